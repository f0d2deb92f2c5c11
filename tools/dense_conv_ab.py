"""GPU box: the dense-layer 3x3 convs (forward with one-bit masks, mirror-form gradient) at the benchmark's size, the
persistent kernel (default) against the one-tile-per-workgroup kernel (nvq_conv_desc::tile_rows = 16): time and equality."""
import os, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, "continual-learning-for-dynamic-video-quality-enhancement_amd"))
import torch
from nerve_cl import _nvq as K

N, H, W = int(os.environ.get("AB_N", 8)), int(os.environ.get("AB_H", 540)), int(os.environ.get("AB_W", 960))


def timeit(fn, n=10):
    for _ in range(2): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


g = torch.Generator(device="cuda").manual_seed(1)
cat = K.CatBuf("cuda", N, H, W, 64, 5, 256, torch.bfloat16, planar=True)
cat.lead.copy_(torch.randn(cat.lead.shape, device="cuda", generator=g) * 0.5)
for sl in cat.slices:
    sl.copy_(torch.randn(sl.shape, device="cuda", generator=g) * 0.5)
bits = torch.zeros(N, H, W, dtype=torch.int32, device="cuda")
for mode in ("fwd", "bwd"):
    for i in range(5):
        cin = 64 + 32 * i
        if mode == "bwd" and i == 0:
            continue
        w = torch.randn(32, cin, 3, 3, device="cuda", generator=g) * 0.05
        ctr = 0
        if mode == "bwd":
            ctr = 64
            centre = w[:, :64, 1, 1].clone(); w[:, :64] = 0; w[:, :64, 1, 1] = centre
        wp = K.conv_pack(w, False, cin, math=K.MATH_BF16)
        b = torch.randn(32, device="cuda", generator=g) * 0.1
        res = {}
        for rows in (16, 0):
            if mode == "fwd":
                run = lambda: K.conv_forward(cat.inp(cin), wp, b, cat.y(4 if i < 4 else 3) if False else cat.y(i), 3, relu=True, math=K.MATH_BF16,
                                             bits=bits, bits_mode=1, tile_rows=rows)
            else:
                run = lambda: K.conv_forward(cat.inp(cin), wp, None, cat.y(i), 3, math=K.MATH_BF16, bits=bits, bits_mode=2,
                                             center_cin=ctr, tile_rows=rows)
            t = timeit(run)
            res[rows] = (t, cat.slices[i].clone(), bits.clone())
        (t0, o0, b0), (t1, o1, b1) = res[16], res[0]
        same = torch.equal(o0, o1) and torch.equal(b0, b1)
        md = (o0.float() - o1.float()).abs().max().item()
        nbytes = N * H * W * ((cin + 32) * 2 + 4)
        print(f"{mode} cin{cin:4d}: one-tile {t0:7.1f} us ({nbytes/t0/1e6:5.2f} TB/s)  persistent {t1:7.1f} us ({nbytes/t1/1e6:5.2f} TB/s)  x{t0/t1:.2f}  "
              f"equal {same} (max diff {md:.2e})", flush=True)
