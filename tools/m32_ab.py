"""GPU box: the dense-layer 3x3 convs (ResidualDenseBlock layers and their mirror-form gradient convs) on the 16x16x32 kernel
(tile_rows 0 / 16) against the 32x32x16 kernels of conv_m32.hip (tile_rows 162: two rows per wave, 8 waves; 164: four rows
per wave, 4 waves), same launch, same slice-planar buffers, B = AB_N clips of 540 x 960.
usage: python tools/m32_ab.py [variants ...]   (default 16 162 164)"""
import os, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, "continual-learning-for-dynamic-video-quality-enhancement_amd"))
import torch
from nerve_cl import _nvq as K

N, H, W = int(os.environ.get("AB_N", 8)), int(os.environ.get("AB_H", 540)), int(os.environ.get("AB_W", 960))
REPS = int(os.environ.get("AB_REPS", 20))
variants = [int(v) for v in sys.argv[1:]] or [16, 162, 164]
dev = torch.device("cuda")
torch.manual_seed(0)

def timeit(fn, n=REPS):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3

cat = K.CatBuf(dev, N, H, W, 64, 5, 224, torch.bfloat16, True)
cat.flat.copy_(torch.randn(cat.flat.numel(), device=dev).clamp_(-3, 3).to(torch.bfloat16))
bits = torch.zeros(N, H, W, dtype=torch.int32, device=dev)
bits_r = torch.randint(-2**31, 2**31 - 1, (N, H, W), dtype=torch.int32, device=dev)
rows = []
for mode, cins in (("fwd", (64, 96, 128, 160)), ("bwd", (64, 96, 128, 160, 192))):
    for cin in cins:
        w = torch.randn(32, cin, 3, 3, device=dev) * 0.05
        ctr = 0
        if mode == "bwd":
            w[:, :64, :, :] = 0
            w[:, :64, 1, 1] = torch.randn(32, 64, device=dev) * 0.05
            ctr = 64
        wp = K.conv_pack(w, False, cin, math=K.MATH_BF16)
        b = torch.randn(32, device=dev) * 0.1 if mode == "fwd" else None
        outs = {}
        res = {}
        for v in [0] + variants:
            out = torch.zeros(N, H, W, 32, device=dev, dtype=torch.bfloat16)
            bw = torch.zeros_like(bits)
            def run():
                if mode == "fwd":
                    K.conv_forward(cat.inp(cin), wp, b, K.Sl(out), 3, relu=True, math=K.MATH_BF16, bits=bw, bits_mode=1,
                                   tile_rows=v)
                else:
                    K.conv_forward(cat.inp(cin), wp, None, K.Sl(out), 3, math=K.MATH_BF16, bits=bits_r, bits_mode=2,
                                   center_cin=ctr, tile_rows=v)
            try:
                res[v] = timeit(run)
            except RuntimeError:                       # a forced form that does not take this launch
                res[v] = float('nan')
            outs[v] = (out.float(), bw.clone())
        ref = outs[0]
        nbytes = N * H * W * ((cin + 32) * 2 + 4)
        line = f"{mode} cin {cin:3d}: " + "  ".join(
            f"[{v}] {res[v]:7.1f} us ({nbytes / res[v] / 1e6:5.2f} TB/s)" for v in [0] + variants)
        for v in variants:
            dmax = (outs[v][0] - ref[0]).abs().max().item()
            nbits = (outs[v][1] != ref[1]).sum().item()
            line += f"  |d{v}| {dmax:.2e} bits!= {nbits}"
        print(line, flush=True)
