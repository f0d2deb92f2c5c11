"""GPU box: time the 3x3 bf16 weight gradient on the benchmark's shapes (B = 8, 540x960) with the strip kernel and with the
previous kernel (NVQ_WGRAD_STRIP=0 / 1 in the environment select them), and compare their results."""
import os, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, "continual-learning-for-dynamic-video-quality-enhancement_amd"))
import torch
from nerve_cl import _nvq as K

N, H, W = int(os.environ.get("WG_N", 8)), int(os.environ.get("WG_H", 540)), int(os.environ.get("WG_W", 960))
ws = torch.empty(K.wgrad_workspace_bytes() // 4 + 1024, device="cuda")


def timeit(fn, n=10):
    for _ in range(2): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


shapes = [(64, 32), (96, 32), (128, 32), (160, 32), (192, 32), (192, 64), (128, 64), (96, 128)]
for cin, cout in shapes:
    n = N * 2 if cout == 128 else N
    g = torch.Generator(device="cuda").manual_seed(cin * 7 + cout)
    cat = K.CatBuf("cuda", n, H, W, 64, 5, 256, torch.bfloat16, planar=True) if cin <= 224 and cout == 32 else None
    if cat is not None:
        cat.lead.copy_(torch.randn(cat.lead.shape, device="cuda", generator=g) * 0.5)
        for sl in cat.slices:
            sl.copy_(torch.randn(sl.shape, device="cuda", generator=g) * 0.5)
        x = cat.inp(cin)
    else:
        xt = (torch.randn(n, H, W, cin, device="cuda", generator=g) * 0.5).bfloat16()
        x = K.Sl(xt)
    dy = (torch.randn(n, H, W, cout, device="cuda", generator=g) * 0.1).bfloat16()
    outs = {}
    for old in ("1", "", "d1", "d2"):
        os.environ["NVQ_WGRAD_STRIP"] = "0" if old == "1" else "1"
        os.environ.pop("NVQ_WS_DBG", None)
        if old.startswith("d"): os.environ["NVQ_WS_DBG"] = old[1:]
        dw, db = torch.empty(cout, cin, 3, 3, device="cuda"), torch.empty(cout, device="cuda")
        run = lambda: K.conv_wgrad(x, cin, K.Sl(dy), dw, db, ws, 3, math=K.MATH_BF16)
        t = timeit(run)
        outs[old] = (t, dw.clone(), db.clone())
    (t0, w0, b0), (t1, w1, b1) = outs["1"], outs[""]
    nbytes = n * H * W * (cin + cout) * 2
    rel = ((w1 - w0).abs().max() / w0.abs().max()).item()
    relb = ((b1 - b0).abs().max() / b0.abs().max()).item()
    print(f"cin{cin:4d} cout{cout:4d} n{n}: old {t0:7.1f} us ({nbytes/t0/1e6:5.2f} TB/s)  strip {t1:7.1f} us ({nbytes/t1/1e6:5.2f} TB/s)  "
          f"x{t0/t1:.2f}  no-mfma {outs['d1'][0]:7.1f}  no-dma {outs['d2'][0]:7.1f}   max rel diff dw {rel:.1e} db {relb:.1e}", flush=True)
