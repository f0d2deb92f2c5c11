"""GPU box: 3x3 weight gradient of the dense layers (cout = 32, slice-planar bf16 x, bf16 dy): the (pixel split, ci chunk, co
chunk) kernel (variant 1) against the all-input-channel kernel of wgrad_m32.hip (variant 2 = forced; cin 64 3x3 is not taken by it and is skipped), same launch.
usage: python tools/wgrad_ab.py            (AB_N clips of AB_H x AB_W, default 8 x 540 x 960)"""
import os, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, "continual-learning-for-dynamic-video-quality-enhancement_amd"))
import torch
from nerve_cl import _nvq as K

N, H, W = int(os.environ.get("AB_N", 8)), int(os.environ.get("AB_H", 540)), int(os.environ.get("AB_W", 960))
F = int(os.environ.get("AB_F", 64))
REPS = int(os.environ.get("AB_REPS", 20))
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

cat = K.CatBuf(dev, N, H, W, F, 5, F + 160, torch.bfloat16, True)
cat.flat.copy_(torch.randn(cat.flat.numel(), device=dev).clamp_(-3, 3).to(torch.bfloat16))
dy = (torch.randn(N, H, W, 32, device=dev) * 0.1).to(torch.bfloat16)
ws = torch.empty(K.wgrad_workspace_bytes() // 4, device=dev)
for cin in [F + 32 * i for i in range(5)]:
    if cin < 96: continue
    res, outs = {}, {}
    for v in (1, 2):
        dw = torch.zeros(32, cin, 3, 3, device=dev)
        db = torch.zeros(32, device=dev)
        def run():
            K.conv_wgrad(cat.inp(cin), cin, K.Sl(dy), dw, db, ws, 3, math=K.MATH_BF16, variant=v)
        res[v] = timeit(run)
        outs[v] = (dw.clone(), db.clone())
    nbytes = N * H * W * (cin + 32) * 2
    ddw = ((outs[2][0] - outs[1][0]).abs().max() / outs[1][0].abs().max()).item()
    ddb = ((outs[2][1] - outs[1][1]).abs().max() / outs[1][1].abs().max()).item()
    extra = ""
    for v, name in ((16, "no-mfma"), (64, "no-dma"), (256, "no-setprio")) if os.environ.get("AB_DEBUG") else ():
        dw = torch.zeros(32, cin, 3, 3, device=dev); db = torch.zeros(32, device=dev)
        def run():
            K.conv_wgrad(cat.inp(cin), cin, K.Sl(dy), dw, db, ws, 3, math=K.MATH_BF16, variant=v)
        extra += f"  {name} {timeit(run):7.1f}"
    print(f"cin {cin:3d} -> 32 n{N}: split kernel {res[1]:7.1f} us ({nbytes / res[1] / 1e6:5.2f} TB/s)   all-ci kernel "
          f"{res[2]:7.1f} us ({nbytes / res[2] / 1e6:5.2f} TB/s)  x{res[1] / res[2]:.2f}   max rel diff dw {ddw:.1e} db {ddb:.1e}" + extra,
          flush=True)

# ---- 1x1: the blocks' local feature fusion (cin = F + 160 -> F)
CAT = F + 160
dy1 = (torch.randn(N, H, W, F, device=dev) * 0.1).to(torch.bfloat16)
res, outs = {}, {}
for v in (1, 2):
    dw = torch.zeros(F, CAT, 1, 1, device=dev)
    db = torch.zeros(F, device=dev)
    def run():
        K.conv_wgrad(cat.inp(CAT), CAT, K.Sl(dy1), dw, db, ws, 1, math=K.MATH_BF16, variant=v)
    res[v] = timeit(run)
    outs[v] = (dw.clone(), db.clone())
nbytes = N * H * W * (CAT + F) * 2
ddw = ((outs[2][0] - outs[1][0]).abs().max() / outs[1][0].abs().max()).item()
ddb = ((outs[2][1] - outs[1][1]).abs().max() / outs[1][1].abs().max()).item()
print(f"1x1 cin {CAT} -> {F} n{N}: split kernel {res[1]:7.1f} us ({nbytes / res[1] / 1e6:5.2f} TB/s)   all-ci kernel "
      f"{res[2]:7.1f} us ({nbytes / res[2] / 1e6:5.2f} TB/s)  x{res[1] / res[2]:.2f}   max rel diff dw {ddw:.1e} db {ddb:.1e}", flush=True)
