"""GPU box: where does the time of one bf16 conv launch go?  Times the same launch normally, with the MFMA section
skipped (staging only) and with the per-chunk loads skipped (compute only)."""
import os, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
# the diagnostic build (NVQ_DEBUG_TOOLS=1 bash build.sh): the shipped libnvq.so has no nvq_debug_* entry points
os.environ.setdefault("NVQ_LIB", os.path.join(REPO, "continual-learning-for-dynamic-video-quality-enhancement_amd", "libnvq_debug.so"))
sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, "continual-learning-for-dynamic-video-quality-enhancement_amd"))
import torch
from nerve_cl import _nvq as K

def timeit(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3

N, H, W = int(os.environ.get("PH_N", 2)), int(os.environ.get("PH_H", 540)), 960
LDS_ = [int(v) for v in os.environ.get("PH_LD", "256").split(",")]
for cin, cout, k, dt, ld in [(192, 32, 3, torch.bfloat16, l) for l in LDS_] + [(128, 32, 3, torch.bfloat16, l) for l in LDS_] + [
                             (64, 32, 3, torch.bfloat16, LDS_[0]), (192, 32, 3, torch.float32, 224),
                             (224, 64, 3, torch.bfloat16, LDS_[-1]), (224, 64, 1, torch.bfloat16, LDS_[-1])]:
    x = torch.randn(N, H, W, ld, device="cuda").to(dt)
    out = torch.empty(N, H, W, ld, device="cuda", dtype=dt)
    w = torch.randn(cout, cin, k, k, device="cuda") * 0.05
    wp = K.conv_pack(w, False, cin, math=K.MATH_BF16)
    b = torch.zeros(cout, device="cuda")
    def run():
        K.conv_forward(K.Sl(x, cin, 0), wp, b, K.Sl(out, cout, 0), k, relu=True, math=K.MATH_BF16)
    res = []
    modes = (0, 1, 2) + tuple(int(m) for m in os.environ.get("PH_MODES", "").split(",") if m)
    for mode in modes:
        K.lib().nvq_debug_set_conv_mode(mode)
        res.append(timeit(run))
    K.lib().nvq_debug_set_conv_mode(0)
    nbytes = N * H * W * (cin + cout) * (2 if dt == torch.bfloat16 else 4)
    print(f"cin{cin} cout{cout} k{k} ld{ld} {str(dt)[6:]:9s}: normal {res[0]:7.1f} us ({nbytes/res[0]/1e6:6.2f} TB/s alg)  "
          f"no-mfma {res[1]:7.1f} us  no-loads {res[2]:7.1f} us  " + "  ".join(f"mode{m} {r:7.1f}" for m, r in zip(modes[3:], res[3:])))
