"""GPU box, with a library built from an INSTRUMENTED copy of csrc/conv_bf16.hip: where does a tile iteration of
wgrad_bf16_kernel go?  The instrumentation (not shipped: it hijacks the bias-gradient pointer as the stamp buffer) is, inside the
kernel's tile loop,

    long long* stamps = d.dbias ? reinterpret_cast<long long*>(d.dbias + 64) : nullptr;
    const bool stamping = stamps && (blockIdx.x == 0 || blockIdx.x == 37) && blockIdx.y == 0 && blockIdx.z == 0 && (tid & 63) == 0;
    auto stamp = [&](int k) { if (stamping && it < 32)
        stamps[(((blockIdx.x == 0 ? 0 : 1) * 4 + (tid >> 6)) * 32 + it) * 8 + k] = __builtin_readcyclecounter(); };

with stamp(0) at the loop top, (1) after the first barrier, (2) after commit(), (3) after the second barrier, (4) after fetch(),
(5) after the MFMA section, `it` counting iterations, and the bias-partial block at the end of the kernel disabled.  Link it into a
copy of libnvq.so and run  NVQ_LIB=<that .so> python tools/wgrad_stamps.py  (results: DESIGN.md section 5)."""
import os, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, "continual-learning-for-dynamic-video-quality-enhancement_amd"))
import numpy as np
import torch
from nerve_cl import _nvq as K, _engine

N, H, W = 8, 540, 960
ws = _engine.workspace(torch.device("cuda"))
for cin, cout in [(192, 32), (64, 32)]:
    x = torch.randn(N, H, W, 256, device="cuda").bfloat16()
    dy = torch.randn(N, H, W, 64, device="cuda").bfloat16()
    dw = torch.empty(cout, cin, 3, 3, device="cuda")
    stamps = torch.zeros(64 + 2 * 4 * 32 * 8 * 2, device="cuda")
    for _ in range(3):
        stamps.zero_()
        K.conv_wgrad(K.Sl(x, cin, 0), cin, K.Sl(dy, cout, 0), dw, stamps, ws, 3, math=K.MATH_BF16)
    torch.cuda.synchronize()
    t = stamps[64:].cpu().numpy().view(np.int64).reshape(2, 4, 32, 8)[..., :6]
    print(f"cin{cin}: stamps per tile iteration (shader cycles), median over iterations 2..: ")
    names = ["barrier1", "commit", "barrier2", "fetch issue", "MFMA section", "(loop back)"]
    for b in range(2):
        for w in range(4):
            tt = t[b, w]
            nit = int((tt[:, 0] > 0).sum())
            if nit < 4:
                continue
            d = np.diff(tt[2:nit], axis=1)                      # phases inside an iteration
            loop = tt[3:nit, 0] - tt[2:nit - 1, 5]
            total = tt[3:nit, 0] - tt[2:nit - 1, 0]
            med = np.median(d, axis=0)
            print(f"  block {'0' if b == 0 else '37'} wave {w}: iterations {nit}  " +
                  "  ".join(f"{n} {int(m)}" for n, m in zip(names[:5], med)) +
                  f"  | iteration {int(np.median(total))}")
