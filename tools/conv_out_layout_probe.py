"""GPU box: the cout-32 3x3 bf16 conv writing its 64-byte pixel rows (a) into a slice of the 256-channel concat buffer
(half of a 128-B line per pixel, the other half belongs to another layer) and (b) into a compact 32-channel tensor (two
pixels per line, whole lines written).  Run plainly for the timings; run under `rocprofv3 --kernel-trace --pmc FETCH_SIZE`
(and WRITE_SIZE) to compare the HBM traffic of the dispatches: 12 dispatches, alternating (a), (b)."""
import os, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, "continual-learning-for-dynamic-video-quality-enhancement_amd"))
import torch
from nerve_cl import _nvq as K

N, H, W, cin = 8, 540, 960, 192
x = torch.randn(N, H, W, 256, device="cuda").bfloat16()
cat_out = torch.empty(N, H, W, 256, device="cuda", dtype=torch.bfloat16)
flat_out = torch.empty(N, H, W, 32, device="cuda", dtype=torch.bfloat16)
w = torch.randn(32, cin, 3, 3, device="cuda") * 0.05
wp = K.conv_pack(w, False, cin, math=K.MATH_BF16)
b = torch.zeros(32, device="cuda")
runs = {"a": lambda: K.conv_forward(K.Sl(x, cin, 0), wp, b, K.Sl(cat_out, 32, 192), 3, relu=True, math=K.MATH_BF16),
        "b": lambda: K.conv_forward(K.Sl(x, cin, 0), wp, b, K.Sl(flat_out), 3, relu=True, math=K.MATH_BF16)}
for _ in range(6):
    runs["a"](); runs["b"]()
torch.cuda.synchronize()
if os.environ.get("PROBE_TIME", "1") == "1":
    for name, fn in runs.items():
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20): fn()
        e1.record(); torch.cuda.synchronize()
        print(f"({name}) {e0.elapsed_time(e1) / 20 * 1e3:7.1f} us per launch", flush=True)
    print("outputs equal:", torch.equal(cat_out[..., 192:224], flat_out))
