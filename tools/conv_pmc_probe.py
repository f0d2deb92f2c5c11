"""GPU box, under `rocprofv3 --pmc ...`: a few launches of the dominant conv shapes (no debug modes), so that the
per-dispatch counters of conv_bf16_kernel can be read from the counter_collection csv."""
import os, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, "continual-learning-for-dynamic-video-quality-enhancement_amd"))
import torch
from nerve_cl import _nvq as K

N, H, W = 2, 540, 960
for cin, cout, k in ((192, 32, 3), (224, 64, 3), (224, 64, 1)):
    x = torch.randn(N, H, W, 256, device="cuda").bfloat16()
    out = torch.empty(N, H, W, 256, device="cuda", dtype=torch.bfloat16)
    w = torch.randn(cout, cin, k, k, device="cuda") * 0.05
    wp = K.conv_pack(w, False, cin, math=K.MATH_BF16)
    b = torch.zeros(cout, device="cuda")
    for _ in range(6):
        K.conv_forward(K.Sl(x, cin, 0), wp, b, K.Sl(out, cout, 0), k, relu=True, math=K.MATH_BF16)
    dy = torch.randn(N, H, W, 256, device="cuda").bfloat16()
    dw = torch.empty(cout, cin, k, k, device="cuda"); db = torch.empty(cout, device="cuda")
    ws = torch.empty(K.wgrad_workspace_bytes() // 4 + (1 << 20), dtype=torch.float32, device="cuda")
    for _ in range(6):
        K.conv_wgrad(K.Sl(x, cin, 0), cin, K.Sl(dy, cout, 0), dw, db, ws, k, math=K.MATH_BF16)
torch.cuda.synchronize()
