"""GPU box (development): s_memtime stamps of workgroup 0 of the all-ci weight-gradient kernel (variant bit 7 = dbg 8)."""
import os, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, "continual-learning-for-dynamic-video-quality-enhancement_amd"))
import torch, numpy as np
from nerve_cl import _nvq as K
N, H, W, cin = 8, 540, 960, int(os.environ.get("CIN", 128))
dev = torch.device("cuda")
cat = K.CatBuf(dev, N, H, W, 64, 5, 224, torch.bfloat16, True)
cat.flat.copy_(torch.randn(cat.flat.numel(), device=dev).clamp_(-3, 3).to(torch.bfloat16))
dy = (torch.randn(N, H, W, 32, device=dev) * 0.1).to(torch.bfloat16)
ws = torch.zeros(K.wgrad_workspace_bytes() // 4 + 8 * 2048 * 2, device=dev)
OFF = K.wgrad_workspace_bytes() // 4
dw = torch.zeros(32, cin, 3, 3, device=dev); db = torch.zeros(32, device=dev)
for _ in range(3):
    K.conv_wgrad(cat.inp(cin), cin, K.Sl(dy), dw, db, ws, 3, math=K.MATH_BF16, variant=128)
torch.cuda.synchronize()
st = ws[OFF:OFF + 8 * 2048 * 2].cpu().numpy().view(np.uint64).reshape(8, 2048)
t0 = st[st > 0].min()
cons = (st[0] - t0).astype(np.int64)   # consumer wave 0: (before barrier, after barrier) per unit
prod = (st[6] - t0).astype(np.int64)   # producer wave 6: (after vm_wait, after write, after load, after barrier) per iteration
print("per unit and wave: barrier -> staging start (last MFMA batch) | wait for loads | stores + loads | staging end -> next barrier (3 MFMA batches) | wait at barrier")
for u in range(20, 28):
    row = []
    for w in range(8):
        c = (st[w] - t0).astype(np.int64)
        b0, b1, s0, s1, s2 = c[5 * u: 5 * u + 5]      # unit u: [before barrier, after], [staging start, loads waited, staged]
        nb0 = c[5 * u + 5]
        row.append(f"{s0 - b1:5d}|{s1 - s0:4d}|{s2 - s1:4d}|{nb0 - s2:5d}|{b1 - b0:5d}")
    print(f"  unit {u}: " + "   ".join(row))
