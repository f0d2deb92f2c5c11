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
print("consumer waves: per unit, cycles from leaving barrier u to arriving at barrier u + 1 | wait at the barrier")
for u in range(20, 30):
    row = []
    for w in range(6):
        c = (st[w] - t0).astype(np.int64)
        a, b = c[2 * u], c[2 * u + 1]
        row.append(f"{a - c[2 * u - 1]:5d}|{b - a:5d}")
    print(f"  unit {u}: " + "  ".join(row))
print("producer wave 6: per iteration [vm_wait done, writes issued, loads issued, barrier left]")
for i in range(20, 32):
    w, x, y, z = prod[4 * i:4 * i + 4]
    print(f"  it {i}: wait-done {w:8d}  write {x - w:6d}  load {y - x:6d}  barrier {z - y:6d}   vm_wait took {w - prod[4 * i - 1]:6d}")
