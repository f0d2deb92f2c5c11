"""GPU box: does the bf16 mode drift from the exact-fp32 mode over a longer run?  Same initial weights, same data stream,
AdamW, 60 steps at 64x64 (F=32, 4 blocks); prints the two loss curves and the final eval-mode PSNR between the two nets."""
import os, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, "continual-learning-for-dynamic-video-quality-enhancement_amd"))
import copy, math, torch
import torch.nn.functional as F
from nerve_cl import _nvq
from nerve_cl.models import SuperResolutionNet

torch.manual_seed(0)
base = SuperResolutionNet(3, 2, 32, 4, 1).cuda().train()
nets = {"fp32": base, "bf16": copy.deepcopy(base)}
nets["bf16"].math_mode, nets["bf16"].bf16_activations = _nvq.MATH_BF16, True
opts = {k: torch.optim.AdamW(n.parameters(), lr=2e-4, weight_decay=1e-5) for k, n in nets.items()}
g = torch.Generator(device="cuda").manual_seed(1)
curves = {k: [] for k in nets}
for step in range(60):
    hr = torch.rand(8, 3, 128, 128, device="cuda", generator=g)
    hr = F.avg_pool2d(hr, 5, 1, 2)                                   # some spatial structure
    lr = F.avg_pool2d(hr, 2)
    clip = torch.stack([torch.roll(lr, s, 3) for s in (-1, 0, 1)], 1)
    for k, n in nets.items():
        opts[k].zero_grad()
        loss = F.mse_loss(n(clip), hr)
        loss.backward()
        opts[k].step()
        curves[k].append(loss.item())
for k in nets:
    print(k, " ".join(f"{v:.5f}" for v in curves[k][::6]), f"... last {curves[k][-1]:.6f}")
rel = max(abs(a - b) / a for a, b in zip(curves["fp32"], curves["bf16"]))
for n in nets.values():
    n.eval()
with torch.no_grad():
    oa, ob = nets["fp32"](clip), nets["bf16"](clip)
psnr = 10 * math.log10(1.0 / max((oa - ob).pow(2).mean().item(), 1e-12))
print(f"largest relative loss difference over 60 steps: {rel:.2e}; eval PSNR between the two trained nets: {psnr:.1f} dB")
