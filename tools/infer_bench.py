"""GPU box: forward-only (eval, no_grad) throughput of SuperResolutionNet at cfg2 geometry, bf16 mode and exact fp32 mode."""
import os, sys, time
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, "continual-learning-for-dynamic-video-quality-enhancement_amd"))
import torch
from nerve_cl import _nvq
from nerve_cl.models import SuperResolutionNet

B = int(os.environ.get("INF_B", 4))
for mode in ("bf16", "f32"):
    net = SuperResolutionNet(3, 2, 64, 8, 1).cuda().eval()
    if mode == "bf16":
        net.math_mode, net.bf16_activations = _nvq.MATH_BF16, True
    x = torch.rand(B, 3, 3, 540, 960, device="cuda")
    with torch.no_grad():
        for _ in range(2): net(x)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        n = 6
        for _ in range(n): y = net(x)
        torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print(f"{mode}: {B * n / dt:.1f} output frames/s (1080p), {dt / n * 1e3:.1f} ms per batch of {B}")
