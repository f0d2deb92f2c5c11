"""Time the bf16 correlation kernels alone at the cfg2 shape: 16 reference images of 540 x 960 against 8 centre images,
64 channels, bf16 features and volumes.  usage: python tools/corr_probe.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..",
                                "continual-learning-for-dynamic-video-quality-enhancement_amd"))
from nerve_cl import _nvq as K  # noqa: E402
from dwpw_probe import timed  # noqa: E402

B, R, H, W, C = 8, 2, 540, 960, 64
N = B * R
dev = "cuda"
oth = torch.randn(N, H, W, C, device=dev).bfloat16()
cen = torch.randn(B, H, W, C, device=dev).bfloat16()
corr = torch.empty(N, H, W, 128, device=dev, dtype=torch.bfloat16)
dcorr = torch.randn(N, H, W, 128, device=dev).bfloat16()
d_oth = torch.zeros(N, H, W, C, device=dev)
d_cen = torch.zeros(B, H, W, C, device=dev)
ms = timed(lambda: K.correlation_forward(K.Sl(oth), K.Sl(cen), corr, math=K.MATH_BF16), 20)
print(f"correlation_forward: {ms:.3f} ms")
ms = timed(lambda: K.correlation_backward(1, dcorr, K.Sl(cen), K.Sl(d_oth), True, math=K.MATH_BF16), 20)
print(f"correlation_backward(1): {ms:.3f} ms")
ms = timed(lambda: K.correlation_backward(2, dcorr, K.Sl(oth), K.Sl(d_cen), True, math=K.MATH_BF16, groups=R), 20)
print(f"correlation_backward(2, groups={R}): {ms:.3f} ms")
