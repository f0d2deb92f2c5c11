"""Time nvq_dwpw_forward (and the three launches it replaces) alone at the cfg2 extractor shape: N = 24 frames of 540 x 960,
64 channels.  usage: python tools/dwpw_probe.py [--iters 20]"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..",
                                "continual-learning-for-dynamic-video-quality-enhancement_amd"))
from nerve_cl import _nvq as K  # noqa: E402


def timed(fn, iters):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--iters", type=int, default=20)
    ap.add_argument("--B", type=int, default=8)
    ap.add_argument("--G", type=int, default=3)
    ap.add_argument("--H", type=int, default=540)
    ap.add_argument("--W", type=int, default=960)
    a = ap.parse_args()
    N, C = a.B * a.G, 64
    dev = "cuda"
    x = torch.randn(N, a.H, a.W, C, device=dev).bfloat16()
    wd = torch.randn(C, 1, 3, 3, device=dev) * 0.3
    wp = torch.randn(C, C, 1, 1, device=dev) * 0.1
    d, p = torch.empty_like(x), torch.empty_like(x)
    mean, invstd = torch.empty(a.G, C, device=dev), torch.empty(a.G, C, device=dev)
    ws = torch.empty(K.wgrad_workspace_bytes() // 4 + (1 << 20), device=dev)
    gamma, beta = torch.ones(C, device=dev), torch.zeros(C, device=dev)
    K.bn_stats(x, a.B, list(range(a.G)), mean, invstd, None, None, ws)
    bn = (mean.clone(), invstd.clone(), gamma, beta, a.B)
    order = list(range(a.G))
    unit = N * a.H * a.W * C * 2 / 1e9
    for name, b in (("plain", None), ("bn-in", bn)):
        ms = timed(lambda: K.dwpw_forward(x, b, wd, wp, d, p, a.B, order, mean, invstd, None, None, ws), a.iters)
        print(f"dwpw_forward {name}: {ms:.3f} ms  ({3 * unit / ms:.2f} TB/s over 3 tensor passes)")
        wpk = K.conv_pack(wp, False, C, math=K.MATH_BF16)

        def three():
            K.dwconv_forward(x, wd, d, bn=b)
            K.conv_forward(K.Sl(d), wpk, None, K.Sl(p), 1, math=K.MATH_BF16)
            K.bn_stats(p, a.B, order, mean, invstd, None, None, ws)
        ms = timed(three, a.iters)
        print(f"three launches {name}: {ms:.3f} ms")


if __name__ == "__main__":
    main()
