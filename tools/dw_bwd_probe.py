"""Time nvq_dwconv_backward (and the two launches it replaces) alone at the cfg2 extractor shape: N = 24 frames of 540 x 960,
64 channels.  usage: python tools/dw_bwd_probe.py [--iters 20]"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..",
                                "continual-learning-for-dynamic-video-quality-enhancement_amd"))
from nerve_cl import _nvq as K  # noqa: E402
from dwpw_probe import timed  # noqa: E402


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
    dy = torch.randn(N, a.H, a.W, C, device=dev).bfloat16()
    wd = torch.randn(C, 1, 3, 3, device=dev) * 0.3
    dx, dw = torch.empty_like(x), torch.empty(C, 1, 3, 3, device=dev)
    mean, invstd = torch.empty(a.G, C, device=dev), torch.empty(a.G, C, device=dev)
    ws = torch.empty(K.wgrad_workspace_bytes() // 4 + (1 << 20), device=dev)
    gamma, beta = torch.ones(C, device=dev), torch.zeros(C, device=dev)
    K.bn_stats(x, a.B, list(range(a.G)), mean, invstd, None, None, ws)
    bn = (mean, invstd, gamma, beta, a.B)
    add = torch.randn(N, a.H, a.W, C, device=dev)
    mask = torch.randn(N, a.H, a.W, C, device=dev).bfloat16()
    unit = N * a.H * a.W * C * 2 / 1e9
    for name, b, ad, mk, units in (("plain", None, None, None, 3), ("bn-in", bn, None, None, 3), ("epilogue", None, add, mask, 6)):
        for var in os.environ.get("VARS", "").split(",") if os.environ.get("VARS") else [None]:
            if var is not None:
                os.environ["NVQ_DWB_RU"] = var
            ms = timed(lambda: K.dwconv_backward(x, b, dy, wd, dx, dw, ws, add=ad, mask=mk), a.iters)
            print(f"dwconv_backward {name} {var or ''}: {ms:.3f} ms  ({units * unit / ms:.2f} TB/s over {units} bf16 tensor passes)")

        def two():
            K.dwconv_wgrad(x, dy, dw, ws, bn=b)
            K.dwconv_forward(dy, wd, dx, flip=True, add=ad, mask=mk)
        ms = timed(two, a.iters)
        print(f"two launches {name}: {ms:.3f} ms")


if __name__ == "__main__":
    main()
