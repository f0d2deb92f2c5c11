"""Shared bits of the experiment scripts: path setup, PSNR, optional one-process-per-GPU launch."""
import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(REPO, "continual-learning-for-dynamic-video-quality-enhancement_amd")
if PKG not in sys.path:
    sys.path.insert(0, PKG)

import torch  # noqa: E402


def compute_psnr(pred: torch.Tensor, target: torch.Tensor) -> float:
    """20*log10(1/sqrt(mse)) over the whole batch (reference experiments/train_baseline.py:27-32)."""
    mse = torch.mean((pred - target) ** 2)
    if mse == 0:
        return float("inf")
    return 20 * torch.log10(1.0 / torch.sqrt(mse)).item()


def pick_device():
    """'cuda' when a HIP device is visible (reference train_baseline.py:37); the SR path has no CPU mode."""
    from nerve_cl import parallel
    rank, world, local = parallel.init_from_env()
    if not torch.cuda.is_available():
        raise SystemExit("no HIP device visible: nerve_cl's super-resolution path runs only on the GPU")
    torch.cuda.set_device(local)
    return torch.device("cuda", local), rank, world


def make_optimizer(cls, params, **kw):
    """The reference's `torch.optim.Adam / AdamW(model.parameters(), ...)` call (train_continual.py:27,51, train_baseline.py:41)
    with PyTorch's fused single-kernel implementation when every parameter is an fp32 tensor on the GPU: the same update rule;
    the default (foreach) form is ~10 launches and 0.6 ms of host time per step over the network's 131 tensors, which the
    launch-bound 64x64 steps feel."""
    params = list(params)
    if params and all(p.is_cuda and p.dtype == torch.float32 for p in params):
        kw.setdefault("fused", True)
    return cls(params, **kw)


def shard(n: int, rank: int, world: int) -> slice:
    """Contiguous shard of a dataset of n samples for this rank.  Every rank gets the SAME number of samples (the
    remainder n % world is dropped): the gradient all-reduce runs once per training batch inside backward, so ranks with
    different batch counts would wait for each other forever."""
    per = n // world
    return slice(rank * per, (rank + 1) * per)
