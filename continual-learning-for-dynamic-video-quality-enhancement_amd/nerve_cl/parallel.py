"""Data parallelism for the SR hot path: one process per GPU, RCCL over xGMI.

The reference is single-process (no torch.distributed anywhere); this is
the one parallel strategy the path needs.  Every rank holds a full replica (8 MB of
parameters), runs forward/backward on its own shard of the minibatch, and the backward of
``SuperResolutionNet`` hands its flat gradient bucket (one contiguous fp32 tensor, ~8 MB) to
ONE all-reduce; the result is divided by the world size (mean of per-rank batch-mean losses =
loss over the global batch).  The message is small, so the collective is latency-bound and
sits after the last gradient kernel on the same stream; nothing is bucketed or overlapped
because there is nothing to overlap with.

``torch.distributed`` with backend "nccl" is RCCL on ROCm; "gloo" runs the same code on CPU
tensors (used by the world_size-2 tests).
"""
from __future__ import annotations

import os
from typing import Dict, Iterable, Optional

import torch
import torch.distributed as dist


def init_from_env(backend: Optional[str] = None) -> "tuple[int, int, int]":
    """Join the process group described by RANK / WORLD_SIZE / LOCAL_RANK / MASTER_* (the
    variables torch.distributed.run exports).  Returns (rank, world, local_rank)."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        if backend == "nccl":
            # RCCL: one device per rank, bound BEFORE the first allocation; the communicator is created lazily on the
            # current device by the first collective (barrier() names the device explicitly)
            torch.cuda.set_device(local)
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, world, local


def barrier(group=None) -> None:
    """dist.barrier(); under RCCL with this rank's own device named (otherwise the barrier picks `rank % device_count`
    and warns, and a rank that has not bound its device would create a context on GPU 0)."""
    if world_size(group) <= 1:
        return
    if dist.get_backend(group) == "nccl":
        dist.barrier(group=group, device_ids=[torch.cuda.current_device()])
    else:
        dist.barrier(group=group)


def world_size(group=None) -> int:
    return dist.get_world_size(group) if dist.is_available() and dist.is_initialized() else 1


def allreduce_mean_(flat: torch.Tensor, group=None) -> torch.Tensor:
    """In-place mean over ranks of one flat bucket (a single collective)."""
    w = world_size(group)
    if w > 1:
        dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=group)
        flat.div_(w)
    return flat


def allreduce_sum_(flat: torch.Tensor, group=None) -> torch.Tensor:
    if world_size(group) > 1:
        dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=group)
    return flat


def broadcast_state_(module: torch.nn.Module, src: int = 0, group=None) -> None:
    """Make every replica identical to rank `src` (parameters and buffers)."""
    if world_size(group) <= 1:
        return
    # one collective per (dtype, device) instead of one per tensor (140 for the SR net): parameters and floating-point
    # buffers travel as one flat fp32 tensor, the BatchNorm step counters as one int64 tensor
    groups: Dict[tuple, list] = {}
    for t in list(module.parameters()) + list(module.buffers()):
        groups.setdefault((t.dtype, t.device), []).append(t.data)
    for tensors in groups.values():
        if len(tensors) == 1:
            dist.broadcast(tensors[0], src=src, group=group)
            continue
        flat = flatten(tensors)
        dist.broadcast(flat, src=src, group=group)
        unflatten_into_(flat, tensors)


def flatten(tensors: Iterable[torch.Tensor]) -> torch.Tensor:
    return torch.cat([t.reshape(-1) for t in tensors])


def unflatten_into_(flat: torch.Tensor, tensors: Iterable[torch.Tensor]) -> None:
    off = 0
    for t in tensors:
        n = t.numel()
        t.copy_(flat[off:off + n].view_as(t))
        off += n


def enable_data_parallel(module: torch.nn.Module, group=None, broadcast: bool = True) -> torch.nn.Module:
    """Turn on gradient all-reduce for every bucketed network (SuperResolutionNet, LightweightSuperResolution,
    FrameRecoveryNet) inside `module`.

    After this call ``loss.backward()`` leaves rank-averaged gradients in ``.grad`` exactly
    where a single-process run would leave them, so the reference training loops need no change.
    (``EWC.compute_fisher`` switches the hook off while it runs: the Fisher needs per-rank gradients.)"""
    from nerve_cl._bucket import BucketedNet
    if broadcast:
        broadcast_state_(module, 0, group)
    for m in module.modules():
        if isinstance(m, BucketedNet):
            m._grad_bucket_hook = (lambda flat, g=group: allreduce_mean_(flat, g))
    return module


def allreduce_scalars(values, group=None, device=None) -> "list[float]":
    """Sum a few python floats over the ranks in one collective (rank-uniform decisions in the training scripts)."""
    if world_size(group) <= 1:
        return [float(v) for v in values]
    t = torch.tensor([float(v) for v in values], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)
    return t.tolist()


def average_bn_buffers_(module: torch.nn.Module, group=None) -> None:
    """Average BatchNorm running statistics across ranks (done at checkpoint time; training
    itself uses per-rank batch statistics, i.e. standard DDP semantics)."""
    w = world_size(group)
    if w <= 1:
        return
    bufs = [b for _, b in module.named_buffers() if b.dtype.is_floating_point]
    if not bufs:
        return
    flat = flatten([b.detach().float() for b in bufs])       # one collective for all of them
    dist.all_reduce(flat, group=group)
    flat.div_(w)
    unflatten_into_(flat, [b.data for b in bufs])
