"""HIP-graph replay of the SuperResolutionNet step for launch-bound (small-frame) shapes.

A 64x64 training step of the network is ~700 kernel launches of a few microseconds each (BASELINE configs[0] and [4]):
the device waits for the host.  The schedule of ``nerve_cl._engine.forward`` / ``backward`` is static for a given input shape
and mode (no host reads, no data-dependent launches), so it is captured once into two HIP graphs - one for the forward, one
for the backward, sharing one memory pool so that the backward graph reads the forward graph's saved state in place - and
replayed from then on: one graph launch instead of hundreds of kernel launches.

Measured (tools/graph_bench.py, MI355X, ROCm 7.2): the replayed step takes exactly as long as the eager one (8 clips of
64x64, F=64, 8 blocks: 13.0 ms either way; 128x128: 27.0 vs 26.8 ms), with identical results.  The small-frame step is
bound on the device - by the critical path of ~500 dependent kernels that each fill a quarter of the CUs (DESIGN.md section
5) - so a replay cannot shorten it; what it saves is the host thread (one call per pass instead of ~700 ctypes calls).
It is therefore OFF by default (``net.use_hip_graphs = True`` or ``NVQ_GRAPH=1`` to enable).

Rules (checked, never assumed):
  * parameters and buffers are baked into the graphs by address: an entry is re-captured when any address changes
    (``.to()``, ``.half()``; in-place optimizer / ``load_state_dict`` updates keep addresses);
  * the saved state of a forward lives in the entry, so a second forward of the same shape while the first one still waits
    for its backward (MAML inner loops, gradient checkpointing) runs eagerly instead, and a backward whose state has been
    overwritten by a later replay raises;
  * outputs and gradients handed to autograd are copies: autograd may keep a gradient tensor as ``p.grad``, and the next
    replay would overwrite it;
  * the kernel timer of ``bench.py`` and ``return_intermediate`` use the eager path.
"""
from __future__ import annotations

import weakref
from typing import Dict, Optional, Tuple

import torch

from nerve_cl import _engine, _nvq


class _Token:
    """Lives as long as the autograd node of one forward does."""
    __slots__ = ("__weakref__",)


class _Entry:
    def __init__(self):
        self.ptrs: Tuple[int, ...] = ()
        self.pool = None
        self.fwd: Optional[torch.cuda.CUDAGraph] = None
        self.bwd: Optional[torch.cuda.CUDAGraph] = None
        self.static_in: Optional[torch.Tensor] = None
        self.static_out: Optional[torch.Tensor] = None
        self.static_dout: Optional[torch.Tensor] = None
        self.flat: Optional[torch.Tensor] = None
        self.sv = None
        self.gen = 0                 # replays of the forward graph so far
        self.pending = None          # weakref to the token of the forward whose backward has not run yet
        self.eager_calls = 0


class StepGraphs:
    """Per-module cache {(shape, mode): _Entry}."""
    WARMUP = 2           # eager calls of a shape before it is captured (first-use allocations, autotuned nothing: just caution)
    MAX_ENTRIES = 8
    MAX_RECAPTURES = 2

    def __init__(self):
        self.entries: Dict[tuple, _Entry] = {}
        self.recaptures: Dict[tuple, int] = {}
        self.replays = 0
        self.eager_fallbacks = 0

    # ------------------------------------------------------------------ forward
    def forward(self, net, frames: torch.Tensor, need_grad: bool, act_dtype):
        """-> (out, entry, token, gen) or None when this call has to run eagerly."""
        if _nvq.TIMER is not None:
            return None
        key = (tuple(frames.shape), bool(net.training), net.math_mode, act_dtype, bool(need_grad), frames.device.index)
        P = net._tensor_dict()
        ptrs = tuple(t.data_ptr() for t in P.values())
        e = self.entries.get(key)
        if e is not None and e.ptrs != ptrs and e.fwd is not None:
            del self.entries[key]                              # parameters were re-allocated: capture again ...
            self.recaptures[key] = self.recaptures.get(key, 0) + 1
            e = None
        if e is None:
            # ... unless they keep moving (functional "fast weights"): such a caller stays on the eager path
            if len(self.entries) >= self.MAX_ENTRIES or self.recaptures.get(key, 0) > self.MAX_RECAPTURES:
                self.eager_fallbacks += 1
                return None
            e = self.entries[key] = _Entry()
        if e.fwd is None:
            if e.eager_calls < self.WARMUP:
                e.eager_calls += 1
                return None
            self._capture_forward(e, net, P, ptrs, frames, need_grad, act_dtype)
        elif e.pending is not None and e.pending() is not None:
            self.eager_fallbacks += 1                          # the previous forward's state is still needed
            return None
        e.static_in.copy_(frames)
        e.fwd.replay()
        e.gen += 1
        self.replays += 1
        token = None
        if need_grad:
            token = _Token()
            e.pending = weakref.ref(token)
        return e.static_out.clone(), e, token, e.gen

    def _capture_forward(self, e: _Entry, net, P, ptrs, frames, need_grad, act_dtype):
        e.ptrs = ptrs
        e.pool = torch.cuda.graph_pool_handle()
        e.static_in = frames.clone()
        g = torch.cuda.CUDAGraph()
        torch.cuda.synchronize()
        with torch.cuda.graph(g, pool=e.pool):
            out, sv = _engine.forward(P, e.static_in, net._F, net._NB, net.scale_factor, net.training, net.math_mode, act_dtype)
            if not need_grad:
                sv = None
        e.fwd, e.static_out, e.sv = g, out, sv

    # ------------------------------------------------------------------ backward
    def backward(self, net, e: _Entry, gen: int, dout: torch.Tensor):
        """-> (flat gradient bucket, views): copies, safe to hand to autograd."""
        if gen != e.gen:
            raise RuntimeError("SuperResolutionNet backward: the saved state of this forward was overwritten by a later forward "
                               "of the same shape (HIP-graph mode keeps one state per shape; set net.use_hip_graphs = False "
                               "for this access pattern)")
        dout = dout.contiguous().float()
        if e.bwd is None:
            e.static_dout = dout.clone()
            g = torch.cuda.CUDAGraph()
            torch.cuda.synchronize()
            with torch.cuda.graph(g, pool=e.pool):
                flat, views = net._new_grad_bucket()
                _engine.backward(net._tensor_dict(), e.sv, e.static_dout, views)
            e.bwd, e.flat = g, flat
            e.sv = None                                        # the graphs own the state now
        else:
            e.static_dout.copy_(dout)
        e.bwd.replay()
        e.pending = None
        flat = e.flat.clone()
        return flat, net._bucket_views(flat)
