"""Flat parameter / gradient buckets shared by the HIP-backed networks (SuperResolutionNet,
LightweightSuperResolution, FrameRecoveryNet).

Every such network is ONE autograd node whose backward writes all parameter gradients into one flat fp32 bucket
(16-byte aligned slots).  That bucket is

  * what a data-parallel run all-reduces (``nerve_cl.parallel`` installs ``_grad_bucket_hook``),
  * what ``EWC.compute_fisher`` squares and accumulates (``_last_grad_bucket``: no per-tensor ``cat``),
  * where the EWC penalty gradient ``lambda * F * (theta - theta*)`` is added by ONE kernel after the all-reduce
    (``_deferred_adds``, filled by the penalty's autograd node, drained by the network's backward) instead of 131
    ``AccumulateGrad`` additions (reference nerve_cl/continual/ewc.py:225-232 + experiments/train_continual.py:56-57).

For the last point the parameters themselves must be readable as one flat tensor in the same layout: ``flat_theta()``
re-homes every ``param.data`` as a view of one persistent buffer (checked by address on every call, rebuilt after
``.to()`` / ``deepcopy``; in-place optimizer and ``load_state_dict`` updates keep it valid).
"""
from __future__ import annotations

import weakref
from typing import Dict, List, Optional, Tuple

import torch
import torch.nn as nn

from nerve_cl import _nvq


class _Token:
    """lives exactly as long as the autograd node of one forward does"""
    __slots__ = ("__weakref__",)


class BucketedNet(nn.Module):
    """Mixin-style base: bucket layout, gradient bucket, flat parameter view, deferred bucket additions."""

    def _init_bucket(self) -> None:
        self._param_names: List[str] = [n for n, _ in self.named_parameters()]
        self._grad_bucket_hook = None          # data-parallel all-reduce (nerve_cl.parallel)
        self._last_grad_bucket: Optional[torch.Tensor] = None
        self._theta_flat: Optional[torch.Tensor] = None
        self._layout: Optional[Tuple[Dict[str, Tuple[int, int]], int]] = None
        self._pending = None                   # weakref to the token of the latest forward-with-grad (see _mark_awaiting)
        self._deferred_adds: list = []         # (lam, star_flat, fisher_flat, scale_dev) to add into the next bucket

    # ------------------------------------------------------------------ layout
    def _slots(self):
        """([(name, owner._parameters, key)], [(name, owner._buffers, key)]) in named_parameters() / named_buffers() order, built
        once.  Every forward and backward needs all ~180 tensors by name; walking the module tree for them (named_parameters +
        named_buffers, ~150 modules) cost 1 ms of the 4 ms continual-learning step.  The owner dicts are looked up per call, so
        a parameter or buffer that is REPLACED (load_state_dict(assign=True), module.to() with swapped tensors) is still
        found; adding or removing sub-modules after construction is not supported by the engine anyway."""
        c = self.__dict__.get("_slot_cache")
        if c is None:
            def locate(name: str, kind: str):
                path, _, key = name.rpartition(".")
                mod = self.get_submodule(path) if path else self
                return (name, getattr(mod, kind), key)
            c = ([locate(n, "_parameters") for n, _ in self.named_parameters()],
                 [locate(n, "_buffers") for n, _ in self.named_buffers()])
            self.__dict__["_slot_cache"] = c
        return c

    def _named_params(self):
        """list(self.named_parameters()) without the module-tree walk"""
        return [(n, d[k]) for n, d, k in self._slots()[0]]

    def _tensor_dict(self) -> Dict[str, torch.Tensor]:
        ps, bs = self._slots()
        d = {n: o[k].data for n, o, k in ps}
        for n, o, k in bs:
            d[n] = o[k]
        return d

    def _bucket_layout(self) -> "Tuple[Dict[str, Tuple[int, int]], int]":
        """{name: (offset, numel)} with 16-byte aligned offsets, total floats."""
        if self._layout is None:
            lay, off = {}, 0
            for n, p in self.named_parameters():
                lay[n] = (off, p.numel())
                off += (p.numel() + 3) // 4 * 4
            self._layout = (lay, off)
        return self._layout

    def _bucket_views(self, flat: torch.Tensor) -> Dict[str, torch.Tensor]:
        lay, _ = self._bucket_layout()
        shapes = {n: p.shape for n, p in self._named_params()}
        return {n: flat[o:o + k].view(shapes[n]) for n, (o, k) in lay.items()}

    def _new_grad_bucket(self):
        _, total = self._bucket_layout()
        dev = self._named_params()[0][1].device
        flat = torch.zeros(total, dtype=torch.float32, device=dev)
        return flat, self._bucket_views(flat)

    # ------------------------------------------------------------------ flat parameters
    def flat_theta(self) -> torch.Tensor:
        """All parameters as one flat fp32 tensor in bucket layout (padding = 0); the parameters are views of it."""
        lay, total = self._bucket_layout()
        flat = self._theta_flat
        named = self._named_params()
        dev = named[0][1].device
        ok = flat is not None and flat.device == dev
        if ok:
            base = flat.data_ptr()
            for n, p in named:
                if p.dtype != torch.float32 or p.data_ptr() != base + 4 * lay[n][0] or not p.is_contiguous():
                    ok = False
                    break
        if not ok:
            flat = torch.zeros(total, dtype=torch.float32, device=dev)
            with torch.no_grad():
                for n, p in named:
                    o, k = lay[n]
                    v = flat[o:o + k].view(p.shape)
                    v.copy_(p.data)
                    p.data = v
            self._theta_flat = flat
        return flat

    # ------------------------------------------------------------------ "is my backward still to come?"
    def _mark_awaiting(self, ctx) -> None:
        """Called by the network's autograd Function in forward when gradients are needed.  The token hangs on the node's
        ctx: if the graph is dropped without a backward (a validation forward outside no_grad), the token dies with it and
        the network stops 'awaiting' - a penalty gradient is then never parked for a backward that will not happen."""
        ctx._nvq_token = _Token()
        self._pending = weakref.ref(ctx._nvq_token)
        self._deferred_adds = []               # leftovers could only come from a pass that was abandoned half-way

    @property
    def _awaiting_backward(self) -> bool:
        return self._pending is not None and self._pending() is not None

    # ------------------------------------------------------------------ backward epilogue
    def _finish_bucket(self, flat: torch.Tensor) -> None:
        """Called by the network's backward once every gradient is in `flat`: data-parallel all-reduce, then the
        deferred penalty gradients (identical on every rank, hence after the reduce)."""
        hook = self._grad_bucket_hook
        if hook is not None:
            hook(flat)
        if self._deferred_adds:
            theta = self.flat_theta()
            for lam, star, fisher, scale in self._deferred_adds:
                _nvq.ewc_penalty_grad(theta, star, fisher, lam, scale, flat, True)
            self._deferred_adds = []
        self._pending = None
        self._last_grad_bucket = flat
