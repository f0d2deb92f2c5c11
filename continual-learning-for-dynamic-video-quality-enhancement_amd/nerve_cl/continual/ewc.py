"""Elastic Weight Consolidation on flat parameter buckets (MI355X).

Same interface and semantics as the reference's ``EWC`` (nerve_cl/continual/ewc.py:19-287):
``EWC(model, ewc_lambda, mode, decay)``, ``compute_fisher``, ``register_task``, ``penalty``,
``state_dict``/``load_state_dict``, ``fisher_dict``/``optpar_dict`` keyed by parameter name.

What differs is the execution: the reference evaluates the penalty with ~5 tiny torch ops per
parameter tensor (131 tensors -> ~1300 kernel launches forward+backward on a GPU).  Here Fisher,
theta* and the gradient live in ONE flat fp32 bucket each and the work is three libnvq kernels:
``nvq_ewc_penalty`` (forward, one reduction), ``nvq_ewc_penalty_grad`` (backward: lambda*F*(theta-theta*)
times the upstream scalar, written for all tensors at once) and ``nvq_fisher_accumulate``
(``fisher += grad**2`` per batch).  Data parallel: every rank accumulates over its own batches
and the Fisher bucket is all-reduced (sum) once per task.

The model's parameters must live on the GPU; there is no CPU fallback.
"""
from __future__ import annotations

from typing import Dict, Iterator, List, Optional

import torch
import torch.nn as nn

from nerve_cl import _engine, _nvq, ops, parallel
from nerve_cl._bucket import BucketedNet


def _flat_views(flat: torch.Tensor, named: "List[tuple]") -> Dict[str, torch.Tensor]:
    out, off = {}, 0
    for name, p in named:
        out[name] = flat[off:off + p.numel()].view(p.shape)
        off += p.numel()
    return out


class _Segment:
    """One flat bucket of the penalty: either all parameters of one bucketed network (``net`` set: Fisher / theta* are
    stored in that network's gradient-bucket layout, so the penalty gradient is ONE accumulate kernel into the bucket) or
    a plain concatenation of loose parameters (``net`` None: e.g. the engine's ``enhancement_strength``)."""

    def __init__(self, net: Optional[BucketedNet], named: "List[tuple]"):
        self.net, self.named = net, named          # named: [(full name, parameter)]
        self.names = [n for n, _ in named]

    def numel(self) -> int:
        if self.net is not None:
            return self.net._bucket_layout()[1]
        return sum(p.numel() for _, p in self.named)

    def views(self, flat: torch.Tensor) -> Dict[str, torch.Tensor]:
        if self.net is None:
            return _flat_views(flat, self.named)
        local = self.net._bucket_views(flat)
        return {full: local[loc] for (full, _), loc in zip(self.named, self.net._param_names)}

    def theta(self) -> torch.Tensor:
        if self.net is not None:
            return self.net.flat_theta()           # persistent flat view of the parameters: no per-step cat
        return torch.cat([p.detach().reshape(-1).float() for _, p in self.named])

    def grads(self) -> Optional[torch.Tensor]:
        """Flat gradient of the last backward in this segment's layout (None when nothing arrived)."""
        if self.net is not None:
            return self.net._last_grad_bucket
        if all(p.grad is None for _, p in self.named):
            return None
        return torch.cat([(p.grad if p.grad is not None else torch.zeros_like(p)).detach().reshape(-1).float()
                          for _, p in self.named])


def segments_of(model: nn.Module) -> "List[_Segment]":
    """Split model.named_parameters() (requires_grad only, reference ewc.py:67-71) into bucketed networks + loose rest."""
    owner: Dict[int, "tuple[BucketedNet, str]"] = {}
    for prefix, m in model.named_modules():
        if isinstance(m, BucketedNet) and all(p.requires_grad for p in m.parameters()):
            for p in m.parameters():
                owner.setdefault(id(p), (m, prefix))
    segs: "Dict[int, _Segment]" = {}
    order: "List[_Segment]" = []
    loose: "List[tuple]" = []
    for name, p in model.named_parameters():
        if not p.requires_grad:
            continue
        own = owner.get(id(p))
        if own is None:
            loose.append((name, p))
            continue
        net = own[0]
        if id(net) not in segs:
            segs[id(net)] = _Segment(net, [])
            order.append(segs[id(net)])
        segs[id(net)].named.append((name, p))
    for sg in order:
        sg.names = [n for n, _ in sg.named]
        assert len(sg.named) == len(sg.net._param_names)
    if loose:
        order.append(_Segment(None, loose))
    return order


class _PenaltyFn(torch.autograd.Function):
    """lambda/2 * sum F (theta - theta*)^2 over one segment (reference ewc.py:225-232).

    Backward, bucketed network whose own backward is still to come in this pass (the usual ``loss = mse + penalty``
    order: autograd runs the later-created node first): the term ``go * lambda * F * (theta - theta*)`` is queued on the
    network and added to its flat gradient bucket by one kernel after the data-parallel all-reduce; no per-tensor
    gradients are returned, so autograd performs no AccumulateGrad additions for it.  Otherwise (loose parameters, or
    the network's backward already ran): the gradient is written by the same kernel into a fresh flat tensor and its
    views are returned."""

    @staticmethod
    def forward(ctx, seg: _Segment, lam: float, star: torch.Tensor, fisher: torch.Tensor, *params):
        theta = seg.theta() if seg is not None else torch.cat([p.detach().reshape(-1).float() for p in params])
        out = torch.empty(1, dtype=torch.float32, device=theta.device)
        with _nvq.device_guard(theta.device):
            _nvq.ewc_penalty(theta, star, fisher, float(lam), out, _engine.workspace(theta.device))
        ctx.seg, ctx.lam, ctx.star, ctx.fisher = seg, float(lam), star, fisher
        # a bucketed segment reads theta again at backward time from the persistent flat view (same values)
        ctx.theta = None if (seg is not None and seg.net is not None) else theta
        ctx.shapes = [p.shape for p in params]
        return out.reshape(())

    @staticmethod
    def backward(ctx, go):
        seg = ctx.seg
        scale = go.detach().to(torch.float32).reshape(1).contiguous()
        net = seg.net if seg is not None else None
        if net is not None and net._awaiting_backward and getattr(net, "fuse_penalty_gradient", True):
            net._deferred_adds.append((ctx.lam, ctx.star, ctx.fisher, scale))
            return (None, None, None, None) + (None,) * len(ctx.shapes)
        theta = net.flat_theta() if net is not None else ctx.theta
        g = torch.empty_like(theta)
        with _nvq.device_guard(theta.device):
            _nvq.ewc_penalty_grad(theta, ctx.star, ctx.fisher, ctx.lam, scale, g, False)
        if net is not None:
            views = net._bucket_views(g)
            grads = [views[n] for n in net._param_names]
        else:
            grads, off = [], 0
            for shp in ctx.shapes:
                n = int(torch.Size(shp).numel())
                grads.append(g[off:off + n].view(shp))
                off += n
        return (None, None, None, None) + tuple(grads)


class EWC:
    """Elastic Weight Consolidation (Kirkpatrick et al. 2017), 'online' or 'separate' mode."""

    def __init__(self, model: nn.Module, ewc_lambda: float = 5000.0, mode: str = "online",
                 decay: float = 0.999, process_group=None):
        self.model = model
        self.ewc_lambda = ewc_lambda
        self.mode = mode
        self.decay = decay
        self.process_group = process_group
        self.fisher_dict: Dict[str, torch.Tensor] = {}
        self.optpar_dict: Dict[str, torch.Tensor] = {}
        self.task_fisher: Dict[int, Dict[str, torch.Tensor]] = {}
        self.task_optpar: Dict[int, Dict[str, torch.Tensor]] = {}
        self.num_tasks = 0
        # flat buckets behind the dicts above (the dict entries are views into them):
        # key ('online' | task id) -> [(segment, fisher_flat, optpar_flat)]
        self._flat: Dict[object, "List[tuple]"] = {}
        self._live: Dict[object, "List[bool]"] = {}      # per key: which segments ever received a gradient
        self._segs: "Optional[List[_Segment]]" = None

    # ------------------------------------------------------------------ helpers
    def _get_params(self) -> Iterator[tuple]:
        for name, param in self.model.named_parameters():
            if param.requires_grad:
                yield name, param

    def _segments(self) -> "List[_Segment]":
        if self._segs is None:
            self._segs = segments_of(self.model)
        return self._segs

    def _device(self) -> torch.device:
        dev = next(self.model.parameters()).device
        if dev.type != "cuda":
            raise RuntimeError(
                f"EWC needs the model on the GPU (found {dev}): the penalty, its gradient and the Fisher "
                "accumulation run as libnvq HIP kernels; there is no CPU fallback.")
        return dev

    # ------------------------------------------------------------------ Fisher
    def compute_fisher(self, dataloader, num_samples: Optional[int] = None,
                       empirical: bool = True) -> Dict[str, torch.Tensor]:
        """Diagonal empirical Fisher exactly as the reference defines it (ewc.py:73-149): eval mode,
        per batch zero_grad -> forward -> batch-mean MSE -> backward -> fisher += grad**2, finally
        divided by the number of samples seen (so it depends on the loader's batch size).

        Data parallel: every rank squares ITS OWN batch gradients - the gradient all-reduce hook of
        ``nerve_cl.parallel`` is switched off for the duration - and the Fisher buckets and the sample count are summed
        over the ranks in one all-reduce at the end, so the result equals the single-process Fisher over the union of
        the ranks' batches.  No collective runs inside the loop, so ranks may see different batch counts."""
        dev = self._device()
        segs = self._segments()
        flats = [torch.zeros(sg.numel(), dtype=torch.float32, device=dev) for sg in segs]
        nets = [m for m in self.model.modules() if isinstance(m, BucketedNet)]
        hooks = [m._grad_bucket_hook for m in nets]
        for m in nets:
            m._grad_bucket_hook = None
        self.model.eval()
        seen = 0
        touched = [False] * len(segs)       # did any batch deliver a gradient to this segment (else its Fisher is exactly 0)
        try:
            for batch in dataloader:
                if num_samples is not None and seen >= num_samples:
                    break
                if isinstance(batch, (tuple, list)):
                    inputs = batch[0]
                    targets = batch[1] if len(batch) > 1 else None
                else:
                    inputs, targets = batch, None
                inputs = inputs.to(dev)
                self.model.zero_grad()
                for m in nets:
                    m._last_grad_bucket = None
                outputs = self.model(inputs)
                if empirical and targets is not None:
                    loss = ops.mse_loss(outputs, targets.to(dev))      # F.mse_loss of ewc.py:125, as a HIP kernel
                else:
                    loss = -0.5 * (outputs ** 2).sum() if outputs.dim() > 1 else outputs.sum()
                loss.backward()
                with _nvq.device_guard(dev):
                    for i, (sg, flat) in enumerate(zip(segs, flats)):
                        g = sg.grads()
                        if g is not None:
                            _nvq.fisher_accumulate(g, flat)
                            touched[i] = True
                seen += inputs.size(0)
        finally:
            for m, h in zip(nets, hooks):
                m._grad_bucket_hook = h
        if parallel.world_size(self.process_group) > 1:
            packed = torch.cat(flats + [torch.tensor([float(seen)], device=dev)])
            parallel.allreduce_sum_(packed, self.process_group)
            off = 0
            for flat in flats:
                flat.copy_(packed[off:off + flat.numel()])
                off += flat.numel()
            seen = int(round(packed[off].item()))
        for flat in flats:
            flat /= max(seen, 1)
        self._last_fisher_flats = flats
        self._last_fisher_touched = touched
        out: Dict[str, torch.Tensor] = {}
        for sg, flat in zip(segs, flats):
            out.update(sg.views(flat))
        return {n: out[n] for n, _ in self._get_params()}

    def register_task(self, task_id: int, dataloader, num_samples: Optional[int] = None) -> None:
        fisher = self.compute_fisher(dataloader, num_samples)
        segs, f_flats, touched = self._segments(), self._last_fisher_flats, self._last_fisher_touched
        o_flats = [sg.theta().clone() for sg in segs]
        optpar: Dict[str, torch.Tensor] = {}
        for sg, o in zip(segs, o_flats):
            optpar.update(sg.views(o))
        optpar = {n: optpar[n] for n, _ in self._get_params()}
        if self.mode == "separate":
            self.task_fisher[task_id] = fisher
            self.task_optpar[task_id] = optpar
            self._flat[task_id] = list(zip(segs, f_flats, o_flats))
            self._live[task_id] = touched
        elif self.mode == "online":
            if len(self.fisher_dict) == 0:
                self.fisher_dict = fisher
            else:
                self._ensure_flat("online")                       # e.g. right after load_state_dict
                merged = [self.decay * old + (1 - self.decay) * new
                          for (_, old, _), new in zip(self._flat["online"], f_flats)]
                fd: Dict[str, torch.Tensor] = {}
                for sg, m in zip(segs, merged):
                    fd.update(sg.views(m))
                self.fisher_dict = {n: fd[n] for n, _ in self._get_params()}
                f_flats = merged
            self.optpar_dict = optpar
            self._flat["online"] = list(zip(segs, f_flats, o_flats))
            old_live = self._live.get("online", [False] * len(segs))
            self._live["online"] = [a or b for a, b in zip(old_live, touched)]
        self.num_tasks += 1

    # ------------------------------------------------------------------ penalty
    def _penalty_one(self, model: nn.Module, key) -> torch.Tensor:
        total = None
        if model is self.model:
            live = self._live.get(key)
            for i, (sg, f_flat, o_flat) in enumerate(self._flat[key]):
                if live is not None and not live[i]:
                    continue                 # a segment no batch ever reached (e.g. an unused head): Fisher 0, term 0
                term = _PenaltyFn.apply(sg, self.ewc_lambda, o_flat, f_flat, *[p for _, p in sg.named])
                total = term if total is None else total + term
            if total is None:
                total = torch.zeros((), dtype=torch.float32, device=self._device())
            return total
        # another module with the same parameter names (reference signature penalty(model)): plain concatenation
        have = dict(model.named_parameters())
        for sg, f_flat, o_flat in self._flat[key]:
            fv, ov = sg.views(f_flat), sg.views(o_flat)
            params = [have[n] for n in sg.names]
            f = torch.cat([fv[n].reshape(-1) for n in sg.names])
            o = torch.cat([ov[n].reshape(-1) for n in sg.names])
            term = _PenaltyFn.apply(None, self.ewc_lambda, o, f, *params)
            total = term if total is None else total + term
        if total is None:                                      # no trainable parameter: si_lambda * 0.0 (reference ewc.py:370-379)
            return torch.zeros((), device=next(self.model.parameters()).device)
        return total

    def penalty(self, model: Optional[nn.Module] = None):
        """lambda/2 * sum_i F_i (theta_i - theta*_i)^2 ; python 0.0 before any task is registered
        (as in the reference, ewc.py:210,232)."""
        if model is None:
            model = self.model
        if self.mode == "separate":
            total = 0.0
            for task_id in self.task_fisher:
                self._ensure_flat(task_id)
                total = total + self._penalty_one(model, task_id)
            return total
        if len(self.fisher_dict) == 0:
            return self.ewc_lambda / 2 * 0.0
        self._ensure_flat("online")
        return self._penalty_one(model, "online")

    def _ensure_flat(self, key) -> None:
        """Rebuild the flat buckets after load_state_dict (dicts of CPU tensors)."""
        if key in self._flat:
            return
        dev = self._device()
        fd, od = (self.fisher_dict, self.optpar_dict) if key == "online" else (self.task_fisher[key], self.task_optpar[key])
        entry = []
        for sg in self._segments():
            f_flat = torch.zeros(sg.numel(), dtype=torch.float32, device=dev)
            o_flat = torch.zeros(sg.numel(), dtype=torch.float32, device=dev)
            fv, ov = sg.views(f_flat), sg.views(o_flat)
            for n in sg.names:
                if n in fd:
                    fv[n].copy_(fd[n].to(dev, torch.float32))
                    ov[n].copy_(od[n].to(dev, torch.float32))
            entry.append((sg, f_flat, o_flat))
        self._flat[key] = entry

    # ------------------------------------------------------------------ bookkeeping
    def get_importance_stats(self) -> Dict[str, dict]:
        if self.mode == "online":
            fisher = self.fisher_dict
        else:
            fisher: Dict[str, torch.Tensor] = {}
            for tf in self.task_fisher.values():
                for n, f in tf.items():
                    fisher[n] = f.clone() if n not in fisher else fisher[n] + f
        return {n: {"mean": f.mean().item(), "max": f.max().item(), "std": f.std().item(),
                    "nonzero": (f > 0).float().mean().item()} for n, f in fisher.items()}

    def state_dict(self) -> Dict:
        cpu = lambda d: {k: v.detach().cpu().clone() for k, v in d.items()}  # noqa: E731
        return {
            "ewc_lambda": self.ewc_lambda, "mode": self.mode, "decay": self.decay, "num_tasks": self.num_tasks,
            "fisher_dict": cpu(self.fisher_dict), "optpar_dict": cpu(self.optpar_dict),
            "task_fisher": {t: cpu(f) for t, f in self.task_fisher.items()},
            "task_optpar": {t: cpu(o) for t, o in self.task_optpar.items()},
        }

    def load_state_dict(self, state: Dict) -> None:
        self.ewc_lambda, self.mode, self.decay = state["ewc_lambda"], state["mode"], state["decay"]
        self.num_tasks = state["num_tasks"]
        self.fisher_dict, self.optpar_dict = state["fisher_dict"], state["optpar_dict"]
        self.task_fisher, self.task_optpar = state["task_fisher"], state["task_optpar"]
        self._flat, self._live = {}, {}


class OnlineEWC(EWC):
    """Online EWC (single running Fisher), reference ewc.py:290-303."""

    def __init__(self, model: nn.Module, ewc_lambda: float = 5000.0, decay: float = 0.999):
        super().__init__(model, ewc_lambda, mode="online", decay=decay)



class SynapticIntelligence:
    """Synaptic Intelligence (Zenke et al. 2017) with the reference's interface (nerve_cl/continual/ewc.py:306-379):
    ``update_importance()`` after every optimizer step (W += -grad * (theta - theta_prev); theta_prev = theta),
    ``register_task()`` at a task end (omega += W / (delta^2 + damping)), ``penalty()`` = si_lambda * sum omega (theta - p_old)^2.

    Like EWC, the state lives on flat buckets, one per segment (``segments_of``): for a bucketed network W / p_old / omega are
    flat fp32 tensors in the network's gradient-bucket layout, so a step's update is ONE ``nvq_si_update`` launch reading the
    persistent flat parameter view and the flat gradient bucket the backward just wrote - no per-step ``cat`` over 131
    tensors; ``register_task`` is one ``nvq_si_consolidate``; the penalty and its gradient are the two EWC kernels
    (``nvq_ewc_penalty`` with lambda = 2 * si_lambda, Fisher := omega, theta* := p_old), fused into the network's bucket when
    its backward is still to come.  ``W`` / ``p_old`` / ``omega`` are dicts of views keyed by parameter name, as in the
    reference.  (Reference quirk kept: ``p_old`` is refreshed by every ``update_importance``, so a loop that calls it after
    each step sees a zero penalty; oracle/cl_cases.si_drive.)"""

    def __init__(self, model: nn.Module, si_lambda: float = 1.0, damping: float = 0.1):
        self.model = model
        self.si_lambda = si_lambda
        self.damping = damping
        dev = next(model.parameters()).device
        if dev.type != "cuda":
            raise RuntimeError(f"SynapticIntelligence needs the model on the GPU (found {dev}); there is no CPU fallback")
        self._segs = segments_of(model)
        self._W = [torch.zeros(sg.numel(), dtype=torch.float32, device=dev) for sg in self._segs]
        self._omega = [torch.zeros(sg.numel(), dtype=torch.float32, device=dev) for sg in self._segs]
        self._p_old = [sg.theta().clone() for sg in self._segs]
        self.W: Dict[str, torch.Tensor] = {}
        self.omega: Dict[str, torch.Tensor] = {}
        self.p_old: Dict[str, torch.Tensor] = {}
        for sg, w, o, p in zip(self._segs, self._W, self._omega, self._p_old):
            self.W.update(sg.views(w))
            self.omega.update(sg.views(o))
            self.p_old.update(sg.views(p))

    @staticmethod
    def _bucket_is_grad(sg: _Segment) -> bool:
        """Are the parameters' .grad tensors the views of the network's last gradient bucket (the plain zero_grad ->
        backward -> step loop)?  Otherwise (gradient accumulation over several backwards) .grad is gathered instead."""
        flat = sg.net._last_grad_bucket
        if flat is None:
            return False
        lay, _ = sg.net._bucket_layout()
        base = flat.data_ptr()
        for (_, p), loc in zip(sg.named, sg.net._param_names):
            if p.grad is None or p.grad.data_ptr() != base + 4 * lay[loc][0]:
                return False
        return True

    def update_importance(self) -> None:
        """Call after each optimizer step (parameters without a gradient contribute nothing, as in the reference)."""
        for sg, W, p_old in zip(self._segs, self._W, self._p_old):
            theta = sg.theta()
            with _nvq.device_guard(theta.device):
                if sg.net is not None and self._bucket_is_grad(sg):
                    _nvq.si_update(theta, sg.net._last_grad_bucket, p_old, W)
                    continue
                views_w, views_p = sg.views(W), sg.views(p_old)
                for n, p in sg.named:                         # loose parameters / accumulated gradients: per tensor
                    if p.grad is not None and p.numel() > 0:
                        g = p.grad.detach().float().contiguous()
                        _nvq.si_update(p.detach().float().contiguous(), g, views_p[n].view(-1), views_w[n].view(-1))

    def register_task(self) -> None:
        for sg, W, p_old, omega in zip(self._segs, self._W, self._p_old, self._omega):
            theta = sg.theta()
            with _nvq.device_guard(theta.device):
                _nvq.si_consolidate(theta, float(self.damping), p_old, W, omega)

    def penalty(self) -> torch.Tensor:
        total = None
        for sg, p_old, omega in zip(self._segs, self._p_old, self._omega):
            term = _PenaltyFn.apply(sg, 2.0 * self.si_lambda, p_old, omega, *[p for _, p in sg.named])
            total = term if total is None else total + term
        return total
