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

from nerve_cl import _engine, _nvq, parallel


def _flat_views(flat: torch.Tensor, named: "List[tuple]") -> Dict[str, torch.Tensor]:
    out, off = {}, 0
    for name, p in named:
        out[name] = flat[off:off + p.numel()].view(p.shape)
        off += p.numel()
    return out


class _PenaltyFn(torch.autograd.Function):
    """lambda/2 * sum F (theta - theta*)^2 over a list of parameter tensors."""

    @staticmethod
    def forward(ctx, lam: float, star: torch.Tensor, fisher: torch.Tensor, *params):
        theta = torch.cat([p.detach().reshape(-1) for p in params])
        out = torch.empty(1, dtype=torch.float32, device=theta.device)
        _nvq.ewc_penalty(theta, star, fisher, float(lam), out, _engine.workspace(theta.device))
        ctx.lam, ctx.theta, ctx.star, ctx.fisher = float(lam), theta, star, fisher
        ctx.shapes = [p.shape for p in params]
        return out.reshape(())

    @staticmethod
    def backward(ctx, go):
        g = torch.empty_like(ctx.theta)
        scale = go.detach().to(torch.float32).reshape(1).contiguous()
        _nvq.ewc_penalty_grad(ctx.theta, ctx.star, ctx.fisher, ctx.lam, scale, g, False)
        grads, off = [], 0
        for shp in ctx.shapes:
            n = int(torch.Size(shp).numel())
            grads.append(g[off:off + n].view(shp))
            off += n
        return (None, None, None) + tuple(grads)


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
        # flat buckets behind the dicts above (the dict entries are views into them)
        self._flat: Dict[object, "tuple[torch.Tensor, torch.Tensor, List[str]]"] = {}

    # ------------------------------------------------------------------ helpers
    def _get_params(self) -> Iterator[tuple]:
        for name, param in self.model.named_parameters():
            if param.requires_grad:
                yield name, param

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
        divided by the number of samples seen (so it depends on the loader's batch size)."""
        dev = self._device()
        named = list(self._get_params())
        flat = torch.zeros(sum(p.numel() for _, p in named), dtype=torch.float32, device=dev)
        self.model.eval()
        seen = 0
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
            outputs = self.model(inputs)
            if empirical and targets is not None:
                loss = nn.functional.mse_loss(outputs, targets.to(dev))
            else:
                loss = -0.5 * (outputs ** 2).sum() if outputs.dim() > 1 else outputs.sum()
            loss.backward()
            g = torch.cat([(p.grad if p.grad is not None else torch.zeros_like(p)).detach().reshape(-1)
                           for _, p in named])
            _nvq.fisher_accumulate(g, flat)
            seen += inputs.size(0)
        if parallel.world_size(self.process_group) > 1:
            parallel.allreduce_sum_(flat, self.process_group)
            cnt = torch.tensor([float(seen)], device=dev)
            parallel.allreduce_sum_(cnt, self.process_group)
            seen = int(cnt.item())
        flat /= max(seen, 1)
        self._last_fisher_flat = flat
        return _flat_views(flat, named)

    def register_task(self, task_id: int, dataloader, num_samples: Optional[int] = None) -> None:
        fisher = self.compute_fisher(dataloader, num_samples)
        f_flat = self._last_fisher_flat
        named = list(self._get_params())
        names = [n for n, _ in named]
        o_flat = torch.cat([p.detach().reshape(-1) for _, p in named])
        optpar = _flat_views(o_flat, named)
        if self.mode == "separate":
            self.task_fisher[task_id] = fisher
            self.task_optpar[task_id] = optpar
            self._flat[task_id] = (f_flat, o_flat, names)
        elif self.mode == "online":
            if len(self.fisher_dict) == 0:
                self.fisher_dict = fisher
            else:
                old = self._flat["online"][0]
                merged = self.decay * old + (1 - self.decay) * f_flat
                self.fisher_dict = _flat_views(merged, named)
                f_flat = merged
            self.optpar_dict = optpar
            self._flat["online"] = (f_flat, o_flat, names)
        self.num_tasks += 1

    # ------------------------------------------------------------------ penalty
    def _penalty_one(self, model: nn.Module, key) -> torch.Tensor:
        f_flat, o_flat, names = self._flat[key]
        have = dict(model.named_parameters())
        params = [have[n] for n in names]
        return _PenaltyFn.apply(self.ewc_lambda, o_flat, f_flat, *params)

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
        names = [n for n, _ in self.model.named_parameters() if n in fd]
        f_flat = torch.cat([fd[n].to(dev, torch.float32).reshape(-1) for n in names])
        o_flat = torch.cat([od[n].to(dev, torch.float32).reshape(-1) for n in names])
        self._flat[key] = (f_flat, o_flat, names)

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
        self._flat = {}


class OnlineEWC(EWC):
    """Online EWC (single running Fisher), reference ewc.py:290-303."""

    def __init__(self, model: nn.Module, ewc_lambda: float = 5000.0, decay: float = 0.999):
        super().__init__(model, ewc_lambda, mode="online", decay=decay)



class SynapticIntelligence:
    """Synaptic Intelligence (Zenke et al. 2017) with the reference's interface (nerve_cl/continual/ewc.py:306-379):
    ``update_importance()`` after every optimizer step (W += -grad * (theta - theta_prev)), ``register_task()`` at a task
    end (omega += W / (delta^2 + damping)), ``penalty()`` = si_lambda * sum omega (theta - theta_old)^2.

    W, theta_old and omega live in one flat fp32 bucket each (``W`` / ``p_old`` / ``omega`` are dicts of views into them,
    keyed by parameter name as in the reference); the penalty and its gradient are the same two libnvq kernels EWC uses
    (``nvq_ewc_penalty`` with lambda = 2 * si_lambda, Fisher := omega, theta* := theta_old)."""

    def __init__(self, model: nn.Module, si_lambda: float = 1.0, damping: float = 0.1):
        self.model = model
        self.si_lambda = si_lambda
        self.damping = damping
        self._named = [(n, p) for n, p in model.named_parameters() if p.requires_grad]
        dev = next(model.parameters()).device
        if dev.type != "cuda":
            raise RuntimeError(f"SynapticIntelligence needs the model on the GPU (found {dev}); there is no CPU fallback")
        n = sum(p.numel() for _, p in self._named)
        self._W = torch.zeros(n, dtype=torch.float32, device=dev)
        self._omega = torch.zeros(n, dtype=torch.float32, device=dev)
        self._p_old = self._theta()
        self.W = _flat_views(self._W, self._named)
        self.omega = _flat_views(self._omega, self._named)
        self.p_old = _flat_views(self._p_old, self._named)

    def _theta(self) -> torch.Tensor:
        return torch.cat([p.detach().reshape(-1).float() for _, p in self._named])

    def _set_p_old(self, theta: torch.Tensor) -> None:
        self._p_old.copy_(theta)                              # the views in self.p_old stay valid

    def update_importance(self) -> None:
        """Call after each optimizer step (parameters without a gradient contribute nothing, as in the reference)."""
        theta = self._theta()
        g = torch.cat([(p.grad.detach().reshape(-1).float() if p.grad is not None else torch.zeros(p.numel(), device=theta.device))
                       for _, p in self._named])
        has = torch.cat([torch.full((p.numel(),), p.grad is not None, dtype=torch.bool, device=theta.device)
                         for _, p in self._named])
        self._W.add_(torch.where(has, -g * (theta - self._p_old), torch.zeros_like(g)))
        self._set_p_old(torch.where(has, theta, self._p_old))

    def register_task(self) -> None:
        theta = self._theta()
        delta = theta - self._p_old
        self._omega.add_(self._W / (delta * delta + self.damping))
        self._W.zero_()
        self._set_p_old(theta)

    def penalty(self) -> torch.Tensor:
        return _PenaltyFn.apply(2.0 * self.si_lambda, self._p_old, self._omega, *[p for _, p in self._named])
