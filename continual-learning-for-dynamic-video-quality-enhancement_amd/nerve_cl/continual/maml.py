"""Meta-learning helpers with the reference's interface (nerve_cl/continual/maml.py): ``MAML`` / ``FOMAML`` (:23-244),
``Reptile`` (:247-372), ``ContentAdaptiveMAML`` (:375-431).  They are host-side loops around plain forward / backward /
deepcopy of the model, which is all they need from the MI355X path.

Deviations, both where the reference cannot run: its first-order ``meta_step`` adapts under ``torch.no_grad()`` (the inner
``loss.backward()`` raises) and evaluates the query loss on a deep copy (no gradient could reach the meta-parameters); here
the query-loss gradient at the adapted weights is applied to the meta-parameters, i.e. first-order MAML as published.
Second-order MAML needs the ``higher`` package (absent) and raises NotImplementedError."""
from __future__ import annotations

from copy import deepcopy
from typing import Callable, Dict, List, Optional, Tuple

import torch
import torch.nn as nn


def _sgd_steps(model: nn.Module, data: Tuple[torch.Tensor, torch.Tensor], loss_fn: Callable, lr: float, steps: int) -> float:
    device = next(model.parameters()).device
    inputs, targets = data[0].to(device), data[1].to(device)
    opt = torch.optim.SGD(model.parameters(), lr=lr)
    loss = torch.zeros((), device=device)
    for _ in range(steps):
        opt.zero_grad()
        loss = loss_fn(model(inputs), targets)
        loss.backward()
        opt.step()
    return float(loss.item())


class _MetaLoss:
    """What first-order ``meta_step`` returns: ``backward()`` hands the averaged query gradients to the meta-parameters,
    ``item()`` / ``float()`` give the averaged query loss (the two things the reference's callers do with the tensor)."""

    def __init__(self, value: float, params: List[nn.Parameter], grads: List[Optional[torch.Tensor]]):
        self._value, self._params, self._grads = value, params, grads

    def backward(self) -> None:
        for p, g in zip(self._params, self._grads):
            if g is not None:
                p.grad = g.clone() if p.grad is None else p.grad + g

    def item(self) -> float:
        return self._value

    def __float__(self) -> float:
        return self._value


class MAML:
    def __init__(self, model: nn.Module, inner_lr: float = 0.01, outer_lr: float = 0.001, inner_steps: int = 5,
                 first_order: bool = True):
        self.model = model
        self.inner_lr, self.outer_lr, self.inner_steps, self.first_order = inner_lr, outer_lr, inner_steps, first_order
        self.meta_optimizer = torch.optim.Adam(self.model.parameters(), lr=outer_lr)

    def _inner_loop(self, model: nn.Module, support_data, loss_fn: Callable, steps: Optional[int] = None) -> nn.Module:
        adapted = deepcopy(model)
        _sgd_steps(adapted, support_data, loss_fn, self.inner_lr, steps or self.inner_steps)
        return adapted

    def meta_step(self, task_batch: List[Dict], loss_fn: Callable) -> _MetaLoss:
        if not self.first_order:
            raise NotImplementedError("second-order MAML needs the `higher` package, which is not available here")
        params = list(self.model.parameters())
        grads: List[Optional[torch.Tensor]] = [None] * len(params)
        total = 0.0
        for task in task_batch:
            adapted = self._inner_loop(self.model, task["support"], loss_fn)
            adapted.zero_grad()
            device = next(adapted.parameters()).device
            q_in, q_tgt = task["query"]
            loss = loss_fn(adapted(q_in.to(device)), q_tgt.to(device))
            loss.backward()
            total += float(loss.item())
            for i, ap in enumerate(adapted.parameters()):
                if ap.grad is not None:
                    g = ap.grad.detach() / len(task_batch)
                    grads[i] = g if grads[i] is None else grads[i] + g
        return _MetaLoss(total / max(len(task_batch), 1), params, grads)

    def adapt(self, data, loss_fn: Callable, steps: Optional[int] = None) -> nn.Module:
        return self._inner_loop(self.model, data, loss_fn, steps)

    def train_step(self, task_batch: List[Dict], loss_fn: Callable) -> float:
        self.meta_optimizer.zero_grad()
        meta_loss = self.meta_step(task_batch, loss_fn)
        meta_loss.backward()
        self.meta_optimizer.step()
        return meta_loss.item()

    def state_dict(self) -> Dict:
        return {"model": self.model.state_dict(), "meta_optimizer": self.meta_optimizer.state_dict(),
                "inner_lr": self.inner_lr, "outer_lr": self.outer_lr, "inner_steps": self.inner_steps,
                "first_order": self.first_order}

    def load_state_dict(self, state: Dict) -> None:
        self.model.load_state_dict(state["model"])
        self.meta_optimizer.load_state_dict(state["meta_optimizer"])
        self.inner_lr, self.outer_lr = state["inner_lr"], state["outer_lr"]
        self.inner_steps, self.first_order = state["inner_steps"], state["first_order"]


class FOMAML(MAML):
    def __init__(self, model: nn.Module, inner_lr: float = 0.01, outer_lr: float = 0.001, inner_steps: int = 5):
        super().__init__(model, inner_lr, outer_lr, inner_steps, first_order=True)


class Reptile:
    """Move the initialisation toward the average of the per-task adapted parameters (Nichol et al. 2018)."""

    def __init__(self, model: nn.Module, inner_lr: float = 0.01, outer_lr: float = 0.1, inner_steps: int = 10):
        self.model, self.inner_lr, self.outer_lr, self.inner_steps = model, inner_lr, outer_lr, inner_steps

    def train_step(self, task_batch: List[Dict], loss_fn: Callable) -> float:
        init = {n: p.data.clone() for n, p in self.model.named_parameters()}
        mean = {n: torch.zeros_like(v) for n, v in init.items()}
        total = 0.0
        for task in task_batch:
            for n, p in self.model.named_parameters():
                p.data.copy_(init[n])
            total += _sgd_steps(self.model, task["support"], loss_fn, self.inner_lr, self.inner_steps)
            for n, p in self.model.named_parameters():
                mean[n] += p.data / len(task_batch)
        with torch.no_grad():
            for n, p in self.model.named_parameters():
                p.data.copy_(init[n] + self.outer_lr * (mean[n] - init[n]))
        return total / len(task_batch)

    def adapt(self, data, loss_fn: Callable, steps: Optional[int] = None) -> nn.Module:
        adapted = deepcopy(self.model)
        _sgd_steps(adapted, data, loss_fn, self.inner_lr, steps or self.inner_steps)
        return adapted


class ContentAdaptiveMAML(MAML):
    """First-order MAML with one adaptation learning rate per content type."""

    def __init__(self, model: nn.Module, content_types: List[str], inner_lr: float = 0.01, outer_lr: float = 0.001,
                 inner_steps: int = 5):
        super().__init__(model, inner_lr, outer_lr, inner_steps, first_order=True)
        self.content_types = content_types
        self.content_lr = nn.ParameterDict({ct: nn.Parameter(torch.tensor(inner_lr)) for ct in content_types})

    def adapt_to_content(self, data, content_type: str, loss_fn: Callable, steps: Optional[int] = None) -> nn.Module:
        lr = self.content_lr[content_type].item() if content_type in self.content_lr else self.inner_lr
        adapted = deepcopy(self.model)
        _sgd_steps(adapted, data, loss_fn, lr, steps or self.inner_steps)
        return adapted
