"""First-order MAML helper kept importable for train_continual.py:12 (no script branch uses it,
SURVEY.md appendix item 1).  Only needs a model that survives copy.deepcopy and plain forward/backward."""
from __future__ import annotations

import copy
from typing import Callable, Sequence, Tuple

import torch
import torch.nn as nn


class FOMAML:
    def __init__(self, model: nn.Module, inner_lr: float = 0.01, outer_lr: float = 1e-3, inner_steps: int = 5):
        self.model, self.inner_lr, self.inner_steps = model, inner_lr, inner_steps
        self.meta_opt = torch.optim.Adam(model.parameters(), lr=outer_lr)

    def adapt(self, data: Tuple[torch.Tensor, torch.Tensor], loss_fn: Callable, steps: int = None) -> nn.Module:
        """Return a fine-tuned copy after `steps` SGD steps on `data`."""
        learner = copy.deepcopy(self.model)
        opt = torch.optim.SGD(learner.parameters(), lr=self.inner_lr)
        x, y = data
        for _ in range(self.inner_steps if steps is None else steps):
            opt.zero_grad()
            loss_fn(learner(x), y).backward()
            opt.step()
        return learner

    def meta_step(self, tasks: Sequence[Tuple[Tuple[torch.Tensor, torch.Tensor], Tuple[torch.Tensor, torch.Tensor]]],
                  loss_fn: Callable) -> float:
        """First-order update: gradients of the query loss at the adapted weights are applied to the
        meta-parameters."""
        self.meta_opt.zero_grad()
        total = 0.0
        for support, query in tasks:
            learner = self.adapt(support, loss_fn)
            learner.zero_grad()
            q = loss_fn(learner(query[0]), query[1])
            q.backward()
            total += q.item()
            for p, lp in zip(self.model.parameters(), learner.parameters()):
                if lp.grad is not None:
                    p.grad = lp.grad.clone() / len(tasks) if p.grad is None else p.grad + lp.grad / len(tasks)
        self.meta_opt.step()
        return total / max(len(tasks), 1)
