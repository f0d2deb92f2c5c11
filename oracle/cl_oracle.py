"""TEST INFRASTRUCTURE ONLY (see oracle/sr_oracle.py): CPU restatement of the reference's continual-learning loops that
consume the SR hot path, as plain functions over any ``nn.Module`` (the tests drive them with ``sr_oracle.OracleSR``).

Pinned by ``tests/golden/{meta_f16,si_f16,cfg5_loop}.npz``, which ``oracle/make_goldens.py`` writes from the imported
reference (``FOMAML.adapt``, ``Reptile.train_step``, ``SynapticIntelligence``, ``EWC`` inside the ``train_with_ewc`` loop)
after asserting that these restatements agree with it; ``tests/test_oracle_golden.py`` re-checks them on every CPU run.
"""
from __future__ import annotations

from copy import deepcopy
from typing import Callable, Dict, List, Sequence, Tuple

import torch
import torch.nn.functional as F

from oracle import sr_oracle


def sgd_inner_loop(model: torch.nn.Module, data, loss_fn: Callable, lr: float, steps: int) -> float:
    """The inner loop shared by MAML._inner_loop (maml.py:74-110), Reptile.train_step (:305-318) and Reptile.adapt
    (:347-372): plain SGD, ``steps`` times zero_grad -> forward -> loss -> backward -> step.  Returns the last loss."""
    inputs, targets = data
    opt = torch.optim.SGD(model.parameters(), lr=lr)
    loss = None
    for _ in range(steps):
        opt.zero_grad()
        loss = loss_fn(model(inputs), targets)
        loss.backward()
        opt.step()
    return float(loss.item())


def fomaml_adapt(model: torch.nn.Module, data, loss_fn: Callable, inner_lr: float, steps: int) -> torch.nn.Module:
    """MAML.adapt -> _inner_loop, maml.py:168-186,74-110: a deep copy of the model after ``steps`` SGD steps."""
    adapted = deepcopy(model)
    sgd_inner_loop(adapted, data, loss_fn, inner_lr, steps)
    return adapted


def reptile_train_step(model: torch.nn.Module, task_batch: Sequence[Dict], loss_fn: Callable, inner_lr: float,
                       outer_lr: float, inner_steps: int) -> float:
    """Reptile.train_step, maml.py:276-345.  Per task the PARAMETERS are reset to their initial values (buffers - the
    BatchNorm running statistics - are not: they carry over from task to task), the model takes ``inner_steps`` SGD steps
    on the task's support set; then theta <- theta0 + outer_lr * (mean of the adapted thetas - theta0).  Returns the mean
    of the tasks' last inner losses."""
    init = {n: p.data.clone() for n, p in model.named_parameters()}
    adapted: List[Dict[str, torch.Tensor]] = []
    total = 0.0
    for task in task_batch:
        for n, p in model.named_parameters():
            p.data.copy_(init[n])
        total += sgd_inner_loop(model, task["support"], loss_fn, inner_lr, inner_steps)
        adapted.append({n: p.data.clone() for n, p in model.named_parameters()})
    with torch.no_grad():
        for n, p in model.named_parameters():
            avg = torch.stack([a[n] for a in adapted]).mean(dim=0)
            p.data.copy_(init[n] + outer_lr * (avg - init[n]))
    return total / len(task_batch)


class SI:
    """SynapticIntelligence, ewc.py:306-379: W += -grad * (theta - theta_prev) after every optimizer step,
    omega += W / ((theta - theta_prev)^2 + damping) at a task end, penalty = si_lambda * sum omega (theta - theta_old)^2.
    (``p_old`` is refreshed by EVERY update_importance call, so at register_task time delta is the last step only.)"""

    def __init__(self, model: torch.nn.Module, si_lambda: float = 1.0, damping: float = 0.1):
        self.model, self.si_lambda, self.damping = model, si_lambda, damping
        self.named = [(n, p) for n, p in model.named_parameters() if p.requires_grad]
        self.W = {n: torch.zeros_like(p) for n, p in self.named}
        self.omega = {n: torch.zeros_like(p) for n, p in self.named}
        self.p_old = {n: p.data.clone() for n, p in self.named}

    def update_importance(self) -> None:
        for n, p in self.named:
            if p.grad is not None:
                self.W[n] += -p.grad.data * (p.data - self.p_old[n])
                self.p_old[n] = p.data.clone()

    def register_task(self) -> None:
        for n, p in self.named:
            delta = p.data - self.p_old[n]
            self.omega[n] += self.W[n] / (delta ** 2 + self.damping)
            self.W[n] = torch.zeros_like(p)
            self.p_old[n] = p.data.clone()

    def penalty(self) -> torch.Tensor:
        total = 0.0
        for n, p in self.named:
            total = total + (self.omega[n] * (p - self.p_old[n]) ** 2).sum()
        return self.si_lambda * total


def train_with_ewc(model: torch.nn.Module, tasks: Sequence[Tuple[str, List]], lam: float, lr: float, epochs: int,
                   decay: float = 0.999) -> Dict[str, List[float]]:
    """The loop of experiments/train_continual.py:26-69 (``train_with_ewc``) with the reference's EWC semantics restated by
    sr_oracle.ewc_*: Adam(lr), per batch ``loss = mse + penalty`` (penalty is the python float 0.0 before the first
    register_task), ``register_task`` (online Fisher merge, theta* = theta) after each task's epochs.  ``tasks`` =
    [(name, [(inputs, targets), ...])]; the model maps the inputs to the output tensor itself (the 4-D -> 5-D adapter of
    SURVEY.md 3.4 is the caller's).  Returns the per-step 'task_loss' / 'ewc_loss' series the script logs."""
    named = [(n, p) for n, p in model.named_parameters() if p.requires_grad]
    opt = torch.optim.Adam(model.parameters(), lr=lr)
    fisher, optpar = None, None
    log: Dict[str, List[float]] = {"task_loss": [], "ewc_loss": []}
    for _, batches in tasks:
        for _ in range(epochs):
            model.train()
            for inputs, targets in batches:
                opt.zero_grad()
                task_loss = F.mse_loss(model(inputs), targets)
                ewc_loss = sr_oracle.ewc_penalty(named, fisher, optpar, lam) if fisher else 0.0
                (task_loss + ewc_loss).backward()
                opt.step()
                log["task_loss"].append(float(task_loss.item()))
                log["ewc_loss"].append(float(ewc_loss.item()) if torch.is_tensor(ewc_loss) else float(ewc_loss))
        f_new = sr_oracle.ewc_fisher(model, batches, named)          # leaves the model in eval(), as the reference does
        fisher = sr_oracle.ewc_online_merge(fisher, f_new, decay)
        optpar = {n: p.detach().clone() for n, p in named}
    log["fisher"] = fisher
    return log
