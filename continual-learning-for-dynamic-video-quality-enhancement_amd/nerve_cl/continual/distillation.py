"""Teacher/student MSE distillation helper kept importable for train_continual.py:12 (never called by
the scripts).  Works on any module that survives copy.deepcopy."""
from __future__ import annotations

import copy
from typing import Dict, Optional

import torch
import torch.nn as nn
import torch.nn.functional as F


class DistillationLoss(nn.Module):
    def __init__(self, temperature: float = 4.0, alpha: float = 0.5):
        super().__init__()
        self.temperature, self.alpha = temperature, alpha

    def forward(self, student_output, teacher_output, target: Optional[torch.Tensor] = None):
        d = F.mse_loss(student_output, teacher_output.detach())
        if target is None:
            return d
        return self.alpha * d + (1 - self.alpha) * F.mse_loss(student_output, target)


class ContinualDistillation:
    def __init__(self, model: nn.Module, temperature: float = 4.0, alpha: float = 0.5):
        self.student, self.teacher = model, None
        self.distill_loss = DistillationLoss(temperature, alpha)
        self.task_count = 0

    def register_task(self) -> None:
        self.teacher = copy.deepcopy(self.student).eval()
        for p in self.teacher.parameters():
            p.requires_grad_(False)
        self.task_count += 1

    def compute_loss(self, inputs, targets, task_loss_fn) -> Dict[str, torch.Tensor]:
        out = self.student(inputs)
        task = task_loss_fn(out, targets)
        losses = {"task": task, "distill": torch.zeros((), device=out.device), "total": task}
        if self.teacher is not None:
            with torch.no_grad():
                t_out = self.teacher(inputs)
            losses["distill"] = self.distill_loss(out, t_out, targets)
            losses["total"] = task + losses["distill"]
        return losses
