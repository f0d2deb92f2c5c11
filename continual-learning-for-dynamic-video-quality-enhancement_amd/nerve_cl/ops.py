"""Differentiable single-kernel ops of the training loop that sit OUTSIDE the network modules.

``mse_loss`` / ``MSELoss`` replace ``nn.MSELoss()`` / ``F.mse_loss`` of the reference's loops
(experiments/train_baseline.py:64,86, train_continual.py:31,55, nerve_cl/continual/ewc.py:125): mean over all elements,
gradient 2 (x - y) / numel w.r.t. the prediction only (the target carries no gradient in any caller).
HIP tensors only; there is no CPU fallback.
"""
from __future__ import annotations

import torch
import torch.nn as nn

from nerve_cl import _engine, _nvq


class _MSEFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, pred: torch.Tensor, target: torch.Tensor):
        _nvq.require_device(pred, "prediction")
        _nvq.require_device(target, "target")
        if pred.shape != target.shape:
            raise RuntimeError(f"mse_loss: shapes differ, {tuple(pred.shape)} vs {tuple(target.shape)}")
        a, b = pred.detach().float().contiguous(), target.detach().float().contiguous()
        out = torch.empty(1, dtype=torch.float32, device=a.device)
        with _nvq.device_guard(a.device):
            _nvq.mse_forward(a, b, out, _engine.workspace(a.device))
        ctx.save_for_backward(a, b)
        ctx.shape = pred.shape
        return out.reshape(())

    @staticmethod
    def backward(ctx, go):
        a, b = ctx.saved_tensors
        da = torch.empty_like(a)
        with _nvq.device_guard(a.device):
            _nvq.mse_backward(a, b, go.detach().float().reshape(1).contiguous(), da)
        return da.view(ctx.shape), None


def mse_loss(pred: torch.Tensor, target: torch.Tensor) -> torch.Tensor:
    """mean((pred - target)^2) as two libnvq launches forward and one backward."""
    return _MSEFn.apply(pred, target)


class MSELoss(nn.Module):
    """Drop-in for ``nn.MSELoss()`` (mean reduction) on HIP tensors."""

    def forward(self, pred: torch.Tensor, target: torch.Tensor) -> torch.Tensor:
        return mse_loss(pred, target)
