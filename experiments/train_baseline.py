#!/usr/bin/env python3
"""Baseline SR training (BASELINE configs[0]): same CLI, model arguments, optimiser, loss, PSNR formula,
T=3 `expand` trick, progress line and best-checkpoint rule as the reference's
experiments/train_baseline.py (:35-147), running on libnvq.  Launch with torch.distributed.run for
data-parallel training (each rank trains on its shard; gradients are all-reduced inside backward)."""
import argparse
import time
from pathlib import Path

import _common  # noqa: F401  (sys.path)
import torch
import torch.nn as nn
from torch.utils.data import DataLoader, TensorDataset

from _common import compute_psnr, make_optimizer, pick_device, shard
from nerve_cl import ops, parallel
from nerve_cl.models import SuperResolutionNet


def load_split(data_dir: str, split: str, rank: int, world: int) -> TensorDataset:
    blob = torch.load(f"{data_dir}/{split}/data.pt")
    sl = shard(len(blob["lr"]), rank, world)
    return TensorDataset(blob["lr"][sl], blob["hr"][sl])


def as_clip(lr: torch.Tensor) -> torch.Tensor:
    return lr.unsqueeze(1).expand(-1, 3, -1, -1, -1)      # (B,3,C,H,W), stride-0 like the reference (:82)


def train(args) -> None:
    device, rank, world = pick_device()
    say = print if rank == 0 else (lambda *a, **k: None)
    say(f"Using device: {device}")
    say("Loading dataset...")
    train_set, val_set = load_split(args.data_dir, "train", rank, world), load_split(args.data_dir, "val", rank, world)
    say(f"  Train samples: {len(train_set)}")
    say(f"  Val samples: {len(val_set)}")
    train_loader = DataLoader(train_set, batch_size=args.batch_size, shuffle=True)
    val_loader = DataLoader(val_set, batch_size=args.batch_size)

    say("Creating model...")
    model = SuperResolutionNet(scale_factor=2, num_features=32, num_residual_blocks=4, temporal_window=1).to(device)
    if world > 1:
        parallel.enable_data_parallel(model)
    say(f"  Parameters: {sum(p.numel() for p in model.parameters()):,}")
    optimizer = make_optimizer(torch.optim.AdamW, model.parameters(), lr=args.lr, weight_decay=1e-5)
    scheduler = torch.optim.lr_scheduler.CosineAnnealingLR(optimizer, T_max=args.epochs)
    criterion = ops.MSELoss()       # nn.MSELoss() of the reference, as libnvq kernels

    say(f"\nTraining for {args.epochs} epochs...")
    say("-" * 60)
    best_psnr, t_start = 0, time.time()
    for epoch in range(args.epochs):
        model.train()
        running = 0.0
        for lr, hr in train_loader:
            lr, hr = lr.to(device), hr.to(device)
            optimizer.zero_grad()
            loss = criterion(model(as_clip(lr)), hr)
            loss.backward()
            optimizer.step()
            running += loss.item()
        running /= max(len(train_loader), 1)

        model.eval()
        val_loss = val_psnr = 0.0
        with torch.no_grad():
            for lr, hr in val_loader:
                lr, hr = lr.to(device), hr.to(device)
                out = model(as_clip(lr))
                val_loss += criterion(out, hr).item()
                val_psnr += compute_psnr(out, hr)
        # rank-uniform validation numbers (each rank validated its shard): every rank must take the same branch below,
        # because the branch contains collectives
        val_loss, val_psnr, nval = parallel.allreduce_scalars([val_loss, val_psnr, len(val_loader)], device=device)
        val_loss /= max(nval, 1)
        val_psnr /= max(nval, 1)
        scheduler.step()
        say(f"Epoch {epoch + 1:3d}/{args.epochs} | Train Loss: {running:.4f} | Val Loss: {val_loss:.4f} | "
            f"Val PSNR: {val_psnr:.2f} dB | Time: {time.time() - t_start:.1f}s")
        if val_psnr > best_psnr:
            best_psnr = val_psnr
            if world > 1:
                parallel.average_bn_buffers_(model)
            if rank == 0:
                torch.save({"epoch": epoch, "model_state_dict": model.state_dict(),
                            "optimizer_state_dict": optimizer.state_dict(), "psnr": best_psnr},
                           "checkpoints/best_model.pt")
    say("-" * 60)
    say("Training complete!")
    say(f"  Best PSNR: {best_psnr:.2f} dB")
    say(f"  Total time: {time.time() - t_start:.1f}s")
    say("  Model saved: checkpoints/best_model.pt")


def main() -> None:
    ap = argparse.ArgumentParser(description="Train NERVE baseline")
    ap.add_argument("--data-dir", type=str, default="data")
    ap.add_argument("--batch-size", type=int, default=16)
    ap.add_argument("--epochs", type=int, default=10)
    ap.add_argument("--lr", type=float, default=1e-3)
    args = ap.parse_args()
    Path("checkpoints").mkdir(exist_ok=True)
    train(args)


if __name__ == "__main__":
    main()
