#!/usr/bin/env python3
"""Write data/{train,val}/data.pt in the format train_baseline.py reads: dict{'lr': (N,3,64,64),
'hr': (N,3,128,128)}.  --dist randn reproduces the reference's scripts/download_data.sh:36-37 recipe
(PSNR < 0, so no checkpoint is ever written); --dist rand gives targets in [0,1]."""
import argparse
import os

import torch

ap = argparse.ArgumentParser()
ap.add_argument("--out", default="data")
ap.add_argument("--train", type=int, default=500)
ap.add_argument("--val", type=int, default=100)
ap.add_argument("--dist", choices=["rand", "randn"], default="rand")
ap.add_argument("--seed", type=int, default=0)
a = ap.parse_args()
torch.manual_seed(a.seed)
gen = torch.rand if a.dist == "rand" else torch.randn
for split, n in (("train", a.train), ("val", a.val)):
    os.makedirs(os.path.join(a.out, split), exist_ok=True)
    torch.save({"lr": gen(n, 3, 64, 64), "hr": gen(n, 3, 128, 128)}, os.path.join(a.out, split, "data.pt"))
    print(f"wrote {a.out}/{split}/data.pt ({n} samples, {a.dist})")
