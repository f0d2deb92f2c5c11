"""TEST INFRASTRUCTURE ONLY: the closed-form inputs of the continual-learning fixtures (tests/golden/meta_f16.npz,
si_f16.npz, cfg5_loop.npz), shared by oracle/make_goldens.py (which drives the reference with them) and the tests (which
drive the oracle and the HIP path with them)."""
from __future__ import annotations

from oracle import synth

META = dict(F=16, N=1, win=1, s=2, H=8, W=12)
FOMAML = dict(inner_lr=0.05, steps=3, data_seed=61)
REPTILE = dict(inner_lr=0.05, outer_lr=0.5, inner_steps=2, data_seeds=(71, 72))
SI = dict(si_lambda=2000.0, damping=0.1, lr=0.05)
CFG5 = dict(lam=5000.0, lr=1e-4, offsets=(0.2, -0.2), per_task=6, batch=2)


def state():
    c = META
    return synth.formula_state(3, c["s"], c["F"], c["N"], c["win"], gain=synth.GOLDEN_GAIN)


def clip_pair(seed: int, B: int = 2):
    """(5-D clip, target) for the meta-learning cases"""
    c = META
    return (synth.formula_clip(B, 3, c["H"], c["W"], seed=seed),
            synth.formula_target(B, c["H"] * c["s"], c["W"] * c["s"], seed=seed + 50))


def si_pair():
    c = META
    return (synth.formula_clip(2, 3, c["H"], c["W"], seed=81),
            synth.formula_target(2, c["H"] * c["s"], c["W"] * c["s"], seed=82))


def cfg5_tasks():
    """[(name, [(lr 4-D, hr), ...])]: create_task_data-style offsets (reference train_continual.py:15-23) on formula data"""
    c = META
    tasks = []
    n, b = CFG5["per_task"], CFG5["batch"]
    for k, off in enumerate(CFG5["offsets"]):
        lr = synth.formula_clip(n, 1, c["H"], c["W"], seed=91 + k)[:, 0] + off
        hr = synth.formula_target(n, c["H"] * c["s"], c["W"] * c["s"], seed=95 + k) + off
        tasks.append((f"task{k}", [(lr[i:i + b], hr[i:i + b]) for i in range(0, n, b)]))
    return tasks


def si_drive(model, si, named, x, t, loss_fn=None):
    """The SynapticIntelligence fixture's step sequence (the SAME code drives the reference, the oracle and the HIP path):
    3 SGD steps + update_importance, register_task, 2 steps of loss + penalty with update_importance, 2 without.  Returns
    (W before register_task, omega after it, the penalty series, the loss series)."""
    import torch
    import torch.nn.functional as F
    loss_fn = loss_fn or F.mse_loss
    opt = torch.optim.SGD(model.parameters(), lr=SI["lr"])
    pens, losses = [], []
    model.train()
    for _ in range(3):
        opt.zero_grad()
        loss = loss_fn(model(x), t)
        loss.backward()
        opt.step()
        si.update_importance()
        losses.append(loss.item())
    W = {n: si.W[n].detach().clone() for n in named}
    si.register_task()
    omega1 = {n: si.omega[n].detach().clone() for n in named}
    for _ in range(2):
        opt.zero_grad()
        pen = si.penalty()
        loss = loss_fn(model(x), t)
        (loss + pen).backward()
        opt.step()
        si.update_importance()
        pens.append(float(pen.item()) if torch.is_tensor(pen) else float(pen))
        losses.append(loss.item())
    # (so far every penalty is exactly 0: update_importance() moves p_old to the current parameters after EVERY step,
    # ewc.py:351 - the reference's p_old is both "previous step" and "task start".  Two more steps WITHOUT
    # update_importance, so that the penalty and its gradient are non-zero and shape the trajectory.)
    for _ in range(2):
        opt.zero_grad()
        pen = si.penalty()
        loss = loss_fn(model(x), t)
        (loss + pen).backward()
        opt.step()
        pens.append(float(pen.item()))
        losses.append(loss.item())
    with torch.no_grad():
        pens.append(float(si.penalty().item()))
    return W, omega1, pens, losses

