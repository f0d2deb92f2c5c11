"""GPU: the callers either side of the hot path running on the HIP networks:
 * experiments/train_baseline.py end to end on a tiny dataset: the best-checkpoint dictionary of the reference
   (experiments/train_baseline.py:124-129: epoch / model_state_dict / optimizer_state_dict / psnr) with the reference's
   state_dict keys, loadable with weights_only=True;
 * experiments/train_continual.py --strategy replay (reference :72-112): replay batches concatenated to the task batch
   (16 -> 24 samples), EpisodicMemory in the loop, the saved engine state_dict;
 * FOMAML / Reptile / ContinualDistillation (reference maml.py:74-110,276-345, distillation.py:48-71) on a
   SuperResolutionNet living on the GPU: deepcopy + forward / backward through the HIP path."""
import os
import subprocess
import sys

import pytest
import torch
import torch.nn as nn

pytestmark = pytest.mark.gpu
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXP = os.path.join(REPO, "experiments")


def _run(args, cwd):
    env = dict(os.environ, OMP_NUM_THREADS="2")
    r = subprocess.run([sys.executable] + args, cwd=cwd, env=env, capture_output=True, text=True, timeout=500)
    assert r.returncode == 0, r.stdout[-1500:] + r.stderr[-3000:]
    return r.stdout


@pytest.mark.timeout(600)
def test_train_baseline_script_writes_the_reference_checkpoint_format(tmp_path):
    from oracle import sr_oracle
    _run([os.path.join(EXP, "make_dummy_data.py"), "--out", "data", "--train", "32", "--val", "16", "--dist", "rand"], tmp_path)
    out = _run([os.path.join(EXP, "train_baseline.py"), "--epochs", "2", "--batch-size", "16"], tmp_path)
    assert "Epoch   2/2" in out and "Training complete!" in out and "Val PSNR" in out
    ck = torch.load(tmp_path / "checkpoints" / "best_model.pt", weights_only=True)
    assert set(ck) == {"epoch", "model_state_dict", "optimizer_state_dict", "psnr"}
    want = dict(sr_oracle.param_shapes(3, 2, 32, 4, 1))
    want.update(sr_oracle.buffer_shapes(32))
    got = {k: tuple(v.shape) for k, v in ck["model_state_dict"].items()}
    assert got == {k: tuple(v) for k, v in want.items()}      # the reference module's keys and shapes (F=32, 4 blocks)
    assert ck["psnr"] > 0 and set(ck["optimizer_state_dict"]) == {"state", "param_groups"}
    ora = sr_oracle.OracleSR(3, 2, 32, 4, 1)
    ora.load_named(ck["model_state_dict"])                    # loads into the oracle's (= the reference's) parameter names


@pytest.mark.timeout(600)
def test_train_continual_replay_strategy(tmp_path):
    out = _run([os.path.join(EXP, "train_continual.py"), "--strategy", "replay", "--tasks", "2", "--samples", "24",
                "--epochs", "2", "--features", "16", "--blocks", "1", "--memory-size", "40"], tmp_path)
    assert "=== Training on Task 1: animation ===" in out and "Memory size: 40" in out and "Training complete!" in out
    sd = torch.load(tmp_path / "checkpoints" / "continual_model.pt", weights_only=True)
    assert "enhancement_strength" in sd and any(k.startswith("super_resolution.residual_blocks.0.") for k in sd)


def _small_net():
    from nerve_cl.models import SuperResolutionNet
    torch.manual_seed(0)
    return SuperResolutionNet(3, 2, 16, 1, 1).cuda()


def test_fomaml_reptile_and_distillation_on_the_hip_network():
    from nerve_cl.continual import ContinualDistillation, FOMAML, Reptile
    net = _small_net()
    g = torch.Generator().manual_seed(3)
    x, y = torch.rand(4, 3, 3, 16, 16, generator=g).cuda(), torch.rand(4, 3, 32, 32, generator=g).cuda()
    crit = nn.MSELoss()
    before = crit(net(x), y).item()
    maml = FOMAML(net, inner_lr=0.05, inner_steps=3)
    w_meta = [p.detach().clone() for p in net.parameters()]
    adapted = maml.adapt((x, y), crit)
    assert adapted is not net and next(adapted.parameters()).is_cuda
    assert crit(adapted(x), y).item() < before                # three SGD steps on the support set, on a deep copy
    assert all(torch.equal(a, b) for a, b in zip(w_meta, net.parameters()))     # the meta-parameters are untouched
    tasks = [{"support": (x[:2], y[:2]), "query": (x[2:], y[2:])}]
    assert maml.train_step(tasks, crit) > 0                   # first-order meta-gradient applied by the meta-optimizer
    rep = Reptile(net, inner_lr=0.05, outer_lr=0.5, inner_steps=2)
    w0 = [p.detach().clone() for p in net.parameters()]
    rep.train_step(tasks, crit)
    assert any(not torch.equal(a, b) for a, b in zip(w0, net.parameters()))
    cd = ContinualDistillation(net)
    assert float(cd.compute_loss(x, y, crit)["distill"]) == 0.0     # no teacher before the first task
    cd.register_task()
    losses = cd.compute_loss(x, y, crit)
    losses["total"].backward()
    assert float(losses["distill"]) >= 0.0 and all(p.grad is not None for p in net.parameters())
