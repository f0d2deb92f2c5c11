"""GPU: the continual-learning consumers of the SR hot path through the PRODUCT classes on the HIP
network, against numbers captured from the reference's own classes (oracle/make_goldens.py --only-cl):

 * nerve_cl.continual.FOMAML.adapt / Reptile.train_step vs tests/golden/meta_f16.npz (reference maml.py:74-110,168-186,276-345),
 * nerve_cl.continual.SynapticIntelligence (flat-bucket kernels nvq_si_update / nvq_si_consolidate) vs si_f16.npz
   (reference ewc.py:306-379),
 * the `mse + ewc.penalty()` loop of experiments/train_continual.py:26-69 with the product EWC (penalty gradient fused into the
   gradient bucket) vs cfg5_loop.npz: per-step task_loss / ewc_loss and the final Fisher,
 * experiments/train_continual.py --strategy ewc as a process.
The inputs are the closed-form ones of oracle/cl_cases.py; exact-fp32 kernels (the default math mode)."""
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

from oracle import cl_cases
from oracle.make_goldens import Adapter4D, delta_summaries, grad_summary

pytestmark = pytest.mark.gpu
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _net():
    from nerve_cl.models import SuperResolutionNet
    c = cl_cases.META
    net = SuperResolutionNet(3, c["s"], c["F"], c["N"], c["win"])
    net.load_state_dict(cl_cases.state())
    return net.cuda().train()


def _cuda(pair):
    return pair[0].cuda(), pair[1].cuda()


def _summ_close(got, ref, what, tol):
    err = np.abs(np.asarray(got) - np.asarray(ref)).max() / max(abs(ref[1]), 1e-30)
    assert err <= tol, f"{what}: {err:.3e}"


def _mse():
    from nerve_cl import ops
    return ops.mse_loss                 # the loops' nn.MSELoss as libnvq kernels


def test_fomaml_adapt_and_reptile_train_step_match_the_reference(golden_dir):
    from nerve_cl.continual import FOMAML, Reptile
    g = np.load(os.path.join(golden_dir, "meta_f16.npz"))
    fo, rp = cl_cases.FOMAML, cl_cases.REPTILE
    names = [k[len("fomaml_delta/"):] for k in g.files if k.startswith("fomaml_delta/")]
    net = _net()
    before = {n: p.detach().clone() for n, p in net.named_parameters()}
    assert sorted(before) == sorted(names)
    data = _cuda(cl_cases.clip_pair(fo["data_seed"]))
    adapted = FOMAML(net, inner_lr=fo["inner_lr"], inner_steps=fo["steps"]).adapt(data, _mse())
    assert adapted is not net and all(torch.equal(p, before[n]) for n, p in net.named_parameters())
    d = delta_summaries({n: v.cpu() for n, v in before.items()}, {n: p.cpu() for n, p in adapted.named_parameters()})
    for n in names:
        _summ_close(d[n], g["fomaml_delta/" + n], "fomaml " + n, 5e-3)
    adapted.eval()
    with torch.no_grad():
        assert abs(_mse()(adapted(data[0]), data[1]).item() - float(g["fomaml_eval_loss"])) < 2e-5
    sd = adapted.state_dict()
    for k in g.files:
        if k.startswith("fomaml_buf/"):
            ref = g[k].astype(np.float64)
            assert np.abs(sd[k[len("fomaml_buf/"):]].cpu().double().numpy() - ref).max() <= 1e-4 * max(np.abs(ref).max(), 1e-30), k

    net = _net()
    before = {n: p.detach().clone().cpu() for n, p in net.named_parameters()}
    tasks = [{"support": _cuda(cl_cases.clip_pair(s))} for s in rp["data_seeds"]]
    loss = Reptile(net, inner_lr=rp["inner_lr"], outer_lr=rp["outer_lr"], inner_steps=rp["inner_steps"]).train_step(tasks, _mse())
    assert abs(loss - float(g["reptile_loss"])) < 2e-5
    d = delta_summaries(before, {n: p.cpu() for n, p in net.named_parameters()})
    for n in names:
        _summ_close(d[n], g["reptile_delta/" + n], "reptile " + n, 5e-3)
    sd = net.state_dict()
    for k in g.files:
        if k.startswith("reptile_buf/"):
            ref = g[k].astype(np.float64)
            assert np.abs(sd[k[len("reptile_buf/"):]].cpu().double().numpy() - ref).max() <= 1e-4 * max(np.abs(ref).max(), 1e-30), k


def test_synaptic_intelligence_on_the_flat_bucket_matches_the_reference(golden_dir):
    from nerve_cl.continual import SynapticIntelligence
    g = np.load(os.path.join(golden_dir, "si_f16.npz"))
    net = _net()
    si = SynapticIntelligence(net, si_lambda=cl_cases.SI["si_lambda"], damping=cl_cases.SI["damping"])
    names = [k[2:] for k in g.files if k.startswith("W/")]
    assert sorted(si.W) == sorted(names)
    x, t = _cuda(cl_cases.si_pair())
    W, omega, pens, losses = cl_cases.si_drive(net, si, names, x, t, _mse())
    assert np.allclose(losses, g["losses"], rtol=2e-5)
    assert pens[:3] == [0.0, 0.0, 0.0]                         # the reference's p_old quirk, kept
    assert np.allclose(pens[3:], g["penalties"][3:], rtol=1e-2)
    for n in names:
        _summ_close(grad_summary(W[n].cpu()), g["W/" + n], "W " + n, 1e-2)
        _summ_close(grad_summary(omega[n].cpu()), g["omega/" + n], "omega " + n, 1e-2)
    # after a plain zero_grad -> backward -> step the .grad tensors ARE the views of the network's gradient bucket, i.e.
    # update_importance is the one-launch path (a backward that also accumulated a separately returned penalty gradient,
    # like the last steps of si_drive, replaces .grad by a sum and takes the per-tensor path)
    opt = torch.optim.SGD(net.parameters(), lr=0.01)
    opt.zero_grad()
    _mse()(net(x), t).backward()
    opt.step()
    assert si._bucket_is_grad(si._segs[0])
    w0 = si._W[0].clone()
    si.update_importance()
    assert not torch.equal(w0, si._W[0]) and torch.equal(si._p_old[0], net.flat_theta())


def test_train_with_ewc_loop_matches_the_reference_trajectory(golden_dir):
    """experiments/train_continual.py:26-69 with reference semantics: Adam(1e-4), loss = mse + penalty, register_task per
    task - the product EWC around the HIP network through the 4-D -> 5-D adapter."""
    from nerve_cl.continual import EWC
    g = np.load(os.path.join(golden_dir, "cfg5_loop.npz"))
    model = Adapter4D(_net())
    ewc = EWC(model, ewc_lambda=cl_cases.CFG5["lam"])
    opt = torch.optim.Adam(model.parameters(), lr=cl_cases.CFG5["lr"])
    crit = _mse()
    tl, el = [], []
    for task_id, (_, batches) in enumerate(cl_cases.cfg5_tasks()):
        batches = [(a.cuda(), b.cuda()) for a, b in batches]
        model.train()
        for lr_b, hr_b in batches:
            opt.zero_grad()
            task_loss = crit(model(lr_b), hr_b)
            ewc_loss = ewc.penalty()
            (task_loss + ewc_loss).backward()
            opt.step()
            tl.append(task_loss.item())
            el.append(float(ewc_loss.item()) if torch.is_tensor(ewc_loss) else float(ewc_loss))
        ewc.register_task(task_id, batches)
    assert np.allclose(tl, g["task_loss"], rtol=5e-5), (tl, g["task_loss"])
    assert el[:4] == [0.0] * 4
    assert np.allclose(el[4:], g["ewc_loss"][4:], rtol=2e-2), (el, g["ewc_loss"])
    for k in g.files:
        if k.startswith("fisher/"):
            _summ_close(grad_summary(ewc.fisher_dict["net." + k[7:]].cpu()), g[k], k, 5e-3)


@pytest.mark.timeout(600)
def test_train_continual_ewc_strategy_as_a_process(tmp_path):
    env = dict(os.environ, OMP_NUM_THREADS="2")
    r = subprocess.run([sys.executable, os.path.join(REPO, "experiments", "train_continual.py"), "--strategy", "ewc", "--tasks", "2",
                        "--samples", "32", "--epochs", "2", "--features", "16", "--blocks", "1"], cwd=tmp_path, env=env,
                       capture_output=True, text=True, timeout=500)
    assert r.returncode == 0, r.stdout[-1500:] + r.stderr[-3000:]
    out = r.stdout
    assert "=== Training on Task 0: sports ===" in out and "=== Training on Task 1: animation ===" in out
    assert out.count("Registered task") == 2 and "Training complete!" in out
    losses = [float(ln.split("Loss=")[1]) for ln in out.splitlines() if "Loss=" in ln]
    assert len(losses) == 4 and all(np.isfinite(losses))
    sd = torch.load(tmp_path / "checkpoints" / "continual_model.pt", weights_only=True)
    assert "enhancement_strength" in sd and any(k.startswith("super_resolution.residual_blocks.0.") for k in sd)
