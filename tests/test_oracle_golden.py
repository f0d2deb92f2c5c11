"""CPU: the oracle restatement against the fixtures captured from the reference
(tests/golden, written by oracle/make_goldens.py).  This is what pins the oracle."""
import glob
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import sr_oracle, synth
from oracle.make_goldens import grad_summary


def _rel(a, b):
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    return np.abs(a - b).max() / max(np.abs(b).max(), 1e-30)


CASES = sorted(glob.glob(os.path.join(os.path.dirname(__file__), "golden", "sr_*.npz")))


def test_fixtures_present():
    assert len(CASES) >= 4


@pytest.mark.parametrize("path", CASES, ids=[os.path.basename(p)[3:-4] for p in CASES])
def test_forward_backward_matches_reference_fixture(path):
    g = np.load(path)
    Fc, N, win, s, B, H, W, train = [int(v) for v in g["cfg"]]
    T = 2 * win + 1
    sd = synth.formula_state(3, s, Fc, N, win, gain=synth.GOLDEN_GAIN)
    x = synth.formula_clip(B, T, H, W)
    tgt = synth.formula_target(B, H * s, W * s)
    m = sr_oracle.OracleSR(3, s, Fc, N, win)
    m.load_named(sd)
    m.train(bool(train))
    out, inter = m(x, return_intermediate=True)
    loss = F.mse_loss(out, tgt)
    loss.backward()
    assert out.shape == (B, 3, H * s, W * s)
    assert _rel(out.detach().numpy(), g["output"]) < 1e-5
    assert abs(loss.item() - float(g["loss"])) < 1e-6
    assert _rel(inter["features"][0].detach().numpy(), g["feat0"]) < 1e-5
    assert _rel(inter["aligned"][0].detach().numpy(), g["aligned0"]) < 2e-5
    assert _rel(inter["aggregated"].detach().numpy(), g["aggregated"]) < 2e-5
    named = m.named()
    for key in g.files:
        if key.startswith("gsum/"):
            n = key[5:]
            ref = g[key]
            got = grad_summary(named[n].grad)
            # l2 norm to 1e-4 rel, samples relative to the tensor's max sample
            assert abs(got[1] - ref[1]) <= 1e-4 * max(ref[1], 1e-12), n
            assert np.abs(got[2:] - ref[2:]).max() <= 1e-4 * max(np.abs(ref[2:]).max(), ref[1] * 1e-2), n
        elif key.startswith("gfull/"):
            n = key[6:]
            assert _rel(named[n].grad.numpy(), g[key]) < 1e-4, n
        elif key.startswith("buf/"):
            n = key[4:]
            assert _rel(named[n].double().numpy(), g[key]) < 1e-5, n


def test_baseline_trajectory():
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "traj_baseline.npz"))
    Fc, N, win, s, B, H, W = [int(v) for v in g["cfg"]]
    sd = synth.formula_state(3, s, Fc, N, win, gain=synth.GOLDEN_GAIN)
    lr = synth.formula_clip(B, 1, H, W, seed=5)[:, 0]
    hr = synth.formula_target(B, H * s, W * s, seed=7)
    m = sr_oracle.OracleSR(3, s, Fc, N, win)
    m.load_named(sd)
    opt = torch.optim.AdamW(m.parameters(), lr=1e-3, weight_decay=1e-5)
    m.train()
    losses = []
    for _ in range(3):
        opt.zero_grad()
        loss = F.mse_loss(m(lr.unsqueeze(1).expand(-1, 3, -1, -1, -1)), hr)
        loss.backward()
        opt.step()
        losses.append(loss.item())
    assert np.allclose(losses, g["losses"], rtol=2e-5)
    m.eval()
    with torch.no_grad():
        out = m(lr.unsqueeze(1).expand(-1, 3, -1, -1, -1))
    assert np.abs(out.numpy() - g["eval_output"]).max() < 2e-4
    assert abs(sr_oracle.compute_psnr(out, hr) - float(g["psnr"])) < 1e-3


def test_param_inventory_counts():
    # 131 tensors / 1 987 283 params (cfg2), 83 / 820 339 (cfg1), 131 / 2 082 937 (cfg4)
    for kw, n_t, n_p in (
        (dict(num_features=64, num_residual_blocks=8), 131, 1987283),
        (dict(num_features=32, num_residual_blocks=4), 83, 820339),
        (dict(num_features=64, num_residual_blocks=8, temporal_window=2, scale_factor=4), 131, 2082937),
    ):
        shp = sr_oracle.param_shapes(**kw)
        assert len(shp) == n_t
        assert sum(int(np.prod(v)) for v in shp.values()) == n_p


def test_bicubic_phase_weights():
    # separable Keys cubic A=-0.75 phase weights for s=2
    x = torch.zeros(1, 1, 1, 9)
    x[0, 0, 0, 4] = 1.0
    y = sr_oracle.bicubic_up(x, 2)[0, 0, 0]
    # output 2*4+0 = 8 uses phase0 weights [-0.03515625, 0.26171875, 0.87890625, -0.10546875] on taps 2..5
    assert abs(y[8].item() - 0.87890625) < 1e-7
    assert abs(y[9].item() - 0.87890625) < 1e-7
    assert abs(y[10].item() - 0.26171875) < 1e-7
    assert abs(y[7].item() - 0.26171875) < 1e-7


LIGHT = sorted(glob.glob(os.path.join(os.path.dirname(__file__), "golden", "light_*.npz")))


@pytest.mark.parametrize("path", LIGHT, ids=[os.path.basename(p)[:-4] for p in LIGHT])
def test_light_matches_reference_fixture(path):
    """LightweightSuperResolution restatement (oracle light_forward) against the reference's numbers."""
    g = np.load(path)
    s, B, H, W, train = [int(v) for v in g["cfg"]]
    sd = synth.formula_state_light(s, gain=synth.GOLDEN_GAIN)
    x = synth.formula_clip(B, 1, H, W)[:, 0].contiguous()
    tgt = synth.formula_target(B, H * s, W * s)
    P = {k: v.clone().requires_grad_(v.is_floating_point() and "running" not in k) for k, v in sd.items()}
    out = sr_oracle.light_forward(P, x, bool(train))
    loss = F.mse_loss(out, tgt)
    loss.backward()
    assert _rel(out.detach().numpy(), g["output"]) < 1e-5
    assert abs(loss.item() - float(g["loss"])) < 1e-6
    for key in g.files:
        if key.startswith("gsum/"):
            ref, got = g[key], grad_summary(P[key[5:]].grad)
            assert abs(got[1] - ref[1]) <= 1e-4 * max(ref[1], 1e-12), key
            assert np.abs(got[2:] - ref[2:]).max() <= 1e-4 * max(np.abs(ref[2:]).max(), ref[1] * 1e-2), key
        elif key.startswith("gfull/"):
            assert _rel(P[key[6:]].grad.numpy(), g[key]) < 1e-4, key
        elif key.startswith("buf/"):
            assert _rel(P[key[4:]].double().numpy(), g[key]) < 1e-5, key


FR = sorted(glob.glob(os.path.join(os.path.dirname(__file__), "golden", "fr_b*.npz")))


@pytest.mark.parametrize("path", FR, ids=[os.path.basename(p)[:-4] for p in FR])
def test_frame_recovery_oracle_matches_reference_fixture(path):
    """oracle/fr_oracle.py (FrameRecoveryNet restatement, groundwork for SURVEY 8f row 1) against the reference's numbers:
    output, loss, every parameter's gradient summary and the small BatchNorm buffers after one training-mode call."""
    from oracle import fr_oracle
    g = np.load(path)
    base, B, T, H, W, train = [int(v) for v in g["cfg"]]
    sd = synth.formula_state_fr(3, base, gain=synth.GOLDEN_GAIN)
    shapes, buffers = fr_oracle.shapes(3, base)
    assert set(sd) == set(shapes) | set(buffers) and all(tuple(sd[n].shape) == tuple(s) for n, s in shapes.items())
    clip = synth.formula_clip(B, T + 1, H, W)
    corrupted, refs = clip[:, 0].contiguous(), clip[:, 1:].contiguous()
    mask = torch.zeros(B, 1, H, W)
    mask[:, :, H // 4:H // 4 + H // 2, W // 3:W // 3 + W // 2] = 1.0
    corrupted = corrupted * (1 - mask)
    tgt = synth.formula_target(B, H, W)
    P = {k: v.clone().requires_grad_(v.is_floating_point() and "running" not in k) for k, v in sd.items()}
    out = fr_oracle.frame_recovery_forward(P, corrupted, refs, mask, bool(train))
    loss = F.mse_loss(out, tgt)
    loss.backward()
    assert _rel(out.detach().numpy(), g["output"]) < 2e-5
    assert abs(loss.item() - float(g["loss"])) < 1e-6
    assert torch.equal(out * (1 - mask), corrupted * (1 - mask))          # uncorrupted pixels pass through (:439)
    seen = 0
    for key in g.files:
        if key.startswith("gsum/"):
            ref, got = g[key], grad_summary(P[key[5:]].grad)
            assert abs(got[1] - ref[1]) <= 2e-4 * max(ref[1], 1e-12), key
            assert np.abs(got[2:] - ref[2:]).max() <= 2e-4 * max(np.abs(ref[2:]).max(), ref[1] * 1e-2), key
            seen += 1
        elif key.startswith("buf/"):
            assert _rel(P[key[4:]].double().numpy(), g[key]) < 1e-5, key
    assert seen == len(shapes)


# ---------------------------------------------------------------- continual-learning consumers
def _summ_close(got, ref, what, tol):
    """summary vectors [sum, l2, 16 samples] at the scale of the tensor's own l2"""
    err = np.abs(np.asarray(got) - np.asarray(ref)).max() / max(abs(ref[1]), 1e-30)
    assert err <= tol, f"{what}: {err:.3e}"


def _oracle_net():
    from oracle import cl_cases
    c = cl_cases.META
    m = sr_oracle.OracleSR(3, c["s"], c["F"], c["N"], c["win"])
    m.load_named(cl_cases.state())
    return m.train()


def test_meta_learning_oracle_matches_reference_fixture(golden_dir):
    """oracle/cl_oracle.py's FOMAML.adapt / Reptile.train_step restatements against the parameter changes the reference's
    own classes produced (tests/golden/meta_f16.npz)."""
    from oracle import cl_cases, cl_oracle
    from oracle.make_goldens import delta_summaries
    g = np.load(os.path.join(golden_dir, "meta_f16.npz"))
    fo, rp = cl_cases.FOMAML, cl_cases.REPTILE
    m = _oracle_net()
    names = [k[len("fomaml_delta/"):] for k in g.files if k.startswith("fomaml_delta/")]
    assert len(names) == 47
    before = {n: m.named()[n].detach().clone() for n in names}
    ad = cl_oracle.fomaml_adapt(m, cl_cases.clip_pair(fo["data_seed"]), F.mse_loss, fo["inner_lr"], fo["steps"])
    assert all(torch.equal(m.named()[n], before[n]) for n in names)
    d = delta_summaries(before, {n: ad.named()[n] for n in names})
    for n in names:
        _summ_close(d[n], g["fomaml_delta/" + n], "fomaml " + n, 2e-3)
    ad.eval()
    with torch.no_grad():
        x, t = cl_cases.clip_pair(fo["data_seed"])
        assert abs(F.mse_loss(ad(x), t).item() - float(g["fomaml_eval_loss"])) < 1e-6
    for k in g.files:
        if k.startswith("fomaml_buf/"):
            assert _rel(ad.named()[k[len("fomaml_buf/"):]].double().numpy(), g[k]) < 1e-5, k

    m = _oracle_net()
    before = {n: m.named()[n].detach().clone() for n in names}
    tasks = [{"support": cl_cases.clip_pair(s)} for s in rp["data_seeds"]]
    loss = cl_oracle.reptile_train_step(m, tasks, F.mse_loss, rp["inner_lr"], rp["outer_lr"], rp["inner_steps"])
    assert abs(loss - float(g["reptile_loss"])) < 1e-6
    d = delta_summaries(before, {n: m.named()[n] for n in names})
    for n in names:
        _summ_close(d[n], g["reptile_delta/" + n], "reptile " + n, 2e-3)
    for k in g.files:
        if k.startswith("reptile_buf/"):
            assert _rel(m.named()[k[len("reptile_buf/"):]].double().numpy(), g[k]) < 1e-5, k


def test_synaptic_intelligence_oracle_matches_reference_fixture(golden_dir):
    from oracle import cl_cases, cl_oracle
    g = np.load(os.path.join(golden_dir, "si_f16.npz"))
    m = _oracle_net()
    si = cl_oracle.SI(m, cl_cases.SI["si_lambda"], cl_cases.SI["damping"])
    names = [k[2:] for k in g.files if k.startswith("W/")]
    key = {n: n.replace(".", "|") for n in names}
    x, t = cl_cases.si_pair()
    W, omega, pens, losses = cl_cases.si_drive(m, si, [key[n] for n in names], x, t)
    assert np.allclose(losses, g["losses"], rtol=1e-5)
    assert pens[:3] == [0.0, 0.0, 0.0] and list(g["penalties"][:3]) == [0.0, 0.0, 0.0]     # the p_old quirk (cl_cases.si_drive)
    assert np.allclose(pens[3:], g["penalties"][3:], rtol=2e-3) and g["penalties"][4] > 1e-4
    for n in names:
        _summ_close(grad_summary(W[key[n]]), g["W/" + n], "W " + n, 5e-3)
        _summ_close(grad_summary(omega[key[n]]), g["omega/" + n], "omega " + n, 5e-3)


def test_train_with_ewc_loop_oracle_matches_reference_fixture(golden_dir):
    """The `mse + penalty` optimisation trajectory of experiments/train_continual.py:26-69: per-step
    task_loss / ewc_loss captured from reference EWC + reference SR net through the 4-D -> 5-D adapter."""
    from oracle import cl_cases, cl_oracle
    from oracle.make_goldens import Adapter4D
    g = np.load(os.path.join(golden_dir, "cfg5_loop.npz"))
    m = Adapter4D(_oracle_net())
    log = cl_oracle.train_with_ewc(m, cl_cases.cfg5_tasks(), cl_cases.CFG5["lam"], cl_cases.CFG5["lr"], 1)
    assert np.allclose(log["task_loss"], g["task_loss"], rtol=1e-5)
    assert list(g["ewc_loss"][:4]) == [0.0] * 4 and log["ewc_loss"][:4] == [0.0] * 4
    assert np.allclose(log["ewc_loss"][4:], g["ewc_loss"][4:], rtol=2e-3) and g["ewc_loss"][5] > 1e-6
    for k in g.files:
        if k.startswith("fisher/"):
            _summ_close(grad_summary(log["fisher"]["net." + k[7:].replace(".", "|")]), g[k], k, 1e-3)
