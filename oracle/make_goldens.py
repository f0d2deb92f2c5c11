"""Generate tests/golden/*.npz by running the REFERENCE implementation.

Run in the build container only (needs /root/reference on disk; the reference
never travels to the GPU box):

    python -m oracle.make_goldens

For every case it (1) builds the reference ``SuperResolutionNet`` / ``EWC``
(imported from /root/reference, nothing copied), loads the formula weights of
``oracle.synth``, runs forward + backward on formula inputs, (2) runs the oracle
restatement on the same data and asserts agreement, (3) stores the reference's
numbers.  Fixtures hold data only: outputs, selected intermediates, per-tensor
gradient summaries, BN running statistics, loss trajectories, EWC quantities.
"""
from __future__ import annotations

import os
import sys

import numpy as np
import torch
import torch.nn.functional as F

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = os.environ.get("NVQ_REFERENCE", "/root/reference")
OUT = os.path.join(REPO, "tests", "golden")

sys.path.insert(0, REPO)
from oracle import sr_oracle, synth  # noqa: E402

CASES = {
    # name: (F, N, window, scale, B, H, W, train)
    "a_f32n2_t3_s2_train": (32, 2, 1, 2, 2, 12, 16, True),
    "b_f16n1_t5_s4_train": (16, 1, 2, 4, 1, 10, 12, True),
    "c_f32n1_t3_s3_eval": (32, 1, 1, 3, 1, 9, 11, False),
    "d_f64n1_t3_s2_train": (64, 1, 1, 2, 1, 8, 40, True),
}
SAMPLE_IDX = 16
GAIN = synth.GOLDEN_GAIN


def grad_summary(g: torch.Tensor) -> np.ndarray:
    """[sum, l2, 16 strided samples] of a gradient tensor, float64."""
    flat = g.detach().double().reshape(-1)
    n = flat.numel()
    idx = (np.arange(SAMPLE_IDX) * max(n // SAMPLE_IDX, 1)) % n
    return np.concatenate([[flat.sum().item(), flat.norm().item()], flat[idx].numpy()])


def import_reference():
    sys.path.insert(0, REF)
    import nerve_cl  # noqa: F401  (the reference package)
    from nerve_cl.models import SuperResolutionNet
    from nerve_cl.continual import EWC
    assert os.path.abspath(nerve_cl.__file__).startswith(os.path.abspath(REF)), nerve_cl.__file__
    return SuperResolutionNet, EWC


def run_case(name, cfg, RefSR):
    Fc, N, win, s, B, H, W, train = cfg
    T = 2 * win + 1
    sd = synth.formula_state(3, s, Fc, N, win, gain=GAIN)
    x = synth.formula_clip(B, T, H, W)
    tgt = synth.formula_target(B, H * s, W * s)

    ref = RefSR(3, s, Fc, N, win)
    ref.load_state_dict(sd, strict=True)
    ref.train(train)
    out, inter = ref(x, return_intermediate=True)
    loss = F.mse_loss(out, tgt)
    loss.backward()

    ora = sr_oracle.OracleSR(3, s, Fc, N, win)
    ora.load_named(sd)
    ora.train(train)
    o_out, o_inter = ora(x, return_intermediate=True)
    o_loss = F.mse_loss(o_out, tgt)
    o_loss.backward()

    # oracle vs reference, at generation time
    def close(a, b, what, tol=2e-5):
        a, b = a.detach().double(), b.detach().double()
        err = (a - b).abs().max().item() / max(b.abs().max().item(), 1e-30)
        assert err <= tol, f"{name}: oracle != reference for {what}: rel {err:.3e}"
        return err
    worst = close(o_out, out, "output")
    for t in range(T):
        worst = max(worst, close(o_inter["features"][t], inter["features"][t], f"features[{t}]"))
        worst = max(worst, close(o_inter["aligned"][t], inter["aligned"][t], f"aligned[{t}]"))
    worst = max(worst, close(o_inter["aggregated"], inter["aggregated"], "aggregated"))
    ref_named = dict(ref.named_parameters())
    for n, p in ora.named().items():
        if n in ref_named:
            worst = max(worst, close(p.grad, ref_named[n].grad, "grad " + n, tol=1e-4))
    ref_sd = ref.state_dict()
    for n in sr_oracle.buffer_shapes(Fc):
        worst = max(worst, close(ora.named()[n].double(), ref_sd[n].double(), "buffer " + n))

    blob = {
        "cfg": np.array([Fc, N, win, s, B, H, W, int(train)], dtype=np.int64),
        "output": out.detach().numpy(),
        "loss": np.array(loss.item(), dtype=np.float64),
        "feat0": inter["features"][0].detach().numpy(),
        "aligned0": inter["aligned"][0].detach().numpy(),
        "aggregated": inter["aggregated"].detach().numpy(),
        "clamped_frac": np.array(((out == 0) | (out == 1)).float().mean().item()),
    }
    for n, p in ref.named_parameters():
        g = p.grad
        blob["gsum/" + n] = grad_summary(g)
        if g.numel() <= 256:
            blob["gfull/" + n] = g.detach().numpy()
    for n in sr_oracle.buffer_shapes(Fc):
        blob["buf/" + n] = ref_sd[n].detach().numpy()
    np.savez_compressed(os.path.join(OUT, f"sr_{name}.npz"), **blob)
    print(f"{name}: loss {loss.item():.6f} clamped {blob['clamped_frac']:.3f} "
          f"oracle-vs-ref worst rel {worst:.2e}")


def run_trajectory(RefSR):
    """Three train_baseline-style steps (experiments/train_baseline.py:51-64,79-88):
    F=32, N=4, AdamW(lr=1e-3, wd=1e-5), MSE, input expanded to T=3 identical frames."""
    Fc, N, win, s, B, H, W = 32, 4, 1, 2, 2, 16, 16
    sd = synth.formula_state(3, s, Fc, N, win, gain=GAIN)
    lr = synth.formula_clip(B, 1, H, W, seed=5)[:, 0]
    hr = synth.formula_target(B, H * s, W * s, seed=7)

    def drive(model, steps=3):
        opt = torch.optim.AdamW(model.parameters(), lr=1e-3, weight_decay=1e-5)
        losses = []
        model.train()
        for _ in range(steps):
            opt.zero_grad()
            out = model(lr.unsqueeze(1).expand(-1, 3, -1, -1, -1))
            loss = F.mse_loss(out, hr)
            loss.backward()
            opt.step()
            losses.append(loss.item())
        model.eval()
        with torch.no_grad():
            out = model(lr.unsqueeze(1).expand(-1, 3, -1, -1, -1))
        return np.array(losses), out.numpy(), sr_oracle.compute_psnr(out, hr)

    ref = RefSR(3, s, Fc, N, win)
    ref.load_state_dict(sd)
    rl, rout, rpsnr = drive(ref)
    ora = sr_oracle.OracleSR(3, s, Fc, N, win)
    ora.load_named(sd)
    ol, oout, opsnr = drive(ora)
    assert np.allclose(rl, ol, rtol=1e-5), (rl, ol)
    assert np.abs(rout - oout).max() < 1e-4
    np.savez_compressed(os.path.join(OUT, "traj_baseline.npz"),
                        cfg=np.array([Fc, N, win, s, B, H, W]), losses=rl,
                        eval_output=rout, psnr=np.array(rpsnr))
    print("trajectory losses", rl, "psnr", rpsnr)


def run_ewc(RefSR, RefEWC):
    """EWC on the bare SR net through the 4-D -> 5-D adapter of SURVEY.md 3.4:
    two tasks, loader batch 2, lambda 5000, online mode decay 0.999."""
    Fc, N, win, s, H, W = 32, 1, 1, 2, 8, 8

    class Adapter(torch.nn.Module):
        def __init__(self, net):
            super().__init__()
            self.net = net

        def forward(self, x):
            return self.net(x.unsqueeze(1).expand(-1, 3, -1, -1, -1))

    sd = synth.formula_state(3, s, Fc, N, win, gain=GAIN)
    tasks = []
    for k in range(2):
        lr = synth.formula_clip(4, 1, H, W, seed=31 + k)[:, 0]
        hr = synth.formula_target(4, H * s, W * s, seed=41 + k)
        tasks.append([(lr[0:2], hr[0:2]), (lr[2:4], hr[2:4])])

    ref = Adapter(RefSR(3, s, Fc, N, win))
    ref.net.load_state_dict(sd)
    ewc = RefEWC(ref, ewc_lambda=5000.0)
    ora = Adapter(sr_oracle.OracleSR(3, s, Fc, N, win))
    ora.net.load_named(sd)
    o_named = [("net." + n, p) for n, p in ora.net.named().items() if p.requires_grad]

    blob = {"cfg": np.array([Fc, N, win, s, 4, H, W])}
    o_fisher, o_opt = None, None
    for k, batches in enumerate(tasks):
        ewc.register_task(k, batches)
        f_new = sr_oracle.ewc_fisher(ora, batches, o_named)
        o_fisher = sr_oracle.ewc_online_merge(o_fisher, f_new)
        o_opt = {n: p.detach().clone() for n, p in o_named}
        for n, p in o_named:
            a, b = o_fisher[n].double(), ewc.fisher_dict[n].double()
            err = (a - b).abs().max().item() / max(b.abs().max().item(), 1e-30)
            assert err < 2e-5, (n, err)
        # perturb the weights deterministically (stands in for training on the next task)
        with torch.no_grad():
            for (n, p), (_, q) in zip(ref.named_parameters(), o_named):
                d = torch.from_numpy((synth.hash01(p.numel(), synth.name_seed(n) + k + 1)
                                      .reshape(p.shape) * 2 - 1).astype(np.float32)) * 0.01
                p.add_(d)
                q.add_(d)
        ref.zero_grad()
        pen = ewc.penalty()
        pen.backward()
        o_pen = sr_oracle.ewc_penalty(o_named, o_fisher, o_opt, 5000.0)
        assert abs(o_pen.item() - pen.item()) <= 1e-5 * abs(pen.item()), (o_pen.item(), pen.item())
        blob[f"penalty{k}"] = np.array(pen.item(), dtype=np.float64)
        for n, p in ref.named_parameters():
            key = n[len("net."):]
            blob[f"fisher{k}/{key}"] = grad_summary(ewc.fisher_dict[n])
            blob[f"pgrad{k}/{key}"] = grad_summary(p.grad)
    np.savez_compressed(os.path.join(OUT, "ewc_two_tasks.npz"), **blob)
    print("ewc penalties", blob["penalty0"], blob["penalty1"])


def run_default_init(RefSR):
    """Default (PyTorch) initialisation of the reference module under torch.manual_seed(0):
    per-tensor [sum, l2, first 4 values] so the drop-in's construction order can be checked."""
    blob = {}
    for tag, kw in (("cfg1", dict(scale_factor=2, num_features=32, num_residual_blocks=4, temporal_window=1)),
                    ("cfg2", dict(scale_factor=2, num_features=64, num_residual_blocks=8, temporal_window=1))):
        torch.manual_seed(0)
        ref = RefSR(**kw)
        for n, t in ref.state_dict().items():
            f = t.detach().double().reshape(-1)
            head = np.zeros(4)
            head[:min(4, f.numel())] = f[:4].numpy()
            blob[f"{tag}/{n}"] = np.concatenate([[f.sum().item(), f.norm().item()], head])
    np.savez_compressed(os.path.join(OUT, "default_init_seed0.npz"), **blob)
    print("default init:", len(blob), "tensors")


def run_default_init_fr():
    """Default initialisation of the reference's FrameRecoveryNet (base 16 and the default 64) and of a default
    EnhancementEngine() under torch.manual_seed(0): the construction-order pin of the drop-in's inpainting head."""
    from nerve_cl.models import EnhancementEngine as RefEngine
    from nerve_cl.models.frame_recovery import FrameRecoveryNet as RefFR
    blob = {}

    def add(tag, module):
        for n, t in module.state_dict().items():
            f = t.detach().double().reshape(-1)
            head = np.zeros(4)
            head[:min(4, f.numel())] = f[:4].numpy()
            blob[f"{tag}/{n}"] = np.concatenate([[f.sum().item(), f.norm().item()], head])

    torch.manual_seed(0)
    add("fr16", RefFR(3, 16, 2))
    torch.manual_seed(0)
    add("fr64", RefFR())
    torch.manual_seed(0)
    add("engine", RefEngine())
    np.savez_compressed(os.path.join(OUT, "fr_default_init_seed0.npz"), **blob)
    print("FR / engine default init:", len(blob), "tensors")


LIGHT_CASES = {"light_s2_train": (2, 2, 12, 20, True), "light_s3_eval": (3, 1, 9, 14, False)}


def run_light(name, cfg):
    """LightweightSuperResolution (reference super_resolution.py:434-470): output, loss, gradients, BN buffers."""
    from nerve_cl.models import LightweightSuperResolution as RefLight
    s, B, H, W, train = cfg
    sd = synth.formula_state_light(s, gain=GAIN)
    x = synth.formula_clip(B, 1, H, W)[:, 0].contiguous()
    tgt = synth.formula_target(B, H * s, W * s)
    ref = RefLight(s)
    ref.load_state_dict(sd, strict=True)
    ref.train(train)
    out = ref(x)
    loss = F.mse_loss(out, tgt)
    loss.backward()
    P = {k: v.clone().requires_grad_(v.is_floating_point() and "running" not in k) for k, v in sd.items()}
    o_out = sr_oracle.light_forward(P, x, train)
    F.mse_loss(o_out, tgt).backward()
    err = (o_out - out).abs().max().item()
    assert err < 2e-5, f"{name}: oracle != reference output {err:.3e}"
    ref_sd = ref.state_dict()
    for n, p in ref.named_parameters():
        e = (P[n].grad - p.grad).abs().max().item() / max(p.grad.abs().max().item(), 1e-30)
        assert e < 1e-4, f"{name}: oracle != reference grad {n} {e:.3e}"
    for n in sr_oracle.light_buffer_shapes():
        assert (P[n].double() - ref_sd[n].double()).abs().max().item() < 1e-5, n
    blob = {"cfg": np.array([s, B, H, W, int(train)], dtype=np.int64), "output": out.detach().numpy(),
            "loss": np.array(loss.item(), dtype=np.float64),
            "clamped_frac": np.array(((out == 0) | (out == 1)).float().mean().item())}
    for n, p in ref.named_parameters():
        blob["gsum/" + n] = grad_summary(p.grad)
        if p.grad.numel() <= 256:
            blob["gfull/" + n] = p.grad.detach().numpy()
    for n in sr_oracle.light_buffer_shapes():
        blob["buf/" + n] = ref_sd[n].detach().numpy()
    np.savez_compressed(os.path.join(OUT, f"{name}.npz"), **blob)
    print(f"{name}: loss {loss.item():.6f} clamped {blob['clamped_frac']:.3f} oracle-vs-ref out err {err:.2e}")


def run_fr(name, cfg):
    """FrameRecoveryNet (reference frame_recovery.py:335-446): output, loss, gradients, BatchNorm buffers - the pin of
    oracle/fr_oracle.py, groundwork for the HIP build of that network."""
    from nerve_cl.models.frame_recovery import FrameRecoveryNet as RefFR
    from oracle import fr_oracle
    base, B, T, H, W, train = cfg
    sd = synth.formula_state_fr(3, base, gain=GAIN)
    clip = synth.formula_clip(B, T + 1, H, W)
    corrupted, refs = clip[:, 0].contiguous(), clip[:, 1:].contiguous()
    mask = torch.zeros(B, 1, H, W)
    mask[:, :, H // 4:H // 4 + H // 2, W // 3:W // 3 + W // 2] = 1.0
    corrupted = corrupted * (1 - mask)                      # the corrupted region carries no signal
    tgt = synth.formula_target(B, H, W)
    ref = RefFR(3, base, T)
    ref.load_state_dict(sd, strict=True)
    ref.train(train)
    out = ref(corrupted, refs, mask)
    loss = F.mse_loss(out, tgt)
    loss.backward()
    P = {k: v.clone().requires_grad_(v.is_floating_point() and "running" not in k) for k, v in sd.items()}
    o_out = fr_oracle.frame_recovery_forward(P, corrupted, refs, mask, train)
    F.mse_loss(o_out, tgt).backward()
    err = (o_out - out).abs().max().item()
    assert err < 2e-5, f"{name}: oracle != reference output {err:.3e}"
    ref_sd = ref.state_dict()
    worst = 0.0
    for n, p in ref.named_parameters():
        e = (P[n].grad - p.grad).abs().max().item() / max(p.grad.abs().max().item(), 1e-30)
        worst = max(worst, e)
        assert e < 2e-4, f"{name}: oracle != reference grad {n} {e:.3e}"
    _, bufs = fr_oracle.shapes(3, base)
    for n in bufs:
        assert (P[n].double() - ref_sd[n].double()).abs().max().item() < 1e-5, n
    blob = {"cfg": np.array([base, B, T, H, W, int(train)], dtype=np.int64), "output": out.detach().numpy(),
            "loss": np.array(loss.item(), dtype=np.float64)}
    for n, p in ref.named_parameters():
        blob["gsum/" + n] = grad_summary(p.grad)
    for n in bufs:
        if "num_batches" not in n and ref_sd[n].numel() <= 64:
            blob["buf/" + n] = ref_sd[n].detach().numpy()
    np.savez_compressed(os.path.join(OUT, f"{name}.npz"), **blob)
    print(f"{name}: loss {loss.item():.6f} oracle-vs-ref out err {err:.2e}, worst grad err {worst:.2e}, "
          f"{sum(p.numel() for p in ref.parameters())} parameters")


from oracle import cl_cases  # noqa: E402
META = cl_cases.META


class Adapter4D(torch.nn.Module):
    """4-D frames -> the SR net's 5-D clip (T identical frames), tensor out: SURVEY.md 3.4's adapter around the reference's
    own `lr.unsqueeze(1).expand(-1, 3, ...)` (train_continual.py:51)."""

    def __init__(self, net):
        super().__init__()
        self.net = net

    def forward(self, x):
        return self.net(x.unsqueeze(1).expand(-1, 3, -1, -1, -1))


def delta_summaries(before, after):
    """per-parameter [sum, l2, 16 samples] of (after - before), float64"""
    return {n: grad_summary(after[n].detach().double() - before[n].detach().double()) for n in before}


def _close_summ(a, b, what, tol=2e-3):
    """two summary vectors agree relative to the l2 entry (parameter updates are lr * gradient: their fp32 error is an ulp of
    the PARAMETER, so compare at the update's own scale with a tolerance above that)"""
    scale = max(abs(b[1]), 1e-30)
    err = np.abs(a - b).max() / scale
    assert err <= tol, f"{what}: {err:.3e}"
    return err


def run_meta(RefSR):
    """reference FOMAML.adapt (maml.py:168-186 -> :74-110) and Reptile.train_step (:276-345) on the
    reference SR net; stored: per-parameter summaries of the parameter CHANGE, losses, BatchNorm buffers."""
    from nerve_cl.continual import FOMAML as RefFOMAML, Reptile as RefReptile
    from oracle import cl_oracle
    c = META
    sd = synth.formula_state(3, c["s"], c["F"], c["N"], c["win"], gain=GAIN)

    data = cl_cases.clip_pair

    def make_ref():
        m = RefSR(3, c["s"], c["F"], c["N"], c["win"])
        m.load_state_dict(sd)
        return m.train()

    def make_ora():
        m = sr_oracle.OracleSR(3, c["s"], c["F"], c["N"], c["win"])
        m.load_named(sd)
        return m.train()

    blob = {"cfg": np.array([c["F"], c["N"], c["win"], c["s"], 2, c["H"], c["W"]]),
            "fomaml_inner_lr": np.array(0.05), "fomaml_steps": np.array(3),
            "reptile_inner_lr": np.array(0.05), "reptile_outer_lr": np.array(0.5), "reptile_inner_steps": np.array(2)}
    # --- FOMAML.adapt, 3 inner steps
    ref = make_ref()
    before = {n: p.detach().clone() for n, p in ref.named_parameters()}
    adapted = RefFOMAML(ref, inner_lr=0.05, inner_steps=3).adapt(data(61), F.mse_loss)
    assert all(torch.equal(p, before[n]) for n, p in ref.named_parameters())      # the meta-model itself is untouched
    ora = make_ora()
    o_adapted = cl_oracle.fomaml_adapt(ora, data(61), F.mse_loss, 0.05, 3)
    ref_d = delta_summaries(before, dict(adapted.named_parameters()))
    ora_d = delta_summaries({n: ora.named()[n] for n in before}, {n: o_adapted.named()[n] for n in before})
    worst = max(_close_summ(ora_d[n], ref_d[n], "fomaml " + n) for n in ref_d)
    adapted.eval()
    with torch.no_grad():
        x, t = data(61)
        blob["fomaml_eval_loss"] = np.array(F.mse_loss(adapted(x), t).item(), dtype=np.float64)
    for n, v in ref_d.items():
        blob["fomaml_delta/" + n] = v
    for n in sr_oracle.buffer_shapes(c["F"]):
        blob["fomaml_buf/" + n] = adapted.state_dict()[n].numpy()
    # --- Reptile.train_step, 2 tasks x 2 inner steps
    ref = make_ref()
    before = {n: p.detach().clone() for n, p in ref.named_parameters()}
    tasks = [{"support": data(71)}, {"support": data(72)}]
    rl = RefReptile(ref, inner_lr=0.05, outer_lr=0.5, inner_steps=2).train_step(tasks, F.mse_loss)
    ora = make_ora()
    o_before = {n: ora.named()[n].detach().clone() for n in before}
    ol = cl_oracle.reptile_train_step(ora, tasks, F.mse_loss, 0.05, 0.5, 2)
    assert abs(rl - ol) <= 1e-5 * abs(rl), (rl, ol)
    ref_d = delta_summaries(before, dict(ref.named_parameters()))
    ora_d = delta_summaries(o_before, {n: ora.named()[n] for n in before})
    worst = max(worst, max(_close_summ(ora_d[n], ref_d[n], "reptile " + n) for n in ref_d))
    blob["reptile_loss"] = np.array(rl, dtype=np.float64)
    for n, v in ref_d.items():
        blob["reptile_delta/" + n] = v
    for n in sr_oracle.buffer_shapes(c["F"]):
        blob["reptile_buf/" + n] = ref.state_dict()[n].numpy()
    np.savez_compressed(os.path.join(OUT, "meta_f16.npz"), **blob)
    print(f"meta: fomaml eval loss {blob['fomaml_eval_loss']:.6f}, reptile loss {rl:.6f}, oracle-vs-ref worst {worst:.2e}")


def run_si(RefSR):
    """Reference SynapticIntelligence (ewc.py:306-379) around the reference SR net: 3 SGD steps with update_importance,
    register_task, then 2 steps of `mse + penalty` with update_importance after each (penalty stays 0: a reference quirk, see below) and 2
    without it (non-zero penalty)."""
    from nerve_cl.continual import SynapticIntelligence as RefSI
    from oracle import cl_oracle
    c = META
    sd = synth.formula_state(3, c["s"], c["F"], c["N"], c["win"], gain=GAIN)
    x, t = cl_cases.si_pair()

    drive = lambda model, si, named: cl_cases.si_drive(model, si, named, x, t)  # noqa: E731

    ref = RefSR(3, c["s"], c["F"], c["N"], c["win"])
    ref.load_state_dict(sd)
    names = [n for n, _ in ref.named_parameters()]
    rW, rO, rp, rl = drive(ref, RefSI(ref, si_lambda=2000.0, damping=0.1), names)
    ora = sr_oracle.OracleSR(3, c["s"], c["F"], c["N"], c["win"])
    ora.load_named(sd)
    osi = cl_oracle.SI(ora, si_lambda=2000.0, damping=0.1)
    key = {n: n.replace(".", "|") for n in names}
    oW, oO, op, ol = drive(ora, osi, [key[n] for n in names])
    assert np.allclose(rl, ol, rtol=1e-5), (rl, ol)
    assert np.allclose(rp, op, rtol=2e-3), (rp, op)
    worst = 0.0
    for n in names:
        worst = max(worst, _close_summ(grad_summary(oW[key[n]]), grad_summary(rW[n]), "si W " + n, 5e-3))
        worst = max(worst, _close_summ(grad_summary(oO[key[n]]), grad_summary(rO[n]), "si omega " + n, 5e-3))
    blob = {"cfg": np.array([c["F"], c["N"], c["win"], c["s"], 2, c["H"], c["W"]]), "si_lambda": np.array(2000.0),
            "damping": np.array(0.1), "lr": np.array(0.05), "penalties": np.array(rp), "losses": np.array(rl)}
    for n in names:
        blob["W/" + n] = grad_summary(rW[n])
        blob["omega/" + n] = grad_summary(rO[n])
    np.savez_compressed(os.path.join(OUT, "si_f16.npz"), **blob)
    print("si: penalties", rp, "losses", rl, f"oracle-vs-ref worst {worst:.2e}")


def run_cfg5_loop(RefSR, RefEWC):
    """the loop of experiments/train_continual.py:26-69 (`train_with_ewc`) driven with the reference EWC and
    the reference SR net through the 4-D -> 5-D adapter: 2 tasks (create_task_data-style offsets +0.2 / -0.2 on formula
    data) x 1 epoch of 3 batches of 2, Adam(1e-4), loss = mse + penalty, lambda 5000, register_task after each task on the
    same batches.  Stored: per-step task_loss / ewc_loss, final Fisher summaries."""
    from oracle import cl_oracle
    c = META
    sd = synth.formula_state(3, c["s"], c["F"], c["N"], c["win"], gain=GAIN)
    tasks = cl_cases.cfg5_tasks()

    ref = Adapter4D(RefSR(3, c["s"], c["F"], c["N"], c["win"]))
    ref.net.load_state_dict(sd)
    ewc = RefEWC(ref, ewc_lambda=5000.0)
    opt = torch.optim.Adam(ref.parameters(), lr=1e-4)
    crit = torch.nn.MSELoss()
    tl, el = [], []
    for task_id, (_, batches) in enumerate(tasks):
        ref.train()
        for lr_b, hr_b in batches:
            opt.zero_grad()
            task_loss = crit(ref(lr_b), hr_b)
            ewc_loss = ewc.penalty()
            (task_loss + ewc_loss).backward()
            opt.step()
            tl.append(task_loss.item())
            el.append(float(ewc_loss.item()) if torch.is_tensor(ewc_loss) else float(ewc_loss))
        ewc.register_task(task_id, batches)

    ora = Adapter4D(sr_oracle.OracleSR(3, c["s"], c["F"], c["N"], c["win"]))
    ora.net.load_named(sd)
    log = cl_oracle.train_with_ewc(ora, tasks, 5000.0, 1e-4, 1)
    assert np.allclose(tl, log["task_loss"], rtol=1e-5), (tl, log["task_loss"])
    assert np.allclose(el, log["ewc_loss"], rtol=2e-3, atol=1e-12), (el, log["ewc_loss"])
    blob = {"cfg": np.array([c["F"], c["N"], c["win"], c["s"], 2, c["H"], c["W"]]), "task_loss": np.array(tl),
            "ewc_loss": np.array(el), "offsets": np.array([0.2, -0.2]), "lr": np.array(1e-4), "ewc_lambda": np.array(5000.0)}
    for n, f in ewc.fisher_dict.items():
        blob["fisher/" + n[len("net."):]] = grad_summary(f)
    np.savez_compressed(os.path.join(OUT, "cfg5_loop.npz"), **blob)
    print("cfg5 loop: task_loss", tl, "ewc_loss", el)


FR_CASES = {"fr_b16_train": (16, 2, 2, 32, 48, True), "fr_b16_eval": (16, 1, 2, 40, 40, False)}


def main():
    torch.manual_seed(0)
    torch.set_num_threads(8)
    os.makedirs(OUT, exist_ok=True)
    RefSR, RefEWC = import_reference()
    if "--only-fr-init" in sys.argv:
        run_default_init_fr()
        return
    if "--only-cl" in sys.argv:
        run_meta(RefSR)
        run_si(RefSR)
        run_cfg5_loop(RefSR, RefEWC)
        return
    if "--only-fr" in sys.argv:
        for name, cfg in FR_CASES.items():
            run_fr(name, cfg)
        return
    for name, cfg in LIGHT_CASES.items():
        run_light(name, cfg)
    if "--only-light" in sys.argv:
        return
    for name, cfg in CASES.items():
        run_case(name, cfg, RefSR)
    run_trajectory(RefSR)
    run_ewc(RefSR, RefEWC)
    run_meta(RefSR)
    run_si(RefSR)
    run_cfg5_loop(RefSR, RefEWC)
    run_default_init(RefSR)
    for name, cfg in FR_CASES.items():
        run_fr(name, cfg)
    run_default_init_fr()
    total = sum(os.path.getsize(os.path.join(OUT, f)) for f in os.listdir(OUT))
    print(f"wrote {OUT}: {total/1024:.0f} KiB")


if __name__ == "__main__":
    main()
