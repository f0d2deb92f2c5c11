"""GPU: nerve_cl.models.SuperResolutionNet (HIP kernels through the C ABI) against
 (1) the golden fixtures captured from the reference (tests/golden), and
 (2) the CPU oracle evaluated in the same process on the same seeded inputs.
Tolerance: north_star asks for 1e-3 relative in fp32; every check below is at 1e-3 of the
reference tensor's max magnitude (gradients: of that tensor's own max), most pass at 1e-5."""
import copy
import glob
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import sr_oracle, synth
from oracle.make_goldens import grad_summary

pytestmark = pytest.mark.gpu

GOLD = os.path.join(os.path.dirname(__file__), "golden")
CASES = sorted(glob.glob(os.path.join(GOLD, "sr_*.npz")))
REL = 1e-3


def rel(a, b):
    a = torch.as_tensor(np.asarray(a)).double() if not torch.is_tensor(a) else a.detach().double().cpu()
    b = torch.as_tensor(np.asarray(b)).double() if not torch.is_tensor(b) else b.detach().double().cpu()
    return ((a - b).abs().max() / b.abs().max().clamp_min(1e-30)).item()


@pytest.fixture(scope="module")
def SR():
    if not torch.cuda.is_available():
        pytest.fail("GPU tests need a HIP device")
    from nerve_cl import _nvq
    from nerve_cl.models import SuperResolutionNet
    _nvq.lib()
    return SuperResolutionNet


def build_pair(SR, Fc, N, win, s, train, gain=synth.GOLDEN_GAIN):
    sd = synth.formula_state(3, s, Fc, N, win, gain=gain)
    net = SR(3, s, Fc, N, win)
    net.load_state_dict(sd, strict=True)
    net = net.cuda().train(train)
    ora = sr_oracle.OracleSR(3, s, Fc, N, win)
    ora.load_named(sd)
    ora.train(train)
    return net, ora


@pytest.mark.parametrize("path", CASES, ids=[os.path.basename(p)[3:-4] for p in CASES])
def test_against_reference_fixture_and_oracle(SR, path):
    g = np.load(path)
    Fc, N, win, s, B, H, W, train = [int(v) for v in g["cfg"]]
    T = 2 * win + 1
    net, ora = build_pair(SR, Fc, N, win, s, bool(train))
    x = synth.formula_clip(B, T, H, W)
    tgt = synth.formula_target(B, H * s, W * s)

    out, inter = net(x.cuda(), return_intermediate=True)
    loss = F.mse_loss(out, tgt.cuda())
    loss.backward()
    o_out, o_inter = ora(x, return_intermediate=True)
    F.mse_loss(o_out, tgt).backward()

    # --- forward: fixtures from the reference
    assert out.shape == (B, 3, H * s, W * s)
    assert rel(out, g["output"]) < REL
    assert abs(loss.item() - float(g["loss"])) < REL * float(g["loss"])
    assert rel(inter["features"][0], g["feat0"]) < REL
    assert rel(inter["aligned"][0], g["aligned0"]) < REL
    assert rel(inter["aggregated"], g["aggregated"]) < REL
    # --- forward: oracle, every intermediate
    for t in range(T):
        assert rel(inter["features"][t], o_inter["features"][t]) < REL, t
        assert rel(inter["aligned"][t], o_inter["aligned"][t]) < REL, t
    assert rel(inter["aggregated"], o_inter["aggregated"]) < REL
    assert rel(out, o_out) < REL

    # --- gradients: all tensors vs the oracle, summaries vs the reference fixture
    named = dict(net.named_parameters())
    onamed = ora.named()
    worst = 0.0
    for n, p in named.items():
        assert p.grad is not None, n
        e = rel(p.grad, onamed[n].grad)
        worst = max(worst, e)
        assert e < REL, (n, e)
        ref = g["gsum/" + n]
        got = grad_summary(p.grad.cpu())
        assert abs(got[1] - ref[1]) <= REL * max(ref[1], 1e-12), n
        assert np.abs(got[2:] - ref[2:]).max() <= REL * max(np.abs(ref[2:]).max(), ref[1] * 1e-2), n
        if "gfull/" + n in g.files:
            assert rel(p.grad, g["gfull/" + n]) < REL, n
    # --- BatchNorm buffers after the step
    sd = net.state_dict()
    for key in g.files:
        if key.startswith("buf/"):
            assert rel(sd[key[4:]].double(), g[key]) < REL, key
    print(f"{os.path.basename(path)}: worst grad rel err vs oracle {worst:.2e}")


def test_baseline_training_trajectory(SR):
    """train_baseline-style loop (AdamW, MSE, frame expanded to T=3): losses of 3 steps and the
    eval-mode output afterwards, against values captured from the reference."""
    g = np.load(os.path.join(GOLD, "traj_baseline.npz"))
    Fc, N, win, s, B, H, W = [int(v) for v in g["cfg"]]
    net, _ = build_pair(SR, Fc, N, win, s, True)
    lr = synth.formula_clip(B, 1, H, W, seed=5)[:, 0].cuda()
    hr = synth.formula_target(B, H * s, W * s, seed=7).cuda()
    opt = torch.optim.AdamW(net.parameters(), lr=1e-3, weight_decay=1e-5)
    losses = []
    for _ in range(3):
        opt.zero_grad()
        out = net(lr.unsqueeze(1).expand(-1, 3, -1, -1, -1))     # stride-0 expanded input
        loss = F.mse_loss(out, hr)
        loss.backward()
        opt.step()
        losses.append(loss.item())
    assert np.allclose(losses, g["losses"], rtol=REL), (losses, g["losses"])
    net.eval()
    with torch.no_grad():
        out = net(lr.unsqueeze(1).expand(-1, 3, -1, -1, -1))
    assert np.abs(out.cpu().numpy() - g["eval_output"]).max() < 2e-3
    assert abs(sr_oracle.compute_psnr(out.cpu(), hr.cpu()) - float(g["psnr"])) < 0.02


def test_cfg1_shape_against_oracle(SR):
    """BASELINE configs[0]: F=32, N=4, T=3, s=2 on 64x64 clips (B reduced to 2 for CPU time).

    Every parameter gradient must be within 1e-3 (relative to that tensor's max) of the fp32
    oracle, with one documented exception: ReLU is discontinuous, and among the ~1e6 flow-net
    activations of this case one pre-activation has |value| ~1.5e-7, i.e. fp32 rounding order
    decides its sign (measured with tools/debug_grad_stages.py: exactly 1 of 1 048 576 mask
    elements differs per frame; every stage up to that mask agrees to 3e-6).  One flipped mask
    element moves flow_net.0's tiny gradient (|g| ~1e-7) by ~1e-3 of its max.  So: at most two
    tensors may exceed 1e-3, none may exceed 1e-2, and the whole gradient vector must agree to
    1e-3 in relative L2."""
    net, ora = build_pair(SR, 32, 4, 1, 2, True)
    x = synth.formula_clip(2, 3, 64, 64, seed=3)
    tgt = synth.formula_target(2, 128, 128, seed=4)
    out = net(x.cuda())
    F.mse_loss(out, tgt.cuda()).backward()
    o_out = ora(x)
    F.mse_loss(o_out, tgt).backward()
    assert rel(out, o_out) < REL
    onamed = ora.named()
    over, num, den = [], 0.0, 0.0
    for n, p in net.named_parameters():
        e = rel(p.grad, onamed[n].grad)
        d = p.grad.detach().double().cpu() - onamed[n].grad.double()
        num += float((d * d).sum())
        den += float((onamed[n].grad.double() ** 2).sum())
        if e >= REL:
            over.append((n, e))
        assert e < 1e-2, (n, e)
    print("  tensors over 1e-3:", over, " global rel L2:", (num / den) ** 0.5)
    assert len(over) <= 2, over
    assert (num / den) ** 0.5 < REL
    for n in sr_oracle.buffer_shapes(32):
        assert rel(net.state_dict()[n].double(), onamed[n].double()) < REL, n


def test_default_config_shapes_like_reference_tests(SR):
    """reference tests/test_models.py:61-73: default net on (2,3,3,64,64); scale factors 2,3,4."""
    torch.manual_seed(0)
    net = SR().cuda()
    with torch.no_grad():
        out = net(torch.randn(2, 3, 3, 64, 64, device="cuda"))
    assert out.shape == (2, 3, 128, 128)
    assert out.min().item() >= 0 and out.max().item() <= 1
    for s in (2, 3, 4):
        n2 = SR(scale_factor=s, num_features=16, num_residual_blocks=1).cuda()
        with torch.no_grad():
            o = n2(torch.randn(1, 3, 3, 32, 32, device="cuda"))
        assert o.shape == (1, 3, 32 * s, 32 * s)
    o1 = net.forward_single(torch.rand(1, 3, 16, 24, device="cuda"))
    assert o1.shape == (1, 3, 32, 48)


def test_modes_and_repeatability(SR):
    net, _ = build_pair(SR, 32, 2, 1, 2, False)
    x = synth.formula_clip(1, 3, 40, 70, seed=9).cuda()
    with torch.no_grad():
        a = net(x)
        b = net(x)
    assert torch.equal(a, b)                      # forward is deterministic
    c = net(x)                                     # eval mode with grad enabled
    assert torch.equal(a, c.detach())
    c.sum().backward()
    assert all(p.grad is not None for p in net.parameters())
    nbt = net.state_dict()["feature_extractor.body.0.bn.num_batches_tracked"].item()
    net.train()
    net(x)
    assert net.state_dict()["feature_extractor.body.0.bn.num_batches_tracked"].item() == nbt + 3
    clone = copy.deepcopy(net).eval()
    net.eval()
    with torch.no_grad():
        assert torch.equal(clone(x), net(x))


def test_second_backward_accumulates(SR):
    net, _ = build_pair(SR, 16, 1, 1, 2, True)
    x = synth.formula_clip(1, 3, 12, 16).cuda()
    net(x).sum().backward()
    g1 = {n: p.grad.clone() for n, p in net.named_parameters()}
    net.eval()                                     # same statistics both times
    net.zero_grad()
    net(x).sum().backward()
    ga = {n: p.grad.clone() for n, p in net.named_parameters()}
    net(x).sum().backward()
    for n, p in net.named_parameters():
        assert rel(p.grad, 2 * ga[n]) < 1e-5, n
    assert g1.keys() == ga.keys()


def test_bf16_math_mode_quality(SR):
    """NVQ_MATH_BF16 (bf16 MFMA operands, fp32 accumulate/storage) has no counterpart in the reference, so it
    is judged against the fp32 oracle: PSNR of the output, relative loss difference, and direction of the
    large gradients (SURVEY.md 7 hard part (v))."""
    from nerve_cl import _nvq
    net, ora = build_pair(SR, 32, 4, 1, 2, True)
    net.math_mode = _nvq.MATH_BF16
    x = synth.formula_clip(2, 3, 64, 64, seed=3)
    tgt = synth.formula_target(2, 128, 128, seed=4)
    out = net(x.cuda())
    loss = F.mse_loss(out, tgt.cuda())
    loss.backward()
    o_out = ora(x)
    o_loss = F.mse_loss(o_out, tgt)
    o_loss.backward()
    psnr = sr_oracle.compute_psnr(out.detach().cpu(), o_out.detach())
    onamed = ora.named()
    cos_min, worst = 1.0, None
    for n, p in net.named_parameters():
        a, b = p.grad.detach().double().cpu().reshape(-1), onamed[n].grad.double().reshape(-1)
        cos = float((a @ b) / (a.norm() * b.norm()).clamp_min(1e-300))
        if "motion_estimator" not in n and cos < cos_min:
            cos_min, worst = cos, n
    print(f"  bf16 math: PSNR vs fp32 oracle {psnr:.1f} dB, loss {loss.item():.6f} vs {o_loss.item():.6f}, "
          f"min grad cosine (non-flow tensors) {cos_min:.5f} at {worst}")
    assert psnr > 45.0
    assert abs(loss.item() - o_loss.item()) < 2e-3 * o_loss.item()
    assert cos_min > 0.99


def test_bf16_activation_storage_quality(SR):
    """bf16 MFMA operands AND bf16 storage of the conv-internal tensors (dense-block buffers, flow-net and
    attention hidden activations and their gradients): judged against the fp32 oracle like the test above."""
    from nerve_cl import _nvq
    net, ora = build_pair(SR, 32, 4, 1, 2, True)
    net.math_mode, net.bf16_activations = _nvq.MATH_BF16, True
    x = synth.formula_clip(2, 3, 64, 64, seed=3)
    tgt = synth.formula_target(2, 128, 128, seed=4)
    out, inter = net(x.cuda(), return_intermediate=True)
    loss = F.mse_loss(out, tgt.cuda())
    loss.backward()
    o_out = ora(x)
    o_loss = F.mse_loss(o_out, tgt)
    o_loss.backward()
    psnr = sr_oracle.compute_psnr(out.detach().cpu(), o_out.detach())
    onamed = ora.named()
    cos_min, worst = 1.0, None
    for n, p in net.named_parameters():
        a, b = p.grad.detach().double().cpu().reshape(-1), onamed[n].grad.double().reshape(-1)
        cos = float((a @ b) / (a.norm() * b.norm()).clamp_min(1e-300))
        if "motion_estimator" not in n and cos < cos_min:
            cos_min, worst = cos, n
    print(f"  bf16 storage: PSNR vs fp32 oracle {psnr:.1f} dB, loss {loss.item():.6f} vs {o_loss.item():.6f}, "
          f"min grad cosine (non-flow tensors) {cos_min:.5f} at {worst}")
    assert inter["aggregated"].dtype == torch.float32
    assert psnr > 40.0
    assert abs(loss.item() - o_loss.item()) < 5e-3 * o_loss.item()
    assert cos_min > 0.98
    # a few optimiser steps keep tracking the fp32 trajectory
    net2, ora2 = build_pair(SR, 32, 4, 1, 2, True)
    net2.math_mode, net2.bf16_activations = _nvq.MATH_BF16, True
    o1 = torch.optim.AdamW(net2.parameters(), lr=1e-3, weight_decay=1e-5)
    o2 = torch.optim.AdamW(ora2.parameters(), lr=1e-3, weight_decay=1e-5)
    l1, l2 = [], []
    for _ in range(3):
        o1.zero_grad(); a = F.mse_loss(net2(x.cuda()), tgt.cuda()); a.backward(); o1.step(); l1.append(a.item())
        o2.zero_grad(); b = F.mse_loss(ora2(x), tgt); b.backward(); o2.step(); l2.append(b.item())
    print("  bf16 storage losses", l1, "fp32 oracle", l2)
    assert np.allclose(l1, l2, rtol=1e-2)


def test_bf16_feature_storage_switch(SR, monkeypatch):
    """NVQ_BF16_FEATURES=0 keeps the frames' feature tensors (and the gradient w.r.t. them) in fp32 inside the bf16 mode:
    both settings run, and differ only by the bf16 rounding of those tensors."""
    from nerve_cl import _nvq
    x = synth.formula_clip(2, 3, 32, 40, seed=3).cuda()
    res = []
    for flag in ("1", "0"):
        monkeypatch.setenv("NVQ_BF16_FEATURES", flag)
        net, _ = build_pair(SR, 32, 2, 1, 2, True)
        net.math_mode, net.bf16_activations = _nvq.MATH_BF16, True
        out, inter = net(x, return_intermediate=True)
        out.square().mean().backward()
        res.append((out.detach(), torch.cat([p.grad.flatten() for p in net.parameters()]), inter))
    assert not torch.equal(res[0][0], res[1][0])
    mse = (res[0][0] - res[1][0]).pow(2).mean().item()
    assert 10 * np.log10(1.0 / max(mse, 1e-12)) > 50.0
    assert F.cosine_similarity(res[0][1], res[1][1], dim=0).item() > 0.999
    a0, a1 = res[0][2]["aligned"][1], res[1][2]["aligned"][1]          # centre frame: the same values, rounded once to bf16 in run 0
    assert a0.dtype == a1.dtype == torch.float32 and torch.equal(a0, a1.bfloat16().float())


LIGHT = sorted(glob.glob(os.path.join(GOLD, "light_*.npz")))


@pytest.mark.parametrize("path", LIGHT, ids=[os.path.basename(p)[:-4] for p in LIGHT])
def test_lightweight_against_reference_fixture_and_oracle(SR, path):
    """LightweightSuperResolution (reference super_resolution.py:434-470) through the HIP kernels: output, loss,
    every gradient and the BN running statistics against the reference fixture and the oracle."""
    from nerve_cl.models import LightweightSuperResolution
    g = np.load(path)
    s, B, H, W, train = [int(v) for v in g["cfg"]]
    sd = synth.formula_state_light(s, gain=synth.GOLDEN_GAIN)
    net = LightweightSuperResolution(s)
    assert set(net.state_dict().keys()) == set(sd.keys())
    net.load_state_dict(sd, strict=True)
    net = net.cuda().train(bool(train))
    x = synth.formula_clip(B, 1, H, W)[:, 0].contiguous()
    tgt = synth.formula_target(B, H * s, W * s)
    out = net(x.cuda())
    loss = F.mse_loss(out, tgt.cuda())
    loss.backward()
    assert rel(out, g["output"]) < REL
    assert abs(loss.item() - float(g["loss"])) < 1e-5
    P = {k: v.clone().requires_grad_(v.is_floating_point() and "running" not in k) for k, v in sd.items()}
    F.mse_loss(sr_oracle.light_forward(P, x, bool(train)), tgt).backward()
    worst = 0.0
    for n, p in net.named_parameters():
        e = rel(p.grad, P[n].grad)
        worst = max(worst, e)
        assert e < REL, (n, e)
        if "gfull/" + n in g.files:
            assert rel(p.grad, g["gfull/" + n]) < REL, n
        ref = g["gsum/" + n]
        got = grad_summary(p.grad.cpu())
        assert abs(got[1] - ref[1]) <= REL * max(ref[1], 1e-12), n
    for n, b in net.named_buffers():
        assert rel(b.double(), g["buf/" + n]) < 1e-5, n
    print(f"light {os.path.basename(path)}: worst grad rel {worst:.2e}")
    # eval-mode inference through EnhancementEngine(use_lightweight_sr)
    from nerve_cl.models import EnhancementConfig, EnhancementEngine
    eng = EnhancementEngine(EnhancementConfig(frame_recovery_enabled=False, use_lightweight_sr=True, scale_factor=s))
    eng.super_resolution.load_state_dict(net.state_dict())
    eng = eng.cuda().eval()
    with torch.no_grad():
        y = eng(x.cuda().unsqueeze(1).expand(-1, 3, -1, -1, -1))["enhanced"]
        assert torch.equal(y, net.eval()(x.cuda()))


def test_lightweight_bf16_mode_runs_close_to_fp32(SR):
    """LightweightSuperResolution in the bf16 throughput mode (F = 32: the 32-channel kernel variants): forward + backward
    run and stay close to the fp32 mode."""
    from nerve_cl import _nvq
    from nerve_cl.models import LightweightSuperResolution
    sd = synth.formula_state_light(2, gain=synth.GOLDEN_GAIN)
    x = synth.formula_clip(2, 1, 40, 56)[:, 0].contiguous().cuda()
    tgt = synth.formula_target(2, 80, 112).cuda()
    outs, grads = [], []
    for bf16 in (False, True):
        net = LightweightSuperResolution(2)
        net.load_state_dict(sd, strict=True)
        net = net.cuda().train()
        if bf16:
            net.math_mode, net.bf16_activations = _nvq.MATH_BF16, True
        out = net(x)
        F.mse_loss(out, tgt).backward()
        outs.append(out.detach())
        grads.append(torch.cat([p.grad.flatten() for p in net.parameters()]))
    mse = (outs[0] - outs[1]).pow(2).mean().item()
    assert 10 * np.log10(1.0 / max(mse, 1e-12)) > 40.0
    cos = F.cosine_similarity(grads[0], grads[1], dim=0).item()
    assert cos > 0.99, cos

