"""GPU: the mode bench.py times - bf16 MFMA operands + bf16 storage of the conv-internal tensors at F = 64 (which alone
reaches rdb_tail_kernel, the 64-channel depthwise / correlation / weight-gradient variants and the slice-planar CatBuf) -
end to end against (a) the reference fixture sr_d_f64n1_t3_s2_train and (b) the fp32 CPU oracle at the benchmark's
F=64 / 8 blocks / T=3 on 64x96 clips, under both buffer layouts (NVQ_PLANAR) and both feature storages
(NVQ_BF16_FEATURES).  The reference has no reduced-precision path, so the bf16 mode is judged as SURVEY.md 8(d) says:
PSNR of the output (formula of experiments/train_baseline.py:27-32), relative loss, direction of every non-flow gradient,
and a 3-step AdamW trajectory.  Also nerve_cl.ops.mse_loss against torch."""
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import sr_oracle, synth

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(__file__), "golden")
MODES = [("1", "1"), ("0", "1"), ("1", "0"), ("0", "0")]        # (NVQ_PLANAR, NVQ_BF16_FEATURES)


def _pair(Fc, N, win, s, gain=synth.GOLDEN_GAIN):
    from nerve_cl import _nvq
    from nerve_cl.models import SuperResolutionNet
    sd = synth.formula_state(3, s, Fc, N, win, gain=gain)
    net = SuperResolutionNet(3, s, Fc, N, win)
    net.load_state_dict(sd, strict=True)
    net = net.cuda().train()
    net.math_mode, net.bf16_activations = _nvq.MATH_BF16, True
    ora = sr_oracle.OracleSR(3, s, Fc, N, win)
    ora.load_named(sd)
    ora.train()
    return net, ora


def _grad_cosines(net, ora):
    onamed = ora.named()
    cos_min, worst, flow_min = 1.0, None, 1.0
    for n, p in net.named_parameters():
        a, b = p.grad.detach().double().cpu().reshape(-1), onamed[n].grad.double().reshape(-1)
        cos = float((a @ b) / (a.norm() * b.norm()).clamp_min(1e-300))
        if "motion_estimator" in n:
            flow_min = min(flow_min, cos)
        elif cos < cos_min:
            cos_min, worst = cos, n
    return cos_min, worst, flow_min


@pytest.mark.parametrize("planar,bf16feat", MODES)
def test_f64_fixture_in_the_benchmarked_mode(monkeypatch, planar, bf16feat):
    monkeypatch.setenv("NVQ_PLANAR", planar)
    monkeypatch.setenv("NVQ_BF16_FEATURES", bf16feat)
    g = np.load(os.path.join(GOLD, "sr_d_f64n1_t3_s2_train.npz"))
    Fc, N, win, s, B, H, W, train = [int(v) for v in g["cfg"]]
    assert Fc == 64 and train == 1
    net, ora = _pair(Fc, N, win, s)
    x = synth.formula_clip(B, 2 * win + 1, H, W)
    tgt = synth.formula_target(B, H * s, W * s)
    out = net(x.cuda())
    loss = F.mse_loss(out, tgt.cuda())
    loss.backward()
    o_loss = F.mse_loss(ora(x), tgt)
    o_loss.backward()
    ref_out = torch.from_numpy(g["output"])                      # the REFERENCE's output for these weights / inputs
    psnr = sr_oracle.compute_psnr(out.detach().cpu(), ref_out)
    cos_min, worst, flow_min = _grad_cosines(net, ora)
    print(f"  f64 fixture planar={planar} bf16feat={bf16feat}: PSNR vs reference {psnr:.1f} dB, loss {loss.item():.6f} vs "
          f"{float(g['loss']):.6f}, min grad cosine {cos_min:.5f} at {worst} (flow net {flow_min:.4f})")
    assert psnr > 43.0                                        # measured 46.6 dB
    assert abs(loss.item() - float(g["loss"])) < 2e-3 * float(g["loss"])     # measured 2.3e-4
    assert cos_min > 0.975                                    # measured 0.983 (attention.0, a 1e-4-sized gradient)


@pytest.mark.parametrize("planar,bf16feat", MODES[:2] + MODES[3:])
def test_bench_config_f64_n8_against_fp32_oracle(monkeypatch, planar, bf16feat):
    """bench.py's network (F=64, 8 dense blocks, T=3, s=2) on clips small enough for the CPU oracle."""
    monkeypatch.setenv("NVQ_PLANAR", planar)
    monkeypatch.setenv("NVQ_BF16_FEATURES", bf16feat)
    net, ora = _pair(64, 8, 1, 2)
    x = synth.formula_clip(2, 3, 64, 96, seed=5)
    tgt = synth.formula_target(2, 128, 192, seed=6)
    out = net(x.cuda())
    loss = F.mse_loss(out, tgt.cuda())
    loss.backward()
    o_out = ora(x)
    o_loss = F.mse_loss(o_out, tgt)
    o_loss.backward()
    psnr = sr_oracle.compute_psnr(out.detach().cpu(), o_out.detach())
    cos_min, worst, flow_min = _grad_cosines(net, ora)
    print(f"  F=64 N=8 planar={planar} bf16feat={bf16feat}: PSNR vs fp32 oracle {psnr:.1f} dB, loss {loss.item():.6f} vs "
          f"{o_loss.item():.6f}, min grad cosine {cos_min:.5f} at {worst} (flow net {flow_min:.4f})")
    assert psnr > 41.0                                        # measured 44.5 dB
    assert abs(loss.item() - o_loss.item()) < 2e-3 * o_loss.item()     # measured 3.5e-4
    assert cos_min > 0.985                                    # measured 0.995
    if (planar, bf16feat) != ("1", "1"):
        return
    # three AdamW steps from the same start: the bf16 trajectory tracks the fp32 one
    net2, ora2 = _pair(64, 8, 1, 2)
    o1 = torch.optim.AdamW(net2.parameters(), lr=1e-3, weight_decay=1e-5)
    o2 = torch.optim.AdamW(ora2.parameters(), lr=1e-3, weight_decay=1e-5)
    l1, l2 = [], []
    for _ in range(3):
        o1.zero_grad(); a = F.mse_loss(net2(x.cuda()), tgt.cuda()); a.backward(); o1.step(); l1.append(a.item())
        o2.zero_grad(); b = F.mse_loss(ora2(x), tgt); b.backward(); o2.step(); l2.append(b.item())
    print("  F=64 N=8 bf16 losses", l1, "fp32 oracle", l2)
    assert np.allclose(l1, l2, rtol=1e-2)


def test_mse_loss_kernels_against_torch():
    from nerve_cl import ops
    torch.manual_seed(0)
    for shape in ((2, 3, 36, 52), (1, 3, 7, 5), (3, 1, 1, 1)):
        a = torch.rand(shape, device="cuda", requires_grad=True)
        b = torch.rand(shape, device="cuda")
        loss = ops.mse_loss(a, b)
        (3.0 * loss).backward()
        a2 = a.detach().clone().requires_grad_(True)
        ref = F.mse_loss(a2.double(), b.double())
        (3.0 * ref).backward()
        assert abs(loss.item() - ref.item()) <= 1e-6 * ref.item()
        assert (a.grad - a2.grad).abs().max() <= 1e-6 * a2.grad.abs().max()
    crit = ops.MSELoss()
    assert crit(b, b).item() == 0.0
    with pytest.raises(RuntimeError):
        ops.mse_loss(torch.zeros(2), torch.zeros(2))            # CPU tensors: no fallback
