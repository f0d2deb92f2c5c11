"""GPU: VALUE parity of the SR hot path at real sizes (reference nerve_cl/models/super_resolution.py:327-391), where
tests/test_full_size_gpu.py only checks size-independent properties:

 (a) F=64, 8 blocks, T=3 on a 135x240 clip (1/16 of cfg2's pixels: 17 x 8 tiles of the conv kernels, image edges that are not
     tile edges), one TRAINING step: output, loss, every parameter gradient and the BatchNorm buffers against the fp32 CPU
     oracle - exact-fp32 mode to 1e-3, the benchmarked bf16 mode by PSNR / gradient direction / loss;
 (b) ONE full 540x960 clip, eval-mode forward, both modes, against the oracle's forward (~10 s of CPU);
 (c) the benchmarked mode at bench.py's batch: 8 clips of 540x960 (8160 tiles, 32-bit offsets up to 1.06e9 elements): clip 7's
     output inside the batch equals its output alone bit for bit, and the parameter gradients of a loss on clip 7 alone,
     computed through the 8-clip batch, equal those of the single-clip run (other clips contribute exact zeros; only the
     fp32 summation order of the weight-gradient splits differs).
Inputs and weights are the closed-form ones of oracle/synth.py."""
import time

import pytest
import torch
import torch.nn.functional as F

from oracle import sr_oracle, synth

pytestmark = pytest.mark.gpu


def _pair(bf16, train):
    from nerve_cl import _nvq
    from nerve_cl.models import SuperResolutionNet
    sd = synth.formula_state(3, 2, 64, 8, 1, gain=synth.GOLDEN_GAIN)
    net = SuperResolutionNet(3, 2, 64, 8, 1)
    net.load_state_dict(sd, strict=True)
    net = net.cuda().train(train)
    net.math_mode, net.bf16_activations = (_nvq.MATH_BF16, True) if bf16 else (_nvq.MATH_F32, False)
    ora = sr_oracle.OracleSR(3, 2, 64, 8, 1)
    ora.load_named(sd)
    ora.train(train)
    return net, ora


@pytest.fixture(scope="module")
def oracle_135():
    """the oracle's training step on the 135x240 clip, computed once for both modes"""
    _, ora = _pair(False, True)
    x = synth.formula_clip(1, 3, 135, 240, seed=13)
    tgt = synth.formula_target(1, 270, 480, seed=14)
    t0 = time.perf_counter()
    out = ora(x)
    loss = F.mse_loss(out, tgt)
    loss.backward()
    print(f"  oracle 135x240 train step: {time.perf_counter() - t0:.1f} s")
    return x, tgt, out.detach(), loss.item(), ora


@pytest.fixture(scope="module")
def oracle_135_f64(oracle_135):
    """the same step with the oracle evaluated in float64: the yardstick that says how far an fp32 evaluation of this network
    may sit from the exact gradient (conditioning), as opposed to how far two fp32 evaluations sit from each other"""
    x, tgt = oracle_135[0], oracle_135[1]
    _, ora = _pair(False, True)
    ora = ora.double()
    t0 = time.perf_counter()
    F.mse_loss(ora(x.double()), tgt.double()).backward()
    print(f"  oracle 135x240 train step in float64: {time.perf_counter() - t0:.1f} s")
    return {n: p.grad.detach() for n, p in ora.named().items() if p.grad is not None}


@pytest.mark.timeout(900)
def test_135x240_training_step_fp32_every_gradient(oracle_135, oracle_135_f64):
    """Every one of the 131 gradients of the exact-fp32 mode against the fp32 CPU oracle: whole-tensor relative L2 and the
    max-normalised element error both under 1e-3 - or, for a tensor that is not, ATTRIBUTED with the float64 oracle: the HIP
    gradient may sit at most 4x as far from the float64 gradient as the fp32 CPU oracle itself does (+ 2e-5, the kernels'
    own tolerance).  A tensor listed under that rule is one whose fp32 evaluation is ill-conditioned at this size (tiny
    gradients formed by cancellation, ReLU pre-activations within rounding of zero), not one the kernels get wrong: an
    ordering / accumulation defect would put the HIP gradient far from float64 where the CPU oracle is close."""
    x, tgt, o_out, o_loss, ora = oracle_135
    g64 = oracle_135_f64
    net, _ = _pair(False, True)
    out = net(x.cuda())
    loss = F.mse_loss(out, tgt.cuda())
    loss.backward()
    assert (out.detach().cpu() - o_out).abs().max().item() < 1e-3
    assert abs(loss.item() - o_loss) < 1e-5 * o_loss
    onamed = ora.named()
    listed, failed, num, den = [], [], 0.0, 0.0
    for n, p in net.named_parameters():
        g, r, t = p.grad.detach().double().cpu(), onamed[n].grad.double(), g64[n]
        e = ((g - r).abs().max() / r.abs().max().clamp_min(1e-300)).item()
        l2 = ((g - r).norm() / r.norm().clamp_min(1e-300)).item()
        num += float(((g - r) ** 2).sum())
        den += float((r ** 2).sum())
        if e < 1e-3 and l2 < 1e-3:
            continue
        # attribution: distance to the float64 gradient, max-normalised and whole-tensor L2, HIP vs the fp32 CPU oracle
        tm, tn = t.abs().max().clamp_min(1e-300), t.norm().clamp_min(1e-300)
        hip_e, ora_e = ((g - t).abs().max() / tm).item(), ((r - t).abs().max() / tm).item()
        hip_l2, ora_l2 = ((g - t).norm() / tn).item(), ((r - t).norm() / tn).item()
        row = (n, f"vs fp32 oracle: max {e:.2e} L2 {l2:.2e}", f"vs float64: HIP max {hip_e:.2e} L2 {hip_l2:.2e}, "
               f"CPU fp32 max {ora_e:.2e} L2 {ora_l2:.2e}")
        listed.append(row)
        if not (hip_e <= 4 * ora_e + 2e-5 and hip_l2 <= 4 * ora_l2 + 2e-5):
            failed.append(row)
    glob = (num / den) ** 0.5
    print(f"  135x240 fp32: loss {loss.item():.7f} vs {o_loss:.7f}, whole-vector rel L2 {glob:.2e}; tensors over 1e-3 of the fp32 "
          f"oracle, attributed with float64 ({len(listed)}):")
    for row in listed:
        print("    ", *row)
    assert not failed, failed
    assert len(listed) <= 12, listed          # (a handful of ill-conditioned tensors, not a systematic offset)
    assert glob < 1e-4
    sd = net.state_dict()
    for n in sr_oracle.buffer_shapes(64):
        a, b = sd[n].double().cpu(), onamed[n].double()
        assert ((a - b).abs().max() / b.abs().max().clamp_min(1e-30)).item() < 1e-3, n


@pytest.mark.timeout(900)
def test_135x240_training_step_benchmarked_bf16_mode(oracle_135):
    x, tgt, o_out, o_loss, ora = oracle_135
    net, _ = _pair(True, True)
    out = net(x.cuda())
    loss = F.mse_loss(out, tgt.cuda())
    loss.backward()
    psnr = sr_oracle.compute_psnr(out.detach().cpu(), o_out)
    onamed = ora.named()
    cos_min, at, flow_min, dot, na, nb = 1.0, None, 1.0, 0.0, 0.0, 0.0
    for n, p in net.named_parameters():
        a, b = p.grad.detach().double().cpu().reshape(-1), onamed[n].grad.double().reshape(-1)
        cos = float((a @ b) / (a.norm() * b.norm()).clamp_min(1e-300))
        dot += float(a @ b); na += float(a @ a); nb += float(b @ b)
        if "motion_estimator" in n:
            flow_min = min(flow_min, cos)
        elif cos < cos_min:
            cos_min, at = cos, n
    whole = dot / (na * nb) ** 0.5
    print(f"  135x240 bf16: PSNR vs fp32 oracle {psnr:.1f} dB, loss {loss.item():.6f} vs {o_loss:.6f}, min non-flow gradient "
          f"cosine {cos_min:.5f} at {at} (flow net {flow_min:.4f}), whole-gradient cosine {whole:.6f}")
    assert psnr > 41.0
    assert abs(loss.item() - o_loss) < 2e-3 * o_loss
    assert cos_min > 0.98 and whole > 0.995


@pytest.mark.timeout(900)
def test_540x960_eval_forward_both_modes():
    x = synth.formula_clip(1, 3, 540, 960, seed=15)
    _, ora = _pair(False, False)
    t0 = time.perf_counter()
    with torch.no_grad():
        o_out = ora(x)
    print(f"  oracle 540x960 eval forward: {time.perf_counter() - t0:.1f} s")
    frac = ((o_out == 0) | (o_out == 1)).float().mean().item()
    assert 0.0 < frac < 0.9                                    # some, not all, outputs sit on a clamp rail
    for bf16 in (False, True):
        net, _ = _pair(bf16, False)
        with torch.no_grad():
            out = net(x.cuda()).cpu()
        err = (out - o_out).abs().max().item()
        psnr = sr_oracle.compute_psnr(out, o_out)
        print(f"  540x960 eval forward, {'bf16' if bf16 else 'fp32'} mode: max abs err {err:.2e}, PSNR vs oracle {psnr:.1f} dB "
              f"(clamped fraction {frac:.3f})")
        if bf16:
            assert psnr > 41.0 and err < 0.1
        else:
            assert err < 1e-3 and psnr > 90.0


@pytest.mark.timeout(900)
def test_batch_of_8_full_size_clip_7_alone_vs_in_the_batch():
    """the benchmarked mode at the benchmark's batch (eval mode: BatchNorm uses running statistics, so clips are independent
    in forward AND backward)"""
    net, _ = _pair(True, False)
    g = torch.Generator(device="cuda").manual_seed(17)
    x8 = torch.rand(8, 3, 3, 540, 960, device="cuda", generator=g)
    tgt = torch.rand(1, 3, 1080, 1920, device="cuda", generator=g)
    out8 = net(x8)
    F.mse_loss(out8[7:8], tgt).backward()
    g8 = torch.cat([p.grad.flatten() for p in net.parameters()]).double()
    out8 = out8.detach()
    net.zero_grad(set_to_none=True)
    out1 = net(x8[7:8])
    F.mse_loss(out1, tgt).backward()
    g1 = torch.cat([p.grad.flatten() for p in net.parameters()]).double()
    assert torch.equal(out8[7:8], out1.detach())
    with torch.no_grad():
        assert torch.equal(net(x8[:1]), out8[:1])             # ... and clip 0, the other end of the tile range
    rel = ((g8 - g1).norm() / g1.norm()).item()
    worst = 0.0
    off = 0
    for n, p in net.named_parameters():
        k = p.numel()
        a, b = g8[off:off + k], g1[off:off + k]
        worst = max(worst, ((a - b).abs().max() / b.abs().max().clamp_min(1e-300)).item())
        off += k
    print(f"  B=8 vs B=1 gradients of a clip-7 loss: whole-vector rel L2 {rel:.2e}, worst tensor (max-normalised) {worst:.2e}")
    assert g1.abs().max() > 0 and rel < 1e-4 and worst < 1e-3
