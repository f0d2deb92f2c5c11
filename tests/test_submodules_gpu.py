"""GPU: the sub-modules of the SR network called on their own (reference nerve_cl/models/super_resolution.py:22-253 are
ordinary nn.Modules; inside SuperResolutionNet ours are parameter holders of one fused kernel schedule).  FeatureExtractor,
MotionEstimator, TemporalAggregator, ResidualDenseBlock and the module-level warp_features compute through the libnvq kernels
chained by nerve_cl._ops: forward values and every gradient (inputs and parameters) against the oracle's component functions
on the same formula weights; BatchNorm buffers after a training-mode call."""
import pytest
import torch

from oracle import sr_oracle, synth

pytestmark = pytest.mark.gpu
Fc, NB, T = 32, 1, 3


def rnd(*shape, seed=0, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return (torch.rand(*shape, generator=g) * 2 - 1) * scale


def rel(a, b):
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    return ((a - b).abs().max() / b.abs().max().clamp_min(1e-30)).item()


@pytest.fixture()
def pair():
    from nerve_cl.models import SuperResolutionNet
    sd = synth.formula_state(3, 2, Fc, NB, 1, gain=synth.GOLDEN_GAIN)
    net = SuperResolutionNet(3, 2, Fc, NB, 1)
    net.load_state_dict(sd)
    net = net.cuda().train()
    P = {k: (v.clone().requires_grad_(True) if v.is_floating_point() and "running" not in k else v.clone()) for k, v in sd.items()}
    return net, P


def _check_param_grads(module, prefix, P, tol=2e-4):
    for n, p in module.named_parameters():
        ref = P[prefix + n].grad
        assert p.grad is not None and ref is not None, n
        assert rel(p.grad, ref) < tol, (n, rel(p.grad, ref))


def test_feature_extractor_forward_backward_and_bn_buffers(pair):
    net, P = pair
    x = rnd(2, 3, 19, 37, seed=1).abs()
    xg = x.cuda().requires_grad_(True)
    xo = x.clone().requires_grad_(True)
    dy = rnd(2, Fc, 19, 37, seed=2)
    out = net.feature_extractor(xg)
    out.backward(dy.cuda())
    ref = sr_oracle.feature_extractor(P, xo, True)
    ref.backward(dy)
    assert out.shape == ref.shape and rel(out, ref) < 2e-5
    assert rel(xg.grad, xo.grad) < 2e-4
    _check_param_grads(net.feature_extractor, "feature_extractor.", P)
    sd = net.state_dict()
    for k in range(3):
        for b in ("running_mean", "running_var"):
            name = f"feature_extractor.body.{k}.bn.{b}"
            assert rel(sd[name], P[name]) < 1e-5, name
        nbt = f"feature_extractor.body.{k}.bn.num_batches_tracked"
        assert int(sd[nbt]) == int(P[nbt])                    # one more than the loaded state, as in the oracle's call
    net.eval()
    with torch.no_grad():
        assert rel(net.feature_extractor(x.cuda()), sr_oracle.feature_extractor(P, x, False)) < 2e-5


def test_motion_estimator_and_warp_features(pair):
    from nerve_cl.models.super_resolution import warp_features
    net, P = pair
    f1, f2 = rnd(2, Fc, 17, 23, seed=3), rnd(2, Fc, 17, 23, seed=4)
    a, b = f1.cuda().requires_grad_(True), f2.cuda().requires_grad_(True)
    ao, bo = f1.clone().requires_grad_(True), f2.clone().requires_grad_(True)
    dfl = rnd(2, 2, 17, 23, seed=5)
    flow = net.motion_estimator(a, b)
    flow.backward(dfl.cuda())
    ref = sr_oracle.flow_net(P, sr_oracle.correlation(ao, bo))
    ref.backward(dfl)
    assert flow.shape == (2, 2, 17, 23) and rel(flow, ref) < 2e-5
    assert rel(a.grad, ao.grad) < 2e-4 and rel(b.grad, bo.grad) < 2e-4
    _check_param_grads(net.motion_estimator, "motion_estimator.", P)
    # warp_features with flows of a pixel or two, both gradients
    fl = rnd(2, 2, 17, 23, seed=6, scale=1.7)
    fg, flg = f1.cuda().requires_grad_(True), fl.cuda().requires_grad_(True)
    fo, flo = f1.clone().requires_grad_(True), fl.clone().requires_grad_(True)
    dw = rnd(2, Fc, 17, 23, seed=7)
    w = warp_features(fg, flg)
    w.backward(dw.cuda())
    wr = sr_oracle.warp(fo, flo)
    wr.backward(dw)
    assert rel(w, wr) < 2e-5 and rel(fg.grad, fo.grad) < 2e-4 and rel(flg.grad, flo.grad) < 2e-4


def test_temporal_aggregator_and_residual_dense_block(pair):
    net, P = pair
    feats = [rnd(2, Fc, 12, 20, seed=10 + t) for t in range(T)]
    fg = [f.cuda().requires_grad_(True) for f in feats]
    fo = [f.clone().requires_grad_(True) for f in feats]
    dy = rnd(2, Fc, 12, 20, seed=20)
    out = net.temporal_aggregator(fg)
    out.backward(dy.cuda())
    ref = sr_oracle.temporal_aggregator(P, fo)
    ref.backward(dy)
    assert rel(out, ref) < 2e-5
    for t in range(T):
        assert rel(fg[t].grad, fo[t].grad) < 2e-4, t
    _check_param_grads(net.temporal_aggregator, "temporal_aggregator.", P)
    with pytest.raises(RuntimeError, match="aligned feature maps"):
        net.temporal_aggregator(fg[:2])
    # one dense block
    x = rnd(2, Fc, 13, 21, seed=30)
    xg, xo = x.cuda().requires_grad_(True), x.clone().requires_grad_(True)
    blk = net.residual_blocks[0]
    y = blk(xg)
    y.backward(dy[:, :, :13, :20].new_ones(2, Fc, 13, 21).cuda() * 0.5)
    yr = sr_oracle.residual_dense_block(P, 0, xo)
    yr.backward(torch.full((2, Fc, 13, 21), 0.5))
    assert rel(y, yr) < 2e-5 and rel(xg.grad, xo.grad) < 2e-4
    _check_param_grads(blk, "residual_blocks.0.", P)


def test_cpu_tensors_are_refused(pair):
    net, _ = pair
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        net.feature_extractor(torch.zeros(1, 3, 8, 8))
