"""GPU: every differentiable NHWC op of nerve_cl._ops (the layers FrameRecoveryNet and the stand-alone layer modules are
built from; kernels in csrc/fr_ops.hip + the conv / depthwise / CBAM / correlation kernels of the SR path) against the
PyTorch CPU operator it replaces, forward and backward, fp32 math mode, tolerance 2e-5 of the reference tensor's maximum
(1e-3 is what north_star asks).  Shapes include odd sizes, channel counts that are not multiples of 4 (230) and ties."""
import pytest
import torch
import torch.nn as nn
import torch.nn.functional as F

pytestmark = pytest.mark.gpu
TOL = 2e-5


def rel(a, b):
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    return ((a - b).abs().max() / b.abs().max().clamp_min(1e-30)).item()


def nhwc(x):                       # (N,C,H,W) cpu -> [N,H,W,pad4(C)] cuda
    from nerve_cl import _ops
    return _ops.ToNHWC.apply(x.cuda())


def nchw(y, C):
    from nerve_cl import _ops
    return _ops.ToNCHW.apply(y, C)


def run_pair(gpu_fn, cpu_fn, inputs, params, C_out, tol=TOL):
    """inputs / params: lists of CPU tensors (requires_grad where a gradient is wanted).  gpu_fn gets NHWC cuda inputs and
    cuda params, returns NHWC; cpu_fn gets NCHW cpu tensors."""
    gi = [t.detach().clone().requires_grad_(t.requires_grad) for t in inputs]
    gp = [t.detach().cuda().requires_grad_(t.requires_grad) for t in params]
    y = nchw(gpu_fn([nhwc(t) for t in gi], gp), C_out)
    r = cpu_fn(inputs, params)
    assert y.shape == r.shape, (y.shape, r.shape)
    assert rel(y, r) < tol, ("forward", rel(y, r))
    torch.manual_seed(5)
    dout = torch.randn_like(r)
    y.backward(dout.cuda())
    r.backward(dout)
    for k, (a, b) in enumerate(zip(gi + gp, inputs + params)):
        if b.requires_grad:
            assert a.grad is not None, k
            assert rel(a.grad, b.grad) < tol, ("grad", k, rel(a.grad, b.grad))
    return y


def T(*shape, grad=True, scale=1.0):
    return (torch.randn(*shape) * scale).requires_grad_(grad)


def test_layout_round_trip_and_padding():
    from nerve_cl import _ops
    torch.manual_seed(0)
    for N, C, H, W in ((2, 3, 7, 9), (1, 230, 5, 6), (3, 64, 17, 33), (1, 1, 4, 4)):
        x = torch.randn(N, C, H, W)
        y = _ops.ToNHWC.apply(x.cuda())
        assert y.shape == (N, H, W, (C + 3) // 4 * 4)
        assert torch.equal(y[..., :C].cpu(), x.permute(0, 2, 3, 1))
        assert (y[..., C:] == 0).all()
        assert torch.equal(_ops.ToNCHW.apply(y, C).cpu(), x)


@pytest.mark.parametrize("cin,cout,k,relu,bias", [(64, 32, 3, True, True), (4, 32, 3, False, False), (230, 128, 1, False, False),
                                                  (64, 230, 3, False, False), (512, 64, 1, False, True), (8, 3, 3, False, True)])
def test_conv(cin, cout, k, relu, bias):
    from nerve_cl import _nvq, _ops
    torch.manual_seed(1)
    creal = 3 if cin == 4 else cin
    x, w = T(2, creal, 9, 13), T(cout, creal, k, k, scale=(creal * k * k) ** -0.5)
    ps = [w] + ([T(cout, scale=0.1)] if bias else [])
    run_pair(lambda i, p: _ops.Conv.apply(i[0], p[0], p[1] if bias else None, relu, _nvq.MATH_F32),
             lambda i, p: (F.relu if relu else (lambda t: t))(F.conv2d(i[0], p[0], p[1] if bias else None, padding=k // 2)),
             [x], ps, cout)


@pytest.mark.parametrize("C", [16, 64, 256])
def test_depthwise_conv(C):
    from nerve_cl import _ops
    torch.manual_seed(2)
    run_pair(lambda i, p: _ops.DwConv.apply(i[0], p[0]), lambda i, p: F.conv2d(i[0], p[0], None, padding=1, groups=C),
             [T(2, C, 10, 11)], [T(C, 1, 3, 3, scale=0.3)], C)


@pytest.mark.parametrize("C,training,relu,with_res", [(16, True, True, False), (230, True, True, False), (64, True, True, True),
                                                      (128, True, False, False), (460, False, True, False),
                                                      (32, False, True, True)])
def test_batchnorm_modes(C, training, relu, with_res):
    from nerve_cl import _ops
    torch.manual_seed(3)
    x = T(3, C, 6, 7)
    res = [T(3, C, 6, 7)] if with_res else []
    gamma, beta = (1 + 0.2 * torch.randn(C)).requires_grad_(True), T(C, scale=0.1)
    rm, rv = 0.1 * torch.randn(C), 0.5 + torch.rand(C)
    bn_cpu = nn.BatchNorm2d(C)
    bn_cpu.running_mean.copy_(rm)
    bn_cpu.running_var.copy_(rv)
    bn_cpu.train(training)
    rm_g, rv_g = rm.clone().cuda(), rv.clone().cuda()

    def cpu(i, p):
        y = F.batch_norm(i[0], bn_cpu.running_mean, bn_cpu.running_var, p[0], p[1], training, 0.1, 1e-5)
        if with_res:
            y = y + i[1]
        return F.relu(y) if relu else y

    run_pair(lambda i, p: _ops.BatchNorm.apply(i[0], p[0], p[1], i[1] if with_res else None, rm_g, rv_g, training, relu),
             cpu, [x] + res, [gamma, beta], C)
    assert rel(rm_g, bn_cpu.running_mean) < TOL and rel(rv_g, bn_cpu.running_var) < TOL


@pytest.mark.parametrize("k,s,p,H,W", [(3, 2, 1, 16, 24), (3, 2, 1, 9, 7), (2, 2, 0, 10, 12), (2, 2, 0, 5, 7)])
def test_maxpool_with_ties(k, s, p, H, W):
    """after ReLU whole windows are zero: the FIRST maximum takes the gradient, like PyTorch's CPU kernel"""
    from nerve_cl import _ops
    torch.manual_seed(4)
    x = F.relu(torch.randn(2, 8, H, W)).requires_grad_(True)
    run_pair(lambda i, q: _ops.MaxPool.apply(i[0], k, s, p), lambda i, q: F.max_pool2d(i[0], k, s, p), [x], [], 8, tol=0.0 + 1e-7)


def test_subsample_resize_depth_to_space_groupmean_cat():
    from nerve_cl import _ops
    torch.manual_seed(5)
    run_pair(lambda i, p: _ops.Subsample2.apply(i[0]), lambda i, p: i[0][:, :, ::2, ::2], [T(2, 8, 9, 12)], [], 8)
    for (H, W, OH, OW) in ((10, 12, 3, 3), (6, 6, 40, 40), (7, 5, 7, 5), (48, 48, 40, 40), (5, 9, 2, 4)):
        run_pair(lambda i, p: _ops.Resize.apply(i[0], OH, OW),
                 lambda i, p: F.interpolate(i[0], size=(OH, OW), mode="bilinear", align_corners=False), [T(2, 8, H, W)], [], 8)
    x = T(2, 32, 5, 6)
    # phase-major depth-to-space: channel (i*2+j)*Co + c -> pixel (2y+i, 2x+j)
    run_pair(lambda i, p: _ops.DepthToSpace2.apply(i[0]),
             lambda i, p: i[0].view(2, 2, 2, 8, 5, 6).permute(0, 3, 4, 1, 5, 2).reshape(2, 8, 10, 12), [x], [], 8)
    xt = T(6, 8, 4, 5)                       # time-major [T*B] with T = 3, B = 2
    run_pair(lambda i, p: _ops.GroupMean.apply(i[0], 3, 8), lambda i, p: i[0].view(3, 2, 8, 4, 5).mean(0), [xt], [], 8)
    run_pair(lambda i, p: _ops.Cat2.apply(i[0], i[1]), lambda i, p: torch.cat(i, dim=1), [T(2, 8, 4, 5), T(2, 12, 4, 5)], [], 20)


@pytest.mark.parametrize("ci,co", [(64, 32), (16, 8), (128, 64)])
def test_conv_transpose_k4s2p1(ci, co):
    from nerve_cl import _nvq, _ops
    torch.manual_seed(6)
    run_pair(lambda i, p: _ops.ConvT.apply(i[0], p[0], _nvq.MATH_F32),
             lambda i, p: F.conv_transpose2d(i[0], p[0], None, stride=2, padding=1),
             [T(2, ci, 5, 7)], [T(ci, co, 4, 4, scale=(ci * 4) ** -0.5)], co)


@pytest.mark.parametrize("T_,B,ci,co", [(2, 2, 32, 64), (4, 1, 232, 128), (1, 2, 16, 16)])
def test_temporal_conv_3x1x1(T_, B, ci, co):
    from nerve_cl import _nvq, _ops
    torch.manual_seed(7)
    creal = 230 if ci == 232 else ci
    x = T(T_ * B, creal, 5, 6)               # time-major images

    def cpu(i, p):
        v = i[0].view(T_, B, creal, 5, 6).permute(1, 2, 0, 3, 4)          # (B,C,T,H,W)
        y = F.conv3d(v, p[0], None, padding=(1, 0, 0))
        return y.permute(2, 0, 1, 3, 4).reshape(T_ * B, co, 5, 6)

    run_pair(lambda i, p: _ops.TemporalConv.apply(i[0], p[0], T_, _nvq.MATH_F32), cpu, [x],
             [T(co, creal, 3, 1, 1, scale=(3 * creal) ** -0.5)], co)


def to_tc(v, Cp):
    """(B,C,T,H,W) -> time-in-channels [B,H,W,T*Cp] (frame t = channels t*Cp .. t*Cp+C-1, the rest zero); torch ops, so
    gradients flow back to v"""
    B, C, T_, H, W = v.shape
    return F.pad(v.permute(0, 3, 4, 2, 1), (0, Cp - C)).reshape(B, H, W, T_ * Cp).contiguous()


def from_tc(y, C, T_):
    B, H, W, ld = y.shape
    return y.view(B, H, W, T_, ld // T_)[..., :C].permute(0, 4, 3, 1, 2)


@pytest.mark.parametrize("T_,B,ci,co", [(2, 2, 32, 64), (4, 1, 230, 128), (3, 2, 16, 16), (2, 1, 3, 16)])
def test_2plus1d_convs_time_in_channels(T_, B, ci, co):
    """the (1,3,3) and (3,1,1) convolutions of TemporalConv3D (reference efficient_layers.py:259-278) on the time-in-channels
    layout against F.conv3d; mean over time against the mean"""
    from nerve_cl import _nvq, _ops
    torch.manual_seed(13)
    H, W, Cpi, Cpo = 5, 6, (ci + 3) // 4 * 4, (co + 3) // 4 * 4
    x = T(B, ci, T_, H, W)
    wt, ws = T(co, ci, 3, 1, 1, scale=(3 * ci) ** -0.5), T(co, ci, 1, 3, 3, scale=(9 * ci) ** -0.5)
    for name, w, fn, pad in (("temporal", wt, lambda a, b: _ops.TemporalConvTC.apply(a, b, T_, _nvq.MATH_F32), (1, 0, 0)),
                             ("spatial", ws, lambda a, b: _ops.SpatialConvTC.apply(a, b.view(co, ci, 3, 3), T_, _nvq.MATH_F32), (0, 1, 1))):
        xg, wg = x.detach().cuda().requires_grad_(True), w.detach().cuda().requires_grad_(True)
        y = from_tc(fn(to_tc(xg, Cpi), wg), co, T_)
        r = F.conv3d(x, w, None, padding=pad)
        assert rel(y, r) < TOL, (name, rel(y, r))
        d = torch.randn_like(r)
        y.backward(d.cuda())
        gx, gw = torch.autograd.grad(r, [x, w], d)
        assert rel(xg.grad, gx) < TOL and rel(wg.grad, gw) < TOL, (name, rel(xg.grad, gx), rel(wg.grad, gw))
    xg = x.detach().cuda().requires_grad_(True)
    m = _ops.GroupMeanTC.apply(to_tc(xg, Cpi), T_, ci)
    assert rel(m[..., :ci].permute(0, 3, 1, 2), x.mean(2)) < TOL and (m[..., ci:] == 0).all()
    m.sum().backward()
    assert rel(xg.grad, torch.full_like(x, 1.0 / T_)) < TOL


@pytest.mark.parametrize("Co,H,W", [(16, 32, 48), (64, 21, 17)])
def test_stem_7x7_stride2(Co, H, W):
    from nerve_cl import _ops
    torch.manual_seed(8)
    x = torch.randn(2, 4, H, W)
    w = T(Co, 4, 7, 7, scale=196 ** -0.5)
    run_pair(lambda i, p: _ops.Stem7.apply(i[0], p[0]), lambda i, p: F.conv2d(i[0], p[0], None, stride=2, padding=3), [x], [w], Co)


@pytest.mark.parametrize("C", [64, 256])
def test_cbam(C):
    from nerve_cl import _ops
    torch.manual_seed(9)
    w1, w2, w7 = T(C // 16, C, scale=C ** -0.5), T(C, C // 16, scale=0.3), T(1, 2, 7, 7, scale=0.1)

    def cpu(i, p):
        x = i[0]
        ca = torch.sigmoid(F.relu(x.mean(dim=(2, 3)) @ p[0].t()) @ p[1].t())
        xc = x * ca[:, :, None, None]
        sm = torch.cat([xc.mean(1, keepdim=True), xc.max(1, keepdim=True)[0]], 1)
        return xc * torch.sigmoid(F.conv2d(sm, p[2], None, padding=3))

    run_pair(lambda i, p: _ops.CBAMFn.apply(i[0], p[0], p[1], p[2]), cpu, [T(2, C, 9, 11)], [w1, w2, w7], C)


def test_fusion_mix_tanh_blend():
    from nerve_cl import _ops
    torch.manual_seed(10)
    C = 64
    al, lg, sp, tp = T(2, C, 5, 6), T(2, 2, 5, 6), T(2, C, 5, 6), T(2, C, 5, 6)

    def cpu(i, p):
        a = torch.softmax(i[1], dim=1)
        return i[0] + a[:, 0:1] * i[2].mean(1, keepdim=True) + a[:, 1:2] * i[3].mean(1, keepdim=True)

    run_pair(lambda i, p: _ops.FusionMix.apply(*i), cpu, [al, lg, sp, tp], [], C)
    run_pair(lambda i, p: _ops.Tanh.apply(i[0]), lambda i, p: torch.tanh(i[0]), [T(2, 3, 7, 9)], [], 3)
    # mask blend: NCHW frame and mask are data, the recovered image is NHWC
    frame, mask = torch.rand(2, 3, 6, 8), (torch.rand(2, 1, 6, 8) > 0.5).float()
    rec = T(2, 3, 6, 8)
    rec_g = rec.detach().clone().requires_grad_(True)
    out = _ops.MaskBlend.apply(frame.cuda(), nhwc(rec_g), mask.cuda())
    ref = frame * (1 - mask) + rec * mask
    assert rel(out, ref) < TOL
    d = torch.randn_like(ref)
    out.backward(d.cuda())
    ref.backward(d)
    assert rel(rec_g.grad, rec.grad) < TOL


def test_bf16_storage_variants_of_the_frame_recovery_ops():
    """the same kernels with bf16-stored activations (FrameRecoveryNet's throughput mode): each op on a bf16 tensor against
    the fp32 op on the same (bf16-representable) values; differences = one bf16 rounding of the result (2^-8 relative)"""
    from nerve_cl import _nvq, _ops
    torch.manual_seed(12)
    B16, TOLB = torch.bfloat16, 1.2e-2

    def pair(fn, x, C, *params, grad_in=True, keeps_dtype=True):
        """fn(x_nhwc, *params) -> nhwc; returns (bf16 result, fp32 result) and checks outputs and input gradients"""
        xq = x.bfloat16().float()
        outs, grads = [], []
        for dt in (B16, torch.float32):
            xi = nhwc(xq.clone().requires_grad_(grad_in))
            xin = _ops.Cast.apply(xi, dt, C)
            leaf = xin.detach().requires_grad_(grad_in)
            y = fn(leaf, *[p.detach().cuda().requires_grad_(True) for p in params])
            assert y.dtype == (dt if keeps_dtype else torch.float32)
            yf = _ops.Cast.apply(y, torch.float32, y.shape[-1])
            torch.manual_seed(1)
            d = torch.randn_like(yf).bfloat16().float()
            if grad_in:
                yf.backward(d)
                grads.append(_ops.Cast.apply(leaf.grad, torch.float32, C))
            outs.append(yf)
        assert rel(outs[0], outs[1]) < TOLB, rel(outs[0], outs[1])
        if grad_in:
            assert rel(grads[0], grads[1]) < 2 * TOLB, rel(grads[0], grads[1])

    C = 64
    gamma, beta = 1 + 0.2 * torch.randn(C), 0.1 * torch.randn(C)
    rm, rv = torch.zeros(C).cuda(), torch.ones(C).cuda()
    pair(lambda x, g, b: _ops.BatchNorm.apply(x, g, b, None, rm.clone(), rv.clone(), True, True), torch.randn(2, C, 9, 10), C, gamma, beta)
    pair(lambda x, g, b: _ops.BatchNorm.apply(x, g, b, x, rm.clone(), rv.clone(), True, True), torch.randn(2, C, 9, 10), C, gamma, beta)
    g230, b230 = 1 + 0.2 * torch.randn(230), 0.1 * torch.randn(230)
    pair(lambda x, g, b: _ops.BatchNorm.apply(x, g, b, None, torch.zeros(230).cuda(), torch.ones(230).cuda(), True, True),
         torch.randn(2, 230, 6, 7), 230, g230, b230)
    pair(lambda x: _ops.MaxPool.apply(x, 3, 2, 1), F.relu(torch.randn(2, C, 12, 14)), C)
    pair(lambda x: _ops.MaxPool.apply(x, 2, 2, 0), torch.randn(2, C, 12, 14), C)
    pair(lambda x: _ops.Subsample2.apply(x), torch.randn(2, C, 9, 12), C)
    pair(lambda x: _ops.DepthToSpace2.apply(x), torch.randn(2, C, 5, 6), C)
    pair(lambda x: _ops.GroupMean.apply(x, 2, C), torch.randn(4, C, 5, 6), C, keeps_dtype=False)   # the mean is fp32
    pair(lambda x, w: _ops.DwConv.apply(x, w), torch.randn(2, C, 9, 10), C, 0.3 * torch.randn(C, 1, 3, 3))
    pair(lambda x, w: _ops.Conv.apply(x, w, None, False, _nvq.MATH_BF16), torch.randn(2, C, 9, 10), C, torch.randn(230, C, 3, 3) / 24)
    pair(lambda x, w: _ops.Conv.apply(x, w, None, False, _nvq.MATH_BF16), torch.randn(2, 230, 9, 10), 230, torch.randn(128, 230, 1, 1) / 15)
    pair(lambda x, w: _ops.TemporalConv.apply(x, w, 2, _nvq.MATH_BF16), torch.randn(4, 230, 5, 6), 230, torch.randn(128, 230, 3, 1, 1) / 26)
    pair(lambda x, w: _ops.ConvT.apply(x, w, _nvq.MATH_BF16), torch.randn(2, C, 5, 6), C, torch.randn(C, 32, 4, 4) / 16)
    # time-in-channels (2+1)D convolutions: 2 frames x 64 channels side by side
    pair(lambda x, w: _ops.TemporalConvTC.apply(x, w, 2, _nvq.MATH_BF16), torch.randn(2, 2 * C, 5, 6), 2 * C, torch.randn(128, C, 3, 1, 1) / 14)
    pair(lambda x, w: _ops.SpatialConvTC.apply(x, w, 2, _nvq.MATH_BF16), torch.randn(2, 2 * C, 5, 6), 2 * C, torch.randn(128, C, 3, 3) / 24)
    pair(lambda x: _ops.GroupMeanTC.apply(x, 2, C), torch.randn(2, 2 * C, 5, 6), 2 * C, keeps_dtype=False)
    x4 = torch.randn(2, 4, 20, 24)
    w7 = (torch.randn(16, 4, 7, 7) / 14).cuda()
    a = _ops.Stem7.apply(nhwc(x4), w7, B16)
    b = _ops.Stem7.apply(nhwc(x4), w7, torch.float32)
    assert a.dtype == B16 and rel(a.float(), b) < TOLB


def test_standalone_layer_modules_match_the_reference_layers_semantics():
    """the calls of the reference's tests/test_models.py:19-38 (on HIP tensors) plus values against torch modules holding
    the same parameters"""
    from nerve_cl.models.layers import (CBAM, DepthwiseSeparableConv, LiteFlowNetCorrelation, PixelShuffleUpsampler,
                                        ResidualBlock, TemporalConv3D)
    torch.manual_seed(11)
    layer = DepthwiseSeparableConv(32, 64).cuda()
    x = torch.randn(2, 32, 16, 16)
    y = layer(x.cuda())
    assert y.shape == (2, 64, 16, 16)
    ref = F.relu(F.batch_norm(F.conv2d(F.conv2d(x, layer.depthwise.weight.cpu(), None, padding=1, groups=32),
                                       layer.pointwise.weight.cpu()), None, None, layer.bn.weight.cpu(), layer.bn.bias.cpu(), True))
    assert rel(y, ref) < TOL
    y.sum().backward()
    assert layer.depthwise.weight.grad is not None and layer.bn.weight.grad is not None
    up = PixelShuffleUpsampler(64, scale_factor=2).cuda()
    x = torch.randn(2, 64, 16, 16)
    y = up(x.cuda())
    assert y.shape == (2, 3, 32, 32)
    assert rel(y, F.pixel_shuffle(F.conv2d(x, up.conv.weight.cpu(), up.conv.bias.cpu(), padding=1), 2)) < TOL
    y.square().mean().backward()
    xg = x.clone().requires_grad_(True)
    F.pixel_shuffle(F.conv2d(xg, up.conv.weight.detach().cpu().requires_grad_(True), up.conv.bias.detach().cpu(), padding=1), 2).square().mean().backward()
    rb = ResidualBlock(64).cuda()
    assert rb(torch.randn(2, 64, 16, 16).cuda()).shape == (2, 64, 16, 16)
    assert CBAM(64).cuda()(torch.randn(2, 64, 8, 8).cuda()).shape == (2, 64, 8, 8)
    tc = TemporalConv3D(3, 64).cuda()
    assert tc(torch.randn(2, 3, 4, 8, 8).cuda()).shape == (2, 64, 4, 8, 8)
    v = torch.randn(2, 3, 4, 8, 8)
    want = tc(v.cuda())                                        # time-major path
    got = from_tc(tc.forward_tc(to_tc(v.cuda(), 4), 4), 64, 4)  # time-in-channels path, same module
    assert rel(got, want) < TOL
    corr = LiteFlowNetCorrelation()
    a, b = torch.randn(1, 16, 6, 7), torch.randn(1, 16, 6, 7)
    c = corr(a.cuda(), b.cuda())
    bp = F.pad(b, (4, 4, 4, 4))
    ref = torch.stack([(a * bp[:, :, i:i + 6, j:j + 7]).mean(1) for i in range(9) for j in range(9)], 1)
    assert rel(c, ref) < TOL
    with pytest.raises(RuntimeError):
        layer(torch.randn(2, 32, 16, 16))                     # CPU tensor: no fallback
