"""GPU: every libnvq kernel, called through the C ABI, against a plain PyTorch fp32 CPU
reference of the same op (torch.nn.functional + autograd).  Tolerances are relative to the
reference tensor's max magnitude and written next to each check."""
import pytest
import torch
import torch.nn.functional as F

from oracle import sr_oracle

pytestmark = pytest.mark.gpu

TOL = 2e-5      # fp32 kernels vs fp32 CPU reference (different summation order only)


@pytest.fixture(scope="module")
def K():
    if not torch.cuda.is_available():
        pytest.fail("GPU tests need a HIP device")
    from nerve_cl import _nvq
    _nvq.lib()
    return _nvq


def rnd(*shape, seed=0, scale=1.0):
    g = torch.Generator().manual_seed(seed + sum(shape))
    return (torch.rand(*shape, generator=g) * 2 - 1) * scale


def to_nhwc(x, ld=None, coff=0, fill=0.0):
    """CPU NCHW -> cuda [N,H,W,ld] with the channels at [coff, coff+C)."""
    n, c, h, w = x.shape
    ld = ld or c
    buf = torch.full((n, h, w, ld), fill, dtype=torch.float32)
    buf[..., coff:coff + c] = x.permute(0, 2, 3, 1)
    return buf.cuda()


def from_nhwc(buf, c=None, coff=0):
    c = buf.shape[-1] - coff if c is None else c
    return buf[..., coff:coff + c].permute(0, 3, 1, 2).contiguous().cpu()


def rel(a, b):
    a, b = a.double().cpu(), b.double().cpu()
    return ((a - b).abs().max() / b.abs().max().clamp_min(1e-30)).item()


def ws_tensor(K):
    return torch.empty(K.wgrad_workspace_bytes() // 4 + (1 << 20), dtype=torch.float32, device="cuda")


# ----------------------------------------------------------------------------- convolution
CONV_CASES = [
    # cin_real, cin_store, cout, k, N, H, W
    (64, 64, 32, 3, 2, 13, 37),
    (81, 96, 128, 3, 1, 9, 33),
    (224, 224, 64, 1, 1, 11, 40),
    (32, 32, 2, 3, 2, 8, 32),
    (64, 64, 12, 3, 1, 17, 5),
    (16, 16, 16, 3, 1, 40, 70),
    (192, 192, 64, 3, 1, 8, 35),
    (64, 64, 5, 3, 1, 6, 9),
]


@pytest.mark.parametrize("cin,cs,cout,k,N,H,W", CONV_CASES)
def test_conv_forward_bias_relu(K, cin, cs, cout, k, N, H, W):
    x, w, b = rnd(N, cin, H, W), rnd(cout, cin, k, k, scale=0.2), rnd(cout)
    ref = F.relu(F.conv2d(x, w, b, padding=k // 2))
    xin = to_nhwc(x, cs)
    wp = K.conv_pack(w.cuda(), False, cs)
    cst = K.pad4(cout)
    out = torch.full((N, H, W, cst + 4), 7.0, device="cuda")
    K.conv_forward(K.Sl(xin), wp, b.cuda(), K.Sl(out, cout, 0), k, relu=True, cout_store=cst)
    assert rel(from_nhwc(out, cout), ref) < TOL
    if cst > cout:
        assert out[..., cout:cst].abs().max().item() == 0.0      # padded channels written as zero
    assert (out[..., cst:] == 7.0).all()                          # nothing beyond cout_store touched


def test_conv_epilogue_scale_residual_slices(K):
    """lff form: out = 0.2*(conv1x1(cat)+b) + cat[:, :F], written into a channel slice."""
    N, H, W, Fc, CAT = 1, 9, 34, 64, 224
    cat, w, b = rnd(N, CAT, H, W), rnd(Fc, CAT, 1, 1, scale=0.1), rnd(Fc)
    ref = F.conv2d(cat, w, b) * 0.2 + cat[:, :Fc]
    catb = to_nhwc(cat)
    out = torch.zeros(N, H, W, CAT, device="cuda")
    K.conv_forward(K.Sl(catb), K.conv_pack(w.cuda(), False, CAT), b.cuda(), K.Sl(out, Fc, 32), 1, alpha=0.2,
                   res=K.Sl(catb, Fc, 0))
    assert rel(from_nhwc(out, Fc, 32), ref) < TOL
    assert out[..., :32].abs().max().item() == 0 and out[..., 96:].abs().max().item() == 0


def test_conv_dense_layer_in_place_concat(K):
    """RDB layer: reads channels [0,96) of the concat buffer, writes relu(conv) at [96,128)."""
    N, H, W = 2, 10, 33
    x, w, b = rnd(N, 96, H, W), rnd(32, 96, 3, 3, scale=0.1), rnd(32)
    ref = F.relu(F.conv2d(x, w, b, padding=1))
    cat = to_nhwc(x, 224)
    K.conv_forward(K.Sl(cat, 96, 0), K.conv_pack(w.cuda(), False, 96), b.cuda(), K.Sl(cat, 32, 96), 3, relu=True)
    assert rel(from_nhwc(cat, 32, 96), ref) < TOL
    assert rel(from_nhwc(cat, 96, 0), x) == 0.0


def test_conv_out2_and_residual_after_relu(K):
    """gff form: out2 = relu(conv+b), out = out2 + centre."""
    N, H, W, Fc = 1, 12, 20, 32
    x, w, b, cen = rnd(N, Fc, H, W), rnd(Fc, Fc, 3, 3, scale=0.1), rnd(Fc), rnd(N, Fc, H, W, seed=5)
    r = F.relu(F.conv2d(x, w, b, padding=1))
    out, out2 = torch.empty(N, H, W, Fc, device="cuda"), torch.empty(N, H, W, Fc, device="cuda")
    al = to_nhwc(cen, 96, 32)
    K.conv_forward(K.Sl(to_nhwc(x)), K.conv_pack(w.cuda(), False, Fc), b.cuda(), K.Sl(out), 3, relu=True,
                   out2=K.Sl(out2), res=K.Sl(al, Fc, 32))
    assert rel(from_nhwc(out2), r) < TOL
    assert rel(from_nhwc(out), r + cen) < TOL


@pytest.mark.parametrize("cin,cout,k", [(96, 32, 3), (224, 64, 1), (81, 128, 3), (64, 2, 3), (64, 12, 3)])
def test_conv_input_gradient_accumulate_mask(K, cin, cout, k):
    """dgrad = forward kernel with the transposed pack; accumulate + ReLU mask on a channel range."""
    N, H, W = 1, 9, 35
    x = rnd(N, cin, H, W).requires_grad_()
    w = rnd(cout, cin, k, k, scale=0.2)
    dy = rnd(N, cout, H, W, seed=3)
    F.conv2d(x, w, None, padding=k // 2).backward(dy)
    old = rnd(N, cin, H, W, seed=9)
    act = rnd(N, cin, H, W, seed=11)
    c0, c1 = (cin // 8) * 4, cin - cin % 4 if cin % 4 else cin
    want = x.grad + old
    m = torch.ones_like(want)
    m[:, c0:c1] = (act[:, c0:c1] > 0).float()
    want = want * m
    cs = K.pad4(cout)
    dyb = to_nhwc(dy, cs)
    wp = K.conv_pack(w.cuda(), True, cs, cin)
    ld = K.pad4(cin)
    out = to_nhwc(old, ld)
    K.conv_forward(K.Sl(dyb), wp, None, K.Sl(out, cin, 0), k, accumulate=True, cout_store=ld,
                   mask=K.Sl(to_nhwc(act, ld)), mask_c0=c0, mask_c1=c1)
    assert rel(from_nhwc(out, cin), want) < TOL


@pytest.mark.parametrize("cin,cin_store,cout,k,N,H,W", [
    (64, 64, 32, 3, 2, 16, 40), (81, 96, 128, 3, 1, 9, 33), (224, 224, 64, 1, 1, 11, 37),
    (32, 32, 2, 3, 1, 8, 31), (64, 64, 12, 3, 1, 17, 33), (192, 192, 64, 3, 1, 8, 64), (64, 64, 3, 3, 2, 5, 7),
])
def test_conv_weight_gradient(K, cin, cin_store, cout, k, N, H, W):
    x = rnd(N, cin, H, W)
    w = rnd(cout, cin, k, k).requires_grad_()
    b = rnd(cout).requires_grad_()
    dy = rnd(N, cout, H, W, seed=3)
    (F.conv2d(x, w, b, padding=k // 2) * 0.5).backward(dy)
    dw = torch.zeros(cout, cin, k, k, device="cuda")
    db = torch.zeros(cout, device="cuda")
    K.conv_wgrad(K.Sl(to_nhwc(x, cin_store)), cin, K.Sl(to_nhwc(dy, K.pad4(cout)), cout, 0), dw, db, ws_tensor(K), k,
                 alpha=0.5)
    assert rel(dw, w.grad) < TOL
    assert rel(db, b.grad) < TOL


def test_conv_weight_gradient_slices_and_accumulate(K):
    N, H, W = 1, 10, 33
    x, dy = rnd(N, 96, H, W), rnd(N, 32, H, W, seed=2)
    w = rnd(32, 96, 3, 3).requires_grad_()
    F.conv2d(x, w, None, padding=1).backward(dy)
    cat = to_nhwc(x, 224)
    dcat = to_nhwc(dy, 224, 96)
    dw = torch.ones(32, 96, 3, 3, device="cuda")
    K.conv_wgrad(K.Sl(cat, 96, 0), 96, K.Sl(dcat, 32, 96), dw, None, ws_tensor(K), 3, accumulate=True)
    assert rel(dw - 1.0, w.grad) < TOL


# ----------------------------------------------------------------------------- feature extractor
@pytest.mark.parametrize("Fc,B,T,H,W,out16", [(64, 2, 3, 37, 70, True), (32, 1, 3, 16, 64, True), (16, 2, 1, 5, 9, False),
                                              (64, 1, 3, 40, 130, False)])
def test_head_forward_matrix_core_mode(K, Fc, B, T, H, W, out16):
    """NVQ_MATH_BF16 head: weights rounded to bf16, the frame as a hi + lo pair of bf16 values (~16 bits), fp32 accumulation
    (head_mfma_kernel, permuted output rows, second K step for the window's last pixel): equal to the fp32 conv of the frame
    with the rounded weights; border, ragged and multi-tile shapes; fp32 and bf16 outputs; the bf16 NHWC-8 copy of the frames."""
    frames = rnd(B, T, 3, H, W).abs()
    w, b = rnd(Fc, 3, 3, 3, scale=0.4), rnd(Fc, scale=0.1)
    c = T // 2
    slots = [c] + [t for t in range(T) if t != c]
    ref = torch.stack([F.relu(F.conv2d(frames[:, t], bf(w), b, padding=1)) for t in slots], 0).reshape(T * B, Fc, H, W)
    out = torch.full((T * B, H, W, Fc), 3.0, device="cuda", dtype=torch.bfloat16 if out16 else torch.float32)
    img8 = torch.full((T * B, H, W, 8), 3.0, device="cuda", dtype=torch.bfloat16)
    K.head_forward(frames.cuda(), slots, w.cuda(), b.cuda(), out, img8=img8, math=K.MATH_BF16)
    if out16:
        assert rel(from_nhwc(out.float()), ref) < 5e-3
    else:
        assert rel(from_nhwc(out), ref) < TOL
    want8 = torch.zeros(T * B, H, W, 8)
    want8[..., :3] = torch.stack([bf(frames[:, t]) for t in slots], 0).reshape(T * B, 3, H, W).permute(0, 2, 3, 1)
    assert torch.equal(img8.float().cpu(), want8)


@pytest.mark.parametrize("Fc,B,T,H,W", [(32, 2, 3, 9, 14), (64, 1, 5, 8, 33)])
def test_head_forward_and_wgrad(K, Fc, B, T, H, W):
    frames = rnd(B, T, 3, H, W).abs()
    w = rnd(Fc, 3, 3, 3, scale=0.4).requires_grad_()
    b = rnd(Fc, scale=0.1).requires_grad_()
    c = T // 2
    slots = [c] + [t for t in range(T) if t != c]
    ref = torch.stack([F.relu(F.conv2d(frames[:, t], w, b, padding=1)) for t in slots], 0)   # [slot,B,F,H,W]
    dout = rnd(T, B, Fc, H, W, seed=4)
    ref.backward(dout)
    out = torch.empty(T * B, H, W, Fc, device="cuda")
    K.head_forward(frames.cuda(), slots, w.detach().cuda(), b.detach().cuda(), out)
    assert rel(from_nhwc(out), ref.detach().reshape(T * B, Fc, H, W)) < TOL
    dw, db = torch.empty(Fc, 3, 3, 3, device="cuda"), torch.empty(Fc, device="cuda")
    K.head_wgrad(frames.cuda(), slots, to_nhwc(dout.reshape(T * B, Fc, H, W)), out, dw, db, ws_tensor(K))
    assert rel(dw, w.grad) < TOL
    assert rel(db, b.grad) < TOL
    # bf16-stored head features: same values rounded once; the ReLU mask read from the bf16 tensor is the same mask
    out16 = torch.empty(T * B, H, W, Fc, device="cuda", dtype=torch.bfloat16)
    K.head_forward(frames.cuda(), slots, w.detach().cuda(), b.detach().cuda(), out16)
    assert torch.equal(out16, out.to(torch.bfloat16))
    dw2, db2 = torch.empty_like(dw), torch.empty_like(db)
    K.head_wgrad(frames.cuda(), slots, to_nhwc(dout.reshape(T * B, Fc, H, W)), out16, dw2, db2, ws_tensor(K))
    assert rel(dw2, w.grad) < TOL and rel(db2, b.grad) < TOL
    # gradient given as the sum of two tensors (skip path fused into the kernel)
    d1 = to_nhwc((0.25 * dout).reshape(T * B, Fc, H, W))
    d2 = to_nhwc((0.75 * dout).reshape(T * B, Fc, H, W), Fc + 4)
    K.head_wgrad(frames.cuda(), slots, d1, out16, dw2, db2, ws_tensor(K), dout2=d2)
    assert rel(dw2, w.grad) < TOL and rel(db2, b.grad) < TOL


@pytest.mark.parametrize("C,N,H,W", [(32, 2, 9, 14), (64, 1, 7, 33), (16, 3, 5, 5)])
def test_depthwise_forward_flip_wgrad(K, C, N, H, W):
    x = rnd(N, C, H, W).requires_grad_()
    w = rnd(C, 1, 3, 3).requires_grad_()
    dy = rnd(N, C, H, W, seed=6)
    y = F.conv2d(x, w, None, padding=1, groups=C)
    y.backward(dy)
    out = torch.empty(N, H, W, C, device="cuda")
    K.dwconv_forward(to_nhwc(x.detach()), w.detach().cuda(), out)
    assert rel(from_nhwc(out), y.detach()) < TOL
    dx = torch.empty(N, H, W, C, device="cuda")
    K.dwconv_forward(to_nhwc(dy), w.detach().cuda(), dx, flip=True)
    assert rel(from_nhwc(dx), x.grad) < TOL
    dw = torch.empty(C, 1, 3, 3, device="cuda")
    K.dwconv_wgrad(to_nhwc(x.detach()), to_nhwc(dy), dw, ws_tensor(K))
    assert rel(dw, w.grad) < TOL


@pytest.mark.parametrize("training", [True, False])
@pytest.mark.parametrize("C,B,G,H,W", [(32, 2, 3, 9, 14), (64, 1, 5, 6, 33)])
def test_batchnorm_relu_groups(K, training, C, B, G, H, W):
    """G groups of B images = G separate BatchNorm2d calls sharing weights and running stats."""
    N = B * G
    x = (rnd(N, C, H, W) * 1.5 + 0.3).requires_grad_()
    gamma = (1 + 0.2 * rnd(C)).requires_grad_()
    beta = (0.1 * rnd(C, seed=2)).requires_grad_()
    res = rnd(N, C, H, W, seed=8)
    rm, rv = 0.1 * rnd(C, seed=3), 1 + 0.3 * rnd(C, seed=4)
    order = list(range(1, G)) + [0]            # running stats update order (slot != time order)
    rm_ref, rv_ref = rm.clone(), rv.clone()
    ys = [None] * G
    for g in order:
        ys[g] = F.relu(F.batch_norm(x[g * B:(g + 1) * B], rm_ref, rv_ref, gamma, beta, training, 0.1, 1e-5))
    y = torch.cat(ys, 0)
    dy = rnd(N, C, H, W, seed=12)
    y.backward(dy)

    xb = to_nhwc(x.detach())
    mean, invstd = torch.empty(G, C, device="cuda"), torch.empty(G, C, device="cuda")
    rmc, rvc = rm.cuda(), rv.cuda()
    ws = ws_tensor(K)
    if training:
        K.bn_stats(xb, B, order, mean, invstd, rmc, rvc, ws)
        assert rel(rmc, rm_ref) < TOL and rel(rvc, rv_ref) < TOL
    else:
        K.bn_eval_stats(rmc, rvc, G, mean, invstd)
    split = B                                   # first group -> slice of a wide buffer, rest -> plain buffer
    outA = torch.zeros(B, H, W, 3 * C, device="cuda")
    outB = torch.empty(N - B, H, W, C, device="cuda")
    K.bn_apply_relu(xb, B, mean, invstd, gamma.detach().cuda(), beta.detach().cuda(), to_nhwc(res),
                    K.Sl(outA, C, C), split, K.Sl(outB))
    want = y.detach() + res
    assert rel(from_nhwc(outA, C, C), want[:B]) < TOL
    assert rel(from_nhwc(outB), want[B:]) < TOL
    dx = torch.empty(N, H, W, C, device="cuda")
    dg, db = torch.empty(C, device="cuda"), torch.empty(C, device="cuda")
    K.bn_relu_backward(to_nhwc(dy), xb, B, mean, invstd, gamma.detach().cuda(), beta.detach().cuda(), training, dx,
                       dg, db, ws)
    assert rel(from_nhwc(dx), x.grad) < 5e-5
    assert rel(dg, gamma.grad) < 5e-5 and rel(db, beta.grad) < 5e-5


# ----------------------------------------------------------------------------- motion
@pytest.mark.parametrize("C,B,R,H,W", [(32, 2, 2, 9, 14), (64, 1, 1, 11, 37), (16, 1, 4, 20, 6)])
def test_correlation_forward_backward(K, C, B, R, H, W):
    """R reference-frame groups of B images share B centre images."""
    N = B * R
    x1 = rnd(N, C, H, W).requires_grad_()
    x2 = rnd(B, C, H, W, seed=3).requires_grad_()
    out = torch.cat([sr_oracle.correlation(x1[r * B:(r + 1) * B], x2) for r in range(R)], 0)
    dy = rnd(N, 81, H, W, seed=5)
    out.backward(dy)
    x1b = to_nhwc(x1.detach())
    al = to_nhwc(x2.detach(), 3 * C, C)
    corr = torch.full((N, H, W, 96), 3.0, device="cuda")
    K.correlation_forward(K.Sl(x1b), K.Sl(al, C, C), corr)
    assert rel(from_nhwc(corr, 81), out.detach()) < TOL
    assert corr[..., 81:].abs().max().item() == 0
    dcorr = to_nhwc(dy, 96)
    dx1 = to_nhwc(rnd(N, C, H, W, seed=7))
    base1 = from_nhwc(dx1)
    K.correlation_backward(1, dcorr, K.Sl(al, C, C), K.Sl(dx1), True)
    assert rel(from_nhwc(dx1) - base1, x1.grad) < TOL
    dx2 = torch.zeros(B, H, W, C, device="cuda")
    for r in range(R):
        K.correlation_backward(2, dcorr[r * B:(r + 1) * B], K.Sl(x1b).images(r * B, (r + 1) * B), K.Sl(dx2), True)
    assert rel(from_nhwc(dx2), x2.grad) < TOL
    # the R reference frames in one call (groups): same sums, one pass over the centre-frame gradient
    dx2m = torch.full((B, H, W, C), 2.0, device="cuda")
    K.correlation_backward(2, dcorr, K.Sl(x1b), K.Sl(dx2m), True, groups=R)
    assert rel(from_nhwc(dx2m) - 2.0, x2.grad) < TOL


@pytest.mark.parametrize("gather", [True, False])
@pytest.mark.parametrize("C,N,H,W,mag", [(32, 2, 9, 14, 1.5), (64, 1, 11, 37, 4.0), (16, 1, 6, 7, 30.0), (64, 2, 19, 45, 0.8),
                                         (128, 1, 9, 33, 6.0)])
def test_warp_forward_backward(K, C, N, H, W, mag, gather):
    """gather=True: the atomics-free two-pass backward (flows up to 30 px exercise its scatter fallback and the image border);
    gather=False: the scatter form."""
    feat = rnd(N, C, H, W).requires_grad_()
    flow = (rnd(N, 2, H, W, seed=2) * mag).requires_grad_()
    out = sr_oracle.warp(feat, flow)
    dy = rnd(N, C, H, W, seed=4)
    out.backward(dy)
    fb = to_nhwc(feat.detach())
    flb = to_nhwc(flow.detach(), 4)
    al = torch.zeros(N, H, W, 3 * C, device="cuda")
    K.warp_forward(K.Sl(fb), flb, K.Sl(al, C, 2 * C))
    assert rel(from_nhwc(al, C, 2 * C), out.detach()) < 5e-5
    dal = to_nhwc(dy, 3 * C, 2 * C)
    dfeat = torch.zeros(N, H, W, C, device="cuda")
    dflow = torch.full((N, H, W, 4), 9.0, device="cuda")
    dfeat = to_nhwc(rnd(N, C, H, W, seed=8))              # the gradient is ADDED to what is there
    base = from_nhwc(dfeat)
    K.warp_backward(K.Sl(dal, C, 2 * C), K.Sl(fb), flb, K.Sl(dfeat), dflow, gather=gather)
    assert rel(from_nhwc(dfeat) - base, feat.grad) < 5e-5
    assert rel(from_nhwc(dflow, 2), flow.grad) < 2e-4
    assert dflow[..., 2:].abs().max().item() == 0
    if gather:
        # overwrite mode: dfeat is written (whatever it held), far sources scattered behind the gather pass
        dfeat2 = torch.full((N, H, W, C), float("nan"), device="cuda")
        dflow2 = torch.empty(N, H, W, 4, device="cuda")
        K.warp_backward(K.Sl(dal, C, 2 * C), K.Sl(fb), flb, K.Sl(dfeat2), dflow2, overwrite=True)
        assert rel(from_nhwc(dfeat2), feat.grad) < 5e-5 and torch.equal(dflow2, dflow)
        # ... and as bf16 (far sources then add with a compare-and-swap loop per word: rounded after every add)
        dfeat3 = torch.full((N, H, W, C), float("nan"), device="cuda", dtype=torch.bfloat16)
        K.warp_backward(K.Sl(dal, C, 2 * C), K.Sl(fb), flb, K.Sl(dfeat3), dflow2, overwrite=True)
        assert rel(from_nhwc(dfeat3.float()), feat.grad) < (8e-3 if mag < 4 else 3e-2) and torch.equal(dflow2, dflow)
    else:
        with pytest.raises(RuntimeError, match="overwrite mode needs the gather form"):
            K.warp_backward(K.Sl(dal, C, 2 * C), K.Sl(fb), flb, K.Sl(dfeat), dflow, gather=False, overwrite=True)


@pytest.mark.parametrize("dfeat_bf16", [False, True])
def test_warp_backward_with_a_fifth_of_the_sources_far_away(K, dfeat_bf16):
    """Overwrite-mode warp backward on a motion field where 20 % of the pixels move 5 - 8 px (beyond the gather pass's 9 x 9
    window): those sources go through warp_bwd_far_kernel (a wave per far source: corner x 16-byte channel piece per lane),
    here thousands of them in every direction, several per wave, onto destinations that also receive near sources; fp32 and
    bf16 gradient tensors, against autograd through the oracle's grid_sample."""
    C, N, H, W = 64, 2, 40, 72
    g = torch.Generator().manual_seed(5)
    feat = rnd(N, C, H, W).requires_grad_()
    far = torch.rand(N, 1, H, W, generator=g) < 0.2
    ang = torch.rand(N, 1, H, W, generator=g) * 6.2831853
    r = 5.0 + 3.0 * torch.rand(N, 1, H, W, generator=g)
    fl = 0.5 * rnd(N, 2, H, W, seed=2) + torch.where(far, r, torch.zeros(())) * torch.cat([torch.cos(ang), torch.sin(ang)], 1)
    flow = fl.requires_grad_()
    dy = rnd(N, C, H, W, seed=4)
    sr_oracle.warp(feat, flow).backward(dy)
    assert 0.15 < far.float().mean().item() < 0.25
    fb, flb, dal = to_nhwc(feat.detach()), to_nhwc(flow.detach(), 4), to_nhwc(dy)
    dfeat = torch.full((N, H, W, C), float("nan"), device="cuda", dtype=torch.bfloat16 if dfeat_bf16 else torch.float32)
    dflow = torch.empty(N, H, W, 4, device="cuda")
    K.warp_backward(K.Sl(dal), K.Sl(fb), flb, K.Sl(dfeat), dflow, overwrite=True)
    # bf16: a destination's addends are rounded one by one (compare-and-swap per word), a few of them per pixel here
    assert rel(from_nhwc(dfeat.float()), feat.grad) < (2e-2 if dfeat_bf16 else 5e-5)
    assert rel(from_nhwc(dflow, 2), flow.grad) < 2e-4
    # the accumulate mode (all sources through the src pass) gives the same sums
    dfeat2 = torch.zeros(N, H, W, C, device="cuda")
    K.warp_backward(K.Sl(dal), K.Sl(fb), flb, K.Sl(dfeat2), dflow)
    assert rel(from_nhwc(dfeat2), feat.grad) < 5e-5


@pytest.mark.parametrize("overwrite", [False, True])
def test_warp_backward_contracting_flow_overflows_the_hit_list(K, overwrite):
    """A flow that sends a 7 x 7 neighbourhood to (nearly) one point gives destination pixels more contributing sources than the
    gather pass lists (12): the rest is added by the pixel's own thread behind the listed ones, in both modes."""
    C, N, H, W = 64, 1, 16, 40
    feat = rnd(N, C, H, W).requires_grad_()
    ys, xs = torch.meshgrid(torch.arange(H, dtype=torch.float32), torch.arange(W, dtype=torch.float32), indexing="ij")
    cy, cx = 8.3, 20.6
    near = ((ys - cy).abs() <= 3.5) & ((xs - cx).abs() <= 3.5)
    fl = torch.stack([torch.where(near, 0.97 * (cx - xs), torch.zeros(())), torch.where(near, 0.97 * (cy - ys), torch.zeros(()))])
    flow = (fl[None] + 0.05 * rnd(N, 2, H, W, seed=2)).requires_grad_()
    sr_oracle.warp(feat, flow).backward(rnd(N, C, H, W, seed=4))
    dal = to_nhwc(rnd(N, C, H, W, seed=4))
    base = rnd(N, C, H, W, seed=8)
    dfeat = to_nhwc(base)
    dflow = torch.empty(N, H, W, 4, device="cuda")
    K.warp_backward(K.Sl(dal), K.Sl(to_nhwc(feat.detach())), to_nhwc(flow.detach(), 4), K.Sl(dfeat), dflow, overwrite=overwrite)
    got = from_nhwc(dfeat) - (0 if overwrite else base)
    assert rel(got, feat.grad) < 5e-5
    assert rel(from_nhwc(dflow, 2), flow.grad) < 2e-4
    if overwrite:                                             # bf16 dfeat: the overflow adds round after every term
        d16 = torch.zeros(N, H, W, C, device="cuda", dtype=torch.bfloat16)
        K.warp_backward(K.Sl(dal), K.Sl(to_nhwc(feat.detach())), to_nhwc(flow.detach(), 4), K.Sl(d16), dflow, overwrite=True)
        assert rel(from_nhwc(d16.float()), feat.grad) < 3e-2


# ----------------------------------------------------------------------------- aggregation
@pytest.mark.parametrize("C,T,B,H,W", [(32, 3, 2, 9, 14), (64, 3, 1, 40, 37), (16, 5, 1, 6, 7)])
def test_softmax_weighted_sum(K, C, T, B, H, W):
    al = rnd(B, T * C, H, W).requires_grad_()
    lg = (rnd(B, T, H, W, seed=3) * 3).requires_grad_()
    attn = torch.softmax(lg, 1)
    wt = (al.view(B, T, C, H, W) * attn[:, :, None]).sum(1)
    dw = rnd(B, C, H, W, seed=6)
    dgap = rnd(B, C, seed=8) * 0.1
    (wt * (dw + dgap[:, :, None, None])).sum().backward()
    Tp = K.pad4(T)
    alb, lgb = to_nhwc(al.detach()), to_nhwc(lg.detach(), Tp)
    attn_b, wt_b = torch.empty(B, H, W, Tp, device="cuda"), torch.empty(B, H, W, C, device="cuda")
    nblk = K.tsum_blocks(H, W)
    gp = torch.empty(B, nblk, C, device="cuda")
    K.tsum_forward(alb, lgb, T, C, attn_b, wt_b, gp)
    assert rel(from_nhwc(wt_b), wt.detach()) < TOL
    assert rel(from_nhwc(attn_b, T), attn.detach()) < TOL
    assert rel(gp.sum(1).cpu(), wt.detach().sum((2, 3))) < TOL
    dal, dlg = torch.empty(B, H, W, T * C, device="cuda"), torch.full((B, H, W, Tp), 5.0, device="cuda")
    K.tsum_backward(to_nhwc(dw), dgap.cuda(), alb, attn_b, T, C, dal, dlg)
    assert rel(from_nhwc(dal), al.grad) < TOL
    assert rel(from_nhwc(dlg, T), lg.grad) < 5e-5
    assert dlg[..., T:].abs().max().item() == 0


@pytest.mark.parametrize("C,B,H,W", [(32, 2, 9, 14), (64, 1, 35, 37), (16, 1, 6, 20)])
def test_cbam_forward_backward(K, C, B, H, W):
    R = C // 16
    x = rnd(B, C, H, W).requires_grad_()
    P = {"temporal_aggregator.refine.channel_attention.fc.0.weight": rnd(R, C, seed=1).requires_grad_(),
         "temporal_aggregator.refine.channel_attention.fc.2.weight": rnd(C, R, seed=2).requires_grad_(),
         "temporal_aggregator.refine.spatial_attention.conv.weight": (rnd(1, 2, 7, 7, seed=3) * 0.3).requires_grad_()}
    y = sr_oracle.cbam(P, x)
    dy = rnd(B, C, H, W, seed=7)
    y.backward(dy)
    w1, w2, w7 = [v.detach().cuda() for v in P.values()]
    xb = to_nhwc(x.detach())
    # the pooled sums normally come from the weighted-sum kernel: emulate one block per image
    nblk = 1
    gp = x.detach().sum((2, 3)).reshape(B, 1, C).cuda().contiguous()
    gap, hid, ca = torch.empty(B, C, device="cuda"), torch.empty(B, R, device="cuda"), torch.empty(B, C, device="cuda")
    K.cbam_channel(gp, nblk, C, R, B, H * W, w1, w2, gap, hid, ca)
    sm, amax, sa = torch.empty(B, H, W, 2, device="cuda"), torch.empty(B, H, W, dtype=torch.int32, device="cuda"), \
        torch.empty(B, H, W, device="cuda")
    K.cbam_pool(xb, ca, sm, amax)
    cat = torch.zeros(B, H, W, C + 32, device="cuda")
    K.cbam_spatial_apply(xb, ca, sm, w7, sa, K.Sl(cat, C, 0))
    assert rel(from_nhwc(cat, C), y.detach()) < TOL
    dcat = to_nhwc(dy, C + 32)
    dpre = torch.empty(B, H, W, device="cuda")
    K.cbam_bwd_spatial_pre(K.Sl(dcat, C, 0), xb, ca, sa, dpre)
    dsm, dw7 = torch.empty(B, H, W, 2, device="cuda"), torch.empty(1, 2, 7, 7, device="cuda")
    ws = ws_tensor(K)
    K.cbam_bwd_spatial_conv(dpre, sm, w7, dsm, dw7, ws)
    nb2 = K.tsum_blocks(H, W)
    dx, dcap = torch.empty(B, H, W, C, device="cuda"), torch.empty(B, nb2, C, device="cuda")
    K.cbam_bwd_scale(K.Sl(dcat, C, 0), xb, ca, sa, dsm, amax, dx, dcap)
    dw1, dw2, dgp = torch.empty(R, C, device="cuda"), torch.empty(C, R, device="cuda"), torch.empty(B, C, device="cuda")
    K.cbam_bwd_channel(dcap, nb2, C, R, B, H * W, w1, w2, gap, hid, ca, dw1, dw2, dgp)
    got_dx = from_nhwc(dx) + dgp.cpu()[:, :, None, None]
    g1, g2, g7 = [v.grad for v in P.values()]
    assert rel(got_dx, x.grad) < 5e-5
    assert rel(dw7, g7) < 5e-5
    assert rel(dw1, g1) < 5e-5 and rel(dw2, g2) < 5e-5


# ----------------------------------------------------------------------------- upsampler tail
@pytest.mark.parametrize("s,cin,B,T,H,W", [(2, 64, 2, 3, 19, 45), (3, 32, 1, 3, 8, 32), (4, 64, 1, 5, 17, 33), (2, 32, 3, 3, 9, 14)])
def test_upsampler_tail_fused_in_the_conv_epilogue(K, s, cin, B, T, H, W):
    """nvq_upsampler_tail_forward (bf16 mode): the 3x3 conv to 3 s^2 channels with the pixel shuffle done as an LDS transpose
    in its epilogue, the bicubic skip and the clamp - against the reference ops (PixelShuffleUpsampler
    efficient_layers.py:94-106, super_resolution.py:378-382) and, bit for bit, against the two-kernel path
    (nvq_conv_forward + nvq_shuffle_bicubic_clamp); interior and border tiles, all three scales."""
    U = 3 * s * s
    frames = rnd(B, T, 3, H, W, seed=1).abs()
    x = bf(rnd(B, cin, H, W, seed=2))
    w, b = rnd(U, cin, 3, 3, scale=0.08, seed=3), rnd(U, seed=4) * 0.1
    xb = to_nhwc_bf16(x)
    wp = K.conv_pack(w.cuda(), False, cin, math=K.MATH_BF16)
    out = torch.empty(B, 3, H * s, W * s, device="cuda")
    pm = torch.empty(B, 3, H * s, W * s, dtype=torch.uint8, device="cuda")
    K.upsampler_tail_forward(K.Sl(xb), wp, b.cuda(), frames.cuda(), T // 2, s, out, pm)
    Up = K.pad4(U)
    u = torch.zeros(B, H, W, Up, device="cuda")
    K.conv_forward(K.Sl(xb), wp, b.cuda(), K.Sl(u, U), 3, cout_store=Up, math=K.MATH_BF16)
    out2 = torch.empty_like(out)
    pm2 = torch.empty_like(pm)
    K.shuffle_bicubic_clamp(u, frames.cuda(), T // 2, s, out2, pm2)
    assert torch.equal(out, out2) and torch.equal(pm, pm2)
    pre = sr_oracle.bicubic_up(frames[:, T // 2], s) + F.pixel_shuffle(F.conv2d(x, bf(w), b, padding=1), s)
    assert (out.cpu() - pre.clamp(0, 1)).abs().max().item() < 2e-5
    frac = ((out == 0) | (out == 1)).float().mean().item()
    assert 0.02 < frac < 0.9
    with pytest.raises(RuntimeError, match="upsampler_tail"):   # the exact-fp32 mode has no fused form
        K.upsampler_tail_forward(K.Sl(to_nhwc(x)), wp, b.cuda(), frames.cuda(), T // 2, s, out, pm)


@pytest.mark.parametrize("s,B,T,H,W", [(2, 2, 3, 9, 14), (3, 1, 3, 7, 11), (4, 1, 5, 6, 10)])
def test_shuffle_bicubic_clamp(K, s, B, T, H, W):
    frames = rnd(B, T, 3, H, W).abs()
    u = (rnd(B, 3 * s * s, H, W, seed=2) * 0.6).requires_grad_()
    pre = sr_oracle.bicubic_up(frames[:, T // 2], s) + F.pixel_shuffle(u, s)
    out = torch.clamp(pre, 0, 1)
    dy = rnd(B, 3, H * s, W * s, seed=3)
    out.backward(dy)
    Up = K.pad4(3 * s * s)
    got = torch.empty(B, 3, H * s, W * s, device="cuda")
    pm = torch.empty(B, 3, H * s, W * s, dtype=torch.uint8, device="cuda")
    K.shuffle_bicubic_clamp(to_nhwc(u.detach(), Up), frames.cuda(), T // 2, s, got, pm)
    assert (got.cpu() - out.detach()).abs().max().item() < 2e-6     # absolute: values live in [0,1]
    frac = ((out == 0) | (out == 1)).float().mean().item()
    assert 0.02 < frac < 0.9                                           # both clamp rails exercised
    du = torch.full((B, H, W, Up), 4.0, device="cuda")
    K.shuffle_clamp_backward(dy.cuda(), pm, s, du)
    # compare where the pre-clamp value is not within rounding of a rail
    safe = ((pre.detach() - 0).abs() > 1e-5) & ((pre.detach() - 1).abs() > 1e-5)
    safe_u = F.pixel_unshuffle(safe.float(), s) > 0
    diff = (from_nhwc(du, 3 * s * s) - u.grad).abs() * safe_u
    assert diff.max().item() < 1e-6
    if Up > 3 * s * s:
        assert du[..., 3 * s * s:].abs().max().item() == 0
    # the same through the any-shape kernel (a wider du row is not the vectorised kernels' layout)
    du_wide = torch.full((B, H, W, Up + 4), 4.0, device="cuda")
    K.shuffle_clamp_backward(dy.cuda(), pm, s, du_wide)
    assert torch.equal(du_wide[..., :Up], du) and du_wide[..., Up:].abs().max().item() == 0


# ----------------------------------------------------------------------------- helpers / EWC
def test_axpy_slice_and_colsum(K):
    N, H, W = 2, 7, 9
    a, b, m = rnd(N, 32, H, W), rnd(N, 32, H, W, seed=2), rnd(N, 32, H, W, seed=3)
    dst, src, msk = to_nhwc(a, 64, 16), to_nhwc(b, 96, 32), to_nhwc(m, 32)
    K.axpy_slice(K.Sl(dst, 32, 16), K.Sl(src, 32, 32), alpha=0.5, accumulate=True, mask=K.Sl(msk))
    assert rel(from_nhwc(dst, 32, 16), a + 0.5 * b * (m > 0)) < 1e-6
    out = torch.ones(32, device="cuda")
    K.colsum(K.Sl(src, 32, 32), out, ws_tensor(K), alpha=2.0, accumulate=True)
    assert rel(out, 1 + 2 * b.sum((0, 2, 3))) < TOL


def test_ewc_flat_kernels(K):
    n = 1_000_003
    th, st, fi = rnd(n), rnd(n, seed=2), rnd(n, seed=3).abs()
    lam = 5000.0
    want = lam / 2 * (fi.double() * (th.double() - st.double()) ** 2).sum()
    out = torch.zeros(1, device="cuda")
    ws = ws_tensor(K)
    K.ewc_penalty(th.cuda(), st.cuda(), fi.cuda(), lam, out, ws)
    assert abs(out.item() - want.item()) < 1e-5 * abs(want.item())
    scale = torch.tensor([0.25], device="cuda")
    g = torch.ones(n, device="cuda")
    K.ewc_penalty_grad(th.cuda(), st.cuda(), fi.cuda(), lam, scale, g, True)
    assert rel(g - 1, 0.25 * lam * fi * (th - st)) < 1e-5
    K.ewc_penalty_grad(th.cuda(), st.cuda(), fi.cuda(), lam, None, g, False)
    assert rel(g, lam * fi * (th - st)) < 1e-6
    acc = fi.cuda().clone()
    K.fisher_accumulate(th.cuda(), acc)
    assert rel(acc, fi + th * th) < 1e-6


# ----------------------------------------------------------------------------- NVQ_MATH_BF16 convolutions
# Reference = fp32 convolution of bf16-ROUNDED operands: bf16 x bf16 products are exact in fp32, so the
# kernels may differ from it by summation order only (same 2e-5 bound as the fp32 kernels).
def bf(x):
    return x.bfloat16().float()


@pytest.mark.parametrize("cin,cs,cout,k,N,H,W", CONV_CASES)
def test_conv_bf16_forward_bias_relu(K, cin, cs, cout, k, N, H, W):
    x, w, b = rnd(N, cin, H, W), rnd(cout, cin, k, k, scale=0.2), rnd(cout)
    ref = F.relu(F.conv2d(bf(x), bf(w), b, padding=k // 2))
    xin = to_nhwc(x, cs)
    wp = K.conv_pack(w.cuda(), False, cs, math=K.MATH_BF16)
    cst = K.pad4(cout)
    out = torch.full((N, H, W, cst + 4), 7.0, device="cuda")
    K.conv_forward(K.Sl(xin), wp, b.cuda(), K.Sl(out, cout, 0), k, relu=True, cout_store=cst, math=K.MATH_BF16)
    assert rel(from_nhwc(out, cout), ref) < TOL
    assert (out[..., cst:] == 7.0).all()
    # and the rounding itself costs at most ~2^-8 relative against the unrounded fp32 convolution
    assert rel(from_nhwc(out, cout), F.relu(F.conv2d(x, w, b, padding=k // 2))) < 2e-2


def test_conv_bf16_dense_layer_and_lff_epilogues(K):
    N, H, W, Fc, CAT = 1, 10, 33, 64, 224
    cat, w, b = rnd(N, CAT, H, W), rnd(Fc, CAT, 1, 1, scale=0.1), rnd(Fc)
    ref = F.conv2d(bf(cat), bf(w), b) * 0.2 + cat[:, :Fc]
    catb = to_nhwc(cat)
    out = torch.zeros(N, H, W, CAT, device="cuda")
    K.conv_forward(K.Sl(catb), K.conv_pack(w.cuda(), False, CAT, math=K.MATH_BF16), b.cuda(), K.Sl(out, Fc, 32), 1,
                   alpha=0.2, res=K.Sl(catb, Fc, 0), math=K.MATH_BF16)
    assert rel(from_nhwc(out, Fc, 32), ref) < TOL
    x, w3, b3 = rnd(N, 96, H, W), rnd(32, 96, 3, 3, scale=0.1), rnd(32)
    ref3 = F.relu(F.conv2d(bf(x), bf(w3), b3, padding=1))
    cat2 = to_nhwc(x, 224)
    K.conv_forward(K.Sl(cat2, 96, 0), K.conv_pack(w3.cuda(), False, 96, math=K.MATH_BF16), b3.cuda(),
                   K.Sl(cat2, 32, 96), 3, relu=True, math=K.MATH_BF16)
    assert rel(from_nhwc(cat2, 32, 96), ref3) < TOL


@pytest.mark.parametrize("cin,cout,k", [(96, 32, 3), (224, 64, 1), (81, 128, 3), (64, 2, 3), (64, 12, 3)])
def test_conv_bf16_input_gradient(K, cin, cout, k):
    N, H, W = 1, 9, 35
    x = rnd(N, cin, H, W).requires_grad_()
    w = rnd(cout, cin, k, k, scale=0.2)
    dy = rnd(N, cout, H, W, seed=3)
    F.conv2d(x, bf(w), None, padding=k // 2).backward(bf(dy))
    old = rnd(N, cin, H, W, seed=9)
    cs, ld = K.pad4(cout), K.pad4(cin)
    out = to_nhwc(old, ld)
    K.conv_forward(K.Sl(to_nhwc(dy, cs)), K.conv_pack(w.cuda(), True, cs, cin, math=K.MATH_BF16), None,
                   K.Sl(out, cin, 0), k, accumulate=True, cout_store=ld, math=K.MATH_BF16)
    assert rel(from_nhwc(out, cin), x.grad + old) < TOL


@pytest.mark.parametrize("cin,cin_store,cout,k,N,H,W", [
    (64, 64, 32, 3, 2, 16, 40), (81, 96, 128, 3, 1, 9, 33), (224, 224, 64, 1, 1, 11, 37),
    (32, 32, 2, 3, 1, 8, 31), (64, 64, 12, 3, 1, 17, 33), (192, 192, 64, 3, 1, 8, 64), (64, 64, 3, 3, 2, 5, 7),
])
def test_conv_bf16_weight_gradient(K, cin, cin_store, cout, k, N, H, W):
    x = rnd(N, cin, H, W)
    w = rnd(cout, cin, k, k).requires_grad_()
    dy = rnd(N, cout, H, W, seed=3)
    F.conv2d(bf(x), w, None, padding=k // 2).backward(bf(dy))
    dw = torch.zeros(cout, cin, k, k, device="cuda")
    db = torch.zeros(cout, device="cuda")
    K.conv_wgrad(K.Sl(to_nhwc(x, cin_store)), cin, K.Sl(to_nhwc(dy, K.pad4(cout)), cout, 0), dw, db, ws_tensor(K), k,
                 math=K.MATH_BF16)
    assert rel(dw, w.grad) < TOL
    assert rel(db, dy.sum((0, 2, 3))) < TOL      # bias gradient is summed in fp32 from the unrounded dy


# ----------------------------------------------------------------------------- bf16-STORED activation tensors
def to_nhwc_bf16(x, ld=None, coff=0):
    return to_nhwc(x, ld, coff).bfloat16()


@pytest.mark.parametrize("cin,cout,k,N,H,W", [(64, 32, 3, 2, 13, 37), (224, 64, 1, 1, 11, 40), (192, 32, 3, 1, 8, 35),
                                              (128, 64, 3, 1, 9, 33), (32, 2, 3, 1, 8, 32),
                                              (96, 32, 3, 2, 41, 37), (192, 32, 3, 1, 16, 70), (64, 24, 3, 1, 33, 32)])
def test_conv_bf16_stored_input_and_output(K, cin, cout, k, N, H, W):
    """Input read as stored bf16 (no rounding in the kernel), output written as bf16 (rounded once) or fp32."""
    x, w, b = bf(rnd(N, cin, H, W)), rnd(cout, cin, k, k, scale=0.2), rnd(cout)
    ref = F.relu(F.conv2d(x, bf(w), b, padding=k // 2))
    wp = K.conv_pack(w.cuda(), False, cin, math=K.MATH_BF16)
    xin = to_nhwc_bf16(x)
    cst = K.pad4(cout)
    out32 = torch.zeros(N, H, W, cst, device="cuda")
    K.conv_forward(K.Sl(xin), wp, b.cuda(), K.Sl(out32, cout, 0), k, relu=True, cout_store=cst, math=K.MATH_BF16)
    assert rel(from_nhwc(out32, cout), ref) < TOL
    if cout % 8 == 0:
        out16 = torch.zeros(N, H, W, cout + 8, device="cuda", dtype=torch.bfloat16)
        K.conv_forward(K.Sl(xin), wp, b.cuda(), K.Sl(out16, cout, 8), k, relu=True, math=K.MATH_BF16)
        assert torch.equal(out16[..., 8:].float().cpu(), to_nhwc(ref).bfloat16().float().cpu()) or \
            rel(out16[..., 8:].float().permute(0, 3, 1, 2), ref) < 5e-3       # one bf16 rounding of the result
        assert out16[..., :8].abs().max().item() == 0


def test_conv_bf16_stored_dense_block_epilogues(K):
    """All tensors bf16: in-place concat write, lff form (scale + residual), mask + second output."""
    N, H, W, Fc, CAT = 1, 10, 33, 64, 224
    cat = bf(rnd(N, CAT, H, W))
    w3, b3 = rnd(32, 96, 3, 3, scale=0.1), rnd(32)
    catb = to_nhwc_bf16(cat)
    K.conv_forward(K.Sl(catb, 96, 0), K.conv_pack(w3.cuda(), False, 96, math=K.MATH_BF16), b3.cuda(), K.Sl(catb, 32, 96), 3,
                   relu=True, math=K.MATH_BF16)
    ref3 = F.relu(F.conv2d(cat[:, :96], bf(w3), b3, padding=1))
    assert rel(catb[..., 96:128].float().permute(0, 3, 1, 2), ref3) < 5e-3
    assert torch.equal(catb[..., :96].float().cpu(), to_nhwc(cat[:, :96]).cpu())
    w1, b1 = rnd(Fc, CAT, 1, 1, scale=0.1), rnd(Fc)
    cat2 = to_nhwc_bf16(cat)
    nxt = torch.zeros(N, H, W, CAT, device="cuda", dtype=torch.bfloat16)
    K.conv_forward(K.Sl(cat2), K.conv_pack(w1.cuda(), False, CAT, math=K.MATH_BF16), b1.cuda(), K.Sl(nxt, Fc, 0), 1,
                   alpha=0.2, res=K.Sl(cat2, Fc, 0), math=K.MATH_BF16)
    ref1 = F.conv2d(cat, bf(w1), b1) * 0.2 + cat[:, :Fc]
    assert rel(nxt[..., :Fc].float().permute(0, 3, 1, 2), ref1) < 5e-3
    # mirror-backward form: bf16 gradient buffer, mask from the bf16 forward buffer at another channel offset
    dcat = to_nhwc_bf16(bf(rnd(N, CAT, H, W, seed=5)))
    wb = rnd(32, 128, 3, 3, scale=0.1)
    K.conv_forward(K.Sl(dcat, 128, 0), K.conv_pack(wb.cuda(), False, 128, math=K.MATH_BF16), None, K.Sl(dcat, 32, 128), 3,
                   mask=K.Sl(cat2, 32, 96), mask_c0=0, mask_c1=32, math=K.MATH_BF16)
    refb = F.conv2d(dcat[..., :128].float().permute(0, 3, 1, 2).cpu(), bf(wb), None, padding=1) * (cat[:, 96:128] > 0)
    assert rel(dcat[..., 128:160].float().permute(0, 3, 1, 2), refb) < 5e-3


@pytest.mark.parametrize("xb,yb", [(True, True), (True, False), (False, True)])
@pytest.mark.parametrize("cin,cout,k,N,H,W", [(64, 32, 3, 2, 16, 40), (224, 64, 1, 1, 11, 37), (192, 64, 3, 1, 8, 64),
                                              (96, 72, 1, 2, 9, 33), (64, 128, 1, 1, 17, 32),
                                              # an odd number of 32-channel units: 64-channel chunks + a 32-channel tail with
                                              # its own pixel-split count in one launch (wgrad_bf16_mixed_kernel)
                                              (96, 32, 3, 2, 19, 70), (160, 32, 3, 3, 33, 41), (96, 128, 3, 2, 24, 33),
                                              (224, 64, 3, 1, 17, 50)])
def test_conv_bf16_stored_weight_gradient(K, xb, yb, cin, cout, k, N, H, W):
    x, dy = bf(rnd(N, cin, H, W)), bf(rnd(N, cout, H, W, seed=3))
    w = rnd(cout, cin, k, k).requires_grad_()
    F.conv2d(x, w, None, padding=k // 2).backward(dy)
    dw = torch.zeros(cout, cin, k, k, device="cuda")
    db = torch.zeros(cout, device="cuda")
    xs = to_nhwc_bf16(x) if xb else to_nhwc(x)
    ds = to_nhwc_bf16(dy, cout + 8, 8) if yb else to_nhwc(dy, cout + 8, 8)
    K.conv_wgrad(K.Sl(xs), cin, K.Sl(ds, cout, 8), dw, db, ws_tensor(K), k, math=K.MATH_BF16)
    assert rel(dw, w.grad) < TOL
    assert rel(db, dy.sum((0, 2, 3))) < TOL


def test_conv_wgrad_in_two_halves_with_batched_reduces(K):
    """nvq_conv_wgrad_partial + nvq_wgrad_reduce_batch: 19 weight gradients of different shapes (1x1 / 3x3, bf16 and fp32 tensors,
    with and without bias, alpha / accumulate) left as partial sums in workspaces of their own and finished by two batched
    launches (16 + 3 jobs): bit-identical to nvq_conv_wgrad on the same operands."""
    shapes = [(64, 32, 3), (96, 32, 3), (224, 64, 1), (64, 64, 3), (128, 32, 3), (160, 32, 3), (96, 72, 1), (192, 32, 3),
              (96, 128, 3)]
    N, H, W = 2, 14, 40
    jobs, ref, got = [], [], []
    for i in range(19):
        cin, cout, k = shapes[i % len(shapes)]
        bf = i % 3 != 2
        x = rnd(N, cin, H, W, seed=i)
        dy = rnd(N, cout, H, W, seed=100 + i)
        xs = to_nhwc_bf16(x) if bf else to_nhwc(x)
        ds = to_nhwc_bf16(dy, cout + 8, 8) if bf else to_nhwc(dy, cout + 8, 8)
        acc, alpha, bias = i % 4 == 1, (0.2 if i % 5 == 0 else 1.0), i % 6 != 3
        mk = lambda: (torch.full((cout, cin, k, k), 0.5, device="cuda"), torch.full((cout,), -0.25, device="cuda") if bias else None)
        dw0, db0 = mk()
        K.conv_wgrad(K.Sl(xs), cin, K.Sl(ds, cout, 8), dw0, db0, ws_tensor(K), k, alpha=alpha, accumulate=acc, math=K.MATH_BF16)
        dw1, db1 = mk()
        K.conv_wgrad(K.Sl(xs), cin, K.Sl(ds, cout, 8), dw1, db1, torch.empty_like(ws_tensor(K)), k, alpha=alpha, accumulate=acc,
                     math=K.MATH_BF16, defer=jobs)
        ref.append((dw0, db0)); got.append((dw1, db1))
    assert len(jobs) == 19 and all(torch.equal(g[0], torch.full_like(g[0], 0.5)) for g in got)   # nothing reduced yet
    K.wgrad_reduce_batch(jobs)
    assert jobs == []
    for (dw0, db0), (dw1, db1) in zip(ref, got):
        assert torch.equal(dw0, dw1) and (db0 is None or torch.equal(db0, db1))


@pytest.mark.parametrize("Fc,cin,N,H,W", [(64, 96, 2, 16, 64), (64, 96, 1, 21, 45), (64, 128, 2, 9, 33), (64, 160, 3, 17, 70),
                                            (64, 192, 1, 40, 96), (32, 96, 2, 8, 32), (32, 160, 1, 13, 50), (128, 192, 1, 12, 40),
                                            (64, 192, 5, 37, 130)])
def test_conv_bf16_weight_gradient_all_input_channels(K, Fc, cin, N, H, W):
    """nvq_conv_wgrad on a slice-planar bf16 dense-block buffer, 3x3, cout = 32: the all-input-channel kernel (wgrad_m32.hip: x
    units without halo, dy with halo, six MFMA waves + two dy waves, persistent over tiles) against autograd's weight / bias
    gradient of F.conv2d on the same bf16 values, and against the (pixel split, ci chunk) kernel (variant = 1; 2 forces the
    all-input-channel kernel, which the automatic choice takes from 1024 tiles on).  Sizes with
    partial tiles in both directions, one to many tiles per workgroup, leading tensors of 32 / 64 / 128 channels, alpha and
    accumulate."""
    x, dy = bf(rnd(N, cin, H, W)), bf(rnd(N, 32, H, W, seed=3))
    w = rnd(32, cin, 3, 3).requires_grad_()
    F.conv2d(x, w, None, padding=1).backward(dy)
    cat = K.CatBuf("cuda", N, H, W, Fc, (cin - Fc) // 32 + 1, 256, torch.bfloat16, planar=True)
    xn = to_nhwc_bf16(x)
    cat.lead.copy_(xn[..., :Fc])
    for j in range((cin - Fc) // 32):
        cat.slices[j].copy_(xn[..., Fc + 32 * j:Fc + 32 * (j + 1)])
    ds = to_nhwc_bf16(dy, 40, 8)
    ws = ws_tensor(K)
    outs = []
    for variant in (2, 1):
        dw, db = torch.full((32, cin, 3, 3), 7.0, device="cuda"), torch.full((32,), 7.0, device="cuda")
        K.conv_wgrad(cat.inp(cin), cin, K.Sl(ds, 32, 8), dw, db, ws, 3, math=K.MATH_BF16, variant=variant)
        outs.append((dw, db))
    assert rel(outs[0][0], w.grad) < TOL and rel(outs[0][1], dy.sum((0, 2, 3))) < TOL
    assert rel(outs[0][0], outs[1][0]) < TOL and rel(outs[0][1], outs[1][1]) < TOL
    dw, db = outs[0][0].clone(), outs[0][1].clone()
    K.conv_wgrad(cat.inp(cin), cin, K.Sl(ds, 32, 8), dw, db, ws, 3, alpha=0.5, accumulate=True, math=K.MATH_BF16, variant=2)
    assert rel(dw, 1.5 * w.grad) < TOL and rel(db, 1.5 * dy.sum((0, 2, 3))) < TOL


def test_extractor_kernels_with_bf16_stored_tensors(K):
    """Depthwise conv and BatchNorm kernels reading / writing bf16-stored tensors (fp32 arithmetic inside)."""
    C, B, G, H, W = 32, 2, 3, 9, 14
    N = B * G
    x = bf(rnd(N, C, H, W) * 1.5 + 0.3)
    w = rnd(C, 1, 3, 3)
    ref = F.conv2d(x, w, None, padding=1, groups=C)
    out = torch.empty(N, H, W, C, device="cuda", dtype=torch.bfloat16)
    K.dwconv_forward(to_nhwc_bf16(x), w.cuda(), out)
    assert rel(out.float().permute(0, 3, 1, 2), ref) < 5e-3            # one bf16 rounding of the result
    dy = bf(rnd(N, C, H, W, seed=6))
    wg = w.clone().requires_grad_()
    F.conv2d(x, wg, None, padding=1, groups=C).backward(dy)
    dw = torch.empty(C, 1, 3, 3, device="cuda")
    K.dwconv_wgrad(to_nhwc_bf16(x), to_nhwc_bf16(dy), dw, ws_tensor(K))
    assert rel(dw, wg.grad) < TOL
    # BatchNorm (training statistics) on a bf16 tensor, bf16 output, bf16 gradient in/out
    xg = x.clone().requires_grad_()
    gamma, beta = (1 + 0.2 * rnd(C)).requires_grad_(), (0.1 * rnd(C, seed=2)).requires_grad_()
    ys = [F.relu(F.batch_norm(xg[g * B:(g + 1) * B], None, None, gamma, beta, True, 0.1, 1e-5)) for g in range(G)]
    y = torch.cat(ys, 0)
    y.backward(dy)
    xb = to_nhwc_bf16(x)
    mean, invstd = torch.empty(G, C, device="cuda"), torch.empty(G, C, device="cuda")
    ws = ws_tensor(K)
    K.bn_stats(xb, B, list(range(G)), mean, invstd, None, None, ws)
    yb = torch.empty(N, H, W, C, device="cuda", dtype=torch.bfloat16)
    K.bn_apply_relu(xb, B, mean, invstd, gamma.detach().cuda(), beta.detach().cuda(), None, K.Sl(yb), N)
    assert rel(yb.float().permute(0, 3, 1, 2), y.detach()) < 5e-3
    dx = torch.empty(N, H, W, C, device="cuda", dtype=torch.bfloat16)
    dg, db = torch.empty(C, device="cuda"), torch.empty(C, device="cuda")
    K.bn_relu_backward(to_nhwc_bf16(dy), xb, B, mean, invstd, gamma.detach().cuda(), beta.detach().cuda(), True, dx, dg,
                       db, ws)
    assert rel(dx.float().permute(0, 3, 1, 2), xg.grad) < 8e-3
    assert rel(dg, gamma.grad) < 5e-5 and rel(db, beta.grad) < 5e-5


@pytest.mark.parametrize("training", [True, False])
@pytest.mark.parametrize("dy16", [True, False])
@pytest.mark.parametrize("B,G,H,W", [(2, 3, 9, 14), (1, 2, 16, 40), (3, 1, 5, 7)])
def test_pointwise_batchnorm_backward_in_one_pass(K, training, dy16, B, G, H, W):
    """nvq_pw_bn_backward: ReLU + BatchNorm backward, the 1x1 conv's input gradient and its weight gradient from one staged
    tile (dp never stored) against (a) the three-launch path it replaces and (b) autograd of the reference ops
    (efficient_layers.py:49-66); groups whose pixel count is not a multiple of the 128-pixel tile, fp32 / bf16 dy, train / eval."""
    C, N = 64, B * G
    d = bf(rnd(N, C, H, W, seed=1))
    w = bf(rnd(C, C, 1, 1, scale=0.15, seed=2))
    gamma, beta = 1 + 0.2 * rnd(C, seed=3), 0.1 * rnd(C, seed=4)
    dy = bf(rnd(N, C, H, W, seed=5))
    dg, wg, gg, bg = d.clone().requires_grad_(), w.clone().requires_grad_(), gamma.clone().requires_grad_(), beta.clone().requires_grad_()
    p_ref = F.conv2d(dg, wg)
    p16 = bf(p_ref.detach())                                   # what the forward stored: bf16
    rm, rv = 0.1 * rnd(C, seed=6), 1 + 0.3 * rnd(C, seed=7).abs()
    # autograd reference on the bf16-stored conv output (per frame group statistics)
    pl = p16.clone().requires_grad_()
    outs = [F.relu(F.batch_norm(pl[g * B:(g + 1) * B], rm.clone(), rv.clone(), gg, bg, training, 0.1, 1e-5)) for g in range(G)]
    torch.cat(outs, 0).backward(dy)
    dp_ref = pl.grad
    dd_ref = F.conv_transpose2d(bf(dp_ref), w)                  # dd = dp W (dp as the kernels see it: bf16)
    dw_ref = torch.einsum("nohw,nihw->oi", bf(dp_ref), d)
    pb, db_ = to_nhwc_bf16(p16), to_nhwc_bf16(d)
    dyb = to_nhwc_bf16(dy) if dy16 else to_nhwc(dy)
    mean, invstd = torch.empty(G, C, device="cuda"), torch.empty(G, C, device="cuda")
    ws = ws_tensor(K)
    if training:
        K.bn_stats(pb, B, list(range(G)), mean, invstd, None, None, ws)
    else:
        K.bn_eval_stats(rm.cuda(), rv.cuda(), G, mean, invstd)
    dd = torch.empty(N, H, W, C, device="cuda", dtype=torch.bfloat16)
    dga, dbe, dwt = torch.empty(C, device="cuda"), torch.empty(C, device="cuda"), torch.empty(C, C, 1, 1, device="cuda")
    K.pw_bn_backward(dyb, pb, db_, B, mean, invstd, gamma.cuda(), beta.cuda(), training, w.cuda(), dd, dga, dbe, dwt, ws)
    # (a) the three launches
    dp3 = torch.empty(N, H, W, C, device="cuda", dtype=torch.bfloat16)
    dga3, dbe3 = torch.empty(C, device="cuda"), torch.empty(C, device="cuda")
    K.bn_relu_backward(dyb, pb, B, mean, invstd, gamma.cuda(), beta.cuda(), training, dp3, dga3, dbe3, ws)
    dw3 = torch.empty(C, C, 1, 1, device="cuda")
    K.conv_wgrad(K.Sl(db_), C, K.Sl(dp3), dw3, None, ws, 1, math=K.MATH_BF16)
    dd3 = torch.empty(N, H, W, C, device="cuda", dtype=torch.bfloat16)
    K.conv_forward(K.Sl(dp3), K.conv_pack(w.cuda(), True, C, C, math=K.MATH_BF16), None, K.Sl(dd3), 1, math=K.MATH_BF16)
    assert torch.equal(dga, dga3) and torch.equal(dbe, dbe3)
    assert rel(dd.float(), dd3.float()) < 1e-2 and rel(dwt, dw3) < 2e-5
    # (b) autograd
    assert rel(from_nhwc(dd.float()), dd_ref) < 1.5e-2          # dp and dd each rounded to bf16 once
    assert rel(dwt.reshape(C, C), dw_ref) < 1e-2
    assert rel(dga, gg.grad) < 1e-4 and rel(dbe, bg.grad) < 1e-4


@pytest.mark.parametrize("with_bn", [False, True])
@pytest.mark.parametrize("stats", [True, False])
@pytest.mark.parametrize("B,G,H,W", [(2, 3, 9, 14), (1, 2, 16, 64), (3, 1, 5, 7), (9, 3, 17, 40)])
def test_depthwise_pointwise_batchnorm_sums_forward_in_one_pass(K, with_bn, stats, B, G, H, W):
    """nvq_dwpw_forward: depthwise 3x3 -> pointwise 1x1 -> BatchNorm statistics from one staged halo tile against (a) the three
    launches it replaces and (b) the reference ops (efficient_layers.py:49-66); ragged tiles, more tiles than workgroups per
    group (B = 9: persistent loop), with / without the previous layer's BatchNorm + ReLU on the input, with / without sums."""
    C, N = 64, B * G
    x = bf(rnd(N, C, H, W, seed=1) * 1.5 + 0.2)
    wd, wp = rnd(C, 1, 3, 3, seed=2), bf(rnd(C, C, 1, 1, scale=0.15, seed=3))
    xb = to_nhwc_bf16(x)
    ws = ws_tensor(K)
    bn, xin = None, x
    if with_bn:
        gamma, beta = (1 + 0.2 * rnd(C, seed=4)).cuda(), (0.1 * rnd(C, seed=5)).cuda()
        m0, i0 = torch.empty(G, C, device="cuda"), torch.empty(G, C, device="cuda")
        K.bn_stats(xb, B, list(range(G)), m0, i0, None, None, ws)
        bn = (m0, i0, gamma, beta, B)
        r = torch.empty(N, H, W, C, device="cuda", dtype=torch.bfloat16)
        K.bn_apply_relu(xb, B, m0, i0, gamma, beta, None, K.Sl(r), N)
        xin = from_nhwc(r.float()).cpu()                        # what the staged tile holds
    order = list(range(G))[::-1]
    rm, rv = (0.1 * rnd(C, seed=6)).cuda(), (1 + 0.3 * rnd(C, seed=7).abs()).cuda()
    rm3, rv3 = rm.clone(), rv.clone()
    d = torch.zeros(N, H, W, C, device="cuda", dtype=torch.bfloat16)
    p = torch.zeros(N, H, W, C, device="cuda", dtype=torch.bfloat16)
    mean, invstd = torch.zeros(G, C, device="cuda"), torch.zeros(G, C, device="cuda")
    K.dwpw_forward(xb, bn, wd.cuda(), wp.cuda(), d, p, B, order if stats else None, mean, invstd, rm, rv, ws)
    # (a) the three launches
    d3 = torch.empty_like(d)
    K.dwconv_forward(xb, wd.cuda(), d3, bn=bn)
    p3 = torch.empty_like(p)
    K.conv_forward(K.Sl(d3), K.conv_pack(wp.cuda(), False, C, math=K.MATH_BF16), None, K.Sl(p3), 1, math=K.MATH_BF16)
    assert torch.equal(d, d3)
    assert rel(p.float(), p3.float()) < 2e-3                    # same operands; summation order / one bf16 rounding apart
    if stats:
        mean3, invstd3 = torch.empty(G, C, device="cuda"), torch.empty(G, C, device="cuda")
        K.bn_stats(p, B, order, mean3, invstd3, rm3, rv3, ws)
        assert (mean - mean3).abs().max() < 1e-5 * (1 + mean3.abs().max()) and rel(invstd, invstd3) < 1e-5
        assert rel(rm, rm3) < 1e-5 and rel(rv, rv3) < 1e-5
    else:
        assert not mean.any() and not invstd.any() and torch.equal(rm, rm3)
    # (b) the reference ops on the same bf16 operands
    d_ref = F.conv2d(xin, wd, None, padding=1, groups=C)
    assert rel(from_nhwc(d.float()).cpu(), d_ref) < 5e-3
    p_ref = F.conv2d(bf(d_ref), wp)
    assert rel(from_nhwc(p.float()).cpu(), p_ref) < 8e-3


@pytest.mark.parametrize("with_bn", [False, True])
@pytest.mark.parametrize("with_epi", [False, True])
@pytest.mark.parametrize("B,G,H,W", [(2, 3, 9, 14), (1, 2, 16, 64), (3, 1, 5, 7), (9, 3, 17, 40)])
def test_depthwise_backward_input_and_weight_gradient_from_one_tile(K, with_bn, with_epi, B, G, H, W):
    """nvq_dwconv_backward against (a) the two launches it replaces (weight gradient; flipped conv with the add / mask epilogue)
    and (b) autograd of the reference's nn.Conv2d(groups = C) (efficient_layers.py:49-66); ragged tiles, persistent loop (B = 9),
    with / without the previous layer's BatchNorm + ReLU on the conv input."""
    C, N = 64, B * G
    x = bf(rnd(N, C, H, W, seed=1) * 1.5 + 0.2)
    dy = bf(rnd(N, C, H, W, seed=2))
    wd = rnd(C, 1, 3, 3, seed=3)
    xb, dyb = to_nhwc_bf16(x), to_nhwc_bf16(dy)
    ws = ws_tensor(K)
    bn, xin = None, x
    if with_bn:
        gamma, beta = (1 + 0.2 * rnd(C, seed=4)).cuda(), (0.1 * rnd(C, seed=5)).cuda()
        m0, i0 = torch.empty(G, C, device="cuda"), torch.empty(G, C, device="cuda")
        K.bn_stats(xb, B, list(range(G)), m0, i0, None, None, ws)
        bn = (m0, i0, gamma, beta, B)
        r = torch.empty(N, H, W, C, device="cuda", dtype=torch.bfloat16)
        K.bn_apply_relu(xb, B, m0, i0, gamma, beta, None, K.Sl(r), N)
        xin = from_nhwc(r.float())                              # what the conv was fed
    add = to_nhwc(rnd(N, C, H, W, seed=6)) if with_epi else None
    mask_c = bf(rnd(N, C, H, W, seed=7)).clamp_min(0) if with_epi else None
    mask = to_nhwc_bf16(mask_c) if with_epi else None
    dx = torch.zeros(N, H, W, C, device="cuda", dtype=torch.bfloat16)
    dw = torch.zeros(C, 1, 3, 3, device="cuda")
    K.dwconv_backward(xb, bn, dyb, wd.cuda(), dx, dw, ws, add=add, mask=mask)
    # (a) the two launches
    dw2 = torch.empty(C, 1, 3, 3, device="cuda")
    K.dwconv_wgrad(xb, dyb, dw2, ws, bn=bn)
    dx2 = torch.empty_like(dx)
    K.dwconv_forward(dyb, wd.cuda(), dx2, flip=True, add=add, mask=mask)
    assert torch.equal(dx, dx2)
    assert rel(dw, dw2) < 2e-5
    # (b) autograd
    xg, wg = xin.clone().requires_grad_(), wd.clone().requires_grad_()
    F.conv2d(xg, wg, None, padding=1, groups=C).backward(dy)
    want = xg.grad
    if with_epi:
        want = (want + from_nhwc(add)) * (mask_c > 0)
    assert rel(from_nhwc(dx.float()), want) < 5e-3             # one bf16 rounding of the result
    assert rel(dw, wg.grad) < TOL
    if with_bn and not with_epi:
        # the same call also returning the backward sums of the input's BatchNorm (x = its input, dx = the gradient of its
        # activation): equal to what the BatchNorm backward computes from (dx, x) in a pass of its own, and usable by
        # nvq_pw_bn_backward in place of that pass
        sums = torch.zeros(G, 2, C, device="cuda")
        dga, dbe = torch.zeros(C, device="cuda"), torch.zeros(C, device="cuda")
        dx_s, dw_s = torch.zeros_like(dx), torch.zeros_like(dw)
        K.dwconv_backward(xb, bn, dyb, wd.cuda(), dx_s, dw_s, ws, bn_sums=sums, bn_dgamma=dga, bn_dbeta=dbe)
        assert torch.equal(dx_s, dx) and rel(dw_s, dw) < 2e-5
        dp_ref = torch.empty_like(dx)
        dga_ref, dbe_ref = torch.empty(C, device="cuda"), torch.empty(C, device="cuda")
        K.bn_relu_backward(dx, xb, B, m0, i0, gamma, beta, True, dp_ref, dga_ref, dbe_ref, ws)
        assert rel(dga, dga_ref) < 2e-5 and rel(dbe, dbe_ref) < 2e-5
        # through the pointwise backward: the hand-off gives the same dd / dweight as its own reduce pass
        wpw = bf(rnd(C, C, 1, 1, scale=0.15, seed=9)).cuda()
        dprev = to_nhwc_bf16(bf(rnd(N, C, H, W, seed=10)))
        outs = []
        for s_in in (None, sums):
            dd = torch.zeros_like(dx)
            g1, g2, gw = torch.zeros(C, device="cuda"), torch.zeros(C, device="cuda"), torch.zeros(C, C, 1, 1, device="cuda")
            K.pw_bn_backward(dx, xb, dprev, B, m0, i0, gamma, beta, True, wpw, dd, g1, g2, gw, ws, sums_in=s_in)
            outs.append((dd, gw, g1))
        assert rel(outs[1][0].float(), outs[0][0].float()) < 1e-2 and rel(outs[1][1], outs[0][1]) < 1e-4
        assert rel(outs[0][2], dga_ref) < 2e-5 and not outs[1][2].any()      # with sums_in, dgamma / dbeta are left alone


@pytest.mark.parametrize("C,N,H,W", [(64, 2, 11, 37), (128, 1, 8, 32), (64, 3, 17, 70)])
def test_dwconv_bf16_full_line_kernels(K, C, N, H, W):
    """The 64-channel bf16 depthwise kernels (forward, flipped = input gradient, weight gradient), ragged tiles,
    fp32 and bf16 outputs."""
    x = bf(rnd(N, C, H, W) * 1.5 + 0.3)
    w = rnd(C, 1, 3, 3)
    ref = F.conv2d(x, w, None, padding=1, groups=C)
    for odt, tol in ((torch.float32, TOL), (torch.bfloat16, 5e-3)):
        out = torch.empty(N, H, W, C, device="cuda", dtype=odt)
        K.dwconv_forward(to_nhwc_bf16(x), w.cuda(), out)
        assert rel(out.float().permute(0, 3, 1, 2), ref) < tol
    dy = bf(rnd(N, C, H, W, seed=6))
    xg, wg = x.clone().requires_grad_(), w.clone().requires_grad_()
    F.conv2d(xg, wg, None, padding=1, groups=C).backward(dy)
    dx = torch.empty(N, H, W, C, device="cuda")
    K.dwconv_forward(to_nhwc_bf16(dy), w.cuda(), dx, flip=True)
    assert rel(dx.permute(0, 3, 1, 2), xg.grad) < TOL
    dw = torch.empty(C, 1, 3, 3, device="cuda")
    K.dwconv_wgrad(to_nhwc_bf16(x), to_nhwc_bf16(dy), dw, ws_tensor(K))
    assert rel(dw, wg.grad) < TOL


@pytest.mark.parametrize("C,B,R,H,W", [(64, 2, 2, 9, 14), (64, 1, 1, 11, 37), (32, 1, 3, 20, 6), (64, 1, 2, 17, 33)])
@pytest.mark.parametrize("store_bf16", [False, True])
def test_correlation_mfma_forward_backward(K, C, B, R, H, W, store_bf16):
    """NVQ_MATH_BF16 correlation (matrix cores): inputs pre-rounded to bf16 so that only the fp32 summation order
    differs from the oracle; ragged tiles, shared centre images, fp32- and bf16-stored corr / dcorr."""
    N = B * R
    x1 = bf(rnd(N, C, H, W)).requires_grad_()
    x2 = bf(rnd(B, C, H, W, seed=3)).requires_grad_()
    out = torch.cat([sr_oracle.correlation(x1[r * B:(r + 1) * B], x2) for r in range(R)], 0)
    dy = bf(rnd(N, 81, H, W, seed=5))
    out.backward(dy)
    x1b = to_nhwc(x1.detach())
    al = to_nhwc(x2.detach(), 3 * C, C)
    ld = 128 if store_bf16 else 96
    corr = torch.full((N, H, W, ld), 3.0, device="cuda", dtype=torch.bfloat16 if store_bf16 else torch.float32)
    K.correlation_forward(K.Sl(x1b), K.Sl(al, C, C), corr, math=K.MATH_BF16)
    assert rel(from_nhwc(corr.float(), 81), out.detach()) < (5e-3 if store_bf16 else TOL)
    assert corr[..., 81:].float().abs().max().item() == 0
    dcorr = to_nhwc(dy, ld)
    if store_bf16:
        dcorr = dcorr.bfloat16()
    dx1 = to_nhwc(rnd(N, C, H, W, seed=7))
    base1 = from_nhwc(dx1)
    K.correlation_backward(1, dcorr, K.Sl(al, C, C), K.Sl(dx1), True, math=K.MATH_BF16)
    assert rel(from_nhwc(dx1) - base1, x1.grad) < TOL
    dx2 = torch.zeros(B, H, W, C, device="cuda")
    for r in range(R):
        K.correlation_backward(2, dcorr[r * B:(r + 1) * B], K.Sl(x1b).images(r * B, (r + 1) * B), K.Sl(dx2), True,
                               math=K.MATH_BF16)
    assert rel(from_nhwc(dx2), x2.grad) < TOL
    dx2m = torch.zeros(B, H, W, C, device="cuda")
    K.correlation_backward(2, dcorr, K.Sl(x1b), K.Sl(dx2m), False, math=K.MATH_BF16, groups=R)
    assert rel(from_nhwc(dx2m), x2.grad) < TOL
    assert rel(dx2m, dx2) < 1e-6                              # same products, same order per pixel
    # the last pass over an accumulated gradient can leave as bf16 instead (dx is then only read)
    acc0 = to_nhwc(rnd(N, C, H, W, seed=7))
    keep = acc0.clone()
    o16 = torch.zeros(N, H, W, C, device="cuda", dtype=torch.bfloat16)
    K.correlation_backward(1, dcorr, K.Sl(al, C, C), K.Sl(acc0), True, math=K.MATH_BF16, out16=o16)
    assert torch.equal(o16, dx1.bfloat16()) and torch.equal(acc0, keep)
    acc2 = to_nhwc(rnd(B, C, H, W, seed=8))
    want2 = acc2.clone()
    K.correlation_backward(2, dcorr, K.Sl(x1b), K.Sl(want2), True, math=K.MATH_BF16, groups=R)
    o2 = torch.zeros(B, H, W, C, device="cuda", dtype=torch.bfloat16)
    K.correlation_backward(2, dcorr, K.Sl(x1b), K.Sl(acc2), True, math=K.MATH_BF16, groups=R, out16=o2)
    assert torch.equal(o2, want2.bfloat16())
    # ... and take the other terms of the sum as bf16 addends in the same pass (no fp32 accumulator at all)
    ta = to_nhwc_bf16(bf(rnd(B, C, H, W, seed=12)))
    tb = to_nhwc_bf16(bf(rnd(B, 3 * C, H, W, seed=13)))
    o3 = torch.zeros(B, H, W, C, device="cuda", dtype=torch.bfloat16)
    K.correlation_backward(2, dcorr, K.Sl(x1b), K.Sl(acc2), False, math=K.MATH_BF16, groups=R, out16=o3,
                           addends=(K.Sl(ta), K.Sl(tb, C, 2 * C)))
    assert torch.equal(acc2, want2 * 0 + acc2)                # (dx untouched)
    assert torch.equal(o3, (dx2m + ta.float() + tb[..., 2 * C:].float()).bfloat16())


@pytest.mark.parametrize("C,B,R,H,W", [(64, 1, 2, 70, 20), (32, 2, 1, 45, 37), (64, 1, 1, 8, 16), (64, 1, 1, 131, 9)])
@pytest.mark.parametrize("store_bf16", [False, True])
def test_correlation_forward_strip_kernel(K, C, B, R, H, W, store_bf16):
    """bf16-stored features take the strip form of the forward (a workgroup walks down a column block with the halo rows as a
    ring in LDS): segments of several tiles (H = 70, 131: the ring rotates more than once), of one tile, and empty ones (H = 8);
    against the tile form on the same values stored as fp32 (same operands, same MFMA order: equal) and the oracle."""
    N = B * R
    x1 = bf(rnd(N, C, H, W))
    x2 = bf(rnd(B, C, H, W, seed=3))
    want = torch.cat([sr_oracle.correlation(x1[r * B:(r + 1) * B], x2) for r in range(R)], 0)
    ld = 96
    odt = torch.bfloat16 if store_bf16 else torch.float32
    got = torch.full((N, H, W, ld), 3.0, device="cuda", dtype=odt)
    K.correlation_forward(K.Sl(to_nhwc_bf16(x1)), K.Sl(to_nhwc_bf16(x2, 3 * C, C), C, C), got, math=K.MATH_BF16)
    tile = torch.full((N, H, W, ld), 3.0, device="cuda", dtype=odt)
    K.correlation_forward(K.Sl(to_nhwc(x1)), K.Sl(to_nhwc(x2, 3 * C, C), C, C), tile, math=K.MATH_BF16)
    assert torch.equal(got, tile)
    assert rel(from_nhwc(got.float(), 81), want) < (5e-3 if store_bf16 else TOL)
    assert got[..., 81:].float().abs().max().item() == 0


@pytest.mark.parametrize("C,B,R,H,W", [(64, 1, 2, 70, 20), (32, 2, 1, 45, 37), (64, 1, 1, 8, 16), (64, 1, 1, 131, 9)])
@pytest.mark.parametrize("mode", ["overwrite", "accumulate", "bf16 out + addend"])
def test_correlation_x1_gradient_strip_kernel(K, C, B, R, H, W, mode):
    """All-bf16 tensors take the strip form of the x1 gradient (ring of halo rows, prefetched dcorr rows); against the tile form
    on the same values with `other` stored as fp32 (same operands, same MFMA order: equal), in every epilogue mode."""
    N = B * R
    oth = bf(rnd(B, C, H, W, seed=3))
    dcorr = to_nhwc_bf16(bf(rnd(N, 81, H, W, seed=5)), 96)
    outs = []
    for other in (to_nhwc_bf16(oth), to_nhwc(oth)):            # strip form / tile form
        dx = to_nhwc(rnd(N, C, H, W, seed=7))
        if mode == "bf16 out + addend":
            o16 = torch.zeros(N, H, W, C, device="cuda", dtype=torch.bfloat16)
            ta = to_nhwc_bf16(bf(rnd(N, 2 * C, H, W, seed=12)))
            K.correlation_backward(1, dcorr, K.Sl(other), K.Sl(dx), False, math=K.MATH_BF16, out16=o16, addends=(K.Sl(ta, C, C),))
            outs.append(o16)
        else:
            K.correlation_backward(1, dcorr, K.Sl(other), K.Sl(dx), mode == "accumulate", math=K.MATH_BF16)
            outs.append(dx)
    assert torch.equal(outs[0], outs[1])


@pytest.mark.parametrize("C,B,R,H,W", [(64, 1, 2, 70, 20), (32, 2, 1, 45, 37), (64, 3, 2, 8, 16), (64, 40, 1, 9, 17)])
@pytest.mark.parametrize("mode", ["overwrite", "accumulate", "bf16 out + addends"])
def test_correlation_x2_gradient_prefetching_kernel(K, C, B, R, H, W, mode):
    """All-bf16 tensors take the prefetching form of the x2 gradient (persistent workgroups over (tile, frame group) steps, the
    next step's halos fetched under the MFMAs; more tiles than workgroups at B = 40); against the tile form on the same values
    with `other` stored as fp32 (same operands, same order: equal), in every epilogue mode, one and two frame groups."""
    N = B * R
    oth = bf(rnd(N, C, H, W, seed=3))
    dcorr = to_nhwc_bf16(bf(rnd(N, 81, H, W, seed=5)), 96)
    outs = []
    for other in (to_nhwc_bf16(oth), to_nhwc(oth)):            # prefetching form / tile form
        dx = to_nhwc(rnd(B, C, H, W, seed=7))
        if mode == "bf16 out + addends":
            o16 = torch.zeros(B, H, W, C, device="cuda", dtype=torch.bfloat16)
            ta, tb = to_nhwc_bf16(bf(rnd(B, 2 * C, H, W, seed=12))), to_nhwc_bf16(bf(rnd(B, C, H, W, seed=13)))
            K.correlation_backward(2, dcorr, K.Sl(other), K.Sl(dx), False, math=K.MATH_BF16, groups=R, out16=o16,
                                   addends=(K.Sl(ta, C, C), K.Sl(tb)))
            outs.append(o16)
        else:
            K.correlation_backward(2, dcorr, K.Sl(other), K.Sl(dx), mode == "accumulate", math=K.MATH_BF16, groups=R)
            outs.append(dx)
    assert torch.equal(outs[0], outs[1])


@pytest.mark.parametrize("Fc,N,H,W", [(64, 2, 19, 37), (64, 1, 16, 32), (32, 2, 9, 20)])
def test_slice_planar_dense_block_buffer(K, Fc, N, H, W):
    """nvq_conv_desc::in_plane / nvq_wgrad_desc::x_plane: the dense-block buffer as compact tensors [x | y_0 | ..] in one
    allocation (K.CatBuf).  Dense layers, the fused tail (F = 64), the mirror-style 1x1 / 3x3 convs over a channel prefix
    and the weight gradients give bit-identical results to the interleaved buffer."""
    CAT, ld = Fc + 160, 256
    full = bf(rnd(N, CAT, H, W))
    inter = K.CatBuf("cuda", N, H, W, Fc, 5, ld, torch.bfloat16, planar=False)
    inter.t.zero_()
    inter.t[..., :CAT] = to_nhwc(full).bfloat16()
    plan = K.CatBuf("cuda", N, H, W, Fc, 5, ld, torch.bfloat16, planar=True)
    plan.lead.copy_(inter.t[..., :Fc])
    for j in range(5):
        plan.slices[j].copy_(inter.t[..., Fc + 32 * j:Fc + 32 * (j + 1)])
    ws = ws_tensor(K)
    for i in (0, 1, 3):                                      # dense layer i: prefix of Fc + 32 i channels -> slice i
        cin = Fc + 32 * i
        w3, b3 = rnd(32, cin, 3, 3, scale=0.1, seed=i), rnd(32, seed=20 + i)
        wp = K.conv_pack(w3.cuda(), False, cin, math=K.MATH_BF16)
        bits = [torch.zeros(N, H, W, dtype=torch.int32, device="cuda") for _ in range(2)]
        for buf, bt in ((inter, bits[0]), (plan, bits[1])):
            K.conv_forward(buf.inp(cin), wp, b3.cuda(), buf.y(i), 3, relu=True, math=K.MATH_BF16, bits=bt, bits_mode=1)
        assert torch.equal(inter.t[..., cin:cin + 32], plan.slices[i]) and torch.equal(bits[0], bits[1])
        dws = []
        for buf, variant in ((inter, 0), (plan, 1), (plan, 2 if cin >= 96 else 0)):   # weight gradient of the same layer
            dw, db = torch.empty(32, cin, 3, 3, device="cuda"), torch.empty(32, device="cuda")
            K.conv_wgrad(buf.inp(cin), cin, buf.y(4), dw, db, ws, 3, math=K.MATH_BF16, variant=variant)
            dws.append((dw, db))
        assert torch.equal(dws[0][0], dws[1][0]) and torch.equal(dws[0][1], dws[1][1])   # the same kernel on either layout
        # variant 2 on the slice-planar buffer is the all-input-channel kernel (wgrad_m32.hip): other fp32 summation order
        assert rel(dws[2][0], dws[1][0]) < TOL and rel(dws[2][1], dws[1][1]) < TOL
    # 1x1 over all channels (lff) with residual x, 64-channel output into the next block's x; its weight gradient
    wl, bl = rnd(Fc, CAT, 1, 1, scale=0.1, seed=7), rnd(Fc, seed=8)
    wlp = K.conv_pack(wl.cuda(), False, CAT, math=K.MATH_BF16)
    nxt = [K.CatBuf("cuda", N, H, W, Fc, 5, ld, torch.bfloat16, planar=pl) for pl in (False, True)]
    for buf, nb in zip((inter, plan), nxt):
        K.conv_forward(buf.inp(CAT), wlp, bl.cuda(), nb.x(), 1, alpha=0.2, res=buf.x(), math=K.MATH_BF16)
    assert torch.equal(nxt[0].t[..., :Fc], nxt[1].lead)
    dws = []
    for buf, nb in zip((inter, plan), nxt):
        dw, db = torch.empty(Fc, CAT, 1, 1, device="cuda"), torch.empty(Fc, device="cuda")
        K.conv_wgrad(buf.inp(CAT), CAT, nb.x(), dw, db, ws, 1, alpha=0.2, math=K.MATH_BF16, variant=1)
        dws.append(dw)
        if CAT % 32 == 0 and Fc % 32 == 0 and 64 <= CAT <= 256 and Fc <= 64:   # the all-input-channel 1x1 kernel, either layout
            dw2, db2 = torch.empty_like(dw), torch.empty_like(db)
            K.conv_wgrad(buf.inp(CAT), CAT, nb.x(), dw2, db2, ws, 1, alpha=0.2, math=K.MATH_BF16, variant=2)
            assert rel(dw2, dw) < TOL and rel(db2, db) < TOL
    assert torch.equal(dws[0], dws[1])
    # 3x3 over all channels -> 64 (the block-input gradient conv) with the centre-tap hint
    wx = rnd(Fc, CAT, 3, 3, scale=0.1, seed=9)
    wx[:, :Fc, [0, 0, 0, 1, 1, 2, 2, 2], [0, 1, 2, 0, 2, 0, 1, 2]] = 0
    wxp = K.conv_pack(wx.cuda(), False, CAT, math=K.MATH_BF16)
    outs = []
    for buf in (inter, plan):
        o = torch.empty(N, H, W, Fc, device="cuda", dtype=torch.bfloat16)
        K.conv_forward(buf.inp(CAT), wxp, None, K.Sl(o), 3, res=buf.x(), math=K.MATH_BF16, center_cin=Fc if Fc % 32 == 0 else 0)
        outs.append(o)
    assert torch.equal(outs[0], outs[1])
    if Fc == 64:                                             # fused tail: last dense layer + lff in one launch
        cin = Fc + 128
        w3, b3 = rnd(32, cin, 3, 3, scale=0.1, seed=11), rnd(32, seed=12)
        w3p = K.conv_pack(w3.cuda(), False, cin, math=K.MATH_BF16)
        for buf, nb in zip((inter, plan), nxt):
            K.rdb_tail_forward(buf.inp(cin), w3p, b3.cuda(), buf.y(4), wlp, bl.cuda(), nb.x(), alpha=0.2, res=buf.x())
        assert torch.equal(inter.t[..., cin:cin + 32], plan.slices[4]) and torch.equal(nxt[0].t[..., :Fc], nxt[1].lead)
    with pytest.raises(RuntimeError, match="slice-planar"):
        K.conv_forward(K.Sl(plan.lead, Fc + 32, 0, plane=plan.plane + 32), wp, None, plan.y(4), 3, math=K.MATH_BF16)


@pytest.mark.parametrize("C,B,H,W", [(64, 2, 19, 37), (32, 1, 9, 20)])
def test_bf16_stored_feature_tensors(K, C, B, H, W):
    """The storage flags of the non-conv readers of the feature tensors (correlation on the matrix cores, warp, softmax-
    weighted sum): with bf16-stored inputs every kernel returns exactly what it returns for fp32 tensors holding the same
    (bf16-representable) values - the flag changes how a value is loaded, nothing else."""
    T, R = 3, 2
    N = B * R
    feat = bf(rnd(N, C, H, W))                               # neighbours' features
    al = bf(rnd(B, T * C, H, W, seed=2))                     # aligned: [centre | warped neighbours]
    f32, a32 = to_nhwc(feat), to_nhwc(al)
    f16, a16 = f32.bfloat16(), a32.bfloat16()
    assert torch.equal(f16.float(), f32) and torch.equal(a16.float(), a32)
    # correlation forward / both gradients
    outs = []
    for fb, ab in ((f32, a32), (f16, a16)):
        corr = torch.empty(N, H, W, 128, device="cuda", dtype=torch.bfloat16)
        K.correlation_forward(K.Sl(fb), K.Sl(ab, C, C), corr, math=K.MATH_BF16)
        dcorr = to_nhwc(bf(rnd(N, 81, H, W, seed=5)), 128).bfloat16()
        dx1 = torch.zeros(N, H, W, C, device="cuda")
        K.correlation_backward(1, dcorr, K.Sl(ab, C, C), K.Sl(dx1), False, math=K.MATH_BF16)
        dx2 = torch.zeros(B, H, W, C, device="cuda")
        for r in range(R):
            K.correlation_backward(2, dcorr[r * B:(r + 1) * B], K.Sl(fb).images(r * B, (r + 1) * B), K.Sl(dx2), True,
                                   math=K.MATH_BF16)
        outs.append((corr, dx1, dx2))
    for a, b in zip(*outs):
        assert torch.equal(a, b)
    # warp forward (fp32 and bf16 output) and backward
    flow = to_nhwc(rnd(B, 2, H, W, seed=7) * 1.7, 4)
    outs = []
    dal32 = to_nhwc(bf(rnd(B, T * C, H, W, seed=9)))         # gradient w.r.t. aligned, bf16-representable
    for fb, odt in ((f32, torch.float32), (f16, torch.float32), (f16, torch.bfloat16)):
        wo = torch.zeros(B, H, W, T * C, device="cuda", dtype=odt)
        K.warp_forward(K.Sl(fb).images(0, B), flow, K.Sl(wo, C, 2 * C))
        dfeat, dflow = torch.zeros(B, H, W, C, device="cuda"), torch.empty(B, H, W, 4, device="cuda")
        K.warp_backward(K.Sl(dal32.to(odt), C, 2 * C), K.Sl(fb).images(0, B), flow, K.Sl(dfeat), dflow)
        outs.append((wo, dfeat, dflow))
    assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[2][0], outs[0][0].bfloat16())
    for k in (1, 2):
        assert torch.equal(outs[0][k], outs[1][k]) and torch.equal(outs[0][k], outs[2][k])
    assert outs[0][0][..., :2 * C].abs().max().item() == 0
    for fb, db in ((f16, dal32), (f32, dal32.bfloat16())):
        with pytest.raises(RuntimeError, match="gather form"):
            K.warp_backward(K.Sl(db, C, 2 * C), K.Sl(fb).images(0, B), flow, K.Sl(torch.zeros(B, H, W, C, device="cuda")),
                            torch.empty(B, H, W, 4, device="cuda"), gather=False)
    # slice axpy with a bf16-stored source
    d0 = to_nhwc(rnd(B, C, H, W, seed=11))
    da, db = d0.clone(), d0.clone()
    K.axpy_slice(K.Sl(da), K.Sl(dal32, C, C), alpha=0.5)
    K.axpy_slice(K.Sl(db), K.Sl(dal32.bfloat16(), C, C), alpha=0.5)
    assert torch.equal(da, db) and not torch.equal(da, d0)
    # softmax-weighted sum forward / backward
    lg = to_nhwc(rnd(B, T, H, W, seed=3) * 3, 4)
    outs = []
    for ab in (a32, a16):
        attn, wt = torch.empty(B, H, W, 4, device="cuda"), torch.empty(B, H, W, C, device="cuda")
        gp = torch.empty(B, K.tsum_blocks(H, W), C, device="cuda")
        K.tsum_forward(ab, lg, T, C, attn, wt, gp)
        dal, dlg = torch.empty(B, H, W, T * C, device="cuda"), torch.empty(B, H, W, 4, device="cuda")
        K.tsum_backward(to_nhwc(rnd(B, C, H, W, seed=6)), (rnd(B, C, seed=8) * 0.1).cuda(), ab, attn, T, C, dal, dlg)
        outs.append((attn, wt, gp, dal, dlg))
    for a, b in zip(*outs):
        assert torch.equal(a, b)
    dal16, dlg16 = torch.empty(B, H, W, T * C, device="cuda", dtype=torch.bfloat16), torch.empty(B, H, W, 4, device="cuda")
    K.tsum_backward(to_nhwc(rnd(B, C, H, W, seed=6)), (rnd(B, C, seed=8) * 0.1).cuda(), a16, outs[1][0], T, C, dal16, dlg16)
    assert torch.equal(dal16, outs[1][3].bfloat16()) and torch.equal(dlg16, outs[1][4])     # bf16-stored gradient output
    # exact-fp32 correlation refuses bf16-stored inputs
    with pytest.raises(RuntimeError, match="NVQ_MATH_BF16"):
        K.correlation_forward(K.Sl(f16), K.Sl(a16, C, C), torch.empty(N, H, W, 96, device="cuda"), math=K.MATH_F32)


@pytest.mark.parametrize("cin,N,H,W", [(96, 2, 35, 41), (64, 1, 16, 32), (160, 3, 50, 70)])
def test_conv_bf16_tall_tiles_mask_and_bias(K, cin, N, H, W):
    """The 16x32-tile 3x3 kernel (bf16 input, cout 32, H >= 16): bias + ReLU written in place into the concat buffer, and
    the mirror-form gradient form (no bias, ReLU mask from a bf16 slice); interior and border tiles."""
    ld = 256
    cat = bf(rnd(N, ld, H, W))
    w3, b3 = rnd(32, cin, 3, 3, scale=0.1), rnd(32)
    catb = to_nhwc_bf16(cat)
    K.conv_forward(K.Sl(catb, cin, 0), K.conv_pack(w3.cuda(), False, cin, math=K.MATH_BF16), b3.cuda(), K.Sl(catb, 32, cin), 3,
                   relu=True, math=K.MATH_BF16)
    ref3 = F.relu(F.conv2d(cat[:, :cin], bf(w3), b3, padding=1))
    assert rel(catb[..., cin:cin + 32].float().permute(0, 3, 1, 2), ref3) < 5e-3
    assert torch.equal(catb[..., :cin].float().cpu(), to_nhwc(cat[:, :cin]).cpu())
    # gradient form: out = conv(x) masked by (m > 0), fp32 output
    m = bf(rnd(N, 32, H, W, seed=9))
    mb = to_nhwc_bf16(m, 64, 32)
    out = torch.full((N, H, W, 32), 5.0, device="cuda")
    K.conv_forward(K.Sl(to_nhwc_bf16(cat), cin, 0), K.conv_pack(w3.cuda(), False, cin, math=K.MATH_BF16), None, K.Sl(out), 3,
                   mask=K.Sl(mb, 32, 32), mask_c0=0, mask_c1=32, math=K.MATH_BF16)
    refm = F.conv2d(cat[:, :cin], bf(w3), None, padding=1) * (m > 0)
    assert rel(from_nhwc(out), refm) < TOL


@pytest.mark.parametrize("cin,cout,ctr,res,H,W", [(96, 32, 0, False, 35, 41), (192, 32, 64, False, 33, 70), (224, 64, 64, True, 19, 40),
                                                   (96, 128, 0, False, 24, 33), (64, 81, 0, False, 17, 50)])
def test_conv_bf16_eight_wave_kernels_equal_the_four_wave_ones(K, cin, cout, ctr, res, H, W):
    """The two 8-wave 3x3 kernels the benchmark runs - 16x32 tiles for cout <= 32, and the channel-split one (two halves of the
    workgroup = 32 output channels each) for cout 64 / 128 - against the single-tile 4-wave kernels they replaced
    (nvq_conv_desc::tile_rows = 8 selects those): the same sums in the same order, bit-identical, with bias / residual /
    centre-tap chunks / a second 64-channel slab (cout 128, 81)."""
    N = 2
    w = rnd(cout, cin, 3, 3, scale=0.1)
    if ctr:
        centre = w[:, :ctr, 1, 1].clone()
        w[:, :ctr] = 0
        w[:, :ctr, 1, 1] = centre
    x = bf(rnd(N, cin, H, W, seed=5))
    xin = to_nhwc_bf16(x, 256)
    wp = K.conv_pack(w.cuda(), False, cin, math=K.MATH_BF16)
    bias = rnd(cout, seed=6).cuda() if cout % 4 == 0 else None   # (the 16-byte epilogue wants whole 4-channel groups of bias)
    r16 = to_nhwc_bf16(bf(rnd(N, 64, H, W, seed=7)), 64) if res else None
    cs = (cout + 7) // 8 * 8
    outs = []
    for rows in (16, 8, 0):                                  # (16: the 16x16x32 eight-wave form whatever the automatic choice is)
        out = torch.full((N, H, W, cs), 3.0, device="cuda").bfloat16()
        K.conv_forward(K.Sl(xin, cin, 0), wp, bias, K.Sl(out, cout), 3, cout_store=cs, math=K.MATH_BF16, center_cin=ctr,
                       res=K.Sl(r16) if res else None, tile_rows=rows)
        outs.append(out)
    assert torch.equal(outs[0], outs[1])
    # the automatic choice may be the 32x32x16 form (cout <= 32, cin <= 128): other fp32 order inside a chunk, at most one
    # bf16 ulp of the stored result
    assert rel(outs[2].float(), outs[0].float()) < 8e-3
    ref = F.conv2d(x, bf(w), bias.cpu() if bias is not None else None, padding=1)
    if res:
        ref = ref + from_nhwc(r16.float())[:, :cout]
    assert rel(from_nhwc(outs[0][..., :cout].float()), ref) < 1.2e-2          # one bf16 rounding of the stored result


@pytest.mark.parametrize("rows", [162, 164])
@pytest.mark.parametrize("cin,ctr,N,H,W", [(64, 0, 2, 16, 32), (96, 64, 1, 35, 41), (160, 0, 2, 33, 70), (192, 64, 1, 50, 96),
                                            (40, 0, 1, 19, 37)])
def test_conv_bf16_dense_layer_on_32x32x16_mfma(K, rows, cin, ctr, N, H, W):
    """conv_m32.hip: the cout <= 32 3x3 conv of a bf16 input on v_mfma_f32_32x32x16_bf16 (tile_rows 162: two tile rows per wave,
    eight waves; 164: four rows per wave, four waves), every epilogue form the dense blocks and their mirror-form gradient convs
    use - bias + ReLU + one-bit masks written, bit masks read with centre-tap-only leading channels, residual / second output /
    accumulate / tensor mask with fp32 and bf16 outputs - against F.conv2d on the same bf16 values and against the 16x16x32
    kernel (tile_rows 16); border and interior tiles, a cin that is not a multiple of 32."""
    w, b = rnd(32, cin, 3, 3, scale=0.1), rnd(32, seed=2)
    if ctr:
        centre = w[:, :ctr, 1, 1].clone()
        w[:, :ctr] = 0
        w[:, :ctr, 1, 1] = centre
    x = bf(rnd(N, cin, H, W, seed=5))
    xin = to_nhwc_bf16(x, 256)
    wp = K.conv_pack(w.cuda(), False, cin, math=K.MATH_BF16)
    ref = F.conv2d(x, bf(w), b, padding=1)
    # (a) forward form: bias + ReLU, bf16 output staged through LDS, bit masks written
    res = {}
    for r_ in (rows, 16):
        out = torch.full((N, H, W, 40), 3.0, device="cuda").bfloat16()
        bits = torch.zeros(N, H, W, dtype=torch.int32, device="cuda")
        K.conv_forward(K.Sl(xin, cin, 0), wp, b.cuda(), K.Sl(out, 32, 8), 3, relu=True, math=K.MATH_BF16, bits=bits, bits_mode=1,
                       center_cin=ctr, tile_rows=r_)
        res[r_] = (out, bits)
    out, bits = res[rows]
    assert rel(from_nhwc(out[..., 8:].float()), F.relu(ref)) < 8e-3          # one bf16 rounding of the stored result
    assert out[..., :8].float().abs().max().item() == 3.0                    # the neighbouring channels are untouched
    want = ((out[..., 8:].float() > 0).to(torch.int64) << torch.arange(32, device="cuda")).sum(-1)
    assert torch.equal(bits.to(torch.int64) & 0xFFFFFFFF, want)              # the bits describe the STORED values
    assert rel(out.float(), res[16][0].float()) < 8e-3
    # (b) gradient form: no bias, bit masks read, fp32 output (direct stores)
    g32 = torch.full((N, H, W, 32), 5.0, device="cuda")
    K.conv_forward(K.Sl(xin, cin, 0), wp, None, K.Sl(g32), 3, math=K.MATH_BF16, bits=bits, bits_mode=2, center_cin=ctr, tile_rows=rows)
    refm = F.conv2d(x, bf(w), None, padding=1) * (from_nhwc(out[..., 8:].float()) > 0)
    assert rel(from_nhwc(g32), refm) < TOL
    # (c) alpha, residual, second output, accumulate, tensor mask on a channel range, 24 stored channels of the 32
    r16 = to_nhwc_bf16(bf(rnd(N, 32, H, W, seed=7)), 32)
    m = bf(rnd(N, 32, H, W, seed=9))
    mb = to_nhwc_bf16(m, 64, 32)
    acc0 = rnd(N, 24, H, W, seed=11)
    o1, o2 = to_nhwc(acc0, 24).clone(), torch.zeros(N, H, W, 24, device="cuda")
    w24 = K.conv_pack(w[:24].cuda(), False, cin, math=K.MATH_BF16)
    K.conv_forward(K.Sl(xin, cin, 0), w24, b[:24].cuda(), K.Sl(o1), 3, alpha=0.2, res=K.Sl(r16, 24, 0), out2=K.Sl(o2), accumulate=True,
                   mask=K.Sl(mb, 32, 32), mask_c0=8, mask_c1=16, math=K.MATH_BF16, tile_rows=rows)
    pre = 0.2 * ref[:, :24]
    want1 = pre + from_nhwc(r16.float())[:, :24] + acc0
    want1[:, 8:16] = want1[:, 8:16] * (m[:, 8:16] > 0)
    assert rel(from_nhwc(o2), pre) < TOL and rel(from_nhwc(o1), want1) < TOL


@pytest.mark.parametrize("math_name", ["MATH_F32", "MATH_BF16"])
def test_conv_pack_batch_equals_single_packs(K, math_name):
    """nvq_conv_pack_batch: forward and transposed packs, 1x1 and 3x3, all three cout tile widths, more jobs than one launch
    holds (48) - bit-identical to nvq_conv_pack, job by job."""
    math = getattr(K, math_name)
    shapes = [(32, 96, 3, False, 96, None), (64, 224, 1, False, 224, None), (12, 64, 3, True, 12, 64), (64, 64, 3, True, 64, 64),
              (2, 32, 3, False, 32, None), (128, 81, 3, True, 128, 81), (3, 64, 3, True, 4, 64), (64, 192, 3, True, 64, 192)]
    reqs = []
    for rep in range(7):                                    # 56 jobs
        for i, (co, ci_, k, tr, cs, keep) in enumerate(shapes):
            reqs.append((rnd(co, ci_, k, k, seed=10 * rep + i).cuda(), tr, cs, keep))
    many = K.conv_pack_many(reqs, math)
    assert len(many) == len(reqs)
    for (w, tr, cs, keep), got in zip(reqs, many):
        assert torch.equal(got, K.conv_pack(w, tr, cs, keep, math=math))
    assert K.conv_pack_many([], math) == []


@pytest.mark.parametrize("cin,cout,ctr,in16,H,W", [(96, 32, 64, True, 35, 41), (64, 32, 64, True, 20, 33), (224, 64, 64, True, 19, 40),
                                                    (128, 32, 32, False, 12, 37), (96, 16, 64, True, 9, 30)])
def test_conv_bf16_center_cin_hint(K, cin, cout, ctr, in16, H, W):
    """nvq_conv_desc::center_cin: input channels [0, ctr) whose weights are zero outside the centre tap (the lff^T part of the
    mirror-form dense-block gradient convs) skip the other eight taps: bit-identical to the same launch without the hint,
    and equal to the reference conv; all tile shapes (cout 16 / 32 tall and short / 64, fp32 and bf16 inputs)."""
    N = 2
    w = rnd(cout, cin, 3, 3, scale=0.1)
    centre = w[:, :ctr, 1, 1].clone()
    w[:, :ctr] = 0
    w[:, :ctr, 1, 1] = centre
    x = bf(rnd(N, cin, H, W, seed=3))
    xin = to_nhwc_bf16(x, 256) if in16 else to_nhwc(x)
    wp = K.conv_pack(w.cuda(), False, cin, math=K.MATH_BF16)
    outs = []
    for hint in (0, ctr):
        out = torch.full((N, H, W, cout), 7.0, device="cuda")
        K.conv_forward(K.Sl(xin, cin, 0), wp, None, K.Sl(out), 3, math=K.MATH_BF16, center_cin=hint)
        outs.append(out)
    assert torch.equal(outs[0], outs[1])
    assert rel(from_nhwc(outs[1]), F.conv2d(x, bf(w), None, padding=1)) < TOL
    with pytest.raises(RuntimeError, match="center_cin"):
        K.conv_forward(K.Sl(xin, cin, 0), wp, None, K.Sl(out), 3, math=K.MATH_BF16, center_cin=48)


@pytest.mark.parametrize("N,H,W", [(2, 35, 41), (1, 9, 33)])
def test_conv_bf16_one_bit_relu_masks(K, N, H, W):
    """bits_mode 1 writes bit c = (stored output channel c > 0) per pixel; bits_mode 2 uses the word as the ReLU mask of the
    mirror-form gradient conv: same result as masking with the activation tensor itself (both tile heights)."""
    cin, ld = 96, 256
    cat = bf(rnd(N, ld, H, W))
    w3, b3 = rnd(32, cin, 3, 3, scale=0.1), rnd(32)
    wp = K.conv_pack(w3.cuda(), False, cin, math=K.MATH_BF16)
    catb = to_nhwc_bf16(cat)
    bits = torch.zeros(N, H, W, dtype=torch.int32, device="cuda")
    K.conv_forward(K.Sl(catb, cin, 0), wp, b3.cuda(), K.Sl(catb, 32, cin), 3, relu=True, math=K.MATH_BF16, bits=bits,
                   bits_mode=1)
    act = catb[..., cin:cin + 32].float()
    want = ((act > 0).to(torch.int64) << torch.arange(32, device="cuda")).sum(-1)
    assert torch.equal(bits.to(torch.int64) & 0xFFFFFFFF, want)
    dyb = to_nhwc_bf16(bf(rnd(N, cin, H, W, seed=4)), ld)
    wb = K.conv_pack(rnd(32, cin, 3, 3, scale=0.1, seed=5).cuda(), False, cin, math=K.MATH_BF16)
    out_mask = torch.empty(N, H, W, 32, device="cuda")
    out_bits = torch.empty(N, H, W, 32, device="cuda")
    K.conv_forward(K.Sl(dyb, cin, 0), wb, None, K.Sl(out_mask), 3, mask=K.Sl(catb, 32, cin), mask_c0=0, mask_c1=32,
                   math=K.MATH_BF16)
    K.conv_forward(K.Sl(dyb, cin, 0), wb, None, K.Sl(out_bits), 3, math=K.MATH_BF16, bits=bits, bits_mode=2)
    assert torch.equal(out_mask, out_bits)


@pytest.mark.parametrize("cin,cout,N,H,W", [(96, 128, 1, 20, 37), (128, 64, 2, 9, 33), (192, 64, 1, 17, 40)])
def test_conv_bf16_one_bit_relu_masks_of_wide_layers(K, cin, cout, N, H, W):
    """The same with more than 32 output channels (the flow net's and the attention's hidden layers): cout / 32 words per pixel,
    written by the channel-split 3x3 kernel's two halves (and its channel chunks) and read back by the input-gradient conv
    of the layer behind, whose own output is that wide."""
    nw = cout // 32
    x = to_nhwc_bf16(bf(rnd(N, cin, H, W)))
    w, b = rnd(cout, cin, 3, 3, scale=0.1), rnd(cout)
    y = torch.zeros(N, H, W, cout, device="cuda", dtype=torch.bfloat16)
    bits = torch.zeros(N, H, W, nw, dtype=torch.int32, device="cuda")
    K.conv_forward(K.Sl(x), K.conv_pack(w.cuda(), False, cin, math=K.MATH_BF16), b.cuda(), K.Sl(y), 3, relu=True, math=K.MATH_BF16,
                   bits=bits, bits_mode=1)
    y_plain = torch.zeros_like(y)
    K.conv_forward(K.Sl(x), K.conv_pack(w.cuda(), False, cin, math=K.MATH_BF16), b.cuda(), K.Sl(y_plain), 3, relu=True,
                   math=K.MATH_BF16)
    assert torch.equal(y, y_plain)
    act = (y.float() > 0).to(torch.int64).view(N, H, W, nw, 32)
    want = (act << torch.arange(32, device="cuda")).sum(-1)
    assert torch.equal(bits.to(torch.int64) & 0xFFFFFFFF, want)
    # the input gradient of a following 3x3 conv (cout -> 32), masked by this layer's ReLU: bits vs the activation tensor
    dy = to_nhwc_bf16(bf(rnd(N, 32, H, W, seed=4)))
    wt = K.conv_pack(rnd(32, cout, 3, 3, scale=0.1, seed=5).cuda(), True, 32, cout, math=K.MATH_BF16)
    out_mask = torch.empty(N, H, W, cout, device="cuda", dtype=torch.bfloat16)
    out_bits = torch.empty_like(out_mask)
    K.conv_forward(K.Sl(dy), wt, None, K.Sl(out_mask), 3, mask=K.Sl(y), mask_c0=0, mask_c1=cout, math=K.MATH_BF16)
    K.conv_forward(K.Sl(dy), wt, None, K.Sl(out_bits), 3, math=K.MATH_BF16, bits=bits, bits_mode=2)
    assert torch.equal(out_mask, out_bits)
    with pytest.raises(RuntimeError, match="bit masks"):      # the four-wave 64-channel kernel has no bit path
        K.conv_forward(K.Sl(x), K.conv_pack(w.cuda(), False, cin, math=K.MATH_BF16), b.cuda(), K.Sl(y), 3, relu=True,
                       math=K.MATH_BF16, bits=bits, bits_mode=1, tile_rows=8)


@pytest.mark.parametrize("cin,cout,N,H,W", [(96, 128, 2, 19, 37), (128, 64, 1, 16, 64), (224, 64, 1, 9, 33), (64, 192, 2, 8, 40)])
def test_conv_bf16_wide_layers_on_32x32x16_mfma(K, cin, cout, N, H, W):
    """tile_rows = 264 / 265: the cout >= 64 3x3 conv on v_mfma_f32_32x32x16_bf16 with a wave = 2 rows x 32 pixels x 64 channels /
    as eight channel-split waves (conv_m32w_kernel) against the shipped channel-split kernel and the fp32 conv of the bf16-rounded operands: bias + ReLU with
    one-bit masks written (cout / 32 words per pixel), a bf16 residual, masks read, centre-tap-only leading chunks; partial
    tiles in both directions, one and several 64-channel slabs."""
    nw = cout // 32
    xf = bf(rnd(N, cin, H, W))
    x = to_nhwc_bf16(xf)
    w, b = rnd(cout, cin, 3, 3, scale=0.1), rnd(cout)
    wp = K.conv_pack(w.cuda(), False, cin, math=K.MATH_BF16)
    res = to_nhwc_bf16(bf(rnd(N, cout, H, W, seed=9)))
    ref = F.relu(F.conv2d(xf, bf(w), b, padding=1))
    outs = {}
    for rows in (0, 264, 265):
        y = torch.zeros(N, H, W, cout, device="cuda", dtype=torch.bfloat16)
        bits = torch.zeros(N, H, W, nw, dtype=torch.int32, device="cuda")
        K.conv_forward(K.Sl(x), wp, b.cuda(), K.Sl(y), 3, relu=True, math=K.MATH_BF16, bits=bits, bits_mode=1, tile_rows=rows)
        act = (y.float() > 0).to(torch.int64).view(N, H, W, nw, 32)
        assert torch.equal(bits.to(torch.int64) & 0xFFFFFFFF, (act << torch.arange(32, device="cuda")).sum(-1))
        assert rel(from_nhwc(y.float()), ref) < 5e-3
        y2 = torch.zeros_like(y)                            # residual + alpha, masks read back
        K.conv_forward(K.Sl(x), wp, b.cuda(), K.Sl(y2), 3, alpha=0.2, res=K.Sl(res), math=K.MATH_BF16, tile_rows=rows)
        assert rel(from_nhwc(y2.float()), 0.2 * F.conv2d(xf, bf(w), b, padding=1) + from_nhwc(res.float()).cpu()) < 5e-3
        y3 = torch.zeros_like(y)
        K.conv_forward(K.Sl(x), wp, None, K.Sl(y3), 3, math=K.MATH_BF16, bits=bits, bits_mode=2, tile_rows=rows)
        assert torch.equal(y3 != 0, (y3 != 0) & (y > 0))    # nothing passes where the mask bit is clear
        outs[rows] = (y, y2, y3)
    for v in (264, 265):
        for a, c in zip(outs[0], outs[v]):
            assert rel(a.float(), c.float()) < 5e-3
    if cin >= 96:                                            # centre-tap-only leading chunks (the blocks' input-gradient conv)
        wc = w.clone()
        wc[:, :64, [0, 0, 0, 1, 1, 2, 2, 2], [0, 1, 2, 0, 2, 0, 1, 2]] = 0
        wcp = K.conv_pack(wc.cuda(), False, cin, math=K.MATH_BF16)
        ya, yb = torch.zeros(N, H, W, cout, device="cuda", dtype=torch.bfloat16), torch.zeros(N, H, W, cout, device="cuda", dtype=torch.bfloat16)
        for v in (264, 265):
            K.conv_forward(K.Sl(x), wcp, None, K.Sl(ya), 3, math=K.MATH_BF16, center_cin=64, tile_rows=v)
            K.conv_forward(K.Sl(x), wcp, None, K.Sl(yb), 3, math=K.MATH_BF16, tile_rows=v)
            assert torch.equal(ya, yb) and rel(from_nhwc(ya.float()), F.conv2d(xf, bf(wc), None, padding=1)) < 5e-3


def test_dwconv_bf16_with_fused_batchnorm_input(K):
    """bn=(mean, invstd, gamma, beta, group_images): the depthwise kernels consume relu(bn(x)) evaluated while staging x;
    must equal bn_apply_relu (bf16 output) followed by the plain kernels, bit for bit."""
    C, B, G, H, W = 64, 2, 3, 19, 37
    N = B * G
    p = to_nhwc_bf16(bf(rnd(N, C, H, W) * 1.5 + 0.2))
    w = rnd(C, 1, 3, 3).cuda()
    mean, invstd = (0.3 * rnd(G, C)).cuda(), (1.0 + 0.5 * rnd(G, C, seed=2).abs()).cuda()
    gamma, beta = (1 + 0.2 * rnd(C, seed=3)).cuda(), (0.1 * rnd(C, seed=4)).cuda()
    r = torch.empty(N, H, W, C, device="cuda", dtype=torch.bfloat16)
    K.bn_apply_relu(p, B, mean, invstd, gamma, beta, None, K.Sl(r), N)
    bn = (mean, invstd, gamma, beta, B)
    a, b = torch.empty_like(r), torch.empty_like(r)
    K.dwconv_forward(r, w, a)
    K.dwconv_forward(p, w, b, bn=bn)
    assert torch.equal(a, b)
    dy = to_nhwc_bf16(bf(rnd(N, C, H, W, seed=6)))
    dwa, dwb = torch.empty(C, 1, 3, 3, device="cuda"), torch.empty(C, 1, 3, 3, device="cuda")
    K.dwconv_wgrad(r, dy, dwa, ws_tensor(K))
    K.dwconv_wgrad(p, dy, dwb, ws_tensor(K), bn=bn)
    assert torch.equal(dwa, dwb)


def test_head_gradient_through_matrix_core_wgrad(K):
    """bf16 mode's head weight gradient: head_forward also emits the frames as bf16 NHWC-8, the last depthwise input
    gradient leaves its kernel as (dx + skip) masked by the head features, and nvq_conv_wgrad does the rest."""
    Fc, B, T, H, W = 64, 2, 3, 19, 37
    frames = bf(rnd(B, T, 3, H, W).abs())
    w = rnd(Fc, 3, 3, 3, scale=0.4).requires_grad_()
    b = rnd(Fc, scale=0.1).requires_grad_()
    slots = [1, 0, 2]
    feat = torch.stack([F.relu(F.conv2d(frames[:, t], w, b, padding=1)) for t in slots], 0)   # [slot,B,F,H,W]
    NI = T * B
    feat16 = torch.empty(NI, H, W, Fc, device="cuda", dtype=torch.bfloat16)
    img8 = torch.full((NI, H, W, 8), 7.0, device="cuda", dtype=torch.bfloat16)
    K.head_forward(frames.cuda(), slots, w.detach().cuda(), b.detach().cuda(), feat16, img8=img8)
    want_img = torch.stack([frames[:, t] for t in slots], 0).reshape(NI, 3, H, W)
    assert torch.equal(img8[..., :3].float().cpu(), want_img.permute(0, 2, 3, 1)) and img8[..., 3:].abs().max().item() == 0
    # depthwise input gradient with the add / mask epilogue
    dd = bf(rnd(NI, Fc, H, W, seed=3))
    wd = rnd(Fc, 1, 3, 3)
    skip = rnd(NI, Fc, H, W, seed=4)
    xg = torch.zeros(NI, Fc, H, W, requires_grad=True)
    F.conv2d(xg, wd, None, padding=1, groups=Fc).backward(dd)
    mask_ref = from_nhwc(feat16.float()) > 0
    ghead = (xg.grad + skip) * mask_ref
    dx = torch.empty(NI, H, W, Fc, device="cuda", dtype=torch.bfloat16)
    K.dwconv_forward(to_nhwc_bf16(dd), wd.cuda(), dx, flip=True, add=to_nhwc(skip), mask=feat16)
    assert rel(from_nhwc(dx.float()), ghead) < 5e-3
    # weight / bias gradient of the head conv from (img8, dx)
    gh = from_nhwc(dx.float()).reshape(T, B, Fc, H, W)
    feat.backward(gh * (feat.detach() > 0))          # reference: the same (bf16-rounded) gradient through conv + ReLU
    dw, db = torch.empty(Fc, 3, 3, 3, device="cuda"), torch.empty(Fc, device="cuda")
    K.conv_wgrad(K.Sl(img8), 3, K.Sl(dx), dw, db, ws_tensor(K), 3, math=K.MATH_BF16)
    assert rel(dw, w.grad) < TOL and rel(db, b.grad) < TOL


@pytest.mark.parametrize("N,H,W,out16", [(2, 19, 37, True), (1, 8, 64, False), (3, 41, 70, True)])
def test_rdb_tail_forward_fused(K, N, H, W, out16):
    """nvq_rdb_tail_forward = last dense 3x3 layer (in place, + bit masks) followed by the 1x1 lff with 0.2 scaling and
    residual: same results as the two separate launches."""
    Fc, cin, ld = 64, 192, 256
    cat = bf(rnd(N, ld, H, W))
    w3, b3 = rnd(32, cin, 3, 3, scale=0.1), rnd(32)
    wl, bl = rnd(Fc, cin + 32, 1, 1, scale=0.1, seed=3), rnd(Fc, seed=4)
    w3p = K.conv_pack(w3.cuda(), False, cin, math=K.MATH_BF16)
    wlp = K.conv_pack(wl.cuda(), False, cin + 32, math=K.MATH_BF16)
    odt = torch.bfloat16 if out16 else torch.float32
    a, b = to_nhwc_bf16(cat), to_nhwc_bf16(cat)
    oa = torch.zeros(N, H, W, ld if out16 else Fc, device="cuda", dtype=odt)
    ob = torch.zeros_like(oa)
    bits_a = torch.zeros(N, H, W, dtype=torch.int32, device="cuda")
    bits_b = torch.zeros_like(bits_a)
    K.conv_forward(K.Sl(a, cin, 0), w3p, b3.cuda(), K.Sl(a, 32, cin), 3, relu=True, math=K.MATH_BF16, bits=bits_a, bits_mode=1)
    K.conv_forward(K.Sl(a, cin + 32, 0), wlp, bl.cuda(), K.Sl(oa, Fc, 0), 1, alpha=0.2, res=K.Sl(a, Fc, 0), math=K.MATH_BF16)
    K.rdb_tail_forward(K.Sl(b, cin, 0), w3p, b3.cuda(), K.Sl(b, 32, cin), wlp, bl.cuda(), K.Sl(ob, Fc, 0), alpha=0.2,
                       res=K.Sl(b, Fc, 0), bits=bits_b)
    assert torch.equal(a, b) and torch.equal(bits_a, bits_b)
    # the four-wave kernel (tile_rows = 4) and the eight-wave, two-role one (automatic): the same sums in the same order
    b4, ob4, bits_4 = to_nhwc_bf16(cat), torch.zeros_like(oa), torch.zeros_like(bits_a)
    K.rdb_tail_forward(K.Sl(b4, cin, 0), w3p, b3.cuda(), K.Sl(b4, 32, cin), wlp, bl.cuda(), K.Sl(ob4, Fc, 0), alpha=0.2,
                       res=K.Sl(b4, Fc, 0), bits=bits_4, tile_rows=4)
    assert torch.equal(b4, b) and torch.equal(bits_4, bits_b) and torch.equal(ob4, ob)
    assert rel(ob.float(), oa.float()) < (5e-3 if out16 else 1e-6)
    # and against the fp32 reference of the two ops on the bf16-rounded data
    y4 = F.relu(F.conv2d(cat[:, :cin], bf(w3), b3, padding=1))
    full = torch.cat([cat[:, :cin], bf(y4)], 1)
    ref = 0.2 * F.conv2d(full, bf(wl), bl) + cat[:, :Fc]
    assert rel(from_nhwc(ob.float(), Fc), ref) < 5e-3
