"""GPU: BASELINE cfg2 geometry (540x960 -> 1080x1920, F=64, 8 blocks, T=3).  The CPU oracle cannot run at this size in test
time (~75 GB of autograd state), so these are size-independent properties of the HIP path itself:
batch independence, run-to-run determinism, linearity of the backward in the output gradient, and the bicubic + clamp tail
against torch's own bicubic."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu
H, W = 540, 960


@pytest.fixture(scope="module")
def net():
    if not torch.cuda.is_available():
        pytest.fail("GPU tests need a HIP device")
    from nerve_cl.models import SuperResolutionNet
    torch.manual_seed(3)
    return SuperResolutionNet(3, 2, 64, 8, 1).cuda()


def _clips(b, seed):
    g = torch.Generator(device="cuda").manual_seed(seed)
    return torch.rand(b, 3, 3, H, W, device="cuda", generator=g)


@pytest.mark.parametrize("bf16", [False, True])
def test_batch_independence_and_determinism(net, bf16):
    """eval mode: a clip's output does not depend on what else is in the batch (other tiles, other XCD assignment), and two
    runs give identical bits; train mode backward: identical gradient bits on a second run (no atomics left on the path)."""
    from nerve_cl import _nvq
    net.math_mode, net.bf16_activations = (_nvq.MATH_BF16, True) if bf16 else (_nvq.MATH_F32, False)
    x = _clips(2, 5)
    net.eval()
    with torch.no_grad():
        both = net(x)
        one = net(x[1:2])
        again = net(x)
    assert both.shape == (2, 3, 2 * H, 2 * W)
    assert torch.equal(both[1:2], one)
    assert torch.equal(both, again)
    net.train()
    grads, bufs = [], []
    start = {n: b.clone() for n, b in net.named_buffers()}
    for _ in range(2):
        net.zero_grad(set_to_none=True)
        with torch.no_grad():
            for n, b in net.named_buffers():        # same BatchNorm running state at the start of both runs
                b.copy_(start[n])
        F.mse_loss(net(x[:1]), torch.full((1, 3, 2 * H, 2 * W), 0.5, device="cuda")).backward()
        grads.append(torch.cat([p.grad.flatten() for p in net.parameters()]))
        bufs.append(torch.cat([b.flatten().double() for b in net.buffers()]))
    assert torch.isfinite(grads[0]).all()
    assert torch.equal(grads[0], grads[1])
    assert torch.equal(bufs[0], bufs[1])


@pytest.mark.parametrize("bf16", [False, True])
def test_cfg4_sr_geometry_batch_independence_and_determinism(bf16):
    """BASELINE cfg4's SR network (reference super_resolution.py:279-291 with scale_factor=4, temporal_window=2: T=5, the
    320->64 attention conv, 64->5 logits, 64->48 upsampler) at its full size, 270x480 -> 1080x1920: the same
    size-independent properties as above."""
    from nerve_cl import _nvq
    from nerve_cl.models import SuperResolutionNet
    torch.manual_seed(4)
    net4 = SuperResolutionNet(3, 4, 64, 8, 2).cuda()
    net4.math_mode, net4.bf16_activations = (_nvq.MATH_BF16, True) if bf16 else (_nvq.MATH_F32, False)
    g = torch.Generator(device="cuda").manual_seed(21)
    x = torch.rand(2, 5, 3, 270, 480, device="cuda", generator=g)
    net4.eval()
    with torch.no_grad():
        both = net4(x)
        one = net4(x[1:2])
        again = net4(x)
    assert both.shape == (2, 3, 1080, 1920)
    assert torch.equal(both[1:2], one) and torch.equal(both, again)
    # the skip path alone: zeroed upsampler conv => clamp(bicubic x4 of the centre frame)
    sd = {k: v.clone() for k, v in net4.state_dict().items()}
    with torch.no_grad():
        net4.upsampler.conv.weight.zero_()
        net4.upsampler.conv.bias.zero_()
        ref = F.interpolate(x[:, 2], scale_factor=4.0, mode="bicubic", align_corners=False).clamp(0, 1)
        assert (net4(x) - ref).abs().max().item() < 2e-5
    net4.load_state_dict(sd)
    net4.train()
    start = {n: b.clone() for n, b in net4.named_buffers()}
    grads = []
    for _ in range(2):
        net4.zero_grad(set_to_none=True)
        with torch.no_grad():
            for n, b in net4.named_buffers():
                b.copy_(start[n])
        F.mse_loss(net4(x), torch.full((2, 3, 1080, 1920), 0.5, device="cuda")).backward()
        grads.append(torch.cat([p.grad.flatten() for p in net4.parameters()]))
    assert torch.isfinite(grads[0]).all() and grads[0].abs().max() > 0
    assert torch.equal(grads[0], grads[1])


def test_backward_is_linear_in_the_output_gradient(net):
    """fp32 mode: d(loss)/d(theta) for dout = a + b equals the sum of the two separate backward passes through the same
    forward state (the clamp mask, ReLU masks and softmax are fixed by the forward), to fp32 summation accuracy."""
    from nerve_cl import _nvq
    net.math_mode, net.bf16_activations = _nvq.MATH_F32, False
    net.train()
    net.retain_backward_state = True          # three backward passes through one forward (see _SRFunction.backward)
    x = _clips(1, 9)
    ga = torch.rand(1, 3, 2 * H, 2 * W, device="cuda") - 0.5
    gb = torch.rand(1, 3, 2 * H, 2 * W, device="cuda") - 0.5
    out = net(x)
    params = list(net.parameters())
    da = torch.autograd.grad(out, params, ga, retain_graph=True)
    db = torch.autograd.grad(out, params, gb, retain_graph=True)
    dab = torch.autograd.grad(out, params, ga + gb)
    num = torch.sqrt(sum(((u + v - w).double() ** 2).sum() for u, v, w in zip(da, db, dab)))
    den = torch.sqrt(sum((w.double() ** 2).sum() for w in dab))
    net.retain_backward_state = False
    assert (num / den).item() < 1e-4, (num / den).item()


def test_tail_is_bicubic_plus_residual_clamped(net):
    """With the upsampler conv zeroed the network output is clamp(bicubic(centre frame)): checks the fused pixel-shuffle +
    bicubic + clamp kernel at 1080p against torch's bicubic (an independent implementation of the same operator)."""
    from nerve_cl import _nvq
    net.math_mode, net.bf16_activations = _nvq.MATH_F32, False
    net.eval()
    sd = {k: v.clone() for k, v in net.state_dict().items()}
    try:
        with torch.no_grad():
            net.upsampler.conv.weight.zero_()
            net.upsampler.conv.bias.zero_()
            x = _clips(1, 11)
            out = net(x)
            ref = F.interpolate(x[:, 1], scale_factor=2.0, mode="bicubic", align_corners=False).clamp(0, 1)
        assert (out - ref).abs().max().item() < 2e-5
    finally:
        net.load_state_dict(sd)
