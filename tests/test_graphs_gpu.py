"""GPU: HIP-graph replay of the SuperResolutionNet step (nerve_cl/_graphs.py) gives the results of the eager kernel schedule
bit for bit, and takes the eager path whenever one saved state per shape is not enough."""
import copy

import pytest
import torch
import torch.nn.functional as F

from oracle import synth

pytestmark = pytest.mark.gpu


def make(F_=16, blocks=1, graphs=True, train=True, seed_gain=synth.GOLDEN_GAIN):
    from nerve_cl.models import SuperResolutionNet
    net = SuperResolutionNet(3, 2, F_, blocks, 1)
    net.load_state_dict(synth.formula_state(3, 2, F_, blocks, 1, gain=seed_gain), strict=True)
    net = net.cuda().train(train)
    net.use_hip_graphs = graphs
    return net


def train_steps(net, steps, B=2, H=20, W=24):
    opt = torch.optim.AdamW(net.parameters(), lr=1e-3, weight_decay=1e-5)
    losses = []
    for i in range(steps):
        x = synth.formula_clip(B, 3, H, W, seed=10 + i).cuda()
        y = synth.formula_target(B, 2 * H, 2 * W, seed=50 + i).cuda()
        opt.zero_grad()
        loss = F.mse_loss(net(x), y)
        loss.backward()
        opt.step()
        losses.append(loss.item())
    return losses


@pytest.mark.parametrize("bf16", [False, True])
def test_graph_training_equals_eager_bit_for_bit(bf16):
    """6 AdamW steps (2 eager warm-up calls, capture on the 3rd, replays after): losses, parameters and BatchNorm running
    statistics equal those of the eager schedule exactly - same kernels, same order, same addresses-independent results."""
    from nerve_cl import _nvq
    nets = [make(graphs=g) for g in (True, False)]
    for n in nets:
        if bf16:
            n.math_mode, n.bf16_activations = _nvq.MATH_BF16, True
    la, lb = train_steps(nets[0], 6), train_steps(nets[1], 6)
    assert la == lb
    sa, sb = nets[0].state_dict(), nets[1].state_dict()
    for k in sa:
        assert torch.equal(sa[k], sb[k]), k
    g = nets[0]._step_graphs
    assert g.replays == 4 and g.eager_fallbacks == 0 and nets[1]._step_graphs.replays == 0
    # eval / no_grad is a graph of its own
    x = synth.formula_clip(2, 3, 20, 24, seed=3).cuda()
    outs = []
    for n in nets:
        n.eval()
        with torch.no_grad():
            outs.append([n(x).clone() for _ in range(4)])
    for a, b in zip(*outs):
        assert torch.equal(a, b)
    assert g.replays == 6


def test_graph_outputs_and_gradients_are_copies():
    """p.grad of step k must survive the replay of step k+1 (autograd keeps the tensor it is handed), and so must an output."""
    net = make()
    train_steps(net, 3)                                   # captured
    x1, x2 = synth.formula_clip(2, 3, 20, 24, seed=1).cuda(), synth.formula_clip(2, 3, 20, 24, seed=2).cuda()
    net.zero_grad(set_to_none=True)
    o1 = net(x1)
    o1.sum().backward()
    keep_o, keep_g = o1.clone(), {n: p.grad.clone() for n, p in net.named_parameters()}
    o2 = net(x2)                                          # replays the same graphs
    o2.sum().backward()                                   # accumulates into p.grad
    assert torch.equal(o1, keep_o) and not torch.equal(o1, o2)
    ref = make(graphs=False)
    ref.load_state_dict(net.state_dict())
    ref.zero_grad(set_to_none=True)
    ref(x1).sum().backward()
    for n, p in ref.named_parameters():
        assert torch.equal(p.grad, keep_g[n]), n
    ref(x2).sum().backward()
    for (n, p), (_, q) in zip(net.named_parameters(), ref.named_parameters()):
        assert torch.equal(p.grad, q.grad), n


def test_two_forwards_before_their_backwards_fall_back_to_eager():
    """MAML-style access: forward, forward, backward, backward of one shape.  The second forward must not overwrite the
    state the first backward needs: it runs eagerly, and both gradients equal the eager ones."""
    net, ref = make(), make(graphs=False)
    train_steps(net, 3)
    ref.load_state_dict(net.state_dict())
    xs = [synth.formula_clip(2, 3, 20, 24, seed=s).cuda() for s in (4, 5)]
    got, want = [], []
    for m, sink in ((net, got), (ref, want)):
        outs = [m(x) for x in xs]
        for o in outs:
            m.zero_grad(set_to_none=True)
            o.square().mean().backward()
            sink.append({n: p.grad.clone() for n, p in m.named_parameters()})
    assert net._step_graphs.eager_fallbacks == 1
    for a, b in zip(got, want):
        for n in a:
            assert torch.equal(a[n], b[n]), n


def test_stale_state_raises_and_moved_parameters_recapture():
    net = make()
    train_steps(net, 3)
    x = synth.formula_clip(2, 3, 20, 24, seed=6).cuda()
    o1 = net(x)
    loss1 = o1.mean()
    loss1.backward(retain_graph=True)                     # state consumed, node kept alive
    net(x).mean().backward()                              # replay: overwrites the state of o1's node
    with pytest.raises(RuntimeError, match="overwritten by a later forward"):
        loss1.backward()
    # parameters re-allocated (new storage, as after .to() / .half().float()): the entry is captured again, results stay right
    before = net._step_graphs.replays
    for p in net.parameters():
        p.data = p.data.clone()
    ref = make(graphs=False)
    ref.load_state_dict(net.state_dict())
    net.zero_grad(set_to_none=True)
    ref.zero_grad(set_to_none=True)
    for _ in range(4):                                    # 2 eager warm-ups, capture, replay
        net.zero_grad(set_to_none=True)
        net(x).mean().backward()
    ref(x).mean().backward()
    assert net._step_graphs.replays == before + 2
    for (n, p), (_, q) in zip(net.named_parameters(), ref.named_parameters()):
        assert torch.equal(p.grad, q.grad), n


def test_auto_mode_only_for_launch_bound_shapes():
    net = make(graphs="auto")
    small = torch.empty(8, 3, 3, 64, 64, device="cuda")
    big = torch.empty(1, 3, 3, 540, 960, device="cuda")
    assert net._graphs_wanted(small) and not net._graphs_wanted(big)
