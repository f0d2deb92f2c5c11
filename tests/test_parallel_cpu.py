"""CPU, world_size 2 over gloo: the data-parallel plumbing of nerve_cl.parallel (bucket all-reduce, state
broadcast, Fisher all-reduce).  Gradients come from the CPU oracle (test infrastructure) because the product
kernels need a GPU; what is under test is the collective logic that bench.py / the scripts run over RCCL."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp
import torch.nn.functional as F

from oracle import sr_oracle, synth

CFG = dict(scale_factor=2, num_features=16, num_residual_blocks=1, temporal_window=1)


def _free_port() -> int:
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _get(q, procs, timeout: float = 240.0):
    """q.get() that fails instead of hanging when a worker died (the payload must be read BEFORE the producer exits: its
    tensors travel as shared-memory handles)"""
    import time
    t0 = time.time()
    while q.empty():
        dead = [p.exitcode for p in procs if p.exitcode not in (None, 0)]
        assert not dead, f"worker exited with {dead}"
        assert time.time() - t0 < timeout, "workers timed out"
        time.sleep(0.05)
    return q.get()


def _flat_grads(model, x, y):
    model.zero_grad()
    F.mse_loss(model(x), y).backward()
    return torch.cat([p.grad.reshape(-1) for p in model.parameters()])


def _worker(rank: int, world: int, port: int, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    torch.set_num_threads(2)
    from nerve_cl import parallel
    r, w, _ = parallel.init_from_env("gloo")
    assert (r, w) == (rank, world)
    sd = synth.formula_state(3, 2, 16, 1, 1, gain=synth.GOLDEN_GAIN)
    model = sr_oracle.OracleSR(3, 2, 16, 1, 1)
    model.load_named(sd)
    if rank == 1:                                   # replicas start different; broadcast must fix that
        with torch.no_grad():
            for p in model.parameters():
                p.add_(1.0)
    parallel.broadcast_state_(model, 0)
    for n, p in model.named().items():
        assert torch.equal(p.detach(), sd[n].to(p.dtype)), n
    model.eval()                                    # running statistics: per-sample independence
    x = synth.formula_clip(4, 3, 8, 10)
    y = synth.formula_target(4, 16, 20)
    mine = slice(2 * rank, 2 * rank + 2)
    flat = _flat_grads(model, x[mine], y[mine])
    parallel.allreduce_mean_(flat)
    # Fisher: per-rank sum of squared batch gradients, all-reduced (sum) and divided by the global count
    fisher = torch.zeros_like(flat)
    for k in range(2):
        g = _flat_grads(model, x[mine][k:k + 1], y[mine][k:k + 1])
        fisher += g * g
    parallel.allreduce_sum_(fisher)
    fisher /= 4
    if rank == 0:
        q.put((flat.numpy(), fisher.numpy()))          # by value: shared-memory handles die with the producer
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_world2_bucket_allreduce_broadcast_and_fisher():
    ctx = mp.get_context("spawn")
    q = ctx.SimpleQueue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    flat, fisher = [torch.from_numpy(a) for a in _get(q, procs)]
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    # single-process oracle on the full batch (eval mode => mean of shard gradients == full-batch gradient)
    sd = synth.formula_state(3, 2, 16, 1, 1, gain=synth.GOLDEN_GAIN)
    model = sr_oracle.OracleSR(3, 2, 16, 1, 1)
    model.load_named(sd)
    model.eval()
    x = synth.formula_clip(4, 3, 8, 10)
    y = synth.formula_target(4, 16, 20)
    want = _flat_grads(model, x, y)
    assert (flat - want).abs().max() <= 1e-5 * want.abs().max()
    batches = [(x[k:k + 1], y[k:k + 1]) for k in range(4)]
    f_ref = sr_oracle.ewc_fisher(model, batches)    # the reference's definition at loader batch = per-rank batch
    f_ref = torch.cat([f_ref[n].reshape(-1) for n, _ in model.named_parameters()])
    assert (fisher - f_ref).abs().max() <= 1e-5 * f_ref.abs().max()


def test_single_process_helpers_are_noops():
    from nerve_cl import parallel
    t = torch.arange(6.0)
    assert parallel.world_size() == 1
    assert torch.equal(parallel.allreduce_mean_(t.clone()), t)
    assert torch.equal(parallel.allreduce_sum_(t.clone()), t)
    a, b = torch.zeros(2, 3), torch.zeros(4)
    parallel.unflatten_into_(torch.arange(10.0), [a, b])
    assert torch.equal(parallel.flatten([a, b]), torch.arange(10.0))
    assert parallel.shard if hasattr(parallel, "shard") else True


def test_enable_data_parallel_installs_bucket_hook():
    from nerve_cl import parallel
    from nerve_cl.models import EnhancementConfig, EnhancementEngine
    eng = EnhancementEngine(EnhancementConfig(frame_recovery_enabled=False, sr_num_features=16,
                                              sr_num_residual_blocks=1))
    assert eng.super_resolution._grad_bucket_hook is None
    parallel.enable_data_parallel(eng, broadcast=False)
    hook = eng.super_resolution._grad_bucket_hook
    assert callable(hook)
    t = torch.ones(8)
    hook(t)                                          # world size 1: identity
    assert torch.equal(t, torch.ones(8))


# ----------------------------------------------------------------------------- EWC + data parallel on a toy bucketed net
# The product's Fisher / penalty arithmetic are libnvq kernels (GPU only).  What is checked here is the HOST logic
# around them - the gradient hook being switched off inside compute_fisher, the packed Fisher all-reduce, the deferred
# penalty gradient applied AFTER the bucket all-reduce - so the three kernels are replaced by their torch formulas (in this
# test only) and the network is a two-tensor linear model with a hand-written bucket backward.
def _install_cpu_kernels():
    from nerve_cl import _engine, _nvq, ops
    from nerve_cl.continual import ewc as ewc_mod
    ops.mse_loss = F.mse_loss
    _nvq.fisher_accumulate = lambda g, f: f.add_(g * g)
    _nvq.ewc_penalty = lambda th, st, fi, lam, out, ws: out.copy_((0.5 * lam * (fi * (th - st) ** 2).sum()).reshape(1))

    def pen_grad(th, st, fi, lam, scale, grad, acc):
        g = lam * fi * (th - st) * scale
        grad.add_(g) if acc else grad.copy_(g)
    _nvq.ewc_penalty_grad = pen_grad
    _engine.workspace = lambda dev: None
    ewc_mod.EWC._device = lambda self: next(self.model.parameters()).device


def _toy_net():
    from nerve_cl._bucket import BucketedNet

    class _ToyFn(torch.autograd.Function):
        @staticmethod
        def forward(ctx, net, x, w, b):
            ctx.net, ctx.x = net, x
            net._mark_awaiting(ctx)
            return x @ w.t() + b

        @staticmethod
        def backward(ctx, dy):
            net = ctx.net
            flat, views = net._new_grad_bucket()
            views["lin.weight"].copy_(dy.t() @ ctx.x)
            views["lin.bias"].copy_(dy.sum(0))
            net._finish_bucket(flat)
            return None, None, views["lin.weight"], views["lin.bias"]

    class Toy(BucketedNet):
        def __init__(self):
            super().__init__()
            self.lin = torch.nn.Linear(3, 3)
            self._init_bucket()

        def forward(self, x):
            return _ToyFn.apply(self, x, self.lin.weight, self.lin.bias)

    return Toy()


class _Wrap(torch.nn.Module):
    """bucketed net + one loose parameter (like EnhancementEngine.enhancement_strength)"""

    def __init__(self, net):
        super().__init__()
        self.net = net
        self.loose = torch.nn.Parameter(torch.ones(1))

    def forward(self, x):
        return self.net(x) * self.loose


def _toy_data():
    g = torch.Generator().manual_seed(7)
    return torch.randn(8, 3, generator=g), torch.randn(8, 3, generator=g)


def _toy_reference(lam):
    """single-process, plain autograd: Fisher over the union of the ranks' batches (batch size 2), then the gradient of
    mean-over-ranks(mse) + penalty at perturbed parameters"""
    torch.manual_seed(3)
    lin = torch.nn.Linear(3, 3)
    loose = torch.nn.Parameter(torch.ones(1))
    params = [lin.weight, lin.bias, loose]
    x, y = _toy_data()
    fisher = [torch.zeros_like(p) for p in params]
    for k in range(0, 8, 2):
        gs = torch.autograd.grad(F.mse_loss(lin(x[k:k + 2]) * loose, y[k:k + 2]), params)
        for f, g in zip(fisher, gs):
            f += g * g
    fisher = [f / 8 for f in fisher]
    star = [p.detach().clone() for p in params]
    with torch.no_grad():
        for p in params:
            p.add_(0.05)
    loss = 0.5 * (F.mse_loss(lin(x[:4]) * loose, y[:4]) + F.mse_loss(lin(x[4:]) * loose, y[4:]))
    pen = sum((lam / 2 * f * (p - s) ** 2).sum() for f, p, s in zip(fisher, params, star))
    gs = torch.autograd.grad(loss + pen, params)
    return fisher, gs


def _ewc_worker(rank: int, world: int, port: int, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    torch.set_num_threads(1)
    from nerve_cl import parallel
    from nerve_cl.continual import EWC
    _install_cpu_kernels()
    parallel.init_from_env("gloo")
    torch.manual_seed(3 + 10 * rank)                    # replicas start different: enable_data_parallel broadcasts rank 0
    model = _Wrap(_toy_net())
    if rank == 0:
        torch.manual_seed(3)
        model.net.lin.reset_parameters()
    parallel.enable_data_parallel(model)
    x, y = _toy_data()
    mine = slice(4 * rank, 4 * rank + 4)
    # rank 1 sees its two batches and then an EMPTY extra iteration budget (num_samples) - different loop lengths are fine
    loader = [(x[mine][k:k + 2], y[mine][k:k + 2]) for k in (0, 2)]
    ewc = EWC(model, ewc_lambda=50.0)
    ewc.register_task(0, loader)
    assert callable(model.net._grad_bucket_hook)        # restored
    fisher = [ewc.fisher_dict[n].clone() for n in ("net.lin.weight", "net.lin.bias", "loose")]
    with torch.no_grad():
        for p in model.parameters():
            p.add_(0.05)
    model.zero_grad()
    loss = F.mse_loss(model(x[mine]), y[mine]) + ewc.penalty()
    loss.backward()
    assert model.net._deferred_adds == []               # drained by the net's backward
    # the loose parameter is outside the bucket: its data gradient is averaged here by hand, the penalty part is local
    gl = model.loose.grad.clone()
    pen_l = 50.0 * ewc.fisher_dict["loose"] * (model.loose.detach() - ewc.optpar_dict["loose"])
    data_l = gl - pen_l
    dist.all_reduce(data_l)
    gl = data_l / world + pen_l
    grads = [model.net.lin.weight.grad.clone(), model.net.lin.bias.grad.clone(), gl]
    if rank == 0:
        q.put(([f.numpy() for f in fisher], [g.numpy() for g in grads]))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_world2_ewc_fisher_and_deferred_penalty_through_the_bucket_hook():
    ctx = mp.get_context("spawn")
    q = ctx.SimpleQueue()
    port = _free_port()
    procs = [ctx.Process(target=_ewc_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    fisher, grads = [[torch.from_numpy(a) for a in part] for part in _get(q, procs)]
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    f_ref, g_ref = _toy_reference(50.0)
    for a, b in zip(fisher, f_ref):
        assert (a - b).abs().max() <= 1e-6 * b.abs().max().clamp_min(1e-12), (a, b)
    for a, b in zip(grads, g_ref):
        assert (a - b).abs().max() <= 1e-5 * b.abs().max().clamp_min(1e-12), (a, b)
