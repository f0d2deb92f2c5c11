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
        q.put((flat, fisher))
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
    flat, fisher = q.get()
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
