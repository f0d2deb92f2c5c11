"""Data parallelism through the HIP path: two ranks share the box's one GPU (gloo transport; RCCL wants one
device per rank) and run tests/dp_worker.py; the parent then reproduces every quantity in a single process on the same GPU
at loader batch = per-rank batch, which is what SURVEY.md 8e defines as the parity target."""
import os
import subprocess
import sys

import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))


def _free_port() -> int:
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


@pytest.mark.timeout(600)
def test_two_ranks_bucket_mean_fisher_and_fused_penalty(tmp_path):
    out = tmp_path / "dp.pt"
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", OMP_NUM_THREADS="2")
    # children are ordinary child processes of a launcher that never touches the GPU
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2",
                        "--master-addr", "127.0.0.1", "--master-port", str(_free_port()),
                        os.path.join(HERE, "dp_worker.py"), str(out)], env=env, capture_output=True, text=True, timeout=540)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    got = torch.load(out, weights_only=True)

    sys.path.insert(0, HERE)
    import dp_worker as W
    from nerve_cl.continual import EWC
    dev = torch.device("cuda", 0)
    x, y = W.data(2)
    x, y = x.to(dev), y.to(dev)
    b = W.B_PER_RANK
    state0 = {k: v.clone() for k, v in W.make_engine().state_dict().items()}

    def fresh():
        eng = W.make_engine()
        eng.load_state_dict(state0)
        return eng.to(dev)

    # step 1: mean over the two micro-batches of the train-mode gradient bucket (each with its own BatchNorm statistics)
    buckets = []
    for r_ in range(2):
        eng = fresh().train()
        F.mse_loss(eng(x[b * r_:b * r_ + b])["enhanced"], y[b * r_:b * r_ + b]).backward()
        buckets.append(eng.super_resolution._last_grad_bucket.clone())
    want1 = (buckets[0] + buckets[1]) / 2
    assert (got["bucket1"].to(dev) - want1).abs().max() <= 1e-5 * want1.abs().max()

    # Fisher: single process over the union of the ranks' batches (batch size 1), after the SAME step-1 BN updates as rank 0
    eng = fresh().train()
    F.mse_loss(eng(x[:b])["enhanced"], y[:b]).backward()       # rank 0's step 1 (moves its BatchNorm running statistics)
    model = W.Adapter(eng)
    # the other rank's replica saw different running statistics in step 1: evaluate its batches with ITS buffers
    eng1 = fresh().train()
    F.mse_loss(eng1(x[b:2 * b])["enhanced"], y[b:2 * b]).backward()
    f0 = EWC(model, ewc_lambda=W.LAM).compute_fisher([(x[k:k + 1], y[k:k + 1]) for k in range(b)])
    f1 = EWC(W.Adapter(eng1), ewc_lambda=W.LAM).compute_fisher([(x[k:k + 1], y[k:k + 1]) for k in range(b, 2 * b)])
    names = [n for n, _ in model.named_parameters()]
    want_f = torch.cat([((f0[n] * b + f1[n] * b) / (2 * b)).reshape(-1) for n in names])
    assert want_f.abs().max() > 0
    assert (got["fisher"].to(dev) - want_f).abs().max() <= 1e-5 * want_f.abs().max()

    # step 2: rank-mean data gradient + lambda * F * (theta - theta*) with the all-reduced Fisher
    sr_names = [n for n, _ in eng.super_resolution.named_parameters()]
    star = torch.cat([p.detach().reshape(-1) for p in eng.super_resolution.parameters()])
    gsum = None
    for r_, e in enumerate((eng, eng1)):
        W.perturb_(e)
        e.train()
        e.zero_grad()
        F.mse_loss(e(x[b * r_:b * r_ + b])["enhanced"], y[b * r_:b * r_ + b]).backward()
        g = torch.cat([p.grad.reshape(-1) for p in e.super_resolution.parameters()])
        gsum = g if gsum is None else gsum + g
    theta = torch.cat([p.detach().reshape(-1) for p in eng.super_resolution.parameters()])
    fish_sr = torch.cat([((f0["engine.super_resolution." + n] + f1["engine.super_resolution." + n]) / 2).reshape(-1)
                         for n in sr_names])
    want2 = gsum / 2 + W.LAM * fish_sr * (theta - star)
    pen_part = (W.LAM * fish_sr * (theta - star)).abs().max()
    assert pen_part > 1e-3 * want2.abs().max()                 # the penalty term is visible in the comparison
    assert (got["grads2"].to(dev) - want2).abs().max() <= 2e-5 * want2.abs().max()
