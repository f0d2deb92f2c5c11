"""GPU box: cProfile of the host side of the cfg5 step loop (bf16, graph replay, batch 8): the step is host-bound (2.5 ms of
kernels in a 4+ ms step), so this is where its time goes."""
import cProfile, os, pstats, sys, time
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "experiments")); sys.path.insert(0, os.path.join(REPO, "continual-learning-for-dynamic-video-quality-enhancement_amd"))
import torch
import train_continual as TC
from nerve_cl import ops
from nerve_cl.continual import EWC
from nerve_cl.models import EnhancementConfig, EnhancementEngine

dev = torch.device("cuda", 0)
torch.manual_seed(0)
model = EnhancementEngine(EnhancementConfig(frame_recovery_enabled=False, super_resolution_enabled=True)).to(dev)
TC.configure_precision(model, "bf16", "auto")
adapter = TC._ClipAdapter(model)
ewc = EWC(adapter, ewc_lambda=5000)
opt = torch.optim.Adam(model.parameters(), lr=1e-4, **({"fused": True} if os.environ.get("FUSED") else {}))
crit = ops.MSELoss()
lr, hr = TC.create_task_data("sports", 64)
lr, hr = lr.to(dev), hr.to(dev)
loader = [(lr[i:i + 8], hr[i:i + 8]) for i in range(0, 64, 8)]

def step(a, b):
    opt.zero_grad()
    out = model(a.unsqueeze(1).expand(-1, 3, -1, -1, -1))["enhanced"]
    loss = crit(out, b) + ewc.penalty()
    loss.backward()
    opt.step()
    return loss

model.train()
for a, b in loader[:3]: step(a, b)
ewc.register_task(0, loader[:2])
model.train()
for _ in range(3):
    for a, b in loader: step(a, b)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(6):
    for a, b in loader: step(a, b)
torch.cuda.synchronize()
print(f"{(time.perf_counter() - t0) / 48 * 1e3:.2f} ms per step")
# host time per section (no synchronisation inside: what the CPU spends enqueueing)
acc = [0.0] * 5
def step_timed(a, b):
    t = [time.perf_counter()]
    opt.zero_grad(); t.append(time.perf_counter())
    out = model(a.unsqueeze(1).expand(-1, 3, -1, -1, -1))["enhanced"]; t.append(time.perf_counter())
    loss = crit(out, b) + ewc.penalty(); t.append(time.perf_counter())
    loss.backward(); t.append(time.perf_counter())
    opt.step(); t.append(time.perf_counter())
    for i in range(5): acc[i] += t[i + 1] - t[i]
for _ in range(6):
    for a, b in loader: step_timed(a, b)
torch.cuda.synchronize()
print("host ms per step: zero_grad %.2f  forward %.2f  loss+penalty %.2f  backward %.2f  optimizer %.2f" % tuple(v / 48 * 1e3 for v in acc))
# the two graph launches alone (host time of CUDAGraph.replay, device idle in between)
sg = model.super_resolution._step_graphs
for key, e in sg.entries.items():
    if e.fwd is None or e.bwd is None: continue
    for name, g in (("forward", e.fwd), ("backward", e.bwd)):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(20): g.replay()
        t1 = time.perf_counter()
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        print(f"{name} graph: replay() returns after {(t1 - t0) / 20 * 1e3:.2f} ms of host time; 20 replays done after {(t2 - t0) / 20 * 1e3:.2f} ms each")
    t0 = time.perf_counter()
    for _ in range(200): P = model.super_resolution._tensor_dict(); ptrs = tuple(t.data_ptr() for t in P.values())
    print(f"_tensor_dict + address tuple: {(time.perf_counter() - t0) / 200 * 1e3:.3f} ms")
if not os.environ.get("CPROFILE"): sys.exit(0)
pr = cProfile.Profile()
pr.enable()
for _ in range(4):
    for a, b in loader: step(a, b)
torch.cuda.synchronize()
pr.disable()
st = pstats.Stats(pr); st.sort_stats("cumulative"); st.print_stats(45)
