"""Debug helper (GPU box): per-stage gradient error of the HIP backward vs the oracle's autograd."""
import os, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, "continual-learning-for-dynamic-video-quality-enhancement_amd"))
import torch, torch.nn.functional as F
from oracle import sr_oracle, synth
from nerve_cl.models import SuperResolutionNet
from nerve_cl import _engine

Fc, N, win, s, B, H, W = 32, 4, 1, 2, 2, 64, 64
dt = torch.float64 if len(sys.argv) > 1 and sys.argv[1] == "f64" else torch.float32
sd = synth.formula_state(3, s, Fc, N, win, gain=synth.GOLDEN_GAIN)
net = SuperResolutionNet(3, s, Fc, N, win); net.load_state_dict(sd); net = net.cuda().train()
ora = sr_oracle.OracleSR(3, s, Fc, N, win); ora.load_named(sd); ora.train(); ora = ora.to(dt)
x = synth.formula_clip(B, 3, H, W, seed=3); tgt = synth.formula_target(B, H * s, W * s, seed=4)
cap = {}
_engine.DEBUG_CAPTURE = cap
out = net(x.cuda()); F.mse_loss(out, tgt.cuda()).backward()
o_out, inter = ora(x.to(dt), return_intermediate=True)
keep = {"flow0": inter["flows"][0], "flow2": inter["flows"][2], "al0": inter["aligned"][0], "al1": inter["aligned"][1],
        "al2": inter["aligned"][2], "f0": inter["features"][0], "f1": inter["features"][1], "f2": inter["features"][2],
        "agg": inter["aggregated"], "fused": inter["fused"], "res": inter["residual"]}
for v in keep.values(): v.retain_grad()
F.mse_loss(o_out, tgt.to(dt)).backward()
def rel(a, b):
    a, b = a.double().cpu(), b.double().cpu()
    return ((a - b).abs().max() / b.abs().max().clamp_min(1e-30)).item()
def nchw(t, c=None, off=0):
    c = t.shape[-1] - off if c is None else c
    return t[..., off:off + c].permute(0, 3, 1, 2)
print("out", rel(out, o_out))
print("dres", rel(nchw(cap["dres"]), keep["res"].grad))
print("dagg", rel(nchw(cap["dagg"]), keep["agg"].grad))
da = cap["daligned"]
for t in range(3):
    print(f"daligned[{t}]", rel(nchw(da, Fc, t * Fc), keep[f"al{t}"].grad) if t != 1 else "(centre: shared leaf)")
df = cap["dflow"]
print("dflow t=0", rel(nchw(df[:B], 2), keep["flow0"].grad), " t=2", rel(nchw(df[B:], 2), keep["flow2"].grad))
dfa = cap["dfeat_all"]
print("dfeat centre", rel(nchw(dfa[:B]), keep["f1"].grad), " t0", rel(nchw(dfa[B:2 * B]), keep["f0"].grad), " t2", rel(nchw(dfa[2 * B:]), keep["f2"].grad))
on = ora.named()
for n, p in net.named_parameters():
    e = rel(p.grad, on[n].grad)
    if e > 1e-4: print(f"  {n}: {e:.2e}")

# --- ReLU mask agreement of the flow net's first layer
sv_f1 = None
import nerve_cl._engine as E
P = {n: p.detach().to(dt).cpu() for n, p in net.named_parameters()}
with torch.no_grad():
    feats = inter["features"]
    for j, t in enumerate((0, 2)):
        corr = sr_oracle.correlation(feats[t], feats[1])
        pre = F.conv2d(corr, ora.named()["motion_estimator.flow_net.0.weight"], ora.named()["motion_estimator.flow_net.0.bias"], padding=1)
        hip_f1 = cap["f1"][j * B:(j + 1) * B].permute(0, 3, 1, 2).cpu()
        dis = ((pre > 0) != (hip_f1 > 0))
        print(f"t={t}: f1 mask disagreements {int(dis.sum())} of {dis.numel()}; |pre| at those:", pre[dis].abs().tolist()[:8],
              " min |pre| overall", pre.abs().min().item())
