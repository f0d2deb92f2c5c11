"""CPU oracle for the NERVE-CL super-resolution hot path.

TEST INFRASTRUCTURE ONLY.  Nothing under ``oracle/`` is part of the product:
only ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of
``bench.py`` may import it, and only as the checker / the timed CPU baseline.
The product path (``nerve_cl`` + ``libnvq.so``) never routes through here.

This file restates, as plain functions over a flat ``{state_dict name: tensor}``
dictionary, the floating-point algorithm of the reference's
``nerve_cl.models.SuperResolutionNet`` forward pass (autograd supplies the
backward) and of ``nerve_cl.continual.EWC``.  Each function cites the reference
lines it follows (paths relative to the reference checkout).

Parity pin: ``oracle/make_goldens.py`` imports the reference in the build
container, drives both it and this restatement with identical formula-generated
weights/inputs and writes ``tests/golden/*.npz``; ``tests/test_oracle_golden.py``
checks this file against those fixtures on every run (the reference's own tests
hold shapes only).
"""
from __future__ import annotations

import math
from typing import Dict, List, Optional, Tuple

import torch
import torch.nn.functional as F

Params = Dict[str, torch.Tensor]

RDB_GROWTH = 32      # fixed in the reference: super_resolution.py:227
RDB_LAYERS = 5       # fixed in the reference: super_resolution.py:228
CORR_DISP = 4        # super_resolution.py:70
BN_EPS = 1e-5        # nn.BatchNorm2d default (efficient_layers.py:59)
BN_MOMENTUM = 0.1


# --------------------------------------------------------------------------
# parameter inventory (measured from the reference)
# --------------------------------------------------------------------------
def param_shapes(in_channels: int = 3, scale_factor: int = 2, num_features: int = 64,
                 num_residual_blocks: int = 8, temporal_window: int = 1
                 ) -> "Dict[str, Tuple[int, ...]]":
    """Ordered {name: shape} of every trainable tensor of the SR net.

    Order equals ``named_parameters()`` order of the reference module
    (super_resolution.py:294-318)."""
    Fc, T = num_features, 2 * temporal_window + 1
    s: Dict[str, Tuple[int, ...]] = {}
    s["feature_extractor.head.0.weight"] = (Fc, in_channels, 3, 3)
    s["feature_extractor.head.0.bias"] = (Fc,)
    for k in range(3):
        b = f"feature_extractor.body.{k}."
        s[b + "depthwise.weight"] = (Fc, 1, 3, 3)
        s[b + "pointwise.weight"] = (Fc, Fc, 1, 1)
        s[b + "bn.weight"] = (Fc,)
        s[b + "bn.bias"] = (Fc,)
    chans = [(2 * CORR_DISP + 1) ** 2, 128, 64, 32, 2]
    for li, idx in enumerate((0, 2, 4, 6)):
        s[f"motion_estimator.flow_net.{idx}.weight"] = (chans[li + 1], chans[li], 3, 3)
        s[f"motion_estimator.flow_net.{idx}.bias"] = (chans[li + 1],)
    achans = [Fc * T, Fc, Fc, T]
    for li, idx in enumerate((0, 2, 4)):
        s[f"temporal_aggregator.attention.{idx}.weight"] = (achans[li + 1], achans[li], 3, 3)
        s[f"temporal_aggregator.attention.{idx}.bias"] = (achans[li + 1],)
    s["temporal_aggregator.refine.channel_attention.fc.0.weight"] = (Fc // 16, Fc)
    s["temporal_aggregator.refine.channel_attention.fc.2.weight"] = (Fc, Fc // 16)
    s["temporal_aggregator.refine.spatial_attention.conv.weight"] = (1, 2, 7, 7)
    for k in range(num_residual_blocks):
        for i in range(RDB_LAYERS):
            s[f"residual_blocks.{k}.layers.{i}.0.weight"] = (RDB_GROWTH, Fc + RDB_GROWTH * i, 3, 3)
            s[f"residual_blocks.{k}.layers.{i}.0.bias"] = (RDB_GROWTH,)
        s[f"residual_blocks.{k}.lff.weight"] = (Fc, Fc + RDB_GROWTH * RDB_LAYERS, 1, 1)
        s[f"residual_blocks.{k}.lff.bias"] = (Fc,)
    s["gff.0.weight"] = (Fc, Fc, 3, 3)
    s["gff.0.bias"] = (Fc,)
    s["upsampler.conv.weight"] = (in_channels * scale_factor ** 2, Fc, 3, 3)
    s["upsampler.conv.bias"] = (in_channels * scale_factor ** 2,)
    return s


def buffer_shapes(num_features: int = 64) -> "Dict[str, Tuple[int, ...]]":
    s: Dict[str, Tuple[int, ...]] = {}
    for k in range(3):
        b = f"feature_extractor.body.{k}.bn."
        s[b + "running_mean"] = (num_features,)
        s[b + "running_var"] = (num_features,)
        s[b + "num_batches_tracked"] = ()
    return s


# --------------------------------------------------------------------------
# individual stages
# --------------------------------------------------------------------------
def batch_norm_call(x: torch.Tensor, P: Params, prefix: str, training: bool) -> torch.Tensor:
    """nn.BatchNorm2d semantics (efficient_layers.py:59,65).

    train: normalise with the biased batch variance over (B,H,W); update
    running_mean / running_var (unbiased) with momentum 0.1 and bump
    num_batches_tracked.  eval: normalise with the running statistics."""
    g, b = P[prefix + "weight"], P[prefix + "bias"]
    rm, rv = P[prefix + "running_mean"], P[prefix + "running_var"]
    if training:
        n = x.numel() // x.shape[1]
        mean = x.mean(dim=(0, 2, 3))
        var = x.var(dim=(0, 2, 3), unbiased=False)
        with torch.no_grad():
            rm.mul_(1 - BN_MOMENTUM).add_(BN_MOMENTUM * mean.detach())
            rv.mul_(1 - BN_MOMENTUM).add_(BN_MOMENTUM * var.detach() * (n / max(n - 1, 1)))
            P[prefix + "num_batches_tracked"] += 1
    else:
        mean, var = rm, rv
    inv = torch.rsqrt(var + BN_EPS)
    return (x - mean[None, :, None, None]) * (inv * g)[None, :, None, None] + b[None, :, None, None]


def feature_extractor(P: Params, frame: torch.Tensor, training: bool) -> torch.Tensor:
    """FeatureExtractor.forward, super_resolution.py:40-53; DepthwiseSeparableConv
    efficient_layers.py:62-67."""
    h = F.relu(F.conv2d(frame, P["feature_extractor.head.0.weight"],
                        P["feature_extractor.head.0.bias"], padding=1))
    y = h
    C = h.shape[1]
    for k in range(3):
        b = f"feature_extractor.body.{k}."
        y = F.conv2d(y, P[b + "depthwise.weight"], None, padding=1, groups=C)
        y = F.conv2d(y, P[b + "pointwise.weight"], None)
        y = F.relu(batch_norm_call(y, P, b + "bn.", training))
    return y + h


def correlation(x1: torch.Tensor, x2: torch.Tensor, d: int = CORR_DISP) -> torch.Tensor:
    """LiteFlowNetCorrelation.forward, efficient_layers.py:313-343.

    out[b, i*(2d+1)+j, y, x] = mean_c x1[b,c,y,x] * pad(x2)[b,c,y+i,x+j]."""
    B, C, H, W = x1.shape
    n = 2 * d + 1
    x2p = F.pad(x2, [d, d, d, d])
    planes = []
    for i in range(n):
        for j in range(n):
            planes.append((x1 * x2p[:, :, i:i + H, j:j + W]).sum(dim=1))
    return torch.stack(planes, dim=1) / C


def flow_net(P: Params, corr: torch.Tensor) -> torch.Tensor:
    """MotionEstimator.flow_net, super_resolution.py:74-82."""
    y = corr
    for idx in (0, 2, 4, 6):
        y = F.conv2d(y, P[f"motion_estimator.flow_net.{idx}.weight"],
                     P[f"motion_estimator.flow_net.{idx}.bias"], padding=1)
        if idx != 6:
            y = F.relu(y)
    return y


def warp(feat: torch.Tensor, flow: torch.Tensor) -> torch.Tensor:
    """warp_features, super_resolution.py:104-143: bilinear sample of ``feat`` at
    pixel coordinates (x+flow_x, y+flow_y), zeros outside, align_corners=True."""
    B, C, H, W = feat.shape
    ys = torch.arange(H, dtype=feat.dtype, device=feat.device)
    xs = torch.arange(W, dtype=feat.dtype, device=feat.device)
    gx = xs[None, None, :] + flow[:, 0]
    gy = ys[None, :, None] + flow[:, 1]
    gxn = 2.0 * gx / (W - 1) - 1.0
    gyn = 2.0 * gy / (H - 1) - 1.0
    grid = torch.stack([gxn, gyn], dim=-1)
    return F.grid_sample(feat, grid, mode="bilinear", padding_mode="zeros", align_corners=True)


def cbam(P: Params, x: torch.Tensor) -> torch.Tensor:
    """CBAM = ChannelAttention then SpatialAttention, efficient_layers.py:176-180,
    200-205, 225-228."""
    pre = "temporal_aggregator.refine."
    gap = x.mean(dim=(2, 3))
    hid = F.relu(gap @ P[pre + "channel_attention.fc.0.weight"].t())
    ca = torch.sigmoid(hid @ P[pre + "channel_attention.fc.2.weight"].t())
    xc = x * ca[:, :, None, None]
    sm = torch.cat([xc.mean(dim=1, keepdim=True), xc.max(dim=1, keepdim=True)[0]], dim=1)
    sa = torch.sigmoid(F.conv2d(sm, P[pre + "spatial_attention.conv.weight"], None, padding=3))
    return xc * sa


def temporal_aggregator(P: Params, aligned: List[torch.Tensor]) -> torch.Tensor:
    """TemporalAggregator.forward, super_resolution.py:180-209."""
    stacked = torch.stack(aligned, dim=1)
    B, T, C, H, W = stacked.shape
    y = stacked.reshape(B, T * C, H, W)
    for idx in (0, 2, 4):
        y = F.conv2d(y, P[f"temporal_aggregator.attention.{idx}.weight"],
                     P[f"temporal_aggregator.attention.{idx}.bias"], padding=1)
        if idx != 4:
            y = F.relu(y)
    attn = torch.softmax(y, dim=1)
    weighted = (stacked * attn[:, :, None]).sum(dim=1)
    return cbam(P, weighted)


def residual_dense_block(P: Params, k: int, x: torch.Tensor) -> torch.Tensor:
    """ResidualDenseBlock.forward, super_resolution.py:245-253."""
    feats = [x]
    for i in range(RDB_LAYERS):
        y = F.conv2d(torch.cat(feats, dim=1), P[f"residual_blocks.{k}.layers.{i}.0.weight"],
                     P[f"residual_blocks.{k}.layers.{i}.0.bias"], padding=1)
        feats.append(F.relu(y))
    y = F.conv2d(torch.cat(feats, dim=1), P[f"residual_blocks.{k}.lff.weight"],
                 P[f"residual_blocks.{k}.lff.bias"])
    return y * 0.2 + x


def bicubic_up(x: torch.Tensor, s: int) -> torch.Tensor:
    """nn.Upsample(scale_factor=s, mode='bicubic', align_corners=False),
    super_resolution.py:321-325."""
    return F.interpolate(x, scale_factor=float(s), mode="bicubic", align_corners=False)


LIGHT_F = 32
LIGHT_BLOCKS = (2, 3, 4, 5)


def light_param_shapes(scale_factor: int = 2) -> Dict[str, tuple]:
    """state_dict parameter names / shapes of LightweightSuperResolution, super_resolution.py:450-459."""
    Fc = LIGHT_F
    sh = {"net.0.weight": (Fc, 3, 3, 3), "net.0.bias": (Fc,)}
    for k in LIGHT_BLOCKS:
        sh[f"net.{k}.depthwise.weight"] = (Fc, 1, 3, 3)
        sh[f"net.{k}.pointwise.weight"] = (Fc, Fc, 1, 1)
        sh[f"net.{k}.bn.weight"] = (Fc,)
        sh[f"net.{k}.bn.bias"] = (Fc,)
    sh["net.6.weight"] = (3 * scale_factor ** 2, Fc, 3, 3)
    sh["net.6.bias"] = (3 * scale_factor ** 2,)
    return sh


def light_buffer_shapes() -> Dict[str, tuple]:
    sh = {}
    for k in LIGHT_BLOCKS:
        sh[f"net.{k}.bn.running_mean"] = (LIGHT_F,)
        sh[f"net.{k}.bn.running_var"] = (LIGHT_F,)
        sh[f"net.{k}.bn.num_batches_tracked"] = ()
    return sh


def light_forward(P: Params, x: torch.Tensor, training: bool = True) -> torch.Tensor:
    """LightweightSuperResolution.forward, super_resolution.py:467-470 (net: :450-459)."""
    y = F.relu(F.conv2d(x, P["net.0.weight"], P["net.0.bias"], padding=1))
    for k in LIGHT_BLOCKS:
        b = f"net.{k}."
        y = F.conv2d(y, P[b + "depthwise.weight"], None, padding=1, groups=y.shape[1])
        y = F.conv2d(y, P[b + "pointwise.weight"], None)
        y = F.relu(batch_norm_call(y, P, b + "bn.", training))
    y = F.conv2d(y, P["net.6.weight"], P["net.6.bias"], padding=1)
    s = int(round(math.sqrt(P["net.6.weight"].shape[0] / 3)))
    return torch.clamp(bicubic_up(x, s) + F.pixel_shuffle(y, s), 0, 1)


def num_blocks(P: Params) -> int:
    k = 0
    while f"residual_blocks.{k}.lff.weight" in P:
        k += 1
    return k


def scale_of(P: Params, in_channels: int = 3) -> int:
    return int(round(math.sqrt(P["upsampler.conv.weight"].shape[0] / in_channels)))


# --------------------------------------------------------------------------
# whole forward (super_resolution.py:327-391)
# --------------------------------------------------------------------------
def sr_forward(P: Params, lr_frames: torch.Tensor, training: bool = True,
               return_intermediate: bool = False):
    B, T, C, H, W = lr_frames.shape
    c = T // 2
    s = scale_of(P, C)
    feats = [feature_extractor(P, lr_frames[:, t], training) for t in range(T)]
    center = feats[c]
    aligned, flows = [], {}
    for t in range(T):
        if t == c:
            aligned.append(center)
            continue
        fl = flow_net(P, correlation(feats[t], center))
        flows[t] = fl
        aligned.append(warp(feats[t], fl))
    agg = temporal_aggregator(P, aligned)
    y = agg
    for k in range(num_blocks(P)):
        y = residual_dense_block(P, k, y)
    fused = F.relu(F.conv2d(y, P["gff.0.weight"], P["gff.0.bias"], padding=1)) + center
    up = F.pixel_shuffle(F.conv2d(fused, P["upsampler.conv.weight"],
                                  P["upsampler.conv.bias"], padding=1), s)
    out = torch.clamp(bicubic_up(lr_frames[:, c], s) + up, 0, 1)
    if return_intermediate:
        return out, {"features": feats, "aligned": aligned, "aggregated": agg,
                     "flows": flows, "residual": y, "fused": fused}
    return out


def compute_psnr(pred: torch.Tensor, target: torch.Tensor) -> float:
    """experiments/train_baseline.py:27-32."""
    mse = torch.mean((pred - target) ** 2)
    if mse == 0:
        return float("inf")
    return 20 * torch.log10(1.0 / torch.sqrt(mse)).item()


# --------------------------------------------------------------------------
# module wrapper so optimisers / state_dict work on the oracle
# --------------------------------------------------------------------------
class OracleSR(torch.nn.Module):
    """Holds the tensors under the reference's state_dict names (dots replaced
    internally) and evaluates :func:`sr_forward`.  Default initialisation is NOT
    reproduced here; tests load explicit weights."""

    def __init__(self, in_channels=3, scale_factor=2, num_features=64,
                 num_residual_blocks=8, temporal_window=1):
        super().__init__()
        self.scale_factor = scale_factor
        self.temporal_window = temporal_window
        self.num_frames = 2 * temporal_window + 1
        self._names: List[str] = []
        self._bufs: List[str] = []
        for n, shp in param_shapes(in_channels, scale_factor, num_features,
                                   num_residual_blocks, temporal_window).items():
            self.register_parameter(n.replace(".", "|"), torch.nn.Parameter(torch.zeros(shp)))
            self._names.append(n)
        for n, shp in buffer_shapes(num_features).items():
            init = torch.ones(shp) if n.endswith("running_var") else torch.zeros(shp)
            if n.endswith("num_batches_tracked"):
                init = torch.zeros((), dtype=torch.long)
            self.register_buffer(n.replace(".", "|"), init)
            self._bufs.append(n)

    def P(self) -> Params:
        d = {n: getattr(self, n.replace(".", "|")) for n in self._names}
        d.update({n: getattr(self, n.replace(".", "|")) for n in self._bufs})
        return d

    def load_named(self, sd: Dict[str, torch.Tensor]) -> None:
        with torch.no_grad():
            for n in self._names + self._bufs:
                getattr(self, n.replace(".", "|")).copy_(sd[n])

    def named(self) -> Dict[str, torch.Tensor]:
        return {n: getattr(self, n.replace(".", "|")) for n in self._names + self._bufs}

    def forward(self, lr_frames, return_intermediate=False):
        return sr_forward(self.P(), lr_frames, self.training, return_intermediate)


# --------------------------------------------------------------------------
# EWC (nerve_cl/continual/ewc.py)
# --------------------------------------------------------------------------
def ewc_fisher(model: torch.nn.Module, batches, names_params=None) -> Dict[str, torch.Tensor]:
    """EWC.compute_fisher, ewc.py:73-149 (empirical branch): eval mode; per batch
    zero grads, batch-mean MSE, backward, accumulate grad**2; divide by the
    number of samples seen."""
    named = list(names_params) if names_params is not None else list(model.named_parameters())
    fisher = {n: torch.zeros_like(p) for n, p in named}
    model.eval()
    seen = 0
    for inputs, targets in batches:
        model.zero_grad()
        loss = F.mse_loss(model(inputs), targets)
        loss.backward()
        for n, p in named:
            if p.grad is not None:
                fisher[n] += p.grad.detach() ** 2
        seen += inputs.shape[0]
    for n in fisher:
        fisher[n] /= max(seen, 1)
    return fisher


def ewc_online_merge(old: Optional[Dict[str, torch.Tensor]], new: Dict[str, torch.Tensor],
                     decay: float = 0.999) -> Dict[str, torch.Tensor]:
    """EWC.register_task online branch, ewc.py:180-191."""
    if not old:
        return new
    return {n: decay * old[n] + (1 - decay) * new[n] for n in new}


def ewc_penalty(named_params, fisher: Dict[str, torch.Tensor],
                optpar: Dict[str, torch.Tensor], lam: float) -> torch.Tensor:
    """EWC.penalty online branch, ewc.py:225-232."""
    total = 0.0
    for n, p in named_params:
        if n in fisher:
            total = total + (fisher[n] * (p - optpar[n]) ** 2).sum()
    return lam / 2 * total
