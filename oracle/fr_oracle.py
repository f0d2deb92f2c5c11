"""CPU oracle for the reference's FrameRecoveryNet (BASELINE cfg4) - groundwork for the HIP
build of that network, which does not exist yet.

TEST INFRASTRUCTURE ONLY, like everything under ``oracle/``: nothing in the product imports it.

The forward pass of ``nerve_cl.models.frame_recovery.FrameRecoveryNet`` (reference ``nerve_cl/models/frame_recovery.py``
and the layers of ``nerve_cl/models/layers/efficient_layers.py`` it is built from) restated as plain functions over a flat
``{state_dict name: tensor}`` dictionary; autograd supplies the backward.  Each function cites the reference lines it follows.

Parity pin: ``oracle/make_goldens.py --only-fr`` imports the reference in the build container, drives it and this
restatement with the same formula-generated weights / inputs, asserts agreement and writes ``tests/golden/fr_*.npz``;
``tests/test_oracle_golden.py`` checks this file against those fixtures on every CPU run.
"""
from __future__ import annotations

from typing import Dict, List, Tuple

import torch
import torch.nn.functional as F

Params = Dict[str, torch.Tensor]
BN_EPS, BN_MOMENTUM = 1e-5, 0.1        # nn.BatchNorm2d / BatchNorm3d defaults


# ---------------------------------------------------------------------------- parameter inventory
def _bn(shapes, buffers, pre: str, c: int) -> None:
    shapes[pre + "weight"], shapes[pre + "bias"] = (c,), (c,)
    buffers[pre + "running_mean"], buffers[pre + "running_var"], buffers[pre + "num_batches_tracked"] = (c,), (c,), ()


def _res_block(shapes, buffers, pre: str, c: int) -> None:
    """ResidualBlock(use_efficient=True), efficient_layers.py:118-143."""
    shapes[pre + "conv1.depthwise.weight"], shapes[pre + "conv1.pointwise.weight"] = (c, 1, 3, 3), (c, c, 1, 1)
    _bn(shapes, buffers, pre + "conv1.bn.", c)
    shapes[pre + "conv2.0.weight"], shapes[pre + "conv2.1.weight"] = (c, 1, 3, 3), (c, c, 1, 1)
    _bn(shapes, buffers, pre + "conv2.2.", c)


def _cbam(shapes, pre: str, c: int) -> None:
    shapes[pre + "channel_attention.fc.0.weight"] = (c // 16, c)
    shapes[pre + "channel_attention.fc.2.weight"] = (c, c // 16)
    shapes[pre + "spatial_attention.conv.weight"] = (1, 2, 7, 7)


def tconv_mid(cin: int, cout: int, tk: int = 3) -> int:
    """TemporalConv3D's intermediate width, efficient_layers.py:253-257."""
    return max((cin * cout * 9 * tk) // (cin * 9 + cout * tk), cout // 2)


def shapes(in_channels: int = 3, base: int = 64, num_blocks: int = 2) -> "Tuple[Dict[str, tuple], Dict[str, tuple]]":
    """(parameter shapes, buffer shapes) in the reference's state_dict order of names (frame_recovery.py:35-57,124-137,
    183-207,272-309,361-395)."""
    P: Dict[str, tuple] = {}
    Bf: Dict[str, tuple] = {}
    se = "spatial_encoder."
    P[se + "stem.0.weight"] = (base, in_channels + 1, 7, 7)
    _bn(P, Bf, se + "stem.1.", base)
    cin = base
    for si, cout in ((1, base), (2, base * 2), (3, base * 4)):
        idx = 0
        if si > 1:                                            # stride 2 and a channel change: 1x1 conv + BN first
            P[f"{se}stage{si}.0.0.weight"] = (cout, cin, 1, 1)
            _bn(P, Bf, f"{se}stage{si}.0.1.", cout)
            idx = 1
        for b in range(num_blocks):
            _res_block(P, Bf, f"{se}stage{si}.{idx + b}.", cout)
        cin = cout
    _cbam(P, se + "attention.", base * 4)
    te = "temporal_encoder."
    for name, (ci, co) in (("conv1", (in_channels, 64)), ("conv2", (64, 128)), ("conv3", (128, base * 4))):
        mid = tconv_mid(ci, co)
        P[f"{te}{name}.spatial.0.weight"] = (mid, ci, 1, 3, 3)
        _bn(P, Bf, f"{te}{name}.spatial.1.", mid)
        P[f"{te}{name}.temporal.0.weight"] = (co, mid, 3, 1, 1)
        _bn(P, Bf, f"{te}{name}.temporal.1.", co)
    c4 = base * 4
    P["fusion.align.weight"], P["fusion.align.bias"] = (c4, 2 * c4, 1, 1), (c4,)
    P["fusion.attention.0.weight"], P["fusion.attention.0.bias"] = (c4 // 4, c4, 1, 1), (c4 // 4,)
    P["fusion.attention.2.weight"], P["fusion.attention.2.bias"] = (2, c4 // 4, 1, 1), (2,)
    _res_block(P, Bf, "fusion.refine.0.", c4)
    _res_block(P, Bf, "fusion.refine.1.", c4)
    _cbam(P, "fusion.refine.2.", c4)
    for i, (ci, co) in enumerate(((c4, base * 4), (base * 4, base * 2), (base * 2, base), (base, base // 2)), 1):
        P[f"decoder.up{i}.0.weight"] = (ci, co, 4, 4)       # ConvTranspose2d layout [in, out, k, k]
        _bn(P, Bf, f"decoder.up{i}.1.", co)
    P["decoder.final.0.weight"], P["decoder.final.0.bias"] = (in_channels, base // 2, 3, 3), (in_channels,)
    return P, Bf


# ---------------------------------------------------------------------------- layers
def batch_norm(x: torch.Tensor, P: Params, pre: str, training: bool) -> torch.Tensor:
    """nn.BatchNorm2d / BatchNorm3d: statistics over every dimension but the channel one; train = biased batch variance for
    the normalisation, unbiased one into running_var, momentum 0.1."""
    dims = [d for d in range(x.dim()) if d != 1]
    shape = [1, -1] + [1] * (x.dim() - 2)
    if training:
        n = x.numel() // x.shape[1]
        mean, var = x.mean(dim=dims), x.var(dim=dims, unbiased=False)
        with torch.no_grad():
            P[pre + "running_mean"].mul_(1 - BN_MOMENTUM).add_(BN_MOMENTUM * mean.detach())
            P[pre + "running_var"].mul_(1 - BN_MOMENTUM).add_(BN_MOMENTUM * var.detach() * (n / max(n - 1, 1)))
            P[pre + "num_batches_tracked"] += 1
    else:
        mean, var = P[pre + "running_mean"], P[pre + "running_var"]
    inv = torch.rsqrt(var + BN_EPS) * P[pre + "weight"]
    return (x - mean.view(shape)) * inv.view(shape) + P[pre + "bias"].view(shape)


def residual_block(P: Params, pre: str, x: torch.Tensor, training: bool) -> torch.Tensor:
    """ResidualBlock.forward, efficient_layers.py:145-151 (conv1 = DepthwiseSeparableConv :62-67)."""
    c = x.shape[1]
    y = F.conv2d(x, P[pre + "conv1.depthwise.weight"], None, padding=1, groups=c)
    y = F.conv2d(y, P[pre + "conv1.pointwise.weight"], None)
    y = F.relu(batch_norm(y, P, pre + "conv1.bn.", training))
    y = F.conv2d(y, P[pre + "conv2.0.weight"], None, padding=1, groups=c)
    y = F.conv2d(y, P[pre + "conv2.1.weight"], None)
    y = batch_norm(y, P, pre + "conv2.2.", training)
    return F.relu(y + x)


def cbam(P: Params, pre: str, x: torch.Tensor) -> torch.Tensor:
    """CBAM, efficient_layers.py:176-180,200-205,225-228."""
    hid = F.relu(x.mean(dim=(2, 3)) @ P[pre + "channel_attention.fc.0.weight"].t())
    ca = torch.sigmoid(hid @ P[pre + "channel_attention.fc.2.weight"].t())
    xc = x * ca[:, :, None, None]
    sm = torch.cat([xc.mean(dim=1, keepdim=True), xc.max(dim=1, keepdim=True)[0]], dim=1)
    return xc * torch.sigmoid(F.conv2d(sm, P[pre + "spatial_attention.conv.weight"], None, padding=3))


def spatial_encoder(P: Params, x: torch.Tensor, training: bool, num_blocks: int = 2) -> "Tuple[torch.Tensor, List[torch.Tensor]]":
    """SpatialEncoder.forward, frame_recovery.py:83-108: 7x7 stride-2 stem + BN + ReLU + max-pool, three stages, CBAM."""
    pre = "spatial_encoder."
    y = F.conv2d(x, P[pre + "stem.0.weight"], None, stride=2, padding=3)
    y = F.max_pool2d(F.relu(batch_norm(y, P, pre + "stem.1.", training)), 3, 2, 1)
    skips = [y]
    for si in (1, 2, 3):
        idx = 0
        if si > 1:
            y = batch_norm(F.conv2d(y, P[f"{pre}stage{si}.0.0.weight"], None, stride=2), P, f"{pre}stage{si}.0.1.", training)
            idx = 1
        for b in range(num_blocks):
            y = residual_block(P, f"{pre}stage{si}.{idx + b}.", y, training)
        if si < 3:
            skips.append(y)
    return cbam(P, pre + "attention.", y), skips


def temporal_conv3d(P: Params, pre: str, x: torch.Tensor, training: bool) -> torch.Tensor:
    """TemporalConv3D.forward, efficient_layers.py:284-294: (1,3,3) conv + BN3d + ReLU, (3,1,1) conv + BN3d + ReLU."""
    y = F.relu(batch_norm(F.conv3d(x, P[pre + "spatial.0.weight"], None, padding=(0, 1, 1)), P, pre + "spatial.1.", training))
    return F.relu(batch_norm(F.conv3d(y, P[pre + "temporal.0.weight"], None, padding=(1, 0, 0)), P, pre + "temporal.1.", training))


def temporal_encoder(P: Params, frames: torch.Tensor, training: bool) -> torch.Tensor:
    """TemporalEncoder.forward, frame_recovery.py:142-167: (B,T,C,H,W) -> (B,C',H/4,W/4), mean over T at the end."""
    pre = "temporal_encoder."
    x = frames.permute(0, 2, 1, 3, 4)
    x = F.max_pool3d(temporal_conv3d(P, pre + "conv1.", x, training), (1, 2, 2))
    x = F.max_pool3d(temporal_conv3d(P, pre + "conv2.", x, training), (1, 2, 2))
    return temporal_conv3d(P, pre + "conv3.", x, training).mean(dim=2)


def fusion(P: Params, spatial: torch.Tensor, temporal: torch.Tensor, training: bool) -> torch.Tensor:
    """FusionModule.forward, frame_recovery.py:211-257.  The two 'projections' are channel means broadcast to C_out."""
    if spatial.shape[2:] != temporal.shape[2:]:
        temporal = F.interpolate(temporal, size=spatial.shape[2:], mode="bilinear", align_corners=False)
    aligned = F.conv2d(torch.cat([spatial, temporal], dim=1), P["fusion.align.weight"], P["fusion.align.bias"])
    a = F.relu(F.conv2d(aligned, P["fusion.attention.0.weight"], P["fusion.attention.0.bias"]))
    attn = torch.softmax(F.conv2d(a, P["fusion.attention.2.weight"], P["fusion.attention.2.bias"]), dim=1)
    c = aligned.shape[1]
    sp = spatial.mean(dim=1, keepdim=True).expand(-1, c, -1, -1)        # conv2d with ones / C_in, :244-251
    tp = temporal.mean(dim=1, keepdim=True).expand(-1, c, -1, -1)
    y = aligned + attn[:, 0:1] * sp + attn[:, 1:2] * tp
    y = residual_block(P, "fusion.refine.0.", y, training)
    y = residual_block(P, "fusion.refine.1.", y, training)
    return cbam(P, "fusion.refine.2.", y)


def decoder(P: Params, x: torch.Tensor, training: bool) -> torch.Tensor:
    """Decoder.forward, frame_recovery.py:311-332 (the skip connections are accepted and ignored there too)."""
    for i in (1, 2, 3, 4):
        x = F.conv_transpose2d(x, P[f"decoder.up{i}.0.weight"], None, stride=2, padding=1)
        x = F.relu(batch_norm(x, P, f"decoder.up{i}.1.", training))
    return torch.tanh(F.conv2d(x, P["decoder.final.0.weight"], P["decoder.final.0.bias"], padding=1))


def frame_recovery_forward(P: Params, corrupted: torch.Tensor, references: torch.Tensor, mask: torch.Tensor = None,
                           training: bool = True, num_blocks: int = 2) -> torch.Tensor:
    """FrameRecoveryNet.forward, frame_recovery.py:397-442."""
    B, C, H, W = corrupted.shape
    if mask is None:
        mask = torch.zeros(B, 1, H, W, device=corrupted.device)
    sp, _ = spatial_encoder(P, torch.cat([corrupted, mask], dim=1), training, num_blocks)
    tp = temporal_encoder(P, references, training)
    rec = decoder(P, fusion(P, sp, tp, training), training)
    if rec.shape[2:] != (H, W):
        rec = F.interpolate(rec, size=(H, W), mode="bilinear", align_corners=False)
    return corrupted * (1 - mask) + rec * mask
