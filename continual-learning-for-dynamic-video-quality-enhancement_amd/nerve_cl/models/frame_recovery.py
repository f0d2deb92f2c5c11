"""FrameRecoveryNet on MI355X: the reference's inpainting network (nerve_cl/models/frame_recovery.py:23-446) with the
same constructor, sub-module / ``state_dict`` names, default initialisation and forward signature, every arithmetic step
a libnvq kernel (``nerve_cl._ops``: the implicit-GEMM MFMA convolutions of the SR path for all 1x1 / 3x3 / (2+1)D /
transposed convolutions, plus the generic NHWC kernels of ``csrc/fr_ops.hip``).

Like ``SuperResolutionNet`` the whole network is ONE node for the caller's autograd: its backward delivers every parameter
gradient in one flat bucket (data-parallel all-reduce, EWC Fisher and the fused EWC penalty gradient work on it unchanged,
``nerve_cl._bucket``).  Inside the node the layer ops are chained by autograd, which contributes no arithmetic.

Layout: activations fp32 NHWC; the T reference frames are a time-major image batch ``[T*B, H, W, C]`` so that BatchNorm3d
is BatchNorm over that batch, the (1,3,3) convolutions are ordinary 3x3 convolutions and the (3,1,1) convolutions are
three accumulating 1x1 convolutions over shifted image ranges.
"""
from __future__ import annotations

import os
from typing import List, Optional, Tuple

import torch
import torch.nn as nn

from nerve_cl import _nvq, _ops
from nerve_cl._bucket import BucketedNet
from nerve_cl.models.layers import CBAM, ResidualBlock, Stack, Act, TemporalConv3D
from nerve_cl.models.layers.efficient_layers import _Holder


class SpatialEncoder(_Holder):
    """7x7 stride-2 stem + BN + ReLU + max-pool, three residual stages (the last two entered through a 1x1 stride-2 conv +
    BN), CBAM (reference :23-108)."""

    def __init__(self, in_channels: int = 3, base_channels: int = 64, num_blocks: int = 2):
        super().__init__()
        self.stem = Stack(nn.Conv2d(in_channels, base_channels, 7, 2, 3, bias=False), nn.BatchNorm2d(base_channels), Act(), Act())
        self.stage1 = self._make_stage(base_channels, base_channels, num_blocks)
        self.stage2 = self._make_stage(base_channels, base_channels * 2, num_blocks, stride=2)
        self.stage3 = self._make_stage(base_channels * 2, base_channels * 4, num_blocks, stride=2)
        self.attention = CBAM(base_channels * 4)

    @staticmethod
    def _make_stage(cin: int, cout: int, num_blocks: int, stride: int = 1) -> Stack:
        layers: List[nn.Module] = []
        if stride != 1 or cin != cout:
            layers.append(Stack(nn.Conv2d(cin, cout, 1, stride, bias=False), nn.BatchNorm2d(cout)))
        layers += [ResidualBlock(cout) for _ in range(num_blocks)]
        return Stack(*layers)

    def forward_nhwc(self, x4: torch.Tensor, math: int, act_dtype=torch.float32) -> "Tuple[torch.Tensor, List[torch.Tensor]]":
        y = _ops.Stem7.apply(x4, self.stem[0].weight, act_dtype)
        y = _ops.bn(y, self.stem[1], self.training, relu=True)
        y = _ops.MaxPool.apply(y, 3, 2, 1)
        skips = [y]
        for si, stage in enumerate((self.stage1, self.stage2, self.stage3)):
            for layer in stage:
                if isinstance(layer, ResidualBlock):
                    y = layer.forward_nhwc(y, math)
                else:                                             # 1x1 stride-2 conv + BN
                    y = _ops.Conv.apply(_ops.Subsample2.apply(y), layer[0].weight, None, False, math)
                    y = _ops.bn(y, layer[1], self.training, relu=False)
            if si < 2:
                skips.append(y)
        # the attention kernels (global pooling, 7x7 conv over channel statistics) are fp32; at 1/16 resolution
        return self.attention.forward_nhwc(_ops.Cast.apply(y, torch.float32, y.shape[-1])), skips


class TemporalEncoder(_Holder):
    """three (2+1)D convolution blocks with 2x2 spatial max-pooling after the first two, mean over time (reference :111-167)"""

    def __init__(self, in_channels: int = 3, out_channels: int = 256, temporal_window: int = 3):
        super().__init__()
        self.temporal_window = temporal_window
        self.conv1 = TemporalConv3D(in_channels, 64, temporal_kernel=3)
        self.conv2 = TemporalConv3D(64, 128, temporal_kernel=3)
        self.conv3 = TemporalConv3D(128, out_channels, temporal_kernel=3)
        self.temporal_pool = Act()

    def forward_nhwc(self, frames: torch.Tensor, T: int, math: int, act_dtype=torch.float32) -> torch.Tensor:
        """frames: time-major image batch [T*B, H, W, 4]"""
        y = _ops.MaxPool.apply(self.conv1.forward_nhwc(frames, T, math, act_dtype), 2, 2, 0)
        y = _ops.MaxPool.apply(self.conv2.forward_nhwc(y, T, math), 2, 2, 0)
        return _ops.GroupMean.apply(self.conv3.forward_nhwc(y, T, math), T, self.conv3.out_channels)   # fp32 out

    def forward_tc(self, frames_tc: torch.Tensor, T: int, math: int, act_dtype=torch.float32) -> torch.Tensor:
        """frames_tc: time-in-channels tensor [B, H, W, T*4] (frame t = channels 4t..4t+2); T >= 2"""
        y = _ops.MaxPool.apply(self.conv1.forward_tc(frames_tc, T, math, act_dtype), 2, 2, 0)
        y = _ops.MaxPool.apply(self.conv2.forward_tc(y, T, math), 2, 2, 0)
        return _ops.GroupMeanTC.apply(self.conv3.forward_tc(y, T, math), T, self.conv3.out_channels)    # fp32 out


class FusionModule(_Holder):
    """1x1 alignment of [spatial | temporal], 2-way softmax attention over the channel means, two residual blocks, CBAM
    (reference :170-257)"""

    def __init__(self, spatial_channels: int = 256, temporal_channels: int = 256, out_channels: int = 256):
        super().__init__()
        if not (spatial_channels == temporal_channels == out_channels):
            raise NotImplementedError("libnvq's fusion kernel needs equal spatial / temporal / output widths (as FrameRecoveryNet builds)")
        self.align = nn.Conv2d(spatial_channels + temporal_channels, out_channels, 1)
        self.attention = Stack(nn.Conv2d(out_channels, out_channels // 4, 1), Act(), nn.Conv2d(out_channels // 4, 2, 1), Act())
        self.refine = Stack(ResidualBlock(out_channels), ResidualBlock(out_channels), CBAM(out_channels))

    def forward_nhwc(self, sp: torch.Tensor, tp: torch.Tensor, math: int) -> torch.Tensor:
        if sp.shape[1:3] != tp.shape[1:3]:
            tp = _ops.Resize.apply(tp, sp.shape[1], sp.shape[2])
        aligned = _ops.Conv.apply(_ops.Cat2.apply(sp, tp), self.align.weight, self.align.bias, False, math)
        a = _ops.Conv.apply(aligned, self.attention[0].weight, self.attention[0].bias, True, math)
        logits = _ops.Conv.apply(a, self.attention[2].weight, self.attention[2].bias, False, math)
        y = _ops.FusionMix.apply(aligned, logits, sp, tp)
        y = self.refine[0].forward_nhwc(y, math)
        y = self.refine[1].forward_nhwc(y, math)
        return self.refine[2].forward_nhwc(y)


class Decoder(_Holder):
    """four ConvTranspose2d(4,2,1) + BN + ReLU stages, conv3x3 + tanh (reference :260-332; the skip connections it is handed
    are not used there either)"""

    def __init__(self, in_channels: int = 256, out_channels: int = 3, base_channels: int = 64):
        super().__init__()
        widths = [(in_channels, base_channels * 4), (base_channels * 4, base_channels * 2), (base_channels * 2, base_channels),
                  (base_channels, base_channels // 2)]
        for i, (ci, co) in enumerate(widths, 1):
            setattr(self, f"up{i}", Stack(nn.ConvTranspose2d(ci, co, 4, 2, 1, bias=False), nn.BatchNorm2d(co), Act()))
        self.final = Stack(nn.Conv2d(base_channels // 2, out_channels, 3, 1, 1), Act())

    def forward_nhwc(self, x: torch.Tensor, math: int, act_dtype=torch.float32) -> torch.Tensor:
        x = _ops.Cast.apply(x, act_dtype, x.shape[-1])
        for i in (1, 2, 3, 4):
            up = getattr(self, f"up{i}")
            x = _ops.bn(_ops.ConvT.apply(x, up[0].weight, math), up[1], self.training, relu=True)
        return _ops.Tanh.apply(_ops.Conv.apply(x, self.final[0].weight, self.final[0].bias, False, math, torch.float32))


class _FRFunction(torch.autograd.Function):
    """One autograd node for the whole network: runs the layer graph (an inner autograd graph over the libnvq ops), keeps
    it, and on backward collects its parameter gradients into the flat bucket."""

    @staticmethod
    def forward(ctx, net: "FrameRecoveryNet", frame, refs, mask, *params):
        need_grad = any(ctx.needs_input_grad[4:])
        ctx.net = net
        if not need_grad:
            ctx.inner = None
            with torch.no_grad():
                return net._graph(frame, refs, mask)
        # Function.forward runs with autograd switched off: switch it back on for the layer graph, whose leaves are the
        # real parameters (torch.autograd.grad in backward reads their gradients without touching .grad)
        with torch.enable_grad():
            out = net._graph(frame, refs, mask)
            leaves = list(params)
        ctx.inner = (out, leaves)
        net._mark_awaiting(ctx)
        return out.detach()

    @staticmethod
    def backward(ctx, dout):
        net = ctx.net
        if ctx.inner is None:
            raise RuntimeError("FrameRecoveryNet backward called twice (or without saved state)")
        out, leaves = ctx.inner
        ctx.inner = None
        with _nvq.device_guard(dout.device):
            grads = torch.autograd.grad(out, leaves, dout.contiguous(), allow_unused=True)
            lay, total = net._bucket_layout()
            pieces, off = [], 0
            for (name, (o, k)), g, leaf in zip(lay.items(), grads, leaves):
                if o > off:
                    pieces.append(torch.zeros(o - off, dtype=torch.float32, device=dout.device))
                pieces.append(g.reshape(-1) if g is not None else torch.zeros(k, dtype=torch.float32, device=dout.device))
                off = o + k
            if total > off:
                pieces.append(torch.zeros(total - off, dtype=torch.float32, device=dout.device))
            flat = torch.cat(pieces)
            net._finish_bucket(flat)
        views = net._bucket_views(flat)
        return (None, None, None, None) + tuple(views[n] for n in net._param_names)


class FrameRecoveryNet(BucketedNet):
    """Recovers a corrupted frame from its neighbours (reference :335-446).

    Args (same as the reference): in_channels (3), base_channels (a power of two, 16..64), temporal_window.
    forward(corrupted_frame (B,C,H,W), reference_frames (B,T,C,H,W), corruption_mask (B,1,H,W) or None) -> (B,C,H,W).
    """

    def __init__(self, in_channels: int = 3, base_channels: int = 64, temporal_window: int = 2):
        super().__init__()
        if in_channels != 3:
            raise NotImplementedError("libnvq's stem kernel takes the 3 + 1 (mask) channel input of the reference's default")
        if base_channels < 16 or base_channels > 64 or base_channels & (base_channels - 1):
            raise NotImplementedError("libnvq needs base_channels to be a power of two in [16, 64]")
        self.temporal_window = temporal_window
        self.spatial_encoder = SpatialEncoder(in_channels=in_channels + 1, base_channels=base_channels)
        self.temporal_encoder = TemporalEncoder(in_channels=in_channels, out_channels=base_channels * 4,
                                                temporal_window=temporal_window)
        self.fusion = FusionModule(base_channels * 4, base_channels * 4, base_channels * 4)
        self.decoder = Decoder(base_channels * 4, in_channels, base_channels)
        # Precision knobs (as on SuperResolutionNet): math_mode MATH_BF16 = bf16 MFMA operands in every convolution;
        # bf16_activations (only honoured then) = the encoders' and the decoder's activations and their gradients stored as
        # bf16 (statistics, the 1/16-resolution attention / fusion stage and the output image stay fp32)
        bf16 = os.environ.get("NVQ_MATH", "f32").lower() in ("bf16", "bfloat16")
        self.math_mode = _nvq.MATH_BF16 if bf16 else _nvq.MATH_F32
        self.bf16_activations = os.environ.get("NVQ_BF16_ACTIVATIONS", "1") != "0"
        # temporal encoder layout: True = reference frames side by side in the channel dimension ([B,H,W,T*C]: each (3,1,1)
        # convolution is one 1x1 convolution per frame, no accumulating passes); False = time-major image batch [T*B,H,W,C]
        self.time_in_channels = os.environ.get("NVQ_FR_TIME_IN_CHANNELS", "1") != "0"
        self._init_bucket()

    # ------------------------------------------------------------------ the layer graph (NHWC, libnvq ops)
    def _graph(self, frame: torch.Tensor, refs: torch.Tensor, mask: torch.Tensor) -> torch.Tensor:
        B, C, H, W = frame.shape
        T = refs.shape[1]
        math = self.math_mode
        act = torch.bfloat16 if (math == _nvq.MATH_BF16 and self.bf16_activations) else torch.float32
        x4 = torch.empty(B, H, W, 4, dtype=torch.float32, device=frame.device)
        _ops.nchw_to_nhwc_(frame, C * H * W, B, C, H, W, x4, 0)
        _ops.nchw_to_nhwc_(mask, H * W, B, 1, H, W, x4, C)
        sp, _skips = self.spatial_encoder.forward_nhwc(x4, math, act)
        if T >= 2 and self.time_in_channels:
            # frames side by side in the channel dimension: the (3,1,1) convolutions become single 1x1 convolutions
            r = torch.empty(B, H, W, T * 4, dtype=torch.float32, device=frame.device)
            for t in range(T):
                _ops.nchw_to_nhwc_(refs, T * C * H * W, B, C, H, W, r, 4 * t, czero=4, src_offset=t * C * H * W)
            tp = self.temporal_encoder.forward_tc(r, T, math, act)
        else:
            r = torch.empty(T * B, H, W, 4, dtype=torch.float32, device=frame.device)
            for t in range(T):                                    # time-major image batch, channel 3 = 0
                _ops.nchw_to_nhwc_(refs, T * C * H * W, B, C, H, W, r[t * B:(t + 1) * B], 0, czero=4, src_offset=t * C * H * W)
            tp = self.temporal_encoder.forward_nhwc(r, T, math, act)
        rec = self.decoder.forward_nhwc(self.fusion.forward_nhwc(sp, tp, math), math, act)
        if rec.shape[1:3] != (H, W):
            rec = _ops.Resize.apply(rec, H, W)
        return _ops.MaskBlend.apply(frame, rec, mask)

    # ------------------------------------------------------------------ reference API
    def forward(self, corrupted_frame: torch.Tensor, reference_frames: torch.Tensor,
                corruption_mask: Optional[torch.Tensor] = None) -> torch.Tensor:
        B, C, H, W = corrupted_frame.shape
        _nvq.require_device(corrupted_frame, "corrupted_frame")
        named = self._named_params()
        _nvq.require_device(named[0][1], "FrameRecoveryNet parameters")
        if C != 3:
            raise RuntimeError(f"expected 3 image channels, got {C}")
        if reference_frames.dim() != 5 or reference_frames.shape[0] != B or tuple(reference_frames.shape[2:]) != (C, H, W):
            raise RuntimeError(f"reference_frames must be (B,T,{C},{H},{W}), got {tuple(reference_frames.shape)}")
        if H < 32 or W < 32:
            raise RuntimeError("frames must be at least 32x32 (the encoder downsamples by 16)")
        frame = corrupted_frame.detach().to(torch.float32).contiguous()
        refs = reference_frames.detach().to(torch.float32).contiguous()
        if corruption_mask is None:
            mask = torch.zeros(B, 1, H, W, dtype=torch.float32, device=frame.device)
        else:
            mask = corruption_mask.detach().to(frame.device, torch.float32).contiguous()
        params = [p for _, p in named]
        with torch.cuda.device(frame.device):
            return _FRFunction.apply(self, frame, refs, mask, *params)

    def get_num_parameters(self) -> int:
        return sum(p.numel() for p in self.parameters() if p.requires_grad)
