"""SuperResolutionNet on MI355X: the reference's module surface over libnvq HIP kernels.

Constructor, attribute names, ``state_dict`` keys and forward signature follow the
reference (nerve_cl/models/super_resolution.py:256-431) so that
``experiments/train_baseline.py`` / ``train_continual.py`` run unchanged.  The whole
forward is ONE autograd node: its forward launches the kernel schedule of
``nerve_cl._engine.forward`` and its backward the hand-written gradient schedule, writing
every parameter gradient into one flat bucket (which is what a data-parallel run
all-reduces over RCCL, see ``nerve_cl.parallel``).
"""
from __future__ import annotations

import os
from typing import Dict, List, Optional, Tuple

import torch
import torch.nn as nn

from nerve_cl import _engine, _graphs, _nvq, _ops
from nerve_cl._bucket import BucketedNet
from nerve_cl.models.layers import (
    CBAM,
    DepthwiseSeparableConv,
    LiteFlowNetCorrelation,
    PixelShuffleUpsampler,
    Stack,
    Act,
)
from nerve_cl.models.layers.efficient_layers import _Holder, _nchw_call

# The four sub-modules below are parameter holders inside SuperResolutionNet (whose forward / backward is ONE fused kernel
# schedule, nerve_cl._engine).  Called on their own - the reference's classes are ordinary nn.Modules - they compute through
# the same libnvq kernels, chained by the differentiable ops of nerve_cl._ops (exact-fp32 mode, fp32 NCHW tensors in and out
# like the reference; no CPU / PyTorch fallback).


class FeatureExtractor(_Holder):
    """head conv3x3+ReLU, three depthwise-separable blocks, skip (reference :22-54)."""

    def __init__(self, in_channels: int = 3, num_features: int = 64):
        super().__init__()
        self.head = Stack(nn.Conv2d(in_channels, num_features, 3, 1, 1), Act())
        self.body = Stack(*[DepthwiseSeparableConv(num_features, num_features) for _ in range(3)])
        self.num_features = num_features

    def forward_nhwc(self, x: torch.Tensor) -> torch.Tensor:
        h = _ops.Conv.apply(x, self.head[0].weight, self.head[0].bias, True, _nvq.MATH_F32)
        y = h
        for blk in self.body:
            y = blk.forward_nhwc(y)
        return _ops.ScaleAdd.apply(y, h, 1.0)                 # features = body(h) + h  (reference :53)

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        """(N, C, H, W) frame -> (N, F, H, W) features (reference :51-54)"""
        return _nchw_call(self.forward_nhwc, x, self.num_features)


class MotionEstimator(_Holder):
    """correlation(d=4) + flow_net 81->128->64->32->2 (reference :57-101)."""

    def __init__(self, in_channels: int = 64):
        super().__init__()
        self.correlation = LiteFlowNetCorrelation(max_displacement=4)
        self.flow_net = Stack(nn.Conv2d(81, 128, 3, 1, 1), Act(), nn.Conv2d(128, 64, 3, 1, 1), Act(),
                              nn.Conv2d(64, 32, 3, 1, 1), Act(), nn.Conv2d(32, 2, 3, 1, 1))
        self.in_channels = in_channels

    def forward(self, feat1: torch.Tensor, feat2: torch.Tensor) -> torch.Tensor:
        """(N, F, H, W) x 2 -> (N, 2, H, W) flow, channel 0 = dx, 1 = dy (reference :84-101)"""
        _nvq.require_device(feat1, "feat1")
        _nvq.require_device(feat2, "feat2")
        if feat1.dim() != 4 or feat1.shape != feat2.shape:
            raise RuntimeError(f"expected two (N,C,H,W) tensors of one shape, got {tuple(feat1.shape)} and {tuple(feat2.shape)}")
        C = feat1.shape[1]
        with _nvq.device_guard(feat1.device):
            y = _ops.Correlation.apply(_ops.ToNHWC.apply(feat1), _ops.ToNHWC.apply(feat2), C)
            for i in (0, 2, 4, 6):
                conv = self.flow_net[i]
                y = _ops.Conv.apply(y, conv.weight, conv.bias, i != 6, _nvq.MATH_F32)
            return _ops.ToNCHW.apply(y, 2)


class TemporalAggregator(_Holder):
    """attention convs T*F->F->F->T + softmax-weighted sum + CBAM (reference :146-209)."""

    def __init__(self, num_features: int = 64, num_frames: int = 3):
        super().__init__()
        self.num_frames = num_frames
        self.attention = Stack(nn.Conv2d(num_features * num_frames, num_features, 3, 1, 1), Act(),
                               nn.Conv2d(num_features, num_features, 3, 1, 1), Act(),
                               nn.Conv2d(num_features, num_frames, 3, 1, 1), Act())
        self.refine = CBAM(num_features)
        self.num_features = num_features

    def forward(self, aligned_features: List[torch.Tensor]) -> torch.Tensor:
        """T aligned (N, F, H, W) feature maps -> (N, F, H, W) (reference :180-209)"""
        T, C = self.num_frames, self.num_features
        if len(aligned_features) != T:
            raise RuntimeError(f"expected {T} aligned feature maps, got {len(aligned_features)}")
        for f in aligned_features:
            _nvq.require_device(f, "aligned features")
        with _nvq.device_guard(aligned_features[0].device):
            cat = _ops.ToNHWC.apply(aligned_features[0])
            for f in aligned_features[1:]:
                cat = _ops.Cat2.apply(cat, _ops.ToNHWC.apply(f))
            a = cat
            for i in (0, 2, 4):
                conv = self.attention[i]
                a = _ops.Conv.apply(a, conv.weight, conv.bias, i != 4, _nvq.MATH_F32)
            weighted = _ops.SoftmaxWeightedSum.apply(cat, a, T, C)
            return _ops.ToNCHW.apply(self.refine.forward_nhwc(weighted), C)


class ResidualDenseBlock(_Holder):
    """five dense 3x3 layers (growth 32) + 1x1 fusion, 0.2 residual scaling (reference :212-253)."""

    def __init__(self, num_features: int = 64, growth_rate: int = 32, num_layers: int = 5):
        super().__init__()
        if (growth_rate, num_layers) != (_engine.GROWTH, _engine.LAYERS):
            raise NotImplementedError("libnvq implements growth_rate=32, num_layers=5 (the values the SR net uses)")
        self.layers = nn.ModuleList()
        ch = num_features
        for _ in range(num_layers):
            self.layers.append(Stack(nn.Conv2d(ch, growth_rate, 3, 1, 1), Act()))
            ch += growth_rate
        self.lff = nn.Conv2d(ch, num_features, 1)
        self.num_features = num_features

    def forward_nhwc(self, x: torch.Tensor) -> torch.Tensor:
        cat = x
        for layer in self.layers:
            y = _ops.Conv.apply(cat, layer[0].weight, layer[0].bias, True, _nvq.MATH_F32)
            cat = _ops.Cat2.apply(cat, y)
        out = _ops.Conv.apply(cat, self.lff.weight, self.lff.bias, False, _nvq.MATH_F32)
        return _ops.ScaleAdd.apply(out, x, 0.2)                # local residual learning, scaled (reference :253)

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        """(N, F, H, W) -> (N, F, H, W) (reference :245-253)"""
        return _nchw_call(self.forward_nhwc, x, self.num_features)


def warp_features(features: torch.Tensor, flow: torch.Tensor) -> torch.Tensor:
    """Reference super_resolution.py:104-143: sample `features` (N, C, H, W) at pixel coordinates (x + flow[:, 0],
    y + flow[:, 1]) - bilinear, zeros outside the image, align_corners=True.  Differentiable w.r.t. both arguments."""
    _nvq.require_device(features, "features")
    _nvq.require_device(flow, "flow")
    if features.dim() != 4 or flow.dim() != 4 or flow.shape[1] != 2 or flow.shape[0] != features.shape[0] \
            or flow.shape[2:] != features.shape[2:]:
        raise RuntimeError(f"expected (N,C,H,W) features and (N,2,H,W) flow, got {tuple(features.shape)} and {tuple(flow.shape)}")
    C = features.shape[1]
    with _nvq.device_guard(features.device):
        return _ops.ToNCHW.apply(_ops.Warp.apply(_ops.ToNHWC.apply(features), _ops.ToNHWC.apply(flow), C), C)


class _SRFunction(torch.autograd.Function):
    """One autograd node for the whole network."""

    @staticmethod
    def forward(ctx, net: "SuperResolutionNet", frames: torch.Tensor, want_inter: bool, *params):
        P = net._tensor_dict()
        act = torch.bfloat16 if (net.bf16_activations and net.math_mode == _nvq.MATH_BF16) else torch.float32
        need_grad = any(ctx.needs_input_grad[3:])
        ctx.net = net
        ctx.graph = ctx.token = None
        if need_grad:
            net._mark_awaiting(ctx)
        if net._graphs_wanted(frames) and not want_inter:
            hit = net._step_graphs.forward(net, frames, need_grad, act)
            if hit is not None:
                out, entry, ctx.token, gen = hit
                ctx.graph = (entry, gen) if need_grad else None
                ctx.sv = None
                net._last_intermediates = None
                return out
        out, sv = _engine.forward(P, frames, net._F, net._NB, net.scale_factor, net.training, net.math_mode, act)
        ctx.sv = sv if need_grad else None
        net._last_intermediates = _engine.intermediates(sv) if want_inter else None
        return out

    @staticmethod
    def backward(ctx, dout):
        net, sv = ctx.net, ctx.sv
        names = net._param_names
        if ctx.graph is not None:
            flat, views = net._step_graphs.backward(net, ctx.graph[0], ctx.graph[1], dout)
        else:
            if sv is None:
                raise RuntimeError("SuperResolutionNet backward called without saved forward state (a second backward through "
                                   "the same forward needs net.retain_backward_state = True, the analogue of retain_graph)")
            flat, views = net._new_grad_bucket()
            with torch.cuda.device(dout.device):
                _engine.backward(net._tensor_dict(), sv, dout.contiguous().float(), views)
            # A custom Function cannot see retain_graph, and the state of a 540p step is tens of GB that must not outlive
            # the backward (autograd frees its own saved tensors here too), so it is dropped unless the module asks to keep it.
            if not getattr(net, "retain_backward_state", False):
                ctx.sv = None
        # data-parallel all-reduce of the whole bucket (nerve_cl.parallel), then the deferred EWC penalty gradient
        net._finish_bucket(flat)
        return (None, None, None) + tuple(views[n] for n in names)


class SuperResolutionNet(BucketedNet):
    """Temporal super-resolution network (reference :256-431), MI355X-native.

    Args (same as the reference):
        in_channels, scale_factor (2/3/4), num_features (power of two >= 16),
        num_residual_blocks, temporal_window (T = 2*window + 1 frames).
    """

    def __init__(self, in_channels: int = 3, scale_factor: int = 2, num_features: int = 64,
                 num_residual_blocks: int = 8, temporal_window: int = 1):
        super().__init__()
        if num_features < 16 or num_features & (num_features - 1) or num_features > 256:
            raise NotImplementedError("libnvq needs num_features to be a power of two in [16, 256]")
        if 2 * temporal_window + 1 > _nvq.MAX_T:
            raise NotImplementedError(f"libnvq supports at most {_nvq.MAX_T} frames")
        self.scale_factor = scale_factor
        self.temporal_window = temporal_window
        self.num_frames = 2 * temporal_window + 1
        self._F, self._NB, self._Cimg = num_features, num_residual_blocks, in_channels

        self.feature_extractor = FeatureExtractor(in_channels, num_features)
        self.motion_estimator = MotionEstimator(num_features)
        self.temporal_aggregator = TemporalAggregator(num_features, self.num_frames)
        self.residual_blocks = Stack(*[ResidualDenseBlock(num_features) for _ in range(num_residual_blocks)])
        self.gff = Stack(nn.Conv2d(num_features, num_features, 3, 1, 1), Act())
        self.upsampler = PixelShuffleUpsampler(num_features, scale_factor, in_channels)
        self.bicubic_upsample = Act()   # nn.Upsample(bicubic) in the reference; fused into the tail kernel

        # Precision knobs (not part of the reference surface).  math_mode: MFMA operand type of the convolutions
        # (fp32 = the parity mode, 1e-3 of the CPU path; bf16 = the throughput mode of BASELINE cfg2).
        # bf16_activations: additionally store the conv-internal tensors (dense-block buffers, flow-net and
        # attention hidden activations, their gradients) as bf16; only honoured with math_mode = MATH_BF16.
        # Environment default (for unmodified caller scripts): NVQ_MATH=bf16 selects the throughput mode.
        bf16 = os.environ.get("NVQ_MATH", "f32").lower() in ("bf16", "bfloat16")
        self.math_mode = _nvq.MATH_BF16 if bf16 else _nvq.MATH_F32
        self.bf16_activations = bf16 and os.environ.get("NVQ_BF16_ACTIVATIONS", "1") != "0"
        self._init_bucket()
        self._last_intermediates = None
        # HIP-graph replay of the step (nerve_cl/_graphs.py): True / False / "auto" = only for small frame sizes.  Off by
        # default: measured on MI355X it frees the host thread but does not shorten the step (see _graphs.py).
        # NVQ_GRAPH=1|0|auto sets the default for unmodified caller scripts.
        env = os.environ.get("NVQ_GRAPH", "0").lower()
        self.use_hip_graphs = True if env in ("1", "on", "true") else "auto" if env == "auto" else False
        self._step_graphs = _graphs.StepGraphs()

    GRAPH_AUTO_MAX_PIXELS = 8 * 3 * 128 * 128     # B*T*H*W up to which a step is launch-bound on MI355X

    def _graphs_wanted(self, frames: torch.Tensor) -> bool:
        if self.use_hip_graphs == "auto":
            return frames.numel() // frames.shape[2] <= self.GRAPH_AUTO_MAX_PIXELS
        return bool(self.use_hip_graphs)

    # ------------------------------------------------------------------ reference API
    def forward(self, lr_frames: torch.Tensor, return_intermediate: bool = False):
        """(B,T,C,H,W) low-resolution clip -> (B,C,H*s,W*s) super-resolved centre frame."""
        B, T, C, H, W = lr_frames.shape
        _nvq.require_device(lr_frames, "lr_frames")
        named = self._named_params()
        _nvq.require_device(named[0][1], "SuperResolutionNet parameters")
        if T != self.num_frames:
            raise RuntimeError(f"expected {self.num_frames} frames (temporal_window={self.temporal_window}), got {T}")
        if C != self._Cimg:
            raise RuntimeError(f"expected {self._Cimg} image channels, got {C}")
        if H < 2 or W < 2:
            raise RuntimeError("frames must be at least 2x2 (grid_sample normalisation divides by size-1)")
        frames = lr_frames.detach().to(torch.float32).contiguous()
        params = [p for _, p in named]
        with torch.cuda.device(frames.device):      # the kernels launch on the CURRENT device's stream
            out = _SRFunction.apply(self, frames, bool(return_intermediate), *params)
        if return_intermediate:
            inter, self._last_intermediates = self._last_intermediates, None
            return out, inter
        return out

    # ------------------------------------------------------------------ inference with cached per-frame features
    def _act_dtype(self):
        return torch.bfloat16 if (self.bf16_activations and self.math_mode == _nvq.MATH_BF16) else torch.float32

    @torch.no_grad()
    def extract_features(self, frames: torch.Tensor) -> torch.Tensor:
        """Eval-mode FeatureExtractor output of every frame of a video, (B,Tv,C,H,W) -> opaque [Tv,B,H,W,F] tensor for
        `forward_cached`.  In eval mode a frame's features do not depend on the window it is used in, so a sliding-window
        caller (EnhancementEngine.enhance_video) needs them once per frame instead of once per window."""
        if self.training:
            raise RuntimeError("extract_features is an eval-mode (running BatchNorm statistics) inference path")
        _nvq.require_device(frames, "frames")
        with torch.cuda.device(frames.device):
            return _engine.extract_features(self._tensor_dict(), frames.detach().to(torch.float32).contiguous(), self._F,
                                            self.math_mode, self._act_dtype())

    @torch.no_grad()
    def forward_cached(self, lr_frames: torch.Tensor, features: torch.Tensor) -> torch.Tensor:
        """forward(lr_frames) with the frames' features (extract_features(...)[window indices]) supplied by the caller."""
        if self.training:
            raise RuntimeError("forward_cached is an eval-mode inference path")
        B, T, C, H, W = lr_frames.shape
        if T != self.num_frames:
            raise RuntimeError(f"expected {self.num_frames} frames, got {T}")
        with torch.cuda.device(lr_frames.device):
            out, _ = _engine.forward(self._tensor_dict(), lr_frames.detach().to(torch.float32).contiguous(), self._F,
                                     self._NB, self.scale_factor, False, self.math_mode, self._act_dtype(),
                                     features=features.contiguous())
        return out

    def forward_single(self, lr_frame: torch.Tensor) -> torch.Tensor:
        """Upscale one frame: it is repeated T times (reference :393-405)."""
        return self.forward(lr_frame.unsqueeze(1).expand(-1, self.num_frames, -1, -1, -1))

    def get_num_parameters(self) -> int:
        return sum(p.numel() for p in self.parameters() if p.requires_grad)

    def get_flops(self, input_size: Tuple[int, int] = (128, 128)) -> int:
        """The reference's closed-form estimate, reproduced verbatim including its fixed
        F=64 / 8 blocks (reference :411-431)."""
        H, W = input_size
        C, F = 3, 64
        flops = H * W * C * F * 9
        flops += H * W * F * 81 * (self.num_frames - 1)
        flops += H * W * F * F * 9 * 8
        flops += H * W * F * (C * self.scale_factor * self.scale_factor) * 9
        return flops


class _LightFunction(torch.autograd.Function):
    @staticmethod
    def forward(ctx, net: "LightweightSuperResolution", x: torch.Tensor, *params):
        act = torch.bfloat16 if (net.bf16_activations and net.math_mode == _nvq.MATH_BF16) else torch.float32
        out, sv = _engine.light_forward(net._tensor_dict(), x, net.scale_factor, net.training, net.math_mode, act)
        ctx.net = net
        ctx.sv = sv if any(ctx.needs_input_grad[2:]) else None
        if ctx.sv is not None:
            net._mark_awaiting(ctx)
        return out

    @staticmethod
    def backward(ctx, dout):
        net, sv = ctx.net, ctx.sv
        if sv is None:
            raise RuntimeError("LightweightSuperResolution backward called without saved forward state")
        flat, views = net._new_grad_bucket()
        with torch.cuda.device(dout.device):
            _engine.light_backward(net._tensor_dict(), sv, dout.contiguous().float(), views)
        if not getattr(net, "retain_backward_state", False):
            ctx.sv = None
        net._finish_bucket(flat)
        return (None, None) + tuple(views[n] for n in net._param_names)


class LightweightSuperResolution(BucketedNet):
    """Single-frame variant (reference :434-470): conv3x3+ReLU, 4 depthwise-separable blocks (32 features),
    conv3x3 -> PixelShuffle, + bicubic(x), clamp.  ``state_dict`` keys are the reference's (``net.<i>.*``)."""

    def __init__(self, scale_factor: int = 2):
        super().__init__()
        self.scale_factor = scale_factor
        F = _engine.LIGHT_F
        self.net = Stack(nn.Conv2d(3, F, 3, 1, 1), Act(),
                         DepthwiseSeparableConv(F, F), DepthwiseSeparableConv(F, F),
                         DepthwiseSeparableConv(F, F), DepthwiseSeparableConv(F, F),
                         nn.Conv2d(F, 3 * scale_factor ** 2, 3, 1, 1), Act())
        self.bicubic = Act()
        bf16 = os.environ.get("NVQ_MATH", "f32").lower() in ("bf16", "bfloat16")
        self.math_mode = _nvq.MATH_BF16 if bf16 else _nvq.MATH_F32
        self.bf16_activations = bf16 and os.environ.get("NVQ_BF16_ACTIVATIONS", "1") != "0"
        self._init_bucket()

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        """(B,3,H,W) -> (B,3,H*s,W*s)."""
        _nvq.require_device(x, "x")
        named = self._named_params()
        _nvq.require_device(named[0][1], "LightweightSuperResolution parameters")
        if x.dim() != 4 or x.shape[1] != 3:
            raise RuntimeError(f"expected (B,3,H,W), got {tuple(x.shape)}")
        params = [p for _, p in named]
        with torch.cuda.device(x.device):
            return _LightFunction.apply(self, x.detach().to(torch.float32).contiguous(), *params)
