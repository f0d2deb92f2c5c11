"""Building blocks of the NERVE-CL networks (reference nerve_cl/models/layers/efficient_layers.py:9-343).

Every class owns its parameters under the reference's attribute names, so ``state_dict()`` keys, shapes, dtypes and
default initialisation (same construction order => same RNG stream) are identical.  Inside ``SuperResolutionNet`` the
arithmetic is scheduled by ``nerve_cl._engine`` (fused kernels, the modules act as parameter holders); called on their
own - ``layer(x)`` with an (N,C,H,W) HIP tensor, as the reference's tests do (tests/test_models.py:19-38) - they run the
same libnvq kernels through the differentiable ops of ``nerve_cl._ops``.  There is no CPU / PyTorch fallback.

The ``*_nhwc`` methods are the internal form (fp32 [N,H,W,ld] activations) that ``FrameRecoveryNet`` chains without
layout round trips.
"""
from __future__ import annotations

import torch
import torch.nn as nn

from nerve_cl import _nvq, _ops

MATH_F32 = _nvq.MATH_F32


def _nchw_call(fn, x: torch.Tensor, cout: int) -> torch.Tensor:
    """run an NHWC op chain on an (N,C,H,W) HIP tensor"""
    _nvq.require_device(x, "input")
    if x.dim() != 4:
        raise RuntimeError(f"expected (N,C,H,W), got {tuple(x.shape)}")
    with _nvq.device_guard(x.device):
        return _ops.ToNCHW.apply(fn(_ops.ToNHWC.apply(x)), cout)


class _Holder(nn.Module):
    """Base: owns parameters, refuses to compute."""

    def forward(self, *args, **kwargs):  # pragma: no cover - guard
        raise RuntimeError(
            f"{type(self).__name__} only holds parameters; its arithmetic runs inside "
            "nerve_cl.models.SuperResolutionNet.forward as HIP kernels (no PyTorch fallback).")


class Stack(_Holder):
    """nn.Sequential-like container (children named '0','1',...) without a forward."""

    def __init__(self, *mods: nn.Module):
        super().__init__()
        for i, m in enumerate(mods):
            self.add_module(str(i), m)

    def __getitem__(self, i: int) -> nn.Module:
        return self._modules[str(i)]

    def __len__(self) -> int:
        return len(self._modules)

    def __iter__(self):
        return iter(self._modules.values())


class Act(_Holder):
    """Placeholder occupying the index an activation has in the reference's nn.Sequential."""


class DepthwiseSeparableConv(_Holder):
    """depthwise 3x3 (groups=C, no bias) -> pointwise 1x1 (no bias) -> BatchNorm2d -> ReLU
    (reference efficient_layers.py:9-67)."""

    def __init__(self, in_channels: int, out_channels: int, kernel_size: int = 3, stride: int = 1,
                 padding: int = 1, bias: bool = False):
        super().__init__()
        if (kernel_size, stride, padding, bias) != (3, 1, 1, False):
            raise NotImplementedError("libnvq implements the 3x3 / stride 1 / padding 1 / no-bias form the networks use")
        self.in_channels, self.out_channels = in_channels, out_channels
        self.depthwise = nn.Conv2d(in_channels, in_channels, 3, 1, 1, groups=in_channels, bias=False)
        self.pointwise = nn.Conv2d(in_channels, out_channels, 1, 1, 0, bias=False)
        self.bn = nn.BatchNorm2d(out_channels)
        self.act = Act()
        self.math_mode = MATH_F32

    def forward_nhwc(self, x: torch.Tensor, math: int = MATH_F32) -> torch.Tensor:
        y = _ops.DwConv.apply(x, self.depthwise.weight)
        y = _ops.Conv.apply(y, self.pointwise.weight, None, False, math)
        return _ops.bn(y, self.bn, self.training, relu=True)

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        """(N,C_in,H,W) -> (N,C_out,H,W)"""
        return _nchw_call(lambda t: self.forward_nhwc(t, self.math_mode), x, self.out_channels)


class PixelShuffleUpsampler(_Holder):
    """conv3x3(C -> out*s^2) + PixelShuffle(s) (reference efficient_layers.py:70-106)."""

    def __init__(self, in_channels: int, scale_factor: int = 2, out_channels: int = 3):
        super().__init__()
        self.scale_factor, self.out_channels = scale_factor, out_channels
        self.conv = nn.Conv2d(in_channels, out_channels * scale_factor ** 2, 3, 1, 1)
        self.pixel_shuffle = Act()
        self.math_mode = MATH_F32

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        """(N,C,H,W) -> (N,out,H*s,W*s)"""
        _nvq.require_device(x, "input")
        with _nvq.device_guard(x.device):
            u = _ops.Conv.apply(_ops.ToNHWC.apply(x), self.conv.weight, self.conv.bias, False, self.math_mode)
            return _ops.PixelShuffleNCHW.apply(u, self.out_channels, self.scale_factor)


class ResidualBlock(_Holder):
    """relu(conv2(conv1(x)) + x) with conv1 = DepthwiseSeparableConv and conv2 = depthwise 3x3 -> pointwise 1x1 ->
    BatchNorm2d (reference efficient_layers.py:109-151, use_efficient=True, the only form the networks build)."""

    def __init__(self, channels: int, use_efficient: bool = True):
        super().__init__()
        if not use_efficient:
            raise NotImplementedError("libnvq implements the depthwise-separable form (use_efficient=True)")
        self.channels = channels
        self.conv1 = DepthwiseSeparableConv(channels, channels)
        self.conv2 = Stack(nn.Conv2d(channels, channels, 3, 1, 1, groups=channels, bias=False),
                           nn.Conv2d(channels, channels, 1, 1, 0, bias=False), nn.BatchNorm2d(channels))
        self.relu = Act()
        self.math_mode = MATH_F32

    def forward_nhwc(self, x: torch.Tensor, math: int = MATH_F32) -> torch.Tensor:
        y = self.conv1.forward_nhwc(x, math)
        y = _ops.DwConv.apply(y, self.conv2[0].weight)
        y = _ops.Conv.apply(y, self.conv2[1].weight, None, False, math)
        return _ops.bn(y, self.conv2[2], self.training, relu=True, res=x)     # relu(bn(.) + identity)

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        return _nchw_call(lambda t: self.forward_nhwc(t, self.math_mode), x, self.channels)


class ChannelAttention(_Holder):
    """GAP -> Linear(C, C/r) -> ReLU -> Linear(C/r, C) -> sigmoid (reference :154-180); computed inside CBAM."""

    def __init__(self, channels: int, reduction: int = 16):
        super().__init__()
        self.avg_pool = Act()
        self.fc = Stack(nn.Linear(channels, channels // reduction, bias=False), Act(),
                        nn.Linear(channels // reduction, channels, bias=False), Act())


class SpatialAttention(_Holder):
    """conv7x7([mean_c, max_c]) -> sigmoid (reference :183-205); computed inside CBAM."""

    def __init__(self, kernel_size: int = 7):
        super().__init__()
        if kernel_size != 7:
            raise NotImplementedError("libnvq implements the 7x7 spatial attention used by CBAM")
        self.conv = nn.Conv2d(2, 1, 7, padding=3, bias=False)
        self.sigmoid = Act()


class CBAM(_Holder):
    """channel attention then spatial attention (reference :208-228)."""

    def __init__(self, channels: int, reduction: int = 16):
        super().__init__()
        self.channels = channels
        self.channel_attention = ChannelAttention(channels, reduction)
        self.spatial_attention = SpatialAttention()

    def forward_nhwc(self, x: torch.Tensor) -> torch.Tensor:
        c = self.channels
        if c < 16 or c > 256 or c & (c - 1) or c // self.channel_attention.fc[0].weight.shape[0] < 1:
            raise NotImplementedError("libnvq's CBAM kernels need a power-of-two channel count in [16, 256]")
        return _ops.CBAMFn.apply(x, self.channel_attention.fc[0].weight, self.channel_attention.fc[2].weight,
                                 self.spatial_attention.conv.weight)

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        return _nchw_call(self.forward_nhwc, x, self.channels)


class TemporalConv3D(_Holder):
    """(2+1)D factorised 3-D convolution (reference efficient_layers.py:231-294): Conv3d(1,3,3) + BatchNorm3d + ReLU, then
    Conv3d(T,1,1) + BatchNorm3d + ReLU; the intermediate width follows the reference's formula (:253-257)."""

    def __init__(self, in_channels: int, out_channels: int, temporal_kernel: int = 3):
        super().__init__()
        if temporal_kernel != 3:
            raise NotImplementedError("libnvq implements temporal_kernel=3")
        mid = (in_channels * out_channels * 3 * 3 * temporal_kernel) // (in_channels * 3 * 3 + out_channels * temporal_kernel)
        mid = max(mid, out_channels // 2)
        self.in_channels, self.mid_channels, self.out_channels = in_channels, mid, out_channels
        self.spatial = Stack(nn.Conv3d(in_channels, mid, (1, 3, 3), 1, (0, 1, 1), bias=False), nn.BatchNorm3d(mid), Act())
        self.temporal = Stack(nn.Conv3d(mid, out_channels, (temporal_kernel, 1, 1), 1, (temporal_kernel // 2, 0, 0), bias=False),
                              nn.BatchNorm3d(out_channels), Act())
        self.math_mode = MATH_F32

    def forward_nhwc(self, x: torch.Tensor, T: int, math: int = MATH_F32, act_dtype=None) -> torch.Tensor:
        """x: time-major image batch [T*B, H, W, ld]; act_dtype: storage type of the block's activations (default: x's)"""
        w = self.spatial[0].weight
        y = _ops.Conv.apply(x, w.view(w.shape[0], w.shape[1], 3, 3), None, False, math, act_dtype)
        y = _ops.bn(y, self.spatial[1], self.training, relu=True)
        y = _ops.TemporalConv.apply(y, self.temporal[0].weight, T, math)
        return _ops.bn(y, self.temporal[1], self.training, relu=True)

    def forward_tc(self, x: torch.Tensor, T: int, math: int = MATH_F32, act_dtype=None) -> torch.Tensor:
        """x: time-in-channels tensor [B, H, W, T * Cp] (see nerve_cl._ops); T >= 2.  Every activation is read and written
        once: no accumulating passes over memory for the temporal taps."""
        w = self.spatial[0].weight
        y = _ops.SpatialConvTC.apply(x, w.view(w.shape[0], w.shape[1], 3, 3), T, math, act_dtype)
        y = _ops.bn_tc(y, self.spatial[1], T, self.training, relu=True)
        y = _ops.TemporalConvTC.apply(y, self.temporal[0].weight, T, math)
        return _ops.bn_tc(y, self.temporal[1], T, self.training, relu=True)

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        """(B,C,T,H,W) -> (B,C',T,H,W)"""
        _nvq.require_device(x, "input")
        B, C, T, H, W = x.shape
        with _nvq.device_guard(x.device):
            xt = x.permute(2, 0, 1, 3, 4).reshape(T * B, C, H, W)
            y = self.forward_nhwc(_ops.ToNHWC.apply(xt), T, self.math_mode)
            out = _ops.ToNCHW.apply(y, self.out_channels)
        return out.view(T, B, self.out_channels, H, W).permute(1, 2, 0, 3, 4)


class LiteFlowNetCorrelation(_Holder):
    """9x9 local correlation, no parameters (reference :297-343); forward through the exact-fp32 correlation kernel."""

    def __init__(self, max_displacement: int = 4):
        super().__init__()
        if max_displacement != 4:
            raise NotImplementedError("libnvq implements max_displacement=4 (81 channels)")
        self.max_displacement = max_displacement
        self.pad = max_displacement

    def forward(self, x1: torch.Tensor, x2: torch.Tensor) -> torch.Tensor:
        """(B,C,H,W) x 2 -> (B,81,H,W)"""
        _nvq.require_device(x1, "x1")
        with _nvq.device_guard(x1.device):
            return _ops.ToNCHW.apply(_ops.Correlation.apply(_ops.ToNHWC.apply(x1), _ops.ToNHWC.apply(x2), x1.shape[1]), 81)
