"""Parameter holders for the building blocks of the SR hot path.

The reference implements these as torch.nn modules with their own forward
(nerve_cl/models/layers/efficient_layers.py:9-343).  Here they only OWN the parameters,
under the reference's attribute names so that ``state_dict()`` keys, shapes, dtypes and
default initialisation (same construction order => same RNG stream) are identical; the
arithmetic lives in libnvq and is scheduled by ``nerve_cl._engine``.  Calling one of these
holders directly is refused: there is no PyTorch fallback path.
"""
from __future__ import annotations

import torch.nn as nn


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
        if (kernel_size, stride, padding, bias) != (3, 1, 1, False) or in_channels != out_channels:
            raise NotImplementedError("libnvq implements the 3x3 / stride 1 / C->C / no-bias form used by the SR net")
        self.depthwise = nn.Conv2d(in_channels, in_channels, 3, 1, 1, groups=in_channels, bias=False)
        self.pointwise = nn.Conv2d(in_channels, out_channels, 1, 1, 0, bias=False)
        self.bn = nn.BatchNorm2d(out_channels)
        self.act = Act()


class PixelShuffleUpsampler(_Holder):
    """conv3x3(C -> out*s^2) + PixelShuffle(s) (reference efficient_layers.py:70-106)."""

    def __init__(self, in_channels: int, scale_factor: int = 2, out_channels: int = 3):
        super().__init__()
        self.scale_factor = scale_factor
        self.conv = nn.Conv2d(in_channels, out_channels * scale_factor ** 2, 3, 1, 1)
        self.pixel_shuffle = Act()


class ChannelAttention(_Holder):
    """GAP -> Linear(C, C/r) -> ReLU -> Linear(C/r, C) -> sigmoid (reference :154-180)."""

    def __init__(self, channels: int, reduction: int = 16):
        super().__init__()
        self.avg_pool = Act()
        self.fc = Stack(nn.Linear(channels, channels // reduction, bias=False), Act(),
                        nn.Linear(channels // reduction, channels, bias=False), Act())


class SpatialAttention(_Holder):
    """conv7x7([mean_c, max_c]) -> sigmoid (reference :183-205)."""

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
        self.channel_attention = ChannelAttention(channels, reduction)
        self.spatial_attention = SpatialAttention()


class LiteFlowNetCorrelation(_Holder):
    """9x9 local correlation, no parameters (reference :297-343)."""

    def __init__(self, max_displacement: int = 4):
        super().__init__()
        if max_displacement != 4:
            raise NotImplementedError("libnvq implements max_displacement=4 (81 channels)")
        self.max_displacement = max_displacement
        self.pad = max_displacement
