"""Parameter-holder layers of the SR hot path (see efficient_layers.py)."""
from nerve_cl.models.layers.efficient_layers import (
    DepthwiseSeparableConv,
    PixelShuffleUpsampler,
    ResidualBlock,
    ChannelAttention,
    SpatialAttention,
    CBAM,
    TemporalConv3D,
    LiteFlowNetCorrelation,
    Stack,
    Act,
)

__all__ = [
    "DepthwiseSeparableConv", "PixelShuffleUpsampler", "ResidualBlock", "ChannelAttention", "SpatialAttention",
    "CBAM", "TemporalConv3D", "LiteFlowNetCorrelation", "Stack", "Act",
]
