"""Closed-form test tensors (TEST INFRASTRUCTURE, see oracle/sr_oracle.py header).

Weights and inputs for the golden fixtures come from a sine hash of the flat
index, so they need not be stored and do not depend on any RNG or on module
construction order.
"""
from __future__ import annotations

import zlib
from typing import Dict

import numpy as np
import torch

from . import sr_oracle

GOLDEN_GAIN = 0.6   # weight gain used by every fixture under tests/golden/


def hash01(n: int, seed: int) -> np.ndarray:
    """n pseudo-random float64 values in [0,1), deterministic across machines."""
    i = np.arange(n, dtype=np.float64)
    v = np.sin(i * 12.9898 + (seed % 1000) * 78.233 + 0.5) * 43758.5453
    return v - np.floor(v)


def name_seed(name: str) -> int:
    return zlib.crc32(name.encode()) % 997


def formula_state(in_channels=3, scale_factor=2, num_features=64, num_residual_blocks=8,
                  temporal_window=1, gain: float = 1.0) -> Dict[str, torch.Tensor]:
    """Full state_dict (params + BN buffers), fp32.

    Conv / linear weights ~ U(-a, a) with a = gain*sqrt(3/fan_in) (unit output
    variance for unit-variance input), biases U(-0.1, 0.1), BN weight in
    [0.8, 1.2], BN bias in [-0.1, 0.1], running_mean in [-0.2, 0.2], running_var
    in [0.5, 1.5], num_batches_tracked = 3."""
    shapes = sr_oracle.param_shapes(in_channels, scale_factor, num_features,
                                    num_residual_blocks, temporal_window)
    return _fill(shapes, sr_oracle.buffer_shapes(num_features), gain)


def formula_state_light(scale_factor: int = 2, gain: float = 1.0) -> Dict[str, torch.Tensor]:
    """Same recipe for LightweightSuperResolution's state_dict."""
    return _fill(sr_oracle.light_param_shapes(scale_factor), sr_oracle.light_buffer_shapes(), gain)


def _fill(shapes, buffers, gain) -> Dict[str, torch.Tensor]:
    sd: Dict[str, torch.Tensor] = {}
    for name, shp in shapes.items():
        n = int(np.prod(shp))
        u = hash01(n, name_seed(name)) * 2 - 1
        if name.endswith("bn.weight"):
            v = 1.0 + 0.2 * u
        elif name.endswith("bias"):
            v = 0.1 * u
        else:
            fan_in = int(np.prod(shp[1:]))
            v = gain * np.sqrt(3.0 / fan_in) * u
        sd[name] = torch.from_numpy(v.reshape(shp).astype(np.float32))
    for name, shp in buffers.items():
        if name.endswith("num_batches_tracked"):
            sd[name] = torch.tensor(3, dtype=torch.long)
            continue
        n = int(np.prod(shp))
        u = hash01(n, name_seed(name))
        v = (0.4 * u - 0.2) if name.endswith("running_mean") else (0.5 + u)
        sd[name] = torch.from_numpy(v.reshape(shp).astype(np.float32))
    return sd


def formula_state_fr(in_channels: int = 3, base: int = 16, gain: float = 1.0) -> Dict[str, torch.Tensor]:
    """The same recipe for FrameRecoveryNet's state_dict (oracle/fr_oracle.py).  Its BatchNorm layers are Sequential
    members ("stem.1.weight"), so a BatchNorm weight is recognised by its shape: 1-D, not a bias."""
    from . import fr_oracle
    shapes, buffers = fr_oracle.shapes(in_channels, base)
    renamed = {(n[:-6] + "bn.weight" if (len(s) == 1 and n.endswith("weight") and not n.endswith("bn.weight")) else n): (n, s)
               for n, s in shapes.items()}
    filled = _fill({k: s for k, (_, s) in renamed.items()}, buffers, gain)
    out = {orig: filled[k] for k, (orig, _) in renamed.items()}
    out.update({n: filled[n] for n in buffers})
    return out


def formula_clip(B: int, T: int, H: int, W: int, C: int = 3, seed: int = 11) -> torch.Tensor:
    """(B,T,C,H,W) frames in [0,1): a smooth moving pattern plus hash noise, so
    that correlation/flow see real inter-frame structure."""
    b = np.arange(B)[:, None, None, None, None]
    t = np.arange(T)[None, :, None, None, None]
    c = np.arange(C)[None, None, :, None, None]
    y = np.arange(H)[None, None, None, :, None]
    x = np.arange(W)[None, None, None, None, :]
    smooth = 0.5 + 0.25 * np.sin(0.55 * (x + 0.7 * t) + 0.3 * c + 0.9 * b) \
        * np.cos(0.45 * (y - 0.4 * t) + 0.2 * c)
    noise = hash01(B * T * C * H * W, seed).reshape(B, T, C, H, W)
    return torch.from_numpy((0.7 * smooth + 0.3 * noise).astype(np.float32))


def formula_target(B: int, H: int, W: int, C: int = 3, seed: int = 23) -> torch.Tensor:
    return torch.from_numpy(hash01(B * C * H * W, seed).reshape(B, C, H, W).astype(np.float32))
