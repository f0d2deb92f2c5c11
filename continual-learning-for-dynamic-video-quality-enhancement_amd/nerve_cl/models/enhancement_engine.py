"""EnhancementEngine: the dict-returning wrapper ("plugin surface") around the SR net.

Mirrors the reference's interface (nerve_cl/models/enhancement_engine.py:18-292):
``EnhancementConfig`` fields and defaults, ``EnhancementEngine(config)`` attributes
(``config``, ``frame_recovery``, ``super_resolution``, ``enhancement_strength``) and
``forward(frames, center_idx, corruption_mask, enhancement_strength) -> dict`` with keys
'enhanced' (always), 'super_resolved', 'recovered'.  Both branches run as libnvq HIP kernels: the
super-resolution network is the hot path, the frame-recovery head its first "next" row (8f).
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import Any, Dict, Optional

import torch
import torch.nn as nn

from nerve_cl import _nvq
from nerve_cl.models.frame_recovery import FrameRecoveryNet
from nerve_cl.models.super_resolution import LightweightSuperResolution, SuperResolutionNet


class _BlendFn(torch.autograd.Function):
    """strength * sr + (1 - strength) * bicubic(frame): one kernel; the frame carries no gradient."""

    @staticmethod
    def forward(ctx, sr, frames, t_center, scale, strength):
        out = torch.empty_like(sr)
        _nvq.bicubic_blend(sr.contiguous(), frames, t_center, scale, strength, out)
        ctx.strength = strength
        return out

    @staticmethod
    def backward(ctx, g):
        return g * ctx.strength, None, None, None, None


@dataclass
class EnhancementConfig:
    """Same fields and defaults as the reference (enhancement_engine.py:22-37)."""
    frame_recovery_enabled: bool = True
    recovery_base_channels: int = 64
    recovery_temporal_window: int = 2
    super_resolution_enabled: bool = True
    scale_factor: int = 2
    sr_num_features: int = 64
    sr_num_residual_blocks: int = 8
    sr_temporal_window: int = 1
    use_lightweight_sr: bool = False
    enhancement_mode: str = "sequential"
    upscale_first: bool = False


class EnhancementEngine(nn.Module):
    def __init__(self, config: Optional[EnhancementConfig] = None):
        super().__init__()
        self.config = config or EnhancementConfig()
        # same construction order as the reference (:65-93), so default initialisation draws the same RNG stream
        if self.config.frame_recovery_enabled:
            self.frame_recovery = FrameRecoveryNet(base_channels=self.config.recovery_base_channels,
                                                   temporal_window=self.config.recovery_temporal_window)
        else:
            self.frame_recovery = None
        if self.config.super_resolution_enabled:
            if self.config.use_lightweight_sr:
                self.super_resolution = LightweightSuperResolution(scale_factor=self.config.scale_factor)
            else:
                self.super_resolution = SuperResolutionNet(
                    scale_factor=self.config.scale_factor,
                    num_features=self.config.sr_num_features,
                    num_residual_blocks=self.config.sr_num_residual_blocks,
                    temporal_window=self.config.sr_temporal_window,
                )
        else:
            self.super_resolution = None
        self.enhancement_strength = nn.Parameter(torch.ones(1))

    def forward(self, frames: torch.Tensor, center_idx: Optional[int] = None,
                corruption_mask: Optional[torch.Tensor] = None,
                enhancement_strength: Optional[float] = None) -> Dict[str, torch.Tensor]:
        B, T, C, H, W = frames.shape
        if center_idx is None:
            center_idx = T // 2
        results: Dict[str, torch.Tensor] = {}
        current = frames[:, center_idx]
        if self.frame_recovery is not None and corruption_mask is not None:
            # same guard as the reference (:130-131): recovery only runs on a non-empty mask (one host sync, as there)
            if bool(corruption_mask.sum() > 0):
                refs = frames[:, [i for i in range(T) if i != center_idx]]
                results["recovered"] = self.frame_recovery(corrupted_frame=current, reference_frames=refs,
                                                           corruption_mask=corruption_mask)
                current = results["recovered"]
                # (as in the reference, :139-163: the lightweight single-frame SR net is fed the recovered frame, the
                # temporal SR net still reads the ORIGINAL frames; with SR disabled 'enhanced' is the recovered frame)
        if self.super_resolution is not None:
            w = self.config.sr_temporal_window
            lo, hi = max(0, center_idx - w), min(T, center_idx + w + 1)
            sr_frames = frames[:, lo:hi]
            need = 2 * w + 1
            if sr_frames.shape[1] < need:   # right-pad by repeating the last frame (:152-158)
                sr_frames = torch.cat(
                    [sr_frames, sr_frames[:, -1:].expand(-1, need - sr_frames.shape[1], -1, -1, -1)], dim=1)
            if isinstance(self.super_resolution, LightweightSuperResolution):
                current = self.super_resolution(current)      # single-frame net (:161-162)
            else:
                current = self.super_resolution(sr_frames)
            results["super_resolved"] = current
        strength = enhancement_strength if enhancement_strength is not None else self.enhancement_strength.item()
        if strength < 1.0 and "super_resolved" in results:
            # blend with the bicubic-upsampled original (reference :172-180)
            current = _BlendFn.apply(current, frames.detach().to(torch.float32).contiguous(), center_idx,
                                     self.config.scale_factor, float(strength))
        results["enhanced"] = current
        return results

    def enhance_video(self, video: torch.Tensor, corruption_masks: Optional[torch.Tensor] = None,
                      batch_size: int = 4, cache_features: bool = True) -> torch.Tensor:
        """Sliding-window enhancement of a whole clip, (T,C,H,W) or (B,T,C,H,W) -> same rank, upscaled
        (reference :186-248: window = 2*max(recovery window, sr window)+1, clipped at the clip boundaries).
        cache_features (eval mode, gradients off, temporal SR net): every frame's features are extracted once for the whole
        clip instead of once per window that contains it; same results."""
        squeeze = video.dim() == 4
        if squeeze:
            video = video.unsqueeze(0)
        B, T, C, H, W = video.shape
        win = 2 * max(self.config.recovery_temporal_window, self.config.sr_temporal_window) + 1
        sr = self.super_resolution
        cached = (cache_features and isinstance(sr, SuperResolutionNet) and not self.training and not torch.is_grad_enabled()
                  and corruption_masks is None)
        feats = sr.extract_features(video) if cached else None
        frames_out = []
        for t in range(T):
            lo, hi = max(0, t - win // 2), min(T, t + win // 2 + 1)
            if cached:
                # the frames forward() would hand to the SR net: centre +- sr window inside [lo, hi), right-padded
                w = self.config.sr_temporal_window
                idx = list(range(max(lo, t - w), min(hi, t + w + 1)))
                idx += [idx[-1]] * (2 * w + 1 - len(idx))
                out = sr.forward_cached(video[:, idx], feats[idx])
                strength = self.enhancement_strength.item()
                if strength < 1.0:
                    out = _BlendFn.apply(out, video[:, lo:hi].detach().to(torch.float32).contiguous(), t - lo,
                                         self.config.scale_factor, float(strength))
                frames_out.append(out)
                continue
            mask = corruption_masks[t:t + 1] if corruption_masks is not None else None
            frames_out.append(self.forward(video[:, lo:hi], center_idx=t - lo, corruption_mask=mask)["enhanced"])
        out = torch.stack(frames_out, dim=1)
        return out.squeeze(0) if squeeze else out

    def get_model_info(self) -> Dict[str, Any]:
        info: Dict[str, Any] = {
            "config": {
                "frame_recovery_enabled": self.config.frame_recovery_enabled,
                "super_resolution_enabled": self.config.super_resolution_enabled,
                "scale_factor": self.config.scale_factor,
                "use_lightweight_sr": self.config.use_lightweight_sr,
            },
            "parameters": {
                "total": sum(p.numel() for p in self.parameters()),
                "trainable": sum(p.numel() for p in self.parameters() if p.requires_grad),
            },
        }
        if self.frame_recovery is not None:
            info["parameters"]["frame_recovery"] = sum(p.numel() for p in self.frame_recovery.parameters())
        if self.super_resolution is not None:
            info["parameters"]["super_resolution"] = sum(p.numel() for p in self.super_resolution.parameters())
        return info

    def set_enhancement_mode(self, mode: str) -> None:
        if mode == "full":
            self.config.frame_recovery_enabled, self.config.super_resolution_enabled = True, True
        elif mode == "recovery_only":
            self.config.frame_recovery_enabled, self.config.super_resolution_enabled = True, False
        elif mode == "sr_only":
            self.config.frame_recovery_enabled, self.config.super_resolution_enabled = False, True
        elif mode == "lightweight":
            self.config.frame_recovery_enabled, self.config.super_resolution_enabled = False, True
            self.config.use_lightweight_sr = True



class AdaptiveEnhancementEngine(EnhancementEngine):
    """Enhancement strength / mode chosen from a content-complexity estimate, a resource budget and a user preference
    (reference enhancement_engine.py:295-381).  The estimator is a 3x8x8 -> 64 -> 1 MLP on the pooled centre frame - a few
    hundred FLOPs, left to stock PyTorch ops; the enhancement itself is the HIP path of EnhancementEngine."""

    def __init__(self, config: Optional[EnhancementConfig] = None):
        super().__init__(config)
        self.complexity_estimator = nn.Sequential(nn.AdaptiveAvgPool2d(8), nn.Flatten(), nn.Linear(3 * 8 * 8, 64),
                                                  nn.ReLU(inplace=True), nn.Linear(64, 1), nn.Sigmoid())

    def estimate_complexity(self, frame: torch.Tensor) -> torch.Tensor:
        """(B,C,H,W) -> (B,1) in [0,1]."""
        return self.complexity_estimator(frame)

    def adaptive_forward(self, frames: torch.Tensor, resource_budget: float = 1.0,
                         user_quality_preference: float = 0.5) -> Dict[str, torch.Tensor]:
        B, T, C, H, W = frames.shape
        complexity = self.estimate_complexity(frames[:, T // 2])
        strength = 0.3 * resource_budget + 0.3 * user_quality_preference + 0.4 * complexity.mean().item()
        strength = min(1.0, max(0.3, strength))
        # like the reference, the mode only flips config flags; the modules built in __init__ stay as they are
        self.set_enhancement_mode("lightweight" if resource_budget < 0.3 else "sr_only" if resource_budget < 0.6 else "full")
        results = self.forward(frames=frames, enhancement_strength=strength)
        results["complexity"] = complexity
        results["enhancement_strength"] = strength
        return results
