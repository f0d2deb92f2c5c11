"""Models of the NERVE-CL hot path, MI355X-native (reference nerve_cl/models/__init__.py:3-24: the same names)."""
from nerve_cl.models.frame_recovery import FrameRecoveryNet
from nerve_cl.models.super_resolution import SuperResolutionNet, LightweightSuperResolution
from nerve_cl.models.enhancement_engine import EnhancementEngine, AdaptiveEnhancementEngine, EnhancementConfig

__all__ = ["FrameRecoveryNet", "SuperResolutionNet", "LightweightSuperResolution", "EnhancementEngine",
           "AdaptiveEnhancementEngine", "EnhancementConfig"]
