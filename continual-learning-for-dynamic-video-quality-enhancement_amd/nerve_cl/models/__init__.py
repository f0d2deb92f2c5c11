"""Models of the NERVE-CL hot path, MI355X-native (reference nerve_cl/models/__init__.py:3-24)."""
from nerve_cl.models.super_resolution import SuperResolutionNet, LightweightSuperResolution
from nerve_cl.models.enhancement_engine import EnhancementEngine, EnhancementConfig

__all__ = ["SuperResolutionNet", "LightweightSuperResolution", "EnhancementEngine", "EnhancementConfig"]
