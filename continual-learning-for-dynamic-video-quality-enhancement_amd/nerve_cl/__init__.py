"""nerve_cl on MI355X: the super-resolution hot path of NERVE-CL as HIP kernels (libnvq.so)
behind the reference's Python surface (reference nerve_cl/__init__.py:24-44: the same top-level names).
See DESIGN.md and INTEGRATION.md at the repo root."""
__version__ = "0.1.0"

from nerve_cl.models import (FrameRecoveryNet, SuperResolutionNet, LightweightSuperResolution, EnhancementEngine,
                             AdaptiveEnhancementEngine, EnhancementConfig)
from nerve_cl.continual import EpisodicMemory, EWC, MAML

__all__ = ["FrameRecoveryNet", "SuperResolutionNet", "LightweightSuperResolution", "EnhancementEngine",
           "AdaptiveEnhancementEngine", "EnhancementConfig", "EpisodicMemory", "EWC", "MAML"]
