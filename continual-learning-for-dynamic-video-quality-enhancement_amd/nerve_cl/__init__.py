"""nerve_cl on MI355X: the super-resolution hot path of NERVE-CL as HIP kernels (libnvq.so)
behind the reference's Python surface.  See DESIGN.md and INTEGRATION.md at the repo root."""
__version__ = "0.1.0"

from nerve_cl.models import SuperResolutionNet, EnhancementEngine, EnhancementConfig

__all__ = ["SuperResolutionNet", "EnhancementEngine", "EnhancementConfig"]
