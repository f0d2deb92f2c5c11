"""FrameRecoveryNet is SURVEY section 8(f) row 1 (needed only by BASELINE config 4) and is not built yet: the name is kept
importable so that ``from nerve_cl.models import FrameRecoveryNet`` works, and constructing it fails loudly instead of
silently running somewhere else.  (In the reference its output never feeds the super-resolution result either:
enhancement_engine.py:143-148 hands the ORIGINAL frames to the SR net.)"""
import torch.nn as nn


class FrameRecoveryNet(nn.Module):
    def __init__(self, *args, **kwargs):
        super().__init__()
        raise NotImplementedError(
            "FrameRecoveryNet (reference nerve_cl/models/frame_recovery.py:335-446) has no MI355X implementation yet; "
            "use EnhancementConfig(frame_recovery_enabled=False)")
