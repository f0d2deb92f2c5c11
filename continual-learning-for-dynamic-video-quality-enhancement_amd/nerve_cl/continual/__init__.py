"""Continual-learning components around the SR hot path (reference nerve_cl/continual/__init__.py).
EWC is the accelerated one (flat-bucket HIP kernels); the others are small host-side helpers kept so that
`from nerve_cl.continual import EpisodicMemory, EWC, FOMAML, ContinualDistillation` works."""
from nerve_cl.continual.memory import EpisodicMemory
from nerve_cl.continual.ewc import EWC, OnlineEWC
from nerve_cl.continual.maml import FOMAML
from nerve_cl.continual.distillation import DistillationLoss, ContinualDistillation

__all__ = ["EpisodicMemory", "EWC", "OnlineEWC", "FOMAML", "DistillationLoss", "ContinualDistillation"]
