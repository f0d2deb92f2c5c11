"""Continual-learning components around the SR hot path (reference nerve_cl/continual/__init__.py: the same names).
EWC and SynapticIntelligence run their penalty / Fisher arithmetic as flat-bucket HIP kernels; the others are host-side
loops that only need forward / backward / deepcopy of the model."""
from nerve_cl.continual.memory import EpisodicMemory, StreamingEpisodicMemory
from nerve_cl.continual.ewc import EWC, OnlineEWC, SynapticIntelligence
from nerve_cl.continual.maml import MAML, FOMAML, Reptile, ContentAdaptiveMAML
from nerve_cl.continual.distillation import DistillationLoss, ContinualDistillation

__all__ = ["EpisodicMemory", "StreamingEpisodicMemory", "EWC", "OnlineEWC", "SynapticIntelligence", "MAML", "FOMAML",
           "Reptile", "ContentAdaptiveMAML", "DistillationLoss", "ContinualDistillation"]
