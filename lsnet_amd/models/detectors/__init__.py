from .base import BaseDetector
from .lsnet import LSDetector, SingleStageDetector

__all__ = ['BaseDetector', 'SingleStageDetector', 'LSDetector']
