from .base import BaseDetector
from .lscpv import LSCPVDetector
from .lsnet import LSDetector, SingleStageDetector

__all__ = ['BaseDetector', 'SingleStageDetector', 'LSDetector', 'LSCPVDetector']
