from .backbones import ResNet, ResNeXt
from .builder import (BACKBONES, DETECTORS, HEADS, LOSSES, NECKS, build_backbone, build_detector, build_head,
                      build_loss, build_neck)
from .dense_heads import LSHead
from .detectors import BaseDetector, LSDetector, SingleStageDetector
from .losses import CrossIOULoss, FocalLoss
from .necks import FPN

__all__ = ['BACKBONES', 'NECKS', 'HEADS', 'LOSSES', 'DETECTORS', 'build_backbone', 'build_neck', 'build_head',
           'build_loss', 'build_detector', 'ResNet', 'ResNeXt', 'FPN', 'LSHead', 'CrossIOULoss', 'FocalLoss',
           'BaseDetector', 'SingleStageDetector', 'LSDetector']
