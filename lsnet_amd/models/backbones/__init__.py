from .res2net import Res2Net
from .resnet import ResNet, ResNeXt

__all__ = ['ResNet', 'ResNeXt', 'Res2Net']
