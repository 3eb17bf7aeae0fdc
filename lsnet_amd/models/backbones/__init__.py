from .resnet import ResNet, ResNeXt

__all__ = ['ResNet', 'ResNeXt']
