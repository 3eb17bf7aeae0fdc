from . import mask

__all__ = ['mask']
