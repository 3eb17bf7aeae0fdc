from .ls_head import DCNConvModule, LSHead

__all__ = ['LSHead', 'DCNConvModule']
