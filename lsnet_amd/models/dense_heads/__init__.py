from .ls_head import DCNConvModule, LSHead
from .lscpv_head import LSCPVHead

__all__ = ['LSHead', 'LSCPVHead', 'DCNConvModule']
