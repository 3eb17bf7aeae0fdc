from .backend import get_backend, register_backend, unregister_backend
from .dcn import (DeformConv, DeformConvPack, ModulatedDeformConv, ModulatedDeformConvPack, PyramidDeformConv,
                  dcn_multi, deform_conv, modulated_deform_conv, pyramid_deform_conv)
from .focal import SigmoidFocalLoss, sigmoid_focal_loss, sigmoid_focal_loss_sum
from .nms import batched_nms, nms, nms_match, soft_nms

__all__ = [
    'DeformConv', 'DeformConvPack', 'ModulatedDeformConv', 'ModulatedDeformConvPack', 'PyramidDeformConv',
    'deform_conv', 'modulated_deform_conv', 'pyramid_deform_conv', 'dcn_multi', 'SigmoidFocalLoss',
    'sigmoid_focal_loss', 'sigmoid_focal_loss_sum', 'nms', 'batched_nms', 'soft_nms', 'nms_match', 'get_backend', 'register_backend',
    'unregister_backend'
]
