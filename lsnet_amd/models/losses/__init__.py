from .cross_iou_loss import CrossIOULoss, cross_iou_loss
from .cpv_losses import GaussianFocalLoss, SEPFocalLoss, SmoothL1Loss, gaussian_focal_loss, smooth_l1_loss
from .focal_loss import FocalLoss, sigmoid_focal_loss
from .utils import reduce_loss, weight_reduce_loss, weighted_loss

__all__ = ['CrossIOULoss', 'cross_iou_loss', 'FocalLoss', 'sigmoid_focal_loss', 'reduce_loss',
           'weight_reduce_loss', 'weighted_loss', 'GaussianFocalLoss', 'SEPFocalLoss', 'SmoothL1Loss',
           'gaussian_focal_loss', 'smooth_l1_loss']
