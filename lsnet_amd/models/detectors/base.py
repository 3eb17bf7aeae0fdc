"""Detector base: the train/test call protocol of the reference
(mmdet/models/detectors/base.py:17-243) -- `forward(img, img_metas, return_loss=...)`,
`train_step(data, optimizer) -> {loss, log_vars, num_samples}`, `_parse_losses`.

Difference by design: `_parse_losses` keeps the logged scalars on the DEVICE (one fused
all-reduce for all of them when distributed) and converts to Python floats lazily, so a training
step has no device->host synchronisation; the reference calls `.item()` on every scalar every
iteration (base.py:202-207)."""
from collections import OrderedDict

import torch
import torch.distributed as dist
import torch.nn as nn


class LazyScalars(OrderedDict):
    """name -> 0-d device tensor; `float(d[name])` / `.item()` synchronises only when asked."""

    def items_as_float(self):
        vals = torch.stack([v.detach().float().reshape(()) for v in self.values()]).cpu().tolist()
        return OrderedDict(zip(self.keys(), vals))


class BaseDetector(nn.Module):

    def __init__(self):
        super().__init__()
        self.fp16_enabled = False

    with_neck = property(lambda self: getattr(self, 'neck', None) is not None)
    with_bbox = property(lambda self: getattr(self, 'bbox_head', None) is not None)

    def extract_feat(self, imgs):
        raise NotImplementedError

    def extract_feats(self, imgs):
        assert isinstance(imgs, list)
        return [self.extract_feat(img) for img in imgs]

    def init_weights(self, pretrained=None):
        pass

    def forward_train(self, imgs, img_metas, **kwargs):
        raise NotImplementedError

    def simple_test(self, img, img_metas, **kwargs):
        raise NotImplementedError

    def aug_test(self, imgs, img_metas, **kwargs):
        raise NotImplementedError('multi-scale / flip test-time augmentation is outside the hot path')

    def forward_test(self, imgs, img_metas, **kwargs):
        """imgs / img_metas are lists over test-time augmentations (base.py:117-159)."""
        for var, name in ((imgs, 'imgs'), (img_metas, 'img_metas')):
            if not isinstance(var, list):
                raise TypeError(f'{name} must be a list, but got {type(var)}')
        if len(imgs) != len(img_metas):
            raise ValueError(f'num of augmentations ({len(imgs)}) != num of image meta ({len(img_metas)})')
        if len(imgs) == 1:
            return self.simple_test(imgs[0], img_metas[0], **kwargs)
        return self.aug_test(imgs, img_metas, **kwargs)

    def forward(self, img, img_metas, return_loss=True, **kwargs):
        if return_loss:
            return self.forward_train(img, img_metas, **kwargs)
        return self.forward_test(img, img_metas, **kwargs)

    def _parse_losses(self, losses):
        """dict of tensors / lists of tensors -> (total loss, log_vars).  Keys containing 'loss'
        are summed into the total (base.py:176-209)."""
        log_vars = LazyScalars()

        def mean(v):     # the mean of a 0-dim tensor is the tensor: no launch for it
            return v if v.dim() == 0 else v.mean()

        def total(vals):   # Python's sum() without its leading `0 +` (exact either way, one launch less per list)
            vals = list(vals)
            if not vals:
                return 0
            acc = vals[0]
            for v in vals[1:]:
                acc = acc + v
            return acc
        for name, value in losses.items():
            if isinstance(value, torch.Tensor):
                log_vars[name] = mean(value)
            elif isinstance(value, list):
                base = getattr(value, 'base', None)     # ls_head.LevelTerms: the levels of the term are one (L,) tensor
                log_vars[name] = base.sum() if base is not None and len(value) > 1 else total(mean(v) for v in value)
            else:
                raise TypeError(f'{name} is not a tensor or list of tensors')
        loss = total(v for k, v in log_vars.items() if 'loss' in k)
        log_vars['loss'] = loss
        capturing = torch.cuda.is_available() and torch.cuda.is_current_stream_capturing()
        # `_log_reduce_elsewhere`: a GraphedForwardBackward owns the ONE reduction of the log scalars of its steps -- warm-up,
        # capture and replay alike, so that every rank issues the same sequence of collectives whatever its call count
        elsewhere = getattr(self, '_log_reduce_elsewhere', False)
        if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1 and not capturing and not elsewhere:
            packed = torch.stack([v.detach().reshape(()) for v in log_vars.values()])
            dist.all_reduce(packed.div_(dist.get_world_size()))   # one collective for all scalars
            for i, k in enumerate(list(log_vars.keys())):
                log_vars[k] = packed[i]
        else:
            for k in list(log_vars.keys()):
                log_vars[k] = log_vars[k].detach()
        return loss, log_vars

    def train_step(self, data, optimizer):
        losses = self(**data)
        loss, log_vars = self._parse_losses(losses)
        return dict(loss=loss, log_vars=log_vars, num_samples=len(data['img_metas']))

    def val_step(self, data, optimizer):
        return self.train_step(data, optimizer)
