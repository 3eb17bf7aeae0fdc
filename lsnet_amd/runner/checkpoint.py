"""Checkpoint IO with the reference's on-disk contract (mmcv/runner/checkpoint.py:154-300):
`{meta, state_dict, optimizer}`, state-dict keys without the DDP `module.` prefix, CPU tensors."""
import os
import time
from collections import OrderedDict

import torch


def _strip_module(state_dict):
    return OrderedDict((k[7:] if k.startswith('module.') else k, v) for k, v in state_dict.items())


def load_state_dict(module, state_dict, strict=False, logger=None):
    missing, unexpected = module.load_state_dict(state_dict, strict=False)
    missing = [k for k in missing if 'num_batches_tracked' not in k]
    msgs = []
    if unexpected:
        msgs.append('unexpected key in source state_dict: ' + ', '.join(unexpected))
    if missing:
        msgs.append('missing keys in source state_dict: ' + ', '.join(missing))
    if msgs:
        text = 'The model and loaded state dict do not match exactly\n' + '\n'.join(msgs)
        if strict:
            raise RuntimeError(text)
        (logger.warning if logger is not None else print)(text)


def load_checkpoint(model, filename, map_location='cpu', strict=False, logger=None):
    if filename.startswith(('torchvision://', 'open-mmlab://', 'http://', 'https://')):
        raise IOError(f'{filename}: there is no network here; pass a local checkpoint path')
    if not os.path.isfile(filename):
        raise IOError(f'{filename} is not a checkpoint file')
    ckpt = torch.load(filename, map_location=map_location, weights_only=False)
    if isinstance(ckpt, OrderedDict) or 'state_dict' not in ckpt:
        state = ckpt
    else:
        state = ckpt['state_dict']
    target = model.module if hasattr(model, 'module') else model
    load_state_dict(target, _strip_module(state), strict, logger)
    from ..ops.conv import invalidate_weight_images
    invalidate_weight_images()    # the parameters changed under the prepared weight images (ops/conv.py)
    return ckpt


def save_checkpoint(model, filename, optimizer=None, meta=None):
    meta = dict(meta or {})
    meta.update(time=time.asctime())
    target = model.module if hasattr(model, 'module') else model
    ckpt = dict(meta=meta, state_dict=OrderedDict((k, v.cpu()) for k, v in target.state_dict().items()))
    if optimizer is not None:
        ckpt['optimizer'] = optimizer.state_dict()
    os.makedirs(os.path.dirname(os.path.abspath(filename)), exist_ok=True)
    torch.save(ckpt, filename)
