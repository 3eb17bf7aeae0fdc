"""`DataContainer`, `collate` and `scatter`: how a list of per-image dicts becomes the keyword arguments of
`detector.forward` (mmcv/parallel/data_container.py:20-89, collate.py:10-84, scatter_gather.py:7-55).

Three kinds of fields:
  * `cpu_only`   -- meta data and polygon masks: stay python objects, one list per device group;
  * `stack`      -- images: padded bottom/right to the largest in the group with `padding_value`, then stacked;
  * neither      -- per-image ground truth of ragged length: a list of tensors.

One process drives one GPU here, so a batch holds exactly one group and `scatter` simply unwraps it onto the
device (the reference's `Scatter.forward` streams are replaced by `non_blocking` copies from pinned memory)."""
from collections.abc import Mapping, Sequence

import torch
import torch.nn.functional as F
from torch.utils.data.dataloader import default_collate


class DataContainer:

    def __init__(self, data, stack=False, padding_value=0, cpu_only=False, pad_dims=2):
        assert pad_dims in (None, 1, 2, 3)
        self._data, self._stack, self._padding_value = data, stack, padding_value
        self._cpu_only, self._pad_dims = cpu_only, pad_dims

    data = property(lambda self: self._data)
    stack = property(lambda self: self._stack)
    padding_value = property(lambda self: self._padding_value)
    cpu_only = property(lambda self: self._cpu_only)
    pad_dims = property(lambda self: self._pad_dims)

    @property
    def datatype(self):
        return self._data.type() if isinstance(self._data, torch.Tensor) else type(self._data)

    def size(self, *args, **kwargs):
        return self._data.size(*args, **kwargs)

    def dim(self):
        return self._data.dim()

    def __len__(self):
        return len(self._data)

    def __repr__(self):
        return f'{type(self).__name__}({self._data!r})'


def _stack_group(group):
    first = group[0]
    if first.pad_dims is None:
        return default_collate([s.data for s in group])
    nd, pd = first.dim(), first.pad_dims
    assert nd > pd
    for s in group:
        assert s.data.shape[:nd - pd] == first.data.shape[:nd - pd]
    target = [max(s.size(-d) for s in group) for d in range(1, pd + 1)]       # innermost dimension first
    out = []
    for s in group:
        pad = []
        for d in range(1, pd + 1):
            pad += [0, target[d - 1] - s.size(-d)]
        out.append(F.pad(s.data, pad, value=s.padding_value))
    return default_collate(out)


def collate(batch, samples_per_gpu=1):
    if not isinstance(batch, Sequence):
        raise TypeError(f'{type(batch)} is not supported.')
    head = batch[0]
    if isinstance(head, DataContainer):
        assert len(batch) % samples_per_gpu == 0
        groups = [batch[i:i + samples_per_gpu] for i in range(0, len(batch), samples_per_gpu)]
        if head.cpu_only:
            return DataContainer([[s.data for s in g] for g in groups], head.stack, head.padding_value, cpu_only=True)
        if head.stack:
            for g in groups:
                assert isinstance(g[0].data, torch.Tensor)
            return DataContainer([_stack_group(g) for g in groups], True, head.padding_value)
        return DataContainer([[s.data for s in g] for g in groups], False, head.padding_value)
    if isinstance(head, Sequence) and not isinstance(head, (str, bytes)):
        return [collate(samples, samples_per_gpu) for samples in zip(*batch)]
    if isinstance(head, Mapping):
        return {k: collate([d[k] for d in batch], samples_per_gpu) for k in head}
    return default_collate(batch)


def _to_device(obj, device, channels_last):
    if isinstance(obj, torch.Tensor):
        if device is None:
            return obj
        if obj.device.type == 'cpu' and torch.device(device).type == 'cuda' and not obj.is_pinned():
            obj = obj.pin_memory()
        obj = obj.to(device, non_blocking=True)
        if channels_last and obj.dim() == 4:
            obj = obj.contiguous(memory_format=torch.channels_last)
        return obj
    if isinstance(obj, (list, tuple)):
        return type(obj)(_to_device(o, device, channels_last) for o in obj)
    return obj


def scatter(data, device=None, group=0, channels_last=False):
    """Unwrap the `DataContainer`s of a collated batch for the device this process owns.  `group` selects the device
    group (always 0 with one process per GPU)."""
    if isinstance(data, DataContainer):
        part = data.data[group]
        return part if data.cpu_only else _to_device(part, device, channels_last)
    if isinstance(data, torch.Tensor):
        return _to_device(data, device, channels_last)
    if isinstance(data, Mapping):
        out = {k: scatter(v, device, group, channels_last) for k, v in data.items() if k != 'img'}
        if 'img' in data:
            img = data['img']
            deferred = (isinstance(img, DataContainer) and not img.stack and not img.cpu_only and len(img.data[group])
                        and isinstance(img.data[group][0], torch.Tensor) and img.data[group][0].dtype == torch.uint8)
            if deferred:      # decoded 8-bit images + recorded plans -> the float batch, made on the target device
                from ..data.device_prep import prepare_batch
                out['img'] = prepare_batch(img.data[group], out['img_metas'], device)
            else:
                out['img'] = scatter(img, device, group, channels_last)
        return out
    if isinstance(data, (list, tuple)) and not isinstance(data, (str, bytes)):
        return type(data)(scatter(v, device, group, channels_last) for v in data)
    return data
