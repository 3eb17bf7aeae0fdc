"""COCO-shaped synthetic training batches: images N(0,1) padded to /32,
~7 boxes per image with log-uniform sizes, labels, extreme points on the box edges, 36-vertex
polygons and 17 keypoints -- the tensors LSDetector.forward_train consumes.  Everything is built
on the host once and moved to the device; the timed loop never touches the host."""
import math
import types

import numpy as np
import torch


def _boxes(g, num, h, w, min_size=16., max_size=600.):
    cx, cy = torch.rand(num, generator=g) * w, torch.rand(num, generator=g) * h
    lo, hi = math.log(min_size), math.log(min(max_size, 0.9 * min(h, w)))
    bw = torch.exp(torch.rand(num, generator=g) * (hi - lo) + lo)
    bh = torch.exp(torch.rand(num, generator=g) * (hi - lo) + lo)
    x1, y1 = (cx - bw / 2).clamp(0, w - 4), (cy - bh / 2).clamp(0, h - 4)
    x2, y2 = torch.min((cx + bw / 2), torch.tensor(float(w))), torch.min((cy + bh / 2), torch.tensor(float(h)))
    return torch.stack([x1, y1, torch.max(x2, x1 + 4), torch.max(y2, y1 + 4)], 1)


def _extremes(g, b):
    u = torch.rand(b.shape[0], 4, generator=g)
    x1, y1, x2, y2 = b.unbind(1)
    return torch.stack([x1 + u[:, 0] * (x2 - x1), y1, x1, y1 + u[:, 1] * (y2 - y1), x1 + u[:, 2] * (x2 - x1), y2,
                        x2, y1 + u[:, 3] * (y2 - y1), (x1 + x2) / 2, (y1 + y2) / 2], 1)


def _polygons(b, nv=36):
    masks = []
    ang = -np.pi / 2 + 2 * np.pi * np.arange(nv) / nv
    for x1, y1, x2, y2 in b.tolist():
        cx, cy, rx, ry = (x1 + x2) / 2, (y1 + y2) / 2, (x2 - x1) / 2, (y2 - y1) / 2
        masks.append([np.stack([cx + rx * np.cos(ang), cy + ry * np.sin(ang)], 1).astype(np.float32).reshape(-1)])
    return types.SimpleNamespace(masks=masks, areas=None)


def _keypoints(g, b, nk=17):
    u = torch.rand(b.shape[0], nk, 2, generator=g)
    x = b[:, None, 0] + u[..., 0] * (b[:, None, 2] - b[:, None, 0])
    y = b[:, None, 1] + u[..., 1] * (b[:, None, 3] - b[:, None, 1])
    r = torch.rand(b.shape[0], nk, generator=g)
    v = (r > 0.3).float() + (r > 0.5).float()
    v[:, 0] = 2.0
    return torch.stack([x, y, v], 2).reshape(b.shape[0], -1)


def synthetic_batch(task='bbox', batch=2, height=800, width=1344, boxes_per_img=7, num_classes=80, seed=1234,
                    device='cuda', channels_last=True):
    """Returns the kwargs of `detector(**data)` / `detector.train_step(data, None)`."""
    g = torch.Generator().manual_seed(int(seed))
    img = torch.randn(batch, 3, height, width, generator=g)
    metas = [dict(pad_shape=(height, width, 3), img_shape=(height, width, 3), scale_factor=1.0,
                  ori_shape=(height, width, 3), flip=False) for _ in range(batch)]
    gt_b = [_boxes(g, boxes_per_img, height, width) for _ in range(batch)]
    gt_l = [torch.randint(0, num_classes, (boxes_per_img,), generator=g) for _ in range(batch)]
    img = img.to(device)
    if channels_last:
        img = img.contiguous(memory_format=torch.channels_last)
    data = dict(img=img, img_metas=metas, gt_bboxes=[b.to(device) for b in gt_b],
                gt_labels=[l.to(device) for l in gt_l])
    if task in ('bbox', 'pose_bbox', 'bbox_cpv'):
        data['gt_extremes'] = [_extremes(g, b).to(device) for b in gt_b]
    if task == 'bbox_cpv':      # stride-8 box-level class maps + 1/area weights (pipelines.LoadRPDV2Annotations)
        sem = torch.zeros(batch, num_classes, height // 8, width // 8)
        wts = torch.zeros_like(sem)
        for i, (b, l) in enumerate(zip(gt_b, gt_l)):
            area = (b[:, 2] - b[:, 0]) * (b[:, 3] - b[:, 1])
            for j in torch.argsort(area, descending=True).tolist():
                x1, y1, x2, y2 = (int(v / 8) for v in b[j].tolist())
                sem[i, l[j], y1:y2 + 1, x1:x2 + 1] = 1
                wts[i, l[j], y1:y2 + 1, x1:x2 + 1] = 1 / float(area[j])
        data['gt_sem_map'], data['gt_sem_weights'] = sem.to(device), wts.to(device)
    if task == 'segm':
        data['gt_masks'] = [_polygons(b) for b in gt_b]
    if 'pose' in task:
        data['gt_keypoints'] = [_keypoints(g, b).to(device) for b in gt_b]
    return data
