"""Anchor points of the FPN grids (mmdet/core/anchor/point_generator.py:8-37): point (x*s, y*s, s)
for every cell, row-major, no half-stride shift."""
import torch

from ..utils.registry import Registry, build_from_cfg

ANCHOR_GENERATORS = Registry('Anchor generator')


@ANCHOR_GENERATORS.register_module()
class PointGenerator:

    def grid_points(self, featmap_size, stride=16, device='cuda'):
        h, w = featmap_size
        xs = torch.arange(0., w, device=device) * stride
        ys = torch.arange(0., h, device=device) * stride
        xx, yy = xs.repeat(h), ys.view(-1, 1).repeat(1, w).view(-1)
        return torch.stack([xx, yy, xx.new_full((xx.shape[0],), stride)], dim=-1)

    def valid_flags(self, featmap_size, valid_size, device='cuda'):
        (h, w), (vh, vw) = featmap_size, valid_size
        assert vh <= h and vw <= w
        vx = torch.zeros(w, dtype=torch.bool, device=device)
        vy = torch.zeros(h, dtype=torch.bool, device=device)
        vx[:vw] = True
        vy[:vh] = True
        return vx.repeat(h) & vy.view(-1, 1).repeat(1, w).view(-1)


def build_anchor_generator(cfg, default_args=None):
    return build_from_cfg(cfg, ANCHOR_GENERATORS, default_args)
