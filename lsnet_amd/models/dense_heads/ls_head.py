"""LSHead -- the per-pixel location-sensitive head of LSNet
(reference: mmdet/models/dense_heads/lsnet_head.py:16-1849).

Every FPN point predicts class scores plus, per regression branch, N+1 landmark vectors
(N landmarks + the vector to the object centre); each vector component is split into two
non-negative halves [y_up, y_down, x_left, x_right] produced by softplus.  An init stage predicts
the vectors from the tower features; their end points become the sampling offsets of
PyramidDeformConvs that gather features from three neighbouring pyramid levels for the refine
stage and for classification.

Same parameters, state-dict keys, outputs and losses as the reference, organised differently:
  * one table-driven implementation for the four tasks (bbox / segm / pose_bbox / pose_kbox)
    instead of per-task copies;
  * level-batched native ops: a tower layer runs its modulated DCN over the five FPN levels in one
    launch, each PyramidDeformConv evaluates its 15 (target level, source level) pairs in one
    launch (ops.dcn_multi);
  * target building without data-dependent shapes (no nonzero / boolean indexing / .item()), so
    the training step never synchronises with the host.
"""
import math
import os

import numpy as np
import torch
import torch.nn as nn

from ...cnn import ConvModule, bias_init_with_prob, kaiming_init, normal_init
from ...ops.conv import Conv2d
from ...ops.group_norm import GroupNorm
from ...core import PointGenerator, build_assigner, build_sampler, multiclass_nms_lsvr
from ...ops import ModulatedDeformConvPack, PyramidDeformConv
from ...ops.dcn import offset_scale_chain
from ...ops import cross_iou as fused_ciou
from ...ops.focal import level_rows_ok, level_sums
from ...ops.streams import side_stream
from ..builder import HEADS, build_loss

# LSNET_FUSED_LEVEL_SUMS=0: every loss term level by level (the host path's form) on the device as well (A/B switch)
FUSED_LEVEL_SUMS = os.environ.get('LSNET_FUSED_LEVEL_SUMS', '1') != '0'
# LSNET_SIDE_STREAM_TARGETS=0: the init-stage targets on the main stream, between the head's forward and the loss (A/B switch)
SIDE_STREAM_TARGETS = os.environ.get('LSNET_SIDE_STREAM_TARGETS', '1') != '0'

# regression branches of each task; the LAST branch's offsets also drive the classification
# PyramidDeformConv (lsnet_head.py:638, 653, 680, 695)
TASK_BRANCHES = {'bbox': ('bbox',), 'segm': ('segm',), 'pose_bbox': ('bbox', 'pose'), 'pose_kbox': ('pose',)}


class DCNConvModule(nn.Module):
    """ModulatedDeformConvPack -> GroupNorm -> ReLU (lsnet_head.py:1830-1849); keys `conv`, `bn`."""

    def __init__(self, in_channels=256, out_channels=256, kernel_size=3, dilation=1, num_groups=1, dcn_pad=1):
        super().__init__()
        self.conv = ModulatedDeformConvPack(in_channels, out_channels, kernel_size, 1, dcn_pad)
        self.bn = GroupNorm(num_groups, out_channels)
        self.relu = nn.ReLU(inplace=True)

    def forward(self, x):
        return self.bn.forward_act(self.conv(x))

    def forward_multi(self, xs):
        return self.bn.forward_multi(self.conv.forward_multi(xs), relu=True)   # all levels: one fused GN+ReLU


def _split_px_views(x, shapes):
    B, C = x.shape[:2]
    flat = x.permute(0, 2, 3, 1).reshape(B, -1, C)
    outs, o = [], 0
    for h, w in shapes:
        outs.append(flat[:, o:o + h * w].reshape(B, h, w, C).permute(0, 3, 1, 2))
        o += h * w
    return outs


class _SplitPxFn(torch.autograd.Function):
    """`_split_px` with ONE launch in backward: autograd's own backward of the five slices is, per level, a zero fill of
    the whole (B, N_all, C) tensor + a copy of the slice, and then four additions of the five full-size tensors -- 14
    launches and ~9x the bytes of the concatenation below (4 calls per step: 0.25 ms of the LSNet R-50 step)."""

    @staticmethod
    def forward(ctx, x, shapes):
        ctx.shapes, ctx.dims = shapes, (x.shape[0], x.shape[1])
        return tuple(_split_px_views(x, shapes))

    @staticmethod
    def backward(ctx, *grads):
        B, C = ctx.dims
        ref = next(g for g in grads if g is not None)
        parts = [g.permute(0, 2, 3, 1).reshape(B, h * w, C) if g is not None else ref.new_zeros((B, h * w, C))
                 for g, (h, w) in zip(grads, ctx.shapes)]
        flat = torch.cat(parts, dim=1)                                  # (B, N_all, C)
        return flat.unsqueeze(2).permute(0, 3, 1, 2), None              # (B, C, N_all, 1), channels-last memory


class LevelTerms(list):
    """The per-level values of one loss term as the reference returns them (a list of 0-dim tensors) that also remembers the
    (L,) tensor they are views of: `_parse_losses` (detectors/base.py) then adds the levels up with one launch."""

    def __init__(self, base):
        super().__init__(base.unbind(0))
        self.base = base


def _signed_pairs(t, dim):
    """Collapse (neg, pos) pairs along `dim` (size 2): pos if pos > neg else -neg.  Index 0 wins ties
    and is negated, which is what torch.max(dim) + `inds == 0` gives (lsnet_head.py:323-325)."""
    neg, pos = t.select(dim, 0), t.select(dim, 1)
    return torch.where(pos > neg, pos, -neg)


@HEADS.register_module()
class LSHead(nn.Module):

    def __init__(self, num_classes, in_channels, feat_channels=256, stacked_convs=4, strides=(4, 8, 16, 32, 64),
                 point_feat_channels=256, num_kernel_points=9, gradient_mul=0.1,
                 point_strides=[8, 16, 32, 64, 128], point_base_scale=4, task='bbox', num_vectors=4,
                 conv_module_type='norm', background_label=None, conv_cfg=None, norm_cfg=None,
                 loss_cls=dict(type='FocalLoss', use_sigmoid=True, gamma=2.0, alpha=0.25, loss_weight=1.0),
                 loss_bbox_init=dict(type='CrossIOULoss', loss_weight=1.0),
                 loss_bbox_refine=dict(type='CrossIOULoss', loss_weight=2.0),
                 loss_segm_init=None, loss_segm_refine=None, loss_pose_init=None, loss_pose_refine=None,
                 train_cfg=None, test_cfg=None, **kwargs):
        super().__init__()
        assert task in TASK_BRANCHES, task
        self.task, self.branches = task, TASK_BRANCHES[task]
        self.num_classes = self.cls_out_channels = num_classes
        self.in_channels, self.feat_channels, self.stacked_convs = in_channels, feat_channels, stacked_convs
        self.strides, self.conv_cfg, self.norm_cfg = strides, conv_cfg, norm_cfg
        self.train_cfg, self.test_cfg = train_cfg, test_cfg
        self.background_label = num_classes if background_label is None else background_label
        assert self.background_label in (0, num_classes)
        self.num_vectors, self.num_kernel_points = num_vectors, num_kernel_points
        self.point_feat_channels, self.conv_module_type = point_feat_channels, conv_module_type
        self.gradient_mul, self.point_base_scale = gradient_mul, point_base_scale
        self.point_strides = list(point_strides)
        self.fpn_levels = list(range(len(self.point_strides)))
        self.point_generators = [PointGenerator() for _ in self.point_strides]

        self.dcn_kernel = int(np.sqrt(num_kernel_points))
        self.dcn_pad = (self.dcn_kernel - 1) // 2
        assert self.dcn_kernel * self.dcn_kernel == num_kernel_points, 'The points number should be a square number.'
        assert self.dcn_kernel % 2 == 1, 'The points number should be an odd square number.'
        base = np.arange(-self.dcn_pad, self.dcn_pad + 1).astype(np.float64)
        base_yx = np.stack([np.repeat(base, self.dcn_kernel), np.tile(base, self.dcn_kernel)], axis=1).reshape(-1)
        # regular 3x3 grid, (y, x) per tap.  A non-persistent buffer: it follows the module to the device (an
        # H2D copy inside forward would break hipGraph capture) and stays out of the state dict like the
        # reference's plain attribute (lsnet_head.py:84-91)
        self.register_buffer('dcn_base_offset', torch.tensor(base_yx).view(1, -1, 1, 1), persistent=False)
        self._consts = {}

        self.loss_cls = build_loss(loss_cls)
        loss_cfgs = dict(bbox=(loss_bbox_init, loss_bbox_refine), segm=(loss_segm_init, loss_segm_refine),
                         pose=(loss_pose_init, loss_pose_refine))
        for b in self.branches:
            setattr(self, f'loss_{b}_init', build_loss(loss_cfgs[b][0]))
            setattr(self, f'loss_{b}_refine', build_loss(loss_cfgs[b][1]))
        if self.train_cfg:
            self.init_assigner = build_assigner(self.train_cfg.init.assigner)
            self.refine_assigner = build_assigner(self.train_cfg.refine.assigner)
            self.sampler = build_sampler(dict(type='PseudoSampler'), context=self)
        self._init_layers()

    # ------------------------------------------------------------------------------------ layers
    def _out_dims(self, branch):
        """(init_out channels, refine_out channels).  The bbox branch predicts 5 landmark vectors
        (top, left, bottom, right, centre) plus raw offsets for the remaining kernel points; with
        task='pose_bbox' the reference hard-codes 28 / 20 (lsnet_head.py:170-176, 207-211)."""
        if branch == 'bbox':
            if self.task == 'bbox':
                nv = self.num_vectors
                return 4 * (nv + 1) + (self.num_kernel_points - nv - 1) * 2, 4 * (nv + 1)
            return 28, 20
        d = (self.num_vectors + 1) * 4
        return d, d

    def _tower(self):
        ng = self.norm_cfg.num_groups if hasattr(self.norm_cfg, 'num_groups') else self.norm_cfg['num_groups']
        layers = nn.ModuleList()
        for i in range(self.stacked_convs):
            chn = self.in_channels if i == 0 else self.feat_channels
            if self.conv_module_type == 'norm':
                layers.append(ConvModule(chn, self.feat_channels, 3, stride=1, padding=1, conv_cfg=self.conv_cfg,
                                         norm_cfg=self.norm_cfg))
            else:
                layers.append(DCNConvModule(chn, self.feat_channels, self.dcn_kernel, 1, ng, self.dcn_pad))
        return layers

    def _init_layers(self):
        fc, pc = self.feat_channels, self.point_feat_channels
        ng = self.norm_cfg.num_groups if hasattr(self.norm_cfg, 'num_groups') else self.norm_cfg['num_groups']
        self.relu = nn.ReLU(inplace=True)
        self.softplus = nn.Softplus()
        self.cls_GN = GroupNorm(ng, fc)
        self.cls_convs = self._tower()
        for b in self.branches:
            setattr(self, f'{b}_GN', GroupNorm(ng, fc))
            setattr(self, f'{b}_convs', self._tower())
        self.pts_cls_conv = PyramidDeformConv(fc, pc, self.dcn_kernel, 1, self.dcn_pad)
        self.pts_cls_out = Conv2d(pc, self.cls_out_channels, 1, 1, 0)
        self.cls_af_dcn_conv = nn.Sequential(Conv2d(3 * pc, pc, 1, 1, 0), nn.ReLU())
        self.cls_feat_conv = Conv2d(fc, pc, 3, 1, 1)
        for b in self.branches:
            d_init, d_refine = self._out_dims(b)
            setattr(self, f'pts_{b}_init_conv', Conv2d(fc, pc, 3, 1, 1))
            setattr(self, f'pts_{b}_init_out', Conv2d(pc, d_init, 1, 1, 0))
            setattr(self, f'pts_{b}_refine_conv', PyramidDeformConv(fc, pc, self.dcn_kernel, 1, self.dcn_pad))
            setattr(self, f'pts_{b}_refine_out', Conv2d(pc, d_refine, 1, 1, 0))
            setattr(self, f'{b}_af_dcn_conv', nn.Sequential(Conv2d(3 * pc, pc, 1, 1, 0), nn.ReLU()))
            setattr(self, f'{b}_feat_conv', Conv2d(fc, pc, 3, 1, 1))

    def init_weights(self):
        """lsnet_head.py:259-319 (same RNG consumption order: towers, then cls, then each branch)."""
        for m in self.cls_convs:
            normal_init(m.conv, std=0.01)
        for b in self.branches:
            for m in getattr(self, f'{b}_convs'):
                normal_init(m.conv, std=0.01)
        kaiming_init(self.pts_cls_conv)
        normal_init(self.pts_cls_out, std=0.01, bias=bias_init_with_prob(0.01))
        normal_init(self.cls_feat_conv, std=0.01)
        normal_init(self.cls_af_dcn_conv[0], std=0.01)
        for b in self.branches:
            normal_init(getattr(self, f'pts_{b}_init_conv'), std=0.01)
            normal_init(getattr(self, f'pts_{b}_init_out'), std=0.01)
            kaiming_init(getattr(self, f'pts_{b}_refine_conv'))
            normal_init(getattr(self, f'pts_{b}_refine_out'), std=0.01)
            normal_init(getattr(self, f'{b}_feat_conv'), std=0.01)
            normal_init(getattr(self, f'{b}_af_dcn_conv')[0], std=0.01)

    # ------------------------------------------------------------------------- vector decoding
    def extreme_points2bbox(self, pts, y_first=True, extreme=False):
        """(B, 20, H, W) extreme-point vectors -> (B, 4, H, W) box [x_left, y_top, x_right, y_bottom]
        from landmarks (top, left, bottom, right); optionally the 4 extreme points as (x, y) pairs
        (lsnet_head.py:321-347)."""
        v = _signed_pairs(pts.reshape(pts.shape[0], -1, 2, *pts.shape[2:]), 2)     # (B, 10, H, W)
        v = v.reshape(v.shape[0], -1, 2, *v.shape[2:])                             # (B, 5, [y,x], H, W)
        py, px = (v[:, :, 0], v[:, :, 1]) if y_first else (v[:, :, 1], v[:, :, 0])
        bbox = torch.stack([px[:, 1], py[:, 0], px[:, 3], py[:, 2]], dim=1)
        if not extreme:
            return bbox
        ext = torch.stack([px[:, 0], py[:, 0], px[:, 1], py[:, 1], px[:, 2], py[:, 2], px[:, 3], py[:, 3]], dim=1)
        return ext, bbox

    def vectors2bbox(self, pts, y_first=True, vector=False):
        """(B, 4*(nv+1), H, W) -> bounding box of the nv vectors (centre dropped); optionally the
        vectors as interleaved (x, y) (lsnet_head.py:349-370)."""
        body = pts[:, :-4]
        v = _signed_pairs(body.reshape(body.shape[0], -1, 2, *body.shape[2:]), 2)
        v = v.reshape(v.shape[0], -1, 2, *v.shape[2:])
        py, px = (v[:, :, 0], v[:, :, 1]) if y_first else (v[:, :, 1], v[:, :, 0])
        bbox = torch.stack([px.min(1)[0], py.min(1)[0], px.max(1)[0], py.max(1)[0]], 1)
        if not vector:
            return bbox
        return torch.stack([px, py], 2).reshape(py.shape[0], -1, *py.shape[2:]), bbox

    def get_pred_reg(self, raw_reg1, raw_reg2):
        """The 9 sampling offsets (y, x per kernel point) derived from the predicted vectors
        (lsnet_head.py:372-400)."""
        if raw_reg2 is not None:   # bbox: 5 landmark vectors -> 10 signed values, + 8 free offsets
            signed = _signed_pairs(raw_reg1.reshape(raw_reg1.shape[0], -1, 2, *raw_reg1.shape[2:]), 2)
            return torch.cat((signed, raw_reg2), dim=1)
        r = raw_reg1.reshape(raw_reg1.shape[0], -1, 4, *raw_reg1.shape[2:])
        if self.task == 'segm':   # every ceil(nv/8)-th contour vector
            picks = r[:, :-1][:, ::math.ceil(self.num_vectors / (self.num_kernel_points - 1))]
        else:                     # pose: every second keypoint, starting at 1
            picks = r[:, :-1][:, 1::2]
        sel = torch.cat([picks, r[:, -1:]], dim=1)                                  # (B, 9, 4, H, W)
        return _signed_pairs(sel.reshape(sel.shape[0], -1, 2, *sel.shape[3:]), 2)    # (B, 18, H, W)

    # ------------------------------------------------------------------------------------ forward
    def _run_tower(self, convs, feats):
        if self.conv_module_type == 'norm':
            out = list(feats)
            for conv in convs:
                out = [conv(x) for x in out]
            return out
        out = list(feats)
        for conv in convs:
            out = conv.forward_multi(out)
        return out

    def _scale_const(self, sh, sw, like):
        """(1, 2*taps, 1, 1) tensor [sh, sw, sh, sw, ...], built once per (scale, device): constants are never
        uploaded inside the step (no H2D copy in the hot loop, hipGraph-capturable)."""
        key = (float(sh), float(sw), like.device, like.dtype)
        t = self._consts.get(key)
        if t is None:
            t = like.new_tensor([sh, sw]).repeat(self.num_kernel_points).view(1, -1, 1, 1)
            self._consts[key] = t
        return t

    @staticmethod
    def _level_list(lvl, num_levels):
        if lvl == 0:
            return [0, 1, 2]
        if lvl == num_levels - 1:
            return [lvl, lvl - 1, lvl - 2]
        return [lvl, lvl - 1, lvl + 1]

    def _gn_cat(self, gn, maps):
        """relu(GN(maps)) as _cat_px lays it out; the library's GroupNorm writes the concatenated tensor itself"""
        if hasattr(gn, 'forward_cat_px'):
            return gn.forward_cat_px(maps, relu=True)
        return self._cat_px(gn.forward_multi(maps, relu=True))

    @staticmethod
    def _cat_px(maps):
        """Per-level maps (B, C, H_l, W_l) -> ONE (B, C, N_all, 1) tensor in channels-last memory (pixel rows of all
        levels back to back).  1x1 convolutions and pointwise functions do not care where a pixel came from, so the
        levels share one launch (and one weight-gradient accumulation) instead of five."""
        B, C = maps[0].shape[:2]
        x = torch.cat([m.permute(0, 2, 3, 1).reshape(B, -1, C) for m in maps], dim=1)      # (B, N_all, C)
        return x.unsqueeze(2).permute(0, 3, 1, 2)

    @staticmethod
    def _split_px(x, shapes):
        """Inverse of `_cat_px`: per-level (B, C, H_l, W_l) VIEWS of a (B, C, N_all, 1) tensor."""
        if x.requires_grad and torch.is_grad_enabled():
            outs = list(_SplitPxFn.apply(x, tuple(shapes)))
        else:
            outs = _split_px_views(x, shapes)
        for i, o in enumerate(outs):
            o._px_cat = (x, i)     # `_px_base`: the loss reads the concatenated tensor itself instead of gluing the views together
        return outs

    @staticmethod
    def _px_base(maps):
        """The (B, C, N_all, 1) tensor the per-level maps are `_split_px` views of, as (B, N_all, C) rows -- or None."""
        tags = [getattr(m, '_px_cat', None) for m in maps]
        if any(t is None for t in tags) or any(t[0] is not tags[0][0] or t[1] != i for i, t in enumerate(tags)):
            return None
        x = tags[0][0]
        if sum(m.shape[2] * m.shape[3] for m in maps) != x.shape[2] or x.shape[3] != 1:
            return None
        return x.permute(0, 2, 3, 1).reshape(x.shape[0], x.shape[2], x.shape[1])

    def forward(self, feats, after_init=None):
        """feats: tuple of 5 FPN maps.  Returns the reference's 7-tuple of per-level lists
        (cls, bbox_init, bbox_refine, segm_init, segm_refine, pose_init, pose_refine); branches the
        task does not have are lists of None (lsnet_head.py:479-500).
        after_init: called with {branch: per-level init predictions} as soon as those are issued, before the pyramid
        convolutions (forward_train starts the refine-stage assignment there, on its second stream)."""
        nl = len(feats)
        shapes = [tuple(f.shape[2:]) for f in feats]
        base_offset = self.dcn_base_offset.type_as(feats[0])
        cls_feats = self._run_tower(self.cls_convs, feats)
        st = {}
        for b in self.branches:
            tower = self._run_tower(getattr(self, f'{b}_convs'), feats)
            init_conv, init_out = getattr(self, f'pts_{b}_init_conv'), getattr(self, f'pts_{b}_init_out')
            n_sp = self._out_dims(b)[1]
            # 3x3 conv per level; everything after it is pixel-wise: one pass over the concatenated levels
            raw = init_out(self._cat_px(init_conv.forward_multi(tower, relu=True)))   # (ReLU in the 3x3 launch's epilogue)
            sp = self.softplus(raw[:, :n_sp])
            reg = self.get_pred_reg(sp, raw[:, n_sp:] if raw.shape[1] > n_sp else None)
            reg = (1 - self.gradient_mul) * reg.detach() + self.gradient_mul * reg
            st[b] = dict(feat=tower, sp_all=sp, sp=self._split_px(sp, shapes),
                         off=self._split_px(reg - base_offset, shapes))

        if after_init is not None:
            after_init({b: st[b]['sp'] for b in self.branches})

        # --- offsets handed to the pyramid convs.  The reference rescales the offset tensor IN PLACE
        # while looping over the three source levels, so the multipliers accumulate:
        # level_list [l, l-1, l+1] -> y offsets x [s0, s0*s1, s0*s1*s2] (lsnet_head.py:622-638).
        # Reproduced with out-of-place multiplies in the same order (same values, same gradients).
        pairs = []   # (dst level, src level, scale_h, scale_w)
        mults = []
        for l in range(nl):
            bh, bw = cls_feats[l].shape[2:]
            trio = []
            for s in self._level_list(l, nl):
                sh, sw = cls_feats[s].shape[2] / bh, cls_feats[s].shape[3] / bw
                trio.append((sh, sw))
                pairs.append((l, s, sh, sw))
            mults.append(trio)
        # (one launch per branch and direction on the device; three multiplications per level elsewhere.  The last branch's
        # fields also steer the classification gather: it gets its own handles, so the two gradients meet in the launch)
        driver = self.branches[-1]
        scaled, scaled_cls = {}, None
        for b in self.branches:
            sets = offset_scale_chain(st[b]['off'], mults, copies=2 if b == driver else 1)
            scaled[b] = [t for trio in sets[0] for t in trio]
            if b == driver:
                scaled_cls = [t for trio in sets[1] for t in trio]
        scales = [(p[2], p[3]) for p in pairs]

        def gather(conv, src_feats, offsets):
            # the three maps of a destination level side by side in one tensor (the reference concatenates them next)
            return conv.forward_multi([src_feats[p[1]] for p in pairs], offsets, scales, concat=3)

        cls_raw = gather(self.pts_cls_conv, cls_feats, scaled_cls)
        outs = {}
        def fuse(af, fc, raw, feat):
            # relu(1x1 over the three gathered maps of a level) + 3x3 over the level's tower output: each of the two
            # convolutions runs over all levels in one launch
            # (round 5: the sum rides in the epilogue of the second launch -- (conv + bias) + a, the additions and roundings of
            # `f + a` -- instead of five element-wise launches per branch)
            a = af[0].forward_multi(raw, relu=True)
            if hasattr(fc, 'forward_multi') and 'residuals' in fc.forward_multi.__code__.co_varnames:
                return fc.forward_multi(feat, residuals=a)
            f = fc.forward_multi(feat)
            return [x + y for x, y in zip(a, f)]

        fused = fuse(self.cls_af_dcn_conv, self.cls_feat_conv, cls_raw, cls_feats)
        outs['cls'] = self._split_px(self.pts_cls_out(self._gn_cat(self.cls_GN, fused)), shapes)
        for b in self.branches:
            raw = gather(getattr(self, f'pts_{b}_refine_conv'), st[b]['feat'], scaled[b])
            af, fc = getattr(self, f'{b}_af_dcn_conv'), getattr(self, f'{b}_feat_conv')
            gn, ro = getattr(self, f'{b}_GN'), getattr(self, f'pts_{b}_refine_out')
            fused = fuse(af, fc, raw, st[b]['feat'])
            refine = self.softplus(ro(self._gn_cat(gn, fused)) + st[b]['sp_all'].detach())
            outs[b] = self._split_px(refine, shapes)

        none = [None] * nl
        res = [outs['cls']]
        for b in ('bbox', 'segm', 'pose'):
            res += [st[b]['sp'], outs[b]] if b in self.branches else [none, none]
        return tuple(res)

    def forward_train(self, x, img_metas, gt_bboxes, gt_extremes=None, gt_keypoints=None, gt_masks=None,
                      gt_labels=None, gt_bboxes_ignore=None, proposal_cfg=None, **kwargs):
        # The init-stage targets (ground-truth preparation, grid points, CentroidAssigner, dense targets: ~100 launches of a few
        # microseconds) depend on the ground truth and the map sizes only: they are issued on a second stream BEFORE the head's
        # forward and run in the shadow of its deformable convolutions instead of between forward and backward.
        # The refine-stage assignment needs the init predictions and nothing later: it starts on that stream as soon as the head
        # has issued them (`after_init`) and runs beside the pyramid convolutions.
        pre, stage2, hook = None, [], None
        if SIDE_STREAM_TARGETS and x[0].is_cuda:
            main = torch.cuda.current_stream(x[0].device)
            side = self._side_stream(x[0].device)
            side.wait_stream(main)                      # the ground truth may have been produced on the main stream
            with torch.cuda.stream(side):
                pre = self.init_stage_targets([tuple(f.shape[-2:]) for f in x], x[0], gt_bboxes, gt_extremes, gt_keypoints,
                                              gt_masks, gt_labels, img_metas)

            def hook(init_preds):
                side.wait_stream(main)                  # the init predictions are main-stream work
                with torch.cuda.stream(side):
                    stage2.append(self.refine_stage_targets(pre, [p.detach() for p in init_preds[self._box_branch()]]))
        outs = self(x, after_init=hook)
        if pre is not None:
            # (no record_stream: the next use of the side stream starts with wait_stream(main) again, i.e. behind every reader of
            # these tensors, and that is the only place where the allocator can hand their memory out again)
            main.wait_stream(side)
        losses = self.loss(*outs, gt_bboxes, gt_extremes, gt_keypoints, gt_masks, gt_labels, img_metas,
                           gt_bboxes_ignore=gt_bboxes_ignore, init_stage=pre, refine_stage=stage2[0] if stage2 else None)
        if proposal_cfg is None:
            return losses
        return losses, self.get_bboxes(*outs, img_metas, cfg=proposal_cfg)

    # ------------------------------------------------------------------------------------ targets
    def get_points(self, featmap_sizes, img_metas, device):
        """Per-level grid points (shared by all images) and, per image, the flags of the cells that
        lie inside the padded image (lsnet_head.py:757-794).  Both depend only on the grid geometry and the padded
        shapes: built once per geometry and reused (they are read-only downstream), so a training step with a
        constant input shape creates none of these ~50 small tensors again."""
        key = ('points', tuple(tuple(s) for s in featmap_sizes), tuple(tuple(m['pad_shape'][:2]) for m in img_metas),
               str(device))
        hit = self._consts.get(key)
        if hit is not None:
            return hit
        points = [self.point_generators[i].grid_points(featmap_sizes[i], self.point_strides[i], device)
                  for i in range(len(featmap_sizes))]
        flags, all_valid = [], True
        for meta in img_metas:
            per_level = []
            h, w = meta['pad_shape'][:2]
            for i, (fh, fw) in enumerate(featmap_sizes):
                s = self.point_strides[i]
                vh, vw = min(int(np.ceil(h / s)), fh), min(int(np.ceil(w / s)), fw)
                all_valid = all_valid and vh == fh and vw == fw
                per_level.append(self.point_generators[i].valid_flags((fh, fw), (vh, vw), device))
            flags.append(per_level)
        if len(self._consts) > 64:            # multi-scale training visits many geometries: keep the table small
            self._consts.clear()
        self._consts[key] = (points, flags, all_valid)
        return points, flags, all_valid

    def _dense_targets(self, gt_inds, gt_labels, gt_bboxes, extra):
        """Per-point targets from an assignment WITHOUT index lists: positives are rows with
        gt_inds > 0 (lsnet_head.py:834-890 written densely).  `extra`: dict name -> (G, d) tensors
        gathered the same way (extreme points, polygons, keypoints, visibilities)."""
        pos = gt_inds > 0
        idx = (gt_inds - 1).clamp(min=0)
        posf = pos.unsqueeze(1)
        out = dict(bboxes_gt=torch.where(posf, gt_bboxes[idx], 0.0),          # python scalars: no tensor is made for them
                   bbox_weights=posf.to(gt_bboxes.dtype).expand(-1, 4),
                   labels=torch.where(pos, gt_labels[idx] if gt_labels is not None else torch.ones_like(idx),
                                      self.background_label),
                   label_weights=torch.ones_like(pos, dtype=gt_bboxes.dtype), num_pos=pos.sum())
        for k, v in extra.items():
            out[k] = torch.where(posf, v[idx], 0.0)
        return out

    def _assign_image(self, stage, proposals, flags, all_valid, num_level, gt_bboxes, gt_labels, extra):
        """One image, one stage -> dense target dict over ALL points (lsnet_head.py:796-917)."""
        n_all = proposals.shape[0]
        if gt_bboxes.shape[0] == 0:   # nothing to assign: all background
            zeros = proposals.new_zeros
            out = dict(bboxes_gt=zeros((n_all, 4)), bbox_weights=zeros((n_all, 4)),
                       labels=proposals.new_full((n_all,), self.background_label, dtype=torch.long),
                       label_weights=proposals.new_ones((n_all,)), num_pos=proposals.new_zeros((), dtype=torch.long))
            for k, v in extra.items():
                out[k] = zeros((n_all, v.shape[1]))
            return out
        inside = None if all_valid else flags
        props = proposals if inside is None else proposals[inside]
        if stage == 'init':
            res = self.init_assigner.assign(props, gt_bboxes, extra.get('extremes_gt'), None, gt_labels)
        else:
            if inside is not None:
                num_level = [int(f.sum()) for f in torch.split(inside, num_level)]
            res = self.refine_assigner.assign(props, num_level, gt_bboxes, None, gt_labels)
        out = self._dense_targets(res.gt_inds, gt_labels, gt_bboxes, extra)
        if inside is not None:   # scatter back to the full grid; outside cells: zeros, weight 0
            full = {}
            for k, v in out.items():
                if k == 'num_pos':
                    full[k] = v
                    continue
                buf = v.new_zeros((n_all,) + tuple(v.shape[1:]))
                buf[inside] = v
                full[k] = buf
            out = full
        return out

    def get_targets(self, proposals_list, flags_list, all_valid, num_level, gt_bboxes_list, gt_labels_list,
                    extra_list, stage):
        """All images of the batch -> per-level target tensors (B, N_l, ...) plus the positive count
        sum_img max(n_pos, 1) as a DEVICE scalar (lsnet_head.py:919-1019)."""
        per_img = [self._assign_image(stage, proposals_list[i], flags_list[i], all_valid, num_level,
                                      gt_bboxes_list[i], None if gt_labels_list is None else gt_labels_list[i],
                                      extra_list[i]) for i in range(len(proposals_list))]
        num_total_pos = sum(t['num_pos'].clamp(min=1) for t in per_img)
        out = {}
        for k in per_img[0]:
            if k == 'num_pos':
                continue
            out[k] = torch.stack([t[k] for t in per_img], 0)     # (B, N_all, ...): levels stay concatenated
        return out, num_total_pos

    # --------------------------------------------------------------------------------------- loss
    def get_bbox_gt_reg(self, gt_pts, anchor_pts, bbox_weights):
        return self._gt_reg(gt_pts, anchor_pts, bbox_weights)

    def get_poly_gt_reg(self, gt_pts, anchor_pts, bbox_weights):
        return self._gt_reg(gt_pts, anchor_pts, bbox_weights)

    @staticmethod
    def _gt_reg(gt_pts, anchor_pts, weights):
        """Ground-truth landmark points (N, 2m) as (x, y) pairs -> regression targets (N, 4m) in the
        head's layout [y_up, y_down, x_left, x_right] per landmark, and the mask of the active half of
        every pair.  Rows with zero weight get all-zero targets (lsnet_head.py:402-454)."""
        m = gt_pts.shape[1] // 2
        d = (gt_pts - anchor_pts[:, :2].repeat(1, m)).reshape(-1, m, 2)   # (N, m, [dx, dy])
        nonneg = d >= 0
        mag = torch.abs(d) * (weights[:, :1] > 0).to(d.dtype).unsqueeze(-1)
        zero = torch.zeros_like(mag)
        # per landmark: [y_up (dy<0), y_down (dy>=0), x_left (dx<0), x_right (dx>=0)]
        reg = torch.stack([torch.where(nonneg[..., 1], zero[..., 1], mag[..., 1]),
                           torch.where(nonneg[..., 1], mag[..., 1], zero[..., 1]),
                           torch.where(nonneg[..., 0], zero[..., 0], mag[..., 0]),
                           torch.where(nonneg[..., 0], mag[..., 0], zero[..., 0])], dim=-1)
        act = torch.stack([~nonneg[..., 1], nonneg[..., 1], ~nonneg[..., 0], nonneg[..., 0]], dim=-1)
        return reg.reshape(-1, 4 * m), act.reshape(-1, 4 * m)

    def _level_rows(self, num_level, like):
        """(N_all,) stride of every point's level, built once per grid geometry (constants are not re-created
        inside the step)."""
        key = ('strides', tuple(num_level), like.device, like.dtype)
        t = self._consts.get(key)
        if t is None:
            t = torch.cat([like.new_full((n,), float(s)) for n, s in zip(num_level, self.point_strides)])
            self._consts[key] = t
        return t

    def loss_levels(self, cls_scores, preds, tg_init, tg_refine, points, num_level, n_init, n_refine):
        """The per-level losses of the reference's `loss_single` (lsnet_head.py:1021-1270) for ALL levels at once.
        The regression targets and the cross-IOU terms are row-wise functions of (prediction, target, anchor,
        stride): they are evaluated once on the concatenation of the levels -- a fifth of the elementwise launches
        -- and only the final sums are taken per level, so the returned lists hold the same per-level values."""
        B = cls_scores[0].shape[0]
        losses = {'cls': []}
        # On the device every term's levels are ONE launch (ops/focal.py: per-level focal sums, per-level row sums) over the head's
        # concatenated tensors, which `_split_px` left on the per-level views; elsewhere level by level as the reference does.
        by_level = FUSED_LEVEL_SUMS and level_rows_ok(cls_scores[0], B, num_level)
        cls_cat = self._px_base(cls_scores) if by_level and hasattr(self.loss_cls, 'forward_levels') \
            and getattr(self.loss_cls, 'reduction', None) == 'mean' else None
        if cls_cat is not None and cls_cat.is_contiguous() and tg_refine['labels'].is_contiguous() \
                and tg_refine['label_weights'].is_contiguous():
            lv_cls = self.loss_cls.forward_levels(cls_cat.reshape(-1, self.cls_out_channels), tg_refine['labels'].reshape(-1),
                                                  tg_refine['label_weights'].reshape(-1), B, num_level, avg_factor=n_refine)
            losses['cls'] = LevelTerms(lv_cls)
        else:
            labels = torch.split(tg_refine['labels'], num_level, dim=1)
            label_w = torch.split(tg_refine['label_weights'], num_level, dim=1)
            for lvl, cs in enumerate(cls_scores):
                cs = cs.permute(0, 2, 3, 1).reshape(-1, self.cls_out_channels)
                losses['cls'].append(self.loss_cls(cs, labels[lvl].reshape(-1), label_w[lvl].reshape(-1),
                                                   avg_factor=n_refine))

        def per_level_terms(rows, n):
            if by_level and rows.dtype == torch.float32:
                return LevelTerms(level_sums(rows.reshape(-1), B, num_level) / n)
            return [r.sum() / n for r in torch.split(rows.reshape(B, -1), num_level, dim=1)]
        # (constants of the grid geometry and the batch size: built once, like the points themselves)
        key = ('loss_rows', B, tuple(id(p) for p in points))     # (the entry holds `points`: the ids stay theirs while it lives)
        hit = self._consts.get(key)
        if hit is None:
            stride = self._level_rows(num_level, points[0]).repeat(B).unsqueeze(1)            # (B*N_all, 1)
            hit = self._consts[key] = (stride, self.point_base_scale * stride,
                                       torch.cat(points)[None].expand(B, -1, -1).reshape(-1, 3), list(points))
        stride, norm, anchor = hit[:3]
        for b in self.branches:
            for stage, tg, plist, n in (('init', tg_init, preds[b][0], n_init), ('refine', tg_refine, preds[b][1], n_refine)):
                bw = tg['bbox_weights'].reshape(-1, 4)
                bbox_gt = tg['bboxes_gt'].reshape(-1, 4)
                if b == 'bbox':
                    gt_pts = tg['extremes_gt'].reshape(-1, tg['extremes_gt'].shape[-1])
                    kw = dict(bbox_gt=bbox_gt / norm)
                elif b == 'segm':
                    gt_pts = tg['polygons_gt'].reshape(-1, tg['polygons_gt'].shape[-1])
                    kw = dict(bbox_gt=bbox_gt / norm)
                else:
                    gt_pts = tg['keypoints_gt'].reshape(-1, tg['keypoints_gt'].shape[-1])
                    kw = dict(bbox_gt=None, vs=tg['keypoints_vs'].reshape(-1, self.num_vectors))
                width = gt_pts.shape[1] * 2
                weights = bw[:, :1].expand(-1, width)
                pred = self._px_base(plist) if by_level else None
                if pred is None or pred.shape[2] != width:
                    pred = torch.cat([p.permute(0, 2, 3, 1).reshape(B, -1, width) for p in plist], dim=1)
                loss_fn = getattr(self, f'loss_{b}_{stage}')
                if (b == 'bbox' and width == 20 and fused_ciou.enabled() and pred.is_cuda and pred.dtype == torch.float32
                        and getattr(loss_fn, 'loss_type', None) == 'bbox'):
                    # (default; LSNET_FUSED_CIOU=0 switches it off) scaling, normalisation, target construction and the loss
                    # in one launch; the segm / pose stages below reach the fused polygon / keypoint rows through loss_fn
                    rows = loss_fn.loss_weight * fused_ciou.cross_iou_bbox_stage_rows(
                        pred.reshape(-1, width), gt_pts, anchor, bbox_gt, bw[:, 0], self.point_base_scale, loss_fn.alpha,
                        loss_fn.eps)
                    losses[f'{b}_{stage}'] = per_level_terms(rows, n)
                    continue
                pred = pred.reshape(-1, width) * stride
                gt_reg, active = self._gt_reg(gt_pts, anchor, weights)
                rows = loss_fn(pred / norm, gt_reg / norm, weights, reduction_override='none',
                               anchor_pts=anchor[:, :-1] / norm, pos_inds=active, **kw)   # weighted, per row
                losses[f'{b}_{stage}'] = per_level_terms(rows, n)
        return losses

    def _side_stream(self, device):
        return side_stream(device)      # (ONE per process and device: ops/streams.py says why)

    def init_stage_targets(self, featmap_sizes, like, gt_bboxes, gt_extremes, gt_keypoints_vs, gt_masks, gt_labels, img_metas):
        """Everything of `loss` (lsnet_head.py:1272-1437) that does not depend on a prediction: the ground truth in the head's
        formats, grid points and flags, the init-stage assignment and its dense targets.  `like`: a tensor of the head's device
        and dtype."""
        device = like.device
        num_imgs = len(img_metas)
        extra = [dict() for _ in range(num_imgs)]
        if self.task in ('bbox', 'pose_bbox'):
            if gt_extremes is None:
                gt_extremes = self.get_border_center(gt_bboxes)
            for i in range(num_imgs):
                extra[i]['extremes_gt'] = gt_extremes[i]
        if self.task == 'segm':
            gt_polygons, gt_bboxes = self.process_polygons(gt_masks, [like])
            for i in range(num_imgs):
                extra[i]['polygons_gt'] = gt_polygons[i]
        elif self.task == 'pose_bbox':
            kps, vs = self.process_keypoints_with_bbox(gt_bboxes, gt_keypoints_vs)
            for i in range(num_imgs):
                extra[i].update(keypoints_gt=kps[i], keypoints_vs=vs[i])
        elif self.task == 'pose_kbox':
            kps, gt_bboxes, vs = self.process_keypoints_with_kbox(gt_keypoints_vs)
            for i in range(num_imgs):
                extra[i].update(keypoints_gt=kps[i], keypoints_vs=vs[i])

        featmap_sizes = [tuple(s) for s in featmap_sizes]
        assert len(featmap_sizes) == len(self.point_generators)
        points, flags, all_valid = self.get_points(featmap_sizes, img_metas, device)
        num_level = [p.shape[0] for p in points]
        flat_points = torch.cat(points)
        flat_flags = [torch.cat(f) for f in flags]
        tg_init, n_init = self.get_targets([flat_points] * num_imgs, flat_flags, all_valid, num_level, gt_bboxes,
                                           gt_labels, extra, 'init')
        return dict(featmap_sizes=featmap_sizes, extra=extra, gt_bboxes=gt_bboxes, gt_labels=gt_labels, points=points,
                    all_valid=all_valid, num_level=num_level, flat_flags=flat_flags, tg_init=tg_init, n_init=n_init)

    def _box_branch(self):
        return 'bbox' if 'bbox' in self.branches else self.branches[0]

    def refine_stage_targets(self, init_stage, init_preds):
        """The refine-stage assignment: on the boxes decoded from the (detached) init predictions of the box branch
        (lsnet_head.py:1342-1378).  Returns (targets, positive count)."""
        points, box_branch = init_stage['points'], self._box_branch()
        num_imgs = init_preds[0].shape[0]
        decoded = []
        for lvl, p in enumerate(init_preds):
            p = p.detach()
            box = self.extreme_points2bbox(p) if box_branch == 'bbox' else self.vectors2bbox(p)
            box = box * self.point_strides[lvl]
            centre = torch.cat([points[lvl][:, :2], points[lvl][:, :2]], dim=1)
            decoded.append(centre[None] + box.permute(0, 2, 3, 1).reshape(num_imgs, -1, 4))
        boxes = torch.cat(decoded, dim=1)
        return self.get_targets([boxes[i] for i in range(num_imgs)], init_stage['flat_flags'], init_stage['all_valid'],
                                init_stage['num_level'], init_stage['gt_bboxes'], init_stage['gt_labels'], init_stage['extra'],
                                'refine')

    def loss(self, cls_scores, bbox_pts_preds_init, bbox_pts_preds_refine, segm_pts_preds_init,
             segm_pts_preds_refine, pose_pts_preds_init, pose_pts_preds_refine, gt_bboxes, gt_extremes,
             gt_keypoints_vs, gt_masks, gt_labels, img_metas, gt_bboxes_ignore=None, init_stage=None, refine_stage=None):
        """lsnet_head.py:1272-1437.  init_stage / refine_stage: the results of `init_stage_targets` / `refine_stage_targets`
        when the caller has computed them already (forward_train does, on a second stream)."""
        preds = dict(bbox=(bbox_pts_preds_init, bbox_pts_preds_refine),
                     segm=(segm_pts_preds_init, segm_pts_preds_refine),
                     pose=(pose_pts_preds_init, pose_pts_preds_refine))
        num_imgs = len(img_metas)
        featmap_sizes = [tuple(m.shape[-2:]) for m in cls_scores]
        if init_stage is None or init_stage['featmap_sizes'] != featmap_sizes:
            refine_stage = None
            init_stage = self.init_stage_targets(featmap_sizes, cls_scores[0], gt_bboxes, gt_extremes, gt_keypoints_vs, gt_masks,
                                                 gt_labels, img_metas)
        extra, gt_bboxes, points = init_stage['extra'], init_stage['gt_bboxes'], init_stage['points']
        all_valid, num_level, flat_flags = init_stage['all_valid'], init_stage['num_level'], init_stage['flat_flags']
        tg_init, n_init = init_stage['tg_init'], init_stage['n_init']
        if refine_stage is None:
            refine_stage = self.refine_stage_targets(init_stage, preds[self._box_branch()][0])
        tg_refine, n_refine = refine_stage

        lv = self.loss_levels(cls_scores, {b: preds[b] for b in self.branches}, tg_init, tg_refine, points, num_level,
                              n_init, n_refine)
        out = {'loss_cls': lv['cls']}
        for b in self.branches:
            out[f'loss_{b}_init'] = lv[f'{b}_init']
            out[f'loss_{b}_refine'] = lv[f'{b}_refine']
        return out

    # ----------------------------------------------------------------------------------- decoding
    def get_bboxes(self, cls_scores, bbox_pts_preds_init, bbox_pts_preds_refine, segm_pts_preds_init,
                   segm_pts_preds_refine, pose_pts_preds_init, pose_pts_preds_refine, img_metas, cfg=None,
                   rescale=False, nms=True):
        """lsnet_head.py:1439-1511: refine-stage vectors -> boxes + landmark vectors, per image
        top-k, decode and NMS."""
        if self.task in ('bbox', 'pose_bbox'):
            dec = [self.extreme_points2bbox(p, extreme=True) for p in bbox_pts_preds_refine]
            box_maps = [d[1] for d in dec]
            vec_maps = [d[0] for d in dec]
        if self.task == 'segm':
            dec = [self.vectors2bbox(p, vector=True) for p in segm_pts_preds_refine]
            box_maps, vec_maps = [d[1] for d in dec], [d[0] for d in dec]
        elif self.task in ('pose_bbox', 'pose_kbox'):
            dec = [self.vectors2bbox(p, vector=True) for p in pose_pts_preds_refine]
            vec_maps = [d[0] for d in dec]
            if self.task == 'pose_kbox':
                box_maps = [d[1] for d in dec]
        device = cls_scores[0].device
        points = [self.point_generators[i].grid_points(cls_scores[i].shape[-2:], self.point_strides[i], device)
                  for i in range(len(cls_scores))]
        results = []
        for i, meta in enumerate(img_metas):
            results.append(self._get_bboxes_single([c[i].detach() for c in cls_scores],
                                                   [m[i].detach() for m in box_maps],
                                                   [m[i].detach() for m in vec_maps], points, meta['img_shape'],
                                                   meta['scale_factor'], cfg, rescale, nms))
        return results

    def _get_bboxes_single(self, cls_scores, bbox_preds, vec_preds, mlvl_points, img_shape, scale_factor, cfg,
                           rescale=False, nms=True):
        """lsnet_head.py:1513-1668"""
        cfg = self.test_cfg if cfg is None else cfg
        assert len(cls_scores) == len(mlvl_points)
        nv = self.num_vectors
        boxes_all, vecs_all, scores_all = [], [], []
        for lvl, (cls_score, bbox_pred, vec_pred, points) in enumerate(zip(cls_scores, bbox_preds, vec_preds,
                                                                           mlvl_points)):
            assert cls_score.shape[-2:] == bbox_pred.shape[-2:]
            stride = self.point_strides[lvl]
            scores = cls_score.permute(1, 2, 0).reshape(-1, self.cls_out_channels).sigmoid()
            bbox_pred = bbox_pred.permute(1, 2, 0).reshape(-1, 4)
            vec_pred = vec_pred.permute(1, 2, 0).reshape(-1, nv * 2)
            nms_pre = cfg.get('nms_pre', -1)
            if 0 < nms_pre < scores.shape[0]:
                _, keep = scores.max(dim=1)[0].topk(nms_pre)
                points, bbox_pred, vec_pred, scores = points[keep], bbox_pred[keep], vec_pred[keep], scores[keep]
            xy = points[:, :2]
            bboxes = bbox_pred * stride + torch.cat([xy, xy], dim=1)
            vecs = vec_pred * stride + xy.repeat(1, nv)
            x1 = bboxes[:, 0].clamp(min=0, max=img_shape[1])
            y1 = bboxes[:, 1].clamp(min=0, max=img_shape[0])
            x2 = bboxes[:, 2].clamp(min=0, max=img_shape[1])
            y2 = bboxes[:, 3].clamp(min=0, max=img_shape[0])
            boxes_all.append(torch.stack([x1, y1, x2, y2], dim=-1))
            if self.task == 'bbox':
                # [x_top, y1, x1, y_left, x_bottom, y2, x2, y_right] (lsnet_head.py:1579-1583)
                xt = vecs[:, 0].clamp(min=0, max=img_shape[1])
                yl = vecs[:, 3].clamp(min=0, max=img_shape[0])
                xb = vecs[:, 4].clamp(min=0, max=img_shape[1])
                yr = vecs[:, 7].clamp(min=0, max=img_shape[0])
                vecs_all.append(torch.stack([xt, y1, x1, yl, xb, y2, x2, yr], dim=-1))
            else:
                vx = vecs[:, 0::2].clamp(min=0, max=img_shape[1])
                vy = vecs[:, 1::2].clamp(min=0, max=img_shape[0])
                vecs_all.append(torch.stack([vx, vy], 2).reshape(vecs.size(0), -1))
            scores_all.append(scores)
        mlvl_bboxes, mlvl_vecs, mlvl_scores = torch.cat(boxes_all), torch.cat(vecs_all), torch.cat(scores_all)
        if rescale:
            sf = np.atleast_1d(np.asarray(scale_factor, dtype=np.float32))
            sf = np.tile(sf, 4)[:4] if sf.size < 4 else sf
            mlvl_bboxes = mlvl_bboxes / mlvl_bboxes.new_tensor(sf)
            reps = 2 if self.task == 'bbox' else None
            vsf = np.tile(sf, reps) if reps else np.tile(sf[:2], nv)
            mlvl_vecs = mlvl_vecs / mlvl_vecs.new_tensor(vsf)
        mlvl_scores = torch.cat([mlvl_scores, mlvl_scores.new_zeros(mlvl_scores.shape[0], 1)], dim=1)
        if not nms:
            return mlvl_bboxes, mlvl_vecs, mlvl_scores
        return multiclass_nms_lsvr(mlvl_bboxes, mlvl_vecs, mlvl_scores, nv, cfg.score_thr, cfg.nms,
                                   cfg.max_per_img)

    # -------------------------------------------------------------------------- ground-truth prep
    @staticmethod
    def get_border_center(gt_bboxes_list):
        """Fallback extreme points: the mid-points of the box borders + the box centre, as
        [top, left, bottom, right, centre] (x, y) pairs (lsnet_head.py:1677-1697)."""
        out = []
        for g in gt_bboxes_list:
            cx, cy = (g[:, 2] + g[:, 0]) / 2.0, (g[:, 3] + g[:, 1]) / 2.0
            out.append(torch.stack([cx, g[:, 1], g[:, 0], cy, cx, g[:, 3], g[:, 2], cy, cx, cy], dim=1))
        return out

    @staticmethod
    def component_polygon_area(poly):
        x, y = poly[:, 0], poly[:, 1]   # shoelace formula
        return 0.5 * np.abs(np.dot(x, np.roll(y, 1)) - np.dot(y, np.roll(x, 1)))

    def process_polygons(self, gt_masks_list, cls_scores):
        """Largest component of every instance polygon (already resampled to nv points by the data
        pipeline) + the centre of its bounding box; also returns those boxes (lsnet_head.py:1717-1756)."""
        device, dtype = cls_scores[0].device, cls_scores[0].dtype
        polys_out, boxes_out = [], []
        for gt_masks in gt_masks_list:
            inst = []
            for comps in gt_masks.masks:
                areas = [self.component_polygon_area(c.reshape(-1, 2)) for c in comps]
                best = 0
                for j in range(1, len(areas)):   # first maximum wins, as in the reference
                    if areas[j] > areas[best]:
                        best = j
                inst.append(torch.tensor(comps[best].reshape(-1, 2), dtype=dtype, device=device))
            p = torch.stack(inst)
            lo, hi = p.min(1)[0], p.max(1)[0]
            centre = ((lo + hi) / 2).unsqueeze(1)
            polys_out.append(torch.cat([p, centre], dim=1).reshape(p.size(0), -1))
            boxes_out.append(torch.cat([lo, hi], dim=1))
        return polys_out, boxes_out

    @staticmethod
    def process_keypoints_with_bbox(gt_bboxes_list, gt_keypoints_vs_list):
        """(G, 17*3) [x, y, v] -> (G, 2*18) keypoints + bbox centre, and (G, 17) visibilities
        (lsnet_head.py:1758-1784)."""
        kps_out, vs_out = [], []
        for g, kv in zip(gt_bboxes_list, gt_keypoints_vs_list):
            xy = torch.stack((kv[:, 0::3], kv[:, 1::3]), dim=2).reshape(kv.size(0), -1)
            centre = torch.stack([(g[:, 0] + g[:, 2]) / 2, (g[:, 1] + g[:, 3]) / 2], 1)
            kps_out.append(torch.cat((xy, centre), 1))
            vs_out.append(kv[:, 2::3])
        return kps_out, vs_out

    @staticmethod
    def process_keypoints_with_kbox(gt_keypoints_vs_list):
        """As above, but the box is the extent of the VISIBLE keypoints (lsnet_head.py:1786-1828)."""
        kps_out, box_out, vs_out = [], [], []
        for kv in gt_keypoints_vs_list:
            x, y, v = kv[:, 0::3], kv[:, 1::3], kv[:, 2::3]
            hidden = v == 0
            big, small = x.new_full((), 10000000.), x.new_full((), -1.)
            xmin, ymin = torch.where(hidden, big, x).min(1)[0], torch.where(hidden, big, y).min(1)[0]
            xmax, ymax = torch.where(hidden, small, x).max(1)[0], torch.where(hidden, small, y).max(1)[0]
            xy = torch.stack((x, y), dim=2).reshape(kv.size(0), -1)
            centre = torch.stack([(xmin + xmax) / 2, (ymin + ymax) / 2], 1)
            kps_out.append(torch.cat((xy, centre), 1))
            box_out.append(torch.stack([xmin, ymin, xmax, ymax], 1))
            vs_out.append(v)
        return kps_out, box_out, vs_out
