"""LSCPVHead -- LSNet's bbox head with corner-point verification (the "LSNet-CPV" rows of the reference's README;
reference: mmdet/models/dense_heads/lscpvnet_head.py:16-1123, after RepPoints v2).

On top of the bbox task of `LSHead` (towers, init / refine landmark regression, pyramid deformable gathers,
cross-IOU losses) it adds, per FPN level:
  * a shared tower on the regression features feeding
      - a box-level semantic branch (`reppoints_sem_out`, target: stride-8 class maps from the data pipeline,
        SEPFocalLoss) whose embedding is added back into the classification / regression / corner features;
      - a corner branch: top-left and bottom-right corner pooling (`hem_tl`, `hem_br`), each predicting a corner
        heat-map and a sub-cell offset (GaussianFocalLoss / SmoothL1Loss, targets from `PointHMAssigner`);
  * the 2 + 4 corner maps are appended to the features the pyramid deformable convolutions gather from
    (`feat_channels + 6` inputs);
  * at test time the decoded box corners of levels >= 1 snap to the strongest corner response in a 2x2 window of the
    stride-8 (levels 1, 2) or stride-16 (levels 3, 4) heat-map, plus that cell's predicted offset.

Same parameter names / shapes and outputs as the reference.  Everything the bbox task shares with `LSHead` IS
`LSHead` (targets, cross-IOU loss evaluation over concatenated levels, the accumulating offset rescale of the pyramid
gather); the additions follow its rules: level-batched launches, no data-dependent shapes in the training step."""
import torch
import torch.nn as nn
import torch.nn.functional as F

from ...cnn import ConvModule, bias_init_with_prob, normal_init
from ...core import build_assigner, multiclass_nms
from ...ops import PyramidDeformConv
from ...ops.conv import Conv2d
from ...ops.corner_pool import BRPool, TLPool
from ...ops.group_norm import GroupNorm
from ..builder import HEADS, build_loss
from .ls_head import DCNConvModule, LSHead


@HEADS.register_module()
class LSCPVHead(LSHead):

    def __init__(self, num_classes, in_channels, point_feat_channels=256, shared_stacked_convs=1, first_kernel_size=3,
                 kernel_size=1, corner_dim=64, num_points=9, gradient_mul=0.1, point_strides=[8, 16, 32, 64, 128],
                 point_base_scale=4, conv_module_type='norm',
                 loss_cls=dict(type='FocalLoss', use_sigmoid=True, gamma=2.0, alpha=0.25, loss_weight=1.0),
                 loss_bbox_init=dict(type='SmoothL1Loss', beta=1.0 / 9.0, loss_weight=0.5),
                 loss_bbox_refine=dict(type='SmoothL1Loss', beta=1.0 / 9.0, loss_weight=1.0),
                 loss_heatmap=dict(type='GaussianFocalLoss', alpha=2.0, gamma=4.0, loss_weight=0.25),
                 loss_offset=dict(type='SmoothL1Loss', beta=1.0 / 9.0, loss_weight=1.0),
                 loss_sem=dict(type='SEPFocalLoss', gamma=2.0, alpha=0.25, loss_weight=0.1),
                 use_grid_points=False, center_init=True, moment_mul=0.01, **kwargs):
        # plain attributes `_init_layers` (called by the parent constructor) needs
        self.num_points, self.shared_stacked_convs = num_points, shared_stacked_convs
        self.first_kernel_size, self.kernel_size, self.corner_dim = first_kernel_size, kernel_size, corner_dim
        self.use_grid_points, self.center_init = use_grid_points, center_init
        super().__init__(num_classes, in_channels, point_feat_channels=point_feat_channels, num_kernel_points=num_points,
                         gradient_mul=gradient_mul, point_strides=point_strides, point_base_scale=point_base_scale,
                         task='bbox', num_vectors=4, conv_module_type=conv_module_type, loss_cls=loss_cls,
                         loss_bbox_init=loss_bbox_init, loss_bbox_refine=loss_bbox_refine, **kwargs)
        self.loss_heatmap, self.loss_offset = build_loss(loss_heatmap), build_loss(loss_offset)
        self.loss_sem = build_loss(loss_sem)
        if self.train_cfg:
            self.hm_assigner = build_assigner(self.train_cfg.heatmap.assigner)

    # ------------------------------------------------------------------------------------ layers
    def _shared_tower(self):
        ng = self.norm_cfg.num_groups if hasattr(self.norm_cfg, 'num_groups') else self.norm_cfg['num_groups']
        layers = nn.ModuleList()
        for _ in range(self.shared_stacked_convs):
            if self.conv_module_type == 'norm':
                layers.append(ConvModule(self.feat_channels, self.feat_channels, 3, stride=1, padding=1,
                                         conv_cfg=self.conv_cfg, norm_cfg=self.norm_cfg))
            else:
                layers.append(DCNConvModule(self.feat_channels, self.feat_channels, 3, 1, ng, self.dcn_pad))
        return layers

    def _init_layers(self):
        """lscpvnet_head.py:99-183"""
        fc, pc = self.feat_channels, self.point_feat_channels
        ng = self.norm_cfg.num_groups if hasattr(self.norm_cfg, 'num_groups') else self.norm_cfg['num_groups']
        self.relu, self.softplus = nn.ReLU(inplace=True), nn.Softplus()
        self.cls_GN, self.bbox_GN = GroupNorm(ng, fc), GroupNorm(ng, fc)
        self.cls_convs, self.bbox_convs = self._tower(), self._tower()
        self.shared_convs = self._shared_tower()
        pool_kw = dict(first_kernel_size=self.first_kernel_size, kernel_size=self.kernel_size, corner_dim=self.corner_dim)
        self.hem_tl = TLPool(fc, self.conv_cfg, self.norm_cfg, **pool_kw)
        self.hem_br = BRPool(fc, self.conv_cfg, self.norm_cfg, **pool_kw)
        gather_in = fc + 6                                   # features + 2 corner scores + 4 corner offsets
        # The matrix-pipe kernels step through channels in eights: the gathered maps get zero channels up to the next
        # multiple of 8 and the four weights that read them a matching zero-padded VIEW per step (parameters keep the
        # reference's shapes: 262 input channels for feat_channels = 256).
        self.gather_pad = (-gather_in) % 8
        self.pts_cls_conv = PyramidDeformConv(gather_in, pc, self.dcn_kernel, 1, self.dcn_pad)
        self.pts_cls_out = Conv2d(pc, self.cls_out_channels, 1, 1, 0)
        self.pts_bbox_init_conv = Conv2d(fc, pc, 3, 1, 1)
        self.pts_bbox_init_out = Conv2d(pc, 4 * 5 + (self.num_points - 5) * 2, 1, 1, 0)
        self.pts_bbox_refine_conv = PyramidDeformConv(gather_in, pc, self.dcn_kernel, 1, self.dcn_pad)
        self.pts_bbox_refine_out = Conv2d(pc, 20, 1, 1, 0)
        self.reppoints_hem_tl_score_out = Conv2d(fc, 1, 3, 1, 1)
        self.reppoints_hem_br_score_out = Conv2d(fc, 1, 3, 1, 1)
        self.reppoints_hem_tl_offset_out = Conv2d(fc, 2, 3, 1, 1)
        self.reppoints_hem_br_offset_out = Conv2d(fc, 2, 3, 1, 1)
        self.reppoints_sem_out = Conv2d(fc, self.cls_out_channels, 1, 1, 0)
        self.reppoints_sem_embedding = ConvModule(fc, fc, 1, conv_cfg=self.conv_cfg, norm_cfg=self.norm_cfg)
        self.cls_af_dcn_conv = nn.Sequential(Conv2d(3 * pc, pc, 1, 1, 0), nn.ReLU())
        self.bbox_af_dcn_conv = nn.Sequential(Conv2d(3 * pc, pc, 1, 1, 0), nn.ReLU())
        self.cls_feat_conv = Conv2d(gather_in, pc, 3, 1, 1)
        self.bbox_feat_conv = Conv2d(gather_in, pc, 3, 1, 1)

    def init_weights(self):
        """lscpvnet_head.py:185-213 (same order of random draws; the corner-pool blocks keep their constructor
        initialisation, as there)."""
        for tower in (self.cls_convs, self.bbox_convs, self.shared_convs):
            for m in tower:
                normal_init(m.conv, std=0.01)
        bias_cls = bias_init_with_prob(0.01)
        normal_init(self.pts_cls_conv, std=0.01)
        normal_init(self.pts_cls_out, std=0.01, bias=bias_cls)
        normal_init(self.pts_bbox_init_conv, std=0.01)
        normal_init(self.pts_bbox_init_out, std=0.01)
        normal_init(self.pts_bbox_refine_conv, std=0.01)
        normal_init(self.pts_bbox_refine_out, std=0.01)
        normal_init(self.reppoints_hem_tl_score_out, std=0.01, bias=bias_cls)
        normal_init(self.reppoints_hem_tl_offset_out, std=0.01)
        normal_init(self.reppoints_hem_br_score_out, std=0.01, bias=bias_cls)
        normal_init(self.reppoints_hem_br_offset_out, std=0.01)
        normal_init(self.reppoints_sem_out, std=0.01, bias=bias_cls)
        normal_init(self.cls_feat_conv, std=0.01)
        normal_init(self.bbox_feat_conv, std=0.01)
        normal_init(self.cls_af_dcn_conv[0], std=0.01)
        normal_init(self.bbox_af_dcn_conv[0], std=0.01)

    # ------------------------------------------------------------------------------------ forward
    def forward(self, feats):
        """-> (cls, bbox_init, bbox_refine, corner scores (B,2,H,W), corner offsets (B,4,H,W), semantic scores),
        each a list over the FPN levels (lscpvnet_head.py:262-368)."""
        nl = len(feats)
        shapes = [tuple(f.shape[2:]) for f in feats]
        base_offset = self.dcn_base_offset.type_as(feats[0])
        cls_tower = self._run_tower(self.cls_convs, feats)
        bbox_tower = self._run_tower(self.bbox_convs, feats)
        shared = self._run_tower(self.shared_convs, bbox_tower)

        sem_scores = self._split_px(self.reppoints_sem_out(self._cat_px(shared)), shapes)       # 1x1: one launch
        cls_feats, bbox_feats, hem_scores, hem_offsets, init_feats = [], [], [], [], []
        for l in range(nl):
            sem = self.reppoints_sem_embedding(shared[l])
            corner_in = shared[l] + sem
            tl, br = self.hem_tl(corner_in), self.hem_br(corner_in)
            score = torch.cat([self.reppoints_hem_tl_score_out(tl), self.reppoints_hem_br_score_out(br)], dim=1)
            offset = torch.cat([self.reppoints_hem_tl_offset_out(tl), self.reppoints_hem_br_offset_out(br)], dim=1)
            hem_scores.append(score)
            hem_offsets.append(offset)
            bbox_feat = bbox_tower[l] + sem
            init_feats.append(self.pts_bbox_init_conv(bbox_feat))
            extra = [score, offset]
            if self.gather_pad:
                extra.append(score.new_zeros(score.shape[0], self.gather_pad, *score.shape[2:]))
            cls_feats.append(torch.cat([cls_tower[l] + sem] + extra, dim=1))
            bbox_feats.append(torch.cat([bbox_feat] + extra, dim=1))

        raw = self.pts_bbox_init_out(self.relu(self._cat_px(init_feats)))          # pixel-wise from here: all levels
        sp_all = self.softplus(raw[:, :20])
        reg = self.get_pred_reg(sp_all, raw[:, 20:])
        reg = (1 - self.gradient_mul) * reg.detach() + self.gradient_mul * reg
        init_sp = self._split_px(sp_all, shapes)
        offsets = self._split_px(reg - base_offset, shapes)

        # the three source levels of every target level, offsets rescaled cumulatively as in LSHead.forward
        pairs, scaled = [], []
        for l in range(nl):
            bh, bw = shapes[l]
            cur = offsets[l]
            for s in self._level_list(l, nl):
                sh, sw = shapes[s][0] / bh, shapes[s][1] / bw
                cur = cur * self._scale_const(sh, sw, cur)
                scaled.append(cur)
                pairs.append((l, s, sh, sw))
        scales = [(p[2], p[3]) for p in pairs]

        def wide(w):                                           # (Co, C, kh, kw) -> (Co, C + pad, kh, kw)
            return F.pad(w, (0, 0, 0, 0, 0, self.gather_pad)) if self.gather_pad else w
        cls_raw = self.pts_cls_conv.forward_multi([cls_feats[p[1]] for p in pairs], scaled, scales,
                                                  weight=wide(self.pts_cls_conv.weight))
        box_raw = self.pts_bbox_refine_conv.forward_multi([bbox_feats[p[1]] for p in pairs], scaled, scales,
                                                          weight=wide(self.pts_bbox_refine_conv.weight))
        w_cls, w_box = wide(self.cls_feat_conv.weight), wide(self.bbox_feat_conv.weight)
        cls_fused = [self.cls_af_dcn_conv(torch.cat(cls_raw[3 * l:3 * l + 3], dim=1)) +
                     self.cls_feat_conv(cls_feats[l], weight=w_cls) for l in range(nl)]
        box_fused = [self.bbox_af_dcn_conv(torch.cat(box_raw[3 * l:3 * l + 3], dim=1)) +
                     self.bbox_feat_conv(bbox_feats[l], weight=w_box) for l in range(nl)]
        cls_out = self._split_px(self.pts_cls_out(self._cat_px(self.cls_GN.forward_multi(cls_fused, relu=True))), shapes)
        refine = self.pts_bbox_refine_out(self._cat_px(self.bbox_GN.forward_multi(box_fused, relu=True)))
        refine_sp = self._split_px(self.softplus(refine + sp_all.detach()), shapes)
        return cls_out, init_sp, refine_sp, hem_scores, hem_offsets, sem_scores

    def forward_train(self, x, img_metas, gt_bboxes, gt_labels=None, gt_bboxes_ignore=None, gt_sem_map=None,
                      gt_sem_weights=None, gt_extremes=None, **kwargs):
        return self.loss(*self(x), gt_bboxes, gt_extremes, gt_sem_map, gt_sem_weights, gt_labels, img_metas,
                         gt_bboxes_ignore=gt_bboxes_ignore)

    # --------------------------------------------------------------------------------------- loss
    def get_hm_targets(self, flat_points, flat_flags, all_valid, gt_bboxes):
        """Corner heat-map / offset targets for every image over ALL points (lscpvnet_head.py:559-668 written
        densely): dict of (B, N_all[, 2]) tensors + the positive counts sum_img max(n_pos, 1) as device scalars."""
        per_img = []
        for i, boxes in enumerate(gt_bboxes):
            inside = None if all_valid else flat_flags[i]
            pts = flat_points if inside is None else flat_points[inside]
            hm_tl, off_tl, hm_br, off_br = self.hm_assigner.assign_dense(pts, boxes, strides=self.point_strides)
            t = dict(hm_tl=hm_tl.float(), off_tl=off_tl, hm_br=hm_br.float(), off_br=off_br)
            for c in ('tl', 'br'):
                pos = t[f'hm_{c}'] == 1
                t[f'hm_w_{c}'] = (pos | (t[f'hm_{c}'] < 1)).float()
                t[f'off_w_{c}'] = pos.float().unsqueeze(1).expand(-1, 2)
                t[f'n_{c}'] = pos.sum()
            if inside is not None:
                for k, v in list(t.items()):
                    if v.dim() == 0:
                        continue
                    buf = v.new_zeros((flat_points.shape[0],) + tuple(v.shape[1:]))
                    buf[inside] = v
                    t[k] = buf
            per_img.append(t)
        out = {k: torch.stack([t[k] for t in per_img], 0) for k in per_img[0] if not k.startswith('n_')}
        n_tl = sum(t['n_tl'].clamp(min=1) for t in per_img)
        n_br = sum(t['n_br'].clamp(min=1) for t in per_img)
        return out, n_tl, n_br

    def loss(self, cls_scores, bbox_pts_preds_init, bbox_pts_preds_refine, hm_scores, hm_offsets, sem_scores,
             gt_bboxes, gt_extremes, gt_sem_map, gt_sem_weights, gt_labels, img_metas, gt_bboxes_ignore=None):
        """lscpvnet_head.py:670-903: the bbox-task losses of LSHead + corner heat-map, corner offset and semantic
        losses."""
        none = [None] * len(cls_scores)
        out = LSHead.loss(self, cls_scores, bbox_pts_preds_init, bbox_pts_preds_refine, none, none, none, none,
                          gt_bboxes, gt_extremes, None, None, gt_labels, img_metas, gt_bboxes_ignore=gt_bboxes_ignore)
        device = cls_scores[0].device
        featmap_sizes = [tuple(m.shape[-2:]) for m in cls_scores]
        points, flags, all_valid = self.get_points(featmap_sizes, img_metas, device)
        num_level = [p.shape[0] for p in points]
        tg, n_tl, n_br = self.get_hm_targets(torch.cat(points), [torch.cat(f) for f in flags], all_valid, gt_bboxes)
        assert cls_scores[0].shape[0] == gt_sem_map.shape[0] == len(gt_bboxes)

        def levels(t):
            return torch.split(t, num_level, dim=1)
        heat, offs = [], []
        for lvl in range(len(cls_scores)):
            score = hm_scores[lvl].permute(0, 2, 3, 1).reshape(-1, 2).sigmoid()
            off = hm_offsets[lvl].permute(0, 2, 3, 1).reshape(-1, 4)
            lh = lo = 0
            for c, n, col in (('tl', n_tl, 0), ('br', n_br, 1)):
                lh = lh + self.loss_heatmap(score[:, col], levels(tg[f'hm_{c}'])[lvl].reshape(-1),
                                            levels(tg[f'hm_w_{c}'])[lvl].reshape(-1), avg_factor=n)
                lo = lo + self.loss_offset(off[:, 2 * col:2 * col + 2], levels(tg[f'off_{c}'])[lvl].reshape(-1, 2),
                                           levels(tg[f'off_w_{c}'])[lvl].reshape(-1, 2), avg_factor=n)
            heat.append(lh / 2.0)
            offs.append(lo / 2.0)
        out['loss_heatmap'], out['loss_offset'] = heat, offs

        sem_pred = torch.cat([s.reshape(-1) for s in sem_scores])
        sem_gt = torch.cat([F.interpolate(gt_sem_map, s.shape[-2:]).reshape(-1) for s in sem_scores])
        sem_w = torch.cat([F.interpolate(gt_sem_weights, s.shape[-2:]).reshape(-1) for s in sem_scores])
        out['loss_sem'] = self.loss_sem(sem_pred, sem_gt, sem_w, avg_factor=(sem_gt > 0).sum())
        return out

    # ----------------------------------------------------------------------------------- decoding
    def get_bboxes(self, cls_scores, bbox_pts_preds_init, bbox_pts_preds_refine, hm_scores, hm_offsets, sem_scores,
                   img_metas, cfg=None, rescale=False, nms=True):
        """lscpvnet_head.py:905-951"""
        assert len(cls_scores) == len(bbox_pts_preds_refine)
        box_maps = [self.extreme_points2bbox(p) for p in bbox_pts_preds_refine]
        device = cls_scores[0].device
        points = [self.point_generators[i].grid_points(cls_scores[i].shape[-2:], self.point_strides[i], device)
                  for i in range(len(cls_scores))]
        results = []
        for i, meta in enumerate(img_metas):
            results.append(self._get_bboxes_single([c[i].detach() for c in cls_scores], [m[i].detach() for m in box_maps],
                                                   [h[i].detach() for h in hm_scores], [h[i].detach() for h in hm_offsets],
                                                   points, meta['img_shape'], meta['scale_factor'], cfg, rescale, nms))
        return results

    def _snap_to_corner(self, score_map, x, y, stride):
        """Corner verification of one corner type: the cell (x, y)/stride floor-ed, the arg-max of the 2x2 window
        whose bottom-right cell it is (max_pool2d kernel 2, stride 1, padding 0 -> window starting at the cell; the
        pooled map is one cell smaller, indices clamp to it) -> (col, row) of that maximum (lscpvnet_head.py:965-986)."""
        H, W = score_map.shape[-2:]
        pooled, idx = F.max_pool2d_with_indices(score_map.sigmoid()[None, None], kernel_size=2, stride=1, padding=0)
        idx = idx[0, 0]
        xr = torch.floor((x / stride).clamp(min=0, max=pooled.shape[-1] - 1)).long()
        yr = torch.floor((y / stride).clamp(min=0, max=pooled.shape[-2] - 1)).long()
        sel = idx[yr, xr]
        return sel % W, sel // W

    def _get_bboxes_single(self, cls_scores, bbox_preds, hm_scores, hm_offsets, mlvl_points, img_shape, scale_factor,
                           cfg, rescale=False, nms=True):
        """lscpvnet_head.py:953-1090"""
        cfg = self.test_cfg if cfg is None else cfg
        assert len(cls_scores) == len(bbox_preds) == len(mlvl_points)
        boxes_all, scores_all = [], []
        for lvl, (cls_score, bbox_pred, points) in enumerate(zip(cls_scores, bbox_preds, mlvl_points)):
            assert cls_score.shape[-2:] == bbox_pred.shape[-2:]
            scores = cls_score.permute(1, 2, 0).reshape(-1, self.cls_out_channels).sigmoid()
            bbox_pred = bbox_pred.permute(1, 2, 0).reshape(-1, 4)
            nms_pre = cfg.get('nms_pre', -1)
            if 0 < nms_pre < scores.shape[0]:
                _, keep = scores.max(dim=1)[0].topk(nms_pre)
                points, bbox_pred, scores = points[keep], bbox_pred[keep], scores[keep]
            xy = points[:, :2]
            bboxes = bbox_pred * self.point_strides[lvl] + torch.cat([xy, xy], dim=1)
            x1 = bboxes[:, 0].clamp(min=0, max=img_shape[1])
            y1 = bboxes[:, 1].clamp(min=0, max=img_shape[0])
            x2 = bboxes[:, 2].clamp(min=0, max=img_shape[1])
            y2 = bboxes[:, 3].clamp(min=0, max=img_shape[0])
            if lvl > 0:
                src = 0 if lvl in (1, 2) else 1                 # verify against the stride-8 or stride-16 corner maps
                stride = self.point_strides[src]
                off = hm_offsets[src].permute(1, 2, 0)
                cx1, cy1 = self._snap_to_corner(hm_scores[src][0], x1, y1, stride)
                cx2, cy2 = self._snap_to_corner(hm_scores[src][1], x2, y2, stride)
                x1 = ((cx1.float() + off[cy1, cx1, 0]) * stride).clamp(min=0, max=img_shape[1])
                y1 = ((cy1.float() + off[cy1, cx1, 1]) * stride).clamp(min=0, max=img_shape[0])
                x2 = ((cx2.float() + off[cy2, cx2, 2]) * stride).clamp(min=0, max=img_shape[1])
                y2 = ((cy2.float() + off[cy2, cx2, 3]) * stride).clamp(min=0, max=img_shape[0])
            boxes_all.append(torch.stack([x1, y1, x2, y2], dim=-1))
            scores_all.append(scores)
        mlvl_bboxes, mlvl_scores = torch.cat(boxes_all), torch.cat(scores_all)
        if rescale:
            mlvl_bboxes = mlvl_bboxes / mlvl_bboxes.new_tensor(scale_factor)
        mlvl_scores = torch.cat([mlvl_scores, mlvl_scores.new_zeros(mlvl_scores.shape[0], 1)], dim=1)
        if not nms:
            return mlvl_bboxes, mlvl_scores
        return multiclass_nms(mlvl_bboxes, mlvl_scores, cfg.score_thr, cfg.nms, cfg.max_per_img)
