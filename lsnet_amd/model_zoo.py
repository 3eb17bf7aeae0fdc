"""Programmatic equivalents of the reference's LSNet configs (configs/lsnet/*.py +
configs/_base_/{schedules,default_runtime}.py), for places where the config FILES are not available
(the reference tree does not travel to the GPU box).  `Config.fromfile` on the original files gives
the same dicts; tests/test_configs.py checks that where /root/reference exists.

Hyper-parameters only (data); see the cited config lines for their origin."""
import copy

from .utils import Config

NORM_GN = dict(type='GN', num_groups=32, requires_grad=True)


def backbone_cfg(name='r50'):
    """r50 / r101 (lsnet_bbox_r50_fpn_1x_coco.py:9-18), res2-101 = Res2Net-101 26w x 4s
    (lsnet_segm_res2_101_fpn_dconv_c3-c5_mstrain_30e_coco.py:6-14), x101 = ResNeXt-101-64x4d
    (lsnet_bbox_x101_fpn_mstrain_2x_coco.py), `-dcn` suffix = DCNv2 in c3-c5
    (lsnet_bbox_x101_fpn_dconv_c3-c5_mstrain_2x_coco.py.py:4-17)."""
    dcn = name.endswith('-dcn')
    base = name[:-4] if dcn else name
    common = dict(num_stages=4, out_indices=(0, 1, 2, 3), frozen_stages=1,
                  norm_cfg=dict(type='BN', requires_grad=True), norm_eval=True, style='pytorch')
    if base in ('r50', 'r101'):
        cfg = dict(type='ResNet', depth=int(base[1:]), **common)
    elif base == 'x101':
        cfg = dict(type='ResNeXt', depth=101, groups=64, base_width=4, **common)
    elif base in ('res2-50', 'res2-101'):   # lsnet_*_res2_101_fpn_dconv_c3-c5_*: 26w x 4s, always with `-dcn`
        cfg = dict(type='Res2Net', depth=int(base.split('-')[1]), scales=4, base_width=26, **common)
    else:
        raise KeyError(base)
    if dcn:
        cfg.update(dcn=dict(type='DCNv2', deformable_groups=1, fallback_on_stride=False),
                   stage_with_dcn=(False, True, True, True))
        if base in ('x101', 'res2-50', 'res2-101'):
            cfg['with_cp'] = True
    return cfg


def head_cfg(task='bbox', conv_module_type='dcn'):
    """bbox: lsnet_bbox_r50_fpn_1x_coco.py:27-47; segm: lsnet_segm_r50_fpn_1x_coco.py:40-59;
    pose_bbox / pose_kbox: lsnet_pose_bbox_r50_fpn_1x_coco.py:27-46."""
    if task == 'bbox_cpv':     # lsnet_bbox_cpv_x101_fpn_dconv_c3-c5_mstrain_2x_coco.py:20-56
        return dict(type='LSCPVHead', num_classes=80, in_channels=256, feat_channels=256, point_feat_channels=256,
                    stacked_convs=3, shared_stacked_convs=1, first_kernel_size=3, kernel_size=1, corner_dim=64,
                    num_points=9, gradient_mul=0.1, point_strides=[8, 16, 32, 64, 128], point_base_scale=4,
                    norm_cfg=NORM_GN, conv_module_type=conv_module_type,
                    loss_cls=dict(type='FocalLoss', use_sigmoid=True, gamma=2.0, alpha=0.25, loss_weight=1.0),
                    loss_bbox_init=dict(type='CrossIOULoss', loss_weight=1.0),
                    loss_bbox_refine=dict(type='CrossIOULoss', loss_weight=2.0),
                    loss_heatmap=dict(type='GaussianFocalLoss', alpha=2.0, gamma=4.0, loss_weight=0.25),
                    loss_offset=dict(type='SmoothL1Loss', beta=1.0 / 9.0, loss_weight=1.0),
                    loss_sem=dict(type='SEPFocalLoss', gamma=2.0, alpha=0.25, loss_weight=0.1))
    nv = {'bbox': 4, 'segm': 36, 'pose_bbox': 17, 'pose_kbox': 17}[task]
    cfg = dict(type='LSHead', task=task, num_vectors=nv, num_classes=80 if 'pose' not in task else 1,
               in_channels=256, feat_channels=256, point_feat_channels=256, stacked_convs=3, num_kernel_points=9,
               gradient_mul=0.1, point_strides=[8, 16, 32, 64, 128], point_base_scale=4, norm_cfg=NORM_GN,
               conv_module_type=conv_module_type,
               loss_cls=dict(type='FocalLoss', use_sigmoid=True, gamma=2.0, alpha=0.25, loss_weight=1.0))
    if task == 'bbox':
        cfg.update(loss_bbox_init=dict(type='CrossIOULoss', loss_weight=1.0),
                   loss_bbox_refine=dict(type='CrossIOULoss', loss_weight=2.0))
    elif task == 'segm':
        cfg.update(loss_segm_init=dict(type='CrossIOULoss', loss_weight=1.0, loss_type='polygon', stride=9),
                   loss_segm_refine=dict(type='CrossIOULoss', loss_weight=2.0, loss_type='polygon', stride=9))
    else:
        if task == 'pose_bbox':
            cfg.update(loss_bbox_init=dict(type='CrossIOULoss', loss_weight=0.1, loss_type='bbox'),
                       loss_bbox_refine=dict(type='CrossIOULoss', loss_weight=0.2, loss_type='bbox'))
        cfg.update(loss_pose_init=dict(type='CrossIOULoss', loss_weight=1.0, loss_type='keypoint'),
                   loss_pose_refine=dict(type='CrossIOULoss', loss_weight=2.0, loss_type='keypoint'))
    return cfg


def lsnet_config(task='bbox', backbone='r50', conv_module_type='dcn', lr=0.01, max_per_img=None):
    cpv = task == 'bbox_cpv'
    model = dict(type='LSCPVDetector' if cpv else 'LSDetector', pretrained=None, backbone=backbone_cfg(backbone),
                 neck=dict(type='FPN', in_channels=[256, 512, 1024, 2048], out_channels=256, start_level=1,
                           add_extra_convs='on_input', num_outs=5, norm_cfg=NORM_GN),
                 bbox_head=head_cfg(task, conv_module_type))
    train_cfg = dict(init=dict(assigner=dict(type='CentroidAssigner', scale=4, pos_num=1, iou_type='center'),
                               allowed_border=-1, pos_weight=-1, debug=False),
                     refine=dict(assigner=dict(type='ATSSAssigner', topk=9), allowed_border=-1, pos_weight=-1,
                                 debug=False))
    if cpv:                    # lsnet_bbox_cpv_x101_*.py:58-73
        train_cfg['heatmap'] = dict(assigner=dict(type='PointHMAssigner', gaussian_bump=True, gaussian_iou=0.7),
                                    allowed_border=-1, pos_weight=-1, debug=False)
    pose = 'pose' in task
    test_cfg = dict(nms_pre=100 if pose else 1000, min_bbox_size=0, score_thr=0.05,
                    nms=dict(type='nms', iou_thr=0.6), max_per_img=max_per_img or (20 if pose else 100))
    return Config(dict(
        model=model, train_cfg=train_cfg, test_cfg=test_cfg,
        # schedule_1x.py:2-11 with the LSNet overrides (lsnet_bbox_r50_fpn_1x_coco.py:64-65)
        optimizer=dict(type='SGD', lr=lr, momentum=0.9, weight_decay=0.0001),
        optimizer_config=dict(grad_clip=dict(max_norm=35, norm_type=2)),
        lr_config=dict(policy='step', warmup='linear', warmup_iters=500, warmup_ratio=0.001, step=[8, 11]),
        total_epochs=12, checkpoint_config=dict(interval=1),
        log_config=dict(interval=50, hooks=[dict(type='TextLoggerHook')]),
        dist_params=dict(backend='nccl'), log_level='INFO', load_from=None, resume_from=None,
        workflow=[('train', 1)], data=dict(samples_per_gpu=2, workers_per_gpu=2)))


def build_lsnet(task='bbox', backbone='r50', **kw):
    from .models import build_detector
    cfg = lsnet_config(task, backbone, **kw)
    return build_detector(copy.deepcopy(cfg.model), train_cfg=cfg.train_cfg, test_cfg=cfg.test_cfg), cfg
