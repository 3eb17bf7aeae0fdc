"""Config machinery, model zoo vs the reference's config files, registry build of every LSNet variant."""
import glob
import os

import pytest

from lsnet_amd.model_zoo import build_lsnet, lsnet_config
from lsnet_amd.utils import Config

HERE = os.path.dirname(os.path.abspath(__file__))
REF_CFG = '/root/reference/code/configs/lsnet'


def test_base_merge_delete_and_override():
    cfg = Config.fromfile(os.path.join(HERE, 'data', 'child_cfg.py'))
    assert cfg.optimizer == dict(type='SGD', lr=0.01, momentum=0.9, weight_decay=0.0001)
    assert cfg.lr_config.step == [16, 22] and cfg.lr_config.warmup == 'linear'
    assert cfg.total_epochs == 24
    assert cfg.model.backbone == dict(type='ResNeXt', depth=101, groups=64, base_width=4)   # _delete_
    assert cfg.model.neck == dict(type='FPN', num_outs=5, norm_cfg=dict(type='GN', num_groups=32))
    cfg.merge_from_dict({'model.neck.num_outs': 4, 'optimizer.lr': 0.5})
    assert cfg.model.neck.num_outs == 4 and cfg.optimizer.lr == 0.5
    assert cfg.get('missing', 7) == 7
    with pytest.raises(AttributeError):
        cfg.nope


@pytest.mark.skipif(not os.path.isdir(REF_CFG), reason='reference configs are not on this machine')
def test_all_reference_lsnet_configs_load():
    files = sorted(glob.glob(os.path.join(REF_CFG, '*.py')))
    assert len(files) == 17
    for f in files:
        cfg = Config.fromfile(f)
        assert cfg.model.type in ('LSDetector', 'LSCPVDetector')
        assert cfg.model.bbox_head.type in ('LSHead', 'LSCPVHead')
        assert 'train_cfg' in cfg and 'test_cfg' in cfg and cfg.optimizer.type == 'SGD'


@pytest.mark.skipif(not os.path.isdir(REF_CFG), reason='reference configs are not on this machine')
@pytest.mark.parametrize('fname,task,backbone', [
    ('lsnet_bbox_r50_fpn_1x_coco.py', 'bbox', 'r50'),
    ('lsnet_segm_r50_fpn_1x_coco.py', 'segm', 'r50'),
    ('lsnet_pose_bbox_r50_fpn_1x_coco.py', 'pose_bbox', 'r50'),
    ('lsnet_bbox_x101_fpn_dconv_c3-c5_mstrain_2x_coco.py.py', 'bbox', 'x101-dcn'),
    ('lsnet_segm_res2_101_fpn_dconv_c3-c5_mstrain_30e_coco.py', 'segm', 'res2-101-dcn'),
    ('lsnet_bbox_cpv_x101_fpn_dconv_c3-c5_mstrain_2x_coco.py', 'bbox_cpv', 'x101-dcn'),
])
def test_model_zoo_equals_reference_config(fname, task, backbone):
    """The zoo is what bench.py uses on the GPU box (no config files there): same model dicts."""
    ref = Config.fromfile(os.path.join(REF_CFG, fname))
    zoo = lsnet_config(task, backbone)

    def norm(d):   # tuples vs lists and ConfigDict vs dict do not matter
        if isinstance(d, dict):
            return {k: norm(v) for k, v in d.items()}
        if isinstance(d, (list, tuple)):
            return [norm(v) for v in d]
        return d
    r, z = norm(ref.model), norm(zoo.model)
    r.pop('pretrained', None), z.pop('pretrained', None)
    assert r['backbone'] == z['backbone']
    assert r['neck'] == z['neck']
    assert r['bbox_head'] == z['bbox_head']
    assert norm(ref.train_cfg) == norm(zoo.train_cfg)
    assert norm(ref.test_cfg) == norm(zoo.test_cfg)
    assert norm(ref.optimizer) == norm(zoo.optimizer)
    assert norm(ref.optimizer_config) == norm(zoo.optimizer_config)


@pytest.mark.parametrize('task,backbone,params_m', [
    ('bbox', 'r50', 38.80), ('segm', 'r50', 38.87), ('pose_bbox', 'r50', 42.74), ('bbox', 'x101-dcn', 104.41)])
def test_registry_builds_every_variant(task, backbone, params_m):
    model, cfg = build_lsnet(task, backbone)
    n = sum(p.numel() for p in model.parameters()) / 1e6
    assert abs(n - params_m) < 0.12, n        # measured on the reference
    keys = set(model.state_dict())
    for k in ('bbox_head.cls_convs.0.conv.weight', 'bbox_head.cls_convs.0.conv.conv_offset.weight',
              'bbox_head.pts_cls_conv.weight', 'bbox_head.cls_GN.weight', 'neck.lateral_convs.0.conv.weight',
              'backbone.layer1.0.conv1.weight'):
        assert k in keys, k
    frozen = [n_ for n_, p in model.named_parameters() if not p.requires_grad]
    assert any(n_.startswith('backbone.layer1') for n_ in frozen) and not any('layer2' in n_ for n_ in frozen)


@pytest.mark.skipif(not os.path.isdir(REF_CFG), reason='reference configs are not on this machine')
def test_data_sections_of_reference_configs_build():
    """Every `data.train/val/test` pipeline of configs/lsnet/*.py is made of stages this package registers, with the
    arguments those files pass (the dataset classes themselves need the annotation files, so only the type is checked)."""
    from lsnet_amd.data import DATASETS, Compose
    from lsnet_amd.utils import Config
    seen = set()
    for f in sorted(os.listdir(REF_CFG)):
        cfg = Config.fromfile(os.path.join(REF_CFG, f))
        for split in ('train', 'val', 'test'):
            d = cfg.data[split]
            assert d['type'] in DATASETS, (f, d['type'])
            pipe = Compose(d['pipeline'])
            seen |= {type(t).__name__ for t in pipe.transforms}
            assert repr(pipe)
        assert cfg.data.samples_per_gpu in (2, 6)                 # 6 for the pose configs
    assert {'LoadImageFromFile', 'LoadAnnotations', 'Resize', 'RandomFlip', 'Normalize', 'Pad', 'DefaultFormatBundle',
            'Collect', 'MultiScaleFlipAug'} <= seen
