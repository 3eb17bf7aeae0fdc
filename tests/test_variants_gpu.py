"""One training step of the other BASELINE configurations on the GPU: R-101-DCN bbox (config 3: DCNv2 in the backbone)
and X-101-64x4d-DCN segm (config 4: grouped DCNv2 + activation checkpointing) run by default; R-50 pose head and
Res2Net-101-DCN are opt-in (LSNET_SLOW_TESTS=1).  Loss finite, the head's parameters move, the backbone's deformable
convs receive finite gradients.  (Feature / gradient parity of these backbones against the reference:
tests/test_golden_gpu.py::test_dcn_backbones_of_configs_3_and_4.)"""
import os

import pytest
import torch

from lsnet_amd.data import synthetic_batch
from lsnet_amd.model_zoo import build_lsnet
from lsnet_amd.runner import EpochBasedRunner, build_optimizer


SLOW = pytest.mark.skipif(os.environ.get('LSNET_SLOW_TESTS') != '1', reason='opt-in: LSNET_SLOW_TESTS=1')


@pytest.mark.gpu
@pytest.mark.parametrize('task,backbone', [('bbox', 'r101-dcn'), ('segm', 'x101-dcn'),
                                           pytest.param('pose_bbox', 'r50', marks=SLOW),
                                           pytest.param('bbox', 'res2-101-dcn', marks=SLOW)])
def test_one_training_step(task, backbone):
    dev = torch.device('cuda:0')
    torch.manual_seed(0)
    model, cfg = build_lsnet(task, backbone)
    model = model.to(dev).to(memory_format=torch.channels_last).train()
    opt = build_optimizer(model, cfg.optimizer)
    r = EpochBasedRunner(model, optimizer=opt, logger=lambda s: None)
    r.register_training_hooks(cfg.lr_config, cfg.optimizer_config, None, dict(interval=10 ** 9, hooks=[]))
    batch = synthetic_batch(task, 2, 384, 480, seed=9, device=dev)
    before = {k: v.detach().clone() for k, v in model.bbox_head.named_parameters() if v.requires_grad}
    r.run([[batch, batch]], [('train', 1)], 1)
    loss = float(r.outputs['log_vars']['loss'])
    assert loss == loss and 0 < loss < 100, loss
    moved = sum(int(not torch.equal(before[k], v.detach())) for k, v in model.bbox_head.named_parameters()
                if v.requires_grad)
    assert moved > 0.8 * len(before), (moved, len(before))
    if 'dcn' in backbone:   # the backbone's deformable convs received gradients
        g = [p.grad for n, p in model.backbone.named_parameters() if 'conv_offset' in n]
        assert g and all(t is not None and torch.isfinite(t).all() for t in g)
