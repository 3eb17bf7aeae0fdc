"""One training step of the other BASELINE configurations on the GPU: R-101-DCN bbox (DCNv2 in the backbone),
X-101-64x4d-DCN segm (grouped DCNv2 + activation checkpointing), R-50 pose head, Res2Net-101-DCN.

Opt-in (LSNET_SLOW_TESTS=1): each case spends about a minute in MIOpen's first-call kernel search.  State at the end of
round 1: the R-101-DCN case ran forward, backward and the optimizer step on an MI355X; its first version then failed
on an over-strict "90 % of all parameters moved" check (with zero-initialised last norms the residual branches get
exactly zero gradient at the first iteration) -- relaxed below to the head's parameters, not re-run since (GPU budget)."""
import os

import pytest
import torch

from lsnet_amd.data import synthetic_batch
from lsnet_amd.model_zoo import build_lsnet
from lsnet_amd.runner import EpochBasedRunner, build_optimizer


@pytest.mark.gpu
@pytest.mark.skipif(os.environ.get('LSNET_SLOW_TESTS') != '1', reason='slow (MIOpen kernel search per new shape); LSNET_SLOW_TESTS=1')
@pytest.mark.parametrize('task,backbone', [('bbox', 'r101-dcn'), ('segm', 'x101-dcn'), ('pose_bbox', 'r50'),
                                           ('bbox', 'res2-101-dcn')])
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
