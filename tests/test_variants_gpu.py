"""One training step of the other BASELINE configurations on the GPU: R-101-DCN bbox (config 3: DCNv2 in the backbone),
X-101-64x4d-DCN segm (config 4: grouped DCNv2 + activation checkpointing) and the R-50 pose head, at a reduced image size and
-- `test_full_size_*` -- at the 800 x 1344 of BASELINE.json (tile tables, 32-bit offset guards and checkpointing memory
at the real shapes); one inference batch of config 5 (pose head, 4 images) at full size; Res2Net-101-DCN (the headline
53.5-AP backbone) at the reduced size.  Loss finite, the head's parameters move, the backbone's deformable
convs receive finite gradients.  (Feature / gradient parity of these backbones against the reference:
tests/test_golden_gpu.py::test_dcn_backbones_of_configs_3_and_4.)"""

import pytest
import torch

from lsnet_amd.data import synthetic_batch
from lsnet_amd.model_zoo import build_lsnet
from lsnet_amd.runner import EpochBasedRunner, build_optimizer


@pytest.mark.gpu
@pytest.mark.parametrize('task,backbone', [('bbox', 'r101-dcn'), ('segm', 'x101-dcn'),
                                           ('pose_bbox', 'r50'),
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


@pytest.mark.gpu
@pytest.mark.parametrize('task,backbone', [('bbox', 'r101-dcn'), ('segm', 'x101-dcn')])
def test_full_size_training_step(task, backbone):
    """BASELINE configs 3 and 4 at 2 x 3 x 800 x 1344: forward, losses, backward, clip, SGD through the runner's hooks --
    twice, so that the second step runs on refreshed weight images and learnt gradient-contribution counts."""
    from lsnet_amd.parallel import DataParallelModel
    dev = torch.device('cuda:0')
    torch.manual_seed(0)
    model, cfg = build_lsnet(task, backbone)
    model = DataParallelModel(model.to(dev).to(memory_format=torch.channels_last).train())
    opt = build_optimizer(model, cfg.optimizer)
    r = EpochBasedRunner(model, optimizer=opt, logger=lambda s: None)
    r.register_training_hooks(cfg.lr_config, cfg.optimizer_config, None, dict(interval=10 ** 9, hooks=[]))
    batch = synthetic_batch(task, 2, 800, 1344, seed=9, device=dev)
    before = {k: v.detach().clone() for k, v in model.module.bbox_head.named_parameters() if v.requires_grad}
    r.run([[batch, batch]], [('train', 1)], 1)
    loss = float(r.outputs['log_vars']['loss'])
    assert loss == loss and 0 < loss < 200, loss
    moved = sum(int(not torch.equal(before[k], v.detach())) for k, v in model.module.bbox_head.named_parameters()
                if v.requires_grad)
    assert moved > 0.8 * len(before), (moved, len(before))
    g = [p.grad for n, p in model.module.backbone.named_parameters() if 'conv_offset' in n]
    assert g and all(t is not None and torch.isfinite(t).all() for t in g)
    assert all(torch.isfinite(p).all() for p in model.parameters())


@pytest.mark.gpu
def test_config3_multiscale_training_shapes_change_every_iteration():
    """BASELINE config 3 the way it trains (configs/lsnet/lsnet_bbox_r50_fpn_mstrain_2x_coco.py:13-15: img_scale 480 .. 960
    rows, multiscale_mode='range'; group_sampler.py:60-140: landscape and portrait groups): R-101-DCN bbox through the
    runner for 12 iterations whose padded batch shape cycles 800x1344, 960x1344, 640x1344, 1344x800 (portrait), 480x1344.
    Tile tables, stream-K plans, 32-bit offset guards, the library's scratch growth and the batched weight gradients all see
    a new shape every step.  Finite losses, parameters move, and after the first pass over the five shapes the library asks
    the HIP runtime for NOTHING any more: no hipMalloc, no blocking synchronisation (lsn_scratch_stats)."""
    from lsnet_amd import _lib
    from lsnet_amd.parallel import DataParallelModel
    dev = torch.device('cuda:0')
    torch.manual_seed(0)
    model, cfg = build_lsnet('bbox', 'r101-dcn')
    model = DataParallelModel(model.to(dev).to(memory_format=torch.channels_last).train())
    opt = build_optimizer(model, cfg.optimizer)
    r = EpochBasedRunner(model, optimizer=opt, logger=lambda s: None)
    r.register_training_hooks(cfg.lr_config, cfg.optimizer_config, None, dict(interval=10 ** 9, hooks=[]))
    shapes = [(800, 1344), (960, 1344), (640, 1344), (1344, 800), (480, 1344)]
    batches = [synthetic_batch('bbox', 2, h, w, seed=20 + i, device=dev) for i, (h, w) in enumerate(shapes)]
    before = {k: v.detach().clone() for k, v in model.module.named_parameters() if v.requires_grad}
    losses, stats = [], []

    from lsnet_amd.runner import Hook

    class Probe(Hook):   # after every iteration (behind the optimizer hook): the loss and the library's counters
        priority = 99

        def after_train_iter(self, runner):
            losses.append(float(runner.outputs['log_vars']['loss']))
            stats.append(_lib.scratch_stats())
    r.register_hook(Probe())
    r.run([[batches[i % 5] for i in range(12)]], [('train', 1)], 1)
    assert len(losses) == 12 and all(l == l and 0 < l < 200 for l in losses), losses
    moved = sum(int(not torch.equal(before[k], v.detach())) for k, v in model.module.named_parameters() if v.requires_grad)
    assert moved > 0.9 * len(before), (moved, len(before))
    assert all(torch.isfinite(p).all() for p in model.parameters())
    first_pass, last = stats[4], stats[-1]
    assert last['mallocs'] == first_pass['mallocs'], ('the library allocated after it had seen every shape', first_pass, last)
    assert last['blocking_syncs'] == first_pass['blocking_syncs'], (first_pass, last)
    print('config 3 multi-scale: losses', [round(l, 3) for l in losses], 'library scratch', last)


@pytest.mark.gpu
def test_full_size_pose_inference_batch():
    """BASELINE config 5: R-50-FPN pose head, 4 images 3 x 800 x 1344, forward + decode + NMS on the device."""
    dev = torch.device('cuda:0')
    torch.manual_seed(0)
    model, _ = build_lsnet('pose_kbox', 'r50')
    model = model.to(dev).to(memory_format=torch.channels_last).eval()
    with torch.no_grad():
        model.bbox_head.pts_cls_out.bias.add_(2.0)    # random-init logits sit below score_thr: let detections through
        img = torch.randn(4, 3, 800, 1344, device=dev).contiguous(memory_format=torch.channels_last)
        metas = [dict(pad_shape=(800, 1344, 3), img_shape=(800, 1344, 3), scale_factor=1.0, ori_shape=(800, 1344, 3),
                      flip=False)] * 4
        dets = model.simple_test_batch(img, metas)
    assert len(dets) == 4
    for boxes, vectors, labels in dets:
        assert boxes.shape[0] == labels.shape[0] == vectors.shape[0] and boxes.shape[0] > 0
        assert torch.isfinite(boxes).all() and torch.isfinite(vectors).all()
        assert boxes.shape[1] == 5 and int(labels.min()) >= 0


@pytest.mark.gpu
@pytest.mark.parametrize('task,backbone', [('bbox', 'r50'), ('bbox', 'r101-dcn'), ('segm', 'x101-dcn')])
def test_training_step_is_bit_reproducible(task, backbone):
    """The same model, the same batch, twice: the loss and EVERY parameter gradient must come back with the same bits.
    What it takes: anchor lists sorted by sample id whatever their length, split partial tiles +
    ordered reduce in every weight-gradient kernel, block partials instead of fp32 atomics in the GroupNorm / BatchNorm
    parameter sums and the focal-loss sum.  (The reference scatters grad_input with fp32 atomics,
    deform_conv_cuda_kernel.cu:913-970: it is not reproducible.)"""
    dev = torch.device('cuda:0')

    def run():
        torch.manual_seed(3)
        model, _ = build_lsnet(task, backbone)
        model = model.to(dev).to(memory_format=torch.channels_last).train()
        data = synthetic_batch(task, 2, 384, 480, boxes_per_img=5, num_classes=80, seed=11, device='cuda:0', channels_last=True)
        losses = model(**data)
        loss = sum(v if torch.is_tensor(v) else sum(v) for k, v in losses.items() if 'loss' in k)
        loss.backward()
        torch.cuda.synchronize()
        return loss.detach().clone(), {n: p.grad.detach().clone() for n, p in model.named_parameters() if p.grad is not None}

    l1, g1 = run()
    l2, g2 = run()
    assert torch.equal(l1, l2), (float(l1), float(l2))
    assert len(g1) > 100
    diff = [n for n in g1 if not torch.equal(g1[n], g2[n])]
    assert not diff, f'{len(diff)} of {len(g1)} parameter gradients differ between two identical steps, e.g. {diff[:4]}'


@pytest.mark.gpu
@pytest.mark.parametrize('task', ['bbox', 'segm'])
def test_fused_level_sums_and_side_stream_targets_leave_the_step_unchanged(task, monkeypatch):
    """Round 5: every loss term's levels in one launch over the head's concatenated tensors (LSNET_FUSED_LEVEL_SUMS) and the
    target assignment on a second stream (LSNET_SIDE_STREAM_TARGETS) against the level-by-level, single-stream form: the same
    per-level loss terms (summation order only) and the same parameter gradients; identical targets, hence no tolerance for
    anything the assignment decides."""
    from lsnet_amd.models.dense_heads import ls_head
    dev = torch.device('cuda:0')

    def run(fused, side):
        monkeypatch.setattr(ls_head, 'FUSED_LEVEL_SUMS', fused)
        monkeypatch.setattr(ls_head, 'SIDE_STREAM_TARGETS', side)
        torch.manual_seed(3)
        model, _ = build_lsnet(task, 'r50')
        model = model.to(dev).to(memory_format=torch.channels_last).train()
        data = synthetic_batch(task, 2, 384, 480, boxes_per_img=5, num_classes=80, seed=11, device='cuda:0', channels_last=True)
        losses = model(**data)
        loss = sum(v if torch.is_tensor(v) else sum(v) for k, v in losses.items() if 'loss' in k)
        loss.backward()
        torch.cuda.synchronize()
        terms = {k: torch.stack([t.detach().reshape(()) for t in (v if isinstance(v, list) else [v])]) for k, v in losses.items()}
        return terms, {n: p.grad.detach().clone() for n, p in model.named_parameters() if p.grad is not None}

    t0, g0 = run(False, False)
    t_side, g_side = run(False, True)
    for k in t0:
        assert torch.equal(t0[k], t_side[k]), k                  # the stream does not change a bit
    assert all(torch.equal(g0[n], g_side[n]) for n in g0)
    t1, g1 = run(True, True)
    for k in t0:
        assert t0[k].shape == t1[k].shape == (5,)
        assert torch.allclose(t0[k], t1[k], rtol=1e-5, atol=1e-7), (k, t0[k], t1[k])
    worst = max(float((g0[n] - g1[n]).abs().max() / g0[n].abs().max().clamp(min=1e-12)) for n in g0)
    print(f'fused level sums: worst parameter-gradient deviation {worst:.2e} of the tensor range')
    assert worst < 1e-4
