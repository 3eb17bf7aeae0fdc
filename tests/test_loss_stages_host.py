"""Host-side checks of the round-5 structure of LSHead's loss: the loss computed from target stages prepared ahead
of time (what forward_train does on its second stream) equals the inline loss bit for bit; `_split_px` views remember the
concatenated tensor they come from; per-level lists that are views of one tensor are added up with one operation."""
import torch

from tests import golden_cases as gc
from tests import golden_util as gu


def _head_and_inputs(task):
    dev = torch.device('cpu')
    head = gc.build_head(task, dev, 32)
    head.train()
    feats = [f.to(dev).requires_grad_() for f in gu.head_inputs(11, 32)]
    boxes, labels, extremes, masks, kps, metas = gc.gt_for(task, dev)
    gt = (boxes, extremes if task in ('bbox', 'pose_bbox') else None, [k.clone() for k in kps] if 'pose' in task else None,
          masks if task == 'segm' else None, labels, metas)
    return head, feats, gt


def test_loss_from_prepared_target_stages_equals_the_inline_loss(cpu_oracle_backend):
    for task in ('bbox', 'segm'):
        head, feats, gt = _head_and_inputs(task)
        seen = {}
        outs = head(feats, after_init=lambda preds: seen.update(preds))
        assert sorted(seen) == sorted(head.branches) and all(len(v) == len(feats) for v in seen.values())
        inline = head.loss(*outs, *gt)
        sizes = [tuple(f.shape[-2:]) for f in feats]
        pre = head.init_stage_targets(sizes, feats[0], *gt)
        stage2 = head.refine_stage_targets(pre, [p.detach() for p in seen[head._box_branch()]])
        staged = head.loss(*outs, *gt, init_stage=pre, refine_stage=stage2)
        assert sorted(staged) == sorted(inline)
        for k in inline:
            for a, b in zip(inline[k], staged[k]):
                assert torch.equal(a, b), k
        # stages of another geometry are not used
        wrong = dict(pre, featmap_sizes=[(1, 1)] * len(sizes))
        again = head.loss(*outs, *gt, init_stage=wrong, refine_stage=stage2)
        for k in inline:
            for a, b in zip(inline[k], again[k]):
                assert torch.equal(a, b), k


def test_split_views_remember_their_concatenated_tensor():
    from lsnet_amd.models.dense_heads.ls_head import LSHead
    shapes = [(5, 4), (3, 2), (1, 1)]
    n = sum(h * w for h, w in shapes)
    x = torch.randn(2, n, 6).unsqueeze(2).permute(0, 3, 1, 2).requires_grad_()       # (B, C, N_all, 1), channels-last memory
    maps = LSHead._split_px(x, shapes)
    assert [tuple(m.shape) for m in maps] == [(2, 6, h, w) for h, w in shapes]
    base = LSHead._px_base(maps)
    assert base is not None and base.shape == (2, n, 6)
    assert torch.equal(base, torch.cat([m.permute(0, 2, 3, 1).reshape(2, -1, 6) for m in maps], dim=1))
    assert base.data_ptr() == x.data_ptr()                                            # a view, nothing was copied
    g, = torch.autograd.grad((base * 2).sum(), x)
    assert torch.equal(g, torch.full_like(x, 2.0))
    assert LSHead._px_base(maps[::-1]) is None and LSHead._px_base([m.clone() for m in maps]) is None
    assert LSHead._px_base(maps[:2]) is None
    with torch.no_grad():
        assert LSHead._px_base(LSHead._split_px(x.detach(), shapes)) is not None


def test_level_terms_are_added_with_one_operation():
    from lsnet_amd.models.dense_heads.ls_head import LevelTerms
    from lsnet_amd.models.detectors.base import BaseDetector
    base = torch.tensor([0.5, 1.25, 2.0, 0.125, 3.0], requires_grad=True)
    terms = LevelTerms(base * 1.0)
    assert isinstance(terms, list) and len(terms) == 5 and all(t.dim() == 0 for t in terms)
    det = BaseDetector.__new__(BaseDetector)
    loss, log_vars = det._parse_losses({'loss_a': terms, 'loss_b': [base[0] * 2, base[1] * 2], 'acc': base[2].detach()})
    assert abs(float(log_vars['loss_a']) - 6.875) < 1e-6 and abs(float(log_vars['loss_b']) - 3.5) < 1e-6
    assert abs(float(loss.detach()) - 10.375) < 1e-6 and abs(float(log_vars['loss']) - 10.375) < 1e-6
    loss.backward()
    assert torch.equal(base.grad, torch.tensor([3.0, 3.0, 1.0, 1.0, 1.0]))
