"""Multi-scale vote testing: the per-class merge against the reference's `instances_vote` (fixture
tests/golden/vote.npz from oracle/ref_harness/make_golden.py) and the mapping-back / end-to-end plumbing."""
import os

import numpy as np
import torch

from lsnet_amd.core.vote import instance_mapping_back, instances_vote, remove_boxes, vote_merge

REF = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'vote.npz'))


def test_instances_vote_matches_reference():
    case = 0
    while f'{case}/boxes' in REF.files:
        b, v, s = (torch.from_numpy(REF[f'{case}/{k}']) for k in ('boxes', 'vectors', 'scores'))
        ob, ov, os_ = instances_vote(b, v, s)
        assert ob.shape[0] == REF[f'{case}/out_boxes'].shape[0], case
        assert np.allclose(ob.numpy(), REF[f'{case}/out_boxes'], rtol=1e-6, atol=1e-4), case
        assert np.allclose(ov.numpy(), REF[f'{case}/out_vectors'], rtol=1e-6, atol=1e-4), case
        assert np.allclose(os_.numpy(), REF[f'{case}/out_scores'], rtol=1e-6, atol=1e-6), case
        assert (os_[:-1] >= os_[1:]).all()
        case += 1
    assert case == 4
    assert REF['2/out_boxes'].shape[0] == 0          # a single detection votes to nothing (reference quirk)


def test_mapping_back_and_merge():
    shape, sf = (400, 600, 3), np.array([2.0, 2.0, 2.0, 2.0], dtype=np.float32)
    box = torch.tensor([[100., 80., 300., 240.]])
    ext = torch.tensor([[200., 80., 100., 160., 200., 240., 300., 160.]])       # top, left, bottom, right
    b, v = instance_mapping_back(box, ext, shape, sf, flip=False, task='bbox')
    assert torch.allclose(b, box / 2) and torch.allclose(v, ext / 2)
    # the flipped view of the same object maps back onto it
    fbox = torch.tensor([[300., 80., 500., 240.]])
    fext = torch.tensor([[400., 80., 300., 160., 400., 240., 500., 160.]])
    b2, v2 = instance_mapping_back(fbox, fext, shape, sf, flip=True, task='bbox')
    assert torch.allclose(b2, b) and torch.allclose(v2, v)
    assert remove_boxes(torch.tensor([[0., 0., 10., 10.], [0., 0., 100., 100.]]), 32, 1000).tolist() == [1]

    metas = [[dict(img_shape=shape, scale_factor=sf, flip=False)], [dict(img_shape=shape, scale_factor=sf, flip=True)]]
    dets = [torch.cat([box, torch.tensor([[0.9]])], 1), torch.cat([fbox, torch.tensor([[0.7]])], 1)]
    ob, ov, ol = vote_merge(dets, [ext, fext], [torch.tensor([3]), torch.tensor([3])], metas, 'bbox', 80, 4)
    assert ol.tolist()[0] == 3 and torch.allclose(ob[0, :4], box[0] / 2, atol=1e-4) and abs(float(ob[0, 4]) - 0.9) < 1e-6
    assert torch.allclose(ov[0], ext[0] / 2, atol=1e-4)
    none = vote_merge([dets[0][:0]], [ext[:0]], [torch.zeros(0, dtype=torch.long)], metas[:1], 'bbox', 80, 4)
    assert none[0].shape == (0, 5) and none[1].shape == (0, 8)


def test_detector_aug_test_vote_runs(cpu_oracle_backend):
    """forward_test with two views (plain + flipped) goes through LSDetector.aug_test (plumbing; random weights)."""
    from lsnet_amd.model_zoo import build_lsnet
    torch.manual_seed(0)
    model, cfg = build_lsnet('bbox', 'r50')
    model.eval()
    model.test_cfg = cfg.test_cfg
    model.test_cfg.update(method='vote', scale_ranges=[[0, 10000]], score_thr=0.0, nms_pre=20,
                          max_per_img=20)
    model.bbox_head.test_cfg = model.test_cfg
    imgs, metas = [], []
    for (h, w), s in (((288, 352), 1.0),):
        for flip in (False, True):
            imgs.append(torch.randn(1, 3, h, w))
            metas.append([dict(img_shape=(h, w, 3), pad_shape=(h, w, 3), ori_shape=(288, 352, 3), flip=flip,
                               scale_factor=np.array([s, s, s, s], dtype=np.float32))])
    with torch.no_grad():
        boxes, vectors = model(imgs, metas, return_loss=False, rescale=True)
    assert len(boxes) == len(vectors) == 80
    n = sum(b.shape[0] for b in boxes)
    assert n > 0 and all(b.shape[1] == 5 for b in boxes) and all(v.shape[1] == 8 for v in vectors)
    model.test_cfg.method = 'simple'                       # all views merged by ONE NMS: per-class boxes only
    with torch.no_grad():
        merged = model(imgs, metas, return_loss=False, rescale=True)
    assert len(merged) == 80 and all(b.shape[1] == 5 for b in merged) and 0 < sum(len(b) for b in merged) <= 20
