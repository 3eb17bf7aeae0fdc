"""GT formatting of the real-data pipeline against fixtures produced by the reference's own functions
(oracle/ref_harness/make_golden.py: golden_gt_formats)."""
import os

import numpy as np
import torch

from lsnet_amd.data.gt_formats import (flip_extremes, flip_keypoints, flip_polygons, polygon_landmarks,
                                       resample_polygon)

REF = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'gt_formats.npz'))


def test_resample_polygon_matches_reference():
    keys = sorted({k.split('/')[1] for k in REF.files if k.startswith('resample/')})
    assert len(keys) == 5
    for k in keys:
        n = int(k.split('_')[1])
        got = resample_polygon(REF[f'resample/{k}/in'], n)
        want = REF[f'resample/{k}/out']
        assert got.shape == want.shape == (n, 2), k
        assert np.allclose(got, want, rtol=0, atol=1e-9), k


def test_polygon_landmarks_match_reference():
    i = 0
    while f'poly/{i}/n' in REF.files:
        comps = [REF[f'poly/{i}/in{j}'] for j in range(int(REF[f'poly/{i}/n']))]
        got = polygon_landmarks([c.reshape(-1).tolist() for c in comps], REF[f'poly/{i}/bbox'])
        want = REF[f'poly/{i}/out']
        assert len(got) == want.shape[0], i
        for g, w in zip(got, want):
            assert g.shape == (72,)
            assert np.allclose(g, w, rtol=0, atol=1e-9), i
            pts = g.reshape(-1, 2)       # clockwise in image coordinates (= negative shoelace area in a y-up frame)
            assert np.dot(pts[:, 0], np.roll(pts[:, 1], -1)) - np.dot(pts[:, 1], np.roll(pts[:, 0], -1)) <= 0
        i += 1
    assert i == 7


def test_flips_match_reference():
    shape = (480, 640, 3)
    ext, pol, kps = (torch.from_numpy(REF[k]) for k in ('ext', 'pol', 'kps'))
    for d in ('horizontal', 'vertical'):
        assert torch.equal(flip_extremes(ext, shape, d), torch.from_numpy(REF[f'flip/{d}/ext']))
        assert torch.equal(flip_polygons(pol, shape, d), torch.from_numpy(REF[f'flip/{d}/pol']))
        assert torch.equal(flip_keypoints(kps, shape, d), torch.from_numpy(REF[f'flip/{d}/kps']))
        # an involution on extremes and keypoints
        assert torch.allclose(flip_extremes(flip_extremes(ext, shape, d), shape, d), ext, atol=1e-4)
        assert torch.allclose(flip_keypoints(flip_keypoints(kps, shape, d), shape, d), kps, atol=1e-4)
    assert flip_polygons(pol[:0], shape).shape == (0, 72)
