"""Known-answer vectors held by the reference's OWN tests for pieces of this path (data: inputs and expected outputs),
checked against this package's implementations:

  * corner pooling      -- tests/test_ops/test_corner_pool.py:21-58 of the reference;
  * polygon masks       -- tests/test_masks.py:330-478, 531-546: rescale / resize / crop with their rasterised truth
                           bitmaps (these pin COCO's polygon rasterisation in liblsnet_host.so independently of the
                           reference-built maskapi.so), areas;
  * NMS                 -- tests/test_ops/test_nms.py:18-24 lives in tests/test_oracle.py / test_ops_gpu.py."""
import numpy as np
import pytest
import torch

from lsnet_amd.data import PolygonMasks
from lsnet_amd.ops.corner_pool import CornerPool


def test_corner_pool_known_answers():
    with pytest.raises(AssertionError):
        CornerPool('corner')
    lr = torch.tensor([[[[0, 0, 0, 0, 0], [2, 1, 3, 0, 2], [5, 4, 1, 1, 6], [0, 0, 0, 0, 0], [0, 0, 0, 0, 0]]]])
    tb = torch.tensor([[[[0, 3, 1, 0, 0], [0, 1, 1, 0, 0], [0, 3, 4, 0, 0], [0, 2, 2, 0, 0], [0, 0, 2, 0, 0]]]])
    answers = dict(
        left=(lr, [[0, 0, 0, 0, 0], [3, 3, 3, 2, 2], [6, 6, 6, 6, 6], [0, 0, 0, 0, 0], [0, 0, 0, 0, 0]]),
        right=(lr, [[0, 0, 0, 0, 0], [2, 2, 3, 3, 3], [5, 5, 5, 5, 6], [0, 0, 0, 0, 0], [0, 0, 0, 0, 0]]),
        top=(tb, [[0, 3, 4, 0, 0], [0, 3, 4, 0, 0], [0, 3, 4, 0, 0], [0, 2, 2, 0, 0], [0, 0, 2, 0, 0]]),
        bottom=(tb, [[0, 3, 1, 0, 0], [0, 3, 1, 0, 0], [0, 3, 4, 0, 0], [0, 3, 4, 0, 0], [0, 3, 4, 0, 0]]))
    for mode, (x, want) in answers.items():
        got = CornerPool(mode)(x)
        assert got.type() == x.type() and torch.equal(got, torch.tensor([[want]])), mode


PENTAGON = np.array([1, 1, 3, 1, 4, 3, 2, 4, 1, 3], dtype=np.float64)
TRUTH_10 = np.array([[0, 0, 0, 0, 0, 0, 0, 0, 0, 0], [0, 0, 0, 0, 0, 0, 0, 0, 0, 0], [0, 0, 1, 1, 1, 1, 0, 0, 0, 0],
                     [0, 0, 1, 1, 1, 1, 1, 0, 0, 0], [0, 0, 1, 1, 1, 1, 1, 0, 0, 0], [0, 0, 1, 1, 1, 1, 1, 1, 0, 0],
                     [0, 0, 0, 1, 1, 1, 1, 0, 0, 0], [0, 0, 0, 0, 1, 0, 0, 0, 0, 0], [0, 0, 0, 0, 0, 0, 0, 0, 0, 0],
                     [0, 0, 0, 0, 0, 0, 0, 0, 0, 0]], np.uint8)
TRUTH_6 = np.array([[0, 1, 0, 0, 0, 0], [0, 0, 0, 0, 0, 0], [0, 0, 1, 1, 0, 0], [0, 0, 1, 1, 0, 0], [0, 0, 0, 0, 0, 0],
                    [0, 0, 0, 0, 0, 0]], np.uint8)


def test_polygon_mask_known_answers():
    empty = PolygonMasks([], 28, 28)
    r = empty.rescale((56, 72))
    assert (len(r), r.height, r.width) == (0, 56, 56) and r.to_ndarray().shape == (0, 56, 56)
    r = empty.resize((56, 72))
    assert (len(r), r.height, r.width) == (0, 56, 72) and r.to_ndarray().shape == (0, 56, 72)
    assert empty.flip('horizontal').to_ndarray().shape == (0, 28, 28)
    c = empty.crop(np.array([0, 10, 10, 27]))
    assert (len(c), c.height, c.width) == (0, 17, 10)

    one = PolygonMasks([[PENTAGON.copy()]], 5, 5)
    r = one.rescale((12, 10))
    assert (len(r), r.height, r.width) == (1, 10, 10) and (r.to_ndarray() == TRUTH_10).all()
    assert (one.resize((10, 10)).to_ndarray() == TRUTH_10).all()
    parts = [[np.array([0., 0., 1., 0., 1., 1.]), np.array([1., 1., 2., 1., 2., 2., 1., 2.])]]
    two = PolygonMasks(parts, 3, 3).resize((6, 6))
    assert (two.height, two.width) == (6, 6) and (two.to_ndarray() == TRUTH_6).all()
    both = PolygonMasks([[PENTAGON.copy()], parts[0]], 5, 5).resize((10, 10))
    assert (both.to_ndarray() == np.stack([TRUTH_10, np.pad(TRUTH_6, ((0, 4), (0, 4)), 'constant')])).all()

    crop = PolygonMasks([[np.array([1., 3., 5., 1., 5., 6., 1, 6])]], 7, 7).crop(np.array([0, 0, 3, 4]))
    assert (crop.height, crop.width) == (4, 3)
    assert (crop.to_ndarray() == np.array([[0, 0, 0], [0, 0, 0], [0, 0, 1], [0, 1, 1]])).all()
    with pytest.raises(AssertionError):
        PolygonMasks([[PENTAGON.copy()]], 5, 5).crop(np.array([[0, 0, 1, 1]]))       # a 2-D box

    rng = np.random.RandomState(0)
    three = PolygonMasks([[rng.rand(10) * 28] for _ in range(3)], 28, 28)
    for d in ('horizontal', 'vertical'):
        back = three.flip(d).flip(d)
        assert (three.to_ndarray() == back.to_ndarray()).all() and three.flip(d).to_ndarray().shape == (3, 28, 28)
    padded = three.pad((56, 56))
    assert (padded.height, padded.width) == (56, 56) and padded.to_ndarray().shape == (3, 56, 56)

    # areas: shoelace, summed over the parts of an object (test_masks.py:531-546)
    assert PolygonMasks([], 28, 28).areas.shape == (0,)
    square = PolygonMasks([[np.array([1., 1., 5., 1., 5., 5., 1., 5.])]], 7, 7)
    assert square.areas[0] == 16 and square.areas.shape == (1,)
    assert abs(square.to_ndarray().sum() - 16) <= 9          # the bitmap agrees with the polygon up to its border


def test_dataset_wrappers_index_as_the_reference_expects():
    """The index arithmetic the reference's tests/test_dataset.py:89-127 checks on ConcatDataset / RepeatDataset."""
    from lsnet_amd.data import ConcatDataset, RepeatDataset

    class Stub:
        CLASSES = ('a',)

        def __init__(self, n, seed):
            rng = np.random.RandomState(seed)
            self.cats = [rng.randint(0, 80, k).tolist() for k in rng.randint(1, 20, n)]

        def __len__(self):
            return len(self.cats)

        def __getitem__(self, i):
            return i

        def get_cat_ids(self, i):
            return self.cats[i]
    a, b = Stub(10, 0), Stub(20, 1)
    cat = ConcatDataset([a, b])
    assert cat[5] == 5 and cat[25] == 15 and len(cat) == 30
    assert cat.get_cat_ids(5) == a.cats[5] and cat.get_cat_ids(25) == b.cats[15] and cat.get_cat_ids(-1) == b.cats[19]
    rep = RepeatDataset(a, 10)
    assert rep[5] == 5 and rep[15] == 5 and rep[27] == 7 and len(rep) == 100
    assert rep.get_cat_ids(15) == a.cats[5] and rep.get_cat_ids(27) == a.cats[7]


def test_soft_nms_and_nms_match_known_answers_and_reference_library():
    """The reference's doctest vector for soft NMS (mmdet/ops/nms/nms_wrapper.py:78-87) and its tests/test_ops/
    test_nms.py:86-108 / test_soft_nms.py expectations; where oracle/_ref/nms_ext.so (built from the reference's
    nms_cpu.cpp) exists, random boxes index for index against it, all three decay methods."""
    from lsnet_amd.ops import nms_match, soft_nms
    dets = np.array([[4., 3., 5., 3., 0.9], [4., 3., 5., 4., 0.9], [3., 1., 3., 1., 0.5], [3., 1., 3., 1., 0.5],
                     [3., 1., 3., 1., 0.4], [3., 1., 3., 1., 0.0]], dtype=np.float32)
    new_dets, inds = soft_nms(dets, 0.6, sigma=0.5)
    assert len(inds) == len(new_dets) == 5 and new_dets.dtype == np.float32 and inds.dtype == np.int64
    t_dets, t_inds = soft_nms(torch.from_numpy(dets), 0.6, method='gaussian')
    assert isinstance(t_dets, torch.Tensor) and t_inds.dtype == torch.long and len(t_inds) == len(t_dets)
    with pytest.raises(ValueError):
        soft_nms(dets, 0.6, method='cubic')
    assert nms_match(np.zeros((0, 5), np.float32), 0.5) == []
    from lsnet_amd.ops import nms
    # tests/test_ops/test_nms.py:14-50 (test_nms_device_and_dtypes_cpu): the reference's vector, float32 and float64,
    # ndarray and tensor
    base = np.array([[49.1, 32.4, 51.0, 35.9, 0.1], [49.3, 32.9, 51.0, 35.3, 0.05], [35.3, 11.5, 39.9, 14.5, 0.9],
                     [35.2, 11.7, 39.7, 15.7, 0.3]])
    expected = np.array([[35.3, 11.5, 39.9, 14.5, 0.9], [49.1, 32.4, 51.0, 35.9, 0.1]])
    for dt in (np.float32, np.float64):
        kept, keep_inds = nms(base.astype(dt), 0.6)
        assert kept.dtype == dt and np.array_equal(kept, expected.astype(dt)) and keep_inds.tolist() == [2, 0]
        t_kept, t_inds = nms(torch.from_numpy(base.astype(dt)), 0.6)
        assert t_kept.dtype == torch.from_numpy(base.astype(dt)).dtype and torch.equal(t_kept, torch.from_numpy(expected.astype(dt)))
    e_dets, e_inds = nms(np.zeros((0, 5), np.float32), 0.5)
    assert len(e_dets) == 0 and len(e_inds) == 0
    boxes = np.array([[49.1, 32.4, 51.0, 35.9, 0.9], [49.3, 32.9, 51.0, 35.3, 0.9], [35.3, 11.5, 39.9, 14.5, 0.4],
                      [35.2, 11.7, 39.7, 15.7, 0.3]], dtype=np.float32)       # test_nms.py's boxes
    groups = nms_match(boxes, 0.1)
    assert sorted(int(i) for g in groups for i in g) == [0, 1, 2, 3] and all(len(g) == 2 for g in groups)
    tg = nms_match(torch.from_numpy(boxes), 0.1)
    assert all(isinstance(g, torch.Tensor) and g.dtype == torch.long for g in tg)

    from oracle import build_ref
    ref = build_ref.load() if build_ref.available() else None
    if ref is None:
        return
    rng = np.random.RandomState(0)
    for n in (1, 7, 60, 300):
        xy = rng.rand(n, 2) * 60
        wh = rng.rand(n, 2) * 30 + 1
        d = np.concatenate([xy, xy + wh, rng.rand(n, 1)], 1).astype(np.float32)
        for method, code in (('linear', 1), ('gaussian', 2)):
            want = ref.soft_nms(torch.from_numpy(d), 0.5, code, 0.5, 0.05)
            got_d, got_i = soft_nms(d, 0.5, method=method, sigma=0.5, min_score=0.05)
            assert np.array_equal(want[:, 5].numpy().astype(np.int64), got_i), (n, method)
            np.testing.assert_allclose(got_d, want[:, :5].numpy(), rtol=1e-6, atol=1e-7)
        scores_unique = d.copy()
        scores_unique[:, 4] = rng.permutation(n) / n                          # no ties: the sort order is unambiguous
        want = ref.nms_match(torch.from_numpy(scores_unique), 0.3)
        got = nms_match(scores_unique, 0.3)
        assert [list(map(int, g)) for g in got] == [list(g) for g in want], n
        # hard NMS of host tensors (nms_cpu, the CPU branch of nms_wrapper.py:33-37), float32 and float64
        from lsnet_amd.ops import nms
        for arr in (scores_unique, scores_unique.astype(np.float64)):
            want_keep = ref.nms(torch.from_numpy(arr), 0.4).numpy()
            got_dets, got_keep = nms(arr, 0.4)
            assert got_keep.dtype == np.int64 and np.array_equal(got_keep, want_keep), (n, arr.dtype)
            assert got_dets.dtype == arr.dtype and np.array_equal(got_dets, arr[want_keep])
            t_dets, t_keep = nms(torch.from_numpy(arr), 0.4)
            assert t_dets.dtype == torch.from_numpy(arr).dtype and t_keep.tolist() == want_keep.tolist()
