"""Samplers of the real-data pipeline.  Where /root/reference is present the indices are compared with the reference's
own DistributedGroupSampler; everywhere, with the invariants that define it."""
import os
import types

import numpy as np
import pytest

from lsnet_amd.data.samplers import DistributedGroupSampler


def _dataset(n, seed):
    rng = np.random.RandomState(seed)
    return types.SimpleNamespace(flag=(rng.rand(n) < 0.7).astype(np.uint8), __len__=lambda: n)


@pytest.mark.parametrize('n,spg,world', [(103, 2, 8), (64, 4, 2), (37, 3, 1), (10, 2, 4)])
def test_group_sampler_invariants(n, spg, world):
    ds = _dataset(n, n)
    per_rank = []
    for r in range(world):
        s = DistributedGroupSampler(ds, spg, world, r)
        s.set_epoch(3)
        idx = list(s)
        assert len(idx) == len(s) and len(idx) % spg == 0
        for b in range(0, len(idx), spg):                       # one group per mini-batch
            assert len({int(ds.flag[i]) for i in idx[b:b + spg]}) == 1
        per_rank.append(idx)
    assert len({len(p) for p in per_rank}) == 1                 # same number of batches on every rank
    seen = set(i for p in per_rank for i in p)
    assert seen == set(range(n))                                # every image is used (padding repeats some)
    s = DistributedGroupSampler(ds, spg, world, 0)
    s.set_epoch(4)
    assert list(s) != per_rank[0]                               # another epoch, another order


@pytest.mark.skipif(not os.path.isdir('/root/reference/code'), reason='the reference tree is not on this machine')
def test_group_sampler_equals_reference():
    os.environ['PYTHONDONTWRITEBYTECODE'] = '1'
    from oracle.ref_harness import bootstrap
    bootstrap.load_reference()
    from mmdet.datasets.samplers import DistributedGroupSampler as Ref
    for n, spg, world in ((103, 2, 8), (64, 4, 2), (37, 3, 1)):
        ds = _dataset(n, n)
        for r in range(world):
            a, b = DistributedGroupSampler(ds, spg, world, r), Ref(ds, spg, world, r)
            for ep in (0, 5):
                a.set_epoch(ep)
                b.set_epoch(ep)
                assert list(a) == [int(i) for i in b], (n, spg, world, r, ep)
