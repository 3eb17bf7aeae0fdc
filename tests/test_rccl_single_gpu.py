"""RCCL on a 1-GPU box: a one-rank 'nccl' process group with LSNET_FORCE_COLLECTIVES=1 sends every gradient bucket of the
hook-driven reducer through RCCL (async work objects on RCCL's stream, ordered against the kernels that fill the
buckets, waited for in finish()) -- the part of the data-parallel path that tests/test_rccl_gpu.py can only run with two
GPUs.  The gloo twin of this test runs on every CPU run (tests/test_runner_dist.py).

Opt-in (LSNET_RCCL_SINGLE=1) until it has been seen green on the pool's boxes: the driver runs the suite with -x."""
import os

import pytest
import torch

from tests.test_runner_dist import one_rank_forced_collectives_case

pytestmark = [pytest.mark.gpu,
              pytest.mark.skipif(not torch.cuda.is_available(), reason='needs a GPU'),
              pytest.mark.skipif(os.environ.get('LSNET_RCCL_SINGLE') != '1', reason='opt-in: LSNET_RCCL_SINGLE=1')]


def test_one_rank_rccl_group_runs_every_bucket():
    one_rank_forced_collectives_case('nccl', 'cuda:0')
