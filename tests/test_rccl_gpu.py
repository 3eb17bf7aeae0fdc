"""The data-parallel path over RCCL (torch.distributed backend 'nccl'), two processes, one GPU each: the hook-driven
gradient-view reducer and the hipGraph step (no collective inside the captured graph; log scalars averaged eagerly).
Self-skipping on boxes with fewer than two GPUs (the 1-GPU boxes of `gpurun`); the same code paths run over gloo in
tests/test_runner_dist.py on every CPU run."""
import os
import tempfile

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from tests.test_runner_dist import Toy, _batches, _free_port

pytestmark = [pytest.mark.gpu,
              pytest.mark.skipif(not torch.cuda.is_available() or torch.cuda.device_count() < 2,
                                 reason='needs two GPUs for a 2-rank RCCL run')]


def _nccl_worker(rank, world, port, out_dir, graphed):
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
    torch.cuda.set_device(rank)
    dist.init_process_group('nccl', rank=rank, world_size=world)
    try:
        from lsnet_amd.parallel import DataParallelModel
        from lsnet_amd.runner import EpochBasedRunner, build_optimizer
        dev = torch.device('cuda', rank)
        torch.manual_seed(100 + rank)
        model = DataParallelModel(Toy().to(dev))
        opt = build_optimizer(model, dict(type='SGD', lr=0.1, momentum=0.9))
        r = EpochBasedRunner(model, optimizer=opt, logger=lambda s: None)
        r.register_training_hooks(dict(policy='step', step=[100]), dict(grad_clip=dict(max_norm=35, norm_type=2)),
                                  None, dict(interval=1000, hooks=[]))
        if graphed:
            r.enable_hip_graph(warmup=1)
        data = [{k: v.to(dev) for k, v in b.items()} for b in _batches(5, 10 + rank)]
        r.run([data], [('train', 1)], 2)
        torch.cuda.synchronize()
        torch.save({k: v.detach().cpu().clone() for k, v in model.module.state_dict().items()},
                   os.path.join(out_dir, f'rank{rank}.pt'))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize('graphed', [False, True], ids=['hooks', 'hipgraph'])
def test_two_rank_rccl_equals_large_batch_sgd(graphed):
    world = 2
    with tempfile.TemporaryDirectory() as d:
        mp.spawn(_nccl_worker, args=(world, _free_port(), d, graphed), nprocs=world, join=True)
        sd = [torch.load(os.path.join(d, f'rank{r}.pt')) for r in range(world)]
    for k in sd[0]:
        assert torch.equal(sd[0][k], sd[1][k]), f'ranks diverged at {k}'
    torch.manual_seed(100)
    ref = Toy()
    opt = torch.optim.SGD(ref.parameters(), lr=0.1, momentum=0.9)
    shards = [_batches(5, 10), _batches(5, 11)]
    for _ in range(2):
        for i in range(5):
            opt.zero_grad()
            for s in shards:
                (ref.train_step(s[i], None)['loss'] / world).backward()
            torch.nn.utils.clip_grad_norm_([p for p in ref.parameters() if p.requires_grad and p.grad is not None], 35)
            opt.step()
    for k, v in ref.state_dict().items():
        assert torch.allclose(sd[0][k], v, rtol=1e-4, atol=1e-5), k
