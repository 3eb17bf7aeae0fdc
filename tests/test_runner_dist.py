"""Runner / hooks (lr schedule, optimizer hook, checkpoints) and the data-parallel gradient reducer.

The multi-process tests run world_size 2 over gloo on CPU: the reducer is backend-agnostic, on the GPU box
the same code runs over RCCL (backend 'nccl')."""
import os
import socket
import tempfile

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp
import torch.nn as nn

from lsnet_amd.parallel import BucketedGradReducer, DataParallelModel
from lsnet_amd.runner import EpochBasedRunner, build_optimizer


class Toy(nn.Module):
    """A model with the detector protocol: train_step(data, optimizer) -> dict(loss, log_vars, num_samples)."""

    def __init__(self):
        super().__init__()
        self.a = nn.Linear(8, 16)
        self.b = nn.Linear(16, 4)
        self.unused = nn.Parameter(torch.zeros(3))     # never receives a gradient
        self.frozen = nn.Parameter(torch.ones(2), requires_grad=False)

    def forward(self, x):
        return self.b(torch.relu(self.a(x)))

    def train_step(self, data, optimizer):
        loss = (self(data['x']) - data['y']).pow(2).mean()
        return dict(loss=loss, log_vars=dict(loss=loss.detach()), num_samples=data['x'].shape[0])


def _batches(n, seed):
    g = torch.Generator().manual_seed(seed)
    return [dict(x=torch.randn(4, 8, generator=g), y=torch.randn(4, 4, generator=g)) for _ in range(n)]


def test_step_lr_warmup_and_steps():
    m = Toy()
    opt = build_optimizer(m, dict(type='SGD', lr=0.01, momentum=0.9, weight_decay=1e-4))
    r = EpochBasedRunner(m, optimizer=opt, logger=lambda s: None)
    r.register_training_hooks(dict(policy='step', warmup='linear', warmup_iters=10, warmup_ratio=0.001, step=[2, 3]),
                              dict(grad_clip=dict(max_norm=35, norm_type=2)), None, dict(interval=1000, hooks=[]))
    lrs = []

    class Spy(type(r.hooks[0]).__mro__[1]):   # Hook
        priority = 90

        def before_train_iter(self, runner):
            lrs.append(runner.optimizer.param_groups[0]['lr'])
    r.register_hook(Spy())
    r.run([_batches(6, 0)], [('train', 1)], 4)
    # reference rule (mmcv lr_updater.py:153-181): during warm-up lr = regular * (1 - (1 - it/warmup_iters) * (1 - ratio))
    for it in range(10):
        k = (1 - it / 10) * (1 - 0.001)
        assert lrs[it] == pytest.approx(0.01 * (1 - k), rel=1e-6), it
    assert lrs[10] == pytest.approx(0.01) and lrs[11] == pytest.approx(0.01)
    assert lrs[12] == pytest.approx(0.001) and lrs[17] == pytest.approx(0.001)      # epoch 2
    assert lrs[18] == pytest.approx(0.0001)                                          # epoch 3
    assert r.epoch == 4 and r.iter == 24


def test_training_reduces_loss_and_checkpoint_resume():
    torch.manual_seed(0)
    with tempfile.TemporaryDirectory() as d:
        m = Toy()
        opt = build_optimizer(m, dict(type='SGD', lr=0.05, momentum=0.9))
        r = EpochBasedRunner(m, optimizer=opt, work_dir=d, logger=lambda s: None)
        r.register_training_hooks(dict(policy='step', step=[100]), dict(grad_clip=None), dict(interval=1),
                                  dict(interval=1000, hooks=[]))
        data = _batches(8, 1)
        first = float(m.train_step(data[0], None)['loss'])
        r.run([data], [('train', 1)], 3)
        assert float(m.train_step(data[0], None)['loss']) < first
        assert os.path.exists(os.path.join(d, 'epoch_3.pth')) and os.path.islink(os.path.join(d, 'latest.pth'))
        m2 = Toy()
        opt2 = build_optimizer(m2, dict(type='SGD', lr=0.05, momentum=0.9))
        r2 = EpochBasedRunner(m2, optimizer=opt2, work_dir=d, logger=lambda s: None)
        r2.resume(os.path.join(d, 'latest.pth'))
        assert r2.epoch == 3 and r2.iter == 24
        for (k, a), (_, b) in zip(m.state_dict().items(), m2.state_dict().items()):
            assert torch.equal(a, b), k
        assert opt2.state_dict()['state'].keys() == opt.state_dict()['state'].keys()


def test_reducer_is_a_noop_without_process_group():
    m = Toy()
    red = BucketedGradReducer(m.parameters())
    m.train_step(_batches(1, 0)[0], None)['loss'].backward()
    g = m.a.weight.grad.clone()
    red.finish()
    assert torch.equal(m.a.weight.grad, g)


def test_reducer_clears_stale_slot_of_an_unused_parameter():
    """A parameter that gets no gradient in a step must hold zeros afterwards even when the caller used
    `p.grad = None` instead of the reducer's zero_grad() (round-2 advisor finding: the slot kept last step's value)."""
    from lsnet_amd.parallel.reducer import BucketedGradReducer
    torch.manual_seed(0)
    a, b = torch.nn.Linear(3, 3), torch.nn.Linear(3, 3)
    red = BucketedGradReducer(list(a.parameters()) + list(b.parameters()))
    x = torch.randn(4, 3)
    (a(x).sum() + b(x).sum()).backward()
    red.finish()
    assert float(b.weight.grad.abs().sum()) > 0
    for p in list(a.parameters()) + list(b.parameters()):
        p.grad = None                        # the old protocol: no zero_grad() of the reducer
    a(x).sum().backward()                    # b is unused in this step
    red.finish()
    assert float(b.weight.grad.abs().sum()) == 0 and float(b.bias.grad.abs().sum()) == 0
    assert torch.allclose(a.weight.grad, x.sum(0).expand(3, 3))


def test_gradient_sinks_accumulate_into_the_buckets():
    """ops/grad_sink.py protocol on the reducer: an operator that adds its parameter gradient straight into the sink
    and reports `done` is equivalent to returning the gradient to autograd; classic and sunk contributions mix."""
    from lsnet_amd.ops import grad_sink
    from lsnet_amd.parallel.reducer import BucketedGradReducer

    class SunkLinear(torch.autograd.Function):
        @staticmethod
        def forward(ctx, x, w):
            ctx.save_for_backward(x, w)
            return x @ w.t()

        @staticmethod
        def backward(ctx, g):
            x, w = ctx.saved_tensors
            gw = g.t() @ x
            s = grad_sink.sink(w)
            if s is not None:
                s.add_(gw)
                grad_sink.done(w)
                gw = None
            return g @ w, gw

    torch.manual_seed(1)
    w = torch.nn.Parameter(torch.randn(3, 3))
    x = torch.randn(5, 3)
    ref = torch.autograd.grad((x @ w.t()).sum() * 2 + (x @ w.t()).pow(2).sum(), w)[0]
    red = BucketedGradReducer([w])
    assert grad_sink.sink(w) is None         # nothing registered before zero_grad()
    for _ in range(3):                       # (the reducer learns the contribution count in the first step)
        red.zero_grad()
        assert grad_sink.sink(w) is w.grad
        # two sunk uses and one classic use of the same parameter
        (SunkLinear.apply(x, w).sum() + SunkLinear.apply(x, w).sum() + (x @ w.t()).pow(2).sum()).backward()
        red.finish()
        assert torch.allclose(w.grad, ref, atol=1e-5)
    w.grad = None                            # user replaced the gradient: the sink is off, the classic path works
    assert grad_sink.sink(w) is None
    SunkLinear.apply(x, w).sum().backward()
    assert torch.allclose(w.grad, torch.ones(3, 5) @ x, atol=1e-6)


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, out_dir, bucket_mb):
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        torch.manual_seed(100 + rank)           # different initial weights per rank: broadcast must fix that
        model = DataParallelModel(Toy(), bucket_mb=bucket_mb)
        opt = build_optimizer(model, dict(type='SGD', lr=0.1, momentum=0.9))
        r = EpochBasedRunner(model, optimizer=opt, logger=lambda s: None)
        r.register_training_hooks(dict(policy='step', step=[100]), dict(grad_clip=dict(max_norm=35, norm_type=2)),
                                  None, dict(interval=1000, hooks=[]))
        r.run([_batches(5, 10 + rank)], [('train', 1)], 2)        # each rank sees its own shard
        torch.save({k: v.clone() for k, v in model.module.state_dict().items()},
                   os.path.join(out_dir, f'rank{rank}.pt'))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize('bucket_mb', [64.0, 0.0002])   # one bucket / one bucket per few parameters
def test_two_rank_data_parallel_equals_large_batch_sgd(bucket_mb):
    world, port = 2, _free_port()
    with tempfile.TemporaryDirectory() as d:
        mp.spawn(_worker, args=(world, port, d, bucket_mb), nprocs=world, join=True)
        sd = [torch.load(os.path.join(d, f'rank{r}.pt')) for r in range(world)]
    for k in sd[0]:
        assert torch.equal(sd[0][k], sd[1][k]), f'ranks diverged at {k}'
    # single-process restatement: mean of the two per-rank gradients, same clip and SGD
    torch.manual_seed(100)                       # rank 0's weights are broadcast to everyone
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
        assert torch.allclose(sd[0][k], v, rtol=1e-5, atol=1e-6), k
    assert torch.equal(sd[0]['unused'], torch.zeros(3))


def _one_rank_worker(rank, world, port, out_dir, backend, device):
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK='0', WORLD_SIZE='1', LOCAL_RANK='0',
                      LSNET_FORCE_COLLECTIVES='1')
    os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
    if device != 'cpu':
        torch.cuda.set_device(0)
    dist.init_process_group(backend, rank=0, world_size=1)
    try:
        torch.manual_seed(100)
        model = DataParallelModel(Toy().to(device), bucket_mb=0.0002)
        assert model.reducer.collective and model.reducer.world == 1
        opt = build_optimizer(model, dict(type='SGD', lr=0.1, momentum=0.9))
        r = EpochBasedRunner(model, optimizer=opt, logger=lambda s: None)
        r.register_training_hooks(dict(policy='step', step=[100]), dict(grad_clip=dict(max_norm=35, norm_type=2)),
                                  None, dict(interval=1000, hooks=[]))
        launched = []
        orig = dist.all_reduce

        def counting(*a, **k):
            launched.append(1)
            return orig(*a, **k)
        dist.all_reduce = counting
        try:
            r.run([[{k: v.to(device) for k, v in b.items()} for b in _batches(5, 10)]], [('train', 1)], 2)
        finally:
            dist.all_reduce = orig
        assert len(launched) >= 10 * len(model.reducer.buckets), len(launched)   # every bucket of every step went out
        torch.save({k: v.detach().cpu().clone() for k, v in model.module.state_dict().items()},
                   os.path.join(out_dir, 'rank0.pt'))
    finally:
        dist.destroy_process_group()


def one_rank_forced_collectives_case(backend, device):
    """LSNET_FORCE_COLLECTIVES=1: a one-rank group still all-reduces every bucket (hook-driven launches during backward,
    the wait in finish()); the training run equals plain SGD.  With backend 'nccl' this is an RCCL execution of the
    reducer on a 1-GPU box (tests/test_rccl_gpu.py)."""
    with tempfile.TemporaryDirectory() as d:
        mp.spawn(_one_rank_worker, args=(1, _free_port(), d, backend, device), nprocs=1, join=True)
        sd = torch.load(os.path.join(d, 'rank0.pt'))
    torch.manual_seed(100)
    ref = Toy()
    opt = torch.optim.SGD(ref.parameters(), lr=0.1, momentum=0.9)
    data = _batches(5, 10)
    for _ in range(2):
        for i in range(5):
            opt.zero_grad()
            ref.train_step(data[i], None)['loss'].backward()
            torch.nn.utils.clip_grad_norm_([p for p in ref.parameters() if p.requires_grad and p.grad is not None], 35)
            opt.step()
    for k, v in ref.state_dict().items():
        assert torch.allclose(sd[k], v, rtol=1e-4, atol=1e-5), k


def test_one_rank_group_with_forced_collectives_equals_plain_sgd():
    one_rank_forced_collectives_case('gloo', 'cpu')


# ---------------------------------------------------------------------------------------------------
# GraphedForwardBackward: without a GPU it runs its eager path -- gradient-view buckets, in-place accumulation,
# bucket all-reduce -- which is everything except the hipGraph capture itself (covered by the gpu tests).
def test_graph_step_matches_plain_training_on_cpu():
    torch.manual_seed(3)
    ref, m = Toy(), Toy()
    m.load_state_dict(ref.state_dict())
    data = _batches(6, 5)
    runs = []
    for model, graphed in ((ref, False), (m, True)):
        opt = build_optimizer(model, dict(type='SGD', lr=0.05, momentum=0.9, weight_decay=1e-3))
        r = EpochBasedRunner(model, optimizer=opt, logger=lambda s: None)
        r.register_training_hooks(dict(policy='step', step=[100]), dict(grad_clip=dict(max_norm=1.0, norm_type=2)),
                                  None, dict(interval=1000, hooks=[]))
        if graphed:
            gs = r.enable_hip_graph(warmup=1, bucket_mb=0.0002)
            assert len(gs.buckets) > 1
        r.run([data], [('train', 1)], 2)
        runs.append(model.state_dict())
    for k in runs[0]:
        assert torch.allclose(runs[0][k], runs[1][k], rtol=1e-6, atol=1e-7), k


def _graph_worker(rank, world, port, out_dir):
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        torch.manual_seed(100 + rank)
        model = DataParallelModel(Toy())
        opt = build_optimizer(model, dict(type='SGD', lr=0.1, momentum=0.9))
        r = EpochBasedRunner(model, optimizer=opt, logger=lambda s: None)
        r.register_training_hooks(dict(policy='step', step=[100]), dict(grad_clip=dict(max_norm=35, norm_type=2)),
                                  None, dict(interval=1000, hooks=[]))
        r.enable_hip_graph(warmup=1)
        r.run([_batches(5, 10 + rank)], [('train', 1)], 2)
        torch.save({k: v.clone() for k, v in model.module.state_dict().items()},
                   os.path.join(out_dir, f'rank{rank}.pt'))
    finally:
        dist.destroy_process_group()


def test_two_rank_graph_step_equals_hooked_reducer():
    """The graph step's bucket all-reduce gives the same training as the hook-driven reducer."""
    world = 2
    with tempfile.TemporaryDirectory() as d1, tempfile.TemporaryDirectory() as d2:
        mp.spawn(_graph_worker, args=(world, _free_port(), d1), nprocs=world, join=True)
        mp.spawn(_worker, args=(world, _free_port(), d2, 64.0), nprocs=world, join=True)
        a = [torch.load(os.path.join(d1, f'rank{r}.pt')) for r in range(world)]
        b = torch.load(os.path.join(d2, 'rank0.pt'))
    for k in b:
        assert torch.equal(a[0][k], a[1][k]), k
        assert torch.allclose(a[0][k], b[k], rtol=1e-5, atol=1e-6), k


# ---------------------------------------------------------------------------------------------------
# the real detector under two ranks (gloo, CPU oracle behind the native ops): fused log-scalar all-reduce,
# hook-driven gradient buckets, identical parameters on both ranks after the step
def _lsnet_worker(rank, world, port, out_dir):
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    torch.set_num_threads(2)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        from lsnet_amd.data import synthetic_batch
        from lsnet_amd.model_zoo import build_lsnet
        from lsnet_amd.ops import register_backend
        from tests.oracle_backend import OracleBackend
        register_backend('cpu', OracleBackend())
        torch.manual_seed(7 + rank)
        model, cfg = build_lsnet('bbox', 'r50')
        model = DataParallelModel(model.train())
        opt = build_optimizer(model, cfg.optimizer)
        r = EpochBasedRunner(model, optimizer=opt, logger=lambda s: None)
        r.register_training_hooks(cfg.lr_config, cfg.optimizer_config, None, dict(interval=10 ** 9, hooks=[]))
        batch = synthetic_batch('bbox', 1, 288, 352, boxes_per_img=3, seed=50 + rank, device='cpu', channels_last=False)
        r.run([[batch]], [('train', 1)], 1)
        logs = r.outputs['log_vars']
        torch.save(dict(loss=float(logs['loss']), w=model.module.bbox_head.pts_cls_out.weight.detach().clone(),
                        b=model.module.backbone.layer4[0].conv1.weight.detach().clone()),
                   os.path.join(out_dir, f'rank{rank}.pt'))
    finally:
        dist.destroy_process_group()


def test_two_rank_lsnet_step():
    world = 2
    with tempfile.TemporaryDirectory() as d:
        mp.spawn(_lsnet_worker, args=(world, _free_port(), d), nprocs=world, join=True)
        a, b = (torch.load(os.path.join(d, f'rank{r}.pt')) for r in range(world))
    assert a['loss'] == b['loss'] and a['loss'] > 0          # the logged loss is the mean over ranks
    assert torch.equal(a['w'], b['w']) and torch.equal(a['b'], b['b'])
