"""RCCL on a 1-GPU box: a one-rank 'nccl' process group with LSNET_FORCE_COLLECTIVES=1 sends every gradient bucket of the
hook-driven reducer through RCCL (async work objects on RCCL's stream, ordered against the kernels that fill the
buckets, waited for in finish()) -- the part of the data-parallel path that tests/test_rccl_gpu.py can only run with two
GPUs (mmcv/parallel/distributed.py:10-53, mmdet/apis/train.py:74-78).  The gloo twin runs on every CPU run
(tests/test_runner_dist.py).  Seen green on the pool's boxes in round 4 (profiles/r4_rccl_single.log): on by default."""
import os
import tempfile

import pytest
import torch
import torch.multiprocessing as mp

from tests.test_runner_dist import _free_port, one_rank_forced_collectives_case

pytestmark = [pytest.mark.gpu, pytest.mark.skipif(not torch.cuda.is_available(), reason='needs a GPU')]


def test_one_rank_rccl_group_runs_every_bucket():
    one_rank_forced_collectives_case('nccl', 'cuda:0')


def _detector_worker(_, collective, port, out_path):
    """Two training iterations of the real LSNet R-50 bbox detector (gradient sinks, fused SGD, the runner's own hooks)."""
    import torch.distributed as dist
    os.environ['HSA_ENABLE_IPC_MODE_LEGACY'] = '0'
    if collective:
        os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK='0', WORLD_SIZE='1',
                          LSNET_FORCE_COLLECTIVES='1')
        torch.cuda.set_device(0)
        dist.init_process_group('nccl', rank=0, world_size=1)
    else:
        os.environ.pop('LSNET_FORCE_COLLECTIVES', None)
    try:
        from lsnet_amd.data import synthetic_batch
        from lsnet_amd.model_zoo import build_lsnet
        from lsnet_amd.parallel import DataParallelModel
        from lsnet_amd.runner import EpochBasedRunner, build_optimizer
        dev = torch.device('cuda:0')
        torch.manual_seed(0)
        model, cfg = build_lsnet('bbox', 'r50')
        model = DataParallelModel(model.to(dev).to(memory_format=torch.channels_last).train(), bucket_mb=25.0)
        assert model.reducer.collective == bool(collective)
        opt = build_optimizer(model, cfg.optimizer)
        runner = EpochBasedRunner(model, optimizer=opt, logger=lambda m: None)
        runner.register_training_hooks(cfg.lr_config, cfg.optimizer_config, None, dict(interval=10 ** 9, hooks=[]))
        runner.epoch_len = 10 ** 9
        runner.call_hook('before_run')
        runner.call_hook('before_train_epoch')
        data = synthetic_batch('bbox', 2, 384, 480, boxes_per_img=5, num_classes=80, seed=7, device=dev, channels_last=True)
        losses = []
        for _ in range(2):
            runner.call_hook('before_train_iter')
            runner.outputs = runner.run_iter(data)
            runner.call_hook('after_train_iter')
            runner.iter += 1
            losses.append(float(runner.outputs['loss']))
        torch.cuda.synchronize()
        nb = len(model.reducer.buckets)
        torch.save({'state': {k: v.detach().cpu() for k, v in model.module.state_dict().items()}, 'losses': losses,
                    'buckets': nb}, out_path)
    finally:
        if collective:
            dist.destroy_process_group()


def test_real_detector_through_one_rank_rccl_equals_plain_step_bit_for_bit():
    """The all-reduce of one rank is the identity: two iterations of the real detector with every bucket sent through RCCL
    must leave exactly the parameters of the same two iterations without a process group -- any mis-ordering between the
    kernels that fill a bucket (weight-gradient sinks), its all-reduce and the optimizer would show."""
    with tempfile.TemporaryDirectory() as d:
        a, b = os.path.join(d, 'rccl.pt'), os.path.join(d, 'plain.pt')
        mp.spawn(_detector_worker, args=(True, _free_port(), a), nprocs=1, join=True)
        mp.spawn(_detector_worker, args=(False, 0, b), nprocs=1, join=True)
        ra, rb = torch.load(a), torch.load(b)
    assert ra['buckets'] >= 4
    assert ra['losses'] == rb['losses'], (ra['losses'], rb['losses'])
    diff = [k for k in rb['state'] if not torch.equal(ra['state'][k], rb['state'][k])]
    assert not diff, f'{len(diff)} tensors differ, e.g. {diff[:4]}'


def test_bench_two_ranks_share_the_gpu():
    """`bench.py --gpus 2` as the driver launches it (torch.distributed.run, one process per rank), on a 1-GPU box through
    LSNET_BENCH_BACKEND=gloo (the ranks share the device, the buckets travel through the host): every rank must issue the
    SAME sequence of collectives.  Round 4 added a rank-0-only pass whose loss bookkeeping all-reduced -- the 2-rank run
    aborted with gloo's size-mismatch error, and only this self-test saw it (the driver's 8-GPU run would have been next)."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, LSNET_BENCH_BACKEND='gloo', HSA_ENABLE_IPC_MODE_LEGACY='0')
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '2', '--master-addr', '127.0.0.1',
           '--master-port', str(_free_port()), 'bench.py', '--gpus', '2', '--steps', '2', '--warmup', '1', '--no-extra',
           '--no-cpu-baseline', '--height', '384', '--width', '480']
    res = subprocess.run(cmd, cwd=root, env=env, capture_output=True, text=True, timeout=900)
    assert res.returncode == 0, res.stdout[-2000:] + res.stderr[-4000:]
    line = json.loads(res.stdout.strip().splitlines()[-1])
    assert line['n_gpus'] == 2 and line['steps'] == 2 and line['value'] > 0 and line['scaling'] == 'weak'
    assert line['config']['world_size'] == 2


def _timing_worker(_, port, out_path):
    """The benchmark step (2 x 3 x 800 x 1344) of a process that builds its model the way a data-parallel rank does, timed
    with and without the bucket all-reduces of every step (toggled on the one reducer, alternating)."""
    import sys
    import torch.distributed as dist
    os.environ['HSA_ENABLE_IPC_MODE_LEGACY'] = '0'
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK='0', WORLD_SIZE='1', LSNET_FORCE_COLLECTIVES='1')
    torch.cuda.set_device(0)
    dist.init_process_group('nccl', rank=0, world_size=1)
    try:
        root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
        sys.path.insert(0, root)
        sys.argv = sys.argv[:1]
        import bench
        from lsnet_amd.data import synthetic_batch
        from lsnet_amd.model_zoo import build_lsnet
        from lsnet_amd.parallel import DataParallelModel
        dev = torch.device('cuda:0')
        torch.manual_seed(0)
        model, cfg = build_lsnet('bbox', 'r50')
        model = DataParallelModel(model.to(dev).to(memory_format=torch.channels_last).train())
        assert model.reducer.collective
        step, _ = bench.build_step(model, cfg)
        data = synthetic_batch('bbox', 2, 800, 1344, seed=1234, device=dev, channels_last=True)
        bench.timed_steps(step, data, 1, 4)
        with_c, plain = [], []
        for _ in range(3):
            model.reducer.collective = True
            with_c.append(bench.timed_steps(step, data, 5, 1))
            model.reducer.collective = False
            plain.append(bench.timed_steps(step, data, 5, 1))
        torch.save({'with': min(with_c) * 1e3, 'plain': min(plain) * 1e3}, out_path)
    finally:
        dist.destroy_process_group()


def test_collectives_cost_the_step_a_few_percent_not_a_hardware_queue():
    """Round 6: with RCCL's streams alive the order in which the process's streams first submit work decides
    whether the step's stream shares a hardware queue -- eight of ten orders ran the benchmark step at 49 ms instead of 31.5
    (profiles/r6_rccl_streams.txt), among them the one DataParallelModel arranged until round 5.  The package's order (second
    stream submits at creation, library stream warmed, then the first collective) must keep the step with the seven bucket
    all-reduces within 8 % of the same step without them (measured 2 %; a shared queue costs 55 %)."""
    with tempfile.TemporaryDirectory() as d:
        path = os.path.join(d, 't.pt')
        mp.spawn(_timing_worker, args=(_free_port(), path), nprocs=1, join=True)
        r = torch.load(path)
    print(f"step with the bucket all-reduces {r['with']:.2f} ms, without {r['plain']:.2f} ms")
    assert r['with'] <= 1.08 * r['plain'], r
