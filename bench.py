#!/usr/bin/env python
"""Throughput of the LSNet training hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

One "step" = one full training iteration of LSNet R-50-FPN (bbox task, conv_module_type='dcn') on
one synthetic COCO-shaped batch of 2 images 3x800x1344 per GPU (BASELINE.json configs[1]):
backbone + FPN + LSHead forward, target assignment (CentroidAssigner + ATSS), focal + cross-IOU
losses, backward, RCCL gradient all-reduce (N > 1), clip-grad-norm 35 and the SGD update -- the same
hooks a real run uses (lsnet_amd/runner).  Inputs are resident in HBM before the timed region.

Prints ONE JSON line (rank 0):  value = total images / s over all N GPUs (weak scaling, 2 img/GPU).
  roofline     : the hand-written kernel family with the most GPU time in the timed steps (a deformable-convolution
                 kernel): algorithmic FLOPs of its launches / their HIP-event time (events recorded by the library
                 around each kernel on its launch stream), against the MFMA peak of its arithmetic -- 2516 / 3 TFLOP/s
                 for split-bf16 products, 157.3 for exact fp32 (MI355X_MICROARCH.md); `traffic` from rocprofv3 counters.
  fp32_exact   : the same step with --math fp32 (exact fp32 MFMA kernels, MIOpen fp32 convolutions), 5 steps.
  cpu_baseline : the same training step on the host CPU (this repo's host code with the CPU oracle
                 standing behind the native ops -- the reference has no CPU path for them), on a
                 bounded sample, rank 0 at N=1 only.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

os.environ.setdefault('DEBUG_CLR_GRAPH_PACKET_CAPTURE', '0')   # ROCm 7.2 hipGraph workaround, see lsnet_amd/__init__.py
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

# HBM traffic per launch (GB) from the committed PMC passes, profiles/r1c_pmc_hbm.txt: mean over the 8 launches
# of a step of FETCH_SIZE (KB; doubled as MI355X_MICROARCH.md "HBM" prescribes for gfx950) + WRITE_SIZE (KB).
# bwd_data's 2 GB of writes are its fp32 atomics reaching the memory side (36 atomic adds per input element).
HBM_TRAFFIC_GB = {'dcn_fwd': 0.93, 'dcn_bwd_data': 3.06, 'dcn_wgrad': 1.08}
# split-bf16 kernels with the XCD-aware work order, profiles/r1r_pmc_hbm_ops_xcd.txt (before it: r1q_...): counters of ONE tower-shaped launch (52.8 GFLOP, i.e. 2/3 of the mean
# launch of the step, random offsets): FETCH_SIZE x 2 + WRITE_SIZE
HBM_TRAFFIC_X3_GB = {'dcn_fwd': 2 * 0.057 + 0.045, 'dcn_bwd_data': 2 * 0.187 + 1.461, 'dcn_wgrad': 2 * 0.083 + 0.067}
BF16_MFMA_PEAK_TFLOPS = 2516.6   # MI355X_MICROARCH.md: ~2.5 PF dense bf16
FP32_MFMA_PEAK_TFLOPS = 157.3   # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32, dense, 2.4 GHz x 256 CU


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=10)
    ap.add_argument('--warmup', type=int, default=3)
    ap.add_argument('--task', default='bbox')
    ap.add_argument('--backbone', default='r50')
    ap.add_argument('--height', type=int, default=800)
    ap.add_argument('--width', type=int, default=1344)
    ap.add_argument('--batch', type=int, default=2, help='images per GPU')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-kernel-timing', action='store_true')
    ap.add_argument('--nchw', action='store_true', help='run in contiguous NCHW memory format (slower)')
    ap.add_argument('--math', default='bf16x3', choices=['bf16x3', 'fp32'],
                    help="arithmetic of the conv / deformable-conv contractions: split-bf16 products on the matrix pipe "
                         "with fp32 accumulation (rel. err 5e-6, the library default) or exact fp32 MFMA")
    ap.add_argument('--graph', action='store_true',
                    help='replay forward+backward from one captured hipGraph (runner/graph_step.py) instead of '
                         'launching every kernel eagerly; same speed while the step is GPU-bound')
    return ap.parse_args()


KERNEL_NOTES = {
    'bf16x3': {
        'dcn_fwd': 'lsn::dcn_fwd_x3_kernel (fused bilinear gather + split-bf16 MFMA implicit GEMM, forward)',
        'dcn_bwd_data': 'lsn::dcn_bwd_data_x3_kernel (gout x W^T as split-bf16 MFMA, merged bilinear scatter: grad '
                        'input/offset/mask)',
        'dcn_wgrad': 'lsn::dcn_wgrad_x3_kernel (gathered columns^T x gout as split-bf16 MFMA: grad weight/bias)',
    },
    'fp32': {
        'dcn_fwd': 'lsn::dcn_fwd_pipe_kernel (fused bilinear gather + fp32 MFMA implicit GEMM, forward)',
        'dcn_bwd_data': 'lsn::dcn_bwd_data_kernel / _win_kernel (gout x W^T on fp32 MFMA, fused bilinear scatter)',
        'dcn_wgrad': 'lsn::dcn_wgrad_kernel (gathered columns^T x gout on fp32 MFMA: grad weight/bias)',
    },
}


class KernelTimer:
    """Per-kernel launch durations from the library's own HIP-event log (lsn_prof_*): the events are
    recorded on the stream each deformable-conv kernel is launched on, around that kernel alone."""

    def __init__(self):
        from lsnet_amd import _lib
        self.lib = _lib

    def start(self):
        self.lib.prof_enable(True)

    def stop(self):
        torch.cuda.synchronize()
        rec = self.lib.prof_read()
        self.lib.prof_enable(False)
        out = {}
        for k, r in rec.items():
            if r['launches']:
                sec = r['total_ms'] * 1e-3
                out[k] = dict(launches=r['launches'], avg_ms=r['total_ms'] / r['launches'], total_ms=r['total_ms'],
                              tflops=r['flops'] / sec / 1e12, gflop_per_launch=r['flops'] / r['launches'] / 1e9,
                              alg_gbytes_per_launch=r['bytes'] / r['launches'] / 1e9, alg_gbps=r['bytes'] / sec / 1e9)
        return out


def build_step(model, cfg):
    """One training iteration through the runner's own hooks (lr schedule, optimizer hook)."""
    from lsnet_amd.runner import EpochBasedRunner, build_optimizer
    opt = build_optimizer(model, cfg.optimizer)
    runner = EpochBasedRunner(model, optimizer=opt, logger=lambda m: None)
    runner.register_training_hooks(cfg.lr_config, cfg.optimizer_config, None, dict(interval=10 ** 9, hooks=[]))
    runner.epoch_len = 10 ** 9
    runner.call_hook('before_run')
    runner.call_hook('before_train_epoch')

    def step(data):
        runner.call_hook('before_train_iter')
        runner.outputs = runner.run_iter(data)
        runner.log_buffer_update(runner.outputs['log_vars'], runner.outputs['num_samples'])
        runner.call_hook('after_train_iter')
        runner.iter += 1
        return runner.outputs

    return step, runner


def cpu_baseline(args):
    """The training step on the host CPU: this repo's host code + the CPU oracle behind the native
    ops (test infrastructure used as the timed baseline, never as the product path)."""
    from lsnet_amd.data import synthetic_batch
    from lsnet_amd.model_zoo import build_lsnet
    from lsnet_amd.ops import register_backend, unregister_backend
    from tests.oracle_backend import OracleBackend
    from oracle import oracle_py
    h, w, b = 416, 672, 1          # bounded sample: 1 image at ~1/4 of the 800x1344 area
    cores = os.cpu_count() or 1
    threads = min(cores, 64)
    torch.set_num_threads(threads)
    os.environ.setdefault('OMP_NUM_THREADS', str(threads))
    register_backend('cpu', OracleBackend())
    try:
        torch.manual_seed(0)
        model, cfg = build_lsnet(args.task, args.backbone)
        model.train()
        step, _ = build_step(model, cfg)
        data = synthetic_batch(args.task, b, h, w, seed=99, device='cpu', channels_last=False)
        nstep = 2
        t0 = time.time()
        for _ in range(nstep):
            out = step(data)
            loss = float(out['loss'].detach())
        dt = (time.time() - t0) / nstep
    finally:
        unregister_backend('cpu')
    return dict(value=b / dt, unit='img/s', cores=threads, kind='port',
                sample=f'{nstep} training steps (fwd+bwd+clip+SGD) of the same model on {b} image 3x{h}x{w} '
                       f'({oracle_py.num_threads()} OpenMP threads in the oracle, {threads} torch threads, '
                       f'{cores} host cores), {dt:.1f} s per step, last loss {loss:.3f}')


def main():
    args = parse()
    rank = int(os.environ.get('RANK', 0))
    world = int(os.environ.get('WORLD_SIZE', 1))
    local = int(os.environ.get('LOCAL_RANK', 0))
    assert torch.cuda.is_available(), 'bench.py measures the HIP path: a GPU is required'
    torch.cuda.set_device(local)
    dev = torch.device('cuda', local)
    if world > 1:
        os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
        dist.init_process_group('nccl')
    assert world == args.gpus, f'--gpus {args.gpus} but WORLD_SIZE={world}'

    from lsnet_amd.data import synthetic_batch
    from lsnet_amd.model_zoo import build_lsnet
    from lsnet_amd.ops import get_backend
    from lsnet_amd.parallel import DataParallelModel

    from lsnet_amd import _lib
    _lib.set_math_mode(args.math)
    torch.manual_seed(0)
    model, cfg = build_lsnet(args.task, args.backbone)
    model = model.to(dev)
    if not args.nchw:
        model = model.to(memory_format=torch.channels_last)
    model.train()
    if world > 1:
        model = DataParallelModel(model)
    step, runner = build_step(model, cfg)
    data = synthetic_batch(args.task, args.batch, args.height, args.width, seed=1234 + rank, device=dev,
                           channels_last=not args.nchw)
    timer = None if args.no_kernel_timing else KernelTimer()
    use_graph = args.graph
    if use_graph:
        # forward+backward replayed from one hipGraph.  Library autotuning and the capture itself happen here,
        # before the W warm-up steps of the contract (which then already replay the graph).
        gs = runner.enable_hip_graph(warmup=2)
        for _ in range(3):
            step(data)
        assert gs.graph is not None

    for _ in range(args.warmup):
        step(data)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    markers = os.environ.get('LSNET_PROF_MARKERS') == '1'
    if markers:   # a recognisable dispatch (lsn::selftest32_kernel) delimits the timed region in rocprof traces
        mk = torch.ones(32, 32, device=dev)
        get_backend(mk).selftest_mfma(mk, mk, 0)
        torch.cuda.synchronize()
    if timer and not use_graph:
        timer.start()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        out = step(data)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    ks = timer.stop() if (timer and not use_graph) else {}
    if timer and use_graph:
        # HIP events cannot bracket kernels inside a graph replay: the same kernels on the same tensors are
        # launched eagerly for a few forward+backward passes right after the timed region and timed there
        timer.start()
        for _ in range(min(args.steps, 5)):
            gs._fwd_bwd(data)
        ks = timer.stop()
    if markers:
        get_backend(mk).selftest_mfma(mk, mk, 0)
        torch.cuda.synchronize()
    if world > 1:
        t = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    losses = out['log_vars'].items_as_float() if hasattr(out['log_vars'], 'items_as_float') else {}
    alt = None
    if args.math == 'bf16x3' and world == 1 and not use_graph:
        # the same step with exact fp32 MFMA everywhere, for readers who only accept that arithmetic
        _lib.set_math_mode('fp32')
        for _ in range(2):
            step(data)
        torch.cuda.synchronize()
        ta = time.perf_counter()
        for _ in range(5):
            step(data)
        torch.cuda.synchronize()
        ta = (time.perf_counter() - ta) / 5
        alt = {'value': args.batch / ta, 'unit': 'img/s', 'ms_per_step': ta * 1e3, 'steps': 5,
               'math': 'exact fp32 MFMA / MIOpen fp32 for every contraction (--math fp32)'}
        _lib.set_math_mode(args.math)

    if rank == 0:
        imgs = args.batch * world * args.steps
        res = {
            'metric': 'img/s train LSNet R-50-FPN 1333x800 bs2/GPU',
            'value': imgs / dt, 'unit': 'img/s', 'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup,
            'ms_per_step': dt / args.steps * 1e3, 'higher_is_better': True, 'scaling': 'weak',
            'vs_baseline': None, 'dtype': 'f32', 'data': 'synthetic',
            'config': {'workload': f'LSNet {args.backbone.upper()}-FPN {args.task} (conv_module_type=dcn), '
                                   f'{args.batch} img/GPU 3x{args.height}x{args.width} (1333x800 padded to /32), '
                                   f'7 gt/img, fwd+bwd+RCCL grad all-reduce+clip35+SGD',
                       'global_batch': args.batch * world, 'parallelism': f'dp{world}',
                       'memory_format': 'nchw' if args.nchw else 'channels_last',
                       'launch': 'hipGraph replay of forward+backward; all-reduce, clip, SGD eager' if use_graph
                       else 'eager',
                       'math': 'fp32 tensors; conv / deformable-conv products as 3 bf16 MFMAs on split operands '
                               '(hi*hi + hi*lo + lo*hi), fp32 accumulation, rel. err 5e-6 vs exact fp32 (tests); '
                               'small / grouped / stride-2-backward convs and all conv weight gradients stay on MIOpen fp32'
                       if args.math == 'bf16x3' else 'exact fp32 MFMA / MIOpen fp32'},
            'loss': {k: round(v, 5) for k, v in losses.items()},
        }
        x3 = {'dcn_fwd', 'dcn_bwd_data', 'dcn_wgrad'} if args.math == 'bf16x3' else set()   # split-bf16 MFMA kernels

        def peak_of(k):
            return BF16_MFMA_PEAK_TFLOPS / 3.0 if k in x3 else FP32_MFMA_PEAK_TFLOPS

        def peak_note(k):
            return ('dense bf16 MFMA peak / 3 (three bf16 products per fp32 product)' if k in x3
                    else 'dense fp32 MFMA peak (v_mfma_f32_*_f32)')
        if ks:
            dom = max(ks, key=lambda k: ks[k]['total_ms'])   # the kernel with the most GPU time in the timed steps
            k = ks[dom]
            res['kernels'] = ks
            res['roofline'] = {'kernel': KERNEL_NOTES[args.math].get(dom, dom), 'bound': 'mfma', 'achieved': k['tflops'],
                               'peak': peak_of(dom), 'unit': 'TFLOP/s',
                               'frac': k['tflops'] / peak_of(dom), 'peak_note': peak_note(dom), 'traffic': (HBM_TRAFFIC_GB if args.math == 'fp32' else HBM_TRAFFIC_X3_GB).get(dom),
                               'traffic_unit': 'GB/launch, rocprofv3 FETCH_SIZE x2 + WRITE_SIZE: ' + (
                                   'mean launch of the step, profiles/r1c_pmc_hbm.txt' if args.math == 'fp32' else
                                   'one 52.8-GFLOP tower launch of the micro-benchmark (the mean launch of the step is '
                                   '79.3 GFLOP), profiles/r1r_pmc_hbm_ops_xcd.txt'),
                               'launches_timed': k['launches'], 'avg_launch_ms': k['avg_ms'],
                               'gflop_per_launch': k['gflop_per_launch'],
                               'alg_gbytes_per_launch': k['alg_gbytes_per_launch'],
                               'ms_per_step': k['total_ms'] / args.steps}
        if alt is not None:
            res['fp32_exact'] = alt
        if world == 1 and not args.no_cpu_baseline:
            try:
                res['cpu_baseline'] = cpu_baseline(args)
            except Exception as ex:   # the baseline must never take the bench line down
                res['cpu_baseline'] = {'value': None, 'unit': 'img/s', 'cores': 0, 'kind': 'port',
                                       'sample': f'failed: {type(ex).__name__}: {ex}'}
        print(json.dumps(res))
    if world > 1:
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
