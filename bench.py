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
  value        : at the library's default arithmetic 'bf16x6' = fp32-equivalent (include/lsnet_hip.h).
  roofline     : the hand-written kernel family with the most GPU time in the step, chosen over ALL instrumented families
                 (deformable conv forward / backward-data / weight gradient, dense conv forward / data gradient / weight
                 gradient, normalisation passes) by a survey pass of 3 untimed steps with HIP events around every launch;
                 the K timed steps then carry events around that ONE family only (a few dozen launches per step, so the
                 events do not slow the step they measure): algorithmic FLOPs (bytes, for the HBM-bound norm family) of
                 its launches / their HIP-event time on the launch stream, against the MFMA peak of its arithmetic --
                 2516 / 6 TFLOP/s for 'bf16x6', 157.3 for exact fp32 -- or 8 TB/s of HBM (MI355X_MICROARCH.md);
                 `traffic`: HBM bytes of this step's own launch shapes from committed rocprofv3 counter passes.
  kernels      : the survey pass: per family launches, ms per step, TFLOP/s and algorithmic GB/s.
  extra        : the same step in the other arithmetic modes (bf16x3: 3-term split; fp32: exact fp32 MFMA for the deformable family, the dense
                 convolutions stay on the fp32-equivalent 6-term split -- no vendor convolution in any mode),
                 5 steps each, and BASELINE config 5 (pose-head inference, bs 4).
  cpu_baseline : BASELINE config 1 (2 x 3x800x800) on the host CPU: this repo's host code with the CPU oracle standing
                 behind the native ops -- the reference has no CPU path for them -- rank 0 at N=1 only.
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
os.environ.setdefault('HIP_FORCE_DEV_KERNARG', '1')             # kernel arguments in device memory, see lsnet_amd/__init__.py
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

BF16_MFMA_PEAK_TFLOPS = 2516.6   # MI355X_MICROARCH.md: ~2.5 PF dense bf16
FP32_MFMA_PEAK_TFLOPS = 157.3    # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32, dense, 2.4 GHz x 256 CU
PRODUCTS = {'bf16x6': 6, 'bf16x3': 3}
# HBM traffic of the deformable-conv launches of THIS step (tower launch over 5 levels, pyramid launch over 15 pairs),
# written by tools/pmc_step_shapes.sh from rocprofv3 FETCH_SIZE / WRITE_SIZE passes (separate passes; FETCH doubled as
# MI355X_MICROARCH.md prescribes for gfx950) and committed with the round's profiles
TRAFFIC_FILE = os.path.join(ROOT, 'profiles', 'r6_hbm_traffic.json')
HBM_PEAK_GBPS = 8000.0           # MI355X_MICROARCH.md: HBM3E, 8 TB/s
HBM_BOUND = ('norm',)            # families that are streaming passes, priced against HBM bandwidth


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
    ap.add_argument('--math', default='bf16x6', choices=['bf16x6', 'bf16x3', 'fp32'],
                    help="arithmetic of the conv / deformable-conv contractions: 'bf16x6' = fp32-equivalent (exact 3-way "
                         "bf16 split of every operand, 6 product terms, fp32 accumulation; the library default), "
                         "'bf16x3' = 2-way split, 3 terms (rel. err 5e-6), 'fp32' = exact fp32 MFMA")
    ap.add_argument('--cpu-baseline-worker', action='store_true', help=argparse.SUPPRESS)
    ap.add_argument('--no-extra', action='store_true', help='skip the extra legs (other arithmetic modes, inference)')
    ap.add_argument('--graph', action='store_true',
                    help='replay forward+backward from one captured hipGraph (runner/graph_step.py) instead of '
                         'launching every kernel eagerly; same speed while the step is GPU-bound')
    return ap.parse_args()


_COMMON_NOTES = {
    'conv_fwd': 'lsn::conv_mm_kernel (dense conv forward: implicit GEMM, pixel planes split into a swizzled LDS image, weight '
                'fragments straight from L2, two workgroups per CU) + conv_splitk_reduce_kernel',
    'conv_bwd_data': 'lsn::conv_mm_kernel on grad_output with transposed / flipped weight images (one launch per residue '
                     'class of a strided convolution)',
    'conv_wgrad': 'lsn::conv_wgrad_kernel + conv_wgrad_reduce_kernel (3x3 / 1x1: input patch + grad_output rows staged once '
                  'per 16-pixel segment, ds_read_b64_tr_b16 fragments, split-pixel partial tiles); wide 3x3 / deep 1x1 layers: '
                  'dcn_tap_table_kernel + dcn_gout_frag_kernel + dcn_wgrad_mm_kernel<NP, DENSE> (grad_output pre-split once into '
                  'fragment order); other strided 3x3 and tap counts: dcn_wgrad_xn_kernel<PLAIN>',
    'norm': 'lsn::bn_act_fwd / bn_act_bwd / bn_param_reduce and gn_stats / gn_apply / gn_bwd_reduce / gn_bwd_apply / '
            'gn_param_grad kernels (frozen-statistics BatchNorm + add + ReLU, GroupNorm + ReLU; streaming passes)',
    'gconv': 'lsn::gconv_kernel / gconv_wgrad_kernel (grouped convolution, exact fp32)',
}
KERNEL_NOTES = {
    'split': dict(_COMMON_NOTES, **{
        'dcn_fwd': 'lsn::dcn_fwd_mm_kernel (bilinear gather fused into the dense split-bf16 implicit-GEMM skeleton: two '
                   'workgroups per CU, weights as MFMA fragments from L2)',
        'dcn_bwd_data': 'lsn::conv_mm_kernel as a 1x1 convolution of grad_output (gout x W^T -> unweighted column gradients) '
                        '+ dcn_anchor_sum_kernel / dcn_anchor_combine_kernel (atomic-free grad input from per-anchor sample '
                        'lists, corner sums of grad offset/mask formed from the same rows) + dcn_offgrad_kernel; list '
                        'building (dcn_bin / scan / fill / sort_lists) inside the timed bracket',
        'dcn_wgrad': 'lsn::dcn_gout_frag_kernel + dcn_wgrad_mm_kernel + conv_wgrad_reduce_kernel (gathered columns^T x gout '
                     'as split-bf16 MFMA, grad_output pre-arranged in fragment order, split partial tiles: grad weight/bias)',
    }),
    'fp32': dict(_COMMON_NOTES, **{
        'dcn_fwd': 'lsn::dcn_fwd_kernel (fused bilinear gather + fp32 MFMA implicit GEMM, forward)',
        'dcn_bwd_data': 'lsn::dcn_gcol_mfma_kernel (gout x W^T on fp32 MFMA -> unweighted column gradients) + the atomic-free '
                        'per-anchor gather of the default mode (dcn_anchor_sum / _combine / dcn_offgrad kernels): bit-reproducible',
        'dcn_wgrad': 'lsn::dcn_wgrad_kernel (gathered columns^T x gout on fp32 MFMA, per-split partial tiles + ordered reduce)',
    }),
}
MATH_NOTES = {
    'bf16x6': 'fp32 tensors; every conv / deformable-conv product a*b evaluated as 6 bf16 MFMA terms of the EXACT 3-way '
              'bf16 splits a = h+m+l, b = h+m+l (hh+hm+mh+mm+hl+lh; dropped terms <= 2^-25 relative, below the 2^-24 '
              'rounding of an fp32 product), fp32 accumulation: fp32-equivalent, <= 1e-6 of the output range against the '
              'exact-fp32 MFMA kernels (tests/test_ops_gpu.py::test_split6_matches_exact_fp32)',
    'bf16x3': 'fp32 tensors; products as 3 bf16 MFMA terms of 2-way splits (hh+hl+lh), fp32 accumulation, rel. err 5e-6',
    'fp32': 'exact fp32 MFMA for the deformable family (round 6: atomic-free, bit-reproducible like the default mode); dense '
            'convolutions on the 6-term split kernels (fp32-equivalent): no vendor convolution in any mode',
}


class KernelTimer:
    """Per-kernel launch durations from the library's own HIP-event log (lsn_prof_*): the events are
    recorded on the stream each deformable-conv kernel is launched on, around that kernel alone."""

    def __init__(self):
        from lsnet_amd import _lib
        self.lib = _lib

    def start(self, families=None):
        self.lib.prof_enable(True, families)

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


CPU_BASELINE_THREADS = 64     # more threads were slower on the 256-core host of the GPU box (a 256-thread run of this
                              # leg did not finish one step in 15 minutes: oversubscribed OpenMP team + torch pool)
CPU_BASELINE_LIMIT_S = 300    # hard wall-clock limit of the leg (it runs in a child process)


def cpu_baseline_worker(args):
    """Child process: BASELINE config 1 on the host CPU.  Prints one JSON line."""
    from lsnet_amd.data import synthetic_batch
    from lsnet_amd.model_zoo import build_lsnet
    from lsnet_amd.ops import register_backend
    from tests.oracle_backend import OracleBackend
    from oracle import oracle_py
    h, w, b = 800, 800, 2
    cores = os.cpu_count() or 1
    threads = min(cores, CPU_BASELINE_THREADS)
    torch.set_num_threads(threads)
    register_backend('cpu', OracleBackend())
    torch.manual_seed(0)
    model, cfg = build_lsnet(args.task, args.backbone)
    model.train()
    step, _ = build_step(model, cfg)
    data = synthetic_batch(args.task, b, h, w, seed=99, device='cpu', channels_last=False)
    t0 = time.time()
    step(data)                                   # warm-up (allocations, thread pools), not timed
    warm = time.time() - t0
    nstep = 3 if warm < 45 else 1                # (~25 s per step on the GPU box's host: 3 timed steps, ~110 s in all)
    t0 = time.time()
    for _ in range(nstep):
        out = step(data)
        loss = float(out['loss'].detach())
    dt = (time.time() - t0) / nstep
    print(json.dumps(dict(
        value=b / dt, unit='img/s', cores=threads, kind='port',
        sample=f'BASELINE config 1: {nstep} timed training step(s) (fwd+bwd+clip+SGD; 1 warm-up step of {warm:.1f} s '
               f'excluded) of LSNet {args.backbone.upper()}-FPN {args.task} on {b} images 3x{h}x{w}; '
               f'torch.set_num_threads({threads}), {oracle_py.num_threads()} OpenMP threads in the oracle, host has '
               f'{cores} cores; {dt:.1f} s per step, last loss {loss:.3f}')))


def cpu_baseline(args):
    """BASELINE config 1 (2 images 3x800x800) on the host: the training step of the same model with this repo's host
    code and the CPU oracle behind the native ops (the reference has no CPU path for them: SURVEY.md section 0.2), in a
    child process with a hard time limit.  Test infrastructure used as the timed baseline, never as the product."""
    import subprocess
    cores = os.cpu_count() or 1
    threads = min(cores, CPU_BASELINE_THREADS)
    env = dict(os.environ, OMP_NUM_THREADS=str(threads), MKL_NUM_THREADS=str(threads), HIP_VISIBLE_DEVICES='',
               CUDA_VISIBLE_DEVICES='')
    cmd = [sys.executable, os.path.abspath(__file__), '--cpu-baseline-worker', '--task', args.task, '--backbone',
           args.backbone]
    try:
        p = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=CPU_BASELINE_LIMIT_S)
    except subprocess.TimeoutExpired:
        return {'value': None, 'unit': 'img/s', 'cores': threads, 'kind': 'port',
                'sample': f'BASELINE config 1 (2 x 3x800x800) did not finish within {CPU_BASELINE_LIMIT_S} s'}
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith('{')]
    if p.returncode != 0 or not lines:
        return {'value': None, 'unit': 'img/s', 'cores': threads, 'kind': 'port',
                'sample': 'failed: ' + (p.stderr.strip().splitlines() or ['no output'])[-1][:300]}
    return json.loads(lines[-1])


def infer_leg(dev):
    """BASELINE config 5: LSNet R-50-FPN pose head (17 keypoints), 4 images 3x800x1344, forward + decode + NMS on the
    device (the reference's tools/benchmark.py:63-89 times the same region at batch 1)."""
    from lsnet_amd.model_zoo import build_lsnet
    B, H, W = 4, 800, 1344
    torch.manual_seed(0)
    model, _ = build_lsnet('pose_kbox', 'r50')
    model = model.to(dev).to(memory_format=torch.channels_last).eval()
    with torch.no_grad():
        # random-init classification logits sit at -4.6 (bias init) and nothing would pass score_thr = 0.05: shift the
        # bias so that a realistic share of the points survives and the decode / NMS stages do real work
        model.bbox_head.pts_cls_out.bias.add_(2.0)
    img = torch.randn(B, 3, H, W, device=dev).contiguous(memory_format=torch.channels_last)
    metas = [dict(pad_shape=(H, W, 3), img_shape=(H, W, 3), scale_factor=1.0, ori_shape=(H, W, 3), flip=False)] * B
    with torch.no_grad():
        for _ in range(2):
            dets = model.simple_test_batch(img, metas)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        n = 5
        for _ in range(n):
            dets = model.simple_test_batch(img, metas)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / n
    return dict(metric='inference latency LSNet R-50-FPN pose head (17 kps), bs 4, 3x800x1344, decode + NMS included',
                ms_per_batch=dt * 1e3, img_per_s=B / dt, detections_img0=int(dets[0][0].shape[0]))


def config_leg(dev, task, backbone, n=3, warm=2):
    """Another BASELINE config through the same step (runner hooks, gradient arena, clip, SGD) at 2 x 3x800x1344:
    config 4 = X-101-64x4d-DCN segm (grouped DCNv2 in c3-c5, 36-landmark polygon head, activation checkpointing)."""
    from lsnet_amd.data import synthetic_batch
    from lsnet_amd.model_zoo import build_lsnet
    from lsnet_amd.parallel import DataParallelModel
    torch.manual_seed(0)
    model, cfg = build_lsnet(task, backbone)
    model = DataParallelModel(model.to(dev).to(memory_format=torch.channels_last).train())
    step, _ = build_step(model, cfg)
    data = synthetic_batch(task, 2, 800, 1344, seed=1234, device=dev, channels_last=True)
    dt = timed_steps(step, data, n, warm)
    out = dict(metric=f'img/s train LSNet {backbone.upper()} {task}, 2 x 3x800x1344', value=2 / dt, unit='img/s',
               ms_per_step=dt * 1e3, steps=n)
    del model, step, data
    torch.cuda.empty_cache()
    return out


MS_SHAPES = [(800, 1344), (960, 1344), (640, 1344), (1344, 800), (480, 1344)]


def mstrain_leg(dev, n_cycles=2):
    """BASELINE config 3 the way it trains: R-101-DCN bbox with `img_scale=[(1333, 480), (1333, 960)],
    multiscale_mode='range'` (configs/lsnet/lsnet_bbox_r50_fpn_mstrain_2x_coco.py:13-15) and GroupSampler's landscape /
    portrait batches (mmdet/datasets/samplers/group_sampler.py:60-140): the padded batch shape changes EVERY iteration.
    One pass over MS_SHAPES untimed (scratch buffers and the allocator meet every shape), then n_cycles timed passes;
    beside it the same model at the fixed 800 x 1344, and what the library asked of the HIP runtime during the timed passes
    (lsn_scratch_stats: must be nothing)."""
    from lsnet_amd import _lib
    from lsnet_amd.data import synthetic_batch
    from lsnet_amd.model_zoo import build_lsnet
    from lsnet_amd.parallel import DataParallelModel
    torch.manual_seed(0)
    model, cfg = build_lsnet('bbox', 'r101-dcn')
    model = DataParallelModel(model.to(dev).to(memory_format=torch.channels_last).train())
    step, _ = build_step(model, cfg)
    batches = [synthetic_batch('bbox', 2, h, w, seed=1234 + i, device=dev, channels_last=True) for i, (h, w) in enumerate(MS_SHAPES)]
    fixed = timed_steps(step, batches[0], 3, 2)
    for b in batches:
        step(b)
    torch.cuda.synchronize()
    s0 = _lib.scratch_stats()
    a0 = torch.cuda.memory_stats().get('num_device_alloc', 0)
    t0 = time.perf_counter()
    for _ in range(n_cycles):
        for b in batches:
            out = step(b)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / (n_cycles * len(batches))
    s1 = _lib.scratch_stats()
    a1 = torch.cuda.memory_stats().get('num_device_alloc', 0)
    # image-area weighted: the cycle's mean image has 1.0 x the pixels of 800 x 1344 only by accident of the list
    px = sum(h * w for h, w in MS_SHAPES) / len(MS_SHAPES) / (800 * 1344)
    res = dict(metric='img/s train LSNet R101-DCN bbox, 2 img/GPU, padded shape cycling ' + ' / '.join(f'{h}x{w}' for h, w in MS_SHAPES),
               value=2 / dt, unit='img/s', ms_per_step=dt * 1e3, steps=n_cycles * len(batches),
               fixed_shape_800x1344={'value': 2 / fixed, 'ms_per_step': fixed * 1e3},
               mean_pixels_vs_800x1344=px, value_per_800x1344_image=2 / dt * px,
               ratio_to_fixed_shape_pixel_normalised=(2 / dt * px) / (2 / fixed),
               library_hipMalloc_calls_during_timed_passes=s1['mallocs'] - s0['mallocs'],
               library_blocking_syncs_during_timed_passes=s1['blocking_syncs'] - s0['blocking_syncs'],
               library_scratch_mbytes=s1['held_bytes'] / 2 ** 20,
               torch_allocator_device_allocs_during_timed_passes=int(a1 - a0),
               loss=float(out['log_vars']['loss']))
    del model, step, batches
    torch.cuda.empty_cache()
    return res


def timed_steps(step, data, n, warm):
    for _ in range(warm):
        step(data)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        step(data)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n


ITER0_FIXTURE = os.path.join(ROOT, 'tests', 'golden', 'bench_iter0.npz')


def iteration0_losses(model, data):
    """The losses of the untouched model on the bench batch (one forward, no update), as the reference's
    `_parse_losses` (detectors/base.py:176-209) reports them."""
    mod = model.module if hasattr(model, 'module') else model
    prev = getattr(mod, '_log_reduce_elsewhere', False)
    mod._log_reduce_elsewhere = True   # rank 0 alone runs this pass: no collective in it (N > 1: _parse_losses would all-reduce)
    try:
        with torch.no_grad():
            _, log_vars = mod._parse_losses(model(**data))
    finally:
        mod._log_reduce_elsewhere = prev
    return {k: float(v) for k, v in log_vars.items()}


def iteration0_parity(args, iter0):
    """iter0 against tests/golden/bench_iter0.npz = the REFERENCE's detector on the same weights and batch (one CPU forward
    in the build container, oracle/ref_harness/make_golden.py::golden_bench_iter0).  Only the default workload has one."""
    if iter0 is None or (args.task, args.backbone, args.batch, args.height, args.width) != ('bbox', 'r50', 2, 800, 1344):
        return None
    try:
        import numpy as np
        ref = np.load(ITER0_FIXTURE)
    except OSError:
        return None
    keys = ('loss_cls', 'loss_bbox_init', 'loss_bbox_refine', 'loss')
    want = {k: float(ref[k]) for k in keys}
    return {'loss_iter0': {k: round(iter0[k], 6) for k in keys}, 'loss_iter0_ref': {k: round(want[k], 6) for k in keys},
            'loss_ref_rel_err': max(abs(iter0[k] - want[k]) / abs(want[k]) for k in keys)}


def flush_c_stdio():
    """RCCL writes its version banner with C stdio (fully buffered when stdout is a file: it would surface at process
    exit, AFTER the JSON line).  Everything buffered so far goes out now, so that the JSON line stays the last line."""
    import ctypes
    sys.stdout.flush()
    try:
        ctypes.CDLL(None).fflush(None)
    except Exception:
        pass


def allreduce_probe(model, dev, world, reps=5):
    """Stand-alone time of one gradient all-reduce of the step (all buckets, back to back, nothing to overlap with)."""
    flats = [b['flat'] for b in model.reducer.buckets]
    for _ in range(2):
        for f in flats:
            dist.all_reduce(f)
    torch.cuda.synchronize()
    dist.barrier()
    t0 = time.perf_counter()
    for _ in range(reps):
        works = [dist.all_reduce(f, async_op=True) for f in flats]
        for w in works:
            w.wait()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / reps
    nbytes = sum(f.numel() * 4 for f in flats)
    return dict(allreduce_ms_per_step_standalone=dt * 1e3, allreduce_mbytes=nbytes / 1e6, buckets=len(flats),
                allreduce_algbw_gbps=nbytes / dt / 1e9)


def main():
    args = parse()
    if args.cpu_baseline_worker:
        return cpu_baseline_worker(args)
    rank = int(os.environ.get('RANK', 0))
    world = int(os.environ.get('WORLD_SIZE', 1))
    local = int(os.environ.get('LOCAL_RANK', 0))
    assert torch.cuda.is_available(), 'bench.py measures the HIP path: a GPU is required'
    # one process per GPU.  (LSNET_BENCH_BACKEND=gloo with more ranks than GPUs is the self-test of the N > 1 code path
    # on a single-GPU box: ranks then share a device and the gradients travel through the host.)
    backend = os.environ.get('LSNET_BENCH_BACKEND', 'nccl')
    if backend != 'nccl':
        local %= torch.cuda.device_count()
    torch.cuda.set_device(local)
    dev = torch.device('cuda', local)
    if world > 1:
        os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
        dist.init_process_group(backend)
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
    model = DataParallelModel(model)   # N = 1: the gradient arena alone (no communication), as train_detector builds it
    exposed = []   # N > 1: (start, end) events around the wait for the gradient all-reduces = what backward did not hide
    if world > 1:
        finish = model.reduce_gradients

        def timed_finish():
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            finish()
            e1.record()
            exposed.append((e0, e1))
        model.reduce_gradients = timed_finish
    step, runner = build_step(model, cfg)
    data = synthetic_batch(args.task, args.batch, args.height, args.width, seed=1234 + rank, device=dev,
                           channels_last=not args.nchw)
    iter0 = iteration0_losses(model, data) if rank == 0 else None   # before any update: SURVEY 8(d) "Loss parity"
    timer = None if args.no_kernel_timing else KernelTimer()
    survey, dominant = {}, None
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
    if timer and not use_graph:
        # survey: events around every launch of every family for 3 untimed steps -> which family dominates the step
        timer.start()
        for _ in range(3):
            step(data)
        survey = timer.stop()
        for v in survey.values():
            v['ms_per_step'] = v['total_ms'] / 3
        if survey:
            dominant = max(survey, key=lambda k: survey[k]['total_ms'])
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    markers = os.environ.get('LSNET_PROF_MARKERS') == '1'
    if markers:   # a recognisable dispatch (lsn::selftest32_kernel) delimits the timed region in rocprof traces
        mk = torch.ones(32, 32, device=dev)
        get_backend(mk).selftest_mfma(mk, mk, 0)
        torch.cuda.synchronize()
    if timer and not use_graph and dominant:
        timer.start([dominant])      # HIP events around the launches of the dominant family only
    del exposed[:]
    t0 = time.perf_counter()
    for _ in range(args.steps):
        out = step(data)
    torch.cuda.synchronize()
    dt_rank = time.perf_counter() - t0     # this rank's own K steps, before it waits for the others
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    exposed_ms = sum(a.elapsed_time(b) for a, b in exposed) / max(len(exposed), 1) if exposed else 0.0
    ks = timer.stop() if (timer and not use_graph) else {}
    if markers:
        get_backend(mk).selftest_mfma(mk, mk, 0)
        torch.cuda.synchronize()
    per_rank = None
    if world > 1:
        t = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
        mine = torch.tensor([dt_rank / args.steps * 1e3, exposed_ms], device=dev, dtype=torch.float64)
        allr = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(allr, mine)
        per_rank = [[round(float(v), 3) for v in r] for r in allr]
    losses = out['log_vars'].items_as_float() if hasattr(out['log_vars'], 'items_as_float') else {}

    nk = args.steps
    if timer and use_graph:
        # HIP events cannot bracket kernels inside a graph replay: the same kernels on the same tensors are launched
        # eagerly for a few forward+backward passes right after the timed region and timed there
        nk = min(args.steps, 5)
        timer.start()
        for _ in range(nk):
            gs._fwd_bwd(data)
        ks = timer.stop()
    comm = allreduce_probe(model, dev, world) if world > 1 else None
    flush_c_stdio()   # (the communicators exist by now: RCCL's banner, if any, is out before the JSON line)
    comm1 = None
    if world == 1 and not args.no_extra:
        # the same buckets through a ONE-rank RCCL group: what a stand-alone gradient all-reduce of this step costs on this
        # box before any second GPU is involved (launch + RCCL's own kernels; tests/test_rccl_single_gpu.py runs the
        # hook-driven path the same way).  Reported under config, never part of `value`.
        try:
            os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
            import socket
            with socket.socket() as sk:
                sk.bind(('127.0.0.1', 0))
                port = sk.getsockname()[1]
            dist.init_process_group('nccl', init_method=f'tcp://127.0.0.1:{port}', rank=0, world_size=1)
            comm1 = allreduce_probe(model, dev, 1)
            comm1['rccl_ranks'] = dist.get_world_size()
            # the TIMED step with the seven bucket all-reduces of every step sent through that group (launched from the hooks
            # during backward, on RCCL's stream, waited for in finish()) beside the plain step, alternating: the closest
            # single-GPU proxy of what a rank of an N-GPU run pays besides the wire
            red = model.reducer
            t_plain, t_coll = [], []
            for _ in range(2):
                red.collective = False
                t_plain.append(timed_steps(step, data, 5, 1))
                red.collective = True
                t_coll.append(timed_steps(step, data, 5, 1))
            red.collective = False
            comm1['step_ms_plain'] = min(t_plain) * 1e3
            comm1['step_ms_with_collectives'] = min(t_coll) * 1e3
            comm1['step_overhead_frac'] = min(t_coll) / min(t_plain) - 1.0
            dist.destroy_process_group()
            flush_c_stdio()
        except Exception as ex:   # must never take the bench line down
            comm1 = {'allreduce_ms_per_step_standalone': None, 'error': f'{type(ex).__name__}: {ex}'}

    extra = {}
    if world == 1 and not use_graph and not args.no_extra:
        for mode in ('bf16x3', 'fp32'):
            if mode == args.math:
                continue
            _lib.set_math_mode(mode)
            ta = timed_steps(step, data, 5, 2)
            extra[mode] = {'value': args.batch / ta, 'unit': 'img/s', 'ms_per_step': ta * 1e3, 'steps': 5,
                           'math': MATH_NOTES[mode]}
        _lib.set_math_mode(args.math)
        try:
            extra['infer_pose_bs4'] = infer_leg(dev)
        except Exception as ex:   # an extra leg must never take the bench line down
            extra['infer_pose_bs4'] = {'error': f'{type(ex).__name__}: {ex}'}
        try:
            extra['config4_segm_x101_dcn'] = config_leg(dev, 'segm', 'x101-dcn')
        except Exception as ex:
            extra['config4_segm_x101_dcn'] = {'error': f'{type(ex).__name__}: {ex}'}
        try:
            extra['config3_r101_dcn_mstrain'] = mstrain_leg(dev)
        except Exception as ex:
            extra['config3_r101_dcn_mstrain'] = {'error': f'{type(ex).__name__}: {ex}'}

    if rank == 0:
        imgs = args.batch * world * args.steps
        np_ = PRODUCTS.get(args.math, 0)
        res = {
            'metric': 'img/s train LSNet R-50-FPN 1333x800 bs2/GPU',
            'value': imgs / dt, 'unit': 'img/s', 'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup,
            'ms_per_step': dt / args.steps * 1e3, 'higher_is_better': True, 'scaling': 'weak',
            'vs_baseline': None,
            'dtype': {'bf16x6': 'f32 (fp32-equivalent: 6 bf16 MFMA terms per product on exact 3-way operand splits, '
                                'fp32 accumulate)',
                      'bf16x3': 'f32 tensors, bf16x3 split products (16 mantissa bits per operand)',
                      'fp32': 'f32'}[args.math],
            'data': 'synthetic',
            'config': {'workload': f'LSNet {args.backbone.upper()}-FPN {args.task} (conv_module_type=dcn), '
                                   f'{args.batch} img/GPU 3x{args.height}x{args.width} (1333x800 padded to /32), '
                                   f'7 gt/img, fwd+bwd+' +
                                   ('RCCL grad all-reduce (bucketed, overlapped with backward)+' if world > 1 else
                                    'gradient arena (the reducer\'s buckets, no collective on one rank)+') + 'clip35+SGD',
                       'global_batch': args.batch * world, 'parallelism': f'dp{world}', 'world_size': world,
                       'memory_format': 'nchw' if args.nchw else 'channels_last',
                       'launch': 'hipGraph replay of forward+backward; all-reduce, clip, SGD eager' if use_graph
                       else 'eager',
                       'math': MATH_NOTES[args.math]},
            'loss': {k: round(v, 5) for k, v in losses.items()},
        }
        par = iteration0_parity(args, iter0)
        if par:
            res.update(par)   # iteration-0 losses vs the reference's on the same weights / batch (tolerance 1e-3)
        if comm1 is not None:
            res['config']['one_rank_rccl'] = comm1
        if world > 1:
            try:
                res['config']['rccl_version'] = '.'.join(str(v) for v in torch.cuda.nccl.version())
            except Exception:
                res['config']['rccl_version'] = 'unknown'
            res['config'].update(comm)
            res['config']['rccl_ranks'] = dist.get_world_size()    # what the process group reports, not the flag
            res['config']['per_rank_ms_per_step'] = [r[0] for r in per_rank]
            # GPU time the compute stream spent waiting for the bucket all-reduces after backward (mean per step, per rank):
            # the part of `allreduce_ms_per_step_standalone` that backward did NOT hide
            res['config']['allreduce_exposed_ms_per_step'] = [r[1] for r in per_rank]
            res['config']['grad_buckets_mb'] = [round(b['flat'].numel() * 4 / 2 ** 20, 1) for b in model.reducer.buckets]

        peak = BF16_MFMA_PEAK_TFLOPS / np_ if np_ else FP32_MFMA_PEAK_TFLOPS
        peak_note = (f'dense bf16 MFMA peak / {np_} ({np_} bf16 products per fp32 product)' if np_
                     else 'dense fp32 MFMA peak (v_mfma_f32_*_f32)')
        if use_graph and ks:
            survey = ks
            for v in survey.values():
                v['ms_per_step'] = v['total_ms'] / max(nk, 1)
            dominant = max(ks, key=lambda k: ks[k]['total_ms'])
        if ks and dominant in ks:
            dom = dominant   # the kernel family with the most GPU time (survey pass), timed here over the K steps
            k = ks[dom]
            for fam, v in survey.items():      # every family against its own roof
                if fam in HBM_BOUND:
                    v['bound'], v['frac'] = 'hbm', v['alg_gbps'] / HBM_PEAK_GBPS
                else:
                    v['bound'], v['frac'] = 'mfma', v['tflops'] / peak
            res['kernels'] = survey
            traffic, traffic_note = None, 'not measured'
            try:
                with open(TRAFFIC_FILE) as f:
                    tf = json.load(f)
                from lsnet_amd.csrc.build import kernel_signature
                if tf.get('kernel_signature') != kernel_signature():
                    traffic_note = 'not reported: the kernel sources changed since the counters were collected'
                elif tf.get('math') == args.math and dom in tf.get('kernels', {}):
                    e = tf['kernels'][dom]
                    traffic = e['gbytes_per_mean_launch']
                    traffic_note = e['note']
            except (OSError, ValueError, KeyError):
                pass
            hbm = dom in HBM_BOUND
            res['roofline'] = {'family': dom, 'kernel': KERNEL_NOTES['split' if np_ else 'fp32'].get(dom, dom),
                               'bound': 'hbm' if hbm else 'mfma',
                               'achieved': k['alg_gbps'] if hbm else k['tflops'],
                               'peak': HBM_PEAK_GBPS if hbm else peak, 'unit': 'GB/s' if hbm else 'TFLOP/s',
                               'frac': k['alg_gbps'] / HBM_PEAK_GBPS if hbm else k['tflops'] / peak,
                               'peak_note': 'HBM3E 8 TB/s' if hbm else peak_note,
                               'traffic': traffic, 'traffic_unit': 'GB per mean launch: ' + traffic_note,
                               'launches_timed': k['launches'], 'avg_launch_ms': k['avg_ms'],
                               'gflop_per_launch': k['gflop_per_launch'],
                               'alg_gbytes_per_launch': k['alg_gbytes_per_launch'],
                               'ms_per_step': k['total_ms'] / max(nk, 1)}
            # the whole step: algorithmic flops of every matrix-pipe family of one step (survey pass) / the timed step
            nsurvey = 3 if not use_graph else max(nk, 1)
            step_flops = sum(v['gflop_per_launch'] * 1e9 * v['launches'] / nsurvey for f, v in survey.items() if f not in HBM_BOUND)
            res['roofline']['step_tflops'] = step_flops / (dt / args.steps) / 1e12
            res['roofline']['step_frac'] = res['roofline']['step_tflops'] / peak
            if traffic is not None and not hbm:
                # what the counters say about the same launches: counted HBM bytes per launch / its duration.  A family whose
                # counted rate is a large share of HBM's while its flop rate is a small share of the matrix pipe's is bound by
                # the memory system in practice, whatever its algorithmic intensity promises ('bound' stays the paper roof)
                cg = traffic / (k['avg_ms'] * 1e-3)
                res['roofline']['counted_gbps'] = cg
                res['roofline']['counted_hbm_frac'] = cg / HBM_PEAK_GBPS
                res['roofline']['bound_by_counters'] = 'hbm' if cg / HBM_PEAK_GBPS > k['tflops'] / peak else 'mfma'
        if extra:
            res['extra'] = extra
        if world == 1 and not args.no_cpu_baseline:
            try:
                res['cpu_baseline'] = cpu_baseline(args)
            except Exception as ex:   # the baseline must never take the bench line down
                res['cpu_baseline'] = {'value': None, 'unit': 'img/s', 'cores': 0, 'kind': 'port',
                                       'sample': f'failed: {type(ex).__name__}: {ex}'}
        flush_c_stdio()
        print(json.dumps(res), flush=True)
    if world > 1:
        flush_c_stdio()
        dist.destroy_process_group()
        flush_c_stdio()


if __name__ == '__main__':
    main()
