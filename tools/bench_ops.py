"""Micro-benchmarks of the hand-written kernels at the LSHead shapes (R-50-FPN, 800x1344, bs 2).

Prints per-op average time (HIP events on the launch stream), achieved fp32 TFLOP/s against the
157.3 TF MFMA peak, and the same for torch's (MIOpen) conv2d as a yardstick."""
import argparse
import json
import sys
import os

import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lsnet_amd import ops  # noqa: E402

PEAK = 157.3e12
LEVELS = [(100, 168), (50, 84), (25, 42), (13, 21), (7, 11)]


def timeit(fn, iters=10, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e-3


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--B', type=int, default=2)
    ap.add_argument('--C', type=int, default=256)
    ap.add_argument('--iters', type=int, default=10)
    ap.add_argument('--what', default='all')
    args = ap.parse_args()
    dev = torch.device('cuda:0')
    B, C = args.B, args.C
    cl = torch.channels_last
    torch.manual_seed(0)
    res = {}
    w = (torch.randn(C, C, 3, 3, device=dev) * 0.02).contiguous(memory_format=cl).requires_grad_()
    b = torch.zeros(C, device=dev, requires_grad=True)
    xs = [torch.randn(B, C, h, ww, device=dev).contiguous(memory_format=cl).requires_grad_() for h, ww in LEVELS]
    offs = [(torch.randn(B, 18, h, ww, device=dev) * 0.5).contiguous(memory_format=cl).requires_grad_()
            for h, ww in LEVELS]
    msks = [torch.rand(B, 9, h, ww, device=dev).contiguous(memory_format=cl).requires_grad_() for h, ww in LEVELS]
    npix = [B * h * ww for h, ww in LEVELS]
    flops_lv = [2.0 * p * C * C * 9 for p in npix]

    def run(name, lv):
        xl, ol, ml = [xs[i] for i in lv], [offs[i] for i in lv], [msks[i] for i in lv]
        fl = sum(flops_lv[i] for i in lv)
        with torch.no_grad():
            t = timeit(lambda: ops.dcn_multi(xl, ol, ml, w, b, 1, 1, 1), args.iters)
        res[name + '_fwd'] = dict(ms=t * 1e3, tflops=fl / t / 1e12, frac=fl / t / PEAK)
        outs = ops.dcn_multi(xl, ol, ml, w, b, 1, 1, 1)
        gos = [torch.randn_like(o) for o in outs]

        def bwd():
            torch.autograd.grad(outs, [w, b] + xl + ol + ml, gos, retain_graph=True)
        t = timeit(bwd, args.iters)
        res[name + '_bwd'] = dict(ms=t * 1e3, tflops=2 * fl / t / 1e12, frac=2 * fl / t / PEAK)

    if args.what == 'dcn_all5':
        run('dcn_all5', [0, 1, 2, 3, 4])
    if args.what in ('all', 'dcn'):
        run('dcn_p3', [0])
        run('dcn_p4', [1])
        run('dcn_p7', [4])
        run('dcn_all5', [0, 1, 2, 3, 4])
    if args.what in ('all', 'conv'):
        for fmt, name in ((cl, 'nhwc'), (torch.contiguous_format, 'nchw')):
            x = torch.randn(B, C, 100, 168, device=dev).contiguous(memory_format=fmt).requires_grad_()
            wt = (torch.randn(C, C, 3, 3, device=dev) * 0.02).contiguous(memory_format=fmt).requires_grad_()
            fl = flops_lv[0]
            with torch.no_grad():
                t = timeit(lambda: F.conv2d(x, wt, None, 1, 1), args.iters)
            res[f'torch_conv3x3_p3_{name}_fwd'] = dict(ms=t * 1e3, tflops=fl / t / 1e12, frac=fl / t / PEAK)
            out = F.conv2d(x, wt, None, 1, 1)
            go = torch.randn_like(out)
            t = timeit(lambda: torch.autograd.grad(out, [x, wt], go, retain_graph=True), args.iters)
            res[f'torch_conv3x3_p3_{name}_bwd'] = dict(ms=t * 1e3, tflops=2 * fl / t / 1e12, frac=2 * fl / t / PEAK)
        x = torch.randn(B, 1024, 50, 84, device=dev).contiguous(memory_format=cl)
        wt = (torch.randn(256, 1024, 1, 1, device=dev) * 0.02).contiguous(memory_format=cl)
        fl = 2.0 * B * 50 * 84 * 1024 * 256
        with torch.no_grad():
            t = timeit(lambda: F.conv2d(x, wt), args.iters)
        res['torch_conv1x1_c4_nhwc_fwd'] = dict(ms=t * 1e3, tflops=fl / t / 1e12, frac=fl / t / PEAK)
        a, bm = torch.randn(8192, 2304, device=dev), torch.randn(2304, 256, device=dev)
        fl = 2.0 * 8192 * 2304 * 256
        t = timeit(lambda: a @ bm, args.iters)
        res['torch_sgemm_8192x2304x256'] = dict(ms=t * 1e3, tflops=fl / t / 1e12, frac=fl / t / PEAK)
    for k, v in res.items():
        print(f'{k:36s} {v["ms"]:9.3f} ms  {v["tflops"]:7.2f} TF  {100 * v["frac"]:5.1f}% of fp32 MFMA peak')
    print(json.dumps(res))


if __name__ == '__main__':
    main()
