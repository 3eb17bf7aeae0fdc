"""Per-chunk cycle anatomy of the DCN forward / backward-data kernels (lsn_debug_phase_clocks)."""
import sys, os, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from lsnet_amd import _lib
from lsnet_amd.ops import get_backend

dev = torch.device('cuda:0')
cl = torch.channels_last
LEVELS = [(100, 168), (50, 84), (25, 42), (13, 21), (7, 11)]
B, C = 2, 256
torch.manual_seed(0)
be = get_backend(torch.zeros(1, device=dev))
lib = _lib.load()
w = (torch.randn(C, C, 3, 3, device=dev) * 0.02).contiguous(memory_format=cl)
xs = [torch.randn(B, C, h, ww, device=dev).contiguous(memory_format=cl) for h, ww in LEVELS]
offs = [(torch.randn(B, 18, h, ww, device=dev) * 0.5).contiguous(memory_format=cl) for h, ww in LEVELS]
msks = [torch.rand(B, 9, h, ww, device=dev).contiguous(memory_format=cl) for h, ww in LEVELS]
gos = [torch.randn(B, C, h, ww, device=dev).contiguous(memory_format=cl) for h, ww in LEVELS]
cfg = dict(stride=1, pad=1, dil=1, groups=1, dg=1, scales=[(1.0, 1.0)] * 5, pyramid=False)
need = dict(input=[True] * 5, offset=[True] * 5, mask=[True] * 5, weight=False, bias=False)
NAMES = {0: 'start', 1: 'table+sync', 2: 'loop top', 3: 'store/stage done', 4: 'sync1 done', 5: 'loads issued / mfma done(bwd)',
         6: 'mfma done / post done(bwd)', 7: 'sync2 done'}


def run(fn, block, label, flags=0):
    buf = torch.zeros(512, dtype=torch.int64, device=dev)
    for _ in range(3):
        fn()
    lib.lsn_debug_phase_clocks(None, flags)
    for _ in range(2):
        fn()
    lib.lsn_debug_phase_clocks(ctypes.c_void_p(buf.data_ptr()), block | flags)
    fn()
    torch.cuda.synchronize()
    lib.lsn_debug_phase_clocks(None, 0)
    v = buf.cpu().tolist()
    st = [(x >> 56, x & ((1 << 56) - 1)) for x in v if x != 0]
    print(f'--- {label}, block {block}: {len(st)} stamps')
    # per-phase deltas, averaged over chunks 2.. (skip first)
    from collections import defaultdict
    acc = defaultdict(list)
    for (p0, t0), (p1, t1) in zip(st[:-1], st[1:]):
        acc[(p0, p1)].append(t1 - t0)
    for k in sorted(acc):
        d = acc[k]
        d2 = d[2:] if len(d) > 4 else d
        print(f'   {NAMES.get(k[0], k[0]):32s} -> {NAMES.get(k[1], k[1]):32s} n={len(d):3d} mean {sum(d2) / len(d2):9.0f} cyc  min {min(d2):7d} max {max(d2):7d}')
    if len(st) > 2:
        print(f'   total {st[-1][1] - st[0][1]} cycles for {len(st)} stamps')


import sys
if len(sys.argv) > 1 and sys.argv[1] == 'wg':
    # weight-gradient kernel alone: 2 step top, 3 staged (VALU + LDS), 1 next tap table built, 4 barrier, 5 next loads issued, 6 MFMAs, 7 barrier
    need_w = dict(input=[False] * 5, offset=[False] * 5, mask=[False] * 5, weight=True, bias=True)
    for blk in (0, 17):
        run(lambda: be.dcn_backward(xs, offs, msks, w, gos, cfg, need_w), blk, 'weight gradient (split kernel)')
elif len(sys.argv) > 1 and sys.argv[1] == 'conv':
    # dense conv kernel: 2 loop top, 5 operand reads of the chunk landed, 6 MFMAs + staging slices done, 7 barrier
    from lsnet_amd.ops.conv import conv2d
    for name, ci, co, k, hw in (('head 3x3 256->256 P3', 256, 256, 3, (100, 168)), ('l1 1x1 64->256', 64, 256, 1, (200, 336)),
                                ('l3 3x3 256->256', 256, 256, 3, (50, 84)), ('l2 3x3 128->128', 128, 128, 3, (100, 168))):
        xc = torch.randn(2, ci, *hw, device=dev).contiguous(memory_format=cl)
        wc = (torch.randn(co, ci, k, k, device=dev) * 0.05).contiguous(memory_format=cl)
        with torch.no_grad():
            run(lambda: conv2d(xc, wc, None, 1, k // 2), 17, 'conv forward ' + name)
elif len(sys.argv) > 1 and sys.argv[1] == 'wgab':
    # weight gradient inside a full backward call (the backward-data pass leaves its sampling table for it):
    # bit 24 = ignore that table, bit 25 = scalar loads
    need_all = dict(input=[True] * 5, offset=[True] * 5, mask=[True] * 5, weight=True, bias=True)
    for name, fl in (('table + 8-byte loads', 0), ('computed taps', 1 << 24), ('scalar loads', 1 << 25)):
        lib.lsn_debug_phase_clocks(None, fl)
        for _ in range(3):
            be.dcn_backward(xs, offs, msks, w, gos, cfg, need_all)
        _lib.prof_enable(True)
        for _ in range(10):
            be.dcn_backward(xs, offs, msks, w, gos, cfg, need_all)
        torch.cuda.synchronize()
        pr = _lib.prof_read()
        _lib.prof_enable(False)
        lib.lsn_debug_phase_clocks(None, 0)
        print(f'== {name}: ' + ', '.join(f"{k} {v['total_ms'] / max(v['launches'], 1):.3f} ms" for k, v in pr.items()))
        run(lambda: be.dcn_backward(xs, offs, msks, w, gos, cfg, need_all), 17, 'weight gradient, ' + name, fl)
elif len(sys.argv) > 1 and sys.argv[1] == 'bwd1':
    # split kernels: 2 loop top, 4 slab landed (barrier), 5 MFMAs done, 3 barrier + next slab issued, 6 epilogue done
    for blk in (0, 300, 600):
        run(lambda: be.dcn_backward(xs, offs, msks, w, gos, cfg, need), blk, 'backward-data (split kernel)')
elif len(sys.argv) > 1 and sys.argv[1] == 'bwd':
    for name, fl in (('full', 0), ('no atomics', 1 << 26), ('no offset/mask grads', 1 << 27),
                     ('neither atomics nor offset grads', 3 << 26)):
        run(lambda: be.dcn_backward(xs, offs, msks, w, gos, cfg, need), 300, f'backward-data {name}', fl)
elif len(sys.argv) > 1 and sys.argv[1] == 'ablate':
    for name, fl in (('full', 0), ('no issue', 2 << 20), ('no issue, commit VALU only', 6 << 20), ('no issue, commit LDS only', 10 << 20)):
        run(lambda: be.dcn_forward(xs, offs, msks, w, None, cfg, LEVELS), 300, 'forward ' + name, fl)
else:
    for blk in (0, 300):
        run(lambda: be.dcn_forward(xs, offs, msks, w, None, cfg, LEVELS), blk, 'forward')
        run(lambda: be.dcn_backward(xs, offs, msks, w, gos, cfg, need), blk, 'backward-data')
