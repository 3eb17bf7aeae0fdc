"""every dense-convolution call of the benchmark step (forward and data gradient), its duration INSIDE the
step beside the duration of the very same call replayed on its own.

    python tools/instep_vs_isolated.py [steps] > profiles/r6_instep_vs_isolated.txt

In the step: the library's per-call log (lsn_prof_launch_log: a pair of HIP events on the launch stream around each call).
Isolated, per distinct call signature, four arms on fresh random tensors:
    warm      20 calls back to back on the same tensors (what tools/ubench/conv_step measures: operands in the caches)
    cold      the same call behind a 1 GB fill each time (operands from HBM, an empty chip in front)
    warm-epi  warm without the residual / gate operands of the epilogue (only for calls that have them)
    chain     the call behind another large kernel of the library on the same stream, no idle gap (dependent-kernel hand-over)
"""
import ctypes
import os
import sys
from collections import OrderedDict

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault('HIP_FORCE_DEV_KERNARG', '1')
import torch  # noqa: E402

sys.argv, steps = sys.argv[:1], int(sys.argv[1]) if len(sys.argv) > 1 else 3
import bench  # noqa: E402
from lsnet_amd import _lib  # noqa: E402
from lsnet_amd.data import synthetic_batch  # noqa: E402
from lsnet_amd.model_zoo import build_lsnet  # noqa: E402
from lsnet_amd.parallel import DataParallelModel  # noqa: E402

dev = torch.device('cuda:0')
lib = _lib.load()
FIELDS = ('kind', 'C', 'Co', 'kh', 'kw', 'stride', 'pad', 'dil', 'relu', 'xpitch', 'n_levels', 'has_residual', 'has_gate')


def key_of(r):
    n = r.n_levels
    return tuple(getattr(r, f) for f in FIELDS) + tuple((r.B[i], r.H[i], r.W[i]) for i in range(n))


def log_steps(step, data, n):
    _lib.check(lib.lsn_prof_launch_log(1))
    for _ in range(n):
        step(data)
    torch.cuda.synchronize()
    cnt = lib.lsn_prof_read_launches(None, 0)
    arr = (_lib.ProfLaunch * cnt)()
    got = lib.lsn_prof_read_launches(arr, cnt)
    assert got == cnt
    recs = OrderedDict()
    for i in range(cnt):
        k = key_of(arr[i])
        e = recs.setdefault(k, [0, 0.0])
        e[0] += 1
        e[1] += arr[i].ms
    _lib.check(lib.lsn_prof_launch_log(0))
    return recs


def out_size(i, k, s, p, d):
    return (i + 2 * p - (d * (k - 1) + 1)) // s + 1


class Replay:
    """One logged call on fresh tensors."""

    def __init__(self, key):
        self.k = dict(zip(FIELDS, key[:len(FIELDS)]))
        self.lv = key[len(FIELDS):]
        k = self.k
        g = torch.Generator(device=dev).manual_seed(1)
        self.w = torch.randn(k['Co'] * k['kh'] * k['kw'] * k['C'], device=dev, generator=g) * (k['kh'] * k['kw'] * k['C']) ** -0.5
        nb = lib.lsn_conv2d_prepared_bytes(k['kind'], k['C'], k['Co'], k['kh'], k['kw'], k['stride'], k['pad'], k['dil'])
        assert nb > 0, key
        self.img = torch.empty(nb, dtype=torch.uint8, device=dev)
        st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
        _lib.check(lib.lsn_conv2d_prepare_weights(k['kind'], ctypes.c_void_p(self.w.data_ptr()), ctypes.c_void_p(self.img.data_ptr()),
                                                  k['C'], k['Co'], k['kh'], k['kw'], k['stride'], k['pad'], k['dil'], st))
        self.bias = torch.randn(k['Co'], device=dev, generator=g)
        self.t = []
        for (B, H, W) in self.lv:
            if k['xpitch'] != k['C']:       # row-merged stem form: W counts pixels of xpitch floats
                Ho, Wo = out_size(H, k['kh'], k['stride'], 0, 1), (W - k['C'] // k['xpitch']) // k['stride'] + 1
                n_in = B * H * W * k['xpitch']
            else:
                Ho, Wo = out_size(H, k['kh'], k['stride'], k['pad'], k['dil']), out_size(W, k['kw'], k['stride'], k['pad'], k['dil'])
                n_in = B * H * W * k['C']
            n_out = B * Ho * Wo * k['Co']
            if k['kind'] == 0:
                x, out = torch.randn(n_in, device=dev, generator=g), torch.empty(n_out, device=dev)
            else:
                x, out = torch.randn(n_out, device=dev, generator=g), torch.empty(n_in, device=dev)
            res = torch.randn(out.numel(), device=dev, generator=g) if k['has_residual'] else None
            gate = torch.randn(out.numel(), device=dev, generator=g) if k['has_gate'] else None
            self.t.append((x, out, res, gate))

    def levels(self, epi=True):
        n = len(self.lv)
        arr = (_lib.ConvLevel * n)()
        for i, ((B, H, W), (x, out, res, gate)) in enumerate(zip(self.lv, self.t)):
            L = arr[i]
            L.x, L.out, L.B, L.H, L.W = x.data_ptr(), out.data_ptr(), B, H, W
            L.residual = res.data_ptr() if (res is not None and epi) else None
            L.gate = gate.data_ptr() if (gate is not None and epi) else None
        return arr

    def call(self, lv):
        k = self.k
        st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
        if k['kind'] == 0:
            _lib.check(lib.lsn_conv2d_forward_prepared(len(self.lv), lv, ctypes.c_void_p(self.img.data_ptr()),
                                                       ctypes.c_void_p(self.bias.data_ptr()), k['C'], k['xpitch'], k['Co'], k['kh'], k['kw'],
                                                       k['stride'], k['pad'], k['dil'], k['relu'], st))
        else:
            _lib.check(lib.lsn_conv2d_backward_data_prepared(len(self.lv), lv, ctypes.c_void_p(self.img.data_ptr()), k['C'], k['Co'],
                                                             k['kh'], k['kw'], k['stride'], k['pad'], k['dil'], st))

    def flops(self):
        k, px = self.k, 0
        for (B, H, W) in self.lv:
            if k['xpitch'] != k['C']:
                px += B * out_size(H, k['kh'], k['stride'], 0, 1) * ((W - k['C'] // k['xpitch']) // k['stride'] + 1)
            else:
                px += B * out_size(H, k['kh'], k['stride'], k['pad'], k['dil']) * out_size(W, k['kw'], k['stride'], k['pad'], k['dil'])
        return 2.0 * px * k['Co'] * k['C'] * k['kh'] * k['kw']


def ev():
    return torch.cuda.Event(enable_timing=True)


def time_warm(rp, lv, reps=20):
    for _ in range(3):
        rp.call(lv)
    a, b = ev(), ev()
    a.record()
    for _ in range(reps):
        rp.call(lv)
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / reps


_flush = None


def time_cold(rp, lv, reps=6):
    global _flush
    if _flush is None:
        _flush = torch.empty(256 << 20, dtype=torch.float32, device=dev)     # 1 GB > Infinity Cache + L2s
    tot = 0.0
    for _ in range(reps):
        _flush.fill_(1.0)
        torch.cuda.synchronize()
        a, b = ev(), ev()
        a.record()
        rp.call(lv)
        b.record()
        torch.cuda.synchronize()
        tot += a.elapsed_time(b)
    return tot / reps


def time_chain(rp, lv, other, olv, reps=10):
    """rp behind `other` on the same stream, no idle in front: (time of other + rp) - (time of other alone back to back)."""
    for _ in range(2):
        other.call(olv), rp.call(lv)
    a, b, c = ev(), ev(), ev()
    a.record()
    for _ in range(reps):
        other.call(olv)
        rp.call(lv)
    b.record()
    for _ in range(reps):
        other.call(olv)
    c.record()
    torch.cuda.synchronize()
    return (a.elapsed_time(b) - b.elapsed_time(c)) / reps


def name_of(key):
    k = dict(zip(FIELDS, key[:len(FIELDS)]))
    lv = key[len(FIELDS):]
    s = f"{'fwd' if k['kind'] == 0 else 'dgrad'} {k['C']}->{k['Co']} {k['kh']}x{k['kw']}"
    if k['stride'] != 1:
        s += f' s{k["stride"]}'
    if k['xpitch'] != k['C']:
        s += f' pitch{k["xpitch"]}'
    s += f' @{lv[0][1]}x{lv[0][2]}' + (f' +{len(lv) - 1}lv' if len(lv) > 1 else '')
    s += (' relu' if k['relu'] else '') + (' +res' if k['has_residual'] else '') + (' +gate' if k['has_gate'] else '')
    return s


def main():
    torch.manual_seed(0)
    model, cfg = build_lsnet('bbox', 'r50')
    model = DataParallelModel(model.to(dev).to(memory_format=torch.channels_last).train())
    step, _ = bench.build_step(model, cfg)
    data = synthetic_batch('bbox', 2, 800, 1344, seed=1234, device=dev, channels_last=True)
    plain = bench.timed_steps(step, data, 5, 4)
    recs = log_steps(step, data, steps)
    logged = bench.timed_steps(step, data, 5, 0)
    del model, step
    torch.cuda.empty_cache()
    print(f'# LSNet R-50 bbox step, 2 x 3x800x1344: {plain * 1e3:.2f} ms/step untimed; dense-convolution calls of {steps} steps logged '
          f'({sum(v[0] for v in recs.values()) // steps} calls/step, {len(recs)} distinct signatures)')
    print('# in-step: HIP events around the call on the launch stream (incl. the hand-over from the kernel in front); isolated arms: see the '
          'header of tools/instep_vs_isolated.py; us per call; x/step = calls per step')
    print(f'{"call":58s} {"x/step":>6s} {"in-step":>8s} {"warm":>8s} {"cold":>8s} {"warm-epi":>8s} {"chain":>8s} {"TF in":>6s} {"TF warm":>7s} '
          f'{"ms/step in":>10s} {"ms/step warm":>12s}')
    # the partner of the chain arm: the head's 3x3 256 -> 256 over five levels (a typical large neighbour)
    big_key = next((k for k in recs if k[0] == 0 and k[1] == 256 and k[2] == 256 and k[3] == 3 and len(k) - len(FIELDS) == 5), None)
    big = Replay(big_key) if big_key else None
    tot = {0: [0.0] * 5, 1: [0.0] * 5}
    rows = []
    for key, (cnt, ms) in recs.items():
        rp = Replay(key)
        lv = rp.levels()
        t_in = ms / cnt * 1e3
        warm = time_warm(rp, lv) * 1e3
        cold = time_cold(rp, lv) * 1e3
        has_epi = key[FIELDS.index('has_residual')] or key[FIELDS.index('has_gate')]
        wepi = time_warm(rp, rp.levels(epi=False)) * 1e3 if has_epi else warm
        chain = time_chain(rp, lv, big, big.levels()) * 1e3 if big is not None else float('nan')
        per = cnt / steps
        fl = rp.flops()
        rows.append((t_in * per, name_of(key), per, t_in, warm, cold, wepi, chain, fl / t_in / 1e6, fl / warm / 1e6))
        t = tot[key[0]]
        for i, v in enumerate((t_in, warm, cold, wepi, chain)):
            t[i] += v * per / 1e3
        del rp
    for r in sorted(rows, reverse=True):
        print(f'{r[1]:58s} {r[2]:6.1f} {r[3]:8.1f} {r[4]:8.1f} {r[5]:8.1f} {r[6]:8.1f} {r[7]:8.1f} {r[8]:6.1f} {r[9]:7.1f} '
              f'{r[0] / 1e3:10.3f} {r[4] * r[2] / 1e3:12.3f}')
    for kind, nm in ((0, 'forward'), (1, 'data gradient')):
        t = tot[kind]
        print(f'# {nm}: in-step {t[0]:.3f} ms/step; replayed warm {t[1]:.3f}, cold {t[2]:.3f}, warm without epilogue operands {t[3]:.3f}, '
              f'chained behind a large kernel {t[4]:.3f}')
    print(f'# step with the log on: {logged * 1e3:.2f} ms/step')


if __name__ == '__main__':
    main()
