"""Parity of the HIP path (through the C ABI) against the CPU oracle on seeded inputs.

Tolerances: north_star asks for 1e-3 relative on fp32 conv/loss.  The default arithmetic of the contractions is
'bf16x6' (fp32-equivalent split products on the bf16 matrix pipe, include/lsnet_hip.h); 'bf16x3' and exact fp32 MFMA
are run as well.  All of them are held to 1e-4 of the output range against the oracle (measured ~1e-6; the oracle's own
fp32 summation order is part of that), and the default mode to 1e-6 against the exact-fp32 kernels
(test_split6_matches_exact_fp32).  Index outputs (NMS keep) must be identical."""
import ctypes

import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

from oracle import oracle_py as orc  # noqa: E402  (the checker)

TOL = 1e-4


def _dev():
    assert torch.cuda.is_available(), 'gpu tests need the MI355X'
    return torch.device('cuda:0')


def _err(a, b):
    a, b = a.detach().float().cpu(), b.detach().float().cpu()
    assert a.shape == b.shape, (a.shape, b.shape)
    scale = max(b.abs().max().item(), 1e-12)
    return (a - b).abs().max().item() / scale


def _report(name, got, ref):
    e = _err(got, ref)
    if not (e < TOL):
        d = (got.detach().float().cpu() - ref.detach().float().cpu()).abs()
        idx = np.unravel_index(int(d.argmax()), d.shape)
        print(f'[{name}] rel err {e:.3e}; worst at {idx}: got {got.detach().cpu()[idx].item():.6f} '
              f'ref {ref[idx].item():.6f}; ref absmax {ref.abs().max().item():.4f}; '
              f'frac bad {(d > TOL * ref.abs().max()).float().mean().item():.4f}')
    return e


@pytest.mark.parametrize('variant', [0, 1])
def test_mfma_fragment_maps(variant):
    """Asymmetric operands: catches a transposed C/D map (guide rule 16)."""
    from lsnet_amd.ops import get_backend
    dev = _dev()
    torch.manual_seed(0)
    for M, N, K in [(32, 32, 8), (64, 96, 36), (48, 40, 20)]:
        A, B = torch.randn(M, K, device=dev), torch.randn(K, N, device=dev)
        D = get_backend(A).selftest_mfma(A, B, variant)
        ref = (A.double() @ B.double()).float()
        assert _err(D, ref) < 1e-5, (variant, M, N, K)


DCN_CASES = [
    # name, C, Co, groups, dg, stride, pad, dil, mask, (Hs,Ws), dst or None, bias
    dict(name='v2_small', C=16, Co=24, hw=(13, 21)),
    dict(name='v2_s2_g4_dg2', C=16, Co=24, groups=4, dg=2, stride=2, hw=(13, 21)),
    dict(name='v2_dil2', C=16, Co=24, dil=2, pad=2, hw=(13, 21)),
    dict(name='v2_c40_co72', C=40, Co=72, hw=(9, 14)),             # channel tails (C % 32 != 0)
    dict(name='v2_c6_odd', C=6, Co=10, hw=(7, 9)),                 # C % 4 != 0: scalar weight path
    dict(name='v2_head_p6', C=256, Co=256, hw=(13, 21)),           # the LSHead tower shape at P6
    dict(name='v2_wide', C=64, Co=320, hw=(10, 12)),               # Co > 256: two co blocks
    dict(name='v1', C=32, Co=48, mask=False, hw=(13, 21)),
    dict(name='pyr_down', C=32, Co=32, mask=False, hw=(25, 42), dst=(13, 21)),
    dict(name='pyr_up', C=32, Co=32, mask=False, hw=(7, 11), dst=(13, 21)),
    dict(name='pyr_head', C=256, Co=256, mask=False, hw=(25, 42), dst=(13, 21)),
    dict(name='v2_g2', C=64, Co=128, groups=2, hw=(8, 8)),
    dict(name='v2_dg4', C=128, Co=64, dg=4, hw=(6, 10)),
    dict(name='v2_dg2_wide', C=512, Co=128, dg=2, hw=(9, 12)),     # two deformable groups of 256 channels (dcn_mm_kernels.h)
    dict(name='v2_dg2_co256', C=128, Co=256, dg=2, hw=(9, 12)),    # 64-channel deformable groups under the 256-co weight-gradient tile
    # backbone-shaped calls of BASELINE configs 3 / 4: R-101-DCN conv2 (g = 1, first block of a stage
    # stride 2) and X-101-64x4d-DCN conv2 (g = 64: 8 / 16 / 32 channels per group, no bias)
    dict(name='r101_l2_s2', C=128, Co=128, stride=2, hw=(50, 84), bias=False),
    dict(name='r101_l3', C=256, Co=256, hw=(25, 42), bias=False),
    dict(name='r101_l4_s2', C=512, Co=512, stride=2, hw=(26, 42), bias=False),
    dict(name='x101_l2_s2', C=512, Co=512, groups=64, stride=2, hw=(50, 84), bias=False),
    dict(name='x101_l3', C=1024, Co=1024, groups=64, hw=(13, 21), bias=False),
    dict(name='x101_l4', C=2048, Co=2048, groups=64, hw=(7, 11), bias=False),
    # round 6 (dcn_grouped_kernels.h): bias, two deformable groups, a ragged last pixel tile, 16 / 4 channels per group with one
    # 64-channel span (4 per group: the new weight gradient, the general forward)
    dict(name='g64_c8_dg2_bias', C=512, Co=512, groups=64, dg=2, hw=(9, 14)),
    dict(name='g4_c16', C=64, Co=64, groups=4, hw=(9, 12), mask=False),
    dict(name='g16_c4', C=64, Co=64, groups=16, hw=(7, 9)),
    # Res2Net-50/101 26w x 4s (the headline 53.5-AP backbone): per-scale widths 52 / 104 / 208, the first block of a stage
    # with stride 2 (round 4: the backbone's gradient fixture localised a device-only error to these calls)
    dict(name='r2_l2_s2', C=52, Co=52, stride=2, hw=(24, 32), bias=False),
    dict(name='r2_l3_s2', C=104, Co=104, stride=2, hw=(12, 16), bias=False),
    dict(name='r2_l4_s2', C=208, Co=208, stride=2, hw=(6, 8), bias=False),
    dict(name='r2_l4', C=208, Co=208, hw=(3, 4), bias=False),
]


def _make(case, dev, seed=0):
    g = torch.Generator().manual_seed(seed)
    B, C, Co = 2, case['C'], case['Co']
    groups, dg = case.get('groups', 1), case.get('dg', 1)
    stride, pad, dil = case.get('stride', 1), case.get('pad', 1), case.get('dil', 1)
    Hs, Ws = case['hw']
    if 'dst' in case:
        Ho, Wo = case['dst']
        sh, sw = Hs / Ho, Ws / Wo
        pyramid = True
    else:
        Ho, Wo = orc.out_size(Hs, 3, stride, pad, dil), orc.out_size(Ws, 3, stride, pad, dil)
        sh = sw = 1.0
        pyramid = False
    has_mask = case.get('mask', True)
    x = torch.randn(B, C, Hs, Ws, generator=g)
    w = torch.randn(Co, C // groups, 3, 3, generator=g) * (1.0 / (3 * (C // groups) ** 0.5))
    b = torch.randn(Co, generator=g) if (has_mask and case.get('bias', True)) else None
    off = torch.rand(B, dg * 18, Ho, Wo, generator=g) * 6 - 3  # reaches outside the map
    mask = torch.rand(B, dg * 9, Ho, Wo, generator=g) if has_mask else None
    go = torch.randn(B, Co, Ho, Wo, generator=g)
    cfg = dict(stride=stride, pad=pad, dil=dil, groups=groups, dg=dg, sh=sh, sw=sw, pyramid=pyramid,
               out_hw=(Ho, Wo))
    return x, w, b, off, mask, go, cfg


def _to(t, dev, cl):
    if t is None:
        return None
    t = t.to(dev)
    return t.contiguous(memory_format=torch.channels_last) if cl else t


KERNEL_CHOICES = {
    # name: (math mode, debug word).  Bit 23: atomic scatter instead of the anchor-list gather (lsn_debug_phase_clocks).  The split
    # weight-gradient kernel reads bits 24 / 25 as "compute the sampling table instead of copying the launch-wide one" / "scalar
    # loads": the one atomic choice below also covers those fallback paths.  (Round 6: the windowed scatter kernels and the
    # pipelined fp32 forward are gone, 7 -> 5 choices; the exact mode takes the atomic-free gather as well.)
    'default': ('bf16x6', 0),                       # fp32-equivalent products, atomic-free grad_input
    'x6_first_gemms': ('bf16x6', 1 << 28),          # bit 28: dcn_kernels.h GEMMs where dcn_mm_kernels.h would serve
    'x3_gather': ('bf16x3', 0),
    'x3_atomic_fallbacks': ('bf16x3', (1 << 23) | (1 << 24) | (1 << 25)),
    'math_fp32': ('fp32', 0),                       # exact fp32: fp32 MFMA forward / weight gradient, fmaf column gradients + gather
}


@pytest.fixture(params=list(KERNEL_CHOICES))
def dcn_kernel_choice(request):
    """The parity cases run under the defaults, with every other backward-data kernel forced, in the 3-product split
    mode and with exact fp32 MFMA everywhere."""
    from lsnet_amd import _lib
    mode, flag = KERNEL_CHOICES[request.param]
    _lib.load().lsn_debug_phase_clocks(None, flag)
    old = _lib.get_math_mode()
    _lib.set_math_mode(mode)
    yield request.param
    _lib.set_math_mode(old)
    _lib.load().lsn_debug_phase_clocks(None, 0)


@pytest.mark.parametrize('layout', ['nchw', 'nhwc'])
@pytest.mark.parametrize('case', DCN_CASES, ids=[c['name'] for c in DCN_CASES])
def test_dcn_forward_backward(case, layout, dcn_kernel_choice):
    from lsnet_amd import ops
    dev = _dev()
    cl = layout == 'nhwc'
    x, w, b, off, mask, go, cfg = _make(case, dev)
    ref = orc.deform_conv_forward(x, w, b, off, mask, cfg['stride'], cfg['pad'], cfg['dil'], cfg['groups'],
                                  cfg['dg'], cfg['sh'], cfg['sw'], out_hw=cfg['out_hw'])
    gref = orc.deform_conv_backward(x, w, off, mask, go, cfg['stride'], cfg['pad'], cfg['dil'], cfg['groups'],
                                    cfg['dg'], cfg['sh'], cfg['sw'])
    xd, wd, od, md = _to(x, dev, cl).requires_grad_(), _to(w, dev, cl).requires_grad_(), \
        _to(off, dev, cl).requires_grad_(), _to(mask, dev, cl)
    bd = None if b is None else b.to(dev).requires_grad_()
    if md is not None:
        md.requires_grad_()
    out = ops.dcn_multi([xd], [od], [md], wd, bd, cfg['stride'], cfg['pad'], cfg['dil'], cfg['groups'],
                        cfg['dg'], scales=[(cfg['sh'], cfg['sw'])], pyramid=cfg['pyramid'])[0]
    torch.cuda.synchronize()
    errs = {'out': _report(case['name'] + '/out', out, ref)}
    wrt = [t for t in (xd, od, md, wd, bd) if t is not None]
    grads = torch.autograd.grad(out, wrt, _to(go, dev, cl))
    torch.cuda.synchronize()
    names = ['gx', 'goff'] + (['gmask'] if md is not None else []) + ['gw'] + (['gb'] if bd is not None else [])
    for n, gt in zip(names, grads):
        errs[n] = _report(case['name'] + '/' + n, gt, gref[n])
    print(case['name'], layout, {k: f'{v:.1e}' for k, v in errs.items()})
    assert all(e < TOL for e in errs.values()), errs


def test_dcn_multi_level_launch_equals_single():
    """Five levels with shared weights in ONE launch == five launches (and == oracle)."""
    from lsnet_amd import ops
    dev = _dev()
    torch.manual_seed(1)
    C = Co = 64
    sizes = [(20, 28), (10, 14), (5, 7), (3, 4), (2, 2)]
    w = (torch.randn(Co, C, 3, 3, device=dev) * 0.05).requires_grad_()
    b = torch.randn(Co, device=dev).requires_grad_()
    xs = [torch.randn(2, C, h, ww, device=dev).contiguous(memory_format=torch.channels_last).requires_grad_()
          for h, ww in sizes]
    offs = [(torch.rand(2, 18, h, ww, device=dev) * 4 - 2).requires_grad_() for h, ww in sizes]
    msks = [torch.rand(2, 9, h, ww, device=dev).requires_grad_() for h, ww in sizes]
    outs = ops.dcn_multi(xs, offs, msks, w, b, 1, 1, 1)
    loss = sum((o * (i + 1)).square().sum() for i, o in enumerate(outs))
    g_multi = torch.autograd.grad(loss, [w, b] + xs + offs + msks)
    singles = [ops.modulated_deform_conv(x, o, m, w, b, 1, 1, 1) for x, o, m in zip(xs, offs, msks)]
    loss1 = sum((o * (i + 1)).square().sum() for i, o in enumerate(singles))
    g_single = torch.autograd.grad(loss1, [w, b] + xs + offs + msks)
    for a, c in zip(outs, singles):
        assert _err(a, c.cpu()) < 1e-6
    for a, c in zip(g_multi, g_single):
        assert _err(a, c.cpu()) < 1e-4
    ref = orc.deform_conv_forward(xs[2].detach().cpu(), w.detach().cpu(), b.detach().cpu(), offs[2].detach().cpu(),
                                  msks[2].detach().cpu(), 1, 1, 1)
    assert _err(outs[2], ref) < TOL


def test_reference_named_entry_points():
    """The one-to-one C-ABI replacements of deform_conv_ext (NCHW, reference argument order)."""
    from lsnet_amd import _lib
    lib = _lib.load()
    dev = _dev()
    case = dict(name='abi', C=32, Co=48, mask=True, hw=(11, 13))
    x, w, b, off, mask, go, cfg = _make(case, dev, seed=5)
    B, C, H, W = x.shape
    Co = w.shape[0]
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    p = lambda t: ctypes.c_void_p(t.data_ptr())
    xd, wd, bd, od, md, gd = [t.to(dev).contiguous() for t in (x, w, b, off, mask, go)]
    out = torch.empty(B, Co, H, W, device=dev)
    _lib.check(lib.lsn_modulated_deform_conv_forward(p(xd), p(wd), p(bd), p(od), p(md), p(out), B, C, H, W, Co,
                                                     3, 3, 1, 1, 1, 1, 1, 1, 1, 1, 1, st))
    ref = orc.deform_conv_forward(x, w, b, off, mask, 1, 1, 1)
    assert _report('abi/mdcn_fwd', out, ref) < TOL
    gx, gw, gb = torch.empty_like(xd), torch.empty_like(wd), torch.empty_like(bd)
    goff, gm = torch.empty_like(od), torch.empty_like(md)
    _lib.check(lib.lsn_modulated_deform_conv_backward(p(xd), p(wd), p(bd), p(od), p(md), p(gx), p(gw), p(gb),
                                                      p(goff), p(gm), p(gd), B, C, H, W, Co, 3, 3, 1, 1, 1, 1, 1,
                                                      1, 1, 1, 1, st))
    gref = orc.deform_conv_backward(x, w, off, mask, go, 1, 1, 1)
    for n, t in (('gx', gx), ('gw', gw), ('gb', gb), ('goff', goff), ('gmask', gm)):
        assert _report('abi/mdcn_' + n, t, gref[n]) < TOL
    # DCNv1: forward / backward_input / backward_parameters (W-before-H argument order)
    out1 = torch.empty(B, Co, H, W, device=dev)
    _lib.check(lib.lsn_deform_conv_forward(p(xd), p(wd), p(od), p(out1), B, C, H, W, Co, 3, 3, 1, 1, 1, 1, 1, 1,
                                           1, 1, B, st))
    assert _report('abi/dcn1_fwd', out1, orc.deform_conv_forward(x, w, None, off, None, 1, 1, 1)) < TOL
    g1 = orc.deform_conv_backward(x, w, off, None, go, 1, 1, 1)
    _lib.check(lib.lsn_deform_conv_backward_input(p(xd), p(od), p(gd), p(gx), p(goff), p(wd), B, C, H, W, Co, 3,
                                                  3, 1, 1, 1, 1, 1, 1, 1, 1, B, st))
    _lib.check(lib.lsn_deform_conv_backward_parameters(p(xd), p(od), p(gd), p(gw), B, C, H, W, Co, 3, 3, 1, 1, 1,
                                                       1, 1, 1, 1, 1, ctypes.c_float(1.0), B, st))
    for n, t in (('gx', gx), ('goff', goff), ('gw', gw)):
        assert _report('abi/dcn1_' + n, t, g1[n]) < TOL
    # pyramid: offset grid 7x9 over the 11x13 source
    Ho, Wo = 7, 9
    offp = (torch.rand(B, 18, Ho, Wo) * 4 - 2)
    gop = torch.randn(B, Co, Ho, Wo)
    sh, sw = H / Ho, W / Wo
    outp = torch.empty(B, Co, Ho, Wo, device=dev)
    offpd, gopd = offp.to(dev), gop.to(dev)
    _lib.check(lib.lsn_pyramid_deform_conv_forward(p(xd), p(wd), p(offpd), p(outp), B, C, H, W, Co, Ho, Wo, 3, 3,
                                                   1, 1, 1, 1, 1, 1, ctypes.c_float(sw), ctypes.c_float(sh), 1, 1,
                                                   B, st))
    assert _report('abi/pyr_fwd', outp, orc.deform_conv_forward(x, w, None, offp, None, 1, 1, 1, 1, 1, sh, sw,
                                                                out_hw=(Ho, Wo))) < TOL
    gp = orc.deform_conv_backward(x, w, offp, None, gop, 1, 1, 1, 1, 1, sh, sw)
    goffp = torch.empty_like(offpd)
    _lib.check(lib.lsn_pyramid_deform_conv_backward_input(p(xd), p(offpd), p(gopd), p(gx), p(goffp), p(wd), B, C,
                                                          H, W, Co, Ho, Wo, 3, 3, 1, 1, 1, 1, 1, 1,
                                                          ctypes.c_float(sw), ctypes.c_float(sh), 1, 1, B, st))
    _lib.check(lib.lsn_pyramid_deform_conv_backward_parameters(p(xd), p(offpd), p(gopd), p(gw), B, C, H, W, Co,
                                                               Ho, Wo, 3, 3, 1, 1, 1, 1, 1, 1, ctypes.c_float(sw),
                                                               ctypes.c_float(sh), 1, 1, ctypes.c_float(1.0), B,
                                                               st))
    torch.cuda.synchronize()
    for n, t in (('gx', gx), ('goff', goffp), ('gw', gw)):
        assert _report('abi/pyr_' + n, t, gp[n]) < TOL


def test_focal_loss_parity():
    from lsnet_amd import ops
    dev = _dev()
    torch.manual_seed(2)
    for N, C in [(1000, 80), (33, 7), (4096, 80)]:
        lg = torch.randn(N, C) * 4
        tg = torch.randint(0, C + 1, (N,))
        d = torch.randn(N, C)
        ref = orc.sigmoid_focal_loss_forward(lg, tg, 2.0, 0.25)
        gref = orc.sigmoid_focal_loss_backward(lg, tg, d, 2.0, 0.25)
        x = lg.to(dev).requires_grad_()
        out = ops.sigmoid_focal_loss(x, tg.to(dev), 2.0, 0.25)
        g, = torch.autograd.grad(out, x, d.to(dev))
        assert torch.allclose(out.cpu(), ref, rtol=1e-4, atol=1e-6)
        assert torch.allclose(g.cpu(), gref, rtol=1e-4, atol=1e-6)
        w = torch.rand(N)
        s = ops.sigmoid_focal_loss_sum(x, tg.to(dev), w.to(dev), 2.0, 0.25)
        assert abs(s.item() - (ref * w[:, None]).sum().item()) < 1e-4 * abs((ref * w[:, None]).sum().item()) + 1e-5
        gs, = torch.autograd.grad(s * 0.37, x)
        assert torch.allclose(gs.cpu(), orc.sigmoid_focal_loss_backward(lg, tg, (w[:, None] * 0.37).expand(N, C)
                                                                         .contiguous(), 2.0, 0.25),
                              rtol=1e-4, atol=1e-6)


def test_per_level_loss_sums_match_the_level_by_level_form():
    """lsn_sigmoid_focal_loss_level_sums / lsn_level_sums over LSHead's concatenated rows (B images x N_all rows, levels back to
    back per image) against what the reference's per-level calls compute (focal_loss.py:74-116 per level slice, then .sum()):
    values to fp32 summation order, gradients element-wise, the same bits on every run."""
    from lsnet_amd import ops
    from lsnet_amd.ops.focal import level_sums, sigmoid_focal_loss_level_sums
    dev = _dev()
    torch.manual_seed(12)
    for B, num_level, C in [(2, [1200, 300, 80, 20, 6], 80), (3, [7, 5], 11), (1, [4000], 80), (8, [33, 17, 9, 5, 3, 2, 1, 1], 4)]:
        nall = sum(num_level)
        lg = (torch.randn(B * nall, C) * 4).to(dev).requires_grad_()
        tg = torch.randint(0, C + 1, (B * nall,), device=dev)
        w = torch.rand(B * nall, device=dev)
        up = torch.randn(len(num_level), device=dev)
        got = sigmoid_focal_loss_level_sums(lg, tg, w, B, num_level, 2.0, 0.25)
        assert got.shape == (len(num_level),)
        ggot, = torch.autograd.grad(got, lg, up)
        again = sigmoid_focal_loss_level_sums(lg, tg, w, B, num_level, 2.0, 0.25)
        assert torch.equal(got, again)
        x2 = lg.detach().clone().requires_grad_()
        ref, o = [], 0
        for n in num_level:
            sl = torch.cat([torch.arange(b * nall + o, b * nall + o + n, device=dev) for b in range(B)])
            ref.append(ops.sigmoid_focal_loss_sum(x2[sl], tg[sl], w[sl], 2.0, 0.25))
            o += n
        ref = torch.stack(ref)
        gref, = torch.autograd.grad(ref, x2, up)
        assert torch.allclose(got, ref, rtol=2e-6, atol=1e-6), (got, ref)
        assert torch.allclose(ggot, gref, rtol=1e-6, atol=1e-9)
        # no weights
        g0 = sigmoid_focal_loss_level_sums(lg, tg, None, B, num_level, 2.0, 0.25)
        r0 = ops.sigmoid_focal_loss(lg.detach(), tg, 2.0, 0.25).sum(1).reshape(B, nall)
        r0 = torch.stack([c.sum() for c in torch.split(r0, num_level, dim=1)])
        assert torch.allclose(g0, r0, rtol=2e-6, atol=1e-6)

        rows = torch.randn(B * nall, device=dev).requires_grad_()
        s = level_sums(rows, B, num_level)
        sref = torch.stack([c.double().sum() for c in torch.split(rows.detach().reshape(B, nall), num_level, dim=1)])
        assert torch.allclose(s.double(), sref, rtol=1e-6, atol=1e-5)
        gr, = torch.autograd.grad(s, rows, up)
        exp = torch.cat([up[l].expand(n) for l, n in enumerate(num_level)]).repeat(B)
        assert torch.equal(gr, exp)
    with pytest.raises(RuntimeError):
        level_sums(torch.zeros(10, device=dev), 1, [10, 0])       # an empty level


def test_relu_gate_multi_reads_sliced_gradients_where_they_lie():
    """lsn_relu_gate_multi: the ReLU gates of all maps of a multi-level convolution in one launch; a gradient that is a level
    sliced out of the head's concatenated (B, N_all, C) tensor (images N_all * C floats apart) is gated without a copy.  Through
    the convolution: conv2d_multi(relu=True) followed by the head's concatenation gives the gradients of the level-by-level form."""
    from lsnet_amd.ops import conv as cv
    dev = _dev()
    torch.manual_seed(14)
    shapes = [(2, 64, 9, 7), (2, 64, 5, 4), (2, 64, 3, 2), (2, 64, 1, 1)]
    ys = [torch.randn(s, device=dev).contiguous(memory_format=torch.channels_last) for s in shapes]
    B, C = 2, 64
    nall = sum(h * w for _, _, h, w in shapes)
    cat = torch.randn(B, nall, C, device=dev)
    gys, o = [], 0
    for _, _, h, w in shapes:
        gys.append(cat[:, o:o + h * w].reshape(B, h, w, C).permute(0, 3, 1, 2))
        o += h * w
    assert cv._images_apart(gys[0]) == nall * C and not gys[0].is_contiguous(memory_format=torch.channels_last)
    got = cv.relu_gate_multi(gys, ys)
    for g, gy, y in zip(got, gys, ys):
        assert g.is_contiguous(memory_format=torch.channels_last)
        assert torch.equal(g, torch.where(y > 0, gy, torch.zeros_like(gy)))
    # dense gradients, odd layouts (fallback), and the whole thing behind a convolution
    got = cv.relu_gate_multi([g.contiguous(memory_format=torch.channels_last) for g in gys], ys)
    assert all(torch.equal(g, torch.where(y > 0, gy, torch.zeros_like(gy))) for g, gy, y in zip(got, gys, ys))
    got = cv.relu_gate_multi([g.contiguous() for g in gys], ys)
    assert all(torch.equal(g, torch.where(y > 0, gy, torch.zeros_like(gy))) for g, gy, y in zip(got, gys, ys))

    w = (torch.randn(64, 64, 3, 3, device=dev) * 0.05).contiguous(memory_format=torch.channels_last).requires_grad_()
    b = torch.randn(64, device=dev).requires_grad_()
    xs = [torch.randn(s, device=dev).contiguous(memory_format=torch.channels_last).requires_grad_() for s in shapes]
    up = torch.randn(B, nall, C, device=dev)

    def run(multi):
        outs = cv.conv2d_multi(xs, w, b, 1, 1, relu=True) if multi else \
            [torch.relu(cv.conv2d_multi([x], w, b, 1, 1)[0]) for x in xs]
        flat = torch.cat([t.permute(0, 2, 3, 1).reshape(B, -1, C) for t in outs], dim=1)
        return torch.autograd.grad(flat, xs + [w, b], up)
    for a, r in zip(run(True), run(False)):
        assert torch.allclose(a, r, rtol=1e-4, atol=1e-5), float((a - r).abs().max())


def test_topk_columns_matches_torch():
    """lsn_topk_columns against torch.topk on the CPU (the reference's call, centroid_assigner.py:74 and
    atss_assigner.py:103-111): same values in the same order, same rows; equal values come out by ascending row."""
    from lsnet_amd.core.assigners import topk_columns
    dev = _dev()
    g = torch.Generator().manual_seed(11)
    levels = [16800, 4200, 1050, 273, 77]          # the points of a 800 x 1344 image per FPN level
    for P, G, k, segs, largest in [(22400, 7, 3, None, False), (22400, 1, 1, None, False), (22400, 13, 9, levels, False),
                                   (5000, 5, 9, [2500, 2500], True), (40000, 3, 4, None, False), (300, 64, 9, None, True)]:
        x = torch.randperm(P * G, generator=g).float().reshape(P, G)         # distinct values: one right answer
        segments = None
        if segs is not None:
            segments, o = [], 0
            for n in segs:
                segments.append((o, n))
                o += n
        v_ref, i_ref = topk_columns(x, k, segments, largest)                    # torch on the CPU
        v, i = topk_columns(x.to(dev), k, segments, largest)
        assert torch.equal(v.cpu(), v_ref) and torch.equal(i.cpu(), i_ref), (P, G, k, largest)
    # a strided matrix (a column slice of a wider one)
    wide = torch.randperm(3000 * 12, generator=g).float().reshape(3000, 12)
    v, i = topk_columns(wide.to(dev)[:, 2:9], 5)
    v_ref, i_ref = wide[:, 2:9].topk(5, dim=0, largest=False)
    assert torch.equal(v.cpu(), v_ref) and torch.equal(i.cpu(), i_ref)
    # ties: ascending row among equal values; INF entries (the centroid assigner's other-level points) and NaN go last
    x = torch.full((1000, 2), 1e8)
    x[[5, 900, 17, 400], 0] = torch.tensor([2.0, 1.0, 2.0, 2.0])
    x[[3, 4], 1] = torch.tensor([float('nan'), 7.0])
    v, i = topk_columns(x.to(dev), 3)
    assert i[:, 0].cpu().tolist() == [900, 5, 17] and v[:, 0].cpu().tolist() == [1.0, 2.0, 2.0]
    assert i[:, 1].cpu().tolist() == [4, 0, 1] and v[:, 1].cpu().tolist() == [7.0, 1e8, 1e8]
    v, i = topk_columns(x.to(dev), 2, largest=True)
    assert i[:, 1].cpu().tolist()[0] == 3 and torch.isnan(v[0, 1])             # NaN is the largest value, as in torch


def test_offset_scale_chain_bitwise():
    """lsn_offset_chain_forward / _backward against the operator sequence they replace (lsnet_head.py:622-638: three
    multiplications per level, autograd's additions behind them): the same bits, for slices of a concatenated tensor too."""
    from lsnet_amd.ops.dcn import offset_scale_chain
    dev = _dev()
    g = torch.Generator().manual_seed(5)
    shapes = [(25, 42), (13, 21), (7, 11)]
    B, C = 2, 18
    n_all = sum(h * w for h, w in shapes)
    flat = torch.randn(B, n_all, C, generator=g).to(dev).requires_grad_()
    mults = [((13 / 25, 21 / 42), (25 / 13, 2.0), (0.28, 11 / 42)), ((1.0, 1.0), (25 / 13, 42 / 21), (7 / 13, 11 / 21)),
             ((1.0, 1.0), (13 / 7, 21 / 11), (25 / 13, 2.0))]
    ws = [[torch.randn(B, C, h, w, generator=g).to(dev) for _ in range(3)] for h, w in shapes]

    def run(fused):
        flat.grad = None
        offs, o = [], 0
        for h, w in shapes:
            offs.append(flat[:, o:o + h * w].reshape(B, h, w, C).permute(0, 3, 1, 2))
            o += h * w
        if fused:
            first, second = offset_scale_chain(offs, mults, copies=2)     # two consumers, each with its own handles
        else:
            res = []
            for off, m in zip(offs, mults):
                cur, trio = off, []
                for sh, sw in m:
                    cur = cur * off.new_tensor([sh, sw]).repeat(C // 2).view(1, -1, 1, 1)
                    trio.append(cur)
                res.append(trio)
            first = second = res
        loss = sum((t * w).sum() for trio, wt in zip(first, ws) for t, w in zip(trio[:2], wt[:2])) + (first[0][2] * ws[0][2]).sum()
        loss = loss + sum((t * w.flip(0)).sum() for trio, wt in zip(second, ws) for t, w in zip(trio[1:], wt[1:]))
        loss.backward()    # (first consumer: the third field of levels 1 and 2 unused; second: no first field)
        return [t.detach().clone() for trio in first for t in trio], flat.grad.clone()

    o_ref, g_ref = run(False)
    o, gr = run(True)
    assert all(torch.equal(a, b) for a, b in zip(o, o_ref))
    assert torch.equal(gr, g_ref)


def test_nms_bit_exact():
    from lsnet_amd import ops
    dev = _dev()
    # the reference's known-answer vector (tests/test_ops/test_nms.py:18-24)
    dets = torch.tensor([[49.1, 32.4, 51.0, 35.9, 0.1], [49.3, 32.9, 51.0, 35.3, 0.05],
                         [35.3, 11.5, 39.9, 14.5, 0.9], [35.2, 11.7, 39.7, 15.7, 0.3]])
    kept, inds = ops.nms(dets.to(dev), 0.6)
    assert inds.tolist() == [2, 0]
    assert torch.equal(kept.cpu(), dets[[2, 0]])
    assert ops.nms(torch.zeros(0, 5, device=dev), 0.5)[1].numel() == 0
    g = torch.Generator().manual_seed(3)
    for n in (1, 63, 64, 65, 700, 3000):
        xy = torch.rand(n, 2, generator=g) * 200
        wh = torch.rand(n, 2, generator=g) * 60 + 1
        sc = torch.rand(n, generator=g)          # distinct scores: the sort order is unambiguous
        d = torch.cat([xy, xy + wh, sc[:, None]], 1)
        ref = orc.nms(d, 0.6)
        _, inds = ops.nms(d.to(dev), 0.6)
        assert inds.cpu().tolist() == ref.tolist(), n
    # per-class batching with the coordinate-offset trick
    labels = torch.randint(0, 5, (700,), generator=g)
    boxes, scores = d[:700, :4], d[:700, 4]
    dets_g, keep_g = ops.batched_nms(boxes.to(dev), scores.to(dev), labels.to(dev), dict(type='nms', iou_thr=0.5))
    offs = labels.float() * (boxes.max() + 1)
    ref = orc.nms(torch.cat([boxes + offs[:, None], scores[:, None]], 1), 0.5)
    assert keep_g.cpu().tolist() == ref.tolist()


@pytest.mark.parametrize('layout', ['nchw', 'nhwc'])
def test_dcn_pack_fused_offset_mask_logits(layout):
    """ModulatedDeformConvPack hands conv_offset's output to the op as one tensor (mask logits ->
    sigmoid inside the kernels).  Must equal the explicit chunk / sigmoid composition of the
    reference (deform_conv.py:527-534), forward and all gradients."""
    from lsnet_amd import ops
    dev = _dev()
    torch.manual_seed(3)
    pack = ops.ModulatedDeformConvPack(32, 48, 3, 1, 1).to(dev)
    torch.nn.init.normal_(pack.conv_offset.weight, std=0.05)
    torch.nn.init.normal_(pack.conv_offset.bias, std=0.5)
    x = torch.randn(2, 32, 17, 23, device=dev)
    if layout == 'nhwc':
        pack = pack.to(memory_format=torch.channels_last)
        x = x.contiguous(memory_format=torch.channels_last)
    x.requires_grad_()
    out = pack(x)
    go = torch.randn_like(out)
    g1 = torch.autograd.grad(out, [x] + list(pack.parameters()), go)
    om = pack.conv_offset(x)
    out2 = ops.modulated_deform_conv(x, om[:, :18], torch.sigmoid(om[:, 18:]), pack.weight, pack.bias, 1, 1, 1)
    g2 = torch.autograd.grad(out2, [x] + list(pack.parameters()), go)
    assert _err(out, out2.detach().cpu()) < 1e-6
    for a, b in zip(g1, g2):
        assert _err(a, b.detach().cpu()) < 1e-5
    # and against the oracle
    omc = om.detach().cpu()
    ref = orc.deform_conv_forward(x.detach().cpu(), pack.weight.detach().cpu(), pack.bias.detach().cpu(),
                                  omc[:, :18].contiguous(), torch.sigmoid(omc[:, 18:]).contiguous(), 1, 1, 1)
    assert _err(out, ref) < TOL


# ---------------------------------------------------------------------------------- group norm (+ReLU)
@pytest.mark.gpu
@pytest.mark.parametrize('C,G,relu,shapes', [
    (256, 32, True, [(2, 25, 42), (2, 13, 21), (2, 7, 11)]),      # LSHead towers: levels share one module
    (256, 32, False, [(2, 50, 84)]),                               # FPN ConvModule: no activation
    (64, 8, True, [(3, 9, 5)]),
    (1024, 32, False, [(1, 6, 7), (2, 3, 3)]),
    (128, 32, True, [(2, 130, 67)]),                               # HW not a multiple of the block size
    (32, 32, True, [(2, 25, 42), (2, 13, 21)]),                    # one channel per group (the test-size heads' towers)
    (16, 8, False, [(2, 17, 9)]),                                  # two channels per group
    (32, 16, True, [(1, 130, 67), (3, 4, 5)]),
])
def test_group_norm_matches_torch(C, G, relu, shapes):
    """fp32 reference: torch.nn.functional.group_norm (+relu) on the same device and on the CPU."""
    from lsnet_amd.ops.group_norm import GroupNorm
    torch.manual_seed(5)
    dev = torch.device('cuda:0')
    m = GroupNorm(G, C).to(dev)
    with torch.no_grad():
        m.weight.copy_(torch.randn(C) * 0.5 + 1.0)
        m.bias.copy_(torch.randn(C) * 0.3)
    # large mean / small std in some groups stresses the variance computation
    xs = [(torch.randn(b, C, h, w) * (0.2 + torch.rand(1, C, 1, 1) * 3) + torch.randn(1, C, 1, 1) * 8).to(dev)
          .contiguous(memory_format=torch.channels_last).requires_grad_() for b, h, w in shapes]
    gos = [torch.randn_like(x) for x in xs]
    ys = m.forward_multi(xs, relu=relu)
    assert all(y.is_contiguous(memory_format=torch.channels_last) for y in ys)
    torch.autograd.backward(ys, gos)
    got = [y.detach().clone() for y in ys], [x.grad.clone() for x in xs], m.weight.grad.clone(), m.bias.grad.clone()

    for where in ('cuda', 'cpu'):
        w = m.weight.detach().to(where).clone().requires_grad_()
        b = m.bias.detach().to(where).clone().requires_grad_()
        xr = [x.detach().to(where).contiguous().requires_grad_() for x in xs]
        yr = [F.group_norm(x, G, w, b, m.eps) for x in xr]
        if relu:
            yr = [F.relu(y) for y in yr]
        torch.autograd.backward(yr, [g.to(where) for g in gos])
        for y, r in zip(got[0], yr):
            assert _err(y, r.detach().to(dev)) < 1e-5
        for gx, r in zip(got[1], xr):
            assert _err(gx, r.grad.to(dev)) < 2e-5
        # (one or two channels per group: x_hat of a channel is normalised by that channel's own few-pixel statistics, and
        # two fp32 orders of the per-channel sum of dy * x_hat are ~3e-5 of the range apart -- ATen's CPU and device
        # kernels differ from each other by as much)
        ptol = 2e-5 if (C // G) % 4 == 0 else 6e-5
        assert _err(got[2], w.grad.to(dev)) < ptol
        assert _err(got[3], b.grad.to(dev)) < ptol


@pytest.mark.gpu
@pytest.mark.parametrize('relu', [False, True])
def test_group_norm_writes_the_concatenated_pixel_tensor(relu):
    """GroupNorm.forward_cat_px: the levels' outputs as ONE (B, C, N_all, 1) tensor, pixel rows of all levels back to back,
    written by the kernels themselves (lsn_gn_level.y_batch_stride), and the backward reading the levels of that tensor's
    gradient where they lie (dy_batch_stride) -- the same BITS as concatenating forward_multi's outputs and as backward over
    per-level copies: output, input gradients, gamma / beta gradients."""
    from lsnet_amd.ops.group_norm import GroupNorm
    torch.manual_seed(8)
    dev = torch.device('cuda:0')
    C, G = 256, 32
    m = GroupNorm(G, C).to(dev)
    with torch.no_grad():
        m.weight.copy_(torch.randn(C) * 0.5 + 1.0)
        m.bias.copy_(torch.randn(C) * 0.3)
    shapes = [(2, 25, 42), (2, 13, 21), (2, 7, 11), (2, 4, 6), (2, 2, 3)]
    xs = [torch.randn(b, C, h, w, device=dev).contiguous(memory_format=torch.channels_last).requires_grad_() for b, h, w in shapes]

    def cat(maps):
        B = maps[0].shape[0]
        return torch.cat([t.permute(0, 2, 3, 1).reshape(B, -1, C) for t in maps], dim=1).unsqueeze(2).permute(0, 3, 1, 2)
    y_ref = cat(m.forward_multi(xs, relu=relu))
    go = torch.randn_like(y_ref)
    ref = torch.autograd.grad(y_ref, xs + [m.weight, m.bias], go)
    y = m.forward_cat_px(xs, relu=relu)
    assert y.shape == y_ref.shape and y.stride() == y_ref.stride()
    got = torch.autograd.grad(y, xs + [m.weight, m.bias], go)
    assert torch.equal(y, y_ref)
    for a, b in zip(got, ref):
        assert torch.equal(a, b)
    # a gradient that arrives in another layout is brought into the concatenated one first
    got2 = torch.autograd.grad(m.forward_cat_px(xs, relu=relu), xs, go.contiguous())
    for a, b in zip(got2, ref):
        assert torch.equal(a, b)


@pytest.mark.gpu
def test_group_norm_statistics_buffers_alternate_between_calls():
    """The statistics sums live in two library-owned buffers per stream; each call's statistics kernel clears what the call
    before left in the other one (norm.hip gn_sums) -- no memset launch.  Calls of very different sizes back to back, then
    one with more sums than a buffer holds (the caller's workspace + memset path), then a small one again."""
    from lsnet_amd.ops.group_norm import GroupNorm
    torch.manual_seed(9)
    dev = torch.device('cuda:0')
    m = GroupNorm(32, 128).to(dev)
    with torch.no_grad():
        m.weight.copy_(torch.randn(128) * 0.5 + 1.0)
        m.bias.copy_(torch.randn(128) * 0.3)
    sizes = [[(2, 9, 7), (2, 5, 4)], [(1, 3, 3)], [(16, 6, 5), (16, 3, 3), (16, 2, 2)], [(2, 9, 7), (2, 5, 4)], [(1, 3, 3)],
             [(136, 2, 2)],          # 136 images x 32 groups x 2 sums > 8192
             [(1, 3, 3)], [(3, 11, 5)]]
    first = {}
    for rep, shapes in enumerate(sizes):
        xs = [(torch.randn(b, 128, h, w, device=dev) * 2 + 5).contiguous(memory_format=torch.channels_last) for b, h, w in shapes]
        with torch.no_grad():
            ys = m.forward_multi(xs, relu=False)
        for x, y in zip(xs, ys):
            assert _err(y, F.group_norm(x, 32, m.weight, m.bias, m.eps)) < 1e-5, (rep, tuple(x.shape))
    # the same input gives the same bits whichever buffer the call lands on
    x = torch.randn(2, 128, 9, 7, device=dev).contiguous(memory_format=torch.channels_last)
    with torch.no_grad():
        outs = [m.forward_multi([x], relu=True)[0].clone() for _ in range(3)]
    assert torch.equal(outs[0], outs[1]) and torch.equal(outs[0], outs[2])


@pytest.mark.gpu
def test_group_norm_unsupported_shape_uses_aten():
    from lsnet_amd.ops.group_norm import GroupNorm
    dev = torch.device('cuda:0')
    m = GroupNorm(3, 6).to(dev)                 # C = 6: a pixel row is not a whole number of float4
    x = torch.randn(2, 6, 5, 5, device=dev).contiguous(memory_format=torch.channels_last)
    assert torch.allclose(m(x), F.group_norm(x, 3, m.weight, m.bias, m.eps), atol=1e-6)
    y = GroupNorm(32, 256)(torch.randn(2, 256, 5, 5))      # CPU tensors take ATen's kernel as well
    assert y.shape == (2, 256, 5, 5)


# ---------------------------------------------------------------------------------- dense conv (split bf16)
@pytest.fixture(params=['bf16x6', 'bf16x3'])
def split_mode(request):
    from lsnet_amd import _lib
    old = _lib.get_math_mode()
    _lib.set_math_mode(request.param)
    yield request.param
    _lib.set_math_mode(old)


CONV_CASES = [
    # B, C, Co, k, s, p, d, H, W, bias
    (2, 256, 256, 3, 1, 1, 1, 25, 42, True),      # 64x256 tiles
    (2, 128, 128, 3, 2, 1, 1, 40, 52, False),     # 128x128 tiles, stride 2: backward-data by residue classes
    (1, 256, 64, 1, 1, 0, 1, 33, 31, True),       # 256x64 tiles, ragged pixel count
    (2, 64, 512, 1, 1, 0, 1, 20, 28, False),      # two column blocks
    (2, 768, 256, 1, 1, 0, 1, 13, 21, True),
    (2, 40, 72, 3, 1, 2, 2, 17, 19, True),        # C and Co not multiples of 32, dilation 2
    (1, 2048, 256, 3, 2, 1, 1, 25, 42, True),     # FPN P6: 2048 -> 256 stride 2
    (2, 64, 64, 1, 1, 0, 1, 50, 84, False),       # layer1-like shallow 1x1 (was MIOpen in round 1)
    (2, 256, 27, 3, 1, 1, 1, 25, 42, True),       # a DCNv2 pack's conv_offset: Co % 8 != 0
    (2, 256, 20, 1, 1, 0, 1, 13, 21, True),       # refine_out
    (2, 256, 512, 1, 2, 0, 1, 51, 85, False),     # a ResNet downsample: 1x1 stride 2, odd map (tap-less residue classes)
    (2, 64, 128, 3, 2, 1, 1, 27, 41, True),       # 3x3 stride 2, odd map
    (1, 32, 48, 3, 3, 1, 1, 20, 23, True),        # stride 3
    (1, 32, 48, 5, 2, 2, 1, 19, 22, False),       # 5x5 stride 2: 3- and 2-tap residue classes
    (1, 16, 24, 3, 2, 2, 2, 21, 26, True),        # stride 2 with dilation 2 (one residue class per axis has all taps)
    (2, 3, 64, 7, 2, 3, 1, 64, 96, False),        # the ResNet stem WITH gradients (generic path on 4 padded channels)
    (1, 256, 256, 1, 1, 0, 1, 184, 180, True),    # >= 512 tiles of 64 px x 256 channels: the 64x256 configuration
    (2, 256, 27, 3, 1, 1, 1, 100, 168, True),     # conv_offset at the P3 size: its data gradient reduces over 27 channels
                                                  # into 256 (unaligned slabs, which the 64x256 configuration lacks)
    (2, 256, 27, 3, 2, 1, 1, 26, 42, True),       # conv_offset of a STRIDED DCNv2 pack (first block of a DCN stage)
    (1, 208, 27, 3, 2, 1, 1, 6, 8, True),         # ... at the Res2Net per-scale widths
    (1, 104, 27, 3, 2, 1, 1, 12, 16, True),
    (1, 52, 27, 3, 2, 1, 1, 24, 32, True),
    # weight gradients on the deformable family's fragment-order kernel (csrc/dcn.hip conv_wgrad_dense_mm: 256 | Co, 64 | C and
    # 3x3 at >= 4096 output pixels / wide 1x1 at >= 2048 / strided 1x1 from >= 512 channels) -- the rows above stay below
    # its thresholds, and a pitch the dense caller left unset went unnoticed in round 4 until the benchmark's loss moved
    (2, 64, 256, 3, 1, 1, 1, 48, 50, True),
    (1, 1024, 512, 1, 1, 0, 1, 50, 48, False),
    (2, 512, 256, 1, 2, 0, 1, 80, 84, True),
    # 265 pixel tiles x 2 column blocks = 530 tiles on 512 workgroup slots under a deep reduction: all 530 are spread evenly by
    # chunks over 512 workgroups (round 5, csrc/conv.hip sk_plan: stream-K pieces, last arriver of a tile adds its partial
    # sums in chunk order), forward and data gradient
    (1, 192, 512, 3, 1, 1, 1, 130, 130, True),
    # the other seams of sk_plan: whole tiles in front of the stream-K part (1048 tiles: 512 plain + 536 shared), few tiles
    # under a deep reduction (66 tiles, 144 chunks: 512 pieces of ~18 chunks, up to nine partial sums per tile), a strided
    # data gradient whose residue classes take stream-K pieces with an output step, and pieces over levels of unequal size
    (1, 64, 512, 3, 1, 1, 1, 184, 182, False),
    (2, 512, 512, 3, 1, 1, 1, 25, 42, True),
    (2, 256, 256, 3, 2, 1, 1, 100, 168, False),
]


@pytest.mark.gpu
@pytest.mark.parametrize('B,C,Co,k,s,p,d,H,W,bias', CONV_CASES)
def test_conv2d_matches_torch(B, C, Co, k, s, p, d, H, W, bias, split_mode):
    """Split-bf16 implicit GEMM -- forward, data gradient (any stride), weight and bias gradient, no vendor kernel
    involved -- against an fp64 evaluation of F.conv2d: 3e-6 of the output range in the fp32-equivalent mode
    (fp32 accumulation noise), 5e-5 in the 3-product mode."""
    from lsnet_amd.ops.conv import conv2d
    torch.manual_seed(2)
    dev = _dev()
    tol = 3e-6 if split_mode == 'bf16x6' else 5e-5
    x = torch.randn(B, C, H, W, device=dev).contiguous(memory_format=torch.channels_last).requires_grad_()
    w = (torch.randn(Co, C, k, k, device=dev) / (C * k * k) ** 0.5).contiguous(memory_format=torch.channels_last) \
        .requires_grad_()
    b = torch.randn(Co, device=dev).requires_grad_() if bias else None
    y = conv2d(x, w, b, s, p, d)
    go = torch.randn_like(y)
    grads = torch.autograd.grad(y, [x, w] + ([b] if bias else []), go)
    xr, wr = x.detach().double().cpu().requires_grad_(), w.detach().double().cpu().contiguous().requires_grad_()
    br = b.detach().double().cpu().requires_grad_() if bias else None
    yr = F.conv2d(xr, wr, br, s, p, d)
    gr = torch.autograd.grad(yr, [xr, wr] + ([br] if bias else []), go.double().cpu())
    assert y.is_contiguous(memory_format=torch.channels_last)
    assert _err(y.double(), yr) < tol
    for g, r, n in zip(grads, gr, ('gx', 'gw', 'gb')):
        assert _err(g.double(), r) < tol, (n, _err(g.double(), r))


@pytest.mark.gpu
@pytest.mark.parametrize('C,Co,k,s,res,relu', [(64, 256, 1, 1, True, True), (128, 128, 3, 2, False, True),
                                                (256, 512, 1, 2, False, False), (512, 2048, 1, 1, True, True)])
def test_conv_bn_act_folded(C, Co, k, s, res, relu, split_mode):
    """relu(bn(conv(x)) + residual) with the eval-mode BatchNorm folded into the convolution's weight image (one forward
    launch, the raw convolution output never stored; ops/conv.py conv_bn_act) against an fp64 evaluation of the reference's
    three operators (resnet.py:261-301): output, and the gradients of input, residual, weight, gamma and beta -- the data
    gradient on the backward image of the SCALED weight, the three parameter gradients from ONE weight-gradient launch
    (lsn_conv2d_backward_weight_bn: conv = w . x pulled out of the pixel sum).  Channels with gamma = 0 and gamma = 1e-6
    are part of every case: nothing divides by gamma (round 3 recovered x_hat as (y - residual - beta) / gamma)."""
    from lsnet_amd.ops.conv import Conv2d, conv_bn_act
    torch.manual_seed(5)
    dev = _dev()
    tol = 5e-6 if split_mode == 'bf16x6' else 5e-4   # (3-product mode: 5e-6 relative per product, 2048-wide sums)
    B, H, W = 2, 19, 23
    conv = Conv2d(C, Co, k, stride=s, padding=k // 2, bias=False).to(dev).to(memory_format=torch.channels_last)
    bn = torch.nn.BatchNorm2d(Co).to(dev).eval()
    with torch.no_grad():
        bn.weight.copy_(torch.rand(Co) + 0.5)
        bn.weight[:4] = 0.0       # zero_init_residual (resnet.py:607-612)
        bn.weight[4:8] = 1e-6
        bn.weight[8:12] *= -1.0
        bn.bias.copy_(torch.randn(Co) * 0.3)
        bn.running_mean.copy_(torch.randn(Co) * 0.2)
        bn.running_var.copy_(torch.rand(Co) + 0.5)
    x = torch.randn(B, C, H, W, device=dev).contiguous(memory_format=torch.channels_last).requires_grad_()
    Ho, Wo = (H + 2 * (k // 2) - k) // s + 1, (W + 2 * (k // 2) - k) // s + 1
    r = torch.randn(B, Co, Ho, Wo, device=dev).contiguous(memory_format=torch.channels_last).requires_grad_() if res else None
    y = conv_bn_act(conv, bn, x, relu=relu, residual=r)
    assert y is not None and y.is_contiguous(memory_format=torch.channels_last)
    go = torch.randn_like(y)
    wrt = [x, conv.weight, bn.weight, bn.bias] + ([r] if res else [])
    grads = torch.autograd.grad(y, wrt, go)
    d = lambda t: t.detach().double().cpu()
    xr, wr, gr, br = d(x).requires_grad_(), d(conv.weight).contiguous().requires_grad_(), d(bn.weight).requires_grad_(), \
        d(bn.bias).requires_grad_()
    rr = d(r).requires_grad_() if res else None
    z = F.batch_norm(F.conv2d(xr, wr, None, s, k // 2), d(bn.running_mean), d(bn.running_var), gr, br, False, 0.0, bn.eps)
    if res:
        z = z + rr
    yr = F.relu(z) if relu else z
    gref = torch.autograd.grad(yr, [xr, wr, gr, br] + ([rr] if res else []), go.double().cpu())
    assert _err(y.double(), yr) < tol
    for g, ref, n in zip(grads, gref, ('gx', 'gw', 'ggamma', 'gbeta', 'gres')):
        e = _err(g.double(), ref)
        if split_mode == 'bf16x3' and relu and e >= tol:
            # an output within the 3-product mode's 5e-5 of zero has its ReLU gate on the other side than in fp64: that
            # one element of the pre-activation gradient is then all or nothing (measured: 1.4e-2 of the range of gx
            # under a 2048-wide sum, 0.26 of the range of the residual gradient, which IS that element).  Legitimate kink
            # behaviour -- held to a share of elements, not by the maximum.
            rel = (g.double().cpu() - ref).abs() / ref.abs().max()
            assert float((rel > tol).double().mean()) < 0.01, (n, e, float((rel > tol).double().mean()))
            continue
        assert e < tol, (n, e)


GROUP_CONV_CASES = [
    # B, C (= Co), groups, k, stride, pad, dil, H, W, bias
    (2, 256, 64, 3, 1, 1, 1, 50, 84, False),     # ResNeXt-101 64x4d layer1 conv2 (4 channels per group)
    (2, 512, 64, 3, 2, 1, 1, 27, 41, False),     # layer2.0 conv2: stride 2, odd map (8 per group)
    (1, 1024, 64, 3, 1, 1, 1, 13, 21, False),    # layer3 width (16 per group)
    (1, 2048, 64, 3, 2, 1, 1, 14, 11, True),     # layer4.0 width, stride 2 (32 per group)
    (2, 128, 32, 3, 1, 2, 2, 17, 19, True),      # dilation 2
    (1, 96, 24, 1, 1, 0, 1, 9, 70, True),        # 1x1 grouped, 24 groups: a partial 32-channel slab at the end
    (1, 40, 10, 3, 3, 1, 1, 20, 23, False),      # stride 3, 10 groups of 4
]


@pytest.mark.gpu
@pytest.mark.parametrize('B,C,G,k,s,p,d,H,W,bias', GROUP_CONV_CASES)
def test_grouped_conv2d_matches_torch(B, C, G, k, s, p, d, H, W, bias):
    """Grouped convolution (ResNeXt bottleneck conv2; reference: nn.Conv2d(groups=64) of resnext.py:11-83) through the
    exact-fp32 kernels of csrc/gconv.hip -- forward, data gradient, weight and bias gradient, in EVERY math mode --
    against an fp64 evaluation of F.conv2d: fp32 rounding only (1e-6 of the range)."""
    from lsnet_amd.ops.conv import Conv2d, hip_group_conv_ok
    torch.manual_seed(8)
    dev = _dev()
    m = Conv2d(C, C, k, stride=s, padding=p, dilation=d, groups=G, bias=bias).to(dev).to(memory_format=torch.channels_last)
    x = torch.randn(B, C, H, W, device=dev).contiguous(memory_format=torch.channels_last).requires_grad_()
    assert hip_group_conv_ok(x, m.weight, m.stride, m.padding, m.dilation, m.groups)
    from lsnet_amd import _lib
    before = _lib.get_math_mode()
    try:
        for mode in ('bf16x6', 'fp32'):
            _lib.set_math_mode(mode)
            y = m(x)
            go = torch.randn_like(y)
            params = list(m.parameters())
            grads = torch.autograd.grad(y, [x] + params, go)
            xr = x.detach().double().cpu().requires_grad_()
            pr = [q.detach().double().cpu().contiguous().requires_grad_() for q in params]
            yr = F.conv2d(xr, pr[0], pr[1] if bias else None, s, p, d, G)
            gr = torch.autograd.grad(yr, [xr] + pr, go.double().cpu())
            assert y.is_contiguous(memory_format=torch.channels_last) and y.shape == yr.shape
            assert _err(y.double(), yr) < 1e-6
            for g, r, n in zip(grads, gr, ('gx', 'gw', 'gb')):
                assert g.shape == r.shape
                assert _err(g.double(), r) < 2e-6, (mode, n, _err(g.double(), r))
    finally:
        _lib.set_math_mode(before)


@pytest.mark.gpu
def test_grouped_conv2d_unsupported_shape_uses_aten():
    """5 channels per group is outside the kernels' list: the module keeps ATen's operator, the C entry says so."""
    import ctypes
    from lsnet_amd import _lib
    from lsnet_amd.ops.conv import Conv2d, hip_group_conv_ok
    dev = _dev()
    m = Conv2d(20, 20, 3, padding=1, groups=4).to(dev).to(memory_format=torch.channels_last)
    x = torch.randn(1, 20, 9, 11, device=dev).contiguous(memory_format=torch.channels_last)
    assert not hip_group_conv_ok(x, m.weight, m.stride, m.padding, m.dilation, m.groups)
    assert torch.allclose(m(x), F.conv2d(x, m.weight, m.bias, 1, 1, 1, 4), atol=1e-5)
    lib = _lib.load()
    out = torch.empty_like(x)
    p = lambda t: ctypes.c_void_p(t.data_ptr())
    rc = lib.lsn_grouped_conv2d_forward(p(x), p(m.weight), None, p(out), 1, 9, 11, 20, 20, 3, 3, 1, 1, 1, 4, 0, None)
    assert rc == -2 and b'channels per group' in lib.lsn_last_error()


@pytest.mark.gpu
@pytest.mark.parametrize('C,Co,k,bias,relu', [(256, 27, 3, True, False), (256, 256, 3, True, False), (768, 256, 1, True, True),
                                              (64, 20, 1, False, False)])
def test_conv2d_multi_level_equals_single(C, Co, k, bias, relu, split_mode):
    """The FPN levels of a shared convolution in ONE launch each way == level-by-level calls (forward, input
    gradients, summed weight / bias gradient), and == an fp64 evaluation."""
    from lsnet_amd.ops.conv import Conv2d
    torch.manual_seed(6)
    dev = _dev()
    tol = 3e-6 if split_mode == 'bf16x6' else 5e-5
    m = Conv2d(C, Co, k, padding=k // 2, bias=bias).to(dev).to(memory_format=torch.channels_last)
    xs = [torch.randn(2, C, h, w, device=dev).contiguous(memory_format=torch.channels_last).requires_grad_()
          for h, w in [(50, 84), (25, 42), (13, 21), (7, 11), (4, 6)]]
    outs = m.forward_multi(xs, relu=relu)
    gos = [torch.randn_like(o) for o in outs]
    params = list(m.parameters())
    g_multi = torch.autograd.grad(outs, xs + params, gos)
    singles = [F.relu(m(x)) if relu else m(x) for x in xs]
    g_single = torch.autograd.grad(singles, xs + params, gos)
    for a_, b_ in zip(outs, singles):   # (a small level on its own may take a split reduction: another fp32 summation order)
        assert _err(a_, b_.detach().cpu()) < tol
    for a_, b_ in zip(g_multi, g_single):
        assert _err(a_, b_.detach().cpu()) < tol
    xr = [x.detach().double().cpu().requires_grad_() for x in xs]
    pr = [p_.detach().double().cpu().requires_grad_() for p_ in params]
    yr = [F.conv2d(x, pr[0], pr[1] if bias else None, 1, k // 2) for x in xr]
    yr = [F.relu(y) for y in yr] if relu else yr
    gr = torch.autograd.grad(yr, xr + pr, [g.double().cpu() for g in gos])
    for a_, b_ in zip(list(outs) + list(g_multi), yr + list(gr)):
        assert _err(a_.double(), b_) < tol


@pytest.mark.gpu
@pytest.mark.parametrize('C,Co,k,s,H,W', [(512, 512, 3, 1, 25, 42), (256, 256, 3, 1, 50, 84), (192, 512, 3, 1, 130, 130),
                                          (256, 256, 3, 2, 100, 168)])
def test_conv2d_stream_k_is_deterministic(C, Co, k, s, H, W):
    """Stream-K launches (csrc/conv.hip sk_plan) add a tile's partial sums in chunk order whichever workgroup arrives last:
    the same bits on every run, forward and data gradient, and the arrival counters are left re-armed (the second and third
    calls would hang short of a tile or double-count otherwise)."""
    from lsnet_amd.ops.conv import conv2d
    torch.manual_seed(11)
    dev = _dev()
    x = torch.randn(2, C, H, W, device=dev).contiguous(memory_format=torch.channels_last).requires_grad_()
    w = (torch.randn(Co, C, k, k, device=dev) / (C * k * k) ** 0.5).contiguous(memory_format=torch.channels_last)
    b = torch.randn(Co, device=dev)
    ys, gs = [], []
    for _ in range(3):
        y = conv2d(x, w, b, s, k // 2, 1)
        go = torch.ones_like(y) * 0.5 + y.detach() * 0.25
        ys.append(y.detach().clone())
        gs.append(torch.autograd.grad(y, x, go)[0].clone())
    assert torch.equal(ys[0], ys[1]) and torch.equal(ys[0], ys[2])
    assert torch.equal(gs[0], gs[1]) and torch.equal(gs[0], gs[2])
    ref = F.conv2d(x.detach().double().cpu(), w.double().cpu(), b.double().cpu(), s, k // 2)
    assert _err(ys[0].double(), ref) < 3e-6


@pytest.mark.gpu
@pytest.mark.parametrize('relu', [False, True])
def test_conv2d_multi_level_residual_in_the_epilogue(relu):
    """Conv2d.forward_multi(xs, residuals=rs): the per-level sum conv(x) + r rides in the launch's epilogue (in front of the
    ReLU).  Without ReLU it is LSHead's `3x3(tower) + relu(1x1(gathered))`: the same additions and roundings as the separate
    element-wise launch it replaces -- the same BITS, outputs and all gradients (the residual's gradient is the output
    gradient itself)."""
    from lsnet_amd.ops.conv import Conv2d
    torch.manual_seed(9)
    dev = _dev()
    m = Conv2d(256, 256, 3, padding=1).to(dev).to(memory_format=torch.channels_last)
    sizes = [(50, 84), (25, 42), (13, 21), (7, 11), (4, 6)]
    xs = [torch.randn(2, 256, h, w, device=dev).contiguous(memory_format=torch.channels_last).requires_grad_() for h, w in sizes]
    rs = [torch.randn(2, 256, h, w, device=dev).contiguous(memory_format=torch.channels_last).requires_grad_() for h, w in sizes]
    outs = m.forward_multi(xs, relu=relu, residuals=rs)
    gos = [torch.randn_like(o) for o in outs]
    got = torch.autograd.grad(outs, xs + rs + list(m.parameters()), gos)
    plain = [a + b for a, b in zip(m.forward_multi(xs), rs)]
    plain = [F.relu(o) for o in plain] if relu else plain
    ref = torch.autograd.grad(plain, xs + rs + list(m.parameters()), gos)
    for a, b in zip(outs, plain):
        assert torch.equal(a, b)
    for a, b in zip(got, ref):
        assert torch.equal(a, b)


@pytest.mark.gpu
def test_stem_row_merged_forward(split_mode):
    """The frozen 7x7 stride-2 stem on the 3-channel image: row-merged form (lsn_conv2d_forward_pitched)."""
    from lsnet_amd.ops.conv import Conv2d
    torch.manual_seed(3)
    dev = _dev()
    m = Conv2d(3, 64, 7, stride=2, padding=3, bias=False).to(dev).to(memory_format=torch.channels_last)
    for p_ in m.parameters():
        p_.requires_grad_(False)
    for hw in ((64, 96), (75, 101)):
        x = torch.randn(2, 3, *hw, device=dev).contiguous(memory_format=torch.channels_last)
        ref = F.conv2d(x.double().cpu(), m.weight.double().cpu(), None, 2, 3)
        assert _err(m(x).double(), ref) < (3e-6 if split_mode == 'bf16x6' else 5e-5)


@pytest.mark.gpu
def test_conv2d_module_dispatch():
    """Which kernels a Conv2d module reaches: the own ones for every CUDA fp32 tensor, in every math mode and layout."""
    from lsnet_amd import _lib
    from lsnet_amd.ops.conv import Conv2d
    dev = _dev()
    old = _lib.get_math_mode()
    try:
        _lib.set_math_mode('bf16x6')
        m = Conv2d(256, 256, 3, padding=1).to(dev).to(memory_format=torch.channels_last)
        x = torch.randn(2, 256, 100, 168, device=dev).contiguous(memory_format=torch.channels_last)
        ref = F.conv2d(x, m.weight, m.bias, 1, 1)
        assert _err(m(x), ref) < 5e-6                       # own kernel (different summation order than MIOpen)
        assert not torch.equal(m(x), ref)
        small = Conv2d(64, 64, 1).to(dev).to(memory_format=torch.channels_last)
        xs = torch.randn(2, 64, 20, 20, device=dev).contiguous(memory_format=torch.channels_last)
        assert _err(small(xs), F.conv2d(xs, small.weight, small.bias)) < 5e-6   # small layers too: no vendor kernels
        _lib.set_math_mode('fp32')
        y32 = m(x)                                          # exact mode: still the own (fp32-equivalent) kernels,
        assert _err(y32, ref) < 5e-6 and not torch.equal(y32, ref)   # no vendor convolution behind a CUDA fp32 tensor
        # a contiguous (NCHW) input is re-laid channels-last and takes the same kernels; the result is channels-last
        xn = x.contiguous()
        mn = Conv2d(256, 256, 3, padding=1).to(dev)
        mn.load_state_dict(m.state_dict())
        yn = mn(xn)
        assert yn.is_contiguous(memory_format=torch.channels_last) and _err(yn, ref) < 5e-6 and not torch.equal(yn, ref)
    finally:
        _lib.set_math_mode(old)


# ---------------------------------------------------------------------------------- frozen BN + add + ReLU
@pytest.mark.gpu
@pytest.mark.parametrize('C,shape,relu,res', [(64, (2, 40, 52), True, False), (256, (2, 25, 42), True, True),
                                               (1024, (1, 13, 21), False, False), (512, (2, 9, 7), True, True),
                                               (2048, (2, 7, 11), True, True), (2048, (1, 5, 9), False, False)])
def test_bn_eval_act_matches_torch(C, shape, relu, res):
    from lsnet_amd.ops.batch_norm import bn_act
    torch.manual_seed(4)
    dev = _dev()
    bn = torch.nn.BatchNorm2d(C).to(dev)
    with torch.no_grad():
        bn.weight.copy_(torch.randn(C) * 0.5 + 1)
        bn.bias.copy_(torch.randn(C) * 0.2)
        bn.running_mean.copy_(torch.randn(C))
        bn.running_var.copy_(torch.rand(C) + 0.3)
    bn.eval()
    b, h, w = shape
    x = torch.randn(b, C, h, w, device=dev).contiguous(memory_format=torch.channels_last).requires_grad_()
    r = torch.randn(b, C, h, w, device=dev).contiguous(memory_format=torch.channels_last).requires_grad_() if res else None
    y = bn_act(bn, x, relu=relu, residual=r)
    go = torch.randn_like(y)
    ins = [x] + ([r] if res else []) + [bn.weight, bn.bias]
    got = torch.autograd.grad(y, ins, go)
    x2 = x.detach().clone().requires_grad_()
    r2 = r.detach().clone().requires_grad_() if res else None
    ref = F.batch_norm(x2, bn.running_mean, bn.running_var, bn.weight, bn.bias, False, 0.0, bn.eps)
    if res:
        ref = ref + r2
    if relu:
        ref = F.relu(ref)
    gref = torch.autograd.grad(ref, [x2] + ([r2] if res else []) + [bn.weight, bn.bias], go)
    assert _err(y, ref) < 1e-6
    for g, gr in zip(got, gref):
        assert _err(g, gr) < 2e-5
    if C == 64:                     # batch statistics: not this kernel's business (one case is enough: MIOpen's
        bn.train()                  # training-mode BN segfaults on the 1x1024x13x21 channels-last case here)
        assert torch.allclose(bn_act(bn, x.detach(), relu=True), F.relu(bn(x.detach())), atol=1e-5)


# ---------------------------------------------------------------------------------- fp32 equivalence of 'bf16x6'
def _dcn_all(ops, x, w, b, off, mask, go, cfg, dev, cl=True):
    xd, wd, od = _to(x, dev, cl).requires_grad_(), _to(w, dev, cl).requires_grad_(), _to(off, dev, cl).requires_grad_()
    md = _to(mask, dev, cl)
    bd = None if b is None else b.to(dev).requires_grad_()
    if md is not None:
        md.requires_grad_()
    out = ops.dcn_multi([xd], [od], [md], wd, bd, cfg['stride'], cfg['pad'], cfg['dil'], cfg['groups'], cfg['dg'],
                        scales=[(cfg['sh'], cfg['sw'])], pyramid=cfg['pyramid'])[0]
    wrt = [t for t in (xd, od, md, wd, bd) if t is not None]
    grads = torch.autograd.grad(out, wrt, _to(go, dev, cl))
    names = ['gx', 'goff'] + (['gmask'] if md is not None else []) + ['gw'] + (['gb'] if bd is not None else [])
    res = {'out': out.detach()}
    res.update({n: g.detach() for n, g in zip(names, grads)})
    return res


def _dcn_fp64(x, w, b, off, mask, go, cfg):
    """fp64 evaluation (tests/torch_dcn_ref.py, autograd) at the kernels' own fp32 sampling positions."""
    from tests.torch_dcn_ref import torch_dcn
    xd, wd = x.double().requires_grad_(), w.double().requires_grad_()
    od = off.float().requires_grad_()        # positions (and therefore the bilinear fractions) stay fp32
    md = None if mask is None else mask.double().requires_grad_()
    bd = None if b is None else b.double().requires_grad_()
    out = torch_dcn(xd, od, md, wd, bd, cfg['stride'], cfg['pad'], cfg['dil'], cfg['groups'], cfg['dg'], cfg['sh'], cfg['sw'])
    wrt = [t for t in (xd, od, md, wd, bd) if t is not None]
    grads = torch.autograd.grad(out, wrt, go.double())
    names = ['gx', 'goff'] + (['gmask'] if md is not None else []) + ['gw'] + (['gb'] if bd is not None else [])
    res = {'out': out.detach()}
    res.update({n: g.detach().double() for n, g in zip(names, grads)})
    return res


@pytest.mark.parametrize('case', [c for c in DCN_CASES if c['name'] in ('v2_head_p6', 'pyr_head', 'r101_l3', 'v2_c40_co72')],
                         ids=lambda c: c['name'])
def test_split6_matches_exact_fp32(case):
    """'bf16x6' is the library's fp32-equivalent arithmetic.  Measured against an fp64 evaluation, every output of the
    deformable family is at least as accurate as the exact fp32 MFMA kernels' (whose error is the rounding of an fp32
    fmaf chain over the 2304-term reduction), and the two agree to a few 1e-6 of the output range -- the sum of their
    own rounding errors; the 3-product mode is an order of magnitude off."""
    from lsnet_amd import _lib, ops
    dev = _dev()
    x, w, b, off, mask, go, cfg = _make(case, dev, seed=11)
    truth = _dcn_fp64(x, w, b, off, mask, go, cfg)
    old = _lib.get_math_mode()
    got = {}
    try:
        for mode in ('fp32', 'bf16x6', 'bf16x3'):
            _lib.set_math_mode(mode)
            got[mode] = _dcn_all(ops, x, w, b, off, mask, go, cfg, dev)
    finally:
        _lib.set_math_mode(old)
    err = {m: {k: _err(v[k].double(), truth[k]) for k in truth} for m, v in got.items()}
    for m in err:
        print(case['name'], m, 'vs fp64', {k: f'{v:.1e}' for k, v in err[m].items()})
    cross = {k: _err(got['bf16x6'][k], got['fp32'][k].cpu()) for k in truth}
    print(case['name'], 'x6 vs fp32 kernels', {k: f'{v:.1e}' for k, v in cross.items()})
    for k in truth:
        assert err['bf16x6'][k] <= 1.25 * err['fp32'][k] + 2e-7, (k, err['bf16x6'][k], err['fp32'][k])
        assert cross[k] <= 5e-6, (k, cross[k])


@pytest.mark.parametrize('B,C,Co,k,s,p,d,H,W', [(2, 256, 256, 3, 1, 1, 1, 50, 84), (2, 1024, 512, 1, 1, 0, 1, 25, 42),
                                                (1, 2048, 256, 3, 2, 1, 1, 25, 42), (2, 128, 128, 3, 2, 1, 1, 40, 52)])
def test_conv_split6_matches_fp64(B, C, Co, k, s, p, d, H, W):
    """Dense convolution in 'bf16x6' against an fp64 evaluation: the error must be fp32 rounding (as MIOpen's exact
    fp32 kernels'), two orders below the 3-product mode."""
    from lsnet_amd import _lib
    from lsnet_amd.ops.conv import conv2d
    torch.manual_seed(7)
    dev = _dev()
    x = torch.randn(B, C, H, W, device=dev).contiguous(memory_format=torch.channels_last).requires_grad_()
    w = (torch.randn(Co, C, k, k, device=dev) / (C * k * k) ** 0.5).contiguous(memory_format=torch.channels_last) \
        .requires_grad_()
    old = _lib.get_math_mode()
    res = {}
    try:
        for mode in ('bf16x6', 'bf16x3'):
            _lib.set_math_mode(mode)
            y = conv2d(x, w, None, s, p, d)
            go = torch.randn(y.shape, device=dev, generator=torch.Generator(device=dev).manual_seed(3)) \
                .contiguous(memory_format=torch.channels_last)
            res[mode] = [y.detach()] + [g.detach() for g in torch.autograd.grad(y, [x, w], go)]
    finally:
        _lib.set_math_mode(old)
    ym = F.conv2d(x, w, None, s, p, d)
    res['miopen'] = [ym.detach()] + [g.detach() for g in torch.autograd.grad(ym, [x, w], go)]
    xr, wr = x.detach().double().cpu().requires_grad_(), w.detach().double().cpu().contiguous().requires_grad_()
    yr = F.conv2d(xr, wr, None, s, p, d)
    ref = [yr.detach()] + list(torch.autograd.grad(yr, [xr, wr], go.double().cpu()))
    errs = {m: [(_err(a.double(), r)) for a, r in zip(v, ref)] for m, v in res.items()}
    print((B, C, Co, k, s), {m: [f'{e:.1e}' for e in v] for m, v in errs.items()})
    # fp32 accumulation noise grows with the reduction length (MIOpen blocks its sums; a plain fmaf chain does not)
    floor = max(1e-6, 2e-8 * (C * k * k) ** 0.5)
    for e6, em in zip(errs['bf16x6'], errs['miopen']):
        assert e6 <= max(floor, 2 * em), errs


# ---------------------------------------------------------------------------------- bench-shaped launches
FPN_SIZES = [(100, 168), (50, 84), (25, 42), (13, 21), (7, 11)]   # 800 x 1344 input, strides 8 .. 128


@pytest.mark.parametrize('choice', list(KERNEL_CHOICES))
def test_tower_launch_at_bench_shape(choice):
    """The LSHead tower call of BASELINE config 2: DCNv2 256 -> 256 over all five FPN levels (B = 2) in ONE launch,
    forward and every gradient against the oracle.  These sizes take the multi-wave tile counts, the XCD remap and the
    per-level tile table that the small cases do not."""
    from lsnet_amd import _lib, ops
    dev = _dev()
    mode, flag = KERNEL_CHOICES[choice]
    g = torch.Generator().manual_seed(21)
    C = Co = 256
    w = torch.randn(Co, C, 3, 3, generator=g) / (3 * C ** 0.5)
    b = torch.randn(Co, generator=g)
    xs = [torch.randn(2, C, h, ww, generator=g) for h, ww in FPN_SIZES]
    offs = [torch.randn(2, 18, h, ww, generator=g) * 1.5 for h, ww in FPN_SIZES]
    msks = [torch.rand(2, 9, h, ww, generator=g) for h, ww in FPN_SIZES]
    gos = [torch.randn(2, Co, h, ww, generator=g) for h, ww in FPN_SIZES]
    old = _lib.get_math_mode()
    _lib.load().lsn_debug_phase_clocks(None, flag)
    _lib.set_math_mode(mode)
    try:
        wd, bd = _to(w, dev, True).requires_grad_(), b.to(dev).requires_grad_()
        xd = [_to(t, dev, True).requires_grad_() for t in xs]
        od = [_to(t, dev, True).requires_grad_() for t in offs]
        md = [_to(t, dev, True).requires_grad_() for t in msks]
        outs = ops.dcn_multi(xd, od, md, wd, bd, 1, 1, 1)
        god = [_to(t, dev, True) for t in gos]
        grads = torch.autograd.grad(outs, [wd, bd] + xd + od + md, god, retain_graph=True)
        torch.cuda.synchronize()
        if choice == 'default':
            # the atomic-free backward at the benchmark shape (two workgroups per CU, LDS-DMA slabs): bit-identical
            # grad_input / grad_offset / grad_mask on a second run (round-2 advisor: timing-dependent errors hide here)
            again = torch.autograd.grad(outs, xd + od + md, god)
            for k, (g1, g2) in enumerate(zip(grads[2:], again)):
                assert torch.equal(g1, g2), f'tensor {k} of the data-side gradients differs between two runs'
    finally:
        _lib.set_math_mode(old)
        _lib.load().lsn_debug_phase_clocks(None, 0)
    gw_ref, gb_ref = torch.zeros_like(w), torch.zeros_like(b)
    errs = {}
    for i in range(5):
        ref = orc.deform_conv_forward(xs[i], w, b, offs[i], msks[i], 1, 1, 1)
        gr = orc.deform_conv_backward(xs[i], w, offs[i], msks[i], gos[i], 1, 1, 1)
        gw_ref += gr['gw']
        gb_ref += gr['gb']
        errs[f'out{i}'] = _report(f'tower/out{i}', outs[i], ref)
        errs[f'gx{i}'] = _report(f'tower/gx{i}', grads[2 + i], gr['gx'])
        errs[f'goff{i}'] = _report(f'tower/goff{i}', grads[7 + i], gr['goff'])
        errs[f'gmask{i}'] = _report(f'tower/gmask{i}', grads[12 + i], gr['gmask'])
    errs['gw'] = _report('tower/gw', grads[0], gw_ref)
    errs['gb'] = _report('tower/gb', grads[1], gb_ref)
    print(choice, {k: f'{v:.1e}' for k, v in errs.items()})
    assert all(e < TOL for e in errs.values()), errs


@pytest.mark.parametrize('sizes', [[(25, 42), (13, 21), (7, 11)], [(50, 84), (25, 42), (13, 21), (7, 11)], [(9, 13)]],
                         ids=['3lv', '4lv_one_round_and_more', 'one_tile_row'])
def test_dcn_forward_stream_k_pieces(sizes):
    """Round 6: the deformable forward's stream-K pieces (dcn_mm_kernels.h DcnSk; production: launches of two rounds of workgroups
    and more, i.e. the 15-pair pyramid launch -- test_pyramid_launch_at_bench_shape runs that one against the oracle).  Debug bit
    18 asks for pieces at any size: the result must equal the oracle's, agree with the whole-tile launch up to the summation order of
    the pieces, and come out bit for bit the same on every run (the tile's last arriver adds the slots in chunk order)."""
    from lsnet_amd import _lib, ops
    dev = _dev()
    g = torch.Generator().manual_seed(41)
    C = Co = 256
    w = torch.randn(Co, C, 3, 3, generator=g) / (3 * C ** 0.5)
    b = torch.randn(Co, generator=g)
    xs = [torch.randn(2, C, h, ww, generator=g) for h, ww in sizes]
    offs = [torch.randn(2, 18, h, ww, generator=g) * 1.5 for h, ww in sizes]
    msks = [torch.rand(2, 9, h, ww, generator=g) for h, ww in sizes]
    wd, bd = _to(w, dev, True), b.to(dev)
    xd, od, md = [_to(t, dev, True) for t in xs], [_to(t, dev, True) for t in offs], [_to(t, dev, True) for t in msks]

    def run(flag):
        _lib.load().lsn_debug_phase_clocks(None, flag)
        try:
            with torch.no_grad():
                return [o.clone() for o in ops.dcn_multi(xd, od, md, wd, bd, 1, 1, 1)]
        finally:
            _lib.load().lsn_debug_phase_clocks(None, 0)

    whole = run(1 << 19)
    pieces = run(1 << 18)
    again = run(1 << 18)
    for i in range(len(sizes)):
        ref = orc.deform_conv_forward(xs[i], w, b, offs[i], msks[i], 1, 1, 1)
        assert _report(f'sk/out{i}', pieces[i], ref) < TOL
        assert torch.equal(pieces[i], again[i]), f'level {i}: two launches with pieces differ'
        scale = float(whole[i].abs().max())
        assert float((pieces[i] - whole[i]).abs().max()) <= 2e-6 * scale


def test_dcn_weight_images_cached_per_optimizer_step():
    """Round 6: a deformable layer whose weight is a channels-last Parameter reads its fragment images from
    the cache of the dense convolutions (lsn_dcn_shape.weights_prepared: forward = kind 0, backward GEMM = kind 2) instead of
    rebuilding them in every call.  Same bits as the in-call build, forward and every gradient, also after the weight moved."""
    from lsnet_amd import ops
    from lsnet_amd.ops import hip_backend
    dev = _dev()
    g = torch.Generator().manual_seed(51)
    C = Co = 256
    sizes = [(25, 42), (13, 21)]
    w = torch.nn.Parameter(_to(torch.randn(Co, C, 3, 3, generator=g) / (3 * C ** 0.5), dev, True))
    b = torch.nn.Parameter(torch.randn(Co, generator=g).to(dev))
    xs = [_to(torch.randn(2, C, h, ww, generator=g), dev, True).requires_grad_() for h, ww in sizes]
    offs = [_to(torch.randn(2, 18, h, ww, generator=g) * 1.5, dev, True).requires_grad_() for h, ww in sizes]
    msks = [_to(torch.rand(2, 9, h, ww, generator=g), dev, True).requires_grad_() for h, ww in sizes]
    gos = [_to(torch.randn(2, Co, h, ww, generator=g), dev, True) for h, ww in sizes]

    def run(cached):
        old = hip_backend.CACHE_DCN_IMAGES
        hip_backend.CACHE_DCN_IMAGES = cached
        try:
            outs = ops.dcn_multi(xs, offs, msks, w, b, 1, 1, 1)
            grads = torch.autograd.grad(outs, [w, b] + xs + offs + msks, gos)
            return [o.detach().clone() for o in outs] + [t.clone() for t in grads]
        finally:
            hip_backend.CACHE_DCN_IMAGES = old

    for step in range(2):
        ref = run(False)
        got = run(True)
        again = run(True)        # (second call: the images come from the cache untouched)
        for k, (a, c, d) in enumerate(zip(ref, got, again)):
            assert torch.equal(a, c) and torch.equal(a, d), (step, k)
        with torch.no_grad():
            w.add_(0.01)          # the version counter moves: the cached images are stale and rebuilt on the next call


@pytest.mark.parametrize('mode', ['bf16x6', 'fp32'])
def test_pyramid_outputs_side_by_side(mode):
    """`concat=3` of the pyramid op (lsn_dcn_shape.out_pitch: the three maps of a destination level written into one
    768-channel tensor, their gradient read from where the next operator leaves it) against torch.cat of the separate
    outputs: the same bits forward and backward in the matrix-pipe mode; the exact-fp32 mode has no pitched kernels and
    takes the fallback (separate outputs + ATen's concatenation), same result by construction."""
    from lsnet_amd import _lib, ops
    dev = _dev()
    g = torch.Generator().manual_seed(31)
    C = Co = 256
    sizes = [(25, 42), (13, 21), (7, 11)]
    level_lists = [[0, 1, 2], [1, 0, 2], [2, 1, 0]]
    w = torch.randn(Co, C, 3, 3, generator=g) / (3 * C ** 0.5)
    feats = [torch.randn(2, C, h, ww, generator=g) for h, ww in sizes]
    pairs = [(l, s) for l, lst in enumerate(level_lists) for s in lst]
    offs, scales = [], []
    for l, s in pairs:
        h, ww = sizes[l]
        scales.append((sizes[s][0] / h, sizes[s][1] / ww))
        offs.append(torch.randn(2, 18, h, ww, generator=g) * 2.0)
    gos = [torch.randn(2, 3 * Co, h, ww, generator=g) for h, ww in sizes]
    old = _lib.get_math_mode()
    _lib.set_math_mode(mode)
    try:
        def run(concat):
            wd = _to(w, dev, True).requires_grad_()
            fd = [_to(t, dev, True).requires_grad_() for t in feats]
            od = [_to(t, dev, True).requires_grad_() for t in offs]
            outs = ops.dcn_multi([fd[s] for _, s in pairs], od, None, wd, None, 1, 1, 1, scales=scales, pyramid=True,
                                 concat=3 if concat else 0)
            if not concat:
                outs = [torch.cat(outs[3 * l:3 * l + 3], dim=1) for l in range(3)]
            grads = torch.autograd.grad(outs, [wd] + fd + od, [_to(t, dev, True) for t in gos])
            return [o.detach() for o in outs], grads
        o1, g1 = run(True)
        o0, g0 = run(False)
    finally:
        _lib.set_math_mode(old)
    assert all(o.shape[1] == 3 * Co and o.is_contiguous(memory_format=torch.channels_last) for o in o1)
    assert all(torch.equal(a, b) for a, b in zip(o1, o0))
    if mode == 'bf16x6':
        assert all(torch.equal(a, b) for a, b in zip(g1, g0))
    else:   # (the exact mode scatters its data gradients with fp32 atomics: equal up to their order)
        assert all(float((a - b).abs().max()) <= 2e-5 * float(b.abs().max()) for a, b in zip(g1, g0))


@pytest.mark.parametrize('choice', ['default', 'x3_gather', 'x3_atomic_fallbacks', 'math_fp32'])
def test_pyramid_launch_at_bench_shape(choice):
    """One PyramidDeformConv of LSHead.forward_single2 (lsnet_head.py:600-755) at BASELINE config 2: the 15 (level,
    source) pairs in ONE launch; sources are shared by several pairs, so their gradients accumulate in one buffer."""
    from lsnet_amd import _lib, ops
    dev = _dev()
    mode, flag = KERNEL_CHOICES[choice]
    g = torch.Generator().manual_seed(22)
    C = Co = 256
    level_lists = [[0, 1, 2], [1, 0, 2], [2, 1, 3], [3, 2, 4], [4, 3, 2]]
    w = torch.randn(Co, C, 3, 3, generator=g) / (3 * C ** 0.5)
    feats = [torch.randn(2, C, h, ww, generator=g) for h, ww in FPN_SIZES]
    pairs = [(l, s) for l, lst in enumerate(level_lists) for s in lst]
    offs, gos, scales = [], [], []
    for l, s in pairs:
        h, ww = FPN_SIZES[l]
        hs, ws = FPN_SIZES[s]
        sc = (hs / h, ws / ww)
        # landmark-like offsets: most samples of an object point at the same few places of the source map
        offs.append(torch.randn(2, 18, h, ww, generator=g) * 2.0 * max(sc[0], 1.0))
        gos.append(torch.randn(2, Co, h, ww, generator=g))
        scales.append(sc)
    old = _lib.get_math_mode()
    _lib.load().lsn_debug_phase_clocks(None, flag)
    _lib.set_math_mode(mode)
    try:
        wd = _to(w, dev, True).requires_grad_()
        fd = [_to(t, dev, True).requires_grad_() for t in feats]
        od = [_to(t, dev, True).requires_grad_() for t in offs]
        outs = ops.dcn_multi([fd[s] for _, s in pairs], od, None, wd, None, 1, 1, 1, scales=scales, pyramid=True)
        grads = torch.autograd.grad(outs, [wd] + fd + od, [_to(t, dev, True) for t in gos])
        torch.cuda.synchronize()
    finally:
        _lib.set_math_mode(old)
        _lib.load().lsn_debug_phase_clocks(None, 0)
    gw_ref = torch.zeros_like(w)
    gx_ref = [torch.zeros_like(f) for f in feats]
    errs = {}
    for i, (l, s) in enumerate(pairs):
        sh, sw = scales[i]
        ref = orc.deform_conv_forward(feats[s], w, None, offs[i], None, 1, 1, 1, 1, 1, sh, sw, out_hw=FPN_SIZES[l])
        gr = orc.deform_conv_backward(feats[s], w, offs[i], None, gos[i], 1, 1, 1, 1, 1, sh, sw)
        gw_ref += gr['gw']
        gx_ref[s] += gr['gx']
        errs[f'out{l}<{s}'] = _report(f'pyr/out{l}<{s}', outs[i], ref)
        errs[f'goff{l}<{s}'] = _report(f'pyr/goff{l}<{s}', grads[6 + i], gr['goff'])
    for s in range(5):
        errs[f'gx{s}'] = _report(f'pyr/gx{s}', grads[1 + s], gx_ref[s])
    errs['gw'] = _report('pyr/gw', grads[0], gw_ref)
    print(choice, {k: f'{v:.1e}' for k, v in errs.items()})
    assert all(e < TOL for e in errs.values()), errs


def test_gather_backward_is_deterministic():
    """grad_input of the default path is formed without atomics in a fixed summation order: two runs are bitwise
    equal (grad_offset / grad_mask never used atomics)."""
    from lsnet_amd import ops
    dev = _dev()
    case = dict(name='det', C=256, Co=256, hw=(50, 84))
    x, w, b, off, mask, go, cfg = _make(case, dev, seed=4)
    a = _dcn_all(ops, x, w, b, off, mask, go, cfg, dev)
    c = _dcn_all(ops, x, w, b, off, mask, go, cfg, dev)
    for k in ('out', 'gx', 'goff', 'gmask'):
        assert torch.equal(a[k], c[k]), k


@pytest.mark.gpu
@pytest.mark.parametrize('mode', ['bf16x6', 'fp32'])
def test_grouped_backward_is_deterministic(mode):
    """BASELINE config 4 (ResNeXt-101 64x4d-DCN: 64 groups): grad_input / grad_offset / grad_mask through
    dcn_gcol_grouped_kernel + the anchor-list gather -- no atomics, in EVERY math mode (round 3: fp32 atomic scatter):
    two runs are bitwise equal."""
    from lsnet_amd import _lib, ops
    dev = _dev()
    before = _lib.get_math_mode()
    _lib.set_math_mode(mode)
    try:
        case = dict(name='det_g64', C=512, Co=512, groups=64, stride=2, hw=(50, 84), bias=False)
        x, w, b, off, mask, go, cfg = _make(case, dev, seed=4)
        a = _dcn_all(ops, x, w, b, off, mask, go, cfg, dev)
        c = _dcn_all(ops, x, w, b, off, mask, go, cfg, dev)
        for k in ('out', 'gx', 'goff', 'gmask'):
            assert torch.equal(a[k], c[k]), k
    finally:
        _lib.set_math_mode(before)


@pytest.mark.gpu
def test_anchor_sums_by_channel_block_give_the_same_bits():
    """Round 6: for layers of more than 256 channels the per-anchor sums of the backward-data pass put the 256-channel blocks
    on blockIdx.y (dcn_gather_kernels.h AnchorArgs::ncb) and dcn_offgrad_kernel adds the blocks' corner dot products in block
    order -- the order in which one wave used to accumulate them (debug bit 16): grad_input, grad_offset and grad_mask are
    bitwise the same in both forms."""
    from lsnet_amd import _lib, ops
    dev = _dev()
    case = dict(name='cb_g64', C=1024, Co=1024, groups=64, hw=(25, 42), bias=False)
    x, w, b, off, mask, go, cfg = _make(case, dev, seed=6)
    try:
        a = _dcn_all(ops, x, w, b, off, mask, go, cfg, dev)
        _lib.load().lsn_debug_phase_clocks(None, 1 << 16)
        c = _dcn_all(ops, x, w, b, off, mask, go, cfg, dev)
    finally:
        _lib.load().lsn_debug_phase_clocks(None, 0)
    for k in ('gx', 'goff', 'gmask'):
        assert torch.equal(a[k], c[k]), k


@pytest.mark.gpu
@pytest.mark.parametrize('case', [dict(name='det_dense', C=256, Co=256, hw=(25, 42)),
                                  dict(name='det_dense_s2', C=128, Co=128, stride=2, hw=(50, 84), bias=False),
                                  dict(name='det_pyr', C=256, Co=256, mask=False, hw=(25, 42), dst=(13, 21))],
                         ids=lambda c: c['name'])
def test_exact_mode_is_deterministic(case):
    """Round 6: LSN_MATH_FP32 no longer scatters with fp32 atomics.  Its data gradients take the exact
    fmaf column gradients + the anchor-list gather of the default mode, its weight gradient leaves per-split partial tiles for
    the ordered reduce: every output of a dense deformable call is bitwise equal on two runs -- weight and bias gradient
    included -- and stays within 2e-6 of the fp32-equivalent default mode's."""
    from lsnet_amd import _lib, ops
    dev = _dev()
    before = _lib.get_math_mode()
    x, w, b, off, mask, go, cfg = _make(case, dev, seed=5)
    try:
        _lib.set_math_mode('fp32')
        a = _dcn_all(ops, x, w, b, off, mask, go, cfg, dev)
        c = _dcn_all(ops, x, w, b, off, mask, go, cfg, dev)
        _lib.set_math_mode('bf16x6')
        d = _dcn_all(ops, x, w, b, off, mask, go, cfg, dev)
    finally:
        _lib.set_math_mode(before)
    for k in a:
        assert torch.equal(a[k], c[k]), f'{k}: two exact-mode runs differ'
        assert float((a[k] - d[k]).abs().max()) <= 5e-6 * float(d[k].abs().max()) + 1e-30, k


@pytest.mark.gpu
@pytest.mark.parametrize('k,s,p,ceil,cip', [(3, 2, 1, False, True), (2, 2, 0, True, False)])
def test_avg_pool_backward(k, s, p, ceil, cip):
    """ops/pool.py: the Res2Net pools (Bottle2neck's 3x3 stride-2 pool of the last scale, res2net.py:80-99; the avg_down
    shortcut's 2x2 ceil-mode pool, :213-231) on a strided channel slice of a channels-last tensor against the host, forward
    and backward.  (ATen's own channels-last backward is off by 0.68 of the range for the first configuration on this
    stack -- which is why these run through the NCHW kernels.)"""
    from lsnet_amd.ops.pool import avg_pool_nchw
    torch.manual_seed(0)
    pool = torch.nn.AvgPool2d(k, s, p, ceil_mode=ceil, count_include_pad=cip)
    x = torch.randn(2, 832, 13, 17)
    xh = x.clone().requires_grad_()
    yh = pool(xh[:, 624:])
    go = torch.randn_like(yh)
    yh.backward(go)
    xd = x.to(_dev()).contiguous(memory_format=torch.channels_last).requires_grad_()
    yd = avg_pool_nchw(xd[:, 624:], pool)
    yd.backward(go.to(_dev()).contiguous(memory_format=torch.channels_last))
    assert torch.allclose(yd.cpu(), yh.detach(), rtol=1e-6, atol=1e-6)
    assert torch.allclose(xd.grad.cpu(), xh.grad, rtol=1e-6, atol=1e-6)
