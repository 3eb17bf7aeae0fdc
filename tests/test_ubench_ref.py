"""The host reference of the torch-free GPU harness (tools/ubench/dcn_ref.h, used by tools/ubench/dcn_step.hip to check the
library on the GPU box) against the CPU oracle: DCNv2 with mask logits behind the offsets, DCNv1, and the pyramid form
with scales -- every output and gradient element of a small case.  CPU only; keeps the harness's checker honest."""
import ctypes
import os
import subprocess

import numpy as np
import pytest
import torch

from oracle import oracle_py as orc

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, 'tools', 'ubench', 'dcn_ref_capi.cpp')


@pytest.fixture(scope='module')
def ref_lib(tmp_path_factory):
    so = str(tmp_path_factory.mktemp('ubench') / 'libdcnref.so')
    subprocess.check_call(['g++', '-O2', '-std=c++17', '-shared', '-fPIC', '-ffp-contract=off', SRC, '-o', so])
    return ctypes.CDLL(so)


def _nhwc(t):
    return np.ascontiguousarray(t.permute(0, 2, 3, 1).numpy(), dtype=np.float32)


@pytest.mark.parametrize('name,fused,H,W,Ho,Wo', [('v2_fused_logits', True, 6, 7, 6, 7), ('v1', False, 5, 6, 5, 6),
                                                  ('pyramid_down', False, 9, 11, 5, 6), ('pyramid_up', False, 4, 5, 7, 9)])
def test_harness_reference_equals_oracle(ref_lib, name, fused, H, W, Ho, Wo):
    g = torch.Generator().manual_seed(5)
    B, C, Co, K = 2, 8, 12, 9
    sh, sw = H / Ho, W / Wo
    x = torch.randn(B, C, H, W, generator=g)
    w = torch.randn(Co, C, 3, 3, generator=g) * 0.2
    bias = torch.randn(Co, generator=g) if fused else None
    off = (torch.rand(B, 18, Ho, Wo, generator=g) * 2 - 1) * (1.5 * max(sh, 1.0))
    off[0, :, 0, 0] = 40.0          # a pixel whose nine samples all leave the map
    off[1, 0, 1, 1] = -1.0          # exactly on the open border py = -1 + (ho - 1 + i) ...
    logits = torch.randn(B, 9, Ho, Wo, generator=g) if fused else None
    mask = torch.sigmoid(logits) if fused else None
    gout = torch.randn(B, Co, Ho, Wo, generator=g)
    want_out = orc.deform_conv_forward(x, w, bias, off, mask, 1, 1, 1, 1, 1, np.float32(sh), np.float32(sw), (Ho, Wo))
    want = orc.deform_conv_backward(x, w, off, mask, gout, 1, 1, 1, 1, 1, np.float32(sh), np.float32(sw))

    och = 27 if fused else 18
    offl = torch.cat([off, logits], 1) if fused else off
    xs, offs, gouts = _nhwc(x), _nhwc(offl), _nhwc(gout)
    ws = np.ascontiguousarray(w.permute(0, 2, 3, 1).numpy(), dtype=np.float32)     # (Co, 3, 3, C)
    bs = None if bias is None else np.ascontiguousarray(bias.numpy(), dtype=np.float32)
    out = np.zeros((B, Ho, Wo, Co)); gx = np.zeros((B, H, W, C)); goff = np.zeros((B, Ho, Wo, och))
    gw = np.zeros((Co, 3, 3, C)); gb = np.zeros(Co)
    fp = lambda a: None if a is None else a.ctypes.data_as(ctypes.c_void_p)
    ref_lib.dcnref_all(B, H, W, Ho, Wo, och, ctypes.c_float(sh), ctypes.c_float(sw), C, Co, fp(xs), fp(offs), fp(gouts),
                       fp(ws), fp(bs), fp(out), fp(gx), fp(goff), fp(gw), fp(gb))

    def close(got, ref, what):
        ref = ref.numpy().astype(np.float64)
        err = np.abs(got - ref).max() / max(np.abs(ref).max(), 1e-12)
        assert err < 2e-5, (name, what, err)       # the oracle accumulates in float, the harness reference in double

    close(out.transpose(0, 3, 1, 2), want_out, 'out')
    close(gx.transpose(0, 3, 1, 2), want['gx'], 'gx')
    close(goff[..., :18].transpose(0, 3, 1, 2), want['goff'], 'goff')
    close(gw.transpose(0, 3, 1, 2), want['gw'], 'gw')
    close(gb, want['gb'], 'gb')
    if fused:
        glogit = want['gmask'] * mask * (1 - mask)
        close(goff[..., 18:].transpose(0, 3, 1, 2), glogit, 'gmask (logits)')
