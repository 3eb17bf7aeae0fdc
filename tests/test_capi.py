"""The C-ABI library loads and exports every symbol include/lsnet_hip.h declares (no GPU needed)."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope='module')
def built_lib():
    from lsnet_amd.csrc import build
    return build.build()


def _declared_symbols():
    text = open(os.path.join(ROOT, 'include', 'lsnet_hip.h')).read()
    text = re.sub(r'/\*.*?\*/', '', text, flags=re.S)
    return sorted(set(re.findall(r'\b(lsn_[a-z0-9_]+)\s*\(', text)))


def test_header_symbols_are_exported(built_lib):
    lib = ctypes.CDLL(built_lib)
    names = _declared_symbols()
    assert len(names) >= 19
    for n in names:
        assert hasattr(lib, n), f'{n} declared in lsnet_hip.h but not exported'


def test_loader_export_list_matches_header(built_lib):
    from lsnet_amd import _lib
    assert sorted(_lib.EXPORTS) == _declared_symbols()
    lib = _lib.load()
    assert lib.lsn_version() >= 100
    assert isinstance(lib.lsn_last_error(), bytes)


def test_argument_checks_need_no_gpu(built_lib):
    """Shape validation happens before any device work (mirrors the reference's TORCH_CHECKs)."""
    from lsnet_amd import _lib
    lib = _lib.load()
    # kernel size 0 -> LSN_ERR_INVALID with the reference's message
    rc = lib.lsn_deform_conv_forward(None, None, None, None, 1, 4, 8, 8, 4, 0, 0, 1, 1, 0, 0, 1, 1, 1, 1, 1, None)
    assert rc == -1 and b'kernel size should be greater than zero' in lib.lsn_last_error()
    # input smaller than kernel (deform_conv_cuda.cpp:128)
    rc = lib.lsn_deform_conv_forward(None, None, None, None, 1, 4, 2, 2, 4, 3, 3, 1, 1, 1, 1, 1, 1, 1, 1, 1, None)
    assert rc == -1 and b'input image is smaller than kernel' in lib.lsn_last_error()
    with pytest.raises(RuntimeError):
        _lib.check(rc)


def test_product_ops_refuse_cpu_tensors():
    """No CPU fallback in the product: like the reference (deform_conv.py:46-47) CPU tensors raise in the device-only ops."""
    import torch
    from lsnet_amd import ops
    x = torch.zeros(1, 4, 5, 5)
    with pytest.raises(NotImplementedError):
        ops.deform_conv(x, torch.zeros(1, 18, 5, 5), torch.zeros(4, 4, 3, 3), padding=1)
    with pytest.raises(NotImplementedError):
        ops.sigmoid_focal_loss(torch.zeros(4, 3), torch.zeros(4, dtype=torch.long))
    # NMS is the one op the reference itself serves on the host (nms_wrapper.py:33-37 -> nms_cpu): CPU tensors take the
    # host library's counterpart, they never reach (or replace) the device kernel
    dets, keep = ops.nms(torch.tensor([[0., 0., 2., 2., .9], [0., 0., 2., 2., .8], [5., 5., 6., 6., .7]]), 0.5)
    assert keep.tolist() == [0, 2] and dets.shape == (2, 5)
