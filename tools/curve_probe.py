"""Prints the per-iteration relative deviation of the training curve from the reference fixture on this device, for both
arithmetic modes (sets the tolerance bands of tests/test_golden_gpu.py::test_training_curve).  python tools/curve_probe.py"""
import os
import sys

os.environ.setdefault('MIOPEN_FIND_MODE', '2')
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402

from lsnet_amd import _lib  # noqa: E402
from tests import golden_cases as gc  # noqa: E402

dev = torch.device('cuda:0' if torch.cuda.is_available() else 'cpu')
for mode in ('fp32', 'bf16x3'):
    if dev.type == 'cuda':
        _lib.set_math_mode(mode)
    try:
        gc.train_curve_case(dev, early_tol=-1.0, late_tol=-1.0, rtol_weight=1.0, channels_last=dev.type == 'cuda')
    except AssertionError as e:
        k, got, want, rel = e.args[0]
        print(mode, k, 'rel', np.array2string(rel, precision=2, separator=','), flush=True)
