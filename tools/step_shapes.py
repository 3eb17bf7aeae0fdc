"""The deformable-conv launches of ONE benchmark step (LSNet R-50 bbox, B = 2, 800x1344) replayed on their own: 6 tower
launches (DCNv2 256 -> 256 over the five FPN levels) and 2 pyramid launches (15 (level, source) pairs), forward and
backward, with offsets of the magnitudes the head produces.  Used under rocprofv3 --pmc (tools/pmc_step_shapes.sh) to
count the HBM traffic of the step's own launch shapes without profiling the whole step."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from lsnet_amd import ops, _lib

dev = torch.device('cuda:0')
cl = torch.channels_last
SIZES = [(100, 168), (50, 84), (25, 42), (13, 21), (7, 11)]
LISTS = [[0, 1, 2], [1, 0, 2], [2, 1, 3], [3, 2, 4], [4, 3, 2]]
torch.manual_seed(0)
C = 256
reps = int(os.environ.get('STEP_SHAPES_REPS', '2'))


def t(*shape, scale=1.0):
    return (torch.randn(*shape, device=dev) * scale).contiguous(memory_format=cl).requires_grad_()


w = t(C, C, 3, 3, scale=0.02)
b = torch.zeros(C, device=dev, requires_grad=True)
feats = [t(2, C, h, ww) for h, ww in SIZES]
om = [t(2, 27, h, ww, scale=0.5) for h, ww in SIZES]                     # fused offsets | mask logits of a DCNv2 pack
pairs = [(l, s) for l, lst in enumerate(LISTS) for s in lst]
poffs = [t(2, 18, *SIZES[l], scale=2.0 * max(SIZES[s][0] / SIZES[l][0], 1.0)) for l, s in pairs]
scales = [(SIZES[s][0] / SIZES[l][0], SIZES[s][1] / SIZES[l][1]) for l, s in pairs]
wp = t(C, C, 3, 3, scale=0.02)

for _ in range(reps):
    outs = ops.dcn_multi(feats, om, None, w, b, 1, 1, 1, fused_om=True)
    torch.autograd.grad(outs, [w, b] + feats + om, [torch.randn_like(o) for o in outs])
    outs = ops.dcn_multi([feats[s] for _, s in pairs], poffs, None, wp, None, 1, 1, 1, scales=scales, pyramid=True)
    torch.autograd.grad(outs, [wp] + feats + poffs, [torch.randn_like(o) for o in outs])
torch.cuda.synchronize()
print('step shapes replayed', reps, 'x; math', _lib.get_math_mode())
