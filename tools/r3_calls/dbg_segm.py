"""Which operator makes the segm head fixture's feature gradient wrong?  Run the fixture case with own ops swapped for ATen
one family at a time."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from tests import golden_cases as gc
import lsnet_amd.ops.conv as cv
dev = torch.device('cuda:0')

def run(tag):
    try:
        gc.head_case('segm', dev, True)
        print(tag, 'PASS', flush=True)
    except AssertionError as e:
        print(tag, 'FAIL', str(e)[:160], flush=True)

run('own everything')
orig_ok = cv.hip_conv_ok
cv.hip_conv_ok = lambda *a, **k: False
run('dense convs -> ATen')
cv.hip_conv_ok = orig_ok
# forward own, backward-data via ATen
orig_bwd = cv._ConvMultiFn.backward
os.environ['LSNET_WGRAD_OLD'] = '1'
# conv_multi only -> per-level singles
orig_multi = cv.Conv2d.forward_multi
def single(self, xs, relu=False):
    outs = [self._run(x, self.weight) for x in xs]
    return [torch.relu(o) for o in outs] if relu else outs
cv.Conv2d.forward_multi = single
run('multi-level conv launches -> per-level')
cv.Conv2d.forward_multi = orig_multi
for t in ('bbox', 'segm'):
    try:
        gc.head_case(t, dev, True); print(t, 'again PASS')
    except AssertionError as e:
        print(t, 'again FAIL', str(e)[:120])
