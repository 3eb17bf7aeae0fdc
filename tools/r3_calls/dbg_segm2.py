"""segm head fixture: feature gradients under different conv arithmetic (own split kernels / ATen / exact mode), against the
fixture sample and against each other."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from tests import golden_cases as gc, golden_util as gu
import lsnet_amd.ops.conv as cv
from lsnet_amd import _lib
dev = torch.device('cuda:0')

def grads(task='segm'):
    ref = gc.load(f'head_{task}')
    head = gc.build_head(task, dev).to(memory_format=torch.channels_last); head.train()
    feats = [f.to(dev).contiguous(memory_format=torch.channels_last).requires_grad_() for f in gu.head_inputs(11)]
    outs = head(feats)
    boxes, labels, extremes, masks, kps, metas = gc.gt_for(task, dev)
    losses = head.loss(*outs, boxes, None, None, masks, labels, metas)
    sum(sum(v) for v in losses.values()).backward()
    return ref, [f.grad.detach().clone() for f in feats], [o.detach() for lv in outs for o in lv if o is not None]

def frac(prefix, t, ref, tol=2e-4):
    s = gu.summary(t, 13); want = ref[f'{prefix}/sample']; sc = max(float(np.abs(want).max()), 1e-12)
    rel = np.abs(s['sample'] - want) / sc
    return float((rel > tol).mean()), float(rel.max())

ref, g_own, o_own = grads()
orig = cv.hip_conv_ok
cv.hip_conv_ok = lambda *a, **k: False
_, g_aten, o_aten = grads()
_lib.set_math_mode('fp32')
_, g_exact, o_exact = grads()
_lib.set_math_mode('bf16x6'); cv.hip_conv_ok = orig
for i in range(5):
    print(f'level {i}: vs fixture  own {frac(f"grad/feat/{i}", g_own[i], ref)}  aten {frac(f"grad/feat/{i}", g_aten[i], ref)}  exact {frac(f"grad/feat/{i}", g_exact[i], ref)}')
    d = (g_own[i] - g_exact[i]).abs(); sc = g_exact[i].abs().max()
    bad = (d > 2e-4 * sc)
    print(f'          own vs exact: {100 * float(bad.float().mean()):.2f}% of ALL elements beyond 2e-4, worst {float(d.max() / sc):.2e}; '
          f'pixels touched {int(bad.any(1).sum())} of {bad.shape[0] * bad.shape[2] * bad.shape[3]}')
print('forward outputs own vs exact, max rel:', max(float((a - b).abs().max() / b.abs().max()) for a, b in zip(o_own, o_exact)))
