"""Which of the round-4 glue changes moves a gradient?  One forward + backward of LSNet R-50 bbox at the benchmark size per
arm, every parameter gradient against arm 0 (all switches off): the list-building side stream of the deformable backward (debug bit
21), lsn_topk_columns, the one-launch backward of _split_px, the concatenated pyramid outputs."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from lsnet_amd import _lib
from lsnet_amd.data import synthetic_batch
from lsnet_amd.model_zoo import build_lsnet
from lsnet_amd.core import assigners
from lsnet_amd.models.dense_heads import ls_head
from lsnet_amd.ops import dcn as dcn_ops

dev = 'cuda:0'
lib = _lib.load()
real_topk, real_split = assigners.topk_columns, ls_head.LSHead._split_px
real_fm = dcn_ops.PyramidDeformConv.forward_multi


def torch_topk(x, k, segments=None, largest=False):
    return real_topk(x.cpu(), k, segments, largest)[0].to(x.device), real_topk(x.cpu(), k, segments, largest)[1].to(x.device)


def no_concat(self, xs, offsets, scales, weight=None, concat=0):
    outs = real_fm(self, xs, offsets, scales, weight, 0)
    if concat and concat > 1:
        return [torch.cat(outs[j:j + concat], dim=1) for j in range(0, len(outs), concat)]
    return outs


def run(bits, topk, split, concat):
    lib.lsn_debug_phase_clocks(None, bits)
    assigners.topk_columns = real_topk if topk else torch_topk
    ls_head.LSHead._split_px = staticmethod(real_split if split else ls_head._split_px_views)
    dcn_ops.PyramidDeformConv.forward_multi = real_fm if concat else no_concat
    torch.manual_seed(3)
    model, cfg = build_lsnet('bbox', 'r50')
    model = model.to(dev).to(memory_format=torch.channels_last).train()
    data = synthetic_batch('bbox', 2, 800, 1344, seed=1234, device=dev)
    losses = model(**data)
    loss = sum(v if torch.is_tensor(v) else sum(v) for k, v in losses.items() if 'loss' in k)
    loss.backward()
    torch.cuda.synchronize()
    return float(loss), {n: p.grad.detach().clone() for n, p in model.named_parameters() if p.grad is not None}


OFF = 1 << 21      # (bit 19 switched the tail experiment, since removed)
arms = [('all off', OFF, 0, 0, 0), ('all off again', OFF, 0, 0, 0), ('side lists', 0, 0, 0, 0),
        ('topk', OFF, 1, 0, 0), ('split_px', OFF, 0, 1, 0), ('concat', OFF, 0, 0, 1), ('all on', 0, 1, 1, 1)]
base = None
for name, bits, tk, sp, cc in arms:
    l, g = run(bits, tk, sp, cc)
    if base is None:
        base = (l, g)
    worst = max(((float((g[n] - base[1][n]).abs().max() / base[1][n].abs().max().clamp_min(1e-30)), n) for n in g), default=(0, ''))
    ndiff = sum(1 for n in g if not torch.equal(g[n], base[1][n]))
    print(f'{name:20s} loss {l:.6f}  gradients differing from arm 0: {ndiff} of {len(g)}, worst {worst[0]:.2e} ({worst[1]})', flush=True)
