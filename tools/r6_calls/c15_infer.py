"""round 6: pose inference (bs 4) with / without stream-K pieces in the deformable forward (debug bit 19), alternating"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
sys.argv = sys.argv[:1]
import bench
from lsnet_amd import _lib
dev = torch.device('cuda:0')
for rep in range(3):
    for name, bits in (('pieces', 0), ('whole tiles', 1 << 19)):
        _lib.load().lsn_debug_phase_clocks(None, bits)
        r = bench.infer_leg(dev)
        print(name, round(r['ms_per_batch'], 3), 'ms per batch of 4', flush=True)
_lib.load().lsn_debug_phase_clocks(None, 0)
