"""lsnet_amd -- MI355X-native (gfx950) LSNet hot path behind the reference's mmdet-style API.

The compute path is hand-written HIP in lsnet_amd/csrc (C ABI: include/lsnet_hip.h); this
package is the host-side mirror of the reference's operator / registry interface for that path.
"""
__version__ = '0.1.0'

import os as _os

# ROCm 7.2 hipGraph workaround.  With the runtime's AQL "graph packet capture" fast path (the default), replaying
# the captured forward+backward graph (runner/graph_step.py, ~5000 kernel nodes) after eager kernels have written
# the parameters ends in HSA_STATUS_ERROR_EXCEPTION 0x1016 a few replays later -- reproduced with plain
# per-tensor torch ops between replays, not reproducible with AMD_SERIALIZE_KERNEL=3, gone with the fast path off
# (tools/try_capture4.py).  The variable is read when the HIP runtime initialises, i.e. at the first device call,
# so setting it at import time is early enough; an explicit user setting wins.
_os.environ.setdefault('DEBUG_CLR_GRAPH_PACKET_CAPTURE', '0')
# Kernel arguments in device memory.  The step is ~1 050 dispatches, most of them short: with HIP_FORCE_DEV_KERNARG=0 it takes
# 35.0 ms instead of 32.6 (profiles/r5_env_switches.txt).  1 is the default of this ROCm build; stated here so that the
# step does not depend on it (read at runtime initialisation like the switch above; a user setting wins).
_os.environ.setdefault('HIP_FORCE_DEV_KERNARG', '1')
# (GPU_MAX_HW_QUEUES is deliberately NOT set here.  With two hardware queues every order in which the process's five
# kernel-carrying streams -- the step's, the package's second one, the library's, two of RCCL's -- first submit work runs the step at
# 32.1 - 32.3 ms, where the runtime's default leaves eight of ten orders at 49 ms (profiles/r6_rccl_streams.txt); but a hipGraph
# replay of the captured step, whose branches want queues of their own, segfaults inside the runtime with two
# (tests/test_graph_gpu.py, profiles/r6_graph_queues.txt).  The package instead arranges the one order that is fast under the
# default: ops/streams.py makes the second stream the process's second submitter, DataParallelModel warms the library's stream
# before the first collective.)
