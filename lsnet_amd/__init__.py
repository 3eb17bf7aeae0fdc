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
# Two hardware queues per process.  A training process has five streams that carry kernels: the step's, the package's second
# stream (ops/streams.py), the library's own (csrc/dcn.hip), and two of RCCL's once a process group exists.  With the runtime's
# default number of hardware queues the ORDER in which those streams first submit work decides which of them share a queue, and
# in eight of ten orders the step of a process with a (one-rank) RCCL group ran 49 ms instead of 31.5 (profiles/r6_rccl_streams.txt);
# with 8 queues six of ten.  With 2 every order runs 32.1 - 32.3 ms and the step without a process group is unchanged (31.25 vs
# 31.30 ms, profiles/r6_side_wgrad.txt).  Read at runtime initialisation; a user setting wins.
_os.environ.setdefault('GPU_MAX_HW_QUEUES', '2')
