"""lsnet_amd -- MI355X-native (gfx950) LSNet hot path behind the reference's mmdet-style API.

The compute path is hand-written HIP in lsnet_amd/csrc (C ABI: include/lsnet_hip.h); this
package is the host-side mirror of the reference's operator / registry interface for that path.
"""
__version__ = '0.1.0'
