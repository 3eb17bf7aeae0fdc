"""Device dispatch of the native ops.

The product registers exactly one backend: 'cuda' (the HIP device under PyTorch-ROCm), backed by
liblsnet_hip.so.  Tensors on any other device raise NotImplementedError, as the reference does
for non-CUDA tensors (mmdet/ops/dcn/deform_conv.py:46-47,136-137,221-222).  Tests and
bench.py's cpu_baseline leg may register a 'cpu' backend (the oracle) through
`register_backend`; nothing inside lsnet_amd does.
"""
_BACKENDS = {}


def register_backend(device_type, impl):
    _BACKENDS[device_type] = impl


def unregister_backend(device_type):
    _BACKENDS.pop(device_type, None)


def get_backend(tensor):
    dev = tensor.device.type
    impl = _BACKENDS.get(dev)
    if impl is None:
        if dev == 'cuda':
            from . import hip_backend  # registers itself; raises RuntimeError if the .so is absent
            impl = _BACKENDS['cuda']
        else:
            raise NotImplementedError(
                f'lsnet_amd native ops run on the HIP device only (got a {dev} tensor)')
    return impl
