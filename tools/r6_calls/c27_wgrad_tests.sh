cd "${GRAFT_REPO_ROOT:-/root/repo}"
timeout 900 python -m pytest tests/test_ops_gpu.py tests/test_resblock_gpu.py tests/test_zz_grad_sink_gpu.py -q -m gpu -x -k "conv or resblock or bottleneck or wgrad or weight_grad or deferred" 2>&1 | tail -3
