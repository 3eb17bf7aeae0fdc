cd "${GRAFT_REPO_ROOT:-/root/repo}"
timeout 900 python -m pytest tests/test_ops_gpu.py tests/test_golden_gpu.py tests/test_variants_gpu.py -m gpu -q --tb=line -p no:cacheprovider -k "x101 or g4_dg2 or v2_g2 or grouped_backward or (test_one_training_step and x101) or (full_size and x101)" --durations=8 2>&1 | grep -v "Warn\|^$" | cut -c1-300 | tail -40
