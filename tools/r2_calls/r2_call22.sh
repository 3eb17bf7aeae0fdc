#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export PYTHONUNBUFFERED=1
timeout 300 python tools/phase_clocks.py conv 2>&1 | grep -v amdgpu | head -60
