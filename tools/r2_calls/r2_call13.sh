#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export PYTHONUNBUFFERED=1
timeout 400 python tools/torch_prof_sites.py > gpurun_out/c13_sites.log 2>&1
tail -75 gpurun_out/c13_sites.log
