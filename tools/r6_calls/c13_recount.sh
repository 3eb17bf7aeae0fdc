cd "${GRAFT_REPO_ROOT:-/root/repo}"
bash tools/pmc_step_shapes.sh r6 > gpurun_out/r6_pmc_step.log 2>&1
tail -4 gpurun_out/r6_pmc_hbm.txt
