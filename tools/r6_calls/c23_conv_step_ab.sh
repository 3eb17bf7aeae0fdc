cd "${GRAFT_REPO_ROOT:-/root/repo}"
for v in ab_oldepi liblsnet_hip ab_oldepi liblsnet_hip; do
  LSNET_SO=lsnet_amd/csrc/$v.so timeout 100 tools/ubench/conv_step 10 > gpurun_out/cs_$v.txt 2>&1
  tail -1 gpurun_out/cs_$v.txt
done
