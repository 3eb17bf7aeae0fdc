#!/bin/bash
# round 6: stream-K pieces planned for the workgroups a tile shape really holds per CU (registers / LDS allow three of the
# 64 x 128, 128 x 64 and 128 x 32 tiles) instead of 512 for every shape.  LSNET_SLOTS_A / _B / _C of the LSNET_AB build.
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
out=gpurun_out/r6_slots.txt
: > $out
run() {
  echo "== A=$1 B=$2 C=$3" >> $out
  LSNET_SLOTS_A=$1 LSNET_SLOTS_B=$2 LSNET_SLOTS_C=$3 LSNET_SO=lsnet_amd/csrc/ab_env.so timeout 100 tools/ubench/conv_step 10 >> $out 2>&1
}
run 512 512 512
run 768 512 512
run 512 768 512
run 512 512 768
run 768 768 768
run 512 512 512
run 768 768 768
run 640 640 640
grep -E "^==|per step" $out
