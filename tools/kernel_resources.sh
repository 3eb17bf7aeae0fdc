#!/bin/bash
# register / LDS / spill figures of the kernels of one translation unit whose name matches a pattern:
#   tools/kernel_resources.sh dcn.hip dcn_fwd_mm_kernel
cd "$(dirname "$0")/../lsnet_amd/csrc"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=fast -fno-slp-vectorize -Wno-unused-result ${EXTRA_FLAGS:-} -c "$1" -o /dev/null \
  -Rpass-analysis=kernel-resource-usage 2>&1 | python3 -c "
import re, sys, subprocess
pat = sys.argv[1]
cur = None
rows = {}
for ln in sys.stdin:
    m = re.search(r'remark: +(.*?) \[-Rpass-analysis', ln)
    if not m: continue
    t = m.group(1).strip()
    if t.startswith('Function Name:') or t.startswith('Name:'):
        cur = t.split(':', 1)[1].strip(); rows[cur] = {}
    elif cur and ':' in t:
        k, v = t.split(':', 1); rows[cur][k.strip()] = v.strip()
for name, r in rows.items():
    try: dem = subprocess.run(['/opt/rocm/lib/llvm/bin/llvm-cxxfilt', name], capture_output=True, text=True).stdout.strip()
    except Exception: dem = name
    if pat not in dem: continue
    print(f\"{dem.split('(')[0][:110]:110s} VGPR {r.get('VGPRs','?'):>4s} AGPR {r.get('AGPRs','?'):>3s} SGPR {r.get('TotalSGPRs','?'):>4s} scratch {r.get('ScratchSize [bytes/lane]','?'):>4s} \"
          f\"spillS {r.get('SGPRs Spill','?'):>3s} spillV {r.get('VGPRs Spill','?'):>3s} occ {r.get('Occupancy [waves/SIMD]','?')} LDS {r.get('LDS Size [bytes/block]','?')}\")
" "$2"
