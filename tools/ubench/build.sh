#!/bin/bash
# Rebuilds every harness whose source or the C header is newer than its binary (a stale binary built against an older
# lsn_conv_level / lsn_dcn_level layout hands the library garbage pointers).
cd "$(dirname "$0")"
for t in wgrad_ab dcn_step conv_step norm_step tile_sweep; do
    if [ ! -x $t ] || [ $t.hip -nt $t ] || [ ../../include/lsnet_hip.h -nt $t ] || [ dcn_ref.h -nt $t ]; then
        hipcc --offload-arch=gfx950 -O2 $t.hip -o $t -ldl || exit 1
        echo "built $t"
    fi
done
