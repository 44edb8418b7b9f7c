#!/bin/bash
# developer A/B: build k_conv_wino4w with the given -D flags, run the K sweep under rocprofv3 on the GPU box, print the launch times
# usage: scripts/w4w_ab.sh "-DHL_W4W_SCALAR_FMA=1"
set -e
cd "$(dirname "$0")/.."
touch humanliff_amd/csrc/hl_conv_wino4w.hip
HL_W4W_FLAGS="$1" python -m humanliff_amd.build > /dev/null
/usr/local/graft/bin/gpurun --timeout 600 -- 'cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT; HL_WINO4W=1 rocprofv3 --kernel-trace -d /tmp/ks -- python scripts/wino4_ksweep.py > /dev/null 2>&1; python scripts/rocpd_list.py /tmp/ks k_conv_wino4w | awk "NR%4==3 || NR%4==0" | cut -c1-40 | tr "\n" " "' 2>&1 | tail -2
