#!/bin/bash
# developer A/B: build k_conv_h16 with the given -D flags and run scripts/h16_probe.py on the GPU box
set -e
cd "$(dirname "$0")/.."
touch humanliff_amd/csrc/hl_conv_h16.hip
HL_H16_FLAGS="$1" python -m humanliff_amd.build > /dev/null
/usr/local/graft/bin/gpurun --timeout 600 -- 'timeout 300 python scripts/h16_probe.py 2>&1 | grep -v amdgpu.ids | cut -c1-260' 2>&1 | grep "^N"
