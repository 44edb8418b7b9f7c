#!/bin/bash
# A/B of the 3x3 fp16x2 kernel against the Winograd kernels by level (same box)
cd /root/repo
export HL_B=1,4,8
mkdir -p gpurun_out
{
echo "== tests (conv + fullsize oracle)"; timeout 900 python -m pytest tests/test_unet_gpu.py tests/test_fullsize_gpu.py -m gpu -x -q 2>&1 | tail -8
for cfg in "-1 0" "48 1099511627776" "600 1099511627776" "48 600" "300 600" "100 300" "48 100"; do
  set -- $cfg
  echo "== HL_H2_CONV3_MIN_BLOCKS=$1 MAX=$2"
  HL_H2_CONV3_MIN_BLOCKS=$1 HL_H2_CONV3_MAX_BLOCKS=$2 timeout 300 python scripts/fwd_time.py 2>&1 | grep "B="
done
} > gpurun_out/h3_ab.log 2>&1
