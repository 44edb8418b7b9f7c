#!/bin/bash
# A/B of the dispatch thresholds of the small-tile fp16x2 kernels (same box); thresholds in workgroups of 256 pixels x 192 channels
cd /root/repo
export HL_B=1,4,8 HL_H2_CONV3_MAX_BLOCKS=1099511627776
for cfg in "100 12" "48 12" "24 12" "100 12" "48 6" "100 0"; do
  set -- $cfg
  echo "== HL_H2_CONV3_MIN_BLOCKS=$1 HL_H2_MIN_BLOCKS=$2"
  HL_H2_CONV3_MIN_BLOCKS=$1 HL_H2_MIN_BLOCKS=$2 timeout 300 python scripts/fwd_time.py 2>&1 | grep "B="
done
