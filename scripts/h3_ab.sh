#!/bin/bash
# A/B of the 3x3 fp16x2 kernel's dispatch window (workgroups of 256 pixels x 192 channels), same box
cd /root/repo
export HL_B=1,4,8
for cfg in "-1 0" "100 300" "100 1100" "100 1099511627776" "48 1099511627776" "100 300"; do
  set -- $cfg
  echo "== HL_H2_CONV3_MIN_BLOCKS=$1 MAX=$2"
  HL_H2_CONV3_MIN_BLOCKS=$1 HL_H2_CONV3_MAX_BLOCKS=$2 timeout 300 python scripts/fwd_time.py 2>&1 | grep "B="
done
