#!/bin/bash
# A/B of the 3x3 fp16x2 kernel's dispatch (same box): window by workgroup count, and no upper bound in the decoder (HL_H2_CONV3_SOLO)
cd /root/repo
export HL_B=1,4,8
for cfg in "100 300 0" "100 300 1" "-1 0 0" "100 300 1" "100 300 0"; do
  set -- $cfg
  echo "== HL_H2_CONV3_MIN_BLOCKS=$1 MAX=$2 SOLO=$3"
  HL_H2_CONV3_MIN_BLOCKS=$1 HL_H2_CONV3_MAX_BLOCKS=$2 HL_H2_CONV3_SOLO=$3 timeout 300 python scripts/fwd_time.py 2>&1 | grep "B="
done
