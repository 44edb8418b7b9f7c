#!/bin/bash
# A/B (same box): lower bound of the small-tile 1x1 kernel (HL_H2_MIN_BLOCKS) now that it has split-K
cd /root/repo
export HL_B=1,4,8
for v in 12 4 2 12 4 1; do
  echo "== HL_H2_MIN_BLOCKS=$v"
  HL_H2_MIN_BLOCKS=$v timeout 300 python scripts/fwd_time.py 2>&1 | grep "B="
done
