#!/bin/bash
# A/B (same box): upper bound (query tiles x batch x heads) of the key-split attention kernel with fp16x2 products
cd /root/repo
export HL_B=4,8,16
for v in 1024 1025 2049 4097 1024 2049; do
  echo "== HL_ATT_KS_MAX=$v"
  HL_ATT_KS_MAX=$v timeout 300 python scripts/fwd_time.py 2>&1 | grep "B="
done
