#!/bin/bash
# A/B (same box): split-K on the small-tile fp16x2 3x3 kernel for layers below the 100-workgroup threshold (HL_H2_CONV3_SPLIT_MIN = smallest layer, in 256-pixel x 192-channel units)
cd /root/repo
timeout 900 python -m pytest tests/test_unet_gpu.py -m gpu -q -x 2>&1 | tail -2
export HL_B=1,4,8
for v in -1 8 16 32 4 -1 8; do
  echo "== HL_H2_CONV3_SPLIT_MIN=$v"
  HL_H2_CONV3_SPLIT_MIN=$v timeout 300 python scripts/fwd_time.py 2>&1 | grep "B="
done
