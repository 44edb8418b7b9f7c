#!/bin/bash
# A/B (same box): GroupNorm(+SiLU) applied by k_conv_h2s while staging (HL_H2_FUSE_GN=1) against the two-plane pre-pass (0)
cd /root/repo
timeout 900 python -m pytest tests/test_unet_gpu.py tests/test_fullsize_gpu.py -m gpu -q -x 2>&1 | tail -3
export HL_B=1,4,8
for v in 0 1 0 1; do
  echo "== HL_H2_FUSE_GN=$v"
  HL_H2_FUSE_GN=$v timeout 300 python scripts/fwd_time.py 2>&1 | grep "B="
done
