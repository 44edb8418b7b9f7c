#!/bin/bash
# A/B (same box): threshold of the stride-2 fp16x2 kernel (HL_H2_CONV3S2_MIN_BLOCKS, workgroups' worth of 256 output pixels x 192 channels)
cd /root/repo
export HL_B=1,4,8
for v in 32 8 16 64 -1 32; do
  echo "== HL_H2_CONV3S2_MIN_BLOCKS=$v"
  HL_H2_CONV3S2_MIN_BLOCKS=$v timeout 300 python scripts/fwd_time.py 2>&1 | grep "B="
done
timeout 1500 python -m pytest tests/test_unet_gpu.py tests/test_fullsize_gpu.py tests/test_e2e_gpu.py tests/test_unet_train_gpu.py -m gpu -q -x 2>&1 | tail -3
