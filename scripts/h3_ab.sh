#!/bin/bash
# A/B (same box): the library in the tree against a previous build kept as humanliff_amd/exp/lib_prev.so
cd /root/repo
timeout 900 python -m pytest tests/test_unet_gpu.py tests/test_fullsize_gpu.py -m gpu -q -x 2>&1 | tail -3
export HL_B=1,4,8
for v in prev new prev new; do
  echo "== $v"
  if [ $v = prev ]; then HL_LIB_PATH=/root/repo/humanliff_amd/exp/lib_prev.so timeout 300 python scripts/fwd_time.py 2>&1 | grep "B="; else timeout 300 python scripts/fwd_time.py 2>&1 | grep "B="; fi
done
