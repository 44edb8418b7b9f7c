#!/bin/bash
# same-box A/B of the fp16x2 evaluate kernel: log2-domain softplus (default) against the natural-log form, and the VALU-per-MFMA count of its schedule
cd /root/repo
echo "== default (log2-domain, K=5)"; python scripts/render_products_probe.py 2>&1 | grep -E "fp16x2|bf16x3"
for v in nolog2 k3 k4 k6; do echo "== $v"; HL_LIB_PATH=/root/repo/humanliff_amd/exp/lib_$v.so python scripts/render_products_probe.py 2>&1 | grep -E "fp16x2"; done
echo "== default again"; python scripts/render_products_probe.py 2>&1 | grep -E "fp16x2"
timeout 900 python -m pytest tests/test_render_gpu.py -m gpu -q -x 2>&1 | tail -4
