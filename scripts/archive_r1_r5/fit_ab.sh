#!/bin/bash
# A/B (same box) of the fitting step: one configuration of environment knobs ("NAME=value ...") per line on stdin, e.g. HL_FIT_FP32=1 (the fp32-MFMA kernels of rounds 1-4)
cd /root/repo
while read -r cfg; do
  echo "== $cfg"
  env $cfg timeout 600 python bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-bf16x3-leg --no-parity --no-train --no-render --no-e2e --no-batch-sweep --sustained-steps 0 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); f=d['fit']
print(f['value'], f['ms_per_iteration'], {k: f['stages_ms_per_subject'][k] for k in ('eval_acts_coarse','mlp_backward_coarse','plane_grads','weight_grads')}, f['one_stream']['value'], f['uniforms_on_device']['value'])"
done
