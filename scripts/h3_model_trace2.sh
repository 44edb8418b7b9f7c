#!/bin/bash
source /root/repo/scripts/h3_model_trace.sh.inc
export HL_B=1,4,8; python $R/scripts/fwd_time.py 2>&1 | grep "B="
export HL_B=4
run h2
