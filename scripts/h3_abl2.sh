#!/bin/bash
source /root/repo/scripts/h3_abl.sh.inc
run h2
for v in $VARIANTS; do run h2_$v HL_LIB_PATH=$R/humanliff_amd/exp/lib_$v.so; done
