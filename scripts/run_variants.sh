#!/bin/bash
# On the GPU box: run a probe under rocprofv3 for each library variant of scripts/build_variants.sh and list the wino4 launch times.
# usage: bash scripts/run_variants.sh "<probe script and args>" name1 name2 ...
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
probe=$1; shift
cp humanliff_amd/libhumanliff_hip.so /tmp/lib_keep.so
for v in "$@"; do
  cp humanliff_amd/exp/lib_$v.so humanliff_amd/libhumanliff_hip.so
  rm -rf gpurun_out/var_$v
  timeout 90 rocprofv3 --kernel-trace -d gpurun_out/var_$v -o w -- python $probe > gpurun_out/var_$v.log 2>&1
  echo "$v rc=$? : $(python scripts/rocpd_list.py gpurun_out/var_$v ${KPAT:-k_conv_wino4} 2>/dev/null | grep -v pack | awk '{print $1}' | tr '\n' ' ')"
done
cp /tmp/lib_keep.so humanliff_amd/libhumanliff_hip.so
