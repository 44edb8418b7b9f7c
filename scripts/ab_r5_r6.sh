#!/bin/bash
# Developer A / B on ONE box (boxes differ by 3-5 %): the round-5 tree (unpacked in _r5tree by `git archive ab01fed | tar -x -C _r5tree` + build) against this tree
# and its variant libraries (HL_LIB_PATH), forward times, then a kernel trace of each at B = 4 summarised per kernel.
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
O=gpurun_out/ab; rm -rf $O; mkdir -p $O
run() { echo "== $1"; shift; "$@" 2>&1 | tail -2; }
[ -d _r5tree ] && (cd _r5tree && run "round 5 tree" env HL_B=1,4 python scripts/fwd_time.py)
for v in libhumanliff_hip.so $HL_VARIANTS; do run "r6 [$v]" env HL_LIB_PATH=$PWD/humanliff_amd/$v HL_B=1,4 python scripts/fwd_time.py; done
[ -d _r5tree ] && (cd _r5tree && run "round 5 tree" env HL_B=1,4 python scripts/fwd_time.py)
if [ -n "$HL_TRACE" ]; then
  [ -d _r5tree ] && (cd _r5tree && HL_NO_OVERLAP=1 HL_B=4 rocprofv3 --kernel-trace --stats -d ../$O/t5 -- python scripts/fwd_time.py > ../$O/t5.log 2>&1)
  HL_NO_OVERLAP=1 HL_B=4 rocprofv3 --kernel-trace --stats -d $O/t6 -- python scripts/fwd_time.py > $O/t6.log 2>&1
  for t in t5 t6; do python scripts/rocpd_summary.py $(ls $O/$t/*/*results.db | head -1) $O/$t.md > /dev/null; rm -rf $O/$t; done
fi
