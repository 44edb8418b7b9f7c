#!/bin/bash
# Run on the GPU box from the repo root: SQ counters (scripts/pmc_wino.txt, three passes) of a probe script's conv kernels.
# usage: bash scripts/pmc_probe.sh <out-dir> <python script and args...>
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
O=$1; shift; rm -rf $O; mkdir -p $O
i=0
while read -r line; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc ${line#pmc: } --output-format csv -d $O/sq$i -- python "$@" > $O/sq$i.log 2>&1
done < scripts/pmc_wino.txt
python scripts/pmc_probe_summary.py $O/sq1 $O/sq2 $O/sq3 | tee $O/summary.txt
rm -rf $O/sq1 $O/sq2 $O/sq3
