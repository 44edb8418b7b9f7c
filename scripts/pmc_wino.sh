#!/bin/bash
# SQ counters of the conv kernels inside the bench step (single stream), three rocprofv3 passes (scripts/pmc_wino.txt)
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
O=gpurun_out/pmc_wino; mkdir -p $O
i=0
while read -r line; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc ${line#pmc: } --output-format csv -d $O/p$i -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-bf16x3-leg --no-overlap --no-render --no-fit --no-parity --sustained-steps 0 > $O/p$i.log 2>&1
done < scripts/pmc_wino.txt
python scripts/pmc_sq_summary.py $O/summary.md $O/p1 $O/p2 $O/p3
python scripts/pmc_report.py $O/p2 "k_conv_wino<false>" | head -40 > $O/raw_p2.txt; rm -rf $O/p1 $O/p2 $O/p3
