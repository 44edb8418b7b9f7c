#!/bin/bash
# Run on the GPU box from the repo root: regenerates the bench line, the kernel traces and the PMC traffic summary
# under gpurun_out/ (copy what should be judged into profiles/).
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
O=gpurun_out/refresh; mkdir -p $O
python bench.py > $O/bench_full.log 2>&1; grep '^{"metric"' $O/bench_full.log | tail -1 > $O/bench.json
rocprofv3 --kernel-trace --stats -d $O/tr_a -- python bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-bf16x3-leg > $O/tr_a.log 2>&1
python scripts/rocpd_summary.py $(ls $O/tr_a/*/*results.db | head -1) $O/trace_overlap.md > /dev/null
grep '^{"metric"' $O/tr_a.log | tail -1 > $O/trace_overlap_bench.json
rocprofv3 --kernel-trace --stats -d $O/tr_b -- python bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-bf16x3-leg --no-overlap > $O/tr_b.log 2>&1
python scripts/rocpd_summary.py $(ls $O/tr_b/*/*results.db | head -1) $O/trace_single.md > /dev/null
grep '^{"metric"' $O/tr_b.log | tail -1 > $O/trace_single_bench.json
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/pmc_f -- python bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-bf16x3-leg --no-overlap > $O/pmc_f.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/pmc_w -- python bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-bf16x3-leg --no-overlap > $O/pmc_w.log 2>&1
python scripts/pmc_bench_summary.py $O/pmc_f $O/pmc_w $O/pmc_bench.md
rm -rf $O/tr_a $O/tr_b $O/pmc_f $O/pmc_w
ls -la $O
