#!/bin/bash
# Run on the GPU box from the repo root: regenerates the bench line, the kernel traces, the PMC traffic summary, the SQ counter summary of
# the conv kernels and the training / fitting traces under gpurun_out/refresh (copy what should be judged into profiles/, named per round).
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
O=gpurun_out/refresh; rm -rf $O; mkdir -p $O
BFLAGS="--steps 3 --warmup 2 --no-cpu-baseline --no-bf16x3-leg --no-parity --no-train --sustained-steps 0"
python bench.py --steps 20 --warmup 5 > $O/bench_full.log 2>&1; grep '^{"metric"' $O/bench_full.log | tail -1 > $O/bench.json
rocprofv3 --kernel-trace --stats -d $O/tr_a -- python bench.py $BFLAGS > $O/tr_a.log 2>&1
python scripts/rocpd_summary.py $(ls $O/tr_a/*/*results.db | head -1) $O/trace_overlap.md > /dev/null
grep '^{"metric"' $O/tr_a.log | tail -1 > $O/trace_overlap_bench.json
rocprofv3 --kernel-trace --stats -d $O/tr_b -- python bench.py $BFLAGS --no-overlap > $O/tr_b.log 2>&1
python scripts/rocpd_summary.py $(ls $O/tr_b/*/*results.db | head -1) $O/trace_single.md > /dev/null
grep '^{"metric"' $O/tr_b.log | tail -1 > $O/trace_single_bench.json
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/pmc_f -- python bench.py $BFLAGS --no-overlap --no-fit > $O/pmc_f.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/pmc_w -- python bench.py $BFLAGS --no-overlap --no-fit > $O/pmc_w.log 2>&1
python scripts/pmc_bench_summary.py $O/pmc_f $O/pmc_w $O/pmc_bench.md > /dev/null
i=0
while read -r line; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc ${line#pmc: } --output-format csv -d $O/sq$i -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-bf16x3-leg --no-overlap --no-render --no-fit --no-parity --no-train --sustained-steps 0 > $O/sq$i.log 2>&1
done < scripts/pmc_wino.txt
python scripts/pmc_sq_summary.py $O/pmc_sq_conv.md $O/sq1 $O/sq2 $O/sq3 > /dev/null
rocprofv3 --kernel-trace --stats -d $O/tr_t -- python scripts/unet_train_bench.py 2 2 > $O/tr_t.log 2>&1
python scripts/rocpd_summary.py $(ls $O/tr_t/*/*results.db | head -1) $O/trace_unet_train.md > /dev/null
python scripts/unet_train_bench.py 5 2 2>/dev/null | tail -1 > $O/unet_train_wall.txt; python scripts/unet_train_bench.py 3 2 twin 2>/dev/null | tail -1 >> $O/unet_train_wall.txt
rocprofv3 --kernel-trace --stats -d $O/tr_f -- python scripts/train_bench.py 10 > $O/tr_f.log 2>&1
python scripts/rocpd_summary.py $(ls $O/tr_f/*/*results.db | head -1) $O/trace_fit.md > /dev/null
rm -rf $O/tr_a $O/tr_b $O/pmc_f $O/pmc_w $O/sq1 $O/sq2 $O/sq3 $O/tr_t $O/tr_f
ls -la $O
