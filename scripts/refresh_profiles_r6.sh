#!/bin/bash
# Round 6: run on the GPU box from the repo root.  Kernel traces of the headline command (two streams / single stream), the per-forward grid breakdown at B = 4 and
# B = 1, PMC traffic of the UNet step, the renderer's traffic in both schedules -> gpurun_out/refresh6 (copy what should be judged into profiles/r06_*).
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
O=gpurun_out/refresh6; rm -rf $O; mkdir -p $O
BFLAGS="--steps 3 --warmup 2 --no-cpu-baseline --no-bf16x3-leg --no-parity --no-train --no-fit --no-render --no-e2e --no-batch-sweep --sustained-steps 0"
rocprofv3 --kernel-trace --stats -d $O/tr_a -- python bench.py $BFLAGS > $O/tr_a.log 2>&1
python scripts/rocpd_summary.py $(ls $O/tr_a/*/*results.db | head -1) $O/bench_kernel_trace_two_streams.md > /dev/null
rocprofv3 --kernel-trace --stats -d $O/tr_b -- python bench.py $BFLAGS --no-overlap > $O/tr_b.log 2>&1
python scripts/rocpd_summary.py $(ls $O/tr_b/*/*results.db | head -1) $O/bench_kernel_trace_single_stream.md > /dev/null
python scripts/conv_grid_breakdown.py $O/tr_b 6 > $O/grid_breakdown_per_forward.txt 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/pmc_f -- python bench.py $BFLAGS --no-overlap > $O/pmc_f.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/pmc_w -- python bench.py $BFLAGS --no-overlap > $O/pmc_w.log 2>&1
python scripts/pmc_bench_summary.py $O/pmc_f $O/pmc_w $O/pmc_hbm_traffic.md > /dev/null
rm -rf $O/tr_a $O/tr_b $O/pmc_f $O/pmc_w
HL_NO_OVERLAP=1 rocprofv3 --kernel-trace -d $O/b1 -- python scripts/batch1_trace.py > $O/b1.log 2>&1
python scripts/conv_grid_breakdown.py $O/b1 6 > $O/grid_breakdown_b1.txt 2>&1; rm -rf $O/b1
HL_MODES=fp16x2 rocprofv3 --kernel-trace --stats -d $O/rs -- python scripts/render_onepass_check.py > $O/render_onepass_check.txt 2>&1
python scripts/rocpd_summary.py $(ls $O/rs/*/*results.db | head -1) $O/render_kernel_trace.md > /dev/null; rm -rf $O/rs
bash scripts/pmc_render_traffic_r6.sh > /dev/null 2>&1; cp gpurun_out/r06_pmc_render_traffic.md $O/pmc_render_traffic.md
ls -la $O
