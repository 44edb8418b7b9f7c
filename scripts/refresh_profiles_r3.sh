#!/bin/bash
# Round 3: run on the GPU box from the repo root.  Bench line, kernel traces (two streams / single stream), PMC traffic and SQ counters
# of the UNet step -> gpurun_out/refresh3 (copy what should be judged into profiles/r03_*).
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
O=gpurun_out/refresh3; rm -rf $O; mkdir -p $O
BFLAGS="--steps 3 --warmup 2 --no-cpu-baseline --no-bf16x3-leg --no-parity --no-train --no-fit --no-render --no-e2e --no-batch-sweep --sustained-steps 0"
python bench.py --steps 20 --warmup 5 > $O/bench_full.log 2>&1; grep '^{"metric"' $O/bench_full.log | tail -1 > $O/bench.json
rocprofv3 --kernel-trace --stats -d $O/tr_a -- python bench.py $BFLAGS > $O/tr_a.log 2>&1
python scripts/rocpd_summary.py $(ls $O/tr_a/*/*results.db | head -1) $O/trace_overlap.md > /dev/null
rocprofv3 --kernel-trace --stats -d $O/tr_b -- python bench.py $BFLAGS --no-overlap > $O/tr_b.log 2>&1
python scripts/rocpd_summary.py $(ls $O/tr_b/*/*results.db | head -1) $O/trace_single.md > /dev/null
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/pmc_f -- python bench.py $BFLAGS --no-overlap > $O/pmc_f.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/pmc_w -- python bench.py $BFLAGS --no-overlap > $O/pmc_w.log 2>&1
python scripts/pmc_bench_summary.py $O/pmc_f $O/pmc_w $O/pmc_bench.md > /dev/null
i=0
while read -r line; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc ${line#pmc: } --output-format csv -d $O/sq$i -- python bench.py $BFLAGS --no-overlap > $O/sq$i.log 2>&1
done < scripts/pmc_wino.txt
python scripts/pmc_sq_summary.py $O/pmc_sq_conv.md $O/sq1 $O/sq2 $O/sq3 > /dev/null
rm -rf $O/tr_a $O/tr_b $O/pmc_f $O/pmc_w $O/sq1 $O/sq2 $O/sq3
ls -la $O
scripts/microbench/mfma_fill > $O/microbench_mfma_fill.txt 2>&1
HL_WINO4W=1 rocprofv3 --kernel-trace -d $O/ks1 -- python scripts/wino4_ksweep.py > /dev/null 2>&1; python scripts/rocpd_list.py $O/ks1 k_conv_wino4w > $O/ksweep_wino4w.txt
HL_WINO4W=0 rocprofv3 --kernel-trace -d $O/ks0 -- python scripts/wino4_ksweep.py > /dev/null 2>&1; python scripts/rocpd_list.py $O/ks0 k_conv_wino4 > $O/ksweep_wino4.txt
python scripts/wino4w_probe.py 10 > $O/wino4w_probe.txt 2>&1
python scripts/bf16_probe.py > $O/bf16_probe.txt 2>&1
rm -rf $O/ks0 $O/ks1
ls $O
# late round 3: the 16-bit kernels (probe, SQ counters), the fp16-operand inference mode and the training step in fp32 / bf16
python scripts/h16_probe.py 2>&1 | grep -v amdgpu.ids > $O/h16_probe.txt
bash scripts/pmc_h16.sh 2>&1 | grep "^[abc] waves" > $O/pmc_h16.txt
rocprofv3 --kernel-trace --stats -d $O/f16t -- python scripts/fp16_mode_trace.py > /dev/null 2>&1
python scripts/rocpd_summary.py $(ls $O/f16t/*/*results.db | head -1) $O/trace_fp16_mode.md > /dev/null; rm -rf $O/f16t
for a in fp32 bf16 fp16; do HL_TRAIN_ARITH=$a python scripts/unet_train_bench.py 5 2 2>&1 | grep "UNet training"; done > $O/unet_train_wall.txt
HL_TRAIN_ARITH=fp32 rocprofv3 --kernel-trace --stats -d $O/trf -- python scripts/unet_train_bench.py 2 2 > /dev/null 2>&1
python scripts/rocpd_summary.py $(ls $O/trf/*/*results.db | head -1) $O/trace_unet_train_fp32.md > /dev/null; rm -rf $O/trf
HL_TRAIN_ARITH=bf16 rocprofv3 --kernel-trace --stats -d $O/trb -- python scripts/unet_train_bench.py 2 2 > /dev/null 2>&1
python scripts/rocpd_summary.py $(ls $O/trb/*/*results.db | head -1) $O/trace_unet_train_bf16.md > /dev/null; rm -rf $O/trb
ls $O
# the renderer's stages in fp32 and in the opt-in fp16-operand mode (two views each)
rocprofv3 --kernel-trace --stats -d $O/rs -- python scripts/render_stage_probe.py > /dev/null 2>&1
python scripts/rocpd_summary.py $(ls $O/rs/*/*results.db | head -1) $O/trace_render_fp32_fp16.md > /dev/null; rm -rf $O/rs
python scripts/render_fp16_probe.py 2>&1 | grep -v amdgpu.ids > $O/render_fp16_probe.txt
ls $O
