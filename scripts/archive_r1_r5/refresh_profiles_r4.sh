#!/bin/bash
# Round 4: run on the GPU box from the repo root.  Kernel traces (two streams / single stream), PMC traffic and SQ counters of the UNet step,
# per-forward grid breakdown at B = 4 and B = 1, the renderer's counters in both product modes, the training step -> gpurun_out/refresh4
# (copy what should be judged into profiles/r04_*: scripts/copy_profiles.py).
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
O=gpurun_out/refresh4; rm -rf $O; mkdir -p $O
BFLAGS="--steps 3 --warmup 2 --no-cpu-baseline --no-bf16x3-leg --no-parity --no-train --no-fit --no-render --no-e2e --no-batch-sweep --sustained-steps 0"
rocprofv3 --kernel-trace --stats -d $O/tr_a -- python bench.py $BFLAGS > $O/tr_a.log 2>&1
python scripts/rocpd_summary.py $(ls $O/tr_a/*/*results.db | head -1) $O/trace_overlap.md > /dev/null
rocprofv3 --kernel-trace --stats -d $O/tr_b -- python bench.py $BFLAGS --no-overlap > $O/tr_b.log 2>&1
python scripts/rocpd_summary.py $(ls $O/tr_b/*/*results.db | head -1) $O/trace_single.md > /dev/null
python scripts/conv_grid_breakdown.py $O/tr_b 6 > $O/grid_breakdown_per_forward.txt 2>&1
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
# batch 1 (the shipped script's batch size)
HL_NO_OVERLAP=1 rocprofv3 --kernel-trace -d $O/b1 -- python scripts/batch1_trace.py > $O/b1.log 2>&1
python scripts/conv_grid_breakdown.py $O/b1 6 > $O/grid_breakdown_b1.txt 2>&1; rm -rf $O/b1
# the renderer: stages of a view in both product modes (kernel trace) and the counters of the evaluate pass
rocprofv3 --kernel-trace --stats -d $O/rs -- python scripts/render_b3_probe.py > $O/render_b3_probe.txt 2>&1
python scripts/rocpd_summary.py $(ls $O/rs/*/*results.db | head -1) $O/trace_render_bf16x3_fp32.md > /dev/null; rm -rf $O/rs
bash scripts/pmc_b3.sh > /dev/null 2>&1; cp gpurun_out/pmc_b3/summary.md $O/pmc_render_b3_vs_fp32.md
for f in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --kernel-trace --pmc $f --output-format csv -d $O/rp_$f -- python scripts/render_b3_abl.py bf16x3 > /dev/null 2>&1
done
python - <<'PY' > $O/pmc_render_traffic.md
import csv, glob, collections
out = collections.defaultdict(dict)
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    f = glob.glob(f"gpurun_out/refresh4/rp_{c}/**/*counter_collection.csv", recursive=True)
    if not f: continue
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(f[0])):
        if r["Counter_Name"] == c and "k_" in r["Kernel_Name"]:
            agg[r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0][:60]].append(float(r["Counter_Value"]))
    for k, v in agg.items(): out[k][c] = (len(v), sum(v) / len(v))
print("| kernel | launches | FETCH_SIZE per launch (KB) | read per launch, doubled per the gfx950 calibration (MB) | WRITE_SIZE per launch (KB) | written per launch (MB) |\n|---|---|---|---|---|---|")
for k, d in sorted(out.items(), key=lambda kv: -kv[1].get("FETCH_SIZE", (0, 0))[1]):
    fn, fv = d.get("FETCH_SIZE", (0, 0.0)); wn, wv = d.get("WRITE_SIZE", (0, 0.0))
    print(f"| `{k}` | {fn} | {fv:.0f} | {fv * 2 * 1024 / 1e6:.1f} | {wv:.0f} | {wv * 1024 / 1e6:.1f} |")
PY
rm -rf $O/rp_FETCH_SIZE $O/rp_WRITE_SIZE
# the UNet training step (attention backward now a HIP kernel)
for a in fp32 bf16; do HL_TRAIN_ARITH=$a python scripts/unet_train_bench.py 5 2 2>&1 | grep "UNet training"; done > $O/unet_train_wall.txt
HL_TRAIN_ARITH=fp32 rocprofv3 --kernel-trace --stats -d $O/trf -- python scripts/unet_train_bench.py 2 2 > /dev/null 2>&1
python scripts/rocpd_summary.py $(ls $O/trf/*/*results.db | head -1) $O/trace_unet_train_fp32.md > /dev/null; rm -rf $O/trf
ls -la $O
