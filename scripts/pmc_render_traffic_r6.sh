#!/bin/bash
# HBM traffic per launch of the renderer's kernels, both schedules (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in SEPARATE passes with --kernel-trace only; 512x512 views
# in the default fp16x2 mode through scripts/render_onepass_check.py, which renders with the four-launch and with the two-launch one-pass schedule).
# FETCH_SIZE is doubled per the gfx950 calibration of MI355X_MICROARCH.md.  Writes gpurun_out/r06_pmc_render_traffic.md
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
O=gpurun_out/prt; rm -rf $O; mkdir -p $O
for f in FETCH_SIZE WRITE_SIZE; do
  HL_MODES=fp16x2 rocprofv3 --kernel-trace --pmc $f --output-format csv -d $O/rp_$f -- python scripts/render_onepass_check.py > $O/log_$f.txt 2>&1
done
python - <<'PY' > gpurun_out/r06_pmc_render_traffic.md
import csv, glob, collections
out = collections.defaultdict(dict)
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    f = glob.glob(f"gpurun_out/prt/rp_{c}/**/*counter_collection.csv", recursive=True)
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
rm -rf $O
cat gpurun_out/r06_pmc_render_traffic.md
