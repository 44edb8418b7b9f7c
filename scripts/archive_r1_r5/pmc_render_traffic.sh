#!/bin/bash
# HBM traffic per launch of the renderer's kernels (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes, five 512x512 views in the default
# bf16x3 mode).  Writes gpurun_out/pmc_render_traffic.md
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
O=gpurun_out/prt; rm -rf $O; mkdir -p $O
for f in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --kernel-trace --pmc $f --output-format csv -d $O/rp_$f -- python scripts/archive_r1_r5/render_b3_abl.py bf16x3 > /dev/null 2>&1
done
python - <<'PY' > gpurun_out/pmc_render_traffic.md
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
cat gpurun_out/pmc_render_traffic.md
