#!/bin/bash
# per-kernel totals of the production forward (HL_B, default 4) under rocprofv3 for a dispatch configuration given in the environment
cd /tmp && export TMPDIR=/tmp
R=/root/repo
export HL_B=${HL_B:-4}
name=${1:-run}
rm -rf /tmp/prof_$name
timeout 600 rocprofv3 --kernel-trace --output-format csv -d /tmp/prof_$name -- python $R/scripts/fwd_time.py > /tmp/prof_$name.log 2>&1; grep "B=" /tmp/prof_$name.log
f=$(find /tmp/prof_$name -name "*kernel_trace.csv" | head -1)
python - "$f" <<'PY'
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
d = collections.OrderedDict()
for r in rows:
    n = r["Kernel_Name"].replace("void hl::(anonymous namespace)::", "").replace("hl::(anonymous namespace)::", "")
    d.setdefault((n[:44], r["Grid_Size_X"]), []).append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
tot = 0
for k, v in sorted(d.items(), key=lambda kv: -sum(kv[1])):
    per_fwd = sum(v) / 13
    tot += per_fwd
    if per_fwd > 150: print(f"  {k[0]:44s} grid {k[1]:>8s}: {sum(v) / len(v):8.1f} us x {len(v) / 13:5.1f} = {per_fwd:8.1f} us / forward")
print(f"  total {tot:.0f} us / forward")
PY
