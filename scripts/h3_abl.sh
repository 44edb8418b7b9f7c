#!/bin/bash
# per-kernel durations of the 3x3 fp16x2 kernel, full and with parts removed (H16_ABL variants), against the Winograd kernels
cd /tmp && export TMPDIR=/tmp
R=/root/repo
export HL_SHAPES=${HL_SHAPES:-0,2,4,6}
run() { # name, env...
  name=$1; shift
  rm -rf /tmp/prof_$name
  env "$@" timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/prof_$name -- python $R/scripts/h3_probe.py > /tmp/prof_$name.log 2>&1; tail -3 /tmp/prof_$name.log
  f=$(find /tmp/prof_$name -name "*kernel_trace.csv" | head -1)
  echo "== $name"
  python - "$f" <<'PY'
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
d = collections.OrderedDict()
for r in rows:
    n = r["Kernel_Name"]
    if "pack" in n or "fill" in n.lower() or "elementwise" in n or "distribution" in n: continue
    key = (n[:60], r["Grid_Size_X"] if "Grid_Size_X" in r else r.get("Grid_Size"))
    d.setdefault(key, []).append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
for k, v in d.items():
    v = v[1:] if len(v) > 1 else v
    print(f"  {k[0]:60s} grid {k[1]:>8s}: {sum(v) / len(v):8.1f} us  (n={len(v)})")
PY
}
run wino HL_H2_CONV3_MIN_BLOCKS=-1
run h2
for v in a1 a2 a4 a8 a7 a64 a32; do run h2_$v HL_LIB_PATH=$R/humanliff_amd/exp/lib_$v.so; done
