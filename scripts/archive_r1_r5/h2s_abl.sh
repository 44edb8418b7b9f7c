#!/bin/bash
# per-kernel durations of k_conv_h2s (single-op probe scripts/h3_probe.py, raw fp32 input), full and with parts removed (H2S_ABL variants in humanliff_amd/exp)
cd /tmp && export TMPDIR=/tmp
R=/root/repo
export HL_SHAPES=${HL_SHAPES:-0,1,2,4}
run() {
  name=$1; shift
  rm -rf /tmp/prof_$name
  env "$@" timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/prof_$name -- python $R/scripts/h3_probe.py > /tmp/prof_$name.log 2>&1
  f=$(find /tmp/prof_$name -name "*kernel_trace.csv" | head -1)
  echo "== $name"
  python - "$f" <<'PY'
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
d = collections.OrderedDict()
for r in rows:
    n = r["Kernel_Name"]
    if "k_conv_h2s" not in n and "wino4" not in n: continue
    d.setdefault((n[:46], r["Grid_Size_X"]), []).append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
for k, v in d.items():
    v = v[1:] if len(v) > 1 else v
    print(f"  {k[0]:46s} grid {k[1]:>8s}: {sum(v) / len(v):8.1f} us  (n={len(v)})")
PY
}
run full
for v in $VARIANTS; do run $v HL_LIB_PATH=$R/humanliff_amd/exp/lib_$v.so; done
run wino HL_H2_CONV3_MIN_BLOCKS=-1
