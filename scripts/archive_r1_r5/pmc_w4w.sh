#!/bin/bash
# SQ counters of k_conv_wino4w alone (K sweep at 256x256, batch 4): where do the wave cycles go?
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
O=gpurun_out/pmc_w4w; rm -rf $O; mkdir -p $O
HL_WINO4W=1 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM --output-format csv -d $O/a -- python scripts/wino4_ksweep.py > /dev/null 2>&1
HL_WINO4W=1 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE --output-format csv -d $O/b -- python scripts/wino4_ksweep.py > /dev/null 2>&1
python - <<'PY'
import csv, glob, collections
for d in ("gpurun_out/pmc_w4w/a", "gpurun_out/pmc_w4w/b"):
    rows = collections.defaultdict(dict)
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if "k_conv_wino4w" not in r["Kernel_Name"]: continue
            k = rows[r["Dispatch_Id"]]; k[r["Counter_Name"]] = float(r["Counter_Value"]); k["us"] = (float(r["End_Timestamp"]) - float(r["Start_Timestamp"])) / 1e3; k["waves"] = int(r["Grid_Size"]) / 64
    for i, (did, k) in enumerate(sorted(rows.items(), key=lambda kv: int(kv[0]))):
        if i % 4 != 3: continue
        w = k["waves"]
        print(d[-1], f"{k['us']:8.1f} us", "  ".join(f"{c}={v / w:,.0f}/wave" for c, v in k.items() if c not in ("us", "waves")))
PY
rm -rf $O/a $O/b
