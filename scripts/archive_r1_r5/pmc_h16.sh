#!/bin/bash
# SQ counters of k_conv_h16 alone (scripts/h16_probe.py shapes): where do the wave cycles go?
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
O=gpurun_out/pmc_h16; rm -rf $O; mkdir -p $O
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM --output-format csv -d $O/a -- python scripts/h16_probe.py > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE --output-format csv -d $O/b -- python scripts/h16_probe.py > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_CMD_FIFO_FULL SQ_LDS_DATA_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_VMEM_TA_ADDR_FIFO_FULL SQ_INST_CYCLES_VMEM_RD SQ_ACTIVE_INST_MISC --output-format csv -d $O/c -- python scripts/h16_probe.py > /dev/null 2>&1
python - <<'PY'
import csv, glob, collections
for d in ("gpurun_out/pmc_h16/a", "gpurun_out/pmc_h16/b", "gpurun_out/pmc_h16/c"):
    rows = collections.defaultdict(dict)
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if "k_conv_h16" not in r["Kernel_Name"]: continue
            k = rows[r["Dispatch_Id"]]; k[r["Counter_Name"]] = float(r["Counter_Value"]); k["us"] = (float(r["End_Timestamp"]) - float(r["Start_Timestamp"])) / 1e3; k["waves"] = int(r["Grid_Size"]) / 64
    seen = set()
    for did, k in sorted(rows.items(), key=lambda kv: int(kv[0])):
        key = (k["waves"], round(k["us"] / 20))
        if key in seen: continue
        seen.add(key)
        w = k["waves"]
        print(d[-1], f"waves {w:.0f} {k['us']:8.1f} us", "  ".join(f"{c}={v / w:,.0f}" for c, v in k.items() if c not in ("us", "waves")))
PY
rm -rf $O/a $O/b $O/c
