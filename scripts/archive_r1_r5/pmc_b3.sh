#!/bin/bash
# Shader clock, matrix-pipe occupancy and instruction mix of the renderer's evaluate pass in the three product modes (one rocprofv3 PMC pass each
# over scripts/render_b3_abl.py: five 512x512 views at 128 + 128 samples).  Writes gpurun_out/pmc_b3/summary.md
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
O=gpurun_out/pmc_b3; rm -rf $O; mkdir -p $O
C1="SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_VALU_TRANS_F32 SQ_WAIT_INST_ANY SQ_WAVE_CYCLES SQ_INSTS_LDS"
C2="SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_VMEM GRBM_GUI_ACTIVE"
for mode in fp16x2 bf16x3 fp32; do
  rocprofv3 --kernel-trace --pmc $C1 --output-format csv -d $O/${mode}_1 -- python scripts/render_b3_abl.py $mode > $O/${mode}_1.log 2>&1
  rocprofv3 --kernel-trace --pmc $C2 --output-format csv -d $O/${mode}_2 -- python scripts/render_b3_abl.py $mode > $O/${mode}_2.log 2>&1
done
python - <<'PY'
import csv, glob, collections, re
rows = ["| mode | kernel | workgroups | launches | avg us | clock GHz | matrix pipe busy | MFMA/wave | VALU/wave | transcendental/wave | LDS/wave | wait-inst-any / wave-cycles | wait-any / wave-cycles | active-inst-any / wave-cycles | VALU-active / wave-cycles |", "|" + "---|" * 15]
for mode in ("fp32", "bf16x3", "fp16x2"):
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for p in (1, 2):
        fs = glob.glob(f"gpurun_out/pmc_b3/{mode}_{p}/**/*counter_collection.csv", recursive=True)
        if not fs: continue
        disp = collections.defaultdict(dict)
        for r in csv.DictReader(open(fs[0])):
            if not re.search(r"k_march", r["Kernel_Name"]): continue
            k = disp[r["Dispatch_Id"]]; k[r["Counter_Name"]] = float(r["Counter_Value"])
            k["name"] = re.sub(r"\(.*", "", r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", ""))
            k["wg"] = int(r["Grid_Size"]) // max(int(r["Workgroup_Size"]), 1); k["waves"] = int(r["Grid_Size"]) / 64.0
            k["us"] = (float(r["End_Timestamp"]) - float(r["Start_Timestamp"])) / 1e3
        for k in disp.values():
            for c, v in k.items():
                if c != "name": agg[(k["name"], k["wg"])][c].append(v)
    for (name, wg), cs in agg.items():
        m = lambda c: (sum(cs[c]) / len(cs[c])) if cs[c] else float("nan")
        cyc = m("GRBM_GUI_ACTIVE") / 8; us = m("us"); w = m("waves")
        rows.append(f"| {mode} | `{name}` | {wg} | {len(cs['us']) // 2} | {us:.1f} | {cyc / us / 1e3:.3f} | {m('SQ_VALU_MFMA_BUSY_CYCLES') / (cyc * 1024):.3f} | {m('SQ_INSTS_MFMA') / w:.0f} | "
                    f"{m('SQ_INSTS_VALU') / w:.0f} | {m('SQ_INSTS_VALU_TRANS_F32') / w:.0f} | {m('SQ_INSTS_LDS') / w:.0f} | {m('SQ_WAIT_INST_ANY') / m('SQ_WAVE_CYCLES'):.3f} | "
                    f"{m('SQ_WAIT_ANY') / m('SQ_WAVE_CYCLES'):.3f} | {m('SQ_ACTIVE_INST_ANY') / m('SQ_WAVE_CYCLES'):.3f} | {m('SQ_ACTIVE_INST_VALU') / m('SQ_WAVE_CYCLES'):.3f} |")
open("gpurun_out/pmc_b3/summary.md", "w").write("\n".join(rows) + "\n")
print("\n".join(rows))
PY
rm -rf $O/*_1 $O/*_2
