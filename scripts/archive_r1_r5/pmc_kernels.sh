#!/bin/bash
# Matrix-pipe busy / shader clock / instruction mix of the ray-march evaluate pass and of the UNet training kernels (one rocprofv3 PMC pass each)
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
O=gpurun_out/pmck; rm -rf $O; mkdir -p $O
C="SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_VALU_TRANS_F32 SQ_WAIT_INST_ANY SQ_WAVE_CYCLES SQ_INSTS_LDS"
rocprofv3 --kernel-trace --pmc $C --output-format csv -d $O/r -- python scripts/render_stages.py > $O/r.log 2>&1
rocprofv3 --kernel-trace --pmc $C --output-format csv -d $O/t -- python scripts/unet_train_bench.py 1 2 > $O/t.log 2>&1
python - <<'PY'
import csv, glob, collections, re
out = ["| run | kernel | workgroups | launches | avg us | clock GHz | matrix pipe busy | MFMA/wave | VALU/wave | transcendental/wave | LDS/wave | wait-inst-any / wave-cycles |", "|" + "---|" * 12]
for tag, d, keep in (("render (scripts/render_stages.py)", "gpurun_out/pmck/r", r"k_march|k_importance|k_composite"),
                     ("UNet training step (scripts/unet_train_bench.py 1 2)", "gpurun_out/pmck/t", r"k_conv_wgrad|k_conv_wino|k_conv_dma|k_gn_bwd")):
    f = glob.glob(d + "/**/*counter_collection.csv", recursive=True)[0]
    disp = collections.defaultdict(dict)
    for r in csv.DictReader(open(f)):
        if not re.search(keep, r["Kernel_Name"]): continue
        k = disp[r["Dispatch_Id"]]; k[r["Counter_Name"]] = float(r["Counter_Value"])
        k["name"] = re.sub(r"\(.*", "", r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "")).replace("hl::", "")
        k["wg"] = int(r["Grid_Size"]) // max(int(r["Workgroup_Size"]), 1); k["waves"] = int(r["Grid_Size"]) / 64.0
        k["us"] = (float(r["End_Timestamp"]) - float(r["Start_Timestamp"])) / 1e3
    shapes = collections.defaultdict(list)
    for k in disp.values(): shapes[(k["name"], k["wg"])].append(k)
    for (name, wg), ks in sorted(shapes.items(), key=lambda kv: -sum(k["us"] for k in kv[1]))[:8]:
        m = lambda c: sum(k.get(c, 0.0) for k in ks) / len(ks)
        cyc = m("GRBM_GUI_ACTIVE") / 8; us = m("us"); w = m("waves")
        out.append(f"| {tag} | `{name}` | {wg} | {len(ks)} | {us:.1f} | {cyc / us / 1e3:.3f} | {m('SQ_VALU_MFMA_BUSY_CYCLES') / (cyc * 1024):.3f} | {m('SQ_INSTS_MFMA') / w:.0f} | "
                   f"{m('SQ_INSTS_VALU') / w:.0f} | {m('SQ_INSTS_VALU_TRANS_F32') / w:.0f} | {m('SQ_INSTS_LDS') / w:.0f} | {m('SQ_WAIT_INST_ANY') / max(m('SQ_WAVE_CYCLES'), 1):.3f} |")
open("gpurun_out/pmck/summary.md", "w").write("\n".join(out) + "\n")
print("\n".join(out))
PY
rm -rf $O/r $O/t
