"""Per-kernel / per-launch-shape SQ counter summary of the conv kernels from rocprofv3 --pmc passes of bench.py (one directory per
pass, counters of scripts/pmc_wino.txt).  usage: python scripts/pmc_sq_summary.py <out.md> <dir> [<dir> ...]
Derived columns (per dispatch, then averaged over the dispatches of a shape):
  clock GHz   = GRBM_GUI_ACTIVE / 8 XCDs / duration          (only in passes that carry GRBM_GUI_ACTIVE)
  mfma busy   = SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x cycles)
  per-wave instruction counts = SQ_INSTS_* / (grid threads / 64)
"""
import collections, csv, glob, re, sys

out, dirs = sys.argv[1], sys.argv[2:]
KEEP = re.compile(r"k_conv_wino4w<|k_conv_wino4<|k_conv_wino<|k_conv_dma<|k_conv_h16<|k_conv_h2s|k_conv1_h2|k_conv1_h16<|k_gn_apply|k_splitk_finish|k_gn_coef_st")
shape = collections.defaultdict(lambda: collections.defaultdict(list))
for d in dirs:
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        disp = collections.defaultdict(dict)
        for r in csv.DictReader(open(f)):
            if not KEEP.search(r["Kernel_Name"]):
                continue
            k = disp[r["Dispatch_Id"]]
            k[r["Counter_Name"]] = float(r["Counter_Value"])
            k["_name"] = re.sub(r"\(.*", "", r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "")).replace("hl::", "")
            k["_waves"] = int(r["Grid_Size"]) / 64.0
            k["_grid"] = int(r["Grid_Size"]) // max(int(r["Workgroup_Size"]), 1)
            k["_us"] = (float(r["End_Timestamp"]) - float(r["Start_Timestamp"])) / 1e3
        for k in disp.values():
            key = (k["_name"], k["_grid"])
            for c, v in k.items():
                if not c.startswith("_"):
                    shape[key][c].append(v)
            shape[key]["_us"].append(k["_us"])
            shape[key]["_waves"].append(k["_waves"])
            if "GRBM_GUI_ACTIVE" in k:
                cyc = k["GRBM_GUI_ACTIVE"] / 8
                shape[key]["clock_ghz"].append(cyc / k["_us"] / 1e3)
                if "SQ_VALU_MFMA_BUSY_CYCLES" in k:
                    shape[key]["mfma_busy"].append(k["SQ_VALU_MFMA_BUSY_CYCLES"] / (cyc * 1024))
mean = lambda v: sum(v) / len(v) if v else float("nan")  # noqa: E731
ratio = lambda a, b: (mean(a) / mean(b)) if a and b and mean(b) else float("nan")  # noqa: E731
cols = ["SQ_WAVES", "SQ_INSTS_MFMA", "SQ_INSTS_VALU", "SQ_INSTS_LDS", "SQ_INSTS_VMEM_RD", "SQ_INSTS_SALU"]
lines = ["| kernel | workgroups | launches | avg us | clock GHz | mfma busy | MFMA/wave | VALU/wave | LDS/wave | VMEM rd/wave | SALU/wave | "
         "wait-inst-any / wave-cycles | LDS bank-conflict / LDS active | TA addr FIFO full / busy |", "|" + "---|" * 14]
for (name, grid), cs in sorted(shape.items(), key=lambda kv: -sum(kv[1]["_us"])):
    tot_us = sum(cs["_us"]) / max(1, len(dirs))
    if tot_us < 500:
        continue
    w = mean(cs["_waves"])
    per = [mean(cs[c]) / w if cs[c] and w else float("nan") for c in cols[1:]]
    wait = ratio(cs["SQ_WAIT_INST_ANY"], cs["SQ_WAVE_CYCLES"])
    bank = ratio(cs["SQ_LDS_BANK_CONFLICT"], cs["SQ_LDS_IDX_ACTIVE"])
    ta = ratio(cs["SQ_VMEM_TA_ADDR_FIFO_FULL"], cs["SQ_BUSY_CYCLES"])
    lines.append(f"| `{name}` | {grid} | {len(cs['_us']) // max(1, len(dirs))} | {mean(cs['_us']):.1f} | {mean(cs['clock_ghz']):.3f} | {mean(cs['mfma_busy']):.3f} | "
                 + " | ".join(f"{v:.0f}" for v in per) + f" | {wait:.3f} | {bank:.3f} | {ta:.3f} |")
open(out, "w").write("\n".join(lines) + "\n")
print("\n".join(lines))
