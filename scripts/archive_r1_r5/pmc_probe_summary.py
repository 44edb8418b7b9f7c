"""All counters of the passes of scripts/pmc_probe.sh per (kernel, grid): per wave for instruction counts, as a fraction of the wave
cycles for the SQ wait / active cycle counters, plus clock and matrix-pipe busy.  usage: python scripts/pmc_probe_summary.py <dir>..."""
import collections, csv, glob, re, sys
shape = collections.defaultdict(lambda: collections.defaultdict(list))
for d in sys.argv[1:]:
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        disp = collections.defaultdict(dict)
        for r in csv.DictReader(open(f)):
            if "k_conv" not in r["Kernel_Name"]:
                continue
            k = disp[r["Dispatch_Id"]]
            k[r["Counter_Name"]] = float(r["Counter_Value"])
            k["_name"] = re.sub(r"\(hl::.*", "", r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "")).replace("hl::", "")
            k["_waves"] = int(r["Grid_Size"]) / 64.0
            k["_grid"] = int(r["Grid_Size"]) // max(int(r["Workgroup_Size"]), 1)
            k["_us"] = (float(r["End_Timestamp"]) - float(r["Start_Timestamp"])) / 1e3
        for k in disp.values():
            key = (k["_name"], k["_grid"])
            for c, v in k.items():
                if not c.startswith("_name"):
                    shape[key][c].append(v)
mean = lambda v: sum(v) / len(v) if v else float("nan")  # noqa: E731
for (name, grid), cs in sorted(shape.items()):
    w, us = mean(cs["_waves"]), mean(cs["_us"])
    cyc = mean(cs["GRBM_GUI_ACTIVE"]) / 8 if cs["GRBM_GUI_ACTIVE"] else float("nan")
    wc = mean(cs["SQ_WAVE_CYCLES"]) if cs["SQ_WAVE_CYCLES"] else float("nan")
    print(f"== {name} grid {grid}: {us:.1f} us, clock {cyc / us / 1e3:.3f} GHz, mfma busy {mean(cs['SQ_VALU_MFMA_BUSY_CYCLES']) / (cyc * 1024):.3f}")
    for c in sorted(cs):
        if c.startswith("_") or c == "GRBM_GUI_ACTIVE":
            continue
        v = mean(cs[c])
        if c.startswith("SQ_INSTS"):
            print(f"   {c:32s} {v / w:10.1f} per wave")
        else:
            print(f"   {c:32s} {v / wc:10.4f} of wave cycles   ({v / (cyc * 256):.4f} per CU cycle)")
