"""Mean counter values per (kernel, grid) from a rocprofv3 --pmc ... --output-format csv run.  usage: python scripts/pmc_list_kernels.py <dir> [substring]"""
import collections, csv, glob, re, sys
sub = sys.argv[2] if len(sys.argv) > 2 else "k_conv"
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if sub not in r["Kernel_Name"]:
            continue
        name = re.sub(r"\(hl::.*", "", r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "")).replace("hl::", "")
        key = (name, int(r["Grid_Size"]) // max(int(r["Workgroup_Size"]), 1))
        agg[key][r["Counter_Name"]].append(float(r["Counter_Value"]))
        agg[key]["_us"].append((float(r["End_Timestamp"]) - float(r["Start_Timestamp"])) / 1e3)
for key, cs in sorted(agg.items()):
    print(key, {c: round(sum(v) / len(v), 1) for c, v in cs.items()})
