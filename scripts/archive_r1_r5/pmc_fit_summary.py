"""FETCH_SIZE / WRITE_SIZE (KB) per kernel of the fitting iteration from two rocprofv3 --pmc runs of scripts/train_bench.py.
usage: python scripts/pmc_fit_summary.py <fetch_dir> <write_dir> [out.md]     (FETCH doubled: gfx950 under-count for 16 B/lane streams,
see profiles/notes_design_rounds_1_to_3.md, section 6)"""
import collections, csv, glob, re, sys

KERNELS = [("k_march<.., ACTS> (evaluate + activation matrix)", r"k_march<true, true, 8, false, true>"), ("k_mlp_bwd", r"k_mlp_bwd"), ("k_wgrad", r"k_wgrad"),
           ("k_plane_scatter", r"k_plane_scatter"), ("k_composite_wave", r"k_composite_wave"), ("k_importance", r"k_importance")]


def load(d, counter):
    tot, cnt = collections.defaultdict(float), collections.defaultdict(int)
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] != counter:
                continue
            for name, rx in KERNELS:
                if re.search(rx, r["Kernel_Name"]):
                    tot[name] += float(r["Counter_Value"]); cnt[name] += 1
    return tot, cnt


ft, fc = load(sys.argv[1], "FETCH_SIZE")
wt, wc = load(sys.argv[2], "WRITE_SIZE")
out = ["| kernel | launches | corrected read per launch (MB) | written per launch (MB) |", "|---|---|---|---|"]
for name, _ in KERNELS:
    n = max(fc.get(name, 0), wc.get(name, 0))
    if n:
        out.append(f"| {name} | {n} | {ft.get(name, 0) * 2 * 1024 / 1e6 / max(fc.get(name, 1), 1):.1f} | {wt.get(name, 0) * 1024 / 1e6 / max(wc.get(name, 1), 1):.1f} |")
text = "\n".join(out)
if len(sys.argv) > 3:
    open(sys.argv[3], "w").write(text + "\n")
print(text)
