"""Sum FETCH_SIZE / WRITE_SIZE (KB) per kernel family over a rocprofv3 --pmc run of bench.py.
usage: python scripts/pmc_bench_summary.py <fetch_dir> <write_dir> [out.md]
FETCH_SIZE under-counts 16 B/lane streams by 2x on gfx950 (calibrated with a device copy, see
profiles/notes_design_rounds_1_to_3.md, section 6), so reads are reported doubled ("corrected")."""
import csv, glob, re, sys, collections

FAMILIES = [("conv (k_conv_wino4w / k_conv_wino4 / k_conv_wino / k_conv_dma / k_conv / k_conv_h2s / k_conv1_h2s + k_gn_apply + k_splitk_finish[_st])", r"k_conv<|k_conv_dma<|k_conv_wino<|k_conv_wino4<|k_conv_wino4w<|k_conv_bf3<|k_conv_h16<|k_conv_h2s|k_conv1_h2|k_conv1_h16<|k_gn_apply|k_splitk_finish"),
            ("k_conv_h2s only (3x3, fp16x2 products, 8x16-pixel tiles)", r"k_conv_h2s|k_conv_h16<"),
            ("k_conv1_h2s only (1x1, fp16x2 products, 128-pixel tiles)", r"k_conv1_h2"),
            ("k_conv_wino4w only", r"k_conv_wino4w<"),
            ("k_conv_wino4 only", r"k_conv_wino4<"),
            ("k_conv_wino only", r"k_conv_wino<"),
            ("k_conv_dma only", r"k_conv_dma<"),
            ("k_gn_apply", r"k_gn_apply"),
            ("k_splitk_finish", r"k_splitk_finish"),
            ("GroupNorm statistics (k_gn_coef_st; k_gn_partial / k_gn_small / k_gn_coef where no producer statistics exist)", r"k_gn_partial|k_gn_small|k_gn_coef"),
            ("attention", r"k_attention"),
            ("k_march<true, true> (evaluate pass, 2 per view)", r"k_march<true, true"),
            ("k_march<true, false> (fine pass of the reevaluate schedule)", r"k_march<true, false"),
            ("k_march<false, false> (coarse pass of the reevaluate schedule)", r"k_march<false, false"),
            ("k_composite", r"k_composite"),
            ("k_importance", r"k_importance")]


def load(d, counter):
    tot = collections.defaultdict(float); cnt = collections.defaultdict(int)
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] != counter:
                continue
            for fam, rx in FAMILIES:
                if re.search(rx, r["Kernel_Name"]):
                    tot[fam] += float(r["Counter_Value"]); cnt[fam] += 1
    return tot, cnt


def main():
    ft, fc = load(sys.argv[1], "FETCH_SIZE")
    wt, wc = load(sys.argv[2], "WRITE_SIZE")
    out = ["| kernels | launches | FETCH_SIZE sum (KB) | corrected read (GB) | WRITE_SIZE sum (KB) | written (GB) |", "|---|---|---|---|---|---|"]
    for fam, _ in FAMILIES:
        if fc.get(fam, 0) == 0 and wc.get(fam, 0) == 0:
            continue
        out.append(f"| {fam} | {fc.get(fam, 0)} | {ft.get(fam, 0):.0f} | {ft.get(fam, 0) * 2 * 1024 / 1e9:.2f} | {wt.get(fam, 0):.0f} | {wt.get(fam, 0) * 1024 / 1e9:.2f} |")
    text = "\n".join(out)
    if len(sys.argv) > 3:
        open(sys.argv[3], "w").write(text + "\n")
    print(text)


if __name__ == "__main__":
    main()
