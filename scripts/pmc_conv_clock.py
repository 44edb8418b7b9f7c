"""Per-dispatch shader clock and matrix-pipe utilisation of the conv kernels from a rocprofv3 run with
`--pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv` (e.g. of scripts/conv_sweep.py).
GRBM_GUI_ACTIVE is summed over the 8 XCDs; SQ_VALU_MFMA_BUSY_CYCLES over the 1024 SIMDs.
usage: python scripts/pmc_conv_clock.py <dir> [min_us]"""
import csv, collections, sys, glob
f = glob.glob(sys.argv[1] + '/**/*counter_collection.csv', recursive=True)[0]
min_us = float(sys.argv[2]) if len(sys.argv) > 2 else 1000.0
by = collections.defaultdict(dict)
for r in csv.DictReader(open(f)):
    if any(k in r['Kernel_Name'] for k in ('k_conv_dma', 'k_conv_bf3', 'k_conv_wino')):
        d = by[r['Dispatch_Id']]
        d[r['Counter_Name']] = float(r['Counter_Value'])
        d['t0'] = float(r['Start_Timestamp']); d['t1'] = float(r['End_Timestamp'])
        d['g'] = r['Grid_Size']; d['k'] = r['Kernel_Name'].split('(hl::')[0].split('::')[-1]
print("| dispatch | kernel | grid (threads) | duration us | shader clock GHz | matrix pipe busy |")
print("|---|---|---|---|---|---|")
for k, v in by.items():
    dur = (v['t1'] - v['t0']) / 1e3
    if dur < min_us:
        continue
    cyc = v['GRBM_GUI_ACTIVE'] / 8
    print(f"| {k} | `{v['k']}` | {v['g']} | {dur:.1f} | {cyc / dur / 1e3:.3f} | {v['SQ_VALU_MFMA_BUSY_CYCLES'] / (cyc * 1024):.3f} |")
