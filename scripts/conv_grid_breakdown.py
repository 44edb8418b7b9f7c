"""Per-(kernel, grid) totals of the UNet forward from a rocprofv3 kernel trace: which launches the step time sits in.
usage: python scripts/conv_grid_breakdown.py <dir> <n_forwards>"""
import collections, glob, re, sqlite3, sys
db = sqlite3.connect(glob.glob(sys.argv[1] + "/**/*results.db", recursive=True)[0])
nf = float(sys.argv[2])
rows = db.execute("select name, grid_x, grid_y, grid_z, workgroup_x, workgroup_y, workgroup_z, duration from kernels order by start").fetchall()
agg = collections.defaultdict(lambda: [0, 0.0])
for name, gx, gy, gz, wx, wy, wz, d in rows:
    if "hl::" in name:
        short = re.sub(r"\(hl::.*|\(float.*|\(long.*|\(int.*", "", name.replace("(anonymous namespace)::", "").replace("void ", "").replace("hl::", ""))[:40]
        k = (short, gx // max(wx, 1), gy // max(wy, 1), gz // max(wz, 1))
        agg[k][0] += 1
        agg[k][1] += d / 1e3
tot = sum(v[1] for v in agg.values())
print(f"total {tot / nf:.1f} us per forward")
for k, v in sorted(agg.items(), key=lambda kv: -kv[1][1])[:45]:
    print(f"{v[1] / nf:9.1f} us/fwd  n/fwd {v[0] / nf:6.1f}  avg {v[1] / v[0]:8.1f} us  {k}")
