"""Timeline of the LAST UNet forward in a rocprofv3 kernel trace (rocpd sqlite): every launch with its start offset, duration, queue and the
idle time since the previous kernel ended on ANY queue - where the wall time of a small-batch forward goes.
usage: python scripts/timeline.py <dir> [out.txt]"""
import glob, re, sqlite3, sys
db = sqlite3.connect(glob.glob(sys.argv[1] + "/**/*results.db", recursive=True)[0])
cols = [r[1] for r in db.execute("pragma table_info(kernels)").fetchall()]
qcol = "queue_id" if "queue_id" in cols else ("stream_id" if "stream_id" in cols else None)
rows = db.execute(f"select name, grid_x, grid_y, grid_z, workgroup_x, workgroup_y, workgroup_z, start, end, {qcol or '0'} from kernels order by start").fetchall()


def short(name):
    return re.sub(r"\(hl::.*|\(float.*|\(long.*|\(int.*|\(void.*|\(unsigned.*", "", name.replace("(anonymous namespace)::", "").replace("void ", "").replace("hl::", ""))[:44]


# forwards start at k_timestep_embedding
starts = [i for i, r in enumerate(rows) if "k_timestep_embedding" in r[0]]
a = starts[-2] if len(starts) > 1 else starts[-1]
b = starts[-1] if len(starts) > 1 else len(rows)
seg = rows[a:b]
t0 = seg[0][7]
out = [f"columns: {cols}", f"forward of {len(seg)} launches, wall {(max(r[8] for r in seg) - t0) / 1e3:.1f} us"]
busy_end = t0
idle = 0.0
union = 0.0
for r in seg:
    name, gx, gy, gz, wx, wy, wz, s, e, q = r
    gap = max(0.0, (s - busy_end) / 1e3)
    idle += gap
    union += max(0, e - max(s, busy_end)) / 1e3
    busy_end = max(busy_end, e)
    out.append(f"{(s - t0) / 1e3:9.1f} +{(e - s) / 1e3:7.1f}  idle {gap:5.1f}  q{q}  {short(name)} ({gx // max(wx, 1)},{gy // max(wy, 1)},{gz // max(wz, 1)})")
out.insert(2, f"time with no kernel running: {idle:.1f} us; union of kernel time {union:.1f} us; sum of durations {sum((r[8] - r[7]) for r in seg) / 1e3:.1f} us")
text = "\n".join(out)
if len(sys.argv) > 2:
    open(sys.argv[2], "w").write(text + "\n")
print("\n".join(out[:3]))
