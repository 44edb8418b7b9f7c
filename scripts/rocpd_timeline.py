"""Timeline of a rocprofv3 kernel trace: start / end (ms from the first listed dispatch) of the dispatches whose name contains one of the substrings, last N rows.
usage: python scripts/rocpd_timeline.py <dir> <n_rows> <substr> [<substr> ...]"""
import glob, sqlite3, sys
db = sqlite3.connect(glob.glob(sys.argv[1] + "/**/*results.db", recursive=True)[0])
rows = db.execute("select name, start, end, grid_x, workgroup_x from kernels order by start").fetchall()
rows = [r for r in rows if any(s in r[0] for s in sys.argv[3:])][-int(sys.argv[2]):]
t0 = rows[0][1]
for name, s, e, gx, wx in rows:
    nm = name.replace("(anonymous namespace)::", "")[:46]
    print(f"{(s - t0) / 1e6:9.3f} -> {(e - t0) / 1e6:9.3f} ms  ({(e - s) / 1e6:7.3f})  wg {gx // max(wx, 1):5d}  {nm}")
