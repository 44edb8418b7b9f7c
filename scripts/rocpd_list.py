"""List the dispatches of a kernel (substring) from a rocprofv3 kernel trace in launch order: duration in us, grid in workgroups.
usage: python scripts/rocpd_list.py <dir> <kernel-substring>"""
import glob, sqlite3, sys
db = sqlite3.connect(glob.glob(sys.argv[1] + "/**/*results.db", recursive=True)[0])
rows = db.execute("select name, grid_x, workgroup_x, duration, start from kernels order by start").fetchall()
for name, gx, wx, d, _ in rows:
    if sys.argv[2] in name:
        print(f"{d / 1e3:9.1f} us  grid {gx // max(wx, 1):6d}  {name[:60]}")
