"""Summarise a rocprofv3 (rocpd sqlite) kernel trace: per-kernel totals (the --stats view) and, for the
hot kernels, per-launch-shape statistics.  Usage: python scripts/rocpd_summary.py <results.db> [out.md]"""
import re
import sqlite3
import sys


def short(name):
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    name = re.sub(r"^void ", "", name)
    if len(name) > 90:
        name = name[:87] + "..."
    return name


def main():
    db = sqlite3.connect(sys.argv[1])
    cur = db.cursor()
    rows = cur.execute("select name, grid_x, grid_y, grid_z, workgroup_x, duration, vgpr_count, accum_vgpr_count, lds_size, "
                       "scratch_size from kernels").fetchall()
    out = []
    tot = sum(r[5] for r in rows)
    by = {}
    for r in rows:
        by.setdefault(r[0], []).append(r)
    out.append("| kernel | calls | total ms | avg us | min us | max us | % | VGPR | AGPR | LDS | scratch |")
    out.append("|---|---|---|---|---|---|---|---|---|---|---|")
    for name, rs in sorted(by.items(), key=lambda kv: -sum(r[5] for r in kv[1])):
        d = [r[5] for r in rs]
        out.append(f"| `{short(name)}` | {len(d)} | {sum(d)/1e6:.3f} | {sum(d)/len(d)/1e3:.1f} | {min(d)/1e3:.1f} | "
                   f"{max(d)/1e3:.1f} | {100*sum(d)/tot:.2f} | {rs[0][6]} | {rs[0][7]} | {rs[0][8]} | {rs[0][9]} |")
    out.append("")
    out.append("Per launch shape of the kernels above 2% (grid in workgroups):")
    out.append("")
    out.append("| kernel | grid (wg) | calls | avg us | total ms |")
    out.append("|---|---|---|---|---|")
    for name, rs in sorted(by.items(), key=lambda kv: -sum(r[5] for r in kv[1])):
        if sum(r[5] for r in rs) < 0.02 * tot:
            continue
        shapes = {}
        for r in rs:
            key = (r[1] // max(r[4], 1), r[2], r[3])
            shapes.setdefault(key, []).append(r[5])
        for key, d in sorted(shapes.items(), key=lambda kv: -sum(kv[1])):
            out.append(f"| `{short(name)[:50]}` | {key[0]}x{key[1]}x{key[2]} | {len(d)} | {sum(d)/len(d)/1e3:.1f} | {sum(d)/1e6:.3f} |")
    text = "\n".join(out)
    if len(sys.argv) > 2:
        open(sys.argv[2], "w").write(text + "\n")
    print(text)


if __name__ == "__main__":
    main()
