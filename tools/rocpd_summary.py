"""Summarise a rocprofv3 (ROCm 7.x rocpd SQLite) kernel trace: per-kernel and per-(kernel, grid) totals -> CSV + markdown.
usage: python tools/rocpd_summary.py <results.db> <out_prefix> [steps]"""
import re
import sqlite3
import sys


def short(name: str) -> str:
    name = re.sub(r"\(.*$", "", name)
    name = name.replace("void ", "").replace("(anonymous namespace)::", "")
    return name[:110]


def main():
    db, out = sys.argv[1], sys.argv[2]
    c = sqlite3.connect(db)
    cols = [r[1] for r in c.execute("pragma table_info(kernels)")]
    rows = c.execute("select name, start, end, grid_x, grid_y, grid_z, workgroup_x from kernels").fetchall() if "grid_x" in cols else \
        c.execute("select name, start, end, grid_size_x, grid_size_y, grid_size_z, workgroup_size_x from kernels").fetchall()
    per, per_shape = {}, {}
    tot = 0.0
    for name, s, e, gx, gy, gz, wx in rows:
        d = (e - s) / 1e3
        tot += d
        k = short(name)
        a = per.setdefault(k, [0, 0.0, 1e30, 0.0]); a[0] += 1; a[1] += d; a[2] = min(a[2], d); a[3] = max(a[3], d)
        ks = (k, gx // max(wx, 1), gy, gz)
        b = per_shape.setdefault(ks, [0, 0.0]); b[0] += 1; b[1] += d
    with open(out + "_kernels.csv", "w") as f:
        f.write("kernel,calls,total_us,avg_us,min_us,max_us,pct\n")
        for k, a in sorted(per.items(), key=lambda kv: -kv[1][1]):
            f.write(f"\"{k}\",{a[0]},{a[1]:.1f},{a[1] / a[0]:.2f},{a[2]:.2f},{a[3]:.2f},{100 * a[1] / tot:.2f}\n")
    with open(out + "_kernels_by_grid.csv", "w") as f:
        f.write("kernel,blocks_x,grid_y,grid_z,calls,total_us,avg_us,pct\n")
        for (k, bx, gy, gz), b in sorted(per_shape.items(), key=lambda kv: -kv[1][1])[:80]:
            f.write(f"\"{k}\",{bx},{gy},{gz},{b[0]},{b[1]:.1f},{b[1] / b[0]:.2f},{100 * b[1] / tot:.2f}\n")
    print(f"total kernel time {tot / 1e3:.2f} ms over {len(rows)} dispatches")
    for k, a in sorted(per.items(), key=lambda kv: -kv[1][1])[:14]:
        print(f"{100 * a[1] / tot:6.2f}%  {a[1] / 1e3:9.2f} ms  {a[0]:5d} calls  avg {a[1] / a[0]:9.1f} us  {k}")


if __name__ == "__main__":
    main()
