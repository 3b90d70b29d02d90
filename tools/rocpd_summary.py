#!/usr/bin/env python
"""Summarise a rocprofv3 rocpd sqlite database (kernel-trace) into per-kernel stats (text/CSV).
usage: python tools/rocpd_summary.py <results.db> [--by-grid] [--skip-first N]"""
import re
import sqlite3
import sys


def short(name):
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    name = re.sub(r"void ", "", name)
    return name[:110]


def main():
    path = sys.argv[1]
    by_grid = "--by-grid" in sys.argv
    db = sqlite3.connect(path)
    cur = db.cursor()
    cols = [r[1] for r in cur.execute("pragma table_info(kernels)").fetchall()]
    name_col = "name" if "name" in cols else "kernel_name"
    q = "select %s, start, end, grid_x, grid_y, workgroup_x from kernels order by start" % name_col
    rows = cur.execute(q).fetchall()
    agg = {}
    t_first, t_last = rows[0][1], rows[-1][2]
    for name, s, e, gx, gy, wx in rows:
        key = short(name) + ((" grid=%dx%d" % (gx // max(wx, 1), gy)) if by_grid else "")
        d = agg.setdefault(key, [0, 0.0, 1e30, 0.0])
        dur = (e - s) / 1e3
        d[0] += 1; d[1] += dur; d[2] = min(d[2], dur); d[3] = max(d[3], dur)
    tot = sum(v[1] for v in agg.values())
    print("# kernels: %d dispatches, %.3f ms busy, %.3f ms wall span" % (len(rows), tot / 1e3, (t_last - t_first) / 1e6))
    print("%-118s %8s %12s %10s %10s %10s %6s" % ("kernel", "calls", "total_us", "avg_us", "min_us", "max_us", "pct"))
    for k, v in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        print("%-118s %8d %12.1f %10.2f %10.2f %10.2f %6.2f" % (k, v[0], v[1], v[1] / v[0], v[2], v[3], 100 * v[1] / tot))


if __name__ == "__main__":
    main()
