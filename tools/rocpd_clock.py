#!/usr/bin/env python
"""Effective shader clock per kernel from a rocprofv3 --pmc GRBM_GUI_ACTIVE run (rocpd sqlite): the counter is the busy
cycles of every XCD summed (8 on MI355X), so clock = value / 8 / dispatch duration.
usage: python tools/rocpd_clock.py <results.db> [pattern ...]   (patterns: substrings of kernel names to keep)"""
import re
import sqlite3
import sys


def short(name):
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    return re.sub(r"void ", "", name)[:100]


db = sqlite3.connect(sys.argv[1])
pats = sys.argv[2:]
rows = db.execute("select kernel_name, value, start, end from counters_collection where counter_name = 'GRBM_GUI_ACTIVE'").fetchall()
agg = {}
for name, v, s, e in rows:
    k = short(name)
    if pats and not any(p in k for p in pats):
        continue
    d = agg.setdefault(k, [0, 0.0, 0.0])
    d[0] += 1; d[1] += float(v); d[2] += (e - s)
print("%-102s %6s %10s %10s" % ("kernel", "calls", "avg_us", "clock_GHz"))
for k, (n, cyc, ns) in sorted(agg.items(), key=lambda kv: -kv[1][2]):
    print("%-102s %6d %10.1f %10.3f" % (k, n, ns / n / 1e3, cyc / 8.0 / ns))
