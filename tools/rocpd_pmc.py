#!/usr/bin/env python
"""Aggregate rocprofv3 --pmc results (rocpd sqlite) per kernel: sum of each counter over dispatches + calls.
usage: python tools/rocpd_pmc.py <results.db> [--schema]"""
import re
import sqlite3
import sys


def short(name):
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    return re.sub(r"void ", "", name)[:90]


def main():
    db = sqlite3.connect(sys.argv[1])
    cur = db.cursor()
    views = [r[0] for r in cur.execute("select name from sqlite_master where type='view'").fetchall()]
    if "--schema" in sys.argv or "counters_collection" not in views:
        for v in views:
            print(v, [r[1] for r in cur.execute("pragma table_info(%s)" % v).fetchall()])
        if "counters_collection" not in views:
            return
    cols = [r[1] for r in cur.execute("pragma table_info(counters_collection)").fetchall()]
    print("# counters_collection columns:", cols)
    kcol = "kernel_name" if "kernel_name" in cols else ("name" if "name" in cols else cols[0])
    ccol = "counter_name" if "counter_name" in cols else "name"
    vcol = "value" if "value" in cols else "counter_value"
    dcol = "dispatch_id" if "dispatch_id" in cols else None
    q = "select %s, %s, sum(%s), count(distinct %s) from counters_collection group by %s, %s" % (
        kcol, ccol, vcol, dcol or kcol, kcol, ccol)
    agg = {}
    for k, c, v, n in cur.execute(q).fetchall():
        d = agg.setdefault(short(k), {})
        d[c] = v
        d["_calls"] = n
    names = sorted({c for d in agg.values() for c in d if c != "_calls"})
    print("%-92s %6s " % ("kernel", "calls") + " ".join("%22s" % n for n in names))
    key = names[0] if names else None
    for k, d in sorted(agg.items(), key=lambda kv: -(kv[1].get("SQ_WAVE_CYCLES", kv[1].get(key, 0)) or 0)):
        print("%-92s %6d " % (k, d["_calls"]) + " ".join("%22.4g" % d.get(n, float("nan")) for n in names))


if __name__ == "__main__":
    main()
