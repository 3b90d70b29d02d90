#!/usr/bin/env python
"""MFMA-pipe utilisation per kernel INSIDE the training step from two rocprofv3 --pmc passes over bench.py (tools/gpu_session.sh step `pmc`):
   pass A: SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_INSTS_VALU SQ_WAIT_INST_ANY SQ_WAVE_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE
   pass B: GRBM_GUI_ACTIVE SQ_BUSY_CU_CYCLES SQ_INSTS_LDS SQ_INST_LEVEL_VMEM
MFMA busy % = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 XCDs x 1024 SIMDs)   (the formula of profiles/r03_pmc_gemm_*.txt);
GRBM_GUI_ACTIVE is summed over the 8 XCDs by rocprofv3, the SQ counters over all SIMDs.
usage: python tools/pmc_step_summary.py <passA results.db> <passB results.db>"""
import re
import sqlite3
import sys


def table(path):
    cur = sqlite3.connect(path).cursor()
    cols = [r[1] for r in cur.execute("pragma table_info(counters_collection)").fetchall()]
    kcol = "kernel_name" if "kernel_name" in cols else "name"
    ccol = "counter_name" if "counter_name" in cols else "name"
    vcol = "value" if "value" in cols else "counter_value"
    dcol = "dispatch_id" if "dispatch_id" in cols else kcol
    out = {}
    for k, c, v, n in cur.execute("select %s, %s, sum(%s), count(distinct %s) from counters_collection group by %s, %s"
                                  % (kcol, ccol, vcol, dcol, kcol, ccol)).fetchall():
        d = out.setdefault(k, {})
        d[c] = float(v)
        d["_calls"] = int(n)
    return out


def short(name):
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    name = re.sub(r"_ZN12_GLOBAL__N_1\d+", "", name)
    return re.sub(r"void ", "", name)[:74]


def main():
    a, b = table(sys.argv[1]), table(sys.argv[2])
    rows = []
    for k, da in a.items():
        db = b.get(k)
        if not db or "GRBM_GUI_ACTIVE" not in db or "SQ_VALU_MFMA_BUSY_CYCLES" not in da:
            continue
        simd_cycles = db["GRBM_GUI_ACTIVE"] / 8.0 * 1024.0
        rows.append((db["GRBM_GUI_ACTIVE"], short(k), da["_calls"], 100.0 * da["SQ_VALU_MFMA_BUSY_CYCLES"] / simd_cycles,
                     da.get("SQ_INSTS_VALU", 0.0) / max(da.get("SQ_INSTS_MFMA", 0.0), 1.0),
                     100.0 * da.get("SQ_WAIT_INST_ANY", 0.0) / max(da.get("SQ_WAVE_CYCLES", 1.0), 1.0),
                     100.0 * da.get("SQ_LDS_BANK_CONFLICT", 0.0) / max(da.get("SQ_LDS_IDX_ACTIVE", 1.0), 1.0)))
    tot = sum(r[0] for r in rows)
    print("%-74s %6s %8s %10s %10s %9s %9s" % ("kernel (sorted by GPU-active cycles)", "calls", "% cycles", "MFMA busy%", "VALU/MFMA", "wait-inst%", "LDS conf%"))
    share = 0.0
    for r in sorted(rows, reverse=True)[:16]:
        share += 100.0 * r[0] / tot * r[3] / 100.0
        print("%-74s %6d %8.1f %10.1f %10s %9.1f %9.1f" % (r[1], r[2], 100.0 * r[0] / tot, r[3], ("%.1f" % r[4]) if r[3] > 0 else "-", r[5], r[6]))
    print("# MFMA-pipe busy share of ALL GPU-active cycles of the step (sum of %% cycles x MFMA busy%% over the kernels listed): %.1f %%" % share)


if __name__ == "__main__":
    main()
