#!/usr/bin/env python
"""Reduce the two rocpd_pmc.py tables written by tools/gpu_pmc_traffic.sh to the per-launch HBM traffic of the bf16->bf16
NT GEMM kernels (each family with all its epilogue instantiations together; bench.py's dominant kernel is one of them).
FETCH_SIZE / WRITE_SIZE are reported in KB; on gfx950 FETCH_SIZE tallies the 128-byte requests of wide coalesced reads
at 64 B (MI355X_MICROARCH.md, HBM section) -> doubled; WRITE_SIZE is taken as reported (uncalibrated in that guide).
usage: python tools/pmc_traffic.py <fetch table> <write table> <batch>"""
import json
import re
import sys

FAMILIES = {"gemm_nt_dual_kernel<bf16->bf16>": ("gemm_nt_dual_kernel", "IDF16bL"),       # TO = bf16 (mangled DF16b)
            "gemm_nt_8ph_kernel<bf16->bf16>": ("gemm_nt_8ph_kernel", "DF16bDF16b")}


def table(path, name, tag):
    total, calls = 0.0, 0
    for line in open(path):
        if name not in line or tag not in line:
            continue
        m = re.search(r"\s(\d+)\s+([0-9.eE+\-]+)\s*$", line.rstrip())
        if not m:
            continue
        calls += int(m.group(1))
        total += float(m.group(2))
    return total, calls


out = {"per_gpu_batch": int(sys.argv[3]),
       "fetch_correction": "FETCH_SIZE x 1024 x 2 (gfx950 counts 128-B requests of wide coalesced reads at 64 B)",
       "source": "rocprofv3 --kernel-trace --pmc FETCH_SIZE / WRITE_SIZE, separate passes (tools/gpu_pmc_traffic.sh)",
       "kernels": {}}
for fam, (name, tag) in FAMILIES.items():
    fetch_kb, calls_f = table(sys.argv[1], name, tag)
    write_kb, calls_w = table(sys.argv[2], name, tag)
    if not calls_f or not calls_w:
        continue
    fetch = 2.0 * fetch_kb * 1024.0 / calls_f
    write = write_kb * 1024.0 / calls_w
    out["kernels"][fam] = {"launches_profiled": calls_f, "fetch_bytes_per_launch": round(fetch),
                           "write_bytes_per_launch": round(write), "traffic_bytes_per_launch": round(fetch + write)}
if not out["kernels"]:
    raise SystemExit("no bf16->bf16 NT GEMM rows found")
print(json.dumps(out))
