#!/usr/bin/env python
"""Reduce the two rocpd_pmc.py tables written by tools/gpu_pmc_traffic.sh to the per-launch HBM traffic of the bf16->bf16
persistent NT GEMM (all its epilogue instantiations together = bench.py's dominant kernel).
FETCH_SIZE / WRITE_SIZE are reported in KB; on gfx950 FETCH_SIZE tallies the 128-byte requests of wide coalesced reads
at 64 B (MI355X_MICROARCH.md, HBM section) -> doubled; WRITE_SIZE is taken as reported (uncalibrated in that guide).
usage: python tools/pmc_traffic.py <fetch table> <write table> <batch>"""
import json
import re
import sys


def table(path, counter):
    total, calls = 0.0, 0
    for line in open(path):
        if "gemm_nt_8ph_kernel" not in line or "DF16bDF16b" not in line:
            continue
        m = re.search(r"\s(\d+)\s+([0-9.eE+\-]+)\s*$", line.rstrip())
        if not m:
            continue
        calls += int(m.group(1))
        total += float(m.group(2))
    return total, calls


fetch_kb, calls_f = table(sys.argv[1], "FETCH_SIZE")
write_kb, calls_w = table(sys.argv[2], "WRITE_SIZE")
if not calls_f or not calls_w:
    raise SystemExit("no gemm_nt_8ph_kernel<bf16,bf16> rows found")
fetch = 2.0 * fetch_kb * 1024.0 / calls_f
write = write_kb * 1024.0 / calls_w
print(json.dumps({
    "kernel": "gemm_nt_8ph_kernel<bf16->bf16> (all epilogue instantiations)",
    "per_gpu_batch": int(sys.argv[3]),
    "launches_profiled": calls_f,
    "fetch_bytes_per_launch": round(fetch),
    "fetch_correction": "FETCH_SIZE x 1024 x 2 (gfx950 counts 128-B requests of wide coalesced reads at 64 B)",
    "write_bytes_per_launch": round(write),
    "traffic_bytes_per_launch": round(fetch + write),
    "source": "rocprofv3 --kernel-trace --pmc FETCH_SIZE / WRITE_SIZE, separate passes (tools/gpu_pmc_traffic.sh)",
}))
