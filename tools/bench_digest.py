#!/usr/bin/env python
"""One-screen digest of a bench.py JSON line (tools/gpu_session.sh prints it at the end of a session)."""
import json
import sys


def main():
    for path in sys.argv[1:]:
        try:
            d = json.loads(open(path).read().strip().splitlines()[-1])
        except (OSError, ValueError, IndexError) as e:
            print("%s: unreadable (%s)" % (path, e))
            continue
        r = d.get("roofline") or {}
        print("%s: %s %s  value %.1f %s  %.2f ms/step (median %s)  step_mfu %s / executed %s" % (
            path, d.get("dtype"), (d.get("config") or {}).get("per_gpu_batch"), d["value"], d["unit"], d["ms_per_step"], d.get("ms_per_step_median"),
            d.get("step_mfu"), d.get("step_mfu_executed")))
        if r:
            print("  roofline: %s frac %s  achieved %s %s  avg %s us  traffic %s" % (r.get("kernel", "")[:50], r.get("frac"), r.get("achieved"), r.get("unit"),
                                                                                 r.get("avg_launch_us"), r.get("traffic")))
            for k, v in sorted((r.get("hbm_bound") or {}).items()):
                print("    hbm %-62s %7.1f us %7.1f GB/s" % (k[:62], v["us"], v["GBps"]))
        s = d.get("strict_mode")
        if s:
            print("  strict %s: %.1f samples/s  max_dlogit %s  frac %s" % (s.get("dtype"), s["value"], s.get("max_dlogit"), (s.get("roofline") or {}).get("frac")))
            if "fp32_kernels" in s:
                print("  fp32 kernels: %.1f" % s["fp32_kernels"]["value"])
        v, c, p = d.get("vendor_plain_gemms"), d.get("cpu_baseline"), d.get("parity")
        if v:
            print("  vendor plain GEMMs: %.1f (%+.1f %%)" % (v["value"], 100 * (v["value"] / d["value"] - 1)))
        if c:
            print("  cpu: %s %s on %s cores" % (c["value"], c["unit"], c["cores"]))
        if p:
            print("  parity side batch: max %.3e  top1 %.4f  absmax %.2f" % (p["max_dlogit_vs_fp32_ref"], p["top1_agree"], p["logits_absmax"]))


if __name__ == "__main__":
    main()
