#!/usr/bin/env python
"""Socket power and shader clock per kernel (VERDICT r03 "Next" 3): the step runs power-limited (GEMMs at 1.7-1.9 GHz, attention
alone at 2.2), so joules per FLOP -- not alone-time -- decide what a kernel is worth inside the step.

    python tools/power_by_kernel.py [--seconds 1.5] [--rows 167936] [--no-step] > profiles/r04_power_by_kernel.txt

For each (GEMM variant x layer shape): launches back to back for `--seconds`, a sampler thread reads socket power and the gfx
clock every ~40 ms (amdsmi -> sysfs hwmon -> rocm-smi CLI, whichever works on the box; the source is printed), the window's
mean is reported next to HIP-event TF/s and TF/s per watt.  Then the same for the full training step (bf16 B=1024, bf16x3
B=512), the register-only MFMA ceiling loop and idle.  Developer library (experiment arms + the vendor yardstick)."""
import argparse
import glob
import os
import subprocess
import sys
import threading
import time

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402


class Sampler(threading.Thread):
    """(t, watts, MHz) samples; source chosen at construction"""

    def __init__(self, period=0.04):
        super().__init__(daemon=True)
        self.period, self.samples, self.stop_flag = period, [], False
        self.read, self.source = self._pick()

    def _pick(self):
        # 1. amdsmi python bindings
        try:
            import amdsmi
            amdsmi.amdsmi_init()
            h = amdsmi.amdsmi_get_processor_handles()[0]

            def rd():
                p = amdsmi.amdsmi_get_power_info(h)
                w = p.get("current_socket_power")
                if not isinstance(w, (int, float)) or w <= 0:
                    w = p.get("average_socket_power")
                c = amdsmi.amdsmi_get_clock_info(h, amdsmi.AmdSmiClkType.GFX)
                mhz = c.get("clk", c.get("cur_clk"))
                return float(w), float(mhz) if isinstance(mhz, (int, float)) else float("nan")
            w, m = rd()
            if w > 0:
                return rd, "amdsmi (current_socket_power | average_socket_power, GFX clk)"
        except Exception as e:                         # noqa: BLE001 -- any failure: next source
            sys.stderr.write("amdsmi unavailable: %r\n" % (e,))
        # 2. sysfs hwmon
        for hw in glob.glob("/sys/class/drm/card*/device/hwmon/hwmon*"):
            pw = [f for f in ("power1_input", "power1_average") if os.path.isfile(os.path.join(hw, f))]
            if not pw:
                continue
            pf, ff = os.path.join(hw, pw[0]), os.path.join(hw, "freq1_input")

            def rd(pf=pf, ff=ff):
                w = int(open(pf).read()) / 1e6
                mhz = int(open(ff).read()) / 1e6 if os.path.isfile(ff) else float("nan")
                return w, mhz
            try:
                if rd()[0] > 0:
                    return rd, "sysfs %s (+ freq1_input)" % pf
            except (OSError, ValueError):
                pass
        # 3. rocm-smi CLI (slow: ~5 Hz)
        def rd():
            import json
            out = subprocess.run(["rocm-smi", "--showpower", "--showclocks", "--json"], capture_output=True, text=True, timeout=10).stdout
            d = json.loads(out)
            card = d[sorted(d)[0]]
            w = [float(v) for k, v in card.items() if "ower" in k and "(W)" in k]
            mhz = [float(str(v).strip("()Mhz")) for k, v in card.items() if k.startswith("sclk clock speed")]
            return (w[0] if w else float("nan")), (mhz[0] if mhz else float("nan"))
        return rd, "rocm-smi --showpower --showclocks --json"

    def run(self):
        while not self.stop_flag:
            try:
                w, m = self.read()
                self.samples.append((time.perf_counter(), w, m))
            except Exception:                          # noqa: BLE001
                pass
            time.sleep(self.period)

    def window(self, t0, t1):
        xs = [(w, m) for t, w, m in self.samples if t0 <= t <= t1]
        if not xs:
            return float("nan"), float("nan"), 0
        ws = [w for w, _ in xs if w == w]
        ms = [m for _, m in xs if m == m]
        return (sum(ws) / len(ws) if ws else float("nan")), (sum(ms) / len(ms) if ms else float("nan")), len(xs)


def loop_for(fn, seconds, settle=0.4):
    """run fn() back to back for `seconds` after `settle` seconds of the same load; -> (t0, t1, launches, ms per launch)"""
    dev_sync = torch.cuda.synchronize
    t_end = time.perf_counter() + settle
    while time.perf_counter() < t_end:
        for _ in range(8):
            fn()
        dev_sync()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    n = 0
    t0 = time.perf_counter()
    e0.record()
    while time.perf_counter() - t0 < seconds:
        for _ in range(8):
            fn()
        n += 8
        dev_sync()                                  # bounded queue: the window is wall-clock accurate
    e1.record()
    dev_sync()
    t1 = time.perf_counter()
    return t0, t1, n, e0.elapsed_time(e1) / n


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seconds", type=float, default=1.5)
    ap.add_argument("--rows", type=int, default=1024 * 164)
    ap.add_argument("--no-step", action="store_true")
    ap.add_argument("--variants", default="81,90,100,200")
    args = ap.parse_args()
    import _knobs
    from visualbert_amd import _lib, ops
    L = _knobs.L
    dev = torch.device("cuda", 0)
    smp = Sampler()
    smp.start()
    print("# power source: %s" % smp.source)
    M = args.rows
    g = torch.Generator().manual_seed(0)
    time.sleep(1.0)
    t0 = time.perf_counter(); time.sleep(1.0); t1 = time.perf_counter()
    w, mhz, n = smp.window(t0, t1)
    print("%-58s %8s %8s %9s %9s %9s %6s" % ("what", "us", "TF/s", "W", "MHz", "TF/s/kW", "smpl"))
    print("%-58s %8s %8s %9.1f %9.0f %9s %6d" % ("idle", "-", "-", w, mhz, "-", n))
    rows = []

    def report(name, flops, t0, t1, ms):
        w, mhz, n = smp.window(t0 + 0.15, t1)        # the first 150 ms of a window still carry the previous load's average
        tf = flops / (ms * 1e-3) / 1e12 if flops else float("nan")
        rows.append((name, ms * 1e3, tf, w, mhz, n))
        print("%-58s %8.1f %8.1f %9.1f %9.0f %9.2f %6d" % (name, ms * 1e3, tf, w, mhz, tf / w * 1e3 if w == w and w > 0 else float("nan"), n), flush=True)

    shapes = [("QKV fwd  N=2304 K=768  bias", 2304, 768, "bias"), ("attn-out N=768  K=768  bias", 768, 768, "bias"),
              ("FFN-in   N=3072 K=768  bias+GELU+GELU'", 3072, 768, "gelu"), ("FFN-in   N=3072 K=768  bias (plain)", 3072, 768, "bias"),
              ("FFN-out  N=768  K=3072 bias", 768, 3072, "bias"), ("FFN-out dgrad N=3072 K=768 xGELU'+colsum", 3072, 768, "mulaux"),
              ("FFN-in dgrad N=768 K=3072 +addend", 768, 3072, "add")]
    for name, n, k, epi in shapes:
        a = (torch.randn(M, k, generator=g) * 0.5).to(torch.bfloat16).to(dev)
        wt = (torch.randn(n, k, generator=g) * 0.05).to(torch.bfloat16).to(dev)
        out = torch.empty(M, n, dtype=torch.bfloat16, device=dev)
        kw = dict(out=out)
        if epi in ("bias", "gelu"):
            kw["bias"] = torch.randn(n, generator=g).to(dev)
        if epi == "gelu":
            kw.update(act=_lib.VB_ACT_GELU_SAVE_GRAD, aux_out=torch.empty(M, n, dtype=torch.bfloat16, device=dev))
        elif epi == "mulaux":
            kw.update(act=_lib.VB_ACT_MUL_AUX, aux_in=torch.randn(M, n, device=dev).to(torch.bfloat16), colsum_out=torch.zeros(n, device=dev))
        elif epi == "add":
            kw["addend"] = torch.randn(M, n, device=dev).to(torch.bfloat16)
        for v in [int(x) for x in args.variants.split(",")]:
            if v == 200 and epi not in ("bias", "add"):
                continue                              # the vendor arm takes plain epilogues only
            _knobs.variant(v)
            try:
                t0, t1, cnt, ms = loop_for(lambda: ops.gemm(a, wt, M, n, k, **kw), args.seconds)
            except RuntimeError as e:
                print("%-58s %s" % ("%s  k%d" % (name, v), "failed: %s" % e))
                continue
            report("%s  k%d" % (name, v), 2.0 * M * n * k, t0, t1, ms)
        _knobs.variant(1)
        del a, wt, out, kw
        torch.cuda.empty_cache()

    # register-only MFMA loop (no memory traffic): what the matrix pipes alone draw
    sink = torch.empty(256 * 512, device=dev)
    t0, t1, cnt, ms = loop_for(lambda: L.vb_mfma_peak(2, 4000, 256, _lib.ptr(sink), _lib.stream_ptr()), args.seconds)
    report("mfma_peak (register-only, changing operands)", 256 * 8 * 4000 * 524288.0, t0, t1, ms)

    if not args.no_step:
        from visualbert_amd.data import synthetic_batch
        from visualbert_amd.model import AttrDict, ModelWrapper, VisualBERTFixedImageEmbedding
        from visualbert_amd.modeling import BertConfig
        for dtype, B, label in ((torch.bfloat16, 1024, "training step bf16 B=1024"), ("bf16x3", 512, "training step bf16x3 B=512")):
            torch.manual_seed(0)
            cfg = BertConfig(30522)
            model = VisualBERTFixedImageEmbedding(config=cfg, training_head_type="pretraining", visual_embedding_dim=2048,
                                                  compute_dtype=dtype).to(dev)
            model.train()
            mw = ModelWrapper(AttrDict(train_batch_size=B, learning_rate=5e-5, warmup_proportion=0.1, num_train_epochs=1), 100000 * B,
                              model=model)
            batch = synthetic_batch("pretraining", B, 128, 36, 2048, 30522, seed=0, device=dev)
            for _ in range(3):
                mw.step(batch)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            nst = 0
            while time.perf_counter() - t0 < max(3.0, 2 * args.seconds):
                mw.step(batch)
                nst += 1
                torch.cuda.synchronize()
            e1.record()
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            ms = e0.elapsed_time(e1) / nst
            report("%s (%.0f samples/s)" % (label, B / ms * 1e3), 110.54e9 * B, t0, t1, ms)
            del mw, model, batch
            torch.cuda.empty_cache()
    smp.stop_flag = True


if __name__ == "__main__":
    main()
