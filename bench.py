#!/usr/bin/env python
"""bench.py -- VisualBERT training throughput on MI355X (BASELINE.json metric).

One "step" = ModelWrapper.step on one synthetic batch: zero_grad -> forward (region projection, embeddings, 12
BertLayers, pooler, head, loss) -> backward -> gradient all-reduce (N > 1) -> fused BertAdam.  Nothing is skipped inside
the timed region: dropout is on (p = 0.1), the pre-training workload runs the tied 30522-wide decoder over all 164
positions like the reference, and the optimizer updates all 111.5 M optimised parameters.

    python bench.py                                   # N = 1, BASELINE configs[1] (pre-training, 36 regions + 128 tokens)
    python bench.py --gpus 8 --steps 20 --warmup 5    # spawns 8 ranks itself (torch.distributed.run, RCCL), rank 0 prints
    python bench.py --workload vqa | nlvr2            # BASELINE configs[3] / configs[4] shapes (fine-tuning heads)

Called under torch.distributed.run (WORLD_SIZE set) it is one rank of that job; called plainly with --gpus N > 1 it
re-executes itself through `python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1`.

Prints ONE JSON line on rank 0 (contract in the task statement) with extra objects:
  roofline     -- dominant kernel (the MFMA GEMM instantiation with the largest total time): algorithmic FLOPs per
                  launch / HIP-event duration per launch, against the dense bf16 MFMA peak; `hbm_bound` lists the
                  HBM-bound kernels (LayerNorm, BertAdam, MLM cross-entropy) as GB/s against the 8 TB/s peak
  cpu_baseline -- the oracle restatement of the reference (oracle/visualbert_oracle.py, "port") timed on this node's
                  host cores on a bounded sample, training mode (dropout on) (rank 0, N = 1 only)
  parity       -- bf16 kernels against the fp32 oracle on a B = 2 side batch (rank 0, N = 1 only)
  strict_mode  -- the split-operand bf16x3 mode (the MFMA mode that MEETS the north-star's 1e-3 logit tolerance) measured like
                  the headline: same step, same loop (>= 50 timed steps after warm-up, barrier + synchronize on both sides,
                  whole-loop rate + per-step median), its own `roofline` (dominant split-operand GEMM, ALGORITHMIC FLOPs --
                  not x 3 -- per HIP-event duration) and its own max|dlogit| against the fp32 oracle; + the fp32 kernels (short)
  vendor_plain_gemms -- the same step with the plain GEMMs handed to hipBLASLt (developer library, nt_kernel 200): a yardstick,
                  never `value`
  value_with_h2d -- the same step with every batch streamed from pinned host memory (SURVEY 8d's metric definition);
                  `value` is the HBM-resident rate the contract asks for
The driver's record of this line keeps the SCALAR entries of `roofline` and drops nested objects (round 5), so what a reader needs to
judge a round is ALSO there flat: strict_value / strict_ms_per_step / strict_max_dlogit / strict_frac / strict_meets_tolerance (the
compliant mode), parity_bf16_max_dlogit, mfma_ceiling_before_loop / _after_loop and all_gemm_frac_of_measured_ceiling (every GEMM launch
against the register-only MFMA ceiling measured right before and right after the timed loop: comparable across boxes), value_with_h2d,
value_sparse_mlm_head_optin (the opt-in head that never materialises the dense logits; a short extra leg, never `value`),
batch_curve_b<B>_ms for per-GPU batches 8 ... 512 + the headline's (the regime of the reference's own configs: 6-8 per GPU).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

PEAK_BF16_TFLOPS = 2500.0      # dense bf16 MFMA peak, /opt/skills/guides/MI355X_MICROARCH.md
PEAK_F32_TFLOPS = 157.3
PEAK_HBM_GBPS = 8000.0

# --dtype -> compute dtype handed to the model (visualbert_amd.modeling.set_compute_dtype)
DTYPES = {"bf16": "bfloat16", "fp32": "float32", "bf16x3": "bf16x3"}
# the short re-run of this command line that the PMC passes profile (measure_traffic)
PMC_CHILD_FLAGS = ["--steps", "2", "--warmup", "1", "--no-cpu-baseline", "--no-profile", "--no-h2d", "--no-parity", "--strict-dtype", "none",
                   "--no-vendor-leg", "--pmc-traffic", "off", "--no-batch-curve", "--no-sparse-leg"]

# BASELINE.json configs -> (head, text tokens, regions, feature width, label for config.workload)
WORKLOADS = {
    "pretrain": dict(head="pretraining", T=128, R=36, Dv=2048, batch=1024, cfg="configs[1]",
                     what="MLM + image-text-match pre-training step incl. dropout, dense 30522-wide decoder, BertAdam"),
    "vqa": dict(head="vqa", T=20, R=36, Dv=2048, batch=3072, cfg="configs[3]",
                what="VQA2.0 fine-tuning step (3129-answer head, KL-div on soft scores) incl. dropout, BertAdam"),
    "nlvr2": dict(head="nlvr", T=40, R=72, Dv=2048, batch=1536, cfg="configs[4]",
                  what="NLVR2 paired-image fine-tuning step (2 x 36 regions, 2-way head) incl. dropout, BertAdam"),
    # NLVR2 as the reference really runs it (configs/nlvr2/fine-tune.json:5,7; dataloaders/nlvr_dataset.py:98-106): 2 x 144 regions of
    # 1024-d detectron features + 128 text tokens, S = 416 -- the long-sequence attention stress SURVEY 8d names (232 GF trained / sample)
    "nlvr2-real": dict(head="nlvr", T=128, R=288, Dv=1024, batch=384, cfg="configs[4] at the reference's own shape (nlvr2/fine-tune.json)",
                       what="NLVR2 paired-image fine-tuning step (2 x 144 regions x 1024-d, 2-way head) incl. dropout, BertAdam"),
}


def compute_dtype_of(name):
    """what the model's compute_dtype argument takes: a torch dtype, or the string "bf16x3" (split-operand GEMM mode)"""
    import torch
    v = DTYPES[name]
    return v if v == "bf16x3" else getattr(torch, v)


def flops_per_sample(L, H, I, V, S, R, Dv, head):
    """forward FLOPs (multiply-add = 2); training = 3x."""
    if head == "pretraining":
        hd = 2 * S * H * H + 2 * S * H * V + 4 * H
    elif head == "vqa":
        hd = 2 * H * 3129
    else:
        hd = 4 * H
    fwd = 2 * R * Dv * H + L * (8 * S * H * H + 4 * S * S * H + 4 * S * H * I) + 2 * H * H + hd
    return 3 * fwd


def physical_cores():
    """physical cores of this host (unique (package, core) pairs of /proc/cpuinfo); half the logical count as a fallback"""
    try:
        pairs, phys = set(), None
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.startswith("physical id"):
                    phys = line.split(":")[1].strip()
                elif line.startswith("core id"):
                    pairs.add((phys, line.split(":")[1].strip()))
        if pairs:
            return len(pairs)
    except OSError:
        pass
    return max(1, (os.cpu_count() or 2) // 2)


def free_port():
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def respawn_as_ranks(n):
    """`python bench.py --gpus N` typed plainly: become N ranks through torch.distributed.run (one process per GPU)."""
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n),
           "--master-addr", "127.0.0.1", "--master-port", str(free_port()), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", "8")
    sys.stdout.flush()
    os.execvpe(sys.executable, cmd, env)


def pmc_traffic(batch, key, workload):
    """HBM bytes per launch of the dominant kernel from the COMMITTED PMC passes (tools/gpu_pmc_traffic.sh ->
    profiles/pmc_traffic.json): the fallback when the in-run passes (measure_traffic) are switched off or fail;
    (None, why) when the file does not match this run."""
    path = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    try:
        with open(path) as f:
            d = json.load(f)
    except (OSError, ValueError):
        return None, "no profiles/pmc_traffic.json"
    if workload != "pretrain" or d.get("per_gpu_batch") != batch or (key & 15) or (key & 256):
        return None, "committed PMC pass is for another configuration"
    fam = "gemm_nt_dual_kernel<bf16->bf16>" if key & 64 else ("gemm_nt_8ph_kernel<bf16->bf16>" if key & 16 else None)
    ent = d.get("kernels", {}).get(fam)
    if ent is None:
        return None, "committed PMC pass has no entry for this kernel"
    return ent["traffic_bytes_per_launch"], ("committed PMC pass profiles/pmc_traffic.json (tools/gpu_pmc_traffic.sh: separate "
                                             "FETCH_SIZE / WRITE_SIZE runs of this command, 2 x FETCH_SIZE + WRITE_SIZE); "
                                             "not measured inside this run")


def kernel_name_filter(key):
    """predicate over rocprofv3's kernel names that picks the dominant GEMM family's instantiations.  rocprofv3 demangles some
    instantiations and leaves others mangled (its demangler trips over the __bf16 template argument), so both spellings are
    matched: family name, split-operand flag (last template argument), output type."""
    if key & 3:
        return None
    fam = "gemm_nt_dual_kernel" if key & 64 else ("gemm_nt_8ph_kernel" if key & 16 else None)
    if fam is None:
        return None
    x3, to_f32 = bool(key & 256), bool(key & 4)

    def match(name):
        if fam not in name:
            return False
        if (("Lb1E" in name) or (", true>" in name)) != x3:
            return False
        if x3:
            return True                                       # every split-operand instantiation writes fp32 (or a split image)
        if fam + "I" in name:                                 # mangled: dual <TO, ...>, 8ph <T, TO, ...>
            tag = fam + ("I" if key & 64 else "IDF16b") + ("f" if to_f32 else "DF16b")
            return tag in name
        if key & 64:                                          # demangled: the first template argument is the output type
            return (fam + "<float" in name) == to_f32
        return ("_Accum, float" in name) == to_f32
    return match


def measure_traffic(key, child_args, timeout=240):
    """HBM traffic per launch of the dominant kernel, MEASURED FOR THIS RUN: rocprofv3 cannot wrap a process from the inside, so
    bench.py re-runs its own command line twice (2 steps each) as `rocprofv3 --kernel-trace --pmc <counter>` children -- separate
    FETCH_SIZE and WRITE_SIZE passes, kernel trace only, as /opt/skills/guides/MI355X_MICROARCH.md prescribes -- and sums the
    counters over the launches of that kernel family: 2 x FETCH_SIZE (gfx950 tallies the 128-byte requests of wide coalesced
    reads at 64 B) + WRITE_SIZE, both reported in KB.  -> (bytes per launch, source string) or (None, why)."""
    import shutil
    import sqlite3
    import subprocess
    import tempfile
    match = kernel_name_filter(key)
    exe = shutil.which("rocprofv3")
    if match is None or exe is None:
        return None, "no rocprofv3 on this machine" if exe is None else "dominant kernel is not an NT GEMM family"
    per_launch, launches = {}, 0
    env = dict(os.environ, TMPDIR="/tmp")
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK"):
        env.pop(k, None)
    for counter in ("FETCH_SIZE", "WRITE_SIZE"):
        d = tempfile.mkdtemp(prefix="vb_pmc_", dir="/tmp")
        cmd = [exe, "--kernel-trace", "--pmc", counter, "-d", d, "-o", "p", "--", sys.executable, os.path.abspath(__file__)] + child_args
        try:
            # own session = own process group: on a timeout the WHOLE group goes (rocprofv3 and the python grandchild that holds
            # GPU memory), not just the direct child
            proc = subprocess.Popen(cmd, env=env, cwd="/tmp", stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, start_new_session=True)
            try:
                proc.wait(timeout=timeout)
            except subprocess.TimeoutExpired:
                import signal
                try:
                    os.killpg(proc.pid, signal.SIGKILL)
                except OSError:
                    pass
                proc.wait()
                raise
            dbs = [os.path.join(r, f) for r, _, fs in os.walk(d) for f in fs if f.endswith("_results.db")]
            if not dbs:
                return None, "rocprofv3 --pmc %s pass left no results database" % counter
            cur = sqlite3.connect(dbs[0]).cursor()
            cols = [r[1] for r in cur.execute("pragma table_info(counters_collection)").fetchall()]
            kcol = "kernel_name" if "kernel_name" in cols else "name"
            vcol = "value" if "value" in cols else "counter_value"
            dcol = "dispatch_id" if "dispatch_id" in cols else kcol
            total, calls = 0.0, 0
            for name, v, n in cur.execute("select %s, sum(%s), count(distinct %s) from counters_collection group by %s"
                                          % (kcol, vcol, dcol, kcol)).fetchall():
                if match(name):
                    total += float(v)
                    calls += int(n)
            if not calls:
                return None, "no launches of the dominant kernel family in the %s pass" % counter
            per_launch[counter], launches = total * 1024.0 / calls, calls
        except (subprocess.TimeoutExpired, sqlite3.Error, OSError) as e:
            return None, "rocprofv3 --pmc %s pass failed: %r" % (counter, e)
        finally:
            shutil.rmtree(d, ignore_errors=True)
    traffic = 2.0 * per_launch["FETCH_SIZE"] + per_launch["WRITE_SIZE"]
    return int(round(traffic)), ("measured by this run: two `rocprofv3 --kernel-trace --pmc` children of this command line "
                                 "(FETCH_SIZE, WRITE_SIZE; %d launches of the kernel family; 2 x FETCH_SIZE + WRITE_SIZE, KB -> bytes: the "
                                 "doubling is MI355X_MICROARCH.md's gfx950 correction for wide coalesced reads, calibrated there on a "
                                 "streaming copy -- the RAW counters are FETCH_SIZE %.0f MB, WRITE_SIZE %.0f MB per launch, so the true "
                                 "figure lies between %.0f and %.0f MB); fetch %.0f MB + write %.0f MB per launch"
                                 % (launches, per_launch["FETCH_SIZE"] / 1e6, per_launch["WRITE_SIZE"] / 1e6,
                                    (per_launch["FETCH_SIZE"] + per_launch["WRITE_SIZE"]) / 1e6, traffic / 1e6,
                                    2.0 * per_launch["FETCH_SIZE"] / 1e6, per_launch["WRITE_SIZE"] / 1e6))


def measured_mfma_ceiling(dev):
    """what this chip's matrix pipes deliver on bf16 operands that change every instruction, with NO memory traffic (a
    register-only v_mfma_f32_16x16x32_bf16 loop on every CU, ~12 ms; the measurement kernel lives in the developer build
    of the library, include/visualbert_hip_dev.h).  None when that build is absent."""
    import torch
    from visualbert_amd import _lib
    L = _lib.dev_lib()
    if L is None:
        return None
    blocks, iters = 256, 20000
    out = torch.empty(blocks * 512, device=dev)
    L.vb_mfma_peak(2, iters, blocks, _lib.ptr(out), _lib.stream_ptr())
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(3):
        L.vb_mfma_peak(2, iters, blocks, _lib.ptr(out), _lib.stream_ptr())
    e1.record()
    torch.cuda.synchronize()
    return blocks * 8 * iters * 524288.0 / (e0.elapsed_time(e1) / 3) / 1e9


def cpu_baseline(batch_size, T, R, head, steps=3):
    """the oracle (a restatement of TrainVisualBERTObjective + ModelWrapper.step + BertAdam) on host cores, in training
    mode: dropout p = 0.1 at the reference's four sites, as the reference trains."""
    import torch
    from oracle import visualbert_oracle as vo
    # under torch.distributed.run every rank starts with OMP_NUM_THREADS=8; by now the other ranks are idle at the final
    # barrier, so rank 0 takes the node's PHYSICAL cores like the single-process run does (never the SMT siblings: 256
    # OpenMP threads on 128 cores made this leg >10x slower and timed the r03 GPU sessions out)
    if torch.get_num_threads() < physical_cores():
        torch.set_num_threads(physical_cores())
    cfg = vo.OracleConfig(**vo.CONFIGS["base"])
    sd = vo.synth_state_dict(cfg, head, 0, perturb=False)
    batch = vo.synth_batch(cfg, batch_size, T, R, 0, head, ragged=False)
    state = {}
    with vo.dropout(0.1, 0.1):
        vo.train_step(sd, cfg, head, batch, state, 5e-5, 0.1, 1000)          # warm-up
        t0 = time.time()
        for _ in range(steps):
            vo.train_step(sd, cfg, head, batch, state, 5e-5, 0.1, 1000)
        dt = (time.time() - t0) / steps
    return dict(value=round(batch_size / dt, 3), unit="samples/s", cores=torch.get_num_threads(), kind="port",
                sample="%d full fp32 training steps (forward+backward+BertAdam, model.train(): dropout 0.1) of BERT-base "
                       "VisualBERT (%s head), batch %d x (%d tok + %d regions), oracle/visualbert_oracle.py on torch CPU, "
                       "%d threads, %.2f s/step" % (steps, head, batch_size, T, R, torch.get_num_threads(), dt))


SIDE_BATCH = 16


def parity_side_batch(model, dev, head, T, R, Dv=0):
    """bf16 kernels against the fp32 oracle (the reference's arithmetic) on a B = 16 ragged side batch, eval mode (B = 2 until
    round 4; the weights are whatever the timed steps left: |logit|max ~ 12-15, trained-like statistics)."""
    import torch
    from oracle import visualbert_oracle as vo
    cfg = vo.OracleConfig(**dict(vo.CONFIGS["base"], **({"visual_embedding_dim": Dv} if Dv else {})))
    sd = {k: v.detach().float().cpu() for k, v in model.bert.state_dict().items() if k in vo.param_shapes(cfg, head)}
    batch = vo.synth_batch(cfg, SIDE_BATCH, T, R, 77, head, ragged=True)
    with torch.no_grad():
        ref = vo.objective_forward(sd, cfg, head, mode="fp32", **batch)
    was = model.training
    model.eval()
    with torch.no_grad():
        out = model(**{k: v.to(dev) for k, v in batch.items()})
    model.train(was)
    lg = out["logits"].float().cpu().reshape(ref["logits"].shape)
    d = (lg - ref["logits"]).abs().flatten()
    k = max(1, int(d.numel() * 1e-3))
    return dict(reference="fp32 oracle (oracle/visualbert_oracle.py, pinned to the real reference by tests/golden), B=%d ragged, eval" % SIDE_BATCH,
                max_dlogit_vs_fp32_ref=float(d.max()), mean=float(d.mean()), p999=float(d.topk(k).values[-1]),
                top1_agree=float((lg.argmax(-1) == ref["logits"].argmax(-1)).float().mean()),
                dloss=abs(float(out["loss"]) - float(ref["loss"])), logits_absmax=float(ref["logits"].abs().max()),
                north_star_tolerance=1e-3,
                note="plain bf16 MFMA operands cannot meet 1e-3 (profiles/r02_bf16_error_budget.txt); the split-operand bf16x3 "
                     "mode and the fp32 kernels do (tests/test_parity_at_scale.py; `strict_mode` in this line)")


def timed_steps(mw, batch, steps, warmup, barrier, profile=True):
    """THE measurement loop (headline and strict mode alike): `warmup` untimed steps, then exactly `steps` steps between
    barrier + synchronize pairs.  -> (wall seconds of the loop, median ms of the per-step HIP events, per-GEMM-kernel HIP-event
    summary of the timed region or None)."""
    import torch
    from visualbert_amd import ops
    for _ in range(warmup):
        mw.step(batch)
    if profile:
        ops.gemm_profile_start()
    barrier()
    marks = [torch.cuda.Event(enable_timing=True) for _ in range(steps + 1)]
    t0 = time.perf_counter()
    marks[0].record()
    for i in range(steps):
        mw.step(batch)
        marks[i + 1].record()                              # no host sync: the median step time is read after the loop
    barrier()
    elapsed = time.perf_counter() - t0
    per_step = sorted(marks[i].elapsed_time(marks[i + 1]) for i in range(steps))
    median_ms = per_step[len(per_step) // 2] if len(per_step) % 2 else 0.5 * (per_step[len(per_step) // 2 - 1] + per_step[len(per_step) // 2])
    summ = ops.gemm_profile_stop() if profile else None
    return elapsed, median_ms, summ


def traffic_for(key, batch, workload, child_args):
    """roofline.traffic: measured for this run when child_args is given (measure_traffic), else / on failure the committed pass"""
    traffic, src = (None, "switched off (--pmc-traffic off, or N > 1)")
    if child_args is not None:
        traffic, src = measure_traffic(key, child_args)
    if traffic is None:
        why = src
        traffic, src = pmc_traffic(batch, key, workload)
        src = "%s [in-run PMC passes: %s]" % (src, why)
    return traffic, src


def dominant_key(summ):
    return max(summ.items(), key=lambda kv: kv[1]["ms"])[0]


def roofline_of(summ, steps, peak, batch, workload, traffic_child_args=None):
    """dominant kernel = the GEMM instantiation with the largest total HIP-event time inside the timed region; achieved =
    its algorithmic FLOPs (2 M N K per launch: the split-operand mode's three MFMA passes are NOT counted three times) over its
    summed launch durations."""
    from visualbert_amd import ops
    key, d = max(summ.items(), key=lambda kv: kv[1]["ms"])
    ach = d["flops"] / (d["ms"] * 1e-3) / 1e12
    traffic, traffic_src = traffic_for(key, batch, workload, traffic_child_args)
    # the dispatch rule moves shapes between the two K-contiguous kernels from round to round (round 5: the long-K + addend
    # dgrads left the two-workgroup kernel for the persistent one), so the dominant kernel's own rate is not comparable across
    # rounds -- this entry is: every K-contiguous GEMM with the dominant kernel's output type, whichever kernel ran it
    tag = ops.gemm_key_name(key).split(",")[0].split("<")[1]          # "bf16->bf16" / "bf16->fp32"
    x3 = "[bf16x3" in ops.gemm_key_name(key)
    fam = [v for k, v in summ.items() if ops.gemm_key_name(k).startswith("gemm_nt_") and ("<" + tag) in ops.gemm_key_name(k)
           and (("[bf16x3" in ops.gemm_key_name(k)) == x3)]
    fam = fam or [d]
    fam_tf = sum(v["flops"] for v in fam) / (sum(v["ms"] for v in fam) * 1e-3) / 1e12
    return dict(bound="mfma", achieved=round(ach, 2), peak=peak, unit="TFLOP/s", frac=round(ach / peak, 4),
                nt_gemms_same_output_type=dict(tflops=round(fam_tf, 1), frac=round(fam_tf / peak, 4),
                                               launches_per_step=sum(v["launches"] for v in fam) / steps,
                                               ms_per_step=round(sum(v["ms"] for v in fam) / steps, 3)),
                traffic=traffic, traffic_source=traffic_src,
                kernel=ops.gemm_key_name(key),
                launches_per_step=d["launches"] / steps,
                avg_launch_us=round(d["ms"] * 1e3 / d["launches"], 2),
                gflop_per_launch=round(d["flops"] / d["launches"] / 1e9, 3),
                all_gemm_tflops=round(sum(x["flops"] for x in summ.values()) /
                                      (sum(x["ms"] for x in summ.values()) * 1e-3) / 1e12, 2),
                gemm_ms_per_step=round(sum(x["ms"] for x in summ.values()) / steps, 3),
                by_kernel={ops.gemm_key_name(k):
                           dict(ms_per_step=round(v["ms"] / steps, 3),
                                launches_per_step=v["launches"] / steps,
                                tflops=round(v["flops"] / (v["ms"] * 1e-3) / 1e12, 1))
                           for k, v in summ.items()})


def strict_mode(dev, head, T, R, Dv, V, batch, steps, warmup, dtype_name, flops_per_sample_, workload, full=True, pmc=False):
    """the mode that MEETS the north-star's logit tolerance (<= 1e-3 against the fp32 reference), measured on the same step
    (dropout on, dense decoder, BertAdam) with the same loop as the headline (timed_steps) and -- full=True -- its own roofline
    object, plus its max|dlogit| against the oracle on the B = 2 side batch.  The headline `value` stays the bf16 number
    BASELINE.json's config names; this object is the compliant mode's, first-class."""
    import torch
    from visualbert_amd.data import synthetic_batch
    from visualbert_amd.model import AttrDict, ModelWrapper, VisualBERTFixedImageEmbedding
    from visualbert_amd.modeling import BertConfig
    torch.manual_seed(1234)
    config = BertConfig(V, hidden_size=768, num_hidden_layers=12, num_attention_heads=12, intermediate_size=3072)
    model = VisualBERTFixedImageEmbedding(config=config, training_head_type=head, visual_embedding_dim=Dv,
                                          compute_dtype=compute_dtype_of(dtype_name)).to(dev)
    model.train()
    mw = ModelWrapper(AttrDict(train_batch_size=batch, learning_rate=5e-5, warmup_proportion=0.1, num_train_epochs=1,
                               gradient_accumulation_steps=1), 1000 * batch, model=model)
    b = synthetic_batch(head, batch, T, R, Dv, V, seed=0, device=dev)
    elapsed, median_ms, summ = timed_steps(mw, b, steps, warmup, torch.cuda.synchronize, profile=full)
    dt = elapsed / steps
    par = parity_side_batch(model, dev, head, T, R, Dv)
    peak = PEAK_BF16_TFLOPS if dtype_name != "fp32" else PEAK_F32_TFLOPS
    out = dict(dtype=dtype_name, value=round(batch / dt, 2), unit="samples/s", per_gpu_batch=batch, steps=steps, warmup=warmup,
               ms_per_step=round(dt * 1e3, 3), ms_per_step_median=round(median_ms, 3),
               max_dlogit=par["max_dlogit_vs_fp32_ref"], mean_dlogit=par["mean"],
               top1_agree=par["top1_agree"], dloss=par["dloss"], north_star_tolerance=1e-3,
               meets_tolerance=bool(par["max_dlogit_vs_fp32_ref"] <= 1e-3),
               tflops=round(batch / dt * flops_per_sample_ / 1e12, 1),
               step_mfu=round(batch / dt * flops_per_sample_ / (peak * 1e12), 4))
    if full and summ:
        del mw, model, b                            # the PMC children need the memory
        torch.cuda.empty_cache()
        child = ["--workload", workload, "--dtype", dtype_name, "--batch", str(batch)] + PMC_CHILD_FLAGS if pmc else None
        out["roofline"] = roofline_of(summ, steps, peak, batch, workload, child)
        out["roofline"]["note"] = ("algorithmic FLOPs: the three bf16 MFMA passes of a split-operand product count once; the matrix "
                                   "pipe executes 3 x `achieved`")
    mw = model = b = None
    torch.cuda.empty_cache()
    return out


def hbm_bound_kernels(model, M, H, dev, optimizer=None, V=0):
    """the HBM-bound kernels of the step timed alone: algorithmic bytes / event time against the 8 TB/s peak."""
    import torch
    from visualbert_amd import _lib
    L = _lib.lib()
    dt = torch.bfloat16
    x = torch.randn(M, H, device=dev).to(dt)
    r = torch.randn(M, H, device=dev).to(dt)
    gamma, beta = torch.ones(H, device=dev), torch.zeros(H, device=dev)
    y, z = torch.empty_like(x), torch.empty_like(x)
    mean, rstd = torch.empty(M, device=dev), torch.empty(M, device=dev)
    dz, dx = torch.empty_like(x), torch.empty_like(x)
    dg, db, dbias = torch.zeros(H, device=dev), torch.zeros(H, device=dev), torch.zeros(H, device=dev)
    ws = torch.empty(L.vb_ln_bwd_ws_bytes(M, H) // 4, device=dev)

    def timed(fn, n=10):
        for _ in range(2):
            fn()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n):
            fn()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / n * 1e-3

    t_f = timed(lambda: L.vb_ln_fwd(_lib.VB_BF16, _lib.ptr(x), _lib.ptr(r), _lib.ptr(z), _lib.ptr(y), _lib.ptr(mean),
                                    _lib.ptr(rstd), _lib.ptr(gamma), _lib.ptr(beta), M, H, 1e-12, 0.1, 11, 0.0, 12, 5,
                                    _lib.stream_ptr()))
    t_b = timed(lambda: L.vb_ln_bwd(_lib.VB_BF16, _lib.ptr(x), _lib.ptr(z), _lib.ptr(mean), _lib.ptr(rstd), _lib.ptr(gamma),
                                    _lib.ptr(dz), _lib.ptr(dx), _lib.ptr(dg), _lib.ptr(db), _lib.ptr(dbias), M, H, 0.1, 11,
                                    0.0, 12, 5, _lib.ptr(ws), _lib.stream_ptr()))
    rows_list = [("ln_fwd 4-tensor form (dropout + residual + LayerNorm, z written: heads / embeddings / H > 768)", t_f, 4 * M * H * 2),
                 ("ln_bwd 4-tensor form", t_b, 4 * M * H * 2)]
    # what the encoder layers SHIP (vb_bert_layer_fwd / _bwd -> vb_ln_fwd_rb / vb_ln_bwd_rb): the forward skips the pre-LN sum, the
    # backward rebuilds x-hat from y -- three [M, H] streams per launch (x, resid in, y out | dy, y in, dz out; dx = dz here: no input
    # dropout on this site's gradient path is a separate stream only when p_in > 0, as in the layer: dy, y in; dz, dx out)
    flag = torch.zeros(1, dtype=torch.int32, device=dev)
    t_f3 = timed(lambda: L.vb_ln_fwd_rb(_lib.VB_BF16, _lib.ptr(x), _lib.ptr(r), _lib.ptr(z), _lib.ptr(y), _lib.ptr(mean),
                                        _lib.ptr(rstd), _lib.ptr(gamma), _lib.ptr(beta), M, H, 1e-12, 0.1, 11, 5, _lib.ptr(flag),
                                        _lib.stream_ptr()))
    assert int(flag.item()) == 1, "gamma = 1, beta = 0 must be rebuildable"
    t_b3 = timed(lambda: L.vb_ln_bwd_rb(_lib.VB_BF16, _lib.ptr(x), _lib.ptr(z), _lib.ptr(mean), _lib.ptr(rstd), _lib.ptr(gamma),
                                        _lib.ptr(dz), _lib.ptr(dx), _lib.ptr(dg), _lib.ptr(db), _lib.ptr(dbias), M, H, 0.1, 11, 5,
                                        _lib.ptr(ws), _lib.ptr(y), _lib.ptr(beta), _lib.ptr(flag), _lib.stream_ptr()))
    rows_list += [("ln_fwd as shipped in the layers (3 tensors: x-hat rebuilt from y in the backward)", t_f3, 3 * M * H * 2),
                  ("ln_bwd as shipped in the layers (dy, y in; dz, dx out)", t_b3, 4 * M * H * 2)]
    if optimizer is not None:                            # the fused multi-tensor BertAdam step on the live arena (after the timed region)
        n_par = sum(p.numel() for p in model.parameters())
        t_a = timed(lambda: optimizer.step(), n=5)
        # fp32 p, g, m, v read (16 B) + p, m, v written (12 B) + the bf16 shadow written (2 B) per parameter
        rows_list.append(("bert_adam_step (fused multi-tensor)", t_a, 30 * n_par))
    if V:                                                # MLM cross-entropy over the labelled rows (~11.7 % of the tokens), compact gradient
        ld = (V + 63) // 64 * 64
        g = torch.Generator().manual_seed(0)
        lab = torch.full((M,), -1, dtype=torch.int64)
        sel = torch.rand(M, generator=g) < 0.117
        lab[sel] = torch.randint(0, V, (int(sel.sum()),), generator=g)
        lab = lab.to(dev)
        rows = torch.nonzero(lab != -1).reshape(-1)
        n = rows.numel()
        n_pad = (n + 63) // 64 * 64
        logits = torch.randn(M, ld, device=dev)
        acc, loss = torch.empty(66, device=dev), torch.empty(1, device=dev)
        dlc = torch.empty(n_pad, ld, dtype=dt, device=dev)
        t_c = timed(lambda: L.vb_ce_fwd_bwd_rows(_lib.VB_BF16, _lib.ptr(logits), ld, _lib.ptr(lab), -1, _lib.ptr(rows), n, n_pad,
                                                 _lib.ptr(acc), _lib.ptr(loss), _lib.ptr(dlc), ld, M, V, _lib.stream_ptr()))
        rows_list.append(("mlm_cross_entropy (labelled rows, gradient written compactly)", t_c, n * V * 4 + n_pad * ld * 2))
        del logits, dlc
    out = {}
    for name, t, nbytes in rows_list:
        out[name] = dict(us=round(t * 1e6, 1), GBps=round(nbytes / t / 1e9, 1), frac_of_peak=round(nbytes / t / 1e9 / PEAK_HBM_GBPS, 3),
                         algorithmic_bytes=nbytes)
    return out


BATCH_CURVE = [8, 16, 32, 64, 128, 256, 512]


def batch_curve(dev, head, T, R, Dv, V, dtype_name, fps, peak, steps=12, warmup=3):
    """the same training step at the per-GPU batches the reference's own configs live in (global batch 48-64 over 8 GPUs =
    6-8 per GPU: configs/vqa/coco-pre-train.json:17, models/train.py:146) up to 512: ms per step = median of `steps` per-step HIP
    events after `warmup` steps, a fresh model per batch size.  VERDICT r05 item 1: the driver's record keeps this curve."""
    import torch
    from visualbert_amd.data import synthetic_batch
    from visualbert_amd.model import AttrDict, ModelWrapper, VisualBERTFixedImageEmbedding
    from visualbert_amd.modeling import BertConfig
    out = {}
    for B in BATCH_CURVE:
        torch.manual_seed(1234)
        config = BertConfig(V, hidden_size=768, num_hidden_layers=12, num_attention_heads=12, intermediate_size=3072)
        model = VisualBERTFixedImageEmbedding(config=config, training_head_type=head, visual_embedding_dim=Dv,
                                              compute_dtype=compute_dtype_of(dtype_name)).to(dev)
        model.train()
        mw = ModelWrapper(AttrDict(train_batch_size=B, learning_rate=5e-5, warmup_proportion=0.1, num_train_epochs=1,
                                   gradient_accumulation_steps=1), (steps + warmup + 20) * B, model=model)
        b = synthetic_batch(head, B, T, R, Dv, V, seed=0, device=dev)
        _, med, _ = timed_steps(mw, b, steps, warmup, torch.cuda.synchronize, profile=False)
        out[B] = dict(ms_per_step=round(med, 3), samples_per_s=round(B / med * 1e3, 1),
                      step_mfu=round(B / med * 1e3 * fps / (peak * 1e12), 4))
        del mw, model, b
        torch.cuda.empty_cache()
    return out


def selftest_launch():
    """plumbing check of the self-launch path without a GPU (tests/test_bench_launch.py): every rank joins a gloo group,
    all-reduces a one, and rank 0 prints the JSON line."""
    import torch
    import torch.distributed as dist
    world, rank = int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("RANK", "0"))
    dist.init_process_group(backend="gloo", rank=rank, world_size=world)
    one = torch.ones(1)
    dist.all_reduce(one)
    if rank == 0:
        print(json.dumps({"selftest": "launch", "n_gpus": world, "ranks_seen": int(one.item())}), flush=True)
    dist.destroy_process_group()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--workload", default="pretrain", choices=sorted(WORKLOADS))
    ap.add_argument("--batch", type=int, default=0, help="per-GPU batch (weak scaling); 0 = the workload's default")
    ap.add_argument("--dtype", default="bf16", choices=list(DTYPES))
    ap.add_argument("--strict-dtype", default="both", choices=[d for d in DTYPES if d != "bf16"] + ["both", "none"],
                    help="after the timed bf16 run, also time a short run of the mode that meets the north-star's 1e-3 "
                         "logit tolerance and report it as `strict_mode` (rank 0, N = 1)")
    ap.add_argument("--strict-batch", type=int, default=1024, help="per-GPU batch of the strict-mode leg (fp32 activations: ~190 GB at 1024)")
    ap.add_argument("--strict-steps", type=int, default=0,
                    help="timed steps of the bf16x3 leg; 0 = max(--steps, 50): SURVEY 8d asks for >= 50 steps and the median")
    ap.add_argument("--strict-warmup", type=int, default=0, help="0 = --warmup")
    ap.add_argument("--text-len", type=int, default=0)
    ap.add_argument("--regions", type=int, default=0)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-batch", type=int, default=8, help="batch of the CPU baseline leg (SURVEY 8d: B = 8, >= 3 timed steps after 1 warm-up)")
    ap.add_argument("--no-profile", action="store_true", help="skip the per-GEMM HIP-event timing")
    ap.add_argument("--no-overlap", action="store_true")
    ap.add_argument("--no-h2d", action="store_true", help="skip the second timed loop that streams every batch from pinned host memory")
    ap.add_argument("--no-parity", action="store_true")
    ap.add_argument("--force-dist", action="store_true", help="initialise torch.distributed/RCCL even with one rank")
    ap.add_argument("--comm", default="abi", choices=["abi", "torch"],
                    help="gradient all-reduce through the C ABI's RCCL communicator (vb_comm_*) or torch.distributed")
    ap.add_argument("--sparse-mlm-head", action="store_true",
                    help="also time the opt-in MLM head over the labelled positions only (SURVEY 8f/N1); reported as an extra field")
    ap.add_argument("--no-vendor-leg", action="store_true",
                    help="skip the extra timed leg with the plain GEMMs on hipBLASLt (N = 1 only; never part of `value`)")
    ap.add_argument("--nt-kernel", type=int, default=0,
                    help="vb_stream_opts.nt_kernel for the whole run (0 = chosen per shape; 81 / 90 for A/B runs)")
    ap.add_argument("--attn-two-pass", type=int, default=0,
                    help="vb_stream_opts.attn_two_pass for the whole run (A/B: 1 = two-pass attention backward)")
    ap.add_argument("--lib-path", default="", help="developer A/B runs only: bind this build of libvisualbert_hip.so instead of the in-tree one")
    ap.add_argument("--dev-debug", type=int, default=0,
                    help="developer A/B runs only: bind libvisualbert_hip_dev.so for the whole run and set vb_gemm_set_debug(bits)")
    ap.add_argument("--pmc-traffic", default="auto", choices=["auto", "off"],
                    help="auto (N = 1): measure roofline.traffic for THIS run with two rocprofv3 --pmc children of the same command "
                         "line (adds ~1 min per timed mode); off: report the committed PMC pass (profiles/pmc_traffic.json) or null")
    ap.add_argument("--no-sparse-leg", action="store_true",
                    help="skip the short extra leg with the opt-in sparse MLM head (N = 1, pre-training; reported as roofline.value_sparse_mlm_head_optin)")
    ap.add_argument("--no-batch-curve", action="store_true",
                    help="skip the per-GPU batch sweep 8 ... 512 (N = 1, pre-training workload; ~15 s) reported as roofline.batch_curve")
    ap.add_argument("--selftest-launch", action="store_true", help=argparse.SUPPRESS)
    args = ap.parse_args()

    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        respawn_as_ranks(args.gpus)                       # does not return
    if args.selftest_launch:
        return selftest_launch()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    if os.environ.get("VB_BENCH_HANG_DUMP"):                 # debugging aid: every thread's stack to stderr after N seconds
        import faulthandler
        faulthandler.dump_traceback_later(float(os.environ["VB_BENCH_HANG_DUMP"]), exit=False)
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit("bench.py: --gpus %d but WORLD_SIZE=%d" % (args.gpus, world))

    import torch
    assert torch.cuda.is_available(), "bench.py needs an MI355X"
    # VB_BENCH_ONE_DEVICE=1 (tests/test_bench_launch.py on a 1-GPU box): every rank drives device 0 and the process group is
    # gloo -- RCCL refuses two ranks on one device ("Duplicate GPU detected").  Everything else is the driver's path: the
    # self-spawn, rank environment, gradient hooks, max-over-ranks timing, one JSON line from rank 0.
    one_device = os.environ.get("VB_BENCH_ONE_DEVICE") == "1"
    if one_device:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    import torch.distributed as dist
    use_dist = world > 1 or args.force_dist
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        from visualbert_amd.parallel import configure_rccl_env
        configure_rccl_env()
        if one_device:
            dist.init_process_group(backend="gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group(backend="nccl", device_id=dev, rank=rank, world_size=world)

    from visualbert_amd import ops
    from visualbert_amd.data import synthetic_batch, FeatureStager, pin_batch
    from visualbert_amd.model import AttrDict, ModelWrapper, VisualBERTFixedImageEmbedding
    from visualbert_amd.modeling import BertConfig
    from visualbert_amd.parallel import DataParallelGradSync

    wl = WORKLOADS[args.workload]
    head = wl["head"]
    L, H, I, V, Dv = 12, 768, 3072, 30522, wl["Dv"]
    T, R = args.text_len or wl["T"], args.regions or wl["R"]
    S = T + R
    B = args.batch or wl["batch"]
    dtype = compute_dtype_of(args.dtype)
    torch.manual_seed(1234)
    config = BertConfig(V, hidden_size=H, num_hidden_layers=L, num_attention_heads=H // 64, intermediate_size=I)
    model = VisualBERTFixedImageEmbedding(config=config, training_head_type=head, visual_embedding_dim=Dv,
                                          compute_dtype=dtype).to(dev)
    model.train()
    sync = None
    ranks_seen, comm_kind = 1, None
    if use_dist:
        sync = DataParallelGradSync(model.bert, overlap=not args.no_overlap, use_abi_comm=(args.comm == "abi"))
        sync.broadcast_parameters(0)
        one = torch.ones(1, device=dev)
        dist.all_reduce(one)
        ranks_seen = int(one.item())
        comm_kind = sync.comm_kind
    total_steps = args.steps * 3 + args.warmup + 20
    mw = ModelWrapper(AttrDict(train_batch_size=B * world, learning_rate=5e-5, warmup_proportion=0.1,
                               num_train_epochs=1, gradient_accumulation_steps=1),
                      total_steps * B * world, model=model, grad_sync=sync)
    batch = synthetic_batch(head, B, T, R, Dv, V, seed=rank, device=dev)

    def barrier():
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()

    if args.lib_path:
        from visualbert_amd import _lib
        _lib.set_library(os.path.abspath(args.lib_path))
    if args.dev_debug:
        from visualbert_amd import _lib
        _lib.use_dev_library().vb_gemm_set_debug(args.dev_debug)
    if args.nt_kernel or args.attn_two_pass:
        from visualbert_amd import _lib
        _lib.set_opts(nt_kernel=args.nt_kernel, attn_two_pass=args.attn_two_pass)
    # the matrix pipes' ceiling on THIS box right before and right after the timed loop (boxes differ by +-3.5 %, a kernel gain of
    # 2 % is invisible in `value` alone): roofline.all_gemm_frac_of_measured_ceiling divides by the mean of the two
    ceil_before = None
    if dtype == torch.bfloat16 and not args.no_profile:
        for _ in range(min(args.warmup, 3)):
            mw.step(batch)                                     # clocks and caches in the state the loop will see
        ceil_before = measured_mfma_ceiling(dev)
    elapsed, median_ms, summ = timed_steps(mw, batch, args.steps, args.warmup, barrier, profile=not args.no_profile)
    ceil_after = measured_mfma_ceiling(dev) if ceil_before else None
    et = torch.tensor([elapsed], dtype=torch.float64, device=dev)
    if use_dist:
        dist.all_reduce(et, op=dist.ReduceOp.MAX)
    elapsed = float(et.item())
    loss = float(mw.step(batch)["loss"].detach())

    h2d = None
    if not args.no_h2d:
        stager = FeatureStager(dev)
        host = pin_batch(synthetic_batch(head, B, T, R, Dv, V, seed=rank, device="cpu"))
        nxt, ev = stager.stage(host, 0)
        barrier()
        t1 = time.perf_counter()
        for i in range(args.steps):
            cur, cur_ev = nxt, ev
            nxt, ev = stager.stage(host, (i + 1) & 1)             # prefetch the next batch while this one trains
            torch.cuda.current_stream().wait_event(cur_ev)
            mw.step(cur)
        barrier()
        e2 = torch.tensor([time.perf_counter() - t1], dtype=torch.float64, device=dev)
        if use_dist:
            dist.all_reduce(e2, op=dist.ReduceOp.MAX)
        h2d = B * world * args.steps / float(e2.item())

    allreduce = None
    replicas_same = None
    if sync is not None:
        # replicas must have stayed bit-identical through the synchronised steps: every rank's parameter arena, summed in fp64
        mine = model.bert.arena.data.double().sum().reshape(1)
        sums = [torch.empty_like(mine) for _ in range(world)]
        dist.all_gather(sums, mine)
        replicas_same = all(bool(torch.equal(x, sums[0])) for x in sums)
        allreduce = sync.measure_allreduce(barrier)             # stand-alone all-reduce of the whole gradient arena

    sparse = None
    if (args.sparse_mlm_head or (world == 1 and not args.no_sparse_leg)) and head == "pretraining":
        # the opt-in MLM head over the labelled positions only: the pre-training path never reads the dense fp32
        # logits it materialises (models/model.py:290-297 returns them; ModelWrapper.step uses the loss) -- 20.5 GB of stores per
        # step at B = 1024.  A short extra leg, reported BESIDE `value`, never as `value`: the drop-in contract returns all logits.
        n_sp = args.steps if args.sparse_mlm_head else max(5, args.steps // 4)
        mw.model.bert.sparse_mlm_head = True
        for _ in range(2):
            mw.step(batch)
        barrier()
        t1 = time.perf_counter()
        for _ in range(n_sp):
            mw.step(batch)
        barrier()
        sparse = B * world * n_sp / (time.perf_counter() - t1)
        mw.model.bert.sparse_mlm_head = False

    vendor = None
    if world == 1 and not args.no_vendor_leg and args.nt_kernel == 0 and args.dtype == "bf16":
        # the same step with the PLAIN GEMMs (bias only / "+ addend": 76 of the step's 119) handed to hipBLASLt (nt_kernel 200,
        # csrc/vendor_gemm.hip): what the vendor's hand-scheduled kernel is worth inside the step.  Reported next to `value`,
        # never as `value`: the product path is the hand-written kernels.
        # The yardstick lives in the DEVELOPER library (include/visualbert_hip_dev.h); the product library does not contain it.
        from visualbert_amd import _lib
        vs, tv, nv = {}, None, max(5, args.steps // 5)
        if _lib.dev_lib() is not None:
            with _lib.dev_library():
                _lib.set_opts(nt_kernel=200)
                tv, _, vs = timed_steps(mw, batch, nv, 3, barrier, profile=True)
                tv /= nv
                _lib.set_opts(nt_kernel=0)
        lib_ms = sum(v["ms"] for k, v in vs.items() if k & 512) / nv
        lib_launches = sum(v["launches"] for k, v in vs.items() if k & 512) / nv
        lib_flops = sum(v["flops"] for k, v in vs.items() if k & 512)
        vendor = None if tv is None else dict(value=round(B / tv, 2), unit="samples/s", ms_per_step=round(tv * 1e3, 3), steps=nv,
                      library_launches_per_step=lib_launches, library_ms_per_step=round(lib_ms, 3),
                      library_tflops=round(lib_flops / nv / (lib_ms * 1e-3) / 1e12, 1) if lib_ms > 0 else None,
                      note="plain GEMMs through hipBLASLt (libvisualbert_hip_dev.so, vb_stream_opts.nt_kernel = 200), fused-epilogue GEMMs, weight gradients "
                           "and everything else unchanged; library_launches_per_step = 0 means the library was not found and the "
                           "step ran on our kernels")

    ms_per_step = elapsed / args.steps * 1e3
    value = B * world * args.steps / elapsed
    fps = flops_per_sample(L, H, I, V, S, R, Dv, head)
    peak = PEAK_BF16_TFLOPS if args.dtype in ("bf16", "bf16x3") else PEAK_F32_TFLOPS     # bf16x3 runs on the bf16 matrix pipe

    roofline = par = None
    executed = None
    if rank == 0:
        if summ:
            # what the hardware EXECUTED per step: every GEMM launch's 2 M N K from the per-launch profile (the MLM decoder's dgrad /
            # wgrad run over the labelled rows only -- exact, the skipped gradient rows are identically zero -- so this is below the
            # 3 x forward rule `step_mfu` prices) + the attention core (forward 4 S^2 d per head, backward 2.5 x that)
            gemm_flops = sum(x["flops"] for x in summ.values()) / args.steps
            attn_flops = L * 3.5 * 4.0 * S * S * 64 * (H // 64) * B
            executed = dict(tflop_per_step=round((gemm_flops + attn_flops) / 1e12, 2), gemm_tflop_per_step=round(gemm_flops / 1e12, 2),
                            attention_tflop_per_step=round(attn_flops / 1e12, 2),
                            step_mfu_executed=round((gemm_flops + attn_flops) / (elapsed / args.steps) / (peak * 1e12), 4),
                            note="step_mfu prices SURVEY 8d's 3 x forward FLOPs per sample; this is the arithmetic the step really ran")
            roofline = roofline_of(summ, args.steps, peak, B, args.workload)
            if ceil_before and ceil_after:
                ceil_tf = 0.5 * (ceil_before + ceil_after)
                roofline["mfma_ceiling_measured"] = round(ceil_tf, 1)
                roofline["mfma_ceiling_before_loop"] = round(ceil_before, 1)
                roofline["mfma_ceiling_after_loop"] = round(ceil_after, 1)
                roofline["frac_of_measured_ceiling"] = round(roofline["achieved"] / ceil_tf, 4)
                # the round-to-round, box-independent figure: every GEMM launch of the step against what THIS chip's matrix
                # pipes sustained around the loop
                roofline["all_gemm_frac_of_measured_ceiling"] = round(roofline["all_gemm_tflops"] / ceil_tf, 4)
                roofline["mfma_ceiling_note"] = ("register-only bf16 MFMA loop, operands changing every instruction, all "
                                                 "CUs, run right before and right after the timed loop (mean of the two)")
            if dtype == torch.bfloat16:
                roofline["hbm_bound"] = hbm_bound_kernels(model, B * S, H, dev, optimizer=mw.optimizer,
                                                            V=30522 if head == "pretraining" else 0)
        if not args.no_parity:
            par = parity_side_batch(model, dev, head, T, R, Dv)
    # The collective part of the job ends HERE: the communicator and the process group are torn down before rank 0 starts its
    # long host-side legs (strict-mode models, the CPU baseline), so no rank sits in an RCCL barrier under a watchdog meanwhile.
    if use_dist:
        dist.barrier()
    if sync is not None:
        sync.close()
    if use_dist:
        dist.destroy_process_group()
    if rank == 0:
        cpu = strict = None
        pmc_on = world == 1 and args.pmc_traffic == "auto"
        if pmc_on and roofline is not None:
            # the PMC children re-run this command line: give them the memory first
            del mw, model, batch
            batch = mw = model = None
            torch.cuda.empty_cache()
            child = ["--workload", args.workload, "--dtype", args.dtype, "--batch", str(B)] + \
                    (["--nt-kernel", str(args.nt_kernel)] if args.nt_kernel else []) + PMC_CHILD_FLAGS
            roofline["traffic"], roofline["traffic_source"] = traffic_for(dominant_key(summ), B, args.workload, child)
        if world == 1 and args.strict_dtype != "none" and args.dtype == "bf16":
            batch = mw = model = None
            torch.cuda.empty_cache()
            kinds = ["bf16x3", "fp32"] if args.strict_dtype == "both" else [args.strict_dtype]
            n_strict = args.strict_steps or max(args.steps, 50)
            strict = strict_mode(dev, head, T, R, Dv, V, args.strict_batch, n_strict, args.strict_warmup or args.warmup,
                                 kinds[0], fps, args.workload, pmc=pmc_on)
            for extra in kinds[1:]:                         # the exact fp32 kernels: a short run (they are 3x slower still)
                strict[extra + "_kernels"] = strict_mode(dev, head, T, R, Dv, V, min(args.strict_batch, 256), 6, 2, extra, fps,
                                                         args.workload, full=False)
        curve = None
        if world == 1 and not args.no_batch_curve and args.workload == "pretrain" and roofline is not None:
            batch = mw = model = None
            torch.cuda.empty_cache()
            curve = batch_curve(dev, head, T, R, Dv, V, args.dtype, fps, peak)
            if B not in curve:
                curve[B] = dict(ms_per_step=round(median_ms, 3), samples_per_s=round(B / median_ms * 1e3, 1),
                                step_mfu=round(B / median_ms * 1e3 * fps / (peak * 1e12), 4))
        if roofline is not None:
            # the driver's record keeps the SCALAR entries of `roofline` (nested objects and the other top-level objects of this line
            # survive by name only -- VERDICT r05 item 4): what a reader needs to judge the round goes here, flat
            if par is not None:
                roofline["parity_bf16_max_dlogit"] = par["max_dlogit_vs_fp32_ref"]
            if strict is not None:
                roofline["strict_dtype"] = strict["dtype"]
                roofline["strict_value"] = strict["value"]
                roofline["strict_ms_per_step"] = strict["ms_per_step"]
                roofline["strict_max_dlogit"] = strict["max_dlogit"]
                roofline["strict_meets_tolerance"] = strict["meets_tolerance"]
                if "roofline" in strict:
                    roofline["strict_frac"] = strict["roofline"]["frac"]
                    roofline["strict_all_gemm_tflops"] = strict["roofline"]["all_gemm_tflops"]
                roofline["strict_mode"] = dict(value=strict["value"], ms_per_step=strict["ms_per_step"], max_dlogit=strict["max_dlogit"],
                                               frac=strict.get("roofline", {}).get("frac"))
            if h2d is not None:
                roofline["value_with_h2d"] = round(h2d, 2)
            if sparse is not None:
                roofline["value_sparse_mlm_head_optin"] = round(sparse, 2)
            if curve:
                roofline["batch_curve"] = {str(k): v for k, v in sorted(curve.items())}
                for k, v in sorted(curve.items()):
                    roofline["batch_curve_b%d_ms" % k] = v["ms_per_step"]
        if not args.no_cpu_baseline:
            cpu = cpu_baseline(args.cpu_batch, T, R, head)
        metric = {"pretrain": "pretrain samples/sec (BERT-base, 36 regions+128 tok)",
                  "vqa": "VQA2.0 fine-tune samples/sec (BERT-base, 36 regions+20 tok)",
                  "nlvr2": "NLVR2 fine-tune samples/sec (BERT-base, 2x36 regions+40 tok)",
                  "nlvr2-real": "NLVR2 fine-tune samples/sec (BERT-base, 2x144 regions x 1024-d + 128 tok, the reference's shape)"}[args.workload]
        out = {
            "metric": metric,
            "value": round(value, 2), "unit": "samples/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(ms_per_step, 3), "ms_per_step_median": round(median_ms, 3), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": args.dtype, "data": "synthetic",
            "config": {"workload": "BASELINE.json %s: BERT-base 12L/768 VisualBERT, %d regions x %d-d + %d text tokens (S=%d), %s"
                                   % (wl["cfg"], R, Dv, T, S, wl["what"]),
                       "per_gpu_batch": B, "global_batch": B * world, "seq_len": S, "parallelism": "dp%d" % world,
                       "h2d": "excluded from `value` (batch resident in HBM, the bench contract); roofline.value_with_h2d = the same loop with "
                              "every batch streamed from pinned host memory (SURVEY 8d's definition)",
                       "grad_allreduce": ("fp32 %s (%s), %s" % ("gloo" if one_device else "RCCL", comm_kind, "overlapped with backward" if not args.no_overlap
                                                                  else "after backward")) if use_dist else "none (1 rank)"},
            "value_with_h2d": round(h2d, 2) if h2d is not None else None,
            "train_gflop_per_sample": round(fps / 1e9, 2),
            "step_mfu": round(value * fps / (world * peak * 1e12), 4),
            "step_mfu_executed": executed["step_mfu_executed"] if executed else None,
            "executed": executed,
            "final_loss": round(loss, 4),
            "rccl_ranks_seen": ranks_seen,
            "replicas_bit_identical": replicas_same,
            "allreduce": allreduce,
            "roofline": roofline,
            "cpu_baseline": cpu,
            "parity": par,
            "strict_mode": strict,
            "vendor_plain_gemms": vendor,
        }
        if one_device:
            out["one_device_test_mode"] = "all %d ranks on cuda:0, gloo process group (VB_BENCH_ONE_DEVICE=1)" % world
        if sparse is not None:
            out["samples_per_s_sparse_mlm_head_optin"] = round(sparse, 2)
        try:                                    # RCCL's start-up banner sits in a C stdio buffer: push it out first so that
            import ctypes                       # the JSON line is the LAST line of stdout
            ctypes.CDLL(None).fflush(None)
        except OSError:
            pass
        print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
