"""`python bench.py --gpus N` must work as typed: without WORLD_SIZE it re-executes itself as N ranks through
torch.distributed.run on 127.0.0.1 and rank 0 prints the one JSON line.  Driven here on CPU through the plumbing-only
mode (gloo group, an all-reduce of ones) -- the GPU path behind it is the same launcher."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(args, extra_env=None, timeout=300):
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update(extra_env or {})
    try:
        out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, env=env, cwd=ROOT, timeout=timeout,
                             stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
    except subprocess.TimeoutExpired as e:                  # keep what the ranks printed (VB_BENCH_HANG_DUMP stacks) for the log
        tail = (e.stderr or b"")
        tail = tail.decode("utf-8", "replace") if isinstance(tail, bytes) else tail
        try:
            os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
            with open(os.path.join(ROOT, "gpurun_out", "bench_launch_timeout_stderr.log"), "w") as f:
                f.write(tail)
        except OSError:
            pass
        raise AssertionError("bench.py %s timed out after %ds; stderr tail:\n%s" % (" ".join(args), timeout, tail[-3000:]))
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout                      # ONE JSON line, from rank 0 only
    return json.loads(lines[0])


def test_bench_spawns_its_own_ranks():
    d = _run(["--gpus", "2", "--selftest-launch"])
    assert d["n_gpus"] == 2 and d["ranks_seen"] == 2


def test_bench_runs_as_a_rank_of_an_existing_launch():
    """the driver's form: python -m torch.distributed.run ... bench.py --gpus N (WORLD_SIZE already set: no re-exec)."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
                          "127.0.0.1", "--master-port", "29531", os.path.join(ROOT, "bench.py"), "--gpus", "2",
                          "--selftest-launch"], env=env, cwd=ROOT, timeout=300, stdout=subprocess.PIPE,
                         stderr=subprocess.PIPE, text=True)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1 and json.loads(lines[0])["ranks_seen"] == 2


@pytest.mark.gpu
@pytest.mark.parametrize("world,batch,extra", [(2, 8, []), (8, 2, []), (8, 2, ["--no-overlap"])])
def test_driver_command_ranks_end_to_end(dev, world, batch, extra):
    """The driver's multi-GPU command, `python bench.py --gpus N ...` (N = 2, and N = 8 = the node north_star names, with and
    without --no-overlap), executed for real on the one GPU this box has: all
    ranks drive cuda:0 (VB_BENCH_ONE_DEVICE=1; the process group is gloo because RCCL refuses two ranks on one device).
    Everything but the transport is the N > 1 path the scaling bench runs: self-spawn through torch.distributed.run, rank
    environment, per-rank shards, gradient hooks firing during backward, max-over-ranks timing, ONE JSON line from rank 0
    that still carries cpu_baseline and parity."""
    if dev.type != "cuda":
        pytest.skip("needs the GPU")
    d = _run(["--gpus", str(world), "--batch", str(batch), "--steps", "2", "--warmup", "1", "--cpu-batch", "1", "--no-h2d"] + extra,
             extra_env={"VB_BENCH_ONE_DEVICE": "1", "VB_BENCH_HANG_DUMP": "300"}, timeout=240 if world == 2 else 600)
    assert d["n_gpus"] == world and d["rccl_ranks_seen"] == world
    assert d["config"]["global_batch"] == world * batch and d["config"]["parallelism"] == "dp%d" % world
    assert ("after backward" if extra else "overlapped with backward") in d["config"]["grad_allreduce"]
    assert d["allreduce"] and d["allreduce"]["ranks"] == world and d["allreduce"]["bus_GBps"] > 0
    assert d["allreduce"]["buckets"] == 12 + 2                      # heads | 12 layers | embeddings
    assert d["replicas_bit_identical"] is True                      # parameter arenas of all ranks after the timed steps
    assert d["cpu_baseline"] and d["cpu_baseline"]["value"] > 0
    assert d["parity"] and d["parity"]["max_dlogit_vs_fp32_ref"] < 0.1
    assert d["value"] > 0 and d["final_loss"] == d["final_loss"]
    assert 5.0 < d["final_loss"] < 15.0                             # ~ln(30522) + ln 2 after three steps from random weights


def test_pmc_kernel_name_matcher():
    """bench.py's in-run PMC traffic leg picks the dominant GEMM family out of rocprofv3's kernel names, which arrive mangled for
    some instantiations and (mis)demangled for others (the demangler trips over __bf16): both spellings, output type and the
    split-operand flag must be told apart (names as rocprofv3 printed them on MI355X, profiles/r04_final_kernel_stats_*.txt)."""
    sys.path.insert(0, ROOT)
    import bench
    mangled_dual_bf16 = "_ZN12_GLOBAL__N_119gemm_nt_dual_kernelIDF16bLi0ELi1ELi3ELb0EEEvNS_8GemmArgsE"
    demangled_dual_bf16 = "gemm_nt_dual_kernel<bool _Accum, int, ELi4E, 3, false>(GemmArgs)"
    dual_f32 = "gemm_nt_dual_kernel<float, 0, 2, 3, false>(GemmArgs)"
    dual_x3 = "gemm_nt_dual_kernel<float, 4, 8, 3, true>(GemmArgs)"
    ph8_bf16 = "_ZN12_GLOBAL__N_118gemm_nt_8ph_kernelIDF16bDF16bLi0ELi0ELi1ELb0EEEvNS_8GemmArgsE"
    ph8_x3 = "gemm_nt_8ph_kernel<bool _Accum, 0, 0, 1, true>(GemmArgs)"
    ph8_x3_mangled = "_ZN12_GLOBAL__N_118gemm_nt_8ph_kernelIDF16bfLi0ELi0ELi1ELb1EEEvNS_8GemmArgsE"
    names = [mangled_dual_bf16, demangled_dual_bf16, dual_f32, dual_x3, ph8_bf16, ph8_x3, ph8_x3_mangled, "gemm_tn_8ph_kernel(TnArgs)"]
    pick = lambda key: [n for n in names if bench.kernel_name_filter(key)(n)]
    assert pick(64) == [mangled_dual_bf16, demangled_dual_bf16]            # the bf16 headline's dominant family
    assert pick(64 | 4) == [dual_f32]                                      # its fp32-logit instantiation
    assert pick(64 | 4 | 256) == [dual_x3]
    assert pick(16) == [ph8_bf16]
    assert pick(16 | 4 | 256) == [ph8_x3, ph8_x3_mangled]                  # the strict leg's dominant family
    assert bench.kernel_name_filter(1) is None and bench.kernel_name_filter(19) is None    # not an NT GEMM family


def test_roofline_reports_the_dominant_kernel_and_the_comparable_family():
    """roofline.frac is defined on the kernel with the largest summed launch time; the dispatch rule moves shapes between the two
    K-contiguous kernels from round to round, so the object also carries every K-contiguous GEMM of the dominant kernel's output type
    together (`nt_gemms_same_output_type`) -- checked here on a synthetic per-kernel summary (no GPU)."""
    sys.path.insert(0, ROOT)
    from unittest import mock
    import bench
    steps = 20
    summ = {16: dict(ms=33.6 * steps, flops=1141.0e12 * 33.6e-3 * steps, launches=74 * steps),          # persistent kernel, bf16 -> bf16
            64: dict(ms=24.3 * steps, flops=790.0e12 * 24.3e-3 * steps, launches=25 * steps),           # two-workgroup kernel, bf16 -> bf16
            64 | 4: dict(ms=10.0 * steps, flops=887.0e12 * 10.0e-3 * steps, launches=2 * steps),        # fp32 logits: another output type
            19: dict(ms=23.2 * steps, flops=1283.0e12 * 23.2e-3 * steps, launches=16 * steps)}          # grouped weight gradients
    with mock.patch.object(bench, "traffic_for", lambda *a, **k: (None, "not measured")):
        r = bench.roofline_of(summ, steps, 2500.0, 1024, "pretraining")
    assert r["kernel"].startswith("gemm_nt_8ph_kernel<bf16->bf16") and abs(r["frac"] - 0.4564) < 1e-3
    fam = r["nt_gemms_same_output_type"]
    assert fam["launches_per_step"] == 99 and abs(fam["ms_per_step"] - 57.9) < 1e-6
    assert abs(fam["tflops"] - (1141.0 * 33.6 + 790.0 * 24.3) / 57.9) < 0.1 and abs(fam["frac"] - fam["tflops"] / 2500.0) < 1e-4
    with mock.patch.object(bench, "traffic_for", lambda *a, **k: (None, "not measured")):
        r = bench.roofline_of({19: summ[19]}, steps, 2500.0, 1024, "pretraining")          # no NT GEMM at all: the family is the kernel itself
    assert r["nt_gemms_same_output_type"]["launches_per_step"] == 16
