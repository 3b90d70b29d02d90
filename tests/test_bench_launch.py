"""`python bench.py --gpus N` must work as typed: without WORLD_SIZE it re-executes itself as N ranks through
torch.distributed.run on 127.0.0.1 and rank 0 prints the one JSON line.  Driven here on CPU through the plumbing-only
mode (gloo group, an all-reduce of ones) -- the GPU path behind it is the same launcher."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(args, extra_env=None):
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update(extra_env or {})
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, env=env, cwd=ROOT, timeout=300,
                         stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
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
