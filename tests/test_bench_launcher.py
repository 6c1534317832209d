"""CPU: `bench.py --gpus N` without a launcher around it starts its N ranks itself (one process per GPU, rendezvous on 127.0.0.1,
rank 0 prints the one JSON line), and refuses a WORLD_SIZE that contradicts --gpus.  The ranks drive the CPU harness of the kernels
(--emu, gloo) because this container has no GPU: what is under test is bench.py's N > 1 branch -- spawn, env_index_base /
seed_stride per rank, barrier, MAX all-reduce of the wall time, whole-job value."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(args, env=None):
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, capture_output=True, text=True, timeout=900,
                          env=dict(os.environ, **(env or {})))


def test_bench_gpus_2_launches_its_own_ranks():
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--emu", "--gpus", "2", "--envs", "12", "--steps", "3", "--warmup", "1"],
                       capture_output=True, text=True, timeout=900, env=env)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout  # ONE line, from rank 0
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["steps"] == 3 and out["warmup"] == 1 and out["scaling"] == "weak"
    assert out["config"]["envs_per_gpu"] == 12 and out["config"]["parallelism"].startswith("batch shard x2")
    # whole-job value: both ranks' slots over the max-over-ranks time
    assert abs(out["value"] - 2 * 12 * 3 / (out["ms_per_step"] * 3e-3)) < 1e-6 * out["value"]


def test_bench_refuses_a_world_size_that_contradicts_gpus():
    r = _run(["--emu", "--gpus", "2", "--envs", "4", "--steps", "1", "--warmup", "0"], env=dict(WORLD_SIZE="1", RANK="0", LOCAL_RANK="0"))
    assert r.returncode != 0 and "WORLD_SIZE" in r.stderr
