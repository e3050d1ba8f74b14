"""bench.py's N > 1 path (torch.distributed.run, one process per rank, sharded particle filter, barrier + max-over-ranks
timing, one JSON line from rank 0) exercised end to end on a one-GPU box: both ranks on cuda:0, collectives over gloo.
The numbers mean nothing here; the run must complete and the sharded result must equal the single-rank result."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _cmd(nproc, launcher):
    if launcher:
        # the command the driver's contract quotes for N > 1
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={nproc}", "--master-addr", "127.0.0.1",
               "--master-port", str(29700 + os.getpid() % 200), os.path.join(ROOT, "bench.py")]
    else:
        cmd = [sys.executable, os.path.join(ROOT, "bench.py")]          # bench.py starts its own ranks when --gpus > 1
    return cmd + ["--gpus", str(nproc), "--steps", "12", "--warmup", "3", "--cpu-steps", "0", "--particles", "6000"]


def _run(nproc, extra_env, launcher=False):
    env = dict(os.environ, **extra_env)
    out = subprocess.run(_cmd(nproc, launcher), env=env, cwd=ROOT, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600)
    assert out.returncode == 0, out.stderr.decode()[-3000:]
    lines = [l for l in out.stdout.decode().splitlines() if l.startswith('{"metric"')]
    assert len(lines) == 1, out.stdout.decode()[-2000:]                # exactly one JSON line, from rank 0
    return json.loads(lines[0])


def test_bench_two_ranks_on_one_device():
    one = _run(1, {})
    two = _run(2, {"BENCH_TEST_ONE_DEVICE": "1"})                      # plain `python bench.py --gpus 2`: it launches its ranks
    assert two["n_gpus"] == 2 and two["steps"] == 12 and two["scaling"] == "strong" and two["value"] > 0
    assert two["config"]["parallelism"] == "particle-shard x2"
    for key in ("roofline", "metric", "unit", "ms_per_step", "higher_is_better", "vs_baseline", "dtype", "data"):
        assert key in two
    # Philox noise is keyed by the global particle index and the estimate is formed from the gathered record: the pose after
    # the same steps is the same for one and two ranks up to the last bit of the fused / record-based finish
    assert all(abs(a - b) <= 1e-5 * max(1.0, abs(b)) for a, b in zip(two["final_pose"], one["final_pose"]))


def test_bench_two_ranks_under_the_contract_launcher():
    two = _run(2, {"BENCH_TEST_ONE_DEVICE": "1"}, launcher=True)
    assert two["n_gpus"] == 2 and two["value"] > 0


def test_bench_more_ranks_than_gpus_fails_loudly():
    # a one-GPU box cannot run --gpus 2 for real: non-zero exit and no result line (never a silent n_gpus: 1)
    env = {k: v for k, v in os.environ.items() if k != "BENCH_TEST_ONE_DEVICE"}
    out = subprocess.run(_cmd(2, False), env=env, cwd=ROOT, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600)
    import torch
    if torch.cuda.device_count() >= 2:
        pytest.skip("this box has two GPUs")
    assert out.returncode != 0
    assert not [l for l in out.stdout.decode().splitlines() if l.startswith('{"metric"')]
    assert b"has no GPU" in out.stderr
