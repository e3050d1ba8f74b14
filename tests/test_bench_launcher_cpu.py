"""bench.py --gpus N without a launcher must start N ranks itself, from a parent that never imports torch or touches HIP, and
must fail loudly (non-zero, no JSON line) when the ranks cannot run -- checked here without a GPU: every rank exits with
"bench.py needs a GPU"."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_gpus_flag_launches_ranks_and_propagates_failure():
    env = dict(os.environ, BENCH_TEST_ONE_DEVICE="1")
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--cpu-steps", "0",
                          "--particles", "500"], env=env, cwd=ROOT, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600)
    err = out.stderr.decode()
    import torch
    if torch.cuda.is_available():
        assert out.returncode == 0, err[-2000:]
        return
    assert out.returncode != 0
    assert "bench.py needs a GPU" in err                      # the ranks ran (and said why they stopped)
    assert '"metric"' not in out.stdout.decode()


def test_mismatched_world_size_is_refused():
    env = dict(os.environ, WORLD_SIZE="1", RANK="0", LOCAL_RANK="0")
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "4", "--steps", "2"], env=env, cwd=ROOT,
                         stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=300)
    assert out.returncode != 0 and b"WORLD_SIZE=1" in out.stderr


def test_launcher_parent_does_not_import_torch():
    """What runs before a child is started -- the module level, self_launch and run_other_configs -- must not import torch or the
    library (either would initialise HIP in the parent)."""
    import ast
    src = open(os.path.join(ROOT, "bench.py")).read()
    tree = ast.parse(src)

    def imports(node):
        names = []
        for n in ast.walk(node):
            if isinstance(n, ast.Import):
                names += [a.name for a in n.names]
            elif isinstance(n, ast.ImportFrom):
                names.append(n.module or "")
        return names

    top = [n for n in tree.body if not isinstance(n, (ast.FunctionDef, ast.ClassDef))]
    for node in top:
        assert not any(m.startswith(("torch", "botlab_amd")) for m in imports(node)), ast.dump(node)[:200]
    funcs = {n.name: n for n in tree.body if isinstance(n, ast.FunctionDef)}
    for name in ("self_launch", "run_other_configs"):
        assert not any(m.startswith(("torch", "botlab_amd")) for m in imports(funcs[name])), name
    main_src = ast.get_source_segment(src, funcs["main"])
    assert main_src.index("run_other_configs(") < main_src.index("import torch") and main_src.index("self_launch(") < main_src.index("import torch")


def test_watchdog_ends_a_hung_rank_group_and_names_the_phase():
    """A multi-rank run that stops making progress (first contact with several devices: a hang inside a collective or a peer
    mapping) is ended by the launcher parent: the child process group is killed, the exit status is non-zero, the phase is named."""
    env = dict(os.environ, BENCH_TEST_ONE_DEVICE="1", BENCH_TEST_HANG="1", BENCH_WATCHDOG_SCALE="0.05")      # 'start' may last 12 s
    import time
    t0 = time.time()
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--cpu-steps", "0",
                          "--particles", "500"], env=env, cwd=ROOT, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=300)
    err = out.stderr.decode()
    assert out.returncode == 3, err[-1500:]
    assert "in phase 'start'" in err and "ending its process group" in err
    assert time.time() - t0 < 120
    assert '"metric"' not in out.stdout.decode()
