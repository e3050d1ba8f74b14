"""The line bench.py ends stdout with must be readable by the driver: ONE compact JSON object, serialised length < 4 096 bytes, carrying the
contract's keys, `roofline` and `cpu_baseline` -- whatever the detail record holds (round 5's 21.7 KB line was not read).  The detail record
(other_configs, astar_fixtures, ...) goes to gpurun_out/bench_detail.json and stderr."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import bench  # noqa: E402

CONTRACT_KEYS = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype",
                 "data", "config", "roofline", "cpu_baseline")


def _records():
    # full records of earlier rounds (the 21.7 KB one among them) as inputs of the compaction
    for name in ("r05_bench_driver_form.json", "r04_bench_driver_form.json", "r05_bench_default.json"):
        yield name, json.load(open(os.path.join(ROOT, "profiles", name)))


def test_compact_line_is_short_and_complete():
    for name, full in _records():
        full["summary"] = bench.summary_of(full)
        full["detail"] = "gpurun_out/bench_detail.json"
        c = bench.compact_line(full)
        line = json.dumps(c)
        assert len(line) < bench.COMPACT_LIMIT, (name, len(line))
        back = json.loads(line)
        for k in CONTRACT_KEYS:
            assert k in back, (name, k)
        r = back["roofline"]
        for k in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
            assert k in r, (name, k)
        assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-5
        b = back["cpu_baseline"]
        for k in ("value", "unit", "cores", "kind", "sample"):
            assert k in b, (name, k)
        assert "workload" in back["config"] and "model" not in back["config"]
        assert back["value"] == round(full["value"], 6) and back["steps"] == full["steps"]


def test_compact_line_survives_a_bloated_record():
    _, full = next(_records())
    full["config"]["workload"] = "w" * 5000
    full["cpu_baseline"]["sample"] = "s" * 5000
    full["config"]["collective"] = "c" * 5000
    full["summary"] = {f"k{i}": "x" * 50 for i in range(200)}           # an optional block that cannot fit is dropped, never truncated mid-JSON
    full["particle_sweep_steps_per_s"] = {str(i): float(i) for i in range(500)}
    full["detail"] = "d" * 100
    line = json.dumps(bench.compact_line(full))
    assert len(line) < bench.COMPACT_LIMIT
    back = json.loads(line)
    assert "roofline" in back and "cpu_baseline" in back and "summary" not in back


def test_summary_lifts_the_reviewed_numbers():
    full = json.load(open(os.path.join(ROOT, "profiles", "r05_bench_driver_form.json")))
    s = bench.summary_of(full)
    assert s["closed_loop_ms"] > 0 and s["config4_goal400_closed_loop_steps_s"] > 0 and s["config5_as_written_steps_s"] > 0
    assert s["astar_maze_success_mean_us"]["hip"] > 0 and s["astar_maze_success_mean_us"]["cpu_1_core"] > 0


def test_stdout_carries_exactly_one_write():
    # the only writes to the saved stdout descriptor are the phase lines of a launcher child and emit()'s one line
    src = open(os.path.join(ROOT, "bench.py")).read()
    assert src.count("os.write(json_fd") == 3                            # phase(), --astar-fixtures, emit()
