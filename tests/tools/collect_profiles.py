"""Runs on the GPU box (through gpurun): collects the rocprofv3 evidence bench.py's roofline object and DESIGN.md refer to and
writes the summaries under gpurun_out/profiles_new/ (copied into profiles/ and committed afterwards).  For the default command
(BASELINE.json configs[1]) and for configs[3] / configs[4] (`--config 4`, `--config 5`):
  1. kernel trace + stats                                     -> <round>_<tag>_kernel_stats.csv, <round>_<tag>_under_rocprof.json
  2. FETCH_SIZE and WRITE_SIZE in two separate --pmc passes   -> <round>_<tag>_pmc_hbm_traffic.csv (every kernel), and for the
     default command <round>_mcl_main_traffic.json
  3. (default command) SQ instruction / busy counters of k_mcl_main -> <round>_mcl_main_pmc_sq.csv
Every rocprofv3 invocation puts the program itself after `--` and never mixes --pmc with other trace domains."""
import csv
import glob
import json
import os
import sqlite3
import subprocess
import sys

ROUND = "r06"
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
OUT = os.path.join(ROOT, "gpurun_out", "profiles_new")
os.makedirs(OUT, exist_ok=True)
ENV = dict(os.environ, TMPDIR="/tmp")


def run(args, tag, bench, script="bench.py"):
    d = os.path.join(ROOT, "gpurun_out", "prof_" + tag)
    subprocess.run(["rm", "-rf", d])
    cmd = ["rocprofv3"] + args + ["-d", d, "-o", "p", "--", "python3", os.path.join(ROOT, script)] + bench
    log = open(os.path.join(OUT, tag + ".log"), "w")
    subprocess.run(cmd, cwd="/tmp", env=ENV, stdout=log, stderr=subprocess.STDOUT, check=False, timeout=400)
    dbs = glob.glob(os.path.join(d, "**", "*.db"), recursive=True)
    print("[collect]", tag, "done", flush=True)
    return dbs[0] if dbs else None, os.path.join(OUT, tag + ".log")


def bench_line(logpath):
    for line in open(logpath, errors="replace"):
        if line.startswith('{"metric"'):
            return json.loads(line)
    return None


def short(name):
    return name.split("(")[0][:70]


def collect(tag, bench, config=None, sq=False):
    cmdline = "python3 bench.py " + " ".join(bench)
    # ---- 1. kernel stats
    db, log = run(["--kernel-trace", "--stats"], tag + "_stats", bench)
    j = bench_line(log)
    if j:
        json.dump(j, open(os.path.join(OUT, f"{ROUND}_{tag}_under_rocprof.json"), "w"))
    avg_ns = {}
    if db:
        con = sqlite3.connect(db)
        rows = list(con.execute("select name, count(*), sum(end-start), avg(end-start), min(end-start), max(end-start) from kernels group by name order by 3 desc"))
        tot = sum(r[2] for r in rows)
        with open(os.path.join(OUT, f"{ROUND}_{tag}_kernel_stats.csv"), "w") as f:
            f.write(f"# rocprofv3 --kernel-trace --stats -- {cmdline}   (summary of the kernel table of the rocpd database)\n")
            w = csv.writer(f)
            w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs"])
            for r in rows:
                w.writerow([r[0], r[1], r[2], "%.1f" % r[3], "%.2f" % (100.0 * r[2] / tot), r[4], r[5]])
                avg_ns[short(r[0])] = r[3]
    # ---- 2. HBM traffic of every kernel: separate passes
    pmc_bench = [a for a in bench]
    for k, v in (("--steps", "40"), ("--warmup", "8")):
        if k in pmc_bench:
            pmc_bench[pmc_bench.index(k) + 1] = v
        else:
            pmc_bench += [k, v]
    per = {}
    for counter in ("FETCH_SIZE", "WRITE_SIZE"):
        db, log = run(["--pmc", counter, "--kernel-trace"], tag + "_" + counter.lower(), pmc_bench)
        if not db:
            continue
        con = sqlite3.connect(db)
        tabs = [r[0] for r in con.execute("select name from sqlite_master where type in ('table','view')")]
        if "counters_collection" not in tabs:
            print("no counters view; tables:", tabs)
            continue
        cols = [d[1] for d in con.execute("pragma table_info(counters_collection)")]
        kcol = "kernel_name" if "kernel_name" in cols else "name"
        q = f"select {kcol}, count(*), avg(value), min(value), max(value) from counters_collection where counter_name='{counter}' group by {kcol} order by 3 desc"
        for name, n, mean, mn, mx in con.execute(q):
            per.setdefault(short(name), {})[counter] = (n, mean, mn, mx)
    with open(os.path.join(OUT, f"{ROUND}_{tag}_pmc_hbm_traffic.csv"), "w") as f:
        f.write(f"# rocprofv3 --pmc FETCH_SIZE (pass 1) / --pmc WRITE_SIZE (pass 2) --kernel-trace -- {'python3 bench.py ' + ' '.join(pmc_bench)} ; KB per dispatch.\n"
                "# hbm_bytes = (2 x FETCH_SIZE + WRITE_SIZE) x 1024: on gfx950 FETCH_SIZE counts 64 B per 128 B request of a wide coalesced read (MI355X_MICROARCH.md, HBM section)\n")
        w = csv.writer(f)
        w.writerow(["kernel", "dispatches", "FETCH_SIZE_mean_KB", "WRITE_SIZE_mean_KB", "hbm_bytes_per_dispatch", "avg_duration_ns_from_kernel_stats", "hbm_GBps"])
        for name, c in sorted(per.items(), key=lambda kv: -(2 * kv[1].get("FETCH_SIZE", (0, 0))[1] + kv[1].get("WRITE_SIZE", (0, 0))[1])):
            fk, wk = c.get("FETCH_SIZE", (0, 0.0))[1], c.get("WRITE_SIZE", (0, 0.0))[1]
            hb = (2.0 * fk + wk) * 1024.0
            ns = avg_ns.get(name)
            w.writerow([name, c.get("FETCH_SIZE", c.get("WRITE_SIZE"))[0], "%.3f" % fk, "%.3f" % wk, "%.0f" % hb, "%.1f" % ns if ns else "", "%.1f" % (hb / ns) if ns else ""])
    if config is not None:
        for name, c in per.items():
            if ("k_mcl_main<0" in name or "k_mcl_mainILi0" in name) and len(c) == 2:
                fk, wk = c["FETCH_SIZE"][1], c["WRITE_SIZE"][1]
                json.dump({"kernel": name, "config": config, "FETCH_SIZE_KB": fk, "WRITE_SIZE_KB": wk,
                           "traffic_bytes_per_launch": (2.0 * fk + wk) * 1024.0, "traffic_bytes_per_launch_uncorrected": (fk + wk) * 1024.0,
                           "note": "MI355X_MICROARCH.md HBM section: on gfx950 FETCH_SIZE counts 64 B per 128 B request for wide coalesced reads, so it is doubled; "
                                   f"WRITE_SIZE is exact. Source: profiles/{ROUND}_{tag}_pmc_hbm_traffic.csv (separate --pmc passes)."},
                          open(os.path.join(OUT, f"{ROUND}_mcl_main_traffic.json"), "w"), indent=1)
    # ---- 3. SQ counters of the dominant kernel
    if sq:
        names = ["SQ_INSTS_VALU", "SQ_ACTIVE_INST_VALU", "SQ_INSTS_SALU", "SQ_INSTS_LDS", "SQ_INSTS_VMEM_RD", "SQ_WAVE_CYCLES", "SQ_BUSY_CYCLES", "SQ_WAIT_INST_LDS"]
        db, log = run(["--pmc"] + names + ["--kernel-trace"], tag + "_sq", pmc_bench)
        if db:
            con = sqlite3.connect(db)
            tabs = [r[0] for r in con.execute("select name from sqlite_master where type in ('table','view')")]
            if "counters_collection" in tabs:
                cols = [d[1] for d in con.execute("pragma table_info(counters_collection)")]
                kcol = "kernel_name" if "kernel_name" in cols else "name"
                with open(os.path.join(OUT, f"{ROUND}_mcl_main_pmc_sq.csv"), "w") as f:
                    f.write("# rocprofv3 --pmc " + " ".join(names) + f" --kernel-trace -- python3 bench.py {' '.join(pmc_bench)} ; mean per dispatch of k_mcl_main\n")
                    w = csv.writer(f)
                    w.writerow(["kernel", "counter", "dispatches", "mean"])
                    for name, cn, n, mean in con.execute(f"select {kcol}, counter_name, count(*), avg(value) from counters_collection where {kcol} like '%k_mcl_main%' group by {kcol}, counter_name"):
                        w.writerow([name.split("(")[0][:40], cn, n, "%.1f" % mean])


def collect_streaming():
    """The HBM-bound kernels one grid / one particle set per launch (tests/tools/stream_kernels_probe.py): duration from the kernel
    trace, HBM bytes from the two counter passes, beside the bytes the kernel has to move."""
    script = os.path.join("tests", "tools", "stream_kernels_probe.py")
    S, N = 4096 * 4096, 1_000_000
    alg = {"k_planner_snapshot": 2 * S, "k_dist_rows_wide": 3 * S, "k_dist_cols_summary": 2 * S, "k_dist_cols_carry": 4 * 32 * 4096 * 4, "k_dist_cols_apply": 4 * S,
           "k_dist_floats": 6 * S, "k_pf_export": 88 * N, "k_pf_encode_lcm": 80 * N, "k_scan_tile_sums": 16 * N, "k_scan_finish_prefix": 24 * N}
    db, _ = run(["--kernel-trace", "--stats"], "stream_stats", [], script)
    dur = {}
    if db:
        for name, n, avg, mn in sqlite3.connect(db).execute("select name, count(*), avg(end-start), min(end-start) from kernels group by name"):
            dur[short(name)] = (n, avg, mn)
    per = {}
    for counter in ("FETCH_SIZE", "WRITE_SIZE"):
        db, _ = run(["--pmc", counter, "--kernel-trace"], "stream_" + counter.lower(), [], script)
        if not db:
            continue
        con = sqlite3.connect(db)
        cols = [d[1] for d in con.execute("pragma table_info(counters_collection)")]
        kcol = "kernel_name" if "kernel_name" in cols else "name"
        for name, n, mean in con.execute(f"select {kcol}, count(*), avg(value) from counters_collection where counter_name='{counter}' group by {kcol}"):
            per.setdefault(short(name), {})[counter] = mean
    with open(os.path.join(OUT, f"{ROUND}_streaming_kernels.csv"), "w") as f:
        f.write("# rocprofv3 --kernel-trace --stats / --pmc FETCH_SIZE / --pmc WRITE_SIZE (three runs) -- python3 tests/tools/stream_kernels_probe.py\n"
                "# 4096 x 4096 grid, 1 000 000 particles, one grid per launch.  hbm_bytes = (2 x FETCH_SIZE + WRITE_SIZE) x 1024 (gfx950 correction, MI355X_MICROARCH.md)\n")
        w = csv.writer(f)
        w.writerow(["kernel", "launches", "avg_ns", "min_ns", "algorithmic_bytes", "algorithmic_GBps_at_avg", "frac_of_8TBps", "hbm_bytes_counters", "counters_over_algorithmic"])
        for k, a in alg.items():
            names = [nm for nm in dur if k in nm]           # (a template instance is "void k<...>")
            if not names:
                continue
            n, avg, mn = dur[names[0]]
            c = per.get(names[0], {})
            hb = (2.0 * c.get("FETCH_SIZE", 0.0) + c.get("WRITE_SIZE", 0.0)) * 1024.0 if len(c) == 2 else None
            w.writerow([k, n, "%.0f" % avg, mn, a, "%.1f" % (a / avg), "%.3f" % (a / avg / 8000.0), "%.0f" % hb if hb else "", "%.3f" % (hb / a) if hb else ""])


def collect_dist_fused():
    """Whole-grid setDistances at 2000^2 and 4096^2 (tests/tools/dist_fused_probe.py: the one-launch form, then the four-launch form):
    durations from the kernel trace, HBM bytes from the two counter passes."""
    script = os.path.join("tests", "tools", "dist_fused_probe.py")
    with open(os.path.join(OUT, f"{ROUND}_dist_fused.csv"), "w") as f:
        f.write("# rocprofv3 --kernel-trace --stats / --pmc FETCH_SIZE / --pmc WRITE_SIZE (three runs per size) -- python3 tests/tools/dist_fused_probe.py <side>\n"
                "# k_dist_fused moves 3 B per cell (int8 in, uint16 out); the four kernels of the other form 3 + 2 + 4 (+ the carries).  hbm_bytes = (2 x FETCH_SIZE + WRITE_SIZE) x 1024\n")
        w = csv.writer(f)
        w.writerow(["side", "kernel", "launches", "avg_ns", "min_ns", "algorithmic_bytes", "algorithmic_GBps_at_avg", "frac_of_8TBps", "hbm_bytes_counters"])
        for side in (2000, 4096):
            S = side * side
            alg = {"k_dist_fused": 3 * S, "k_dist_rows_wide": 3 * S, "k_dist_cols_summary": 2 * S, "k_dist_cols_carry": 0, "k_dist_cols_apply": 4 * S}
            db, _ = run(["--kernel-trace", "--stats"], f"distfused{side}_stats", [str(side)], script)
            dur = {}
            if db:
                for name, n, avg, mn in sqlite3.connect(db).execute("select name, count(*), avg(end-start), min(end-start) from kernels group by name"):
                    dur[short(name)] = (n, avg, mn)
            per = {}
            for counter in ("FETCH_SIZE", "WRITE_SIZE"):
                db, _ = run(["--pmc", counter, "--kernel-trace"], f"distfused{side}_" + counter.lower(), [str(side)], script)
                if not db:
                    continue
                con = sqlite3.connect(db)
                cols = [d[1] for d in con.execute("pragma table_info(counters_collection)")]
                kcol = "kernel_name" if "kernel_name" in cols else "name"
                for name, n, mean in con.execute(f"select {kcol}, count(*), avg(value) from counters_collection where counter_name='{counter}' group by {kcol}"):
                    per.setdefault(short(name), {})[counter] = mean
            for k, a in alg.items():
                for nm in [nm for nm in dur if k in nm]:
                    n, avg, mn = dur[nm]
                    c = per.get(nm, {})
                    hb = (2.0 * c.get("FETCH_SIZE", 0.0) + c.get("WRITE_SIZE", 0.0)) * 1024.0 if len(c) == 2 else None
                    w.writerow([side, nm[:40], n, "%.0f" % avg, mn, a, "%.1f" % (a / avg), "%.3f" % (a / avg / 8000.0), "%.0f" % hb if hb else ""])


which = sys.argv[1:] or ["default", "config4", "config5", "config5explore", "frontiers", "stream", "distfused"]
if "distfused" in which:
    collect_dist_fused()
if "stream" in which:
    collect_streaming()
if "default" in which:
    collect("bench_default", ["--cpu-steps", "0", "--no-other-configs"], config={"particles": 100000, "grid": [200, 200], "rays": 290}, sq=True)
if "config4" in which:
    collect("config4", ["--config", "4", "--goal-l1", "400", "--cpu-steps", "0", "--sub", "--steps", "600", "--warmup", "150"])
if "config5" in which:
    collect("config5_fixed_goal", ["--config", "5", "--fixed-goal", "--goal-l1", "40", "--cpu-steps", "0", "--sub", "--steps", "600", "--warmup", "150"])
if "config5explore" in which:
    collect("config5_explore", ["--config", "5", "--cpu-steps", "0", "--sub", "--steps", "600", "--warmup", "150"])
if "frontiers" in which:
    # find_map_frontiers on the explored-disc worlds (2000^2, 4096^2): kernel durations of the flood and the sweep
    db, _ = run(["--kernel-trace", "--stats"], "frontier_stats", [], os.path.join("tests", "tools", "frontier_flood_probe.py"))
    if db:
        rows = list(sqlite3.connect(db).execute("select name, count(*), sum(end-start), avg(end-start), min(end-start), max(end-start) from kernels group by name order by 3 desc"))
        with open(os.path.join(OUT, f"{ROUND}_frontier_kernels.csv"), "w") as f:
            f.write("# rocprofv3 --kernel-trace --stats -- python3 tests/tools/frontier_flood_probe.py   (3 x find_map_frontiers at 2000^2 and at 4096^2, explored disc)\n")
            w = csv.writer(f)
            w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "MinNs", "MaxNs"])
            for r in rows:
                w.writerow([r[0][:90], r[1], r[2], "%.1f" % r[3], r[4], r[5]])
# the rocpd databases stay on the box: together they exceed what gpurun carries back, and then nothing comes back at all
for d in glob.glob(os.path.join(ROOT, "gpurun_out", "prof_*")):
    subprocess.run(["rm", "-rf", d])
for fn in sorted(os.listdir(OUT)):
    if fn.endswith(".csv") and ROUND in fn:
        print("====", fn)
        print(open(os.path.join(OUT, fn)).read()[:2500])
