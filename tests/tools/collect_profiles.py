"""Runs on the GPU box (through gpurun): collects the rocprofv3 evidence bench.py's roofline object refers to and writes
the summaries under gpurun_out/profiles_new/ (copied into profiles/ and committed afterwards).
  1. kernel trace + stats of the default bench command  -> r02_bench_default_kernel_stats.csv, r02_bench_default.json
  2. FETCH_SIZE and WRITE_SIZE in two separate --pmc passes -> r02_pmc_hbm_traffic.csv, r02_mcl_main_traffic.json
  3. SQ instruction / busy counters for k_mcl_main        -> r02_mcl_main_pmc_sq.csv
Every rocprofv3 invocation puts the program itself after `--` and never mixes --pmc with other trace domains."""
import csv
import glob
import json
import os
import sqlite3
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
OUT = os.path.join(ROOT, "gpurun_out", "profiles_new")
os.makedirs(OUT, exist_ok=True)
ENV = dict(os.environ, TMPDIR="/tmp")


def run(args, tag):
    d = os.path.join(ROOT, "gpurun_out", "prof_" + tag)
    subprocess.run(["rm", "-rf", d])
    cmd = ["rocprofv3"] + args + ["-d", d, "-o", "p", "--", "python3", os.path.join(ROOT, "bench.py")] + BENCH
    log = open(os.path.join(OUT, tag + ".log"), "w")
    subprocess.run(cmd, cwd="/tmp", env=ENV, stdout=log, stderr=subprocess.STDOUT, check=False)
    dbs = glob.glob(os.path.join(d, "**", "*.db"), recursive=True)
    return dbs[0] if dbs else None, os.path.join(OUT, tag + ".log")


def bench_line(logpath):
    for line in open(logpath, errors="replace"):
        if line.startswith('{"metric"'):
            return json.loads(line)
    return None


# ---- 1. kernel stats of the default command
BENCH = ["--cpu-steps", "0", "--no-other-configs"]
db, log = run(["--kernel-trace", "--stats"], "stats")
j = bench_line(log)
if j:
    json.dump(j, open(os.path.join(OUT, "r02_bench_default_under_rocprof.json"), "w"))
con = sqlite3.connect(db)
rows = list(con.execute("select name, count(*), sum(end-start), avg(end-start), min(end-start), max(end-start) from kernels group by name order by 3 desc"))
tot = sum(r[2] for r in rows)
with open(os.path.join(OUT, "r02_bench_default_kernel_stats.csv"), "w") as f:
    f.write("# rocprofv3 --kernel-trace --stats -- python3 bench.py --cpu-steps 0   (summary of the kernel table of the rocpd database)\n")
    w = csv.writer(f)
    w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs"])
    for r in rows:
        w.writerow([r[0], r[1], r[2], "%.1f" % r[3], "%.2f" % (100.0 * r[2] / tot), r[4], r[5]])

# ---- 2. HBM traffic: separate passes
BENCH = ["--cpu-steps", "0", "--no-other-configs", "--steps", "40", "--warmup", "5"]
traffic = {}
lines = []
for counter in ("FETCH_SIZE", "WRITE_SIZE"):
    db, log = run(["--pmc", counter, "--kernel-trace"], counter.lower())
    if not db:
        continue
    con = sqlite3.connect(db)
    tabs = [r[0] for r in con.execute("select name from sqlite_master where type in ('table','view')")]
    view = "counters_collection" if "counters_collection" in tabs else None
    if not view:
        print("no counters view; tables:", tabs)
        continue
    cols = [d[1] for d in con.execute(f"pragma table_info({view})")]
    kcol = "kernel_name" if "kernel_name" in cols else "name"
    q = f"select {kcol}, count(*), avg(value), min(value), max(value) from {view} where counter_name='{counter}' group by {kcol} order by 3 desc"
    for name, n, mean, mn, mx in con.execute(q):
        lines.append([name.split("(")[0][:60], counter, n, "%.3f" % mean, "%.3f" % mn, "%.3f" % mx])
        if "k_mcl_main<0" in name or "k_mcl_mainILi0" in name:
            traffic[counter] = (name, mean)
with open(os.path.join(OUT, "r02_pmc_hbm_traffic.csv"), "w") as f:
    f.write("# rocprofv3 --pmc FETCH_SIZE (pass 1) / --pmc WRITE_SIZE (pass 2) --kernel-trace -- python3 bench.py --cpu-steps 0 --steps 40 --warmup 5 ; values in KB per dispatch\n")
    w = csv.writer(f)
    w.writerow(["kernel", "counter", "dispatches", "mean_KB", "min_KB", "max_KB"])
    w.writerows(lines)
if len(traffic) == 2:
    fk, wk = traffic["FETCH_SIZE"][1], traffic["WRITE_SIZE"][1]
    json.dump({"kernel": traffic["FETCH_SIZE"][0].split("(")[0], "config": {"particles": 100000, "grid": [200, 200], "rays": 290},
               "FETCH_SIZE_KB": fk, "WRITE_SIZE_KB": wk,
               "traffic_bytes_per_launch": (2.0 * fk + wk) * 1024.0,
               "traffic_bytes_per_launch_uncorrected": (fk + wk) * 1024.0,
               "note": "MI355X_MICROARCH.md HBM section: on gfx950 FETCH_SIZE counts 64 B per 128 B request for wide coalesced reads, so it is doubled; "
                       "WRITE_SIZE is exact. Source: profiles/r02_pmc_hbm_traffic.csv (separate --pmc passes)."},
              open(os.path.join(OUT, "r02_mcl_main_traffic.json"), "w"), indent=1)

# ---- 3. SQ counters of the dominant kernel
sq = ["SQ_INSTS_VALU", "SQ_ACTIVE_INST_VALU", "SQ_INSTS_SALU", "SQ_INSTS_LDS", "SQ_INSTS_VMEM_RD", "SQ_WAVE_CYCLES", "SQ_BUSY_CYCLES", "SQ_WAIT_INST_LDS"]
db, log = run(["--pmc"] + sq + ["--kernel-trace"], "sq")
if db:
    con = sqlite3.connect(db)
    tabs = [r[0] for r in con.execute("select name from sqlite_master where type in ('table','view')")]
    if "counters_collection" in tabs:
        cols = [d[1] for d in con.execute("pragma table_info(counters_collection)")]
        kcol = "kernel_name" if "kernel_name" in cols else "name"
        with open(os.path.join(OUT, "r02_mcl_main_pmc_sq.csv"), "w") as f:
            f.write("# rocprofv3 --pmc " + " ".join(sq) + " --kernel-trace -- python3 bench.py --cpu-steps 0 --steps 40 --warmup 5 ; mean per dispatch of k_mcl_main\n")
            w = csv.writer(f)
            w.writerow(["kernel", "counter", "dispatches", "mean"])
            for name, cn, n, mean in con.execute(f"select {kcol}, counter_name, count(*), avg(value) from counters_collection where {kcol} like '%k_mcl_main%' group by {kcol}, counter_name"):
                w.writerow([name.split("(")[0][:40], cn, n, "%.1f" % mean])
print(open(os.path.join(OUT, "r02_bench_default_kernel_stats.csv")).read()[:1500])
for fn in ("r02_pmc_hbm_traffic.csv", "r02_mcl_main_traffic.json", "r02_mcl_main_pmc_sq.csv"):
    p = os.path.join(OUT, fn)
    print("====", fn); print(open(p).read()[:1500] if os.path.exists(p) else "MISSING")
