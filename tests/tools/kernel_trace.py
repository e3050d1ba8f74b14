"""Runs on the GPU box: rocprofv3 --kernel-trace --stats of `python3 bench.py <args>` and prints the per-kernel table
(name, calls, total / avg / min / max ns) from the rocpd database.  Usage: python3 tests/tools/kernel_trace.py <bench args...>"""
import glob, json, os, sqlite3, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
d = "/tmp/ktrace"
subprocess.run(["rm", "-rf", d])
cmd = ["rocprofv3", "--kernel-trace", "--stats", "-d", d, "-o", "p", "--", "python3", os.path.join(ROOT, "bench.py")] + sys.argv[1:]
r = subprocess.run(cmd, cwd="/tmp", env=dict(os.environ, TMPDIR="/tmp"), stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
for line in r.stdout.decode(errors="replace").splitlines():
    if line.startswith('{"metric"'):
        j = json.loads(line)
        print("bench:", round(j["value"], 1), j["unit"], j["stage_ms"], "pops/step", j.get("astar_pops_per_step"))
dbs = glob.glob(os.path.join(d, "**", "*.db"), recursive=True)
if not dbs:
    print(r.stdout.decode(errors="replace")[-2000:])
    raise SystemExit("no rocpd database")
con = sqlite3.connect(dbs[0])
rows = list(con.execute("select name, count(*), sum(end-start), avg(end-start), min(end-start), max(end-start) from kernels group by name order by 3 desc"))
for row in rows[:16]:
    print("%-60s calls %6d total %12d avg %10.0f min %9d max %10d" % (row[0][:60], row[1], row[2], row[3], row[4], row[5]))
