"""us per pop of single searches (time of bl_astar_search on the host, launch and fetch included) for the split-storage kernel
(k_astar2: LDS loop, deep loop, C++ forms) and for round 4's k_astar (BOTLAB_ASTAR_V1=1), with the stamped build's cycle shares of
the straight-line loop -> gpurun_out/profiles_new/r06_astar_pop.csv.  Searches: the four fixture searches of astar_probe.py (open
lists of ~1e3, ~1.2e4, ~2.1e4 and ~2.4e5 entries) and convex case 2 (1.8e6 pops)."""
import os, re, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
OUT = os.path.join(ROOT, "gpurun_out", "profiles_new")
os.makedirs(OUT, exist_ok=True)
probe = os.path.join(ROOT, "tests", "tools", "astar_probe.py")


def run(env):
    e = dict(os.environ); e.update(env)
    r = subprocess.run([sys.executable, probe], env=e, capture_output=True, text=True, timeout=600)
    rows, stamps = [], []
    for line in (r.stdout + r.stderr).splitlines():
        m = re.match(r"(\S+) (\d+) pops (\d+) pushes (\d+) len (\d+) ([\d.]+) ms\s+([\d.]+) us/pop", line)
        if m:
            rows.append((m.group(1) + " case " + m.group(2), int(m.group(3)), int(m.group(4)), float(m.group(6)), float(m.group(7))))
        m = re.search(r"cycles/pop: all (\d+) = issue (-?\d+) \+ adjust (\d+) \+ loadwait (\d+) \+ expand (\d+) \+ pushes (\d+) \| clock ([\d.]+)", line)
        if m:
            stamps.append(tuple(m.groups()))
    return rows, stamps


v2, _ = run({})
d1, _ = run({"BOTLAB_ASTAR_DEEP_AHEAD": "0"})
duo, _ = run({"BOTLAB_ASTAR_AHEAD": "0"})
one, _ = run({"BOTLAB_ASTAR_DUO": "0"})
v1, _ = run({"BOTLAB_ASTAR_V1": "1"})
cpp, _ = run({"BOTLAB_ASTAR_NO_TURBO": "1", "BOTLAB_ASTAR_DUO": "0"})
_, st2 = run({"STAMPS": "1", "BOTLAB_ASTAR_DUO": "0"})
with open(os.path.join(OUT, "r06_astar_pop.csv"), "w") as f:
    f.write("# python3 tests/tools/astar_pop_table.py (tests/tools/astar_probe.py per column; best of 3 host-timed calls of bl_astar_search)\n"
            "# k_astar2 = split-storage open list with the straight-line loops on three wavefronts -- pops with the next walk taken beside the\n"
            "# pushes / pushes / expansions made ahead (bl_astar2_ahead.h, round 6) -- in the LDS regime and beyond it; deep_one_wave =\n"
            "# BOTLAB_ASTAR_DEEP_AHEAD=0: one wave beyond LDS (bl_astar2_deep.h, the default until this round's last change);\n"
            "# duo = BOTLAB_ASTAR_AHEAD=0 (round 5's two-wave loop, bl_astar2_duo.h); one_wave = BOTLAB_ASTAR_DUO=0 (bl_astar2_turbo.h: what the replanner's units run); cpp = the same kernel\n"
            "# with BOTLAB_ASTAR_NO_TURBO=1 (C++ forms only); k_astar = round 4's kernel (BOTLAB_ASTAR_V1=1).  stamped = cycles per pop of the\n"
            "# ONE-wave loop in the -DBL_ASTAR_STAMPS build (every mark drains the LDS queue: shares, not the undisturbed loop's time)\n")
    f.write("search,pops,pushes,k_astar2_ms,k_astar2_us_per_pop,deep_one_wave_us_per_pop,duo_r05_us_per_pop,one_wave_us_per_pop,cpp_us_per_pop,k_astar_r04_us_per_pop,speedup_vs_r04,stamped_all,stamped_top_and_loads,stamped_pop,stamped_loadwait,stamped_expand,stamped_pushes,clock_GHz\n")
    for i, r in enumerate(v2):
        a = v1[i] if i < len(v1) else None
        c = cpp[i] if i < len(cpp) else None
        o = one[i] if i < len(one) else None
        du = duo[i] if i < len(duo) else None
        dd = d1[i] if i < len(d1) else None
        s = st2[3 * i + 2] if 3 * i + 2 < len(st2) and r[1] <= 20000 else ("",) * 7      # (the deep loop carries no marks)
        f.write("%s,%d,%d,%.2f,%.3f,%s,%s,%s,%s,%s,%s,%s\n" % (r[0], r[1], r[2], r[3], r[4], "%.3f" % dd[4] if dd else "", "%.3f" % du[4] if du else "", "%.3f" % o[4] if o else "", "%.3f" % c[4] if c else "", "%.3f" % a[4] if a else "",
                                                    "%.2f" % (a[4] / r[4]) if a else "", ",".join(s)))
print(open(os.path.join(OUT, "r06_astar_pop.csv")).read())
