"""How often does a map update change the set of cells with log-odds >= 0 (the only thing the distance transform reads)?
Runs the bench's config-4 inputs for a number of steps and counts, per step, the cells whose sign class changed."""
import os, sys, types
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
import botlab_amd as bl
import bench
grid_side = int(os.environ.get("GRID", "2000")); steps = int(os.environ.get("STEPS", "300"))
args = types.SimpleNamespace(grid=grid_side, max_range=8.0)
ctx = bl.default_context()
m, truth, poses, odo, scans, rands = bench.build_inputs(args, steps, ctx)
cpm = np.float32(1.0 / np.float64(np.float32(0.05)))
g = bl.OccupancyGrid.from_cells(m["cells"], m["origin"], m["mpc"], cellsPerMeter=cpm, ctx=ctx)
mapper = bl.Mapping(5.0, 4, 1, ctx=ctx)
prev = g.cells() >= 0
flips = []
for k in range(steps):
    sc = scans[k]
    mapper.updateMap(sc, bl.make_pose(*poses[k + 1], utime=sc.utime), g)     # true poses: the question is about the map, not the filter
    cur = g.cells() >= 0
    flips.append(int((cur != prev).sum()))
    prev = cur
f = np.array(flips)
print("steps", steps, "with a sign-class change:", int((f > 0).sum()), "cells changed per step: mean %.1f max %d" % (f.mean(), f.max()))
print("first 40:", flips[:40]); print("last 40:", flips[-40:])
