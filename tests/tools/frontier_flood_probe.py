"""Timing probe of find_map_frontiers on the large explored-disc worlds only (flood / sweep stamps: BOTLAB_FRONTIER_STAMPS=1)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import botlab_amd as bl, helpers

ctx = bl.default_context()


def explored_world(S, radius_cells):
    cells = np.full((S, S), -100, np.int8)
    for oy in range(20, S, 40):
        for ox in range(20, S, 40):
            cells[oy:oy + 4, ox:ox + 4] = 100
    cells[0, :] = cells[-1, :] = 100; cells[:, 0] = cells[:, -1] = 100
    yy, xx = np.mgrid[0:S, 0:S]
    cx = cy = S // 2 + 6
    far = (xx - cx) ** 2 + (yy - cy) ** 2 > radius_cells ** 2
    cells[far] = 0
    half = S * 0.05 / 2
    return cells, (np.float32(-half), np.float32(-half)), np.float32(0.05), (-half + (cx + radius_cells - 40 + 0.5) * 0.05, -half + (cy + 0.5) * 0.05, 0.0)


for S, r in ((2000, 400), (4096, 1500)):
    cells, origin, mpc, robot = explored_world(S, r)
    grid = bl.OccupancyGrid.from_cells(cells, origin, mpc, cellsPerMeter=helpers.CPM_DEFAULT, ctx=ctx)
    rp = bl.make_pose(*robot)
    for rep in range(3):
        ctx.sync(); t0 = time.perf_counter(); fr = bl.find_map_frontiers(grid, rp); t_find = time.perf_counter() - t0
    print(f"tiled{S}: find {t_find * 1e3:.2f} ms, bfs {fr.stats()}", flush=True)
