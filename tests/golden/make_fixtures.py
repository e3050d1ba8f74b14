#!/usr/bin/env python3
"""Regenerates tests/golden/reference_maps.npz and astar_cases.json from the DATA files the reference ships and its
own tests read (data/*.map, data/astar/*.map, data/astar/*_poses.txt; src/planning/astar_test.cpp:160-195 names
them).  Runs only in the build container (needs /root/reference); the outputs are committed.

.map format (src/slam/occupancy_grid.cpp:111-175): header `origin_x origin_y width height metersPerCell`, then
`height` rows of `width` signed integers.  Header floats are stored as float32, exactly what `in >> float` yields.
"""
import json
import os
import sys

import numpy as np

REF = os.environ.get("BOTLAB_REFERENCE", "/root/reference")
HERE = os.path.dirname(os.path.abspath(__file__))


def load_map(path):
    with open(path) as f:
        tok = f.read().split()
    ox, oy = np.float32(tok[0]), np.float32(tok[1])
    w, h = int(tok[2]), int(tok[3])
    mpc = np.float32(tok[4])
    cells = np.array(tok[5:5 + w * h], dtype=np.int64)
    assert cells.size == w * h, path
    assert cells.min() >= -128 and cells.max() <= 127
    return dict(origin=np.array([ox, oy], np.float32), mpc=np.array([mpc], np.float32),
                cells=cells.astype(np.int8).reshape(h, w))


def main():
    out = {}
    names = []
    for sub in ("data", "data/astar"):
        d = os.path.join(REF, sub)
        for fn in sorted(os.listdir(d)):
            if fn.endswith(".map"):
                key = (sub.replace("data", "").strip("/") + "_" + fn[:-4]).strip("_")
                m = load_map(os.path.join(d, fn))
                out[key + "__cells"] = m["cells"]
                out[key + "__origin"] = m["origin"]
                out[key + "__mpc"] = m["mpc"]
                names.append(key)
    np.savez_compressed(os.path.join(HERE, "reference_maps.npz"), **out)

    cases = {}
    for name in ("empty", "filled", "narrow", "wide", "convex", "maze"):
        with open(os.path.join(REF, "data/astar", name + "_poses.txt")) as f:
            tok = f.read().split()
        n = int(tok[0])
        rows = []
        for i in range(n):
            # token-stream semantics of `poseIn >> start.x >> start.y >> goal.x >> goal.y >> shouldExist`
            # (astar_test.cpp:236): line breaks are irrelevant and a failed extraction at EOF leaves 0
            # (convex_poses.txt has a 4-token line, so its last case is read across the line break).
            five = (tok[1 + 5 * i: 6 + 5 * i] + ["0"] * 5)[:5]
            sx, sy, gx, gy, ex = five
            rows.append(dict(start=[float(sx), float(sy)], goal=[float(gx), float(gy)], should_exist=bool(int(ex))))
        cases[name] = rows
    with open(os.path.join(HERE, "astar_cases.json"), "w") as f:
        json.dump(cases, f, indent=1)
    print("maps:", names)
    print("astar cases:", {k: len(v) for k, v in cases.items()})


if __name__ == "__main__":
    sys.exit(main())
