#!/usr/bin/env python3
"""bench.py -- SLAM steps/sec (map + MCL + A* replan) on MI355X.

Workload (BASELINE.json configs[1], the configuration the metric is quoted on): full SLAM on the 200x200 @5 cm
obstacle_slam map with 100k particles.  The reference's .log inputs are absent from its checkout, so scans and
odometry are synthesised over the shipped .map (botlab_amd/synth.py) -- "data": "synthetic".

One step = OccupancyGridSLAM::runSLAMIteration (src/slam/slam.cpp:191-207) + the planner's per-map work
(src/planning/exploration.cpp:300-317):
    ParticleFilter::updateFilter(odometry, scan, map)      pre-update map
    Mapping::updateMap(scan, pose, map)
    ObstacleDistanceGrid::setDistances(map)  +  search_for_path(pose, goal)     on a second stream, on a snapshot
Multi-GPU (--gpus N under torch.distributed.run): the particles are block-sharded over the ranks (RCCL all-gather of
the 16-byte exchange record, the only collective); the map update and the replan are replicated.
Total work is fixed as N grows ("scaling": "strong").

Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0            # MI355X_MICROARCH.md: HBM3E 8 TB/s
EVENT_STRIDE = 7                 # every 7th k_mcl_main launch of the timed region carries start/stop events (a timed launch costs ~4 us of stream
                                 # time); 7 shares no factor with the replanner batches (4, 8): 8 sampled the launches a lane's burst of
                                 # distance transforms runs beside, every time (0.075 ms against 0.066 in the trace of all launches)
EVENT_STRIDE_LONG = 17           # ... runs of more than 400 steps: the events of every 7th launch cost 1 % of the steps/s (10 860 against 10 965 without any)
EVENT_STRIDE_SHORT_RUN = 1       # ... runs of at most 50 steps (the driver's 20): every launch, or the roofline leg rests on three samples
EXPLORE_EVERY = 5                # one exploration step per published map = every 5th SLAM step (slam.cpp:284-289, exploration.cpp:300-317)


def load_map(name):
    z = np.load(os.path.join(ROOT, "tests", "golden", "reference_maps.npz"))
    return dict(cells=z[name + "__cells"], origin=tuple(z[name + "__origin"]), mpc=z[name + "__mpc"][0])


def build_inputs(args, total_steps, ctx=None):
    from botlab_amd import synth
    m = load_map(args.map)
    if getattr(args, "explore", False):
        # The exploration loop's world: the shipped obstacle_slam arena in every 200-cell tile, sealed (whatever the shipped map does
        # not know to be free is solid), so that the explored region stays an arena of the reference's own size: with the reference's
        # cost function plan_path_to_frontier costs ~100 pops per explored free cell (1e6 on this arena; an open hall of 7.5 m
        # radius overflows a 1.2e7-pop open list -- tests/test_gpu_exploring_map_large.py).  The map starts with the centre
        # tile's arena known up to column 110 and everything else unknown.
        arena = load_map("obstacle_slam_10mx10m_5cm")["cells"]
        G = args.grid
        reps = (G + 199) // 200
        truth = np.tile(np.where(arena < 0, -127, 127).astype(np.int8), (reps, reps))[:G, :G].copy()
        truth[0, :] = truth[-1, :] = 127
        truth[:, 0] = truth[:, -1] = 127
        t0 = (G // 2 // 200) * 200
        cells = np.zeros((G, G), np.int8)
        a = arena.copy()
        a[:, 110:] = 0
        cells[t0:t0 + 200, t0:t0 + 200] = a
        half = G * 0.05 / 2.0
        origin = (np.float32(-half), np.float32(-half))
        off = (float(origin[0]) + t0 * 0.05 + 5.0, float(origin[1]) + t0 * 0.05 + 5.0)
        m = dict(cells=cells, origin=origin, mpc=np.float32(0.05))
        start = (-0.75 + off[0], 0.2 + off[1], 0.0)
        rands = np.random.default_rng(99).integers(0, 2**31 - 1, size=total_steps + 4)
        poses, odo, scans = record_exploration(args, total_steps, ctx, m, truth, start, rands)
        return m, truth, poses, odo, scans, rands
    if args.grid != 200:
        cells = synth.tile_world(load_map("astar_maze")["cells"], args.grid)
        half = args.grid * 0.05 / 2.0
        m = dict(cells=np.where(cells > 0, 127, -100).astype(np.int8), origin=(np.float32(-half), np.float32(-half)),
                 mpc=np.float32(0.05))
        start, side = (0.3, 0.3, 0.0), 0.5
    elif args.map == "convex_10mx10m_5cm":
        # BASELINE.json configs[2]: localization on the shipped convex room (3.5 m across, centred on the origin)
        start, side = (-0.4, -0.4, 0.0), 0.8
    else:
        # a 0.8 m square loop that keeps >= 0.2 m clearance inside the mapped arena of obstacle_slam (searched offline)
        start, side = (-0.75, 0.2, 0.0), 0.8
    if getattr(args, "start", None) is not None and args.grid == 200:
        start, side = tuple(args.start), 0.6
    truth = np.where(m["cells"] > 0, 127, -127).astype(np.int8)
    rng = np.random.default_rng(1234)
    poses = synth.square_trajectory(start, total_steps, step_len=0.02, turn=0.05, side=side)
    odo = synth.odometry_from_truth(poses, rng)
    max_range = getattr(args, "max_range", synth.MAX_RANGE)
    if args.grid != 200 and ctx is not None:
        # large worlds: every beam of every scan marched in ONE launch of the simulator's lidar kernel (bl_sim_cast_beams,
        # SURVEY.md section 8 row f4) -- the numpy ray caster needs minutes for a few hundred scans of a 4096 x 4096 world
        scans = synth.raycast_scans_gpu(truth, m["origin"], float(m["mpc"]), poses, 1_000_000, 100_000, ctx, max_range=max_range,
                                        noise_sigma=0.005, rng=rng)
    else:
        scans = []
        for k in range(1, len(poses)):
            scans.append(synth.raycast_scan(truth, m["origin"], float(m["mpc"]), poses[k - 1], poses[k], 1_000_000 + k * 100_000,
                                            max_range=max_range, noise_sigma=0.005, rng=rng))
    rands = np.random.default_rng(99).integers(0, 2**31 - 1, size=total_steps + 4)
    return m, truth, poses, odo, scans, rands


def record_exploration(args, total_steps, ctx, m, truth, start, rands):
    """The inputs of an exploration run are not known in advance: the robot drives where plan_path_to_frontier sends it, and that
    depends on the map the SLAM loop has built so far.  So the loop is run ONCE here, untimed and synchronously -- scan cast from
    the truth world at the robot's pose, filter update, map update, on every 5th step the exploration step, the robot then
    following currentPath_ at 0.02 m (or 0.05 rad) per step as the reference's motion controller would -- and its scans and
    odometry are recorded.  The timed run replays them through the pipelined loop: the filter and the map are deterministic, so
    the exploration steps meet the same maps and poses and take the same decisions, at the rate a real exploration takes them."""
    import botlab_amd as bl
    from botlab_amd import synth
    cpm = np.float32(1.0 / np.float64(np.float32(0.05)))
    rng = np.random.default_rng(1234)
    origin, mpc = m["origin"], float(m["mpc"])
    max_range = getattr(args, "max_range", synth.MAX_RANGE)
    grid = bl.OccupancyGrid.from_cells(m["cells"], origin, m["mpc"], cellsPerMeter=cpm, ctx=ctx)
    mapper = bl.Mapping(5.0, 4, 1, ctx=ctx)
    pf = bl.ParticleFilter(args.particles, ctx=ctx)
    ex = bl.AsyncExplorer(ctx=ctx, lanes=1, robotRadius=0.2)
    pose = np.array(start, dtype=np.float64)
    poses, odo, scans = [pose.copy()], [pose.copy()], []
    path, wp = [], 0
    plans = 0
    t_rec = time.perf_counter()
    for k in range(total_steps):
        # ---- the robot's motion of this step: along the path, turning in place towards the next waypoint first
        nxt = pose.copy()
        while wp < len(path) and np.hypot(path[wp][0] - pose[0], path[wp][1] - pose[1]) < 0.02:
            wp += 1
        if wp < len(path):
            dx, dy = path[wp][0] - pose[0], path[wp][1] - pose[1]
            want = np.arctan2(dy, dx)
            dth = np.arctan2(np.sin(want - pose[2]), np.cos(want - pose[2]))
            if abs(dth) > 0.05:
                nxt[2] = pose[2] + np.sign(dth) * 0.05
            else:
                step = min(0.02, float(np.hypot(dx, dy)))
                nxt[:] = (pose[0] + step * np.cos(want), pose[1] + step * np.sin(want), want)
        else:
            nxt[2] = pose[2] + 0.05                           # no path (yet, or any more): look around
        nxt[2] = np.arctan2(np.sin(nxt[2]), np.cos(nxt[2]))
        sc = synth.raycast_scan(truth, origin, mpc, pose, nxt, 1_000_000 + (k + 1) * 100_000, max_range=max_range, noise_sigma=0.005, rng=rng)
        # odometry: the truth motion + drift, accumulated in the odometry frame (synth.odometry_from_truth, one step)
        d = nxt - pose
        dist_ = float(np.hypot(d[0], d[1]))
        head = np.arctan2(d[1], d[0]) - pose[2] if dist_ > 1e-12 else 0.0
        dth_ = np.arctan2(np.sin(d[2]), np.cos(d[2]))
        dist_n = dist_ + (rng.normal(0, 1e-3) if dist_ > 0 else 0.0)
        dth_n = dth_ + rng.normal(0, 3e-3)
        po = odo[-1]
        odo.append(np.array([po[0] + dist_n * np.cos(po[2] + head), po[1] + dist_n * np.sin(po[2] + head),
                             np.arctan2(np.sin(po[2] + dth_n), np.cos(po[2] + dth_n))]))
        poses.append(nxt.copy())
        scans.append(sc)
        pose = nxt
        # ---- the SLAM step and, on every 5th, the exploration step
        if k == 0:
            pf.initializeFilterAtPose(bl.make_pose(*odo[0], utime=int(scans[0].times[0])), seed=42)
        oo = odo[-1]
        est = pf.updateFilter(bl.make_pose(oo[0], oo[1], oo[2], utime=sc.utime), sc, grid, rand_value=int(rands[k]))
        mapper.updateMap(sc, est, grid)
        if k % EXPLORE_EVERY == 0:
            ex.submit(grid, pf.poseDevicePtr())
            r, pth = ex.fetch()
            if r.planned and r.path_length > 1:
                path, wp = [(q.x, q.y) for q in pth], 1
                plans += 1
    sys.stderr.write(f"[bench] exploration recorded: {total_steps} steps, {plans} plans, {time.perf_counter() - t_rec:.1f} s\n")
    ex.close(); pf.close(); mapper.close(); grid.close()
    return poses, odo, scans


def pick_goal(dist_cells, origin, start_xy, radius, max_l1_cells):
    """A reachable-looking replan goal: the free cell with clearance > radius + 1 cell that is farthest (L1) from the
    start but no farther than max_l1_cells (SURVEY.md section 8d picks the farthest such cell)."""
    h, w = dist_cells.shape
    sx = int((start_xy[0] - origin[0]) * 20.0)
    sy = int((start_xy[1] - origin[1]) * 20.0)
    ys, xs = np.nonzero(dist_cells > radius + 0.05)
    l1 = np.abs(xs - sx) + np.abs(ys - sy)
    ok = l1 <= max_l1_cells
    if not ok.any():
        return None
    k = np.argmax(np.where(ok, l1, -1))
    return (float(origin[0]) + (xs[k] + 0.5) * 0.05, float(origin[1]) + (ys[k] + 0.5) * 0.05)


def cpu_baseline(args, m, odo, scans, rands, goal, n_steps):
    """The CPU oracle (a serial restatement of the reference path, oracle/botlab_oracle.cpp) timed on ONE host core on
    a bounded sample of the same workload."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import helpers
    import oracle_lib
    orc = oracle_lib.load_oracle()
    cpm = helpers.CPM_DEFAULT
    cells = m["cells"].copy()
    N = args.particles
    pf = oracle_lib.OraclePF(orc, N)
    pf.init_at_pose(orc.pose(*odo[0], utime=int(scans[0].times[0])), 42)
    om = oracle_lib.OracleMapping(orc, 5.0, 4, 1)
    # one untimed step latches odometry / previous pose (first calls are no-ops in the reference)
    o = odo[1]
    res = pf.update(orc.pose(*o, utime=scans[0].utime), scans[0], cells, m["mpc"], cpm, m["origin"], int(rands[0]))
    om.update(scans[0], res["pose"], cells, m["mpc"], cpm, m["origin"])
    t0 = time.perf_counter()
    pops, plans, t_plan = 0, 0, 0.0
    for k in range(1, 1 + n_steps):
        o = odo[k + 1]
        res = pf.update(orc.pose(*o, utime=scans[k].utime), scans[k], cells, m["mpc"], cpm, m["origin"], int(rands[k]))
        om.update(scans[k], res["pose"], cells, m["mpc"], cpm, m["origin"])
        if goal is not None:
            d = orc.set_distances(cells, m["mpc"], cpm, m["origin"])
            tp = time.perf_counter()
            _, st = orc.search(res["pose"], orc.pose(goal[0], goal[1], 0.0), d, m["mpc"], cpm, m["origin"], 0.2, 2.0)
            t_plan += time.perf_counter() - tp
            pops += st[0]
        elif getattr(args, "explore", False) and k % 5 == 0:
            # the exploration step on every 5th map (slam.cpp:285-289 publishes it, exploration.cpp:277-369 plans on it)
            tp = time.perf_counter()
            d = orc.set_distances(cells, m["mpc"], cpm, m["origin"])
            fr = orc.find_frontiers(cells, m["mpc"], cpm, m["origin"], res["pose"])
            if fr:
                _, _, st = orc.plan_path_to_frontier(fr, res["pose"], d, m["mpc"], cpm, m["origin"], 0.2, 0.2, 2.0)
                pops += st[0]
                plans += 1
            t_plan += time.perf_counter() - tp
    dt = time.perf_counter() - t0
    # (the searches of the sampled steps are those of the trajectory's first poses: the HIP row's astar_pops_per_step is its timed
    # region's average, which lies elsewhere on the trajectory -- the per-pop figure is what carries over)
    what = f"A* {pops // max(n_steps, 1)} pops/step" + (f", {1e3 * t_plan / n_steps:.2f} ms per search on this core (its grid-sized set-up included)" if pops else "")
    if getattr(args, "explore", False):
        what = f"an exploration step on every 5th map: {plans} plans to a frontier, {pops} pops, {t_plan:.2f} s of the {dt:.2f} s"
    extra = {}
    if goal is not None and pops:
        extra = dict(astar_pops_per_step=pops / n_steps, astar_ms_per_search=round(1e3 * t_plan / n_steps, 3), non_planner_s_per_step=round((dt - t_plan) / n_steps, 4))
    return dict(value=n_steps / dt, unit="steps/s", cores=1, kind="port", **extra,
                sample=f"{n_steps} full steps of the same workload ({N} particles, {scans[0].num_ranges} rays, "
                       f"{cells.shape[1]}x{cells.shape[0]} grid, {what}), oracle on 1 thread, "
                       f"{os.cpu_count()} host cores visible")


def same_search(ctx, grid, planner, pose, goal_pose, m):
    """The planner leg on IDENTICAL inputs for both sides: the map and the pose estimate as the timed region left them, the run's goal.
    The oracle's sampled steps lie at the start of the trajectory, the HIP rows average over all of it -- the searches differ in size;
    this one is the same search: pops must be equal, the two times are one search's latency (HIP: setDistances was done, the call is
    bl_astar_search through the C ABI, best of three; CPU: search_for_path of the oracle on one core, its distance grid made before)."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import helpers
    import oracle_lib
    import botlab_amd as bl
    orc = oracle_lib.load_oracle()
    cells = grid.cells().copy()
    planner.distances_.forget()
    planner.setMap(grid)
    start = bl.make_pose(pose.x, pose.y, pose.theta)
    best, st_gpu = None, None
    for _ in range(3):
        t0 = time.perf_counter()
        _, st_gpu = bl.search_for_path(start, goal_pose, planner.distances_, planner.searchParams_, return_stats=True)
        dt = time.perf_counter() - t0
        best = dt if best is None or dt < best else best
    d = orc.set_distances(cells, m["mpc"], helpers.CPM_DEFAULT, m["origin"])
    t0 = time.perf_counter()
    _, st_cpu = orc.search(orc.pose(pose.x, pose.y, pose.theta), orc.pose(goal_pose.x, goal_pose.y, 0.0), d, m["mpc"], helpers.CPM_DEFAULT, m["origin"], 0.2, 2.0)
    t_cpu = time.perf_counter() - t0
    return {"pops": int(st_gpu[0]), "pops_equal": bool(int(st_gpu[0]) == int(st_cpu[0]) and int(st_gpu[1]) == int(st_cpu[1])),
            "hip_ms": round(1e3 * best, 3), "cpu_oracle_ms": round(1e3 * t_cpu, 3), "cores": 1,
            "inputs": "map and pose estimate at the end of the timed region, the run's goal"}


def same_search_fixture(ctx, name="maze", case=0, radius=0.1):
    """The same comparison on a search that IS a search: the reference's own fixture (data/astar/maze_poses.txt pair 0 on maze.map,
    robotRadius 0.1 as astar_test.cpp:227-228 sets it: 1 156 pops).  The headline's replan to a point on the driven loop is a handful of
    pops -- its two times are launch latency, not a search."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import helpers
    import oracle_lib
    import botlab_amd as bl
    orc = oracle_lib.load_oracle()
    cpm = helpers.CPM_DEFAULT
    m = load_map("astar_" + name)
    row = helpers.load_astar_cases()[name][case]
    g = bl.OccupancyGrid.from_cells(m["cells"], m["origin"], m["mpc"], cellsPerMeter=cpm, ctx=ctx)
    planner = bl.MotionPlanner(bl.MotionPlannerParams(radius), ctx=ctx)
    planner.setMap(g)
    s, gl = bl.make_pose(row["start"][0], row["start"][1], 0.0), bl.make_pose(row["goal"][0], row["goal"][1], 0.0)
    best, st_gpu = None, None
    for _ in range(5):
        ctx.sync()
        t0 = time.perf_counter()
        _, st_gpu = bl.search_for_path(s, gl, planner.distances_, planner.searchParams_, return_stats=True)
        dt = time.perf_counter() - t0
        best = dt if best is None or dt < best else best
    d = orc.set_distances(m["cells"], m["mpc"], cpm, m["origin"])
    t_cpu = None
    for _ in range(3):
        t0 = time.perf_counter()
        _, st_cpu = orc.search(orc.pose(*row["start"], 0.0), orc.pose(*row["goal"], 0.0), d, m["mpc"], cpm, m["origin"], radius, 10.0 * radius, cap=4096)
        dt = time.perf_counter() - t0
        t_cpu = dt if t_cpu is None or dt < t_cpu else t_cpu
    return {"pops": int(st_gpu[0]), "pops_equal": bool(int(st_gpu[0]) == int(st_cpu[0]) and int(st_gpu[1]) == int(st_cpu[1])),
            "hip_ms": round(1e3 * best, 3), "cpu_oracle_ms": round(1e3 * t_cpu, 3), "cores": 1,
            "inputs": f"astar fixture {name} pair {case} (tests/golden), robotRadius {radius}; best of 5 / 3 calls"}


ASTAR_FIXTURE_MAPS = ["empty", "filled", "narrow", "wide", "convex", "maze"]
ASTAR_FIXTURE_EXCLUDED = {("narrow", 2): "the search must exhaust the whole free side (2.6e8 pops in the reference's own algorithm): excluded on both sides"}


def astar_fixtures(ctx, reps=5, long_pops=100_000, cpu=True):
    """The reference's own published table: astar_test (src/planning/astar_test.cpp:196-332, 407-426) runs planner.planPath on the
    start / goal pairs of data/astar/*_poses.txt for its six maps (robotRadius 0.1, :227-228), files the wall time of every call
    under 'successful' (path_length > 1) or 'failed' planning attempts per map and prints mean / median / std in microseconds
    (report/saptadeb-botlab.tex:172 quotes them).  The same table for the HIP path (MotionPlanner.planPath through the C ABI: the
    validity gather + k_astar2 + the path's way back to the host, `reps` calls per pair -- one for searches of more than
    `long_pops` pops) and for the CPU oracle on one host core, with the pops behind every row.  The median is the exact one
    (astar_test's boost accumulator estimates it)."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import helpers
    import botlab_amd as bl
    orc = None
    if cpu:
        import oracle_lib
        orc = oracle_lib.load_oracle()
    cpm = helpers.CPM_DEFAULT
    cases = helpers.load_astar_cases()

    def stats(v):
        if not v:
            return None
        a = np.asarray(v, dtype=np.float64)
        return {"n": int(a.size), "mean_us": round(float(a.mean()), 1), "median_us": round(float(np.median(a)), 1), "std_us": round(float(a.std()), 1)}

    rows = {}
    for name in ASTAR_FIXTURE_MAPS:
        m = load_map("astar_" + name)
        g = bl.OccupancyGrid.from_cells(m["cells"], m["origin"], m["mpc"], cellsPerMeter=cpm, ctx=ctx)
        planner = bl.MotionPlanner(bl.MotionPlannerParams(0.1), ctx=ctx)
        planner.setMap(g)
        dist = orc.set_distances(m["cells"], m["mpc"], cpm, m["origin"]) if orc else None
        gs, gf, cs, cf, pops, excluded = [], [], [], [], 0, []
        for i, row in enumerate(cases[name]):
            if (name, i) in ASTAR_FIXTURE_EXCLUDED:
                excluded.append({"case": i, "why": ASTAR_FIXTURE_EXCLUDED[(name, i)]})
                continue
            s, gl = bl.make_pose(row["start"][0], row["start"][1], 0.0), bl.make_pose(row["goal"][0], row["goal"][1], 0.0)
            n_rep = reps
            for r in range(reps):
                if r >= n_rep:
                    break
                ctx.sync()
                t0 = time.perf_counter()
                path, st = planner.planPath(s, gl, return_stats=True)
                dt = (time.perf_counter() - t0) * 1e6
                (gs if len(path) > 1 else gf).append(dt)
                if r == 0:
                    pops += st[0]
                    if st[0] > long_pops:
                        n_rep = 1
            if orc:
                os_, og = orc.pose(*row["start"], 0.0), orc.pose(*row["goal"], 0.0)
                t0 = time.perf_counter()
                if orc.is_valid_goal(og, dist, m["mpc"], cpm, m["origin"], 0.1, 0.1):
                    exp, est = orc.search(os_, og, dist, m["mpc"], cpm, m["origin"], 0.1, 1.0)
                    ok = len(exp) > 1
                else:
                    ok = False
                (cs if ok else cf).append((time.perf_counter() - t0) * 1e6)
        rows[name] = {"pairs": len(cases[name]) - len(excluded), "pops": int(pops), "hip": {"success": stats(gs), "failed": stats(gf)},
                      "cpu_oracle_1_core": {"success": stats(cs), "failed": stats(cf)} if orc else None}
        if excluded:
            rows[name]["excluded"] = excluded
    return {"source": "data/astar fixtures (tests/golden), astar_test.cpp:306-332, 407-426", "reps_per_pair": reps, "rows": rows}


PHASE_LIMIT_S = {                  # how long a multi-rank run may stay in one phase before the launcher ends it (first contact with
    "start": 240.0,                # several devices: a hang in a collective or in a peer mapping must end the run with the phase's name)
    "process-group": 180.0, "ipc-probe": 120.0, "engine": 180.0, "inputs": 600.0, "filter-init": 240.0, "planner-setup": 180.0,
    "warmup": 300.0, "timed": 600.0, "stage-pass": 300.0, "report": 300.0, "teardown": 120.0,
}


def phase(json_fd, name):
    """One line per phase on the launcher's pipe (rank 0 only): the watchdog of self_launch reads them."""
    if int(os.environ.get("RANK", "0")) == 0 and os.environ.get("BENCH_PHASES"):
        os.write(json_fd, f"[bench-phase] {name}\n".encode())


def self_launch(n):
    """`python bench.py --gpus N` without a launcher: run N ranks under torch.distributed.run on this node (the same command the
    contract quotes) as a CHILD process group, pass rank 0's single JSON line through, return non-zero if any rank failed or no
    line came out.  A watchdog: rank 0 reports its phases; a phase that outlasts its limit gets the whole child group killed (the
    group this launcher started, by its id) and the launcher exits non-zero with the phase's name -- never a re-exec."""
    import selectors
    import signal
    import socket
    import subprocess
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", "1")
    env["BENCH_PHASES"] = "1"
    proc = subprocess.Popen(cmd, stdout=subprocess.PIPE, env=env, start_new_session=True)
    sel = selectors.DefaultSelector()
    sel.register(proc.stdout, selectors.EVENT_READ)
    os.set_blocking(proc.stdout.fileno(), False)
    line, cur, t_phase, buf = None, "start", time.monotonic(), b""
    scale = float(os.environ.get("BENCH_WATCHDOG_SCALE", "1"))
    while True:
        ready = sel.select(timeout=1.0)
        if ready:
            chunk = proc.stdout.read()
            if chunk:
                buf += chunk
                while b"\n" in buf:
                    raw, buf = buf.split(b"\n", 1)
                    txt = raw.decode(errors="replace").strip()
                    if txt.startswith("[bench-phase] "):
                        cur, t_phase = txt[len("[bench-phase] "):], time.monotonic()
                        print(f"[bench] {n} ranks: {cur}", file=sys.stderr)
                    elif txt.startswith("{") and '"metric"' in txt:
                        line = txt
                    elif txt:
                        print(txt, file=sys.stderr)
            elif proc.poll() is not None:
                break
        elif proc.poll() is not None:
            break
        if time.monotonic() - t_phase > scale * PHASE_LIMIT_S.get(cur, 300.0):
            print(f"bench.py: the {n}-rank run spent more than {scale * PHASE_LIMIT_S.get(cur, 300.0):.0f} s in phase '{cur}': ending its process group",
                  file=sys.stderr)
            try:
                os.killpg(proc.pid, signal.SIGTERM)
                try:
                    proc.wait(timeout=15)
                except subprocess.TimeoutExpired:
                    os.killpg(proc.pid, signal.SIGKILL)
            except ProcessLookupError:
                pass
            proc.wait()
            return 3
    rc = proc.wait()
    if rc != 0:
        print(f"bench.py: the {n}-rank launch failed (exit {rc}) in phase '{cur}'", file=sys.stderr)
        return rc
    if line is None:
        print("bench.py: the ranks finished without a result line", file=sys.stderr)
        return 1
    print(line, flush=True)
    return 0


OTHER_CONFIGS = [                                      # (preset, goal_l1 in cells, extra flags): short runs printed beside the headline
    (0, 0, ["--no-astar"]),                            # updateFilter + updateMap alone from the default start, and ...
    (0, 0, ["--no-astar", "--start", "0", "0", "0"]),  # ... from the reference's own start pose (slam.cpp:64-66), where the float sums of
                                                       # estimatePosteriorPose hover around zero: what that costs a step.
                                                       # (no replan: from there the reference's search to a point 0.4 m ahead takes 9e4 pops)
    (3, 0, []), (4, 40, []), (4, 400, []),                 # preset 3 (1M particles on one GPU) has no replan: its goal is unused
    (5, 0, []),                                            # BASELINE.json configs[4] as written: SLAM + the exploration step on every 5th map
    (5, 0, ["--explore-mode", "newest-map"]),              # ... with the reference's process arrangement: SLAM never waits for the explorer
    (5, 40, ["--fixed-goal"]),                             # ... and its fixed-goal form (SLAM + a replan per step), as in earlier rounds
    (0, 0, ["--particles", "1000"]), (0, 0, ["--particles", "10000"]),          # north_star's particle sweep below the headline's 100k
    # closed-loop latency: every step's pose and path fetched before the next step is enqueued (slam.cpp:191-207 is synchronous per scan)
    (0, 0, ["--depth", "0", "--lanes", "1", "--batch", "1", "--sync-steps", "200"]),
    (4, 400, ["--depth", "0", "--lanes", "1", "--batch", "1", "--sync-steps", "200"]),
    (4, 40, ["--depth", "0", "--lanes", "1", "--batch", "1", "--sync-steps", "200"]),      # ... with a replan to a goal 2 m away (~850 pops): the north star's 2000 x 2000 step, closed loop
]
OTHER_BUDGET_S = 420.0                                 # wall clock all of them together may take: a slow or hung child costs the others, never the headline


def run_other_configs(steps, warmup):
    """BASELINE.json configs[2] on one GPU (1M particles, no replan) and configs[3] / configs[4] (2000 x 2000 with 100k particles,
    4096 x 4096 with 256k) as short child runs of this script, the latter with the replan goal 40 and 400 cells (2 m, 20 m) from the start.  SURVEY.md section 8d asks for the
    farthest free cell; with the reference's cost function and its open list without de-duplication a goal 1600 cells away in
    this world exhausts 64 GB of host memory in the CPU oracle before it returns (measured), and 400 cells away on the 4096 x 4096
    SLAM-built map overflows a 16 M-entry open list after 1.9e7 pops here; so the sweep stops where the reference's own
    algorithm still terminates, and says how many pops each goal costs (an entry with "error" is a search that did not).  Started BEFORE this process
    initialises HIP (a child is a fork + exec)."""
    import subprocess
    out = []
    t_all = time.perf_counter()
    for cfg, l1, extra in OTHER_CONFIGS:
        left = OTHER_BUDGET_S - (time.perf_counter() - t_all)
        if left < 10.0:
            out.append({"config": cfg, "goal_l1_cells": l1, "flags": extra, "error": "skipped: the other_configs wall-clock budget was spent"})
            continue
        child_steps = steps
        if "--sync-steps" in extra:
            i = extra.index("--sync-steps")
            child_steps = int(extra[i + 1])
            extra = extra[:i] + extra[i + 2:]
        # rows with a planner leg carry their own CPU baseline: the oracle on a short sample of the same inputs (pops stated)
        planner_leg = cfg in (4, 5) and "--explore-mode" not in extra
        cpu_steps = (5 if cfg == 5 and "--fixed-goal" not in extra else 3) if planner_leg else 0
        cmd = [sys.executable, os.path.abspath(__file__), "--config", str(cfg), "--goal-l1", str(max(l1, 1)), "--steps", str(child_steps), "--warmup", str(warmup),
               "--cpu-steps", str(cpu_steps), "--sub"] + extra
        t0 = time.perf_counter()
        try:
            r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=min(150.0, left))
            line = [l for l in r.stdout.decode(errors="replace").splitlines() if l.startswith("{")]
            if r.returncode != 0 or not line:
                err = [l for l in r.stderr.decode(errors="replace").splitlines() if l.strip()]
                out.append({"config": cfg, "goal_l1_cells": l1, "flags": extra, "error": err[-1][-300:] if err else f"exit {r.returncode}"})
                continue
            d = json.loads(line[-1])
            out.append({"config": cfg, "goal_l1_cells": l1, "flags": extra, "workload": d["config"]["workload"], "value": d["value"], "unit": d["unit"],
                        "steps": d["steps"], "warmup": d["warmup"], "ms_per_step": d["ms_per_step"], "astar_pops_per_step": d["astar_pops_per_step"],
                        "stage_ms": d["stage_ms"], "planner": [d["config"]["planner_lanes"], d["config"]["planner_batch"], d["config"]["pipeline_depth"]],
                        "streaming_kernels": d.get("streaming_kernels"), "explore": d.get("explore"), "particles": d["config"]["particles"],
                        "cpu_baseline": d.get("cpu_baseline"), "same_search": d.get("same_search"),
                        "wall_s": round(time.perf_counter() - t0, 1)})
        except subprocess.TimeoutExpired:
            out.append({"config": cfg, "goal_l1_cells": l1, "flags": extra, "error": "timed out"})
    return out


def streaming_kernels(ctx, grid, planner, aplanner, pose_dev, goal_pose, W, H):
    """The HBM-bound kernels of the path at this grid size, each timed by its own HIP events in this run (alone on the device,
    after the timed region): bytes the kernel must read + write / its average duration, against the 8 TB/s peak."""
    import torch
    from botlab_amd import _capi
    ids = [_capi.BL_K_DIST_ROWS, _capi.BL_K_DIST_COLS_SUMMARY, _capi.BL_K_DIST_COLS_APPLY, _capi.BL_K_SNAPSHOT, _capi.BL_K_DIST_FUSED]
    ctx.timing_reset()
    ctx.timing_stride(1)
    ctx.timing_enable(True, kernels=ids)
    reps = 12
    for _ in range(reps):
        planner.distances_.forget()                        # the whole-grid kernels, not the window the last map update allows
        planner.setMap(grid)                               # ObstacleDistanceGrid::setDistances on the live map
    if goal_pose is not None:
        for _ in range(reps):                              # the stand-alone snapshot copy (the bench's own rides in the map kernel)
            aplanner.submit(grid, pose_dev, goal_pose)
            aplanner.fetch()
    torch.cuda.synchronize()
    ctx.timing_enable(False)
    # the whole-grid transform again, 60 times back to back without events (an event pair around a launch adds ~3 us to a 11 us
    # kernel): wall time per transform between two synchronisations -- an upper bound of the kernel's duration that still holds
    # the gaps between launches (rocprofv3's kernel trace: profiles/r04_dist_fused.csv)
    for _ in range(5):
        planner.distances_.forget(); planner.setMap(grid)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(60):
        planner.distances_.forget(); planner.setMap(grid)
    torch.cuda.synchronize()
    dist_b2b_ms = (time.perf_counter() - t0) / 60.0 * 1e3
    cells = float(W) * H
    wide = W >= 1024 and W % 16 == 0
    tall = H >= 512 and W >= 256 and W % 2 == 0
    spec = [("k_dist_rows_wide" if wide else "k_dist_rows", _capi.BL_K_DIST_ROWS, 3.0 * cells),             # int8 in, uint16 out
            ("k_dist_cols_summary+carry", _capi.BL_K_DIST_COLS_SUMMARY, 2.0 * cells),                     # uint16 in (+ 1 MB of strip summaries)
            ("k_dist_cols_apply" if tall else "k_dist_cols", _capi.BL_K_DIST_COLS_APPLY, 4.0 * cells),   # uint16 in; uint16 out
            ("k_planner_snapshot", _capi.BL_K_SNAPSHOT, 2.0 * cells),                                     # int8 in, int8 out
            # the whole transform as one launch (grids of 512 x 512 .. 4096 x 4096 cells, W % 16 == 0): int8 in, uint16 out; the
            # row grid never exists.  The three entries above then have no launches.
            ("k_dist_fused", _capi.BL_K_DIST_FUSED, 3.0 * cells)]
    out = {}
    for name, kid, nbytes in spec:
        ms, n = ctx.timing_get(kid)
        if n:
            gbs = nbytes / (ms / n * 1e-3) / 1e9
            out[name] = {"bytes_per_launch": nbytes, "avg_launch_ms": round(ms / n, 5), "launches": int(n), "achieved_GBps": round(gbs, 1),
                         "frac_of_hbm_peak": round(gbs / HBM_PEAK_GBS, 4)}
            if name == "k_dist_fused":
                out[name]["back_to_back_ms_per_transform"] = round(dist_b2b_ms, 5)
                out[name]["achieved_GBps_back_to_back"] = round(nbytes / (dist_b2b_ms * 1e-3) / 1e9, 1)
                # the same time priced at the 9 B/cell the four-launch form moves (what round 3's review asked for: >= 2 800)
                out[name]["at_four_launch_bytes_GBps"] = round(9.0 * cells / (dist_b2b_ms * 1e-3) / 1e9, 1)
    return out


COMPACT_LIMIT = 4096             # bytes: the last stdout line must stay well below what the driver's reader takes (a 21.7 KB line was not read in round 5)


def _short(v, n=160):
    return v if not isinstance(v, str) or len(v) <= n else v[:n - 3] + "..."


def compact_line(out):
    """The ONE line stdout carries: the contract's keys, `roofline` (with traffic and the VALU figure), `cpu_baseline` and a summary of
    the other rows.  Everything else (other_configs, astar_fixtures, same_search, explore, streaming kernels, slowest steps) is the
    detail record: gpurun_out/bench_detail.json and stderr.  Serialised length < COMPACT_LIMIT, whatever the run produced: optional
    keys are dropped from the end of `optional` until it fits."""
    def rnd(x, n=6):
        return round(x, n) if isinstance(x, float) else x
    c = {k: out[k] for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "steady_ms_per_step", "higher_is_better",
                             "scaling", "vs_baseline", "dtype", "data") if k in out}
    for k in ("value", "ms_per_step", "steady_ms_per_step"):
        if k in c:
            c[k] = rnd(c[k], 6)
    cfg = out["config"]
    c["config"] = {"workload": _short(cfg["workload"], 200), "particles": cfg["particles"], "grid": cfg["grid"], "rays": cfg["rays"],
                   "pipeline_depth": cfg["pipeline_depth"], "planner_lanes": cfg["planner_lanes"], "planner_batch": cfg["planner_batch"],
                   "parallelism": cfg["parallelism"], "collective": _short(cfg["collective"], 100)}
    r = out["roofline"]
    c["roofline"] = {"bound": r["bound"], "achieved": rnd(r["achieved"], 3), "peak": r["peak"], "unit": r["unit"], "frac": rnd(r["frac"], 6),
                     "traffic": r["traffic"], "kernel": r["kernel"], "algorithmic_bytes_per_launch": r["algorithmic_bytes_per_launch"],
                     "avg_launch_ms": rnd(r["avg_launch_ms"], 6), "launches_timed": r["launches_timed"], "event_stride": r["event_stride"],
                     "binding_resource": "VALU issue",
                     "valu_issue_frac": rnd(r["valu"]["valu_issue_frac_at_4_cycles"], 4) if r.get("valu") else None,
                     "valu_per_particle_ray": rnd(r["valu"]["per_particle_ray"], 2) if r.get("valu") else None,
                     "traffic_source": _short((r.get("traffic_source") or "").split(" (")[0], 60) or None,
                     "step_frac": rnd(r["step"]["frac"], 6)}
    if "cpu_baseline" in out:
        b = out["cpu_baseline"]
        c["cpu_baseline"] = {"value": rnd(b["value"], 4), "unit": b["unit"], "cores": b["cores"], "kind": b["kind"], "sample": _short(b["sample"], 220)}
    c["preheat_ms"] = out.get("preheat_ms")
    c["stage_ms"] = out.get("stage_ms")
    c["astar_pops_per_step"] = rnd(out.get("astar_pops_per_step"), 2)
    c["final_pose"] = out.get("final_pose")
    optional = []
    if out.get("timed_region_note"):
        optional.append(("timed_region_note", _short(out["timed_region_note"], 200)))
    summ = out.get("summary")
    if summ:
        optional.append(("summary", summ))
    if "particle_sweep_steps_per_s" in out:
        optional.append(("particle_sweep_steps_per_s", out["particle_sweep_steps_per_s"]))
    if "shard_exchange" in out:
        se = out["shard_exchange"]
        optional.append(("shard_exchange", {"form": _short(se["form"], 80), "measured_on_more_than_one_device": se["measured_on_more_than_one_device"],
                                            "bytes_sent_per_rank_per_step": se["bytes_sent_per_rank_per_step"]}))
    if out.get("detail"):
        optional.append(("detail", out["detail"]))
    for k, v in optional:
        c[k] = v
    while len(json.dumps(c)) >= COMPACT_LIMIT and optional:
        k, _ = optional.pop()
        del c[k]
    if len(json.dumps(c)) >= COMPACT_LIMIT:                  # cannot happen with the bounded strings above; never print an unreadable line
        c["config"]["workload"] = _short(c["config"]["workload"], 80)
        c.get("cpu_baseline", {}).pop("sample", None)
    return c


def summary_of(out):
    """Five numbers lifted out of the detail rows (the review's list): closed-loop latency, config 4 / goal 400 closed loop, config 5 as
    written, and the maze row of the reference's A* table, HIP beside the CPU oracle."""
    s = {}
    lat = out.get("latency") or {}
    if "headline" in lat:
        s["closed_loop_ms"] = round(lat["headline"]["sync_ms_per_step"], 4)
    if "config4_goal400" in lat:
        s["config4_goal400_closed_loop_steps_s"] = round(lat["config4_goal400"]["sync_steps_per_s"], 1)
    if "config4_goal40" in lat:
        s["config4_goal40_closed_loop_steps_s"] = round(lat["config4_goal40"]["sync_steps_per_s"], 1)
    for r in out.get("other_configs") or []:
        if "error" in r:
            continue
        if r["config"] == 5 and not r["flags"]:
            s["config5_as_written_steps_s"] = round(r["value"], 1)
        if r["config"] == 4 and r["goal_l1_cells"] == 400 and r.get("planner") and r["planner"][2] != 0:
            s["config4_goal400_pipelined_steps_s"] = round(r["value"], 1)
        if r["config"] == 3:
            s["config3_1m_steps_s"] = round(r["value"], 1)
    fx = (out.get("astar_fixtures") or {}).get("rows") or {}
    for name in ("maze", "convex"):
        row = fx.get(name)
        if row and row["hip"]["success"]:
            s[f"astar_{name}_success_mean_us"] = {"hip": row["hip"]["success"]["mean_us"],
                                                  "cpu_1_core": (row.get("cpu_oracle_1_core") or {}).get("success", {}).get("mean_us") if row.get("cpu_oracle_1_core") and row["cpu_oracle_1_core"]["success"] else None}
    ss = out.get("same_search")
    if ss:
        s["same_search"] = {"pops": ss["pops"], "pops_equal": ss["pops_equal"], "hip_ms": ss["hip_ms"], "cpu_oracle_ms": ss["cpu_oracle_ms"]}
    errs = sum(1 for r in out.get("other_configs") or [] if "error" in r)
    if errs:
        s["other_configs_with_error"] = errs
    return s or None


def emit(out, json_fd, full):
    """Detail record first (stderr + gpurun_out/bench_detail.json), then ONE compact line on stdout (the full record instead for
    --full-line: the child runs of other_configs and the profile collector read it)."""
    out["summary"] = summary_of(out)
    detail_path = None
    if not full:
        try:
            d = os.path.join(ROOT, "gpurun_out")
            os.makedirs(d, exist_ok=True)
            detail_path = os.path.join(d, "bench_detail.json")
            with open(detail_path, "w") as f:
                json.dump(out, f)
                f.write("\n")
        except OSError:
            detail_path = None
        sys.stderr.write("[bench-detail] " + json.dumps(out) + "\n")
        sys.stderr.flush()
        out["detail"] = ("gpurun_out/bench_detail.json and the [bench-detail] line on stderr: other_configs, astar_fixtures, same_search, latency, "
                         "explore, streaming_kernels, host times") if detail_path else "the [bench-detail] line on stderr"
    line = json.dumps(out if full else compact_line(out))
    sys.stdout.flush()
    os.write(json_fd, (line + "\n").encode())


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2000)
    ap.add_argument("--warmup", type=int, default=100)
    ap.add_argument("--particles", type=int, default=100_000)
    ap.add_argument("--grid", type=int, default=200, help="grid side in cells (200 = a shipped map, see --map)")
    ap.add_argument("--map", default="obstacle_slam_10mx10m_5cm", choices=["obstacle_slam_10mx10m_5cm", "convex_10mx10m_5cm"],
                    help="the shipped 200x200 map (tests/golden/reference_maps.npz) scans are cast on and the filter localises in")
    ap.add_argument("--max-range", type=float, default=8.0, help="range of the synthetic lidar in metres (rays that hit nothing "
                    "report it)")
    ap.add_argument("--start", type=float, nargs=3, default=None, metavar=("X", "Y", "THETA"),
                    help="start pose of the driven loop on a 200x200 map (default: a loop with clearance, see build_inputs)")
    ap.add_argument("--no-astar", action="store_true", help="skip the replan (distance grid + A*) in the step")
    ap.add_argument("--goal", type=float, nargs=2, default=None, metavar=("X", "Y"),
                    help="replan goal in metres (default: a point on the driven loop, see build_inputs)")
    ap.add_argument("--goal-l1", type=int, default=40, help="without --goal on a non-default grid: max L1 distance (cells) "
                    "of the replan goal from the start")
    ap.add_argument("--cpu-steps", type=int, default=25, help="steps of the CPU baseline sample (0 = skip)")
    ap.add_argument("--depth", type=int, default=14, help="steps enqueued ahead of fetching a result (0 = synchronous steps); a replan "
                    "(distance grid + A*, ~0.4 ms) spans several steps and a lane sends its batch off when it is full, so fewer than "
                    "~lanes x batch + 2 leaves the SLAM stream waiting for the host")
    ap.add_argument("--lanes", type=int, default=2, help="replanner streams: consecutive replans run concurrently (1..4)")
    ap.add_argument("--batch", type=int, default=6, help="replans a lane collects and searches in one launch (1..64): lanes x batch "
                    "searches overlap; for grids where a search outlasts several steps (use with --depth >= lanes x batch)")
    ap.add_argument("--config", type=int, default=0, choices=[0, 2, 3, 4, 5],
                    help="BASELINE.json configs[i - 1] as a preset (0/2: the default, configs[1]; 3: 1M-particle MCL, no replan; "
                         "4: 2000x2000 maze with a replan per step; 5: 4096x4096, 256k particles); explicit flags still win")
    ap.add_argument("--explore", action="store_true", help="the exploration loop (BASELINE.json configs[4]): instead of a replan to a fixed goal, every "
                    "5th step submits Exploration::executeExploringMap's step -- setMap + find_map_frontiers + plan_path_to_frontier under the "
                    "0.5 m rule -- to the explorer lanes (preset 5 turns it on)")
    ap.add_argument("--fixed-goal", action="store_true", help="preset 5 without the exploration step: SLAM + a replan to a fixed goal per step")
    ap.add_argument("--explore-lanes", type=int, default=4, help="explorer side streams: exploration steps of consecutive maps overlap")
    ap.add_argument("--explore-mode", choices=["every-map", "newest-map"], default="every-map",
                    help="every-map: every published map gets its exploration step; the SLAM thread takes finished steps back itself and "
                         "waits when all lanes are busy (a plan_path_to_frontier of seconds stalls it).  newest-map: the reference's "
                         "arrangement -- the exploration process is a separate LCM subscriber that works on the newest map it has when it "
                         "is free (exploration.cpp:93-109, 296-298) and the SLAM process never waits for it: a second host thread takes the "
                         "steps back, and a map published while every lane is busy is not explored (counted)")
    ap.add_argument("--preheat-ms", type=float, default=300.0, help="milliseconds of the path's own k_mcl_main on a scratch filter before the warmup steps, "
                    "so that a short run is not a measurement of the clock ramp after the idle seconds of input synthesis (0 = off)")
    ap.add_argument("--astar-fixtures", action="store_true", help="only the reference's own A* table (astar_test's six maps): one JSON line")
    ap.add_argument("--sub", action="store_true", help="a child run of the default invocation (other_configs): no children of its own; prints the full record")
    ap.add_argument("--full-line", action="store_true", help="print the full record on stdout instead of the compact line (tools; --sub implies it)")
    ap.add_argument("--no-other-configs", action="store_true", help="default run: skip the short runs of configs 4 and 5")
    ap.add_argument("--other-steps", type=int, default=1000, help="timed steps of each other_configs run")
    args = ap.parse_args()
    presets = {3: dict(particles=1_000_000, no_astar=True, map="convex_10mx10m_5cm"),     # slam.cpp:36-45: --localization-only <map>
               4: dict(grid=2000, lanes=3, batch=64, depth=256),
               5: dict(grid=4096, particles=256_000, lanes=3, batch=16, depth=64)}
    for key, val in presets.get(args.config, {}).items():
        if getattr(args, key) == ap.get_default(key):
            setattr(args, key, val)
    if args.config == 5 and not args.fixed_goal:
        args.explore = True
    if args.explore:
        args.no_astar = True                       # the exploration step is the planner's work; no fixed-goal replan beside it

    # stdout carries ONE JSON line: anything native code prints there (RCCL writes a version banner to stdout when its first
    # communicator comes up) goes to stderr instead
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # Started without a launcher: this process becomes the launcher.  It has not imported torch or touched HIP, starts
        # one rank per GPU as a CHILD process group (never an exec), forwards rank 0's JSON line and exits with the group's status.
        os.dup2(json_fd, 1)
        os.close(json_fd)
        raise SystemExit(self_launch(args.gpus))
    other = None
    if args.config == 0 and args.gpus == 1 and args.grid == 200 and not args.sub and not args.no_other_configs and "WORLD_SIZE" not in os.environ:
        other = run_other_configs(args.other_steps, 150)
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if os.environ.get("BENCH_TEST_HANG") and "WORLD_SIZE" in os.environ:     # test hook (tests/test_bench_launcher_cpu.py): a rank that never gets anywhere
        time.sleep(3600)
    if world != args.gpus:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}: start it as `python bench.py --gpus N` or under "
                         f"torch.distributed.run with --nproc-per-node equal to --gpus")

    import torch
    import torch.distributed as dist
    import botlab_amd as bl
    from botlab_amd import _capi, sharded

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: botlab_amd has no CPU path")
    if args.astar_fixtures:
        torch.cuda.init()
        table = astar_fixtures(bl.default_context(), reps=5)
        os.write(json_fd, (json.dumps({"astar_fixtures": table}) + "\n").encode())
        return
    phase(json_fd, "process-group")
    # Test hook (tests/test_gpu_bench_two_ranks.py): exercise the N > 1 code path on a one-GPU box -- every rank on cuda:0 and
    # the collectives over gloo (RCCL refuses two ranks on one device).  Never set by the driver.
    one_device = bool(os.environ.get("BENCH_TEST_ONE_DEVICE"))
    if one_device:
        local_rank = 0
    elif local_rank >= torch.cuda.device_count():
        raise SystemExit(f"bench.py: rank {rank} of {world} has no GPU (this node shows {torch.cuda.device_count()}): --gpus {args.gpus} "
                         f"cannot run here")
    torch.cuda.set_device(local_rank)
    if world > 1 or os.environ.get("BOTLAB_FORCE_COLLECTIVES"):
        import datetime
        if one_device:
            dist.init_process_group("gloo", timeout=datetime.timedelta(seconds=300))
        else:
            # (a collective that never completes -- first contact with several devices -- ends the run after five minutes, not thirty)
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank), timeout=datetime.timedelta(seconds=300))

    total = args.steps + args.warmup + 34
    cpm = np.float32(1.0 / np.float64(np.float32(0.05)))

    # Sharded runs take the composed finish (own blocks, sources read from their owners' memory, two small all-gathers: DESIGN.md
    # section 6) when every rank can map every other rank's memory -- asked before the engines are made, because the two forms
    # cut the particle set at different bounds; otherwise the replicated form (all-gather of the whole record).
    composed = False
    phase(json_fd, "ipc-probe")
    if world > 1:
        # first contact with several devices: say what the runtime reports before anything relies on it
        nd = torch.cuda.device_count()
        if rank == 0 and not one_device:
            pairs = [f"{a}->{b}:{int(torch.cuda.can_device_access_peer(a, b))}" for a in range(min(world, nd)) for b in range(min(world, nd)) if a != b]
            sys.stderr.write(f"[bench] can_device_access_peer: {' '.join(pairs)}\n")
    if world > 1 and sharded.composed_possible(args.particles, world):
        probe = bl.Context(local_rank)
        composed = sharded.ipc_probe(probe, rank, world)
        probe.close()
        if rank == 0:
            sys.stderr.write(f"[bench] shard finish: {'composed (own blocks, sources read from their owners)' if composed else 'replicated record (no IPC mapping between the ranks)'}\n")
    phase(json_fd, "engine")
    engine = sharded.HipShardEngine(args.particles, rank, world, local_rank, composed=composed)
    ctx = engine.ctx
    # The replanner on its own stream(s) (the reference's planner process); consecutive replans overlap on `lanes` streams.
    # Created before anything touches the null stream: the HIP runtime multiplexes streams onto 4 hardware queues (raising
    # GPU_MAX_HW_QUEUES costs ~50 us of launch latency per kernel, measured), and two lanes sharing a queue serialise.
    aplanner = bl.AsyncPlanner(ctx=ctx, lanes=args.lanes, batch=args.batch)
    phase(json_fd, "inputs")
    m, truth, poses, odo, scans, rands = build_inputs(args, total, ctx)
    spf = sharded.ShardedParticleFilter(engine)
    grid = bl.OccupancyGrid.from_cells(m["cells"], m["origin"], m["mpc"], cellsPerMeter=cpm, ctx=ctx)
    mapper = bl.Mapping(5.0, 4, 1, ctx=ctx)                 # slam.cpp:24, slam_main.cpp:22-23
    planner = bl.MotionPlanner(ctx=ctx)                       # robotRadius 0.2 (motion_planner.hpp:31)
    planner.setMap(grid)
    goal = None
    if not args.no_astar:
        if args.goal is not None:
            goal = tuple(args.goal)
        elif args.grid == 200:
            # Replan target: a fixed point on the driven loop, 0-0.9 m from the robot (0-450 pops per search).  The
            # reference's cost function (negative obstacle cost, duplicate re-expansion) makes most farther goals on this
            # map take 1e5-2e6 pops, which the reference itself cannot finish; see DESIGN.md "A*".
            goal = (-0.35, 0.2) if args.start is None else (args.start[0] + 0.4, args.start[1])
        else:
            goal = pick_goal(planner.distances_.cells(), m["origin"], poses[0][:2], 0.2, args.goal_l1)
    goal_pose = bl.make_pose(goal[0], goal[1], 0.0) if goal else None

    phase(json_fd, "filter-init")
    spf.initializeFilterAtPose(bl.make_pose(*odo[0], utime=int(scans[0].times[0])), seed=42)
    if world > 1 and rank == 0:
        form = "peer stores (no collective)" if getattr(spf, "peer", False) else ("two small all-gathers" if spf.composed else "all-gather of the whole record")
        sys.stderr.write(f"[bench] shard exchange form: {form}" + (f" -- peer stores not taken: {spf.peer_why}" if spf.composed and not spf.peer and spf.peer_why else "") + "\n")
    phase(json_fd, "planner-setup")
    explorer = bl.AsyncExplorer(ctx=ctx, lanes=args.explore_lanes, robotRadius=0.2) if args.explore else None
    ex_log = []                       # one entry per fetched exploration step

    if goal_pose is not None:
        # setup, not measurement: every replanner unit creates its scratch (open-list heap, snapshot grids, result slots) on
        # first use -- milliseconds of hipMalloc each.  Two rounds of submissions against the initial map and pose touch every
        # unit and both of its snapshot slots, so that runs with a short warmup do not time allocations.
        for _ in range(2):
            for _ in range(args.lanes * args.batch):
                aplanner.submit(grid, engine.pf.poseDevicePtr(), goal_pose)
            for _ in range(args.lanes * args.batch):
                aplanner.fetch()

    pops_total = [0]

    pose_dev = engine.pf.poseDevicePtr()
    # the filter's end rides in the map kernel (bl_mapping_update_finishing_pf), behind the record exchange when sharded
    prefetch = not os.environ.get("BENCH_NO_PREFETCH")
    ride_finish = not os.environ.get("BENCH_NO_RIDE")
    in_flight = []                  # steps enqueued whose result has not been fetched yet

    def explore_fetch():
        r, _ = explorer.fetch(want_path=False)
        ex_log.append((r.status, r.num_frontiers, r.frontier_cells, r.planned, r.pops, r.searches, r.path_length, r.bfs_cells, r.bfs_levels,
                       r.frontiers_ms, r.plan_ms))

    ex_skipped = [0]
    ex_thread = [None]
    ex_outstanding = [0]              # newest-map mode: submissions the worker thread has not taken back yet
    import threading
    ex_cv = threading.Condition()

    def explore_worker():
        while True:
            with ex_cv:
                while ex_outstanding[0] == 0:
                    ex_cv.wait()
                if ex_outstanding[0] < 0:
                    return
            explore_fetch()               # (the library call releases the interpreter lock)
            with ex_cv:
                ex_outstanding[0] -= 1
                ex_cv.notify_all()

    def explore_submit():
        if args.explore_mode == "newest-map":
            if ex_thread[0] is None:
                ex_thread[0] = threading.Thread(target=explore_worker, daemon=True)
                ex_thread[0].start()
            if explorer.pending() >= args.explore_lanes:
                ex_skipped[0] += 1        # the explorer is busy (planning): this map is not explored
                return
            explorer.submit(grid, pose_dev)
            with ex_cv:
                ex_outstanding[0] += 1
                ex_cv.notify_all()
            return
        # every-map: the published map and the pose of this step go to an explorer lane; the oldest step is taken back first when
        # every lane holds one (its plan_path_to_frontier, when due, runs inside that fetch: the host waits, the SLAM stream runs on)
        while explorer.pending() >= args.explore_lanes:
            explore_fetch()
        explorer.submit(grid, pose_dev)

    def enqueue(k):
        # Everything of one step is enqueued on the ctx stream; the pose estimate stays on the device and feeds the map
        # update and the A* start there.
        o = odo[k + 1]
        sc = scans[k]
        odo_pose = bl.make_pose(o[0], o[1], o[2], utime=sc.utime)
        if ride_finish:
            # the end of the filter update (pose estimate + weight prefix) rides in the map kernel's launch; on shards the
            # record exchange comes first and every rank ends the update the same way
            spf.updateBegin(odo_pose, sc, grid, int(rands[k]))
            if prefetch and k + 1 < len(scans):
                ctx.scanPrefetch(scans[k + 1])       # the next scan is queued (slam.cpp:96-104): it rides in this step's map kernel
            if goal_pose is not None:
                aplanner.submit_with_map_update_finishing(mapper, sc, engine.pf, sc.utime, grid, goal_pose)
            else:
                mapper.updateMapFinishingFilter(sc, engine.pf, sc.utime, grid)
            if explorer is not None and k % EXPLORE_EVERY == 0:
                explore_submit()
            in_flight.append(k)
            return
        spf.updateFilter(odo_pose, sc, grid, int(rands[k]), want_pose=False)
        if prefetch and k + 1 < len(scans):
            ctx.scanPrefetch(scans[k + 1])
        if goal_pose is not None:
            # updateMap with the device-resident pose, then snapshot map + pose for the replanner (one library call);
            # setDistances + search_for_path overlap the next step on a replanner lane
            aplanner.submit_with_map_update(mapper, sc, pose_dev, sc.utime, grid, goal_pose)
        else:
            mapper.updateMapDevicePose(sc, pose_dev, sc.utime, grid)
        if explorer is not None and k % EXPLORE_EVERY == 0:
            explore_submit()
        in_flight.append(k)

    last_pose = [None]

    def fetch():
        # hands back the pose and the path of the oldest enqueued step (waits for that step only)
        in_flight.pop(0)
        if goal_pose is not None:
            path, st = aplanner.fetch(return_stats=True)
            pops_total[0] += st[0]
            last_pose[0] = path[0]               # the start pose of the path is that step's pose estimate
            return path[0]
        return None

    host_t = [0.0, 0.0]
    step_wall = []
    trace_every = int(os.environ.get("BENCH_TRACE_RATE", "0"))

    def step(k):
        # software pipeline of depth args.depth: step k is enqueued before the result of step k - depth is fetched, so the
        # GPU never waits for the host between steps; every result is still delivered, in order
        t0 = time.perf_counter()
        enqueue(k)
        t1 = time.perf_counter()
        last = None
        while len(in_flight) > args.depth:
            last = fetch()
        t2 = time.perf_counter()
        host_t[0] += t1 - t0
        host_t[1] += t2 - t1
        step_wall.append((t2 - t0, k, t1 - t0))
        if trace_every and k % trace_every == 0:            # BENCH_TRACE_RATE=<steps>: where a long run spends its time
            sys.stderr.write(f"[bench] step {k}: enqueue {host_t[0]:.3f} s, fetch {host_t[1]:.3f} s so far\n")
        return last

    def drain(wait_explorer=True):
        # newest-map mode: the exploration process is its own consumer -- the SLAM loop's steps are complete without it, and the
        # steps it still holds when the timed region ends are finished behind it (wait_explorer=False there; they are reported)
        if explorer is not None and args.explore_mode == "newest-map" and wait_explorer:
            with ex_cv:
                while ex_outstanding[0] > 0:
                    ex_cv.wait()
        while explorer is not None and args.explore_mode != "newest-map" and explorer.pending():
            explore_fetch()
        if goal_pose is not None and in_flight and not os.environ.get("BENCH_NO_FLUSH"):
            aplanner.flush()                 # end of the scan stream: the batches still collecting go out beside the SLAM stream's last steps
        while in_flight:
            fetch()
        if goal_pose is None or last_pose[0] is None:
            return engine.pf.poseEstimate()
        return last_pose[0]

    # The measured loop is a few hundred microseconds per step; a generational GC pass over the interpreter's (torch-sized)
    # heap is tens of milliseconds.  Collect once now and keep the collector out of the timed region.
    import gc
    gc.collect()
    gc.freeze()
    gc.disable()

    phase(json_fd, "warmup")
    # Setup, not measurement: the device sits idle for seconds while the host synthesises scans, and its clock ramps back over
    # ~0.3 s of load -- longer than a whole 25-step run (measured: k_mcl_main 71-74 us in a 20-step run after idling, 63 us
    # after 300 ms of load, the figure every long run shows).  A SLAM loop is a continuously running service, so the clock is
    # brought up first -- by the path's own dominant kernel: a scratch particle filter of the same size runs updateFilter on the
    # first scan against the same map (k_mcl_main + its finish; nothing of the measured filter, map or planner is touched); the W
    # warmup steps and the K timed steps follow unchanged.  --preheat-ms 0 switches it off; the line says which was used.
    preheat_ms = args.preheat_ms
    if preheat_ms > 0:
        heat = bl.ParticleFilter(min(args.particles, 250_000), ctx=ctx)
        heat.initializeFilterAtPose(bl.make_pose(*odo[0], utime=int(scans[0].times[0])), seed=7)
        t_ph = time.perf_counter()
        j = 0
        while (time.perf_counter() - t_ph) * 1e3 < preheat_ms:
            for _ in range(16):
                o = odo[1 + (j & 1)]
                heat.updateFilter(bl.make_pose(o[0], o[1], o[2], utime=int(scans[0].utime) + j), scans[0], grid)
                j += 1
        del heat
    k = 0
    for _ in range(args.warmup):
        step(k)
        k += 1
    drain()
    ctx.timing_reset()
    event_stride = EVENT_STRIDE_SHORT_RUN if args.steps <= 50 else (EVENT_STRIDE if args.steps <= 400 else EVENT_STRIDE_LONG)
    if not os.environ.get("BENCH_NO_EVENTS"):
        # HIP events of the dominant kernel only (roofline leg), on every EVENT_STRIDE-th launch of the timed region: the
        # launch carries its own start / stop events (hipExtLaunchKernelGGL), which hold the kernel's begin and end time stamps
        # -- a pair of hipEventRecord calls around it also measures the launch gap (~13 us here)
        ctx.timing_stride(event_stride)
        ctx.timing_enable(True, kernels=[_capi.BL_K_MCL_MAIN])
    pops_total[0] = 0
    host_t[0] = host_t[1] = 0.0
    del step_wall[:]
    del ex_log[:]
    ex_skipped[0] = 0
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    phase(json_fd, "timed")
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step(k)
        k += 1
    t_steady = time.perf_counter() - t0  # the K steps enqueued, all but the last `depth` results fetched: the loop without its drain tail
    pose = drain(wait_explorer=False)    # all K results delivered inside the timed region
    if explorer is not None and args.explore_mode == "newest-map":
        # the one deviation from "synchronise the device on both sides": the exploration process of this mode is an independent
        # consumer whose plan_path_to_frontier (seconds of A* on its own stream) is NOT the SLAM loop's work; a device-wide
        # synchronise would wait for it.  The SLAM stream is synchronised instead, and the line says so.
        engine.stream.synchronize()
    else:
        torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    ctx.timing_enable(False)
    ctx.timing_stride(1)
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    phase(json_fd, "stage-pass")
    ex_timed = list(ex_log)
    ex_skipped_timed = ex_skipped[0]
    ex_in_flight_at_end = ex_outstanding[0]
    tail_s = 0.0
    if explorer is not None and args.explore_mode == "newest-map":
        t_tail = time.perf_counter()
        drain()                      # the explorer's last steps (and their plans), outside the timed region
        torch.cuda.synchronize()
        tail_s = time.perf_counter() - t_tail
    main_ms_total, main_n = ctx.timing_get(_capi.BL_K_MCL_MAIN)
    host_ms = (1e3 * host_t[0] / args.steps, 1e3 * host_t[1] / args.steps)
    slowest = sorted(step_wall, reverse=True)[:4]
    pops_timed = pops_total[0]
    final_pose, final_k = pose, k
    # per-stage kernel times: a short untimed pass with every timer on (events between launches cost a few us each, so
    # they stay out of the timed region)
    ctx.timing_reset()
    ctx.timing_enable(True)
    aplanner.timing(1)
    for _ in range(min(args.steps, 30, total - k - 1)):
        step(k)
        k += 1
    drain()
    torch.cuda.synchronize()
    ctx.timing_enable(False)
    stage_ms = {}
    for name, kid in (("mcl_main", _capi.BL_K_MCL_MAIN), ("mcl_scan", _capi.BL_K_MCL_SCAN), ("map", _capi.BL_K_MAP)):
        ms, n = ctx.timing_get(kid)
        stage_ms[name] = (ms / n if n else 0.0, n)
    d_ms, a_ms, pn = aplanner.timing(-1)
    aplanner.timing(0)
    stage_ms["dist"] = (d_ms / pn if pn else 0.0, pn)
    stage_ms["astar"] = (a_ms / pn if pn else 0.0, pn)
    if main_n:
        stage_ms["mcl_main"] = (main_ms_total / main_n, main_n)                      # the timed-region figure
    # (a run too short for the event stride to catch a launch keeps the post-pass figure)
    pose, k = final_pose, final_k
    pops_total[0] = pops_timed
    stream_k = None
    if world == 1 and (args.sub or args.grid != 200):
        stream_k = streaming_kernels(ctx, grid, planner, aplanner, pose_dev, goal_pose, m["cells"].shape[1], m["cells"].shape[0])

    if os.environ.get("BOTLAB_FINISH_LOOKAHEAD") and world == 1:
        engine.pf.debugEstimateStats()          # the library prints the map update's look-ahead counters to stderr
    phase(json_fd, "report")
    if rank == 0:
        N, R = args.particles, scans[0].num_ranges
        W, H = m["cells"].shape[1], m["cells"].shape[0]
        n_local = engine.hi - engine.lo
        # algorithmic bytes of one k_mcl_main launch: 128 B per particle of this shard
        # + the grid once + 20 B per ray
        alg_bytes = 128.0 * n_local + W * H + 20.0 * R
        main_ms = stage_ms["mcl_main"][0]
        # HBM bytes per launch from the rocprofv3 PMC passes committed under profiles/ (FETCH_SIZE / WRITE_SIZE, separate
        # passes, gfx950 correction applied); only valid for the configuration they were collected on
        traffic = None
        traffic_src = None
        for prof in ("r06_mcl_main_traffic.json", "r05_mcl_main_traffic.json", "r03_mcl_main_traffic.json", "r02_mcl_main_traffic.json", "r01_mcl_main_traffic.json"):
            try:
                tj = json.load(open(os.path.join(ROOT, "profiles", prof)))
                if world == 1 and tj["config"] == {"particles": N, "grid": [W, H], "rays": R}:
                    traffic = tj["traffic_bytes_per_launch"]
                    traffic_src = "profiles/" + prof + " (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this command, not measured in this run)"
                    break
            except (OSError, KeyError, ValueError):
                continue
        achieved = alg_bytes / (main_ms * 1e-3) / 1e9 if main_ms > 0 else 0.0
        # The other kernels of the step against the same roof, with SURVEY.md section 8d's algorithmic bytes:
        # map 16 R + 2 C (C = traced cells), distance grid 5 W H per setDistances, A* 52 B per pop; and the whole step.
        traced = float(np.mean([np.sum(np.minimum(sc.ranges[sc.ranges <= 5.0], 5.0)) * 20.0 for sc in scans[:64]]))
        pops_step = pops_total[0] / max(args.steps, 1)
        per_launch = max(args.batch, 1)
        others = {}
        map_bytes = 16.0 * R + 2.0 * traced + (24.0 * N if ride_finish else 0.0)     # riding finish: 16 N read + 8 N prefix written
        for name, b in (("map", map_bytes), ("dist", 5.0 * W * H * per_launch), ("astar", 52.0 * pops_step * per_launch)):
            ms = stage_ms[name][0]
            if ms > 0 and (goal is not None or name == "map"):
                gbs = b / (ms * 1e-3) / 1e9
                others[name] = {"algorithmic_bytes_per_launch": round(b, 1), "avg_launch_ms": round(ms, 5), "achieved": round(gbs, 3),
                                "frac": gbs / HBM_PEAK_GBS}
        step_bytes = alg_bytes * world + 24.0 * N + 16.0 * R + 2.0 * traced + (5.0 * W * H + 52.0 * pops_step if goal is not None else 0.0)
        step_gbs = step_bytes / (elapsed / args.steps) / 1e9
        # secondary bound of k_mcl_main (it is VALU-bound, not HBM-bound): wave-level VALU instructions per launch from the
        # committed SQ counter pass, against the 1024 SIMDs issuing one per 4 cycles at 2.4 GHz
        valu = None
        import csv
        for prof in ("r06_mcl_main_pmc_sq.csv", "r05_mcl_main_pmc_sq.csv", "r03_mcl_main_pmc_sq.csv", "r02_mcl_main_pmc_sq.csv", "r01_mcl_main_pmc_sq.csv"):
            try:
                if world == 1 and traffic is not None and valu is None:
                    for row in csv.reader(open(os.path.join(ROOT, "profiles", prof))):
                        if len(row) == 4 and row[0].startswith("void k_mcl_main<0") and row[1] == "SQ_INSTS_VALU":
                            insts = float(row[3])
                            valu = {"wave_valu_insts_per_launch": insts, "per_particle_ray": insts * 64.0 / (n_local * R),
                                    "valu_issue_frac_at_4_cycles": insts * 4.0 / (1024.0 * main_ms * 1e-3 * 2.4e9),
                                    "source": "profiles/" + prof + " (rocprofv3 --pmc SQ_INSTS_VALU pass of this command, not measured in this run)"}
            except (OSError, ValueError):
                continue
        if args.explore:
            world_name = "sealed obstacle_slam arena in every 200-cell tile; the map starts with the centre arena known up to column 110, the rest unknown"
        elif args.grid == 200:
            world_name = f"continuing from the shipped, already built {args.map} map"
        else:
            world_name = "tiled astar/maze world, the map starts as the truth-derived map"
        out = {
            # (newest-map mode: the value counts the SLAM loop's steps alone -- the exploration process runs beside it and its tail is
            # NOT in the timed region; explore.exploration_steps_per_s_with_tail is the other half)
            "metric": ("SLAM steps/sec (map+MCL; the exploration process beside it, not waited for)" if args.explore and args.explore_mode == "newest-map"
                       else "SLAM steps/sec (map+MCL+A*)"),
            "value": args.steps / elapsed,
            "unit": "steps/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": 1e3 * elapsed / args.steps,
            # the same K steps without the drain tail (the last `depth` results -- their distance grids and the longest search of the
            # last replanner batch -- are fetched behind it): what a long run converges to; `value` stays the whole region
            "steady_ms_per_step": 1e3 * t_steady / args.steps,
            "higher_is_better": True,
            "scaling": "strong",
            "vs_baseline": None,
            "dtype": "int8 grid / f32 poses with f64 intermediates / int64 weights",
            "data": "synthetic",
            "preheat_ms": preheat_ms,         # the path's own k_mcl_main on a scratch filter before the W warmup steps (the clock ramps over ~0.3 s after the idle input synthesis)
            "config": {"workload": f"full SLAM step on {W}x{H} @5cm grid ({world_name}), {N} particles, "
                                   f"{R} rays, " + ("exploration step (setMap + find_map_frontiers + plan_path_to_frontier under the 0.5 m rule) on every "
                                                    f"{EXPLORE_EVERY}th map" if args.explore else f"A* replan {'off' if goal is None else 'on'}"),
                       "particles": N, "grid": [W, H], "rays": R, "pipeline_depth": args.depth, "planner_lanes": args.lanes, "planner_batch": args.batch,
                       "parallelism": f"particle-shard x{world}" if world > 1 else "single GPU",
                       "collective": ("none" if not (world > 1 or spf.force_collectives) else
                                      "none: peer stores over the ranks' IPC mappings + counter waits" if getattr(spf, "peer", False) else
                                      (("two small RCCL all-gathers (tile sums; records + tables)" if spf.composed else "RCCL all-gather of the whole record")
                                       + " enqueued by the library on the filter's stream" if spf.comm is not None
                                       else "torch.distributed all_gather_into_tensor" + (" x2 (composed finish)" if spf.composed else "")))},
            # ("bound" is the roof SURVEY.md section 8d prescribes for the whole path -- HBM; what actually binds k_mcl_main is VALU
            # issue: "binding_resource" and the "valu" block say so)
            "roofline": {"bound": "hbm", "binding_resource": "VALU issue (the kernel's working set is LDS / L2 resident)", "kernel": "k_mcl_main", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "traffic_source": traffic_src,
                         "algorithmic_bytes_per_launch": alg_bytes, "avg_launch_ms": main_ms,
                         "launches_timed": int(stage_ms["mcl_main"][1]), "event_stride": event_stride,
                         "particle_rays_per_s": (n_local * R) / (main_ms * 1e-3) if main_ms > 0 else 0.0,
                         "valu": valu, "other_kernels": others,
                         "step": {"algorithmic_bytes": round(step_bytes, 1), "achieved": round(step_gbs, 3), "frac": step_gbs / HBM_PEAK_GBS}},
            "stage_note": ("map = k_map_update carrying the filter's end (pose estimate + weight prefix); mcl_scan is its stand-alone launch, "
                           "unused here" if ride_finish else "mcl_scan = weight prefix + pose estimate launches"),
            "stage_ms": {k_: round(v[0], 5) for k_, v in stage_ms.items()},
            "astar_pops_per_step": pops_total[0] / args.steps,
            "host_ms_per_step": {"enqueue": round(host_ms[0], 4), "fetch_wait": round(host_ms[1], 4)},
            "slowest_steps_ms": [[round(1e3 * t, 3), kk, round(1e3 * te, 3)] for t, kk, te in slowest],   # wall, step, of which enqueue
            "final_pose": [pose.x, pose.y, pose.theta],
            "truth_pose": [float(v) for v in poses[k]],
        }
        if args.steps <= 50:
            # a region of a few milliseconds: K steps of the steady rate + the drain tail (the last batch's distance grids and its longest
            # search, ~0.2 ms).  The long form of the same command (--steps 2000 --warmup 100, the default) is the figure DESIGN.md quotes.
            note = (f"{args.steps} timed steps = {1e3 * elapsed:.2f} ms, of which the drain tail {1e3 * (elapsed - t_steady):.2f} ms; "
                    f"steady_ms_per_step is the region without it")
            for prof in ("r06_bench_default.json", "r05_bench_default.json"):
                try:
                    lj = json.load(open(os.path.join(ROOT, "profiles", prof)))
                    if lj["config"]["particles"] == N and lj["config"]["grid"] == [W, H] and lj["n_gpus"] == world:
                        note += f"; the {lj['steps']}-step run of this command: {lj['value']:.0f} steps/s (profiles/{prof}, not measured in this run)"
                        break
                except (OSError, KeyError, ValueError):
                    continue
            out["timed_region_note"] = note
        if world > 1 or spf.force_collectives:
            sent, received, own = spf.exchange_bytes_per_update()
            out["shard_exchange"] = {"form": ("composed finish, peer stores (no collective)" if getattr(spf, "peer", False) else
                                              ("composed finish, two small all-gathers" if spf.composed else "replicated record, one all-gather")),
                                     "peer_stores_not_taken_because": (spf.peer_why if spf.composed and not getattr(spf, "peer", False) else None),
                                     "measured_on_more_than_one_device": bool(world > 1 and not one_device),
                                     "bytes_sent_per_rank_per_step": sent,
                                     "bytes_received_per_rank_per_step": received, "source_record_bytes_read_by_k_mcl_main": own,
                                     "replicated_form_would_receive": (world - 1) * engine.S * 16}
        if args.explore:
            n_ex = len(ex_timed)
            planned = [e for e in ex_timed if e[3]]
            out["explore"] = {
                "mode": args.explore_mode, "exploration_steps": n_ex, "maps_published": args.steps // EXPLORE_EVERY, "maps_not_explored": ex_skipped_timed,
                "exploration_steps_still_running_when_the_timed_region_ended": ex_in_flight_at_end,
                "closing_synchronise": ("SLAM stream only: the explorer's running plan is not waited for" if args.explore_mode == "newest-map" else "device"),
                # exploration steps COMPLETED per second when the steps still running at the end of the region are waited for
                "exploration_steps_per_s_with_tail": round((n_ex + ex_in_flight_at_end) / (elapsed + tail_s), 2),
                "tail_s": round(tail_s, 3),
                "every_nth_slam_step": EXPLORE_EVERY, "lanes": args.explore_lanes,
                "status_counts": {"in_progress": sum(e[0] == 0 for e in ex_timed), "complete": sum(e[0] == 1 for e in ex_timed), "failed": sum(e[0] == 2 for e in ex_timed)},
                "frontiers_per_step": (sum(e[1] for e in ex_timed) / n_ex) if n_ex else 0.0,
                "frontier_cells_per_step": (sum(e[2] for e in ex_timed) / n_ex) if n_ex else 0.0,
                "flooded_cells_per_step": (sum(e[7] for e in ex_timed) / n_ex) if n_ex else 0.0,
                "flood_levels_per_step": (sum(e[8] for e in ex_timed) / n_ex) if n_ex else 0.0,
                "plans": len(planned), "pops_per_plan": (sum(e[4] for e in planned) / len(planned)) if planned else 0.0,
                "searches_per_plan": (sum(e[5] for e in planned) / len(planned)) if planned else 0.0,
                "stage_ms": {"frontiers": round(sum(e[9] for e in ex_timed) / n_ex, 4) if n_ex else 0.0,
                             "plan_to_frontier": round(sum(e[10] for e in planned) / len(planned), 3) if planned else 0.0},
                "plan_ms_total": round(sum(e[10] for e in planned), 2),
            }
        if world == 1 and stream_k is not None:
            out["streaming_kernels"] = stream_k
        if other is not None:
            out["other_configs"] = other
            # closed-loop latency and the particle sweep, lifted out of the rows above
            lat = {}
            for r in other:
                if "error" in r:
                    continue
                if r.get("planner") and r["planner"][2] == 0:
                    lat["headline" if r["config"] == 0 else f"config{r['config']}_goal{r['goal_l1_cells']}"] = {
                        "sync_ms_per_step": r["ms_per_step"], "sync_steps_per_s": r["value"], "steps": r["steps"],
                        "meets_1kHz_closed_loop": bool(r["value"] >= 1000.0)}
            out["latency"] = lat
            out["particle_sweep_steps_per_s"] = {str(r["particles"]): round(r["value"], 1) for r in other
                                                 if "error" not in r and r["config"] in (0, 3) and not r["flags"][:1] == ["--no-astar"]
                                                 and (not r.get("planner") or r["planner"][2] != 0)}
            out["particle_sweep_steps_per_s"][str(args.particles)] = round(args.steps / elapsed, 1)
        if args.cpu_steps > 0 and world == 1:
            out["cpu_baseline"] = cpu_baseline(args, m, odo, scans, rands, goal, args.cpu_steps)
            if goal is not None:
                ss = same_search(ctx, grid, planner, pose, goal_pose, m)
                if ss["pops"] < 1000:
                    # (the run's own replan is a handful of pops: kept as "run_goal", the comparison is made on a fixture search)
                    fx = same_search_fixture(ctx)
                    fx["run_goal"] = ss
                    ss = fx
                out["same_search"] = ss
        if other is not None and world == 1:
            # the reference's own published table (astar_test's six maps), HIP path beside the CPU oracle
            out["astar_fixtures"] = astar_fixtures(ctx, reps=3)
        emit(out, json_fd, full=bool(args.sub or args.full_line))
    phase(json_fd, "teardown")
    if dist.is_initialized():
        dist.barrier()
        torch.cuda.synchronize()
        spf.close()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
