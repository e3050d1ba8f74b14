"""More random worlds than the committed fixture holds (tests/golden/astar_random_cases.json: seeds 1-4): fresh seeds, pairs the CPU
oracle finishes within a few seconds (child process with a time limit, as tests/tools/make_astar_random_cases.py), every one searched
on the device under whatever form the environment selects and compared with the oracle -- poses, pops, pushes.
    python tests/tools/astar_soak.py [first_seed] [n_seeds] [pairs_per_seed]"""
import multiprocessing as mp, os, sys, time
import numpy as np
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE)); sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
import make_astar_random_cases as gen


def main():
    import helpers, oracle_lib
    import botlab_amd as bl
    from scipy import ndimage
    first, n_seeds, per_seed = (int(a) for a in (sys.argv[1:] + ["11", "6", "10"])[:3])
    orc = oracle_lib.load_oracle()
    ctx = bl.default_context()
    bad = done = 0
    pops_total = 0
    biggest = 0
    t0 = time.time()
    for seed in range(first, first + n_seeds):
        cells, rng = gen.world(seed)
        mpc = np.float32(0.05); origin = (np.float32(-8.0), np.float32(-6.0))
        dist = orc.set_distances(cells, mpc, helpers.CPM_DEFAULT, origin)
        g = bl.OccupancyGrid.from_cells(cells, origin, mpc, cellsPerMeter=helpers.CPM_DEFAULT, ctx=ctx)
        for radius in (0.1, 0.2):
            planner = bl.MotionPlanner(bl.MotionPlannerParams(radius), ctx=ctx); planner.setMap(g)
            comp, _ = ndimage.label(dist > np.float32(radius) * np.float32(1.000001))
            ys, xs = np.nonzero(dist > radius * 1.5)
            kept = 0
            for _ in range(4 * per_seed):
                if kept >= per_seed:
                    break
                a = int(rng.integers(0, xs.size))
                near = np.nonzero((comp[ys, xs] == comp[ys[a], xs[a]]) & (np.abs(xs - xs[a]) + np.abs(ys - ys[a]) <= 70))[0]
                b = int(near[rng.integers(0, near.size)])
                sp = (float(origin[0]) + (xs[a] + 0.5) * 0.05, float(origin[1]) + (ys[a] + 0.5) * 0.05)
                gp = (float(origin[0]) + (xs[b] + 0.5) * 0.05, float(origin[1]) + (ys[b] + 0.5) * 0.05)
                q = mp.Queue()
                p = mp.Process(target=gen.one, args=((seed, radius, sp, gp), q))
                p.start(); p.join(3.0)
                if p.is_alive():
                    p.terminate(); p.join()
                    continue
                pops, pushes, n = q.get()
                if pops > gen.MAX_POPS:
                    continue
                kept += 1
                exp, est = orc.search(orc.pose(*sp, 0.3), orc.pose(*gp, 0.0), dist, mpc, helpers.CPM_DEFAULT, origin, radius, 10.0 * radius, cap=1 << 16)
                path, stats = bl.search_for_path(bl.make_pose(*sp, 0.3), bl.make_pose(*gp, 0.0), planner.distances_, planner.searchParams_, return_stats=True)
                got = np.array([(p_.utime, p_.x, p_.y, p_.theta) for p_ in path], dtype=exp.dtype)
                ok = tuple(stats) == tuple(est) and got.tobytes() == exp.tobytes()
                done += 1; bad += not ok; pops_total += int(est[0]); biggest = max(biggest, int(est[0]))
                if not ok:
                    print("MISMATCH seed %d radius %.1f %s -> %s: got %s expected %s" % (seed, radius, sp, gp, tuple(stats), tuple(est)), flush=True)
        print("seed %d: %d searches so far, %d mismatches, %.2e pops, largest %.2e  (%.0f s)" % (seed, done, bad, pops_total, biggest, time.time() - t0), flush=True)
        g.close()
    print("RESULT: %d searches, %d mismatches, %.3e pops" % (done, bad, pops_total))
    return 1 if bad else 0


if __name__ == "__main__":
    mp.set_start_method("fork")
    sys.exit(main())
