"""Large-N parity measurement against the CPU oracle (minutes of CPU time; not part of the test suite).
Reports, per moved update: resample-index disagreements, likelihood mismatches, max relative weight error, max pose
difference, and the pose-estimate difference (the reference accumulates x, y in a float; the kernel reduces in double).
Usage: python tests/tools/large_n_parity.py N steps"""
import sys, time
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
import numpy as np
import botlab_amd as bl, helpers, oracle_lib
from botlab_amd import synth
N = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
STEPS = int(sys.argv[2]) if len(sys.argv) > 2 else 5
orc = oracle_lib.load_oracle()
m = helpers.load_reference_maps()["obstacle_slam_10mx10m_5cm"]
truth = np.where(m["cells"] > 0, 127, -127).astype(np.int8)
cpm = helpers.CPM_DEFAULT
ctx = bl.default_context()
g = bl.OccupancyGrid.from_cells(m["cells"], m["origin"], m["mpc"], cellsPerMeter=cpm, ctx=ctx)
poses = synth.square_trajectory((-0.75, 0.2, 0.0), STEPS, step_len=0.02, turn=0.05, side=0.8)
scans = [synth.raycast_scan(truth, m["origin"], 0.05, poses[k - 1], poses[k], 1_000_000 + k * 100_000) for k in range(1, STEPS + 1)]
opf = oracle_lib.OraclePF(orc, N)
opf.init_at_pose(orc.pose(-0.75, 0.2, 0.0, utime=int(scans[0].times[0])), 5)
pf = bl.ParticleFilter(N, ctx=ctx)
pf.setParticles(opf.particles())
pf.debugEnable(True)
rng = np.random.default_rng(1)
for k, sc in enumerate(scans):
    o = poses[k + 1]
    rv = int(rng.integers(0, 2**31 - 1))
    t0 = time.time()
    res = opf.update(orc.pose(*o, utime=sc.utime), sc, m["cells"], m["mpc"], cpm, m["origin"], rv)
    t1 = time.time()
    pose = pf.updateFilter(bl.make_pose(*o, utime=sc.utime), sc, g, rand_value=rv, noise=res["noise"])
    if not res["moved"]:
        continue
    idx, like = pf.debugLast()
    bad_idx = int((idx != res["idx"]).sum())
    adj = int((np.abs(idx - res["idx"]) > 1).sum())
    got, exp = pf.particles(), opf.particles()
    same_src = idx == res["idx"]
    bad_like = int((like[same_src] * 0.5 != res["raw"][same_src]).sum())
    pose_bits = int(((got["x"] != exp["x"]) | (got["y"] != exp["y"]) | (got["theta"] != exp["theta"]))[same_src].sum())
    wrel = float(np.max(np.abs(got["weight"][same_src] - exp["weight"][same_src]) / exp["weight"][same_src]))
    print(f"N={N} step {k}: oracle {t1 - t0:.1f}s | resample idx differ {bad_idx} (non-adjacent {adj}) | likelihood differ {bad_like} | "
          f"pose bits differ {pose_bits} | max rel weight err {wrel:.2e} | estimate dx {pose.x - res['pose'].x:+.3e} dy {pose.y - res['pose'].y:+.3e} "
          f"dtheta {pose.theta - res['pose'].theta:+.3e} (rel x {abs(pose.x - res['pose'].x) / abs(res['pose'].x):.2e})")
    if bad_idx:
        # keep the two filters in lock-step for the next update
        pf.setParticles(exp)
        pf.debugEnable(True)
