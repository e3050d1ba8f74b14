"""Multi-process CPU test of the particle sharding (botlab_amd/sharded.py): world_size 2 over gloo.  The orchestration
(shard bounds, in-place all-gather of the exchange record -- the only collective) must give every rank the
record a single-rank run produces, bit for bit, and the resampled poses must match the oracle's ParticleFilter."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import helpers
import oracle_lib
from botlab_amd import host, sharded, synth

HERE = os.path.dirname(os.path.abspath(__file__))
N, STEPS = 601, 6          # odd N: the last shard is shorter than the padded block


def _inputs():
    maps = helpers.load_reference_maps()
    m = maps["obstacle_slam_10mx10m_5cm"]
    truth = np.where(m["cells"] > 0, 127, -127).astype(np.int8)
    poses = synth.square_trajectory((-0.75, 0.2, 0.0), STEPS, step_len=0.03, turn=0.05, side=0.8)
    odo = synth.odometry_from_truth(poses, np.random.default_rng(5))
    scans = [synth.raycast_scan(truth, m["origin"], 0.05, poses[k - 1], poses[k], 1_000_000 + k * 100_000) for k in range(1, STEPS + 1)]
    rng = np.random.default_rng(11)
    noise = [rng.normal(0, 1, 3 * N).astype(np.float32) * np.tile(np.float32([0.05, 0.005, 0.05]), N) for _ in range(STEPS)]
    rands = [int(v) for v in rng.integers(0, 2**31 - 1, STEPS)]
    orc = oracle_lib.load_oracle()
    opf = oracle_lib.OraclePF(orc, N)
    opf.init_at_pose(orc.pose(-0.75, 0.2, 0.0, utime=int(scans[0].times[0])), 77)
    return m, odo, scans, noise, rands, opf.particles()


def _run(rank, world, port, out_dir):
    sys.path.insert(0, HERE)
    import cpu_shard_engine
    if world > 1:
        os.environ["MASTER_ADDR"] = "127.0.0.1"
        os.environ["MASTER_PORT"] = str(port)
        dist.init_process_group("gloo", rank=rank, world_size=world)
    m, odo, scans, noise, rands, parts = _inputs()
    eng = cpu_shard_engine.CpuShardEngine(N, rank, world, m["cells"], m["mpc"], helpers.CPM_DEFAULT, m["origin"])
    spf = sharded.ShardedParticleFilter(eng)
    spf.setParticles(parts)
    recs, poses = [], []
    for k in range(STEPS):
        # the action noise added to the means is formed inside the engine from rot1/trans/rot2; here noise carries the
        # already-sampled values, so add the means the oracle would: keep it simple and pass raw samples
        o = odo[k + 1]
        pose = spf.updateFilter(host.make_pose(o[0], o[1], o[2], utime=scans[k].utime), scans[k], None, rands[k], noise=noise[k])
        recs.append(eng.record())
        poses.append(pose)
    np.save(os.path.join(out_dir, f"rec_w{world}_r{rank}.npy"), np.stack(recs))
    np.save(os.path.join(out_dir, f"pose_w{world}_r{rank}.npy"), np.array([[0, 0, 0] if p is None else list(p) for p in poses], np.float32))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def test_shard_bounds():
    assert sharded.shard_bounds(100000, 0, 8) == (0, 12500, 12500)
    assert sharded.shard_bounds(601, 1, 2) == (301, 601, 301)
    assert sharded.shard_bounds(10, 3, 4) == (9, 10, 3)
    # the composed finish cuts at whole finish groups
    assert sharded.shard_bounds(1_000_000, 7, 8, sharded.COMPOSED_ALIGN) == (7 * 126976, 1_000_000, 126976)
    assert sharded.shard_bounds(30001, 2, 3, sharded.composed_align(30001)) == (20480, 30001, 10240)
    assert sharded.composed_align(1_000_000) == 2048 and sharded.composed_align(100_000) == 512
    assert sharded.composed_possible(100_000, 8) and sharded.shard_bounds(100_000, 7, 8, 512) == (89600, 100_000, 12800)
    assert sharded.composed_possible(1_000_000, 8) and sharded.composed_possible(30001, 3)
    assert sharded.composed_possible(601, 2) and not sharded.composed_possible(500, 2) and not sharded.composed_possible(100_000, 1)
    with pytest.raises(ValueError):
        sharded.shard_bounds(4, 5, 6)


@pytest.mark.parametrize("world", [2, 3])
def test_gloo_ranks_match_single_rank_and_oracle(tmp_path, world):
    out = str(tmp_path)
    _run(0, 1, 0, out)                                           # single-rank run in this process
    port = 29500 + (os.getpid() % 2000) + world
    mp.spawn(_run, args=(world, port, out), nprocs=world, join=True)     # `world` ranks over gloo
    one = np.load(os.path.join(out, "rec_w1_r0.npy"))
    for r in range(world):
        two = np.load(os.path.join(out, f"rec_w{world}_r{r}.npy"))
        assert one.view(np.uint32).tobytes() == two.view(np.uint32).tobytes(), f"rank {r} record differs from the single-rank run"
        p1, p2 = np.load(os.path.join(out, "pose_w1_r0.npy")), np.load(os.path.join(out, f"pose_w{world}_r{r}.npy"))
        assert p1.tobytes() == p2.tobytes()      # the estimate is formed from the gathered record: independent of the shard count

    # and the same sequence through the oracle's ParticleFilter consuming the same noise
    m, odo, scans, noise, rands, parts = _inputs()
    orc = oracle_lib.load_oracle()
    opf = oracle_lib.OraclePF(orc, N)
    opf.set_particles(parts)
    for k in range(STEPS):
        o = odo[k + 1]
        res = opf.update(orc.pose(o[0], o[1], o[2], utime=scans[k].utime), scans[k], m["cells"], m["mpc"], helpers.CPM_DEFAULT,
                         m["origin"], rands[k], noise_in=noise[k])
        if res["moved"]:
            exp = opf.particles()
            assert np.array_equal(one[k][:, 0], exp["x"]) and np.array_equal(one[k][:, 1], exp["y"]) and np.array_equal(one[k][:, 2], exp["theta"])
            units = one[k][:, 3].copy().view(np.uint32).astype(np.float64)
            assert np.allclose(units / units.sum(), exp["weight"], rtol=1e-9)
