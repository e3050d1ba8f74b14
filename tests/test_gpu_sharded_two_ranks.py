"""The real sharded path (HipShardEngine kernels + ShardedParticleFilter collectives) with TWO ranks.  The test box has
one GPU, so both ranks share cuda:0 and the collectives go over gloo (which stages device tensors through the host);
the kernels, the shard bounds, the in-place all-gather layout and the estimate-from-the-record finish are exactly what runs under RCCL.
2 ranks must reproduce the 1-rank particle set bit for bit (exact integer weights, Philox keyed by global index)."""
import os
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))
N, STEPS = 30001, 6           # odd N: last shard shorter than the padded block


def _run(rank, world, port, out_dir, riding=False, composed=False, n=N, peer=True):
    sys.path.insert(0, HERE)
    sys.path.insert(0, os.path.dirname(HERE))
    import torch
    import torch.distributed as dist
    import helpers
    import botlab_amd as bl
    from botlab_amd import sharded, synth
    if world > 1:
        os.environ["MASTER_ADDR"] = "127.0.0.1"
        os.environ["MASTER_PORT"] = str(port)
        dist.init_process_group("gloo", rank=rank, world_size=world)
    m = helpers.load_reference_maps()["obstacle_slam_10mx10m_5cm"]
    truth = np.where(m["cells"] > 0, 127, -127).astype(np.int8)
    poses = synth.square_trajectory((-0.75, 0.2, 0.0), STEPS, step_len=0.03, turn=0.05, side=0.8)
    scans = [synth.raycast_scan(truth, m["origin"], 0.05, poses[k - 1], poses[k], 1_000_000 + k * 100_000) for k in range(1, STEPS + 1)]
    # one rank would take the fused single-shard finish (sums from k_mcl_main's workgroup partials, another addition
    # order); the scan-based finish is the one every shard count shares, so the comparison below can be bit-exact
    # (riding=True with two ranks: the shards' own default -- the tile sums from the record, the rest of the finish riding in the
    # map kernel -- against the same record-based finish as separate launches on one rank)
    if not (riding and world > 1):
        os.environ["BOTLAB_MCL_NO_FUSED_FINISH"] = "1"
    if not peer:
        os.environ["BOTLAB_SHARD_NO_PEER_STORES"] = "1"
    if composed and world > 1:
        # the composed finish: ranks keep their own block, map each other's memory (IPC), exchange two small
        # all-gathers.  First the question bench.py asks: can the ranks map each other's memory at all?
        assert sharded.composed_possible(n, world)
        probe = bl.Context(0)
        assert sharded.ipc_probe(probe, rank, world), "hipIpc between the ranks' processes does not work here"
        probe.close()
    eng = sharded.HipShardEngine(n, rank, world, 0, composed=composed and world > 1)            # every rank on device 0
    spf = sharded.ShardedParticleFilter(eng)
    grid = bl.OccupancyGrid.from_cells(m["cells"], m["origin"], m["mpc"], cellsPerMeter=helpers.CPM_DEFAULT, ctx=eng.ctx)
    mapper = bl.Mapping(5.0, 4, 1, ctx=eng.ctx)
    spf.initializeFilterAtPose(bl.make_pose(-0.75, 0.2, 0.0, utime=int(scans[0].times[0])), seed=21)
    est = []
    for k, sc in enumerate(scans):
        odo = bl.make_pose(*poses[k + 1], utime=sc.utime)
        if riding and world > 1:
            spf.updateBegin(odo, sc, grid, 900 + k)
            mapper.updateMapFinishingFilter(sc, eng.pf, sc.utime, grid)
            p = eng.pf.poseEstimate()
        else:
            p = spf.updateFilter(odo, sc, grid, 900 + k)
            if riding:
                mapper.updateMapDevicePose(sc, eng.pf.poseDevicePtr(), sc.utime, grid)
        est.append((p.utime, p.x, p.y, p.theta))
    parts = spf.particles()
    if composed and world > 1:
        assert spf.composed
        sent, received, own = spf.exchange_bytes_per_update()
        # the point of the composed form: per rank and update O(block) + O(100 KB), not N x 16 B
        per_peer = 90_000 + (eng.S // 128) * 42                                      # 32 B of records + 10 B of sums per 128 particles + tables
        if spf.peer:                 # no collective: a rank stores its slice and block once per other rank
            assert sent == received and received <= (world - 1) * per_peer
        else:
            assert sent <= per_peer and received == (world - 1) * sent
        with open(os.path.join(out_dir, f"form_w{world}_r{rank}.txt"), "w") as f:
            f.write("peer" if spf.peer else "collective")
        with open(os.path.join(out_dir, f"traffic_w{world}_r{rank}.txt"), "w") as f:
            f.write(f"{sent} {received} {own} {n * 16}")
    np.save(os.path.join(out_dir, f"grid_w{world}_r{rank}.npy"), grid.cells())
    np.save(os.path.join(out_dir, f"parts_w{world}_r{rank}.npy"), parts)
    np.save(os.path.join(out_dir, f"est_w{world}_r{rank}.npy"), np.array(est, dtype=np.float64))
    with open(os.path.join(out_dir, f"shard_w{world}_r{rank}.txt"), "w") as f:
        f.write(f"{eng.lo} {eng.hi}")
    if world > 1:
        spf.close()
        dist.barrier()
        dist.destroy_process_group()


@pytest.mark.parametrize("world,riding,n,peer", [(2, False, N, True), (2, True, N, True), (3, True, N, False), (2, True, 200_001, True), (4, True, 3500, True),   # (3500 over four ranks: blocks of 1024, the small groups' alignment)
                                                 (2, True, N, False), (6, True, 100_000, True)])
def test_composed_finish_matches_single_rank(tmp_path, world, riding, n, peer):
    """The composed finish -- own blocks only, sources read from their owners' memory, two small all-gathers -- with 2 and 3 ranks
    on one device (one process per rank, IPC mappings, collectives over gloo): particles, estimates and the replicated map equal
    the single rank's bit for bit; 200 001 particles take the large finish groups (1024 threads, 2048 particles)."""
    import torch.multiprocessing as mp
    out = str(tmp_path)
    mp.spawn(_run, args=(1, 0, out, riding, False, n), nprocs=1, join=True)
    mp.spawn(_run, args=(world, 29900 + os.getpid() % 300 + 7 * world + (300 if riding else 0), out, riding, True, n, peer), nprocs=world, join=True)
    one = np.load(os.path.join(out, "parts_w1_r0.npy"))
    e1 = np.load(os.path.join(out, "est_w1_r0.npy"))
    g1 = np.load(os.path.join(out, "grid_w1_r0.npy"))
    got = []
    for r in range(world):
        lo, hi = map(int, open(os.path.join(out, f"shard_w{world}_r{r}.txt")).read().split())
        from botlab_amd import sharded
        assert lo % sharded.composed_align(n) == 0
        part = np.load(os.path.join(out, f"parts_w{world}_r{r}.npy"))
        assert part.size == hi - lo
        got.append(part)
        assert e1.tobytes() == np.load(os.path.join(out, f"est_w{world}_r{r}.npy")).tobytes(), f"rank {r}: estimates differ"
        assert np.array_equal(g1, np.load(os.path.join(out, f"grid_w{world}_r{r}.npy"))), f"rank {r}: map differs"
        sent, received, own, full = map(int, open(os.path.join(out, f"traffic_w{world}_r{r}.txt")).read().split())
        assert received < full or n < 100_000               # less than the replicated form's N x 16 B once N is large
        assert open(os.path.join(out, f"form_w{world}_r{r}.txt")).read() == ("peer" if peer else "collective")
    assert np.concatenate(got).tobytes() == one.tobytes()


@pytest.mark.parametrize("riding", [False, True])
def test_two_ranks_one_device_match_single_rank(tmp_path, riding):
    import torch.multiprocessing as mp
    out = str(tmp_path)
    mp.spawn(_run, args=(1, 0, out, riding), nprocs=1, join=True)
    mp.spawn(_run, args=(2, 29600 + os.getpid() % 300 + (300 if riding else 0), out, riding), nprocs=2, join=True)   # replicated form
    one = np.load(os.path.join(out, "parts_w1_r0.npy"))
    got = []
    for r in range(2):
        lo, hi = map(int, open(os.path.join(out, f"shard_w2_r{r}.txt")).read().split())
        part = np.load(os.path.join(out, f"parts_w2_r{r}.npy"))
        assert part.size == hi - lo
        got.append(part)
        e1, e2 = np.load(os.path.join(out, "est_w1_r0.npy")), np.load(os.path.join(out, f"est_w2_r{r}.npy"))
        assert e1.tobytes() == e2.tobytes()      # pose estimates bit-identical: formed from the gathered record in an order fixed by N
    two = np.concatenate(got)
    assert two.tobytes() == one.tobytes()
    g1 = np.load(os.path.join(out, "grid_w1_r0.npy"))
    for r in range(2):
        assert np.array_equal(g1, np.load(os.path.join(out, f"grid_w2_r{r}.npy")))      # the replicated map stays identical
