"""The search's open list on the device (k_astar2's split-storage heap, botlab_amd/csrc/bl_astar2.h) against libstdc++'s
std::push_heap / std::pop_heap, which is what the reference's std::priority_queue<Node, vector, greater> runs
(src/planning/astar.hpp:41-44, astar.cpp:75-76,117-135): the order in which entries of EQUAL fCost leave the queue decides the
path, so every popped (key, payload) pair must match, on every storage tier (LDS keys / global keys, LDS / global payloads)."""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def make_ops(rng, n, spread, push_bias, new_best_every=7):
    """A search-like operation mix: keys near the current scale with many ties, now and then a new best (rises to the root)."""
    keys = np.empty(n, dtype=np.int32)
    pays = np.arange(n, dtype=np.uint32)
    base = 20000
    size = 0
    r = rng.random(n)
    k = rng.integers(0, 2 * spread + 1, size=n)
    nb = rng.integers(0, new_best_every, size=n)
    low = rng.integers(0, 50, size=n)
    for i in range(n):
        if size == 0 or r[i] < push_bias:
            f = base + int(k[i]) - spread // 2
            if nb[i] == 0:
                base = max(1000, base - int(low[i]) // 8)
                f = base - int(low[i])
            keys[i] = min(65534, max(1, f))
            size += 1
        else:
            keys[i] = -1
            size -= 1
    return keys, pays


def replay_both(gpu_ctx, oracle, keys, pays, cfg, cap):
    n = len(keys)
    ok = np.zeros(n, dtype=np.uint32); op = np.zeros(n, dtype=np.uint32)
    no = oracle.lib.orc_heap_replay(keys.ctypes.data, pays.ctypes.data, n, cap, ok.ctypes.data, op.ctypes.data)
    gk = np.zeros(n, dtype=np.uint32); gp = np.zeros(n, dtype=np.uint32)
    gn = C.c_int(0)
    cyc = (C.c_uint64 * 4)()
    rc = gpu_ctx.lib.bl_debug_heap2_replay(gpu_ctx.h, keys.ctypes.data, pays.ctypes.data, n, cfg, cap, gk.ctypes.data, gp.ctypes.data,
                                           C.byref(gn), cyc)
    assert rc == 0
    return (ok[:no], op[:no]), (gk[:gn.value], gp[:gn.value]), list(cyc)


@pytest.mark.parametrize("cfg,n,spread,bias", [
    (2, 4000, 3, 0.70), (2, 60000, 3, 0.62), (2, 60000, 40, 0.58), (2, 200000, 4000, 0.56), (2, 200000, 6, 0.55),
    (1, 120000, 5, 0.65), (1, 300000, 60, 0.58),
    (0, 60000, 5, 0.60), (0, 400000, 30, 0.70), (0, 6000, 4, 0.60), (1, 6000, 4, 0.55),
    (16 + 2, 60000, 3, 0.62), (16 + 0, 60000, 5, 0.60), (16 + 1, 6000, 4, 0.55),      # + 16: the general forms only
])
def test_heap_replay_matches_libstdcxx(gpu_ctx, oracle, cfg, n, spread, bias):
    rng = np.random.default_rng(1000 * (cfg & 15) + n + spread)
    keys, pays = make_ops(rng, n, spread, bias)
    ref, got, cyc = replay_both(gpu_ctx, oracle, keys, pays, cfg, 1 << 20)
    assert len(ref[0]) == len(got[0])
    bad = np.nonzero((ref[0] != got[0]) | (ref[1] != got[1]))[0]
    assert bad.size == 0, "first differing pop %d of %d: ref (%d, %d) device (%d, %d)" % (
        bad[0], len(ref[0]), ref[0][bad[0]], ref[1][bad[0]], got[0][bad[0]], got[1][bad[0]])
    if cyc[1] and cyc[3]:
        print("cfg %d n %d: %.0f cycles/push, %.0f cycles/pop" % (cfg, n, cyc[0] / cyc[1], cyc[2] / cyc[3]))


def test_heap_replay_drains_and_refills(gpu_ctx, oracle):
    """The heap shrinks back through every tier boundary and grows again (slots behind the heap must read as 'no entry')."""
    rng = np.random.default_rng(7)
    parts = []
    for phase, (n, bias) in enumerate([(30000, 0.9), (40000, 0.2), (30000, 0.8), (60000, 0.3), (5000, 0.6)]):
        k, _ = make_ops(rng, n, 4, bias)
        parts.append(k)
    keys = np.concatenate(parts)
    pays = np.arange(len(keys), dtype=np.uint32)
    ref, got, _ = replay_both(gpu_ctx, oracle, keys, pays, 2, 1 << 20)
    assert len(ref[0]) == len(got[0]) and np.array_equal(ref[0], got[0]) and np.array_equal(ref[1], got[1])


def test_heap_replay_capacity_is_respected(gpu_ctx, oracle):
    rng = np.random.default_rng(11)
    keys, pays = make_ops(rng, 20000, 4, 0.8)
    ref, got, _ = replay_both(gpu_ctx, oracle, keys, pays, 2, 5000)
    assert np.array_equal(ref[0], got[0]) and np.array_equal(ref[1], got[1])
