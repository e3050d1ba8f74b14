"""Timing probe for row f4: 254 scans x 290 beams on a 4096^2 world, GPU beam march vs the host-side numpy ray caster."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import botlab_amd as bl, helpers
from botlab_amd import sim, synth
maps = helpers.load_reference_maps()
S = 4096
world = synth.tile_world(maps["astar_maze"]["cells"], S)
half = S * 0.05 / 2
lid = sim.SimLidar(world, -half, -half, 0.05)
poses = synth.square_trajectory((0.3, 0.3, 0.0), 254, step_len=0.02, turn=0.05, side=0.5)
def pose_at(t):
    k = min(int((t - 99.9) * 10), 253)
    return tuple(poses[max(k, 0)])
nows = [100.0 + 0.1 * k for k in range(254)]
for rep in range(2):
    t0 = time.perf_counter(); out = lid.scans(pose_at, nows); t_gpu = time.perf_counter() - t0
t0 = time.perf_counter()
x = np.repeat([p[0] for p in poses[:254]], 290); y = np.repeat([p[1] for p in poses[:254]], 290)
ang = np.tile(np.linspace(-3.1, 3.1, 290), 254)
r = lid.cast(x, y, ang); t_cast = time.perf_counter() - t0
t0 = time.perf_counter()
for k in range(1, 40):
    synth.raycast_scan(world, (np.float32(-half), np.float32(-half)), 0.05, poses[k - 1], poses[k], 1000 + k)
t_np = (time.perf_counter() - t0) / 39 * 254
print(f"254 scans x 290 beams on {S}^2: GPU scans() {t_gpu*1e3:.1f} ms (host loop included), cast() alone {t_cast*1e3:.1f} ms; numpy ray caster {t_np*1e3:.0f} ms (extrapolated from 39 scans)")
