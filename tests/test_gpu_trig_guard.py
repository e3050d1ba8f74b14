"""The guard band of the sensor model's fast trigonometry, checked on the hardware the tests run on.

SensorModel::scoreRay (src/slam/sensor_model.cpp:34-38) truncates range * cosf / sinf(theta') * cellsPerMeter + start to the
cell a ray ends in.  k_mcl_main takes v_sin_f32 / v_cos_f32 of the unwrapped angle instead and falls back to the exact
polynomial for rays whose endpoint lies within a band of a cell boundary; the band is derived from MCL_TRIG_EPS, a bound on the
hardware pair's distance from the reference's values.  That bound is a MEASUREMENT: this test repeats it (exhaustively, every
float of the angle's range, through the very device function the ray loop calls), keeps the figures under gpurun_out/, and
checks that switching the fast path off changes nothing."""
import ctypes as C
import json
import os

import numpy as np
import pytest

import helpers
import oracle_lib
import botlab_amd as bl
from botlab_amd import synth
from botlab_amd._capi import check

pytestmark = pytest.mark.gpu


def test_hardware_trig_stays_inside_the_guard_band(gpu_ctx):
    ms, mc, eps, n = C.c_float(), C.c_float(), C.c_float(), C.c_uint64()
    check(gpu_ctx.lib.bl_debug_trig_probe(gpu_ctx.h, C.byref(ms), C.byref(mc), C.byref(eps), C.byref(n)))
    report = dict(floats_checked=int(n.value), max_sin_err=float(ms.value), max_cos_err=float(mc.value), eps_used=float(eps.value),
                  margin=float(eps.value) / max(float(ms.value), float(mc.value)) - 1.0)
    os.makedirs("gpurun_out", exist_ok=True)
    with open(os.path.join("gpurun_out", "trig_probe.json"), "w") as fh:
        json.dump(report, fh)
    assert n.value > 2_100_000_000                          # every float of [-3 pi - 0.01, pi + 0.01]
    assert 0.0 < ms.value and 0.0 < mc.value                # (a probe that measured nothing would pass every bound)
    # both maxima at least 5 % below the constant the band is built from
    assert ms.value * 1.05 <= eps.value and mc.value * 1.05 <= eps.value, report


def test_trig_by_addition_stays_inside_the_guard_band(gpu_ctx):
    """The default form of the fast path: the ray's direction from the particle's and the ray's (cos, sin) pairs by the addition
    theorems (bl_mcl.hip, ray_cells_fast), with the particle's pair carried scaled by the cells per metre as the ray loop carries it
    (round 6).  Its distance from the reference's sinf / cosf is bounded analytically by 8.8e-7 (8.2e-7 unscaled); this
    measures it over 4e9 random pairs with the functions the ray loop calls (the committed figure, profiles/r04_trig_addition_probe.json,
    is from 1e10) and asserts the same constant the band is built from, with at least 25 % to spare."""
    ms, mc, eps, n = C.c_float(), C.c_float(), C.c_float(), C.c_uint64()
    check(gpu_ctx.lib.bl_debug_trig_addition_probe(gpu_ctx.h, 4_000_000_000, 20261004, C.byref(ms), C.byref(mc), C.byref(eps), C.byref(n)))
    report = dict(pairs_checked=int(n.value), max_sin_err=float(ms.value), max_cos_err=float(mc.value), eps_used=float(eps.value),
                  analytic_bound=8.8e-7, margin=float(eps.value) / max(float(ms.value), float(mc.value)) - 1.0)
    os.makedirs("gpurun_out", exist_ok=True)
    with open(os.path.join("gpurun_out", "trig_addition_probe.json"), "w") as fh:
        json.dump(report, fh)
    assert n.value >= 4_000_000_000
    assert 0.0 < ms.value and 0.0 < mc.value
    assert ms.value <= 8.8e-7 and mc.value <= 8.8e-7, report          # the derived bound holds for every pair seen
    assert ms.value * 1.25 <= eps.value and mc.value * 1.25 <= eps.value, report


@pytest.mark.parametrize("form", ["addition", "hardware"])
def test_fast_trig_on_and_off_give_identical_likelihoods(oracle, maps, gpu_ctx, monkeypatch, form):
    """100 000 particles x 3 updates on the shipped obstacle_slam map, Philox noise: the fast path with guard band -- the default
    form (direction by the addition theorems) and the hardware sine / cosine form (BOTLAB_MCL_HW_TRIG: read once per process, so
    this case runs in a child) -- and BOTLAB_MCL_NO_FAST_TRIG (exact sinf / cosf for every ray) must agree in every likelihood,
    every resampling index, every particle and the estimate -- 5.8e7 particle-rays per run."""
    if form == "hardware":
        import subprocess, sys
        env = dict(os.environ, BOTLAB_MCL_HW_TRIG="1")
        r = subprocess.run([sys.executable, "-m", "pytest", "-x", "-q", "-m", "gpu", os.path.abspath(__file__), "-k", "identical_likelihoods and addition"],
                           env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=600)
        assert r.returncode == 0, r.stdout.decode()[-2000:]
        return
    N = 100_000
    m = maps["obstacle_slam_10mx10m_5cm"]
    truth = np.where(m["cells"] > 0, 127, -127).astype(np.int8)
    poses = synth.square_trajectory((-0.75, 0.2, 0.0), 4, step_len=0.02, turn=0.05, side=0.8)
    scans = [synth.raycast_scan(truth, m["origin"], 0.05, poses[k - 1], poses[k], 1_000_000 + k * 100_000) for k in range(1, 5)]

    def run(no_fast):
        if no_fast:
            monkeypatch.setenv("BOTLAB_MCL_NO_FAST_TRIG", "1")          # read when the filter is created
        else:
            monkeypatch.delenv("BOTLAB_MCL_NO_FAST_TRIG", raising=False)
        g = bl.OccupancyGrid.from_cells(m["cells"], m["origin"], m["mpc"], cellsPerMeter=helpers.CPM_DEFAULT, ctx=gpu_ctx)
        pf = bl.ParticleFilter(N, ctx=gpu_ctx)
        pf.debugEnable(True)
        pf.initializeFilterAtPose(bl.make_pose(-0.75, 0.2, 0.0, utime=int(scans[0].times[0])), seed=21)
        pf.setNoiseSeed(99)
        out = []
        for k, sc in enumerate(scans):
            est = pf.updateFilter(bl.make_pose(*poses[k + 1], utime=sc.utime), sc, g, rand_value=1804289383 + k)
            if k == 0:
                continue
            idx, like = pf.debugLast()
            out.append((idx.copy(), like.copy(), pf.particles().copy(), (np.float32(est.x), np.float32(est.y), np.float32(est.theta))))
        pf.close(); g.close()
        return out

    a, b = run(False), run(True)
    assert len(a) == len(b) == 3
    for k, (x, y) in enumerate(zip(a, b)):
        assert np.array_equal(x[1], y[1]), f"likelihoods differ at update {k}"
        assert np.array_equal(x[0], y[0]), f"indices differ at update {k}"
        assert x[2].tobytes() == y[2].tobytes(), f"particles differ at update {k}"
        assert x[3] == y[3]
        assert x[1].max() > 0
