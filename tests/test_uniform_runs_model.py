"""The equal-weight cumulative as runs (botlab_amd/csrc/bl_mcl_finish.h: uni_build / uni_search), restated line for line on Python floats
(IEEE doubles) and held against the plain loop of resamplePosteriorDistribution (particle_filter.cpp:84-103): c_0 = w, c_i = fl(c_{i-1} + w).
Weights that tie on every step of a binade (w = (q + 1/2) ulp there) are among the cases."""
import math
import struct

import numpy as np
import pytest


def _exp(v):
    return (struct.unpack("<q", struct.pack("<d", v))[0] >> 52) & 0x7FF


def uni_build(w, N):
    segs, i, c = [], 0, w
    while True:
        if i >= N - 1:
            segs.append((i, c, 0.0, 1))
            break
        c1 = c + w
        e = _exp(c)
        run = _exp(c1) == e and i + 2 < N
        if run:
            c2 = c1 + w
            run = _exp(c2) == e
        if not run:
            segs.append((i, c, 0.0, 1)); i += 1; c = c1
            continue
        inc = c2 - c1
        u = math.ldexp(1.0, e - 1023 - 52)
        M1, Q = int(c1 / u), int(inc / u)
        nrun = ((1 << 53) - 2 - M1) // Q + 1 if Q > 0 else 1
        nrun = max(nrun, 1)
        if i + 1 + nrun > N:
            nrun = N - (i + 1)
        segs.append((i, c, 0.0, 1))
        segs.append((i + 1, c1, inc, nrun))
        tail = c1 + float(nrun - 1) * inc
        i = i + 1 + nrun
        if i >= N:
            c = tail
            break
        c = tail + w
    return segs, c


def expand(segs, N):
    out = np.empty(N)
    for i0, c0, inc, n in segs:
        out[i0:i0 + n] = c0 + np.arange(n, dtype=np.float64) * inc
    return out


def plain(w, N):
    out = np.empty(N)
    c = w
    out[0] = c
    for i in range(1, N):
        c = c + w
        out[i] = c
    return out


def uni_search(segs, T, N):
    lo, hi = 0, len(segs) - 1
    last = [c0 + float(n - 1) * inc for _, c0, inc, n in segs]
    while lo < hi:
        mid = (lo + hi) >> 1
        if last[mid] >= T:
            hi = mid
        else:
            lo = mid + 1
    i0, c0, inc, n = segs[lo]
    if not last[lo] >= T:
        return N - 1
    if T <= c0 or n <= 1:
        return i0
    k = int((T - c0) / inc)
    k = min(max(k, 0), n - 1)
    while k < n - 1 and c0 + float(k) * inc < T:
        k += 1
    while k > 0 and c0 + float(k - 1) * inc >= T:
        k -= 1
    return i0 + k


CASES = [(1.0 / 200, 200), (0.001 / 0.20000000000000015, 200), (1.0 / 2, 2), (1.0 / 12345, 12345), (1.0 / 4096, 4096), (1.0 / 100000, 100000), (1.0 / 300000, 300000), (1.0 / 1000, 1000), (1.0 / 7, 7), (1.0 / 3, 3), (0.5, 2), (1.0, 1),
         (0.001, 100000), (0.001, 4096), (0.001 / 100.00000000000001, 100000), (2.0 ** -17, 1 << 17), (3.0 * 2.0 ** -19, 150000),
         # ties: w = (q + 1/2) ulp of the binade the sum runs through -- 1 + 2^-52 is (2^51 + 1/2) ulps of [2, 4)
         (1.0 + 2.0 ** -52, 5000), (1.0 + 3 * 2.0 ** -52, 5000), (2.0 ** -10 * (1.0 + 2.0 ** -52), 200000), (2.0 ** -10 * (1.0 + 7 * 2.0 ** -52), 200000)]


@pytest.mark.parametrize("w,N", CASES)
def test_runs_equal_the_plain_loop(w, N):
    segs, last = uni_build(w, N)
    want = plain(w, N)
    got = expand(segs, N)
    assert np.array_equal(got.view(np.uint64), want.view(np.uint64))
    assert last == want[-1]
    assert len(segs) <= 192
    assert sum(n for _, _, _, n in segs) == N and all(a[0] + a[3] == b[0] for a, b in zip(segs, segs[1:]))


def test_random_weights_and_sizes():
    rng = np.random.default_rng(5)
    for _ in range(60):
        N = int(rng.integers(1, 60000))
        w = float(rng.random() * 2.0 ** float(-rng.integers(0, 40)))
        if w <= 0:
            continue
        segs, _ = uni_build(w, N)
        assert np.array_equal(expand(segs, N).view(np.uint64), plain(w, N).view(np.uint64)), (w, N)


@pytest.mark.parametrize("N", [4096, 100000])
def test_search_picks_the_references_index(N):
    RAND_MAX = 2147483647
    for w in (1.0 / N, 0.001 / plain(0.001, N)[-1]):
        segs, _ = uni_build(w, N)
        c = plain(w, N)
        for rv in (0, 1, 1000, 1 << 30, RAND_MAX, 1804289383):
            r = (rv / RAND_MAX) * (1.0 / N)
            U = r + np.arange(N) * (1.0 / N)
            want = np.minimum(np.searchsorted(c, U, side="left"), N - 1)          # first i with U <= c_i
            ms = list(range(0, N, max(1, N // 997))) + [N - 1, N - 2]
            for m in ms:
                assert uni_search(segs, float(U[m]), N) == want[m], (w, rv, m)
