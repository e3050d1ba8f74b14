// bl_serial_sum.h -- the reference's SERIALLY ROUNDED float accumulator, reproduced bit for bit by parallel code.
//
// estimatePosteriorPose (src/slam/particle_filter.cpp:144-160, lines 151-152) forms
//       pose.x += p.weight * p.pose.x;          float += double * float
// i.e.  acc <- fl32( fl64( (double)acc + t_i ) ),  t_i = fl64(w_i * x_i),  over the particles in order, from acc = 0.
// Every step rounds, so the result depends on the order and no reduction tree reproduces it.  What makes it parallel:
//
//   While acc stays inside one binade, |acc| in [2^e, 2^(e+1)), write acc = s * M * u with u = 2^(e-23) (acc's ulp),
//   M an integer in [2^23, 2^24), s = +-1.  A term with |t| < 2^(e-1) then advances M by an INTEGER that does not depend on M:
//       fl64(acc + t) = acc + t',   t' = t rounded to a multiple of v = 2^(e-52)      (acc is a multiple of v; the tie rule
//                                                                                      sees an even acc / v)
//       fl32(acc + t') = s * (M + d) * u,   d = round(s * t' / u)                     (unless s*t'/u ends in exactly .5)
//   as long as the result stays strictly inside the binade (2^23 < M + d < 2^24).  So inside a binade the accumulator is an
//   integer prefix sum of the d_i; ss_quantize() forms d_i (and says "bad" for a tie or an oversized term, which must be
//   stepped exactly).
//
//   A run of terms is summarised for a PREDICTED binade by ss_rec = (key, D, lo, hi): D = sum of the d_i, lo / hi = the
//   smallest / largest inclusive prefix.  If the accumulator really arrives in that binade with magnitude M and
//   2^23 < M + lo, M + hi < 2^24, every step of the run was an in-binade step and the run leaves M + D: a record is a
//   function that is only ever APPLIED after this check (ss_rec_fits), so a wrong prediction costs time, never correctness.
//   Records of consecutive runs compose associatively (ss_rec_join).  Wherever a record does not fit -- the accumulator
//   crosses a binade (about log2 N times per sum), a tie, a change of sign, a term as large as the sum -- the run is replayed
//   from the true accumulator (ss_replay): in-binade prefix sums up to the first step that leaves, that one step in real
//   arithmetic (ss_exact_step), and on.
//
// The scalar core below is shared by the kernels (bl_mcl_finish.h) and by a CPU model of the whole scheme
// (tests/cpp/serial_sum_model.cpp), which checks it against the plain loop on adversarial sequences without a GPU.
#ifndef BL_SERIAL_SUM_H
#define BL_SERIAL_SUM_H

#include <math.h>
#include <stdint.h>
#include <string.h>

#if defined(__HIPCC__)
#define SS_HD __host__ __device__ __forceinline__
#else
#define SS_HD inline
#endif

#define SS_MLO (1 << 23)
#define SS_MHI (1 << 24)
#define SS_SAT (1 << 29)                          // |D|, |lo|, |hi| saturate here: sums of two stay inside int, and anything
                                                  // this large is out of every binade's range anyway

#define SS_ID 0x7fffffff                          // key of the empty run's record: fits everything, joins to the other side
struct ss_rec { int key, D, lo, hi; };            // key 0: never fits

SS_HD uint32_t ss_f2u(float f) { uint32_t u; memcpy(&u, &f, 4); return u; }
SS_HD float ss_u2f(uint32_t u) { float f; memcpy(&f, &u, 4); return f; }

// one step of the reference's loop
SS_HD float ss_exact_step(float acc, double t) { return (float)((double)acc + t); }

// Binade key of an accumulator: 0 if it has none that the integer form can use (zero, subnormal or tiny, inf, nan),
// else 0x400 | sign << 9 | biased exponent.  Magnitude M = the 24-bit significand.
SS_HD int ss_key(float acc)
{
    const uint32_t b = ss_f2u(acc);
    const int ex = (int)((b >> 23) & 0xff);
    if (ex < 16 || ex == 255) return 0;
    return 0x400 | (int)((b >> 31) << 9) | ex;
}
SS_HD int ss_mag(float acc) { return (int)((ss_f2u(acc) & 0x7fffffu) | 0x800000u); }
SS_HD float ss_from(int key, int M)                // 2^23 <= M <= 2^24 (M = 2^24 is the next binade's first value)
{
    const int ex = key & 0xff;
    const float mag = ldexpf((float)M, ex - 127 - 23);            // exact: M has at most 25 significant bits only when it is 2^24
    return (key & 0x200) ? -mag : mag;
}

// What the in-binade step needs of the binade: computed once per key.
struct ss_bin { double lim, C; int down; int neg; };
SS_HD ss_bin ss_bin_of(int key)
{
    const int e = (key & 0xff) - 127;
    ss_bin b;
    b.lim = ldexp(1.0, e - 1);          // terms must be smaller than a quarter of the binade's upper end
    b.C = ldexp(1.5, e);                // middle of the binade: C + t keeps the binade's double ulp, C / v is even like acc / v
    b.down = 23 - e;                    // scaling to units of u
    b.neg = (key & 0x200) ? 1 : 0;
    return b;
}

// d and bad for a term t while the accumulator is in binade b.  Branch-free: an oversized (or nan) term is quantized as 0
// and flagged.
SS_HD int ss_quantize(const ss_bin& b, double t, int* bad)
{
    const double ta = b.neg ? -t : t;
    const bool big = !(fabs(ta) < b.lim);                       // also catches nan
    const double tm = big ? 0.0 : ta;
    const double tp = (tm + b.C) - b.C;                         // tm rounded to a multiple of 2^(e-52), ties to even
    const double q = ldexp(tp, b.down);                         // exact, |q| < 2^22, a multiple of 2^-29
    const double f = floor(q);
    const double r = q - f;                                     // exact
    *bad |= (big || r == 0.5) ? 1 : 0;
    return (int)f + (r > 0.5 ? 1 : 0);
}
SS_HD int ss_quantize(int key, double t, int* bad) { return ss_quantize(ss_bin_of(key), t, bad); }

SS_HD int ss_sat(long long v) { return v > SS_SAT ? SS_SAT : (v < -SS_SAT ? -SS_SAT : (int)v); }
SS_HD int ss_sat_i(int v) { return v > SS_SAT ? SS_SAT : (v < -SS_SAT ? -SS_SAT : v); }

SS_HD ss_rec ss_rec_make(int key, int D, int lo, int hi) { ss_rec r; r.key = key; r.D = D; r.lo = lo; r.hi = hi; return r; }
SS_HD ss_rec ss_rec_identity() { return ss_rec_make(SS_ID, 0, 0, 0); }

// a, then b (branch-free)
SS_HD ss_rec ss_rec_join(const ss_rec& a, const ss_rec& b)
{
    const bool ida = a.key == SS_ID, idb = b.key == SS_ID;
    const int D = ss_sat_i(a.D + b.D), blo = ss_sat_i(a.D + b.lo), bhi = ss_sat_i(a.D + b.hi);
    ss_rec r;
    r.key = ida ? b.key : (idb ? a.key : (a.key == b.key ? a.key : 0));
    r.D = ida ? b.D : (idb ? a.D : D);
    r.lo = ida ? b.lo : (idb ? a.lo : (a.lo < blo ? a.lo : blo));
    r.hi = ida ? b.hi : (idb ? a.hi : (a.hi > bhi ? a.hi : bhi));
    return r;
}

// may the record be applied to an accumulator with this key and magnitude?
SS_HD bool ss_rec_fits(const ss_rec& r, int key, int M)
{
    const bool in = r.key != 0 && r.key == key && M + r.lo > SS_MLO && M + r.hi < SS_MHI;
    return r.key == SS_ID || in;
}

// ---------------------------------------------------------------------------------------------------------------------------
// Runs that CROSS binades ("wild" runs): a sum that hovers around zero changes its binade -- and its sign -- every few terms, and
// a record for one binade never fits.  But the binade SEQUENCE is predictable (a double-precision prefix sum of the terms gives
// every intermediate value to ~1e-9 of its size), and for a given sequence the run is again a function of the incoming integer
// alone.  With m = +-M the signed magnitude in units of the current binade's ulp u, a step into a binade whose ulp is u' = u 2^k
//       m' = round((m u + t'') / u'),     t'' = t rounded to a multiple of v' = 2^(e' - 52)  (what fl64(acc + t) keeps of t)
// is, with T = t'' / v' (an integer) and u' = 2^29 v':
//       k >= 0:  m' = floor((m 2^(29-k) + T + 2^28) / 2^29) = (m + ((T + 2^28) >> (29 - k))) >> k          (nested floors)
//       k <  0:  m' = (m << -k) + ((T + 2^28) >> 29)
// (round half up differs from the hardware's round half even only on exact ties, which are flagged and never applied).  Maps
//       G(m) = (((m + a) >> q) << r) + c
// are closed under composition (ssw_join), so a run is summarised by ONE of them plus the interval [L, H] of inputs for which
// every intermediate value stays strictly inside its predicted binade (each condition is an interval: the maps are monotone).
// A wild record is only ever applied to an accumulator of its input binade whose magnitude lies in [L, H]: a wrong prediction
// costs time (the run is replayed), never correctness.  tests/cpp/serial_sum_model.cpp checks all of it against the plain loop.
#define SSW_INF (1ll << 60)
#define SSW_MAXSHIFT 30
struct ss_wild { int key_in, key_out, q, r; long long a, c, L, H; };      // key_in 0: applies to nothing

SS_HD long long ssw_signed(int key, int M) { return (key & 0x200) ? -(long long)M : (long long)M; }
SS_HD long long ssw_apply(const ss_wild& w, long long m) { return (long long)((unsigned long long)((m + w.a) >> w.q) << w.r) + w.c; }
// the float of binade `key` with magnitude M, 2^23 <= M < 2^24, put together from its bits
SS_HD float ss_from_bits(int key, int M) { return ss_u2f(((uint32_t)(key & 0x200) << 22) | ((uint32_t)(key & 0xff) << 23) | ((uint32_t)M & 0x7fffffu)); }
SS_HD bool ssw_fits(const ss_wild& w, int key, long long m) { return w.key_in != 0 && w.key_in == key && m >= w.L && m <= w.H; }
SS_HD ss_wild ssw_invalid() { ss_wild w; w.key_in = 0; w.key_out = 0; w.q = w.r = 0; w.a = w.c = 0; w.L = 1; w.H = 0; return w; }
SS_HD ss_wild ssw_identity(int key) { ss_wild w = ssw_invalid(); w.key_in = w.key_out = key; w.L = -SSW_INF; w.H = SSW_INF; return w; }

// inputs m with lo <= G(m) <= hi for G = (a, q, r, c); empty: L > H
SS_HD void ssw_preimage(long long a, int q, int r, long long c, long long lo, long long hi, long long* L, long long* H)
{
    *L = -SSW_INF; *H = SSW_INF;
    if (lo > -SSW_INF / 2) { const long long alpha = -((-(lo - c)) >> r); *L = alpha * (1ll << q) - a; }          // ceil((lo - c) / 2^r)
    if (hi < SSW_INF / 2) { const long long beta = (hi - c) >> r; *H = (beta + 1) * (1ll << q) - 1 - a; }         // floor((hi - c) / 2^r)
}

// One step: an accumulator in binade key0 takes the term t and is predicted to land in binade key1.
SS_HD ss_wild ssw_step(int key0, int key1, double t)
{
    ss_wild w = ssw_invalid();
    if (!key0 || !key1) return w;
    const int e1 = (key1 & 0xff) - 127, k = (key1 & 0xff) - (key0 & 0xff);
    if (k > 28 || k < -8) return w;
    const double qd = ldexp(t, 52 - e1);
    if (!(fabs(qd) < 1.0e18)) return w;                                  // also nan
    const long long T = (long long)rint(qd) + (1ll << 28);               // rint: to nearest even, like the double addition
    long long lo, hi;
    if (key1 & 0x200) { lo = -(long long)SS_MHI + 1; hi = -(long long)SS_MLO - 1; } else { lo = SS_MLO + 1; hi = SS_MHI - 1; }
    if (k >= 0) {
        const int sh = 29 - k;
        if ((T & ((1ll << sh) - 1)) == 0) return w;                      // a tie (k = 0), or one for some m (k > 0)
        w.a = T >> sh; w.q = k; w.r = 0; w.c = 0;
    } else {
        if ((T & ((1ll << 29) - 1)) == 0) return w;
        w.a = 0; w.q = 0; w.r = -k; w.c = T >> 29;
    }
    ssw_preimage(w.a, w.q, w.r, w.c, lo, hi, &w.L, &w.H);
    if (w.L > w.H) return w;
    w.key_in = key0; w.key_out = key1;
    return w;
}

// a, then b (b's input binade must be a's output binade)
SS_HD ss_wild ssw_join(const ss_wild& A, const ss_wild& B)
{
    ss_wild w = ssw_invalid();
    if (!A.key_in || !B.key_in || A.key_out != B.key_in) return w;
    if (B.q <= A.r) {
        w.a = A.a; w.q = A.q; w.r = A.r - B.q + B.r;
        w.c = (((A.c + B.a) >> B.q) * (1ll << B.r)) + B.c;
    } else {
        const long long Bp = (A.c + B.a) >> A.r;
        w.a = A.a + Bp * (1ll << A.q); w.q = A.q + B.q - A.r; w.r = B.r; w.c = B.c;
    }
    if (w.q > SSW_MAXSHIFT || w.r > SSW_MAXSHIFT) return ssw_invalid();
    long long L, H;
    ssw_preimage(A.a, A.q, A.r, A.c, B.L, B.H, &L, &H);
    w.L = L > A.L ? L : A.L; w.H = H < A.H ? H : A.H;
    if (w.L > w.H) return ssw_invalid();
    w.key_in = A.key_in; w.key_out = B.key_out;
    return w;
}

// An in-binade record as a wild map: m' = m +- D on exactly the inputs ss_rec_fits accepts (the record's key must be a plain
// binade key: not 0, not SS_ID).  With it a whole batch of sub-tiles -- plain records and wild maps alike -- composes into ONE
// map (ssw_join is associative), and the true accumulator lies in the composite's interval exactly when it would have fitted
// every record of the batch one after the other (CPU model: composed_scheme; kernels: mclf_tree_helper / mclf_walk_trees).
SS_HD ss_wild ssw_from_rec(const ss_rec& r)
{
    ss_wild w = ssw_invalid();
    if (r.key == 0 || r.key == SS_ID) return w;
    const long long mlo = (long long)SS_MLO + 1 - (long long)r.lo, mhi = (long long)SS_MHI - 1 - (long long)r.hi;      // magnitudes that stay inside
    if (mlo > mhi) return w;
    const bool neg = (r.key & 0x200) != 0;
    w.key_in = w.key_out = r.key;
    w.c = neg ? -(long long)r.D : (long long)r.D;
    w.L = neg ? -mhi : mlo; w.H = neg ? -mlo : mhi;
    return w;
}

// ---------------------------------------------------------------------------------------------------------------------------
// The same for a DOUBLE accumulator stepped by doubles, c <- fl64(c + w): resamplePosteriorDistribution's cumulative weight
// (src/slam/particle_filter.cpp:94-99).  One rounding per step (no float stage), 53-bit magnitudes in 64-bit integers.
#define SSD_MLO (1ll << 52)
#define SSD_MHI (1ll << 53)

SS_HD uint64_t ssd_bits(double d) { uint64_t u; memcpy(&u, &d, 8); return u; }
SS_HD double ssd_exact_step(double c, double w) { return c + w; }
// 0: no usable binade (zero, subnormal or tiny, inf, nan); else 0x1000 | sign << 11 | biased exponent
SS_HD int ssd_key(double c)
{
    const uint64_t b = ssd_bits(c);
    const int ex = (int)((b >> 52) & 0x7ff);
    if (ex < 128 || ex == 2047) return 0;
    return 0x1000 | (int)((b >> 63) << 11) | ex;
}
SS_HD long long ssd_mag(double c) { return (long long)((ssd_bits(c) & 0xfffffffffffffull) | 0x10000000000000ull); }
SS_HD double ssd_from(int key, long long M)        // 2^52 <= M <= 2^53
{
    const double mag = ldexp((double)M, (key & 0x7ff) - 1023 - 52);
    return (key & 0x800) ? -mag : mag;
}
struct ssd_bin { double lim; int down; int neg; };
SS_HD ssd_bin ssd_bin_of(int key)
{
    const int e = (key & 0x7ff) - 1023;
    ssd_bin b;
    b.lim = ldexp(1.0, e - 1);          // |t| < 2^(e-1): the scaled term stays below 2^51, where doubles still carry quarters
    b.down = 52 - e;
    b.neg = (key & 0x800) ? 1 : 0;
    return b;
}
// integer the magnitude advances by for a term t, bad for a tie or an oversized term (branch-free)
SS_HD long long ssd_quantize(const ssd_bin& b, double t, int* bad)
{
    const double ta = b.neg ? -t : t;
    const bool big = !(fabs(ta) < b.lim);
    const double q = ldexp(big ? 0.0 : ta, b.down);             // exact
    const double f = floor(q);
    const double r = q - f;                                     // exact: |q| < 2^51
    *bad |= (big || r == 0.5) ? 1 : 0;
    return (long long)f + (r > 0.5 ? 1 : 0);
}

#endif  // BL_SERIAL_SUM_H
