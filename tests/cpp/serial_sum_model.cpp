// CPU model of the serially-rounded-accumulator scheme of botlab_amd/csrc/bl_serial_sum.h, lane for lane as the kernels run it
// (sub-tiles of 64 lanes x SS_ITEMS terms, records for a predicted binade, batches of 64 records scanned by composition, replay
// of what does not fit), checked against the plain loop of estimatePosteriorPose (particle_filter.cpp:151-152) on random and
// adversarial sequences.  Test infrastructure: built and run by tests/test_serial_sum_model.py, no GPU.
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <random>
#include <vector>

#include "../../botlab_amd/csrc/bl_serial_sum.h"

static const int ITEMS = 2, LANES = 64, SUB = ITEMS * LANES;

static float plain_loop(const std::vector<double>& t)
{
    float acc = 0.0f;
    for (double v : t) acc = (float)((double)acc + v);
    return acc;
}

struct Stats { long records = 0, fitted = 0, replays = 0, phases = 0, exact_steps = 0, batches = 0; };

// ---- stage A: the record of sub-tile s for a predicted start value
static ss_rec make_record(const std::vector<double>& t, int s, double predicted_start)
{
    const int key = ss_key((float)predicted_start);
    ss_rec r = ss_rec_make(key, 0, 0, 0);
    if (!key) return r;
    int bad = 0;
    long long run = 0, lo = (1ll << 40), hi = -(1ll << 40);
    for (int i = s * SUB; i < (s + 1) * SUB && i < (int)t.size(); ++i) {
        run += ss_quantize(key, t[i], &bad);
        if (run < lo) lo = run;
        if (run > hi) hi = run;
    }
    if (bad) { r.key = 0; return r; }
    r.D = ss_sat(run); r.lo = ss_sat(lo); r.hi = ss_sat(hi);
    return r;
}

// ---- replay of sub-tile s from the true accumulator: the wave's phase loop
static float replay(const std::vector<double>& t, int s, float acc, Stats& st)
{
    const int base = s * SUB, n = std::min(SUB, (int)t.size() - base);
    int pos = 0;
    st.replays++;
    while (pos < n) {
        const int key = ss_key(acc);
        if (!key) { acc = ss_exact_step(acc, t[base + pos]); pos++; st.exact_steps++; continue; }
        st.phases++;
        const int M = ss_mag(acc);
        // every lane: its terms from pos on, inclusive prefix, exit test
        int d[SUB], bad[SUB];
        for (int i = 0; i < n; ++i) { bad[i] = 0; d[i] = i >= pos ? ss_quantize(key, t[base + i], &bad[i]) : 0; }
        long long run = 0;
        int first_exit = -1;
        long long Mi[SUB];
        for (int i = 0; i < n; ++i) {
            run += d[i];
            Mi[i] = M + run;
            if (i >= pos && first_exit < 0 && (bad[i] || Mi[i] <= SS_MLO || Mi[i] >= SS_MHI)) first_exit = i;
        }
        if (first_exit < 0) { acc = ss_from(key, (int)Mi[n - 1]); pos = n; break; }
        const long long Mb = first_exit == pos ? M : Mi[first_exit - 1];
        acc = ss_exact_step(ss_from(key, (int)Mb), t[base + first_exit]);
        st.exact_steps++;
        pos = first_exit + 1;
    }
    return acc;
}

// ---- the chain: batches of 64 records; inside a batch the inclusive composition scan finds the first record that does not fit
static float chain(const std::vector<double>& t, const std::vector<ss_rec>& recs, Stats& st)
{
    float acc = 0.0f;
    const int nrec = (int)recs.size();
    int r0 = 0;
    while (r0 < nrec) {
        st.batches++;
        const int nb = std::min(LANES, nrec - r0);
        const int key = ss_key(acc), M = key ? ss_mag(acc) : 0;
        ss_rec pre[LANES];
        int first_bad = nb;
        for (int l = 0; l < nb; ++l) {
            pre[l] = l ? ss_rec_join(pre[l - 1], recs[r0 + l]) : recs[r0 + l];
            if (first_bad == nb && !(key && ss_rec_fits(pre[l], key, M))) first_bad = l;
        }
        if (first_bad > 0 && pre[first_bad - 1].key != SS_ID) acc = ss_from(key, M + pre[first_bad - 1].D);
        st.fitted += first_bad;
        if (first_bad < nb) acc = replay(t, r0 + first_bad, acc, st);
        r0 += first_bad + (first_bad < nb ? 1 : 0);
    }
    return acc;
}

static float scheme(const std::vector<double>& t, int predict_mode, std::mt19937_64& rng, Stats& st)
{
    const int nrec = ((int)t.size() + SUB - 1) / SUB;
    std::vector<ss_rec> recs(nrec);
    double P = 0.0;                                    // the prediction: a double-precision running sum
    for (int s = 0; s < nrec; ++s) {
        double start = P;
        if (predict_mode == 1) start = P * (1.0 + 1e-3 * ((double)(rng() % 2001) - 1000.0) / 1000.0);      // sloppy
        if (predict_mode == 2) start = ldexp(1.0 + (double)(rng() % 1000) / 1000.0, (int)(rng() % 60) - 40) * ((rng() & 1) ? 1 : -1);  // nonsense
        recs[s] = make_record(t, s, start);
        for (int i = s * SUB; i < (s + 1) * SUB && i < (int)t.size(); ++i) P += t[i];
    }
    st.records += nrec;
    return chain(t, recs, st);
}

// ---- wild runs: per-term binade predictions from a double prefix sum, one map + interval per sub-tile (ssw_*), applied when the
// true accumulator fits and replaced by plain steps when it does not
struct WildStats { long subtiles = 0, built = 0, applied = 0, stepped = 0; };
static float wild_scheme(const std::vector<double>& t, int predict_mode, int join_order, std::mt19937_64& rng, WildStats& ws)
{
    const int n = (int)t.size();
    float acc = 0.0f;
    double P = 0.0;
    for (int base = 0; base < n; base += SUB) {
        const int m = std::min(SUB, n - base);
        ws.subtiles++;
        // predictions: the key before the run and after every term
        double drift = 0.0;
        if (predict_mode == 1) drift = P * 1e-6 * ((double)(rng() % 2001) - 1000.0) / 1000.0;
        if (predict_mode == 2) drift = ldexp(1.0, (int)(rng() % 30) - 40);
        std::vector<int> keys(m + 1);
        double Q = P + drift;
        keys[0] = ss_key((float)Q);
        std::vector<ss_wild> steps(m);
        for (int i = 0; i < m; ++i) { Q += t[base + i]; keys[i + 1] = ss_key((float)Q); steps[i] = ssw_step(keys[i], keys[i + 1], t[base + i]); }
        ss_wild rec;
        if (join_order == 0) {                                   // left to right
            rec = steps[0];
            for (int i = 1; i < m; ++i) rec = ssw_join(rec, steps[i]);
        } else {                                                 // a balanced tree, as a wave's scan joins them
            std::vector<ss_wild> level = steps;
            while (level.size() > 1) {
                std::vector<ss_wild> next;
                for (size_t i = 0; i + 1 < level.size(); i += 2) next.push_back(ssw_join(level[i], level[i + 1]));
                if (level.size() & 1) next.push_back(level.back());
                level.swap(next);
            }
            rec = level[0];
        }
        if (rec.key_in) ws.built++;
        const int key = ss_key(acc);
        const long long ms = key ? ssw_signed(key, ss_mag(acc)) : 0;
        if (key && ssw_fits(rec, key, ms)) {
            const long long mo = ssw_apply(rec, ms);
            acc = ss_from(rec.key_out, (int)(mo < 0 ? -mo : mo));
            ws.applied++;
        } else {
            for (int i = 0; i < m; ++i) acc = ss_exact_step(acc, t[base + i]);
            ws.stepped++;
        }
        for (int i = 0; i < m; ++i) P += t[base + i];
    }
    return acc;
}

// ---- batches composed: per sub-tile either the in-binade record as a map (ssw_from_rec) or, where the double prediction sees
// the sub-tile leave its binade, its wild map; 64 of them joined into one map (the way a wave's inclusive scan joins them: lane l
// holds the join of maps first..l), the longest prefix that fits the true accumulator applied at once, the sub-tile behind it
// stepped, and on from there -- mclf_walk's loop
struct ComposedStats { long batches = 0, scans = 0, jumped = 0, stepped = 0; };
static float composed_scheme(const std::vector<double>& t, int predict_mode, std::mt19937_64& rng, ComposedStats& cs)
{
    const int n = (int)t.size();
    const int nsub = (n + SUB - 1) / SUB;
    // stage A: one map per sub-tile from a double-precision prediction of its start value
    std::vector<ss_wild> maps(nsub);
    double P = 0.0;
    for (int s = 0; s < nsub; ++s) {
        const int base = s * SUB, m = std::min(SUB, n - base);
        double drift = 0.0;
        if (predict_mode == 1) drift = P * 1e-6 * ((double)(rng() % 2001) - 1000.0) / 1000.0;
        if (predict_mode == 2) drift = ldexp(1.0, (int)(rng() % 30) - 40);
        double Q = P + drift;
        const int key0 = ss_key((float)Q);
        bool leaves = false;
        std::vector<ss_wild> steps(m);
        int kprev = key0;
        for (int i = 0; i < m; ++i) { Q += t[base + i]; const int k = ss_key((float)Q); steps[i] = ssw_step(kprev, k, t[base + i]); if (k != key0) leaves = true; kprev = k; }
        if (!leaves && key0) maps[s] = ssw_from_rec(make_record(t, s, P + drift));
        else { ss_wild rec = steps[0]; for (int i = 1; i < m; ++i) rec = ssw_join(rec, steps[i]); maps[s] = rec; }
        for (int i = 0; i < m; ++i) P += t[base + i];
    }
    // the chain
    float acc = 0.0f;
    for (int b0 = 0; b0 < nsub; b0 += LANES) {
        const int nb = std::min(LANES, nsub - b0);
        cs.batches++;
        int pos = 0;
        while (pos < nb) {
            cs.scans++;
            std::vector<ss_wild> pre(nb);
            for (int l = pos; l < nb; ++l) pre[l] = l == pos ? maps[b0 + l] : ssw_join(pre[l - 1], maps[b0 + l]);
            const int key = ss_key(acc);
            const long long ms = key ? ssw_signed(key, ss_mag(acc)) : 0;
            int p = 0;
            while (pos + p < nb && key && ssw_fits(pre[pos + p], key, ms)) ++p;
            if (p > 0) {
                const ss_wild& w = pre[pos + p - 1];
                const long long mo = ssw_apply(w, ms);
                acc = ss_from(w.key_out, (int)(mo < 0 ? -mo : mo));
                cs.jumped += p;
                pos += p;
            }
            if (pos < nb) {
                const int base = (b0 + pos) * SUB, m = std::min(SUB, n - base);
                for (int i = 0; i < m; ++i) acc = ss_exact_step(acc, t[base + i]);
                cs.stepped++;
                pos += 1;
            }
        }
    }
    return acc;
}

// ---- the double accumulator (resampling cumulative): every prefix value, by the phase loop alone
static long check_double_scheme(std::mt19937_64& rng, long* cases)
{
    long failures = 0;
    for (int kind = 0; kind < 6; ++kind) {
        const int n = 1 + (int)(rng() % 3000);
        std::vector<double> w(n), want(n), got(n);
        double S = 0;
        std::vector<double> u(n);
        for (int i = 0; i < n; ++i) {
            switch (kind) {
            case 0: u[i] = 1.0; break;                                              // a fresh filter: all equal
            case 1: u[i] = 1000.0 * (double)(40 + rng() % 36000); break;               // an update's likelihood units
            case 2: u[i] = (rng() % 50 == 0) ? 2.0 : 1000.0 * (double)(1 + rng() % 500); break;   // with floored weights
            case 3: u[i] = (double)(1ull << (rng() % 40)); break;                      // powers of two: ties
            case 4: u[i] = (i % 7 == 0) ? 1e12 : 1.0; break;                           // a few dominant weights
            default: u[i] = (double)(1 + rng() % 3); break;
            }
            S += u[i];
        }
        for (int i = 0; i < n; ++i) w[i] = u[i] / S;
        double c = w[0];
        want[0] = c;
        for (int i = 1; i < n; ++i) { c = c + w[i]; want[i] = c; }
        // the scheme: sub-tiles of SUB terms, phases inside
        double acc = 0.0;
        for (int base = 0; base < n; base += SUB) {
            const int m = std::min(SUB, n - base);
            int pos = 0;
            while (pos < m) {
                const int key = ssd_key(acc);
                if (!key) { acc = ssd_exact_step(acc, w[base + pos]); got[base + pos] = acc; pos++; continue; }
                const long long M = ssd_mag(acc);
                const ssd_bin b = ssd_bin_of(key);
                long long run = 0;
                int exit_at = -1;
                long long Mi[SUB];
                for (int i = pos; i < m; ++i) {
                    int bad = 0;
                    run += ssd_quantize(b, w[base + i], &bad);
                    Mi[i] = M + run;
                    if (bad || Mi[i] <= SSD_MLO || Mi[i] >= SSD_MHI) { exit_at = i; break; }
                }
                const int upto = exit_at < 0 ? m : exit_at;
                for (int i = pos; i < upto; ++i) got[base + i] = ssd_from(key, Mi[i]);
                if (exit_at < 0) { acc = got[base + m - 1]; pos = m; break; }
                const double before = exit_at == pos ? acc : got[base + exit_at - 1];
                acc = ssd_exact_step(before, w[base + exit_at]);
                got[base + exit_at] = acc;
                pos = exit_at + 1;
            }
        }
        (*cases)++;
        for (int i = 0; i < n; ++i)
            if (ssd_bits(got[i]) != ssd_bits(want[i])) { if (failures < 5) fprintf(stderr, "DOUBLE MISMATCH kind %d n %d i %d\n", kind, n, i); failures++; break; }
    }
    return failures;
}

int main(int argc, char** argv)
{
    const int rounds = argc > 1 ? atoi(argv[1]) : 60;
    std::mt19937_64 rng(12345);
    std::normal_distribution<double> nd(0.0, 1.0);
    long cases = 0, failures = 0;
    Stats st;
    for (int round = 0; round < rounds; ++round) {
        for (int kind = 0; kind < 10; ++kind) {
            int n = 1 + (int)(rng() % (round % 7 == 0 ? 300000 : 5000));
            if (kind == 9) n = 1 + (int)(rng() % 400);
            std::vector<double> t(n);
            const double centre = (kind & 1) ? 0.0 : ldexp(1.0, (int)(rng() % 12) - 6) * ((rng() & 2) ? 1 : -1);
            const double spread = ldexp(1.0, (int)(rng() % 10) - 8);
            const double w = 1.0 / n;
            for (int i = 0; i < n; ++i) {
                const float x = (float)(centre + spread * nd(rng));
                switch (kind) {
                case 0: case 1: t[i] = (w * (1.0 + 0.5 * nd(rng))) * (double)x; break;            // weights x poses: the real thing
                case 2: t[i] = w * (double)(float)centre; break;                                   // all terms equal
                case 3: t[i] = ldexp((double)((int)(rng() % 33) - 16), -20 - (int)(rng() % 4)); break;          // few significant bits: ties galore
                case 4: t[i] = ldexp(nd(rng), (int)(rng() % 80) - 60); break;                      // wild dynamic range
                case 5: t[i] = (i % 97 == 0) ? -0.9 * (double)i * w : w * (double)x; break;          // big negative jolts: falls through binades and zero
                case 6: t[i] = (rng() % 5 == 0) ? 0.0 : w * (double)x; break;
                case 7: t[i] = ldexp(1.0, -24) * (double)((rng() % 3) + 1) * ((rng() & 1) ? 0.5 : 1.0); break; // halves of an ulp near 1.0 once summed
                case 8: t[i] = (i & 1) ? w * (double)x : -w * (double)x * (1.0 - 1e-7); break;    // cancelling pairs around zero
                default: t[i] = w * (double)x; break;
                }
            }
            const float want = plain_loop(t);
            for (int mode = 0; mode < 3; ++mode) {
                const float got = scheme(t, mode, rng, st);
                cases++;
                if (ss_f2u(got) != ss_f2u(want) && !(got != got && want != want)) {
                    if (failures < 10) fprintf(stderr, "MISMATCH kind %d n %d mode %d: got %.9g want %.9g\n", kind, n, mode, got, want);
                    failures++;
                }
            }
        }
    }
    for (int round = 0; round < rounds; ++round) failures += check_double_scheme(rng, &cases);
    // wild runs: sums that hover around zero (the reference's default start pose), and the adversarial kinds again
    WildStats ws, ws_zero;
    ComposedStats cs, cs_zero;
    for (int round = 0; round < rounds; ++round) {
        for (int kind = 0; kind < 12; ++kind) {
            int n = 1 + (int)(rng() % (round % 5 == 0 ? 200000 : 6000));
            std::vector<double> t(n);
            const double spread = ldexp(1.0, (int)(rng() % 8) - 7);
            const double centre = kind < 4 ? 0.0 : (kind < 8 ? spread * 1e-3 * nd(rng) : ldexp(1.0, (int)(rng() % 12) - 8) * ((rng() & 2) ? 1 : -1));
            const double w = 1.0 / n;
            for (int i = 0; i < n; ++i) {
                const float x = (float)(centre + spread * nd(rng));
                switch (kind % 4) {
                case 0: t[i] = (w * (1.0 + 0.3 * nd(rng))) * (double)x; break;
                case 1: t[i] = w * (double)x; break;
                case 2: t[i] = ldexp((double)((int)(rng() % 65) - 32), -24 - (int)(rng() % 3)); break;       // few bits: ties
                default: t[i] = (i % 53 == 0) ? -0.7 * (double)i * w * centre : w * (double)x; break;       // jolts
                }
            }
            const float want = plain_loop(t);
            for (int mode = 0; mode < 3; ++mode) {
                const float got = composed_scheme(t, mode, rng, kind < 8 && mode == 0 ? cs_zero : cs);
                cases++;
                if (ss_f2u(got) != ss_f2u(want) && !(got != got && want != want)) {
                    if (failures < 10) fprintf(stderr, "COMPOSED MISMATCH kind %d n %d mode %d: got %.9g want %.9g\n", kind, n, mode, got, want);
                    failures++;
                }
            }
            for (int mode = 0; mode < 3; ++mode)
                for (int order = 0; order < 2; ++order) {
                    WildStats& tgt = (kind < 8 && mode == 0) ? ws_zero : ws;
                    const float got = wild_scheme(t, mode, order, rng, tgt);
                    cases++;
                    if (ss_f2u(got) != ss_f2u(want) && !(got != got && want != want)) {
                        if (failures < 10) fprintf(stderr, "WILD MISMATCH kind %d n %d mode %d order %d: got %.9g want %.9g\n", kind, n, mode, order, got, want);
                        failures++;
                    }
                }
        }
    }
    printf("cases %ld failures %ld records %ld fitted %ld replays %ld phases %ld exact_steps %ld batches %ld wild_subtiles %ld wild_built %ld wild_applied %ld "
           "zero_subtiles %ld zero_applied %ld composed_batches %ld composed_scans %ld composed_jumped %ld composed_stepped %ld zero_batches %ld zero_scans %ld zero_jumped %ld zero_stepped %ld\n", cases, failures, st.records,
           st.fitted, st.replays, st.phases, st.exact_steps, st.batches, ws.subtiles + ws_zero.subtiles, ws.built + ws_zero.built, ws.applied + ws_zero.applied,
           ws_zero.subtiles, ws_zero.applied, cs.batches, cs.scans, cs.jumped, cs.stepped, cs_zero.batches, cs_zero.scans, cs_zero.jumped, cs_zero.stepped);
    return failures ? 1 : 0;
}
