// CPU model of k_astar2's open list (bl_astar2.h): the wave-parallel std::push_heap / std::pop_heap on split storage
// (16-bit biased keys, 32-bit payloads), lane for lane, checked against libstdc++ on random operation sequences with many
// equal keys.  Test infrastructure only: g++ -O2 -o heap2_model heap2_model.cpp && ./heap2_model
#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <functional>
#include <random>
#include <vector>

struct Node { int f; unsigned pay; };
struct Greater { bool operator()(const Node& a, const Node& b) const { return a.f > b.f; } };

static const unsigned INF = 0xFFFFu;

struct Heap2 {
    int KLV;                                  // key levels 0..KLV are "LDS" (no validity checks: slots >= len hold INF)
    int FD;                                   // levels the first round descends
    std::vector<unsigned> key;                // slot = idx + 1; slot 0 = 0 (-inf)
    std::vector<unsigned> pay;                // by idx
    int len = 0;
    long rounds = 0, slow = 0;
    Heap2(int klv, size_t cap) : KLV(klv), FD(klv - 10), key(std::max<size_t>(2 * cap + 8, ((size_t)2 << klv) + 8), 0xDEAD), pay(cap + 8, 0) {
        const int kslots = 1 << (KLV + 1);
        for (int i = 0; i < kslots && i < (int)key.size(); ++i) key[i] = INF;
        key[0] = 0;
    }
    int kslots() const { return 1 << (KLV + 1); }
    unsigned top_key() const { return key[1]; }
    unsigned top_pay() const { return pay[0]; }

    void push(unsigned kb, unsigned p) {
        const unsigned hp = (unsigned)len + 1u;
        unsigned ka[64]; unsigned long long GT = 0;
        for (int a = 0; a < 64; ++a) {
            const unsigned slot = (a + 1 < 32) ? (hp >> (a + 1)) : 0;
            ka[a] = key[slot];                 // slot 0 -> 0: never greater
            if (ka[a] > kb) GT |= 1ull << a;
        }
        const int t = __builtin_ffsll((long long)~GT) - 1;
        unsigned pa[64];
        for (int a = 0; a < t; ++a) pa[a] = pay[(hp >> (a + 1)) - 1];
        for (int a = 0; a < t; ++a) { key[hp >> a] = ka[a]; pay[(hp >> a) - 1] = pa[a]; }
        key[hp >> t] = kb; pay[(hp >> t) - 1] = p;
        len += 1;
    }

    // pop_heap + pop_back.  Returns the popped top.
    Node pop() {
        Node topn{(int)key[1] - 32768, pay[0]};
        const int last = len - 1;
        len = last;
        const unsigned vk = key[last + 1], vp = pay[last];
        if (last + 1 < kslots()) key[last + 1] = INF;
        if (last == 0) return topn;
        struct Rnd { unsigned long long Q; unsigned node[64], child[64], knext[64], pnext[64]; };
        std::vector<Rnd> R;
        unsigned hp = 1; int start = 0;
        int leaf = 0;
        for (int r = 0;; ++r) {
            const int DL = r == 0 ? FD : 5;
            const bool lds_keys = start + DL <= KLV;        // children of this round's internal nodes at levels <= KLV
            Rnd X; X.Q = 0;
            unsigned long long M = 0, V = 0;
            for (int lane = 0; lane < 63; ++lane) {
                const int lk = 31 - __builtin_clz(lane + 1), lj = (lane + 1) - (1 << lk);
                if (lk > DL) continue;
                const unsigned node = (hp << lk) + lj - 1;
                X.node[lane] = node;
                if ((long long)node < len) V |= 1ull << lane;
                unsigned fl, fr;
                if (lds_keys && lk < DL) { fl = key[2 * node + 2]; fr = key[2 * node + 3]; }
                else { fl = (long long)2 * node + 1 < len ? key[2 * node + 2] : INF; fr = (long long)2 * node + 2 < len ? key[2 * node + 3] : INF; }
                const bool pr = fr <= fl;
                if (pr) M |= 1ull << lane;
                X.knext[lane] = pr ? fr : fl;
                X.child[lane] = 2 * node + 1 + (pr ? 1 : 0);
            }
            unsigned long long P = 0;
            for (int lane = 0; lane < 63; ++lane) {
                const int lk = 31 - __builtin_clz(lane + 1), lj = (lane + 1) - (1 << lk);
                if (lk > DL) continue;
                unsigned amask = 0, areq = 0;
                for (int t = 0; t < lk; ++t) {
                    const int anc_lane = ((1 << t) - 1) + (lj >> (lk - t));
                    amask |= 1u << anc_lane;
                    areq |= (unsigned)((lj >> (lk - t - 1)) & 1) << anc_lane;
                }
                if ((((unsigned)M) & amask) == areq && ((V >> lane) & 1)) P |= 1ull << lane;
            }
            const int cur = 63 - __builtin_clzll(P);
            X.Q = P & ~(1ull << cur);
            for (int lane = 0; lane < 63; ++lane) if ((X.Q >> lane) & 1) X.pnext[lane] = pay[X.child[lane]];
            R.push_back(X);
            rounds++;
            const unsigned hole = X.node[cur];
            leaf = (int)hole;
            const int first_bottom = (1 << DL) - 1;
            if (!(cur >= first_bottom && (long long)2 * hole + 1 < len)) break;
            hp = hole + 1; start += DL;
        }
        // the climb: deepest position whose successor's key is not greater than the value's
        int rs = -1, L = -1;
        for (int r = (int)R.size() - 1; r >= 0 && rs < 0; --r) {
            unsigned long long S = 0;
            for (int lane = 0; lane < 63; ++lane) if (((R[r].Q >> lane) & 1) && !(R[r].knext[lane] > vk)) S |= 1ull << lane;
            if (S) { rs = r; L = 63 - __builtin_clzll(S); }
            else if (r == (int)R.size() - 1) slow++;
        }
        unsigned land = 0;
        if (rs >= 0) land = R[rs].child[L];
        for (int r = 0; r < (int)R.size(); ++r) {
            unsigned long long W = 0;
            if (rs >= 0) { if (r < rs) W = R[r].Q; else if (r == rs) W = R[r].Q & ((2ull << L) - 1ull); }
            for (int lane = 0; lane < 63; ++lane) if ((W >> lane) & 1) { key[R[r].node[lane] + 1] = R[r].knext[lane]; pay[R[r].node[lane]] = R[r].pnext[lane]; }
        }
        key[land + 1] = vk; pay[land] = vp;
        (void)leaf;
        return topn;
    }
};

int main(int argc, char** argv)
{
    const int klv = argc > 1 ? atoi(argv[1]) : 15;
    std::mt19937 rng(1234);
    long checked = 0;
    for (int trial = 0; trial < 60; ++trial) {
        const size_t cap = trial < 40 ? 5000 : (trial < 55 ? 200000 : 2000000);
        Heap2 h(klv, cap);
        std::vector<Node> ref;
        const int spread = trial % 3 == 0 ? 4 : (trial % 3 == 1 ? 40 : 4000);
        unsigned serial = 0;
        const long ops = trial < 40 ? 20000 : (trial < 55 ? 600000 : 5000000);
        double push_bias = trial % 2 ? 0.62 : 0.5;
        for (long op = 0; op < ops; ++op) {
            const bool do_push = ref.empty() || ((rng() % 1000) < push_bias * 1000 && ref.size() + 1 < cap);
            if (do_push) {
                int base = ref.empty() ? 0 : ref.front().f;
                int f = base + (int)(rng() % (2 * spread + 1)) - spread / 2;
                if (rng() % 7 == 0) f = base - (int)(rng() % 50);             // a new best: rises to the root
                if (f < -32000) f = -32000;
                if (f > 32766) f = 32766;
                Node n{f, serial++};
                ref.push_back(n); std::push_heap(ref.begin(), ref.end(), Greater());
                h.push((unsigned)(f + 32768), n.pay);
            } else {
                std::pop_heap(ref.begin(), ref.end(), Greater());
                Node e = ref.back(); ref.pop_back();
                Node g = h.pop();
                if (e.f != g.f || e.pay != g.pay) { printf("MISMATCH pop trial %d op %ld: ref (%d,%u) got (%d,%u)\n", trial, op, e.f, e.pay, g.f, g.pay); return 1; }
            }
            if ((int)ref.size() != h.len) { printf("len mismatch\n"); return 1; }
            if (op % 997 == 0 || ref.size() < 40) {
                for (size_t i = 0; i < ref.size(); ++i)
                    if ((unsigned)(ref[i].f + 32768) != h.key[i + 1] || ref[i].pay != h.pay[i]) { printf("MISMATCH array trial %d op %ld idx %zu\n", trial, op, i); return 1; }
                for (int s = (int)ref.size() + 1; s < h.kslots() && s < (int)ref.size() + 70; ++s) if (h.key[s] != INF) { printf("INF invariant broken trial %d op %ld slot %d\n", trial, op, s); return 1; }
                checked++;
            }
        }
        printf("trial %d ok: final size %zu, pop rounds %ld, climbs past the last round %ld\n", trial, ref.size(), h.rounds, h.slow);
    }
    printf("all ok (%ld full array checks)\n", checked);
    return 0;
}
