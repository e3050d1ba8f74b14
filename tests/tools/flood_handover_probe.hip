// What would a level of the frontier flood cost to hand over between K workgroups?  ("frontier flood": the one-workgroup
// flood takes 6.4 us per level; a split over K workgroups needs every workgroup to see every other's winners before the next level.)
// The exchange measured here is the cheapest one the flood could use: tagged data, one hop.  Every level each workgroup stores
// `cnt` 64-bit slots {entry, level tag} and a header {count, level tag} into its own segment (two segments by level parity), then
// reads the K headers and every slot of every segment, polling until the tag is the level's.  No compute between: the time per level
// is the hand-over alone.  Variants: the workgroups on ONE XCD (one L2) or spread over all of them; stores plain or sc1
// (write-through to memory); loads sc1 or sc0 sc1.
//   hipcc -O2 --offload-arch=gfx950 tests/tools/flood_handover_probe.hip -o /tmp/fhp && /tmp/fhp
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define SEG 1024                       // slots per segment (header included)
#define T 256

struct args {
    unsigned long long* slots;         // [2][K][SEG]
    int* tickets;                      // [8] + abort word at [8]
    long long* out;                    // [K][4]: ticks, checksum, xcc, polls that had to be repeated
    int K, levels, cnt, target_xcc;
};

template <int LD> __device__ __forceinline__ unsigned long long ld64(const unsigned long long* p)
{
    unsigned long long v;
    if (LD == 0) asm volatile("global_load_dwordx2 %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
    else         asm volatile("global_load_dwordx2 %0, %1, off sc0 sc1\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
    return v;
}
template <int LD> __device__ __forceinline__ void ld64_issue(unsigned long long& v, const unsigned long long* p)
{
    if (LD == 0) asm volatile("global_load_dwordx2 %0, %1, off sc1" : "=v"(v) : "v"(p) : "memory");
    else         asm volatile("global_load_dwordx2 %0, %1, off sc0 sc1" : "=v"(v) : "v"(p) : "memory");
}
template <int ST> __device__ __forceinline__ void st64(unsigned long long* p, unsigned long long v)
{
    if (ST == 0) asm volatile("global_store_dwordx2 %0, %1, off" :: "v"(p), "v"(v) : "memory");
    else         asm volatile("global_store_dwordx2 %0, %1, off sc1" :: "v"(p), "v"(v) : "memory");
}
__host__ __device__ __forceinline__ unsigned int entry_of(int L, int k, int i) { return (unsigned int)(L * 2654435761u) ^ (unsigned int)(k << 20) ^ (unsigned int)i; }

template <int KK, int LD, int ST> __global__ __launch_bounds__(T) void k_probe(args a)
{
    __shared__ int s_k;
    const int tid = threadIdx.x;
    int xcc = (int)(__builtin_amdgcn_s_getreg(63508) & 15u);
    if (tid == 0) {
        if (a.target_xcc >= 0 && xcc != a.target_xcc) s_k = -1;
        else { const int t = atomicAdd(&a.tickets[a.target_xcc >= 0 ? xcc : 0], 1); s_k = t < KK ? t : -1; }
    }
    __syncthreads();
    const int k = s_k;
    if (k < 0) return;
    volatile int* abort_word = a.tickets + 8;
    unsigned long long sum = 0; long long repeats = 0;
    const long long t0 = wall_clock64();
    for (int L = 1; L <= a.levels; ++L) {
        unsigned long long* mine = a.slots + ((size_t)(L & 1) * KK + k) * SEG;
        const unsigned long long tag = (unsigned long long)((unsigned int)L << 8) << 32;
        for (int i = tid; i < a.cnt; i += T) st64<ST>(mine + 1 + i, tag | entry_of(L, k, i));
        if (tid == 0) st64<ST>(mine, tag | (unsigned int)a.cnt);
        const unsigned long long* base = a.slots + (size_t)(L & 1) * KK * SEG;
        // the K headers and the first slot of every segment in one round of loads; what has not arrived is asked for again, all of
        // it in one round (a poll per segment in turn costs a round trip per segment: 0.46 us x K, measured with the first form)
        unsigned long long h[KK], e[KK];
        int spins = 0;
        for (;;) {
#pragma unroll
            for (int j = 0; j < KK; ++j) {
                if (spins == 0 || (h[j] >> 32) != (tag >> 32)) ld64_issue<LD>(h[j], base + (size_t)j * SEG);
                if (spins == 0 || (e[j] >> 32) != (tag >> 32)) ld64_issue<LD>(e[j], base + (size_t)j * SEG + 1 + tid);
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            bool all = true;
#pragma unroll
            for (int j = 0; j < KK; ++j) {
                const bool hv = (h[j] >> 32) == (tag >> 32);
                all = all && hv && ((e[j] >> 32) == (tag >> 32) || tid >= (int)(unsigned int)h[j]);
            }
            if (all) break;
            repeats += 1;
            if ((++spins & 1023) == 0 && (*abort_word || spins > (1 << 21))) { *abort_word = 1; goto out; }
        }
#pragma unroll
        for (int j = 0; j < KK; ++j) {
            const int cj = (int)(unsigned int)h[j];
            if (tid < cj) sum += (unsigned int)e[j];
            for (int i = tid + T; i < cj; i += T) {               // segments longer than a round of the workgroup: rare in the flood
                unsigned long long v = ld64<LD>(base + (size_t)j * SEG + 1 + i);
                int sp = 0;
                while ((v >> 32) != (tag >> 32)) {
                    v = ld64<LD>(base + (size_t)j * SEG + 1 + i); repeats += 1;
                    if ((++sp & 1023) == 0 && (*abort_word || sp > (1 << 21))) { *abort_word = 1; goto out; }
                }
                sum += (unsigned int)v;
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    }
out:
    const long long t1 = wall_clock64();
    // the workgroup's checksum: every thread's sum, added up through LDS atomics
    __shared__ unsigned long long s_sum; __shared__ long long s_rep;
    if (tid == 0) { s_sum = 0; s_rep = 0; }
    __syncthreads();
    atomicAdd(&s_sum, sum); atomicAdd((unsigned long long*)&s_rep, (unsigned long long)repeats);
    __syncthreads();
    if (tid == 0) { a.out[k * 4 + 0] = t1 - t0; a.out[k * 4 + 1] = (long long)s_sum; a.out[k * 4 + 2] = xcc; a.out[k * 4 + 3] = s_rep; }
}

template <int KK, int LD, int ST> static void run(int levels, int cnt, int target_xcc, const char* what)
{
    args a; a.K = KK; a.levels = levels; a.cnt = cnt; a.target_xcc = target_xcc;
    hipMalloc(&a.slots, (size_t)2 * KK * SEG * 8); hipMemset(a.slots, 0, (size_t)2 * KK * SEG * 8);
    hipMalloc(&a.tickets, 16 * 4); hipMemset(a.tickets, 0, 16 * 4);
    hipMalloc(&a.out, KK * 4 * 8); hipMemset(a.out, 0, KK * 4 * 8);
    // one XCD: eight times as many workgroups as needed, those that are not on the target XCD (or come too late) leave at once
    const int grid = target_xcc >= 0 ? 8 * KK : KK;
    hipLaunchKernelGGL((k_probe<KK, LD, ST>), dim3(grid), dim3(T), 0, 0, a);
    hipDeviceSynchronize();
    std::vector<long long> out(KK * 4); int tickets[16];
    hipMemcpy(out.data(), a.out, KK * 4 * 8, hipMemcpyDeviceToHost); hipMemcpy(tickets, a.tickets, 16 * 4, hipMemcpyDeviceToHost);
    unsigned long long expect = 0;
    for (int L = 1; L <= levels; ++L) for (int k = 0; k < KK; ++k) for (int i = 0; i < cnt; ++i) expect += entry_of(L, k, i);
    long long worst = 0, rep = 0; bool ok = !tickets[8]; int xccs = 0;
    for (int k = 0; k < KK; ++k) { if (out[k * 4] > worst) worst = out[k * 4]; ok = ok && (unsigned long long)out[k * 4 + 1] == expect; rep += out[k * 4 + 3]; xccs |= 1 << out[k * 4 + 2]; }
    printf("%-34s K %2d  %4d slots per workgroup and level  %7.3f us per level  %s  polls repeated %.1f per workgroup and level  (XCD mask 0x%02x%s)\n",
           what, KK, cnt, worst * 0.01 / levels, ok ? "data ok" : "DATA WRONG / GAVE UP", (double)rep / KK / levels, xccs, tickets[8] ? ", aborted" : "");
    hipFree(a.slots); hipFree(a.tickets); hipFree(a.out);
}

int main()
{
    const int levels = 3000;
    printf("hand-over of a flood level between K workgroups of 256 threads, %d levels, no compute between\n", levels);
    run<4, 0, 0>(levels, 256, 0, "one XCD, plain store, sc1 load");
    run<8, 0, 0>(levels, 256, 0, "one XCD, plain store, sc1 load");
    run<16, 0, 0>(levels, 128, 0, "one XCD, plain store, sc1 load");
    run<8, 1, 0>(levels, 256, 0, "one XCD, plain store, sc0 sc1 load");
    run<8, 0, 1>(levels, 256, 0, "one XCD, sc1 store, sc1 load");
    run<8, 0, 1>(levels, 256, -1, "any XCD, sc1 store, sc1 load");
    run<8, 1, 1>(levels, 256, -1, "any XCD, sc1 store, sc0 sc1 load");
    run<16, 0, 1>(levels, 128, -1, "any XCD, sc1 store, sc1 load");
    run<8, 0, 0>(levels, 256, -1, "any XCD, plain store, sc1 load");
    run<8, 0, 0>(levels, 64, 0, "one XCD, plain store, sc1 load");
    run<2, 0, 0>(levels, 512, 0, "one XCD, plain store, sc1 load");
    return 0;
}
