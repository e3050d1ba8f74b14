// bl_ctx.hip -- context, error reporting, kernel timers, scan staging and the device OccupancyGrid.
#include <sched.h>
#include <stdarg.h>
#include <stdio.h>
#include <string.h>

#include <chrono>

#include "bl_internal.h"

static thread_local char g_err[512] = "";

void bl_set_error(const char* fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

extern "C" const char* bl_last_error(void) { return g_err; }

// Kernels of `device` may read and write memory of `peer_device` through pointers of the same process (a host that drives
// several devices from one process: include/botlab/sharded_filter.hpp).  BL_OK when it is (or already was) enabled.
extern "C" int bl_dev_enable_peer_access(int device, int peer_device)
{
    if (device == peer_device) return BL_OK;
    int can = 0;
    BL_HIP(hipDeviceCanAccessPeer(&can, device, peer_device));
    if (!can) { bl_set_error("device %d cannot access device %d", device, peer_device); return BL_ERR_STATE; }
    BL_HIP(hipSetDevice(device));
    const hipError_t e = hipDeviceEnablePeerAccess(peer_device, 0);
    if (e != hipSuccess && e != hipErrorPeerAccessAlreadyEnabled) { bl_set_error("hipDeviceEnablePeerAccess(%d -> %d): %s", device, peer_device, hipGetErrorString(e)); return BL_ERR_HIP; }
    (void)hipGetLastError();
    return BL_OK;
}
extern "C" const char* bl_version(void) { return "botlab_hip 0.1 (gfx950)"; }

void bl_astar_free(bl_ctx* ctx);   // bl_planning.hip
void bl_frontier_scratch_free(bl_ctx* ctx);   // bl_frontiers.hip

extern "C" int bl_ctx_create(int device, void* stream, bl_ctx** out)
{
    BL_CHECK_ARG(out != nullptr);
    int ndev = 0;
    BL_HIP(hipGetDeviceCount(&ndev));
    if (device < 0 || device >= ndev) {
        bl_set_error("device %d out of range (%d visible)", device, ndev);
        return BL_ERR_ARG;
    }
    BL_HIP(hipSetDevice(device));
    bl_ctx* c = new bl_ctx();
    c->device = device;
    if (stream) {
        c->stream = (hipStream_t)stream;
        c->own_stream = false;
    } else {
        hipError_t e = hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking);
        if (e != hipSuccess) {
            bl_set_error("hipStreamCreate failed: %s", hipGetErrorString(e));
            delete c;
            return BL_ERR_HIP;
        }
        c->own_stream = true;
    }
    *out = c;
    return BL_OK;
}

// A context whose own stream has the device's LOWEST priority: for work that may keep a hardware queue busy for seconds (a plan to
// a frontier is one kernel of up to 10^6 pops).  Streams of one priority share the runtime's few hardware queues, and a stream
// that lands behind such a kernel on its queue waits for it -- the SLAM stream did, in one run out of a few, for longer than the
// scan staging's patience.  Streams of another priority have queues of their own.
int bl_ctx_create_low_priority(int device, bl_ctx** out)
{
    BL_CHECK_ARG(out != nullptr);
    BL_HIP(hipSetDevice(device));
    int least = 0, greatest = 0;
    BL_HIP(hipDeviceGetStreamPriorityRange(&least, &greatest));
    hipStream_t st = nullptr;
    BL_HIP(hipStreamCreateWithPriority(&st, hipStreamNonBlocking, least));
    int rc = bl_ctx_create(device, (void*)st, out);
    if (rc) { (void)hipStreamDestroy(st); return rc; }
    (*out)->own_stream = true;
    return BL_OK;
}

extern "C" void bl_ctx_destroy(bl_ctx* ctx)
{
    if (!ctx) return;
    (void)hipSetDevice(ctx->device);
    (void)hipStreamSynchronize(ctx->stream);
    bl_astar_free(ctx);
    bl_frontier_scratch_free(ctx);
    bl_scan_free(ctx);
    for (int i = 0; i < BL_K_COUNT; ++i) {
        for (auto& p : ctx->timers[i].pending) { (void)hipEventDestroy(p.first); (void)hipEventDestroy(p.second); }
        for (auto& p : ctx->timers[i].pool) { (void)hipEventDestroy(p.first); (void)hipEventDestroy(p.second); }
    }
    if (ctx->own_stream) (void)hipStreamDestroy(ctx->stream);
    delete ctx;
}

extern "C" int bl_ctx_sync(bl_ctx* ctx)
{
    BL_CHECK_ARG(ctx != nullptr);
    BL_HIP(hipStreamSynchronize(ctx->stream));
    return BL_OK;
}

// ---------------------------------------------------------------- timers
extern "C" int bl_ctx_timing_enable(bl_ctx* ctx, int on)
{
    BL_CHECK_ARG(ctx != nullptr);
    ctx->timing = on != 0;
    ctx->timing_mask = (on == 1) ? 0x3fu : (unsigned int)on;            // 1: every kernel (ids 0..5); otherwise bit i = BL_K_* id i
    return BL_OK;
}

extern "C" int bl_ctx_timing_stride(bl_ctx* ctx, int every)
{
    BL_CHECK_ARG(ctx != nullptr && every >= 1);
    ctx->timing_stride = every;
    return BL_OK;
}

int bl_timer_begin(bl_ctx* ctx, int id, hipEvent_t* a, hipEvent_t* b)
{
    *a = nullptr; *b = nullptr;
    if (!ctx->timing || !((ctx->timing_mask >> id) & 1u)) return BL_OK;
    bl_timer& t = ctx->timers[id];
    if (ctx->timing_stride > 1 && (t.seen++ % ctx->timing_stride) != 0) return BL_OK;
    if (!t.pool.empty()) {
        *a = t.pool.back().first; *b = t.pool.back().second;
        t.pool.pop_back();
    } else {
        BL_HIP(hipEventCreate(a));
        BL_HIP(hipEventCreate(b));
    }
    BL_HIP(hipEventRecord(*a, ctx->stream));
    return BL_OK;
}

// The same bookkeeping for a launch that takes its own start/stop events (hipExtLaunchKernelGGL: the events carry the
// kernel's begin and end time stamps themselves, with no barrier packets around it -- a pair of hipEventRecord calls measures
// the launch gap as well, ~13 us here).  *a stays null when this launch is not to be timed.
int bl_timer_pair(bl_ctx* ctx, int id, hipEvent_t* a, hipEvent_t* b)
{
    *a = nullptr; *b = nullptr;
    if (!ctx->timing || !((ctx->timing_mask >> id) & 1u)) return BL_OK;
    bl_timer& t = ctx->timers[id];
    if (ctx->timing_stride > 1 && (t.seen++ % ctx->timing_stride) != 0) return BL_OK;
    if (!t.pool.empty()) {
        *a = t.pool.back().first; *b = t.pool.back().second;
        t.pool.pop_back();
    } else {
        BL_HIP(hipEventCreate(a));
        BL_HIP(hipEventCreate(b));
    }
    return BL_OK;
}

// A timer that is never read must not pile up events (a few ten thousand live events make every hipEventRecord slow -- a
// 60 000-step run with the planner's timers on fell from 10 000 to 500 steps/s): once a few dozen pairs are pending, the ones
// at the front that have completed are read and go back to the pool.
static void timer_harvest(bl_timer& t)
{
    if (t.pending.size() < 64) return;
    size_t done = 0;
    while (done < t.pending.size() && hipEventQuery(t.pending[done].second) == hipSuccess) {
        float ms = 0;
        if (hipEventElapsedTime(&ms, t.pending[done].first, t.pending[done].second) == hipSuccess) { t.total_ms += ms; t.launches += 1; }
        t.pool.push_back(t.pending[done]);
        ++done;
    }
    (void)hipGetLastError();                                 // hipErrorNotReady of the first unfinished pair is not an error
    if (done) t.pending.erase(t.pending.begin(), t.pending.begin() + (long)done);
}

int bl_timer_commit(bl_ctx* ctx, int id, hipEvent_t a, hipEvent_t b)
{
    if (!a) return BL_OK;
    ctx->timers[id].pending.emplace_back(a, b);
    timer_harvest(ctx->timers[id]);
    return BL_OK;
}

int bl_timer_end(bl_ctx* ctx, int id, hipEvent_t a, hipEvent_t b)
{
    if (!ctx->timing || !a) return BL_OK;
    BL_HIP(hipEventRecord(b, ctx->stream));
    ctx->timers[id].pending.emplace_back(a, b);
    timer_harvest(ctx->timers[id]);
    return BL_OK;
}

static int timer_drain(bl_ctx* ctx, int id)
{
    bl_timer& t = ctx->timers[id];
    for (auto& p : t.pending) {
        BL_HIP(hipEventSynchronize(p.second));
        float ms = 0;
        BL_HIP(hipEventElapsedTime(&ms, p.first, p.second));
        t.total_ms += ms;
        t.launches += 1;
        t.pool.push_back(p);
    }
    t.pending.clear();
    return BL_OK;
}

extern "C" int bl_ctx_timing_get(bl_ctx* ctx, int kernel_id, double* total_ms, int64_t* launches)
{
    BL_CHECK_ARG(ctx != nullptr && kernel_id >= 0 && kernel_id < BL_K_COUNT);
    int rc = timer_drain(ctx, kernel_id);
    if (rc) return rc;
    if (total_ms) *total_ms = ctx->timers[kernel_id].total_ms;
    if (launches) *launches = ctx->timers[kernel_id].launches;
    return BL_OK;
}

extern "C" int bl_ctx_timing_reset(bl_ctx* ctx)
{
    BL_CHECK_ARG(ctx != nullptr);
    for (int i = 0; i < BL_K_COUNT; ++i) {
        int rc = timer_drain(ctx, i);
        if (rc) return rc;
        ctx->timers[i].total_ms = 0;
        ctx->timers[i].launches = 0;
    }
    return BL_OK;
}

// ---------------------------------------------------------------- scan staging
// One lidar_t (lcmtypes/lidar_t.lcm) is packed [times | ranges | thetas] (kept rays only) into a pinned slot and pulled
// into the ctx's device block by a one-workgroup kernel.  A scan identical to the previous one (same kept rays, bit for
// bit: the two calls of one SLAM step) is not fetched again.
static const int kScanSlots = 16;

// The scan block is pulled out of the pinned slot by a kernel instead of a hipMemcpyAsync: on the SLAM stream every
// kernel <-> copy-engine transition costs ~7 us of queue handshake, a kernel -> kernel transition costs none.
// When the block has been read, the kernel publishes its sequence number to a pinned word: the host reuses a slot once
// the number has passed the slot's last use -- no HIP event (an event record costs ~6.5 us of stream time here).
__global__ __launch_bounds__(256) void k_scan_fetch(const int64_t* __restrict__ h_times, const float* __restrict__ h_ranges,
                                                    const float* __restrict__ h_thetas, int kept, int64_t* __restrict__ d_times,
                                                    float* __restrict__ d_ranges, float* __restrict__ d_thetas,
                                                    unsigned long long* h_seq, unsigned long long seq)
{
    for (int i = threadIdx.x; i < kept; i += 256) {
        d_times[i] = h_times[i];
        d_ranges[i] = h_ranges[i];
        d_thetas[i] = h_thetas[i];
    }
    __syncthreads();                                    // every lane's loads from the slot have returned (their data was stored)
    if (threadIdx.x == 0) __hip_atomic_store(h_seq, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}

struct bl_scan_slots {
    void* host[kScanSlots];
    void* host_dev[kScanSlots];          // the slot's address as the device sees it
    unsigned long long slot_seq[kScanSlots];   // sequence number of the fetch that last read the slot (0: never used)
    unsigned long long seq;              // fetches launched so far
    unsigned long long* h_seq;           // pinned: sequence number of the last COMPLETED fetch (written by the kernel)
    unsigned long long* h_seq_dev;
    int next;
    int last;                            // slot holding the scan that is in the device block (-1: none)
};

// wait until fetch number `seq` has read its slot (fetches complete in stream order)
static int scan_wait_seq(bl_scan_slots* sl, unsigned long long seq)
{
    if (seq == 0) return BL_OK;
    // (patience by the clock, not by the count of polls: a fetch may sit behind another stream's long kernel on a shared hardware
    // queue; two minutes of nothing is a dead device)
    long spins = 0;
    std::chrono::steady_clock::time_point t0;
    while (__atomic_load_n(sl->h_seq, __ATOMIC_ACQUIRE) < seq) {
        if ((++spins & 1023) == 0) {
            sched_yield();
            if (spins == 1024) t0 = std::chrono::steady_clock::now();
            else if ((spins & 0xFFFFF) == 0 && std::chrono::steady_clock::now() - t0 > std::chrono::seconds(120)) {
                bl_set_error("scan staging: fetch %llu never completed", seq);
                return BL_ERR_STATE;
            }
        }
    }
    return BL_OK;
}

// Packs the kept rays of `scan` into the next free pinned slot (not yet taken: the caller advances `next` when it uses it)
// and says whether the device block already holds exactly this scan.
static int scan_pack(bl_ctx* ctx, const bl_lidar_t* scan, int* kept_out, float* max_range_out, bool* simple_out, bool* resident)
{
    bl_scan_dev* sd = &ctx->scan;
    const int R = scan->num_ranges;
    if (R > sd->capacity) {
        int cap = R < 512 ? 512 : R;
        BL_HIP(hipStreamSynchronize(ctx->stream));
        bl_scan_free(ctx);
        size_t bytes = (size_t)cap * (8 + 4 + 4);
        char* d = nullptr;
        BL_HIP(hipMalloc((void**)&d, 2 * bytes));                // two blocks: the current one and the prefetch target
        sd->capacity = cap;
        sd->base = d;
        sd->times = (int64_t*)d;                                 // 8-byte aligned part first
        sd->ranges = (float*)(d + (size_t)cap * 8);
        sd->thetas = (float*)(d + (size_t)cap * 12);
        sd->alt_times = (int64_t*)(d + bytes);
        sd->alt_ranges = (float*)(d + bytes + (size_t)cap * 8);
        sd->alt_thetas = (float*)(d + bytes + (size_t)cap * 12);
        sd->pre_pending = false;
        bl_scan_slots* sl = new bl_scan_slots();
        memset(sl, 0, sizeof(*sl));
        sl->last = -1;
        sd->staging_bytes = bytes;
        for (int i = 0; i < kScanSlots; ++i) {
            BL_HIP(hipHostMalloc(&sl->host[i], bytes, hipHostMallocDefault));
            BL_HIP(hipHostGetDevicePointer(&sl->host_dev[i], sl->host[i], 0));
        }
        BL_HIP(hipHostMalloc((void**)&sl->h_seq, 64, hipHostMallocDefault));
        *sl->h_seq = 0;
        BL_HIP(hipHostGetDevicePointer((void**)&sl->h_seq_dev, sl->h_seq, 0));
        sd->staging = sl;
    }
    bl_scan_slots* sl = (bl_scan_slots*)sd->staging;
    const int s = sl->next;
    { int wrc = scan_wait_seq(sl, sl->slot_seq[s]); if (wrc) return wrc; }
    char* h = (char*)sl->host[s];
    const size_t cap = (size_t)sd->capacity;
    int64_t* ht = (int64_t*)h;
    float* hrange = (float*)(h + cap * 8);
    float* htheta = (float*)(h + cap * 12);
    // MovingLaserScan keeps a ray only if its range exceeds 0.15f (moving_laser_scan.cpp:24); the kept rays are packed
    // in scan order, so no kernel branches on validity.
    int kept = 0;
    float max_range = 0;
    bool thetas_simple = true;
    for (int n = 0; n < R; ++n) {
        if (!(scan->ranges[n] > 0.15f)) continue;
        if (!(scan->thetas[n] >= 0.0f && scan->thetas[n] <= BL_THETA_SIMPLE_MAX)) thetas_simple = false;
        if (scan->ranges[n] > max_range) max_range = scan->ranges[n];
        hrange[kept] = scan->ranges[n];
        htheta[kept] = scan->thetas[n];
        ht[kept] = scan->times[n];
        ++kept;
    }
    *kept_out = kept; *max_range_out = max_range; *simple_out = thetas_simple;
    *resident = false;
    if (kept > 0 && sl->last >= 0 && sd->kept == kept) {          // the same scan as the one in the device block?
        const char* p = (const char*)sl->host[sl->last];
        if (memcmp(p, h, (size_t)kept * 8) == 0 && memcmp(p + cap * 8, hrange, (size_t)kept * 4) == 0 &&
            memcmp(p + cap * 12, htheta, (size_t)kept * 4) == 0)
            *resident = true;                                    // the slot stays free for the next scan
    }
    return BL_OK;
}

int bl_scan_upload(bl_ctx* ctx, const bl_lidar_t* scan, int* num_rays)
{
    BL_CHECK_ARG(scan != nullptr && scan->num_ranges >= 0);
    bl_scan_dev* sd = &ctx->scan;
    *num_rays = 0;
    if (scan->num_ranges == 0) { sd->kept = 0; if (sd->staging) ((bl_scan_slots*)sd->staging)->last = -1; return BL_OK; }
    BL_CHECK_ARG(scan->ranges && scan->thetas && scan->times);
    int kept = 0; float max_range = 0; bool thetas_simple = false, resident = false;
    int prc = scan_pack(ctx, scan, &kept, &max_range, &thetas_simple, &resident);
    if (prc) return prc;
    bl_scan_slots* sl = (bl_scan_slots*)sd->staging;
    const int s = sl->next;
    const size_t cap = (size_t)sd->capacity;
    *num_rays = kept;
    if (kept == 0) { sd->kept = 0; sl->last = -1; return BL_OK; }
    if (resident) return BL_OK;
    sd->pre_pending = false;                                     // a packed scan nobody has carried over by now is dropped
    const char* hd = (const char*)sl->host_dev[s];
    hipLaunchKernelGGL(k_scan_fetch, dim3(1), dim3(256), 0, ctx->stream, (const int64_t*)hd, (const float*)(hd + cap * 8),
                       (const float*)(hd + cap * 12), kept, sd->times, sd->ranges, sd->thetas, sl->h_seq_dev, sl->seq + 1);
    BL_HIP(hipGetLastError());
    sl->seq += 1;
    sl->slot_seq[s] = sl->seq;
    sl->last = s;
    sl->next = (s + 1) % kScanSlots;
    sd->kept = kept;
    sd->max_range = max_range;
    sd->thetas_simple = thetas_simple;
    return BL_OK;
}

// The NEXT scan, handed over early (a SLAM host has it queued: slam.cpp:96-104): packed into a pinned slot now, brought to
// the second device block by the next map kernel of this ctx beside its own work (bl_scan_prefetch_take), so that the
// bl_pf_update / bl_mapping_update call that brings the same scan finds it resident and launches no fetch kernel.
// Handing over a scan that is not used next costs nothing but the packing.
extern "C" int bl_scan_prefetch(bl_ctx* ctx, const bl_lidar_t* scan)
{
    BL_CHECK_ARG(ctx != nullptr && scan != nullptr && scan->num_ranges >= 0);
    bl_scan_dev* sd = &ctx->scan;
    sd->pre_pending = false;
    if (scan->num_ranges == 0) return BL_OK;
    BL_CHECK_ARG(scan->ranges && scan->thetas && scan->times);
    BL_HIP(hipSetDevice(ctx->device));
    int kept = 0; float max_range = 0; bool thetas_simple = false, resident = false;
    int rc = scan_pack(ctx, scan, &kept, &max_range, &thetas_simple, &resident);
    if (rc) return rc;
    if (kept == 0 || resident) return BL_OK;
    bl_scan_slots* sl = (bl_scan_slots*)sd->staging;
    sd->pre_slot = sl->next;
    sl->next = (sl->next + 1) % kScanSlots;                      // the slot is taken until the copy (if any) has read it
    sd->pre_kept = kept; sd->pre_max_range = max_range; sd->pre_thetas_simple = thetas_simple;
    sd->pre_pending = true;
    return BL_OK;
}

int bl_scan_prefetch_take(bl_ctx* ctx, bl_scan_prefetch_args* out)
{
    bl_scan_dev* sd = &ctx->scan;
    if (!sd->pre_pending || !sd->staging) return 0;
    bl_scan_slots* sl = (bl_scan_slots*)sd->staging;
    const size_t cap = (size_t)sd->capacity;
    const char* hd = (const char*)sl->host_dev[sd->pre_slot];
    out->h_times = (const int64_t*)hd; out->h_ranges = (const float*)(hd + cap * 8); out->h_thetas = (const float*)(hd + cap * 12);
    out->kept = sd->pre_kept;
    out->d_times = sd->alt_times; out->d_ranges = sd->alt_ranges; out->d_thetas = sd->alt_thetas;
    out->h_seq = sl->h_seq_dev; out->seq = sl->seq + 1;
    sl->seq += 1;
    sl->slot_seq[sd->pre_slot] = sl->seq;
    sl->last = sd->pre_slot;
    // the blocks swap: everything launched from here on reads the prefetched scan
    int64_t* t = sd->times; sd->times = sd->alt_times; sd->alt_times = t;
    float* r = sd->ranges; sd->ranges = sd->alt_ranges; sd->alt_ranges = r;
    float* th = sd->thetas; sd->thetas = sd->alt_thetas; sd->alt_thetas = th;
    sd->kept = sd->pre_kept; sd->max_range = sd->pre_max_range; sd->thetas_simple = sd->pre_thetas_simple;
    sd->pre_pending = false;
    return 1;
}

void bl_scan_free(bl_ctx* ctx)
{
    bl_scan_dev* sd = &ctx->scan;
    if (sd->base) (void)hipFree(sd->base);
    if (sd->staging) {
        bl_scan_slots* sl = (bl_scan_slots*)sd->staging;
        (void)scan_wait_seq(sl, sl->seq);                 // no fetch still reads a slot
        for (int i = 0; i < kScanSlots; ++i)
            if (sl->host[i]) (void)hipHostFree(sl->host[i]);
        if (sl->h_seq) (void)hipHostFree(sl->h_seq);
        delete sl;
    }
    sd->capacity = 0; sd->kept = 0;
    sd->times = nullptr; sd->ranges = nullptr; sd->thetas = nullptr; sd->staging = nullptr;
    sd->base = nullptr; sd->alt_times = nullptr; sd->alt_ranges = nullptr; sd->alt_thetas = nullptr; sd->pre_pending = false;
}

// plain device allocations for a host that has no allocator of its own at hand (botlab_amd/sharded.py probes whether the ranks
// can map each other's memory before it commits to the composed finish)
extern "C" int bl_dev_alloc(bl_ctx* ctx, size_t bytes, void** out)
{
    BL_CHECK_ARG(ctx != nullptr && out != nullptr && bytes > 0);
    BL_HIP(hipSetDevice(ctx->device));
    BL_HIP(hipMalloc(out, bytes));
    BL_HIP(hipMemsetAsync(*out, 0, bytes, ctx->stream));
    BL_HIP(hipStreamSynchronize(ctx->stream));
    return BL_OK;
}
extern "C" int bl_dev_free(void* p)
{
    if (p) BL_HIP(hipFree(p));
    return BL_OK;
}

// One 32-bit word of device memory written (write != 0) or read by a KERNEL of this context's device -- the access the particle
// filter's kernels make to another rank's arrays: a mapping that opens but cannot be read this way must not be trusted.
__global__ void k_dev_word(unsigned int* p, int write, unsigned int v, unsigned int* out)
{
    if (write) __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    else *out = __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}
extern "C" int bl_dev_word(bl_ctx* ctx, void* dev_ptr, int write, uint32_t* value)
{
    BL_CHECK_ARG(ctx != nullptr && dev_ptr != nullptr && value != nullptr);
    BL_HIP(hipSetDevice(ctx->device));
    unsigned int* out = nullptr;
    BL_HIP(hipMalloc((void**)&out, sizeof(unsigned int)));
    hipLaunchKernelGGL(k_dev_word, dim3(1), dim3(1), 0, ctx->stream, (unsigned int*)dev_ptr, write, (unsigned int)*value, out);
    hipError_t e = hipGetLastError();
    if (e == hipSuccess && !write) e = hipMemcpyAsync(value, out, sizeof(unsigned int), hipMemcpyDeviceToHost, ctx->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
    (void)hipFree(out);
    if (e != hipSuccess) { bl_set_error("bl_dev_word: %s", hipGetErrorString(e)); return BL_ERR_HIP; }
    return BL_OK;
}

// ---------------------------------------------------------------- grid lineage + dirty log (bl_internal.h)
#include <atomic>
static std::atomic<uint64_t> g_next_lineage{1};

static void log_release(bl_dirty_log* l)
{
    if (l && --l->refs == 0) { if (l->dev) (void)hipFree(l->dev); delete l; }
}

uint64_t bl_grid_new_lineage(bl_grid* g)
{
    g->id = g_next_lineage.fetch_add(1);
    g->version = 0;
    log_release(g->log);
    g->log = nullptr;
    return g->id;
}

uint64_t bl_grid_lineage_id(const bl_grid* g)
{
    if (g->id == 0) const_cast<bl_grid*>(g)->id = g_next_lineage.fetch_add(1);
    return g->id;
}

void bl_grid_adopt_lineage(bl_grid* snap, const bl_grid* src)
{
    (void)bl_grid_lineage_id(src);
    snap->id = src->id;
    snap->version = src->version;
    if (snap->log != src->log) {
        log_release(snap->log);
        snap->log = src->log;
        if (snap->log) snap->log->refs += 1;
    }
}

int4* bl_grid_log_next(bl_grid* g, uint64_t* version)
{
    if (g->id == 0) g->id = g_next_lineage.fetch_add(1);
    if (!g->log) {
        bl_dirty_log* l = new bl_dirty_log();
        l->refs = 1; l->dev = nullptr;
        if (hipMalloc((void**)&l->dev, sizeof(int4) * BL_DIRTY_LOG) != hipSuccess ||
            hipMemsetAsync(l->dev, 0xff, sizeof(int4) * BL_DIRTY_LOG, g->ctx->stream) != hipSuccess) {
            // no log: every consumer of this lineage takes the full transform (a version without an entry never matches its tag)
            if (l->dev) (void)hipFree(l->dev);
            delete l;
            (void)bl_grid_new_lineage(g);
            *version = 0;
            return nullptr;
        }
        g->log = l;
    }
    g->version += 1;
    *version = g->version;
    return g->log->dev + (g->version % BL_DIRTY_LOG);
}

// ---------------------------------------------------------------- OccupancyGrid (src/slam/occupancy_grid.cpp)
extern "C" int bl_grid_create(bl_ctx* ctx, int width, int height, float meters_per_cell, float cells_per_meter,
                              float origin_x, float origin_y, bl_grid** out)
{
    BL_CHECK_ARG(ctx != nullptr && out != nullptr);
    BL_CHECK_ARG(width > 0 && height > 0 && (int64_t)width * height < (1ll << 31));
    BL_HIP(hipSetDevice(ctx->device));
    bl_grid* g = new bl_grid();
    g->ctx = ctx;
    g->frame.width = width; g->frame.height = height;
    g->frame.mpc = meters_per_cell; g->frame.cpm = cells_per_meter;
    g->frame.ox = origin_x; g->frame.oy = origin_y;
    hipError_t e = hipMalloc((void**)&g->cells, (size_t)width * height);
    if (e != hipSuccess) {
        bl_set_error("hipMalloc(grid %dx%d) failed: %s", width, height, hipGetErrorString(e));
        delete g;
        return BL_ERR_HIP;
    }
    BL_HIP(hipMemsetAsync(g->cells, 0, (size_t)width * height, ctx->stream));
    *out = g;
    return BL_OK;
}

extern "C" void bl_grid_destroy(bl_grid* g)
{
    if (!g) return;
    (void)hipStreamSynchronize(g->ctx->stream);
    (void)hipFree(g->cells);
    if (g->mirror) (void)hipFree(g->mirror);
    log_release(g->log);
    delete g;
}

extern "C" int bl_grid_upload(bl_grid* g, const int8_t* cells)
{
    BL_CHECK_ARG(g != nullptr && cells != nullptr);
    size_t n = (size_t)g->frame.width * g->frame.height;
    g->mirror_valid = false;
    (void)bl_grid_new_lineage(g);
    BL_HIP(hipMemcpyAsync(g->cells, cells, n, hipMemcpyHostToDevice, g->ctx->stream));
    BL_HIP(hipStreamSynchronize(g->ctx->stream));    // the host buffer is caller-owned and may be reused at once
    return BL_OK;
}

extern "C" int bl_grid_download(bl_grid* g, int8_t* cells)
{
    BL_CHECK_ARG(g != nullptr && cells != nullptr);
    size_t n = (size_t)g->frame.width * g->frame.height;
    BL_HIP(hipMemcpyAsync(cells, g->cells, n, hipMemcpyDeviceToHost, g->ctx->stream));
    BL_HIP(hipStreamSynchronize(g->ctx->stream));
    return BL_OK;
}

extern "C" int bl_grid_reset(bl_grid* g)
{
    BL_CHECK_ARG(g != nullptr);
    g->mirror_valid = false;
    (void)bl_grid_new_lineage(g);
    BL_HIP(hipMemsetAsync(g->cells, 0, (size_t)g->frame.width * g->frame.height, g->ctx->stream));
    return BL_OK;
}

extern "C" int bl_grid_set_frame(bl_grid* g, float meters_per_cell, float cells_per_meter, float origin_x, float origin_y)
{
    BL_CHECK_ARG(g != nullptr);
    g->frame.mpc = meters_per_cell; g->frame.cpm = cells_per_meter;
    g->frame.ox = origin_x; g->frame.oy = origin_y;
    return BL_OK;
}

extern "C" int bl_grid_copy(bl_grid* dst, const bl_grid* src)
{
    BL_CHECK_ARG(dst != nullptr && src != nullptr);
    BL_CHECK_ARG(dst->frame.width == src->frame.width && dst->frame.height == src->frame.height);
    dst->frame = src->frame;
    dst->mirror_valid = false;
    (void)bl_grid_new_lineage(dst);               // (a plain copy runs on dst's stream, unordered against src's later updates: no shared log)
    BL_HIP(hipMemcpyAsync(dst->cells, src->cells, (size_t)src->frame.width * src->frame.height,
                          hipMemcpyDeviceToDevice, dst->ctx->stream));
    return BL_OK;
}

extern "C" void* bl_grid_device_ptr(bl_grid* g)
{
    if (!g) return nullptr;
    g->mirror_external = true;          // the caller may write the cells without the library seeing it
    g->mirror_valid = false;
    (void)bl_grid_new_lineage(g);
    return (void*)g->cells;
}

extern "C" int bl_grid_shape(const bl_grid* g, int* width, int* height)
{
    BL_CHECK_ARG(g != nullptr);
    if (width) *width = g->frame.width;
    if (height) *height = g->frame.height;
    return BL_OK;
}
