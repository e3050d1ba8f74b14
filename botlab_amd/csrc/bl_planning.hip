// bl_planning.hip -- ObstacleDistanceGrid::setDistances (src/planning/obstacle_distance_grid.cpp:73-181) and
// search_for_path (src/planning/astar.cpp:9-274) as gfx950 kernels.
//
// Distance grid.  The reference floods a 4-connected min-heap from every cell with log-odds >= 0 (free cells start at
// -1) and gives each newly reached cell parent + 0.1f.  Pops come in non-decreasing distance, so a cell whose
// 4-connected (L1) distance to the nearest non-free cell is n receives f[n], f[0] = 0, f[n] = f[n-1] + 0.1f (float);
// a map with no non-free cell keeps -1 everywhere.  The kernels compute the exact integer L1 distance transform
// (separable: nearest source within the row, then a min-plus sweep down and up the columns) and map it through the
// float table f -- bit-identical floats, no heap.
//
// A*.  The reference's result depends on the pop order of libstdc++'s binary heap among equal fCost and on its
// re-expansion of duplicate open-list entries, so the search is executed with exactly those heap index operations
// (std::push_heap / std::pop_heap semantics, stl_heap.h) by one wavefront; the closed list is an int32 parent grid
// (first closing of a cell wins, which is all is_member/get_member ever observe).  Single-search latency is bound by
// dependent memory accesses, not bandwidth (DESIGN.md "A*").
#include <math.h>
#include <string.h>

#include "bl_internal.h"

// =============================================================================================== distance grid
struct bl_dist {
    bl_ctx* ctx;
    bl_frame frame;
    size_t capacity;          // cells allocated
    uint16_t* row;            // per-row nearest-source distance (0xFFFF: none in row)
    uint16_t* l1;             // L1 distance (0xFFFF: no source anywhere)
    float* cells;             // float distances handed to callers
    float* lut;               // device f[n]
    int lut_n;
    std::vector<float>* lut_host;
    bool valid;
};

#define DIST_INF (1 << 28)

// Row pass: one workgroup per row; d_row[x] = min over sources x' in the row of |x - x'|.
__global__ __launch_bounds__(256) void k_dist_rows(const int8_t* __restrict__ cells, int W, uint16_t* __restrict__ row)
{
    __shared__ int s_wave[4];
    __shared__ int s_carry;
    const int y = blockIdx.x;
    const int8_t* c = cells + (size_t)y * W;
    uint16_t* out = row + (size_t)y * W;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    // forward: nearest source at or left of x
    if (threadIdx.x == 0) s_carry = -DIST_INF;
    __syncthreads();
    for (int base = 0; base < W; base += 256) {
        int x = base + threadIdx.x;
        int v = (x < W && c[x] >= 0) ? x : -DIST_INF;          // is_cell_occupied: logOdds >= 0 (obstacle_distance_grid.cpp:125-128)
        int incl = v;
        for (int off = 1; off < 64; off <<= 1) {
            int t = __shfl_up(incl, off, 64);
            if (lane >= off) incl = max(incl, t);
        }
        if (lane == 63) s_wave[wave] = incl;
        __syncthreads();
        int best = max(incl, s_carry);
        for (int w = 0; w < wave; ++w) best = max(best, s_wave[w]);
        if (x < W) out[x] = (uint16_t)min(x - best, 0xFFFF);
        __syncthreads();
        if (threadIdx.x == 255) s_carry = best;
        __syncthreads();
    }
    // backward: nearest source at or right of x
    if (threadIdx.x == 0) s_carry = DIST_INF;
    __syncthreads();
    const int nseg = (W + 255) / 256;
    for (int seg = nseg - 1; seg >= 0; --seg) {
        int x = seg * 256 + threadIdx.x;
        int v = (x < W && c[x] >= 0) ? x : DIST_INF;
        int incl = v;                                           // suffix min within the wave
        for (int off = 1; off < 64; off <<= 1) {
            int t = __shfl_down(incl, off, 64);
            if (lane + off < 64) incl = min(incl, t);
        }
        if (lane == 0) s_wave[wave] = incl;
        __syncthreads();
        int best = min(incl, s_carry);
        for (int w = wave + 1; w < 4; ++w) best = min(best, s_wave[w]);
        if (x < W) {
            int d = min(best - x, 0xFFFF);
            out[x] = (uint16_t)min((int)out[x], d);
        }
        __syncthreads();
        if (threadIdx.x == 0) s_carry = best;
        __syncthreads();
    }
}

// Column pass: a workgroup owns 64 columns; its 16 thread rows split the H rows into strips.  d[y] = min(row[y],
// d[y-1]+1) downwards and the mirror upwards; strips are chained through LDS summaries.
#define DCOL_TX 64
#define DCOL_TY 16
__global__ __launch_bounds__(DCOL_TX * DCOL_TY) void k_dist_cols(const uint16_t* __restrict__ row, int W, int H,
                                                                 uint16_t* __restrict__ l1, float* __restrict__ out,
                                                                 const float* __restrict__ lut)
{
    __shared__ int s_fwd[DCOL_TY][DCOL_TX];
    __shared__ int s_bwd[DCOL_TY][DCOL_TX];
    const int tx = threadIdx.x, ty = threadIdx.y;
    const int x = blockIdx.x * DCOL_TX + tx;
    const int rows_per = (H + DCOL_TY - 1) / DCOL_TY;
    const int y0 = min(ty * rows_per, H), y1 = min(y0 + rows_per, H);
    const bool live = x < W;
    int a_f = DIST_INF, a_b = DIST_INF;
    if (live) {
        for (int y = y0; y < y1; ++y) {
            int g = row[(size_t)y * W + x];
            if (g == 0xFFFF) g = DIST_INF;
            a_f = min(a_f, g + (y1 - 1 - y));
            a_b = min(a_b, g + (y - y0));
        }
    }
    s_fwd[ty][tx] = a_f;
    s_bwd[ty][tx] = a_b;
    __syncthreads();
    if (!live) return;
    // E = distance at the last row of the previous strip (downward chain); B = at the first row of the next strip
    int E = DIST_INF;
    for (int s = 0; s < ty; ++s) {
        int sy0 = min(s * rows_per, H), sy1 = min(sy0 + rows_per, H);
        E = min(s_fwd[s][tx], E + (sy1 - sy0));
    }
    int B = DIST_INF;
    for (int s = DCOL_TY - 1; s > ty; --s) {
        int sy0 = min(s * rows_per, H), sy1 = min(sy0 + rows_per, H);
        B = min(s_bwd[s][tx], B + (sy1 - sy0));
    }
    int d = E;
    for (int y = y0; y < y1; ++y) {
        int g = row[(size_t)y * W + x];
        if (g == 0xFFFF) g = DIST_INF;
        d = min(g, d + 1);
        l1[(size_t)y * W + x] = (uint16_t)min(d, 0xFFFF);
    }
    int b = B;
    for (int y = y1 - 1; y >= y0; --y) {
        int g = row[(size_t)y * W + x];
        if (g == 0xFFFF) g = DIST_INF;
        b = min(g, b + 1);
        int f = l1[(size_t)y * W + x];
        if (f == 0xFFFF) f = DIST_INF;
        int v = min(f, b);
        bool none = v >= 0xFFFF;
        l1[(size_t)y * W + x] = none ? (uint16_t)0xFFFF : (uint16_t)v;
        out[(size_t)y * W + x] = none ? -1.0f : lut[v];
    }
}

extern "C" int bl_dist_create(bl_ctx* ctx, bl_dist** out)
{
    BL_CHECK_ARG(ctx != nullptr && out != nullptr);
    bl_dist* d = new bl_dist();
    memset((void*)d, 0, sizeof(*d));
    d->ctx = ctx;
    d->frame.mpc = 0.05f; d->frame.cpm = 20.0f;          // ObstacleDistanceGrid() (obstacle_distance_grid.cpp:31-37)
    d->lut_host = new std::vector<float>();
    *out = d;
    return BL_OK;
}

extern "C" void bl_dist_destroy(bl_dist* d)
{
    if (!d) return;
    (void)hipStreamSynchronize(d->ctx->stream);
    if (d->row) (void)hipFree(d->row);
    if (d->l1) (void)hipFree(d->l1);
    if (d->cells) (void)hipFree(d->cells);
    if (d->lut) (void)hipFree(d->lut);
    delete d->lut_host;
    delete d;
}

extern "C" int bl_dist_set_distances(bl_dist* d, const bl_grid* map)
{
    BL_CHECK_ARG(d != nullptr && map != nullptr);
    bl_ctx* ctx = d->ctx;
    BL_HIP(hipSetDevice(ctx->device));
    const int W = map->frame.width, H = map->frame.height;
    BL_CHECK_ARG(W + H < 0xFFFF);
    size_t n = (size_t)W * H;
    if (n > d->capacity) {                                // resetGrid (obstacle_distance_grid.cpp:100-118)
        BL_HIP(hipStreamSynchronize(ctx->stream));
        if (d->row) BL_HIP(hipFree(d->row));
        if (d->l1) BL_HIP(hipFree(d->l1));
        if (d->cells) BL_HIP(hipFree(d->cells));
        d->row = nullptr; d->l1 = nullptr; d->cells = nullptr;
        BL_HIP(hipMalloc((void**)&d->row, n * 2));
        BL_HIP(hipMalloc((void**)&d->l1, n * 2));
        BL_HIP(hipMalloc((void**)&d->cells, n * 4));
        d->capacity = n;
    }
    d->frame = map->frame;
    if (d->lut_n < W + H + 1) {
        BL_HIP(hipStreamSynchronize(ctx->stream));
        if (d->lut) BL_HIP(hipFree(d->lut));
        d->lut = nullptr;
        int ln = W + H + 1;
        d->lut_host->resize(ln);
        float f = 0.0f;                                   // f[n] = f[n-1] + 0.1f (obstacle_distance_grid.cpp:174)
        for (int i = 0; i < ln; ++i) { (*d->lut_host)[i] = f; f = f + 0.1f; }
        BL_HIP(hipMalloc((void**)&d->lut, (size_t)ln * 4));
        BL_HIP(hipMemcpy(d->lut, d->lut_host->data(), (size_t)ln * 4, hipMemcpyHostToDevice));
        d->lut_n = ln;
    }
    hipEvent_t e0, e1;
    int rc = bl_timer_begin(ctx, BL_K_DIST, &e0, &e1);
    if (rc) return rc;
    hipLaunchKernelGGL(k_dist_rows, dim3(H), dim3(256), 0, ctx->stream, map->cells, W, d->row);
    hipLaunchKernelGGL(k_dist_cols, dim3((W + DCOL_TX - 1) / DCOL_TX), dim3(DCOL_TX, DCOL_TY), 0, ctx->stream, d->row, W, H,
                       d->l1, d->cells, d->lut);
    BL_HIP(hipGetLastError());
    rc = bl_timer_end(ctx, BL_K_DIST, e0, e1);
    if (rc) return rc;
    d->valid = true;
    return BL_OK;
}

extern "C" int bl_dist_download(bl_dist* d, float* cells)
{
    BL_CHECK_ARG(d != nullptr && cells != nullptr && d->valid);
    BL_HIP(hipMemcpyAsync(cells, d->cells, (size_t)d->frame.width * d->frame.height * 4, hipMemcpyDeviceToHost, d->ctx->stream));
    BL_HIP(hipStreamSynchronize(d->ctx->stream));
    return BL_OK;
}

extern "C" int bl_dist_shape(const bl_dist* d, int* width, int* height)
{
    BL_CHECK_ARG(d != nullptr);
    if (width) *width = d->frame.width;
    if (height) *height = d->frame.height;
    return BL_OK;
}

extern "C" int bl_dist_frame(const bl_dist* d, float* mpc, float* cpm, float* ox, float* oy)
{
    BL_CHECK_ARG(d != nullptr);
    if (mpc) *mpc = d->frame.mpc;
    if (cpm) *cpm = d->frame.cpm;
    if (ox) *ox = d->frame.ox;
    if (oy) *oy = d->frame.oy;
    return BL_OK;
}

extern "C" void* bl_dist_device_ptr(bl_dist* d) { return d ? (void*)d->cells : nullptr; }

// =============================================================================================== A*
#define ASTAR_INVALID_COST INT32_MIN
#define ASTAR_ST_FOUND 0
#define ASTAR_ST_NOPATH 1        // early exit or open list exhausted: 1-pose path
#define ASTAR_ST_CAPACITY 2
#define ASTAR_ST_LIMIT 3

struct astar_result { int status; int path_len; long long pops; long long pushes; };

struct bl_astar_state {
    int4* heap; int64_t heap_cap;
    int32_t* closed; size_t closed_cap;
    int32_t* path; size_t path_cap;
    int32_t* cost_lut; int cost_lut_cap;
    astar_result* d_result;
    astar_result* h_result;            // pinned
    int32_t* h_cost;                   // pinned staging for the cost table
    // pending search (async form)
    bool pending;
    bl_pose_xyt_t start;
    bl_frame frame;
};

struct astar_args {
    const uint16_t* l1; int W, H;
    const int32_t* cost_lut;           // per L1 distance: obstacle cost, or ASTAR_INVALID_COST if the cell is not valid
    int4* heap; long long heap_cap;
    int32_t* closed;
    int32_t* path; long long path_cap;
    astar_result* result;
    int sx, sy, gx, gy;
    long long max_pops;
};

__device__ __forceinline__ int astar_cell_cost(const astar_args& a, int x, int y)
{
    // isValid (astar.cpp:140-149, D5) folded with get_oCost (astar.cpp:181-186): both depend only on the cell's
    // distance value, i.e. on its integer L1 distance
    if (x < 0 || y < 0 || x >= a.W || y >= a.H) return ASTAR_INVALID_COST;
    int n = a.l1[(size_t)y * a.W + x];
    if (n == 0xFFFF) return ASTAR_INVALID_COST;          // distance -1: never > minDist
    return a.cost_lut[n];
}

// node layout: x = fCost, y = gCost, z = cell index, w = parent cell index
__global__ __launch_bounds__(64) void k_astar(astar_args a)
{
    if (threadIdx.x != 0) return;                        // v1: the heap is walked by one lane (see DESIGN.md "A*")
    astar_result res; res.status = ASTAR_ST_NOPATH; res.path_len = 0; res.pops = 0; res.pushes = 0;
    const int start = a.sy * a.W + a.sx, goal = a.gy * a.W + a.gx;
    bool ok = astar_cell_cost(a, a.gx, a.gy) != ASTAR_INVALID_COST       // astar.cpp:40-44
              && astar_cell_cost(a, a.sx, a.sy) != ASTAR_INVALID_COST    // :46-50
              && !(a.sx == a.gx && a.sy == a.gy);                        // :52-56
    if (!ok) { *a.result = res; return; }
    int4* h = a.heap;
    long long len = 1;
    h[0] = make_int4(0, 0, start, 0);                    // firstNode: costs 0, parent Point() == (0,0)
    const int xD[4] = {1, -1, 0, 0};
    const int yD[4] = {0, 0, 1, -1};
    while (len > 0) {
        if (res.pops >= a.max_pops) { res.status = ASTAR_ST_LIMIT; break; }
        const int4 top = h[0];
        if (a.closed[top.z] < 0) a.closed[top.z] = top.w;             // closedList.push_back: first entry per cell wins
        // ---- std::pop_heap + pop_back (stl_heap.h __pop_heap/__adjust_heap/__push_heap, comp = fCost greater)
        len -= 1;
        if (len > 0) {
            const int4 value = h[len];
            long long hole = 0, child = 0;
            while (child < (len - 1) / 2) {
                child = 2 * (child + 1);
                if (h[child].x > h[child - 1].x) child--;
                h[hole] = h[child];
                hole = child;
            }
            if ((len & 1) == 0 && child == (len - 2) / 2) {
                child = 2 * (child + 1);
                h[hole] = h[child - 1];
                hole = child - 1;
            }
            long long parent = (hole - 1) / 2;
            while (hole > 0 && h[parent].x > value.x) {
                h[hole] = h[parent];
                hole = parent;
                parent = (hole - 1) / 2;
            }
            h[hole] = value;
        }
        res.pops += 1;
        const int cx = top.z % a.W, cy = top.z / a.W;
        bool done = false;
        for (int k = 0; k < 4; ++k) {                                  // expand_node (astar.cpp:213-233)
            const int kx = cx + xD[k], ky = cy + yD[k];
            const int cost = astar_cell_cost(a, kx, ky);
            if (cost == ASTAR_INVALID_COST) continue;                  // off grid or !isValid
            const int kc = ky * a.W + kx;
            if (kc == goal) {                                          // :107-114 -> makePath (:235-274)
                long long n = 0;
                int cell = kc, parent = top.z;
                while (cell != start) {
                    if (n < a.path_cap) a.path[n] = cell;
                    n += 1;
                    cell = parent;
                    parent = a.closed[cell];
                }
                res.status = ASTAR_ST_FOUND;
                res.path_len = (int)n;
                done = true;
                break;
            }
            if (a.closed[kc] >= 0) continue;                           // member of closedList: never pushed (:123)
            const int g = top.y + 10;                                  // get_gCost: 4-connected step
            const int ax = abs(a.gx - kx), ay = abs(a.gy - ky);        // get_hCost (:170-179)
            const int hc = (ax >= ay) ? 14 * ay + 10 * (ax - ay) : 14 * ax + 10 * (ay - ax);
            const int f = g + hc + cost;
            if (32767 > f) {                                           // ngbr.fCost = INT16_MAX > fNew (:103,124)
                if (len >= a.heap_cap) { res.status = ASTAR_ST_CAPACITY; done = true; break; }
                // std::push_heap
                const int4 value = make_int4(f, g, kc, top.z);
                long long hole = len;
                len += 1;
                long long parent = (hole - 1) / 2;
                while (hole > 0 && h[parent].x > value.x) {
                    h[hole] = h[parent];
                    hole = parent;
                    parent = (hole - 1) / 2;
                }
                h[hole] = value;
                res.pushes += 1;
            }
        }
        if (done) break;
    }
    *a.result = res;
}

void bl_astar_free(bl_ctx* ctx)
{
    bl_astar_state* s = ctx->astar;
    if (!s) return;
    if (s->heap) (void)hipFree(s->heap);
    if (s->closed) (void)hipFree(s->closed);
    if (s->path) (void)hipFree(s->path);
    if (s->cost_lut) (void)hipFree(s->cost_lut);
    if (s->d_result) (void)hipFree(s->d_result);
    if (s->h_result) (void)hipHostFree(s->h_result);
    if (s->h_cost) (void)hipHostFree(s->h_cost);
    delete s;
    ctx->astar = nullptr;
}

extern "C" int bl_astar_set_open_capacity(bl_ctx* ctx, int64_t nodes)
{
    BL_CHECK_ARG(ctx != nullptr && nodes >= 0);
    ctx->astar_capacity = nodes;
    return BL_OK;
}

static int astar_prepare(bl_ctx* ctx, const bl_dist* d)
{
    if (!ctx->astar) {
        ctx->astar = new bl_astar_state();
        memset((void*)ctx->astar, 0, sizeof(bl_astar_state));
        BL_HIP(hipMalloc((void**)&ctx->astar->d_result, sizeof(astar_result)));
        BL_HIP(hipHostMalloc((void**)&ctx->astar->h_result, sizeof(astar_result), hipHostMallocDefault));
    }
    bl_astar_state* s = ctx->astar;
    int64_t want = ctx->astar_capacity > 0 ? ctx->astar_capacity : (int64_t)1 << 24;     // 16M nodes = 256 MB
    if (s->heap_cap != want) {
        BL_HIP(hipStreamSynchronize(ctx->stream));
        if (s->heap) BL_HIP(hipFree(s->heap));
        s->heap = nullptr;
        BL_HIP(hipMalloc((void**)&s->heap, (size_t)want * sizeof(int4)));
        s->heap_cap = want;
    }
    size_t n = (size_t)d->frame.width * d->frame.height;
    if (n > s->closed_cap) {
        BL_HIP(hipStreamSynchronize(ctx->stream));
        if (s->closed) BL_HIP(hipFree(s->closed));
        if (s->path) BL_HIP(hipFree(s->path));
        s->closed = nullptr; s->path = nullptr;
        BL_HIP(hipMalloc((void**)&s->closed, n * 4));
        BL_HIP(hipMalloc((void**)&s->path, n * 4));
        s->closed_cap = n; s->path_cap = n;
    }
    int ln = d->frame.width + d->frame.height + 1;
    if (ln > s->cost_lut_cap) {
        BL_HIP(hipStreamSynchronize(ctx->stream));
        if (s->cost_lut) BL_HIP(hipFree(s->cost_lut));
        if (s->h_cost) BL_HIP(hipHostFree(s->h_cost));
        s->cost_lut = nullptr; s->h_cost = nullptr;
        BL_HIP(hipMalloc((void**)&s->cost_lut, (size_t)ln * 4));
        BL_HIP(hipHostMalloc((void**)&s->h_cost, (size_t)ln * 4, hipHostMallocDefault));
        s->cost_lut_cap = ln;
    }
    return BL_OK;
}

extern "C" int bl_astar_search_async(bl_ctx* ctx, const bl_dist* d, const bl_pose_xyt_t* start, const bl_pose_xyt_t* goal,
                                     const bl_search_params_t* params)
{
    BL_CHECK_ARG(ctx != nullptr && d != nullptr && start != nullptr && goal != nullptr && params != nullptr);
    BL_CHECK_ARG(d->valid && d->ctx == ctx);
    BL_HIP(hipSetDevice(ctx->device));
    int rc = astar_prepare(ctx, d);
    if (rc) return rc;
    bl_astar_state* s = ctx->astar;
    if (s->pending) { bl_set_error("an A* search is already pending on this ctx"); return BL_ERR_STATE; }
    // per-distance cell table: validity (astar.cpp:141) and obstacle cost (astar.cpp:181-186) from the float value
    // f[n] a cell at L1 distance n holds.  The host's pow() is the reference's pow().
    BL_HIP(hipStreamSynchronize(ctx->stream));          // h_cost is reused by every search
    const int ln = d->frame.width + d->frame.height + 1;
    const std::vector<float>& f = *d->lut_host;
    for (int n = 0; n < ln; ++n) {
        float dist = f[n];
        int32_t c;
        if (!(dist > params->minDistanceToObstacle * 1.000001)) c = ASTAR_INVALID_COST;
        else {
            c = 0;
            if (dist > params->minDistanceToObstacle && dist < params->maxDistanceWithCost) {
                double v = pow(params->maxDistanceWithCost - dist * 2000, params->distanceCostExponent);   // float product
                c = (v == v && fabs(v) < 2.0e9) ? static_cast<int>(v) : 0;
                if (c == ASTAR_INVALID_COST) c = ASTAR_INVALID_COST + 1;
            }
        }
        s->h_cost[n] = c;
    }
    BL_HIP(hipMemcpyAsync(s->cost_lut, s->h_cost, (size_t)ln * 4, hipMemcpyHostToDevice, ctx->stream));
    astar_args a;
    a.l1 = d->l1; a.W = d->frame.width; a.H = d->frame.height;
    a.cost_lut = s->cost_lut;
    a.heap = s->heap; a.heap_cap = s->heap_cap;
    a.closed = s->closed;
    a.path = s->path; a.path_cap = (long long)s->path_cap;
    a.result = s->d_result;
    bl_global_to_cell((double)goal->x, (double)goal->y, d->frame, &a.gx, &a.gy);     // astar.cpp:23-33
    bl_global_to_cell((double)start->x, (double)start->y, d->frame, &a.sx, &a.sy);
    a.max_pops = 1ll << 31;
    hipEvent_t e0, e1;
    rc = bl_timer_begin(ctx, BL_K_ASTAR, &e0, &e1);
    if (rc) return rc;
    BL_HIP(hipMemsetAsync(s->closed, 0xFF, (size_t)a.W * a.H * 4, ctx->stream));
    hipLaunchKernelGGL(k_astar, dim3(1), dim3(64), 0, ctx->stream, a);
    BL_HIP(hipGetLastError());
    rc = bl_timer_end(ctx, BL_K_ASTAR, e0, e1);
    if (rc) return rc;
    BL_HIP(hipMemcpyAsync(s->h_result, s->d_result, sizeof(astar_result), hipMemcpyDeviceToHost, ctx->stream));
    s->pending = true;
    s->start = *start;
    s->frame = d->frame;
    return BL_OK;
}

extern "C" int bl_astar_search_result(bl_ctx* ctx, bl_pose_xyt_t* out_path, int cap, int* out_len, int64_t* stats)
{
    BL_CHECK_ARG(ctx != nullptr && out_path != nullptr && cap >= 1 && out_len != nullptr);
    bl_astar_state* s = ctx->astar;
    if (!s || !s->pending) { bl_set_error("no A* search pending"); return BL_ERR_STATE; }
    BL_HIP(hipStreamSynchronize(ctx->stream));
    s->pending = false;
    astar_result r = *s->h_result;
    if (stats) { stats[0] = r.pops; stats[1] = r.pushes; }
    out_path[0] = s->start;                                            // path.path.push_back(start) (astar.cpp:21)
    *out_len = 1;
    if (r.status == ASTAR_ST_CAPACITY) { bl_set_error("A* open list exceeded its capacity (%lld pops)", r.pops); return BL_ERR_CAPACITY; }
    if (r.status == ASTAR_ST_LIMIT) { bl_set_error("A* pop limit reached"); return BL_ERR_CAPACITY; }
    if (r.status != ASTAR_ST_FOUND) return BL_OK;
    // makePath (astar.cpp:235-274): cells come goal-first; poses are emitted start-side first
    std::vector<int32_t> cells((size_t)r.path_len);
    BL_HIP(hipMemcpy(cells.data(), s->path, (size_t)r.path_len * 4, hipMemcpyDeviceToHost));
    std::vector<bl_pose_xyt_t> rev((size_t)r.path_len);
    float prevX = 0, prevY = 0;
    for (int i = 0; i < r.path_len; ++i) {
        int cx = cells[i] % s->frame.width, cy = cells[i] / s->frame.width;
        bl_pose_xyt_t p;
        p.utime = 0;                                                   // D6
        p.x = (float)((double)s->frame.ox + (double)cx * (double)s->frame.mpc);     // grid_utils.hpp:14-19
        p.y = (float)((double)s->frame.oy + (double)cy * (double)s->frame.mpc);
        if (i == 0) p.theta = (float)(double)s->start.theta;
        else p.theta = (float)atan2((double)prevY - (double)cy, (double)prevX - (double)cx);
        prevX = (float)cx; prevY = (float)cy;
        rev[i] = p;
    }
    int total = 1 + r.path_len;
    for (int i = 0; i < r.path_len && 1 + i < cap; ++i) out_path[1 + i] = rev[r.path_len - 1 - i];
    *out_len = total;
    return BL_OK;
}

extern "C" int bl_astar_search(bl_ctx* ctx, const bl_dist* d, const bl_pose_xyt_t* start, const bl_pose_xyt_t* goal,
                               const bl_search_params_t* params, bl_pose_xyt_t* out_path, int cap, int* out_len, int64_t* stats)
{
    int rc = bl_astar_search_async(ctx, d, start, goal, params);
    if (rc) return rc;
    return bl_astar_search_result(ctx, out_path, cap, out_len, stats);
}
