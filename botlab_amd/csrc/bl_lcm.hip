// bl_lcm.hip -- LCM wire codec and log-event framing for the seven message types on the hot path's boundary
// (lcmtypes/{pose_xyt_t,odometry_t,lidar_t,particle_t,particles_t,occupancy_grid_t,robot_path_t}.lcm), SURVEY.md section 8
// row f1.  LCM 1.4.0 (docker/Dockerfile:30-31) is neither vendored in the reference nor installed here, so this follows
// the published format:
//   * a message = 8-byte fingerprint + the members in declaration order, every scalar big-endian, arrays as their
//     elements in order (the length is an ordinary member), nested structs without their own fingerprint;
//   * fingerprint = rotate-left-1 of (base hash of the struct + the fingerprints-before-rotation of its struct-typed
//     members), base hash = lcm-gen's hash over member names, primitive type names and array dimensions;
//   * a log event = sync word 0xEDA1DA01, event number (i64), timestamp us (i64), channel length (i32), data length
//     (i32), channel bytes, data bytes, all big-endian.
// PARITY UNPINNED: no golden LCM bytes exist in the reference (its .log files are absent) and no LCM build is available
// to produce any; checked only against an independent restatement kept with the test infrastructure.
// particles_t and occupancy_grid_t can also be encoded straight from device state (the per-step particles() copy of
// slam.cpp:265-268 becomes one kernel writing wire bytes + one D2H into the caller's buffer).
#include <stdio.h>
#include <string.h>
#include <string>
#include <vector>

#include "bl_internal.h"

namespace {
// ---------------------------------------------------------------- type descriptions (lcmtypes/*.lcm)
struct lcm_member { const char* name; const char* type; const char* dim; int nested; };   // dim: variable array length member or null
const lcm_member M_POSE[] = {{"utime", "int64_t", nullptr, -1}, {"x", "float", nullptr, -1}, {"y", "float", nullptr, -1}, {"theta", "float", nullptr, -1}};
const lcm_member M_LIDAR[] = {{"utime", "int64_t", nullptr, -1}, {"num_ranges", "int32_t", nullptr, -1}, {"ranges", "float", "num_ranges", -1},
                              {"thetas", "float", "num_ranges", -1}, {"times", "int64_t", "num_ranges", -1}, {"intensities", "float", "num_ranges", -1}};
const lcm_member M_PARTICLE[] = {{"pose", "pose_xyt_t", nullptr, BL_LCM_POSE_XYT}, {"parent_pose", "pose_xyt_t", nullptr, BL_LCM_POSE_XYT},
                                 {"weight", "double", nullptr, -1}};
const lcm_member M_PARTICLES[] = {{"utime", "int64_t", nullptr, -1}, {"num_particles", "int32_t", nullptr, -1},
                                  {"particles", "particle_t", "num_particles", BL_LCM_PARTICLE}};
const lcm_member M_GRID[] = {{"utime", "int64_t", nullptr, -1}, {"origin_x", "float", nullptr, -1}, {"origin_y", "float", nullptr, -1},
                             {"meters_per_cell", "float", nullptr, -1}, {"width", "int32_t", nullptr, -1}, {"height", "int32_t", nullptr, -1},
                             {"num_cells", "int32_t", nullptr, -1}, {"cells", "int8_t", "num_cells", -1}};
const lcm_member M_PATH[] = {{"utime", "int64_t", nullptr, -1}, {"path_length", "int32_t", nullptr, -1},
                             {"path", "pose_xyt_t", "path_length", BL_LCM_POSE_XYT}};
struct lcm_type { const lcm_member* m; int n; };
const lcm_type TYPES[BL_LCM_TYPE_COUNT] = {{M_POSE, 4}, {M_POSE, 4}, {M_LIDAR, 6}, {M_PARTICLE, 3}, {M_PARTICLES, 3}, {M_GRID, 8}, {M_PATH, 3}};

// lcm-gen's struct hash (lcmgen.c lcm_struct_hash): signed 64-bit, arithmetic right shift
int64_t hash_update(int64_t v, char c) { v = (int64_t)(((uint64_t)v << 8) ^ (uint64_t)(v >> 55)) + c; return v; }
int64_t hash_string_update(int64_t v, const char* s)
{
    v = hash_update(v, (char)strlen(s));
    for (; *s != 0; s++) v = hash_update(v, *s);
    return v;
}
int64_t base_hash(int type)
{
    int64_t v = 0x12345678;
    for (int i = 0; i < TYPES[type].n; ++i) {
        const lcm_member& m = TYPES[type].m[i];
        v = hash_string_update(v, m.name);
        if (m.nested < 0) v = hash_string_update(v, m.type);          // primitive members carry their type name
        const int ndim = m.dim ? 1 : 0;
        v = hash_update(v, (char)ndim);
        if (m.dim) { v = hash_update(v, 1 /* LCM_VAR */); v = hash_string_update(v, m.dim); }
    }
    return v;
}
uint64_t compute_hash(int type)                                        // generated _computeHash(): base + nested, rotated left by one
{
    uint64_t h = (uint64_t)base_hash(type);
    for (int i = 0; i < TYPES[type].n; ++i)
        if (TYPES[type].m[i].nested >= 0) h += compute_hash(TYPES[type].m[i].nested);
    return (h << 1) + ((h >> 63) & 1);
}

// ---------------------------------------------------------------- big-endian writer / reader
struct wr {
    uint8_t* p; int64_t cap, n;
    void put(const void* src, int k)                                   // k bytes, reversed (host is little-endian)
    {
        if (p && n + k <= cap) { const uint8_t* s = (const uint8_t*)src; for (int i = 0; i < k; ++i) p[n + i] = s[k - 1 - i]; }
        n += k;
    }
    void i64(int64_t v) { put(&v, 8); }
    void u64(uint64_t v) { put(&v, 8); }
    void i32(int32_t v) { put(&v, 4); }
    void f32(float v) { put(&v, 4); }
    void f64(double v) { put(&v, 8); }
    void bytes(const void* src, int64_t k) { if (p && n + k <= cap && k > 0) memcpy(p + n, src, (size_t)k); n += k; }
    void pose(const bl_pose_xyt_t& q) { i64(q.utime); f32(q.x); f32(q.y); f32(q.theta); }
};
struct rd {
    const uint8_t* p; int64_t len, n; bool ok;
    void get(void* dst, int k)
    {
        if (n + k > len) { ok = false; memset(dst, 0, (size_t)k); return; }
        uint8_t* d = (uint8_t*)dst; for (int i = 0; i < k; ++i) d[i] = p[n + k - 1 - i];
        n += k;
    }
    int64_t i64() { int64_t v; get(&v, 8); return v; }
    uint64_t u64() { uint64_t v; get(&v, 8); return v; }
    int32_t i32() { int32_t v; get(&v, 4); return v; }
    float f32() { float v; get(&v, 4); return v; }
    double f64() { double v; get(&v, 8); return v; }
    bl_pose_xyt_t pose() { bl_pose_xyt_t q; q.utime = i64(); q.x = f32(); q.y = f32(); q.theta = f32(); return q; }
};
int64_t finish(const wr& w, int64_t cap)
{
    if (w.p && w.n > cap) { bl_set_error("LCM encode: %lld bytes do not fit the %lld-byte buffer", (long long)w.n, (long long)cap); return -(int64_t)BL_ERR_CAPACITY; }
    return w.n;
}
bool check_hash(rd& r, int type)
{
    const uint64_t h = r.u64();
    if (!r.ok || h != compute_hash(type)) { bl_set_error("LCM decode: fingerprint mismatch or short message"); return false; }
    return true;
}
}  // namespace

extern "C" uint64_t bl_lcm_fingerprint(int type) { return (type >= 0 && type < BL_LCM_TYPE_COUNT) ? compute_hash(type) : 0; }

// Encoders return the encoded size (also when buf is null: size query) or a negative status.
extern "C" int64_t bl_lcm_encode_pose(int type, const bl_pose_xyt_t* pose, uint8_t* buf, int64_t cap)
{
    if (!pose || (type != BL_LCM_POSE_XYT && type != BL_LCM_ODOMETRY)) { bl_set_error("bad argument"); return -(int64_t)BL_ERR_ARG; }
    wr w = {buf, cap, 0};
    w.u64(compute_hash(type)); w.pose(*pose);
    return finish(w, cap);
}
extern "C" int64_t bl_lcm_encode_lidar(const bl_lidar_t* scan, const float* intensities, uint8_t* buf, int64_t cap)
{
    if (!scan || scan->num_ranges < 0) { bl_set_error("bad argument"); return -(int64_t)BL_ERR_ARG; }
    wr w = {buf, cap, 0};
    const int n = scan->num_ranges;
    w.u64(compute_hash(BL_LCM_LIDAR)); w.i64(scan->utime); w.i32(n);
    for (int i = 0; i < n; ++i) w.f32(scan->ranges[i]);
    for (int i = 0; i < n; ++i) w.f32(scan->thetas[i]);
    for (int i = 0; i < n; ++i) w.i64(scan->times[i]);
    for (int i = 0; i < n; ++i) w.f32(intensities ? intensities[i] : 0.0f);
    return finish(w, cap);
}
extern "C" int64_t bl_lcm_encode_particles(int64_t utime, const bl_particle_t* particles, int32_t n, uint8_t* buf, int64_t cap)
{
    if (n < 0 || (n > 0 && !particles)) { bl_set_error("bad argument"); return -(int64_t)BL_ERR_ARG; }
    wr w = {buf, cap, 0};
    w.u64(compute_hash(BL_LCM_PARTICLES)); w.i64(utime); w.i32(n);
    for (int i = 0; i < n; ++i) { w.pose(particles[i].pose); w.pose(particles[i].parent_pose); w.f64(particles[i].weight); }
    return finish(w, cap);
}
extern "C" int64_t bl_lcm_encode_grid(int64_t utime, float origin_x, float origin_y, float meters_per_cell, int32_t width,
                                      int32_t height, const int8_t* cells, uint8_t* buf, int64_t cap)
{
    if (width < 0 || height < 0 || ((int64_t)width * height > 0 && !cells) || (int64_t)width * height > 0x7fffffffll) { bl_set_error("bad argument"); return -(int64_t)BL_ERR_ARG; }
    wr w = {buf, cap, 0};
    const int32_t nc = width * height;
    w.u64(compute_hash(BL_LCM_OCCUPANCY_GRID)); w.i64(utime); w.f32(origin_x); w.f32(origin_y); w.f32(meters_per_cell);
    w.i32(width); w.i32(height); w.i32(nc);
    w.bytes(cells, nc);
    return finish(w, cap);
}
extern "C" int64_t bl_lcm_encode_path(int64_t utime, const bl_pose_xyt_t* path, int32_t n, uint8_t* buf, int64_t cap)
{
    if (n < 0 || (n > 0 && !path)) { bl_set_error("bad argument"); return -(int64_t)BL_ERR_ARG; }
    wr w = {buf, cap, 0};
    w.u64(compute_hash(BL_LCM_ROBOT_PATH)); w.i64(utime); w.i32(n);
    for (int i = 0; i < n; ++i) w.pose(path[i]);
    return finish(w, cap);
}

// Decoders: arrays are caller-owned with capacities in elements; *n receives the message's count (decode stops filling
// at the capacity but still validates the length).  Return BL_OK or a status.
extern "C" int bl_lcm_decode_pose(int type, const uint8_t* buf, int64_t len, bl_pose_xyt_t* out)
{
    BL_CHECK_ARG(buf != nullptr && out != nullptr && (type == BL_LCM_POSE_XYT || type == BL_LCM_ODOMETRY));
    rd r = {buf, len, 0, true};
    if (!check_hash(r, type)) return BL_ERR_ARG;
    *out = r.pose();
    if (!r.ok) { bl_set_error("LCM decode: short message"); return BL_ERR_ARG; }
    return BL_OK;
}
extern "C" int bl_lcm_decode_lidar(const uint8_t* buf, int64_t len, int64_t* utime, int32_t* n, float* ranges, float* thetas,
                                   int64_t* times, float* intensities, int32_t cap)
{
    BL_CHECK_ARG(buf != nullptr && utime != nullptr && n != nullptr);
    rd r = {buf, len, 0, true};
    if (!check_hash(r, BL_LCM_LIDAR)) return BL_ERR_ARG;
    *utime = r.i64(); *n = r.i32();
    if (!r.ok || *n < 0 || r.n + (int64_t)*n * 20 > len) { bl_set_error("LCM decode: short or corrupt lidar_t"); return BL_ERR_ARG; }
    for (int i = 0; i < *n; ++i) { float v = r.f32(); if (ranges && i < cap) ranges[i] = v; }
    for (int i = 0; i < *n; ++i) { float v = r.f32(); if (thetas && i < cap) thetas[i] = v; }
    for (int i = 0; i < *n; ++i) { int64_t v = r.i64(); if (times && i < cap) times[i] = v; }
    for (int i = 0; i < *n; ++i) { float v = r.f32(); if (intensities && i < cap) intensities[i] = v; }
    return BL_OK;
}
extern "C" int bl_lcm_decode_particles(const uint8_t* buf, int64_t len, int64_t* utime, int32_t* n, bl_particle_t* out, int32_t cap)
{
    BL_CHECK_ARG(buf != nullptr && utime != nullptr && n != nullptr);
    rd r = {buf, len, 0, true};
    if (!check_hash(r, BL_LCM_PARTICLES)) return BL_ERR_ARG;
    *utime = r.i64(); *n = r.i32();
    if (!r.ok || *n < 0 || r.n + (int64_t)*n * 48 > len) { bl_set_error("LCM decode: short or corrupt particles_t"); return BL_ERR_ARG; }
    for (int i = 0; i < *n; ++i) {
        bl_particle_t q; memset(&q, 0, sizeof(q));                    // member-wise: the padding of the output stays zero
        const bl_pose_xyt_t a = r.pose(), b = r.pose();
        q.pose.utime = a.utime; q.pose.x = a.x; q.pose.y = a.y; q.pose.theta = a.theta;
        q.parent_pose.utime = b.utime; q.parent_pose.x = b.x; q.parent_pose.y = b.y; q.parent_pose.theta = b.theta;
        q.weight = r.f64();
        if (out && i < cap) out[i] = q;
    }
    return BL_OK;
}
extern "C" int bl_lcm_decode_grid(const uint8_t* buf, int64_t len, int64_t* utime, float* origin_xy_mpc, int32_t* width_height_ncells,
                                  int8_t* cells, int64_t cap)
{
    BL_CHECK_ARG(buf != nullptr && utime != nullptr && origin_xy_mpc != nullptr && width_height_ncells != nullptr);
    rd r = {buf, len, 0, true};
    if (!check_hash(r, BL_LCM_OCCUPANCY_GRID)) return BL_ERR_ARG;
    *utime = r.i64();
    for (int i = 0; i < 3; ++i) origin_xy_mpc[i] = r.f32();
    for (int i = 0; i < 3; ++i) width_height_ncells[i] = r.i32();
    const int32_t nc = width_height_ncells[2];
    if (!r.ok || nc < 0 || r.n + nc > len) { bl_set_error("LCM decode: short or corrupt occupancy_grid_t"); return BL_ERR_ARG; }
    if (cells) memcpy(cells, buf + r.n, (size_t)(nc < cap ? nc : cap));
    return BL_OK;
}
extern "C" int bl_lcm_decode_path(const uint8_t* buf, int64_t len, int64_t* utime, int32_t* n, bl_pose_xyt_t* path, int32_t cap)
{
    BL_CHECK_ARG(buf != nullptr && utime != nullptr && n != nullptr);
    rd r = {buf, len, 0, true};
    if (!check_hash(r, BL_LCM_ROBOT_PATH)) return BL_ERR_ARG;
    *utime = r.i64(); *n = r.i32();
    if (!r.ok || *n < 0 || r.n + (int64_t)*n * 20 > len) { bl_set_error("LCM decode: short or corrupt robot_path_t"); return BL_ERR_ARG; }
    for (int i = 0; i < *n; ++i) {
        const bl_pose_xyt_t q = r.pose();
        if (path && i < cap) { memset(&path[i], 0, sizeof(bl_pose_xyt_t)); path[i].utime = q.utime; path[i].x = q.x; path[i].y = q.y; path[i].theta = q.theta; }
    }
    return BL_OK;
}

// ---------------------------------------------------------------- log events (lcm-logger / lcm-logplayer files)
extern "C" int64_t bl_lcm_log_event_size(int32_t channel_len, int32_t data_len) { return 4 + 8 + 8 + 4 + 4 + (int64_t)channel_len + data_len; }
extern "C" int64_t bl_lcm_log_write_event(int64_t event_number, int64_t timestamp_us, const char* channel, const uint8_t* data,
                                          int32_t data_len, uint8_t* buf, int64_t cap)
{
    if (!channel || data_len < 0 || (data_len > 0 && !data)) { bl_set_error("bad argument"); return -(int64_t)BL_ERR_ARG; }
    const int32_t cl = (int32_t)strlen(channel);
    wr w = {buf, cap, 0};
    const uint32_t sync = 0xEDA1DA01u;
    w.put(&sync, 4); w.i64(event_number); w.i64(timestamp_us); w.i32(cl); w.i32(data_len);
    w.bytes(channel, cl); w.bytes(data, data_len);
    return finish(w, cap);
}
// Parses one event at buf[0..len): returns its total size (0: not enough bytes yet, < 0: not an event here).
extern "C" int64_t bl_lcm_log_read_event(const uint8_t* buf, int64_t len, int64_t* event_number, int64_t* timestamp_us,
                                         int64_t* channel_off, int32_t* channel_len, int64_t* data_off, int32_t* data_len)
{
    if (!buf) { bl_set_error("bad argument"); return -(int64_t)BL_ERR_ARG; }
    if (len < 28) return 0;
    rd r = {buf, len, 0, true};
    uint32_t sync; r.get(&sync, 4);
    if (sync != 0xEDA1DA01u) { bl_set_error("LCM log: no sync word at this offset"); return -(int64_t)BL_ERR_ARG; }
    const int64_t en = r.i64(), ts = r.i64();
    const int32_t cl = r.i32(), dl = r.i32();
    if (cl < 0 || dl < 0) { bl_set_error("LCM log: corrupt event header"); return -(int64_t)BL_ERR_ARG; }
    if (28 + (int64_t)cl + dl > len) return 0;
    if (event_number) *event_number = en;
    if (timestamp_us) *timestamp_us = ts;
    if (channel_off) *channel_off = 28;
    if (channel_len) *channel_len = cl;
    if (data_off) *data_off = 28 + cl;
    if (data_len) *data_len = dl;
    return 28 + (int64_t)cl + dl;
}

// ---------------------------------------------------------------- encode straight from device state
// occupancy_grid_t: header on the host, the cells copied from HBM straight behind it (OccupancyGrid::toLCM + encode of
// slam.cpp:285-289 without the intermediate std::vector copies).  particles_t: bl_pf_encode_particles_lcm (bl_mcl.hip).
extern "C" int64_t bl_grid_encode_lcm(bl_grid* grid, int64_t utime, uint8_t* buf, int64_t cap)
{
    if (!grid || !buf) { bl_set_error("bad argument"); return -(int64_t)BL_ERR_ARG; }
    const int32_t w0 = grid->frame.width, h0 = grid->frame.height;
    const int64_t nc = (int64_t)w0 * h0;
    wr w = {buf, cap, 0};
    w.u64(compute_hash(BL_LCM_OCCUPANCY_GRID)); w.i64(utime); w.f32(grid->frame.ox); w.f32(grid->frame.oy); w.f32(grid->frame.mpc);
    w.i32(w0); w.i32(h0); w.i32((int32_t)nc);
    const int64_t head = w.n;
    if (head + nc > cap) { bl_set_error("LCM encode: %lld bytes do not fit the %lld-byte buffer", (long long)(head + nc), (long long)cap); return -(int64_t)BL_ERR_CAPACITY; }
    if (hipSetDevice(grid->ctx->device) != hipSuccess || hipMemcpyAsync(buf + head, grid->cells, (size_t)nc, hipMemcpyDeviceToHost, grid->ctx->stream) != hipSuccess ||
        hipStreamSynchronize(grid->ctx->stream) != hipSuccess) { bl_set_error("bl_grid_encode_lcm: HIP copy failed"); return -(int64_t)BL_ERR_HIP; }
    return head + nc;
}
