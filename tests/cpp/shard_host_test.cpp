// shard_host_test.cpp -- include/botlab/sharded_filter.hpp (a C++ host driving R particle shards from one process through the C
// ABI: composed finish, peer-store exchange) against ONE rank: the same scans and odometry (a binary script written by
// tests/test_gpu_shard_host_cpp.py), particles / estimates / map must be equal bit for bit.  Every rank on device 0.
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include <botlab/sharded_filter.hpp>

static void rd(FILE* f, void* p, size_t n) { if (fread(p, 1, n, f) != n) { std::fprintf(stderr, "short read\n"); std::exit(2); } }
static void ok(int rc) { if (rc != BL_OK) { std::fprintf(stderr, "%s\n", bl_last_error()); std::exit(3); } }

int main(int argc, char** argv)
{
    if (argc < 2) return 2;
    FILE* in = std::fopen(argv[1], "rb");
    if (!in) return 2;
    int32_t N, world, W, H, steps;
    float mpc, cpm, ox, oy;
    rd(in, &N, 4); rd(in, &world, 4); rd(in, &W, 4); rd(in, &H, 4); rd(in, &steps, 4);
    rd(in, &mpc, 4); rd(in, &cpm, 4); rd(in, &ox, 4); rd(in, &oy, 4);
    std::vector<int8_t> cells((size_t)W * H);
    rd(in, cells.data(), cells.size());
    bl_pose_xyt_t start; rd(in, &start, sizeof(start));

    // the single rank: updateFilter with the record-based finish riding in the map kernel (BOTLAB_MCL_NO_FUSED_FINISH in the environment)
    bl_ctx* ctx; bl_grid* grid; bl_mapping* mapping; bl_pf* pf;
    ok(bl_ctx_create(0, nullptr, &ctx));
    ok(bl_grid_create(ctx, W, H, mpc, cpm, ox, oy, &grid));
    ok(bl_grid_upload(grid, cells.data()));
    ok(bl_mapping_create(ctx, 5.0f, 4, 1, &mapping));
    ok(bl_pf_create(ctx, N, 0, N, &pf));
    ok(bl_pf_init_at_pose(pf, &start, 21));

    std::vector<int> devices((size_t)world, 0);
    botlab_hip::ShardedFilterGroup group(N, devices, W, H, mpc, cpm, ox, oy, cells.data());
    group.initializeFilterAtPose(start, 21);

    int bad = 0;
    for (int k = 0; k < steps; ++k) {
        bl_pose_xyt_t odo; rd(in, &odo, sizeof(odo));
        int32_t n, rv; int64_t utime;
        rd(in, &utime, 8); rd(in, &n, 4); rd(in, &rv, 4);
        std::vector<float> ranges((size_t)n), thetas((size_t)n);
        std::vector<int64_t> times((size_t)n);
        rd(in, ranges.data(), 4 * (size_t)n); rd(in, thetas.data(), 4 * (size_t)n); rd(in, times.data(), 8 * (size_t)n);
        bl_lidar_t scan; scan.utime = utime; scan.num_ranges = n; scan.ranges = ranges.data(); scan.thetas = thetas.data(); scan.times = times.data(); scan.intensities = nullptr;
        int moved = 0;
        ok(bl_pf_update_begin(pf, &odo, &scan, grid, rv, nullptr, &moved));
        ok(bl_mapping_update_finishing_pf(mapping, &scan, pf, utime, grid));
        bl_pose_xyt_t one; ok(bl_pf_pose_estimate(pf, &one));
        group.step(odo, scan, rv);
        for (int r = 0; r < group.world(); ++r) {
            const bl_pose_xyt_t p = group.poseEstimate(r);
            if (std::memcmp(&p, &one, sizeof(p)) != 0) { std::fprintf(stderr, "step %d rank %d: estimate differs (%.9g %.9g %.9g | %.9g %.9g %.9g)\n", k, r, p.x, p.y, p.theta, one.x, one.y, one.theta); ++bad; }
        }
    }
    std::vector<bl_particle_t> ref((size_t)N);
    ok(bl_pf_get_particles(pf, ref.data()));
    const std::vector<bl_particle_t> got = group.particles();
    if (std::memcmp(ref.data(), got.data(), sizeof(bl_particle_t) * (size_t)N) != 0) { std::fprintf(stderr, "particles differ\n"); ++bad; }
    std::vector<int8_t> m1((size_t)W * H);
    ok(bl_grid_download(grid, m1.data()));
    for (int r = 0; r < group.world(); ++r)
        if (group.mapCells(r, W, H) != m1) { std::fprintf(stderr, "rank %d: map differs\n", r); ++bad; }
    bl_pf_destroy(pf); bl_mapping_destroy(mapping); bl_grid_destroy(grid); bl_ctx_destroy(ctx);
    if (bad) return 1;
    std::printf("shard_host_test ok: %d ranks x %d steps, %d particles\n", world, steps, N);
    return 0;
}
