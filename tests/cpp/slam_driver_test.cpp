// slam_driver_test.cpp -- drives include/botlab/slam_driver.hpp (PoseTrace + OccupancyGridSLAM without LCM) from a
// binary event script written by tests/test_gpu_slam_driver.py and dumps what the oracle's restatement is compared with.
// Events: 'O' odometry, 'P' SLAM_POSE ground truth, 'L' lidar, 'T' optitrack; after every event the driver runs while ready.
#include <cstdio>
#include <cstdlib>
#include <memory>
#include "dropin_test_types.hpp"
#include <botlab/slam_driver.hpp>

struct odometry_t { int64_t utime = 0; float x = 0, y = 0, theta = 0; };
typedef botlab_hip::OccupancyGridSLAMT<pose_xyt_t, lidar_t, odometry_t, particle_t, particles_t, occupancy_grid_t> SLAM;

static void rd(FILE* f, void* p, size_t n) { if (fread(p, 1, n, f) != n) { std::fprintf(stderr, "short read\n"); std::exit(2); } }

int main(int argc, char** argv)
{
    if (argc < 3) return 2;
    FILE* in = std::fopen(argv[1], "rb");
    FILE* out = std::fopen(argv[2], "wb");
    if (!in || !out) return 2;
    // mode 0 = mapping-only, 1 = localization-only, 2 = action-only (both on the .map file named by argv[3]: slam.cpp:36-45),
    // 3 = full SLAM, -1 = PoseTrace only (no GPU touched)
    int32_t mode, nparticles, nevents, wait_opti;
    rd(in, &mode, 4); rd(in, &nparticles, 4); rd(in, &nevents, 4); rd(in, &wait_opti, 4);
    int published_pose = 0, published_map = 0, published_particles = 0;
    SLAM::Publisher pub;
    pub.slamPose = [&](const pose_xyt_t&) { ++published_pose; };
    pub.slamParticles = [&](const particles_t& p) { published_particles += p.num_particles > 0; };
    pub.slamMap = [&](const occupancy_grid_t&) { ++published_map; };
    std::unique_ptr<SLAM> slamp;
    const std::string loc_map = (mode == 1 || mode == 2) && argc > 3 ? argv[3] : "";
    if ((mode == 1 || mode == 2) && loc_map.empty()) { std::fprintf(stderr, "mode %d needs a map file\n", mode); return 2; }
    if (mode >= 0) slamp.reset(new SLAM(nparticles, 4, 1, pub, wait_opti != 0, mode == 0, mode == 2, loc_map));
    // PoseTrace checks ride along: a private trace fed with the same 'P' events, queried by 'Q' events
    botlab_hip::PoseTraceT<pose_xyt_t> trace;
    for (int e = 0; e < nevents; ++e) {
        char kind; rd(in, &kind, 1);
        if (kind == 'O') { odometry_t o; rd(in, &o.utime, 8); rd(in, &o.x, 4); rd(in, &o.y, 4); rd(in, &o.theta, 4); if (slamp) slamp->handleOdometry(o); }
        else if (kind == 'P' || kind == 'T') {
            pose_xyt_t p; rd(in, &p.utime, 8); rd(in, &p.x, 4); rd(in, &p.y, 4); rd(in, &p.theta, 4);
            if (kind == 'P') { if (slamp) slamp->handlePose(p); trace.addPose(p); } else if (slamp) slamp->handleOptitrack(p);
        } else if (kind == 'L') {
            lidar_t s; int32_t n; rd(in, &s.utime, 8); rd(in, &n, 4);
            s.num_ranges = n; s.ranges.resize(n); s.thetas.resize(n); s.times.resize(n);
            rd(in, s.ranges.data(), 4 * n); rd(in, s.thetas.data(), 4 * n); rd(in, s.times.data(), 8 * n);
            if (slamp) slamp->handleLaser(s);
        } else if (kind == 'Q') {                       // PoseTrace query: poseAt(t), containsPoseAtTime(t)
            int64_t t; rd(in, &t, 8);
            pose_xyt_t p = trace.poseAt(t);
            int32_t c = trace.containsPoseAtTime(t) ? 1 : 0;
            std::fwrite(&p.utime, 8, 1, out); std::fwrite(&p.x, 4, 1, out); std::fwrite(&p.y, 4, 1, out); std::fwrite(&p.theta, 4, 1, out);
            std::fwrite(&c, 4, 1, out);
            continue;
        } else if (kind == 'R') {                       // setReferencePose
            pose_xyt_t p; rd(in, &p.utime, 8); rd(in, &p.x, 4); rd(in, &p.y, 4); rd(in, &p.theta, 4);
            trace.setReferencePose(p);
            continue;
        } else if (kind == 'X') {                       // eraseTraceUntil(t): writes the erase count and the new size
            int64_t t; rd(in, &t, 8);
            int32_t r[2]; r[0] = trace.eraseTraceUntil(t); r[1] = (int32_t)trace.size();
            std::fwrite(r, 4, 2, out);
            continue;
        }
        if (!slamp) continue;
        SLAM& slam = *slamp;
        while (slam.isReadyToUpdate()) {
            slam.runSLAMIteration();
            pose_xyt_t c = slam.currentPose();
            int32_t st[3] = {slam.numIgnoredScans(), (int32_t)slam.queuedScans(), slam.mapUpdateCount()};
            std::fwrite("I", 1, 1, out);
            std::fwrite(&c.utime, 8, 1, out); std::fwrite(&c.x, 4, 1, out); std::fwrite(&c.y, 4, 1, out); std::fwrite(&c.theta, 4, 1, out);
            std::fwrite(st, 4, 3, out);
        }
    }
    std::fwrite("E", 1, 1, out);
    if (!slamp) { std::fclose(out); std::printf("slam_driver_test ok: trace only\n"); return 0; }
    SLAM& slam = *slamp;
    int32_t fin[5] = {slam.numIgnoredScans(), (int32_t)slam.queuedScans(), slam.mapUpdateCount(), published_pose, published_map};
    std::fwrite(fin, 4, 5, out);
    const botlab_hip::OccupancyGrid& m = slam.map();
    for (int y = 0; y < m.heightInCells(); ++y) for (int x = 0; x < m.widthInCells(); ++x) { int8_t v = m(x, y); std::fwrite(&v, 1, 1, out); }
    std::fclose(out);
    std::printf("slam_driver_test ok: %d map updates, %d poses published (%d particle sets), %d maps published\n", fin[2], published_pose,
                published_particles, published_map);
    return 0;
}
