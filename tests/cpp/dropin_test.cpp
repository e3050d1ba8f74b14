// dropin_test.cpp -- run-time check of include/botlab/botlab_dropin.hpp on a GPU.  Reads a .map file and a binary scan
// script written by tests/test_gpu_cpp_dropin.py, drives OccupancyGrid / Mapping / ParticleFilter /
// ObstacleDistanceGrid / search_for_path exactly as OccupancyGridSLAM::runSLAMIteration + MotionPlanner would, and
// writes the results for the Python side to compare with the oracle.
#include <cstdio>
#include <cstdlib>
#include "dropin_test_types.hpp"

typedef botlab_hip::MappingT<pose_xyt_t, lidar_t> Mapping;
typedef botlab_hip::ParticleFilterT<pose_xyt_t, lidar_t, particle_t, particles_t> ParticleFilter;

static void rd(FILE* f, void* p, size_t n) { if (fread(p, 1, n, f) != n) { std::fprintf(stderr, "short read\n"); std::exit(2); } }

int main(int argc, char** argv)
{
    if (argc < 4) { std::fprintf(stderr, "usage: dropin_test map.map script.bin out.bin\n"); return 2; }
    botlab_hip::OccupancyGrid map;
    if (!map.loadFromFile(argv[1])) return 2;
    FILE* in = std::fopen(argv[2], "rb");
    FILE* out = std::fopen(argv[3], "wb");
    if (!in || !out) return 2;
    int32_t nsteps, nparticles, nrays;
    rd(in, &nsteps, 4); rd(in, &nparticles, 4); rd(in, &nrays, 4);
    Mapping mapper(5.0f, 4, 1);
    ParticleFilter filter(nparticles);
    pose_xyt_t init;
    rd(in, &init.utime, 8); rd(in, &init.x, 4); rd(in, &init.y, 4); rd(in, &init.theta, 4);
    filter.initializeFilterAtPose(init);
    pose_xyt_t pose = init;
    for (int k = 0; k < nsteps; ++k) {
        lidar_t scan; pose_xyt_t odo;
        scan.num_ranges = nrays; scan.ranges.resize(nrays); scan.thetas.resize(nrays); scan.times.resize(nrays);
        rd(in, &scan.utime, 8); rd(in, scan.ranges.data(), 4 * nrays); rd(in, scan.thetas.data(), 4 * nrays); rd(in, scan.times.data(), 8 * nrays);
        rd(in, &odo.utime, 8); rd(in, &odo.x, 4); rd(in, &odo.y, 4); rd(in, &odo.theta, 4);
        pose = filter.updateFilter(odo, scan, map);          // slam.cpp:262
        mapper.updateMap(scan, pose, map);                   // slam.cpp:279
        std::fwrite(&pose.x, 4, 1, out); std::fwrite(&pose.y, 4, 1, out); std::fwrite(&pose.theta, 4, 1, out);
    }
    particles_t ps = filter.particles();
    double wsum = 0;
    for (auto& p : ps.particles) wsum += p.weight;
    std::fwrite(&wsum, 8, 1, out);
    // host-visible reads through the lazy mirror, a host write, and a copy
    botlab_hip::OccupancyGrid copy = map;
    copy.setLogOdds(3, 4, 77);
    int8_t a = copy.logOdds(3, 4), b = map.logOdds(3, 4);
    std::fwrite(&a, 1, 1, out); std::fwrite(&b, 1, 1, out);
    for (int y = 0; y < map.heightInCells(); ++y) for (int x = 0; x < map.widthInCells(); ++x) { int8_t v = map(x, y); std::fwrite(&v, 1, 1, out); }
    botlab_hip::ObstacleDistanceGrid dist;
    dist.setDistances(map);
    for (int y = 0; y < dist.heightInCells(); ++y) for (int x = 0; x < dist.widthInCells(); ++x) { float v = dist(x, y); std::fwrite(&v, 4, 1, out); }
    botlab_hip::SearchParams sp = {0.2, 2.0, 1.0};
    pose_xyt_t goal; goal.x = -0.35f; goal.y = 0.2f;
    robot_path_t path = botlab_hip::search_for_path_t<robot_path_t, pose_xyt_t>(pose, goal, dist, sp);
    std::fwrite(&path.path_length, 4, 1, out);
    for (auto& p : path.path) { std::fwrite(&p.x, 4, 1, out); std::fwrite(&p.y, 4, 1, out); std::fwrite(&p.theta, 4, 1, out); }
    std::fclose(out);
    std::printf("dropin_test ok: %d steps, path_length %d, weight sum %.12f\n", nsteps, path.path_length, wsum);
    return 0;
}
