// Plain message records with the fields of botLab's lcm types (lcmtypes/*.lcm), used ONLY by this repository's own
// compile check and C++ GPU test.  In the botLab tree the real lcm-gen headers are used instead.
#ifndef BOTLAB_DROPIN_TEST_TYPES_HPP
#define BOTLAB_DROPIN_TEST_TYPES_HPP
#include <cstdint>
#include <vector>
#include <botlab/botlab_dropin.hpp>

struct pose_xyt_t { int64_t utime = 0; float x = 0, y = 0, theta = 0; };
struct lidar_t { int64_t utime = 0; int32_t num_ranges = 0; std::vector<float> ranges, thetas; std::vector<int64_t> times; std::vector<float> intensities; };
struct particle_t { pose_xyt_t pose, parent_pose; double weight = 0; };
struct particles_t { int64_t utime = 0; int32_t num_particles = 0; std::vector<particle_t> particles; };
struct occupancy_grid_t { int64_t utime = 0; float origin_x = 0, origin_y = 0, meters_per_cell = 0; int32_t width = 0, height = 0, num_cells = 0; std::vector<int8_t> cells; };
struct robot_path_t { int64_t utime = 0; int32_t path_length = 0; std::vector<pose_xyt_t> path; };
#endif
