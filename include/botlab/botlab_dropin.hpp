// botlab_dropin.hpp -- C++ drop-in classes for botLab's hot path, forwarding to libbotlab_hip.so (include/botlab_hip.h).
//
// Each class keeps the reference's name, method signatures, argument meaning and error behaviour:
//   OccupancyGrid            src/slam/occupancy_grid.hpp:51-209
//   Mapping                  src/slam/mapping.hpp:25-34
//   ParticleFilter           src/slam/particle_filter.hpp:38-77
//   ObstacleDistanceGrid     src/planning/obstacle_distance_grid.hpp:28-96
//   search_for_path          src/planning/astar.hpp:58-61
// The reference's message structs (pose_xyt_t, lidar_t, particle_t, particles_t, occupancy_grid_t, robot_path_t) are
// lcm-gen output that lives in the botLab tree, not here, so the classes are templates over those types; in the botLab
// tree one header gives them their reference names:
//
//     #include <lcmtypes/pose_xyt_t.hpp> ... <lcmtypes/robot_path_t.hpp>
//     #include <botlab/botlab_dropin.hpp>
//     typedef botlab_hip::MappingT<pose_xyt_t, lidar_t> Mapping;
//     typedef botlab_hip::ParticleFilterT<pose_xyt_t, lidar_t, particle_t, particles_t> ParticleFilter; ...
//
// Ownership follows the reference (value semantics): copying an OccupancyGrid / ObstacleDistanceGrid deep-copies the
// device buffer; destructors free it.  No exceptions cross the C ABI; a failing call prints bl_last_error() to stderr
// and aborts, matching the reference's assert-style preconditions.
#ifndef BOTLAB_DROPIN_HPP
#define BOTLAB_DROPIN_HPP

#include <cassert>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <string>
#include <vector>

#include "../botlab_hip.h"

namespace botlab_hip {

inline void check(int rc, const char* what)
{
    if (rc != BL_OK) {
        std::fprintf(stderr, "botlab_hip: %s failed (%d): %s\n", what, rc, bl_last_error());
        std::abort();
    }
}

// One context per thread that drives the hot path (the reference touches these objects from the runSLAM thread only).
inline bl_ctx* default_ctx()
{
    static thread_local bl_ctx* ctx = nullptr;
    if (!ctx) check(bl_ctx_create(0, nullptr, &ctx), "bl_ctx_create");
    return ctx;
}

template <typename T>
struct PointT { T x, y; PointT() : x(0), y(0) {} PointT(T a, T b) : x(a), y(b) {} };

typedef int8_t CellOdds;

// ------------------------------------------------------------------------------------------------ OccupancyGrid
class OccupancyGrid {
public:
    OccupancyGrid() : h_(nullptr), width_(0), height_(0), metersPerCell_(0.05f), cellsPerMeter_(1.0 / 0.05f),
                      origin_(0, 0), hostValid_(true), deviceValid_(true) {}
    OccupancyGrid(float widthInMeters, float heightInMeters, float metersPerCell)
        : h_(nullptr), metersPerCell_(metersPerCell), origin_(-widthInMeters / 2.0f, -heightInMeters / 2.0f),
          hostValid_(true), deviceValid_(true)
    {
        assert(widthInMeters > 0.0f);
        assert(heightInMeters > 0.0f);
        assert(metersPerCell_ <= widthInMeters);
        assert(metersPerCell_ <= heightInMeters);
        cellsPerMeter_ = 1.0f / metersPerCell_;
        width_ = widthInMeters * cellsPerMeter_;
        height_ = heightInMeters * cellsPerMeter_;
        host_.assign(static_cast<size_t>(width_) * height_, 0);
        allocate();
    }
    OccupancyGrid(const OccupancyGrid& o) : h_(nullptr) { copyFrom(o); }
    OccupancyGrid& operator=(const OccupancyGrid& o) { if (this != &o) { release(); copyFrom(o); } return *this; }
    ~OccupancyGrid() { release(); }

    int widthInCells() const { return width_; }
    float widthInMeters() const { return width_ * metersPerCell_; }
    int heightInCells() const { return height_; }
    float heightInMeters() const { return height_ * metersPerCell_; }
    float metersPerCell() const { return metersPerCell_; }
    float cellsPerMeter() const { return cellsPerMeter_; }
    PointT<float> originInGlobalFrame() const { return origin_; }

    void setOrigin(float x, float y)            // occupancy_grid.cpp:38-46 (resets, then SUBTRACTS)
    {
        reset();
        origin_.x -= x;
        origin_.y -= y;
        if (h_) check(bl_grid_set_frame(h_, metersPerCell_, cellsPerMeter_, origin_.x, origin_.y), "bl_grid_set_frame");
    }
    void reset()
    {
        std::fill(host_.begin(), host_.end(), 0);
        hostValid_ = true;
        if (h_) { check(bl_grid_reset(h_), "bl_grid_reset"); deviceValid_ = true; }
    }
    bool isCellInGrid(int x, int y) const { return (x >= 0) && (x < width_) && (y >= 0) && (y < height_); }
    CellOdds logOdds(int x, int y) const { return isCellInGrid(x, y) ? (*this)(x, y) : 0; }
    void setLogOdds(int x, int y, CellOdds v) { if (isCellInGrid(x, y)) (*this)(x, y) = v; }
    CellOdds& operator()(int x, int y) { syncToHost(); deviceValid_ = false; return host_[cellIndex(x, y)]; }
    CellOdds operator()(int x, int y) const { syncToHost(); return host_[cellIndex(x, y)]; }

    template <class GridMsg> GridMsg toLCM() const                      // occupancy_grid.cpp:83-96
    {
        syncToHost();
        GridMsg g;
        g.origin_x = origin_.x; g.origin_y = origin_.y; g.meters_per_cell = metersPerCell_;
        g.width = width_; g.height = height_; g.num_cells = static_cast<int32_t>(host_.size());
        g.cells.assign(host_.begin(), host_.end());
        return g;
    }
    template <class GridMsg> void fromLCM(const GridMsg& g)             // occupancy_grid.cpp:99-108
    {
        release();
        origin_.x = g.origin_x; origin_.y = g.origin_y;
        metersPerCell_ = g.meters_per_cell; cellsPerMeter_ = 1.0f / g.meters_per_cell;
        height_ = g.height; width_ = g.width;
        host_.assign(g.cells.begin(), g.cells.end());
        hostValid_ = true; deviceValid_ = false;
        allocate();
    }
    bool saveToFile(const std::string& filename) const                  // occupancy_grid.cpp:111-136
    {
        std::ofstream out(filename);
        if (!out.is_open()) { std::fprintf(stderr, "ERROR: OccupancyGrid::saveToFile: Failed to save to %s\n", filename.c_str()); return false; }
        out << origin_.x << ' ' << origin_.y << ' ' << width_ << ' ' << height_ << ' ' << metersPerCell_ << '\n';
        for (int y = 0; y < height_; ++y) { for (int x = 0; x < width_; ++x) out << +logOdds(x, y) << ' '; out << '\n'; }
        return out.good();
    }
    bool loadFromFile(const std::string& filename)                      // occupancy_grid.cpp:138-175 (cellsPerMeter_ untouched)
    {
        std::ifstream in(filename);
        if (!in.is_open()) { std::fprintf(stderr, "ERROR: OccupancyGrid::loadFromFile: Failed to load from %s\n", filename.c_str()); return false; }
        release();
        width_ = -1; height_ = -1;
        in >> origin_.x >> origin_.y >> width_ >> height_ >> metersPerCell_;
        assert(width_ > 0); assert(height_ > 0); assert(metersPerCell_ > 0.0f);
        host_.assign(static_cast<size_t>(width_) * height_, 0);
        int odds = 0;
        for (int y = 0; y < height_; ++y) for (int x = 0; x < width_; ++x) { in >> odds; host_[cellIndex(x, y)] = static_cast<CellOdds>(odds); }
        hostValid_ = true; deviceValid_ = false;
        allocate();
        return true;
    }

    // ---- device side (used by the classes below)
    bl_grid* device() const { syncToDevice(); return h_; }
    void markDeviceWritten() { hostValid_ = false; deviceValid_ = true; }

private:
    mutable bl_grid* h_;
    int width_, height_;
    float metersPerCell_, cellsPerMeter_;
    PointT<float> origin_;
    mutable std::vector<CellOdds> host_;      // host mirror, synchronised lazily
    mutable bool hostValid_, deviceValid_;

    int cellIndex(int x, int y) const { return y * width_ + x; }
    void allocate()
    {
        if (width_ <= 0 || height_ <= 0) return;
        check(bl_grid_create(default_ctx(), width_, height_, metersPerCell_, cellsPerMeter_, origin_.x, origin_.y, &h_), "bl_grid_create");
        deviceValid_ = false;
        syncToDevice();
    }
    void release() { if (h_) { bl_grid_destroy(h_); h_ = nullptr; } }
    void copyFrom(const OccupancyGrid& o)
    {
        o.syncToHost();
        width_ = o.width_; height_ = o.height_; metersPerCell_ = o.metersPerCell_; cellsPerMeter_ = o.cellsPerMeter_;
        origin_ = o.origin_; host_ = o.host_; hostValid_ = true; deviceValid_ = false;
        allocate();
    }
    void syncToHost() const
    {
        if (!hostValid_ && h_) { check(bl_grid_download(h_, host_.data()), "bl_grid_download"); hostValid_ = true; }
    }
    void syncToDevice() const
    {
        if (!deviceValid_ && h_) { check(bl_grid_upload(h_, host_.data()), "bl_grid_upload"); deviceValid_ = true; }
    }
};

// ------------------------------------------------------------------------------------------------ helpers
template <class Lidar>
inline bl_lidar_t lidar_view(const Lidar& scan)
{
    bl_lidar_t v;
    v.utime = scan.utime; v.num_ranges = scan.num_ranges;
    v.ranges = scan.ranges.data(); v.thetas = scan.thetas.data(); v.times = scan.times.data();
    v.intensities = nullptr;
    return v;
}
template <class Pose> inline bl_pose_xyt_t pose_in(const Pose& p) { bl_pose_xyt_t o; o.utime = p.utime; o.x = p.x; o.y = p.y; o.theta = p.theta; return o; }
template <class Pose> inline Pose pose_out(const bl_pose_xyt_t& p) { Pose o; o.utime = p.utime; o.x = p.x; o.y = p.y; o.theta = p.theta; return o; }

// ------------------------------------------------------------------------------------------------ Mapping
template <class Pose, class Lidar>
class MappingT {
public:
    MappingT(float maxLaserDistance, int8_t hitOdds, int8_t missOdds) : h_(nullptr)
    {
        check(bl_mapping_create(default_ctx(), maxLaserDistance, hitOdds, missOdds, &h_), "bl_mapping_create");
    }
    ~MappingT() { bl_mapping_destroy(h_); }
    MappingT(const MappingT&) = delete;
    MappingT& operator=(const MappingT&) = delete;

    void updateMap(const Lidar& scan, const Pose& pose, OccupancyGrid& map)     // mapping.hpp:34
    {
        bl_lidar_t v = lidar_view(scan);
        bl_pose_xyt_t p = pose_in(pose);
        check(bl_mapping_update(h_, &v, &p, map.device()), "bl_mapping_update");
        map.markDeviceWritten();
    }
    // updateMap with the END of a filter update begun by ParticleFilterT::updateFilterBegin folded in: one launch forms the pose
    // estimate, updates the map with it and writes the filter's weight prefix (bl_mapping_update_finishing_pf).  The pose is
    // the filter's poseEstimate() afterwards; poseUtime is the odometry utime of the update (posteriorPose_.utime).
    template <class Filter>
    void updateMapFinishingFilter(const Lidar& scan, Filter& filter, int64_t poseUtime, OccupancyGrid& map)
    {
        bl_lidar_t v = lidar_view(scan);
        check(bl_mapping_update_finishing_pf(h_, &v, filter.device(), poseUtime, map.device()), "bl_mapping_update_finishing_pf");
        map.markDeviceWritten();
    }
private:
    bl_mapping* h_;
};

// The NEXT scan handed over early (a SLAM host has it queued, slam.cpp:96-104): the next map kernel copies it to the device
// beside its own work (bl_scan_prefetch).
template <class Lidar>
inline void prefetch_scan(const Lidar& scan)
{
    bl_lidar_t v = lidar_view(scan);
    check(bl_scan_prefetch(default_ctx(), &v), "bl_scan_prefetch");
}

// ------------------------------------------------------------------------------------------------ ParticleFilter
template <class Pose, class Lidar, class Particle, class Particles>
class ParticleFilterT {
public:
    explicit ParticleFilterT(int numParticles) : h_(nullptr), n_(numParticles)
    {
        assert(numParticles > 1);                                               // particle_filter.cpp:11
        check(bl_pf_create(default_ctx(), numParticles, 0, numParticles, &h_), "bl_pf_create");
    }
    ~ParticleFilterT() { bl_pf_destroy(h_); }
    ParticleFilterT(const ParticleFilterT&) = delete;
    ParticleFilterT& operator=(const ParticleFilterT&) = delete;

    void initializeFilterAtPose(const Pose& pose)                               // particle_filter.cpp:16-34
    {
        bl_pose_xyt_t p = pose_in(pose);
        uint64_t seed = 0;                                                      // reference: std::random_device
        std::ifstream rnd("/dev/urandom", std::ios::binary);
        if (rnd) rnd.read(reinterpret_cast<char*>(&seed), sizeof(seed));
        check(bl_pf_init_at_pose(h_, &p, seed), "bl_pf_init_at_pose");
    }
    Pose updateFilter(const Pose& odometry, const Lidar& laser, const OccupancyGrid& map)   // particle_filter.cpp:37-52
    {
        bl_lidar_t v = lidar_view(laser);
        bl_pose_xyt_t o = pose_in(odometry), out;
        // the reference draws the low-variance sampler's offset from rand() once per moved update (particle_filter.cpp:92)
        check(bl_pf_update(h_, &o, &v, map.device(), rand(), nullptr, &out), "bl_pf_update");
        return pose_out<Pose>(out);
    }
    // First half of updateFilter: resampling, action and sensor model are launched; the update is ended by
    // MappingT::updateMapFinishingFilter (one launch with the map update) -- or by updateFilterEnd().  Returns "moved".
    bool updateFilterBegin(const Pose& odometry, const Lidar& laser, const OccupancyGrid& map)
    {
        bl_pose_xyt_t o = pose_in(odometry);
        bl_lidar_t v = lidar_view(laser);
        int moved = 0;
        check(bl_pf_update_begin(h_, &o, &v, map.device(), rand(), nullptr, &moved), "bl_pf_update_begin");
        return moved != 0;
    }
    Pose updateFilterEnd()
    {
        bl_pose_xyt_t out;
        check(bl_pf_update_end(h_, &out), "bl_pf_update_end");
        return pose_out<Pose>(out);
    }

    Pose updateFilterActionOnly(const Pose& odometry)                           // particle_filter.cpp:54-65
    {
        bl_pose_xyt_t o = pose_in(odometry), out;
        check(bl_pf_update_action_only(h_, &o, nullptr, &out), "bl_pf_update_action_only");
        return pose_out<Pose>(out);
    }
    Pose poseEstimate() const                                                   // particle_filter.cpp:69-72
    {
        bl_pose_xyt_t out;
        check(bl_pf_pose_estimate(h_, &out), "bl_pf_pose_estimate");
        return pose_out<Pose>(out);
    }
    Particles particles() const                                                 // particle_filter.cpp:75-81
    {
        std::vector<bl_particle_t> raw(n_);
        check(bl_pf_get_particles(h_, raw.data()), "bl_pf_get_particles");
        Particles out;
        out.num_particles = n_;
        out.particles.resize(n_);
        for (int i = 0; i < n_; ++i) {
            out.particles[i].pose = pose_out<Pose>(raw[i].pose);
            out.particles[i].parent_pose = pose_out<Pose>(raw[i].parent_pose);
            out.particles[i].weight = raw[i].weight;
        }
        return out;
    }
    // (extension) resample against the reference's own sequentially rounded cumulative weight: identical source indices for
    // every rand() value, at the price of an extra launch per update (botlab_hip.h, bl_pf_set_strict_resampling)
    void setStrictResampling(bool on) { check(bl_pf_set_strict_resampling(h_, on ? 1 : 0), "bl_pf_set_strict_resampling"); }
    bl_pf* device() const { return h_; }                                        // for the device-side extras (bl_pf_encode_particles_lcm ...)
private:
    bl_pf* h_;
    int n_;
};

// ------------------------------------------------------------------------------------------------ ObstacleDistanceGrid
class ObstacleDistanceGrid {
public:
    ObstacleDistanceGrid() : h_(nullptr), hostValid_(false) { check(bl_dist_create(default_ctx(), &h_), "bl_dist_create"); }
    ObstacleDistanceGrid(const ObstacleDistanceGrid& o) : h_(nullptr), hostValid_(false)
    {
        check(bl_dist_create(default_ctx(), &h_), "bl_dist_create");
        src_ = o.src_;
        if (src_.widthInCells() > 0) setDistances(src_);       // deep copy = recompute from the remembered map (bit-identical)
    }
    ObstacleDistanceGrid& operator=(const ObstacleDistanceGrid& o)
    {
        if (this != &o) { src_ = o.src_; hostValid_ = false; if (src_.widthInCells() > 0) setDistances(src_); }
        return *this;
    }
    ~ObstacleDistanceGrid() { bl_dist_destroy(h_); }

    int widthInCells() const { int w = 0, h = 0; bl_dist_shape(h_, &w, &h); return w; }
    int heightInCells() const { int w = 0, h = 0; bl_dist_shape(h_, &w, &h); return h; }
    float metersPerCell() const { float m, c, x, y; bl_dist_frame(h_, &m, &c, &x, &y); return m; }
    float cellsPerMeter() const { float m, c, x, y; bl_dist_frame(h_, &m, &c, &x, &y); return c; }
    float widthInMeters() const { return widthInCells() * metersPerCell(); }
    float heightInMeters() const { return heightInCells() * metersPerCell(); }
    PointT<float> originInGlobalFrame() const { float m, c, x, y; bl_dist_frame(h_, &m, &c, &x, &y); return PointT<float>(x, y); }

    void setDistances(const OccupancyGrid& map)                                 // obstacle_distance_grid.cpp:73-91
    {
        check(bl_dist_set_distances(h_, map.device()), "bl_dist_set_distances");
        if (&map != &src_) src_ = map;
        hostValid_ = false;
    }
    bool isCellInGrid(int x, int y) const { return (x >= 0) && (x < widthInCells()) && (y >= 0) && (y < heightInCells()); }
    float operator()(int x, int y) const
    {
        if (!hostValid_) {
            host_.resize(static_cast<size_t>(widthInCells()) * heightInCells());
            check(bl_dist_download(h_, host_.data()), "bl_dist_download");
            hostValid_ = true;
        }
        return host_[static_cast<size_t>(y) * widthInCells() + x];
    }
    bl_dist* device() const { return h_; }
private:
    bl_dist* h_;
    OccupancyGrid src_;
    mutable std::vector<float> host_;
    mutable bool hostValid_;
};

// ------------------------------------------------------------------------------------------------ search_for_path
struct SearchParams {                       // astar.hpp:15-27
    double minDistanceToObstacle;
    double maxDistanceWithCost;
    double distanceCostExponent;
};

template <class Path, class Pose>
Path search_for_path_t(Pose start, Pose goal, const ObstacleDistanceGrid& distances, const SearchParams& params)
{
    bl_search_params_t sp = {params.minDistanceToObstacle, params.maxDistanceWithCost, params.distanceCostExponent};
    bl_pose_xyt_t s = pose_in(start), g = pose_in(goal);
    std::vector<bl_pose_xyt_t> buf(1024);
    int len = 0;
    int rc = bl_astar_search(default_ctx(), distances.device(), &s, &g, &sp, buf.data(), static_cast<int>(buf.size()), &len, nullptr);
    if (rc == BL_OK && len > static_cast<int>(buf.size())) {                    // longer than the first buffer: fetch again
        buf.resize(len);
        rc = bl_astar_search(default_ctx(), distances.device(), &s, &g, &sp, buf.data(), len, &len, nullptr);
    }
    check(rc, "bl_astar_search");
    Path path;
    path.utime = start.utime;                                                   // astar.cpp:20
    for (int i = 0; i < len; ++i) path.path.push_back(pose_out<Pose>(buf[i]));
    path.path_length = static_cast<int32_t>(path.path.size());
    return path;
}

}  // namespace botlab_hip

#endif  // BOTLAB_DROPIN_HPP
