// planning_dropin.hpp -- MotionPlanner, frontier_t, find_map_frontiers and plan_path_to_frontier with the reference's
// names and call signatures (src/planning/motion_planner.hpp:51-167, src/planning/frontiers.hpp:17-57), forwarding to
// libbotlab_hip.so.  Header-only host C++ over the C ABI; templates over the lcm-gen message types like
// botlab_dropin.hpp (INTEGRATION.md shows the typedefs a botLab checkout adds).
#ifndef BOTLAB_PLANNING_DROPIN_HPP
#define BOTLAB_PLANNING_DROPIN_HPP

#include <chrono>
#include <cmath>
#include <vector>

#include <botlab/botlab_dropin.hpp>

namespace botlab_hip {

struct MotionPlannerParams {                       // motion_planner.hpp:16-35
    double robotRadius;
    MotionPlannerParams() : robotRadius(0.2) {}
};

struct frontier_t {                                // frontiers.hpp:17-20
    std::vector<PointT<float>> cells;
};

inline int64_t utime_now_us()                       // common/timestamp.c utime_now(): wall clock in microseconds
{
    return std::chrono::duration_cast<std::chrono::microseconds>(std::chrono::system_clock::now().time_since_epoch()).count();
}

template <class Pose, class Path>
class MotionPlannerT {
public:
    explicit MotionPlannerT(const MotionPlannerParams& params = MotionPlannerParams()) : params_(params)   // motion_planner.cpp:9-13
    {
        init_state();
        setParams(params);
    }
    MotionPlannerT(const MotionPlannerParams& params, const SearchParams& searchParams) : params_(params), searchParams_(searchParams)
    {
        init_state();                                                       // motion_planner.cpp:16-20
    }

    Path planPath(const Pose& start, const Pose& goal, const SearchParams& searchParams) const   // motion_planner.cpp:23-43
    {
        if (!isValidGoal(goal)) {
            Path failedPath;
            failedPath.utime = utime_now_us();
            failedPath.path_length = 1;
            failedPath.path.push_back(start);
            return failedPath;
        }
        return search_for_path_t<Path, Pose>(start, goal, distances_, searchParams);
    }
    Path planPath(const Pose& start, const Pose& goal) const { return planPath(start, goal, searchParams_); }

    bool isValidGoal(const Pose& goal) const                                // motion_planner.cpp:52-74
    {
        float dx = goal.x - prev_goal_.x, dy = goal.y - prev_goal_.y;
        float distanceFromPrev = std::sqrt(dx * dx + dy * dy);
        if (num_frontiers_ != 1 && distanceFromPrev < 2 * searchParams_.minDistanceToObstacle) return false;
        PointT<float> o = distances_.originInGlobalFrame();
        int32_t cell[2];
        cell[0] = static_cast<int>((static_cast<double>(goal.x) - o.x) * distances_.cellsPerMeter());   // grid_utils.hpp:33-38
        cell[1] = static_cast<int>((static_cast<double>(goal.y) - o.y) * distances_.cellsPerMeter());
        if (distances_.isCellInGrid(cell[0], cell[1])) {
            float d = 0;
            check(bl_dist_gather(distances_.device(), cell, 1, &d), "bl_dist_gather");
            return d > params_.robotRadius;
        }
        return false;
    }

    bool isPathSafe(const Path& path) const                                 // motion_planner.cpp:77-96 (outside the grid: unsafe)
    {
        const float mpc = distances_.metersPerCell();
        const int w = distances_.widthInCells(), h = distances_.heightInCells();
        std::vector<int32_t> q;
        for (unsigned i = 0; i < path.path.size(); i++) {
            int x = path.path[i].x / mpc + w / 2;
            int y = path.path[i].y / mpc + h / 2;
            q.push_back(x); q.push_back(y);
        }
        std::vector<float> d(path.path.size());
        check(bl_dist_gather(distances_.device(), q.data(), static_cast<int>(d.size()), d.data()), "bl_dist_gather");
        for (size_t i = 0; i < d.size(); ++i)
            if (d[i] != d[i] || d[i] <= searchParams_.minDistanceToObstacle) return false;
        return true;
    }

    void setMap(const OccupancyGrid& map) { distances_.setDistances(map); }   // motion_planner.cpp:99-102
    void setParams(const MotionPlannerParams&)                              // motion_planner.cpp:105-110 reads params_, not the argument
    {
        searchParams_.minDistanceToObstacle = params_.robotRadius;
        searchParams_.maxDistanceWithCost = 10.0 * searchParams_.minDistanceToObstacle;
        searchParams_.distanceCostExponent = 1.0;
    }
    void setPrevGoal(const Pose& goal) { prev_goal_ = pose_in(goal); }
    void setNumFrontiers(const size_t& num_f) { num_frontiers_ = num_f; }
    ObstacleDistanceGrid obstacleDistances(void) const { return distances_; }

    // what plan_path_to_frontier_t hands to the library
    const ObstacleDistanceGrid& distances() const { return distances_; }
    bl_motion_planner_t state() const
    {
        bl_motion_planner_t s;
        s.robot_radius = params_.robotRadius;
        s.search.minDistanceToObstacle = searchParams_.minDistanceToObstacle;
        s.search.maxDistanceWithCost = searchParams_.maxDistanceWithCost;
        s.search.distanceCostExponent = searchParams_.distanceCostExponent;
        s.num_frontiers = static_cast<int32_t>(num_frontiers_);
        s.prev_goal = prev_goal_;
        return s;
    }

private:
    // num_frontiers / prev_goal are uninitialised in the reference (motion_planner.hpp:164-165); defined here as "one
    // frontier, previous goal far away" so that isValidGoal's proximity test is off until the setters are called.
    void init_state() { num_frontiers_ = 1; prev_goal_.utime = 0; prev_goal_.x = 1e9f; prev_goal_.y = 1e9f; prev_goal_.theta = 0; }

    ObstacleDistanceGrid distances_;
    MotionPlannerParams params_;
    SearchParams searchParams_;
    size_t num_frontiers_;
    bl_pose_xyt_t prev_goal_;
};

// find_map_frontiers (frontiers.hpp:34-36)
template <class Pose>
std::vector<frontier_t> find_map_frontiers_t(const OccupancyGrid& map, const Pose& robotPose, double minFrontierLength = 0.35)
{
    bl_pose_xyt_t p = pose_in(robotPose);
    bl_frontiers* f = nullptr;
    check(bl_frontiers_find(default_ctx(), map.device(), &p, minFrontierLength, &f), "bl_frontiers_find");
    const int n = bl_frontiers_count(f), total = bl_frontiers_total_cells(f);
    std::vector<int32_t> offs(static_cast<size_t>(n) + 1);
    std::vector<float> xy(static_cast<size_t>(total) * 2 + 2);
    check(bl_frontiers_get(f, offs.data(), xy.data()), "bl_frontiers_get");
    bl_frontiers_destroy(f);
    std::vector<frontier_t> out(static_cast<size_t>(n));
    for (int k = 0; k < n; ++k)
        for (int i = offs[k]; i < offs[k + 1]; ++i) out[k].cells.push_back(PointT<float>(xy[2 * static_cast<size_t>(i)], xy[2 * static_cast<size_t>(i) + 1]));
    return out;
}

// plan_path_to_frontier (frontiers.hpp:50-53); `map` is unused by the reference as well
template <class Path, class Pose, class Planner>
Path plan_path_to_frontier_t(const std::vector<frontier_t>& frontiers, const Pose& robotPose, const OccupancyGrid&, const Planner& planner)
{
    std::vector<int32_t> offs(1, 0);
    std::vector<float> xy;
    for (const frontier_t& f : frontiers) {
        for (const PointT<float>& c : f.cells) { xy.push_back(c.x); xy.push_back(c.y); }
        offs.push_back(static_cast<int32_t>(xy.size() / 2));
    }
    bl_frontiers* h = nullptr;
    check(bl_frontiers_from_host(offs.data(), static_cast<int>(frontiers.size()), xy.empty() ? nullptr : xy.data(), &h), "bl_frontiers_from_host");
    bl_pose_xyt_t p = pose_in(robotPose);
    bl_motion_planner_t st = planner.state();
    std::vector<bl_pose_xyt_t> buf(4096);
    int len = 0;
    int rc = bl_plan_path_to_frontier(default_ctx(), h, &p, planner.distances().device(), &st, buf.data(), static_cast<int>(buf.size()), &len, nullptr, nullptr);
    if (rc == BL_OK && len > static_cast<int>(buf.size())) {
        buf.resize(len);
        rc = bl_plan_path_to_frontier(default_ctx(), h, &p, planner.distances().device(), &st, buf.data(), len, &len, nullptr, nullptr);
    }
    bl_frontiers_destroy(h);
    check(rc, "bl_plan_path_to_frontier");
    Path path;                                                                  // emptyPath when there is no frontier (frontiers.cpp:117-120)
    if (len == 0) return path;
    path.utime = robotPose.utime;                                               // astar.cpp:20 (planPath of the chosen goal)
    for (int i = 0; i < len; ++i) path.path.push_back(pose_out<Pose>(buf[i]));
    path.path_length = static_cast<int32_t>(path.path.size());
    return path;
}
// Exploration::executeExploringMap (src/planning/exploration.cpp:277-369) without its LCM calls (the caller publishes the
// status and the path): the per-map step of the exploration loop -- setMap, find_map_frontiers, plan_path_to_frontier when
// the robot is within 0.5 m of the current target (or has none) -- and the status / next-state rule of :332-368.  The
// reference leaves status.status unset when frontiers remain but no path was found (:344-347 is commented out) and so ends
// in the default branch of :365-367: FAILED_EXPLORATION (definition D10; `status` then holds STATUS_FAILED).
// States and statuses: lcmtypes/exploration_status_t.lcm:3-11.
template <class Pose, class Path, class Planner = MotionPlannerT<Pose, Path> >
class ExploringMapT {
public:
    enum { STATE_INITIALIZING = 0, STATE_EXPLORING_MAP = 1, STATE_RETURNING_HOME = 2, STATE_COMPLETED_EXPLORATION = 3, STATE_FAILED_EXPLORATION = 4 };
    enum { STATUS_IN_PROGRESS = 0, STATUS_COMPLETE = 1, STATUS_FAILED = 2 };

    explicit ExploringMapT(Planner& planner) : planner_(planner), status(STATUS_FAILED)
    {
        currentTarget_.utime = 0; currentTarget_.x = 0; currentTarget_.y = 0; currentTarget_.theta = 0;
        currentPath_.utime = 0; currentPath_.path_length = 0;
    }

    int8_t execute(const OccupancyGrid& currentMap, const Pose& currentPose)
    {
        planner_.setMap(currentMap);                                                              // :299
        frontiers_ = find_map_frontiers_t<Pose>(currentMap, currentPose);                        // :300
        planner_.setNumFrontiers(frontiers_.size());                                              // :302
        const float distThreshold = 0.5f;
        float currDist;
        if (currentTarget_.x != 0 || currentTarget_.y != 0)                                       // :307-311
            currDist = std::sqrt(std::pow(currentPose.x - currentTarget_.x, 2) + std::pow(currentPose.y - currentTarget_.y, 2));
        else
            currDist = 0;
        if (currDist <= distThreshold && !frontiers_.empty()) {                                   // :316-321
            currentPath_ = plan_path_to_frontier_t<Path, Pose, Planner>(frontiers_, currentPose, currentMap, planner_);
            if (currentPath_.path_length > 1) currentTarget_ = currentPath_.path[currentPath_.path_length - 1];
        }
        if (frontiers_.empty()) status = STATUS_COMPLETE;                                         // :335-347
        else if (currentPath_.path.size() > 1) status = STATUS_IN_PROGRESS;
        else status = STATUS_FAILED;                                                              // D10
        switch (status) {                                                                         // :352-368
        case STATUS_IN_PROGRESS: return STATE_EXPLORING_MAP;
        case STATUS_COMPLETE: return STATE_RETURNING_HOME;
        default: return STATE_FAILED_EXPLORATION;
        }
    }

    Planner& planner_;
    Pose currentTarget_;
    Path currentPath_;
    std::vector<frontier_t> frontiers_;
    int8_t status;
};

// The same step for a host that runs SLAM and exploration in one process: bl_explorer (botlab_hip.h) snapshots the map and the
// filter's device-resident pose behind a map update and runs setMap + find_map_frontiers on a side stream; fetch() -- from this
// thread or from an exploration thread of its own -- applies the 0.5 m rule and plans when due (exploration.cpp:299-368).
template <class Pose, class Path>
class AsyncExploringMapT {
public:
    explicit AsyncExploringMapT(int lanes = 2, double robotRadius = 0.2) : status(2), h_(nullptr), lanes_(lanes)
    {
        if (bl_explorer_create(default_ctx(), lanes, robotRadius, &h_) != BL_OK) { std::fprintf(stderr, "bl_explorer_create: %s\n", bl_last_error()); std::abort(); }
        currentTarget_.utime = 0; currentTarget_.x = 0; currentTarget_.y = 0; currentTarget_.theta = 0;
        currentPath_.utime = 0; currentPath_.path_length = 0;
    }
    ~AsyncExploringMapT() { if (h_) bl_explorer_destroy(h_); }
    AsyncExploringMapT(const AsyncExploringMapT&) = delete;
    AsyncExploringMapT& operator=(const AsyncExploringMapT&) = delete;

    // behind Mapping::updateMap of the step whose pose `filter` holds on the device; false: every lane holds a step (fetch first)
    template <class Filter>
    bool submit(const OccupancyGrid& currentMap, const Filter& filter)
    {
        if (bl_explorer_pending(h_) >= lanes()) return false;
        if (bl_explorer_submit(h_, currentMap.device(), bl_pf_pose_device_ptr(filter.device())) != BL_OK) { std::fprintf(stderr, "bl_explorer_submit: %s\n", bl_last_error()); std::abort(); }
        return true;
    }
    int pending() const { return bl_explorer_pending(h_); }
    // the oldest submitted step: the next state (exploration_status_t); status / currentTarget_ / currentPath_ as ExploringMapT keeps them
    int8_t fetch(bl_explore_result_t* info = nullptr)
    {
        std::vector<bl_pose_xyt_t> buf(65536);
        bl_explore_result_t r;
        if (bl_explorer_fetch(h_, &r, buf.data(), (int)buf.size()) != BL_OK) { std::fprintf(stderr, "bl_explorer_fetch: %s\n", bl_last_error()); std::abort(); }
        status = (int8_t)r.status;
        currentTarget_.utime = r.target.utime; currentTarget_.x = r.target.x; currentTarget_.y = r.target.y; currentTarget_.theta = r.target.theta;
        currentPath_.path.resize((size_t)r.path_length);
        for (int i = 0; i < r.path_length; ++i) {
            currentPath_.path[(size_t)i].utime = buf[(size_t)i].utime; currentPath_.path[(size_t)i].x = buf[(size_t)i].x;
            currentPath_.path[(size_t)i].y = buf[(size_t)i].y; currentPath_.path[(size_t)i].theta = buf[(size_t)i].theta;
        }
        currentPath_.path_length = r.path_length;
        if (info) *info = r;
        return (int8_t)r.next_state;
    }
    bl_explorer* device() const { return h_; }

    Pose currentTarget_;
    Path currentPath_;
    int8_t status;

private:
    int lanes() const { return lanes_; }
    bl_explorer* h_;
    int lanes_;
};

}  // namespace botlab_hip

#endif  // BOTLAB_PLANNING_DROPIN_HPP
