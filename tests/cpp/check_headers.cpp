// Compile check of the drop-in classes (g++ -fsyntax-only): instantiates every template with plain message structs
// that carry the fields of the lcm-gen types (lcmtypes/*.lcm).  Not part of the product; see tests/cpp/dropin_test.cpp
// for the run-time check on a GPU.
#include "dropin_test_types.hpp"

typedef botlab_hip::MappingT<pose_xyt_t, lidar_t> Mapping;
typedef botlab_hip::ParticleFilterT<pose_xyt_t, lidar_t, particle_t, particles_t> ParticleFilter;

robot_path_t plan(pose_xyt_t a, pose_xyt_t b, const botlab_hip::ObstacleDistanceGrid& d, const botlab_hip::SearchParams& p)
{
    return botlab_hip::search_for_path_t<robot_path_t, pose_xyt_t>(a, b, d, p);
}

void touch()
{
    botlab_hip::OccupancyGrid g(10.0f, 10.0f, 0.05f), g2;
    g2 = g;
    occupancy_grid_t m = g.toLCM<occupancy_grid_t>();
    g2.fromLCM(m);
    Mapping mapper(5.0f, 4, 1);
    ParticleFilter pf(200);
    lidar_t scan;
    pose_xyt_t p;
    mapper.updateMap(scan, p, g);
    pf.initializeFilterAtPose(p);
    p = pf.updateFilter(p, scan, g);
    p = pf.updateFilterActionOnly(p);
    particles_t ps = pf.particles();
    (void)ps;
    botlab_hip::ObstacleDistanceGrid d, d2(d);
    d.setDistances(g);
    (void)d(0, 0);
}

// slam_driver.hpp (row f2)
#include <botlab/slam_driver.hpp>
struct odometry_t { int64_t utime = 0; float x = 0, y = 0, theta = 0; };
typedef botlab_hip::OccupancyGridSLAMT<pose_xyt_t, lidar_t, odometry_t, particle_t, particles_t, occupancy_grid_t> OccupancyGridSLAM;
void touch_driver()
{
    OccupancyGridSLAM::Publisher pub;
    OccupancyGridSLAM slam(200, 4, 1, pub, false, false, false, "");
    lidar_t scan; odometry_t odo; pose_xyt_t p;
    slam.handleOdometry(odo); slam.handlePose(p); slam.handleOptitrack(p); slam.handleLaser(scan);
    if (slam.isReadyToUpdate()) slam.runSLAMIteration();
    botlab_hip::PoseTraceT<pose_xyt_t> t; t.addPose(p); (void)t.poseAt(0); t.setReferencePose(p); (void)t.eraseTraceUntil(0);
}

// planning_dropin.hpp instantiations
#include <botlab/planning_dropin.hpp>
typedef botlab_hip::MotionPlannerT<pose_xyt_t, robot_path_t> CheckMotionPlanner;
static void check_planning_dropin(const botlab_hip::OccupancyGrid& map, const pose_xyt_t& pose)
{
    CheckMotionPlanner planner;
    planner.setMap(map);
    std::vector<botlab_hip::frontier_t> fr = botlab_hip::find_map_frontiers_t(map, pose);
    planner.setNumFrontiers(fr.size());
    robot_path_t p = botlab_hip::plan_path_to_frontier_t<robot_path_t>(fr, pose, map, planner);
    (void)planner.isPathSafe(p); (void)planner.isValidGoal(pose); (void)planner.planPath(pose, pose); (void)planner.obstacleDistances();
}
static void check_exploring_map(const botlab_hip::OccupancyGrid& map, const pose_xyt_t& pose)
{
    CheckMotionPlanner planner;
    botlab_hip::ExploringMapT<pose_xyt_t, robot_path_t> ex(planner);
    int8_t next = ex.execute(map, pose);
    (void)next; (void)ex.status; (void)ex.currentPath_; (void)ex.currentTarget_;
}

// sharded_filter.hpp
#include <botlab/sharded_filter.hpp>
static void check_sharded_filter(const int8_t* cells, const bl_pose_xyt_t& pose, const bl_lidar_t& scan)
{
    botlab_hip::ShardedFilterGroup g(100000, std::vector<int>(4, 0), 200, 200, 0.05f, 20.0f, -5.0f, -5.0f, cells);
    g.initializeFilterAtPose(pose, 1);
    g.step(pose, scan, 7);
    (void)g.poseEstimate(); (void)g.particles(); (void)g.mapCells(0, 200, 200); (void)g.world(); (void)g.rank(0);
}

template <class F>
static void check_async_exploring_map(const botlab_hip::OccupancyGrid& map, const F& filter)
{
    botlab_hip::AsyncExploringMapT<pose_xyt_t, robot_path_t> ex(2, 0.2);
    if (ex.submit(map, filter)) { bl_explore_result_t info; (void)ex.fetch(&info); }
    (void)ex.pending(); (void)ex.status; (void)ex.currentPath_; (void)ex.currentTarget_; (void)ex.device();
}

void touch_async_exploring_map(const botlab_hip::OccupancyGrid& map, const ParticleFilter& filter) { check_async_exploring_map(map, filter); }
