// planning_test.cpp -- drives include/botlab/planning_dropin.hpp the way Exploration::executeExploringMap does
// (src/planning/exploration.cpp:300-317): setMap, find_map_frontiers, setNumFrontiers, plan_path_to_frontier; dumps the
// frontiers and the path for tests/test_gpu_frontiers.py to compare with the oracle.
#include <cstdio>
#include <cstdlib>
#include "dropin_test_types.hpp"
#include <botlab/planning_dropin.hpp>

typedef botlab_hip::MotionPlannerT<pose_xyt_t, robot_path_t> MotionPlanner;
using botlab_hip::frontier_t;

int main(int argc, char** argv)
{
    if (argc < 7) return 2;
    botlab_hip::OccupancyGrid map;
    if (!map.loadFromFile(argv[1])) return 2;
    pose_xyt_t pose; pose.utime = 42; pose.x = std::atof(argv[2]); pose.y = std::atof(argv[3]); pose.theta = std::atof(argv[4]);
    botlab_hip::MotionPlannerParams params; params.robotRadius = std::atof(argv[5]);
    FILE* out = std::fopen(argv[6], "wb");
    if (!out) return 2;
    MotionPlanner planner(params);
    planner.setMap(map);
    std::vector<frontier_t> frontiers = botlab_hip::find_map_frontiers_t(map, pose);
    planner.setNumFrontiers(frontiers.size());
    robot_path_t path = botlab_hip::plan_path_to_frontier_t<robot_path_t>(frontiers, pose, map, planner);
    int32_t nf = (int32_t)frontiers.size();
    std::fwrite(&nf, 4, 1, out);
    for (const frontier_t& f : frontiers) {
        int32_t n = (int32_t)f.cells.size();
        std::fwrite(&n, 4, 1, out);
        for (const auto& c : f.cells) { std::fwrite(&c.x, 4, 1, out); std::fwrite(&c.y, 4, 1, out); }
    }
    int32_t flags[3];
    flags[0] = path.path_length;
    flags[1] = path.path_length > 1 ? (planner.isPathSafe(path) ? 1 : 0) : -1;
    flags[2] = path.path_length > 1 ? (planner.isValidGoal(path.path.back()) ? 1 : 0) : -1;
    std::fwrite(flags, 4, 3, out);
    for (const pose_xyt_t& p : path.path) { std::fwrite(&p.utime, 8, 1, out); std::fwrite(&p.x, 4, 1, out); std::fwrite(&p.y, 4, 1, out); std::fwrite(&p.theta, 4, 1, out); }
    // planPath to an invalid goal: the failed path is [start] (motion_planner.cpp:28-40)
    pose_xyt_t bad; bad.x = 1e6f; bad.y = 0;
    robot_path_t failed = planner.planPath(pose, bad);
    int32_t fl = failed.path_length; std::fwrite(&fl, 4, 1, out);
    // Exploration::executeExploringMap (exploration.cpp:277-369) twice from the same pose: the second call has a target
    // (the end of the first path) more than 0.5 m away, so it keeps the path and does not plan again
    MotionPlanner planner2(params);
    botlab_hip::ExploringMapT<pose_xyt_t, robot_path_t> ex(planner2);
    int32_t ex_out[6];
    ex_out[0] = ex.execute(map, pose); ex_out[1] = ex.status; ex_out[2] = ex.currentPath_.path_length;
    ex_out[3] = ex.execute(map, pose); ex_out[4] = ex.status; ex_out[5] = ex.currentPath_.path_length;
    std::fwrite(ex_out, 4, 6, out);
    std::fwrite(&ex.currentTarget_.x, 4, 1, out); std::fwrite(&ex.currentTarget_.y, 4, 1, out);
    std::fclose(out);
    std::printf("planning_test ok: %d frontiers, path of %d poses\n", nf, path.path_length);
    return 0;
}
