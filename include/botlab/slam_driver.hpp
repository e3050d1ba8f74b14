// slam_driver.hpp -- the step scheduler around the kernels: PoseTrace
// (src/common/pose_trace.{hpp,cpp}) and OccupancyGridSLAM (src/slam/slam.{hpp,cpp}) with the LCM transport replaced by
// plain method calls (handlers in, publisher callbacks out).  Host-only C++; every decision -- which scans are queued,
// when an update may run, which pose feeds the mapper, what is published and how often -- follows the reference line by
// line.  The classes are templates over the lcm-gen message types like the rest of include/botlab/.
#ifndef BOTLAB_SLAM_DRIVER_HPP
#define BOTLAB_SLAM_DRIVER_HPP

#include <algorithm>
#include <cmath>
#include <cstdint>
#include <deque>
#include <functional>
#include <iostream>
#include <string>
#include <vector>

#include "botlab_dropin.hpp"

namespace botlab_hip {

// ---------------------------------------------------------------- angle / interpolation helpers (host doubles, as the reference)
inline float host_wrap_to_pi(float angle)                              // src/common/angle_functions.hpp:12-24
{
    if (angle < -M_PI) { for (; angle < -M_PI; angle += 2.0 * M_PI); }
    else if (angle > M_PI) { for (; angle > M_PI; angle -= 2.0 * M_PI); }
    return angle;
}
inline double host_angle_diff(double l, double r)                      // :78-87
{
    double diff = l - r;
    if (std::fabs(diff) > M_PI) diff -= (diff > 0) ? M_PI * 2 : M_PI * -2;
    return diff;
}
inline double host_angle_sum(double a, double b)                       // :128-138
{
    double sum = a + b;
    if (std::fabs(sum) > M_PI) sum -= (sum > 0) ? M_PI * 2 : M_PI * -2;
    return sum;
}
template <class Pose>
Pose host_interpolate_pose_by_time(int64_t time, const Pose& before, const Pose& after)   // src/common/interpolation.hpp:23-50
{
    if (before.utime == after.utime) { Pose p = after; p.utime = time; return p; }
    double ratio = static_cast<double>(time - before.utime) / static_cast<double>(after.utime - before.utime);
    double xStep = (after.x - before.x) * ratio;
    double yStep = (after.y - before.y) * ratio;
    double thetaStep = host_angle_diff(after.theta, before.theta) * ratio;
    Pose out;
    out.utime = time;
    out.x = before.x + xStep;
    out.y = before.y + yStep;
    out.theta = host_angle_sum(before.theta, thetaStep);
    return out;
}

// ---------------------------------------------------------------- PoseTrace (src/common/pose_trace.cpp)
template <class Pose>
class PoseTraceT {
public:
    PoseTraceT() { frameTransform_.utime = 0; frameTransform_.x = 0.0f; frameTransform_.y = 0.0f; frameTransform_.theta = 0.0f; }

    void addPose(const Pose& pose) { trace_.push_back(applyFrameTransform(pose, frameTransform_)); }     // :19-22

    int eraseTraceUntil(int64_t time)                                                                     // :25-35
    {
        auto it = std::remove_if(trace_.begin(), trace_.end(), [time](const Pose& p) { return p.utime < time; });
        int numRemoved = static_cast<int>(std::distance(it, trace_.end()));
        trace_.erase(it, trace_.end());
        return numRemoved;
    }

    Pose poseAt(int64_t time) const                                                                       // :38-68
    {
        if (trace_.empty()) {
            std::cerr << "ERROR: PoseTrace::poseAt: No odometry measurements to interpolate.\n";
            Pose zero; zero.utime = 0; zero.x = zero.y = zero.theta = 0;
            return zero;
        } else if (time < trace_.front().utime) {
            std::cerr << "ERROR: PoseTrace::poseAt: No odometry measurements before " << time << " Closest time:"
                      << trace_.front().utime << " Returning that pose.\n";
            return trace_.front();
        } else if (time > trace_.back().utime) {
            std::cerr << "ERROR: PoseTrace::poseAt: No odometry measurements after " << time << " Closest time:"
                      << trace_.back().utime << " Returning that pose.\n";
            return trace_.back();
        }
        Pose interpolated; interpolated.utime = 0; interpolated.x = interpolated.y = interpolated.theta = 0;
        for (std::size_t i = 1; i < trace_.size(); ++i) {
            if ((trace_[i - 1].utime <= time) && (time <= trace_[i].utime)) {
                interpolated = host_interpolate_pose_by_time(time, trace_[i - 1], trace_[i]);
                break;
            }
        }
        return interpolated;
    }

    bool containsPoseAtTime(int64_t time) const                                                           // :71-79
    {
        if (trace_.empty()) return false;
        return (trace_.front().utime <= time) && (time <= trace_.back().utime);
    }

    void setReferencePose(const Pose& initialInReferenceFrame)                                            // :82-114
    {
        Pose initialPose; initialPose.utime = 0;
        if (trace_.empty()) { initialPose.x = 0.0f; initialPose.y = 0.0f; initialPose.theta = 0.0f; }
        else { initialPose.x = trace_.front().x; initialPose.y = trace_.front().y; initialPose.theta = trace_.front().theta; }
        double deltaTheta = initialInReferenceFrame.theta - initialPose.theta;
        double xRotated = initialPose.x * std::cos(deltaTheta) - initialPose.y * std::sin(deltaTheta);
        double yRotated = initialPose.x * std::sin(deltaTheta) + initialPose.y * std::cos(deltaTheta);
        frameTransform_.x = initialInReferenceFrame.x - xRotated;
        frameTransform_.y = initialInReferenceFrame.y - yRotated;
        frameTransform_.theta = deltaTheta;
        for (auto& p : trace_) p = applyFrameTransform(p, frameTransform_);
    }

    Pose getFrameTransform() const { return frameTransform_; }
    void clear() { trace_.clear(); }
    bool empty() const { return trace_.empty(); }
    std::size_t size() const { return trace_.size(); }
    const Pose& operator[](int i) const { return trace_[i]; }
    const Pose& front() const { return trace_.front(); }
    const Pose& back() const { return trace_.back(); }

    static Pose applyFrameTransform(const Pose& pose, const Pose& transform)                              // :117-128
    {
        Pose out;
        out.utime = pose.utime;
        out.x = (pose.x * std::cos(transform.theta) - pose.y * std::sin(transform.theta)) + transform.x;   // float cosf/sinf
        out.y = (pose.x * std::sin(transform.theta) + pose.y * std::cos(transform.theta)) + transform.y;
        out.theta = host_wrap_to_pi(pose.theta + transform.theta);
        return out;
    }

private:
    std::vector<Pose> trace_;
    Pose frameTransform_;
};

// ---------------------------------------------------------------- OccupancyGridSLAM (src/slam/slam.cpp)
enum SlamMode { kModeMappingOnly, kModeLocalizationOnly, kModeActionOnly, kModeFullSlam };                // slam.hpp:72-78

template <class Pose, class Lidar, class Odometry, class Particle, class Particles, class GridMsg>
class OccupancyGridSLAMT {
public:
    struct Publisher {                                   // stands where lcm_.publish(...) stands (slam.cpp:267-268, 285-289)
        std::function<void(const Pose&)> slamPose;
        std::function<void(const Particles&)> slamParticles;
        std::function<void(const GridMsg&)> slamMap;
    };

    OccupancyGridSLAMT(int numParticles, int8_t hitOddsIncrease, int8_t missOddsDecrease, const Publisher& pub,
                       bool waitForOptitrack, bool mappingOnlyMode, bool actionOnlyMode, const std::string& localizationOnlyMap)
        : mode_(kModeFullSlam), haveInitializedPoses_(false), waitingForOptitrack_(waitForOptitrack), haveMap_(false),
          numIgnoredScans_(0), filter_(numParticles), map_(10.0f, 10.0f, 0.05f), mapper_(5.0f, hitOddsIncrease, missOddsDecrease),
          pub_(pub), mapUpdateCount_(0)
    {
        if (mappingOnlyMode) mode_ = kModeMappingOnly;
        else if (localizationOnlyMap.length() > 0) {
            haveMap_ = map_.loadFromFile(localizationOnlyMap);
            mode_ = actionOnlyMode ? kModeActionOnly : kModeLocalizationOnly;
        }
        currentOdometry_ = zeroPose();
        currentScan_.utime = 0;
        initialPose_ = zeroPose(); previousPose_ = zeroPose(); currentPose_ = zeroPose();
    }

    // ---- message handlers (slam.cpp:90-160)
    void handleLaser(const Lidar& scan)
    {
        bool haveOdom = (mode_ != kModeMappingOnly) && !odometryPoses_.empty() && (odometryPoses_.front().utime <= scan.times.front());
        bool havePose = (mode_ == kModeMappingOnly) && !groundTruthPoses_.empty() && (groundTruthPoses_.front().utime <= scan.times.front());
        if (haveOdom || havePose) {
            incomingScans_.push_back(scan);
            if (numIgnoredScans_ >= 10) numIgnoredScans_ = 0;
        } else {
            ++numIgnoredScans_;
        }
    }
    void handleOdometry(const Odometry& odometry)
    {
        Pose p; p.utime = odometry.utime; p.x = odometry.x; p.y = odometry.y; p.theta = odometry.theta;
        odometryPoses_.addPose(p);
    }
    void handlePose(const Pose& pose) { groundTruthPoses_.addPose(pose); }
    void handleOptitrack(const Pose& pose) { if (waitingForOptitrack_) { initialPose_ = pose; waitingForOptitrack_ = false; } }

    // One launch for "end of updateFilter + updateMap + fetch of the next queued scan" (the default), or the reference's
    // call-by-call order (false).  Results are bit-identical; with the fused step SLAM_POSE / SLAM_PARTICLES of an
    // iteration are published after its map update has been enqueued instead of before.
    void setFusedStep(bool on) { fusedStep_ = on; }

    bool isReadyToUpdate() const                                                                          // :163-188
    {
        bool haveData = false;
        if (!incomingScans_.empty()) {
            const Lidar& nextScan = incomingScans_.front();
            bool haveNewOdom = (mode_ != kModeMappingOnly) && odometryPoses_.containsPoseAtTime(nextScan.times.front());
            bool haveNewPose = (mode_ == kModeMappingOnly) && groundTruthPoses_.containsPoseAtTime(nextScan.times.front());
            haveData = haveNewOdom || haveNewPose;
        }
        return haveData && !waitingForOptitrack_;
    }

    void runSLAMIteration()                                                                               // :191-207
    {
        copyDataForSLAMUpdate();
        initializePosesIfNeeded();
        if (currentScan_.num_ranges > 100) { updateLocalization(); updateMap(); }
        else std::cerr << "ERROR: OccupancyGridSLAM: Detected invalid laser scan with " << currentScan_.num_ranges << " ranges.\n";
    }

    // ---- inspection (tests)
    const OccupancyGrid& map() const { return map_; }
    Pose currentPose() const { return currentPose_; }
    int numIgnoredScans() const { return numIgnoredScans_; }
    std::size_t queuedScans() const { return incomingScans_.size(); }
    int mapUpdateCount() const { return mapUpdateCount_; }

private:
    typedef ParticleFilterT<Pose, Lidar, Particle, Particles> Filter;
    typedef MappingT<Pose, Lidar> Mapper;

    SlamMode mode_;
    std::deque<Lidar> incomingScans_;
    PoseTraceT<Pose> groundTruthPoses_, odometryPoses_;
    Lidar currentScan_;
    Pose currentOdometry_;
    Pose initialPose_, previousPose_, currentPose_;
    bool haveInitializedPoses_, waitingForOptitrack_, haveMap_;
    int numIgnoredScans_;
    Filter filter_;
    OccupancyGrid map_;
    Mapper mapper_;
    Publisher pub_;
    int mapUpdateCount_;
    bool fusedStep_ = true;      // updateFilter's end and the next scan's fetch ride in the map kernel (setFusedStep)
    bool endRides_ = false;

    static Pose zeroPose() { Pose p; p.utime = 0; p.x = p.y = p.theta = 0.0f; return p; }

    void copyDataForSLAMUpdate()                                                                          // :210-229
    {
        currentScan_ = incomingScans_.front();
        incomingScans_.pop_front();
        if (mode_ == kModeMappingOnly) { previousPose_ = currentPose_; currentPose_ = groundTruthPoses_.poseAt(currentScan_.times.back()); }
        else currentOdometry_ = odometryPoses_.poseAt(currentScan_.times.back());
    }
    void initializePosesIfNeeded()                                                                        // :232-250
    {
        if (!haveInitializedPoses_) {
            previousPose_ = initialPose_;
            previousPose_.utime = currentScan_.times.front();
            currentPose_ = previousPose_;
            currentPose_.utime = currentScan_.times.back();
            haveInitializedPoses_ = true;
            filter_.initializeFilterAtPose(previousPose_);
        }
    }
    void updateLocalization()                                                                             // :253-271
    {
        if (haveMap_ && (mode_ != kModeMappingOnly)) {
            previousPose_ = currentPose_;
            if (mode_ == kModeActionOnly) currentPose_ = filter_.updateFilterActionOnly(currentOdometry_);
            else if (fusedStep_) {
                // the filter's end rides in the map update's launch (updateMap below): the pose and the particles are read
                // -- and published -- there, with the same values
                filter_.updateFilterBegin(currentOdometry_, currentScan_, map_);
                if (!incomingScans_.empty()) prefetch_scan(incomingScans_.front());   // the next scan is already queued
                endRides_ = true;
                return;
            }
            else currentPose_ = filter_.updateFilter(currentOdometry_, currentScan_, map_);
            Particles particles = filter_.particles();
            if (pub_.slamPose) pub_.slamPose(currentPose_);                  // SLAM_POSE, then SLAM_PARTICLES (slam.cpp:267-268)
            if (pub_.slamParticles) pub_.slamParticles(particles);
        }
    }
    void updateMap()                                                                                      // :274-294
    {
        // the reference's guard `mode_ != localization_only || mode_ != action_only` is always true: the mapper runs in every mode
        if (endRides_) {
            mapper_.updateMapFinishingFilter(currentScan_, filter_, currentOdometry_.utime, map_);
            endRides_ = false;
            currentPose_ = filter_.poseEstimate();
            Particles particles = filter_.particles();
            if (pub_.slamPose) pub_.slamPose(currentPose_);                  // SLAM_POSE, then SLAM_PARTICLES (slam.cpp:267-268)
            if (pub_.slamParticles) pub_.slamParticles(particles);
        } else
        mapper_.updateMap(currentScan_, currentPose_, map_);
        haveMap_ = true;
        if (mapUpdateCount_ % 5 == 0 && pub_.slamMap) pub_.slamMap(map_.template toLCM<GridMsg>());
        ++mapUpdateCount_;
    }
};

}  // namespace botlab_hip

#endif  // BOTLAB_SLAM_DRIVER_HPP
