// slam_driver.hpp -- the host-side step scheduler around the kernels.
//
// Two classes with the reference's public surfaces and observable behaviour, written from scratch around this library:
//   PoseTraceT          <->  PoseTrace          (src/common/pose_trace.hpp:28-124)
//   OccupancyGridSLAMT  <->  OccupancyGridSLAM  (src/slam/slam.hpp:21-70), LCM subscriptions replaced by plain handler calls
//                                                and lcm_.publish by three callbacks
// Design notes (what differs from the reference's implementation while giving the same answers):
//   * the trace remembers whether its time stamps ascend; if so poseAt() finds the bracketing samples by bisection
//     (proof of equality with the first-match scan below), otherwise it scans;
//   * one accessor names the trace a mode takes its poses from, so "is there a pose for this scan" is written once;
//   * queued scans live in a ring that reuses its slots (a 10 Hz lidar never makes it grow past a handful);
//   * an iteration is by default the fused two-launch step of DESIGN.md section 5 (filter begin; filter end + map update +
//     fetch of the next queued scan in one launch); setFusedStep(false) gives the call-by-call order.
// Threads: like the reference, handlers and runSLAMIteration() must be called under one lock (slam.cpp holds dataMutex_).
#ifndef BOTLAB_SLAM_DRIVER_HPP
#define BOTLAB_SLAM_DRIVER_HPP

#include <cmath>
#include <cstddef>
#include <cstdint>
#include <functional>
#include <iostream>
#include <string>
#include <utility>
#include <vector>

#include "botlab_dropin.hpp"

namespace botlab_hip {

namespace slam_detail {

const double kTwoPi = 2.0 * M_PI;

// An angle that left [-pi, pi] by less than a turn comes back by one turn (what angle_diff / angle_sum do to their
// result, src/common/angle_functions.hpp:78-87, 128-138).
inline double one_turn_back(double a)
{
    if (std::fabs(a) <= M_PI) return a;
    return a > 0 ? a - kTwoPi : a + kTwoPi;
}

// wrap_to_pi of angle_functions.hpp:12-24 on a float: repeated float additions of the double constant.
inline float wrap_pi(float a)
{
    if (a < -M_PI) { do a += kTwoPi; while (a < -M_PI); }
    else if (a > M_PI) { do a -= kTwoPi; while (a > M_PI); }
    return a;
}

template <class Pose>
inline Pose pose_of(int64_t utime, float x, float y, float theta)
{
    Pose p;
    p.utime = utime; p.x = x; p.y = y; p.theta = theta;
    return p;
}

// Pose between two samples, linear in time (src/common/interpolation.hpp:23-50): float differences, double steps, the
// heading along the shorter arc.  Samples with one time stamp give the later one.
template <class Pose>
inline Pose between(const Pose& a, const Pose& b, int64_t t)
{
    if (a.utime == b.utime) { Pose p = b; p.utime = t; return p; }
    const double s = static_cast<double>(t - a.utime) / static_cast<double>(b.utime - a.utime);
    const double turn = one_turn_back(static_cast<double>(b.theta) - static_cast<double>(a.theta)) * s;
    return pose_of<Pose>(t, static_cast<float>(a.x + (b.x - a.x) * s), static_cast<float>(a.y + (b.y - a.y) * s),
                         static_cast<float>(one_turn_back(static_cast<double>(a.theta) + turn)));
}

// A pose expressed in a frame that is rotated by f.theta and shifted by (f.x, f.y) -- float arithmetic throughout
// (pose_trace.cpp:117-128 uses the float overloads of cos / sin).
template <class Pose>
inline Pose into_frame(const Pose& p, const Pose& f)
{
    const float c = std::cos(f.theta), s = std::sin(f.theta);
    return pose_of<Pose>(p.utime, (p.x * c - p.y * s) + f.x, (p.x * s + p.y * c) + f.y, wrap_pi(p.theta + f.theta));
}

}  // namespace slam_detail

// ---------------------------------------------------------------- PoseTrace
template <class Pose>
class PoseTraceT {
public:
    typedef typename std::vector<Pose>::const_iterator const_iterator;

    PoseTraceT() : frame_(slam_detail::pose_of<Pose>(0, 0.0f, 0.0f, 0.0f)), ascending_(true) {}

    // Samples arrive in the source's frame and are stored in the reference frame (pose_trace.hpp:19-27).
    void addPose(const Pose& pose)
    {
        const Pose placed = slam_detail::into_frame(pose, frame_);
        if (!samples_.empty() && placed.utime < samples_.back().utime) ascending_ = false;
        samples_.push_back(placed);
    }

    // Drops every sample older than `time`, wherever it sits; the others keep their order.  Returns how many went.
    int eraseTraceUntil(int64_t time)
    {
        std::size_t kept = 0;
        for (std::size_t i = 0; i < samples_.size(); ++i)
            if (!(samples_[i].utime < time)) {
                if (kept != i) samples_[kept] = samples_[i];
                ++kept;
            }
        const int dropped = static_cast<int>(samples_.size() - kept);
        samples_.resize(kept);
        if (dropped) recheckOrder();
        return dropped;
    }

    // Pose at `time`: interpolated inside the trace, the nearest end outside it (with a complaint, never extrapolated),
    // the zero pose for an empty trace or when no neighbouring pair brackets the time (possible only in a trace whose
    // stamps do not ascend).
    Pose poseAt(int64_t time) const
    {
        const Pose nothing = slam_detail::pose_of<Pose>(0, 0.0f, 0.0f, 0.0f);
        if (samples_.empty()) {
            std::cerr << "PoseTrace::poseAt(" << time << "): the trace is empty\n";
            return nothing;
        }
        if (time < samples_.front().utime || time > samples_.back().utime) {
            const Pose& end = time < samples_.front().utime ? samples_.front() : samples_.back();
            std::cerr << "PoseTrace::poseAt(" << time << "): outside the trace, answering with the sample at " << end.utime << "\n";
            return end;
        }
        const std::size_t hit = ascending_ ? bracketByBisection(time) : bracketByScan(time);
        return hit ? slam_detail::between(samples_[hit - 1], samples_[hit], time) : nothing;
    }

    bool containsPoseAtTime(int64_t time) const
    {
        return !samples_.empty() && !(time < samples_.front().utime) && !(samples_.back().utime < time);
    }

    // Chooses the frame in which the FIRST sample of the trace (the origin if there is none) becomes the given pose; every
    // stored sample moves there, later ones follow on arrival.  A second call composes with the first, as documented for
    // the reference (pose_trace.hpp:76-90).  Rotation of the offset in double, the stored transform in float
    // (pose_trace.cpp:82-114).
    void setReferencePose(const Pose& initialInReferenceFrame)
    {
        float x0 = 0.0f, y0 = 0.0f, th0 = 0.0f;
        if (!samples_.empty()) { x0 = samples_.front().x; y0 = samples_.front().y; th0 = samples_.front().theta; }
        const double turn = initialInReferenceFrame.theta - th0;
        const double c = std::cos(turn), s = std::sin(turn);
        frame_.x = static_cast<float>(initialInReferenceFrame.x - (x0 * c - y0 * s));
        frame_.y = static_cast<float>(initialInReferenceFrame.y - (x0 * s + y0 * c));
        frame_.theta = static_cast<float>(turn);
        for (std::size_t i = 0; i < samples_.size(); ++i) samples_[i] = slam_detail::into_frame(samples_[i], frame_);
    }

    Pose getFrameTransform() const { return frame_; }
    void clear() { samples_.clear(); ascending_ = true; }

    bool empty() const { return samples_.empty(); }
    std::size_t size() const { return samples_.size(); }
    const_iterator begin() const { return samples_.begin(); }
    const_iterator end() const { return samples_.end(); }
    const Pose& operator[](int index) const { return samples_[index]; }
    const Pose& at(int index) const { return samples_.at(index); }
    const Pose& front() const { return samples_.front(); }
    const Pose& back() const { return samples_.back(); }

private:
    std::vector<Pose> samples_;
    Pose frame_;            // rotation + shift applied to every incoming sample
    bool ascending_;        // utimes never decrease along samples_

    void recheckOrder()
    {
        ascending_ = true;
        for (std::size_t i = 1; i < samples_.size() && ascending_; ++i) ascending_ = !(samples_[i].utime < samples_[i - 1].utime);
    }

    // Index i >= 1 of the FIRST pair (i-1, i) with samples_[i-1].utime <= time <= samples_[i].utime, 0 if there is none.
    // The definition, by scanning (the reference's loop, pose_trace.cpp:57-65):
    std::size_t bracketByScan(int64_t time) const
    {
        for (std::size_t i = 1; i < samples_.size(); ++i)
            if (!(time < samples_[i - 1].utime) && !(samples_[i].utime < time)) return i;
        return 0;
    }
    // Ascending stamps, front <= time <= back (checked by the caller): let i* be the first index >= 1 whose stamp is >= time.
    // Every i in [1, i*) has a stamp < time, so no pair before i* brackets the time; and the stamp at i* - 1 is <= time --
    // for i* = 1 by the caller's check, otherwise because i* - 1 is such an i.  So i* is the scan's answer, equal stamps
    // (several samples with one utime) included.  A single sample has no pair: 0, as the scan.
    std::size_t bracketByBisection(int64_t time) const
    {
        if (samples_.size() < 2) return 0;
        std::size_t lo = 1, hi = samples_.size() - 1;       // the stamp at hi (= back) is >= time
        while (lo < hi) {
            const std::size_t mid = lo + (hi - lo) / 2;
            if (samples_[mid].utime < time) lo = mid + 1; else hi = mid;
        }
        return lo;
    }
};

// ---------------------------------------------------------------- queue of scans waiting for their pose
template <class Scan>
class ScanRing {
public:
    ScanRing() : slots_(8), head_(0), count_(0) {}
    bool empty() const { return count_ == 0; }
    std::size_t size() const { return count_; }
    const Scan& front() const { return slots_[head_]; }
    void push(const Scan& s)
    {
        if (count_ == slots_.size()) grow();
        slots_[(head_ + count_) % slots_.size()] = s;
        ++count_;
    }
    // moves the oldest scan out
    void take(Scan* out)
    {
        std::swap(*out, slots_[head_]);
        head_ = (head_ + 1) % slots_.size();
        --count_;
    }
private:
    std::vector<Scan> slots_;
    std::size_t head_, count_;
    void grow()
    {
        std::vector<Scan> wider(slots_.size() * 2);
        for (std::size_t i = 0; i < count_; ++i) std::swap(wider[i], slots_[(head_ + i) % slots_.size()]);
        slots_.swap(wider);
        head_ = 0;
    }
};

// ---------------------------------------------------------------- OccupancyGridSLAM
template <class Pose, class Lidar, class Odometry, class Particle, class Particles, class GridMsg>
class OccupancyGridSLAMT {
public:
    struct Publisher {                                   // SLAM_POSE, SLAM_PARTICLES, SLAM_MAP (slam.cpp:265-268, 284-289)
        std::function<void(const Pose&)> slamPose;
        std::function<void(const Particles&)> slamParticles;
        std::function<void(const GridMsg&)> slamMap;
    };

    // Same arguments as the reference's constructor (slam.hpp:41-48) with the publisher where the lcm::LCM& stands.
    // 10 m x 10 m grid at 5 cm, 5 m mapping range (slam.cpp:22-24).  Mapping-only and a localization map exclude each other.
    OccupancyGridSLAMT(int numParticles, int8_t hitOddsIncrease, int8_t missOddsDecrease, const Publisher& pub,
                       bool waitForOptitrack, bool mappingOnlyMode = false, bool actionOnlyMode = false,
                       const std::string& localizationOnlyMap = std::string())
        : pf_(numParticles), grid_(10.0f, 10.0f, 0.05f), mapping_(5.0f, hitOddsIncrease, missOddsDecrease), out_(pub)
    {
        how_.posesGiven = mappingOnlyMode;
        how_.awaitingFrame = waitForOptitrack;
        if (!mappingOnlyMode && !localizationOnlyMap.empty()) {
            how_.mapKnown = grid_.loadFromFile(localizationOnlyMap);
            how_.odometryOnly = actionOnlyMode;
        }
        const Pose origin = slam_detail::pose_of<Pose>(0, 0.0f, 0.0f, 0.0f);
        start_ = before_ = now_ = odomAtScan_ = origin;
        scan_.utime = 0;
        scan_.num_ranges = 0;
    }

    // ---- inputs.  A scan is kept only if the pose source of this mode already reaches back to its first ray
    // (slam.cpp:90-128); the counter of dropped scans restarts once ten or more were dropped and one is kept.
    void handleLaser(const Lidar& scan)
    {
        const PoseTraceT<Pose>& src = poseSource();
        if (!src.empty() && !(scan.times.front() < src.front().utime)) {
            waiting_.push(scan);
            if (dropped_ >= kDroppedBeforeNotice) {
                std::cout << "OccupancyGridSLAM: poses are arriving, scans are being queued again\n";
                dropped_ = 0;
            }
            return;
        }
        if (++dropped_ == kDroppedBeforeNotice)
            std::cout << "OccupancyGridSLAM: dropping scans, no odometry / pose reaches back to them yet\n";
    }
    void handleOdometry(const Odometry& odometry)
    {
        odom_.addPose(slam_detail::pose_of<Pose>(odometry.utime, odometry.x, odometry.y, odometry.theta));
    }
    void handlePose(const Pose& pose) { truth_.addPose(pose); }
    // the first pose from the tracking system fixes the SLAM frame's start (slam.cpp:150-160)
    void handleOptitrack(const Pose& pose)
    {
        if (!how_.awaitingFrame) return;
        start_ = pose;
        how_.awaitingFrame = false;
    }

    // One launch for "end of updateFilter + updateMap + fetch of the next queued scan" (the default), or the reference's
    // call-by-call order (false).  Results are bit-identical; with the fused step SLAM_POSE / SLAM_PARTICLES of an
    // iteration are published after its map update has been enqueued instead of before.
    void setFusedStep(bool on) { fused_ = on; }

    // The oldest queued scan can be processed once the pose source covers the time of its first ray (slam.cpp:163-188).
    bool isReadyToUpdate() const
    {
        return !how_.awaitingFrame && !waiting_.empty() && poseSource().containsPoseAtTime(waiting_.front().times.front());
    }

    // slam.cpp:191-207: take the scan and its pose / odometry, start the poses on the first call, then localise and map --
    // unless the scan has 100 ranges or fewer.
    void runSLAMIteration()
    {
        waiting_.take(&scan_);
        const int64_t t_end = scan_.times.back();
        if (how_.posesGiven) { before_ = now_; now_ = truth_.poseAt(t_end); }
        else odomAtScan_ = odom_.poseAt(t_end);
        if (!how_.started) startPoses();
        if (scan_.num_ranges <= kFewestRanges) {
            std::cerr << "OccupancyGridSLAM: scan with only " << scan_.num_ranges << " ranges skipped\n";
            return;
        }
        const bool riding = localize();
        extendMap(riding);
    }

    // ---- inspection (tests)
    const OccupancyGrid& map() const { return grid_; }
    Pose currentPose() const { return now_; }
    int numIgnoredScans() const { return dropped_; }
    std::size_t queuedScans() const { return waiting_.size(); }
    int mapUpdateCount() const { return mapsMade_; }

private:
    enum { kDroppedBeforeNotice = 10, kFewestRanges = 100, kMapEvery = 5 };

    // What the four modes of slam.hpp:72-78 differ in:  mapping-only = posesGiven;  localization-only = mapKnown from a file;
    // action-only = that + odometryOnly;  full SLAM = none of them (mapKnown turns true with the first map update).
    struct How {
        bool posesGiven, odometryOnly, mapKnown, awaitingFrame, started;
        How() : posesGiven(false), odometryOnly(false), mapKnown(false), awaitingFrame(false), started(false) {}
    };

    How how_;
    PoseTraceT<Pose> truth_, odom_;
    ScanRing<Lidar> waiting_;
    Lidar scan_;                       // the scan of the running iteration
    Pose odomAtScan_;                  // odometry at its last ray
    Pose start_, before_, now_;        // SLAM frame start; pose estimate of the previous / this iteration
    ParticleFilterT<Pose, Lidar, Particle, Particles> pf_;
    OccupancyGrid grid_;
    MappingT<Pose, Lidar> mapping_;
    Publisher out_;
    int dropped_ = 0, mapsMade_ = 0;
    bool fused_ = true;

    const PoseTraceT<Pose>& poseSource() const { return how_.posesGiven ? truth_ : odom_; }

    // slam.cpp:232-250: both poses start at the frame's start pose, stamped with the first scan's first and last ray; the
    // filter is spread around the earlier one.
    void startPoses()
    {
        before_ = now_ = start_;
        before_.utime = scan_.times.front();
        now_.utime = scan_.times.back();
        pf_.initializeFilterAtPose(before_);
        how_.started = true;
    }

    void announce()
    {
        const Particles cloud = pf_.particles();
        if (out_.slamPose) out_.slamPose(now_);
        if (out_.slamParticles) out_.slamParticles(cloud);
    }

    // slam.cpp:253-271.  Returns true when the update's end was left to ride in the map launch (fused step).
    bool localize()
    {
        if (how_.posesGiven || !how_.mapKnown) return false;
        before_ = now_;
        if (!how_.odometryOnly && fused_) {
            pf_.updateFilterBegin(odomAtScan_, scan_, grid_);
            if (!waiting_.empty()) prefetch_scan(waiting_.front());      // the next scan is already queued: it rides along
            return true;
        }
        now_ = how_.odometryOnly ? pf_.updateFilterActionOnly(odomAtScan_) : pf_.updateFilter(odomAtScan_, scan_, grid_);
        announce();
        return false;
    }

    // slam.cpp:274-294 (its mode test is always true: the map is extended in every mode); the map goes out with every fifth update.
    void extendMap(bool riding)
    {
        if (riding) {
            mapping_.updateMapFinishingFilter(scan_, pf_, odomAtScan_.utime, grid_);
            now_ = pf_.poseEstimate();
            announce();
        } else {
            mapping_.updateMap(scan_, now_, grid_);
        }
        how_.mapKnown = true;
        if (mapsMade_++ % kMapEvery == 0 && out_.slamMap) out_.slamMap(grid_.template toLCM<GridMsg>());
    }
};

}  // namespace botlab_hip

#endif  // BOTLAB_SLAM_DRIVER_HPP
