// sharded_filter.hpp -- a C++ host for ParticleFilter sharded over the GPUs of one node from ONE process (no Python, no
// torch): ranks[r] is a ctx + filter + replicated map on device devices[r]; the per-update exchange of the composed finish runs
// as peer stores between the ranks' buffers (form 3; bl_pf_shard_* in botlab_hip.h).  What is sharded:
// particle_filter.cpp:84-160 (resample, proposal, normalise, estimate); Mapping::updateMap runs replicated on every rank and
// carries the end of the filter update (bl_mapping_update_finishing_pf), so every rank owns the identical estimate and map.
//
// One process addresses every device, so the ranks hand each other their device pointers directly (the same-process path of
// the ABI) after peer access between the devices has been enabled; the one-process-per-GPU arrangement (bench.py,
// botlab_amd/sharded.py) exchanges IPC handles instead and is otherwise the same sequence of calls.  Every rank's exchange phase
// p is enqueued before any rank's phase p + 1: streams that share a hardware queue run in submission order, so a wait is never
// enqueued in front of the push it waits for.
#ifndef BOTLAB_SHARDED_FILTER_HPP
#define BOTLAB_SHARDED_FILTER_HPP

#include <cstdio>
#include <cstdlib>
#include <vector>

#include "../botlab_hip.h"

namespace botlab_hip {

class ShardedFilterGroup {
public:
    struct Rank { bl_ctx* ctx = nullptr; bl_pf* pf = nullptr; bl_grid* grid = nullptr; bl_mapping* mapping = nullptr; int lo = 0, hi = 0; };

    // devices[r]: the HIP device of rank r (all equal: every rank on one device, the test arrangement)
    ShardedFilterGroup(int numParticles, const std::vector<int>& devices, int width, int height, float metersPerCell, float cellsPerMeter,
                       float originX, float originY, const int8_t* cells, float maxLaserDistance = 5.0f, int8_t hitOdds = 4, int8_t missOdds = 1)
        : n_(numParticles), world_((int)devices.size())
    {
        const int align = numParticles >= 160000 ? 2048 : 512;           // whole finish groups and scan tiles (bl_pf_shard_setup)
        block_ = (numParticles + world_ - 1) / world_;
        block_ = (block_ + align - 1) / align * align;
        if (world_ < 2 || world_ > 8 || (long long)(world_ - 1) * block_ >= numParticles) die("particle count and rank count do not give every rank a block");
        ranks_.resize((size_t)world_);
        for (int a = 0; a < world_; ++a)
            for (int b = 0; b < world_; ++b) ok(bl_dev_enable_peer_access(devices[(size_t)a], devices[(size_t)b]));
        for (int r = 0; r < world_; ++r) {
            Rank& k = ranks_[(size_t)r];
            k.lo = r * block_; k.hi = numParticles < (r + 1) * block_ ? numParticles : (r + 1) * block_;
            ok(bl_ctx_create(devices[(size_t)r], nullptr, &k.ctx));
            ok(bl_grid_create(k.ctx, width, height, metersPerCell, cellsPerMeter, originX, originY, &k.grid));
            ok(bl_grid_upload(k.grid, cells));
            ok(bl_mapping_create(k.ctx, maxLaserDistance, hitOdds, missOdds, &k.mapping));
            ok(bl_pf_create(k.ctx, numParticles, k.lo, k.hi, &k.pf));
        }
    }
    ~ShardedFilterGroup()
    {
        for (Rank& k : ranks_) {
            if (k.ctx) (void)bl_ctx_sync(k.ctx);
        }
        for (Rank& k : ranks_) {
            if (k.pf) bl_pf_destroy(k.pf);
            if (k.mapping) bl_mapping_destroy(k.mapping);
            if (k.grid) bl_grid_destroy(k.grid);
            if (k.ctx) bl_ctx_destroy(k.ctx);
        }
    }
    ShardedFilterGroup(const ShardedFilterGroup&) = delete;
    ShardedFilterGroup& operator=(const ShardedFilterGroup&) = delete;

    // ParticleFilter::initializeFilterAtPose on every rank (counter-based: each generates the identical set), then the shards
    // are tied together: every rank's records / prefix and tile sums / exchange block / counters to every rank.
    void initializeFilterAtPose(const bl_pose_xyt_t& pose, uint64_t seed)
    {
        for (Rank& k : ranks_) ok(bl_pf_init_at_pose(k.pf, &pose, seed));
        for (int r = 0; r < world_; ++r) ok(bl_pf_shard_setup(ranks_[(size_t)r].pf, r, world_, block_));
        std::vector<void*> a((size_t)world_), b((size_t)world_), c((size_t)world_);
        for (int r = 0; r < world_; ++r) ok(bl_pf_shard_local_ptrs(ranks_[(size_t)r].pf, &a[(size_t)r], &b[(size_t)r], &c[(size_t)r]));
        for (Rank& k : ranks_) {
            for (int r = 0; r < world_; ++r) ok(bl_pf_shard_set_peer(k.pf, r, a[(size_t)r], b[(size_t)r], c[(size_t)r]));
            ok(bl_pf_shard_commit(k.pf));
        }
        for (int r = 0; r < world_; ++r) ok(bl_pf_shard_local_ptrs_peer(ranks_[(size_t)r].pf, &a[(size_t)r], &b[(size_t)r], &c[(size_t)r]));
        for (Rank& k : ranks_) {
            for (int r = 0; r < world_; ++r) ok(bl_pf_shard_set_peer_buffers(k.pf, r, a[(size_t)r], b[(size_t)r], c[(size_t)r]));
            ok(bl_pf_shard_peer_commit(k.pf));
        }
    }

    // One SLAM step on every rank: updateFilter(odometry, scan, map) sharded, then updateMap(scan, estimate, map) replicated, the
    // filter's end riding in the map kernel.  Nothing waits; poseEstimate() synchronises.
    void step(const bl_pose_xyt_t& odometry, const bl_lidar_t& scan, int randValue)
    {
        int moved_all = -1;
        for (Rank& k : ranks_) {
            int moved = 0;
            ok(bl_pf_update_begin(k.pf, &odometry, &scan, k.grid, randValue, nullptr, &moved));
            if (moved_all >= 0 && moved != moved_all) die("ranks disagree on whether the robot moved");
            moved_all = moved;
        }
        if (moved_all)
            for (int phase = 0; phase < 3; ++phase)
                for (Rank& k : ranks_) ok(bl_pf_shard_exchange_peer_phase(k.pf, phase));
        for (Rank& k : ranks_) ok(bl_mapping_update_finishing_pf(k.mapping, &scan, k.pf, scan.utime, k.grid));
    }

    bl_pose_xyt_t poseEstimate(int rank = 0)
    {
        bl_pose_xyt_t p;
        ok(bl_pf_pose_estimate(ranks_[(size_t)rank].pf, &p));
        return p;
    }
    // the whole particle set, rank after rank (particles() of every shard)
    std::vector<bl_particle_t> particles()
    {
        std::vector<bl_particle_t> all((size_t)n_);
        for (Rank& k : ranks_) ok(bl_pf_get_particles(k.pf, all.data() + k.lo));
        return all;
    }
    std::vector<int8_t> mapCells(int rank, int width, int height)
    {
        std::vector<int8_t> c((size_t)width * height);
        ok(bl_grid_download(ranks_[(size_t)rank].grid, c.data()));
        return c;
    }
    int world() const { return world_; }
    const Rank& rank(int r) const { return ranks_[(size_t)r]; }

private:
    static void ok(int rc) { if (rc != BL_OK) die(bl_last_error()); }
    static void die(const char* why) { std::fprintf(stderr, "botlab_hip::ShardedFilterGroup: %s\n", why); std::abort(); }
    int n_, world_, block_ = 0;
    std::vector<Rank> ranks_;
};

}  // namespace botlab_hip
#endif
