/* botlab_hip.h -- C ABI of libbotlab_hip.so: the MI355X (gfx950) implementation of botLab's SLAM / MCL / planning
 * hot path.  Plain C, POD arguments, caller-owned buffers, int status returns (0 = ok), no exceptions, no torch types.
 *
 * Each entry point names the reference interface it stands in for (paths relative to the botLab checkout).  The C++
 * classes in include/botlab/ (OccupancyGrid, Mapping, ParticleFilter, ObstacleDistanceGrid, search_for_path) keep the
 * reference's signatures and forward to these calls; INTEGRATION.md shows the binding.
 *
 * Threading: one bl_ctx per host thread (the reference touches Mapping / ParticleFilter / map_ only from the runSLAM
 * thread, src/slam/slam_main.cpp:56-58).  All work of a ctx is stream-ordered on ONE HIP stream (its own, or a
 * caller-supplied one); calls return after enqueueing unless they hand a result back to host memory, in which case
 * they synchronise that stream first.
 */
#ifndef BOTLAB_HIP_H
#define BOTLAB_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ------------------------------------------------------------------ status */
#define BL_OK 0
#define BL_ERR_HIP 1        /* a HIP runtime call failed; text in bl_last_error() */
#define BL_ERR_ARG 2        /* bad argument (null handle, shape mismatch, out-of-range parameter) */
#define BL_ERR_CAPACITY 3   /* a device-side work list (A* open list) ran out of its configured capacity */
#define BL_ERR_STATE 4      /* call order violated (e.g. update before init) */

const char* bl_last_error(void);          /* thread-local message of the last failing call */
const char* bl_version(void);

/* ------------------------------------------------------------------ message records (in-memory layout of the
 * lcm-gen structs the reference passes around; field order from the lcmtypes .lcm files) */
typedef struct bl_pose_xyt_t {            /* lcmtypes/pose_xyt_t.lcm:1-8 ; 24 bytes */
    int64_t utime;
    float x, y, theta;
} bl_pose_xyt_t;

typedef struct bl_particle_t {            /* lcmtypes/particle_t.lcm:4-9 ; 56 bytes */
    bl_pose_xyt_t pose;
    bl_pose_xyt_t parent_pose;
    double weight;
} bl_particle_t;

typedef struct bl_lidar_t {               /* lcmtypes/lidar_t.lcm:1-14 ; arrays are HOST pointers, num_ranges long */
    int64_t utime;
    int32_t num_ranges;
    const float* ranges;
    const float* thetas;
    const int64_t* times;
    const float* intensities;             /* may be NULL (unused on the hot path) */
} bl_lidar_t;

typedef struct bl_search_params_t {       /* src/planning/astar.hpp:15-27 */
    double minDistanceToObstacle;
    double maxDistanceWithCost;
    double distanceCostExponent;
} bl_search_params_t;

/* ------------------------------------------------------------------ context */
typedef struct bl_ctx bl_ctx;

/* device: HIP device ordinal.  stream: NULL -> the ctx creates its own non-blocking stream; otherwise a hipStream_t
 * owned by the caller (e.g. torch's current stream, so collectives issued by the caller order with this ctx). */
int bl_ctx_create(int device, void* stream, bl_ctx** out);
void bl_ctx_destroy(bl_ctx* ctx);
int bl_ctx_sync(bl_ctx* ctx);
/* Per-kernel HIP-event timing on the ctx stream (bench.py's roofline leg).  kernel ids: BL_K_* below. */
int bl_ctx_timing_enable(bl_ctx* ctx, int on);   /* 0: off; 1: every kernel; else a bit mask, bit i = kernel id i */
/* Time only every `every`-th launch of each enabled kernel (default 1): an event pair costs ~13 us of stream time per
 * launch on this stack, which would distort a pipelined step. */
int bl_ctx_timing_stride(bl_ctx* ctx, int every);
int bl_ctx_timing_get(bl_ctx* ctx, int kernel_id, double* total_ms, int64_t* launches);
int bl_ctx_timing_reset(bl_ctx* ctx);
#define BL_K_MCL_MAIN 0      /* resample-gather + action + sensor model, one thread per particle */
#define BL_K_MCL_SCAN 1      /* weight prefix scan + pose estimate (3 small launches, timed together) */
#define BL_K_MAP 2           /* Mapping::updateMap */
#define BL_K_DIST 3          /* ObstacleDistanceGrid::setDistances (2 launches, timed together) */
#define BL_K_ASTAR 4         /* search_for_path */
#define BL_K_FRONTIERS 5     /* find_map_frontiers */
/* the launches inside setDistances and the replanner's snapshot copy one by one (timed only when asked for by id: "every
 * kernel" means ids 0..5; an event pair between two kernels costs a few microseconds of stream time) */
#define BL_K_DIST_ROWS 6
#define BL_K_DIST_COLS_SUMMARY 7
#define BL_K_DIST_COLS_APPLY 8
#define BL_K_SNAPSHOT 9
#define BL_K_DIST_FUSED 10   /* the whole-grid transform as one launch (grids of at least 512 x 512, width a multiple of 16) */
#define BL_K_COUNT 11

/* ------------------------------------------------------------------ OccupancyGrid  (src/slam/occupancy_grid.hpp:51-209)
 * Device-resident int8 log-odds cells, row-major y*width+x.  meters_per_cell and cells_per_meter are both carried
 * because the reference carries both (occupancy_grid.cpp:19-36 computes cpm = 1.0f/mpc; loadFromFile :138-175 does
 * not touch cpm). */
typedef struct bl_grid bl_grid;
int bl_grid_create(bl_ctx* ctx, int width, int height, float meters_per_cell, float cells_per_meter,
                   float origin_x, float origin_y, bl_grid** out);       /* cells zeroed (reset(), :48-52) */
void bl_grid_destroy(bl_grid* g);
int bl_grid_upload(bl_grid* g, const int8_t* cells);                     /* host -> device, width*height bytes */
int bl_grid_download(bl_grid* g, int8_t* cells);                         /* device -> host (synchronises) */
int bl_grid_reset(bl_grid* g);                                           /* OccupancyGrid::reset */
int bl_grid_set_frame(bl_grid* g, float meters_per_cell, float cells_per_meter, float origin_x, float origin_y);
int bl_grid_copy(bl_grid* dst, const bl_grid* src);                      /* device -> device, same shape */
void* bl_grid_device_ptr(bl_grid* g);                                    /* int8_t* in HBM */
int bl_grid_shape(const bl_grid* g, int* width, int* height);

/* ------------------------------------------------------------------ Mapping  (src/slam/mapping.hpp:25-34, mapping.cpp:8-127) */
typedef struct bl_mapping bl_mapping;
/* 0 <= hit_odds, miss_odds <= 127 */
int bl_mapping_create(bl_ctx* ctx, float max_laser_distance, int8_t hit_odds, int8_t miss_odds, bl_mapping** out);
void bl_mapping_destroy(bl_mapping* m);
/* Mapping::updateMap(scan, pose, map): first call ever latches the pose and changes no cell (initialized_). */
int bl_mapping_update(bl_mapping* m, const bl_lidar_t* scan, const bl_pose_xyt_t* pose, bl_grid* map);
/* Same, the pose read from device memory (the particle filter's estimate of this step) -- no host round trip. */
int bl_mapping_update_dev_pose(bl_mapping* m, const bl_lidar_t* scan, const void* d_pose /* bl_pose_xyt_t* */,
                               int64_t pose_utime, bl_grid* map);

/* ------------------------------------------------------------------ ParticleFilter  (src/slam/particle_filter.hpp:38-77)
 * Particles [shard_lo, shard_hi) of num_particles live on this device; the exchange record of all num_particles
 * (x, y, theta, weight-units: 16 bytes each) is replicated.  Single GPU: shard = [0, N). */
typedef struct bl_pf bl_pf;
int bl_pf_create(bl_ctx* ctx, int num_particles, int shard_lo, int shard_hi, bl_pf** out);
void bl_pf_destroy(bl_pf* pf);
/* Optional, before init: use caller-allocated device buffers for the two exchange records (each at least
 * num_particles*16 B) so a caller can run the all-gather on them in place. */
int bl_pf_set_exchange_buffers(bl_pf* pf, void* d_rec0, void* d_rec1);
void* bl_pf_exchange_rec_ptr(bl_pf* pf);     /* the record written by the last update_begin (all N; own slice filled) */
/* initializeFilterAtPose (particle_filter.cpp:16-34): N(pose, 0.01) per coordinate from a counter-based Philox stream
 * keyed by seed (reference: std::random_device), last particle = pose, weights 1/N. */
int bl_pf_init_at_pose(bl_pf* pf, const bl_pose_xyt_t* pose, uint64_t seed);
/* Replace the whole posterior from a host AoS array of num_particles records (weights must be uniform or the
 * weight-unit integers in `units` given; units == NULL -> uniform). */
int bl_pf_set_particles(bl_pf* pf, const bl_particle_t* particles, const uint32_t* units);
/* particles(): the local shard as lcm particle_t records (synchronises). */
int bl_pf_get_particles(bl_pf* pf, bl_particle_t* out_local);
/* noise source of the action model: seed for the Philox stream used when update() gets noise == NULL */
int bl_pf_set_noise_seed(bl_pf* pf, uint64_t seed);

/* updateFilter (particle_filter.cpp:37-52), single GPU or replicated-call form.
 *   rand_value: the value the reference takes from rand() for the low-variance sampler (particle_filter.cpp:92).
 *   noise: NULL -> Philox; else HOST array of 3*num_particles floats (sampledRot1, sampledTrans, sampledRot2 per
 *          output particle, global index order) -- the parity mode.
 *   out_pose may be NULL (pose stays on device, see bl_pf_pose_device_ptr). */
int bl_pf_update(bl_pf* pf, const bl_pose_xyt_t* odometry, const bl_lidar_t* scan, const bl_grid* map, int rand_value,
                 const float* noise, bl_pose_xyt_t* out_pose);
/* Sharded form: begin enqueues action + sensor model for the shard and fills its slice of the exchange record; the
 * caller then all-gathers the record (the ONLY collective of an update); end scans the weight units of all N particles
 * and forms the pose estimate from the gathered record, in an addition order that depends on N alone -- every rank, and
 * every shard count, gets the identical estimate.  *moved == 0 -> nothing was enqueued (robot did not move). */
int bl_pf_update_begin(bl_pf* pf, const bl_pose_xyt_t* odometry, const bl_lidar_t* scan, const bl_grid* map,
                       int rand_value, const float* noise, int* moved);
int bl_pf_update_end(bl_pf* pf, bl_pose_xyt_t* out_pose);
/* updateFilterActionOnly (particle_filter.cpp:54-65) */
int bl_pf_update_action_only(bl_pf* pf, const bl_pose_xyt_t* odometry, const float* noise, bl_pose_xyt_t* out_pose);
int bl_pf_pose_estimate(bl_pf* pf, bl_pose_xyt_t* out_pose);              /* poseEstimate() (synchronises) */
const void* bl_pf_pose_device_ptr(bl_pf* pf);                             /* bl_pose_xyt_t in HBM */
/* estimatePosteriorPose(posterior_) (particle_filter.cpp:144-160) of the particles as they stand (all N on this device or
 * replicated): x / y the reference's serially rounded float sums, bit for bit; becomes poseEstimate().  out_pose may be NULL. */
int bl_pf_estimate_posterior_pose(bl_pf* pf, bl_pose_xyt_t* out_pose);
/* diagnostics of the last estimate, eight values: for the x sum, then for the y sum -- sub-tiles replayed generically, phases
 * of those replays, sub-tiles stepped through by their table, gaps walked the slow way (bl_serial_sum.h, bl_mcl_finish.h) */
int bl_pf_debug_estimate_stats(bl_pf* pf, uint32_t* out8);
int bl_pf_debug_set_finish_generation(bl_pf* pf, uint32_t generation);   /* tests: the record tags of the finish launches wrap every 256 launches */
/* Resampling rule.  The update's resampler compares U_m * S with an exact integer prefix of the weight units; the reference
 * (particle_filter.cpp:84-103) compares U_m with a sequentially rounded double sum of the normalised weights.  The two agree unless
 * U_m falls within that sum's rounding error of a partial sum -- measured: never for weights an update leaves behind
 * (tests/test_gpu_resample_sweep.py, tests/test_gpu_config3_1m.py).  The one kind of weights on which they did part ways is ALL EQUAL
 * weights -- a fresh filter's, an upload's, and the set an update leaves when EVERY particle ends at the likelihood floor (a lost filter:
 * the total is then exactly 2 N units): rand() <= ~1000 or == RAND_MAX puts every U_m on a partial sum, and about half of the particles
 * took the neighbouring source.  All three are recognised (the first two by the host, the all-floor set on the device by the launch that
 * writes the total) and resampled against the reference's own cumulative, which has a closed form for equal weights (a few runs of
 * constant increment per binade: bl_mcl_finish.h, uni_seg) -- on one device and on composed shards alike, at no cost to other updates.
 * BOTLAB_NO_AUTO_STRICT=1: the integer rule there too (tests).  bl_pf_debug_uniform_runs: the number of runs in force (0: the weights
 * of the record are not known to be equal; synchronises).
 * Strict mode (off by default): EVERY finish is followed by the launches that form the reference's cumulative bit
 * for bit and the resampler searches that one: identical indices for every rand() value, at ~50 us per update at 100k particles,
 * ~140 us at 1M (three launches: the chunks' sums with the binade predicted from the integer prefix, one wave walking the chunks'
 * records with the true sum, the chunks filled in side by side) behind the finish -- also behind the map kernel that carries it
 * (bl_mapping_update_finishing_pf): a 100k-particle SLAM step is 140 us instead of 90. */
int bl_pf_debug_uniform_runs(bl_pf* pf, int* out_runs);
int bl_pf_set_strict_resampling(bl_pf* pf, int on);
/* resamplePosteriorDistribution alone (particle_filter.cpp:84-103): the source index each output particle would take for this
 * rand() value, by the very search the update kernel runs; num_particles entries (whole set on this device; synchronises) */
int bl_pf_debug_resample(bl_pf* pf, int rand_value, int32_t* out_idx);
/* diagnostics for the parity tests: resample source index and raw likelihood (half-units) of the local shard of the last
 * update; recorded only while enabled (8 B per particle of extra stores) */
int bl_pf_debug_enable(bl_pf* pf, int on);
/* The hardware measurement behind the sensor model's fast trigonometry (SensorModel::scoreRay's endpoint cells,
 * sensor_model.cpp:34-38): the largest difference between v_sin_f32 / v_cos_f32 of the unwrapped ray angle and the
 * reference's sinf / cosf(wrap_to_pi(angle)), exhaustively over every float of the angle's range; *eps_used is the bound
 * the kernel's guard band is built on (both maxima must stay below it). */
int bl_debug_trig_probe(bl_ctx* ctx, float* max_sin_err, float* max_cos_err, float* eps_used, uint64_t* floats_checked);
/* The same for the form the ray loop takes by default -- the direction of pose.theta - ray theta by the addition theorems from the
 * particle's and the ray's (cos, sin) pairs: maxima of |that - the reference's sinf / cosf of wrap_to_pi(fl(p - r))| over `pairs`
 * random (p, r), p in [-pi, pi], r in [0, 6.2831], with the functions the loop calls (Monte Carlo: the pairs are 2^48). */
int bl_debug_trig_addition_probe(bl_ctx* ctx, uint64_t pairs, uint32_t seed, float* max_sin_err, float* max_cos_err, float* eps_used,
                                 uint64_t* pairs_checked);
int bl_pf_debug_last(bl_pf* pf, int32_t* resample_idx, int32_t* likelihood_half_units);

/* ------------------------------------------------------------------ ObstacleDistanceGrid  (src/planning/obstacle_distance_grid.hpp:28-96) */
typedef struct bl_dist bl_dist;
int bl_dist_create(bl_ctx* ctx, bl_dist** out);
void bl_dist_destroy(bl_dist* d);
/* setDistances(map), obstacle_distance_grid.cpp:73-91.  When `map` is a later state of the very map `d` last transformed --
 * Mapping::updateMap calls in between, nothing else; a replanner snapshot counts as the map it was taken from -- only the window
 * those updates can influence is transformed again (grids of at least 1024 cells a side); the result is the full transform's,
 * bit for bit.  BOTLAB_DIST_NO_INCREMENTAL=1 always transforms the whole grid. */
int bl_dist_set_distances(bl_dist* d, const bl_grid* map);
int bl_dist_forget(bl_dist* d);             /* the next bl_dist_set_distances transforms the whole map, whatever d holds now */
/* diagnostic, six counts: setDistances calls that went out as an incremental launch / as a whole-grid launch / found the map
 * unchanged; and of the incremental launches those the device ended with nothing to do / a window / the whole grid */
int bl_dist_debug_stats(bl_dist* d, int64_t* out6);
/* diagnostic: the bound D (an upper bound of every finite L1 distance of the grid, in cells) the next incremental transform
 * dilates its window by, and whether it has been formed (the whole-grid kernels leave none; the first incremental transform
 * after one forms it) */
int bl_dist_debug_bound(bl_dist* d, int* formed, unsigned int* bound);
/* diagnostic, the one-launch whole-grid transform (grids of 512 x 512 .. 4096 x 4096 cells, width a multiple of 16): workgroups
 * that gave up waiting for another tile's summary (never, unless the device is broken: the next bl_dist_set_distances then
 * returns BL_ERR_STATE), and tile summaries a workgroup computed in place of one that had not started yet.
 * BOTLAB_DIST_FUSED_TEST_DELAY=<n> holds every second workgroup back at its start (tests of that path). */
int bl_dist_debug_fused(bl_dist* d, int64_t* out2);
int bl_dist_download(bl_dist* d, float* cells);                           /* width*height floats (synchronises) */
int bl_dist_shape(const bl_dist* d, int* width, int* height);
int bl_dist_frame(const bl_dist* d, float* meters_per_cell, float* cells_per_meter, float* origin_x, float* origin_y);
/* float* in HBM.  A replan never needs the floats (the search reads the integer distances), so they are only formed for callers
 * that ask: bl_dist_download / bl_dist_gather on demand, and -- once this pointer has been handed out -- with every later
 * bl_dist_set_distances on `d`, on d's stream behind the transform, so that a caller who keeps the pointer keeps reading the
 * current transform.  (NULL before the first request on a grid that has never been transformed.) */
void* bl_dist_device_ptr(bl_dist* d);

/* ------------------------------------------------------------------ search_for_path  (src/planning/astar.hpp:58-61, astar.cpp:9-274)
 * out_path[0] is always the start pose; *out_len == 1 means "no path" (lcmtypes/robot_path_t.lcm:7).  If the path is
 * longer than cap, *out_len is the full length and only cap poses are written.  stats (optional, 2 x int64): pops,
 * pushes.  open_capacity (nodes) bounds the device open list; 0 -> default. */
int bl_astar_search(bl_ctx* ctx, const bl_dist* distances, const bl_pose_xyt_t* start, const bl_pose_xyt_t* goal,
                    const bl_search_params_t* params, bl_pose_xyt_t* out_path, int cap, int* out_len, int64_t* stats);
int bl_astar_set_open_capacity(bl_ctx* ctx, int64_t nodes);
/* Which kernel the last search launched on this ctx took: 2 = k_astar2 (16-bit keys apart from payloads), 1 = k_astar (8-byte entries:
 * a cost table that can take an fCost to -32768 or below, e.g. maxDistanceWithCost > 16.4 m at the reference's d * 2000, or an odd
 * distanceCostExponent > 1), 0 = no search yet.  Diagnostics for the parity tests. */
int bl_astar_debug_last_kernel(bl_ctx* ctx);
/* Test entry for the search's open list (std::priority_queue<Node, vector, greater>, astar.cpp:75-76,117-135, as the wave-parallel
 * std::push_heap / std::pop_heap of k_astar2): replays n operations -- keys[i] in [1, 65534]: push (keys[i], pays[i]); keys[i] < 0:
 * pop -- and returns the popped (key, payload) pairs in order.  cfg 0 / 1 / 2: the storage tiers of a lone search, of a
 * co-running search, of the tests (every tier within a few thousand entries).  cycles (optional, 4 x uint64): device cycles and
 * counts of pushes and of pops. */
int bl_debug_heap2_replay(bl_ctx* ctx, const int32_t* keys, const uint32_t* pays, int n, int cfg, int64_t capacity,
                          uint32_t* out_keys, uint32_t* out_pays, int* out_n, uint64_t* cycles);
/* Asynchronous form for step pipelines: enqueue the search, fetch the result later.  Up to 4 searches may be in flight;
 * results are fetched in launch order and fetching waits for that search only (work enqueued after it keeps running). */
int bl_astar_search_async(bl_ctx* ctx, const bl_dist* distances, const bl_pose_xyt_t* start, const bl_pose_xyt_t* goal,
                          const bl_search_params_t* params);
/* Same, the start pose read from device memory (e.g. bl_pf_pose_device_ptr: the estimate of this very step) so a
 * step pipeline needs no host round trip between localisation and replanning; out_path[0] of the result is that pose. */
int bl_astar_search_async_dev_start(bl_ctx* ctx, const bl_dist* distances, const void* d_start /* bl_pose_xyt_t* */,
                                    const bl_pose_xyt_t* goal, const bl_search_params_t* params);
int bl_astar_search_result(bl_ctx* ctx, bl_pose_xyt_t* out_path, int cap, int* out_len, int64_t* stats);

/* ------------------------------------------------------------------ asynchronous replanner
 * The reference's planner is a separate process fed by the maps/poses the SLAM process publishes
 * (src/planning/exploration.cpp:300-317).  bl_planner is the same arrangement on one device: submit() snapshots the
 * map and the (device-resident) pose on the SLAM ctx's stream and runs setDistances + search_for_path on a second
 * stream, overlapping the next scan's particle filter; fetch() returns results in submission order (up to 2 per lane in
 * flight; submit blocks the SLAM stream, not the host, while both snapshot slots of the lane are still being read). */
typedef struct bl_planner bl_planner;
/* lanes (1..4): consecutive submissions go to consecutive side streams, so up to `lanes` replans run concurrently (each
 * is one wavefront on its own CU and latency-bound; independent searches are what the GPU can overlap). */
int bl_planner_create(bl_ctx* ctx, int lanes, bl_planner** out);
/* batch (1..64): a lane collects `batch` consecutive submissions and issues their searches as ONE launch, a workgroup each,
 * so lanes x batch replans overlap although the runtime multiplexes streams onto four hardware queues.  For grids where a
 * search outlasts a step (2000x2000: ~1.5 ms against 0.2 ms); a result is then available `batch` - 1 submissions later
 * (a fetch that cannot wait for the batch to fill sends it off as it is).  bl_planner_create is batch = 1. */
int bl_planner_create_batched(bl_ctx* ctx, int lanes, int batch, bl_planner** out);
void bl_planner_destroy(bl_planner* p);
int bl_planner_submit(bl_planner* p, const bl_grid* map, const void* d_start_pose /* bl_pose_xyt_t* on the device */,
                      const bl_pose_xyt_t* goal, const bl_search_params_t* params);
int bl_planner_fetch(bl_planner* p, bl_pose_xyt_t* out_path, int cap, int* out_len, int64_t* stats);
/* End of input (the scan stream pauses, a run ends): sends off the batch every lane is still collecting, so that its work runs beside
 * whatever the SLAM stream still holds instead of behind the fetch that would have sent it.  Results are fetched as ever. */
int bl_planner_flush(bl_planner* p);
/* on: -1 = just read; 0/1 = disable/enable(+reset) HIP-event timing of the planner stream's kernels (totals in ms) */
int bl_planner_timing(bl_planner* p, int on, double* dist_ms, double* astar_ms, int64_t* launches);
/* bl_mapping_update_dev_pose followed by bl_planner_submit(p, map, d_pose, goal, params) as one call: on grids up to
 * 256 K cells the map kernel itself leaves the snapshot behind (one dependent launch less on the SLAM stream). */
int bl_planner_submit_with_map_update(bl_planner* p, bl_mapping* m, const bl_lidar_t* scan, const void* d_pose,
                                      int64_t pose_utime, bl_grid* map, const bl_pose_xyt_t* goal,
                                      const bl_search_params_t* params);
/* updateLocalization + updateMap of one runSLAMIteration (src/slam/slam.cpp:191-207, 262, 279) with the END of the filter
 * update folded into the map kernel: `pf` has an update begun with bl_pf_update_begin, and this call is its
 * bl_pf_update_end(pf, NULL) followed by bl_mapping_update_dev_pose(m, scan, bl_pf_pose_device_ptr(pf), pose_utime, map)
 * -- in ONE launch.  The map kernel's own workgroup forms the pose estimate right before Mapping::updateMap reads it and
 * further workgroups of the launch write the weight prefix meanwhile, so the SLAM stream carries one kernel less per step.
 * A filter with nothing pending (the robot did not move) or whose end cannot ride (sharded particle set) is ended the
 * ordinary way first.  Results are bit-identical to the separate calls. */
/* ------------------------------------------------------------------ particle shards over RCCL  
 * The shards' one collective -- the in-place all-gather of the 16-byte exchange record -- enqueued from this library on the
 * ctx stream, between the two halves of bl_pf_update (bl_pf_update_begin / bl_pf_update_end).  RCCL is not linked: the
 * caller names the librccl.so its process already uses (with PyTorch: torch/lib/librccl.so).  Rendezvous is the caller's:
 * rank 0 makes the id with bl_comm_unique_id, every rank receives its 128 bytes by whatever transport the host has
 * (torch.distributed in botlab_amd/sharded.py) and calls bl_comm_create (collective). */
typedef struct bl_comm bl_comm;
int bl_comm_load(const char* rccl_path);          /* dlopen + symbols only: 0 if the library is usable */
int bl_comm_unique_id(const char* rccl_path, char* out_id128);
int bl_comm_create(bl_ctx* ctx, const char* rccl_path, const char* id128, int rank, int world, bl_comm** out);
void bl_comm_destroy(bl_comm* c);
/* rec: world x per_rank_floats floats, this rank's slice already at its offset (bl_pf_exchange_rec_ptr) */
int bl_comm_all_gather_inplace(bl_comm* c, void* rec, size_t per_rank_floats);

/* Composed finish of a sharded particle set: instead of all-gathering the whole record (N x 16 B into every
 * rank) each rank keeps its own block, reads the resampling sources it needs from their owners' memory, and the end of an
 * update exchanges two SMALL all-gathers -- tile sums (40 B per 512 particles), then sub-tile records + tables (32 B per 128
 * particles + 80 KB) -- before every rank runs the (replicated, ~10 us) chain of estimatePosteriorPose.  Results are the single
 * rank's, bit for bit.  Set-up, once the filter holds particles: bl_pf_shard_setup (block = particles per rank, a multiple of
 * 2048 -- of 512 for fewer than 160 000 particles --, shard = [rank * block, min(N, (rank + 1) * block)) as given to bl_pf_create); hand every rank's three arrays to every
 * rank (bl_pf_shard_local_ptrs -> bl_ipc_export -> the host's transport -> bl_ipc_open -> bl_pf_shard_set_peer; a rank of the
 * same process passes the pointers themselves); bl_pf_shard_commit.  Per update: bl_pf_update_begin, bl_pf_shard_exchange (or
 * bl_pf_shard_stage(1), all-gather of the sums buffer, bl_pf_shard_stage(2), all-gather of the exchange buffer, both in place),
 * then bl_pf_update_end or one of the *_finishing_pf calls.  The all-gathers are what orders a rank's kernels against the other
 * ranks' reads of its memory: every rank must run them, on the filter's stream. */
int bl_dev_enable_peer_access(int device, int peer_device);   /* one process driving several devices (include/botlab/sharded_filter.hpp): kernels of `device` may use `peer_device`'s pointers */
int bl_dev_alloc(bl_ctx* ctx, size_t bytes, void** out);     /* plain zeroed device memory (the probe of botlab_amd/sharded.py) */
int bl_dev_free(void* dev_ptr);
int bl_dev_word(bl_ctx* ctx, void* dev_ptr, int write, uint32_t* value);   /* one word written / read by a kernel of ctx's device (the probe: a peer mapping must be readable by kernels) */
int bl_ipc_export(const void* dev_ptr, char* out_handle64);
int bl_ipc_open(const char* handle64, void** out_dev_ptr);
int bl_ipc_close(void* dev_ptr);
int bl_pf_shard_setup(bl_pf* pf, int rank, int world, int block);
int bl_pf_shard_local_ptrs(bl_pf* pf, void** rec0, void** rec1, void** prefix);
int bl_pf_shard_set_peer(bl_pf* pf, int rank, const void* rec0, const void* rec1, const void* prefix);
int bl_pf_shard_commit(bl_pf* pf);
int bl_pf_shard_buffers(bl_pf* pf, void** sums, size_t* sums_bytes_per_rank, void** xchg, size_t* xchg_bytes_per_rank);
int bl_pf_shard_stage(bl_pf* pf, int stage);
int bl_pf_shard_exchange(bl_pf* pf, bl_comm* c);
/* bytes per rank and update: sent into the two all-gathers, received from them, and the rank's own block of records (about what
 * its k_mcl_main reads of source records, from wherever they lie) */
int bl_pf_shard_traffic(bl_pf* pf, int64_t* out3);
/* The same exchange WITHOUT a collective (peer-store form): every rank copies its slice of the tile sums, then its exchange
 * block, straight into every other rank's buffers (mapped like the records: bl_pf_shard_local_ptrs_peer -> bl_ipc_export -> ...
 * -> bl_ipc_open -> bl_pf_shard_set_peer_buffers for every rank, after bl_pf_shard_commit; then bl_pf_shard_peer_commit), raises a
 * per-source counter there, and waits for the other ranks' counters in front of the launches that consume the data: two ~40 us
 * collective latencies per update become two pushes over xGMI.  Layouts and results are the collective form's.  First contact:
 * bl_pf_shard_peer_selftest pushes a pattern to every rank and checks every rank's pattern (device-side spin limit: it cannot
 * hang); the ranks agree on the outcome over their rendezvous and either keep the form (bl_pf_shard_peer_reset(pf, 1)) or all
 * leave it (..., 0) for the collective forms.  Per update: bl_pf_update_begin, bl_pf_shard_exchange_peer, then as above.
 * A rank that waits for another rank's part longer than the cross-rank limit (30 s on the device's real-time clock;
 * BOTLAB_SHARD_WAIT_MS for tests) gives up FOR GOOD: a sticky flag on the device turns the update's groups, finish and map store
 * and every later update of the set into no-ops, and every call that fetches the pose or the particles (bl_pf_update_end,
 * bl_map_update_finishing_pf + bl_pf_pose_estimate, bl_pf_get_particles) and every later bl_pf_update_begin returns
 * BL_ERR_STATE until bl_pf_shard_setup or bl_pf_init_at_pose: nothing is computed from another rank's stale data. */
int bl_pf_shard_local_ptrs_peer(bl_pf* pf, void** sums, void** xchg, void** flags);
int bl_pf_shard_set_peer_buffers(bl_pf* pf, int rank, void* sums, void* xchg, void* flags);
int bl_pf_shard_peer_commit(bl_pf* pf);
int bl_pf_shard_peer_active(const bl_pf* pf);
int bl_pf_shard_peer_selftest(bl_pf* pf, int* ok);
int bl_pf_shard_peer_reset(bl_pf* pf, int keep);
int bl_pf_shard_exchange_peer(bl_pf* pf);
int bl_pf_shard_exchange_peer_phase(bl_pf* pf, int phase);   /* 0 sums + push, 1 wait + groups + push, 2 wait: one process driving several ranks enqueues every rank's phase p before any rank's p + 1 */

/* The NEXT lidar scan handed over early (a SLAM host has it queued, src/slam/slam.cpp:96-104): it is packed into pinned
 * memory now and copied to the device by the next bl_mapping_update* / bl_planner_submit_with_map_update* launch of this
 * ctx, beside that kernel's own work; the bl_pf_update* / bl_mapping_update* call that later brings the same scan then
 * launches no fetch kernel.  Purely an optimisation: a scan that is not the next one used costs only its packing. */
int bl_scan_prefetch(bl_ctx* ctx, const bl_lidar_t* scan);
int bl_mapping_update_finishing_pf(bl_mapping* m, const bl_lidar_t* scan, bl_pf* pf, int64_t pose_utime, bl_grid* map);
/* the same for bl_planner_submit_with_map_update */
int bl_planner_submit_with_map_update_finishing_pf(bl_planner* p, bl_mapping* m, const bl_lidar_t* scan, bl_pf* pf,
                                                   int64_t pose_utime, bl_grid* map, const bl_pose_xyt_t* goal,
                                                   const bl_search_params_t* params);

/* ------------------------------------------------------------------ batched searches, frontiers  
 * n independent search_for_path calls (astar.hpp:58-61) from ONE start on one distance grid, run concurrently (one
 * wavefront each).  Path i is written to out_paths + i*cap_each (at most cap_each poses; out_lens[i] is the true
 * length, 1 = no path); stats, if given, receives {pops, pushes} per search. */
int bl_astar_search_batch(bl_ctx* ctx, const bl_dist* d, const bl_pose_xyt_t* start, const bl_pose_xyt_t* goals, int n,
                          const bl_search_params_t* params, bl_pose_xyt_t* out_paths, int cap_each, int* out_lens,
                          int64_t* stats);
/* distances_(x, y) (obstacle_distance_grid.hpp:63) for n cells given as x0,y0,x1,y1,... in one round trip; a cell
 * outside the grid yields NaN. */
int bl_dist_gather(bl_dist* d, const int32_t* xy_cells, int n, float* out);

/* find_map_frontiers (src/planning/frontiers.hpp:34-36, frontiers.cpp:25-85): the frontiers reachable through free
 * space from the robot cell, in the reference's discovery order, each frontier's cells in its growth order; frontier k
 * = cells offsets[k] .. offsets[k+1] of xy (global x, y per cell). */
typedef struct bl_frontiers bl_frontiers;
int bl_frontiers_find(bl_ctx* ctx, const bl_grid* map, const bl_pose_xyt_t* robot_pose, double min_frontier_length,
                      bl_frontiers** out);
int bl_frontiers_from_host(const int32_t* offsets, int count, const float* xy, bl_frontiers** out);   /* caller-made std::vector<frontier_t> */
int bl_frontiers_count(const bl_frontiers* f);
int bl_frontiers_total_cells(const bl_frontiers* f);
int bl_frontiers_get(const bl_frontiers* f, int32_t* offsets /* count + 1 */, float* xy /* 2 * total_cells */);
int bl_frontiers_stats(const bl_frontiers* f, int* bfs_cells, int* bfs_levels);   /* free-space flood size / depth (diagnostic) */
/* which kernels grew the frontiers of this result (diagnostic, for tests of the fall-backs): 0 the one-workgroup form of grids up to
 * 96 K cells (or a caller-made list), 1 the one-workgroup sweep of larger grids, 2 k_frontier_grow (visited set in LDS, classes from
 * global memory), 3 k_frontier_grow2 (all frontier-class cells of the grid in one LDS set: up to 16 384 of them) */
int bl_frontiers_debug_sweep_kernel(const bl_frontiers* f);
void bl_frontiers_destroy(bl_frontiers* f);

/* ------------------------------------------------------------------ simulator lidar  
 * Lidar._beam_scan (src/sim/lidar.py:106-138) for n beams on a truth world (cells > 0 are occupied, Map.at_xy's index
 * arithmetic without bounds checks included, src/sim/map.py:80-87): beam i starts at (x[i], y[i]) and points along
 * angle[i] (the clamped pose.theta - theta); out_ranges[i] is the marched distance or max_distance.  World origin and
 * resolution are doubles, as the simulator reads them from the .map header. */
int bl_sim_cast_beams(bl_ctx* ctx, const bl_grid* world, double origin_x, double origin_y, double meters_per_cell,
                      const double* x, const double* y, const double* angle, int n, double max_distance, double* out_ranges);

/* ------------------------------------------------------------------ LCM wire codec  
 * The seven message types on the hot path's boundary (the .lcm files under lcmtypes/), LCM 1.4.0 wire format: 8-byte fingerprint, then
 * the members in declaration order, scalars big-endian.  Encoders return the encoded size (buf == NULL: size query) or a
 * negative status; decoders fill caller-owned arrays (capacities in elements) and report the message's counts.  Host
 * code (usable without a GPU) except bl_pf_encode_particles_lcm / bl_grid_encode_lcm, which produce the bytes from
 * device state.  PARITY UNPINNED: no LCM build or LCM-encoded data exists in the reference checkout. */
#define BL_LCM_POSE_XYT 0
#define BL_LCM_ODOMETRY 1
#define BL_LCM_LIDAR 2
#define BL_LCM_PARTICLE 3
#define BL_LCM_PARTICLES 4
#define BL_LCM_OCCUPANCY_GRID 5
#define BL_LCM_ROBOT_PATH 6
#define BL_LCM_TYPE_COUNT 7
uint64_t bl_lcm_fingerprint(int type);
int64_t bl_lcm_encode_pose(int type /* BL_LCM_POSE_XYT or BL_LCM_ODOMETRY */, const bl_pose_xyt_t* pose, uint8_t* buf, int64_t cap);
int64_t bl_lcm_encode_lidar(const bl_lidar_t* scan, const float* intensities /* NULL: zeros */, uint8_t* buf, int64_t cap);
int64_t bl_lcm_encode_particles(int64_t utime, const bl_particle_t* particles, int32_t n, uint8_t* buf, int64_t cap);
int64_t bl_lcm_encode_grid(int64_t utime, float origin_x, float origin_y, float meters_per_cell, int32_t width, int32_t height,
                           const int8_t* cells, uint8_t* buf, int64_t cap);
int64_t bl_lcm_encode_path(int64_t utime, const bl_pose_xyt_t* path, int32_t n, uint8_t* buf, int64_t cap);
int bl_lcm_decode_pose(int type, const uint8_t* buf, int64_t len, bl_pose_xyt_t* out);
int bl_lcm_decode_lidar(const uint8_t* buf, int64_t len, int64_t* utime, int32_t* n, float* ranges, float* thetas, int64_t* times,
                        float* intensities, int32_t cap);
int bl_lcm_decode_particles(const uint8_t* buf, int64_t len, int64_t* utime, int32_t* n, bl_particle_t* out, int32_t cap);
int bl_lcm_decode_grid(const uint8_t* buf, int64_t len, int64_t* utime, float* origin_xy_mpc /* 3 */,
                       int32_t* width_height_ncells /* 3 */, int8_t* cells, int64_t cap);
int bl_lcm_decode_path(const uint8_t* buf, int64_t len, int64_t* utime, int32_t* n, bl_pose_xyt_t* path, int32_t cap);
/* lcm-logger files: one event = sync 0xEDA1DA01, event number, timestamp (us), channel length, data length, channel, data */
int64_t bl_lcm_log_event_size(int32_t channel_len, int32_t data_len);
int64_t bl_lcm_log_write_event(int64_t event_number, int64_t timestamp_us, const char* channel, const uint8_t* data, int32_t data_len,
                               uint8_t* buf, int64_t cap);
/* returns the event's total size, 0 if buf[0..len) does not hold a whole event yet, < 0 if there is no event at buf */
int64_t bl_lcm_log_read_event(const uint8_t* buf, int64_t len, int64_t* event_number, int64_t* timestamp_us, int64_t* channel_off,
                              int32_t* channel_len, int64_t* data_off, int32_t* data_len);
/* particles() + encode (slam.cpp:265-268) and toLCM() + encode (slam.cpp:285-289) from device state, one D2H into buf */
int64_t bl_pf_encode_particles_lcm(bl_pf* pf, int64_t utime, uint8_t* buf, int64_t cap);
int64_t bl_grid_encode_lcm(bl_grid* grid, int64_t utime, uint8_t* buf, int64_t cap);

/* The MotionPlanner members plan_path_to_frontier reads (motion_planner.hpp:153-165). */
typedef struct {
    double robot_radius;               /* params_.robotRadius */
    bl_search_params_t search;         /* searchParams_ */
    int32_t num_frontiers;             /* setNumFrontiers() */
    bl_pose_xyt_t prev_goal;           /* setPrevGoal() */
} bl_motion_planner_t;
/* plan_path_to_frontier (frontiers.hpp:50-53, frontiers.cpp:104-214): closest frontier, its middle cell, then the
 * expanding-square sweep of candidate goals -- every ring's planPath calls run as one batch of searches.  An empty
 * frontier list gives *out_len = 0 (the reference's empty path); a sweep that never finds a goal gives the 1-pose path
 * (DESIGN.md D8).  stats, if given: {pops, pushes, searches run}. */
int bl_plan_path_to_frontier(bl_ctx* ctx, const bl_frontiers* frontiers, const bl_pose_xyt_t* robot_pose, bl_dist* dist,
                             const bl_motion_planner_t* planner, bl_pose_xyt_t* out_path, int cap, int* out_len,
                             bl_pose_xyt_t* chosen_goal, int64_t* stats);


/* ------------------------------------------------------------------ the exploration step, asynchronously  (src/planning/exploration.cpp:277-369)
 * Exploration::executeExploringMap on every published map: planner_.setMap, find_map_frontiers, and -- when the robot is within
 * 0.5 m of currentTarget_ or has none -- plan_path_to_frontier; then the status / next-state rule (:332-368; D10).  A submission
 * snapshots the map and the device-resident pose on ctx's stream; one of `lanes` side streams runs the distance transform and the
 * frontier search against the snapshot; bl_explorer_fetch hands back the steps in submission order and applies the rule with the
 * state consecutive steps share (currentTarget_, currentPath_), running plan_path_to_frontier on that lane when it is due.  At
 * most `lanes` submissions may be pending.  bl_explorer_submit / bl_explorer_pending on one thread and bl_explorer_fetch on
 * another (the reference's exploration PROCESS beside its SLAM process) may run concurrently; bl_explorer_pending counts the
 * submission a fetch is still working on. */
typedef struct bl_explorer bl_explorer;
typedef struct {
    int32_t next_state;      /* exploration_status_t: 1 EXPLORING_MAP, 2 RETURNING_HOME, 4 FAILED_EXPLORATION */
    int32_t status;          /* 0 IN_PROGRESS, 1 COMPLETE, 2 FAILED */
    int32_t num_frontiers;   /* frontiers_.size() */
    int32_t frontier_cells;
    int32_t planned;         /* 1: plan_path_to_frontier ran in this step */
    int32_t path_length;     /* currentPath_.path_length after the step */
    int64_t pops, pushes, searches;   /* of this step's plan_path_to_frontier */
    int32_t bfs_cells, bfs_levels;    /* free cells the frontier search flooded, and its depth */
    bl_pose_xyt_t pose;      /* currentPose_ of the step (the snapshot's) */
    bl_pose_xyt_t target;    /* currentTarget_ after the step */
    float frontiers_ms;      /* device time of find_map_frontiers' kernels */
    float plan_ms;           /* host wall time of plan_path_to_frontier (0 when it did not run) */
} bl_explore_result_t;
int bl_explorer_create(bl_ctx* ctx, int lanes, double robot_radius, bl_explorer** out);  /* lanes 1..16; MotionPlannerParams::robotRadius (a double: motion_planner.hpp:27-35) */
void bl_explorer_destroy(bl_explorer* e);
int bl_explorer_set_state(bl_explorer* e, const bl_pose_xyt_t* target, const bl_pose_xyt_t* prev_goal);   /* currentTarget_ / setPrevGoal; null: unchanged */
int bl_explorer_submit(bl_explorer* e, const bl_grid* map, const void* d_pose);          /* d_pose: bl_pose_xyt_t in HBM (e.g. bl_pf_pose_device_ptr) */
int bl_explorer_pending(const bl_explorer* e);
int bl_explorer_fetch(bl_explorer* e, bl_explore_result_t* out, bl_pose_xyt_t* out_path, int cap);   /* out_path: currentPath_, up to cap poses */
int bl_explorer_frontiers(const bl_explorer* e, bl_frontiers** out);                     /* frontiers_ of the last fetched step (caller destroys) */

#ifdef __cplusplus
}
#endif
#endif /* BOTLAB_HIP_H */
