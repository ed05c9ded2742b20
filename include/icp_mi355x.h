/*
 * icp_mi355x.h — C ABI of the MI355X-native (gfx950, HIP) frame-to-model ICP odometry hot path.
 *
 * Drop-in boundary for pyLiDAR-SLAM's `ICPFrameToModel` and the inner plugin seams it is built from.
 * Every entry point cites the reference interface it replaces (paths relative to the reference repo root).
 * Plain pointers and sizes only: no torch / numpy types cross this boundary.  All matrices are row-major 4x4 float.
 *
 * Pointers tagged `mem` may live on the host (ICP_MEM_HOST: the library stages them through HBM) or on the device
 * (ICP_MEM_DEVICE: e.g. `tensor.data_ptr()` of a torch-ROCm tensor; zero-copy).  All work is enqueued on the context's
 * stream (`icp_set_stream`, default stream otherwise); entry points that return values to the host synchronise that
 * stream before returning, the others are asynchronous.
 *
 * Threading: one context per odometry instance, calls on one context must be serialised by the caller (this is the
 * reference's contract too: a single-threaded frame loop, slam/odometry/odometry_runner.py:170-182).
 *
 * Return value: ICP_OK (0) or a negative icp_status; `icp_last_error(ctx)` gives the message.
 */
#ifndef ICP_MI355X_H
#define ICP_MI355X_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct icp_ctx icp_ctx;

typedef enum {
    ICP_OK = 0,
    ICP_ERR_INVALID_ARGUMENT = -1, /* maps to the reference's AssertionError (slam/common/utils.py:30-38) */
    ICP_ERR_HIP = -2,              /* HIP runtime failure (message in icp_last_error) */
    ICP_ERR_INVALID_JACOBIAN = -3, /* RuntimeError("Invalid Jacobian in Gauss Newton minimization"),
                                      slam/common/optimization.py:334-336 (|det H| < 1e-7) */
    ICP_ERR_EMPTY_MAP = -4,        /* nearest-neighbour search against an empty local map */
    ICP_ERR_NO_DEVICE = -5,        /* no gfx950 device visible: the product path never falls back to the CPU */
    ICP_ERR_EXCHANGE = -6          /* multi-GPU exchange: a peer did not deliver its normal equations in time */
} icp_status;

typedef enum { ICP_MEM_HOST = 0, ICP_MEM_DEVICE = 1 } icp_mem;

/* Robust weighting schemes of slam/common/optimization.py:210-226 (`_LS_SCHEME`). */
typedef enum {
    ICP_SCHEME_LEAST_SQUARE = 0, /* "default" / "least_square" :66-72 */
    ICP_SCHEME_HUBER = 1,        /* :76-97  */
    ICP_SCHEME_EXP = 2,          /* :101-117 */
    ICP_SCHEME_NEIGHBORHOOD = 3, /* :121-145 */
    ICP_SCHEME_GEMAN_MCCLURE = 4,        /* :149-166 */
    ICP_SCHEME_SQUARE_GEMAN_MCCLURE = 5, /* :170-187 */
    ICP_SCHEME_CAUCHY = 6                /* :191-208 */
} icp_scheme;

/* Target-point masking of `ICPFrameToModel.sample_points` (slam/odometry/icp_odometry.py:301-308):
 * rows containing a NaN are always skipped (remove_nan, :357); ICP_TARGETS_SKIP_NULL additionally skips zero-norm rows
 * (the "non-null pixels of the vertex map" case, :303-305).  Skipped rows contribute nothing, exactly as if removed. */
typedef enum { ICP_TARGETS_ALL = 0, ICP_TARGETS_SKIP_NULL = 1 } icp_target_mode;

/* Cost minimised by every alignment of the registration loop: RIGID_ALIGNMENT `mode` of the reference
 * (slam/odometry/alignment.py:200-208, selected at slam/odometry/icp_odometry.py:98). */
typedef enum {
    ICP_COST_POINT_TO_PLANE = 0, /* point_to_plane_gauss_newton: GaussNewtonPointToPlaneAlignment :80-127 */
    ICP_COST_POINT_TO_POINT = 1  /* point_to_point_gauss_newton: GaussNewtonPointToPointAlignment :143-189, one step
                                    of PointToPointCost linearised at x0 = 0 per ICP iteration; no map normals */
} icp_cost;

typedef struct {
    /* SphericalProjector(height, width, 3, up_fov, down_fov), slam/common/projection.py:426-445 */
    int32_t height;
    int32_t width;
    float up_fov;   /* degrees */
    float down_fov; /* degrees */
    /* ICPFrameToModelConfig, slam/odometry/icp_odometry.py:29-64 */
    int32_t max_num_alignments;
    float threshold_delta_pose;
    /* GaussNewtonPointToPlaneConfig.gauss_newton_config {scheme, sigma}, slam/odometry/alignment.py:69-77 */
    int32_t scheme; /* icp_scheme */
    float sigma;
    /* KdTreeLocalMapConfig, slam/odometry/local_map.py:243-251 */
    int32_t local_map_size;
    int32_t num_neighbors_normals;
    /* MI355X-side knobs (no reference counterpart) */
    float cell_size;   /* voxel-hash cell edge in metres; <= 0: auto-tuned towards ~10 map points per occupied cell.
                          Search results never depend on it (the search is exact), only speed does */
    int32_t max_rings; /* fine-level rings searched before the coarse level / exhaustive fallback (default 2; exactness is kept either way) */
    int32_t device;    /* HIP device ordinal */
    int32_t poll_every; /* host polls the device-side `done` flag every N iterations (0: never; implied by threshold <= 0) */
} icp_config;

typedef struct {
    float pose[16];     /* relative pose new frame -> map frame, `new_pose_matrix` of register_new_frame :299 */
    float params[6];    /* [tx,ty,tz,ex,ey,ez], `new_pose_params` :299 */
    int32_t iterations; /* number of align() calls made (= len(losses) :289) */
    int32_t converged;  /* 1 if stopped by ||dx|| < threshold_delta_pose (:292) or the residual-norm guard */
    int32_t status;     /* icp_status of the Gauss-Newton loop */
    int32_t num_targets; /* target rows that took part (after NaN / null masking) */
    int64_t normals_computed; /* map normals estimated during this registration (lazy cache, local_map.py:397-422) */
} icp_register_result;

/* ---- lifecycle ------------------------------------------------------------------------------------------------- */
void icp_default_config(icp_config* cfg);
/* ICPFrameToModel.__init__ (slam/odometry/icp_odometry.py:77-121) */
int icp_create(const icp_config* cfg, icp_ctx** out);
void icp_destroy(icp_ctx* ctx);
const char* icp_last_error(const icp_ctx* ctx);
const char* icp_version(void);
/* hipStream_t to enqueue on (e.g. torch.cuda.current_stream().cuda_stream); NULL = default stream */
int icp_set_stream(icp_ctx* ctx, void* hip_stream);
int icp_synchronize(icp_ctx* ctx);
/* MI355X-side tuning options by name (no reference counterpart; none of them changes a result, only the schedule — with ONE
 * exception, "carry_normals", which changes results by float32 rounding and says so):
 *   "nn_cache" 0 | 1 | 2 (2)        exact nearest-neighbour cache across ICP iterations (2: a missed entry seeds the search)
 *   "hit_records" 0 | 1 (0)         the first cache hit behind a search leaves a record — the winner's point, its normal and a
 *                                   bound on every OTHER map point — from which later iterations decide the hit with 48 coalesced
 *                                   bytes and no gather (the entry's candidate set only where the record does not certify);
 *                                   measured slower than the speculative gathers it replaces: off by default
 *   "late_from" n (-1: never)       fused launches from ICP iteration n on run the LATE kernel: the same iteration built for
 *                                   launches the hit records settle (64 registers, 33 KB of LDS: four workgroups per CU instead
 *                                   of two; the few queries left are searched by a wave each); needs "hit_records"; same bits;
 *                                   off by default: a frame whose queries still miss in those launches pays 20x
 *   "late_waves" 8 | 6 (8)          ... its build: 64 registers (four workgroups per CU) or 80 (three)
 *   "fuse_iteration" 0 | 1 (1)      search + rows + partial sums in one kernel when every normal is ready
 *   "iterate_dense" 0 | 1 (1)       64-register build of that kernel (the whole scan resident in one round of workgroups)
 *   "wave_misses" n (48)            workgroups with up to n cache misses search each of them with a whole wave
 *   "wave_misses_dense" n (4)       the same threshold in the 128-queries-per-block launches of the early iterations
 *   "narrow_from" n (0; -1: never)  from ICP iteration n on the fused kernel takes 512 queries per block, one lane each,
 *                                   instead of 128 with a 4-lane group each; same bits (round 3: 3 — the one-lane ball
 *                                   search of round 4 suits the 512-query shape from the first iteration on)
 *   "frame_seed" 0 | 1 (1)          the neighbours of the last frame seed the first iteration of the next one
 *   "exchange_timeout_ms" (5000)    how long a rank waits for its peers inside the in-library exchange
 *   "knn_rings" n (-1: auto), "knn_lanes" 2 | 4 (4), "target_occupancy" points per cell (16), "search_stats" 0 | 1 | 2 | 3 | 4 (0; dev: 3 / 4 = 1 / 2 + a per-workgroup dump on stderr)
 *   "cell_lists" 0 | 1 (0)          grid build: the table slots a build claims are listed; the next build empties those slots, not the
 *                                   whole table, and the cells' starts are scanned over the lists instead of over every slot
 *                                   (measured: slower for one map per launch, faster for a batch of maps — DESIGN §0)
 *   "scan_poll_limit" n (2^20)      grid build (without "cell_lists"): polls of a predecessor tile's descriptor before a tile of the one-launch table
 *                                   scan computes its prefix from the table itself (a safeguard; tests set 0 to walk that path)
 *   "prune_guard" m (0.002)         searches scan neighbour cells whose box is within m metres of the best distance instead of
 *                                   pruning them (the gap of a pruned cell bounds the cache's L: a guard keeps L off the best)
 *   "refresh_at" n (2), "refresh_margin" m (2e-3)   in launch n, NN-cache entries with less slack than m are searched again
 *   "xcd_sectors" 0 | 1 (1)         the workgroups one XCD receives take one azimuth sector of the range image
 *   "lead_after_dense" 0 | 1 (1)    the first 512-query launch also solves the last 128-query one
 *   "lead_solve" 0 | 1 (1)          launched / unpolled registrations: the 6x6 solve of iteration k runs in an extra workgroup
 *                                   at the head of the (late, 512-queries-per-block) launch k + 1, which publishes the pose to
 *                                   the workgroups of that launch through a mailbox, instead of a launch of its own; same bits
 *   "resident_tail" n (0: never)    from ICP iteration n on (and never before the "wide_until" launches are through) ONE launch
 *                                   runs every remaining iteration of a launched / unpolled registration AND the solve behind the
 *                                   last one: its workgroups (one per CU, 512 queries each) stay resident, keep their queries'
 *                                   targets, cache entries and candidate sets in registers, take every pose from the mailbox and
 *                                   hand their partial rows to the lead — workgroup 0, which also takes its share of the queries
 *                                   — as tagged 8-byte granules; no launch boundary, no cold L2, no k_sum_solve launch per late
 *                                   iteration, and a loop with a live stop threshold ends on the device (no chunks).  Needs every
 *                                   workgroup resident at once (scans of up to 512 x CU count points); every wait is bounded by
 *                                   "lead_timeout_ms": on a GPU shared with foreign work a wait may run out — the registration is
 *                                   then finished on per-iteration launches (same bits) and the context keeps to those
 *                                   (icp_handoff_fallbacks counts such registrations); same bits.  OFF by default: measured at the
 *                                   headline size it costs 7 % (0.382 vs 0.357 ms per frame: what the launch boundary costs per
 *                                   late iteration, 1.1 us, is less than the tagged rows and the skew of 256 resident workgroups
 *                                   add), and nothing is gained on the published configuration's 12-workgroup scans
 *   "resident_tail_max_blocks" n (4096) ... for scans of up to n x 512 points
 *   "lead_timeout_ms" t (50)        wall-clock bound of that poll: a workgroup that does not see the pose in time gives up and
 *                                   the registration ends with ICP_ERR_HIP instead of hanging the GPU (the hand-off assumes the
 *                                   lead workgroup — blockIdx 0 — is dispatched before the pollers fill the machine; raise the
 *                                   bound, or set "lead_solve" 0, where a context shares its GPU with heavy foreign work)
 *   "ball_search" 0 | 1 (1)         NN-cache misses of the fused kernel are first searched by ONE lane each: own cell, then the
 *                                   cells across its nearer faces that a ball of the best distance reaches (99 % of the queries
 *                                   of an ordinary frame); whatever does not fit goes to the 4-lane / whole-wave searches; same
 *                                   bits.  With it the 512-query shape may serve the first iteration as well ("narrow_from" 0)
 *   "ball_lanes" 1 | 2 | 4 | 8 (8)  ... by 2 neighbouring lanes each where the workgroup has at most 256 misses, by 4 where it has
 *                                   at most 128, by 8 where it has at most 64 (the groups of four points of a cell alternate between the lanes, one butterfly
 *                                   merge of their sorted keys at the end); 1: one lane always, and workgroups with up to
 *                                   "wave_misses" misses go straight to the whole-wave search
 *   "ball_empty" 0 | 1 (1)          a miss whose own cell is EMPTY (no neighbour row to read) stays with the ball search when it
 *                                   brings a seed: the other seven cells of its 2x2x2 block by hashed probes, then the same
 *                                   pruning and scan (frames whose initial guess is half a metre off put a sixth of the scan
 *                                   into empty cells; 0: those queries go to the cooperative searches, round 5's schedule)
 *   "far_lanes" 0 | 16 (16)         what the ball search hands back (own cell empty, a ball that leaves its block, more than
 *                                   "ball_max" candidates) where a workgroup has more than "far_min" of them and at most
 *                                   "far_max": every such query searched by 16 lanes, THREADS / 16 queries at a time — seven
 *                                   cell lookups per lane in flight together, the points of the cells found 64 per round;
 *                                   0: the whole-wave / 4-lane searches take them; same bits
 *   "far_max" n (128)               see "far_lanes"
 *   "far_min" n (16)                see "far_lanes": more than n handed-back queries (up to n: a whole wave each)
 *   "wide_until" n (3)              the launches of ICP iterations below n run 1024 threads per 512 queries (two lanes for every
 *                                   miss even when all 512 search: the slowest wave of a launch is a lane walking the candidates
 *                                   of a dense cell alone); same bits
 *   "ball_max" n (256; <= 256)      ... if own cell + surviving cells hold at most n candidates: one lane walks them alone, and the
 *                                   longest walk of a launch sets its duration (heavier queries: the cooperative searches)
 *   "chunked_launch" 0 | 1 (1)      icp_register_launch with threshold_delta_pose > 0 enqueues as many iterations as the last
 *                                   registration ran, plus one; icp_register_end enqueues more while the loop is still running
 *   "flat_rows" 0 | 1 | 2 (2)       how the 4-lane search reads the neighbour cells that survive the box test: 2 = cell by cell,
 *                                   the four lanes striding each cell together; 1 = laid end to end and dealt out candidate by
 *                                   candidate; 0 = every lane walks its own cells (round 2's schedule)
 *   "hoods" 0 | 1 | 2 (2)           neighbourhood lists: the points of every occupied cell's 27-neighbourhood copied into one
 *                                   contiguous run at each grid build (<= 27 x 16 B per map point, maps up to 2^22 points whose
 *                                   normals will be estimated all at once, point-to-plane cost, 5 or 10 neighbours); the
 *                                   kNN normals stream ring 1 from it — 2: one lane per map point, sorted 32-bit keys, the
 *                                   uncertified points finished by a launch of their own, one wave each; 1: four lanes per point
 *                                   (round 3); 0: no lists, the 27 cells of the neighbour row are walked
 *   "carry_normals" 0 | 1 (1)       a POSE-ONLY map update (icp_map_update without a cloud and without an eviction: every map
 *                                   point keeps its neighbours) rotates the normals the grid already holds with the points
 *                                   (n' = R^-1 n) instead of clearing them; the reference clears its cache on every build_model
 *                                   (local_map.py:365-369) and re-estimates on first touch — the same vectors up to the float32
 *                                   rounding of the re-expressed points (poses agree to ~1e-7 m).  0: the reference's schedule,
 *                                   every rebuild clears the cache.  Updates that insert or evict always clear it
 *   "lazy_fused" 0 | 1 | 2 (0)      normals ON DEMAND inside the fused iteration kernel (KdTreeLocalMap.__get_normals' own schedule,
 *                                   local_map.py:397-422: estimate on first touch, cache until the next build_model): a query whose
 *                                   neighbour has no normal yet waits in its workgroup, whole waves estimate those normals behind
 *                                   the searches — the same exact kNN, covariance and eigen-solve as the all-at-once kernels: the
 *                                   same bits — and store them for every later launch.  1: where the map holds more than
 *                                   "lazy_fused_ratio" times the valid targets of the last registration (a grid-sampled frame
 *                                   against a window of key frames: 6 000 normals estimated instead of 180 000 per frame); 2:
 *                                   wherever the kernel exists (point-to-plane, 10 or 5 neighbours, one GPU); 0 (default): never
 *                                   — measured on the published configuration it moves 137 us of estimation from behind the map
 *                                   update (in the shadow of the host's next frame) onto the path to the pose: 0.55 vs 0.46 ms
 *   "lazy_fused_ratio" r (4)        see "lazy_fused"
 *   "overlap_map_update" 0 | 1 (0)  a map update that needs none of the context's scratch buffers — a pose-only update, or the
 *                                   insertion of a cloud staged with icp_map_stage_cloud — runs on a stream of the context's own,
 *                                   behind everything enqueued so far and beside what the caller enqueues next as long as that
 *                                   touches neither the map nor a registration: the upload, grid sample and projection of the
 *                                   next frame overlap the re-expression, grid rebuild and normal estimation of this one; every
 *                                   other entry point first orders the caller's stream behind the update.  Off by default:
 *                                   measured, the two event hand-offs between the streams cost more than the overlap returns
 *                                   (headline loop 2549 vs 2840 scans/s; published-configuration loop unchanged)
 *   "insert_by_cell" 0 | 1 (1)      behind a map update the points claim their cells of the new grid in the cell order of the
 *                                   previous grid (a rigid step leaves the points of an old cell in one or two new ones: the lanes
 *                                   of a wave share their claims) instead of in insertion order
 *   "normals_list" 0 | 1 (0)        the stragglers of the eager kNN normals — the ~0.3 % of the map points whose k-th neighbour
 *                                   the pair pass does not certify — go to a list and a launch of their own right behind the
 *                                   estimating one: sixteen lanes per point, exact keys, DPP merges, ring 2 by hashed probes
 *                                   (0: each is finished by a whole wave of the workgroup that met it).  Same normals, bit for
 *                                   bit.  Off by default: measured, the pair pass alone takes 37 us and the stragglers' launch
 *                                   55 behind it, against 62 us with the stragglers inside (their chains overlap the other
 *                                   workgroups' pair passes there)
 *   "normals_tail_stream" 0 | 1 (0) the eager kNN normals behind a map update finish their stragglers (the ~0.2 % of the map
 *                                   points whose k-th neighbour ring 1 does not certify: a ~30 us chain of dependent probes
 *                                   each) on that stream of the context's own instead of inside the estimating launch: they run
 *                                   beside the next frame's preprocessing, and the next entry point that touches the map or a
 *                                   registration orders the caller's stream behind them.  Same normals, bit for bit.  Off by
 *                                   default: measured, neither the published-configuration loop (0.354-0.382 vs 0.360-0.366 ms
 *                                   per frame) nor the headline loop under the reference's schedule (2365-2374 vs 2370 scans/s)
 *                                   moves — behind a map update the GPU waits for the host's next upload, not the reverse
 *   "eager_normals_limit" m (2^20)  maps of up to m points get all their normals in one launch behind every map update (and
 *                                   the fused iteration kernel) whatever the scan size; larger maps only when m <= 2 n
 *   "profile_rotate" 0 | 1 (0)      icp_profile_enable brackets one iteration launch per registration (see icp_profile_read_iterations)
 *   "profile_every" n (1)           icp_profile_enable times the kernels of every n-th registration only (an event pair
 *                                   costs ~2 us of stream time: 40 pairs per frame are 10 % of a 0.8 ms registration)
 * The library reads no environment variables. */
int icp_set_option(icp_ctx* ctx, const char* name, double value);
/* runtime re-configuration of the alignment (RIGID_ALIGNMENT.load, slam/odometry/alignment.py:200-208) */
int icp_set_alignment(icp_ctx* ctx, int32_t scheme, float sigma, int32_t max_num_alignments,
                      float threshold_delta_pose);
/* the alignment mode of the registration loop (icp_cost; default point to plane) */
int icp_set_cost(icp_ctx* ctx, int32_t cost);

/* ---- projection: Projector.build_projection_map (slam/common/projection.py:331-418) ------------------------------
 * xyz [n,3] -> vertex map [3,H,W] planar (zeros where empty), nearest point wins each pixel.
 * index_out (optional, [H*W] int32): winning point index per pixel, -1 where empty. */
int icp_project(icp_ctx* ctx, const float* xyz, int64_t n, int mem, float* vmap_out, int32_t* index_out, int out_mem);
/* The same projection for the DEVICE-RESIDENT pipeline (device pointers only): vmap_out [3,H,W] as above and rows_out
 * [H*W,3] = the same pixels as rows, i.e. vmap.permute(1, 2, 0).reshape(-1, 3) — what ICPFrameToModel.sample_points
 * (slam/odometry/icp_odometry.py:301-308) indexes and the registration takes as targets — written by the same launch
 * instead of a transposing copy per frame. */
int icp_project_rows(icp_ctx* ctx, const float* xyz, int64_t n, float* vmap_out, float* rows_out);
/* torch__spherical_projection (slam/common/projection.py:11-73): float pixel coordinates rows/cols [n] (diagnostics) */
int icp_project_pixels(icp_ctx* ctx, const float* xyz, int64_t n, int mem, float* rows_out, float* cols_out,
                       int out_mem);

/* ---- dataset side: KITTIOdometrySequence.correct_scan (slam/dataset/kitti_dataset.py:202-231) ----------------------
 * scan [n, stride] float32 rows (stride = 4 for KITTI's x, y, z, reflectance .bin records, 3 for plain xyz) ->
 * corrected xyz [n,3] float64 (the reference's einsum promotes to float64). */
int icp_kitti_correct_scan(icp_ctx* ctx, const float* scan, int64_t n, int stride, int mem, double* xyz_out, int out_mem);

/* ---- voxel grid sampling: voxelise / voxel_hashing / sample_from_hashes (slam/common/pointcloud.py:13-79,170-195),
 * GridSample.filter (slam/preprocessing.py:213-226) -----------------------------------------------------------------
 * indices_out [>= n] int64: original index of the first point of every distinct voxel hash, ordered by ascending
 * int64 hash; *count_out = number of samples; points_out (optional) [count,3] gathered samples. */
int icp_grid_sample(icp_ctx* ctx, const float* xyz, int64_t n, int mem, double voxel_size, int64_t* indices_out,
                    float* points_out, int64_t* count_out, int out_mem);
/* voxel coordinates [n,3] int64 and hashes [n] int64 (either output may be NULL) */
int icp_voxel_hash(icp_ctx* ctx, const float* xyz, int64_t n, int mem, double voxel_size, int64_t* voxels_out,
                   int64_t* hashes_out, int out_mem);

/* float64-input variants: `GridSample` applied to the float64 output of `Distortion` (config/slam/preprocessing/
 * grid_sample.yaml: distortion -> grid_sample on "distorted"); points_out is float64 [count,3] */
int icp_grid_sample_f64(icp_ctx* ctx, const double* xyz, int64_t n, int mem, double voxel_size, int64_t* indices_out,
                        double* points_out, int64_t* count_out, int out_mem);

/* The same selections for the DEVICE-RESIDENT pipeline (device pointers only; no reference counterpart beyond GridSample
 * itself): nothing is read back to the host — no synchronisation, every launch asynchronous on the context's stream.  The
 * outputs hold n rows: the V samples first (ascending int64 hash, as above), then NaN points and index -1; V goes to
 * *count_out, a DEVICE int32 (NULL: not reported).  Consumers that mask NaN rows — icp_project, the registration entry
 * points, icp_map_stage_cloud / icp_map_update — take the padded array as it is: a frame then needs ONE synchronisation, the
 * one that hands its pose to the host. */
int icp_grid_sample_padded(icp_ctx* ctx, const float* xyz, int64_t n, double voxel_size, int64_t* indices_out,
                           float* points_out, int32_t* count_out);
int icp_grid_sample_padded_f64(icp_ctx* ctx, const double* xyz, int64_t n, double voxel_size, int64_t* indices_out,
                               double* points_out, int32_t* count_out);

/* Voxelization.filter / voxel_normal_distribution (slam/preprocessing.py:63-98, slam/common/pointcloud.py:83-167):
 * voxel coordinates [n,3] and hashes [n] (optional), voxel_ids_out [n] = rank of the point's hash among the distinct
 * hashes, *num_voxels_out = V, and (all three or none) sizes_out [V] int64 point counts, means_out [V,3] float32,
 * covs_out [V,3,3] float32 = sum (p - mean)(p - mean)^T (unnormalised, as in the reference), voxels ordered by
 * ascending int64 hash.  Per-voxel outputs must hold n entries (V <= n). */
int icp_voxel_statistics(icp_ctx* ctx, const float* xyz, int64_t n, int mem, double voxel_size, int64_t* voxels_out,
                         int64_t* hashes_out, int64_t* voxel_ids_out, int64_t* num_voxels_out, int64_t* sizes_out,
                         float* means_out, float* covs_out, int out_mem);

/* ---- de-skew: Distortion.filter (slam/preprocessing.py:144-191) --------------------------------------------------
 * Every point moves by the fraction alpha = (t - t_min) / (t_max - t_min) of the initial motion estimate `rel_pose`:
 * out = slerp(I, R, alpha) p + alpha t  (alpha = 0 when all timestamps are equal).  xyz [n,3] float32, timestamps [n]
 * float64, rel_pose row-major 4x4 float64, out [n,3] float64 (the reference's einsum promotes to float64). */
int icp_distort(icp_ctx* ctx, const float* xyz, const double* timestamps, int64_t n, int mem, const double rel_pose[16],
                double* xyz_out, int out_mem);

/* ---- local map: KdTreeLocalMap (slam/odometry/local_map.py:254-427) ---------------------------------------------- */
int icp_map_init(icp_ctx* ctx);                                             /* init()               :279-288 */
int icp_map_set(icp_ctx* ctx, const float* xyz, int64_t m, int mem);        /* set_map_pointcloud() :289-299 */
/* update() :302-362 — move the map by inv(rel), append `new_xyz` (rows with NaN dropped; NULL / n = 0: pose-only
 * update), evict the oldest cloud beyond local_map_size, rebuild the search structure and clear the normal cache.
 * *inserted_out (optional) = number of rows appended.  rel_pose = NULL: the pose of the last registration on this
 * context, read on the device (no host round trip; valid after icp_register / icp_register_launch).  If that
 * registration stopped on an error (ICP_ERR_INVALID_JACOBIAN, ICP_ERR_EXCHANGE) the map is NOT moved (the reference
 * raises before it would touch the map; the error itself reaches the caller through icp_register_end).  While the
 * result of a launched registration is still pending, only the pose-only form (new_xyz = NULL) is accepted with
 * rel_pose = NULL: an insertion / eviction must not follow a registration whose status the host has not seen. */
int icp_map_update(icp_ctx* ctx, const float rel_pose[16], const float* new_xyz, int64_t n, int mem, int row_mode,
                   int64_t* inserted_out);
/* The same update in two steps, for a caller that knows BEFORE a registration which cloud it will insert after it
 * (ICPFrameToModel.do_process_next_frame inserts the frame it has just registered, icp_odometry.py:229-231,360-376):
 * icp_map_stage_cloud flags and compacts the valid rows (row_mode as above) into the context and sends their count to the
 * host without waiting for it; icp_map_update_staged then performs update() :302-362 with those rows — same map, same
 * *inserted_out — and has nothing to wait for when anything that synchronises (icp_register_end) ran in between: one host
 * round trip per frame less than icp_map_update with a cloud.  A staged cloud is consumed by the first
 * icp_map_update_staged and replaced by the next icp_map_stage_cloud; rel_pose = NULL as for icp_map_update. */
int icp_map_stage_cloud(icp_ctx* ctx, const float* xyz, int64_t n, int mem, int row_mode);
int icp_map_update_staged(icp_ctx* ctx, const float rel_pose[16], int64_t* inserted_out);
/* update(new_vertex_map=...) :320-324 — appends the pixels of a [3,H,W] vertex map with norm > 0.01 */
int icp_map_update_vertex_map(icp_ctx* ctx, const float rel_pose[16], const float* vmap, int mem,
                              int64_t* inserted_out);
int64_t icp_map_size(const icp_ctx* ctx);
int icp_map_num_clouds(const icp_ctx* ctx);
/* registrations of this context that were finished on per-iteration launches because a hand-off between workgroups ran out
 * of its wall-clock budget ("resident_tail", "lead_solve": a GPU shared with foreign work); the first one switches the
 * context to per-iteration launches for good — results are unaffected (diagnostics; no reference counterpart) */
int icp_handoff_fallbacks(const icp_ctx* ctx);
int icp_map_get(icp_ctx* ctx, float* xyz_out, int out_mem); /* current map [M,3] in insertion order */
/* nearest_neighbor_search() :372-395 + __get_normals :397-422 — exact Euclidean 1-NN (no distance cap) and the
 * lazily estimated normal of every hit.  Outputs [n,3], [n,3], [n] (any may be NULL). */
int icp_nearest_neighbor_search(icp_ctx* ctx, const float* xyz, int64_t n, int mem, float* neighbor_points_out,
                                float* neighbor_normals_out, int32_t* neighbor_index_out, int out_mem);
/* Test support for the search schedule (no reference counterpart; the reference recomputes every neighbour in every
 * iteration, slam/odometry/icp_odometry.py:274-297): the ORIGINAL map index of the point each target of the last
 * registration was matched with in its LAST iteration (-1: masked row), read back from the exact nearest-neighbour cache
 * the fused iteration kernel keeps, and rows 0-2 of the pose that iteration ran with (pose_out, host memory, may be
 * NULL).  Valid right after icp_register / icp_register_end, before the map changes; ICP_ERR_INVALID_ARGUMENT otherwise. */
int icp_last_neighbors(icp_ctx* ctx, int32_t* neighbor_index_out, float pose_out[12], int out_mem);

/* ---- projective local map: ProjectiveLocalMap (slam/odometry/local_map.py:91-240), the reference's "GPU" variant -----
 * compute_normal_map (slam/common/geometry.py:240-295): vertex map [3,H,W] -> normal map [3,H,W] (box-filter plane fit,
 * zero where the pixel is null or |det| <= 1e-6). */
int icp_compute_normal_map(icp_ctx* ctx, const float* vmap, int mem, int kernel_size, float* nmap_out, int out_mem);
/* compute_neighbors (slam/common/geometry.py:397-439): per pixel the closest of k_maps reference vertex maps
 * [k_maps,3,H,W] to the target map [3,H,W] (null pixels never match; ties -> the first map); optionally gathers
 * ref_fields [k_maps,c_fields,H,W] along.  neighbors_out [3,H,W], fields_out [c_fields,H,W]. */
int icp_compute_neighbors(icp_ctx* ctx, const float* tgt_vmap, const float* ref_vmaps, const float* ref_fields,
                          int k_maps, int c_fields, int mem, float* neighbors_out, float* fields_out, int out_mem);
int icp_pmap_init(icp_ctx* ctx); /* init() :113-119 */
/* update() :122-174 — re-express the kept maps by inv(rel_pose), append `vmap` [3,H,W] with its normal map (NULL:
 * pose-only update), drop the oldest beyond local_map_size, rebuild the model (build_model :177-202). */
int icp_pmap_update(icp_ctx* ctx, const float rel_pose[16], const float* vmap, int mem, int normals_kernel_size);
int icp_pmap_num_maps(const icp_ctx* ctx);
/* the model: `_model_vmap` / `_model_nmap` as [K, H*W, 4] float rows (xyz + valid flag, normal + 0) */
int icp_pmap_get_model(icp_ctx* ctx, float* model_v4_out, float* model_n4_out, int out_mem);
/* nearest_neighbor_search() :205-235 — projective association of the (already transformed) points xyz [n,3]:
 * rows9_out [count,9] = neighbour point, neighbour normal, target point per matched pixel, in pixel order. */
int icp_pmap_nearest_neighbor_search(icp_ctx* ctx, const float* xyz, int64_t n, int mem, float* rows9_out,
                                     int64_t* count_out, int out_mem);
/* register_new_frame (slam/odometry/icp_odometry.py:248-299) against the projective map; same contract as
 * icp_register. */
int icp_pmap_register(icp_ctx* ctx, const float* xyz, int64_t n, int mem, int target_mode, const float init_pose[16],
                      icp_register_result* result, double* loss_per_iter_out, float* dx_per_iter_out);

/* ---- rigid alignment: GaussNewtonPointToPlaneAlignment.align (slam/odometry/alignment.py:91-127) on given
 * correspondences; one Gauss-Newton step from x0 = 0 (slam/common/optimization.py:296-344).
 * dx_out[6], pose_out[16] = build_pose_matrix(dx), loss_out = sum (w r)^2, normal_eq_out (optional) = 32 doubles:
 * 21 upper-triangular JtJ, 6 Jtr, sum (w r)^2, sum r^2, row count, 2 pad; residuals_out (optional, [n] float, in the
 * memory space `mem` of the inputs) = (w r)^2 per row, the residual tensor `align` returns (optimization.py:342-344). */
int icp_align_point_to_plane(icp_ctx* ctx, const float* ref_points, const float* tgt_points, const float* ref_normals,
                             int64_t n, int mem, float dx_out[6], float pose_out[16], double* loss_out,
                             double* normal_eq_out, float* residuals_out);

/* GaussNewtonPointToPointAlignment.align (slam/odometry/alignment.py:143-189) on given correspondences: one
 * Gauss-Newton step of PointToPointCost (slam/common/optimization.py:458-560) linearised at x0 (NULL = zeros; with
 * `initialize_with_svd` the caller passes from_pose_matrix(icp_weighted_procrustes(...))).  params_out = x0 + dx,
 * pose_out = build_pose_matrix(params_out); loss_out / normal_eq_out / residuals_out as above. */
int icp_align_point_to_point(icp_ctx* ctx, const float* ref_points, const float* tgt_points, int64_t n, int mem,
                             const float x0[6], float params_out[6], float pose_out[16], double* loss_out,
                             double* normal_eq_out, float* residuals_out);
/* weighted_procrustes (slam/common/registration.py:15-74): closed-form rigid transform target -> reference.  weights
 * [n] float32 or NULL (ones); as in the reference they enter the centroids only.  pose_out row-major 4x4 float64. */
int icp_weighted_procrustes(icp_ctx* ctx, const float* tgt_points, const float* ref_points, const float* weights,
                            int64_t n, int mem, double pose_out[16]);

/* ---- registration: ICPFrameToModel.register_new_frame (slam/odometry/icp_odometry.py:248-299) --------------------
 * All iterations run on the device without host round trips.  loss_per_iter_out / dx_per_iter_out (optional) receive
 * `iterations` entries ([.] double, [.,6] float). */
int icp_register(icp_ctx* ctx, const float* xyz, int64_t n, int mem, int target_mode, const float init_pose[16],
                 icp_register_result* result, double* loss_per_iter_out, float* dx_per_iter_out);

/* Device-side helper of ICPFrameToModel.sample_points for vertex-map targets (slam/odometry/icp_odometry.py:301-308:
 * `target_points[target_points.norm(dim=-1) > 0.0]`) without a host round trip for the count: the rows of xyz [n,3]
 * that pass `target_mode` (rows with a NaN never do; null rows do not under ICP_TARGETS_SKIP_NULL) are written, in
 * order, to the head of out [cap,3]; the rest of `out` is zero-filled, so registering `out` with ICP_TARGETS_SKIP_NULL
 * registers exactly the passing rows — with cap instead of n rows to walk (a 64x2048 vertex map built from a 6000-point
 * grid sample has at most 6000 non-null pixels).  Both pointers are device memory.  cap must be >= the number of passing
 * rows (the caller knows such a bound by construction); passing rows beyond cap are dropped.  Asynchronous. */
int icp_compact_targets(icp_ctx* ctx, const float* xyz, int64_t n, int target_mode, float* out, int64_t cap);

/* ---- multi-GPU seam: one ICP iteration split around the exchange of the packed normal equations -------------------
 * begin -> { accumulate -> [all-reduce the 32 doubles at icp_normal_equations_ptr over RCCL] -> solve } x iters -> end.
 * Every rank registers its own slice of the target points against a replicated map and applies the identical solve.*/
int icp_register_begin(icp_ctx* ctx, const float* xyz, int64_t n, int mem, int target_mode, const float init_pose[16]);
/* begin + every iteration enqueued + the result copied to pinned host memory behind the last iteration, without waiting:
 * work enqueued afterwards on the same context (icp_map_update with rel_pose = NULL: the pose-only branch of
 * ICPFrameToModel.__update_map, icp_odometry.py:379) overlaps the host's wait in icp_register_end, which then blocks on
 * the registration only.  With threshold_delta_pose > 0 only a first chunk of iterations is on the stream when this
 * returns ("chunked_launch": as many as the last registration ran, plus one; launches behind an early stop are
 * device-side no-ops); icp_register_end enqueues more while the loop is still running.  Every entry point that changes
 * what those held-back iterations would see — any icp_map_update / icp_map_update_staged / icp_map_stage_cloud / icp_map_set / icp_map_init / icp_map_update_vertex_map,
 * icp_set_option / icp_set_cost / icp_set_alignment / icp_set_stream, icp_map_normals_owned / _install, icp_pmap_register,
 * the next icp_register_launch — first enqueues ALL of them, so the stream order is the call order, as if every iteration
 * had been enqueued here. */
int icp_register_launch(icp_ctx* ctx, const float* xyz, int64_t n, int mem, int target_mode, const float init_pose[16]);
/* the same with the initial guess = the pose of the previous registration on this context, READ ON THE DEVICE: the
 * constant-velocity initialisation (`ConstantVelocityInitialization`, slam/initialization.py:103-119 — the last relative
 * pose) without the host in the loop.  Up to two launched registrations may await their icp_register_end (which returns
 * them oldest first), so a frame loop can enqueue frame t + 1 (and the map update by the device-resident pose of frame
 * t) before it has collected the pose of frame t: the GPU never idles between frames while the host still receives
 * every pose, one frame later. */
int icp_register_launch_from_last(icp_ctx* ctx, const float* xyz, int64_t n, int mem, int target_mode);
int icp_iteration_accumulate(icp_ctx* ctx); /* search + normals + reduce into the 32-double device vector */
int icp_iteration_solve(icp_ctx* ctx);      /* 6x6 solve + pose update from the (possibly all-reduced) vector */
int icp_register_end(icp_ctx* ctx, icp_register_result* result, double* loss_per_iter_out, float* dx_per_iter_out);
void* icp_normal_equations_ptr(icp_ctx* ctx); /* device pointer, 32 doubles */
/* use caller-owned device memory (e.g. a torch tensor RCCL can reduce in place) for the 32-double vector */
int icp_set_normal_equations_buffer(icp_ctx* ctx, void* device_ptr);

/* ---- B sequences per launch: batched registration ------------------------------------------------------------------
 * The reference registers ONE sequence, one frame at a time (ICPFrameToModel.register_new_frame,
 * slam/odometry/icp_odometry.py:248-299; one `SLAM` object per process, slam/slam.py:84-163): a chain of dependent,
 * latency-bound iterations that leaves most of an MI355X idle.  A batch ties B contexts of ONE device — B independent
 * sequences, each with its own local map, scan and registration state — together so that iteration k of all B
 * registrations is ONE kernel launch ("HBM-bound operation is only approachable by batching many
 * independent registrations per launch"): the per-sequence arguments sit in a descriptor table in device memory, every
 * sequence has a lead workgroup of its own, and one host thread enqueues the launches of all B.  Per sequence the
 * arithmetic is that of icp_register_launch on the same context with the same options: the same poses, bit for bit.
 *   icp_batch_create           ties `count` contexts (1..ICP_BATCH_MAX_SEQUENCES, same device, same alignment configuration
 *                              — max_num_alignments, scheme, sigma — and same schedule options) together; they stay usable on
 *                              their own between batched calls and must outlive the batch;
 *   icp_batch_set_stream       icp_set_stream on every member (a batch enqueues everything on ONE stream);
 *   icp_batch_register_launch  icp_register_launch (from_last = 0; init_poses = count x 16 floats or NULL: identity) or
 *                              icp_register_launch_from_last (from_last = 1) on every member: xyz[b] / n[b] = member b's
 *                              scan.  With a forced iteration count every iteration is enqueued; with a live stop
 *                              threshold a first chunk (as many as the slowest member ran last time, plus one), further
 *                              chunks from icp_batch_register_end while a member is still running — a member whose loop has
 *                              ended idles on the device.  While the batch holds iterations back its members refuse the
 *                              single-context entry points (ICP_ERR_INVALID_ARGUMENT).  Point-to-plane registrations on the
 *                              fused path only (eager normals, no exchange, no profiling): ICP_ERR_INVALID_ARGUMENT otherwise;
 *   icp_batch_project          icp_project for every member in two launches: xyz[b] [n[b],3] -> vmap_out[b] [3,H,W] of member b
 *                              (Projector.build_projection_map, slam/common/projection.py:331-418, as ICPFrameToModel._read_input
 *                              calls it per frame, icp_odometry.py:333); DEVICE pointers only;
 *   icp_batch_map_update       icp_map_update(member, NULL, NULL, ...) for every member: the pose-only update by the
 *                              device-resident pose of the registration just launched (icp_odometry.py:379) — the B grid
 *                              rebuilds in four launches (held-back iterations are enqueued first: the update reads the END of
 *                              the registration);
 *   icp_batch_register_end     icp_register_end for every member (results[b]; loss_per_iter_out / dx_per_iter_out:
 *                              count x max_num_alignments (x 6) entries or NULL): ONE wait for all of them.  Returns the
 *                              first member's non-zero status, every member's own in results[b].status. */
#define ICP_BATCH_MAX_SEQUENCES 32
typedef struct icp_batch icp_batch;
int icp_batch_create(icp_ctx* const* ctxs, int32_t count, icp_batch** out);
void icp_batch_destroy(icp_batch* batch);
const char* icp_batch_last_error(const icp_batch* batch);
int icp_batch_set_stream(icp_batch* batch, void* hip_stream);
int icp_batch_register_launch(icp_batch* batch, const float* const* xyz, const int64_t* n, int mem, int target_mode,
                              const float* init_poses, int from_last);
int icp_batch_project(icp_batch* batch, const float* const* xyz, const int64_t* n, float* const* vmap_out);
int icp_batch_map_update(icp_batch* batch);
int icp_batch_register_end(icp_batch* batch, icp_register_result* results, double* loss_per_iter_out,
                           float* dx_per_iter_out);

/* ---- multi-GPU exchange inside the library ("one-shot P2P write+flag all-reduce") -----------------
 * The per-iteration exchange of the scan-sharded registration without leaving the library: after these three calls
 * icp_register / icp_register_launch on every rank enqueue, per ICP iteration, the iteration kernel and ONE kernel that
 * sums this rank's partial rows, writes the 32 doubles into every peer's inbox over xGMI, waits for all ranks'
 * contributions, adds them in rank order and solves — identical poses on all ranks, no host involvement, no RCCL call.
 *   icp_exchange_create  allocates this rank's inbox (uncached device memory) and returns its IPC handle (64 bytes,
 *                        hipIpcMemHandle_t) for the caller to all-gather over its own side channel (torch.distributed
 *                        `all_gather_object`, MPI, a file);
 *   icp_exchange_connect opens the `world` handles (rank order; the own entry is ignored) and switches the context to
 *                        exchange mode; every rank must then issue the same sequence of registrations;
 *   icp_exchange_destroy leaves exchange mode and releases the mappings.
 * A peer that does not deliver within the budget (option "exchange_timeout_ms", default 5000) ends the registration
 * with ICP_ERR_EXCHANGE instead of hanging the GPU — and the context LEAVES exchange mode: the ranks' sequence counters
 * and inbox tags may have diverged, so registrations fall back to the rank-local solve until every rank has called
 * icp_exchange_create + icp_exchange_connect again (fresh inboxes, counters at zero).  world <= 16. */
#define ICP_EXCHANGE_HANDLE_BYTES 64
int icp_exchange_create(icp_ctx* ctx, int32_t rank, int32_t world, void* handle_out);
int icp_exchange_connect(icp_ctx* ctx, const void* handles);
int icp_exchange_destroy(icp_ctx* ctx);

/* ---- multi-GPU, map-sharded normal estimation (BASELINE configs[3]: 1M-point map over 8 GPUs) ------
 * KdTreeLocalMap.__get_normals (slam/odometry/local_map.py:397-422) costs O(map) per map update once the map is much
 * larger than the scan.  Every rank keeps the whole map (queries stay local) but estimates only the normals of the points
 * whose 1-metre spatial bucket hashes to it: icp_map_normals_owned fills normals_by_index [M,4] float (device memory of
 * the caller, e.g. a torch tensor; nx, ny, nz, 1 at the ORIGINAL index of every owned point, zeros elsewhere), the
 * caller sums the arrays of all ranks (one RCCL all-reduce: every index has exactly one non-zero contribution, so the
 * sum is exact) and icp_map_normals_install scatters the result into this rank's normal cache, after which
 * registrations run the fused path.  world = 1 reduces to the eager single-GPU estimation (same values). */
int icp_map_normals_owned(icp_ctx* ctx, int32_t rank, int32_t world, float* normals_by_index);
int icp_map_normals_install(icp_ctx* ctx, const float* normals_by_index);

/* ---- profiling hooks ---------------------------------------------------------------------------------------------
 * Accumulated HIP-event time (ms) and launch count of the dominant kernel (the per-iteration nearest-neighbour
 * search) since the last reset; measured on the context's stream.  `enable` is a bit mask: 1 = search kernel,
 * 2 = reduction, 4 = normal estimation (0 disables). */
int icp_profile_enable(icp_ctx* ctx, int enable);
int icp_profile_read(icp_ctx* ctx, double* search_ms_out, int64_t* search_launches_out, double* reduce_ms_out,
                     double* normals_ms_out);
/* the search kernel's time by ICP iteration index: ms_out / launches_out [cap] (index cap - 1 and beyond: not reported
 * separately beyond 64).  With the option "profile_rotate" 1 only ONE iteration launch per registration is bracketed —
 * iteration (number of the registration) mod max_num_alignments — so that every frame of a run can be sampled at the cost
 * of two event records, and every iteration index is seen equally often. */
int icp_profile_read_iterations(icp_ctx* ctx, double* ms_out, int64_t* launches_out, int32_t cap);
/* what an event pair ADDS to the launch it brackets, on the context's stream (median of `samples`, microseconds): the
 * pair is put around a kernel that spins for exactly 20 us of the device's wall clock, with a busy predecessor and a
 * successor like a launch inside a registration has them; the result is the measured time minus those 20 us — the dispatch
 * latency behind the barrier packet of the first event, which back-to-back launches do not pay and rocprofv3's kernel
 * durations do not contain.  bench.py subtracts it from its live event timing of the dominant kernel. */
int icp_profile_event_floor(icp_ctx* ctx, int32_t samples, double* median_us_out);

#ifdef __cplusplus
}
#endif
#endif /* ICP_MI355X_H */
