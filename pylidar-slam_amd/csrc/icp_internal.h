// Internal declarations shared by the HIP translation units of libicp_mi355x.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string>
#include <vector>

#include "icp_mi355x.h"

namespace icp {

// ---------------------------------------------------------------------------------------------------------------------
// Voxel-hash grid over the local map (the search structure that replaces the reference's pykdtree KDTree,
// slam/odometry/local_map.py:365-369).  Open addressing, 16-byte entries so one dwordx4 load resolves a probe.
// ---------------------------------------------------------------------------------------------------------------------
struct alignas(16) GridEntry {
    unsigned long long key;  // packed cell coordinates, GRID_EMPTY if free
    int start;               // first point of the cell in the cell-sorted point array
    int count;               // number of points of the cell
};
static constexpr unsigned long long GRID_EMPTY = ~0ull;
static constexpr int CELL_OFFSET = 1 << 20;  // cells are in [-2^20, 2^20)

struct GridView {
    const GridEntry* table;
    unsigned int mask;   // table size - 1 (power of two)
    float h;             // cell edge
    float inv_h;
    const float4* pts;   // cell-sorted map points: (x, y, z, bits(original index))
    int m;               // number of map points
    // neighbour rows: for every occupied cell the (start, count) of its 27-neighbourhood (index = (oz+1)*9+(oy+1)*3+
    // (ox+1), own cell at 13), so that a query in an occupied cell needs ONE hash probe instead of 27.  Indexed by the
    // table slot of the cell (sparse: only the rows of occupied slots are ever written or read), so the entry and the
    // row of a query's cell are fetched in the same round
    const int2* rows;        // [table slots][ROW_STRIDE]
    const int* row_of_pos;   // cell-sorted point position -> slot (= row) of its cell
    // coarse level (cell edge COARSE_FACTOR * h, hashed, own cell-sorted copy of the points): takes over when a query is
    // farther than `max_rings` fine cells from the map, so that the exhaustive scan stays a last resort
    const GridEntry* ctable;
    unsigned int cmask;
    float ch, cinv_h;
    const float4* cpts;
    const int* pos_of_orig;  // original map index -> position in the fine cell-sorted array
    const float4* hood;      // neighbourhood lists (hash_grid.hip::k_hood_build; header = row entry 27) or nullptr
    int flat_rows;           // option "flat_rows": how the 4-lane search scans its surviving neighbour cells (0 / 1 / 2)
    float prune_guard;       // option "prune_guard" (m): neighbour cells closer than best + guard are scanned, not pruned
    int* dbg;                // dev-only path counters (option "search_stats" = 1), nullptr in production
    long long* stamps;       // dev-only phase timestamps (option "search_stats" = 1 | 2), nullptr in production
};
static constexpr float COARSE_FACTOR = 4.0f;
static constexpr int COARSE_RINGS = 6;
static constexpr int HOOD_IDLE_RELEASE = 8;  // builds in a row without a reader before the neighbourhood lists are given back
static constexpr int HOOD_PER_POINT = 30;  // neighbourhood-list entries reserved per map point (27 + padding of the runs); the counters follow
static constexpr int SORTED_PAD = 4;   // +inf entries behind the last cell-sorted point (search.hip::search_ball_lane reads groups of four)
static constexpr int ROW_STRIDE = 28;  // 27 cells + 1 pad: rows are 224 B, 16-byte aligned

__host__ __device__ inline unsigned long long pack_cell(int cx, int cy, int cz) {
    return ((unsigned long long)((unsigned)(cx + CELL_OFFSET) & 0x1FFFFFu)) |
           ((unsigned long long)((unsigned)(cy + CELL_OFFSET) & 0x1FFFFFu) << 21) |
           ((unsigned long long)((unsigned)(cz + CELL_OFFSET) & 0x1FFFFFu) << 42);
}

__host__ __device__ inline unsigned int hash_cell(unsigned long long k) {
    k ^= k >> 33;
    k *= 0xff51afd7ed558ccdull;
    k ^= k >> 33;
    k *= 0xc4ceb9fe1a85ec53ull;
    k ^= k >> 33;
    return (unsigned int)k;
}

__device__ inline int cell_coord(float v, float inv_h) {
    float c = floorf(v * inv_h);
    c = fminf(fmaxf(c, (float)(-CELL_OFFSET + 8)), (float)(CELL_OFFSET - 8));
    return (int)c;
}

// ---------------------------------------------------------------------------------------------------------------------
// Device-resident registration state (one per context): the whole Gauss-Newton loop reads / writes this, the host only
// copies it back once per frame.
// ---------------------------------------------------------------------------------------------------------------------
static constexpr int NEQ = 32;          // packed normal equations: 21 H + 6 g + loss + r2 + count + 2 pad
static constexpr int NEQ_USED = 30;

struct RegState {
    float pose[16];
    float pose_prev[16];  // pose of the previous iteration (the NN cache bounds how far each target moved since)
    float params[6];
    int iter;        // align() calls made so far
    int done;        // 1: loop finished (converged / guard / error); later launches are no-ops
    int converged;
    int status;      // icp_status
    int n_targets;   // valid target rows (from the last reduction)
    int n_worklist;  // map points queued for normal estimation in the current iteration
    long long normals_computed;
    int grid_cells;  // occupied fine cells of the last grid build (feeds the cell-size auto-tuning; rides home with the result)
    int handoff_timeouts;  // workgroups that gave up waiting for the pose of the lead workgroup (never seen; -> ICP_ERR_HIP)
};

// ---------------------------------------------------------------------------------------------------------------------
// Pose mailbox of the fused iteration launches ("lead solve").  The 6x6 solve of iteration k does not get a launch of
// its own: workgroup 0 of launch k + 1 (the first one the hardware dispatches) sums the partial rows launch k left, solves,
// updates the RegState and publishes the new pose here, while the other workgroups — which already have their targets,
// cache entries, cached neighbours and normals in flight — poll for it.  A generation of the pose is 14 granules of 8
// bytes, (tag << 32) | payload, each written and read by ONE sc1 (agent-scope, relaxed) access, so no fence orders
// anything: a reader takes a granule when its tag is the generation it waits for.  Two parities: generation g lives in
// parity g & 1, generation g - 1 (the pose the NN-cache bounds refer to) stays readable in the other one.
//   granule 0..11: rows 0-2 of the 4x4 pose; 12: done (the loop is finished: the launch has nothing to do); 13: iteration
// ---------------------------------------------------------------------------------------------------------------------
// Every generation is written into BOX_REPLICAS copies, each on a cache line of its own; workgroup b polls copy
// b % BOX_REPLICAS: a thousand workgroups polling ONE line queue up behind one memory channel.
static constexpr int BOX_WORDS = 16;
static constexpr int BOX_USED = 14;
static constexpr int BOX_REPLICAS = 16;
static constexpr size_t BOX_BYTES = (size_t)2 * BOX_REPLICAS * BOX_WORDS * sizeof(unsigned long long);

__device__ inline const unsigned long long* box_granule(const unsigned long long* box, unsigned gen, int replica, int k) {
    return box + ((size_t)(gen & 1u) * BOX_REPLICAS + (size_t)replica) * BOX_WORDS + k;
}

__device__ inline void box_store(unsigned long long* __restrict__ box, unsigned gen, int k, unsigned bits) {
    const unsigned long long v = ((unsigned long long)gen << 32) | bits;
#pragma unroll
    for (int r = 0; r < BOX_REPLICAS; ++r)
        __hip_atomic_store(const_cast<unsigned long long*>(box_granule(box, gen, r, k)), v, __ATOMIC_RELAXED,
                           __HIP_MEMORY_SCOPE_AGENT);
}

// what a fused iteration launch needs to know about the hand-off (box == nullptr: classic launch, pose from the RegState)
struct LeadArgs {
    unsigned long long* box = nullptr;
    unsigned gen = 0;                // generation of the pose this launch consumes
    int solve = 0;                   // 1: a lead workgroup (grid + 1) solves the previous iteration first and publishes `gen`
    const double* prev_partials = nullptr;
    int prev_rows = 0, prev_quad = 0;
    double* neq = nullptr;
    double* loss_hist = nullptr;
    float* dx_hist = nullptr;
    int hist_cap = 0;
    long long timeout_ticks = 0;     // 100 MHz wall clock
};

// The start of a registration by the 64 lanes of ONE wave (round 5: the serial form — state_init, then 224 mailbox stores
// and 12 history stores by one thread, each behind the one before — was 5-6 us of every frame's critical path): the state
// (state_init's fields, same values), generation `gen` of the pose mailbox = the initial guess, entry 0 of the pose history.
// Call with all 64 lanes of a wave active; `lane` = lane index.
// keep_pose: the initial guess is the pose the state already holds — the result of the previous registration, i.e. the
// constant-velocity initialisation (slam/initialization.py:103-119) without a host round trip.
__device__ inline void state_init_wave(RegState* st, const float* init, int keep_pose, unsigned long long* __restrict__ box,
                                       unsigned gen, float* __restrict__ hist, int lane) {
    float p = 0.f;
    if (lane < 16) {
        p = keep_pose ? st->pose[lane] : init[lane];
        st->pose[lane] = p;
        st->pose_prev[lane] = p;
    } else if (lane < 22) {
        st->params[lane - 16] = 0.f;  // new_pose_params = zeros (icp_odometry.py:267)
    } else if (lane == 22) {
        st->handoff_timeouts = 0;
        st->iter = 0;
        st->done = 0;
        st->converged = 0;
        st->status = 0;
        st->n_targets = 0;
        st->n_worklist = 0;
        st->normals_computed = 0;
    }
    if (hist && lane < 12) hist[lane] = p;
    if (box) {
        for (int idx = lane; idx < BOX_USED * BOX_REPLICAS; idx += 64) {
            const int k = idx / BOX_REPLICAS, r = idx % BOX_REPLICAS;
            const float v = __shfl(p, k < 12 ? k : 0, 64);  // (all lanes of the wave take part in the shuffle)
            const unsigned bits = k < 12 ? __float_as_uint(v) : 0u;  // granule 12: done = 0, 13: iteration 0
            __hip_atomic_store(const_cast<unsigned long long*>(box_granule(box, gen, r, k)),
                               ((unsigned long long)gen << 32) | bits, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
}

static constexpr size_t STATE_BLOCK = 256;  // bytes reserved for the RegState at the head of the state allocation

// ---------------------------------------------------------------------------------------------------------------------
// Multi-GPU exchange of the packed normal equations inside the library (include/icp_mi355x.h, icp_exchange_*): every
// rank owns an INBOX in its own HBM with one slot per rank and parity; a rank publishes its 32 doubles by writing them
// straight into slot [parity][its rank] of EVERY peer's inbox (peer-mapped over xGMI) followed by a tag, then waits for
// the tags of all ranks in its own inbox and adds the slots up in rank order — the same sum on every rank, bit for bit.
// ---------------------------------------------------------------------------------------------------------------------
static constexpr int EXCHANGE_MAX_RANKS = 16;
struct alignas(64) ExchangeSlot {
    double v[32];
    unsigned long long tag;  // sequence number of the exchange the payload belongs to (0: never written)
    unsigned long long pad[7];
};
struct ExchangeView {
    ExchangeSlot* inbox[EXCHANGE_MAX_RANKS];  // inbox[r] = rank r's inbox ([2][world] slots) as mapped in THIS process
    unsigned long long* seq;                  // device counter of completed exchanges (own memory)
    int rank, world;
    long long timeout_ticks;                  // 100 MHz wall-clock ticks to wait for the peers
};

// Inside the loop of a persistent kernel (the resident tail, search.hip) the optimiser hoists every value derived from the
// thread index — dozens of per-lane addresses, group and lane numbers — out of the loop and keeps them in registers around
// its whole body: 80 spilled VGPRs at a budget of 256, measured.  A local `threadIdx` that shadows the builtin and is
// re-read through an opaque move per trip keeps those values local to where they are used.
struct LocalTid {
    unsigned x;
};
__device__ __forceinline__ LocalTid reloaded_tid() {
    unsigned v = ::threadIdx.x;
    asm volatile("" : "+v"(v));
    return LocalTid{v};
}

struct Pose16 {
    float m[16];
};

// Re-expression of the map (local_map.py:346-348) that the next grid build carries out in its first launch
// (map_move_device.h): moved = inv(pose) applied to `m` points.  The relative pose comes by value from the host or,
// st != nullptr, from the device-resident result of the last registration (no host round trip).
struct MapMoveJob {
    const float* in = nullptr;
    float* out = nullptr;
    long long m = 0;               // 0: nothing to move
    Pose16 rel{};
    const RegState* st = nullptr;  // device-resident pose instead of `rel`
};

// The cell lists of a grid build (hash_grid.hip): the table slots it claims, fine and coarse level apart, with their counts;
// `prev_*` = what the previous build left (the slots this build has to empty).  counts == nullptr: no lists (the table is
// cleared and scanned as a whole); prev_counts == nullptr: the whole table is cleared (a first build, another table)
struct CellLists {
    int* fine = nullptr;
    int* coarse = nullptr;
    int* counts = nullptr;  // [2]: fine, coarse
    const int* prev_fine = nullptr;
    const int* prev_coarse = nullptr;
    const int* prev_counts = nullptr;
};

// Everything the four launches of a grid build take (hash_grid.hip): filled by build_grid, handed to the kernels by value
// (one map) or through a table in device memory (B maps per launch: icp_batch_map_update)
struct GridBuildDesc {
    // k_grid_clear: the table of both levels emptied; + NN cache -> frame seeds, + the re-expression of the kept points, + the
    // normals carried through a pose-only update
    GridEntry* table;
    unsigned n2;              // slots of both levels
    unsigned tsize;           // slots of one level
    const int4* nn_cache;
    const float4* old_pts;    // cell-sorted points of the PREVIOUS grid
    int seed_n, seed_m, seed_evicted;
    int* seed;
    MapMoveJob move;
    int* scan_ticket;
    unsigned long long* hood_used;
    const float4* old_normals;
    const int* old_nflag;
    int carry_m;
    float4* carry;
    unsigned clear_blocks;
    CellLists lists;
    // k_grid_insert2
    const float* xyz;
    int m;
    float inv_h, inv_hc;
    int *slot_of, *rank_of, *cslot_of, *crank_of;
    const float4* order_pts;  // = old_pts when the claims follow the previous grid's cell order, else nullptr
    int order_old_m, order_evicted, order_kept;
    int* visit;
    int visit_n;
    unsigned visit_blocks;    // workgroups of the claiming / scattering threads
    // k_grid_scan
    unsigned long long *desc_a, *desc_b;
    unsigned scan_gen;
    int poll_limit;
    int* slot_of_cell;
    int* ncells;
    unsigned scan_blocks;
    // k_grid_rows_scatter
    int2* rows;
    unsigned row_blocks;
    float4 *sorted, *csorted, *normals;
    int *nflag, *row_of_pos, *pos_of_orig;
    // behind the four launches (host side): neighbourhood lists wanted for this build
    int with_hoods;
    unsigned long long hood_cap;
};

// the arguments of k_pack_targets (grid_sample.hip), as the batched launch reads them from device memory; n = 0: nothing to do
struct PackDesc {
    const float* xyz;
    float4* out;
    RegState* st;
    unsigned long long* box;
    float* hist;
    Pose16 init;
    int n, keep_pose;
    unsigned gen;
    int pad;
};

struct AlignParams {
    int scheme;
    float sigma;
    float threshold_delta_pose;
    int max_iters;
    float* pose_hist;  // [iteration][12]: rows 0-2 of the pose every iteration of the registration runs with (NN cache)
};

// ---------------------------------------------------------------------------------------------------------------------
// Host-side context
// ---------------------------------------------------------------------------------------------------------------------
struct DeviceBuffer {
    void* ptr = nullptr;
    size_t bytes = 0;
    hipError_t reserve(size_t need, bool keep = false, hipStream_t stream = nullptr);
    void release();
    template <typename T>
    T* as() const { return reinterpret_cast<T*>(ptr); }
};

struct Profile {
    bool enabled = false;
    int mask = 0;  // bit0 search, bit1 reduce, bit2 normals
    int every = 1;             // "profile_every": registrations between two timed ones
    long long registrations = 0;
    bool sample_now = true;
    std::vector<std::pair<hipEvent_t, hipEvent_t>> pool;
    struct Rec { int kind; int ev; int iter; };
    std::vector<Rec> pending;
    double ms[3] = {0, 0, 0};
    long long launches[3] = {0, 0, 0};
    // "profile_rotate": of the fused iteration launches of a sampled registration only ONE is bracketed — iteration
    // (number of the registration) mod max_num_alignments — so every frame can be sampled (two event records instead of
    // forty: an event pair breaks the back-to-back dispatch around it) and every iteration index is seen equally often
    int rotate = 0;
    int rotate_index = 0;
    static constexpr int ITER_SLOTS = 64;      // search-kernel time by iteration index (the last slot takes the rest)
    double ms_iter[ITER_SLOTS] = {};
    long long launches_iter[ITER_SLOTS] = {};
};

}  // namespace icp

struct icp_ctx {
    icp_config cfg;
    hipStream_t stream = nullptr;
    std::string error;
    // ---- local map
    int64_t map_m = 0;                 // number of map points
    std::vector<int64_t> cloud_sizes;  // `_local_map_num_elements`
    int map_cur = 0;                   // which of map_xyz[2] is current
    icp::DeviceBuffer map_xyz[2];      // [M,3] float, insertion order
    // ---- hash grid
    icp::DeviceBuffer table;           // GridEntry[T]
    unsigned int table_size = 0;
    icp::DeviceBuffer sorted_pts;      // float4[M]
    icp::DeviceBuffer normals;         // float4[M] (by cell-sorted position)
    icp::DeviceBuffer nflag;           // int[M]: 0 none, 2 queued, 1 ready
    icp::DeviceBuffer slot_of, rank_of;  // int[M] temporaries of the build
    icp::DeviceBuffer slot_of_cell, rows, row_of_pos;
    icp::DeviceBuffer csorted, pos_of_orig, cslot_of, crank_of;  // coarse level (its table follows the fine one)
    icp::GridEntry* ctable_ptr = nullptr;
    unsigned int ctable_size = 0;
    icp::DeviceBuffer scan_tmp;
    // "cell_lists": the slots a grid build claims are listed, the next build empties those (not the table) and the cells'
    // starts are scanned over the lists (hash_grid.hip: CellLists).  Two sets in alternation, [set][fine | coarse].
    // MEASURED (round 6): for ONE sequence the build got slower — 7.0 + 12.4 + 14.2 + 8.0 = 41.6 us against 6.7 + 8.2 +
    // 9.2 + 8.2 = 32.3 (the list appends cost the insertion 800 same-address counter updates, and one workgroup per level
    // scanning 6 500 scattered counts is slower than 128 tiles streaming 262 144 slots) — 2722 vs 2835 scans/s; with eight or
    // sixteen maps per launch it is the faster build (6865 vs 6657 scans/s at B = 16: the bytes of B whole tables count there).
    // Off by default; the batched bench leg switches it on
    int cell_lists = 0;
    icp::DeviceBuffer cell_list[2][2];
    icp::DeviceBuffer cell_counts;     // int[2 sets][2 levels] (+ padding)
    int cell_set = 0;                  // the set the NEXT build writes
    const void* cells_table = nullptr; // the table (and its size, and the point count) the other set describes; nullptr: nothing
    unsigned cells_tsize = 0;
    int64_t cells_m = 0;
    icp::DeviceBuffer scan_desc;       // descriptors of the one-launch table scan of the grid build (k_grid_scan)
    uint64_t scan_builds = 0;          // launches of that scan so far (its descriptors are tagged with it)
    int scan_poll_limit = 1 << 20;     // "scan_poll_limit" (dev): polls of a predecessor's descriptor before a tile sums the table itself
    icp::DeviceBuffer worklist;        // int[M]
    icp::MapMoveJob move_job;          // pending re-expression of the kept points (consumed by build_grid)
    bool grid_valid = false;
    uint64_t grid_gen = 0;             // bumped by every grid build: cell-sorted positions are only valid within one
    bool normals_ready = false;        // every map normal already estimated (eager mode) since the last rebuild
    int64_t normals_eager_count = 0;
    float cell_h = 0.5f;               // cell edge of the current grid (auto-tuned when cfg.cell_size <= 0)
    int occupied_cells = 0;
    double target_occupancy = 16.0;    // auto-tuning target, map points per occupied cell (option "target_occupancy")
    int64_t stats_m = 0;               // map size the occupancy figure belongs to
    bool stats_pending = false;
    int64_t stats_m_pending = 0;
    float stats_h = 0.f, stats_h_pending = 0.f;  // cell edge of the build the figure was measured on
    // ---- registration
    icp::DeviceBuffer targets;         // staged copy of host targets
    const float* tgt_ptr = nullptr;    // device pointer of the current targets
    int64_t tgt_n = 0;
    int tgt_mode = 0;
    icp::DeviceBuffer nn_pos;          // int[N]
    icp::DeviceBuffer nn_cache;        // int4[N]: (NN position | iteration << 24, bits(L), two more candidate positions or -1) — L = lower bound on the distance to every map point outside that set
    int iter_in_registration = 0;      // iterations of the registration in progress enqueued so far (= the device's RegState.iter while it runs)
    bool cache_fresh = false;          // a fused launch of THIS registration has (re)written every NN-cache entry
    int cost = 0;                      // icp_cost of the registration loop (icp_set_cost)
    // tuning options (icp_set_option; none of them changes a result)
    int knn_rings = -1;                // "knn_rings": fine rings of the kNN before the coarse level (-1: min(max_rings, 2))
    int knn_lanes = 4;                 // "knn_lanes": lanes per map point in the kNN kernels (4 or 2)
    int use_nn_cache = 2;              // "nn_cache": 0 off, 1 exact NN cache, 2 + a missed entry seeds the search
    // "late_from": fused launches from that iteration on run the late kernel (search.hip: records settle nearly every query); -1
    // (the default): never.  MEASURED (round 6): 26.5 vs 31 us per late launch of a batch of eight sequences (four
    // workgroups per CU instead of two), no difference for one sequence (its late iterations wait for the lead workgroup) — and
    // a frame whose constant-velocity guess is off (the two turn-around frames of the benchmark's trajectory: every query still
    // misses in iterations 3 and 4) searches its 512 misses per workgroup two waves at a time: 440 us per launch instead of 20
    int late_from = -1;
    int late_waves = 8;                // "late_waves": 8 | 6 — waves per SIMD the late kernel is built for (64 / 80 registers: 4 / 3 workgroups per CU)
    // "hit_records": the first hit behind a search leaves a record (winner, normal, bound) later hits are decided from
    // (search.hip).  Bit-identical, 64 instead of 128 bytes requested per hit — and MEASURED slower (round 6): 2684 vs
    // 2831 scans/s on the headline, 6472 vs 6586 with sixteen sequences per launch: the 2 % of queries whose winner and runner-up
    // are closer than the pose still moves fail the record's test and fetch the candidate set in a DEPENDENT round trip behind
    // the pose, and the slowest workgroup of a launch is one that holds such a query (phase A max 7.04 vs 6.32 us; the
    // speculative gathers of the set were free: they ride behind the lead's solve).  Off by default
    int hit_records = 0;
    icp::DeviceBuffer nn_rec;          // float4[2 N]: the hit records
    int fuse_iteration = 1;            // "fuse_iteration": search + rows + partial sums in one kernel when normals are ready
    int narrow_from = 0;               // "narrow_from": ICP iteration from which the fused kernel runs with 128 threads per block (-1: never)
    int wave_misses_dense = 4;         // "wave_misses_dense": the same threshold in the 128-query shape (early iterations)
    int wave_misses = 48;              // "wave_misses": blocks with up to that many NN-cache misses search them a wave each
    int iterate_dense = 1;             // "iterate_dense": 64-VGPR build of that kernel (4 blocks per CU resident)
    int frame_seed = 1;                // "frame_seed": last frame's neighbours seed the first iteration of the next one
    int search_stats = 0;              // "search_stats": count which path resolved each query (dev)
    bool search_stats_blocks = false;  // ... "search_stats" 3 | 4: + the stamps of every workgroup in the dump (dev)
    icp::DeviceBuffer dbg_counts;
    // a cloud staged for the next map update (icp_map_stage_cloud): its valid rows, in order; their count travels to the host
    // behind the compaction, beside whatever is enqueued after it — the update then needs no synchronisation of its own
    icp::DeviceBuffer staged_xyz;
    int* staged_count_host = nullptr;  // pinned
    hipEvent_t staged_event = nullptr;
    int64_t staged_rows = -1;          // rows handed to the staging call (-1: nothing staged)
    icp::DeviceBuffer tgt4;            // float4[N]: the targets the kernels read
    // what nn_cache currently describes: `cache_n` targets against the grid of generation `cache_gen` (map of cache_m points)
    int64_t cache_n = 0, cache_m = 0;
    uint64_t cache_gen = 0;
    icp::DeviceBuffer seed_orig;       // int[N]: original map index of the neighbour each scan slot had in the last frame
    int64_t seed_n = 0;                // valid entries of seed_orig (0: none)
    icp::DeviceBuffer partials;        // double[2][blocks][NEQ]: two parities (a lead launch sums the rows of the previous launch while its own workgroups write theirs)
    size_t partials_half = 0;          // bytes of one parity
    int partials_parity = 0;           // parity the NEXT fused launch writes
    icp::DeviceBuffer pose_hist_buf;   // float[hist_cap + 1][12]: pose history of the registration in progress
    float* pose_hist = nullptr;
    icp::DeviceBuffer posebox;         // pose mailbox of the lead launches (BOX_BYTES)
    unsigned box_gen = 0;              // last pose generation published (or enqueued to be)
    long long eager_normals_limit = 1 << 20;  // "eager_normals_limit": maps up to that many points get all their normals at once whatever the scan size
    int flat_rows = 2;                 // "flat_rows" (GridView): 0 lane by lane, 1 flattened list, 2 cell by cell with four lanes
    float prune_guard = 2e-3f;         // "prune_guard" (GridView)
    float refresh_margin = 2e-3f;      // "refresh_margin" (m) / "refresh_at" (iteration): NN-cache entries with less slack than
    int refresh_at = 2;                // that are searched again in that one launch, which searches anyway (IterInputs)
    int searches_in_registration = 0;  // k_search_rows launches of the registration in progress (unfused loops): from the
    long long nn_pos_n = -1;           // second on, nn_pos (of nn_pos_n targets against grid generation nn_pos_gen) seeds
    unsigned long long nn_pos_gen = 0; // the searches
    int lead_after_dense = 1;          // "lead_after_dense": the first narrow launch solves the last dense one (enqueue_iterations)
    int xcd_sectors = 1;               // "xcd_sectors": workgroups of one XCD take one sector of the scan (launch_iterate_fused)
    int hoods = 2;                     // "hoods": neighbourhood lists for the kNN normals (1: four lanes per point, 2: one lane per point + a straggler queue)
    bool hoods_valid = false;          // ... built for the current grid
    int hood_idle_builds = 0;          // grid builds in a row that had no use for the lists (their space goes back after HOOD_IDLE_RELEASE of them)
    // "carry_normals": a pose-only map update (no insertion, no eviction: every point keeps its neighbours) rotates the
    // normals already estimated with the points instead of clearing them (local_map.py:368 zeroes them; re-estimated in
    // the new frame they are the same vectors up to float32 rounding of the re-expressed points).  0: the reference's
    // schedule — every rebuild clears the cache
    int carry_normals = 1;
    bool carry_job = false;            // the pending grid build carries the normals of the `carry_m` points of the old grid over
    int64_t carry_m = 0;
    icp::DeviceBuffer normals_carry;   // float4[M] by ORIGINAL index: (rotated normal, 1) or zeros
    bool sharded_normals = false;      // icp_map_normals_owned has been used on this context (the lists serve it too)
    icp::DeviceBuffer hood;            // float4[<= 30 M] + the fill counter behind it
    int chunked_launch = 1;            // "chunked_launch": launched registrations with a live threshold are enqueued in chunks
    int ball_search = 1;               // "ball_search": NN-cache misses of the fused kernel searched by one lane each first (search_ball_lane)
    int wide_until = 3;                // "wide_until": fused launches of iterations below that run with 1024 threads per 512 queries
    int far_lanes = 16;                // "far_lanes": 16 lanes for a query the ball search hands back, where a workgroup has a few dozen (0 | 16)
    int far_min = 16;                  // "far_min": ... more than that many (fewer: a wave each)
    int far_max = 128;                 // "far_max": ... up to that many of them (THREADS / 16 at a time)
    int ball_lanes = 8;                // "ball_lanes": a miss of a workgroup with few of them gets 2 or 8 lanes of the ball search (IterInputs)
    int ball_empty = 1;                // "ball_empty": a seeded miss in an EMPTY cell is searched by the ball search too (its 2x2x2 block by seven hashed probes) instead of the cooperative searches
    int ball_max = 256;                // "ball_max": ... if they have at most that many candidates (a lane walks them alone: the longest walk of a launch sets its duration)
    double lead_timeout_ms = 50.0;     // "lead_timeout_ms": how long a workgroup of a lead launch polls the pose mailbox before it gives up (-> ICP_ERR_HIP)
    // "lazy_fused": normals on demand INSIDE the fused iteration kernel (its LAZY instantiation, search.hip) instead of all of
    // them behind every map update: 0 never (default), 1 where the map holds more than `lazy_fused_ratio` times the valid
    // targets of the last registration, 2 wherever the kernel exists (point-to-plane, 10 or 5 neighbours, fused iterations).
    // Built for VERDICT r4 item 3(i) and MEASURED on the published configuration (6 100 valid targets, 181 000 map points):
    // the same bits as the all-at-once estimation, 6 100 normals instead of 181 000 per frame — and 0.55 instead of 0.46 ms
    // per frame: the all-at-once kernel (89 + 15 us for its lists) runs behind the map update, in the shadow of the host's
    // preparation of the next frame, while the on-demand estimation (24 waiting queries per workgroup, four lanes each: a
    // 45 us chain at two busy waves per CU) sits inside the first iteration launch, on the path to the frame's pose:
    // 222 us of iteration launches per frame instead of 85.  Off by default
    int lazy_fused = 0;
    double lazy_fused_ratio = 4.0;
    bool lazy_now = false;             // ... the registration in progress runs on it
    int64_t last_valid_targets = 0;    // valid target rows of the last collected registration (RegState.n_targets)
    int lead_solve = 1;                // "lead_solve": the solve of iteration k in the head of launch k + 1 (no k_sum_solve launches)
    // "resident_tail": from that iteration on (0: never — the default) ONE launch runs all remaining iterations of a launched /
    // unpolled registration — its workgroups stay resident and hand their partial rows to its lead as tagged granules
    // (search.hip).  Built for VERDICT r4 item 1 and MEASURED (round 5): 256 workgroups x 20 forced iterations
    // 0.382 vs 0.357 ms per frame — per late iteration the tail saves the launch boundary (1.1 us) and a cold L2, and pays
    // 2.4 instead of 1.8 us for the rows to reach the lead (tagged granules through the fabric instead of plain loads behind
    // a kernel boundary) and ~1 us of skew among 256 resident workgroups: 9.95 vs 9.4 us; 12 workgroups with a live stop
    // threshold (the published configuration): 0.606-0.613 vs 0.605-0.633 ms per frame, no difference.  Off by default;
    // the option, its tests and the timed-out-hand-off recovery it made necessary stay
    int resident_tail = 0;
    int resident_tail_max_blocks = 4096;  // "resident_tail_max_blocks": ... for scans of up to that many 512-query workgroups
    bool tail_disabled = false;        // a hand-off of this context timed out once (a GPU shared with foreign work): per-iteration launches from then on
    bool handoff_disabled = false;     // ... and no lead launches either (the user's "lead_solve" is left as set; setting it again re-arms them)
    bool lead_latched = false;         // lead launches for the registration in progress: decided once, in register_begin
    int tail_capacity = -1;            // workgroups of the tail's shape the device holds at once (-1: not asked yet)
    int handoff_fallbacks = 0;         // registrations finished on per-iteration launches behind a timed-out hand-off
    bool batch_hold = false;           // a batched registration of this context holds iterations back (api.hip: icp_batch): individual entry points refuse
    bool counted_registering = false;  // this context is counted among the registering contexts of its device (api.hip)
    bool update_behind_registration = false;  // a pose-only map update by the device pose is enqueued behind an uncollected registration
    icp::DeviceBuffer tail_rows;       // tagged super-rows of the tail: [rows][NEQ][2] granules of 8 bytes
    icp::DeviceBuffer vox_out;         // staging of icp_voxel_statistics' host outputs
    icp::DeviceBuffer state;           // RegState + histories
    // per-iteration histories live behind the RegState in the same allocation (one D2H copy brings back everything):
    // [RegState, padded to STATE_BLOCK bytes | double loss[hist_cap] | float dx[hist_cap][6]]
    double* loss_hist = nullptr;
    float* dx_hist = nullptr;
    icp::DeviceBuffer neq_own;         // double[NEQ]
    double* neq = nullptr;             // active normal-equation vector (own or caller supplied)
    int hist_cap = 0;
    int reduce_blocks = 0;
    bool in_registration = false;
    // asynchronous result hand-off (icp_register_launch): state + grid stats + histories copied to pinned host memory
    // behind the last iteration, an event recorded after the copies.  Two slots: a second registration may be launched
    // (icp_register_launch_from_last: its initial guess is read on the device) before the first one's result has been
    // collected, so the GPU never waits for the host between frames
    bool have_device_pose = false;   // RegState holds the pose of a finished / enqueued registration
    struct ResultSlot {
        void* host = nullptr;
        size_t bytes = 0;
        hipEvent_t event = nullptr;
        hipEvent_t wait = nullptr;   // what icp_register_end waits for: `event`, or the one event of a batched registration
        bool stats = false;          // a grid-stats copy travels with this result
        int64_t stats_m = 0;
        float stats_h = 0.f;
        int64_t eager_normals = 0;   // normals estimated eagerly for this registration
    } rslot[2];
    // the pinned slot (device-visible address, bytes) the LAST solving launch of the registration being enqueued writes the
    // result block into itself (launch_sum_solve); result_folded: it has done so — no copy launch behind it
    char* result_fold_to = nullptr;
    size_t result_fold_bytes = 0;
    bool result_folded = false;
    // a launched registration with a live stop threshold is enqueued in CHUNKS (as many iterations as the previous frame
    // needed, plus one): icp_register_end looks at the result and enqueues the next chunk only if the loop is still
    // running, instead of paying for max_num_alignments launches of which most find `done` set.  launch_remaining =
    // iterations of the NEWEST pending registration not enqueued yet (anything that must follow the whole registration on
    // the stream — a map update by the device pose, another launch — enqueues them first)
    int launch_remaining = 0;
    int launch_enqueued = 0;         // iterations enqueued so far for it
    int last_iterations = 0;         // iterations the last collected registration ran (sizes the next first chunk)
    int r_head = 0;                  // oldest pending slot
    int r_count = 0;                 // results launched and not yet collected (0..2)
    bool result_pending() const { return r_count > 0; }
    // ---- in-library multi-GPU exchange (icp_exchange_*)
    bool exchange_on = false;
    icp::ExchangeView xview{};
    void* x_inbox = nullptr;          // own inbox (uncached device memory, exported through IPC)
    void* x_peer[icp::EXCHANGE_MAX_RANKS] = {};  // peers' inboxes opened through IPC (nullptr for the own rank)
    icp::DeviceBuffer x_seq;
    double exchange_timeout_ms = 5000.0;  // option "exchange_timeout_ms"
    // "overlap_map_update": a map update that needs none of the context's scratch buffers runs on a stream of its own
    // (api.hip::map_update_impl); entry points that touch the map join it first (DeviceGuard)
    // MEASURED (round 5) and left off: the two event hand-offs between the streams cost more than the overlap returns — the
    // headline loop 2549 vs 2840 scans/s (its next frame has 9 us of projection to overlap), the published configuration's
    // loop 0.44-0.48 vs 0.44-0.46 ms per frame
    int overlap_map_update = 0;
    int normals_tail_stream = 0;        // "normals_tail_stream": the stragglers of the eager kNN normals behind a map update run on the map stream (measured: no gain, off)
    int normals_list = 0;               // "normals_list": the stragglers of the two-lane kNN-normal kernel go to a list and a launch of their own, sixteen lanes each (k_normals_tail16: built in round 6, bit-identical, measured slower — 37 + 55 us against 62); 0: a wave of their workgroup each
    bool normals_tail_on_map_stream = false;  // (argument of the launch in flight)
    icp::DeviceBuffer normals_tail;     // [2 + M] int: count, workgroups of the tail launch that are through, positions of the stragglers
    int* normals_tail_list = nullptr;   // (argument of the launch in flight)
    hipStream_t map_stream = nullptr;
    hipEvent_t map_done_event = nullptr, map_start_event = nullptr;
    bool map_stream_busy = false;
    hipEvent_t switch_event = nullptr;  // orders a change of stream (icp_set_stream) behind the work of the old one
    // ---- scratch for projection / sampling / io
    // the map update in front of the coming grid build kept `order_kept` points of the `order_old_m` the current grid holds
    // (the oldest `order_evicted` dropped): k_grid_insert2 may claim in the old grid's cell order (option "insert_by_cell")
    bool order_job = false;
    int64_t order_old_m = 0, order_evicted = 0, order_kept = 0;
    int insert_by_cell = 1;
    int seed_job_n = 0, seed_job_m = 0, seed_job_evicted = 0;  // NN cache -> frame seeds, pending for the next grid build
    const void* zbuf_clean = nullptr;  // the z-buffer allocation the resolve kernel has left all-clear, and its size
    int zbuf_clean_pixels = 0;
    icp::DeviceBuffer zbuf, stage_in, stage_out, stage_out2, flags, scan_a, scan_b, sort_tmp, keys_a, keys_b, vals_a,
        vals_b, counter;
    // ---- projective local map (row a19)
    icp::DeviceBuffer pm_v, pm_n, pm_mv, pm_mn, pm_z, pm_tmp;
    std::vector<int> pm_slots;                 // storage slot of every kept map, oldest first
    struct PmPose { float m[16]; };
    std::vector<PmPose> pm_poses;              // pose of every kept map's frame in the current frame
    icp::Profile prof;
};

namespace icp {

#define ICP_HIP(ctx, expr)                                                                          \
    do {                                                                                            \
        hipError_t _e = (expr);                                                                     \
        if (_e != hipSuccess) {                                                                     \
            (ctx)->error = std::string(#expr) + ": " + hipGetErrorString(_e);                       \
            return ICP_ERR_HIP;                                                                     \
        }                                                                                           \
    } while (0)

// ---- hash_grid.hip
// defer != nullptr: everything but the launches — *defer receives their arguments (the caller launches them, e.g. for B maps
// at once: launch_grid_build_batch) and calls build_grid_finish afterwards
int build_grid(icp_ctx* ctx, GridBuildDesc* defer = nullptr);
int build_grid_finish(icp_ctx* ctx, const GridBuildDesc& d);  // the neighbourhood lists, where the build wants them
int launch_grid_build_batch(icp_ctx* first, const GridBuildDesc* table_host, const GridBuildDesc* table_dev, int count);
// exclusive scan of n ints (in place allowed); total written to *total_dev (device int) if non-null
int exclusive_scan_i32(icp_ctx* ctx, const int* in, int* out, int64_t n, int* total_dev);
// ordered compaction: copies rows (row_floats floats each) whose flag != 0; count to *count_dev
int compact_rows(icp_ctx* ctx, const float* in, const int* flags, int64_t n, int row_floats, float* out,
                 int* count_dev, int64_t cap = -1);
// the valid rows of an [n,3] cloud (no NaN; not null under skip_null), in order, in two launches; the count lands in
// *count_dev and (when given) in *count_host_mapped — pinned host memory, device-visible address
int compact_valid_rows(icp_ctx* ctx, const float* xyz, int64_t n, bool skip_null, float* out, int* count_dev, int64_t cap,
                       int* count_host_mapped);

// ---- search.hip
int launch_search_raw(icp_ctx* ctx);  // 1-NN without the pose transform (LocalMap seam)
int launch_search(icp_ctx* ctx);    // 1-NN of the current targets -> nn_pos, queues missing normals
int launch_normals(icp_ctx* ctx);   // kNN normals for the worklist
// kNN normals of every map point (eager mode); tail_may_overlap: called at the end of a map update — the stragglers of the
// estimation may run on the map stream (option "normals_tail_stream")
int launch_normals_all(icp_ctx* ctx, bool tail_may_overlap = false);
int launch_gather_neighbors(icp_ctx* ctx, int64_t n, float* pts_out, float* nrm_out, int32_t* idx_out);
int launch_last_neighbors(icp_ctx* ctx, int iteration, int* out_dev);  // NN cache -> matched map point per target (tests)
// map-sharded normals: the owned share by original index (zeros elsewhere) / install the all-reduced array
int launch_normals_owned(icp_ctx* ctx, int rank, int world, float* by_index_dev);
int launch_normals_install(icp_ctx* ctx, const float* by_index_dev);
// before a grid rebuild: nn_cache of the last registration -> seeds of the next frame (`evicted` oldest points dropped;
// indices_survive = false when the map is replaced wholesale)
int stash_frame_seeds(icp_ctx* ctx, int64_t evicted, bool indices_survive);
int run_seed_job(icp_ctx* ctx);

// ---- gauss_newton.hip
AlignParams make_align_params(const icp_ctx* ctx);
int launch_reduce(icp_ctx* ctx);    // residual / Jacobian rows -> packed normal equations (ctx->neq)
int launch_solve(icp_ctx* ctx);     // 6x6 solve + pose update on the device
int launch_sum_solve(icp_ctx* ctx, int rows, int quad = 1, const double* partials = nullptr, bool publish = false,
                     bool last = false);  // (with an exchange connected: + the all-reduce over the ranks; partials: ctx->partials unless given)
int launch_sum_partials(icp_ctx* ctx, int rows, int quad = 1);
// fused search + point-to-plane rows + per-block partial sums (needs every touched normal ready); *blocks_out = rows
// rows written; quad: base rows (1) or super-rows (0).  lead: the launch takes its pose from the mailbox; with prev_rows > 0
// its lead workgroup first solves the equations the previous fused launch left (prev_rows / prev_quad rows of the other
// parity of ctx->partials)
int launch_iterate_fused(icp_ctx* ctx, int* rows_out, int* quad_out, bool lead = false, int prev_rows = 0, int prev_quad = 1,
                         int tail_iters = 0);  // tail_iters > 0: a resident tail over that many iterations (ask fused_tail_possible first)
// ---- B sequences per launch (icp_batch_*): one fused iteration / one sum + solve of every member in ONE launch, the
// members' arguments in descriptor tables (pinned host memory -> device memory, copied once per frame by the caller)
struct BatchedIteration {
    int count = 0, per_seq = 0, shape = 0;
    int records = 0;  // the members run with hit records: the builds with the record test compiled in
    int rows[ICP_BATCH_MAX_SEQUENCES] = {};
    int quad[ICP_BATCH_MAX_SEQUENCES] = {};
};
size_t iterate_desc_bytes();
size_t sum_solve_desc_bytes();
int prepare_iterate_batch(icp_ctx* const* ctxs, int count, bool lead_mode, const int* prev_rows, const int* prev_quad,
                          void* table_host, BatchedIteration* out);
int launch_iterate_batch(icp_ctx* first, const BatchedIteration& it, const void* table_dev);
int prepare_sum_solve_batch(icp_ctx* const* ctxs, int count, const int* rows, const int* quad, bool publish, bool last,
                            void* table_host);
int launch_sum_solve_batch(icp_ctx* first, int count, const void* table_dev);
bool fused_tail_possible(icp_ctx* ctx, int prev_rows, int tail_iters);
bool fused_tail_planned(icp_ctx* ctx, int iters);
unsigned long long* pose_box(icp_ctx* ctx);   // device pointer of the mailbox (allocated by ensure_state)
unsigned next_box_generation(icp_ctx* ctx);    // a new generation number for a pose about to be published
bool next_fused_launch_is_narrow(const icp_ctx* ctx);  // the shape launch_iterate_fused will pick for the next iteration
int launch_reduce_solve(icp_ctx* ctx);  // single-GPU path: reduction, final sum and solve without the exchange seam
int launch_align_given(icp_ctx* ctx, const float* ref, const float* tgt, const float* nrm, int64_t n,
                       float* residuals_dev);
int launch_align_p2p(icp_ctx* ctx, const float* ref, const float* tgt, int64_t n, const float* x0,
                     float* residuals_dev);
int launch_reduce_p2p(icp_ctx* ctx, bool solve);  // point-to-point rows of a registration iteration
int launch_procrustes_pass(icp_ctx* ctx, const float* tgt, const float* ref, const float* w, int64_t n,
                           const float* mu_tgt, const float* mu_ref, double* host_out);

// ---- projection.hip
// keep_keys: the z-buffer keys stay for the caller; rows_dev (optional): the same pixels as [H*W, 3] rows
int project_device(icp_ctx* ctx, const float* xyz_dev, int64_t n, float* vmap_dev, int32_t* index_dev, bool keep_keys = false,
                   float* rows_dev = nullptr);
int project_pixels_device(icp_ctx* ctx, const float* xyz_dev, int64_t n, float* rows_dev, float* cols_dev);
int project_batch_device(icp_ctx* const* ctxs, int count, const float* const* xyz_dev, const int64_t* n, float* const* vmap_dev);
int kitti_correct_device(icp_ctx* ctx, const float* scan_dev, int64_t n, int stride, double* out_dev);

// ---- projective.hip
int normal_map_device(icp_ctx* ctx, const float* vmap_dev, int ks, float* nmap_dev);
int neighbors_device(icp_ctx* ctx, const float* tgt, const float* ref, const float* fld, int k_maps, int c_fld,
                     float* nb_out, float* fld_out);
int pmap_store_slot(icp_ctx* ctx, int slot, const float* vmap_dev, const float* nmap_dev);
int pmap_build(icp_ctx* ctx);
int pmap_iterate(icp_ctx* ctx, int* blocks_out);
int pmap_associate(icp_ctx* ctx, const float* xyz_dev, int64_t n, float* rows9_dev, int* flags_dev);

// ---- grid_sample.hip
int voxel_hash_device(icp_ctx* ctx, const float* xyz_dev, int64_t n, double voxel, long long* voxels_dev,
                      long long* hashes_dev);
int grid_sample_f64_device(icp_ctx* ctx, const double* xyz_dev, int64_t n, double voxel, long long* indices_dev,
                           double* points_dev, int* count_dev, int* count_host, bool padded = false);
int distort_device(icp_ctx* ctx, const float* xyz_dev, const double* ts_dev, int64_t n, const double* rel_pose16,
                   double* out_dev);
// targets -> float4 rows (x, y, z, bits(row)) in ctx->tgt4
// targets -> float4 rows; with `init` the registration state is initialised by the same launch (init->m = the initial
// guess; keep_pose: the pose the state already holds — the previous result — is the guess)
int prepare_targets(icp_ctx* ctx, const float* xyz_dev, int64_t n, const Pose16* init = nullptr, bool keep_pose = false,
                    PackDesc* defer = nullptr);  // defer: the launch is left to the caller (launch_pack_targets_batch)
int launch_pack_targets_batch(icp_ctx* first, const PackDesc* table_host, const PackDesc* table_dev, int count);
int voxel_statistics_device(icp_ctx* ctx, const float* xyz_dev, int64_t n, double voxel, long long* voxels_dev,
                            long long* hashes_dev, long long* ids_dev, long long* sizes_dev, float* means_dev,
                            float* covs_dev, int* count_dev);
int grid_sample_device(icp_ctx* ctx, const float* xyz_dev, int64_t n, double voxel, long long* indices_dev,
                       float* points_dev, int* count_dev, int* count_host, bool padded = false);

// ---- profiling helpers (api.hip)
int prof_begin(icp_ctx* ctx, int kind, int iter = -1);  // iter: index of the ICP iteration (search kernel)
void prof_end(icp_ctx* ctx, int token);

inline RegState* reg_state(icp_ctx* ctx) { return ctx->state.as<RegState>(); }

// Normals of the whole map at once (one dense launch, then the fused iteration kernel) or lazily for the map points the
// scan touches (local_map.py:397-422; three more launches per iteration, each a latency-bound chain when the scan is
// small)?  Same values either way.  Eager costs ~1 us per 1000 map points; measured, it wins whenever the map is at most
// twice the scan, and for any scan while the map stays below ~10^6 points (a 6 000-point grid sample against 180 000
// map points: 0.19 ms eager vs 0.45 ms for four lazy iterations; 200 000 points against 10^6: 2.8 vs 4.7 ms per frame of
// twenty iterations).  n = the number of targets (of the registration at hand, or of the last one as a stand-in).
// normals on demand inside the fused kernel (option "lazy_fused")?  n = target rows of the registration at hand (valid or
// not: a padded frame has 131 072 of which 6 000 count — the last registration's valid count is the better estimate)
inline bool wants_lazy_fused(const icp_ctx* ctx, int64_t n) {
    const int kn = ctx->cfg.num_neighbors_normals + 1;
    if (ctx->cost != ICP_COST_POINT_TO_PLANE || !ctx->fuse_iteration || ctx->lazy_fused <= 0 || (kn != 11 && kn != 6) ||
        ctx->exchange_on || ctx->sharded_normals || ctx->map_m >= (1 << 24))
        return false;
    if (ctx->lazy_fused >= 2) return true;
    const int64_t valid = ctx->last_valid_targets > 0 && ctx->last_valid_targets < n ? ctx->last_valid_targets : n;
    return (double)ctx->map_m > ctx->lazy_fused_ratio * (double)valid;
}

inline bool wants_eager_normals(const icp_ctx* ctx, int64_t n) {
    if (ctx->cost != ICP_COST_POINT_TO_PLANE) return false;
    if (wants_lazy_fused(ctx, n)) return false;
    return ctx->map_m <= 2 * n || ctx->map_m <= (int64_t)ctx->eager_normals_limit;
}

}  // namespace icp
