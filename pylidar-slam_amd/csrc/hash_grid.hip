// Voxel-hash grid construction over the local map + the scan / compaction primitives used around it.
//
// Replaces `KDTree(self._model_points)` of the reference (slam/odometry/local_map.py:365-369), which is rebuilt on
// every `update()`.  Build = counting sort through the hash table itself (no comparison sort):
//   1. insert: every point claims / finds the slot of its cell (64-bit CAS) and takes a rank in it (atomic add)
//   2. exclusive scan of the per-slot counts -> start of every cell in the cell-sorted array
//   3. scatter: point -> start[slot] + rank, stored as float4 (x, y, z, bits(original index))
// The rank order inside a cell is not deterministic, but every consumer breaks distance ties on the ORIGINAL index,
// so search results are.
#include <math.h>

#include "icp_internal.h"
#include "map_move_device.h"

namespace icp {

static constexpr int SCAN_THREADS = 256;
static constexpr int SCAN_ITEMS = 8;
static constexpr int SCAN_TILE = SCAN_THREADS * SCAN_ITEMS;

// ---------------------------------------------------------------------------------------------------------------------
// block-wide exclusive scan of one value per thread (256 threads = 4 waves)
// ---------------------------------------------------------------------------------------------------------------------
__device__ inline int block_exclusive_scan(int v, int* total, int* lds /* >= 8 ints */) {
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    int incl = v;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        int t = __shfl_up(incl, o, 64);
        if (lane >= o) incl += t;
    }
    if (lane == 63) lds[wave] = incl;
    __syncthreads();
    int wave_off = 0, tot = 0;
    const int nw = (blockDim.x + 63) >> 6;
    for (int w = 0; w < nw; ++w) {
        int s = lds[w];
        if (w < wave) wave_off += s;
        tot += s;
    }
    __syncthreads();
    *total = tot;
    return wave_off + incl - v;
}

__global__ __launch_bounds__(SCAN_THREADS) void k_scan_tile_sums(const int* __restrict__ in, long long n,
                                                                  int* __restrict__ sums) {
    __shared__ int lds[8];
    const long long base = (long long)blockIdx.x * SCAN_TILE + (long long)threadIdx.x * SCAN_ITEMS;
    int s = 0;
#pragma unroll
    for (int k = 0; k < SCAN_ITEMS; ++k) {
        long long i = base + k;
        if (i < n) s += in[i];
    }
    int tot;
    block_exclusive_scan(s, &tot, lds);
    if (threadIdx.x == 0) sums[blockIdx.x] = tot;
}

// single block: exclusive scan of `nb` tile sums in place (sequential over chunks of blockDim), total -> *total_out
__global__ __launch_bounds__(1024) void k_scan_sums(int* __restrict__ sums, int nb, int* __restrict__ total_out) {
    __shared__ int lds[32];
    __shared__ int carry_s;
    if (threadIdx.x == 0) carry_s = 0;
    __syncthreads();
    for (int base = 0; base < nb; base += blockDim.x) {
        int i = base + threadIdx.x;
        int v = (i < nb) ? sums[i] : 0;
        // 1024-thread scan: wave scan + wave totals
        const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
        int incl = v;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            int t = __shfl_up(incl, o, 64);
            if (lane >= o) incl += t;
        }
        if (lane == 63) lds[wave] = incl;
        __syncthreads();
        int wave_off = 0, tot = 0;
        for (int w = 0; w < (int)(blockDim.x >> 6); ++w) {
            int s = lds[w];
            if (w < wave) wave_off += s;
            tot += s;
        }
        int carry = carry_s;
        if (i < nb) sums[i] = carry + wave_off + incl - v;
        __syncthreads();
        if (threadIdx.x == 0) carry_s = carry + tot;
        __syncthreads();
    }
    if (threadIdx.x == 0 && total_out) *total_out = carry_s;
}

__global__ __launch_bounds__(SCAN_THREADS) void k_scan_apply(const int* __restrict__ in, int* __restrict__ out,
                                                              long long n, const int* __restrict__ sums) {
    __shared__ int lds[8];
    const long long base = (long long)blockIdx.x * SCAN_TILE + (long long)threadIdx.x * SCAN_ITEMS;
    int v[SCAN_ITEMS];
    int s = 0;
#pragma unroll
    for (int k = 0; k < SCAN_ITEMS; ++k) {
        long long i = base + k;
        v[k] = (i < n) ? in[i] : 0;
        s += v[k];
    }
    int tot;
    int off = block_exclusive_scan(s, &tot, lds) + sums[blockIdx.x];
#pragma unroll
    for (int k = 0; k < SCAN_ITEMS; ++k) {
        long long i = base + k;
        if (i < n) out[i] = off;
        off += v[k];
    }
}

int exclusive_scan_i32(icp_ctx* ctx, const int* in, int* out, int64_t n, int* total_dev) {
    if (n <= 0) {
        if (total_dev) ICP_HIP(ctx, hipMemsetAsync(total_dev, 0, sizeof(int), ctx->stream));
        return ICP_OK;
    }
    const int nb = (int)((n + SCAN_TILE - 1) / SCAN_TILE);
    ICP_HIP(ctx, ctx->scan_tmp.reserve((size_t)nb * sizeof(int)));
    int* sums = ctx->scan_tmp.as<int>();
    hipLaunchKernelGGL(k_scan_tile_sums, dim3(nb), dim3(SCAN_THREADS), 0, ctx->stream, in, (long long)n, sums);
    hipLaunchKernelGGL(k_scan_sums, dim3(1), dim3(1024), 0, ctx->stream, sums, nb, total_dev);
    hipLaunchKernelGGL(k_scan_apply, dim3(nb), dim3(SCAN_THREADS), 0, ctx->stream, in, out, (long long)n, sums);
    ICP_HIP(ctx, hipGetLastError());
    return ICP_OK;
}

// ---------------------------------------------------------------------------------------------------------------------
// ordered compaction of rows
// ---------------------------------------------------------------------------------------------------------------------
__global__ void k_compact_scatter(const float* __restrict__ in, const int* __restrict__ flags,
                                  const int* __restrict__ offs, long long n, int row_floats, float* __restrict__ out,
                                  long long cap) {
    long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n || !flags[i]) return;
    long long o = offs[i];
    if (o >= cap) return;  // (an output of bounded capacity: icp_compact_targets)
    for (int c = 0; c < row_floats; ++c) out[o * row_floats + c] = in[i * row_floats + c];
}

int compact_rows(icp_ctx* ctx, const float* in, const int* flags, int64_t n, int row_floats, float* out,
                 int* count_dev, int64_t cap) {
    ICP_HIP(ctx, ctx->scan_a.reserve((size_t)(n > 0 ? n : 1) * sizeof(int)));
    int* offs = ctx->scan_a.as<int>();
    int rc = exclusive_scan_i32(ctx, flags, offs, n, count_dev);
    if (rc) return rc;
    if (n > 0) {
        hipLaunchKernelGGL(k_compact_scatter, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, ctx->stream, in, flags,
                           offs, (long long)n, row_floats, out, (long long)(cap >= 0 ? cap : n));
        ICP_HIP(ctx, hipGetLastError());
    }
    return ICP_OK;
}

// ---------------------------------------------------------------------------------------------------------------------
// ordered compaction of the VALID rows of an [n,3] cloud (no NaN; not null under skip_null) in two launches (round 5: the
// generic form above — flags, three scan launches, scatter, count copy — was seven launches on the critical path of every
// frame of the device-resident pipeline): tile counts, then every tile sums the counts of the tiles in front of it (a few
// hundred integers), scans its own flags and scatters.  The last tile leaves the total in *count_dev and, when given, in
// *count_host (pinned host memory mapped into the device: no copy launch).
// ---------------------------------------------------------------------------------------------------------------------
static constexpr int CV_THREADS = 256, CV_ROUNDS = 4, CV_TILE = CV_THREADS * CV_ROUNDS;

__device__ inline int row_is_valid(const float* __restrict__ xyz, long long i, long long n, int skip_null) {
    if (i >= n) return 0;
    const float x = xyz[3 * i], y = xyz[3 * i + 1], z = xyz[3 * i + 2];
    if (!(x == x && y == y && z == z)) return 0;
    return (skip_null && x == 0.f && y == 0.f && z == 0.f) ? 0 : 1;
}

__global__ __launch_bounds__(CV_THREADS) void k_valid_tile_counts(const float* __restrict__ xyz, long long n,
                                                                  int skip_null, int* __restrict__ counts) {
    __shared__ int lds[8];
    const long long base = (long long)blockIdx.x * CV_TILE + threadIdx.x;
    int s = 0;
#pragma unroll
    for (int k = 0; k < CV_ROUNDS; ++k) s += row_is_valid(xyz, base + (long long)k * CV_THREADS, n, skip_null);
    int tot;
    block_exclusive_scan(s, &tot, lds);
    if (threadIdx.x == 0) counts[blockIdx.x] = tot;
}

__global__ __launch_bounds__(CV_THREADS) void k_valid_compact(const float* __restrict__ xyz, long long n, int skip_null,
                                                              const int* __restrict__ counts, float* __restrict__ out,
                                                              long long cap, int* __restrict__ count_dev,
                                                              int* __restrict__ count_host) {
    __shared__ int lds[8];
    int before = 0;
    for (int t = threadIdx.x; t < (int)blockIdx.x; t += CV_THREADS) before += counts[t];
    int running;
    block_exclusive_scan(before, &running, lds);  // (total of the tiles in front)
    const long long base = (long long)blockIdx.x * CV_TILE + threadIdx.x;
#pragma unroll
    for (int k = 0; k < CV_ROUNDS; ++k) {  // rows base + k * 256 + thread: round by round in row order
        const long long i = base + (long long)k * CV_THREADS;
        const int f = row_is_valid(xyz, i, n, skip_null);
        int tot;
        const long long o = running + block_exclusive_scan(f, &tot, lds);
        if (f && o < cap) {
            out[3 * o] = xyz[3 * i];
            out[3 * o + 1] = xyz[3 * i + 1];
            out[3 * o + 2] = xyz[3 * i + 2];
        }
        running += tot;
    }
    if (blockIdx.x == gridDim.x - 1 && threadIdx.x == 0) {
        if (count_dev) *count_dev = running;
        if (count_host) __hip_atomic_store(count_host, running, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}

int compact_valid_rows(icp_ctx* ctx, const float* xyz, int64_t n, bool skip_null, float* out, int* count_dev,
                       int64_t cap, int* count_host_mapped) {
    if (n <= 0) {
        if (count_dev) ICP_HIP(ctx, hipMemsetAsync(count_dev, 0, sizeof(int), ctx->stream));
        return ICP_OK;  // (*count_host: the caller's zero stands)
    }
    const int nb = (int)((n + CV_TILE - 1) / CV_TILE);
    ICP_HIP(ctx, ctx->scan_tmp.reserve((size_t)nb * sizeof(int)));
    int* counts = ctx->scan_tmp.as<int>();
    hipLaunchKernelGGL(k_valid_tile_counts, dim3(nb), dim3(CV_THREADS), 0, ctx->stream, xyz, (long long)n,
                       skip_null ? 1 : 0, counts);
    hipLaunchKernelGGL(k_valid_compact, dim3(nb), dim3(CV_THREADS), 0, ctx->stream, xyz, (long long)n, skip_null ? 1 : 0,
                       counts, out, (long long)(cap >= 0 ? cap : n), count_dev, count_host_mapped);
    ICP_HIP(ctx, hipGetLastError());
    return ICP_OK;
}

// ---------------------------------------------------------------------------------------------------------------------
// grid build
// ---------------------------------------------------------------------------------------------------------------------
// + (seed_n > 0) the neighbours the last registration left in the NN cache -> original map indices, shifted by the
// `evicted` oldest points this update drops: the seeds of the next frame's first iteration.  Must run while the old
// cell-sorted positions still mean something, i.e. before the scatter of this build: it shares the first launch.
// + (move.m > 0) the re-expression of the kept map points in the new frame (local_map.py:346-348): independent of the
// other two, one launch less per frame.
// + (carry_m > 0: a pose-only update, option "carry_normals") the normals the old grid holds, rotated into the new frame
// like the points (n' = R^-1 n; re-normalised, so that a thousand pose-only updates in a row leave unit vectors) and filed
// by ORIGINAL index: the scatter of this build puts them at the points' new cell-sorted positions instead of zeros.
// CELL LISTS (round 6; VERDICT r3-r5: "k_grid_scan walks a 1.25 M-slot table to find 8 k occupied cells"): every table slot a
// build claims is appended to a list (k_grid_insert2: fine and coarse level apart, one counter update per WORKGROUP); the next
// build empties exactly those slots instead of the whole table, and the prefix sums that turn the cells' counts into their
// starts run over the lists (k_cells_scan) — 9 000 entries instead of 262 144 slots at the headline sizes.  Two list sets in
// alternation (`lists.prev` was written by the previous build).  The ORDER of the cells in the cell-sorted arrays becomes the
// order of the claims — as immaterial as the order of the points inside a cell already was: every consumer breaks ties on the
// original index, every sum over a neighbourhood is order-independent.
__device__ __forceinline__ void grid_clear_body(GridEntry* __restrict__ table, unsigned int size, const int4* __restrict__ nn_cache,
                             const float4* __restrict__ old_pts, int seed_n, int old_m, int evicted,
                             int* __restrict__ seed, const MapMoveJob& move, int* __restrict__ scan_ticket,
                             unsigned long long* __restrict__ hood_used, const float4* __restrict__ old_normals,
                             const int* __restrict__ old_nflag, int carry_m, float4* __restrict__ carry, const CellLists& lists,
                             const unsigned bx) {
    __shared__ float T[16];
    if (bx == 0 && threadIdx.x == 0) {
        *scan_ticket = 0;               // the tile numbers of this build's k_grid_scan
        if (hood_used) *hood_used = 0;  // the list space k_hood_build hands out (a memset launch of its own cost 5 us)
        if (lists.counts) lists.counts[0] = lists.counts[1] = 0;  // this build's lists start empty (nobody reads them here)
    }
    const long long first = (long long)bx * blockDim.x;
    const bool moves = first < move.m;  // block-uniform (a carry job comes with a move job over the same points)
    if (moves && threadIdx.x == 0) map_move_prepare(move, T);
    unsigned int i = bx * blockDim.x + threadIdx.x;
    {
        GridEntry e;
        e.key = GRID_EMPTY;
        e.start = 0;
        e.count = 0;
        if (lists.prev_counts) {  // the slots the previous build claimed, nothing else
            if ((int)i < lists.prev_counts[0]) table[lists.prev_fine[i]] = e;
            if ((int)i < lists.prev_counts[1]) table[lists.prev_coarse[i]] = e;
        } else if (i < size) {
            table[i] = e;
        }
    }
    if ((int)i < seed_n) {
        const int x = nn_cache[i].x, pos = x & 0xffffff;  // (.x = position | iteration << 24, search.hip)
        int o = -1;
        if (x >= 0 && pos < old_m) o = __float_as_int(old_pts[pos].w) - evicted;
        seed[i] = o;
    }
    if (moves) {
        __syncthreads();
        if ((long long)i < move.m) move_point(T, move.in, (long long)i, move.out);
        if ((int)i < carry_m) {  // (carry_m == move.m: this block has T)
            float4 out = make_float4(0.f, 0.f, 0.f, 0.f);
            const float4 p = old_pts[i];
            if (old_nflag[i] == 1) {
                const float4 nv = old_normals[i];
                const float x = T[0] * nv.x + T[1] * nv.y + T[2] * nv.z, y = T[4] * nv.x + T[5] * nv.y + T[6] * nv.z,
                            z = T[8] * nv.x + T[9] * nv.y + T[10] * nv.z;
                // (re-normalised only once rounding has moved the length off 1 by more than an ulp or two: a rotation by the
                // identity — the update behind a registration that failed — leaves every bit where it was)
                const float n2 = x * x + y * y + z * z;
                const float sc = fabsf(n2 - 1.f) > 4e-7f ? rsqrtf(n2) : 1.f;
                if (n2 > 0.f && sc < INFINITY) out = make_float4(x * sc, y * sc, z * sc, 1.f);  // (a zero normal stays unestimated)
            }
            const int o = __float_as_int(p.w);
            if (o >= 0 && o < carry_m) carry[o] = out;
        }
    }
}

__global__ void k_grid_clear(GridEntry* __restrict__ table, unsigned int size, const int4* __restrict__ nn_cache,
                             const float4* __restrict__ old_pts, int seed_n, int old_m, int evicted,
                             int* __restrict__ seed, MapMoveJob move, int* __restrict__ scan_ticket,
                             unsigned long long* __restrict__ hood_used, const float4* __restrict__ old_normals,
                             const int* __restrict__ old_nflag, int carry_m, float4* __restrict__ carry, CellLists lists) {
    grid_clear_body(table, size, nn_cache, old_pts, seed_n, old_m, evicted, seed, move, scan_ticket, hood_used, old_normals,
                    old_nflag, carry_m, carry, lists, blockIdx.x);
}

// rows[cell][c] = (start, count) of the neighbour cell c of every occupied cell (0,0 if that neighbour is empty)
__device__ inline void build_rows_part(const GridEntry* __restrict__ table, unsigned int mask,
                                       const int* __restrict__ slot_of_cell, const int* __restrict__ ncells_dev,
                                       int2* __restrict__ rows, int block, int blocks) {
    const long long total = (long long)(*ncells_dev) * 27;
    for (long long t = (long long)block * blockDim.x + threadIdx.x; t < total; t += (long long)blocks * blockDim.x) {
        const int j = (int)(t / 27), c = (int)(t % 27);
        const GridEntry own = table[slot_of_cell[j]];
        int2 out = make_int2(0, 0);
        if (c == 13) {
            out = make_int2(own.start, own.count);
        } else {
            const int cx = (int)(own.key & 0x1FFFFFull) - CELL_OFFSET, cy = (int)((own.key >> 21) & 0x1FFFFFull) - CELL_OFFSET,
                      cz = (int)((own.key >> 42) & 0x1FFFFFull) - CELL_OFFSET;
            const unsigned long long key = pack_cell(cx + c % 3 - 1, cy + (c / 3) % 3 - 1, cz + c / 9 - 1);
            unsigned int slot = hash_cell(key) & mask;
            while (true) {
                const GridEntry e = table[slot];
                if (e.key == key) {
                    out = make_int2(e.start, e.count);
                    break;
                }
                if (e.key == GRID_EMPTY) break;
                slot = (slot + 1) & mask;
            }
        }
        // rows are indexed by the table SLOT of their cell: a query fetches entry and row together (no row-id hop)
        const size_t base = (size_t)slot_of_cell[j] * ROW_STRIDE;
        rows[base + c] = out;
        if (c == 26) rows[base + 27] = make_int2(0, 0);  // padding entry
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// Fused build of BOTH levels: the fine and the coarse table live back to back in one array ([0,T) fine, [T,2T) coarse),
// so clearing, inserting, the count scan and the scatter (with the neighbour rows) are one launch each: 4 per rebuild.
// One 64-bit scan carries two sums at once: low word = points before the slot (cell start; the fine level holds exactly
// m points, so coarse starts are that prefix minus m), high word = occupied FINE cells before the slot (row id).
// ---------------------------------------------------------------------------------------------------------------------
// (the occupied cells are counted by the scan, not here: one atomic counter for ten thousand claims was two thirds of the
// insertion kernel — 31 us)
// *fresh: this call put the key into an empty slot (exactly one caller per key and build sees that)
__device__ inline unsigned int grid_claim(GridEntry* __restrict__ table, unsigned int mask, unsigned long long key, bool* fresh) {
    unsigned int slot = hash_cell(key) & mask;
    while (true) {
        unsigned long long old = atomicCAS(&table[slot].key, GRID_EMPTY, key);
        if (old == GRID_EMPTY || old == key) {
            *fresh = old == GRID_EMPTY;
            break;
        }
        slot = (slot + 1) & mask;
    }
    return slot;
}

// lanes of a wave that hold the same key form a group: `leader` = its lowest lane, `rank` = position of this lane in the
// group, `size` = lanes in the group (ballots only, no memory traffic; one trip per distinct key in the wave)
__device__ inline void wave_group_by_key(unsigned long long key, bool active, int& leader, int& rank, int& size) {
    const int lane = threadIdx.x & 63;
    unsigned long long todo = __ballot(active);
    leader = lane;
    rank = 0;
    size = 1;
    while (todo) {
        const int l = __ffsll((long long)todo) - 1;
        const unsigned lo = __shfl((unsigned)(key & 0xffffffffull), l, 64), hi = __shfl((unsigned)(key >> 32), l, 64);
        const unsigned long long k = ((unsigned long long)hi << 32) | lo;
        const unsigned long long same = __ballot(active && key == k);
        if (active && key == k) {
            leader = l;
            rank = __popcll(same & ((1ull << lane) - 1ull));
            size = __popcll(same);
        }
        todo &= ~same;
    }
}

// Consecutive map points are spatial neighbours (scan order), so the lanes of a wave share a handful of cells — and, on the
// coarse level, one or two: every point claiming its cell and bumping the cell's counter by itself put hundreds of
// same-address atomics in a row (31 us for 100 000 points).  One claim and one counter update per (wave, cell) instead:
// the group leader adds the group's size and every lane takes base + its rank.
// Round 5 ("insert_by_cell"): the points of a map that has just been re-expressed are claimed in the order of the PREVIOUS
// grid (`old_pts`: its cell-sorted points, .w = original index; kept points have the index minus `evicted`, the inserted ones
// follow in their own order): a rigid step of a few centimetres leaves the points of an old cell in one or two new cells, so
// the lanes of a wave share a handful of keys again — in insertion order a map merged from many grid-sampled clouds has 64
// different cells per wave (64 trips of the grouping loop, 64 claims, 64 counter updates: 16.9 us at C2, 24 us at 181 695
// points).  Which thread claims a point does not matter: ranks inside a cell come from atomics either way.
__device__ __forceinline__ void grid_insert2_body(const float* __restrict__ xyz, int m, float inv_h, float inv_hc,
                               GridEntry* __restrict__ table, unsigned int tsize, int* __restrict__ slot_of,
                               int* __restrict__ rank_of, int* __restrict__ cslot_of, int* __restrict__ crank_of,
                               const float4* __restrict__ old_pts, int old_m, int evicted, int kept,
                               int* __restrict__ visit, const CellLists& lists, const unsigned bx) {
    // (with `old_pts` the four outputs are indexed by THREAD and visit[thread] names the point, -1: none — the scatter walks
    // the same order, so that its stores by new position are clustered as well)
    const int t = bx * blockDim.x + threadIdx.x;
    int i = t;
    bool active = t < m;
    if (old_pts) {
        if (t < old_m) {
            i = __float_as_int(old_pts[t].w) - evicted;
            active = i >= 0 && i < kept;
        } else {
            i = kept + (t - old_m);
            active = i < m;
        }
    }
    float x = 0.f, y = 0.f, z = 0.f;
    if (active) {
        x = xyz[3 * i + 0];
        y = xyz[3 * i + 1];
        z = xyz[3 * i + 2];
    }
    const unsigned long long key = pack_cell(cell_coord(x, inv_h), cell_coord(y, inv_h), cell_coord(z, inv_h));
    const unsigned long long ckey = pack_cell(cell_coord(x, inv_hc), cell_coord(y, inv_hc), cell_coord(z, inv_hc));
    int leader, rank, size, cleader, crank, csize;
    wave_group_by_key(key, active, leader, rank, size);
    wave_group_by_key(ckey, active, cleader, crank, csize);
    const int lane = threadIdx.x & 63;
    int slot = 0, base = 0, cslot = 0, cbase = 0;
    bool fresh = false, cfresh = false;
    if (active && leader == lane) slot = (int)grid_claim(table, tsize - 1, key, &fresh);
    if (active && cleader == lane) cslot = (int)(tsize + grid_claim(table + tsize, tsize - 1, ckey, &cfresh));
    if (active && leader == lane) base = atomicAdd(&table[slot].count, size);  // the two requests are in flight together
    if (active && cleader == lane) cbase = atomicAdd(&table[cslot].count, csize);
    if (lists.counts) {
        // the slots this workgroup has just claimed, appended to the build's cell lists: collected in LDS, ONE counter update
        // per workgroup and level (a counter update per claim was two thirds of an earlier insertion kernel)
        __shared__ int nf_s, nc_s, bf_s, bc_s;
        __shared__ int lf_s[256], lc_s[256];
        if (threadIdx.x == 0) nf_s = nc_s = 0;
        __syncthreads();
        if (fresh) lf_s[atomicAdd(&nf_s, 1)] = slot;
        if (cfresh) lc_s[atomicAdd(&nc_s, 1)] = cslot;
        __syncthreads();
        if (threadIdx.x == 0) {
            bf_s = nf_s ? atomicAdd(&lists.counts[0], nf_s) : 0;
            bc_s = nc_s ? atomicAdd(&lists.counts[1], nc_s) : 0;
        }
        __syncthreads();
        if ((int)threadIdx.x < nf_s) lists.fine[bf_s + threadIdx.x] = lf_s[threadIdx.x];
        if ((int)threadIdx.x < nc_s) lists.coarse[bc_s + threadIdx.x] = lc_s[threadIdx.x];
    }
    slot = __shfl(slot, leader, 64);
    base = __shfl(base, leader, 64);
    cslot = __shfl(cslot, cleader, 64);
    cbase = __shfl(cbase, cleader, 64);
    if (visit) {
        const int total = old_m + (m - kept);
        if (t < total) visit[t] = active ? i : -1;
        if (!active) return;
        slot_of[t] = slot;
        rank_of[t] = base + rank;
        cslot_of[t] = cslot;
        crank_of[t] = cbase + crank;
        return;
    }
    if (!active) return;
    slot_of[i] = slot;
    rank_of[i] = base + rank;
    cslot_of[i] = cslot;
    crank_of[i] = cbase + crank;
}

__global__ void k_grid_insert2(const float* __restrict__ xyz, int m, float inv_h, float inv_hc,
                               GridEntry* __restrict__ table, unsigned int tsize, int* __restrict__ slot_of,
                               int* __restrict__ rank_of, int* __restrict__ cslot_of, int* __restrict__ crank_of,
                               const float4* __restrict__ old_pts, int old_m, int evicted, int kept,
                               int* __restrict__ visit, CellLists lists) {
    grid_insert2_body(xyz, m, inv_h, inv_hc, table, tsize, slot_of, rank_of, cslot_of, crank_of, old_pts, old_m, evicted, kept,
                      visit, lists, blockIdx.x);
}

// The starts of the cells from their counts, over the build's CELL LISTS: workgroup 0 the fine level, workgroup 1 the coarse
// level, each an exclusive scan of its list in chunks of CS_THREADS x CS_ITEMS entries (8 192: one or two chunks at the headline
// sizes; a 1M-point map has ~80 000 cells: ten chunks).  Both levels hold all m points, so both scans run from 0.
static constexpr int CS_THREADS = 1024, CS_ITEMS = 8;

__device__ __forceinline__ void cells_scan_body(GridEntry* __restrict__ table, const CellLists& lists, int* __restrict__ ncells_out,
                                                const int level) {
    __shared__ int wsum[CS_THREADS / 64];
    __shared__ int carry_s;
    const int* __restrict__ list = level == 0 ? lists.fine : lists.coarse;
    const int n = lists.counts[level];
    if (level == 0 && threadIdx.x == 0) *ncells_out = n;
    if (threadIdx.x == 0) carry_s = 0;
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int base = 0; base < n; base += CS_THREADS * CS_ITEMS) {  // block-uniform
        int slot[CS_ITEMS], cnt[CS_ITEMS], s = 0;
#pragma unroll
        for (int k = 0; k < CS_ITEMS; ++k) {  // (consecutive entries per thread: the scan order is the list order)
            const int i = base + (int)threadIdx.x * CS_ITEMS + k;
            slot[k] = i < n ? list[i] : -1;
        }
#pragma unroll
        for (int k = 0; k < CS_ITEMS; ++k) {
            cnt[k] = slot[k] >= 0 ? table[slot[k]].count : 0;
            s += cnt[k];
        }
        int incl = s;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const int t = __shfl_up(incl, o, 64);
            if (lane >= o) incl += t;
        }
        if (lane == 63) wsum[wave] = incl;
        __syncthreads();
        int woff = 0, tot = 0;
#pragma unroll
        for (int w = 0; w < CS_THREADS / 64; ++w) {
            const int v = wsum[w];
            if (w < wave) woff += v;
            tot += v;
        }
        int off = carry_s + woff + incl - s;
#pragma unroll
        for (int k = 0; k < CS_ITEMS; ++k) {
            if (slot[k] >= 0) table[slot[k]].start = off;
            off += cnt[k];
        }
        __syncthreads();
        if (threadIdx.x == 0) carry_s += tot;
        __syncthreads();
    }
}

__global__ __launch_bounds__(CS_THREADS) void k_cells_scan(GridEntry* __restrict__ table, CellLists lists, int* __restrict__ ncells_out) {
    cells_scan_body(table, lists, ncells_out, (int)blockIdx.x);
}

__device__ inline unsigned long long grid_scan_item(const GridEntry* __restrict__ table, long long i, long long n,
                                                    unsigned int tsize) {
    if (i >= n) return 0ull;
    const int c = table[i].count;
    return (unsigned long long)(unsigned)c | ((i < (long long)tsize && c > 0) ? (1ull << 32) : 0ull);
}

__device__ inline unsigned long long block_exclusive_scan_u64(unsigned long long v, unsigned long long* total,
                                                              unsigned long long* lds /* >= 16 */) {
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    unsigned long long incl = v;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const unsigned lo = __shfl_up((unsigned)(incl & 0xffffffffull), o, 64);
        const unsigned hi = __shfl_up((unsigned)(incl >> 32), o, 64);
        if (lane >= o) incl += ((unsigned long long)hi << 32) | lo;
    }
    if (lane == 63) lds[wave] = incl;
    __syncthreads();
    unsigned long long wave_off = 0, tot = 0;
    const int nw = (blockDim.x + 63) >> 6;
    for (int w = 0; w < nw; ++w) {
        const unsigned long long s = lds[w];
        if (w < wave) wave_off += s;
        tot += s;
    }
    __syncthreads();
    *total = tot;
    return wave_off + incl - v;
}

// ---------------------------------------------------------------------------------------------------------------------
// The scan of the table in ONE launch (decoupled look-back): a tile of SCAN_TILE slots per workgroup; the workgroup sums
// its tile, publishes the sum, and its first wave walks back over the predecessors' descriptors — 64 at a time — until
// it meets one that already knows its inclusive prefix; then it publishes its own and applies the offsets to the counts
// it still holds in registers.  (Three launches before: tile sums, scan of the sums by one workgroup, apply — the table
// read twice and two kernel boundaries, 17.8 us per rebuild of a 100 000-point map.)
// A descriptor is two 64-bit words, each (tag << 32 | 32-bit sum): A carries the point count, B the count of occupied
// fine cells; tag = generation of this build << 2 | state (1: sum of the tile, 2: inclusive prefix).  Each word is written
// and read by ONE relaxed agent-scope atomic, so no fence orders them: a reader takes a descriptor only when both tags
// agree.  Stale descriptors of earlier builds carry another generation (the descriptors are zeroed when the generation
// counter wraps).  A workgroup takes its tile number from a ticket counter (reset by k_grid_clear), so it only ever waits
// for tiles whose workgroups are already running, whatever order the hardware dispatches blockIdx in.
// ---------------------------------------------------------------------------------------------------------------------
static constexpr unsigned SCAN_STATE_SUM = 1, SCAN_STATE_PREFIX = 2;

__device__ inline void scan_publish(unsigned long long* __restrict__ desc_a, unsigned long long* __restrict__ desc_b,
                                    int tile, unsigned long long value, unsigned gen, unsigned state) {
    const unsigned long long tag = (unsigned long long)((gen << 2) | state) << 32;
    __hip_atomic_store(&desc_a[tile], tag | (value & 0xffffffffull), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_store(&desc_b[tile], tag | (value >> 32), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// waits for the descriptor of `tile` of this generation; false after `poll_limit` polls (2^20 in production: never seen;
// the caller then sums the table itself instead of hanging the GPU — the option "scan_poll_limit" lets a test go there)
__device__ inline bool scan_wait(const unsigned long long* __restrict__ desc_a,
                                 const unsigned long long* __restrict__ desc_b, int tile, unsigned gen, int poll_limit,
                                 unsigned long long& value, unsigned& state) {
    for (int polls = 0; polls < poll_limit; ++polls) {
        const unsigned long long a = __hip_atomic_load(&desc_a[tile], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const unsigned long long b = __hip_atomic_load(&desc_b[tile], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const unsigned ta = (unsigned)(a >> 32), tb = (unsigned)(b >> 32);
        if (ta == tb && (ta >> 2) == gen && (ta & 3u) != 0u) {
            state = ta & 3u;
            value = ((b & 0xffffffffull) << 32) | (a & 0xffffffffull);
            return true;
        }
        __builtin_amdgcn_s_sleep(1);
    }
    return false;
}

__device__ __forceinline__ void grid_scan_body(GridEntry* __restrict__ table, long long n,
                                                            unsigned int tsize, int m,
                                                            unsigned long long* __restrict__ desc_a,
                                                            unsigned long long* __restrict__ desc_b, unsigned gen,
                                                            int poll_limit, int* __restrict__ slot_of_cell,
                                                            int* __restrict__ ncells_out, int* __restrict__ ticket,
                                                            const int tiles) {
    __shared__ unsigned long long lds[16];
    __shared__ unsigned long long prefix_s;
    __shared__ int gave_up;
    __shared__ int tile_s;
    if (threadIdx.x == 0) {
        tile_s = atomicAdd(ticket, 1);
        gave_up = 0;
    }
    __syncthreads();
    const int tile = tile_s;
    const long long base = (long long)tile * SCAN_TILE + (long long)threadIdx.x * SCAN_ITEMS;
    unsigned long long v[SCAN_ITEMS];
    unsigned long long s = 0;
#pragma unroll
    for (int k = 0; k < SCAN_ITEMS; ++k) {
        v[k] = grid_scan_item(table, base + k, n, tsize);
        s += v[k];
    }
    unsigned long long tot;
    const unsigned long long excl = block_exclusive_scan_u64(s, &tot, lds);  // (two barriers inside)
    if (threadIdx.x < 64) {
        const int lane = threadIdx.x;
        unsigned long long prefix = 0;
        if (tile == 0) {
            if (lane == 0) scan_publish(desc_a, desc_b, 0, tot, gen, SCAN_STATE_PREFIX);
        } else {
            if (lane == 0) scan_publish(desc_a, desc_b, tile, tot, gen, SCAN_STATE_SUM);
            bool failed = false;
            for (int hi = tile - 1;; hi -= 64) {  // wave-uniform
                const int idx = hi - lane;
                unsigned long long val = 0;
                unsigned state = SCAN_STATE_PREFIX;  // lanes in front of tile 0: a prefix of nothing ends the walk
                bool ok = true;
                if (idx >= 0) ok = scan_wait(desc_a, desc_b, idx, gen, poll_limit, val, state);
                if (__ballot(!ok)) {
                    failed = true;
                    break;
                }
                const unsigned long long ends = __ballot(state == SCAN_STATE_PREFIX);
                const int stop = ends ? __ffsll((long long)ends) - 1 : 64;  // nearest predecessor that knows its prefix
                unsigned long long part = lane <= stop ? val : 0ull;
#pragma unroll
                for (int o = 1; o < 64; o <<= 1) part += __shfl_xor(part, o, 64);
                prefix += part;
                if (stop < 64) break;
            }
            if (failed) {  // never seen: a predecessor did not publish in ~a second — the prefix from the table itself
                if (lane == 0) gave_up = 1;
            } else if (lane == 0) {
                scan_publish(desc_a, desc_b, tile, prefix + tot, gen, SCAN_STATE_PREFIX);
            }
        }
        if (lane == 0) prefix_s = prefix;
    }
    __syncthreads();
    if (gave_up) {  // block-uniform
        unsigned long long mine = 0;
        for (long long i = threadIdx.x; i < (long long)tile * SCAN_TILE; i += SCAN_THREADS)
            mine += grid_scan_item(table, i, n, tsize);
        unsigned long long all;
        block_exclusive_scan_u64(mine, &all, lds);
        if (threadIdx.x == 0) {
            prefix_s = all;
            scan_publish(desc_a, desc_b, tile, all + tot, gen, SCAN_STATE_PREFIX);
        }
        __syncthreads();
    }
    unsigned long long off = excl + prefix_s;
    if (threadIdx.x == 0 && tile == tiles - 1) *ncells_out = (int)((prefix_s + tot) >> 32);
#pragma unroll
    for (int k = 0; k < SCAN_ITEMS; ++k) {
        const long long i = base + k;
        if (i < n) {
            const int start = (int)(unsigned)(off & 0xffffffffull);
            if (i < (long long)tsize) {
                table[i].start = start;
                if (v[k] >> 32) slot_of_cell[(int)(off >> 32)] = (int)i;  // dense list of the occupied fine cells
            } else {
                table[i].start = start - m;  // the fine level holds exactly m points
            }
        }
        off += v[k];
    }
}

__global__ __launch_bounds__(SCAN_THREADS) void k_grid_scan(GridEntry* __restrict__ table, long long n,
                                                            unsigned int tsize, int m,
                                                            unsigned long long* __restrict__ desc_a,
                                                            unsigned long long* __restrict__ desc_b, unsigned gen,
                                                            int poll_limit, int* __restrict__ slot_of_cell,
                                                            int* __restrict__ ncells_out, int* __restrict__ ticket) {
    grid_scan_body(table, n, tsize, m, desc_a, desc_b, gen, poll_limit, slot_of_cell, ncells_out, ticket, (int)gridDim.x);
}

// `visit` (k_grid_insert2's order, "insert_by_cell"): thread t handles point visit[t] (-1: none), its claims are filed by t
__device__ inline void grid_scatter_part(int t, const float* __restrict__ xyz, int m, const GridEntry* __restrict__ table,
                                const int* __restrict__ slot_of, const int* __restrict__ rank_of,
                                const int* __restrict__ cslot_of, const int* __restrict__ crank_of,
                                float4* __restrict__ sorted,
                                float4* __restrict__ csorted, float4* __restrict__ normals, int* __restrict__ nflag,
                                int* __restrict__ row_of_pos, int* __restrict__ pos_of_orig,
                                const float4* __restrict__ carry, int carry_m, const int* __restrict__ visit,
                                int visit_n) {
    // four +inf pads behind the last point: search_ball_lane reads cells in whole groups of four
    if (t == 0)
        for (int k = 0; k < SORTED_PAD; ++k) sorted[m + k] = make_float4(INFINITY, INFINITY, INFINITY, __int_as_float(0x7fffffff));
    int i = t;
    if (visit) {
        if (t >= visit_n) return;
        i = visit[t];
        if (i < 0) return;
    } else if (i >= m) {
        return;
    }
    const int slot = slot_of[t];
    const int pos = table[slot].start + rank_of[t];
    const int cpos = table[cslot_of[t]].start + crank_of[t];
    const float4 p = make_float4(xyz[3 * i + 0], xyz[3 * i + 1], xyz[3 * i + 2], __int_as_float(i));
    row_of_pos[pos] = slot;  // rows are indexed by slot
    pos_of_orig[i] = pos;
    sorted[pos] = p;
    csorted[cpos] = p;
    // the normal cache is cleared on every rebuild (local_map.py:368) — or, behind a pose-only update with
    // "carry_normals", holds the rotated normals of the old grid (k_grid_clear filed them by original index)
    if (carry_m > 0) {  // (kernel-uniform; carry_m == m)
        const float4 c = i < carry_m ? carry[i] : make_float4(0.f, 0.f, 0.f, 0.f);
        normals[pos] = c;
        nflag[pos] = c.w == 1.f ? 1 : 0;
    } else {  // (indexed by position: every position is some point's)
        normals[pos] = make_float4(0.f, 0.f, 0.f, 0.f);
        nflag[pos] = 0;
    }
}


// The neighbour rows and the scatter both need the scanned table and nothing of each other: one launch, the first
// `row_blocks` workgroups build rows (grid-stride over the occupied cells x 27), the others scatter the points.
__device__ __forceinline__ void grid_rows_scatter_body(const float* __restrict__ xyz, int m, const GridEntry* __restrict__ table,
                                    unsigned int mask, const int* __restrict__ slot_of_cell,
                                    const int* __restrict__ ncells_dev, int2* __restrict__ rows, int row_blocks,
                                    const int* __restrict__ slot_of, const int* __restrict__ rank_of,
                                    const int* __restrict__ cslot_of, const int* __restrict__ crank_of,
                                    float4* __restrict__ sorted, float4* __restrict__ csorted,
                                    float4* __restrict__ normals, int* __restrict__ nflag, int* __restrict__ row_of_pos,
                                    int* __restrict__ pos_of_orig, const float4* __restrict__ carry, int carry_m,
                                    const int* __restrict__ visit, int visit_n, const int bx) {
    if (bx < row_blocks) {  // block-uniform
        build_rows_part(table, mask, slot_of_cell, ncells_dev, rows, bx, row_blocks);
        return;
    }
    grid_scatter_part((bx - row_blocks) * blockDim.x + threadIdx.x, xyz, m, table, slot_of, rank_of, cslot_of,
                      crank_of, sorted, csorted, normals, nflag, row_of_pos, pos_of_orig, carry, carry_m, visit, visit_n);
}

__global__ void k_grid_rows_scatter(const float* __restrict__ xyz, int m, const GridEntry* __restrict__ table,
                                    unsigned int mask, const int* __restrict__ slot_of_cell,
                                    const int* __restrict__ ncells_dev, int2* __restrict__ rows, int row_blocks,
                                    const int* __restrict__ slot_of, const int* __restrict__ rank_of,
                                    const int* __restrict__ cslot_of, const int* __restrict__ crank_of,
                                    float4* __restrict__ sorted, float4* __restrict__ csorted,
                                    float4* __restrict__ normals, int* __restrict__ nflag, int* __restrict__ row_of_pos,
                                    int* __restrict__ pos_of_orig, const float4* __restrict__ carry, int carry_m,
                                    const int* __restrict__ visit, int visit_n) {
    grid_rows_scatter_body(xyz, m, table, mask, slot_of_cell, ncells_dev, rows, row_blocks, slot_of, rank_of, cslot_of, crank_of,
                           sorted, csorted, normals, nflag, row_of_pos, pos_of_orig, carry, carry_m, visit, visit_n,
                           (int)blockIdx.x);
}

// ---------------------------------------------------------------------------------------------------------------------
// The four launches of a grid build for B maps at once (icp_batch_map_update, api.hip): blockIdx.y = the map, its arguments
// — exactly those of the four kernels above — in a GridBuildDesc in device memory; a map with fewer workgroups than the
// largest of the batch lets the surplus return.  Same bodies, same results.
// ---------------------------------------------------------------------------------------------------------------------
__global__ void k_grid_clear_batch(const GridBuildDesc* __restrict__ t) {
    const GridBuildDesc& d = t[blockIdx.y];
    if (blockIdx.x >= d.clear_blocks) return;
    grid_clear_body(d.table, d.n2, d.nn_cache, d.old_pts, d.seed_n, d.seed_m, d.seed_evicted, d.seed, d.move, d.scan_ticket,
                    d.hood_used, d.old_normals, d.old_nflag, d.carry_m, d.carry, d.lists, blockIdx.x);
}

__global__ void k_grid_insert2_batch(const GridBuildDesc* __restrict__ t) {
    const GridBuildDesc& d = t[blockIdx.y];
    if (blockIdx.x >= d.visit_blocks) return;
    grid_insert2_body(d.xyz, d.m, d.inv_h, d.inv_hc, d.table, d.tsize, d.slot_of, d.rank_of, d.cslot_of, d.crank_of, d.order_pts,
                      d.order_old_m, d.order_evicted, d.order_kept, d.visit, d.lists, blockIdx.x);
}

__global__ __launch_bounds__(SCAN_THREADS) void k_grid_scan_batch(const GridBuildDesc* __restrict__ t) {
    const GridBuildDesc& d = t[blockIdx.y];
    if (blockIdx.x >= d.scan_blocks || d.lists.counts) return;  // (before the ticket is drawn: the tiles of a map number 0 .. scan_blocks - 1; a member on cell lists is scanned by k_cells_scan_batch)
    grid_scan_body(d.table, (long long)d.n2, d.tsize, d.m, d.desc_a, d.desc_b, d.scan_gen, d.poll_limit, d.slot_of_cell,
                   d.ncells, d.scan_ticket, (int)d.scan_blocks);
}

__global__ __launch_bounds__(CS_THREADS) void k_cells_scan_batch(const GridBuildDesc* __restrict__ t) {
    const GridBuildDesc& d = t[blockIdx.y];
    if (!d.lists.counts || d.clear_blocks == 0) return;  // (a member on the table scan / an empty map)
    cells_scan_body(d.table, d.lists, d.ncells, (int)blockIdx.x);
}

__global__ void k_grid_rows_scatter_batch(const GridBuildDesc* __restrict__ t) {
    const GridBuildDesc& d = t[blockIdx.y];
    if (blockIdx.x >= d.row_blocks + d.visit_blocks) return;
    grid_rows_scatter_body(d.xyz, d.m, d.table, d.tsize - 1, d.slot_of_cell, d.ncells, d.rows, (int)d.row_blocks, d.slot_of,
                           d.rank_of, d.cslot_of, d.crank_of, d.sorted, d.csorted, d.normals, d.nflag, d.row_of_pos,
                           d.pos_of_orig, d.carry, d.carry_m, d.visit, d.visit_n, (int)blockIdx.x);
}

// ---------------------------------------------------------------------------------------------------------------------
// Neighbourhood lists ("hoods"): for every occupied fine cell the points of its 27-neighbourhood, copied END TO END into
// one contiguous run (header = entry 27 of the cell's row: start, count).  The kNN normal estimation then streams ~120
// candidates per map point from one address range — no 27 (start, count) pairs to walk, no per-cell loop state: that
// bookkeeping, not the candidates, was 90 % of the 6 700 VALU instructions a wave of 16 points spent there (round 2's
// counters).  Memory for speed: every point appears in up to 27 lists (19 MB for the 100 000-point map, bounded by
// 27 x 16 B x M), rebuilt with the grid — an HBM3E-sized trade.
// 32 lanes per cell (lane c = neighbour c of the row), 32 cells per workgroup, ONE atomicAdd per workgroup for its share
// of the list space (the order of the runs in memory is immaterial; within a run: neighbour by neighbour, cell order).
// ---------------------------------------------------------------------------------------------------------------------
static constexpr int HOOD_THREADS = 1024;

__global__ __launch_bounds__(HOOD_THREADS) void k_hood_build(const int* __restrict__ slot_of_cell,
                                                             const int* __restrict__ ncells_dev, int2* __restrict__ rows,
                                                             const float4* __restrict__ pts, float4* __restrict__ hood,
                                                             long long capacity, unsigned long long* __restrict__ used) {
    __shared__ int cell_tot[HOOD_THREADS / 32];
    __shared__ unsigned long long base_s;
    const int ncells = *ncells_dev;
    const int j = blockIdx.x * (HOOD_THREADS / 32) + (threadIdx.x >> 5), c = threadIdx.x & 31;
    if (blockIdx.x * (HOOD_THREADS / 32) >= ncells) return;  // block-uniform
    int2 e = make_int2(0, 0);
    size_t row = 0;
    if (j < ncells) {
        row = (size_t)slot_of_cell[j] * ROW_STRIDE;
        if (c < 27) e = rows[row + c];
    }
    const int cnt = e.y;
    int incl = cnt;
#pragma unroll
    for (int o = 1; o < 32; o <<= 1) {
        const int t = __shfl_up(incl, o, 32);
        if (c >= o) incl += t;
    }
    const int tot = __shfl(incl, 31, 32);
    // (a run occupies a whole number of groups of four entries, the tail filled with points at +inf: the one-lane kNN
    // streams groups of four without masking a candidate)
    const int tot4 = (tot + 3) & ~3;
    if (c == 0) cell_tot[threadIdx.x >> 5] = tot4;
    __syncthreads();
    if (threadIdx.x == 0) {
        unsigned long long sum = 0;
        for (int k = 0; k < HOOD_THREADS / 32; ++k) sum += (unsigned long long)cell_tot[k];
        base_s = atomicAdd(used, sum);
    }
    __syncthreads();
    long long start = (long long)base_s;
    for (int k = 0; k < (int)(threadIdx.x >> 5); ++k) start += cell_tot[k];
    if (j >= ncells) return;
    if (start + tot4 > capacity) {  // (cannot happen: the capacity is the 27 M + 3 per cell bound)
        if (c == 0) rows[row + 27] = make_int2(0, 0);
        return;
    }
    if (c == 0) rows[row + 27] = make_int2((int)start, tot);
    // the 32 lanes of the cell copy its run together, entry i by lane i % 32 (coalesced stores; the segment of entry i =
    // the first neighbour whose inclusive count exceeds i, found by a 5-step search over the lanes' counts)
    const int excl = incl - cnt;
    float4* dst = hood + start;
    for (int i = c; i < ((tot + 31) & ~31); i += 32) {  // (same trip count in all 32 lanes: the shuffles below need them)
        int seg = 0;
#pragma unroll
        for (int step = 16; step > 0; step >>= 1) {
            const int v = __shfl(incl, seg + step - 1, 32);
            if (v <= i) seg += step;
        }
        const int seg_start = __shfl(e.x, seg & 31, 32), seg_excl = __shfl(excl, seg & 31, 32);
        if (i < tot) dst[i] = pts[seg_start + (i - seg_excl)];
        else if (i < tot4) dst[i] = make_float4(INFINITY, INFINITY, INFINITY, __int_as_float(0x7fffffff));
    }
}

static unsigned int next_pow2(unsigned int v) {
    unsigned int p = 1024;
    while (p < v) p <<= 1;
    return p;
}

int build_grid(icp_ctx* ctx, GridBuildDesc* defer) {
    const int64_t m = ctx->map_m;
    ctx->grid_valid = false;
    if (m <= 0) {
        ctx->move_job = MapMoveJob();
        ctx->seed_job_n = 0;  // nothing to seed against
        ctx->seed_n = 0;
        return ICP_OK;
    }
    if (m > (int64_t)1 << 30) {
        ctx->error = "local map too large";
        return ICP_ERR_INVALID_ARGUMENT;
    }
    // the pending NN-cache -> seed conversion reads the OLD cell-sorted points: a launch of its own if they are about to be
    // reallocated, otherwise part of the clearing launch below
    if (ctx->seed_job_n > 0 && ctx->sorted_pts.bytes < (size_t)(m + SORTED_PAD) * sizeof(float4)) {
        const int rc = run_seed_job(ctx);
        if (rc) return rc;
    }
    // table slots: the next power of two above 1.25 M (round 3: above 2 M).  A map fills at most one slot per point, so
    // the load stays below 0.8 whatever the cloud (linear probing: a handful of probes even there); a LiDAR map with its
    // ~16 points per cell fills 5 % — and the clearing launch, the one-launch scan and the sparse row array all walk or
    // reserve the WHOLE table every build (8 MB cleared and scanned for 6 000 cells at the headline sizes)
    // (claims filed by thread under "insert_by_cell": one thread per point of the OLD grid and per inserted point)
    int64_t claim_n = m;
    if (ctx->order_job && ctx->insert_by_cell && ctx->order_old_m + (m - ctx->order_kept) > claim_n)
        claim_n = ctx->order_old_m + (m - ctx->order_kept);
    const unsigned int tsize = next_pow2((unsigned int)(m + m / 4));
    ICP_HIP(ctx, ctx->table.reserve((size_t)2 * tsize * sizeof(GridEntry)));  // fine level, then the coarse level
    ICP_HIP(ctx, ctx->csorted.reserve((size_t)m * sizeof(float4)));
    ICP_HIP(ctx, ctx->cslot_of.reserve((size_t)claim_n * sizeof(int)));
    ICP_HIP(ctx, ctx->crank_of.reserve((size_t)claim_n * sizeof(int)));
    const bool sorted_in_place = ctx->sorted_pts.ptr && ctx->sorted_pts.bytes >= (size_t)(m + SORTED_PAD) * sizeof(float4);
    ICP_HIP(ctx, ctx->sorted_pts.reserve((size_t)(m + SORTED_PAD) * sizeof(float4)));
    ICP_HIP(ctx, ctx->normals.reserve((size_t)m * sizeof(float4)));
    ICP_HIP(ctx, ctx->nflag.reserve((size_t)m * sizeof(int)));
    ICP_HIP(ctx, ctx->slot_of.reserve((size_t)claim_n * sizeof(int)));
    ICP_HIP(ctx, ctx->rank_of.reserve((size_t)claim_n * sizeof(int)));
    ICP_HIP(ctx, ctx->worklist.reserve((size_t)claim_n * sizeof(int)));
    ICP_HIP(ctx, ctx->slot_of_cell.reserve((size_t)m * sizeof(int)));
    ICP_HIP(ctx, ctx->row_of_pos.reserve((size_t)m * sizeof(int)));
    ICP_HIP(ctx, ctx->pos_of_orig.reserve((size_t)m * sizeof(int)));
    ICP_HIP(ctx, ctx->rows.reserve((size_t)tsize * ROW_STRIDE * sizeof(int2)));  // one row per table slot (sparse)
    ctx->table_size = tsize;
    GridEntry* table = ctx->table.as<GridEntry>();
    const float* xyz = ctx->map_xyz[ctx->map_cur].as<float>();
    // cell edge: fixed by the configuration, or auto-tuned towards ~4 map points per occupied cell from the occupancy
    // measured on the previous build (surface-like scaling: points per cell ~ h^2)
    if (ctx->cfg.cell_size > 0.f) {
        ctx->cell_h = ctx->cfg.cell_size;
    } else if (ctx->occupied_cells > 0 && ctx->stats_m > 0) {
        // the occupancy figure belongs to the build it was measured on (edge stats_h) — with the asynchronous result
        // hand-off that may be the build before the previous one
        const double mean = (double)ctx->stats_m / (double)ctx->occupied_cells;
        double f = sqrt(ctx->target_occupancy / mean);
        if (f < 0.5) f = 0.5;
        if (f > 2.0) f = 2.0;
        const double wanted = fmin(fmax((double)ctx->stats_h * f, 0.05), 8.0);
        const double change = wanted / (double)ctx->cell_h;
        if (change < 0.85 || change > 1.18) ctx->cell_h = (float)wanted;
    }
    const float inv_h = 1.0f / ctx->cell_h;
    // a pose-only update with "carry_normals" (map_update_impl): the normals of the old grid travel with their points — the
    // old cell-sorted array, normals and flags are read by the clearing launch, the new ones written by the scatter
    int carry_m = 0;
    if (ctx->carry_job && ctx->carry_m == m && ctx->move_job.m == m && ctx->cost == ICP_COST_POINT_TO_PLANE) {
        ICP_HIP(ctx, ctx->normals_carry.reserve((size_t)m * sizeof(float4)));
        carry_m = (int)m;
    }
    ctx->carry_job = false;
    const bool carried_all = carry_m > 0 && ctx->normals_ready;  // every normal was there, every normal still is
    ctx->normals_ready = carried_all;
    ctx->stats_pending = true;
    ctx->stats_m_pending = m;
    ctx->stats_h_pending = ctx->cell_h;
    // (the order of the previous grid for the claims of this one: its cell-sorted points must still be where they were)
    const bool by_cell = ctx->order_job && ctx->insert_by_cell && sorted_in_place && ctx->order_kept <= m &&
                         ctx->order_old_m > 0 && ctx->order_old_m + (m - ctx->order_kept) < (1ll << 30);
    ctx->order_job = false;
    const long long n2 = 2ll * tsize;  // both levels
    const int nb = (int)((n2 + SCAN_TILE - 1) / SCAN_TILE);
    // descriptors of the one-launch scan: [nb] point-count words, then [nb] cell-count words; a fresh allocation is
    // zeroed (tag 0 belongs to no build), later builds tell their descriptors from stale ones by the generation
    // (the first 64 bytes of the buffer hold the ticket counter the tiles draw their numbers from)
    const unsigned scan_gen = (unsigned)((ctx->scan_builds++ % 0x3ffffffeull) + 1ull);  // 1 .. 2^30 - 2, a new one per launch
    {
        const size_t need = 64 + (size_t)2 * nb * sizeof(unsigned long long);
        const bool wrapped = scan_gen == 1u && ctx->scan_builds > 1ull;  // generations start over: no stale tag may match
        if (ctx->scan_desc.bytes < need || wrapped) {
            ICP_HIP(ctx, ctx->scan_desc.reserve(need));
            ICP_HIP(ctx, hipMemsetAsync(ctx->scan_desc.ptr, 0, ctx->scan_desc.bytes, ctx->stream));
        }
    }
    // (the neighbourhood lists are allocated before the first launch of the build: their space counter is zeroed by it)
    // ... and only where something will read them: the eager estimation of point-to-plane normals with k = 5 or 10
    // neighbours (k_normals_hood2 / k_normals_hood), by the library's own schedule or map-sharded through
    // icp_map_normals_owned — a point-to-point loop, a lazily estimated map or another k never does (480 B per map point
    // and a launch per build otherwise); the list space of a map that stops needing it is given back
    const int kn_hood = ctx->cfg.num_neighbors_normals + 1;
    const bool hoods_possible = ctx->hoods && m <= (1ll << 22) && (kn_hood == 11 || kn_hood == 6);
    // (nothing reads them behind a build whose normals were all carried over, unless the map-sharded estimation will)
    const bool with_hoods = hoods_possible && ((wants_eager_normals(ctx, ctx->tgt_n > 0 ? ctx->tgt_n : m) && !carried_all) ||
                                               ctx->sharded_normals);
    // the list space goes back once HOOD_IDLE_RELEASE builds in a row had no use for it (or cannot have any): a map
    // whose scan size sits at the eager / lazy boundary must not pay a stream synchronisation and a free / malloc pair
    // per flip
    if (!carried_all) ctx->hood_idle_builds = with_hoods ? 0 : ctx->hood_idle_builds + 1;  // (a carried build says nothing)
    if (ctx->hood.ptr && !with_hoods && (!hoods_possible || ctx->hood_idle_builds >= HOOD_IDLE_RELEASE)) {
        ICP_HIP(ctx, hipStreamSynchronize(ctx->stream));  // (a launch of the previous build may still read them)
        ctx->hood.release();
    }
    const size_t hood_cap = (size_t)HOOD_PER_POINT * (size_t)m;  // 27 per point + the padding of every run to a multiple of four
    unsigned long long* hood_used = nullptr;
    if (with_hoods) {
        ICP_HIP(ctx, ctx->hood.reserve(hood_cap * sizeof(float4) + 64));
        hood_used = (unsigned long long*)(ctx->hood.as<char>() + hood_cap * sizeof(float4));  // 64 bytes of counters behind the lists
    }
    int* scan_ticket = ctx->scan_desc.as<int>();
    unsigned long long* desc = ctx->scan_desc.as<unsigned long long>() + 8;
    int* ncells_dev = &reg_state(ctx)->grid_cells;  // written by the scan, read by k_build_rows and, with the result, by the host
    GridBuildDesc d{};
    // cell lists ("cell_lists"): this build appends to one set; the other set — if the previous build wrote it for THIS table —
    // names the slots to empty
    CellLists lists;
    long long prev_cells_bound = -1;  // >= 0: the previous build's lists are usable (at most that many entries per level)
    if (ctx->cell_lists) {
        const int cur = ctx->cell_set, prev = cur ^ 1;
        ICP_HIP(ctx, ctx->cell_list[cur][0].reserve((size_t)m * sizeof(int)));
        ICP_HIP(ctx, ctx->cell_list[cur][1].reserve((size_t)m * sizeof(int)));
        if (!ctx->cell_counts.ptr) {
            ICP_HIP(ctx, ctx->cell_counts.reserve(64));
            ICP_HIP(ctx, hipMemsetAsync(ctx->cell_counts.ptr, 0, ctx->cell_counts.bytes, ctx->stream));
            ctx->cells_table = nullptr;
        }
        lists.fine = ctx->cell_list[cur][0].as<int>();
        lists.coarse = ctx->cell_list[cur][1].as<int>();
        lists.counts = ctx->cell_counts.as<int>() + 2 * cur;
        if (ctx->cells_table == (const void*)table && ctx->cells_tsize == tsize && ctx->cell_list[prev][0].ptr &&
            ctx->cell_list[prev][1].ptr) {
            lists.prev_fine = ctx->cell_list[prev][0].as<int>();
            lists.prev_coarse = ctx->cell_list[prev][1].as<int>();
            lists.prev_counts = ctx->cell_counts.as<int>() + 2 * prev;
            prev_cells_bound = ctx->cells_m;
        }
        ctx->cells_table = table;
        ctx->cells_tsize = tsize;
        ctx->cells_m = m;
        ctx->cell_set = prev;
    } else {
        ctx->cells_table = nullptr;  // (a build without lists: the next one with lists clears the whole table)
    }
    {
        const int seed_n = ctx->seed_job_n;
        ctx->seed_job_n = 0;
        // threads of the clearing launch: one per table slot — or, with the previous build's cell lists, one per entry of those
        long long span = prev_cells_bound >= 0 ? (prev_cells_bound > 1 ? prev_cells_bound : 1) : n2;
        if (seed_n > span) span = seed_n;
        const MapMoveJob move = ctx->move_job;  // the kept points re-expressed in the new frame (map update)
        ctx->move_job = MapMoveJob();
        if (move.m > span) span = move.m;
        d.lists = lists;
        d.table = table;
        d.n2 = (unsigned)n2;
        d.tsize = tsize;
        d.nn_cache = ctx->nn_cache.as<int4>();
        d.old_pts = ctx->sorted_pts.as<float4>();
        d.seed_n = seed_n;
        d.seed_m = ctx->seed_job_m;
        d.seed_evicted = ctx->seed_job_evicted;
        d.seed = ctx->seed_orig.as<int>();
        d.move = move;
        d.scan_ticket = scan_ticket;
        d.hood_used = hood_used;
        d.old_normals = ctx->normals.as<float4>();
        d.old_nflag = ctx->nflag.as<int>();
        d.carry_m = carry_m;
        d.carry = ctx->normals_carry.as<float4>();
        d.clear_blocks = (unsigned)((span + 255) / 256);
    }
    const long long visit_n = by_cell ? ctx->order_old_m + (m - ctx->order_kept) : 0;
    int* visit = by_cell ? ctx->worklist.as<int>() : (int*)nullptr;  // (the lazy-normal worklist is idle during a build)
    d.xyz = xyz;
    d.m = (int)m;
    d.inv_h = inv_h;
    d.inv_hc = inv_h / COARSE_FACTOR;
    d.slot_of = ctx->slot_of.as<int>();
    d.rank_of = ctx->rank_of.as<int>();
    d.cslot_of = ctx->cslot_of.as<int>();
    d.crank_of = ctx->crank_of.as<int>();
    d.order_pts = by_cell ? (const float4*)ctx->sorted_pts.as<float4>() : (const float4*)nullptr;
    d.order_old_m = (int)ctx->order_old_m;
    d.order_evicted = (int)ctx->order_evicted;
    d.order_kept = (int)ctx->order_kept;
    d.visit = visit;
    d.visit_n = (int)visit_n;
    d.visit_blocks = (unsigned)(((by_cell ? visit_n : m) + 255) / 256);  // workgroups of the claiming / scattering threads
    d.desc_a = desc;
    d.desc_b = desc + nb;
    d.scan_gen = scan_gen;
    d.poll_limit = ctx->scan_poll_limit;
    d.slot_of_cell = lists.counts ? lists.fine : ctx->slot_of_cell.as<int>();  // (the fine list IS the dense list of occupied fine cells)
    d.ncells = ncells_dev;
    d.scan_blocks = (unsigned)nb;
    {
        long long want = ((long long)m * 27 + 255) / 256;
        d.row_blocks = (unsigned)(want < 4096 ? (want < 1 ? 1 : want) : 4096);
    }
    d.rows = ctx->rows.as<int2>();
    d.sorted = ctx->sorted_pts.as<float4>();
    d.csorted = ctx->csorted.as<float4>();
    d.normals = ctx->normals.as<float4>();
    d.nflag = ctx->nflag.as<int>();
    d.row_of_pos = ctx->row_of_pos.as<int>();
    d.pos_of_orig = ctx->pos_of_orig.as<int>();
    d.with_hoods = with_hoods ? 1 : 0;
    d.hood_cap = (unsigned long long)hood_cap;
    // neighbourhood lists for the kNN normals (option "hoods"; maps beyond 2^22 points keep the row walk: 27 x 16 B per
    // point would be gigabytes).  The start of a run is an int: 27 M < 2^31 holds for every map that gets here
    ctx->hoods_valid = false;
    ctx->ctable_ptr = table + tsize;
    ctx->ctable_size = tsize;
    ctx->grid_valid = true;
    ctx->grid_gen += 1;  // cell-sorted positions handed out before this build are void
    if (defer) {  // the caller launches (B maps per launch) and finishes
        *defer = d;
        return ICP_OK;
    }
    hipLaunchKernelGGL(k_grid_clear, dim3(d.clear_blocks), dim3(256), 0, ctx->stream, d.table, d.n2, d.nn_cache, d.old_pts,
                       d.seed_n, d.seed_m, d.seed_evicted, d.seed, d.move, d.scan_ticket, d.hood_used, d.old_normals,
                       d.old_nflag, d.carry_m, d.carry, d.lists);
    hipLaunchKernelGGL(k_grid_insert2, dim3(d.visit_blocks), dim3(256), 0, ctx->stream, d.xyz, d.m, d.inv_h, d.inv_hc, d.table,
                       d.tsize, d.slot_of, d.rank_of, d.cslot_of, d.crank_of, d.order_pts, d.order_old_m, d.order_evicted,
                       d.order_kept, d.visit, d.lists);
    if (d.lists.counts)
        hipLaunchKernelGGL(k_cells_scan, dim3(2), dim3(CS_THREADS), 0, ctx->stream, d.table, d.lists, d.ncells);
    else
        hipLaunchKernelGGL(k_grid_scan, dim3(d.scan_blocks), dim3(SCAN_THREADS), 0, ctx->stream, d.table, (long long)d.n2, d.tsize,
                           d.m, d.desc_a, d.desc_b, d.scan_gen, d.poll_limit, d.slot_of_cell, d.ncells, d.scan_ticket);
    hipLaunchKernelGGL(k_grid_rows_scatter, dim3(d.row_blocks + d.visit_blocks), dim3(256), 0, ctx->stream, d.xyz, d.m, d.table,
                       d.tsize - 1, d.slot_of_cell, d.ncells, d.rows, (int)d.row_blocks, d.slot_of, d.rank_of, d.cslot_of,
                       d.crank_of, d.sorted, d.csorted, d.normals, d.nflag, d.row_of_pos, d.pos_of_orig, d.carry, d.carry_m,
                       d.visit, d.visit_n);
    ICP_HIP(ctx, hipGetLastError());
    return build_grid_finish(ctx, d);
}

int build_grid_finish(icp_ctx* ctx, const GridBuildDesc& d) {
    if (d.with_hoods) {
        const unsigned hb = (unsigned)((d.m + HOOD_THREADS / 32 - 1) / (HOOD_THREADS / 32));  // (cells <= points)
        hipLaunchKernelGGL(k_hood_build, dim3(hb), dim3(HOOD_THREADS), 0, ctx->stream, d.slot_of_cell, d.ncells, d.rows, d.sorted,
                           ctx->hood.as<float4>(), (long long)d.hood_cap, d.hood_used);
        ctx->hoods_valid = true;
        ICP_HIP(ctx, hipGetLastError());
    }
    return ICP_OK;
}

// the four launches for `count` maps (table_host: what the prepared builds returned, for the grid sizes; table_dev: the same
// table in device memory)
int launch_grid_build_batch(icp_ctx* first, const GridBuildDesc* th, const GridBuildDesc* td, int count) {
    unsigned clear = 0, visit = 0, scan = 0, scatter = 0;
    bool any_lists = false;
    for (int b = 0; b < count; ++b) {
        clear = th[b].clear_blocks > clear ? th[b].clear_blocks : clear;
        visit = th[b].visit_blocks > visit ? th[b].visit_blocks : visit;
        if (th[b].lists.counts) any_lists = any_lists || th[b].clear_blocks > 0;
        else scan = th[b].scan_blocks > scan ? th[b].scan_blocks : scan;
        const unsigned sc = th[b].row_blocks + th[b].visit_blocks;
        scatter = sc > scatter ? sc : scatter;
    }
    hipLaunchKernelGGL(k_grid_clear_batch, dim3(clear, count), dim3(256), 0, first->stream, td);
    hipLaunchKernelGGL(k_grid_insert2_batch, dim3(visit, count), dim3(256), 0, first->stream, td);
    if (any_lists) hipLaunchKernelGGL(k_cells_scan_batch, dim3(2, count), dim3(CS_THREADS), 0, first->stream, td);
    if (scan > 0) hipLaunchKernelGGL(k_grid_scan_batch, dim3(scan, count), dim3(SCAN_THREADS), 0, first->stream, td);
    hipLaunchKernelGGL(k_grid_rows_scatter_batch, dim3(scatter, count), dim3(256), 0, first->stream, td);
    ICP_HIP(first, hipGetLastError());
    return ICP_OK;
}

}  // namespace icp
