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
                                  const int* __restrict__ offs, long long n, int row_floats, float* __restrict__ out) {
    long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n || !flags[i]) return;
    long long o = offs[i];
    for (int c = 0; c < row_floats; ++c) out[o * row_floats + c] = in[i * row_floats + c];
}

int compact_rows(icp_ctx* ctx, const float* in, const int* flags, int64_t n, int row_floats, float* out,
                 int* count_dev) {
    ICP_HIP(ctx, ctx->scan_a.reserve((size_t)(n > 0 ? n : 1) * sizeof(int)));
    int* offs = ctx->scan_a.as<int>();
    int rc = exclusive_scan_i32(ctx, flags, offs, n, count_dev);
    if (rc) return rc;
    if (n > 0) {
        hipLaunchKernelGGL(k_compact_scatter, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, ctx->stream, in, flags,
                           offs, (long long)n, row_floats, out);
        ICP_HIP(ctx, hipGetLastError());
    }
    return ICP_OK;
}

// ---------------------------------------------------------------------------------------------------------------------
// grid build
// ---------------------------------------------------------------------------------------------------------------------
// + (seed_n > 0) the neighbours the last registration left in the NN cache -> original map indices, shifted by the
// `evicted` oldest points this update drops: the seeds of the next frame's first iteration.  Must run while the old
// cell-sorted positions still mean something, i.e. before the scatter of this build: it shares the first launch.
__global__ void k_grid_clear(GridEntry* __restrict__ table, unsigned int size, const int2* __restrict__ nn_cache,
                             const float4* __restrict__ old_pts, int seed_n, int old_m, int evicted,
                             int* __restrict__ seed) {
    unsigned int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < size) {
        GridEntry e;
        e.key = GRID_EMPTY;
        e.start = 0;
        e.count = 0;
        table[i] = e;
    }
    if ((int)i < seed_n) {
        const int pos = nn_cache[i].x;
        int o = -1;
        if (pos >= 0 && pos < old_m) o = __float_as_int(old_pts[pos].w) - evicted;
        seed[i] = o;
    }
}

// rows[cell][c] = (start, count) of the neighbour cell c of every occupied cell (0,0 if that neighbour is empty)
__global__ void k_build_rows(const GridEntry* __restrict__ table, unsigned int mask,
                             const int* __restrict__ slot_of_cell, const int* __restrict__ ncells_dev,
                             int2* __restrict__ rows) {
    const long long total = (long long)(*ncells_dev) * 27;
    for (long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x; t < total;
         t += (long long)gridDim.x * blockDim.x) {
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
// so clearing, inserting, the count scan and the scatter are one launch each (7 launches per rebuild instead of 21).
// One 64-bit scan carries two sums at once: low word = points before the slot (cell start; the fine level holds exactly
// m points, so coarse starts are that prefix minus m), high word = occupied FINE cells before the slot (row id).
// ---------------------------------------------------------------------------------------------------------------------
// (the occupied cells are counted by the scan, not here: one atomic counter for ten thousand claims was two thirds of the
// insertion kernel — 31 us)
__device__ inline unsigned int grid_claim(GridEntry* __restrict__ table, unsigned int mask, unsigned long long key) {
    unsigned int slot = hash_cell(key) & mask;
    while (true) {
        unsigned long long old = atomicCAS(&table[slot].key, GRID_EMPTY, key);
        if (old == GRID_EMPTY || old == key) break;
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
__global__ void k_grid_insert2(const float* __restrict__ xyz, int m, float inv_h, float inv_hc,
                               GridEntry* __restrict__ table, unsigned int tsize, int* __restrict__ slot_of,
                               int* __restrict__ rank_of, int* __restrict__ cslot_of, int* __restrict__ crank_of) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    const bool active = i < m;
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
    if (active && leader == lane) slot = (int)grid_claim(table, tsize - 1, key);
    if (active && cleader == lane) cslot = (int)(tsize + grid_claim(table + tsize, tsize - 1, ckey));
    if (active && leader == lane) base = atomicAdd(&table[slot].count, size);  // the two requests are in flight together
    if (active && cleader == lane) cbase = atomicAdd(&table[cslot].count, csize);
    slot = __shfl(slot, leader, 64);
    base = __shfl(base, leader, 64);
    cslot = __shfl(cslot, cleader, 64);
    cbase = __shfl(cbase, cleader, 64);
    if (!active) return;
    slot_of[i] = slot;
    rank_of[i] = base + rank;
    cslot_of[i] = cslot;
    crank_of[i] = cbase + crank;
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

__global__ __launch_bounds__(SCAN_THREADS) void k_grid_tile_sums(const GridEntry* __restrict__ table, long long n,
                                                                 unsigned int tsize,
                                                                 unsigned long long* __restrict__ sums) {
    __shared__ unsigned long long lds[16];
    const long long base = (long long)blockIdx.x * SCAN_TILE + (long long)threadIdx.x * SCAN_ITEMS;
    unsigned long long s = 0;
#pragma unroll
    for (int k = 0; k < SCAN_ITEMS; ++k) s += grid_scan_item(table, base + k, n, tsize);
    unsigned long long tot;
    block_exclusive_scan_u64(s, &tot, lds);
    if (threadIdx.x == 0) sums[blockIdx.x] = tot;
}

// single block: exclusive scan of the tile sums in place; the number of occupied fine cells -> *ncells_out
__global__ __launch_bounds__(1024) void k_grid_scan_sums(unsigned long long* __restrict__ sums, int nb,
                                                         int* __restrict__ ncells_out) {
    __shared__ unsigned long long lds[16];
    __shared__ unsigned long long carry_s;
    if (threadIdx.x == 0) carry_s = 0;
    __syncthreads();
    for (int base = 0; base < nb; base += blockDim.x) {
        const int i = base + threadIdx.x;
        const unsigned long long v = (i < nb) ? sums[i] : 0ull;
        unsigned long long tot;
        const unsigned long long excl = block_exclusive_scan_u64(v, &tot, lds);
        const unsigned long long carry = carry_s;
        if (i < nb) sums[i] = carry + excl;
        __syncthreads();
        if (threadIdx.x == 0) carry_s = carry + tot;
        __syncthreads();
    }
    if (threadIdx.x == 0) *ncells_out = (int)(carry_s >> 32);
}

__global__ __launch_bounds__(SCAN_THREADS) void k_grid_apply(GridEntry* __restrict__ table, long long n,
                                                             unsigned int tsize, int m,
                                                             const unsigned long long* __restrict__ sums,
                                                             int* __restrict__ slot_of_cell) {
    __shared__ unsigned long long lds[16];
    const long long base = (long long)blockIdx.x * SCAN_TILE + (long long)threadIdx.x * SCAN_ITEMS;
    unsigned long long v[SCAN_ITEMS];
    unsigned long long s = 0;
#pragma unroll
    for (int k = 0; k < SCAN_ITEMS; ++k) {
        v[k] = grid_scan_item(table, base + k, n, tsize);
        s += v[k];
    }
    unsigned long long tot;
    unsigned long long off = block_exclusive_scan_u64(s, &tot, lds) + sums[blockIdx.x];
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

__global__ void k_grid_scatter2(const float* __restrict__ xyz, int m, const GridEntry* __restrict__ table,
                                const int* __restrict__ slot_of, const int* __restrict__ rank_of,
                                const int* __restrict__ cslot_of, const int* __restrict__ crank_of,
                                float4* __restrict__ sorted,
                                float4* __restrict__ csorted, float4* __restrict__ normals, int* __restrict__ nflag,
                                int* __restrict__ row_of_pos, int* __restrict__ pos_of_orig) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= m) return;
    const int slot = slot_of[i];
    const int pos = table[slot].start + rank_of[i];
    const int cpos = table[cslot_of[i]].start + crank_of[i];
    const float4 p = make_float4(xyz[3 * i + 0], xyz[3 * i + 1], xyz[3 * i + 2], __int_as_float(i));
    row_of_pos[pos] = slot;  // rows are indexed by slot
    pos_of_orig[i] = pos;
    sorted[pos] = p;
    csorted[cpos] = p;
    // the normal cache is cleared on every rebuild (local_map.py:368)
    normals[i] = make_float4(0.f, 0.f, 0.f, 0.f);
    nflag[i] = 0;
}

static unsigned int next_pow2(unsigned int v) {
    unsigned int p = 1024;
    while (p < v) p <<= 1;
    return p;
}

int build_grid(icp_ctx* ctx) {
    const int64_t m = ctx->map_m;
    ctx->grid_valid = false;
    if (m <= 0) {
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
    if (ctx->seed_job_n > 0 && ctx->sorted_pts.bytes < (size_t)m * sizeof(float4)) {
        const int rc = run_seed_job(ctx);
        if (rc) return rc;
    }
    const unsigned int tsize = next_pow2((unsigned int)(2 * m));
    ICP_HIP(ctx, ctx->table.reserve((size_t)2 * tsize * sizeof(GridEntry)));  // fine level, then the coarse level
    ICP_HIP(ctx, ctx->csorted.reserve((size_t)m * sizeof(float4)));
    ICP_HIP(ctx, ctx->cslot_of.reserve((size_t)m * sizeof(int)));
    ICP_HIP(ctx, ctx->crank_of.reserve((size_t)m * sizeof(int)));
    ICP_HIP(ctx, ctx->sorted_pts.reserve((size_t)m * sizeof(float4)));
    ICP_HIP(ctx, ctx->normals.reserve((size_t)m * sizeof(float4)));
    ICP_HIP(ctx, ctx->nflag.reserve((size_t)m * sizeof(int)));
    ICP_HIP(ctx, ctx->slot_of.reserve((size_t)m * sizeof(int)));
    ICP_HIP(ctx, ctx->rank_of.reserve((size_t)m * sizeof(int)));
    ICP_HIP(ctx, ctx->worklist.reserve((size_t)m * sizeof(int)));
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
    ctx->normals_ready = false;
    ctx->stats_pending = true;
    ctx->stats_m_pending = m;
    ctx->stats_h_pending = ctx->cell_h;
    const unsigned mb = (unsigned)((m + 255) / 256);
    const long long n2 = 2ll * tsize;  // both levels
    const int nb = (int)((n2 + SCAN_TILE - 1) / SCAN_TILE);
    ICP_HIP(ctx, ctx->scan_tmp.reserve((size_t)nb * sizeof(unsigned long long)));
    unsigned long long* sums = ctx->scan_tmp.as<unsigned long long>();
    int* ncells_dev = &reg_state(ctx)->grid_cells;  // written by the scan, read by k_build_rows and, with the result, by the host
    {
        const int seed_n = ctx->seed_job_n;
        ctx->seed_job_n = 0;
        const long long span = n2 > seed_n ? n2 : (long long)seed_n;
        hipLaunchKernelGGL(k_grid_clear, dim3((unsigned)((span + 255) / 256)), dim3(256), 0, ctx->stream, table,
                           (unsigned int)n2, ctx->nn_cache.as<int2>(), ctx->sorted_pts.as<float4>(), seed_n,
                           ctx->seed_job_m, ctx->seed_job_evicted, ctx->seed_orig.as<int>());
    }
    hipLaunchKernelGGL(k_grid_insert2, dim3(mb), dim3(256), 0, ctx->stream, xyz, (int)m, inv_h, inv_h / COARSE_FACTOR,
                       table, tsize, ctx->slot_of.as<int>(), ctx->rank_of.as<int>(), ctx->cslot_of.as<int>(),
                       ctx->crank_of.as<int>());
    hipLaunchKernelGGL(k_grid_tile_sums, dim3(nb), dim3(SCAN_THREADS), 0, ctx->stream, table, n2, tsize, sums);
    hipLaunchKernelGGL(k_grid_scan_sums, dim3(1), dim3(1024), 0, ctx->stream, sums, nb, ncells_dev);
    hipLaunchKernelGGL(k_grid_apply, dim3(nb), dim3(SCAN_THREADS), 0, ctx->stream, table, n2, tsize, (int)m, sums,
                       ctx->slot_of_cell.as<int>());
    {
        long long want = ((long long)m * 27 + 255) / 256;
        const unsigned rb = (unsigned)(want < 4096 ? (want < 1 ? 1 : want) : 4096);
        hipLaunchKernelGGL(k_build_rows, dim3(rb), dim3(256), 0, ctx->stream, table, tsize - 1,
                           ctx->slot_of_cell.as<int>(), ncells_dev, ctx->rows.as<int2>());
    }
    hipLaunchKernelGGL(k_grid_scatter2, dim3(mb), dim3(256), 0, ctx->stream, xyz, (int)m, table,
                       ctx->slot_of.as<int>(), ctx->rank_of.as<int>(), ctx->cslot_of.as<int>(),
                       ctx->crank_of.as<int>(), ctx->sorted_pts.as<float4>(),
                       ctx->csorted.as<float4>(), ctx->normals.as<float4>(), ctx->nflag.as<int>(),
                       ctx->row_of_pos.as<int>(), ctx->pos_of_orig.as<int>());
    ctx->ctable_ptr = table + tsize;
    ctx->ctable_size = tsize;
    ICP_HIP(ctx, hipGetLastError());
    ctx->grid_valid = true;
    ctx->grid_gen += 1;  // cell-sorted positions handed out before this build are void
    return ICP_OK;
}

}  // namespace icp
