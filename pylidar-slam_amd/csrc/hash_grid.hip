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
__global__ void k_grid_clear(GridEntry* __restrict__ table, unsigned int size) {
    unsigned int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < size) {
        GridEntry e;
        e.key = GRID_EMPTY;
        e.start = 0;
        e.count = 0;
        table[i] = e;
    }
}

__global__ void k_grid_insert(const float* __restrict__ xyz, int m, float inv_h, GridEntry* __restrict__ table,
                              unsigned int mask, int* __restrict__ slot_of, int* __restrict__ rank_of,
                              int* __restrict__ stats) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= m) return;
    const float x = xyz[3 * i + 0], y = xyz[3 * i + 1], z = xyz[3 * i + 2];
    const unsigned long long key = pack_cell(cell_coord(x, inv_h), cell_coord(y, inv_h), cell_coord(z, inv_h));
    unsigned int slot = hash_cell(key) & mask;
    while (true) {
        unsigned long long old = atomicCAS(&table[slot].key, GRID_EMPTY, key);
        if (old == GRID_EMPTY) atomicAdd(&stats[0], 1);  // occupied cells (feeds the cell-size auto-tuning)
        if (old == GRID_EMPTY || old == key) break;
        slot = (slot + 1) & mask;
    }
    slot_of[i] = (int)slot;
    rank_of[i] = atomicAdd(&table[slot].count, 1);
}

__global__ void k_grid_counts(const GridEntry* __restrict__ table, unsigned int size, int* __restrict__ counts,
                              int* __restrict__ flags) {
    unsigned int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < size) {
        const int c = table[i].count;
        counts[i] = c;
        flags[i] = c > 0 ? 1 : 0;
    }
}

__global__ void k_grid_starts(GridEntry* __restrict__ table, unsigned int size, const int* __restrict__ starts,
                              const int* __restrict__ flags, const int* __restrict__ cell_ids,
                              int* __restrict__ row_of_slot, int* __restrict__ slot_of_cell) {
    unsigned int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= size) return;
    table[i].start = starts[i];
    const int r = flags[i] ? cell_ids[i] : -1;
    row_of_slot[i] = r;
    if (r >= 0) slot_of_cell[r] = (int)i;
}

__global__ void k_grid_starts_plain(GridEntry* __restrict__ table, unsigned int size, const int* __restrict__ starts) {
    unsigned int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < size) table[i].start = starts[i];
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
        rows[(size_t)j * ROW_STRIDE + c] = out;
        if (c == 26) rows[(size_t)j * ROW_STRIDE + 27] = make_int2(0, 0);  // padding entry
    }
}

__global__ void k_grid_scatter(const float* __restrict__ xyz, int m, const GridEntry* __restrict__ table,
                               const int* __restrict__ slot_of, const int* __restrict__ rank_of,
                               const int* __restrict__ row_of_slot, float4* __restrict__ sorted,
                               float4* __restrict__ normals, int* __restrict__ nflag, int* __restrict__ row_of_pos,
                               int* __restrict__ pos_of_orig) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= m) return;
    const int pos = table[slot_of[i]].start + rank_of[i];
    row_of_pos[pos] = row_of_slot[slot_of[i]];
    pos_of_orig[i] = pos;
    sorted[pos] = make_float4(xyz[3 * i + 0], xyz[3 * i + 1], xyz[3 * i + 2], __int_as_float(i));
    // the normal cache is cleared on every rebuild (local_map.py:368)
    normals[i] = make_float4(0.f, 0.f, 0.f, 0.f);
    nflag[i] = 0;
}

__global__ void k_grid_scatter_plain(const float* __restrict__ xyz, int m, const GridEntry* __restrict__ table,
                                     const int* __restrict__ slot_of, const int* __restrict__ rank_of,
                                     float4* __restrict__ sorted) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= m) return;
    sorted[table[slot_of[i]].start + rank_of[i]] =
        make_float4(xyz[3 * i + 0], xyz[3 * i + 1], xyz[3 * i + 2], __int_as_float(i));
}

static unsigned int next_pow2(unsigned int v) {
    unsigned int p = 1024;
    while (p < v) p <<= 1;
    return p;
}

int build_grid(icp_ctx* ctx) {
    const int64_t m = ctx->map_m;
    ctx->grid_valid = false;
    if (m <= 0) return ICP_OK;
    if (m > (int64_t)1 << 30) {
        ctx->error = "local map too large";
        return ICP_ERR_INVALID_ARGUMENT;
    }
    const unsigned int tsize = next_pow2((unsigned int)(2 * m));
    ICP_HIP(ctx, ctx->table.reserve((size_t)tsize * sizeof(GridEntry)));
    ICP_HIP(ctx, ctx->sorted_pts.reserve((size_t)m * sizeof(float4)));
    ICP_HIP(ctx, ctx->normals.reserve((size_t)m * sizeof(float4)));
    ICP_HIP(ctx, ctx->nflag.reserve((size_t)m * sizeof(int)));
    ICP_HIP(ctx, ctx->slot_of.reserve((size_t)m * sizeof(int)));
    ICP_HIP(ctx, ctx->rank_of.reserve((size_t)m * sizeof(int)));
    ICP_HIP(ctx, ctx->worklist.reserve((size_t)m * sizeof(int)));
    ICP_HIP(ctx, ctx->scan_b.reserve((size_t)tsize * sizeof(int)));
    ICP_HIP(ctx, ctx->cell_flags.reserve((size_t)tsize * sizeof(int)));
    ICP_HIP(ctx, ctx->cell_ids.reserve((size_t)tsize * sizeof(int)));
    ICP_HIP(ctx, ctx->row_of_slot.reserve((size_t)tsize * sizeof(int)));
    ICP_HIP(ctx, ctx->slot_of_cell.reserve((size_t)m * sizeof(int)));
    ICP_HIP(ctx, ctx->row_of_pos.reserve((size_t)m * sizeof(int)));
    ICP_HIP(ctx, ctx->pos_of_orig.reserve((size_t)m * sizeof(int)));
    ICP_HIP(ctx, ctx->rows.reserve((size_t)m * ROW_STRIDE * sizeof(int2)));  // worst case: one cell per point
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
    ICP_HIP(ctx, ctx->grid_stats.reserve(16));
    ICP_HIP(ctx, hipMemsetAsync(ctx->grid_stats.ptr, 0, 16, ctx->stream));
    ctx->normals_ready = false;
    ctx->stats_pending = true;
    ctx->stats_m_pending = m;
    ctx->stats_h_pending = ctx->cell_h;
    const unsigned tb = (tsize + 255) / 256, mb = (unsigned)((m + 255) / 256);
    hipLaunchKernelGGL(k_grid_clear, dim3(tb), dim3(256), 0, ctx->stream, table, tsize);
    hipLaunchKernelGGL(k_grid_insert, dim3(mb), dim3(256), 0, ctx->stream, xyz, (int)m, inv_h, table, tsize - 1,
                       ctx->slot_of.as<int>(), ctx->rank_of.as<int>(), ctx->grid_stats.as<int>());
    int* counts = ctx->scan_b.as<int>();
    int* flags = ctx->cell_flags.as<int>();
    int* cell_ids = ctx->cell_ids.as<int>();
    int* ncells_dev = ctx->grid_stats.as<int>() + 1;
    hipLaunchKernelGGL(k_grid_counts, dim3(tb), dim3(256), 0, ctx->stream, table, tsize, counts, flags);
    int rc = exclusive_scan_i32(ctx, counts, counts, tsize, nullptr);
    if (rc) return rc;
    if ((rc = exclusive_scan_i32(ctx, flags, cell_ids, tsize, ncells_dev))) return rc;
    hipLaunchKernelGGL(k_grid_starts, dim3(tb), dim3(256), 0, ctx->stream, table, tsize, counts, flags, cell_ids,
                       ctx->row_of_slot.as<int>(), ctx->slot_of_cell.as<int>());
    {
        long long want = ((long long)m * 27 + 255) / 256;
        const unsigned rb = (unsigned)(want < 4096 ? (want < 1 ? 1 : want) : 4096);
        hipLaunchKernelGGL(k_build_rows, dim3(rb), dim3(256), 0, ctx->stream, table, tsize - 1,
                           ctx->slot_of_cell.as<int>(), ncells_dev, ctx->rows.as<int2>());
    }
    hipLaunchKernelGGL(k_grid_scatter, dim3(mb), dim3(256), 0, ctx->stream, xyz, (int)m, table,
                       ctx->slot_of.as<int>(), ctx->rank_of.as<int>(), ctx->row_of_slot.as<int>(),
                       ctx->sorted_pts.as<float4>(), ctx->normals.as<float4>(), ctx->nflag.as<int>(),
                       ctx->row_of_pos.as<int>(), ctx->pos_of_orig.as<int>());
    // ---- coarse level: the same counting sort with cells COARSE_FACTOR times larger (table sized for fewer cells)
    {
        const unsigned int csize = tsize;  // worst case (every point in its own coarse cell) must still fit
        ICP_HIP(ctx, ctx->ctable.reserve((size_t)csize * sizeof(GridEntry)));
        ICP_HIP(ctx, ctx->csorted.reserve((size_t)m * sizeof(float4)));
        ctx->ctable_size = csize;
        GridEntry* ctable = ctx->ctable.as<GridEntry>();
        const unsigned cb = (csize + 255) / 256;
        hipLaunchKernelGGL(k_grid_clear, dim3(cb), dim3(256), 0, ctx->stream, ctable, csize);
        hipLaunchKernelGGL(k_grid_insert, dim3(mb), dim3(256), 0, ctx->stream, xyz, (int)m, inv_h / COARSE_FACTOR,
                           ctable, csize - 1, ctx->slot_of.as<int>(), ctx->rank_of.as<int>(),
                           ctx->grid_stats.as<int>() + 2);
        hipLaunchKernelGGL(k_grid_counts, dim3(cb), dim3(256), 0, ctx->stream, ctable, csize, counts, flags);
        if ((rc = exclusive_scan_i32(ctx, counts, counts, csize, nullptr))) return rc;
        hipLaunchKernelGGL(k_grid_starts_plain, dim3(cb), dim3(256), 0, ctx->stream, ctable, csize, counts);
        hipLaunchKernelGGL(k_grid_scatter_plain, dim3(mb), dim3(256), 0, ctx->stream, xyz, (int)m, ctable,
                           ctx->slot_of.as<int>(), ctx->rank_of.as<int>(), ctx->csorted.as<float4>());
    }
    ICP_HIP(ctx, hipGetLastError());
    ctx->grid_valid = true;
    return ICP_OK;
}

}  // namespace icp
