// Voxel-grid subsampling: one point per voxel hash.
//
// Replaces the reference's numba kernels `voxelise` / `voxel_hashing` (slam/common/pointcloud.py:13-79) and the
// `np.unique(hashes, return_index=True)` of `sample_from_hashes` (:170-179) used by `GridSample.filter`
// (slam/preprocessing.py:213-226): the sample of a voxel is its FIRST point in input order, and the samples come out
// ordered by ascending signed int64 hash (hash collisions merge voxels, exactly as in the reference).
//
//   voxel  = int64(round_half_even(double(p) / voxel_size))      numba promotes f32 / f64 to f64
//   hash   = 73856093 x + 19349669 y + 83492791 z                wrapping int64
//   dedupe (hash -> smallest index) through a hash table, then sort the V distinct (hash, index) pairs by hash.
// The sort of the V pairs is rocPRIM's device radix sort (a library primitive); everything else is hand-written.
#include <cmath>
#include <cstring>
#include <string.h>

#include <rocprim/rocprim.hpp>

#include "icp_internal.h"

namespace icp {

template <typename T>
__device__ inline long long voxel_coord(T v, double voxel) {
    // int(np.round_(p / voxel)); rint = round-half-even
    return (long long)rint((double)v / voxel);
}

template <typename T>
__global__ void k_voxel_hash(const T* __restrict__ xyz, int n, double voxel, long long* __restrict__ voxels,
                             long long* __restrict__ hashes, unsigned long long* __restrict__ sort_keys,
                             int* __restrict__ sort_vals) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const long long vx = voxel_coord(xyz[3 * i], voxel), vy = voxel_coord(xyz[3 * i + 1], voxel),
                    vz = voxel_coord(xyz[3 * i + 2], voxel);
    // wrapping arithmetic: do it unsigned
    const unsigned long long h = 73856093ull * (unsigned long long)vx + 19349669ull * (unsigned long long)vy +
                                 83492791ull * (unsigned long long)vz;
    if (voxels) {
        voxels[3 * i] = vx;
        voxels[3 * i + 1] = vy;
        voxels[3 * i + 2] = vz;
    }
    if (hashes) hashes[i] = (long long)h;
    if (sort_keys) {
        sort_keys[i] = h ^ 0x8000000000000000ull;  // signed order under an unsigned sort
        sort_vals[i] = i;
    }
}

__global__ void k_run_heads(const unsigned long long* __restrict__ keys, int n, int* __restrict__ flags) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    flags[i] = (i == 0 || keys[i] != keys[i - 1]) ? 1 : 0;
}

template <typename T>
__global__ void k_emit_samples(const T* __restrict__ xyz, const int* __restrict__ vals,
                               const int* __restrict__ flags, const int* __restrict__ offs, int n,
                               long long* __restrict__ indices, T* __restrict__ points) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n || !flags[i]) return;
    const int o = offs[i];
    const int src = vals[i];
    if (indices) indices[o] = src;
    if (points) {
        points[3 * o] = xyz[3 * src];
        points[3 * o + 1] = xyz[3 * src + 1];
        points[3 * o + 2] = xyz[3 * src + 2];
    }
}

int voxel_hash_device(icp_ctx* ctx, const float* xyz_dev, int64_t n, double voxel, long long* voxels_dev,
                      long long* hashes_dev) {
    if (n <= 0) return ICP_OK;
    hipLaunchKernelGGL(k_voxel_hash<float>, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, ctx->stream, xyz_dev, (int)n,
                       voxel, voxels_dev, hashes_dev, (unsigned long long*)nullptr, (int*)nullptr);
    ICP_HIP(ctx, hipGetLastError());
    return ICP_OK;
}

// ---------------------------------------------------------------------------------------------------------------------
// The sample of a voxel is the point with the smallest index among those that share its hash: a DEDUPE, then a sort of
// the V << n distinct hashes (a 64x2048 scan at 0.4 m: 131 072 points, ~6 000 voxels).  Round 2 sorted all n (hash,
// index) pairs (rocPRIM, 64-bit keys: ~75 of the op's ~100 us at n = 131 072, profiles/r03_secondary_*).
//   1. k_hash_dedupe: every point claims the slot of its hash in an open-addressing table (atomicCAS on the key) and
//      lowers the slot's index (atomicMin); consecutive points mostly share their voxel, so a wave first groups its
//      lanes by hash (ballots) and only the group leaders — carrying the group's smallest index — touch memory;
//   2. k_hash_collect: the occupied slots -> dense (hash ^ sign bit, index) pairs, appended wave by wave (the order does
//      not matter: they are sorted next);
//   3. rocPRIM radix sort of the V pairs by signed hash;  4. k_emit_sorted: indices + gathered points.
// The empty-slot sentinel is a key value no slot ever stores: the one hash equal to it goes to a side cell.
// ---------------------------------------------------------------------------------------------------------------------
static constexpr unsigned long long DEDUPE_EMPTY = ~0ull;

struct DedupeSlot {
    unsigned long long key;
    int idx;
    int pad;
};

__global__ void k_dedupe_clear(DedupeSlot* __restrict__ table, unsigned int size, int* __restrict__ side) {
    const unsigned int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < size) {
        DedupeSlot e;
        e.key = DEDUPE_EMPTY;
        e.idx = 0x7fffffff;
        e.pad = 0;
        table[i] = e;
    }
    if (i == 0) {
        side[0] = 0x7fffffff;  // smallest index with hash == DEDUPE_EMPTY
        side[1] = 0;           // pairs collected
    }
}

template <typename T>
__global__ void k_hash_dedupe(const T* __restrict__ xyz, int n, double voxel, DedupeSlot* __restrict__ table,
                              unsigned int mask, int* __restrict__ side) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    const bool active = i < n;
    unsigned long long h = 0;
    if (active) {
        const long long vx = voxel_coord(xyz[3 * i], voxel), vy = voxel_coord(xyz[3 * i + 1], voxel),
                        vz = voxel_coord(xyz[3 * i + 2], voxel);
        h = 73856093ull * (unsigned long long)vx + 19349669ull * (unsigned long long)vy +
            83492791ull * (unsigned long long)vz;
    }
    // lanes of the wave with the same hash: the lowest lane holds the smallest index (indices ascend with the lane)
    const int lane = threadIdx.x & 63;
    unsigned long long todo = __ballot(active);
    bool leader = false;
    while (todo) {
        const int l = __ffsll((long long)todo) - 1;
        const unsigned lo = __shfl((unsigned)(h & 0xffffffffull), l, 64), hi = __shfl((unsigned)(h >> 32), l, 64);
        const unsigned long long k = ((unsigned long long)hi << 32) | lo;
        const unsigned long long same = __ballot(active && h == k);
        if (lane == l) leader = true;
        todo &= ~same;
    }
    if (!leader) return;
    if (h == DEDUPE_EMPTY) {
        atomicMin(&side[0], i);
        return;
    }
    unsigned long long x = h;  // slot from a mixed copy of the hash (its low bits alone cluster: small multiples)
    x ^= x >> 33;
    x *= 0xff51afd7ed558ccdull;
    x ^= x >> 33;
    unsigned int slot = (unsigned int)x & mask;
    while (true) {
        const unsigned long long old = atomicCAS(&table[slot].key, DEDUPE_EMPTY, h);
        if (old == DEDUPE_EMPTY || old == h) break;
        slot = (slot + 1) & mask;
    }
    atomicMin(&table[slot].idx, i);
}

// (a few large workgroups, ONE atomicAdd each: with one per wave — ~3 000 same-address device-scope atomics at 131 072
// points, ~16 ns apiece — this kernel alone took 52 us)
static constexpr int COLLECT_THREADS = 1024;
static constexpr int COLLECT_BLOCKS = 64;

__global__ __launch_bounds__(COLLECT_THREADS) void k_hash_collect(const DedupeSlot* __restrict__ table,
                                                                  unsigned int size, int* __restrict__ side,
                                                                  unsigned long long* __restrict__ keys,
                                                                  int* __restrict__ vals) {
    __shared__ int wave_tot[COLLECT_THREADS / 64];
    __shared__ int base_s;
    // slots [lo, hi) of this workgroup; the side cell (hash == DEDUPE_EMPTY) rides as slot `size`
    const unsigned int total = size + 1u;
    const unsigned int per = (total + COLLECT_BLOCKS - 1) / COLLECT_BLOCKS;
    const unsigned int lo = blockIdx.x * per, hi = min(lo + per, total);
    int mine = 0;
    for (unsigned int i = lo + threadIdx.x; i < hi; i += COLLECT_THREADS)
        mine += (i < size) ? (table[i].key != DEDUPE_EMPTY ? 1 : 0) : (side[0] != 0x7fffffff ? 1 : 0);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    int incl = mine;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const int t = __shfl_up(incl, o, 64);
        if (lane >= o) incl += t;
    }
    if (lane == 63) wave_tot[wave] = incl;
    __syncthreads();
    int off = incl - mine, tot = 0;
    for (int w = 0; w < COLLECT_THREADS / 64; ++w) {
        if (w < wave) off += wave_tot[w];
        tot += wave_tot[w];
    }
    if (threadIdx.x == 0) base_s = tot > 0 ? atomicAdd(&side[1], tot) : 0;
    __syncthreads();
    int o = base_s + off;
    for (unsigned int i = lo + threadIdx.x; i < hi; i += COLLECT_THREADS) {
        if (i < size) {
            const DedupeSlot e = table[i];
            if (e.key == DEDUPE_EMPTY) continue;
            keys[o] = e.key ^ 0x8000000000000000ull;  // signed order under an unsigned sort
            vals[o] = e.idx;
            ++o;
        } else if (side[0] != 0x7fffffff) {
            keys[o] = DEDUPE_EMPTY ^ 0x8000000000000000ull;
            vals[o] = side[0];
            ++o;
        }
    }
}

template <typename T>
__global__ void k_emit_sorted(const T* __restrict__ xyz, const int* __restrict__ vals, const int* __restrict__ count,
                              long long* __restrict__ indices, T* __restrict__ points) {
    const int o = blockIdx.x * blockDim.x + threadIdx.x;
    if (o >= *count) return;
    const int src = vals[o];
    if (indices) indices[o] = src;
    if (points) {
        points[3 * o] = xyz[3 * src];
        points[3 * o + 1] = xyz[3 * src + 1];
        points[3 * o + 2] = xyz[3 * src + 2];
    }
}

template <typename T>
static int grid_sample_impl(icp_ctx* ctx, const T* xyz_dev, int64_t n, double voxel, long long* indices_dev,
                            T* points_dev, int* count_dev, int* count_host) {
    *count_host = 0;
    if (n <= 0) {
        ICP_HIP(ctx, hipMemsetAsync(count_dev, 0, sizeof(int), ctx->stream));
        return ICP_OK;
    }
    unsigned int tsize = 1024;
    while ((int64_t)tsize < 2 * n) tsize <<= 1;
    ICP_HIP(ctx, ctx->keys_a.reserve((size_t)tsize * sizeof(DedupeSlot)));  // the table
    ICP_HIP(ctx, ctx->keys_b.reserve((size_t)n * 8));                        // at most n distinct hashes
    ICP_HIP(ctx, ctx->vals_a.reserve((size_t)n * 4));
    ICP_HIP(ctx, ctx->vals_b.reserve((size_t)n * 4));
    ICP_HIP(ctx, ctx->scan_a.reserve((size_t)n * 8));                        // sorted keys
    ICP_HIP(ctx, ctx->flags.reserve(64));
    DedupeSlot* table = ctx->keys_a.as<DedupeSlot>();
    unsigned long long* ka = ctx->keys_b.as<unsigned long long>();
    unsigned long long* kb = ctx->scan_a.as<unsigned long long>();
    int* va = ctx->vals_a.as<int>();
    int* vb = ctx->vals_b.as<int>();
    int* side = ctx->flags.as<int>();
    const unsigned nb = (unsigned)((n + 255) / 256);
    hipLaunchKernelGGL(k_dedupe_clear, dim3((tsize + 255) / 256), dim3(256), 0, ctx->stream, table, tsize, side);
    hipLaunchKernelGGL(k_hash_dedupe<T>, dim3(nb), dim3(256), 0, ctx->stream, xyz_dev, (int)n, voxel, table, tsize - 1,
                       side);
    hipLaunchKernelGGL(k_hash_collect, dim3(COLLECT_BLOCKS), dim3(COLLECT_THREADS), 0, ctx->stream, table, tsize, side,
                       ka, va);
    // the number of pairs is only known on the device: the host needs it to size the sort (and returns it anyway)
    int v = 0;
    ICP_HIP(ctx, hipMemcpyAsync(&v, side + 1, sizeof(int), hipMemcpyDeviceToHost, ctx->stream));
    ICP_HIP(ctx, hipStreamSynchronize(ctx->stream));
    ICP_HIP(ctx, hipMemcpyAsync(count_dev, side + 1, sizeof(int), hipMemcpyDeviceToDevice, ctx->stream));
    *count_host = v;
    if (v <= 0) return ICP_OK;
    size_t tmp_bytes = 0;
    ICP_HIP(ctx, rocprim::radix_sort_pairs(nullptr, tmp_bytes, ka, kb, va, vb, (size_t)v, 0, 64, ctx->stream));
    ICP_HIP(ctx, ctx->sort_tmp.reserve(tmp_bytes));
    ICP_HIP(ctx, rocprim::radix_sort_pairs(ctx->sort_tmp.ptr, tmp_bytes, ka, kb, va, vb, (size_t)v, 0, 64,
                                           ctx->stream));
    hipLaunchKernelGGL(k_emit_sorted<T>, dim3((unsigned)((v + 255) / 256)), dim3(256), 0, ctx->stream, xyz_dev, vb,
                       count_dev, indices_dev, points_dev);
    ICP_HIP(ctx, hipGetLastError());
    return ICP_OK;
}

int grid_sample_device(icp_ctx* ctx, const float* xyz_dev, int64_t n, double voxel, long long* indices_dev,
                       float* points_dev, int* count_dev, int* count_host) {
    return grid_sample_impl<float>(ctx, xyz_dev, n, voxel, indices_dev, points_dev, count_dev, count_host);
}

int grid_sample_f64_device(icp_ctx* ctx, const double* xyz_dev, int64_t n, double voxel, long long* indices_dev,
                           double* points_dev, int* count_dev, int* count_host) {
    return grid_sample_impl<double>(ctx, xyz_dev, n, voxel, indices_dev, points_dev, count_dev, count_host);
}

// ---------------------------------------------------------------------------------------------------------------------
// Per-voxel normal distribution (`voxel_normal_distribution`, slam/common/pointcloud.py:83-167; `Voxelization.filter`,
// slam/preprocessing.py:63-98): the points are ordered by voxel hash (same stable radix sort), every run of equal
// hashes is one voxel; voxel id = rank of the hash; per voxel the point count, the float32 mean and the float32
// UNNORMALISED covariance sum (p - mean)(p - mean)^T, accumulated in the sorted order like the numba loop.
// ---------------------------------------------------------------------------------------------------------------------
__global__ void k_voxel_ids(const int* __restrict__ vals, const int* __restrict__ flags, const int* __restrict__ offs,
                            int n, long long* __restrict__ ids, int* __restrict__ starts) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int id = offs[i] + flags[i] - 1;  // heads before i, plus this one if it is a head
    ids[vals[i]] = id;
    if (flags[i]) starts[id] = i;
}

__global__ void k_voxel_stats(const float* __restrict__ xyz, const int* __restrict__ vals,
                              const int* __restrict__ starts, const int* __restrict__ count_dev, int n,
                              long long* __restrict__ sizes, float* __restrict__ means, float* __restrict__ covs) {
    const int v = blockIdx.x * blockDim.x + threadIdx.x;
    const int nv = *count_dev;
    if (v >= nv) return;
    const int b = starts[v], e = (v + 1 < nv) ? starts[v + 1] : n;
    float sx = 0.f, sy = 0.f, sz = 0.f;
    for (int k = b; k < e; ++k) {
        const int i = vals[k];
        sx = __fadd_rn(sx, xyz[3 * i]);
        sy = __fadd_rn(sy, xyz[3 * i + 1]);
        sz = __fadd_rn(sz, xyz[3 * i + 2]);
    }
    const float cnt = (float)(e - b);
    const float mx = sx / cnt, my = sy / cnt, mz = sz / cnt;
    float c[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    for (int k = b; k < e; ++k) {
        const int i = vals[k];
        const float dx = __fsub_rn(xyz[3 * i], mx), dy = __fsub_rn(xyz[3 * i + 1], my), dz = __fsub_rn(xyz[3 * i + 2], mz);
        c[0] = __fadd_rn(c[0], __fmul_rn(dx, dx));
        c[1] = __fadd_rn(c[1], __fmul_rn(dx, dy));
        c[2] = __fadd_rn(c[2], __fmul_rn(dx, dz));
        c[3] = __fadd_rn(c[3], __fmul_rn(dy, dy));
        c[4] = __fadd_rn(c[4], __fmul_rn(dy, dz));
        c[5] = __fadd_rn(c[5], __fmul_rn(dz, dz));
    }
    sizes[v] = e - b;
    means[3 * v] = mx;
    means[3 * v + 1] = my;
    means[3 * v + 2] = mz;
    float* C = covs + 9 * (size_t)v;
    C[0] = c[0]; C[1] = c[1]; C[2] = c[2];
    C[3] = c[1]; C[4] = c[3]; C[5] = c[4];
    C[6] = c[2]; C[7] = c[4]; C[8] = c[5];
}

int voxel_statistics_device(icp_ctx* ctx, const float* xyz_dev, int64_t n, double voxel, long long* voxels_dev,
                            long long* hashes_dev, long long* ids_dev, long long* sizes_dev, float* means_dev,
                            float* covs_dev, int* count_dev) {
    if (n <= 0) {
        ICP_HIP(ctx, hipMemsetAsync(count_dev, 0, sizeof(int), ctx->stream));
        return ICP_OK;
    }
    ICP_HIP(ctx, ctx->keys_a.reserve((size_t)n * 8));
    ICP_HIP(ctx, ctx->keys_b.reserve((size_t)n * 8));
    ICP_HIP(ctx, ctx->vals_a.reserve((size_t)n * 4));
    ICP_HIP(ctx, ctx->vals_b.reserve((size_t)n * 4));
    ICP_HIP(ctx, ctx->flags.reserve((size_t)n * 4));
    ICP_HIP(ctx, ctx->scan_a.reserve((size_t)n * 4));
    ICP_HIP(ctx, ctx->scan_b.reserve((size_t)n * 4));
    unsigned long long* ka = ctx->keys_a.as<unsigned long long>();
    unsigned long long* kb = ctx->keys_b.as<unsigned long long>();
    int* va = ctx->vals_a.as<int>();
    int* vb = ctx->vals_b.as<int>();
    const unsigned nb = (unsigned)((n + 255) / 256);
    hipLaunchKernelGGL(k_voxel_hash<float>, dim3(nb), dim3(256), 0, ctx->stream, xyz_dev, (int)n, voxel, voxels_dev,
                       hashes_dev, ka, va);
    size_t tmp_bytes = 0;
    ICP_HIP(ctx, rocprim::radix_sort_pairs(nullptr, tmp_bytes, ka, kb, va, vb, (size_t)n, 0, 64, ctx->stream));
    ICP_HIP(ctx, ctx->sort_tmp.reserve(tmp_bytes));
    ICP_HIP(ctx, rocprim::radix_sort_pairs(ctx->sort_tmp.ptr, tmp_bytes, ka, kb, va, vb, (size_t)n, 0, 64,
                                           ctx->stream));
    int* flags = ctx->flags.as<int>();
    int* offs = ctx->scan_a.as<int>();
    int* starts = ctx->scan_b.as<int>();
    hipLaunchKernelGGL(k_run_heads, dim3(nb), dim3(256), 0, ctx->stream, kb, (int)n, flags);
    int rc = exclusive_scan_i32(ctx, flags, offs, n, count_dev);
    if (rc) return rc;
    hipLaunchKernelGGL(k_voxel_ids, dim3(nb), dim3(256), 0, ctx->stream, vb, flags, offs, (int)n, ids_dev, starts);
    if (sizes_dev)
        hipLaunchKernelGGL(k_voxel_stats, dim3(nb), dim3(256), 0, ctx->stream, xyz_dev, vb, starts, count_dev, (int)n,
                           sizes_dev, means_dev, covs_dev);
    ICP_HIP(ctx, hipGetLastError());
    return ICP_OK;
}

// ---------------------------------------------------------------------------------------------------------------------
// De-skew (`Distortion.filter`, slam/preprocessing.py:144-191): timestamp range by a two-level f64 min/max reduction,
// then per point the Rodrigues rotation by alpha * theta about the axis of the initial motion + alpha * translation,
// all in float64 like the reference (scipy Slerp + a float64 einsum).
// ---------------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_minmax_f64(const double* __restrict__ v, int n, double* __restrict__ part) {
    __shared__ double smin[4], smax[4];
    double mn = INFINITY, mx = -INFINITY;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        const double x = v[i];
        mn = fmin(mn, x);
        mx = fmax(mx, x);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        mn = fmin(mn, __shfl_down(mn, o, 64));
        mx = fmax(mx, __shfl_down(mx, o, 64));
    }
    if ((threadIdx.x & 63) == 0) {
        smin[threadIdx.x >> 6] = mn;
        smax[threadIdx.x >> 6] = mx;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        part[2 * blockIdx.x] = fmin(fmin(smin[0], smin[1]), fmin(smin[2], smin[3]));
        part[2 * blockIdx.x + 1] = fmax(fmax(smax[0], smax[1]), fmax(smax[2], smax[3]));
    }
}

struct DistortArg {
    double axis[3];
    double theta;
    double t[3];
};

__global__ void k_distort(const float* __restrict__ xyz, const double* __restrict__ ts, int n,
                          const double* __restrict__ part, int nparts, DistortArg a, double* __restrict__ out) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    double tmin = INFINITY, tmax = -INFINITY;
    for (int k = 0; k < nparts; ++k) {  // a few dozen values, L2-resident
        tmin = fmin(tmin, part[2 * k]);
        tmax = fmax(tmax, part[2 * k + 1]);
    }
    const double diff = tmax - tmin;
    const double alpha = diff == 0.0 ? 0.0 : (ts[i] - tmin) / diff;  // :177-179
    const double px = (double)xyz[3 * i], py = (double)xyz[3 * i + 1], pz = (double)xyz[3 * i + 2];
    double s, c;
    sincos(alpha * a.theta, &s, &c);
    const double ux = a.axis[0], uy = a.axis[1], uz = a.axis[2];
    const double dot = ux * px + uy * py + uz * pz;
    const double cx = uy * pz - uz * py, cy = uz * px - ux * pz, cz = ux * py - uy * px;
    out[3 * i] = px * c + cx * s + ux * dot * (1.0 - c) + alpha * a.t[0];
    out[3 * i + 1] = py * c + cy * s + uy * dot * (1.0 - c) + alpha * a.t[1];
    out[3 * i + 2] = pz * c + cz * s + uz * dot * (1.0 - c) + alpha * a.t[2];
}

int distort_device(icp_ctx* ctx, const float* xyz_dev, const double* ts_dev, int64_t n, const double* rel_pose16,
                   double* out_dev) {
    if (n <= 0) return ICP_OK;
    const int nparts = 64;
    ICP_HIP(ctx, ctx->scan_b.reserve((size_t)nparts * 2 * sizeof(double)));
    double* part = ctx->scan_b.as<double>();
    hipLaunchKernelGGL(k_minmax_f64, dim3(nparts), dim3(256), 0, ctx->stream, ts_dev, (int)n, part);
    // log map of the rotation: (R - R^T) / 2 = sin(theta) [axis]x, trace = 1 + 2 cos(theta)
    const double* R = rel_pose16;
    const double vx = 0.5 * (R[9] - R[6]), vy = 0.5 * (R[2] - R[8]), vz = 0.5 * (R[4] - R[1]);
    const double nv = sqrt(vx * vx + vy * vy + vz * vz);
    DistortArg a;
    a.theta = atan2(nv, 0.5 * (R[0] + R[5] + R[10] - 1.0));
    a.axis[0] = nv > 0 ? vx / nv : 0.0;
    a.axis[1] = nv > 0 ? vy / nv : 0.0;
    a.axis[2] = nv > 0 ? vz / nv : 0.0;
    a.t[0] = R[3];
    a.t[1] = R[7];
    a.t[2] = R[11];
    hipLaunchKernelGGL(k_distort, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, ctx->stream, xyz_dev, ts_dev, (int)n,
                       (const double*)part, nparts, a, out_dev);
    ICP_HIP(ctx, hipGetLastError());
    return ICP_OK;
}

// ---------------------------------------------------------------------------------------------------------------------
// Target preparation for a registration: [n,3] rows -> float4 (x, y, z, bits(row)), one aligned 16-byte load per target
// in the iteration kernels.
// ---------------------------------------------------------------------------------------------------------------------
__global__ void k_pack_targets(const float* __restrict__ xyz, int n, float4* __restrict__ out, RegState* st, Pose16 init,
                               int keep_pose, unsigned long long* box, unsigned gen, float* hist) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (st && i == 0) {
        state_init(st, init.m, keep_pose);  // the registration state, by the same launch
        box_publish_serial(box, gen, st->pose, 0, 0);  // ... and the initial guess as generation `gen` of the pose mailbox
        for (int k = 0; k < 12; ++k) hist[k] = st->pose[k];  // ... and as entry 0 of the pose history
    }
    if (i >= n) return;
    out[i] = make_float4(xyz[3 * i], xyz[3 * i + 1], xyz[3 * i + 2], __int_as_float(i));
}

int prepare_targets(icp_ctx* ctx, const float* xyz_dev, int64_t n, const Pose16* init, bool keep_pose) {
    ICP_HIP(ctx, ctx->tgt4.reserve((size_t)(n > 0 ? n : 1) * sizeof(float4)));
    if (n <= 0) return ICP_OK;
    Pose16 p;
    memset(p.m, 0, sizeof(p.m));
    if (init) p = *init;
    hipLaunchKernelGGL(k_pack_targets, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, ctx->stream, xyz_dev, (int)n,
                       ctx->tgt4.as<float4>(), init ? reg_state(ctx) : (RegState*)nullptr, p, keep_pose ? 1 : 0,
                       pose_box(ctx), init ? next_box_generation(ctx) : 0u, ctx->pose_hist);
    ICP_HIP(ctx, hipGetLastError());
    return ICP_OK;
}

}  // namespace icp
