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

// + (padded outputs) the 0xFF bytes behind the samples — NaN points, index -1: the same launch writes them (two memset
// launches less in front of every frame)
__global__ void k_dedupe_clear(DedupeSlot* __restrict__ table, unsigned int size, int* __restrict__ side,
                               unsigned* __restrict__ fill_a, unsigned long long words_a, unsigned* __restrict__ fill_b,
                               unsigned long long words_b) {
    const unsigned int i = blockIdx.x * blockDim.x + threadIdx.x;
    const unsigned long long stride = (unsigned long long)gridDim.x * blockDim.x;
    for (unsigned long long w = i; w < words_a; w += stride) fill_a[w] = 0xFFFFFFFFu;
    for (unsigned long long w = i; w < words_b; w += stride) fill_b[w] = 0xFFFFFFFFu;
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

// ---------------------------------------------------------------------------------------------------------------------
// Sort of the V distinct (hash, index) pairs + emission of the samples in ONE launch of ONE workgroup, V read on the
// device: no host round trip to size a library sort, no library kernels in the frame (five rocPRIM launches
// and two synchronisations per grid sample for ~6 000 pairs).  LSD radix sort over the bits in which the keys differ
// (key - min: a LiDAR frame's hashes span ~2^37, five 8-bit passes instead of eight), pairs ping-ponging between two
// global buffers (L2-resident), per-pass: every wave owns a contiguous chunk, counts its digits tile by tile (64 pairs; the
// lanes of a tile with the same digit found by eight ballots), one scan over (digit, wave) gives every wave its bases,
// the waves scatter their tiles in the same order — stable.  ~5 us per pass at V = 6 000; any V is sorted (a workgroup's
// worth of threads per pass: meant for thousands of voxels — the exact-shape entry point hands V > 32 768 to rocPRIM).
// ---------------------------------------------------------------------------------------------------------------------
static constexpr int SORT_THREADS = 1024;
static constexpr int SORT_WAVES = SORT_THREADS / 64;

template <typename T>
__global__ __launch_bounds__(SORT_THREADS) void k_sort_emit(unsigned long long* __restrict__ ka, int* __restrict__ va,
                                                            unsigned long long* __restrict__ kb, int* __restrict__ vb,
                                                            const int* __restrict__ side, const T* __restrict__ xyz,
                                                            long long* __restrict__ indices, T* __restrict__ points,
                                                            int* __restrict__ count_out) {
    __shared__ int hist[SORT_WAVES][256];
    __shared__ unsigned long long red[2][SORT_WAVES];
    __shared__ int wave_tot[SORT_WAVES];
    const int V = side[1];
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    if (count_out && tid == 0) *count_out = V;
    if (V <= 0) return;
    // ---- the bits in which the keys differ
    unsigned long long kmin = ~0ull, kmax = 0ull;
    for (int i = tid; i < V; i += SORT_THREADS) {
        const unsigned long long k = ka[i];
        kmin = k < kmin ? k : kmin;
        kmax = k > kmax ? k : kmax;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const unsigned long long a = ((unsigned long long)__shfl_xor((unsigned)(kmin >> 32), o, 64) << 32) | __shfl_xor((unsigned)kmin, o, 64);
        const unsigned long long b = ((unsigned long long)__shfl_xor((unsigned)(kmax >> 32), o, 64) << 32) | __shfl_xor((unsigned)kmax, o, 64);
        kmin = a < kmin ? a : kmin;
        kmax = b > kmax ? b : kmax;
    }
    if (lane == 0) {
        red[0][w] = kmin;
        red[1][w] = kmax;
    }
    __syncthreads();
    for (int k = 0; k < SORT_WAVES; ++k) {
        kmin = red[0][k] < kmin ? red[0][k] : kmin;
        kmax = red[1][k] > kmax ? red[1][k] : kmax;
    }
    const unsigned long long range = kmax - kmin;
    const int bits = range ? 64 - __clzll((long long)range) : 0, passes = (bits + 7) / 8;
    // every wave's chunk: a whole number of 64-pair tiles
    const int tiles_total = (V + 63) / 64, tiles_per_wave = (tiles_total + SORT_WAVES - 1) / SORT_WAVES;
    const int first = w * tiles_per_wave * 64;
    int last = first + tiles_per_wave * 64;
    last = last < V ? last : V;
    unsigned long long* src_k = ka;
    unsigned long long* dst_k = kb;
    int* src_v = va;
    int* dst_v = vb;
    const unsigned long long below = lane ? (~0ull >> (64 - lane)) : 0ull;  // lanes in front of this one
    constexpr int REG_TILES = 8;  // up to 8 x 64 pairs per wave (V <= 8192) stay in registers through a pass
    const bool in_regs = tiles_per_wave <= REG_TILES;
    for (int p = 0; p < passes; ++p) {
        const int shift = 8 * p;
        for (int k = tid; k < SORT_WAVES * 256; k += SORT_THREADS) (&hist[0][0])[k] = 0;
        if (in_regs) {
            // every pair of the wave's chunk is requested at once — ONE memory latency per pass instead of one per tile and
            // phase (six tiles, two phases: 83 us for 6 000 pairs in five passes, measured) — and kept for the scatter
            unsigned long long key[REG_TILES];
            int val[REG_TILES];
#pragma unroll
            for (int t = 0; t < REG_TILES; ++t) {
                const int i = first + t * 64 + lane;
                key[t] = 0ull;
                val[t] = 0;
                if (i < last) {
                    key[t] = src_k[i];
                    val[t] = src_v[i];
                }
            }
            __syncthreads();
            unsigned long long same[REG_TILES];
#pragma unroll
            for (int t = 0; t < REG_TILES; ++t) {
                const bool valid = first + t * 64 + lane < last;
                const unsigned d = (unsigned)(((key[t] - kmin) >> shift) & 255ull);
                unsigned long long m = __ballot(valid);
#pragma unroll
                for (int b = 0; b < 8; ++b) {
                    const unsigned long long bb = __ballot(valid && ((d >> b) & 1u));
                    m &= ((d >> b) & 1u) ? bb : ~bb;
                }
                same[t] = valid ? m : 0ull;
                if (valid && (m & below) == 0ull) hist[w][d] += __popcll(m);
            }
            __syncthreads();
            {
                int v4[4], sum = 0;
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const int L = 4 * tid + k;
                    v4[k] = hist[L % SORT_WAVES][L / SORT_WAVES];
                    sum += v4[k];
                }
                int incl = sum;
#pragma unroll
                for (int o = 1; o < 64; o <<= 1) {
                    const int t2 = __shfl_up(incl, o, 64);
                    if (lane >= o) incl += t2;
                }
                if (lane == 63) wave_tot[w] = incl;
                __syncthreads();
                int off = incl - sum;
                for (int k = 0; k < w; ++k) off += wave_tot[k];
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const int L = 4 * tid + k;
                    hist[L % SORT_WAVES][L / SORT_WAVES] = off;
                    off += v4[k];
                }
            }
            __syncthreads();
#pragma unroll
            for (int t = 0; t < REG_TILES; ++t) {
                const bool valid = same[t] != 0ull;
                const unsigned d = (unsigned)(((key[t] - kmin) >> shift) & 255ull);
                const int start = valid ? hist[w][d] : 0;
                const int rank = __popcll(same[t] & below);
                if (valid) {
                    dst_k[start + rank] = key[t];
                    dst_v[start + rank] = val[t];
                    if (rank == 0) hist[w][d] = start + __popcll(same[t]);
                }
            }
            __syncthreads();
            unsigned long long* tk = src_k;
            src_k = dst_k;
            dst_k = tk;
            int* tv = src_v;
            src_v = dst_v;
            dst_v = tv;
            continue;
        }
        __syncthreads();
        for (int base = first; base < last; base += 64) {  // (wave-uniform bounds)
            const int i = base + lane;
            const bool valid = i < last;
            const unsigned d = valid ? (unsigned)(((src_k[i] - kmin) >> shift) & 255ull) : 0u;
            unsigned long long same = __ballot(valid);
#pragma unroll
            for (int b = 0; b < 8; ++b) {
                const unsigned long long bb = __ballot(valid && ((d >> b) & 1u));
                same &= ((d >> b) & 1u) ? bb : ~bb;
            }
            if (valid && (same & below) == 0ull) hist[w][d] += __popcll(same);  // the first lane of every digit group
        }
        __syncthreads();
        // exclusive scan over (digit major, wave minor): linear index L = d * SORT_WAVES + wave, four consecutive L per thread
        {
            int v4[4], sum = 0;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int L = 4 * tid + k;
                v4[k] = hist[L % SORT_WAVES][L / SORT_WAVES];
                sum += v4[k];
            }
            int incl = sum;
#pragma unroll
            for (int o = 1; o < 64; o <<= 1) {
                const int t = __shfl_up(incl, o, 64);
                if (lane >= o) incl += t;
            }
            if (lane == 63) wave_tot[w] = incl;
            __syncthreads();
            int off = incl - sum;
            for (int k = 0; k < w; ++k) off += wave_tot[k];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int L = 4 * tid + k;
                hist[L % SORT_WAVES][L / SORT_WAVES] = off;
                off += v4[k];
            }
        }
        __syncthreads();
        for (int base = first; base < last; base += 64) {
            const int i = base + lane;
            const bool valid = i < last;
            unsigned long long key = 0ull;
            int val = 0;
            if (valid) {
                key = src_k[i];
                val = src_v[i];
            }
            const unsigned d = valid ? (unsigned)(((key - kmin) >> shift) & 255ull) : 0u;
            unsigned long long same = __ballot(valid);
#pragma unroll
            for (int b = 0; b < 8; ++b) {
                const unsigned long long bb = __ballot(valid && ((d >> b) & 1u));
                same &= ((d >> b) & 1u) ? bb : ~bb;
            }
            const int start = valid ? hist[w][d] : 0;  // (every lane of the group reads it before its first lane moves it on)
            const int rank = __popcll(same & below);
            if (valid) {
                dst_k[start + rank] = key;
                dst_v[start + rank] = val;
                if (rank == 0) hist[w][d] = start + __popcll(same);
            }
        }
        __syncthreads();  // (the pairs written above are read by other waves in the next pass: workgroup-scope release / acquire)
        unsigned long long* tk = src_k;
        src_k = dst_k;
        dst_k = tk;
        int* tv = src_v;
        src_v = dst_v;
        dst_v = tv;
    }
    // ---- the samples, by ascending hash: original index + gathered point
    for (int o = tid; o < V; o += SORT_THREADS) {
        const int src = src_v[o];
        if (indices) indices[o] = src;
        if (points) {
            points[3 * o] = xyz[3 * src];
            points[3 * o + 1] = xyz[3 * src + 1];
            points[3 * o + 2] = xyz[3 * src + 2];
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// The same result from BUCKET_BLOCKS workgroups in one launch (round 5: the one-workgroup radix sort above is a chain of five
// passes of load -> count -> scan -> scatter, 72 us for 6 000 pairs, whatever is kept in registers).  Every workgroup reads
// ALL V pairs (72 KB from L2), finds the keys' range, keeps the pairs of ITS slice of that range — slice b = (key - min)
// >> shift, monotone in the key — in LDS and counts the pairs of the slices below it; a member's place in the output is
// that count + its rank among the members (a compare against every member: a slice holds V / 64 pairs, ~100).  No
// ordering between workgroups, no atomics on global memory, V read on the device.  A slice that outgrows the LDS list (a
// degenerate key distribution) is ranked against all V keys instead: slow, correct.
// ---------------------------------------------------------------------------------------------------------------------
static constexpr int BUCKET_BLOCKS = 64;
static constexpr int BUCKET_THREADS = 512;
static constexpr int BUCKET_CAP = 4096;

template <typename T>
__global__ __launch_bounds__(BUCKET_THREADS) void k_bucket_sort_emit(const unsigned long long* __restrict__ keys,
                                                                     const int* __restrict__ vals,
                                                                     const int* __restrict__ side, const T* __restrict__ xyz,
                                                                     long long* __restrict__ indices, T* __restrict__ points,
                                                                     int* __restrict__ count_out) {
    __shared__ unsigned long long mk[BUCKET_CAP];
    __shared__ int mv[BUCKET_CAP];
    __shared__ unsigned long long red[2][BUCKET_THREADS / 64];
    __shared__ int lower_s[BUCKET_THREADS / 64];
    __shared__ int members;
    const int V = side[1];
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, me = blockIdx.x;
    if (count_out && me == 0 && tid == 0) *count_out = V;
    if (V <= 0) return;
    if (tid == 0) members = 0;
    unsigned long long kmin = ~0ull, kmax = 0ull;
    for (int i = tid; i < V; i += BUCKET_THREADS) {
        const unsigned long long k = keys[i];
        kmin = k < kmin ? k : kmin;
        kmax = k > kmax ? k : kmax;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const unsigned long long a = ((unsigned long long)__shfl_xor((unsigned)(kmin >> 32), o, 64) << 32) | __shfl_xor((unsigned)kmin, o, 64);
        const unsigned long long b = ((unsigned long long)__shfl_xor((unsigned)(kmax >> 32), o, 64) << 32) | __shfl_xor((unsigned)kmax, o, 64);
        kmin = a < kmin ? a : kmin;
        kmax = b > kmax ? b : kmax;
    }
    if (lane == 0) {
        red[0][w] = kmin;
        red[1][w] = kmax;
    }
    __syncthreads();
    for (int k = 0; k < BUCKET_THREADS / 64; ++k) {
        kmin = red[0][k] < kmin ? red[0][k] : kmin;
        kmax = red[1][k] > kmax ? red[1][k] : kmax;
    }
    const unsigned long long range = kmax - kmin;
    const int bits = range ? 64 - __clzll((long long)range) : 0;
    const int shift = bits > 6 ? bits - 6 : 0;  // (range >> shift) < 64 = BUCKET_BLOCKS slices
    // ---- my slice's members into LDS, the pairs of lower slices counted
    int lower = 0;
    for (int i = tid; i < V; i += BUCKET_THREADS) {
        const unsigned long long k = keys[i];
        const int b = (int)((k - kmin) >> shift);
        lower += b < me ? 1 : 0;
        if (b == me) {
            const int slot = atomicAdd(&members, 1);
            if (slot < BUCKET_CAP) {
                mk[slot] = k;
                mv[slot] = vals[i];
            }
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) lower += __shfl_xor(lower, o, 64);
    if (lane == 0) lower_s[w] = lower;
    __syncthreads();
    int base = 0;
    for (int k = 0; k < BUCKET_THREADS / 64; ++k) base += lower_s[k];
    const int m = members;
    if (m <= BUCKET_CAP) {
        for (int i = tid; i < m; i += BUCKET_THREADS) {
            const unsigned long long k = mk[i];
            int rank = 0;
            for (int j = 0; j < m; ++j) rank += mk[j] < k ? 1 : 0;  // (the keys are distinct hashes)
            const int o = base + rank, src = mv[i];
            if (indices) indices[o] = src;
            if (points) {
                points[3 * o] = xyz[3 * src];
                points[3 * o + 1] = xyz[3 * src + 1];
                points[3 * o + 2] = xyz[3 * src + 2];
            }
        }
        return;
    }
    // ---- a slice larger than the list (never seen on LiDAR frames): every member ranked against all V keys
    for (int i = tid; i < V; i += BUCKET_THREADS) {
        const unsigned long long k = keys[i];
        if ((int)((k - kmin) >> shift) != me) continue;
        int rank = 0;
        for (int j = 0; j < V; ++j) rank += keys[j] < k ? 1 : 0;
        const int src = vals[i];
        if (indices) indices[rank] = src;
        if (points) {
            points[3 * rank] = xyz[3 * src];
            points[3 * rank + 1] = xyz[3 * src + 1];
            points[3 * rank + 2] = xyz[3 * src + 2];
        }
    }
}

// padded = true: nothing is read back — the outputs hold n rows, the V samples first, NaN points / index -1 behind them
// (0xFF bytes, written by the first launch), *count_dev = V on the device only
template <typename T>
static int grid_sample_impl(icp_ctx* ctx, const T* xyz_dev, int64_t n, double voxel, long long* indices_dev,
                            T* points_dev, int* count_dev, int* count_host, bool padded = false) {
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
    // (padded: the outputs hold n rows — NaN points / index -1 behind the samples the sort below emits)
    hipLaunchKernelGGL(k_dedupe_clear, dim3((tsize + 255) / 256), dim3(256), 0, ctx->stream, table, tsize, side,
                       padded ? reinterpret_cast<unsigned*>(points_dev) : nullptr,
                       padded && points_dev ? (unsigned long long)n * 3 * (sizeof(T) / 4) : 0ull,
                       padded ? reinterpret_cast<unsigned*>(indices_dev) : nullptr,
                       padded && indices_dev ? (unsigned long long)n * 2 : 0ull);
    hipLaunchKernelGGL(k_hash_dedupe<T>, dim3(nb), dim3(256), 0, ctx->stream, xyz_dev, (int)n, voxel, table, tsize - 1,
                       side);
    hipLaunchKernelGGL(k_hash_collect, dim3(COLLECT_BLOCKS), dim3(COLLECT_THREADS), 0, ctx->stream, table, tsize, side,
                       ka, va);
    if (padded) {  // the device-resident pipeline: one more launch, nothing read back
        if (n <= 262144)  // (every workgroup reads all pairs: thousands of voxels — a frame; beyond: the one-workgroup radix sort)
            hipLaunchKernelGGL(k_bucket_sort_emit<T>, dim3(BUCKET_BLOCKS), dim3(BUCKET_THREADS), 0, ctx->stream, ka, va, side,
                               xyz_dev, indices_dev, points_dev, count_dev);
        else
            hipLaunchKernelGGL(k_sort_emit<T>, dim3(1), dim3(SORT_THREADS), 0, ctx->stream, ka, va, kb, vb, side, xyz_dev,
                               indices_dev, points_dev, count_dev);
        ICP_HIP(ctx, hipGetLastError());
        *count_host = -1;
        return ICP_OK;
    }
    // exact shapes: the host hands back V rows, so it needs V (one round trip; it also picks the sort)
    int v = 0;
    ICP_HIP(ctx, hipMemcpyAsync(&v, side + 1, sizeof(int), hipMemcpyDeviceToHost, ctx->stream));
    ICP_HIP(ctx, hipStreamSynchronize(ctx->stream));
    *count_host = v;
    if (v <= 32768) {  // thousands of voxels: the in-tree bucket sort (no library launches in the frame)
        hipLaunchKernelGGL(k_bucket_sort_emit<T>, dim3(BUCKET_BLOCKS), dim3(BUCKET_THREADS), 0, ctx->stream, ka, va, side,
                           xyz_dev, indices_dev, points_dev, count_dev);
        ICP_HIP(ctx, hipGetLastError());
        return ICP_OK;
    }
    ICP_HIP(ctx, hipMemcpyAsync(count_dev, side + 1, sizeof(int), hipMemcpyDeviceToDevice, ctx->stream));
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
                       float* points_dev, int* count_dev, int* count_host, bool padded) {
    return grid_sample_impl<float>(ctx, xyz_dev, n, voxel, indices_dev, points_dev, count_dev, count_host, padded);
}

int grid_sample_f64_device(icp_ctx* ctx, const double* xyz_dev, int64_t n, double voxel, long long* indices_dev,
                           double* points_dev, int* count_dev, int* count_host, bool padded) {
    return grid_sample_impl<double>(ctx, xyz_dev, n, voxel, indices_dev, points_dev, count_dev, count_host, padded);
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
    // the registration state, the initial guess as generation `gen` of the pose mailbox and as entry 0 of the pose history,
    // by the first wave of the same launch
    if (st && i < 64) state_init_wave(st, init.m, keep_pose, box, gen, hist, i);
    if (i >= n) return;
    out[i] = make_float4(xyz[3 * i], xyz[3 * i + 1], xyz[3 * i + 2], __int_as_float(i));
}

// ... for B registrations in one launch (icp_batch_register_launch): blockIdx.y = the member, its arguments in a table in
// device memory
__global__ void k_pack_targets_batch(const PackDesc* __restrict__ table) {
    const PackDesc& d = table[blockIdx.y];
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (d.st && i < 64) state_init_wave(d.st, d.init.m, d.keep_pose, d.box, d.gen, d.hist, i);
    if (i >= d.n) return;
    d.out[i] = make_float4(d.xyz[3 * i], d.xyz[3 * i + 1], d.xyz[3 * i + 2], __int_as_float(i));
}

int prepare_targets(icp_ctx* ctx, const float* xyz_dev, int64_t n, const Pose16* init, bool keep_pose, PackDesc* defer) {
    ICP_HIP(ctx, ctx->tgt4.reserve((size_t)(n > 0 ? n : 1) * sizeof(float4)));
    if (n <= 0) return ICP_OK;
    PackDesc d;
    memset(&d, 0, sizeof(d));
    if (init) d.init = *init;
    d.xyz = xyz_dev;
    d.n = (int)n;
    d.out = ctx->tgt4.as<float4>();
    d.st = init ? reg_state(ctx) : (RegState*)nullptr;
    d.keep_pose = keep_pose ? 1 : 0;
    d.box = pose_box(ctx);
    d.gen = init ? next_box_generation(ctx) : 0u;
    d.hist = ctx->pose_hist;
    if (defer) {  // the caller launches (B registrations per launch)
        *defer = d;
        return ICP_OK;
    }
    hipLaunchKernelGGL(k_pack_targets, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, ctx->stream, d.xyz, d.n, d.out, d.st,
                       d.init, d.keep_pose, d.box, d.gen, d.hist);
    ICP_HIP(ctx, hipGetLastError());
    return ICP_OK;
}

int launch_pack_targets_batch(icp_ctx* first, const PackDesc* table_host, const PackDesc* table_dev, int count) {
    int max_n = 0;
    for (int b = 0; b < count; ++b) max_n = table_host[b].n > max_n ? table_host[b].n : max_n;
    if (max_n <= 0) return ICP_OK;
    hipLaunchKernelGGL(k_pack_targets_batch, dim3((unsigned)((max_n + 255) / 256), count), dim3(256), 0, first->stream, table_dev);
    ICP_HIP(first, hipGetLastError());
    return ICP_OK;
}

}  // namespace icp
