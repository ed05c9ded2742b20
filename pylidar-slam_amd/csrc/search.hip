// Exact nearest-neighbour search in the voxel-hash grid + lazy kNN normal estimation.
//
// Replaces, per ICP iteration, `KdTreeLocalMap.nearest_neighbor_search` (slam/odometry/local_map.py:372-395, i.e.
// `pykdtree.KDTree.query`, k = 1, NO distance cap) and `__get_normals` (:397-422: k+1 NN of every hit map point, first
// one dropped, covariance centred on the point, smallest singular vector).
//
// Exactness: rings of cells (Chebyshev distance r around the query's cell) are visited until the best distance found
// is provably not beaten by anything outside the visited block: best <= r*h + min_axis(min(f, h - f)), f = offset of
// the query inside its cell.  Cells whose box is farther than the current best are skipped without a table probe.
// Queries that exhaust `max_rings` fall back to an exhaustive scan, so the result is the exact NN for every input.
// Distance ties are broken on the smaller original map index (the kd-tree's tie order is unspecified).
#include "icp_internal.h"

namespace icp {

struct Best {
    float d2;
    int idx;  // original index (tie-break)
    int pos;  // cell-sorted position
};

__device__ inline bool better(float d2, int idx, float bd2, int bidx) { return d2 < bd2 || (d2 == bd2 && idx < bidx); }

__device__ inline bool grid_lookup(const GridView& g, int cx, int cy, int cz, int& start, int& count) {
    const unsigned long long key = pack_cell(cx, cy, cz);
    unsigned int slot = hash_cell(key) & g.mask;
    while (true) {
        const GridEntry e = g.table[slot];
        if (e.key == key) {
            start = e.start;
            count = e.count;
            return true;
        }
        if (e.key == GRID_EMPTY) return false;
        slot = (slot + 1) & g.mask;
    }
}

// squared distance from the query (offset f inside its own cell, per axis) to the box of the cell at offset o
__device__ inline float axis_gap(int o, float f, float h) {
    if (o == 0) return 0.f;
    return o < 0 ? f + (float)(-o - 1) * h : (h - f) + (float)(o - 1) * h;
}

__device__ inline void consider(const float4 q, int pos, float px, float py, float pz, Best& b) {
    const float dx = q.x - px, dy = q.y - py, dz = q.z - pz;
    const float d2 = fmaf(dz, dz, fmaf(dy, dy, dx * dx));
    const int idx = __float_as_int(q.w);
    if (better(d2, idx, b.d2, b.idx)) {
        b.d2 = d2;
        b.idx = idx;
        b.pos = pos;
    }
}

// candidates are fetched four at a time (independent 16-byte loads in flight together); the tail re-reads the last
// point of the cell, which cannot change the (d2, index) minimum
__device__ inline void scan_cell_1nn(const GridView& g, int start, int count, float px, float py, float pz, Best& b) {
    const int last = start + count - 1;
    for (int k = start; k <= last; k += 4) {
        const int k1 = min(k + 1, last), k2 = min(k + 2, last), k3 = min(k + 3, last);
        const float4 q0 = g.pts[k], q1 = g.pts[k1], q2 = g.pts[k2], q3 = g.pts[k3];
        consider(q0, k, px, py, pz, b);
        consider(q1, k1, px, py, pz, b);
        consider(q2, k2, px, py, pz, b);
        consider(q3, k3, px, py, pz, b);
    }
}

__device__ inline Best nearest_in_grid(const GridView& g, float px, float py, float pz, int max_rings) {
    Best b;
    b.d2 = INFINITY;
    b.idx = 0x7fffffff;
    b.pos = -1;
    const int cx = cell_coord(px, g.inv_h), cy = cell_coord(py, g.inv_h), cz = cell_coord(pz, g.inv_h);
    const float h = g.h;
    const float fx = fminf(fmaxf(px - (float)cx * h, 0.f), h);
    const float fy = fminf(fmaxf(py - (float)cy * h, 0.f), h);
    const float fz = fminf(fmaxf(pz - (float)cz * h, 0.f), h);
    const float edge = fminf(fminf(fminf(fx, h - fx), fminf(fy, h - fy)), fminf(fz, h - fz));
    int start, count;
    if (grid_lookup(g, cx, cy, cz, start, count)) scan_cell_1nn(g, start, count, px, py, pz, b);
    for (int r = 1; r <= max_rings; ++r) {
        for (int oz = -r; oz <= r; ++oz) {
            const float gz = axis_gap(oz, fz, h);
            const float gz2 = gz * gz;
            if (gz2 > b.d2) continue;
            const int az = oz < 0 ? -oz : oz;
            for (int oy = -r; oy <= r; ++oy) {
                const float gy = axis_gap(oy, fy, h);
                const float gyz2 = fmaf(gy, gy, gz2);
                if (gyz2 > b.d2) continue;
                const int ay = oy < 0 ? -oy : oy;
                const bool shell_yz = (az == r) || (ay == r);
                // on the shell in y/z every x is visited, otherwise only x = -r and x = +r
                const int step = shell_yz ? 1 : 2 * r;
                for (int ox = -r; ox <= r; ox += step) {
                    const float gx = axis_gap(ox, fx, h);
                    if (fmaf(gx, gx, gyz2) > b.d2) continue;
                    if (grid_lookup(g, cx + ox, cy + oy, cz + oz, start, count))
                        scan_cell_1nn(g, start, count, px, py, pz, b);
                }
            }
        }
        const float bound = (float)r * h + edge;
        if (b.d2 <= bound * bound * 0.999999f) return b;
    }
    // exhaustive fallback: keeps the search exact for queries farther than max_rings cells from the map
    b.d2 = INFINITY;
    b.idx = 0x7fffffff;
    b.pos = -1;
    scan_cell_1nn(g, 0, g.m, px, py, pz, b);
    return b;
}

__device__ inline bool target_valid(float x, float y, float z, int mode) {
    if (!(x == x) || !(y == y) || !(z == z)) return false;  // remove_nan, icp_odometry.py:357
    if (mode == ICP_TARGETS_SKIP_NULL && x == 0.f && y == 0.f && z == 0.f) return false;  // :303-305
    return true;
}

// p' = p R^T + t  (Pose.apply_transformation, slam/common/pose.py:169-186)
__device__ inline void transform_point(const float* __restrict__ T, float x, float y, float z, float& px, float& py,
                                       float& pz) {
    px = fmaf(z, T[2], fmaf(y, T[1], x * T[0])) + T[3];
    py = fmaf(z, T[6], fmaf(y, T[5], x * T[4])) + T[7];
    pz = fmaf(z, T[10], fmaf(y, T[9], x * T[8])) + T[11];
}

// ---------------------------------------------------------------------------------------------------------------------
// K1: transform + exact 1-NN; queue the hit map points that have no normal yet
// ---------------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_search(GridView g, const float* __restrict__ tgt, int n, int mode,
                                                int transform, RegState* __restrict__ st, int max_rings,
                                                int* __restrict__ nn_pos, int* __restrict__ nflag,
                                                int* __restrict__ worklist, int queue_normals) {
    if (st->done) return;
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float x = tgt[3 * i + 0], y = tgt[3 * i + 1], z = tgt[3 * i + 2];
    if (!target_valid(x, y, z, mode)) {
        nn_pos[i] = -1;
        return;
    }
    float px = x, py = y, pz = z;
    if (transform) transform_point(st->pose, x, y, z, px, py, pz);
    const Best b = nearest_in_grid(g, px, py, pz, max_rings);
    nn_pos[i] = b.pos;
    if (queue_normals && b.pos >= 0 && nflag[b.pos] == 0) {
        if (atomicCAS(&nflag[b.pos], 0, 2) == 0) worklist[atomicAdd(&st->n_worklist, 1)] = b.pos;
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// K2: kNN normals for the queued map points
// ---------------------------------------------------------------------------------------------------------------------
template <int KN>
struct TopK {
    float d2[KN];
    int idx[KN];
    int pos[KN];
    __device__ inline void init() {
#pragma unroll
        for (int k = 0; k < KN; ++k) {
            d2[k] = INFINITY;
            idx[k] = 0x7fffffff;
            pos[k] = -1;
        }
    }
    __device__ inline void insert(float d, int i, int p) {
        if (!better(d, i, d2[KN - 1], idx[KN - 1])) return;
        d2[KN - 1] = d;
        idx[KN - 1] = i;
        pos[KN - 1] = p;
#pragma unroll
        for (int k = KN - 1; k > 0; --k) {
            const bool sw = better(d2[k], idx[k], d2[k - 1], idx[k - 1]);
            const float td = d2[k];
            const int ti = idx[k], tp = pos[k];
            d2[k] = sw ? d2[k - 1] : td;
            idx[k] = sw ? idx[k - 1] : ti;
            pos[k] = sw ? pos[k - 1] : tp;
            d2[k - 1] = sw ? td : d2[k - 1];
            idx[k - 1] = sw ? ti : idx[k - 1];
            pos[k - 1] = sw ? tp : pos[k - 1];
        }
    }
};

template <int KN>
__device__ inline void scan_cell_knn(const GridView& g, int start, int count, float px, float py, float pz,
                                     TopK<KN>& t) {
    for (int k = 0; k < count; ++k) {
        const float4 q = g.pts[start + k];
        const float dx = q.x - px, dy = q.y - py, dz = q.z - pz;
        t.insert(fmaf(dz, dz, fmaf(dy, dy, dx * dx)), __float_as_int(q.w), start + k);
    }
}

// smallest-eigenvalue eigenvector of a symmetric 3x3 (cyclic Jacobi in f64).  The reference takes vh[2] of an f32
// LAPACK SVD of the same matrix (local_map.py:414-416); for a symmetric PSD matrix that is this eigenvector up to sign,
// and the sign cancels in J^T J and J^T r.
__device__ inline void smallest_eigenvector(double a00, double a01, double a02, double a11, double a12, double a22,
                                            float& nx, float& ny, float& nz) {
    double A[3][3] = {{a00, a01, a02}, {a01, a11, a12}, {a02, a12, a22}};
    double V[3][3] = {{1, 0, 0}, {0, 1, 0}, {0, 0, 1}};
    const double scale = fabs(a00) + fabs(a11) + fabs(a22);
    for (int sweep = 0; sweep < 12; ++sweep) {
        const double off = fabs(A[0][1]) + fabs(A[0][2]) + fabs(A[1][2]);
        if (off <= 1e-17 * scale || off == 0.0) break;
#pragma unroll
        for (int pq = 0; pq < 3; ++pq) {
            const int p = pq == 2 ? 1 : 0;
            const int q = pq == 0 ? 1 : 2;
            const double apq = A[p][q];
            if (apq == 0.0) continue;
            const double theta = (A[q][q] - A[p][p]) / (2.0 * apq);
            const double t = (theta >= 0 ? 1.0 : -1.0) / (fabs(theta) + sqrt(theta * theta + 1.0));
            const double c = 1.0 / sqrt(t * t + 1.0), s = t * c;
#pragma unroll
            for (int k = 0; k < 3; ++k) {  // A <- A J
                const double akp = A[k][p], akq = A[k][q];
                A[k][p] = c * akp - s * akq;
                A[k][q] = s * akp + c * akq;
            }
#pragma unroll
            for (int k = 0; k < 3; ++k) {  // A <- J^T A
                const double apk = A[p][k], aqk = A[q][k];
                A[p][k] = c * apk - s * aqk;
                A[q][k] = s * apk + c * aqk;
            }
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                const double vkp = V[k][p], vkq = V[k][q];
                V[k][p] = c * vkp - s * vkq;
                V[k][q] = s * vkp + c * vkq;
            }
        }
    }
    int m = 2;  // ties -> the last axis, like vh[2] of an already diagonal input
    if (A[1][1] < A[m][m]) m = 1;
    if (A[0][0] < A[m][m]) m = 0;
    const double x = V[0][m], y = V[1][m], z = V[2][m];
    const double inv = 1.0 / sqrt(x * x + y * y + z * z);
    nx = (float)(x * inv);
    ny = (float)(y * inv);
    nz = (float)(z * inv);
}

template <int KN>
__device__ inline void estimate_normal(const GridView& g, int s, int max_rings, float4* __restrict__ normals,
                                       int* __restrict__ nflag) {
    const float4 P = g.pts[s];
    const float px = P.x, py = P.y, pz = P.z;
    TopK<KN> t;
    t.init();
    const int cx = cell_coord(px, g.inv_h), cy = cell_coord(py, g.inv_h), cz = cell_coord(pz, g.inv_h);
    const float h = g.h;
    const float fx = fminf(fmaxf(px - (float)cx * h, 0.f), h);
    const float fy = fminf(fmaxf(py - (float)cy * h, 0.f), h);
    const float fz = fminf(fmaxf(pz - (float)cz * h, 0.f), h);
    const float edge = fminf(fminf(fminf(fx, h - fx), fminf(fy, h - fy)), fminf(fz, h - fz));
    int start, count;
    if (grid_lookup(g, cx, cy, cz, start, count)) scan_cell_knn<KN>(g, start, count, px, py, pz, t);
    bool exact = false;
    for (int r = 1; r <= max_rings && !exact; ++r) {
        for (int oz = -r; oz <= r; ++oz) {
            const float gz = axis_gap(oz, fz, h);
            const float gz2 = gz * gz;
            if (gz2 > t.d2[KN - 1]) continue;
            const int az = oz < 0 ? -oz : oz;
            for (int oy = -r; oy <= r; ++oy) {
                const float gy = axis_gap(oy, fy, h);
                const float gyz2 = fmaf(gy, gy, gz2);
                if (gyz2 > t.d2[KN - 1]) continue;
                const int ay = oy < 0 ? -oy : oy;
                const int step = ((az == r) || (ay == r)) ? 1 : 2 * r;
                for (int ox = -r; ox <= r; ox += step) {
                    const float gx = axis_gap(ox, fx, h);
                    if (fmaf(gx, gx, gyz2) > t.d2[KN - 1]) continue;
                    if (grid_lookup(g, cx + ox, cy + oy, cz + oz, start, count))
                        scan_cell_knn<KN>(g, start, count, px, py, pz, t);
                }
            }
        }
        const float bound = (float)r * h + edge;
        exact = t.d2[KN - 1] <= bound * bound * 0.999999f;
    }
    if (!exact) {
        t.init();
        scan_cell_knn<KN>(g, 0, g.m, px, py, pz, t);
    }
    // covariance of the k neighbours (first of the k+1 dropped, :407) centred on the query point (:411-413), f32
    float c00 = 0, c01 = 0, c02 = 0, c11 = 0, c12 = 0, c22 = 0;
    int used = 0;
#pragma unroll
    for (int k = 1; k < KN; ++k) {
        if (t.pos[k] < 0) continue;  // map smaller than k + 1 points
        const float4 q = g.pts[t.pos[k]];
        const float dx = q.x - px, dy = q.y - py, dz = q.z - pz;
        c00 += dx * dx;
        c01 += dx * dy;
        c02 += dx * dz;
        c11 += dy * dy;
        c12 += dy * dz;
        c22 += dz * dz;
        ++used;
    }
    const float invk = used > 0 ? 1.0f / (float)used : 0.f;
    float nx, ny, nz;
    smallest_eigenvector((double)(c00 * invk), (double)(c01 * invk), (double)(c02 * invk), (double)(c11 * invk),
                         (double)(c12 * invk), (double)(c22 * invk), nx, ny, nz);
    normals[s] = make_float4(nx, ny, nz, 1.f);
    nflag[s] = 1;
}

// lazy: the map points queued by the search of this iteration
template <int KN>
__global__ __launch_bounds__(128) void k_normals(GridView g, RegState* __restrict__ st,
                                                 const int* __restrict__ worklist, int max_rings,
                                                 float4* __restrict__ normals, int* __restrict__ nflag) {
    if (st->done) return;
    const int nw = st->n_worklist;
    for (int w = blockIdx.x * blockDim.x + threadIdx.x; w < nw; w += gridDim.x * blockDim.x)
        estimate_normal<KN>(g, worklist[w], max_rings, normals, nflag);
    if (blockIdx.x == 0 && threadIdx.x == 0) atomicAdd((unsigned long long*)&st->normals_computed, (unsigned long long)nw);
}

// eager: every map point, right after a rebuild (chosen when the map is not much larger than the scan; the values
// are the same either way: a normal depends on the map only)
template <int KN>
__global__ __launch_bounds__(128) void k_normals_all(GridView g, int max_rings, float4* __restrict__ normals,
                                                     int* __restrict__ nflag) {
    const int s = blockIdx.x * blockDim.x + threadIdx.x;
    if (s < g.m) estimate_normal<KN>(g, s, max_rings, normals, nflag);
}

// generic k (rare): top-k list in scratch memory
__global__ __launch_bounds__(128) void k_normals_generic(GridView g, RegState* __restrict__ st,
                                                         const int* __restrict__ worklist, int kn,
                                                         float4* __restrict__ normals, int* __restrict__ nflag) {
    if (st->done) return;
    const int nw = st->n_worklist;
    constexpr int KMAX = 65;
    for (int w = blockIdx.x * blockDim.x + threadIdx.x; w < nw; w += gridDim.x * blockDim.x) {
        const int s = worklist[w];
        const float4 P = g.pts[s];
        float d2[KMAX];
        int idx[KMAX], pos[KMAX];
        for (int k = 0; k < kn; ++k) {
            d2[k] = INFINITY;
            idx[k] = 0x7fffffff;
            pos[k] = -1;
        }
        for (int j = 0; j < g.m; ++j) {  // exhaustive: this path only serves unusual k
            const float4 q = g.pts[j];
            const float dx = q.x - P.x, dy = q.y - P.y, dz = q.z - P.z;
            const float d = fmaf(dz, dz, fmaf(dy, dy, dx * dx));
            const int id = __float_as_int(q.w);
            if (!better(d, id, d2[kn - 1], idx[kn - 1])) continue;
            int k = kn - 1;
            while (k > 0 && better(d, id, d2[k - 1], idx[k - 1])) {
                d2[k] = d2[k - 1];
                idx[k] = idx[k - 1];
                pos[k] = pos[k - 1];
                --k;
            }
            d2[k] = d;
            idx[k] = id;
            pos[k] = j;
        }
        float c00 = 0, c01 = 0, c02 = 0, c11 = 0, c12 = 0, c22 = 0;
        int used = 0;
        for (int k = 1; k < kn; ++k) {
            if (pos[k] < 0) continue;
            const float4 q = g.pts[pos[k]];
            const float dx = q.x - P.x, dy = q.y - P.y, dz = q.z - P.z;
            c00 += dx * dx;
            c01 += dx * dy;
            c02 += dx * dz;
            c11 += dy * dy;
            c12 += dy * dz;
            c22 += dz * dz;
            ++used;
        }
        const float invk = used > 0 ? 1.0f / (float)used : 0.f;
        float nx, ny, nz;
        smallest_eigenvector((double)(c00 * invk), (double)(c01 * invk), (double)(c02 * invk), (double)(c11 * invk),
                             (double)(c12 * invk), (double)(c22 * invk), nx, ny, nz);
        normals[s] = make_float4(nx, ny, nz, 1.f);
        nflag[s] = 1;
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) atomicAdd((unsigned long long*)&st->normals_computed, (unsigned long long)nw);
}

// ---------------------------------------------------------------------------------------------------------------------
// API helper: gather neighbour point / normal / original index per query (LocalMap.NeighborhoodResult)
// ---------------------------------------------------------------------------------------------------------------------
__global__ void k_gather_neighbors(GridView g, const int* __restrict__ nn_pos, const float4* __restrict__ normals,
                                   int n, float* __restrict__ pts_out, float* __restrict__ nrm_out,
                                   int* __restrict__ idx_out) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int s = nn_pos[i];
    float4 q = make_float4(NAN, NAN, NAN, __int_as_float(-1));
    float4 nn = make_float4(NAN, NAN, NAN, 0.f);
    if (s >= 0) {
        q = g.pts[s];
        nn = normals[s];
    }
    if (pts_out) {
        pts_out[3 * i + 0] = q.x;
        pts_out[3 * i + 1] = q.y;
        pts_out[3 * i + 2] = q.z;
    }
    if (nrm_out) {
        nrm_out[3 * i + 0] = nn.x;
        nrm_out[3 * i + 1] = nn.y;
        nrm_out[3 * i + 2] = nn.z;
    }
    if (idx_out) idx_out[i] = __float_as_int(q.w);
}

static GridView make_view(icp_ctx* ctx) {
    GridView g;
    g.table = ctx->table.as<GridEntry>();
    g.mask = ctx->table_size - 1;
    g.h = ctx->cell_h;
    g.inv_h = 1.0f / ctx->cell_h;
    g.pts = ctx->sorted_pts.as<float4>();
    g.m = (int)ctx->map_m;
    return g;
}

int launch_search(icp_ctx* ctx) {
    const int n = (int)ctx->tgt_n;
    if (n <= 0) return ICP_OK;
    const int tok = prof_begin(ctx, 0);
    hipLaunchKernelGGL(k_search, dim3((n + 255) / 256), dim3(256), 0, ctx->stream, make_view(ctx), ctx->tgt_ptr, n,
                       ctx->tgt_mode, 1, reg_state(ctx), ctx->cfg.max_rings, ctx->nn_pos.as<int>(),
                       ctx->nflag.as<int>(), ctx->worklist.as<int>(), ctx->normals_ready ? 0 : 1);
    prof_end(ctx, tok);
    ICP_HIP(ctx, hipGetLastError());
    return ICP_OK;
}

// eager estimation of every map normal (only for the k with a register-resident top-k list)
int launch_normals_all(icp_ctx* ctx) {
    const int kn = ctx->cfg.num_neighbors_normals + 1;
    if (ctx->normals_ready || ctx->map_m <= 0) return ICP_OK;
    if (kn != 11 && kn != 6 && kn != 21) return ICP_OK;  // generic k stays lazy
    const int blocks = (int)((ctx->map_m + 127) / 128);
    GridView g = make_view(ctx);
    const int tok = prof_begin(ctx, 2);
    if (kn == 11)
        hipLaunchKernelGGL(k_normals_all<11>, dim3(blocks), dim3(128), 0, ctx->stream, g, ctx->cfg.max_rings,
                           ctx->normals.as<float4>(), ctx->nflag.as<int>());
    else if (kn == 6)
        hipLaunchKernelGGL(k_normals_all<6>, dim3(blocks), dim3(128), 0, ctx->stream, g, ctx->cfg.max_rings,
                           ctx->normals.as<float4>(), ctx->nflag.as<int>());
    else
        hipLaunchKernelGGL(k_normals_all<21>, dim3(blocks), dim3(128), 0, ctx->stream, g, ctx->cfg.max_rings,
                           ctx->normals.as<float4>(), ctx->nflag.as<int>());
    prof_end(ctx, tok);
    ICP_HIP(ctx, hipGetLastError());
    ctx->normals_ready = true;
    ctx->normals_eager_count += ctx->map_m;
    return ICP_OK;
}

int launch_normals(icp_ctx* ctx) {
    if (ctx->normals_ready) return ICP_OK;
    const int kn = ctx->cfg.num_neighbors_normals + 1;
    // the worklist length lives on the device: launch a fixed grid and stride over it
    int64_t cap = ctx->tgt_n < ctx->map_m ? ctx->tgt_n : ctx->map_m;
    int blocks = (int)((cap + 127) / 128);
    if (blocks < 1) blocks = 1;
    if (blocks > 2048) blocks = 2048;
    const int tok = prof_begin(ctx, 2);
    GridView g = make_view(ctx);
    if (kn == 11) {
        hipLaunchKernelGGL(k_normals<11>, dim3(blocks), dim3(128), 0, ctx->stream, g, reg_state(ctx),
                           ctx->worklist.as<int>(), ctx->cfg.max_rings, ctx->normals.as<float4>(),
                           ctx->nflag.as<int>());
    } else if (kn == 6) {
        hipLaunchKernelGGL(k_normals<6>, dim3(blocks), dim3(128), 0, ctx->stream, g, reg_state(ctx),
                           ctx->worklist.as<int>(), ctx->cfg.max_rings, ctx->normals.as<float4>(),
                           ctx->nflag.as<int>());
    } else if (kn == 21) {
        hipLaunchKernelGGL(k_normals<21>, dim3(blocks), dim3(128), 0, ctx->stream, g, reg_state(ctx),
                           ctx->worklist.as<int>(), ctx->cfg.max_rings, ctx->normals.as<float4>(),
                           ctx->nflag.as<int>());
    } else {
        hipLaunchKernelGGL(k_normals_generic, dim3(blocks), dim3(128), 0, ctx->stream, g, reg_state(ctx),
                           ctx->worklist.as<int>(), kn, ctx->normals.as<float4>(), ctx->nflag.as<int>());
    }
    prof_end(ctx, tok);
    ICP_HIP(ctx, hipGetLastError());
    return ICP_OK;
}

// search without pose transform (LocalMap.nearest_neighbor_search API): state must have done = 0
int launch_search_raw(icp_ctx* ctx) {
    const int n = (int)ctx->tgt_n;
    if (n <= 0) return ICP_OK;
    hipLaunchKernelGGL(k_search, dim3((n + 255) / 256), dim3(256), 0, ctx->stream, make_view(ctx), ctx->tgt_ptr, n,
                       ICP_TARGETS_ALL, 0, reg_state(ctx), ctx->cfg.max_rings, ctx->nn_pos.as<int>(),
                       ctx->nflag.as<int>(), ctx->worklist.as<int>(), ctx->normals_ready ? 0 : 1);
    ICP_HIP(ctx, hipGetLastError());
    return ICP_OK;
}

int launch_gather_neighbors(icp_ctx* ctx, int64_t n, float* pts_out, float* nrm_out, int32_t* idx_out) {
    if (n <= 0) return ICP_OK;
    hipLaunchKernelGGL(k_gather_neighbors, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, ctx->stream,
                       make_view(ctx), ctx->nn_pos.as<int>(), ctx->normals.as<float4>(), (int)n, pts_out, nrm_out,
                       idx_out);
    ICP_HIP(ctx, hipGetLastError());
    return ICP_OK;
}

}  // namespace icp
