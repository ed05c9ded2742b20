// Device-side Gauss-Newton solve and pose update shared by gauss_newton.hip (separate solve kernels) and search.hip
// (the fused iteration kernel solves the previous iteration's equations in its prologue).
#pragma once
#include "icp_internal.h"

namespace icp {

__device__ inline void euler_trig_to_mat_f32(float cx, float sx, float cy, float sy, float cz, float sz, float* R);

__device__ inline void euler_to_mat_f32(float ex, float ey, float ez, float* R) {
    euler_trig_to_mat_f32(cosf(ex), sinf(ex), cosf(ey), sinf(ey), cosf(ez), sinf(ez), R);
}

__device__ inline void build_pose_f32(const float* p, float* T) {  // slam/common/pose.py:120-144
    float R[9];
    euler_to_mat_f32(p[3], p[4], p[5], R);
    T[0] = R[0]; T[1] = R[1]; T[2] = R[2]; T[3] = p[0];
    T[4] = R[3]; T[5] = R[4]; T[6] = R[5]; T[7] = p[1];
    T[8] = R[6]; T[9] = R[7]; T[10] = R[8]; T[11] = p[2];
    T[12] = 0.f; T[13] = 0.f; T[14] = 0.f; T[15] = 1.f;
}

// Cholesky H = L L^T in f64 with fully unrolled static indexing (everything stays in registers): returns det(H)
// (0 if a pivot is not positive: H = J^T J is PSD, so that only happens for a numerically singular system) and solves
// H x = b.
__device__ inline double solve6(double A[6][6], double* b, double* x) {
    double L[6][6], rinv[6];
    double det = 1.0;
    bool ok = true;
#pragma unroll
    for (int j = 0; j < 6; ++j) {
        double d = A[j][j];
#pragma unroll
        for (int k = 0; k < 6; ++k)
            if (k < j) d -= L[j][k] * L[j][k];
        if (!(d > 0.0)) ok = false;
        det *= d;
        const double inv = rsqrt(d);  // 1 / L[j][j] (L[j][j] itself is never needed: one rsqrt instead of a sqrt and a divide)
        rinv[j] = inv;
#pragma unroll
        for (int i = 0; i < 6; ++i) {
            if (i > j) {
                double v = A[i][j];
#pragma unroll
                for (int k = 0; k < 6; ++k)
                    if (k < j) v -= L[i][k] * L[j][k];
                L[i][j] = v * inv;
            }
        }
    }
    double y[6];
#pragma unroll
    for (int i = 0; i < 6; ++i) {
        double v = b[i];
#pragma unroll
        for (int k = 0; k < 6; ++k)
            if (k < i) v -= L[i][k] * y[k];
        y[i] = v * rinv[i];  // 1 / L[i][i], already formed for the column scaling
    }
#pragma unroll
    for (int i = 5; i >= 0; --i) {
        double v = y[i];
#pragma unroll
        for (int k = 0; k < 6; ++k)
            if (k > i) v -= L[k][i] * x[k];
        x[i] = v * rinv[i];
    }
    return ok ? det : 0.0;
}

// Solves one Gauss-Newton step from the packed normal equations.  Returns status; dx (f32) and loss are written.
__device__ inline int gauss_newton_from_neq(const double* neq, float* dx, double* loss, int* stopped) {
    *stopped = 0;
    const double r2 = neq[28];
    if (sqrt(r2) < 1.0e-7) {  // optimization.py:323-327: return x unchanged, loss = res * res (unweighted)
        for (int a = 0; a < 6; ++a) dx[a] = 0.f;
        *loss = r2;
        *stopped = 1;
        return ICP_OK;
    }
    double H[6][6], g[6], x[6];
    int k = 0;
#pragma unroll
    for (int a = 0; a < 6; ++a)
#pragma unroll
        for (int b = a; b < 6; ++b) {
            H[a][b] = neq[k];
            H[b][a] = neq[k];
            ++k;
        }
#pragma unroll
    for (int a = 0; a < 6; ++a) g[a] = neq[21 + a];
    const double det = solve6(H, g, x);
    *loss = neq[27];
    if (!(fabs(det) >= 1.0e-7)) {  // optimization.py:334-336 (also catches NaN)
        for (int a = 0; a < 6; ++a) dx[a] = 0.f;
        return ICP_ERR_INVALID_JACOBIAN;
    }
    for (int a = 0; a < 6; ++a) dx[a] = (float)(-x[a]);  // dx = -H^-1 J^T r (:338)
    return ICP_OK;
}

// SGPR broadcasts of a lane's value (constant lane: v_readlane, not the LDS crossbar of __shfl)
__device__ inline float wave_bcast_f32(float v, int lane_const) {
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), lane_const));
}

// (Round 5 also built the 6x6 solve row-parallel — lane r holding row r of [H | g], Gauss-Jordan with the pivot row
// travelling by v_readlane: 0.80 us against the 0.78 us of the one-lane Cholesky, tools/dev/t/solve_bench.hip: either way
// the step is a chain of six dependent float64 reciprocals / reciprocal square roots.  Not kept.)

// rotation matrix from the sines / cosines of the three Euler angles: Rz(ez) @ Ry(ey) @ Rx(ex)
// (torch_euler_to_mat, slam/common/rotation.py:144-150), float32
__device__ inline void euler_trig_to_mat_f32(float cx, float sx, float cy, float sy, float cz, float sz, float* R) {
    const float a00 = cy, a01 = sy * sx, a02 = sy * cx;
    const float a10 = 0.f, a11 = cx, a12 = -sx;
    const float a20 = -sy, a21 = cy * sx, a22 = cy * cx;
    R[0] = cz * a00 - sz * a10;
    R[1] = cz * a01 - sz * a11;
    R[2] = cz * a02 - sz * a12;
    R[3] = sz * a00 + cz * a10;
    R[4] = sz * a01 + cz * a11;
    R[5] = sz * a02 + cz * a12;
    R[6] = a20;
    R[7] = a21;
    R[8] = a22;
}

// sines and cosines of three angles, one angle per lane (lane % 3), gathered into every lane of the wave: three
// independent libm chains run side by side instead of six calls in a row on one lane
__device__ inline void wave_sincos3(float e0, float e1, float e2, float* sn, float* cs) {
    const int a = (int)(threadIdx.x & 63) % 3;
    const float ang = a == 0 ? e0 : (a == 1 ? e1 : e2);
    float s, c;
    sincosf(ang, &s, &c);  // (one range reduction for both)
#pragma unroll
    for (int k = 0; k < 3; ++k) {  // lanes 0, 1, 2 hold the three angles: SGPR broadcasts, not the LDS crossbar
        sn[k] = wave_bcast_f32(s, k);
        cs[k] = wave_bcast_f32(c, k);
    }
}

__device__ inline void wave_build_pose_f32(const float* p, float* T) {  // slam/common/pose.py:120-144
    float sn[3], cs[3], R[9];
    wave_sincos3(p[3], p[4], p[5], sn, cs);
    euler_trig_to_mat_f32(cs[0], sn[0], cs[1], sn[1], cs[2], sn[2], R);
    T[0] = R[0]; T[1] = R[1]; T[2] = R[2]; T[3] = p[0];
    T[4] = R[3]; T[5] = R[4]; T[6] = R[5]; T[7] = p[1];
    T[8] = R[6]; T[9] = R[7]; T[10] = R[8]; T[11] = p[2];
    T[12] = 0.f; T[13] = 0.f; T[14] = 0.f; T[15] = 1.f;
}

__device__ inline void wave_from_pose_f32(const float* T, float* p) {  // pose.py:188-207, rotation.py:253-270
    const float r00 = T[0], r10 = T[4], r20 = T[8], r21 = T[9], r22 = T[10], r11 = T[5], r12 = T[6];
    const float sy = sqrtf(r00 * r00 + r10 * r10);
    const bool regular = !(sy < 1.0e-6f);
    const int a = (int)(threadIdx.x & 63) % 3;  // one atan2 per lane
    float y, x;
    if (a == 0) {
        y = regular ? r21 : -r12;
        x = regular ? r22 : r11;
    } else if (a == 1) {
        y = -r20;
        x = sy;
    } else {
        y = r10;
        x = r00;
    }
    const float e = atan2f(y, x);
    p[0] = T[3];
    p[1] = T[7];
    p[2] = T[11];
    p[3] = wave_bcast_f32(e, 0);
    p[4] = wave_bcast_f32(e, 1);
    p[5] = regular ? wave_bcast_f32(e, 2) : 0.f;
}

// One Gauss-Newton step + the pose update of register_new_frame, executed by ALL 64 lanes of one wave: the f64 Cholesky
// runs redundantly (identical in every lane), the trigonometry of the pose algebra — most of the serial time — is
// spread one angle per lane.  Pure function of its inputs (every lane returns the same SolveOut): the callers decide who
// writes what where.  `it` / `pose_in` / `params_in`: iteration count, pose and parameters before the step.
struct SolveOut {
    float pose[16];
    float params[6];
    float dx[6];
    double loss;
    int status;     // icp_status
    int done;       // the loop is finished after this step (converged / guard / error / iteration budget)
    int converged;
    int moved;      // pose / params changed (otherwise they are the inputs)
};

__device__ inline void solve_core(const double* __restrict__ neq, AlignParams ap, int it, const float* pose_in,
                                  const float* params_in, SolveOut& o) {
    int stopped;
    o.status = gauss_newton_from_neq(neq, o.dx, &o.loss, &stopped);
    o.done = 0;
    o.converged = 0;
    o.moved = 0;
#pragma unroll
    for (int k = 0; k < 16; ++k) o.pose[k] = pose_in[k];
#pragma unroll
    for (int k = 0; k < 6; ++k) o.params[k] = params_in[k];
    if (o.status != ICP_OK) {
        o.done = 1;
        return;
    }
    // if delta_pose.norm() < threshold: break     (icp_odometry.py:292) — also taken by the residual guard (dx = 0)
    float nrm2 = 0.f;
    for (int a = 0; a < 6; ++a) nrm2 += o.dx[a] * o.dx[a];
    if (sqrtf(nrm2) < ap.threshold_delta_pose || stopped) {
        o.done = 1;
        o.converged = 1;
        return;
    }
    // new_pose_params = from_pose_matrix(delta @ pose); pose = build_pose_matrix(params)   (:296-297), float32
    float D[16], P[16];
    wave_build_pose_f32(o.dx, D);
    {   // P = D @ pose_in, the twelve elements of rows 0-2 one per lane (row 3 of a rigid product is 0 0 0 1) — the same
        // products added in the same order as the scalar triple loop: s = ((0 + d0 p0) + d1 p1) + d2 p2) + d3 p3
        const int l = (int)(threadIdx.x & 63), rr = (l >> 2) % 3, cc = l & 3;
        float s = 0.f;
#pragma unroll
        for (int k2 = 0; k2 < 4; ++k2) {
            const float dv = rr == 0 ? D[k2] : (rr == 1 ? D[4 + k2] : D[8 + k2]);
            const float pv = cc == 0 ? pose_in[4 * k2] : (cc == 1 ? pose_in[4 * k2 + 1] : (cc == 2 ? pose_in[4 * k2 + 2] : pose_in[4 * k2 + 3]));
            s += dv * pv;
        }
#pragma unroll
        for (int e = 0; e < 12; ++e) P[e] = wave_bcast_f32(s, e);
        P[12] = 0.f, P[13] = 0.f, P[14] = 0.f, P[15] = 1.f;
    }
    wave_from_pose_f32(P, o.params);
    wave_build_pose_f32(o.params, o.pose);
    o.moved = 1;
    if (it + 1 >= ap.max_iters) o.done = 1;
}

// the step applied to a RegState in place (lane 0 of the wave writes)
__device__ inline void solve_and_update(RegState* __restrict__ st, const double* __restrict__ neq, AlignParams ap,
                                        double* __restrict__ loss_hist, float* __restrict__ dx_hist, int hist_cap,
                                        int it, const float* pose_in, const float* params_in,
                                        unsigned long long* __restrict__ box = nullptr, unsigned gen = 0,
                                        SolveOut* __restrict__ carry = nullptr) {
    float* __restrict__ pose_hist = ap.pose_hist;
    SolveOut o;
    solve_core(neq, ap, it, pose_in, params_in, o);
    if (carry) *carry = o;  // (every lane: the resident tail's lead solves the next iteration from it, without reading the RegState back)
    if (box) {  // the pose mailbox first (workgroups of this very launch may be polling for it), one granule per lane
        const int lane = threadIdx.x & 63;
        float v = 0.f;
#pragma unroll
        for (int k = 0; k < 12; ++k)
            if (lane == k) v = o.pose[k];
        const unsigned bits = lane < 12 ? __float_as_uint(v) : (lane == 12 ? (unsigned)o.done : (unsigned)(it + 1));
        if (lane < BOX_USED) box_store(box, gen, lane, bits);
    }
    if ((threadIdx.x & 63) != 0) return;
    st->n_worklist = 0;
    st->n_targets = (int)neq[29];
    if (it < hist_cap) {
        loss_hist[it] = o.loss;
        for (int a = 0; a < 6; ++a) dx_hist[6 * it + a] = o.dx[a];
    }
    st->iter = it + 1;
    if (o.status != ICP_OK) st->status = o.status;
    if (o.converged) st->converged = 1;
    if (o.moved) {
        for (int k2 = 0; k2 < 16; ++k2) st->pose_prev[k2] = pose_in[k2];
        for (int a = 0; a < 6; ++a) st->params[a] = o.params[a];
        for (int k2 = 0; k2 < 16; ++k2) st->pose[k2] = o.pose[k2];
        // the pose iteration it + 1 runs with: the NN cache measures how far a target has moved since its search
        if (pose_hist && it + 1 < hist_cap)
            for (int k2 = 0; k2 < 12; ++k2) pose_hist[(size_t)(it + 1) * 12 + k2] = o.pose[k2];
    }
    if (o.done) st->done = 1;
}


// ---------------------------------------------------------------------------------------------------------------------
// Fixed-order sum of the partial rows -> out[NEQ].  The order of the additions depends only on the geometry, never on
// timing, and is defined for 1024 VIRTUAL threads (32 row groups x 32 columns): a workgroup of THREADS threads plays
// 1024 / THREADS of them each, so the 1024-thread summing kernel and the 512-thread lead workgroup of the fused iteration
// launch form the same bits.  Canonical order: with B base rows (128 queries each) and S = ceil(B / 4), super-row s =
// (r[s] + r[s + S]) + (r[s + 2S] + r[s + 3S]) — four base rows a QUARTER OF THE SCAN apart (rows past the end count as
// 0.0; round 3 grouped four consecutive ones: the 512-query workgroup that forms a super-row then searched 512 consecutive
// queries, and the one in the densest region of the map set the duration of the launch); the super-rows are added in the
// strided 8-accumulator pattern below.  `quad` = 1: `partials`
// holds base rows (grouped here); 0: the producer already wrote super-rows (the 512-queries-per-block shape of the fused
// iteration kernel: a quarter of the bytes for this single workgroup to load).  `lds` = 32 x NEQ doubles of scratch.
// ---------------------------------------------------------------------------------------------------------------------
__device__ inline double load_super_row(const double* __restrict__ partials, int nrows, int quad, int sr, int col) {
    if (!quad) return partials[(size_t)sr * NEQ + col];
    const int S = (nrows + 3) / 4;  // super-row sr = base rows sr, sr + S, sr + 2S, sr + 3S
    const double r0 = partials[(size_t)sr * NEQ + col];
    const double r1 = sr + S < nrows ? partials[(size_t)(sr + S) * NEQ + col] : 0.0;
    const double r2 = sr + 2 * S < nrows ? partials[(size_t)(sr + 2 * S) * NEQ + col] : 0.0;
    const double r3 = sr + 3 * S < nrows ? partials[(size_t)(sr + 3 * S) * NEQ + col] : 0.0;
    return (r0 + r1) + (r2 + r3);
}

template <int THREADS, bool RELOAD_TID = false>
__device__ inline void sum_partials_vt(const double* __restrict__ partials, int nrows, int quad, double* out,
                                       double (*lds)[NEQ]) {
    static_assert(1024 % THREADS == 0, "virtual threads");
    const LocalTid threadIdx = RELOAD_TID ? reloaded_tid() : LocalTid{::threadIdx.x};  // (icp_internal.h)
    const int ns = quad ? (nrows + 3) / 4 : nrows;  // super-rows
    if (!quad && ns == 256 && THREADS < 1024) {
        // 256 super-rows (a 131 072-point scan in the 512-query shape): every virtual thread makes exactly one eight-wide
        // trip of the loop below (its accumulators are then the loaded values themselves), so a thread can have the loads
        // of ALL its virtual threads in flight at once instead of one dependent round per virtual thread.  Same additions.
        constexpr int V = 1024 / THREADS;
        double x[V][8];
#pragma unroll
        for (int k = 0; k < V; ++k) {
            const int v = threadIdx.x + k * THREADS, col = v & 31, grp = v >> 5;
#pragma unroll
            for (int i = 0; i < 8; ++i) x[k][i] = partials[(size_t)(grp + 32 * i) * NEQ + col];
        }
#pragma unroll
        for (int k = 0; k < V; ++k) {
            const int v = threadIdx.x + k * THREADS;
            lds[v >> 5][v & 31] = ((x[k][0] + x[k][1]) + (x[k][2] + x[k][3])) + ((x[k][4] + x[k][5]) + (x[k][6] + x[k][7]));
        }
        __syncthreads();
        if (threadIdx.x < NEQ) {
            double t = 0.0;
#pragma unroll
            for (int g = 0; g < 32; ++g) t += lds[g][threadIdx.x];
            out[threadIdx.x] = t;
        }
        return;
    }
#pragma unroll 1
    for (int v = threadIdx.x; v < 1024; v += THREADS) {
        const int col = v & 31, grp = v >> 5;
        double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0, s4 = 0.0, s5 = 0.0, s6 = 0.0, s7 = 0.0;
        int b = grp;
        for (; b + 224 < ns; b += 256) {  // eight independent super-rows (8 or 32 loads) in flight
            const double v0 = load_super_row(partials, nrows, quad, b, col);
            const double v1 = load_super_row(partials, nrows, quad, b + 32, col);
            const double v2 = load_super_row(partials, nrows, quad, b + 64, col);
            const double v3 = load_super_row(partials, nrows, quad, b + 96, col);
            const double v4 = load_super_row(partials, nrows, quad, b + 128, col);
            const double v5 = load_super_row(partials, nrows, quad, b + 160, col);
            const double v6 = load_super_row(partials, nrows, quad, b + 192, col);
            const double v7 = load_super_row(partials, nrows, quad, b + 224, col);
            s0 += v0;
            s1 += v1;
            s2 += v2;
            s3 += v3;
            s4 += v4;
            s5 += v5;
            s6 += v6;
            s7 += v7;
        }
        for (; b < ns; b += 32) s0 += load_super_row(partials, nrows, quad, b, col);
        lds[grp][col] = ((s0 + s1) + (s2 + s3)) + ((s4 + s5) + (s6 + s7));
    }
    __syncthreads();
    if (threadIdx.x < NEQ) {
        double t = 0.0;
#pragma unroll
        for (int g = 0; g < 32; ++g) t += lds[g][threadIdx.x];
        out[threadIdx.x] = t;
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// Partial rows handed over INSIDE a launch (the resident tail of the fused iteration kernel, search.hip): a row is NEQ
// doubles = 2 NEQ granules of 8 bytes, (tag << 32) | 32 payload bits (low half, high half), each written and read by ONE
// relaxed agent-scope access — the data-tagged granule of the pose mailbox: no fence orders anything, a reader takes a
// granule when its tag is the one it waits for (tag = the mailbox generation of the iteration that produced the row).
// ---------------------------------------------------------------------------------------------------------------------
__device__ inline void tagged_row_store(unsigned long long* __restrict__ rows, int row, int col, unsigned tag, double v) {
    const unsigned long long bits = (unsigned long long)__double_as_longlong(v), t = (unsigned long long)tag << 32;
    unsigned long long* p = rows + ((size_t)row * NEQ + col) * 2;
    __hip_atomic_store(p, t | (bits & 0xffffffffull), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_store(p + 1, t | (bits >> 32), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// one element = both granules in ONE 16-byte load (agent-coherent: sc1, as the 8-byte atomic loads of the mailbox): the
// lead pulls 256 rows x 32 elements per iteration, and it is the NUMBER of wave-level load instructions that bounds it
// (1.7 us with two 8-byte loads per element, measured).  A 16-byte aligned load cannot tear inside an aligned 8-byte half.
// Only the FIRST look at an element is such a load: an element that has not arrived yet is polled with the 8-byte atomic
// loads of the mailbox — measured inside the fused kernel, a buffer load repeated in the spin loop kept returning the stale
// element for the whole 50 ms of the wait (with and without the volatile bit; the same loop in a stand-alone kernel,
// tools/dev/t/handoff_bench.hip, sees the new element after 2 us), the atomic loads see it at once.
typedef unsigned int tagged_u32x4 __attribute__((ext_vector_type(4)));
__device__ inline __amdgpu_buffer_rsrc_t tagged_rows_rsrc(const unsigned long long* rows, int nrows) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned long long*>(rows), 0, nrows * NEQ * 16, 0x00020000);
}
__device__ inline tagged_u32x4 tagged_pair_load(__amdgpu_buffer_rsrc_t rsrc, int row, int col) {
    // (bit 31: volatile — the load is repeated in a spin loop and must not be hoisted out of it or merged with its predecessor)
    return __builtin_amdgcn_raw_buffer_load_b128(rsrc, (row * NEQ + col) * 16, 0, (int)0x80000010u /* sc1 | volatile */);
}

// polls one element until both granules carry `tag`; *failed is raised (and 0.0 returned) once `deadline` has passed
__device__ inline double tagged_row_wait(const unsigned long long* __restrict__ rows, int row, int col, unsigned tag,
                                         tagged_u32x4 v, long long deadline, int* __restrict__ failed) {
    const unsigned long long* p = rows + ((size_t)row * NEQ + col) * 2;
    for (;;) {
        if (v.y == tag && v.w == tag) return __longlong_as_double((long long)(((unsigned long long)v.z << 32) | v.x));
        if (wall_clock64() > deadline) {
            *failed = 1;
            return 0.0;
        }
        __builtin_amdgcn_s_sleep(1);
        // (the re-poll: two 8-byte agent-scope atomic loads, as the mailbox polls)
        const unsigned long long lo = __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const unsigned long long hi = __hip_atomic_load(p + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        v.x = (unsigned)lo, v.y = (unsigned)(lo >> 32), v.z = (unsigned)hi, v.w = (unsigned)(hi >> 32);
    }
}

// sum of `ns` tagged SUPER-rows in the canonical order of sum_partials_vt (quad = 0) -> out[NEQ]; the same bits as the
// summing kernels form from plain rows.  *failed (LDS) is raised when a row did not arrive before `deadline`.
template <int THREADS>
__device__ inline void sum_tagged_rows_vt(const unsigned long long* __restrict__ rows, int ns, unsigned tag, double* out,
                                          double (*lds)[NEQ], long long deadline, int* __restrict__ failed) {
    static_assert(1024 % THREADS == 0, "virtual threads");
    constexpr int V = 1024 / THREADS;
    const LocalTid threadIdx = reloaded_tid();  // (icp_internal.h: this runs inside the loop of a persistent kernel)
    const __amdgpu_buffer_rsrc_t rsrc = tagged_rows_rsrc(rows, ns);
    if (ns == 256 && V <= 2) {
        // (a 131 072-point scan: one eight-wide trip per virtual thread — all 8 V elements requested at once)
        tagged_u32x4 q[V][8];
#pragma unroll
        for (int k = 0; k < V; ++k) {
            const int v = threadIdx.x + k * THREADS, col = v & 31, grp = v >> 5;
#pragma unroll
            for (int i = 0; i < 8; ++i) q[k][i] = tagged_pair_load(rsrc, grp + 32 * i, col);
        }
#pragma unroll
        for (int k = 0; k < V; ++k) {
            const int v = threadIdx.x + k * THREADS, col = v & 31, grp = v >> 5;
            double x[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) x[i] = tagged_row_wait(rows, grp + 32 * i, col, tag, q[k][i], deadline, failed);
            lds[grp][col] = ((x[0] + x[1]) + (x[2] + x[3])) + ((x[4] + x[5]) + (x[6] + x[7]));
        }
    } else {
#pragma unroll 1
        for (int v = threadIdx.x; v < 1024; v += THREADS) {
            const int col = v & 31, grp = v >> 5;
            double s[8] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
            int b = grp;
            for (; b + 224 < ns; b += 256) {
                tagged_u32x4 q[8];
#pragma unroll
                for (int i = 0; i < 8; ++i) q[i] = tagged_pair_load(rsrc, b + 32 * i, col);
#pragma unroll
                for (int i = 0; i < 8; ++i) s[i] += tagged_row_wait(rows, b + 32 * i, col, tag, q[i], deadline, failed);
            }
            for (; b < ns; b += 32) s[0] += tagged_row_wait(rows, b, col, tag, tagged_pair_load(rsrc, b, col), deadline, failed);
            lds[grp][col] = ((s[0] + s[1]) + (s[2] + s[3])) + ((s[4] + s[5]) + (s[6] + s[7]));
        }
    }
    __syncthreads();
    if (threadIdx.x < NEQ) {
        double t = 0.0;
#pragma unroll
        for (int g = 0; g < 32; ++g) t += lds[g][threadIdx.x];
        out[threadIdx.x] = t;
    }
}

// the 1024-thread form with its own scratch
__device__ inline void sum_partials_block(const double* __restrict__ partials, int nrows, int quad,
                                          double* out /* LDS or global */) {
    __shared__ double lds[32][NEQ];
    sum_partials_vt<1024>(partials, nrows, quad, out, lds);
}

}  // namespace icp
