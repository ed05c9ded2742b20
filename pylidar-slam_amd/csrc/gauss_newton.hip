// Point-to-plane residual / Jacobian rows, robust weights and the 6x6 normal-equation reduction + solve.
//
// Replaces one `GaussNewtonPointToPlaneAlignment.align` call (slam/odometry/alignment.py:91-127), i.e.
// `PointToPlaneCost.get_residual_fun / get_residual_jac_fun` (slam/common/optimization.py:356-435) evaluated at
// x0 = 0 and `GaussNewton.compute` with max_iters = 1 (:296-344), plus the pose update of
// `ICPFrameToModel.register_new_frame` (slam/odometry/icp_odometry.py:289-297).
//
// The reference materialises J [N,6] and forms J^T J in f32; here every row is built in f32 exactly as there
// (r = (p - q).n, J = [n, p x n], w = sqrt(cost)/clamp(|r|,1e-4)), but the 21 + 6 + 3 sums are accumulated in f64:
// wave64 shuffle reduction -> LDS across the 4 waves of a block -> one partial row per block -> fixed-order final sum
// (bit-reproducible run to run).  No MFMA: this is a gather + reduce, not a dense contraction.
#include "gn_device.h"
#include "icp_internal.h"
#include "solve_device.h"

namespace icp {

static constexpr int RED_THREADS = 256;

struct RowAcc {
    double v[NEQ_USED];
    __device__ inline void zero() {
#pragma unroll
        for (int k = 0; k < NEQ_USED; ++k) v[k] = 0.0;
    }
    // one correspondence: p (transformed target), q (map point), n (map normal)
    __device__ inline float add(float px, float py, float pz, float qx, float qy, float qz, float nx, float ny, float nz,
                                int scheme, float sigma) {
        const float dx = __fsub_rn(px, qx), dy = __fsub_rn(py, qy), dz = __fsub_rn(pz, qz);
        // r = ((p - q) * n).sum(-1)   (optimization.py:427-431)
        const float r = __fadd_rn(__fadd_rn(__fmul_rn(dx, nx), __fmul_rn(dy, ny)), __fmul_rn(dz, nz));
        // J = [n, p x n]              (optimization.py:378-390 at x0 = 0)
        float J[6];
        J[0] = nx;
        J[1] = ny;
        J[2] = nz;
        J[3] = __fsub_rn(__fmul_rn(py, nz), __fmul_rn(pz, ny));
        J[4] = __fsub_rn(__fmul_rn(pz, nx), __fmul_rn(px, nz));
        J[5] = __fsub_rn(__fmul_rn(px, ny), __fmul_rn(py, nx));
        const float d2 = __fadd_rn(__fadd_rn(__fmul_rn(dx, dx), __fmul_rn(dy, dy)), __fmul_rn(dz, dz));
        return add_row(J, r, robust_weight(scheme, sigma, r, d2));
    }
    // one residual r with Jacobian row J and weight w: res *= w, J *= w (optimization.py:329-330), f64 sums; returns
    // (w r)^2, the row's entry of the residual vector `GaussNewton.compute` hands back (:342-344)
    __device__ inline float add_row(const float* J, float r, float w) {
        const float rw = __fmul_rn(r, w);
        double Jw[6];
#pragma unroll
        for (int a = 0; a < 6; ++a) Jw[a] = (double)__fmul_rn(J[a], w);
        int k = 0;
#pragma unroll
        for (int a = 0; a < 6; ++a)
#pragma unroll
            for (int b = a; b < 6; ++b) v[k++] += Jw[a] * Jw[b];  // H = J^T J (:332-333), upper triangle
#pragma unroll
        for (int a = 0; a < 6; ++a) v[21 + a] += Jw[a] * (double)rw;  // J^T r
        const float rw2 = __fmul_rn(rw, rw);
        v[27] += (double)rw2;                                          // loss = sum (w r)^2
        v[28] += (double)__fmul_rn(r, r);                              // ||r||^2 for the 1e-7 guard (:323)
        v[29] += 1.0;
        return rw2;
    }
};

__device__ inline double wave_sum(double x) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) x += __shfl_down(x, o, 64);
    return x;
}

__device__ inline void block_store_partial(RowAcc& acc, double* __restrict__ partial_row) {
    __shared__ double lds[RED_THREADS / 64][NEQ];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int k = 0; k < NEQ_USED; ++k) {
        const double s = wave_sum(acc.v[k]);
        if (lane == 0) lds[wave][k] = s;
    }
    __syncthreads();
    if (threadIdx.x < NEQ) {
        double s = 0.0;
        if (threadIdx.x < NEQ_USED) {
#pragma unroll
            for (int w = 0; w < RED_THREADS / 64; ++w) s += lds[w][threadIdx.x];
        }
        partial_row[threadIdx.x] = s;
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// K3: rows from the search result
// ---------------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(RED_THREADS) void k_reduce(const float4* __restrict__ map_pts,
                                                        const float4* __restrict__ normals,
                                                        const float4* __restrict__ tgt, const int* __restrict__ nn_pos,
                                                        int n, const RegState* __restrict__ st, AlignParams ap,
                                                        double* __restrict__ partials) {
    if (st->done) return;
    RowAcc acc;
    acc.zero();
    const float* T = st->pose;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        const int s = nn_pos[i];
        if (s < 0) continue;
        const float4 t4 = tgt[i];
        const float x = t4.x, y = t4.y, z = t4.z;
        const float px = fmaf(z, T[2], fmaf(y, T[1], x * T[0])) + T[3];
        const float py = fmaf(z, T[6], fmaf(y, T[5], x * T[4])) + T[7];
        const float pz = fmaf(z, T[10], fmaf(y, T[9], x * T[8])) + T[11];
        const float4 q = map_pts[s];
        const float4 nn = normals[s];
        acc.add(px, py, pz, q.x, q.y, q.z, nn.x, nn.y, nn.z, ap.scheme, ap.sigma);
    }
    block_store_partial(acc, partials + (size_t)blockIdx.x * NEQ);
}

// rows from caller-supplied correspondences (RigidAlignment.align seam)
__global__ __launch_bounds__(RED_THREADS) void k_reduce_given(const float* __restrict__ ref,
                                                              const float* __restrict__ tgt,
                                                              const float* __restrict__ nrm, int n, AlignParams ap,
                                                              double* __restrict__ partials,
                                                              float* __restrict__ residuals) {
    RowAcc acc;
    acc.zero();
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        const float rw2 = acc.add(tgt[3 * i], tgt[3 * i + 1], tgt[3 * i + 2], ref[3 * i], ref[3 * i + 1],
                                  ref[3 * i + 2], nrm[3 * i], nrm[3 * i + 1], nrm[3 * i + 2], ap.scheme, ap.sigma);
        if (residuals) residuals[i] = rw2;
    }
    block_store_partial(acc, partials + (size_t)blockIdx.x * NEQ);
}

// Point-to-point rows (`PointToPointCost`, slam/common/optimization.py:458-560) at the linearisation point x0:
//   d = (R0 p + t0) - q,  r = ||d||                         residual  (:540-550)
//   J_k = (dT/dx_k p) . d  with dT/dx_k = e_k for the translations and dR/de_k p for the Euler angles (:483-499) —
//   the reference's own (unnormalised) form, reproduced as is
//   w from the scheme on |r|; `neighborhood` sees the RAW target point (alignment.py:183: target_points=tgt_points)
struct P2PLinearisation {
    float T0[12];   // R0 (row-major 3x3) then t0
    float dR[27];   // d R / d ex, ey, ez at x0 (rotation.py:166-187)
};

// one point-to-point row: p = target (raw), q = reference
__device__ inline float p2p_row(RowAcc& acc, float px, float py, float pz, float qx, float qy, float qz,
                                const P2PLinearisation& L, AlignParams ap) {
    float d[3], J[6];
#pragma unroll
    for (int a = 0; a < 3; ++a) {  // apply_transformation: einsum over j, then + t (pose.py:169-186)
        const float rp = __fadd_rn(__fadd_rn(__fmul_rn(L.T0[3 * a], px), __fmul_rn(L.T0[3 * a + 1], py)),
                                   __fmul_rn(L.T0[3 * a + 2], pz));
        d[a] = __fsub_rn(__fadd_rn(rp, L.T0[9 + a]), a == 0 ? qx : (a == 1 ? qy : qz));
    }
    const float r = sqrtf(__fadd_rn(__fadd_rn(__fmul_rn(d[0], d[0]), __fmul_rn(d[1], d[1])), __fmul_rn(d[2], d[2])));
    J[0] = d[0];
    J[1] = d[1];
    J[2] = d[2];
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        float s = 0.f;
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            const float* M = L.dR + 9 * k + 3 * a;
            const float v = __fadd_rn(__fadd_rn(__fmul_rn(M[0], px), __fmul_rn(M[1], py)), __fmul_rn(M[2], pz));
            s = __fadd_rn(s, __fmul_rn(v, d[a]));
        }
        J[3 + k] = s;
    }
    const float ex = __fsub_rn(px, qx), ey = __fsub_rn(py, qy), ez = __fsub_rn(pz, qz);
    const float d2raw = __fadd_rn(__fadd_rn(__fmul_rn(ex, ex), __fmul_rn(ey, ey)), __fmul_rn(ez, ez));
    return acc.add_row(J, r, robust_weight(ap.scheme, ap.sigma, r, d2raw));
}

__global__ __launch_bounds__(RED_THREADS) void k_reduce_p2p(const float* __restrict__ ref, const float* __restrict__ tgt,
                                                            int n, P2PLinearisation L, AlignParams ap,
                                                            double* __restrict__ partials,
                                                            float* __restrict__ residuals) {
    RowAcc acc;
    acc.zero();
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        const float rw2 = p2p_row(acc, tgt[3 * i], tgt[3 * i + 1], tgt[3 * i + 2], ref[3 * i], ref[3 * i + 1],
                                  ref[3 * i + 2], L, ap);
        if (residuals) residuals[i] = rw2;
    }
    block_store_partial(acc, partials + (size_t)blockIdx.x * NEQ);
}

// point-to-point rows of a registration iteration: targets transformed by the current pose (icp_odometry.py:275) against
// their nearest neighbours, linearised at x0 = 0 (what `GaussNewtonPointToPointAlignment.align` does without an initial
// estimate, alignment.py:173-176)
__global__ __launch_bounds__(RED_THREADS) void k_reduce_p2p_nn(const float4* __restrict__ map_pts,
                                                               const float4* __restrict__ tgt,
                                                               const int* __restrict__ nn_pos, int n,
                                                               const RegState* __restrict__ st, P2PLinearisation L,
                                                               AlignParams ap, double* __restrict__ partials) {
    if (st->done) return;
    RowAcc acc;
    acc.zero();
    const float* T = st->pose;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        const int s = nn_pos[i];
        if (s < 0) continue;
        const float4 t4 = tgt[i];
        const float x = t4.x, y = t4.y, z = t4.z;
        const float px = fmaf(z, T[2], fmaf(y, T[1], x * T[0])) + T[3];
        const float py = fmaf(z, T[6], fmaf(y, T[5], x * T[4])) + T[7];
        const float pz = fmaf(z, T[10], fmaf(y, T[9], x * T[8])) + T[11];
        const float4 q = map_pts[s];
        p2p_row(acc, px, py, pz, q.x, q.y, q.z, L, ap);
    }
    block_store_partial(acc, partials + (size_t)blockIdx.x * NEQ);
}

// weighted means and cross-covariance for `weighted_procrustes` (slam/common/registration.py:15-74): two passes.
// pass 1: sum w, sum w p_tgt, sum w p_ref  -> partial rows (7 used columns)
__global__ __launch_bounds__(RED_THREADS) void k_procrustes_means(const float* __restrict__ tgt,
                                                                  const float* __restrict__ ref,
                                                                  const float* __restrict__ w, int n,
                                                                  double* __restrict__ partials) {
    RowAcc acc;
    acc.zero();
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        const double wi = w ? (double)w[i] : 1.0;
        acc.v[0] += wi;
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            acc.v[1 + a] += wi * (double)tgt[3 * i + a];
            acc.v[4 + a] += wi * (double)ref[3 * i + a];
        }
    }
    block_store_partial(acc, partials + (size_t)blockIdx.x * NEQ);
}

// pass 2: C[i][j] = sum (ref - mu_ref)_i (tgt - mu_tgt)_j, the centred differences formed in float32 (:41-44)
struct ProcrustesMeans {
    float mu_tgt[3], mu_ref[3];
};

__global__ __launch_bounds__(RED_THREADS) void k_procrustes_cov(const float* __restrict__ tgt,
                                                                const float* __restrict__ ref, int n,
                                                                ProcrustesMeans mu, double* __restrict__ partials) {
    RowAcc acc;
    acc.zero();
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        float dt[3], dr[3];
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            dt[a] = __fsub_rn(tgt[3 * i + a], mu.mu_tgt[a]);
            dr[a] = __fsub_rn(ref[3 * i + a], mu.mu_ref[a]);
        }
#pragma unroll
        for (int a = 0; a < 3; ++a)
#pragma unroll
            for (int b = 0; b < 3; ++b) acc.v[3 * a + b] += (double)dr[a] * (double)dt[b];
    }
    block_store_partial(acc, partials + (size_t)blockIdx.x * NEQ);
}

// (the fixed-order sum of the partial rows — `sum_partials_block` — lives in solve_device.h: the lead workgroup of the
// fused iteration launch runs it too)
__global__ __launch_bounds__(1024) void k_sum_partials(const double* __restrict__ partials, int nblocks, int quad,
                                                       const RegState* __restrict__ st, int check_done,
                                                       double* __restrict__ neq) {
    if (check_done && st->done) return;
    sum_partials_block(partials, nblocks, quad, neq);
}

// ---------------------------------------------------------------------------------------------------------------------
// K4: solve + pose update (one thread)
// ---------------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(64) void k_solve(RegState* __restrict__ st, const double* __restrict__ neq, AlignParams ap,
                                              double* __restrict__ loss_hist, float* __restrict__ dx_hist,
                                              int hist_cap) {
    if (blockIdx.x != 0) return;
    if (st->done) return;  // wave-uniform
    float pose_in[16], params_in[6];
    for (int k = 0; k < 16; ++k) pose_in[k] = st->pose[k];
    for (int k = 0; k < 6; ++k) params_in[k] = st->params[k];
    solve_and_update(st, neq, ap, loss_hist, dx_hist, hist_cap, st->iter, pose_in, params_in);
}

// single-GPU path: final sum of the partial rows + solve + pose update in one launch
__device__ __forceinline__ void sum_solve_body(const double* __restrict__ partials, int nblocks, int quad,
                                               RegState* __restrict__ st, AlignParams ap,
                                               double* __restrict__ neq, double* __restrict__ loss_hist,
                                               float* __restrict__ dx_hist, int hist_cap,
                                               unsigned long long* __restrict__ box, unsigned gen,
                                               const unsigned* __restrict__ result_src, unsigned* __restrict__ result_dst,
                                               int result_words) {
    // result_dst (the last solving launch of an enqueued registration): the state allocation — RegState and histories, the
    // block icp_register_end reads — goes to the pinned result slot (mapped into the device) by this launch instead of a
    // copy launch of its own behind it
    // the state words are requested first and consumed last: their latency hides behind the partial-row loads
    // (a hand-off that timed out in an earlier launch of this registration — handoff_timeouts — left incomplete rows behind:
    // nothing is solved from them; icp_register_end re-runs the rest of the loop from the iteration the state holds)
    const int done = st->done | (st->handoff_timeouts > 0 ? 1 : 0);
    int it = 0;
    float pose_in[16], params_in[6];
    if (threadIdx.x < 64) {  // the solving wave (uniform addresses: one transaction)
        it = st->iter;
#pragma unroll
        for (int k = 0; k < 16; ++k) pose_in[k] = st->pose[k];
#pragma unroll
        for (int k = 0; k < 6; ++k) params_in[k] = st->params[k];
    }
    __shared__ double total[NEQ];
    sum_partials_block(partials, nblocks, quad, total);
    __syncthreads();
    if (done) {
        // (box: the next fused launch takes its pose — and the news that the loop is over — from the mailbox)
        if (box && threadIdx.x < BOX_USED) {
            const int k = threadIdx.x;
            box_store(box, gen, k, k < 12 ? __float_as_uint(st->pose[k]) : (k == 12 ? 1u : (unsigned)st->iter));
        }
    } else {
        if (threadIdx.x < NEQ) neq[threadIdx.x] = total[threadIdx.x];
        if (threadIdx.x < 64) solve_and_update(st, total, ap, loss_hist, dx_hist, hist_cap, it, pose_in, params_in, box, gen);
    }
    if (result_dst) {  // (block-uniform)
        __syncthreads();  // the solving wave's stores to the state and the histories: visible to the workgroup
        for (int w = threadIdx.x; w < result_words; w += blockDim.x)
            __hip_atomic_store(result_dst + w, __hip_atomic_load(result_src + w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP),
                               __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}

__global__ __launch_bounds__(1024) void k_sum_solve(const double* __restrict__ partials, int nblocks, int quad,
                                                    RegState* __restrict__ st, AlignParams ap,
                                                    double* __restrict__ neq, double* __restrict__ loss_hist,
                                                    float* __restrict__ dx_hist, int hist_cap,
                                                    unsigned long long* __restrict__ box, unsigned gen,
                                                    const unsigned* __restrict__ result_src, unsigned* __restrict__ result_dst,
                                                    int result_words) {
    sum_solve_body(partials, nblocks, quad, st, ap, neq, loss_hist, dx_hist, hist_cap, box, gen, result_src, result_dst,
                   result_words);
}

// ... of B sequences in one launch (icp_batch_*, api.hip): workgroup b sums and solves sequence b from its descriptor
struct SumSolveDesc {
    const double* partials;
    RegState* st;
    AlignParams ap;
    double* neq;
    double* loss_hist;
    float* dx_hist;
    unsigned long long* box;
    const unsigned* result_src;
    unsigned* result_dst;
    int nblocks, quad, hist_cap, result_words;
    unsigned gen;
    int pad[3];
};

__global__ __launch_bounds__(1024) void k_sum_solve_batch(const SumSolveDesc* __restrict__ table) {
    const SumSolveDesc d = table[blockIdx.x];
    sum_solve_body(d.partials, d.nblocks, d.quad, d.st, d.ap, d.neq, d.loss_hist, d.dx_hist, d.hist_cap, d.box, d.gen,
                   d.result_src, d.result_dst, d.result_words);
}

// ---------------------------------------------------------------------------------------------------------------------
// Multi-GPU: final sum + ALL-REDUCE over the ranks + solve in one launch (icp_exchange_*).
//
// The message is 256 bytes: an RCCL ring all-reduce costs tens of microseconds of latency per ICP iteration and — driven
// from the host — three enqueues and a stream hand-off.  Here the block that has just summed this rank's partial rows
// writes its 32 doubles into slot [parity][rank] of every rank's inbox (its own included; peers' inboxes are mapped
// through IPC, the stores travel over xGMI as system-scope write-through stores), drains them, publishes the tag,
// waits for the `world` tags of this exchange in its OWN inbox (relaxed system-scope polls with s_sleep, bounded by a
// wall-clock budget) and adds the slots in RANK ORDER, so every rank forms the identical sum and applies the identical
// solve: poses stay bit-identical across ranks without a broadcast.
// Slot reuse: parity = exchange number & 1.  A rank can write exchange s + 2 only after it completed s + 1, which needs
// this rank's s + 1 contribution, which is sent after this rank has read every slot of s: two parities suffice.
// On a timeout (a peer died or never launched) the registration stops with ICP_ERR_EXCHANGE; later launches are no-ops.
// ---------------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(1024) void k_sum_exchange_solve(const double* __restrict__ partials, int nblocks, int quad,
                                                             RegState* __restrict__ st, AlignParams ap,
                                                             double* __restrict__ neq, double* __restrict__ loss_hist,
                                                             float* __restrict__ dx_hist, int hist_cap,
                                                             ExchangeView x) {
    const int done = st->done;
    int it = 0;
    float pose_in[16], params_in[6];
    if (threadIdx.x < 64) {
        it = st->iter;
#pragma unroll
        for (int k = 0; k < 16; ++k) pose_in[k] = st->pose[k];
#pragma unroll
        for (int k = 0; k < 6; ++k) params_in[k] = st->params[k];
    }
    __shared__ double total[NEQ];
    __shared__ int arrived;
    sum_partials_block(partials, nblocks, quad, total);
    if (threadIdx.x == 0) arrived = 1;
    __syncthreads();
    if (done) return;  // identical on every rank (same state after the same solves): nobody exchanges
    const unsigned long long seq = *x.seq + 1;  // number of THIS exchange (1, 2, ..): identical on every rank
    const int par = (int)(seq & 1ull);
    // ---- publish: payload into every inbox, drained, then the tags
    if ((int)threadIdx.x < NEQ * x.world) {
        const int p = threadIdx.x / NEQ, e = threadIdx.x % NEQ;
        __hip_atomic_store(&x.inbox[p][par * x.world + x.rank].v[e], total[e], __ATOMIC_RELAXED,
                           __HIP_MEMORY_SCOPE_SYSTEM);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if ((int)threadIdx.x < x.world) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "");  // system scope
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __hip_atomic_store(&x.inbox[threadIdx.x][par * x.world + x.rank].tag, seq, __ATOMIC_RELAXED,
                           __HIP_MEMORY_SCOPE_SYSTEM);
        // ---- wait for rank `threadIdx.x`'s contribution in the own inbox
        const ExchangeSlot* mine = &x.inbox[x.rank][par * x.world + threadIdx.x];
        const long long t0 = wall_clock64();
        while (__hip_atomic_load(&mine->tag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) != seq) {
            if (wall_clock64() - t0 > x.timeout_ticks) {
                arrived = 0;
                break;
            }
            __builtin_amdgcn_s_sleep(8);
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "");
    }
    __syncthreads();
    if (!arrived) {  // block-uniform
        if (threadIdx.x == 0) {
            st->status = ICP_ERR_EXCHANGE;
            st->done = 1;
            st->iter = it + 1;
        }
        return;
    }
    // ---- the same fixed-order sum on every rank
    if (threadIdx.x < NEQ) {
        double s = 0.0;
        for (int r = 0; r < x.world; ++r)
            s += __hip_atomic_load(&x.inbox[x.rank][par * x.world + r].v[threadIdx.x], __ATOMIC_RELAXED,
                                   __HIP_MEMORY_SCOPE_SYSTEM);
        total[threadIdx.x] = s;
        neq[threadIdx.x] = s;
    }
    if (threadIdx.x == 0) *x.seq = seq;
    __syncthreads();
    if (threadIdx.x < 64) solve_and_update(st, total, ap, loss_hist, dx_hist, hist_cap, it, pose_in, params_in);
}

// align() on given correspondences: writes params = x0 + dx [6], pose[16] = build_pose_matrix(params) (f32) and loss
// into `out` (device, 6 + 16 floats; loss double)
struct Params6 {
    float v[6];
};

__global__ void k_solve_given(const double* __restrict__ neq, Params6 x0, float* __restrict__ out_f,
                              double* __restrict__ out_loss, int* __restrict__ out_status) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    float dx[6];
    double loss;
    int stopped;
    *out_status = gauss_newton_from_neq(neq, dx, &loss, &stopped);
    for (int a = 0; a < 6; ++a) out_f[a] = __fadd_rn(x0.v[a], dx[a]);  // x = x0 + dx (optimization.py:338-340)
    build_pose_f32(out_f, out_f + 6);
    *out_loss = loss;
}

AlignParams make_align_params(const icp_ctx* ctx) {
    AlignParams ap;
    ap.scheme = ctx->cfg.scheme;
    ap.sigma = ctx->cfg.sigma;
    ap.threshold_delta_pose = ctx->cfg.threshold_delta_pose;
    ap.max_iters = ctx->cfg.max_num_alignments;
    ap.pose_hist = ctx->pose_hist;
    return ap;
}

static int reduce_grid(int64_t n) {
    int blocks = (int)((n + RED_THREADS - 1) / RED_THREADS);
    if (blocks < 1) blocks = 1;
    if (blocks > 256) blocks = 256;  // one block per CU: 256 partial rows to sum
    return blocks;
}

int launch_reduce(icp_ctx* ctx) {
    const int n = (int)ctx->tgt_n;
    const int blocks = reduce_grid(n);
    ICP_HIP(ctx, ctx->partials.reserve((size_t)blocks * NEQ * sizeof(double)));
    const int tok = prof_begin(ctx, 1);
    hipLaunchKernelGGL(k_reduce, dim3(blocks), dim3(RED_THREADS), 0, ctx->stream, ctx->sorted_pts.as<float4>(),
                       ctx->normals.as<float4>(), ctx->tgt4.as<float4>(), ctx->nn_pos.as<int>(), n, reg_state(ctx),
                       make_align_params(ctx), ctx->partials.as<double>());
    hipLaunchKernelGGL(k_sum_partials, dim3(1), dim3(1024), 0, ctx->stream, ctx->partials.as<double>(), blocks, 1,
                       reg_state(ctx), 1, ctx->neq);
    prof_end(ctx, tok);
    ICP_HIP(ctx, hipGetLastError());
    return ICP_OK;
}

// final sum + solve over partial rows produced by the fused iteration kernel (search.hip::launch_iterate_fused)
int launch_sum_solve(icp_ctx* ctx, int blocks, int quad, const double* partials, bool publish, bool last) {
    if (!partials) partials = ctx->partials.as<double>();
    // the last solving launch of an enqueued registration delivers the result block itself (api.hip::enqueue_result_copy)
    unsigned* fold = (last && !ctx->exchange_on) ? reinterpret_cast<unsigned*>(ctx->result_fold_to) : nullptr;
    if (fold) ctx->result_folded = true;
    if (ctx->exchange_on) {
        ExchangeView x = ctx->xview;
        x.timeout_ticks = (long long)(ctx->exchange_timeout_ms * 1.0e5);  // 100 MHz
        hipLaunchKernelGGL(k_sum_exchange_solve, dim3(1), dim3(1024), 0, ctx->stream, partials,
                           blocks, quad, reg_state(ctx), make_align_params(ctx), ctx->neq, ctx->loss_hist,
                           ctx->dx_hist, ctx->hist_cap, x);
    } else {
        hipLaunchKernelGGL(k_sum_solve, dim3(1), dim3(1024), 0, ctx->stream, partials, blocks, quad,
                           reg_state(ctx), make_align_params(ctx), ctx->neq, ctx->loss_hist, ctx->dx_hist,
                           ctx->hist_cap, publish ? pose_box(ctx) : (unsigned long long*)nullptr,
                           publish ? next_box_generation(ctx) : 0u, ctx->state.as<unsigned>(), fold,
                           fold ? (int)(ctx->result_fold_bytes / 4) : 0);
    }
    ICP_HIP(ctx, hipGetLastError());
    return ICP_OK;
}

// the batched form of launch_sum_solve (no exchange): member b's descriptor into table_host[b]
int prepare_sum_solve_batch(icp_ctx* const* ctxs, int count, const int* rows, const int* quad, bool publish, bool last,
                            void* table_host) {
    SumSolveDesc* table = reinterpret_cast<SumSolveDesc*>(table_host);
    for (int b = 0; b < count; ++b) {
        icp_ctx* ctx = ctxs[b];
        SumSolveDesc d{};
        // the rows of the launch just prepared: the parity it has written (lead launches alternate; classic ones write parity 0)
        d.partials = publish ? (const double*)(ctx->partials.as<char>() + (size_t)(ctx->partials_parity ^ 1) * ctx->partials_half)
                             : ctx->partials.as<double>();
        d.st = reg_state(ctx);
        d.ap = make_align_params(ctx);
        d.neq = ctx->neq;
        d.loss_hist = ctx->loss_hist;
        d.dx_hist = ctx->dx_hist;
        d.hist_cap = ctx->hist_cap;
        d.nblocks = rows[b];
        d.quad = quad[b];
        d.box = publish ? pose_box(ctx) : nullptr;
        d.gen = publish ? next_box_generation(ctx) : 0u;
        d.result_src = ctx->state.as<unsigned>();
        d.result_dst = last ? reinterpret_cast<unsigned*>(ctx->result_fold_to) : nullptr;
        d.result_words = d.result_dst ? (int)(ctx->result_fold_bytes / 4) : 0;
        if (d.result_dst) ctx->result_folded = true;
        table[b] = d;
    }
    return ICP_OK;
}

int launch_sum_solve_batch(icp_ctx* first, int count, const void* table_dev) {
    hipLaunchKernelGGL(k_sum_solve_batch, dim3(count), dim3(1024), 0, first->stream,
                       reinterpret_cast<const SumSolveDesc*>(table_dev));
    ICP_HIP(first, hipGetLastError());
    return ICP_OK;
}

size_t sum_solve_desc_bytes() { return sizeof(SumSolveDesc); }

int launch_sum_partials(icp_ctx* ctx, int blocks, int quad) {
    hipLaunchKernelGGL(k_sum_partials, dim3(1), dim3(1024), 0, ctx->stream, ctx->partials.as<double>(), blocks, quad,
                       reg_state(ctx), 1, ctx->neq);
    ICP_HIP(ctx, hipGetLastError());
    return ICP_OK;
}

int launch_reduce_solve(icp_ctx* ctx) {
    const int n = (int)ctx->tgt_n;
    const int blocks = reduce_grid(n);
    ICP_HIP(ctx, ctx->partials.reserve((size_t)blocks * NEQ * sizeof(double)));
    const int tok = prof_begin(ctx, 1);
    hipLaunchKernelGGL(k_reduce, dim3(blocks), dim3(RED_THREADS), 0, ctx->stream, ctx->sorted_pts.as<float4>(),
                       ctx->normals.as<float4>(), ctx->tgt4.as<float4>(), ctx->nn_pos.as<int>(), n, reg_state(ctx),
                       make_align_params(ctx), ctx->partials.as<double>());
    const int rc = launch_sum_solve(ctx, blocks);
    prof_end(ctx, tok);
    if (rc) return rc;
    ICP_HIP(ctx, hipGetLastError());
    return ICP_OK;
}

int launch_solve(icp_ctx* ctx) {
    hipLaunchKernelGGL(k_solve, dim3(1), dim3(64), 0, ctx->stream, reg_state(ctx), ctx->neq, make_align_params(ctx),
                       ctx->loss_hist, ctx->dx_hist, ctx->hist_cap);
    ICP_HIP(ctx, hipGetLastError());
    return ICP_OK;
}

// out layout in ctx->stage_out: [0..21] floats (params, pose), then at byte 128 the loss (double), at byte 136 status (int)
static int launch_solve_given(icp_ctx* ctx, int blocks, const float* x0) {
    hipLaunchKernelGGL(k_sum_partials, dim3(1), dim3(1024), 0, ctx->stream, ctx->partials.as<double>(), blocks, 1,
                       reg_state(ctx), 0, ctx->neq);
    Params6 p;
    for (int a = 0; a < 6; ++a) p.v[a] = x0 ? x0[a] : 0.f;
    char* out = ctx->stage_out.as<char>();
    hipLaunchKernelGGL(k_solve_given, dim3(1), dim3(64), 0, ctx->stream, ctx->neq, p, (float*)out,
                       (double*)(out + 128), (int*)(out + 136));
    ICP_HIP(ctx, hipGetLastError());
    return ICP_OK;
}

int launch_align_given(icp_ctx* ctx, const float* ref, const float* tgt, const float* nrm, int64_t n,
                       float* residuals_dev) {
    const int blocks = reduce_grid(n);
    ICP_HIP(ctx, ctx->partials.reserve((size_t)blocks * NEQ * sizeof(double)));
    ICP_HIP(ctx, ctx->stage_out.reserve(256));
    hipLaunchKernelGGL(k_reduce_given, dim3(blocks), dim3(RED_THREADS), 0, ctx->stream, ref, tgt, nrm, (int)n,
                       make_align_params(ctx), ctx->partials.as<double>(), residuals_dev);
    return launch_solve_given(ctx, blocks, nullptr);
}

// R = Rz(ez) Ry(ey) Rx(ex) and its derivatives in float32, the products in the reference's order
// (torch_euler_to_mat / torch_euler_jacobian, slam/common/rotation.py:144-187)
static void mat3_mul(const float* A, const float* B, float* C) {
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) {
            float s = 0.f;
            for (int k = 0; k < 3; ++k) s += A[3 * i + k] * B[3 * k + j];
            C[3 * i + j] = s;
        }
}

static P2PLinearisation linearise_euler(const float* x0) {
    P2PLinearisation L;
    const float z[6] = {0, 0, 0, 0, 0, 0};
    if (!x0) x0 = z;
    const float cx = cosf(x0[3]), sx = sinf(x0[3]), cy = cosf(x0[4]), sy = sinf(x0[4]), cz = cosf(x0[5]), sz = sinf(x0[5]);
    const float Rx[9] = {1, 0, 0, 0, cx, -sx, 0, sx, cx}, Ry[9] = {cy, 0, sy, 0, 1, 0, -sy, 0, cy},
                Rz[9] = {cz, -sz, 0, sz, cz, 0, 0, 0, 1};
    const float Jx[9] = {0, 0, 0, 0, -sx, -cx, 0, cx, -sx}, Jy[9] = {-sy, 0, cy, 0, 0, 0, -cy, 0, -sy},
                Jz[9] = {-sz, -cz, 0, cz, -sz, 0, 0, 0, 0};
    float zy[9], t[9];
    mat3_mul(Rz, Ry, zy);
    mat3_mul(zy, Rx, L.T0);            // (Rz @ Ry) @ Rx
    mat3_mul(zy, Jx, L.dR);            // (Rz @ Ry) @ dRx
    mat3_mul(Rz, Jy, t);
    mat3_mul(t, Rx, L.dR + 9);         // (Rz @ dRy) @ Rx
    mat3_mul(Jz, Ry, t);
    mat3_mul(t, Rx, L.dR + 18);        // (dRz @ Ry) @ Rx
    for (int a = 0; a < 3; ++a) L.T0[9 + a] = x0[a];
    return L;
}

int launch_align_p2p(icp_ctx* ctx, const float* ref, const float* tgt, int64_t n, const float* x0,
                     float* residuals_dev) {
    const int blocks = reduce_grid(n);
    ICP_HIP(ctx, ctx->partials.reserve((size_t)blocks * NEQ * sizeof(double)));
    ICP_HIP(ctx, ctx->stage_out.reserve(256));
    hipLaunchKernelGGL(k_reduce_p2p, dim3(blocks), dim3(RED_THREADS), 0, ctx->stream, ref, tgt, (int)n,
                       linearise_euler(x0), make_align_params(ctx), ctx->partials.as<double>(), residuals_dev);
    return launch_solve_given(ctx, blocks, x0);
}

// point-to-point cost in the registration loop: rows from the search result -> partial rows (`solve`: + final sum and
// solve, the single-GPU path; otherwise + final sum only, the exchange seam)
int launch_reduce_p2p(icp_ctx* ctx, bool solve) {
    const int n = (int)ctx->tgt_n;
    const int blocks = reduce_grid(n);
    ICP_HIP(ctx, ctx->partials.reserve((size_t)blocks * NEQ * sizeof(double)));
    const int tok = prof_begin(ctx, 1);
    hipLaunchKernelGGL(k_reduce_p2p_nn, dim3(blocks), dim3(RED_THREADS), 0, ctx->stream, ctx->sorted_pts.as<float4>(),
                       ctx->tgt4.as<float4>(), ctx->nn_pos.as<int>(), n, reg_state(ctx), linearise_euler(nullptr),
                       make_align_params(ctx), ctx->partials.as<double>());
    if (solve) {
        const int rc = launch_sum_solve(ctx, blocks);
        if (rc) return rc;
    } else
        hipLaunchKernelGGL(k_sum_partials, dim3(1), dim3(1024), 0, ctx->stream, ctx->partials.as<double>(), blocks, 1,
                           reg_state(ctx), 1, ctx->neq);
    prof_end(ctx, tok);
    ICP_HIP(ctx, hipGetLastError());
    return ICP_OK;
}

// sums for weighted_procrustes; host_out[NEQ] receives the fixed-order total of the pass (synchronises)
int launch_procrustes_pass(icp_ctx* ctx, const float* tgt, const float* ref, const float* w, int64_t n,
                           const float* mu_tgt, const float* mu_ref, double* host_out) {
    const int blocks = reduce_grid(n);
    ICP_HIP(ctx, ctx->partials.reserve((size_t)blocks * NEQ * sizeof(double)));
    if (!mu_tgt) {
        hipLaunchKernelGGL(k_procrustes_means, dim3(blocks), dim3(RED_THREADS), 0, ctx->stream, tgt, ref, w, (int)n,
                           ctx->partials.as<double>());
    } else {
        ProcrustesMeans mu;
        for (int a = 0; a < 3; ++a) {
            mu.mu_tgt[a] = mu_tgt[a];
            mu.mu_ref[a] = mu_ref[a];
        }
        hipLaunchKernelGGL(k_procrustes_cov, dim3(blocks), dim3(RED_THREADS), 0, ctx->stream, tgt, ref, (int)n, mu,
                           ctx->partials.as<double>());
    }
    hipLaunchKernelGGL(k_sum_partials, dim3(1), dim3(1024), 0, ctx->stream, ctx->partials.as<double>(), blocks, 1,
                       reg_state(ctx), 0, ctx->neq);
    ICP_HIP(ctx, hipGetLastError());
    ICP_HIP(ctx, hipMemcpyAsync(host_out, ctx->neq, NEQ * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
    ICP_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return ICP_OK;
}

}  // namespace icp
