// Re-expression of the local map in a new frame (local_map.py:346-348), shared by api.hip (host-side checks) and
// hash_grid.hip (the re-expression rides in the first launch of the grid build).
#pragma once
#include <math.h>

#include "icp_internal.h"

namespace icp {

// general 4x4 inverse in f64 (np.linalg.inv(relative_pose), local_map.py:346); the same code on host and device
__host__ __device__ inline bool invert4(const float* m, float* out) {
    double a[4][8];
    for (int r = 0; r < 4; ++r)
        for (int c = 0; c < 4; ++c) {
            a[r][c] = m[4 * r + c];
            a[r][4 + c] = r == c ? 1.0 : 0.0;
        }
    for (int c = 0; c < 4; ++c) {
        int piv = c;
        for (int r = c + 1; r < 4; ++r)
            if (fabs(a[r][c]) > fabs(a[piv][c])) piv = r;
        if (a[piv][c] == 0.0) return false;
        if (piv != c)
            for (int k = 0; k < 8; ++k) {
                const double t = a[c][k];
                a[c][k] = a[piv][k];
                a[piv][k] = t;
            }
        const double inv = 1.0 / a[c][c];
        for (int k = 0; k < 8; ++k) a[c][k] *= inv;
        for (int r = 0; r < 4; ++r)
            if (r != c) {
                const double f = a[r][c];
#if defined(__HIP_DEVICE_COMPILE__)
                for (int k = 0; k < 8; ++k) a[r][k] = __dsub_rn(a[r][k], __dmul_rn(f, a[c][k]));  // no fma: host bits
#else
                for (int k = 0; k < 8; ++k) a[r][k] -= f * a[c][k];
#endif
            }
    }
    for (int r = 0; r < 4; ++r)
        for (int c = 0; c < 4; ++c) out[4 * r + c] = (float)a[r][4 + c];
    return true;
}

// moved = R^-1 x + t^-1 over the kept part of the map (local_map.py:346-348)
__device__ inline void move_point(const float* T, const float* __restrict__ in, long long i, float* __restrict__ out) {
    const float x = in[3 * i], y = in[3 * i + 1], z = in[3 * i + 2];
    // np.einsum("ij,nj->ni", R, map) + t
    out[3 * i] = __fadd_rn(__fadd_rn(__fadd_rn(__fmul_rn(T[0], x), __fmul_rn(T[1], y)), __fmul_rn(T[2], z)), T[3]);
    out[3 * i + 1] = __fadd_rn(__fadd_rn(__fadd_rn(__fmul_rn(T[4], x), __fmul_rn(T[5], y)), __fmul_rn(T[6], z)), T[7]);
    out[3 * i + 2] = __fadd_rn(__fadd_rn(__fadd_rn(__fmul_rn(T[8], x), __fmul_rn(T[9], y)), __fmul_rn(T[10], z)), T[11]);
}

// Every block inverts the 4x4 once: one arithmetic for both sources of the pose (host value / device-resident result),
// so the two give the same map bit for bit.
// T (LDS, 16 floats) <- the inverse this launch applies; call from one thread, barrier afterwards
__device__ inline void map_move_prepare(const MapMoveJob& job, float* __restrict__ T) {
    float pose[16], inv[16];
    for (int k = 0; k < 16; ++k) pose[k] = job.st ? job.st->pose[k] : job.rel.m[k];
    // a registration that stopped on an error moves nothing: the reference raises before it would touch the map
    // (slam/common/optimization.py:334-336 inside icp_odometry.py:286), the host learns of it in icp_register_end
    // ... and so does one whose loop was cut short by a timed-out hand-off (icp_register_end finishes it on per-iteration
    // launches and repeats this update with the final pose)
    const bool failed = job.st && (job.st->status != ICP_OK || job.st->handoff_timeouts > 0);
    if (failed || !invert4(pose, inv))  // a pose built from Euler angles is never singular; the host path checks
        for (int k = 0; k < 16; ++k) inv[k] = (k % 5 == 0) ? 1.f : 0.f;
    for (int k = 0; k < 16; ++k) T[k] = inv[k];
}

}  // namespace icp
