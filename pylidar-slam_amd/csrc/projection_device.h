// Float pixel coordinates of `torch__spherical_projection` (slam/common/projection.py:11-73) for the two kernels that
// round them to a pixel (projection.hip, projective.hip).
//
// The reference evaluates atan2 / asin in float32 and rounds the scaled result to the nearest integer (:395-396), so a
// point whose coordinate falls within an ulp or two of x.5 lands in one pixel or the other depending on the LAST BIT of
// those two functions — which differs between the reference's own code paths (PyTorch's Sleef kernels under AVX2 /
// AVX512 vs the scalar libm path: 2 % of the coordinates of a 64x2048 scan, by up to 2.3e-5 pixel;
// oracle/make_golden_projection_spread.py, tests/golden/projection_spread.npz).  No float32 routine can match all of
// them, so the device does the one defensible thing: away from a rounding boundary the fast float32 ocml functions decide
// (their few ulp cannot change the pixel); within 2e-3 pixel of a boundary the angles are re-evaluated in float64 and
// rounded once to float32 — the correctly rounded float32 value, at most one ulp from any faithful float32 routine.
#pragma once
#include <hip/hip_runtime.h>

namespace icp {

// angles in float32; exact = correctly rounded (through float64)
__device__ inline void spherical_angles(float x, float y, float q /* z / r */, bool exact, float& theta, float& phi) {
    if (exact) {
        theta = -(float)atan2((double)y, (double)x);
        phi = (float)asin((double)q);
    } else {
        theta = -atan2f(y, x);  // :64
        phi = asinf(q);         // :65
    }
}

// float32 operations in the reference's order (:52-73); r = norm of the point (> 0)
__device__ inline void spherical_rowcol(float x, float y, float z, float r, float fov_down_abs, float fov, int height,
                                        int width, bool exact, float& row, float& col) {
    float theta, phi;
    spherical_angles(x, y, z / r, exact, theta, phi);
    const float pc = 0.5f * (theta / 3.14159265358979323846f + 1.0f);  // :67
    const float pr = 1.0f - (phi + fov_down_abs) / fov;                // :68
    col = pc * (float)width;                                           // :70
    row = pr * (float)height;                                          // :71
}

__device__ inline bool near_half(float v) { return fabsf(v - floorf(v) - 0.5f) < 2.0e-3f; }

// the coordinates the pixel is rounded from: fast path, refined next to a rounding boundary
__device__ inline void spherical_rowcol_for_rounding(float x, float y, float z, float r, float fov_down_abs, float fov,
                                                     int height, int width, float& row, float& col) {
    spherical_rowcol(x, y, z, r, fov_down_abs, fov, height, width, false, row, col);
    if (near_half(row) || near_half(col)) spherical_rowcol(x, y, z, r, fov_down_abs, fov, height, width, true, row, col);
}

}  // namespace icp
