// Device-side pieces of the point-to-plane Gauss-Newton rows shared by gauss_newton.hip and the fused iteration kernel.
#pragma once
#include "icp_internal.h"

namespace icp {

__device__ inline float robust_weight(int scheme, float sigma, float r, float dist2_pq) {
    // slam/common/optimization.py:45-50 with the per-scheme cost(); least_square short-circuits to 1 (:70-72).
    // Evaluated operation by operation like the reference (sigma + r * r is a rounded product then a rounded sum): the
    // library is built with -ffp-contract=off, so every kernel that inlines this forms the same weight bit for bit.
    if (scheme == ICP_SCHEME_LEAST_SQUARE) return 1.0f;
    const float a = fabsf(r);
    float cost;
    switch (scheme) {
        case ICP_SCHEME_HUBER:  // :87-97
            cost = a < sigma ? r * r : (2.0f * sigma * a - sigma * sigma);
            break;
        case ICP_SCHEME_EXP:  // :110-117
            cost = (r * r) * expf(-(r * r) / (sigma * sigma));
            break;
        case ICP_SCHEME_NEIGHBORHOOD: {  // :132-145  exp(-||p - q||^2 / sigma^2), norm taken then squared
            const float nrm = sqrtf(dist2_pq);
            cost = r * r * expf(-(nrm * nrm) / (sigma * sigma));
            break;
        }
        case ICP_SCHEME_GEMAN_MCCLURE: {  // :158-166
            const float r2 = r * r;
            cost = sigma * r2 / (sigma + r2);
            break;
        }
        case ICP_SCHEME_SQUARE_GEMAN_MCCLURE: {  // :179-187
            const float r2 = r * r;
            const float q = sigma / (sigma + r2);
            cost = r2 * (q * q);
            break;
        }
        case ICP_SCHEME_CAUCHY: {  // :200-208
            const float q = r / sigma;
            cost = logf(1.0f + q * q);
            break;
        }
        default:
            cost = r * r;
    }
    return sqrtf(cost) / fmaxf(a, 1.0e-4f);
}

// One correspondence p (transformed target) <-> q (map point) with normal n, float32 exactly as the reference forms it:
//   r = ((p - q) * n).sum(-1)                      slam/common/optimization.py:427-431
//   J = [n, p x n]                                 :378-390 at x0 = 0
//   w = sqrt(cost(r)) / clamp(|r|, 1e-4)           :45-50  ;  res *= w, J *= w  (:329-330)
// out = { Jw[0..5], r*w, r, 1 }
__device__ inline void point_to_plane_row(float px, float py, float pz, float qx, float qy, float qz, float nx, float ny,
                                          float nz, int scheme, float sigma, float* __restrict__ out) {
    const float dx = __fsub_rn(px, qx), dy = __fsub_rn(py, qy), dz = __fsub_rn(pz, qz);
    const float r = __fadd_rn(__fadd_rn(__fmul_rn(dx, nx), __fmul_rn(dy, ny)), __fmul_rn(dz, nz));
    const float j3 = __fsub_rn(__fmul_rn(py, nz), __fmul_rn(pz, ny));
    const float j4 = __fsub_rn(__fmul_rn(pz, nx), __fmul_rn(px, nz));
    const float j5 = __fsub_rn(__fmul_rn(px, ny), __fmul_rn(py, nx));
    const float d2 = __fadd_rn(__fadd_rn(__fmul_rn(dx, dx), __fmul_rn(dy, dy)), __fmul_rn(dz, dz));
    const float w = robust_weight(scheme, sigma, r, d2);
    out[0] = __fmul_rn(nx, w);
    out[1] = __fmul_rn(ny, w);
    out[2] = __fmul_rn(nz, w);
    out[3] = __fmul_rn(j3, w);
    out[4] = __fmul_rn(j4, w);
    out[5] = __fmul_rn(j5, w);
    out[6] = __fmul_rn(r, w);
    out[7] = r;
    out[8] = 1.0f;
}

// packed normal-equation element e (0..29) = sum over rows of x[NEQ_A[e]] * x[NEQ_B[e]] (x = the 9 floats above):
// 21 upper-triangular JtJ, 6 Jtr, loss = sum (w r)^2, sum r^2, row count
__device__ inline void neq_operands(int e, int& a, int& b) {
    if (e < 21) {
        int k = e, row = 0;
        while (k >= 6 - row) {
            k -= 6 - row;
            ++row;
        }
        a = row;
        b = row + k;
    } else if (e < 27) {
        a = e - 21;
        b = 6;
    } else if (e == 27) {
        a = 6;
        b = 6;
    } else if (e == 28) {
        a = 7;
        b = 7;
    } else {
        a = 8;
        b = 8;
    }
}

// the same pair from two packed tables (four bits per element): the loop above costs the summing threads of the fused kernel
// ~20 integer instructions each — a thirtieth of what a late launch issues (profiles/r06_valu_attribution.txt); e in 0 .. 31
// (30, 31: unused elements, operands 0 x 0 of a thread that adds nothing)
__device__ inline void neq_operands_lut(int e, int& a, int& b) {
    const unsigned long long ta = e < 16 ? 0x3222211111000000ull : 0x0087654321054433ull;
    const unsigned long long tb = e < 16 ? 0x3543254321543210ull : 0x0087666666655454ull;
    const int sh = (e & 15) * 4;
    a = (int)((ta >> sh) & 15ull);
    b = (int)((tb >> sh) & 15ull);
}

}  // namespace icp
