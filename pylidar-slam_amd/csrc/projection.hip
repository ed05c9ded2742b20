// Spherical range-image projection with a z-buffer.
//
// Replaces `torch__spherical_projection` (slam/common/projection.py:11-73) and `Projector.build_projection_map`
// (:331-418).  The reference sorts all points by descending range and scatters them so that the last (= nearest) write
// wins each pixel (:404-415) — deterministic only single-threaded.  Here every point does one 64-bit atomicMin on
// (range bits << 32 | ~index) per pixel, then a resolve pass gathers the winner: nearest point wins, equal ranges go to
// the highest point index (the outcome of a stable descending sort + last-write-wins).
#include <string.h>

#include "icp_internal.h"
#include "projection_device.h"

namespace icp {

struct ProjParams {
    int height, width;
    float fov_down_abs;  // |down_fov| in rad
    float fov;           // |down| + |up| in rad
};

// float pixel coordinates, float32 operations in the reference's order (:52-73).  `exact`: the angles correctly rounded
// everywhere (the diagnostic output); otherwise only where the rounding to a pixel could depend on them
// (projection_device.h)
__device__ inline void spherical_pixel(float x, float y, float z, const ProjParams& pp, float& row, float& col,
                                       float& range, bool exact = false) {
    const float r_raw = sqrtf(__fadd_rn(__fadd_rn(__fmul_rn(x, x), __fmul_rn(y, y)), __fmul_rn(z, z)));
    range = r_raw;
    if (r_raw == 0.0f) {  // mask_0: row = col = -1 (:55-57,:73)
        row = -1.0f;
        col = -1.0f;
        return;
    }
    if (exact)
        spherical_rowcol(x, y, z, r_raw, pp.fov_down_abs, pp.fov, pp.height, pp.width, true, row, col);
    else
        spherical_rowcol_for_rounding(x, y, z, r_raw, pp.fov_down_abs, pp.fov, pp.height, pp.width, row, col);
}

__global__ void k_zbuf_clear(unsigned long long* __restrict__ zbuf, int npix) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < npix) zbuf[i] = ~0ull;
}

__global__ void k_project(const float* __restrict__ xyz, int n, ProjParams pp, unsigned long long* __restrict__ zbuf) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float x = xyz[3 * i], y = xyz[3 * i + 1], z = xyz[3 * i + 2];
    float row, col, r;
    spherical_pixel(x, y, z, pp, row, col, r);
    const float prow = rintf(row), pcol = rintf(col);  // torch.round = half-to-even (:395-396)
    // invalid outside the image (:398-401); NaN fails every comparison; r must be > 0 (:409)
    if (!(prow >= 0.0f && prow <= (float)(pp.height - 1) && pcol >= 0.0f && pcol <= (float)(pp.width - 1))) return;
    if (!(r > 0.0f)) return;
    const int pix = (int)prow * pp.width + (int)pcol;
    const unsigned long long key = ((unsigned long long)__float_as_uint(r) << 32) | (unsigned long long)(~(unsigned)i);
    atomicMin(&zbuf[pix], key);
}

__global__ void k_project_resolve(const float* __restrict__ xyz, unsigned long long* __restrict__ zbuf, int npix,
                                  float* __restrict__ vmap, int* __restrict__ index, int leave_clean,
                                  float* __restrict__ rows) {
    const int p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= npix) return;
    const unsigned long long k = zbuf[p];
    if (leave_clean && k != ~0ull) zbuf[p] = ~0ull;  // clean for the next projection: no clearing launch per frame
    float x = 0.f, y = 0.f, z = 0.f;
    int idx = -1;
    if (k != ~0ull) {
        idx = (int)(~(unsigned)(k & 0xffffffffull));
        x = xyz[3 * idx];
        y = xyz[3 * idx + 1];
        z = xyz[3 * idx + 2];
    }
    if (vmap) {
        vmap[p] = x;
        vmap[npix + p] = y;
        vmap[2 * npix + p] = z;
    }
    if (index) index[p] = idx;
    if (rows) {  // the same pixels as [H*W, 3] rows (= vmap.permute(1, 2, 0).reshape(-1, 3): what sample_points reads)
        rows[3 * p] = x;
        rows[3 * p + 1] = y;
        rows[3 * p + 2] = z;
    }
}

// B projections per launch (icp_batch_project): blockIdx.y = the member; its arguments ride in the kernel-argument segment
// (48 bytes per member).  Same arithmetic per point and per pixel as k_project / k_project_resolve.
struct ProjBatchEntry {
    const float* xyz;
    unsigned long long* zbuf;
    float* vmap;
    int n, npix;
    ProjParams pp;
};
struct ProjBatchArgs {
    ProjBatchEntry e[ICP_BATCH_MAX_SEQUENCES];
};

__global__ void k_project_batch(ProjBatchArgs a) {
    const ProjBatchEntry& e = a.e[blockIdx.y];
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= e.n) return;
    const float x = e.xyz[3 * i], y = e.xyz[3 * i + 1], z = e.xyz[3 * i + 2];
    float row, col, r;
    spherical_pixel(x, y, z, e.pp, row, col, r);
    const float prow = rintf(row), pcol = rintf(col);
    if (!(prow >= 0.0f && prow <= (float)(e.pp.height - 1) && pcol >= 0.0f && pcol <= (float)(e.pp.width - 1))) return;
    if (!(r > 0.0f)) return;
    const int pix = (int)prow * e.pp.width + (int)pcol;
    const unsigned long long key = ((unsigned long long)__float_as_uint(r) << 32) | (unsigned long long)(~(unsigned)i);
    atomicMin(&e.zbuf[pix], key);
}

__global__ void k_project_resolve_batch(ProjBatchArgs a) {
    const ProjBatchEntry& e = a.e[blockIdx.y];
    const int p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= e.npix) return;
    const unsigned long long k = e.zbuf[p];
    if (k != ~0ull) e.zbuf[p] = ~0ull;  // clean for the next projection
    float x = 0.f, y = 0.f, z = 0.f;
    if (k != ~0ull) {
        const int idx = (int)(~(unsigned)(k & 0xffffffffull));
        x = e.xyz[3 * idx];
        y = e.xyz[3 * idx + 1];
        z = e.xyz[3 * idx + 2];
    }
    e.vmap[p] = x;
    e.vmap[e.npix + p] = y;
    e.vmap[2 * e.npix + p] = z;
}

__global__ void k_project_pixels(const float* __restrict__ xyz, int n, ProjParams pp, float* __restrict__ rows,
                                 float* __restrict__ cols) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float row, col, r;
    spherical_pixel(xyz[3 * i], xyz[3 * i + 1], xyz[3 * i + 2], pp, row, col, r, true);
    rows[i] = row;
    cols[i] = col;
}

// KITTI HDL-64 calibration correction (`KITTIOdometrySequence.correct_scan`, slam/dataset/kitti_dataset.py:202-231):
// every point is rotated by 0.205 degrees about the axis (p x z) / |p x z| (Rodrigues, float32 like the numpy code;
// cos/sin of the angle are float64 scalars there, so the products promote to float64 and the einsum result is float64).
__global__ void k_kitti_correct(const float* __restrict__ scan, int n, int stride, double* __restrict__ out) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float x = scan[(size_t)i * stride], y = scan[(size_t)i * stride + 1], z = scan[(size_t)i * stride + 2];
    // axes = cross(xyz, [0,0,1]) = (y, -x, 0), normalised in float32 (:209-211)
    const float nrm = sqrtf(__fadd_rn(__fmul_rn(y, y), __fmul_rn(x, x)));
    const float ax = y / nrm, ay = -x / nrm;  // az = 0 (0/nrm; NaN only when x = y = 0, as in the reference)
    const float az = 0.0f / nrm;
    const double theta = 0.205 * 3.14159265358979323846 / 180.0;
    const double c = cos(theta), s = sin(theta);
    // rotations = c * eye + s * u_cross + (1 - c) * u_outer   (:216-228), row i of R times xyz
    const double o00 = (double)__fmul_rn(ax, ax), o01 = (double)__fmul_rn(ax, ay), o02 = (double)__fmul_rn(ax, az);
    const double o11 = (double)__fmul_rn(ay, ay), o12 = (double)__fmul_rn(ay, az), o22 = (double)__fmul_rn(az, az);
    const double r00 = c + (1 - c) * o00, r01 = s * (double)(-az) + (1 - c) * o01, r02 = s * (double)ay + (1 - c) * o02;
    const double r10 = s * (double)az + (1 - c) * o01, r11 = c + (1 - c) * o11, r12 = s * (double)(-ax) + (1 - c) * o12;
    const double r20 = s * (double)(-ay) + (1 - c) * o02, r21 = s * (double)ax + (1 - c) * o12, r22 = c + (1 - c) * o22;
    out[3 * (size_t)i] = r00 * x + r01 * y + r02 * z;
    out[3 * (size_t)i + 1] = r10 * x + r11 * y + r12 * z;
    out[3 * (size_t)i + 2] = r20 * x + r21 * y + r22 * z;
}

int kitti_correct_device(icp_ctx* ctx, const float* scan_dev, int64_t n, int stride, double* out_dev) {
    if (n <= 0) return ICP_OK;
    hipLaunchKernelGGL(k_kitti_correct, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, ctx->stream, scan_dev, (int)n,
                       stride, out_dev);
    ICP_HIP(ctx, hipGetLastError());
    return ICP_OK;
}

static ProjParams proj_params(const icp_ctx* ctx) {
    ProjParams pp;
    pp.height = ctx->cfg.height;
    pp.width = ctx->cfg.width;
    const double up = (double)ctx->cfg.up_fov / 180.0 * 3.14159265358979323846;
    const double down = (double)ctx->cfg.down_fov / 180.0 * 3.14159265358979323846;
    const double a_down = down < 0 ? -down : down, a_up = up < 0 ? -up : up;
    pp.fov_down_abs = (float)a_down;
    pp.fov = (float)(a_down + a_up);
    return pp;
}

int project_device(icp_ctx* ctx, const float* xyz_dev, int64_t n, float* vmap_dev, int32_t* index_dev, bool keep_keys,
                   float* rows_dev) {
    const int npix = ctx->cfg.height * ctx->cfg.width;
    ICP_HIP(ctx, ctx->zbuf.reserve((size_t)npix * sizeof(unsigned long long)));
    unsigned long long* zb = ctx->zbuf.as<unsigned long long>();
    const ProjParams pp = proj_params(ctx);
    if (ctx->zbuf_clean != zb || ctx->zbuf_clean_pixels != npix) {  // a fresh allocation / another image size
        hipLaunchKernelGGL(k_zbuf_clear, dim3((npix + 255) / 256), dim3(256), 0, ctx->stream, zb, npix);
        ctx->zbuf_clean = zb;
        ctx->zbuf_clean_pixels = npix;
    }
    if (n > 0)
        hipLaunchKernelGGL(k_project, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, ctx->stream, xyz_dev, (int)n, pp,
                           zb);
    hipLaunchKernelGGL(k_project_resolve, dim3((npix + 255) / 256), dim3(256), 0, ctx->stream, xyz_dev, zb, npix,
                       vmap_dev, index_dev, keep_keys ? 0 : 1, rows_dev);
    if (keep_keys) ctx->zbuf_clean = nullptr;  // the caller reads the (range, ~index) keys: cleared by the next projection
    ICP_HIP(ctx, hipGetLastError());
    return ICP_OK;
}

// vertex maps of `count` scans (device pointers) in two launches; every member keeps its own z-buffer
int project_batch_device(icp_ctx* const* ctxs, int count, const float* const* xyz_dev, const int64_t* n, float* const* vmap_dev) {
    ProjBatchArgs a;
    memset(&a, 0, sizeof(a));
    int max_n = 0, max_pix = 0;
    icp_ctx* first = ctxs[0];
    for (int b = 0; b < count; ++b) {
        icp_ctx* ctx = ctxs[b];
        const int npix = ctx->cfg.height * ctx->cfg.width;
        ICP_HIP(ctx, ctx->zbuf.reserve((size_t)npix * sizeof(unsigned long long)));
        unsigned long long* zb = ctx->zbuf.as<unsigned long long>();
        if (ctx->zbuf_clean != zb || ctx->zbuf_clean_pixels != npix) {  // a fresh allocation / another image size
            hipLaunchKernelGGL(k_zbuf_clear, dim3((npix + 255) / 256), dim3(256), 0, ctx->stream, zb, npix);
            ctx->zbuf_clean = zb;
            ctx->zbuf_clean_pixels = npix;
        }
        a.e[b].xyz = xyz_dev[b];
        a.e[b].zbuf = zb;
        a.e[b].vmap = vmap_dev[b];
        a.e[b].n = (int)n[b];
        a.e[b].npix = npix;
        a.e[b].pp = proj_params(ctx);
        max_n = (int)n[b] > max_n ? (int)n[b] : max_n;
        max_pix = npix > max_pix ? npix : max_pix;
    }
    if (max_n > 0)
        hipLaunchKernelGGL(k_project_batch, dim3((unsigned)((max_n + 255) / 256), count), dim3(256), 0, first->stream, a);
    hipLaunchKernelGGL(k_project_resolve_batch, dim3((unsigned)((max_pix + 255) / 256), count), dim3(256), 0, first->stream, a);
    ICP_HIP(first, hipGetLastError());
    return ICP_OK;
}

int project_pixels_device(icp_ctx* ctx, const float* xyz_dev, int64_t n, float* rows_dev, float* cols_dev) {
    if (n <= 0) return ICP_OK;
    hipLaunchKernelGGL(k_project_pixels, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, ctx->stream, xyz_dev, (int)n,
                       proj_params(ctx), rows_dev, cols_dev);
    ICP_HIP(ctx, hipGetLastError());
    return ICP_OK;
}

}  // namespace icp
