// Projective local map: the reference's own "GPU" variant of the ICP hot path, as fused HIP kernels.
//
//   compute_normal_map   slam/common/geometry.py:240-295   k_normal_map     box-filter plane fit per pixel
//   compute_neighbors    slam/common/geometry.py:397-439   k_neighbors      per-pixel argmin over K vertex maps
//   ProjectiveLocalMap   slam/odometry/local_map.py:91-240 pmap_update / pmap_build / k_pm_iterate
//
// The reference forms sum_box(p p^T) and its adjugate inverse in float32; with |p| ~ 10-30 m that cancels
// catastrophically — its own normals sit a median 3e-4 rad (p99 5e-3) from the exact value (measured in
// oracle/make_golden_projective.py's data).  Here the window sums and the 3x3 inverse are done in float64 and rounded
// once, i.e. the result is the exact value the reference approximates.
//
// Per ICP iteration the reference projects the N transformed target points again (local_map.py:216), takes per pixel
// the closest of the K stored maps and masks null pixels.  Fused here into: z-buffer projection of the transformed
// targets (atomicMin) -> one kernel that, per pixel, re-derives the winning target point, scans the K model maps,
// forms the point-to-plane row and reduces the block's rows to one partial (same packed layout / final solve as the
// kd-tree path).
#include <string.h>

#include "gn_device.h"
#include "icp_internal.h"
#include "projection_device.h"

namespace icp {

struct PoseArg {
    float m[16];
};

// ---------------------------------------------------------------------------------------------------------------------
// normal map
// ---------------------------------------------------------------------------------------------------------------------
__global__ void k_normal_map(const float* __restrict__ vmap, int h, int w, int ks, float* __restrict__ nmap) {
    const int npix = h * w;
    const int p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= npix) return;
    const int y = p / w, x = p % w, r = ks / 2;
    const float vx = vmap[p], vy = vmap[npix + p], vz = vmap[2 * npix + p];
    float nx = 0.f, ny = 0.f, nz = 0.f;
    // mask_null: norm == 0 (geometry.py:279)
    if (!(vx == 0.f && vy == 0.f && vz == 0.f)) {
        double s0 = 0, s1 = 0, s2 = 0, a00 = 0, a01 = 0, a02 = 0, a11 = 0, a12 = 0, a22 = 0;
        for (int dy = -r; dy <= r; ++dy) {
            const int yy = y + dy;
            if (yy < 0 || yy >= h) continue;  // zero padding
            for (int dx = -r; dx <= r; ++dx) {
                const int xx = x + dx;
                if (xx < 0 || xx >= w) continue;
                const int q = yy * w + xx;
                const double qx = vmap[q], qy = vmap[npix + q], qz = vmap[2 * npix + q];
                s0 += qx;
                s1 += qy;
                s2 += qz;
                a00 += qx * qx;
                a01 += qx * qy;
                a02 += qx * qz;
                a11 += qy * qy;
                a12 += qy * qz;
                a22 += qz * qz;
            }
        }
        // adjugate of the symmetric A and det (geometry.py:63-98); n = A^-1 s
        const double c00 = a11 * a22 - a12 * a12, c01 = a02 * a12 - a01 * a22, c02 = a01 * a12 - a02 * a11;
        const double c11 = a00 * a22 - a02 * a02, c12 = a01 * a02 - a00 * a12, c22 = a00 * a11 - a01 * a01;
        const double det = a00 * c00 + a01 * c01 + a02 * c02;
        if (fabs(det) > 1.0e-6) {  // :272-273
            const double inv = 1.0 / det;
            const double mx = (c00 * s0 + c01 * s1 + c02 * s2) * inv;
            const double my = (c01 * s0 + c11 * s1 + c12 * s2) * inv;
            const double mz = (c02 * s0 + c12 * s1 + c22 * s2) * inv;
            const double nrm = sqrt(mx * mx + my * my + mz * mz);
            if (nrm > 0.0) {
                nx = (float)(mx / nrm);
                ny = (float)(my / nrm);
                nz = (float)(mz / nrm);
            }
        }
    }
    nmap[p] = nx;
    nmap[npix + p] = ny;
    nmap[2 * npix + p] = nz;
}

int normal_map_device(icp_ctx* ctx, const float* vmap_dev, int ks, float* nmap_dev) {
    const int npix = ctx->cfg.height * ctx->cfg.width;
    hipLaunchKernelGGL(k_normal_map, dim3((npix + 255) / 256), dim3(256), 0, ctx->stream, vmap_dev, ctx->cfg.height,
                       ctx->cfg.width, ks, nmap_dev);
    ICP_HIP(ctx, hipGetLastError());
    return ICP_OK;
}

// ---------------------------------------------------------------------------------------------------------------------
// compute_neighbors on caller-supplied planar maps: target [3,HW], reference [K,3,HW], fields [K,C,HW] (optional)
// ---------------------------------------------------------------------------------------------------------------------
__global__ void k_neighbors(const float* __restrict__ tgt, const float* __restrict__ ref, const float* __restrict__ fld,
                            int k_maps, int c_fld, int npix, float* __restrict__ nb_out, float* __restrict__ fld_out) {
    const int p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= npix) return;
    const float tx = tgt[p], ty = tgt[npix + p], tz = tgt[2 * npix + p];
    const bool t_ok = fmaxf(fmaxf(fabsf(tx), fabsf(ty)), fabsf(tz)) > 0.f;  // mask_not_null
    float best = INFINITY;
    int bi = 0;  // all-inf -> index 0, like torch.min
    for (int k = 0; k < k_maps; ++k) {
        const float* r = ref + (size_t)k * 3 * npix;
        const float rx = r[p], ry = r[npix + p], rz = r[2 * npix + p];
        const bool r_ok = fmaxf(fmaxf(fabsf(rx), fabsf(ry)), fabsf(rz)) > 0.f;
        const float dx = tx - rx, dy = ty - ry, dz = tz - rz;
        const float d = (t_ok && r_ok) ? sqrtf(__fadd_rn(__fadd_rn(__fmul_rn(dx, dx), __fmul_rn(dy, dy)), __fmul_rn(dz, dz)))
                                       : INFINITY;
        if (d < best) {  // first minimum wins
            best = d;
            bi = k;
        }
    }
    const float* r = ref + (size_t)bi * 3 * npix;
    nb_out[p] = t_ok ? r[p] : 0.f;  // :431
    nb_out[npix + p] = t_ok ? r[npix + p] : 0.f;
    nb_out[2 * npix + p] = t_ok ? r[2 * npix + p] : 0.f;
    if (fld && fld_out)
        for (int c = 0; c < c_fld; ++c) fld_out[(size_t)c * npix + p] = fld[((size_t)bi * c_fld + c) * npix + p];
}

int neighbors_device(icp_ctx* ctx, const float* tgt, const float* ref, const float* fld, int k_maps, int c_fld,
                     float* nb_out, float* fld_out) {
    const int npix = ctx->cfg.height * ctx->cfg.width;
    hipLaunchKernelGGL(k_neighbors, dim3((npix + 255) / 256), dim3(256), 0, ctx->stream, tgt, ref, fld, k_maps, c_fld,
                       npix, nb_out, fld_out);
    ICP_HIP(ctx, hipGetLastError());
    return ICP_OK;
}

// ---------------------------------------------------------------------------------------------------------------------
// the stored maps and the model (ProjectiveLocalMap.update / build_model)
//   slots: pm_v[k] / pm_n[k] = float4 [HW]: own-frame vertex (w = 1 valid / 0 null) and normal; pm_pose[k] = pose of
//   the slot's frame in the CURRENT frame.  model_v / model_n [K][HW] float4 = every slot re-expressed in the current
//   frame and re-projected (6 channels riding on the z-buffer winner).
// ---------------------------------------------------------------------------------------------------------------------
struct ProjArg {
    int height, width;
    float fov_down_abs, fov;
};

__device__ inline bool pixel_of(float x, float y, float z, const ProjArg& pp, int& pix, float& range) {
    const float r = sqrtf(__fadd_rn(__fadd_rn(__fmul_rn(x, x), __fmul_rn(y, y)), __fmul_rn(z, z)));
    range = r;
    if (!(r > 0.f)) return false;
    float row, col;  // (refined next to a rounding boundary: projection_device.h)
    spherical_rowcol_for_rounding(x, y, z, r, pp.fov_down_abs, pp.fov, pp.height, pp.width, row, col);
    const float prow = rintf(row), pcol = rintf(col);
    if (!(prow >= 0.f && prow <= (float)(pp.height - 1) && pcol >= 0.f && pcol <= (float)(pp.width - 1))) return false;
    pix = (int)prow * pp.width + (int)pcol;
    return true;
}

__global__ void k_pm_store(const float* __restrict__ vmap, const float* __restrict__ nmap, int npix,
                           float4* __restrict__ v4, float4* __restrict__ n4) {
    const int p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= npix) return;
    const float x = vmap[p], y = vmap[npix + p], z = vmap[2 * npix + p];
    const bool ok = fmaxf(fmaxf(fabsf(x), fabsf(y)), fabsf(z)) > 0.f;  // mask_not_null (local_map.py:141)
    v4[p] = make_float4(x, y, z, ok ? 1.f : 0.f);
    n4[p] = make_float4(nmap[p], nmap[npix + p], nmap[2 * npix + p], 0.f);
}

__global__ void k_zclear(unsigned long long* __restrict__ z, long long n) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) z[i] = ~0ull;
}

// one slot: transform by its pose, project, z-buffer
__global__ void k_pm_project(const float4* __restrict__ v4, int npix, PoseArg T, ProjArg pp,
                             unsigned long long* __restrict__ zbuf) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= npix) return;
    const float4 v = v4[i];
    if (v.w == 0.f) return;  // model_points *= mask (local_map.py:192-193): null points project nowhere
    const float x = fmaf(v.z, T.m[2], fmaf(v.y, T.m[1], v.x * T.m[0])) + T.m[3];
    const float y = fmaf(v.z, T.m[6], fmaf(v.y, T.m[5], v.x * T.m[4])) + T.m[7];
    const float z = fmaf(v.z, T.m[10], fmaf(v.y, T.m[9], v.x * T.m[8])) + T.m[11];
    int pix;
    float r;
    if (!pixel_of(x, y, z, pp, pix, r)) return;
    atomicMin(&zbuf[pix], ((unsigned long long)__float_as_uint(r) << 32) | (unsigned long long)(~(unsigned)i));
}

__global__ void k_pm_resolve(const float4* __restrict__ v4, const float4* __restrict__ n4, int npix, PoseArg T,
                             const unsigned long long* __restrict__ zbuf, float4* __restrict__ mv,
                             float4* __restrict__ mn) {
    const int p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= npix) return;
    const unsigned long long k = zbuf[p];
    float4 ov = make_float4(0.f, 0.f, 0.f, 0.f), on = make_float4(0.f, 0.f, 0.f, 0.f);
    if (k != ~0ull) {
        const int i = (int)(~(unsigned)(k & 0xffffffffull));
        const float4 v = v4[i], n = n4[i];
        ov.x = fmaf(v.z, T.m[2], fmaf(v.y, T.m[1], v.x * T.m[0])) + T.m[3];
        ov.y = fmaf(v.z, T.m[6], fmaf(v.y, T.m[5], v.x * T.m[4])) + T.m[7];
        ov.z = fmaf(v.z, T.m[10], fmaf(v.y, T.m[9], v.x * T.m[8])) + T.m[11];
        ov.w = 1.f;
        on.x = fmaf(n.z, T.m[2], fmaf(n.y, T.m[1], n.x * T.m[0]));  // apply_rotation (pose.py:154-167)
        on.y = fmaf(n.z, T.m[6], fmaf(n.y, T.m[5], n.x * T.m[4]));
        on.z = fmaf(n.z, T.m[10], fmaf(n.y, T.m[9], n.x * T.m[8]));
    }
    mv[p] = ov;
    mn[p] = on;
}

// ---------------------------------------------------------------------------------------------------------------------
// per-iteration kernels
// ---------------------------------------------------------------------------------------------------------------------
__global__ void k_pm_project_targets(const float4* __restrict__ tgt, int n, int mode, const RegState* __restrict__ st,
                                     ProjArg pp, unsigned long long* __restrict__ zbuf) {
    if (st->done) return;
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float4 t = tgt[i];
    if (!(t.x == t.x) || !(t.y == t.y) || !(t.z == t.z)) return;
    if (mode == ICP_TARGETS_SKIP_NULL && t.x == 0.f && t.y == 0.f && t.z == 0.f) return;
    const float* T = st->pose;
    const float x = fmaf(t.z, T[2], fmaf(t.y, T[1], t.x * T[0])) + T[3];
    const float y = fmaf(t.z, T[6], fmaf(t.y, T[5], t.x * T[4])) + T[7];
    const float z = fmaf(t.z, T[10], fmaf(t.y, T[9], t.x * T[8])) + T[11];
    int pix;
    float r;
    if (!pixel_of(x, y, z, pp, pix, r)) return;
    atomicMin(&zbuf[pix], ((unsigned long long)__float_as_uint(r) << 32) | (unsigned long long)(~(unsigned)i));
}

static constexpr int PM_THREADS = 256;

__global__ __launch_bounds__(PM_THREADS) void k_pm_iterate(const float4* __restrict__ tgt, int npix, int k_maps,
                                                           const unsigned long long* __restrict__ zbuf,
                                                           const float4* __restrict__ mv, const float4* __restrict__ mn,
                                                           const RegState* __restrict__ st, AlignParams ap,
                                                           double* __restrict__ partials) {
    __shared__ float rowbuf[PM_THREADS][9];
    __shared__ double part[8][NEQ];
    if (st->done) return;
    const int p = blockIdx.x * blockDim.x + threadIdx.x;
    float row[9];
#pragma unroll
    for (int k = 0; k < 9; ++k) row[k] = 0.f;
    if (p < npix) {
        const unsigned long long key = zbuf[p];
        if (key != ~0ull) {  // a target point landed here (mask_not_null(new_points), local_map.py:222)
            const int i = (int)(~(unsigned)(key & 0xffffffffull));
            const float4 t = tgt[i];
            const float* T = st->pose;
            const float px = fmaf(t.z, T[2], fmaf(t.y, T[1], t.x * T[0])) + T[3];
            const float py = fmaf(t.z, T[6], fmaf(t.y, T[5], t.x * T[4])) + T[7];
            const float pz = fmaf(t.z, T[10], fmaf(t.y, T[9], t.x * T[8])) + T[11];
            if (fmaxf(fmaxf(fabsf(px), fabsf(py)), fabsf(pz)) > 0.f) {
                float best = INFINITY;
                int bk = -1;
                for (int k = 0; k < k_maps; ++k) {  // compute_neighbors (geometry.py:415-424)
                    const float4 q = mv[(size_t)k * npix + p];
                    if (q.w == 0.f) continue;
                    const float dx = px - q.x, dy = py - q.y, dz = pz - q.z;
                    const float d = sqrtf(__fadd_rn(__fadd_rn(__fmul_rn(dx, dx), __fmul_rn(dy, dy)), __fmul_rn(dz, dz)));
                    if (d < best) {
                        best = d;
                        bk = k;
                    }
                }
                if (bk >= 0) {
                    const float4 q = mv[(size_t)bk * npix + p];
                    if (fmaxf(fmaxf(fabsf(q.x), fabsf(q.y)), fabsf(q.z)) > 0.f) {  // mask_not_null(neighbor_points)
                        const float4 nn = mn[(size_t)bk * npix + p];
                        point_to_plane_row(px, py, pz, q.x, q.y, q.z, nn.x, nn.y, nn.z, ap.scheme, ap.sigma, row);
                    }
                }
            }
        }
    }
#pragma unroll
    for (int k = 0; k < 9; ++k) rowbuf[threadIdx.x][k] = row[k];
    __syncthreads();
    if (threadIdx.x < 8 * NEQ) {  // 8 x 30 threads: element e of eighth `g8` of the block's pixels, fixed order
        const int e = threadIdx.x & (NEQ - 1), g8 = threadIdx.x / NEQ;
        double acc = 0.0;
        if (e < NEQ_USED) {
            int a, b;
            neq_operands(e, a, b);
            const int j0 = g8 * (PM_THREADS / 8);
#pragma unroll 8
            for (int j = 0; j < PM_THREADS / 8; ++j) acc += (double)rowbuf[j0 + j][a] * (double)rowbuf[j0 + j][b];
        }
        part[g8][e] = acc;
    }
    __syncthreads();
    if (threadIdx.x < NEQ) {
        double s = 0.0;
#pragma unroll
        for (int g8 = 0; g8 < 8; ++g8) s += part[g8][threadIdx.x];
        partials[(size_t)blockIdx.x * NEQ + threadIdx.x] = s;
    }
}

// gather of the per-pixel association for the LocalMap.nearest_neighbor_search seam (local_map.py:205-235):
// planar neighbour points / normals / target points per pixel + validity flag
__global__ void k_pm_assoc(const float* __restrict__ tgt_xyz, int npix, int k_maps,
                           const unsigned long long* __restrict__ zbuf, const float4* __restrict__ mv,
                           const float4* __restrict__ mn, float* __restrict__ rows9, int* __restrict__ flags) {
    const int p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= npix) return;
    int ok = 0;
    float o[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
    const unsigned long long key = zbuf[p];
    if (key != ~0ull) {
        const int i = (int)(~(unsigned)(key & 0xffffffffull));
        const float px = tgt_xyz[3 * i], py = tgt_xyz[3 * i + 1], pz = tgt_xyz[3 * i + 2];
        float best = INFINITY;
        int bk = -1;
        for (int k = 0; k < k_maps; ++k) {
            const float4 q = mv[(size_t)k * npix + p];
            if (q.w == 0.f) continue;
            const float dx = px - q.x, dy = py - q.y, dz = pz - q.z;
            const float d = sqrtf(__fadd_rn(__fadd_rn(__fmul_rn(dx, dx), __fmul_rn(dy, dy)), __fmul_rn(dz, dz)));
            if (d < best) {
                best = d;
                bk = k;
            }
        }
        if (bk >= 0 && fmaxf(fmaxf(fabsf(px), fabsf(py)), fabsf(pz)) > 0.f) {
            const float4 q = mv[(size_t)bk * npix + p], nn = mn[(size_t)bk * npix + p];
            if (fmaxf(fmaxf(fabsf(q.x), fabsf(q.y)), fabsf(q.z)) > 0.f) {
                ok = 1;
                o[0] = q.x; o[1] = q.y; o[2] = q.z;
                o[3] = nn.x; o[4] = nn.y; o[5] = nn.z;
                o[6] = px; o[7] = py; o[8] = pz;
            }
        }
    }
    flags[p] = ok;
    for (int c = 0; c < 9; ++c) rows9[(size_t)p * 9 + c] = o[c];
}

static ProjArg proj_arg(const icp_ctx* ctx) {
    ProjArg pp;
    pp.height = ctx->cfg.height;
    pp.width = ctx->cfg.width;
    const double up = (double)ctx->cfg.up_fov / 180.0 * 3.14159265358979323846;
    const double down = (double)ctx->cfg.down_fov / 180.0 * 3.14159265358979323846;
    const double a_down = down < 0 ? -down : down, a_up = up < 0 ? -up : up;
    pp.fov_down_abs = (float)a_down;
    pp.fov = (float)(a_down + a_up);
    return pp;
}

// ---- host side of the map --------------------------------------------------------------------------------------------
int pmap_store_slot(icp_ctx* ctx, int slot, const float* vmap_dev, const float* nmap_dev) {
    const int npix = ctx->cfg.height * ctx->cfg.width;
    const size_t cap = (size_t)(ctx->cfg.local_map_size + 1);
    ICP_HIP(ctx, ctx->pm_v.reserve(cap * npix * sizeof(float4), true, ctx->stream));
    ICP_HIP(ctx, ctx->pm_n.reserve(cap * npix * sizeof(float4), true, ctx->stream));
    hipLaunchKernelGGL(k_pm_store, dim3((npix + 255) / 256), dim3(256), 0, ctx->stream, vmap_dev, nmap_dev, npix,
                       ctx->pm_v.as<float4>() + (size_t)slot * npix, ctx->pm_n.as<float4>() + (size_t)slot * npix);
    ICP_HIP(ctx, hipGetLastError());
    return ICP_OK;
}

// build_model (local_map.py:177-202): every stored slot -> current frame -> re-projection
int pmap_build(icp_ctx* ctx) {
    const int npix = ctx->cfg.height * ctx->cfg.width;
    const int k_maps = (int)ctx->pm_slots.size();
    if (k_maps == 0) return ICP_OK;
    ICP_HIP(ctx, ctx->pm_mv.reserve((size_t)k_maps * npix * sizeof(float4)));
    ICP_HIP(ctx, ctx->pm_mn.reserve((size_t)k_maps * npix * sizeof(float4)));
    ICP_HIP(ctx, ctx->pm_z.reserve((size_t)k_maps * npix * sizeof(unsigned long long)));
    unsigned long long* z = ctx->pm_z.as<unsigned long long>();
    const long long nz = (long long)k_maps * npix;
    hipLaunchKernelGGL(k_zclear, dim3((unsigned)((nz + 255) / 256)), dim3(256), 0, ctx->stream, z, nz);
    const ProjArg pp = proj_arg(ctx);
    const unsigned nb = (npix + 255) / 256;
    for (int k = 0; k < k_maps; ++k) {
        PoseArg T;
        memcpy(T.m, ctx->pm_poses[k].m, sizeof(T.m));
        const float4* v4 = ctx->pm_v.as<float4>() + (size_t)ctx->pm_slots[k] * npix;
        const float4* n4 = ctx->pm_n.as<float4>() + (size_t)ctx->pm_slots[k] * npix;
        hipLaunchKernelGGL(k_pm_project, dim3(nb), dim3(256), 0, ctx->stream, v4, npix, T, pp, z + (size_t)k * npix);
        hipLaunchKernelGGL(k_pm_resolve, dim3(nb), dim3(256), 0, ctx->stream, v4, n4, npix, T, z + (size_t)k * npix,
                           ctx->pm_mv.as<float4>() + (size_t)k * npix, ctx->pm_mn.as<float4>() + (size_t)k * npix);
    }
    ICP_HIP(ctx, hipGetLastError());
    return ICP_OK;
}

// one ICP iteration against the projective model: project targets, associate + rows + partials
int pmap_iterate(icp_ctx* ctx, int* blocks_out) {
    const int npix = ctx->cfg.height * ctx->cfg.width;
    const int n = (int)ctx->tgt_n;
    const int k_maps = (int)ctx->pm_slots.size();
    ICP_HIP(ctx, ctx->zbuf.reserve((size_t)npix * sizeof(unsigned long long)));
    unsigned long long* z = ctx->zbuf.as<unsigned long long>();
    ctx->zbuf_clean = nullptr;  // the target keys stay behind: the next icp_project on this context must clear first
    const int blocks = (npix + PM_THREADS - 1) / PM_THREADS;
    ICP_HIP(ctx, ctx->partials.reserve((size_t)blocks * NEQ * sizeof(double)));
    hipLaunchKernelGGL(k_zclear, dim3((npix + 255) / 256), dim3(256), 0, ctx->stream, z, (long long)npix);
    if (n > 0)
        hipLaunchKernelGGL(k_pm_project_targets, dim3((n + 255) / 256), dim3(256), 0, ctx->stream,
                           ctx->tgt4.as<float4>(), n, ctx->tgt_mode, reg_state(ctx), proj_arg(ctx), z);
    const int tok = prof_begin(ctx, 0);
    hipLaunchKernelGGL(k_pm_iterate, dim3(blocks), dim3(PM_THREADS), 0, ctx->stream, ctx->tgt4.as<float4>(), npix,
                       k_maps, z, ctx->pm_mv.as<float4>(), ctx->pm_mn.as<float4>(), reg_state(ctx),
                       make_align_params(ctx), ctx->partials.as<double>());
    prof_end(ctx, tok);
    ICP_HIP(ctx, hipGetLastError());
    *blocks_out = blocks;
    return ICP_OK;
}

// association of untransformed `xyz_dev` [n,3] against the model: rows9 [HW,9] + flags [HW]
int pmap_associate(icp_ctx* ctx, const float* xyz_dev, int64_t n, float* rows9_dev, int* flags_dev) {
    const int npix = ctx->cfg.height * ctx->cfg.width;
    const int k_maps = (int)ctx->pm_slots.size();
    int rc = project_device(ctx, xyz_dev, n, nullptr, nullptr, true);  // fills ctx->zbuf with (range, ~index) keys
    if (rc) return rc;
    hipLaunchKernelGGL(k_pm_assoc, dim3((npix + 255) / 256), dim3(256), 0, ctx->stream, xyz_dev, npix, k_maps,
                       ctx->zbuf.as<unsigned long long>(), ctx->pm_mv.as<float4>(), ctx->pm_mn.as<float4>(), rows9_dev,
                       flags_dev);
    ICP_HIP(ctx, hipGetLastError());
    return ICP_OK;
}

}  // namespace icp
