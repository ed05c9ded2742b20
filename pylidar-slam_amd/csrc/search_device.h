// Device-side building blocks of the exact voxel-hash nearest-neighbour search (search.hip).
#pragma once
#include "icp_internal.h"

namespace icp {

// Exchanges inside a 4-lane group (the lanes of one query) by DPP quad permutes: one VALU instruction each, where
// __shfl_xor is an address computation + ds_bpermute + a wait for the LDS crossbar.  All four lanes must be active (the
// groups' control flow is group-uniform).  X = 1, 2, 3: the lane `sub ^ X`.
template <int X>
__device__ inline int quad_xor(int v) {
    static_assert(X >= 1 && X <= 3, "lane ^ X within a quad");
    constexpr int ctrl = X == 1 ? 0xB1 : (X == 2 ? 0x4E : 0x1B);  // quad_perm [1,0,3,2] / [2,3,0,1] / [3,2,1,0]
    return __builtin_amdgcn_update_dpp(v, v, ctrl, 0xf, 0xf, false);
}
template <int X>
__device__ inline float quad_xor(float v) { return __int_as_float(quad_xor<X>(__float_as_int(v))); }
template <int X>
__device__ inline double quad_xor(double v) {
    const long long b = __double_as_longlong(v);
    const int lo = quad_xor<X>((int)(b & 0xffffffffll)), hi = quad_xor<X>((int)(b >> 32));
    return __longlong_as_double(((long long)hi << 32) | (unsigned int)lo);
}
// the value of lane L (0..3) of the group, in all four lanes
template <int L>
__device__ inline int quad_bcast(int v) {
    static_assert(L >= 0 && L <= 3, "lane of the quad");
    return __builtin_amdgcn_update_dpp(v, v, L * 0x55, 0xf, 0xf, false);
}
__device__ inline float quad_min(float v) {
    v = fminf(v, quad_xor<1>(v));
    return fminf(v, quad_xor<2>(v));
}

// Exchanges inside a ROW of 16 lanes (the lanes of one straggler of the kNN normals, search.hip::k_normals_tail16) by DPP:
// STEP 0 / 1 = the lane ^ 1 / ^ 2 (quad permutes), STEP 2 = the mirror of each half row (lane i <-> 7 - i: the other quad of
// the half), STEP 3 = the mirror of the row (i <-> 15 - i: the other half).  After the four steps of a minimum — or of an
// order-independent sum — every lane of the row holds the row's result; one VALU instruction per 32-bit word and step, where
// __shfl_xor pays an address computation, a ds_bpermute and the wait for the LDS crossbar.  All sixteen lanes must be active.
template <int STEP>
__device__ inline int row16_step(int v) {
    static_assert(STEP >= 0 && STEP <= 3, "four steps cover a row of 16");
    constexpr int ctrl = STEP == 0 ? 0xB1 : (STEP == 1 ? 0x4E : (STEP == 2 ? 0x141 : 0x140));  // quad_perm x2, row_half_mirror, row_mirror
    return __builtin_amdgcn_update_dpp(v, v, ctrl, 0xf, 0xf, false);
}
template <int STEP>
__device__ inline unsigned long long row16_step(unsigned long long v) {
    const unsigned lo = (unsigned)row16_step<STEP>((int)(unsigned)(v & 0xffffffffull));
    const unsigned hi = (unsigned)row16_step<STEP>((int)(unsigned)(v >> 32));
    return ((unsigned long long)hi << 32) | lo;
}
template <int STEP>
__device__ inline double row16_step(double v) {
    return __longlong_as_double((long long)row16_step<STEP>((unsigned long long)__double_as_longlong(v)));
}
__device__ inline double row16_sum(double v) {  // (sums of multiples of 2^-40 below 2^13: exact in any order — CovSums)
    v += row16_step<0>(v);
    v += row16_step<1>(v);
    v += row16_step<2>(v);
    return v + row16_step<3>(v);
}

struct Best {
    float d2;
    int idx;       // original index (tie-break)
    int pos;       // cell-sorted position
    float second;  // lower bound on the squared distance of every OTHER map point seen or pruned so far (NN cache)
};

__device__ inline bool better(float d2, int idx, float bd2, int bidx) { return d2 < bd2 || (d2 == bd2 && idx < bidx); }

// What a 4-lane search of the row path knows at its end: the nearest map point (b: distance, original index, position),
// up to two more candidates by position (-1: none) and, in b.second, a lower bound on the squared distance of every map
// point that is NOT one of those — the NN cache keeps the set and settles later iterations by comparing its members.
struct Near3 {
    Best b;
    int pos1, pos2;
};

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

// `skip` (optional): an original index this lane must not take — the seed of a 4-lane search belongs to lane 0 alone
// (the lanes keep local bests until the end: the same point in two of them would waste a place of the candidate set)
__device__ inline void consider(const float4 q, int pos, float px, float py, float pz, Best& b, int skip = -2) {
    const float dx = q.x - px, dy = q.y - py, dz = q.z - pz;
    const float d2 = fmaf(dz, dz, fmaf(dy, dy, dx * dx));
    const int idx = __float_as_int(q.w);
    if (idx == b.idx || idx == skip) return;  // the current best again (clamped tail of a 4-wide fetch, a seed met in its
                                              // cell, or the same point seen through the coarse level): original indices
                                              // are unique
    if (better(d2, idx, b.d2, b.idx)) {
        b.second = fminf(b.second, b.d2);
        b.d2 = d2;
        b.idx = idx;
        b.pos = pos;
    } else {
        b.second = fminf(b.second, d2);
    }
}

// candidates are fetched four at a time (independent 16-byte loads in flight together); the tail re-reads the last
// point of the cell, which cannot change the (d2, index) minimum
__device__ inline void scan_cell_1nn(const GridView& g, int start, int count, float px, float py, float pz, Best& b,
                                     int skip = -2) {
    const int last = start + count - 1;
    for (int k = start; k <= last; k += 4) {
        const int k1 = min(k + 1, last), k2 = min(k + 2, last), k3 = min(k + 3, last);
        const float4 q0 = g.pts[k], q1 = g.pts[k1], q2 = g.pts[k2], q3 = g.pts[k3];
        consider(q0, k, px, py, pz, b, skip);
        consider(q1, k1, px, py, pz, b, skip);
        consider(q2, k2, px, py, pz, b, skip);
        consider(q3, k3, px, py, pz, b, skip);
    }
}

// ring search on ONE level; returns true when the result is provably exact within `max_rings`
__device__ inline bool nearest_in_level(const GridView& g, float px, float py, float pz, int max_rings, Best& b) {
    b.d2 = INFINITY;
    b.idx = 0x7fffffff;
    b.pos = -1;
    b.second = 0.f;  // the generic path does not feed the NN cache
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
        if (b.d2 <= bound * bound * 0.999999f) return true;
    }
    return false;
}

// view of the coarse level through the same struct (only the fields the generic search reads)
__device__ inline GridView coarse_view(const GridView& g) {
    GridView c = g;
    c.table = g.ctable;
    c.mask = g.cmask;
    c.h = g.ch;
    c.inv_h = g.cinv_h;
    c.pts = g.cpts;
    return c;
}

// the tail of `nearest_in_grid` for a caller that has already exhausted the fine rings
__device__ inline Best nearest_beyond_fine(const GridView& g, float px, float py, float pz) {
    Best b;
    if (g.dbg) atomicAdd(&g.dbg[3], 1);
    if (g.ctable) {
        const GridView c = coarse_view(g);
        if (nearest_in_level(c, px, py, pz, COARSE_RINGS, b)) {
            b.pos = g.pos_of_orig[b.idx];
            return b;
        }
    }
    if (g.dbg) atomicAdd(&g.dbg[4], 1);
    b.d2 = INFINITY;
    b.idx = 0x7fffffff;
    b.pos = -1;
    b.second = 0.f;
    scan_cell_1nn(g, 0, g.m, px, py, pz, b);
    b.second = 0.f;
    return b;
}

// exact NN for any input: fine rings, then coarse rings, then (queries farther than COARSE_RINGS coarse cells from
// every map point) the exhaustive scan
__device__ inline Best nearest_in_grid(const GridView& g, float px, float py, float pz, int max_rings) {
    Best b;
    if (nearest_in_level(g, px, py, pz, max_rings, b)) return b;
    if (g.dbg) atomicAdd(&g.dbg[3], 1);
    if (g.ctable) {
        const GridView c = coarse_view(g);
        if (nearest_in_level(c, px, py, pz, COARSE_RINGS, b)) {
            b.pos = g.pos_of_orig[b.idx];
            return b;
        }
    }
    if (g.dbg) atomicAdd(&g.dbg[4], 1);
    b.d2 = INFINITY;
    b.idx = 0x7fffffff;
    b.pos = -1;
    b.second = 0.f;
    scan_cell_1nn(g, 0, g.m, px, py, pz, b);
    b.second = 0.f;
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

}  // namespace icp
