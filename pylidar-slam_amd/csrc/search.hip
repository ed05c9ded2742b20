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
#include <type_traits>

#include "gn_device.h"
#include "icp_internal.h"
#include "search_device.h"
#include "solve_device.h"

namespace icp {

// ---------------------------------------------------------------------------------------------------------------------
// K1: transform + exact 1-NN, neighbour rows, 4 lanes per query.
//
// What the hardware counters said about a one-lane-per-query ring search (round 1): L1 hit rate
// > 90 %, HBM traffic = the compulsory bytes, yet 160 dependent load instructions and ~3000 VALU instructions per wave:
// the 27-cell loop runs once per cell for the union of the lanes, each pass = hash + dependent probe + dependent
// candidate rounds.  So:
//   * one hash probe per query (its own cell); the 27 neighbour (start, count) pairs come from the cell's ROW
//     (contiguous, loaded in one round, no hashing);
//   * FOUR lanes per query: own-cell candidates strided over the lanes, the 26 neighbours split 7/6/7/6, two
//     shuffle min-reductions.  4x the waves in flight (32 per CU instead of 8) and a 4x shorter dependent chain.
// A query whose own cell is empty (no row) splits the 26 hashed probes over its 4 lanes instead.  Anything not provably
// exact after ring 1 continues with rings 2.. / the coarse level / the exhaustive scan, split over the same 4 lanes.
// ---------------------------------------------------------------------------------------------------------------------
// e-th (0..47) shell cell of the three middle z-slabs of the 5x5x5 block: each slab contributes its 16 border cells
__device__ inline int shell_mid(int e) {
    const int slab = e >> 4, j = e & 15;  // slab 0..2 -> z index 1..3
    // border of a 5x5 square in row-major order: row 0 (5), rows 1-3 (2 each), row 4 (5)
    int cell;
    if (j < 5) cell = j;
    else if (j < 11) cell = 5 * (1 + ((j - 5) >> 1)) + (((j - 5) & 1) ? 4 : 0);
    else cell = 20 + (j - 11);
    return 25 * (slab + 1) + cell;
}

template <int X>
__device__ inline void group_min_step(Best& b) {
    const float d2 = quad_xor<X>(b.d2);
    const int idx = quad_xor<X>(b.idx);
    const int pos = quad_xor<X>(b.pos);
    float sec = fminf(b.second, quad_xor<X>(b.second));
    if (idx != b.idx) sec = fminf(sec, fmaxf(d2, b.d2));  // the loser of two distinct bests is an "other" point
    if (better(d2, idx, b.d2, b.idx)) {
        b.d2 = d2;
        b.idx = idx;
        b.pos = pos;
    }
    b.second = sec;
}
__device__ inline void group_min4(Best& b) {
    group_min_step<1>(b);
    group_min_step<2>(b);
}

__device__ inline void scan_strided4(const GridView& g, int start, int count, int sub, float px, float py, float pz,
                                     Best& b, int skip = -2) {
    // 4 independent loads per lane and round (16 candidates per group): a cell of ~10 points is one round
    const int last = start + count - 1;
    for (int k = start + sub; k <= last; k += 16) {
        const int k1 = min(k + 4, last), k2 = min(k + 8, last), k3 = min(k + 12, last);
        const float4 q0 = g.pts[k], q1 = g.pts[k1], q2 = g.pts[k2], q3 = g.pts[k3];
        consider(q0, k, px, py, pz, b, skip);
        consider(q1, k1, px, py, pz, b, skip);
        consider(q2, k2, px, py, pz, b, skip);
        consider(q3, k3, px, py, pz, b, skip);
    }
}

// Rings r_begin..r_end of ONE level searched by the 4 lanes of a query (hashed probes; ring 0 = the query's own cell with
// its candidates strided over the lanes, ring r >= 1 = the shell split over the lanes), pruned by box distance against
// the best so far, one group reduction per ring.  Returns true as soon as the best is provably exact on this level.
__device__ inline bool coop_rings(const GridView& lv, float px, float py, float pz, int sub, int r_begin, int r_end,
                                  Best& b, int2* __restrict__ stack, int stride) {
    const int cx = cell_coord(px, lv.inv_h), cy = cell_coord(py, lv.inv_h), cz = cell_coord(pz, lv.inv_h);
    const float h = lv.h;
    const float fx = fminf(fmaxf(px - (float)cx * h, 0.f), h);
    const float fy = fminf(fmaxf(py - (float)cy * h, 0.f), h);
    const float fz = fminf(fmaxf(pz - (float)cz * h, 0.f), h);
    const float edge = fminf(fminf(fminf(fx, h - fx), fminf(fy, h - fy)), fminf(fz, h - fz));
    for (int r = r_begin; r <= r_end; ++r) {
        int start, count;
        if (r == 0) {
            if (grid_lookup(lv, cx, cy, cz, start, count)) scan_strided4(lv, start, count, sub, px, py, pz, b);
        } else {
            // the shell's cells are PROBED one lane each, seven per lane at a time, and SCANNED by the four lanes together
            // (16 candidates per round): a coarse cell of several hundred points used to be one lane's to walk alone
            const int side = 2 * r + 1, total = side * side * side;
            for (int c0 = 0; c0 < total; c0 += 28) {  // group-uniform
                int nl = 0;
                for (int k = 0; k < 7; ++k) {
                    const int c = c0 + sub + 4 * k;
                    if (c >= total) break;
                    const int ox = c % side - r, oy = (c / side) % side - r, oz = c / (side * side) - r;
                    const int m = max(max(ox < 0 ? -ox : ox, oy < 0 ? -oy : oy), oz < 0 ? -oz : oz);
                    if (m < r) continue;  // interior: visited by the previous rings
                    const float gx = axis_gap(ox, fx, h), gy = axis_gap(oy, fy, h), gz = axis_gap(oz, fz, h);
                    const float gap2 = fmaf(gx, gx, fmaf(gy, gy, gz * gz));
                    if (gap2 > b.d2) {
                        b.second = fminf(b.second, gap2);  // every point of a pruned cell is at least that far
                        continue;
                    }
                    if (grid_lookup(lv, cx + ox, cy + oy, cz + oz, start, count)) {
                        stack[nl * stride] = make_int2(start, count);
                        ++nl;
                    }
                }
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
                const int n_of[4] = {quad_bcast<0>(nl), quad_bcast<1>(nl), quad_bcast<2>(nl), quad_bcast<3>(nl)};
#pragma unroll
                for (int l = 0; l < 4; ++l) {
                    const int n_l = n_of[l];  // group-uniform
                    for (int k = 0; k < n_l; ++k) {
                        const int2 e = stack[k * stride + (l - sub)];  // lane l's column sits next to this lane's
                        scan_strided4(lv, e.x, e.y, sub, px, py, pz, b);
                    }
                }
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();  // the next chunk overwrites the lists
            }
        }
        group_min4(b);
        const float bound = (float)r * h + edge;
        if (b.d2 <= bound * bound * 0.999999f) {
            b.second = fminf(b.second, bound * bound * 0.999999f);  // nothing outside the visited block is closer
            return true;
        }
    }
    return false;
}

// the search of one query by its 4 lanes; the result is valid in every lane of the group (b is shared after the
// reductions) except when the per-lane fallback ran, which only lane 0 executes (and only lane 0 consumes)
// `stack` = this lane's column of an LDS array [7][blockDim] (stride = blockDim): the neighbour cells it still has to
// scan.  Scanning them from a per-lane list lets every lane walk ITS cells back to back; looping over the 7 row entries
// in lockstep instead makes the whole wave pay one pass (probe of the predicate + 4 loads + wait) per entry.
// `seed_*` (optional, seed_pos < 0: none): a map point already known to be a candidate — the neighbour cached by the
// previous iteration.  It only tightens the starting upper bound (neighbour cells farther than it are pruned before
// they are read); the (distance, index) minimum over the map is the same with or without it.
__device__ inline Near3 search_rows_group(const GridView& g, float px, float py, float pz, int sub, int max_rings,
                                          int2* __restrict__ stack, int stride, float seed_d2 = INFINITY,
                                          int seed_idx = 0x7fffffff, int seed_pos = -1) {
    // Through ring 1 every lane keeps its LOCAL best (and the bound of the other points it has seen or pruned): only the
    // pruning radius r2 — the best squared distance any lane has found — is shared.  What the four lanes hold at the end
    // is the candidate set of Near3; a merge after every step would throw three of the four away.
    Near3 out;
    out.pos1 = out.pos2 = -1;
    Best& b = out.b;
    const int skip = sub == 0 ? -2 : seed_idx;  // the seed is lane 0's
    b.d2 = sub == 0 ? seed_d2 : INFINITY;
    b.idx = sub == 0 ? seed_idx : 0x7fffffff;
    b.pos = sub == 0 ? seed_pos : -1;
    b.second = INFINITY;
    float r2 = seed_d2;
    const int cx = cell_coord(px, g.inv_h), cy = cell_coord(py, g.inv_h), cz = cell_coord(pz, g.inv_h);
    const float h = g.h;
    const float fx = fminf(fmaxf(px - (float)cx * h, 0.f), h);
    const float fy = fminf(fmaxf(py - (float)cy * h, 0.f), h);
    const float fz = fminf(fmaxf(pz - (float)cz * h, 0.f), h);
    const float edge = fminf(fminf(fminf(fx, h - fx), fminf(fy, h - fy)), fminf(fz, h - fz));
    // own cell: the entry of the first slot and — rows are indexed by slot — this lane's 7 row entries are fetched
    // together (a first-probe hit is the common case; the speculative row is simply reloaded after a collision)
    const unsigned long long key = pack_cell(cx, cy, cz);
    unsigned int slot = hash_cell(key) & g.mask;
    GridEntry e = g.table[slot];
    int2 cell[7];
    {
        const int2* __restrict__ r = g.rows + (size_t)slot * ROW_STRIDE;
#pragma unroll
        for (int k = 0; k < 7; ++k) cell[k] = r[sub * 7 + k];
    }
    if (e.key != key && e.key != GRID_EMPTY) {
        while (true) {
            slot = (slot + 1) & g.mask;
            e = g.table[slot];
            if (e.key == key || e.key == GRID_EMPTY) break;
        }
        const int2* __restrict__ r = g.rows + (size_t)slot * ROW_STRIDE;
#pragma unroll
        for (int k = 0; k < 7; ++k) cell[k] = r[sub * 7 + k];
    }
    if (e.key == key) {
        // ---- row path
        scan_strided4(g, e.start, e.count, sub, px, py, pz, b, skip);
        r2 = fminf(r2, quad_min(b.d2));
        // neighbour cells are pruned when their box is farther than the best so far by `prune_guard` (see search_rows_wave)
        const float prune_r = sqrtf(r2) + g.prune_guard, prune2 = prune_r * prune_r;
        int nl = 0;
#pragma unroll
        for (int k = 0; k < 7; ++k) {
            const int c = sub * 7 + k;
            if (c == 13 || c >= 27 || cell[k].y <= 0) continue;  // own cell done above; entry 27 is padding
            const float gx = axis_gap(c % 3 - 1, fx, h), gy = axis_gap((c / 3) % 3 - 1, fy, h),
                        gz = axis_gap(c / 9 - 1, fz, h);
            const float gap2 = fmaf(gx, gx, fmaf(gy, gy, gz * gz));
            if (gap2 > prune2) {
                b.second = fminf(b.second, gap2);  // every point of a pruned cell is at least that far
                continue;
            }
            stack[nl * stride] = make_int2(cell[k].x, cell[k].y | (c << 24));
            ++nl;
        }
        if (g.flat_rows == 2) {
            // every surviving cell of the group scanned by its four lanes together, 16 candidates per round and cell: a
            // round more than the flattened list when several small cells survive, a third of its instructions
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            const int n_of[4] = {quad_bcast<0>(nl), quad_bcast<1>(nl), quad_bcast<2>(nl), quad_bcast<3>(nl)};
#pragma unroll
            for (int l = 0; l < 4; ++l) {
                for (int k = 0; k < n_of[l]; ++k) {  // group-uniform
                    const int2 e = stack[k * stride + (l - sub)];  // lane l's column sits next to this lane's
                    scan_strided4(g, e.x, e.y & 0xffffff, sub, px, py, pz, b, skip);
                }
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            nl = 0;
        } else if (g.flat_rows) {
            // The surviving neighbour cells of the GROUP laid end to end (round 3): every lane's cells are announced in a
            // list the four lanes share (their four LDS columns, 28 slots), with the running candidate count in front of
            // each; the four lanes then walk candidates 0 .. T-1 together, four loads in flight each, finding the cell of
            // candidate j by a 5-step search of the list.  ~25 surviving candidates are two rounds of loads instead of one
            // round per (cell, 4 candidates) of whichever lane drew the most cells.
            int tl = 0;
#pragma unroll
            for (int i = 0; i < 7; ++i)
                if (i < nl) tl += stack[i * stride].y & 0xffffff;
            int cbase = 0, tbase = 0, C = 0, T = 0;
            const int n_of[4] = {quad_bcast<0>(nl), quad_bcast<1>(nl), quad_bcast<2>(nl), quad_bcast<3>(nl)};
            const int t_of[4] = {quad_bcast<0>(tl), quad_bcast<1>(tl), quad_bcast<2>(tl), quad_bcast<3>(tl)};
#pragma unroll
            for (int l = 0; l < 4; ++l) {
                const int n_l = n_of[l], t_l = t_of[l];
                if (l < sub) {
                    cbase += n_l;
                    tbase += t_l;
                }
                C += n_l;
                T += t_l;
            }
            int2 mine[7];
#pragma unroll
            for (int i = 0; i < 7; ++i) mine[i] = i < nl ? stack[i * stride] : make_int2(0, 0);
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();  // (every lane has read its own column: the columns now hold the shared list)
            {
                int cum = tbase;
#pragma unroll
                for (int i = 0; i < 7; ++i)
                    if (i < nl) {
                        const int e = cbase + i;
                        stack[(e >> 2) * stride + ((e & 3) - sub)] = make_int2(mine[i].x - cum, cum);
                        cum += mine[i].y & 0xffffff;
                    }
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            for (int j0 = 0; j0 < T; j0 += 16) {  // group-uniform
                int pos[4];
#pragma unroll
                for (int m = 0; m < 4; ++m) {
                    const int j = min(j0 + 4 * m + sub, T - 1);  // (a clamped tail re-reads the last candidate: harmless)
                    int e = 0;
#pragma unroll
                    for (int s2 = 16; s2 > 0; s2 >>= 1) {
                        const int t = e + s2;
                        if (t < C && stack[(t >> 2) * stride + ((t & 3) - sub)].y <= j) e = t;
                    }
                    pos[m] = j + stack[(e >> 2) * stride + ((e & 3) - sub)].x;
                }
                const float4 q0 = g.pts[pos[0]], q1 = g.pts[pos[1]], q2 = g.pts[pos[2]], q3 = g.pts[pos[3]];
                consider(q0, pos[0], px, py, pz, b, skip);
                consider(q1, pos[1], px, py, pz, b, skip);
                consider(q2, pos[2], px, py, pz, b, skip);
                consider(q3, pos[3], px, py, pz, b, skip);
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();  // (the columns go back to their lanes)
            nl = 0;
        }
        int ci = 0, st = 0, cnt = 0, k = 0;
        for (;;) {
            if (k >= cnt) {
                if (ci >= nl) break;
                const int2 nx = stack[ci * stride];
                ++ci;
                const int c = (int)((unsigned)nx.y >> 24);
                // the best may have shrunk since the cell was queued
                const float gx = axis_gap(c % 3 - 1, fx, h), gy = axis_gap((c / 3) % 3 - 1, fy, h),
                            gz = axis_gap(c / 9 - 1, fz, h);
                const float gap2 = fmaf(gx, gx, fmaf(gy, gy, gz * gz));
                if (gap2 > prune2) {
                    b.second = fminf(b.second, gap2);
                    continue;
                }
                st = nx.x;
                cnt = nx.y & 0xffffff;
                k = 0;
            }
            const int last = st + cnt - 1, k0 = st + k;
            const int k1 = min(k0 + 1, last), k2 = min(k0 + 2, last), k3 = min(k0 + 3, last);
            const float4 q0 = g.pts[k0], q1 = g.pts[k1], q2 = g.pts[k2], q3 = g.pts[k3];
            consider(q0, k0, px, py, pz, b, skip);
            consider(q1, k1, px, py, pz, b, skip);
            consider(q2, k2, px, py, pz, b, skip);
            consider(q3, k3, px, py, pz, b, skip);
            k += 4;
        }
        r2 = fminf(r2, quad_min(b.d2));
    } else {
        // ---- own cell empty: hashed probes of the 26 neighbours, split over the 4 lanes
        for (int c0 = sub; c0 < 26; c0 += 4) {
            const int c = c0 + (c0 >= 13 ? 1 : 0);
            const int ox = c % 3 - 1, oy = (c / 3) % 3 - 1, oz = c / 9 - 1;
            const float gx = axis_gap(ox, fx, h), gy = axis_gap(oy, fy, h), gz = axis_gap(oz, fz, h);
            const float gap2 = fmaf(gx, gx, fmaf(gy, gy, gz * gz));
            if (gap2 > fminf(r2, b.d2)) {
                b.second = fminf(b.second, gap2);
                continue;
            }
            int start, count;
            if (grid_lookup(g, cx + ox, cy + oy, cz + oz, start, count))
                scan_cell_1nn(g, start, count, px, py, pz, b, skip);
        }
        r2 = fminf(r2, quad_min(b.d2));
    }
    float bound = h + edge;
    bool resolved = r2 <= bound * bound * 0.999999f;  // group-uniform: r2 is shared
    if (g.dbg && sub == 0) atomicAdd(&g.dbg[resolved ? 0 : 1], 1);
    if (g.dbg && sub == 0 && e.key != key) atomicAdd(&g.dbg[5], 1);
    if (resolved) {
        // ---- settled by ring 1: the candidate set.  Nothing outside the 27-cell block is closer than `bound`; every
        // point inside it that is not one of the four local bests is bounded by its lane's `second`.
        float L2 = fminf(quad_min(b.second), bound * bound * 0.999999f);
        {   // the same point in two lanes (a clamped tail of a 4-wide fetch): the higher lane gives it up
            const int i1 = quad_xor<1>(b.idx), i2 = quad_xor<2>(b.idx), i3 = quad_xor<3>(b.idx);
            const bool dup = (i1 == b.idx && (sub ^ 1) < sub) || (i2 == b.idx && (sub ^ 2) < sub) ||
                             (i3 == b.idx && (sub ^ 3) < sub);
            if (dup || b.pos < 0) {
                b.d2 = INFINITY;
                b.idx = 0x7fffffff;
                b.pos = -1;
            }
        }
        int rank = 0;  // place of this lane's candidate among the four, by (distance, index, lane)
        {
            float od2 = quad_xor<1>(b.d2);
            int oi = quad_xor<1>(b.idx);
            rank += (better(od2, oi, b.d2, b.idx) || (od2 == b.d2 && oi == b.idx && (sub ^ 1) < sub)) ? 1 : 0;
            od2 = quad_xor<2>(b.d2);
            oi = quad_xor<2>(b.idx);
            rank += (better(od2, oi, b.d2, b.idx) || (od2 == b.d2 && oi == b.idx && (sub ^ 2) < sub)) ? 1 : 0;
            od2 = quad_xor<3>(b.d2);
            oi = quad_xor<3>(b.idx);
            rank += (better(od2, oi, b.d2, b.idx) || (od2 == b.d2 && oi == b.idx && (sub ^ 3) < sub)) ? 1 : 0;
        }
        int p0 = rank == 0 ? b.pos : -1, p1 = rank == 1 ? b.pos : -1, p2 = rank == 2 ? b.pos : -1;
        int i0 = rank == 0 ? b.idx : -1;
        float d0 = rank == 0 ? b.d2 : INFINITY, d3 = rank == 3 ? b.d2 : INFINITY;
        p0 = max(p0, quad_xor<1>(p0)), p1 = max(p1, quad_xor<1>(p1)), p2 = max(p2, quad_xor<1>(p2));
        i0 = max(i0, quad_xor<1>(i0));
        d0 = fminf(d0, quad_xor<1>(d0)), d3 = fminf(d3, quad_xor<1>(d3));
        p0 = max(p0, quad_xor<2>(p0)), p1 = max(p1, quad_xor<2>(p1)), p2 = max(p2, quad_xor<2>(p2));
        i0 = max(i0, quad_xor<2>(i0));
        d0 = fminf(d0, quad_xor<2>(d0)), d3 = fminf(d3, quad_xor<2>(d3));
        b.d2 = d0;
        b.idx = p0 >= 0 ? i0 : 0x7fffffff;
        b.pos = p0;
        b.second = fminf(L2, d3);  // the fourth local best is one of the others
        out.pos1 = p1;
        out.pos2 = p2;
        return out;
    }
    // ---- not settled: the merged best of the group from here on (shared by the four lanes, as the rings expect)
    group_min4(b);
    {
        // rings 2..max_rings of the fine level, then the coarse level (4x cells), every ring split over the 4 lanes
        if (g.dbg && sub == 0) atomicAdd(&g.dbg[2], 1);
        resolved = max_rings >= 2 && coop_rings(g, px, py, pz, sub, 2, max_rings, b, stack, stride);
        if (!resolved && g.ctable) {
            if (g.dbg && sub == 0) atomicAdd(&g.dbg[3], 1);
            // candidates of the coarse level carry positions of ITS point array: the winner is identified by its original
            // index (`consider` recognises the best so far by index, so meeting it again changes nothing)
            resolved = coop_rings(coarse_view(g), px, py, pz, sub, 0, COARSE_RINGS, b, stack, stride);
            if (b.idx != 0x7fffffff) b.pos = g.pos_of_orig[b.idx];
        }
        if (!resolved && sub == 0) {  // farther than COARSE_RINGS coarse cells from every map point
            if (g.dbg) atomicAdd(&g.dbg[4], 1);
            b.d2 = INFINITY;
            b.idx = 0x7fffffff;
            b.pos = -1;
            b.second = INFINITY;
            scan_cell_1nn(g, 0, g.m, px, py, pz, b);  // every other point is seen: b.second = the second-nearest
        }
    }
    return out;
}

// ---- the generic exact search with W = 16 lanes per query (round 4, item 48) -------------------------------------
// For the queries the ball search hands back when a workgroup has a few dozen of them: own cell empty, a ball that leaves
// its 2x2x2 block, more candidates than `ball_max`.  A 4-lane group takes them one dependent round trip after the other
// (seven lookups per lane and chunk, a round of loads per cell: 12-35 us per query, and the launch lasts as long as the
// workgroup that has them); here a lane looks up at most seven cells AT ONCE (keys first, then the ranges: two round trips
// for a whole ring 2 at W = 16), the cells found are laid end to end in the group's LDS columns and their points taken
// 4 W per round.  Same minimum, same tie-break, same bound on every other point as the 4-lane search.
template <int W>
__device__ __forceinline__ void group_min_w(Best& b) {
#pragma unroll
    for (int o = 1; o < W; o <<= 1) {
        const float d2 = __shfl_xor(b.d2, o, W);
        const int idx = __shfl_xor(b.idx, o, W);
        const int pos = __shfl_xor(b.pos, o, W);
        float sec = fminf(b.second, __shfl_xor(b.second, o, W));
        if (idx != b.idx) sec = fminf(sec, fmaxf(d2, b.d2));  // the loser of two distinct bests is an "other" point
        if (better(d2, idx, b.d2, b.idx)) {
            b.d2 = d2;
            b.idx = idx;
            b.pos = pos;
        }
        b.second = sec;
    }
}

// seven cells of one lane looked up together: first-slot keys (one round trip), then the ranges of those that matched (a
// second); a collision walks on alone.  found[k] = (start, count), (0, 0) where there is no such cell or bit k of `want` is
// clear.  (found[k] itself holds the key while it is in flight: no registers beside the list.)
template <typename KeyOf>
__device__ __forceinline__ void grid_lookup7(const GridView& g, int want, KeyOf key_of, int2 (&found)[7]) {
#pragma unroll
    for (int k = 0; k < 7; ++k) {
        found[k] = make_int2(0, 0);
        if (want >> k & 1) {
            const unsigned long long v = g.table[hash_cell(key_of(k)) & g.mask].key;
            found[k] = make_int2((int)(unsigned int)(v & 0xffffffffull), (int)(unsigned int)(v >> 32));
        }
    }
#pragma unroll
    for (int k = 0; k < 7; ++k) {
        if (!(want >> k & 1)) continue;
        const unsigned long long key = key_of(k);
        unsigned int slot = hash_cell(key) & g.mask;
        unsigned long long got = ((unsigned long long)(unsigned int)found[k].y << 32) | (unsigned int)found[k].x;
        while (got != key && got != GRID_EMPTY) {
            slot = (slot + 1) & g.mask;
            got = g.table[slot].key;
        }
        found[k] = got == key ? *reinterpret_cast<const int2*>(&g.table[slot].start) : make_int2(0, 0);
    }
}

// the cells in found[] (count > 0) of the W lanes of a group, laid end to end in the group's 7 x W LDS entries
// (`col0` = column of the group's lane 0, rows `stride` apart) and walked together, 4 W candidates per round
template <int W>
__device__ __forceinline__ void walk_found_w(const GridView& lv, const int2 (&found)[7], int2* __restrict__ col0, int stride, int sub,
                                    float px, float py, float pz, Best& b) {
    int nl = 0, tl = 0;
#pragma unroll
    for (int k = 0; k < 7; ++k)
        if (found[k].y > 0) {
            ++nl;
            tl += found[k].y;
        }
    int cincl = nl, tincl = tl;  // inclusive prefix sums over the lanes of the group
#pragma unroll
    for (int o = 1; o < W; o <<= 1) {
        const int cn = __shfl_up(cincl, o, W), tn = __shfl_up(tincl, o, W);
        if (sub >= o) {
            cincl += cn;
            tincl += tn;
        }
    }
    const int C = __shfl(cincl, W - 1, W), T = __shfl(tincl, W - 1, W);  // group-uniform
    if (T == 0) return;
    {
        int e = cincl - nl, cum = tincl - tl;
#pragma unroll
        for (int k = 0; k < 7; ++k)
            if (found[k].y > 0) {
                col0[(e / W) * stride + (e % W)] = make_int2(found[k].x - cum, cum);  // candidate j of this cell: j + .x
                cum += found[k].y;
                ++e;
            }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    for (int j0 = 0; j0 < T; j0 += 4 * W) {  // group-uniform
        int pos[4];
#pragma unroll
        for (int m = 0; m < 4; ++m) {
            const int j = min(j0 + W * m + sub, T - 1);  // (a clamped tail re-reads the last candidate: harmless)
            int e = 0;
#pragma unroll
            for (int s2 = (7 * W > 128 ? 256 : 64); s2 > 0; s2 >>= 1) {  // (C <= 7 W: 112 at W = 16, 448 at W = 64)
                const int t = e + s2;
                if (t < C && col0[(t / W) * stride + (t % W)].y <= j) e = t;
            }
            pos[m] = j + col0[(e / W) * stride + (e % W)].x;
        }
        const float4 q0 = lv.pts[pos[0]], q1 = lv.pts[pos[1]], q2 = lv.pts[pos[2]], q3 = lv.pts[pos[3]];
        consider(q0, pos[0], px, py, pz, b);
        consider(q1, pos[1], px, py, pz, b);
        consider(q2, pos[2], px, py, pz, b);
        consider(q3, pos[3], px, py, pz, b);
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();  // (the next chunk overwrites the list)
}

// rings r_begin .. r_end of ONE level: the whole cube of radius r_begin first, then shell after shell; true as soon as the
// best (shared by the group after every ring) is provably exact on this level
template <int W>
__device__ __forceinline__ bool far_rings_w(const GridView& lv, float px, float py, float pz, int sub, int r_begin, int r_end, Best& b,
                                   int2* __restrict__ col0, int stride) {
    const int cx = cell_coord(px, lv.inv_h), cy = cell_coord(py, lv.inv_h), cz = cell_coord(pz, lv.inv_h);
    const float h = lv.h;
    const float fx = fminf(fmaxf(px - (float)cx * h, 0.f), h);
    const float fy = fminf(fmaxf(py - (float)cy * h, 0.f), h);
    const float fz = fminf(fmaxf(pz - (float)cz * h, 0.f), h);
    const float edge = fminf(fminf(fminf(fx, h - fx), fminf(fy, h - fy)), fminf(fz, h - fz));
    for (int r = r_begin; r <= r_end; ++r) {
        const int side = 2 * r + 1, total = side * side * side;
        for (int c0 = 0; c0 < total; c0 += 7 * W) {  // group-uniform
            int want = 0;
#pragma unroll
            for (int k = 0; k < 7; ++k) {
                const int c = c0 + sub + W * k;
                if (c >= total) continue;
                const int ox = c % side - r, oy = (c / side) % side - r, oz = c / (side * side) - r;
                const int m = max(max(ox < 0 ? -ox : ox, oy < 0 ? -oy : oy), oz < 0 ? -oz : oz);
                if (m < r && r > r_begin) continue;  // interior: visited by the previous rings
                const float gx = axis_gap(ox, fx, h), gy = axis_gap(oy, fy, h), gz = axis_gap(oz, fz, h);
                const float gap2 = fmaf(gx, gx, fmaf(gy, gy, gz * gz));
                if (gap2 > b.d2) {
                    b.second = fminf(b.second, gap2);  // every point of a pruned cell is at least that far
                    continue;
                }
                want |= 1 << k;
            }
            int2 found[7];
            grid_lookup7(lv, want, [&](int k) {
                const int c = c0 + sub + W * k;
                return pack_cell(cx + c % side - r, cy + (c / side) % side - r, cz + c / (side * side) - r);
            }, found);
            walk_found_w<W>(lv, found, col0, stride, sub, px, py, pz, b);
        }
        group_min_w<W>(b);
        const float bound = (float)r * h + edge;
        if (b.d2 <= bound * bound * 0.999999f) {
            b.second = fminf(b.second, bound * bound * 0.999999f);  // nothing outside the visited block is closer
            return true;
        }
    }
    return false;
}

// one query by its W lanes: fine rings 1 .. max_rings, the coarse level, the exhaustive scan.  The result is the same in
// every lane of the group: nearest point (distance, original index, position) and, in .second, a lower bound on the
// squared distance of every other map point.  `seed_*`: a map point already known (the cached neighbour), lane 0's.
template <int W>
__device__ __forceinline__ Best search_far_w(const GridView& g, float px, float py, float pz, int sub, int max_rings,
                                    int2* __restrict__ col0, int stride, float seed_d2, int seed_idx, int seed_pos) {
    Best b;
    b.d2 = sub == 0 ? seed_d2 : INFINITY;
    b.idx = sub == 0 ? seed_idx : 0x7fffffff;
    b.pos = sub == 0 ? seed_pos : -1;
    b.second = INFINITY;
    if (far_rings_w<W>(g, px, py, pz, sub, 1, max(max_rings, 1), b, col0, stride)) return b;
    if (g.ctable) {
        // candidates of the coarse level carry positions of ITS point array: the winner is identified by its original index
        const bool ok = far_rings_w<W>(coarse_view(g), px, py, pz, sub, 0, COARSE_RINGS, b, col0, stride);
        if (b.idx != 0x7fffffff) b.pos = g.pos_of_orig[b.idx];
        if (ok) return b;
    }
    // farther than COARSE_RINGS coarse cells from every map point: every point is seen, .second = the second nearest
    b.d2 = INFINITY;
    b.idx = 0x7fffffff;
    b.pos = -1;
    b.second = INFINITY;
    for (int k = sub; k < g.m; k += W) consider(g.pts[k], k, px, py, pz, b);
    group_min_w<W>(b);
    return b;
}

// one query by its W lanes from the COARSE level on (rings 0 .. COARSE_RINGS of the 4x cells — every map point is in
// them, so the level is exact by itself — then the exhaustive scan): for a query the fine rings have not settled.  Round 5:
// a whole wave takes it (W = 64) where a workgroup has a handful of misses; the 4-lane group such a query used to fall
// back to walked 27+ hashed probes and several hundred candidates of the coarse cells in ~100 us — the duration of the
// launch, in every iteration of a frame with a dozen targets a metre away from the map (the published-configuration loop:
// frames of 0.70 ms among frames of 0.33).
template <int W>
__device__ __forceinline__ Best search_coarse_w(const GridView& g, float px, float py, float pz, int sub,
                                                int2* __restrict__ col0, int stride, float seed_d2, int seed_idx,
                                                int seed_pos) {
    Best b;
    b.d2 = sub == 0 ? seed_d2 : INFINITY;
    b.idx = sub == 0 ? seed_idx : 0x7fffffff;
    b.pos = sub == 0 ? seed_pos : -1;
    b.second = INFINITY;
    if (g.ctable) {
        // candidates of the coarse level carry positions of ITS point array: the winner is identified by its original index
        const bool ok = far_rings_w<W>(coarse_view(g), px, py, pz, sub, 0, COARSE_RINGS, b, col0, stride);
        if (b.idx != 0x7fffffff) b.pos = g.pos_of_orig[b.idx];
        if (ok) return b;
    }
    b.d2 = INFINITY;
    b.idx = 0x7fffffff;
    b.pos = -1;
    b.second = INFINITY;
    for (int k = sub; k < g.m; k += W) consider(g.pts[k], k, px, py, pz, b);
    group_min_w<W>(b);
    return b;
}

// (round 6: the four steps inside a row of 16 lanes by DPP — lane ^ 1, ^ 2, the mirror of the half row, the mirror of the row:
// every step pairs two disjoint groups of lanes, which is all the merge needs — and only the two steps across the rows through
// the LDS crossbar: 8 ds_bpermute round trips instead of 24 per call, three calls per whole-wave search.  All 64 lanes active.)
template <int STEP>
__device__ __forceinline__ void wave_min64_step(Best& b) {
    float d2, second;
    int idx, pos;
    if constexpr (STEP < 4) {
        d2 = __int_as_float(row16_step<STEP>(__float_as_int(b.d2)));
        idx = row16_step<STEP>(b.idx);
        pos = row16_step<STEP>(b.pos);
        second = __int_as_float(row16_step<STEP>(__float_as_int(b.second)));
    } else {
        constexpr int o = STEP == 4 ? 16 : 32;
        d2 = __shfl_xor(b.d2, o, 64);
        idx = __shfl_xor(b.idx, o, 64);
        pos = __shfl_xor(b.pos, o, 64);
        second = __shfl_xor(b.second, o, 64);
    }
    float sec = fminf(b.second, second);
    if (idx != b.idx) sec = fminf(sec, fmaxf(d2, b.d2));  // the loser of two distinct bests is an "other" point
    if (better(d2, idx, b.d2, b.idx)) {
        b.d2 = d2;
        b.idx = idx;
        b.pos = pos;
    }
    b.second = sec;
}
__device__ inline void wave_min64(Best& b) {
    wave_min64_step<0>(b);
    wave_min64_step<1>(b);
    wave_min64_step<2>(b);
    wave_min64_step<3>(b);
    wave_min64_step<4>(b);
    wave_min64_step<5>(b);
}

// One query searched by a WHOLE WAVE (the fused iteration kernel uses it for workgroups with few cache misses: in the
// late iterations a handful of queries search at all, and the slowest of them sets the duration of the launch — a
// 4-lane search is a chain of ~20 dependent memory round trips, 12-13 us by the in-kernel timers).  Here the chain is
// three: (1) table entry + the 27 row entries, one per lane; (2) ALL surviving candidates at once — the cells that pass
// the box test against the seed are laid end to end (prefix sum over the lanes, cell of candidate j found by a 5-step
// search in LDS) and dealt out 2 x 64 per round; (3) the winner's normal.  Same minimum, same tie-break.
// An empty own cell costs one hashed probe per lane instead of the row; a search that ring 1 does not settle goes on to
// the 98 cells of ring 2, at most two hashed probes per lane.
// `wl` = 64 ints of LDS private to the wave.  Returns false (b undefined) when ring 2 does not settle the search either:
// the caller hands the query to the generic 4-lane path (coarse level, exhaustive scan).
__device__ inline bool search_rows_wave(const GridView& g, float px, float py, float pz, int lane, int max_rings,
                                        int* __restrict__ wl, float seed_d2, int seed_idx, int seed_pos, Best& b,
                                        Best* runner = nullptr, Best* third = nullptr) {
    // `runner` (optional): the SECOND nearest map point, with runner->second = a lower bound on the squared distance of
    // every point other than the two — what lets the NN cache settle a query that sits between two map points by
    // comparing the pair instead of searching again (pos = -1 when the search cannot name it cheaply: more than 128
    // candidates, or not settled by ring 1)
    // `third` (with `runner`): the third nearest, third->second bounding every point other than the three
    if (runner) runner->pos = -1;
    if (third) third->pos = -1;
    const int cx = cell_coord(px, g.inv_h), cy = cell_coord(py, g.inv_h), cz = cell_coord(pz, g.inv_h);
    const float h = g.h;
    const float fx = fminf(fmaxf(px - (float)cx * h, 0.f), h);
    const float fy = fminf(fmaxf(py - (float)cy * h, 0.f), h);
    const float fz = fminf(fmaxf(pz - (float)cz * h, 0.f), h);
    const float edge = fminf(fminf(fminf(fx, h - fx), fminf(fy, h - fy)), fminf(fz, h - fz));
    const unsigned long long key = pack_cell(cx, cy, cz);
    unsigned int slot = hash_cell(key) & g.mask;
    const int c = min(lane, 27);  // entry 27 is the (0, 0) padding
    GridEntry e = g.table[slot];
    int2 rc = g.rows[(size_t)slot * ROW_STRIDE + c];
    if (e.key != key && e.key != GRID_EMPTY) {  // wave-uniform
        while (true) {
            slot = (slot + 1) & g.mask;
            e = g.table[slot];
            if (e.key == key || e.key == GRID_EMPTY) break;
        }
        rc = g.rows[(size_t)slot * ROW_STRIDE + c];
    }
    if (e.key != key) {
        // own cell empty (no row): the 26 neighbours by hashed probes, one per lane
        rc = make_int2(0, 0);
        if (lane < 27 && lane != 13) {
            int start, count;
            if (grid_lookup(g, cx + c % 3 - 1, cy + (c / 3) % 3 - 1, cz + c / 9 - 1, start, count))
                rc = make_int2(start, count);
        }
    }
    b.d2 = seed_d2;
    b.idx = seed_idx;
    b.pos = seed_pos;
    b.second = INFINITY;
    const float prune_r = sqrtf(seed_d2) + g.prune_guard, prune2 = prune_r * prune_r;
    int cnt = 0;
    float pruned_gap = INFINITY;  // this lane's pruned cell (if any): all its points are at least that far
    if (lane < 27) {
        const float gx = axis_gap(c % 3 - 1, fx, h), gy = axis_gap((c / 3) % 3 - 1, fy, h), gz = axis_gap(c / 9 - 1, fz, h);
        const float gap2 = fmaf(gx, gx, fmaf(gy, gy, gz * gz));
        cnt = rc.y;
        // (pruned only when the box is farther than the seed by `prune_guard`: the gap of a pruned cell bounds L, and a
        // ball that all but touches a cell face would leave the entry a slack of micrometres — such queries, ~20 of a
        // 131 072-point scan, missed in every late launch; inside the guard band the cell is scanned and its POINTS bound L)
        if (cnt > 0 && gap2 > prune2) {
            b.second = gap2;  // every point of a pruned cell is at least that far
            pruned_gap = gap2;
            cnt = 0;
        }
    }
    int incl = cnt;
#pragma unroll
    for (int o = 1; o <= 16; o <<= 1) {
        const int t = __shfl_up(incl, o, 64);
        if (lane >= o) incl += t;
    }
    const int total = __shfl(incl, 31, 64);  // lanes 27..31 add nothing
    const int excl = incl - cnt;
    if (lane < 32) {
        wl[lane] = excl;
        wl[32 + lane] = rc.x - excl;  // candidate j of cell c sits at position j + wl[32 + c]
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    float4 qa = make_float4(0.f, 0.f, 0.f, 0.f), qb = qa;  // (the last trip's candidates stay around for the runner-up)
    int pa = -1, pb = -1;
    for (int j0 = 0; j0 < total; j0 += 128) {
        const int ja = j0 + lane, jb = j0 + 64 + lane;
        int ca = 0, cb = 0;
#pragma unroll
        for (int s2 = 16; s2 > 0; s2 >>= 1) {
            if (wl[ca + s2] <= ja) ca += s2;
            if (wl[cb + s2] <= jb) cb += s2;
        }
        pa = ja < total ? ja + wl[32 + ca] : -1;
        pb = jb < total ? jb + wl[32 + cb] : -1;
        if (pa >= 0) qa = g.pts[pa];
        if (pb >= 0) qb = g.pts[pb];
        if (pa >= 0) consider(qa, pa, px, py, pz, b);
        if (pb >= 0) consider(qb, pb, px, py, pz, b);
    }
    wave_min64(b);
    float bound = h + edge;
    if (b.d2 <= bound * bound * 0.999999f) {
        b.second = fminf(b.second, bound * bound * 0.999999f);  // nothing outside the 27-cell block is closer
        if (runner && total <= 128 && b.pos >= 0) {
            // every candidate is still in a register: the nearest of those that are not the winner, and what bounds the rest
            Best r;
            r.d2 = INFINITY;
            r.idx = 0x7fffffff;
            r.pos = -1;
            r.second = pruned_gap;
            if (pa >= 0 && __float_as_int(qa.w) != b.idx) consider(qa, pa, px, py, pz, r);
            if (pb >= 0 && __float_as_int(qb.w) != b.idx) consider(qb, pb, px, py, pz, r);
            if (seed_pos >= 0 && seed_idx != b.idx && lane == 0) {
                // (the seed is a map point like any other; if a cell the box test pruned holds it, no lane has seen it)
                const float4 qs = g.pts[seed_pos];
                consider(qs, seed_pos, px, py, pz, r);
            }
            wave_min64(r);
            r.second = fminf(r.second, bound * bound * 0.999999f);
            *runner = r;
            if (third && r.pos >= 0) {
                Best t;
                t.d2 = INFINITY;
                t.idx = 0x7fffffff;
                t.pos = -1;
                t.second = pruned_gap;
                const int ia = __float_as_int(qa.w), ib = __float_as_int(qb.w);
                if (pa >= 0 && ia != b.idx && ia != r.idx) consider(qa, pa, px, py, pz, t);
                if (pb >= 0 && ib != b.idx && ib != r.idx) consider(qb, pb, px, py, pz, t);
                if (seed_pos >= 0 && seed_idx != b.idx && seed_idx != r.idx && lane == 0) {
                    const float4 qs = g.pts[seed_pos];
                    consider(qs, seed_pos, px, py, pz, t);
                }
                wave_min64(t);
                t.second = fminf(t.second, bound * bound * 0.999999f);
                *third = t;
            }
        }
        return true;
    }
    if (max_rings < 2) return false;
    // ring 2 of the fine level: the 98 shell cells of the 5x5x5 block by hashed probes, at most two per lane
    for (int s5 = lane; s5 < 125; s5 += 64) {
        const int ox = s5 % 5 - 2, oy = (s5 / 5) % 5 - 2, oz = s5 / 25 - 2;
        if (max(max(ox < 0 ? -ox : ox, oy < 0 ? -oy : oy), oz < 0 ? -oz : oz) < 2) continue;  // rings 0 and 1: done
        const float gx = axis_gap(ox, fx, h), gy = axis_gap(oy, fy, h), gz = axis_gap(oz, fz, h);
        const float gap2 = fmaf(gx, gx, fmaf(gy, gy, gz * gz));
        if (gap2 > b.d2) {
            b.second = fminf(b.second, gap2);
            continue;
        }
        int start, count;
        if (grid_lookup(g, cx + ox, cy + oy, cz + oz, start, count))
            for (int k = start; k < start + count; ++k) consider(g.pts[k], k, px, py, pz, b);  // rare: kept small
    }
    wave_min64(b);
    bound = 2.f * h + edge;
    if (!(b.d2 <= bound * bound * 0.999999f)) return false;
    b.second = fminf(b.second, bound * bound * 0.999999f);
    return true;
}

// `seeded`: nn_pos holds the neighbours the previous iteration of this registration found for the same targets — each
// seeds its query's search (a candidate with a tight bound: most neighbour cells fail the box test unread; like any seed
// it cannot change the minimum).  The unfused loops (point-to-point alignment, lazily estimated normals) searched from
// scratch in every iteration: 128 us per launch at the headline sizes.
__global__ __launch_bounds__(256) void k_search_rows(GridView g, const float4* __restrict__ tgt, int n, int mode,
                                                     int transform, RegState* __restrict__ st, int max_rings,
                                                     int* nn_pos, int* __restrict__ nflag,
                                                     int* __restrict__ worklist, int queue_normals, int seeded) {
    __shared__ int2 cellstack[7][256];
    if (st->done) return;
    const int gid = blockIdx.x * blockDim.x + threadIdx.x;
    const int qi = gid >> 2, sub = gid & 3;
    if (qi >= n) return;  // group-uniform
    const float4 t4 = tgt[qi];
    if (!target_valid(t4.x, t4.y, t4.z, mode)) {  // group-uniform
        if (sub == 0) nn_pos[qi] = -1;
        return;
    }
    float px = t4.x, py = t4.y, pz = t4.z;
    if (transform) transform_point(st->pose, t4.x, t4.y, t4.z, px, py, pz);
    float seed_d2 = INFINITY;
    int seed_idx = 0x7fffffff, seed_pos = -1;
    if (seeded) {
        const int sp = nn_pos[qi];  // (read by the four lanes before lane 0 replaces it below)
        if (sp >= 0 && sp < g.m) {
            const float4 q = g.pts[sp];
            const float dx = q.x - px, dy = q.y - py, dz = q.z - pz;
            seed_d2 = fmaf(dz, dz, fmaf(dy, dy, dx * dx));
            seed_idx = __float_as_int(q.w);
            seed_pos = sp;
        }
    }
    const Best b = search_rows_group(g, px, py, pz, sub, max_rings, &cellstack[0][threadIdx.x], 256, seed_d2, seed_idx,
                                     seed_pos).b;
    if (sub != 0) return;
    nn_pos[qi] = b.pos;
    if (queue_normals && b.pos >= 0 && nflag[b.pos] == 0) {
        if (atomicCAS(&nflag[b.pos], 0, 2) == 0) worklist[atomicAdd(&st->n_worklist, 1)] = b.pos;
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// Ball search (round 4): ONE lane per query, no cooperation, no per-cell bookkeeping.
//
// Round 3's counters put the early iteration launches at 16.6 M VALU + 7.5 M SALU wave-instructions — 8 100 lane-
// instructions per query for a 1-NN over ~50 candidates: box tests of 26 neighbours, LDS lists, group reductions, done by
// four lanes each.  What a query actually needs, measured offline on the bench's frames (tools/ball_stats.py): its own
// cell holds a point within 0.15 h of it (median; h = the cell edge), and a ball of that radius reaches 1 (50 %), 2 (33 %),
// 4 (13 %) or 8 (4 %) cells — the own cell and the ones across its NEARER faces — with ~40 candidates in all; 99.4-99.9 %
// of the queries of an ordinary frame fit that pattern.  So:
//   1. own cell (one hashed probe; the 7 row entries of the cells across the nearer faces are requested with it);
//   2. r = the best distance so far (own cell, seed); cells whose box is farther than r + prune_guard are pruned, their
//      box distance bounds L; r + guard must stay inside the 2x2x2 block (else: the generic paths);
//   3. the surviving cells, candidates four at a time.
// Per candidate: the squared distance (6 instructions) and a sorted insertion into FOUR 32-bit keys by min / med3 (4):
// key = distance bits with the low 8 bits replaced by the candidate's running number j (positive floats order like their
// bits).  The keys name the candidate set the NN cache wants (three points + a bound on everybody else): everybody else
// has a key >= K3, hence a distance >= K3 with its low bits cleared.  Exactness: the true nearest neighbour is one of
// K0..K2 unless four candidates agree in the 15 leading mantissa bits (then: the generic paths); where K1 ties K0 in
// those bits the tied points are compared exactly, (distance, original index) like everywhere else.
// Cells are read in whole groups of four points — past the end of a cell into the next one of the array (real map points:
// more candidates never hurt) or into the four +inf pads behind the last point — so no candidate is masked.
// `stack` = this lane's column of an LDS array [7][stride] int2.  Returns false when the query does not fit the pattern.
// ---------------------------------------------------------------------------------------------------------------------
static constexpr unsigned BALL_JMASK = 255u;
static constexpr int BALL_MAX_CAND = 256;  // candidates (cells rounded up to groups of four) a key's j can number
static constexpr int BALL_W = 4;           // groups of four points a lane has in flight per trip of its scan

__device__ inline unsigned umed3(unsigned a, unsigned b, unsigned c) {  // -> v_med3_u32
    return max(min(a, b), min(max(a, b), c));
}

struct Top4 {
    unsigned k0, k1, k2, k3;  // ascending
    __device__ inline void insert(unsigned c) {
        const unsigned n3 = umed3(k2, k3, c), n2 = umed3(k1, k2, c), n1 = umed3(k0, k1, c);
        k0 = min(k0, c);
        k1 = n1;
        k2 = n2;
        k3 = n3;
    }
};

__device__ inline unsigned ball_key(const float4 q, float px, float py, float pz, int j) {
    const float dx = q.x - px, dy = q.y - py, dz = q.z - pz;
    const float d2 = fmaf(dz, dz, fmaf(dy, dy, dx * dx));
    return (__float_as_uint(d2) & ~BALL_JMASK) | (unsigned)j;
}

// one trip of the scan: W groups of four consecutive points, all W * 4 loads in flight together (a lane's chain of
// dependent round trips, not its instruction count, is what a one-lane search costs: the 4-wide loop of the first build
// took 22 trips for the 64 candidates of the slowest lane of a wave); `a[g]` = the group's first point or the pads
template <int W>
__device__ inline void ball_trip(const float4* const (&a)[W], const int (&tag)[W], float px, float py, float pz, Top4& t) {
    float4 q[W][4];
#pragma unroll
    for (int g = 0; g < W; ++g) {
        q[g][0] = a[g][0];
        q[g][1] = a[g][1];
        q[g][2] = a[g][2];
        q[g][3] = a[g][3];
    }
#pragma unroll
    for (int g = 0; g < W; ++g) {
#pragma unroll
        for (int i = 0; i < 4; ++i) t.insert(ball_key(q[g][i], px, py, pz, tag[g] + i));
    }
}

// exchange inside a group of LPQ lanes (the lanes of one query): lane ^ X — DPP inside a quad, the LDS crossbar beyond
template <int X>
__device__ inline unsigned ball_xchg(unsigned v) {
    if constexpr (X <= 2) return (unsigned)quad_xor<X>((int)v);
    else return (unsigned)__shfl_xor((int)v, X, 64);
}
// the four smallest keys of this lane's and its partner's lists, sorted again (identical in the two lanes: element-wise
// minimum of one sorted list with the other reversed = the smallest four of the union)
template <int X>
__device__ inline void ball_merge(Top4& t) {
    const unsigned b0 = ball_xchg<X>(t.k0), b1 = ball_xchg<X>(t.k1), b2 = ball_xchg<X>(t.k2), b3 = ball_xchg<X>(t.k3);
    const unsigned a0 = min(t.k0, b3), a1 = min(t.k1, b2), a2 = min(t.k2, b1), a3 = min(t.k3, b0);
    t.k0 = t.k1 = t.k2 = t.k3 = ~0u;
    t.insert(a0);
    t.insert(a1);
    t.insert(a2);
    t.insert(a3);
}

// Returns true when the query is settled (pos0..2, L).  false: it goes to the generic paths; seed_pos >= 0 then names the
// best point the own-cell scan found (closer than the caller's seed): their searches start from it.
// LPQ = lanes per query (1, 2, 4 or 8 neighbouring lanes; `sub` = this lane's place among them): the groups of four points
// of every cell alternate between the lanes, each keeps its own four keys, one butterfly merge at the end — a workgroup
// with few misses (iterations 1-6) gives each of them more lanes and a chain of dependent trips that much shorter.  All
// lanes of a group return the same.
template <int W, int LPQ>
__device__ inline bool search_ball_lane(const GridView& g, float px, float py, float pz, float seed_d2, int max_cand,
                                        int2* __restrict__ stack, int stride, int sub, int& pos0, int& pos1, int& pos2,
                                        float& L, int& seed_pos, bool through_empty = true) {
    seed_pos = -1;
    const int cx = cell_coord(px, g.inv_h), cy = cell_coord(py, g.inv_h), cz = cell_coord(pz, g.inv_h);
    const float h = g.h;
    const float fx = fminf(fmaxf(px - (float)cx * h, 0.f), h);
    const float fy = fminf(fmaxf(py - (float)cy * h, 0.f), h);
    const float fz = fminf(fmaxf(pz - (float)cz * h, 0.f), h);
    // the nearer face of every axis: only a ball that reaches it leaves the cell on that side
    const bool lx = fx < h - fx, ly = fy < h - fy, lz = fz < h - fz;
    const float nx = lx ? fx : h - fx, ny = ly ? fy : h - fy, nz = lz ? fz : h - fz;
    // every cell outside the 2x2x2 block on the nearer sides is at least that far
    const float outer = h - fmaxf(nx, fmaxf(ny, nz));
    const unsigned long long key = pack_cell(cx, cy, cz);
    unsigned int slot = hash_cell(key) & g.mask;
    const int o1 = lx ? -1 : 1, o2 = ly ? -3 : 3, o4 = lz ? -9 : 9;
    GridEntry e = g.table[slot];
    int2 rc[7];
    {
        const int2* __restrict__ r = g.rows + (size_t)slot * ROW_STRIDE + 13;
#pragma unroll
        for (int m = 1; m < 8; ++m) rc[m - 1] = r[((m & 1) ? o1 : 0) + ((m & 2) ? o2 : 0) + ((m & 4) ? o4 : 0)];
    }
    if (e.key != key && e.key != GRID_EMPTY) {
        while (true) {
            slot = (slot + 1) & g.mask;
            e = g.table[slot];
            if (e.key == key || e.key == GRID_EMPTY) break;
        }
        const int2* __restrict__ r = g.rows + (size_t)slot * ROW_STRIDE + 13;
#pragma unroll
        for (int m = 1; m < 8; ++m) rc[m - 1] = r[((m & 1) ? o1 : 0) + ((m & 2) ? o2 : 0) + ((m & 4) ? o4 : 0)];
    }
    // own cell EMPTY (round 6): no neighbour row to read — rows belong to occupied cells — but a query that brings a seed still
    // has its ball: the other seven cells of the 2x2x2 block by hashed probes, all in flight together, then the same pruning
    // and the same scan.  (A frame whose constant-velocity guess is off by half a metre puts a sixth of the scan into empty
    // cells in front of the walls it faces; those queries went to the cooperative searches — 110 us first launches.)
    const bool own_empty = e.key != key;
    if (own_empty) {
        if (!through_empty || !(seed_d2 < INFINITY)) return false;  // (option "ball_empty" 0 / nothing bounds the ball)
        grid_lookup7(g, 0x7f, [&](int k) {
            const int m = k + 1;
            return pack_cell(cx + ((m & 1) ? (lx ? -1 : 1) : 0), cy + ((m & 2) ? (ly ? -1 : 1) : 0), cz + ((m & 4) ? (lz ? -1 : 1) : 0));
        }, rc);
        e.start = 0;
        e.count = 0;
    }
    const int cum0 = (e.count + 3) & ~3;
    if (cum0 > max_cand) return false;
    const float4* __restrict__ pad = g.pts + g.m;  // SORTED_PAD points at +inf
    Top4 t;
    t.k0 = t.k1 = t.k2 = t.k3 = ~0u;
    {
        const float4* __restrict__ base = g.pts + e.start;
        for (int j = 0; j < cum0; j += 4 * W * LPQ) {
            const float4* a[W];
            int tag[W];
#pragma unroll
            for (int gr = 0; gr < W; ++gr) {
                tag[gr] = j + 4 * (gr * LPQ + sub);
                a[gr] = tag[gr] < cum0 ? base + tag[gr] : pad;
            }
            ball_trip<W>(a, tag, px, py, pz, t);
        }
    }
    unsigned kbest = t.k0;  // the best of the own cell over the lanes of the query
    if constexpr (LPQ >= 2) kbest = min(kbest, ball_xchg<1>(kbest));
    if constexpr (LPQ >= 4) kbest = min(kbest, ball_xchg<2>(kbest));
    if constexpr (LPQ >= 8) kbest = min(kbest, ball_xchg<4>(kbest));
    if (kbest >= 0x7f800000u && !own_empty) return false;  // nothing finite in the own cell
    // the pruning radius: the nearest so far is no farther than its key with the low bits set; the seed is a map point
    // like any other and is met in its cell
    const float r2 = own_empty ? seed_d2 : fminf(seed_d2, __uint_as_float(kbest | BALL_JMASK));
    const float R = sqrtf(r2) * 1.000001f + g.prune_guard;
    const bool inside = R < outer;
    const float R2 = R * R;
    // a ball that leaves the 2x2x2 block but stays inside the 3x3x3 one (round 6: a query half a metre from its surface in
    // 0.5 m cells — a sixth of the scan when the initial guess is that far off): every cell of the 26 whose box the ball
    // reaches, from the own cell's row or, where the own cell is empty, by hashed probes seven at a time
    const float outer3 = h + fminf(fminf(fminf(fx, h - fx), fminf(fy, h - fy)), fminf(fz, h - fz));
    bool block3 = !inside && through_empty && R < outer3;
    float Lb2 = block3 ? outer3 * outer3 : outer * outer;
    int ns = 0, total = cum0;
    if (block3) {
        unsigned surv = 0;  // bit idx = (oz + 1) * 9 + (oy + 1) * 3 + (ox + 1) of the cells the ball reaches
#pragma unroll
        for (int idx = 0; idx < 27; ++idx) {
            if (idx == 13) continue;
            const float gx = axis_gap(idx % 3 - 1, fx, h), gy = axis_gap((idx / 3) % 3 - 1, fy, h), gz = axis_gap(idx / 9 - 1, fz, h);
            const float gap2 = fmaf(gx, gx, fmaf(gy, gy, gz * gz));
            if (gap2 > R2) Lb2 = fminf(Lb2, gap2);  // (every point such a cell may hold is at least that far)
            else surv |= 1u << idx;
        }
        while (surv && block3) {  // (group-uniform: the lanes of a query hold the same mask)
            unsigned long long pick = 0;  // up to seven cell numbers, five bits each (+ 1: 0 = none)
            int want = 0;
#pragma unroll
            for (int k = 0; k < 7; ++k) {
                if (surv) {
                    const int idx = __ffs((int)surv) - 1;
                    surv &= surv - 1u;
                    pick |= (unsigned long long)(idx + 1) << (5 * k);
                    want |= 1 << k;
                }
            }
            int2 found[7];
            if (own_empty) {
                grid_lookup7(g, want, [&](int k) {
                    const int idx = (int)((pick >> (5 * k)) & 31ull) - 1;
                    return pack_cell(cx + idx % 3 - 1, cy + (idx / 3) % 3 - 1, cz + idx / 9 - 1);
                }, found);
            } else {
                const int2* __restrict__ r = g.rows + (size_t)slot * ROW_STRIDE;
#pragma unroll
                for (int k = 0; k < 7; ++k) {
                    const int idx = (int)((pick >> (5 * k)) & 31ull) - 1;
                    found[k] = (want >> k & 1) ? r[idx] : make_int2(0, 0);
                }
            }
#pragma unroll
            for (int k = 0; k < 7; ++k) {
                if ((want >> k & 1) && found[k].y > 0) {
                    if (ns == 7) {  // (the lane's cell stack holds seven: the generic paths take such a query)
                        block3 = false;
                    } else {
                        stack[ns * stride] = make_int2(found[k].x - total, total);
                        total += (found[k].y + 3) & ~3;
                        ++ns;
                    }
                }
            }
        }
    }
    if (inside) {
#pragma unroll
        for (int m = 1; m < 8; ++m) {
            const int2 c = rc[m - 1];
            if (c.y > 0) {
                const float gap2 = ((m & 1) ? nx * nx : 0.f) + ((m & 2) ? ny * ny : 0.f) + ((m & 4) ? nz * nz : 0.f);
                if (gap2 > R2) {
                    Lb2 = fminf(Lb2, gap2);  // every point of a pruned cell is at least that far
                } else {
                    stack[ns * stride] = make_int2(c.x - total, total);  // candidate j of this cell sits at position j + .x
                    total += (c.y + 3) & ~3;
                    ++ns;
                }
            }
        }
    }
    if ((!inside && !block3) || total > max_cand) {
        // the generic paths take over, from the best point of the own cell if that beats the caller's seed
        if (!own_empty && __uint_as_float(kbest & ~BALL_JMASK) < seed_d2) seed_pos = e.start + (int)(kbest & BALL_JMASK);
        return false;
    }
    {
        int s = 0, off = 0, jnext = cum0;
        for (int j = cum0; j < total; j += 4 * W * LPQ) {
            const float4* a[W];
            int tag[W];
#pragma unroll
            for (int gr = 0; gr < W; ++gr) {
                const int jg = j + 4 * (gr * LPQ + sub);
                while (jg >= jnext && s < ns) {  // (cells start at multiples of four: a group never straddles two)
                    off = stack[s * stride].x;
                    ++s;
                    jnext = s < ns ? stack[s * stride].y : total;
                }
                tag[gr] = jg;
                a[gr] = jg < total ? g.pts + (off + jg) : pad;
            }
            ball_trip<W>(a, tag, px, py, pz, t);
        }
    }
    if constexpr (LPQ >= 2) ball_merge<1>(t);
    if constexpr (LPQ >= 4) ball_merge<2>(t);
    if constexpr (LPQ >= 8) ball_merge<4>(t);
    // pads behind the last map point (+inf) and NaN distances are no candidates
    if (t.k1 >= 0x7f800000u) t.k1 = ~0u;
    if (t.k2 >= 0x7f800000u) t.k2 = ~0u;
    if (t.k3 >= 0x7f800000u) t.k3 = ~0u;
    if (t.k3 != ~0u && ((t.k3 ^ t.k0) & ~BALL_JMASK) == 0u) {  // four candidates within the key's resolution
        return false;
    }
    const auto position = [&](unsigned k) -> int {
        if (k == ~0u) return -1;
        const int j = (int)(k & BALL_JMASK);
        if (j < cum0) return e.start + j;
        int s = 0;
        if (ns > 4 && stack[4 * stride].y <= j) s = 4;
        if (s + 2 < ns && stack[(s + 2) * stride].y <= j) s += 2;
        if (s + 1 < ns && stack[(s + 1) * stride].y <= j) s += 1;
        return stack[s * stride].x + j;
    };
    int p0 = position(t.k0), p1 = position(t.k1), p2 = position(t.k2);
    if (t.k1 != ~0u && ((t.k1 ^ t.k0) & ~BALL_JMASK) == 0u) {
        // K1 (and perhaps K2) ties K0 in the bits a key keeps: the exact (distance, original index) order decides
        const float4 q0 = g.pts[p0], q1 = g.pts[p1];
        Best b;
        b.d2 = INFINITY;
        b.idx = 0x7fffffff;
        b.pos = -1;
        b.second = INFINITY;
        consider(q0, p0, px, py, pz, b);
        consider(q1, p1, px, py, pz, b);
        if (t.k2 != ~0u && ((t.k2 ^ t.k0) & ~BALL_JMASK) == 0u) consider(g.pts[p2], p2, px, py, pz, b);
        if (b.pos == p1) {
            p1 = p0;
            p0 = b.pos;
        } else if (b.pos == p2) {
            p2 = p0;
            p0 = b.pos;
        }
    }
    // a point read twice (past the end of a cell into a cell that is scanned as well) names one candidate
    if (p1 == p0) p1 = -1;
    if (p2 == p0 || p2 == p1) p2 = -1;
    float L2 = Lb2;
    if (t.k3 != ~0u) L2 = fminf(L2, __uint_as_float(t.k3 & ~BALL_JMASK));
    pos0 = p0;
    pos1 = p1;
    pos2 = p2;
    L = sqrtf(L2);
    return true;
}

// ---------------------------------------------------------------------------------------------------------------------
// Fused iteration kernel (every needed normal is ready): the search above + the point-to-plane row of each query +
// the per-block partial normal equations, in one launch — no nn_pos round trip, no second pass over the targets.
// (Measured and dropped in round 2: solving iteration k in the prologue of launch k + 1 — every workgroup redundantly,
// group rows published with agent-scope stores and a ticket per 32 workgroups — instead of the separate single-block
// sum-and-solve launch: 638 vs 516 + 154 us per frame of kernel time, no gain in frame time, 11 % slower with four
// sequences per GPU.  The solve launch costs 5.4 us with ONE partial row and 7.9 us with 1024: it is launch and
// cold-load latency, which moving the work does not remove.)
// 128 queries per block: their 9-float rows go to LDS, then 4 x 30 threads each own one packed element of a quarter of
// the queries and add up its products in f64 in a fixed order (bit-reproducible), one partial row per block.
// ---------------------------------------------------------------------------------------------------------------------
static constexpr int IT_THREADS = 512;            // 128 queries x 4 lanes per block -> N/128 partial rows
static constexpr int IT_QUERIES = IT_THREADS / 4;

// per-block partial normal equations from the 9-float rows of the block's queries: for every 128 queries 4 x 30 threads,
// element e of quarter `qtr` of them, f64, fixed order (bit-reproducible).  The canonical order of the whole sum
// (solve_device.h::sum_partials_vt): 128 queries -> base row r = (p0 + p1) + (p2 + p3); base rows s, s + S, s + 2S, s + 3S
// (S = a quarter of the base rows) -> super-row (r0 + r1) + (r2 + r3); super-rows in the strided 8-accumulator pattern.  A
// block of 128 queries (Q = 128) writes its base row, a block of 512 queries — four 128-query chunks S base rows apart —
// its super-row: a quarter of the rows for the summing kernel, same bits.
// RET: the sum is returned (threads < NEQ; 0.0 elsewhere) instead of being stored
// (Round 6, measured and dropped: every row converted to float64 ONCE into LDS the searches no longer need, the summing
// threads reading doubles — 9 conversions per lane instead of 64.  The summing is bound by LDS bandwidth, not by VALU issue:
// 512 threads x 64 operands x 4 B = 131 KB per workgroup = 0.43 us of the CU's 128 B per clock, twice that with doubles;
// late launches 10.5 -> 11.1 us, headline and batched throughput -4 %: profiles/r06_ab_sessions.txt.)
// PADDED (round 6): the caller's row buffer has one spare row behind every 32 (row of slot s at s + (s >> 5)), and a block of
// 512 queries sums differently: thread = (element e = tid / 16, lane L = 4 * base row + quarter), so that the SIXTEEN partial
// sums of an element sit in one DPP row of 16 lanes and the canonical tree — (p0 + p1) + (p2 + p3) per base row, then
// (r0 + r1) + (r2 + r3) — is four DPP exchanges (lane ^ 1, ^ 2, the mirror of the half row, the mirror of the row: each adds
// exactly the pair the tree adds; a sum does not care which of its two operands comes first) instead of a round through LDS:
// no `part` array, no barrier, no second stage of 16 reads and 15 dependent adds.  The spare rows spread the sixteen lanes'
// rows (32 rows = 288 floats apart: one LDS bank) over sixteen banks (33 rows = 297 floats: 9 L mod 32).  Same products,
// same order, same association: the same bits.
__device__ __forceinline__ int padded_row(int slot) { return slot + (slot >> 5); }

template <int Q, bool RET = false, bool PADDED = false>
__device__ inline double block_reduce_rows(const float (*rowbuf)[9], double (*part)[NEQ], double* __restrict__ partials,
                                           int block) {
    const LocalTid threadIdx = RET ? reloaded_tid() : LocalTid{::threadIdx.x};  // (RET: inside the resident tail's loop)
    constexpr int SUB = Q / IT_QUERIES;  // base rows per block
    if constexpr (PADDED && SUB == 4 && !RET) {
        static_assert(NEQ == 32, "one DPP row of 16 lanes per element, 32 elements = 512 threads");
        if ((int)threadIdx.x < 16 * NEQ) {  // (whole waves: every lane of the rows below is active)
            const int e = (int)threadIdx.x >> 4, L = (int)threadIdx.x & 15, qtr = L & 3, sb = L >> 2;
            double acc = 0.0;
            if (e < NEQ_USED) {
                int a, b2;
                neq_operands_lut(e, a, b2);
                const int j0 = sb * IT_QUERIES + qtr * (IT_QUERIES / 4);
                const float(*r)[9] = rowbuf + padded_row(j0);  // (32 consecutive rows: j0 is a multiple of 32)
#pragma unroll 8
                for (int j = 0; j < IT_QUERIES / 4; ++j) acc = fma((double)r[j][a], (double)r[j][b2], acc);
            }
            acc += row16_step<0>(acc);  // p0 + p1 | p2 + p3
            acc += row16_step<1>(acc);  // (p0 + p1) + (p2 + p3) = the base row's sum, in its four lanes
            acc += row16_step<2>(acc);  // r0 + r1 | r2 + r3
            acc += row16_step<3>(acc);  // (r0 + r1) + (r2 + r3)
            if (L == 0) partials[(size_t)block * NEQ + e] = acc;
        }
        return 0.0;
    }
    if ((int)threadIdx.x < SUB * 4 * NEQ) {
        const int e = threadIdx.x & (NEQ - 1), qtr = (threadIdx.x / NEQ) & 3, sb = threadIdx.x / (4 * NEQ);
        double acc = 0.0;
        if (e < NEQ_USED) {
            int a, b2;
            neq_operands_lut(e, a, b2);
            const int j0 = sb * IT_QUERIES + qtr * (IT_QUERIES / 4);
            const float(*r)[9] = rowbuf + (PADDED ? padded_row(j0) : j0);
            // (fma: the product of two floats is exact in float64 — 48 mantissa bits — so the fused form rounds once where the
            // separate multiply and add round once too: the same bits, one instruction less per row)
#pragma unroll 8
            for (int j = 0; j < IT_QUERIES / 4; ++j) acc = fma((double)r[j][a], (double)r[j][b2], acc);
        }
        part[sb * 4 + qtr][e] = acc;
    }
    __syncthreads();
    if (threadIdx.x < NEQ) {
        const int e = threadIdx.x;
        double r[SUB];
#pragma unroll
        for (int sb = 0; sb < SUB; ++sb)
            r[sb] = (part[4 * sb][e] + part[4 * sb + 1][e]) + (part[4 * sb + 2][e] + part[4 * sb + 3][e]);
        double v = r[0];
        if (SUB == 4) v = (r[0] + r[1]) + (r[2] + r[3]);
        if (RET) return v;
        partials[(size_t)block * NEQ + e] = v;
    }
    return 0.0;
}


// ---------------------------------------------------------------------------------------------------------------------
// The kernel, with in-block compaction of the cache misses.
//
// NN cache (exact): a search leaves a CANDIDATE SET — the neighbour and up to two more map points (the local bests of
// the other lanes of a 4-lane search; the runner-up and the third of a whole-wave search) — and a lower bound L on the
// distance to EVERY map point outside the set (the next candidate, the box distance of every pruned cell, the bound of
// the last ring), plus the iteration k it ran in.  The target has moved by delta = |T_now p - T_k p| since (T_k from the
// pose history of the registration), so every point outside the set is still >= L - delta away: if the nearest member of
// the set is strictly closer than that it is THE nearest neighbour and the search is skipped.  (One cached point made
// the slack the gap between the first and the second neighbour: in a map merged from eight scans that is millimetres,
// and half the scan searched again after a 5 mm step.  Cells are pruned only beyond a guard band of the best distance
// — "prune_guard" — because the box distance of a pruned cell is part of L: a ball that all but touches a cell face left
// a slack of micrometres, and those ~20 queries of a scan missed in every late launch.)  (Round 2 subtracted the step of every iteration from L instead: a sum of
// |steps| that keeps growing while the pose merely jitters around its fixed point — it eroded the slack of the same few
// dozen boundary queries in every late iteration, and rewrote the 1 MB cache per launch to do so.  The displacement since
// the search stops growing once the pose has converged, and a hit writes nothing.)
// From the third iteration on most queries keep their neighbour, but a wave holds 16 queries
// and runs the whole search path as soon as ONE of them misses: with a few per cent of misses nearly every wave still
// paid for a search.  Here the block works in two phases:
//   A. one lane per query (2 of the 8 waves): transform, cache test; a hit forms its row at once (map point and normal
//      are fetched together, speculatively, behind the cache entry); a miss is appended to a list in LDS;
//   B. the 4-lane groups of the whole block take the list entries in order: only ceil(misses / 16) waves search, the
//      others go straight to the reduction.
// The list order is not deterministic, the result is: every row lands in the slot of its query.
// HIT RECORD (round 6; VERDICT r5 item 2: "make a cache hit cost what it uses").  A hit used to request 128 bytes — target,
// cache entry, the three candidate points AND their three normals, six of the eight loads 16-byte gathers — to use 36, and
// decided in two dependent round trips (entry, then candidates).  Now the first hit behind a search leaves a RECORD next to
// the entry: the winner's point, its normal, the iteration k' of that hit and a bound B on the distance of EVERY OTHER map
// point at pose k' — B = min(L - delta(k -> k'), |candidate 2|, |candidate 3|) with the entry's L (everything outside the
// set, at the search's pose k) and the other members' exact distances at k'.  A later iteration tests the record alone:
// |winner - T p| < B - delta(k' -> now) certifies the winner as THE nearest neighbour (every other point is still farther),
// from 48 coalesced bytes per query and no gather; the record is a lower-bound argument like the entry's, exact.  Only when
// that test fails (or there is no record: the iteration right behind a search) the entry's candidate set is fetched and
// tested as before — so every query the record certifies is one the set test certifies too (triangle inequality), the
// searches are the same searches, the neighbours the same neighbours.  A search invalidates the record of its query.
// (Not in the resident tail, which keeps the set in registers, nor with normals on demand, whose rows wait for flags.)
// First iteration of a frame: no cache yet, but `frame_seed` (optional) names, per scan slot, the map point that was the
// neighbour of the same slot at the end of the PREVIOUS frame (original map index, shifted by the points evicted
// since) — a candidate that starts the search with a tight bound; like any seed it cannot change the minimum.
// ---------------------------------------------------------------------------------------------------------------------
struct IterInputs {
    const float4* tgt;       // targets (x, y, z, row)
    const float4* normals;   // by cell-sorted position
    int4* nn_cache;          // per query: (position | iteration of the search << 24, bits(L), positions of up to two more
                             // candidates or -1) — L bounds every map point that is not a member of that candidate set
    float4* rec;             // HIT RECORDS, two float4 per query, or nullptr (see "hit record" at the kernel): (winner xyz,
                             // bits(bound)), (winner normal, bits(iteration the bound refers to; -1: no record))
    const float* pose_hist;  // [iteration][12]: the pose every earlier iteration of this registration ran with
    const int* frame_seed;   // original map index per query or nullptr
    double* partials;
    int n, mode, max_rings, use_cache;
    int iter;                // index of this iteration within the registration (0, 1, ..)
    int wave_misses;         // up to that many cache misses in a block: a whole wave per miss (0: never)
    int ball;                // option "ball_search": the misses go through search_ball_lane first (one lane each)
    int ball_max;            // option "ball_max": ... those with up to that many candidates (cells rounded up to fours)
    int far_lanes;           // option "far_lanes": lanes of a handed-back query in phase BF (0: no such phase, 16)
    int far_max;             // option "far_max": the most handed-back queries of a workgroup phase BF takes
    int far_min;             // option "far_min": ... and it takes more than that many (fewer: a wave each, phase B1)
    int ball_lanes;          // option "ball_lanes": the most lanes a miss gets in phase B0 (1, 2 or 8)
    int ball_empty;          // option "ball_empty": a seeded miss whose own cell is empty stays with the ball search (seven hashed probes)
    int chunk_stride;        // 512-query shape: S = base rows between the four 128-query chunks of a workgroup (= its super-rows)
    float4* normals_rw;      // LAZY instantiation: the normal cache and its flags, writable; fine rings of its kNN
    int* nflag;
    int knn_rings;
    int tail_iters;          // resident tail (TAIL instantiation): iterations this launch runs (iter .. iter + tail_iters - 1)
    int refresh_at;          // ... the one of them the refresh margin applies to (a launch of its own: `refresh_margin` is set for it alone)
    unsigned long long* tail_rows;  // ... the tagged super-rows its workgroups hand to its lead: [rows][NEQ][2] granules
    // XCD sectors (option "xcd_sectors"): the hardware deals consecutive workgroups round-robin to the 8 XCDs, each with
    // an L2 of its own — with consecutive queries in consecutive workgroups every L2 has to hold the rows and points of
    // the WHOLE map.  With swz_bpr_shift >= 0 the workgroups of one XCD take one azimuth sector (x elevation band) of the
    // range image instead: physical workgroup p -> XCD c = p % 8 -> sector c % swz_sectors, band c / swz_sectors; the
    // j-th workgroup of the class -> row band * swz_band_rows + (j >> swz_bpr_shift), block (j & bpr - 1) of the
    // sector's bpr blocks of that row.  A permutation of the logical workgroups (queries and partial rows belong to the
    // LOGICAL index): same bits.
    float refresh_margin;    // ONE early launch (it searches anyway): an entry whose slack is about to run out counts as
                             // a miss and is searched again there, not alone in a late launch that nothing else holds up
    int swz_bpr_shift;       // log2(blocks per row and sector), -1: identity
    int swz_sectors;         // azimuth sectors (8, 4, 2 or 1)
    int swz_band_rows;       // image rows per elevation band
    int swz_row_blocks;      // blocks per image row
};

// logical workgroup of physical workgroup `p` (`off` = 1 when workgroup 0 is the lead of a lead launch)
__device__ inline int logical_block(const IterInputs& in, int p, int off) {
    if (in.swz_bpr_shift < 0) return p - off;
    const int c = p & 7, first = off + ((c - off) & 7), j = (p - first) >> 3;
    const int sector = c % in.swz_sectors, band = c / in.swz_sectors;
    return (band * in.swz_band_rows + (j >> in.swz_bpr_shift)) * in.swz_row_blocks +
           (sector << in.swz_bpr_shift) + (j & ((1 << in.swz_bpr_shift) - 1));
}

// One sequence's arguments of a fused iteration launch, as the batched launch reads them from device memory
// (k_iterate_batch): exactly what k_iterate_compact takes by value.
struct alignas(16) IterateDesc {
    GridView g;
    IterInputs in;
    RegState* st;
    AlignParams ap;
    LeadArgs lead;
    int blocks;  // workgroups of this sequence's share of the launch (its lead not counted)
    int pad[3];
};
static constexpr int BATCH_LEAD_SLOTS = 32;  // lead workgroups at the head of a batched launch = the most sequences of a batch
static_assert(BATCH_LEAD_SLOTS % 8 == 0 && BATCH_LEAD_SLOTS == ICP_BATCH_MAX_SEQUENCES, "XCD alignment of the work workgroups");

// cache entry .x = cell-sorted position of the neighbour | iteration of the search << 24 (-1: no neighbour); a position
// needs 24 bits (maps of up to 16.7 M points use the cache), iterations wrap into 7 bits — harmlessly: an entry older
// than CACHE_HIST launches is a miss
// The square root of the cache tests: the hardware's v_sqrt_f32 (1 ulp) instead of the correctly rounded sqrtf() (which the
// compiler expands to 14 instructions: scaling for denormals, two correction steps, a class test).  The tests are BOUNDS, not
// results — a hit is certified when an upper bound of the winner's distance is below a lower bound of everybody else's — and
// their factors (1.000001 up, 0.999999 down: 8 ulp) cover the 1.5 ulp of the squared distance's three roundings plus this one;
// a denormal argument may come back as 0 — 1e-19 below the truth, against the 1e-7 every test adds to the displacement.
// Whether a query is certified or searched never changes its neighbour (both are exact): same bits.
__device__ __forceinline__ float sqrt_1ulp(float x) { return __builtin_amdgcn_sqrtf(x); }

static constexpr int CACHE_ITER_SHIFT = 24;
static constexpr int CACHE_POS_MASK = (1 << CACHE_ITER_SHIFT) - 1;
static constexpr int CACHE_HIST = 24;  // poses kept in LDS (1152 B: what the 128-query shape has left of its 40 KB)

__device__ inline int pack_cache(int pos, int iteration) {
    return pos < 0 ? -1 : (pos | ((iteration & 127) << CACHE_ITER_SHIFT));
}

// MINW = minimum waves per SIMD the register allocation must leave room for: 8 keeps all 4 blocks of a CU (the whole
// 131 072-point scan) resident in one round at 64 VGPRs (some spilled dwords), 6 allows 80 VGPRs (3 blocks per CU, a
// quarter of the blocks in a second round)
// Two shapes, 512 threads each:
//   Q = 128 queries per block (a 4-lane group for every query: the early iterations, where most queries search);
//   Q = 512 queries per block (one lane per query: the late iterations, where a handful of queries search — three
//       quarters of the waves of the other shape would only be launched to wait at the barriers — and a quarter of the
//       partial rows for the single workgroup that sums them).
// Same bits either way (block_reduce_rows).
// The LEAD workgroup of a lead launch (LeadArgs, icp_internal.h; one workgroup more than the queries need): fixed-order
// sum of the partial rows the PREVIOUS launch left, 6x6 solve, pose update, RegState — what k_sum_solve does in a launch
// of its own — and the new pose into the mailbox, for the workgroups of this launch that poll for it.  A separate
// workgroup on a path of its own (it returns before the iteration proper): folded into workgroup 0's regular work the
// f64 Cholesky set the register allocation of everybody's main path (69 spilled VGPRs instead of 37 in the 64-register
// shape, every launch 20 % slower), and as a non-inlined call it was itself three times slower (both measured).
// `scratch` = LDS the iteration does not use yet.
template <int THREADS>
__device__ inline void lead_solve(const LeadArgs& lead, RegState* __restrict__ st, AlignParams ap, double* scratch,
                                  long long* stamps) {
    double(*lds)[NEQ] = reinterpret_cast<double(*)[NEQ]>(scratch);
    double* total = scratch + 32 * NEQ;
    if (stamps && threadIdx.x == 0) stamps[0] = wall_clock64();  // dev: entry | rows summed | solved and published
    // (block-uniform; a hand-off that timed out in an earlier launch of this registration left incomplete rows behind: the
    // chain of solves stops there, as behind the end of the loop — icp_register_end re-runs the rest)
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
    // (the rows are requested before `done` is looked at: one round trip instead of two in a row; behind the end of the
    // loop the sum is simply discarded)
    sum_partials_vt<THREADS>(lead.prev_partials, lead.prev_rows, lead.prev_quad, total, lds);
    __syncthreads();
    if (done) {  // the loop ended earlier: this launch has nothing to do, and its workgroups must hear it
        if (threadIdx.x < BOX_USED) {
            const int k = threadIdx.x;
            box_store(lead.box, lead.gen, k,
                      k < 12 ? __float_as_uint(st->pose[k]) : (k == 12 ? 1u : (unsigned)st->iter));
        }
        return;
    }
    if (stamps && threadIdx.x == 0) stamps[1] = wall_clock64();
    if (threadIdx.x < NEQ) lead.neq[threadIdx.x] = total[threadIdx.x];
    if (threadIdx.x < 64)
        solve_and_update(st, total, ap, lead.loss_hist, lead.dx_hist, lead.hist_cap, it, pose_in, params_in, lead.box,
                         lead.gen);
    if (stamps && threadIdx.x == 0) stamps[2] = wall_clock64();
}

// The lead's work inside the RESIDENT TAIL (k_iterate_compact<.., TAIL>): workgroup 0 carries out solve j of the launch in
// front of its own share of iteration j — j = 0 from the rows the PREVIOUS launch left (plain loads), j >= 1 from the tagged
// rows the workgroups of this launch handed over in iteration j - 1 (sum_tagged_rows_vt; tag = the generation that
// iteration consumed), j = tail_iters behind the last iteration (what a k_sum_solve launch used to do) — and publishes the
// pose as generation `gen`.  Pose, parameters and iteration count travel from solve to solve in `carry_s` (LDS; the
// RegState is written for the host and for later launches, never read back here).  False: the launch is over for this
// workgroup (the loop had ended before the launch, or a row did not arrive in time).
template <int THREADS>
__device__ inline bool tail_lead_step(const LeadArgs& lead, RegState* __restrict__ st, AlignParams ap, double* scratch,
                                      float* __restrict__ carry_s, int j, unsigned gen,
                                      const unsigned long long* __restrict__ tail_rows, int nrows, long long* stamps) {
    const LocalTid threadIdx = reloaded_tid();  // (icp_internal.h)
    double(*lds)[NEQ] = reinterpret_cast<double(*)[NEQ]>(scratch);
    double* total = scratch + 32 * NEQ;
    int* failed = reinterpret_cast<int*>(total + NEQ);
    if (stamps && threadIdx.x == 0) stamps[0] = wall_clock64();  // dev: entry | rows summed | solved and published
    int done = 0, it = 0;
    float pose_in[16], params_in[6];
    if (threadIdx.x == 0) *failed = 0;
    if (j == 0) {
        done = st->done | (st->handoff_timeouts > 0 ? 1 : 0);  // (block-uniform, as in lead_solve)
        if (threadIdx.x < 64) {
            it = st->iter;
#pragma unroll
            for (int k = 0; k < 16; ++k) pose_in[k] = st->pose[k];
#pragma unroll
            for (int k = 0; k < 6; ++k) params_in[k] = st->params[k];
        }
        sum_partials_vt<THREADS, true>(lead.prev_partials, lead.prev_rows, lead.prev_quad, total, lds);
    } else {
        __syncthreads();  // (*failed)
        sum_tagged_rows_vt<THREADS>(tail_rows, nrows, gen - 1u, total, lds, wall_clock64() + lead.timeout_ticks, failed);
        if (threadIdx.x < 64) {
#pragma unroll
            for (int k = 0; k < 16; ++k) pose_in[k] = carry_s[k];
#pragma unroll
            for (int k = 0; k < 6; ++k) params_in[k] = carry_s[16 + k];
            it = __float_as_int(carry_s[22]);
        }
    }
    __syncthreads();
    if (done) {  // the loop ended before this launch: its workgroups must hear it
        if (threadIdx.x < BOX_USED) {
            const int k = threadIdx.x;
            box_store(lead.box, gen, k, k < 12 ? __float_as_uint(st->pose[k]) : (k == 12 ? 1u : (unsigned)st->iter));
        }
        return false;
    }
    if (*failed) {  // never seen: a workgroup of this launch is not running (a GPU shared with foreign work): the host re-runs
        if (threadIdx.x == 0) atomicAdd(&st->handoff_timeouts, 1);  // the rest on per-iteration launches
        return false;
    }
    if (stamps && threadIdx.x == 0) stamps[1] = wall_clock64();
    if (threadIdx.x < NEQ) lead.neq[threadIdx.x] = total[threadIdx.x];
    if (threadIdx.x < 64) {
        SolveOut o;
        solve_and_update(st, total, ap, lead.loss_hist, lead.dx_hist, lead.hist_cap, it, pose_in, params_in, lead.box, gen, &o);
        const int lane = threadIdx.x;  // every lane holds the same `o`: lane k parks element k
        float v = 0.f;
#pragma unroll
        for (int k = 0; k < 16; ++k)
            if (lane == k) v = o.pose[k];
#pragma unroll
        for (int k = 0; k < 6; ++k)
            if (lane == 16 + k) v = o.params[k];
        if (lane == 22) v = __int_as_float(it + 1);
        if (lane < 23) carry_s[lane] = v;
    }
    if (stamps && threadIdx.x == 0) stamps[2] = wall_clock64();
    __syncthreads();  // (carry_s; the scratch goes back to the searches)
    return true;
}

// STATS: the dev instrumentation ("search_stats": path counters, phase stamps) is compiled into an instantiation of its
// own; the product kernel carries none of it (72 s_memrealtime and their branches in a kernel that spills SGPRs).
// TAIL (the resident tail): ONE launch runs iterations in.iter .. in.iter + in.tail_iters - 1.  Every workgroup stays
// resident and loops: pose of iteration i from the mailbox (generation lead.gen + i - in.iter) -> cache test / searches /
// rows / block sums as in a launch of its own -> its super-row handed to the lead as TAGGED granules (solve_device.h) instead
// of a store the next launch reads.  The lead workgroup loops with them (lead_solve<.., true>).  What a launch boundary
// cost per late iteration — the dispatch of 257 workgroups, a cold L2 (the XCDs' L2s are invalidated at every kernel start:
// 4.7 MB of fabric reads per launch for a working set that had not changed), the k_sum_solve launch behind the last
// iteration — is paid once.  The workgroups wait for one another: all of them must be RESIDENT together (257 of the
// 512-thread shape on 256 CUs); every wait has a wall-clock bound, and a wait that runs out (a GPU shared
// with foreign work) ends the launch with RegState.handoff_timeouts raised — icp_register_end then re-runs what is left on
// per-iteration launches and the context stops using the tail.
// (the tail is built for MINW = 2: ONE workgroup per CU with the whole register file — the loop keeps the launch's uniform
// state and per-lane addresses alive around the searches, which at 128 registers spilled 65 of them inside the cache test of
// every iteration — so its lead cannot be a workgroup of its own beside 256 others: workgroup 0 is lead AND takes its share
// of the queries, tail_lead_step in front of each of its iterations.)
// LAZY_KN > 0 (the point-to-plane normals ON DEMAND, `KdTreeLocalMap.__get_normals`, local_map.py:397-422: a normal is
// estimated when a scan first touches its map point and cached until the next build_model): a neighbour whose normal has
// not been estimated yet does not form its row at once — the query waits in an LDS list, and behind the searches whole
// waves estimate those normals (the exact LAZY_KN-nearest neighbours from scratch: wave_knn_rings over rings 0.., coarse
// level, exhaustive scan — the straggler path of the eager kernels, same neighbours, same order-independent covariance
// sums, same eigen-solve: the same bits), store them for every later launch and form the rows.  For a scan that touches a
// few per cent of the map (the published configuration: 6 000 targets, 180 000 map points) this replaces the estimation
// of EVERY normal behind every map update (89 + 15 us per frame) by ~6 000 whole-wave searches spread over the chip
// inside the first iteration launch.  Built for 256 registers (MINW = 2) and the 512-query shape.
template <int KN>
__device__ inline void lazy_cov_wave(const GridView& g, int s, int lane, int max_rings, int* __restrict__ wl,
                                     float* __restrict__ cov);
template <int KN>
__device__ inline bool lazy_cov_group(const GridView& g, int s, int sub, float* __restrict__ cov, int2* __restrict__ stack,
                                      int stride);
__device__ inline void normal_from_cov(const float* __restrict__ cov, int s, float4* __restrict__ normals,
                                       int* __restrict__ nflag);

// The body of the launch, shared by the kernel with by-value arguments (one sequence: k_iterate_compact) and the kernel
// that takes them from a descriptor table in device memory (B sequences per launch: k_iterate_batch).  `is_lead`: this
// workgroup is the lead of its sequence (block-uniform); `pb` / `off`: its physical number among the workgroups of the
// sequence and the number of lead workgroups in front of them (logical_block); `bx` / `gx`: blockIdx.x / gridDim.x of a
// launch that holds ONE sequence (the resident tail, the dev stamps).
template <int THREADS, int Q, bool STATS, bool TAIL, int LAZY_KN, bool REC_BUILD>
__device__ __forceinline__ void iterate_body(GridView g, IterInputs in, RegState* __restrict__ st, AlignParams ap,
                                             LeadArgs lead, const bool is_lead, const int pb, const int off, const int bx,
                                             const int gx) {
    if (!STATS) {  // (constant-folds every `if (g.dbg)` / `if (stamps)` below and inside the inlined searches)
        g.dbg = nullptr;
        g.stamps = nullptr;
    }
    __shared__ float rowbuf[Q + Q / 32][9];  // (row of slot s at padded_row(s): block_reduce_rows)
    __shared__ double part[Q / 32][NEQ];
    __shared__ int2 cellstack[7][THREADS];
    __shared__ float4 miss_p[Q];   // transformed target + bits(query slot)
    __shared__ int4 miss_seed[Q];  // bits(seed d2), seed index, seed position
    __shared__ int nmiss;
    __shared__ float pose_s[12];               // rows 0-2 of the pose of this iteration
    __shared__ float hist_s[CACHE_HIST][12];   // ... and of the CACHE_HIST iterations before it (slot = iteration % CACHE_HIST)
    __shared__ int ctl_s[4];                  // logical block | done | iteration | hand-off failures
    __shared__ int dbg_s[16];                 // dev-only ("search_stats" = 1): this workgroup's path counters
    __shared__ float carry_s[24];             // TAIL, workgroup 0: pose, parameters and iteration count from solve to solve
    constexpr bool LAZY = LAZY_KN > 0;
    __shared__ float4 lazy_p[LAZY ? Q : 1];   // LAZY: transformed target + bits(query slot) of a query that waits for a normal
    __shared__ int lazy_pos[LAZY ? Q : 1];    // ... the map point whose normal it waits for
    __shared__ float lazy_cov[LAZY ? Q : 1][7];
    __shared__ int nlazy;
    static_assert(sizeof(cellstack) >= (32 * NEQ + NEQ + 1) * sizeof(double), "the lead's scratch lives in the cell stacks");
    // In a lead launch (LeadArgs) workgroup 0 is the lead: it solves the previous iteration and publishes the pose the
    // others poll for.  The hardware dispatches workgroups in ascending order, so whoever polls, polls for a workgroup
    // placed before it (a role ticket drawn from a counter would make that independent of the dispatch order — and costs
    // a thousand same-address device-scope atomics per launch, ~10 us: measured); should the order ever differ, the
    // wall-clock bound of the poll turns the wait into ICP_ERR_HIP instead of a hang.
    if (lead.box) {
        if (is_lead) {  // block-uniform
            lead_solve<THREADS>(lead, st, ap, reinterpret_cast<double*>(&cellstack[0][0]),
                                (g.stamps && in.iter < 24) ? g.stamps + 4 * ((size_t)in.iter * 1024 + 1023) : nullptr);
            return;
        }
    } else if (st->done) {
        return;  // classic launch behind the end of the loop (block-uniform)
    }
    long long t_entry = g.stamps ? wall_clock64() : 0;
    const int vb = logical_block(in, pb, off);  // which queries, which partial row
    // which query a local slot stands for: the 128-query shape takes consecutive queries (one BASE row of the canonical sum,
    // solve_device.h); the 512-query shape takes FOUR base rows a quarter of the scan apart — base rows vb, vb + S, vb + 2S,
    // vb + 3S, exactly the four that form super-row vb — so a workgroup mixes four regions of the scan (the rings near the
    // floor hit dense map cells, the upper rings sparse ones: with 512 consecutive queries per workgroup the slowest
    // workgroup of an early launch searched for 30 us against a mean of 14, and the launch lasts as long as it does)
    const auto query_of = [&](int slot) -> int {
        if (Q == IT_QUERIES || in.chunk_stride <= 0) return vb * Q + slot;
        return (vb + (slot / IT_QUERIES) * in.chunk_stride) * IT_QUERIES + (slot % IT_QUERIES);
    };
    // ---- phase A, first half: everything that does not depend on the pose is requested now — target, cache entry and,
    // behind it, the cached neighbour and its normal (or the frame seed and its map point): in a lead launch these loads
    // are in flight while the lead workgroup solves
    int* const dbg_global = g.dbg;
    if (g.dbg) g.dbg = dbg_s;  // the search paths count into LDS (flat atomics); flushed, and stored per workgroup, at the end
    // (TAIL: the poses of the iterations this launch runs come from the mailbox and are kept in hist_s by the workgroup
    // itself — the lead's stores to pose_hist are for later launches and the host)
    if (in.use_cache && (int)threadIdx.x < 12 * CACHE_HIST) {
        const int e = threadIdx.x / 12, j = in.iter - 1 - e;  // iteration j, most recent first
        if (j >= 0) hist_s[j % CACHE_HIST][threadIdx.x % 12] = in.pose_hist[(size_t)j * 12 + threadIdx.x % 12];
    }
    // what a query brings to its cache test — target, cache entry, the candidate set's points and normals: 32 registers.  The
    // TAIL keeps them from trip to trip (it is built for 256 registers) and loads again only what a search has rewritten: in
    // the late iterations, where every query hits, a trip reads nothing but the pose
    bool valid = false, need_load = true;
    float4 t4 = make_float4(0.f, 0.f, 0.f, 0.f), cq = make_float4(0.f, 0.f, 0.f, 0.f), cn = cq, cq2 = cq, cn2 = cq,
           cq3 = cq, cn3 = cq;
    int4 c = make_int4(-1, 0, -1, -1);
    int seed_o = -1, seed_sp = -1;
    // hit records (see above): compiled into the REC_BUILD instantiations only (round 6: the option is off by default — its
    // tests and the late kernel select those builds; the default kernels carry neither the record test nor its branches)
    constexpr bool REC = REC_BUILD && !TAIL && LAZY_KN == 0;
    float4 r0 = make_float4(0.f, 0.f, 0.f, 0.f), r1 = make_float4(0.f, 0.f, 0.f, __int_as_float(-1));
#pragma unroll 1
    for (int tj = 0;; ++tj) {  // (one trip unless TAIL)
    const LocalTid threadIdx = TAIL ? reloaded_tid() : LocalTid{::threadIdx.x};  // (icp_internal.h)
    const int lq = threadIdx.x, qi = query_of(lq);
    const int iter_now = in.iter + tj;  // (= the RegState's / the mailbox's iteration count while the loop is running)
    const unsigned gen_now = lead.gen + (unsigned)tj;
    if (lq < Q && need_load) {
        need_load = false;
        c = make_int4(-1, 0, -1, -1);
        r1.w = __int_as_float(-1);
        valid = qi < in.n;
        if (valid) {
            t4 = in.tgt[qi];
            valid = target_valid(t4.x, t4.y, t4.z, in.mode);
            if (!valid && !in.use_cache) in.nn_cache[qi] = make_int4(-1, 0, -1, -1);  // masked row: no neighbour, no seed
        }
        if (valid) {
            if (in.use_cache) {
                c = in.nn_cache[qi];
                if (REC && in.rec) {  // entry and record together: 48 coalesced bytes
                    r0 = in.rec[2 * (size_t)qi];
                    r1 = in.rec[2 * (size_t)qi + 1];
                }
                // the candidate set only where there is no record to try first (the iteration behind a search)
                if (c.x >= 0 && !(REC && in.rec && __float_as_int(r1.w) >= 0)) {
                    cq = g.pts[c.x & CACHE_POS_MASK];
                    cn = in.normals[c.x & CACHE_POS_MASK];  // speculative: needed on a hit only
                    if (c.z >= 0) {  // the other members of the candidate set the search left
                        cq2 = g.pts[c.z];
                        cn2 = in.normals[c.z];
                    }
                    if (c.w >= 0) {
                        cq3 = g.pts[c.w];
                        cn3 = in.normals[c.w];
                    }
                }
            } else if (in.frame_seed) {
                const int o = in.frame_seed[qi];
                if (o >= 0 && o < g.m) {
                    seed_o = o;
                    seed_sp = g.pos_of_orig[o];
                    cq = g.pts[seed_sp];
                }
            }
        }
    }
    // (the poses of the last CACHE_HIST iterations, for the cache test, were requested above: written by the solves of
    // EARLIER launches, plain loads, in flight before the wait below like everything else that does not need this pose)
    // ---- the pose: from the mailbox (lead launch) or from the RegState (classic launch)
    if (threadIdx.x < 4) ctl_s[threadIdx.x] = threadIdx.x == 0 ? vb : 0;
    if (dbg_global && threadIdx.x < 16) dbg_s[threadIdx.x] = 0;
    __syncthreads();
    if (TAIL && bx == 0) {  // (block-uniform) the lead: solve tj, published as generation gen_now, in front of its own share
        if (!tail_lead_step<THREADS>(lead, st, ap, reinterpret_cast<double*>(&cellstack[0][0]), carry_s, tj, gen_now, in.tail_rows,
                                     gx,
                                     (g.stamps && iter_now < 24) ? g.stamps + 4 * ((size_t)iter_now * 1024 + 1023) : nullptr))
            return;
    }
    if (lead.box) {
        if (threadIdx.x < BOX_USED) {
            const unsigned long long* p = box_granule(lead.box, gen_now, vb % BOX_REPLICAS, threadIdx.x);
            const long long t0 = wall_clock64();
            unsigned long long v;
            for (;;) {
                v = __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if ((unsigned)(v >> 32) == gen_now) break;
                if (wall_clock64() - t0 > lead.timeout_ticks) {
                    atomicAdd(&ctl_s[3], 1);
                    break;
                }
                __builtin_amdgcn_s_sleep(1);
            }
            const unsigned bits = (unsigned)(v & 0xffffffffull);
            if (threadIdx.x < 12) pose_s[threadIdx.x] = __uint_as_float(bits);
            else ctl_s[threadIdx.x - 11] = (int)bits;  // 12 -> done, 13 -> iteration
        }
    } else if (threadIdx.x < 12) {
        pose_s[threadIdx.x] = st->pose[threadIdx.x];
        if (threadIdx.x == 0) ctl_s[2] = st->iter;
    }
    if (threadIdx.x == 0) {
        nmiss = 0;
        nlazy = 0;
    }
    __syncthreads();

    if (ctl_s[3]) {  // never seen: the hand-off did not arrive within its wall-clock budget -> a loud error, not a hang
        if (threadIdx.x == 0) atomicAdd(&st->handoff_timeouts, 1);
        return;
    }
    if (ctl_s[1]) return;  // the loop is finished (block-uniform)
    // dev-only phase timestamps ("search_stats"): 4 x wall_clock64 (100 MHz) per block and iteration behind the 16 path
    // counters
    long long* stamps = nullptr;
    if (g.stamps && gx <= 1025 && vb < 1024 && iter_now < 24)
        stamps = g.stamps + 4 * ((size_t)iter_now * 1024 + vb);
    if (stamps && threadIdx.x == 0) stamps[0] = t_entry;
    // ---- phase A, second half: one lane per query
    if (lq < Q) {
        float row[9];
#pragma unroll
        for (int k = 0; k < 9; ++k) row[k] = 0.f;
        if (valid) {
            float px, py, pz;
            transform_point(pose_s, t4.x, t4.y, t4.z, px, py, pz);
            bool hit = false;
            float seed_d2 = INFINITY;
            int seed_idx = 0x7fffffff, seed_pos = -1;
            if (in.use_cache) {
                const float margin = (!TAIL || iter_now == in.refresh_at) ? in.refresh_margin : 0.f;
                if (REC && in.rec && c.x >= 0 && __float_as_int(r1.w) >= 0) {  // the record first
                    const int k2 = __float_as_int(r1.w), age2 = iter_now - k2;
                    if (age2 >= 1 && age2 <= CACHE_HIST) {
                        const float dx = r0.x - px, dy = r0.y - py, dz = r0.z - pz;
                        const float d2 = fmaf(dz, dz, fmaf(dy, dy, dx * dx));
                        float ox, oy, oz;  // where the target was when the record's bound was formed
                        transform_point(hist_s[k2 % CACHE_HIST], t4.x, t4.y, t4.z, ox, oy, oz);
                        const float mx = px - ox, my = py - oy, mz = pz - oz;
                        const float delta = sqrt_1ulp(fmaf(mz, mz, fmaf(my, my, mx * mx))) * 1.000001f + 1e-7f;
                        hit = sqrt_1ulp(d2) * 1.000001f < r0.w - delta - margin;
                    }
                    if (hit) {
                        point_to_plane_row(px, py, pz, r0.x, r0.y, r0.z, r1.x, r1.y, r1.z, ap.scheme, ap.sigma, row);
                    } else {  // not certified by the record alone: the entry's candidate set after all (a dependent round trip)
                        cq = g.pts[c.x & CACHE_POS_MASK];
                        cn = in.normals[c.x & CACHE_POS_MASK];
                        if (c.z >= 0) {
                            cq2 = g.pts[c.z];
                            cn2 = in.normals[c.z];
                        }
                        if (c.w >= 0) {
                            cq3 = g.pts[c.w];
                            cn3 = in.normals[c.w];
                        }
                    }
                }
                if (!hit && c.x >= 0) {
                    const int k = (int)((unsigned)c.x >> CACHE_ITER_SHIFT), age = iter_now - k;  // searched `age` launches ago
                    float dx = cq.x - px, dy = cq.y - py, dz = cq.z - pz;
                    float d2 = fmaf(dz, dz, fmaf(dy, dy, dx * dx));
                    int hit_pos = c.x & CACHE_POS_MASK;
                    float4 wq = cq, wn = cn;  // the winner (the set itself stays as loaded: the tail keeps it for the next trip)
                    float others2 = INFINITY;  // squared distance of the nearest member of the set that is NOT the winner
                    // the candidate set: its nearest member is THE neighbour as long as everything else (>= L) stays farther
                    if (c.z >= 0) {
                        dx = cq2.x - px, dy = cq2.y - py, dz = cq2.z - pz;
                        const float e2 = fmaf(dz, dz, fmaf(dy, dy, dx * dx));
                        if (better(e2, __float_as_int(cq2.w), d2, __float_as_int(wq.w))) {
                            others2 = fminf(others2, d2);
                            d2 = e2;
                            wq = cq2;
                            wn = cn2;
                            hit_pos = c.z;
                        } else {
                            others2 = fminf(others2, e2);
                        }
                    }
                    if (c.w >= 0) {
                        dx = cq3.x - px, dy = cq3.y - py, dz = cq3.z - pz;
                        const float e2 = fmaf(dz, dz, fmaf(dy, dy, dx * dx));
                        if (better(e2, __float_as_int(cq3.w), d2, __float_as_int(wq.w))) {
                            others2 = fminf(others2, d2);
                            d2 = e2;
                            wq = cq3;
                            wn = cn3;
                            hit_pos = c.w;
                        } else {
                            others2 = fminf(others2, e2);
                        }
                    }
                    float outside = 0.f;  // lower bound on every map point outside the set, at THIS pose
                    if (age >= 1 && age <= CACHE_HIST) {
                        float ox, oy, oz;  // where the target was when its neighbour was searched
                        transform_point(hist_s[k % CACHE_HIST], t4.x, t4.y, t4.z, ox, oy, oz);
                        const float mx = px - ox, my = py - oy, mz = pz - oz;
                        const float delta = sqrt_1ulp(fmaf(mz, mz, fmaf(my, my, mx * mx))) * 1.000001f + 1e-7f;
                        outside = __int_as_float(c.y) - delta;
                        hit = sqrt_1ulp(d2) * 1.000001f < outside - margin;
                    }
                    if (REC && in.rec && hit) {
                        // the record of this hit: winner, normal, and what bounds every OTHER map point at this pose — the
                        // set's other members exactly (rounded down), everything outside the set by `outside`
                        const float bound = fminf(outside, sqrt_1ulp(others2) * 0.999999f);
                        in.rec[2 * (size_t)qi] = make_float4(wq.x, wq.y, wq.z, bound);
                        in.rec[2 * (size_t)qi + 1] = make_float4(wn.x, wn.y, wn.z, __int_as_float(iter_now));
                    }
                    if (hit && LAZY && wn.w != 1.f) {  // the neighbour's normal has not been estimated yet: phase N forms the row
                        const int k2 = atomicAdd(&nlazy, 1);
                        lazy_p[k2] = make_float4(px, py, pz, __int_as_float(lq));
                        lazy_pos[k2] = hit_pos;
                        need_load = true;
                    } else if (hit) {  // (nothing to write: the entry stays as the search left it)
                        point_to_plane_row(px, py, pz, wq.x, wq.y, wq.z, wn.x, wn.y, wn.z, ap.scheme, ap.sigma, row);
                    } else if (in.use_cache > 1) {  // a candidate all the same: it seeds the search
                        seed_d2 = d2;
                        seed_idx = __float_as_int(wq.w);
                        seed_pos = hit_pos;
                    }
                }
            } else if (seed_sp >= 0) {
                const float dx = cq.x - px, dy = cq.y - py, dz = cq.z - pz;
                seed_d2 = fmaf(dz, dz, fmaf(dy, dy, dx * dx));
                seed_idx = seed_o;
                seed_pos = seed_sp;
            }
            if (!hit) {
                need_load = true;  // (TAIL: the search below rewrites this query's entry)
                const int k = atomicAdd(&nmiss, 1);
                miss_p[k] = make_float4(px, py, pz, __int_as_float(lq));
                miss_seed[k] = make_int4(__float_as_int(seed_d2), seed_idx, seed_pos, 0);
            }
        }
#pragma unroll
        for (int k = 0; k < 9; ++k) rowbuf[padded_row(lq)][k] = row[k];
    }
    __syncthreads();
    if (stamps && threadIdx.x == 0) {
        stamps[1] = wall_clock64() | ((long long)nmiss << 48);  // the block's miss count rides in the top bits
        if (g.dbg) {
            atomicAdd(&g.dbg[6], nmiss);
            atomicAdd(&g.dbg[8 + min(iter_now, 21) / 3], nmiss);  // misses by iteration: 0-2, 3-5, .., 18-20
        }
    }
    // ---- phase B0 (round 4): every miss by ONE lane — or, where the workgroup has few of them, by 2 or 8 neighbouring
    // lanes (search_ball_lane); what does not fit its pattern (own cell empty, a ball that leaves the 2x2x2 block, more than
    // 256 candidates, four candidates within a key's resolution) goes back on the list for the generic paths below
    // the row of a query whose neighbour a search has just named — or (LAZY) the query on the waiting list of phase N
    // what a search leaves for its query: the entry — and no hit record (the first hit on the new entry forms one)
    const auto store_entry = [&](int slot, const int4 e) {
        const int q = query_of(slot);
        in.nn_cache[q] = e;
        if (REC && in.rec) in.rec[2 * (size_t)q + 1] = make_float4(0.f, 0.f, 0.f, __int_as_float(-1));
    };
    const auto row_or_wait = [&](int slot, const float4 mp, const float4 q, const float4 nn, int pos) {
        if (LAZY && nn.w != 1.f) {
            const int k2 = atomicAdd(&nlazy, 1);
            lazy_p[k2] = make_float4(mp.x, mp.y, mp.z, __int_as_float(slot));
            lazy_pos[k2] = pos;
            return;
        }
        float row[9];
        point_to_plane_row(mp.x, mp.y, mp.z, q.x, q.y, q.z, nn.x, nn.y, nn.z, ap.scheme, ap.sigma, row);
#pragma unroll
        for (int k = 0; k < 9; ++k) rowbuf[padded_row(slot)][k] = row[k];
    };
    // (round 6) a workgroup WITHOUT a miss — nearly every one of a late launch — goes straight to the summing: the rows of its
    // hits are behind the barrier above; the phases below would cost it four dependent reads of the list's counter and a barrier
    const bool any_miss = nmiss > 0;  // block-uniform
    if (any_miss) {
    const int ball_lpq = !in.ball ? 0
                         : (in.ball_lanes >= 8 && nmiss * 8 <= THREADS)   ? 8
                         : (in.ball_lanes >= 4 && nmiss * 4 <= THREADS)   ? 4
                         : (in.ball_lanes >= 2 && nmiss * 2 <= THREADS)   ? 2
                         : (in.ball_lanes >= 2 || nmiss > in.wave_misses) ? 1
                                                                          : 0;  // ("ball_lanes" 1: a handful of misses -> B1)
    const auto ball_phase = [&](auto lpq_c) {
        constexpr int LPQ = decltype(lpq_c)::value;
        // (the 128-query shape is built for 64 registers: one group of four in flight; more lanes per query, fewer groups each)
        constexpr int W = Q == IT_QUERIES ? 1 : (LPQ == 1 ? BALL_W : (LPQ == 2 ? 2 : 1));
        const int listed = nmiss;
        const int m = (int)threadIdx.x / LPQ, sub = (int)threadIdx.x % LPQ;
        const bool mine = m < listed;  // (uniform over the LPQ lanes of a query)
        float4 mp = make_float4(0.f, 0.f, 0.f, 0.f);
        int4 ms = make_int4(0, 0, 0, 0);
        if (mine) {
            mp = miss_p[m];
            ms = miss_seed[m];
        }
        __syncthreads();
        if (threadIdx.x == 0) nmiss = 0;
        __syncthreads();
        if (mine) {
            int p0, p1, p2, sp;
            float L;
            const bool ok = search_ball_lane<W, LPQ>(g, mp.x, mp.y, mp.z, __int_as_float(ms.x), in.ball_max,
                                                     &cellstack[0][threadIdx.x], THREADS, sub, p0, p1, p2, L, sp, in.ball_empty != 0);
            if (sub == 0) {
                if (ok) {
                    const int lq2 = __float_as_int(mp.w);
                    const float4 q = g.pts[p0];
                    const float4 nn = in.normals[p0];
                    store_entry(lq2, make_int4(pack_cache(p0, iter_now), __float_as_int(L * 0.999999f), p1, p2));
                    row_or_wait(lq2, mp, q, nn, p0);
                    if (g.dbg) atomicAdd(&g.dbg[0], 1);
                } else {
                    if (sp >= 0) {  // a better seed than the one it came with: the nearest point of its own cell (exact distance)
                        const float4 q = g.pts[sp];
                        const float dx = q.x - mp.x, dy = q.y - mp.y, dz = q.z - mp.z;
                        ms = make_int4(__float_as_int(fmaf(dz, dz, fmaf(dy, dy, dx * dx))), __float_as_int(q.w), sp, 0);
                    }
                    const int k = atomicAdd(&nmiss, 1);
                    miss_p[k] = mp;
                    miss_seed[k] = ms;
                }
            }
        }
        __syncthreads();
    };
    // (the resident tail runs the late iterations — a handful of misses chip-wide: the whole-wave search, then the 4-lane
    // groups; the ball search and the 16-lane search are not compiled into it: their registers would be live around its loop)
    if (!TAIL && nmiss > 0) {  // block-uniform
        if (ball_lpq == 8) ball_phase(std::integral_constant<int, 8>{});
        else if (ball_lpq == 4) ball_phase(std::integral_constant<int, 4>{});
        else if (ball_lpq == 2) ball_phase(std::integral_constant<int, 2>{});
        else if (ball_lpq == 1) ball_phase(std::integral_constant<int, 1>{});
    }
    if (stamps && !dbg_global && threadIdx.x == 0) {  // dev ("search_stats" 2): what the ball search left, and when
        dbg_s[12] = nmiss;
        dbg_s[13] = (int)(wall_clock64() - (stamps[1] & ((1ll << 48) - 1)));
    }
    // ---- phase BF (round 4, item 48): a few dozen queries left — too many for a wave each, too few to fill the workgroup
    // four lanes each: 16 lanes per query, THREADS / 16 queries at a time (search_far_w); nothing is left behind
    if (!TAIL && Q != IT_QUERIES && in.far_lanes >= 16 && nmiss > in.far_min && nmiss <= in.far_max) {  // block-uniform
        constexpr int W = 16;
        const int listed = nmiss, sub = (int)threadIdx.x % W;
        __syncthreads();
        if (threadIdx.x == 0) nmiss = 0;
        for (int m = (int)threadIdx.x / W; m < listed; m += THREADS / W) {  // group-uniform
            const float4 mp = miss_p[m];
            const int4 ms = miss_seed[m];
            const Best b = search_far_w<W>(g, mp.x, mp.y, mp.z, sub, in.max_rings, &cellstack[0][threadIdx.x - sub], THREADS,
                                           __int_as_float(ms.x), ms.y, ms.z);
            if (sub == 0) {
                const int lq = __float_as_int(mp.w);
                store_entry(lq, make_int4(pack_cache(b.pos, iter_now), __float_as_int(sqrtf(b.second) * 0.999999f), -1, -1));
                if (b.pos >= 0) {
                    const float4 q = g.pts[b.pos];
                    const float4 nn = in.normals[b.pos];
                    row_or_wait(lq, mp, q, nn, b.pos);
                }
            }
        }
        __syncthreads();
    }
    // ---- phase B1: few misses (the late iterations): a whole wave per miss — the latency of the slowest search is the
    // duration of the launch.  Whatever the wave path does not settle stays on the list for B2.
    if (nmiss > 0 && nmiss <= in.wave_misses) {  // block-uniform
        const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, listed = nmiss;
        // (the cell stacks are idle until B2: the wave's own 64 columns — row 0 for the ring-1 lists, all seven rows for a
        // search that goes on to the coarse level)
        int* wl = reinterpret_cast<int*>(&cellstack[0][wave * 64]);
        for (int m = wave; m < listed; m += THREADS / 64) {
            const float4 mp = miss_p[m];
            const int4 ms = miss_seed[m];
            Best b, r2, r3;
            if (!search_rows_wave(g, mp.x, mp.y, mp.z, lane, in.max_rings, wl, __int_as_float(ms.x), ms.y, ms.z, b, &r2,
                                  &r3)) {
                // not settled by the fine rings 1-2: the coarse level by the same wave (the resident tail leaves it to the
                // 4-lane groups of B2: the 64-lane lists are not compiled into its loop)
                if constexpr (TAIL) {
                    continue;
                } else {
                    if (g.dbg && lane == 0) atomicAdd(&g.dbg[3], 1);
                    b = search_coarse_w<64>(g, mp.x, mp.y, mp.z, lane, &cellstack[0][wave * 64], THREADS,
                                            __int_as_float(ms.x), ms.y, ms.z);
                    r2.pos = r3.pos = -1;
                }
            }
            if (lane == 0) {
                const int lq = __float_as_int(mp.w);
                // the runner-up and the third ride along: L then bounds everything but the set
                const bool pair = r2.pos >= 0 && b.pos >= 0, triple = pair && r3.pos >= 0;
                store_entry(lq, make_int4(pack_cache(b.pos, iter_now),
                                          __float_as_int(sqrtf(triple ? r3.second : (pair ? r2.second : b.second)) * 0.999999f),
                                          pair ? r2.pos : -1, triple ? r3.pos : -1));
                if (b.pos >= 0) {
                    const float4 q = g.pts[b.pos];
                    const float4 nn = in.normals[b.pos];
                    row_or_wait(lq, mp, q, nn, b.pos);
                }
                miss_seed[m].w = 1;  // settled
            }
        }
        __syncthreads();
        if (threadIdx.x == 0) {
            int k = 0;
            for (int m = 0; m < listed; ++m)
                if (!miss_seed[m].w) {
                    miss_p[k] = miss_p[m];
                    miss_seed[k] = miss_seed[m];
                    ++k;
                }
            nmiss = k;
        }
        __syncthreads();
    }
    // ---- phase B2: the misses, 4 lanes each, dense over the block's groups
    {
        const int sub = threadIdx.x & 3;
        // (a loop with a single trip in the 512-thread build — written as one there: the loop form costs it spills)
        for (int grp = threadIdx.x >> 2; grp < nmiss; grp += THREADS / 4) {  // group-uniform
            const float4 mp = miss_p[grp];
            const int4 ms = miss_seed[grp];
            const Near3 near = search_rows_group(g, mp.x, mp.y, mp.z, sub, in.max_rings, &cellstack[0][threadIdx.x],
                                                 THREADS, __int_as_float(ms.x), ms.y, ms.z);
            const Best& b = near.b;
            if (sub == 0) {
                const int lq = __float_as_int(mp.w);
                store_entry(lq, make_int4(pack_cache(b.pos, iter_now), __float_as_int(sqrtf(b.second) * 0.999999f), near.pos1,
                                          near.pos2));
                if (b.pos >= 0) {
                    const float4 q = g.pts[b.pos];
                    const float4 nn = in.normals[b.pos];
                    row_or_wait(lq, mp, q, nn, b.pos);
                }
            }
            if (THREADS >= 4 * Q) break;  // a group for every query: one trip
        }
    }
    __syncthreads();
    }  // (any_miss)
    if (LAZY && nlazy > 0) {  // ---- phase N (block-uniform): the normals the queries of this workgroup wait for
        const int listed = nlazy, wave = threadIdx.x >> 6, lane = threadIdx.x & 63, sub = threadIdx.x & 3;
        // (another query of this workgroup may wait for the same map point — estimated twice, the same bits; so may another
        // workgroup of this launch)
        // four lanes per waiting query: ring 1 from the cell's neighbour row (estimate_cov, the round-2 kernel's group: a
        // whole wave per query — 64-lane merges of sorted lists — took 60 us each); what ring 1 does not certify is marked
        // and finished by whole waves below (the stragglers' path of the eager kernels)
        if (threadIdx.x == 0) nmiss = 0;  // (the miss list is through: its counter numbers the stragglers)
        __syncthreads();
        for (int m = threadIdx.x >> 2; m < listed; m += THREADS / 4) {  // group-uniform
            if (!lazy_cov_group<LAZY_KN>(g, lazy_pos[m], sub, lazy_cov[m], &cellstack[0][threadIdx.x], THREADS) && sub == 0)
                miss_seed[atomicAdd(&nmiss, 1)].x = m;
        }
        __syncthreads();
        int* wl = reinterpret_cast<int*>(&cellstack[0][0]) + wave * 128;
        for (int k = wave; k < nmiss; k += THREADS / 64) {
            const int m = miss_seed[k].x;
            lazy_cov_wave<LAZY_KN>(g, lazy_pos[m], lane, in.knn_rings, wl, lazy_cov[m]);
        }
        __syncthreads();
        for (int m = threadIdx.x; m < listed; m += THREADS) {  // the eigen-solves, one lane per waiting query
            const int pos = lazy_pos[m];
            normal_from_cov(lazy_cov[m], pos, in.normals_rw, in.nflag);
            const float4 nn = in.normals_rw[pos], q = g.pts[pos], mp = lazy_p[m];
            float row[9];
            point_to_plane_row(mp.x, mp.y, mp.z, q.x, q.y, q.z, nn.x, nn.y, nn.z, ap.scheme, ap.sigma, row);
            const int slot = __float_as_int(mp.w);
#pragma unroll
            for (int k = 0; k < 9; ++k) rowbuf[padded_row(slot)][k] = row[k];
        }
        if (threadIdx.x == 0) atomicAdd((unsigned long long*)&st->normals_computed, (unsigned long long)listed);
        __syncthreads();
    }
    if (stamps && threadIdx.x == 0) stamps[2] = wall_clock64();
    if (TAIL) {  // the super-row goes to the lead of THIS launch: tagged granules, no fence (solve_device.h)
        const double v = block_reduce_rows<Q, true, true>(rowbuf, part, nullptr, vb);
        if (threadIdx.x < NEQ) tagged_row_store(in.tail_rows, vb, threadIdx.x, gen_now, v);
    } else {
        block_reduce_rows<Q, false, true>(rowbuf, part, in.partials, vb);
    }
    if (stamps && threadIdx.x == 0) stamps[3] = wall_clock64();
    if (stamps && !dbg_global && threadIdx.x == 0) {  // (top bits of stamps 3 / 0: queries left to B1 / B2, ticks of B0)
        stamps[3] |= (long long)min(dbg_s[12], 65535) << 48;
        stamps[0] |= (long long)min(dbg_s[13], 65535) << 48;
    }
    if (dbg_global) {  // dev: counters to the context's totals; per workgroup: beyond ring 1 | own cell empty | coarse level
        if (threadIdx.x < 16 && dbg_s[threadIdx.x]) atomicAdd(&dbg_global[threadIdx.x], dbg_s[threadIdx.x]);
        if (stamps && threadIdx.x == 0) {
            stamps[2] |= (long long)dbg_s[2] << 48;
            stamps[3] |= (long long)dbg_s[5] << 48;
            stamps[0] |= (long long)dbg_s[3] << 48;
        }
    }
    if (!TAIL) break;
    if (tj + 1 >= in.tail_iters) {  // behind the last iteration: the solve a k_sum_solve launch used to carry out
        if (bx == 0) {
            __syncthreads();
            tail_lead_step<THREADS>(lead, st, ap, reinterpret_cast<double*>(&cellstack[0][0]), carry_s, tj + 1, gen_now + 1u,
                                    in.tail_rows, gx,
                                    (g.stamps && iter_now + 1 < 24) ? g.stamps + 4 * ((size_t)(iter_now + 1) * 1024 + 1023) : nullptr);
        }
        break;
    }
    // the next iteration of the tail: its pose history entry is the pose this one ran with (slot iter % CACHE_HIST held
    // iteration iter - CACHE_HIST, which this iteration's cache test may still have read: hence behind the barriers above)
    if (threadIdx.x < 12) hist_s[iter_now % CACHE_HIST][threadIdx.x] = pose_s[threadIdx.x];
    if (g.stamps) t_entry = wall_clock64();
    __syncthreads();  // (pose_s, ctl_s, rowbuf, part: read above, rewritten by the next trip)
    }
}

template <int MINW, int THREADS, int Q, bool STATS = false, bool TAIL = false, int LAZY_KN = 0, bool REC_BUILD = false>
__global__ __launch_bounds__(THREADS, MINW) void k_iterate_compact(GridView g, IterInputs in,
                                                                   RegState* __restrict__ st, AlignParams ap,
                                                                   LeadArgs lead) {
    // In a lead launch (LeadArgs) workgroup 0 is the lead (the resident tail's lead is its workgroup 0 too, but takes its
    // share of the queries as well: tail_lead_step inside the body)
    const int lead_blocks = (lead.box && !TAIL) ? lead.solve : 0;
    iterate_body<THREADS, Q, STATS, TAIL, LAZY_KN, REC_BUILD>(g, in, st, ap, lead, (int)blockIdx.x < lead_blocks, (int)blockIdx.x,
                                                              lead_blocks, (int)blockIdx.x, (int)gridDim.x);
}

// ---------------------------------------------------------------------------------------------------------------------
// B sequences per launch ("a batched registration").  One sequence is a chain of dependent, latency-bound
// launches: a late iteration streams 4.7 MB in ~3 us of work behind ~5 us of the lead's sum -> solve -> publish.  Here B
// independent registrations (B contexts: B maps, B scans, B states) advance by one iteration in ONE launch: the arguments
// of every sequence — the very structs the single launch takes by value — sit in a descriptor table in device memory
// (IterateDesc[B], written by the host once per frame for all its launches), a workgroup finds its sequence from its number
// and reads its descriptor with scalar loads.  Layout of the grid: LEAD_SLOTS lead workgroups first (lead b solves the
// previous iteration of sequence b and publishes its pose; slots beyond B return at once: the region is a multiple of 8
// so that the XCD of a work workgroup stays its number mod 8, logical_block), then per_seq workgroups per sequence,
// sequence after sequence.  The hardware dispatches in ascending order: every lead is placed before any workgroup that
// polls for it, B solves overlap, and the streaming of sequence b runs beside the solve of sequence b + 1.
// Per sequence the body is the single launch's, same arguments, same order of every sum: the same bits.
// ---------------------------------------------------------------------------------------------------------------------
template <typename T>
__device__ __forceinline__ T load_descriptor(const T* table, int index) {
    // through the CONSTANT address space: the table is not written while the launch runs, so every word may be fetched by
    // a scalar load wherever the body first needs it (and fetched again instead of being kept in registers)
    static_assert(sizeof(T) % 4 == 0, "descriptor words");
    typedef const int __attribute__((address_space(4))) * cwords;
    cwords p = (cwords)(const void*)(table + index);
    int w[sizeof(T) / 4];
#pragma unroll
    for (int i = 0; i < (int)(sizeof(T) / 4); ++i) w[i] = p[i];
    T out;
    __builtin_memcpy(&out, w, sizeof(T));
    return out;
}

template <int MINW, int THREADS, int Q, bool REC_BUILD = false>
__global__ __launch_bounds__(THREADS, MINW) void k_iterate_batch(const IterateDesc* __restrict__ table, int nseq, int per_seq) {
    const int p = (int)blockIdx.x;
    int seq, pb;
    bool is_lead = false;
    if (p < BATCH_LEAD_SLOTS) {  // block-uniform
        if (p >= nseq) return;
        seq = p;
        pb = 0;
        is_lead = true;
    } else {
        seq = (p - BATCH_LEAD_SLOTS) / per_seq;
        pb = (p - BATCH_LEAD_SLOTS) - seq * per_seq;
    }
    const IterateDesc d = load_descriptor(table, seq);
    if (is_lead ? !(d.lead.box && d.lead.solve) : pb >= d.blocks) return;  // (no solve pending for this sequence / a shorter scan)
    iterate_body<THREADS, Q, false, false, 0, REC_BUILD>(d.g, d.in, d.st, d.ap, d.lead, is_lead, pb, 0, pb, d.blocks);
}

// ---------------------------------------------------------------------------------------------------------------------
// The LATE kernel (round 6).  From the third or fourth iteration on nearly every query is settled by its hit record — one
// lane, 48 coalesced bytes, a dozen registers — yet the launches ran the kernel above: 128 registers and 69 KB of LDS per
// workgroup for the searches it carries, i.e. two workgroups (16 waves) per CU, four rounds of workgroups per batched launch,
// every round a latency chain (loads -> pose -> rows -> barrier -> block sums) nothing overlaps with.  This kernel is the same
// iteration built for those launches: phase A is the generic kernel's (record first, then the entry's candidate set, same
// tests, same record written on a set hit), what neither settles is listed and searched by a whole wave each (search_rows_wave, then the coarse level by the same wave:
// exact, every path), two waves at a time.  64 registers and 33 KB of LDS: four workgroups (32 waves) per CU.  Same rows for the
// same queries in the same slots, same block sums: the same bits as the generic kernel.
// ---------------------------------------------------------------------------------------------------------------------
static constexpr int LATE_MISS_CAP = 64;
static constexpr int LATE_SEARCH_WAVES = 2;

template <int THREADS>
__device__ __forceinline__ void iterate_late_body(GridView g, IterInputs in, RegState* __restrict__ st, AlignParams ap,
                                                  LeadArgs lead, const bool is_lead, const int pb, const int off) {
    constexpr int Q = THREADS;
    g.dbg = nullptr;
    g.stamps = nullptr;
    __shared__ float rowbuf[Q][9];
    __shared__ double part[Q / 32][NEQ];
    __shared__ int2 cellstack[7][64 * LATE_SEARCH_WAVES];
    __shared__ float4 miss_p[LATE_MISS_CAP];
    __shared__ int4 miss_seed[LATE_MISS_CAP];
    __shared__ unsigned short over_q[Q];  // the slots of the misses beyond LATE_MISS_CAP (searched without a seed: never seen in a converging loop)
    __shared__ int nmiss;
    __shared__ float pose_s[12];
    __shared__ float hist_s[CACHE_HIST][12];
    __shared__ int ctl_s[4];
    static_assert(sizeof(rowbuf) >= (32 * NEQ + NEQ + 1) * sizeof(double), "the lead's scratch lives in the row buffer");
    static_assert(Q <= 65536, "slots are listed as 16-bit numbers");
    if (lead.box) {
        if (is_lead) {  // block-uniform
            lead_solve<THREADS>(lead, st, ap, reinterpret_cast<double*>(&rowbuf[0][0]), nullptr);
            return;
        }
    } else if (st->done) {
        return;
    }
    const int vb = logical_block(in, pb, off);
    const int lq = threadIdx.x;
    const int qi = in.chunk_stride <= 0 ? vb * Q + lq
                                        : (vb + (lq / IT_QUERIES) * in.chunk_stride) * IT_QUERIES + (lq % IT_QUERIES);
    if ((int)threadIdx.x < 12 * CACHE_HIST) {
        const int e = threadIdx.x / 12, j = in.iter - 1 - e;  // iteration j, most recent first
        if (j >= 0) hist_s[j % CACHE_HIST][threadIdx.x % 12] = in.pose_hist[(size_t)j * 12 + threadIdx.x % 12];
    }
    // ---- what does not depend on the pose: target and hit record (48 coalesced bytes), in flight while the lead solves
    bool valid = qi < in.n;
    float4 t4 = make_float4(0.f, 0.f, 0.f, 0.f), r0 = t4, r1 = make_float4(0.f, 0.f, 0.f, __int_as_float(-1));
    if (valid) {
        t4 = in.tgt[qi];
        valid = target_valid(t4.x, t4.y, t4.z, in.mode);
    }
    if (valid) {
        r0 = in.rec[2 * (size_t)qi];
        r1 = in.rec[2 * (size_t)qi + 1];
    }
    // ---- the pose: from the mailbox (lead launch) or from the RegState (classic launch)
    if (threadIdx.x < 4) ctl_s[threadIdx.x] = threadIdx.x == 0 ? vb : 0;
    __syncthreads();
    if (lead.box) {
        if (threadIdx.x < BOX_USED) {
            const unsigned long long* p = box_granule(lead.box, lead.gen, vb % BOX_REPLICAS, threadIdx.x);
            const long long t0 = wall_clock64();
            unsigned long long v;
            for (;;) {
                v = __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if ((unsigned)(v >> 32) == lead.gen) break;
                if (wall_clock64() - t0 > lead.timeout_ticks) {
                    atomicAdd(&ctl_s[3], 1);
                    break;
                }
                __builtin_amdgcn_s_sleep(1);
            }
            const unsigned bits = (unsigned)(v & 0xffffffffull);
            if (threadIdx.x < 12) pose_s[threadIdx.x] = __uint_as_float(bits);
            else ctl_s[threadIdx.x - 11] = (int)bits;  // 12 -> done, 13 -> iteration
        }
    } else if (threadIdx.x < 12) {
        pose_s[threadIdx.x] = st->pose[threadIdx.x];
        if (threadIdx.x == 0) ctl_s[2] = st->iter;
    }
    if (threadIdx.x == 0) nmiss = 0;
    __syncthreads();
    if (ctl_s[3]) {  // the hand-off did not arrive within its wall-clock budget -> a loud error, not a hang
        if (threadIdx.x == 0) atomicAdd(&st->handoff_timeouts, 1);
        return;
    }
    if (ctl_s[1]) return;  // the loop is finished (block-uniform)
    const int iter_now = in.iter;
    // ---- phase A: the record, then the entry's candidate set (the generic kernel's tests, word for word)
    {
        float row[9];
#pragma unroll
        for (int k = 0; k < 9; ++k) row[k] = 0.f;
        if (valid) {
            float px, py, pz;
            transform_point(pose_s, t4.x, t4.y, t4.z, px, py, pz);
            bool hit = false;
            float seed_d2 = INFINITY;
            int seed_idx = 0x7fffffff, seed_pos = -1;
            const float margin = in.refresh_margin;
            const int k2 = __float_as_int(r1.w), age2 = iter_now - k2;
            if (k2 >= 0 && age2 >= 1 && age2 <= CACHE_HIST) {
                const float dx = r0.x - px, dy = r0.y - py, dz = r0.z - pz;
                const float d2 = fmaf(dz, dz, fmaf(dy, dy, dx * dx));
                float ox, oy, oz;
                transform_point(hist_s[k2 % CACHE_HIST], t4.x, t4.y, t4.z, ox, oy, oz);
                const float mx = px - ox, my = py - oy, mz = pz - oz;
                const float delta = sqrt_1ulp(fmaf(mz, mz, fmaf(my, my, mx * mx))) * 1.000001f + 1e-7f;
                hit = sqrt_1ulp(d2) * 1.000001f < r0.w - delta - margin;
                if (hit) point_to_plane_row(px, py, pz, r0.x, r0.y, r0.z, r1.x, r1.y, r1.z, ap.scheme, ap.sigma, row);
            }
            if (!hit) {
                const int4 c = in.nn_cache[qi];
                if (c.x >= 0) {
                    const int k = (int)((unsigned)c.x >> CACHE_ITER_SHIFT), age = iter_now - k;
                    int hit_pos = c.x & CACHE_POS_MASK;
                    float4 wq = g.pts[hit_pos];
                    float4 cq2 = wq, cq3 = wq;
                    if (c.z >= 0) cq2 = g.pts[c.z];
                    if (c.w >= 0) cq3 = g.pts[c.w];
                    float dx = wq.x - px, dy = wq.y - py, dz = wq.z - pz;
                    float d2 = fmaf(dz, dz, fmaf(dy, dy, dx * dx));
                    float others2 = INFINITY;
                    if (c.z >= 0) {
                        dx = cq2.x - px, dy = cq2.y - py, dz = cq2.z - pz;
                        const float e2 = fmaf(dz, dz, fmaf(dy, dy, dx * dx));
                        if (better(e2, __float_as_int(cq2.w), d2, __float_as_int(wq.w))) {
                            others2 = fminf(others2, d2);
                            d2 = e2;
                            wq = cq2;
                            hit_pos = c.z;
                        } else {
                            others2 = fminf(others2, e2);
                        }
                    }
                    if (c.w >= 0) {
                        dx = cq3.x - px, dy = cq3.y - py, dz = cq3.z - pz;
                        const float e2 = fmaf(dz, dz, fmaf(dy, dy, dx * dx));
                        if (better(e2, __float_as_int(cq3.w), d2, __float_as_int(wq.w))) {
                            others2 = fminf(others2, d2);
                            d2 = e2;
                            wq = cq3;
                            hit_pos = c.w;
                        } else {
                            others2 = fminf(others2, e2);
                        }
                    }
                    float outside = 0.f;
                    if (age >= 1 && age <= CACHE_HIST) {
                        float ox, oy, oz;  // where the target was when its neighbour was searched
                        transform_point(hist_s[k % CACHE_HIST], t4.x, t4.y, t4.z, ox, oy, oz);
                        const float mx = px - ox, my = py - oy, mz = pz - oz;
                        const float delta = sqrt_1ulp(fmaf(mz, mz, fmaf(my, my, mx * mx))) * 1.000001f + 1e-7f;
                        outside = __int_as_float(c.y) - delta;
                        hit = sqrt_1ulp(d2) * 1.000001f < outside - margin;
                    }
                    if (hit) {
                        const float4 wn = in.normals[hit_pos];
                        const float bound = fminf(outside, sqrt_1ulp(others2) * 0.999999f);
                        in.rec[2 * (size_t)qi] = make_float4(wq.x, wq.y, wq.z, bound);
                        in.rec[2 * (size_t)qi + 1] = make_float4(wn.x, wn.y, wn.z, __int_as_float(iter_now));
                        point_to_plane_row(px, py, pz, wq.x, wq.y, wq.z, wn.x, wn.y, wn.z, ap.scheme, ap.sigma, row);
                    } else if (in.use_cache > 1) {  // a candidate all the same: it seeds the search
                        seed_d2 = d2;
                        seed_idx = __float_as_int(wq.w);
                        seed_pos = hit_pos;
                    }
                }
            }
            if (!hit) {
                const int k = atomicAdd(&nmiss, 1);
                if (k < LATE_MISS_CAP) {
                    miss_p[k] = make_float4(px, py, pz, __int_as_float(lq));
                    miss_seed[k] = make_int4(__float_as_int(seed_d2), seed_idx, seed_pos, 0);
                } else {
                    over_q[k - LATE_MISS_CAP] = (unsigned short)lq;
                }
            }
        }
#pragma unroll
        for (int k = 0; k < 9; ++k) rowbuf[lq][k] = row[k];
    }
    // ---- the searches: a whole wave per listed query, LATE_SEARCH_WAVES at a time (the first LATE_MISS_CAP with the seed
    // phase A found, the rest — never seen in a converging loop — from their slot alone: target reloaded, no seed)
    __syncthreads();
    {
        const int total = nmiss;  // block-uniform
        const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
        if (total > 0 && wave < LATE_SEARCH_WAVES) {
            int* wl = reinterpret_cast<int*>(&cellstack[0][wave * 64]);
            for (int m = wave; m < total; m += LATE_SEARCH_WAVES) {
                float4 mp;
                int4 ms;
                if (m < LATE_MISS_CAP) {
                    mp = miss_p[m];
                    ms = miss_seed[m];
                } else {
                    const int slot = (int)over_q[m - LATE_MISS_CAP];
                    const int q = in.chunk_stride <= 0 ? vb * Q + slot
                                                       : (vb + (slot / IT_QUERIES) * in.chunk_stride) * IT_QUERIES + (slot % IT_QUERIES);
                    const float4 t = in.tgt[q];
                    transform_point(pose_s, t.x, t.y, t.z, mp.x, mp.y, mp.z);
                    mp.w = __int_as_float(slot);
                    ms = make_int4(__float_as_int(INFINITY), 0x7fffffff, -1, 0);
                }
                Best b, r2, r3;
                if (!search_rows_wave(g, mp.x, mp.y, mp.z, lane, in.max_rings, wl, __int_as_float(ms.x), ms.y, ms.z, b, &r2, &r3)) {
                    b = search_coarse_w<64>(g, mp.x, mp.y, mp.z, lane, &cellstack[0][wave * 64], 64 * LATE_SEARCH_WAVES,
                                            __int_as_float(ms.x), ms.y, ms.z);
                    r2.pos = r3.pos = -1;
                }
                if (lane == 0) {
                    const int slot = __float_as_int(mp.w);
                    const int q = in.chunk_stride <= 0 ? vb * Q + slot
                                                       : (vb + (slot / IT_QUERIES) * in.chunk_stride) * IT_QUERIES + (slot % IT_QUERIES);
                    const bool pair = r2.pos >= 0 && b.pos >= 0, triple = pair && r3.pos >= 0;
                    in.nn_cache[q] = make_int4(pack_cache(b.pos, iter_now),
                                               __float_as_int(sqrtf(triple ? r3.second : (pair ? r2.second : b.second)) * 0.999999f),
                                               pair ? r2.pos : -1, triple ? r3.pos : -1);
                    in.rec[2 * (size_t)q + 1] = make_float4(0.f, 0.f, 0.f, __int_as_float(-1));  // (the first hit on the new entry forms a record)
                    if (b.pos >= 0) {
                        const float4 qp = g.pts[b.pos];
                        const float4 nn = in.normals[b.pos];
                        float row[9];
                        point_to_plane_row(mp.x, mp.y, mp.z, qp.x, qp.y, qp.z, nn.x, nn.y, nn.z, ap.scheme, ap.sigma, row);
#pragma unroll
                        for (int k = 0; k < 9; ++k) rowbuf[slot][k] = row[k];
                    }
                }
            }
        }
    }
    __syncthreads();
    block_reduce_rows<Q>(rowbuf, part, in.partials, vb);
}

template <int MINW, int THREADS>
__global__ __launch_bounds__(THREADS, MINW) void k_iterate_late(GridView g, IterInputs in, RegState* __restrict__ st,
                                                                AlignParams ap, LeadArgs lead) {
    const int lead_blocks = lead.box ? lead.solve : 0;
    iterate_late_body<THREADS>(g, in, st, ap, lead, (int)blockIdx.x < lead_blocks, (int)blockIdx.x, lead_blocks);
}

template <int MINW, int THREADS>
__global__ __launch_bounds__(THREADS, MINW) void k_iterate_late_batch(const IterateDesc* __restrict__ table, int nseq, int per_seq) {
    const int p = (int)blockIdx.x;
    int seq, pb;
    bool is_lead = false;
    if (p < BATCH_LEAD_SLOTS) {  // block-uniform
        if (p >= nseq) return;
        seq = p;
        pb = 0;
        is_lead = true;
    } else {
        seq = (p - BATCH_LEAD_SLOTS) / per_seq;
        pb = (p - BATCH_LEAD_SLOTS) - seq * per_seq;
    }
    const IterateDesc d = load_descriptor(table, seq);
    if (is_lead ? !(d.lead.box && d.lead.solve) : pb >= d.blocks) return;
    iterate_late_body<THREADS>(d.g, d.in, d.st, d.ap, d.lead, is_lead, pb, 0);
}

// nn_cache positions of the finished registration -> original map indices, shifted by the `evicted` oldest points the
// coming map update drops (runs right before the grid is rebuilt, while the positions still mean something)
__global__ void k_cache_to_seed(const int4* __restrict__ nn_cache, const float4* __restrict__ pts, int n, int m,
                                int evicted, int* __restrict__ seed) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int x = nn_cache[i].x, pos = x & CACHE_POS_MASK;
    int o = -1;
    if (x >= 0 && pos < m) o = __float_as_int(pts[pos].w) - evicted;
    seed[i] = o;
}

// ---------------------------------------------------------------------------------------------------------------------
// K2: kNN normals for the queued map points
// ---------------------------------------------------------------------------------------------------------------------
// k smallest (distance, original index) pairs as 64-bit keys (d2 bits << 32 | index): d2 >= 0, so the float bits order
// like the values and one unsigned 64-bit compare gives the lexicographic (distance, index) order the search breaks
// ties with.  Sorted ascending; a compare-swap step is 1 compare + 4 selects.
static constexpr unsigned long long KEY_EMPTY = (0x7f800000ull << 32) | 0x7fffffffull;  // (+inf, max index)

__device__ inline unsigned long long make_key(float d2, int idx) {
    return ((unsigned long long)__float_as_uint(d2) << 32) | (unsigned long long)(unsigned)idx;
}
__device__ inline float key_d2(unsigned long long k) { return __uint_as_float((unsigned)(k >> 32)); }
__device__ inline int key_idx(unsigned long long k) { return (int)(unsigned)(k & 0xffffffffull); }

__device__ inline unsigned long long point_key(const float4 q, float px, float py, float pz) {
    const float dx = q.x - px, dy = q.y - py, dz = q.z - pz;
    return make_key(fmaf(dz, dz, fmaf(dy, dy, dx * dx)), __float_as_int(q.w));
}

template <int KN>
struct TopK {
    unsigned long long key[KN];
    __device__ inline void init() {
#pragma unroll
        for (int k = 0; k < KN; ++k) key[k] = KEY_EMPTY;
    }
    __device__ inline float kth() const { return key_d2(key[KN - 1]); }
    __device__ inline void insert(unsigned long long c) {
        if (!(c < key[KN - 1])) return;
        key[KN - 1] = c;
#pragma unroll
        for (int k = KN - 1; k > 0; --k) {
            const unsigned long long lo = key[k] < key[k - 1] ? key[k] : key[k - 1];
            const unsigned long long hi = key[k] < key[k - 1] ? key[k - 1] : key[k];
            key[k - 1] = lo;
            key[k] = hi;
        }
    }
};

template <int KN>
__device__ inline void scan_cell_knn(const GridView& g, int start, int count, float px, float py, float pz,
                                     TopK<KN>& t) {
    const int last = start + count - 1;
    for (int k = start; k <= last; k += 4) {
        const int k1 = min(k + 1, last), k2 = min(k + 2, last), k3 = min(k + 3, last);
        const float4 q0 = g.pts[k], q1 = g.pts[k1], q2 = g.pts[k2], q3 = g.pts[k3];
        t.insert(point_key(q0, px, py, pz));
        if (k1 != k) t.insert(point_key(q1, px, py, pz));
        if (k2 != k1) t.insert(point_key(q2, px, py, pz));
        if (k3 != k2) t.insert(point_key(q3, px, py, pz));
    }
}

// smallest-eigenvalue eigenvector of a symmetric 3x3 (cyclic Jacobi in f64).  The reference takes vh[2] of an f32
// LAPACK SVD of the same matrix (local_map.py:414-416); for a symmetric PSD matrix that is this eigenvector up to sign,
// and the sign cancels in J^T J and J^T r.
__device__ inline void smallest_eigenvector(float a00, float a01, float a02, float a11, float a12, float a22,
                                            float& nx, float& ny, float& nz) {
    // cyclic Jacobi in f32 (the reference's own decomposition is an f32 LAPACK SVD); the matrix is first scaled to
    // unit trace so that tiny covariances do not underflow the rotation tests
    const float tr = a00 + a11 + a22;
    const float sc = tr > 0.f ? 1.0f / tr : 1.0f;
    float A[3][3] = {{a00 * sc, a01 * sc, a02 * sc}, {a01 * sc, a11 * sc, a12 * sc}, {a02 * sc, a12 * sc, a22 * sc}};
    float V[3][3] = {{1, 0, 0}, {0, 1, 0}, {0, 0, 1}};
    for (int sweep = 0; sweep < 8; ++sweep) {
        const float off = fabsf(A[0][1]) + fabsf(A[0][2]) + fabsf(A[1][2]);
        if (off <= 1e-9f) break;
#pragma unroll
        for (int pq = 0; pq < 3; ++pq) {
            const int p = pq == 2 ? 1 : 0;
            const int q = pq == 0 ? 1 : 2;
            const float apq = A[p][q];
            if (apq == 0.0f) continue;
            const float theta = (A[q][q] - A[p][p]) / (2.0f * apq);
            const float t = (theta >= 0 ? 1.0f : -1.0f) / (fabsf(theta) + sqrtf(fmaf(theta, theta, 1.0f)));
            const float c = 1.0f / sqrtf(fmaf(t, t, 1.0f)), s = t * c;
#pragma unroll
            for (int k = 0; k < 3; ++k) {  // A <- A J
                const float akp = A[k][p], akq = A[k][q];
                A[k][p] = c * akp - s * akq;
                A[k][q] = s * akp + c * akq;
            }
#pragma unroll
            for (int k = 0; k < 3; ++k) {  // A <- J^T A
                const float apk = A[p][k], aqk = A[q][k];
                A[p][k] = c * apk - s * aqk;
                A[q][k] = s * apk + c * aqk;
            }
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                const float vkp = V[k][p], vkq = V[k][q];
                V[k][p] = c * vkp - s * vkq;
                V[k][q] = s * vkp + c * vkq;
            }
        }
    }
    // ties -> the last axis, like vh[2] of an already diagonal input (selects, not indexing: V stays in registers)
    // (two rounds of conditional moves on values; written as `if (..) x = V[0][1]` the optimiser turned the choice of the
    // COLUMN into an index and parked V in 48 bytes of scratch memory — in every normal kernel)
    const bool c1 = A[1][1] < A[2][2];
    float lam = c1 ? A[1][1] : A[2][2];
    float x = c1 ? V[0][1] : V[0][2], y = c1 ? V[1][1] : V[1][2], z = c1 ? V[2][1] : V[2][2];
    asm volatile("" : "+v"(x), "+v"(y), "+v"(z), "+v"(lam));  // (keeps the two rounds apart: values, not a column number)
    const bool c0 = A[0][0] < lam;
    x = c0 ? V[0][0] : x;
    y = c0 ? V[1][0] : y;
    z = c0 ? V[2][0] : z;
    const float inv = 1.0f / sqrtf(x * x + y * y + z * z);
    nx = x * inv;
    ny = y * inv;
    nz = z * inv;
}

// covariance of the k neighbours (first of the k+1 dropped, :407) centred on the query point (:411-413), f32:
// cov = {c00, c01, c02, c11, c12, c22}
// The six sums of a covariance, ORDER-INDEPENDENT: every float32 product is first rounded to a multiple of 2^-40
// ((v + C) - C in float64 with ulp(C) = 2^-40: exact for |v| < 2^11, i.e. neighbours up to 45 m away), and sums of such
// multiples below 2^13 are exact in float64 whatever their order.  The kernels that estimate a normal reach its
// neighbours in different orders (sorted by distance here, in list order in estimate_cov_hood, lane by lane and wave by
// wave elsewhere) and must agree bit for bit.
struct CovSums {
    double c[6];
    __device__ inline void zero() {
#pragma unroll
        for (int k = 0; k < 6; ++k) c[k] = 0.0;
    }
    static __device__ inline double grid(float v) {
        const double C = 6144.0;  // 1.5 * 2^12: ulp = 2^-40
        return ((double)v + C) - C;
    }
    __device__ inline void add(float dx, float dy, float dz) {
        c[0] += grid(dx * dx);
        c[1] += grid(dx * dy);
        c[2] += grid(dx * dz);
        c[3] += grid(dy * dy);
        c[4] += grid(dy * dz);
        c[5] += grid(dz * dz);
    }
    __device__ inline void store(int used, float* __restrict__ cov) const {
        const float invk = used > 0 ? 1.0f / (float)used : 0.f;
#pragma unroll
        for (int k = 0; k < 6; ++k) cov[k] = (float)c[k] * invk;
    }
};

template <int KN>
__device__ inline void neighbourhood_cov(const GridView& g, float px, float py, float pz, const TopK<KN>& t,
                                         float* __restrict__ cov) {
    CovSums cs;
    cs.zero();
    int used = 0;
#pragma unroll
    for (int k = 1; k < KN; ++k) {
        if (t.key[k] == KEY_EMPTY) continue;  // map smaller than k + 1 points
        const float4 q = g.pts[g.pos_of_orig[key_idx(t.key[k])]];
        cs.add(q.x - px, q.y - py, q.z - pz);
        ++used;
    }
    cs.store(used, cov);
}

__device__ inline void normal_from_cov(const float* __restrict__ cov, int s, float4* __restrict__ normals,
                                       int* __restrict__ nflag) {
    float nx, ny, nz;
    smallest_eigenvector(cov[0], cov[1], cov[2], cov[3], cov[4], cov[5], nx, ny, nz);
    normals[s] = make_float4(nx, ny, nz, 1.f);
    nflag[s] = 1;
}

// hashed ring search on one level from ring `first_ring` on (rings below it were already visited by the caller);
// true when the k-th neighbour is provably exact
template <int KN>
__device__ inline bool knn_level(const GridView& g, float px, float py, float pz, int first_ring, int max_rings,
                                 bool translate, TopK<KN>& t) {
    const int cx = cell_coord(px, g.inv_h), cy = cell_coord(py, g.inv_h), cz = cell_coord(pz, g.inv_h);
    const float h = g.h;
    const float fx = fminf(fmaxf(px - (float)cx * h, 0.f), h);
    const float fy = fminf(fmaxf(py - (float)cy * h, 0.f), h);
    const float fz = fminf(fmaxf(pz - (float)cz * h, 0.f), h);
    const float edge = fminf(fminf(fminf(fx, h - fx), fminf(fy, h - fy)), fminf(fz, h - fz));
    int start, count;
    if (first_ring == 0 && grid_lookup(g, cx, cy, cz, start, count)) scan_cell_knn<KN>(g, start, count, px, py, pz, t);
    bool exact = false;
    for (int r = first_ring < 1 ? 1 : first_ring; r <= max_rings && !exact; ++r) {
        for (int oz = -r; oz <= r; ++oz) {
            const float gz = axis_gap(oz, fz, h);
            const float gz2 = gz * gz;
            if (gz2 > t.kth()) continue;
            const int az = oz < 0 ? -oz : oz;
            for (int oy = -r; oy <= r; ++oy) {
                const float gy = axis_gap(oy, fy, h);
                const float gyz2 = fmaf(gy, gy, gz2);
                if (gyz2 > t.kth()) continue;
                const int ay = oy < 0 ? -oy : oy;
                const int step = ((az == r) || (ay == r)) ? 1 : 2 * r;
                for (int ox = -r; ox <= r; ox += step) {
                    const float gx = axis_gap(ox, fx, h);
                    if (fmaf(gx, gx, gyz2) > t.kth()) continue;
                    if (grid_lookup(g, cx + ox, cy + oy, cz + oz, start, count))
                        scan_cell_knn<KN>(g, start, count, px, py, pz, t);
                }
            }
        }
        const float bound = (float)r * h + edge;
        exact = t.kth() <= bound * bound * 0.999999f;
    }
    (void)translate;  // keys carry original indices: nothing to translate between levels
    return exact;
}

// continuation after the fine ring 1 (rows): fine ring 2.., then the coarse level, then the exhaustive scan
template <int KN>
__device__ inline void knn_rings(const GridView& g, float px, float py, float pz, int first_ring, int max_rings,
                                 TopK<KN>& t) {
    if (knn_level<KN>(g, px, py, pz, first_ring, max_rings, false, t)) return;
    if (g.ctable) {
        t.init();
        if (knn_level<KN>(coarse_view(g), px, py, pz, 0, COARSE_RINGS, true, t)) return;
    }
    t.init();
    scan_cell_knn<KN>(g, 0, g.m, px, py, pz, t);
}

template <int KN, int NL>
__device__ inline void merge_group(TopK<KN>& t, TopK<KN>& m);

static constexpr int NRM_THREADS = 256;

// kNN counterpart of `coop_rings` for a WHOLE WAVE: rings r_begin..r_end of one level around a map point.  `m` is the
// merged list so far (identical in all lanes); per ring lane 0 continues from it, the others from empty lists.  The cells
// of a ring are taken 64 at a time, one hashed probe per lane; the candidates of the cells that pass the box test are
// then laid end to end (prefix sum over the lanes, cell of candidate j found by a 6-step search in LDS) and dealt out
// evenly — a coarse cell of a thousand points costs every lane sixteen candidates instead of one lane a thousand.
// `limit` = a squared distance beyond which nothing can be among the k nearest (k points are already known inside it).
// Keys carry original indices, so fine and coarse levels mix freely.  `wl` = 128 ints of LDS private to the wave.
// Returns true when the k-th neighbour is provably exact on this level.
template <int KN>
__device__ inline bool wave_knn_rings(const GridView& lv, float px, float py, float pz, int lane, int r_begin, int r_end,
                                      float limit, TopK<KN>& m, int* __restrict__ wl) {
    const int cx = cell_coord(px, lv.inv_h), cy = cell_coord(py, lv.inv_h), cz = cell_coord(pz, lv.inv_h);
    const float h = lv.h;
    const float fx = fminf(fmaxf(px - (float)cx * h, 0.f), h);
    const float fy = fminf(fmaxf(py - (float)cy * h, 0.f), h);
    const float fz = fminf(fmaxf(pz - (float)cz * h, 0.f), h);
    const float edge = fminf(fminf(fminf(fx, h - fx), fminf(fy, h - fy)), fminf(fz, h - fz));
    for (int r = r_begin; r <= r_end; ++r) {
        TopK<KN> t;
        if (lane == 0) {
            t = m;
        } else {
            t.init();
        }
        const float kth = fminf(m.kth(), limit);
        const int side = 2 * r + 1, total = side * side * side;
        for (int c0 = 0; c0 < total; c0 += 64) {  // wave-uniform trip count
            const int c = c0 + lane;
            int start = 0, count = 0;
            if (c < total) {
                const int ox = c % side - r, oy = (c / side) % side - r, oz = c / (side * side) - r;
                const int mx = max(max(ox < 0 ? -ox : ox, oy < 0 ? -oy : oy), oz < 0 ? -oz : oz);
                if (mx == r) {  // the shell only: the interior belongs to the previous rings
                    const float gx = axis_gap(ox, fx, h), gy = axis_gap(oy, fy, h), gz = axis_gap(oz, fz, h);
                    if (!(fmaf(gx, gx, fmaf(gy, gy, gz * gz)) > kth) &&
                        !grid_lookup(lv, cx + ox, cy + oy, cz + oz, start, count))
                        count = 0;
                }
            }
            int incl = count;
#pragma unroll
            for (int o = 1; o <= 32; o <<= 1) {
                const int up = __shfl_up(incl, o, 64);
                if (lane >= o) incl += up;
            }
            const int cand = __shfl(incl, 63, 64);
            const int excl = incl - count;
            wl[lane] = excl;
            wl[64 + lane] = start - excl;  // candidate j of the cell of lane l sits at position j + wl[64 + l]
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            for (int j0 = 0; j0 < cand; j0 += 128) {
                const int ja = j0 + lane, jb = j0 + 64 + lane;
                int ca = 0, cb = 0;
#pragma unroll
                for (int s2 = 32; s2 > 0; s2 >>= 1) {
                    if (wl[ca + s2] <= ja) ca += s2;
                    if (wl[cb + s2] <= jb) cb += s2;
                }
                const int pa = ja + wl[64 + ca], pb = jb + wl[64 + cb];
                float4 qa, qb;
                if (ja < cand) qa = lv.pts[pa];
                if (jb < cand) qb = lv.pts[pb];
                if (ja < cand) {
                    const unsigned long long key = point_key(qa, px, py, pz);
                    if (!(key_d2(key) > limit)) t.insert(key);
                }
                if (jb < cand) {
                    const unsigned long long key = point_key(qb, px, py, pz);
                    if (!(key_d2(key) > limit)) t.insert(key);
                }
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();  // the next round overwrites wl
        }
        merge_group<KN, 64>(t, m);
        const float bound = (float)r * h + edge;
        if (m.kth() <= bound * bound * 0.999999f) return true;
    }
    return false;
}

// Neighbourhood covariance of one map point by NL (4 or 2) lanes (lane 0 of the group writes cov[6]); the eigen-solve
// that turns it into a normal runs afterwards on dense waves, one lane per point (see k_normals_all).  A map point
// always lies in an occupied cell, so its 27-neighbourhood comes
// from the cell's row (no hashing).  Each lane keeps the top-k of its share of the candidates (own cell strided,
// neighbour cells split 7/6/7/6), then the four sorted lists are merged by k rounds of "group-min of the heads, winner
// pops".  Only if the k-th neighbour is not provably inside ring 1 does lane 0 continue with the hashed rings / coarse
// level.  4x the waves and ~1/4 of the serial insert chain of a one-lane-per-point search; same result.
template <int KN, int NL>
__device__ inline bool estimate_cov(const GridView& g, int s, int sub, float* __restrict__ cov, int2* __restrict__ stack,
                                    int stride, TopK<KN>& m) {
    const float4 P = g.pts[s];
    const float px = P.x, py = P.y, pz = P.z;
    TopK<KN> t;
    t.init();
    const float h = g.h;
    const int cx = cell_coord(px, g.inv_h), cy = cell_coord(py, g.inv_h), cz = cell_coord(pz, g.inv_h);
    const float fx = fminf(fmaxf(px - (float)cx * h, 0.f), h);
    const float fy = fminf(fmaxf(py - (float)cy * h, 0.f), h);
    const float fz = fminf(fmaxf(pz - (float)cz * h, 0.f), h);
    const float edge = fminf(fminf(fminf(fx, h - fx), fminf(fy, h - fy)), fminf(fz, h - fz));
    const int2* __restrict__ r = g.rows + (size_t)g.row_of_pos[s] * ROW_STRIDE;
    const int2 own = r[13];
    constexpr int CPL = ROW_STRIDE / NL;  // row entries per lane (entry 27 is padding)
    int2 cell[CPL];
#pragma unroll
    for (int k = 0; k < CPL; ++k) cell[k] = r[sub * CPL + k];
    // candidates are fetched four at a time (independent 16-byte loads in flight together): one load per insert made
    // the whole kernel wait a full memory latency per candidate
    {
        const int last = own.x + own.y - 1;
        for (int k = own.x + sub; k <= last; k += 4 * NL) {
            const bool v1 = k + NL <= last, v2 = k + 2 * NL <= last, v3 = k + 3 * NL <= last;
            const float4 q0 = g.pts[k], q1 = g.pts[v1 ? k + NL : k], q2 = g.pts[v2 ? k + 2 * NL : k],
                         q3 = g.pts[v3 ? k + 3 * NL : k];
            t.insert(point_key(q0, px, py, pz));
            if (v1) t.insert(point_key(q1, px, py, pz));
            if (v2) t.insert(point_key(q2, px, py, pz));
            if (v3) t.insert(point_key(q3, px, py, pz));
        }
    }
    // the lane's neighbour cells go to its LDS list, then ONE loop streams their candidates (an insert is ~100 VALU:
    // walking the 7 row entries in lockstep would make the wave pay every lane's longest cell 7 times over)
    int nl = 0;
#pragma unroll
    for (int k = 0; k < CPL; ++k) {
        const int c = sub * CPL + k;
        if (c == 13 || c >= 27 || cell[k].y <= 0) continue;
        stack[nl * stride] = make_int2(cell[k].x, cell[k].y | (c << 24));
        ++nl;
    }
    {
        int ci = 0, st = 0, cnt = 0, k = 0;
        for (;;) {
            if (k >= cnt) {
                if (ci >= nl) break;
                const int2 nx = stack[ci * stride];
                ++ci;
                const int c = (int)((unsigned)nx.y >> 24);
                const float gx = axis_gap(c % 3 - 1, fx, h), gy = axis_gap((c / 3) % 3 - 1, fy, h),
                            gz = axis_gap(c / 9 - 1, fz, h);
                if (fmaf(gx, gx, fmaf(gy, gy, gz * gz)) > t.kth()) continue;
                st = nx.x;
                cnt = nx.y & 0xffffff;
                k = 0;
            }
            const int last = st + cnt - 1, k0 = st + k;
            const int k1 = min(k0 + 1, last), k2 = min(k0 + 2, last), k3 = min(k0 + 3, last);
            const float4 q0 = g.pts[k0], q1 = g.pts[k1], q2 = g.pts[k2], q3 = g.pts[k3];
            t.insert(point_key(q0, px, py, pz));
            if (k1 != k0) t.insert(point_key(q1, px, py, pz));
            if (k2 != k1) t.insert(point_key(q2, px, py, pz));
            if (k3 != k2) t.insert(point_key(q3, px, py, pz));
            k += 4;
        }
    }
    merge_group<KN, NL>(t, m);
    const float bound1 = h + edge;
    const bool exact = m.kth() <= bound1 * bound1 * 0.999999f;  // group-uniform: m is shared
    if (exact && sub == 0) neighbourhood_cov<KN>(g, px, py, pz, m, cov);
    return exact;
}

// ---------------------------------------------------------------------------------------------------------------------
// Ring 1 of the kNN from the cell's NEIGHBOURHOOD LIST (hash_grid.hip::k_hood_build), 4 lanes per map point:
//   pass 1  every lane streams its quarter of the list (4 loads in flight) into a sorted list of its KN smallest squared
//           DISTANCES — plain floats, updated branch-free by one v_med3 per slot; two butterfly rounds (element-wise min
//           of one sorted list with the other reversed = the KN smallest of both) give T, the KN-th smallest distance of
//           the whole list;
//   pass 2  the list again (L1-hot): candidates with d < T, and those with d == T, are noted in LDS and counted;
//   pass 3  the ~KN noted ones feed the order-independent covariance sums.
// The k + 1 nearest neighbours are exactly {d < T} plus the ties at T when together they are KN; more ties than that
// (duplicates, lattices) need the (distance, index) order — and a T beyond what ring 1 certifies needs more rings: both
// return false and the point goes to the whole-wave continuation with 64-bit keys (finish_cov_wave), like before.
// The first neighbour the reference drops (:407) is the point itself or a twin at distance 0: it adds nothing to the sums.
// Returns true when the covariance has been written (by lane 0 of the group).  `sel` = this lane's column of an LDS array
// [KN][blockDim] of 16-bit list positions.
// ---------------------------------------------------------------------------------------------------------------------
template <int KN>
__device__ inline void sorted_insert_med3(float (&D)[KN], float c) {
#pragma unroll
    for (int i = KN - 1; i > 0; --i) D[i] = __builtin_amdgcn_fmed3f(c, D[i - 1], D[i]);  // (old D[i - 1]: going down)
    D[0] = fminf(c, D[0]);
}

template <int KN>
__device__ inline bool estimate_cov_hood(const GridView& g, int s, int sub, float* __restrict__ cov,
                                         unsigned short* __restrict__ sel, int stride) {
    const float4 P = g.pts[s];
    const float px = P.x, py = P.y, pz = P.z;
    const float h = g.h;
    const int cx = cell_coord(px, g.inv_h), cy = cell_coord(py, g.inv_h), cz = cell_coord(pz, g.inv_h);
    const float fx = fminf(fmaxf(px - (float)cx * h, 0.f), h);
    const float fy = fminf(fmaxf(py - (float)cy * h, 0.f), h);
    const float fz = fminf(fmaxf(pz - (float)cz * h, 0.f), h);
    const float edge = fminf(fminf(fminf(fx, h - fx), fminf(fy, h - fy)), fminf(fz, h - fz));
    const int2 hh = g.rows[(size_t)g.row_of_pos[s] * ROW_STRIDE + 27];
    const float4* __restrict__ H = g.hood + hh.x;
    const int n = hh.y;
    // ---- pass 1
    float D[KN];
#pragma unroll
    for (int i = 0; i < KN; ++i) D[i] = INFINITY;
    for (int j = sub; j < n; j += 16) {
        const bool v1 = j + 4 < n, v2 = j + 8 < n, v3 = j + 12 < n;
        const float4 q0 = H[j], q1 = H[v1 ? j + 4 : j], q2 = H[v2 ? j + 8 : j], q3 = H[v3 ? j + 12 : j];
        float dx = q0.x - px, dy = q0.y - py, dz = q0.z - pz;
        sorted_insert_med3<KN>(D, fmaf(dz, dz, fmaf(dy, dy, dx * dx)));
        dx = q1.x - px, dy = q1.y - py, dz = q1.z - pz;
        sorted_insert_med3<KN>(D, v1 ? fmaf(dz, dz, fmaf(dy, dy, dx * dx)) : INFINITY);  // (+inf changes nothing)
        dx = q2.x - px, dy = q2.y - py, dz = q2.z - pz;
        sorted_insert_med3<KN>(D, v2 ? fmaf(dz, dz, fmaf(dy, dy, dx * dx)) : INFINITY);
        dx = q3.x - px, dy = q3.y - py, dz = q3.z - pz;
        sorted_insert_med3<KN>(D, v3 ? fmaf(dz, dz, fmaf(dy, dy, dx * dx)) : INFINITY);
    }
    // ---- the KN-th smallest of the four lists
    float T;
    {
        float E[KN];
#pragma unroll
        for (int i = 0; i < KN; ++i) E[i] = INFINITY;
#pragma unroll
        for (int i = 0; i < KN; ++i)  // the KN smallest of this lane's and its neighbour's lists, sorted again
            sorted_insert_med3<KN>(E, fminf(D[i], quad_xor<1>(D[KN - 1 - i])));
        T = -INFINITY;
#pragma unroll
        for (int i = 0; i < KN; ++i) T = fmaxf(T, fminf(E[i], quad_xor<2>(E[KN - 1 - i])));
    }
    const float bound = h + edge;
    if (!(T <= bound * bound * 0.999999f)) return false;  // group-uniform (T is): ring 1 does not certify the KN-th neighbour
    if (n > 65535) return false;                          // (the notes below are 16-bit list positions)
    // ---- pass 2: who is in
    int nsel = 0, over = 0;
    for (int j = sub; j < n; j += 16) {
        const bool v1 = j + 4 < n, v2 = j + 8 < n, v3 = j + 12 < n;
        const float4 q0 = H[j], q1 = H[v1 ? j + 4 : j], q2 = H[v2 ? j + 8 : j], q3 = H[v3 ? j + 12 : j];
        float dx = q0.x - px, dy = q0.y - py, dz = q0.z - pz;
        const float d0 = fmaf(dz, dz, fmaf(dy, dy, dx * dx));
        dx = q1.x - px, dy = q1.y - py, dz = q1.z - pz;
        const float d1 = v1 ? fmaf(dz, dz, fmaf(dy, dy, dx * dx)) : INFINITY;
        dx = q2.x - px, dy = q2.y - py, dz = q2.z - pz;
        const float d2 = v2 ? fmaf(dz, dz, fmaf(dy, dy, dx * dx)) : INFINITY;
        dx = q3.x - px, dy = q3.y - py, dz = q3.z - pz;
        const float d3 = v3 ? fmaf(dz, dz, fmaf(dy, dy, dx * dx)) : INFINITY;
        // (a lane that meets more than KN candidates within T has met ties beyond the KN-th place)
        if (d0 <= T) { if (nsel < KN) { sel[nsel * stride] = (unsigned short)j;        ++nsel; } else over = 1; }
        if (d1 <= T) { if (nsel < KN) { sel[nsel * stride] = (unsigned short)(j + 4);  ++nsel; } else over = 1; }
        if (d2 <= T) { if (nsel < KN) { sel[nsel * stride] = (unsigned short)(j + 8);  ++nsel; } else over = 1; }
        if (d3 <= T) { if (nsel < KN) { sel[nsel * stride] = (unsigned short)(j + 12); ++nsel; } else over = 1; }
    }
    int tot = nsel;
    tot += quad_xor<1>(tot);
    tot += quad_xor<2>(tot);
    over |= quad_xor<1>(over);
    over |= quad_xor<2>(over);
    if (tot != KN || over) return false;  // ties at T beyond the KN-th place: the keyed (distance, index) search decides
    // ---- pass 3: the covariance sums
    CovSums cs;
    cs.zero();
    for (int i = 0; i < nsel; ++i) {
        const float4 q = H[sel[i * stride]];
        cs.add(q.x - px, q.y - py, q.z - pz);
    }
#pragma unroll
    for (int k = 0; k < 6; ++k) {
        cs.c[k] += quad_xor<1>(cs.c[k]);
        cs.c[k] += quad_xor<2>(cs.c[k]);
    }
    if (sub == 0) cs.store(KN - 1, cov);
    return true;
}

// The points ring 1 does not settle (isolated points: 1-2 % of a LiDAR map) are finished by a WHOLE WAVE each: a few of
// them per launch walked the hashed rings and the coarse level with 4 lanes — 25 dependent probes per lane and ring,
// coarse cells of hundreds of points — and set the duration of the kernel (the same tail as in the iteration kernel).
// `m` = the merged list after ring 1 (identical in every lane).  Fine rings 2..max_rings, the coarse level, then the
// exhaustive scan, every one split over the 64 lanes; lane 0 writes the covariance.
template <int KN>
__device__ inline void finish_cov_wave(const GridView& g, int s, int lane, int max_rings, TopK<KN>& m,
                                       float* __restrict__ cov, int* __restrict__ wl) {
    const float4 P = g.pts[s];
    const float px = P.x, py = P.y, pz = P.z;
    if (g.dbg && lane == 0) atomicAdd(&g.dbg[7], 1);
    bool exact = max_rings >= 2 && wave_knn_rings<KN>(g, px, py, pz, lane, 2, max_rings, INFINITY, m, wl);
    if (g.dbg && lane == 0 && !exact) atomicAdd(&g.dbg[15], 1);
    if (!exact && g.ctable) {
        // the coarse rings start at ring 0 and re-find the fine results: the list starts empty (no duplicates), but what
        // the fine rings found still bounds the search — k points are known within its k-th distance
        const float limit = m.kth();
        m.init();
        exact = wave_knn_rings<KN>(coarse_view(g), px, py, pz, lane, 0, COARSE_RINGS, limit, m, wl);
    }
    if (!exact) {  // farther than COARSE_RINGS coarse cells from k map points: exhaustive
        TopK<KN> t;
        t.init();
        m.init();
        for (int k = lane; k < g.m; k += 64) t.insert(point_key(g.pts[k], px, py, pz));
        merge_group<KN, 64>(t, m);
    }
    if (lane == 0) neighbourhood_cov<KN>(g, px, py, pz, m, cov);
}

// ring 1 by a 4-lane group (estimate_cov without a caller for its merged list)
template <int KN>
__device__ inline bool lazy_cov_group(const GridView& g, int s, int sub, float* __restrict__ cov, int2* __restrict__ stack,
                                      int stride) {
    TopK<KN> m;
    return estimate_cov<KN, 4>(g, s, sub, cov, stack, stride, m);
}

// The covariance of map point `s` from scratch by a whole wave (the fused iteration kernel's LAZY instantiation: normals on
// demand): rings 0 .. max_rings of the fine level, then the continuation of finish_cov_wave.  lane 0 writes cov[0..6].
template <int KN>
__device__ inline void lazy_cov_wave(const GridView& g, int s, int lane, int max_rings, int* __restrict__ wl,
                                     float* __restrict__ cov) {
    const float4 P = g.pts[s];
    const float px = P.x, py = P.y, pz = P.z;
    TopK<KN> m;
    m.init();
    bool exact = wave_knn_rings<KN>(g, px, py, pz, lane, 0, max_rings, INFINITY, m, wl);
    if (!exact && g.ctable) {
        const float limit = m.kth();
        m.init();
        exact = wave_knn_rings<KN>(coarse_view(g), px, py, pz, lane, 0, COARSE_RINGS, limit, m, wl);
    }
    if (!exact) {
        TopK<KN> t;
        t.init();
        m.init();
        for (int k = lane; k < g.m; k += 64) t.insert(point_key(g.pts[k], px, py, pz));
        merge_group<KN, 64>(t, m);
    }
    if (lane == 0) neighbourhood_cov<KN>(g, px, py, pz, m, cov);
}

// the unsettled points of a block wait in LDS: point, slot of its covariance, merged list after ring 1
template <int KN, int PTS>
struct PendingKnn {
    unsigned long long key[PTS][KN];
    int s[PTS];
    int lq[PTS];

    int wl[NRM_THREADS / 64][128];  // per-wave scratch of wave_knn_rings
    int n;
};

template <int KN, int PTS>
__device__ inline void pend_push(PendingKnn<KN, PTS>& p, int s, int lq, const TopK<KN>& m) {
    const int k = atomicAdd(&p.n, 1);
    p.s[k] = s;
    p.lq[k] = lq;
#pragma unroll
    for (int j = 0; j < KN; ++j) p.key[k][j] = m.key[j];
}

// every thread of the block calls (between two barriers of the caller)
template <int KN, int PTS>
__device__ inline void pend_finish(const GridView& g, PendingKnn<KN, PTS>& p, int max_rings, float (*covs)[7]) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    for (int k = wave; k < p.n; k += NRM_THREADS / 64) {
        TopK<KN> m;
#pragma unroll
        for (int j = 0; j < KN; ++j) m.key[j] = p.key[k][j];
        finish_cov_wave<KN>(g, p.s[k], lane, max_rings, m, covs[p.lq[k]], p.wl[wave]);
    }
}

// merge of the NL lanes' sorted lists: k rounds of "group-min of the heads, the winner pops" (t is consumed)
template <int KN, int NL>
__device__ inline void merge_group(TopK<KN>& t, TopK<KN>& m) {
#pragma unroll
    for (int round = 0; round < KN; ++round) {
        unsigned long long best = t.key[0];
        if (NL == 16) {  // a row of 16 lanes: DPP (search_device.h::row16_step)
            unsigned long long other = row16_step<0>(best);
            best = other < best ? other : best;
            other = row16_step<1>(best);
            best = other < best ? other : best;
            other = row16_step<2>(best);
            best = other < best ? other : best;
            other = row16_step<3>(best);
            best = other < best ? other : best;
        } else if (NL == 64) {
            // a whole wave (round 6): the minimum of every row of 16 by DPP, then the four rows' minima read into scalar
            // registers (v_readlane) — 4 x 5 + 8 + 9 VALU instructions where six __shfl_xor steps of a 64-bit key were twelve
            // ds_bpermute round trips through the LDS crossbar (~800 cycles per round, eleven rounds per merge, two or three
            // merges per straggler of the kNN normals).  Every lane of the wave must be active.
            unsigned long long other = row16_step<0>(best);
            best = other < best ? other : best;
            other = row16_step<1>(best);
            best = other < best ? other : best;
            other = row16_step<2>(best);
            best = other < best ? other : best;
            other = row16_step<3>(best);
            best = other < best ? other : best;
            const unsigned blo = (unsigned)(best & 0xffffffffull), bhi = (unsigned)(best >> 32);
            unsigned long long r0 = ((unsigned long long)(unsigned)__builtin_amdgcn_readlane((int)bhi, 0) << 32) |
                                    (unsigned)__builtin_amdgcn_readlane((int)blo, 0);
            const unsigned long long r1 = ((unsigned long long)(unsigned)__builtin_amdgcn_readlane((int)bhi, 16) << 32) |
                                          (unsigned)__builtin_amdgcn_readlane((int)blo, 16);
            const unsigned long long r2 = ((unsigned long long)(unsigned)__builtin_amdgcn_readlane((int)bhi, 32) << 32) |
                                          (unsigned)__builtin_amdgcn_readlane((int)blo, 32);
            const unsigned long long r3 = ((unsigned long long)(unsigned)__builtin_amdgcn_readlane((int)bhi, 48) << 32) |
                                          (unsigned)__builtin_amdgcn_readlane((int)blo, 48);
            r0 = r1 < r0 ? r1 : r0;
            const unsigned long long r23 = r3 < r2 ? r3 : r2;
            best = r23 < r0 ? r23 : r0;
        } else if (NL > 4) {  // (other group sizes: the LDS crossbar)
#pragma unroll
            for (int o = 1; o < NL; o <<= 1) {
                const unsigned lo = __shfl_xor((unsigned)(best & 0xffffffffull), o, 64);
                const unsigned hi = __shfl_xor((unsigned)(best >> 32), o, 64);
                const unsigned long long other = ((unsigned long long)hi << 32) | lo;
                best = other < best ? other : best;
            }
        } else {
            const unsigned lo = (unsigned)quad_xor<1>((int)(best & 0xffffffffull));
            const unsigned hi = (unsigned)quad_xor<1>((int)(best >> 32));
            const unsigned long long other = ((unsigned long long)hi << 32) | lo;
            best = other < best ? other : best;
        }
        if (NL == 4) {
            const unsigned lo = (unsigned)quad_xor<2>((int)(best & 0xffffffffull));
            const unsigned hi = (unsigned)quad_xor<2>((int)(best >> 32));
            const unsigned long long other = ((unsigned long long)hi << 32) | lo;
            best = other < best ? other : best;
        }
        m.key[round] = best;
        const bool won = (t.key[0] == best) && (best != KEY_EMPTY);  // original indices are unique: one winner
#pragma unroll
        for (int k = 0; k < KN - 1; ++k) t.key[k] = won ? t.key[k + 1] : t.key[k];
        if (won) t.key[KN - 1] = KEY_EMPTY;
    }
}

// Block = NRM_THREADS / NL map points x NL lanes: the groups leave their covariances in LDS, then the first threads
// (whole waves, every lane busy) run the Jacobi eigen-solves — a group would otherwise spend the ~1.5k-instruction solve
// with one lane in NL active.
// lazy schedule: the map points queued by the search of this iteration (a no-op once the registration is done; the
// count feeds `normals_computed`)
template <int KN, int NL>
__global__ __launch_bounds__(NRM_THREADS) void k_normals(GridView g, RegState* __restrict__ st,
                                                         const int* __restrict__ worklist, int max_rings,
                                                         float4* __restrict__ normals, int* __restrict__ nflag) {
    constexpr int PTS = NRM_THREADS / NL;
    if (st->done) return;
    __shared__ int2 cellstack[ROW_STRIDE / NL][NRM_THREADS];
    __shared__ float covs[PTS][7];
    __shared__ PendingKnn<KN, PTS> pend;
    const int nw = st->n_worklist;
    const int sub = threadIdx.x % NL, lq = threadIdx.x / NL;
    for (int base = blockIdx.x * PTS; base < nw; base += gridDim.x * PTS) {  // block-uniform trip count
        if (threadIdx.x == 0) pend.n = 0;
        __syncthreads();
        const int w = base + lq;
        if (w < nw) {
            TopK<KN> m;
            const int s = worklist[w];
            if (!estimate_cov<KN, NL>(g, s, sub, covs[lq], &cellstack[0][threadIdx.x], NRM_THREADS, m) && sub == 0)
                pend_push(pend, s, lq, m);
        }
        __syncthreads();
        pend_finish(g, pend, max_rings, covs);
        __syncthreads();
        if (threadIdx.x < PTS && base + (int)threadIdx.x < nw)
            normal_from_cov(covs[threadIdx.x], worklist[base + threadIdx.x], normals, nflag);
        __syncthreads();
    }
    if (blockIdx.x == 0 && threadIdx.x == 0)
        atomicAdd((unsigned long long*)&st->normals_computed, (unsigned long long)nw);
}

// eager: every map point, right after a rebuild (chosen when the map is not much larger than the scan; the values are
// the same under both schedules: a normal depends on the map only).  Measured and dropped in round 2: a two-pass variant
// (ring 1 for everyone, then a dense pass over the unsettled points: 276-332 vs 210 us — at cell sizes tuned for the
// 1-NN search most points need ring 2 for their 10th neighbour) and a cell-centric variant (one wave per cell, its
// 27-cell candidates staged once in LDS, brute-force top-k per point: 137 us for the ring-1 part alone + 166 us for the
// unsettled points, vs 154 us here).
template <int KN, int NL>
__global__ __launch_bounds__(NRM_THREADS, NL == 4 ? 5 : 2) void k_normals_all(GridView g, int max_rings, float4* __restrict__ normals,
                                                             int* __restrict__ nflag) {
    constexpr int PTS = NRM_THREADS / NL;
    __shared__ int2 cellstack[ROW_STRIDE / NL][NRM_THREADS];
    __shared__ float covs[PTS][7];
    __shared__ PendingKnn<KN, PTS> pend;
    long long* stamps = (g.stamps && gridDim.x <= 8192 && NL == 4) ? g.stamps + 24 * 1024 * 4 + 4 * blockIdx.x : nullptr;  // dev
    if (threadIdx.x == 0) {
        pend.n = 0;
        if (stamps) stamps[0] = wall_clock64();
    }
    __syncthreads();
    const int lq = threadIdx.x / NL, sub = threadIdx.x % NL;
    const int s = blockIdx.x * PTS + lq;
    if (s < g.m) {
        TopK<KN> m;
        if (!estimate_cov<KN, NL>(g, s, sub, covs[lq], &cellstack[0][threadIdx.x], NRM_THREADS, m) && sub == 0)
            pend_push(pend, s, lq, m);
    }
    __syncthreads();
    if (stamps && threadIdx.x == 0) stamps[1] = wall_clock64();
    pend_finish(g, pend, max_rings, covs);
    __syncthreads();
    if (stamps && threadIdx.x == 0) stamps[2] = wall_clock64();
    const int s2 = blockIdx.x * PTS + threadIdx.x;
    if (threadIdx.x < PTS && s2 < g.m) normal_from_cov(covs[threadIdx.x], s2, normals, nflag);
    if (stamps && threadIdx.x == 0) stamps[3] = wall_clock64();
}

// The eager estimation through the neighbourhood lists (round 3; kernel of its own so that its lighter register and LDS
// footprint — 7 workgroups per CU: every workgroup of a 100 000-point map resident at once — is not set by the row walk
// above, which stays for 2 lanes per point, k > 13 and maps without lists).  Phase 1: estimate_cov_hood, 4 lanes per
// point.  Phase 2: what it does not settle (1.5 % of a LiDAR map: isolated points; everything with ties at the k-th
// distance) is finished by a whole wave each, exactly like before: the wave first rebuilds the merged (distance, index)
// list of ring 1 from the neighbourhood list, then continues with finish_cov_wave (fine ring 2, coarse level,
// exhaustive).  Phase 3: the eigen-solves on dense waves.  OWNED: the map-sharded variant (only the points whose
// spatial bucket `rank` owns, results by original index).
// ---------------------------------------------------------------------------------------------------------------------
__device__ inline int bucket_owner(float x, float y, float z, int world);

template <int KN, bool OWNED>
__global__ __launch_bounds__(NRM_THREADS, 7) void k_normals_hood(GridView g, int max_rings, int rank, int world,
                                                                 float4* __restrict__ out, int* __restrict__ nflag) {
    constexpr int PTS = NRM_THREADS / 4;
    __shared__ unsigned short sel[KN][NRM_THREADS];
    __shared__ float covs[PTS][7];
    __shared__ int pend_s[PTS], pend_lq[PTS], owned[PTS];
    __shared__ int wl[NRM_THREADS / 64][128];
    __shared__ int npend;
    long long* stamps = (g.stamps && gridDim.x <= 8192) ? g.stamps + 24 * 1024 * 4 + 4 * blockIdx.x : nullptr;  // dev
    if (threadIdx.x == 0) {
        npend = 0;
        if (stamps) stamps[0] = wall_clock64();
    }
    __syncthreads();
    const int lq = threadIdx.x >> 2, sub = threadIdx.x & 3;
    const int s = blockIdx.x * PTS + lq;
    bool mine = s < g.m;
    if (OWNED && mine) {
        const float4 P = g.pts[s];
        mine = bucket_owner(P.x, P.y, P.z, world) == rank;  // group-uniform
    }
    if (mine && !estimate_cov_hood<KN>(g, s, sub, covs[lq], &sel[0][threadIdx.x], NRM_THREADS) && sub == 0) {
        const int k = atomicAdd(&npend, 1);
        pend_s[k] = s;
        pend_lq[k] = lq;
    }
    if (sub == 0) owned[lq] = mine ? 1 : 0;
    __syncthreads();
    if (stamps && threadIdx.x == 0) stamps[1] = wall_clock64();
    {
        const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
        for (int k = wave; k < npend; k += NRM_THREADS / 64) {  // wave-uniform
            const int ps = pend_s[k];
            const float4 P = g.pts[ps];
            const int2 hh = g.rows[(size_t)g.row_of_pos[ps] * ROW_STRIDE + 27];
            TopK<KN> t, m;
            t.init();
            for (int j = lane; j < hh.y; j += 64) t.insert(point_key(g.hood[hh.x + j], P.x, P.y, P.z));
            merge_group<KN, 64>(t, m);  // = the merged list estimate_cov leaves behind ring 1
            finish_cov_wave<KN>(g, ps, lane, max_rings, m, covs[pend_lq[k]], wl[wave]);
        }
    }
    __syncthreads();
    if (stamps && threadIdx.x == 0) stamps[2] = wall_clock64();
    const int s2 = blockIdx.x * PTS + threadIdx.x;
    if (threadIdx.x < PTS && s2 < g.m && owned[threadIdx.x]) {
        float nx, ny, nz;
        const float* c = covs[threadIdx.x];
        smallest_eigenvector(c[0], c[1], c[2], c[3], c[4], c[5], nx, ny, nz);
        if (OWNED) {
            out[__float_as_int(g.pts[s2].w)] = make_float4(nx, ny, nz, 1.f);
        } else {
            out[s2] = make_float4(nx, ny, nz, 1.f);
            nflag[s2] = 1;
        }
    }
    if (stamps && threadIdx.x == 0) stamps[3] = wall_clock64();
}

// ---------------------------------------------------------------------------------------------------------------------
// Map-sharded normal estimation (multi-GPU, SURVEY.md §8e "Map-sharded"; BASELINE configs[3]: 1M-point map on 8 GPUs).
// Every rank holds the whole map and its grid (the 1-NN queries stay local), but estimates the normals only of the
// points whose spatial bucket it owns; the results travel BY ORIGINAL MAP INDEX (the cell-sorted order differs from rank
// to rank: the counting sort's within-cell order is not deterministic), are summed over the ranks by one collective
// (every index has exactly one owner, the other ranks contribute zeros: the sum is exact) and scattered back into the
// rank's own cell-sorted normal cache.  Ownership must not depend on anything rank-local (the auto-tuned cell edge is):
// it is the hash of the point's 1-metre bucket.
// ---------------------------------------------------------------------------------------------------------------------
__device__ inline int bucket_owner(float x, float y, float z, int world) {
    const unsigned long long key = pack_cell(cell_coord(x, 1.0f), cell_coord(y, 1.0f), cell_coord(z, 1.0f));
    return (int)(hash_cell(key) % (unsigned)world);
}

// ---------------------------------------------------------------------------------------------------------------------
// The eager estimation through the neighbourhood lists with TWO lanes per map point and ONE pass over the list (round 4;
// "hoods" 2).  Round 3's k_normals_hood gives a point four lanes and walks its list three times (distances -> T, members,
// covariance) with two butterfly merges in between: ~2 600 VALU instructions per wave of 16 points.  Here the two lanes of
// a point stream alternate groups of four entries, sixteen entries in flight each, into KN + 1 sorted 32-bit keys
// (distance bits, the low 11 bits replaced by the entry's number; min + med3 per slot — the device of search_ball_lane);
// one butterfly step (element-wise min of one list with the other reversed = the KN + 1 smallest of both) merges them.
// The keys name the KN nearest entries exactly whenever key KN differs from key KN - 1 in the bits above the number (every
// other entry then lies a whole truncation step farther than all KN), and ring 1 certifies them when the KN-th distance —
// bounded by its key with the low bits set — is within h + edge.  The covariance sums are order-independent (CovSums), so
// the members need no order and the two lanes split them.  (One lane per point was built first: 43 us — 1 563 waves on
// 1 024 SIMDs, each a chain of eight list trips, the Jacobi solve behind it.)
// Anything else (an uncertified KN-th neighbour: ~0.2 % of a LiDAR map; two entries within 2^-12 of each other at the
// KN-th place; lists beyond 2 048 entries) is finished by a whole wave of the workgroup (finish_cov_wave, as before).
// (Tried and dropped: the stragglers in a chip-wide queue — a launch of its own behind the main one: +31 us, a chain of
// dependent probes per point whoever runs it; drained inside the launch by the waves that are done with ring 1: a shared
// head counter serialises a thousand waves on one address, 7 ms.)
// ---------------------------------------------------------------------------------------------------------------------
static constexpr unsigned HOOD_JMASK = 2047u;
static constexpr int TAIL_HEADER = 2;  // words in front of the stragglers' list: their count, the workgroups of the tail launch that are through
static constexpr int NRM2_THREADS = 128;  // 64 map points x 2 lanes

template <int N>
struct TopKeys {
    unsigned k[N];  // ascending
    __device__ inline void init() {
#pragma unroll
        for (int i = 0; i < N; ++i) k[i] = ~0u;
    }
    __device__ inline void insert(unsigned c) {
#pragma unroll
        for (int i = N - 1; i > 0; --i) k[i] = umed3(k[i - 1], k[i], c);  // (the old k[i - 1]: going down)
        k[0] = min(k[0], c);
    }
};

// the two lanes of a point (sub = 0 / 1, neighbours in a DPP quad); true: lane 0 has written the covariance
template <int KN>
__device__ inline bool cov_hood_pair(const GridView& g, int s, int sub, float* __restrict__ cov) {
    const float4 P = g.pts[s];
    const float px = P.x, py = P.y, pz = P.z;
    const float h = g.h;
    const int cx = cell_coord(px, g.inv_h), cy = cell_coord(py, g.inv_h), cz = cell_coord(pz, g.inv_h);
    const float fx = fminf(fmaxf(px - (float)cx * h, 0.f), h);
    const float fy = fminf(fmaxf(py - (float)cy * h, 0.f), h);
    const float fz = fminf(fmaxf(pz - (float)cz * h, 0.f), h);
    const float edge = fminf(fminf(fminf(fx, h - fx), fminf(fy, h - fy)), fminf(fz, h - fz));
    const int2 hh = g.rows[(size_t)g.row_of_pos[s] * ROW_STRIDE + 27];
    const int n = hh.y, n4 = (n + 3) & ~3;  // (runs are padded to whole groups of four with points at +inf)
    if (n < KN || n4 > (int)HOOD_JMASK + 1) return false;  // pair-uniform
    const float4* __restrict__ H = g.hood + hh.x;
    const float4* __restrict__ pad = g.pts + g.m;
    TopKeys<KN + 1> t;
    t.init();
    for (int j = 0; j < n4; j += 32) {  // pair-uniform trip count: the DPP exchange below needs both lanes
        float4 q[4][4];
        int first[4];
#pragma unroll
        for (int gr = 0; gr < 4; ++gr) {
            first[gr] = j + 8 * gr + 4 * sub;
            const float4* __restrict__ a = first[gr] < n4 ? H + first[gr] : pad;
            q[gr][0] = a[0];
            q[gr][1] = a[1];
            q[gr][2] = a[2];
            q[gr][3] = a[3];
        }
#pragma unroll
        for (int gr = 0; gr < 4; ++gr) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const float dx = q[gr][i].x - px, dy = q[gr][i].y - py, dz = q[gr][i].z - pz;
                const float d2 = fmaf(dz, dz, fmaf(dy, dy, dx * dx));
                t.insert((__float_as_uint(d2) & ~HOOD_JMASK) | (unsigned)((first[gr] + i) & (int)HOOD_JMASK));
            }
        }
    }
    // the KN + 1 smallest of both lanes' lists, sorted again (identical in the two lanes)
    TopKeys<KN + 1> m;
    m.init();
#pragma unroll
    for (int i = 0; i <= KN; ++i) m.insert(min(t.k[i], (unsigned)quad_xor<1>((int)t.k[KN - i])));
    if (m.k[KN - 1] >= 0x7f800000u) return false;  // fewer than KN finite distances
    // ring 1 certifies the KN-th neighbour ...
    const float bound = h + edge;
    if (!(__uint_as_float(m.k[KN - 1] | HOOD_JMASK) <= bound * bound * 0.999999f)) return false;
    // ... and the keys name the KN nearest when everybody else is a whole truncation step farther
    if ((m.k[KN] & ~HOOD_JMASK) == (m.k[KN - 1] & ~HOOD_JMASK)) return false;
    CovSums cs;
    cs.zero();
#pragma unroll
    for (int i = 0; i < KN; ++i) {  // (the nearest — the point itself or a twin at distance 0 — adds nothing: :407)
        if ((i & 1) == sub) {
            const float4 q = H[m.k[i] & HOOD_JMASK];
            cs.add(q.x - px, q.y - py, q.z - pz);
        }
    }
#pragma unroll
    for (int k = 0; k < 6; ++k) cs.c[k] += quad_xor<1>(cs.c[k]);  // (exact sums of multiples of 2^-40: any order)
    if (sub == 0) cs.store(KN - 1, cov);
    return true;
}

// one straggler of the pair pass by a whole wave: the merged list of ring 1 from the cell's neighbourhood list, then
// finish_cov_wave (fine ring 2, coarse level, exhaustive) and the eigen-solve; `wcov` / `wl` = the wave's LDS scratch
// `cov_out` (round 6): the covariance is left there (LDS, six floats) and nothing is solved — the caller's dense pass of
// eigen-solves takes the point with the others (the Jacobi solve is a 5 us chain on one lane: per straggler here, once per
// workgroup there)
template <int KN, bool OWNED>
__device__ inline void normal_of_straggler(const GridView& g, int ps, int lane, int max_rings, float* __restrict__ wcov,
                                           int* __restrict__ wl, float4* __restrict__ out, int* __restrict__ nflag,
                                           float* __restrict__ cov_out = nullptr) {
    const float4 P = g.pts[ps];
    const int2 hh = g.rows[(size_t)g.row_of_pos[ps] * ROW_STRIDE + 27];
    TopK<KN> t, m;
    t.init();
    for (int j = lane; j < hh.y; j += 64) t.insert(point_key(g.hood[hh.x + j], P.x, P.y, P.z));
    merge_group<KN, 64>(t, m);  // = the merged list estimate_cov leaves behind ring 1
    // (round 6) three stragglers in ten are points of DENSE cells whose KN-th and KN + 1-th neighbours the pair pass's
    // truncated keys could not tell apart: the exact keys settle them inside ring 1, no ring 2
    const float h = g.h;
    const float fx = fminf(fmaxf(P.x - (float)cell_coord(P.x, g.inv_h) * h, 0.f), h);
    const float fy = fminf(fmaxf(P.y - (float)cell_coord(P.y, g.inv_h) * h, 0.f), h);
    const float fz = fminf(fmaxf(P.z - (float)cell_coord(P.z, g.inv_h) * h, 0.f), h);
    const float bound1 = h + fminf(fminf(fminf(fx, h - fx), fminf(fy, h - fy)), fminf(fz, h - fz));
    if (m.kth() <= bound1 * bound1 * 0.999999f) {  // (wave-uniform: m is the same in every lane)
        if (lane == 0) neighbourhood_cov<KN>(g, P.x, P.y, P.z, m, wcov);
    } else {
        finish_cov_wave<KN>(g, ps, lane, max_rings, m, wcov, wl);
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    if (cov_out) {
        if (lane < 6) cov_out[lane] = wcov[lane];
    } else if (lane == 0) {
        float nx, ny, nz;
        smallest_eigenvector(wcov[0], wcov[1], wcov[2], wcov[3], wcov[4], wcov[5], nx, ny, nz);
        if (OWNED) {
            out[__float_as_int(P.w)] = make_float4(nx, ny, nz, 1.f);
        } else {
            out[ps] = make_float4(nx, ny, nz, 1.f);
            nflag[ps] = 1;
        }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();  // (the next point overwrites the wave's scratch)
}

// `tail` (round 5, "normals_tail_stream"): the stragglers are not finished here but appended to a chip-wide list —
// tail[0] = their count, tail[1..] = their positions — which k_normals_tail works through on the context's map stream, beside
// the next frame's preprocessing on the caller's stream (api.hip: DeviceGuard's join orders every reader of the normals behind
// it).  A launch of its own behind this one on the SAME stream was tried in round 4 (+31 us: the 30 us chain per straggler sets
// the duration of whatever launch runs it); on a stream of its own that chain has ~50 us of independent work to hide behind.
template <int KN, bool OWNED, bool LIST = false>
__global__ __launch_bounds__(NRM2_THREADS) void k_normals_hood2(GridView g, int max_rings, int rank, int world,
                                                                float4* __restrict__ out, int* __restrict__ nflag,
                                                                int* __restrict__ tail) {
    constexpr int PTS = NRM2_THREADS / 2;
    __shared__ float covs[PTS][7];
    __shared__ int settled[PTS];
    __shared__ int pend_s[PTS];
    __shared__ int npend;
    __shared__ int wl[NRM2_THREADS / 64][128];
    __shared__ float wcov[NRM2_THREADS / 64][8];
    if (threadIdx.x == 0) npend = 0;
    __syncthreads();
    const int lq = threadIdx.x >> 1, sub = threadIdx.x & 1;
    const int s = blockIdx.x * PTS + lq;
    bool mine = s < g.m;
    if (OWNED && mine) {
        const float4 P = g.pts[s];
        mine = bucket_owner(P.x, P.y, P.z, world) == rank;  // pair-uniform
    }
    bool ok = false;
    if (mine) ok = cov_hood_pair<KN>(g, s, sub, covs[lq]);
    if (sub == 0) {
        settled[lq] = (mine && ok) ? 1 : 0;
        if (mine && !ok) pend_s[atomicAdd(&npend, 1)] = s;
    }
    __syncthreads();
    if constexpr (LIST) {  // the stragglers go to the chip-wide list (k_normals_tail16 / k_normals_tail behind this launch)
        if (npend > 0) {
            __shared__ int tail_base;
            if (threadIdx.x == 0) tail_base = atomicAdd(&tail[0], npend);
            __syncthreads();
            if ((int)threadIdx.x < npend) tail[TAIL_HEADER + tail_base + threadIdx.x] = pend_s[threadIdx.x];
        }
    } else if (npend > 0) {  // (block-uniform)
        // ---- the stragglers of this workgroup, a wave per point: their covariances join the others' (round 6: in front of
        // the dense pass below, which solves them with the rest — the solve was a 5 us chain on one lane per straggler)
        const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
        for (int k = wave; k < npend; k += NRM2_THREADS / 64) {  // wave-uniform
            const int ps = pend_s[k], slot = ps - blockIdx.x * PTS;
            normal_of_straggler<KN, OWNED>(g, ps, lane, max_rings, wcov[wave], wl[wave], out, nflag, covs[slot]);
            if (lane == 0) settled[slot] = 1;
        }
        __syncthreads();
    }
    if (threadIdx.x < PTS && settled[threadIdx.x]) {  // the eigen-solves on a dense wave
        const int s2 = blockIdx.x * PTS + threadIdx.x;
        float nx, ny, nz;
        const float* c = covs[threadIdx.x];
        smallest_eigenvector(c[0], c[1], c[2], c[3], c[4], c[5], nx, ny, nz);
        if (OWNED) {
            out[__float_as_int(g.pts[s2].w)] = make_float4(nx, ny, nz, 1.f);
        } else {
            out[s2] = make_float4(nx, ny, nz, 1.f);
            nflag[s2] = 1;
        }
    }
}

// The stragglers of k_normals_hood2<.., LIST> — map points whose KN-th neighbour the pair pass does not certify, 0.3 % of a
// LiDAR map: 70 % of them in sparse cells (ring 2 needed), the rest points of dense cells whose KN-th and KN + 1-th neighbours
// the truncated 32-bit keys cannot tell apart — in a launch of their own, SIXTEEN lanes per point (round 6; option
// "normals_list", OFF by default).  Inside k_normals_hood2 each is finished by a whole wave of the workgroup that met it (ring 1
// again with exact keys, a 64-lane merge through the LDS crossbar, the cells of ring 2 the KN-th distance reaches, another
// merge, ten dependent loads and the Jacobi solve on one lane): the launch lasts 62 us at the benchmark sizes where the pair
// pass alone takes 37 (88 against 42 us on the published configuration's 181 695-point map).  Here lane l of a row of 16 takes
// entries l, l + 16, .. of the cell's neighbourhood list with exact 64-bit keys, the sixteen sorted lists are merged by DPP
// row exchanges (no LDS); what ring 1 does not certify probes the shell cells of ring 2 its KN-th distance reaches (seven
// hashed probes per lane, grid_lookup7), scans them and merges again; the members' covariance terms are formed by lanes
// 1 .. KN - 1, one member each, and summed over the row (order-independent sums: CovSums); lane 0 solves.  What ring 2 does
// not certify either (a point or two per frame) and lists beyond 2 048 entries take the whole-wave path, wave-uniformly,
// inside this launch.  Same neighbours, same sums: the same bits (tests: `normals_list` variants).
// MEASURED (profiles/r06_normals_stragglers.txt): the pair pass without its stragglers 37 us (96 registers: occupancy 5) —
// and this launch 55 us behind it (35 at best): 92 against 62.  A straggler is a chain — list, merge, probes, cells, merge,
// members, Jacobi: 13 us for ring 1, 27 more for ring 2 and the members, 5.5 for the solve — whoever runs it; inside the pair
// kernel that chain starts when a workgroup's pair pass ends and overlaps the other workgroups' passes, in a launch of its own
// it starts when ALL of them have ended.  Sixteen lanes walk a cell list four times as long as 64 do.  Kept behind its option
// as the record of the experiment; the published configuration hides its normals behind the host's upload of the next frame
// either way (frame time 0.417 -> 0.410 ms with NO stragglers at all).
// tail[0] = count, tail[1] = workgroups of this launch that are through (the last one zeroes both: the list is ready for the
// next build without a memset launch), positions from tail[TAIL_HEADER].
static constexpr int TAIL16_THREADS = 256;

// shell cell number s (0 .. 97) of ring 2 -> its offsets: the two 5 x 5 faces z = -2 / +2, then for z = -1, 0, 1 the 16 cells
// around the 3 x 3 interior
__device__ inline void ring2_shell_cell(int s, int& ox, int& oy, int& oz) {
    if (s < 50) {
        oz = s < 25 ? -2 : 2;
        const int xy = s < 25 ? s : s - 25;
        ox = xy % 5 - 2;
        oy = xy / 5 - 2;
    } else {
        const int t = s - 50, r = t & 15;
        oz = (t >> 4) - 1;
        if (r < 5) {
            ox = r - 2;
            oy = -2;
        } else if (r < 10) {
            ox = r - 7;
            oy = 2;
        } else {
            ox = ((r - 10) & 1) ? 2 : -2;
            oy = ((r - 10) >> 1) - 1;
        }
    }
}

// true: lane 0 of the row holds the covariance in cov6[0..5]
template <int KN>
__device__ inline bool cov_ring2_row16(const GridView& g, int ps, int sub, float (&cov6)[6]) {
    const float4 P = g.pts[ps];
    const float px = P.x, py = P.y, pz = P.z;
    const float h = g.h;
    const int cx = cell_coord(px, g.inv_h), cy = cell_coord(py, g.inv_h), cz = cell_coord(pz, g.inv_h);
    const float fx = fminf(fmaxf(px - (float)cx * h, 0.f), h);
    const float fy = fminf(fmaxf(py - (float)cy * h, 0.f), h);
    const float fz = fminf(fmaxf(pz - (float)cz * h, 0.f), h);
    const float edge = fminf(fminf(fminf(fx, h - fx), fminf(fy, h - fy)), fminf(fz, h - fz));
    const int2 hh = g.rows[(size_t)g.row_of_pos[ps] * ROW_STRIDE + 27];
    if (hh.y > 2048) return false;  // (row-uniform) the wave path
    if (g.dbg && sub == 0) atomicAdd(&g.dbg[7], 1);  // ("knn: beyond ring 1")
    TopK<KN> t, m;
    t.init();
    {  // ring 1: the cell's neighbourhood list with exact 64-bit keys, four entries of the lane in flight
        const float4* __restrict__ H = g.hood + hh.x;
        const int n = hh.y;
        for (int j = sub; j < n; j += 64) {
            const bool v1 = j + 16 < n, v2 = j + 32 < n, v3 = j + 48 < n;
            const float4 q0 = H[j], q1 = H[v1 ? j + 16 : j], q2 = H[v2 ? j + 32 : j], q3 = H[v3 ? j + 48 : j];
            t.insert(point_key(q0, px, py, pz));
            if (v1) t.insert(point_key(q1, px, py, pz));
            if (v2) t.insert(point_key(q2, px, py, pz));
            if (v3) t.insert(point_key(q3, px, py, pz));
        }
    }
    merge_group<KN, 16>(t, m);
    // most stragglers are points of DENSE cells whose KN-th and KN + 1-th neighbours the pair pass's truncated 32-bit keys
    // could not tell apart: the exact keys settle them inside ring 1
    const float bound1 = h + edge;
    if (!(m.kth() <= bound1 * bound1 * 0.999999f)) {  // (row-uniform: m is the same in the sixteen lanes)
        // the shell of ring 2: the cells whose box the KN-th distance so far reaches, seven hashed probes per lane in flight
        const float kth = m.kth();
        int2 found[7];
        int want = 0;
#pragma unroll
        for (int k = 0; k < 7; ++k) {
            if (sub + 16 * k < 98) {
                int ox, oy, oz;
                ring2_shell_cell(sub + 16 * k, ox, oy, oz);
                const float gx = axis_gap(ox, fx, h), gy = axis_gap(oy, fy, h), gz = axis_gap(oz, fz, h);
                if (!(fmaf(gx, gx, fmaf(gy, gy, gz * gz)) > kth)) want |= 1 << k;
            }
        }
        grid_lookup7(g, want, [&](int k) {
            int ox, oy, oz;
            ring2_shell_cell(sub + 16 * k, ox, oy, oz);
            return pack_cell(cx + ox, cy + oy, cz + oz);
        }, found);
        if (sub == 0) {
            t = m;  // (what ring 1 found rides in lane 0's list)
        } else {
            t.init();
        }
#pragma unroll
        for (int k = 0; k < 7; ++k)
            if (found[k].y > 0) scan_cell_knn<KN>(g, found[k].x, found[k].y, px, py, pz, t);
        merge_group<KN, 16>(t, m);
        // ring 2 certifies the KN-th neighbour (finite: the map holds KN points within its reach)
        const float bound = 2.0f * h + edge;
        if (!(m.kth() <= bound * bound * 0.999999f)) return false;
    }
    // the covariance (neighbourhood_cov's terms: members 1 .. KN - 1, the nearest — the point itself or a twin — dropped): one
    // member per lane, summed over the row
    CovSums cs;
    cs.zero();
    unsigned long long mine = KEY_EMPTY;
#pragma unroll
    for (int i = 1; i < KN; ++i)
        if (i % 16 == sub) mine = m.key[i];  // (KN - 1 <= 15 members: lanes 1 .. KN - 1)
    static_assert(KN <= 16, "one member per lane of the row");
    if (sub >= 1 && sub < KN) {
        const float4 q = g.pts[g.pos_of_orig[key_idx(mine)]];
        cs.add(q.x - px, q.y - py, q.z - pz);
    }
#pragma unroll
    for (int k = 0; k < 6; ++k) cs.c[k] = row16_sum(cs.c[k]);
    cs.store(KN - 1, cov6);
    return true;
}

template <int KN>
__global__ __launch_bounds__(TAIL16_THREADS) void k_normals_tail16(GridView g, int max_rings, int* __restrict__ tail,
                                                                   float4* __restrict__ out, int* __restrict__ nflag) {
    __shared__ int wl[TAIL16_THREADS / 64][128];
    __shared__ float wcov[TAIL16_THREADS / 64][8];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, row = lane >> 4, sub = lane & 15;
    const int count = tail[0];
    const int waves = gridDim.x * (TAIL16_THREADS / 64);
    for (int k0 = 4 * (blockIdx.x * (TAIL16_THREADS / 64) + wave); k0 < count; k0 += 4 * waves) {  // wave-uniform
        const int k = k0 + row;
        const bool mine = k < count;  // row-uniform
        const int ps = mine ? tail[TAIL_HEADER + k] : 0;
        bool ok = !mine;
        if (mine) {
            float cov6[6];
            ok = cov_ring2_row16<KN>(g, ps, sub, cov6);
            if (ok && sub == 0) {
                float nx, ny, nz;
                smallest_eigenvector(cov6[0], cov6[1], cov6[2], cov6[3], cov6[4], cov6[5], nx, ny, nz);
                out[ps] = make_float4(nx, ny, nz, 1.f);
                nflag[ps] = 1;
            }
        }
        // what the rows could not settle: the whole wave, one point after the other (wave-uniform)
        const unsigned long long failed = __ballot(!ok);
        for (int r = 0; r < 4; ++r) {
            if (!((failed >> (16 * r)) & 1ull)) continue;
            const int ps2 = __builtin_amdgcn_readlane(ps, 16 * r);
            normal_of_straggler<KN, false>(g, ps2, lane, max_rings, wcov[wave], wl[wave], out, nflag);
        }
    }
    // the last workgroup through empties the list for the next build
    __syncthreads();
    if (threadIdx.x == 0) {
        __threadfence();
        if (atomicAdd(&tail[1], 1) == (int)gridDim.x - 1) {
            tail[0] = 0;
            tail[1] = 0;
        }
    }
}

// the stragglers of k_normals_hood2 (`tail` list), a wave per point, on a stream of their own ("normals_tail_stream")
template <int KN>
__global__ __launch_bounds__(NRM2_THREADS) void k_normals_tail(GridView g, int max_rings, int* __restrict__ tail,
                                                               float4* __restrict__ out, int* __restrict__ nflag) {
    __shared__ int wl[NRM2_THREADS / 64][128];
    __shared__ float wcov[NRM2_THREADS / 64][8];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int count = tail[0], waves = gridDim.x * (NRM2_THREADS / 64);
    for (int k = blockIdx.x * (NRM2_THREADS / 64) + wave; k < count; k += waves)  // wave-uniform
        normal_of_straggler<KN, false>(g, tail[TAIL_HEADER + k], lane, max_rings, wcov[wave], wl[wave], out, nflag);
    __syncthreads();
    if (threadIdx.x == 0) {
        __threadfence();
        if (atomicAdd(&tail[1], 1) == (int)gridDim.x - 1) {
            tail[0] = 0;
            tail[1] = 0;
        }
    }
}

template <int KN, int NL>
__global__ __launch_bounds__(NRM_THREADS) void k_normals_owned(GridView g, int max_rings, int rank, int world,
                                                               float4* __restrict__ by_index) {
    constexpr int PTS = NRM_THREADS / NL;
    __shared__ int2 cellstack[ROW_STRIDE / NL][NRM_THREADS];
    __shared__ float covs[PTS][7];
    __shared__ int owned[PTS];
    __shared__ PendingKnn<KN, PTS> pend;
    if (threadIdx.x == 0) pend.n = 0;
    __syncthreads();
    const int lq = threadIdx.x / NL, sub = threadIdx.x % NL;
    const int s = blockIdx.x * PTS + lq;
    bool mine = false;
    if (s < g.m) {
        const float4 P = g.pts[s];
        mine = bucket_owner(P.x, P.y, P.z, world) == rank;  // group-uniform
        if (mine) {
            TopK<KN> m;
            if (!estimate_cov<KN, NL>(g, s, sub, covs[lq], &cellstack[0][threadIdx.x], NRM_THREADS, m) && sub == 0)
                pend_push(pend, s, lq, m);
        }
    }
    if (sub == 0) owned[lq] = mine ? 1 : 0;
    __syncthreads();
    pend_finish(g, pend, max_rings, covs);
    __syncthreads();
    const int s2 = blockIdx.x * PTS + threadIdx.x;
    if (threadIdx.x < PTS && s2 < g.m && owned[threadIdx.x]) {
        float nx, ny, nz;
        const float* c = covs[threadIdx.x];
        smallest_eigenvector(c[0], c[1], c[2], c[3], c[4], c[5], nx, ny, nz);
        by_index[__float_as_int(g.pts[s2].w)] = make_float4(nx, ny, nz, 1.f);
    }
}

// normals by original index -> this rank's cell-sorted cache
__global__ void k_normals_install(const float4* __restrict__ pts, int m, const float4* __restrict__ by_index,
                                  float4* __restrict__ normals, int* __restrict__ nflag) {
    const int s = blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= m) return;
    const float4 n = by_index[__float_as_int(pts[s].w)];
    normals[s] = make_float4(n.x, n.y, n.z, 1.f);
    nflag[s] = 1;
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
    g.rows = ctx->rows.as<int2>();
    g.row_of_pos = ctx->row_of_pos.as<int>();
    g.ctable = ctx->ctable_ptr;
    g.cmask = ctx->ctable_size ? ctx->ctable_size - 1 : 0;
    g.ch = ctx->cell_h * COARSE_FACTOR;
    g.cinv_h = 1.0f / g.ch;
    g.cpts = ctx->csorted.as<float4>();
    g.pos_of_orig = ctx->pos_of_orig.as<int>();
    g.flat_rows = ctx->flat_rows;
    g.prune_guard = ctx->prune_guard;
    g.hood = ctx->hoods_valid ? ctx->hood.as<float4>() : nullptr;
    g.dbg = ctx->search_stats == 1 ? ctx->dbg_counts.as<int>() : nullptr;
    g.stamps = ctx->search_stats ? reinterpret_cast<long long*>(ctx->dbg_counts.as<int>() + 16) : nullptr;
    return g;
}

static void launch_search_rows(icp_ctx* ctx, int n, int mode, int transform, int seeded) {
    hipLaunchKernelGGL(k_search_rows, dim3((unsigned)(((long long)n * 4 + 255) / 256)), dim3(256), 0, ctx->stream,
                       make_view(ctx), ctx->tgt4.as<float4>(), n, mode, transform, reg_state(ctx), ctx->cfg.max_rings,
                       ctx->nn_pos.as<int>(), ctx->nflag.as<int>(), ctx->worklist.as<int>(),
                       (ctx->normals_ready || ctx->cost == ICP_COST_POINT_TO_POINT) ? 0 : 1,  // p2p needs no normals
                       seeded);
}

int launch_search(icp_ctx* ctx) {
    const int n = (int)ctx->tgt_n;
    if (n <= 0) return ICP_OK;
    const int tok = prof_begin(ctx, 0);
    // from the second search of a registration on, nn_pos describes these targets against this grid: seeds
    const int seeded = (ctx->use_nn_cache > 1 && ctx->searches_in_registration > 0 && ctx->nn_pos_n == n &&
                        ctx->nn_pos_gen == ctx->grid_gen) ? 1 : 0;
    launch_search_rows(ctx, n, ctx->tgt_mode, 1, seeded);
    prof_end(ctx, tok);
    ICP_HIP(ctx, hipGetLastError());
    ctx->searches_in_registration += 1;
    ctx->nn_pos_n = n;
    ctx->nn_pos_gen = ctx->grid_gen;
    return ICP_OK;
}

// search without pose transform (LocalMap.nearest_neighbor_search seam): the state must have done = 0
int launch_search_raw(icp_ctx* ctx) {
    const int n = (int)ctx->tgt_n;
    if (n <= 0) return ICP_OK;
    launch_search_rows(ctx, n, ICP_TARGETS_ALL, 0, 0);
    ICP_HIP(ctx, hipGetLastError());
    ctx->nn_pos_n = -1;  // (other queries: no seeds for a registration)
    return ICP_OK;
}

// fine rings tried by the kNN before it moves to the coarse level: ring 3 means 218 hashed probes for a handful of extra
// candidates, the coarse level reaches the same points through a few 4x larger cells (measured: 233 -> 199 us per
// 100k-point map with 2 instead of 3).  Option "knn_rings" overrides.
static int knn_fine_rings(const icp_ctx* ctx) {
    if (ctx->knn_rings >= 0) return ctx->knn_rings;
    return ctx->cfg.max_rings < 2 ? ctx->cfg.max_rings : 2;
}

// the worklist kernel for the three compiled neighbourhood sizes
template <int NL>
static void launch_worklist_t(icp_ctx* ctx, int kn, const GridView& g, RegState* st, int blocks) {
    const int rings = knn_fine_rings(ctx);
    const int* wl = ctx->worklist.as<int>();
    float4* nrm = ctx->normals.as<float4>();
    int* nf = ctx->nflag.as<int>();
    if (kn == 11)
        hipLaunchKernelGGL((k_normals<11, NL>), dim3(blocks), dim3(NRM_THREADS), 0, ctx->stream, g, st, wl,
                           rings, nrm, nf);
    else if (kn == 6)
        hipLaunchKernelGGL((k_normals<6, NL>), dim3(blocks), dim3(NRM_THREADS), 0, ctx->stream, g, st, wl,
                           rings, nrm, nf);
    else
        hipLaunchKernelGGL((k_normals<21, NL>), dim3(blocks), dim3(NRM_THREADS), 0, ctx->stream, g, st, wl,
                           rings, nrm, nf);
}

static int worklist_blocks(int64_t cap, int nl) {
    const int pts = NRM_THREADS / nl;
    int64_t blocks = (cap + pts - 1) / pts;
    if (blocks < 1) blocks = 1;
    if (blocks > 4096) blocks = 4096;
    return (int)blocks;
}

template <int NL>
static void launch_normals_all_t(icp_ctx* ctx, int kn, const GridView& g) {
    const int blocks = (int)((ctx->map_m + NRM_THREADS / NL - 1) / (NRM_THREADS / NL));
    const int rings = knn_fine_rings(ctx);
    float4* nrm = ctx->normals.as<float4>();
    int* nf = ctx->nflag.as<int>();
    if (NL == 4 && g.hood && ctx->hoods >= 2 && (kn == 11 || kn == 6)) {  // two lanes per point, one pass over the list
        const int b2 = (int)((ctx->map_m + NRM2_THREADS / 2 - 1) / (NRM2_THREADS / 2));
        int* tail = ctx->normals_tail_list;  // (set by launch_normals_all: the stragglers go to a list and a launch of their own)
        if (!tail) {  // "normals_list" 0: every workgroup finishes its own stragglers, a wave each (rounds 4-5)
            if (kn == 11)
                hipLaunchKernelGGL((k_normals_hood2<11, false>), dim3(b2), dim3(NRM2_THREADS), 0, ctx->stream, g, rings, 0, 1, nrm,
                                   nf, tail);
            else
                hipLaunchKernelGGL((k_normals_hood2<6, false>), dim3(b2), dim3(NRM2_THREADS), 0, ctx->stream, g, rings, 0, 1, nrm,
                                   nf, tail);
            return;
        }
        if (kn == 11)
            hipLaunchKernelGGL((k_normals_hood2<11, false, true>), dim3(b2), dim3(NRM2_THREADS), 0, ctx->stream, g, rings, 0, 1, nrm,
                               nf, tail);
        else
            hipLaunchKernelGGL((k_normals_hood2<6, false, true>), dim3(b2), dim3(NRM2_THREADS), 0, ctx->stream, g, rings, 0, 1, nrm,
                               nf, tail);
        if (ctx->normals_tail_on_map_stream) {  // "normals_tail_stream": a wave per straggler on the map stream, behind the launch above
            (void)hipEventRecord(ctx->map_start_event, ctx->stream);
            (void)hipStreamWaitEvent(ctx->map_stream, ctx->map_start_event, 0);
            int tb = b2 / 2;  // (a sparse map is all stragglers: as many waves as the pair pass had workgroups; at least 64)
            if (tb < 64) tb = 64;
            if (tb > 1024) tb = 1024;
            if (kn == 11)
                hipLaunchKernelGGL((k_normals_tail<11>), dim3(tb), dim3(NRM2_THREADS), 0, ctx->map_stream, g, rings, tail, nrm, nf);
            else
                hipLaunchKernelGGL((k_normals_tail<6>), dim3(tb), dim3(NRM2_THREADS), 0, ctx->map_stream, g, rings, tail, nrm, nf);
            (void)hipEventRecord(ctx->map_done_event, ctx->map_stream);
            ctx->map_stream_busy = true;
        } else {  // sixteen lanes per straggler, right behind (a LiDAR map leaves 0.3 % of its points: a few hundred rows of 16)
            int tb = b2 / 16;  // (a sparse map is all stragglers: 16 rows per workgroup, a few trips each)
            if (tb < 64) tb = 64;
            if (tb > 1024) tb = 1024;
            if (kn == 11)
                hipLaunchKernelGGL((k_normals_tail16<11>), dim3(tb), dim3(TAIL16_THREADS), 0, ctx->stream, g, rings, tail, nrm, nf);
            else
                hipLaunchKernelGGL((k_normals_tail16<6>), dim3(tb), dim3(TAIL16_THREADS), 0, ctx->stream, g, rings, tail, nrm, nf);
        }
        return;
    }
    if (NL == 4 && g.hood && (kn == 11 || kn == 6)) {  // through the neighbourhood lists, four lanes per point (round 3)
        if (kn == 11)
            hipLaunchKernelGGL((k_normals_hood<11, false>), dim3(blocks), dim3(NRM_THREADS), 0, ctx->stream, g, rings, 0, 1,
                               nrm, nf);
        else
            hipLaunchKernelGGL((k_normals_hood<6, false>), dim3(blocks), dim3(NRM_THREADS), 0, ctx->stream, g, rings, 0, 1,
                               nrm, nf);
        return;
    }
    if (kn == 11)
        hipLaunchKernelGGL((k_normals_all<11, NL>), dim3(blocks), dim3(NRM_THREADS), 0, ctx->stream, g, rings, nrm, nf);
    else if (kn == 6)
        hipLaunchKernelGGL((k_normals_all<6, NL>), dim3(blocks), dim3(NRM_THREADS), 0, ctx->stream, g, rings, nrm, nf);
    else
        hipLaunchKernelGGL((k_normals_all<21, NL>), dim3(blocks), dim3(NRM_THREADS), 0, ctx->stream, g, rings, nrm, nf);
}

int launch_normals_all(icp_ctx* ctx, bool tail_may_overlap) {
    const int kn = ctx->cfg.num_neighbors_normals + 1;
    if (ctx->normals_ready || ctx->map_m <= 0) return ICP_OK;
    if (kn != 11 && kn != 6 && kn != 21) return ICP_OK;  // generic k stays lazy
    GridView g = make_view(ctx);
    // "normals_list" (round 6, default): the stragglers of the two-lane kernel — points ring 1 does not certify — go to a list
    // and a launch of their own right behind it (k_normals_tail16).  "normals_tail_stream": behind a map update (nothing reads
    // a normal before the next entry point that joins the map stream) that launch is round 5's wave-per-straggler kernel on
    // the map stream, beside the next frame's preprocessing (measured neutral then; kept for comparison)
    ctx->normals_tail_list = nullptr;
    ctx->normals_tail_on_map_stream = false;
    if (ctx->knn_lanes != 2 && g.hood && ctx->hoods >= 2 && (kn == 11 || kn == 6)) {
        const bool other_stream = tail_may_overlap && ctx->normals_tail_stream && !ctx->exchange_on && !ctx->prof.enabled &&
                                  !ctx->search_stats && ctx->stream != ctx->map_stream;
        if (other_stream && !ctx->map_stream) {
            ICP_HIP(ctx, hipStreamCreateWithFlags(&ctx->map_stream, hipStreamNonBlocking));
            ICP_HIP(ctx, hipEventCreateWithFlags(&ctx->map_done_event, hipEventDisableTiming));
            ICP_HIP(ctx, hipEventCreateWithFlags(&ctx->map_start_event, hipEventDisableTiming));
        }
        if (other_stream || ctx->normals_list) {
            const size_t need = ((size_t)ctx->map_m + TAIL_HEADER) * sizeof(int);
            if (ctx->normals_tail.bytes < need) {  // (a fresh list: its header zeroed once — every tail launch leaves it zeroed)
                ICP_HIP(ctx, ctx->normals_tail.reserve(need + need / 4));  // (hipFree of the old block synchronises the device)
                ICP_HIP(ctx, hipMemsetAsync(ctx->normals_tail.ptr, 0, TAIL_HEADER * sizeof(int), ctx->stream));
            }
            ctx->normals_tail_list = ctx->normals_tail.as<int>();
            ctx->normals_tail_on_map_stream = other_stream;
        }
    }
    const int tok = prof_begin(ctx, 2);
    if (ctx->knn_lanes == 2)
        launch_normals_all_t<2>(ctx, kn, g);
    else
        launch_normals_all_t<4>(ctx, kn, g);
    prof_end(ctx, tok);
    ctx->normals_tail_list = nullptr;
    ICP_HIP(ctx, hipGetLastError());
    ctx->normals_ready = true;
    ctx->normals_eager_count += ctx->map_m;
    return ICP_OK;
}

int launch_normals_owned(icp_ctx* ctx, int rank, int world, float* by_index_dev) {
    const int kn = ctx->cfg.num_neighbors_normals + 1;
    if (kn != 11 && kn != 6 && kn != 21) {
        ctx->error = "map-sharded normals support num_neighbors_normals = 5, 10 or 20";
        return ICP_ERR_INVALID_ARGUMENT;
    }
    const int64_t m = ctx->map_m;
    ICP_HIP(ctx, hipMemsetAsync(by_index_dev, 0, (size_t)m * sizeof(float4), ctx->stream));
    const GridView g = make_view(ctx);
    const int blocks = (int)((m + NRM_THREADS / 4 - 1) / (NRM_THREADS / 4));
    const int rings = knn_fine_rings(ctx);
    float4* out = (float4*)by_index_dev;
    const int tok = prof_begin(ctx, 2);
    if (g.hood && ctx->hoods >= 2 && (kn == 11 || kn == 6)) {
        const int b2 = (int)((m + NRM2_THREADS / 2 - 1) / (NRM2_THREADS / 2));
        if (kn == 11)
            hipLaunchKernelGGL((k_normals_hood2<11, true>), dim3(b2), dim3(NRM2_THREADS), 0, ctx->stream, g, rings, rank, world,
                               out, (int*)nullptr, (int*)nullptr);
        else
            hipLaunchKernelGGL((k_normals_hood2<6, true>), dim3(b2), dim3(NRM2_THREADS), 0, ctx->stream, g, rings, rank, world,
                               out, (int*)nullptr, (int*)nullptr);
    } else if (g.hood && kn == 11)
        hipLaunchKernelGGL((k_normals_hood<11, true>), dim3(blocks), dim3(NRM_THREADS), 0, ctx->stream, g, rings, rank,
                           world, out, (int*)nullptr);
    else if (g.hood && kn == 6)
        hipLaunchKernelGGL((k_normals_hood<6, true>), dim3(blocks), dim3(NRM_THREADS), 0, ctx->stream, g, rings, rank,
                           world, out, (int*)nullptr);
    else if (kn == 11)
        hipLaunchKernelGGL((k_normals_owned<11, 4>), dim3(blocks), dim3(NRM_THREADS), 0, ctx->stream, g, rings, rank, world, out);
    else if (kn == 6)
        hipLaunchKernelGGL((k_normals_owned<6, 4>), dim3(blocks), dim3(NRM_THREADS), 0, ctx->stream, g, rings, rank, world, out);
    else
        hipLaunchKernelGGL((k_normals_owned<21, 4>), dim3(blocks), dim3(NRM_THREADS), 0, ctx->stream, g, rings, rank, world, out);
    prof_end(ctx, tok);
    ICP_HIP(ctx, hipGetLastError());
    return ICP_OK;
}

int launch_normals_install(icp_ctx* ctx, const float* by_index_dev) {
    const int m = (int)ctx->map_m;
    hipLaunchKernelGGL(k_normals_install, dim3((m + 255) / 256), dim3(256), 0, ctx->stream, ctx->sorted_pts.as<float4>(),
                       m, (const float4*)by_index_dev, ctx->normals.as<float4>(), ctx->nflag.as<int>());
    ICP_HIP(ctx, hipGetLastError());
    ctx->normals_ready = true;
    ctx->normals_eager_count += ctx->map_m;
    return ICP_OK;
}

unsigned long long* pose_box(icp_ctx* ctx) { return ctx->posebox.as<unsigned long long>(); }

unsigned next_box_generation(icp_ctx* ctx) {
    ctx->box_gen += 1u;
    if (ctx->box_gen == 0u) ctx->box_gen = 1u;  // tag 0 = never written
    return ctx->box_gen;
}

// from iteration `narrow_from` on (few NN-cache misses expected) 512 queries per block, one lane each: a wrong guess costs
// time only
static int fused_cache_mode(const icp_ctx* ctx) {
    // (positions share the cache word with the iteration tag: maps of 2^24 points and more search every iteration)
    // entries carry the index of the iteration that searched them, and the pose history is kept by the same index — the
    // count of ALL iterations of the registration, fused or not (the seam API may install normals mid-registration: the
    // unfused iterations before that write no entries).  The cache is used once a fused launch of THIS registration has
    // written every entry: the first one runs without it
    return (ctx->use_nn_cache && ctx->cache_fresh && ctx->map_m < (1 << 24)) ? ctx->use_nn_cache : 0;
}

// (may later launches of this registration use the NN cache at all?)
static bool use_cache_possible(const icp_ctx* ctx) { return ctx->use_nn_cache && ctx->map_m < (1 << 24); }

bool next_fused_launch_is_narrow(const icp_ctx* ctx) {
    // (without the NN cache — the first iteration of a registration — every query searches: one lane each with the ball
    // search, which suits the 512-query shape; the 4-lane groups of the generic path want the 128-query shape)
    const int use_cache = fused_cache_mode(ctx);
    if (ctx->lazy_now) return true;  // (normals on demand exist in the 512-query shape only)
    return (use_cache || ctx->ball_search) && ctx->narrow_from >= 0 && ctx->iter_in_registration >= ctx->narrow_from;
}

// the most workgroups of the resident tail's shape the device holds at once (cached per context; 0: unknown -> no tail)
static int tail_capacity(icp_ctx* ctx) {
    if (ctx->tail_capacity < 0) {
        int per_cu = 0, cus = 0;
        hipDeviceProp_t prop;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, k_iterate_compact<2, IT_THREADS, IT_THREADS, false, true>,
                                                         IT_THREADS, 0) == hipSuccess &&
            hipGetDeviceProperties(&prop, ctx->cfg.device) == hipSuccess)
            cus = prop.multiProcessorCount;
        ctx->tail_capacity = per_cu * cus;
    }
    return ctx->tail_capacity;
}

// may the launch of the NEXT iteration be a resident tail over `tail_iters` iterations?  (a lead launch of the plain
// 512-query shape with rows to solve in front of it, every workgroup resident at once, the NN cache live)
bool fused_tail_possible(icp_ctx* ctx, int prev_rows, int tail_iters) {
    if (ctx->resident_tail <= 0 || ctx->tail_disabled || tail_iters < 2 || prev_rows <= 0 || ctx->lazy_now) return false;
    if (ctx->iter_in_registration < ctx->resident_tail || !fused_cache_mode(ctx) || !next_fused_launch_is_narrow(ctx)) return false;
    if (ctx->iter_in_registration < ctx->wide_until && ctx->ball_search) return false;  // (the 1024-thread shape has no tail)
    const int blocks = (int)((ctx->tgt_n + IT_THREADS - 1) / IT_THREADS);
    return ctx->tgt_n > 0 && blocks <= tail_capacity(ctx) && blocks <= ctx->resident_tail_max_blocks;
}

// before the first launch of a registration of up to `iters` iterations: will its lead launches end in a resident tail?
// (then a live stop threshold needs no chunked launches: the tail ends on the device when the loop does)
bool fused_tail_planned(icp_ctx* ctx, int iters) {
    if (ctx->resident_tail <= 0 || ctx->tail_disabled || !ctx->use_nn_cache || ctx->map_m >= (1 << 24) || ctx->tgt_n <= 0 || ctx->lazy_now) return false;
    const int from = (ctx->ball_search && ctx->wide_until > ctx->resident_tail) ? ctx->wide_until : ctx->resident_tail;
    if (ctx->narrow_from < 0 || ctx->narrow_from > from || iters - from < 2) return false;
    const int blocks = (int)((ctx->tgt_n + IT_THREADS - 1) / IT_THREADS);
    return blocks <= tail_capacity(ctx) && blocks <= ctx->resident_tail_max_blocks;
}

// what launch_iterate_fused has decided about the next fused launch of a context, before anything is launched: the
// arguments, the grid and the instantiation.  prepare_iterate_fused updates the context's bookkeeping (iteration count,
// parity, mailbox generation, cache state) as if the launch had happened: the caller launches — alone, or as one sequence
// of a batched launch.
enum FusedShape { SHAPE_LAZY11, SHAPE_LAZY6, SHAPE_TAIL, SHAPE_WIDE, SHAPE_NARROW, SHAPE_DENSE8, SHAPE_DENSE6, SHAPE_LATE };
struct FusedLaunch {
    IterateDesc d;
    int grid = 0;       // workgroups of the single launch (the lead included)
    FusedShape shape = SHAPE_NARROW;
    bool stats = false;
    int rows = 0, quad = 1;
    int iter = 0;       // (profiling: index of the iteration)
};

static int prepare_iterate_fused(icp_ctx* ctx, bool lead_mode, int prev_rows, int prev_quad, int tail_iters, FusedLaunch& fl) {
    const int n = (int)ctx->tgt_n;
    const int use_cache = fused_cache_mode(ctx);
    const bool narrow = next_fused_launch_is_narrow(ctx);
    const bool tail = tail_iters > 0;  // (the caller has asked fused_tail_possible)
    const int per_block = narrow ? IT_THREADS : IT_QUERIES;
    const int blocks = n > 0 ? (n + per_block - 1) / per_block : 1;
    {
        const size_t half = (size_t)(n > 0 ? (n + IT_QUERIES - 1) / IT_QUERIES : 1) * NEQ * sizeof(double);
        if (half > ctx->partials_half) {  // (only ever at the first launch of a registration: nothing is pending in it)
            ICP_HIP(ctx, ctx->partials.reserve(2 * half));
            ctx->partials_half = ctx->partials.bytes / 2 / (NEQ * sizeof(double)) * (NEQ * sizeof(double));
        }
    }
    ICP_HIP(ctx, ctx->nn_cache.reserve((size_t)(n > 0 ? n : 1) * sizeof(int4)));
    fl.iter = ctx->iter_in_registration;
    IterInputs in;
    in.tgt = ctx->tgt4.as<float4>();
    in.normals = ctx->normals.as<float4>();
    in.nn_cache = ctx->nn_cache.as<int4>();
    in.rec = nullptr;
    if (ctx->hit_records && use_cache_possible(ctx)) {
        ICP_HIP(ctx, ctx->nn_rec.reserve((size_t)(n > 0 ? n : 1) * 2 * sizeof(float4)));
        in.rec = ctx->nn_rec.as<float4>();
    }
    in.pose_hist = ctx->pose_hist;
    // previous frame's neighbours as seeds of the first, cache-less iteration (same scan shape only)
    in.frame_seed = (!ctx->cache_fresh && ctx->frame_seed && ctx->seed_n == n && n > 0)
                        ? ctx->seed_orig.as<int>() : nullptr;
    // the classic launch writes parity 0 (what launch_sum_solve reads by default); lead launches alternate
    const int parity = lead_mode ? ctx->partials_parity : 0;
    in.partials = (double*)(ctx->partials.as<char>() + (size_t)parity * ctx->partials_half);
    LeadArgs lead;
    if (lead_mode) {
        lead.box = pose_box(ctx);
        lead.solve = prev_rows > 0 ? 1 : 0;
        lead.gen = lead.solve ? next_box_generation(ctx) : ctx->box_gen;  // the lead publishes it / already published
        if (tail) {  // generations lead.gen .. lead.gen + tail_iters: one per iteration + the solve behind the last one
            if (lead.gen > 0xffffffffu - (unsigned)tail_iters - 2u) {  // (no wrap inside a tail: tag 0 = never written)
                ctx->box_gen = 0;
                lead.gen = next_box_generation(ctx);
            }
            ctx->box_gen = lead.gen + (unsigned)tail_iters;
        }
        lead.prev_partials = (const double*)(ctx->partials.as<char>() + (size_t)(parity ^ 1) * ctx->partials_half);
        lead.prev_rows = prev_rows;
        lead.prev_quad = prev_quad;
        lead.neq = ctx->neq;
        lead.loss_hist = ctx->loss_hist;
        lead.dx_hist = ctx->dx_hist;
        lead.hist_cap = ctx->hist_cap;
        lead.timeout_ticks = (long long)(ctx->lead_timeout_ms * 1.0e5);  // 100 MHz wall clock (option "lead_timeout_ms")
        ctx->partials_parity = parity ^ 1;
    }
    const int grid = blocks + ((lead_mode && !tail) ? lead.solve : 0);  // (the tail's lead is its workgroup 0)
    in.n = n;
    in.iter = ctx->iter_in_registration;
    in.mode = ctx->tgt_mode;
    in.max_rings = ctx->cfg.max_rings;
    in.use_cache = use_cache;
    // (the 128-query shape runs while most workgroups search: a wave per miss is a round of ~8 us under that load, the
    // 4-lane groups take up to 128 misses in 10-15 us — whole waves only for a handful)
    in.wave_misses = min(narrow ? ctx->wave_misses : ctx->wave_misses_dense, IT_QUERIES);
    in.chunk_stride = narrow ? blocks : 0;  // (S = ceil(base rows / 4) = the number of 512-query workgroups)
    in.normals_rw = ctx->normals.as<float4>();
    in.nflag = ctx->nflag.as<int>();
    in.knn_rings = ctx->knn_rings >= 0 ? ctx->knn_rings : (ctx->cfg.max_rings < 2 ? ctx->cfg.max_rings : 2);
    in.tail_iters = tail ? tail_iters : 0;
    in.refresh_at = ctx->refresh_at;
    in.tail_rows = nullptr;
    if (tail) {
        const size_t need = (size_t)blocks * NEQ * 2 * sizeof(unsigned long long);
        if (ctx->tail_rows.bytes < need) {  // (tag 0 belongs to no generation: a fresh allocation is zeroed)
            ICP_HIP(ctx, ctx->tail_rows.reserve(need));
            ICP_HIP(ctx, hipMemsetAsync(ctx->tail_rows.ptr, 0, ctx->tail_rows.bytes, ctx->stream));
        }
        in.tail_rows = ctx->tail_rows.as<unsigned long long>();
    }
    in.ball = ctx->ball_search;
    in.ball_lanes = ctx->ball_lanes;
    in.ball_empty = ctx->ball_empty;
    in.far_lanes = ctx->far_lanes;
    in.far_max = min(ctx->far_max, IT_THREADS);
    in.far_min = ctx->far_min;
    in.ball_max = ctx->ball_max < BALL_MAX_CAND ? ctx->ball_max : BALL_MAX_CAND;
    in.refresh_margin = (tail || ctx->iter_in_registration == ctx->refresh_at) ? ctx->refresh_margin : 0.f;  // (a tail applies it in iteration `refresh_at` only)
    in.swz_bpr_shift = -1;
    in.swz_sectors = in.swz_band_rows = in.swz_row_blocks = 1;
    if (ctx->xcd_sectors && blocks >= 64 && blocks % 8 == 0) {
        // a range image of cfg.height x cfg.width pixels in row-major order: azimuth sectors (x elevation bands when a
        // row has fewer than 8 blocks); any other target array: eight contiguous runs of workgroups
        const int W = ctx->cfg.width, H = ctx->cfg.height;
        int row_blocks = 1, rows = blocks;
        if (narrow && (int64_t)H * W == n && W % IT_QUERIES == 0 && H % 4 == 0) {
            // (a 512-query workgroup = the same 128 columns of four image rows H / 4 apart: logical workgroup = that segment
            // of the first of them)
            row_blocks = W / IT_QUERIES;
            rows = H / 4;
        } else if (!narrow && (int64_t)H * W == n && W % per_block == 0) {
            row_blocks = W / per_block;
            rows = H;
        }
        const int sectors = row_blocks >= 8 ? 8 : row_blocks;  // 8, or a divisor of 8 when row_blocks is one
        int bpr = row_blocks / (sectors > 0 ? sectors : 1), shift = 0;
        while ((1 << shift) < bpr) ++shift;
        if (sectors > 0 && 8 % sectors == 0 && row_blocks % sectors == 0 && (1 << shift) == bpr &&
            rows % (8 / sectors) == 0) {
            in.swz_bpr_shift = shift;
            in.swz_sectors = sectors;
            in.swz_band_rows = rows / (8 / sectors);
            in.swz_row_blocks = row_blocks;
        }
    }
    // the first launches of a registration (most queries search) with 1024 threads per workgroup: twice the lanes for the
    // same 512 queries, so every miss gets two (the slowest WAVE sets these launches: a lane that walks 200 candidates of a
    // dense cell alone); same super-rows, same bits
    const bool wide = narrow && ctx->iter_in_registration < ctx->wide_until && ctx->ball_search && !ctx->lazy_now;
    const int kn_lazy = ctx->lazy_now ? ctx->cfg.num_neighbors_normals + 1 : 0;
    fl.stats = ctx->search_stats != 0;  // (dev: the instrumented instantiations exist for the tail and the two default shapes only)
    // the late kernel: the 512-query shape once the NN cache and its hit records carry the launch ("late_from")
    const bool late = narrow && !wide && !tail && !kn_lazy && !fl.stats && use_cache && in.rec && ctx->late_from >= 0 &&
                      ctx->iter_in_registration >= ctx->late_from;
    fl.shape = kn_lazy == 11 ? SHAPE_LAZY11  // (normals on demand: the 512-query shape from the first iteration on, built for 256 registers)
               : kn_lazy == 6 ? SHAPE_LAZY6
               : tail         ? SHAPE_TAIL
               : wide         ? SHAPE_WIDE
               : late         ? SHAPE_LATE
               : narrow       ? SHAPE_NARROW
               : ctx->iterate_dense ? SHAPE_DENSE8 : SHAPE_DENSE6;
    fl.d.g = make_view(ctx);
    fl.d.in = in;
    fl.d.st = reg_state(ctx);
    fl.d.ap = make_align_params(ctx);
    fl.d.lead = lead;
    fl.d.blocks = blocks;
    fl.d.pad[0] = fl.d.pad[1] = fl.d.pad[2] = 0;
    fl.grid = grid;
    fl.rows = blocks;
    fl.quad = narrow ? 0 : 1;  // base rows (summed four by four first) or super-rows already
    ctx->iter_in_registration += tail ? tail_iters : 1;
    ctx->cache_fresh = true;
    ctx->cache_n = n;  // nn_cache now describes these targets against the current grid
    ctx->cache_m = ctx->map_m;
    ctx->cache_gen = ctx->grid_gen;
    return ICP_OK;
}

int launch_iterate_fused(icp_ctx* ctx, int* rows_out, int* quad_out, bool lead_mode, int prev_rows, int prev_quad,
                         int tail_iters) {
    FusedLaunch fl;
    const int rc = prepare_iterate_fused(ctx, lead_mode, prev_rows, prev_quad, tail_iters, fl);
    if (rc) return rc;
    const IterateDesc& d = fl.d;
    const dim3 grid(fl.grid);
    const int tok = prof_begin(ctx, 0, fl.iter);
    switch (fl.shape) {
        case SHAPE_LAZY11:
            hipLaunchKernelGGL((k_iterate_compact<2, IT_THREADS, IT_THREADS, false, false, 11>), grid, dim3(IT_THREADS), 0,
                               ctx->stream, d.g, d.in, d.st, d.ap, d.lead);
            break;
        case SHAPE_LAZY6:
            hipLaunchKernelGGL((k_iterate_compact<2, IT_THREADS, IT_THREADS, false, false, 6>), grid, dim3(IT_THREADS), 0,
                               ctx->stream, d.g, d.in, d.st, d.ap, d.lead);
            break;
        case SHAPE_TAIL:
            if (fl.stats)
                hipLaunchKernelGGL((k_iterate_compact<2, IT_THREADS, IT_THREADS, true, true>), grid, dim3(IT_THREADS), 0,
                                   ctx->stream, d.g, d.in, d.st, d.ap, d.lead);
            else
                hipLaunchKernelGGL((k_iterate_compact<2, IT_THREADS, IT_THREADS, false, true>), grid, dim3(IT_THREADS), 0,
                                   ctx->stream, d.g, d.in, d.st, d.ap, d.lead);
            break;
        // (`d.in.rec` set — option "hit_records" — selects the builds with the record test compiled in; with the dev stamps
        // they are not built: the stamped launch of a context with records runs without stamps)
        case SHAPE_WIDE:
            if (d.in.rec)
                hipLaunchKernelGGL((k_iterate_compact<4, 2 * IT_THREADS, IT_THREADS, false, false, 0, true>), grid, dim3(2 * IT_THREADS), 0,
                                   ctx->stream, d.g, d.in, d.st, d.ap, d.lead);
            else if (fl.stats)
                hipLaunchKernelGGL((k_iterate_compact<4, 2 * IT_THREADS, IT_THREADS, true>), grid, dim3(2 * IT_THREADS), 0,
                                   ctx->stream, d.g, d.in, d.st, d.ap, d.lead);
            else
                hipLaunchKernelGGL((k_iterate_compact<4, 2 * IT_THREADS, IT_THREADS>), grid, dim3(2 * IT_THREADS), 0,
                                   ctx->stream, d.g, d.in, d.st, d.ap, d.lead);
            break;
        case SHAPE_NARROW:
            if (d.in.rec)
                hipLaunchKernelGGL((k_iterate_compact<4, IT_THREADS, IT_THREADS, false, false, 0, true>), grid, dim3(IT_THREADS), 0, ctx->stream,
                                   d.g, d.in, d.st, d.ap, d.lead);
            else if (fl.stats)
                hipLaunchKernelGGL((k_iterate_compact<4, IT_THREADS, IT_THREADS, true>), grid, dim3(IT_THREADS), 0, ctx->stream,
                                   d.g, d.in, d.st, d.ap, d.lead);
            else
                hipLaunchKernelGGL((k_iterate_compact<4, IT_THREADS, IT_THREADS>), grid, dim3(IT_THREADS), 0, ctx->stream,
                                   d.g, d.in, d.st, d.ap, d.lead);
            break;
        case SHAPE_LATE:
            if (ctx->late_waves >= 8)
                hipLaunchKernelGGL((k_iterate_late<8, IT_THREADS>), grid, dim3(IT_THREADS), 0, ctx->stream, d.g, d.in, d.st, d.ap, d.lead);
            else
                hipLaunchKernelGGL((k_iterate_late<6, IT_THREADS>), grid, dim3(IT_THREADS), 0, ctx->stream, d.g, d.in, d.st, d.ap, d.lead);
            break;
        // (the 128-query shapes of rounds 2-3, behind their options: built WITH the record test as before — a launch that
        // searches must clear the records of the queries it searches, whichever shape follows it)
        case SHAPE_DENSE8:
            hipLaunchKernelGGL((k_iterate_compact<8, IT_THREADS, IT_QUERIES, false, false, 0, true>), grid, dim3(IT_THREADS), 0, ctx->stream,
                               d.g, d.in, d.st, d.ap, d.lead);
            break;
        case SHAPE_DENSE6:
            hipLaunchKernelGGL((k_iterate_compact<6, IT_THREADS, IT_QUERIES, false, false, 0, true>), grid, dim3(IT_THREADS), 0, ctx->stream,
                               d.g, d.in, d.st, d.ap, d.lead);
            break;
    }
    prof_end(ctx, tok);
    ICP_HIP(ctx, hipGetLastError());
    *rows_out = fl.rows;
    *quad_out = fl.quad;
    return ICP_OK;
}

// ---- the batched launch: one fused iteration of every member's registration (api.hip: icp_batch_*) ---------------------
// `table_host` / `table_dev`: room for `count` descriptors (pinned host memory the caller copies to the device in front of
// the launches of the frame; the launch reads table_dev).  All members must come out with the same instantiation (same
// options, same iteration index): anything else is refused — the caller then falls back to one launch per member.
int prepare_iterate_batch(icp_ctx* const* ctxs, int count, bool lead_mode, const int* prev_rows, const int* prev_quad,
                          void* table_host, BatchedIteration* out) {
    IterateDesc* table = reinterpret_cast<IterateDesc*>(table_host);
    int per_seq = 0;
    for (int b = 0; b < count; ++b) {
        FusedLaunch fl;
        const int rc = prepare_iterate_fused(ctxs[b], lead_mode, prev_rows[b], prev_quad[b], 0, fl);
        if (rc) return rc;
        if (fl.stats || (fl.shape != SHAPE_WIDE && fl.shape != SHAPE_NARROW && fl.shape != SHAPE_LATE) ||
            (b > 0 && (int)fl.shape != out->shape)) {
            ctxs[b]->error = "batched registration: the members must run the same fused shape (narrow / wide; same options)";
            return ICP_ERR_INVALID_ARGUMENT;
        }
        out->shape = (int)fl.shape;
        out->records = fl.d.in.rec != nullptr ? 1 : 0;  // (same options in every member: checked by the caller)
        out->rows[b] = fl.rows;
        out->quad[b] = fl.quad;
        table[b] = fl.d;
        if (fl.d.blocks > per_seq) per_seq = fl.d.blocks;
    }
    out->per_seq = per_seq;
    out->count = count;
    return ICP_OK;
}

int launch_iterate_batch(icp_ctx* first, const BatchedIteration& it, const void* table_dev) {
    const IterateDesc* table = reinterpret_cast<const IterateDesc*>(table_dev);
    const dim3 grid((unsigned)(BATCH_LEAD_SLOTS + it.count * it.per_seq));
    if (it.shape == (int)SHAPE_LATE && first->late_waves >= 8)
        hipLaunchKernelGGL((k_iterate_late_batch<8, IT_THREADS>), grid, dim3(IT_THREADS), 0, first->stream, table, it.count, it.per_seq);
    else if (it.shape == (int)SHAPE_LATE)
        hipLaunchKernelGGL((k_iterate_late_batch<6, IT_THREADS>), grid, dim3(IT_THREADS), 0, first->stream, table, it.count, it.per_seq);
    else if (it.shape == (int)SHAPE_WIDE && it.records)
        hipLaunchKernelGGL((k_iterate_batch<4, 2 * IT_THREADS, IT_THREADS, true>), grid, dim3(2 * IT_THREADS), 0, first->stream, table,
                           it.count, it.per_seq);
    else if (it.shape == (int)SHAPE_WIDE)
        hipLaunchKernelGGL((k_iterate_batch<4, 2 * IT_THREADS, IT_THREADS>), grid, dim3(2 * IT_THREADS), 0, first->stream, table,
                           it.count, it.per_seq);
    else if (it.records)
        hipLaunchKernelGGL((k_iterate_batch<4, IT_THREADS, IT_THREADS, true>), grid, dim3(IT_THREADS), 0, first->stream, table,
                           it.count, it.per_seq);
    else
        hipLaunchKernelGGL((k_iterate_batch<4, IT_THREADS, IT_THREADS>), grid, dim3(IT_THREADS), 0, first->stream, table,
                           it.count, it.per_seq);
    ICP_HIP(first, hipGetLastError());
    return ICP_OK;
}

size_t iterate_desc_bytes() { return sizeof(IterateDesc); }

// Called by the grid build before it overwrites the cell-sorted points: the neighbours the last registration left in
// nn_cache become the seeds of the next frame's first iteration.  `evicted` = oldest map points about to be dropped.
int stash_frame_seeds(icp_ctx* ctx, int64_t evicted, bool indices_survive) {
    ctx->seed_n = 0;
    ctx->seed_job_n = 0;
    if (!indices_survive || !ctx->frame_seed || ctx->cache_n <= 0 || ctx->cache_gen != ctx->grid_gen) return ICP_OK;
    const int n = (int)ctx->cache_n;
    ICP_HIP(ctx, ctx->seed_orig.reserve((size_t)n * sizeof(int)));
    // the conversion rides in the clearing launch of the grid build that follows (build_grid -> run_seed_job)
    ctx->seed_job_n = n;
    ctx->seed_job_m = (int)ctx->cache_m;
    ctx->seed_job_evicted = (int)evicted;
    ctx->seed_n = n;
    return ICP_OK;
}

// the pending conversion as a launch of its own (the build is about to reallocate the cell-sorted points)
int run_seed_job(icp_ctx* ctx) {
    const int n = ctx->seed_job_n;
    ctx->seed_job_n = 0;
    if (n <= 0) return ICP_OK;
    hipLaunchKernelGGL(k_cache_to_seed, dim3((n + 255) / 256), dim3(256), 0, ctx->stream, ctx->nn_cache.as<int4>(),
                       ctx->sorted_pts.as<float4>(), n, ctx->seed_job_m, ctx->seed_job_evicted, ctx->seed_orig.as<int>());
    ICP_HIP(ctx, hipGetLastError());
    return ICP_OK;
}

int launch_normals(icp_ctx* ctx) {
    if (ctx->normals_ready) return ICP_OK;
    const int kn = ctx->cfg.num_neighbors_normals + 1;
    // the worklist length lives on the device: launch a fixed grid and stride over it
    const int64_t cap = ctx->tgt_n < ctx->map_m ? ctx->tgt_n : ctx->map_m;
    const int tok = prof_begin(ctx, 2);
    GridView g = make_view(ctx);
    RegState* st = reg_state(ctx);
    if (kn == 11 || kn == 6 || kn == 21) {
        if (ctx->knn_lanes == 2)
            launch_worklist_t<2>(ctx, kn, g, st, worklist_blocks(cap, 2));
        else
            launch_worklist_t<4>(ctx, kn, g, st, worklist_blocks(cap, 4));
    } else {
        int blocks = (int)((cap + 127) / 128);
        if (blocks < 1) blocks = 1;
        if (blocks > 4096) blocks = 4096;
        hipLaunchKernelGGL(k_normals_generic, dim3(blocks), dim3(128), 0, ctx->stream, g, st,
                           ctx->worklist.as<int>(), kn, ctx->normals.as<float4>(), ctx->nflag.as<int>());
    }
    prof_end(ctx, tok);
    ICP_HIP(ctx, hipGetLastError());
    return ICP_OK;
}

// The map point (original index, -1: none) every target was matched with in fused iteration `iteration` of the last
// registration, read back from the NN cache: the nearest member of the entry's candidate set under the pose that
// iteration ran with (pose_hist) — what a cache hit of that launch used, and what its searches wrote.  Test support
// (icp_last_neighbors): the benchmark-size parity test checks it against brute force.
__global__ void k_last_neighbors(const float4* __restrict__ tgt, const int4* __restrict__ cache,
                                 const float4* __restrict__ pts, const float* __restrict__ pose12, int n, int mode,
                                 int m, int* __restrict__ out) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float4 t4 = tgt[i];
    const int4 c = cache[i];
    int best = -1;
    if (target_valid(t4.x, t4.y, t4.z, mode) && c.x >= 0) {
        float px, py, pz;
        transform_point(pose12, t4.x, t4.y, t4.z, px, py, pz);
        Best b;
        b.d2 = INFINITY;
        b.idx = 0x7fffffff;
        b.pos = -1;
        b.second = INFINITY;
        const int p0 = c.x & CACHE_POS_MASK;
        if (p0 < m) consider(pts[p0], p0, px, py, pz, b);
        if (c.z >= 0 && c.z < m) consider(pts[c.z], c.z, px, py, pz, b);
        if (c.w >= 0 && c.w < m) consider(pts[c.w], c.w, px, py, pz, b);
        if (b.pos >= 0) best = b.idx;
    }
    out[i] = best;
}

int launch_last_neighbors(icp_ctx* ctx, int iteration, int* out_dev) {
    const int n = (int)ctx->tgt_n;
    if (n <= 0) return ICP_OK;
    hipLaunchKernelGGL(k_last_neighbors, dim3((n + 255) / 256), dim3(256), 0, ctx->stream, ctx->tgt4.as<float4>(),
                       ctx->nn_cache.as<int4>(), ctx->sorted_pts.as<float4>(), ctx->pose_hist + (size_t)iteration * 12, n,
                       ctx->tgt_mode, (int)ctx->map_m, out_dev);
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
