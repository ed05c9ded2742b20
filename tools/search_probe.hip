// Developer micro-benchmark: isolates where the time of the search kernel goes (not part of the product).
// build: hipcc -O3 -std=c++17 --offload-arch=gfx950 -Iinclude -Ipylidar-slam_amd/csrc tools/search_probe.hip -o gpurun_out/search_probe
// input: /tmp/probe.bin written by tools/search_probe.py  (int32 M, int32 N, float h, M*3 floats, N*3 floats)
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "search_device.h"

using namespace icp;

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)

__global__ void v0_noop(GridView g, const float4* tgt, int n, int* out) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float4 t = tgt[i];
    out[i] = (int)(t.x + t.y + t.z);
}
__global__ void v1_probe_own(GridView g, const float4* tgt, int n, int* out) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float4 t = tgt[i];
    int s = -1, c = 0;
    grid_lookup(g, cell_coord(t.x, g.inv_h), cell_coord(t.y, g.inv_h), cell_coord(t.z, g.inv_h), s, c);
    out[i] = s + c;
}
__global__ void v2_own_scan(GridView g, const float4* tgt, int n, int* out) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float4 t = tgt[i];
    int s = -1, c = 0;
    Best b; b.d2 = INFINITY; b.idx = 0x7fffffff; b.pos = -1;
    if (grid_lookup(g, cell_coord(t.x, g.inv_h), cell_coord(t.y, g.inv_h), cell_coord(t.z, g.inv_h), s, c))
        scan_cell_1nn(g, s, c, t.x, t.y, t.z, b);
    out[i] = b.pos;
}
__global__ void v3_full(GridView g, const float4* tgt, int n, int* out) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float4 t = tgt[i];
    out[i] = nearest_in_grid(g, t.x, t.y, t.z, 4).pos;
}
// all 27 probes, no candidates
__global__ void v4_probe27(GridView g, const float4* tgt, int n, int* out) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float4 t = tgt[i];
    int cx = cell_coord(t.x, g.inv_h), cy = cell_coord(t.y, g.inv_h), cz = cell_coord(t.z, g.inv_h);
    int acc = 0;
    for (int oz = -1; oz <= 1; ++oz) for (int oy = -1; oy <= 1; ++oy) for (int ox = -1; ox <= 1; ++ox) {
        int s = 0, c = 0;
        grid_lookup(g, cx + ox, cy + oy, cz + oz, s, c);
        acc += s + c;
    }
    out[i] = acc;
}


// v5: own cell first, then the 26 neighbour probes issued together (pruned by box distance), then ONE flattened
// candidate loop per lane over an LDS-resident per-lane cell list (no per-cell serialisation across lanes)
template <int BLOCK, int WIDE>
__global__ __launch_bounds__(BLOCK) void v5_flat(GridView g, const float4* tgt, int n, int* out) {
    __shared__ uint2 stack[26][BLOCK];
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float4 t = tgt[i];
    const float px = t.x, py = t.y, pz = t.z;
    Best b; b.d2 = INFINITY; b.idx = 0x7fffffff; b.pos = -1;
    const int cx = cell_coord(px, g.inv_h), cy = cell_coord(py, g.inv_h), cz = cell_coord(pz, g.inv_h);
    const float h = g.h;
    const float fx = fminf(fmaxf(px - (float)cx * h, 0.f), h);
    const float fy = fminf(fmaxf(py - (float)cy * h, 0.f), h);
    const float fz = fminf(fmaxf(pz - (float)cz * h, 0.f), h);
    const float edge = fminf(fminf(fminf(fx, h - fx), fminf(fy, h - fy)), fminf(fz, h - fz));
    int start, count;
    if (grid_lookup(g, cx, cy, cz, start, count)) scan_cell_1nn(g, start, count, px, py, pz, b);
    // batched first probes
    GridEntry e[26];
    unsigned long long keys[26];
    float gap2[26];
    int c = 0;
#pragma unroll
    for (int oz = -1; oz <= 1; ++oz)
#pragma unroll
        for (int oy = -1; oy <= 1; ++oy)
#pragma unroll
            for (int ox = -1; ox <= 1; ++ox) {
                if (ox == 0 && oy == 0 && oz == 0) continue;
                const float gx = axis_gap(ox, fx, h), gy = axis_gap(oy, fy, h), gz = axis_gap(oz, fz, h);
                gap2[c] = gx * gx + gy * gy + gz * gz;
                keys[c] = pack_cell(cx + ox, cy + oy, cz + oz);
                GridEntry v; v.key = GRID_EMPTY; v.start = 0; v.count = 0;
                if (gap2[c] <= b.d2) v = g.table[hash_cell(keys[c]) & g.mask];
                e[c] = v;
                ++c;
            }
    int nl = 0;
#pragma unroll
    for (int k = 0; k < 26; ++k) {
        GridEntry v = e[k];
        if (v.key != GRID_EMPTY && v.key != keys[k]) {  // collision: continue the linear probe
            unsigned slot = (hash_cell(keys[k]) + 1) & g.mask;
            while (true) {
                v = g.table[slot];
                if (v.key == keys[k] || v.key == GRID_EMPTY) break;
                slot = (slot + 1) & g.mask;
            }
        }
        if (v.key == keys[k] && v.count > 0) {
            stack[nl][threadIdx.x] = make_uint2((unsigned)v.start, ((unsigned)v.count << 8) | (unsigned)k);
            ++nl;
        }
    }
    // flattened scan
    int li = 0, k = 0, cnt = 0, st = 0;
    while (true) {
        while (k >= cnt && li < nl) {
            const uint2 s2 = stack[li][threadIdx.x];
            ++li;
            st = (int)s2.x; cnt = (int)(s2.y >> 8); k = 0;
            const int ci = (int)(s2.y & 0xff);
            // re-prune with the current best
            int cc = ci >= 13 ? ci + 1 : ci;
            const int ox = cc % 3 - 1, oy = (cc / 3) % 3 - 1, oz = cc / 9 - 1;
            const float gx = axis_gap(ox, fx, h), gy = axis_gap(oy, fy, h), gz = axis_gap(oz, fz, h);
            if (gx * gx + gy * gy + gz * gz > b.d2) cnt = 0;
        }
        if (k >= cnt) break;
        const int last = st + cnt - 1;
        const int base = st + k;
#pragma unroll
        for (int u = 0; u < WIDE; ++u) {
            const int idx = min(base + u, last);
            const float4 q = g.pts[idx];
            consider(q, idx, px, py, pz, b);
        }
        k += WIDE;
    }
    const float bound = h + edge;
    if (!(b.d2 <= bound * bound * 0.999999f)) b = nearest_in_grid(g, px, py, pz, 4);
    out[i] = b.pos;
}

// v6: G lanes per query.  Own cell first (candidates strided over the G lanes), group-min, then the 26 neighbour
// cells split over the lanes with box-distance pruning, group-min.
template <int G>
__device__ inline void group_min(Best& b) {
#pragma unroll
    for (int o = 1; o < G; o <<= 1) {
        const float d2 = __shfl_xor(b.d2, o, 64);
        const int idx = __shfl_xor(b.idx, o, 64);
        const int pos = __shfl_xor(b.pos, o, 64);
        if (better(d2, idx, b.d2, b.idx)) { b.d2 = d2; b.idx = idx; b.pos = pos; }
    }
}
template <int G>
__global__ __launch_bounds__(256) void v6_group(GridView g, const float4* tgt, int n, int* out) {
    const int gid = blockIdx.x * blockDim.x + threadIdx.x;
    const int qi = gid / G, sub = gid % G;
    if (qi >= n) return;
    const float4 t = tgt[qi];
    const float px = t.x, py = t.y, pz = t.z;
    Best b; b.d2 = INFINITY; b.idx = 0x7fffffff; b.pos = -1;
    const int cx = cell_coord(px, g.inv_h), cy = cell_coord(py, g.inv_h), cz = cell_coord(pz, g.inv_h);
    const float h = g.h;
    const float fx = fminf(fmaxf(px - (float)cx * h, 0.f), h);
    const float fy = fminf(fmaxf(py - (float)cy * h, 0.f), h);
    const float fz = fminf(fmaxf(pz - (float)cz * h, 0.f), h);
    const float edge = fminf(fminf(fminf(fx, h - fx), fminf(fy, h - fy)), fminf(fz, h - fz));
    int start, count;
    if (grid_lookup(g, cx, cy, cz, start, count)) {
        const int last = start + count - 1;
        for (int k = start + sub; k <= last; k += 2 * G) {
            const int k1 = min(k + G, last);
            const float4 q0 = g.pts[k], q1 = g.pts[k1];
            consider(q0, k, px, py, pz, b);
            consider(q1, k1, px, py, pz, b);
        }
    }
    group_min<G>(b);
    for (int c = sub; c < 26; c += G) {
        const int cc = c + (c >= 13 ? 1 : 0);
        const int ox = cc % 3 - 1, oy = (cc / 3) % 3 - 1, oz = cc / 9 - 1;
        const float gx = axis_gap(ox, fx, h), gy = axis_gap(oy, fy, h), gz = axis_gap(oz, fz, h);
        if (gx * gx + gy * gy + gz * gz > b.d2) continue;
        if (grid_lookup(g, cx + ox, cy + oy, cz + oz, start, count)) scan_cell_1nn(g, start, count, px, py, pz, b);
    }
    group_min<G>(b);
    const float bound = h + edge;
    if (sub == 0) {
        if (!(b.d2 <= bound * bound * 0.999999f)) b = nearest_in_grid(g, px, py, pz, 4);
        out[qi] = b.pos;
    }
}

int main(int argc, char** argv) {
    FILE* f = fopen(argc > 1 ? argv[1] : "/tmp/probe.bin", "rb");
    if (!f) { printf("no input\n"); return 1; }
    int M, N; float h;
    if (fread(&M, 4, 1, f) != 1 || fread(&N, 4, 1, f) != 1 || fread(&h, 4, 1, f) != 1) return 1;
    std::vector<float> model(3 * M), q(3 * N);
    if (fread(model.data(), 4, 3 * M, f) != (size_t)3 * M || fread(q.data(), 4, 3 * N, f) != (size_t)3 * N) return 1;
    fclose(f);
    if (argc > 2) h = atof(argv[2]);
    unsigned T = 1024; while (T < 2u * M) T <<= 1;
    std::vector<GridEntry> table(T);
    for (auto& e : table) { e.key = GRID_EMPTY; e.start = 0; e.count = 0; }
    float inv_h = 1.0f / h;
    auto cc = [&](float v) { float c = floorf(v * inv_h); return (int)c; };
    std::vector<unsigned> slot_of(M);
    for (int i = 0; i < M; ++i) {
        unsigned long long key = pack_cell(cc(model[3*i]), cc(model[3*i+1]), cc(model[3*i+2]));
        unsigned s = hash_cell(key) & (T - 1);
        while (table[s].key != GRID_EMPTY && table[s].key != key) s = (s + 1) & (T - 1);
        table[s].key = key; table[s].count++; slot_of[i] = s;
    }
    int run = 0, occ = 0;
    for (unsigned s = 0; s < T; ++s) { table[s].start = run; run += table[s].count; if (table[s].count) occ++; table[s].count = 0; }
    std::vector<float4> sorted(M);
    for (int i = 0; i < M; ++i) {
        GridEntry& e = table[slot_of[i]];
        sorted[e.start + e.count++] = make_float4(model[3*i], model[3*i+1], model[3*i+2], [&]{ float fv; int iv = i; memcpy(&fv, &iv, 4); return fv; }());
    }
    std::vector<float4> q4(N);
    for (int i = 0; i < N; ++i) q4[i] = make_float4(q[3*i], q[3*i+1], q[3*i+2], 0.f);
    printf("M=%d N=%d h=%.3f table=%u occupied=%d pts/cell=%.1f\n", M, N, h, T, occ, (double)M / occ);
    GridEntry* dt; float4 *dp, *dq; int* dout;
    CK(hipMalloc(&dt, T * sizeof(GridEntry))); CK(hipMalloc(&dp, M * sizeof(float4)));
    CK(hipMalloc(&dq, N * sizeof(float4))); CK(hipMalloc(&dout, N * 4));
    CK(hipMemcpy(dt, table.data(), T * sizeof(GridEntry), hipMemcpyHostToDevice));
    CK(hipMemcpy(dp, sorted.data(), M * sizeof(float4), hipMemcpyHostToDevice));
    CK(hipMemcpy(dq, q4.data(), N * sizeof(float4), hipMemcpyHostToDevice));
    GridView g; g.table = dt; g.mask = T - 1; g.h = h; g.inv_h = inv_h; g.pts = dp; g.m = M; g.row_of_slot = nullptr; g.rows = nullptr; g.row_of_pos = nullptr; g.ctable = nullptr; g.cmask = 0; g.ch = 0; g.cinv_h = 0; g.cpts = nullptr; g.pos_of_orig = nullptr; g.dbg = nullptr;
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    auto timeit = [&](const char* name, auto kern, int block) {
        float best = 1e9, tot = 0; const int reps = getenv("PROBE_REPS") ? atoi(getenv("PROBE_REPS")) : 20;
        for (int r = 0; r < reps + 3; ++r) {
            CK(hipEventRecord(a));
            hipLaunchKernelGGL(kern, dim3((N + block - 1) / block), dim3(block), 0, 0, g, (const float4*)dq, N, dout);
            CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
            float ms; CK(hipEventElapsedTime(&ms, a, b));
            if (r >= 3) { best = std::min(best, ms); tot += ms; }
        }
        printf("%-14s block=%d  avg %.1f us  min %.1f us\n", name, block, tot / reps * 1e3, best * 1e3);
    };
    for (int block : {256}) {
        timeit("v0_noop", v0_noop, block);
        timeit("v1_probe_own", v1_probe_own, block);
        timeit("v2_own_scan", v2_own_scan, block);
        timeit("v4_probe27", v4_probe27, block);
        timeit("v3_full", v3_full, block);
    }
    std::vector<int> ref(N), got(N);
    hipLaunchKernelGGL(v3_full, dim3((N + 255) / 256), dim3(256), 0, 0, g, (const float4*)dq, N, dout);
    CK(hipMemcpy(ref.data(), dout, N * 4, hipMemcpyDeviceToHost));
    timeit("v5_flat<256,4>", v5_flat<256, 4>, 256);
    CK(hipMemcpy(got.data(), dout, N * 4, hipMemcpyDeviceToHost));
    { int bad = 0; for (int i = 0; i < N; ++i) bad += ref[i] != got[i]; printf("   mismatches vs v3_full: %d\n", bad); }
    auto time_group = [&](const char* name, auto kern, int G) {
        float best = 1e9, tot = 0; const int reps = 20;
        for (int r = 0; r < reps + 3; ++r) {
            CK(hipEventRecord(a));
            hipLaunchKernelGGL(kern, dim3((unsigned)(((long long)G * N + 255) / 256)), dim3(256), 0, 0, g, (const float4*)dq, N, dout);
            CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
            float ms; CK(hipEventElapsedTime(&ms, a, b));
            if (r >= 3) { best = std::min(best, ms); tot += ms; }
        }
        CK(hipMemcpy(got.data(), dout, N * 4, hipMemcpyDeviceToHost));
        int bad = 0; for (int i = 0; i < N; ++i) bad += ref[i] != got[i];
        printf("%-14s avg %.1f us  min %.1f us  mismatches %d\n", name, tot / reps * 1e3, best * 1e3, bad);
    };
    time_group("v6_group<2>", v6_group<2>, 2);
    time_group("v6_group<4>", v6_group<4>, 4);
    time_group("v6_group<8>", v6_group<8>, 8);
    time_group("v6_group<16>", v6_group<16>, 16);
    if (!getenv("PROBE_REPS")) {
        timeit("v5_flat<256,8>", v5_flat<256, 8>, 256);
        timeit("v5_flat<128,4>", v5_flat<128, 4>, 128);
        timeit("v5_flat<64,4>", v5_flat<64, 4>, 64);
        timeit("v5_flat<64,2>", v5_flat<64, 2>, 64);
    }
    return 0;
}
