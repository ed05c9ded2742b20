// dev: latency and reliability of an in-kernel hand-off of tagged 16-byte elements between workgroups on different XCDs
// (producer blocks 1..P store one 512-byte row each per round, tagged with the round; block 0 polls all rows, then publishes
// the next round through a one-line mailbox the producers poll).  Variants: how the consumer loads (8-byte atomic pairs /
// 16-byte buffer sc1 / sc0 sc1), how the producers store, and the memory type of the rows (hipMalloc / uncached).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
static constexpr int NEQ = 32;
template <int LOADM>
__device__ inline u32x4 load_pair(const unsigned long long* rows, __amdgpu_buffer_rsrc_t r, int row, int col) {
    if (LOADM == 0) {
        const unsigned long long* p = rows + ((size_t)row * NEQ + col) * 2;
        const unsigned long long lo = __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const unsigned long long hi = __hip_atomic_load(p + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        u32x4 v; v.x = (unsigned)lo; v.y = (unsigned)(lo >> 32); v.z = (unsigned)hi; v.w = (unsigned)(hi >> 32); return v;
    } else if (LOADM == 1) {
        return __builtin_amdgcn_raw_buffer_load_b128(r, (row * NEQ + col) * 16, 0, (int)0x80000010u);
    } else {
        return __builtin_amdgcn_raw_buffer_load_b128(r, (row * NEQ + col) * 16, 0, (int)0x80000011u);
    }
}
template <int LOADM, int STOREM>
__global__ __launch_bounds__(512) void k_handoff(unsigned long long* rows, unsigned long long* box, int P, int rounds, long long* lat, int* fails) {
    __shared__ int bad;
    const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(rows, 0, P * NEQ * 16, 0x00020000);
    if (blockIdx.x == 0) {
        for (int k = 1; k <= rounds; ++k) {
            if (threadIdx.x == 0) bad = 0;
            __syncthreads();
            const long long t0 = wall_clock64();
            // every thread waits for its share of the P x 32 elements tagged k
            for (int e = threadIdx.x; e < P * NEQ; e += 512) {
                u32x4 v = load_pair<LOADM>(rows, r, e / NEQ, e % NEQ);
                while (v.y != (unsigned)k || v.w != (unsigned)k) {
                    if (wall_clock64() - t0 > 2000000) { bad = 1; break; }
                    __builtin_amdgcn_s_sleep(1);
                    v = load_pair<LOADM>(rows, r, e / NEQ, e % NEQ);
                }
            }
            __syncthreads();
            if (threadIdx.x == 0) {
                lat[k] = wall_clock64() - t0;
                if (bad) atomicAdd(fails, 1);
                __hip_atomic_store(box, (unsigned long long)(k + 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            __syncthreads();
        }
        return;
    }
    const int row = blockIdx.x - 1;
    for (int k = 1; k <= rounds; ++k) {
        if (threadIdx.x == 0) {
            const long long t0 = wall_clock64();
            while (__hip_atomic_load(box, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < (unsigned long long)k) {
                if (wall_clock64() - t0 > 4000000) break;
                __builtin_amdgcn_s_sleep(1);
            }
        }
        __syncthreads();
        if (threadIdx.x < NEQ) {
            unsigned long long* p = rows + ((size_t)row * NEQ + threadIdx.x) * 2;
            const unsigned long long t = (unsigned long long)k << 32;
            if (STOREM == 0) {
                __hip_atomic_store(p, t | 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __hip_atomic_store(p + 1, t | 2u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            } else {
                u32x4 v; v.x = 1u; v.y = (unsigned)k; v.z = 2u; v.w = (unsigned)k;
                __builtin_amdgcn_raw_buffer_store_b128(v, r, (row * NEQ + (int)threadIdx.x) * 16, 0, (int)0x80000010u);
            }
        }
    }
}
template <int LOADM, int STOREM>
void run(const char* name, int P, bool uncached) {
    const int rounds = 200;
    unsigned long long *rows, *box; long long* lat; int* fails;
    if (uncached) hipExtMallocWithFlags((void**)&rows, (size_t)P * NEQ * 16, hipDeviceMallocUncached);
    else hipMalloc(&rows, (size_t)P * NEQ * 16);
    hipMalloc(&box, 256); hipMalloc(&lat, (rounds + 1) * 8); hipMalloc(&fails, 4);
    hipMemset(rows, 0, (size_t)P * NEQ * 16); hipMemset(fails, 0, 4);
    unsigned long long one = 1; hipMemcpy(box, &one, 8, hipMemcpyHostToDevice);
    hipLaunchKernelGGL((k_handoff<LOADM, STOREM>), dim3(P + 1), dim3(512), 0, 0, rows, box, P, rounds, lat, fails);
    hipDeviceSynchronize();
    std::vector<long long> h(rounds + 1); int f = 0;
    hipMemcpy(h.data(), lat, (rounds + 1) * 8, hipMemcpyDeviceToHost); hipMemcpy(&f, fails, 4, hipMemcpyDeviceToHost);
    double s = 0; long long mx = 0; for (int k = 20; k <= rounds; ++k) { s += h[k]; if (h[k] > mx) mx = h[k]; }
    printf("%-34s P %3d %-8s: mean %.2f us max %.2f us per round, rounds that timed out %d (%s)\n", name, P, uncached ? "uncached" : "hipMalloc",
           s / (rounds - 19) * 0.01, mx * 0.01, f, hipGetErrorString(hipGetLastError()));
    hipFree(rows); hipFree(box); hipFree(lat); hipFree(fails);
}
int main() {
    for (int P : {63, 255}) for (int unc = 0; unc < 2; ++unc) {
        run<0, 0>("load 2x8B atomic, store 2x8B atomic", P, unc);
        run<1, 0>("load 16B buffer sc1, store 2x8B", P, unc);
        run<2, 0>("load 16B buffer sc0 sc1, store 2x8B", P, unc);
        run<1, 1>("load 16B buffer sc1, store 16B sc1", P, unc);
        run<0, 1>("load 2x8B atomic, store 16B sc1", P, unc);
    }
    return 0;
}
