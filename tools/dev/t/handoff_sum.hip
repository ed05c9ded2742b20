// dev: the in-kernel hand-off with the product's own consumer (sum_tagged_rows_vt) and producer (tagged_row_store)
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
#include "solve_device.h"
using namespace icp;
__global__ __launch_bounds__(512) void k_handoff(unsigned long long* rows, unsigned long long* box, int P, int rounds, long long* lat, int* fails, double* sums, int self) {
    __shared__ double lds[32][NEQ];
    __shared__ double total[NEQ];
    __shared__ int failed;
    const int row = self ? (int)blockIdx.x : (int)blockIdx.x - 1;
    for (int k = 1; k <= rounds; ++k) {
        if (blockIdx.x == 0) {
            if (threadIdx.x == 0) failed = 0;
            __syncthreads();
            const long long t0 = wall_clock64();
            if (k > 1) {
                sum_tagged_rows_vt<512>(rows, P, (unsigned)(k - 1), total, lds, wall_clock64() + 2000000ll, &failed);
                __syncthreads();
            }
            if (threadIdx.x == 0) {
                lat[k] = wall_clock64() - t0;
                if (failed) atomicAdd(fails, 1);
                sums[k] = total[0];
                __hip_atomic_store(box, (unsigned long long)k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            __syncthreads();
            if (!self) continue;
        }
        if (threadIdx.x == 0) {
            const long long t0 = wall_clock64();
            while (__hip_atomic_load(box, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < (unsigned long long)k) {
                if (wall_clock64() - t0 > 4000000) break;
                __builtin_amdgcn_s_sleep(1);
            }
        }
        __syncthreads();
        if (threadIdx.x < NEQ) tagged_row_store(rows, row, threadIdx.x, (unsigned)k, 1.0 + row * 0.001 + k);
        __syncthreads();
    }
}
int main() {
    for (int self = 0; self < 2; ++self) for (int P : {12, 64, 128, 256}) {
        const int rounds = 200;
        unsigned long long *rows, *box; long long* lat; int* fails; double* sums;
        hipMalloc(&rows, (size_t)P * NEQ * 16); hipMalloc(&box, 256); hipMalloc(&lat, (rounds + 1) * 8); hipMalloc(&fails, 4); hipMalloc(&sums, (rounds + 1) * 8);
        hipMemset(rows, 0, (size_t)P * NEQ * 16); hipMemset(fails, 0, 4); hipMemset(box, 0, 256);
        hipLaunchKernelGGL(k_handoff, dim3(self ? P : P + 1), dim3(512), 0, 0, rows, box, P, rounds, lat, fails, sums, self);
        hipDeviceSynchronize();
        std::vector<long long> h(rounds + 1); int f = 0; std::vector<double> sm(rounds + 1);
        hipMemcpy(h.data(), lat, (rounds + 1) * 8, hipMemcpyDeviceToHost); hipMemcpy(&f, fails, 4, hipMemcpyDeviceToHost); hipMemcpy(sm.data(), sums, (rounds + 1) * 8, hipMemcpyDeviceToHost);
        double s = 0; long long mx = 0; for (int k = 20; k <= rounds; ++k) { s += h[k]; if (h[k] > mx) mx = h[k]; }
        printf("self %d P %3d: mean %.2f us max %.2f us per sum, rounds that timed out %d, sum[200] %.6f (%s)\n", self, P, s / (rounds - 19) * 0.01, mx * 0.01, f, sm[rounds], hipGetErrorString(hipGetLastError()));
        hipFree(rows); hipFree(box); hipFree(lat); hipFree(fails); hipFree(sums);
    }
    return 0;
}
