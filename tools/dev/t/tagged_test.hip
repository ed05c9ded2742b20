// dev: sum_tagged_rows_vt against sum_partials_vt for several row counts
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
#include "solve_device.h"
using namespace icp;
__global__ __launch_bounds__(512) void k_store(unsigned long long* rows, const double* plain, int ns, unsigned tag) {
    const int r = blockIdx.x;
    if (threadIdx.x < NEQ) tagged_row_store(rows, r, threadIdx.x, tag, plain[(size_t)r * NEQ + threadIdx.x]);
}
__global__ __launch_bounds__(512) void k_sum(const unsigned long long* rows, const double* plain, int ns, unsigned tag, double* out_t, double* out_p, int* failed_out) {
    __shared__ double lds[32][NEQ];
    __shared__ double total[NEQ];
    __shared__ int failed;
    if (threadIdx.x == 0) failed = 0;
    __syncthreads();
    sum_tagged_rows_vt<512>(rows, ns, tag, total, lds, wall_clock64() + 5000000ll, &failed);
    __syncthreads();
    if (threadIdx.x < NEQ) out_t[threadIdx.x] = total[threadIdx.x];
    if (threadIdx.x == 0) *failed_out = failed;
    __syncthreads();
    sum_partials_vt<512>(plain, ns, 0, total, lds);
    __syncthreads();
    if (threadIdx.x < NEQ) out_p[threadIdx.x] = total[threadIdx.x];
}
int main() {
    for (int ns : {64, 128, 256, 12, 250, 391}) {
        std::vector<double> h((size_t)ns * NEQ);
        for (size_t i = 0; i < h.size(); ++i) h[i] = 1.0 + 0.001 * (double)(i % 977) - 1e-7 * (double)i;
        double *plain, *ot, *op; unsigned long long* rows; int* f;
        hipMalloc(&plain, h.size() * 8); hipMalloc(&rows, h.size() * 16); hipMalloc(&ot, NEQ * 8); hipMalloc(&op, NEQ * 8); hipMalloc(&f, 4);
        hipMemcpy(plain, h.data(), h.size() * 8, hipMemcpyHostToDevice);
        hipMemset(rows, 0, h.size() * 16);
        hipLaunchKernelGGL(k_store, dim3(ns), dim3(512), 0, 0, rows, plain, ns, 7u);
        hipLaunchKernelGGL(k_sum, dim3(1), dim3(512), 0, 0, rows, plain, ns, 7u, ot, op, f);
        double a[NEQ], b[NEQ]; int failed;
        hipMemcpy(a, ot, NEQ * 8, hipMemcpyDeviceToHost); hipMemcpy(b, op, NEQ * 8, hipMemcpyDeviceToHost); hipMemcpy(&failed, f, 4, hipMemcpyDeviceToHost);
        int same = 1; for (int k = 0; k < NEQ; ++k) same &= (a[k] == b[k]);
        printf("ns %d: failed %d, bit-equal %d (%.12g vs %.12g) err %s\n", ns, failed, same, a[0], b[0], hipGetErrorString(hipGetLastError()));
        hipFree(plain); hipFree(rows); hipFree(ot); hipFree(op); hipFree(f);
    }
    return 0;
}
