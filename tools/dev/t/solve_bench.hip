// dev: how long the 6x6 Gauss-Newton step + pose update of the lead takes on ONE wave, stage by stage
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
#include <random>
#include "solve_device.h"
using namespace icp;
__global__ __launch_bounds__(64) void k_bench(const double* neq_in, int reps, long long* ticks, float* out, int stage) {
    __shared__ double neq[NEQ];
    if (threadIdx.x < NEQ) neq[threadIdx.x] = neq_in[threadIdx.x];
    __syncthreads();
    AlignParams ap; ap.scheme = 0; ap.sigma = 0.5f; ap.threshold_delta_pose = 0.f; ap.max_iters = 1 << 30; ap.pose_hist = nullptr;
    float pose[16], params[6];
    for (int k = 0; k < 16; ++k) pose[k] = (k % 5 == 0) ? 1.f : 0.f;
    for (int k = 0; k < 6; ++k) params[k] = 0.f;
    const long long t0 = wall_clock64();
    float acc = 0.f;
    for (int r = 0; r < reps; ++r) {
        if (stage == 0) {  // the whole step
            SolveOut o;
            solve_core(neq, ap, r, pose, params, o);
            for (int k = 0; k < 16; ++k) pose[k] = o.pose[k];
            for (int k = 0; k < 6; ++k) params[k] = o.params[k];
            if (threadIdx.x == 0) neq[21] += 1e-9 * o.dx[0];  // (a dependence from step to step)
            acc += o.dx[1];
        } else if (stage == 1) {  // Cholesky only (one lane's chain, every lane redundantly)
            float dx[6]; double loss; int stopped;
            gauss_newton_from_neq(neq, dx, &loss, &stopped);
            if (threadIdx.x == 0) neq[21] += 1e-9 * dx[0];
            acc += dx[1];
        } else {  // pose algebra only
            float dx[6] = {1e-3f + acc * 1e-9f, 2e-3f, -1e-3f, 1e-4f, -2e-4f, 3e-4f};
            float D[16], P[16];
            wave_build_pose_f32(dx, D);
            for (int rr = 0; rr < 4; ++rr) for (int c = 0; c < 4; ++c) { float s = 0.f; for (int k2 = 0; k2 < 4; ++k2) s += D[4 * rr + k2] * pose[4 * k2 + c]; P[4 * rr + c] = s; }
            wave_from_pose_f32(P, params);
            wave_build_pose_f32(params, pose);
            acc += pose[3];
        }
        __builtin_amdgcn_s_waitcnt(0);
    }
    const long long t1 = wall_clock64();
    if (threadIdx.x == 0) { ticks[0] = t1 - t0; out[0] = acc + pose[3]; }
}
int main() {
    std::mt19937 rng(3); std::normal_distribution<double> nd;
    double J[200][6], r[200]; for (auto& row : J) for (double& v : row) v = nd(rng); for (double& v : r) v = 0.01 * nd(rng);
    std::vector<double> neq(NEQ, 0.0); int k = 0;
    for (int a = 0; a < 6; ++a) for (int b = a; b < 6; ++b) { double s = 0; for (int i = 0; i < 200; ++i) s += J[i][a] * J[i][b]; neq[k++] = s; }
    for (int a = 0; a < 6; ++a) { double s = 0; for (int i = 0; i < 200; ++i) s += J[i][a] * r[i]; neq[21 + a] = s; }
    neq[27] = 1.0; neq[28] = 1.0; neq[29] = 200;
    double* d; long long* t; float* o; hipMalloc(&d, NEQ * 8); hipMalloc(&t, 8); hipMalloc(&o, 4);
    hipMemcpy(d, neq.data(), NEQ * 8, hipMemcpyHostToDevice);
    for (int stage = 0; stage < 3; ++stage) {
        const int reps = 2000;
        hipLaunchKernelGGL(k_bench, dim3(1), dim3(64), 0, 0, d, reps, t, o, stage);
        hipLaunchKernelGGL(k_bench, dim3(1), dim3(64), 0, 0, d, reps, t, o, stage);
        hipDeviceSynchronize();
        long long ticks; float ov; hipMemcpy(&ticks, t, 8, hipMemcpyDeviceToHost); hipMemcpy(&ov, o, 4, hipMemcpyDeviceToHost);
        printf("stage %d (%s): %.3f us per step (%g) %s\n", stage, stage == 0 ? "whole step" : (stage == 1 ? "H assembly + Cholesky + dx" : (stage == 2 ? "pose algebra" : "row-parallel elimination + dx")), ticks * 0.01 / reps, ov, hipGetErrorString(hipGetLastError()));
    }
    return 0;
}
