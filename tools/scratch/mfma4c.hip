// Scratch experiment (GPU box): issue rate of v_mfma_f64_4x4x4_4b_f64, 1 / 2 / 4 independent accumulators, one wave per SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
template <int NACC>
__global__ __launch_bounds__(256) void k(double *out, int iters) {
    double acc[NACC];
    for (int i = 0; i < NACC; ++i) acc[i] = 0.0;
    const double a = 1.0 + threadIdx.x * 1e-3, b = 1.0 - threadIdx.x * 1e-3;
    const long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b + i, acc[i], 0, 0, 0);
    }
    const long long t1 = __builtin_readcyclecounter();
    double s = 0;
    for (int i = 0; i < NACC; ++i) s += acc[i];
    if (threadIdx.x == 0) { out[0] = (double)(t1 - t0); out[1] = s; }
}
int main() {
    double *d, h[2];
    hipMalloc(&d, 16);
    const int iters = 100000;
    k<1><<<1, 256>>>(d, iters); hipMemcpy(h, d, 16, hipMemcpyDeviceToHost); printf("1 acc: %.1f cycles per MFMA\n", h[0] / iters / 1);
    k<2><<<1, 256>>>(d, iters); hipMemcpy(h, d, 16, hipMemcpyDeviceToHost); printf("2 acc: %.1f cycles per MFMA\n", h[0] / iters / 2);
    k<4><<<1, 256>>>(d, iters); hipMemcpy(h, d, 16, hipMemcpyDeviceToHost); printf("4 acc: %.1f cycles per MFMA\n", h[0] / iters / 4);
    k<8><<<1, 256>>>(d, iters); hipMemcpy(h, d, 16, hipMemcpyDeviceToHost); printf("8 acc: %.1f cycles per MFMA\n", h[0] / iters / 8);
    return 0;
}
