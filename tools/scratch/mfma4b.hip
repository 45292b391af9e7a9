// Scratch experiment (GPU box): v_mfma_f64_4x4x4_4b_f64 with the measured layout
//   A[i][k] of block q: lane 16 k + 4 q + i;  B[k][j]: lane 16 k + 4 q + j;  D[i][j]: lane 16 i + 4 q + j
// against v_mfma_f64_16x16x4_f64 and a sequential fma chain, bit for bit.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <cstring>
typedef double double4_t __attribute__((ext_vector_type(4)));
__global__ void k_compare(const double *E, const double *U, int K, double *out4, double *out16, double *outseq) {
    const int l = threadIdx.x;
    double acc = 0.0;
    for (int k0 = 0; k0 < K; k0 += 4) {
        const double a = E[(l & 3) * K + k0 + (l >> 4)];
        const double b = U[(k0 + (l >> 4)) * 16 + (l & 15)];
        acc = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, acc, 0, 0, 0);
    }
    out4[(l >> 4) * 16 + (l & 15)] = acc;
    const int lr = l & 15, lk = l >> 4;
    double4_t c = {0, 0, 0, 0};
    for (int k0 = 0; k0 < K; k0 += 4) {
        const double a = lr < 4 ? E[lr * K + k0 + lk] : 0.0;
        const double b = U[(k0 + lk) * 16 + lr];
        c = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0);
    }
    if (lk < 4) out16[lk * 16 + lr] = c[0];
    const int si = l >> 4, sj = l & 15;
    double s = 0.0;
    for (int k = 0; k < K; ++k) s = __builtin_fma(E[si * K + k], U[k * 16 + sj], s);
    outseq[si * 16 + sj] = s;
}
int main() {
    for (int K : {4, 32, 64, 512}) {
        double *hE = (double *)malloc(4 * K * 8), *hU = (double *)malloc(K * 16 * 8);
        srand(7 + K);
        for (int i = 0; i < 4 * K; ++i) hE[i] = (float)((rand() / (double)RAND_MAX - 0.5) * 0.3);
        for (int i = 0; i < K * 16; ++i) hU[i] = (rand() / (double)RAND_MAX - 0.5) * 0.05;
        double *dE, *dU, *d4, *d16, *ds;
        hipMalloc(&dE, 4 * K * 8); hipMalloc(&dU, K * 16 * 8); hipMalloc(&d4, 512); hipMalloc(&d16, 512); hipMalloc(&ds, 512);
        hipMemcpy(dE, hE, 4 * K * 8, hipMemcpyHostToDevice); hipMemcpy(dU, hU, K * 16 * 8, hipMemcpyHostToDevice);
        double o4[64], o16[64], os[64];
        k_compare<<<1, 64>>>(dE, dU, K, d4, d16, ds);
        hipMemcpy(o4, d4, 512, hipMemcpyDeviceToHost); hipMemcpy(o16, d16, 512, hipMemcpyDeviceToHost); hipMemcpy(os, ds, 512, hipMemcpyDeviceToHost);
        int eq4_16 = 0, eq4_s = 0, eq16_s = 0; double maxrel = 0;
        for (int i = 0; i < 64; ++i) {
            eq4_16 += memcmp(&o4[i], &o16[i], 8) == 0; eq4_s += memcmp(&o4[i], &os[i], 8) == 0; eq16_s += memcmp(&o16[i], &os[i], 8) == 0;
            maxrel = fmax(maxrel, fabs(o4[i] - os[i]) / (fabs(os[i]) + 1e-300));
        }
        printf("K=%3d: 4x4x4==16x16x4 %d/64, 4x4x4==seqfma %d/64, 16x16x4==seqfma %d/64, max rel diff 4x4 vs seq %.3g\n", K, eq4_16, eq4_s, eq16_s, maxrel);
    }
    return 0;
}
