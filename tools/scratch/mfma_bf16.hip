// Scratch experiment (GPU box): operand layout of v_mfma_f32_32x32x16_bf16 (gfx950) and the accuracy of a
// float32 product emulated with three bfloat16 pieces per operand (6 MFMA terms).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <cstring>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float float16v __attribute__((ext_vector_type(16)));

__device__ __forceinline__ unsigned short bf16_rn(float x) {  // round to nearest even, finite inputs
    unsigned u = __float_as_uint(x);
    u += 0x7FFFu + ((u >> 16) & 1u);
    return (unsigned short)(u >> 16);
}
__device__ __forceinline__ float bf16_to_f(unsigned short h) { return __uint_as_float((unsigned)h << 16); }

// A: 32 x 16 (row-major), B: 16 x 32 (row-major) as floats that are exactly bf16; D = A B
__global__ void k_layout(const float *A, const float *B, float *D) {
    const int l = threadIdx.x;
    bf16x8 a, b;
    unsigned short ta[8], tb[8];
    for (int e = 0; e < 8; ++e) {
        ta[e] = bf16_rn(A[(l & 31) * 16 + 8 * (l >> 5) + e]);
        tb[e] = bf16_rn(B[(8 * (l >> 5) + e) * 32 + (l & 31)]);
    }
    memcpy(&a, ta, 16);
    memcpy(&b, tb, 16);
    float16v c;
    for (int r = 0; r < 16; ++r) c[r] = 0.0f;
    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
    for (int r = 0; r < 16; ++r) D[((r & 3) + 8 * (r >> 2) + 4 * (l >> 5)) * 32 + (l & 31)] = c[r];
}

// emulated float32 product, K deep: out[i][j] = sum_k A[i][k] B[k][j], 32 x 32 output
__global__ void k_emul(const float *A, const float *B, int K, float *D6, float *D3) {
    const int l = threadIdx.x;
    float16v c6, c3;
    for (int r = 0; r < 16; ++r) c6[r] = c3[r] = 0.0f;
    for (int k0 = 0; k0 < K; k0 += 16) {
        unsigned short pa[3][8], pb[3][8];
        for (int e = 0; e < 8; ++e) {
            float x = A[(l & 31) * K + k0 + 8 * (l >> 5) + e];
            float y = B[(size_t)(k0 + 8 * (l >> 5) + e) * 32 + (l & 31)];
            for (int p = 0; p < 3; ++p) {
                pa[p][e] = bf16_rn(x); x -= bf16_to_f(pa[p][e]);
                pb[p][e] = bf16_rn(y); y -= bf16_to_f(pb[p][e]);
            }
        }
        bf16x8 a[3], b[3];
        for (int p = 0; p < 3; ++p) { memcpy(&a[p], pa[p], 16); memcpy(&b[p], pb[p], 16); }
        // smallest terms first
        c6 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[2], b[0], c6, 0, 0, 0);
        c6 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[1], b[1], c6, 0, 0, 0);
        c6 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], b[2], c6, 0, 0, 0);
        c6 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[1], b[0], c6, 0, 0, 0);
        c6 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], b[1], c6, 0, 0, 0);
        c6 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], b[0], c6, 0, 0, 0);
        c3 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[1], b[0], c3, 0, 0, 0);
        c3 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], b[1], c3, 0, 0, 0);
        c3 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], b[0], c3, 0, 0, 0);
    }
    for (int r = 0; r < 16; ++r) {
        const int o = ((r & 3) + 8 * (r >> 2) + 4 * (l >> 5)) * 32 + (l & 31);
        D6[o] = c6[r]; D3[o] = c3[r];
    }
}

static float host_bf16(float x) { unsigned u; memcpy(&u, &x, 4); u += 0x7FFFu + ((u >> 16) & 1u); u &= 0xFFFF0000u; memcpy(&x, &u, 4); return x; }

int main() {
    srand(3);
    {
        float hA[32 * 16], hB[16 * 32], hD[32 * 32];
        for (float &v : hA) v = host_bf16((rand() / (float)RAND_MAX - 0.5f) * 2);
        for (float &v : hB) v = host_bf16((rand() / (float)RAND_MAX - 0.5f) * 2);
        float *dA, *dB, *dD; hipMalloc(&dA, sizeof hA); hipMalloc(&dB, sizeof hB); hipMalloc(&dD, sizeof hD);
        hipMemcpy(dA, hA, sizeof hA, hipMemcpyHostToDevice); hipMemcpy(dB, hB, sizeof hB, hipMemcpyHostToDevice);
        k_layout<<<1, 64>>>(dA, dB, dD); hipMemcpy(hD, dD, sizeof hD, hipMemcpyDeviceToHost);
        double worst = 0;
        for (int i = 0; i < 32; ++i) for (int j = 0; j < 32; ++j) {
            double s = 0; for (int k = 0; k < 16; ++k) s += (double)hA[i * 16 + k] * hB[k * 32 + j];
            worst = fmax(worst, fabs(s - hD[i * 32 + j]));
        }
        printf("layout check: max |D - ref| = %.3g (expect ~1e-6)\n", worst);
    }
    for (int K : {512, 4096}) {
        float *hA = (float *)malloc(32 * K * 4), *hB = (float *)malloc((size_t)K * 32 * 4), h6[1024], h3[1024];
        for (int i = 0; i < 32 * K; ++i) hA[i] = (rand() / (float)RAND_MAX - 0.5f) * 0.4f;
        for (int i = 0; i < K * 32; ++i) hB[i] = (rand() / (float)RAND_MAX - 0.3f) * 1.7f;
        float *dA, *dB, *d6, *d3; hipMalloc(&dA, 32 * K * 4); hipMalloc(&dB, (size_t)K * 32 * 4); hipMalloc(&d6, 4096); hipMalloc(&d3, 4096);
        hipMemcpy(dA, hA, 32 * K * 4, hipMemcpyHostToDevice); hipMemcpy(dB, hB, (size_t)K * 32 * 4, hipMemcpyHostToDevice);
        k_emul<<<1, 64>>>(dA, dB, K, d6, d3); hipMemcpy(h6, d6, 4096, hipMemcpyDeviceToHost); hipMemcpy(h3, d3, 4096, hipMemcpyDeviceToHost);
        double w6 = 0, w3 = 0, wf = 0, scale = 0;
        for (int i = 0; i < 32; ++i) for (int j = 0; j < 32; ++j) {
            double s = 0, sa = 0; float f = 0;
            for (int k = 0; k < K; ++k) { s += (double)hA[i * K + k] * hB[(size_t)k * 32 + j]; sa += fabs((double)hA[i * K + k] * hB[(size_t)k * 32 + j]); f = fmaf(hA[i * K + k], hB[(size_t)k * 32 + j], f); }
            w6 = fmax(w6, fabs(s - h6[i * 32 + j]) / sa); w3 = fmax(w3, fabs(s - h3[i * 32 + j]) / sa); wf = fmax(wf, fabs(s - f) / sa); scale = sa;
        }
        printf("K=%4d: max error / sum|a b|:  6 terms %.3g   3 terms %.3g   sequential float32 fma %.3g\n", K, w6, w3, wf);
    }
    return 0;
}
