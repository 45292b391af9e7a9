// Scratch experiment (GPU box): operand layout of v_mfma_f64_4x4x4_4b_f64 and its rounding against
// v_mfma_f64_16x16x4_f64 and a sequential fma chain.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <cstring>
typedef double double4_t __attribute__((ext_vector_type(4)));

__global__ void k_layout(const double *A, const double *B, double *D) {
    // A, B: 64 values (one per lane); D: 64 outputs
    const int l = threadIdx.x;
    double d = __builtin_amdgcn_mfma_f64_4x4x4f64(A[l], B[l], 0.0, 0, 0, 0);
    D[l] = d;
}
// K-deep product of a 4 x K by K x 16 problem, three ways.  E: 4 x K (row-major), U: K x 16
__global__ void k_compare(const double *E, const double *U, int K, double *out4, double *out16, double *outseq,
                          int ai, int ak, int ab, int bj, int bk, int bb, int di, int dj, int db) {
    const int l = threadIdx.x;
    // decode lane -> (x, y, blk) with x = l & 3, y = (l >> 2) & 3, blk = l >> 4
    const int x = l & 3, y = (l >> 2) & 3, blk = l >> 4;
    const int idx[3] = {x, y, blk};
    const int i_a = idx[ai], k_a = idx[ak];
    const int j_b = idx[bj], k_b = idx[bk], b_b = idx[bb];
    (void)ab;
    double acc = 0.0;
    for (int k0 = 0; k0 < K; k0 += 4) {
        const double a = E[i_a * K + k0 + k_a];
        const double b = U[(k0 + k_b) * 16 + 4 * b_b + j_b];
        acc = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, acc, 0, 0, 0);
    }
    const int i_d = idx[di], j_d = idx[dj], b_d = idx[db];
    out4[i_d * 16 + 4 * b_d + j_d] = acc;
    // 16x16x4: rows 0-3 valid
    const int lr = l & 15, lk = l >> 4;
    double4_t c = {0, 0, 0, 0};
    for (int k0 = 0; k0 < K; k0 += 4) {
        const double a = lr < 4 ? E[lr * K + k0 + lk] : 0.0;
        const double b = U[(k0 + lk) * 16 + lr];
        c = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0);
    }
    // D: row = lk + 4 r, col = lr
    if (lk < 4) out16[lk * 16 + lr] = c[0];
    // sequential fma
    const int si = l >> 4, sj = l & 15;
    double s = 0.0;
    for (int k = 0; k < K; ++k) s = __builtin_fma(E[si * K + k], U[k * 16 + sj], s);
    outseq[si * 16 + sj] = s;
}

int main() {
    double hA[64], hB[64], hD[64];
    double *dA, *dB, *dD;
    hipMalloc(&dA, 512); hipMalloc(&dB, 512); hipMalloc(&dD, 512);
    // layout discovery: A one-hot at lane la, B one-hot at lane lb -> which D lanes light up
    // Encode: A[l] = 1 + l (distinct primes would be better); use products: A[l] = 2^(l%... simpler: brute force per (la, lb)
    int a_i[64], a_k[64], a_b[64], b_j[64], b_k[64], b_b[64];
    for (int l = 0; l < 64; ++l) a_i[l] = a_k[l] = a_b[l] = b_j[l] = b_k[l] = b_b[l] = -1;
    // For every pair (la, lb) record the set of D lanes that are non-zero
    static int hit[64][64][64];
    for (int la = 0; la < 64; ++la)
        for (int lb = 0; lb < 64; ++lb) {
            memset(hA, 0, sizeof hA); memset(hB, 0, sizeof hB);
            hA[la] = 1.0; hB[lb] = 1.0;
            hipMemcpy(dA, hA, 512, hipMemcpyHostToDevice); hipMemcpy(dB, hB, 512, hipMemcpyHostToDevice);
            k_layout<<<1, 64>>>(dA, dB, dD);
            hipMemcpy(hD, dD, 512, hipMemcpyDeviceToHost);
            for (int l = 0; l < 64; ++l) hit[la][lb][l] = hD[l] != 0.0;
        }
    // print, for A lane la: the B lanes it pairs with and the D lanes produced
    for (int la = 0; la < 64; la += 1) {
        if (!(la < 20 || la % 16 == 0)) continue;
        printf("A lane %2d:", la);
        for (int lb = 0; lb < 64; ++lb)
            for (int l = 0; l < 64; ++l)
                if (hit[la][lb][l]) printf(" (B%d->D%d)", lb, l);
        printf("\n");
    }
    // hypothesis test of the bitwise comparison for a few layout hypotheses
    const int K = 64;
    double *hE = (double *)malloc(4 * K * 8), *hU = (double *)malloc(K * 16 * 8);
    srand(7);
    for (int i = 0; i < 4 * K; ++i) hE[i] = (float)((rand() / (double)RAND_MAX - 0.5) * 0.3);
    for (int i = 0; i < K * 16; ++i) hU[i] = (rand() / (double)RAND_MAX - 0.5) * 0.05;
    double *dE, *dU, *d4, *d16, *ds;
    hipMalloc(&dE, 4 * K * 8); hipMalloc(&dU, K * 16 * 8); hipMalloc(&d4, 512); hipMalloc(&d16, 512); hipMalloc(&ds, 512);
    hipMemcpy(dE, hE, 4 * K * 8, hipMemcpyHostToDevice); hipMemcpy(dU, hU, K * 16 * 8, hipMemcpyHostToDevice);
    // index meaning: 0 = l&3, 1 = (l>>2)&3, 2 = l>>4
    const int hyp[][9] = {
        {0, 1, 2, 0, 1, 2, 1, 0, 2},  // A: i=x,k=y; B: j=x,k=y; D: i=y, j=x
        {0, 1, 2, 0, 1, 2, 0, 1, 2},  // D: i=x, j=y
        {1, 0, 2, 1, 0, 2, 1, 0, 2},
        {1, 0, 2, 1, 0, 2, 0, 1, 2},
    };
    for (auto &h : hyp) {
        double o4[64], o16[64], os[64];
        hipMemset(d4, 0, 512);
        k_compare<<<1, 64>>>(dE, dU, K, d4, d16, ds, h[0], h[1], h[2], h[3], h[4], h[5], h[6], h[7], h[8]);
        hipMemcpy(o4, d4, 512, hipMemcpyDeviceToHost); hipMemcpy(o16, d16, 512, hipMemcpyDeviceToHost); hipMemcpy(os, ds, 512, hipMemcpyDeviceToHost);
        int eq4_16 = 0, eq4_s = 0, eq16_s = 0; double maxrel = 0;
        for (int i = 0; i < 64; ++i) {
            eq4_16 += memcmp(&o4[i], &o16[i], 8) == 0; eq4_s += memcmp(&o4[i], &os[i], 8) == 0; eq16_s += memcmp(&o16[i], &os[i], 8) == 0;
            maxrel = fmax(maxrel, fabs(o4[i] - os[i]) / (fabs(os[i]) + 1e-300));
        }
        printf("hyp A(i=%d,k=%d) B(j=%d,k=%d) D(i=%d,j=%d): 4x4x4==16x16x4 %d/64, 4x4x4==seqfma %d/64, 16x16x4==seqfma %d/64, max rel diff 4x4 vs seq %.3g\n",
               h[0], h[1], h[3], h[4], h[6], h[7], eq4_16, eq4_s, eq16_s, maxrel);
    }
    return 0;
}
