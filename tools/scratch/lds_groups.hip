// Which lanes does the LDS serve together?  For each access width, time a wave issuing reads (or writes)
// whose per-lane addresses are conflict-free under ONE hypothesis about the grouping and collide under the others.
//   hipcc --offload-arch=gfx950 -O2 tools/scratch/lds_groups.hip -o tools/scratch/lds_groups && ./lds_groups
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

typedef float f4 __attribute__((ext_vector_type(4)));
typedef float f2 __attribute__((ext_vector_type(2)));

template <int WIDTH, bool WRITE>
__global__ void k_time(const int *__restrict__ addr, int iters, long long *out) {
    extern __shared__ char lds[];
    const int lane = threadIdx.x & 63;
    const int a = addr[lane];
    for (int i = threadIdx.x; i < 40960; i += blockDim.x) reinterpret_cast<float *>(lds)[i] = (float)i;
    __syncthreads();
    float acc = 0.0f;
    const long long t0 = clock64();
    for (int i = 0; i < iters; ++i) {
        if (WRITE) {
            if (WIDTH == 16) {
                asm volatile("ds_write_b128 %0, %1\n ds_write_b128 %0, %1\n ds_write_b128 %0, %1\n ds_write_b128 %0, %1\n ds_write_b128 %0, %1\n ds_write_b128 %0, %1\n ds_write_b128 %0, %1\n ds_write_b128 %0, %1\n ds_write_b128 %0, %1\n ds_write_b128 %0, %1\n ds_write_b128 %0, %1\n ds_write_b128 %0, %1\n ds_write_b128 %0, %1\n ds_write_b128 %0, %1\n ds_write_b128 %0, %1\n ds_write_b128 %0, %1\n s_waitcnt lgkmcnt(0)" ::"v"(a),
                             "v"((f4){1.0f, 2.0f, 3.0f, 4.0f})
                             : "memory");
            } else {
                asm volatile("ds_write_b64 %0, %1\n ds_write_b64 %0, %1\n ds_write_b64 %0, %1\n ds_write_b64 %0, %1\n ds_write_b64 %0, %1\n ds_write_b64 %0, %1\n ds_write_b64 %0, %1\n ds_write_b64 %0, %1\n ds_write_b64 %0, %1\n ds_write_b64 %0, %1\n ds_write_b64 %0, %1\n ds_write_b64 %0, %1\n ds_write_b64 %0, %1\n ds_write_b64 %0, %1\n ds_write_b64 %0, %1\n ds_write_b64 %0, %1\n s_waitcnt lgkmcnt(0)" ::"v"(a),
                             "v"((f2){1.0f, 2.0f})
                             : "memory");
            }
        } else if (WIDTH == 16) {
            f4 v0, v1, v2, v3;
            asm volatile("ds_read_b128 %0, %4\n ds_read_b128 %1, %4\n ds_read_b128 %2, %4\n ds_read_b128 %3, %4\n ds_read_b128 %0, %4\n ds_read_b128 %1, %4\n ds_read_b128 %2, %4\n ds_read_b128 %3, %4\n ds_read_b128 %0, %4\n ds_read_b128 %1, %4\n ds_read_b128 %2, %4\n ds_read_b128 %3, %4\n ds_read_b128 %0, %4\n ds_read_b128 %1, %4\n ds_read_b128 %2, %4\n ds_read_b128 %3, %4\n s_waitcnt lgkmcnt(0)"
                         : "=&v"(v0), "=&v"(v1), "=&v"(v2), "=&v"(v3)
                         : "v"(a)
                         : "memory");
            acc += v0.x + v1.y + v2.z + v3.w;
        } else if (WIDTH == 8) {
            f2 v0, v1, v2, v3;
            asm volatile("ds_read_b64 %0, %4\n ds_read_b64 %1, %4\n ds_read_b64 %2, %4\n ds_read_b64 %3, %4\n ds_read_b64 %0, %4\n ds_read_b64 %1, %4\n ds_read_b64 %2, %4\n ds_read_b64 %3, %4\n ds_read_b64 %0, %4\n ds_read_b64 %1, %4\n ds_read_b64 %2, %4\n ds_read_b64 %3, %4\n ds_read_b64 %0, %4\n ds_read_b64 %1, %4\n ds_read_b64 %2, %4\n ds_read_b64 %3, %4\n s_waitcnt lgkmcnt(0)"
                         : "=&v"(v0), "=&v"(v1), "=&v"(v2), "=&v"(v3)
                         : "v"(a)
                         : "memory");
            acc += v0.x + v1.y + v2.x + v3.y;
        } else {
            float v0, v1, v2, v3;
            asm volatile("ds_read_b32 %0, %4\n ds_read_b32 %1, %4\n ds_read_b32 %2, %4\n ds_read_b32 %3, %4\n ds_read_b32 %0, %4\n ds_read_b32 %1, %4\n ds_read_b32 %2, %4\n ds_read_b32 %3, %4\n ds_read_b32 %0, %4\n ds_read_b32 %1, %4\n ds_read_b32 %2, %4\n ds_read_b32 %3, %4\n ds_read_b32 %0, %4\n ds_read_b32 %1, %4\n ds_read_b32 %2, %4\n ds_read_b32 %3, %4\n s_waitcnt lgkmcnt(0)"
                         : "=&v"(v0), "=&v"(v1), "=&v"(v2), "=&v"(v3)
                         : "v"(a)
                         : "memory");
            acc += v0 + v1 + v2 + v3;
        }
    }
    const long long t1 = clock64();
    if (threadIdx.x == 0) out[0] = t1 - t0;
    if (acc == 123.456f) out[1] = 1;
}

template <int WIDTH, bool WRITE>
static double run(const std::vector<int> &addr, int *d_addr, long long *d_out) {
    hipMemcpy(d_addr, addr.data(), 64 * sizeof(int), hipMemcpyHostToDevice);
    const int iters = 2000;
    hipFuncSetAttribute(reinterpret_cast<const void *>(k_time<WIDTH, WRITE>), hipFuncAttributeMaxDynamicSharedMemorySize, 163840);
    k_time<WIDTH, WRITE><<<1, 256, 163840>>>(d_addr, iters, d_out);
    k_time<WIDTH, WRITE><<<1, 256, 163840>>>(d_addr, iters, d_out);
    hipDeviceSynchronize();
    long long t = 0;
    hipMemcpy(&t, d_out, sizeof(t), hipMemcpyDeviceToHost);
    return (double)t / iters / 16.0 / 4.0;  // cycles per instruction and wave (4 waves issue together)
}

// slots: how many lanes one pass can serve without conflict (256-byte bank row / width)
template <int WIDTH, bool WRITE>
static void sweep(const char *name, int *d_addr, long long *d_out) {
    const int slots = 256 / WIDTH;  // 16 for b128, 32 for b64, 64 for b32
    std::vector<int> addr(64);
    for (int l = 0; l < 64; ++l) addr[l] = l * WIDTH;
    printf("%s linear                         : %6.2f ticks / instruction\n", name, run<WIDTH, WRITE>(addr, d_addr, d_out));
    for (int l = 0; l < 64; ++l) addr[l] = (l % 4) * WIDTH + (l / 4) * 1024;  // 16-way style collisions
    printf("%s 4 slots only (heavy conflicts) : %6.2f\n", name, run<WIDTH, WRITE>(addr, d_addr, d_out));
    if (slots >= 64) return;
    // hypothesis: a pass serves `slots` lanes = the lanes that agree on a set of lane-index bits.
    // bank slot of lane l = its index among the lanes of its pass; rows differ per pass (same banks, other address).
    const int nbits = 6, sbits = slots == 16 ? 4 : 5;
    for (int mask = 0; mask < 64; ++mask) {
        if (__builtin_popcount(mask) != sbits) continue;  // lane bits that select the slot inside a pass
        for (int l = 0; l < 64; ++l) {
            int slot = 0, pass = 0, sb = 0, pb = 0;
            for (int b = 0; b < nbits; ++b) {
                if (mask >> b & 1) slot |= ((l >> b) & 1) << sb++;
                else pass |= ((l >> b) & 1) << pb++;
            }
            addr[l] = slot * WIDTH + pass * 4096;
        }
        const double t = run<WIDTH, WRITE>(addr, d_addr, d_out);
        printf("%s slot bits %c%c%c%c%c%c (lane bit 5..0)   : %6.2f\n", name, mask & 32 ? '1' : '0', mask & 16 ? '1' : '0', mask & 8 ? '1' : '0',
               mask & 4 ? '1' : '0', mask & 2 ? '1' : '0', mask & 1 ? '1' : '0', t);
    }
}

int main() {
    int *d_addr;
    long long *d_out;
    hipMalloc(&d_addr, 64 * sizeof(int));
    hipMalloc(&d_out, 2 * sizeof(long long));
    sweep<16, false>("ds_read_b128 ", d_addr, d_out);
    sweep<8, false>("ds_read_b64  ", d_addr, d_out);
    sweep<16, true>("ds_write_b128", d_addr, d_out);
    sweep<8, true>("ds_write_b64 ", d_addr, d_out);
    return 0;
}
