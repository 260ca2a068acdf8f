// tools/mfma_issue.hip -- how many filler instructions per f32 MFMA can a wave carry before the matrix pipe starves?
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));

// MODE 0: FILL independent VALU per MFMA.  MODE 1: operand A produced by a chain of FILL dependent VALU.
// MODE 2: FILL SALU per MFMA.  MODE 3: FILL independent 64-bit integer mads (address math) per MFMA.
template <int NACC, int FILL, int MODE>
__global__ __launch_bounds__(256) void k(float *out, int iters, float seed) {
    f32x16 acc[NACC];
    for (int i = 0; i < NACC; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    float a = seed + threadIdx.x * 1e-3f, b = seed * 0.5f + threadIdx.x * 2e-3f;
    float f0 = a, f1 = b; int s0 = iters; long long m0 = threadIdx.x;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 8; ++u)
#pragma unroll
            for (int i = 0; i < NACC; ++i) {
                float aa = a;
                if (MODE == 0) {
#pragma unroll
                    for (int q = 0; q < FILL; ++q) asm volatile("v_add_f32 %0, %0, %1" : "+v"(f0) : "v"(f1));
                } else if (MODE == 1) {
#pragma unroll
                    for (int q = 0; q < FILL; ++q) asm volatile("v_add_f32 %0, %0, %1" : "+v"(aa) : "v"(f1));
                } else if (MODE == 2) {
#pragma unroll
                    for (int q = 0; q < FILL; ++q) asm volatile("s_add_i32 %0, %0, 1" : "+s"(s0));
                } else if (MODE == 3) {
#pragma unroll
                    for (int q = 0; q < FILL; ++q) asm volatile("v_mad_i64_i32 %0, vcc, %1, %2, %0" : "+v"(m0) : "v"(it), "v"(u) : "vcc");
                }
                acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(aa, b, acc[i], 0, 0, 0);
            }
    }
    float s = f0 + (float)s0 + (float)m0;
    for (int i = 0; i < NACC; ++i) for (int r = 0; r < 16; ++r) s += acc[i][r];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int NACC, int FILL, int MODE>
void run(int bpc, int iters) {
    float *out; const int blocks = 256 * bpc;
    hipMalloc(&out, blocks * 256 * sizeof(float));
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    k<NACC, FILL, MODE><<<blocks, 256>>>(out, iters, 0.37f); hipDeviceSynchronize();
    hipEventRecord(e0); k<NACC, FILL, MODE><<<blocks, 256>>>(out, iters, 0.37f); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double flops = 2.0 * 32 * 32 * 2 * 8.0 * NACC * iters * (double)blocks * 4;
    printf("mode=%d fill=%2d NACC=%d waves/SIMD=%d : %.1f TFLOP/s\n", MODE, FILL, NACC, bpc, flops / ms / 1e9);
    hipFree(out);
}

int main() {
    run<2, 0, 0>(2, 2000);
    run<2, 4, 0>(1, 2000); run<2, 8, 0>(1, 2000); run<2, 12, 0>(1, 2000); run<2, 16, 0>(1, 2000);
    run<2, 4, 0>(4, 1000); run<2, 8, 0>(4, 1000); run<2, 12, 0>(4, 1000); run<2, 16, 0>(4, 1000);
    run<2, 2, 1>(4, 1000); run<2, 4, 1>(4, 1000); run<2, 8, 1>(4, 1000);
    run<2, 2, 1>(1, 2000); run<2, 4, 1>(1, 2000); run<2, 8, 1>(1, 2000);
    run<2, 8, 2>(4, 1000); run<2, 16, 2>(4, 1000);
    run<2, 2, 3>(4, 1000); run<2, 4, 3>(4, 1000); run<2, 8, 3>(4, 1000);
    return 0;
}
