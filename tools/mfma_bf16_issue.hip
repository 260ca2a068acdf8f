// tools/mfma_bf16_issue.hip -- sustained v_mfma_f32_32x32x16_bf16 rate of one wave per SIMD, alone and with
// filler VALU / LDS reads per MFMA (what the x3 kernels interleave).  Prices the "2500/6" ceiling used in bench.py.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

// MODE 0: FILL independent VALU per MFMA.  MODE 1: one ds_read_b128 every FILL MFMAs feeding operand A.
template <int NACC, int FILL, int MODE>
__global__ __launch_bounds__(256) void k(float *out, int iters, float seed) {
    __shared__ uint4 lds[64 * 16];
    f32x16 acc[NACC];
    for (int i = 0; i < NACC; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    uint4 ua = make_uint4(threadIdx.x, 0x3f803f80u, 0x3f003f00u, 0x3e803e80u), ub = make_uint4(0x3f803f80u, threadIdx.x * 3, 0x3f003f00u, 0x3e803e80u);
    for (int i = threadIdx.x; i < 64 * 16; i += 256) lds[i] = ua;
    __syncthreads();
    bf16x8 a = __builtin_bit_cast(bf16x8, ua), b = __builtin_bit_cast(bf16x8, ub);
    float f0 = seed, f1 = seed * 0.5f;
    const uint4 *lp = &lds[threadIdx.x & 63];
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 8; ++u)
#pragma unroll
            for (int i = 0; i < NACC; ++i) {
                if (MODE == 0) {
#pragma unroll
                    for (int q = 0; q < FILL; ++q) asm volatile("v_add_f32 %0, %0, %1" : "+v"(f0) : "v"(f1));
                } else if (MODE == 1) {
                    if ((u * NACC + i) % FILL == 0) a = __builtin_bit_cast(bf16x8, lp[64 * ((u * NACC + i) & 15)]);
                }
                acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[i], 0, 0, 0);
            }
    }
    float s = f0;
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
    const double flops = 2.0 * 32 * 32 * 16 * 8.0 * NACC * iters * (double)blocks * 4;
    printf("mode=%d fill=%2d NACC=%d waves/SIMD=%d : %7.3f ms  %.0f TFLOP/s (x3-equivalent %.1f)\n", MODE, FILL, NACC, bpc, ms, flops / ms / 1e9, flops / ms / 1e9 / 6);
    hipFree(out);
}

int main() {
    run<6, 0, 0>(1, 2000); run<6, 0, 0>(1, 20000); run<6, 0, 0>(2, 10000); run<4, 0, 0>(1, 20000);
    run<6, 1, 0>(1, 10000); run<6, 2, 0>(1, 10000); run<6, 4, 0>(1, 10000); run<6, 6, 0>(1, 10000); run<6, 8, 0>(1, 10000);
    run<6, 4, 0>(2, 5000); run<6, 8, 0>(2, 5000);
    run<6, 2, 1>(1, 10000); run<6, 1, 1>(1, 10000);
    return 0;
}
