// tools/mfma_power.hip -- what the matrix pipe sustains on REAL data.  A pure v_mfma_f32_32x32x16_f16 loop (one
// wave per SIMD, six independent accumulators) is timed with (a) constant operands and (b) eight rotating sets of
// random fp16 operands, on all CUs and on 32 CUs, together with the shader clock it ran at (s_memtime cycles per
// s_memrealtime tick).  The nominal 2.5 PFLOP/s assumes 2.4 GHz; with toggling operands the chip is power-limited.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

// EXTRA bit 0: every MFMA takes its A operand from a fresh ds_read_b128 (what the fused kernel does);
// EXTRA bit 1: three dependent-free VALU instructions per MFMA on live data (the operand split).
template <int RANDOM, int EXTRA = 0>
__global__ __launch_bounds__(256) void k(float *out, const uint4 *ops, int iters, long long *clk) {
    __shared__ uint4 lds[48 * 64];
    for (int i = threadIdx.x; i < 48 * 64; i += 256) lds[i] = ops[i & 4095];
    __syncthreads();
    const uint4 *lp = &lds[threadIdx.x & 63];
    float v0 = __uint_as_float(ops[threadIdx.x].x & 0x3fffffffu), v1 = __uint_as_float(ops[threadIdx.x].y & 0x3fffffffu), v2 = 0.37f;
    f32x16 acc[6];
    for (int i = 0; i < 6; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    f16x8 a[8], b[8];
    for (int s = 0; s < 8; ++s) {
        const uint4 ua = RANDOM ? ops[(s * 2) * 256 + threadIdx.x] : make_uint4(0x3c003c00u, 0x3c003c00u, 0x3c003c00u, 0x3c003c00u);
        const uint4 ub = RANDOM ? ops[(s * 2 + 1) * 256 + threadIdx.x] : make_uint4(0x38003800u, 0x38003800u, 0x38003800u, 0x38003800u);
        a[s] = __builtin_bit_cast(f16x8, ua); b[s] = __builtin_bit_cast(f16x8, ub);
    }
    const long long c0 = __builtin_readcyclecounter(), r0 = (long long)__builtin_amdgcn_s_memrealtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int s = 0; s < 8; ++s)
#pragma unroll
            for (int i = 0; i < 6; ++i) {
                f16x8 av = a[(s + i) & 7];
                if constexpr (EXTRA & 1) av = __builtin_bit_cast(f16x8, lp[64 * (s * 6 + i)]);
                acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(av, b[s], acc[i], 0, 0, 0);
                if constexpr (EXTRA & 2) {
                    asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(v0) : "v"(v1), "v"(v2));
                    asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(v1) : "v"(v2), "v"(v0));
                    asm volatile("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(v2) : "v"(v0), "v"(v1));
                }
                __builtin_amdgcn_sched_barrier(0);
            }
    }
    if constexpr (EXTRA & 2) acc[0][0] += v0 + v1 + v2;
    const long long c1 = __builtin_readcyclecounter(), r1 = (long long)__builtin_amdgcn_s_memrealtime();
    float sum = 0.f;
    for (int i = 0; i < 6; ++i) for (int r = 0; r < 16; ++r) sum += acc[i][r];
    out[blockIdx.x * 256 + threadIdx.x] = sum;
    if (blockIdx.x == 0 && threadIdx.x == 0) { clk[0] = c1 - c0; clk[1] = r1 - r0; }
}

template <int RANDOM, int EXTRA = 0>
void run(int blocks, int iters, const uint4 *ops) {
    float *out; long long *clk, h[2];
    hipMalloc(&out, blocks * 256 * sizeof(float)); hipMalloc(&clk, 16);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    k<RANDOM, EXTRA><<<blocks, 256>>>(out, ops, iters, clk); hipDeviceSynchronize();
    hipEventRecord(e0); k<RANDOM, EXTRA><<<blocks, 256>>>(out, ops, iters, clk); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    hipMemcpy(h, clk, 16, hipMemcpyDeviceToHost);
    const double flops = 2.0 * 32 * 32 * 16 * 48.0 * iters * (double)blocks * 4;
    printf("%-8s operands%s%s, %3d workgroups (1 per CU), %6.2f ms: %6.0f TFLOP/s  = %5.2f TFLOP/s per CU, shader clock %.3f GHz\n",
           RANDOM ? "random" : "constant", (EXTRA & 1) ? " + ds_read_b128/MFMA" : "", (EXTRA & 2) ? " + 3 VALU/MFMA" : "", blocks, ms, flops / ms / 1e9, flops / ms / 1e9 / blocks, (double)h[0] / ((double)h[1] * 10.0));
    hipFree(out); hipFree(clk);
}

int main() {
    uint4 *ops; const size_t n = 16 * 256;
    uint4 *hops = (uint4 *)malloc(n * sizeof(uint4));
    srand(1);
    for (size_t i = 0; i < n; ++i) {                            // fp16 values in (-2, 2) with random mantissas
        unsigned w[4];
        for (int q = 0; q < 4; ++q) { const unsigned lo = (rand() & 0x83ff) | ((13 + rand() % 3) << 10), hi = (rand() & 0x83ff) | ((13 + rand() % 3) << 10); w[q] = lo | (hi << 16); }
        hops[i] = make_uint4(w[0], w[1], w[2], w[3]);
    }
    hipMalloc(&ops, n * sizeof(uint4)); hipMemcpy(ops, hops, n * sizeof(uint4), hipMemcpyHostToDevice);
    for (int rep = 0; rep < 2; ++rep) {
        run<0>(256, 2000, ops); run<1>(256, 2000, ops); run<0>(256, 20000, ops); run<1>(256, 20000, ops);
        run<0>(32, 20000, ops); run<1>(32, 20000, ops); run<1>(128, 20000, ops);
        run<1, 1>(256, 20000, ops); run<1, 2>(256, 20000, ops); run<1, 3>(256, 20000, ops); run<1, 3>(32, 20000, ops); run<0, 3>(256, 20000, ops);
    }
    return 0;
}
