// tools/mfma_peak.hip -- what f32 MFMA rate and shader clock does THIS chip sustain?  (calibrates the roofline)
//   hipcc --offload-arch=gfx950 -O3 tools/mfma_peak.hip -o /tmp/mfma_peak && /tmp/mfma_peak
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int NACC>
__global__ __launch_bounds__(256) void mfma_loop(float *out, long long *clk, int iters, float seed) {
    f32x16 acc[NACC];
    for (int i = 0; i < NACC; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    float a = seed + threadIdx.x * 1e-3f, b = seed * 0.5f + threadIdx.x * 2e-3f;
    long long t0 = clock64(), w0 = wall_clock64();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 8; ++u)
#pragma unroll
            for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[i], 0, 0, 0);
    }
    long long t1 = clock64(), w1 = wall_clock64();
    float s = 0.f;
    for (int i = 0; i < NACC; ++i) for (int r = 0; r < 16; ++r) s += acc[i][r];
    out[blockIdx.x * 256 + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) { clk[0] = t1 - t0; clk[1] = w1 - w0; }
}

template <int NACC>
void run(int blocks_per_cu, int iters) {
    float *out; long long *clk;
    const int blocks = 256 * blocks_per_cu;
    hipMalloc(&out, blocks * 256 * sizeof(float)); hipMalloc(&clk, 16);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    mfma_loop<NACC><<<blocks, 256>>>(out, clk, iters, 0.37f);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    mfma_loop<NACC><<<blocks, 256>>>(out, clk, iters, 0.37f);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    long long h[2]; hipMemcpy(h, clk, 16, hipMemcpyDeviceToHost);
    const double flops = 2.0 * 32 * 32 * 2 * 8.0 * NACC * iters * (double)blocks * 4;
    printf("NACC=%d waves/SIMD=%d : %.3f ms  %.1f TFLOP/s   shader clock %.0f MHz (s_memtime ticks %lld per %lld x 10ns)\n",
           NACC, blocks_per_cu, ms, flops / ms / 1e9, (double)h[0] / ((double)h[1] * 0.01), h[0], h[1]);
    hipFree(out); hipFree(clk);
}

int main() {
    run<1>(1, 20000); run<2>(1, 10000); run<4>(1, 5000); run<2>(2, 5000); run<2>(4, 2500); run<4>(2, 2500);
    return 0;
}
