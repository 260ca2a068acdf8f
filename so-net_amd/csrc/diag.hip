// diag.hip -- a measuring stick, not part of the data path: what the matrix pipe of THIS device sustains.
//
// The fused first-PointNet kernel is MFMA-bound, and on the whole chip the MFMA rate is set by power, not by the
// 2.4 GHz nominal clock: a pure v_mfma_f32_32x32x16_f16 loop (one wave per SIMD, six independent accumulators,
// nothing else) holds 2.46 PFLOP/s at 2.39 GHz on constant operands but only 1.6-1.7 PFLOP/s at 1.6-1.7 GHz on
// operands with random mantissas, while 32 CUs alone keep 2.39 GHz either way (tools/mfma_power.hip,
// profiles/r01j_mfma_power.log).  bench.py calls this entry next to the timed region so that the roofline line can
// quote the fraction of the rate the chip can actually hold on real data beside the nominal one.
#include "common.hpp"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

__device__ __forceinline__ unsigned mix(unsigned x) {          // integer hash (lowbias32)
    x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
    return x;
}
// two fp16 values in (-2, 2) with random sign and mantissa (exponents 13..15)
__device__ __forceinline__ unsigned rand_f16_pair(unsigned seed) {
    const unsigned r = mix(seed), e = mix(seed ^ 0x9e3779b9u);
    const unsigned lo = (r & 0x83ffu) | ((13u + e % 3u) << 10), hi = ((r >> 16) & 0x83ffu) | ((13u + (e >> 8) % 3u) << 10);
    return lo | (hi << 16);
}

__global__ __launch_bounds__(256) void mfma_rate_kernel(int random_operands, int iters, float *sink, long long *clk)
{
    f32x16 acc[6];
#pragma unroll
    for (int i = 0; i < 6; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    f16x8 a[8], b[8];
#pragma unroll
    for (int s = 0; s < 8; ++s) {
        uint4 ua = make_uint4(0x3c003c00u, 0x3c003c00u, 0x3c003c00u, 0x3c003c00u);      // 1.0
        uint4 ub = make_uint4(0x38003800u, 0x38003800u, 0x38003800u, 0x38003800u);      // 0.5
        if (random_operands) {
            const unsigned t = (blockIdx.x * 256u + threadIdx.x) * 64u + s * 8u;
            ua = make_uint4(rand_f16_pair(t), rand_f16_pair(t + 1), rand_f16_pair(t + 2), rand_f16_pair(t + 3));
            ub = make_uint4(rand_f16_pair(t + 4), rand_f16_pair(t + 5), rand_f16_pair(t + 6), rand_f16_pair(t + 7));
        }
        a[s] = __builtin_bit_cast(f16x8, ua);
        b[s] = __builtin_bit_cast(f16x8, ub);
    }
    const long long c0 = __builtin_readcyclecounter(), r0 = (long long)__builtin_amdgcn_s_memrealtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int s = 0; s < 8; ++s)
#pragma unroll
            for (int i = 0; i < 6; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[(s + i) & 7], b[s], acc[i], 0, 0, 0);
    }
    const long long c1 = __builtin_readcyclecounter(), r1 = (long long)__builtin_amdgcn_s_memrealtime();
    float sum = 0.f;
#pragma unroll
    for (int i = 0; i < 6; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) sum += acc[i][r];
    sink[blockIdx.x * 256 + threadIdx.x] = sum;
    if (blockIdx.x == 0 && threadIdx.x == 0) { clk[0] = c1 - c0; clk[1] = r1 - r0; }   // shader cycles, 100 MHz ticks
}

}  // namespace

extern "C" int sonet_diag_mfma_f16_rate(int random_operands, int iters, double *tflops_out, double *ghz_out, sonet_stream_t stream)
{
    const char *what = "sonet_diag_mfma_f16_rate";
    SONET_REQUIRE(tflops_out && ghz_out, "%s: NULL pointer", what);
    SONET_REQUIRE(iters > 0 && iters <= (1 << 22), "%s: iters=%d out of range", what, iters);
    int dev = 0, cus = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus <= 0)
        return sonet::fail(SONET_ERR_NO_DEVICE, "%s: no HIP device is current", what);
    hipStream_t st = sonet::as_stream(stream);
    float *sink = nullptr;
    long long *clk = nullptr;
    hipEvent_t e0 = nullptr, e1 = nullptr;
    int rc = SONET_OK;
    if (hipMalloc(&sink, (size_t)cus * 256 * sizeof(float)) != hipSuccess || hipMalloc(&clk, 2 * sizeof(long long)) != hipSuccess ||
        hipEventCreate(&e0) != hipSuccess || hipEventCreate(&e1) != hipSuccess) {
        rc = sonet::fail(SONET_ERR_LAUNCH, "%s: could not allocate scratch", what);
    } else {
        hipLaunchKernelGGL(mfma_rate_kernel, dim3(cus), dim3(256), 0, st, random_operands, iters / 8 + 1, sink, clk);   // clocks settle
        (void)hipEventRecord(e0, st);
        hipLaunchKernelGGL(mfma_rate_kernel, dim3(cus), dim3(256), 0, st, random_operands, iters, sink, clk);
        (void)hipEventRecord(e1, st);
        float ms = 0.f;
        long long h[2] = {0, 0};
        if (hipEventSynchronize(e1) != hipSuccess || hipEventElapsedTime(&ms, e0, e1) != hipSuccess || ms <= 0.f ||
            hipMemcpy(h, clk, sizeof(h), hipMemcpyDeviceToHost) != hipSuccess) {
            rc = sonet::fail(SONET_ERR_LAUNCH, "%s: %s", what, hipGetErrorString(hipGetLastError()));
        } else {
            const double flops = 2.0 * 32 * 32 * 16 * 48.0 * (double)iters * cus * 4.0;
            *tflops_out = flops / ((double)ms * 1e9);
            *ghz_out = h[1] > 0 ? (double)h[0] / ((double)h[1] * 10.0) : 0.0;
        }
    }
    if (e0) (void)hipEventDestroy(e0);
    if (e1) (void)hipEventDestroy(e1);
    if (sink) (void)hipFree(sink);
    if (clk) (void)hipFree(clk);
    return rc;
}
