// chamfer.hip -- exact 1-nearest-neighbour search for the Chamfer loss ("next" row).
//
// Replaces the per-sample faiss GpuIndexFlatL2 build + search + host round trip of
// models/losses.py:220-235, 260-276.  One thread per query point; the database cloud is streamed
// through LDS in tiles of 1024 points (float4, broadcast reads), the same (dx*dx+dy*dy)+dz*dz
// arithmetic as som_assign, ascending j with strict '<' so ties keep the lowest database index.
#include "common.hpp"

namespace {
constexpr int CH_THREADS = 256;
constexpr int CH_TILE = 1024;

__global__ __launch_bounds__(CH_THREADS) void chamfer_nn_kernel(const float *__restrict__ q, const float *__restrict__ db,
                                                                 int32_t *__restrict__ nn, int Nq, int Nd)
{
    __shared__ float4 tile[CH_TILE];
    const int b = blockIdx.y;
    const int i = blockIdx.x * CH_THREADS + threadIdx.x;
    const float *qb = q + (size_t)b * 3 * Nq, *dbb = db + (size_t)b * 3 * Nd;
    const bool valid = i < Nq;
    const float px = valid ? qb[i] : 0.f, py = valid ? qb[Nq + i] : 0.f, pz = valid ? qb[2 * (size_t)Nq + i] : 0.f;
    float best = __builtin_inff();
    int bi = 0;
    for (int t0 = 0; t0 < Nd; t0 += CH_TILE) {
        const int cnt = min(CH_TILE, Nd - t0);
        __syncthreads();
        for (int t = threadIdx.x; t < cnt; t += CH_THREADS)
            tile[t] = make_float4(dbb[t0 + t], dbb[Nd + t0 + t], dbb[2 * (size_t)Nd + t0 + t], 0.f);
        __syncthreads();
#pragma unroll 4
        for (int t = 0; t < cnt; ++t) {
            const float4 p = tile[t];
            const float dx = __fsub_rn(px, p.x), dy = __fsub_rn(py, p.y), dz = __fsub_rn(pz, p.z);
            const float d = __fadd_rn(__fadd_rn(__fmul_rn(dx, dx), __fmul_rn(dy, dy)), __fmul_rn(dz, dz));
            const bool lt = d < best;
            best = lt ? d : best;
            bi = lt ? t0 + t : bi;
        }
    }
    if (valid) nn[(size_t)b * Nq + i] = bi;
}
}  // namespace

extern "C" int sonet_chamfer_nn_f32(const float *q, const float *db, int32_t *nn, int B, int Nq, int Nd,
                                    sonet_stream_t stream)
{
    const char *what = "sonet_chamfer_nn_f32";
    SONET_REQUIRE(q && db && nn, "%s: NULL pointer", what);
    SONET_REQUIRE(B > 0 && Nq > 0 && Nd > 0, "%s: non-positive size", what);
    if (B > 65535) return sonet::fail(SONET_ERR_UNSUPPORTED, "%s: B=%d > 65535", what, B);
    hipLaunchKernelGGL(chamfer_nn_kernel, dim3(sonet::ceil_div(Nq, CH_THREADS), B), dim3(CH_THREADS), 0,
                       sonet::as_stream(stream), q, db, nn, Nq, Nd);
    return sonet::launched(what);
}
