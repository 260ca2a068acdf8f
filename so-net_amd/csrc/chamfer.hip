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

// Two database points per step in packed f32 (v_pk_add_f32 / v_pk_mul_f32: each lane of a pair is an IEEE single operation, so the
// distances are the bits of the scalar form): the tile holds PAIRS -- (x0, x1, y0, y1) and (z0, z1) -- and the 8 arithmetic operations of a
// pair cost 8 instructions instead of 16; the two compare / select steps stay scalar and in order (ascending j, strict '<').
typedef float ch_f2 __attribute__((ext_vector_type(2)));

__global__ __launch_bounds__(CH_THREADS) void chamfer_nn_kernel(const float *__restrict__ q, const float *__restrict__ db,
                                                                 int32_t *__restrict__ nn, int Nq, int Nd)
{
    __shared__ float4 txy[CH_TILE / 2];                          // (x0, x1, y0, y1) of database points 2 t, 2 t + 1
    __shared__ float2 tz[CH_TILE / 2];                           // (z0, z1)
    const int b = blockIdx.y;
    const int i = blockIdx.x * CH_THREADS + threadIdx.x;
    const float *qb = q + (size_t)b * 3 * Nq, *dbb = db + (size_t)b * 3 * Nd;
    const bool valid = i < Nq;
    const float px = valid ? qb[i] : 0.f, py = valid ? qb[Nq + i] : 0.f, pz = valid ? qb[2 * (size_t)Nq + i] : 0.f;
    const ch_f2 PX = {px, px}, PY = {py, py}, PZ = {pz, pz};
    float best = __builtin_inff();
    int bi = 0;
    for (int t0 = 0; t0 < Nd; t0 += CH_TILE) {
        const int cnt = min(CH_TILE, Nd - t0);
        const int npair = (cnt + 1) >> 1;
        __syncthreads();
        for (int t = threadIdx.x; t < npair; t += CH_THREADS) {
            const int j0 = t0 + 2 * t, j1 = j0 + 1;
            const bool has1 = 2 * t + 1 < cnt;
            // (an odd tail: the second point of the last pair is a copy of the first -- its distance is equal, never strictly smaller)
            const int jb = has1 ? j1 : j0;
            txy[t] = make_float4(dbb[j0], dbb[jb], dbb[Nd + j0], dbb[Nd + jb]);
            tz[t] = make_float2(dbb[2 * (size_t)Nd + j0], dbb[2 * (size_t)Nd + jb]);
        }
        __syncthreads();
#pragma unroll 4
        for (int t = 0; t < npair; ++t) {
            const float4 a = txy[t];
            const float2 c = tz[t];
            const ch_f2 X = {a.x, a.y}, Y = {a.z, a.w}, Z = {c.x, c.y};
            const ch_f2 dx = PX - X, dy = PY - Y, dz = PZ - Z;
            const ch_f2 d = (dx * dx + dy * dy) + dz * dz;         // (-ffp-contract=off: no fused multiply-add)
            const bool lt0 = d[0] < best;
            best = lt0 ? d[0] : best;
            bi = lt0 ? t0 + 2 * t : bi;
            const bool lt1 = d[1] < best;
            best = lt1 ? d[1] : best;
            bi = lt1 ? t0 + 2 * t + 1 : bi;
        }
    }
    if (valid) nn[(size_t)b * Nq + i] = bi;
}
#ifdef SONET_VARIANTS   // (a measured-slower record: variants build only, tools/ + tests/variants)
// ---- both directions in ONE sweep of the distance matrix (models/losses.py:255 and :262 together) -----------------------------
// One thread per point of cloud A (the larger one: more workgroups); cloud B goes through LDS in tiles.  d(a_i, b_j) is
// computed once: the row minimum (a_i's nearest b) is a register update as above; the column minimum (b_j's nearest a) is a
// packed 64-bit key (distance bits << 32 | i) in an LDS bin per b_j -- every lane reads the bin (one broadcast ds_read_b64) and
// only a record-breaker issues the ds_min_u64, as in index_max.  Distances are >= 0, so their bit patterns order like the
// values; equal distances order by the smaller i: exactly "ascending i, strict <".  (dx)^2 == (-dx)^2 bit for bit, so the
// column result equals a separate b -> a launch.  Bins go to memory by atomicMin on the same keys (one per bin, tile and
// workgroup); a NaN distance has a key above +inf and never wins against a finite one; an all-NaN column keeps index 0.
// MEASURED (profiles/r02l_chamfer.log, 1280 x 5000 points): 8.66 ms at B = 64 against 0.34 ms for two one-direction launches,
// which already run at 0.67 of the vector-issue roof (22 lane-ops per pair) -- the dependent LDS read and the divergent branch
// per pair stall a loop that is otherwise pure register arithmetic.  The loss (models/losses.py) therefore keeps the two
// launches; this entry point stays as the tested record of the experiment (identical indices).
constexpr int C2_TILE = 2048;
constexpr unsigned long long C2_INIT = 0x7F800000FFFFFFFFull;          // (+inf, i = 2^32 - 1): what "no candidate yet" compares as

__global__ __launch_bounds__(CH_THREADS) void chamfer_nn2_kernel(const float *__restrict__ a, const float *__restrict__ bdb,
                                                                  int32_t *__restrict__ nn_a, unsigned long long *__restrict__ colkey,
                                                                  int Na, int Nb)
{
    __shared__ float4 tile[C2_TILE];
    __shared__ unsigned long long bins[C2_TILE];
    const int b = blockIdx.y;
    const int i = blockIdx.x * CH_THREADS + threadIdx.x;
    const float *ab = a + (size_t)b * 3 * Na, *bb = bdb + (size_t)b * 3 * Nb;
    const bool valid = i < Na;
    const float px = valid ? ab[i] : 0.f, py = valid ? ab[Na + i] : 0.f, pz = valid ? ab[2 * (size_t)Na + i] : 0.f;
    float best = __builtin_inff();
    int bi = 0;
    for (int t0 = 0; t0 < Nb; t0 += C2_TILE) {
        const int cnt = min(C2_TILE, Nb - t0);
        __syncthreads();
        for (int t = threadIdx.x; t < cnt; t += CH_THREADS) {
            tile[t] = make_float4(bb[t0 + t], bb[Nb + t0 + t], bb[2 * (size_t)Nb + t0 + t], 0.f);
            bins[t] = C2_INIT;
        }
        __syncthreads();
        if (valid) {
#pragma unroll 4
            for (int t = 0; t < cnt; ++t) {
                const float4 p = tile[t];
                const float dx = __fsub_rn(px, p.x), dy = __fsub_rn(py, p.y), dz = __fsub_rn(pz, p.z);
                const float d = __fadd_rn(__fadd_rn(__fmul_rn(dx, dx), __fmul_rn(dy, dy)), __fmul_rn(dz, dz));
                const bool lt = d < best;
                best = lt ? d : best;
                bi = lt ? t0 + t : bi;
                const unsigned long long key = ((unsigned long long)__float_as_uint(d) << 32) | (unsigned)i;
                if (key < bins[t]) atomicMin(&bins[t], key);
            }
        }
        __syncthreads();
        for (int t = threadIdx.x; t < cnt; t += CH_THREADS) {
            const unsigned long long k = bins[t];
            if (k < C2_INIT) atomicMin(colkey + (size_t)b * Nb + t0 + t, k);
        }
    }
    if (valid) nn_a[(size_t)b * Na + i] = bi;
}

__global__ __launch_bounds__(256) void chamfer_nn2_finalize_kernel(const unsigned long long *__restrict__ colkey, int32_t *__restrict__ nn_b, long long n)
{
    const long long t = (long long)blockIdx.x * 256 + threadIdx.x;
    if (t >= n) return;
    const unsigned long long k = colkey[t];
    nn_b[t] = k < C2_INIT ? (int32_t)(unsigned)(k & 0xFFFFFFFFull) : 0;
}
#endif  // SONET_VARIANTS
}  // namespace

#ifdef SONET_VARIANTS
extern "C" size_t sonet_chamfer_nn2_ws_size(int B, int Na, int Nb)
{
    if (B <= 0 || Na <= 0 || Nb <= 0) return 0;
    return (size_t)B * (size_t)(Na > Nb ? Nb : Na) * 8;
}

extern "C" int sonet_chamfer_nn2_f32(const float *pa, const float *pb, int32_t *nn_ab, int32_t *nn_ba, void *ws, int B, int Na, int Nb,
                                     sonet_stream_t stream)
{
    const char *what = "sonet_chamfer_nn2_f32";
    SONET_REQUIRE(pa && pb && nn_ab && nn_ba && ws, "%s: NULL pointer", what);
    SONET_REQUIRE(B > 0 && Na > 0 && Nb > 0, "%s: non-positive size", what);
    if (B > 65535) return sonet::fail(SONET_ERR_UNSUPPORTED, "%s: B=%d > 65535", what, B);
    hipStream_t st = sonet::as_stream(stream);
    // threads over the larger cloud, LDS tiles over the smaller one
    const bool swap = Nb > Na;
    const float *big = swap ? pb : pa, *small = swap ? pa : pb;
    int32_t *nn_big = swap ? nn_ba : nn_ab, *nn_small = swap ? nn_ab : nn_ba;
    const int Nbig = swap ? Nb : Na, Nsmall = swap ? Na : Nb;
    unsigned long long *colkey = reinterpret_cast<unsigned long long *>(ws);
    if (hipMemsetAsync(colkey, 0xFF, (size_t)B * Nsmall * 8, st) != hipSuccess) return sonet::fail(SONET_ERR_LAUNCH, "%s: memset failed", what);
    hipLaunchKernelGGL(chamfer_nn2_kernel, dim3(sonet::ceil_div(Nbig, CH_THREADS), B), dim3(CH_THREADS), 0, st, big, small, nn_big, colkey, Nbig, Nsmall);
    const long long n = (long long)B * Nsmall;
    hipLaunchKernelGGL(chamfer_nn2_finalize_kernel, dim3((unsigned)sonet::ceil_div64(n, 256)), dim3(256), 0, st, colkey, nn_small, n);
    return sonet::launched(what);
}
#endif  // SONET_VARIANTS

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
