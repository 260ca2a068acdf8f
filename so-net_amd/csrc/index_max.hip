// index_max.hip -- per-node segmented arg-max pool for gfx950 (MI355X).
//
// Replaces models/index_max_ext/index_max_cuda.cu:10-26 (one CUDA *thread* per (b,c) row, scanning
// its N' values serially and uncoalesced) and its CPU twin index_max.cpp:73-112.
//
// Design (HBM-bound, 23.2 MB per cloud at C=384, N'=15000 -- DESIGN.md "index_max"):
//   * one 256-thread workgroup owns R consecutive channel rows of ONE cloud; its 4 waves stream
//     disjoint 256-element chunks of those rows with 16-byte loads (64 lanes x float4 = 1 KiB per
//     instruction, fully coalesced along n).  The node-id chunk (int4) is loaded once per chunk and
//     reused for the R rows, so the shared B x N' id row costs 1/R of the data traffic and stays in L2.
//   * per row, K bins live in LDS as packed 64-bit keys  (orderable(value) << 32) | (0xFFFFFFFF - n).
//     "key > bin" is exactly "this element beats the running maximum under the reference's sequential
//     strict-'>' scan" (bigger value wins, equal value -> smaller n wins).  Each element does one
//     ds_read_b64 of its bin and only the rare record-breakers (about ln(n) per bin) issue a
//     ds_max_u64, so LDS atomics never limit the stream.  Integer max is order-independent, hence the
//     result is deterministic and identical to the sequential scan.
//   * bins start at (orderable(-1000) << 32 | 0xFFFFFFFF) == "value -1000 at position 0": values
//     <= -1000, NaN (mapped to key 0) and empty segments leave position 0, as the reference does.
//     -0.0 is canonicalised to +0.0 first because IEEE '>' treats them as equal.
//   * epilogue: one coalesced store of the R x K positions (and, for the gather variant, the value at
//     that position, masked by row_max -- models/networks.py:185).
//   * workgroup ids are remapped so that the ~C/R workgroups of a cloud run on one XCD and share its
//     L2 copy of the id row (speed only).
#include "common.hpp"

namespace {

constexpr int IM_THREADS = 256;
constexpr unsigned long long IM_INIT_KEY = (0x3B85FFFFull << 32) | 0xFFFFFFFFull;  // ord(-1000.0f) = ~0xC47A0000, n = 0

__device__ __forceinline__ unsigned ord_f32(unsigned bits) {
    // total order on floats as unsigned ints; -0.0 == +0.0; NaN -> 0 (never wins)
    if (bits == 0x80000000u) bits = 0u;
    const unsigned mag = bits & 0x7FFFFFFFu;
    const unsigned o = bits ^ ((unsigned)((int)bits >> 31) | 0x80000000u);
    return mag > 0x7F800000u ? 0u : o;
}
static_assert((0xC47A0000u ^ 0xFFFFFFFFu) == 0x3B85FFFFu, "ord(-1000) check");

__device__ __forceinline__ void im_update(unsigned long long *bins, int id, int K, unsigned vbits, unsigned n) {
    if ((unsigned)id >= (unsigned)K) return;  // out-of-range id: ignored (the reference would corrupt memory)
    const unsigned long long key = ((unsigned long long)ord_f32(vbits) << 32) | (unsigned long long)(0xFFFFFFFFu - n);
    if (key > bins[id]) atomicMax(&bins[id], key);
}

template <typename T> struct Vec4;
typedef float im_f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned short im_u16x4 __attribute__((ext_vector_type(4)));
template <> struct Vec4<float> {
    using type = float4;
    static __device__ __forceinline__ float4 load_nt(const float *row, int v) {       // read-once stream: non-temporal
        return __builtin_bit_cast(float4, __builtin_nontemporal_load(reinterpret_cast<const im_f32x4 *>(row) + v));
    }
    static __device__ __forceinline__ void bits(const float4 &v, unsigned (&o)[4]) {
        o[0] = __float_as_uint(v.x); o[1] = __float_as_uint(v.y); o[2] = __float_as_uint(v.z); o[3] = __float_as_uint(v.w);
    }
    static __device__ __forceinline__ unsigned bits1(float v) { return __float_as_uint(v); }
    static __device__ __forceinline__ float to_f32(float v) { return v; }
};
template <> struct Vec4<uint16_t> {  // bfloat16 bits
    using type = ushort4;
    static __device__ __forceinline__ ushort4 load_nt(const uint16_t *row, int v) {
        return __builtin_bit_cast(ushort4, __builtin_nontemporal_load(reinterpret_cast<const im_u16x4 *>(row) + v));
    }
    static __device__ __forceinline__ void bits(const ushort4 &v, unsigned (&o)[4]) {
        o[0] = (unsigned)v.x << 16; o[1] = (unsigned)v.y << 16; o[2] = (unsigned)v.z << 16; o[3] = (unsigned)v.w << 16;
    }
    static __device__ __forceinline__ unsigned bits1(uint16_t v) { return (unsigned)v << 16; }
    static __device__ __forceinline__ float to_f32(uint16_t v) { return __uint_as_float((unsigned)v << 16); }
};

// R rows per workgroup; VEC = true requires Np % 4 == 0 (16-byte aligned rows).
template <typename T, int R, bool VEC>
__global__ __launch_bounds__(IM_THREADS) void index_max_kernel(
    const T *__restrict__ data, const int32_t *__restrict__ index, const int32_t *__restrict__ row_max,
    int32_t *__restrict__ out_idx, float *__restrict__ out_val, int C, int Np, int K)
{
    extern __shared__ __attribute__((aligned(16))) unsigned long long bins[];  // [R][K]
    const int tid = threadIdx.x;
    const int grp = xcd_remap(blockIdx.x, gridDim.x);
    const long long row0 = (long long)grp * R;        // flattened (b*C + c); R divides C
    const int b = (int)(row0 / C);
    const T *drow = data + row0 * (long long)Np;
    const int32_t *irow = index + (long long)b * Np;

    for (int i = tid; i < R * K; i += IM_THREADS) bins[i] = IM_INIT_KEY;
    __syncthreads();

    if constexpr (VEC) {
        using V = typename Vec4<T>::type;
        const int nvec = Np >> 2;
#pragma unroll 2
        for (int v = tid; v < nvec; v += IM_THREADS) {
            const int4 id = reinterpret_cast<const int4 *>(irow)[v];
            V d[R];
#pragma unroll
            for (int r = 0; r < R; ++r) d[r] = Vec4<T>::load_nt(drow + (long long)r * Np, v);
            const unsigned n = (unsigned)v << 2;
#pragma unroll
            for (int r = 0; r < R; ++r) {
                unsigned vb[4];
                Vec4<T>::bits(d[r], vb);
                unsigned long long *bb = bins + r * K;
                im_update(bb, id.x, K, vb[0], n);
                im_update(bb, id.y, K, vb[1], n + 1);
                im_update(bb, id.z, K, vb[2], n + 2);
                im_update(bb, id.w, K, vb[3], n + 3);
            }
        }
    } else {
        for (int n = tid; n < Np; n += IM_THREADS) {
            const int id = irow[n];
#pragma unroll
            for (int r = 0; r < R; ++r)
                im_update(bins + r * K, id, K, Vec4<T>::bits1(drow[(long long)r * Np + n]), (unsigned)n);
        }
    }
    __syncthreads();

    for (int i = tid; i < R * K; i += IM_THREADS) {
        const int r = i / K, m = i - r * K;
        const int pos = (int)(0xFFFFFFFFu - (unsigned)(bins[i] & 0xFFFFFFFFull));
        out_idx[(row0 + r) * K + m] = pos;
        if (out_val != nullptr) {
            // the winner's value is the high half of its key: no second trip to HBM for the gather (1.6 M scattered 4-byte reads
            // at B = 64 cost a 64-byte sector each: +19 % traffic, profiles/pmc_traffic.json).  Only a bin that nothing beat
            // (position 0: the reference gathers element 0 of the row), a masked node, or a zero (the key holds +0 for -0)
            // reads the row itself -- element 0 / the winner, one cache line per row.
            const unsigned okey = (unsigned)(bins[i] >> 32);
            const bool won = bins[i] != IM_INIT_KEY && (row_max == nullptr || row_max[(long long)b * K + m] != 0);
            float v = __uint_as_float((okey & 0x80000000u) ? (okey ^ 0x80000000u) : ~okey);
            if (!won) v = Vec4<T>::to_f32(drow[(long long)r * Np]);
            else if (v == 0.f) v = Vec4<T>::to_f32(drow[(long long)r * Np + pos]);
            out_val[(row0 + r) * K + m] = v;
        }
    }
}

template <typename T>
int launch_index_max(const T *data, const int32_t *index, const int32_t *row_max, int32_t *out_idx,
                     float *out_val, int B, int C, int Np, int K, hipStream_t st, const char *what)
{
    SONET_REQUIRE(data && index && out_idx, "%s: NULL pointer", what);
    SONET_REQUIRE(B > 0 && C > 0 && Np > 0 && K > 0, "%s: non-positive size B=%d C=%d Np=%d K=%d", what, B, C, Np, K);
    if (K > 1024) return sonet::fail(SONET_ERR_UNSUPPORTED, "%s: K=%d > 1024 bins", what, K);
    const bool vec = (Np % 4 == 0) && ((reinterpret_cast<uintptr_t>(data) & 15) == 0) &&
                     ((reinterpret_cast<uintptr_t>(index) & 15) == 0);
    // rows per workgroup: as many as divide C (id-chunk reuse), but keep >= ~4 workgroups per CU in flight
    int R = 1;
    const long long rows = (long long)B * C;
    for (int cand : {8, 4, 2}) {
        if (C % cand == 0 && rows / cand >= 1024 && (size_t)cand * K * 8 <= 32768) { R = cand; break; }
    }
    if (R == 1) for (int cand : {4, 2}) if (C % cand == 0 && (size_t)cand * K * 8 <= 32768) { R = cand; break; }
    const long long nwg = rows / R;
    if (nwg > 0x7FFFFFFFll) return sonet::fail(SONET_ERR_UNSUPPORTED, "%s: too many rows", what);
    const size_t lds = (size_t)R * K * sizeof(unsigned long long);
    dim3 grid((unsigned)nwg), block(IM_THREADS);
#define IM_LAUNCH(RR, VV) \
    hipLaunchKernelGGL((index_max_kernel<T, RR, VV>), grid, block, lds, st, data, index, row_max, out_idx, out_val, C, Np, K)
    if (vec) {
        switch (R) { case 8: IM_LAUNCH(8, true); break; case 4: IM_LAUNCH(4, true); break;
                     case 2: IM_LAUNCH(2, true); break; default: IM_LAUNCH(1, true); }
    } else {
        switch (R) { case 8: IM_LAUNCH(8, false); break; case 4: IM_LAUNCH(4, false); break;
                     case 2: IM_LAUNCH(2, false); break; default: IM_LAUNCH(1, false); }
    }
#undef IM_LAUNCH
    return sonet::launched(what);
}

// ---- the same pool over an activation that only exists as P16 planes (csrc/pointmlp_h3p.hip) -----------------------------------------
// P[b][kc][form][h][l][8] fp16, value = (form 0 + form 1) / 32 exactly (22 significand bits), element e of half h = channel
// 16 kc + 4 h + (e & 3) + 8 (e >> 2).  One workgroup owns (b, kc, h) = 8 channels of a cloud: a lane's two 16-byte loads (one per
// form, consecutive lanes = consecutive columns: 1 KiB per wave and instruction) are the 8 channel values of ONE column, and the bins
// of the 8 rows are updated exactly as above.  The segmenter's first PointNet then need not write first_pn_out in f32 at all.
__device__ __forceinline__ void p16_unpack(const uint4 &hi, const uint4 &mid, float (&v)[8]) {
    const unsigned hw[4] = {hi.x, hi.y, hi.z, hi.w}, mw[4] = {mid.x, mid.y, mid.z, mid.w};
#pragma unroll
    for (int p = 0; p < 4; ++p) {
        typedef _Float16 h2 __attribute__((ext_vector_type(2)));
        const h2 a = __builtin_bit_cast(h2, hw[p]), c = __builtin_bit_cast(h2, mw[p]);
        v[2 * p] = ((float)a[0] + (float)c[0]) * 0.03125f;              // exact: the two pieces do not overlap, 1/32 is a power of two
        v[2 * p + 1] = ((float)a[1] + (float)c[1]) * 0.03125f;
    }
}

__global__ __launch_bounds__(IM_THREADS) void index_max_p16_kernel(const uint4 *__restrict__ planes, const int32_t *__restrict__ index,
                                                                  const int32_t *__restrict__ row_max, int32_t *__restrict__ out_idx,
                                                                  float *__restrict__ out_val, int C, int KC, int Np, int K)
{
    extern __shared__ __attribute__((aligned(16))) unsigned long long bins[];  // [8][K]
    const int tid = threadIdx.x;
    const int grp = xcd_remap(blockIdx.x, gridDim.x);               // (b, kc, h): the 2 KC workgroups of a cloud share its id row
    const int h = grp & 1, kc = (grp >> 1) % KC, b = (grp >> 1) / KC;
    const uint4 *p_hi = planes + ((((size_t)b * KC + kc) * 2 + 0) * 2 + h) * (size_t)Np;
    const uint4 *p_mid = planes + ((((size_t)b * KC + kc) * 2 + 1) * 2 + h) * (size_t)Np;
    const int32_t *irow = index + (long long)b * Np;
    for (int i = tid; i < 8 * K; i += IM_THREADS) bins[i] = IM_INIT_KEY;
    __syncthreads();
#pragma unroll 2
    for (int n = tid; n < Np; n += IM_THREADS) {
        const int id = irow[n];
        float v[8];
        p16_unpack(p_hi[n], p_mid[n], v);
#pragma unroll
        for (int e = 0; e < 8; ++e) im_update(bins + e * K, id, K, __float_as_uint(v[e]), (unsigned)n);
    }
    __syncthreads();
    for (int i = tid; i < 8 * K; i += IM_THREADS) {
        const int e = i / K, m = i - e * K;
        const int c = 16 * kc + 4 * h + (e & 3) + 8 * (e >> 2);
        if (c >= C) continue;                                          // (channels of a padded last chunk)
        const int pos = (int)(0xFFFFFFFFu - (unsigned)(bins[i] & 0xFFFFFFFFull));
        const long long o = ((long long)b * C + c) * K + m;
        out_idx[o] = pos;
        if (out_val != nullptr) {
            const unsigned okey = (unsigned)(bins[i] >> 32);
            const bool won = bins[i] != IM_INIT_KEY && (row_max == nullptr || row_max[(long long)b * K + m] != 0);
            float v = __uint_as_float((okey & 0x80000000u) ? (okey ^ 0x80000000u) : ~okey);
            if (!won || v == 0.f) {                                    // element 0 of the row / the winner itself (a zero: the key holds +0 for -0)
                float w[8];
                const int q = won ? pos : 0;
                p16_unpack(p_hi[q], p_mid[q], w);
                v = w[e];
            }
            out_val[o] = v;
        }
    }
}

}  // namespace

extern "C" int sonet_index_max_f32(const float *data, const int32_t *index, int32_t *out_idx,
                                   int B, int C, int Np, int K, sonet_stream_t stream) {
    return launch_index_max<float>(data, index, nullptr, out_idx, nullptr, B, C, Np, K, sonet::as_stream(stream),
                                   "sonet_index_max_f32");
}

extern "C" int sonet_index_max_bf16(const uint16_t *data, const int32_t *index, int32_t *out_idx,
                                    int B, int C, int Np, int K, sonet_stream_t stream) {
    return launch_index_max<uint16_t>(data, index, nullptr, out_idx, nullptr, B, C, Np, K, sonet::as_stream(stream),
                                      "sonet_index_max_bf16");
}

extern "C" int sonet_index_max_gather_f32(const float *data, const int32_t *index, const int32_t *row_max,
                                          int32_t *out_idx, float *out_val, int B, int C, int Np, int K,
                                          sonet_stream_t stream) {
    SONET_REQUIRE(out_val, "sonet_index_max_gather_f32: out_val is NULL");
    return launch_index_max<float>(data, index, row_max, out_idx, out_val, B, C, Np, K, sonet::as_stream(stream),
                                   "sonet_index_max_gather_f32");
}

extern "C" int sonet_index_max_gather_bf16(const uint16_t *data, const int32_t *index, const int32_t *row_max,
                                           int32_t *out_idx, float *out_val, int B, int C, int Np, int K,
                                           sonet_stream_t stream) {
    SONET_REQUIRE(out_val, "sonet_index_max_gather_bf16: out_val is NULL");
    return launch_index_max<uint16_t>(data, index, row_max, out_idx, out_val, B, C, Np, K, sonet::as_stream(stream),
                                      "sonet_index_max_gather_bf16");
}

/* index_max_gather over an activation given as P16 planes (sonet_p16_size(B, C, Np) bytes, the layout of sonet_pointmlp_h3p): the
 * values are (hi + mid) / 32 exactly; same rules as sonet_index_max_gather_f32 (-1000 / position 0 for empty segments, first maximum
 * wins, row_max masks nodes).  Np below 2^31, K <= 512. */
extern "C" int sonet_index_max_gather_p16(const void *planes, const int32_t *index, const int32_t *row_max,
                                          int32_t *out_idx, float *out_val, int B, int C, int Np, int K, sonet_stream_t stream)
{
    const char *what = "sonet_index_max_gather_p16";
    SONET_REQUIRE(planes && index && out_idx && out_val, "%s: NULL pointer", what);
    SONET_REQUIRE(B > 0 && C > 0 && Np > 0 && K > 0, "%s: non-positive size B=%d C=%d Np=%d K=%d", what, B, C, Np, K);
    if (K > 512) return sonet::fail(SONET_ERR_UNSUPPORTED, "%s: K=%d > 512 bins", what, K);
    if (reinterpret_cast<uintptr_t>(planes) & 15) return sonet::fail(SONET_ERR_INVALID_ARG, "%s: planes must be 16-byte aligned", what);
    const int KC = sonet::ceil_div(C, 16);
    const long long nwg = (long long)B * KC * 2;
    if (nwg > 0x7FFFFFFFll) return sonet::fail(SONET_ERR_UNSUPPORTED, "%s: too many rows", what);
    hipLaunchKernelGGL(index_max_p16_kernel, dim3((unsigned)nwg), dim3(IM_THREADS), (size_t)8 * K * sizeof(unsigned long long), sonet::as_stream(stream),
                       reinterpret_cast<const uint4 *>(planes), index, row_max, out_idx, out_val, C, KC, Np, K);
    return sonet::launched(what);
}
