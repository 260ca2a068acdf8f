// node_stage.hip -- glue kernels of the no-grad node-level stage (KNNModule + final PointNet, models/layers.py:313-367,384-387,
// models/networks.py:187-197) on the third-generation layer kernel (pointmlp_h3p.hip).
//
// The stage runs on a FLAT column axis: one "cloud" whose columns are the B x M nodes of the batch (M-level tensors, column b M + m, the axis
// padded with zero columns to Lm = a multiple of 128: sonet_node_stage_columns) or
// their B x M x K neighbour copies (K-level tensor).  Point-wise layers do not care where a cloud ends, and 64 clouds x 64 nodes are
// 4096 columns -- a launch per cloud-sized piece would leave the chip empty.  The K-level tensor is laid out for the group-max epilogue
// (sonet_pointmlp_h3p_gmax): every 128-column block holds G = min(16, floor(128 / K)) nodes, the K neighbour copies of a node next to each other
// (column 128 i + g K + k is neighbour k of node i G + g), the columns behind G K are zero padding.
//
// KNNModule up to the input of its second layer (models/layers.py:319-352 + the first MyConv2d) is
//   h1[c][node n, neighbour k] = act(scale[c] (z[c][b M + I[b][m][k]] + wl[c][0..2] . (coord[b][:, I[b][m][k]] - center[b][:, m])) + shift[c])
// with z = W[:, 3:] . features (the layer is linear: its 384-channel block is applied once per node by a sonet_pointmlp_h3p launch, not
// once per neighbour copy).  sonet_knn_stage_prepare_f32 (needs only the node coordinates: it runs before the first PointNet) writes a
// 16-byte record per K-level column -- source column, de-centred coordinates -- and the neighbourhood centres, as f32 [B][3][M]
// (KNNModule's first return value) and as a one-chunk P16 panel (the 3 leading channels of the final PointNet's input);
// sonet_knn_stage_input_p16 turns the records and z (pre-split: two 16-byte gathers per lane and chunk) into h1 as P16 planes -- the
// operand format of the next layer.
#include "common.hpp"

namespace {

__device__ __forceinline__ void split_pair(float x0, float x1, unsigned &h, unsigned &m) { p16_split_pair(x0, x1, h, m); }
__device__ __forceinline__ int p16_channel(int h, int e) { return 4 * h + (e & 3) + 8 * (e >> 2); }

constexpr int KS_CPW = 4;            // 16-channel chunks per workgroup of the input kernel

// Per padded column of the K-level tensor: rec = (source column of the neighbour on the flat M-level axis, the three de-centred
// coordinates).  src = -1: neighbour index outside [0, M) (its features read as zeros, its coordinates as 0 - centre, as in
// sonet_knn_group_f32); src = -2: a padding column (all zeros).  The thread of a node's first neighbour also writes the centre: f32
// [B][3][M] and the one-chunk P16 panel (channels 0..2 = elements 0..2 of half 0; everything else zero).  One thread per column.
__global__ __launch_bounds__(256) void knn_stage_prepare_kernel(const float *__restrict__ coord, const int64_t *__restrict__ I, int KI, int avg,
                                                                 int B, int M, int K, int G, long long Lp, long long Lm,
                                                                 float *__restrict__ center, uint4 *__restrict__ center_p16, int4 *__restrict__ rec,
                                                                 unsigned *__restrict__ rlog)
{
    const long long l = (long long)blockIdx.x * 256 + threadIdx.x;
    if (l >= Lp) return;
    const int col = (int)(l & 127);
    const long long BM = (long long)B * M;
    const int g = col / K, k = col - g * K;
    const long long n = (l >> 7) * G + g;                                // flat node b M + m
    if (!(g < G && n < BM)) {
        rec[l] = make_int4(-2, 0, 0, 0);
        return;
    }
    const int b = (int)(n / M), m = (int)(n - (long long)b * M);
    const int64_t *Ib = I + ((long long)b * M + m) * KI;
    const float *cb = coord + (long long)b * 3 * M;
    const long long id = Ib[k];
    const bool ok = (unsigned long long)id < (unsigned long long)M;
    float d[3], ctr[3];
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        const float *src = cb + c * M;
        if (avg) {                                                        // (the order and the division of knn_prepare / the reference's mean)
            float sum = 0.f;
            for (int kk = 0; kk < K; ++kk) {
                const long long ik = Ib[kk];
                sum += ((unsigned long long)ik < (unsigned long long)M) ? src[ik] : 0.f;
            }
            ctr[c] = sum / (float)K;
        } else {
            ctr[c] = src[m];
        }
        d[c] = (ok ? src[id] : 0.f) - ctr[c];
    }
    rec[l] = make_int4(ok ? (int)((long long)b * M + id) : -1, __float_as_int(d[0]), __float_as_int(d[1]), __float_as_int(d[2]));
    if (k == 0) {
#pragma unroll
        for (int c = 0; c < 3; ++c) center[((long long)b * 3 + c) * M + m] = ctr[c];
        unsigned h0, m0, h1v, m1;
        split_pair(ctr[0], ctr[1], h0, m0);
        split_pair(ctr[2], 0.f, h1v, m1);
        center_p16[(0 * 2 + 0) * Lm + n] = make_uint4(h0, h1v, 0u, 0u);
        center_p16[(0 * 2 + 1) * Lm + n] = make_uint4(0u, 0u, 0u, 0u);
        center_p16[(1 * 2 + 0) * Lm + n] = make_uint4(m0, m1, 0u, 0u);
        center_p16[(1 * 2 + 1) * Lm + n] = make_uint4(0u, 0u, 0u, 0u);
        if (rlog != nullptr) {                                            // (one atomic per node at most, and only while it still raises the word)
            RangeAcc xr = {0, 0u};
            range_track(xr, ctr[0], ctr[1]);
            range_track(xr, ctr[2], 0.f);
            const unsigned bits = range_amax_bits(xr);
            if (bits > __atomic_load_n(rlog + 2, __ATOMIC_RELAXED)) atomicMax(rlog + 2, bits);
        }
    }
}

// grid (column blocks, ceil(KC / KS_CPW)); 256 threads = 128 columns x 2 halves.  z arrives pre-split (the P16 output of the per-node
// launch): one lane's 8 channels of a neighbour are TWO 16-byte gathers (hi, residual) instead of eight dword gathers from an f32 map.
// The kernel is bound by its vector arithmetic (19 M values x decode + 3 fma + affine + ReLU + range + split, at the clock the first
// PointNet leaves behind), so the arithmetic is written on channel PAIRS (v_pk_fma_f32 / v_pk_add_f32, packed conversions) and every
// power of two rides in a coefficient: the planes hold 32 z and the split wants 32 h1, so the table keeps 32 w, scale, 32 shift --
// a = 32 z + (32 w) . d is 32 x the reference's sum bit for bit, and (32 s)(a / 32) ... no: act(a * scale + 32 shift) = 32 h1 exactly
// (powers of two commute with the rounding of an fma below overflow, and |32 h1| <= 65504 is the range the split clamps to anyway).
typedef float ks_f2 __attribute__((ext_vector_type(2)));
typedef _Float16 ks_h2 __attribute__((ext_vector_type(2)));

constexpr int KS_MAXG = 8;           // chunk groups one workgroup walks at most (its coefficient table: 5 x 8 x 64 floats)

// A workgroup = 128 columns x the chunk groups blockIdx.y, blockIdx.y + gridDim.y, ...: the gathers of the NEXT group are in flight while
// the current one is computed and stored.  (One group per workgroup and the whole grid resident at once ran every workgroup through
// "gather, compute, store" in lock step: 41 us in the forward's graph, 21 without its stores, 25 without its gathers -- the phases did
// not overlap.)
__global__ __launch_bounds__(256) void knn_stage_input_kernel(const int4 *__restrict__ rec, const uint4 *__restrict__ zp, const float *__restrict__ wl,
                                                               const float *__restrict__ scale, const float *__restrict__ shift, int relu,
                                                               int C, long long Lp, long long Lm, uint4 *__restrict__ h1, unsigned *__restrict__ rlog, int abl)
{
    __shared__ unsigned wmax[4];
    // coefficients of the workgroup's channels, [kind][local group][channel in P16 element order of (chunk, half)]: a lane's pair (e, e + 1)
    // of a half is two consecutive floats of every kind -- one 8-byte broadcast read each.  kind: 32 w0, 32 w1, 32 w2, scale, 32 shift.
    // (Left as scalar loads -- the channel is uniform over a wave -- every element waited for its own round trip to the scalar cache.)
    __shared__ __attribute__((aligned(16))) float coef[5][KS_MAXG * KS_CPW * 16];
    const int col = threadIdx.x & 127;
    const int hh = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 7));
    const long long lcol = (long long)blockIdx.x * 128 + col;
    const int4 r = rec[lcol];
    const int KC = (C + 15) >> 4;
    const int ngrp = (KC + KS_CPW - 1) / KS_CPW, gstride = (int)gridDim.y;
    for (int t = threadIdx.x; t < KS_MAXG * KS_CPW * 16; t += 256) {
        // slot = local group j, chunk i, half h, element e  <->  channel 16 (chunk) + 4 h + (e & 3) + 8 (e >> 2)
        const int j = t >> 6, i = (t >> 4) & 3, h = (t >> 3) & 1, e = t & 7;
        const int grp = (int)blockIdx.y + j * gstride;
        const int c = (grp * KS_CPW + i) * 16 + p16_channel(h, e);
        const bool in = grp < ngrp && c < C;
        coef[0][t] = in ? 32.f * wl[c * 3 + 0] : 0.f;
        coef[1][t] = in ? 32.f * wl[c * 3 + 1] : 0.f;
        coef[2][t] = in ? 32.f * wl[c * 3 + 2] : 0.f;
        coef[3][t] = in ? scale[c] : 0.f;
        coef[4][t] = in ? 32.f * shift[c] : 0.f;
    }
    const bool valid = r.x != -2, ok = r.x >= 0;
    const ks_f2 d0 = {__int_as_float(r.y), __int_as_float(r.y)}, d1 = {__int_as_float(r.z), __int_as_float(r.z)},
                d2 = {__int_as_float(r.w), __int_as_float(r.w)};
    RangeAcc xr = {0, 0u};
    const float lo = relu ? 0.f : -65504.f;                              // ReLU and the lower clamp of the split are one v_med3_f32
    // (plane p = (chunk, form, half) of a P16 tensor with L columns starts at p L)
    const uint4 *zbase = zp + (long long)hh * Lm + (ok ? r.x : 0);
    uint4 *hbase = h1 + (long long)hh * Lp + lcol;

    // the gathers of one group: 2 x 16 bytes per chunk and lane (a chunk past KC reads nothing)
    auto load = [&](uint4 (&zh)[KS_CPW], uint4 (&zm)[KS_CPW], int grp) {
        const uint4 *zq = zbase + (long long)(grp * KS_CPW) * 4 * Lm;
#pragma unroll
        for (int i = 0; i < KS_CPW; ++i) {
            zh[i] = make_uint4(0u, 0u, 0u, 0u);
            zm[i] = zh[i];
#ifdef SONET_VARIANTS
            if (abl & 2) { zh[i] = make_uint4(r.y, r.z, r.w, r.x); zm[i] = zh[i]; } else       // (ablation: no gathers)
#endif
            if (ok && grp * KS_CPW + i < KC) {
                zh[i] = zq[0];
                zm[i] = zq[2 * Lm];
            }
            zq += 4 * Lm;
        }
    };
    auto process = [&](const uint4 (&zhq)[KS_CPW], const uint4 (&zmq)[KS_CPW], int grp, int j) {
        uint4 *hq = hbase + (long long)(grp * KS_CPW) * 4 * Lp;
#pragma unroll
        for (int i = 0; i < KS_CPW; ++i) {
            if (grp * KS_CPW + i >= KC) break;
            const unsigned zh[4] = {zhq[i].x, zhq[i].y, zhq[i].z, zhq[i].w}, zm[4] = {zmq[i].x, zmq[i].y, zmq[i].z, zmq[i].w};
            unsigned hv[4], mv[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) {                                // elements 2 q, 2 q + 1: two consecutive channels
                const int slot = (j * KS_CPW + i) * 16 + hh * 8 + 2 * q;
                const ks_f2 w0 = *reinterpret_cast<const ks_f2 *>(&coef[0][slot]), w1 = *reinterpret_cast<const ks_f2 *>(&coef[1][slot]),
                            w2 = *reinterpret_cast<const ks_f2 *>(&coef[2][slot]), sc = *reinterpret_cast<const ks_f2 *>(&coef[3][slot]),
                            sh = *reinterpret_cast<const ks_f2 *>(&coef[4][slot]);
                const ks_h2 zh2 = __builtin_bit_cast(ks_h2, zh[q]), zm2 = __builtin_bit_cast(ks_h2, zm[q]);
                // 32 z = hi + residual (exact: two fp16 values whose exponents are at most 11 apart)
                ks_f2 a = __builtin_convertvector(zh2, ks_f2) + __builtin_convertvector(zm2, ks_f2);
                a = __builtin_elementwise_fma(w0, d0, a);                // (the reference's order: z, then the three coordinate channels)
                a = __builtin_elementwise_fma(w1, d1, a);
                a = __builtin_elementwise_fma(w2, d2, a);
                a = __builtin_elementwise_fma(a, sc, sh);                // 32 x the pre-activation
                if (!valid) a = ks_f2{0.f, 0.f};
                range_track(xr, a[0], a[1]);
                // (a NaN leaves v_med3_f32 as the lower bound, as in split_act; the range log has seen it)
                const ks_f2 X = {__builtin_amdgcn_fmed3f(a[0], lo, 65504.f), __builtin_amdgcn_fmed3f(a[1], lo, 65504.f)};
                const ks_h2 hp = __builtin_convertvector(X, ks_h2);
                const ks_f2 R = X - __builtin_convertvector(hp, ks_f2);
                hv[q] = __builtin_bit_cast(unsigned, hp);
                mv[q] = __builtin_bit_cast(unsigned, __builtin_convertvector(R, ks_h2));
            }
#ifdef SONET_VARIANTS
            if ((abl & 1) && hv[0] != 0x12345u) { hq += 4 * Lp; continue; }    // (ablation: no stores)
#endif
            hq[0] = make_uint4(hv[0], hv[1], hv[2], hv[3]);
            hq[2 * Lp] = make_uint4(mv[0], mv[1], mv[2], mv[3]);
            hq += 4 * Lp;
        }
    };

    uint4 zhA[KS_CPW], zmA[KS_CPW], zhB[KS_CPW], zmB[KS_CPW];
    int grp = (int)blockIdx.y, j = 0;
    load(zhA, zmA, grp);
    __syncthreads();                                                     // the coefficient table
    for (; grp < ngrp; grp += 2 * gstride, j += 2) {
        const bool more = grp + gstride < ngrp;
        if (more) load(zhB, zmB, grp + gstride);
        process(zhA, zmA, grp, j);
        if (more) {
            if (grp + 2 * gstride < ngrp) load(zhA, zmA, grp + 2 * gstride);
            process(zhB, zmB, grp + gstride, j + 1);
        }
    }
    if (rlog != nullptr) {
        // (32 x was tracked: take the factor out of the exponent; a negative value only matters without the ReLU)
        unsigned bits;
        if (relu) {
            const unsigned pos = xr.mp > 0 ? (unsigned)xr.mp : 0u, nan_neg = xr.mn > 0xFF800000u ? (xr.mn & 0x7FFFFFFFu) : 0u;
            bits = pos > nan_neg ? pos : nan_neg;
        } else {
            bits = range_amax_bits(xr);
        }
        unsigned wm = wave_umax(bits);
        wm = wm > (5u << 23) ? wm - (5u << 23) : 0u;
        if ((threadIdx.x & 63) == 0) wmax[threadIdx.x >> 6] = wm;
        __syncthreads();
        if (threadIdx.x == 0) {
            const unsigned a = wmax[0] > wmax[1] ? wmax[0] : wmax[1], c2 = wmax[2] > wmax[3] ? wmax[2] : wmax[3];
            const unsigned mx = a > c2 ? a : c2;
            if (mx > __atomic_load_n(rlog + 2, __ATOMIC_RELAXED)) atomicMax(rlog + 2, mx);
        }
    }
}

// f32 [G][C] (the group-max layer's output, e.g. the global feature [B][C]) <- nothing to do; P16 planes of a flat 1 x C x L activation
// -> f32 [B][C][M] (L = B M): the lazy decode of the stage's intermediate maps (knn_feature_1) when a caller reads them
__global__ __launch_bounds__(256) void p16_flat_to_bcm_kernel(const uint4 *__restrict__ p, float *__restrict__ x, int C, int M, long long BM, long long Lm, int KC)
{
    const long long l = (long long)blockIdx.x * 256 + threadIdx.x;
    const int kc = blockIdx.y >> 1, hh = blockIdx.y & 1;
    if (l >= BM) return;
    const uint4 hv = p[((long long)(kc * 2 + 0) * 2 + hh) * Lm + l], mv = p[((long long)(kc * 2 + 1) * 2 + hh) * Lm + l];
    typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
    const f16x8 h8 = __builtin_bit_cast(f16x8, hv), m8 = __builtin_bit_cast(f16x8, mv);
    const long long b = l / M;
    const int m = (int)(l - b * M);
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const int c = kc * 16 + p16_channel(hh, e);
        if (c < C) x[(b * C + c) * M + m] = ((float)h8[e] + (float)m8[e]) * 0.03125f;
    }
}

}  // namespace

extern "C" size_t sonet_knn_stage_columns(int B, int M, int K)
{
    if (B <= 0 || M <= 0 || K <= 0 || K > 128) return 0;
    return (size_t)(sonet::ceil_div64((long long)B * M, knn_stage_groups(K)) * 128);
}

extern "C" size_t sonet_node_stage_columns(int B, int M)
{
    if (B <= 0 || M <= 0) return 0;
    return (size_t)(sonet::ceil_div64((long long)B * M, 128) * 128);
}

extern "C" int sonet_knn_stage_prepare_f32(const float *coord, const int64_t *knn_I, int KI, int center_avg, int B, int M, int K,
                                           float *center, void *center_p16, void *rec, sonet_stream_t stream)
{
    const char *what = "sonet_knn_stage_prepare_f32";
    SONET_REQUIRE(coord && knn_I && center && center_p16 && rec, "%s: NULL pointer", what);
    SONET_REQUIRE(B > 0 && M > 0 && K >= 1 && K <= 128 && KI >= K, "%s: bad size B=%d M=%d K=%d (of %d)", what, B, M, K, KI);
    const long long Lp = (long long)sonet_knn_stage_columns(B, M, K), Lm = (long long)sonet_node_stage_columns(B, M);
    if (Lp / 256 + 1 > 0x7FFFFFFFll) return sonet::fail(SONET_ERR_UNSUPPORTED, "%s: too many columns", what);
    hipLaunchKernelGGL(knn_stage_prepare_kernel, dim3((unsigned)sonet::ceil_div64(Lp, 256)), dim3(256), 0, sonet::as_stream(stream),
                       coord, knn_I, KI, center_avg, B, M, K, knn_stage_groups(K), Lp, Lm, center, reinterpret_cast<uint4 *>(center_p16),
                       reinterpret_cast<int4 *>(rec), sonet::range_log());
    return sonet::launched(what);
}

extern "C" int sonet_knn_stage_input_p16(const void *rec, const void *z_p16, const float *wl, const float *scale, const float *shift, int relu,
                                         int B, int M, int K, int C, void *h1_p16, sonet_stream_t stream)
{
    const char *what = "sonet_knn_stage_input_p16";
    SONET_REQUIRE(rec && z_p16 && wl && scale && shift && h1_p16, "%s: NULL pointer", what);
    SONET_REQUIRE(B > 0 && M > 0 && C > 0 && K >= 1 && K <= 128, "%s: bad size B=%d M=%d C=%d K=%d", what, B, M, C, K);
    const long long Lp = (long long)sonet_knn_stage_columns(B, M, K), Lm = (long long)sonet_node_stage_columns(B, M);
    const long long nblk = Lp / 128;
    const int KC = sonet::ceil_div(C, 16);
    if (nblk > 0x7FFFFFFFll || (double)KC * 64.0 * (double)Lp >= 2.0e9) return sonet::fail(SONET_ERR_UNSUPPORTED, "%s: too many columns", what);
    // chunk groups per workgroup: as many as keep >= 2 workgroups per CU in the launch (each walks its groups with the next one's gathers in
    // flight), at most KS_MAXG
    const int ngrp = sonet::ceil_div(KC, KS_CPW);
    int gy = sonet::ceil_div(ngrp, KS_MAXG);
    while (gy < ngrp && nblk * gy < 512) ++gy;
#ifdef SONET_VARIANTS
    if (const char *e = sonet::knob("SONET_KSI_GY")) { const int v = atoi(e); if (v >= sonet::ceil_div(ngrp, KS_MAXG) && v <= ngrp) gy = v; }
#endif
    hipLaunchKernelGGL(knn_stage_input_kernel, dim3((unsigned)nblk, (unsigned)gy), dim3(256), 0, sonet::as_stream(stream),
                       reinterpret_cast<const int4 *>(rec), reinterpret_cast<const uint4 *>(z_p16), wl, scale, shift, relu, C, Lp, Lm,
                       reinterpret_cast<uint4 *>(h1_p16), sonet::range_log(), sonet::knob("SONET_KSI_ABL") ? atoi(sonet::knob("SONET_KSI_ABL")) : 0);
    return sonet::launched(what);
}

extern "C" int sonet_p16_flat_to_bcm_f32(const void *p16, float *x, int B, int C, int M, sonet_stream_t stream)
{
    const char *what = "sonet_p16_flat_to_bcm_f32";
    SONET_REQUIRE(p16 && x, "%s: NULL pointer", what);
    SONET_REQUIRE(B > 0 && C > 0 && M > 0, "%s: non-positive size", what);
    const int KC = sonet::ceil_div(C, 16);
    const long long BM = (long long)B * M;
    if (KC * 2 > 65535) return sonet::fail(SONET_ERR_UNSUPPORTED, "%s: C too large for one launch", what);
    hipLaunchKernelGGL(p16_flat_to_bcm_kernel, dim3((unsigned)sonet::ceil_div64(BM, 256), (unsigned)(KC * 2)), dim3(256), 0, sonet::as_stream(stream),
                       reinterpret_cast<const uint4 *>(p16), x, C, M, BM, (long long)sonet_node_stage_columns(B, M), KC);
    return sonet::launched(what);
}
