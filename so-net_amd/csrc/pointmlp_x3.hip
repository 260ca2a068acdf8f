// pointmlp_x3.hip -- the fused point-wise layer on bf16 MFMA with a 3-way bf16 split of BOTH operands.
//
// Why: on gfx950 the f32-input MFMA (pointmlp.hip) runs at the f32 vector rate (157 TFLOP/s) and shares the
// VALU pipe; the bf16 MFMA runs on the matrix cores at 16x that rate.  Splitting an f32 value into three
// bf16 terms  x = xh + xm + xl  (each the round-to-nearest bf16 of the remaining residual: 3 x 8 = 24
// significand bits) and keeping the six products of weight <= 2^-16
//      W.x  ~=  Wh.xh + Wh.xm + Wm.xh + Wh.xl + Wl.xh + Wm.xm        (f32 accumulate inside the MFMA)
// reproduces the f32 product to ~2^-23 relative (the dropped terms Wm.xl, Wl.xm, Wl.xl are <= 2^-24),
// i.e. f32-class accuracy at 6/16 of the f32-MFMA cost.  Against the reference's fixtures the whole
// classifier forward stays within 3e-6 * max(|ref|, rms) (tolerance 1e-5); a 2-way split does not (4e-5).
//
// Same data flow as the lean f32 kernel: W (pre-split, pre-packed in A-fragment order) goes through LDS in
// stages shared by the 4 waves; X rows are raw-buffer loads (scalar row offsets, hardware zero fill past
// the panel) prefetched one stage ahead as f32 and split in registers right before use:
//   v_mfma_f32_32x32x16_bf16:  A lane l: W[i = l&31][k = 8*(l>>5) .. +7]   (8 bf16 = 4 VGPRs)
//                              B lane l: X[k = 8*(l>>5) .. +7][j = l&31]   -> 8 dword loads per 16-channel chunk
//                              D as in pointmlp.hip (rows = output channels, columns = points).
// Split cost: 11 VALU per value pair (v_cvt_pk_bf16_f32, unpack, exact f32 subtract) = 44 per chunk per lane,
// on the VALU pipe, while the 6*MT MFMAs of the chunk run on the matrix pipe.
//
// F16 variant (sonet_pointmlp_h3_*): the same kernel on v_mfma_f32_32x32x16_f16 with a THREE-term split,
//      x = xh + xm,  xh = fp16(x):   32 W.x ~= Wh.(32 xh) + Wh.fp16(32 xm) + fp16(32 Wm).xh
// (dropped terms and the rounding of the scaled residuals <= 2^-22 relative; the factor 32 keeps the residuals out of the
// fp16 subnormals, rides in the accumulator -- power-of-two scalings are exact -- and leaves through scale / 32).  The
// first two terms share the weight operand Wh: two W slices per chunk and tile go through LDS for three MFMAs.  Same
// 3e-6 accuracy on the fixtures at half the MFMAs, but an fp16 operand RANGE: inputs are clamped to +-2047 (32 x must
// fit) and magnitudes below ~1e-4 lose relative precision -- fine for coordinates and normalised activations (forward),
// not for gradients: the dgrad launches of the backward keep the bf16 split.
#include "common.hpp"
#include <stdlib.h>

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2_t __attribute__((ext_vector_type(2)));
typedef float f32x2_t __attribute__((ext_vector_type(2)));
typedef int i32x4_t __attribute__((ext_vector_type(4)));

constexpr int X3_THREADS = 256;
constexpr int X3_WAVES = 4;

__device__ __forceinline__ unsigned cvt_pk_bf16(float lo, float hi) {
    unsigned r;
    asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(r) : "v"(lo), "v"(hi));
    return r;
}

// (x0, x1) -> three packed bf16 pairs (hi, mid, lo), round-to-nearest at each level, residuals exact in f32
__device__ __forceinline__ void split3_pair(float x0, float x1, unsigned &h, unsigned &m, unsigned &l) {
    h = cvt_pk_bf16(x0, x1);
    const float r0 = x0 - __uint_as_float(h << 16), r1 = x1 - __uint_as_float(h & 0xFFFF0000u);
    m = cvt_pk_bf16(r0, r1);
    const float q0 = r0 - __uint_as_float(m << 16), q1 = r1 - __uint_as_float(m & 0xFFFF0000u);
    l = cvt_pk_bf16(q0, q1);
}

// fp16 flavour, B side: (x0, x1) -> 32 xh = fp16(32 x), fp16(32 x - 32 xh), xh = 32 xh * 2^-5 (packed pairs; exact residual)
constexpr unsigned F16_2_M5_PK = 0x28002800u;
__device__ __forceinline__ unsigned cvt_pk_f16(float lo, float hi) {
    const f32x2_t v = {lo, hi};
    return __builtin_bit_cast(unsigned, __builtin_convertvector(v, f16x2_t));
}
__device__ __forceinline__ void split16_pair(float x0, float x1, unsigned &h, unsigned &m, unsigned &l) {
    x0 = 32.f * __builtin_fminf(__builtin_fmaxf(x0, -2047.f), 2047.f);
    x1 = 32.f * __builtin_fminf(__builtin_fmaxf(x1, -2047.f), 2047.f);
    h = cvt_pk_f16(x0, x1);
    const f16x2_t hv = __builtin_bit_cast(f16x2_t, h);
    m = cvt_pk_f16(x0 - (float)hv[0], x1 - (float)hv[1]);
    l = __builtin_bit_cast(unsigned, hv * __builtin_bit_cast(f16x2_t, F16_2_M5_PK));
}
// the same values with the clamp written as one v_med3_f32 (what hipcc makes of fmin(fmax()) after a canonicalising v_max x, x:
// a NaN input still leaves as -2047, the minimum of the three)
__device__ __forceinline__ void split16_pair_med3(float x0, float x1, unsigned &h, unsigned &m, unsigned &l) {
    x0 = 32.f * __builtin_amdgcn_fmed3f(x0, -2047.f, 2047.f);
    x1 = 32.f * __builtin_amdgcn_fmed3f(x1, -2047.f, 2047.f);
    h = cvt_pk_f16(x0, x1);
    const f16x2_t hv = __builtin_bit_cast(f16x2_t, h);
    m = cvt_pk_f16(x0 - (float)hv[0], x1 - (float)hv[1]);
    l = __builtin_bit_cast(unsigned, hv * __builtin_bit_cast(f16x2_t, F16_2_M5_PK));
}
// A side: (w0, w1) -> fp16(w), fp16(32 * (w - fp16(w)))
__device__ __forceinline__ void split16_w(float w0, float w1, unsigned &h, unsigned &res) {
    w0 = __builtin_fminf(__builtin_fmaxf(w0, -65504.f), 65504.f);
    w1 = __builtin_fminf(__builtin_fmaxf(w1, -65504.f), 65504.f);
    h = cvt_pk_f16(w0, w1);
    const f16x2_t hv = __builtin_bit_cast(f16x2_t, h);
    res = cvt_pk_f16(32.f * (w0 - (float)hv[0]), 32.f * (w1 - (float)hv[1]));
}

// bf16 flavour: Wp3[ct][kc][term][lane] (uint4 = 8 bf16):  W[ct*32 + (lane&31)][kc*16 + 8*(lane>>5) + t], t = 0..7, term = h, m, l.
// fp16 flavour: Wp2[ct][kcp][term][lane], term 0 = fp16(w) (meets 32 xh and the scaled x residual), term 1 = fp16(32 (w - h))
// (meets xh); kcp runs to KCP = KC rounded up to a multiple of H3_KPAD with ZERO chunks past KC: a (tile, chunk pair) is 4 KiB of
// consecutive bytes -- one LDS-DMA base with four instruction offsets -- and a K tail needs no special case.
constexpr int H3_KPAD = 8;
template <bool F16>
__device__ __forceinline__ void x3_pack_body(const float *__restrict__ W, uint4 *__restrict__ Wp3, int Cin, int Cout, int KC, long long t,
                                             unsigned *__restrict__ trailer, long long rs, long long cs)
{
    RangeAcc wr = {0, 0u};
    const int lane = (int)(t & 63);
    const long long r = t >> 6;
    const int kc = (int)(r % KC), ct = (int)(r / KC);
    const int o = ct * 32 + (lane & 31);
    const int c0 = kc * 16 + 8 * (lane >> 5);
    unsigned h[4], m[4], l[4];
#pragma unroll
    for (int p = 0; p < 4; ++p) {
        const int c = c0 + 2 * p;
        const float w0 = (o < Cout && c < Cin) ? W[(long long)o * rs + (long long)c * cs] : 0.f;
        const float w1 = (o < Cout && c + 1 < Cin) ? W[(long long)o * rs + (long long)(c + 1) * cs] : 0.f;
        range_track(wr, w0, w1);
        if constexpr (F16) {
            unsigned hh, res;
            split16_w(w0, w1, hh, res);
            h[p] = hh; m[p] = 0u; l[p] = res;
        } else {
            split3_pair(w0, w1, h[p], m[p], l[p]);
        }
    }
    if constexpr (F16) {
        uint4 *dst = Wp3 + (r * 2) * 64 + lane;
        dst[0] = make_uint4(h[0], h[1], h[2], h[3]);
        dst[64] = make_uint4(l[0], l[1], l[2], l[3]);
    } else {
        uint4 *dst = Wp3 + (r * 3) * 64 + lane;
        dst[0] = make_uint4(h[0], h[1], h[2], h[3]);
        dst[64] = make_uint4(m[0], m[1], m[2], m[3]);
        dst[128] = make_uint4(l[0], l[1], l[2], l[3]);
    }
    range_publish(trailer, wave_umax(range_amax_bits(wr)), lane);         // max |w| of the layer (range log, word 1 of a launch)
}

template <bool F16>
__global__ __launch_bounds__(256) void x3_pack_kernel(const float *__restrict__ W, uint4 *__restrict__ Wp3,
                                                       int Cin, int Cout, int KC /*fp16: KCP*/, long long total, unsigned *__restrict__ trailer,
                                                       long long rs /*element (o, c) = W[o * rs + c * cs]*/, long long cs)
{
    const long long t = (long long)blockIdx.x * 256 + threadIdx.x;       // (ct*KC + kc)*64 + lane
    if (t >= total) return;                                              // (total is a multiple of 64: whole waves leave)
    x3_pack_body<F16>(W, Wp3, Cin, Cout, KC, t, trailer, rs, cs);
}

// ---- every weight pack of a training step in ONE launch (sonet_pack_multi): a device table of the packs to refresh -- the layers' forward
// packs and the transposed packs of their dgrads, any flavour -- and one workgroup range per entry.  The optimizer changes every weight
// once per step; fifteen pack launches (+ fifteen 64-byte trailer memsets) per step were 0.1 ms of device time and as many host calls.
struct PackEntry {
    const float *W;                       // source: element (o, c) = W[o * rs + c * cs]
    void *Wp;                             // destination pack
    long long rs, cs, total;              // total = 64 x (32-row tiles) x (K chunks) work items
    int Cin, rows, KC, flavour;           // flavour 0 = bf16, 1 = x3 (three bf16 pieces), 2 = h3 (fp16 + residual)
    int blk0, nblk;                       // this entry's workgroups: blk0 .. blk0 + nblk - 1
    long long pad_;
};
static_assert(sizeof(PackEntry) == 72, "PackEntry layout is part of the C ABI (sonet_pack_multi)");

__device__ __forceinline__ unsigned *pack_trailer(const PackEntry &e) {
    return reinterpret_cast<unsigned *>(reinterpret_cast<uint4 *>(e.Wp) + e.total * (e.flavour == 2 ? 2 : 3));
}

__global__ __launch_bounds__(64) void pack_multi_zero_kernel(const PackEntry *__restrict__ tab, int n)
{
    const int e = blockIdx.x, w = threadIdx.x;
    if (e < n && tab[e].flavour != 0 && w < 16) pack_trailer(tab[e])[w] = 0u;
}

__global__ __launch_bounds__(256) void pack_multi_kernel(const PackEntry *__restrict__ tab, int n)
{
    __shared__ int which;
    if (threadIdx.x == 0) {
        int e = 0;
        while (e + 1 < n && (int)blockIdx.x >= tab[e + 1].blk0) ++e;
        which = e;
    }
    __syncthreads();
    const PackEntry e = tab[which];
    const long long t = (long long)((int)blockIdx.x - e.blk0) * 256 + threadIdx.x;
    if (t >= e.total) return;
    if (e.flavour == 1) x3_pack_body<false>(e.W, reinterpret_cast<uint4 *>(e.Wp), e.Cin, e.rows, e.KC, t, pack_trailer(e), e.rs, e.cs);
    else if (e.flavour == 2) x3_pack_body<true>(e.W, reinterpret_cast<uint4 *>(e.Wp), e.Cin, e.rows, e.KC, t, pack_trailer(e), e.rs, e.cs);
    else {
        // (the body of bf16_pack_kernel, pointmlp_bf16.hip)
        const int lane = (int)(t & 63);
        const long long r = t >> 6;
        const int kc = (int)(r % e.KC), ct = (int)(r / e.KC);
        const int o = ct * 32 + (lane & 31);
        const int c0 = kc * 16 + 8 * (lane >> 5);
        unsigned w[4];
#pragma unroll
        for (int p = 0; p < 4; ++p) {
            const int c = c0 + 2 * p;
            const float w0 = (o < e.rows && c < e.Cin) ? e.W[(long long)o * e.rs + (long long)c * e.cs] : 0.f;
            const float w1 = (o < e.rows && c + 1 < e.Cin) ? e.W[(long long)o * e.rs + (long long)(c + 1) * e.cs] : 0.f;
            asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(w[p]) : "v"(w0), "v"(w1));
        }
        reinterpret_cast<uint4 *>(e.Wp)[t] = make_uint4(w[0], w[1], w[2], w[3]);
    }
}

// mean / biased variance from the per-workgroup partial sums of the statistics epilogue: one workgroup per channel, thread i adds the
// partials i, i + 256, ... in order, the 256 sums meet in a fixed tree (deterministic).  (Four channels per workgroup and 64 lanes per
// channel took 40 us on the 7,500 partials of a point-level f32 layer.)  With a rider (common.hpp) the same thread goes on to the
// normalisation coefficients and the running-statistics update -- the arithmetic of bn_fwd_coeffs_kernel / bn_running_update_kernel
// (pointwise_bwd.hip), operation for operation.
__device__ __forceinline__ void bn_rider_apply(const sonet::BnRider &rd, int c, float mean, float var) {
    if (rd.gamma == nullptr) return;
    const float is = 1.0f / __fsqrt_rn(var + rd.eps);
    const float s_ = rd.gamma[c] * is;
    rd.invstd[c] = is;
    rd.sc[c] = s_;
    rd.sh[c] = rd.beta[c] - mean * s_;
    if (rd.rmean != nullptr) {
        const float m = rd.momentum;
        rd.rmean[c] = __fmaf_rn(mean, m, __fmul_rn(rd.rmean[c], 1.0f - m));
        rd.rvar[c] = __fmaf_rn(__fmul_rn(var, rd.unbias), m, __fmul_rn(rd.rvar[c], 1.0f - m));
    }
}

__global__ __launch_bounds__(256) void stats_partial_finalize_kernel(const double *__restrict__ partial, int nwg, int C, double inv_n,
                                                                     float *__restrict__ mean, float *__restrict__ var, const sonet::BnRider rd)
{
    __shared__ double r1[256], r2[256];
    const int c = blockIdx.x, t = threadIdx.x;
    double s1 = 0.0, s2 = 0.0;
    for (int k = t; k < nwg; k += 256) {
        const double *p = partial + ((size_t)k * C + c) * 2;
        s1 += p[0];
        s2 += p[1];
    }
    r1[t] = s1;
    r2[t] = s2;
    __syncthreads();
#pragma unroll
    for (int off = 128; off > 0; off >>= 1) {
        if (t < off) { r1[t] += r1[t + off]; r2[t] += r2[t + off]; }
        __syncthreads();
    }
    if (t == 0) {
        const double m = r1[0] * inv_n;
        double v = r2[0] * inv_n - m * m;
        if (v < 0.0) v = 0.0;
        mean[c] = (float)m;
        var[c] = (float)v;
        bn_rider_apply(rd, c, (float)m, (float)v);
    }
}

// partial (sum, sum) pairs of the workgroups -> sums[0 .. C) and sums[C .. 2 C) in double precision, fixed order: the layout
// sonet_bn_bwd_coeffs_f32 reads (BnbArgs::pstats)
__global__ __launch_bounds__(256) void bwd_sums_finalize_kernel(const double *__restrict__ partial, int nwg, int C, double *__restrict__ sums)
{
    __shared__ double r1[256], r2[256];
    const int c = blockIdx.x, t = threadIdx.x;
    double s1 = 0.0, s2 = 0.0;
    for (int k = t; k < nwg; k += 256) {
        const double *p = partial + ((size_t)k * C + c) * 2;
        s1 += p[0];
        s2 += p[1];
    }
    r1[t] = s1;
    r2[t] = s2;
    __syncthreads();
#pragma unroll
    for (int off = 128; off > 0; off >>= 1) {
        if (t < off) { r1[t] += r1[t + off]; r2[t] += r2[t + off]; }
        __syncthreads();
    }
    if (t == 0) { sums[c] = r1[0]; sums[C + c] = r2[0]; }
}

// SEGPOOL (training forward of the first PointNet's last layer when only the pooled map is consumed, models/layers.py:431 +
// models/networks.py:180-185): the columns are NODE-SORTED (som_sort_group), the output is never stored; the per-node arg-max of every
// channel leaves through 64-bit keys (orderable(value) << 32 | 0xFFFFFFFF - column: the order of index_max.hip -- bigger value wins,
// equal values: the smaller column, NaN never, -0 counts as +0) combined by global atomicMax, keys[B][Cout][M] preset to "-1000 at
// column 0".  The MFMA operands are swapped (D^T = X^T W^T: the same products in the same order), so a lane holds ONE channel for 16
// of the wave's 32 points and the maximum over a node's points is a chain over registers; the two half-waves meet in one exchange.
struct SegPoolArgs {
    const int32_t *ids;                   // [B][L] node of every (sorted) column
    const int32_t *pos0;                  // [B] sorted position of original column 0 (what a bin nothing beat gathers)
    unsigned long long *keys;             // [B][Cout][M]
    float *v0;                            // [B][Cout] the layer's value at column pos0[b]
    int M;
};
constexpr unsigned long long SP_INIT_KEY = (0x3B85FFFFull << 32) | 0xFFFFFFFFull;   // ord(-1000.0f), column 0 (index_max.hip)
__device__ __forceinline__ unsigned sp_ord_f32(unsigned bits) {   // total order; -0 == +0; NaN -> 0 (never wins)
    if (bits == 0x80000000u) bits = 0u;
    const unsigned o = bits ^ ((unsigned)((int)bits >> 31) | 0x80000000u);
    return (bits & 0x7FFFFFFFu) > 0x7F800000u ? 0u : o;
}

// XAFF (f32-class training forward): the inputs are the RAW outputs of BatchNorm layers whose normalise + ReLU pass was never run;
// the operand load applies it -- x = act(raw * scale[c] + shift[c]) per input channel, the arithmetic of sonet_channel_affine_act_f32 bit
// for bit -- so the normalised activations never exist in memory (models/layers.py:60-70, :282-296 between two layers).
struct XAffArgs {
    const float *s1, *h1;                 // [C1] (scale, shift) of x1's channels
    const float *s2, *h2;                 // [C2] of x2's
    int relu;                             // bit 0: ReLU on x1's channels, bit 1: on x2's
};

// BNB (f32-class training backward, bf16-split arithmetic): the input gradient of a layer whose OUTPUT went through a training-mode BatchNorm
// + ReLU.  x1 is gy (the gradient of the activation), bnb.raw the layer's raw output; the operand load applies the BatchNorm / ReLU backward
//      g_raw = a[k] * (relu && !(raw * sc[k] + sh[k] > 0) ? 0 : gy) + b[k] * raw + c0[k]
// -- sonet_pointwise_bwd_apply_f32's arithmetic, bit for bit -- and multiplies by W^T; the workgroups of the first output slab also write g_raw
// (the weight gradient's operand).  The separate pass over (gy, raw) and one read of g_raw are gone (models/layers.py:60-70 backward).
struct BnbArgs {
    const float *raw;                     // [B][C1][L]
    const float *a, *b, *c0, *sc, *sh;    // [C1]
    float *g_out;                         // [B][C1][L] (may be NULL)
    int relu;
    // (optional) the OUTPUT of this launch is gy of the layer below; when that layer's raw output and normalisation are at hand (the training
    // forward normalises on load: they are this layer's saved input) the epilogue also computes ITS BatchNorm-backward sums -- per channel
    // sum of gy * mask and of gy * mask * praw, what sonet_pointwise_bwd_stats_f32 would read (gy, praw) once more for
    const float *praw;                    // [B][Cout][L] or NULL
    const float *psc, *psh;               // [Cout]
    int prelu;
    double *pstats;                       // [gridDim.x][Cout][2] partial sums
    // (optional) another gradient of the same tensor, already computed (the layer's input has a second consumer whose backward ran
    // earlier): y = this launch's product + yadd -- autograd's accumulation (one more pass over both, a third tensor) done by the store
    const float *yadd;                    // [B][Cout][L] or NULL
};

// ZADD: the per-node addend form with its gathers issued BEFORE the K loop (64 registers; instantiated for MT = 4 only).
template <int MT, int S, bool F16, bool ZADD = false, bool SEGPOOL = false, bool XAFF = false, bool BNB = false>
__global__ __launch_bounds__(X3_THREADS) void pointmlp_x3_kernel(
    const float *__restrict__ x1, int C1, const float *__restrict__ x2, int C2, const uint4 *__restrict__ Wp3,
    const float *__restrict__ scale, const float *__restrict__ shift, int relu, float *__restrict__ y,
    int Cout, int L, int gpc, long long ngroups, int CT, int KC, int ct_per_y,
    const int32_t *__restrict__ gidx /*optional [B][L]: column l of x1 is x1[:, gidx[b][l]]*/, int L1 /*row length of x1*/,
    unsigned *__restrict__ rlog /*optional (fp16 flavour): range-log slot, word 0 = max |x| bits, word 1 = max |w| bits*/,
    int KCP /*chunks per cout tile in the pack (fp16 flavour: KC rounded up to H3_KPAD; bf16: KC)*/,
    double *__restrict__ stats_partial /*optional [gridDim.x][Cout][2]: sum and sum of squares of the stored output over this workgroup's columns*/,
    const float *__restrict__ zadd /*optional [B][Cout][ZM]: y = act((W x + zadd[b][o][zidx[b][l]]) * scale + shift)*/,
    const int32_t *__restrict__ zidx /*[B][L], out of range: + 0*/, int ZM, const SegPoolArgs sp, const XAffArgs xa, const BnbArgs bnb)
{
    static_assert(!BNB || (!F16 && !ZADD && !SEGPOOL && !XAFF && S == 1), "the BatchNorm-backward operand exists in the bf16-split layer only");
    constexpr int BW = BNB ? 16 : 8;                          // registers per chunk and lane: 8 rows of the input (+ 8 of raw)
    __shared__ __attribute__((aligned(16))) float4 bnb_t[BNB ? 512 : 1];        // (BNB) per input channel (a, b, c0, sc)
    __shared__ float bnb_h[BNB ? 512 : 1];                                       //       ... and sh
    static_assert(!SEGPOOL || (F16 && !ZADD), "the pooled form exists in the fp16-split arithmetic only");
    static_assert(!XAFF || (F16 && !ZADD), "normalise-on-load exists in the fp16-split arithmetic only");
    __shared__ __attribute__((aligned(16))) float2 xaff_t[XAFF ? 1024 : 1];     // (XAFF) per input channel (scale, shift), x1's first; past Cin: (0, 0)
    constexpr int NTW = F16 ? 2 : 3;                          // W slices per (chunk, tile): fp16 terms h and m share one
    constexpr int NSL = S * MT * NTW;                         // 1 KiB W slices per stage
    constexpr int NS = (NSL + X3_WAVES - 1) / X3_WAVES;
    __shared__ uint4 wsm[2][NS * X3_WAVES][64];
    __shared__ float2 affine[1024];
    __shared__ float2 red[X3_WAVES][MT * 32];                   // (statistics epilogue) per wave: (sum, sum of squares) of a row over its 32 columns

    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int j = lane & 31, h = lane >> 5;

    long long q = (long long)blockIdx.x * X3_WAVES + wave;
    const bool wave_valid = q < ngroups;
    q = wave_valid ? q : 0;
    const long long b = q / gpc;
    const int l0 = (int)(q - b * gpc) * 32;
    const bool pv = wave_valid && (l0 + j < L);
    const int lc = (l0 + j < L) ? l0 + j : l0;

    const unsigned rowB = (unsigned)L * 4u, rowB1 = (unsigned)L1 * 4u;
    const __amdgpu_buffer_rsrc_t r1 = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float *>(x1 + b * (long long)C1 * L1), 0, (int)((unsigned)C1 * rowB1), 0x00020000);
    const __amdgpu_buffer_rsrc_t r2 = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float *>(x2 ? x2 + b * (long long)C2 * L : x1), 0, (int)((unsigned)(x2 ? C2 : 0) * rowB), 0x00020000);
    const __amdgpu_buffer_rsrc_t ry = __builtin_amdgcn_make_buffer_rsrc(
        y + b * (long long)Cout * L, 0, (int)((unsigned)Cout * rowB), 0x00020000);
    const __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<uint4 *>(Wp3), 0, (int)((unsigned)CT * (unsigned)KCP * (unsigned)(NTW * 1024)), 0x00020000);
    const unsigned vox = (unsigned)(8 * h * L + lc) * 4u;      // lane byte offset inside a 16-channel chunk
    // x1 through a gather index (the neighbour gather of KNNModule, models/layers.py:313-350, done by the operand load):
    // an index outside [0, L1) reads zeros (lane offset past the panel: the descriptor's bounds check)
    unsigned vox1 = vox;
    if (gidx) {
        const int src = gidx[b * L + lc];
        vox1 = (unsigned)src < (unsigned)L1 ? (unsigned)(8 * h * L1 + src) * 4u : 0x7FFFFF00u;
    }
    const unsigned voy = (unsigned)(4 * h * L + lc) * 4u;
    const unsigned vow = (unsigned)lane * 16u;

    const int KC1 = C2 > 0 ? (C1 >> 4) : KC;                  // chunks fed by x1 (C1 % 16 == 0 when x2 exists)
    const int nstage = (KC + S - 1) / S;
    // (SEGPOOL) node of this lane's column (both half-waves hold the wave's 32 columns), position of original column 0 relative to the wave
    int sp_nid = -1, sp_p0rel = -1;
    if constexpr (SEGPOOL) {
        if (pv) sp_nid = sp.ids[b * L + l0 + j];
        sp_p0rel = wave_valid ? sp.pos0[b] - l0 : -1;
    }

    const __amdgpu_buffer_rsrc_t rbn = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float *>(BNB ? bnb.raw + b * (long long)C1 * L : x1), 0, (int)((unsigned)(BNB ? C1 : 0) * rowB), 0x00020000);
    const __amdgpu_buffer_rsrc_t rgo = __builtin_amdgcn_make_buffer_rsrc(
        (BNB && bnb.g_out) ? bnb.g_out + b * (long long)C1 * L : y, 0, (int)((unsigned)((BNB && bnb.g_out) ? C1 : 0) * rowB), 0x00020000);
    auto load_b = [&](float (&raw)[S][BW], int st) {
#pragma unroll
        for (int i = 0; i < S; ++i) {
            const int kc = st * S + i;
            const bool second = kc >= KC1;
            const unsigned rb = second ? rowB : rowB1;
            const unsigned row0 = (unsigned)(16 * (second ? kc - KC1 : kc)) * rb;
#pragma unroll
            for (int t = 0; t < 8; ++t) {
                const unsigned so = row0 + (unsigned)t * rb;
                raw[i][t] = __builtin_bit_cast(float, second ? __builtin_amdgcn_raw_buffer_load_b32(r2, vox, so, 0)
                                                              : __builtin_amdgcn_raw_buffer_load_b32(r1, vox1, so, 0));
                if constexpr (BNB) raw[i][8 + t] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rbn, vox, so, 0));
            }
        }
    };

    const int ct_begin = blockIdx.y * ct_per_y;
    const int ct_end = min(CT, ct_begin + ct_per_y);
    for (int o = ct_begin * 32 + (int)threadIdx.x; o < ct_end * 32; o += X3_THREADS)
        affine[o - ct_begin * 32] = make_float2(F16 ? scale[o] * 0.03125f : scale[o], shift[o]);   // fp16: accumulators hold 32 W.x
    if constexpr (BNB) {                                        // (published by the barrier in front of the first stage; past C1: zeros)
        for (int c = threadIdx.x; c < KC * 16; c += X3_THREADS) {
            const bool ok = c < C1;
            bnb_t[c] = ok ? make_float4(bnb.a[c], bnb.b[c], bnb.c0[c], bnb.sc[c]) : make_float4(0.f, 0.f, 0.f, 0.f);
            bnb_h[c] = ok ? bnb.sh[c] : 0.f;
        }
    }
    if constexpr (XAFF) {                                       // (published by the barrier in front of the first stage)
        for (int c = threadIdx.x; c < KC * 16; c += X3_THREADS)
            xaff_t[c] = c < C1 ? make_float2(xa.s1[c], xa.h1[c]) : (c - C1 < C2 ? make_float2(xa.s2[c - C1], xa.h2[c - C1]) : make_float2(0.f, 0.f));
    }

    RangeAcc xr = {0, 0u};
    for (int ct0 = ct_begin; ct0 < ct_end; ct0 += MT) {
        const bool track = F16 && rlog != nullptr && ct0 == 0;
        f32x16 acc[MT];
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[mt][r] = 0.f;
        // (ZADD) this group's 16 * MT node values per lane are requested now and used in the epilogue: gathered there they cost the
        // segmenter's first layer 0.28 ms of exposed L2 latency (0.85 vs 0.57 ms for the plain layer)
        float zreg[ZADD ? MT : 1][16];
        if constexpr (ZADD) {
            const int zm = zidx[b * L + lc];
            const bool zok = pv && (unsigned)zm < (unsigned)ZM;
            const float *zb = zadd + ((size_t)b * Cout) * ZM + (zok ? zm : 0);
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    zreg[mt][r] = zok ? zb[(size_t)((ct0 + mt) * 32 + (r & 3) + 8 * (r >> 2) + 4 * h) * ZM] : 0.f;
        }

        // slice sl of a stage: chunk i = sl / (NTW*MT), cout tile mt = (sl / NTW) % MT, term = sl % NTW
        auto stage_load = [&](i32x4_t (&w)[NS], int st) {
#pragma unroll
            for (int t = 0; t < NS; ++t) {
                int sl = wave + t * X3_WAVES;
                sl = sl < NSL ? sl : NSL - 1;
                const int i = sl / (NTW * MT), rem = sl - i * (NTW * MT);
                const int mt = rem / NTW, term = rem - mt * NTW;
                int kc = st * S + i;
                kc = kc < KC ? kc : KC - 1;
                w[t] = __builtin_amdgcn_raw_buffer_load_b128(rw, vow, (unsigned)(((ct0 + mt) * KCP + kc) * NTW + term) * 1024u, 0);
            }
        };
        auto stage_write = [&](const i32x4_t (&w)[NS], int slot) {
#pragma unroll
            for (int t = 0; t < NS; ++t)
                wsm[slot][wave + t * X3_WAVES][lane] = __builtin_bit_cast(uint4, w[t]);
        };
        auto compute = [&](const float (&raw_in)[S][BW], int slot, int st) {
#pragma unroll
            for (int i = 0; i < S; ++i) {
                unsigned bh[4], bm[4], bl[4];
                if constexpr (F16) {
                    float raw[S][8];
                    if constexpr (XAFF) {
                        // this lane's 8 channels of the chunk: 16 kc + 8 h + t (x1's channels first: C1 % 16 == 0 with a second input)
                        const int kc = st * S + i;
                        const float4 *tp = reinterpret_cast<const float4 *>(&xaff_t[kc * 16 + 8 * h]);
                        const bool rl = ((kc >= KC1 ? xa.relu >> 1 : xa.relu) & 1) != 0;
#pragma unroll
                        for (int p = 0; p < 4; ++p) {
                            const float4 c = tp[p];
                            float v0 = __fmaf_rn(raw_in[i][2 * p], c.x, c.y), v1 = __fmaf_rn(raw_in[i][2 * p + 1], c.z, c.w);
                            if (rl) { v0 = (v0 < 0.f) ? 0.f : v0; v1 = (v1 < 0.f) ? 0.f : v1; }
                            raw[i][2 * p] = v0; raw[i][2 * p + 1] = v1;
                        }
                    } else {
#pragma unroll
                        for (int t = 0; t < 8; ++t) raw[i][t] = raw_in[i][t];
                    }
                    if (track) {                                         // wave-uniform: first output-tile group of slab 0 only
#pragma unroll
                        for (int p = 0; p < 4; ++p) range_track(xr, raw[i][2 * p], raw[i][2 * p + 1]);
                    }
#pragma unroll
                    for (int p = 0; p < 4; ++p) split16_pair(raw[i][2 * p], raw[i][2 * p + 1], bh[p], bm[p], bl[p]);
                    const f16x8 Bh = __builtin_bit_cast(f16x8, make_uint4(bh[0], bh[1], bh[2], bh[3]));     // 32 xh
                    const f16x8 Bm = __builtin_bit_cast(f16x8, make_uint4(bm[0], bm[1], bm[2], bm[3]));     // 32 * residual
                    const f16x8 Bl = __builtin_bit_cast(f16x8, make_uint4(bl[0], bl[1], bl[2], bl[3]));     // xh
                    f16x8 Ah[MT];
                    if constexpr (SEGPOOL) {
                        // D^T = X^T W^T: the fragment registers are the same (A lane l: row l & 31, k = 8 (l >> 5) ..; B lane l: column
                        // l & 31, same k), every accumulator sees the same three products in the same order
#pragma unroll
                        for (int mt = 0; mt < MT; ++mt) acc[mt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(
                            Bl, __builtin_bit_cast(f16x8, wsm[slot][(i * MT + mt) * 2 + 1][lane]), acc[mt], 0, 0, 0);
#pragma unroll
                        for (int mt = 0; mt < MT; ++mt) {
                            Ah[mt] = __builtin_bit_cast(f16x8, wsm[slot][(i * MT + mt) * 2 + 0][lane]);
                            acc[mt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(Bm, Ah[mt], acc[mt], 0, 0, 0);
                        }
#pragma unroll
                        for (int mt = 0; mt < MT; ++mt) acc[mt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(Bh, Ah[mt], acc[mt], 0, 0, 0);
                    } else {
#pragma unroll
                    for (int mt = 0; mt < MT; ++mt) acc[mt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(
                        __builtin_bit_cast(f16x8, wsm[slot][(i * MT + mt) * 2 + 1][lane]), Bl, acc[mt], 0, 0, 0);
#pragma unroll
                    for (int mt = 0; mt < MT; ++mt) {
                        Ah[mt] = __builtin_bit_cast(f16x8, wsm[slot][(i * MT + mt) * 2 + 0][lane]);
                        acc[mt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(Ah[mt], Bm, acc[mt], 0, 0, 0);
                    }
#pragma unroll
                    for (int mt = 0; mt < MT; ++mt) acc[mt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(Ah[mt], Bh, acc[mt], 0, 0, 0);
                    }
                } else {
                float gv[8];
                if constexpr (BNB) {
                    // this lane's 8 channels of the chunk: 16 kc + 8 h + t
                    const int kc = st * S + i;
                    const int k0 = kc * 16 + 8 * h;
#pragma unroll
                    for (int t = 0; t < 8; ++t) {
                        const float4 c = bnb_t[k0 + t];
                        const float rv = raw_in[i][8 + t];
                        float gg = raw_in[i][t];
                        if (bnb.relu && !(__fmaf_rn(rv, c.w, bnb_h[k0 + t]) > 0.f)) gg = 0.f;
                        gv[t] = __fmaf_rn(c.x, gg, __fmaf_rn(c.y, rv, c.z));
                    }
                    if (bnb.g_out != nullptr && ct0 == 0 && pv) {          // (the first output slab's first tile group: every (channel, column) once)
#pragma unroll
                        for (int t = 0; t < 8; ++t)
                            __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(int, gv[t]), rgo, vox, (unsigned)(kc * 16 + t) * rowB, 0);
                    }
                } else {
#pragma unroll
                    for (int t = 0; t < 8; ++t) gv[t] = raw_in[i][t];
                }
#pragma unroll
                for (int p = 0; p < 4; ++p) split3_pair(gv[2 * p], gv[2 * p + 1], bh[p], bm[p], bl[p]);
                const bf16x8 Bh = __builtin_bit_cast(bf16x8, make_uint4(bh[0], bh[1], bh[2], bh[3]));
                const bf16x8 Bm = __builtin_bit_cast(bf16x8, make_uint4(bm[0], bm[1], bm[2], bm[3]));
                const bf16x8 Bl = __builtin_bit_cast(bf16x8, make_uint4(bl[0], bl[1], bl[2], bl[3]));
#pragma unroll
                for (int mt = 0; mt < MT; ++mt) {
                    const bf16x8 Ah = __builtin_bit_cast(bf16x8, wsm[slot][(i * MT + mt) * 3 + 0][lane]);
                    const bf16x8 Am = __builtin_bit_cast(bf16x8, wsm[slot][(i * MT + mt) * 3 + 1][lane]);
                    const bf16x8 Al = __builtin_bit_cast(bf16x8, wsm[slot][(i * MT + mt) * 3 + 2][lane]);
                    // smallest terms first
                    acc[mt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(Al, Bh, acc[mt], 0, 0, 0);
                    acc[mt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(Ah, Bl, acc[mt], 0, 0, 0);
                    acc[mt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(Am, Bm, acc[mt], 0, 0, 0);
                    acc[mt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(Am, Bh, acc[mt], 0, 0, 0);
                    acc[mt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(Ah, Bm, acc[mt], 0, 0, 0);
                    acc[mt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(Ah, Bh, acc[mt], 0, 0, 0);
                }
                }
            }
        };

        i32x4_t wreg[NS];
        float b0[S][BW], b1[S][BW];
        __syncthreads();
        stage_load(wreg, 0);
        load_b(b0, 0);
        stage_write(wreg, 0);
        stage_load(wreg, nstage > 1 ? 1 : 0);
#define X3_STAGE(st, bcur, bnxt, slot)                                        \
        {                                                                    \
            __syncthreads();                                                 \
            stage_write(wreg, (slot) ^ 1);                                   \
            stage_load(wreg, (st) + 2 < nstage ? (st) + 2 : nstage - 1);     \
            load_b(bnxt, (st) + 1);                                          \
            compute(bcur, slot, st);                                         \
        }
        int st = 0;
        for (; st + 2 <= nstage; st += 2) {
            X3_STAGE(st, b0, b1, 0)
            X3_STAGE(st + 1, b1, b0, 1)
        }
        if (st < nstage) X3_STAGE(st, b0, b1, 0)
#undef X3_STAGE

        if constexpr (SEGPOOL) {
            // acc[mt][r] = Y[point prow(r) = (r & 3) + 8 (r >> 2) + 4 h of the wave's 32][channel 32 (ct0 + mt) + j]
            if ((unsigned)sp_p0rel < 32u) {                      // wave-uniform: the group that holds original column 0
#pragma unroll
                for (int mt = 0; mt < MT; ++mt) {
                    const float2 ss = affine[(ct0 + mt - ct_begin) * 32 + j];
#pragma unroll
                    for (int r = 0; r < 16; ++r)
                        if ((r & 3) + 8 * (r >> 2) + 4 * h == sp_p0rel) {
                            float v = __fmaf_rn(acc[mt][r], ss.x, ss.y);
                            if (relu) v = (v < 0.f) ? 0.f : v;
                            sp.v0[(size_t)b * Cout + (ct0 + mt) * 32 + j] = v;
                        }
                }
            }
            unsigned remaining = (unsigned)__ballot(pv);         // lanes 0..31 <-> the wave's 32 columns
            while (remaining != 0u) {                            // one turn per node present (ids are sorted: a node's points are the rows [s0, s0 + nrows))
                const int s0 = __builtin_ctz(remaining);
                const int node = __builtin_amdgcn_readlane(sp_nid, s0);
                const unsigned segmask = (unsigned)__ballot(pv && sp_nid == node);
                remaining &= ~segmask;
                if ((unsigned)node >= (unsigned)sp.M) continue;  // an id outside [0, M): nobody's column (index_max.hip ignores it too)
                const unsigned nrows = (unsigned)__builtin_popcount(segmask);
                const int trel = 4 * h - s0;
#pragma unroll
                for (int mt = 0; mt < MT; ++mt) {
                    const float2 ss = affine[(ct0 + mt - ct_begin) * 32 + j];
                    float m = -__builtin_inff();
                    int p = 0;
#pragma unroll
                    for (int r = 0; r < 16; ++r) {               // ascending point order, strict '>': the first of equal values stays
                        const int orow = (r & 3) + 8 * (r >> 2);
                        float v = __fmaf_rn(acc[mt][r], ss.x, ss.y);
                        if (relu) v = (v < 0.f) ? 0.f : v;
                        const bool take = (unsigned)(trel + orow) < nrows && v > m;      // (a NaN fails the compare: it never wins)
                        m = take ? v : m;
                        p = take ? orow + 4 * h : p;
                    }
                    const float mo = __shfl_xor(m, 32, 64);      // the other half-wave's 16 points
                    const int po = __shfl_xor(p, 32, 64);
                    if (mo > m || (mo == m && po < p)) { m = mo; p = po; }
                    if (h == 0 && m > -__builtin_inff()) {
                        const unsigned long long key = ((unsigned long long)sp_ord_f32(__float_as_uint(m)) << 32) |
                                                       (unsigned long long)(0xFFFFFFFFu - (unsigned)(l0 + p));
                        atomicMax(sp.keys + ((size_t)b * Cout + (ct0 + mt) * 32 + j) * sp.M + node, key);
                    }
                }
            }
#ifdef SONET_VARIANTS
        // (variants build only: measured slower than the separate statistics pass, docs/findings.md R5.9 -- kept as a tested record)
        } else if (BNB && bnb.pstats != nullptr) {
            // the output is gy of the layer below: its BatchNorm-backward sums from here (same reduction as the forward statistics below)
            const unsigned voy_s = pv ? voy : 0x7FFFFF00u;
            const __amdgpu_buffer_rsrc_t rpr = __builtin_amdgcn_make_buffer_rsrc(
                const_cast<float *>(bnb.praw + b * (long long)Cout * L), 0, (int)((unsigned)Cout * rowB), 0x00020000);
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) {
                const unsigned so_tile = (unsigned)((ct0 + mt) * 32) * rowB;
                const float2 *aff = affine + (ct0 + mt - ct_begin) * 32 + 4 * h;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int orow = (r & 3) + 8 * (r >> 2);
                    const float2 ss = aff[orow];
                    float v = __fmaf_rn(acc[mt][r], ss.x, ss.y);
                    if (relu) v = (v < 0.f) ? 0.f : v;
                    __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(int, v), ry, voy_s, so_tile + (unsigned)orow * rowB, 0);
                    const int ch = (ct0 + mt) * 32 + orow + 4 * h;
                    // (requested here: 16 MT exposed round trips per lane.  Requested before the K loop -- 64 more registers, one workgroup
                    //  less per CU for every launch of this instantiation -- the launch was no faster: docs/findings.md R5.9)
                    const float pr = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rpr, voy_s, so_tile + (unsigned)orow * rowB, 0));
                    float gm = pv ? v : 0.f;
                    if (bnb.prelu && !(__fmaf_rn(pr, bnb.psc[ch], bnb.psh[ch]) > 0.f)) gm = 0.f;
                    const float s1 = row32_sum(gm), s2 = row32_sum(gm * pr);
                    if (j == 0) red[wave][mt * 32 + orow + 4 * h] = make_float2(s1, s2);
                }
            }
            __syncthreads();
            for (int t = threadIdx.x; t < MT * 32; t += X3_THREADS) {
                const double a_ = ((double)red[0][t].x + (double)red[1][t].x) + ((double)red[2][t].x + (double)red[3][t].x);
                const double q_ = ((double)red[0][t].y + (double)red[1][t].y) + ((double)red[2][t].y + (double)red[3][t].y);
                double *dst = bnb.pstats + ((size_t)blockIdx.x * Cout + (size_t)ct0 * 32 + t) * 2;
                dst[0] = a_;
                dst[1] = q_;
            }
            __syncthreads();                                    // (red is reused by the next tile group)
#endif
        } else if (stats_partial != nullptr) {
            // Training forward: BatchNorm's batch statistics (models/layers.py:60-70) of the output come out of this epilogue instead of
            // a second pass over the tensor.  A row's 32 columns sit in the 32 lanes of a half wave: four DPP adds + one swizzle per
            // quantity, all lanes active (a padded column stores to an out-of-range offset -- dropped by the descriptor -- and adds 0).
            const unsigned voy_s = pv ? voy : 0x7FFFFF00u;
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) {
                const unsigned so_tile = (unsigned)((ct0 + mt) * 32) * rowB;
                const float2 *aff = affine + (ct0 + mt - ct_begin) * 32 + 4 * h;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int orow = (r & 3) + 8 * (r >> 2);
                    const float2 ss = aff[orow];
                    float v = __fmaf_rn(acc[mt][r], ss.x, ss.y);
                    if (relu) v = (v < 0.f) ? 0.f : v;
                    __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(int, v), ry, voy_s, so_tile + (unsigned)orow * rowB, 0);
                    const float sv = pv ? v : 0.f;
                    const float s1 = row32_sum(sv), s2 = row32_sum(sv * sv);
                    if (j == 0) red[wave][mt * 32 + orow + 4 * h] = make_float2(s1, s2);
                }
            }
            __syncthreads();
            for (int t = threadIdx.x; t < MT * 32; t += X3_THREADS) {
                const double a = ((double)red[0][t].x + (double)red[1][t].x) + ((double)red[2][t].x + (double)red[3][t].x);
                const double q = ((double)red[0][t].y + (double)red[1][t].y) + ((double)red[2][t].y + (double)red[3][t].y);
                double *dst = stats_partial + ((size_t)blockIdx.x * Cout + (size_t)ct0 * 32 + t) * 2;
                dst[0] = a;
                dst[1] = q;
            }
            __syncthreads();                                    // (red is reused by the next tile group)
        } else if (pv && zadd != nullptr) {
            // a per-node addend gathered in the epilogue: the layer's input concatenates per-column channels (the GEMM above) with
            // channels that are constant per node -- their block of W . x is computed once per node by another launch (z) and added
            // here, before the folded BatchNorm and the ReLU (segmenter layer 1, models/networks.py:296-326)
            const int zm = zidx[b * L + lc];
            const bool zok = (unsigned)zm < (unsigned)ZM;
            const float *zb = zadd + ((size_t)b * Cout) * ZM + (zok ? zm : 0);
            const float unscale = F16 ? 32.f : 1.f;                // (fp16 flavour: the affine table holds scale / 32)
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) {
                const unsigned so_tile = (unsigned)((ct0 + mt) * 32) * rowB;
                const float2 *aff = affine + (ct0 + mt - ct_begin) * 32 + 4 * h;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int orow = (r & 3) + 8 * (r >> 2);
                    const float2 ss = aff[orow];
                    float zv;
                    if constexpr (ZADD) zv = zreg[mt][r];
                    else zv = zok ? zb[(size_t)((ct0 + mt) * 32 + orow + 4 * h) * ZM] : 0.f;
                    float v = __fmaf_rn(acc[mt][r], ss.x, __fmaf_rn(zv, ss.x * unscale, ss.y));
                    if (relu) v = (v < 0.f) ? 0.f : v;
                    __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(int, v), ry, voy, so_tile + (unsigned)orow * rowB, 0);
                }
            }
        } else if (pv) {
            bool addy = false;
            __amdgpu_buffer_rsrc_t rya = ry;
            if constexpr (BNB) {
                addy = bnb.yadd != nullptr;
                if (addy) rya = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(bnb.yadd + b * (long long)Cout * L), 0, (int)((unsigned)Cout * rowB), 0x00020000);
            }
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) {
                const unsigned so_tile = (unsigned)((ct0 + mt) * 32) * rowB;
                const float2 *aff = affine + (ct0 + mt - ct_begin) * 32 + 4 * h;
#pragma unroll
                for (int r0 = 0; r0 < 16; r0 += 8) {                    // (the addend eight rows at a time: registers)
                    float ya[8];
                    if constexpr (BNB) {
                        if (addy) {
#pragma unroll
                            for (int r = 0; r < 8; ++r)
                                ya[r] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(
                                    rya, voy, so_tile + (unsigned)(((r0 + r) & 3) + 8 * ((r0 + r) >> 2)) * rowB, 0));
                        }
                    }
#pragma unroll
                    for (int r = r0; r < r0 + 8; ++r) {
                        const int orow = (r & 3) + 8 * (r >> 2);
                        const float2 ss = aff[orow];
                        float v = __fmaf_rn(acc[mt][r], ss.x, ss.y);
                        if (relu) v = (v < 0.f) ? 0.f : v;
                        if constexpr (BNB) { if (addy) v = v + ya[r - r0]; }
                        __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(int, v), ry, voy, so_tile + (unsigned)orow * rowB, 0);
                    }
                }
            }
        }
    }
    if constexpr (F16) {
        if (rlog != nullptr && ct_begin == 0) {
            // (lanes of padded columns re-read a valid column, a wave past the last group re-reads group 0: real values only)
            range_publish(rlog, wave_umax(range_amax_bits(xr)), lane);
            if (blockIdx.x == 0 && threadIdx.x == 0)
                atomicMax(rlog + 1, reinterpret_cast<const unsigned *>(Wp3 + (long long)CT * KCP * NTW * 64)[0]);
        }
    }
}


// ---------------------------------------------------------------------------------------------------------------------------
// fp16-split layer, second generation ("h3r").  Same arithmetic, operand order and outputs as pointmlp_x3_kernel<.., true>
// (bit-identical: every accumulator sees the same MFMAs in the same order); a different pipeline:
//   * W goes global -> LDS by LDS-DMA into a ring of H3R_SLOTS stages, each stage = S = 2 K chunks x MT = 4 output tiles
//     x 2 terms = 16 KiB; a wave moves ONE (tile, chunk pair) per stage = 4 KiB of consecutive pack bytes: one scalar base, one M0,
//     four instruction offsets.  Three stages are in flight (the first-generation kernel staged through registers with one stage
//     of look-ahead: at the node-level shapes -- 4096 ... 36864 columns -- every 6-18 MFMAs waited for an L2 round trip).
//   * X rows are prefetched two stages ahead (f32, raw buffer loads), split into the fp16 terms for chunk i + 1 while the 12 MFMAs
//     of chunk i run; A fragments are read one chunk ahead.
//   * one barrier per stage (24 MFMAs per wave), 64 KiB of LDS + the affine table: two workgroups per CU.
// The DMA requests are invisible to hipcc's s_waitcnt bookkeeping: each iteration issues a CONSTANT number of memory
// operations (stage indices are clamped, not skipped), so "the DMA of stage st has landed" is vmcnt(H3R_WAIT) at every stage.
#ifndef H3R_ABL
#define H3R_ABL 0                                              // timing ablations (tools/build_variant.sh): never set in the product
#endif
constexpr int H3R_MT = 4, H3R_S = 2, H3R_SLOTS = 4, H3R_SLOT_SL = H3R_S * H3R_MT * 2;
constexpr int H3R_WAIT = 2 * 4 + 2 * H3R_S * 8;               // the DMAs of two later stages + the X loads of two stages

// NC = 32-column tiles per wave; the product instantiates NC = 1.  NC = 2 (variants build, a measured record): every A fragment read
// from LDS feeds two MFMAs, the weight stream is fetched once per 256 columns instead of 128, one workgroup per CU with 128
// accumulator registers per lane.  A lane then owns the ADJACENT columns 2j, 2j+1 (tile c = column parity), so X loads and Y stores
// are 8 bytes wide and their count per stage -- which the hand-counted vmcnt relies on -- is the same as for NC = 1.  Needs an even
// L and no gather index.
template <int NC>
__global__ __launch_bounds__(X3_THREADS, NC == 1 ? 2 : 1) void pointmlp_h3r_kernel(
    const float *__restrict__ x1, int C1, const float *__restrict__ x2, int C2, const uint4 *__restrict__ Wp2,
    const float *__restrict__ scale, const float *__restrict__ shift, int relu, float *__restrict__ y,
    int Cout, int L, int gpc, long long ngroups, int CT, int KC, int ct_per_y,
    const int32_t *__restrict__ gidx, int L1, unsigned *__restrict__ rlog, int KCP, int nslab, int ncol /*column groups of 128*/,
    const float *__restrict__ zadd /*optional per-node addend, as in pointmlp_x3_kernel*/, const int32_t *__restrict__ zidx, int ZM,
    unsigned *__restrict__ kmax /*optional [B][Cout][KM] ordered keys: the output is max-reduced over the columns l with the same l % KM
                                  (KNNModule: k-major columns, max over the K neighbour planes) and y is not written*/, int KM,
    double *__restrict__ stats_partial /*optional [ncol][Cout][2]: statistics epilogue, as in pointmlp_x3_kernel*/)
{
    constexpr int MT = H3R_MT, S = H3R_S;
    // Workgroup -> (column group, output slab), XCD-aware: consecutive workgroup ids go round-robin over the 8 XCDs, each with
    // its own L2; the `nslab` workgroups that read the SAME X columns get consecutive slots of ONE XCD, so the panel comes from HBM
    // once and from that L2 nslab - 1 times (x-major grids re-read it from HBM / MALL per slab: the node-level layers ran at the
    // bandwidth of that re-read, not at anything the pipeline could fix).  Grid = ceil(ncol / 8) * 8 * nslab.
    const int wg_xcd = blockIdx.x & 7, wg_local = blockIdx.x >> 3;
    const int wg_col = (wg_local / nslab) * 8 + wg_xcd, wg_slab = wg_local - (wg_local / nslab) * nslab;
    if (wg_col >= ncol) return;
    struct Lds { uint4 wsm[H3R_SLOTS][H3R_SLOT_SL][64]; float2 affine[1024]; float2 red[X3_WAVES][H3R_MT * 32]; };   // W ring first: LDS-DMA addresses below 64 KiB
    __shared__ __attribute__((aligned(16))) Lds lds;

    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int j = lane & 31, h = lane >> 5;

    long long q = (long long)wg_col * X3_WAVES + wave;
    const bool wave_valid = q < ngroups;
    q = wave_valid ? q : 0;
    const long long b = q / gpc;
    const int l0 = (int)(q - b * gpc) * (32 * NC);              // gpc = groups of 32 NC columns per cloud
    const bool pv = wave_valid && (l0 + NC * j < L);            // (NC = 2: L is even, so column 2j + 1 is valid with 2j)
    const int lc = (l0 + NC * j < L) ? l0 + NC * j : l0;        // the lane's first column

    const unsigned rowB = (unsigned)L * 4u, rowB1 = (unsigned)L1 * 4u;
    const __amdgpu_buffer_rsrc_t r1 = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float *>(x1 + b * (long long)C1 * L1), 0, (int)((unsigned)C1 * rowB1), 0x00020000);
    const __amdgpu_buffer_rsrc_t r2 = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float *>(x2 ? x2 + b * (long long)C2 * L : x1), 0, (int)((unsigned)(x2 ? C2 : 0) * rowB), 0x00020000);
    const __amdgpu_buffer_rsrc_t ry = __builtin_amdgcn_make_buffer_rsrc(
        y + b * (long long)Cout * L, 0, (int)((unsigned)Cout * rowB), 0x00020000);
    const unsigned vox = (unsigned)(8 * h * L + lc) * 4u;
    unsigned vox1 = vox;
    if constexpr (NC == 1) {
        if (gidx) {
            const int src = gidx[b * L + lc];
            vox1 = (unsigned)src < (unsigned)L1 ? (unsigned)(8 * h * L1 + src) * 4u : 0x7FFFFF00u;
        }
    }
    const unsigned voy = (unsigned)(4 * h * L + lc) * 4u;
    const unsigned vow = (unsigned)lane * 16u;

    const int KC1 = C2 > 0 ? (C1 >> 4) : KC;
    const int nstage = (KC + S - 1) / S;

    auto load_b = [&](float (&raw)[S][8][NC], int st) {
#pragma unroll
        for (int i = 0; i < S; ++i) {
            const int kc = st * S + i;                          // (a chunk past KC reads past both panels: zeros)
            const bool second = kc >= KC1;
            const unsigned rb = second ? rowB : rowB1;
            const unsigned row0 = (unsigned)(16 * (second ? kc - KC1 : kc)) * rb;
#pragma unroll
            for (int t = 0; t < 8; ++t) {
                const unsigned so = row0 + (unsigned)t * rb;
                if constexpr (NC == 1) {
                    raw[i][t][0] = __builtin_bit_cast(float, second ? __builtin_amdgcn_raw_buffer_load_b32(r2, vox, so, 0)
                                                                     : __builtin_amdgcn_raw_buffer_load_b32(r1, vox1, so, 0));
                } else {
                    const f32x2_t v2 = __builtin_bit_cast(f32x2_t, second ? __builtin_amdgcn_raw_buffer_load_b64(r2, vox, so, 0)
                                                                           : __builtin_amdgcn_raw_buffer_load_b64(r1, vox1, so, 0));
                    raw[i][t][0] = v2[0]; raw[i][t][1] = v2[1];
                }
            }
        }
    };

    const int ct_begin = wg_slab * ct_per_y;
    const int ct_end = min(CT, ct_begin + ct_per_y);
    for (int o = ct_begin * 32 + (int)threadIdx.x; o < ct_end * 32; o += X3_THREADS)
        lds.affine[o - ct_begin * 32] = make_float2(scale[o] * 0.03125f, shift[o]);       // accumulators hold 32 W.x

    // this wave's DMA unit of a stage: output tile `wave`, both chunks, both terms = bytes [wave * 4 KiB, +4 KiB) of the slot
    const unsigned wsm_lds = (unsigned)reinterpret_cast<size_t>(&lds.wsm[0][0][0]);
    const unsigned dma_dst_w = wsm_lds + (unsigned)wave * 4096u;
    const uint4 *lds_w = &lds.wsm[0][0][lane];

    RangeAcc xr = {0, 0u};
    for (int ct0 = ct_begin; ct0 < ct_end; ct0 += MT) {
        f32x16 acc[MT][NC];
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int c = 0; c < NC; ++c)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[mt][c][r] = 0.f;

        const char *gw = reinterpret_cast<const char *>(Wp2) + ((size_t)(ct0 + wave) * KCP) * 2048u;
        auto dma = [&](int st) {                               // stage st (clamped) -> slot st % H3R_SLOTS
            const int sc = st < nstage ? st : nstage - 1;
            const char *g = gw + (size_t)sc * (S * 2048);
            const unsigned d = dma_dst_w + (unsigned)(st & (H3R_SLOTS - 1)) * (unsigned)(H3R_SLOT_SL * 1024), vo = vow;
            unsigned keep;
            // (s_nop 4: an operand may arrive in an SGPR written by a VALU instruction -- v_readlane of a spill, v_readfirstlane --,
            // and a VMEM instruction reading such an SGPR needs 5 wait states that hipcc does not add inside inline asm)
            asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 4\n\t"
                         "global_load_lds_dwordx4 %1, %2\n\tglobal_load_lds_dwordx4 %1, %2 offset:1024\n\t"
                         "global_load_lds_dwordx4 %1, %2 offset:2048\n\tglobal_load_lds_dwordx4 %1, %2 offset:3072\n\t"
                         "s_mov_b32 m0, %0"
                         : "=&s"(keep) : "v"(vo), "s"(g), "s"(d) : "memory");
        };
        // slot layout = pack layout of the four tiles side by side: [mt][i][term] -> slice (mt * S + i) * 2 + term
        // (the range maxima are tracked unconditionally -- two v_max3 per value pair in the MFMA shadow; a branch here would cut
        // the scheduling region that interleaves the split with the MFMAs -- and published by slab 0 only)
        auto split = [&](const float (&raw)[8][NC], unsigned (&bh)[NC][4], unsigned (&bm)[NC][4], unsigned (&bl)[NC][4]) {
#pragma unroll
            for (int c = 0; c < NC; ++c) {
#pragma unroll
                for (int p = 0; p < 4; ++p) range_track(xr, raw[2 * p][c], raw[2 * p + 1][c]);
#pragma unroll
                for (int p = 0; p < 4; ++p) split16_pair_med3(raw[2 * p][c], raw[2 * p + 1][c], bh[c][p], bm[c][p], bl[c][p]);
            }
        };
        // one MFMA, then VALU_PER of the split's vector instructions in its shadow (a wave issues in order: twelve MFMAs in a row
        // followed by the split leave the matrix pipe idle for the ~200 cycles of the split; hipcc emits exactly that by itself)
        auto interleave = [&]() {
#pragma unroll
            for (int k = 0; k < 3 * MT * NC; ++k) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x002, 4, 0);
            }
        };
        auto read_a = [&](uint4 (&Ar)[MT], uint4 (&Ah)[MT], int slot, int i) {
            const uint4 *base = lds_w + (size_t)slot * (H3R_SLOT_SL * 64);
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) {
                Ar[mt] = base[((mt * S + i) * 2 + 1) * 64];
                Ah[mt] = base[((mt * S + i) * 2 + 0) * 64];
            }
        };
        auto mfmas = [&](const uint4 (&Ar)[MT], const uint4 (&Ah)[MT], const unsigned (&bh)[NC][4], const unsigned (&bm)[NC][4], const unsigned (&bl)[NC][4]) {
            f16x8 Bh[NC], Bm[NC], Bl[NC];
#pragma unroll
            for (int c = 0; c < NC; ++c) {
                Bh[c] = __builtin_bit_cast(f16x8, make_uint4(bh[c][0], bh[c][1], bh[c][2], bh[c][3]));     // 32 xh
                Bm[c] = __builtin_bit_cast(f16x8, make_uint4(bm[c][0], bm[c][1], bm[c][2], bm[c][3]));     // 32 * residual
                Bl[c] = __builtin_bit_cast(f16x8, make_uint4(bl[c][0], bl[c][1], bl[c][2], bl[c][3]));     // xh
            }
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                for (int c = 0; c < NC; ++c) acc[mt][c] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, Ar[mt]), Bl[c], acc[mt][c], 0, 0, 0);
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                for (int c = 0; c < NC; ++c) acc[mt][c] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, Ah[mt]), Bm[c], acc[mt][c], 0, 0, 0);
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                for (int c = 0; c < NC; ++c) acc[mt][c] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, Ah[mt]), Bh[c], acc[mt][c], 0, 0, 0);
        };

        float xa[S][8][NC], xb[S][8][NC], xc[S][8][NC];
        unsigned bh[2][NC][4], bm[2][NC][4], bl[2][NC][4];
        __syncthreads();                                        // every wave is done with the previous tile group's slots (and the affine table is written)
        dma(0); dma(1); dma(2);
        load_b(xa, 0);
        load_b(xb, nstage > 1 ? 1 : 0);
        split(xa[0], bh[0], bm[0], bl[0]);
        // stage st: X in xcur (chunk 0 already split into set 0), xnxt = stage st + 1, xfar receives stage st + 2
        // (a third stage of X look-ahead measured 5-10 % SLOWER: r02y)
#define H3R_STAGE(st, xcur, xnxt, xfar)                                                                      \
        {                                                                                                    \
            if (!(H3R_ABL & 8)) { asm volatile("s_waitcnt vmcnt(%0)" :: "n"(H3R_WAIT) : "memory");   /* this wave's unit of stage st has landed */ \
            __builtin_amdgcn_s_barrier(); }                                                                  \
            if (!(H3R_ABL & 2)) dma((st) + 3);                  /* into the slot of stage st - 1 */          \
            if (!(H3R_ABL & 1)) load_b(xfar, (st) + 2 < nstage ? (st) + 2 : nstage - 1);                      \
            const int slot = (st) & (H3R_SLOTS - 1);                                                         \
            uint4 Ar0[MT], Ah0[MT], Ar1[MT], Ah1[MT];                                                        \
            read_a(Ar0, Ah0, slot, 0);                                                                       \
            read_a(Ar1, Ah1, slot, 1);                                                                       \
            mfmas(Ar0, Ah0, bh[0], bm[0], bl[0]);                                                            \
            split(xcur[1], bh[1], bm[1], bl[1]);                                                             \
            interleave();                                                                                    \
            mfmas(Ar1, Ah1, bh[1], bm[1], bl[1]);                                                            \
            split(xnxt[0], bh[0], bm[0], bl[0]);                                                             \
            interleave();                                                                                    \
        }
        int st = 0;
        for (; st + 3 <= nstage; st += 3) {
            H3R_STAGE(st, xa, xb, xc)
            H3R_STAGE(st + 1, xb, xc, xa)
            H3R_STAGE(st + 2, xc, xa, xb)
        }
        if (st < nstage) {
            H3R_STAGE(st, xa, xb, xc)
            if (st + 1 < nstage) H3R_STAGE(st + 1, xb, xc, xa)
        }
#undef H3R_STAGE
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");        // the clamped re-loads of the tail: nothing may land after the next group starts

        // y (c = 0 .. NC-1: the lane's adjacent columns) leaves as one 4 NC-byte store per (row, lane)
        auto store_row = [&](const float (&v)[NC], unsigned vo, unsigned so) {
            if constexpr (NC == 1) __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(int, v[0]), ry, vo, so, 0);
            else { const f32x2_t v2 = {v[0], v[1]}; __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(decltype(__builtin_amdgcn_raw_buffer_load_b64(ry, 0u, 0u, 0)), v2), ry, vo, so, 0); }
        };
        if (stats_partial != nullptr) {                        // BatchNorm batch statistics from the epilogue (see pointmlp_x3_kernel)
            const unsigned voy_s = pv ? voy : 0x7FFFFF00u;
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) {
                const unsigned so_tile = (unsigned)((ct0 + mt) * 32) * rowB;
                const float2 *aff = lds.affine + (ct0 + mt - ct_begin) * 32 + 4 * h;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int orow = (r & 3) + 8 * (r >> 2);
                    const float2 ss = aff[orow];
                    float v[NC], sv = 0.f, sq = 0.f;
#pragma unroll
                    for (int c = 0; c < NC; ++c) {
                        v[c] = __fmaf_rn(acc[mt][c][r], ss.x, ss.y);
                        if (relu) v[c] = (v[c] < 0.f) ? 0.f : v[c];
                        const float vv = pv ? v[c] : 0.f;
                        sv = c == 0 ? vv : sv + vv;
                        sq = c == 0 ? vv * vv : sq + vv * vv;
                    }
                    store_row(v, voy_s, so_tile + (unsigned)orow * rowB);
                    const float s1 = row32_sum(sv), s2 = row32_sum(sq);
                    if (j == 0) lds.red[wave][mt * 32 + orow + 4 * h] = make_float2(s1, s2);
                }
            }
            __syncthreads();
            for (int t = threadIdx.x; t < MT * 32; t += X3_THREADS) {
                const double a = ((double)lds.red[0][t].x + (double)lds.red[1][t].x) + ((double)lds.red[2][t].x + (double)lds.red[3][t].x);
                const double qq = ((double)lds.red[0][t].y + (double)lds.red[1][t].y) + ((double)lds.red[2][t].y + (double)lds.red[3][t].y);
                double *dst = stats_partial + ((size_t)wg_col * Cout + (size_t)ct0 * 32 + t) * 2;
                dst[0] = a;
                dst[1] = qq;
            }
            __syncthreads();
        } else if (NC == 1 && kmax != nullptr) {
            // max over the neighbour planes in the epilogue: column l of a k-major tensor belongs to node l % KM; a 32-column tile of
            // one plane is 32 consecutive nodes, so a store instruction's 32 lanes hit 32 consecutive keys of one channel row -- an
            // atomic per element costs what the store would, and B x C x K*M never exists (torch.max(dim=3) of models/layers.py:350).
            // Keys: order-preserving integers (sign flipped for positives, all bits for negatives), memset 0 = below everything.
            if (pv) {
                unsigned *kb = kmax + (size_t)b * Cout * KM + (lc % KM);
#pragma unroll
                for (int mt = 0; mt < MT; ++mt) {
                    const float2 *aff = lds.affine + (ct0 + mt - ct_begin) * 32 + 4 * h;
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int orow = (r & 3) + 8 * (r >> 2);
                        const float2 ss = aff[orow];
                        float v = __fmaf_rn(acc[mt][0][r], ss.x, ss.y);
                        if (relu) v = (v < 0.f) ? 0.f : v;
                        const unsigned u = __float_as_uint(v);
                        atomicMax(kb + (size_t)((ct0 + mt) * 32 + orow + 4 * h) * KM, (u & 0x80000000u) ? ~u : (u | 0x80000000u));
                    }
                }
            }
        } else if (pv && zadd != nullptr) {                    // per-node addend gathered here (see pointmlp_x3_kernel)
            const float *zb[NC];
            bool zok[NC];
#pragma unroll
            for (int c = 0; c < NC; ++c) {
                const int zm = zidx[b * L + lc + c];
                zok[c] = (unsigned)zm < (unsigned)ZM;
                zb[c] = zadd + ((size_t)b * Cout) * ZM + (zok[c] ? zm : 0);
            }
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) {
                const unsigned so_tile = (unsigned)((ct0 + mt) * 32) * rowB;
                const float2 *aff = lds.affine + (ct0 + mt - ct_begin) * 32 + 4 * h;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int orow = (r & 3) + 8 * (r >> 2);
                    const float2 ss = aff[orow];
                    float v[NC];
#pragma unroll
                    for (int c = 0; c < NC; ++c) {
                        const float zv = zok[c] ? zb[c][(size_t)((ct0 + mt) * 32 + orow + 4 * h) * ZM] : 0.f;
                        v[c] = __fmaf_rn(acc[mt][c][r], ss.x, __fmaf_rn(zv, ss.x * 32.f, ss.y));
                        if (relu) v[c] = (v[c] < 0.f) ? 0.f : v[c];
                    }
                    store_row(v, voy, so_tile + (unsigned)orow * rowB);
                }
            }
        } else if (pv) {
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) {
                const unsigned so_tile = (unsigned)((ct0 + mt) * 32) * rowB;
                const float2 *aff = lds.affine + (ct0 + mt - ct_begin) * 32 + 4 * h;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int orow = (r & 3) + 8 * (r >> 2);
                    const float2 ss = aff[orow];
                    float v[NC];
#pragma unroll
                    for (int c = 0; c < NC; ++c) {
                        v[c] = __fmaf_rn(acc[mt][c][r], ss.x, ss.y);
                        if (relu) v[c] = (v[c] < 0.f) ? 0.f : v[c];
                    }
                    store_row(v, voy, so_tile + (unsigned)orow * rowB);
                }
            }
        }
    }
    if (rlog != nullptr && ct_begin == 0) {
        range_publish(rlog, wave_umax(range_amax_bits(xr)), lane);
        if (wg_col == 0 && threadIdx.x == 0)
            atomicMax(rlog + 1, reinterpret_cast<const unsigned *>(Wp2 + (long long)CT * KCP * 2 * 64)[0]);
    }
}


}  // namespace

// (shared with pointmlp_bf16.hip)
int sonet::launch_stats_finalize(const double *partial, int nwg, int C, double inv_n, float *mean, float *var, hipStream_t st)
{
    hipLaunchKernelGGL(stats_partial_finalize_kernel, dim3((unsigned)C), dim3(256), 0, st, partial, nwg, C, inv_n, mean, var, sonet::take_bn_rider());
    return 0;
}

extern "C" size_t sonet_pointmlp_x3_pack_size(int Cin, int Cout)
{
    if (Cin <= 0 || Cout <= 0) return 0;
    // 3 KiB per (cout tile, K chunk) + a 64-byte trailer: word 0 = bits of max |w| (written by the pack kernels, read by the
    // fp16-flavour kernel for the range log)
    // (bf16 flavour: 3 slices per chunk; fp16 flavour: 2 slices per chunk of the K range rounded up to H3_KPAD chunks)
    const size_t kc = (size_t)sonet::ceil_div(Cin, 16), kcp = (size_t)sonet::ceil_div(Cin, 16 * H3_KPAD) * H3_KPAD;
    const size_t per_tile = kc * 3 > kcp * 2 ? kc * 3 : kcp * 2;
    return (size_t)sonet::ceil_div(Cout, 32) * per_tile * 1024 + 64;     // bytes
}

// rows: the rows of W that exist (o >= rows packs zeros: callers pad Cout to a friendly tile count); (rs, cs): element strides of W
static int x3_pack_impl(const char *what, bool f16, const float *W, void *Wp3, int Cin, int Cout, sonet_stream_t stream,
                        int rows = -1, long long rs = 0, long long cs = 1)
{
    SONET_REQUIRE(W && Wp3, "%s: NULL pointer", what);
    SONET_REQUIRE(Cin > 0 && Cout > 0, "%s: non-positive size", what);
    if (rows < 0) { rows = Cout; rs = Cin; cs = 1; }
    SONET_REQUIRE(rows <= Cout, "%s: rows=%d > Cout=%d", what, rows, Cout);
    const int KC = f16 ? sonet::ceil_div(Cin, 16 * H3_KPAD) * H3_KPAD : sonet::ceil_div(Cin, 16);
    const long long total = (long long)sonet::ceil_div(Cout, 32) * KC * 64;
    unsigned *trailer = reinterpret_cast<unsigned *>(reinterpret_cast<uint4 *>(Wp3) + total * (f16 ? 2 : 3));
    if (hipMemsetAsync(trailer, 0, 64, sonet::as_stream(stream)) != hipSuccess) return sonet::fail(SONET_ERR_LAUNCH, "%s: memset failed", what);
    // (the kernel packs ceil(Cout / 32) tiles and zero-fills rows >= its Cout argument: pass the rows that exist)
    if (f16) hipLaunchKernelGGL(x3_pack_kernel<true>, dim3((unsigned)sonet::ceil_div64(total, 256)), dim3(256), 0, sonet::as_stream(stream),
                                W, reinterpret_cast<uint4 *>(Wp3), Cin, rows, KC, total, trailer, rs, cs);
    else     hipLaunchKernelGGL(x3_pack_kernel<false>, dim3((unsigned)sonet::ceil_div64(total, 256)), dim3(256), 0, sonet::as_stream(stream),
                                W, reinterpret_cast<uint4 *>(Wp3), Cin, rows, KC, total, trailer, rs, cs);
    return sonet::launched(what);
}

/* The pack of a matrix given by element strides: element (o, c), o < rows, c < Cin, is W[o * row_stride + c * col_stride]; rows
 * [rows, Cout) pack as zeros.  With (row_stride, col_stride) = (1, ld) and W advanced by a column offset this is the pack of a column
 * block of W TRANSPOSED -- the dgrad's weights -- read along W's own rows (coalesced), without a transposed copy. */
extern "C" int sonet_pointmlp_x3_pack_strided(const float *W, long long row_stride, long long col_stride, void *Wp3, int Cin, int Cout, int rows,
                                              sonet_stream_t stream)
{
    SONET_REQUIRE(rows > 0, "sonet_pointmlp_x3_pack_strided: rows must be positive");
    return x3_pack_impl("sonet_pointmlp_x3_pack_strided", false, W, Wp3, Cin, Cout, stream, rows, row_stride, col_stride);
}

extern "C" int sonet_pointmlp_h3_pack_strided(const float *W, long long row_stride, long long col_stride, void *Wp3, int Cin, int Cout, int rows,
                                              sonet_stream_t stream)
{
    SONET_REQUIRE(rows > 0, "sonet_pointmlp_h3_pack_strided: rows must be positive");
    return x3_pack_impl("sonet_pointmlp_h3_pack_strided", true, W, Wp3, Cin, Cout, stream, rows, row_stride, col_stride);
}

extern "C" int sonet_pointmlp_x3_pack(const float *W, void *Wp3, int Cin, int Cout, sonet_stream_t stream)
{
    return x3_pack_impl("sonet_pointmlp_x3_pack", false, W, Wp3, Cin, Cout, stream);
}

extern "C" int sonet_pointmlp_h3_pack(const float *W, void *Wp3, int Cin, int Cout, sonet_stream_t stream)
{
    return x3_pack_impl("sonet_pointmlp_h3_pack", true, W, Wp3, Cin, Cout, stream);
}

/* Refresh many weight packs in one launch (+ one that clears the max-|w| trailers).  table: n_entries records of 72 bytes on the device,
 *   { const float *W; void *Wp; int64 rs, cs, total; int32 Cin, rows, KC, flavour, blk0, nblk; int64 pad }
 * -- entry i packs the matrix with element (o, c) = W[o rs + c cs], o < rows, c < Cin (rows beyond `rows` up to the pack's tile count as zeros)
 * into Wp exactly as sonet_pointmlp_bf16_pack_strided (flavour 0), sonet_pointmlp_x3_pack_strided (1) or sonet_pointmlp_h3_pack_strided (2)
 * would; KC = the flavour's chunk count of Cin, total = 64 x tiles x KC, and the entry owns workgroups blk0 .. blk0 + nblk - 1 with
 * nblk = ceil(total / 256), blk0 ascending; total_blocks = their sum. */
extern "C" int sonet_pack_multi(const void *table, int n_entries, int total_blocks, sonet_stream_t stream)
{
    const char *what = "sonet_pack_multi";
    SONET_REQUIRE(table, "%s: NULL pointer", what);
    SONET_REQUIRE(n_entries > 0 && total_blocks > 0, "%s: empty table", what);
    hipStream_t st = sonet::as_stream(stream);
    hipLaunchKernelGGL(pack_multi_zero_kernel, dim3((unsigned)n_entries), dim3(64), 0, st, reinterpret_cast<const PackEntry *>(table), n_entries);
    hipLaunchKernelGGL(pack_multi_kernel, dim3((unsigned)total_blocks), dim3(256), 0, st, reinterpret_cast<const PackEntry *>(table), n_entries);
    return sonet::launched(what);
}

/* chunk count (KC) a flavour's pack uses for Cin input channels: what the table's KC field must hold */
extern "C" int sonet_pack_multi_kc(int flavour, int Cin)
{
    return flavour == 2 ? sonet::ceil_div(Cin, 16 * H3_KPAD) * H3_KPAD : sonet::ceil_div(Cin, 16);
}

static int x3_run_impl(const char *what, bool f16, const float *x1, int C1, const float *x2, int C2, const void *Wp3,
                       const float *scale, const float *shift, int relu, float *y,
                       int B, int Cout, int L, sonet_stream_t stream, const int32_t *gidx = nullptr, int L1 = 0,
                       double *stats_ws = nullptr, float *mean = nullptr, float *var = nullptr,
                       const float *zadd = nullptr, const int32_t *zidx = nullptr, int ZM = 0, unsigned *kmax = nullptr, int KM = 0,
                       const SegPoolArgs *segpool = nullptr, const XAffArgs *xaff = nullptr, const BnbArgs *bnbp = nullptr, double *bnb_psums = nullptr)
{
    if (!gidx) L1 = L;
    SONET_REQUIRE(L1 > 0, "%s: non-positive size", what);
    SONET_REQUIRE(x1 && Wp3 && scale && shift && (y || kmax || segpool), "%s: NULL pointer", what);
    SONET_REQUIRE(!segpool || (f16 && !gidx && !stats_ws && !zadd && !kmax), "%s: the pooled form takes the plain fp16-split layer only", what);
    const SegPoolArgs sp = segpool ? *segpool : SegPoolArgs{nullptr, nullptr, nullptr, nullptr, 0};
    SONET_REQUIRE(!xaff || (f16 && !gidx && !zadd && !kmax && xaff->s1 && xaff->h1 && ((C2 == 0) || (xaff->s2 && xaff->h2))),
                  "%s: normalise-on-load takes the plain fp16-split layer and a (scale, shift) pair per input panel", what);
    if (xaff && (C1 + C2 > 1024 || (Cout / 32) % 4 != 0)) return sonet::fail(SONET_ERR_UNSUPPORTED, "%s: normalise-on-load needs Cin <= 1024 and Cout %% 128 == 0", what);
    const XAffArgs xa = xaff ? *xaff : XAffArgs{nullptr, nullptr, nullptr, nullptr, 0};
    SONET_REQUIRE(!bnbp || (!f16 && !gidx && !zadd && !kmax && !segpool && !xaff && !stats_ws && C2 == 0 && bnbp->raw && bnbp->a && bnbp->b && bnbp->c0
                            && bnbp->sc && bnbp->sh), "%s: the BatchNorm-backward operand takes the plain bf16-split layer with one input panel", what);
    if (bnbp && C1 > 512) return sonet::fail(SONET_ERR_UNSUPPORTED, "%s: the BatchNorm-backward operand needs <= 512 input channels", what);
    SONET_REQUIRE(!bnbp || ((bnbp->praw == nullptr) == (bnbp->pstats == nullptr) && (bnbp->pstats == nullptr) == (bnb_psums == nullptr) &&
                            (!bnbp->praw || (bnbp->psc && bnbp->psh))), "%s: the sums of the layer below need praw, psc, psh, a workspace and the output", what);
    SONET_REQUIRE(!bnbp || !bnbp->yadd || !bnbp->pstats, "%s: an accumulated output and the sums of the layer below do not combine", what);
    const BnbArgs bnb = bnbp ? *bnbp : BnbArgs{nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, 0, nullptr, nullptr, nullptr, 0, nullptr, nullptr};
    SONET_REQUIRE(B > 0 && C1 > 0 && C2 >= 0 && Cout > 0 && L > 0, "%s: non-positive size", what);
    SONET_REQUIRE((C2 == 0) == (x2 == nullptr), "%s: x2 and C2 disagree", what);
    SONET_REQUIRE(C2 == 0 || C1 % 16 == 0, "%s: with a second input C1=%d must be a multiple of 16", what, C1);
    if (Cout % 32 != 0) return sonet::fail(SONET_ERR_UNSUPPORTED, "%s: Cout=%d must be a multiple of 32", what, Cout);
    const int Cin = C1 + C2;
    const int CT = Cout / 32, KC = sonet::ceil_div(Cin, 16);
    const int KCP = f16 ? sonet::ceil_div(Cin, 16 * H3_KPAD) * H3_KPAD : KC;
    const int gpc = sonet::ceil_div(L, 32);
    const long long ngroups = (long long)B * gpc;
    if ((double)C1 * L1 * 4.0 >= 2.0e9 || (double)(C1 > C2 ? C1 : C2) * L * 4.0 >= 4.0e9 || (double)Cout * L * 4.0 >= 4.0e9 || (double)CT * KCP * 3072.0 >= 2.0e9)
        return sonet::fail(SONET_ERR_UNSUPPORTED, "%s: a per-cloud panel exceeds 4 GiB", what);
    const long long nwg_x = sonet::ceil_div64(ngroups, (long long)X3_WAVES);
    if (nwg_x > 0x7FFFFFFFll) return sonet::fail(SONET_ERR_UNSUPPORTED, "%s: too many points", what);
    hipStream_t st = sonet::as_stream(stream);
    const uint4 *wp = reinterpret_cast<const uint4 *>(Wp3);
    unsigned *rlog = f16 ? sonet::range_log() : nullptr;
    const char *eg = sonet::knob("SONET_POINTMLP_H3R");          // bench-only: 0 = the first-generation pipeline
    // second-generation pipeline where it measures faster (profiles/r02y_pointmlp_h3r.log, r02zd): inputs that stay in the 256 MB
    // MALL across the CT / 4 passes over X, and point-level inputs whenever the first generation would also run 4-tile groups (CT
    // not a multiple of 6: 1024 -> 512 at 64 x 3072 columns 0.62 vs 0.78 ms).  With 6-tile groups (two passes for 384 channels
    // instead of three) the first generation wins by 15-20 % at 64 x 15000 columns.
    const bool h3r_fits = (double)Cin * (double)B * (double)L * 4.0 <= 128.0e6 && (long long)B * L >= 256;
    // (... and a long K loop: 128 -> 256 at 64 x 15000 columns -- 8 chunks, then 1 GB of output -- stays 15 % faster on the first generation)
    const bool h3r_pick = h3r_fits || (CT % 6 != 0 && Cin >= 512 && (long long)B * L >= 256);
    // (the per-node addend form measured better on the first generation: 0.86 vs 0.98 ms for 393 -> 1024 at 64 x 3072 columns)
    if (kmax && !(f16 && CT % H3R_MT == 0)) return sonet::fail(SONET_ERR_UNSUPPORTED, "%s: the max-reduced form needs Cout %% 128 == 0", what);
#ifdef SONET_VARIANTS
    // Two column tiles per wave (NC = 2), variants build only (SONET_POINTMLP_NC=2; tools/bench_h3w.py): bit-identical, and measured
    // 2 % faster on the two K >= 512 segmenter layers but 3-18 % SLOWER everywhere else (profiles/r03b_pointmlp_two_tiles.log) -- the
    // per-layer kernel is bound by its X stream, where two waves per SIMD hide more than the halved weight stream saves.
    const char *enc = sonet::knob("SONET_POINTMLP_NC");
    const bool wide = enc && atoi(enc) == 2 && f16 && CT % H3R_MT == 0 && !gidx && !kmax && (L % 2 == 0);
    if (wide) {
        const int gpc2 = sonet::ceil_div(L, 64);
        const long long ngroups2 = (long long)B * gpc2, nwg_x2 = sonet::ceil_div64(ngroups2, (long long)X3_WAVES);
        int dev = 0, cus = 256;
        if (hipGetDevice(&dev) == hipSuccess) {
            int v = 0;
            if (hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && v > 0) cus = v;
        }
        // output-channel slabs: as below, with one workgroup per CU
        const int groups = CT / H3R_MT;
        int best = 0;
        long long best_cost = 0;
        for (int d = 1; d <= groups; ++d) {
            if (groups % d != 0 || CT / d > 32) continue;
            const long long cost = sonet::ceil_div64(nwg_x2 * d, (long long)cus) * (groups / d);
            if (best == 0 || cost <= best_cost) { best = d; best_cost = cost; }
        }
        if (const char *e = sonet::knob("SONET_POINTMLP_YSPLIT")) {
            const int want = atoi(e);
            if (want >= 1 && groups % want == 0 && CT / want <= 32) best = want;
        }
        if (best == 0) return sonet::fail(SONET_ERR_UNSUPPORTED, "%s: Cout=%d too large", what, Cout);
        const long long nwg = sonet::ceil_div64(nwg_x2, 8) * 8 * best;
        if (nwg > 0x7FFFFFFFll) return sonet::fail(SONET_ERR_UNSUPPORTED, "%s: too many points", what);
        hipLaunchKernelGGL(pointmlp_h3r_kernel<2>, dim3((unsigned)nwg), dim3(X3_THREADS), 0, st,
                           x1, C1, x2, C2, wp, scale, shift, relu, y, Cout, L, gpc2, ngroups2, CT, KC, CT / best, gidx, L1, rlog, KCP, best, (int)nwg_x2,
                           zadd, zidx, ZM, kmax, KM, stats_ws);
        if (stats_ws) sonet::launch_stats_finalize(stats_ws, (int)nwg_x2, Cout, 1.0 / ((double)B * L), mean, var, st);
        return sonet::launched(what);
    }
#endif
    if (!segpool && !xaff && f16 && CT % H3R_MT == 0 && (kmax || (eg ? atoi(eg) != 0 : (h3r_pick && !zadd)))) {
        // output-channel slabs: the divisor d of the CT / 4 tile groups that needs the fewest rounds of (2 workgroups per CU)
        // x (groups per workgroup); ties go to the larger d (shorter workgroups)
        int dev = 0, cus = 256;
        if (hipGetDevice(&dev) == hipSuccess) {
            int v = 0;
            if (hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && v > 0) cus = v;
        }
        const int groups = CT / H3R_MT;
        const long long slots = 2ll * cus;
        int best = 0;
        long long best_cost = 0;
        for (int d = 1; d <= groups; ++d) {
            if (groups % d != 0 || CT / d > 32) continue;
            const long long cost = sonet::ceil_div64(nwg_x * d, slots) * (groups / d);
            if (best == 0 || cost <= best_cost) { best = d; best_cost = cost; }
        }
        if (const char *e = sonet::knob("SONET_POINTMLP_YSPLIT")) {
            const int want = atoi(e);
            if (want >= 1 && groups % want == 0 && CT / want <= 32) best = want;
        }
        if (best == 0) return sonet::fail(SONET_ERR_UNSUPPORTED, "%s: Cout=%d too large", what, Cout);
        const long long nwg = sonet::ceil_div64(nwg_x, 8) * 8 * best;
        if (nwg > 0x7FFFFFFFll) return sonet::fail(SONET_ERR_UNSUPPORTED, "%s: too many points", what);
        hipLaunchKernelGGL(pointmlp_h3r_kernel<1>, dim3((unsigned)nwg), dim3(X3_THREADS), 0, st,
                           x1, C1, x2, C2, wp, scale, shift, relu, y, Cout, L, gpc, ngroups, CT, KC, CT / best, gidx, L1, rlog, KCP, best, (int)nwg_x,
                           zadd, zidx, ZM, kmax, KM, stats_ws);
        if (stats_ws) sonet::launch_stats_finalize(stats_ws, (int)nwg_x, Cout, 1.0 / ((double)B * L), mean, var, st);
        return sonet::launched(what);
    }
    int MT = 1, S = 1;
    // (8 tiles per group -- half the passes over a point-level input panel -- needs 290 registers: one wave per SIMD, and measured
    // slower than 4 where it matters: 1024 -> 512 at 64 x 3072 columns 0.69 vs 0.77 ms but 128 -> 256 at 64 x 15000 and the
    // segmenter's first layer lose 20-70 %.  Not instantiated.)
    if (CT % 6 == 0) MT = 6;
    else if (CT % 4 == 0 && !(nwg_x < 64)) MT = 4;
    else if (CT % 2 == 0) MT = 2;
    if (const char *e = sonet::knob("SONET_POINTMLP_MT")) {      // tuning knob (bench experiments only)
        const int want = atoi(e);
        if ((want == 6 || want == 4 || want == 2 || want == 1) && CT % want == 0) MT = want;
    }
    if (const char *e = sonet::knob("SONET_POINTMLP_S")) {
        const int want = atoi(e);
        if (want == 1 || want == 2) S = want;
    }
    if (KC == 1) S = 1;
    if (xaff && MT != 6) MT = 4;
    // output-channel slabs per column group: the smallest divisor of the CT / MT tile groups that fills the chip (>= 1024 workgroups) --
    // any divisor, not only powers of two: 768 -> 640 at 4096 columns (the dgrad of the final PointNet's first layer: 5 groups of 4 tiles)
    // ran on 32 workgroups, 0.20 ms for 0.03 ms of matrix work -- and slabs of <= 32 tiles (the affine table)
    int ysplit = 1;
    {
        const int groups = CT / MT;
        int best = 0;
        for (int d = 1; d <= groups; ++d) {
            if (groups % d != 0 || CT / d > 32) continue;
            best = d;
            if (nwg_x * d >= 1024) break;
        }
        if (best > 0) ysplit = best;
    }
    if (const char *e = sonet::knob("SONET_POINTMLP_YSPLIT")) {  // tuning knob (bench experiments only): output-channel slabs per column group.
        const int want = atoi(e);                           // 1152 workgroups on 768 resident slots run 1.5 rounds; 2 slabs of half the work
        if (want >= 1 && (CT / MT) % want == 0 && CT / want <= 32) ysplit = want;   // each would run 3 rounds of half the length (X re-read twice)
    }
    const int ct_per_y = CT / ysplit;
    if (ct_per_y > 32) return sonet::fail(SONET_ERR_UNSUPPORTED, "%s: Cout=%d too large", what, Cout);
    dim3 grid((unsigned)nwg_x, (unsigned)ysplit), block(X3_THREADS);
#define X3_ARGS grid, block, 0, st, x1, C1, x2, C2, wp, scale, shift, relu, y, Cout, L, gpc, ngroups, CT, KC, ct_per_y, gidx, L1, rlog, KCP, stats_ws, zadd, zidx, ZM, sp, xa, bnb
#define X3_LAUNCH(MM) do { if (f16) { if (S == 2) hipLaunchKernelGGL((pointmlp_x3_kernel<MM, 2, true>), X3_ARGS); \
                                      else        hipLaunchKernelGGL((pointmlp_x3_kernel<MM, 1, true>), X3_ARGS); } \
                           else     { if (S == 2) hipLaunchKernelGGL((pointmlp_x3_kernel<MM, 2, false>), X3_ARGS); \
                                      else        hipLaunchKernelGGL((pointmlp_x3_kernel<MM, 1, false>), X3_ARGS); } } while (0)
    if (zadd && f16 && CT % 4 == 0 && !stats_ws && nwg_x >= 512 && !sonet::knob("SONET_POINTMLP_MT")) {      // (small launches: more slabs instead)
        // per-node addend: 4-tile groups with the gathers in flight during the K loop
        dim3 gz((unsigned)nwg_x, (unsigned)sonet::ceil_div(CT, 32));      // (slabs of <= 32 tiles: the affine table; a workgroup walks its groups)
        const int cpy = CT / (int)gz.y;
        if (CT % (int)gz.y == 0 && cpy % 4 == 0 && cpy <= 32) {
            hipLaunchKernelGGL((pointmlp_x3_kernel<4, 1, true, true>), gz, block, 0, st, x1, C1, x2, C2, wp, scale, shift, relu, y, Cout, L, gpc, ngroups,
                               CT, KC, cpy, gidx, L1, rlog, KCP, stats_ws, zadd, zidx, ZM, sp, xa, bnb);
            return sonet::launched(what);
        }
    }
    if (bnbp) {
        switch (MT) {
            case 6: hipLaunchKernelGGL((pointmlp_x3_kernel<6, 1, false, false, false, false, true>), X3_ARGS); break;
            case 4: hipLaunchKernelGGL((pointmlp_x3_kernel<4, 1, false, false, false, false, true>), X3_ARGS); break;
            case 2: hipLaunchKernelGGL((pointmlp_x3_kernel<2, 1, false, false, false, false, true>), X3_ARGS); break;
            default: hipLaunchKernelGGL((pointmlp_x3_kernel<1, 1, false, false, false, false, true>), X3_ARGS);
        }
        if (bnb.pstats) hipLaunchKernelGGL(bwd_sums_finalize_kernel, dim3((unsigned)Cout), dim3(256), 0, st, bnb.pstats, (int)nwg_x, Cout, bnb_psums);
        return sonet::launched(what);
    }
    if (xaff) {                                                  // (MT is 6 or 4 here: Cout % 128 == 0)
        if (segpool) {
            if (MT == 6) hipLaunchKernelGGL((pointmlp_x3_kernel<6, 1, true, false, true, true>), X3_ARGS);
            else         hipLaunchKernelGGL((pointmlp_x3_kernel<4, 1, true, false, true, true>), X3_ARGS);
            return sonet::launched(what);
        }
        if (MT == 6) hipLaunchKernelGGL((pointmlp_x3_kernel<6, 1, true, false, false, true>), X3_ARGS);
        else         hipLaunchKernelGGL((pointmlp_x3_kernel<4, 1, true, false, false, true>), X3_ARGS);
        if (stats_ws) sonet::launch_stats_finalize(stats_ws, (int)nwg_x, Cout, 1.0 / ((double)B * L), mean, var, st);
        return sonet::launched(what);
    }
    if (segpool) {
        // (one K chunk per stage, as the storing launch of the same shape)
        switch (MT) {
            case 6: hipLaunchKernelGGL((pointmlp_x3_kernel<6, 1, true, false, true>), X3_ARGS); break;
            case 4: hipLaunchKernelGGL((pointmlp_x3_kernel<4, 1, true, false, true>), X3_ARGS); break;
            case 2: hipLaunchKernelGGL((pointmlp_x3_kernel<2, 1, true, false, true>), X3_ARGS); break;
            default: hipLaunchKernelGGL((pointmlp_x3_kernel<1, 1, true, false, true>), X3_ARGS);
        }
        return sonet::launched(what);
    }
    switch (MT) {
        case 6: X3_LAUNCH(6); break;
        case 4: X3_LAUNCH(4); break;
        case 2: X3_LAUNCH(2); break;
        default: X3_LAUNCH(1);
    }
#undef X3_LAUNCH
#undef X3_ARGS
    if (stats_ws) sonet::launch_stats_finalize(stats_ws, (int)nwg_x, Cout, 1.0 / ((double)B * L), mean, var, st);
    return sonet::launched(what);
}

/* y = act((W . cat(x1, x2) + zadd[b][o][zidx[b][l]]) * scale + shift): the fp16-split layer with a per-node addend gathered in the
 * epilogue (zadd [B][Cout][ZM], zidx [B][L] i32, out of range: + 0). */
extern "C" int sonet_pointmlp_h3_nodeadd_f32(const float *x1, int C1, const float *x2, int C2, const void *Wp3, const float *scale,
                                             const float *shift, int relu, float *y, int B, int Cout, int L,
                                             const float *zadd, const int32_t *zidx, int ZM, sonet_stream_t stream)
{
    SONET_REQUIRE(zadd && zidx && ZM > 0, "sonet_pointmlp_h3_nodeadd_f32: NULL pointer or ZM <= 0");
    return x3_run_impl("sonet_pointmlp_h3_nodeadd_f32", true, x1, C1, x2, C2, Wp3, scale, shift, relu, y, B, Cout, L, stream, nullptr, 0,
                       nullptr, nullptr, nullptr, zadd, zidx, ZM);
}

namespace {
__global__ __launch_bounds__(256) void segpool_init_kernel(unsigned long long *__restrict__ keys, long long n)
{
    const long long t = (long long)blockIdx.x * 256 + threadIdx.x;
    if (t < n) keys[t] = SP_INIT_KEY;
}
// keys -> (column, value): the epilogue of index_max_kernel (index_max.hip) on the combined keys.  A bin nothing beat (an empty node,
// or nothing above -1000) and a masked node report original column 0 -- sorted position pos0[b] -- and the layer's value there.
__global__ __launch_bounds__(256) void segpool_decode_kernel(const unsigned long long *__restrict__ keys, const float *__restrict__ v0,
                                                              const int32_t *__restrict__ pos0, const int32_t *__restrict__ row_max,
                                                              int32_t *__restrict__ out_idx, float *__restrict__ out_val, int Cout, int M, long long n)
{
    const long long t = (long long)blockIdx.x * 256 + threadIdx.x;
    if (t >= n) return;
    const long long bc = t / M;
    const int m = (int)(t - bc * M);
    const long long b = bc / Cout;
    const unsigned long long key = keys[t];
    const bool won = key != SP_INIT_KEY && (row_max == nullptr || row_max[b * M + m] != 0);
    const unsigned okey = (unsigned)(key >> 32);
    out_idx[t] = won ? (int)(0xFFFFFFFFu - (unsigned)(key & 0xFFFFFFFFull)) : pos0[b];
    out_val[t] = won ? __uint_as_float((okey & 0x80000000u) ? (okey ^ 0x80000000u) : ~okey) : v0[bc];
}
}  // namespace

extern "C" size_t sonet_pointmlp_h3_segpool_ws_size(int B, int Cout, int M)
{
    if (B <= 0 || Cout <= 0 || M <= 0) return 0;
    return (size_t)B * Cout * M * sizeof(unsigned long long) + (size_t)B * Cout * sizeof(float);
}

/* The fp16-split layer and the per-node arg-max pool of its output in ONE pass over NODE-SORTED columns; the output itself is never
 * written (models/layers.py:431 + models/networks.py:180-185 in training, when only the pooled map is consumed).  ids_sorted [B][L] i32
 * non-decreasing per cloud, pos0 [B] = sorted position of original column 0, row_max [B][M] or NULL.  out_idx [B][Cout][M] = winning
 * SORTED column (pos0[b] where nothing beat -1000 or the node is masked), out_val = the layer's value there: what sonet_pointmlp_h3_f32
 * + sonet_index_max_gather_f32 report on the sorted tensor, with position 0 of the original order in place of position 0. */
extern "C" int sonet_pointmlp_h3_segpool_f32(const float *x1, int C1, const float *x2, int C2, const void *Wp3, const float *scale,
                                             const float *shift, int relu, const int32_t *ids_sorted, const int32_t *pos0,
                                             const int32_t *row_max, int M, void *ws, int32_t *out_idx, float *out_val,
                                             int B, int Cout, int L, const float *xs1, const float *xh1, const float *xs2, const float *xh2,
                                             int xrelu, sonet_stream_t stream)
{
    const char *what = "sonet_pointmlp_h3_segpool_f32";
    SONET_REQUIRE(ids_sorted && pos0 && ws && out_idx && out_val, "%s: NULL pointer", what);
    SONET_REQUIRE(B > 0 && Cout > 0 && M > 0 && L > 0, "%s: non-positive size", what);
    const long long n = (long long)B * Cout * M;
    if (n > 0x7FFFFFFFll * 256) return sonet::fail(SONET_ERR_UNSUPPORTED, "%s: too many bins", what);
    unsigned long long *keys = reinterpret_cast<unsigned long long *>(ws);
    float *v0 = reinterpret_cast<float *>(keys + n);
    hipStream_t st = sonet::as_stream(stream);
    hipLaunchKernelGGL(segpool_init_kernel, dim3((unsigned)sonet::ceil_div64(n, 256)), dim3(256), 0, st, keys, n);
    const SegPoolArgs sp = {ids_sorted, pos0, keys, v0, M};
    const XAffArgs xa = {xs1, xh1, xs2, xh2, xrelu};
    const int rc = x3_run_impl(what, true, x1, C1, x2, C2, Wp3, scale, shift, relu, nullptr, B, Cout, L, stream, nullptr, 0,
                               nullptr, nullptr, nullptr, nullptr, nullptr, 0, nullptr, 0, &sp, xs1 ? &xa : nullptr);
    if (rc != SONET_OK) return rc;
    hipLaunchKernelGGL(segpool_decode_kernel, dim3((unsigned)sonet::ceil_div64(n, 256)), dim3(256), 0, st, keys, v0, pos0, row_max, out_idx, out_val, Cout, M, n);
    return sonet::launched(what);
}

#ifdef SONET_VARIANTS   // (max over the neighbour planes from the layer epilogue: measured slower than layer + planes_max; variants build only)
namespace {
__global__ __launch_bounds__(256) void kmax_decode_kernel(const unsigned *__restrict__ keys, float *__restrict__ out, long long n)
{
    const long long t = (long long)blockIdx.x * 256 + threadIdx.x;
    if (t >= n) return;
    const unsigned k = keys[t];
    out[t] = k == 0u ? -__builtin_inff() : __uint_as_float((k & 0x80000000u) ? (k ^ 0x80000000u) : ~k);
}
}  // namespace

/* out[b][o][m] = max over the columns l with l % M == m of act((W . cat(x1, x2)) * scale + shift)[b][o][l]: the layer followed by
 * the max over the K neighbour planes of a k-major B x Cout x (K * M) tensor (KNNModule, models/layers.py:340-350) without
 * materialising it.  keys_ws: B * Cout * M * 4 bytes.  fp16-split arithmetic, Cout % 128 == 0. */
extern "C" int sonet_pointmlp_h3_kmax_f32(const float *x1, int C1, const float *x2, int C2, const void *Wp3, const float *scale,
                                          const float *shift, int relu, float *out, void *keys_ws, int B, int Cout, int L, int M,
                                          sonet_stream_t stream)
{
    const char *what = "sonet_pointmlp_h3_kmax_f32";
    SONET_REQUIRE(out && keys_ws && M > 0 && L % M == 0, "%s: NULL pointer or L %% M != 0", what);
    const size_t n = (size_t)B * Cout * M;
    if (hipMemsetAsync(keys_ws, 0, n * 4, sonet::as_stream(stream)) != hipSuccess) return sonet::fail(SONET_ERR_LAUNCH, "%s: memset failed", what);
    const int rc = x3_run_impl(what, true, x1, C1, x2, C2, Wp3, scale, shift, relu, nullptr, B, Cout, L, stream, nullptr, 0,
                               nullptr, nullptr, nullptr, nullptr, nullptr, 0, reinterpret_cast<unsigned *>(keys_ws), M);
    if (rc != SONET_OK) return rc;
    hipLaunchKernelGGL(kmax_decode_kernel, dim3((unsigned)sonet::ceil_div64((long long)n, 256)), dim3(256), 0, sonet::as_stream(stream),
                       reinterpret_cast<const unsigned *>(keys_ws), out, (long long)n);
    return sonet::launched(what);
}
#endif  // SONET_VARIANTS

extern "C" size_t sonet_pointmlp_stats_ws_size(int B, int Cout, int L)
{
    if (B <= 0 || Cout <= 0 || L <= 0) return 0;
    return (size_t)sonet::ceil_div64((long long)B * sonet::ceil_div(L, 32), X3_WAVES) * Cout * 2 * sizeof(double);
}

/* sonet_pointmlp_{h3,x3}_f32 that also return the per-channel mean and biased variance of y over (B, L) -- BatchNorm's batch
 * statistics (models/layers.py:60-70) -- from the kernel's epilogue.  stats_ws: sonet_pointmlp_stats_ws_size bytes. */
extern "C" int sonet_pointmlp_h3_stats_f32(const float *x1, int C1, const float *x2, int C2, const void *Wp3, const float *scale,
                                           const float *shift, int relu, float *y, int B, int Cout, int L, void *stats_ws,
                                           float *mean, float *var, sonet_stream_t stream)
{
    SONET_REQUIRE(stats_ws && mean && var, "sonet_pointmlp_h3_stats_f32: NULL pointer");
    return x3_run_impl("sonet_pointmlp_h3_stats_f32", true, x1, C1, x2, C2, Wp3, scale, shift, relu, y, B, Cout, L, stream, nullptr, 0,
                       reinterpret_cast<double *>(stats_ws), mean, var);
}

/* sonet_pointmlp_h3_stats_f32 on inputs that are the RAW outputs of BatchNorm layers: x = act(raw * xs[c] + xh[c]) is applied by the operand
 * load (xs1 / xh1 [C1], xs2 / xh2 [C2] or NULL without x2; xrelu bit 0 / 1: ReLU on x1's / x2's channels), as sonet_channel_affine_act_f32
 * would have computed it.  Cin <= 1024, Cout % 128 == 0. */
extern "C" int sonet_pointmlp_h3_stats_xaff_f32(const float *x1, int C1, const float *x2, int C2, const void *Wp3, const float *scale,
                                                const float *shift, int relu, float *y, int B, int Cout, int L, void *stats_ws,
                                                float *mean, float *var, const float *xs1, const float *xh1, const float *xs2, const float *xh2,
                                                int xrelu, sonet_stream_t stream)
{
    SONET_REQUIRE(stats_ws && mean && var && xs1 && xh1, "sonet_pointmlp_h3_stats_xaff_f32: NULL pointer");
    const XAffArgs xa = {xs1, xh1, xs2, xh2, xrelu};
    return x3_run_impl("sonet_pointmlp_h3_stats_xaff_f32", true, x1, C1, x2, C2, Wp3, scale, shift, relu, y, B, Cout, L, stream, nullptr, 0,
                       reinterpret_cast<double *>(stats_ws), mean, var, nullptr, nullptr, 0, nullptr, 0, nullptr, &xa);
}

extern "C" int sonet_pointmlp_x3_stats_f32(const float *x1, int C1, const float *x2, int C2, const void *Wp3, const float *scale,
                                           const float *shift, int relu, float *y, int B, int Cout, int L, void *stats_ws,
                                           float *mean, float *var, sonet_stream_t stream)
{
    SONET_REQUIRE(stats_ws && mean && var, "sonet_pointmlp_x3_stats_f32: NULL pointer");
    return x3_run_impl("sonet_pointmlp_x3_stats_f32", false, x1, C1, x2, C2, Wp3, scale, shift, relu, y, B, Cout, L, stream, nullptr, 0,
                       reinterpret_cast<double *>(stats_ws), mean, var);
}

/* The input gradient of a layer behind a training-mode BatchNorm (+ ReLU) with the BatchNorm / ReLU backward applied by the operand load:
 * y = (W . g_raw) * scale + shift with g_raw[k] = a[k] * (relu && !(raw * sc[k] + sh[k] > 0) ? 0 : gy) + b[k] * raw + c0[k]  (gy, raw [B][C][L],
 * the coefficients [C]: sonet_bn_bwd_coeffs_f32's a, b, c0 and the forward's normalisation sc, sh) -- what sonet_pointwise_bwd_apply_f32 +
 * sonet_pointmlp_x3_f32 compute, bit for bit, in one pass over (gy, raw); g_raw_out (or NULL) receives g_raw for the weight gradient.
 * Wp3: the bf16-split pack (sonet_pointmlp_x3_pack*) of the C x Cout matrix; C <= 512, Cout % 32 == 0. */
extern "C" int sonet_pointmlp_x3_bnb_f32(const float *gy, const float *raw, int C, const void *Wp3, const float *scale, const float *shift,
                                         const float *a, const float *b, const float *c0, const float *sc, const float *sh, int relu,
                                         float *g_raw_out, float *y, int B, int Cout, int L,
                                         const float *praw, const float *psc, const float *psh, int prelu, void *pstats_ws, double *psums,
                                         sonet_stream_t stream)
{
    const char *what = "sonet_pointmlp_x3_bnb_f32";
    SONET_REQUIRE(gy && raw && a && b && c0 && sc && sh && y, "%s: NULL pointer", what);
#ifndef SONET_VARIANTS
    // (the epilogue that also computes the BatchNorm-backward sums of the layer below measured slower than the pass it replaces -- docs/findings.md
    //  R5.9 -- and is compiled into the variants build only)
    if (praw || pstats_ws || psums)
        return sonet::fail(SONET_ERR_UNSUPPORTED, "%s: the sums of the layer below (praw / psums) are a variants-build record, pass NULL", what);
#endif
    const BnbArgs bn = {raw, a, b, c0, sc, sh, g_raw_out, relu, praw, psc, psh, prelu, reinterpret_cast<double *>(pstats_ws), nullptr};
    return x3_run_impl(what, false, gy, C, nullptr, 0, Wp3, scale, shift, 0, y, B, Cout, L, stream, nullptr, 0, nullptr, nullptr, nullptr,
                       nullptr, nullptr, 0, nullptr, 0, nullptr, nullptr, &bn, psums);
}

/* sonet_pointmlp_x3_bnb_f32 with the store accumulating: y = (W . g_raw) * scale + shift + yadd  (yadd [B][Cout][L]: another gradient of the
 * same tensor, computed earlier; yadd == y is allowed -- every element is read and written by one lane).  The f32 sum is the one autograd's
 * accumulation of the two gradients would store. */
extern "C" int sonet_pointmlp_x3_bnb_acc_f32(const float *gy, const float *raw, int C, const void *Wp3, const float *scale, const float *shift,
                                             const float *a, const float *b, const float *c0, const float *sc, const float *sh, int relu,
                                             float *g_raw_out, const float *yadd, float *y, int B, int Cout, int L, sonet_stream_t stream)
{
    const char *what = "sonet_pointmlp_x3_bnb_acc_f32";
    SONET_REQUIRE(gy && raw && a && b && c0 && sc && sh && y && yadd, "%s: NULL pointer", what);
    const BnbArgs bn = {raw, a, b, c0, sc, sh, g_raw_out, relu, nullptr, nullptr, nullptr, 0, nullptr, yadd};
    return x3_run_impl(what, false, gy, C, nullptr, 0, Wp3, scale, shift, 0, y, B, Cout, L, stream, nullptr, 0, nullptr, nullptr, nullptr,
                       nullptr, nullptr, 0, nullptr, 0, nullptr, nullptr, &bn, nullptr);
}

extern "C" int sonet_pointmlp_x3_f32(const float *x1, int C1, const float *x2, int C2, const void *Wp3,
                                     const float *scale, const float *shift, int relu, float *y,
                                     int B, int Cout, int L, sonet_stream_t stream)
{
    return x3_run_impl("sonet_pointmlp_x3_f32", false, x1, C1, x2, C2, Wp3, scale, shift, relu, y, B, Cout, L, stream);
}

extern "C" int sonet_pointmlp_h3_f32(const float *x1, int C1, const float *x2, int C2, const void *Wp3,
                                     const float *scale, const float *shift, int relu, float *y,
                                     int B, int Cout, int L, sonet_stream_t stream)
{
    return x3_run_impl("sonet_pointmlp_h3_f32", true, x1, C1, x2, C2, Wp3, scale, shift, relu, y, B, Cout, L, stream);
}


// The same layer with x1 read through a per-column gather index: x1 is [B][C1][L1], column l of cloud b takes
// x1[b][:, gidx[b][l]] (zeros when the index is outside [0, L1)); x2 [B][C2][L] and y [B][Cout][L] as usual.
extern "C" int sonet_pointmlp_h3_gather_f32(const float *x1, int C1, int L1, const int32_t *gidx, const float *x2, int C2, const void *Wp3,
                                            const float *scale, const float *shift, int relu, float *y,
                                            int B, int Cout, int L, sonet_stream_t stream)
{
    const char *what = "sonet_pointmlp_h3_gather_f32";
    SONET_REQUIRE(gidx, "%s: NULL pointer", what);
    return x3_run_impl(what, true, x1, C1, x2, C2, Wp3, scale, shift, relu, y, B, Cout, L, stream, gidx, L1);
}
