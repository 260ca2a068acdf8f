// pointmlp_bf16.hip -- the fused point-wise layer with bf16 STORAGE and bf16 MFMA (BASELINE configs[1]: "bf16").
//
//   y[b][o][l] = act( (sum_i W[o][i] * xcat[b][i][l]) * scale[o] + shift[o] ),   x, y: bfloat16 [B][C][L]; f32 accumulate
//
// Same operator as pointmlp.hip / pointmlp_x3.hip (models/layers.py:282-296 with the BatchNorm of :60-70 folded into
// (scale, shift)); the reference computes in f32 only, this is the reduced-precision twin SURVEY.md 7 (step 5) schedules:
// ONE v_mfma_f32_32x32x16_bf16 per product (the f32-class paths issue 3 or 6), half the activation bytes.  At 2.5 PFLOP/s
// dense the 320 -> 384 layer has 175 flop per HBM byte, under the machine balance of ~310: the layer-wise kernel is
// HBM-bound again, so it is built around bytes: dword accesses only, every A fragment used twice.
//
// Wave tile = 64 points x (MT x 32) output channels.  The two 32-column MFMA tiles of a wave are the EVEN and the ODD
// points of its 64: lane j reads ONE dword (points 2j, 2j+1) per channel row -- 128 contiguous bytes per half-wave and
// row, exactly the f32 kernels' access pattern at half the bytes -- and a v_perm_b32 pair per two channels sorts the
// halves into the two B fragments (8 VALU per 16-channel chunk; the f32-class split costs 44-52).  On the way out the
// accumulators of the two tiles meet again: v_cvt_pk_bf16_f32(acc_even, acc_odd) is the dword to store.  W goes through
// LDS in stages shared by the 4 waves as in pointmlp_x3.hip (one 1-KiB slice per (cout tile, K chunk)); each slice read
// from LDS feeds two MFMAs, which keeps the LDS at half its bandwidth when the matrix pipe is full.
// Odd L (rows not dword aligned) and the gather variant (columns picked by an index) take 2-byte loads / stores instead.
#include "common.hpp"
#include <stdlib.h>

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef int i32x4_t __attribute__((ext_vector_type(4)));

constexpr int BF_THREADS = 256;
constexpr int BF_WAVES = 4;
#ifdef SONET_VARIANTS
struct T1 { static constexpr bool value = true; };
struct T0 { static constexpr bool value = false; };
#endif

__device__ __forceinline__ unsigned cvt_pk_bf16(float lo, float hi) {       // round to nearest even, NaN stays NaN
    unsigned r;
    asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(r) : "v"(lo), "v"(hi));
    return r;
}

// Wp[ct][kc][lane] (uint4 = 8 bf16):  W[ct*32 + (lane&31)][kc*16 + 8*(lane>>5) + t], t = 0..7, zero padded
__global__ __launch_bounds__(256) void bf16_pack_kernel(const float *__restrict__ W, uint4 *__restrict__ Wp, int Cin, int Cout, int KC, long long total,
                                                         long long rs /*element (o, c) = W[o * rs + c * cs]*/, long long cs)
{
    const long long t = (long long)blockIdx.x * 256 + threadIdx.x;
    if (t >= total) return;
    const int lane = (int)(t & 63);
    const long long r = t >> 6;
    const int kc = (int)(r % KC), ct = (int)(r / KC);
    const int o = ct * 32 + (lane & 31);
    const int c0 = kc * 16 + 8 * (lane >> 5);
    unsigned w[4];
#pragma unroll
    for (int p = 0; p < 4; ++p) {
        const int c = c0 + 2 * p;
        const float w0 = (o < Cout && c < Cin) ? W[(long long)o * rs + (long long)c * cs] : 0.f;
        const float w1 = (o < Cout && c + 1 < Cin) ? W[(long long)o * rs + (long long)(c + 1) * cs] : 0.f;
        w[p] = cvt_pk_bf16(w0, w1);
    }
    Wp[t] = make_uint4(w[0], w[1], w[2], w[3]);
}

// (BNB) BatchNorm / ReLU backward on the operand load: x1 holds gy and the operand of the product is
//   g_raw[k] = bf16( a[k] * (relu && !(raw * sc[k] + sh[k] > 0) ? 0 : gy) + b[k] * raw + c0[k] )
// -- sonet_pointwise_bwd_apply_bf16's arithmetic and rounding, element for element -- computed on the chunk's registers in front of the
// MFMAs; the first pass of the first output slab also stores it (g_raw_out, for the weight gradient).  PAIRED, C1 % 16 == 0, C1 <= BNB_CMAX.
struct BfBnb { const uint16_t *raw; const float *a, *b, *c0, *sc, *sh; uint16_t *g_raw_out; int relu; };
constexpr int BNB_CMAX = 512;

// PAIRED: L even -> one dword per (lane, channel row) covers the lane's two points.  Otherwise 2-byte accesses with two
// independent column offsets per lane (odd L; the gather variant).
template <int MT, int S, bool PAIRED, int NXB = 2, bool BNB = false>
__global__ __launch_bounds__(BF_THREADS, (MT <= 4 ? 2 : 1)) void pointmlp_bf16_kernel(      // <= 4 tiles: two workgroups per CU (<= 256 VGPRs)
    const uint16_t *__restrict__ x1, int C1, const uint16_t *__restrict__ x2, int C2, const uint4 *__restrict__ Wp,
    const float *__restrict__ scale, const float *__restrict__ shift, int relu, uint16_t *__restrict__ y,
    int Cout, int L, int gpc /*64-column groups per cloud*/, long long ngroups, int CT, int KC, int ct_per_y,
    const int32_t *__restrict__ gidx /*optional [B][L]: column l of x1 is x1[:, gidx[b][l]]*/, int L1 /*row length of x1*/,
    double *__restrict__ stats_partial /*optional [gridDim.x][Cout][2]: sum / sum of squares of the STORED (bf16) output over this workgroup's columns*/,
    const uint16_t *__restrict__ yadd = nullptr /*optional [B][Cout][L] (PAIRED, no statistics): y = bf16(float(bf16(result)) + float(yadd)) -- what
                                                  autograd's accumulation of two bf16 gradients of one tensor would store (models/layers.py _GradCarry)*/,
    const BfBnb bnb = BfBnb{})
{
    static_assert(!BNB || (PAIRED && NXB == 2), "the BatchNorm-backward load exists on the paired two-buffer pipeline");
    constexpr int NSL = S * MT;                              // 1 KiB W slices per stage
    constexpr int NS = (NSL + BF_WAVES - 1) / BF_WAVES;
    __shared__ uint4 wsm[2][NS * BF_WAVES][64];
    __shared__ float2 affine[1024];
    __shared__ float2 red[BF_WAVES][MT * 32];                   // (statistics epilogue)
    __shared__ float4 bnb_abcs[BNB ? BNB_CMAX : 1];             // (BNB) (a, b, c0, sc) of input channel k
    __shared__ float bnb_sh[BNB ? BNB_CMAX : 1];                //       sh

    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int j = lane & 31, h = lane >> 5;

    long long q = (long long)blockIdx.x * BF_WAVES + wave;
    const bool wave_valid = q < ngroups;
    q = wave_valid ? q : 0;
    const long long b = q / gpc;
    const int l0 = (int)(q - b * gpc) * 64;
    const int ca = l0 + 2 * j, cb = ca + 1;                   // the lane's two columns (even tile, odd tile)
    const bool pva = wave_valid && ca < L, pvb = wave_valid && cb < L;
    const int cca = ca < L ? ca : l0, ccb = cb < L ? cb : cca;   // clamped: padded lanes re-read a valid column

    const unsigned rowB = (unsigned)L * 2u, rowB1 = (unsigned)L1 * 2u;
    const __amdgpu_buffer_rsrc_t r1 = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<uint16_t *>(x1 + b * (long long)C1 * L1), 0, (int)((unsigned)C1 * rowB1), 0x00020000);
    const __amdgpu_buffer_rsrc_t r2 = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<uint16_t *>(x2 ? x2 + b * (long long)C2 * L : x1), 0, (int)((unsigned)(x2 ? C2 : 0) * rowB), 0x00020000);
    const __amdgpu_buffer_rsrc_t ry = __builtin_amdgcn_make_buffer_rsrc(
        y + b * (long long)Cout * L, 0, (int)((unsigned)Cout * rowB), 0x00020000);
    const __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<uint4 *>(Wp), 0, (int)((unsigned)CT * (unsigned)KC * 1024u), 0x00020000);
    // lane byte offsets inside a 16-channel chunk (rows 8h .. 8h+7 of the chunk)
    const unsigned voa = (unsigned)(8 * h * L + cca) * 2u, vob = (unsigned)(8 * h * L + ccb) * 2u;
    unsigned voa1 = voa, vob1 = vob;                          // ... of x1 (through the gather index when there is one)
    if (gidx) {
        const int sa = gidx[b * L + cca], sb = gidx[b * L + ccb];
        voa1 = (unsigned)sa < (unsigned)L1 ? (unsigned)(8 * h * L1 + sa) * 2u : 0x7FFFFF00u;   // out of range: zeros
        vob1 = (unsigned)sb < (unsigned)L1 ? (unsigned)(8 * h * L1 + sb) * 2u : 0x7FFFFF00u;
    }
    const unsigned voya = (unsigned)(4 * h * L + cca) * 2u, voyb = (unsigned)(4 * h * L + ccb) * 2u;
    const unsigned vow = (unsigned)lane * 16u;

    const int KC1 = C2 > 0 ? (C1 >> 4) : KC;                  // chunks fed by x1 (C1 % 16 == 0 when x2 exists)
#ifdef SONET_VARIANTS   // ablation bits ride in the upper bits of ``relu`` (tools only): 1 no stores, 2 no K loop, 4 stores into four rows
    const int abl = relu >> 1;
    relu &= 1;
    const int nstage = (abl & 2) ? 0 : (KC + S - 1) / S;
#else
    const int nstage = (KC + S - 1) / S;
#endif
    constexpr int NR = PAIRED ? 8 : 16;                       // raw registers per chunk

    auto load_b = [&](unsigned (&raw)[S][NR], int st) {
#pragma unroll
        for (int i = 0; i < S; ++i) {
            const int kc = st * S + i;
            const bool second = kc >= KC1;
            const unsigned rb = second ? rowB : rowB1;
            const unsigned row0 = (unsigned)(16 * (second ? kc - KC1 : kc)) * rb;
#pragma unroll
            for (int t = 0; t < 8; ++t) {
                const unsigned so = row0 + (unsigned)t * rb;
                if constexpr (PAIRED) {
                    raw[i][t] = (unsigned)(second ? __builtin_amdgcn_raw_buffer_load_b32(r2, voa, so, 0)
                                                  : __builtin_amdgcn_raw_buffer_load_b32(r1, voa1, so, 0));
                } else {
                    raw[i][2 * t] = (unsigned)(unsigned short)(second ? __builtin_amdgcn_raw_buffer_load_b16(r2, voa, so, 0)
                                                                      : __builtin_amdgcn_raw_buffer_load_b16(r1, voa1, so, 0));
                    raw[i][2 * t + 1] = (unsigned)(unsigned short)(second ? __builtin_amdgcn_raw_buffer_load_b16(r2, vob, so, 0)
                                                                          : __builtin_amdgcn_raw_buffer_load_b16(r1, vob1, so, 0));
                }
            }
        }
    };

    // (BNB) the raw pre-activations beside gy: the same rows, the same lane offsets
    const __amdgpu_buffer_rsrc_t rr_ = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<uint16_t *>(BNB ? bnb.raw + b * (long long)C1 * L : x1), 0, (int)((unsigned)(BNB ? C1 : 0) * rowB), 0x00020000);
    const __amdgpu_buffer_rsrc_t rg_ = __builtin_amdgcn_make_buffer_rsrc(
        (BNB && bnb.g_raw_out) ? bnb.g_raw_out + b * (long long)C1 * L : y, 0, (int)((unsigned)((BNB && bnb.g_raw_out) ? C1 : 0) * rowB), 0x00020000);
    auto load_r = [&](unsigned (&rw_)[S][NR], int st) {
        if constexpr (BNB) {
#pragma unroll
            for (int i = 0; i < S; ++i) {
                const unsigned row0 = (unsigned)(16 * (st * S + i)) * rowB;
#pragma unroll
                for (int t = 0; t < 8; ++t) rw_[i][t] = (unsigned)__builtin_amdgcn_raw_buffer_load_b32(rr_, voa, row0 + (unsigned)t * rowB, 0);
            }
        }
    };

    const int ct_begin = blockIdx.y * ct_per_y;
    const int ct_end = min(CT, ct_begin + ct_per_y);
    for (int o = ct_begin * 32 + (int)threadIdx.x; o < ct_end * 32; o += BF_THREADS)
        affine[o - ct_begin * 32] = make_float2(scale[o], shift[o]);
    if constexpr (BNB) {
        for (int k = threadIdx.x; k < BNB_CMAX; k += BF_THREADS) {
            const bool in = k < C1;
            bnb_abcs[k] = in ? make_float4(bnb.a[k], bnb.b[k], bnb.c0[k], bnb.sc[k]) : make_float4(0.f, 0.f, 0.f, 0.f);
            bnb_sh[k] = in ? bnb.sh[k] : 0.f;
        }
    }

    for (int ct0 = ct_begin; ct0 < ct_end; ct0 += MT) {
        f32x16 acc[MT][2];
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int r = 0; r < 16; ++r) { acc[mt][0][r] = 0.f; acc[mt][1][r] = 0.f; }

        // slice sl of a stage: chunk i = sl / MT, cout tile mt = sl % MT
        auto stage_load = [&](i32x4_t (&w)[NS], int st) {
#pragma unroll
            for (int t = 0; t < NS; ++t) {
                int sl = wave + t * BF_WAVES;
                sl = sl < NSL ? sl : NSL - 1;
                const int i = sl / MT, mt = sl - i * MT;
                int kc = st * S + i;
                kc = kc < KC ? kc : KC - 1;
                w[t] = __builtin_amdgcn_raw_buffer_load_b128(rw, vow, (unsigned)((ct0 + mt) * KC + kc) * 1024u, 0);
            }
        };
        auto stage_write = [&](const i32x4_t (&w)[NS], int slot) {
#pragma unroll
            for (int t = 0; t < NS; ++t)
                wsm[slot][wave + t * BF_WAVES][lane] = __builtin_bit_cast(uint4, w[t]);
        };
        auto compute = [&](unsigned (&raw)[S][NR], const unsigned (&rawr)[S][NR], int slot, int st) {
#pragma unroll
            for (int i = 0; i < S; ++i) {
                if (st * S + i >= KC) break;                  // (a padded last stage: its W slices repeat the last chunk)
                if constexpr (BNB) {
                    // gy -> g_raw on the chunk's eight dwords (channel rows 16 kc + 8 h + t; the lane's even | odd point)
                    const int k0 = 16 * (st * S + i) + 8 * h;
                    const bool st_out = bnb.g_raw_out != nullptr && blockIdx.y == 0 && ct0 == ct_begin && pva;
                    const unsigned row0 = (unsigned)(16 * (st * S + i)) * rowB;
#pragma unroll
                    for (int t = 0; t < 8; ++t) {
                        const float4 co = bnb_abcs[k0 + t];
                        const float shv = bnb_sh[k0 + t];
                        const unsigned dg = raw[i][t], dr = rawr[i][t];
                        const float r0 = __uint_as_float(dr << 16), r1 = __uint_as_float(dr & 0xFFFF0000u);
                        float g0 = __uint_as_float(dg << 16), g1 = __uint_as_float(dg & 0xFFFF0000u);
                        if (bnb.relu) {
                            g0 = (__fmaf_rn(r0, co.w, shv) > 0.f) ? g0 : 0.f;
                            g1 = (__fmaf_rn(r1, co.w, shv) > 0.f) ? g1 : 0.f;
                        }
                        const unsigned pk = cvt_pk_bf16(__fmaf_rn(co.x, g0, __fmaf_rn(co.y, r0, co.z)), __fmaf_rn(co.x, g1, __fmaf_rn(co.y, r1, co.z)));
                        raw[i][t] = pk;
                        if (st_out) __builtin_amdgcn_raw_buffer_store_b32((int)pk, rg_, voa, row0 + (unsigned)t * rowB, 0);
                    }
                }
                unsigned ba[4], bb[4];
#pragma unroll
                for (int p = 0; p < 4; ++p) {
                    if constexpr (PAIRED) {                   // channel rows 2p, 2p+1: (even, odd) point halves of two dwords
                        ba[p] = __builtin_amdgcn_perm(raw[i][2 * p + 1], raw[i][2 * p], 0x05040100u);
                        bb[p] = __builtin_amdgcn_perm(raw[i][2 * p + 1], raw[i][2 * p], 0x07060302u);
                    } else {
                        ba[p] = raw[i][4 * p] | (raw[i][4 * p + 2] << 16);
                        bb[p] = raw[i][4 * p + 1] | (raw[i][4 * p + 3] << 16);
                    }
                }
                const bf16x8 Ba = __builtin_bit_cast(bf16x8, make_uint4(ba[0], ba[1], ba[2], ba[3]));
                const bf16x8 Bb = __builtin_bit_cast(bf16x8, make_uint4(bb[0], bb[1], bb[2], bb[3]));
#pragma unroll
                for (int mt = 0; mt < MT; ++mt) {
                    const bf16x8 A = __builtin_bit_cast(bf16x8, wsm[slot][i * MT + mt][lane]);
                    acc[mt][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A, Ba, acc[mt][0], 0, 0, 0);
                    acc[mt][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A, Bb, acc[mt][1], 0, 0, 0);
                }
            }
        };

        i32x4_t wreg[NS];
        unsigned b0[S][NR], b1[S][NR];
        unsigned q0[BNB ? S : 1][NR], q1[BNB ? S : 1][NR];          // (BNB) the raw pre-activations of the same chunks
        __syncthreads();
        if constexpr (NXB == 2) {
            stage_load(wreg, 0);
            load_b(b0, 0);
            if constexpr (BNB) load_r(q0, 0);
            stage_write(wreg, 0);
            stage_load(wreg, nstage > 1 ? 1 : 0);
#define BF_STAGE(st, bcur, bnxt, qcur, qnxt, slot)                            \
        {                                                                    \
            __syncthreads();                                                 \
            stage_write(wreg, (slot) ^ 1);                                   \
            stage_load(wreg, (st) + 2 < nstage ? (st) + 2 : nstage - 1);     \
            load_b(bnxt, (st) + 1);                                          \
            if constexpr (BNB) { load_r(qnxt, (st) + 1); compute(bcur, qcur, slot, st); } \
            else compute(bcur, bcur, slot, st);                              \
        }
            int st = 0;
            for (; st + 2 <= nstage; st += 2) {
                BF_STAGE(st, b0, b1, q0, q1, 0)
                BF_STAGE(st + 1, b1, b0, q1, q0, 1)
            }
            if (st < nstage) BF_STAGE(st, b0, b1, q0, q1, 0)
#undef BF_STAGE
        } else {
            // X AND W two stages ahead (three X register sets, two W sets).  One stage ahead, every stage top waited for the W slices
            // requested a stage (~0.25 us of MFMA issue) earlier: an L2 round trip per stage.  Issue order per stage: W(st + 3),
            // X(st + 2); the prologue leaves the same queue behind (hipcc merges the wait-counter states of loop entry and back edge)
            unsigned b2[S][NR];
            i32x4_t wreg1[NS];
            auto clampst = [&](int t) { return t < nstage ? t : nstage - 1; };
            stage_load(wreg1, 0);
            stage_load(wreg, clampst(1));
            load_b(b0, 0);
            stage_write(wreg1, 0);
            __builtin_amdgcn_sched_barrier(0);
            stage_load(wreg1, clampst(2));
            __builtin_amdgcn_sched_barrier(0);
            load_b(b1, clampst(1));
            __builtin_amdgcn_sched_barrier(0);
#define BF_STAGE3(st, bcur, bfar, wset)                                       \
        {                                                                    \
            const int slot_ = (st) & 1;                                      \
            __syncthreads();                                                 \
            stage_write(wset, slot_ ^ 1);                                    \
            stage_load(wset, clampst((st) + 3));                             \
            load_b(bfar, clampst((st) + 2));     /* (unconditional: a branch here makes hipcc wait for vmcnt(0)) */ \
            compute(bcur, bcur, slot_, st);                                  \
        }
            int st = 0;
            for (; st + 6 <= nstage; st += 6) {
                BF_STAGE3(st, b0, b2, wreg)
                BF_STAGE3(st + 1, b1, b0, wreg1)
                BF_STAGE3(st + 2, b2, b1, wreg)
                BF_STAGE3(st + 3, b0, b2, wreg1)
                BF_STAGE3(st + 4, b1, b0, wreg)
                BF_STAGE3(st + 5, b2, b1, wreg1)
            }
            if (st < nstage) BF_STAGE3(st, b0, b2, wreg)
            if (st + 1 < nstage) BF_STAGE3(st + 1, b1, b0, wreg1)
            if (st + 2 < nstage) BF_STAGE3(st + 2, b2, b1, wreg)
            if (st + 3 < nstage) BF_STAGE3(st + 3, b0, b2, wreg1)
            if (st + 4 < nstage) BF_STAGE3(st + 4, b1, b0, wreg)
#undef BF_STAGE3
        }

        if (stats_partial != nullptr) {
            // training forward: batch statistics of the STORED values (what the normalise pass and the backward read) from this
            // epilogue -- see pointmlp_x3.hip; all lanes active, padded columns store out of range and add 0
            const unsigned voya_s = pva ? voya : 0x7FFFFF00u, voyb_s = pvb ? voyb : 0x7FFFFF00u;
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) {
                const unsigned so_tile = (unsigned)((ct0 + mt) * 32) * rowB;
                const float2 *aff = affine + (ct0 + mt - ct_begin) * 32 + 4 * h;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int orow = (r & 3) + 8 * (r >> 2);
                    const float2 ss = aff[orow];
                    float va = __fmaf_rn(acc[mt][0][r], ss.x, ss.y), vb = __fmaf_rn(acc[mt][1][r], ss.x, ss.y);
                    if (relu) { va = (va < 0.f) ? 0.f : va; vb = (vb < 0.f) ? 0.f : vb; }
                    const unsigned pk = cvt_pk_bf16(va, vb);
                    const unsigned so = so_tile + (unsigned)orow * rowB;
                    if constexpr (PAIRED) {
                        __builtin_amdgcn_raw_buffer_store_b32((int)pk, ry, voya_s, so, 0);
                    } else {
                        __builtin_amdgcn_raw_buffer_store_b16((short)(pk & 0xFFFFu), ry, voya_s, so, 0);
                        __builtin_amdgcn_raw_buffer_store_b16((short)(pk >> 16), ry, voyb_s, so, 0);
                    }
                    const float ra = pva ? __uint_as_float(pk << 16) : 0.f, rb = pvb ? __uint_as_float(pk & 0xFFFF0000u) : 0.f;
                    const float s1 = row32_sum(ra + rb), s2 = row32_sum(__fmaf_rn(ra, ra, rb * rb));
                    if (j == 0) red[wave][mt * 32 + orow + 4 * h] = make_float2(s1, s2);
                }
            }
            __syncthreads();
            for (int t = threadIdx.x; t < MT * 32; t += BF_THREADS) {
                const double a = ((double)red[0][t].x + (double)red[1][t].x) + ((double)red[2][t].x + (double)red[3][t].x);
                const double qq = ((double)red[0][t].y + (double)red[1][t].y) + ((double)red[2][t].y + (double)red[3][t].y);
                double *dst = stats_partial + ((size_t)blockIdx.x * Cout + (size_t)ct0 * 32 + t) * 2;
                dst[0] = a;
                dst[1] = qq;
            }
            __syncthreads();
        } else if (pva) {
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) {
                const unsigned so_tile = (unsigned)((ct0 + mt) * 32) * rowB;
                const float2 *aff = affine + (ct0 + mt - ct_begin) * 32 + 4 * h;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int orow = (r & 3) + 8 * (r >> 2);
                    const float2 ss = aff[orow];
                    float va = __fmaf_rn(acc[mt][0][r], ss.x, ss.y), vb = __fmaf_rn(acc[mt][1][r], ss.x, ss.y);
                    if (relu) { va = (va < 0.f) ? 0.f : va; vb = (vb < 0.f) ? 0.f : vb; }     // NaN propagates
                    const unsigned pk = cvt_pk_bf16(va, vb);
#ifdef SONET_VARIANTS
                    const unsigned so = (abl & 4) ? (unsigned)(orow & 3) * rowB : so_tile + (unsigned)orow * rowB;
                    if (abl & 1) { asm volatile("" :: "v"(pk)); continue; }
#else
                    const unsigned so = so_tile + (unsigned)orow * rowB;
#endif
                    if constexpr (PAIRED) {
                        if (yadd != nullptr) {
                            const __amdgpu_buffer_rsrc_t ra_ = __builtin_amdgcn_make_buffer_rsrc(
                                const_cast<uint16_t *>(yadd + b * (long long)Cout * L), 0, (int)((unsigned)Cout * rowB), 0x00020000);
                            const unsigned ad = (unsigned)__builtin_amdgcn_raw_buffer_load_b32(ra_, voya, so, 0);
                            const unsigned sum = cvt_pk_bf16(__uint_as_float(pk << 16) + __uint_as_float(ad << 16),
                                                             __uint_as_float(pk & 0xFFFF0000u) + __uint_as_float(ad & 0xFFFF0000u));
                            __builtin_amdgcn_raw_buffer_store_b32((int)sum, ry, voya, so, 0);
                            continue;
                        }
                        __builtin_amdgcn_raw_buffer_store_b32((int)pk, ry, voya, so, 0);
                    } else {
                        __builtin_amdgcn_raw_buffer_store_b16((short)(pk & 0xFFFFu), ry, voya, so, 0);
                        if (pvb) __builtin_amdgcn_raw_buffer_store_b16((short)(pk >> 16), ry, voyb, so, 0);
                    }
                }
            }
        }
    }
}

// ---- the streaming generation: W resident in LDS, persistent waves, X requests that never go cold -----------------------------
// Why: on the first PointNet's layers (64 x 15000 columns) the kernel above spends a constant ~10 us per (workgroup, pass) whatever
// the K extent -- a cold X request at the head of every pass, the drain of its 64 dword stores at the end, two workgroups per CU to
// hide both (profiles/r04s_bf16_layers.log: 1.6 TB/s of writes in every shape; without stores AND without the K loop the launch
// still takes 0.15 ms).  Here a workgroup of 8 waves (2 per SIMD, one workgroup per CU) copies its slab of W -- tps cout tiles x all
// K chunks, <= 144 KiB -- into the LDS ONCE and never meets a barrier again; every wave walks its own sequence of (64-column group,
// pass of MT tiles) units, and its X requests run a fixed 4 chunks (32 dword loads, 8 KiB) ahead of the MFMAs ACROSS unit and group
// boundaries, so the stores of a unit's epilogue are in flight next to the next unit's loads.  All X loads are inline asm with
// hand-counted s_waitcnt vmcnt (the counter is in order and counts stores: 24 = three chunks behind the one consumed; + the 16 MT
// stores of an epilogue for the first four chunks after one, capped at 63).
typedef int i32x4_t_ __attribute__((ext_vector_type(4)));

struct BfrXaff { const float *xs1, *xh1, *xs2, *xh2; int xrelu; };     // normalise-on-load coefficients (see BfrArgs)

struct BfrArgs {
    const uint16_t *x1, *x2;
    const uint4 *Wp;
    const float *scale, *shift;
    uint16_t *y;
    double *stats_partial;                // [nstream][Cout][2] (STATS)
    int C1, C2, Cout, L, gpc, relu, KC, KC1, tps /*cout tiles per slab*/, nslab, nstream /*column streams = workgroups per slab*/;
    int ngroups;                          // B * gpc
    int sync;                             // workgroup barriers (rows that are not cache-line aligned): 1 per column group, 2 + per four chunks
    // (POOL) the layer's output is never stored: a workgroup = (cloud, output slab) keeps the per-node arg-max bins of its rows in LDS
    // (models/networks.py:180-185: index_max + the masked gather), ids = node of every column [B][L] i32, M <= 256 nodes
    const int32_t *ids, *row_max;         // row_max [B][M] (NULL: nothing masked)
    int32_t *out_idx;                     // [B][Cout][M] winning column (0 where nothing beat -1000, as the reference)
    float *out_val;                       // [B][Cout][M] the stored (bf16) value at out_idx * row_max
    int M;
    int abl;                              // (variants build, POOL) 1: no epilogue at all, 2: no bin reads / updates (the arithmetic stays)
    // (XAFF) normalise-on-load: x1 / x2 hold the RAW (bf16) outputs of BatchNorm layers; the operand is act(raw * xs[c] + xh[c]) rounded to
    // bf16 -- what sonet_channel_affine_act_bf16 would have stored, bit for bit -- computed on the chunk's registers in front of the MFMAs
    const float *xs1, *xh1, *xs2, *xh2;   // [C1], [C2] (x2's may be NULL when C2 == 0)
    int xrelu;                            // bit 0: ReLU on x1, bit 1: on x2
    unsigned xco_off;                     // byte offset of the coefficient table in the dynamic LDS
};

// (POOL) the order of the keys is that of index_max.hip -- bigger value wins, equal values: the smaller column wins, -0 counts as +0, a NaN never
// wins, bins start at "value -1000 at column 0" -- so the launch reports exactly what sonet_index_max_gather_bf16 reports on the tensor the
// storing launch would have written.  The values are bf16 and a cloud has < 65536 columns: a key is 32 bits, (orderable(bf16) << 16) |
// (0xFFFF - column), a bin one LDS word, "this element beats the running maximum" one ds_read_b32 + one compare per value.
constexpr unsigned BFP_INIT_KEY = (0x3B85u << 16) | 0xFFFFu;                 // ord(bf16(-1000)) = ~0xC47A, column 0
static_assert((0xC47Au ^ 0xFFFFu) == 0x3B85u, "ord(-1000) check");
// the two orderable 16-bit patterns of a packed bf16 pair (lo | hi << 16), in place
__device__ __forceinline__ unsigned bfp_ord2(unsigned pk) {
    // -0 -> +0
    pk = (pk & 0xFFFFu) == 0x8000u ? (pk & 0xFFFF0000u) : pk;
    pk = (pk >> 16) == 0x8000u ? (pk & 0x0000FFFFu) : pk;
    // negative: all bits flipped; positive: the sign bit set
    const unsigned neg = (pk >> 15) & 0x00010001u;                          // 1 per negative half
    const unsigned o = pk ^ (neg * 0x7FFFu | 0x80008000u);
    // NaN -> 0 (never wins)
    const unsigned mlo = pk & 0x7FFFu, mhi = (pk >> 16) & 0x7FFFu;
    return (mlo > 0x7F80u ? 0u : (o & 0xFFFFu)) | (mhi > 0x7F80u ? 0u : (o & 0xFFFF0000u));
}

__device__ __forceinline__ i32x4_t_ bfr_rsrc(const void *base, unsigned bytes) {
    const unsigned long long a = reinterpret_cast<unsigned long long>(base);
    i32x4_t_ r;
    r[0] = __builtin_amdgcn_readfirstlane((int)(unsigned)a);
    r[1] = __builtin_amdgcn_readfirstlane((int)(unsigned)((a >> 32) & 0xFFFFu));      // stride 0: raw buffer
    r[2] = __builtin_amdgcn_readfirstlane((int)bytes);
    r[3] = 0x00020000;
    return r;
}
// "at most N vector-memory operations of this wave outstanding", then the chunk's eight dwords (channel rows 2p, 2p + 1: even | odd
// point) sorted into the two B fragments -- ONE statement, the ring registers plain inputs: a wait with the registers as in/out
// operands made hipcc hand it COPIES, taken before the wait (tools/check_bf16r_asm.py looks for any move out of a ring register).
template <int N>
__device__ __forceinline__ void bfr_wait_perm(const unsigned (&x)[8], unsigned (&ba)[4], unsigned (&bb)[4]) {
    asm volatile("s_waitcnt vmcnt(%16)\n\t"
                 "v_perm_b32 %0, %9, %8, %17\n\tv_perm_b32 %4, %9, %8, %18\n\t"
                 "v_perm_b32 %1, %11, %10, %17\n\tv_perm_b32 %5, %11, %10, %18\n\t"
                 "v_perm_b32 %2, %13, %12, %17\n\tv_perm_b32 %6, %13, %12, %18\n\t"
                 "v_perm_b32 %3, %15, %14, %17\n\tv_perm_b32 %7, %15, %14, %18"
                 : "=&v"(ba[0]), "=&v"(ba[1]), "=&v"(ba[2]), "=&v"(ba[3]), "=&v"(bb[0]), "=&v"(bb[1]), "=&v"(bb[2]), "=&v"(bb[3])
                 : "v"(x[0]), "v"(x[1]), "v"(x[2]), "v"(x[3]), "v"(x[4]), "v"(x[5]), "v"(x[6]), "v"(x[7]),
                   "n"(N), "s"(0x05040100u), "s"(0x07060302u)
                 : "memory");
}

// (XAFF) the four channel pairs of a B fragment (elements 2p, 2p + 1 of the chunk's half h, packed bf16) through act(raw * s + h): f32 fma,
// one round-to-nearest-even back to bf16, ReLU -- sonet_channel_affine_act_bf16's arithmetic per element (there: ReLU in f32 in front of the
// rounding; rounding is monotone and keeps the sign, so the order does not matter.  A -0.0 comes out as +0.0 here: as an operand of the
// product that is the same number).  Five vector instructions per pair: two unpacks, v_pk_fma_f32, v_cvt_pk_bf16_f32, v_pk_max_i16 against
// `floor` (0 with ReLU, the most negative i16 -- the identity -- without).  co[p] = (s of 2p, s of 2p + 1, h of 2p, h of 2p + 1).
typedef float bfr_f2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void bfr_xaff(unsigned (&b)[4], const float4 (&co)[4], unsigned floor2) {
#pragma unroll
    for (int p = 0; p < 4; ++p) {
        const bfr_f2 x = {__uint_as_float(b[p] << 16), __uint_as_float(b[p] & 0xFFFF0000u)};
        const bfr_f2 sc = {co[p].x, co[p].y}, sh = {co[p].z, co[p].w};
        bfr_f2 v;
        asm("v_pk_fma_f32 %0, %1, %2, %3" : "=v"(v) : "v"(x), "v"(sc), "v"(sh));
        unsigned r = cvt_pk_bf16(v[0], v[1]);
        asm("v_pk_max_i16 %0, %1, %2" : "=v"(r) : "v"(r), "v"(floor2));
        b[p] = r;
    }
}

template <int MT, bool STATS, bool POOL = false, bool XAFF = false>
__global__ __launch_bounds__(512, 1) void pointmlp_bf16r_kernel(const BfrArgs a)
{
    static_assert(!(STATS && POOL), "statistics belong to BatchNorm layers, the pool to the last (norm-free) layer");
    extern __shared__ uint4 bfr_lds[];
    uint4 *wl = bfr_lds;                                                       // [tps][KC][64]
    float2 *aff = reinterpret_cast<float2 *>(wl + (size_t)a.tps * a.KC * 64);  // [tps * 32]
    float2 *stl = aff + a.tps * 32;                                            // STATS: [8 waves][tps * 32] (sum, sum of squares)
    // POOL: bins [tps * 32][M] (one word each), the value at column 0 of every row (what a bin nothing beat gathers), the cloud's node ids as bytes
    unsigned *bins = reinterpret_cast<unsigned *>(aff + a.tps * 32);
    // ... and a float shadow of every bin's value (never above it): "can this element matter at all" is one LDS read and one float compare;
    // the orderable key, its compare and the LDS atomic are paid by the few that pass (about ln n per bin)
    float *shadow = reinterpret_cast<float *>(bins + (size_t)a.tps * 32 * (POOL ? a.M : 0));
    unsigned *v0s = reinterpret_cast<unsigned *>(shadow + (size_t)a.tps * 32 * (POOL ? a.M : 0));
    unsigned char *idb = reinterpret_cast<unsigned char *>(v0s + a.tps * 32);
    // (XAFF) [KC][2 halves][4] float4 = (s, s, h, h) of elements 2p, 2p + 1: behind everything else (the launch adds KC * 128 bytes)
    float4 *xco = reinterpret_cast<float4 *>(reinterpret_cast<unsigned char *>(bfr_lds) + a.xco_off);

    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int j = lane & 31, h = lane >> 5;
    // workgroup -> (slab, column stream): the nslab workgroups of one stream read the same X, they sit on the same XCD (ids 8 apart)
    const int xcd = blockIdx.x & 7, kq = blockIdx.x >> 3;
    const int slab = kq % a.nslab, stream = xcd + 8 * (kq / a.nslab);
    if constexpr (POOL) { if (stream * a.gpc >= a.ngroups) return; }      // (the stream is the cloud: the grid is rounded up to whole XCD rows)
    const int ct_begin = slab * a.tps;
    const int KC = a.KC, L = a.L;
    const unsigned rowB = (unsigned)L * 2u;

    {
        const uint4 *src = a.Wp + (size_t)ct_begin * KC * 64;
        const int n = a.tps * KC * 64;                          // (tps tiles x KC chunks x 64 lanes; KC a multiple of 4)
        int i0 = 0;
        for (; i0 + 512 * 8 <= n; i0 += 512 * 8) {              // eight requests per thread in flight, then the eight LDS writes
            const uint4 *sp = src + i0 + (int)threadIdx.x;
            uint4 *dp = wl + i0 + (int)threadIdx.x;
            const uint4 t0 = sp[0], t1 = sp[512], t2 = sp[1024], t3 = sp[1536], t4 = sp[2048], t5 = sp[2560], t6 = sp[3072], t7 = sp[3584];
            dp[0] = t0; dp[512] = t1; dp[1024] = t2; dp[1536] = t3; dp[2048] = t4; dp[2560] = t5; dp[3072] = t6; dp[3584] = t7;
        }
        for (; i0 < n; i0 += 512)                               // (n is a multiple of 256: a multiple of 512 only for an even slab)
            if (i0 + (int)threadIdx.x < n) wl[i0 + (int)threadIdx.x] = src[i0 + (int)threadIdx.x];
        for (int o = threadIdx.x; o < a.tps * 32; o += 512) aff[o] = make_float2(a.scale[ct_begin * 32 + o], a.shift[ct_begin * 32 + o]);
        if constexpr (STATS)
            for (int o = threadIdx.x; o < 8 * a.tps * 32; o += 512) stl[o] = make_float2(0.f, 0.f);
        if constexpr (XAFF) {
            // entry (kc, hh, p): channels 16 kc + 8 hh + 2 p, + 1 of x1 (kc < KC1) or x2; rows past the panel's channel count read 0 and
            // must stay 0 (their weights are 0 as well, but act(0 * s + h) need not be)
            for (int o = threadIdx.x; o < KC * 8; o += 512) {
                const int kc = o >> 3, hh = (o >> 2) & 1, pp = o & 3;
                const bool second = kc >= a.KC1;
                const int c = 16 * (second ? kc - a.KC1 : kc) + 8 * hh + 2 * pp, Cp = second ? a.C2 : a.C1;
                const float *sp = second ? a.xs2 : a.xs1, *hp = second ? a.xh2 : a.xh1;
                float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
                if (c < Cp) { v.x = sp[c]; v.z = hp[c]; }
                if (c + 1 < Cp) { v.y = sp[c + 1]; v.w = hp[c + 1]; }
                xco[o] = v;
            }
        }
        if constexpr (POOL) {
            // (the stream IS the cloud: every column group of cloud `stream` passes through this workgroup)
            for (int o = threadIdx.x; o < a.tps * 32 * a.M; o += 512) { bins[o] = BFP_INIT_KEY; shadow[o] = -1000.f; }
            for (int o = threadIdx.x; o < a.tps * 32; o += 512) v0s[o] = 0u;
            const int32_t *idr = a.ids + (size_t)stream * L;
            for (int o = threadIdx.x; o < L; o += 512) {
                const int v = idr[o];
                idb[o] = (unsigned)v < (unsigned)a.M ? (unsigned char)v : (unsigned char)255;      // (255 >= M: nobody's column; M <= 255 here)
            }
        }
    }
    __syncthreads();

    const int npass = a.tps / MT;
    // this wave's column groups: wv, wv + stride, ... -- round-robin over the whole batch, or (POOL) over the workgroup's own cloud
    const int wv = POOL ? stream * a.gpc + wave : stream * 8 + wave, stride = POOL ? 8 : a.nstream * 8;
    const int gend = POOL ? (stream + 1) * a.gpc : a.ngroups;
    const int ngw = wv < gend ? (gend - wv + stride - 1) / stride : 0;
    const int wv0 = POOL ? stream * a.gpc : stream * 8;                               // ... and those of the workgroup's wave 0 (the most)
    const int ngw_wg = wv0 < gend ? (gend - wv0 + stride - 1) / stride : 0;

    // ---- load side: a cursor (group, chunk) that runs 4 chunks ahead of the multiplications
    int lg = 0, lkc = 0, lrep = 0;                  // index into this wave's groups / chunk / pass counter (X is re-read per pass)
    i32x4_t_ r1, r2;
    unsigned voa = 0;
    auto set_load_group = [&](int gi) {
        const int g = wv + (gi < ngw ? gi : ngw - 1) * stride;       // (past the end: the last group again -- harmless re-reads)
        const int b = g / a.gpc;
        const int ca = (g - b * a.gpc) * 64 + 2 * j;
        const int cca = ca < L ? ca : (g - b * a.gpc) * 64;
        r1 = bfr_rsrc(a.x1 + (size_t)b * a.C1 * L, (unsigned)a.C1 * rowB);
        r2 = bfr_rsrc(a.x2 ? a.x2 + (size_t)b * a.C2 * L : a.x1, (unsigned)(a.x2 ? a.C2 : 0) * rowB);
        voa = (unsigned)(8 * h * L + cca) * 2u;
    };
    auto issue = [&](unsigned (&x)[8]) {             // the cursor's chunk -> x, cursor + 1
        const bool second = lkc >= a.KC1;
        const i32x4_t_ rs = second ? r2 : r1;
        const unsigned row0 = (unsigned)(16 * (second ? lkc - a.KC1 : lkc)) * rowB;
#pragma unroll
        for (int t = 0; t < 8; ++t)
            asm volatile("buffer_load_dword %0, %1, %2, %3 offen" : "=v"(x[t]) : "v"(voa), "s"(rs), "s"(row0 + (unsigned)t * rowB) : "memory");
        if (++lkc == KC) {
            lkc = 0;
            if (++lrep == npass) { lrep = 0; set_load_group(++lg); }
        }
    };

    if (ngw > 0) {
        unsigned X[4][8];
        set_load_group(0);
        issue(X[0]); issue(X[1]); issue(X[2]); issue(X[3]);
        // 16 MT stores that land nowhere (offset outside the buffer): the first unit then starts behind the same queue as every other
        // one -- four chunks of loads, then an epilogue's stores -- and ONE wait count serves the first four chunks of every unit.  (Two
        // counts selected by a branch made hipcc hand the wait statements COPIES of the ring registers, taken before the wait.)
        if constexpr (!POOL) {                                 // (POOL: no stores anywhere in the loop -- the queue holds loads only)
            const i32x4_t_ r0 = bfr_rsrc(a.y, 4);              // (asm: as builtins hipcc folds the identical stores into one)
            const unsigned oob = 0x7FFFFF00u, zero = 0u;
#pragma unroll
            for (int t = 0; t < 16 * MT; ++t)
                asm volatile("buffer_store_dword %0, %1, %2, 0 offen" :: "v"(zero), "v"(oob), "s"(r0) : "memory");
        }

        for (int gi = 0; gi < ngw_wg; ++gi) {
            // Rows that do not start on a cache line (15000 columns: 30000 bytes): the 128-byte windows of neighbouring groups
            // (= neighbouring waves) share lines, and waves that drift microseconds apart each fetch the shared line from memory
            // (PMC: 1.8 x the bytes of the aligned case, profiles/r04u_bf16r_alignment.md).  Barriers keep the eight waves within
            // four chunks of each other: 0.54 -> 0.40 ms on 320 -> 384 (rows of 15040 columns, no barrier: 0.35).
            if (a.sync) __builtin_amdgcn_s_barrier();
            if (gi >= ngw) {
                if (a.sync == 2) for (int t = 0; t < npass * (KC / 4); ++t) __builtin_amdgcn_s_barrier();
                continue;
            }
            const int g = wv + gi * stride;
            const int b = g / a.gpc;
            const int ca = (g - b * a.gpc) * 64 + 2 * j;
            const bool pva = ca < L;
            // (POOL) the nodes of this lane's two columns (byte table in LDS; a column nobody owns -- padding, id outside [0, M) -- reads bin 0
            // and never updates), 0xFFFF - column
            unsigned ida_s = 0u, idb_s = 0u, npa = 0u;
            bool oka = false, okb = false;
            if constexpr (POOL) {
                if (pva) {
                    const unsigned idpair = *reinterpret_cast<const unsigned short *>(idb + ca);
                    oka = (idpair & 0xFFu) < (unsigned)a.M;
                    okb = (idpair >> 8) < (unsigned)a.M;
                    ida_s = oka ? (idpair & 0xFFu) : 0u;
                    idb_s = okb ? (idpair >> 8) : 0u;
                }
                npa = 0xFFFFu - (unsigned)ca;
            }
            const __amdgpu_buffer_rsrc_t ry = __builtin_amdgcn_make_buffer_rsrc(a.y + (size_t)b * a.Cout * L, 0, (int)((unsigned)a.Cout * rowB), 0x00020000);
            const unsigned voya = pva ? (unsigned)(4 * h * L + ca) * 2u : 0x7FFFFF00u;       // padded columns: the store falls outside the buffer

            for (int pass = 0; pass < npass; ++pass) {
                f32x16 acc[MT][2];
#pragma unroll
                for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                    for (int r = 0; r < 16; ++r) { acc[mt][0][r] = 0.f; acc[mt][1][r] = 0.f; }
                const uint4 *wpass = wl + (size_t)pass * MT * KC * 64 + lane;

#define BFR_BODY(q, kc, FIRST4)                                                                         \
                {                                                                                       \
                    unsigned ba[4], bb[4];                                                              \
                    bfr_wait_perm<((FIRST4 && !POOL) ? (24 + 16 * MT > 63 ? 63 : 24 + 16 * MT) : 24)>(X[q], ba, bb); \
                    issue(X[q]);                                                                        \
                    if constexpr (XAFF) {                                                               \
                        const float4 *cp_ = xco + ((kc) * 2 + h) * 4;                                   \
                        const float4 co_[4] = {cp_[0], cp_[1], cp_[2], cp_[3]};                         \
                        const unsigned fl_ = ((((kc) >= a.KC1) ? (a.xrelu & 2) : (a.xrelu & 1)) != 0) ? 0u : 0x80008000u; \
                        bfr_xaff(ba, co_, fl_);                                                         \
                        bfr_xaff(bb, co_, fl_);                                                         \
                    }                                                                                   \
                    const bf16x8 Ba = __builtin_bit_cast(bf16x8, make_uint4(ba[0], ba[1], ba[2], ba[3])); \
                    const bf16x8 Bb = __builtin_bit_cast(bf16x8, make_uint4(bb[0], bb[1], bb[2], bb[3])); \
                    _Pragma("unroll") for (int mt = 0; mt < MT; ++mt) {                                 \
                        const bf16x8 A = __builtin_bit_cast(bf16x8, wpass[(size_t)(mt * KC + (kc)) * 64]); \
                        acc[mt][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A, Ba, acc[mt][0], 0, 0, 0); \
                        acc[mt][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A, Bb, acc[mt][1], 0, 0, 0); \
                    }                                                                                   \
                }
                if (a.sync == 2) __builtin_amdgcn_s_barrier();
                BFR_BODY(0, 0, true) BFR_BODY(1, 1, true) BFR_BODY(2, 2, true) BFR_BODY(3, 3, true)
                for (int kc = 4; kc < KC; kc += 4) {
                    if (a.sync == 2) __builtin_amdgcn_s_barrier();
                    BFR_BODY(0, kc, false) BFR_BODY(1, kc + 1, false) BFR_BODY(2, kc + 2, false) BFR_BODY(3, kc + 3, false)
                }
#undef BFR_BODY

                // epilogue: affine + ReLU, two points per dword; statistics of the STORED values (what the normalise pass and the
                // backward read) into this wave's own LDS rows
#pragma unroll
                for (int mt = 0; mt < MT; ++mt) {
                    const int tl = pass * MT + mt;                           // tile inside the slab
                    const unsigned so_tile = (unsigned)((ct_begin + tl) * 32) * rowB;
                    const float2 *af = aff + tl * 32 + 4 * h;
                    float mine = 0.f;
                    if constexpr (POOL) {
#ifdef SONET_VARIANTS
                        if (a.abl == 1) continue;
#endif
                        // The two stored values of this lane per register (columns ca, ca + 1 of row tl * 32 + orow + 4 h) against their nodes'
                        // bins.  All 32 bin reads of the tile FIRST (one wait for the lot: a read - wait - compare - branch chain per value
                        // cost an LDS round trip 96 times a pass), then the compares; only a record breaker issues an LDS atomic.
                        const float *sh0 = shadow + (size_t)(tl * 32 + 4 * h) * a.M;
                        float cura[16], curb[16];
#pragma unroll
                        for (int r = 0; r < 16; ++r) {
                            const int orow = (r & 3) + 8 * (r >> 2);
                            cura[r] = sh0[orow * a.M + (int)ida_s];
                            curb[r] = sh0[orow * a.M + (int)idb_s];
                        }
#pragma unroll
                        for (int r = 0; r < 16; ++r) {
                            const int orow = (r & 3) + 8 * (r >> 2);
                            const float2 ss = af[orow];
                            const float va = __fmaf_rn(acc[mt][0][r], ss.x, ss.y), vb = __fmaf_rn(acc[mt][1][r], ss.x, ss.y);
                            const unsigned pk = cvt_pk_bf16(va, vb);
                            const float sa = __uint_as_float(pk << 16), sb = __uint_as_float(pk & 0xFFFF0000u);     // the STORED values
                            // (>=: an equal value at a smaller column still wins; a NaN fails the compare, as it never wins)
                            bool ca_ = oka && sa >= cura[r], cb_ = okb && sb >= curb[r];
#ifdef SONET_VARIANTS
                            if (a.abl == 2) { ca_ = sa == 12345.f; cb_ = sb == 12345.f; }
#endif
                            if (ca_ || cb_) {
                                const unsigned o2 = bfp_ord2(pk);
                                const int row = tl * 32 + orow + 4 * h;
                                if (ca_) { atomicMax(bins + (size_t)row * a.M + ida_s, (o2 << 16) | npa); shadow[(size_t)row * a.M + ida_s] = sa; }
                                if (cb_) { atomicMax(bins + (size_t)row * a.M + idb_s, (o2 & 0xFFFF0000u) | (npa - 1u)); shadow[(size_t)row * a.M + idb_s] = sb; }
                            }
                            if (ca == 0) v0s[tl * 32 + orow + 4 * h] = pk << 16;
                        }
                        continue;
                    }
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int orow = (r & 3) + 8 * (r >> 2);
                        const float2 ss = af[orow];
                        float va = __fmaf_rn(acc[mt][0][r], ss.x, ss.y), vb = __fmaf_rn(acc[mt][1][r], ss.x, ss.y);
                        if (a.relu & 1) { va = (va < 0.f) ? 0.f : va; vb = (vb < 0.f) ? 0.f : vb; }     // NaN propagates
                        const unsigned pk = cvt_pk_bf16(va, vb);
                        __builtin_amdgcn_raw_buffer_store_b32((int)pk, ry, voya, so_tile + (unsigned)orow * rowB, 0);
                        if constexpr (STATS) {
                            const float ra = pva ? __uint_as_float(pk << 16) : 0.f, rb = pva ? __uint_as_float(pk & 0xFFFF0000u) : 0.f;
                            const float s1 = row32_sum(ra + rb), s2 = row32_sum(__fmaf_rn(ra, ra, rb * rb));
                            // every lane of the half holds both sums: lane j keeps the one of register r = j & 15 (sum for j < 16, sum of
                            // squares for j >= 16) -- ONE LDS update per tile instead of a read-add-write chain per row
                            mine = ((j & 15) == r) ? (j < 16 ? s1 : s2) : mine;
                        }
                    }
                    if constexpr (STATS) {
                        const int rr = j & 15;
                        float *d = reinterpret_cast<float *>(stl + (size_t)wave * a.tps * 32 + tl * 32 + (rr & 3) + 8 * (rr >> 2) + 4 * h) + (j >> 4);
                        *d += mine;
                    }
                }
            }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // the look-ahead requests of the tail: nothing may land after the wave has left
    } else if (a.sync) {
        const int nb = a.sync == 2 ? 1 + npass * (KC / 4) : 1;
        for (int gi = 0; gi < ngw_wg * nb; ++gi) __builtin_amdgcn_s_barrier();
    }
    if constexpr (POOL) {
        // every column of the cloud has been through: the bins ARE the result (the epilogue of index_max_kernel, index_max.hip)
        __syncthreads();
        const int b = stream;
        for (int i = threadIdx.x; i < a.tps * 32 * a.M; i += 512) {
            const int rl = i / a.M, m = i - rl * a.M;
            const unsigned key = bins[i];
            const int pos = (int)(0xFFFFu - (key & 0xFFFFu));
            const size_t o = ((size_t)b * a.Cout + (size_t)ct_begin * 32 + rl) * a.M + m;
            a.out_idx[o] = pos;
            const unsigned okey = key >> 16;
            const bool won = key != BFP_INIT_KEY && (a.row_max == nullptr || a.row_max[(size_t)b * a.M + m] != 0);
            float v = __uint_as_float(((okey & 0x8000u) ? (okey ^ 0x8000u) : (~okey & 0xFFFFu)) << 16);
            if (!won) v = __uint_as_float(v0s[rl]);             // position 0: the value the storing launch would have written there
            a.out_val[o] = v;
        }
    }
    if constexpr (STATS) {
        __syncthreads();
        for (int t = threadIdx.x; t < a.tps * 32; t += 512) {
            double s1 = 0.0, s2 = 0.0;
            for (int w = 0; w < 8; ++w) { const float2 v = stl[(size_t)w * a.tps * 32 + t]; s1 += (double)v.x; s2 += (double)v.y; }
            double *dst = a.stats_partial + ((size_t)stream * a.Cout + (size_t)ct_begin * 32 + t) * 2;
            dst[0] = s1;
            dst[1] = s2;
        }
    }
}

#ifdef SONET_VARIANTS
// ---- the big-L variant: the wave's X tile lives in REGISTERS, W goes by in groups ---------------------------------------
// In bf16 the 320 -> 384 layer carries 175 flop per HBM byte: whatever re-reads X loses.  Here a wave loads its 64 columns x
// all Cin channels ONCE (KC x 8 dwords per lane: 160 registers at Cin = 320) and walks the output channels in G groups of
// MT tiles; the group's weights (KC x MT KiB) sit in one of two LDS buffers shared by the 4 waves, the next group's being
// staged (buffer_load -> registers -> ds_write, one slice per wave and K chunk, two chunks of latency) while this one is
// multiplied.  One barrier per group instead of one per K stage, no X traffic beyond the first touch, and the registers of X
// are refilled with the NEXT tile's columns chunk by chunk during the last group (each right after its last use), so the
// persistent workgroup never waits for a cold load.  G == 1 (all of W in one buffer): W is staged once, no barrier at all.
template <int KC, int MT>
__global__ __launch_bounds__(BF_THREADS, 1) void pointmlp_bf16_xreg_kernel(
    const uint16_t *__restrict__ x1, int C1, const uint16_t *__restrict__ x2, int C2, const uint4 *__restrict__ Wp,
    const float *__restrict__ scale, const float *__restrict__ shift, int relu, uint16_t *__restrict__ y,
    int Cout, int L, int gpc, long long ngroups, int G /*groups of MT cout tiles*/, int KC1 /*chunks fed by x1*/)
{
    constexpr int NSL = KC * MT;                              // slices per W group
    __shared__ uint4 wbuf[2][NSL][64];
    __shared__ float2 aff[1024];

    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int j = lane & 31, h = lane >> 5;
    const unsigned rowB = (unsigned)L * 2u;
    const unsigned vow = (unsigned)lane * 16u;
    const __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<uint4 *>(Wp), 0, (int)((unsigned)G * MT * KC * 1024u), 0x00020000);

    for (int o = threadIdx.x; o < Cout; o += BF_THREADS) aff[o] = make_float2(scale[o], shift[o]);
    // group 0 of W, cooperatively (every tile starts with it: restaged only when G > 1)
    for (int sl = wave; sl < NSL; sl += BF_WAVES) {
        const int kc = sl / MT, mt = sl - kc * MT;
        wbuf[0][sl][lane] = __builtin_bit_cast(uint4, __builtin_amdgcn_raw_buffer_load_b128(rw, vow, (unsigned)(mt * KC + kc) * 1024u, 0));
    }
    __syncthreads();

    const long long ntile = (ngroups + BF_WAVES - 1) / BF_WAVES;
    unsigned X[KC][8];
    long long nbuf = 0;                                         // running group counter: buffer = nbuf & 1

    // column bookkeeping of tile t for this wave
    struct Tile { __amdgpu_buffer_rsrc_t r1, r2, ry; unsigned vo, voy; bool pv; };
    auto tile_of = [&](long long t) {
        long long q = t * BF_WAVES + wave;
        const bool valid = q < ngroups;
        q = valid ? q : 0;
        const long long b = q / gpc;
        const int l0 = (int)(q - b * gpc) * 64;
        const int ca = l0 + 2 * j;
        const int cca = ca < L ? ca : l0;
        Tile T;
        T.r1 = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint16_t *>(x1 + b * (long long)C1 * L), 0, (int)((unsigned)C1 * rowB), 0x00020000);
        T.r2 = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint16_t *>(x2 ? x2 + b * (long long)C2 * L : x1), 0, (int)((unsigned)(x2 ? C2 : 0) * rowB), 0x00020000);
        T.ry = __builtin_amdgcn_make_buffer_rsrc(y + b * (long long)Cout * L, 0, (int)((unsigned)Cout * rowB), 0x00020000);
        T.vo = (unsigned)(8 * h * L + cca) * 2u;
        T.voy = (unsigned)(4 * h * L + cca) * 2u;
        T.pv = valid && ca < L;
        return T;
    };
    auto load_chunk = [&](unsigned (&dst)[8], const Tile &T, int kc) {
        const bool second = kc >= KC1;
        const unsigned row0 = (unsigned)(16 * (second ? kc - KC1 : kc)) * rowB;
#pragma unroll
        for (int t = 0; t < 8; ++t)
            dst[t] = (unsigned)(second ? __builtin_amdgcn_raw_buffer_load_b32(T.r2, T.vo, row0 + (unsigned)t * rowB, 0)
                                       : __builtin_amdgcn_raw_buffer_load_b32(T.r1, T.vo, row0 + (unsigned)t * rowB, 0));
    };

    Tile cur = tile_of(blockIdx.x);
    if ((long long)blockIdx.x < ntile) {
#pragma unroll
        for (int kc = 0; kc < KC; ++kc) load_chunk(X[kc], cur, kc);
    }
    // One group of MT cout tiles over the whole K range.  STAGE: the next group's W is staged meanwhile (every wave moves one
    // slice per K chunk during the first NSL / 4 chunks: buffer_load now, ds_write two chunks later).  PREF: each chunk's X
    // registers are refilled with the next tile's columns right after their last use.  Both are compile-time, so the K loop
    // is straight-line code: the A fragments of chunk kc + 1 are read from LDS before the MFMAs of chunk kc are issued and
    // every wait is a counted one.
    constexpr int NSTG = NSL / BF_WAVES;                        // staging chunks (NSL % 4 == 0 whenever G > 1 is possible)
    auto group_pass = [&](auto stage_c, auto pref_c, int g, int gn, int buf, long long tnext) {
        constexpr bool STAGE = decltype(stage_c)::value, PREF = decltype(pref_c)::value;
        Tile nxt;
        if constexpr (PREF) nxt = tile_of(tnext);
        f32x16 acc[MT][2];
        i32x4_t st[3];
        uint4 Af[2][MT];
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) Af[0][mt] = wbuf[buf][mt][lane];
#pragma unroll
        for (int kc = 0; kc < KC; ++kc) {
            __builtin_amdgcn_sched_barrier(0);                  // chunk boundary: nothing moves across (the compiler otherwise sinks
                                                                // every LDS read next to its MFMA and waits for it there)
            if constexpr (STAGE) {
                if (kc >= 2 && kc - 2 < NSTG) wbuf[buf ^ 1][(kc - 2) * BF_WAVES + wave][lane] = __builtin_bit_cast(uint4, st[(kc - 2) % 3]);
                if (kc < NSTG) {
                    const int sl = kc * BF_WAVES + wave;        // slice of the staged group: K chunk sl / MT, tile sl % MT
                    const int skc = sl / MT, smt = sl - skc * MT;
                    st[kc % 3] = __builtin_amdgcn_raw_buffer_load_b128(rw, vow, (unsigned)((gn * MT + smt) * KC + skc) * 1024u, 0);
                }
            }
            if (kc + 1 < KC) {
#pragma unroll
                for (int mt = 0; mt < MT; ++mt) Af[(kc + 1) & 1][mt] = wbuf[buf][(kc + 1) * MT + mt][lane];
            }
            __builtin_amdgcn_sched_barrier(0);                  // the next chunk's A fragments are requested BEFORE this chunk's MFMAs
            unsigned ba[4], bb[4];
#pragma unroll
            for (int p = 0; p < 4; ++p) {
                ba[p] = __builtin_amdgcn_perm(X[kc][2 * p + 1], X[kc][2 * p], 0x05040100u);
                bb[p] = __builtin_amdgcn_perm(X[kc][2 * p + 1], X[kc][2 * p], 0x07060302u);
            }
            if constexpr (PREF) load_chunk(X[kc], nxt, kc);   // this chunk's registers are free now: next tile's columns
            const bf16x8 Ba = __builtin_bit_cast(bf16x8, make_uint4(ba[0], ba[1], ba[2], ba[3]));
            const bf16x8 Bb = __builtin_bit_cast(bf16x8, make_uint4(bb[0], bb[1], bb[2], bb[3]));
            const f32x16 zero = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) {
                const bf16x8 A = __builtin_bit_cast(bf16x8, Af[kc & 1][mt]);
                acc[mt][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A, Ba, kc == 0 ? zero : acc[mt][0], 0, 0, 0);
                acc[mt][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A, Bb, kc == 0 ? zero : acc[mt][1], 0, 0, 0);
            }
        }
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (STAGE) {
#pragma unroll
            for (int kc = (KC >= 2 ? KC - 2 : 0); kc < KC; ++kc)
                if (kc < NSTG) wbuf[buf ^ 1][kc * BF_WAVES + wave][lane] = __builtin_bit_cast(uint4, st[kc % 3]);
        }
        if (cur.pv) {
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) {
                __builtin_amdgcn_sched_barrier(0);              // (keeps the coefficient reads of all tiles from being hoisted at once)
                const int ct = g * MT + mt;
                const unsigned so_tile = (unsigned)(ct * 32) * rowB;
                const float2 *ap = aff + ct * 32 + 4 * h;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int orow = (r & 3) + 8 * (r >> 2);
                    const float2 ss = ap[orow];
                    float va = __fmaf_rn(acc[mt][0][r], ss.x, ss.y), vb = __fmaf_rn(acc[mt][1][r], ss.x, ss.y);
                    if (relu) { va = (va < 0.f) ? 0.f : va; vb = (vb < 0.f) ? 0.f : vb; }
                    __builtin_amdgcn_raw_buffer_store_b32((int)cvt_pk_bf16(va, vb), cur.ry, cur.voy, so_tile + (unsigned)orow * rowB, 0);
                }
            }
        }
    };
    for (long long tile = blockIdx.x; tile < ntile; tile += gridDim.x) {
        const bool has_next = tile + gridDim.x < ntile;
        const long long nxt = tile + gridDim.x;
        for (int g = 0; g < G; ++g) {
            const int buf = (int)(nbuf & 1);
            const bool last = g == G - 1;
            const int gn = last ? 0 : g + 1;                    // the group staged meanwhile (the next tile starts with group 0 again)
            const bool stage = G > 1 && (!last || has_next);
            const bool pref = last && has_next;
            if (stage) { if (pref) group_pass(T1{}, T1{}, g, gn, buf, nxt); else group_pass(T1{}, T0{}, g, gn, buf, nxt); }
            else       { if (pref) group_pass(T0{}, T1{}, g, gn, buf, nxt); else group_pass(T0{}, T0{}, g, gn, buf, nxt); }
            if (G > 1) {
                __syncthreads();                                // the staged group is complete, this buffer is free
                nbuf += stage ? 1 : 0;
            }
        }
        if (has_next) cur = tile_of(nxt);
    }
}
#endif  // SONET_VARIANTS

}  // namespace

extern "C" size_t sonet_pointmlp_bf16_pack_size(int Cin, int Cout)
{
    if (Cin <= 0 || Cout <= 0) return 0;
    return (size_t)sonet::ceil_div(Cout, 32) * sonet::ceil_div(Cin, 16) * 64 * 16;     // bytes
}

static int bf16_pack_impl(const char *what, const float *W, void *Wp, int Cin, int Cout, int rows, long long rs, long long cs, sonet_stream_t stream)
{
    SONET_REQUIRE(W && Wp, "%s: NULL pointer", what);
    SONET_REQUIRE(Cin > 0 && Cout > 0 && rows > 0 && rows <= Cout, "%s: bad size Cin=%d Cout=%d rows=%d", what, Cin, Cout, rows);
    const int KC = sonet::ceil_div(Cin, 16);
    const long long total = (long long)sonet::ceil_div(Cout, 32) * KC * 64;
    hipLaunchKernelGGL(bf16_pack_kernel, dim3((unsigned)sonet::ceil_div64(total, 256)), dim3(256), 0, sonet::as_stream(stream),
                       W, reinterpret_cast<uint4 *>(Wp), Cin, rows, KC, total, rs, cs);
    return sonet::launched(what);
}

extern "C" int sonet_pointmlp_bf16_pack(const float *W, void *Wp, int Cin, int Cout, sonet_stream_t stream)
{
    return bf16_pack_impl("sonet_pointmlp_bf16_pack", W, Wp, Cin, Cout, Cout, Cin, 1, stream);
}

/* The bf16 pack of a matrix given by element strides (see sonet_pointmlp_x3_pack_strided): element (o, c) = W[o * row_stride + c *
 * col_stride] for o < rows, zeros for rows <= o < Cout. */
extern "C" int sonet_pointmlp_bf16_pack_strided(const float *W, long long row_stride, long long col_stride, void *Wp, int Cin, int Cout, int rows,
                                                sonet_stream_t stream)
{
    return bf16_pack_impl("sonet_pointmlp_bf16_pack_strided", W, Wp, Cin, Cout, rows, row_stride, col_stride, stream);
}

static int bf16_run_impl(const char *what, const uint16_t *x1, int C1, const uint16_t *x2, int C2, const void *Wp,
                         const float *scale, const float *shift, int relu, uint16_t *y,
                         int B, int Cout, int L, sonet_stream_t stream, const int32_t *gidx = nullptr, int L1 = 0,
                         double *stats_ws = nullptr, float *mean = nullptr, float *var = nullptr, const BfrXaff *xaff = nullptr,
                         const uint16_t *yadd = nullptr, const BfBnb *bnb = nullptr)
{
    if (!gidx) L1 = L;
    SONET_REQUIRE(L1 > 0, "%s: non-positive size", what);
    SONET_REQUIRE(x1 && Wp && scale && shift && y, "%s: NULL pointer", what);
    SONET_REQUIRE(B > 0 && C1 > 0 && C2 >= 0 && Cout > 0 && L > 0, "%s: non-positive size", what);
    SONET_REQUIRE((C2 == 0) == (x2 == nullptr), "%s: x2 and C2 disagree", what);
    SONET_REQUIRE(C2 == 0 || C1 % 16 == 0, "%s: with a second input C1=%d must be a multiple of 16", what, C1);
    if (Cout % 32 != 0) return sonet::fail(SONET_ERR_UNSUPPORTED, "%s: Cout=%d must be a multiple of 32", what, Cout);
    const int Cin = C1 + C2;
    const int CT = Cout / 32, KC = sonet::ceil_div(Cin, 16);
    const int gpc = sonet::ceil_div(L, 64);
    const long long ngroups = (long long)B * gpc;
    if ((double)C1 * L1 * 2.0 >= 2.0e9 || (double)(C1 > C2 ? C1 : C2) * L * 2.0 >= 2.0e9 || (double)Cout * L * 2.0 >= 4.0e9 || (double)CT * KC * 1024.0 >= 4.0e9)
        return sonet::fail(SONET_ERR_UNSUPPORTED, "%s: a per-cloud panel is too large", what);
    const long long nwg_x = sonet::ceil_div64(ngroups, (long long)BF_WAVES);
    if (nwg_x > 0x7FFFFFFFll) return sonet::fail(SONET_ERR_UNSUPPORTED, "%s: too many points", what);
    int MT = 1, S = 2;
    // measured on the first PointNet's shapes (profiles/r02b_bench_bf16_kernel_streaming.log): 4 tiles x 2 chunks per stage wins
    // everywhere it divides (two workgroups per CU); 12 tiles -- X read once -- leaves one starved wave per SIMD (3x slower)
    if (CT % 4 == 0) MT = 4;
    else if (CT % 6 == 0) MT = 6;
    else if (CT % 2 == 0) MT = 2;
    // node-level launches (a few thousand columns): with 4 tiles per group 515 -> 768 at 64 x 64 columns is 16 x 6 = 96 workgroups of
    // three groups each on 256 CUs; two tiles per group doubles the workgroups that can run side by side
    if (MT == 4 && nwg_x * (CT / 4) < 256 && CT % 2 == 0) MT = 2;
    if (bnb && MT != 4 && MT != 2 && CT % 2 == 0) MT = 2;     // (the BatchNorm-backward load is instantiated for 4 and 2 tiles)
    if (const char *e = sonet::knob("SONET_BF16_MT")) {            // tuning knob (bench experiments only)
        const int want = atoi(e);
        if ((want == 12 || want == 6 || want == 4 || want == 2 || want == 1) && CT % want == 0) MT = want;
    }
#ifdef SONET_VARIANTS
    if (const char *e = sonet::knob("SONET_BF16_ABL")) relu = (relu & 1) | (atoi(e) << 1);
#endif
    if (const char *e = sonet::knob("SONET_BF16_S")) {
        const int want = atoi(e);
        if (want == 1 || want == 2) S = want;
    }
    if (KC == 1) S = 1;
    // output-channel slabs: small launches spread the channels over workgroups too -- the smallest divisor d of the CT / MT groups
    // that gives >= 512 workgroups (and <= 32 tiles per slab), else the largest (round 2a: powers of two only, which left 768
    // channels = 6 groups at d = 2)
    int ysplit = CT / MT;
    for (int d = CT / MT; d >= 1; --d)
        if ((CT / MT) % d == 0 && nwg_x * d >= 512 && CT / d <= 32) ysplit = d;
    if (const char *e = sonet::knob("SONET_BF16_YSPLIT")) {
        const int want = atoi(e);
        if (want >= 1 && (CT / MT) % want == 0 && CT / want <= 32) ysplit = want;
    }
    const int ct_per_y = CT / ysplit;
    if (ct_per_y > 32) return sonet::fail(SONET_ERR_UNSUPPORTED, "%s: Cout=%d too large", what, Cout);
    const bool paired = (L % 2 == 0) && (L1 % 2 == 0) && gidx == nullptr &&
                        ((reinterpret_cast<uintptr_t>(x1) | reinterpret_cast<uintptr_t>(x2) | reinterpret_cast<uintptr_t>(y)) & 3) == 0;
    hipStream_t st = sonet::as_stream(stream);
    const uint4 *wp = reinterpret_cast<const uint4 *>(Wp);
#ifdef SONET_VARIANTS   // (the X-in-registers kernel: bit-identical, measured slower; variants build only)
    // big launches with dword-aligned rows: X tile in registers, W by groups (above).  (KC, MT) pairs that are instantiated:
    // K chunks 1 / 4 / 8 / 16 / 20 / 24 with the tile count that keeps two W buffers inside the LDS
    {
        int xmt = 0;
        switch (KC) { case 1: xmt = 2; break; case 4: case 8: case 16: xmt = 4; break; case 20: xmt = 3; break; case 24: xmt = 2; break; default: break; }
        // Opt-in (SONET_BF16_XREG=1; 2 = whenever the shape allows, for the tests): bit-identical to the streaming kernel but
        // measured SLOWER on every first-PointNet shape (0.54 vs 0.46 ms on 320 -> 384 at B = 64): one wave per SIMD, and the
        // 160 prefetch loads of the next tile pile up behind the 6-bit vmcnt during the last group.  Kept as the record of
        // the experiment and as the skeleton of the fused bf16 kernel (pointresnet_bf16.hip), where X never comes from HBM.
        const char *e = sonet::knob("SONET_BF16_XREG");
        const bool want = e && atoi(e) >= 1, force = e && atoi(e) == 2;
        if (want && !stats_ws && paired && xmt > 0 && CT % xmt == 0 && Cout <= 1024 && (nwg_x >= 512 || force) && (sonet::knob("SONET_BF16_MT") == nullptr || force)) {
            int dev = 0, cus = 256;
            if (hipGetDevice(&dev) == hipSuccess) {
                int v = 0;
                if (hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && v > 0) cus = v;
            }
            const int KC1x = C2 > 0 ? (C1 >> 4) : KC;
            const int Gx = CT / xmt;
            dim3 gridx((unsigned)(nwg_x < cus ? nwg_x : cus)), blockx(BF_THREADS);
#define BFX_LAUNCH(KK, MM) hipLaunchKernelGGL((pointmlp_bf16_xreg_kernel<KK, MM>), gridx, blockx, 0, st, x1, C1, x2, C2, wp, scale, shift, relu, y, \
                                              Cout, L, gpc, ngroups, Gx, KC1x)
            switch (KC) {
                case 1: BFX_LAUNCH(1, 2); break;
                case 4: BFX_LAUNCH(4, 4); break;
                case 8: BFX_LAUNCH(8, 4); break;
                case 16: BFX_LAUNCH(16, 4); break;
                case 20: BFX_LAUNCH(20, 3); break;
                default: BFX_LAUNCH(24, 2); break;
            }
#undef BFX_LAUNCH
            return sonet::launched(what);
        }
    }
#endif  // SONET_VARIANTS
    // big launches with dword-aligned rows: the streaming generation (W slab resident in LDS, persistent waves)
    {
        // (an accumulating store -- yadd -- runs on the staged kernel below: its epilogue loads would sit in the streaming kernel's hand-counted queue)
        bool want = paired && KC % 4 == 0 && CT % 2 == 0 && ngroups >= 8192 && ngroups < 0x7FFFFFFFll && yadd == nullptr && bnb == nullptr;
        if (const char *e = sonet::knob("SONET_BF16_STREAM")) want = want && atoi(e) != 0;
        int best_ns = 0, best_cost = 1 << 30;
        for (int ns = 1; want && ns <= CT / 2; ++ns) {
            if (CT % ns) continue;
            const int tps = CT / ns;
            if (tps % 2 || (size_t)tps * KC * 1024 + (size_t)tps * 256 + (stats_ws ? (size_t)tps * 2048 : 0) + (xaff ? (size_t)KC * 128 : 0) > 158 * 1024) continue;   // W slab + affine + statistics rows (+ the normalise-on-load table)
            const int cost = ns * (tps / (tps % 4 == 0 ? 4 : 2));            // times X goes through the vector memory path
            if (cost < best_cost) { best_cost = cost; best_ns = ns; }
        }
        int cus = 256;
        {
            int dev = 0, v = 0;
            if (hipGetDevice(&dev) == hipSuccess && hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && v >= 8) cus = v;
        }
        const int spx = best_ns > 0 ? (cus / 8) / best_ns : 0;            // column streams per XCD
        if (want && best_ns > 0 && spx > 0) {
            BfrArgs a;
            a.x1 = x1; a.x2 = x2; a.Wp = wp; a.scale = scale; a.shift = shift; a.y = y; a.stats_partial = stats_ws; a.abl = 0;
            a.C1 = C1; a.C2 = C2; a.Cout = Cout; a.L = L; a.gpc = gpc; a.relu = relu & 1; a.KC = KC; a.KC1 = C2 > 0 ? (C1 >> 4) : KC;
            a.tps = CT / best_ns; a.nslab = best_ns; a.nstream = 8 * spx; a.ngroups = (int)ngroups;
            a.sync = ((unsigned)L * 2u) % 128u != 0 ? 2 : 0;
            if (const char *e = sonet::knob("SONET_BF16_SYNC")) a.sync = atoi(e);
            size_t lds = (size_t)a.tps * KC * 1024 + (size_t)a.tps * 32 * 8 + (stats_ws ? (size_t)8 * a.tps * 32 * 8 : 0);
            a.xs1 = a.xh1 = a.xs2 = a.xh2 = nullptr; a.xrelu = 0; a.xco_off = 0;
            if (xaff) {
                a.xs1 = xaff->xs1; a.xh1 = xaff->xh1; a.xs2 = xaff->xs2; a.xh2 = xaff->xh2; a.xrelu = xaff->xrelu;
                a.xco_off = (unsigned)lds;
                lds += (size_t)KC * 128;
            }
            const dim3 gridr((unsigned)(8 * spx * best_ns)), blockr(512);
#define BFR_LAUNCH(MM, SS) do { static bool attr_set = false;                                                                         \
                if (!attr_set) { if (hipFuncSetAttribute(reinterpret_cast<const void *>(&pointmlp_bf16r_kernel<MM, SS>),             \
                                                         hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess)       \
                                     return sonet::fail(SONET_ERR_UNSUPPORTED, "%s: cannot reserve the LDS", what);                   \
                                 attr_set = true; }                                                                                   \
                hipLaunchKernelGGL((pointmlp_bf16r_kernel<MM, SS>), gridr, blockr, lds, st, a); } while (0)
#define BFR_LAUNCH_X(MM) do { static bool attr_set = false;                                                                             \
                if (!attr_set) { if (hipFuncSetAttribute(reinterpret_cast<const void *>(&pointmlp_bf16r_kernel<MM, true, false, true>), \
                                                         hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess)       \
                                     return sonet::fail(SONET_ERR_UNSUPPORTED, "%s: cannot reserve the LDS", what);                   \
                                 attr_set = true; }                                                                                   \
                hipLaunchKernelGGL((pointmlp_bf16r_kernel<MM, true, false, true>), gridr, blockr, lds, st, a); } while (0)
            if (xaff) {                                        // (normalise-on-load: the training forward, i.e. with statistics)
                if (!stats_ws) return sonet::fail(SONET_ERR_UNSUPPORTED, "%s: normalise-on-load comes with the statistics epilogue", what);
                if (a.tps % 4 == 0) BFR_LAUNCH_X(4); else BFR_LAUNCH_X(2);
            }
            else if (a.tps % 4 == 0) { if (stats_ws) BFR_LAUNCH(4, true); else BFR_LAUNCH(4, false); }
            else                     { if (stats_ws) BFR_LAUNCH(2, true); else BFR_LAUNCH(2, false); }
#undef BFR_LAUNCH_X
#undef BFR_LAUNCH
            if (stats_ws) sonet::launch_stats_finalize(stats_ws, a.nstream, Cout, 1.0 / ((double)B * L), mean, var, st);
            return sonet::launched(what);
        }
    }
    if (xaff) return sonet::fail(SONET_ERR_UNSUPPORTED, "%s: normalise-on-load exists on the streaming kernel only (Cin %% 64 == 0, even L, >= 8192 column groups)", what);
    if (yadd && (!paired || stats_ws)) return sonet::fail(SONET_ERR_UNSUPPORTED, "%s: the accumulating store needs even L, 4-byte aligned rows and no statistics", what);
    dim3 grid((unsigned)nwg_x, (unsigned)ysplit), block(BF_THREADS);
#define BF_ARGS grid, block, 0, st, x1, C1, x2, C2, wp, scale, shift, relu, y, Cout, L, gpc, ngroups, CT, KC, ct_per_y, gidx, L1, stats_ws, yadd
    if (bnb) {
        // BatchNorm / ReLU backward on the operand load (the compiler-scheduled kernel: a second operand stream beside gy)
        if (!paired || C2 != 0 || C1 % 16 != 0 || C1 > BNB_CMAX || stats_ws || S != 2 || (MT != 4 && MT != 2) ||
            ((reinterpret_cast<uintptr_t>(bnb->raw) | reinterpret_cast<uintptr_t>(bnb->g_raw_out)) & 3) != 0)
            return sonet::fail(SONET_ERR_UNSUPPORTED, "%s: needs even L, 4-byte aligned rows, C %% 16 == 0, 32 <= C <= %d, Cout %% 64 == 0", what, BNB_CMAX);
        if (MT == 4) hipLaunchKernelGGL((pointmlp_bf16_kernel<4, 2, true, 2, true>), BF_ARGS, *bnb);
        else         hipLaunchKernelGGL((pointmlp_bf16_kernel<2, 2, true, 2, true>), BF_ARGS, *bnb);
        return sonet::launched(what);
    }
#define BF_LAUNCH(MM) do { if (paired) { if (S == 2) hipLaunchKernelGGL((pointmlp_bf16_kernel<MM, 2, true>), BF_ARGS); \
                                         else        hipLaunchKernelGGL((pointmlp_bf16_kernel<MM, 1, true>), BF_ARGS); } \
                           else        { if (S == 2) hipLaunchKernelGGL((pointmlp_bf16_kernel<MM, 2, false>), BF_ARGS); \
                                         else        hipLaunchKernelGGL((pointmlp_bf16_kernel<MM, 1, false>), BF_ARGS); } } while (0)
    if (MT == 12) S = 1;                                      // (12 slices per chunk already: one chunk per stage)
#ifdef SONET_VARIANTS
    // X and W two stages ahead (NXB = 3): measured 1-5 % SLOWER than one stage ahead on every shape (profiles/r04s_bf16_layers_ablation.log
    // and docs/findings.md R4.8: the staged kernel is bound by the serialisation of a pass, not by its look-ahead) -- kept as the record
    // of the experiment, selectable in the variants build only
    if (const char *e = sonet::knob("SONET_BF16_NXB")) {
        if (atoi(e) == 3 && paired && S == 2 && (MT == 4 || MT == 2)) {
            if (MT == 4) hipLaunchKernelGGL((pointmlp_bf16_kernel<4, 2, true, 3>), BF_ARGS);
            else         hipLaunchKernelGGL((pointmlp_bf16_kernel<2, 2, true, 3>), BF_ARGS);
            if (stats_ws) sonet::launch_stats_finalize(stats_ws, (int)nwg_x, Cout, 1.0 / ((double)B * L), mean, var, st);
            return sonet::launched(what);
        }
    }
#endif
    switch (MT) {
        case 12: BF_LAUNCH(12); break;
        case 6: BF_LAUNCH(6); break;
        case 4: BF_LAUNCH(4); break;
        case 2: BF_LAUNCH(2); break;
        default: BF_LAUNCH(1);
    }
#undef BF_LAUNCH
#undef BF_ARGS
    if (stats_ws) sonet::launch_stats_finalize(stats_ws, (int)nwg_x, Cout, 1.0 / ((double)B * L), mean, var, st);
    return sonet::launched(what);
}

extern "C" size_t sonet_pointmlp_bf16_stats_ws_size(int B, int Cout, int L)
{
    if (B <= 0 || Cout <= 0 || L <= 0) return 0;
    return (size_t)sonet::ceil_div64((long long)B * sonet::ceil_div(L, 64), BF_WAVES) * Cout * 2 * sizeof(double);
}

/* sonet_pointmlp_bf16 that also returns mean / biased variance of the stored (bf16) output over (B, L): BatchNorm's batch statistics
 * (models/layers.py:60-70) from the kernel's epilogue.  stats_ws: sonet_pointmlp_bf16_stats_ws_size bytes. */
extern "C" int sonet_pointmlp_bf16_stats(const uint16_t *x1, int C1, const uint16_t *x2, int C2, const void *Wp,
                                         const float *scale, const float *shift, int relu, uint16_t *y,
                                         int B, int Cout, int L, void *stats_ws, float *mean, float *var, sonet_stream_t stream)
{
    SONET_REQUIRE(stats_ws && mean && var, "sonet_pointmlp_bf16_stats: NULL pointer");
    return bf16_run_impl("sonet_pointmlp_bf16_stats", x1, C1, x2, C2, Wp, scale, shift, relu, y, B, Cout, L, stream, nullptr, 0,
                         reinterpret_cast<double *>(stats_ws), mean, var);
}

/* sonet_pointmlp_bf16_stats with NORMALISE-ON-LOAD (bf16 training forward, hidden layers of the first PointNet): x1 / x2 hold the RAW outputs
 * of training-mode BatchNorm layers (models/layers.py:60-70, :282-296) whose normalise + ReLU pass was never run; the operand load computes
 * act(raw * xs[c] + xh[c]) in f32 and rounds to bf16 -- exactly what sonet_channel_affine_act_bf16 would have stored -- so the outputs equal
 * sonet_pointmlp_bf16_stats on the normalised tensors bit for bit.  xs1, xh1 [C1] (xs2, xh2 [C2] when C2 > 0); xrelu bit 0 / 1: ReLU on x1 / x2.
 * Streaming-kernel shapes only ((C1 + C2) % 64 == 0, even L, 4-byte aligned rows, >= 8192 column groups): SONET_ERR_UNSUPPORTED otherwise. */
extern "C" int sonet_pointmlp_bf16_stats_xaff(const uint16_t *x1, int C1, const uint16_t *x2, int C2, const void *Wp,
                                              const float *scale, const float *shift, int relu, uint16_t *y,
                                              int B, int Cout, int L, void *stats_ws, float *mean, float *var,
                                              const float *xs1, const float *xh1, const float *xs2, const float *xh2, int xrelu,
                                              sonet_stream_t stream)
{
    const char *what = "sonet_pointmlp_bf16_stats_xaff";
    SONET_REQUIRE(stats_ws && mean && var && xs1 && xh1 && (C2 == 0 || (xs2 && xh2)), "%s: NULL pointer", what);
    const BfrXaff xa = {xs1, xh1, xs2, xh2, xrelu};
    return bf16_run_impl(what, x1, C1, x2, C2, Wp, scale, shift, relu, y, B, Cout, L, stream, nullptr, 0,
                         reinterpret_cast<double *>(stats_ws), mean, var, &xa);
}

/* The layer and the per-node arg-max pool of its output in ONE launch; the output itself is never written (the last layer of the first
 * PointNet in training, when only the pooled map is consumed: models/layers.py:431 + models/networks.py:180-185).
 * out_idx [B][Cout][M] = what sonet_index_max_bf16 reports on the tensor sonet_pointmlp_bf16 would have written (first maximum above -1000 in
 * column order, else 0), out_val [B][Cout][M] f32 = that tensor's value at out_idx * row_max (sonet_index_max_gather_bf16): bit for bit.
 * A workgroup = (cloud, slab of output tiles) on the streaming kernel: W slab, the cloud's node ids (bytes) and the bins of its rows x M
 * nodes live in LDS, every column group of the cloud passes through it once.  ids [B][L] i32, row_max [B][M] i32 or NULL.
 * Needs: L even and < 65535 (a key holds the column in 16 bits), 4-byte aligned rows, (C1 + C2) % 64 == 0, Cout % 32 == 0 with a slab shape that fits the LDS, M <= 255. */
static int bf16_pool_impl(const char *what, const uint16_t *x1, int C1, const uint16_t *x2, int C2, const void *Wp,
                          const float *scale, const float *shift, int relu, const int32_t *ids, const int32_t *row_max,
                          int32_t *out_idx, float *out_val, int B, int Cout, int L, int M, sonet_stream_t stream, const BfrXaff *xaff)
{
    SONET_REQUIRE(x1 && Wp && scale && shift && ids && out_idx && out_val, "%s: NULL pointer", what);
    SONET_REQUIRE(B > 0 && C1 > 0 && C2 >= 0 && Cout > 0 && L > 0 && M > 0, "%s: non-positive size", what);
    SONET_REQUIRE((C2 == 0) == (x2 == nullptr), "%s: x2 and C2 disagree", what);
    SONET_REQUIRE(C2 == 0 || C1 % 16 == 0, "%s: with a second input C1=%d must be a multiple of 16", what, C1);
    const int Cin = C1 + C2;
    if (Cout % 32 != 0 || Cin % 64 != 0 || L % 2 != 0 || L > 65534 || M > 255 || B > 65535 ||
        ((reinterpret_cast<uintptr_t>(x1) | reinterpret_cast<uintptr_t>(x2)) & 3) != 0)
        return sonet::fail(SONET_ERR_UNSUPPORTED, "%s: Cout %% 32, Cin %% 64, even L, M <= 255, 4-byte aligned rows (Cout=%d Cin=%d L=%d M=%d)", what, Cout, Cin, L, M);
    const int CT = Cout / 32, KC = Cin / 16, gpc = sonet::ceil_div(L, 64);
    if ((double)C1 * L * 2.0 >= 2.0e9 || (double)C2 * L * 2.0 >= 2.0e9 || (long long)B * gpc >= 0x7FFFFFFFll)
        return sonet::fail(SONET_ERR_UNSUPPORTED, "%s: a per-cloud panel is too large", what);
    int cus = 256;
    {
        int dev = 0, v = 0;
        if (hipGetDevice(&dev) == hipSuccess && hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && v >= 8) cus = v;
    }
    // output slabs: one workgroup per (cloud, slab) and CU -- the slab count whose B x nslab workgroups need the fewest rounds of tile passes
    int best_ns = 0, best_mt = 0;
    long long best_cost = 0;
    size_t best_lds = 0;
    for (int ns = 1; ns <= CT; ++ns) {
        if (CT % ns) continue;
        const int tps = CT / ns;
        const int mt = tps % 4 == 0 ? 4 : tps % 3 == 0 ? 3 : tps % 2 == 0 ? 2 : 0;
        if (mt == 0) continue;
        const size_t lds = (size_t)tps * KC * 1024 + (size_t)tps * 32 * 8 + (size_t)tps * 32 * M * 8 + (size_t)tps * 32 * 4 + (size_t)((L + 15) & ~15)
                           + (xaff ? (size_t)KC * 128 : 0);
        if (lds > 158 * 1024) continue;
        const long long cost = sonet::ceil_div64((long long)B * ns, cus) * tps;
        if (best_ns == 0 || cost < best_cost) { best_ns = ns; best_mt = mt; best_cost = cost; best_lds = lds; }
    }
#ifdef SONET_VARIANTS
    if (const char *e = sonet::knob("SONET_BF16_POOL_NS")) {       // (tools/bench_pool_epilogue.py: a slab count by hand)
        const int ns = atoi(e);
        if (ns >= 1 && CT % ns == 0) {
            const int tps = CT / ns, mt = tps % 4 == 0 ? 4 : tps % 3 == 0 ? 3 : tps % 2 == 0 ? 2 : 0;
            const size_t lds = (size_t)tps * KC * 1024 + (size_t)tps * 32 * 8 + (size_t)tps * 32 * M * 8 + (size_t)tps * 32 * 4 + (size_t)((L + 15) & ~15)
                               + (xaff ? (size_t)KC * 128 : 0);
            if (mt && lds <= 158 * 1024) { best_ns = ns; best_mt = mt; best_lds = lds; }
        }
    }
#endif
    if (best_ns == 0) return sonet::fail(SONET_ERR_UNSUPPORTED, "%s: no slab shape fits the LDS (Cin=%d Cout=%d M=%d L=%d)", what, Cin, Cout, M, L);
    BfrArgs a;
    a.x1 = x1; a.x2 = x2; a.Wp = reinterpret_cast<const uint4 *>(Wp); a.scale = scale; a.shift = shift; a.y = nullptr; a.stats_partial = nullptr;
    a.C1 = C1; a.C2 = C2; a.Cout = Cout; a.L = L; a.gpc = gpc; a.relu = relu & 1; a.KC = KC; a.KC1 = C2 > 0 ? (C1 >> 4) : KC;
    a.tps = CT / best_ns; a.nslab = best_ns; a.nstream = (B + 7) / 8 * 8; a.ngroups = B * gpc;
    // (no workgroup barriers here: the eight waves of a workgroup walk adjacent column groups of ONE cloud and its four slab workgroups share
    //  an XCD's L2 -- the shared cache lines of unaligned rows come from L2 either way, and free-running waves let one wave's epilogue sit
    //  under another's MFMAs: 559 -> 478 us at 64 x 15000 columns, tools/bench_pool_epilogue.py)
    a.sync = 0;
    a.ids = ids; a.row_max = row_max; a.out_idx = out_idx; a.out_val = out_val; a.M = M; a.abl = 0;
    a.xs1 = a.xh1 = a.xs2 = a.xh2 = nullptr; a.xrelu = 0; a.xco_off = 0;
    if (xaff) {
        a.xs1 = xaff->xs1; a.xh1 = xaff->xh1; a.xs2 = xaff->xs2; a.xh2 = xaff->xh2; a.xrelu = xaff->xrelu;
        a.xco_off = (unsigned)(best_lds - (size_t)KC * 128);             // (the table is the last item of the slab's LDS budget)
    }
#ifdef SONET_VARIANTS
    if (const char *e = sonet::knob("SONET_BF16_POOL_ABL")) a.abl = atoi(e);
    if (const char *e = sonet::knob("SONET_BF16_SYNC")) a.sync = atoi(e);
#endif
    hipStream_t st = sonet::as_stream(stream);
    // (workgroup -> (slab, stream) as in the storing launch: the slabs of a cloud on one XCD; streams >= B leave at once)
    const dim3 gridr((unsigned)(a.nstream * best_ns)), blockr(512);
#define BFP_LAUNCH(MM) do { static bool attr_set = false;                                                                             \
        if (!attr_set) { if (hipFuncSetAttribute(reinterpret_cast<const void *>(&pointmlp_bf16r_kernel<MM, false, true>),            \
                                                 hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess)               \
                             return sonet::fail(SONET_ERR_UNSUPPORTED, "%s: cannot reserve the LDS", what);                           \
                         attr_set = true; }                                                                                           \
        hipLaunchKernelGGL((pointmlp_bf16r_kernel<MM, false, true>), gridr, blockr, best_lds, st, a); } while (0)
#define BFP_LAUNCH_X(MM) do { static bool attr_set = false;                                                                           \
        if (!attr_set) { if (hipFuncSetAttribute(reinterpret_cast<const void *>(&pointmlp_bf16r_kernel<MM, false, true, true>),      \
                                                 hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess)               \
                             return sonet::fail(SONET_ERR_UNSUPPORTED, "%s: cannot reserve the LDS", what);                           \
                         attr_set = true; }                                                                                           \
        hipLaunchKernelGGL((pointmlp_bf16r_kernel<MM, false, true, true>), gridr, blockr, best_lds, st, a); } while (0)
    if (xaff) { if (best_mt == 4) BFP_LAUNCH_X(4); else if (best_mt == 3) BFP_LAUNCH_X(3); else BFP_LAUNCH_X(2); }
    else if (best_mt == 4) BFP_LAUNCH(4); else if (best_mt == 3) BFP_LAUNCH(3); else BFP_LAUNCH(2);
#undef BFP_LAUNCH_X
#undef BFP_LAUNCH
    return sonet::launched(what);
}

extern "C" int sonet_pointmlp_bf16_pool(const uint16_t *x1, int C1, const uint16_t *x2, int C2, const void *Wp,
                                        const float *scale, const float *shift, int relu, const int32_t *ids, const int32_t *row_max,
                                        int32_t *out_idx, float *out_val, int B, int Cout, int L, int M, sonet_stream_t stream)
{
    return bf16_pool_impl("sonet_pointmlp_bf16_pool", x1, C1, x2, C2, Wp, scale, shift, relu, ids, row_max, out_idx, out_val, B, Cout, L, M, stream, nullptr);
}

/* sonet_pointmlp_bf16_pool with normalise-on-load of both input panels (see sonet_pointmlp_bf16_stats_xaff): the last layer of the first
 * PointNet reading the RAW outputs of the first and third layer (models/layers.py:431).  Same positions and values, bit for bit, as the
 * plain launch on the normalised tensors. */
extern "C" int sonet_pointmlp_bf16_pool_xaff(const uint16_t *x1, int C1, const uint16_t *x2, int C2, const void *Wp,
                                             const float *scale, const float *shift, int relu, const int32_t *ids, const int32_t *row_max,
                                             int32_t *out_idx, float *out_val, int B, int Cout, int L, int M,
                                             const float *xs1, const float *xh1, const float *xs2, const float *xh2, int xrelu,
                                             sonet_stream_t stream)
{
    const char *what = "sonet_pointmlp_bf16_pool_xaff";
    SONET_REQUIRE(xs1 && xh1 && (C2 == 0 || (xs2 && xh2)), "%s: NULL pointer", what);
    const BfrXaff xa = {xs1, xh1, xs2, xh2, xrelu};
    return bf16_pool_impl(what, x1, C1, x2, C2, Wp, scale, shift, relu, ids, row_max, out_idx, out_val, B, Cout, L, M, stream, &xa);
}

extern "C" int sonet_pointmlp_bf16(const uint16_t *x1, int C1, const uint16_t *x2, int C2, const void *Wp,
                                   const float *scale, const float *shift, int relu, uint16_t *y,
                                   int B, int Cout, int L, sonet_stream_t stream)
{
    return bf16_run_impl("sonet_pointmlp_bf16", x1, C1, x2, C2, Wp, scale, shift, relu, y, B, Cout, L, stream);
}

/* sonet_pointmlp_bf16 with an ACCUMULATING store: y = bf16(float(bf16(result)) + float(yadd)), yadd [B][Cout][L] bf16 -- another gradient of the
 * same tensor, computed earlier (models/layers.py:417-431: the first layer's output feeds the second layer and the last one); exactly what
 * autograd's accumulation of the two bf16 tensors would store, without its pass over three tensors.  yadd == y is allowed (every element is
 * read and written by one lane).  Even L, 4-byte aligned rows. */
extern "C" int sonet_pointmlp_bf16_acc(const uint16_t *x1, int C1, const uint16_t *x2, int C2, const void *Wp,
                                       const float *scale, const float *shift, int relu, const uint16_t *yadd, uint16_t *y,
                                       int B, int Cout, int L, sonet_stream_t stream)
{
    SONET_REQUIRE(yadd, "sonet_pointmlp_bf16_acc: NULL pointer");
    if ((reinterpret_cast<uintptr_t>(yadd) & 3) != 0) return sonet::fail(SONET_ERR_INVALID_ARG, "sonet_pointmlp_bf16_acc: yadd must be 4-byte aligned");
    return bf16_run_impl("sonet_pointmlp_bf16_acc", x1, C1, x2, C2, Wp, scale, shift, relu, y, B, Cout, L, stream, nullptr, 0, nullptr, nullptr, nullptr,
                         nullptr, yadd);
}

/* The input gradient of a bf16 layer behind a training-mode BatchNorm (+ ReLU) with the BatchNorm / ReLU backward applied by the operand load
 * (models/layers.py:60-70, :282-296; the bf16 twin of sonet_pointmlp_x3_bnb_f32 / _bnb_acc_f32):
 *   y = bf16((W . g_raw) * scale + shift) [+ yadd],   g_raw[k] = bf16(a[k] * (relu && !(raw * sc[k] + sh[k] > 0) ? 0 : gy) + b[k] * raw + c0[k])
 * gy, raw [B][C][L] bf16, coefficients [C] f32 (sonet_bn_bwd_coeffs_f32's a, b, c0; the forward's normalisation sc, sh) -- what
 * sonet_pointwise_bwd_apply_bf16 followed by sonet_pointmlp_bf16 (or _acc) computes, bit for bit, in ONE pass over (gy, raw).
 * g_raw_out (or NULL) receives g_raw for the weight gradient; yadd (or NULL): another gradient of the same tensor, added by the store.
 * Wp: the bf16 pack of the C x Cout matrix.  Needs even L, 4-byte aligned rows, C % 16 == 0, 32 <= C <= 512, Cout % 64 == 0. */
extern "C" int sonet_pointmlp_bf16_bnb(const uint16_t *gy, const uint16_t *raw, int C, const void *Wp, const float *scale, const float *shift,
                                       const float *a, const float *b, const float *c0, const float *sc, const float *sh, int relu,
                                       uint16_t *g_raw_out, const uint16_t *yadd, uint16_t *y, int B, int Cout, int L, sonet_stream_t stream)
{
    const char *what = "sonet_pointmlp_bf16_bnb";
    SONET_REQUIRE(gy && raw && a && b && c0 && sc && sh && y, "%s: NULL pointer", what);
    if (yadd && (reinterpret_cast<uintptr_t>(yadd) & 3) != 0) return sonet::fail(SONET_ERR_INVALID_ARG, "%s: yadd must be 4-byte aligned", what);
    if (C < 32) return sonet::fail(SONET_ERR_UNSUPPORTED, "%s: C=%d < 32", what, C);
    const BfBnb bn = {raw, a, b, c0, sc, sh, g_raw_out, relu};
    return bf16_run_impl(what, gy, C, nullptr, 0, Wp, scale, shift, 0, y, B, Cout, L, stream, nullptr, 0, nullptr, nullptr, nullptr, nullptr, yadd, &bn);
}

extern "C" int sonet_pointmlp_bf16_gather(const uint16_t *x1, int C1, int L1, const int32_t *gidx, const uint16_t *x2, int C2, const void *Wp,
                                          const float *scale, const float *shift, int relu, uint16_t *y,
                                          int B, int Cout, int L, sonet_stream_t stream)
{
    const char *what = "sonet_pointmlp_bf16_gather";
    SONET_REQUIRE(gidx, "%s: NULL pointer", what);
    return bf16_run_impl(what, x1, C1, x2, C2, Wp, scale, shift, relu, y, B, Cout, L, stream, gidx, L1);
}
