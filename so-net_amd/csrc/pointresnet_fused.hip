// pointresnet_fused.hip -- the whole first PointNet of the encoder as ONE kernel (eval mode).
//
// Replaces the four EquivariantLayer launches of PointResNet.forward (models/layers.py:419-432, built at
// models/networks.py:82-83 as 6 -> 64 -> 128 -> 256 -> [64 + 256] -> 384 with BN + ReLU on the first
// three layers) when BatchNorm runs on its running statistics.  Per 32-point tile a wave keeps every
// intermediate activation in registers; HBM sees only the 6-channel input and the 384-channel output
// (the unfused path writes and re-reads 64 + 128 + 256 channels per point: 3.6 KB / point).
//
// Arithmetic: the 3 x bf16 split scheme of pointmlp_x3.hip (6 bf16 MFMAs per product set, f32 accumulate).
//
// Register chaining.  v_mfma_f32_32x32x16_bf16 produces D[row = (r&3) + 8*(r>>2) + 4*(l>>5)][col = l&31]
// in register r of lane l and consumes B[k = 8*(l>>5) + e][col = l&31], e = 0..7.  Registers 8q .. 8q+7 of
// an output tile therefore ARE the B operand of a 16-channel chunk of the next layer (after the affine +
// ReLU and the bf16 split), provided the next layer's weights are packed with the matching channel order
//     k = 8h + e   <->   channel 32*t + 16*q + (e&3) + 8*(e>>2) + 4*h          ("chained" packing)
// No shuffle, no LDS round trip, no transposition between layers.
//
// Weights.  All four layers are packed (pointresnet_pack_kernel) into ONE linear stream of 1-KiB slices
// (64 lanes x 8 bf16) in exactly the order the MFMAs consume them, so W staging is a linear copy:
// the 4 waves of a workgroup load the next-next stage (NSTG slices) into registers, ds_write it after the
// stage barrier, and read their A fragments back at (ring slot) + compile-time offsets.  One barrier per 72 MFMAs.
//   L1: 2 tiles x 1 chunk, L2: 4 x 4, L3: 8 x 8  (tile-major),  L4: 2 passes x 20 chunks x 6 tiles.
// Workgroups are persistent (one per CU) and walk the 128-point tiles; the weight stream simply restarts.
#include "common.hpp"
#include <stdlib.h>

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef int i32x4_t __attribute__((ext_vector_type(4)));

constexpr int PF_THREADS = 256, PF_WAVES = 4;
constexpr int T0 = 2, T1 = 4, T2 = 8, T3 = 12;               // output tiles (x32 channels) of the four layers
constexpr int KC1 = 1, KC2 = 2 * T0, KC3 = 2 * T1, KC4 = 2 * T0 + 2 * T2;   // 16-channel K chunks per layer
constexpr int MT4 = 6, NPASS = T3 / MT4;                      // layer-4 cout tiles per accumulator pass
constexpr int NSTG = 36;                                       // slices per LDS stage
constexpr int GS = 4;                                          // cout tiles processed together in layers 2 and 3
constexpr int SL1 = 12 /* 6 used + 6 pad: keeps every step aligned */, SL2 = T1 * KC2 * 3, SL3 = T2 * KC3 * 3;
constexpr int OFF1 = 0, OFF2 = SL1, OFF3 = SL1 + SL2;
constexpr int PRE = ((SL1 + SL2 + SL3 + NSTG - 1) / NSTG) * NSTG;                 // layer 4 starts on a stage boundary
static_assert(OFF2 % (3 * GS) == 0 && OFF3 % (3 * GS) == 0 && NSTG % (3 * GS) == 0 && NSTG % (3 * MT4) == 0 && T1 % GS == 0 && T2 % GS == 0,
              "a step (one K chunk x a group of tiles) must never straddle a stage boundary");
constexpr int SL4 = KC4 * MT4 * 3;                             // slices per layer-4 pass
constexpr int NSLICE = PRE + NPASS * SL4;
constexpr int NSTAGE = NSLICE / NSTG;
static_assert(SL4 % NSTG == 0 && NSLICE % NSTG == 0 && PRE == SL1 + SL2 + SL3, "whole stages, no padding stage");
constexpr int NSW = NSTG / PF_WAVES;                           // slices staged per wave
static_assert(NSTG % PF_WAVES == 0, "");
constexpr int CH_TOTAL = 32 * (T0 + T1 + T2 + T3);

__device__ __forceinline__ unsigned cvt_pk_bf16(float lo, float hi) {
    unsigned r;
    asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(r) : "v"(lo), "v"(hi));
    return r;
}
__device__ __forceinline__ void split3_pair(float x0, float x1, unsigned &h, unsigned &m, unsigned &l) {
    h = cvt_pk_bf16(x0, x1);
    const float r0 = x0 - __uint_as_float(h << 16), r1 = x1 - __uint_as_float(h & 0xFFFF0000u);
    m = cvt_pk_bf16(r0, r1);
    const float q0 = r0 - __uint_as_float(m << 16), q1 = r1 - __uint_as_float(m & 0xFFFF0000u);
    l = cvt_pk_bf16(q0, q1);
}

struct B3 { bf16x8 h, m, l; };
template <int ABL> __device__ __forceinline__ B3 split_chunk_abl(const float (&v)[8]);
__device__ __forceinline__ B3 split_chunk(const float (&v)[8]) {
    unsigned bh[4], bm[4], bl[4];
#pragma unroll
    for (int p = 0; p < 4; ++p) split3_pair(v[2 * p], v[2 * p + 1], bh[p], bm[p], bl[p]);
    B3 b;
    b.h = __builtin_bit_cast(bf16x8, make_uint4(bh[0], bh[1], bh[2], bh[3]));
    b.m = __builtin_bit_cast(bf16x8, make_uint4(bm[0], bm[1], bm[2], bm[3]));
    b.l = __builtin_bit_cast(bf16x8, make_uint4(bl[0], bl[1], bl[2], bl[3]));
    return b;
}

template <int ABL> __device__ __forceinline__ B3 split_chunk_abl(const float (&v)[8]) {
    if constexpr (ABL & 2) {
        B3 b;
        const unsigned u = __float_as_uint(v[0]);
        b.h = __builtin_bit_cast(bf16x8, make_uint4(u, u, u, u)); b.m = b.h; b.l = b.h;
        return b;
    } else {
        return split_chunk(v);
    }
}

// ---- weight stream packing -------------------------------------------------------------------------
// slice s of the stream -> (layer, cout tile, K chunk, split term); one thread per (slice, lane).
__global__ __launch_bounds__(256) void pointresnet_pack_kernel(const float *__restrict__ W1, const float *__restrict__ W2,
                                                                const float *__restrict__ W3, const float *__restrict__ W4,
                                                                int Cin0, uint4 *__restrict__ out)
{
    const int t = blockIdx.x * 256 + threadIdx.x;
    if (t >= NSLICE * 64) return;
    const int lane = t & 63, s = t >> 6;
    const int i = lane & 31, h = lane >> 5;
    const float *W = nullptr;
    int Cin = 0, ct = 0, kc = 0, term = 0;
    bool chained = true, valid = true;
    // consumption order: per layer, tile-group major, then K chunk, then tile within the group, then split term
    if (s < OFF2) {                       // L1: 2 tiles x 1 chunk, standard channel order (input comes from memory)
        const int u = s - OFF1; term = u % 3; kc = 0; ct = u / 3; W = W1; Cin = Cin0; chained = false; valid = u < T0 * 3;
    } else if (s < OFF3) {
        const int u = s - OFF2; term = u % 3; const int mt = (u / 3) % GS; kc = (u / (3 * GS)) % KC2;
        ct = (u / (3 * GS * KC2)) * GS + mt; W = W2; Cin = 32 * T0;
    } else if (s < OFF3 + SL3) {
        const int u = s - OFF3; term = u % 3; const int mt = (u / 3) % GS; kc = (u / (3 * GS)) % KC3;
        ct = (u / (3 * GS * KC3)) * GS + mt; W = W3; Cin = 32 * T1;
    } else if (s < PRE) {
        valid = false;                    // padding up to an even number of stages
    } else {                              // L4: pass-major, then chunk-major, MT4 tiles per chunk
        const int u = (s - PRE) % SL4, pass = (s - PRE) / SL4;
        term = u % 3; ct = pass * MT4 + (u / 3) % MT4; kc = u / (3 * MT4); W = W4; Cin = 32 * (T0 + T2);
    }
    unsigned w[4] = {0, 0, 0, 0};
    if (valid) {
#pragma unroll
        for (int p = 0; p < 4; ++p) {
            float v[2];
#pragma unroll
            for (int z = 0; z < 2; ++z) {
                const int e = 2 * p + z;
                const int c = chained ? kc * 16 + (e & 3) + 8 * (e >> 2) + 4 * h : kc * 16 + 8 * h + e;
                v[z] = c < Cin ? W[(long long)(ct * 32 + i) * Cin + c] : 0.f;
            }
            unsigned hh, mm, ll;
            split3_pair(v[0], v[1], hh, mm, ll);
            w[p] = term == 0 ? hh : term == 1 ? mm : ll;
        }
    }
    out[(long long)s * 64 + lane] = make_uint4(w[0], w[1], w[2], w[3]);
}

// ---- the fused kernel ------------------------------------------------------------------------------
// LDS ring of NSLOT stages.  Before stage n is consumed, stages n and n+1 are resident and visible, the
// registers hold stage n+2.  Boundary(n), run by every wave at the first step of stage n:
//     barrier;  ds_write stage n+2 into the slot stage n-1 occupied;  load stage n+3 into the registers.
// So while a wave computes the last step of stage n it may already read the A fragments of the first step
// of stage n+1: every step prefetches the NEXT step's A fragments (LDS -> registers) and splits the next
// step's B chunk before issuing its own MFMAs.  With one wave per SIMD nothing else hides those latencies.
constexpr int NSLOT = 3;

struct AF { bf16x8 h[MT4], m[MT4], l[MT4]; };                  // A fragments of one step (up to MT4 tiles x 3 terms)

template <int ABL>   // bench-only ablation: 1 = no stores, 2 = no bf16 split (constant B), 4 = no W streaming / barriers
__global__ __launch_bounds__(PF_THREADS, 1) void pointresnet_fused_kernel(
    const float *__restrict__ x, int Cin0, const uint4 *__restrict__ Wst, const float2 *__restrict__ affine_g /*[CH_TOTAL] (scale, shift); last layer (1, bias)*/,
    float *__restrict__ y, int L, int tpc /*128-point tiles per cloud*/, long long ntiles)
{
    __shared__ uint4 wsm[NSLOT * NSTG][64];                    // 3 x 36 KiB
    __shared__ float2 aff[CH_TOTAL];

    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int j = lane & 31, h = lane >> 5;
    for (int c = threadIdx.x; c < CH_TOTAL; c += PF_THREADS) aff[c] = affine_g[c];

    const __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint4 *>(Wst), 0, NSLICE * 1024, 0x00020000);
    const unsigned vow = (unsigned)lane * 16u;
    const unsigned rowB = (unsigned)L * 4u;

    i32x4_t wreg[NSW];
    auto stage_load = [&](int n) {                              // stream stage n (wraps: the stream restarts per tile)
        const int sn = n % NSTAGE;
#pragma unroll
        for (int t = 0; t < NSW; ++t)
            wreg[t] = __builtin_amdgcn_raw_buffer_load_b128(rw, vow, (unsigned)(sn * NSTG + wave + t * PF_WAVES) * 1024u, 0);
    };
    auto stage_write = [&](int slot) {
#pragma unroll
        for (int t = 0; t < NSW; ++t) wsm[slot * NSTG + wave + t * PF_WAVES][lane] = __builtin_bit_cast(uint4, wreg[t]);
    };
    // ring state (wave-uniform scalars)
    int n_cur = 0;                                              // stage being consumed
    int slot_cur = 0, slot_nxt = 1, slot_fill = 2;
    stage_load(0); stage_write(0);
    stage_load(1); stage_write(1);
    stage_load(2);
    const uint4 *lds_cur = &wsm[slot_cur * NSTG][lane];
    const uint4 *lds_nxt = &wsm[slot_nxt * NSTG][lane];
    bool first_boundary = true;

    // boundary of the stage that the CURRENT step opens
    auto boundary = [&]() {
        if constexpr (ABL & 4) return;
        __syncthreads();
        if (!first_boundary) {                                  // rotate: the stage just finished becomes the fill slot
            const int t = slot_cur; slot_cur = slot_nxt; slot_nxt = slot_fill; slot_fill = t;
            n_cur += 1;
        }
        first_boundary = false;
        stage_write(slot_fill);                                 // stage n_cur + 2
        stage_load(n_cur + 3);
        lds_cur = &wsm[slot_cur * NSTG][lane];
        lds_nxt = &wsm[slot_nxt * NSTG][lane];
    };
    // A fragments of the step at slice index sidx (compile-time); `ahead` = the step opens a new stage and is
    // being prefetched from the stage before it
#define PF_LOAD_A(af, NT, sidx, ahead)                                                               \
    {                                                                                                \
        const uint4 *base_ = (ahead) ? lds_nxt : lds_cur;                                            \
        _Pragma("unroll") for (int u_ = 0; u_ < NT; ++u_) {                                          \
            af.h[u_] = __builtin_bit_cast(bf16x8, base_[(((sidx) % NSTG) + 3 * u_ + 0) * 64]);        \
            af.m[u_] = __builtin_bit_cast(bf16x8, base_[(((sidx) % NSTG) + 3 * u_ + 1) * 64]);        \
            af.l[u_] = __builtin_bit_cast(bf16x8, base_[(((sidx) % NSTG) + 3 * u_ + 2) * 64]);        \
        }                                                                                            \
    }
    // six product terms, TERM-major across the NT tiles (consecutive MFMAs never share an accumulator)
#define PF_MFMAS(accarr, tbase, NT, af, b)                                                           \
    {                                                                                                \
        _Pragma("unroll") for (int u_ = 0; u_ < NT; ++u_) accarr[(tbase) + u_] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af.l[u_], b.h, accarr[(tbase) + u_], 0, 0, 0); \
        _Pragma("unroll") for (int u_ = 0; u_ < NT; ++u_) accarr[(tbase) + u_] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af.h[u_], b.l, accarr[(tbase) + u_], 0, 0, 0); \
        _Pragma("unroll") for (int u_ = 0; u_ < NT; ++u_) accarr[(tbase) + u_] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af.m[u_], b.m, accarr[(tbase) + u_], 0, 0, 0); \
        _Pragma("unroll") for (int u_ = 0; u_ < NT; ++u_) accarr[(tbase) + u_] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af.m[u_], b.h, accarr[(tbase) + u_], 0, 0, 0); \
        _Pragma("unroll") for (int u_ = 0; u_ < NT; ++u_) accarr[(tbase) + u_] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af.h[u_], b.m, accarr[(tbase) + u_], 0, 0, 0); \
        _Pragma("unroll") for (int u_ = 0; u_ < NT; ++u_) accarr[(tbase) + u_] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af.h[u_], b.h, accarr[(tbase) + u_], 0, 0, 0); \
    }
#define PF_AFFINE_RELU(accv, chbase)                                                                 \
    {                                                                                                \
        _Pragma("unroll") for (int r = 0; r < 16; ++r) {                                             \
            const float2 ss = aff[(chbase) + (r & 3) + 8 * (r >> 2) + 4 * h];                        \
            const float v = __fmaf_rn(accv[r], ss.x, ss.y);                                          \
            accv[r] = v < 0.f ? 0.f : v;                                                             \
        }                                                                                            \
    }
#define PF_CHUNK(dst, arr, kc)                                                                       \
    { _Pragma("unroll") for (int e = 0; e < 8; ++e) dst[e] = arr[(kc) >> 1][8 * ((kc) & 1) + e]; }

    constexpr int NMID = KC2 * (T1 / GS) + KC3 * (T2 / GS);    // steps of layers 2 and 3 (4 + 16)
    // slice index / tile group of middle step i (layer 2 first, then layer 3)
#define MID_SIDX(i) ((i) < KC2 * (T1 / GS) ? OFF2 + (i) * 3 * GS : OFF3 + ((i) - KC2 * (T1 / GS)) * 3 * GS)

    for (long long tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const long long b = tile / tpc;
        const int l0 = (int)(tile - b * tpc) * 128 + wave * 32;
        const bool pv = l0 + j < L;
        const int lc = pv ? l0 + j : (l0 < L ? l0 : 0);
        const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc(
            const_cast<float *>(x + b * (long long)Cin0 * L), 0, (int)((unsigned)Cin0 * rowB), 0x00020000);
        const __amdgpu_buffer_rsrc_t ry = __builtin_amdgcn_make_buffer_rsrc(
            y + b * (long long)(32 * T3) * L, 0, (int)((unsigned)(32 * T3) * rowB), 0x00020000);

        float xin[8];
#pragma unroll
        for (int e = 0; e < 8; ++e)
            xin[e] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rx, (unsigned)(8 * h * L + lc) * 4u, (unsigned)e * rowB, 0));
        f32x16 act1[T0], act2[T1], act3[T2];
#pragma unroll
        for (int t = 0; t < T0; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) act1[t][r] = 0.f;
#pragma unroll
        for (int t = 0; t < T1; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) act2[t][r] = 0.f;
#pragma unroll
        for (int t = 0; t < T2; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) act3[t][r] = 0.f;

        AF af[2];
        B3 bq[2];
        // ---- layer 1 (slice 0 opens stage 0 of this tile): no prefetch into it, prefetches layer 2's first step ----
        boundary();
        PF_LOAD_A(af[0], T0, OFF1, false)
        bq[0] = split_chunk_abl<ABL>(xin);
        PF_LOAD_A(af[1], GS, MID_SIDX(0), (MID_SIDX(0) % NSTG) == 0)
        PF_MFMAS(act1, 0, T0, af[0], bq[0])
#pragma unroll
        for (int t = 0; t < T0; ++t) PF_AFFINE_RELU(act1[t], 32 * t)
        {
            float v[8];
            PF_CHUNK(v, act1, 0)
            bq[1] = split_chunk_abl<ABL>(v);
        }
        // ---- layers 2 and 3: middle steps i = 0 .. NMID-1, set parity (i + 1) & 1 ----
#define PF_MID(I_)                                                          \
        {                                                                   \
            constexpr int i = (I_);                                         \
            const int sidx = MID_SIDX(i); \
            const int cur = (i + 1) & 1, nxt = i & 1; \
            const bool l2 = i < KC2 * (T1 / GS); \
            const int kc = l2 ? i % KC2 : (i - KC2 * (T1 / GS)) % KC3; \
            const int grp = l2 ? i / KC2 : (i - KC2 * (T1 / GS)) / KC3; \
            if (sidx % NSTG == 0) boundary(); \
 \
            if (i + 1 < NMID) { \
                PF_LOAD_A(af[nxt], GS, MID_SIDX(i + 1), (MID_SIDX(i + 1) % NSTG) == 0) \
            } else { \
                PF_LOAD_A(af[nxt], MT4, PRE, (PRE % NSTG) == 0) \
            } \
 \
            const bool last_of_l2 = (i == KC2 * (T1 / GS) - 1), last_of_l3 = (i == NMID - 1); \
            if (!last_of_l2 && !last_of_l3) { \
                float v[8]; \
                const int kcn = l2 ? (i + 1) % KC2 : (i + 1 - KC2 * (T1 / GS)) % KC3; \
                if (l2) PF_CHUNK(v, act1, (l2 ? kcn : 0)) else PF_CHUNK(v, act2, (l2 ? 0 : kcn)) \
                bq[nxt] = split_chunk_abl<ABL>(v); \
            } \
            if (l2) PF_MFMAS(act2, grp * GS, GS, af[cur], bq[cur]) else PF_MFMAS(act3, grp * GS, GS, af[cur], bq[cur]) \
            if (last_of_l2) { \
_Pragma("unroll") \
                for (int t = 0; t < T1; ++t) PF_AFFINE_RELU(act2[t], 32 * T0 + 32 * t) \
                float v[8]; \
                PF_CHUNK(v, act2, 0) \
                bq[nxt] = split_chunk_abl<ABL>(v); \
            } \
            if (last_of_l3) { \
_Pragma("unroll") \
                for (int t = 0; t < T2; ++t) PF_AFFINE_RELU(act3[t], 32 * (T0 + T1) + 32 * t) \
                float v[8]; \
                PF_CHUNK(v, act1, 0) \
                bq[nxt] = split_chunk_abl<ABL>(v); \
            } \
        }
        static_assert(NMID == 20, "expand PF_MID to NMID steps");
        PF_MID(0) PF_MID(1) PF_MID(2) PF_MID(3) PF_MID(4) PF_MID(5) PF_MID(6) PF_MID(7) PF_MID(8) PF_MID(9) PF_MID(10) PF_MID(11) PF_MID(12) PF_MID(13) PF_MID(14) PF_MID(15) PF_MID(16) PF_MID(17) PF_MID(18) PF_MID(19)
#undef PF_MID
        // ---- layer 4: NPASS passes x KC4 steps of MT4 tiles; set parity of step kc is (NMID + 1 + kc) & 1 ----
        for (int pass = 0; pass < NPASS; ++pass) {
            f32x16 acc[MT4];
#pragma unroll
            for (int mt = 0; mt < MT4; ++mt)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[mt][r] = 0.f;
#define PF_L4(K_)                                                           \
            {                                                               \
                constexpr int kc = (K_);                                    \
                const int sidx = PRE + kc * MT4 * 3; \
                const int cur = (NMID + 1 + kc) & 1, nxt = cur ^ 1; \
                if (sidx % NSTG == 0) boundary(); \
                const int kn = (kc + 1) % KC4; \
                const int sn = PRE + kn * MT4 * 3; \
                PF_LOAD_A(af[nxt], MT4, sn, (sn % NSTG) == 0) \
                { \
                    float v[8]; \
                    if (kn < KC2) PF_CHUNK(v, act1, (kn < KC2 ? kn : 0)) else PF_CHUNK(v, act3, (kn < KC2 ? 0 : kn - KC2)) \
                    bq[nxt] = split_chunk_abl<ABL>(v); \
                } \
                PF_MFMAS(acc, 0, MT4, af[cur], bq[cur]) \
            }
            static_assert(KC4 == 20, "expand PF_L4 to KC4 steps");
            PF_L4(0) PF_L4(1) PF_L4(2) PF_L4(3) PF_L4(4) PF_L4(5) PF_L4(6) PF_L4(7) PF_L4(8) PF_L4(9) PF_L4(10) PF_L4(11) PF_L4(12) PF_L4(13) PF_L4(14) PF_L4(15) PF_L4(16) PF_L4(17) PF_L4(18) PF_L4(19)
#undef PF_L4
            if (pv) {
#pragma unroll
                for (int mt = 0; mt < MT4; ++mt) {
                    const int ct = pass * MT4 + mt;
                    const unsigned so_tile = (unsigned)(ct * 32) * rowB;
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int orow = (r & 3) + 8 * (r >> 2);
                        const float2 ss = aff[32 * (T0 + T1 + T2) + ct * 32 + orow + 4 * h];
                        const float v = __fmaf_rn(acc[mt][r], ss.x, ss.y);
                        if constexpr (ABL & 1) { asm volatile("" ::"v"(v)); } else
                        __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(int, v), ry, (unsigned)(4 * h * L + lc) * 4u,
                                                              so_tile + (unsigned)orow * rowB, 0);
                    }
                }
            }
        }
    }
#undef PF_LOAD_A
#undef PF_MFMAS
#undef PF_AFFINE_RELU
#undef PF_CHUNK
#undef MID_SIDX
}

}  // namespace

extern "C" size_t sonet_pointresnet_pack_size(void) { return (size_t)NSLICE * 1024; }

extern "C" int sonet_pointresnet_pack(const float *W1, const float *W2, const float *W3, const float *W4, int Cin0,
                                      void *stream_out, sonet_stream_t stream)
{
    const char *what = "sonet_pointresnet_pack";
    SONET_REQUIRE(W1 && W2 && W3 && W4 && stream_out, "%s: NULL pointer", what);
    SONET_REQUIRE(Cin0 >= 1 && Cin0 <= 16, "%s: Cin0=%d must be in [1, 16]", what, Cin0);
    hipLaunchKernelGGL(pointresnet_pack_kernel, dim3(sonet::ceil_div(NSLICE * 64, 256)), dim3(256), 0, sonet::as_stream(stream),
                       W1, W2, W3, W4, Cin0, reinterpret_cast<uint4 *>(stream_out));
    return sonet::launched(what);
}

extern "C" int sonet_pointresnet_fused_f32(const float *x, int Cin0, const void *wstream, const float *affine,
                                           float *y, int B, int L, sonet_stream_t stream)
{
    const char *what = "sonet_pointresnet_fused_f32";
    SONET_REQUIRE(x && wstream && affine && y, "%s: NULL pointer", what);
    SONET_REQUIRE(B > 0 && L > 0 && Cin0 >= 1 && Cin0 <= 16, "%s: bad size B=%d L=%d Cin0=%d", what, B, L, Cin0);
    if ((double)(32 * T3) * L * 4.0 >= 4.0e9) return sonet::fail(SONET_ERR_UNSUPPORTED, "%s: a per-cloud panel exceeds 4 GiB", what);
    const int tpc = sonet::ceil_div(L, 128);
    const long long ntiles = (long long)B * tpc;
    int dev = 0, cus = 256;
    if (hipGetDevice(&dev) == hipSuccess) {
        int v = 0;
        if (hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && v > 0) cus = v;
    }
    const long long grid = ntiles < cus ? ntiles : cus;        // persistent: one workgroup per CU
    int abl = 0;
    if (const char *e = getenv("SONET_FUSED_ABLATE")) abl = atoi(e);       // bench-only (tools/microbench.py)
#define PF_LAUNCH(AA) hipLaunchKernelGGL(pointresnet_fused_kernel<AA>, dim3((unsigned)grid), dim3(PF_THREADS), 0, sonet::as_stream(stream), \
                       x, Cin0, reinterpret_cast<const uint4 *>(wstream), reinterpret_cast<const float2 *>(affine), y, L, tpc, ntiles)
    switch (abl) { case 1: PF_LAUNCH(1); break; case 2: PF_LAUNCH(2); break; case 4: PF_LAUNCH(4); break; case 7: PF_LAUNCH(7); break; default: PF_LAUNCH(0); }
#undef PF_LAUNCH
    return sonet::launched(what);
}
