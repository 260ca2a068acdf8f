// pointresnet_bf16.hip -- the whole first PointNet of the encoder as ONE kernel in bf16 (eval mode).
//
// bf16 twin of pointresnet_fused.hip (models/layers.py:419-432 as built at models/networks.py:82-83:
// Cin0 -> 64 -> 128 -> 256 -> [64 + 256] -> 384, BatchNorm + ReLU on the first three layers, running statistics):
// ONE v_mfma_f32_32x32x16_bf16 per product where the f32-class kernel issues three fp16 MFMAs, activations rounded to bf16
// between the layers (exactly what the layer-wise bf16 kernels of pointmlp_bf16.hip store), f32 accumulation.  HBM sees
// the f32 input (Cin0 rows) and the bf16 output only.
//
// Wave tile = 64 points (two 32-column MFMA tiles: the EVEN and the ODD points, so that an output dword is one
// v_cvt_pk_bf16_f32 of the two accumulators, see pointmlp_bf16.hip) through all four layers.  Register chaining as in the
// f32-class kernel: registers 8q..8q+7 of an output tile are the B operand of a 16-channel chunk of the next layer once the
// weights are packed in the matching channel order (k = 8h+e <-> channel 32t + 16q + (e&3) + 8(e>>2) + 4h) -- but here the
// activations are kept ALREADY CONVERTED (affine + ReLU + bf16 pack done once, 4 registers per chunk and tile instead of
// 8 raw accumulators re-split at every use): act1 32 + act3 128 registers feed layer 4, whose 12 output tiles go by in 4
// groups of 3 (96 accumulators).
//
// Weights: one linear stream of 1-KiB slices in consumption order, cut into 7 GROUPS per 256-point tile
//     G0 = layer 1 (2 slices + 2 pad) + layer 2 (4 chunks x 4 tiles),  G1, G2 = layer 3 tiles 0-3 / 4-7 (8 x 4),
//     G3..G6 = layer 4 tiles 3g..3g+2 (20 chunks x 3)
// and two 60-KiB LDS buffers: while a group is multiplied out of one buffer, the 4 waves stage the next group into the other
// (buffer_load -> registers -> ds_write two K chunks later).  One barrier per group, 7 per tile, 644 MFMAs per wave between.
// The K loops are straight-line code with a scheduling barrier per chunk: the A fragments of chunk k+1 are requested before
// the MFMAs of chunk k, every wait is counted (see pointmlp_bf16.hip, the X-in-registers kernel, for what hipcc does
// otherwise).
#include "common.hpp"
#include <stdlib.h>
#include <type_traits>
#include <utility>

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef int i32x4_t __attribute__((ext_vector_type(4)));

constexpr int FB_THREADS = 256, FB_WAVES = 4;
constexpr int G0_SL = 20, L3G_SL = 32, L4G_SL = 60;            // slices per group
constexpr int OFF_L1 = 0, OFF_L2 = 4, OFF_L3 = G0_SL, OFF_L4 = G0_SL + 2 * L3G_SL;
constexpr int NSLICE_BF = OFF_L4 + 4 * L4G_SL;                  // 324
constexpr int NSLICE_PAD = 2;                                   // zero slices after the stream (the pool kernel stages 32-slice groups: the last half of layer 4 is 30)
constexpr int BUF_SL = 60;
constexpr int CH_TOTAL_BF = 64 + 128 + 256 + 384;
constexpr int AFF_L1 = 0, AFF_L2 = 64, AFF_L3 = 192, AFF_L4 = 448;

template <int N> struct IC { static constexpr int value = N; };
typedef short i16x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ unsigned fb_relu_pk(unsigned pk) {
    const i16x2 v = __builtin_bit_cast(i16x2, pk), z = {0, 0};
    return __builtin_bit_cast(unsigned, __builtin_elementwise_max(v, z));
}

__device__ __forceinline__ unsigned fb_cvt_pk_bf16(float lo, float hi) {
    unsigned r;
    asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(r) : "v"(lo), "v"(hi));
    return r;
}

// slice s of the stream -> (layer, cout tile, K chunk); one thread per (slice, lane).  "chained" layers take their input
// channels in the order the producing layer's accumulator registers hold them.
__global__ __launch_bounds__(256) void pointresnet_bf16_pack_kernel(const float *__restrict__ W1, const float *__restrict__ W2,
                                                                     const float *__restrict__ W3, const float *__restrict__ W4,
                                                                     int Cin0, uint4 *__restrict__ out)
{
    const int t = blockIdx.x * 256 + threadIdx.x;
    if (t >= (NSLICE_BF + NSLICE_PAD) * 64) return;
    if (t >= NSLICE_BF * 64) { out[t] = make_uint4(0u, 0u, 0u, 0u); return; }
    const int lane = t & 63, s = t >> 6;
    const int i = lane & 31, h = lane >> 5;
    const float *W = nullptr;
    int Cin = 0, ct = 0, kc = 0;
    bool chained = true, valid = true;
    if (s < OFF_L2) {                          // layer 1: input from memory, standard channel order; slices 2, 3 are padding
        ct = s; kc = 0; W = W1; Cin = Cin0; chained = false; valid = s < 2;
    } else if (s < OFF_L3) {                   // layer 2: chunk-major, 4 tiles per chunk
        const int u = s - OFF_L2; kc = u / 4; ct = u % 4; W = W2; Cin = 64;
    } else if (s < OFF_L4) {                   // layer 3: two groups of 4 tiles
        const int u = s - OFF_L3, grp = u / L3G_SL, v = u % L3G_SL; kc = v / 4; ct = grp * 4 + v % 4; W = W3; Cin = 128;
    } else {                                   // layer 4: four groups of 3 tiles; chunks 0-3 = act1 (the skip), 4-19 = act3
        const int u = s - OFF_L4, grp = u / L4G_SL, v = u % L4G_SL; kc = v / 3; ct = grp * 3 + v % 3; W = W4; Cin = 320;
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
            w[p] = fb_cvt_pk_bf16(v[0], v[1]);
        }
    }
    out[(long long)s * 64 + lane] = make_uint4(w[0], w[1], w[2], w[3]);
}

// ---- one pass: MTn cout tiles x KCn K chunks out of an LDS buffer, optionally staging the next group -----------------
// lds_cur: &buffer[first slice of the pass][lane]; slice (kc, mt) at (kc * MTn + mt) * 64.
// Staging (STAGE): this wave moves NSTG_W slices of the next group (stream slices idx * 4 + wave), PER per K chunk:
// buffer_load at chunk kc, ds_write at chunk kc + 2.  bfrag(IC<kc>, Ba, Bb) supplies the two B fragments of chunk kc.
// SWAP: X as the A operand and W as B (the per-lane register contents of both are the same either way): the accumulator
// tile comes out transposed, rows = points, columns = channels -- what the per-node max-pool epilogue wants.
template <int KCn, int MTn, int PER, int NSTG_W, bool STAGE, bool SWAP = false, typename BF>
__device__ __forceinline__ void mfma_pass(f32x16 (&acc)[MTn][2], const uint4 *lds_cur, uint4 *lds_nxt_w,
                                          const __amdgpu_buffer_rsrc_t &rw, unsigned vow, unsigned gofs_w, BF &&bfrag)
{
    static_assert(!STAGE || NSTG_W <= KCn * PER, "every staged slice needs a K chunk to ride on");
    i32x4_t st[3][PER];
    uint4 Af[2][MTn];
#pragma unroll
    for (int mt = 0; mt < MTn; ++mt) Af[0][mt] = lds_cur[mt * 64];
    const f32x16 zero = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int kc = 0; kc < KCn; ++kc) {
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (STAGE) {
#pragma unroll
            for (int p = 0; p < PER; ++p) {
                const int iw = (kc - 2) * PER + p;
                if (kc >= 2 && iw < NSTG_W) lds_nxt_w[iw * FB_WAVES * 64] = __builtin_bit_cast(uint4, st[(kc - 2) % 3][p]);
            }
#pragma unroll
            for (int p = 0; p < PER; ++p) {
                const int il = kc * PER + p;
                if (il < NSTG_W) st[kc % 3][p] = __builtin_amdgcn_raw_buffer_load_b128(rw, vow, gofs_w + (unsigned)(il * FB_WAVES) * 1024u, 0);
            }
        }
        if (kc + 1 < KCn) {
#pragma unroll
            for (int mt = 0; mt < MTn; ++mt) Af[(kc + 1) & 1][mt] = lds_cur[((kc + 1) * MTn + mt) * 64];
        }
        __builtin_amdgcn_sched_barrier(0);
        bf16x8 Ba, Bb;
        if (kc == 0) bfrag(IC<0>{}, Ba, Bb);
        else if (kc == 1) bfrag(IC<(KCn > 1 ? 1 : 0)>{}, Ba, Bb);
        else if (kc == 2) bfrag(IC<(KCn > 2 ? 2 : 0)>{}, Ba, Bb);
        else if (kc == 3) bfrag(IC<(KCn > 3 ? 3 : 0)>{}, Ba, Bb);
        else if (kc == 4) bfrag(IC<(KCn > 4 ? 4 : 0)>{}, Ba, Bb);
        else if (kc == 5) bfrag(IC<(KCn > 5 ? 5 : 0)>{}, Ba, Bb);
        else if (kc == 6) bfrag(IC<(KCn > 6 ? 6 : 0)>{}, Ba, Bb);
        else if (kc == 7) bfrag(IC<(KCn > 7 ? 7 : 0)>{}, Ba, Bb);
        else if (kc == 8) bfrag(IC<(KCn > 8 ? 8 : 0)>{}, Ba, Bb);
        else if (kc == 9) bfrag(IC<(KCn > 9 ? 9 : 0)>{}, Ba, Bb);
        else if (kc == 10) bfrag(IC<(KCn > 10 ? 10 : 0)>{}, Ba, Bb);
        else if (kc == 11) bfrag(IC<(KCn > 11 ? 11 : 0)>{}, Ba, Bb);
        else if (kc == 12) bfrag(IC<(KCn > 12 ? 12 : 0)>{}, Ba, Bb);
        else if (kc == 13) bfrag(IC<(KCn > 13 ? 13 : 0)>{}, Ba, Bb);
        else if (kc == 14) bfrag(IC<(KCn > 14 ? 14 : 0)>{}, Ba, Bb);
        else if (kc == 15) bfrag(IC<(KCn > 15 ? 15 : 0)>{}, Ba, Bb);
        else if (kc == 16) bfrag(IC<(KCn > 16 ? 16 : 0)>{}, Ba, Bb);
        else if (kc == 17) bfrag(IC<(KCn > 17 ? 17 : 0)>{}, Ba, Bb);
        else if (kc == 18) bfrag(IC<(KCn > 18 ? 18 : 0)>{}, Ba, Bb);
        else bfrag(IC<(KCn > 19 ? 19 : 0)>{}, Ba, Bb);
        static_assert(KCn <= 20, "extend the chunk dispatch");
#pragma unroll
        for (int mt = 0; mt < MTn; ++mt) {
            const bf16x8 A = __builtin_bit_cast(bf16x8, Af[kc & 1][mt]);
            if constexpr (SWAP) {
                acc[mt][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(Ba, A, kc == 0 ? zero : acc[mt][0], 0, 0, 0);
                acc[mt][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(Bb, A, kc == 0 ? zero : acc[mt][1], 0, 0, 0);
            } else {
                acc[mt][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A, Ba, kc == 0 ? zero : acc[mt][0], 0, 0, 0);
                acc[mt][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A, Bb, kc == 0 ? zero : acc[mt][1], 0, 0, 0);
            }
        }
    }
    __builtin_amdgcn_sched_barrier(0);
    if constexpr (STAGE) {
#pragma unroll
        for (int kc = KCn; kc < KCn + 2; ++kc) {
#pragma unroll
            for (int p = 0; p < PER; ++p) {
                const int iw = (kc - 2) * PER + p;
                if (kc >= 2 && iw >= 0 && iw < NSTG_W) lds_nxt_w[iw * FB_WAVES * 64] = __builtin_bit_cast(uint4, st[(kc - 2) % 3][p]);
            }
        }
    }
}

// accumulators of T output tiles -> activation in B-fragment form: chunk 2t+q <- registers 8q..8q+7 of tile t, after the
// producing layer's affine (BatchNorm + bias folded) and ReLU; P[chunk][column tile][word]
template <int T>
__device__ __forceinline__ void pack_tiles(const f32x16 (&acc)[T][2], const float2 *aff_tile0 /*&aff[LB + 32 * first tile + 4h]*/,
                                           unsigned (*P)[2][4] /*first chunk of the first tile*/)
{
#pragma unroll
    for (int t = 0; t < T; ++t) {
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const float4 *ap = reinterpret_cast<const float4 *>(aff_tile0 + 32 * t + 16 * q);
            const float4 c0 = ap[0], c1 = ap[1], c2 = ap[4], c3 = ap[5];     // channels +0,1 | +2,3 | +8,9 | +10,11 (scale, shift pairs)
            const float sc[8] = {c0.x, c0.z, c1.x, c1.z, c2.x, c2.z, c3.x, c3.z};
            const float sh[8] = {c0.y, c0.w, c1.y, c1.w, c2.y, c2.w, c3.y, c3.w};
            float va[8], vb[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                va[e] = __fmaf_rn(acc[t][0][8 * q + e], sc[e], sh[e]);
                vb[e] = __fmaf_rn(acc[t][1][8 * q + e], sc[e], sh[e]);
            }
#pragma unroll
            for (int p = 0; p < 4; ++p) {
                // ReLU on the packed pair: a bf16 with the sign bit set is a negative 16-bit integer, so max(., 0) as signed
                // halves clears exactly the negative values (-0 -> +0; a positive-sign NaN stays a NaN) -- one v_pk_max_i16
                // per two values where compare + select on the f32 values would take four instructions
                P[2 * t + q][0][p] = fb_relu_pk(fb_cvt_pk_bf16(va[2 * p], va[2 * p + 1]));
                P[2 * t + q][1][p] = fb_relu_pk(fb_cvt_pk_bf16(vb[2 * p], vb[2 * p + 1]));
            }
        }
    }
}

__device__ __forceinline__ void frag_of(const unsigned (&p)[2][4], bf16x8 &Ba, bf16x8 &Bb) {
    Ba = __builtin_bit_cast(bf16x8, make_uint4(p[0][0], p[0][1], p[0][2], p[0][3]));
    Bb = __builtin_bit_cast(bf16x8, make_uint4(p[1][0], p[1][1], p[1][2], p[1][3]));
}

// POOL = the per-node max-pool epilogue instead of the y stores (x must then be node-sorted: sonet_som_sort_group_f32):
// replaces index_max + masked gather (models/networks.py:180-185) where only the pooled VALUES are needed.  Every 256-point
// tile pre-reduces the (few) nodes it touches in LDS bins -- integer atomicMax on orderable keys, no float atomics,
// deterministic -- and stores them once; pooled_bf16_decode_kernel combines the tiles of each node.  Values are rounded to
// bf16 before they compete (rounding is monotone: this IS the maximum of the bf16 features the store variant writes).
constexpr int FB_SLOTS = 8;                                    // nodes of a tile pre-reduced in LDS (the rest: global atomics)
constexpr unsigned FB_INIT = 0x3B85FFFFu;                     // orderable(-1000.0f): the reference's initial running max

__device__ __forceinline__ unsigned fb_ord(unsigned bits) {    // total order; -0 == +0; NaN -> 0 (never wins)
    if (bits == 0x80000000u) bits = 0u;
    const unsigned o = bits ^ ((unsigned)((int)bits >> 31) | 0x80000000u);
    return (bits & 0x7FFFFFFFu) > 0x7F800000u ? 0u : o;
}
__device__ __forceinline__ float fb_round_bf16(float v) {      // the f32 value of bf16(v), round to nearest even
    return __uint_as_float(fb_cvt_pk_bf16(v, v) << 16);
}

template <bool POOL>
__global__ __launch_bounds__(FB_THREADS, 1) void pointresnet_bf16_kernel(
    const float *__restrict__ x, int Cin0, const uint4 *__restrict__ Wst, const float2 *__restrict__ affine_g /*[832] (scale, shift)*/,
    uint16_t *__restrict__ y, int L, int tpc /*256-point tiles per cloud*/, long long ntiles, int abl /*bench-only ablations: 1 = no stores*/,
    const int32_t *__restrict__ ids_sorted, const int32_t *__restrict__ pos0, const int32_t *__restrict__ node_off, const int32_t *__restrict__ count,
    unsigned *__restrict__ pooled /*[B][M][384] keys, FB_INIT*/, unsigned *__restrict__ partial /*[ntiles][FB_SLOTS][384]*/, float *__restrict__ v0 /*[B][384]*/, int M)
{
    __shared__ uint4 wbuf[2][BUF_SL][64];                       // 2 x 60 KiB
    __shared__ __attribute__((aligned(16))) float2 aff[CH_TOTAL_BF];
    __shared__ unsigned bins[POOL ? FB_SLOTS : 1][POOL ? 384 : 1];
    if constexpr (POOL) {
        for (int i = threadIdx.x; i < FB_SLOTS * 384; i += FB_THREADS) (&bins[0][0])[i] = FB_INIT;
    }

    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int j = lane & 31, h = lane >> 5;
    const unsigned vow = (unsigned)lane * 16u;
    const unsigned rowX = (unsigned)L * 4u, rowY = (unsigned)L * 2u;
    const __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint4 *>(Wst), 0, NSLICE_BF * 1024, 0x00020000);
    for (int c = threadIdx.x; c < CH_TOTAL_BF; c += FB_THREADS) aff[c] = affine_g[c];
    for (int sl = wave; sl < G0_SL; sl += FB_WAVES)             // group 0 of the first tile
        wbuf[0][sl][lane] = __builtin_bit_cast(uint4, __builtin_amdgcn_raw_buffer_load_b128(rw, vow, (unsigned)sl * 1024u, 0));
    __syncthreads();
    const bool paired = (L % 2 == 0) && ((reinterpret_cast<uintptr_t>(y) & 3) == 0);
    int nb = 0;                                                 // running group counter: this group's buffer = nb & 1

    // the wave's 64 columns of tile t: two f32 values per lane and input row (points 2j, 2j+1)
    float xin[8][2];
    int nid_a = -1, nid_b = -1, n0t_n = 0, nlt_n = 0, p0_n = 0;  // POOL: node ids of the lane's two points, first / last node of the tile, pos0
    auto load_x = [&](long long t) {
        const long long b = t / tpc;
        const int l0 = (int)(t - b * tpc) * 256 + wave * 64;
        const int ca = l0 + 2 * j, cb = ca + 1;
        const int cca = ca < L ? ca : (l0 < L ? l0 : 0), ccb = cb < L ? cb : cca;
        const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc(
            const_cast<float *>(x + b * (long long)Cin0 * L), 0, (int)((unsigned)Cin0 * rowX), 0x00020000);
#pragma unroll
        for (int e = 0; e < 8; ++e) {                           // rows 8h + e; rows >= Cin0 read zeros (descriptor bounds)
            xin[e][0] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rx, (unsigned)(8 * h * L + cca) * 4u, (unsigned)e * rowX, 0));
            xin[e][1] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rx, (unsigned)(8 * h * L + ccb) * 4u, (unsigned)e * rowX, 0));
        }
        if constexpr (POOL) {                                   // everything the pool epilogue needs, requested a tile ahead: no
            const int32_t *idb = ids_sorted + b * (long long)L; // dependent global load sits between the MFMA passes
            nid_a = ca < L ? idb[ca] : -1;
            nid_b = cb < L ? idb[cb] : -1;
            const int t0 = l0 - wave * 64;
            n0t_n = idb[t0 < L ? t0 : 0];
            nlt_n = idb[t0 + 255 < L ? t0 + 255 : L - 1];
            p0_n = pos0[b];
        }
    };
    if ((long long)blockIdx.x < ntiles) load_x(blockIdx.x);

    for (long long tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const long long b = tile / tpc;
        const int l0 = (int)(tile - b * tpc) * 256 + wave * 64;
        const int ca = l0 + 2 * j, cb = ca + 1;
        const bool pva = ca < L, pvb = cb < L;
        const int cca = pva ? ca : (l0 < L ? l0 : 0), ccb = pvb ? cb : cca;
        const bool has_next = tile + gridDim.x < ntiles;

        const int my_a = nid_a, my_b = nid_b, n0t = __builtin_amdgcn_readfirstlane(n0t_n), nlt = __builtin_amdgcn_readfirstlane(nlt_n),
                  p0t = __builtin_amdgcn_readfirstlane(p0_n);
        unsigned P1[4][2][4], P2[8][2][4], P3[16][2][4];
        // ---- group 0: layer 1 (one chunk from memory) and layer 2, staging layer 3's first group ------------------------
        {
            unsigned X0[2][4];
#pragma unroll
            for (int p = 0; p < 4; ++p) {
                X0[0][p] = fb_cvt_pk_bf16(xin[2 * p][0], xin[2 * p + 1][0]);
                X0[1][p] = fb_cvt_pk_bf16(xin[2 * p][1], xin[2 * p + 1][1]);
            }
            const uint4 *cur = &wbuf[nb & 1][0][lane];
            uint4 *nxt = &wbuf[(nb & 1) ^ 1][wave][lane];
            f32x16 a1[2][2];
            mfma_pass<1, 2, 1, 0, false>(a1, cur + OFF_L1 * 64, nxt, rw, vow, 0u, [&](auto, bf16x8 &Ba, bf16x8 &Bb) { frag_of(X0, Ba, Bb); });
            pack_tiles<2>(a1, aff + AFF_L1 + 4 * h, P1);
            f32x16 a2[4][2];
            mfma_pass<4, 4, 2, L3G_SL / FB_WAVES, true>(a2, cur + OFF_L2 * 64, nxt, rw, vow, (unsigned)(OFF_L3 + wave) * 1024u,
                                                        [&](auto kc, bf16x8 &Ba, bf16x8 &Bb) { frag_of(P1[decltype(kc)::value], Ba, Bb); });
            pack_tiles<4>(a2, aff + AFF_L2 + 4 * h, P2);
            __syncthreads();
            ++nb;
        }
        // ---- groups 1, 2: layer 3, tiles 0-3 and 4-7 ---------------------------------------------------------------------
        {
            const uint4 *cur = &wbuf[nb & 1][0][lane];
            uint4 *nxt = &wbuf[(nb & 1) ^ 1][wave][lane];
            f32x16 a3[4][2];
            mfma_pass<8, 4, 1, L3G_SL / FB_WAVES, true>(a3, cur, nxt, rw, vow, (unsigned)(OFF_L3 + L3G_SL + wave) * 1024u,
                                                        [&](auto kc, bf16x8 &Ba, bf16x8 &Bb) { frag_of(P2[decltype(kc)::value], Ba, Bb); });
            pack_tiles<4>(a3, aff + AFF_L3 + 4 * h, P3);
            __syncthreads();
            ++nb;
        }
        {
            const uint4 *cur = &wbuf[nb & 1][0][lane];
            uint4 *nxt = &wbuf[(nb & 1) ^ 1][wave][lane];
            f32x16 a3[4][2];
            mfma_pass<8, 4, 2, L4G_SL / FB_WAVES, true>(a3, cur, nxt, rw, vow, (unsigned)(OFF_L4 + wave) * 1024u,
                                                        [&](auto kc, bf16x8 &Ba, bf16x8 &Bb) { frag_of(P2[decltype(kc)::value], Ba, Bb); });
            pack_tiles<4>(a3, aff + AFF_L3 + 32 * 4 + 4 * h, P3 + 8);
            __syncthreads();
            ++nb;
        }
        // the next tile's input: requested here, consumed after four layer-4 groups
        if (has_next) load_x(tile + gridDim.x);
        // ---- groups 3..6: layer 4, three output tiles each; K chunks 0-3 = act1 (the skip), 4-19 = act3 ------------------
        const __amdgpu_buffer_rsrc_t ry = __builtin_amdgcn_make_buffer_rsrc(y + b * (long long)384 * L, 0, (int)(384u * rowY), 0x00020000);
        const unsigned voya = (unsigned)(4 * h * L + cca) * 2u, voyb = (unsigned)(4 * h * L + ccb) * 2u;
        for (int g = 0; g < 4; ++g) {
            const uint4 *cur = &wbuf[nb & 1][0][lane];
            uint4 *nxt = &wbuf[(nb & 1) ^ 1][wave][lane];
            f32x16 a4[3][2];
            auto b4 = [&](auto kc, bf16x8 &Ba, bf16x8 &Bb) {
                constexpr int k = decltype(kc)::value;
                if constexpr (k < 4) frag_of(P1[k], Ba, Bb); else frag_of(P3[k - 4], Ba, Bb);
            };
            if (g < 3) mfma_pass<20, 3, 1, L4G_SL / FB_WAVES, true, POOL>(a4, cur, nxt, rw, vow, (unsigned)(OFF_L4 + (g + 1) * L4G_SL + wave) * 1024u, b4);
            else if (has_next) mfma_pass<20, 3, 1, G0_SL / FB_WAVES, true, POOL>(a4, cur, nxt, rw, vow, (unsigned)wave * 1024u, b4);
            else mfma_pass<20, 3, 1, 0, false, POOL>(a4, cur, nxt, rw, vow, 0u, b4);
            if constexpr (POOL) {
                // a4[mt][ct][r] = Y[point 2 * prow + ct][channel 96 g + 32 mt + j], prow = (r&3) + 8 (r>>2) + 4 h.  The wave's 64
                // points are consecutive in node-sorted order, so the points of a node are a range [s, e) of them.
                const int lw = l0;                                          // first point of the wave (wave-uniform)
                float bias4[3];
                bool unit = true;
#pragma unroll
                for (int mt = 0; mt < 3; ++mt) {
                    const float2 ss = aff[AFF_L4 + (g * 3 + mt) * 32 + j];
                    bias4[mt] = ss.y;
                    unit = unit && (ss.x == 1.0f);
                }
                if (__builtin_amdgcn_ballot_w64(!unit) != 0ull) {           // (never in the reference: layer 4 has no BatchNorm)
#pragma unroll
                    for (int mt = 0; mt < 3; ++mt) {
                        const float sc = aff[AFF_L4 + (g * 3 + mt) * 32 + j].x;
#pragma unroll
                        for (int ct = 0; ct < 2; ++ct)
#pragma unroll
                            for (int r = 0; r < 16; ++r) a4[mt][ct][r] = __fmaf_rn(a4[mt][ct][r], sc, bias4[mt]);
                        bias4[mt] = 0.f;
                    }
                }
                if (lw < L && !(abl & 16)) {
                    const int nvalid = (L - lw) < 64 ? (L - lw) : 64;
                    // features of original point copy 0 (what the reference gathers for a node that never beat -1000)
                    const int p0 = p0t - lw;
                    if (p0 >= 0 && p0 < 64) {
                        const int row = p0 >> 1, ct0 = p0 & 1, hsel = (row >> 2) & 1, rsel = (row & 3) + 4 * (row >> 3);
#pragma unroll
                        for (int r = 0; r < 16; ++r)
                            if (r == rsel && h == hsel) {
#pragma unroll
                                for (int mt = 0; mt < 3; ++mt)
                                    v0[b * 384 + (g * 3 + mt) * 32 + j] = fb_round_bf16((ct0 ? a4[mt][1][r] : a4[mt][0][r]) + bias4[mt]);
                            }
                    }
                    // node segments from the ids the lanes hold (lanes 0..31 <-> point pairs; sorted, so a node is a range)
                    int s0 = 0;
                    while (s0 < nvalid) {
                        const int node = (s0 & 1) ? __builtin_amdgcn_readlane(my_b, s0 >> 1) : __builtin_amdgcn_readlane(my_a, s0 >> 1);
                        const unsigned ma = (unsigned)__builtin_amdgcn_ballot_w64(my_a == node), mb = (unsigned)__builtin_amdgcn_ballot_w64(my_b == node);
                        const int e0 = s0 + __builtin_popcount(ma) + __builtin_popcount(mb);
                        float mx[3];
                        if (abl & 2) {
#pragma unroll
                            for (int mt = 0; mt < 3; ++mt) mx[mt] = a4[mt][0][0] + a4[mt][1][7];
                        } else if (s0 == 0 && e0 == 64) {
#pragma unroll
                            for (int mt = 0; mt < 3; ++mt) {
                                float m = a4[mt][0][0];
#pragma unroll
                                for (int r = 1; r < 16; ++r) asm("v_max_f32 %0, %1, %2" : "=v"(m) : "v"(m), "v"(a4[mt][0][r]));
#pragma unroll
                                for (int r = 0; r < 16; ++r) asm("v_max_f32 %0, %1, %2" : "=v"(m) : "v"(m), "v"(a4[mt][1][r]));
                                mx[mt] = m;
                            }
                        } else {
#pragma unroll
                            for (int mt = 0; mt < 3; ++mt) mx[mt] = -__builtin_inff();
#pragma unroll
                            for (int r = 0; r < 16; ++r) {
                                const int pa = 2 * ((r & 3) + 8 * (r >> 2) + 4 * h);
                                const bool ina = pa >= s0 && pa < e0, inb = pa + 1 >= s0 && pa + 1 < e0;
#pragma unroll
                                for (int mt = 0; mt < 3; ++mt) {
                                    const float va = ina ? a4[mt][0][r] : -__builtin_inff(), vb = inb ? a4[mt][1][r] : -__builtin_inff();
                                    asm("v_max_f32 %0, %1, %2" : "=v"(mx[mt]) : "v"(mx[mt]), "v"(va));      // (v_max ignores a NaN operand, as the reference's '>' does)
                                    asm("v_max_f32 %0, %1, %2" : "=v"(mx[mt]) : "v"(mx[mt]), "v"(vb));
                                }
                            }
                        }
                        const int slot = node - n0t;
                        s0 = e0;                                                                            // the next node starts here
#pragma unroll
                        for (int mt = 0; mt < 3; ++mt) {
                            float m = mx[mt];
                            const float o = __shfl_xor(m, 32, 64);                                          // the other half-wave's 16 rows
                            asm("v_max_f32 %0, %1, %2" : "=v"(m) : "v"(m), "v"(o));
                            const unsigned key = fb_ord(__float_as_uint(fb_round_bf16(m + bias4[mt])));
                            if (h == 0 && !(abl & 4)) {
                                if (slot < FB_SLOTS) atomicMax(&bins[slot][(g * 3 + mt) * 32 + j], key);
                                else atomicMax(pooled + ((long long)b * M + node) * 384 + (g * 3 + mt) * 32 + j, key);
                            }
                        }
                    }
                }
                __syncthreads();                                            // every wave's maxima of this group are in; W buffers swap
                ++nb;
                if (g == 3 && !(abl & 8)) {                                 // tile done: one store of the slots it used, bins back to INIT
                    int ns = nlt - n0t + 1;
                    ns = ns < FB_SLOTS ? ns : FB_SLOTS;
                    unsigned *dst = partial + tile * (long long)(FB_SLOTS * 384);
                    for (int i = threadIdx.x; i < ns * 384; i += FB_THREADS) {
                        dst[i] = (&bins[0][0])[i];
                        (&bins[0][0])[i] = FB_INIT;
                    }
                }
                continue;
            }
            if (pva && !(abl & 1)) {
                auto store_tiles = [&](auto paired_c) {
#pragma unroll
                    for (int mt = 0; mt < 3; ++mt) {
                        __builtin_amdgcn_sched_barrier(0);
                        const int ct = g * 3 + mt;
                        const unsigned so_tile = (unsigned)(ct * 32) * rowY;
                        const float2 *ap = aff + AFF_L4 + ct * 32 + 4 * h;
#pragma unroll
                        for (int r = 0; r < 16; ++r) {
                            const int orow = (r & 3) + 8 * (r >> 2);
                            const float2 ss = ap[orow];
                            const unsigned pk = fb_cvt_pk_bf16(__fmaf_rn(a4[mt][0][r], ss.x, ss.y), __fmaf_rn(a4[mt][1][r], ss.x, ss.y));
                            const unsigned so = so_tile + (unsigned)orow * rowY;
                            if constexpr (decltype(paired_c)::value != 0) {
                                __builtin_amdgcn_raw_buffer_store_b32((int)pk, ry, voya, so, 0);
                            } else {
                                __builtin_amdgcn_raw_buffer_store_b16((short)(pk & 0xFFFFu), ry, voya, so, 0);
                                if (pvb) __builtin_amdgcn_raw_buffer_store_b16((short)(pk >> 16), ry, voyb, so, 0);
                            }
                        }
                    }
                };
                if (paired) store_tiles(IC<1>{}); else store_tiles(IC<0>{});
            }
            if (g < 3 || has_next) {
                __syncthreads();
                ++nb;
            }
        }
    }
}

// ======================================================================================================================
// Pool variant, second generation: TWO independent workgroups per CU, 4 waves x 32 points each (two waves per SIMD).
// The 4-wave / 64-point kernel above leaves the matrix pipe idle whenever its one wave per SIMD does vector work between the
// MFMA passes (activation packing: ~450 values per lane and tile; the per-node register max of the pool epilogue): 0.30 ms of
// MFMA passes become 0.46-0.49 ms.  Tried first: (a) a second accumulator set with the previous group's epilogue sliced into
// the MFMA shadow -- 137 register spills (the 256 architectural VGPRs hold the packed activations); (b) ONE workgroup of 8
// waves x 32 points -- bit-identical, 233 registers, and no faster (0.483 ms): the per-group barrier keeps all eight waves in
// the same phase, so they fight for the matrix pipe together and do their vector work together.  Two workgroups that share
// nothing drift apart and fill each other's gaps.  To fit two sets of W buffers into the LDS a group is at most 32 slices:
// layer 4's 20 K chunks x 3 tiles go by in two halves of 10 chunks (the accumulators carry over), 64 KiB of W per workgroup.
// Price: every A fragment read from LDS feeds one MFMA instead of two, and the weight stream is pulled once per 128 points
// instead of once per 256.  Same stream, arithmetic and roundings as the kernels above: bit-identical results.
template <int N, typename F, int... I>
__device__ __forceinline__ void static_for_impl(F &&f, std::integer_sequence<int, I...>) { (f(IC<I>{}), ...); }
template <int N, typename F>
__device__ __forceinline__ void static_for(F &&f) { static_for_impl<N>(f, std::make_integer_sequence<int, N>{}); }

constexpr int P2_THREADS = 256, P2_WAVES = 4;
constexpr int P2_BUF = 32;                                      // slices per LDS buffer
constexpr int P2_SLOTS = 4;                                     // nodes of a 128-point tile pre-reduced in LDS
constexpr int L4H_SL = 30;                                      // half a layer-4 group: 10 chunks x 3 tiles

// KC0: first K chunk of this pass (acc is cleared when KC0 == 0), KCn chunks, MTn tiles, ONE 32-column tile.  Staging: this
// wave moves the NSTG_W consecutive slices wave * NSTG_W ... of the next group, PER per K chunk.
// The staged slices go global -> LDS by LDS-DMA (global_load_lds_dwordx4: one 1-KiB slice per wave instruction, lane-linear,
// which is the slice layout): no staging registers, no ds_write, and -- the point -- no wait in the middle of the pass: with
// register staging the ds_write two chunks after its load stalled the wave for an L2 round trip eight times per pass (SQ
// counters of the register-staged version: half of the wave cycles waiting).  hipcc does not see these loads; the caller waits
// for them (s_waitcnt vmcnt(0)) right before the barrier that publishes the group.
// Addressing: slices il and il + 1 ... of one wave are 1 KiB apart in the stream AND in LDS, and the instruction offset moves
// both addresses, so four slices share one (M0, voffset) pair and every slice of a pass shares the scalar base `gsrc`: three
// scalar registers and two vector registers per pass.  (The first version gave every slice its own 64-bit pointer and LDS
// address; hipcc hoisted all 45 of them out of the tile loop into spilled SGPRs, and that build returned garbage whatever
// waits and nops surrounded the DMA -- builds differing only by a debugging printf were right.  Not understood; this form has
// no spilled operands.)
template <int KC0, int KCn, int MTn, int PER, int NSTG_W, bool STAGE, bool SWAP, typename BF>
__device__ __forceinline__ void mfma_pass1(f32x16 (&acc)[MTn], const uint4 *lds_cur, unsigned lds_nxt_addr /*LDS byte address of the next buffer*/,
                                           const char *gsrc /*slice 0 of the next group in the weight stream*/, unsigned vow, int wave, BF &&bfrag)
{
    static_assert(!STAGE || NSTG_W <= KCn * PER, "every staged slice needs a K chunk to ride on");
    uint4 Af[2][MTn];
#pragma unroll
    for (int mt = 0; mt < MTn; ++mt) Af[0][mt] = lds_cur[mt * 64];
    const f32x16 zero = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    const unsigned wofs = (unsigned)wave * (unsigned)(NSTG_W * 1024);
    const unsigned d_lo = lds_nxt_addr + wofs, d_hi = d_lo + 4096u;
    const unsigned v_lo = vow + wofs, v_hi = v_lo + 4096u;
    static_for<KCn>([&](auto kc_c) {
        constexpr int kc = decltype(kc_c)::value;
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (STAGE) {
            static_for<PER>([&](auto p_c) {
                constexpr int il = kc * PER + decltype(p_c)::value;
                if constexpr (il < NSTG_W) {
                    // (locals: clang refuses captured variables as asm operands.)  M0 is the compiler's: saved and restored.
                    // s_nop 4: should an operand arrive in an SGPR written by a VALU instruction (v_readlane of a spill), a
                    // VMEM instruction reading it needs 5 wait states, and hipcc pads its own instructions, not inline asm.
                    const char *g = gsrc;
                    const unsigned d = il < 4 ? d_lo : d_hi, vo = il < 4 ? v_lo : v_hi;
                    unsigned keep;
                    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 4\n\tglobal_load_lds_dwordx4 %1, %2 offset:%4\n\ts_mov_b32 m0, %0"
                                 : "=&s"(keep) : "v"(vo), "s"(g), "s"(d), "n"((il % 4) * 1024) : "memory");
                }
            });
        }
        if constexpr (kc + 1 < KCn) {
#pragma unroll
            for (int mt = 0; mt < MTn; ++mt) Af[(kc + 1) & 1][mt] = lds_cur[((kc + 1) * MTn + mt) * 64];
        }
        __builtin_amdgcn_sched_barrier(0);
        bf16x8 Bx;
        bfrag(IC<KC0 + kc>{}, Bx);
#pragma unroll
        for (int mt = 0; mt < MTn; ++mt) {
            const bf16x8 A = __builtin_bit_cast(bf16x8, Af[kc & 1][mt]);
            if constexpr (SWAP) acc[mt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(Bx, A, (KC0 + kc == 0) ? zero : acc[mt], 0, 0, 0);
            else acc[mt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A, Bx, (KC0 + kc == 0) ? zero : acc[mt], 0, 0, 0);
        }
    });
    __builtin_amdgcn_sched_barrier(0);
}

template <int T>
__device__ __forceinline__ void pack_tiles1(const f32x16 (&acc)[T], const float2 *aff_tile0, unsigned (*P)[4])
{
#pragma unroll
    for (int t = 0; t < T; ++t) {
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const float4 *ap = reinterpret_cast<const float4 *>(aff_tile0 + 32 * t + 16 * q);
            const float4 c0 = ap[0], c1 = ap[1], c2 = ap[4], c3 = ap[5];
            const float sc[8] = {c0.x, c0.z, c1.x, c1.z, c2.x, c2.z, c3.x, c3.z};
            const float sh[8] = {c0.y, c0.w, c1.y, c1.w, c2.y, c2.w, c3.y, c3.w};
            float va[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) va[e] = __fmaf_rn(acc[t][8 * q + e], sc[e], sh[e]);
#pragma unroll
            for (int p = 0; p < 4; ++p) P[2 * t + q][p] = fb_relu_pk(fb_cvt_pk_bf16(va[2 * p], va[2 * p + 1]));
        }
    }
}

__device__ __forceinline__ bf16x8 frag1(const unsigned (&p)[4]) { return __builtin_bit_cast(bf16x8, make_uint4(p[0], p[1], p[2], p[3])); }

__global__ __launch_bounds__(P2_THREADS, 2) void pointresnet_bf16_pool2_kernel(
    const float *__restrict__ x, int Cin0, const uint4 *__restrict__ Wst, const float2 *__restrict__ affine_g, int L, int tpc /*128-point tiles per cloud*/,
    long long ntiles, const int32_t *__restrict__ ids_sorted, const int32_t *__restrict__ pos0,
    unsigned *__restrict__ pooled, unsigned *__restrict__ partial /*[ntiles][P2_SLOTS][384]*/, float *__restrict__ v0, int M)
{
    // One LDS object so that the W buffers sit at its start: the LDS-DMA destination travels in M0, and the slices must stay
    // below 64 KiB of the workgroup's LDS (with separate __shared__ arrays the compiler chose the order, and W buffer 1 ended
    // past 64 KiB: its last slices landed on the affine table).
    struct P2Lds { uint4 wbuf[2][P2_BUF][64]; float2 aff[CH_TOTAL_BF]; unsigned bins[P2_SLOTS][384]; };
    __shared__ __attribute__((aligned(16))) P2Lds lds_;
    auto &wbuf = lds_.wbuf;
    auto &aff = lds_.aff;
    auto &bins = lds_.bins;

    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int j = lane & 31, h = lane >> 5;
    const unsigned vow = (unsigned)lane * 16u;
    const unsigned rowX = (unsigned)L * 4u;
    const __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint4 *>(Wst), 0, NSLICE_BF * 1024, 0x00020000);
    const char *wst = reinterpret_cast<const char *>(Wst);
    const unsigned wbuf_lds = (unsigned)reinterpret_cast<size_t>(&wbuf[0][0][0]);      // LDS byte address of the W buffers (LDS-DMA target)
    bool unit_lane = true;
    for (int c = threadIdx.x; c < CH_TOTAL_BF; c += P2_THREADS) {
        const float2 v = affine_g[c];
        aff[c] = v;
        if (c >= AFF_L4 && v.x != 1.0f) unit_lane = false;
    }
    for (int i = threadIdx.x; i < P2_SLOTS * 384; i += P2_THREADS) (&bins[0][0])[i] = FB_INIT;
    for (int sl = wave; sl < G0_SL; sl += P2_WAVES)
        wbuf[0][sl][lane] = __builtin_bit_cast(uint4, __builtin_amdgcn_raw_buffer_load_b128(rw, vow, (unsigned)sl * 1024u, 0));
    const bool l4_unit = __syncthreads_and(unit_lane) != 0;      // (also publishes aff, bins, group 0)
    int nb = 0;

    float xin[8];
    int nid_n = -1, n0t_n = 0, nlt_n = 0, p0_n = 0;
    auto load_x = [&](long long t) {
        const long long b = t / tpc;
        const int t0 = (int)(t - b * tpc) * 128, l0 = t0 + wave * 32;
        const int ca = l0 + j;
        const int cc = ca < L ? ca : (l0 < L ? l0 : 0);
        const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc(
            const_cast<float *>(x + b * (long long)Cin0 * L), 0, (int)((unsigned)Cin0 * rowX), 0x00020000);
#pragma unroll
        for (int e = 0; e < 8; ++e)
            xin[e] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rx, (unsigned)(8 * h * L + cc) * 4u, (unsigned)e * rowX, 0));
        const int32_t *idb = ids_sorted + b * (long long)L;
        nid_n = ca < L ? idb[ca] : -1;
        n0t_n = idb[t0 < L ? t0 : 0];
        nlt_n = idb[t0 + 127 < L ? t0 + 127 : L - 1];
        p0_n = pos0[b];
    };
    if ((long long)blockIdx.x < ntiles) load_x(blockIdx.x);

    for (long long tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const long long b = tile / tpc;
        const int l0 = (int)(tile - b * tpc) * 128 + wave * 32;
        const bool has_next = tile + gridDim.x < ntiles;
        const int my_id = nid_n, n0t = __builtin_amdgcn_readfirstlane(n0t_n), nlt = __builtin_amdgcn_readfirstlane(nlt_n),
                  p0t = __builtin_amdgcn_readfirstlane(p0_n);
        unsigned P1[4][4], P2[8][4], P3[16][4];
        {   // group 0: layer 1 + layer 2, staging layer 3's first group (32 slices: 8 consecutive per wave over the 4 chunks of layer 2)
            unsigned X0[4];
#pragma unroll
            for (int p = 0; p < 4; ++p) X0[p] = fb_cvt_pk_bf16(xin[2 * p], xin[2 * p + 1]);
            const uint4 *cur = &wbuf[nb & 1][0][lane];
            const unsigned nxt = wbuf_lds + (unsigned)(((nb & 1) ^ 1) * P2_BUF) * 1024u;
            f32x16 a1[2];
            mfma_pass1<0, 1, 2, 1, 0, false, false>(a1, cur + OFF_L1 * 64, nxt, wst, vow, wave, [&](auto, bf16x8 &B) { B = frag1(X0); });
            pack_tiles1<2>(a1, aff + AFF_L1 + 4 * h, P1);
            f32x16 a2[4];
            mfma_pass1<0, 4, 4, 2, 8, true, false>(a2, cur + OFF_L2 * 64, nxt, wst + OFF_L3 * 1024, vow, wave,
                                                           [&](auto kc, bf16x8 &B) { B = frag1(P1[decltype(kc)::value]); });
            pack_tiles1<4>(a2, aff + AFF_L2 + 4 * h, P2);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // this wave's LDS-DMA slices of the next group have landed
            __syncthreads();
            ++nb;
        }
        {   // group 1: layer 3 tiles 0-3, staging tiles 4-7
            const uint4 *cur = &wbuf[nb & 1][0][lane];
            const unsigned nxt = wbuf_lds + (unsigned)(((nb & 1) ^ 1) * P2_BUF) * 1024u;
            f32x16 a3[4];
            mfma_pass1<0, 8, 4, 1, 8, true, false>(a3, cur, nxt, wst + (OFF_L3 + L3G_SL) * 1024, vow, wave,
                                                           [&](auto kc, bf16x8 &B) { B = frag1(P2[decltype(kc)::value]); });
            pack_tiles1<4>(a3, aff + AFF_L3 + 4 * h, P3);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // this wave's LDS-DMA slices of the next group have landed
            __syncthreads();
            ++nb;
        }
        {   // group 2: layer 3 tiles 4-7, staging the first half of layer 4's first group (30 slices; 32 are moved, the last two belong to the next half)
            const uint4 *cur = &wbuf[nb & 1][0][lane];
            const unsigned nxt = wbuf_lds + (unsigned)(((nb & 1) ^ 1) * P2_BUF) * 1024u;
            f32x16 a3[4];
            mfma_pass1<0, 8, 4, 1, 8, true, false>(a3, cur, nxt, wst + OFF_L4 * 1024, vow, wave,
                                                           [&](auto kc, bf16x8 &B) { B = frag1(P2[decltype(kc)::value]); });
            pack_tiles1<4>(a3, aff + AFF_L3 + 32 * 4 + 4 * h, P3 + 8);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // this wave's LDS-DMA slices of the next group have landed
            __syncthreads();
            ++nb;
        }
        if (has_next) load_x(tile + gridDim.x);
        const int nvalid = l0 < L ? ((L - l0) < 32 ? (L - l0) : 32) : 0;
        for (int g = 0; g < 4; ++g) {
            f32x16 a4[3];
            auto b4 = [&](auto kc, bf16x8 &B) {
                constexpr int k = decltype(kc)::value;
                if constexpr (k < 4) B = frag1(P1[k]); else B = frag1(P3[k - 4]);
            };
            {   // K chunks 0-9 (act1 + the first 6 chunks of act3), staging chunks 10-19 of the same three tiles
                const uint4 *cur = &wbuf[nb & 1][0][lane];
                const unsigned nxt = wbuf_lds + (unsigned)(((nb & 1) ^ 1) * P2_BUF) * 1024u;
                mfma_pass1<0, 10, 3, 1, 8, true, true>(a4, cur, nxt, wst + (OFF_L4 + g * L4G_SL + L4H_SL) * 1024, vow, wave, b4);
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __syncthreads();
                ++nb;
            }
            {   // K chunks 10-19, staging the next group's first half (or the next tile's group 0)
                const uint4 *cur = &wbuf[nb & 1][0][lane];
                const unsigned nxt = wbuf_lds + (unsigned)(((nb & 1) ^ 1) * P2_BUF) * 1024u;
                if (g < 3) mfma_pass1<10, 10, 3, 1, 8, true, true>(a4, cur, nxt, wst + (OFF_L4 + (g + 1) * L4G_SL) * 1024, vow, wave, b4);
                else if (has_next) mfma_pass1<10, 10, 3, 1, 5, true, true>(a4, cur, nxt, wst, vow, wave, b4);
                else mfma_pass1<10, 10, 3, 1, 0, false, true>(a4, cur, nxt, wst, vow, wave, b4);
            }
            // a4[mt][r] = Y[point (r&3) + 8 (r>>2) + 4 h][channel 96 g + 32 mt + j]
            float bias4[3];
#pragma unroll
            for (int mt = 0; mt < 3; ++mt) {
                const float2 ss = aff[AFF_L4 + (g * 3 + mt) * 32 + j];
                bias4[mt] = ss.y;
                if (!l4_unit) {
#pragma unroll
                    for (int r = 0; r < 16; ++r) a4[mt][r] = __fmaf_rn(a4[mt][r], ss.x, ss.y);
                    bias4[mt] = 0.f;
                }
            }
            if (nvalid > 0) {
                const int p0 = p0t - l0;
                if (p0 >= 0 && p0 < 32) {                                   // features of original point copy 0
                    const int hsel = (p0 >> 2) & 1, rsel = (p0 & 3) + 4 * (p0 >> 3);
#pragma unroll
                    for (int r = 0; r < 16; ++r)
                        if (r == rsel && h == hsel) {
#pragma unroll
                            for (int mt = 0; mt < 3; ++mt) v0[b * 384 + (g * 3 + mt) * 32 + j] = fb_round_bf16(a4[mt][r] + bias4[mt]);
                        }
                }
                int s0 = 0;
                while (s0 < nvalid) {
                    const int node = __builtin_amdgcn_readlane(my_id, s0);
                    const int e0 = s0 + __builtin_popcount((unsigned)__builtin_amdgcn_ballot_w64(my_id == node));
                    float mx[3];
                    if (s0 == 0 && e0 == 32) {
#pragma unroll
                        for (int mt = 0; mt < 3; ++mt) {
                            float m = a4[mt][0];
#pragma unroll
                            for (int r = 1; r < 16; ++r) asm("v_max_f32 %0, %1, %2" : "=v"(m) : "v"(m), "v"(a4[mt][r]));
                            mx[mt] = m;
                        }
                    } else {
#pragma unroll
                        for (int mt = 0; mt < 3; ++mt) mx[mt] = -__builtin_inff();
#pragma unroll
                        for (int r = 0; r < 16; ++r) {
                            const int pr = (r & 3) + 8 * (r >> 2) + 4 * h;
                            const bool in = pr >= s0 && pr < e0;
#pragma unroll
                            for (int mt = 0; mt < 3; ++mt) {
                                const float v = in ? a4[mt][r] : -__builtin_inff();
                                asm("v_max_f32 %0, %1, %2" : "=v"(mx[mt]) : "v"(mx[mt]), "v"(v));      // (ignores a NaN, as the reference's '>')
                            }
                        }
                    }
                    const int slot = node - n0t;
#pragma unroll
                    for (int mt = 0; mt < 3; ++mt) {
                        float m = mx[mt];
                        const float o = __shfl_xor(m, 32, 64);
                        asm("v_max_f32 %0, %1, %2" : "=v"(m) : "v"(m), "v"(o));
                        const unsigned key = fb_ord(__float_as_uint(fb_round_bf16(m + bias4[mt])));
                        if (h == 0) {
                            if (slot < P2_SLOTS) atomicMax(&bins[slot][(g * 3 + mt) * 32 + j], key);
                            else atomicMax(pooled + ((long long)b * M + node) * 384 + (g * 3 + mt) * 32 + j, key);
                        }
                    }
                    s0 = e0;
                }
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // this wave's LDS-DMA slices of the next group have landed
            __syncthreads();
            ++nb;
            if (g == 3) {
                int ns = nlt - n0t + 1;
                ns = ns < P2_SLOTS ? ns : P2_SLOTS;
                unsigned *dst = partial + tile * (long long)(P2_SLOTS * 384);
                for (int i = threadIdx.x; i < ns * 384; i += P2_THREADS) {
                    dst[i] = (&bins[0][0])[i];
                    (&bins[0][0])[i] = FB_INIT;
                }
            }
        }
    }
}

__global__ __launch_bounds__(256) void pooled_bf16_init_kernel(unsigned *__restrict__ pooled, long long n) {
    const long long t = (long long)blockIdx.x * 256 + threadIdx.x;
    if (t < n) pooled[t] = FB_INIT;
}

// out[b][c][m] = max over the tiles that hold copies of node m of that tile's slot (slot = m - first node of the tile), combined
// with the straight-to-memory fallback in `pooled` (tiles spanning more than FB_SLOTS nodes).  Nodes that never beat -1000
// (empty, or all values <= -1000) take the features of original point copy 0 (models/networks.py:185: gather index 0).
__global__ __launch_bounds__(256) void pooled_bf16_decode_kernel(const unsigned *__restrict__ pooled, const unsigned *__restrict__ partial,
                                                                  const int32_t *__restrict__ ids_sorted, const int32_t *__restrict__ node_off,
                                                                  const int32_t *__restrict__ count, const float *__restrict__ v0,
                                                                  float *__restrict__ out, int M, int L, int tpc, long long total,
                                                                  int tile_pts /*points per tile*/, int slots /*LDS-reduced nodes per tile*/)
{
    const long long t = (long long)blockIdx.x * 256 + threadIdx.x;      // over [B][M][384], c fastest (coalesced partial reads)
    if (t >= total) return;
    const int c = (int)(t % 384);
    const long long bm = t / 384;
    const int m = (int)(bm % M);
    const long long b = bm / M;
    unsigned key = pooled[t];
    const int cnt = count[b * M + m];
    if (cnt > 0) {
        const int off = node_off[b * M + m];
        for (int tl = off / tile_pts; tl <= (off + cnt - 1) / tile_pts; ++tl) {
            const int slot = m - ids_sorted[b * L + tl * tile_pts];
            if (slot < slots) {
                const unsigned k2 = partial[((b * tpc + tl) * slots + slot) * 384ll + c];
                key = k2 > key ? k2 : key;
            }
        }
    }
    float v;
    if (key > FB_INIT) v = __uint_as_float((key & 0x80000000u) ? (key ^ 0x80000000u) : ~key);
    else v = v0[b * 384 + c];
    out[(b * 384 + c) * M + m] = v;
}

}  // namespace

extern "C" size_t sonet_pointresnet_bf16_pack_size(void) { return (size_t)(NSLICE_BF + NSLICE_PAD) * 1024; }

extern "C" int sonet_pointresnet_bf16_pack(const float *W1, const float *W2, const float *W3, const float *W4, int Cin0,
                                           void *stream_out, sonet_stream_t stream)
{
    const char *what = "sonet_pointresnet_bf16_pack";
    SONET_REQUIRE(W1 && W2 && W3 && W4 && stream_out, "%s: NULL pointer", what);
    SONET_REQUIRE(Cin0 >= 1 && Cin0 <= 16, "%s: Cin0=%d must be in [1, 16]", what, Cin0);
    hipLaunchKernelGGL(pointresnet_bf16_pack_kernel, dim3(sonet::ceil_div((NSLICE_BF + NSLICE_PAD) * 64, 256)), dim3(256), 0, sonet::as_stream(stream),
                       W1, W2, W3, W4, Cin0, reinterpret_cast<uint4 *>(stream_out));
    return sonet::launched(what);
}

extern "C" int sonet_pointresnet_bf16(const float *x, int Cin0, const void *wstream, const float *affine,
                                      uint16_t *y, int B, int L, sonet_stream_t stream)
{
    const char *what = "sonet_pointresnet_bf16";
    SONET_REQUIRE(x && wstream && affine && y, "%s: NULL pointer", what);
    SONET_REQUIRE(B > 0 && L > 0 && Cin0 >= 1 && Cin0 <= 16, "%s: bad size B=%d L=%d Cin0=%d", what, B, L, Cin0);
    if ((double)384 * L * 2.0 >= 4.0e9 || (double)Cin0 * L * 4.0 >= 2.0e9) return sonet::fail(SONET_ERR_UNSUPPORTED, "%s: a per-cloud panel is too large", what);
    const int tpc = sonet::ceil_div(L, 256);
    const long long ntiles = (long long)B * tpc;
    int dev = 0, cus = 256;
    if (hipGetDevice(&dev) == hipSuccess) {
        int v = 0;
        if (hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && v > 0) cus = v;
    }
    const long long grid = ntiles < cus ? ntiles : cus;          // persistent: one workgroup per CU
    int abl = 0;
    if (const char *e = sonet::knob("SONET_BF16_FUSED_ABLATE")) abl = atoi(e);      // bench-only (tools/bench_bf16.py)
    hipLaunchKernelGGL(pointresnet_bf16_kernel<false>, dim3((unsigned)grid), dim3(FB_THREADS), 0, sonet::as_stream(stream),
                       x, Cin0, reinterpret_cast<const uint4 *>(wstream), reinterpret_cast<const float2 *>(affine), y, L, tpc, ntiles, abl,
                       (const int32_t *)nullptr, (const int32_t *)nullptr, (const int32_t *)nullptr, (const int32_t *)nullptr,
                       (unsigned *)nullptr, (unsigned *)nullptr, (float *)nullptr, 0);
    return sonet::launched(what);
}

extern "C" size_t sonet_pointresnet_bf16_pool_ws_size(int B, int L, int M)
{
    if (B <= 0 || L <= 0 || M <= 0) return 0;
    const long long n256 = (long long)B * sonet::ceil_div(L, 256) * FB_SLOTS, n128 = (long long)B * sonet::ceil_div(L, 128) * P2_SLOTS;
    return (size_t)((long long)B * M * 384 + (n256 > n128 ? n256 : n128) * 384) * 4 + (size_t)B * 384 * 4;
}

extern "C" int sonet_pointresnet_bf16_pool(const float *x_sorted, int Cin0, const void *wstream, const float *affine,
                                           const int32_t *ids_sorted, const int32_t *pos0, const int32_t *node_off,
                                           const int32_t *count, void *ws, float *out, int B, int L, int M, sonet_stream_t stream)
{
    const char *what = "sonet_pointresnet_bf16_pool";
    SONET_REQUIRE(x_sorted && wstream && affine && ids_sorted && pos0 && node_off && count && ws && out, "%s: NULL pointer", what);
    SONET_REQUIRE(B > 0 && L > 0 && M > 0 && Cin0 >= 1 && Cin0 <= 16, "%s: bad size B=%d L=%d M=%d Cin0=%d", what, B, L, M, Cin0);
    if ((double)Cin0 * L * 4.0 >= 2.0e9) return sonet::fail(SONET_ERR_UNSUPPORTED, "%s: a per-cloud panel is too large", what);
    hipStream_t st = sonet::as_stream(stream);
    const long long npool = (long long)B * M * 384;
    const char *e8 = sonet::knob("SONET_BF16_POOL2");               // bench-only: 0 = the 4-wave / 64-point kernel (one workgroup per CU)
    const bool two = !(e8 && atoi(e8) == 0);
    const int tile_pts = two ? 128 : 256, slots = two ? P2_SLOTS : FB_SLOTS;
    const int tpc = sonet::ceil_div(L, tile_pts);
    const long long ntiles = (long long)B * tpc;
    unsigned *pooled_ws = reinterpret_cast<unsigned *>(ws);
    unsigned *partial_ws = pooled_ws + npool;
    const long long n256 = (long long)B * sonet::ceil_div(L, 256) * FB_SLOTS, n128 = (long long)B * sonet::ceil_div(L, 128) * P2_SLOTS;
    float *v0_ws = reinterpret_cast<float *>(partial_ws + (n256 > n128 ? n256 : n128) * 384);
    hipLaunchKernelGGL(pooled_bf16_init_kernel, dim3((unsigned)sonet::ceil_div64(npool, 256)), dim3(256), 0, st, pooled_ws, npool);
    int dev = 0, cus = 256;
    if (hipGetDevice(&dev) == hipSuccess) {
        int v = 0;
        if (hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && v > 0) cus = v;
    }
    if (two) {
        long long grid = ntiles < 2ll * cus ? ntiles : 2ll * cus;            // persistent: two workgroups per CU
        if (const char *eg = sonet::knob("SONET_BF16_POOL2_GRID")) { const long long v = atoll(eg); if (v > 0 && v < grid) grid = v; }   // debugging
        hipLaunchKernelGGL(pointresnet_bf16_pool2_kernel, dim3((unsigned)grid), dim3(P2_THREADS), 0, st,
                           x_sorted, Cin0, reinterpret_cast<const uint4 *>(wstream), reinterpret_cast<const float2 *>(affine), L, tpc, ntiles,
                           ids_sorted, pos0, pooled_ws, partial_ws, v0_ws, M);
    } else {
        const long long grid = ntiles < cus ? ntiles : cus;
        hipLaunchKernelGGL(pointresnet_bf16_kernel<true>, dim3((unsigned)grid), dim3(FB_THREADS), 0, st,
                           x_sorted, Cin0, reinterpret_cast<const uint4 *>(wstream), reinterpret_cast<const float2 *>(affine), (uint16_t *)nullptr,
                           L, tpc, ntiles, sonet::knob("SONET_BF16_FUSED_ABLATE") ? atoi(sonet::knob("SONET_BF16_FUSED_ABLATE")) : 0,
                           ids_sorted, pos0, node_off, count, pooled_ws, partial_ws, v0_ws, M);
    }
    hipLaunchKernelGGL(pooled_bf16_decode_kernel, dim3((unsigned)sonet::ceil_div64(npool, 256)), dim3(256), 0, st, pooled_ws, partial_ws,
                       ids_sorted, node_off, count, v0_ws, out, M, L, tpc, npool, tile_pts, slots);
    return sonet::launched(what);
}
