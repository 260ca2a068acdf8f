// pointmlp.hip -- fused point-wise layer (Conv 1x1 + per-channel affine + ReLU) on exact-f32 MFMA.
//
// Replaces, per layer, the three aten launches of EquivariantLayer.forward (models/layers.py:282-296:
// conv1d -> batch_norm -> relu) and MyConv2d.forward (:199-211); the channel concat of
// PointResNet.forward (models/layers.py:431) is fused as a second input panel.
//
//   y[b][o][l] = act( (sum_i W[o][i] * xcat[b][i][l]) * scale[o] + shift[o] )
//
// GEMM view: D[Cout x P] = W[Cout x Cin] . X[Cin x P] with P = B*L points.  Cin is tiny (6..515) and
// P is huge (960k at B=64, N=5000, k=3), so the layer is a "tall-skinny" GEMM whose X panel is read
// once and whose W is shared by every workgroup.
//
// MFMA mapping (v_mfma_f32_32x32x2_f32, exact f32 = an fma chain, guide section 3):
//   A (32 x 2)  <- W     lane l: A[i = l&31][k = l>>5]
//   B (2 x 32)  <- X     lane l: B[k = l>>5][j = l&31]     j runs along POINTS -> the 32 lanes of a
//                        half-wave read 128 contiguous bytes of one channel row (the B x C x L layout
//                        is already "N-major" for this operand: no transpose, no LDS staging).
//   D (32 x 32)          lane l, reg r: D[row = (r&3) + 8*(r>>2) + 4*(l>>5)][col = l&31]
//                        -> per register the half-waves store 128 contiguous bytes of one output row.
// W is pre-packed once per weight update (sonet_pointmlp_pack_f32) into exactly the A-fragment order:
//   Wp[ct][g][lane][s] = W[ct*32 + (lane&31)][8*g + 2*s + (lane>>5)],  s = 0..3, zero padded,
// so one global_load_dwordx4 per lane feeds four K-steps and the whole wave reads 1 KiB contiguous.
// W (<= 2 MB) is read by every wave and stays L1/L2-resident; HBM sees X once per cout pass and Y once.
//
// Work split: a wave owns NT groups of 32 consecutive points of one cloud and MT cout-tiles at a time
// (MT*NT accumulators of 16 VGPRs); a 256-thread workgroup = 4 waves on 4 adjacent point groups (they
// hit the same W lines together).  gridDim.y splits the cout tiles when there are too few point
// groups to fill 256 CUs (the node-level layers with L = 64 or 576 columns per cloud).
#include "common.hpp"
#include <stdlib.h>

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int PM_THREADS = 256;
constexpr int PM_WAVES = PM_THREADS / 64;

__global__ __launch_bounds__(256) void pointmlp_pack_kernel(const float *__restrict__ W, float *__restrict__ Wp,
                                                             int Cin, int Cout, int G, long long total)
{
    const long long t = (long long)blockIdx.x * 256 + threadIdx.x;
    if (t >= total) return;
    const int s = (int)(t & 3);
    const int lane = (int)((t >> 2) & 63);
    const long long r = t >> 8;            // ct*G + g
    const int g = (int)(r % G);
    const int ct = (int)(r / G);
    const int o = ct * 32 + (lane & 31);
    const int i = 8 * g + 2 * s + (lane >> 5);
    Wp[t] = (o < Cout && i < Cin) ? W[(long long)o * Cin + i] : 0.f;
}

// Operands of one K-group (8 input channels = 4 MFMA K-steps) for MT cout tiles x NT point groups.
template <int MT, int NT>
struct PmFrag {
    float4 a[MT];
    float b[NT][4];
};

// Loads are UNCONDITIONAL (addresses are clamped by the caller; hipcc would otherwise branch around
// every guarded load and drain vmcnt per element).  TAIL handles the last, partially filled group of
// a panel: channel index clamped into range, value zeroed by a select after the load.
template <int MT, int NT, bool TAIL>
__device__ __forceinline__ void pm_load(PmFrag<MT, NT> &f, const float4 *__restrict__ wp4, long long wstride,
                                        const float *const (&xp)[NT], long long L, int c0 /*first channel of the group*/,
                                        int Cx, int h)
{
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) f.a[mt] = wp4[(long long)mt * wstride];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            const int c = c0 + 2 * s + h;
            if constexpr (TAIL) {
                const int cc = c < Cx ? c : Cx - 1;
                const float v = xp[nt][(long long)cc * L];
                f.b[nt][s] = c < Cx ? v : 0.f;
            } else {
                f.b[nt][s] = xp[nt][(long long)c * L];
            }
        }
    }
}

template <int MT, int NT>
__device__ __forceinline__ void pm_mfma(f32x16 (&acc)[MT][NT], const PmFrag<MT, NT> &f)
{
#pragma unroll
    for (int s = 0; s < 4; ++s) {
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
            const float av = s == 0 ? f.a[mt].x : s == 1 ? f.a[mt].y : s == 2 ? f.a[mt].z : f.a[mt].w;
#pragma unroll
            for (int nt = 0; nt < NT; ++nt)
                acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, f.b[nt][s], acc[mt][nt], 0, 0, 0);
        }
    }
}

// Both input panels as ONE sequence of K-groups (x1's groups, then x2's), software-pipelined with two
// named operand sets in ping-pong: the loads of group g+1 are issued BEFORE the MFMAs of group g and
// stay in flight across them (hipcc then emits a counted s_waitcnt vmcnt(N), not a drain), so L2/HBM
// latency hides under 4*MT*NT MFMAs of 64 cycles each even at 1-2 waves per SIMD.  A single-set
// "load next; compute current; current = next" loop gets rotated back into load-then-wait by hipcc.
// Only the very last group can be partial (C1 % 8 == 0 whenever x2 exists): it runs as the TAIL.
template <int MT, int NT>
__device__ __forceinline__ void pm_accumulate(f32x16 (&acc)[MT][NT], const float *const (&xp1)[NT], int C1,
                                              const float *const (&xp2)[NT], int C2, long long L,
                                              const float4 *__restrict__ wp4 /*cout tile ct0, group 0, + lane*/,
                                              long long wstride /*float4 stride between cout tiles = G*64*/, int h)
{
    const int G1 = C1 >> 3;                                   // full groups of x1
    const int nfull = G1 + (C2 >> 3);
    const int Clast = C2 > 0 ? C2 : C1;
    auto load_full = [&](PmFrag<MT, NT> &f, int g) {
        const bool second = g >= G1;
        const float *xp[NT];
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) xp[nt] = second ? xp2[nt] : xp1[nt];
        pm_load<MT, NT, false>(f, wp4 + (long long)g * 64, wstride, xp, L, 8 * (second ? g - G1 : g), 0, h);
    };
    PmFrag<MT, NT> fa, fb;
    int g = 0;
    if (nfull > 0) {
        load_full(fa, 0);
        for (; g + 2 <= nfull; g += 2) {
            load_full(fb, g + 1);
            pm_mfma<MT, NT>(acc, fa);
            load_full(fa, g + 2 < nfull ? g + 2 : g + 1);      // clamped re-load on the last trip (branch-free)
            pm_mfma<MT, NT>(acc, fb);
        }
        if (g < nfull) pm_mfma<MT, NT>(acc, fa);
    }
    if (Clast & 7) {
        const int gl = nfull;                                  // the partial group
        if (C2 > 0) pm_load<MT, NT, true>(fb, wp4 + (long long)gl * 64, wstride, xp2, L, 8 * (gl - G1), C2, h);
        else        pm_load<MT, NT, true>(fb, wp4 + (long long)gl * 64, wstride, xp1, L, 8 * gl, C1, h);
        pm_mfma<MT, NT>(acc, fb);
    }
}

template <int MT, int NT>
__global__ __launch_bounds__(PM_THREADS) void pointmlp_f32_kernel(
    const float *__restrict__ x1, int C1, const float *__restrict__ x2, int C2, const float *__restrict__ Wp,
    const float *__restrict__ scale, const float *__restrict__ shift, int relu, float *__restrict__ y,
    int Cout, int L, int gpc /*32-point groups per cloud*/, long long ngroups, int CT, int G, int ct_per_y)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int j = lane & 31, h = lane >> 5;

    const float *xp1[NT], *xp2[NT];
    bool pv[NT];
    long long ybase[NT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
        const long long q = ((long long)blockIdx.x * PM_WAVES + wave) * NT + nt;   // global 32-point group
        const long long b = q / gpc;
        const int l = (int)(q - b * gpc) * 32 + j;
        pv[nt] = q < ngroups && l < L;
        const long long bc = pv[nt] ? b : 0;               // invalid lanes read cloud 0 / point 0, never store
        const int lc = pv[nt] ? l : 0;
        xp1[nt] = x1 + (bc * C1) * (long long)L + lc;
        xp2[nt] = x2 ? x2 + (bc * C2) * (long long)L + lc : x1;
        ybase[nt] = (bc * Cout) * (long long)L + lc;
    }

    const int ct_begin = blockIdx.y * ct_per_y;
    const int ct_end = min(CT, ct_begin + ct_per_y);
    for (int ct0 = ct_begin; ct0 < ct_end; ct0 += MT) {     // CT and ct_per_y are multiples of MT
        f32x16 acc[MT][NT];
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[mt][nt][r] = 0.f;

        const float4 *wp4 = reinterpret_cast<const float4 *>(Wp) + (long long)ct0 * G * 64 + lane;
        pm_accumulate<MT, NT>(acc, xp1, C1, xp2, C2, L, wp4, (long long)G * 64, h);

#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int o = (ct0 + mt) * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
                if (o < Cout) {
                    const float sc = scale[o], sh = shift[o];
#pragma unroll
                    for (int nt = 0; nt < NT; ++nt) {
                        float v = __fmaf_rn(acc[mt][nt][r], sc, sh);
                        if (relu) v = (v < 0.f) ? 0.f : v;      // NaN propagates like aten's relu
                        if (pv[nt]) y[ybase[nt] + (long long)o * L] = v;
                    }
                }
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------------
// v2: W through LDS, staged S K-groups at a time.
// In the kernel above every wave streams its own copy of the packed W from L1/L2 (MT KB per K-group per
// wave): identical data for all waves of a CU.  Here the 4 waves of a workgroup share ONE copy per
// STAGE of S K-groups (8*S input channels):
//   * each wave loads 1/4 of the next-next stage's S*MT W slices into registers (async-STAGE split:
//     issue early, ds_write after the next barrier), so global latency hides under a whole stage of
//     MFMAs (S*4*MT MFMAs = 256*S*MT cycles per wave);
//   * LDS holds two stages (2 x S x MT KB); all waves read their A fragments with conflict-free
//     lane-linear ds_read_b128; ONE __syncthreads per stage, not per K-group (measured: with a barrier
//     per group the loop reached only 67 % of the MFMA rate even with stores and X loads ablated);
//   * X (the B operand) stays a direct register load -- it is private to the wave (its own 32 points)
//     and already coalesced -- prefetched a whole stage ahead (4*S dword loads in flight per wave),
//     ping-ponged between two named register sets so hipcc keeps the loads in flight (counted vmcnt).
// ABL: bench-only ablation mask (tools/microbench.py): 1 = skip the epilogue stores, 2 = skip the X loads.
template <int MT, int S, int ABL = 0>
__global__ __launch_bounds__(PM_THREADS) void pointmlp_f32_wlds_kernel(
    const float *__restrict__ x1, int C1, const float *__restrict__ x2, int C2, const float *__restrict__ Wp,
    const float *__restrict__ scale, const float *__restrict__ shift, int relu, float *__restrict__ y,
    int Cout, int L, int gpc, long long ngroups, int CT, int G, int ct_per_y)
{
    __shared__ float4 wsm[2][S * MT][64];
    __shared__ float2 affine[1024];                          // (scale, shift) of this workgroup's cout range
    constexpr int NSL = S * MT;                               // W slices per stage
    constexpr int NS = (NSL + PM_WAVES - 1) / PM_WAVES;      // ... staged per wave
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int j = lane & 31, h = lane >> 5;

    const long long q = (long long)blockIdx.x * PM_WAVES + wave;
    const long long b = q / gpc;
    const int l = (int)(q - b * gpc) * 32 + j;
    const bool pv = q < ngroups && l < L;
    const long long bc = pv ? b : 0;
    const int lc = pv ? l : 0;
    const float *xp1 = x1 + (bc * C1) * (long long)L + lc;
    const float *xp2 = x2 ? x2 + (bc * C2) * (long long)L + lc : x1;
    const long long ybase = (bc * Cout) * (long long)L + lc;
    const int G1 = C2 > 0 ? (C1 >> 3) : G;                   // K-groups fed by x1
    const int nstage = (G + S - 1) / S;

    // B operands of stage st: S groups x 4 values per lane.  Channels past the panel (and the padded
    // groups g >= G of the last stage) are clamped for the load and zeroed by a select afterwards.
    auto load_b = [&](float (&bv)[S][4], int st) {
#pragma unroll
        for (int i = 0; i < S; ++i) {
            const int g = st * S + i;
            const bool second = g >= G1;
            const float *xp = second ? xp2 : xp1;
            const int Cx = (g < G) ? (second ? C2 : C1) : 0;
            const int c0 = 8 * (second ? g - G1 : g);
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                const int c = c0 + 2 * s + h;
                int cc = c < Cx ? c : Cx - 1;
                cc = cc < 0 ? 0 : cc;
                if constexpr (ABL & 2) { bv[i][s] = (float)(cc + lane) * 1e-3f; continue; }
                const float v = xp[(long long)cc * L];
                bv[i][s] = c < Cx ? v : 0.f;
            }
        }
    };

    const int ct_begin = blockIdx.y * ct_per_y;
    const int ct_end = min(CT, ct_begin + ct_per_y);
    // epilogue constants into LDS once: the epilogue must not start with a dependent global load
    for (int o = ct_begin * 32 + (int)threadIdx.x; o < ct_end * 32 && o - ct_begin * 32 < 1024; o += PM_THREADS)
        affine[o - ct_begin * 32] = o < Cout ? make_float2(scale[o], shift[o]) : make_float2(0.f, 0.f);
    for (int ct0 = ct_begin; ct0 < ct_end; ct0 += MT) {
        f32x16 acc[MT];
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[mt][r] = 0.f;

        const float4 *wsrc = reinterpret_cast<const float4 *>(Wp) + (long long)ct0 * G * 64 + lane;
        // slice index sl in [0, S*MT): group i = sl / MT of the stage, cout tile mt = sl % MT
        auto stage_load = [&](float4 (&w)[NS], int st) {
#pragma unroll
            for (int t = 0; t < NS; ++t) {
                int sl = wave + t * PM_WAVES;
                sl = sl < NSL ? sl : NSL - 1;                  // surplus lanes of the last round re-load a valid slice
                const int i = sl / MT, mt = sl - i * MT;
                int g = st * S + i;
                g = g < G ? g : G - 1;                          // padded groups: any valid slice (their B is zero)
                w[t] = wsrc[((long long)mt * G + g) * 64];
            }
        };
        auto stage_write = [&](const float4 (&w)[NS], int slot) {
#pragma unroll
            for (int t = 0; t < NS; ++t) {
                const int sl = wave + t * PM_WAVES;
                if (sl < NSL) wsm[slot][sl][lane] = w[t];
            }
        };
        auto compute = [&](const float (&bv)[S][4], int slot) {
#pragma unroll
            for (int i = 0; i < S; ++i) {
                float4 a[MT];
#pragma unroll
                for (int mt = 0; mt < MT; ++mt) {
                    if constexpr (ABL & 4) a[mt] = make_float4(bv[i][0] + mt, bv[i][1], bv[i][2], bv[i][3] + slot);
                    else a[mt] = wsm[slot][i * MT + mt][lane];
                }
#pragma unroll
                for (int s = 0; s < 4; ++s)
#pragma unroll
                    for (int mt = 0; mt < MT; ++mt) {
                        const float av = s == 0 ? a[mt].x : s == 1 ? a[mt].y : s == 2 ? a[mt].z : a[mt].w;
                        acc[mt] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv[i][s], acc[mt], 0, 0, 0);
                    }
            }
        };

        float4 wreg[NS];
        float b0[S][4], b1[S][4];
        __syncthreads();                                       // previous pass finished reading both slots
        stage_load(wreg, 0);
        load_b(b0, 0);
        stage_write(wreg, 0);
        stage_load(wreg, nstage > 1 ? 1 : 0);
        // one stage: barrier; publish W(st+1); prefetch W(st+2) and X(st+1); compute stage st
#define PM_STAGE(st, bcur, bnxt, slot)                                       \
        {                                                                    \
            if constexpr (!(ABL & 4)) {                                      \
            __syncthreads();                                                 \
            if ((st) + 1 < nstage) stage_write(wreg, (slot) ^ 1);            \
            stage_load(wreg, (st) + 2 < nstage ? (st) + 2 : nstage - 1); }   \
            load_b(bnxt, (st) + 1 < nstage ? (st) + 1 : nstage - 1);         \
            compute(bcur, slot);                                             \
        }
        int st = 0;
        for (; st + 2 <= nstage; st += 2) {
            PM_STAGE(st, b0, b1, 0)
            PM_STAGE(st + 1, b1, b0, 1)
        }
        if (st < nstage) PM_STAGE(st, b0, b1, 0)
#undef PM_STAGE

#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
            __builtin_amdgcn_sched_barrier(0);                 // keep the epilogue tile-by-tile (bounds VGPR live ranges)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int o = (ct0 + mt) * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
                if (o < Cout) {
                    const float2 ss = affine[o - ct_begin * 32];
                    float v = __fmaf_rn(acc[mt][r], ss.x, ss.y);
                    if (relu) v = (v < 0.f) ? 0.f : v;
                    if constexpr (ABL & 1) { asm volatile("" ::"v"(v)); continue; }
                    if (pv) y[ybase + (long long)o * L] = v;
                }
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------------
// v3 ("lean"): same data flow as v2 (W through LDS in stages, X prefetched a stage ahead), but the
// steady-state loop carries NO ordinary VALU instructions.
// Measured on MI355X (tools/mfma_issue.hip, profiles/r01_mfma_issue.log): v_mfma_f32_32x32x2_f32 runs at
// the f32 VECTOR rate and every ordinary VALU instruction of a resident wave takes matrix-pipe time:
// 0 / 4 / 8 / 16 filler VALU per MFMA -> 155 / 126 / 112 / 90 TFLOP/s at 4 waves per SIMD.  v2 carried
// ~5 VALU per MFMA (64-bit address mads, clamps, selects) and topped out at 117 TFLOP/s even with all
// memory traffic ablated.  Here every address is  <buffer descriptor> + <lane offset VGPR, computed once>
// + <wave-uniform SGPR offset>  (raw buffer loads / stores; SALU arithmetic is free), and LDS addresses
// are one VGPR plus compile-time immediates.  The descriptor of an X panel covers exactly the Cx*L floats
// of this wave's cloud, so rows past the panel (the zero-padded channels of a partial K-group, padded
// groups of the last stage) read as 0 in hardware: no clamps, no selects, one loader for every stage.
// Requires Cout % 32 == 0 and panels < 4 GiB per cloud (all layers of the path); other shapes use v2.
typedef int i32x4_t __attribute__((ext_vector_type(4)));

template <int MT, int S>
__global__ __launch_bounds__(PM_THREADS) void pointmlp_f32_lean_kernel(
    const float *__restrict__ x1, int C1, const float *__restrict__ x2, int C2, const float *__restrict__ Wp,
    const float *__restrict__ scale, const float *__restrict__ shift, int relu, float *__restrict__ y,
    int Cout, int L, int gpc, long long ngroups, int CT, int G, int ct_per_y)
{
    constexpr int NSL = S * MT;                               // W slices (1 KiB each) per stage
    constexpr int NS = (NSL + PM_WAVES - 1) / PM_WAVES;      // ... staged per wave
    __shared__ float4 wsm[2][NS * PM_WAVES][64];             // rounded up: surplus slices land in unused rows
    __shared__ float2 affine[1024];

    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int j = lane & 31, h = lane >> 5;

    // wave-uniform placement: this wave's 32 points of cloud b
    long long q = (long long)blockIdx.x * PM_WAVES + wave;
    const bool wave_valid = q < ngroups;
    q = wave_valid ? q : 0;
    const long long b = q / gpc;
    const int l0 = (int)(q - b * gpc) * 32;
    const bool pv = wave_valid && (l0 + j < L);
    const int lc = (l0 + j < L) ? l0 + j : l0;               // clamped point (never stored when invalid)

    const unsigned rowB = (unsigned)L * 4u;                   // bytes per channel row
    const __amdgpu_buffer_rsrc_t r1 = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float *>(x1 + b * (long long)C1 * L), 0, (int)((unsigned)C1 * rowB), 0x00020000);
    const __amdgpu_buffer_rsrc_t r2 = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float *>(x2 ? x2 + b * (long long)C2 * L : x1), 0, (int)((unsigned)(x2 ? C2 : 0) * rowB), 0x00020000);
    const __amdgpu_buffer_rsrc_t ry = __builtin_amdgcn_make_buffer_rsrc(
        y + b * (long long)Cout * L, 0, (int)((unsigned)Cout * rowB), 0x00020000);
    const __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float *>(Wp), 0, (int)((unsigned)CT * (unsigned)G * 1024u), 0x00020000);
    const unsigned vox = (unsigned)(h * L + lc) * 4u;          // lane byte offset inside a channel-pair of rows
    const unsigned voy = (unsigned)(4 * h * L + lc) * 4u;      // ... inside an output row quad (D rows r and r+4)
    const unsigned vow = (unsigned)lane * 16u;

    const int G1 = C2 > 0 ? (C1 >> 3) : G;                   // K-groups fed by x1 (C1 % 8 == 0 when x2 exists)
    const int nstage = (G + S - 1) / S;

    auto load_b = [&](float (&bv)[S][4], int st) {
#pragma unroll
        for (int i = 0; i < S; ++i) {
            const int g = st * S + i;
            const bool second = g >= G1;
            const unsigned row0 = (unsigned)(8 * (second ? g - G1 : g)) * rowB;     // scalar
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                const unsigned so = row0 + (unsigned)(2 * s) * rowB;
                bv[i][s] = __builtin_bit_cast(float, second ? __builtin_amdgcn_raw_buffer_load_b32(r2, vox, so, 0)
                                                            : __builtin_amdgcn_raw_buffer_load_b32(r1, vox, so, 0));
            }
        }
    };

    const int ct_begin = blockIdx.y * ct_per_y;
    const int ct_end = min(CT, ct_begin + ct_per_y);
    for (int o = ct_begin * 32 + (int)threadIdx.x; o < ct_end * 32; o += PM_THREADS)
        affine[o - ct_begin * 32] = make_float2(scale[o], shift[o]);

    for (int ct0 = ct_begin; ct0 < ct_end; ct0 += MT) {
        f32x16 acc[MT];
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[mt][r] = 0.f;

        // slice sl of a stage: K-group i = sl / MT, cout tile mt = sl % MT  (all scalar)
        auto stage_load = [&](i32x4_t (&w)[NS], int st) {
#pragma unroll
            for (int t = 0; t < NS; ++t) {
                int sl = wave + t * PM_WAVES;
                sl = sl < NSL ? sl : NSL - 1;
                const int i = sl / MT, mt = sl - i * MT;
                int g = st * S + i;
                g = g < G ? g : G - 1;
                w[t] = __builtin_amdgcn_raw_buffer_load_b128(rw, vow, (unsigned)((ct0 + mt) * G + g) * 1024u, 0);
            }
        };
        auto stage_write = [&](const i32x4_t (&w)[NS], int slot) {
#pragma unroll
            for (int t = 0; t < NS; ++t)
                wsm[slot][wave + t * PM_WAVES][lane] = __builtin_bit_cast(float4, w[t]);
        };
        auto compute = [&](const float (&bv)[S][4], int slot) {
#pragma unroll
            for (int i = 0; i < S; ++i) {
                float4 a[MT];
#pragma unroll
                for (int mt = 0; mt < MT; ++mt) a[mt] = wsm[slot][i * MT + mt][lane];
#pragma unroll
                for (int s = 0; s < 4; ++s)
#pragma unroll
                    for (int mt = 0; mt < MT; ++mt) {
                        const float av = s == 0 ? a[mt].x : s == 1 ? a[mt].y : s == 2 ? a[mt].z : a[mt].w;
                        acc[mt] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv[i][s], acc[mt], 0, 0, 0);
                    }
            }
        };

        i32x4_t wreg[NS];
        float b0[S][4], b1[S][4];
        __syncthreads();                                       // previous pass done with both slots / affine ready
        stage_load(wreg, 0);
        load_b(b0, 0);
        stage_write(wreg, 0);
        stage_load(wreg, nstage > 1 ? 1 : 0);
        // one stage: barrier; publish W(st+1) (unconditionally: a guarded ds_write lets LLVM sink the W load
        // next to it and expose its latency); prefetch W(st+2) and X(st+1); compute stage st
#define PM_STAGE(st, bcur, bnxt, slot)                                        \
        {                                                                    \
            __syncthreads();                                                 \
            stage_write(wreg, (slot) ^ 1);                                   \
            stage_load(wreg, (st) + 2 < nstage ? (st) + 2 : nstage - 1);     \
            load_b(bnxt, (st) + 1);                                          \
            compute(bcur, slot);                                             \
        }
        int st = 0;
        for (; st + 2 <= nstage; st += 2) {
            PM_STAGE(st, b0, b1, 0)
            PM_STAGE(st + 1, b1, b0, 1)
        }
        if (st < nstage) PM_STAGE(st, b0, b1, 0)
#undef PM_STAGE

        if (pv) {
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) {
                const unsigned so_tile = (unsigned)((ct0 + mt) * 32) * rowB;
                const float2 *aff = affine + (ct0 + mt - ct_begin) * 32 + 4 * h;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int orow = (r & 3) + 8 * (r >> 2);             // + 4h is folded into voy / aff
                    const float2 ss = aff[orow];
                    float v = __fmaf_rn(acc[mt][r], ss.x, ss.y);
                    if (relu) v = (v < 0.f) ? 0.f : v;
                    __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(int, v), ry, voy, so_tile + (unsigned)orow * rowB, 0);
                }
            }
        }
    }
}

// ---- training-mode BatchNorm support: per-channel statistics and the normalise + ReLU pass --------
constexpr int ST_THREADS = 256;

// grid (chunks, C): each workgroup reduces one chunk of one channel over (b, l); f64 partials are
// combined with f64 atomics into stat_ws[2*C] = {sum[c], sumsq[c]}.
__global__ __launch_bounds__(ST_THREADS) void channel_stats_kernel(const float *__restrict__ y, int B, int C, int L,
                                                                    double *__restrict__ stat_ws)
{
    const int c = blockIdx.y;
    const long long per_c = (long long)B * L;
    const long long chunk = (per_c + gridDim.x - 1) / gridDim.x;
    const long long beg = (long long)blockIdx.x * chunk, end = min(per_c, beg + chunk);
    double s = 0.0, s2 = 0.0;
    for (long long t = beg + threadIdx.x; t < end; t += ST_THREADS) {
        const long long b = t / L;
        const float v = y[(b * C + c) * (long long)L + (t - b * L)];
        s += (double)v;
        s2 += (double)v * (double)v;
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        s += __shfl_down(s, off, 64);
        s2 += __shfl_down(s2, off, 64);
    }
    __shared__ double red[2][ST_THREADS / 64];
    if ((threadIdx.x & 63) == 0) { red[0][threadIdx.x >> 6] = s; red[1][threadIdx.x >> 6] = s2; }
    __syncthreads();
    if (threadIdx.x == 0) {
        double a = 0.0, a2 = 0.0;
        for (int w = 0; w < ST_THREADS / 64; ++w) { a += red[0][w]; a2 += red[1][w]; }
        unsafeAtomicAdd(&stat_ws[c], a);
        unsafeAtomicAdd(&stat_ws[C + c], a2);
    }
}

__global__ __launch_bounds__(256) void channel_stats_finalize_kernel(const double *__restrict__ stat_ws, int C, double inv_n,
                                                                      float *__restrict__ mean, float *__restrict__ var, const sonet::BnRider rd)
{
    const int c = blockIdx.x * 256 + threadIdx.x;
    if (c >= C) return;
    const double m = stat_ws[c] * inv_n;
    double v = stat_ws[C + c] * inv_n - m * m;
    if (v < 0.0) v = 0.0;
    mean[c] = (float)m;
    var[c] = (float)v;
    if (rd.gamma != nullptr) {                                  // (BatchNorm rider, common.hpp: coefficients + running update, operation for operation)
        const float mf = (float)m, vf = (float)v;
        const float is = 1.0f / __fsqrt_rn(vf + rd.eps);
        const float s_ = rd.gamma[c] * is;
        rd.invstd[c] = is;
        rd.sc[c] = s_;
        rd.sh[c] = rd.beta[c] - mf * s_;
        if (rd.rmean != nullptr) {
            const float mo = rd.momentum;
            rd.rmean[c] = __fmaf_rn(mf, mo, __fmul_rn(rd.rmean[c], 1.0f - mo));
            rd.rvar[c] = __fmaf_rn(__fmul_rn(vf, rd.unbias), mo, __fmul_rn(rd.rvar[c], 1.0f - mo));
        }
    }
}

__global__ __launch_bounds__(256) void channel_affine_act_kernel(float *__restrict__ y, const float *__restrict__ scale,
                                                                  const float *__restrict__ shift, int relu, int C, int L,
                                                                  long long total)
{
    for (long long t = (long long)blockIdx.x * 256 + threadIdx.x; t < total; t += (long long)gridDim.x * 256) {
        const int c = (int)((t / L) % C);
        float v = __fmaf_rn(y[t], scale[c], shift[c]);
        if (relu) v = (v < 0.f) ? 0.f : v;
        y[t] = v;
    }
}

}  // namespace

extern "C" size_t sonet_pointmlp_pack_size(int Cin, int Cout)
{
    if (Cin <= 0 || Cout <= 0) return 0;
    return (size_t)sonet::ceil_div(Cout, 32) * sonet::ceil_div(Cin, 8) * 256;
}

extern "C" int sonet_pointmlp_pack_f32(const float *W, float *Wp, int Cin, int Cout, sonet_stream_t stream)
{
    const char *what = "sonet_pointmlp_pack_f32";
    SONET_REQUIRE(W && Wp, "%s: NULL pointer", what);
    SONET_REQUIRE(Cin > 0 && Cout > 0, "%s: non-positive size", what);
    const long long total = (long long)sonet_pointmlp_pack_size(Cin, Cout);
    hipLaunchKernelGGL(pointmlp_pack_kernel, dim3((unsigned)sonet::ceil_div64(total, 256)), dim3(256), 0,
                       sonet::as_stream(stream), W, Wp, Cin, Cout, sonet::ceil_div(Cin, 8), total);
    return sonet::launched(what);
}

extern "C" int sonet_pointmlp_f32(const float *x1, int C1, const float *x2, int C2, const float *Wp,
                                  const float *scale, const float *shift, int relu, float *y,
                                  int B, int Cout, int L, sonet_stream_t stream)
{
    const char *what = "sonet_pointmlp_f32";
    SONET_REQUIRE(x1 && Wp && scale && shift && y, "%s: NULL pointer", what);
    SONET_REQUIRE(B > 0 && C1 > 0 && C2 >= 0 && Cout > 0 && L > 0, "%s: non-positive size", what);
    SONET_REQUIRE((C2 == 0) == (x2 == nullptr), "%s: x2 and C2 disagree", what);
    SONET_REQUIRE(C2 == 0 || C1 % 8 == 0, "%s: with a second input C1=%d must be a multiple of 8", what, C1);
    const int Cin = C1 + C2;
    const int CT = sonet::ceil_div(Cout, 32), G = sonet::ceil_div(Cin, 8);
    const int gpc = sonet::ceil_div(L, 32);
    const long long ngroups = (long long)B * gpc;
    hipStream_t st = sonet::as_stream(stream);
    // Tile policy, measured on MI355X (tools/microbench.py, profiles/r01_microbench_pointmlp_lean.log) for the
    // lean kernel: as many cout tiles per pass as divide CT (6, else 4, else 2) -- fewer passes over the X
    // panel -- with one K-group per LDS stage; two per stage for the short-K / few-tile layers.
    // The 32-tile node-level layer (768->1024 on 64 columns per cloud) prefers MT = 2 (more workgroups).
    const long long nwg_x = sonet::ceil_div64(ngroups, (long long)PM_WAVES);
    int MT = 1, S = 1;
    if (CT % 6 == 0) MT = 6;
    else if (CT % 4 == 0 && !(nwg_x < 64)) MT = 4;
    else if (CT % 2 == 0) MT = 2;
    if (MT == 2 || (MT == 4 && CT <= 8)) S = 2;
    if (G == 1) S = 1;
    if (const char *e = sonet::knob("SONET_POINTMLP_MT")) {      // tuning knob (bench experiments only)
        const int want = atoi(e);
        if ((want == 8 || want == 6 || want == 4 || want == 2) && CT % want == 0) MT = want;
    }
    if (nwg_x > 0x7FFFFFFFll) return sonet::fail(SONET_ERR_UNSUPPORTED, "%s: too many points", what);
    // split the cout passes over gridDim.y when the point axis alone cannot fill the chip
    int ysplit = 1;
    while (nwg_x * ysplit < 1024 && (CT / MT) % (ysplit * 2) == 0) ysplit *= 2;
    while (CT / ysplit > 32 && (CT / MT) % (ysplit * 2) == 0) ysplit *= 2;   // LDS affine table holds 1024 channels
    const int ct_per_y = CT / ysplit;
    if (ct_per_y > 32) return sonet::fail(SONET_ERR_UNSUPPORTED, "%s: Cout=%d too large for one pass table", what, Cout);
    dim3 grid((unsigned)nwg_x, (unsigned)ysplit), block(PM_THREADS);
    if (const char *e = sonet::knob("SONET_POINTMLP_S")) {       // tuning knob (bench experiments only)
        const int want = atoi(e);
        if (want == 1 || want == 2 || want == 4) S = want;
    }
    int abl = 0;
    if (const char *e = sonet::knob("SONET_POINTMLP_ABLATE")) abl = atoi(e);   // bench-only: no stores / no X loads
#define PM_ARGS grid, block, 0, st, x1, C1, x2, C2, Wp, scale, shift, relu, y, Cout, L, gpc, ngroups, CT, G, ct_per_y
#ifdef SONET_VARIANTS
#define PM_LAUNCH_V2(MM)                                                                              \
    do {                                                                                              \
        if (abl == 1 && S == 2)      hipLaunchKernelGGL((pointmlp_f32_wlds_kernel<MM, 2, 1>), PM_ARGS); \
        else if (abl == 2 && S == 2) hipLaunchKernelGGL((pointmlp_f32_wlds_kernel<MM, 2, 2>), PM_ARGS); \
        else if (abl == 3 && S == 2) hipLaunchKernelGGL((pointmlp_f32_wlds_kernel<MM, 2, 3>), PM_ARGS); \
        else if (abl == 4 && S == 2) hipLaunchKernelGGL((pointmlp_f32_wlds_kernel<MM, 2, 4>), PM_ARGS); \
        else if (abl == 7 && S == 2) hipLaunchKernelGGL((pointmlp_f32_wlds_kernel<MM, 2, 7>), PM_ARGS); \
        else if (S == 4)             hipLaunchKernelGGL((pointmlp_f32_wlds_kernel<MM, 4, 0>), PM_ARGS); \
        else if (S == 2)             hipLaunchKernelGGL((pointmlp_f32_wlds_kernel<MM, 2, 0>), PM_ARGS); \
        else                         hipLaunchKernelGGL((pointmlp_f32_wlds_kernel<MM, 1, 0>), PM_ARGS); \
    } while (0)
#else                                                          /* product: no ablation instantiations */
#define PM_LAUNCH_V2(MM)                                                                              \
    do {                                                                                              \
        (void)abl;                                                                                    \
        if (S == 4)                  hipLaunchKernelGGL((pointmlp_f32_wlds_kernel<MM, 4, 0>), PM_ARGS); \
        else if (S == 2)             hipLaunchKernelGGL((pointmlp_f32_wlds_kernel<MM, 2, 0>), PM_ARGS); \
        else                         hipLaunchKernelGGL((pointmlp_f32_wlds_kernel<MM, 1, 0>), PM_ARGS); \
    } while (0)
#endif
#define PM_LAUNCH(MM, NN) hipLaunchKernelGGL((pointmlp_f32_kernel<MM, NN>), PM_ARGS)
    const char *kenv = sonet::knob("SONET_POINTMLP_KERNEL");      // "wlds" forces v2 (bench A/B only)
    const bool fits32 = (double)(C1 > C2 ? C1 : C2) * L * 4.0 < 4.0e9 && (double)Cout * L * 4.0 < 4.0e9 &&
                        (double)CT * G * 1024.0 < 4.0e9;
    const bool lean = (Cout % 32 == 0) && (ct_per_y <= 32) && fits32 && !(kenv && kenv[0] == 'w') && abl == 0;
#define PM_LAUNCH_LEAN(MM)                                                                            \
    do {                                                                                              \
        if (S == 4)      hipLaunchKernelGGL((pointmlp_f32_lean_kernel<MM, 4>), PM_ARGS);              \
        else if (S == 2) hipLaunchKernelGGL((pointmlp_f32_lean_kernel<MM, 2>), PM_ARGS);              \
        else             hipLaunchKernelGGL((pointmlp_f32_lean_kernel<MM, 1>), PM_ARGS);              \
    } while (0)
    if (lean && (MT == 2 || MT == 4 || MT == 6)) {
        if (MT == 2) PM_LAUNCH_LEAN(2); else if (MT == 4) PM_LAUNCH_LEAN(4); else PM_LAUNCH_LEAN(6);
        return sonet::launched(what);
    }
#undef PM_LAUNCH_LEAN
    switch (MT) {
        case 8: PM_LAUNCH_V2(8); break;
        case 6: PM_LAUNCH_V2(6); break;
        case 4: PM_LAUNCH_V2(4); break;
        case 2: PM_LAUNCH_V2(2); break;
        default: PM_LAUNCH(1, 1);                       // odd CT (Cout <= 32 or not a multiple of 64)
    }
#undef PM_LAUNCH
#undef PM_LAUNCH_V2
#undef PM_ARGS
    return sonet::launched(what);
}

extern "C" int sonet_channel_stats_f32(const float *y, int B, int C, int L, double *stat_ws, float *mean,
                                       float *var_biased, sonet_stream_t stream)
{
    const char *what = "sonet_channel_stats_f32";
    SONET_REQUIRE(y && stat_ws && mean && var_biased, "%s: NULL pointer", what);
    SONET_REQUIRE(B > 0 && C > 0 && L > 0, "%s: non-positive size", what);
    if (C > 65535) return sonet::fail(SONET_ERR_UNSUPPORTED, "%s: C=%d > 65535", what, C);
    hipStream_t st = sonet::as_stream(stream);
    if (hipMemsetAsync(stat_ws, 0, (size_t)2 * C * sizeof(double), st) != hipSuccess)
        return sonet::fail(SONET_ERR_LAUNCH, "%s: hipMemsetAsync failed", what);
    const long long per_c = (long long)B * L;
    int chunks = (int)sonet::ceil_div64(per_c, 16384);
    if (chunks < 1) chunks = 1;
    if (chunks > 64) chunks = 64;
    hipLaunchKernelGGL(channel_stats_kernel, dim3(chunks, C), dim3(ST_THREADS), 0, st, y, B, C, L, stat_ws);
    hipLaunchKernelGGL(channel_stats_finalize_kernel, dim3(sonet::ceil_div(C, 256)), dim3(256), 0, st,
                       stat_ws, C, 1.0 / (double)per_c, mean, var_biased, sonet::take_bn_rider());
    return sonet::launched(what);
}

extern "C" int sonet_channel_affine_act_f32(float *y, const float *scale, const float *shift, int relu,
                                            int B, int C, int L, sonet_stream_t stream)
{
    const char *what = "sonet_channel_affine_act_f32";
    SONET_REQUIRE(y && scale && shift, "%s: NULL pointer", what);
    SONET_REQUIRE(B > 0 && C > 0 && L > 0, "%s: non-positive size", what);
    const long long total = (long long)B * C * L;
    long long blocks = sonet::ceil_div64(total, 256);
    if (blocks > 256 * 32) blocks = 256 * 32;
    hipLaunchKernelGGL(channel_affine_act_kernel, dim3((unsigned)blocks), dim3(256), 0, sonet::as_stream(stream),
                       y, scale, shift, relu, C, L, total);
    return sonet::launched(what);
}
