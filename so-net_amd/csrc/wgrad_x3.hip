// wgrad_x3.hip -- weight gradient of a point-wise (1x1 conv) layer on the bf16 matrix cores at f32-class accuracy:
//      dW[o][c] = sum_b sum_l g[b][o][l] * x[b][c][l]            g [B][Cout][L], x [B][Cin][L], dW [Cout][Cin]  (f32)
// This is what the reference's autograd computes for nn.Conv1d / nn.Conv2d(1x1) weights (models/layers.py:282-296 ->
// cudnn wgrad); round 1 ran it as torch.bmm(g, x^T).sum(0) = a hipBLASLt f32 GEMM per cloud batch (1.0 ms of the
// 12 ms training step).  The reduction axis is the LONG one (B * L = 960 000 columns at the benchmark shape), the output is
// small (<= 1024 x 768): every workgroup owns a 128 x 128 output block over a slice of the columns, partial blocks go to a
// workspace and are summed in a fixed order by a second kernel (deterministic, no float atomics).
//
// Arithmetic: both operands are f32 tensors with unrelated dynamic ranges (gradients 1e-7 ... 1e-1), so they are split into
// three bf16 pieces each (f32's exponent range, 3 x 8 significand bits) and the six products of weight <= 2^-16 are kept --
// the dgrad launches' arithmetic (pointmlp_x3.hip), ~2^-23 relative per product, f32 accumulation inside the MFMA.
//
// Data flow per 16-column step: wave w loads 32 rows x 16 columns of g (output tile w) and of x (input tile w) -- each lane 8
// consecutive floats of its row, two steps = 64 bytes per lane, 128 consecutive bytes per row --, splits both in registers,
// publishes its three g pieces (3 KiB) in LDS, and multiplies ALL four g tiles (read back from LDS) with its own x pieces:
// 24 MFMAs per wave and step for 88 vector instructions of splitting (interleaved by sched_group_barrier: a wave issues in
// order), one barrier per step, LDS double buffered.  Algorithmic bytes: (Cout + Cin) * B * L * 4 read once per 128-wide
// block of the other operand; 128 -> 256 at 64 x 15000 columns: 1.47 GB = 0.18 ms at 8 TB/s, 63 GFLOP x 6 = 0.23 ms at the
// 1.65 PFLOP/s the chip sustains (DESIGN.md finding 8): balanced by construction.
#include "common.hpp"
#include <stdlib.h>

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

constexpr int WG_THREADS = 256;                                 // 4 waves
constexpr int WG_BLK = 128;                                     // output block edge: 4 tiles of 32
constexpr int WG_UNIT = 32;                                     // columns per loop iteration = two MFMA steps

__device__ __forceinline__ unsigned wg_cvt_pk_bf16(float lo, float hi) {
    unsigned r;
    asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(r) : "v"(lo), "v"(hi));
    return r;
}
// (x0, x1) -> three packed bf16 pairs (hi, mid, lo), round-to-nearest at each level, residuals exact in f32
__device__ __forceinline__ void wg_split3_pair(float x0, float x1, unsigned &h, unsigned &m, unsigned &l) {
    h = wg_cvt_pk_bf16(x0, x1);
    const float r0 = x0 - __uint_as_float(h << 16), r1 = x1 - __uint_as_float(h & 0xFFFF0000u);
    m = wg_cvt_pk_bf16(r0, r1);
    const float q0 = r0 - __uint_as_float(m << 16), q1 = r1 - __uint_as_float(m & 0xFFFF0000u);
    l = wg_cvt_pk_bf16(q0, q1);
}

// 16 floats of one row: columns [l, l + 16) of `row` (zeros past L or when the row does not exist)
// (An out-of-line element-wise form for the rare partial unit halved the code size -- 8000 -> 4800 lines of ISA -- and changed
// nothing for whole blocks (419 vs 421 us) while the scratch arrays of the call cost the guarded variant 20 %: not kept.  The big
// shapes run at 4.7 TB/s of 128-byte row segments 60 KB apart, which is what this access pattern gets from HBM; hipBLASLt's f32
// kernel sits at 3.7 TB/s on the same operands.)
__device__ __forceinline__ void wg_load16(const float *__restrict__ row, bool row_ok, int l, int L, bool vec, float (&v)[16]) {
    if (row_ok && vec && l + 16 <= L) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const float4 t = *reinterpret_cast<const float4 *>(row + l + 4 * q);
            v[4 * q] = t.x; v[4 * q + 1] = t.y; v[4 * q + 2] = t.z; v[4 * q + 3] = t.w;
        }
    } else {
#pragma unroll
        for (int e = 0; e < 16; ++e) v[e] = (row_ok && l + e < L) ? row[l + e] : 0.f;
    }
}

// XAFF: x is the RAW output of a BatchNorm layer whose normalise + ReLU pass was never run -- the split of the x operand applies
// x = act(raw * xs[c] + xh[c]) first, the arithmetic of sonet_channel_affine_act_f32 bit for bit (a lane's 16 floats are one row: one
// coefficient pair per lane).  Columns past L need no mask: g is zero there.
template <bool FULL /*every tile of every block exists: no guards in the step*/, bool XAFF = false>
__global__ __launch_bounds__(WG_THREADS, 2) void wgrad_x3_kernel(
    const float *__restrict__ g, const float *__restrict__ x, float *__restrict__ partial,
    int Cout, int Cin, int L, int nL /*32-column units per cloud*/, long long units /*B * nL*/, int nsplit, int oblocks, int cblocks,
    const float *__restrict__ xs = nullptr, const float *__restrict__ xh = nullptr, int xrelu = 0)
{
    __shared__ uint4 apieces[2][4][3][64];                      // [buffer][g tile][piece h, m, l][lane]: 2 x 12 KiB

    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int i = lane & 31, h = lane >> 5;
    // (output block fastest: the workgroups running together walk the SAME columns, so a g / x row panel fetched for one block is
    // in L2 for the blocks that share it)
    const int nblk = oblocks * cblocks;
    const int blk = blockIdx.x % nblk, sp = blockIdx.x / nblk;
    const int ob = blk / cblocks, cb = blk - ob * cblocks;
    const long long u0 = units * sp / nsplit, u1 = units * (sp + 1) / nsplit;

    const int o_row = ob * WG_BLK + wave * 32 + i, c_row = cb * WG_BLK + wave * 32 + i;
    const bool o_ok = o_row < Cout, c_ok = c_row < Cin;
    const bool vec = (L & 3) == 0;
    // FULL: no branches in the step, the scheduling region stays whole.  Otherwise tiles past Cout / Cin are skipped (wave-uniform
    // guards): a 64 x 6 gradient would spend 15/16 of its MFMAs on zero rows.
    const int n_ot = FULL ? 4 : min(4, (Cout - ob * WG_BLK + 31) >> 5);
    const bool c_tile = FULL || cb * WG_BLK + wave * 32 < Cin;
    float xsc = 0.f, xsh = 0.f;                                 // (a row past Cin: act(0 * 0 + 0) = 0)
    if constexpr (XAFF) { if (c_ok) { xsc = xs[c_row]; xsh = xh[c_row]; } }

    f32x16 acc[4];
#pragma unroll
    for (int ot = 0; ot < 4; ++ot)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[ot][r] = 0.f;

    auto load_unit = [&](long long u, float (&ra)[16], float (&rb)[16]) {
        const int b = (int)(u / nL);
        const int l = (int)(u - (long long)b * nL) * WG_UNIT + 16 * h;
        wg_load16(g + ((size_t)b * Cout + (o_ok ? o_row : 0)) * L, o_ok, l, L, vec, ra);
        wg_load16(x + ((size_t)b * Cin + (c_ok ? c_row : 0)) * L, c_ok, l, L, vec, rb);
    };
    auto split8x = [&](const float (&raw)[16], int s, uint4 (&pc)[3]) {    // the x operand: (XAFF) normalise + ReLU, then the pieces
        unsigned ph[4], pm[4], pl[4];
#pragma unroll
        for (int p = 0; p < 4; ++p) {
            float v0 = raw[8 * s + 2 * p], v1 = raw[8 * s + 2 * p + 1];
            if constexpr (XAFF) {
                v0 = __fmaf_rn(v0, xsc, xsh); v1 = __fmaf_rn(v1, xsc, xsh);
                if (xrelu) { v0 = (v0 < 0.f) ? 0.f : v0; v1 = (v1 < 0.f) ? 0.f : v1; }
            }
            wg_split3_pair(v0, v1, ph[p], pm[p], pl[p]);
        }
        pc[0] = make_uint4(ph[0], ph[1], ph[2], ph[3]);
        pc[1] = make_uint4(pm[0], pm[1], pm[2], pm[3]);
        pc[2] = make_uint4(pl[0], pl[1], pl[2], pl[3]);
    };
    auto split8 = [&](const float (&raw)[16], int s, uint4 (&pc)[3]) {     // floats [8 s, 8 s + 8) -> pieces h, m, l
        unsigned ph[4], pm[4], pl[4];
#pragma unroll
        for (int p = 0; p < 4; ++p) wg_split3_pair(raw[8 * s + 2 * p], raw[8 * s + 2 * p + 1], ph[p], pm[p], pl[p]);
        pc[0] = make_uint4(ph[0], ph[1], ph[2], ph[3]);
        pc[1] = make_uint4(pm[0], pm[1], pm[2], pm[3]);
        pc[2] = make_uint4(pl[0], pl[1], pl[2], pl[3]);
    };
    auto publish = [&](const uint4 (&pa)[3], int buf) {
#pragma unroll
        for (int t = 0; t < 3; ++t) apieces[buf][wave][t][lane] = pa[t];
    };
    // the six products of weight <= 2^-16, smallest first; term-major so that consecutive MFMAs hit different accumulators
    auto mfmas = [&](int buf, const uint4 (&pb)[3]) {
        if (!FULL && !c_tile) return;
        const bf16x8 Bh = __builtin_bit_cast(bf16x8, pb[0]), Bm = __builtin_bit_cast(bf16x8, pb[1]), Bl = __builtin_bit_cast(bf16x8, pb[2]);
        bf16x8 Ah[4], Am[4], Al[4];
#pragma unroll
        for (int ot = 0; ot < 4; ++ot) {
            if (FULL || ot < n_ot) {
                Ah[ot] = __builtin_bit_cast(bf16x8, apieces[buf][ot][0][lane]);
                Am[ot] = __builtin_bit_cast(bf16x8, apieces[buf][ot][1][lane]);
                Al[ot] = __builtin_bit_cast(bf16x8, apieces[buf][ot][2][lane]);
            }
        }
#define WG_TERM(A_, B_) _Pragma("unroll") for (int ot = 0; ot < 4; ++ot) if (FULL || ot < n_ot) acc[ot] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A_[ot], B_, acc[ot], 0, 0, 0);
        WG_TERM(Al, Bh) WG_TERM(Ah, Bl) WG_TERM(Am, Bm) WG_TERM(Am, Bh) WG_TERM(Ah, Bm) WG_TERM(Ah, Bh)
#undef WG_TERM
    };
    auto interleave = [&]() {                                   // one MFMA, four of the next step's split instructions in its shadow
#pragma unroll
        for (int k = 0; k < 24; ++k) {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x002, 4, 0);
        }
    };
    auto step_barrier = [&]() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); };   // (not __syncthreads: it would drain the prefetch)

    float ra0[16], rb0[16], ra1[16], rb1[16], ra2[16], rb2[16];
    uint4 pa[3], pb[3], pa_n[3], pb_n[3];
    if (u0 < u1) {
        load_unit(u0, ra0, rb0);
        if (u0 + 1 < u1) load_unit(u0 + 1, ra1, rb1);
        split8(ra0, 0, pa);
        split8x(rb0, 0, pb);
        int buf = 0;
        // iteration: raw of unit u in (ra_c, rb_c), its step-0 pieces already in (pa, pb), unit u + 1 on its way into (ra_n, rb_n);
        // requests unit u + 2 into (ra_f, rb_f).  (One unit of look-ahead left every 32-column step waiting for memory: 2 us per
        // unit against 0.64 us of MFMAs.)
#define WG_ITER(u, ra_c, rb_c, ra_n, rb_n, ra_f, rb_f)                                      \
        {                                                                                \
            const bool more = (u) + 1 < u1;                                              \
            if ((u) + 2 < u1) load_unit((u) + 2, ra_f, rb_f);                            \
            publish(pa, buf);                                                            \
            step_barrier();                                                              \
            mfmas(buf, pb);                                                              \
            split8(ra_c, 1, pa_n);                                                       \
            split8x(rb_c, 1, pb_n);                                                      \
            interleave();                                                                \
            buf ^= 1;                                                                    \
            publish(pa_n, buf);                                                          \
            step_barrier();                                                              \
            mfmas(buf, pb_n);                                                            \
            if (more) {                                                                  \
                split8(ra_n, 0, pa);                                                     \
                split8x(rb_n, 0, pb);                                                    \
            }                                                                            \
            interleave();                                                                \
            buf ^= 1;                                                                    \
        }
        long long u = u0;
        for (; u + 3 <= u1; u += 3) {
            WG_ITER(u, ra0, rb0, ra1, rb1, ra2, rb2)
            WG_ITER(u + 1, ra1, rb1, ra2, rb2, ra0, rb0)
            WG_ITER(u + 2, ra2, rb2, ra0, rb0, ra1, rb1)
        }
        if (u < u1) {
            WG_ITER(u, ra0, rb0, ra1, rb1, ra2, rb2)
            if (u + 1 < u1) WG_ITER(u + 1, ra1, rb1, ra2, rb2, ra0, rb0)
        }
#undef WG_ITER
    }

    // partial[sp][o][c], padded to whole blocks (the reduction kernel reads the valid part)
    const int Cpad = cblocks * WG_BLK;
    const size_t Opad = (size_t)oblocks * WG_BLK;
    float *pp = partial + ((size_t)sp * Opad + (size_t)ob * WG_BLK) * Cpad + cb * WG_BLK + wave * 32 + i;
#pragma unroll
    for (int ot = 0; ot < 4; ++ot)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int orow = ot * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
            pp[(size_t)orow * Cpad] = acc[ot][r];
        }
}

// ---- bf16 twin (BASELINE configs[1] "bf16": activations and gradients stored in bf16) --------------------------------------------
// g [B][Cout][L], x [B][Cin][L] as bfloat16 bit patterns: ONE v_mfma_f32_32x32x16_bf16 per product, f32 accumulation, f32 partial
// blocks, the same column slices and the same fixed-order reduction as above (a bf16 partial would cost three of the eight
// significand bits).  No split work and a sixth of the matrix work: the kernel is a stream of 16-byte row loads (a lane's 8 bf16 of
// one MFMA step) -- 2 bytes per element and pass -- with the g tiles handed round through LDS as in the f32 kernel.
// Per loop iteration a lane holds 32 consecutive columns of its g row and of its x row (four 16-byte loads each = four MFMA
// steps); the half-waves take alternate 8-column groups, which is a permutation of the reduction axis common to both operands.
constexpr int WB_UNIT = 64;                                     // columns per loop iteration = four MFMA steps
__device__ __forceinline__ void wb_load32(const uint16_t *__restrict__ row, bool row_ok, int l, int L, bool vec, uint4 (&v)[4]) {
    // this lane's four 8-column groups: columns l + 16 q + [0, 8), q = 0..3 (l already carries the half-wave's 8-column offset)
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int c0 = l + 16 * q;
        if (row_ok && vec && c0 + 8 <= L) {
            v[q] = *reinterpret_cast<const uint4 *>(row + c0);
        } else {
            unsigned w[4] = {0u, 0u, 0u, 0u};
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const unsigned val = (row_ok && c0 + e < L) ? (unsigned)row[c0 + e] : 0u;
                w[e >> 1] |= val << (16 * (e & 1));
            }
            v[q] = make_uint4(w[0], w[1], w[2], w[3]);
        }
    }
}

template <bool FULL>
__global__ __launch_bounds__(WG_THREADS, 2) void wgrad_bf16_kernel(
    const uint16_t *__restrict__ g, const uint16_t *__restrict__ x, float *__restrict__ partial,
    int Cout, int Cin, int L, int nL /*64-column units per cloud*/, long long units /*B * nL*/, int nsplit, int oblocks, int cblocks)
{
    __shared__ uint4 apieces[2][4][4][64];                      // [buffer][g tile][step][lane]: 2 x 16 KiB

    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int i = lane & 31, h = lane >> 5;
    const int nblk = oblocks * cblocks;
    const int blk = blockIdx.x % nblk, sp = blockIdx.x / nblk;
    const int ob = blk / cblocks, cb = blk - ob * cblocks;
    const long long u0 = units * sp / nsplit, u1 = units * (sp + 1) / nsplit;
    const int o_row = ob * WG_BLK + wave * 32 + i, c_row = cb * WG_BLK + wave * 32 + i;
    const bool o_ok = o_row < Cout, c_ok = c_row < Cin;
    const bool vec = (L & 7) == 0;                              // 16-byte aligned 8-column groups
    const int n_ot = FULL ? 4 : min(4, (Cout - ob * WG_BLK + 31) >> 5);
    const bool c_tile = FULL || cb * WG_BLK + wave * 32 < Cin;

    f32x16 acc[4];
#pragma unroll
    for (int ot = 0; ot < 4; ++ot)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[ot][r] = 0.f;

    auto load_unit = [&](long long u, uint4 (&ra)[4], uint4 (&rb)[4]) {
        const int b = (int)(u / nL);
        const int l = (int)(u - (long long)b * nL) * WB_UNIT + 8 * h;
        wb_load32(g + ((size_t)b * Cout + (o_ok ? o_row : 0)) * L, o_ok, l, L, vec, ra);
        wb_load32(x + ((size_t)b * Cin + (c_ok ? c_row : 0)) * L, c_ok, l, L, vec, rb);
    };
    auto step_barrier = [&]() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); };   // (not __syncthreads: it would drain the prefetch)

    uint4 ra0[4], rb0[4], ra1[4], rb1[4], ra2[4], rb2[4];
    if (u0 < u1) {
        load_unit(u0, ra0, rb0);
        if (u0 + 1 < u1) load_unit(u0 + 1, ra1, rb1);
        int buf = 0;
        // iteration: unit u in (ra_c, rb_c), unit u + 1 on its way into (ra_n, rb_n); requests unit u + 2 into (ra_f, rb_f)
#define WB_ITER(u, ra_c, rb_c, ra_f, rb_f)                                                \
        {                                                                                \
            if ((u) + 2 < u1) load_unit((u) + 2, ra_f, rb_f);                            \
            _Pragma("unroll") for (int q = 0; q < 4; ++q) apieces[buf][wave][q][lane] = ra_c[q]; \
            step_barrier();                                                              \
            if (FULL || c_tile) {                                                        \
                _Pragma("unroll") for (int q = 0; q < 4; ++q) {                          \
                    const bf16x8 Bq = __builtin_bit_cast(bf16x8, rb_c[q]);               \
                    _Pragma("unroll") for (int ot = 0; ot < 4; ++ot)                     \
                        if (FULL || ot < n_ot)                                           \
                            acc[ot] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, apieces[buf][ot][q][lane]), Bq, acc[ot], 0, 0, 0); \
                }                                                                        \
            }                                                                            \
            buf ^= 1;                                                                    \
        }
        long long u = u0;
        for (; u + 3 <= u1; u += 3) {
            WB_ITER(u, ra0, rb0, ra2, rb2)
            WB_ITER(u + 1, ra1, rb1, ra0, rb0)
            WB_ITER(u + 2, ra2, rb2, ra1, rb1)
        }
        if (u < u1) {
            WB_ITER(u, ra0, rb0, ra2, rb2)
            if (u + 1 < u1) WB_ITER(u + 1, ra1, rb1, ra0, rb0)
        }
#undef WB_ITER
    }
    const int Cpad = cblocks * WG_BLK;
    const size_t Opad = (size_t)oblocks * WG_BLK;
    float *pp = partial + ((size_t)sp * Opad + (size_t)ob * WG_BLK) * Cpad + cb * WG_BLK + wave * 32 + i;
#pragma unroll
    for (int ot = 0; ot < 4; ++ot)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int orow = ot * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
            pp[(size_t)orow * Cpad] = acc[ot][r];
        }
}

// ---- bf16 wgrad, streaming generation: both operands through an LDS ring filled by LDS-DMA ------------------------------------------
// The kernel above hands every lane 16-byte pieces of ITS row: a load instruction touches 32 rows x 32 bytes (2.8 TB/s on the
// point-level shapes; hipBLASLt reaches 3.1-3.9 there).  Here a persistent workgroup (one per CU) owns a (GT x 128) x 128 block of dw
// and a slice of the 64-column units; a unit's g rows and x rows -- (GT x 128 + 128) rows x 128 bytes -- go straight from memory
// into one of three LDS slots (global_load_lds_dwordx4: eight lanes per row, i.e. whole 128-byte row segments per request, no VGPRs),
// two units ahead of the MFMAs; every element is loaded once per workgroup.  16-byte piece q of row r sits at slot position
// q ^ (r & 7) (the DMA writes lanes linearly: the lane fetches the piece that belongs at its position), so that the 16-byte fragment
// reads of 32 rows spread over the banks.  Rows past Cout / Cin and columns past L are fetched from a zeroed 16 bytes.  One barrier
// per unit: "my DMAs of this unit have landed" (s_waitcnt vmcnt(N_DMA): the next unit's may be in flight) + barrier = everyone's
// have, and everyone has finished the previous unit, whose slot the next request then overwrites.  f32 partial blocks per slice,
// summed by wgrad_reduce_kernel in a fixed order.
// XAFF (bf16 training with normalise-on-load): x holds the RAW output of a BatchNorm layer; a wave applies act(raw * xs[c] + xh[c]) -- f32 fma,
// round to nearest even, ReLU: what sonet_channel_affine_act_bf16 would have stored -- to the x fragment it has just read from LDS (a lane's
// eight values are one channel row: one coefficient pair per lane and x tile, held in registers for the whole launch).  Columns past L and
// rows past Cin come from the zeroed 16 bytes and would turn into act(xh): their g partners are zero as well (same zeroed source), the
// product is unchanged.
constexpr int WS_UNIT = 64;
typedef float ws_f2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ unsigned ws_cvt_pk_bf16(float lo, float hi) {
    unsigned r;
    asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(r) : "v"(lo), "v"(hi));
    return r;
}
__device__ __forceinline__ uint4 ws_xaff(uint4 v, float sc, float sh, unsigned floor2) {
    unsigned d[4] = {v.x, v.y, v.z, v.w};
    const ws_f2 s2 = {sc, sc}, h2 = {sh, sh};
#pragma unroll
    for (int p = 0; p < 4; ++p) {
        const ws_f2 xx = {__uint_as_float(d[p] << 16), __uint_as_float(d[p] & 0xFFFF0000u)};
        ws_f2 r;
        asm("v_pk_fma_f32 %0, %1, %2, %3" : "=v"(r) : "v"(xx), "v"(s2), "v"(h2));
        unsigned o = ws_cvt_pk_bf16(r[0], r[1]);
        asm("v_pk_max_i16 %0, %1, %2" : "=v"(o) : "v"(o), "v"(floor2));
        d[p] = o;
    }
    return make_uint4(d[0], d[1], d[2], d[3]);
}
template <int GT, bool XAFF = false>
__global__ __launch_bounds__(256, 1) void wgrad_bf16s_kernel(const uint16_t *__restrict__ g, const uint16_t *__restrict__ x,
                                                             const uint4 *__restrict__ zero16, float *__restrict__ partial,
                                                             int Cout, int Cin, int L, int nL, long long units, int nsplit, int oblocks, int cblocks,
                                                             const float *__restrict__ xs = nullptr, const float *__restrict__ xh = nullptr, int xrelu = 0)
{
    constexpr int OBR = GT * 128, ROWS = OBR + 128, NDMA = ROWS / 32;   // DMA requests per wave and unit (64 lanes x 16 bytes = 8 rows each)
    extern __shared__ uint4 ws_ring[];                                   // [3][ROWS][8] 16-byte pieces
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int m = lane & 31, h = lane >> 5;
    const int nblk = oblocks * cblocks;
    const int blk = blockIdx.x % nblk, sp = blockIdx.x / nblk;
    const int ob = blk / cblocks, cb = blk - ob * cblocks;
    const long long u0 = units * sp / nsplit, u1 = units * (sp + 1) / nsplit;
    const unsigned lds0 = (unsigned)reinterpret_cast<size_t>(ws_ring);
    const int nxt = min(4, (Cin - cb * 128 + 31) >> 5);                  // x tiles that exist
    const bool g_any = ob * OBR + wave * GT * 32 < Cout;
    float xsc[4] = {0.f, 0.f, 0.f, 0.f}, xsh[4] = {0.f, 0.f, 0.f, 0.f};
    const unsigned xfloor = xrelu ? 0u : 0x80008000u;
    if constexpr (XAFF) {
#pragma unroll
        for (int xt = 0; xt < 4; ++xt) {
            const int c = cb * 128 + xt * 32 + m;
            if (c < Cin) { xsc[xt] = xs[c]; xsh[xt] = xh[c]; }
        }
    }

    f32x16 acc[GT][4];
#pragma unroll
    for (int gt = 0; gt < GT; ++gt)
#pragma unroll
        for (int xt = 0; xt < 4; ++xt)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[gt][xt][r] = 0.f;

    auto dma = [&](long long u, int slot) {
        const int b = (int)(u / nL);
        const int l0 = (int)(u - (long long)b * nL) * WS_UNIT;
#pragma unroll
        for (int j = 0; j < NDMA; ++j) {
            const int pos = (wave * NDMA + j) * 64 + lane;                // 16-byte position in the slot
            const int row = pos >> 3, q = (pos & 7) ^ (row & 7);         // the piece that belongs there
            const int col = l0 + 8 * q;
            const uint16_t *src;
            bool ok;
            if (row < OBR) {
                const int o = ob * OBR + row;
                ok = o < Cout && col < L;
                src = g + ((size_t)b * Cout + (ok ? o : 0)) * L + (ok ? col : 0);
            } else {
                const int c = cb * 128 + row - OBR;
                ok = c < Cin && col < L;
                src = x + ((size_t)b * Cin + (ok ? c : 0)) * L + (ok ? col : 0);
            }
            const void *addr = ok ? static_cast<const void *>(src) : static_cast<const void *>(zero16);
            const unsigned d = lds0 + (unsigned)slot * (unsigned)(ROWS * 128) + (unsigned)(wave * NDMA + j) * 1024u;
            unsigned keep;
            asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 4\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                         : "=&s"(keep) : "v"(addr), "s"(d) : "memory");
        }
    };

    if (u0 < u1) {
        dma(u0, 0);
        dma(u0 + 1 < u1 ? u0 + 1 : u1 - 1, 1);
        int slot = 0;
        for (long long u = u0; u < u1; ++u) {
            asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" :: "n"(NDMA) : "memory");
            dma(u + 2 < u1 ? u + 2 : u1 - 1, slot >= 1 ? slot - 1 : 2);         // (slot + 2) % 3: the slot of unit u - 1
            const unsigned char *sl = reinterpret_cast<const unsigned char *>(ws_ring) + (size_t)slot * (ROWS * 128);
            if (g_any) {
#pragma unroll
                for (int ks = 0; ks < 4; ++ks) {
                    const int pq = ((2 * ks + h) ^ (m & 7)) << 4;
                    bf16x8 A[GT];
#pragma unroll
                    for (int gt = 0; gt < GT; ++gt)
                        A[gt] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const uint4 *>(sl + ((wave * GT + gt) * 32 + m) * 128 + pq));
#pragma unroll
                    for (int xt = 0; xt < 4; ++xt) {
                        if (xt < nxt) {
                            uint4 bx = *reinterpret_cast<const uint4 *>(sl + (OBR + xt * 32 + m) * 128 + pq);
                            if constexpr (XAFF) bx = ws_xaff(bx, xsc[xt], xsh[xt], xfloor);
                            const bf16x8 Bx = __builtin_bit_cast(bf16x8, bx);
#pragma unroll
                            for (int gt = 0; gt < GT; ++gt)
                                acc[gt][xt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A[gt], Bx, acc[gt][xt], 0, 0, 0);
                        }
                    }
                }
            }
            slot = slot == 2 ? 0 : slot + 1;
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    const int Cpad = cblocks * 128;
    const size_t Opad = (size_t)oblocks * OBR;
#pragma unroll
    for (int gt = 0; gt < GT; ++gt)
#pragma unroll
        for (int xt = 0; xt < 4; ++xt) {
            float *pp = partial + ((size_t)sp * Opad + (size_t)ob * OBR + (wave * GT + gt) * 32) * Cpad + cb * 128 + xt * 32 + m;
#pragma unroll
            for (int r = 0; r < 16; ++r) pp[(size_t)((r & 3) + 8 * (r >> 2) + 4 * h) * Cpad] = acc[gt][xt][r];
        }
}

struct WsPlan { int gt, oblocks, cblocks, nL, nsplit; long long units; size_t part_bytes; };
static WsPlan ws_plan(int B, int Cout, int Cin, int L)
{
    WsPlan p;
    p.gt = Cout > 128 ? 2 : 1;
    p.oblocks = sonet::ceil_div(Cout, p.gt * 128);
    p.cblocks = sonet::ceil_div(Cin, 128);
    p.nL = sonet::ceil_div(L, WS_UNIT);
    p.units = (long long)B * p.nL;
    const int nblk = p.oblocks * p.cblocks;
    long long ns = 256 / nblk;                                  // one workgroup per CU (144 KiB of LDS)
    if (ns > p.units / 4) ns = p.units / 4;
    if (ns < 1) ns = 1;
    p.nsplit = (int)ns;
    p.part_bytes = (size_t)p.nsplit * p.oblocks * p.gt * 128 * p.cblocks * 128 * sizeof(float);
    return p;
}

// dw[o][c] = sum over the column slices, in a fixed order: thread (element e, kq) adds slices kq, kq + 4, ... with four independent
// chains, the four quarters meet in LDS -- 64 elements per workgroup (a 64 x 6 gradient summed over 1024 slices is 384 elements:
// one thread per element walking all slices took 270 us of dependent loads).
__global__ __launch_bounds__(256) void wgrad_reduce_kernel(const float *__restrict__ partial, float *__restrict__ dw,
                                                            int Cout, int Cin, int nsplit, size_t Opad, int Cpad)
{
    __shared__ float part[4][64];
    const int e = threadIdx.x & 63, kq = threadIdx.x >> 6;
    const long long t = (long long)blockIdx.x * 64 + e;
    const bool ok = t < (long long)Cout * Cin;
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    if (ok) {
        const int o = (int)(t / Cin), c = (int)(t - (long long)o * Cin);
        const float *p = partial + (size_t)o * Cpad + c;
        const size_t stride = Opad * (size_t)Cpad;
        int k = kq;
        for (; k + 12 < nsplit; k += 16) {
            s0 += p[(size_t)k * stride];
            s1 += p[(size_t)(k + 4) * stride];
            s2 += p[(size_t)(k + 8) * stride];
            s3 += p[(size_t)(k + 12) * stride];
        }
        for (; k < nsplit; k += 4) s0 += p[(size_t)k * stride];
    }
    part[kq][e] = (s0 + s1) + (s2 + s3);
    __syncthreads();
    if (kq == 0 && ok) dw[t] = (part[0][e] + part[1][e]) + (part[2][e] + part[3][e]);
}

struct WgPlan { int oblocks, cblocks, nL, nsplit; long long units; size_t ws_bytes; };
static WgPlan wg_plan(int B, int Cout, int Cin, int L, int unit = WG_UNIT)
{
    WgPlan p;
    p.oblocks = sonet::ceil_div(Cout, WG_BLK);
    p.cblocks = sonet::ceil_div(Cin, WG_BLK);
    p.nL = sonet::ceil_div(L, unit);
    p.units = (long long)B * p.nL;
    const int nblk = p.oblocks * p.cblocks;
    long long ns = 1024 / nblk;                                 // ~ 4 workgroups per CU in total
    if (ns > p.units / 4) ns = p.units / 4;                     // at least four 32-column units per slice
    if (ns < 1) ns = 1;
    p.nsplit = (int)ns;
    p.ws_bytes = (size_t)p.nsplit * p.oblocks * WG_BLK * p.cblocks * WG_BLK * sizeof(float);
    return p;
}

}  // namespace

extern "C" size_t sonet_wgrad_x3_ws_size(int B, int Cout, int Cin, int L)
{
    if (B <= 0 || Cout <= 0 || Cin <= 0 || L <= 0) return 0;
    return wg_plan(B, Cout, Cin, L).ws_bytes;
}

static int wgrad_x3_impl(const char *what, const float *g, const float *x, float *dw, void *ws, int B, int Cout, int Cin, int L,
                         const float *xs, const float *xh, int xrelu, sonet_stream_t stream)
{
    SONET_REQUIRE(g && x && dw && ws, "%s: NULL pointer", what);
    SONET_REQUIRE(B > 0 && Cout > 0 && Cin > 0 && L > 0, "%s: non-positive size", what);
    if ((double)Cout * L * 4.0 >= 8.0e9 || (double)Cin * L * 4.0 >= 8.0e9) return sonet::fail(SONET_ERR_UNSUPPORTED, "%s: a per-cloud panel is too large", what);
    const WgPlan p = wg_plan(B, Cout, Cin, L);
    const long long nwg = (long long)p.oblocks * p.cblocks * p.nsplit;
    if (nwg > 0x7FFFFFFFll) return sonet::fail(SONET_ERR_UNSUPPORTED, "%s: too large", what);
    hipStream_t st = sonet::as_stream(stream);
    const bool full = Cout % WG_BLK == 0 && Cin % WG_BLK == 0;
#define WGX_LAUNCH(FF, XX) hipLaunchKernelGGL((wgrad_x3_kernel<FF, XX>), dim3((unsigned)nwg), dim3(WG_THREADS), 0, st, g, x, reinterpret_cast<float *>(ws), \
                                              Cout, Cin, L, p.nL, p.units, p.nsplit, p.oblocks, p.cblocks, xs, xh, xrelu)
    if (xs) { if (full) WGX_LAUNCH(true, true); else WGX_LAUNCH(false, true); }
    else    { if (full) WGX_LAUNCH(true, false); else WGX_LAUNCH(false, false); }
#undef WGX_LAUNCH
    const long long n = (long long)Cout * Cin;
    hipLaunchKernelGGL(wgrad_reduce_kernel, dim3((unsigned)sonet::ceil_div64(n, 64)), dim3(256), 0, st, reinterpret_cast<const float *>(ws), dw,
                       Cout, Cin, p.nsplit, (size_t)p.oblocks * WG_BLK, p.cblocks * WG_BLK);
    return sonet::launched(what);
}

extern "C" int sonet_wgrad_x3_f32(const float *g, const float *x, float *dw, void *ws, int B, int Cout, int Cin, int L, sonet_stream_t stream)
{
    return wgrad_x3_impl("sonet_wgrad_x3_f32", g, x, dw, ws, B, Cout, Cin, L, nullptr, nullptr, 0, stream);
}

/* The same gradient when x is the RAW output of a BatchNorm layer: the operand split applies x = act(raw * xs[c] + xh[c]) first (xs, xh [Cin];
 * xrelu: ReLU), exactly what sonet_channel_affine_act_f32 would have stored. */
extern "C" int sonet_wgrad_x3_xaff_f32(const float *g, const float *x, float *dw, void *ws, int B, int Cout, int Cin, int L,
                                       const float *xs, const float *xh, int xrelu, sonet_stream_t stream)
{
    SONET_REQUIRE(xs && xh, "sonet_wgrad_x3_xaff_f32: NULL pointer");
    return wgrad_x3_impl("sonet_wgrad_x3_xaff_f32", g, x, dw, ws, B, Cout, Cin, L, xs, xh, xrelu, stream);
}

extern "C" size_t sonet_wgrad_bf16_ws_size(int B, int Cout, int Cin, int L)
{
    if (B <= 0 || Cout <= 0 || Cin <= 0 || L <= 0) return 0;
    const size_t a = wg_plan(B, Cout, Cin, L, WB_UNIT).ws_bytes, b = 256 + ws_plan(B, Cout, Cin, L).part_bytes;
    return a > b ? a : b;
}

static int wgrad_bf16_impl(const char *what, const uint16_t *g, const uint16_t *x, float *dw, void *ws, int B, int Cout, int Cin, int L, sonet_stream_t stream,
                           const float *xs, const float *xh, int xrelu)
{
    SONET_REQUIRE(g && x && dw && ws, "%s: NULL pointer", what);
    SONET_REQUIRE(B > 0 && Cout > 0 && Cin > 0 && L > 0, "%s: non-positive size", what);
    if ((double)Cout * L * 2.0 >= 8.0e9 || (double)Cin * L * 2.0 >= 8.0e9) return sonet::fail(SONET_ERR_UNSUPPORTED, "%s: a per-cloud panel is too large", what);
    if (((reinterpret_cast<uintptr_t>(g) | reinterpret_cast<uintptr_t>(x)) & 15) != 0) return sonet::fail(SONET_ERR_INVALID_ARG, "%s: g and x must be 16-byte aligned", what);
    hipStream_t st = sonet::as_stream(stream);
    {
        // long reductions with 16-byte aligned 8-column groups: the streaming generation (both operands through the LDS-DMA ring)
        bool want = (L & 7) == 0 && (long long)B * sonet::ceil_div(L, WS_UNIT) >= 2048;
        if (const char *e = sonet::knob("SONET_WGRAD_BF16_STREAM")) want = want && atoi(e) != 0;
        if (want) {
            const WsPlan q = ws_plan(B, Cout, Cin, L);
            if (hipMemsetAsync(ws, 0, 256, st) != hipSuccess) return sonet::fail(SONET_ERR_LAUNCH, "%s: memset failed", what);
            const uint4 *zero16 = reinterpret_cast<const uint4 *>(ws);
            float *part = reinterpret_cast<float *>(reinterpret_cast<char *>(ws) + 256);
            const size_t lds = (size_t)3 * (q.gt * 128 + 128) * 128;
            const dim3 grid((unsigned)(q.oblocks * q.cblocks * q.nsplit)), block(256);
#define WS_LAUNCH(GG) do { static bool attr_set = false;                                                                              \
                if (!attr_set) { if (hipFuncSetAttribute(reinterpret_cast<const void *>(&wgrad_bf16s_kernel<GG>),                    \
                                                         hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess)       \
                                     return sonet::fail(SONET_ERR_UNSUPPORTED, "%s: cannot reserve the LDS", what);                   \
                                 attr_set = true; }                                                                                   \
                hipLaunchKernelGGL(wgrad_bf16s_kernel<GG>, grid, block, lds, st, g, x, zero16, part, Cout, Cin, L, q.nL, q.units, q.nsplit, q.oblocks, q.cblocks); } while (0)
#define WS_LAUNCH_X(GG) do { static bool attr_set = false;                                                                            \
                if (!attr_set) { if (hipFuncSetAttribute(reinterpret_cast<const void *>(&wgrad_bf16s_kernel<GG, true>),              \
                                                         hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess)       \
                                     return sonet::fail(SONET_ERR_UNSUPPORTED, "%s: cannot reserve the LDS", what);                   \
                                 attr_set = true; }                                                                                   \
                hipLaunchKernelGGL((wgrad_bf16s_kernel<GG, true>), grid, block, lds, st, g, x, zero16, part, Cout, Cin, L, q.nL, q.units, q.nsplit, q.oblocks, q.cblocks, \
                                   xs, xh, xrelu); } while (0)
            if (xs) { if (q.gt == 2) WS_LAUNCH_X(2); else WS_LAUNCH_X(1); }
            else if (q.gt == 2) WS_LAUNCH(2); else WS_LAUNCH(1);
#undef WS_LAUNCH_X
#undef WS_LAUNCH
            const long long n = (long long)Cout * Cin;
            hipLaunchKernelGGL(wgrad_reduce_kernel, dim3((unsigned)sonet::ceil_div64(n, 64)), dim3(256), 0, st, part, dw,
                               Cout, Cin, q.nsplit, (size_t)q.oblocks * q.gt * 128, q.cblocks * 128);
            return sonet::launched(what);
        }
    }
    if (xs) return sonet::fail(SONET_ERR_UNSUPPORTED, "%s: normalise-on-load exists on the streaming kernel only (L %% 8 == 0, >= 2048 column units)", what);
    const WgPlan p = wg_plan(B, Cout, Cin, L, WB_UNIT);
    const long long nwg = (long long)p.oblocks * p.cblocks * p.nsplit;
    if (nwg > 0x7FFFFFFFll) return sonet::fail(SONET_ERR_UNSUPPORTED, "%s: too large", what);
    if (Cout % WG_BLK == 0 && Cin % WG_BLK == 0)
        hipLaunchKernelGGL(wgrad_bf16_kernel<true>, dim3((unsigned)nwg), dim3(WG_THREADS), 0, st, g, x, reinterpret_cast<float *>(ws),
                           Cout, Cin, L, p.nL, p.units, p.nsplit, p.oblocks, p.cblocks);
    else
        hipLaunchKernelGGL(wgrad_bf16_kernel<false>, dim3((unsigned)nwg), dim3(WG_THREADS), 0, st, g, x, reinterpret_cast<float *>(ws),
                           Cout, Cin, L, p.nL, p.units, p.nsplit, p.oblocks, p.cblocks);
    const long long n = (long long)Cout * Cin;
    hipLaunchKernelGGL(wgrad_reduce_kernel, dim3((unsigned)sonet::ceil_div64(n, 64)), dim3(256), 0, st, reinterpret_cast<const float *>(ws), dw,
                       Cout, Cin, p.nsplit, (size_t)p.oblocks * WG_BLK, p.cblocks * WG_BLK);
    return sonet::launched(what);
}

extern "C" int sonet_wgrad_bf16(const uint16_t *g, const uint16_t *x, float *dw, void *ws, int B, int Cout, int Cin, int L, sonet_stream_t stream)
{
    return wgrad_bf16_impl("sonet_wgrad_bf16", g, x, dw, ws, B, Cout, Cin, L, stream, nullptr, nullptr, 0);
}

/* sonet_wgrad_bf16 when x is the RAW (bf16) output of a BatchNorm layer whose normalise pass was never run (bf16 training with
 * normalise-on-load): x = act(raw * xs[c] + xh[c]) rounded to bf16 is applied to the fragments, bit for bit what sonet_channel_affine_act_bf16
 * would have stored; xs, xh [Cin].  Streaming-kernel shapes only (L % 8 == 0, B * ceil(L / 64) >= 2048): SONET_ERR_UNSUPPORTED otherwise. */
extern "C" int sonet_wgrad_bf16_xaff(const uint16_t *g, const uint16_t *x, float *dw, void *ws, int B, int Cout, int Cin, int L,
                                     const float *xs, const float *xh, int xrelu, sonet_stream_t stream)
{
    SONET_REQUIRE(xs && xh, "sonet_wgrad_bf16_xaff: NULL pointer");
    return wgrad_bf16_impl("sonet_wgrad_bf16_xaff", g, x, dw, ws, B, Cout, Cin, L, stream, xs, xh, xrelu);
}
