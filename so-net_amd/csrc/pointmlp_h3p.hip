// pointmlp_h3p.hip -- the fp16-split point-wise layer on PRE-SPLIT activations ("P16" planes): third generation of
//   y = act((W . cat(x1, x2)) * scale + shift)        (EquivariantLayer / MyConv2d 1x1, models/layers.py:282-296)
//
// What the second generation (pointmlp_x3.hip, pointmlp_h3r_kernel) spends per 16-channel chunk and 32 points: 8 dword loads of X,
// ~52 VALU instructions that clamp / scale / split the 8 values into fp16 pieces, 12 MFMAs -- and it does that again for every group
// of 4 output tiles.  Its ablations (docs/findings.md R3.6) price the split at 19 % and the X requests at 17 % of the layer.  Here the
// split runs ONCE, in the epilogue of the layer that PRODUCES the activation (or in a converter pass), and the consumer's operand
// loads are finished MFMA B fragments:
//
//   P16 layout of a B x C x L activation:  P[b][kc][form][h][l][8] fp16,   kc = 0 .. ceil(C/16)-1,
//       form 0 = fp16(32 x)  ("hi"),  form 1 = fp16(32 x - hi)  ("mid"; the residual is exact in f32 before it is rounded),
//       element e = 0..7 of half h  <->  channel 16 kc + 4 h + (e & 3) + 8 (e >> 2)          (zeros past C)
//   = 64 ceil(C/16) L bytes per cloud: the bytes of the f32 tensor.  One lane of a v_mfma_f32_32x32x16_f16 B operand (column j = lane & 31,
//   K half h = lane >> 5) is ONE 16-byte load per form; a half wave reads 512 contiguous bytes.  The channel order inside a chunk is
//   the order in which a 32x32 accumulator tile holds its rows (row (r & 3) + 8 (r >> 2) + 4 h in register r): registers 8q .. 8q+7
//   of output tile t ARE chunk 2t + q, half h, elements 0..7 of the next layer's input -- the producer stores two 16-byte vectors per
//   (tile, q) instead of 8 dwords, no shuffle.  The weight pack carries the same order (h3p_pack_kernel).
//
// Arithmetic: the two-operand-form split of the fused first PointNet (pointresnet_fused.hip):
//   1024 w x  ~=  fp16(32 w - Wh) . Xh  +  Wh . Xm  +  Wh . Xh,     Wh = fp16(32 w), Xh = fp16(32 x), Xm = fp16(32 x - Xh),
// three v_mfma_f32_32x32x16_f16 with f32 accumulation, smallest term first; the factor 1024 is exact, rides in the accumulator and leaves
// through scale / 1024.  Term by term this is 32 x the second generation's sum (Wr . xh + wh . Xm + wh . Xh with wh = fp16(w)), so the
// values differ from it only through the MFMA's internal summation order over the permuted K slots and through weights below the fp16
// normal range (kept more precisely here): f32-class, 1e-5 against the oracle like its predecessors (tests/test_gpu_h3p.py).
// Operand range: |x| <= 2047 is enforced where the split happens (the producer clamps and logs its post-activation maximum in word 2 of
// the range log; a consumer of P16 planes has nothing left to check on the activation side), |w| <= 2047 is logged from the pack.
//
// Pipeline (one workgroup = 4 waves x 32 NC columns, MT output tiles per pass, all passes of its output slab in ONE flat loop):
//   * W: global -> LDS by LDS-DMA (global_load_lds_dwordx4) into a ring of 5 one-chunk stages (MT x 2 KiB each), issued 4 chunks ahead;
//     every A fragment read from LDS feeds 3 NC / 2 MFMAs; fragments are read one tile pair ahead (32 registers, not MT x 8).
//   * X: 2 NC 16-byte loads per chunk, three chunks ahead, into a 4-deep register ring.
//   * EVERY vector-memory instruction of the loop is inline asm and every wait is written by hand: hipcc's s_waitcnt bookkeeping cannot
//     see LDS-DMA requests, and in the second generation its own vmcnt(N) for the X loads therefore waited on DMA requests issued a few
//     cycles earlier -- an L2 round trip per stage.  Here the count is (PB - 1) x (loads + DMA instructions per chunk) at every chunk,
//     the issue order is the same in the prologue and in steady state, and the waited-for registers are tied to the wait statement so
//     that no use can be scheduled above it.  One bare s_barrier per chunk (3 NC MT MFMAs per wave).
//   * the flat loop runs over (pass, chunk) with the chunk count rounded up to a multiple of 4 (the register ring's period; chunks past
//     the input read zeros through the buffer descriptor and meet the pack's zero padding): the pipeline never drains between passes --
//     the next pass's weights and X chunks are in flight while the epilogue of this one runs.
#include "common.hpp"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2_t __attribute__((ext_vector_type(2)));
typedef float f32x2_t __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef int i32x4_t __attribute__((ext_vector_type(4)));

constexpr int P_THREADS = 256;
constexpr int P_KPAD = 8;                                     // the pack's K range is padded with zero chunks to a multiple of this (as the h3 pack)
constexpr unsigned OOB = 0x7FFFFF00u;                          // a lane offset no descriptor covers: loads return 0, stores are dropped

__device__ __forceinline__ unsigned pk_f16(float lo, float hi) {
    const f32x2_t v = {lo, hi};
    return __builtin_bit_cast(unsigned, __builtin_convertvector(v, f16x2_t));
}
// (X0, X1) already scaled by 32 and inside +-65504 -> packed hi, packed mid (residual exact in f32, then rounded)
__device__ __forceinline__ void split2(float X0, float X1, unsigned &h, unsigned &m) {
    h = pk_f16(X0, X1);
    const f16x2_t hv = __builtin_bit_cast(f16x2_t, h);
    m = pk_f16(X0 - (float)hv[0], X1 - (float)hv[1]);
}
// activation side: clamp to the fp16-split range (a NaN leaves as the lower bound, as in the second generation), scale, split.
// RELU: the lower bound is 0 -- ReLU and clamp are one v_med3_f32.
template <bool RELU>
__device__ __forceinline__ void split_act(float x0, float x1, unsigned &h, unsigned &m) {
    constexpr float lo = RELU ? 0.f : -2047.f;
    split2(32.f * __builtin_amdgcn_fmed3f(x0, lo, 2047.f), 32.f * __builtin_amdgcn_fmed3f(x1, lo, 2047.f), h, m);
}

// channel of element e (0..7) of half h in a 16-channel chunk
__device__ __forceinline__ int p16_channel(int h, int e) { return 4 * h + (e & 3) + 8 * (e >> 2); }

// ---- weight pack: Wp[ct][kcp][form][lane] (uint4 = 8 fp16):  row ct*32 + (lane & 31), K slots 8 (lane >> 5) .. +7 of chunk kcp in P16 channel
// order; form 0 = fp16(32 w), form 1 = fp16(32 w - form 0); zero chunks from ceil(Cin/16) to KCP; 64-byte trailer: word 0 = bits of max |w|.
__global__ __launch_bounds__(256) void h3p_pack_kernel(const float *__restrict__ W, uint4 *__restrict__ Wp, int Cin, int Cout, int KCP,
                                                        long long total, unsigned *__restrict__ trailer)
{
    const long long t = (long long)blockIdx.x * 256 + threadIdx.x;       // (ct*KCP + kc)*64 + lane
    if (t >= total) return;
    RangeAcc wr = {0, 0u};
    const int lane = (int)(t & 63);
    const long long r = t >> 6;
    const int kc = (int)(r % KCP), ct = (int)(r / KCP);
    const int o = ct * 32 + (lane & 31), hh = lane >> 5;
    unsigned h[4], m[4];
#pragma unroll
    for (int p = 0; p < 4; ++p) {
        const int c0 = kc * 16 + p16_channel(hh, 2 * p), c1 = c0 + 1;
        float w0 = (o < Cout && c0 < Cin) ? W[(long long)o * Cin + c0] : 0.f;
        float w1 = (o < Cout && c1 < Cin) ? W[(long long)o * Cin + c1] : 0.f;
        range_track(wr, w0, w1);
        w0 = 32.f * __builtin_fminf(__builtin_fmaxf(w0, -2047.f), 2047.f);
        w1 = 32.f * __builtin_fminf(__builtin_fmaxf(w1, -2047.f), 2047.f);
        split2(w0, w1, h[p], m[p]);
    }
    uint4 *dst = Wp + (r * 2) * 64 + lane;
    dst[0] = make_uint4(h[0], h[1], h[2], h[3]);
    dst[64] = make_uint4(m[0], m[1], m[2], m[3]);
    range_publish(trailer, wave_umax(range_amax_bits(wr)), lane);
}

// ---- f32 [B][C][L] -> P16 planes, optionally through a per-channel affine + ReLU (BatchNorm normalise of the training forward:
// the separate normalise pass writes the next layer's operand instead of an f32 tensor of the same size).  One thread = one (chunk, half,
// column): 8 strided reads (each coalesced over the wave's 64 columns), two 16-byte stores.
template <bool AFFINE>
__global__ __launch_bounds__(256) void p16_from_f32_kernel(const float *__restrict__ x, uint4 *__restrict__ p, int C, int L, int KC,
                                                            const float *__restrict__ scale, const float *__restrict__ shift, int relu,
                                                            unsigned *__restrict__ rlog)
{
    __shared__ unsigned wmax[4];
    const int kc = blockIdx.y >> 1, hh = blockIdx.y & 1, b = blockIdx.z;
    RangeAcc xr = {0, 0u};
    const float *xb0 = x + (size_t)b * C * L;
    uint4 *pb = p + (size_t)b * KC * 4 * L;
    float sc[8], sf[8];
    if constexpr (AFFINE) {
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const int c = kc * 16 + p16_channel(hh, e);
            sc[e] = c < C ? scale[c] : 0.f;
            sf[e] = c < C ? shift[c] : 0.f;
        }
    }
    // 1024 columns per workgroup (four per thread, each step coalesced over the workgroup's 256 consecutive columns)
#pragma unroll
    for (int it = 0; it < 4; ++it) {
        const int l = blockIdx.x * 1024 + it * 256 + threadIdx.x;
        if (l >= L) break;
        const float *xb = xb0 + l;
        float v[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const int c = kc * 16 + p16_channel(hh, e);
            float t = c < C ? xb[(size_t)c * L] : 0.f;
            if constexpr (AFFINE) {
                t = __fmaf_rn(t, sc[e], sf[e]);
                if (relu) t = t < 0.f ? 0.f : t;
            }
            v[e] = t;
        }
        unsigned h[4], m[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            range_track(xr, v[2 * q], v[2 * q + 1]);
            split_act<false>(v[2 * q], v[2 * q + 1], h[q], m[q]);
        }
        pb[((size_t)(kc * 2 + 0) * 2 + hh) * L + l] = make_uint4(h[0], h[1], h[2], h[3]);
        pb[((size_t)(kc * 2 + 1) * 2 + hh) * L + l] = make_uint4(m[0], m[1], m[2], m[3]);
    }
    if (rlog != nullptr) {                                     // one atomic per workgroup at most (every wave publishing: ~60 us of contention on a 7 MB tensor)
        const unsigned wm = wave_umax(range_amax_bits(xr));
        if ((threadIdx.x & 63) == 0) wmax[threadIdx.x >> 6] = wm;
        __syncthreads();
        if (threadIdx.x == 0) {
            const unsigned a = wmax[0] > wmax[1] ? wmax[0] : wmax[1], c2 = wmax[2] > wmax[3] ? wmax[2] : wmax[3];
            const unsigned mx = a > c2 ? a : c2;
            if (mx > __atomic_load_n(rlog + 2, __ATOMIC_RELAXED)) atomicMax(rlog + 2, mx);
        }
    }
}

// P16 planes -> f32 [B][C][L]: (hi + mid) / 32 (exact sum: both are fp16 values whose exponents are at most 11 apart)
__global__ __launch_bounds__(256) void p16_to_f32_kernel(const uint4 *__restrict__ p, float *__restrict__ x, int C, int L, int KC)
{
    const int l = blockIdx.x * 256 + threadIdx.x;
    const int kc = blockIdx.y >> 1, hh = blockIdx.y & 1, b = blockIdx.z;
    if (l >= L) return;
    const uint4 *pb = p + (size_t)b * KC * 4 * L;
    const uint4 hv = pb[((size_t)(kc * 2 + 0) * 2 + hh) * L + l], mv = pb[((size_t)(kc * 2 + 1) * 2 + hh) * L + l];
    const f16x8 h8 = __builtin_bit_cast(f16x8, hv), m8 = __builtin_bit_cast(f16x8, mv);
    float *xb = x + (size_t)b * C * L + l;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const int c = kc * 16 + p16_channel(hh, e);
        if (c < C) xb[(size_t)c * L] = ((float)h8[e] + (float)m8[e]) * 0.03125f;
    }
}

// ---------------------------------------------------------------------------------------------------------------------------
struct H3pArgs {
    const void *x1, *x2;                  // P16 planes: x1 [B][KC1][2][2][L1][16 B], x2 [B][KC2][2][2][L][16 B] (or NULL)
    const void *Wp;                       // h3p pack of W [Cout][16 (KC1 + KC2) columns in concat order]
    const float *scale, *shift;
    float *y;                             // optional f32 output [B][Cout][L]
    void *yp;                             // optional P16 output [B][Cout/16][2][2][L][16 B]
    const int32_t *gidx;                  // optional [B][L]: column l of x1 is x1[..][gidx[b][l]] (out of range: zeros)
    const float *zadd;                    // optional per-node addend [B][Cout][ZM] (segmenter layer 1), with zidx [B][L]
    const int32_t *zidx;
    unsigned *rlog;                       // range-log slot (word 1: max |w| bits, word 2: max of the P16 output)
    double *stats_partial;                // optional [ncol][Cout][2] (sum, sum of squares of the f32 output over the workgroup's columns)
    int abl;                              // (variants build) ablations: 1 = outputs dropped (stores issued, out of range), 2 = every X load reads chunk 0, 4 = every W request reads chunk 0, 8 = no epilogue at all
    unsigned long long *prof;             // (variants build) phase cycle counters: [0] workgroups, [1] prologue, [2] chunk loops, [3] epilogues, [4] whole
    long long ngroups;                    // B * gpc
    int KC1, KC2, L1, L, Cout, CT, KC, KCr, KCP, ct_per_y, nslab, ncol, gpc, relu, ZM;
    // (EPI 4 / 5: group-max epilogues of the node-level stage) every workgroup's 128 columns hold G groups of GK consecutive columns
    // (columns G GK .. 127 are padding); the max over a group goes to output column wg_col G + g (< ngout) of yp (P16, Lout = ngout
    // columns; EPI 4) or to y[group][Cout] (f32; EPI 5)
    int GK, G, ngout, Lout;                // (Lout >= ngout: column count = plane stride of yp)
};

__device__ __forceinline__ i32x4_t make_rsrc(const void *base, unsigned bytes) {
    const unsigned long long a = reinterpret_cast<unsigned long long>(base);
    i32x4_t r;
    r[0] = __builtin_amdgcn_readfirstlane((int)(unsigned)a);
    r[1] = __builtin_amdgcn_readfirstlane((int)(unsigned)((a >> 32) & 0xFFFFu));      // stride 0: raw buffer
    r[2] = __builtin_amdgcn_readfirstlane((int)bytes);
    r[3] = 0x00020000;
    return r;
}
__device__ __forceinline__ u32x4 bload16(i32x4_t rsrc, unsigned voff, unsigned soff) {
    u32x4 r;
    asm volatile("buffer_load_dwordx4 %0, %1, %2, %3 offen" : "=v"(r) : "v"(voff), "s"(rsrc), "s"(soff) : "memory");
    return r;
}
// Stores go through the compiler's builtins: nothing about a store needs hiding from hipcc (it never waits for one), and an inline-asm
// buffer_store_dwordx4 needs wait states before its data registers may be rewritten that hipcc does not add behind an asm statement
// (the first version of the transposed epilogue lost the last two values of a quad that way).
__device__ __forceinline__ void bstore16(__amdgpu_buffer_rsrc_t rsrc, unsigned voff, unsigned soff, u32x4 v) {
    __builtin_amdgcn_raw_buffer_store_b128(v, rsrc, voff, soff, 0);
}
__device__ __forceinline__ void bstore4(__amdgpu_buffer_rsrc_t rsrc, unsigned voff, unsigned soff, float v) {
    __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), rsrc, voff, soff, 0);
}
// "at most N vector-memory operations of this wave are outstanding", then the workgroup barrier; the registers the wait is FOR are
// in/out operands, so that every later use depends on this statement
template <int N>
__device__ __forceinline__ void wait_barrier(u32x4 &a, u32x4 &b) {
    asm volatile("s_waitcnt vmcnt(%2)\n\ts_barrier" : "+v"(a), "+v"(b) : "n"(N) : "memory");
}
template <int N>
__device__ __forceinline__ void wait_barrier(u32x4 &a, u32x4 &b, u32x4 &c, u32x4 &d) {
    asm volatile("s_waitcnt vmcnt(%4)\n\ts_barrier" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "n"(N) : "memory");
}

// scheduling pattern of one tile pair: a fragment read (for the NEXT pair), then a quarter of this pair's MFMAs, four times
template <int NC>
__device__ __forceinline__ void sched_pair() {
    __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
    __builtin_amdgcn_sched_group_barrier(0x008, NC == 2 ? 3 : 1, 0);
    __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
    __builtin_amdgcn_sched_group_barrier(0x008, NC == 2 ? 3 : 2, 0);
    __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
    __builtin_amdgcn_sched_group_barrier(0x008, NC == 2 ? 3 : 1, 0);
    __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
    __builtin_amdgcn_sched_group_barrier(0x008, NC == 2 ? 3 : 2, 0);
}

// MT output tiles per pass (even), NC 32-column tiles per wave, OCC workgroups per CU the register budget is cut for.
// EPI: 0 = plain, 1 = per-node addend (zadd) staged through LDS, 3 = per-node addend gathered from global memory (any ZM, any column
// grouping: the fallback), 2 = statistics (training forward).  OUT: bit 0 = f32 output y, bit 1 = P16 output yp.
// (A transposed orientation -- X as the MFMA's A operand, so that a lane holds 16 points of one channel and the f32 rows leave as 16-byte
// stores -- was built and dropped: its stores write 32-byte pieces of 32 rows per instruction and measured 12 % slower than two full
// 128-byte lines per dword store, docs/findings.md R4.2.)
// EPI 4 / 5 (NC = 1, one cloud = the flat column axis of the node-level stage): the max over groups of GK consecutive columns instead of
// the columns themselves -- KNNModule's max over the K neighbours of a node (models/layers.py:365; EPI 4: P16 planes out) and the global
// max over a cloud's M nodes (models/networks.py:197; EPI 5: f32 [cloud][Cout]).  A tile's 32 x 128 post-activation values pass through
// LDS (17 KiB) one tile at a time; NaN wins like torch.max.
constexpr int GM_STRIDE = 136;                                 // floats per staged row: 128 columns + 8 (rows 4 apart land 32 banks apart)
__device__ __forceinline__ float nan_max(float m, float v) { return (v > m || v != v) ? v : m; }
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

template <int MT, int NC, int OCC, int EPI, int OUT>
__global__ __launch_bounds__(P_THREADS, OCC) void pointmlp_h3p_kernel(const H3pArgs a)
{
    static_assert(OUT >= 1 && OUT <= 3 && (EPI != 2 || (OUT & 1)), "output mode");
    static_assert(EPI < 4 || (NC == 1 && ((EPI == 4 && OUT == 2) || (EPI == 5 && OUT == 1))), "group-max epilogues: 32-column waves, P16 (4) or f32 (5) out");
    static_assert(MT % 2 == 0 && (NC == 1 || NC == 2), "tile shape");
    // X look-ahead: 3 chunks (2 measured 3-5 % slower on the 64-column tiles); 2 for the 64-column tiles with the addend rows in LDS at two
    // workgroups per CU: 16 registers less, that instantiation sits at the 256-register limit
    constexpr int PB = (OCC >= 2 && NC == 2 && (EPI == 1 || EPI == 3)) ? 2 : 3, NB = 4, D = 3, NSLOT = D + 2;
    static_assert(D >= PB && PB >= 2, "the wait count below assumes the W request of a chunk is older than its X request");
    constexpr int NSL = MT * 2;                                // 1 KiB slices per chunk: [tile][form]
    constexpr int ND = NSL / 4;                                // LDS-DMA instructions per wave and chunk
    constexpr int NBL = NC * 2;                                // X loads per chunk
    constexpr int NP = MT / 2;                                 // tile pairs per chunk
    constexpr int KWAIT = (PB - 1) * (NBL + ND);
    static_assert(KWAIT < 64, "vmcnt is a 6-bit counter");
    struct Lds {
        u32x4 wsm[NSLOT][NSL][64];                            // W ring first: LDS-DMA addresses below 64 KiB
        // scale / 1024 and shift of the slab's rows (EPI 1: slabs of <= 16 tiles, the addend rows need the space) as [tile][h][scale, shift][16]:
        // the 16 rows of a tile a lane's registers hold (row (r & 3) + 8 (r >> 2) + 4 h in register r), contiguous, so that a tile's 32 values
        // are eight 16-byte reads at the top of the tile instead of 32 reads inside it
        float aff[EPI == 1 ? 1024 : 2048];
        float2 red[EPI == 2 ? 4 : 1][EPI == 2 ? MT * 32 : 1];
        float zl[EPI == 1 ? MT * 32 : 1][EPI == 1 ? 64 : 1];      // (EPI 1) the pass's rows of the per-node addend, when the workgroup sits in one cloud
        float gm[EPI >= 4 ? 32 : 1][EPI >= 4 ? GM_STRIDE : 1];    // (EPI 4 / 5) one output tile x the workgroup's 128 columns, post-activation
    };
    __shared__ __attribute__((aligned(16))) Lds lds;

    // workgroup -> (column group, output slab), XCD-aware: the nslab workgroups that read the same columns get consecutive slots of one XCD
    const int wg_xcd = blockIdx.x & 7, wg_local = blockIdx.x >> 3;
    const int wg_col = (wg_local / a.nslab) * 8 + wg_xcd, wg_slab = wg_local - (wg_local / a.nslab) * a.nslab;
    if (wg_col >= a.ncol) return;

    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int j = lane & 31, h = lane >> 5;
    long long q = (long long)wg_col * 4 + wave;
    const bool wave_valid = q < a.ngroups;
    q = wave_valid ? q : 0;
    const int b = __builtin_amdgcn_readfirstlane((int)(q / a.gpc));
    const int l0 = __builtin_amdgcn_readfirstlane((int)(q - (long long)b * a.gpc) * (32 * NC));
    const int L = a.L, L1 = a.L1;

    bool pv[NC];
    unsigned vo1[NC], vo2[NC], voy[NC], voyp[NC];
    int lcl[NC];
#pragma unroll
    for (int c = 0; c < NC; ++c) {
        const int l = l0 + 32 * c + j;
        pv[c] = wave_valid && l < L;
        lcl[c] = l < L ? l : l0;
        vo2[c] = (unsigned)(h * L + lcl[c]) * 16u;
        vo1[c] = (unsigned)(h * L1 + lcl[c]) * 16u;
        if (a.gidx) {
            const int src = a.gidx[(size_t)b * L + lcl[c]];
            vo1[c] = (unsigned)src < (unsigned)L1 ? (unsigned)(h * L1 + src) * 16u : OOB;
        }
        voy[c] = pv[c] ? (unsigned)(4 * h * L + l) * 4u : OOB;
        voyp[c] = pv[c] ? (unsigned)(h * L + l) * 16u : OOB;
#ifdef SONET_VARIANTS
        if (a.abl & 1) { voy[c] = OOB; voyp[c] = OOB; pv[c] = false; }
#endif
    }
    const i32x4_t r1 = make_rsrc(static_cast<const char *>(a.x1) + (size_t)b * a.KC1 * 64 * L1, (unsigned)a.KC1 * 64u * (unsigned)L1);
    const i32x4_t r2 = make_rsrc(a.x2 ? static_cast<const char *>(a.x2) + (size_t)b * a.KC2 * 64 * L : static_cast<const char *>(a.x1),
                                 a.x2 ? (unsigned)a.KC2 * 64u * (unsigned)L : 0u);
    const __amdgpu_buffer_rsrc_t ry = __builtin_amdgcn_make_buffer_rsrc(
        a.y ? reinterpret_cast<char *>(a.y) + (size_t)b * a.Cout * L * 4 : nullptr, 0, a.y ? (int)((unsigned)a.Cout * (unsigned)L * 4u) : 0, 0x00020000);
    const __amdgpu_buffer_rsrc_t ryp = __builtin_amdgcn_make_buffer_rsrc(
        a.yp ? static_cast<char *>(a.yp) + (size_t)b * (a.Cout / 16) * 64 * L : nullptr, 0, a.yp ? (int)((unsigned)(a.Cout / 16) * 64u * (unsigned)L) : 0, 0x00020000);

    const int ct_begin = wg_slab * a.ct_per_y;
    const int ct_end = min(a.CT, ct_begin + a.ct_per_y);
    const int npass = (ct_end - ct_begin) / MT;                // (the host makes ct_per_y a multiple of MT)
    const int KCr = a.KCr, KC1 = a.KC1;
#ifdef SONET_VARIANTS
    unsigned long long pt0 = __builtin_readcyclecounter(), pt_loop = 0, pt_epi = 0, pt_pro = 0;
#endif
    const unsigned wsm_lds = (unsigned)reinterpret_cast<size_t>(&lds.wsm[0][0][0]);
    const unsigned vow = (unsigned)lane * 16u;
    const char *wp = static_cast<const char *>(a.Wp);

    // X chunk kc (of this wave's columns) -> registers: [c][form]
    auto load_b = [&](u32x4 (&bb)[NBL], int kc) {
#ifdef SONET_VARIANTS
        if (a.abl & 2) kc = 0;
#endif
        const bool second = kc >= KC1;
        const int kk = second ? kc - KC1 : kc;
        const unsigned rowb = (unsigned)(second ? L : L1) * 32u;             // bytes of one (chunk, form) plane pair
        const i32x4_t rs = second ? r2 : r1;
#pragma unroll
        for (int c = 0; c < NC; ++c) {
            const unsigned vo = second ? vo2[c] : vo1[c];
            bb[2 * c + 0] = bload16(rs, vo, (unsigned)(kk * 2 + 0) * rowb);
            bb[2 * c + 1] = bload16(rs, vo, (unsigned)(kk * 2 + 1) * rowb);
        }
    };
    // W chunk (pass, kc) -> ring slot: this wave moves slices wave, wave + 4, ...; slice sl = tile sl / 2, form sl % 2
    auto dma = [&](int pass, int kc, int slot) {
#ifdef SONET_VARIANTS
        if (a.abl & 4) { kc = 0; pass = 0; }
#endif
        const int ps = pass < npass ? pass : npass - 1;
        const char *g0 = wp + ((size_t)(ct_begin + ps * MT) * a.KCP + kc) * 2048u;
        const unsigned d0 = wsm_lds + (unsigned)slot * (unsigned)(NSL * 1024);
#pragma unroll
        for (int t = 0; t < ND; ++t) {
            const int sl = wave + 4 * t;
            const char *g = g0 + (size_t)(sl >> 1) * ((size_t)a.KCP * 2048u) + (size_t)(sl & 1) * 1024u;
            const unsigned d = d0 + (unsigned)sl * 1024u;
            unsigned keep;
            // (s_nop 4: an SGPR operand may have been written by a VALU instruction -- v_readfirstlane -- and a VMEM instruction
            // reading it needs 5 wait states that hipcc does not add inside inline asm)
            asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 4\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
                         : "=&s"(keep) : "v"(vow), "s"(g), "s"(d) : "memory");
        }
    };
    auto read_a = [&](u32x4 (&A)[4], int slot, int p) {        // tile pair p of the chunk in `slot`: [tile 2p: hi, mid][tile 2p+1: hi, mid]
#pragma unroll
        for (int u = 0; u < 4; ++u) A[u] = lds.wsm[slot][4 * p + u][lane];
    };

    f32x16 acc[MT][NC];

    auto mfma_pair = [&](const u32x4 (&A)[4], const u32x4 (&bb)[NBL], int p) {
        f16x8 Bh[NC], Bm[NC];
#pragma unroll
        for (int c = 0; c < NC; ++c) {
            Bh[c] = __builtin_bit_cast(f16x8, bb[2 * c + 0]);
            Bm[c] = __builtin_bit_cast(f16x8, bb[2 * c + 1]);
        }
#pragma unroll
        for (int term = 0; term < 3; ++term)
#pragma unroll
            for (int tt = 0; tt < 2; ++tt)
#pragma unroll
                for (int c = 0; c < NC; ++c)
                {
                    const f16x8 wf = __builtin_bit_cast(f16x8, A[2 * tt + (term == 0 ? 1 : 0)]), xf = term == 1 ? Bm[c] : Bh[c];
                    acc[2 * p + tt][c] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wf, xf, acc[2 * p + tt][c], 0, 0, 0);
                }
    };

    RangeAcc yr = {0, 0u};
    const float relu_thr = a.relu ? 0.f : -__builtin_inff();   // v < thr ? 0 : v  (a NaN stays a NaN, as torch's ReLU leaves it)
    const float split_lo = a.relu ? 0.f : -2047.f;             // lower clamp of the split: ReLU and clamp are one v_med3_f32
    // (EPI 1 / 3) the node of each of this lane's columns: loaded once per workgroup (a compiler-visible load inside the pass loop would make
    // hipcc wait for vmcnt(0) there: the look-ahead requests included)
    int zm[NC];
    bool zok[NC];
    unsigned zvo[NC], zla[NC];
    const i32x4_t rz = make_rsrc(a.zadd ? a.zadd + (size_t)b * a.Cout * a.ZM : nullptr, a.zadd ? (unsigned)a.Cout * (unsigned)a.ZM * 4u : 0u);
    if constexpr (EPI == 1 || EPI == 3) {
#pragma unroll
        for (int c = 0; c < NC; ++c) {
            zm[c] = a.zidx[(size_t)b * L + lcl[c]];
            zok[c] = (unsigned)zm[c] < (unsigned)a.ZM;
            zm[c] = zok[c] ? zm[c] : 0;
            zvo[c] = zok[c] ? (unsigned)(zm[c] + 4 * h * a.ZM) * 4u : OOB;      // (the gather path: row 4 h, node zm; out of range: 0 through the descriptor)
            // (the LDS path: ONE opaque address register per column tile and row offsets in the instruction -- the rows sit beyond the
            // 64 KiB an LDS instruction's offset field reaches from 0, and hipcc otherwise keeps a hoisted address register per row)
            zla[c] = (unsigned)reinterpret_cast<size_t>(&lds.zl[4 * h][zm[c]]);
            asm volatile("" : "+v"(zla[c]));
        }
    }
    // epilogue of one pass: tiles ct0 .. ct0 + MT - 1, one (tile, column tile) at a time (a scheduling barrier in between: left alone,
    // hipcc lifts all 16 MT NC accumulator registers out of the accumulation file at once and spills)
    auto epilogue = [&](int ct0) {
        if constexpr (EPI >= 4) {
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) {
                if ((ct0 + mt) * 32 >= a.Cout) continue;          // (uniform over the workgroup: the barriers below are safe)
                float asc16[16], ash16[16];
                {
                    const float4 *t4 = reinterpret_cast<const float4 *>(lds.aff + (ct0 - ct_begin + mt) * 64 + h * 32);
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        const float4 s4 = t4[u], h4 = t4[4 + u];
                        asc16[4 * u] = s4.x; asc16[4 * u + 1] = s4.y; asc16[4 * u + 2] = s4.z; asc16[4 * u + 3] = s4.w;
                        ash16[4 * u] = h4.x; ash16[4 * u + 1] = h4.y; ash16[4 * u + 2] = h4.z; ash16[4 * u + 3] = h4.w;
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const float v = __fmaf_rn(acc[mt][0][r], asc16[r], ash16[r]);
                    lds.gm[(r & 3) + 8 * (r >> 2) + 4 * h][32 * wave + j] = v < relu_thr ? 0.f : v;
                }
                lds_barrier();
                const int t = (int)threadIdx.x;
                if constexpr (EPI == 4) {
                    // thread = (element pair p of a P16 half, group g, 16-row chunk qq and half h2 of the tile): two rows x GK columns
                    const int p = t & 3, u = t >> 2;
                    const int g = u % a.G, qh = u / a.G;
                    if (qh < 4) {
                        const int qq = qh >> 1, h2 = qh & 1;
                        const float *r0 = &lds.gm[16 * qq + 4 * h2 + 2 * (p & 1) + 8 * (p >> 1)][g * a.GK];
                        float m0 = r0[0], m1 = r0[GM_STRIDE];
                        for (int k = 1; k < a.GK; ++k) {
                            m0 = nan_max(m0, r0[k]);
                            m1 = nan_max(m1, r0[GM_STRIDE + k]);
                        }
                        const long long n_out = (long long)wg_col * a.G + g;
                        if (n_out < a.ngout) {
                            range_track(yr, m0, m1);
                            unsigned hv, mv;
                            // (OUT == 2: the affine table made the values 32 x already, like the plain P16 epilogue)
                            split2(__builtin_amdgcn_fmed3f(m0, split_lo * 32.f, 65504.f), __builtin_amdgcn_fmed3f(m1, split_lo * 32.f, 65504.f), hv, mv);
                            unsigned *dst = reinterpret_cast<unsigned *>(static_cast<char *>(a.yp)
                                + ((((size_t)((ct0 + mt) * 2 + qq) * 2) * 2 + h2) * (size_t)a.Lout + (size_t)n_out) * 16) + p;
                            dst[0] = hv;
                            dst[(size_t)a.Lout * 8] = mv;                  // form 1: two half planes (2 x Lout x 16 bytes) further
                        }
                    }
                } else {
                    // thread = (quarter of the group's columns, row, group): GK / 4 values, then the four quarters meet in a quad
                    const int seg = t & 3, row = (t >> 2) & 31, g = t >> 7;
                    const int cols = a.GK >> 2;
                    const float *r0 = &lds.gm[row][(g < a.G ? g : 0) * a.GK + seg * cols];
                    float m = r0[0];
                    for (int k = 1; k < cols; ++k) m = nan_max(m, r0[k]);
                    m = nan_max(m, __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, m), 0xB1, 0xF, 0xF, true)));
                    m = nan_max(m, __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, m), 0x4E, 0xF, 0xF, true)));
                    const long long n_out = (long long)wg_col * a.G + g;
                    if (seg == 0 && g < a.G && n_out < a.ngout) a.y[(size_t)n_out * a.Cout + (size_t)(ct0 + mt) * 32 + row] = m;
                }
                lds_barrier();
            }
            return;
        }
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
            if ((ct0 + mt) * 32 >= a.Cout) continue;              // (the zero tile behind an odd tile count)
            float asc16[16], ash16[16];
            {
                const float4 *t4 = reinterpret_cast<const float4 *>(lds.aff + (ct0 - ct_begin + mt) * 64 + h * 32);
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const float4 s4 = t4[u], h4 = t4[4 + u];
                    asc16[4 * u] = s4.x; asc16[4 * u + 1] = s4.y; asc16[4 * u + 2] = s4.z; asc16[4 * u + 3] = s4.w;
                    ash16[4 * u] = h4.x; ash16[4 * u + 1] = h4.y; ash16[4 * u + 2] = h4.z; ash16[4 * u + 3] = h4.w;
                }
            }
            __builtin_amdgcn_sched_barrier(0);
            float st1[EPI == 2 ? 16 : 1], st2[EPI == 2 ? 16 : 1];          // (EPI 2) per register: this lane's sum / sum of squares over its column tiles
#pragma unroll
            for (int c = 0; c < NC; ++c) {
#pragma unroll
                for (int qq = 0; qq < 2; ++qq) {               // registers 8 qq .. 8 qq + 7: rows 16 qq + {0..3, 8..11} + 4 h of the tile
                    unsigned hh[4], mm[4];
#pragma unroll
                    for (int hf = 0; hf < 2; ++hf) {           // four registers at a time (a scheduling barrier after each: register pressure)
                        const int r0 = 8 * qq + 4 * hf;         // registers r0 .. r0 + 3: rows 16 qq + 8 hf + (0..3) + 4 h
                        const int row0 = mt * 32 + 16 * qq + 8 * hf;
                        float av[4], v[4], zg[4];
                        // the accumulators leave the accumulation file HERE (pinned: left to itself hipcc copies the whole file into VGPRs at the
                        // top of the epilogue, 16 MT NC registers, and spills)
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            if constexpr (OCC == 1) asm volatile("v_accvgpr_read_b32 %0, %1" : "=v"(av[e]) : "a"(acc[mt][c][r0 + e]));
                            else av[e] = acc[mt][c][r0 + e];
                        }
                        if constexpr (EPI == 3) {
                            // (fallback: gathers waited for at once -- inline asm like every vector-memory access of this kernel, so that hipcc
                            // neither hoists 16 MT NC address pairs out of the pass loop nor miscounts the outstanding requests)
#pragma unroll
                            for (int e = 0; e < 4; ++e) {
                                const unsigned so = (unsigned)((ct0 + mt) * 32 + 16 * qq + 8 * hf + e) * (unsigned)a.ZM * 4u;
                                asm volatile("buffer_load_dword %0, %1, %2, %3 offen" : "=v"(zg[e]) : "v"(zvo[c]), "s"(rz), "s"(so) : "memory");
                            }
                            asm volatile("s_waitcnt vmcnt(0)" : "+v"(zg[0]), "+v"(zg[1]), "+v"(zg[2]), "+v"(zg[3]) :: "memory");
                        }
                        if constexpr (EPI == 1) {
#pragma unroll
                            for (int e = 0; e < 4; ++e)
                                asm volatile("v_add_u32 %0, %2, %1\n\tds_read_b32 %0, %0" : "=&v"(zg[e]) : "v"(zla[c]), "s"((row0 + e) * 256) : "memory");
                            asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(zg[0]), "+v"(zg[1]), "+v"(zg[2]), "+v"(zg[3]) :: "memory");
                        }
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            const float sc = asc16[r0 + e], sf = ash16[r0 + e];
                            if constexpr (EPI == 1 || EPI == 3) {
                                // per-node addend: the layer's input concatenates per-column channels (the GEMM above) with channels that are
                                // constant per node -- their block of W . x is computed once per node by another launch and added here
                                // (segmenter layer 1, models/networks.py:296-326)
                                const float zv = EPI == 1 ? (zok[c] ? zg[e] : 0.f) : zg[e];
                                v[e] = __fmaf_rn(av[e], sc, __fmaf_rn(zv, sc * 1024.f, sf));
                            } else {
                                v[e] = __fmaf_rn(av[e], sc, sf);
                            }
                        }
                        if constexpr ((OUT & 1) != 0) {
#pragma unroll
                            for (int e = 0; e < 4; ++e) {
                                const int orow = 16 * qq + 8 * hf + e;
                                const float o = v[e] < relu_thr ? 0.f : v[e];
                                bstore4(ry, voy[c], (unsigned)((ct0 + mt) * 32 + orow) * (unsigned)L * 4u, o);
                                if constexpr (EPI == 2) {
                                    // Training forward: BatchNorm's batch statistics of the output (models/layers.py:60-70) from this epilogue: a lane
                                    // first sums its column tiles (a padded column adds 0), the 32 lanes of a row meet once per row below
                                    const float vv = pv[c] ? o : 0.f;
                                    st1[r0 + e] = c == 0 ? vv : st1[r0 + e] + vv;
                                    st2[r0 + e] = c == 0 ? vv * vv : st2[r0 + e] + vv * vv;
                                }
                            }
                        }
                        if constexpr ((OUT & 2) != 0) {
#pragma unroll
                            for (int p = 0; p < 2; ++p) {
                                const float x0 = v[2 * p], x1 = v[2 * p + 1];
                                range_track(yr, x0, x1);
                                if constexpr (OUT == 2)             // (the values are 32 x already: see the affine table)
                                    split2(__builtin_amdgcn_fmed3f(x0, split_lo * 32.f, 65504.f), __builtin_amdgcn_fmed3f(x1, split_lo * 32.f, 65504.f),
                                           hh[2 * hf + p], mm[2 * hf + p]);
                                else
                                    split2(32.f * __builtin_amdgcn_fmed3f(x0, split_lo, 2047.f), 32.f * __builtin_amdgcn_fmed3f(x1, split_lo, 2047.f),
                                           hh[2 * hf + p], mm[2 * hf + p]);
                            }
                        }
                        if constexpr (EPI == 1 || EPI == 3 || OUT == 3) __builtin_amdgcn_sched_barrier(0);      // (register pressure; the plain epilogues run 8 values per fence)
                    }
                    if constexpr ((OUT & 2) == 0) __builtin_amdgcn_sched_barrier(0);
                    if constexpr ((OUT & 2) != 0) {
                        const unsigned so = (unsigned)(((ct0 + mt) * 2 + qq) * 2) * (unsigned)L * 32u;
                        const u32x4 hv = {hh[0], hh[1], hh[2], hh[3]}, mv = {mm[0], mm[1], mm[2], mm[3]};
                        bstore16(ryp, voyp[c], so, hv);
                        bstore16(ryp, voyp[c], so + (unsigned)L * 32u, mv);
                        __builtin_amdgcn_sched_barrier(0);
                    }
                }
            }
            if constexpr (EPI == 2) {
                // a row's 32 columns sit in the 32 lanes of a half wave: four DPP adds + one swizzle per quantity, all lanes active
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const float s1 = row32_sum(st1[r]), s2 = row32_sum(st2[r]);
                    if (j == 0) lds.red[wave][mt * 32 + (r & 3) + 8 * (r >> 2) + 4 * h] = make_float2(s1, s2);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        if constexpr (EPI == 2) {
            __syncthreads();
            for (int t = threadIdx.x; t < MT * 32 && ct0 * 32 + t < a.Cout; t += P_THREADS) {
                const double s = ((double)lds.red[0][t].x + (double)lds.red[1][t].x) + ((double)lds.red[2][t].x + (double)lds.red[3][t].x);
                const double qq = ((double)lds.red[0][t].y + (double)lds.red[1][t].y) + ((double)lds.red[2][t].y + (double)lds.red[3][t].y);
                double *dst = a.stats_partial + ((size_t)wg_col * a.Cout + (size_t)ct0 * 32 + t) * 2;
                dst[0] = s;
                dst[1] = qq;
            }
            __syncthreads();
        }
        if constexpr (EPI == 1) __syncthreads();               // every wave is done with the addend rows before the next pass's rows land on them
    };
    // (EPI 1) the pass's MT x 32 rows of the addend, 256 bytes each and contiguous in z[b]: into LDS by LDS-DMA at the start of the pass.
    // The requests are older than every X / W request of the pass's later chunks, so the per-chunk waits cover them; the per-chunk barriers
    // publish them.  Needs the four waves in one cloud and ZM == 64 (a.zlds); otherwise the epilogue gathers from global memory.
    auto z_dma = [&](int ct0) {
        if constexpr (EPI == 1) {
            {
                const char *g0 = reinterpret_cast<const char *>(a.zadd + ((size_t)b * a.Cout + (size_t)ct0 * 32) * 64);
                const unsigned d0 = (unsigned)reinterpret_cast<size_t>(&lds.zl[0][0]);
#pragma unroll
                for (int t = 0; t < MT * 2; ++t) {
                    const int piece = wave + 4 * t;                     // 1 KiB = 4 rows per piece
                    const char *g = g0 + (size_t)piece * 1024u;
                    const unsigned d = d0 + (unsigned)piece * 1024u;
                    unsigned keep;
                    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 4\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
                                 : "=&s"(keep) : "v"(vow), "s"(g), "s"(d) : "memory");
                }
            }
        }
    };

    // ---- the flat loop over (pass, chunk) -------------------------------------------------------------------------------------
    u32x4 bq[NB][NBL];
    u32x4 Aq[2][4];
    // prologue: the issue order of the steady state (DMA of a chunk BEFORE the X loads issued in the same body), so that the wait count
    // below means the same thing at chunk 0 as everywhere else
    // (KCr is a multiple of 4 >= D + 1, so chunks 0 .. D belong to pass 0)
#pragma unroll
    for (int i = 0; i <= D - PB; ++i) dma(0, i, i);
#pragma unroll
    for (int t = -PB; t <= -2; ++t) {                          // the bodies -PB .. -2 of the steady state, without their MFMAs
        load_b(bq[(t + PB) & 3], t + PB);
        dma(0, t + 1 + D, t + 1 + D);
    }
    load_b(bq[(PB - 1) & 3], PB - 1);                          // body -1
    if constexpr (NBL == 2) wait_barrier<KWAIT>(bq[0][0], bq[0][1]);
    else wait_barrier<KWAIT>(bq[0][0], bq[0][1], bq[0][2], bq[0][3]);
    dma(0, D, D);
    read_a(Aq[0], 0, 0);
    // scale / shift of the slab into LDS -- AFTER the prologue's requests have been issued: the two loads per row fly under them.
    // OUT == 2 (P16 planes only): the table holds scale / 32 and 32 shift, so that the epilogue's value already is 32 x (the split's
    // scaling; powers of two commute with the rounding of the fma, the planes are bit-identical to scaling afterwards).
    for (int o = ct_begin * 32 + (int)threadIdx.x; o < ct_end * 32; o += P_THREADS)
    {
        const int rel = o - ct_begin * 32, rr = rel & 31;
        constexpr float pre = OUT == 2 ? 32.f : 1.f;
        const float scv = o < a.Cout ? a.scale[o] * (pre / 1024.f) : 0.f, shv = o < a.Cout ? a.shift[o] * pre : 0.f;   // accumulators hold 1024 W.x
        float *t = lds.aff + (rel >> 5) * 64 + ((rr >> 2) & 1) * 32 + ((rr & 3) | ((rr >> 3) << 2));
        t[0] = scv;
        t[16] = shv;
    }
    __syncthreads();
#ifdef SONET_VARIANTS
    pt_pro = __builtin_readcyclecounter() - pt0;
#endif

    int kcb = PB;                                              // chunk of the X prefetch of body t (t + PB, wrapped)
    int kcd = (1 + D) % KCr, passd = (1 + D) / KCr;            // (pass, chunk) of the DMA of body t (t + 1 + D)
    int slot = 0;                                              // ring slot of chunk t
    if (kcb >= KCr) kcb -= KCr;

#define H3P_BODY(I)                                                                                                   \
    {                                                                                                                 \
        load_b(bq[((I) + PB) & 3], kcb);                                                                              \
        const int slot1 = slot + 1 < NSLOT ? slot + 1 : 0;                                                            \
        int slotd = slot + 1 + D;                                                                                     \
        slotd = slotd >= NSLOT ? slotd - NSLOT : slotd;                                                               \
        _Pragma("unroll") for (int p = 0; p < NP; ++p) {                                                              \
            constexpr int base_par = ((I) * NP) & 1;                                                                  \
            const int cur = (base_par + p) & 1;                                                                       \
            if (p == NP - 1) {                                                                                        \
                if constexpr (NBL == 2) wait_barrier<KWAIT>(bq[((I) + 1) & 3][0], bq[((I) + 1) & 3][1]);              \
                else wait_barrier<KWAIT>(bq[((I) + 1) & 3][0], bq[((I) + 1) & 3][1], bq[((I) + 1) & 3][2], bq[((I) + 1) & 3][3]); \
                dma(passd, kcd, slotd);                                                                               \
                read_a(Aq[cur ^ 1], slot1, 0);                                                                        \
            } else {                                                                                                  \
                read_a(Aq[cur ^ 1], slot, p + 1);                                                                     \
            }                                                                                                         \
            mfma_pair(Aq[cur], bq[(I)], p);                                                                           \
            /* one fragment read (for the NEXT pair) per 3 NC / 2 MFMAs of this pair: left alone, hipcc sinks each read next to its first use */ \
            sched_pair<NC>();                                                                                         \
        }                                                                                                             \
        slot = slot1;                                                                                                 \
        kcb = kcb + 1 < KCr ? kcb + 1 : 0;                                                                            \
        if (kcd + 1 < KCr) kcd += 1; else { kcd = 0; passd += 1; }                                                    \
    }

    // (pass loop outside, chunk loop inside: the look-ahead state -- kcb, kcd, passd, slot, the register rings -- runs through; with ONE flat
    // loop and the epilogue under a condition hipcc merges "old" and "cleared" accumulators at the join and copies the whole accumulation
    // file into VGPRs)
    for (int pass = 0; pass < npass; ++pass) {
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int c = 0; c < NC; ++c)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[mt][c][r] = 0.f;
        z_dma(ct_begin + pass * MT);
#ifdef SONET_VARIANTS
        const unsigned long long pl0 = __builtin_readcyclecounter();
#endif
        for (int k4 = 0; k4 < KCr; k4 += 4) {
            H3P_BODY(0)
            H3P_BODY(1)
            H3P_BODY(2)
            H3P_BODY(3)
        }
#ifdef SONET_VARIANTS
        const unsigned long long pl1 = __builtin_readcyclecounter();
        pt_loop += pl1 - pl0;
#endif
#ifdef SONET_VARIANTS
        if (!(a.abl & 8))
#endif
        epilogue(ct_begin + pass * MT);
#ifdef SONET_VARIANTS
        pt_epi += __builtin_readcyclecounter() - pl1;
#endif
    }
#undef H3P_BODY
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");           // the look-ahead requests of the tail: nothing may land after the workgroup has left
#ifdef SONET_VARIANTS
    if (a.prof != nullptr && threadIdx.x == 0) {
        atomicAdd(a.prof + 0, 1ull);
        atomicAdd(a.prof + 1, pt_pro);
        atomicAdd(a.prof + 2, pt_loop);
        atomicAdd(a.prof + 3, pt_epi);
        atomicAdd(a.prof + 4, (unsigned long long)__builtin_readcyclecounter() - pt0);
    }
#endif

    if (a.rlog != nullptr) {
        // (every output slab publishes the range of the planes IT wrote -- slabs hold different channels; only the weight trailer is
        // a per-launch constant and goes through one thread)
        if (a.yp) {
            // (after a ReLU only the positive side -- and a NaN of either sign -- can leave the range)
            const unsigned pos = yr.mp > 0 ? (unsigned)yr.mp : 0u, nan_neg = yr.mn > 0xFF800000u ? (yr.mn & 0x7FFFFFFFu) : 0u;
            unsigned bits = wave_umax(a.relu ? (pos > nan_neg ? pos : nan_neg) : range_amax_bits(yr));
            if constexpr (OUT == 2) bits = bits > (5u << 23) ? bits - (5u << 23) : 0u;       // (32 x was tracked: take the factor out of the exponent)
            range_publish(a.rlog + 2, bits, lane);
        }
        if (ct_begin == 0 && wg_col == 0 && threadIdx.x == 0)
            atomicMax(a.rlog + 1, reinterpret_cast<const unsigned *>(wp + (size_t)a.CT * a.KCP * 2048u)[0]);
    }
}

}  // namespace

// =========================================================== C ABI ===========================================================

extern "C" size_t sonet_p16_size(int B, int C, int L)
{
    if (B <= 0 || C <= 0 || L <= 0) return 0;
    return (size_t)B * (size_t)sonet::ceil_div(C, 16) * 64 * (size_t)L;
}

extern "C" int sonet_p16_from_f32(const float *x, void *p16, int B, int C, int L, const float *scale, const float *shift, int relu,
                                  sonet_stream_t stream)
{
    const char *what = "sonet_p16_from_f32";
    SONET_REQUIRE(x && p16, "%s: NULL pointer", what);
    SONET_REQUIRE(B > 0 && C > 0 && L > 0, "%s: non-positive size", what);
    SONET_REQUIRE((scale == nullptr) == (shift == nullptr), "%s: scale and shift come together", what);
    const int KC = sonet::ceil_div(C, 16);
    if (B > 65535 || KC * 2 > 65535) return sonet::fail(SONET_ERR_UNSUPPORTED, "%s: B or C too large for one launch", what);
    dim3 grid((unsigned)sonet::ceil_div(L, 1024), (unsigned)(KC * 2), (unsigned)B);
    unsigned *rlog = sonet::range_log();
    if (scale) hipLaunchKernelGGL(p16_from_f32_kernel<true>, grid, dim3(256), 0, sonet::as_stream(stream), x, reinterpret_cast<uint4 *>(p16), C, L, KC, scale, shift, relu, rlog);
    else       hipLaunchKernelGGL(p16_from_f32_kernel<false>, grid, dim3(256), 0, sonet::as_stream(stream), x, reinterpret_cast<uint4 *>(p16), C, L, KC, scale, shift, relu, rlog);
    return sonet::launched(what);
}

extern "C" int sonet_p16_to_f32(const void *p16, float *x, int B, int C, int L, sonet_stream_t stream)
{
    const char *what = "sonet_p16_to_f32";
    SONET_REQUIRE(x && p16, "%s: NULL pointer", what);
    SONET_REQUIRE(B > 0 && C > 0 && L > 0, "%s: non-positive size", what);
    const int KC = sonet::ceil_div(C, 16);
    if (B > 65535 || KC * 2 > 65535) return sonet::fail(SONET_ERR_UNSUPPORTED, "%s: B or C too large for one launch", what);
    dim3 grid((unsigned)sonet::ceil_div(L, 256), (unsigned)(KC * 2), (unsigned)B);
    hipLaunchKernelGGL(p16_to_f32_kernel, grid, dim3(256), 0, sonet::as_stream(stream), reinterpret_cast<const uint4 *>(p16), x, C, L, KC);
    return sonet::launched(what);
}

extern "C" size_t sonet_pointmlp_h3p_pack_size(int Cin, int Cout)
{
    if (Cin <= 0 || Cout <= 0) return 0;
    const size_t kcp = (size_t)sonet::ceil_div(Cin, 16 * P_KPAD) * P_KPAD;
    // (an even number of 32-row tiles -- the kernels take tiles in pairs; the rows past Cout are zeros)
    return (size_t)(sonet::ceil_div(Cout, 64) * 2) * kcp * 2048 + 64;
}

extern "C" int sonet_pointmlp_h3p_pack(const float *W, void *Wp, int Cin, int Cout, sonet_stream_t stream)
{
    const char *what = "sonet_pointmlp_h3p_pack";
    SONET_REQUIRE(W && Wp, "%s: NULL pointer", what);
    SONET_REQUIRE(Cin > 0 && Cout > 0, "%s: non-positive size", what);
    const int KCP = sonet::ceil_div(Cin, 16 * P_KPAD) * P_KPAD;
    const long long total = (long long)(sonet::ceil_div(Cout, 64) * 2) * KCP * 64;
    unsigned *trailer = reinterpret_cast<unsigned *>(reinterpret_cast<uint4 *>(Wp) + total * 2);
    if (sonet::zero_words(trailer, 64, sonet::as_stream(stream)) != 0) return sonet::fail(SONET_ERR_LAUNCH, "%s: clearing the trailer failed", what);
    hipLaunchKernelGGL(h3p_pack_kernel, dim3((unsigned)sonet::ceil_div64(total, 256)), dim3(256), 0, sonet::as_stream(stream),
                       W, reinterpret_cast<uint4 *>(Wp), Cin, Cout, KCP, total, trailer);
    return sonet::launched(what);
}

extern "C" size_t sonet_pointmlp_h3p_stats_ws_size(int B, int Cout, int L)
{
    if (B <= 0 || Cout <= 0 || L <= 0) return 0;
    // the smallest column group any tile shape uses is 4 waves x 32 columns
    return (size_t)sonet::ceil_div64((long long)B * sonet::ceil_div(L, 32), 4) * Cout * 2 * sizeof(double);
}

namespace {
struct Shape { int MT, NC, OCC; };

template <int MT, int NC, int OCC>
int launch_shape(int epi, int out, unsigned nwg, hipStream_t st, const H3pArgs &a)
{
#define H3P_GO(E, O) hipLaunchKernelGGL((pointmlp_h3p_kernel<MT, NC, OCC, E, O>), dim3(nwg), dim3(P_THREADS), 0, st, a)
    if (epi == 2 && out == 1) H3P_GO(2, 1);
    else if (epi == 1 && out == 1) { if constexpr (MT <= 8) H3P_GO(1, 1); else return 1; }     // (ring + addend rows must fit the LDS)
    else if (epi == 1 && out == 2) { if constexpr (MT <= 8) H3P_GO(1, 2); else return 1; }
    else if (epi == 3 && out == 1) H3P_GO(3, 1);
    else if (epi == 3 && out == 2) H3P_GO(3, 2);
    else if (epi == 0 && out == 1) H3P_GO(0, 1);
    else if (epi == 0 && out == 2) H3P_GO(0, 2);
    else if (epi == 0 && out == 3) H3P_GO(0, 3);
    else if (epi == 4 && out == 2) { if constexpr (NC == 1) H3P_GO(4, 2); else return 1; }
    else if (epi == 5 && out == 1) { if constexpr (NC == 1) H3P_GO(5, 1); else return 1; }
    else return 1;
#undef H3P_GO
    return 0;
}
}  // namespace

static int h3p_launch(const char *what, const void *x1p, int C1, int L1, const int32_t *gidx, const void *x2p, int C2, const void *Wp,
                      const float *scale, const float *shift, int relu, float *y, void *yp, int B, int Cout, int L,
                      const float *zadd, const int32_t *zidx, int ZM, void *stats_ws, float *mean, float *var,
                      int GK, int G, int ngout, int Lout, sonet_stream_t stream)
{
    if (!gidx) L1 = L;
    SONET_REQUIRE(x1p && Wp && scale && shift && (y || yp), "%s: NULL pointer", what);
    SONET_REQUIRE(B > 0 && C1 > 0 && C2 >= 0 && Cout > 0 && L > 0 && L1 > 0, "%s: non-positive size", what);
    SONET_REQUIRE((C2 == 0) == (x2p == nullptr), "%s: x2 and C2 disagree", what);
    SONET_REQUIRE(C2 == 0 || C1 % 16 == 0, "%s: with a second input C1=%d must be a multiple of 16", what, C1);
    SONET_REQUIRE((zadd == nullptr) == (zidx == nullptr) && (!zadd || ZM > 0), "%s: zadd, zidx and ZM come together", what);
    SONET_REQUIRE(!stats_ws || (y && !yp && mean && var && !zadd), "%s: the statistics epilogue needs y (only), mean, var and no addend", what);
    if (Cout % 32 != 0) return sonet::fail(SONET_ERR_UNSUPPORTED, "%s: Cout=%d must be a multiple of 32", what, Cout);
    const int KC1 = sonet::ceil_div(C1, 16), KC2 = sonet::ceil_div(C2, 16), KC = KC1 + KC2;
    const int KCP = sonet::ceil_div(C1 + C2, 16 * P_KPAD) * P_KPAD;
    const int KCr = sonet::ceil_div(KC, 4) * 4;
    const int CT = sonet::ceil_div(Cout, 64) * 2;              // tiles incl. the pack's zero tile behind an odd count: nothing of it is stored
    if ((double)KC1 * 64.0 * L1 >= 2.0e9 || (double)KC2 * 64.0 * L >= 2.0e9 || (double)Cout * L * 4.0 >= 2.0e9 || (double)CT * KCP * 2048.0 >= 2.0e9)
        return sonet::fail(SONET_ERR_UNSUPPORTED, "%s: a per-cloud panel exceeds 2 GiB", what);
    const long long cols = (long long)B * L;
    // Tile shape (measured, profiles/r04a_bench_h3p_sweep.log): 4 output tiles per pass and two workgroups per CU everywhere -- 64 columns
    // per wave when the launch still fills the chip that way, 32 otherwise (node-level launches); 2-tile passes for Cout % 128 != 0.
    // (8 x 2 and 6 x 2 tiles with one workgroup per CU measured within 3 % of 4 x 2 on the large layers and 2-4 x slower on the small
    // ones: they exist in the variants build only.)
    Shape sh = (CT % 4 == 0) ? ((cols >= 131072 && GK == 0) ? Shape{4, 2, 2} : Shape{4, 1, 2}) : Shape{2, 1, 2};
#ifdef SONET_VARIANTS
    if (const char *e = sonet::knob("SONET_H3P_SHAPE")) {       // "MT,NC,OCC" (tools/bench_h3p.py)
        int m = 0, n = 0, o = 0;
        if (sscanf(e, "%d,%d,%d", &m, &n, &o) == 3 && CT % m == 0) sh = Shape{m, n, o};
    }
#endif
    const int gpc = sonet::ceil_div(L, 32 * sh.NC);
    const long long ngroups = (long long)B * gpc, ncol = sonet::ceil_div64(ngroups, 4);
    // Output slabs: the workgroups that read the same columns sit next to each other on one XCD, so the input comes from HBM once and from
    // that L2 for the other slabs.  One pass per workgroup on small launches (they need the workgroups); on large ones at most 4 slabs
    // (393 -> 1024 at 64 x 3072 columns: 2 slabs 0.57 ms, 8 slabs 0.60, 1 slab 0.67; 1024 -> 512: 4 slabs 0.53, 1 slab 0.71).
    const int groups = CT / sh.MT;
    int nslab = 1;
    for (int d = 1; d <= groups; ++d)
        if (groups % d == 0 && (d <= 4 || cols < 131072 || CT / nslab > (zadd ? 16 : 32))) nslab = d;
#ifdef SONET_VARIANTS
    if (const char *e = sonet::knob("SONET_H3P_NSLAB")) {
        const int want = atoi(e);
        if (want >= 1 && groups % want == 0 && CT / want <= 32) nslab = want;
    }
#endif
    const long long nwg = sonet::ceil_div64(ncol, 8) * 8 * nslab;
    if (nwg > 0x7FFFFFFFll || ncol > 0x7FFFFFFFll) return sonet::fail(SONET_ERR_UNSUPPORTED, "%s: too many points", what);
    if (stats_ws && sonet::ceil_div64((long long)B * sonet::ceil_div(L, 32), 4) < ncol)
        return sonet::fail(SONET_ERR_UNSUPPORTED, "%s: statistics workspace too small", what);
    H3pArgs a;
    a.x1 = x1p; a.x2 = x2p; a.Wp = Wp; a.scale = scale; a.shift = shift; a.y = y; a.yp = yp; a.gidx = gidx; a.zadd = zadd; a.zidx = zidx;
    a.rlog = sonet::range_log(); a.stats_partial = reinterpret_cast<double *>(stats_ws); a.ngroups = ngroups;
    a.prof = nullptr; a.abl = 0;
#ifdef SONET_VARIANTS
    if (const char *e = sonet::knob("SONET_H3P_ABL")) a.abl = atoi(e);
    if (const char *e = sonet::knob("SONET_H3P_PROF")) a.prof = reinterpret_cast<unsigned long long *>(strtoull(e, nullptr, 0));   // device address of 5 u64 counters
#endif
    a.KC1 = KC1; a.KC2 = KC2; a.L1 = L1; a.L = L; a.Cout = Cout; a.CT = CT; a.KC = KC; a.KCr = KCr; a.KCP = KCP; a.ct_per_y = CT / nslab;
    a.nslab = nslab; a.ncol = (int)ncol; a.gpc = gpc; a.relu = relu; a.ZM = ZM;
    a.GK = GK; a.G = G; a.ngout = ngout; a.Lout = Lout;
    hipStream_t st = sonet::as_stream(stream);
    // (addend rows through LDS: the four waves of a workgroup in one cloud, 256-byte rows, slabs of <= 16 tiles)
    const int epi = GK ? (yp ? 4 : 5)
                       : stats_ws ? 2 : zadd ? ((ZM == 64 && gpc % 4 == 0 && CT / nslab <= 16 && Cout % 64 == 0) ? 1 : 3) : 0;
    const unsigned g = (unsigned)nwg;
    const int out = (y ? 1 : 0) | (yp ? 2 : 0);
    int miss = 1;
    if (sh.MT == 4 && sh.NC == 2) miss = launch_shape<4, 2, 2>(epi, out, g, st, a);
    else if (sh.MT == 4 && sh.NC == 1) miss = launch_shape<4, 1, 2>(epi, out, g, st, a);
    else if (sh.MT == 2 && sh.NC == 1) miss = launch_shape<2, 1, 2>(epi, out, g, st, a);
#ifdef SONET_VARIANTS
    else if (sh.MT == 8 && sh.NC == 2) miss = launch_shape<8, 2, 1>(epi, out, g, st, a);
    else if (sh.MT == 6 && sh.NC == 2) miss = launch_shape<6, 2, 1>(epi, out, g, st, a);
    else if (sh.MT == 12 && sh.NC == 1) miss = launch_shape<12, 1, 1>(epi, out, g, st, a);
#endif
    if (miss) return sonet::fail(SONET_ERR_UNSUPPORTED, "%s: no kernel for tile shape %d x %d with epilogue %d and outputs %d", what, sh.MT, sh.NC, epi, out);
    if (stats_ws) sonet::launch_stats_finalize(reinterpret_cast<const double *>(stats_ws), (int)ncol, Cout, 1.0 / ((double)B * L), mean, var, st);
    return sonet::launched(what);
}

/* y = act((W . cat(x1, x2) [+ zadd[b][o][zidx[b][l]]]) * scale + shift) on P16 inputs; outputs: y (f32 [B][Cout][L]) and / or yp (P16).
 * x1: B x C1 channels x L1 columns (C1 % 16 == 0 when x2 is given), read through gidx [B][L] when given; x2: B x C2 x L.
 * stats_ws / mean / var: training forward (batch statistics of y from the epilogue; needs y). */
extern "C" int sonet_pointmlp_h3p(const void *x1p, int C1, int L1, const int32_t *gidx, const void *x2p, int C2, const void *Wp,
                                  const float *scale, const float *shift, int relu, float *y, void *yp, int B, int Cout, int L,
                                  const float *zadd, const int32_t *zidx, int ZM, void *stats_ws, float *mean, float *var,
                                  sonet_stream_t stream)
{
    return h3p_launch("sonet_pointmlp_h3p", x1p, C1, L1, gidx, x2p, C2, Wp, scale, shift, relu, y, yp, B, Cout, L, zadd, zidx, ZM, stats_ws, mean, var,
                      0, 0, 0, 0, stream);
}

/* The layer followed by a max over groups of columns, in one launch (node-level stage, flat column axis: ONE cloud of L columns, L % 128 == 0).
 * Every 128-column block holds G groups of GK consecutive columns (G GK <= 128; the columns behind them are padding and take no part);
 * group g of block i is output column i G + g, ngout of them.  Exactly one output: yp = P16 planes of a 1 x Cout x Lout activation (Lout >= ngout;
 * the columns from ngout on are not written)
 * (KNNModule's max over the K neighbours, models/layers.py:365), or y = f32 [ngout][Cout] (the global max over a cloud's nodes,
 * models/networks.py:197).  NaN wins, as in torch.max. */
extern "C" int sonet_pointmlp_h3p_gmax(const void *x1p, int C1, const void *x2p, int C2, const void *Wp, const float *scale, const float *shift,
                                       int relu, int Cout, int L, int GK, int G, int ngout, int Lout, float *y, void *yp, sonet_stream_t stream)
{
    const char *what = "sonet_pointmlp_h3p_gmax";
    SONET_REQUIRE((y == nullptr) != (yp == nullptr), "%s: exactly one of y (f32 [ngout][Cout]) and yp (P16 planes)", what);
    SONET_REQUIRE(L > 0 && L % 128 == 0, "%s: L=%d must be a positive multiple of 128", what, L);
    SONET_REQUIRE(GK >= 1 && G >= 1 && G * GK <= 128 && ngout >= 1 && (long long)ngout <= (long long)(L / 128) * G, "%s: bad grouping GK=%d G=%d ngout=%d", what, GK, G, ngout);
    SONET_REQUIRE(!yp || Lout >= ngout, "%s: Lout=%d (the planes' column count) below ngout=%d", what, Lout, ngout);
    if (yp) { if (G > 16) return sonet::fail(SONET_ERR_UNSUPPORTED, "%s: at most 16 groups per 128 columns for P16 output (G=%d)", what, G); }
    else if (G > 2 || GK % 4 != 0) return sonet::fail(SONET_ERR_UNSUPPORTED, "%s: f32 output takes G <= 2 groups of GK %% 4 == 0 columns (G=%d GK=%d)", what, G, GK);
    return h3p_launch(what, x1p, C1, L, nullptr, x2p, C2, Wp, scale, shift, relu, y, yp, 1, Cout, L, nullptr, nullptr, 0, nullptr, nullptr, nullptr,
                      GK, G, ngout, Lout, stream);
}
