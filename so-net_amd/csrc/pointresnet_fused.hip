// pointresnet_fused.hip -- the whole first PointNet of the encoder as ONE kernel (eval mode), third generation.
//
// Replaces the four EquivariantLayer launches of PointResNet.forward (models/layers.py:419-432, built at
// models/networks.py:82-83 as 6 -> 64 -> 128 -> 256 -> [64 + 256] -> 384 with BN + ReLU on the first three layers)
// when BatchNorm runs on its running statistics.  HBM sees only the 6-channel input and the 384-channel output (or,
// pool variant, only the per-node maxima: models/networks.py:175-185).
//
// Arithmetic: fp32 operands split into fp16 pieces, three v_mfma_f32_32x32x16_f16 per product set with fp32
// accumulation ("operand split" below).  Per accumulator the MFMA sequence (K chunk order, term order l, m, h) is the
// one of the second-generation kernel, so the results are bit-identical to it.
//
// Work decomposition (what changed).  The second generation gave every wave its own 32 points and ALL channels: each
// 1-KiB weight fragment read from LDS fed one MFMA per wave, every wave re-split layer 4's input in each of its passes
// (4.9 VALU + 0.92 LDS instructions per MFMA), and a barrier every 36 MFMAs published the shared weight ring.  Here a
// workgroup = 4 waves owns 64 consecutive points (two 32-column MFMA tiles c = 0, 1) and the waves split the OUTPUT
// CHANNELS of layers 2-4:
//   layer 1 (6 -> 64):    every wave computes it for all 64 points (12 MFMAs, 2 % redundant work) and keeps the result;
//   layer 2 (64 -> 128):  wave w computes output tile w;             its input is the wave's own layer-1 result;
//   layer 3 (128 -> 256): wave w computes output tiles 2w, 2w+1;     its input is layer 2 of all waves, through LDS;
//   layer 4 (320 -> 384): wave w computes output tiles 3w .. 3w+2;   input: own layer 1 + layer 3 of all waves (LDS).
// Activations are handed over PRE-SPLIT: when a wave's accumulators are complete they are turned once (BatchNorm affine +
// ReLU + split: a "job" of 36 VALU instructions per 8 values) into the fp16 pieces (32 xh, fp16(32 xm)) that ARE the B
// operand of the next layer, written to LDS in fragment layout (one ds_write_b128 per piece) and read back by every
// wave with ds_read_b128: 4 KiB of B per K chunk feed 18 MFMAs (0.22 KiB per MFMA instead of 0.67-0.92).
// Weights never touch LDS: every wave streams ONLY its own output tiles' fragments (a quarter of the stream, packed
// wave-major in consumption order) from L2 straight into registers, two steps ahead of their use, and every fragment
// feeds both column tiles.  Two barriers per 64-point tile (layer-2 and layer-3 hand-over) instead of 27 per 128.
//
// Software pipeline.  The short dependent front of a tile (x -> layer 1 -> job -> layer 2 -> job -> LDS) would leave the
// matrix pipe idle, so the front of tile i+1 is computed INSIDE layer 4 of tile i (its 36 MFMAs and 12 jobs fill VALU
// slots behind layer 4's MFMAs); the jobs of layer 3 ride behind the first four steps of layer 4 (which read the wave's
// own layer-1 registers, not LDS).  Per tile and wave: barrier, layer 3 (96 MFMAs), layer 4 steps 0-3 + layer-3 jobs,
// barrier, layer 4 steps 4-19 + front of the next tile, epilogue.
//
// Register chaining.  v_mfma_f32_32x32x16_f16 produces D[row = (r&3) + 8*(r>>2) + 4*(l>>5)][col = l&31] in register r
// of lane l and consumes B[k = 8*(l>>5) + e][col = l&31], e = 0..7.  Registers 8q .. 8q+7 of an output tile therefore
// ARE the B operand of a 16-channel chunk of the next layer, provided the next layer's weights are packed with the
// matching channel order     k = 8h + e   <->   channel 32*t + 16*q + (e&3) + 8*(e>>2) + 4*h     ("chained" packing).
#include "common.hpp"
#include <stdlib.h>
#include <type_traits>

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2_t __attribute__((ext_vector_type(2)));
typedef unsigned u32x4_t __attribute__((ext_vector_type(4)));
typedef float f32x2_t __attribute__((ext_vector_type(2)));

constexpr int PF_THREADS = 256, PF_WAVES = 4;
constexpr int TPTS = 64;                                       // points per workgroup tile (two column tiles)
constexpr int T0 = 2, T1 = 4, T2 = 8, T3 = 12;               // output tiles (x32 channels) of the four layers
constexpr int KC2 = 2 * T0, KC3 = 2 * T1, KC4 = 2 * T0 + 2 * T2;   // 16-channel K chunks per layer
constexpr int W2T = T1 / PF_WAVES, W3T = T2 / PF_WAVES, W4T = T3 / PF_WAVES;   // output tiles per wave: 1, 2, 3
static_assert(W2T == 1 && W3T == 2 && W4T == 3, "the step code below is written for this split");
constexpr int NTERM = 2;                                       // W slices per (cout tile, K chunk): h and l
// the weight stream: [layer 1: 4 slices, shared][wave 0: L2 8, L3 32, L4 120][wave 1 ...] ...
constexpr int NS_L1 = T0 * NTERM, NS_L2 = KC2 * W2T * NTERM, NS_L3 = KC3 * W3T * NTERM, NS_L4 = KC4 * W4T * NTERM;
constexpr int NS_WAVE = NS_L2 + NS_L3 + NS_L4;
constexpr int NSLICE = NS_L1 + PF_WAVES * NS_WAVE;
constexpr int CH_TOTAL = 32 * (T0 + T1 + T2 + T3);
constexpr int LB1 = 0, LB2 = 32 * T0, LB3 = 32 * (T0 + T1), LB4 = 32 * (T0 + T1 + T2);
constexpr int C4 = 32 * T3;                                    // 384 output channels

// ---- fp32 -> 3 x fp16 operand split ------------------------------------------------------------------
// x = xh + xm exactly, xh = fp16(x) (11 significand bits), xm the residual; 32 * (a*b) is taken as
//     ah * (32 bh)  +  ah * fp16(32 bm)  +  fp16(32 am) * bh
// i.e. THREE fp16 MFMAs with fp32 accumulation (the dropped am*bm and the rounding of the scaled residuals are
// <= 2^-22 relative).  The factor 32 keeps the residuals out of the fp16 subnormals; it is carried by the ACCUMULATOR
// (every power-of-two scaling is exact) and taken out again by the layer's affine, so the first two terms share ONE
// weight operand: the stream holds two slices per (cout tile, K chunk), ah and fp16(32 am), for three MFMAs.
// Operand range: |x| <= 2047 (32 x must fit fp16); the split clamps and the range log reports it.
// Naming: B side pieces h = 32 bh, m = fp16(32 bm), l = bh = h * 2^-5 (derived when used, never stored);
// terms in accumulation order: (A.l, B.l), (A.h, B.m), (A.h, B.h).
constexpr float F16_MAX = 2047.0f;                             // 32 * 2047 = 65504, the largest fp16
constexpr float F16_MAX32 = 65504.0f;
constexpr float ACC_UNSCALE = 0.03125f;                        // accumulators hold 32 * (W . x)
constexpr unsigned F16_2_M5_PK = 0x28002800u;                  // packed fp16 (2^-5, 2^-5)
__device__ __forceinline__ unsigned cvt_pk_f16(float lo, float hi) {        // one v_cvt_pk_f16_f32 (round to nearest even)
    const f32x2_t v = {lo, hi};
    return __builtin_bit_cast(unsigned, __builtin_convertvector(v, f16x2_t));
}
__device__ __forceinline__ float f16_lo(unsigned pk) { return (float)__builtin_bit_cast(f16x2_t, pk)[0]; }
__device__ __forceinline__ float f16_hi(unsigned pk) { return (float)__builtin_bit_cast(f16x2_t, pk)[1]; }
__device__ __forceinline__ unsigned pk_mul_f16(unsigned a, unsigned b) {
    const f16x2_t r = __builtin_bit_cast(f16x2_t, a) * __builtin_bit_cast(f16x2_t, b);
    return __builtin_bit_cast(unsigned, r);
}
__device__ __forceinline__ float clamp_f16(float x) { return __builtin_fminf(__builtin_fmaxf(x, -F16_MAX), F16_MAX); }
__device__ __forceinline__ f16x8 as_f16x8(u32x4_t v) { return __builtin_bit_cast(f16x8, v); }
__device__ __forceinline__ u32x4_t piece_l(u32x4_t h) {        // xh = (32 xh) * 2^-5 (exact)
    u32x4_t r;
    r[0] = pk_mul_f16(h[0], F16_2_M5_PK); r[1] = pk_mul_f16(h[1], F16_2_M5_PK);
    r[2] = pk_mul_f16(h[2], F16_2_M5_PK); r[3] = pk_mul_f16(h[3], F16_2_M5_PK);
    return r;
}
// the network input (clamped, no affine): 8 channel values of one point -> (h, m)
__device__ __forceinline__ void split_input(const float (&v)[8], u32x4_t &h, u32x4_t &m) {
#pragma unroll
    for (int p = 0; p < 4; ++p) {
        const float x0 = clamp_f16(v[2 * p]), x1 = clamp_f16(v[2 * p + 1]);
        const unsigned hh = cvt_pk_f16(32.f * x0, 32.f * x1);                         // 32 xh
        h[p] = hh;
        m[p] = cvt_pk_f16(32.f * x0 - f16_lo(hh), 32.f * x1 - f16_hi(hh));            // 32 (x - xh), exact before the rounding
    }
}

// ---- a "job": BatchNorm affine + ReLU + split of 8 accumulator values (registers 8Q..8Q+7 of an output tile) ----
// One VALU instruction at a time (volatile asm: instruction selection floats pure VALU ops across sched_barrier and
// clumps them; a clump of dependent VALU between two MFMAs stalls the matrix pipe), so that the steps can place a few
// of them behind each MFMA.  The accumulators hold 32 W.x and the coefficients are (scale, 32 shift): the affine
// delivers 32 x directly (bit-identical to 32 * fl(acc * scale/32 + shift): power-of-two scalings commute with the
// rounding), ReLU and the fp16 range clamp are one v_med3 against 32 * 2047.
__device__ __forceinline__ float pin_fma(float a, float s, float b) { float r; asm volatile("v_fma_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(s), "v"(b)); return r; }
__device__ __forceinline__ float pin_relu_clamp32(float a) { float r; asm volatile("v_med3_f32 %0, %1, 0, %2" : "=v"(r) : "v"(a), "v"(F16_MAX32)); return r; }
__device__ __forceinline__ unsigned pin_cvt(float lo, float hi) { unsigned r; asm volatile("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(r) : "v"(lo), "v"(hi)); return r; }
// x32 - (fp16 half of pk = 32 xh) = 32 * (x - xh), exact
__device__ __forceinline__ float pin_res_lo(unsigned pk, float x32) { float r; asm volatile("v_fma_mix_f32 %0, %1, %2, %3 op_sel_hi:[1,0,0]" : "=v"(r) : "v"(pk), "v"(-1.0f), "v"(x32)); return r; }
__device__ __forceinline__ float pin_res_hi(unsigned pk, float x32) { float r; asm volatile("v_fma_mix_f32 %0, %1, %2, %3 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(r) : "v"(pk), "v"(-1.0f), "v"(x32)); return r; }
// range log: max over the post-affine, pre-clamp activations (x 32) as signed-int-ordered bits (positive side: what ReLU keeps)
__device__ __forceinline__ int pin_max3_i32(int m, float a, float b) { int r; asm volatile("v_max3_i32 %0, %1, %2, %3" : "=v"(r) : "v"(m), "v"(a), "v"(b)); return r; }

struct JobSc { float2 sc[8]; };                                // (scale, 32 shift) of the job's 8 channels
struct JobOut { u32x4_t h, m; };
struct JobX { float x[8]; };                                   // the 8 values in flight
constexpr int JOB_OPS = 36;
// op I of 36 = two groups of 18 (values 4g..4g+3, neighbours independent): affine x4, range max x2, relu+clamp x4,
// 32xh x2 (cvt), residual x4, 32xm x2 (cvt).
template <int Q, int I> __device__ __forceinline__ void job_op(const f32x16 &a, JobX &v, const JobSc &s, JobOut &o, int &rm) {
    static_assert(I >= 0 && I < JOB_OPS, "");
    constexpr int g = I / 18, k = I % 18, R = 8 * Q;
    if constexpr (k < 4) { constexpr int e = 4 * g + k; v.x[e] = pin_fma(a[R + e], s.sc[e].x, s.sc[e].y); }
    else if constexpr (k < 6) { constexpr int e = 4 * g + 2 * (k - 4); rm = pin_max3_i32(rm, v.x[e], v.x[e + 1]); }
    else if constexpr (k < 10) { constexpr int e = 4 * g + k - 6; v.x[e] = pin_relu_clamp32(v.x[e]); }
    else if constexpr (k < 12) { constexpr int P = 2 * g + (k - 10); o.h[P] = pin_cvt(v.x[2 * P], v.x[2 * P + 1]); }
    else if constexpr (k < 16) { constexpr int e = 4 * g + (k - 12); v.x[e] = (e & 1) ? pin_res_hi(o.h[e >> 1], v.x[e]) : pin_res_lo(o.h[e >> 1], v.x[e]); }
    else { constexpr int P = 2 * g + (k - 16); o.m[P] = pin_cvt(v.x[2 * P], v.x[2 * P + 1]); }
}
template <int Q, int I0, int I1> __device__ __forceinline__ void job_ops(const f32x16 &a, JobX &v, const JobSc &s, JobOut &o, int &rm) {
    if constexpr (I0 < I1 && I0 < JOB_OPS) { job_op<Q, I0>(a, v, s, o, rm); job_ops<Q, I0 + 1, I1>(a, v, s, o, rm); }
}

// compile-time loop: f(integral_constant<int, i>) for i in [I0, I1)
template <int I0, int I1, class F> __device__ __forceinline__ void static_for(F &&f) {
    if constexpr (I0 < I1) { f(std::integral_constant<int, I0>{}); static_for<I0 + 1, I1>(f); }
}
template <int V> using IC = std::integral_constant<int, V>;
#define SFOR(var, N) static_for<0, (N)>([&](auto var##_c_) __attribute__((always_inline)) { constexpr int var = decltype(var##_c_)::value;
#define SEND });

// ---- weight stream packing -------------------------------------------------------------------------
// slice s of the stream -> (layer, cout tile, K chunk, split term); one thread per (slice, lane).
__global__ __launch_bounds__(256) void pointresnet_pack_kernel(const float *__restrict__ W1, const float *__restrict__ W2,
                                                                const float *__restrict__ W3, const float *__restrict__ W4,
                                                                int Cin0, uint4 *__restrict__ out, unsigned *__restrict__ trailer)
{
    const int t = blockIdx.x * 256 + threadIdx.x;
    if (t >= NSLICE * 64) return;
    RangeAcc wr = {0, 0u};
    const int lane = t & 63, s = t >> 6;
    const int i = lane & 31, h = lane >> 5;
    const float *W = nullptr;
    int Cin = 0, ct = 0, kc = 0, term = 0;
    bool chained = true;
    if (s < NS_L1) {                      // L1: 2 tiles x 1 chunk, standard channel order (the input comes from memory)
        term = s % NTERM; ct = s / NTERM; kc = 0; W = W1; Cin = Cin0; chained = false;
    } else {                              // per wave: consumption order = K chunk, then the wave's tiles, then the split term
        const int w = (s - NS_L1) / NS_WAVE, r = (s - NS_L1) % NS_WAVE;
        if (r < NS_L2) {
            term = r % NTERM; kc = r / (NTERM * W2T); ct = w * W2T + (r / NTERM) % W2T; W = W2; Cin = 32 * T0;
        } else if (r < NS_L2 + NS_L3) {
            const int u = r - NS_L2; term = u % NTERM; kc = u / (NTERM * W3T); ct = w * W3T + (u / NTERM) % W3T; W = W3; Cin = 32 * T1;
        } else {
            const int u = r - NS_L2 - NS_L3; term = u % NTERM; kc = u / (NTERM * W4T); ct = w * W4T + (u / NTERM) % W4T; W = W4; Cin = 32 * (T0 + T2);
        }
    }
    unsigned w4[4] = {0, 0, 0, 0};
#pragma unroll
    for (int p = 0; p < 4; ++p) {
        float v[2];
#pragma unroll
        for (int z = 0; z < 2; ++z) {
            const int e = 2 * p + z;
            const int c = chained ? kc * 16 + (e & 3) + 8 * (e >> 2) + 4 * h : kc * 16 + 8 * h + e;
            v[z] = c < Cin ? W[(long long)(ct * 32 + i) * Cin + c] : 0.f;
        }
        range_track(wr, v[0], v[1]);
        // A-side slices: 0 = fp16(w) (terms h and m), 1 = fp16(32 * (w - h)) (term l)
        const unsigned hh = cvt_pk_f16(v[0], v[1]);
        const unsigned ll = cvt_pk_f16(32.f * (v[0] - f16_lo(hh)), 32.f * (v[1] - f16_hi(hh)));
        w4[p] = term == 0 ? hh : ll;
    }
    out[(long long)s * 64 + lane] = make_uint4(w4[0], w4[1], w4[2], w4[3]);
    range_publish(trailer, wave_umax(range_amax_bits(wr)), lane);         // max |w| over the four layers (range log, word 1)
}

// ---- the fused kernel ------------------------------------------------------------------------------
// SEGMAX = the per-node max-pool epilogue (see below) instead of the y stores; x must then be node-sorted.
constexpr int SEG_SLOTS = 4;                                  // nodes of a 64-point tile pre-reduced in LDS (the rest: global atomics)
constexpr unsigned SEG_INIT = 0x3B85FFFFu;                    // orderable(-1000.0f): the reference's initial running max

__device__ __forceinline__ unsigned ord_f32(unsigned bits) {   // total order; -0 == +0; NaN -> 0 (never wins)
    if (bits == 0x80000000u) bits = 0u;
    const unsigned o = bits ^ ((unsigned)((int)bits >> 31) | 0x80000000u);
    return (bits & 0x7FFFFFFFu) > 0x7F800000u ? 0u : o;
}
#ifdef SONET_PROF
// Profiling build only (make prof; tools/fused_phases.py): per-wave shader-clock cycles spent in each phase.
constexpr int PROF_N = 32;
__device__ long long g_prof[1024 * PROF_N];
#define PROF_DECL long long prof_[PROF_N] = {}; const long long prof_rt0_ = (long long)__builtin_amdgcn_s_memrealtime(); long long prof_t_ = __builtin_readcyclecounter();
#define PROF_MARK(i) { const long long n_ = __builtin_readcyclecounter(); prof_[i] += n_ - prof_t_; prof_t_ = n_; }
#define PROF_DUMP prof_[31] = (long long)__builtin_amdgcn_s_memrealtime() - prof_rt0_; /* 100 MHz constant clock */ if (lane == 0) { for (int i_ = 0; i_ < PROF_N; ++i_) g_prof[(blockIdx.x * PF_WAVES + wave) * PROF_N + i_] = prof_[i_]; }
#else
#define PROF_DECL
#define PROF_MARK(i)
#define PROF_DUMP
#endif

#define PF_SB __builtin_amdgcn_sched_barrier(0);

// A fragments of one step: up to 3 output tiles x (h, l)
struct AF { f16x8 h[W4T], l[W4T]; };

template <bool SEGMAX>
__global__ __launch_bounds__(PF_THREADS, 1) void pointresnet_fused_kernel(
    const float *__restrict__ x, int Cin0, const u32x4_t *__restrict__ Wst, const float2 *__restrict__ affine_g /*[CH_TOTAL] (scale, shift); last layer (1, bias)*/,
    float *__restrict__ y, int L, int tpc /*64-point tiles per cloud*/, long long ntiles,
    const int32_t *__restrict__ ids_sorted, const int32_t *__restrict__ pos0, unsigned *__restrict__ pooled, float *__restrict__ v0, int M,
    unsigned *__restrict__ partial /*[ntiles][SEG_SLOTS][384] keys of the tile's first SEG_SLOTS nodes*/,
    unsigned *__restrict__ rlog /*optional range-log slot: [0] max |x in|, [1] max |w|, [2] max post-affine input of layers 2-4 (bits)*/)
{
    // activations in B-fragment layout: [16-channel chunk][column tile][piece h, m][lane] x 16 bytes
    __shared__ u32x4_t act2s[KC3][2][2][64];                   // layer-2 output, 32 KiB
    __shared__ u32x4_t act3s[2 * T2][2][2][64];                // layer-3 output, 64 KiB
    __shared__ __attribute__((aligned(16))) float2 aff[CH_TOTAL];
    __shared__ unsigned bins[SEGMAX ? SEG_SLOTS : 1][SEGMAX ? C4 : 1];

    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int j = lane & 31, h = lane >> 5;
    bool l4_unit_lane = true;                                   // layer 4 has no BatchNorm in the reference: scale == 1
    for (int c = threadIdx.x; c < CH_TOTAL; c += PF_THREADS) {
        const float2 v = affine_g[c];
        // layers 1-3: the split wants 32 x = acc * scale + 32 shift (the accumulators carry a factor 32);
        // layer 4: y = acc * scale / 32 + shift
        aff[c] = c < LB4 ? make_float2(v.x, 32.f * v.y) : make_float2(v.x * ACC_UNSCALE, v.y);
        if (c >= LB4 && v.x != 1.0f) l4_unit_lane = false;
    }
    if constexpr (SEGMAX) {
        for (int i = threadIdx.x; i < SEG_SLOTS * C4; i += PF_THREADS) (&bins[0][0])[i] = SEG_INIT;
    }
    const bool l4_unit = __syncthreads_and(l4_unit_lane) != 0;   // then max(x + b) = max(x) + b exactly: bias after the pool
    PROF_DECL

    const unsigned rowB = (unsigned)L * 4u;
    // this wave's part of the weight stream (lane-linear 16-byte fragments), read through buffer descriptors: the lane
    // offset is ONE VGPR, every slice offset a scalar / immediate (with flat pointers hipcc hoists a 64-bit VGPR address
    // per fragment out of the tile loop: 300+ registers)
    const __amdgpu_buffer_rsrc_t rw1 = __builtin_amdgcn_make_buffer_rsrc(const_cast<u32x4_t *>(Wst), 0, NS_L1 * 1024, 0x00020000);
#ifdef SONET_ABL_SAMEW                                          // experiment: every wave reads wave 0's stream (L1 hits)
    const __amdgpu_buffer_rsrc_t rww = __builtin_amdgcn_make_buffer_rsrc(const_cast<u32x4_t *>(Wst) + (size_t)NS_L1 * 64, 0, NS_WAVE * 1024, 0x00020000);
#else
    const __amdgpu_buffer_rsrc_t rww = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<u32x4_t *>(Wst) + (size_t)(NS_L1 + wave * NS_WAVE) * 64, 0, NS_WAVE * 1024, 0x00020000);
#endif
    const unsigned vow = (unsigned)lane * 16u;
#ifndef SONET_WAUX
#define SONET_WAUX 0
#endif
#define PF_WLOAD(rsrc, slice) __builtin_bit_cast(f16x8, __builtin_amdgcn_raw_buffer_load_b128(rsrc, vow, (unsigned)(slice) * 1024u, SONET_WAUX))
#ifdef SONET_ABL_NOW                                            // experiment: the main steps keep their first fragments (no weight traffic)
#define PF_WLOAD_MAIN(rsrc, slice) f_keep_
#else
#define PF_WLOAD_MAIN(rsrc, slice) PF_WLOAD(rsrc, slice)
#endif
    constexpr int WO2 = 0, WO3 = NS_L2, WO4 = NS_L2 + NS_L3;     // slice offsets of layers 2-4 in the wave's stream
    u32x4_t *const act2w = &act2s[0][0][0][lane];               // + ((chunk * 2 + c) * 2 + piece) * 64
    u32x4_t *const act3w = &act3s[0][0][0][lane];
#define PF_ACT(base, chunk, c, piece) (base)[(((chunk) * 2 + (c)) * 2 + (piece)) * 64]

    // ---- the main steps: global step g = 0..7 layer 3 (K chunk g, tiles 2w, 2w+1), g = 8..27 layer 4 (K chunk g - 8,
    // tiles 3w..3w+2).  A fragments live in a ring af[g % AFD]; step g starts by requesting the fragments of step g + AFD - 1
    // into the buffer step g - 1 has just read (with 4 waves pulling 6 KiB each per step the texture path is ~2/3 busy: a
    // request issued one step ahead arrives late).
#ifndef SONET_AFD
#define SONET_AFD 4
#endif
    constexpr int AFD = SONET_AFD;                               // fragment buffers; step g reads af[g % AFD], refilled AFD - 1 steps ahead
    static_assert((KC3 + KC4) % AFD == 0, "the buffer of a step must not depend on the tile");
    AF af[AFD];
    // fragment i of global step g (wraps: next tile): i < NT the `l` slices, then the `h` slices
    auto load_frag = [&](auto gc, auto ic, AF &f) __attribute__((always_inline)) {
        constexpr int g = decltype(gc)::value % (KC3 + KC4), i = decltype(ic)::value;
        const f16x8 f_keep_ = f.l[0]; (void)f_keep_;
        if constexpr (g < KC3) {
            if constexpr (i < W3T) f.l[i] = PF_WLOAD_MAIN(rww, WO3 + g * NTERM * W3T + NTERM * i + 1);
            else if constexpr (i < 2 * W3T) f.h[i - W3T] = PF_WLOAD_MAIN(rww, WO3 + g * NTERM * W3T + NTERM * (i - W3T));
        } else {
            if constexpr (i < W4T) f.l[i] = PF_WLOAD_MAIN(rww, WO4 + (g - KC3) * NTERM * W4T + NTERM * i + 1);
            else if constexpr (i < 2 * W4T) f.h[i - W4T] = PF_WLOAD_MAIN(rww, WO4 + (g - KC3) * NTERM * W4T + NTERM * (i - W4T));
        }
    };
#define PF_MFMA(ACCE, A_, B_, ZEROC, SWAPC)                                                              \
    {                                                                                                    \
        if constexpr (ZEROC) {                                                                           \
            const f32x16 cz_ = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f}; \
            if constexpr (SWAPC) ACCE = __builtin_amdgcn_mfma_f32_32x32x16_f16(B_, A_, cz_, 0, 0, 0);    \
            else ACCE = __builtin_amdgcn_mfma_f32_32x32x16_f16(A_, B_, cz_, 0, 0, 0);                    \
        } else {                                                                                         \
            if constexpr (SWAPC) ACCE = __builtin_amdgcn_mfma_f32_32x32x16_f16(B_, A_, ACCE, 0, 0, 0);   \
            else ACCE = __builtin_amdgcn_mfma_f32_32x32x16_f16(A_, B_, ACCE, 0, 0, 0);                   \
        }                                                                                                \
    }
    // One STEP = a 16-channel K chunk x NT output tiles x 2 column tiles = 6 NT MFMAs, hand-scheduled (sched_barrier
    // after every MFMA: with one wave per SIMD nothing else hides a latency).  Term order (A.l, B.l), (A.h, B.m),
    // (A.h, B.h) per accumulator; SLOT(q) places VALU / LDS work behind MFMA q; G = global step (fragment refills).
#define PF_REQ(G, NT, qq)                                        /* 6 NT MFMAs, up to 6 fragments: one behind every NT-th */ \
    {                                                                                                    \
        if constexpr ((G) >= 0 && (qq) % (NT) == 0) {                                                    \
            constexpr int gn_ = ((G) >= 0 ? (G) : 0) + AFD - 1;                                          \
            load_frag(IC<gn_>{}, IC<(qq) / (NT)>{}, af[gn_ % AFD]);                                      \
        }                                                                                                \
    }
#define PF_STEP(NT, G, FR, ACC, BH0, BM0, BL0, BH1, BM1, BL1, ZERO, SWAP, SLOT)                          \
    {                                                                                                    \
        const f16x8 bh_[2] = {as_f16x8(BH0), as_f16x8(BH1)}, bm_[2] = {as_f16x8(BM0), as_f16x8(BM1)}, bl_[2] = {as_f16x8(BL0), as_f16x8(BL1)}; \
        /* ONE weight request behind every NT-th MFMA (the fragments of step G + AFD - 1, into the buffer the previous step */ \
        /* read): a wave issues in order, and a burst of requests holds its MFMAs back while the texture unit takes them    */ \
        PF_SB                                                                                            \
        SFOR(q, 2 * (NT)) constexpr int c = q / (NT), u = q % (NT);                                      \
            PF_MFMA(ACC(u, c), FR.l[u], bl_[c], (ZERO), (SWAP))                                          \
            PF_REQ((G), (NT), q)                                                                               \
            SLOT(q)                                                                                      \
            PF_SB                                                                                        \
        SEND                                                                                             \
        PF_SB                                                                                            \
        SFOR(q, 2 * (NT)) constexpr int c = q / (NT), u = q % (NT);                                      \
            PF_MFMA(ACC(u, c), FR.h[u], bm_[c], false, (SWAP))                                           \
            PF_REQ((G), (NT), 2 * (NT) + q)                                                                    \
            SLOT(2 * (NT) + q)                                                                           \
            PF_SB                                                                                        \
        SEND                                                                                             \
        SFOR(q, 2 * (NT)) constexpr int c = q / (NT), u = q % (NT);                                      \
            PF_MFMA(ACC(u, c), FR.h[u], bh_[c], false, (SWAP))                                           \
            PF_REQ((G), (NT), 4 * (NT) + q)                                                                    \
            SLOT(4 * (NT) + q)                                                                           \
            PF_SB                                                                                        \
        SEND                                                                                             \
        PF_SB                                                                                            \
    }
    // (scale, 32 shift) of the 8 channels of a job: channels CH + (e&3) + 8(e>>2), CH = layer base + 32 tile + 16 Q + 4 h
#define PF_LOAD_SC(scv, CH)                                                                              \
    {                                                                                                    \
        const float4 *ap_ = reinterpret_cast<const float4 *>(&aff[(CH)]);                                \
        const float4 c0_ = ap_[0], c1_ = ap_[1], c2_ = ap_[4], c3_ = ap_[5];                             \
        scv.sc[0] = make_float2(c0_.x, c0_.y); scv.sc[1] = make_float2(c0_.z, c0_.w);                    \
        scv.sc[2] = make_float2(c1_.x, c1_.y); scv.sc[3] = make_float2(c1_.z, c1_.w);                    \
        scv.sc[4] = make_float2(c2_.x, c2_.y); scv.sc[5] = make_float2(c2_.z, c2_.w);                    \
        scv.sc[6] = make_float2(c3_.x, c3_.y); scv.sc[7] = make_float2(c3_.z, c3_.w);                    \
    }
    // ops [I0, I1) of the 72 of a job PAIR (both column tiles of one (tile, half Q): same coefficients)
#define PF_JOB2(Q, I0, I1, A0, A1, OUT0, OUT1)                                                           \
    {                                                                                                    \
        if constexpr ((I0) < JOB_OPS) job_ops<Q, (I0), ((I1) < JOB_OPS ? (I1) : JOB_OPS)>(A0, jx_, sc_, OUT0, rmax_); \
        if constexpr ((I1) > JOB_OPS) job_ops<Q, ((I0) > JOB_OPS ? (I0) - JOB_OPS : 0), (I1) - JOB_OPS>(A1, jx_, sc_, OUT1, rmax_); \
    }
    // ... placed OPS at a time behind the MFMAs of a step, from the third on (the coefficient reads need that long)
#ifdef SONET_ABL_NOJOB
#define PF_JOBS_ON false
#else
#define PF_JOBS_ON true
#endif
#define PF_SLOT2(q, OPS, Q, A0, A1, OUT0, OUT1)                                                          \
    {                                                                                                    \
        if constexpr (PF_JOBS_ON && (q) >= 2 && (OPS) * ((q) - 2) < 2 * JOB_OPS) {                                     \
            constexpr int i0_ = (OPS) * ((q) - 2);                                                       \
            PF_JOB2(Q, i0_, i0_ + (OPS), A0, A1, OUT0, OUT1)                                             \
        }                                                                                                \
    }

    // ---- the front of a tile: x -> layer 1 (all 64 points, every wave) -> layer 2 (this wave's tile) -> LDS ----
    float xin[2][8];
    int nid_n[2] = {-1, -1}, n0_n = 0, nlast_n = 0, pos0_n = 0;   // pool bookkeeping of the NEXT tile (read with its x: a round trip at the
                                                                // top of the tile sat in front of barrier 1)
    u32x4_t xh[2], xm[2], xl[2];
    AF fl1;                                                     // layer 1's 4 fragments (h, l of 2 tiles)
    f16x8 fl2h, fl2l;                                           // layer 2: one tile, one chunk
    f32x16 acc1[T0][2], acc2[2];
    JobOut a1n[T0][2][2];                                       // layer-1 output of the NEXT tile [tile][half][column tile]
    JobOut a2o[2][2];                                           // this wave's layer-2 tile [half][column tile]
    int rmax_ = 0;                                              // range log: running max of the post-affine inputs of layers 2-4 (x 32)
    unsigned xin_r = 0u;                                        // ... and of |network input|, as ordered bit patterns
    auto front_load_x = [&](long long t) __attribute__((always_inline)) {
        t = t < ntiles ? t : ntiles - 1;                        // past the end: recompute the last tile's front (never consumed)
        const long long bb = t / tpc;
        const int t0 = (int)(t - bb * tpc) * TPTS;
        const __amdgpu_buffer_rsrc_t rxx = __builtin_amdgcn_make_buffer_rsrc(
            const_cast<float *>(x + bb * (long long)Cin0 * L), 0, (int)((unsigned)Cin0 * rowB), 0x00020000);
#pragma unroll
        for (int c = 0; c < 2; ++c) {
            const int ll0 = t0 + 32 * c;
            const int lcc = ll0 + j < L ? ll0 + j : (ll0 < L ? ll0 : 0);
#pragma unroll
            for (int e = 0; e < 8; ++e)                          // rows >= Cin0 are out of range of the descriptor: 0
                xin[c][e] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rxx, (unsigned)(8 * h * L + lcc) * 4u, (unsigned)e * rowB, 0));
            if constexpr (SEGMAX) nid_n[c] = ll0 + j < L ? ids_sorted[bb * (long long)L + ll0 + j] : -1;
        }
        if constexpr (SEGMAX) {
            const int32_t *idb = ids_sorted + bb * (long long)L;
            n0_n = idb[t0];                                      // first / last node of the tile
            nlast_n = idb[(t0 + TPTS - 1 < L ? t0 + TPTS - 1 : L - 1)];
            pos0_n = pos0[bb];
        }
    };
    auto front_load_w1 = [&]() __attribute__((always_inline)) {
        SFOR(u, T0) fl1.h[u] = PF_WLOAD(rw1, NTERM * u); fl1.l[u] = PF_WLOAD(rw1, NTERM * u + 1); SEND
    };
    auto front_split_x = [&]() __attribute__((always_inline)) {
#pragma unroll
        for (int c = 0; c < 2; ++c) {
#pragma unroll
            for (int p = 0; p < 4; ++p) {
                const unsigned a0 = __float_as_uint(xin[c][2 * p]) & 0x7FFFFFFFu, a1 = __float_as_uint(xin[c][2 * p + 1]) & 0x7FFFFFFFu;
                xin_r = max(max(xin_r, a0), a1);
            }
            split_input(xin[c], xh[c], xm[c]);
            xl[c] = piece_l(xh[c]);
        }
    };
#define ACC1(u, c) acc1[u][c]
#define NOSLOT(q)
    auto front_l1 = [&]() __attribute__((always_inline)) {      // 12 MFMAs
        PF_STEP(T0, -1, fl1, ACC1, xh[0], xm[0], xl[0], xh[1], xm[1], xl[1], true, false, NOSLOT)
    };
    auto front_load_w2 = [&](auto kcc) __attribute__((always_inline)) {
        constexpr int kc = decltype(kcc)::value;
        fl2h = PF_WLOAD(rww, WO2 + kc * NTERM); fl2l = PF_WLOAD(rww, WO2 + kc * NTERM + 1);
    };
    auto front_l2 = [&](auto kcc) __attribute__((always_inline)) {    // 6 MFMAs: chunk kc of layer 2, this wave's tile
        constexpr int kc = decltype(kcc)::value, t = kc >> 1, q = kc & 1;
        const f16x8 b0l = as_f16x8(piece_l(a1n[t][q][0].h)), b1l = as_f16x8(piece_l(a1n[t][q][1].h));
        PF_SB
        PF_MFMA(acc2[0], fl2l, b0l, (kc == 0), false) PF_SB
        PF_MFMA(acc2[1], fl2l, b1l, (kc == 0), false) PF_SB
        PF_MFMA(acc2[0], fl2h, as_f16x8(a1n[t][q][0].m), false, false) PF_SB
        PF_MFMA(acc2[1], fl2h, as_f16x8(a1n[t][q][1].m), false, false) PF_SB
        PF_MFMA(acc2[0], fl2h, as_f16x8(a1n[t][q][0].h), false, false) PF_SB
        PF_MFMA(acc2[1], fl2h, as_f16x8(a1n[t][q][1].h), false, false) PF_SB
    };
    auto front_store_a2 = [&]() __attribute__((always_inline)) {  // this wave's layer-2 tile = chunks 2w, 2w+1 of layer 3's input
        SFOR(q, 2) SFOR(c, 2)
            PF_ACT(act2w, 2 * wave + q, c, 0) = a2o[q][c].h; PF_ACT(act2w, 2 * wave + q, c, 1) = a2o[q][c].m;
        SEND SEND
    };
    auto front_exposed = [&](long long t) __attribute__((always_inline)) {   // the whole front back to back (first tile of a workgroup)
        front_load_x(t);
        front_load_w1();
        front_split_x();
        front_l1();
        SFOR(ck, KC2)
            JobSc sc_; JobX jx_;
            PF_LOAD_SC(sc_, LB1 + 32 * (ck >> 1) + 16 * (ck & 1) + 4 * h)
            PF_JOB2((ck & 1), 0, 2 * JOB_OPS, acc1[ck >> 1][0], acc1[ck >> 1][1], a1n[ck >> 1][ck & 1][0], a1n[ck >> 1][ck & 1][1])
        SEND
        SFOR(kc, KC2)
            front_load_w2(IC<kc>{});
            front_l2(IC<kc>{});
        SEND
        SFOR(q, 2)
            JobSc sc_; JobX jx_;
            PF_LOAD_SC(sc_, LB2 + 32 * wave + 16 * q + 4 * h)
            PF_JOB2(q, 0, 2 * JOB_OPS, acc2[0], acc2[1], a2o[q][0], a2o[q][1])
        SEND
        front_store_a2();
    };

    PROF_MARK(0)                                                // kernel prologue
    front_exposed(blockIdx.x);
    // fragments of the first two main steps
#ifdef SONET_ABL_NOW
    SFOR(u, W4T) af[0].l[u] = PF_WLOAD(rww, WO4 + 2 * u + 1); af[0].h[u] = PF_WLOAD(rww, WO4 + 2 * u); af[1].l[u] = PF_WLOAD(rww, WO4 + 6 + 2 * u + 1); af[1].h[u] = PF_WLOAD(rww, WO4 + 6 + 2 * u); SEND
#else
    SFOR(g, AFD - 1) SFOR(i, 2 * W4T) load_frag(IC<g>{}, IC<i>{}, af[g]); SEND SEND
#endif
    PROF_MARK(1)                                                // first front

    int pend_n = 0;
    unsigned *pend = partial;
    for (long long tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const long long b = tile / tpc;
        const int t0 = (int)(tile - b * tpc) * TPTS;             // first point of the tile (same 64 points for all four waves)
        bool pv[2]; int lc[2];
#pragma unroll
        for (int c = 0; c < 2; ++c) {
            const int l0 = t0 + 32 * c;
            pv[c] = l0 + j < L;
            lc[c] = pv[c] ? l0 + j : (l0 < L ? l0 : 0);
        }
        // per-node max-pool bookkeeping of the tile's 2 x 32 (node-sorted) points (used by the epilogue, a tile later)
        int nid[2] = {-1, -1}, n0 = 0, nslots = 0, jpos0[2] = {-1, -1};
        if constexpr (SEGMAX) {                                 // (read by the front of this tile, a tile ago)
            nid[0] = nid_n[0];
            nid[1] = nid_n[1];
            n0 = n0_n;
            const int nlast = nlast_n;
            nslots = nlast - n0 + 1 < SEG_SLOTS ? nlast - n0 + 1 : SEG_SLOTS;
            const int p0 = pos0_n;
#pragma unroll
            for (int c = 0; c < 2; ++c) {
                const int pp = p0 - (t0 + 32 * c);
                jpos0[c] = (pp >= 0 && pp < 32) ? pp : -1;
            }
        }
        // the layer-1 output of THIS tile (made by the front a tile ago) moves out of the front's registers
        JobOut a1[T0][2][2];
        SFOR(t, T0) SFOR(q, 2) SFOR(c, 2) a1[t][q][c] = a1n[t][q][c]; SEND SEND SEND

        __syncthreads();                                        // #1: layer 2 of this tile is in LDS; everyone is done with the previous tile
        if constexpr (SEGMAX) {                                 // the previous tile's maxima: LDS bins -> its partial block
            for (int e = threadIdx.x; e < pend_n; e += PF_THREADS) {
                unsigned *bp = &bins[0][0] + e;
                pend[e] = *bp;
                *bp = SEG_INIT;
            }
        }
        PROF_MARK(2)                                            // tile prologue + barrier 1
        // ---- layer 3: 8 steps, tiles 2w, 2w+1, B = layer 2 from LDS ----
        f32x16 acc3[W3T][2];
#define ACC3(u, c) acc3[u][c]
        u32x4_t bh[2], bm[2], bl[2];                            // B pieces of the current step
        bh[0] = PF_ACT(act2w, 0, 0, 0); bm[0] = PF_ACT(act2w, 0, 0, 1); bh[1] = PF_ACT(act2w, 0, 1, 0); bm[1] = PF_ACT(act2w, 0, 1, 1);
        bl[0] = piece_l(bh[0]); bl[1] = piece_l(bh[1]);
        SFOR(kc, KC3)
            u32x4_t nh[2], nm[2], nl[2];                        // next step's B chunk, read now
            if constexpr (kc + 1 < KC3) {
#ifdef SONET_ABL_NOB
                nh[0] = bh[0]; nm[0] = bm[0]; nh[1] = bh[1]; nm[1] = bm[1];
#else
                nh[0] = PF_ACT(act2w, kc + 1, 0, 0); nm[0] = PF_ACT(act2w, kc + 1, 0, 1); nh[1] = PF_ACT(act2w, kc + 1, 1, 0); nm[1] = PF_ACT(act2w, kc + 1, 1, 1);
#endif
            } else {                                            // layer 4 starts with the wave's own layer-1 chunk 0
                nh[0] = a1[0][0][0].h; nm[0] = a1[0][0][0].m; nh[1] = a1[0][0][1].h; nm[1] = a1[0][0][1].m;
            }
#define SLOT_L3(q) { if constexpr ((q) == 8) nl[0] = piece_l(nh[0]); if constexpr ((q) == 9) nl[1] = piece_l(nh[1]); }
            PF_STEP(W3T, kc, af[kc % AFD], ACC3, bh[0], bm[0], bl[0], bh[1], bm[1], bl[1], (kc == 0), false, SLOT_L3)
#undef SLOT_L3
            bh[0] = nh[0]; bm[0] = nm[0]; bl[0] = nl[0]; bh[1] = nh[1]; bm[1] = nm[1]; bl[1] = nl[1];
        SEND
        PROF_MARK(3)                                            // layer 3
        // ---- layer 4: 20 steps, tiles 3w..3w+2; chunks 0-3 = own layer-1 output, 4-19 = layer 3 from LDS ----
        f32x16 acc[W4T][2];
#define ACC4(u, c) acc[u][c]
        // steps 0-3: the 8 jobs of layer 3 (step s: tile s>>1, half s&1, both column tiles), stored when complete
        SFOR(kc, KC2)
            constexpr int ju = kc >> 1, jq = kc & 1;
            JobSc sc_; JobX jx_;
            JobOut a3o[2];                                      // the job pair's outputs on their way to LDS
            PF_LOAD_SC(sc_, LB3 + 32 * (W3T * wave + ju) + 16 * jq + 4 * h)
            u32x4_t nh[2], nm[2], nl[2];
            if constexpr (kc + 1 < KC2) {
                constexpr int tn = (kc + 1) >> 1, qn = (kc + 1) & 1;
                nh[0] = a1[tn][qn][0].h; nm[0] = a1[tn][qn][0].m; nh[1] = a1[tn][qn][1].h; nm[1] = a1[tn][qn][1].m;
            }
#define SLOT_L4A(q) { PF_SLOT2(q, 6, jq, acc3[ju][0], acc3[ju][1], a3o[0], a3o[1]) \
                      if constexpr (kc + 1 < KC2) { if constexpr ((q) == 14) nl[0] = piece_l(nh[0]); if constexpr ((q) == 15) nl[1] = piece_l(nh[1]); } }
            PF_STEP(W4T, KC3 + kc, af[kc % AFD], ACC4, bh[0], bm[0], bl[0], bh[1], bm[1], bl[1], (kc == 0), SEGMAX, SLOT_L4A)
#undef SLOT_L4A
            SFOR(c, 2)
                PF_ACT(act3w, 2 * (W3T * wave + ju) + jq, c, 0) = a3o[c].h; PF_ACT(act3w, 2 * (W3T * wave + ju) + jq, c, 1) = a3o[c].m;
            SEND
            if constexpr (kc + 1 < KC2) { bh[0] = nh[0]; bm[0] = nm[0]; bl[0] = nl[0]; bh[1] = nh[1]; bm[1] = nm[1]; bl[1] = nl[1]; }
        SEND
        PROF_MARK(4)                                            // layer 4, steps 0-3 (+ layer-3 jobs)
        __syncthreads();                                        // #2: layer 3 of this tile is in LDS
        PROF_MARK(5)
        bh[0] = PF_ACT(act3w, 0, 0, 0); bm[0] = PF_ACT(act3w, 0, 0, 1); bh[1] = PF_ACT(act3w, 0, 1, 0); bm[1] = PF_ACT(act3w, 0, 1, 1);
        bl[0] = piece_l(bh[0]); bl[1] = piece_l(bh[1]);
        // steps 4-19 carry the front of the NEXT tile of this workgroup:
        //   step 4: x loads + layer-1 fragments;  6: split x;  7: layer 1 (12 MFMAs);  8-11: its 8 jobs (a pair per step);
        //   12-15: layer 2, one chunk per step (6 MFMAs, fragments read a step ahead);  16-17: its 4 jobs;  18: stores to LDS.
        SFOR(s, KC4 - KC2)
            constexpr int kc = KC2 + s;
            u32x4_t nh[2], nm[2], nl[2];
            if constexpr (kc + 1 < KC4) {
#ifdef SONET_ABL_NOB
                nh[0] = bh[0]; nm[0] = bm[0]; nh[1] = bh[1]; nm[1] = bm[1];
#else
                nh[0] = PF_ACT(act3w, kc + 1 - KC2, 0, 0); nm[0] = PF_ACT(act3w, kc + 1 - KC2, 0, 1);
                nh[1] = PF_ACT(act3w, kc + 1 - KC2, 1, 0); nm[1] = PF_ACT(act3w, kc + 1 - KC2, 1, 1);
#endif
            }
            constexpr bool j1 = kc >= 8 && kc < 12, j2 = kc >= 16 && kc < 18;      // a layer-1 / layer-2 job pair rides on this step
            constexpr int ck1 = j1 ? kc - 8 : 0, q2 = j2 ? kc - 16 : 0;
            JobSc sc_; JobX jx_;
            if constexpr (kc == 4) { front_load_x(tile + gridDim.x); front_load_w1(); }
            if constexpr (kc == 6) front_split_x();
            if constexpr (j1) PF_LOAD_SC(sc_, LB1 + 32 * (ck1 >> 1) + 16 * (ck1 & 1) + 4 * h)
            if constexpr (j2) PF_LOAD_SC(sc_, LB2 + 32 * wave + 16 * q2 + 4 * h)
#define SLOT_L4B(q) { \
                if constexpr (kc + 1 < KC4) { if constexpr ((q) == 0) nl[0] = piece_l(nh[0]); if constexpr ((q) == 1) nl[1] = piece_l(nh[1]); } \
                if constexpr (j1) PF_SLOT2(q, 6, (ck1 & 1), acc1[ck1 >> 1][0], acc1[ck1 >> 1][1], a1n[ck1 >> 1][ck1 & 1][0], a1n[ck1 >> 1][ck1 & 1][1]) \
                if constexpr (j2) PF_SLOT2(q, 6, q2, acc2[0], acc2[1], a2o[q2][0], a2o[q2][1]) \
            }
            PF_STEP(W4T, KC3 + kc, af[kc % AFD], ACC4, bh[0], bm[0], bl[0], bh[1], bm[1], bl[1], false, SEGMAX, SLOT_L4B)
#undef SLOT_L4B
            if constexpr (kc == 7) front_l1();
            if constexpr (kc >= 12 && kc < 16) front_l2(IC<(kc >= 12 && kc < 16 ? kc - 12 : 0)>{});
            if constexpr (kc >= 11 && kc < 15) front_load_w2(IC<(kc >= 11 && kc < 15 ? kc - 11 : 0)>{});   // (after the chunk before it has been consumed)
            if constexpr (kc == 18) front_store_a2();
            if constexpr (kc + 1 < KC4) { bh[0] = nh[0]; bm[0] = nm[0]; bl[0] = nl[0]; bh[1] = nh[1]; bm[1] = nm[1]; bl[1] = nl[1]; }
        SEND
        PROF_MARK(6)                                            // layer 4, steps 4-19 (+ front of the next tile)
        // ---- epilogue: this wave's 96 channels x 64 points ----
        if constexpr (SEGMAX) {
            // per-node max-pool (replaces index_max + masked gather, models/networks.py:180-185, for the no-grad path:
            // only the VALUES are needed).  Layer 4 of this variant runs with the MFMA operands swapped: the accumulators
            // are TRANSPOSED, acc[mt][c][r] = Y[point 32c + prow(r)][channel 96w + 32mt + j], prow(r) = (r&3) + 8(r>>2) + 4h,
            // so the maximum over a node's points is a maximum over REGISTERS (15 v_max per tile) instead of a cross-lane
            // reduction of every register; the two half-waves (16 points each) meet in one lane exchange.
            float bias4[W4T];
#pragma unroll
            for (int mt = 0; mt < W4T; ++mt) {
                const float2 ss = aff[LB4 + (W4T * wave + mt) * 32 + j];
                bias4[mt] = ss.y;
                if (!l4_unit) {
#pragma unroll
                    for (int c = 0; c < 2; ++c)
#pragma unroll
                        for (int r = 0; r < 16; ++r) acc[mt][c][r] = __fmaf_rn(acc[mt][c][r], ss.x, ss.y);
                }
            }
#pragma unroll
            for (int c = 0; c < 2; ++c) {
                if (jpos0[c] >= 0) {                                              // wave-uniform: features of original copy 0
#pragma unroll
                    for (int r = 0; r < 16; ++r)
                        if ((r & 3) + 8 * (r >> 2) + 4 * h == jpos0[c]) {
#pragma unroll
                            for (int mt = 0; mt < W4T; ++mt) v0[b * C4 + (W4T * wave + mt) * 32 + j] = l4_unit ? __fmaf_rn(acc[mt][c][r], ACC_UNSCALE, bias4[mt]) : acc[mt][c][r];
                        }
                }
                // per node present in this column tile (usually 1, 2 at a node boundary; ids are sorted, so a node's
                // points are the rows [s, e)): max over its rows (v_max_f32 ignores a NaN operand, as the reference's
                // '>' does), published by integer atomicMax on orderable keys -- to the LDS bins of the tile's
                // first SEG_SLOTS nodes, else straight to memory.  All branches are wave-uniform.
                unsigned remaining = (unsigned)__ballot(pv[c]);                   // lanes 0..31 <-> the column tile's 32 points
                while (remaining != 0u) {
                    const int s0 = __builtin_ctz(remaining);
                    const int node = __builtin_amdgcn_readlane(nid[c], s0);
                    const unsigned segmask = (unsigned)__ballot(pv[c] && nid[c] == node);
                    remaining &= ~segmask;
                    const int e0 = s0 + __builtin_popcount(segmask);
                    const bool whole = (s0 == 0 && e0 == 32);
                    const int slot = node - n0;
                    float mx[W4T];
                    if (whole) {
#pragma unroll
                        for (int mt = 0; mt < W4T; ++mt) {
                            float m = acc[mt][c][0];
#pragma unroll
                            for (int r = 1; r < 16; ++r) asm("v_max_f32 %0, %1, %2" : "=v"(m) : "v"(m), "v"(acc[mt][c][r]));
                            mx[mt] = m;
                        }
                    } else {
#pragma unroll
                        for (int mt = 0; mt < W4T; ++mt) mx[mt] = -__builtin_inff();
#pragma unroll
                        for (int r = 0; r < 16; ++r) {
                            const int prow = (r & 3) + 8 * (r >> 2) + 4 * h;
                            const bool in = prow >= s0 && prow < e0;
#pragma unroll
                            for (int mt = 0; mt < W4T; ++mt) {
                                const float v = in ? acc[mt][c][r] : -__builtin_inff();
                                asm("v_max_f32 %0, %1, %2" : "=v"(mx[mt]) : "v"(mx[mt]), "v"(v));
                            }
                        }
                    }
                    if (l4_unit) {                                                // fl(x / 32 + b) is monotone in x: after the max
#pragma unroll
                        for (int mt = 0; mt < W4T; ++mt) mx[mt] = __fmaf_rn(mx[mt], ACC_UNSCALE, bias4[mt]);
                    }
                    // two explicit paths: a generic pointer here would make FLAT atomics (and FLAT operations count on
                    // both vmcnt and lgkmcnt)
                    if (slot < SEG_SLOTS) {
#pragma unroll
                        for (int mt = 0; mt < W4T; ++mt) atomicMax(&bins[slot][(W4T * wave + mt) * 32 + j], ord_f32(__float_as_uint(mx[mt])));
                    } else {
                        unsigned *gdst = pooled + ((long long)b * M + node) * C4 + (W4T * wave) * 32;
#pragma unroll
                        for (int mt = 0; mt < W4T; ++mt) atomicMax(gdst + 32 * mt + j, ord_f32(__float_as_uint(mx[mt])));
                    }
                }
            }
            pend_n = nslots * C4;
            pend = partial + (tile * SEG_SLOTS) * (long long)C4;                   // stored after the next barrier 1
        } else {
            const __amdgpu_buffer_rsrc_t ry = __builtin_amdgcn_make_buffer_rsrc(
                y + b * (long long)C4 * L, 0, (int)((unsigned)C4 * rowB), 0x00020000);
#pragma unroll
            for (int c = 0; c < 2; ++c) {
                if (!pv[c]) continue;
#pragma unroll
                for (int mt = 0; mt < W4T; ++mt) {
                    const int ct = W4T * wave + mt;
                    const unsigned so_tile = (unsigned)(ct * 32) * rowB;
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int orow = (r & 3) + 8 * (r >> 2);
                        const float2 ss = aff[LB4 + ct * 32 + orow + 4 * h];
                        const float v = __fmaf_rn(acc[mt][c][r], ss.x, ss.y);
                        __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(int, v), ry, (unsigned)(4 * h * L + lc[c]) * 4u,
                                                              so_tile + (unsigned)orow * rowB, 0);
                    }
                }
            }
        }
        PROF_MARK(7)                                            // epilogue (pool or stores)
    }
    if constexpr (SEGMAX) {
        __syncthreads();
        for (int e = threadIdx.x; e < pend_n; e += PF_THREADS) pend[e] = (&bins[0][0])[e];
    }
    if (rlog != nullptr) {
        range_publish(rlog, wave_umax(xin_r), lane);
        // the jobs logged 32 x: take the factor out of the exponent (a NaN / inf stays far above the fp16 range)
        unsigned rb = wave_umax((unsigned)(rmax_ > 0 ? rmax_ : 0));
        rb = rb > (5u << 23) ? rb - (5u << 23) : 0u;
        range_publish(rlog + 2, rb, lane);
        if (blockIdx.x == 0 && threadIdx.x == 0) atomicMax(rlog + 1, reinterpret_cast<const unsigned *>(Wst + (long long)NSLICE * 64)[0]);
    }
    PROF_MARK(8)
    PROF_DUMP
}

__global__ __launch_bounds__(256) void pooled_init_kernel(unsigned *__restrict__ pooled, long long n) {
    const long long t = (long long)blockIdx.x * 256 + threadIdx.x;
    if (t < n) pooled[t] = SEG_INIT;
}

// Second kernel of the pooled path: out[b][c][m] = max over the tiles that hold copies of node m of that tile's
// partial (slot = m - first node of the tile), combined with the rare straight-to-memory fallback in `pooled`
// (tiles spanning more than SEG_SLOTS nodes).  Nodes that never beat -1000 (empty, or all values <= -1000) take the
// features of original point copy 0, as gather index 0 does in the reference (models/networks.py:185).
__global__ __launch_bounds__(256) void pooled_decode_kernel(const unsigned *__restrict__ pooled, const unsigned *__restrict__ partial,
                                                             const int32_t *__restrict__ ids_sorted, const int32_t *__restrict__ node_off,
                                                             const int32_t *__restrict__ count, const float *__restrict__ v0,
                                                             float *__restrict__ out, int M, int L, int tpc, long long total)
{
    const long long t = (long long)blockIdx.x * 256 + threadIdx.x;      // over [B][M][384], c fastest (coalesced partial reads)
    if (t >= total) return;
    const int c = (int)(t % C4);
    const long long bm = t / C4;
    const int m = (int)(bm % M);
    const long long b = bm / M;
    unsigned key = pooled[t];
    const int cnt = count[b * M + m];
    if (cnt > 0) {
        const int off = node_off[b * M + m];
        for (int tl = off / TPTS; tl <= (off + cnt - 1) / TPTS; ++tl) {
            const int slot = m - ids_sorted[b * L + tl * TPTS];
            if (slot < SEG_SLOTS) {
                const unsigned k2 = partial[((b * tpc + tl) * SEG_SLOTS + slot) * (long long)C4 + c];
                key = k2 > key ? k2 : key;
            }
        }
    }
    float v;
    if (key > SEG_INIT) v = __uint_as_float((key & 0x80000000u) ? (key ^ 0x80000000u) : ~key);
    else v = v0[b * C4 + c];
    out[(b * C4 + c) * M + m] = v;
}

int cu_count() {
    int dev = 0, cus = 256;
    if (hipGetDevice(&dev) == hipSuccess) {
        int v = 0;
        if (hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && v > 0) cus = v;
    }
    if (const char *e = sonet::knob("SONET_FUSED_MAXCU")) { const int v = atoi(e); if (v > 0 && v < cus) cus = v; }   // (variants build only)
    return cus;
}

}  // namespace

extern "C" size_t sonet_pointresnet_pack_size(void) { return (size_t)NSLICE * 1024 + 64; }   // + trailer: word 0 = bits of max |w|

extern "C" int sonet_pointresnet_pack(const float *W1, const float *W2, const float *W3, const float *W4, int Cin0,
                                      void *stream_out, sonet_stream_t stream)
{
    const char *what = "sonet_pointresnet_pack";
    SONET_REQUIRE(W1 && W2 && W3 && W4 && stream_out, "%s: NULL pointer", what);
    SONET_REQUIRE(Cin0 >= 1 && Cin0 <= 16, "%s: Cin0=%d must be in [1, 16]", what, Cin0);
    unsigned *trailer = reinterpret_cast<unsigned *>(reinterpret_cast<char *>(stream_out) + (size_t)NSLICE * 1024);
    if (hipMemsetAsync(trailer, 0, 64, sonet::as_stream(stream)) != hipSuccess) return sonet::fail(SONET_ERR_LAUNCH, "%s: memset failed", what);
    hipLaunchKernelGGL(pointresnet_pack_kernel, dim3(sonet::ceil_div(NSLICE * 64, 256)), dim3(256), 0, sonet::as_stream(stream),
                       W1, W2, W3, W4, Cin0, reinterpret_cast<uint4 *>(stream_out), trailer);
    return sonet::launched(what);
}

extern "C" int sonet_pointresnet_fused_f32(const float *x, int Cin0, const void *wstream, const float *affine,
                                           float *y, int B, int L, sonet_stream_t stream)
{
    const char *what = "sonet_pointresnet_fused_f32";
    SONET_REQUIRE(x && wstream && affine && y, "%s: NULL pointer", what);
    SONET_REQUIRE(B > 0 && L > 0 && Cin0 >= 1 && Cin0 <= 16, "%s: bad size B=%d L=%d Cin0=%d", what, B, L, Cin0);
    if ((double)C4 * L * 4.0 >= 4.0e9) return sonet::fail(SONET_ERR_UNSUPPORTED, "%s: a per-cloud panel exceeds 4 GiB", what);
    const int tpc = sonet::ceil_div(L, TPTS);
    const long long ntiles = (long long)B * tpc;
    const int cus = cu_count();
    const long long grid = ntiles < cus ? ntiles : cus;        // persistent: one workgroup per CU
    hipLaunchKernelGGL((pointresnet_fused_kernel<false>), dim3((unsigned)grid), dim3(PF_THREADS), 0, sonet::as_stream(stream),
                       x, Cin0, reinterpret_cast<const u32x4_t *>(wstream), reinterpret_cast<const float2 *>(affine), y, L, tpc, ntiles,
                       (const int32_t *)nullptr, (const int32_t *)nullptr, (unsigned *)nullptr, (float *)nullptr, 0, (unsigned *)nullptr, sonet::range_log());
    return sonet::launched(what);
}

extern "C" size_t sonet_pointresnet_pool_ws_size(int B, int L, int M)
{
    if (B <= 0 || L <= 0 || M <= 0) return 0;
    const long long ntiles = (long long)B * sonet::ceil_div(L, TPTS);
    return (size_t)((long long)B * M * C4 + ntiles * SEG_SLOTS * C4) * 4 + (size_t)B * C4 * 4;
}

extern "C" int sonet_pointresnet_fused_pool_f32(const float *x_sorted, int Cin0, const void *wstream, const float *affine,
                                                const int32_t *ids_sorted, const int32_t *pos0, const int32_t *node_off,
                                                const int32_t *count, void *ws, float *out, int B, int L, int M, sonet_stream_t stream)
{
    const char *what = "sonet_pointresnet_fused_pool_f32";
    SONET_REQUIRE(x_sorted && wstream && affine && ids_sorted && pos0 && node_off && count && ws && out, "%s: NULL pointer", what);
    SONET_REQUIRE(B > 0 && L > 0 && M > 0 && Cin0 >= 1 && Cin0 <= 16, "%s: bad size B=%d L=%d M=%d Cin0=%d", what, B, L, M, Cin0);
    hipStream_t st = sonet::as_stream(stream);
    const long long npool = (long long)B * M * C4;
    const int tpc = sonet::ceil_div(L, TPTS);
    const long long ntiles = (long long)B * tpc;
    unsigned *pooled_ws = reinterpret_cast<unsigned *>(ws);
    unsigned *partial_ws = pooled_ws + npool;
    float *v0_ws = reinterpret_cast<float *>(partial_ws + ntiles * SEG_SLOTS * C4);
    hipLaunchKernelGGL(pooled_init_kernel, dim3((unsigned)sonet::ceil_div64(npool, 256)), dim3(256), 0, st, pooled_ws, npool);
    const int cus = cu_count();
    const long long grid = ntiles < cus ? ntiles : cus;
    hipLaunchKernelGGL((pointresnet_fused_kernel<true>), dim3((unsigned)grid), dim3(PF_THREADS), 0, st,
                       x_sorted, Cin0, reinterpret_cast<const u32x4_t *>(wstream), reinterpret_cast<const float2 *>(affine), (float *)nullptr,
                       L, tpc, ntiles, ids_sorted, pos0, pooled_ws, v0_ws, M, partial_ws, sonet::range_log());
    hipLaunchKernelGGL(pooled_decode_kernel, dim3((unsigned)sonet::ceil_div64(npool, 256)), dim3(256), 0, st, pooled_ws, partial_ws,
                       ids_sorted, node_off, count, v0_ws, out, M, L, tpc, npool);
    return sonet::launched(what);
}

#ifdef SONET_PROF
extern "C" int sonet_prof_read(long long *host, int n)
{
    return hipMemcpyFromSymbol(host, HIP_SYMBOL(g_prof), sizeof(long long) * (size_t)n, 0, hipMemcpyDeviceToHost) == hipSuccess ? 0 : 1;
}
#endif
