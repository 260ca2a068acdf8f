// pointresnet_fused.hip -- the whole first PointNet of the encoder as ONE kernel (eval mode):
// 64 points per wave, every weight fragment read from LDS feeds TWO MFMAs, activations pre-split once, and the accumulation
// registers owned by hand.
//
// Replaces the four EquivariantLayer launches of PointResNet.forward (models/layers.py:419-432, built at
// models/networks.py:82-83 as 6 -> 64 -> 128 -> 256 -> [64 + 256] -> 384 with BN + ReLU on the first three layers)
// when BatchNorm runs on its running statistics.  HBM sees only the 6-channel input and the 384-channel output (or,
// pool variant, only the per-node maxima: models/networks.py:175-185).
//
// Arithmetic: fp32 operands split into fp16 pieces, three v_mfma_f32_32x32x16_f16 per product set with fp32
// accumulation ("operand split" below).  Per accumulator the MFMA sequence (K chunk order, term order l, m, h) is the
// one of the second-generation kernel and every product differs from that kernel's by an exact factor 32, so the results are
// bit-identical to it.
//
// What limits this kernel is the traffic of the WEIGHT STREAM per point, not the matrix pipe (measured: the same MFMAs with the
// weight requests removed run in two thirds of the time, at a higher clock).  So a workgroup = 4 waves streams the weights ONCE
// for 256 points: every wave owns 64 points (column tiles c = 0, 1) through all four layers, the stream goes through a 3-slot
// LDS ring by LDS-DMA (one barrier per 72 MFMAs), and each fragment read from LDS feeds both column tiles.
//
// Registers.  64 points x 256 channels of layer-3 output are 256 registers per lane on their own, and hipcc cannot place that
// beside everything else (every attempt spilled 200+ registers: it keeps VALU-produced MFMA operands in the 256 arch VGPRs).
// The accumulation registers a[0:255] are therefore OWNED BY HAND (inline asm with literal register numbers): the accumulators of
// layers 1-3 live there, and when a layer-3 tile is complete it is turned IN PLACE into the fp16 pieces (32 xh, fp16(32 xm)) that
// are the B operand of layer 4 -- the same 32 bits per value.  The compiler keeps the 256 arch VGPRs: layer 4's accumulators (96),
// the fragments, the pieces of layers 1 / 2.  Every MFMA is an asm statement; hipcc knows nothing about a[...] and pads no hazard
// inside or behind an asm statement, so the build is audited (tools/check_fused_asm.py, run by the CPU tests and by build()):
// no spill, no compiler access to the accumulation file, nothing touches an MFMA's VGPR destination while it is in flight.
//   a[  0: 63]  layer-1 accumulators (2 tiles x 2 column tiles)      dead when layer 2 is done
//   a[128:255]  layer-2 accumulators (4 x 2)                          their jobs end during layer 3's first tile group
//   a[  0:255]  layer-3 accumulators (8 x 2), then its pre-split output, live through layer 4
// Activations are split ONCE ("jobs": BatchNorm affine + ReLU + split, placed 4-5 at a time behind the MFMAs of later steps).
// A lane holds 64 + 128 + 256 activation values per tile and has 512 registers: during layer 3, half of layer 2's output waits in
// a 64 KiB lane-private LDS park, and layer 1 (12 of 1944 MFMAs per tile and wave) is simply run a second time after layer 3, its output parked in
// the same space for the four passes of layer 4.
//
// Register chaining.  v_mfma_f32_32x32x16_f16 produces D[row = (r&3) + 8*(r>>2) + 4*(l>>5)][col = l&31] in register r
// of lane l and consumes B[k = 8*(l>>5) + e][col = l&31], e = 0..7.  Registers 8q .. 8q+7 of an output tile therefore
// ARE the B operand of a 16-channel chunk of the next layer, provided the next layer's weights are packed with the
// matching channel order     k = 8h + e   <->   channel 32*t + 16*q + (e&3) + 8*(e>>2) + 4*h     ("chained" packing).
//
// Weights.  All four layers are packed (pointresnet_pack_kernel) into ONE linear stream of 1-KiB slices (64 lanes x
// 8 fp16) in exactly the order the MFMAs consume them.  A STEP = one 16-channel K chunk x NT output tiles x 2 column
// tiles = 6 NT MFMAs fed by 2 NT slices; a stage = 24 slices.
//   L1: 1 step (NT 2),  L2: 4 steps (NT 4),  L3: 2 tile groups x 8 steps (NT 4),  L4: 4 passes x 20 steps (NT 3).
// Workgroups are persistent (one per CU) and walk the 256-point tiles; the weight stream simply restarts (every tile
// is 27 whole stages), and every step -- across layers, passes and tiles -- reads the NEXT step's fragments.
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
constexpr int WPTS = 64, TPTS = PF_WAVES * WPTS;              // points per wave / per workgroup tile
constexpr int T0 = 2, T1 = 4, T2 = 8, T3 = 12;               // output tiles (x32 channels) of the four layers
constexpr int KC2 = 2 * T0, KC3 = 2 * T1, KC4 = 2 * T0 + 2 * T2;   // 16-channel K chunks per layer
constexpr int MT4 = 3, NPASS = T3 / MT4;                      // layer-4 cout tiles per accumulator pass
constexpr int GS = 4;                                          // cout tiles per step in layers 2 and 3
constexpr int NSTG = 24;                                       // slices per LDS stage
constexpr int NTERM = 2;                                       // W slices per (cout tile, K chunk): h and l
constexpr int SL1 = 8 /* 4 used + 4 pad */, SL2 = KC2 * GS * NTERM, SL3 = (T2 / GS) * KC3 * GS * NTERM;
constexpr int OFF1 = 0, OFF2 = SL1, OFF3 = SL1 + SL2, PRE = SL1 + SL2 + SL3;
constexpr int SL4 = KC4 * MT4 * NTERM;                         // slices per layer-4 pass
constexpr int NSLICE = PRE + NPASS * SL4;
constexpr int NSTAGE = NSLICE / NSTG;
static_assert(PRE % NSTG == 0 && SL4 % NSTG == 0 && NSLICE % NSTG == 0, "layer 4 and every pass start on a stage boundary");
static_assert(NSTG % (NTERM * GS) == 0 && NSTG % (NTERM * MT4) == 0 && OFF2 % (NTERM * GS) == 0 && OFF3 % (NTERM * GS) == 0 && T1 == GS && T2 % GS == 0,
              "a step never straddles a stage boundary");
constexpr int NSW = NSTG / PF_WAVES;                           // slices staged per wave
static_assert(NSW == 6, "three DMA statements per wave and stage");
constexpr int CH_TOTAL = 32 * (T0 + T1 + T2 + T3);
constexpr int LB1 = 0, LB2 = 32 * T0, LB3 = 32 * (T0 + T1), LB4 = 32 * (T0 + T1 + T2);

// ---- fp32 -> 3 x fp16 operand split ------------------------------------------------------------------
// x = xh + xm exactly, xh = fp16(x) (11 significand bits), xm the residual; a product is taken as
//     1024 a*b  =  (32 ah) * (32 bh)  +  (32 ah) * fp16(32 bm)  +  fp16(32 am) * (32 bh)
// i.e. THREE fp16 MFMAs with fp32 accumulation (the dropped am*bm and the rounding of the scaled residuals are
// <= 2^-22 relative).  The factors 32 keep the residuals out of the fp16 subnormals; they ride in the ACCUMULATOR (1024 = 2^10;
// every power-of-two scaling is exact, so the bits are those of the unscaled sum) and leave through the layer's affine.  With
// both main pieces carrying their 32 there are only TWO forms of each operand: the stream holds two slices per (cout tile, K
// chunk), 32 ah and fp16(32 am), an activation is two pieces, 32 bh and fp16(32 bm) -- nothing is derived at the point of use
// (a third piece bh = (32 bh) * 2^-5 computed per step cost 2.4 % of the kernel).
// Operand range: |x| <= 2047 and |w| <= 2047 (32 x and 32 w must fit fp16); the split clamps and the range log reports both.
// Naming: B side pieces h = 32 bh, m = fp16(32 bm); A side slices h = 32 ah, l = fp16(32 am);
// terms in accumulation order: (A.l, B.h), (A.h, B.m), (A.h, B.h).
constexpr float F16_MAX = 2047.0f;                             // 32 * 2047 = 65504, the largest fp16
constexpr float F16_MAX32 = 65504.0f;
constexpr float ACC_UNSCALE = 0.0009765625f;                   // accumulators hold 1024 * (W . x)
constexpr float ACC_TO_X32 = 0.03125f;                         // ... and 32 x = acc * (scale / 32) + 32 shift
__device__ __forceinline__ unsigned cvt_pk_f16(float lo, float hi) {        // one v_cvt_pk_f16_f32 (round to nearest even)
    const f32x2_t v = {lo, hi};
    return __builtin_bit_cast(unsigned, __builtin_convertvector(v, f16x2_t));
}
__device__ __forceinline__ float f16_lo(unsigned pk) { return (float)__builtin_bit_cast(f16x2_t, pk)[0]; }
__device__ __forceinline__ float f16_hi(unsigned pk) { return (float)__builtin_bit_cast(f16x2_t, pk)[1]; }
__device__ __forceinline__ float clamp_f16(float x) { return __builtin_fminf(__builtin_fmaxf(x, -F16_MAX), F16_MAX); }
__device__ __forceinline__ f16x8 as_f16x8(u32x4_t v) { return __builtin_bit_cast(f16x8, v); }
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
// of them behind each MFMA.  The accumulators hold 1024 W.x and the coefficients are (scale / 32, 32 shift): the affine
// delivers 32 x directly (bit-identical to 32 * fl(acc * scale/1024 + shift): power-of-two scalings commute with the
// rounding), ReLU and the fp16 range clamp are one v_med3 against 32 * 2047.
__device__ __forceinline__ float pin_fma(float a, float s, float b) { float r; asm volatile("v_fma_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(s), "v"(b)); return r; }
__device__ __forceinline__ float pin_relu_clamp32(float a) { float r; asm volatile("v_med3_f32 %0, %1, 0, %2" : "=v"(r) : "v"(a), "v"(F16_MAX32)); return r; }
__device__ __forceinline__ unsigned pin_cvt(float lo, float hi) { unsigned r; asm volatile("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(r) : "v"(lo), "v"(hi)); return r; }
// x32 - (fp16 half of pk = 32 xh) = 32 * (x - xh), exact
__device__ __forceinline__ float pin_res_lo(unsigned pk, float x32) { float r; asm volatile("v_fma_mix_f32 %0, %1, %2, %3 op_sel_hi:[1,0,0]" : "=v"(r) : "v"(pk), "v"(-1.0f), "v"(x32)); return r; }
__device__ __forceinline__ float pin_res_hi(unsigned pk, float x32) { float r; asm volatile("v_fma_mix_f32 %0, %1, %2, %3 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(r) : "v"(pk), "v"(-1.0f), "v"(x32)); return r; }
// range log: max over the post-affine, pre-clamp activations (x 32) as signed-int-ordered bits (positive side: what ReLU keeps)
__device__ __forceinline__ int pin_max3_i32(int m, float a, float b) { int r; asm volatile("v_max3_i32 %0, %1, %2, %3" : "=v"(r) : "v"(m), "v"(a), "v"(b)); return r; }

// ---- the accumulation registers, by literal number --------------------------------------------------------------------------------
template <int N> __device__ __forceinline__ float agpr_read() { float r; asm volatile("v_accvgpr_read_b32 %0, a[%c1]" : "=v"(r) : "i"(N)); return r; }
template <int N> __device__ __forceinline__ void agpr_write(unsigned v) { asm volatile("v_accvgpr_write_b32 a[%c0], %1" :: "i"(N), "v"(v)); }
// accumulators of layers 1-3 in a[N:N+15]: D = A . B + (ZERO ? 0 : D).  (s_nop 1: the B operand may be a VGPR a VALU instruction of
// the compiler's has just written -- hipcc pads nothing for the inside of an asm statement.)
template <int N, bool ZERO> __device__ __forceinline__ void mfma_agpr(f16x8 a, f16x8 b) {
    if constexpr (ZERO) asm volatile("s_nop 1\n\tv_mfma_f32_32x32x16_f16 a[%c2:%c3], %0, %1, 0" :: "v"(a), "v"(b), "i"(N), "i"(N + 15));
    else asm volatile("s_nop 1\n\tv_mfma_f32_32x32x16_f16 a[%c2:%c3], %0, %1, a[%c2:%c3]" :: "v"(a), "v"(b), "i"(N), "i"(N + 15));
}
// layer 4: accumulator in VGPRs (the compiler's), the activation operand either a VGPR piece or a[N:N+3]; SWAP: the activation is
// the A operand (the accumulator tile comes out transposed: rows = points)
template <bool ZERO, bool SWAP> __device__ __forceinline__ void mfma_vv(f32x16 &acc, f16x8 w, f16x8 x) {
    if constexpr (ZERO) {
        if constexpr (SWAP) asm volatile("s_nop 1\n\tv_mfma_f32_32x32x16_f16 %0, %2, %1, 0" : "=&v"(acc) : "v"(w), "v"(x));
        else asm volatile("s_nop 1\n\tv_mfma_f32_32x32x16_f16 %0, %1, %2, 0" : "=&v"(acc) : "v"(w), "v"(x));
    } else {
        if constexpr (SWAP) asm volatile("s_nop 1\n\tv_mfma_f32_32x32x16_f16 %0, %2, %1, %0" : "+v"(acc) : "v"(w), "v"(x));
        else asm volatile("s_nop 1\n\tv_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(acc) : "v"(w), "v"(x));
    }
}
template <int N, bool SWAP> __device__ __forceinline__ void mfma_va(f32x16 &acc, f16x8 w) {
    if constexpr (SWAP) asm volatile("v_mfma_f32_32x32x16_f16 %0, a[%c2:%c3], %1, %0" : "+v"(acc) : "v"(w), "i"(N), "i"(N + 3));
    else asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, a[%c2:%c3], %0" : "+v"(acc) : "v"(w), "i"(N), "i"(N + 3));
}

struct JobSc { float2 sc[8]; };                                // (scale, 32 shift) of the job's 8 channels
struct JobOut { u32x4_t h, m; };
struct JobX { float x[8]; unsigned h[4], m[4]; };              // the 8 values in flight and their pieces
// A job works on a[R .. R+7] (registers 8Q..8Q+7 of an accumulator tile).  Op list: 8 reads (all of them first: the in-place form
// writes its pieces over the values), then two groups of 18 (values 4g..4g+3, neighbours independent): affine x4, range max x2,
// relu+clamp x4, 32xh x2 (cvt), residual x4, 32xm x2 (cvt); the in-place form ends with 8 writes (h -> a[R..R+3], m -> a[R+4..R+7]).
constexpr int JOB_OPS_V = 44, JOB_OPS_A = 52;
template <int R, bool INPLACE, int I, bool RANGE = true> __device__ __forceinline__ void job_op(JobX &v, const JobSc &s, int &rm) {
    static_assert(I >= 0 && I < (INPLACE ? JOB_OPS_A : JOB_OPS_V), "");
    if constexpr (I < 8) { v.x[I] = agpr_read<R + I>(); }
    else if constexpr (I < 44) {
        constexpr int g = (I - 8) / 18, k = (I - 8) % 18;
        if constexpr (k < 4) { constexpr int e = 4 * g + k; v.x[e] = pin_fma(v.x[e], s.sc[e].x, s.sc[e].y); }
        else if constexpr (k < 6) { constexpr int e = 4 * g + 2 * (k - 4); if constexpr (RANGE) rm = pin_max3_i32(rm, v.x[e], v.x[e + 1]); }
        else if constexpr (k < 10) { constexpr int e = 4 * g + k - 6; v.x[e] = pin_relu_clamp32(v.x[e]); }
        else if constexpr (k < 12) { constexpr int P = 2 * g + (k - 10); v.h[P] = pin_cvt(v.x[2 * P], v.x[2 * P + 1]); }
        else if constexpr (k < 16) { constexpr int e = 4 * g + (k - 12); v.x[e] = (e & 1) ? pin_res_hi(v.h[e >> 1], v.x[e]) : pin_res_lo(v.h[e >> 1], v.x[e]); }
        else { constexpr int P = 2 * g + (k - 16); v.m[P] = pin_cvt(v.x[2 * P], v.x[2 * P + 1]); }
    } else {
        constexpr int w = I - 44;
        if constexpr (w < 4) agpr_write<R + w>(v.h[w]); else agpr_write<R + w>(v.m[w - 4]);
    }
}
template <int R, bool INPLACE, int I0, int I1, bool RANGE = true> __device__ __forceinline__ void job_ops(JobX &v, const JobSc &s, int &rm) {
    if constexpr (I0 < I1 && I0 < (INPLACE ? JOB_OPS_A : JOB_OPS_V)) { job_op<R, INPLACE, I0, RANGE>(v, s, rm); job_ops<R, INPLACE, I0 + 1, I1, RANGE>(v, s, rm); }
}
__device__ __forceinline__ JobOut job_result(const JobX &v) {
    JobOut o;
    o.h[0] = v.h[0]; o.h[1] = v.h[1]; o.h[2] = v.h[2]; o.h[3] = v.h[3];
    o.m[0] = v.m[0]; o.m[1] = v.m[1]; o.m[2] = v.m[2]; o.m[3] = v.m[3];
    return o;
}

// compile-time loop: f(integral_constant<int, i>) for i in [I0, I1)
template <int I0, int I1, class F> __device__ __forceinline__ void static_for(F &&f) {
    if constexpr (I0 < I1) { f(std::integral_constant<int, I0>{}); static_for<I0 + 1, I1>(f); }
}
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
    bool chained = true, valid = true;
    // consumption order: per layer, tile-group major, then K chunk, then tile within the group, then split term
    if (s < OFF2) {                       // L1: 2 tiles x 1 chunk, standard channel order (input comes from memory)
        const int u = s - OFF1; term = u % NTERM; kc = 0; ct = u / NTERM; W = W1; Cin = Cin0; chained = false; valid = u < T0 * NTERM;
    } else if (s < OFF3) {
        const int u = s - OFF2; term = u % NTERM; const int mt = (u / NTERM) % GS; kc = (u / (NTERM * GS)) % KC2;
        ct = (u / (NTERM * GS * KC2)) * GS + mt; W = W2; Cin = 32 * T0;
    } else if (s < PRE) {
        const int u = s - OFF3; term = u % NTERM; const int mt = (u / NTERM) % GS; kc = (u / (NTERM * GS)) % KC3;
        ct = (u / (NTERM * GS * KC3)) * GS + mt; W = W3; Cin = 32 * T1;
    } else {                              // L4: pass-major, then chunk-major, MT4 tiles per chunk
        const int u = (s - PRE) % SL4, pass = (s - PRE) / SL4;
        term = u % NTERM; ct = pass * MT4 + (u / NTERM) % MT4; kc = u / (NTERM * MT4); W = W4; Cin = 32 * (T0 + T2);
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
            range_track(wr, 32.f * v[0], 32.f * v[1]);              // (logged as 32 |w|: the fp16 range applies to that)
            // A-side slices: 0 = fp16(32 w) = 32 wh (terms h and m), 1 = fp16(32 (w - wh)) = fp16(32 w - 32 wh) (term l).  |w| > 2047 is
            // clamped (a finite, wrong product instead of inf - inf = NaN in the residual); the range log above has the true value
            const float w0 = v[0] != v[0] ? v[0] : __builtin_fminf(__builtin_fmaxf(32.f * v[0], -F16_MAX32), F16_MAX32);     // (NaN stays NaN)
            const float w1 = v[1] != v[1] ? v[1] : __builtin_fminf(__builtin_fmaxf(32.f * v[1], -F16_MAX32), F16_MAX32);
            const unsigned hh = cvt_pk_f16(w0, w1);
            const unsigned ll = cvt_pk_f16(w0 - f16_lo(hh), w1 - f16_hi(hh));
            w[p] = term == 0 ? hh : ll;
        }
    }
    out[(long long)s * 64 + lane] = make_uint4(w[0], w[1], w[2], w[3]);
    range_publish(trailer, wave_umax(range_amax_bits(wr)), lane);         // max |w| over the four layers (range log, word 1)
}

// ---- the fused kernel ------------------------------------------------------------------------------
// LDS ring of NSLOT = 3 stages of W (24 KiB each), filled by LDS-DMA.  While stage n is consumed, stage n+1 is
// resident and published (the last step of a stage reads the A fragments of the next stage's first step from it)
// and stage n+2 is landing in the slot stage n-1 occupied.  Boundary(n), run by every wave at the first step of
// stage n:   s_waitcnt vmcnt(0) (this wave's pieces of stage n+1 have landed);  barrier;  issue stage n+2.
constexpr int NSLOT = 3;


struct AF { f16x8 h[GS], l[GS]; };                            // A fragments of one step (up to GS tiles x 2 slices)

// SEGMAX = the per-node max-pool epilogue (see below) instead of the y stores; x must then be node-sorted.
constexpr int SEG_SLOTS = 16;                                 // nodes of a 256-point tile pre-reduced in LDS (the rest: global atomics)
constexpr int PCH = 32 * MT4;                                 // channels per layer-4 pass
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
#define PROF_T0 long long pt_ = __builtin_readcyclecounter();
#define PROF_T1(i) { const long long n_ = __builtin_readcyclecounter(); prof_[i] += n_ - pt_; pt_ = n_; }
#define PROF_DUMP prof_[31] = (long long)__builtin_amdgcn_s_memrealtime() - prof_rt0_; /* 100 MHz constant clock */ if (lane == 0) { for (int i_ = 0; i_ < PROF_N; ++i_) g_prof[(blockIdx.x * PF_WAVES + wave) * PROF_N + i_] = prof_[i_]; }
#else
#define PROF_DECL
#define PROF_MARK(i)
#define PROF_T0
#define PROF_T1(i)
#define PROF_DUMP
#endif

#define PF_SB __builtin_amdgcn_sched_barrier(0);
#define PF_LDA(base, slice) __builtin_bit_cast(f16x8, (base)[(slice) * 64])
// first accumulation register of the accumulator tile (t, c) of layers 1 / 3 (a[0:...]) and of layer 2 (a[128:255])
#define AG13(t, c) ((((t) * 2) + (c)) * 16)
#define AG2(t, c) (128 + (((t) * 2) + (c)) * 16)
// the whole accumulation file belongs to the asm statements of this kernel: naming it once makes the kernel descriptor allocate it
#define PF_A8(n) "a" #n "0", "a" #n "1", "a" #n "2", "a" #n "3", "a" #n "4", "a" #n "5", "a" #n "6", "a" #n "7", "a" #n "8", "a" #n "9"
#define PF_CLAIM_AGPRS asm volatile("" ::: "a0", "a1", "a2", "a3", "a4", "a5", "a6", "a7", "a8", "a9", PF_A8(1), PF_A8(2), PF_A8(3), PF_A8(4), PF_A8(5), \
        PF_A8(6), PF_A8(7), PF_A8(8), PF_A8(9), PF_A8(10), PF_A8(11), PF_A8(12), PF_A8(13), PF_A8(14), PF_A8(15), PF_A8(16), PF_A8(17), PF_A8(18), PF_A8(19), \
        PF_A8(20), PF_A8(21), PF_A8(22), PF_A8(23), PF_A8(24), "a250", "a251", "a252", "a253", "a254", "a255");

// P16OUT (store variant only): y ALSO leaves pre-split, in the P16 planes of csrc/pointmlp_h3p.hip (the segmenter's first layer consumes
// first_pn_out per point copy: its operand loads are then finished MFMA fragments and the separate conversion pass is gone).
template <bool SEGMAX, bool P16OUT = false>
__global__ __launch_bounds__(PF_THREADS, 1) void pointresnet_fused_kernel(
    const float *__restrict__ x, int Cin0, const uint4 *__restrict__ Wst, const float2 *__restrict__ affine_g /*[CH_TOTAL] (scale, shift); last layer (1, bias)*/,
    float *__restrict__ y, int L, int tpc /*256-point tiles per cloud*/, long long ntiles,
    const int32_t *__restrict__ ids_sorted, const int32_t *__restrict__ pos0, unsigned *__restrict__ pooled, float *__restrict__ v0, int M,
    unsigned *__restrict__ partial /*[ntiles][NPASS][SEG_SLOTS][PCH] keys of the tile's first SEG_SLOTS nodes*/,
    unsigned *__restrict__ rlog /*optional range-log slot: [0] max |x in|, [1] max |w|, [2] max post-affine input of layers 2-4 (bits)*/,
    void *__restrict__ yp = nullptr /*P16OUT: [B][24][2][2][L][16 B]*/)
{
    static_assert(!(SEGMAX && P16OUT), "the pooled variant has no per-point output");
    RangeAcc yr4 = {0, 0u};                                    // (P16OUT) magnitudes of the layer-4 output that was split
    __shared__ uint4 wsm[NSLOT * NSTG][64];                    // 3 x 24 KiB
    // 64 KiB lane-private parking space, [wave][chunk][column tile][piece h / m][lane]: chunks 0-3 of layer 2's output between the
    // two tile groups of layer 3, then the four chunks of layer 1's output for the passes of layer 4
    __shared__ u32x4_t park[PF_WAVES][KC2][2][2][64];
    __shared__ uint4 w1s[T0 * NTERM][64];                      // layer 1's weight slices, resident (layer 1 runs twice per tile, see below)
    __shared__ __attribute__((aligned(16))) float2 aff[CH_TOTAL];
    __shared__ unsigned bins[SEGMAX ? SEG_SLOTS : 1][SEGMAX ? PCH : 1];

    PF_CLAIM_AGPRS
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int j = lane & 31, h = lane >> 5;
    bool l4_unit_lane = true;                                   // layer 4 has no BatchNorm in the reference: scale == 1
    for (int c = threadIdx.x; c < CH_TOTAL; c += PF_THREADS) {
        const float2 v = affine_g[c];
        // layers 1-3: the split wants 32 x = acc * scale / 32 + 32 shift (the accumulators carry a factor 1024);
        // layer 4: y = acc * scale / 1024 + shift
        aff[c] = c < LB4 ? make_float2(v.x * ACC_TO_X32, 32.f * v.y) : make_float2(v.x * ACC_UNSCALE, v.y);
        if (c >= LB4 && v.x != 1.0f) l4_unit_lane = false;
    }
    if constexpr (SEGMAX) {
        for (int i = threadIdx.x; i < SEG_SLOTS * PCH; i += PF_THREADS) (&bins[0][0])[i] = SEG_INIT;
    }
    static_assert(T0 * NTERM * 64 == PF_THREADS, "one 16-byte piece of layer 1's slices per thread");
    w1s[threadIdx.x >> 6][threadIdx.x & 63] = Wst[OFF1 * 64 + threadIdx.x];
    const bool l4_unit = __syncthreads_and(l4_unit_lane) != 0;   // then max(x + b) = max(x) + b exactly: bias after the pool
    PROF_DECL

    unsigned vow = (unsigned)lane * 16u;                        // (not const: a const local is not captured by a generic lambda)
    const unsigned rowB = (unsigned)L * 4u;

    // The W stream goes global -> LDS by LDS-DMA (global_load_lds_dwordx4: one 1 KiB slice per wave instruction,
    // lane-linear, which is exactly the slice layout): no staging VGPRs and no ds_write pass.  hipcc does not see
    // these loads; their completion is counted by hand (vmcnt(0) before the barrier that publishes the stage).
    const unsigned wsm_lds = (unsigned)reinterpret_cast<size_t>(&wsm[0][0]);
    const char *dma_g = nullptr;                                // this wave's NSW slices of the stage being streamed
    unsigned dma_dst = 0;
    auto dma_setup = [&](int n, int slot) __attribute__((always_inline)) {   // stream stage n (wraps: the stream restarts per tile)
        const int sn = n % NSTAGE;
        dma_g = reinterpret_cast<const char *>(Wst) + (size_t)(sn * NSTG + wave * NSW) * 1024u;
        dma_dst = wsm_lds + (unsigned)(slot * NSTG + wave * NSW) * 1024u;
    };
    // pieces t and t+1 (t even) of the wave's NSW = 6: they share an M0 / base pair, the instruction offset moves both
    // the global and the LDS address.  (M0 is written in the statement that uses it and not restored: nothing else
    // in this kernel reads it.)  s_nop 4: an SGPR the compiler restored with v_readlane right in front of the statement
    // needs 5 wait states before a VMEM instruction reads it (tools/check_dma_hazard.py).
    auto dma_pair = [&](auto tc) __attribute__((always_inline)) {
        constexpr int t = decltype(tc)::value;
        const char *g = dma_g + (t / 4) * 4096;
        const unsigned d = dma_dst + (unsigned)(t / 4) * 4096u;
        const unsigned vv = vow;                                   // (an asm operand alone does not make a generic lambda capture it)
        if constexpr ((t % 4) == 0) asm volatile("s_mov_b32 m0, %2\n\ts_nop 4\n\tglobal_load_lds_dwordx4 %0, %1\n\tglobal_load_lds_dwordx4 %0, %1 offset:1024" :: "v"(vv), "s"(g), "s"(d) : "memory");
        else                        asm volatile("s_mov_b32 m0, %2\n\ts_nop 4\n\tglobal_load_lds_dwordx4 %0, %1 offset:2048\n\tglobal_load_lds_dwordx4 %0, %1 offset:3072" :: "v"(vv), "s"(g), "s"(d) : "memory");
    };
    auto stage_dma = [&](int n, int slot) __attribute__((always_inline)) {
        dma_setup(n, slot);
        dma_pair(std::integral_constant<int, 0>{}); dma_pair(std::integral_constant<int, 2>{}); dma_pair(std::integral_constant<int, 4>{});
    };
    // ring state (wave-uniform scalars)
    int n_cur = 0;                                              // stage being consumed
    int slot_cur = 0, slot_nxt = 1, slot_fill = 2;
    stage_dma(0, 0);
    stage_dma(1, 1);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();                                            // stages 0 and 1 are published (stage 0 is read cold)
    const uint4 *lds_cur = &wsm[slot_cur * NSTG][lane];
    const uint4 *lds_nxt = &wsm[slot_nxt * NSTG][lane];
    bool first_boundary = true;
    int pend_n = 0;
    // SEGMAX: partial-maxima block whose LDS bins are still to be stored (at the first stage boundary AFTER the pass,
    // between the barrier -- all lanes have published -- and the next W loads: the stores are older than those loads,
    // so the vmcnt wait that the next boundary needs anyway covers them a whole stage later).
    unsigned *pend = partial;
    auto flush_bins = [&]() __attribute__((always_inline)) {
        for (int e = threadIdx.x; e < pend_n; e += PF_THREADS) {         // only the slots the tile's nodes occupy
            unsigned *bp = &bins[0][0] + e;
            pend[e] = *bp;
            *bp = SEG_INIT;
        }
    };
    auto boundary = [&](auto flushc) __attribute__((always_inline)) {
        constexpr bool flush = decltype(flushc)::value;
        PROF_T0
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");        // this wave's pieces of the stage issued one boundary ago have landed
        PROF_T1(28)
        // a bare s_barrier, not __syncthreads(): that one waits for lgkmcnt(0) first, i.e. for the fragment reads the step before has
        // just issued -- an LDS round trip exposed 27 times per tile.  Nothing read from LDS is shared between waves except the
        // stage being published (written by DMA: vmcnt above) and the pool bins (the pass boundaries wait for their atomics).
        if constexpr (SEGMAX && flush) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");                          // (the intrinsic alone does not order the compiler's memory operations)
        PROF_T1(29)
        if (!first_boundary) {                                  // rotate: the stage just finished becomes the fill slot
            n_cur += 1;
            const int t = slot_cur; slot_cur = slot_nxt; slot_nxt = slot_fill; slot_fill = t;
        }
        first_boundary = false;
        lds_cur = &wsm[slot_cur * NSTG][lane];
        lds_nxt = &wsm[slot_nxt * NSTG][lane];
        if constexpr (SEGMAX && flush) flush_bins();
        dma_setup(n_cur + 2, slot_fill);                        // lands during this stage, published at the next boundary
    };

    // ---- one STEP: a 16-channel K chunk x NT output tiles x 2 column tiles = 6 NT MFMAs, hand-scheduled
    // (sched_barrier after every MFMA: with one wave per SIMD nothing else hides a latency):
    //  - on entry af.l / af.h hold this step's fragments (read by the step before) and BL0 / BL1 the derived B piece l;
    //  - a step that opens a stage takes the boundary first and issues its NSW slices between its first MFMAs;
    //  - term order (A.l, B.l), (A.h, B.m), (A.h, B.h) per accumulator; the freed `l` registers take the NEXT step's `l`
    //    fragments after the second term, `h` likewise after the third (they land under the next step's first term);
    //  - `SLOT(q)` places the VALU work of the step (jobs, next step's B piece l) behind MFMA q.
    // PF_STEP_A: layers 1-3, accumulators a[AGB(u, c) ...], B pieces in VGPRs.  PF_STEP_V: layer 4, accumulators ACC(u, c) in VGPRs,
    // B pieces h / m either VGPRs (FROMA false: the parked layer-1 chunk) or a[HN + 128 c', ...] (layer 3's pre-split output).
    AF af;
#define PF_STEP_HEAD(SIDX, SIDXN, FLUSH)                                                                 \
        constexpr int so_ = (SIDX) % NSTG, son_ = (SIDXN) % NSTG;                                        \
        if constexpr (so_ == 0) boundary(std::integral_constant<bool, (FLUSH)>{});                       \
        const uint4 *nb_ = son_ == 0 ? lds_nxt : lds_cur;
#define PF_STEP_A(NT, SIDX, NTN, SIDXN, AGB, BH0, BM0, BH1, BM1, ZERO, SLOT)                             \
    {                                                                                                    \
        PF_STEP_HEAD(SIDX, SIDXN, false)                                                                 \
        const f16x8 bh_[2] = {as_f16x8(BH0), as_f16x8(BH1)}, bm_[2] = {as_f16x8(BM0), as_f16x8(BM1)};    \
        PF_SB                                                                                            \
        SFOR(q, 2 * (NT)) constexpr int c = q / (NT), u = q % (NT);                                      \
            mfma_agpr<AGB(u, c), (ZERO)>(af.l[u], bh_[c]);                                               \
            if constexpr (so_ == 0 && q < 3) dma_pair(std::integral_constant<int, 2 * q>{});             \
            SLOT(q)                                                                                      \
            PF_SB                                                                                        \
        SEND                                                                                             \
        SFOR(q, 2 * (NT)) constexpr int c = q / (NT), u = q % (NT);                                      \
            mfma_agpr<AGB(u, c), false>(af.h[u], bm_[c]);                                                \
            SLOT(2 * (NT) + q)                                                                           \
            PF_SB                                                                                        \
        SEND                                                                                             \
        SFOR(u, NTN) af.l[u] = PF_LDA(nb_, son_ + NTERM * u + 1); SEND                                   \
        PF_SB                                                                                            \
        SFOR(q, 2 * (NT)) constexpr int c = q / (NT), u = q % (NT);                                      \
            mfma_agpr<AGB(u, c), false>(af.h[u], bh_[c]);                                                \
            SLOT(4 * (NT) + q)                                                                           \
            PF_SB                                                                                        \
        SEND                                                                                             \
        SFOR(u, NTN) af.h[u] = PF_LDA(nb_, son_ + NTERM * u); SEND                                       \
        PF_SB                                                                                            \
    }
#define PF_STEP_V(SIDX, SIDXN, ACC, FROMA, HN0, HN1, BH0, BM0, BH1, BM1, ZERO, FLUSH, SLOT)              \
    {                                                                                                    \
        PF_STEP_HEAD(SIDX, SIDXN, FLUSH)                                                                 \
        const f16x8 bh_[2] = {as_f16x8(BH0), as_f16x8(BH1)}, bm_[2] = {as_f16x8(BM0), as_f16x8(BM1)};    \
        constexpr int hn_[2] = {(HN0), (HN1)};                                                           \
        PF_SB                                                                                            \
        SFOR(q, 2 * MT4) constexpr int c = q / MT4, u = q % MT4;                                         \
            if constexpr (FROMA) mfma_va<hn_[c], SEGMAX>(ACC(u, c), af.l[u]);                            \
            else mfma_vv<(ZERO), SEGMAX>(ACC(u, c), af.l[u], bh_[c]);                                    \
            if constexpr (so_ == 0 && q < 3) dma_pair(std::integral_constant<int, 2 * q>{});             \
            SLOT(q)                                                                                      \
            PF_SB                                                                                        \
        SEND                                                                                             \
        SFOR(q, 2 * MT4) constexpr int c = q / MT4, u = q % MT4;                                         \
            if constexpr (FROMA) mfma_va<hn_[c] + 4, SEGMAX>(ACC(u, c), af.h[u]);                        \
            else mfma_vv<false, SEGMAX>(ACC(u, c), af.h[u], bm_[c]);                                     \
            SLOT(2 * MT4 + q)                                                                            \
            PF_SB                                                                                        \
        SEND                                                                                             \
        SFOR(u, MT4) af.l[u] = PF_LDA(nb_, son_ + NTERM * u + 1); SEND                                   \
        PF_SB                                                                                            \
        SFOR(q, 2 * MT4) constexpr int c = q / MT4, u = q % MT4;                                         \
            if constexpr (FROMA) mfma_va<hn_[c], SEGMAX>(ACC(u, c), af.h[u]);                            \
            else mfma_vv<false, SEGMAX>(ACC(u, c), af.h[u], bh_[c]);                                     \
            SLOT(4 * MT4 + q)                                                                            \
            PF_SB                                                                                        \
        SEND                                                                                             \
        SFOR(u, MT4) af.h[u] = PF_LDA(nb_, son_ + NTERM * u); SEND                                       \
        PF_SB                                                                                            \
    }
    // (scale, 32 shift) of the 8 channels of job (tile t of the layer at LB, half Q): channel LB + 32t + 16Q + (e&3) + 8(e>>2) + 4h
#define PF_LOAD_SC(scv, LB, t, Q)                                                                        \
    {                                                                                                    \
        const float4 *ap_ = reinterpret_cast<const float4 *>(&aff[(LB) + 32 * (t) + 16 * (Q) + 4 * h]);  \
        const float4 c0_ = ap_[0], c1_ = ap_[1], c2_ = ap_[4], c3_ = ap_[5];                             \
        scv.sc[0] = make_float2(c0_.x, c0_.y); scv.sc[1] = make_float2(c0_.z, c0_.w);                    \
        scv.sc[2] = make_float2(c1_.x, c1_.y); scv.sc[3] = make_float2(c1_.z, c1_.w);                    \
        scv.sc[4] = make_float2(c2_.x, c2_.y); scv.sc[5] = make_float2(c2_.z, c2_.w);                    \
        scv.sc[6] = make_float2(c3_.x, c3_.y); scv.sc[7] = make_float2(c3_.z, c3_.w);                    \
    }
    // VALU slots of a step that carries TWO jobs (both column tiles of one (tile, half): same coefficients sc_): OPS of the 2 x JOPS
    // operations behind each MFMA from the third on (the coefficient reads need that long).  R0 / R1: first register of the
    // 8 values of each job; INPL: pieces written back in place.
#ifdef SONET_ABL_NOJOB
#define PF_JOBS_ON false
#else
#define PF_JOBS_ON true
#endif
#define PF_SLOT2(q, OPS, JOPS, INPL, R0, R1)                                                             \
    {                                                                                                    \
        if constexpr (PF_JOBS_ON && (q) >= 2 && (OPS) * ((q) - 2) < 2 * (JOPS)) {                                      \
            constexpr int i0_ = (OPS) * ((q) - 2), i1_ = i0_ + (OPS);                                    \
            if constexpr (i0_ < (JOPS)) job_ops<(R0), (INPL), i0_, (i1_ < (JOPS) ? i1_ : (JOPS))>(jx0_, sc_, rmax_); \
            if constexpr (i1_ > (JOPS)) job_ops<(R1), (INPL), (i0_ > (JOPS) ? i0_ - (JOPS) : 0), i1_ - (JOPS)>(jx1_, sc_, rmax_); \
        }                                                                                                \
    }
#define PF_SLOT1(q, OPS, JOPS, INPL, R0)                                                                 \
    {                                                                                                    \
        if constexpr (PF_JOBS_ON && (q) >= 2 && (OPS) * ((q) - 2) < (JOPS)) {                            \
            constexpr int i0_ = (OPS) * ((q) - 2);                                                       \
            job_ops<(R0), (INPL), i0_, i0_ + (OPS)>(jx0_, sc_, rmax_);                                   \
        }                                                                                                \
    }
    PROF_MARK(0)                                                // kernel prologue
    // inputs of a tile, read one tile ahead (in front of the previous tile's last pass): read at the top of the tile,
    // the x / node-id loads put an HBM round trip in front of layer 1
    float xin_n[2][8];
    int nid_n[2] = {-1, -1}, n0_n = 0, nlast_n = 0, pos0_n = 0;
    auto prefetch_tile = [&](long long t) __attribute__((always_inline)) {
        t = t < ntiles ? t : ntiles - 1;                        // unconditional: the registers are dead between layer 1 and here
        const long long bb = t / tpc;
        const int t0 = (int)(t - bb * tpc) * TPTS;
        const __amdgpu_buffer_rsrc_t rxx = __builtin_amdgcn_make_buffer_rsrc(
            const_cast<float *>(x + bb * (long long)Cin0 * L), 0, (int)((unsigned)Cin0 * rowB), 0x00020000);
#pragma unroll
        for (int c = 0; c < 2; ++c) {
            const int ll0 = t0 + wave * WPTS + 32 * c;
            const bool pvv = ll0 + j < L;
            const int lcc = pvv ? ll0 + j : (ll0 < L ? ll0 : 0);
#pragma unroll
            for (int e = 0; e < 8; ++e)                          // rows >= Cin0 are out of range of the descriptor: 0
                xin_n[c][e] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rxx, (unsigned)(8 * h * L + lcc) * 4u, (unsigned)e * rowB, 0));
            if constexpr (SEGMAX) nid_n[c] = pvv ? ids_sorted[bb * (long long)L + ll0 + j] : -1;
        }
        if constexpr (SEGMAX) {
            const int32_t *idb = ids_sorted + bb * (long long)L;
            n0_n = idb[t0];                                                        // first / last node of the workgroup's tile
            nlast_n = idb[(t0 + TPTS - 1 < L ? t0 + TPTS - 1 : L - 1)];
            pos0_n = pos0[bb];
        }
    };
    prefetch_tile(blockIdx.x);
    // range log: running max of the post-affine inputs of layers 2-4 (x 32) and of |network input|, as ordered bit patterns
    int rmax_ = 0;
    unsigned xin_r = 0u;
    u32x4_t *const park_w = &park[wave][0][0][0][lane];           // this lane's 16 bytes of [chunk][column tile][piece]
#define PF_PARK(kc, c, piece) park_w[(((kc) * 2 + (c)) * 2 + (piece)) * 64]

    // cold start: fragments of the very first step (layer 1 of this workgroup's first tile) from stage 0
    SFOR(u, T0) af.l[u] = PF_LDA(lds_cur, NTERM * u + 1); af.h[u] = PF_LDA(lds_cur, NTERM * u); SEND

    for (long long tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const long long b = tile / tpc;
        const int t0w = (int)(tile - b * tpc) * TPTS + wave * WPTS;      // first point of this wave
        bool pv[2]; int lc[2];
#pragma unroll
        for (int c = 0; c < 2; ++c) {
            const int l0 = t0w + 32 * c;
            pv[c] = l0 + j < L;
            lc[c] = pv[c] ? l0 + j : (l0 < L ? l0 : 0);
        }
        // per-node max-pool bookkeeping of this wave's 2 x 32 (node-sorted) points
        int nid[2] = {-1, -1}, n0 = 0, nslots = 0, jpos0[2] = {-1, -1};
        if constexpr (SEGMAX) {
            nid[0] = nid_n[0]; nid[1] = nid_n[1];
            n0 = n0_n;
            nslots = nlast_n - n0_n + 1 < SEG_SLOTS ? nlast_n - n0_n + 1 : SEG_SLOTS;
#pragma unroll
            for (int c = 0; c < 2; ++c) {
                const int p0 = pos0_n - (t0w + 32 * c);
                jpos0[c] = (p0 >= 0 && p0 < 32) ? p0 : -1;
            }
        }
        // ---- layer 1 (slice 0 opens stage 0 of this tile) ----
        u32x4_t xh[2], xm[2];
#pragma unroll
        for (int c = 0; c < 2; ++c) {
#pragma unroll
            for (int p = 0; p < 4; ++p) {
                // (pinned: left to itself hipcc sinks this to the end of the tile and keeps the 16 inputs alive until there)
                unsigned a0, a1;
                asm volatile("v_and_b32 %0, 0x7fffffff, %1" : "=v"(a0) : "v"(xin_n[c][2 * p]));
                asm volatile("v_and_b32 %0, 0x7fffffff, %1" : "=v"(a1) : "v"(xin_n[c][2 * p + 1]));
                asm volatile("v_max3_u32 %0, %0, %1, %2" : "+v"(xin_r) : "v"(a0), "v"(a1));
            }
            split_input(xin_n[c], xh[c], xm[c]);
        }
        PROF_MARK(1)                                            // tile prologue
#define NOSLOT(q)
        PF_STEP_A(T0, OFF1, GS, OFF2, AG13, xh[0], xm[0], xh[1], xm[1], true, NOSLOT)
        // layer-1 output, pre-split: a1[tile][half][column tile]
        JobOut a1[T0][2][2];
        {                                                       // layer transition: chunk 0 of layer 2's input, nothing to overlap with
            asm volatile("s_nop 15" ::: );                      // (the last MFMAs' results: 12 wait states before anything reads them)
            JobSc sc_; JobX jx0_, jx1_;
            PF_LOAD_SC(sc_, LB1, 0, 0)
            job_ops<AG13(0, 0), false, 0, JOB_OPS_V>(jx0_, sc_, rmax_);
            job_ops<AG13(0, 1), false, 0, JOB_OPS_V>(jx1_, sc_, rmax_);
            a1[0][0][0] = job_result(jx0_); a1[0][0][1] = job_result(jx1_);
        }
        PROF_MARK(6)                                            // layer 1
        // ---- layer 2: 4 steps over the K chunks of the layer-1 output, all 4 output tiles (accumulators a[128:255]) ----
        SFOR(k, KC2)
            constexpr int tk = k >> 1, qk = k & 1;              // this step's chunk = a1[tk][qk]
            constexpr int kn = k + 1, tn = (kn >> 1) & 1, qn = kn & 1;   // jobs of this step: chunk k + 1 (k < 3)
            constexpr int sidx = OFF2 + k * NTERM * GS;
            JobSc sc_; JobX jx0_, jx1_;
            if constexpr (k + 1 < KC2) PF_LOAD_SC(sc_, LB1, tn, qn)
#define SLOT_L2(q) { if constexpr (k + 1 < KC2) { PF_SLOT2(q, 4, JOB_OPS_V, false, AG13(tn, 0) + 8 * qn, AG13(tn, 1) + 8 * qn) \
                       if constexpr ((q) == 23) { a1[tn][qn][0] = job_result(jx0_); a1[tn][qn][1] = job_result(jx1_); } } }
            PF_STEP_A(GS, sidx, GS, sidx + NTERM * GS, AG2, a1[tk][qk][0].h, a1[tk][qk][0].m, a1[tk][qk][1].h, a1[tk][qk][1].m, (k == 0), SLOT_L2)
#undef SLOT_L2
        SEND
        // layer-2 output, pre-split
        JobOut a2[T1][2][2];
        {                                                       // layer transition: chunk 0 of layer 3's input
            asm volatile("s_nop 15" ::: );
            JobSc sc_; JobX jx0_, jx1_;
            PF_LOAD_SC(sc_, LB2, 0, 0)
            job_ops<AG2(0, 0), false, 0, JOB_OPS_V>(jx0_, sc_, rmax_);
            job_ops<AG2(0, 1), false, 0, JOB_OPS_V>(jx1_, sc_, rmax_);
            a2[0][0][0] = job_result(jx0_); a2[0][0][1] = job_result(jx1_);
        }
        PROF_MARK(7)                                            // layer 2
        // ---- layer 3: two groups of 4 output tiles x 8 K chunks (accumulators a[0:127], then a[128:255]) ----
#define AG3A(u, c) AG13(u, c)
#define AG3B(u, c) AG13(GS + (u), c)
        u32x4_t pbh[2], pbm[2];                                 // a chunk read back from the park
        SFOR(k, KC3)                                            // group 0; jobs: the layer-2 chunk of the next step
            constexpr int tk = k >> 1, qk = k & 1;
            if constexpr (k + 1 == KC3) {                       // the second group starts with chunk 0 again (parked at step 0)
                pbh[0] = PF_PARK(0, 0, 0); pbm[0] = PF_PARK(0, 0, 1);
                pbh[1] = PF_PARK(0, 1, 0); pbm[1] = PF_PARK(0, 1, 1);
            }
            constexpr int kn = k + 1, tn = (kn >> 1) & 3, qn = kn & 1;
            constexpr int sidx = OFF3 + k * NTERM * GS;
            JobSc sc_; JobX jx0_, jx1_;
            if constexpr (k + 1 < KC3) PF_LOAD_SC(sc_, LB2, tn, qn)
#define SLOT_L3A(q) { if constexpr (k + 1 < KC3) { PF_SLOT2(q, 4, JOB_OPS_V, false, AG2(tn, 0) + 8 * qn, AG2(tn, 1) + 8 * qn) \
                        if constexpr ((q) == 23) { a2[tn][qn][0] = job_result(jx0_); a2[tn][qn][1] = job_result(jx1_); } } }
            PF_STEP_A(GS, sidx, GS, sidx + NTERM * GS, AG3A, a2[tk][qk][0].h, a2[tk][qk][0].m, a2[tk][qk][1].h, a2[tk][qk][1].m, (k == 0), SLOT_L3A)
#undef SLOT_L3A
            if constexpr (k < KC2) {                            // chunks 0-3 wait in LDS for the second tile group, 4-7 in registers
                PF_PARK(k, 0, 0) = a2[tk][qk][0].h; PF_PARK(k, 0, 1) = a2[tk][qk][0].m;
                PF_PARK(k, 1, 0) = a2[tk][qk][1].h; PF_PARK(k, 1, 1) = a2[tk][qk][1].m;
            }
        SEND
        PROF_MARK(8)                                            // layer 3, tiles 0-3
        float xagain[2][8];
        SFOR(k, KC3)                                            // group 1; jobs: the layer-3 outputs of group 0 (tile k>>1, half k&1), in place
            constexpr int tk = k >> 1, qk = k & 1;
            constexpr int sidx = OFF3 + (KC3 + k) * NTERM * GS;
            constexpr int ntn = k + 1 < KC3 ? GS : MT4;         // the last step reads the fragments of layer 4's first step
            constexpr bool cur_park = k < KC2, nxt_park = k + 1 < KC2;
            JobSc sc_; JobX jx0_, jx1_;
            PF_LOAD_SC(sc_, LB3, tk, qk)
            const u32x4_t ch0 = cur_park ? pbh[0] : a2[tk][qk][0].h, cm0 = cur_park ? pbm[0] : a2[tk][qk][0].m;
            const u32x4_t ch1 = cur_park ? pbh[1] : a2[tk][qk][1].h, cm1 = cur_park ? pbm[1] : a2[tk][qk][1].m;
            if constexpr (nxt_park) {
                pbh[0] = PF_PARK(k + 1, 0, 0); pbm[0] = PF_PARK(k + 1, 0, 1);
                pbh[1] = PF_PARK(k + 1, 1, 0); pbm[1] = PF_PARK(k + 1, 1, 1);
            }
            if constexpr (k + 1 == KC3) {                       // the tile's input again (from L2), for the second run of layer 1
                const __amdgpu_buffer_rsrc_t rxx = __builtin_amdgcn_make_buffer_rsrc(
                    const_cast<float *>(x + b * (long long)Cin0 * L), 0, (int)((unsigned)Cin0 * rowB), 0x00020000);
#pragma unroll
                for (int c = 0; c < 2; ++c)
#pragma unroll
                    for (int e = 0; e < 8; ++e)
                        xagain[c][e] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rxx, (unsigned)(8 * h * L + lc[c]) * 4u, (unsigned)e * rowB, 0));
            }
#define SLOT_L3B(q) { PF_SLOT2(q, 5, JOB_OPS_A, true, AG13(tk, 0) + 8 * qk, AG13(tk, 1) + 8 * qk) }
            PF_STEP_A(GS, sidx, ntn, sidx + NTERM * GS, AG3B, ch0, cm0, ch1, cm1, (k == 0), SLOT_L3B)
#undef SLOT_L3B
        SEND
        PROF_MARK(9)                                            // layer 3, tiles 4-7
        // ---- layer 1 once more, for layer 4's first four K chunks (its output could not stay anywhere during layer 3: 64 points x
        // (64 + 128 + 256) channels are more registers than a lane has): same MFMAs, same order, accumulators in VGPRs this time ----
        {
            f32x16 r1[T0][2];
            f16x8 wh[T0], wl[T0];
            SFOR(u, T0) wl[u] = __builtin_bit_cast(f16x8, w1s[NTERM * u + 1][lane]); wh[u] = __builtin_bit_cast(f16x8, w1s[NTERM * u][lane]); SEND
            u32x4_t xh0, xm0, xh1, xm1;
            split_input(xagain[0], xh0, xm0);
            split_input(xagain[1], xh1, xm1);
            const f16x8 bh_[2] = {as_f16x8(xh0), as_f16x8(xh1)}, bm_[2] = {as_f16x8(xm0), as_f16x8(xm1)};
            PF_SB
            SFOR(q, 2 * T0) constexpr int c = q / T0, u = q % T0; mfma_vv<true, false>(r1[u][c], wl[u], bh_[c]); PF_SB SEND
            SFOR(q, 2 * T0) constexpr int c = q / T0, u = q % T0; mfma_vv<false, false>(r1[u][c], wh[u], bm_[c]); PF_SB SEND
            SFOR(q, 2 * T0) constexpr int c = q / T0, u = q % T0; mfma_vv<false, false>(r1[u][c], wh[u], bh_[c]); PF_SB SEND
            asm volatile("s_nop 15" ::: );                      // (MFMA results: 12 wait states before a VALU instruction reads them)
            int rdummy = 0;                                     // (these values went through the range log the first time)
            SFOR(kk, KC2)
                constexpr int tk = kk >> 1, qk = kk & 1;
                JobSc sc_;
                PF_LOAD_SC(sc_, LB1, tk, qk)
                SFOR(c, 2)
                    JobX jx_;
                    SFOR(i, 8) jx_.x[i] = r1[tk][c][8 * qk + i]; SEND
                    job_ops<0, false, 8, JOB_OPS_V, false>(jx_, sc_, rdummy);
                    const JobOut o = job_result(jx_);
                    PF_PARK(kk, c, 0) = o.h; PF_PARK(kk, c, 1) = o.m;
                    if constexpr (kk == 0) { pbh[c] = o.h; pbm[c] = o.m; }
                SEND
            SEND
        }
        PROF_MARK(2)                                            // layer 1 again
        // ---- layer 4: NPASS passes x KC4 steps of MT4 output tiles; chunks 0-3 = layer-1 output (park), 4-19 = layer 3 (a[...]) ----
        // Pass 0 carries the in-place jobs of layer 3's second tile group (one per step, steps 0-15: tile 4 + s/4, half (s/2)&1,
        // column tile s&1 -- tile 4 is complete when step 12 needs it); passes 1-3 run the same steps without jobs.
        // (y == nullptr -- the P16-only store variant: a descriptor of zero bytes, every f32 store falls outside it and writes nothing)
        const __amdgpu_buffer_rsrc_t ry = __builtin_amdgcn_make_buffer_rsrc(
            y ? y + b * (long long)(32 * T3) * L : const_cast<float *>(x), 0, y ? (int)((unsigned)(32 * T3) * rowB) : 0, 0x00020000);
        const __amdgpu_buffer_rsrc_t ryp = __builtin_amdgcn_make_buffer_rsrc(
            P16OUT ? static_cast<char *>(yp) + b * (long long)(2 * T3) * 64 * L : nullptr, 0, P16OUT ? (int)((unsigned)(2 * T3) * 64u * (unsigned)L) : 0, 0x00020000);
#define ACC4(u, c) acc[u][c]
#define PF_L4_BODY(JOBS)                                                                                 \
            SFOR(kc, KC4)                                                                                \
                constexpr int sidx = PRE + kc * NTERM * MT4;                                             \
                constexpr int kn = (kc + 1) % KC4;                                                       \
                constexpr bool cur_park = kc < KC2, nxt_park = kn < KC2;                                 \
                constexpr int t3 = cur_park ? 0 : (kc - KC2) >> 1, q3 = cur_park ? 0 : (kc - KC2) & 1;   \
                constexpr int jt = T2 / 2 + kc / 4, jq = (kc >> 1) & 1, jc = kc & 1;                     \
                constexpr bool job = (JOBS) && kc < 16;                                                  \
                JobSc sc_; JobX jx0_;                                                                    \
                if constexpr (job) PF_LOAD_SC(sc_, LB3, (job ? jt : 0), jq)                              \
                const u32x4_t ch0 = pbh[0], cm0 = pbm[0], ch1 = pbh[1], cm1 = pbm[1];                    \
                if constexpr (nxt_park) {                        /* the step reads this chunk's registers first (copies above) */ \
                    pbh[0] = PF_PARK(kn, 0, 0); pbm[0] = PF_PARK(kn, 0, 1);                              \
                    pbh[1] = PF_PARK(kn, 1, 0); pbm[1] = PF_PARK(kn, 1, 1);                              \
                }                                                                                        \
                PF_STEP_V(sidx, sidx + NTERM * MT4, ACC4, !cur_park, AG13(t3, 0) + 8 * q3, AG13(t3, 1) + 8 * q3, ch0, cm0, ch1, cm1, (kc == 0), (kc == 0), SLOT_L4) \
            SEND
#define SLOT_L4(q) { if constexpr (job) { PF_SLOT1(q, 4, JOB_OPS_A, true, AG13((job ? jt : 0), jc) + 8 * jq) } }
        // the epilogue of a layer-4 pass
        auto epilogue = [&](f32x16 (&acc)[MT4][2], const int pass) __attribute__((always_inline)) {
#ifdef SONET_ABL_NOEPI
            if (M != 12345) return;
#endif
            asm volatile("s_nop 15" ::: );                      // the last MFMAs' results: 12 wait states before the compiler's code reads them
            if constexpr (SEGMAX) {
                // ---- per-node max-pool of this pass's 96 channels (replaces index_max + masked gather,
                //      models/networks.py:180-185, for the no-grad path: only the VALUES are needed) ----
                // Layer 4 of this variant runs with the MFMA operands swapped: the accumulators are TRANSPOSED,
                // acc[mt][c][r] = Y[point 32c + prow(r)][channel 32 mt + j] with prow(r) = (r&3) + 8 (r>>2) + 4 h, so the
                // maximum over a node's points is a maximum over REGISTERS (15 v_max per tile) instead of a cross-lane
                // reduction of every register; the two half-waves (16 points each) meet in the LDS atomic.
                float bias4[MT4];
#pragma unroll
                for (int mt = 0; mt < MT4; ++mt) {
                    const float2 ss = aff[LB4 + (pass * MT4 + mt) * 32 + j];
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
                    if (jpos0[c] >= 0) {                                          // wave-uniform: one wave per cloud; features of original copy 0
#pragma unroll
                        for (int r = 0; r < 16; ++r)
                            if ((r & 3) + 8 * (r >> 2) + 4 * h == jpos0[c]) {
#pragma unroll
                                for (int mt = 0; mt < MT4; ++mt) v0[b * (32 * T3) + (pass * MT4 + mt) * 32 + j] = l4_unit ? __fmaf_rn(acc[mt][c][r], ACC_UNSCALE, bias4[mt]) : acc[mt][c][r];
                            }
                    }
                    // per node present in this column tile (usually 1, 2 at a node boundary; ids are sorted, so a node's
                    // points are the rows [s, e)): max over its rows (v_max_f32 ignores a NaN operand, as the reference's
                    // '>' does), published by integer atomicMax on orderable keys -- to the LDS bins of the workgroup's
                    // first SEG_SLOTS nodes, else straight to memory.  All branches are wave-uniform.
                    auto publish = [&](int node, float (&mx)[MT4]) __attribute__((always_inline)) {
                        if (l4_unit) {                                            // fl(x / 1024 + b) is monotone in x: after the max
#pragma unroll
                            for (int mt = 0; mt < MT4; ++mt) mx[mt] = __fmaf_rn(mx[mt], ACC_UNSCALE, bias4[mt]);
                        }
                        // two explicit paths: a generic pointer here makes FLAT atomics, and with a FLAT operation anywhere in
                        // the loop hipcc replaces every counted lgkmcnt wait of the MFMA steps by lgkmcnt(0)
                        const int slot = node - n0;
                        if (slot < SEG_SLOTS) {
#pragma unroll
                            for (int mt = 0; mt < MT4; ++mt) atomicMax(&bins[slot][32 * mt + j], ord_f32(__float_as_uint(mx[mt])));
                        } else {
                            unsigned *gdst = pooled + ((long long)b * M + node) * (32 * T3) + pass * PCH;
#pragma unroll
                            for (int mt = 0; mt < MT4; ++mt) atomicMax(gdst + 32 * mt + j, ord_f32(__float_as_uint(mx[mt])));
                        }
                    };
                    unsigned remaining = (unsigned)__ballot(pv[c]);               // lanes 0..31 <-> the column tile's 32 points
                    if (remaining == 0xFFFFFFFFu) {
                        // the common boundary case -- exactly two nodes in 32 valid points, rows [0, e) and [e, 32) -- in one sweep:
                        // one comparison per row serves both maxima
                        const int node_a = __builtin_amdgcn_readlane(nid[c], 0), node_b = __builtin_amdgcn_readlane(nid[c], 31);
                        const int e = __builtin_popcount((unsigned)__ballot(nid[c] == node_a));
                        if (node_a != node_b && __builtin_amdgcn_readlane(nid[c], e & 31) == node_b) {
                            float ma[MT4], mb[MT4];
#pragma unroll
                            for (int mt = 0; mt < MT4; ++mt) ma[mt] = mb[mt] = -__builtin_inff();
#pragma unroll
                            for (int r = 0; r < 16; r += 2) {
                                const int prow = (r & 3) + 8 * (r >> 2) + 4 * h;
                                const bool in0 = prow < e, in1 = prow + 1 < e;
#pragma unroll
                                for (int mt = 0; mt < MT4; ++mt) {
                                    const float a0 = in0 ? acc[mt][c][r] : -__builtin_inff(), a1 = in1 ? acc[mt][c][r + 1] : -__builtin_inff();
                                    const float b0 = in0 ? -__builtin_inff() : acc[mt][c][r], b1 = in1 ? -__builtin_inff() : acc[mt][c][r + 1];
                                    asm("v_max3_f32 %0, %1, %2, %3" : "=v"(ma[mt]) : "v"(ma[mt]), "v"(a0), "v"(a1));
                                    asm("v_max3_f32 %0, %1, %2, %3" : "=v"(mb[mt]) : "v"(mb[mt]), "v"(b0), "v"(b1));
                                }
                            }
                            publish(node_a, ma);
                            publish(node_b, mb);
                            remaining = 0u;
                        }
                    }
                    while (remaining != 0u) {
                        const int s0 = __builtin_ctz(remaining);
                        const int node = __builtin_amdgcn_readlane(nid[c], s0);
                        const unsigned segmask = (unsigned)__ballot(pv[c] && nid[c] == node);
                        remaining &= ~segmask;
                        const int e0 = s0 + __builtin_popcount(segmask);
                        const bool whole = (s0 == 0 && e0 == 32);
                        float mx[MT4];
                        // (v_max3_f32: two rows per instruction, same NaN rule as v_max_f32)
                        if (whole) {
#pragma unroll
                            for (int mt = 0; mt < MT4; ++mt) {
                                float m;
                                asm("v_max_f32 %0, %1, %2" : "=v"(m) : "v"(acc[mt][c][0]), "v"(acc[mt][c][1]));
#pragma unroll
                                for (int r = 2; r < 16; r += 2) asm("v_max3_f32 %0, %1, %2, %3" : "=v"(m) : "v"(m), "v"(acc[mt][c][r]), "v"(acc[mt][c][r + 1]));
                                mx[mt] = m;
                            }
                        } else {
#pragma unroll
                            for (int mt = 0; mt < MT4; ++mt) mx[mt] = -__builtin_inff();
                            const unsigned nrows = (unsigned)(e0 - s0);
#pragma unroll
                            for (int r = 0; r < 16; r += 2) {
                                const int prow = (r & 3) + 8 * (r >> 2) + 4 * h;          // rows prow, prow + 1 (r even: same group of four)
                                const bool in0 = (unsigned)(prow - s0) < nrows, in1 = (unsigned)(prow + 1 - s0) < nrows;
#pragma unroll
                                for (int mt = 0; mt < MT4; ++mt) {
                                    const float v0_ = in0 ? acc[mt][c][r] : -__builtin_inff(), v1_ = in1 ? acc[mt][c][r + 1] : -__builtin_inff();
                                    asm("v_max3_f32 %0, %1, %2, %3" : "=v"(mx[mt]) : "v"(mx[mt]), "v"(v0_), "v"(v1_));
                                }
                            }
                        }
                        publish(node, mx);
                    }
                }
                pend_n = nslots * PCH;
                pend = partial + ((tile * NPASS + pass) * SEG_SLOTS) * (long long)PCH;   // stored at the next flush boundary
            } else {
#pragma unroll
                for (int c = 0; c < 2; ++c) {
                    if (!pv[c]) continue;
#pragma unroll
                    for (int mt = 0; mt < MT4; ++mt) {
                        const int ct = pass * MT4 + mt;
                        const unsigned so_tile = (unsigned)(ct * 32) * rowB;
                        float vv[16];
#pragma unroll
                        for (int r = 0; r < 16; ++r) {
                            const int orow = (r & 3) + 8 * (r >> 2);
                            const float2 ss = aff[LB4 + ct * 32 + orow + 4 * h];
                            const float v = __fmaf_rn(acc[mt][c][r], ss.x, ss.y);
                            vv[r] = v;
                            __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(int, v), ry, (unsigned)(4 * h * L + lc[c]) * 4u,
                                                                  so_tile + (unsigned)orow * rowB, 0);
                        }
                        if constexpr (P16OUT) {
                            // registers 8 q .. 8 q + 7 of tile ct ARE chunk 2 ct + q, half h, elements 0..7 of the P16 planes (pointmlp_h3p.hip):
                            // clamp to the fp16-split range, scale by 32, two roundings, two 16-byte stores per q
#pragma unroll
                            for (int q = 0; q < 2; ++q) {
                                unsigned hh[4], mm[4];
#pragma unroll
                                for (int p = 0; p < 4; ++p) {
                                    const float x0 = vv[8 * q + 2 * p], x1 = vv[8 * q + 2 * p + 1];
                                    range_track(yr4, x0, x1);
                                    const float X0 = 32.f * __builtin_amdgcn_fmed3f(x0, -2047.f, 2047.f), X1 = 32.f * __builtin_amdgcn_fmed3f(x1, -2047.f, 2047.f);
                                    hh[p] = cvt_pk_f16(X0, X1);
                                    mm[p] = cvt_pk_f16(X0 - f16_lo(hh[p]), X1 - f16_hi(hh[p]));
                                }
                                const unsigned so = (unsigned)((ct * 2 + q) * 2) * (unsigned)L * 32u;
                                const u32x4_t hv = {hh[0], hh[1], hh[2], hh[3]}, mv = {mm[0], mm[1], mm[2], mm[3]};
                                __builtin_amdgcn_raw_buffer_store_b128(hv, ryp, (unsigned)(h * L + lc[c]) * 16u, so, 0);
                                __builtin_amdgcn_raw_buffer_store_b128(mv, ryp, (unsigned)(h * L + lc[c]) * 16u, so + (unsigned)L * 32u, 0);
                            }
                        }
                    }
                }
            }
        };
        {                                                       // pass 0 (peeled: it carries the jobs of layer 3's second tile group)
            f32x16 acc[MT4][2];
            PF_L4_BODY(true)
            // the next tile's inputs (here and not in the pass loop: a conditional load in the loop would keep the registers occupied
            // from the top of the tile on)
            prefetch_tile(tile + gridDim.x);
            PROF_MARK(10)                                       // layer-4 pass 0 (carries jobs)
            epilogue(acc, 0);
            PROF_MARK(4)                                        // epilogue (pool or stores)
        }
        for (int pass = 1; pass < NPASS; ++pass) {
            f32x16 acc[MT4][2];
            PF_L4_BODY(false)
            PROF_MARK(3)
            epilogue(acc, pass);
            PROF_MARK(4)
        }
#undef SLOT_L4
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");            // the stages streamed ahead of the last tile
    if constexpr (SEGMAX) {
        __syncthreads();
        flush_bins();
    }
    if (rlog != nullptr) {
        range_publish(rlog, wave_umax(xin_r), lane);
        // the job logged 32 x: take the factor out of the exponent (a NaN / inf stays far above the fp16 range)
        unsigned rb = wave_umax((unsigned)(rmax_ > 0 ? rmax_ : 0));
        rb = rb > (5u << 23) ? rb - (5u << 23) : 0u;
        if constexpr (P16OUT) {                                // (the split output is an operand of the next layer: its range counts too)
            const unsigned ob = wave_umax(range_amax_bits(yr4));
            rb = rb > ob ? rb : ob;
        }
        range_publish(rlog + 2, rb, lane);
        if (blockIdx.x == 0 && threadIdx.x == 0) atomicMax(rlog + 1, reinterpret_cast<const unsigned *>(Wst + (long long)NSLICE * 64)[0]);
    }
    PROF_MARK(5)
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
// One workgroup = 32 nodes x 32 channels of one cloud: the keys are read with the channel fastest (as the fused kernel wrote them),
// the output [B][384][M] is written with the node fastest, through a 32 x 33 LDS tile.
__global__ __launch_bounds__(256) void pooled_decode_kernel(const unsigned *__restrict__ pooled, const unsigned *__restrict__ partial,
                                                             const int32_t *__restrict__ ids_sorted, const int32_t *__restrict__ node_off,
                                                             const int32_t *__restrict__ count, const float *__restrict__ v0,
                                                             float *__restrict__ out, int M, int L, int tpc,
                                                             uint4 *__restrict__ out_p16, unsigned *__restrict__ rlog)
{
    constexpr int C = 32 * T3;
    __shared__ float tile[32][33];
    const long long b = blockIdx.z;
    const int m0 = blockIdx.y * 32, c0 = blockIdx.x * 32;
    const int lx = threadIdx.x & 31, ly = threadIdx.x >> 5;
    const int c = c0 + lx;                                      // (C is a multiple of 32)
    const int pass = c / PCH, cl = c - pass * PCH;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int m = m0 + ly + 8 * i;
        if (m < M) {
            unsigned key = pooled[(b * M + m) * C + c];
            const int cnt = count[b * M + m];
            if (cnt > 0) {
                const int off = node_off[b * M + m];
                for (int tl = off / TPTS; tl <= (off + cnt - 1) / TPTS; ++tl) {
                    const int slot = m - ids_sorted[b * L + tl * TPTS];
                    if (slot < SEG_SLOTS) {
                        const unsigned k2 = partial[((((b * tpc + tl) * NPASS + pass) * SEG_SLOTS) + slot) * (long long)PCH + cl];
                        key = k2 > key ? k2 : key;
                    }
                }
            }
            float v;
            if (key > SEG_INIT) v = __uint_as_float((key & 0x80000000u) ? (key ^ 0x80000000u) : ~key);
            else v = v0[b * C + c];
            tile[ly + 8 * i][lx] = v;
        }
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int cc = c0 + ly + 8 * i, m = m0 + lx;
        if (m < M) out[(b * C + cc) * M + m] = tile[lx][ly + 8 * i];
    }
    if (out_p16 != nullptr && threadIdx.x < 128) {
        // the same values pre-split, on the FLAT column axis of the node-level stage (P16 planes of a 1 x 384 x (B M) activation, column
        // b M + m: node_stage.hip): the operand of KNNModule's first layer.  Thread = (node, 16-channel chunk qq of the 32, half hh).
        const int node = threadIdx.x & 31, qq = (threadIdx.x >> 6) & 1, hh = (threadIdx.x >> 5) & 1;
        const int m = m0 + node;
        RangeAcc xr = {0, 0u};
        if (m < M) {
            const long long BM = ((long long)gridDim.z * M + 127) / 128 * 128, l = b * M + m;      // (plane stride: the flat axis padded to 128 columns)
            float v[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] = tile[node][16 * qq + 4 * hh + (e & 3) + 8 * (e >> 2)];
            unsigned hv[4], mv[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                range_track(xr, v[2 * q], v[2 * q + 1]);
                const float X0 = 32.f * __builtin_amdgcn_fmed3f(v[2 * q], -2047.f, 2047.f), X1 = 32.f * __builtin_amdgcn_fmed3f(v[2 * q + 1], -2047.f, 2047.f);
                typedef _Float16 h2_t __attribute__((ext_vector_type(2)));
                typedef float f2_t __attribute__((ext_vector_type(2)));
                const f2_t xv = {X0, X1};
                const h2_t hp = __builtin_convertvector(xv, h2_t);
                const f2_t rv = {X0 - (float)hp[0], X1 - (float)hp[1]};
                hv[q] = __builtin_bit_cast(unsigned, hp);
                mv[q] = __builtin_bit_cast(unsigned, __builtin_convertvector(rv, h2_t));
            }
            const int kc = c0 / 16 + qq;
            out_p16[((long long)(kc * 2 + 0) * 2 + hh) * BM + l] = make_uint4(hv[0], hv[1], hv[2], hv[3]);
            out_p16[((long long)(kc * 2 + 1) * 2 + hh) * BM + l] = make_uint4(mv[0], mv[1], mv[2], mv[3]);
        }
        // (the planes' consumers cannot check the clamp any more: the largest pooled magnitude goes to word 2 of the launch's range slot)
        if (rlog != nullptr) range_publish(rlog + 2, wave_umax(range_amax_bits(xr)), (int)(threadIdx.x & 63));
    }
}

int cu_count() {
    int dev = 0, cus = 256;
    if (hipGetDevice(&dev) == hipSuccess) {
        int v = 0;
        if (hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && v > 0) cus = v;
    }
    return cus;
}

// Persistent grid: one workgroup per CU.  A workgroup owns a whole CU (512 registers per lane), so nothing of another stream runs beside it.
// Leaving CUs out for the other graphs' small launches was measured and is slower (docs/findings.md R6.1: 3776 tiles walk in 15 rounds on
// 252 workgroups as on 256, and the headline still lost 1.5-2 %; 8 / 12 / 20 CUs out: -1 to -3 %), so the knob stays an experiment of the VARIANTS build:
// SONET_FUSED_FREE_CUS = k leaves at least k CUs out, -1 = the smallest grid with the same number of rounds.
long long persistent_grid(long long ntiles, int cus) {
    static const int free_cus = [] { const char *e = sonet::knob("SONET_FUSED_FREE_CUS"); return e ? atoi(e) : 0; }();   // (variants build only)
    if (ntiles <= cus) return ntiles;
    if (free_cus == 0) return cus;
    const long long rounds = (ntiles + cus - 1) / cus;
    long long grid = (ntiles + rounds - 1) / rounds;
    if (free_cus > 0 && grid > cus - free_cus) grid = cus - free_cus > 1 ? cus - free_cus : 1;
    return free_cus > 0 || free_cus == -1 ? grid : cus;
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
    if ((double)(32 * T3) * L * 4.0 >= 4.0e9) return sonet::fail(SONET_ERR_UNSUPPORTED, "%s: a per-cloud panel exceeds 4 GiB", what);
    const int tpc = sonet::ceil_div(L, TPTS);
    const long long ntiles = (long long)B * tpc;
    const int cus = cu_count();
    const long long grid = persistent_grid(ntiles, cus);
    hipLaunchKernelGGL((pointresnet_fused_kernel<false>), dim3((unsigned)grid), dim3(PF_THREADS), 0, sonet::as_stream(stream),
                       x, Cin0, reinterpret_cast<const uint4 *>(wstream), reinterpret_cast<const float2 *>(affine), y, L, tpc, ntiles,
                       (const int32_t *)nullptr, (const int32_t *)nullptr, (unsigned *)nullptr, (float *)nullptr, 0, (unsigned *)nullptr, sonet::range_log());
    return sonet::launched(what);
}

/* sonet_pointresnet_fused_f32 that also (y != NULL) or only (y == NULL) writes y pre-split: yp = the P16 planes of y (sonet_p16_size(B, 384, L) bytes, include/sonet_hip.h),
 * the operand format of sonet_pointmlp_h3p -- the segmenter's first layer reads first_pn_out per point copy (models/networks.py:296-326). */
extern "C" int sonet_pointresnet_fused_p16_f32(const float *x, int Cin0, const void *wstream, const float *affine,
                                               float *y, void *yp, int B, int L, sonet_stream_t stream)
{
    const char *what = "sonet_pointresnet_fused_p16_f32";
    SONET_REQUIRE(x && wstream && affine && yp, "%s: NULL pointer", what);          // (y may be NULL: only the planes are written)
    SONET_REQUIRE(B > 0 && L > 0 && Cin0 >= 1 && Cin0 <= 16, "%s: bad size B=%d L=%d Cin0=%d", what, B, L, Cin0);
    if ((double)(32 * T3) * L * 4.0 >= 2.0e9) return sonet::fail(SONET_ERR_UNSUPPORTED, "%s: a per-cloud panel exceeds 2 GiB", what);
    const int tpc = sonet::ceil_div(L, TPTS);
    const long long ntiles = (long long)B * tpc;
    const int cus = cu_count();
    const long long grid = persistent_grid(ntiles, cus);
    hipLaunchKernelGGL((pointresnet_fused_kernel<false, true>), dim3((unsigned)grid), dim3(PF_THREADS), 0, sonet::as_stream(stream),
                       x, Cin0, reinterpret_cast<const uint4 *>(wstream), reinterpret_cast<const float2 *>(affine), y, L, tpc, ntiles,
                       (const int32_t *)nullptr, (const int32_t *)nullptr, (unsigned *)nullptr, (float *)nullptr, 0, (unsigned *)nullptr, sonet::range_log(), yp);
    return sonet::launched(what);
}

extern "C" size_t sonet_pointresnet_pool_ws_size(int B, int L, int M)
{
    if (B <= 0 || L <= 0 || M <= 0) return 0;
    const long long ntiles = (long long)B * sonet::ceil_div(L, TPTS);
    return (size_t)((long long)B * M * (32 * T3) + ntiles * NPASS * SEG_SLOTS * PCH) * 4 + (size_t)B * (32 * T3) * 4;
}

static int fused_pool_impl(const char *what, const float *x_sorted, int Cin0, const void *wstream, const float *affine,
                           const int32_t *ids_sorted, const int32_t *pos0, const int32_t *node_off,
                           const int32_t *count, void *ws, float *out, void *out_p16, int B, int L, int M, sonet_stream_t stream)
{
    SONET_REQUIRE(x_sorted && wstream && affine && ids_sorted && pos0 && node_off && count && ws && out, "%s: NULL pointer", what);
    SONET_REQUIRE(B > 0 && L > 0 && M > 0 && Cin0 >= 1 && Cin0 <= 16, "%s: bad size B=%d L=%d M=%d Cin0=%d", what, B, L, M, Cin0);
    if (B > 65535) return sonet::fail(SONET_ERR_UNSUPPORTED, "%s: B=%d > 65535", what, B);
    hipStream_t st = sonet::as_stream(stream);
    const long long npool = (long long)B * M * (32 * T3);
    const int tpc = sonet::ceil_div(L, TPTS);
    const long long ntiles = (long long)B * tpc;
    unsigned *pooled_ws = reinterpret_cast<unsigned *>(ws);
    unsigned *partial_ws = pooled_ws + npool;
    float *v0_ws = reinterpret_cast<float *>(partial_ws + ntiles * NPASS * SEG_SLOTS * PCH);
    hipLaunchKernelGGL(pooled_init_kernel, dim3((unsigned)sonet::ceil_div64(npool, 256)), dim3(256), 0, st, pooled_ws, npool);
    const int cus = cu_count();
    const long long grid = persistent_grid(ntiles, cus);
    hipLaunchKernelGGL((pointresnet_fused_kernel<true>), dim3((unsigned)grid), dim3(PF_THREADS), 0, st,
                       x_sorted, Cin0, reinterpret_cast<const uint4 *>(wstream), reinterpret_cast<const float2 *>(affine), (float *)nullptr,
                       L, tpc, ntiles, ids_sorted, pos0, pooled_ws, v0_ws, M, partial_ws, sonet::range_log());
    hipLaunchKernelGGL(pooled_decode_kernel, dim3((unsigned)(32 * T3 / 32), (unsigned)sonet::ceil_div(M, 32), (unsigned)B), dim3(256), 0, st,
                       pooled_ws, partial_ws, ids_sorted, node_off, count, v0_ws, out, M, L, tpc, reinterpret_cast<uint4 *>(out_p16), sonet::range_log());
    return sonet::launched(what);
}

extern "C" int sonet_pointresnet_fused_pool_f32(const float *x_sorted, int Cin0, const void *wstream, const float *affine,
                                                const int32_t *ids_sorted, const int32_t *pos0, const int32_t *node_off,
                                                const int32_t *count, void *ws, float *out, int B, int L, int M, sonet_stream_t stream)
{
    return fused_pool_impl("sonet_pointresnet_fused_pool_f32", x_sorted, Cin0, wstream, affine, ids_sorted, pos0, node_off, count, ws, out, nullptr, B, L, M, stream);
}

/* The same launches; the decode pass also writes the pooled map pre-split on the flat column axis of the node-level stage: out_p16 =
 * P16 planes of a 1 x 384 x (B M) activation (sonet_p16_size(1, 384, B * M) bytes, column b M + m), the operand of KNNModule's first layer
 * (sonet_pointmlp_h3p); the largest pooled magnitude goes to word 2 of the range log. */
extern "C" int sonet_pointresnet_fused_pool_p16_f32(const float *x_sorted, int Cin0, const void *wstream, const float *affine,
                                                    const int32_t *ids_sorted, const int32_t *pos0, const int32_t *node_off,
                                                    const int32_t *count, void *ws, float *out, void *out_p16, int B, int L, int M, sonet_stream_t stream)
{
    const char *what = "sonet_pointresnet_fused_pool_p16_f32";
    SONET_REQUIRE(out_p16, "%s: NULL pointer", what);
    return fused_pool_impl(what, x_sorted, Cin0, wstream, affine, ids_sorted, pos0, node_off, count, ws, out, out_p16, B, L, M, stream);
}

#ifdef SONET_PROF
extern "C" int sonet_prof_read(long long *host, int n)
{
    return hipMemcpyFromSymbol(host, HIP_SYMBOL(g_prof), sizeof(long long) * (size_t)n, 0, hipMemcpyDeviceToHost) == hipSuccess ? 0 : 1;
}
#endif
