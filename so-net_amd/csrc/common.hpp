// common.hpp -- shared host-side helpers of libsonet_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdarg.h>
#include <stdlib.h>

#include "../../include/sonet_hip.h"

namespace sonet {

constexpr int WAVE = 64;            // CDNA4 wavefront
constexpr int NUM_XCD = 8;          // MI355X: 8 XCDs, block b is dispatched to XCD b % 8 (speed only)

// thread-local last-error text behind sonet_last_error()
char *err_buf();
int fail(int code, const char *fmt, ...);

static inline hipStream_t as_stream(sonet_stream_t s) { return reinterpret_cast<hipStream_t>(s); }

// post-launch check: launch-configuration errors surface here; no synchronisation.
static inline int launched(const char *what) {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail(SONET_ERR_LAUNCH, "%s: %s", what, hipGetErrorString(e));
    return SONET_OK;
}

#define SONET_REQUIRE(cond, ...)                                      \
    do {                                                              \
        if (!(cond)) return ::sonet::fail(SONET_ERR_INVALID_ARG, __VA_ARGS__); \
    } while (0)

// Range log of the fp16-split ("h3") kernels: thread-local pointer to the 8-word slot the NEXT h3 launch of this thread
// reports its operand magnitudes into (sonet_range_log_set; NULL = no report).  See include/sonet_hip.h.
uint32_t *range_log();

// Tuning / ablation knobs (SONET_* environment variables of the experiment tools): they exist only in the VARIANTS build
// (make -C so-net_amd/csrc variants -> lib/libsonet_hip_variants.so, -DSONET_VARIANTS).  The product library reads no
// environment variable: every knob is the constant "unset" there and the code behind it is compiled out.
#ifdef SONET_VARIANTS
static inline const char *knob(const char *name) { return getenv(name); }
#else
static inline const char *knob(const char *) { return nullptr; }
#endif

static inline int ceil_div(int a, int b) { return (a + b - 1) / b; }
static inline long long ceil_div64(long long a, long long b) { return (a + b - 1) / b; }

// Zero `bytes` (a multiple of 4, 4-byte aligned) of device memory with a KERNEL instead of hipMemsetAsync.  The forward of a batch is
// captured into a HIP graph and several such graphs are replayed concurrently on different streams (bench.py --in-flight); with
// memset NODES in the graphs that combination faulted intermittently ("write access to a read-only page") on this runtime.
// Defined in api.hip.
int zero_words(void *p, size_t bytes, hipStream_t st);

// statistics epilogue of the layer kernels: mean / biased variance from per-workgroup (sum, sum of squares) partials
// [nwg][C][2] (f64), fixed order.  Defined in pointmlp_x3.hip.
int launch_stats_finalize(const double *partial, int nwg, int C, double inv_n, float *mean, float *var, hipStream_t st);

// "BatchNorm rider" of the training forward (sonet_bn_rider_set, include/sonet_hip.h): what the NEXT statistics finalize of this thread
// also computes per channel -- the normalisation coefficients (sonet_bn_fwd_coeffs_f32) and F.batch_norm's running-statistics update
// (sonet_bn_running_update_f32), same arithmetic in the same order -- instead of two more C-element launches per BatchNorm layer and step.
// take_bn_rider() hands it out once and clears it.  Defined in api.hip.
struct BnRider {
    const float *gamma, *beta;            // NULL gamma: no rider
    float eps, momentum, unbias;
    float *rmean, *rvar;                  // running statistics, updated in place (may be NULL)
    float *invstd, *sc, *sh;              // outputs [C]
};
BnRider take_bn_rider();

}  // namespace sonet

// sum over the 32 lanes of a half wave, result in every lane (all lanes must be active): xor-1, xor-2 inside a quad, mirror inside
// 8 and 16 lanes (DPP modifiers of the add), then the other row of 16 through ds_swizzle (no LDS memory is touched)
__device__ __forceinline__ float row32_sum(float v) {
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1, 0xF, 0xF, true));    // quad_perm [1,0,3,2]
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x4E, 0xF, 0xF, true));    // quad_perm [2,3,0,1]
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x141, 0xF, 0xF, true));   // row_half_mirror
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x140, 0xF, 0xF, true));   // row_mirror
    v += __builtin_bit_cast(float, __builtin_amdgcn_ds_swizzle(__builtin_bit_cast(int, v), 0x401F));                     // lane ^ 16
    return v;
}


// Bijective XCD-aware remap of a 1-D block id: consecutive *virtual* ids land on the same XCD
// (and so share its L2).  Pure speed choice -- correctness never depends on placement.
__device__ __forceinline__ int xcd_remap(int bid, int nwg) {
    const int q = nwg / sonet::NUM_XCD, r = nwg % sonet::NUM_XCD;
    const int xcd = bid % sonet::NUM_XCD, local = bid / sonet::NUM_XCD;
    const int base = xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
    return base + local;
}

// ---- operand-range tracking of the fp16-split kernels (two integer max per value pair) ---------------------------
// |x| as ordered bits over everything a lane has seen: positive floats order as signed ints (mp), negative ones --
// sign bit set -- order by magnitude as unsigned ints (mn); a NaN of either sign lands above +-inf in one of the two.
struct RangeAcc { int mp; unsigned mn; };
__device__ __forceinline__ void range_track(RangeAcc &r, float x0, float x1) {
    const int a = __float_as_int(x0), b = __float_as_int(x1);
    r.mp = max(max(r.mp, a), b);                                           // v_max3_i32
    r.mn = max(max(r.mn, (unsigned)a), (unsigned)b);                       // v_max3_u32
}
__device__ __forceinline__ unsigned range_amax_bits(const RangeAcc &r) {   // bits of max |x| (NaN > inf > finite)
    const unsigned neg = (r.mn & 0x80000000u) ? (r.mn & 0x7FFFFFFFu) : 0u;
    const unsigned pos = (unsigned)r.mp;
    return pos > neg ? pos : neg;
}
__device__ __forceinline__ unsigned wave_umax(unsigned v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { const unsigned t = (unsigned)__shfl_xor((int)v, o, 64); v = t > v ? t : v; }
    return v;
}
// one atomic per wave at most, and only while the wave's value still raises the word
__device__ __forceinline__ void range_publish(unsigned *word, unsigned wave_max_bits, int lane) {
    if (lane == 0 && wave_max_bits > __atomic_load_n(word, __ATOMIC_RELAXED)) atomicMax(word, wave_max_bits);
}

// ---- P16 planes (include/sonet_hip.h): a value pair clamped to the fp16-split range, scaled by 32, as packed fp16 hi + packed fp16 residual
// (the residual is exact in f32 before it is rounded) -- the arithmetic of split_act in pointmlp_h3p.hip, for the producers outside it
__device__ __forceinline__ void p16_split_pair(float x0, float x1, unsigned &h, unsigned &m) {
    typedef _Float16 sonet_h2_t __attribute__((ext_vector_type(2)));
    typedef float sonet_f2_t __attribute__((ext_vector_type(2)));
    const sonet_f2_t X = {32.f * __builtin_amdgcn_fmed3f(x0, -2047.f, 2047.f), 32.f * __builtin_amdgcn_fmed3f(x1, -2047.f, 2047.f)};
    const sonet_h2_t hv = __builtin_convertvector(X, sonet_h2_t);
    const sonet_f2_t R = {X[0] - (float)hv[0], X[1] - (float)hv[1]};
    h = __builtin_bit_cast(unsigned, hv);
    m = __builtin_bit_cast(unsigned, __builtin_convertvector(R, sonet_h2_t));
}
// groups (nodes) per 128-column block of the K-level tensor of the flat node-level stage (node_stage.hip): as many as fit, at most 16
static inline int knn_stage_groups(int K) { const int g = 128 / K; return g > 16 ? 16 : g; }
