// common.hpp -- shared host-side helpers of libsonet_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdarg.h>

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

static inline int ceil_div(int a, int b) { return (a + b - 1) / b; }
static inline long long ceil_div64(long long a, long long b) { return (a + b - 1) / b; }

}  // namespace sonet

// Bijective XCD-aware remap of a 1-D block id: consecutive *virtual* ids land on the same XCD
// (and so share its L2).  Pure speed choice -- correctness never depends on placement.
__device__ __forceinline__ int xcd_remap(int bid, int nwg) {
    const int q = nwg / sonet::NUM_XCD, r = nwg % sonet::NUM_XCD;
    const int xcd = bid % sonet::NUM_XCD, local = bid / sonet::NUM_XCD;
    const int base = xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
    return base + local;
}
