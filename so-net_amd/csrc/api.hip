// api.hip -- library identity and error plumbing of libsonet_hip.so.
#include "common.hpp"
#include <string.h>

namespace sonet {

char *err_buf() {
    static thread_local char buf[512] = {0};
    return buf;
}

int fail(int code, const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(err_buf(), 512, fmt, ap);
    va_end(ap);
    return code;
}

namespace {
__global__ __launch_bounds__(256) void zero_words_kernel(uint32_t *__restrict__ p, size_t n) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) p[i] = 0u;
}
}  // namespace

int zero_words(void *p, size_t bytes, hipStream_t st) {
    const size_t n = bytes / 4;
    if (n == 0) return 0;
    size_t blocks = (n + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(zero_words_kernel, dim3((unsigned)blocks), dim3(256), 0, st, reinterpret_cast<uint32_t *>(p), n);
    return hipGetLastError() == hipSuccess ? 0 : 1;
}

static thread_local uint32_t *g_range_log = nullptr;
uint32_t *range_log() { return g_range_log; }

}  // namespace sonet

extern "C" int sonet_range_log_set(uint32_t *slot) { sonet::g_range_log = slot; return SONET_OK; }

namespace sonet {
static thread_local BnRider g_bn_rider = {nullptr, nullptr, 0.f, 0.f, 0.f, nullptr, nullptr, nullptr, nullptr, nullptr};
BnRider take_bn_rider() {
    const BnRider r = g_bn_rider;
    g_bn_rider.gamma = nullptr;
    return r;
}
}  // namespace sonet

extern "C" int sonet_bn_rider_set(const float *gamma, const float *beta, float eps, float momentum, float unbias,
                                  float *running_mean, float *running_var, float *invstd, float *scale, float *shift)
{
    if (gamma == nullptr) { sonet::g_bn_rider.gamma = nullptr; return SONET_OK; }
    SONET_REQUIRE(beta && invstd && scale && shift, "sonet_bn_rider_set: NULL pointer");
    SONET_REQUIRE((running_mean == nullptr) == (running_var == nullptr), "sonet_bn_rider_set: running_mean and running_var come together");
    sonet::g_bn_rider = sonet::BnRider{gamma, beta, eps, momentum, unbias, running_mean, running_var, invstd, scale, shift};
    return SONET_OK;
}

extern "C" int sonet_abi_version(void) { return 1; }
extern "C" const char *sonet_build_arch(void) { return "gfx950"; }
extern "C" const char *sonet_last_error(void) { return sonet::err_buf(); }

extern "C" int sonet_check_device(void) {
    int dev = -1;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0) {
        (void)hipGetLastError();
        return sonet::fail(SONET_ERR_NO_DEVICE, "sonet_check_device: no HIP device is current");
    }
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, dev) != hipSuccess)
        return sonet::fail(SONET_ERR_NO_DEVICE, "sonet_check_device: hipGetDeviceProperties failed");
    if (strncmp(prop.gcnArchName, "gfx950", 6) != 0)
        return sonet::fail(SONET_ERR_NO_DEVICE, "sonet_check_device: device %d is %s, this library is built for gfx950 only",
                           dev, prop.gcnArchName);
    return SONET_OK;
}
