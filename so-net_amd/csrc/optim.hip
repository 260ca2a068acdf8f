// optim.hip -- torch.optim.Adam's update (models/classifier.py:45-49: Adam, betas (0.9, 0.999), no weight decay, no amsgrad) for ALL
// parameters of an optimizer in ONE launch.
//
// The reference steps two optimizers per iteration; PyTorch's foreach implementation costs ~12 multi-tensor launches of 14-23 us each
// (0.25 ms of a 5.7 ms training step at B = 64: profiles/r04z_kernel_stats_train_bf16.csv).  Here a launch walks a chunk table: chunk j
// = 4096 elements of tensor chunk_tensor[j] starting at chunk_off[j]; per tensor a record of four pointers (param, grad, exp_avg,
// exp_avg_sq) and its own step-dependent scalars (a parameter that received no gradient keeps its step count, as in torch: its grad
// pointer is NULL and its chunks return at once).  Same operations in the same order as torch's `_single_tensor_adam`:
//     m <- lerp(m, g, 1 - beta1);  v <- v * beta2 + (1 - beta2) * g * g;  p <- p - step_size * m / (sqrt(v) / sqrt(bc2) + eps)
// (f32 throughout; the scalars are computed in double on the host, as Python floats are).
#include "common.hpp"

namespace {

struct AdamTensor { float *p; const float *g; float *m; float *v; };     // 32 bytes per tensor, device table

constexpr int AD_CHUNK = 4096;

__global__ __launch_bounds__(256) void adam_multi_kernel(const AdamTensor *__restrict__ tensors, const float *__restrict__ step_size,
                                                         const float *__restrict__ bc2_sqrt, const int32_t *__restrict__ chunk_tensor,
                                                         const long long *__restrict__ chunk_off, const long long *__restrict__ sizes,
                                                         float beta1, float beta2, float w1, float w2, float eps)
{
    const int t = chunk_tensor[blockIdx.x];
    const AdamTensor T = tensors[t];
    if (T.g == nullptr) return;
    const long long off = chunk_off[blockIdx.x];
    const long long n = sizes[t] - off < AD_CHUNK ? sizes[t] - off : AD_CHUNK;
    const float ss = step_size[t], bs = bc2_sqrt[t];
    for (int i = threadIdx.x; i < n; i += 256) {
        const long long e = off + i;
        const float g = T.g[e];
        float m = T.m[e], v = T.v[e];
        // torch.lerp(m, g, w): w < 0.5 ? m + w (g - m) : g - (g - m)(1 - w)
        const float d = __fsub_rn(g, m);
        m = w1 < 0.5f ? __fadd_rn(m, __fmul_rn(w1, d)) : __fsub_rn(g, __fmul_rn(d, __fsub_rn(1.f, w1)));
        v = __fadd_rn(__fmul_rn(v, beta2), __fmul_rn(__fmul_rn(w2, g), g));          // mul_(beta2).addcmul_(g, g, value = 1 - beta2)
        const float denom = __fadd_rn(__fdiv_rn(__fsqrt_rn(v), bs), eps);
        T.p[e] = __fadd_rn(T.p[e], __fmul_rn(-ss, __fdiv_rn(m, denom)));            // addcdiv_(m, denom, value = -step_size)
        T.m[e] = m;
        T.v[e] = v;
    }
}

}  // namespace

/* One Adam step for `ntensors` f32 parameters in one launch (torch.optim.Adam with amsgrad = False, weight_decay = 0, maximize = False).
 * tensors: device table of {param, grad, exp_avg, exp_avg_sq} pointers (grad NULL: the parameter is skipped); step_size[t] = lr / (1 -
 * beta1^step_t), bc2_sqrt[t] = sqrt(1 - beta2^step_t) (device, f32); chunk_tensor / chunk_off: the chunk table (4096 elements per
 * chunk), sizes[t] the element counts (device).  one_minus_beta1/2: 1 - beta computed in double by the caller (as torch passes its Python
 * floats: 1.f - 0.999f is 1.3e-5 off).  All pointers are device pointers. */
extern "C" int sonet_adam_multi_f32(const void *tensors, const float *step_size, const float *bc2_sqrt, const int32_t *chunk_tensor,
                                    const long long *chunk_off, const long long *sizes, int nchunks, float beta1, float beta2,
                                    float one_minus_beta1, float one_minus_beta2, float eps, sonet_stream_t stream)
{
    const char *what = "sonet_adam_multi_f32";
    SONET_REQUIRE(tensors && step_size && bc2_sqrt && chunk_tensor && chunk_off && sizes, "%s: NULL pointer", what);
    SONET_REQUIRE(nchunks > 0, "%s: no chunks", what);
    hipLaunchKernelGGL(adam_multi_kernel, dim3((unsigned)nchunks), dim3(256), 0, sonet::as_stream(stream),
                       reinterpret_cast<const AdamTensor *>(tensors), step_size, bc2_sqrt, chunk_tensor, chunk_off, sizes, beta1, beta2, one_minus_beta1, one_minus_beta2, eps);
    return sonet::launched(what);
}

extern "C" int sonet_adam_chunk(void) { return AD_CHUNK; }
