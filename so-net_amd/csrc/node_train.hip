// node_train.hip -- node-level pieces of the TRAINING step that ran on aten (rocprofv3 of the bf16 step: _scatter_gather_elementwise 0.23 ms,
// reduce_kernel<MaxOps> 0.18 ms, fills and the index expansion of torch.gather's backward):
//   * max over the last dimension WITH the arg-max (first maximum, as torch.max documents) and its backward -- the max over the K'
//     neighbours of KNNModule (models/layers.py:350-365: torch.max(dim=3)) and over the M nodes (models/networks.py:197);
//   * the backward of the neighbour gather (models/operations.py:38-54): gx[b][c][m] = sum over the (m', k) with knn_I[b][m'][k] == m of
//     g[b][c][m'][k], as a GATHER over per-cloud inverse lists (fixed summation order: bitwise reproducible; aten scatter-adds atomically).
#include "common.hpp"

namespace {

__device__ __forceinline__ float ld(const float *p, long long i) { return p[i]; }
__device__ __forceinline__ float ld(const uint16_t *p, long long i) { return __uint_as_float((unsigned)p[i] << 16); }
__device__ __forceinline__ void st(float *p, long long i, float v) { p[i] = v; }
__device__ __forceinline__ void st(uint16_t *p, long long i, float v) {           // v is a value that came from bf16 storage (or 0): exact
    p[i] = (uint16_t)(__float_as_uint(v) >> 16);
}

// one thread per row of K contiguous values: value and index of the FIRST maximum; a NaN wins (the first one), as torch.max
template <typename T>
__global__ __launch_bounds__(256) void lastdim_argmax_kernel(const T *__restrict__ x, T *__restrict__ out, int32_t *__restrict__ idx, int K, long long rows)
{
    const long long r = (long long)blockIdx.x * 256 + threadIdx.x;
    if (r >= rows) return;
    const long long base = r * K;
    float m = ld(x, base);
    int mi = 0;
    for (int k = 1; k < K; ++k) {
        const float v = ld(x, base + k);
        if (v > m || (v != v && m == m)) { m = v; mi = k; }
    }
    st(out, r, m);
    idx[r] = mi;
}

// gx[r][k] = (k == idx[r]) ? g[r] : 0 : the whole row is written (no separate fill)
template <typename T>
__global__ __launch_bounds__(256) void lastdim_max_bwd_kernel(const T *__restrict__ g, const int32_t *__restrict__ idx, T *__restrict__ gx, int K, long long rows)
{
    const long long r = (long long)blockIdx.x * 256 + threadIdx.x;
    if (r >= rows) return;
    const float gv = ld(g, r);
    const int mi = idx[r];
    const long long base = r * K;
    for (int k = 0; k < K; ++k) st(gx, base + k, k == mi ? gv : 0.f);
}

// inverse neighbour lists of one cloud: off[b][m] .. off[b][m+1] index into list[b][.] = the entries e = m' * K + k (ascending) whose
// neighbour is node m.  One workgroup per cloud, thread m owns node m (M <= 1024); entries with an index outside [0, M) belong to nobody.
__global__ __launch_bounds__(1024) void knn_inverse_kernel(const long long *__restrict__ knn_I, int M, int K, int32_t *__restrict__ off, int32_t *__restrict__ list)
{
    __shared__ int cnt[1025];
    extern __shared__ int Is[];                  // the cloud's E indices (out of range: -1).  Every thread walks all of them twice: straight
                                                 // from global memory that was 1152 dependent loads per thread, 87 us of the training step
    const int b = blockIdx.x, m = threadIdx.x;
    const long long *I = knn_I + (long long)b * M * K;
    const int E = M * K;
    for (int e = m; e < E; e += blockDim.x) {
        const long long i = I[e];
        Is[e] = (i >= 0 && i < M) ? (int)i : -1;
    }
    __syncthreads();
    int n = 0;
    if (m < M)
        for (int e = 0; e < E; ++e) n += (Is[e] == m);
    cnt[m] = m < M ? n : 0;
    __syncthreads();
    if (m == 0) {
        int acc = 0;
        for (int i = 0; i < M; ++i) { const int c = cnt[i]; cnt[i] = acc; acc += c; }
        cnt[M] = acc;
    }
    __syncthreads();
    // (M + 1 offsets from at most 1024 threads: at M == 1024 thread 0 also writes the closing one)
    for (int i = m; i <= M; i += blockDim.x) off[(long long)b * (M + 1) + i] = cnt[i];
    if (m < M) {
        int w = cnt[m];
        int32_t *L = list + (long long)b * E;
        for (int e = 0; e < E; ++e)
            if (Is[e] == m) L[w++] = e;
    }
}

// one thread per (b, c, m): the sum of its list's entries of g[b][c][.], in list order, f32 accumulation
template <typename T>
__global__ __launch_bounds__(256) void knn_gather_bwd_kernel(const T *__restrict__ g, const int32_t *__restrict__ off, const int32_t *__restrict__ list,
                                                              float *__restrict__ gx, int C, int M, int K, long long total)
{
    const long long t = (long long)blockIdx.x * 256 + threadIdx.x;
    if (t >= total) return;
    const int m = (int)(t % M);
    const long long bc = t / M;
    const int b = (int)(bc / C);
    const int32_t *o = off + (long long)b * (M + 1);
    const int32_t *L = list + (long long)b * M * K;
    const long long base = bc * (long long)M * K;
    float s = 0.f;
    for (int i = o[m]; i < o[m + 1]; ++i) s += ld(g, base + L[i]);
    gx[t] = s;
}

}  // namespace

template <typename T>
static int argmax_impl(const char *what, const T *x, T *out, int32_t *idx, long long rows, int K, sonet_stream_t stream)
{
    SONET_REQUIRE(x && out && idx, "%s: NULL pointer", what);
    SONET_REQUIRE(rows > 0 && K > 0, "%s: bad size", what);
    const long long blocks = sonet::ceil_div64(rows, 256);
    if (blocks > 0x7FFFFFFFll) return sonet::fail(SONET_ERR_UNSUPPORTED, "%s: too large", what);
    hipLaunchKernelGGL(lastdim_argmax_kernel<T>, dim3((unsigned)blocks), dim3(256), 0, sonet::as_stream(stream), x, out, idx, K, rows);
    return sonet::launched(what);
}
template <typename T>
static int maxbwd_impl(const char *what, const T *g, const int32_t *idx, T *gx, long long rows, int K, sonet_stream_t stream)
{
    SONET_REQUIRE(g && gx && idx, "%s: NULL pointer", what);
    SONET_REQUIRE(rows > 0 && K > 0, "%s: bad size", what);
    const long long blocks = sonet::ceil_div64(rows, 256);
    if (blocks > 0x7FFFFFFFll) return sonet::fail(SONET_ERR_UNSUPPORTED, "%s: too large", what);
    hipLaunchKernelGGL(lastdim_max_bwd_kernel<T>, dim3((unsigned)blocks), dim3(256), 0, sonet::as_stream(stream), g, idx, gx, K, rows);
    return sonet::launched(what);
}

extern "C" int sonet_lastdim_argmax_f32(const float *x, float *out, int32_t *idx, long long rows, int K, sonet_stream_t stream)
{
    return argmax_impl("sonet_lastdim_argmax_f32", x, out, idx, rows, K, stream);
}
extern "C" int sonet_lastdim_argmax_bf16(const uint16_t *x, uint16_t *out, int32_t *idx, long long rows, int K, sonet_stream_t stream)
{
    return argmax_impl("sonet_lastdim_argmax_bf16", x, out, idx, rows, K, stream);
}
extern "C" int sonet_lastdim_max_bwd_f32(const float *g, const int32_t *idx, float *gx, long long rows, int K, sonet_stream_t stream)
{
    return maxbwd_impl("sonet_lastdim_max_bwd_f32", g, idx, gx, rows, K, stream);
}
extern "C" int sonet_lastdim_max_bwd_bf16(const uint16_t *g, const int32_t *idx, uint16_t *gx, long long rows, int K, sonet_stream_t stream)
{
    return maxbwd_impl("sonet_lastdim_max_bwd_bf16", g, idx, gx, rows, K, stream);
}

extern "C" size_t sonet_knn_gather_bwd_ws_size(int B, int M, int K)
{
    if (B <= 0 || M <= 0 || K <= 0) return 0;
    return ((size_t)B * (M + 1) + (size_t)B * M * K) * sizeof(int32_t);
}

template <typename T>
static int gather_bwd_impl(const char *what, const T *g, const int64_t *knn_I, float *gx, void *ws, int B, int C, int M, int K, sonet_stream_t stream)
{
    SONET_REQUIRE(g && knn_I && gx && ws, "%s: NULL pointer", what);
    SONET_REQUIRE(B > 0 && C > 0 && M > 0 && K > 0, "%s: bad size", what);
    if (M > 1024) return sonet::fail(SONET_ERR_UNSUPPORTED, "%s: M=%d > 1024", what, M);
    int32_t *off = reinterpret_cast<int32_t *>(ws), *list = off + (size_t)B * (M + 1);
    hipStream_t s = sonet::as_stream(stream);
    if ((size_t)M * K * 4 > 56 * 1024) return sonet::fail(SONET_ERR_UNSUPPORTED, "%s: M*K=%d indices do not fit the LDS", what, M * K);
    hipLaunchKernelGGL(knn_inverse_kernel, dim3((unsigned)B), dim3(1024), (size_t)M * K * 4, s, reinterpret_cast<const long long *>(knn_I), M, K, off, list);
    const long long total = (long long)B * C * M;
    hipLaunchKernelGGL(knn_gather_bwd_kernel<T>, dim3((unsigned)sonet::ceil_div64(total, 256)), dim3(256), 0, s, g, off, list, gx, C, M, K, total);
    return sonet::launched(what);
}
extern "C" int sonet_knn_gather_bwd_f32(const float *g, const int64_t *knn_I, float *gx, void *ws, int B, int C, int M, int K, sonet_stream_t stream)
{
    return gather_bwd_impl("sonet_knn_gather_bwd_f32", g, knn_I, gx, ws, B, C, M, K, stream);
}
extern "C" int sonet_knn_gather_bwd_bf16(const uint16_t *g, const int64_t *knn_I, float *gx, void *ws, int B, int C, int M, int K, sonet_stream_t stream)
{
    return gather_bwd_impl("sonet_knn_gather_bwd_bf16", g, knn_I, gx, ws, B, C, M, K, stream);
}
