// fc_head.hip -- the B x C fully connected layers of the heads in TRAINING: Linear + BatchNorm1d (batch statistics) + ReLU
// (models/layers.py:123-166 as stacked by models/networks.py:202-227: 1024 -> 512 -> 256 -> classes) as ONE forward launch and TWO
// backward launches per layer.
//
// On aten a layer is addmm + batch_norm + relu forward and threshold_backward + batch_norm_backward + two mm + a bias reduction backward:
// about 30 launches for the classifier, and hipBLASLt needs 10-47 us for each of the 64-row GEMMs (profiles/r04z_kernel_stats_train_bf16.csv:
// six Cijk_* kernels, 0.16 ms of a 5.4 ms step).  The problems are tiny (B <= 128 rows, 0.67 M weights): what matters is the number of
// launches and that each one spreads over many CUs.  f32 throughout, plain FMAs (the matrix cores have no f32 rate to speak of and the
// operands come out of L2), every reduction in a fixed order.
//
//   forward   workgroup = 4 output channels x all rows (one wave per channel, a lane per row and 64-row block): the x tile [B][64] and the
//             W tile [4][64] go through LDS, the batch statistics of a channel are a wave reduction, running statistics updated in place
//   backward  (1) same grid: ReLU mask, BatchNorm backward (two wave reductions), dz written, the four rows of dW = dz^T x (a thread per
//             input channel, dz broadcast from LDS), bias / gamma / beta gradients;  (2) dx = dz W: workgroup = 64 input channels x 16 rows.
#include "common.hpp"

namespace {

constexpr int FC_THREADS = 256, FC_KT = 64, FC_XP = FC_KT + 4;      // x tile row pitch (floats): 16-byte rows, conflict-free b128 reads
constexpr int FC_CS = 4;                                             // output channels per workgroup (one wave each)
constexpr int FC_MAXB = 128;

__device__ __forceinline__ float wave_sum(float v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

template <int RB>
__global__ __launch_bounds__(FC_THREADS) void fc_bn_act_fwd_kernel(const float *__restrict__ x, const float *__restrict__ W,
                                                                   const float *__restrict__ bias, const float *__restrict__ gamma,
                                                                   const float *__restrict__ beta, float *__restrict__ running_mean,
                                                                   float *__restrict__ running_var, float momentum, float eps, int relu,
                                                                   int B, int Cin, int Cout, float *__restrict__ y,
                                                                   float *__restrict__ xhat, float *__restrict__ invstd_out)
{
    __shared__ __attribute__((aligned(16))) float sm[RB * 64 * FC_XP + FC_CS * FC_KT];
    float *xs = sm, *ws = sm + RB * 64 * FC_XP;
    const int t = threadIdx.x, lane = t & 63, wv = t >> 6;
    const int c = blockIdx.x * FC_CS + wv;
    float acc[RB];
#pragma unroll
    for (int j = 0; j < RB; ++j) acc[j] = 0.f;
    // the next tile's global loads are in flight while the current one is multiplied out of LDS (a workgroup is one latency chain
    // of Cin / 64 tiles otherwise: 45 us for 1024 input channels)
    constexpr int XV = RB * 64 * (FC_KT / 4) / FC_THREADS;          // float4 per thread and x tile
    float4 xr[XV], wr = make_float4(0.f, 0.f, 0.f, 0.f);
    auto fetch = [&](int k0) {
#pragma unroll
        for (int i = 0; i < XV; ++i) {
            const int idx = t + i * FC_THREADS, row = idx >> 4, q = idx & 15, k = k0 + q * 4;
            xr[i] = (row < B && k < Cin) ? *reinterpret_cast<const float4 *>(x + (size_t)row * Cin + k) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
        if (t < FC_CS * (FC_KT / 4)) {
            const int cc = t >> 4, q = t & 15, k = k0 + q * 4, co = blockIdx.x * FC_CS + cc;
            wr = (co < Cout && k < Cin) ? *reinterpret_cast<const float4 *>(W + (size_t)co * Cin + k) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
    };
    fetch(0);
    for (int k0 = 0; k0 < Cin; k0 += FC_KT) {
#pragma unroll
        for (int i = 0; i < XV; ++i) {
            const int idx = t + i * FC_THREADS, row = idx >> 4, q = idx & 15;
            *reinterpret_cast<float4 *>(xs + row * FC_XP + q * 4) = xr[i];
        }
        if (t < FC_CS * (FC_KT / 4)) *reinterpret_cast<float4 *>(ws + (t >> 4) * FC_KT + (t & 15) * 4) = wr;
        __syncthreads();
        if (k0 + FC_KT < Cin) fetch(k0 + FC_KT);
#pragma unroll 4
        for (int q = 0; q < FC_KT / 4; ++q) {
            const float4 w4 = *reinterpret_cast<const float4 *>(ws + wv * FC_KT + q * 4);
#pragma unroll
            for (int j = 0; j < RB; ++j) {
                const float4 x4 = *reinterpret_cast<const float4 *>(xs + (lane + 64 * j) * FC_XP + q * 4);
                acc[j] = fmaf(x4.x, w4.x, acc[j]);
                acc[j] = fmaf(x4.y, w4.y, acc[j]);
                acc[j] = fmaf(x4.z, w4.z, acc[j]);
                acc[j] = fmaf(x4.w, w4.w, acc[j]);
            }
        }
        __syncthreads();
    }
    const bool cok = c < Cout;
    const float bv = (cok && bias) ? bias[c] : 0.f;
    float z[RB];
    bool ok[RB];
#pragma unroll
    for (int j = 0; j < RB; ++j) {
        ok[j] = cok && lane + 64 * j < B;
        z[j] = ok[j] ? acc[j] + bv : 0.f;
    }
    if (gamma) {                                                     // training BatchNorm1d: statistics over the B rows (biased variance)
        float s = 0.f;
#pragma unroll
        for (int j = 0; j < RB; ++j) s += z[j];
        const float mean = wave_sum(s) / (float)B;
        float v = 0.f;
#pragma unroll
        for (int j = 0; j < RB; ++j) {
            z[j] = ok[j] ? z[j] - mean : 0.f;
            v = fmaf(z[j], z[j], v);
        }
        const float var = wave_sum(v) / (float)B;
        const float is = 1.0f / sqrtf(var + eps);
        const float ga = cok ? gamma[c] : 0.f, be = cok ? beta[c] : 0.f;
#pragma unroll
        for (int j = 0; j < RB; ++j) {
            const float xh = z[j] * is;
            if (ok[j]) xhat[(size_t)(lane + 64 * j) * Cout + c] = xh;
            z[j] = fmaf(xh, ga, be);
        }
        if (cok && lane == 0) {
            invstd_out[c] = is;
            if (running_mean) running_mean[c] = (1.f - momentum) * running_mean[c] + momentum * mean;
            if (running_var) running_var[c] = (1.f - momentum) * running_var[c] + momentum * (var * ((float)B / (float)(B - 1)));
        }
    }
#pragma unroll
    for (int j = 0; j < RB; ++j)
        if (ok[j]) y[(size_t)(lane + 64 * j) * Cout + c] = relu ? fmaxf(z[j], 0.f) : z[j];
}

// gy -> dz (through the ReLU mask and the BatchNorm backward), the workgroup's four rows of dW = dz^T x, and the vector gradients
template <int RB>
__global__ __launch_bounds__(FC_THREADS) void fc_bn_act_bwd_kernel(const float *__restrict__ gy, const float *__restrict__ y,
                                                                   const float *__restrict__ xhat, const float *__restrict__ invstd,
                                                                   const float *__restrict__ gamma, const float *__restrict__ x, int relu,
                                                                   int B, int Cin, int Cout, float *__restrict__ dz_out,
                                                                   float *__restrict__ dW, float *__restrict__ dbias,
                                                                   float *__restrict__ dgamma, float *__restrict__ dbeta)
{
    __shared__ __attribute__((aligned(16))) float dzs[RB * 64 * FC_CS];      // [row][channel of the workgroup]
    const int t = threadIdx.x, lane = t & 63, wv = t >> 6;
    const int c = blockIdx.x * FC_CS + wv;
    const bool cok = c < Cout;
    float g[RB], xh[RB];
    bool ok[RB];
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int j = 0; j < RB; ++j) {
        const int row = lane + 64 * j;
        ok[j] = cok && row < B;
        const size_t at = (size_t)row * Cout + c;
        g[j] = ok[j] ? gy[at] : 0.f;
        if (relu && ok[j] && !(y[at] > 0.f)) g[j] = 0.f;
        xh[j] = (gamma && ok[j]) ? xhat[at] : 0.f;
        s1 += g[j];
        s2 = fmaf(g[j], xh[j], s2);
    }
    float db = 0.f;
    if (gamma) {
        s1 = wave_sum(s1);
        s2 = wave_sum(s2);
        const float a = cok ? gamma[c] * invstd[c] : 0.f, m1 = s1 / (float)B, m2 = s2 / (float)B;
#pragma unroll
        for (int j = 0; j < RB; ++j) g[j] = ok[j] ? a * (g[j] - m1 - xh[j] * m2) : 0.f;
        if (cok && lane == 0) { dgamma[c] = s2; dbeta[c] = s1; }
    }
#pragma unroll
    for (int j = 0; j < RB; ++j) {
        db += g[j];
        if (ok[j]) dz_out[(size_t)(lane + 64 * j) * Cout + c] = g[j];
        dzs[(lane + 64 * j) * FC_CS + wv] = g[j];
    }
    db = wave_sum(db);
    if (cok && lane == 0 && dbias) dbias[c] = db;
    __syncthreads();
    if (!dW) return;
    const int c0 = blockIdx.x * FC_CS;
    for (int k = t; k < Cin; k += FC_THREADS) {
        float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
        int b = 0;
        for (; b + 4 <= B; b += 4) {                                  // four independent loads in flight; rows added in order
            const float x0 = x[(size_t)b * Cin + k], x1 = x[(size_t)(b + 1) * Cin + k], x2 = x[(size_t)(b + 2) * Cin + k],
                        x3 = x[(size_t)(b + 3) * Cin + k];
            const float4 d0 = *reinterpret_cast<const float4 *>(dzs + b * FC_CS), d1 = *reinterpret_cast<const float4 *>(dzs + (b + 1) * FC_CS),
                         d2 = *reinterpret_cast<const float4 *>(dzs + (b + 2) * FC_CS), d3 = *reinterpret_cast<const float4 *>(dzs + (b + 3) * FC_CS);
            a0 = fmaf(d0.x, x0, a0); a1 = fmaf(d0.y, x0, a1); a2 = fmaf(d0.z, x0, a2); a3 = fmaf(d0.w, x0, a3);
            a0 = fmaf(d1.x, x1, a0); a1 = fmaf(d1.y, x1, a1); a2 = fmaf(d1.z, x1, a2); a3 = fmaf(d1.w, x1, a3);
            a0 = fmaf(d2.x, x2, a0); a1 = fmaf(d2.y, x2, a1); a2 = fmaf(d2.z, x2, a2); a3 = fmaf(d2.w, x2, a3);
            a0 = fmaf(d3.x, x3, a0); a1 = fmaf(d3.y, x3, a1); a2 = fmaf(d3.z, x3, a2); a3 = fmaf(d3.w, x3, a3);
        }
        for (; b < B; ++b) {
            const float x0 = x[(size_t)b * Cin + k];
            const float4 d0 = *reinterpret_cast<const float4 *>(dzs + b * FC_CS);
            a0 = fmaf(d0.x, x0, a0); a1 = fmaf(d0.y, x0, a1); a2 = fmaf(d0.z, x0, a2); a3 = fmaf(d0.w, x0, a3);
        }
        if (c0 + 0 < Cout) dW[(size_t)(c0 + 0) * Cin + k] = a0;
        if (c0 + 1 < Cout) dW[(size_t)(c0 + 1) * Cin + k] = a1;
        if (c0 + 2 < Cout) dW[(size_t)(c0 + 2) * Cin + k] = a2;
        if (c0 + 3 < Cout) dW[(size_t)(c0 + 3) * Cin + k] = a3;
    }
}

// dx[b][k] = sum_c dz[b][c] W[c][k]: workgroup = 64 input channels (a lane each) x 16 rows (four per wave); c in tiles of 32 through LDS
constexpr int FD_CT = 32, FD_DP = FD_CT + 4, FD_ROWS = 16;
__global__ __launch_bounds__(FC_THREADS) void fc_dx_kernel(const float *__restrict__ dz, const float *__restrict__ W, int B, int Cin, int Cout,
                                                           float *__restrict__ dx)
{
    __shared__ __attribute__((aligned(16))) float sm[FD_CT * 64 + FD_ROWS * FD_DP];
    float *ws = sm, *ds = sm + FD_CT * 64;
    const int t = threadIdx.x, lane = t & 63, wv = t >> 6;
    const int k0 = blockIdx.x * 64, r0 = blockIdx.y * FD_ROWS;
    float acc[4] = {0.f, 0.f, 0.f, 0.f};
    float4 wr[2], dr = make_float4(0.f, 0.f, 0.f, 0.f);            // the next tile, in flight during the products of this one
    auto fetch = [&](int c0) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int idx = t + i * FC_THREADS, cc = idx >> 4, q = idx & 15, k = k0 + q * 4;
            wr[i] = (c0 + cc < Cout && k < Cin) ? *reinterpret_cast<const float4 *>(W + (size_t)(c0 + cc) * Cin + k) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
        if (t < FD_ROWS * (FD_CT / 4)) {
            const int rr = t >> 3, q = t & 7, cq = c0 + q * 4;
            dr = (r0 + rr < B && cq < Cout) ? *reinterpret_cast<const float4 *>(dz + (size_t)(r0 + rr) * Cout + cq) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
    };
    fetch(0);
    for (int c0 = 0; c0 < Cout; c0 += FD_CT) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int idx = t + i * FC_THREADS;
            *reinterpret_cast<float4 *>(ws + (idx >> 4) * 64 + (idx & 15) * 4) = wr[i];
        }
        if (t < FD_ROWS * (FD_CT / 4)) *reinterpret_cast<float4 *>(ds + (t >> 3) * FD_DP + (t & 7) * 4) = dr;
        __syncthreads();
        if (c0 + FD_CT < Cout) fetch(c0 + FD_CT);
#pragma unroll
        for (int q = 0; q < FD_CT / 4; ++q) {
            const float w0 = ws[(q * 4 + 0) * 64 + lane], w1 = ws[(q * 4 + 1) * 64 + lane], w2 = ws[(q * 4 + 2) * 64 + lane],
                        w3 = ws[(q * 4 + 3) * 64 + lane];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const float4 d = *reinterpret_cast<const float4 *>(ds + (wv * 4 + i) * FD_DP + q * 4);
                acc[i] = fmaf(d.x, w0, acc[i]);
                acc[i] = fmaf(d.y, w1, acc[i]);
                acc[i] = fmaf(d.z, w2, acc[i]);
                acc[i] = fmaf(d.w, w3, acc[i]);
            }
        }
        __syncthreads();
    }
    const int k = k0 + lane;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int row = r0 + wv * 4 + i;
        if (row < B && k < Cin) dx[(size_t)row * Cin + k] = acc[i];
    }
}

static bool fc_shape_ok(int B, int Cin, int Cout) { return B >= 1 && B <= FC_MAXB && Cin >= 4 && Cin % 4 == 0 && Cout >= 4 && Cout % 4 == 0; }
static bool fc_aligned(const void *a, const void *b = nullptr, const void *c = nullptr)
{
    return ((reinterpret_cast<uintptr_t>(a) | reinterpret_cast<uintptr_t>(b) | reinterpret_cast<uintptr_t>(c)) & 15) == 0;
}

}  // namespace

/* Training forward of one FC layer of the heads (models/layers.py:123-166): y = act(BN(x W^T + bias)), B <= 128 rows, Cin % 4 == 0,
 * Cout % 4 == 0.  gamma == NULL: no normalisation (beta, running_*, xhat, invstd unused, may be NULL).  gamma != NULL: BatchNorm1d with
 * BATCH statistics (B >= 2; biased variance for the output, running_var takes the unbiased one; running_mean / running_var updated in
 * place with `momentum`, either may be NULL), xhat [B][Cout] and invstd [Cout] are kept for the backward.  relu != 0: ReLU. */
extern "C" int sonet_fc_bn_act_fwd_f32(const float *x, const float *W, const float *bias, const float *gamma, const float *beta,
                                       float *running_mean, float *running_var, float momentum, float eps, int relu, int B, int Cin,
                                       int Cout, float *y, float *xhat, float *invstd, sonet_stream_t stream)
{
    const char *what = "sonet_fc_bn_act_fwd_f32";
    SONET_REQUIRE(x && W && y, "%s: NULL pointer", what);
    SONET_REQUIRE(!gamma || (beta && xhat && invstd), "%s: BatchNorm needs beta, xhat and invstd", what);
    if (!fc_shape_ok(B, Cin, Cout) || (gamma && B < 2))
        return sonet::fail(SONET_ERR_UNSUPPORTED, "%s: shape B=%d Cin=%d Cout=%d not supported (B <= %d, channel counts multiples of 4)", what, B, Cin, Cout, FC_MAXB);
    if (!fc_aligned(x, W)) return sonet::fail(SONET_ERR_INVALID_ARG, "%s: x and W must be 16-byte aligned", what);
    const dim3 grid((unsigned)sonet::ceil_div(Cout, FC_CS)), block(FC_THREADS);
    hipStream_t st = sonet::as_stream(stream);
    if (B <= 64)
        hipLaunchKernelGGL(fc_bn_act_fwd_kernel<1>, grid, block, 0, st, x, W, bias, gamma, beta, running_mean, running_var, momentum, eps, relu, B, Cin, Cout, y, xhat, invstd);
    else
        hipLaunchKernelGGL(fc_bn_act_fwd_kernel<2>, grid, block, 0, st, x, W, bias, gamma, beta, running_mean, running_var, momentum, eps, relu, B, Cin, Cout, y, xhat, invstd);
    return sonet::launched(what);
}

/* Backward of the same layer up to its own parameters: gy [B][Cout] -> dz [B][Cout] (the gradient at the Linear's output: ReLU mask from
 * y, BatchNorm backward with batch statistics from xhat / invstd / gamma; gamma == NULL: mask only), dW [Cout][Cin] = dz^T x, dbias =
 * column sums of dz, dgamma, dbeta.  dW / dbias may be NULL (not wanted). */
extern "C" int sonet_fc_bn_act_bwd_f32(const float *gy, const float *y, const float *xhat, const float *invstd, const float *gamma,
                                       const float *x, int relu, int B, int Cin, int Cout, float *dz, float *dW, float *dbias,
                                       float *dgamma, float *dbeta, sonet_stream_t stream)
{
    const char *what = "sonet_fc_bn_act_bwd_f32";
    SONET_REQUIRE(gy && x && dz && (y || !relu), "%s: NULL pointer", what);
    SONET_REQUIRE(!gamma || (xhat && invstd && dgamma && dbeta), "%s: BatchNorm needs xhat, invstd, dgamma and dbeta", what);
    if (!fc_shape_ok(B, Cin, Cout)) return sonet::fail(SONET_ERR_UNSUPPORTED, "%s: shape B=%d Cin=%d Cout=%d not supported", what, B, Cin, Cout);
    const dim3 grid((unsigned)sonet::ceil_div(Cout, FC_CS)), block(FC_THREADS);
    hipStream_t st = sonet::as_stream(stream);
    if (B <= 64)
        hipLaunchKernelGGL(fc_bn_act_bwd_kernel<1>, grid, block, 0, st, gy, y, xhat, invstd, gamma, x, relu, B, Cin, Cout, dz, dW, dbias, dgamma, dbeta);
    else
        hipLaunchKernelGGL(fc_bn_act_bwd_kernel<2>, grid, block, 0, st, gy, y, xhat, invstd, gamma, x, relu, B, Cin, Cout, dz, dW, dbias, dgamma, dbeta);
    return sonet::launched(what);
}

/* dx [B][Cin] = dz [B][Cout] . W [Cout][Cin] (the input gradient of the Linear). */
extern "C" int sonet_fc_dx_f32(const float *dz, const float *W, int B, int Cin, int Cout, float *dx, sonet_stream_t stream)
{
    const char *what = "sonet_fc_dx_f32";
    SONET_REQUIRE(dz && W && dx, "%s: NULL pointer", what);
    if (!fc_shape_ok(B, Cin, Cout)) return sonet::fail(SONET_ERR_UNSUPPORTED, "%s: shape B=%d Cin=%d Cout=%d not supported", what, B, Cin, Cout);
    if (!fc_aligned(dz, W)) return sonet::fail(SONET_ERR_INVALID_ARG, "%s: dz and W must be 16-byte aligned", what);
    const dim3 grid((unsigned)sonet::ceil_div(Cin, 64), (unsigned)sonet::ceil_div(B, FD_ROWS)), block(FC_THREADS);
    hipLaunchKernelGGL(fc_dx_kernel, grid, block, 0, sonet::as_stream(stream), dz, W, B, Cin, Cout, dx);
    return sonet::launched(what);
}

extern "C" int sonet_fc_max_rows(void) { return FC_MAXB; }
