// som.hip -- SOM assignment, grouping, one-hot mask and node-kNN gather kernels for gfx950.
//
//   som_assign : util/som.py:237-269 (BatchSOM.query_topk) + node counts (models/networks.py:128)
//                + per-node coordinate sums (the numerator of the cluster mean, networks.py:140-142)
//   som_group  : models/networks.py:140-172 (cluster mean, centers, de-centre, concat normals)
//   som_mask   : util/som.py:254-265 (the one-hot B x kN x M mask, only when a caller asks for it)
//   knn_gather : models/operations.py:38-54
//
// The reference materialises B x 3 x N x M and two B x 3 x kN x M f32 temporaries (11.5 MB per cloud
// each at N=5000) to do this; here a point is read once (12 B), its k node ids are written once
// (4 B each) and the only cross-point state -- 64 counts and 192 sums per cloud -- lives in LDS.
//
// Arithmetic contract (bit-exact node sets): d = (dx*dx + dy*dy) + dz*dz with separate f32 multiplies
// and adds.  This file is compiled with -ffp-contract=off AND uses __fmul_rn/__fadd_rn so that no
// FMA contraction can change a comparison.  Selection is a k-deep insertion list kept in VGPRs,
// nodes visited in ascending id with strict '<', so ties keep the lower id and slots come out in
// ascending (distance, id) order -- the canonical order of torch.topk(sorted=True).
#include "common.hpp"
#include <stdlib.h>

namespace {

constexpr int SA_THREADS = 256;

__device__ __forceinline__ float sqdist(float px, float py, float pz, const float4 nd) {
    const float dx = __fsub_rn(px, nd.x), dy = __fsub_rn(py, nd.y), dz = __fsub_rn(pz, nd.z);
    return __fadd_rn(__fadd_rn(__fmul_rn(dx, dx), __fmul_rn(dy, dy)), __fmul_rn(dz, dz));
}

// LDS layout (dynamic): float4 nodes[M] | double sums[3][M] | unsigned cnt[M]
template <int KSEL>
__global__ __launch_bounds__(SA_THREADS) void som_assign_kernel(
    const float *__restrict__ x, const float *__restrict__ node, int N, int M,
    int32_t *__restrict__ min32, int64_t *__restrict__ min64, int32_t *__restrict__ count,
    double *__restrict__ sum_ws)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    float4 *nodes = reinterpret_cast<float4 *>(smem);
    double *sums = reinterpret_cast<double *>(smem + (size_t)M * sizeof(float4));
    unsigned *cnt = reinterpret_cast<unsigned *>(smem + (size_t)M * (sizeof(float4) + 3 * sizeof(double)));

    const int tid = threadIdx.x;
    const int b = blockIdx.y;
    const float *xb = x + (size_t)b * 3 * N;
    const float *nb = node + (size_t)b * 3 * M;
    for (int m = tid; m < M; m += SA_THREADS) {
        nodes[m] = make_float4(nb[m], nb[M + m], nb[2 * M + m], 0.f);
        sums[m] = 0.0; sums[M + m] = 0.0; sums[2 * M + m] = 0.0;
        cnt[m] = 0u;
    }
    __syncthreads();

    const int n = blockIdx.x * SA_THREADS + tid;
    if (n < N) {
        const float px = xb[n], py = xb[N + n], pz = xb[2 * (size_t)N + n];
        float bd[KSEL];
        int bi[KSEL];
#pragma unroll
        for (int s = 0; s < KSEL; ++s) { bd[s] = __builtin_inff(); bi[s] = 0; }
#pragma unroll 4
        for (int m = 0; m < M; ++m) {
            const float d = sqdist(px, py, pz, nodes[m]);
            bool c[KSEL];
#pragma unroll
            for (int s = 0; s < KSEL; ++s) c[s] = d < bd[s];
#pragma unroll
            for (int s = KSEL - 1; s >= 1; --s) {
                bd[s] = c[s - 1] ? bd[s - 1] : (c[s] ? d : bd[s]);
                bi[s] = c[s - 1] ? bi[s - 1] : (c[s] ? m : bi[s]);
            }
            bd[0] = c[0] ? d : bd[0];
            bi[0] = c[0] ? m : bi[0];
        }
        const size_t kN = (size_t)KSEL * N;
#pragma unroll
        for (int s = 0; s < KSEL; ++s) {
            const size_t o = (size_t)b * kN + (size_t)s * N + n;
            min32[o] = bi[s];
            if (min64 != nullptr) min64[o] = bi[s];
            atomicAdd(&cnt[bi[s]], 1u);
            atomicAdd(&sums[bi[s]], (double)px);
            atomicAdd(&sums[M + bi[s]], (double)py);
            atomicAdd(&sums[2 * M + bi[s]], (double)pz);
        }
    }
    __syncthreads();
    for (int m = tid; m < M; m += SA_THREADS) {
        const unsigned c = cnt[m];
        if (c != 0u) {
            atomicAdd(&count[(size_t)b * M + m], (int)c);
            double *ws = sum_ws + (size_t)b * 3 * M;
            unsafeAtomicAdd(&ws[m], sums[m]);
            unsafeAtomicAdd(&ws[M + m], sums[M + m]);
            unsafeAtomicAdd(&ws[2 * M + m], sums[2 * M + m]);
        }
    }
}

// ---- the same assignment, selection on packed keys ------------------------------------------------------------------------
// The insertion list above costs ~13 vector operations per (point, node) pair on top of the 8 of the exact distance.  Here the
// distance's bit pattern (d >= 0: it orders like the value) gives up its low IB bits to the node id, key = (bits(d) & ~mask) | m,
// and the KSEL + 1 smallest keys are kept by an unsigned min / median-of-three chain: v_and_or + v_min + KSEL x v_med3 = 5
// operations at k = 3, and the winners' ids are simply the keys' low bits -- no second pass, no distances kept.  Truncation can
// only reorder two candidates whose distances agree in all the kept bits (within 2^IB ulp of each other): a lane sees that as
// equal high parts among its KSEL + 1 smallest keys (the extra one guards the boundary of the list) and then -- like a lane
// whose list reaches +inf / NaN, where the reference leaves id 0 -- redoes its point with the exact insertion list.  Bit-exact
// with som_assign_kernel by construction; the slow branch runs for ~1e-4 of the points.
template <int KSEL, int IB>
__global__ __launch_bounds__(SA_THREADS) void som_assign_keys_kernel(
    const float *__restrict__ x, const float *__restrict__ node, int N, int M,
    int32_t *__restrict__ min32, int64_t *__restrict__ min64, int32_t *__restrict__ count,
    double *__restrict__ sum_ws)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    float4 *nodes = reinterpret_cast<float4 *>(smem);
    double *sums = reinterpret_cast<double *>(smem + (size_t)M * sizeof(float4));
    unsigned *cnt = reinterpret_cast<unsigned *>(smem + (size_t)M * (sizeof(float4) + 3 * sizeof(double)));
    constexpr unsigned IMASK = (1u << IB) - 1u;

    const int tid = threadIdx.x;
    const int b = blockIdx.y;
    const float *xb = x + (size_t)b * 3 * N;
    const float *nb = node + (size_t)b * 3 * M;
    for (int m = tid; m < M; m += SA_THREADS) {
        nodes[m] = make_float4(nb[m], nb[M + m], nb[2 * M + m], 0.f);
        sums[m] = 0.0; sums[M + m] = 0.0; sums[2 * M + m] = 0.0;
        cnt[m] = 0u;
    }
    __syncthreads();

    const int n = blockIdx.x * SA_THREADS + tid;
    if (n < N) {
        const float px = xb[n], py = xb[N + n], pz = xb[2 * (size_t)N + n];
        unsigned t[KSEL + 1];
#pragma unroll
        for (int s = 0; s <= KSEL; ++s) t[s] = 0xFFFFFFFFu;
        const unsigned hi_mask = ~IMASK;
        auto visit = [&](int m) {
            unsigned key;                                                   // (bits(d) & ~IMASK) | m in one instruction
            asm("v_and_or_b32 %0, %1, %2, %3" : "=v"(key) : "v"(__float_as_uint(sqdist(px, py, pz, nodes[m]))), "v"(hi_mask), "s"((unsigned)m));   // (one scalar operand per VOP3)
            // sorted insertion of key into t[0] <= t[1] <= ... : new t[s] = median(t[s-1], t[s], key), new t[0] = min(t[0], key)
#pragma unroll
            for (int s = KSEL; s >= 1; --s) {
                unsigned md;                                                // (hipcc does not form v_med3_u32 from the selects)
                asm("v_med3_u32 %0, %1, %2, %3" : "=v"(md) : "v"(t[s - 1]), "v"(t[s]), "v"(key));
                t[s] = md;
            }
            t[0] = key < t[0] ? key : t[0];
        };
        int m = 0;
        for (; m + 8 <= M; m += 8) {                                       // (unrolled by hand: the asm statements keep the pragma from doing it)
            visit(m); visit(m + 1); visit(m + 2); visit(m + 3); visit(m + 4); visit(m + 5); visit(m + 6); visit(m + 7);
        }
        for (; m < M; ++m) visit(m);
        int bi[KSEL];
        bool exact = t[KSEL - 1] >= 0x7F800000u;                            // the list reaches +inf / NaN: the reference keeps id 0 there
#pragma unroll
        for (int s = 0; s < KSEL; ++s) {
            bi[s] = (int)(t[s] & IMASK);
            exact = exact || ((t[s] & ~IMASK) == (t[s + 1] & ~IMASK));     // a pair the truncation may have ordered by id instead of by distance
        }
        if (exact) {
            float bd[KSEL];
#pragma unroll
            for (int s = 0; s < KSEL; ++s) { bd[s] = __builtin_inff(); bi[s] = 0; }
            for (int m = 0; m < M; ++m) {
                const float d = sqdist(px, py, pz, nodes[m]);
                bool c[KSEL];
#pragma unroll
                for (int s = 0; s < KSEL; ++s) c[s] = d < bd[s];
#pragma unroll
                for (int s = KSEL - 1; s >= 1; --s) {
                    bd[s] = c[s - 1] ? bd[s - 1] : (c[s] ? d : bd[s]);
                    bi[s] = c[s - 1] ? bi[s - 1] : (c[s] ? m : bi[s]);
                }
                bd[0] = c[0] ? d : bd[0];
                bi[0] = c[0] ? m : bi[0];
            }
        }
        const size_t kN = (size_t)KSEL * N;
#pragma unroll
        for (int s = 0; s < KSEL; ++s) {
            const size_t o = (size_t)b * kN + (size_t)s * N + n;
            min32[o] = bi[s];
            if (min64 != nullptr) min64[o] = bi[s];
            atomicAdd(&cnt[bi[s]], 1u);
            atomicAdd(&sums[bi[s]], (double)px);
            atomicAdd(&sums[M + bi[s]], (double)py);
            atomicAdd(&sums[2 * M + bi[s]], (double)pz);
        }
    }
    __syncthreads();
    for (int m = tid; m < M; m += SA_THREADS) {
        const unsigned c = cnt[m];
        if (c != 0u) {
            atomicAdd(&count[(size_t)b * M + m], (int)c);
            double *ws = sum_ws + (size_t)b * 3 * M;
            unsafeAtomicAdd(&ws[m], sums[m]);
            unsafeAtomicAdd(&ws[M + m], sums[M + m]);
            unsafeAtomicAdd(&ws[2 * M + m], sums[2 * M + m]);
        }
    }
}

constexpr int SG_THREADS = 256;
constexpr int SG_PER_THREAD = 4;

__global__ __launch_bounds__(SG_THREADS) void som_group_kernel(
    const float *__restrict__ x, const float *__restrict__ sn, const int32_t *__restrict__ min32,
    const int32_t *__restrict__ count, const double *__restrict__ sum_ws, int N, int M, int k,
    float *__restrict__ som_node, int32_t *__restrict__ row_max, float *__restrict__ centers,
    float *__restrict__ x_dec, float *__restrict__ x_aug)
{
    extern __shared__ __attribute__((aligned(16))) float mean[];  // [3][M]
    const int tid = threadIdx.x;
    const int b = blockIdx.y;
    const double *ws = sum_ws + (size_t)b * 3 * M;
    for (int m = tid; m < M; m += SG_THREADS) {
        const int c = count[(size_t)b * M + m];
        const float denom = __fadd_rn((float)c, 1e-5f);          // networks.py:142
        const float mx = __fdiv_rn((float)ws[m], denom);
        const float my = __fdiv_rn((float)ws[M + m], denom);
        const float mz = __fdiv_rn((float)ws[2 * M + m], denom);
        mean[m] = mx; mean[M + m] = my; mean[2 * M + m] = mz;
        if (blockIdx.x == 0) {
            if (som_node != nullptr) {
                float *o = som_node + (size_t)b * 3 * M;
                o[m] = mx; o[M + m] = my; o[2 * M + m] = mz;
            }
            if (row_max != nullptr) row_max[(size_t)b * M + m] = c > 0;
        }
    }
    __syncthreads();
    if (centers == nullptr && x_dec == nullptr && x_aug == nullptr) return;

    const size_t kN = (size_t)k * N;
    const float *xb = x + (size_t)b * 3 * N;
    const float *snb = sn ? sn + (size_t)b * 3 * N : nullptr;
    const int32_t *ib = min32 + (size_t)b * kN;
    const size_t j0 = (size_t)blockIdx.x * (SG_THREADS * SG_PER_THREAD) + tid;
#pragma unroll
    for (int i = 0; i < SG_PER_THREAD; ++i) {
        const size_t j = j0 + (size_t)i * SG_THREADS;
        if (j >= kN) break;
        const int n = (int)(j % (size_t)N);
        const int m = ib[j];
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const float ctr = mean[c * M + m];
            const float d = __fsub_rn(xb[(size_t)c * N + n], ctr);
            const size_t o = ((size_t)b * 3 + c) * kN + j;
            if (centers != nullptr) centers[o] = ctr;
            if (x_dec != nullptr) x_dec[o] = d;
            if (x_aug != nullptr) {
                x_aug[((size_t)b * 6 + c) * kN + j] = d;
                x_aug[((size_t)b * 6 + 3 + c) * kN + j] = snb[(size_t)c * N + n];
            }
        }
    }
}

__global__ __launch_bounds__(256) void som_mask_kernel(const int32_t *__restrict__ min32, int32_t *__restrict__ mask,
                                                        long long rows, int M)
{
    // one thread per 4 consecutive m of one (b, j) row when M % 4 == 0, else per element
    const long long t = (long long)blockIdx.x * 256 + threadIdx.x;
    if ((M & 3) == 0) {
        const int q = M >> 2;
        const long long row = t / q;
        if (row >= rows) return;
        const int m0 = (int)(t - row * q) << 2;
        const int id = min32[row];
        reinterpret_cast<int4 *>(mask)[t] = make_int4(id == m0, id == m0 + 1, id == m0 + 2, id == m0 + 3);
    } else {
        const long long row = t / M;
        if (row >= rows) return;
        mask[t] = min32[row] == (int)(t - row * M);
    }
}

__global__ __launch_bounds__(256) void knn_gather_kernel(const float *__restrict__ x, const int64_t *__restrict__ I,
                                                          float *__restrict__ out, int C, int M, int K, long long total)
{
    const long long t = (long long)blockIdx.x * 256 + threadIdx.x;
    if (t >= total) return;
    const int MK = M * K;
    const long long bc = t / MK;                 // b*C + c
    const int mk = (int)(t - bc * MK);
    const long long b = bc / C;
    const long long id = I[b * MK + mk];
    out[t] = ((unsigned long long)id < (unsigned long long)M) ? x[bc * M + id] : 0.f;
}

// KNNModule input in one pass (models/layers.py:313-350): out[b][0:3][m][k] = coord[b][:, I[b][m][k]] - center[b][:, m],
// out[b][3:3+C][m][k] = feat[b][:, I[b][m][k]]; center = mean over the K neighbours ("avg", sequential f32 sum / K) or
// the node itself ("center").  Replaces two gathers, a mean, a subtraction and a 387-channel concat.
__global__ __launch_bounds__(256) void knn_group_kernel(const float *__restrict__ coord, const float *__restrict__ feat,
                                                         const int64_t *__restrict__ I, int C, int M, int K, int avg,
                                                         float *__restrict__ center, float *__restrict__ out)
{
    // one thread = 4 consecutive (m, k) positions of one (b, c) row: 32-byte index loads, 16-byte stores
    const int MK = M * K, CC = 3 + C;
    const int mk0 = (blockIdx.x * 256 + threadIdx.x) * 4;
    if (mk0 >= MK) return;
    const int c = blockIdx.y, b = blockIdx.z;
    const int64_t *Ib = I + (size_t)b * MK;
    float *o = out + ((size_t)b * CC + c) * MK;
    const float *src = c >= 3 ? feat + ((size_t)b * C + (c - 3)) * M : coord + ((size_t)b * 3 + c) * M;
    float v[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int mk = mk0 + q;
        float val = 0.f;
        if (mk < MK) {
            const long long id = Ib[mk];
            val = ((unsigned long long)id < (unsigned long long)M) ? src[id] : 0.f;
            if (c < 3) {
                const int m = mk / K;
                float ctr;
                if (avg) {
                    float sum = 0.f;
                    for (int k = 0; k < K; ++k) {
                        const long long ik = Ib[m * K + k];
                        sum += ((unsigned long long)ik < (unsigned long long)M) ? src[ik] : 0.f;
                    }
                    ctr = sum / (float)K;
                } else {
                    ctr = src[m];
                }
                if (mk - m * K == 0) center[((size_t)b * 3 + c) * M + m] = ctr;
                val -= ctr;
            }
        }
        v[q] = val;
    }
    if (mk0 + 3 < MK && (MK & 3) == 0) *reinterpret_cast<float4 *>(o + mk0) = make_float4(v[0], v[1], v[2], v[3]);
    else
        for (int q = 0; q < 4 && mk0 + q < MK; ++q) o[mk0 + q] = v[q];
}

// out[row][m] = max over k of x[row][k * M + m] (k-major neighbourhood planes), NaN-propagating like torch.max
__global__ __launch_bounds__(256) void planes_max_kernel(const float *__restrict__ x, float *__restrict__ out, int K, int M, long long total)
{
    const long long t = (long long)blockIdx.x * 256 + threadIdx.x;       // row * M + m
    if (t >= total) return;
    const long long row = t / M;
    const int m = (int)(t - row * M);
    const float *p = x + row * (long long)K * M + m;
    float best = p[0];
    for (int k = 1; k < K; ++k) {
        const float v = p[(long long)k * M];
        best = (v > best || v != v) ? v : best;
    }
    out[t] = best;
}

// KNNModule input for the gathering layer (sonet_pointmlp_h3_gather_f32): nothing but the 3 de-centred coordinate rows is
// materialised, K-MAJOR (column k * M + m), together with the int32 gather index of every column (-1: index out of range,
// reads as 0 like knn_group) -- the 384 feature rows are gathered by the layer's operand loads.
__global__ __launch_bounds__(256) void knn_prepare_kernel(const float *__restrict__ coord, const int64_t *__restrict__ I, int M, int K, int avg,
                                                           float *__restrict__ center, float *__restrict__ dec, int32_t *__restrict__ gidx,
                                                           long long total)
{
    const long long t = (long long)blockIdx.x * 256 + threadIdx.x;       // (b * K + k) * M + m
    if (t >= total) return;
    const int m = (int)(t % M);
    const long long bk = t / M;
    const int k = (int)(bk % K);
    const long long b = bk / K;
    const int64_t *Ib = I + (b * M + m) * K;
    const float *cb = coord + b * 3 * M;
    const long long id = Ib[k];
    const bool ok = (unsigned long long)id < (unsigned long long)M;
    gidx[t] = ok ? (int32_t)id : -1;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        const float *src = cb + c * M;
        float ctr;
        if (avg) {
            float sum = 0.f;
            for (int kk = 0; kk < K; ++kk) {
                const long long ik = Ib[kk];
                sum += ((unsigned long long)ik < (unsigned long long)M) ? src[ik] : 0.f;
            }
            ctr = sum / (float)K;
        } else {
            ctr = src[m];
        }
        if (k == 0) center[(b * 3 + c) * M + m] = ctr;
        dec[(b * 3 + c) * (long long)K * M + (long long)k * M + m] = (ok ? src[id] : 0.f) - ctr;
    }
}

// Self k-NN of the SOM nodes (the node_knn_I table the reference builds with faiss on the host,
// data/modelnet_shrec_loader.py:116-150,257-259, and KNNModule's fallback, models/layers.py:333-337): for every node the
// K nearest nodes (itself first), ascending (distance, index), distance (dx*dx + dy*dy) + dz*dz.  One thread per
// (cloud, node); the K best live in registers (static compare-exchange chain, K <= 16).
constexpr int KS_MAX = 16;
__global__ __launch_bounds__(256) void knn_self_kernel(const float *__restrict__ node, int64_t *__restrict__ knn_I,
                                                        int M, int K, long long total)
{
    const long long t = (long long)blockIdx.x * 256 + threadIdx.x;       // b * M + m
    if (t >= total) return;
    const long long b = t / M;
    const int m = (int)(t - b * M);
    const float *nb = node + b * 3 * M;
    const float qx = nb[m], qy = nb[M + m], qz = nb[2 * M + m];
    float bd[KS_MAX];
    int bi[KS_MAX];
#pragma unroll
    for (int j = 0; j < KS_MAX; ++j) { bd[j] = __builtin_inff(); bi[j] = 0x7FFFFFFF; }
    for (int i = 0; i < M; ++i) {
        const float dx = __fsub_rn(qx, nb[i]), dy = __fsub_rn(qy, nb[M + i]), dz = __fsub_rn(qz, nb[2 * M + i]);
        float d = __fadd_rn(__fadd_rn(__fmul_rn(dx, dx), __fmul_rn(dy, dy)), __fmul_rn(dz, dz));
        int id = i;
#pragma unroll
        for (int j = 0; j < KS_MAX; ++j) {                               // insert, pushing the larger one down the chain
            const bool lt = d < bd[j] || (d == bd[j] && id < bi[j]);
            const float td = lt ? bd[j] : d;
            const int ti = lt ? bi[j] : id;
            bd[j] = lt ? d : bd[j];
            bi[j] = lt ? id : bi[j];
            d = td; id = ti;
        }
    }
#pragma unroll
    for (int j = 0; j < KS_MAX; ++j)
        if (j < K) knn_I[t * K + j] = bi[j];
}

// out[row] = max over the K contiguous values of the row (NaN wins, as torch.amax): the neighbourhood max of
// KNNModule (K = 9) and the global max over the nodes (K = M = 64), models/layers.py:365, models/networks.py:197.
template <int TPR>   // threads per row (power of two <= 64): each reads a strided share of the row, then a shuffle tree
__global__ __launch_bounds__(256) void lastdim_max_kernel(const float *__restrict__ x, float *__restrict__ out, int K, long long rows)
{
    const long long t = (long long)blockIdx.x * 256 + threadIdx.x;
    const long long r = t / TPR;
    const int q = (int)(t % TPR);
    const bool live = r < rows;
    const float *p = x + (live ? r : 0) * K;
    float m = p[q < K ? q : 0];
    for (int k = q + TPR; k < K; k += TPR) {
        const float v = p[k];
        m = (v > m || v != v) ? v : m;
    }
#pragma unroll
    for (int off = TPR >> 1; off > 0; off >>= 1) {
        const float v = __shfl_xor(m, off, 64);
        m = (v > m || v != v) ? v : m;
    }
    if (live && q == 0) out[r] = m;
}

// out[b][c][j] = feat[b][c][ idx[b][j] ]   (models/segmenter.py:90-98: node features broadcast back to the
// kN point copies).  One thread per 4 consecutive j of one (b, c) row: coalesced 16-byte stores, the 256-byte
// feature row and the id row stay in L1/L2.
__global__ __launch_bounds__(256) void node_gather_kernel(const float *__restrict__ feat, const int32_t *__restrict__ idx,
                                                           float *__restrict__ out, int C, int M, int kN, long long total4)
{
    const long long t = (long long)blockIdx.x * 256 + threadIdx.x;
    if (t >= total4) return;
    const int q = (kN + 3) >> 2;                  // 4-wide groups per row
    const long long bc = t / q;
    const int j0 = (int)(t - bc * q) << 2;
    const long long b = bc / C;
    const float *frow = feat + bc * M;
    const int32_t *irow = idx + b * kN;
    float v[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const int j = j0 + e;
        const int id = j < kN ? irow[j] : 0;
        v[e] = ((unsigned)id < (unsigned)M) ? frow[id] : 0.f;
    }
    float *orow = out + bc * kN + j0;
    if (j0 + 3 < kN && ((reinterpret_cast<uintptr_t>(orow) & 15) == 0)) {
        *reinterpret_cast<float4 *>(orow) = make_float4(v[0], v[1], v[2], v[3]);
    } else {
#pragma unroll
        for (int e = 0; e < 4; ++e) if (j0 + e < kN) orow[e] = v[e];
    }
}

// Node-sorted grouping for the fused eval path: the kN point copies of a cloud are counting-sorted by node id, so a
// 128-point tile of the fused first PointNet spans only a couple of nodes and its per-node max-pool epilogue is a
// wave-level reduction plus a handful of atomics.  Same arithmetic as som_group (mean = sum/(count+1e-5), x - mean).
// Order inside a node is arbitrary (a max-pool does not care); pos0[b] = sorted position of original copy j = 0
// (its features are the reference's fallback for empty nodes: gather index 0, models/networks.py:185).
__global__ __launch_bounds__(SG_THREADS) void som_sort_group_kernel(
    const float *__restrict__ x, const float *__restrict__ sn, const int32_t *__restrict__ min32,
    const int32_t *__restrict__ count, const double *__restrict__ sum_ws, int N, int M, int k,
    float *__restrict__ som_node, int32_t *__restrict__ row_max, float *__restrict__ x_aug_sorted,
    int32_t *__restrict__ ids_sorted, int32_t *__restrict__ pos0, int32_t *__restrict__ cursor, int32_t *__restrict__ node_off)
{
    extern __shared__ __attribute__((aligned(16))) float smem_f[];          // mean[3][M] | offs[M] | hist[M] | base[M]
    float *mean = smem_f;
    int *offs = reinterpret_cast<int *>(smem_f + 3 * M);
    int *hist = offs + M;
    int *base = hist + M;
    const int tid = threadIdx.x;
    const int b = blockIdx.y;
    const double *ws = sum_ws + (size_t)b * 3 * M;
    for (int m = tid; m < M; m += SG_THREADS) {
        const int c = count[(size_t)b * M + m];
        const float denom = __fadd_rn((float)c, 1e-5f);
        const float mx = __fdiv_rn((float)ws[m], denom), my = __fdiv_rn((float)ws[M + m], denom), mz = __fdiv_rn((float)ws[2 * M + m], denom);
        mean[m] = mx; mean[M + m] = my; mean[2 * M + m] = mz;
        hist[m] = 0;
        if (blockIdx.x == 0) {
            if (som_node != nullptr) { float *o = som_node + (size_t)b * 3 * M; o[m] = mx; o[M + m] = my; o[2 * M + m] = mz; }
            if (row_max != nullptr) row_max[(size_t)b * M + m] = c > 0;
        }
    }
    if (tid == 0) {                                                          // exclusive prefix of the node counts
        int acc = 0;
        for (int m = 0; m < M; ++m) {
            offs[m] = acc;
            if (blockIdx.x == 0 && node_off != nullptr) node_off[(size_t)b * M + m] = acc;
            acc += count[(size_t)b * M + m];
        }
    }
    __syncthreads();
    const size_t kN = (size_t)k * N;
    const int32_t *ib = min32 + (size_t)b * kN;
    const size_t j0 = (size_t)blockIdx.x * (SG_THREADS * SG_PER_THREAD) + tid;
    int id[SG_PER_THREAD], rk[SG_PER_THREAD];
#pragma unroll
    for (int i = 0; i < SG_PER_THREAD; ++i) {
        const size_t j = j0 + (size_t)i * SG_THREADS;
        id[i] = j < kN ? ib[j] : -1;
        rk[i] = id[i] >= 0 ? atomicAdd(&hist[id[i]], 1) : 0;
    }
    __syncthreads();
    for (int m = tid; m < M; m += SG_THREADS) base[m] = hist[m] ? atomicAdd(&cursor[(size_t)b * M + m], hist[m]) : 0;
    __syncthreads();
    // Phase A ends here: only the source copy index (parked, as raw bits, in channel plane 5 of x_aug_sorted, where
    // the same thread of phase B overwrites it with the value) and the node id go to their sorted position.  Writing the
    // six channels from here scattered 4-byte stores over six planes: 80 MB of HBM writes for 27 MB of payload.
    int32_t *src_plane = reinterpret_cast<int32_t *>(x_aug_sorted + ((size_t)b * 6 + 5) * kN);
#pragma unroll
    for (int i = 0; i < SG_PER_THREAD; ++i) {
        const size_t j = j0 + (size_t)i * SG_THREADS;
        if (id[i] < 0) continue;
        const int m = id[i];
        const size_t pos = (size_t)(offs[m] + base[m] + rk[i]);
        ids_sorted[(size_t)b * kN + pos] = m;
        src_plane[pos] = (int32_t)j;
        if (j == 0) pos0[b] = (int)pos;
    }
}

// Phase B: one thread per sorted position: coalesced stores of the six channels; the gathers hit the cloud's 120 KB of
// x / sn in L2.  (Four positions per thread with 16-byte index loads and stores measured SLOWER: 26.7 vs 22.1 us at B = 64 --
// a quarter of the threads to hide the load -> gather -> store chain; r02zc.)
__global__ __launch_bounds__(256) void som_sort_fill_kernel(
    const float *__restrict__ x, const float *__restrict__ sn, const int32_t *__restrict__ count,
    const double *__restrict__ sum_ws, const int32_t *__restrict__ ids_sorted, int N, int M, int k, float *__restrict__ x_aug_sorted)
{
    extern __shared__ __attribute__((aligned(16))) float smem_f[];          // mean[3][M]
    float *mean = smem_f;
    const int b = blockIdx.y;
    const double *ws = sum_ws + (size_t)b * 3 * M;
    for (int m = threadIdx.x; m < M; m += 256) {
        const float denom = __fadd_rn((float)count[(size_t)b * M + m], 1e-5f);
        mean[m] = __fdiv_rn((float)ws[m], denom);
        mean[M + m] = __fdiv_rn((float)ws[M + m], denom);
        mean[2 * M + m] = __fdiv_rn((float)ws[2 * M + m], denom);
    }
    __syncthreads();
    const size_t kN = (size_t)k * N;
    const size_t pos = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (pos >= kN) return;
    float *ob = x_aug_sorted + (size_t)b * 6 * kN;
    const int j = reinterpret_cast<const int32_t *>(ob + 5 * kN)[pos];
    const int m = ids_sorted[(size_t)b * kN + pos];
    const int n = j % N;
    const float *xb = x + (size_t)b * 3 * N;
    const float *snb = sn + (size_t)b * 3 * N;
#pragma unroll
    for (int c = 0; c < 3; ++c) ob[(size_t)c * kN + pos] = __fsub_rn(xb[(size_t)c * N + n], mean[c * M + m]);
#pragma unroll
    for (int c = 0; c < 3; ++c) ob[(size_t)(3 + c) * kN + pos] = snb[(size_t)c * N + n];
}


// ---- the SOM stage of the no-grad pooled path in TWO launches (assign + rank + partial sums | totals + node-sorted fill) ----------
// The five-launch form (clear, som_assign_keys, clear, som_sort_group, som_sort_fill) is launch- and atomics-bound at B = 64:
// twelve LDS atomics per point (three counters and nine double-precision coordinate sums) are most of som_assign's 16 us, and the
// sorted positions come from a global cursor that needs its own clear.  Here
//   K1  a workgroup owns SP_T = 512 consecutive points (two per thread): node ids as above (packed keys, exact redo), ONE integer
//       LDS atomic per point copy -- its return value is the copy's RANK among the workgroup's copies of that node --, then the
//       copies' coordinates are staged in LDS at (node start + rank) and 3M threads add up their node's run in double precision,
//       in slot order.  Outputs: ids, ranks, and per-workgroup partial counts / sums by PLAIN stores: nothing to clear.
//   K2  every workgroup of a cloud adds up the (few) partials: totals, cluster means (sum / (count + 1e-5), networks.py:140-142),
//       node offsets, and the start of ITS run inside every node = the partial counts of the workgroups before it.  The sorted
//       position of a copy is node offset + run start + rank: no cursor, no atomics.  The six channels and the ids go to their
//       sorted positions through an LDS transpose (workgroup-local node order), so that consecutive lanes store consecutive
//       positions of a run (~48 copies) instead of 4-byte stores scattered over six planes.
constexpr int SP_THREADS = 256;
constexpr int SP_PPT = 2;                                      // points per thread
constexpr int SP_T = SP_THREADS * SP_PPT;                      // points per workgroup
constexpr int SP_KMAX = 4;

// exclusive prefix sums over M (<= 1024) integers in LDS by the FIRST WAVE (64 per step, carry in a register): a thread-0 loop is a
// 64-deep dependent LDS chain (~1.5 us per prefix at M = 64)
__device__ __forceinline__ void lds_exclusive_scan(const int *in, int *out, int M, int tid)
{
    if (tid >= 64) return;
    int carry = 0;
    for (int m0 = 0; m0 < M; m0 += 64) {
        const int m = m0 + tid;
        const int v = m < M ? in[m] : 0;
        int incl = v;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) { const int t = __shfl_up(incl, o, 64); if (tid >= o) incl += t; }
        if (m < M) out[m] = carry + incl - v;
        carry += __shfl(incl, 63, 64);
    }
}

template <int KSEL, int IB>
__device__ __forceinline__ void som_select_keys(float px, float py, float pz, const float4 *nodes, int M, int (&bi)[KSEL])
{
    constexpr unsigned IMASK = (1u << IB) - 1u;
    unsigned t[KSEL + 1];
#pragma unroll
    for (int s = 0; s <= KSEL; ++s) t[s] = 0xFFFFFFFFu;
    const unsigned hi_mask = ~IMASK;
    auto visit = [&](int m) {
        unsigned key;
        asm("v_and_or_b32 %0, %1, %2, %3" : "=v"(key) : "v"(__float_as_uint(sqdist(px, py, pz, nodes[m]))), "v"(hi_mask), "s"((unsigned)m));
#pragma unroll
        for (int s = KSEL; s >= 1; --s) {
            unsigned md;
            asm("v_med3_u32 %0, %1, %2, %3" : "=v"(md) : "v"(t[s - 1]), "v"(t[s]), "v"(key));
            t[s] = md;
        }
        t[0] = key < t[0] ? key : t[0];
    };
    int m = 0;
    for (; m + 8 <= M; m += 8) { visit(m); visit(m + 1); visit(m + 2); visit(m + 3); visit(m + 4); visit(m + 5); visit(m + 6); visit(m + 7); }
    for (; m < M; ++m) visit(m);
    bool exact = t[KSEL - 1] >= 0x7F800000u;                                // the list reaches +inf / NaN: the reference keeps id 0 there
#pragma unroll
    for (int s = 0; s < KSEL; ++s) {
        bi[s] = (int)(t[s] & IMASK);
        exact = exact || ((t[s] & ~IMASK) == (t[s + 1] & ~IMASK));         // a pair the truncation may have ordered by id instead of by distance
    }
    if (exact) {
        float bd[KSEL];
#pragma unroll
        for (int s = 0; s < KSEL; ++s) { bd[s] = __builtin_inff(); bi[s] = 0; }
        for (int mm = 0; mm < M; ++mm) {
            const float d = sqdist(px, py, pz, nodes[mm]);
            bool c[KSEL];
#pragma unroll
            for (int s = 0; s < KSEL; ++s) c[s] = d < bd[s];
#pragma unroll
            for (int s = KSEL - 1; s >= 1; --s) {
                bd[s] = c[s - 1] ? bd[s - 1] : (c[s] ? d : bd[s]);
                bi[s] = c[s - 1] ? bi[s - 1] : (c[s] ? mm : bi[s]);
            }
            bd[0] = c[0] ? d : bd[0];
            bi[0] = c[0] ? mm : bi[0];
        }
    }
}

// (Two points per thread with the distances in packed f32 -- v_pk_add_f32 / v_pk_mul_f32, bit-identical node sets -- measured no
// faster: 21.1 vs 19.5 us at B = 64; the scalar form stays.)
// LDS (dynamic): float4 nodes[M] | int cnt[M] | int lstart[M] | float stage[3][KSEL * SP_T]
template <int KSEL, int IB>
__global__ __launch_bounds__(SP_THREADS) void som_assign_rank_kernel(
    const float *__restrict__ x, const float *__restrict__ node, int N, int M, int nW,
    int32_t *__restrict__ min32, int64_t *__restrict__ min64, uint16_t *__restrict__ rank16,
    int32_t *__restrict__ cnt_part /*[B][nW][M]*/, double *__restrict__ sum_part /*[B][nW][3][M]*/,
    int det /*ranks independent of the arrival order of the LDS atomics (+ 4 M ints of LDS)*/)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    float4 *nodes = reinterpret_cast<float4 *>(smem);
    int *cnt = reinterpret_cast<int *>(smem + (size_t)M * sizeof(float4));
    int *lstart = cnt + M;
    float *stage = reinterpret_cast<float *>(lstart + M);
    constexpr int CAP = KSEL * SP_T;
    const int tid = threadIdx.x, w = blockIdx.x, b = blockIdx.y;
    const float *xb = x + (size_t)b * 3 * N;
    const float *nb = node + (size_t)b * 3 * M;
    for (int m = tid; m < M; m += SP_THREADS) {
        nodes[m] = make_float4(nb[m], nb[M + m], nb[2 * M + m], 0.f);
        cnt[m] = 0;
    }
    __syncthreads();
    const size_t kN = (size_t)KSEL * N;
    float pc[SP_PPT][3];
    int bi[SP_PPT][KSEL], rk[SP_PPT][KSEL];
#pragma unroll
    for (int p = 0; p < SP_PPT; ++p) {
        const int n = w * SP_T + p * SP_THREADS + tid;
        if (n < N) {
            pc[p][0] = xb[n]; pc[p][1] = xb[N + n]; pc[p][2] = xb[2 * (size_t)N + n];
            som_select_keys<KSEL, IB>(pc[p][0], pc[p][1], pc[p][2], nodes, M, bi[p]);
            if (!det) {
#pragma unroll
                for (int s = 0; s < KSEL; ++s) rk[p][s] = atomicAdd(&cnt[bi[p][s]], 1);   // the copy's rank among this workgroup's copies of the node
            }
        }
    }
    if (det) {
        // Ranks that do not depend on the arrival order of LDS atomics (the f32-class TRAINING forward runs its first PointNet on the sorted
        // copy: BatchNorm's batch sums must see the same column order in every run).  Order inside the workgroup: wave, then slot (p, s),
        // then lane.  Per slot a wave walks the distinct node ids among its lanes: the lanes of an id take the wave's running count of
        // that node plus the number of lower lanes with the same id, one lane adds the group's size.  The waves' counts are prefixed
        // after a barrier.  (~40 turns per slot on 64 nodes: the training path only; the no-grad forward keeps the atomics.)
        int *whist = reinterpret_cast<int *>(stage + 3 * CAP);              // [waves][M], zeroed below before use
        constexpr int NWAVE = SP_THREADS / 64;
        for (int t = tid; t < NWAVE * M; t += SP_THREADS) whist[t] = 0;
        __syncthreads();
        const int wave = tid >> 6, lane = tid & 63;
        const unsigned long long lower = (1ull << lane) - 1ull;
#pragma unroll
        for (int p = 0; p < SP_PPT; ++p) {
            const bool valid = w * SP_T + p * SP_THREADS + tid < N;
#pragma unroll
            for (int s = 0; s < KSEL; ++s) {
                const int id = valid ? bi[p][s] : -1;
                unsigned long long rem = __ballot(id >= 0);
                int r = 0;
                while (rem != 0ull) {
                    const int l0 = __builtin_ctzll(rem);
                    const int id0 = __builtin_amdgcn_readlane(id, l0);
                    const unsigned long long grp = __ballot(id == id0);
                    rem &= ~grp;
                    const int basev = whist[wave * M + id0];
                    if (id == id0) r = basev + __builtin_popcountll(grp & lower);
                    if (lane == l0) whist[wave * M + id0] = basev + __builtin_popcountll(grp);
                }
                rk[p][s] = r;
            }
        }
        __syncthreads();
#pragma unroll
        for (int p = 0; p < SP_PPT; ++p) {
            if (w * SP_T + p * SP_THREADS + tid < N) {
#pragma unroll
                for (int s = 0; s < KSEL; ++s)
                    for (int w2 = 0; w2 < wave; ++w2) rk[p][s] += whist[w2 * M + bi[p][s]];
            }
        }
        for (int m = tid; m < M; m += SP_THREADS) {
            int c = 0;
            for (int w2 = 0; w2 < NWAVE; ++w2) c += whist[w2 * M + m];
            cnt[m] = c;
        }
    }
#pragma unroll
    for (int p = 0; p < SP_PPT; ++p) {
        const int n = w * SP_T + p * SP_THREADS + tid;
        if (n < N) {
#pragma unroll
            for (int s = 0; s < KSEL; ++s) {
                const size_t o = (size_t)b * kN + (size_t)s * N + n;
                min32[o] = bi[p][s];
                if (min64 != nullptr) min64[o] = bi[p][s];
                rank16[o] = (uint16_t)rk[p][s];
            }
        }
    }
    __syncthreads();
    lds_exclusive_scan(cnt, lstart, M, tid);                                // node starts of the workgroup-local order
    __syncthreads();
#pragma unroll
    for (int p = 0; p < SP_PPT; ++p) {
        const int n = w * SP_T + p * SP_THREADS + tid;
        if (n < N) {
#pragma unroll
            for (int s = 0; s < KSEL; ++s) {
                const int slot = lstart[bi[p][s]] + rk[p][s];
                stage[slot] = pc[p][0]; stage[CAP + slot] = pc[p][1]; stage[2 * CAP + slot] = pc[p][2];
            }
        }
    }
    __syncthreads();
    for (int t = tid; t < 3 * M; t += SP_THREADS) {                         // one thread per (coordinate, node): its run, in slot order
        const int c = t / M, m = t - c * M;
        const float *sp = stage + c * CAP + lstart[m];
        const int len = cnt[m];
        double acc = 0.0;
        for (int i = 0; i < len; ++i) acc += (double)sp[i];
        sum_part[(((size_t)b * nW + w) * 3 + c) * M + m] = acc;
        if (c == 0) cnt_part[((size_t)b * nW + w) * M + m] = len;
    }
}

// LDS (dynamic): double tsum[3][M] | float mean[3][M] | int noff[M] | int base[M] | int lstart[M] | int tot[M] | int mine[M]
//                | int gpos[k * SP_T] | float buf[7][k * SP_T]
// Optional rider of som_sort_fill2_kernel: the index / coordinate side of KNNModule on the flat column axis of the node-level stage
// (node_stage.hip, knn_stage_prepare_kernel -- same records, same arithmetic), computed by the cloud's last workgroup from the cluster
// means it already holds in LDS: the no-grad forward needs no launch for it.
struct KnnPrep {
    const int64_t *I;                     // [B][M][KI] neighbour table (NULL: no rider)
    int KI, K, avg, G;
    long long BM, Lm;
    float *center;                        // [B][3][M]
    uint4 *center_p16;                    // one-chunk P16 panel, Lm columns
    int4 *rec;                            // [Lp] (source column | -1 | -2, three de-centred coordinates)
};

__global__ __launch_bounds__(SP_THREADS) void som_sort_fill2_kernel(
    const float *__restrict__ x, const float *__restrict__ sn, const int32_t *__restrict__ min32, const uint16_t *__restrict__ rank16,
    const int32_t *__restrict__ cnt_part, const double *__restrict__ sum_part, int N, int M, int k, int nW,
    int32_t *__restrict__ count, double *__restrict__ sum_ws, float *__restrict__ som_node, int32_t *__restrict__ row_max,
    float *__restrict__ x_aug_sorted, int32_t *__restrict__ ids_sorted, int32_t *__restrict__ pos0, int32_t *__restrict__ node_off,
    const KnnPrep kp)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem2[];
    double *tsum = reinterpret_cast<double *>(smem2);
    float *mean = reinterpret_cast<float *>(tsum + 3 * M);
    int *noff = reinterpret_cast<int *>(mean + 3 * M);
    int *base = noff + M, *lstart = base + M, *tot = lstart + M, *mine = tot + M;
    const int cap = k * SP_T;
    int *gpos = mine + M;
    float *buf = reinterpret_cast<float *>(gpos + cap);
    const int tid = threadIdx.x, w = blockIdx.x, b = blockIdx.y;
    // (rider) the cloud's M K neighbour items are spread over its workgroups, one per thread; the item's K table entries are requested
    // HERE, ahead of everything else, so that their round trip is over by the time the means exist
    constexpr int KP_MAXK = 16;
    int kidx[KP_MAXK];
    const int kp_stride = nW * SP_THREADS;
    if (kp.I != nullptr && kp.K <= KP_MAXK) {
        const int t = w * SP_THREADS + tid;
        if (t < M * kp.K) {
            const int64_t *Ib = kp.I + ((size_t)b * M + t / kp.K) * kp.KI;
#pragma unroll
            for (int q = 0; q < KP_MAXK; ++q) {
                long long v = -1;
                if (q < kp.K) v = Ib[q];
                kidx[q] = ((unsigned long long)v < (unsigned long long)M) ? (int)v : -1;
            }
        }
    }
    // totals over the cloud's workgroups, one thread per (quantity, node): quantity 0 = count (+ this workgroup's run start), 1-3 = sums
    for (int t = tid; t < 4 * M; t += SP_THREADS) {
        const int q = t / M, m = t - q * M;
        if (q == 0) {
            int tt = 0, bs = 0, mi = 0;
            for (int ww = 0; ww < nW; ++ww) {                                // fixed order: the same totals in every workgroup of the cloud
                const int c = cnt_part[((size_t)b * nW + ww) * M + m];
                if (ww < w) bs += c;
                if (ww == w) mi = c;
                tt += c;
            }
            tot[m] = tt; base[m] = bs; mine[m] = mi;
        } else {
            double acc = 0.0;
            for (int ww = 0; ww < nW; ++ww) acc += sum_part[(((size_t)b * nW + ww) * 3 + (q - 1)) * M + m];
            tsum[(q - 1) * M + m] = acc;
        }
    }
    __syncthreads();
    for (int t = tid; t < 3 * M; t += SP_THREADS) {
        const int c = t / M, m = t - c * M;
        const float denom = __fadd_rn((float)tot[m], 1e-5f);                 // networks.py:142
        const float mv = __fdiv_rn((float)tsum[t], denom);
        mean[t] = mv;
        if (w == 0) {
            if (som_node != nullptr) som_node[(size_t)b * 3 * M + t] = mv;
            if (sum_ws != nullptr) sum_ws[(size_t)b * 3 * M + t] = tsum[t];
            if (c == 0) {
                if (row_max != nullptr) row_max[(size_t)b * M + m] = tot[m] > 0;
                if (count != nullptr) count[(size_t)b * M + m] = tot[m];
            }
        }
    }
    lds_exclusive_scan(tot, noff, M, tid);                                   // node offsets of the sorted cloud (first wave)
    if (tid >= 64 && tid < 128) lds_exclusive_scan(mine, lstart, M, tid - 64);   // this workgroup's local node order (second wave)
    __syncthreads();
    if (w == 0) for (int m = tid; m < M; m += SP_THREADS) node_off[(size_t)b * M + m] = noff[m];
    if (kp.I != nullptr) {
        // KNNModule's neighbourhood centres and de-centred neighbour coordinates (models/layers.py:319-350) from the means in LDS
        const int K = kp.K, MK = M * K;
        bool first = K <= KP_MAXK;
        for (int t = w * SP_THREADS + tid; t < MK; t += kp_stride, first = false) {
            const int m = t / K, kk = t - m * K;
            const int64_t *Ib = kp.I + ((size_t)b * M + m) * kp.KI;
            int id = -1;
            if (first) {
#pragma unroll
                for (int q = 0; q < KP_MAXK; ++q) id = q == kk ? kidx[q] : id;
            } else {
                const long long v = Ib[kk];
                id = ((unsigned long long)v < (unsigned long long)M) ? (int)v : -1;
            }
            const bool ok = id >= 0;
            float d[3], ctr[3];
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                const float *src = mean + c * M;
                if (kp.avg) {
                    float sum = 0.f;
                    if (first) {
#pragma unroll
                        for (int q = 0; q < KP_MAXK; ++q)
                            if (q < K) sum += kidx[q] >= 0 ? src[kidx[q]] : 0.f;
                    } else {
                        for (int q = 0; q < K; ++q) {
                            const long long ik = Ib[q];
                            sum += ((unsigned long long)ik < (unsigned long long)M) ? src[ik] : 0.f;
                        }
                    }
                    ctr[c] = sum / (float)K;
                } else {
                    ctr[c] = src[m];
                }
                d[c] = (ok ? src[id] : 0.f) - ctr[c];
            }
            const long long n = (long long)b * M + m, blk = n / kp.G;
            const int g = (int)(n - blk * kp.G);
            kp.rec[blk * 128 + g * K + kk] = make_int4(ok ? (int)((long long)b * M + id) : -1, __float_as_int(d[0]), __float_as_int(d[1]), __float_as_int(d[2]));
            if (kk == 0) {
#pragma unroll
                for (int c = 0; c < 3; ++c) kp.center[((size_t)b * 3 + c) * M + m] = ctr[c];
                unsigned h0, m0, h1v, m1;
                p16_split_pair(ctr[0], ctr[1], h0, m0);
                p16_split_pair(ctr[2], 0.f, h1v, m1);
                kp.center_p16[0 * kp.Lm + n] = make_uint4(h0, h1v, 0u, 0u);
                kp.center_p16[1 * kp.Lm + n] = make_uint4(0u, 0u, 0u, 0u);
                kp.center_p16[2 * kp.Lm + n] = make_uint4(m0, m1, 0u, 0u);
                kp.center_p16[3 * kp.Lm + n] = make_uint4(0u, 0u, 0u, 0u);
                // the padding columns of a block: behind its last node, or behind the very last node of the batch
                if (g == kp.G - 1 || n == kp.BM - 1)
                    for (int col = (g + 1) * K; col < 128; ++col) kp.rec[blk * 128 + col] = make_int4(-2, 0, 0, 0);
            }
        }
    }
    const size_t kN = (size_t)k * N;
    const float *xb = x + (size_t)b * 3 * N;
    const float *snb = sn + (size_t)b * 3 * N;
    int nloc = N - w * SP_T;
    nloc = (nloc > SP_T ? SP_T : nloc) * k;                                  // copies of this workgroup
#pragma unroll
    for (int p = 0; p < SP_PPT; ++p) {
        const int n = w * SP_T + p * SP_THREADS + tid;
        if (n < N) {
            float v[6];
#pragma unroll
            for (int c = 0; c < 3; ++c) { v[c] = xb[(size_t)c * N + n]; v[3 + c] = snb[(size_t)c * N + n]; }
            for (int s_ = 0; s_ < k; ++s_) {
                const size_t o = (size_t)b * kN + (size_t)s_ * N + n;
                const int m = min32[o], r = (int)rank16[o];
                const int ls = lstart[m] + r;
                const int g = noff[m] + base[m] + r;
                gpos[ls] = g;
                if (n == 0 && s_ == 0) pos0[b] = g;
#pragma unroll
                for (int c = 0; c < 3; ++c) { buf[c * cap + ls] = __fsub_rn(v[c], mean[c * M + m]); buf[(3 + c) * cap + ls] = v[3 + c]; }
                buf[6 * cap + ls] = __int_as_float(m);
            }
        }
    }
    __syncthreads();
    float *ob = x_aug_sorted + (size_t)b * 6 * kN;
    for (int i = tid; i < nloc; i += SP_THREADS) {                           // consecutive lanes: consecutive positions of a node's run
        const int g = gpos[i];
#pragma unroll
        for (int c = 0; c < 6; ++c) ob[(size_t)c * kN + g] = buf[c * cap + i];
        ids_sorted[(size_t)b * kN + g] = __float_as_int(buf[6 * cap + i]);
    }
}

}  // namespace

extern "C" size_t sonet_som_assign_sort_ws_size(int B, int N, int M, int k)
{
    if (B <= 0 || N <= 0 || M <= 0 || k <= 0) return 0;
    const size_t nW = (size_t)sonet::ceil_div(N, SP_T);
    // sum_part f64 [B][nW][3][M] | cnt_part i32 [B][nW][M] | rank16 u16 [B][kN] (8-byte aligned blocks)
    return (size_t)B * nW * 3 * M * 8 + (((size_t)B * nW * M * 4 + 7) & ~(size_t)7) + (((size_t)B * k * N * 2 + 7) & ~(size_t)7);
}

static int som_assign_sort_impl(const char *what, const float *x, const float *sn, const float *node, int B, int N, int M, int k,
                                int32_t *min_idx_i32, int64_t *min_idx_i64, int32_t *count, double *sum_ws,
                                float *som_node, int32_t *row_max, float *x_aug_sorted, int32_t *ids_sorted,
                                int32_t *pos0, int32_t *node_off, void *ws, const KnnPrep &kp, sonet_stream_t stream, int det = 0)
{
    SONET_REQUIRE(x && sn && node && min_idx_i32 && count && x_aug_sorted && ids_sorted && pos0 && node_off && ws, "%s: NULL pointer", what);
    SONET_REQUIRE(B > 0 && N > 0 && M > 0, "%s: non-positive size B=%d N=%d M=%d", what, B, N, M);
    SONET_REQUIRE(k >= 1 && k <= SP_KMAX && k <= M, "%s: k=%d must be in [1, min(4, M=%d)]", what, k, M);
    if (M > 1024) return sonet::fail(SONET_ERR_UNSUPPORTED, "%s: M=%d > 1024 nodes", what, M);
    if (B > 65535) return sonet::fail(SONET_ERR_UNSUPPORTED, "%s: B=%d > 65535", what, B);
    hipStream_t st = sonet::as_stream(stream);
    const int nW = sonet::ceil_div(N, SP_T);
    double *sum_part = reinterpret_cast<double *>(ws);
    int32_t *cnt_part = reinterpret_cast<int32_t *>(sum_part + (size_t)B * nW * 3 * M);
    uint16_t *rank16 = reinterpret_cast<uint16_t *>(reinterpret_cast<char *>(cnt_part) + (((size_t)B * nW * M * 4 + 7) & ~(size_t)7));
    dim3 grid((unsigned)nW, (unsigned)B), block(SP_THREADS);
    const size_t lds1 = (size_t)M * (sizeof(float4) + 2 * sizeof(int)) + (size_t)3 * k * SP_T * sizeof(float) + (det ? (size_t)(SP_THREADS / 64) * M * sizeof(int) : 0);
    const size_t lds2 = (size_t)M * (3 * sizeof(double) + 3 * sizeof(float) + 5 * sizeof(int)) + (size_t)8 * k * SP_T * sizeof(float);
#define SP_LAUNCH(KK) do { \
        if (M <= 64) { if (lds1 > 48 * 1024 && hipFuncSetAttribute(reinterpret_cast<const void *>(som_assign_rank_kernel<KK, 6>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds1) != hipSuccess) return sonet::fail(SONET_ERR_LAUNCH, "%s: LDS", what); \
                       hipLaunchKernelGGL((som_assign_rank_kernel<KK, 6>), grid, block, lds1, st, x, node, N, M, nW, min_idx_i32, min_idx_i64, rank16, cnt_part, sum_part, det); } \
        else         { if (lds1 > 48 * 1024 && hipFuncSetAttribute(reinterpret_cast<const void *>(som_assign_rank_kernel<KK, 10>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds1) != hipSuccess) return sonet::fail(SONET_ERR_LAUNCH, "%s: LDS", what); \
                       hipLaunchKernelGGL((som_assign_rank_kernel<KK, 10>), grid, block, lds1, st, x, node, N, M, nW, min_idx_i32, min_idx_i64, rank16, cnt_part, sum_part, det); } } while (0)
    switch (k) { case 1: SP_LAUNCH(1); break; case 2: SP_LAUNCH(2); break; case 3: SP_LAUNCH(3); break; default: SP_LAUNCH(4); }
#undef SP_LAUNCH
    if (lds2 > 48 * 1024 && hipFuncSetAttribute(reinterpret_cast<const void *>(som_sort_fill2_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds2) != hipSuccess)
        return sonet::fail(SONET_ERR_LAUNCH, "%s: LDS", what);
    hipLaunchKernelGGL(som_sort_fill2_kernel, grid, block, lds2, st, x, sn, min_idx_i32, rank16, cnt_part, sum_part, N, M, k, nW,
                       count, sum_ws, som_node, row_max, x_aug_sorted, ids_sorted, pos0, node_off, kp);
    return sonet::launched(what);
}

extern "C" int sonet_som_assign_sort_f32(const float *x, const float *sn, const float *node, int B, int N, int M, int k,
                                         int32_t *min_idx_i32, int64_t *min_idx_i64, int32_t *count, double *sum_ws,
                                         float *som_node, int32_t *row_max, float *x_aug_sorted, int32_t *ids_sorted,
                                         int32_t *pos0, int32_t *node_off, void *ws, sonet_stream_t stream)
{
    KnnPrep kp;
    kp.I = nullptr; kp.KI = kp.K = kp.avg = kp.G = 0; kp.BM = kp.Lm = 0; kp.center = nullptr; kp.center_p16 = nullptr; kp.rec = nullptr;
    return som_assign_sort_impl("sonet_som_assign_sort_f32", x, sn, node, B, N, M, k, min_idx_i32, min_idx_i64, count, sum_ws, som_node, row_max,
                                x_aug_sorted, ids_sorted, pos0, node_off, ws, kp, stream);
}

/* sonet_som_assign_sort_f32 with a sort whose order INSIDE a node does not depend on the arrival order of atomics (wave, slot, lane order
 * inside a 512-point workgroup, workgroups in order): the same sorted copy in every run -- what the f32-class training forward wants, whose
 * BatchNorm batch sums run over the sorted columns.  Same ids, counts, means, node offsets. */
extern "C" int sonet_som_assign_sort_det_f32(const float *x, const float *sn, const float *node, int B, int N, int M, int k,
                                             int32_t *min_idx_i32, int64_t *min_idx_i64, int32_t *count, double *sum_ws,
                                             float *som_node, int32_t *row_max, float *x_aug_sorted, int32_t *ids_sorted,
                                             int32_t *pos0, int32_t *node_off, void *ws, sonet_stream_t stream)
{
    KnnPrep kp;
    kp.I = nullptr; kp.KI = kp.K = kp.avg = kp.G = 0; kp.BM = kp.Lm = 0; kp.center = nullptr; kp.center_p16 = nullptr; kp.rec = nullptr;
    return som_assign_sort_impl("sonet_som_assign_sort_det_f32", x, sn, node, B, N, M, k, min_idx_i32, min_idx_i64, count, sum_ws, som_node, row_max,
                                x_aug_sorted, ids_sorted, pos0, node_off, ws, kp, stream, 1);
}

/* sonet_som_assign_sort_f32 whose second launch also does sonet_knn_stage_prepare_f32 on the cluster means it computes (KNNModule's index /
 * coordinate side on the flat column axis of the node-level stage, include/sonet_hip.h): same outputs center [B][3][M], center_p16, rec --
 * bit-identical to the separate launch on som_node. */
extern "C" int sonet_som_assign_sort_knn_f32(const float *x, const float *sn, const float *node, int B, int N, int M, int k,
                                             int32_t *min_idx_i32, int64_t *min_idx_i64, int32_t *count, double *sum_ws,
                                             float *som_node, int32_t *row_max, float *x_aug_sorted, int32_t *ids_sorted,
                                             int32_t *pos0, int32_t *node_off, void *ws,
                                             const int64_t *knn_I, int KI, int K, int center_avg, float *center, void *center_p16, void *rec,
                                             sonet_stream_t stream)
{
    const char *what = "sonet_som_assign_sort_knn_f32";
    SONET_REQUIRE(knn_I && center && center_p16 && rec, "%s: NULL pointer", what);
    SONET_REQUIRE(K >= 1 && K <= 128 && KI >= K, "%s: bad neighbour count K=%d (of %d)", what, K, KI);
    KnnPrep kp;
    kp.I = knn_I; kp.KI = KI; kp.K = K; kp.avg = center_avg; kp.G = knn_stage_groups(K);
    kp.BM = (long long)B * M; kp.Lm = (kp.BM + 127) / 128 * 128;
    kp.center = center; kp.center_p16 = reinterpret_cast<uint4 *>(center_p16); kp.rec = reinterpret_cast<int4 *>(rec);
    return som_assign_sort_impl(what, x, sn, node, B, N, M, k, min_idx_i32, min_idx_i64, count, sum_ws, som_node, row_max,
                                x_aug_sorted, ids_sorted, pos0, node_off, ws, kp, stream);
}

extern "C" int sonet_som_sort_group_f32(const float *x, const float *sn, const int32_t *min_idx_i32, const int32_t *count,
                                        const double *sum_ws, int B, int N, int M, int k, float *som_node, int32_t *row_max,
                                        float *x_aug_sorted, int32_t *ids_sorted, int32_t *pos0, int32_t *node_off,
                                        int32_t *cursor_ws, sonet_stream_t stream)
{
    const char *what = "sonet_som_sort_group_f32";
    SONET_REQUIRE(x && sn && min_idx_i32 && count && sum_ws && x_aug_sorted && ids_sorted && pos0 && node_off && cursor_ws, "%s: NULL pointer", what);
    SONET_REQUIRE(B > 0 && N > 0 && M > 0 && k >= 1, "%s: non-positive size", what);
    if (M > 4096) return sonet::fail(SONET_ERR_UNSUPPORTED, "%s: M=%d > 4096 nodes", what, M);
    if (B > 65535) return sonet::fail(SONET_ERR_UNSUPPORTED, "%s: B=%d > 65535", what, B);
    hipStream_t st = sonet::as_stream(stream);
    const long long kN = (long long)k * N;
    // (One 1024-thread workgroup per cloud doing the whole sort in LDS -- no global cursor, no memset, one launch -- measured
    //  SLOWER: 50 vs 39 us at B = 64; 64 workgroups and 15000 LDS atomics on 64 counters each.  Not kept.)
    if (sonet::zero_words(cursor_ws, (size_t)B * M * sizeof(int32_t), st) != 0)       // (a kernel, not a memset node: common.hpp)
        return sonet::fail(SONET_ERR_LAUNCH, "%s: zero fill failed", what);
    dim3 grid((unsigned)sonet::ceil_div64(kN, SG_THREADS * SG_PER_THREAD), B), block(SG_THREADS);
    hipLaunchKernelGGL(som_sort_group_kernel, grid, block, (size_t)M * (3 * sizeof(float) + 3 * sizeof(int)), st,
                       x, sn, min_idx_i32, count, sum_ws, N, M, k, som_node, row_max, x_aug_sorted, ids_sorted, pos0, cursor_ws, node_off);
    hipLaunchKernelGGL(som_sort_fill_kernel, dim3((unsigned)sonet::ceil_div64(kN, 256), B), dim3(256), (size_t)M * 3 * sizeof(float), st,
                       x, sn, count, sum_ws, ids_sorted, N, M, k, x_aug_sorted);
    return sonet::launched(what);
}

extern "C" int sonet_node_gather_f32(const float *feat, const int32_t *min_idx_i32, float *out, int B, int C, int M, int kN,
                                     sonet_stream_t stream)
{
    const char *what = "sonet_node_gather_f32";
    SONET_REQUIRE(feat && min_idx_i32 && out, "%s: NULL pointer", what);
    SONET_REQUIRE(B > 0 && C > 0 && M > 0 && kN > 0, "%s: non-positive size", what);
    const long long total4 = (long long)B * C * ((kN + 3) / 4);
    const long long blocks = sonet::ceil_div64(total4, 256);
    if (blocks > 0x7FFFFFFFll) return sonet::fail(SONET_ERR_UNSUPPORTED, "%s: too large", what);
    hipLaunchKernelGGL(node_gather_kernel, dim3((unsigned)blocks), dim3(256), 0, sonet::as_stream(stream),
                       feat, min_idx_i32, out, C, M, kN, total4);
    return sonet::launched(what);
}

extern "C" int sonet_som_assign_f32(const float *x, const float *node, int B, int N, int M, int k,
                                    int32_t *min_idx_i32, int64_t *min_idx_i64, int32_t *count, double *sum_ws,
                                    sonet_stream_t stream)
{
    const char *what = "sonet_som_assign_f32";
    SONET_REQUIRE(x && node && min_idx_i32 && count && sum_ws, "%s: NULL pointer", what);
    SONET_REQUIRE(B > 0 && N > 0 && M > 0, "%s: non-positive size B=%d N=%d M=%d", what, B, N, M);
    SONET_REQUIRE(k >= 1 && k <= 4 && k <= M, "%s: k=%d must be in [1, min(4, M=%d)]", what, k, M);
    if (M > 1024) return sonet::fail(SONET_ERR_UNSUPPORTED, "%s: M=%d > 1024 nodes", what, M);
    if (B > 65535) return sonet::fail(SONET_ERR_UNSUPPORTED, "%s: B=%d > 65535", what, B);
    hipStream_t st = sonet::as_stream(stream);
    const size_t sum_bytes = (size_t)B * 3 * M * sizeof(double), cnt_bytes = (size_t)B * M * sizeof(int32_t);
    const bool adjacent = reinterpret_cast<char *>(sum_ws) + sum_bytes == reinterpret_cast<char *>(count);   // one clear launch then
    if (adjacent ? sonet::zero_words(sum_ws, sum_bytes + cnt_bytes, st) != 0
                 : (sonet::zero_words(count, cnt_bytes, st) != 0 || sonet::zero_words(sum_ws, sum_bytes, st) != 0))
        return sonet::fail(SONET_ERR_LAUNCH, "%s: zero fill failed", what);
    dim3 grid(sonet::ceil_div(N, SA_THREADS), B), block(SA_THREADS);
    const size_t lds = (size_t)M * (sizeof(float4) + 3 * sizeof(double) + sizeof(unsigned));
    const char *ek = sonet::knob("SONET_SOM_KEYS");                  // bench / test switch: 0 = the insertion-list kernel
    const bool keys = !(ek && atoi(ek) == 0);
#define SA_LAUNCH(KK) do { if (keys && M <= 64) hipLaunchKernelGGL((som_assign_keys_kernel<KK, 6>), grid, block, lds, st, x, node, N, M, min_idx_i32, min_idx_i64, count, sum_ws); \
                           else if (keys) hipLaunchKernelGGL((som_assign_keys_kernel<KK, 10>), grid, block, lds, st, x, node, N, M, min_idx_i32, min_idx_i64, count, sum_ws); \
                           else hipLaunchKernelGGL((som_assign_kernel<KK>), grid, block, lds, st, x, node, N, M, min_idx_i32, min_idx_i64, count, sum_ws); } while (0)
    switch (k) { case 1: SA_LAUNCH(1); break; case 2: SA_LAUNCH(2); break; case 3: SA_LAUNCH(3); break; default: SA_LAUNCH(4); }
#undef SA_LAUNCH
    return sonet::launched(what);
}

extern "C" int sonet_som_group_f32(const float *x, const float *sn, const int32_t *min_idx_i32, const int32_t *count,
                                   const double *sum_ws, int B, int N, int M, int k, float *som_node, int32_t *row_max,
                                   float *centers, float *x_decentered, float *x_augmented, sonet_stream_t stream)
{
    const char *what = "sonet_som_group_f32";
    SONET_REQUIRE(x && min_idx_i32 && count && sum_ws, "%s: NULL pointer", what);
    SONET_REQUIRE(B > 0 && N > 0 && M > 0 && k >= 1, "%s: non-positive size", what);
    SONET_REQUIRE(!(x_augmented != nullptr && sn == nullptr), "%s: x_augmented needs sn", what);
    if (M > 4096) return sonet::fail(SONET_ERR_UNSUPPORTED, "%s: M=%d > 4096 nodes", what, M);
    if (B > 65535) return sonet::fail(SONET_ERR_UNSUPPORTED, "%s: B=%d > 65535", what, B);
    const bool per_point = centers || x_decentered || x_augmented;
    const long long kN = (long long)k * N;
    dim3 grid(per_point ? (unsigned)sonet::ceil_div64(kN, SG_THREADS * SG_PER_THREAD) : 1u, B), block(SG_THREADS);
    hipLaunchKernelGGL(som_group_kernel, grid, block, (size_t)3 * M * sizeof(float), sonet::as_stream(stream),
                       x, sn, min_idx_i32, count, sum_ws, N, M, k, som_node, row_max, centers, x_decentered, x_augmented);
    return sonet::launched(what);
}

extern "C" int sonet_som_mask_i32(const int32_t *min_idx_i32, int32_t *mask, int B, int kN, int M, sonet_stream_t stream)
{
    const char *what = "sonet_som_mask_i32";
    SONET_REQUIRE(min_idx_i32 && mask, "%s: NULL pointer", what);
    SONET_REQUIRE(B > 0 && kN > 0 && M > 0, "%s: non-positive size", what);
    const long long rows = (long long)B * kN;
    const long long threads = (M & 3) == 0 ? rows * (M >> 2) : rows * M;
    const long long blocks = sonet::ceil_div64(threads, 256);
    if (blocks > 0x7FFFFFFFll) return sonet::fail(SONET_ERR_UNSUPPORTED, "%s: too large", what);
    hipLaunchKernelGGL(som_mask_kernel, dim3((unsigned)blocks), dim3(256), 0, sonet::as_stream(stream),
                       min_idx_i32, mask, rows, M);
    return sonet::launched(what);
}

extern "C" int sonet_knn_gather_f32(const float *x, const int64_t *knn_I, float *out, int B, int C, int M, int K,
                                    sonet_stream_t stream)
{
    const char *what = "sonet_knn_gather_f32";
    SONET_REQUIRE(x && knn_I && out, "%s: NULL pointer", what);
    SONET_REQUIRE(B > 0 && C > 0 && M > 0 && K > 0, "%s: non-positive size", what);
    const long long total = (long long)B * C * M * K;
    const long long blocks = sonet::ceil_div64(total, 256);
    if (blocks > 0x7FFFFFFFll) return sonet::fail(SONET_ERR_UNSUPPORTED, "%s: too large", what);
    hipLaunchKernelGGL(knn_gather_kernel, dim3((unsigned)blocks), dim3(256), 0, sonet::as_stream(stream),
                       x, knn_I, out, C, M, K, total);
    return sonet::launched(what);
}

extern "C" int sonet_knn_group_f32(const float *coord, const float *feat, const int64_t *knn_I, int B, int C, int M, int K,
                                   int center_avg, float *center, float *out, sonet_stream_t stream)
{
    const char *what = "sonet_knn_group_f32";
    SONET_REQUIRE(coord && feat && knn_I && center && out, "%s: NULL pointer", what);
    SONET_REQUIRE(B > 0 && C > 0 && M > 0 && K > 0, "%s: non-positive size", what);
    if (B > 65535 || 3 + C > 65535 || (long long)M * K > 0x7FFFFFFFll) return sonet::fail(SONET_ERR_UNSUPPORTED, "%s: too large", what);
    hipLaunchKernelGGL(knn_group_kernel, dim3((unsigned)sonet::ceil_div(M * K, 1024), (unsigned)(3 + C), (unsigned)B), dim3(256), 0,
                       sonet::as_stream(stream), coord, feat, knn_I, C, M, K, center_avg, center, out);
    return sonet::launched(what);
}

extern "C" int sonet_planes_max_f32(const float *x, float *out, long long rows, int K, int M, sonet_stream_t stream)
{
    const char *what = "sonet_planes_max_f32";
    SONET_REQUIRE(x && out, "%s: NULL pointer", what);
    SONET_REQUIRE(rows > 0 && K > 0 && M > 0, "%s: non-positive size", what);
    const long long total = rows * M;
    if (sonet::ceil_div64(total, 256) > 0x7FFFFFFFll) return sonet::fail(SONET_ERR_UNSUPPORTED, "%s: too large", what);
    hipLaunchKernelGGL(planes_max_kernel, dim3((unsigned)sonet::ceil_div64(total, 256)), dim3(256), 0, sonet::as_stream(stream), x, out, K, M, total);
    return sonet::launched(what);
}

extern "C" int sonet_knn_prepare_f32(const float *coord, const int64_t *knn_I, int B, int M, int K, int center_avg,
                                     float *center, float *dec, int32_t *gidx, sonet_stream_t stream)
{
    const char *what = "sonet_knn_prepare_f32";
    SONET_REQUIRE(coord && knn_I && center && dec && gidx, "%s: NULL pointer", what);
    SONET_REQUIRE(B > 0 && M > 0 && K > 0, "%s: non-positive size", what);
    const long long total = (long long)B * K * M;
    if (sonet::ceil_div64(total, 256) > 0x7FFFFFFFll) return sonet::fail(SONET_ERR_UNSUPPORTED, "%s: too large", what);
    hipLaunchKernelGGL(knn_prepare_kernel, dim3((unsigned)sonet::ceil_div64(total, 256)), dim3(256), 0, sonet::as_stream(stream),
                       coord, knn_I, M, K, center_avg, center, dec, gidx, total);
    return sonet::launched(what);
}

extern "C" int sonet_lastdim_max_f32(const float *x, float *out, long long rows, int K, sonet_stream_t stream)
{
    const char *what = "sonet_lastdim_max_f32";
    SONET_REQUIRE(x && out, "%s: NULL pointer", what);
    SONET_REQUIRE(rows > 0 && K > 0, "%s: non-positive size", what);
    const int tpr = K >= 48 ? 16 : K >= 6 ? 4 : 1;            // K = 64: 16 lanes x float loads (coalesced 256 B); K = 9: 4 lanes
    const long long blocks = sonet::ceil_div64(rows * tpr, 256);
    if (blocks > 0x7FFFFFFFll) return sonet::fail(SONET_ERR_UNSUPPORTED, "%s: too large", what);
    hipStream_t st = sonet::as_stream(stream);
    if (tpr == 16) hipLaunchKernelGGL(lastdim_max_kernel<16>, dim3((unsigned)blocks), dim3(256), 0, st, x, out, K, rows);
    else if (tpr == 4) hipLaunchKernelGGL(lastdim_max_kernel<4>, dim3((unsigned)blocks), dim3(256), 0, st, x, out, K, rows);
    else hipLaunchKernelGGL(lastdim_max_kernel<1>, dim3((unsigned)blocks), dim3(256), 0, st, x, out, K, rows);
    return sonet::launched(what);
}

extern "C" int sonet_knn_self_f32(const float *node, int64_t *knn_I, int B, int M, int K, sonet_stream_t stream)
{
    const char *what = "sonet_knn_self_f32";
    SONET_REQUIRE(node && knn_I, "%s: NULL pointer", what);
    SONET_REQUIRE(B > 0 && M > 0 && K > 0 && K <= M, "%s: bad size B=%d M=%d K=%d", what, B, M, K);
    if (K > KS_MAX) return sonet::fail(SONET_ERR_UNSUPPORTED, "%s: K=%d > %d", what, K, KS_MAX);
    const long long total = (long long)B * M;
    hipLaunchKernelGGL(knn_self_kernel, dim3((unsigned)sonet::ceil_div64(total, 256)), dim3(256), 0, sonet::as_stream(stream),
                       node, knn_I, M, K, total);
    return sonet::launched(what);
}
