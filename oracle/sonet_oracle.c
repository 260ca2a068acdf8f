/*
 * sonet_oracle.c -- CPU restatement of the SO-Net "SOM assignment -> grouped point feature" hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under so-net_amd/ (the product) may call, link or import this
 * file or anything else in oracle/.  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline
 * leg use it, and only as the checker / the timed CPU baseline -- never as the thing shipped.
 *
 * Parity pinning: the reference has NO tests or golden vectors for this path, so
 * the restatement is pinned differentially: oracle/make_golden.py runs the unmodified reference
 * (/root/reference, imported with empty shims for its unused third-party imports, its own
 * index_max.cpp compiled by oracle/build_ref.py) on seeded inputs and commits the outputs under
 * tests/golden/; tests/test_oracle_golden.py checks every function below against those files.
 *
 * Every function cites the reference file:line (paths under /root/reference) it follows.
 * Plain C99, scalar, single thread:  gcc -O2 -ffp-contract=off -shared -fPIC.
 * (-ffp-contract=off matters: the reference's distance is separate multiplies and adds, no FMA.)
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <math.h>

/* ------------------------------------------------------------------------------------------------
 * index_max  --  models/index_max_ext/index_max.cpp:73-112 (index_max_forward_cpu), same math as
 * the CUDA kernel models/index_max_ext/index_max_cuda.cu:10-26.
 *   data  B x C x N  f32, index B x N i32 (node id of column n), out B x C x K i32.
 *   max_val starts at -1000 (index_max.cpp:82), max_idx at 0 (:81); ascending n; strict '>' (:104).
 * Consequences that the HIP kernel must reproduce: ties -> smallest n; values <= -1000, NaN and
 * empty segments -> 0;  -0.0 does not beat +0.0 and vice versa.
 * ---------------------------------------------------------------------------------------------- */
void oracle_index_max_f32(const float *data, const int32_t *index, int32_t *out,
                          int B, int C, int N, int K)
{
    float *max_val = (float *)malloc(sizeof(float) * (size_t)K);
    for (int b = 0; b < B; ++b) {
        for (int c = 0; c < C; ++c) {
            int32_t *o = out + ((size_t)b * C + c) * K;
            const float *row = data + ((size_t)b * C + c) * N;
            const int32_t *idx = index + (size_t)b * N;
            for (int k = 0; k < K; ++k) { max_val[k] = -1000.0f; o[k] = 0; }
            for (int n = 0; n < N; ++n) {
                int k = idx[n];
                float v = row[n];
                if (v > max_val[k]) { max_val[k] = v; o[k] = n; }
            }
        }
    }
    free(max_val);
}

/* ------------------------------------------------------------------------------------------------
 * BatchSOM.query_topk  --  util/som.py:237-269.
 *   x B x 3 x N f32, node B x 3 x M f32.
 *   diff = x - node (som.py:249); diff_norm = (diff**2).sum(dim=1) (:250): evaluated by aten as
 *   ((dx*dx + dy*dy) + dz*dz) in f32 -- separate multiply / add, ascending channel order.
 *   topk(k, largest=False, sorted=False) (:253): the SET of the k smallest is defined, the slot
 *   order is implementation-defined (aten CPU uses nth_element).  This restatement -- and the HIP
 *   kernel -- emit the canonical order: ascending (distance, node id).  Tests compare per-point
 *   sets against the reference and exact slots against the reference run with sorted=True.
 *   min_idx  B x kN i64, k-major concat (:261-266): min_idx[b, s*N + n] = slot s of point n.
 *   count    B x M i32  = mask.sum(1)      (models/networks.py:128)
 *   row_max  B x M i32  = max_n mask        (util/som.py:267)   == (count > 0)
 * ---------------------------------------------------------------------------------------------- */
void oracle_som_query_topk_f32(const float *x, const float *node, int B, int N, int M, int k,
                               int64_t *min_idx, int32_t *count, int32_t *row_max)
{
    float *bd = (float *)malloc(sizeof(float) * (size_t)k);
    int *bi = (int *)malloc(sizeof(int) * (size_t)k);
    for (int b = 0; b < B; ++b) {
        const float *xb = x + (size_t)b * 3 * N;
        const float *nb = node + (size_t)b * 3 * M;
        int32_t *cnt = count + (size_t)b * M;
        for (int m = 0; m < M; ++m) cnt[m] = 0;
        for (int n = 0; n < N; ++n) {
            float px = xb[n], py = xb[N + n], pz = xb[2 * N + n];
            int filled = 0;
            for (int m = 0; m < M; ++m) {
                float dx = px - nb[m], dy = py - nb[M + m], dz = pz - nb[2 * M + m];
                float d = (dx * dx + dy * dy) + dz * dz;
                /* insertion into the ascending (d, m) list; m ascends so ties keep the lower id */
                int pos = filled;
                while (pos > 0 && d < bd[pos - 1]) --pos;
                if (pos < k) {
                    int last = (filled < k) ? filled : k - 1;
                    for (int j = last; j > pos; --j) { bd[j] = bd[j - 1]; bi[j] = bi[j - 1]; }
                    bd[pos] = d; bi[pos] = m;
                    if (filled < k) ++filled;
                }
            }
            for (int s = 0; s < k; ++s) {
                min_idx[(size_t)b * k * N + (size_t)s * N + n] = bi[s];
                cnt[bi[s]] += 1;
            }
        }
        for (int m = 0; m < M; ++m) row_max[(size_t)b * M + m] = cnt[m] > 0;
    }
    free(bd); free(bi);
}

/* ------------------------------------------------------------------------------------------------
 * Encoder grouping block  --  models/networks.py:128-171.
 *   cluster_mean[b,c,m] = sum_{j: min_idx[b,j]==m} x_stack[b,c,j] / (count[b,m] + 1e-5)   (:140-142)
 *   centers[b,c,j]      = cluster_mean[b,c,min_idx[b,j]]  (one-hot . node sum == gather)  (:168-169)
 *   x_decentered        = x_stack - centers                                                 (:171)
 *   x_stack = k copies of x along the point axis (:132-136), j = s*N + n.
 * The reference accumulates the sum in f32 with aten's cascade order; this restatement accumulates
 * in f64 and rounds once (the value the f32 sum approximates), the division is f32 as in the
 * reference.  Tolerance against the reference: 1e-5 * max(|ref|, rms(ref))  (hard part 4).
 * ---------------------------------------------------------------------------------------------- */
void oracle_som_group_f32(const float *x, const int64_t *min_idx, int B, int N, int M, int k,
                          float *som_node, float *centers, float *x_decentered)
{
    size_t kN = (size_t)k * N;
    double *acc = (double *)malloc(sizeof(double) * 3 * (size_t)M);
    int32_t *cnt = (int32_t *)malloc(sizeof(int32_t) * (size_t)M);
    for (int b = 0; b < B; ++b) {
        const float *xb = x + (size_t)b * 3 * N;
        const int64_t *ib = min_idx + (size_t)b * kN;
        float *nodeb = som_node + (size_t)b * 3 * M;
        memset(acc, 0, sizeof(double) * 3 * (size_t)M);
        memset(cnt, 0, sizeof(int32_t) * (size_t)M);
        for (size_t j = 0; j < kN; ++j) {
            int m = (int)ib[j];
            size_t n = j % (size_t)N;
            cnt[m] += 1;
            acc[m] += xb[n]; acc[M + m] += xb[N + n]; acc[2 * M + m] += xb[2 * (size_t)N + n];
        }
        for (int c = 0; c < 3; ++c)
            for (int m = 0; m < M; ++m)
                nodeb[c * M + m] = (float)acc[c * M + m] / ((float)cnt[m] + 1e-5f);
        if (centers || x_decentered) {
            for (int c = 0; c < 3; ++c)
                for (size_t j = 0; j < kN; ++j) {
                    float ctr = nodeb[c * M + (int)ib[j]];
                    size_t o = ((size_t)b * 3 + c) * kN + j;
                    if (centers) centers[o] = ctr;
                    if (x_decentered) x_decentered[o] = xb[(size_t)c * N + j % (size_t)N] - ctr;
                }
        }
    }
    free(acc); free(cnt);
}

/* ------------------------------------------------------------------------------------------------
 * knn_gather_by_indexing  --  models/operations.py:38-54.
 *   x B x C x M f32, I B x M x K i64  ->  out B x C x M x K,  out[b,c,m,j] = x[b,c,I[b,m,j]].
 * ---------------------------------------------------------------------------------------------- */
void oracle_knn_gather_f32(const float *x, const int64_t *I, float *out, int B, int C, int M, int K)
{
    for (int b = 0; b < B; ++b)
        for (int c = 0; c < C; ++c)
            for (int m = 0; m < M; ++m)
                for (int j = 0; j < K; ++j)
                    out[(((size_t)b * C + c) * M + m) * K + j] =
                        x[((size_t)b * C + c) * M + I[((size_t)b * M + m) * K + j]];
}

/* ------------------------------------------------------------------------------------------------
 * Point-wise layer  --  models/layers.py:282-296 (EquivariantLayer.forward), eval-mode BN
 * (layers.py:60-70 -> F.batch_norm with running stats).
 *   y[b,o,l] = act( (sum_i W[o,i] x[b,i,l] + bias[o] - mean[o]) / sqrt(var[o]+eps) * gamma[o] + beta[o] )
 * f64 accumulation, one rounding to f32 after the conv and f32 BN arithmetic as aten does
 * (aten: (y - mean) * invstd * gamma + beta with invstd = 1/sqrt(var+eps) in f32).
 * bn == 0 skips normalisation, relu == 0 skips the activation (last layer of PointNet/PointResNet,
 * layers.py:383, :416).
 * ---------------------------------------------------------------------------------------------- */
void oracle_pointwise_layer_f32(const float *x, const float *W, const float *bias,
                                const float *gamma, const float *beta, const float *mean,
                                const float *var, float eps, int bn, int relu,
                                float *y, int B, int Cin, int Cout, int L)
{
    double *acc = (double *)malloc(sizeof(double) * (size_t)L);
    for (int b = 0; b < B; ++b)
        for (int o = 0; o < Cout; ++o) {
            for (int l = 0; l < L; ++l) acc[l] = 0.0;
            for (int i = 0; i < Cin; ++i) {
                double w = W[(size_t)o * Cin + i];
                const float *xr = x + ((size_t)b * Cin + i) * L;
                for (int l = 0; l < L; ++l) acc[l] += w * xr[l];
            }
            float *yr = y + ((size_t)b * Cout + o) * L;
            float invstd = bn ? 1.0f / sqrtf(var[o] + eps) : 1.0f;
            for (int l = 0; l < L; ++l) {
                float v = (float)(acc[l] + (double)bias[o]);
                if (bn) v = (v - mean[o]) * invstd * gamma[o] + beta[o];
                if (relu && !(v > 0.0f)) v = (v != v) ? v : 0.0f;
                yr[l] = v;
            }
        }
    free(acc);
}

/* ------------------------------------------------------------------------------------------------
 * Chamfer nearest neighbour ("next" row)  --  models/losses.py:220-235, 260-276.
 * The reference asks faiss (un-vendored, no pinned version; GpuIndexFlatL2) for the exact 1-NN of
 * every query in a database, per sample.  PARITY UNPINNED at the faiss boundary: faiss is absent
 * here, the reference has no test for it.  Published algorithm of IndexFlatL2: exact brute-force
 * squared-L2 argmin.  Restated with the same (dx*dx+dy*dy)+dz*dz arithmetic as above, ties -> lowest
 * database index.   q B x 3 x Nq, db B x 3 x Nd  ->  nn B x Nq i32.
 * ---------------------------------------------------------------------------------------------- */
void oracle_chamfer_nn_f32(const float *q, const float *db, int32_t *nn, int B, int Nq, int Nd)
{
    for (int b = 0; b < B; ++b) {
        const float *qb = q + (size_t)b * 3 * Nq, *dbb = db + (size_t)b * 3 * Nd;
        for (int i = 0; i < Nq; ++i) {
            float best = INFINITY; int bi = 0;
            for (int j = 0; j < Nd; ++j) {
                float dx = qb[i] - dbb[j], dy = qb[Nq + i] - dbb[Nd + j],
                      dz = qb[2 * Nq + i] - dbb[2 * Nd + j];
                float d = (dx * dx + dy * dy) + dz * dz;
                if (d < best) { best = d; bi = j; }
            }
            nn[(size_t)b * Nq + i] = bi;
        }
    }
}
