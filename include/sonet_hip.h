/*
 * sonet_hip.h -- C ABI of libsonet_hip.so: the MI355X (gfx950 / CDNA4) implementation of SO-Net's
 * "SOM assignment -> grouped point feature" hot path.
 *
 * This is the drop-in boundary.  Every entry point is `extern "C"`, takes plain DEVICE pointers,
 * sizes and a HIP stream, launches asynchronously on that stream (no device synchronisation, no
 * allocation) and returns a status code; sonet_last_error() gives the message for the calling
 * thread.  No torch types cross this boundary.  The reference interface each function replaces is
 * cited as file:line under the lijx10/SO-Net tree.  Host-side bindings (the python modules that
 * mirror the reference operator API, and the stub a reference maintainer would add) are described
 * in INTEGRATION.md.
 *
 * Conventions
 *   - all tensors are dense, row-major ("contiguous" in torch terms), batch outermost;
 *   - B = clouds in the batch, N = points per cloud, k = SOM nodes per point, kN = k*N,
 *     M = SOM nodes, C / Cin / Cout = feature channels, L = columns (points) per cloud;
 *   - `stream` is a hipStream_t passed as void* (NULL = the legacy default stream);
 *   - pointers marked "nullable" may be NULL to skip that output;
 *   - workspaces are caller-allocated; the callee initialises them on `stream`.
 */
#ifndef SONET_HIP_H
#define SONET_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef void *sonet_stream_t;

enum {
    SONET_OK = 0,
    SONET_ERR_INVALID_ARG = 1, /* NULL pointer, non-positive size, id out of range, misaligned   */
    SONET_ERR_UNSUPPORTED = 2, /* shape outside what the kernels instantiate (e.g. K > 1024)     */
    SONET_ERR_LAUNCH = 3,      /* hipGetLastError() after the launch was not hipSuccess          */
    SONET_ERR_NO_DEVICE = 4    /* no gfx950 device is current                                    */
};

/* Library identity: ABI version (bumped on any signature change) and build target ("gfx950"). */
int sonet_abi_version(void);
const char *sonet_build_arch(void);
/* Message of the last non-OK status returned to this thread ("" if none). */
const char *sonet_last_error(void);
/* SONET_OK iff the current HIP device reports gcnArchName gfx950*. */
int sonet_check_device(void);
/* Measuring stick (bench.py's roofline line; no reference counterpart): the rate a pure
 * v_mfma_f32_32x32x16_f16 loop -- one wave per SIMD on every CU, six independent accumulators, 48 * iters MFMAs per
 * wave -- sustains on the current device, in TFLOP/s, and the shader clock it ran at, in GHz.  random_operands != 0
 * fills the operands with random fp16 mantissas: on the whole chip that rate is power-limited well below the
 * nominal 2.4 GHz figure (finding 8).  Synchronous: returns after the timed launch has finished. */
int sonet_diag_mfma_f16_rate(int random_operands, int iters, double *tflops_out, double *ghz_out, sonet_stream_t stream);

/* Range log of the fp16-split ("h3") kernels (no reference counterpart: the reference computes in plain f32).
 * The three-term fp16 split has an fp16 operand RANGE: |x| <= 2047 (larger inputs are clamped) and relative precision
 * fades once the largest magnitude of a layer's input drops below ~2^-7; weights likewise (|w| <= 65504).  So that a
 * caller never gets clamped results unknowingly, every h3 launch (sonet_pointmlp_h3_f32, sonet_pointmlp_h3_gather_f32,
 * sonet_pointresnet_fused_f32, sonet_pointresnet_fused_pool_f32) reports the magnitudes it saw into the 8-word DEVICE
 * slot registered for the calling thread (NULL = no report, the default), by atomic max on the IEEE bit patterns
 * (a NaN sorts above +inf):
 *   slot[0] = bits of max |x| over the launch's input (for the fused kernels: the network input),
 *   slot[1] = bits of max |w| (recorded by the pack kernels in the packed weight's trailer),
 *   slot[2] = fused kernels only: bits of the largest post-BatchNorm-affine activation entering layers 2-4.
 * The caller zeroes the slot, launches, and reads it back at its next synchronisation point; a word above
 * bits(2047.0f) (slot[1]: bits(65504.0f)) or a non-zero slot[0] / slot[1] below bits(2^-6) means the launch's
 * results are not f32-class and must be recomputed with sonet_pointmlp_x3_f32 (f32 range).  sonet_hip/ops.py does
 * exactly that (ops.range_scope; Encoder.forward falls back to the x3 arithmetic for the batch). */
int sonet_range_log_set(uint32_t *slot);

/* ------------------------------------------------------------------------------------------------
 * index_max  -- replaces index_max.forward_cuda / forward_cuda_shared_mem
 *   reference: models/index_max_ext/index_max.cpp:132-148 (wrappers), index_max_cuda.cu:10-26,66-82
 *   caller:    models/networks.py:180-184
 * data  [B][C][Np] f32 (bf16 twin: raw bfloat16 bits), index [B][Np] i32 with 0 <= index < K,
 * out_idx [B][C][K] i32 (fully overwritten).
 * out_idx[b][c][m] = the position n of the maximum of { data[b][c][n] : index[b][n] == m } under the
 * reference's sequential semantics: running max starts at -1000, position at 0, ascending n,
 * strict '>'  =>  ties keep the smallest n; NaN, values <= -1000 and empty segments yield 0.
 * Bit-exact with index_max_forward_cpu on identical inputs.
 * sonet_index_max_gather_* additionally writes out_val[b][c][m] = data[b][c][ out_idx * row_max[b][m] ]
 * (the masked gather of models/networks.py:185; row_max nullable = all ones), saving one pass.
 * ---------------------------------------------------------------------------------------------- */
int sonet_index_max_f32(const float *data, const int32_t *index, int32_t *out_idx,
                        int B, int C, int Np, int K, sonet_stream_t stream);
int sonet_index_max_bf16(const uint16_t *data, const int32_t *index, int32_t *out_idx,
                         int B, int C, int Np, int K, sonet_stream_t stream);
int sonet_index_max_gather_f32(const float *data, const int32_t *index, const int32_t *row_max,
                               int32_t *out_idx, float *out_val,
                               int B, int C, int Np, int K, sonet_stream_t stream);
int sonet_index_max_gather_bf16(const uint16_t *data, const int32_t *index, const int32_t *row_max,
                                int32_t *out_idx, float *out_val, int B, int C, int Np, int K,
                                sonet_stream_t stream);
/* The same pool over an activation that exists only as P16 planes (sonet_p16_size(B, C, Np) bytes, layout under sonet_pointmlp_h3p):
 * values = (form 0 + form 1) / 32 exactly, i.e. the 22-bit values the next layer multiplies; otherwise the rules of
 * sonet_index_max_gather_f32 (models/index_max_ext/index_max_cuda.cu:10-26 + the gather of models/networks.py:185).  K <= 512. */
int sonet_index_max_gather_p16(const void *planes, const int32_t *index, const int32_t *row_max,
                               int32_t *out_idx, float *out_val, int B, int C, int Np, int K, sonet_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * som_assign  -- replaces the body of BatchSOM.query_topk (and BatchSOM.query for k = 1)
 *   reference: util/som.py:237-269 (:245-250 distance, :253 top-k, :261-267 outputs)
 *   caller:    models/networks.py:127-128
 * x [B][3][N] f32, node [B][3][M] f32, 1 <= k <= 4, k <= M <= 1024.
 * Distance is ((dx*dx + dy*dy) + dz*dz) in f32 with separate multiplies and adds (no FMA), which
 * is what aten evaluates for (diff**2).sum(dim=1).  The k smallest nodes of every point are written
 * in canonical slot order -- ascending (distance, node id) -- k-major:
 *   min_idx_i32[b][s*N + n] (always), min_idx_i64 (nullable; the dtype query_topk returns).
 * count [B][M] i32   = number of point copies assigned to each node (= mask.sum(1), networks.py:128)
 * sum_ws [B][3][M] f64 workspace = per-node coordinate sums of the assigned copies (f64 accumulation),
 *   consumed by sonet_som_group_f32.  Both are zeroed by the callee on `stream`.
 * ---------------------------------------------------------------------------------------------- */
int sonet_som_assign_f32(const float *x, const float *node, int B, int N, int M, int k,
                         int32_t *min_idx_i32, int64_t *min_idx_i64, int32_t *count, double *sum_ws,
                         sonet_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * som_group  -- replaces the dense grouping block of Encoder.forward
 *   reference: models/networks.py:128-172 (cluster mean :140-144, centers :168-169,
 *              de-centre :171, concat with normals :172)
 * Inputs: x, sn (nullable) [B][3][N] f32, min_idx_i32 [B][kN], count [B][M], sum_ws [B][3][M] f64.
 * Outputs (each nullable):
 *   som_node     [B][3][M]  = sum / (count + 1e-5)          (f32 division as the reference)
 *   row_max      [B][M] i32 = count > 0                     (util/som.py:267 mask_row_max)
 *   centers      [B][3][kN] = som_node gathered by min_idx
 *   x_decentered [B][3][kN] = x_stack - centers
 *   x_augmented  [B][6][kN] = cat(x_decentered, sn_stack)   (requires sn)
 * ---------------------------------------------------------------------------------------------- */
int sonet_som_group_f32(const float *x, const float *sn, const int32_t *min_idx_i32,
                        const int32_t *count, const double *sum_ws, int B, int N, int M, int k,
                        float *som_node, int32_t *row_max, float *centers, float *x_decentered,
                        float *x_augmented, sonet_stream_t stream);

/* one-hot mask [B][kN][M] i32 from min_idx (util/som.py:254-265); materialised only on request
 * (the reference's Encoder.mask attribute, read by models/segmenter.py:90). */
int sonet_som_mask_i32(const int32_t *min_idx_i32, int32_t *mask, int B, int kN, int M,
                       sonet_stream_t stream);

/* node_gather -- the segmenter's back-broadcast of node-level features to the kN point copies
 *   reference: models/segmenter.py:90-98 (argmax(mask) -> min_idx, three torch.gather calls)
 * feat [B][C][M] f32, min_idx_i32 [B][kN] -> out [B][C][kN] = feat[b][c][min_idx[b][j]]. */
int sonet_node_gather_f32(const float *feat, const int32_t *min_idx_i32, float *out,
                          int B, int C, int M, int kN, sonet_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * knn_gather  -- replaces knn_gather_by_indexing / knn_gather_wrapper
 *   reference: models/operations.py:19-54;  caller: models/layers.py:346,360
 * x [B][C][M] f32, knn_I [B][M][K] i64 (0 <= id < M), out [B][C][M][K] f32 = x[b][c][knn_I[b][m][j]].
 * ---------------------------------------------------------------------------------------------- */
int sonet_knn_gather_f32(const float *x, const int64_t *knn_I, float *out,
                         int B, int C, int M, int K, sonet_stream_t stream);

/* Self k-NN of the SOM nodes: knn_I [B][M][K] i64 = the K nearest nodes of every node (itself first), ascending
 * (distance, index), distance (dx*dx + dy*dy) + dz*dz.  Replaces the host-side faiss IndexFlatL2 search of the loaders
 * (data/modelnet_shrec_loader.py:116-150,257-259) and KNNModule's dense fallback (models/layers.py:333-337).  K <= 16. */
int sonet_knn_self_f32(const float *node, int64_t *knn_I, int B, int M, int K, sonet_stream_t stream);

/* KNNModule input in one pass (models/layers.py:313-350): out [B][3+C][M][K] = cat(coord[:, I] - center, feat[:, I]),
 * center [B][3][M] = mean of the K neighbour coordinates (center_avg != 0) or the node itself.  knn_I [B][M][K] i64. */
int sonet_knn_group_f32(const float *coord, const float *feat, const int64_t *knn_I, int B, int C, int M, int K,
                        int center_avg, float *center, float *out, sonet_stream_t stream);
/* KNNModule without the gathered tensor (models/layers.py:313-364, no-grad path).  sonet_knn_prepare_f32 writes the
 * neighbourhood centre [B][3][M] ("avg": sequential f32 mean of the K neighbours, else the node), the de-centred neighbour
 * coordinates K-MAJOR [B][3][K*M] (column k*M + m) and the int32 gather index of every column [B][K*M] (-1 where
 * knn_I is out of range: such a column reads as zeros, as in sonet_knn_group_f32).
 * sonet_pointmlp_h3_gather_f32 is sonet_pointmlp_h3_f32 with x1 = [B][C1][L1] read through that index: column l takes
 * x1[b][:, gidx[b][l]] -- the neighbour gather happens in the operand load; x2 [B][C2][L] and y [B][Cout][L] as usual.
 * sonet_planes_max_f32: out[row][m] = max_k x[row][k*M + m] (torch.max over the neighbourhood, layers.py:361-364, on the
 * k-major layout; NaN-propagating). */
int sonet_knn_prepare_f32(const float *coord, const int64_t *knn_I, int B, int M, int K, int center_avg,
                          float *center, float *dec, int32_t *gidx, sonet_stream_t stream);
int sonet_pointmlp_h3_gather_f32(const float *x1, int C1, int L1, const int32_t *gidx, const float *x2, int C2, const void *Wp3,
                                 const float *scale, const float *shift, int relu, float *y,
                                 int B, int Cout, int L, sonet_stream_t stream);
int sonet_planes_max_f32(const float *x, float *out, long long rows, int K, int M, sonet_stream_t stream);

/* out[row] = max over the K contiguous values of each of `rows` rows (NaN propagates, as torch.amax):
 * the neighbourhood max of KNNModule (models/layers.py:365) and the max over nodes (models/networks.py:197). */
int sonet_lastdim_max_f32(const float *x, float *out, long long rows, int K, sonet_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * pointmlp  -- fused point-wise layer: Conv1d/Conv2d(kernel 1) + per-channel affine + ReLU
 *   reference: models/layers.py:282-296 (EquivariantLayer.forward), :199-211 (MyConv2d.forward),
 *              BN at :60-70 / :112-120 (F.batch_norm); concat of models/layers.py:431 fused as x2.
 *   y[b][o][l] = act( (sum_i W[o][i] * xcat[b][i][l]) * scale[o] + shift[o] ),
 *   xcat = concat(x1 [B][C1][L], x2 [B][C2][L] (nullable, C2 = 0)) along channels, Cin = C1 + C2.
 * Eval mode folds bias and BatchNorm running statistics into (scale, shift) on the host side;
 * a layer without BN passes scale = 1, shift = bias.  relu != 0 applies max(.,0) (NaN propagates).
 * Arithmetic: exact-f32 MFMA (v_mfma_f32_32x32x2_f32), i.e. an f32 fma chain over i.
 * W is passed PACKED: sonet_pointmlp_pack_size(Cin, Cout) floats produced by
 * sonet_pointmlp_pack_f32 from the row-major [Cout][Cin] weight (device to device, on `stream`).
 * ---------------------------------------------------------------------------------------------- */
size_t sonet_pointmlp_pack_size(int Cin, int Cout);
int sonet_pointmlp_pack_f32(const float *W, float *Wp, int Cin, int Cout, sonet_stream_t stream);
int sonet_pointmlp_f32(const float *x1, int C1, const float *x2, int C2, const float *Wp,
                       const float *scale, const float *shift, int relu, float *y,
                       int B, int Cout, int L, sonet_stream_t stream);

/* The input gradient of a layer behind a training-mode BatchNorm (+ ReLU) (models/layers.py:60-70, :282-296 in the backward) with the
 * BatchNorm / ReLU backward applied by the operand load:  y = (W . g_raw) * scale + shift,
 *     g_raw[k] = a[k] * (relu && !(raw * sc[k] + sh[k] > 0) ? 0 : gy) + b[k] * raw + c0[k]
 * (gy, raw [B][C][L]; a, b, c0 from sonet_bn_bwd_coeffs_f32, sc, sh the forward's normalisation) -- what sonet_pointwise_bwd_apply_f32 followed
 * by sonet_pointmlp_x3_f32 compute, bit for bit, in one pass over (gy, raw).  g_raw_out (or NULL) receives g_raw: the weight gradient's operand.
 * Wp3: the bf16-split pack of the C x Cout matrix (W^T of the layer); C <= 512, Cout % 32 == 0.
 * praw / psc / psh / prelu / pstats_ws / psums: VARIANTS build only (the product library returns SONET_ERR_UNSUPPORTED unless they are NULL / 0:
 * the epilogue measured slower than the pass it replaces, docs/findings.md R5.9).  All or none: y is gy of the layer BELOW; with that layer's raw output praw [B][Cout][L] and
 * normalisation psc, psh [Cout] the epilogue also returns its BatchNorm-backward sums psums[0 .. Cout) = sum of gy * mask, psums[Cout .. 2 Cout) =
 * sum of gy * mask * praw (double; what sonet_pointwise_bwd_stats_f32 computes from one more pass over (gy, praw)); pstats_ws:
 * sonet_pointmlp_stats_ws_size(B, Cout, L) bytes. */
int sonet_pointmlp_x3_bnb_f32(const float *gy, const float *raw, int C, const void *Wp3, const float *scale, const float *shift,
                              const float *a, const float *b, const float *c0, const float *sc, const float *sh, int relu,
                              float *g_raw_out, float *y, int B, int Cout, int L,
                              const float *praw, const float *psc, const float *psh, int prelu, void *pstats_ws, double *psums,
                              sonet_stream_t stream);

/* The same launch with an accumulating store: y = (W . g_raw) * scale + shift + yadd, yadd [B][Cout][L] another gradient of the same tensor
 * that was computed earlier (the first layer's output of the first PointNet feeds the second layer AND the last one, models/layers.py:417-431:
 * autograd would add the two gradients in a pass of its own).  The f32 sum is the one that pass would store; yadd == y is allowed. */
int sonet_pointmlp_x3_bnb_acc_f32(const float *gy, const float *raw, int C, const void *Wp3, const float *scale, const float *shift,
                                  const float *a, const float *b, const float *c0, const float *sc, const float *sh, int relu,
                                  float *g_raw_out, const float *yadd, float *y, int B, int Cout, int L, sonet_stream_t stream);

/* The same layer on bf16 MFMA with a 3-way bf16 split of both operands (6 MFMAs per product term set):
 * f32-class accuracy (classifier forward within 3e-6 * max(|ref|, rms) of the reference; tolerance 1e-5) at
 * 6/16 of the f32-MFMA cost.  Requires Cout % 32 == 0 and, with a second input, C1 % 16 == 0.
 * Wp3 = sonet_pointmlp_x3_pack_size(Cin, Cout) BYTES produced by sonet_pointmlp_x3_pack. */
size_t sonet_pointmlp_x3_pack_size(int Cin, int Cout);
int sonet_pointmlp_x3_pack(const float *W, void *Wp3, int Cin, int Cout, sonet_stream_t stream);
int sonet_pointmlp_x3_f32(const float *x1, int C1, const float *x2, int C2, const void *Wp3,
                          const float *scale, const float *shift, int relu, float *y,
                          int B, int Cout, int L, sonet_stream_t stream);

/* The same layer on a THREE-term fp16 split (v_mfma_f32_32x32x16_f16): half the matrix work at the same 3e-6 accuracy,
 * but an fp16 operand range (|x| <= 2047, clamped; magnitudes below ~1e-4 lose relative precision): forward
 * activations and coordinates, not gradients.  Packed size = sonet_pointmlp_x3_pack_size. */
int sonet_pointmlp_h3_pack(const float *W, void *Wp3, int Cin, int Cout, sonet_stream_t stream);
int sonet_pointmlp_h3_f32(const float *x1, int C1, const float *x2, int C2, const void *Wp3,
                          const float *scale, const float *shift, int relu, float *y,
                          int B, int Cout, int L, sonet_stream_t stream);

/* Third generation of the fp16-split layer: PRE-SPLIT activations ("P16" planes; so-net_amd/csrc/pointmlp_h3p.hip).
 *   reference: models/layers.py:282-296 (EquivariantLayer.forward) / :313-367 (the 1x1 Conv2d layers of KNNModule).
 * P16 layout of a B x C x L activation: P[b][kc][form][h][l][8] fp16, kc < ceil(C/16); form 0 = fp16(32 x), form 1 = fp16(32 x - form 0);
 * element e of half h is channel 16 kc + 4 h + (e & 3) + 8 (e >> 2), zeros past C: sonet_p16_size(B, C, L) = 64 B ceil(C/16) L bytes (the
 * bytes of the f32 tensor).  The split (clamp to +-2047, scale, two roundings) runs once where the activation is PRODUCED -- in the
 * epilogue of sonet_pointmlp_h3p (output yp) or in sonet_p16_from_f32 (optionally through a per-channel affine + ReLU: the training
 * forward's normalise pass) -- and the consuming layer's operand loads are finished MFMA B fragments.  The producer reports its largest
 * post-activation magnitude in word 2 of the range log.  sonet_p16_to_f32 returns (form 0 + form 1) / 32.
 * sonet_pointmlp_h3p: y = act((W . cat(x1, x2) [+ zadd[b][o][zidx[b][l]]]) * scale + shift); x1p / x2p P16 (x1: C1 channels x L1 columns,
 * read through gidx [B][L] i32 when given -- out of range: zeros --, L1 = L otherwise; C1 % 16 == 0 when x2 is given), Wp from
 * sonet_pointmlp_h3p_pack (sonet_pointmlp_h3p_pack_size bytes; K slots in P16 channel order, weights as fp16(32 w) + fp16 residual,
 * |w| <= 2047 logged in word 1 of the range log), outputs y (f32 [B][Cout][L]) and / or yp (P16); Cout % 32 == 0.
 * stats_ws / mean / var (all or none; y only): BatchNorm batch statistics of y from the epilogue, as sonet_pointmlp_h3_stats_f32
 * (stats_ws: sonet_pointmlp_h3p_stats_ws_size bytes).  zadd [B][Cout][ZM] f32 / zidx [B][L] i32: as sonet_pointmlp_h3_nodeadd_f32. */
size_t sonet_p16_size(int B, int C, int L);
int sonet_p16_from_f32(const float *x, void *p16, int B, int C, int L, const float *scale, const float *shift, int relu,
                       sonet_stream_t stream);
int sonet_p16_to_f32(const void *p16, float *x, int B, int C, int L, sonet_stream_t stream);
size_t sonet_pointmlp_h3p_pack_size(int Cin, int Cout);
int sonet_pointmlp_h3p_pack(const float *W, void *Wp, int Cin, int Cout, sonet_stream_t stream);
size_t sonet_pointmlp_h3p_stats_ws_size(int B, int Cout, int L);
int sonet_pointmlp_h3p(const void *x1p, int C1, int L1, const int32_t *gidx, const void *x2p, int C2, const void *Wp,
                       const float *scale, const float *shift, int relu, float *y, void *yp, int B, int Cout, int L,
                       const float *zadd, const int32_t *zidx, int ZM, void *stats_ws, float *mean, float *var,
                       sonet_stream_t stream);

/* No-grad node-level stage on the third-generation layer (so-net_amd/csrc/node_stage.hip, pointmlp_h3p.hip): KNNModule + the final
 * PointNet + the global max (models/layers.py:313-367,384-387, models/networks.py:187-197) on a FLAT column axis -- ONE "cloud" whose
 * columns are the B x M nodes of the batch (column b M + m; the axis is padded with zero columns to Lm = sonet_node_stage_columns(B, M),
 * a multiple of 128, which is the column count of every M-level tensor below), so that 64 clouds x 64 nodes fill a launch.
 * sonet_pointresnet_fused_pool_p16_f32 (below) hands the pooled map over as such planes; a sonet_pointmlp_h3p launch (B = 1, L = B M, f32
 * out) applies the 384-channel block of KNNModule's first layer ONCE per node: z (the layer is linear; models/layers.py:351).
 * The rest of that layer per neighbour copy (models/layers.py:319-352),
 *   h1[c][n, k] = act(scale[c] (z[c][b M + I[b][m][k]] + wl[c][0..2] . (coord[b][:, I[b][m][k]] - center[b][:, m])) + shift[c]),
 * as P16 planes of a 1 x C x Lp activation, Lp = sonet_knn_stage_columns(B, M, K): every 128-column block holds G = min(16, 128 / K) nodes, the K
 * neighbour copies of a node next to each other (column 128 i + g K + k = neighbour k of node i G + g), zero padding behind them, in two launches:
 * sonet_knn_stage_prepare_f32 (needs the node coordinates only) writes rec [Lp] int4 = (source column b M + I on the flat axis -- -1: index
 *   outside [0, M), features read as zeros as in sonet_knn_group_f32; -2: padding column --, the three de-centred coordinates as f32 bits),
 *   center [B][3][M] f32 = mean of the K neighbour coordinates (center_avg) or the node (KNNModule's first output) and center_p16 = the same as a
 *   one-chunk P16 panel (sonet_p16_size(1, 3, Lm) bytes: the 3 leading channels of the final PointNet's input); knn_I [B][M][KI] i64, first K used;
 * sonet_knn_stage_input_p16: z_p16 = P16 planes of the 1 x C x Lm map z (sonet_pointmlp_h3p, yp output), wl [C][3] -> h1_p16.  Largest |h1| -> word 2
 *   of the range log.
 * sonet_pointmlp_h3p_gmax: the layer followed by a max over groups of GK consecutive columns in ONE launch (L % 128 == 0, one cloud):
 * every 128-column block holds G groups (G GK <= 128), group g of block i is output column i G + g (< ngout).  Exactly one output:
 * yp = P16 planes (Lout >= ngout columns, the first ngout written) of the maxima (KNNModule's max over the neighbours, models/layers.py:365; G <= 16) or y = f32
 * [ngout][Cout] (the global max over a cloud's M nodes = the feature vector, models/networks.py:197; G <= 2, GK % 4 == 0).  NaN wins.
 * sonet_p16_flat_to_bcm_f32: planes of a flat 1 x C x (B M) activation -> f32 [B][C][M] (an intermediate map a caller asks for). */
size_t sonet_knn_stage_columns(int B, int M, int K);
size_t sonet_node_stage_columns(int B, int M);
int sonet_knn_stage_prepare_f32(const float *coord, const int64_t *knn_I, int KI, int center_avg, int B, int M, int K,
                                float *center, void *center_p16, void *rec, sonet_stream_t stream);
int sonet_knn_stage_input_p16(const void *rec, const void *z_p16, const float *wl, const float *scale, const float *shift, int relu,
                              int B, int M, int K, int C, void *h1_p16, sonet_stream_t stream);
int sonet_pointmlp_h3p_gmax(const void *x1p, int C1, const void *x2p, int C2, const void *Wp, const float *scale, const float *shift,
                            int relu, int Cout, int L, int GK, int G, int ngout, int Lout, float *y, void *yp, sonet_stream_t stream);
int sonet_p16_flat_to_bcm_f32(const void *p16, float *x, int B, int C, int M, sonet_stream_t stream);

/* Node-level pieces of the training step (so-net_amd/csrc/node_train.hip).
 * sonet_lastdim_argmax_*: out[r] = max over the K contiguous values of row r, idx[r] = index of the FIRST maximum (a NaN wins) -- torch.max
 *   over the K' neighbours of KNNModule (models/layers.py:350-365) and over the M nodes (models/networks.py:197) with the routing its
 *   backward needs; sonet_lastdim_max_bwd_*: gx[r][k] = (k == idx[r]) ? g[r] : 0 (the whole row is written).
 * sonet_knn_gather_bwd_*: backward of knn_gather_by_indexing (models/operations.py:38-54): gx[b][c][m] = sum of g[b][c][m'][k] over the
 *   (m', k) with knn_I[b][m'][k] == m, gathered over per-cloud inverse lists in a fixed order (f32 accumulation, bitwise reproducible);
 *   ws: sonet_knn_gather_bwd_ws_size(B, M, K) bytes; M <= 1024; indices outside [0, M) contribute nowhere. */
int sonet_lastdim_argmax_f32(const float *x, float *out, int32_t *idx, long long rows, int K, sonet_stream_t stream);
int sonet_lastdim_argmax_bf16(const uint16_t *x, uint16_t *out, int32_t *idx, long long rows, int K, sonet_stream_t stream);
int sonet_lastdim_max_bwd_f32(const float *g, const int32_t *idx, float *gx, long long rows, int K, sonet_stream_t stream);
int sonet_lastdim_max_bwd_bf16(const uint16_t *g, const int32_t *idx, uint16_t *gx, long long rows, int K, sonet_stream_t stream);
size_t sonet_knn_gather_bwd_ws_size(int B, int M, int K);
int sonet_knn_gather_bwd_f32(const float *g, const int64_t *knn_I, float *gx, void *ws, int B, int C, int M, int K, sonet_stream_t stream);
int sonet_knn_gather_bwd_bf16(const uint16_t *g, const int64_t *knn_I, float *gx, void *ws, int B, int C, int M, int K, sonet_stream_t stream);

/* The same layer with bf16 STORAGE and bf16 MFMA (BASELINE configs[1] "bf16"; the reference is f32-only, so this is the
 * reduced-precision twin of models/layers.py:282-296, not a bit-compatible replacement): x1, x2, y are bfloat16 bit
 * patterns [B][C][L], one v_mfma_f32_32x32x16_bf16 per product with f32 accumulation, the epilogue
 * act(acc * scale + shift) in f32, rounded to nearest-even bf16 on the store.  Requires Cout % 32 == 0 and, with a second
 * input, C1 % 16 == 0.  Wp = sonet_pointmlp_bf16_pack_size(Cin, Cout) BYTES from sonet_pointmlp_bf16_pack (f32 weights in).
 * _gather: column l of x1 ([B][C1][L1]) is x1[b][:, gidx[b][l]] (zeros when out of range), as sonet_pointmlp_h3_gather_f32. */
size_t sonet_pointmlp_bf16_pack_size(int Cin, int Cout);
int sonet_pointmlp_bf16_pack(const float *W, void *Wp, int Cin, int Cout, sonet_stream_t stream);
int sonet_pointmlp_bf16(const uint16_t *x1, int C1, const uint16_t *x2, int C2, const void *Wp,
                        const float *scale, const float *shift, int relu, uint16_t *y,
                        int B, int Cout, int L, sonet_stream_t stream);
int sonet_pointmlp_bf16_gather(const uint16_t *x1, int C1, int L1, const int32_t *gidx, const uint16_t *x2, int C2, const void *Wp,
                               const float *scale, const float *shift, int relu, uint16_t *y,
                               int B, int Cout, int L, sonet_stream_t stream);
/* sonet_pointmlp_bf16 with an ACCUMULATING store: y = bf16(float(bf16(result)) + float(yadd)); yadd [B][Cout][L] bf16 = another gradient of the
 * same tensor, computed earlier (models/layers.py:417-431: the first layer's output feeds the second layer and the last one) -- what autograd's
 * accumulation of the two bf16 tensors would store, bit for bit, without its pass over three tensors.  yadd == y allowed.  Even L, 4-byte
 * aligned rows. */
int sonet_pointmlp_bf16_acc(const uint16_t *x1, int C1, const uint16_t *x2, int C2, const void *Wp,
                            const float *scale, const float *shift, int relu, const uint16_t *yadd, uint16_t *y,
                            int B, int Cout, int L, sonet_stream_t stream);
/* The input gradient of a bf16 layer behind a training-mode BatchNorm (+ ReLU), the BatchNorm / ReLU backward applied by the operand load
 * (replaces autograd's backward of models/layers.py:60-70 + :282-296 for the layer below; bf16 twin of sonet_pointmlp_x3_bnb_f32 / _bnb_acc_f32):
 *   y = bf16((W . g_raw) * scale + shift) [+ yadd],  g_raw[k] = bf16(a[k] * (relu && !(raw * sc[k] + sh[k] > 0) ? 0 : gy) + b[k] * raw + c0[k])
 * gy, raw [B][C][L] bf16; a, b, c0 (sonet_bn_bwd_coeffs_f32) and sc, sh (the forward's normalisation) [C] f32.  Bit for bit what
 * sonet_pointwise_bwd_apply_bf16 followed by sonet_pointmlp_bf16 (yadd NULL) or sonet_pointmlp_bf16_acc computes, in ONE pass over (gy, raw).
 * g_raw_out [B][C][L] (or NULL) receives g_raw for the weight gradient.  Wp: sonet_pointmlp_bf16_pack of the C x Cout matrix.
 * SONET_ERR_UNSUPPORTED unless L even, rows 4-byte aligned, C % 16 == 0, 32 <= C <= 512, Cout % 64 == 0. */
int sonet_pointmlp_bf16_bnb(const uint16_t *gy, const uint16_t *raw, int C, const void *Wp, const float *scale, const float *shift,
                            const float *a, const float *b, const float *c0, const float *sc, const float *sh, int relu,
                            uint16_t *g_raw_out, const uint16_t *yadd, uint16_t *y, int B, int Cout, int L, sonet_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * pointresnet_fused -- the encoder's first PointNet as ONE kernel (eval mode, 3xbf16-split arithmetic)
 *   reference: models/layers.py:419-432 (PointResNet.forward) as built at models/networks.py:82-83:
 *              Cin0 (<= 16) -> 64 -> 128 -> 256 -> [64 + 256] -> 384, BN + ReLU on the first three layers.
 * x [B][Cin0][L] f32 -> y [B][384][L] f32; intermediate activations never leave the registers.
 * wstream: sonet_pointresnet_pack_size() bytes from sonet_pointresnet_pack (the four row-major weights
 *   [64][Cin0], [128][64], [256][128], [384][320] split into bf16 terms and laid out in MFMA consumption order);
 * affine: 832 (scale, shift) float pairs, channels of layer 1, 2, 3, 4 concatenated (eval BN + bias folded;
 *   layer 4: scale 1, shift bias).
 * ---------------------------------------------------------------------------------------------- */
size_t sonet_pointresnet_pack_size(void);
int sonet_pointresnet_pack(const float *W1, const float *W2, const float *W3, const float *W4, int Cin0,
                           void *wstream, sonet_stream_t stream);
int sonet_pointresnet_fused_f32(const float *x, int Cin0, const void *wstream, const float *affine,
                                float *y, int B, int L, sonet_stream_t stream);
/* The same launch also writes y pre-split: yp = the P16 planes of y (sonet_p16_size(B, 384, L) bytes; layout under sonet_pointmlp_h3p
 * above), the operand format of sonet_pointmlp_h3p -- the part segmenter's first layer reads first_pn_out per point copy
 * (models/networks.py:296-326, models/segmenter.py:90-109).  The largest split magnitude joins word 2 of the range log.
 * y == NULL: only the planes are written (the per-node pool then runs on them: sonet_index_max_gather_p16). */
int sonet_pointresnet_fused_p16_f32(const float *x, int Cin0, const void *wstream, const float *affine,
                                    float *y, void *yp, int B, int L, sonet_stream_t stream);

/* bf16 twin of sonet_pointresnet_fused_f32 (BASELINE configs[1] "bf16"): one bf16 MFMA per product, activations rounded to
 * bf16 between the layers (what the layer-wise sonet_pointmlp_bf16 launches would store), f32 accumulation; x [B][Cin0][L] f32
 * -> y [B][384][L] bfloat16 bits.  wstream: sonet_pointresnet_bf16_pack_size() bytes from sonet_pointresnet_bf16_pack (the
 * same four row-major f32 weights); affine as above (832 (scale, shift) pairs). */
size_t sonet_pointresnet_bf16_pack_size(void);
int sonet_pointresnet_bf16_pack(const float *W1, const float *W2, const float *W3, const float *W4, int Cin0,
                                void *wstream, sonet_stream_t stream);
int sonet_pointresnet_bf16(const float *x, int Cin0, const void *wstream, const float *affine,
                           uint16_t *y, int B, int L, sonet_stream_t stream);

/* ... and of sonet_pointresnet_fused_pool_f32 (same inputs from sonet_som_sort_group_f32, same workspace protocol with
 * sonet_pointresnet_bf16_pool_ws_size bytes): out [B][384][M] f32 = per-node maximum of the bf16 features, i.e. exactly
 * index_max_gather_bf16 of the tensor sonet_pointresnet_bf16 would write (bf16 rounding is monotone), which never reaches HBM. */
size_t sonet_pointresnet_bf16_pool_ws_size(int B, int L, int M);
int sonet_pointresnet_bf16_pool(const float *x_sorted, int Cin0, const void *wstream, const float *affine,
                                const int32_t *ids_sorted, const int32_t *pos0, const int32_t *node_off,
                                const int32_t *count, void *ws, float *out, int B, int L, int M, sonet_stream_t stream);

/* Fused first PointNet + per-node max-pool (the no-grad classifier / autoencoder path): nothing of the 23 MB/cloud
 * first_pn_out reaches HBM.  Replaces models/networks.py:175-185 (PointResNet, index_max, masked gather) when only
 * first_pn_out_masked_max is needed.  Inputs come from sonet_som_sort_group_f32 (point copies sorted by node):
 *   x_sorted [B][Cin0][L], ids_sorted [B][L] (non-decreasing per cloud), pos0 [B] (sorted position of copy 0),
 *   node_off [B][M] (first sorted position of each node), count [B][M].
 * out [B][384][M] f32 = max over the node's copies (values > -1000), else the features of copy 0 (the reference's
 * gather index 0).  Two kernels, no atomics on the common path: every 128-point tile stores the per-node maxima of
 * the few nodes it touches, a second kernel combines the tiles of each node.  ws: sonet_pointresnet_pool_ws_size bytes. */
size_t sonet_pointresnet_pool_ws_size(int B, int L, int M);
int sonet_pointresnet_fused_pool_f32(const float *x_sorted, int Cin0, const void *wstream, const float *affine,
                                     const int32_t *ids_sorted, const int32_t *pos0, const int32_t *node_off,
                                     const int32_t *count, void *ws, float *out, int B, int L, int M, sonet_stream_t stream);
/* The same, and the decode pass also writes the pooled map pre-split on the flat column axis of the node-level stage: out_p16 = P16
 * planes of a 1 x 384 x Lm activation (Lm = sonet_node_stage_columns(B, M), column b M + m; the pad columns are not written); largest pooled magnitude -> word 2 of
 * the range log (its consumers cannot check the clamp any more). */
int sonet_pointresnet_fused_pool_p16_f32(const float *x_sorted, int Cin0, const void *wstream, const float *affine,
                                         const int32_t *ids_sorted, const int32_t *pos0, const int32_t *node_off,
                                         const int32_t *count, void *ws, float *out, void *out_p16, int B, int L, int M, sonet_stream_t stream);
/* The SOM stage of the no-grad pooled path in TWO launches: sonet_som_assign_f32 + sonet_som_sort_group_f32 in one call
 * (util/som.py:237-269 + models/networks.py:128-172).  Same outputs -- min_idx_i32 [B][k*N] (k-major; min_idx_i64 optional),
 * count [B][M], sum_ws [B][3][M] f64 (optional), som_node [B][3][M], row_max [B][M] (both optional), x_aug_sorted [B][6][kN],
 * ids_sorted [B][kN], pos0 [B], node_off [B][M] -- with no clear launches, no global atomics and one LDS atomic per point copy:
 * per-workgroup partial counts / sums and per-copy ranks in ws (sonet_som_assign_sort_ws_size bytes), sorted positions = node
 * offset + the counts of the workgroups before + rank.  Node ids and counts are bit-identical to the two separate calls; the
 * order of the copies inside a node differs (any order serves the per-node max-pool). */
size_t sonet_som_assign_sort_ws_size(int B, int N, int M, int k);
int sonet_som_assign_sort_f32(const float *x, const float *sn, const float *node, int B, int N, int M, int k,
                              int32_t *min_idx_i32, int64_t *min_idx_i64, int32_t *count, double *sum_ws,
                              float *som_node, int32_t *row_max, float *x_aug_sorted, int32_t *ids_sorted,
                              int32_t *pos0, int32_t *node_off, void *ws, sonet_stream_t stream);
/* ... with a deterministic order inside a node (wave, slot, lane order of a 512-point workgroup instead of the arrival order of LDS atomics):
 * the same sorted copy in every run.  The f32-class TRAINING forward uses it (its BatchNorm batch sums run over the sorted columns); the
 * no-grad forward, which only takes maxima over a node's points, keeps the atomics. */
int sonet_som_assign_sort_det_f32(const float *x, const float *sn, const float *node, int B, int N, int M, int k,
                              int32_t *min_idx_i32, int64_t *min_idx_i64, int32_t *count, double *sum_ws,
                              float *som_node, int32_t *row_max, float *x_aug_sorted, int32_t *ids_sorted,
                              int32_t *pos0, int32_t *node_off, void *ws, sonet_stream_t stream);
/* ... whose second launch also does sonet_knn_stage_prepare_f32 (below) on the cluster means it computes: center [B][3][M], center_p16 and rec
 * are bit-identical to the separate launch on som_node, and the no-grad forward needs no launch for KNNModule's index / coordinate side
 * (models/layers.py:319-350). */
int sonet_som_assign_sort_knn_f32(const float *x, const float *sn, const float *node, int B, int N, int M, int k,
                                  int32_t *min_idx_i32, int64_t *min_idx_i64, int32_t *count, double *sum_ws,
                                  float *som_node, int32_t *row_max, float *x_aug_sorted, int32_t *ids_sorted,
                                  int32_t *pos0, int32_t *node_off, void *ws,
                                  const int64_t *knn_I, int KI, int K, int center_avg, float *center, void *center_p16, void *rec,
                                  sonet_stream_t stream);
/* som_sort_group: som_group with the kN point copies of every cloud counting-sorted by node id.
 * x_aug_sorted [B][6][kN], ids_sorted [B][kN], pos0 [B], node_off [B][M]; cursor_ws: B*M i32 (zeroed by the callee). */
int sonet_som_sort_group_f32(const float *x, const float *sn, const int32_t *min_idx_i32, const int32_t *count,
                             const double *sum_ws, int B, int N, int M, int k, float *som_node, int32_t *row_max,
                             float *x_aug_sorted, int32_t *ids_sorted, int32_t *pos0, int32_t *node_off,
                             int32_t *cursor_ws, sonet_stream_t stream);

/* bf16 twins of the training-mode element-wise passes below (BASELINE configs[1]: bf16 forward + backward): tensors are
 * bfloat16 bit patterns [B][C][L], the arithmetic is the f32 / f64 one of the *_f32 entry points on the widened values, results
 * are rounded to nearest-even bf16.  sonet_channel_stats_bf16 = sonet_channel_stats_f32 (mean, biased variance, f64 sums). */
int sonet_channel_stats_bf16(const uint16_t *y, int B, int C, int L, double *stat_ws, float *mean, float *var, sonet_stream_t stream);
int sonet_channel_affine_act_out_bf16(const uint16_t *x, const float *scale, const float *shift, int relu, uint16_t *y,
                                      int B, int C, int L, sonet_stream_t stream);
int sonet_pointwise_bwd_stats_bf16(const uint16_t *gy, const uint16_t *raw, const float *scale, const float *shift,
                                   int relu, int B, int C, int L, double *sums, sonet_stream_t stream);
int sonet_pointwise_bwd_apply_bf16(const uint16_t *gy, const uint16_t *raw, const float *scale, const float *shift, int relu,
                                   const float *a, const float *b, const float *c0, uint16_t *g_raw, int B, int C, int L,
                                   sonet_stream_t stream);

/* Backward of act(BN(W x + b)) around the GEMMs (training-mode models/layers.py:282-296, :60-70), two passes:
 *   stats: sums[c] = sum gy*mask, sums[C+c] = sum gy*mask*raw over (b, l), f64 (zeroed by the callee);
 *   apply: g_raw = a[c]*(gy*mask) + b[c]*raw + c0[c];
 * mask = (fma(raw, scale[c], shift[c]) > 0) if relu else 1 -- the forward's own affine, bit for bit.
 * gy, raw, g_raw: [B][C][L] f32; scale, shift, a, b, c0: [C] f32; sums: [2C] f64. */
int sonet_pointwise_bwd_stats_f32(const float *gy, const float *raw, const float *scale, const float *shift,
                                  int relu, int B, int C, int L, double *sums, sonet_stream_t stream);
int sonet_pointwise_bwd_apply_f32(const float *gy, const float *raw, const float *scale, const float *shift, int relu,
                                  const float *a, const float *b, const float *c0, float *g_raw,
                                  int B, int C, int L, sonet_stream_t stream);
/* t = act((t + z[b][c][min_idx[b][l]]) * scale[c] + shift[c]) in place.  t [B][C][L] (the per-point block of a layer's
 * pre-activation), z [B][C][M] (its per-node block, computed once per node), min_idx [B][L] i32 node of every point
 * copy.  Used for the first Segmenter layer, whose 3356 input channels are 393 per-point, 1923 per-node and 1040
 * per-cloud ones (models/networks.py:296-326, models/segmenter.py:90-109). */
int sonet_node_add_affine_act_f32(float *t, const float *z, const int32_t *min_idx_i32, const float *scale,
                                  const float *shift, int relu, int B, int C, int L, int M, sonet_stream_t stream);

/* sonet_pointmlp_h3_f32 / sonet_pointmlp_x3_f32 that also return mean[c] and the biased variance var[c] of y over (B, L) --
 * BatchNorm's batch statistics (models/layers.py:60-70) -- from the kernel's epilogue instead of a second pass over y: per-workgroup
 * (sum, sum of squares) partials in f64 (rows reduced over a wave's 32 columns in f32), summed in a fixed order.
 * stats_ws: sonet_pointmlp_stats_ws_size(B, Cout, L) bytes. */
size_t sonet_pointmlp_stats_ws_size(int B, int Cout, int L);
int sonet_pointmlp_h3_stats_f32(const float *x1, int C1, const float *x2, int C2, const void *Wp3, const float *scale,
                                const float *shift, int relu, float *y, int B, int Cout, int L, void *stats_ws,
                                float *mean, float *var, sonet_stream_t stream);
int sonet_pointmlp_x3_stats_f32(const float *x1, int C1, const float *x2, int C2, const void *Wp3, const float *scale,
                                const float *shift, int relu, float *y, int B, int Cout, int L, void *stats_ws,
                                float *mean, float *var, sonet_stream_t stream);

/* sonet_pointmlp_h3_f32 with a per-node addend gathered in the epilogue: y = act((W . cat(x1, x2) + zadd[b][o][zidx[b][l]]) * scale
 * + shift), zadd [B][Cout][ZM] f32, zidx [B][L] i32 (out of range: + 0).  The first Segmenter layer (models/networks.py:296-326,
 * models/segmenter.py:90-109): its per-node and per-cloud input channels are multiplied once per node (zadd) instead of once per
 * point copy; replaces sonet_pointmlp_h3_f32 + sonet_node_add_affine_act_f32 (one pass over the 1024-channel tensor less). */
int sonet_pointmlp_h3_nodeadd_f32(const float *x1, int C1, const float *x2, int C2, const void *Wp3, const float *scale,
                                  const float *shift, int relu, float *y, int B, int Cout, int L,
                                  const float *zadd, const int32_t *zidx, int ZM, sonet_stream_t stream);

/* bf16 twin: statistics of the STORED bf16 values (what the normalise pass and the backward read). */
size_t sonet_pointmlp_bf16_stats_ws_size(int B, int Cout, int L);
int sonet_pointmlp_bf16_stats(const uint16_t *x1, int C1, const uint16_t *x2, int C2, const void *Wp,
                              const float *scale, const float *shift, int relu, uint16_t *y,
                              int B, int Cout, int L, void *stats_ws, float *mean, float *var, sonet_stream_t stream);
/* The bf16 layer and the per-node arg-max pool of its output in ONE launch; the output itself is never written: the last (norm-free) layer
 * of the first PointNet in training when only the pooled map is consumed -- classifier, autoencoder -- (models/layers.py:431 +
 * models/networks.py:180-185, models/index_max_ext/index_max_cuda.cu:10-26).  out_idx [B][Cout][M] = what sonet_index_max_bf16 reports on the
 * tensor sonet_pointmlp_bf16 would have written (first maximum above -1000 in column order, else 0), out_val [B][Cout][M] f32 = that tensor's
 * value at out_idx * row_max (sonet_index_max_gather_bf16): bit for bit.  ids [B][L] i32 = node of every column, row_max [B][M] i32 or NULL.
 * Needs: even L < 65535, 4-byte aligned rows, (C1 + C2) % 64 == 0, Cout % 32 == 0, M <= 255; SONET_ERR_UNSUPPORTED otherwise. */
int sonet_pointmlp_bf16_pool(const uint16_t *x1, int C1, const uint16_t *x2, int C2, const void *Wp,
                             const float *scale, const float *shift, int relu, const int32_t *ids, const int32_t *row_max,
                             int32_t *out_idx, float *out_val, int B, int Cout, int L, int M, sonet_stream_t stream);
/* Normalise-on-load (bf16 training forward, round 6): x1 / x2 hold the RAW (bf16) outputs of training-mode BatchNorm layers
 * (models/layers.py:60-70, :282-296) whose normalise + ReLU pass was never run.  The operand load computes act(raw * xs[c] + xh[c]) in f32 and
 * rounds to bf16 -- exactly what sonet_channel_affine_act_bf16 would have stored -- so the results equal the plain entry points on the
 * normalised tensors bit for bit.  xs1, xh1 [C1] (xs2, xh2 [C2] when C2 > 0); xrelu bit 0 / 1: ReLU on x1 / x2.  Streaming-kernel shapes only
 * ((C1 + C2) % 64 == 0, even L, 4-byte aligned rows; the statistics form: >= 8192 column groups): SONET_ERR_UNSUPPORTED otherwise. */
int sonet_pointmlp_bf16_stats_xaff(const uint16_t *x1, int C1, const uint16_t *x2, int C2, const void *Wp,
                                   const float *scale, const float *shift, int relu, uint16_t *y,
                                   int B, int Cout, int L, void *stats_ws, float *mean, float *var,
                                   const float *xs1, const float *xh1, const float *xs2, const float *xh2, int xrelu,
                                   sonet_stream_t stream);
int sonet_pointmlp_bf16_pool_xaff(const uint16_t *x1, int C1, const uint16_t *x2, int C2, const void *Wp,
                                  const float *scale, const float *shift, int relu, const int32_t *ids, const int32_t *row_max,
                                  int32_t *out_idx, float *out_val, int B, int Cout, int L, int M,
                                  const float *xs1, const float *xh1, const float *xs2, const float *xh2, int xrelu,
                                  sonet_stream_t stream);

/* f32-class twin on NODE-SORTED columns (sonet_som_sort_group_f32: ids_sorted [B][L] i32 non-decreasing per cloud, pos0 [B] = sorted position
 * of original column 0): the fp16-split layer (sonet_pointmlp_h3_f32) and the per-node arg-max pool of its output in one pass; the output is
 * never written.  out_idx [B][Cout][M] = winning SORTED column = first maximum above -1000 in sorted column order (what
 * models/index_max_ext/index_max_cuda.cu:10-26 reports on the sorted tensor; the sort kernels leave the order INSIDE a node to the arrival of
 * their atomics, so among columns of a node with exactly EQUAL values the winner need not be the one with the lowest original index -- the
 * pooled value is the same), pos0[b] where nothing beat -1000 or
 * row_max[b][m] == 0 (models/networks.py:185: gather index 0 of the original order); out_val [B][Cout][M] = the layer's value there (bit for
 * bit sonet_pointmlp_h3_f32 + sonet_index_max_gather_f32 on the sorted tensor; -0 reported as +0).  ws: sonet_pointmlp_h3_segpool_ws_size
 * bytes.  Cout % 32 == 0, C1 % 16 == 0 when x2 is given.
 *
 * Normalise-on-load (both entry points below): when xs1 is given the inputs are the RAW outputs of BatchNorm layers whose normalise + ReLU pass
 * was never run; the operand load applies x = act(raw * xs[c] + xh[c]) (xs1 / xh1 [C1], xs2 / xh2 [C2]; xrelu bit 0 / 1: ReLU on x1's / x2's
 * channels) exactly as sonet_channel_affine_act_f32 computes it -- the normalised activations of models/layers.py:60-70 never exist in
 * memory between two layers of the training forward.  Needs Cin <= 1024, Cout % 128 == 0. */
size_t sonet_pointmlp_h3_segpool_ws_size(int B, int Cout, int M);
int sonet_pointmlp_h3_segpool_f32(const float *x1, int C1, const float *x2, int C2, const void *Wp3, const float *scale,
                                  const float *shift, int relu, const int32_t *ids_sorted, const int32_t *pos0,
                                  const int32_t *row_max, int M, void *ws, int32_t *out_idx, float *out_val,
                                  int B, int Cout, int L, const float *xs1, const float *xh1, const float *xs2, const float *xh2,
                                  int xrelu, sonet_stream_t stream);
/* sonet_pointmlp_h3_stats_f32 with normalise-on-load (xs1 required). */
int sonet_pointmlp_h3_stats_xaff_f32(const float *x1, int C1, const float *x2, int C2, const void *Wp3, const float *scale,
                                     const float *shift, int relu, float *y, int B, int Cout, int L, void *stats_ws,
                                     float *mean, float *var, const float *xs1, const float *xh1, const float *xs2, const float *xh2,
                                     int xrelu, sonet_stream_t stream);

/* Weight gradient of a point-wise layer: dw[o][c] = sum_b sum_l g[b][o][l] * x[b][c][l]  (g [B][Cout][L], x [B][Cin][L], dw
 * [Cout][Cin], f32) -- what autograd computes for the nn.Conv1d / nn.Conv2d(1x1) weights of models/layers.py:282-296.  Both
 * operands are split into three bf16 pieces, six products kept (f32-class, f32 range), f32 accumulation on the matrix cores;
 * partial 128 x 128 blocks over column slices are summed in a fixed order (deterministic).  ws = sonet_wgrad_x3_ws_size bytes. */
size_t sonet_wgrad_x3_ws_size(int B, int Cout, int Cin, int L);
int sonet_wgrad_x3_f32(const float *g, const float *x, float *dw, void *ws, int B, int Cout, int Cin, int L, sonet_stream_t stream);
/* ... with normalise-on-load of x (see sonet_pointmlp_h3_stats_xaff_f32): x is the RAW output of a BatchNorm layer, the operand split applies
 * x = act(raw * xs[c] + xh[c]) first (xs, xh [Cin]; xrelu != 0: ReLU). */
int sonet_wgrad_x3_xaff_f32(const float *g, const float *x, float *dw, void *ws, int B, int Cout, int Cin, int L,
                            const float *xs, const float *xh, int xrelu, sonet_stream_t stream);
/* bf16 twin (BASELINE configs[1] "bf16"): g [B][Cout][L], x [B][Cin][L] as bfloat16 bit patterns (16-byte aligned), one bf16 MFMA per
 * product, f32 accumulation, f32 partial blocks over the same column slices, the same fixed-order reduction -> dw [Cout][Cin] f32.
 * Replaces torch.bmm(g, x^T, out_dtype=f32).sum(0) (hipBLASLt) in the bf16 training step.  ws: sonet_wgrad_bf16_ws_size bytes.
 * Long reductions (L % 8 == 0, >= 2048 units of 64 columns in the batch) take the streaming generation: both operands through an
 * LDS-DMA ring, one workgroup per CU, (128 or 256) x 128 blocks of dw, partial blocks per column slice summed in a fixed order. */
size_t sonet_wgrad_bf16_ws_size(int B, int Cout, int Cin, int L);
int sonet_wgrad_bf16(const uint16_t *g, const uint16_t *x, float *dw, void *ws, int B, int Cout, int Cin, int L, sonet_stream_t stream);
/* ... when x is the RAW (bf16) output of a BatchNorm layer whose normalise pass was never run (bf16 training, normalise-on-load): the x
 * fragments go through act(raw * xs[c] + xh[c]) rounded to bf16, bit for bit what sonet_channel_affine_act_bf16 would have stored; xs, xh
 * [Cin].  Streaming-kernel shapes only (L % 8 == 0, B * ceil(L / 64) >= 2048): SONET_ERR_UNSUPPORTED otherwise. */
int sonet_wgrad_bf16_xaff(const uint16_t *g, const uint16_t *x, float *dw, void *ws, int B, int Cout, int Cin, int L,
                          const float *xs, const float *xh, int xrelu, sonet_stream_t stream);
/* out[b][c][l] = act((z[b][c][gidx[b][l]] + sum_{i<NL} wl[c][i] * lead[b][i][l]) * scale[c] + shift[c]);  z [B][C][M] = the
 * layer applied to the M node features once (sonet_pointmlp_h3_f32 with unit scale), gidx [B][L] i32 (out of range: 0),
 * lead [B][NL][L] the per-column channels (NL <= 4: the 3 de-centred coordinates), wl [C][NL] their weight columns, exact f32
 * fmas in channel order.  KNNModule layer 1 (models/layers.py:313-350: cat(de-centred coordinates, gathered features) -> 1x1
 * conv) without the K-fold redundant MFMAs over gathered columns. */
int sonet_node_gather_lead_affine_act_f32(const float *z, const int32_t *gidx, const float *lead, const float *wl,
                                          const float *scale, const float *shift, int relu, float *out,
                                          int B, int C, int L, int M, int NL, sonet_stream_t stream);

int sonet_node_gather_lead_affine_act_bf16(const uint16_t *z, const int32_t *gidx, const float *lead, const float *wl,
                                           const float *scale, const float *shift, int relu, uint16_t *out,
                                           int B, int C, int L, int M, int NL, sonet_stream_t stream);   /* z, out: bfloat16 bits */

/* Small-batch fully connected layer: y[b][o] = act((sum_k x[b][k] W[o][k]) * scale[o] + shift[o]); x [B][Cin], W [Cout][Cin]
 * (nn.Linear layout), exact f32 fma chain.  MyLinear = Linear + BatchNorm1d(eval) + ReLU (models/layers.py:123-166) with the
 * bias and the running statistics folded into (scale, shift): the classifier head of models/networks.py:202-227. */
int sonet_linear_act_f32(const float *x, const float *W, const float *scale, const float *shift, int relu, float *y,
                         int B, int Cin, int Cout, sonet_stream_t stream);

/* Sparse dgrad of the pooled last layer of the first PointNet (training, classifier / autoencoder): the gradient of
 * first_pn_out = W . [x1; x2] + b arrives only through the per-node max-pool (models/networks.py:180-185), i.e. as
 * g_pooled [B][C][M] at the arg-max positions pos [B][C][M] (i32 in [0, L); out-of-range entries are ignored).
 *   gx1 [B][C1][L], gx2 [B][C2][L]  <-  sum over entries (c, m) with pos == l of g_pooled[b][c][m] * W[c][:]
 * W [C][C1+C2] row-major f32.  Dense outputs (zero where nothing hits), C*M*(C1+C2) MACs per cloud instead of
 * C*L*(C1+C2).  ws: sonet_pooled_dgrad_ws_size bytes.  Deterministic (entries sorted per 64-column tile). */
size_t sonet_pooled_dgrad_ws_size(int B, int C, int M, int L);
/* The matching wgrad, with g_pooled and pos TRANSPOSED to [B][M][C] (coalesced entry walks):
 * gw_partial[b][c][ci] = sum_m g_pooled[b][m][c] * x[b][ci][pos[b][m][c]] (x [B][Ci][L]; positions
 * outside [0, L) are ignored); the caller sums the per-cloud partials over b.  Replaces scatter-into-zeros + a dense
 * [C x L] x [L x Ci] GEMM over a gradient that is zero everywhere except at the C*M gathered positions of a cloud. */
int sonet_pooled_wgrad_f32(const float *g_pooled, const int32_t *pos, const float *x, int B, int C, int M, int Ci, int L,
                           float *gw_partial, sonet_stream_t stream);
/* ... with normalise-on-load of x (see sonet_pointmlp_h3_stats_xaff_f32): x holds the RAW output of a BatchNorm layer, the rows are normalised
 * on their way into the LDS: x = act(raw * xs[ci] + xh[ci]); xs, xh [Ci], xrelu != 0: ReLU. */
int sonet_pooled_wgrad_xaff_f32(const float *g_pooled, const int32_t *pos, const float *x, int B, int C, int M, int Ci, int L,
                                float *gw_partial, const float *xs, const float *xh, int xrelu, sonet_stream_t stream);
int sonet_pooled_dgrad_f32(const float *g_pooled, const int32_t *pos, const float *W, int B, int C, int M, int C1, int C2,
                           int L, void *ws, float *gx1, float *gx2, sonet_stream_t stream);
/* bf16 training path: x read as bfloat16 bits (sonet_pooled_wgrad_xbf16), gradients written as bfloat16 bits (sonet_pooled_dgrad_obf16) */
int sonet_pooled_wgrad_xbf16(const float *g_pooled, const int32_t *pos, const uint16_t *x, int B, int C, int M, int Ci, int L,
                             float *gw_partial, sonet_stream_t stream);
/* ... with normalise-on-load of the bf16 rows (see sonet_pointmlp_bf16_stats_xaff): x = bf16(act(raw * xs[ci] + xh[ci])); xs, xh [Ci]. */
int sonet_pooled_wgrad_xaff_xbf16(const float *g_pooled, const int32_t *pos, const uint16_t *x, int B, int C, int M, int Ci, int L,
                                  float *gw_partial, const float *xs, const float *xh, int xrelu, sonet_stream_t stream);
int sonet_pooled_dgrad_obf16(const float *g_pooled, const int32_t *pos, const float *W, int B, int C, int M, int C1, int C2,
                             int L, void *ws, uint16_t *gx1, uint16_t *gx2, sonet_stream_t stream);
/* ... the same gradient as a dense product on the matrix cores: the 64-column tile of the (never built) gradient of first_pn_out is
 * assembled in LDS from the entries and multiplied by W^T.  wt_pack = sonet_pointmlp_bf16_pack of W^T ([C1 + C2] x C); g and W rounded
 * to bfloat16, f32 accumulate (the dense bf16 dgrad of the other layers does the same).  L even, C a multiple of 16 and <= 384, an even
 * number (<= 12) of 32-row tiles in C1 + C2; same workspace as above.  SONET_ERR_UNSUPPORTED otherwise (the caller takes _obf16). */
int sonet_pooled_dgrad_mfma_bf16(const float *g_pooled, const int32_t *pos, const void *wt_pack, int B, int C, int M, int C1, int C2,
                                 int L, void *ws, uint16_t *gx1, uint16_t *gx2, sonet_stream_t stream);

/* Per-channel coefficients of training BatchNorm, forward (invstd = 1/sqrt(var+eps), scale = gamma*invstd,
 * shift = beta - mean*scale) and backward (from the two sums of sonet_pointwise_bwd_stats_f32, n = B*L:
 *   sg = invstd*(s2 - mean*s1);  a = gamma*invstd;  b = -a*invstd*sg/n;  c0 = -a*s1/n - b*mean;
 *   g_gamma = sg;  g_beta = s1), all [C]. */
int sonet_bn_fwd_coeffs_f32(const float *mean, const float *var, const float *gamma, const float *beta, float eps, int C,
                            float *invstd, float *scale, float *shift, sonet_stream_t stream);
/* Running-statistics update of training BatchNorm (F.batch_norm semantics, models/layers.py:60-70):
 * running = running*(1 - momentum) + momentum*stat, the variance entering as var*unbias (unbias = n/(n-1)); in place, [C]. */
int sonet_bn_running_update_f32(float *running_mean, float *running_var, const float *mean, const float *var,
                                float momentum, float unbias, int C, sonet_stream_t stream);
/* "BatchNorm rider" of the training forward: registers, for the calling thread, what the NEXT statistics finalize -- the one inside
 * sonet_pointmlp_*_stats_*, sonet_pointmlp_h3p (statistics epilogue) or sonet_channel_stats_* -- also computes per channel from the batch
 * statistics it produces: (invstd, scale, shift) as sonet_bn_fwd_coeffs_f32 and, when running_mean / running_var are given, their update
 * as sonet_bn_running_update_f32 -- the same arithmetic in the same order, one launch instead of three per BatchNorm layer and step
 * (models/layers.py:60-70).  Consumed (and cleared) by that launch; gamma == NULL clears it. */
int sonet_bn_rider_set(const float *gamma, const float *beta, float eps, float momentum, float unbias,
                       float *running_mean, float *running_var, float *invstd, float *scale, float *shift);
int sonet_bn_bwd_coeffs_f32(const double *sums, const float *mean, const float *invstd, const float *gamma, double n, int C,
                            float *a, float *b, float *c0, float *g_gamma, float *g_beta, sonet_stream_t stream);
/* y = act(x*scale[c] + shift[c]) out of place (training forward: raw stays for the backward). */
/* Mean over the k copies of a point (segmenter head, models/networks.py:331-336): out[r][n] = c * ((h[r][n] + h[r][N + n]) + h[r][2N + n]),
 * h [rows][k * N] f32, c = 1/3 (k = 3) / 0.5 (k = 2), k = 1: copy; the reference's order of operations (bit-identical to split + add + mul). */
int sonet_chunk_mean_f32(const float *h, float *out, long long rows, int N, int k, sonet_stream_t stream);
int sonet_channel_affine_act_out_f32(const float *x, const float *scale, const float *shift, int relu, float *y,
                                     int B, int C, int L, sonet_stream_t stream);

/* Per-channel batch statistics of y [B][C][L] for training-mode BatchNorm (F.batch_norm with
 * training=True, models/layers.py:68): mean[c], biased var[c] over (B, L), f64 accumulation.
 * stat_ws: 2*C doubles of workspace, zeroed by the callee. */
int sonet_channel_stats_f32(const float *y, int B, int C, int L, double *stat_ws,
                            float *mean, float *var_biased, sonet_stream_t stream);
/* y = act(y * scale[c] + shift[c]) in place (the normalise + ReLU pass of training mode). */
int sonet_channel_affine_act_f32(float *y, const float *scale, const float *shift, int relu,
                                 int B, int C, int L, sonet_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * chamfer_nn ("next" row) -- replaces the faiss GpuIndexFlatL2 1-NN search
 *   reference: models/losses.py:220-235 (search), :260-276 (per-sample loop)
 * q [B][3][Nq], db [B][3][Nd] f32 -> nn [B][Nq] i32 = argmin_j ((dx*dx+dy*dy)+dz*dz), ties -> lowest j.
 * ---------------------------------------------------------------------------------------------- */
int sonet_chamfer_nn_f32(const float *q, const float *db, int32_t *nn, int B, int Nq, int Nd,
                         sonet_stream_t stream);
/* ------------------------------------------------------------------------------------------------
 * VARIANTS build only (make -C so-net_amd/csrc variants -> libsonet_hip_variants.so, -DSONET_VARIANTS): kernels that measured
 * slower than what the product dispatches, kept as tested records of the experiments (tests/variants).  The product library
 * does not export them and reads no environment variable.
 * ---------------------------------------------------------------------------------------------- */
#ifdef SONET_VARIANTS
/* out[b][o][m] = max over the columns l with l % M == m of act((W . cat(x1, x2)) * scale + shift)[b][o][l]: a layer followed by the
 * max over the K neighbour planes of a k-major B x Cout x (K * M) tensor -- KNNModule's last layer + torch.max(dim=3)
 * (models/layers.py:340-350) -- without materialising that tensor (ordered-integer atomic max from the layer kernel's epilogue).
 * keys_ws: B * Cout * M * 4 bytes; out [B][Cout][M] f32; L % M == 0; Cout % 128 == 0; fp16-split arithmetic (range-guarded). */
int sonet_pointmlp_h3_kmax_f32(const float *x1, int C1, const float *x2, int C2, const void *Wp3, const float *scale,
                               const float *shift, int relu, float *out, void *keys_ws, int B, int Cout, int L, int M,
                               sonet_stream_t stream);

/* Both directions of models/losses.py:255,262 from ONE sweep of the Na x Nb distance matrix: nn_ab[b][i] = nearest point of
 * cloud b for a_i, nn_ba[b][j] = nearest point of cloud a for b_j (same arithmetic, ties -> lowest index: bit-identical to two
 * sonet_chamfer_nn_f32 calls).  ws: sonet_chamfer_nn2_ws_size bytes (64-bit (distance, index) keys of the column minima). */
size_t sonet_chamfer_nn2_ws_size(int B, int Na, int Nb);
int sonet_chamfer_nn2_f32(const float *a, const float *b, int32_t *nn_ab, int32_t *nn_ba, void *ws, int B, int Na, int Nb,
                          sonet_stream_t stream);

/* sonet_pooled_dgrad_f32 with what used to follow the launch riding on its store -- measured slower than the launches it replaces
 * (docs/findings.md R5.14) -- (C1 + C2 a multiple of 4; node-sorted f32-class training path):
 *  col0 [B][C1 + C2], pos0 [B] (both or neither): gx[b][:, pos0[b]] += col0[b] -- every channel of an EMPTY node gathers position 0
 *    (models/networks.py:185): their entries are one dense mat-vec per cloud (the caller's) landing on one column;
 *  sraw [B][C2][L], ssc, ssh [C2], srelu, tail_ws (sonet_pooled_dgrad_tail_ws_size bytes), sums [2 C2] (all or none; C2 > 0): gx2 is gy of the
 *    BatchNorm (+ ReLU) layer (models/layers.py:60-70, :282-296) whose RAW output is sraw; sums[0 .. C2) = sum over (b, l) of gy * mask,
 *    sums[C2 .. 2 C2) = sum of gy * mask * raw with mask = !srelu || raw * ssc + ssh > 0: what sonet_pointwise_bwd_stats_f32 computes from one
 *    more pass over (gy, raw); double precision, fixed order. */
size_t sonet_pooled_dgrad_tail_ws_size(int B, int C2, int L);
int sonet_pooled_dgrad_tail_f32(const float *g_pooled, const int32_t *pos, const float *W, int B, int C, int M, int C1, int C2,
                                int L, void *ws, float *gx1, float *gx2, const float *col0, const int32_t *pos0,
                                const float *sraw, const float *ssc, const float *ssh, int srelu, void *tail_ws, double *sums,
                                sonet_stream_t stream);
#endif /* SONET_VARIANTS */

/* Packs of a matrix given by element strides: element (o, c) = W[o * row_stride + c * col_stride] for o < rows, c < Cin; rows in
 * [rows, Cout) pack as zeros (callers pad Cout to a friendly tile count).  With (row_stride, col_stride) = (1, ld) and W advanced by a
 * column offset: the pack of a column block of W TRANSPOSED -- the dgrad's weights (W^T g of models/layers.py:282-296's conv) -- read
 * along W's own rows, without a transposed copy.  Same pack sizes / layouts as sonet_pointmlp_{x3,h3,bf16}_pack(Cin, Cout). */
int sonet_pointmlp_x3_pack_strided(const float *W, long long row_stride, long long col_stride, void *Wp3, int Cin, int Cout, int rows,
                                   sonet_stream_t stream);
int sonet_pointmlp_h3_pack_strided(const float *W, long long row_stride, long long col_stride, void *Wp3, int Cin, int Cout, int rows,
                                   sonet_stream_t stream);
/* Refresh many weight packs in one launch (+ one that clears the max-|w| trailers): the optimizer changes every weight once per training step
 * (models/classifier.py:93-99), and a launch per layer and pack flavour was fifteen launches and fifteen 64-byte memsets per step.
 * table: n_entries records of 72 bytes on the device,
 *   { const float *W; void *Wp; int64 rs, cs, total; int32 Cin, rows, KC, flavour, blk0, nblk; int64 pad }
 * -- entry i packs the matrix with element (o, c) = W[o rs + c cs], o < rows, c < Cin, into Wp exactly as sonet_pointmlp_bf16_pack_strided
 * (flavour 0), sonet_pointmlp_x3_pack_strided (1) or sonet_pointmlp_h3_pack_strided (2) would; KC = sonet_pack_multi_kc(flavour, Cin),
 * total = 64 x ceil(Cout_pack / 32) x KC, the entry owns workgroups blk0 .. blk0 + nblk - 1, nblk = ceil(total / 256), blk0 ascending;
 * total_blocks = their sum. */
int sonet_pack_multi(const void *table, int n_entries, int total_blocks, sonet_stream_t stream);
int sonet_pack_multi_kc(int flavour, int Cin);
int sonet_pointmlp_bf16_pack_strided(const float *W, long long row_stride, long long col_stride, void *Wp, int Cin, int Cout, int rows,
                                     sonet_stream_t stream);

/* torch.optim.Adam's update (models/classifier.py:45-49; amsgrad = False, weight_decay = 0) for all f32 parameters of an optimizer in
 * ONE launch.  tensors: device table, 32 bytes per tensor: {float *param; const float *grad; float *exp_avg; float *exp_avg_sq}
 * (grad NULL: skipped, the tensor's step count does not advance); step_size[t] = lr / (1 - beta1^step_t), bc2_sqrt[t] = sqrt(1 -
 * beta2^step_t); chunk table: chunk j covers sonet_adam_chunk() elements of tensor chunk_tensor[j] from chunk_off[j]; sizes[t] =
 * element counts; one_minus_beta1/2 = 1 - beta rounded from double (as torch's Python floats are).
 * m <- lerp(m, g, 1 - beta1); v <- v beta2 + (1 - beta2) g g; p <- p - step_size m / (sqrt(v) / bc2_sqrt + eps). */
int sonet_adam_chunk(void);
int sonet_adam_multi_f32(const void *tensors, const float *step_size, const float *bc2_sqrt, const int32_t *chunk_tensor,
                         const long long *chunk_off, const long long *sizes, int nchunks, float beta1, float beta2,
                         float one_minus_beta1, float one_minus_beta2, float eps, sonet_stream_t stream);

/* The B x C fully connected layers of the heads in TRAINING (models/layers.py:123-166: Linear + BatchNorm1d + ReLU, stacked by
 * models/networks.py:202-227): one forward launch, two backward launches per layer instead of aten's addmm / batch_norm / relu /
 * threshold_backward / batch_norm_backward / mm / mm / sum.  f32, B <= sonet_fc_max_rows() rows, Cin % 4 == 0, Cout % 4 == 0.
 * forward: y = act(BN(x W^T + bias)); gamma == NULL: no normalisation; gamma != NULL: batch statistics (B >= 2, biased variance for the
 * output), running_mean / running_var (either may be NULL) updated in place with `momentum` (unbiased variance), xhat [B][Cout] and
 * invstd [Cout] kept for the backward.
 * backward: gy -> dz [B][Cout] (ReLU mask from y, BatchNorm backward from xhat / invstd / gamma), dW [Cout][Cin] = dz^T x, dbias = column
 * sums of dz (dW, dbias may be NULL), dgamma, dbeta;  sonet_fc_dx_f32: dx [B][Cin] = dz W. */
int sonet_fc_max_rows(void);
int sonet_fc_bn_act_fwd_f32(const float *x, const float *W, const float *bias, const float *gamma, const float *beta,
                            float *running_mean, float *running_var, float momentum, float eps, int relu, int B, int Cin, int Cout,
                            float *y, float *xhat, float *invstd, sonet_stream_t stream);
int sonet_fc_bn_act_bwd_f32(const float *gy, const float *y, const float *xhat, const float *invstd, const float *gamma, const float *x,
                            int relu, int B, int Cin, int Cout, float *dz, float *dW, float *dbias, float *dgamma, float *dbeta,
                            sonet_stream_t stream);
int sonet_fc_dx_f32(const float *dz, const float *W, int B, int Cin, int Cout, float *dx, sonet_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* SONET_HIP_H */
