/*
 * cabi_hotpath.c -- TEST: the drop-in boundary from a host that knows nothing about torch or Python.
 *
 * Plain C99, compiled with gcc against include/sonet_hip.h and the HIP runtime's C API only.  It drives the hot
 * path exactly as a C / cgo / JNI binding of the reference would (level 0):
 *     device buffers  ->  sonet_som_assign_f32  ->  sonet_som_group_f32  ->  sonet_index_max_gather_f32
 * on a caller-created HIP stream, and checks every output against the CPU oracle (oracle/_build/libsonet_oracle.so,
 * test infrastructure -- this file is a test, not product code): node ids, counts and arg-max positions bit-exact,
 * float outputs within 1e-5 * max(|ref|, rms).  Also checks the error convention: a non-zero status plus a message from
 * sonet_last_error() for a NULL pointer, and that nothing is synchronised or allocated behind the caller's back
 * (all outputs are caller-allocated, all launches asynchronous on `stream`).
 *
 *   usage: cabi_hotpath [B N M k C]      exit code 0 = all checks passed
 */
#define __HIP_PLATFORM_AMD__ 1
#include <hip/hip_runtime_api.h>
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "sonet_hip.h"

/* the checker (oracle/sonet_oracle.c) */
void oracle_index_max_f32(const float *data, const int32_t *index, int32_t *out, int B, int C, int N, int K);
void oracle_som_query_topk_f32(const float *x, const float *node, int B, int N, int M, int k,
                               int64_t *min_idx, int32_t *count, int32_t *row_max);
void oracle_som_group_f32(const float *x, const int64_t *min_idx, int B, int N, int M, int k,
                          float *som_node, float *centers, float *x_decentered);

#define HIP_OK(call) do { hipError_t e_ = (call); if (e_ != hipSuccess) { \
    fprintf(stderr, "%s:%d: %s -> %s\n", __FILE__, __LINE__, #call, hipGetErrorString(e_)); return 2; } } while (0)
#define SONET_CALL(call) do { int s_ = (call); if (s_ != SONET_OK) { \
    fprintf(stderr, "%s:%d: %s -> status %d: %s\n", __FILE__, __LINE__, #call, s_, sonet_last_error()); return 3; } } while (0)

static uint64_t rng_state = 0x9E3779B97F4A7C15ull;
static float uniform_pm1(void) {                       /* xorshift64*: deterministic inputs, no libc rand */
    rng_state ^= rng_state >> 12; rng_state ^= rng_state << 25; rng_state ^= rng_state >> 27;
    return (float)((double)((rng_state * 0x2545F4914F6CDD1Dull) >> 40) / (double)(1 << 24)) * 2.0f - 1.0f;
}

static int check_close(const char *what, const float *got, const float *ref, size_t n) {
    double ss = 0.0;
    for (size_t i = 0; i < n; ++i) ss += (double)ref[i] * ref[i];
    const double rms = n ? sqrt(ss / (double)n) : 0.0;
    size_t bad = 0;
    for (size_t i = 0; i < n; ++i) {
        const double bound = 1e-5 * fmax(fabs((double)ref[i]), rms);
        if (!(fabs((double)got[i] - (double)ref[i]) <= bound)) ++bad;
    }
    if (bad) fprintf(stderr, "%s: %zu of %zu elements outside 1e-5 * max(|ref|, rms)\n", what, bad, n);
    return bad == 0;
}

static int check_equal_i32(const char *what, const int32_t *got, const int32_t *ref, size_t n) {
    size_t bad = 0;
    for (size_t i = 0; i < n; ++i) bad += got[i] != ref[i];
    if (bad) fprintf(stderr, "%s: %zu of %zu integers differ\n", what, bad, n);
    return bad == 0;
}

int main(int argc, char **argv) {
    const int B = argc > 1 ? atoi(argv[1]) : 3, N = argc > 2 ? atoi(argv[2]) : 1000, M = argc > 3 ? atoi(argv[3]) : 64;
    const int k = argc > 4 ? atoi(argv[4]) : 3, C = argc > 5 ? atoi(argv[5]) : 40;
    const size_t kN = (size_t)k * N;

    if (sonet_abi_version() != 1 || strcmp(sonet_build_arch(), "gfx950") != 0) { fprintf(stderr, "library identity\n"); return 1; }
    SONET_CALL(sonet_check_device());

    /* ---- host inputs ---- */
    float *x = malloc(sizeof(float) * B * 3 * N), *node = malloc(sizeof(float) * B * 3 * M), *data = malloc(sizeof(float) * B * C * kN);
    for (size_t i = 0; i < (size_t)B * 3 * N; ++i) x[i] = uniform_pm1();
    for (size_t i = 0; i < (size_t)B * 3 * M; ++i) node[i] = uniform_pm1();
    for (size_t i = 0; i < (size_t)B * C * kN; ++i) data[i] = uniform_pm1() * 3.0f;

    /* ---- device buffers: all caller-owned ---- */
    hipStream_t stream;
    HIP_OK(hipStreamCreate(&stream));
    float *d_x, *d_node, *d_data, *d_som_node, *d_centers, *d_xdec, *d_val;
    int32_t *d_min32, *d_count, *d_row_max, *d_idx;
    int64_t *d_min64;
    double *d_sum;
    HIP_OK(hipMalloc((void **)&d_x, sizeof(float) * B * 3 * N));
    HIP_OK(hipMalloc((void **)&d_node, sizeof(float) * B * 3 * M));
    HIP_OK(hipMalloc((void **)&d_data, sizeof(float) * B * C * kN));
    HIP_OK(hipMalloc((void **)&d_min32, sizeof(int32_t) * B * kN));
    HIP_OK(hipMalloc((void **)&d_min64, sizeof(int64_t) * B * kN));
    HIP_OK(hipMalloc((void **)&d_count, sizeof(int32_t) * B * M));
    HIP_OK(hipMalloc((void **)&d_sum, sizeof(double) * B * 3 * M));
    HIP_OK(hipMalloc((void **)&d_som_node, sizeof(float) * B * 3 * M));
    HIP_OK(hipMalloc((void **)&d_row_max, sizeof(int32_t) * B * M));
    HIP_OK(hipMalloc((void **)&d_centers, sizeof(float) * B * 3 * kN));
    HIP_OK(hipMalloc((void **)&d_xdec, sizeof(float) * B * 3 * kN));
    HIP_OK(hipMalloc((void **)&d_idx, sizeof(int32_t) * B * C * M));
    HIP_OK(hipMalloc((void **)&d_val, sizeof(float) * B * C * M));
    HIP_OK(hipMemcpyAsync(d_x, x, sizeof(float) * B * 3 * N, hipMemcpyHostToDevice, stream));
    HIP_OK(hipMemcpyAsync(d_node, node, sizeof(float) * B * 3 * M, hipMemcpyHostToDevice, stream));
    HIP_OK(hipMemcpyAsync(d_data, data, sizeof(float) * B * C * kN, hipMemcpyHostToDevice, stream));

    /* ---- the hot path through the C ABI, back to back on one stream, no synchronisation in between ---- */
    SONET_CALL(sonet_som_assign_f32(d_x, d_node, B, N, M, k, d_min32, d_min64, d_count, d_sum, stream));
    SONET_CALL(sonet_som_group_f32(d_x, NULL, d_min32, d_count, d_sum, B, N, M, k, d_som_node, d_row_max, d_centers, d_xdec, NULL, stream));
    SONET_CALL(sonet_index_max_gather_f32(d_data, d_min32, d_row_max, d_idx, d_val, B, C, (int)kN, M, stream));

    int32_t *min32 = malloc(sizeof(int32_t) * B * kN), *count = malloc(sizeof(int32_t) * B * M), *row_max = malloc(sizeof(int32_t) * B * M);
    int32_t *idx = malloc(sizeof(int32_t) * B * C * M);
    int64_t *min64 = malloc(sizeof(int64_t) * B * kN);
    float *som_node = malloc(sizeof(float) * B * 3 * M), *centers = malloc(sizeof(float) * B * 3 * kN), *xdec = malloc(sizeof(float) * B * 3 * kN);
    float *val = malloc(sizeof(float) * B * C * M);
    HIP_OK(hipMemcpyAsync(min32, d_min32, sizeof(int32_t) * B * kN, hipMemcpyDeviceToHost, stream));
    HIP_OK(hipMemcpyAsync(min64, d_min64, sizeof(int64_t) * B * kN, hipMemcpyDeviceToHost, stream));
    HIP_OK(hipMemcpyAsync(count, d_count, sizeof(int32_t) * B * M, hipMemcpyDeviceToHost, stream));
    HIP_OK(hipMemcpyAsync(row_max, d_row_max, sizeof(int32_t) * B * M, hipMemcpyDeviceToHost, stream));
    HIP_OK(hipMemcpyAsync(som_node, d_som_node, sizeof(float) * B * 3 * M, hipMemcpyDeviceToHost, stream));
    HIP_OK(hipMemcpyAsync(centers, d_centers, sizeof(float) * B * 3 * kN, hipMemcpyDeviceToHost, stream));
    HIP_OK(hipMemcpyAsync(xdec, d_xdec, sizeof(float) * B * 3 * kN, hipMemcpyDeviceToHost, stream));
    HIP_OK(hipMemcpyAsync(idx, d_idx, sizeof(int32_t) * B * C * M, hipMemcpyDeviceToHost, stream));
    HIP_OK(hipMemcpyAsync(val, d_val, sizeof(float) * B * C * M, hipMemcpyDeviceToHost, stream));
    HIP_OK(hipStreamSynchronize(stream));

    /* ---- the checker ---- */
    int64_t *r_min = malloc(sizeof(int64_t) * B * kN);
    int32_t *r_count = malloc(sizeof(int32_t) * B * M), *r_row_max = malloc(sizeof(int32_t) * B * M), *r_idx = malloc(sizeof(int32_t) * B * C * M);
    int32_t *r_min32 = malloc(sizeof(int32_t) * B * kN);
    float *r_node = malloc(sizeof(float) * B * 3 * M), *r_centers = malloc(sizeof(float) * B * 3 * kN), *r_xdec = malloc(sizeof(float) * B * 3 * kN);
    float *r_val = malloc(sizeof(float) * B * C * M);
    oracle_som_query_topk_f32(x, node, B, N, M, k, r_min, r_count, r_row_max);
    oracle_som_group_f32(x, r_min, B, N, M, k, r_node, r_centers, r_xdec);
    for (size_t i = 0; i < (size_t)B * kN; ++i) r_min32[i] = (int32_t)r_min[i];
    oracle_index_max_f32(data, r_min32, r_idx, B, C, (int)kN, M);
    for (int b = 0; b < B; ++b)                              /* models/networks.py:185: gather at index * mask_row_max */
        for (int c = 0; c < C; ++c)
            for (int m = 0; m < M; ++m) {
                const size_t o = ((size_t)b * C + c) * M + m;
                r_val[o] = data[((size_t)b * C + c) * kN + (size_t)r_idx[o] * (size_t)r_row_max[(size_t)b * M + m]];
            }

    int ok = 1;
    size_t bad64 = 0;
    for (size_t i = 0; i < (size_t)B * kN; ++i) bad64 += min64[i] != r_min[i];
    if (bad64) { fprintf(stderr, "min_idx_i64: %zu differ\n", bad64); ok = 0; }
    ok &= check_equal_i32("min_idx_i32", min32, r_min32, (size_t)B * kN);
    ok &= check_equal_i32("count", count, r_count, (size_t)B * M);
    ok &= check_equal_i32("row_max", row_max, r_row_max, (size_t)B * M);
    ok &= check_equal_i32("index_max positions", idx, r_idx, (size_t)B * C * M);
    ok &= check_close("som_node", som_node, r_node, (size_t)B * 3 * M);
    ok &= check_close("centers", centers, r_centers, (size_t)B * 3 * kN);
    ok &= check_close("x_decentered", xdec, r_xdec, (size_t)B * 3 * kN);
    size_t badv = 0;
    for (size_t i = 0; i < (size_t)B * C * M; ++i) badv += memcmp(&val[i], &r_val[i], sizeof(float)) != 0;
    if (badv) { fprintf(stderr, "gathered values: %zu differ\n", badv); ok = 0; }

    /* ---- error convention: status code + message, no abort ---- */
    const int st = sonet_index_max_f32(NULL, d_min32, d_idx, B, C, (int)kN, M, stream);
    if (st == SONET_OK || strlen(sonet_last_error()) == 0) { fprintf(stderr, "NULL data pointer was accepted\n"); ok = 0; }
    const int st2 = sonet_index_max_f32(d_data, d_min32, d_idx, B, C, (int)kN, 4096, stream);
    if (st2 != SONET_ERR_UNSUPPORTED) { fprintf(stderr, "K = 4096 should be SONET_ERR_UNSUPPORTED, got %d\n", st2); ok = 0; }

    HIP_OK(hipStreamDestroy(stream));
    printf("cabi_hotpath B=%d N=%d M=%d k=%d C=%d: %s\n", B, N, M, k, C, ok ? "OK" : "FAILED");
    return ok ? 0 : 1;
}
