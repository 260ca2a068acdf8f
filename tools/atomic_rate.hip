// tools/atomic_rate.hip -- device-scope atomicMax(u32) throughput on a small table (pooled[B][384][64] = 6 MB at B=64)
#include <hip/hip_runtime.h>
#include <stdio.h>
__global__ void k(unsigned *tab, unsigned n_entries, int per_thread, unsigned seed) {
    unsigned x = (blockIdx.x * blockDim.x + threadIdx.x) * 2654435761u + seed;
    for (int i = 0; i < per_thread; ++i) {
        x = x * 1664525u + 1013904223u;
        atomicMax(&tab[(x >> 8) % n_entries], x);
    }
}
__global__ void kseq(unsigned *tab, unsigned n_entries, int per_thread) {      // coalesced: consecutive lanes -> consecutive entries
    unsigned base = (blockIdx.x * blockDim.x + threadIdx.x);
    for (int i = 0; i < per_thread; ++i) atomicMax(&tab[(base + i * 977u * 64u) % n_entries], base + i);
}
int main() {
    unsigned n = 64 * 384 * 64; unsigned *tab; hipMalloc(&tab, n * 4); hipMemset(tab, 0, n * 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int mode = 0; mode < 2; ++mode) for (int blocks : {1024, 8192}) {
        const int per = 64; float ms;
        if (mode == 0) k<<<blocks, 256>>>(tab, n, per, 1); else kseq<<<blocks, 256>>>(tab, n, per);
        hipDeviceSynchronize();
        hipEventRecord(e0);
        if (mode == 0) k<<<blocks, 256>>>(tab, n, per, 7); else kseq<<<blocks, 256>>>(tab, n, per);
        hipEventRecord(e1); hipEventSynchronize(e1); hipEventElapsedTime(&ms, e0, e1);
        double total = (double)blocks * 256 * per;
        printf("%s blocks=%d : %.1f M atomics in %.3f ms = %.1f G atomics/s\n", mode ? "coalesced" : "random   ", blocks, total / 1e6, ms, total / ms / 1e6);
    }
    return 0;
}
