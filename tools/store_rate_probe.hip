// Diagnostic (GPU box): what rate can 16-byte-per-lane stores alone reach?  k_learn_chain writes 0.85 GB per minibatch
// (tiles of 96 rows x 512 B from 4,096 workgroups); this kernel writes the same bytes in the same shape and nothing else.
//   hipcc --offload-arch=gfx950 -O3 -o tools/store_rate_probe tools/store_rate_probe.hip ; ./tools/store_rate_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;
template <bool NT, bool LOAD>
__global__ __launch_bounds__(256) void k_tiles(u32x4 *out, int arrays, size_t rows, const u32x4 *in) {
    // one workgroup per 96-row tile: `arrays` arrays of [rows][32] x 16 B, as H1 / H2 / dH2 / dH1
    size_t row0 = (size_t)blockIdx.x * 96;
    unsigned tid = threadIdx.x;
    u32x4 v = {tid, blockIdx.x, 3u, 4u};
    for (int a = 0; a < arrays; a++) {
        u32x4 *base = out + (size_t)a * rows * 32;
#pragma unroll
        for (int it = 0; it < 12; it++) {
            size_t idx = (row0 + it * 8 + (tid >> 5)) * 32 + (tid & 31);
            if (LOAD) { u32x4 t = NT ? __builtin_nontemporal_load(&in[(size_t)a * rows * 32 + idx]) : in[(size_t)a * rows * 32 + idx]; v += t; }
            else if (NT) __builtin_nontemporal_store(v, &base[idx]); else base[idx] = v;
        }
    }
    if (LOAD && v.x == 0x12345u) out[0] = v;
}
int main() {
    const size_t rows = 393216, tiles = rows / 96;
    const int arrays = 4;
    u32x4 *buf, *src;
    size_t bytes = (size_t)arrays * rows * 512;
    hipMalloc(&buf, bytes); hipMalloc(&src, bytes);
    hipMemset(src, 1, bytes);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    auto run = [&](const char *name, auto kernel) {
        for (int w = 0; w < 3; w++) hipLaunchKernelGGL(kernel, dim3(tiles), dim3(256), 0, 0, buf, arrays, rows, src);
        hipEventRecord(e0);
        for (int r = 0; r < 20; r++) hipLaunchKernelGGL(kernel, dim3(tiles), dim3(256), 0, 0, buf, arrays, rows, src);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        printf("%-28s %7.1f us per launch  %6.2f TB/s (%.2f GB)\n", name, ms * 1e3 / 20, bytes / (ms / 20 * 1e-3) / 1e12, bytes / 1e9);
    };
    run("stores, plain", k_tiles<false, false>);
    run("stores, non-temporal", k_tiles<true, false>);
    run("loads, plain", k_tiles<false, true>);
    run("loads, non-temporal", k_tiles<true, true>);
    return 0;
}
