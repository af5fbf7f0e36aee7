// What does an output store cost the wave that issues it?  (gfx950, one wave per SIMD: 256 workgroups x 256 threads.)
// Each loop iteration = 24 plain vector instructions + (mode) one store of a fresh row, the shape of k_play's per-card
// outputs (one byte / one 8-byte word per game and card, row after row); cycles per iteration by s_memtime,
// median over the waves.  The difference to mode 0 is what the store (and its address arithmetic) costs.
//   mode 0  no store
//   mode 1  byte store, non-temporal, 64-bit per-lane pointer (advanced by v_lshl_add_u64: what the compiler emits
//           for a row index that is a 64-bit number)
//   mode 2  8-byte store, the same
//   mode 3  byte store, non-temporal, uniform base pointer + 32-bit per-lane offset (the saddr form)
//   mode 4  8-byte store, the same
//   mode 5  modes 1 + 1 + 2 together (a card's three outputs), 64-bit pointers
//   mode 6  the same three stores in the saddr form
// build: hipcc --offload-arch=gfx950 -O3 -o tools/store_issue tools/store_issue.hip     run: tools/store_issue
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
typedef unsigned long long u64;
typedef uint32_t u32;
#define CHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(2); } } while (0)
#define FILL8 "v_and_b32 %0, %0, %1\nv_xor_b32 %0, %0, %1\nv_add_u32 %0, %0, %1\nv_and_b32 %0, %0, %1\nv_xor_b32 %0, %0, %1\nv_add_u32 %0, %0, %1\nv_and_b32 %0, %0, %1\nv_xor_b32 %0, %0, %1\n"

template <int MODE>
__global__ __launch_bounds__(256) void k_probe(u64 *cycles, uint8_t *b1, uint8_t *b2, u64 *b8, int iters, int64_t n, u32 seed) {
    u32 x = seed + threadIdx.x, y = seed | 5u;
    int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    int64_t row = i;                         // 64-bit row index, advanced by n per iteration
    u32 off = (u32)i, off8 = (u32)i * 8u;    // 32-bit (byte) offsets for the saddr form
    u64 t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; it++) {
        asm volatile(FILL8 FILL8 FILL8 : "+v"(x) : "v"(y));
        if (MODE == 1 || MODE == 5) __builtin_nontemporal_store((uint8_t)x, &b1[row]);
        if (MODE == 5) __builtin_nontemporal_store((uint8_t)(x >> 8), &b2[row]);
        if (MODE == 2 || MODE == 5) __builtin_nontemporal_store(((u64)x << 32) | y, &b8[row]);
        if (MODE == 3 || MODE == 6) __builtin_nontemporal_store((uint8_t)x, &b1[off]);
        if (MODE == 6) __builtin_nontemporal_store((uint8_t)(x >> 8), &b2[off]);
        if (MODE == 4 || MODE == 6) __builtin_nontemporal_store(((u64)x << 32) | y, (u64 *)((char *)b8 + off8));
        row += n; off += (u32)n; off8 += (u32)n * 8u;
    }
    u64 t1 = __builtin_amdgcn_s_memtime();
    if (x == 0x12345678u) b1[0] = 1;
    if ((threadIdx.x & 63) == 0) cycles[i >> 6] = t1 - t0;
}

template <int MODE> static double run(u64 *d_c, uint8_t *b1, uint8_t *b2, u64 *b8, int iters, int64_t n) {
    for (int rep = 0; rep < 2; rep++) hipLaunchKernelGGL(k_probe<MODE>, dim3((unsigned)(n / 256)), dim3(256), 0, 0, d_c, b1, b2, b8, iters, n, 12345u);
    CHK(hipDeviceSynchronize());
    std::vector<u64> h(n / 64);
    CHK(hipMemcpy(h.data(), d_c, h.size() * sizeof(u64), hipMemcpyDeviceToHost));
    std::sort(h.begin(), h.end());
    return (double)h[h.size() / 2] / iters;
}

int main() {
    const int64_t n = 65536; const int iters = 512;
    u64 *d_c; uint8_t *b1, *b2; u64 *b8;
    CHK(hipMalloc((void **)&d_c, n / 64 * sizeof(u64)));
    CHK(hipMalloc((void **)&b1, (size_t)n * iters)); CHK(hipMalloc((void **)&b2, (size_t)n * iters));
    CHK(hipMalloc((void **)&b8, (size_t)n * iters * 8));
    double m[7] = {run<0>(d_c, b1, b2, b8, iters, n), run<1>(d_c, b1, b2, b8, iters, n), run<2>(d_c, b1, b2, b8, iters, n), run<3>(d_c, b1, b2, b8, iters, n),
                   run<4>(d_c, b1, b2, b8, iters, n), run<5>(d_c, b1, b2, b8, iters, n), run<6>(d_c, b1, b2, b8, iters, n)};
    const char *name[7] = {"24 vector instructions, no store", "+ byte store, 64-bit pointer", "+ 8-byte store, 64-bit pointer", "+ byte store, base + 32-bit offset",
                           "+ 8-byte store, base + 32-bit offset", "+ a card's three outputs, 64-bit pointers", "+ a card's three outputs, base + 32-bit offsets"};
    printf("{\n \"unit\": \"shader cycles per loop iteration, one wave per SIMD, median over 1,024 waves\",\n");
    for (int k = 0; k < 7; k++) printf(" \"%s\": {\"cycles\": %.1f, \"over_no_store\": %.1f}%s\n", name[k], m[k], m[k] - m[0], k < 6 ? "," : "");
    printf("}\n");
    return 0;
}
