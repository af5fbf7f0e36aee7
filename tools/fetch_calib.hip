// Calibration of rocprofv3's FETCH_SIZE / WRITE_SIZE on gfx950 for the access shapes of the step kernels
// (MI355X_MICROARCH.md: FETCH_SIZE reads HALF the bytes of a wide coalesced read stream; other shapes are uncalibrated).
// Each kernel touches a known number of bytes of a 1 GiB buffer (far beyond the 256 MB Infinity Cache), once:
//   k_stream16   every lane reads 16 B, consecutive lanes consecutive addresses (the state arrays)        n x 16 B
//   k_stream8    every lane reads 8 B (RNG keys, observation words)                                      n x 8 B
//   k_stream1    every lane reads 1 B (the action array)                                                 n x 1 B
//   k_sparse64   one lane in ten reads a 64-byte record (4 x 16 B) at its slot of an 896-byte-strided array: the
//                next-game line of a lane whose game ended                                               n/10 x 64 B
//   k_sparse32   one lane in ten reads 32 B (2 x 16 B) of a 32-byte-strided array: the slot's Counters    n/10 x 32 B
//   k_wsparse32  one lane in ten writes 16 + 4 B into its 32-byte record (score sums, episode)            n/10 x 20 B (one 32-B sector)
//   k_wstream16  every lane writes 16 B                                                                   n x 16 B
// Run under rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE (separate passes); tools/fetch_calib_summary.py compares the
// counters (KB per dispatch) with the bytes above.
// build: hipcc --offload-arch=gfx950 -O3 -o tools/fetch_calib tools/fetch_calib.hip     run: tools/fetch_calib
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
typedef unsigned long long u64;
typedef uint32_t u32;
#define CHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(2); } } while (0)

__global__ void k_stream16(const uint4 *p, int64_t n, u32 *sink) { int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; if (i < n) { uint4 v = p[i]; if (v.x == 0x12345678u && v.y == 1) sink[0] = v.z; } }
__global__ void k_stream8(const u64 *p, int64_t n, u32 *sink) { int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; if (i < n) { u64 v = p[i]; if (v == 0x123456789ULL) sink[0] = 1; } }
__global__ void k_stream1(const uint8_t *p, int64_t n, u32 *sink) { int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; if (i < n) { u32 v = p[i]; if (v == 77 && i == 3) sink[0] = 1; } }
__device__ __forceinline__ bool chosen(int64_t i) { return ((u32)i * 2654435761u >> 7) % 10u == 0; }
__global__ void k_sparse64(const uint4 *p, int64_t n, u32 *sink) {
    int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n && chosen(i)) { const uint4 *q = p + i * 56 + 4 * ((u32)i % 14u); uint4 a = q[0], b = q[1], c = q[2], d = q[3]; if ((a.x ^ b.y ^ c.z ^ d.w) == 0x12345678u) sink[0] = 1; }
}
__global__ void k_sparse32(const uint4 *p, int64_t n, u32 *sink) {
    int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n && chosen(i)) { uint4 a = p[i * 2], b = p[i * 2 + 1]; if ((a.x ^ b.y) == 0x12345678u) sink[0] = 1; }
}
__global__ void k_wsparse32(uint4 *p, int64_t n) {
    int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n && chosen(i)) { p[i * 2] = make_uint4((u32)i, 1, 2, 3); reinterpret_cast<u32 *>(p + i * 2 + 1)[0] = (u32)i; }
}
__global__ void k_wstream16(uint4 *p, int64_t n) { int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; if (i < n) p[i] = make_uint4((u32)i, 1, 2, 3); }

int main() {
    const int64_t n = 1 << 20;                     // lanes ("slots")
    size_t bytes = (size_t)n * 896;                // 0.94 GB: every kernel's footprint lies inside it
    uint4 *buf; u32 *sink;
    CHK(hipMalloc(&buf, bytes)); CHK(hipMalloc(&sink, 64));
    CHK(hipMemset(buf, 1, bytes));
    dim3 g((unsigned)(n / 256)), b(256);
    for (int rep = 0; rep < 3; rep++) {
        hipLaunchKernelGGL(k_stream16, g, b, 0, 0, buf, n, sink);
        hipLaunchKernelGGL(k_stream8, g, b, 0, 0, (const u64 *)buf + (64 << 20), n, sink);
        hipLaunchKernelGGL(k_stream1, g, b, 0, 0, (const uint8_t *)buf + (512 << 20), n, sink);
        hipLaunchKernelGGL(k_sparse64, g, b, 0, 0, buf, n, sink);
        hipLaunchKernelGGL(k_sparse32, g, b, 0, 0, buf + (32 << 20), n, sink);
        hipLaunchKernelGGL(k_wsparse32, g, b, 0, 0, buf + (40 << 20), n);
        hipLaunchKernelGGL(k_wstream16, g, b, 0, 0, buf + (48 << 20), n);
        CHK(hipDeviceSynchronize());
    }
    printf("{\"lanes\": %lld, \"chosen_fraction\": 0.1}\n", (long long)n);
    return 0;
}
