// Diagnostic (GPU box): does gfx950 have gfx90a's "64-bit shift with the amount in the last allocated VGPR" erratum?
// (LLVM: GCNHazardRecognizer::fixShift64HighRegBug, applied to gfx90a only.)  Round 4: root cause of the refill-role
// corruption — k_step's failing builds had 104 VGPRs and `v_lshlrev_b64 v[70:71], v103, 1`.
//
// Every kernel below owns VGPRs v0..vTOP exactly (the asm clobber of vTOP makes it the highest register used, TOP & 7 == 7,
// so the allocation ends right behind it), puts a shift amount into vTOP (LAST) or vTOP-1 (control), a marker (37) into v0 and
// shifts 1 by it, many times; a wave reports how many results were wrong, how many of the wrong ones were 1 << 37 (the
// hardware's "source out of range -> VGPR0" substitution), its GPR_ALLOC register (physical VGPR base / size) and HW_ID.
//
//   hipcc --offload-arch=gfx950 -O3 -o tools/shift64_probe tools/shift64_probe.hip && tools/shift64_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <vector>
#include <map>
#include <string>

#define STR2(x) #x
#define STR(x) STR2(x)
#define PROBE_KERNEL(NAME, TOP, AMT) PROBE_KERNEL_OP(NAME, TOP, AMT, "v_lshlrev_b64 %0, v" STR(AMT) ", 1", (1ULL << amt), (1ULL << 37))
#define PROBE_KERNEL_OP(NAME, TOP, AMT, INSN, EXPECT, EXPECT_V0)                                                                                  \
    __global__ __launch_bounds__(256) void NAME(uint32_t *out, uint32_t iters, uint32_t spin) {                        \
        uint32_t wrong = 0, wrong_v0 = 0, lane = threadIdx.x & 63;                                                     \
        for (uint32_t it = 0; it < iters; it++) {                                                                      \
            uint32_t amt = (lane * 7 + it * 13 + 3) % 54;                                                              \
            if (amt == 37) amt = 38;                                                                                   \
            unsigned long long r;                                                                                      \
            asm volatile("v_mov_b32 v0, 37\n\tv_mov_b32 v" STR(AMT) ", %1\n\ts_nop 4\n\t" INSN "\n\ts_nop 1"                    \
                         : "=v"(r) : "v"(amt), "v"(0x8123456789ABCDEFULL) : "v0", "v" STR(TOP), "v" STR(AMT), "s10", "s11"); \
            if (r != (unsigned long long)(EXPECT)) { wrong++; if (r == (unsigned long long)(EXPECT_V0)) wrong_v0++; }  \
            for (uint32_t s = 0; s < spin; s++) asm volatile("s_nop 7");                                               \
        }                                                                                                              \
        uint32_t alloc, hwid;                                                                                          \
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_GPR_ALLOC)" : "=s"(alloc));                                        \
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid));                                             \
        uint32_t w = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;                                                     \
        uint32_t tw = __reduce_add_sync_dummy(wrong), tv = __reduce_add_sync_dummy(wrong_v0);                          \
        if (lane == 0) { out[4 * w] = tw; out[4 * w + 1] = tv; out[4 * w + 2] = alloc; out[4 * w + 3] = hwid; }        \
    }

__device__ __forceinline__ uint32_t __reduce_add_sync_dummy(uint32_t v) {     // wave sum
    for (int o = 32; o; o >>= 1) v += __shfl_xor(v, o);
    return v;
}

PROBE_KERNEL(k_last_15, 15, 15)
PROBE_KERNEL(k_ctrl_15, 15, 14)
PROBE_KERNEL(k_last_63, 63, 63)
PROBE_KERNEL(k_ctrl_63, 63, 62)
PROBE_KERNEL(k_last_103, 103, 103)
PROBE_KERNEL(k_ctrl_103, 103, 102)
PROBE_KERNEL(k_mid_103, 103, 95)          // index & 7 == 7 but NOT the last register of the allocation
PROBE_KERNEL(k_last_127, 127, 127)
PROBE_KERNEL(k_ctrl_127, 127, 126)
PROBE_KERNEL_OP(k_lshr_103, 103, 103, "v_lshrrev_b64 %0, v103, %2", (0x8123456789ABCDEFULL >> amt), (0x8123456789ABCDEFULL >> 37))
PROBE_KERNEL_OP(k_ashr_103, 103, 103, "v_ashrrev_i64 %0, v103, %2", ((long long)0x8123456789ABCDEFULL >> amt), ((long long)0x8123456789ABCDEFULL >> 37))
// other instructions that mix a 32-bit VGPR operand with 64-bit ones (is the erratum wider than LLVM's list of three?):
//   v_mad_u64_u32 d, a32, b32, c64 with a32 in the last register; v_cvt_f64_u32 d64, a32; v_ldexp_f64 d64, a64, e32;
//   v_lshl_add_u64 with its 64-bit addend ENDING in the last register (a genuine pair: control)
PROBE_KERNEL_OP(k_mad64_103, 103, 103, "v_mad_u64_u32 %0, s[10:11], v103, 3, %2", (0x8123456789ABCDEFULL + 3ULL * amt), (0x8123456789ABCDEFULL + 3ULL * 37))
PROBE_KERNEL_OP(k_cvt64_103, 103, 103, "v_cvt_f64_u32 %0, v103", __double_as_longlong((double)amt), __double_as_longlong(37.0))
PROBE_KERNEL_OP(k_ldexp_103, 103, 103, "v_ldexp_f64 %0, 1.0, v103", __double_as_longlong(ldexp(1.0, (int)amt)), __double_as_longlong(ldexp(1.0, 37)))

typedef void (*kern_t)(uint32_t *, uint32_t, uint32_t);

int main() {
    struct { const char *name; kern_t k; } ks[] = {
        {"amount in v15  = last of 16", k_last_15}, {"amount in v14  (control)", k_ctrl_15},
        {"amount in v63  = last of 64", k_last_63}, {"amount in v62  (control)", k_ctrl_63},
        {"amount in v103 = last of 104", k_last_103}, {"amount in v102 (control)", k_ctrl_103}, {"amount in v95  (index & 7 == 7, not last)", k_mid_103},
        {"amount in v127 = last of 128", k_last_127}, {"amount in v126 (control)", k_ctrl_127},
        {"v_lshrrev_b64, amount in v103 = last of 104", k_lshr_103}, {"v_ashrrev_i64, amount in v103 = last of 104", k_ashr_103},
        {"v_mad_u64_u32, 32-bit factor in v103 = last of 104", k_mad64_103}, {"v_cvt_f64_u32, source in v103 = last of 104", k_cvt64_103},
        {"v_ldexp_f64, exponent in v103 = last of 104", k_ldexp_103},
    };
    const int blocks = 2048, threads = 256, waves = blocks * threads / 64;
    uint32_t *d;
    (void)hipMalloc(&d, waves * 16);
    std::vector<uint32_t> h(waves * 4);
    for (auto &e : ks) {
        (void)hipMemset(d, 0, waves * 16);
        hipLaunchKernelGGL(e.k, dim3(blocks), dim3(threads), 0, 0, d, 2000u, 4u);
        if (hipDeviceSynchronize() != hipSuccess) { printf("%s: launch failed\n", e.name); return 1; }
        (void)hipMemcpy(h.data(), d, waves * 16, hipMemcpyDeviceToHost);
        uint64_t wrong = 0, wrong_v0 = 0, bad_waves = 0;
        std::map<uint32_t, std::pair<uint32_t, uint32_t>> by_base;      // VGPR base (in granules of 8) -> waves, bad waves
        for (int w = 0; w < waves; w++) {
            wrong += h[4 * w]; wrong_v0 += h[4 * w + 1];
            uint32_t base = h[4 * w + 2] & 63, size = (h[4 * w + 2] >> 8) & 63;
            auto &b = by_base[base | (size << 8)];
            b.first++;
            if (h[4 * w]) { bad_waves++; b.second++; }
        }
        printf("%-44s wrong %8llu of %llu shifts (%.3g), of them 1 << v0: %llu; waves with a wrong result: %llu of %d\n", e.name,
               (unsigned long long)wrong, (unsigned long long)waves * 64 * 2000, (double)wrong / ((double)waves * 64 * 2000),
               (unsigned long long)wrong_v0, (unsigned long long)bad_waves, waves);
        printf("    by GPR_ALLOC (VGPR base*8 .. +(size+1)*8): ");
        for (auto &kv : by_base)
            printf("[%u..%u) %u/%u  ", (kv.first & 63) * 8, (kv.first & 63) * 8 + (((kv.first >> 8) & 63) + 1) * 8, kv.second.second, kv.second.first);
        printf("\n");
    }
    return 0;
}
