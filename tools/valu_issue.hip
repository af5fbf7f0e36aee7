// Issue-rate microbenchmark for the vector/scalar instructions k_play is made of (gfx950).
//
// Question: does a wave64 VALU instruction cost 2 or 4 cycles of its SIMD,
// for THESE ops (v_bitop3, v_bcnt, v_bfe, v_min/max_u32, v_mul_lo/hi_u32, 64-bit shifts, v_cndmask,
// v_readlane, v_cmp ...), alone and with 2 / 4 / 8 waves per SIMD, on independent and on dependent
// chains — and does a wave with only 32 active lanes issue faster?
//
// One kernel per (op, chain kind): a loop of 64 copies of the instruction (inline asm, so the
// compiler cannot fold or reorder them), timed per wave with s_memtime (shader cycles) and
// s_memrealtime (100 MHz), the wave's HW_ID recorded so that the waves-per-SIMD placement is
// verified rather than assumed.  Placement is forced with dynamic LDS: one workgroup of 256 x w
// threads per CU (w <= 4), two of 1024 threads for w = 8.
//
//   hipcc --offload-arch=gfx950 -O3 -o tools/valu_issue tools/valu_issue.hip && tools/valu_issue > out.json
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <algorithm>
#include <string>
#include <vector>

typedef unsigned long long u64;
typedef uint32_t u32;

#define CHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(2); } } while (0)

// ---- instruction templates: I(r) = one instruction reading and writing register r (a 32-bit VGPR
// or, for the *_64 kernels, a 64-bit VGPR pair); %8 / %9 are loop-invariant 32-bit VGPR operands,
// s[10:11] a scalar mask, s12 a scalar.
#define I_and(r)        "v_and_b32 " r ", " r ", %8\n"
#define I_xor(r)        "v_xor_b32 " r ", " r ", %8\n"
#define I_or3(r)        "v_or3_b32 " r ", " r ", %8, %9\n"
#define I_add(r)        "v_add_u32 " r ", " r ", %8\n"
#define I_add3(r)       "v_add3_u32 " r ", " r ", %8, %9\n"
#define I_lshr(r)       "v_lshrrev_b32 " r ", 1, " r "\n"
#define I_lshl_or(r)    "v_lshl_or_b32 " r ", " r ", 1, %8\n"
#define I_mov(r)        "v_mov_b32 " r ", %8\n"
#define I_bitop3(r)     "v_bitop3_b32 " r ", " r ", %8, %9 bitop3:0x96\n"
#define I_bcnt(r)       "v_bcnt_u32_b32 " r ", " r ", %8\n"
#define I_bfe(r)        "v_bfe_u32 " r ", " r ", %8, %9\n"
#define I_min(r)        "v_min_u32 " r ", " r ", %8\n"
#define I_max(r)        "v_max_u32 " r ", " r ", %8\n"
#define I_mul_lo(r)     "v_mul_lo_u32 " r ", " r ", %8\n"
#define I_mul_hi(r)     "v_mul_hi_u32 " r ", " r ", %8\n"
#define I_mul_u24(r)    "v_mul_u32_u24 " r ", " r ", %8\n"
#define I_mad_u24(r)    "v_mad_u32_u24 " r ", " r ", %8, %9\n"
#define I_cndmask(r)    "v_cndmask_b32 " r ", " r ", %8, s[10:11]\n"
#define I_cmp(r)        "v_cmp_lt_u32 s[10:11], " r ", %8\n"
#define I_cmp_vcc(r)    "v_cmp_lt_u32 vcc, " r ", %8\n"
#define I_readlane(r)   "v_readlane_b32 s12, " r ", 3\n"
#define I_sdwa(r)       "v_xor_b32_sdwa " r ", " r ", %8 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_1\n"
#define I_fma(r)        "v_fma_f32 " r ", " r ", %8, %9\n"
#define I_snop(r)       "s_nop 0\n"
#define I_sand(r)       "s_and_b64 s[10:11], s[10:11], exec\n"
#define I_sadd(r)       "s_add_u32 s12, s12, 1\n"
// branches (the 1: label is the next instruction but one: a taken branch skips one s_nop).  br_taken: 2
// instructions executed per copy, br_not: 3, execz (not taken): 2, saveexec (the compiler's `if` on a per-lane
// condition: s_and_saveexec, s_cbranch_execz not taken, body, s_or exec): 4, ballot_br (the `if (__ballot(c))` shape:
// v_cmp into an SGPR pair, s_cmp_lg_u64, s_cbranch_scc1 taken): 3, ballot_nobr (the same, not taken): 4
#define I_br_taken(r)    "s_cmp_eq_u32 s12, s12\ns_cbranch_scc1 1f\ns_nop 0\n1:\n"
#define I_br_not(r)      "s_cmp_eq_u32 s12, s12\ns_cbranch_scc0 1f\ns_nop 0\n1:\n"
#define I_execz(r)       "s_cbranch_execz 1f\ns_nop 0\n1:\n"
#define I_saveexec(r)    "s_and_saveexec_b64 s[10:11], exec\ns_cbranch_execz 1f\ns_nop 0\n1:\ns_or_b64 exec, exec, s[10:11]\n"
#define I_ballot_br(r)   "v_cmp_ge_u32 s[10:11], " r ", " r "\ns_cmp_lg_u64 s[10:11], 0\ns_cbranch_scc1 1f\ns_nop 0\n1:\n"
#define I_ballot_nobr(r) "v_cmp_ge_u32 s[10:11], " r ", " r "\ns_cmp_lg_u64 s[10:11], 0\ns_cbranch_scc0 1f\ns_nop 0\n1:\n"
// a select on a comparison, three ways (3 instruction slots each).  cmpsel: v_cmp into vcc, the s_nop 1 that gfx950
// requires before a VALU reads an SGPR a VALU wrote, v_cndmask; cmpsel_sgpr: the same through an SGPR pair;
// arithsel: no mask register at all: the sign of a difference smeared over the word picks the operand (v_bitop3)
#define I_cmpsel(r)      "v_cmp_lt_u32 vcc, " r ", %8\ns_nop 1\nv_cndmask_b32 " r ", " r ", %9, vcc\n"
#define I_cmpsel_sgpr(r) "v_cmp_lt_u32 s[10:11], " r ", %8\ns_nop 1\nv_cndmask_b32 " r ", " r ", %9, s[10:11]\n"
#define I_arithsel(r)    "v_sub_u32 v100, " r ", %8\nv_ashrrev_i32 v100, 31, v100\nv_bitop3_b32 " r ", " r ", %9, v100 bitop3:0xd8\n"
// 64-bit forms (r = a VGPR pair)
#define I_lshl64(r)     "v_lshlrev_b64 " r ", 1, " r "\n"
#define I_lshl64v(r)    "v_lshlrev_b64 " r ", %8, " r "\n"
#define I_lshr64(r)     "v_lshrrev_b64 " r ", 1, " r "\n"
#define I_lshl_add64(r) "v_lshl_add_u64 " r ", " r ", 1, " r "\n"
#define I_mov64(r)      "v_mov_b64 " r ", " r "\n"
#define I_mad64(r)      "v_mad_u64_u32 " r ", s[10:11], %8, %9, " r "\n"
// the v_cmp -> s_and -> v_cndmask pattern of a select on a compound condition (3 instructions)
#define I_select(r)     "v_cmp_lt_u32 s[10:11], " r ", %8\ns_and_b64 s[10:11], s[10:11], vcc\nv_cndmask_b32 " r ", " r ", %9, s[10:11]\n"

#define IND8(I) I("%0") I("%1") I("%2") I("%3") I("%4") I("%5") I("%6") I("%7")
#define DEP8(I) I("%0") I("%0") I("%0") I("%0") I("%0") I("%0") I("%0") I("%0")
#define REP8(x) x x x x x x x x
#define PER_ITER 64

struct Rec { u64 cycles, real; u32 hw_id, xcc; };

// LANES: 0 = all 64 lanes, 1 = lanes 0..31 only, 2 = even lanes only (32 active, both halves)
#define KERNEL(NAME, T, BODY)                                                                              \
    __global__ __launch_bounds__(1024) void NAME(Rec *out, int iters, int lanes, T seed) {                 \
        extern __shared__ char lds_pad[];                                                                  \
        T x0 = seed + threadIdx.x, x1 = x0 * 3 + 1, x2 = x0 * 5 + 2, x3 = x0 * 7 + 3, x4 = x0 * 11 + 4,     \
          x5 = x0 * 13 + 5, x6 = x0 * 17 + 6, x7 = x0 * 19 + 7;                                            \
        u32 a = (u32)seed | 5u, b = ((u32)seed & 7u) + 9u;                                                 \
        u32 lane = threadIdx.x & 63;                                                                       \
        bool on = lanes == 0 || (lanes == 1 && lane < 32) || (lanes == 2 && !(lane & 1));                  \
        u64 t0 = 0, t1 = 0, r0 = 0, r1 = 0;                                                                \
        if (on) {                                                                                          \
            asm volatile("s_mov_b64 s[10:11], 0x55555555\ns_mov_b32 s12, 0" ::: "s10", "s11", "s12");     \
            r0 = __builtin_amdgcn_s_memrealtime();                                                         \
            t0 = __builtin_amdgcn_s_memtime();                                                             \
            for (int it = 0; it < iters; it++)                                                             \
                asm volatile(BODY : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6),  \
                             "+v"(x7) : "v"(a), "v"(b) : "s10", "s11", "s12", "vcc", "scc", "v100");                      \
            t1 = __builtin_amdgcn_s_memtime();                                                             \
            r1 = __builtin_amdgcn_s_memrealtime();                                                         \
        }                                                                                                  \
        T sink = x0 ^ x1 ^ x2 ^ x3 ^ x4 ^ x5 ^ x6 ^ x7;                                                    \
        if (sink == (T)0x123456789ABCDEFULL) lds_pad[threadIdx.x] = 1;   /* keep the chains alive */       \
        if (lane == 0) {                                                                                   \
            u32 hw, xcc;                                                                                   \
            asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)\ns_getreg_b32 %1, hwreg(HW_REG_XCC_ID)"     \
                         : "=s"(hw), "=s"(xcc));                                                           \
            Rec r = {t1 - t0, r1 - r0, hw, xcc};                                                           \
            out[(blockIdx.x * blockDim.x + threadIdx.x) >> 6] = r;                                         \
        }                                                                                                  \
    }

#define OP32(X) \
    X(and, 1) X(xor, 1) X(or3, 1) X(add, 1) X(add3, 1) X(lshr, 1) X(lshl_or, 1) X(mov, 1) X(bitop3, 1) X(bcnt, 1) X(bfe, 1) \
    X(min, 1) X(max, 1) X(mul_lo, 1) X(mul_hi, 1) X(mul_u24, 1) X(mad_u24, 1) X(cndmask, 1) X(cmp, 1) X(cmp_vcc, 1)          \
    X(readlane, 1) X(sdwa, 1) X(fma, 1) X(snop, 1) X(sand, 1) X(sadd, 1) X(select, 3)                                       \
    X(br_taken, 2) X(br_not, 3) X(execz, 2) X(saveexec, 4) X(ballot_br, 3) X(ballot_nobr, 4)                          \
    X(cmpsel, 3) X(cmpsel_sgpr, 3) X(arithsel, 3)
#define OP64(X) X(lshl64, 1) X(lshl64v, 1) X(lshr64, 1) X(lshl_add64, 1) X(mov64, 1) X(mad64, 1)

#define DEF32(n, k) KERNEL(k32_##n##_ind, u32, REP8(IND8(I_##n))) KERNEL(k32_##n##_dep, u32, REP8(DEP8(I_##n)))
#define DEF64(n, k) KERNEL(k64_##n##_ind, u64, REP8(IND8(I_##n))) KERNEL(k64_##n##_dep, u64, REP8(DEP8(I_##n)))
OP32(DEF32)
OP64(DEF64)

struct Op { const char *name; int per_copy; void (*ind32)(Rec *, int, int, u32); void (*dep32)(Rec *, int, int, u32);
            void (*ind64)(Rec *, int, int, u64); void (*dep64)(Rec *, int, int, u64); };
#define ROW32(n, k) {#n, k, k32_##n##_ind, k32_##n##_dep, nullptr, nullptr},
#define ROW64(n, k) {#n, k, nullptr, nullptr, k64_##n##_ind, k64_##n##_dep},
static Op g_ops[] = {OP32(ROW32) OP64(ROW64)};

struct Result { double cyc_per_instr, clock_mhz, waves_per_simd_seen, wall_us, chip_cyc, cyc_min, cyc_max; int waves; };

static Result run(const Op &op, bool dep, int wps, int lanes, int iters, Rec *d_out, std::vector<Rec> &h) {
    // wps waves per SIMD: w <= 4 -> one workgroup of 256 x w threads per CU (LDS 120 KB: a second
    // one does not fit); w = 8 -> two workgroups of 1024 threads per CU (LDS 70 KB each: a third does not fit)
    int threads = wps <= 4 ? 256 * wps : 1024;
    int blocks = wps <= 4 ? 256 : 512;
    size_t lds = wps <= 4 ? 120 * 1024 : 70 * 1024;
    const void *fn = op.ind32 ? (dep ? (const void *)op.dep32 : (const void *)op.ind32)
                              : (dep ? (const void *)op.dep64 : (const void *)op.ind64);
    static std::vector<const void *> attr_set;
    if (std::find(attr_set.begin(), attr_set.end(), fn) == attr_set.end()) {
        CHK(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, 120 * 1024));
        attr_set.push_back(fn);
    }
    int nw = blocks * threads / 64;
    static hipEvent_t e0 = nullptr, e1 = nullptr;
    if (!e0) { CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1)); }
    for (int rep = 0; rep < 2; rep++) {                    // first launch warms the instruction cache
        CHK(hipEventRecord(e0, 0));
        if (op.ind32) hipLaunchKernelGGL(dep ? op.dep32 : op.ind32, dim3(blocks), dim3(threads), lds, 0, d_out, iters, lanes, 12345u);
        else hipLaunchKernelGGL(dep ? op.dep64 : op.ind64, dim3(blocks), dim3(threads), lds, 0, d_out, iters, lanes, (u64)0x12345678912345ULL);
        CHK(hipEventRecord(e1, 0));
        CHK(hipEventSynchronize(e1));
    }
    CHK(hipGetLastError());
    float ms = 0;
    CHK(hipEventElapsedTime(&ms, e0, e1));
    h.resize(nw);
    CHK(hipMemcpy(h.data(), d_out, nw * sizeof(Rec), hipMemcpyDeviceToHost));
    std::vector<double> cyc, clk;
    std::vector<u32> simd_key;
    for (auto &r : h) {
        cyc.push_back((double)r.cycles);
        if (r.real) clk.push_back((double)r.cycles / (double)r.real * 100.0);
        // HW_ID: wave_id[3:0] simd_id[5:4] pipe[7:6] cu_id[11:8] sh_id[12] se_id[15:13]; + XCC id
        simd_key.push_back(((r.xcc & 15u) << 16) | (r.hw_id & 0xFF30u));
    }
    std::sort(cyc.begin(), cyc.end());
    std::sort(clk.begin(), clk.end());
    std::sort(simd_key.begin(), simd_key.end());
    size_t distinct = std::unique(simd_key.begin(), simd_key.end()) - simd_key.begin();
    Result res;
    res.cyc_per_instr = cyc[cyc.size() / 2] / ((double)iters * PER_ITER * op.per_copy);
    res.clock_mhz = clk.empty() ? 0 : clk[clk.size() / 2];
    res.waves_per_simd_seen = (double)nw / (double)distinct;
    res.wall_us = ms * 1e3;
    res.waves = nw;
    // chip-level check, independent of the per-wave stamps and of the placement: SIMD cycles per
    // wave-instruction = wall time x clock x 1024 SIMDs / (waves x instructions per wave)
    res.chip_cyc = (double)ms * 1e-3 * res.clock_mhz * 1e6 * 1024.0 / ((double)nw * iters * PER_ITER * op.per_copy);
    res.cyc_min = cyc.front() / ((double)iters * PER_ITER * op.per_copy);
    res.cyc_max = cyc.back() / ((double)iters * PER_ITER * op.per_copy);
    return res;
}

int main(int argc, char **argv) {
    int iters = argc > 1 ? atoi(argv[1]) : 1500;
    const char *only = argc > 2 ? argv[2] : nullptr;          // comma-separated op names (default: all)
    CHK(hipSetDevice(0));
    Rec *d_out;
    CHK(hipMalloc((void **)&d_out, 8192 * sizeof(Rec)));
    std::vector<Rec> h;
    const int wpss[] = {1, 2, 4, 8};
    printf("{\n \"iters\": %d, \"instructions_per_iteration\": %d,\n \"unit\": \"shader cycles (s_memtime) per wave-instruction, median over the waves\",\n \"ops\": {\n", iters, PER_ITER);
    size_t nops = sizeof(g_ops) / sizeof(g_ops[0]);
    double clock_sum = 0; int clock_n = 0;
    bool first = true;
    for (size_t o = 0; o < nops; o++) {
        if (only && !strstr((std::string(",") + only + ",").c_str(), (std::string(",") + g_ops[o].name + ",").c_str())) continue;
        printf("%s  \"%s\": {", first ? "" : ",\n", g_ops[o].name);
        first = false;
        for (int dep = 0; dep < 2; dep++) {
            printf("\"%s\": {", dep ? "dep" : "ind");
            for (int k = 0; k < 4; k++) {
                Result r = run(g_ops[o], dep != 0, wpss[k], 0, iters, d_out, h);
                printf("\"w%d\": {\"cyc\": %.3f, \"min\": %.3f, \"max\": %.3f, \"per_simd\": %.3f, \"chip\": %.3f, \"placed\": %.2f, \"mhz\": %.0f, \"wall_us\": %.1f}%s",
                       wpss[k], r.cyc_per_instr, r.cyc_min, r.cyc_max, r.cyc_per_instr / wpss[k], r.chip_cyc, r.waves_per_simd_seen, r.clock_mhz, r.wall_us, k < 3 ? ", " : "");
                if (r.clock_mhz > 0) { clock_sum += r.clock_mhz; clock_n++; }
            }
            printf("}%s", dep ? "" : ", ");
        }
        printf("}");
        fflush(stdout);
    }
    printf("\n },\n \"half_waves\": {\n");
    // does a wave with 32 active lanes issue in fewer cycles?  (lanes: 1 = low half only, 2 = even lanes)
    const char *probe[] = {"and", "bcnt", "mul_lo", "lshl64", "cndmask", "bitop3"};
    for (size_t p = 0; p < 6; p++) {
        const Op *op = nullptr;
        for (size_t o = 0; o < nops; o++) if (!strcmp(g_ops[o].name, probe[p])) op = &g_ops[o];
        printf("  \"%s\": {", probe[p]);
        for (int lanes = 0; lanes < 3; lanes++) {
            printf("\"%s\": {", lanes == 0 ? "all64" : (lanes == 1 ? "low32" : "even32"));
            for (int k = 0; k < 4; k++) {
                Result r = run(*op, false, wpss[k], lanes, iters, d_out, h);
                printf("\"w%d\": %.3f%s", wpss[k], r.cyc_per_instr, k < 3 ? ", " : "");
            }
            printf("}%s", lanes < 2 ? ", " : "");
        }
        printf("}%s\n", p < 5 ? "," : "");
        fflush(stdout);
    }
    printf(" },\n \"clock_mhz_mean\": %.1f\n}\n", clock_n ? clock_sum / clock_n : 0.0);
    CHK(hipFree(d_out));
    return 0;
}
