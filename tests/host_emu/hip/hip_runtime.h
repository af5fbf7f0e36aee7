// Host stand-in for <hip/hip_runtime.h>, for tests/test_device_rules_host.py ONLY: lets g++ compile
// tarok_amd/csrc/tarok_device.h (the device-side rule code) on the CPU, with the gfx950 builtins it
// uses emulated bit for bit (v_bitop3_b32, v_bfe_u32, popcounts, mul_hi).  Test infrastructure.
#pragma once
#include <stdint.h>
#include <algorithm>
#define __device__
#define __forceinline__ inline
#define __global__
using std::max;
using std::min;
static inline int __popc(unsigned x) { return __builtin_popcount(x); }
static inline int __popcll(unsigned long long x) { return __builtin_popcountll(x); }
// v_ffbh_u32 as HIP's __clz: leading zeros, 32 for 0
static inline int __clz(unsigned x) { return x ? __builtin_clz(x) : 32; }
static inline unsigned __umulhi(unsigned a, unsigned b) { return (unsigned)(((unsigned long long)a * b) >> 32); }
// v_bfe_u32: (src >> offset[4:0]) & ((1 << width[4:0]) - 1)
#define TK_KEEP_VGPR(x) ((void)0)
// v_sad_u8: sum of the absolute differences of the four bytes, plus c
static inline unsigned __builtin_amdgcn_sad_u8(unsigned a, unsigned b, unsigned c) {
    for (int i = 0; i < 4; i++) {
        int x = (int)((a >> (8 * i)) & 255u), y = (int)((b >> (8 * i)) & 255u);
        c += (unsigned)(x > y ? x - y : y - x);
    }
    return c;
}
static inline unsigned __builtin_amdgcn_ubfe(unsigned s, unsigned off, unsigned w) {
    off &= 31; w &= 31;
    return w ? (s >> off) & ((1u << w) - 1) : 0;
}
// v_bitop3_b32: bit i of the result = truth_table[{a_i, b_i, c_i}]; the builtin returns a SIGNED int
static inline int emu_bitop3(unsigned a, unsigned b, unsigned c, unsigned tt) {
    unsigned r = 0;
    for (int i = 0; i < 32; i++) {
        unsigned idx = (((a >> i) & 1) << 2) | (((b >> i) & 1) << 1) | ((c >> i) & 1);
        r |= ((tt >> idx) & 1u) << i;
    }
    return (int)r;
}
#define __builtin_amdgcn_bitop3_b32(a, b, c, tt) emu_bitop3((a), (b), (c), (tt))
// wave-level builtins only used by deal_wave (not exercised on the host)
static inline unsigned __lane_id() { return 0; }
static inline unsigned long long __ballot(int) { return 0; }
static inline int __builtin_amdgcn_readlane(int v, int) { return v; }
