// Device-side Tarok rules on bit-packed per-game lanes (gfx950 / CDNA4 only).
//
// One game = 4 x uint64, kept as two 16-byte pairs in two SoA arrays so a wave
// reads/writes 1 KiB contiguous per instruction:
//
//   play pair (rewritten by every card)
//     X0 = C plane [53:0] | n_in_trick<<54 (2) | leader<<56 (2) | trick_no<<58 (4) | phase<<62 (2)
//     X1 = talon 6x6-bit ids [35:0] | current trick 4x6-bit ids [59:36] | tl<<60 (3) | error<<63
//   seat pair (rewritten only when a trick is resolved: every 4th card)
//     Y0 = A plane [53:0] | contract<<54 (4) | declarer<<58 (2) | king<<60 (2) | cprev[3:2]<<62
//     Y1 = B plane [53:0] | team<<54 (4) | epar<<58 (4) | cprev[1:0]<<62
//          epar = the slot's episode number mod TK_AHEAD: says which of the slot's next-game
//          lines holds the next game; cprev = how many of them (the farthest ahead) were put
//          on a refill list by the previous launch, i.e. are being re-dealt during this one
//
// Card c (bit c of every plane) belongs to seat (B_c A_c).  C_c = 0: it is in
// that seat's hand (Igralec.roka).  C_c = 1: it has left the hands — it lies in
// that seat's won pile (Igralec.kupcek), or on the table (attributed to whoever
// played it until the trick is resolved), or it is a talon card nobody owns yet
// (parked in an opponent's pile bits; told apart through the ordered talon ids).
// `tl` is Klop's len(self.talon) (Klop.py:67) or, for Tri..Solo_ena, the chosen
// talon group (7 = none yet).
//
// Reference rules cited per function; the CPU statement of the same rules is
// oracle/tarok_oracle.c (test infrastructure, never linked here).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef unsigned long long u64;
typedef uint32_t u32;

// gfx950 ERRATUM GUARD (DESIGN.md §3 "the refill-role bug", tarok_amd/isa_check.py, profiles/r04_refill_root_cause.txt).
// v_lshlrev_b64 / v_lshrrev_b64 / v_ashrrev_i64 shift by v0 instead of by their amount register when that register is the
// LAST VGPR the wave was allocated (gfx90a's "Shift64HighRegBug"; ROCm 7.2.0's compiler works round it on gfx90a only).
// Every kernel of this library is declared TK_KERNEL(threads, B) — it may use v0 .. v(B-2) only — and starts with
// TK_VGPR_TOP(B, B-1), which marks v(B-1) used: the wave's allocation is always exactly B registers (B a multiple of the
// allocation granule, 8) and no value ever lives in the last one.  B is the kernel's occupancy bucket (<= 64: 8 waves per
// SIMD, 80: 6, 96: 5, 128: 4, 168: 3, 256: 2), so the guard costs one register of the bucket and no occupancy.  (A kernel
// that needs more than 256 registers — k_learn_dw — holds the rest in accumulation registers, which follow the VGPRs in the
// unified file: the row behind its last VGPR is always allocated.)
#define TK_VGPR_BUDGET(B) __attribute__((amdgpu_num_vgpr((B) - 1)))
#define TK_KERNEL(threads, B) __global__ __launch_bounds__(threads) TK_VGPR_BUDGET(B)
#define TK_VGPR_STR2(x) #x
#define TK_VGPR_STR(x) TK_VGPR_STR2(x)
#define TK_VGPR_TOP(B, TOP)                                                                                           \
    do {                                                                                                              \
        static_assert((B) % 8 == 0 && (B) <= 256 && (TOP) == (B) - 1, "TK_VGPR_TOP(B, B - 1), B a multiple of 8");     \
        asm volatile("" ::: "v" TK_VGPR_STR(TOP));                                                                    \
    } while (0)

#define TK_BIT(i) (1ULL << (i))
#define TK_DECK ((1ULL << 54) - 1)
#define TK_TAROK (((1ULL << 22) - 1) << 32)
#define TK_PAGAT TK_BIT(32)
// Roka.vrednost_stiha card values (Roka.py:76-95)
#define TK_V5 (TK_BIT(7) | TK_BIT(15) | TK_BIT(23) | TK_BIT(31) | TK_BIT(32) | TK_BIT(52) | TK_BIT(53))
#define TK_V4 (TK_BIT(6) | TK_BIT(14) | TK_BIT(22) | TK_BIT(30))
#define TK_V3 (TK_BIT(5) | TK_BIT(13) | TK_BIT(21) | TK_BIT(29))
#define TK_V2 (TK_BIT(4) | TK_BIT(12) | TK_BIT(20) | TK_BIT(28))
// Roka.mozno_zalozit (Roka.py:23-27) with Karta.vrednost (Karta.py:10-16):
// suit ranks 1..7 and taroks 2..7
#define TK_DISCARDABLE (0x7F7F7F7FULL | (0x3FULL << 33))

enum { TK_KLOP = 0, TK_TRI, TK_DVE, TK_ENA, TK_SOLO_TRI, TK_SOLO_DVE, TK_SOLO_ENA, TK_BERAC, TK_SOLO_BREZ, TK_ODPRTI_BERAC };
enum { TK_PHASE_EXCHANGE = 1, TK_PHASE_PLAY = 2, TK_PHASE_DONE = 3 };

struct Game {
    u64 A, B, C, talon;
    u32 trick, nt, leader, trick_no, phase, contract, declarer, king, error, team, tl, epar, cprev;
};

// (x0, x1) = play pair, (y0, y1) = seat pair
__device__ __forceinline__ void unpack(Game &g, u64 x0, u64 x1, u64 y0, u64 y1) {
    g.C = x0 & TK_DECK; g.A = y0 & TK_DECK; g.B = y1 & TK_DECK;
    u32 m0 = (u32)(x0 >> 54), m2 = (u32)(y0 >> 54), m3 = (u32)(y1 >> 54);
    g.nt = m0 & 3; g.leader = (m0 >> 2) & 3; g.trick_no = (m0 >> 4) & 15; g.phase = m0 >> 8;
    g.talon = x1 & ((1ULL << 36) - 1);
    g.trick = (u32)(x1 >> 36) & 0xFFFFFF;
    g.tl = (u32)(x1 >> 60) & 7; g.error = (u32)(x1 >> 63);
    g.contract = m2 & 15; g.declarer = (m2 >> 4) & 3; g.king = (m2 >> 6) & 3;
    g.team = m3 & 15; g.epar = (m3 >> 4) & 15; g.cprev = (m3 >> 8) | ((m2 >> 8) << 2);
}

// The same for a line of the dealt-ahead buffer: a game at its first card (deal_into_buffer: setup_game, Bot
// exchange), so no card of a trick and nothing to re-deal.  Written as constants: where the caller is at the end
// of a trick of EVERY lane (the trick-aligned card loops) "no card led" stays a compile-time fact across the
// swap, and the next legal mask is the leader's (the hand, pagat rule) without the follow-suit selects.
__device__ __forceinline__ void unpack_fresh(Game &g, u64 x0, u64 x1, u64 y0, u64 y1) {
    g.C = x0 & TK_DECK; g.A = y0 & TK_DECK; g.B = y1 & TK_DECK;
    u32 m0 = (u32)(x0 >> 54), m2 = (u32)(y0 >> 54), m3 = (u32)(y1 >> 54);
    g.nt = 0; g.leader = (m0 >> 2) & 3; g.trick_no = (m0 >> 4) & 15; g.phase = m0 >> 8;
    g.talon = x1 & ((1ULL << 36) - 1);
    g.trick = 0;
    g.tl = (u32)(x1 >> 60) & 7; g.error = (u32)(x1 >> 63);
    g.contract = m2 & 15; g.declarer = (m2 >> 4) & 3; g.king = (m2 >> 6) & 3;
    g.team = m3 & 15; g.epar = (m3 >> 4) & 15; g.cprev = 0;
}

__device__ __forceinline__ void pack_play(const Game &g, u64 &x0, u64 &x1) {
    x0 = g.C | ((u64)(g.nt | (g.leader << 2) | (g.trick_no << 4) | (g.phase << 8)) << 54);
    x1 = g.talon | ((u64)g.trick << 36) | ((u64)g.tl << 60) | ((u64)g.error << 63);
}
__device__ __forceinline__ void pack_seats(const Game &g, u64 &y0, u64 &y1) {
    y0 = g.A | ((u64)(g.contract | (g.declarer << 4) | (g.king << 6) | ((g.cprev >> 2) << 8)) << 54);
    y1 = g.B | ((u64)(g.team | (g.epar << 4) | ((g.cprev & 3) << 8)) << 54);
}
__device__ __forceinline__ void pack(const Game &g, u64 &x0, u64 &x1, u64 &y0, u64 &y1) {
    pack_play(g, x0, x1);
    pack_seats(g, y0, y1);
}

__device__ __forceinline__ int popc64(u64 m) { return __popcll(m); }

// v_bitop3_b32: any boolean function of three 32-bit words in one instruction.  The immediate
// is the function's truth table, written here by evaluating it on the three selector bytes.
template <class F> constexpr u32 tk_tt(F f) { return f(0xF0u, 0xCCu, 0xAAu) & 0xFFu; }
#define TK_BITOP3(a, b, c, ...) ((u32)__builtin_amdgcn_bitop3_b32((a), (b), (c), tk_tt([](u32 a_, u32 b_, u32 c_) { return (__VA_ARGS__); })))
#define TK_LO(x) ((u32)(x))
#define TK_HI(x) ((u32)((x) >> 32))
#ifndef TK_KEEP_VGPR                         // value materialised in a VGPR here (the CPU build of the tests defines it away)
#define TK_KEEP_VGPR(x) asm volatile("" : "+v"(x))
#endif
#define TK_U64(lo, hi) (((u64)(u32)(hi) << 32) | (u64)(u32)(lo))

__device__ __forceinline__ u64 seat_cards(const Game &g, u32 s) {
    u64 a = (s & 1) ? g.A : ~g.A, b = (s & 2) ? g.B : ~g.B;
    return a & b & TK_DECK;
}
// hand of seat s = ~C & (A == s&1) & (B == s>>1): with ma / mb = all-ones when the seat bit is
// set, (A xnor ma) & (B xnor mb) & ~C is two v_bitop3 per 32-bit half
__device__ __forceinline__ u64 hand_of(const Game &g, u32 s) {
    u32 ma = (u32)((int)(s << 31) >> 31), mb = (u32)((int)(s << 30) >> 31);
    u32 tl = TK_BITOP3(TK_LO(g.A), ma, TK_LO(g.C), ~c_ & ~(a_ ^ b_));
    u32 th = TK_BITOP3(TK_HI(g.A), ma, TK_HI(g.C), ~c_ & ~(a_ ^ b_));
    u32 hl = TK_BITOP3(tl, TK_LO(g.B), mb, a_ & ~(b_ ^ c_));
    u32 hh = TK_BITOP3(th, TK_HI(g.B), mb, a_ & ~(b_ ^ c_)) & 0x3FFFFFu;
    return TK_U64(hl, hh);
}
// won pile of seat S (incl. whatever is parked in its pile bits): C & (A == S&1) & (B == S>>1)
template <int S> __device__ __forceinline__ u64 pile_of(const Game &g) {
    u32 l, h;
    if (S == 0)      { l = TK_BITOP3(TK_LO(g.A), TK_LO(g.B), TK_LO(g.C), ~a_ & ~b_ & c_); h = TK_BITOP3(TK_HI(g.A), TK_HI(g.B), TK_HI(g.C), ~a_ & ~b_ & c_); }
    else if (S == 1) { l = TK_BITOP3(TK_LO(g.A), TK_LO(g.B), TK_LO(g.C), a_ & ~b_ & c_);  h = TK_BITOP3(TK_HI(g.A), TK_HI(g.B), TK_HI(g.C), a_ & ~b_ & c_); }
    else if (S == 2) { l = TK_BITOP3(TK_LO(g.A), TK_LO(g.B), TK_LO(g.C), ~a_ & b_ & c_);  h = TK_BITOP3(TK_HI(g.A), TK_HI(g.B), TK_HI(g.C), ~a_ & b_ & c_); }
    else             { l = TK_BITOP3(TK_LO(g.A), TK_LO(g.B), TK_LO(g.C), a_ & b_ & c_);   h = TK_BITOP3(TK_HI(g.A), TK_HI(g.B), TK_HI(g.C), a_ & b_ & c_); }
    return TK_U64(l, h);
}

__device__ __forceinline__ u64 ids_mask(u64 ids, int first, int n) {
    u64 m = 0;
#pragma unroll
    for (int i = 0; i < 6; i++)
        if (i >= first && i < first + n) m |= 1ULL << ((ids >> (6 * i)) & 63);
    return m;
}
__device__ __forceinline__ u64 talon_all(const Game &g) { return ids_mask(g.talon, 0, 6); }

__device__ __forceinline__ bool klop_family(u32 c) { return c == TK_KLOP || c == TK_BERAC || c == TK_ODPRTI_BERAC; }
__device__ __forceinline__ bool has_exchange(u32 c) { return c >= TK_TRI && c <= TK_SOLO_ENA; }
__device__ __forceinline__ bool has_king(u32 c) { return c >= TK_TRI && c <= TK_ENA; }
// odpri_talon's korak (Navadna_igra.py:36-44) for Tri..Solo_ena: 3,2,1,3,2,1
__device__ __forceinline__ u32 group_size(u32 c) { return 3 - ((c - 1) % 3); }

// talon cards that sit in nobody's pile yet
__device__ __forceinline__ u64 talon_unowned(const Game &g) {
    if (g.contract == TK_KLOP) return ids_mask(g.talon, 0, (int)g.tl);
    if (has_exchange(g.contract) && g.tl != 7) {
        u32 gs = group_size(g.contract);
        return talon_all(g) & ~ids_mask(g.talon, (int)(g.tl * gs), (int)gs);
    }
    return talon_all(g);
}

// Roka.prestej (Roka.py:56-98), order independent:
// sum(val) - 2*floor(n/3) - [n%3 != 0].  The values 1..5 are summed as nested
// popcounts (v_bcnt accumulates for free); n <= 54 so n/3 = (n*43)>>7.
__device__ __forceinline__ int prestej(u64 m) {
    // suits = low word (rank r of a suit is bit r-1 of its byte: ranks 5..8 are worth 2..5),
    // taroks = high word (pagat, mond, skis = bits 0, 20, 21 are worth 5, the others 1)
    u32 lo = TK_LO(m), hi = TK_HI(m);
    u32 n = (u32)__popc(lo) + (u32)__popc(hi);
    u32 v = n + (u32)__popc(lo & 0xF0F0F0F0u) + (u32)__popc(lo & 0xE0E0E0E0u) + (u32)__popc(lo & 0xC0C0C0C0u) +
            (u32)__popc(lo & 0x80808080u) + 4u * (u32)__popc(hi & 0x300001u);
    u32 q = (n * 43u) >> 7;
    return (int)(v - 2 * q - (n != 3 * q ? 1u : 0u));
}

// mozne_karte: Navadna_igra.py:158-168; Klop.py:96-133 adds "pagat only when
// nothing else is allowed" (the over-play filter there is dead code).
// The suits are the low word of a plane and the taroks (pagat = bit 0) the high word, so the
// rule is written on the halves: lead-suit cards if any, else taroks if any, else the hand.
// Written on word masks, not on compares and selects: on gfx950 a select costs a lone wave 16.6 cycles through a
// compare (v_cmp, two idle slots, v_cndmask), 28.5 on a compound condition (v_cmp, s_and, v_cndmask) and 12.5 as
// three plain vector instructions (tools/valu_issue: cmpsel / select / arithsel).
// all ones where x == 0, in two plain vector instructions: v_sad_u8 sums the four bytes of x onto -1, the sign
// of that is smeared over the word.  (Written as min(x, 1) or as a sign trick of x itself the compiler turns it
// back into the compare and the select; it does not look into v_sad_u8.)
__device__ __forceinline__ u32 tk_zero_mask(u32 x) { return (u32)((int)__builtin_amdgcn_sad_u8(x, 0u, ~0u) >> 31); }
__device__ __forceinline__ u64 legal_mask(u64 hand, bool has_lead, u32 lead, u32 contract) {
    u32 hl = TK_LO(hand), hh = TK_HI(hand);
    u32 lm = has_lead ? ~0u : 0u;                       // (a compile-time fact in the trick-aligned card loops)
    u32 tarok_led = (u32)((int)(lead << 26) >> 31);     // bit 5 of the id: no suit to follow, taroks next
    u32 s = TK_BITOP3(hl, 0xFFu << (lead & 24), tarok_led, a_ & b_ & ~c_) & lm;
    u32 zs = tk_zero_mask(s);                           // no card of the suit led (or nobody led)
    u32 zt = tk_zero_mask(hh) | ~lm;                    // no taroks (or nobody led)
    u32 bl = s | TK_BITOP3(hl, zs, zt, a_ & b_ & c_);
    u32 bh = hh & zs;
    u32 kf = (0x281u >> contract) & 1u;                 // Klop, Berac, Odprti berac: the pagat only when nothing else goes
    u32 only_pagat = tk_zero_mask(bl | (bh & ~1u));
    return TK_U64(bl, TK_BITOP3(bh, kf, only_pagat, a_ & ~(b_ & ~c_)));
}

__device__ __forceinline__ u64 legal_now(const Game &g) {
    u32 seat = (g.leader + g.nt) & 3;
    return legal_mask(hand_of(g, seat), g.nt != 0, g.trick & 63, g.contract);
}

// observation word (tarok_env.h TAROK_OBS_*) from an already computed legal mask
// (IN_PLAY: the caller knows that the game is not in the DONE phase)
template <bool IN_PLAY = false>
__device__ __forceinline__ u64 obs_word_with(const Game &g, bool finished_now, u64 legal) {
    u64 o = legal;
    o |= (u64)((g.leader + g.nt) & 3) << 54;
    o |= (u64)(g.trick_no * 4 + g.nt) << 56;
    if (IN_PLAY) o |= (u64)(finished_now ? 1u : 0u) << 62;               // (a shift of the caller's 0 / 1, no select)
    else if (finished_now || g.phase == TK_PHASE_DONE) o |= 1ULL << 62;
    o |= (u64)g.error << 63;
    return o;
}

// observation word (tarok_env.h TAROK_OBS_*)
__device__ __forceinline__ u64 obs_word(const Game &g, bool finished_now) {
    bool play = g.phase == TK_PHASE_PLAY;
    u64 o = play ? legal_now(g) : 0;
    o |= (u64)((g.leader + g.nt) & 3) << 54;
    o |= (u64)(g.trick_no * 4 + g.nt) << 56;
    if (finished_now || g.phase == TK_PHASE_DONE) o |= 1ULL << 62;
    o |= (u64)g.error << 63;
    return o;
}

__device__ __forceinline__ u64 pack_scores(int s0, int s1, int s2, int s3) {
    return (u64)(uint16_t)(int16_t)s0 | ((u64)(uint16_t)(int16_t)s1 << 16) |
           ((u64)(uint16_t)(int16_t)s2 << 32) | ((u64)(uint16_t)(int16_t)s3 << 48);
}

// End-of-game scoring, Klop and Navadna/Solo in one straight-line pass (lanes of a
// wave finish different contracts in the same step; two separate routines would
// both be paid by every such wave).
//   Klop (Klop.py:36-45): -points each; if anybody took more than 35 everybody
//     scores 0 (the -70 branch at Klop.py:39-40 is unreachable).
//   Navadna/Solo (Navadna_igra.py:80-113): team pile (+ the talon rest only in the
//     narrow case of :87), v -> sign(contract) + round-to-5 of (v-35).
// Un-owned talon cards sit in an OPPONENT's pile bits from the start (setup_game),
// so the team pile needs no masking; the rest mask is rebuilt only for :87.
__device__ __forceinline__ u64 score_game(const Game &g) {
    bool klop = g.contract == TK_KLOP;
    u64 p0 = pile_of<0>(g), p1 = pile_of<1>(g), p2 = pile_of<2>(g), p3 = pile_of<3>(g);
    // team pile = the taken cards whose owner seat (B_c A_c) is in the team: per card a 4-way
    // select among the team bits, three bitop3 ("a ? b : c") per half
    u32 t0 = (u32)((int)(g.team << 31) >> 31), t1 = (u32)((int)(g.team << 30) >> 31);
    u32 t2 = (u32)((int)(g.team << 29) >> 31), t3 = (u32)((int)(g.team << 28) >> 31);
    u32 xl = TK_BITOP3(TK_LO(g.A), t1, t0, (a_ & b_) | (~a_ & c_)), yl = TK_BITOP3(TK_LO(g.A), t3, t2, (a_ & b_) | (~a_ & c_));
    u32 xh = TK_BITOP3(TK_HI(g.A), t1, t0, (a_ & b_) | (~a_ & c_)), yh = TK_BITOP3(TK_HI(g.A), t3, t2, (a_ & b_) | (~a_ & c_));
    u32 tl = TK_BITOP3(TK_LO(g.B), yl, xl, (a_ & b_) | (~a_ & c_)) & TK_LO(g.C);
    u32 th = TK_BITOP3(TK_HI(g.B), yh, xh, (a_ & b_) | (~a_ & c_)) & TK_HI(g.C);
    u64 t = TK_U64(tl, th);
    // Navadna_igra.py:87: the declarer plays alone (solo, or called a king of his own) and the
    // called king lies in his pile: the kings are bits 7, 15, 23, 31 of the low word
    u32 kb = g.king * 8 + 7, d = g.declarer;
    bool king_taken = ((TK_LO(g.C) >> kb) & 1) && ((TK_LO(g.A) >> kb) & 1) == (d & 1) && ((TK_LO(g.B) >> kb) & 1) == (d >> 1);
    bool alone_with_king = !klop && g.contract != TK_SOLO_BREZ && __popc(g.team) == 1 && has_king(g.contract) && king_taken;
    if (alone_with_king) t |= talon_unowned(g);
    int c0 = prestej(klop ? p0 : t), c1 = prestej(klop ? p1 : 0), c2 = prestej(klop ? p2 : 0), c3 = prestej(klop ? p3 : 0);
    bool over = c0 > 35 || c1 > 35 || c2 > 35 || c3 > 35;
    int d0 = c0 - 35, ad = d0 < 0 ? -d0 : d0;
    int r = 5 * (int)(((u32)(ad + 2) * 205u) >> 10);                     // int(round(d/5))*5, :103
    if (d0 < 0) r = -r;
    int c = 10 * (int)g.contract;
    int sc = (c0 > 35 ? c : -c) + r;
    int s0 = klop ? (over ? 0 : -c0) : ((g.team & 1) ? sc : 0);
    int s1 = klop ? (over ? 0 : -c1) : ((g.team & 2) ? sc : 0);
    int s2 = klop ? (over ? 0 : -c2) : ((g.team & 4) ? sc : 0);
    int s3 = klop ? (over ? 0 : -c3) : ((g.team & 8) ? sc : 0);
    return pack_scores(s0, s1, s2, s3);
}

// Berac / Odprti_berac at its end (Berac.py:33-44): the declarer took the last trick played (he leads next:
// apply_step has already made the winner the leader) -> -70 / -90, otherwise twelve tricks went by -> +70 / +90.
__device__ __forceinline__ u64 berac_scores(const Game &g) {
    int v = g.contract == TK_BERAC ? 70 : 90;
    int sc = g.leader == g.declarer ? -v : v;
    u32 d = g.declarer;
    return pack_scores(d == 0 ? sc : 0, d == 1 ? sc : 0, d == 2 ? sc : 0, d == 3 ? sc : 0);
}

// One card: the body of krog (Klop.py:47-79, Navadna_igra.py:115-141) after
// igraj_karto returned card `a`, then the per-contract loop bookkeeping
// (Klop.py:27-45, Berac.py:26-44, Navadna_igra.py:72-113).
// Returns 0 = played, 1 = played and the game is finished (scores set),
// -1 = not a legal card: nothing changes except the error bit.
// TRUSTED: `a` was drawn from the legal mask by the in-kernel policy, the membership test
// (Klop.py:57-60, Navadna_igra.py:125-126) cannot fail and is skipped.
// DEFER: a finished game's scores are NOT computed here (scores stays untouched): the caller keeps the
// final state and calls final_scores() on it later (k_play scores the finished games of several tricks
// together, on dense lanes).  want_tv = false skips the trick value (nobody asked for trick_info).
// c_lead: where the caller kept the C plane of the moment the trick's first card was played (the trick-aligned
// card loops), the trick's cards are the bits that plane has gained since; otherwise they come from the ids.
template <bool TRUSTED = false, bool DEFER = false>
__device__ __forceinline__ int apply_step(Game &g, u32 a, u64 &scores, u32 &trick_info, bool want_tv = true, const u64 *c_lead = nullptr) {
    if (!TRUSTED) {
        u64 legal = legal_now(g);
        bool ok = a < 54 && ((legal >> (a & 63)) & 1);
        g.error = ok ? g.error : 1u; // (select, not a conditional store: keeps the struct in registers)
        if (!ok) return -1;
    }
    g.C |= 1ULL << a;
    g.trick |= a << (6 * g.nt);
    g.nt++;
    if (g.nt < 4) return 0;
    // pobere_stih / primerjaj_karti (Klop.py:81-94).  The running-best rule there — a challenger wins with a
    // higher card of the best card's suit, or as a tarok against a non-tarok — makes the winner the highest tarok
    // of the trick if one was played, else the highest card of the suit led (a card of any other suit never
    // becomes the best one): read off the trick's card mask instead of three dependent comparisons.
    u64 tm;
    if (c_lead) tm = g.C ^ *c_lead;
    else {
        u32 c0 = g.trick & 63, c1 = (g.trick >> 6) & 63, c2 = (g.trick >> 12) & 63, c3 = (g.trick >> 18) & 63;
        tm = (1ULL << c0) | (1ULL << c1) | (1ULL << c2) | (1ULL << c3);
    }
    u32 th = TK_HI(tm), tl_ = TK_LO(tm) & (0xFFu << (g.trick & 24));     // taroks played; cards of the suit led
    // The winning card is the top bit of the tarok word, or of the suit word when no tarok was played (the card led
    // is a tarok or of its own suit: the chosen word is never zero), and WHO played it is still written in the
    // owner planes at that bit — played cards keep their player's seat bits until the trick changes hands just
    // below.  Word masks, no compares, no selects (see legal_mask).
    u32 zt = tk_zero_mask(th);
    u32 wbit = 31u - (u32)__builtin_clz(TK_BITOP3(th, tl_, zt, (a_ & ~c_) | (b_ & c_)));
    u32 aw = TK_BITOP3(TK_HI(g.A), TK_LO(g.A), zt, (a_ & ~c_) | (b_ & c_)), bw = TK_BITOP3(TK_HI(g.B), TK_LO(g.B), zt, (a_ & ~c_) | (b_ & c_));
    u32 ws = __builtin_amdgcn_ubfe(aw, wbit, 1) | (__builtin_amdgcn_ubfe(bw, wbit, 1) << 1);
    {   // talon gift, Klop.py:67-71: talon.pop() joins the trick after the winner is known
        u32 gift = tk_zero_mask(g.contract) & ((g.tl + 7u) >> 3);         // 1: a Klop with talon cards left
        g.tl -= gift;
        u64 gb = 1ULL << ((u32)(g.talon >> (6 * g.tl)) & 63u);
        u32 gm = 0u - gift;
        tm |= TK_U64(TK_LO(gb) & gm, TK_HI(gb) & gm);
    }
    {   // the trick's cards change owner: plane bit := winner's seat bit where tm is set (one bitop3 per half)
        u32 m1 = (u32)((int)(ws << 31) >> 31), m2 = (u32)((int)(ws << 30) >> 31);
        g.A = TK_U64(TK_BITOP3(TK_LO(g.A), TK_LO(tm), m1, (a_ & ~b_) | (b_ & c_)), TK_BITOP3(TK_HI(g.A), TK_HI(tm), m1, (a_ & ~b_) | (b_ & c_)));
        g.B = TK_U64(TK_BITOP3(TK_LO(g.B), TK_LO(tm), m2, (a_ & ~b_) | (b_ & c_)), TK_BITOP3(TK_HI(g.B), TK_HI(tm), m2, (a_ & ~b_) | (b_ & c_)));
    }
    // what rezultat_stiha(stih, sem_pobral) is told (Klop.py:76-77, Navadna_igra.py:138-139):
    // Roka.vrednost_stiha of the 4 (Klop: 5) cards (Roka.py:76-95) and who took them
    if (want_tv) {
        u32 tv = (u32)(popc64(tm) + popc64(tm & (TK_V2 | TK_V3 | TK_V4 | TK_V5)) + popc64(tm & (TK_V3 | TK_V4 | TK_V5)) +
                       popc64(tm & (TK_V4 | TK_V5)) + popc64(tm & TK_V5)) - 2;
        trick_info = 0x8000u | (tv << 4) | ws;
    }
    g.leader = ws; g.nt = 0; g.trick = 0; g.trick_no++;
    // the game is over after twelve tricks, a Berac also the moment the declarer takes a trick (Berac.py:33-39);
    // straight-line selects: lanes of one wave sit in every combination of these cases
    // (as a 0 / 1 number made of plain vector instructions: the caller's test of it is then ONE compare, which a
    // wave vote can use as it is; a vote on a compound condition goes compare -> scalar and -> select -> compare)
    u32 berac01 = (0x280u >> g.contract) & 1u;                           // TK_BERAC = 7, TK_ODPRTI_BERAC = 9
    u32 over01 = ((g.trick_no + 4u) >> 4) | (berac01 & tk_zero_mask(ws ^ g.declarer));      // (trick_no <= 12)
    if (!DEFER) {
        if (over01) scores = berac01 ? berac_scores(g) : score_game(g);
    }
    g.phase |= over01;                                                   // TK_PHASE_PLAY (2) -> TK_PHASE_DONE (3)
    return (int)over01;
}

// The scores of a game that apply_step has just finished, from its final state alone.
__device__ __forceinline__ u64 final_scores(const Game &g) {
    return (g.contract == TK_BERAC || g.contract == TK_ODPRTI_BERAC) ? berac_scores(g) : score_game(g);
}

// Igra.razdeli's result + engine constructor (Igra.py:38-55,65-73;
// Navadna_igra.py:20-30; Berac.py:15)
__device__ __forceinline__ void setup_game(Game &g, u64 h0, u64 h1, u64 h2, u64 h3, u64 talon36,
                                           u32 contract, u32 declarer, u32 king) {
    g.A = h1 | h3; g.B = h2 | h3;
    g.talon = talon36;
    u64 tal = ids_mask(talon36, 0, 6);
    g.C = tal;
    g.trick = 0; g.nt = 0; g.trick_no = 0; g.error = 0;
    g.contract = contract; g.declarer = declarer;
    g.king = has_king(contract) ? king : 0;
    g.leader = (contract == TK_BERAC || contract == TK_ODPRTI_BERAC) ? declarer : 0;
    if (contract == TK_KLOP) {
        g.team = 0; g.tl = 6;
    } else {
        u32 team = 1u << declarer;
        if (has_king(contract)) {   // partner = holder of the called king BEFORE the exchange
            u64 kb = 1ULL << (king * 8 + 7);
            team |= (h0 & kb) ? 1u : 0u; team |= (h1 & kb) ? 2u : 0u;
            team |= (h2 & kb) ? 4u : 0u; team |= (h3 & kb) ? 8u : 0u;
        }
        g.team = team;
        g.tl = (has_exchange(contract) || contract == TK_SOLO_BREZ) ? 7 : 0;
        // Park the un-owned talon in the pile bits of the lowest seat OUTSIDE the team: whatever
        // of it is never picked up ends with the opponents (Navadna_igra.py:87-92) except in
        // the alone-with-king case, which score_game handles.
        u32 o = (u32)__builtin_ctz(~team & 15u);
        g.A |= (o & 1) ? tal : 0;
        g.B |= (o & 2) ? tal : 0;
    }
    g.phase = has_exchange(contract) ? TK_PHASE_EXCHANGE : TK_PHASE_PLAY;
}

// odpri_talon + menjaj_iz_talona (Navadna_igra.py:36-66, Igralec.py:161-171).
// false = rejected (bad group, card not in hand, duplicate): error bit set.
__device__ __forceinline__ bool apply_exchange(Game &g, u32 choice, u32 d0, u32 d1, u32 d2) {
    u32 gs = group_size(g.contract);
    u32 ngroups = gs == 3 ? 2u : (gs == 2 ? 3u : 6u);
    u64 grp = ids_mask(g.talon, (int)(choice * gs), (int)gs);
    u64 h = hand_of(g, g.declarer) | grp;
    u32 dpk = (d0 & 255) | ((d1 & 255) << 8) | ((d2 & 255) << 16);
    u64 dm = 0;
    bool ok = choice < ngroups;
#pragma unroll
    for (u32 i = 0; i < 3; i++) {
        u32 di = (dpk >> (8 * i)) & 255;
        bool good = di < 54 && ((h >> (di & 63)) & 1) && !((dm >> (di & 63)) & 1);
        ok = ok && (good || i >= gs);
        dm |= (good && i < gs) ? (1ULL << (di & 63)) : 0;
    }
    // all-select update (no conditional stores to different fields: they would push the struct to scratch)
    u32 s = g.declarer;
    u64 nA = (g.A & ~grp) | ((s & 1) ? grp : 0);
    u64 nB = (g.B & ~grp) | ((s & 2) ? grp : 0);
    u64 nC = (g.C & ~grp) | dm;
    g.A = ok ? nA : g.A;
    g.B = ok ? nB : g.B;
    g.C = ok ? nC : g.C;
    g.tl = ok ? choice : g.tl;
    g.phase = ok ? (u32)TK_PHASE_PLAY : g.phase;
    g.error = ok ? g.error : 1u;
    return ok;
}

// ---------------------------------------------------------------------------
// synthetic inputs: counter-based RNG, deal, contract mix, Bot policy
// (the build's own spec; CPU statement in oracle/tarok_spec.py)
// ---------------------------------------------------------------------------
__device__ __forceinline__ u64 mix64(u64 x) {
    x ^= x >> 30; x *= 0xBF58476D1CE4E5B9ULL;
    x ^= x >> 27; x *= 0x94D049BB133111EBULL;
    x ^= x >> 31;
    return x;
}
__device__ __forceinline__ u64 game_key(u64 seed, u64 gidx, u64 episode) {
    u64 a = gidx * 0x9E3779B97F4A7C15ULL + episode * 0xD1B54A32D192ED03ULL + 0x2545F4914F6CDD1DULL;
    return mix64(seed ^ mix64(a));
}
__device__ __forceinline__ u32 rng32(u32 lo, u32 hi, u32 i) {
    u32 x = lo ^ (i * 0x9E3779B1u);
    x ^= x >> 16; x *= 0x85EBCA6Bu;
    x ^= x >> 13; x *= 0xC2B2AE35u;
    x ^= x >> 16; x ^= hi;
    x *= 0x27D4EB2Fu; x ^= x >> 15;
    return x;
}
__device__ __forceinline__ u32 rng32(u64 key, u32 i) { return rng32((u32)key, (u32)(key >> 32), i); }
__device__ __forceinline__ u32 pick(u32 r, u32 n) { return __umulhi(r, n); }

// index of the k-th (0-based) set bit, k < popcount(m): the smallest p with more than k set bits at or below
// it, by bisection on p, one bit of p per level, without a compare or a select (on gfx950 a v_cndmask needs two
// idle slots after the v_cmp that made its mask, and a lone wave pays for them): v_bcnt adds ~k to its count, so
// the sign of the sum says "count <= k", and a funnel shift (v_alignbit) moves that sign into p.
// Per level: v_lshl_or (the candidate), v_bfe (the bits below it), v_bcnt, v_alignbit; 26 instructions in all
// (the bisection that carried the remaining k along and selected on compares: 41).
__device__ __forceinline__ u32 kth_bit(u64 m, u32 k) {
    u32 lo = TK_LO(m), hi = TK_HI(m);
    u32 nk = ~k;                                     // count + nk is negative  <=>  count <= k
    u32 c = (u32)__popc(lo);
    u32 s0 = (u32)((int)(c + nk) >> 31);             // all ones: the bit is in the high word
    u32 w = TK_BITOP3(lo, hi, s0, (a_ & ~c_) | (b_ & c_));
    nk += c & s0;                                    // (~(k - c) = ~k + c)
    u32 p = ((u32)__popc(w & 0xFFFFu) + nk) >> 31;   // p = position / 16
#pragma unroll
    for (int s = 3; s >= 0; s--) {                   // p = position / 2^s after the level
        u32 q = (p << (s + 1)) | (1u << s);
        u32 d = (u32)__popc(__builtin_amdgcn_ubfe(w, 0, q)) + nk;
        p = (p << 1) | (d >> 31);
    }
    return p | (s0 & 32u);
}

// uniform card among the legal ones (Bot_igralec.igraj_karto, Igralec.py:158-159)
__device__ __forceinline__ u32 policy_action(u64 key, u32 step, u64 mask) {
    return kth_bit(mask, pick(rng32(key, 128 + step), (u32)popc64(mask)));
}

// One bidding round between four Bot players (TAROK_MIX_BOT): the control flow of
// Igra.licitacija (Igra.py:75-114) with Bot_igralec.licitiram (Igralec.py:148-152)
// behind the player-side filter (Igralec.py:58-74).  Bids are int(Tip_igre)
// (Naprej -10, Klop 0, Tri 10, Dve 20, Ena 30); the n-th licitiram call of the
// game consumes draw 72 + n.  The re-bidding loop is cut after 8 rounds.
struct BidCtx { u64 key; u32 calls; };
__device__ __forceinline__ int bot_ask(BidCtx &b, int min_igra, int obvezno /* -100 = None */, bool prednost) {
    u32 w = pick(rng32(b.key, 72 + b.calls), 6);
    b.calls++;
    int wish = w < 3 ? -10 : (int)(w - 2) * 10;
    bool ok = prednost ? wish >= min_igra : wish > min_igra;
    return ok ? wish : (obvezno == -100 ? -10 : obvezno);
}
__device__ __forceinline__ void bot_bidding(u64 key, u32 &contract, u32 &declarer) {
    BidCtx b = {key, 0};
    u32 still = 0;
    int top = 10;
#pragma unroll
    for (u32 seat = 1; seat <= 3; seat++) {
        int v = bot_ask(b, top, -100, false);
        if (v != -10) still |= 1u << seat;
        top = max(top, v);
    }
    if (top == 10) {                                 // nobody bid: seat 0 plays at least Klop
        declarer = 0;
        contract = (u32)(bot_ask(b, -10, 0, false) / 10);
        return;
    }
    int v = bot_ask(b, top, -100, true);             // seat 0 may match (priority)
    if (v != -10) still |= 1u;
    top = max(top, v);
    u32 holder = (u32)__builtin_ctz(still);
    for (int rounds = 0; __popc(still) != 1 && rounds < 8; rounds++) {
        u32 nxt = 0;
#pragma unroll
        for (u32 k = 0; k < 4; k++) {
            u32 seat = (k + 1) & 3;                  // 1, 2, 3, 0: seat 0 is asked last
            if ((still >> seat) & 1) {
                v = bot_ask(b, top, seat == holder ? top : -100, false);
                if (v != -10) { nxt |= 1u << seat; holder = seat; top = v; }
            }
        }
        still = nxt;
    }
    declarer = holder;
    contract = (u32)(top / 10);
}

__device__ __forceinline__ void sample_setup(u64 key, int mix, u32 &contract, u32 &declarer, u32 &king) {
    u32 c;
    if (mix == 2) {
        bot_bidding(key, contract, declarer);
        king = has_king(contract) ? pick(rng32(key, 67), 4) : 0;
        return;
    }
    if (mix >= 16) c = (u32)(mix - 16);
    else if (mix == 1) c = TK_TRI + pick(rng32(key, 65), 3);
    else {
        u32 fam = pick(rng32(key, 64), 3), r = rng32(key, 65);
        u32 nav = pick(r, 7);
        u32 navc = nav < 6 ? TK_TRI + nav : TK_SOLO_BREZ;   // Tri,Dve,Ena,Solo_tri,Solo_dve,Solo_ena,Solo_brez
        c = fam == 0 ? TK_KLOP : (fam == 1 ? (pick(r, 2) == 0 ? TK_BERAC : TK_ODPRTI_BERAC) : navc);
    }
    contract = c;
    declarer = c == TK_KLOP ? 0 : pick(rng32(key, 66), 4);
    king = has_king(c) ? pick(rng32(key, 67), 4) : 0;
}

// Bot_igralec.menjaj_iz_talona (Igralec.py:161-171): group 0, random.sample of
// the discardable cards (whole hand if there are too few: the reference raises)
__device__ __forceinline__ void bot_exchange(Game &g, u64 key) {
    u32 gs = group_size(g.contract);
    u64 h = hand_of(g, g.declarer) | ids_mask(g.talon, 0, (int)gs);
    u64 cand = h & TK_DISCARDABLE;
    if ((u32)popc64(cand) < gs) cand = h;
    u32 dpk = 0xFFFFFF;
#pragma unroll
    for (u32 j = 0; j < 3; j++)
        if (j < gs) {
            u32 c = kth_bit(cand, pick(rng32(key, 68 + j), (u32)popc64(cand)));
            dpk = (dpk & ~(255u << (8 * j))) | (c << (8 * j));
            cand &= ~(1ULL << c);
        }
    apply_exchange(g, 0, dpk & 255, (dpk >> 8) & 255, (dpk >> 16) & 255);
}

// The deal: sort the 54 cards by (random key | card id); position p of the
// sorted order is Igra.razdeli's karte[p] (Igra.py:65-73).
// Per-thread form: 54 keys in registers through a fixed sorting network.
__device__ __forceinline__ void deal_thread(u64 key, u64 &h0, u64 &h1, u64 &h2, u64 &h3, u64 &talon36) {
    u32 a[54];
    u32 lo = (u32)key, hi = (u32)(key >> 32);
#pragma unroll
    for (u32 c = 0; c < 54; c++) a[c] = (rng32(lo, hi, c) & 0xFFFFFFC0u) | c;
#define CS(i, j) { u32 x_ = a[i], y_ = a[j]; a[i] = min(x_, y_); a[j] = max(x_, y_); }
#include "deal_network.inc"
#undef CS
    u64 h[4] = {0, 0, 0, 0};
#pragma unroll
    for (int p = 0; p < 48; p++) h[p / 12] |= 1ULL << (a[p] & 63);
    h0 = h[0]; h1 = h[1]; h2 = h[2]; h3 = h[3];
    u64 t = 0;
#pragma unroll
    for (int p = 0; p < 6; p++) t |= (u64)(a[48 + p] & 63) << (6 * p);
    talon36 = t;
}

// Wave-cooperative form of the same deal for ONE game: lane c holds card c,
// its position is its rank among the 54 keys (54 v_readlane broadcasts), the
// hands fall out of four ballots whose bit index IS the card id.  Must be
// called by all 64 lanes of the wave; klo/khi are wave-uniform.  Used where
// only a few lanes of a wave need a new deal (auto-reset inside a step).
__device__ __forceinline__ void deal_wave(u32 klo, u32 khi, u64 &h0, u64 &h1, u64 &h2, u64 &h3, u64 &talon36) {
    u32 lane = __lane_id();
    u32 key = lane < 54 ? ((rng32(klo, khi, lane) & 0xFFFFFFC0u) | lane) : 0xFFFFFFFFu;
    u32 rank = 0;
#pragma unroll
    for (int j = 0; j < 54; j++) {
        u32 kj = (u32)__builtin_amdgcn_readlane((int)key, j);
        rank += kj < key ? 1u : 0u;
    }
    h0 = __ballot(rank < 12);
    h1 = __ballot(rank >= 12 && rank < 24);
    h2 = __ballot(rank >= 24 && rank < 36);
    h3 = __ballot(rank >= 36 && rank < 48);
    u64 t = 0;
#pragma unroll
    for (u32 p = 0; p < 6; p++) {
        u64 b = __ballot(rank == 48 + p);
        t |= (u64)__builtin_ctzll(b) << (6 * p);
    }
    talon36 = t;
}
