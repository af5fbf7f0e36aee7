// libtarokenv: HIP kernels + C ABI (include/tarok_env.h).  gfx950 only.
//
// One thread per game, 256-thread workgroups (4 waves).  Per-game state is the
// 4 packed uint64 lanes of tarok_device.h, stored as two 16-byte SoA arrays
// (s01[g] = play pair {X0,X1}, s23[g] = seat pair {Y0,Y1}): every wave-level
// load/store moves 1 KiB contiguous.  The work is integer mask algebra + popcounts (no MFMA): the
// one-card step (k_step: the state through HBM on every card) is bounded by HBM bandwidth once the batch
// streams, the multi-card Bot-policy kernel (k_play_wide: the state stays in registers) by instruction issue.
//
// Finished games are replaced at once (auto-reset) without a deal on the step's critical path:
// every slot keeps its next FOURTEEN games ready in the lines of its Aux record (episode e lives in
// line e mod 14).  A launch that consumes episode k pushes "deal k+14 into line k mod 14" onto its
// workgroup's refill list; the NEXT launch carries extra workgroups that work those lists off
// (sorting-network deals on dense lanes) while its own play workgroups run — the ~5 us of deal
// latency overlaps the next launch instead of following this one.  The shortest game, a Berac lost on
// trick 1, is 4 cards: with fourteen lines launches of 128 cards (32 tricks; a slot starts ~3.5 games
// in one) practically never wait for a deal (with seven, round 1, 64 cards were the limit: a slot that
// finishes more games within two launches than it has lines deals in place, and the launch is its
// slowest wave).  Lists are double-buffered by launch parity — the low bit of the launch number, which is kept
// modulo 64 in DEVICE memory as counts of started workgroups that every step launch advances by its grid size
// (launch_count / launch_counted / launch_phase; two kinds of launch with two grid sizes, each counted on its own,
// the launch number is the sum), so eager launches and replays of captured graphs (the library's own or a caller's,
// e.g. torch.cuda.graph) mix freely and in any number.  The one-card step of small batches (kind 0) has no extra
// workgroups: every step workgroup works its own group's lists off after playing its card, and it does not deal the
// lines it empties in the next launch at all but collects them for a bulk deal every thirty-second launch
// (refill_role<true>, TAROK_OPT_LAZY_REFILL).  A line is valid iff its episode tag matches
// and it is not being re-dealt right now (`cprev` in the slot's state), and a slot that ever runs
// out of usable lines just deals the game itself, wave-cooperatively (ballot/readlane), same result.
#include "tarok_device.h"

#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <type_traits>
#include <utility>

#include "../../include/tarok_env.h"

#ifndef TK_BLOCK
#define TK_BLOCK 256               // (512 / 1024: diagnostics builds only, tools/ab_build.sh — more play waves per SIMD on fewer CUs)
#endif
#define TK_PF_SLOTS (4 * TK_BLOCK)
#define TK_GRAPH_CACHE 16          // instantiated graphs kept per env (tarok_run_random)
#define TK_REFILL_CAP (TK_BLOCK * TK_AHEAD) // refill-list entries per play workgroup and launch (<= TK_AHEAD per slot)
#define TK_REFILL_FAN 8            // play workgroups whose lists one refill workgroup works off
// list lengths: one 128-byte line per (play workgroup, parity) — neighbouring workgroups run on
// different XCDs, whose L2s are not coherent: two of them must never write into one line
#define TK_RC(group, k) ((((size_t)(group)) * 4 + (k)) * 32)     // k = 0, 1: the per-launch lists by parity; 2, 3: the one-card step's stretch lists
#ifndef TK_BULK_EVERY
#define TK_BULK_EVERY 32u          // one-card launches per stretch (a power of two; see launch_count): the stretch lists are dealt in bulk this often
#endif
#define TK_BULK_CAP ((TK_BULK_EVERY / 4) * TK_BLOCK)  // entries of a stretch list: a game is at least four cards long, one card per launch

#ifndef TK_AHEAD
#define TK_AHEAD TAROK_GAMES_AHEAD  // next-game lines per slot (<= 15: epar and cprev are 4 bits each)
#endif
#define TK_LINE(episode) ((u32)(episode) % (u32)TK_AHEAD)
#define TK_FINQ 128                 // entries of a play wave's finished-games ring (a power of two >= 128)
// The per-card outputs are written once and never read back by the kernels: non-temporal stores, so that
// they stream out during the launch instead of piling up as dirty L2 lines for the write-back at its end
#ifndef TK_NO_STREAM_STORES
#define TK_STREAM_STORE(ptr, val) __builtin_nontemporal_store((val), (ptr))
#else
#define TK_STREAM_STORE(ptr, val) (*(ptr) = (val))
#endif
// s_waitcnt vmcnt(0) (gfx9 encoding: expcnt and lgkmcnt left at their maxima)
#define TK_WAIT_LOADS() __builtin_amdgcn_s_waitcnt(0x0F70)
// wave-uniform branches of the card loop (play_role): a taken branch costs a lone wave ~25 cycles, one that falls
// through ~7 (tools/valu_issue: br_taken / br_not) — the usual path is laid out as the fall-through.
#define TK_RARE(c) __builtin_expect(!!(c), 0)
#define TK_USUAL(c) __builtin_expect(!!(c), 1)

// Per-slot side record: TK_AHEAD (fourteen) 64-byte next-game lines.  Line b holds the dealt-ahead
// game whose episode number is b mod TK_AHEAD: its packed pairs, its RNG key and the episode number
// it is (the validity tag).  A 64-byte AuxLine is HALF an L2 line (128 B on MI355X), and a slot's
// 448-byte record shares its first / last L2 line with a neighbouring slot.  That is safe because
// (i) only refill workgroups ever write these records — play workgroups read them and write
// nothing here (their Counters / state / gkey / refill lists live in lines of their own, see
// Counters below), (ii) all lines of one play group's 256 slots (a 112 KiB, 128-byte aligned
// range) are re-dealt by ONE refill workgroup per launch, so no L2 line is written by two
// workgroups (two XCDs) within a launch, and (iii) a reader never trusts a line whose episode tag
// does not match or that `cprev` says is being re-dealt right now.
struct __attribute__((aligned(64))) AuxLine {
    ulonglong2 n01, n23; u64 nkey; u32 nep;
    u32 pad[5];
};
struct __attribute__((aligned(64))) Aux { AuxLine line[TK_AHEAD]; };
static_assert(sizeof(AuxLine) == 64 && sizeof(Aux) == 64 * TK_AHEAD, "64 bytes per next-game line");
static_assert((TK_BLOCK * sizeof(Aux)) % 128 == 0, "a play group's next-game lines end on an L2 line boundary");
static_assert(TK_AHEAD >= 2 && TK_AHEAD <= 15, "epar / cprev are 4-bit fields");
// What a finishing game always touches: the slot's episode number and its summed scores.  Kept
// apart from the next-game lines: those are written by refill workgroups, these by the slot's own
// play workgroup, which runs on another XCD (another, non-coherent L2) — the two must not share
// a cache line.
struct __attribute__((aligned(32))) Counters {
    int4 score_sum;                                  // summed scores by seat (Tarok.rezultati)
    u32 episode;                                     // episode number of the slot's current game
    u32 pad[3];
};
static_assert(sizeof(Counters) == 32, "Counters is half a cache line");

struct tarok_env {
    int device;
    int64_t n;
    u64 offset, seed;
    int mix, flags;
    ulonglong2 *s01, *s23;   // packed state
    Aux *aux;                // next-game lines per slot
    Counters *cnt;           // episode number and score sums per slot
    uint16_t *nstale;        // bit k: the game k+1 ahead is missing and not on any refill list
                             // (after tarok_reset; what k_prefetch scans); padded to 1024 slots
    u64 *gkey;               // RNG key of the slot's current game
    uint8_t *hist;           // [48][n] play history (card p of the slot's current game), TAROK_HISTORY envs only
    u64 *rlist;              // refill lists [play workgroups][2 parities][TK_REFILL_CAP]: episode<<32 | slot in group
    u32 *rcount;             // [play workgroups][4] list lengths, one 128-byte line each (TK_RC)
    u64 *elist;              // the one-card step's stretch lists [play workgroups][2][TK_BULK_CAP] (refill_role)
    u32 *epoch;              // 2 x TK_EPOCH_SHARDS counters, 128 bytes apart: workgroups of step launches of either kind started so
                             // far (launch_count): the parity of the refill list the running launch writes (it works the other one off)
    uint32_t refill_fan;     // play workgroups per refill workgroup (1..TK_REFILL_FAN)
    uint32_t lazy_refill;    // the one-card step lists emptied lines for a bulk deal every TK_BULK_EVERY launches (refill_role)
    int n_cus;               // compute units of the device (k_learn_dw's grid), 0 = not asked yet
    float *adam_sumsq;       // k_learn_gnorm's partial sums
    u64 *stamps;             // diagnostics only (tarok_debug_stamps)
    size_t stamps_words;     // its capacity: a kernel that would write more gets no stamps pointer
    int launched;            // a step launch has been issued (tarok_set_option: a change of the grid must restart the launch counters)
    hipStream_t cap_stream;  // capture-only stream for tarok_run_random's graphs
    // tarok_run_random's instantiated graphs.  An exec is NEVER destroyed while launches of it may still be queued (round 3
    // destroyed the one cached exec whenever the segment kind changed, with up to a hundred of its launches in flight):
    // every (launch kind, chunk, flags, buffers, launch tuning) keeps its exec until tarok_destroy, which synchronises the
    // device first; a full cache is emptied behind a device synchronisation.
    struct GraphEntry {
        hipGraphExec_t exec;
        int fused, chunk, flags, prefetch;
        void *action, *reward, *done, *obs, *stamps;
        uint32_t fan, lazy;
    } graphs[TK_GRAPH_CACHE];
    int n_graphs;
};

static thread_local int g_last_hip = 0;

#define HIPCHK(x)                                                   \
    do {                                                            \
        hipError_t e_ = (x);                                        \
        if (e_ != hipSuccess) { g_last_hip = (int)e_; return TAROK_EHIP; } \
    } while (0)

static inline dim3 grid_for(int64_t n) { return dim3((unsigned)((n + TK_BLOCK - 1) / TK_BLOCK)); }

// ---------------------------------------------------------------------------
// kernels
// ---------------------------------------------------------------------------
__device__ __forceinline__ void load_game(Game &g, const ulonglong2 *s01, const ulonglong2 *s23, int64_t i) {
    ulonglong2 a = s01[i], b = s23[i];
    unpack(g, a.x, a.y, b.x, b.y);
}
// the play pair always; the seat pair only when the A/B planes (or the setup fields) changed
__device__ __forceinline__ void store_game(const Game &g, ulonglong2 *s01, ulonglong2 *s23, int64_t i,
                                           bool seats_changed = true) {
    ulonglong2 a;
    pack_play(g, a.x, a.y);
    s01[i] = a;
    if (seats_changed) {
        ulonglong2 b;
        pack_seats(g, b.x, b.y);
        s23[i] = b;
    }
}

// Igra.razdeli + engine construction + talon exchange for every slot.
TK_KERNEL(TK_BLOCK, 80) void k_reset(
    int64_t n, u64 seed, u64 offset, u32 episode, int mix, int flags,
    const uint8_t *__restrict__ deals, const int8_t *__restrict__ contract, const int8_t *__restrict__ declarer,
    const int8_t *__restrict__ king, const int8_t *__restrict__ choice, const uint8_t *__restrict__ discards,
    ulonglong2 *__restrict__ s01, ulonglong2 *__restrict__ s23, Aux *aux, Counters *__restrict__ cnt,
    uint16_t *__restrict__ nstale, u64 *__restrict__ gkey) {
    TK_VGPR_TOP(80, 79);
    int64_t i = (int64_t)blockIdx.x * TK_BLOCK + threadIdx.x;
    if (i >= n) return;
    u64 key = game_key(seed, offset + (u64)i, episode);
    u64 h0 = 0, h1 = 0, h2 = 0, h3 = 0, tal = 0;
    bool bad = false;
#pragma unroll
    for (int b = 0; b < TK_AHEAD; b++) aux[i].line[b].nep = 0xFFFFFFFFu;   // all next-game lines: empty
    nstale[i] = (uint16_t)((1u << TK_AHEAD) - 1);
    if (deals) {
        const uint8_t *p = deals + i * 54;
        u64 h[4] = {0, 0, 0, 0};
        u64 seen = 0;
        for (int k = 0; k < 48; k++) { u32 c = p[k]; bad |= c >= 54; c &= 63; seen |= 1ULL << c; h[k / 12] |= 1ULL << c; }
        for (int k = 0; k < 6; k++) { u32 c = p[48 + k]; bad |= c >= 54; c &= 63; seen |= 1ULL << c; tal |= (u64)c << (6 * k); }
        bad |= seen != TK_DECK;
        h0 = h[0]; h1 = h[1]; h2 = h[2]; h3 = h[3];
    } else {
        deal_thread(key, h0, h1, h2, h3, tal);
    }
    u32 c, d, k;
    if (contract) {
        int ci = contract[i];
        bad |= ci < 0 || ci > 9;
        c = (u32)ci % 10u;
        d = declarer ? ((u32)declarer[i] & 3u) : 0u;
        k = king ? ((u32)king[i] & 3u) : 0u;
    } else {
        sample_setup(key, mix, c, d, k);
    }
    Game g;
    setup_game(g, h0, h1, h2, h3, tal, c, d, k);
    g.epar = TK_LINE(episode); g.cprev = 0;
    if (g.phase == TK_PHASE_EXCHANGE && !(flags & TAROK_DEFER_EXCHANGE)) {
        if (choice && discards) {
            const uint8_t *q = discards + i * 3;
            apply_exchange(g, (u32)(uint8_t)choice[i], q[0], q[1], q[2]);
        } else {
            bot_exchange(g, key);
        }
    }
    if (bad) g.error = 1;
    store_game(g, s01, s23, i);
    gkey[i] = key;
    cnt[i].episode = episode;
    if (flags & TAROK_CLEAR_COUNTERS) cnt[i].score_sum = make_int4(0, 0, 0, 0);
}

// Deal game `episode` of slot j ahead of time into its line (episode mod TK_AHEAD).
__device__ __forceinline__ void deal_into_buffer(Aux *aux, int64_t j, u32 episode, u64 seed, u64 offset, int mix) {
    u64 key = game_key(seed, offset + (u64)j, (u64)episode);
    u64 h0, h1, h2, h3, tal;
    deal_thread(key, h0, h1, h2, h3, tal);
    u32 c, d, k;
    sample_setup(key, mix, c, d, k);
    Game g;
    setup_game(g, h0, h1, h2, h3, tal, c, d, k);
    g.epar = TK_LINE(episode); g.cprev = 0;
    if (g.phase == TK_PHASE_EXCHANGE) bot_exchange(g, key);
    ulonglong2 a, b;
    pack(g, a.x, a.y, b.x, b.y);
    AuxLine *ln = &aux[j].line[TK_LINE(episode)];
    ln->n01 = a; ln->n23 = b; ln->nkey = key; ln->nep = episode;
}

// tarok_prefetch: fill, synchronously, the next-game lines that tarok_reset emptied (flags in
// nstale: bit k = episode+1+k missing).  Each workgroup compacts the missing lines of its
// 1024-slot tile into an LDS list (4 flag bytes per thread) and deals list entry j on thread j,
// so the sorting network runs on dense lanes.
TK_KERNEL(TK_BLOCK, 128) void k_prefetch(int64_t n, u64 seed, u64 offset, int mix,
                                                      Aux *aux, const Counters *__restrict__ cnt,
                                                      uint16_t *__restrict__ nstale) {
    TK_VGPR_TOP(128, 127);
    __shared__ unsigned short list[TK_AHEAD * TK_PF_SLOTS];
    __shared__ u32 count;
    int64_t base = (int64_t)blockIdx.x * TK_PF_SLOTS;
    if (threadIdx.x == 0) count = 0;
    __syncthreads();
    const u64 ALL = (1u << TK_AHEAD) - 1;
    u64 f = reinterpret_cast<const u64 *>(nstale + base)[threadIdx.x];   // 4 slots x 16 flag bits; the array is padded
    if (f) {
        u32 c = (u32)__popcll(f & (0x0001000100010001ULL * ALL));
        u32 pos = atomicAdd(&count, c);
#pragma unroll
        for (u32 k = 0; k < 4; k++) {
            u32 fk = (u32)(f >> (16 * k)) & (u32)ALL;
#pragma unroll
            for (u32 b = 0; b < TK_AHEAD; b++)
                if (fk & (1u << b)) list[pos++] = (unsigned short)((threadIdx.x * 4 + k) * TK_AHEAD + b);
        }
        reinterpret_cast<u64 *>(nstale + base)[threadIdx.x] = 0;
    }
    __syncthreads();
    u32 total = count;
    for (u32 j = threadIdx.x; j < total; j += TK_BLOCK) {
        int64_t i = base + list[j] / TK_AHEAD;
        if (i >= n) continue;
        deal_into_buffer(aux, i, cnt[i].episode + 1 + (list[j] % TK_AHEAD), seed, offset, mix);
    }
}

TK_KERNEL(TK_BLOCK, 64) void k_exchange(int64_t n, const int8_t *__restrict__ choice,
                                                      const uint8_t *__restrict__ discards,
                                                      ulonglong2 *__restrict__ s01, ulonglong2 *__restrict__ s23,
                                                      const u64 *__restrict__ gkey) {
    TK_VGPR_TOP(64, 63);
    int64_t i = (int64_t)blockIdx.x * TK_BLOCK + threadIdx.x;
    if (i >= n) return;
    Game g;
    load_game(g, s01, s23, i);
    if (g.phase != TK_PHASE_EXCHANGE) return;
    if (choice && discards) {
        const uint8_t *q = discards + i * 3;
        apply_exchange(g, (u32)(uint8_t)choice[i], q[0], q[1], q[2]);
    } else {
        bot_exchange(g, gkey[i]);
    }
    store_game(g, s01, s23, i);
}

TK_KERNEL(TK_BLOCK, 64) void k_legal(int64_t n, const ulonglong2 *__restrict__ s01,
                                                   const ulonglong2 *__restrict__ s23, u64 *__restrict__ obs,
                                                   int8_t *__restrict__ seat) {
    TK_VGPR_TOP(64, 63);
    int64_t i = (int64_t)blockIdx.x * TK_BLOCK + threadIdx.x;
    if (i >= n) return;
    Game g;
    load_game(g, s01, s23, i);
    obs[i] = obs_word(g, false);
    if (seat) seat[i] = (int8_t)((g.leader + g.nt) & 3);
}

TK_KERNEL(TK_BLOCK, 64) void k_policy(int64_t n, const u64 *__restrict__ obs,
                                                    const u64 *__restrict__ gkey, uint8_t *__restrict__ action) {
    TK_VGPR_TOP(64, 63);
    int64_t i = (int64_t)blockIdx.x * TK_BLOCK + threadIdx.x;
    if (i >= n) return;
    u64 o = obs[i], key = gkey[i];     // (the key asked for WITH the observation word, not behind it: one memory round trip)
    u64 m = o & TAROK_OBS_MASK;
    u32 a = 255;
    if (m) a = policy_action(key, (u32)(o >> TAROK_OBS_STEP_SHIFT) & 63u, m);
    action[i] = (uint8_t)a;
}

// The same for four games per thread, the form the host picks for batches that stream (launch_policy): a lane of the
// one-game kernel has 16 bytes in flight, the chip 8 MB, and at 4 M games the launch ran at the memory LATENCY
// (16.3 us for 71 MB, profiles/r03_step_durations.txt; this form 13.0 us).  A workgroup takes 1,024 consecutive games; thread t the
// pairs (2t, 2t+1) of its first and of its second half, so that every load is a contiguous 16 bytes per lane and
// every store two bytes per lane of one 128-byte line per wave.
TK_KERNEL(TK_BLOCK, 64) void k_policy_x4(int64_t n, const u64 *__restrict__ obs,
                                                       const u64 *__restrict__ gkey, uint8_t *__restrict__ action) {
    TK_VGPR_TOP(64, 63);
    int64_t base = (int64_t)blockIdx.x * (4 * TK_BLOCK);
    if (base + 4 * TK_BLOCK <= n) {
        int64_t p0 = base + 2 * threadIdx.x, p1 = p0 + 2 * TK_BLOCK;
        ulonglong2 o0 = *reinterpret_cast<const ulonglong2 *>(obs + p0), o1 = *reinterpret_cast<const ulonglong2 *>(obs + p1);
        ulonglong2 k0 = *reinterpret_cast<const ulonglong2 *>(gkey + p0), k1 = *reinterpret_cast<const ulonglong2 *>(gkey + p1);
        auto one = [](u64 o, u64 key) __attribute__((always_inline)) {
            u64 m = o & TAROK_OBS_MASK;
            return m ? policy_action(key, (u32)(o >> TAROK_OBS_STEP_SHIFT) & 63u, m) : 255u;
        };
        u32 a0 = one(o0.x, k0.x) | (one(o0.y, k0.y) << 8), a1 = one(o1.x, k1.x) | (one(o1.y, k1.y) << 8);
        *reinterpret_cast<uint16_t *>(action + p0) = (uint16_t)a0;
        *reinterpret_cast<uint16_t *>(action + p1) = (uint16_t)a1;
        return;
    }
    for (int64_t i = base + threadIdx.x; i < n; i += TK_BLOCK) {       // the last, partial workgroup
        u64 o = obs[i];
        u64 m = o & TAROK_OBS_MASK;
        u32 a = 255;
        if (m) a = policy_action(gkey[i], (u32)(o >> TAROK_OBS_STEP_SHIFT) & 63u, m);
        action[i] = (uint8_t)a;
    }
}

// WHAT WORKGROUPS HAND EACH OTHER, AND THROUGH WHICH CACHE (DESIGN.md §3 has the table).  Five buffers are written by one
// workgroup and read by another: the next-game lines (`aux`: refill role -> play / step role), the per-launch refill lists
// and their lengths (`rlist`, `rcount[0..1]`: play / step role -> refill role), the stretch lists and their lengths (`elist`,
// `rcount[2..3]`: step role -> refill role and, the lengths, back to the step role of a LATER launch) and the launch counters
// (`epoch`).  Except for `epoch` every such word is read in a LATER launch than it was written: the kernel boundary (release at
// the end of a launch, acquire at the start of the next: L2 write-back and invalidate across the XCDs, vector L1 and scalar
// cache invalidated) is the only ordering the protocol relies on, and within a launch no workgroup reads a word another one
// writes in that launch.  The code says so: none of these pointers is `__restrict__` or `const` (nothing licenses the
// compiler to keep such a word in the scalar cache or to merge its reads across a barrier), every length and list entry is
// read with an agent-scope atomic load (tk_ld: a vector load that misses the L1), and `epoch` — the one buffer read and
// written within a launch — only with agent-scope atomics.
template <class T> __device__ __forceinline__ T tk_ld(const T *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

// Launch number modulo TK_PHASES (see the file header) without a host counter.  The workgroups of step launches are
// counted as they START, modulo TK_PHASES times their number G per launch: launch L begins with the count at
// (L mod TK_PHASES) * G; every workgroup reads the count together with its first loads (launch_count: nothing waits for
// it alone) and adds itself afterwards (launch_counted: a wrapping increment WITHOUT a return value, behind the read in
// the lane's program order — it completes somewhere under the card loop).  At any read at most G - 1 workgroups of the
// running launch have added themselves, so the count is still in launch L's band: phase = count / G.  The kernel boundary
// drains the adds before the next launch reads.  The count is kept in TK_EPOCH_SHARDS separate counters (one 128-byte
// line each), workgroup b using counter b mod TK_EPOCH_SHARDS with G_s = the number of such workgroups: the argument
// holds for every counter on its own, and the adds do not queue up behind each other — ONE counter took 512 same-address
// atomics per launch at 65,536 games, serialised at the memory side: +3.4 us on every launch, wherever in the kernel they
// were issued (profiles/r02_ab_launch_parity.txt).
// An env has step launches of TWO grid sizes (round 4): KIND 0, the one-card step of small batches (k_step<., true>):
// `groups` workgroups, each of which plays its group's card and then works its OWN group's refill lists off — at 65,536
// games the 256 refill workgroups of the other kind, idle in fifteen launches of sixteen, cost every launch ~0.9 us
// (profiles/r04_ab_step.txt); KIND 1, every other step launch (k_play_wide, k_policy_step, the streaming k_step):
// groups + ceil(groups / fan) workgroups, play and refill roles side by side.  Each kind counts ITS launches in its own
// set of counters as above, and a workgroup of kind k reads, beside its own counter, one counter of the other kind — at
// rest during this launch: exactly (launches of that kind mod TK_PHASES) * its shard size.  The launch number is the sum.
// The low bit of the phase is the parity of the refill list the launch writes; the one-card step also works the
// next-game lines its slots emptied off in bulk, once per TK_BULK_EVERY launches (step_role, refill_role).
#define TK_EPOCH_SHARDS 256
#define TK_EPOCH_WORDS (32 * TK_EPOCH_SHARDS)      // u32 words of one kind's counters (one 128-byte line per counter)
#define TK_PHASES (2u * TK_BULK_EVERY)
static_assert((TK_BULK_EVERY & (TK_BULK_EVERY - 1)) == 0 && TK_BULK_EVERY >= 4 && TK_BULK_EVERY + 1 <= 4 * (TK_AHEAD - 1),
              "a line is dealt at most TK_BULK_EVERY + 1 launches after it was emptied and is needed again TK_AHEAD - 1 games (four launches each, at least) after that");
struct TkCount { u32 own, other; };
__device__ __forceinline__ u32 tk_shard_size(u32 grid, u32 shard) { return (grid + TK_EPOCH_SHARDS - 1 - shard) / TK_EPOCH_SHARDS; }
// the other kind's grid, and the counter of it this workgroup reads (one that exists: kind 0's grid is the smaller one)
template <int KIND> __device__ __forceinline__ u32 tk_other_grid(u32 groups, u32 fan) { return KIND == 0 ? groups + (groups + fan - 1) / fan : groups; }
template <int KIND> __device__ __forceinline__ u32 tk_other_shard(u32 groups) { return (KIND == 0 ? blockIdx.x : blockIdx.x % groups) % TK_EPOCH_SHARDS; }
template <int KIND> __device__ __forceinline__ TkCount launch_count(const u32 *epoch, u32 groups) {
    TkCount c;
    c.own = tk_ld(epoch + KIND * TK_EPOCH_WORDS + 32 * (blockIdx.x % TK_EPOCH_SHARDS));
    c.other = tk_ld(epoch + (1 - KIND) * TK_EPOCH_WORDS + 32 * tk_other_shard<KIND>(groups));
    return c;
}
// count / G_s: a float estimate (count < TK_PHASES * G_s is exact as a float for any grid a launch can have; the half added
// to the count keeps the quotient away from the integers) put right by one exact integer step either way
__device__ __forceinline__ u32 tk_div_small(u32 count, u32 g) {
    u32 q = (u32)(((float)count + 0.5f) * __builtin_amdgcn_rcpf((float)g));
    q -= (q * g > count) ? 1u : 0u;
    q += ((q + 1u) * g <= count) ? 1u : 0u;
    return q;
}
template <int KIND> __device__ __forceinline__ u32 launch_phase(TkCount c, u32 groups, u32 fan) {
    // (the counts are the same in every lane, and the compiler knows: without this fence it moves the quotients — and the wait
    // for the counts' loads — to the top of the kernel, ahead of the state loads: +4 us per launch at 4 M games)
    asm volatile("" : "+v"(c.own), "+v"(c.other));
    u32 own = tk_div_small(c.own, tk_shard_size(gridDim.x, blockIdx.x % TK_EPOCH_SHARDS));
    u32 other = tk_div_small(c.other, tk_shard_size(tk_other_grid<KIND>(groups, fan), tk_other_shard<KIND>(groups)));
    return (own + other) % TK_PHASES;
}
// call after launch_count in program order — one lane's read and add of one address stay in that order — behind a barrier
// that every wave of the workgroup passes after ITS read of the count (were this the last workgroup of its counter to add,
// a wave that read after the add would see the next launch's band), and before the first use of the count's value: the add
// counts as an outstanding memory operation of the wave, and issued only once the count has arrived it adds a memory-side
// round trip to the life of every step wave (+4 us per launch at 4 M games)
template <int KIND> __device__ __forceinline__ void launch_counted(u32 *epoch) {
    if (threadIdx.x == 0)
        (void)atomicInc(epoch + KIND * TK_EPOCH_WORDS + 32 * (blockIdx.x % TK_EPOCH_SHARDS),
                        TK_PHASES * tk_shard_size(gridDim.x, blockIdx.x % TK_EPOCH_SHARDS) - 1u);
}

// The step kernels.  One launch plays `cards` cards of every game:
//   k_step (step_role), cards = 1: tarok_step — the card comes from `action_in` (an external policy) — and
//                   tarok_step_random (the Bot policy, Igralec.py:158-159, evaluated in-kernel);
//   k_play_wide (play_role), cards >= 2, Bot policy in-kernel: cards = 4 is one whole trick = one pass of the
//                   reference's krog generator (Klop.py:47-79, Navadna_igra.py:115-141).
// The packed state is read once, stays in registers while the cards are played and is written
// once; everything a consumer of the trajectory needs is written for EVERY card: row c of
// action/obs/done/trick/reward (rows `stride` games apart) belongs to the c-th card of the launch.
// Per card: legal mask -> card -> apply -> (4th card of a trick) winner, Klop talon gift, Berac
// end, end-of-game scoring, and with TAROK_AUTO_RESET the finished game's successor swapped in
// from the slot's next-game buffer.
//
// Workgroups [0, play_groups) play; the workgroups after them work off the refill lists the
// PREVIOUS launch wrote (see the file header).
// Refill role of a step launch: work off the lists the PREVIOUS launch wrote for `fan` play
// workgroups (rblock = index among the refill workgroups; tid / nthreads = this thread in its
// workgroup), concatenated so that the sorting-network deals run on dense lanes (~11 % of the
// slots of a group finish per trick: 8 lists fill a 256-thread workgroup).  Small batches use a
// smaller fan: a refill workgroup that needs a second pass would outlast the play.
// `count` = launch_count()'s value, possibly still in flight: the list lengths of BOTH parities are requested
// before it is looked at, so the parity costs this role no memory round trip of its own.
// BULK (k_step): the lines the one-card step's slots emptied are NOT on those lists.  Every slot is at the same card of
// its trick and games end on a trick's 4th card, so they would all be dealt in the launch after it, whose refill
// workgroups (a wave alone issues an instruction every ~4.5 cycles, a deal is 2.7k of them) outlast the step
// workgroups: 9.9 us against 4.3 at 65,536 games, 79 against 44 at 4 M (profiles/r03_step_durations.txt,
// r03_ab_step.txt (e)).  A slot has fourteen lines and takes at most eight in thirty-two launches, so step_role collects
// those entries in a second pair of lists per workgroup, switched every TK_BULK_EVERY launches, and the first launch
// of each such stretch works the previous stretch's list off here, on dense lanes, in one go.  A line waits for its
// deal at most TK_BULK_EVERY launches and is not needed again before thirteen more games of its slot have
// ended (>= 52 launches); lines a slot must not meet half written — the fourteen of a slot that dealt a game in place —
// stay on the per-launch lists.  A BULK launch is of kind 0 (launch_count): it has no refill workgroups — the workgroup
// that has just played its group's card runs this role for that group alone (fan 1, rblock = its group), so the count
// has been read and counted by the step role already.  !BULK (kind 1: k_play_wide, k_policy_step, the streaming k_step;
// slots there can take a line per trick): the stretch lists of the workgroup's groups are emptied unworked — those lines
// stay stale until their slot comes round to them, deals that game in place and lists all fourteen.
template <bool BULK>
__device__ __forceinline__ void refill_role(u32 rblock, u32 tid, u32 nthreads, u64 seed, u64 offset, int mix, u32 play_groups,
                                        TkCount count, u32 *epoch, u32 fan, bool bulk_on, Aux *aux, u64 *rlist, u32 *rcount, u64 *elist) {
    constexpr int KIND = BULK ? 0 : 1;
    const u32 kind_fan = fan;            // (the env's fan: what the other kind's grid is made of)
    if (BULK) fan = 1;                   // the workgroup's own group
    u32 g0 = rblock * fan;
    // every thread loads the two per-launch lengths of every group itself, the stretch lists' lengths only in the pass that
    // works them off.  (Round 3 shipped this form because shorter ones — all four lengths with one load per wave, sums in
    // scalar registers — "failed for reasons not understood".  The reason is understood now and had nothing to do with the
    // lengths: those builds had 104 VGPRs with a shift amount of the deal in v103, and gfx950 mis-executes a 64-bit shift
    // whose amount sits in the last allocated VGPR — TK_VGPR_TOP, tarok_device.h; profiles/r04_refill_root_cause.txt.)
    u32 len0[TK_REFILL_FAN], len1[TK_REFILL_FAN];
#pragma unroll
    for (u32 q = 0; q < TK_REFILL_FAN; q++) {
        bool has = q < fan && g0 + q < play_groups;
        len0[q] = has ? tk_ld(&rcount[TK_RC(g0 + q, 0)]) : 0u;
        len1[q] = has ? tk_ld(&rcount[TK_RC(g0 + q, 1)]) : 0u;
    }
    // The workgroup's add must not overtake the count reads of its OWN later waves (every thread reads the count at the top
    // of the kernel; were this the last workgroup of its counter to add, a wave that read after the add would see the next
    // launch's band: the wrong parity, the list the running launch is writing).  The barrier puts every wave's read into the
    // memory pipe ahead of the add; step_role and play_role add behind their first barrier as well.
    if (!BULK) {
        __syncthreads();
        launch_counted<KIND>(epoch);
    }
    const u32 phase = launch_phase<KIND>(count, play_groups, kind_fan), par = phase & 1u;
    if (!BULK) {                 // empty the stretch lists
        if (tid < 2 * fan && g0 + tid / 2 < play_groups) rcount[TK_RC(g0 + tid / 2, 2 + (tid & 1))] = 0u;
    }
    u32 cum[TK_REFILL_FAN + 1];
    cum[0] = 0;
    u32 which = par ^ 1u;
    const u32 odd = 0u - par;    // (a mask, not a select between the arrays: that becomes a parity-indexed array in scratch)
#pragma unroll
    for (u32 q = 0; q < TK_REFILL_FAN; q++) cum[q + 1] = cum[q] + ((len0[q] & odd) | (len1[q] & ~odd));
    const bool any = cum[TK_REFILL_FAN] != 0;                 // (the same in every thread: nobody writes these lengths in this launch)
    // the lists of the workgroup's groups, one after the other, entry j on thread j mod nthreads; BULK: a second pass (the
    // same code: one copy of the deal) over the stretch lists in the launches that work them off
    u64 *lists = rlist;
    u32 cap = TK_REFILL_CAP;
    const bool bulk = BULK && bulk_on && phase % TK_BULK_EVERY == 0;      // (bulk_on: the env's step workgroups fill stretch lists)
#pragma nounroll
    for (u32 pass = 0; pass < (BULK ? 2u : 1u); pass++) {
        if (pass == 1) {
            if (!bulk) break;
            which = ((phase / TK_BULK_EVERY) & 1u) ^ 1u;     // the stretch before this one
            lists = elist; cap = TK_BULK_CAP;
#pragma unroll
            for (u32 q = 0; q < TK_REFILL_FAN; q++) {
                bool has = q < fan && g0 + q < play_groups;
                cum[q + 1] = cum[q] + (has ? min(tk_ld(&rcount[TK_RC(g0 + q, 2 + which)]), (u32)TK_BULK_CAP) : 0u);
            }
        }
        for (u32 j = tid; j < cum[TK_REFILL_FAN]; j += nthreads) {
            u32 q = 0;
#pragma unroll
            for (u32 r = 1; r < TK_REFILL_FAN; r++) q += j >= cum[r] ? 1u : 0u;
            u32 base = 0;
#pragma unroll
            for (u32 r = 0; r < TK_REFILL_FAN; r++) base = (r == q) ? cum[r] : base;
            u64 en = tk_ld(&lists[((int64_t)(g0 + q) * 2 + which) * cap + (j - base)]);
            deal_into_buffer(aux, (int64_t)(g0 + q) * TK_BLOCK + (u32)(en & 0xFFFF), (u32)(en >> 32), seed, offset, mix);
        }
    }
    // the lengths worked off are cleared (the one-card step's workgroups write a length only when they list something);
    // every thread has had its copies (cum[]) before the barrier
    if (any || bulk) {
        __syncthreads();
        if (any && tid < fan && g0 + tid < play_groups) rcount[TK_RC(g0 + tid, par ^ 1u)] = 0u;
        if (bulk && tid < fan && g0 + tid < play_groups) rcount[TK_RC(g0 + tid, 2 + which)] = 0u;     // (which: the stretch list's, after the second pass)
    }
}

// Play role of a Bot-policy launch (tarok_krog_random: `cards` cards per launch, the card from the in-kernel Bot
// policy, Igralec.py:158-159) for the 256 slots of play workgroup `group`: thread `tid` (0..255) plays slot
// group * 256 + tid.  (One card per launch from an external policy: step_role below.)
// HIST: also record the play history (one byte per card; tarok_create flag TAROK_HISTORY).
template <bool HIST>
__device__ __forceinline__ void play_role(
    u32 group, u32 tid,
    int64_t n, u64 seed, u64 offset, int mix, int flags, int cards, int64_t stride, TkCount count, u32 *epoch, u32 play_groups, u32 fan,
    uint8_t *__restrict__ action_out, int16_t *__restrict__ reward,
    uint8_t *__restrict__ done, uint16_t *__restrict__ trick, u64 *__restrict__ obs, uint8_t *__restrict__ hist,
    ulonglong2 *__restrict__ s01, ulonglong2 *__restrict__ s23, Aux *aux, Counters *__restrict__ cnt,
    u64 *__restrict__ gkey, u64 *rlist, u32 *rcount, u64 *__restrict__ stamps) {
    __shared__ unsigned short push_list[TK_REFILL_CAP];   // (how far ahead) * TK_BLOCK + slot in group
    __shared__ u32 push_ep[TK_BLOCK];                     // the slot's episode number at the end of the launch
    __shared__ u32 push_count;
    // Deferred scoring (the multi-card kernel): a game that ends leaves its final state in a per-wave LDS ring
    // (9 dwords) instead of being scored on the spot — with ~10 % of the slots finishing per trick the
    // scoring code (both contract families, ~250 instructions) ran for every wave on every trick with 7 of 64
    // lanes active: a third of a play wave's time at 65,536 games (tools/card_probe.py).  Whenever 64 entries
    // wait they are scored in ONE pass on full lanes (drain_finished); the rest at the end of the launch.  The
    // scores go to their reward row from there and are summed per slot in LDS (sacc) for the slot's score_sum.
    __shared__ u32 finq[TK_BLOCK / 64][9][TK_FINQ];
    __shared__ int sacc[4][TK_BLOCK];
    if (tid == 0) push_count = 0;
    sacc[0][tid] = 0; sacc[1][tid] = 0; sacc[2][tid] = 0; sacc[3][tid] = 0;
    __syncthreads();
    u32 fq_head = 0, fq_n = 0;               // this wave's ring: first waiting entry, entries waiting (wave uniform)
    u32 (*fq)[TK_FINQ] = finq[tid >> 6];
    // score `cnt_` waiting entries (at most 64) of the wave's ring on dense lanes
    auto drain_finished = [&](u32 cnt_) __attribute__((always_inline)) {
        u32 lane = tid & 63;
        if (lane < cnt_) {
            u32 e = (fq_head + lane) & (TK_FINQ - 1);
            Game f;
            f.A = TK_U64(fq[0][e], fq[1][e]); f.B = TK_U64(fq[2][e], fq[3][e]); f.C = TK_U64(fq[4][e], fq[5][e]);
            u32 m = fq[7][e], ri = fq[8][e];
            f.talon = TK_U64(fq[6][e], (m >> 24) & 15u);
            f.contract = m & 15; f.declarer = (m >> 4) & 3; f.king = (m >> 6) & 3; f.team = (m >> 8) & 15;
            f.tl = (m >> 12) & 7; f.trick_no = (m >> 16) & 15; f.leader = (m >> 20) & 3;
            f.trick = 0; f.nt = 0; f.phase = TK_PHASE_DONE; f.error = 0; f.epar = 0; f.cprev = 0;
            u64 sc = final_scores(f);
            u32 t = ri & 0xFFFFu;
            if (reward) {
                u64 rs = sc;
                if ((flags & TAROK_REWARD_REF) && (f.contract == TK_BERAC || f.contract == TK_ODPRTI_BERAC)) {
                    int dv = f.trick_no >= 12 ? -20 : 20;                  // Igralec.py:434-437 (see the immediate path below)
                    u32 d = f.declarer;
                    rs = pack_scores(d == 0 ? (int)(int16_t)(sc & 0xFFFF) : dv, d == 1 ? (int)(int16_t)((sc >> 16) & 0xFFFF) : dv,
                                     d == 2 ? (int)(int16_t)((sc >> 32) & 0xFFFF) : dv, d == 3 ? (int)(int16_t)(sc >> 48) : dv);
                }
                reinterpret_cast<u64 *>(reward)[(int64_t)group * TK_BLOCK + t + (int64_t)(ri >> 16) * stride] = rs;
            }
            atomicAdd(&sacc[0][t], (int)(int16_t)(sc & 0xFFFF)); atomicAdd(&sacc[1][t], (int)(int16_t)((sc >> 16) & 0xFFFF));
            atomicAdd(&sacc[2][t], (int)(int16_t)((sc >> 32) & 0xFFFF)); atomicAdd(&sacc[3][t], (int)(int16_t)(sc >> 48));
        }
        fq_head = (fq_head + cnt_) & (TK_FINQ - 1);
        fq_n -= cnt_;
    };
    u64 t_real0 = 0, t_cyc0 = 0, t_play = 0;
    if (stamps) { t_real0 = __builtin_amdgcn_s_memrealtime(); t_cyc0 = __builtin_amdgcn_s_memtime(); }
    int64_t i = (int64_t)group * TK_BLOCK + tid;
    bool valid = i < n;
    int64_t ic = valid ? i : n - 1;
    Game g;
    load_game(g, s01, s23, ic);
    u64 key = gkey[ic];
    bool autoreset = (flags & TAROK_AUTO_RESET) != 0;
    // Lanes that can reach the end of their game within this launch (a game only ends on the 4th
    // card of a trick: Berac on any trick, the others in trick 12) issue their finish-path loads
    // NOW, next to the state load, instead of as a second memory round trip after the rules.
    bool berac = g.contract == TK_BERAC || g.contract == TK_ODPRTI_BERAC;
    bool spec = valid && ((g.phase == TK_PHASE_PLAY && (int)g.nt + cards >= 4 &&
                           (berac || (int)(g.trick_no * 4 + g.nt) + cards >= 48)) ||
                          g.phase == TK_PHASE_DONE);
    // Next-game lines this launch may take: the lines the previous launch put on its refill list
    // (the `cprev` farthest ahead) are being written by refill workgroups right now.
    u32 cprev0 = valid ? g.cprev : 0u;
    u32 allowed = TK_AHEAD - min(cprev0, (u32)TK_AHEAD);
    int4 acc = make_int4(0, 0, 0, 0);
    u32 cur_ep = 0;
    // (na, nb, nkey, nep1): the line of the next game (episode cur_ep + 1), (na2, nb2, nkey2, nep2): of the
    // one after it.  ok1 / ok2: the line has been requested (and was one this launch may take); its episode tag
    // is compared when it is taken, so that nothing waits for the load where it is issued (line_tag).  Both
    // are loaded here, before the loop, and topped up at the first card of a trick (below) — never across the
    // loop's back edge: a load in flight there would make every iteration wait for the previous iteration's
    // stores (vmcnt counts both, in order)
    ulonglong2 na = make_ulonglong2(0, 0), nb = na, na2 = na, nb2 = na;
    u64 nkey = 0, nkey2 = 0;
    u32 nep1 = 0, nep2 = 0;
    bool ok1 = false, ok2 = false;
    auto line_tag = [&](bool &ok, u32 &nep, u32 tag) __attribute__((always_inline)) { nep = tag; ok = true; };
    if (spec) {
        acc = cnt[i].score_sum;
        cur_ep = cnt[i].episode;
        if (autoreset && allowed > 0) {
            const AuxLine *ln = &aux[i].line[TK_LINE(g.epar + 1)];
            na = ln->n01; nb = ln->n23; nkey = ln->nkey;
            line_tag(ok1, nep1, ln->nep);
            if (allowed > 1 && cards > 4) {          // a Berac can be over after 4 cards
                const AuxLine *l2 = &aux[i].line[TK_LINE(g.epar + 2)];
                na2 = l2->n01; nb2 = l2->n23; nkey2 = l2->nkey;
                line_tag(ok2, nep2, l2->nep);
            }
        }
    }
    TK_WAIT_LOADS();                        // nothing in flight when the loop starts (see above)
    u32 par = launch_phase<1>(count, play_groups, fan) & 1u;     // (`count` was requested before the state: it has arrived with it)
    launch_counted<1>(epoch);
    g.cprev = 0;
    u32 consumed = 0;                       // games swapped in / dealt during this launch
#ifdef TK_EVENT_STAMPS                      // diagnostics build (tools/ev_probe.py): per-wave event counts
    u32 ev_deal = 0, ev_lazy = 0, ev_renew = 0, ev_early = 0;
#endif
#ifdef TK_CARD_STAMPS
    u32 cs_012 = 0, cs_3 = 0, cs_seg[4] = {0, 0, 0, 0};      // cs_seg: the 4th card's rules | scoring queue | renewal | outputs
    u64 cs_t = 0;
#define TK_SEG(k) do { if constexpr (ALL && NT == 3) { u64 t_ = __builtin_amdgcn_s_memtime(); cs_seg[k] += (u32)(t_ - cs_t); cs_t = t_; } } while (0)
#else
#define TK_SEG(k) do { } while (0)
#endif
    bool resync = false;                    // a line that should have been usable was not: refill them all
    bool blocked = false;                   // a game was dealt in place: no more swap-ins in this launch
    bool acc_dirty = false, seats_dirty = false, touched = false;
    // the legal mask written into the observation after card c is the one the policy needs for
    // card c+1: computed once per card, carried in a register
    u64 legal = (valid && g.phase == TK_PHASE_PLAY) ? legal_now(g) : 0;
    u64 c_lead = 0;
    u64 pending = 0;                         // (wave uniform) lanes that came out of a swap without a line for their next game
    u32 blocked_v = 0, resync_v = 0;         // `blocked` / `resync` of the fast-renewal loop, as numbers (a bool carried through a loop is a
                                             // lane mask, merged with three scalar instructions at every join, used or not)
    // fill the line buffers of the lanes that lack them (lacks: the next game's, lacks2: the one after it)
    auto fetch_lines = [&](bool lacks, bool lacks2) __attribute__((always_inline)) {
        if (lacks) {
            const AuxLine *ln = &aux[i].line[TK_LINE(cur_ep + 1)];
            na = ln->n01; nb = ln->n23; nkey = ln->nkey;
            line_tag(ok1, nep1, ln->nep);
        }
        if (lacks2) {
            const AuxLine *l2 = &aux[i].line[TK_LINE(cur_ep + 2)];
            na2 = l2->n01; nb2 = l2->n23; nkey2 = l2->nkey;
            line_tag(ok2, nep2, l2->nep);
        }
    };
    auto play_card = [&](auto all_tag, auto nt_tag, auto std_tag, int64_t row, int ci) __attribute__((always_inline)) {
        // ALL: every lane of the wave is a valid slot with a game in play (wave uniform, see below):
        // no per-lane predicates around the rules and the output stores.
        // NT >= 0: moreover every lane is at card NT of its trick: the constant propagates through
        // the rules (no trick-end test on cards 0..2, constant shifts, "somebody led" known)
        // STD: the usual set of outputs of a rollout — action_out and done given, trick not — known for the
        // whole launch: no pointer tests per card (two scalar instructions each: as dear as vector ones here)
        constexpr bool ALL = decltype(all_tag)::value;
        constexpr int NT = decltype(nt_tag)::value;
        constexpr bool STD = decltype(std_tag)::value;
        if constexpr (NT >= 0) g.nt = (u32)NT;
        // Trick-aligned loops: the lines of the next games are topped
        // up HERE, at the first card of a trick, as soon as one lane has used its two up — three cards (~1,400
        // cycles) before a game can end and take one: the memory round trip hides behind the rules, and the
        // compiler's wait at the first use counts past the stores issued since.  (A lane's `consumed` only moves
        // at a 4th card, so a lane that still lacks a line at the 4th card could not have fetched one: no fetch
        // on the spot in these loops.)  With four waves on a SIMD the other waves hide the round trip of a
        // fetch on the spot, and the earlier, more frequent top-ups only cost instructions (4 M games: -11 %).
        constexpr bool TOP_UP_EARLY = ALL && NT == 0;
        constexpr bool TOP_UP_LATE = !(ALL && NT == 3);
        if constexpr (TOP_UP_EARLY) {
            // (`pending`: set at a 4th card when a finishing lane came out of its swap without the next line —
            // a scalar test here, not a vote on per-lane conditions at every trick.  Which lane holds which line
            // is read off the tags: a lane holds the line of its next game if nep1 is that game's number.)
            if (TK_RARE(pending != 0)) {
#ifdef TK_EVENT_STAMPS
                ev_early++;
#endif
                pending = 0;
                bool base = spec && blocked_v == 0 && consumed >= 1;
                bool lacks = base && nep1 != cur_ep + 1 && consumed < allowed;
                bool lacks2 = base && nep2 != cur_ep + 2 && consumed + 1 < allowed;
                fetch_lines(lacks, lacks2);
            }
        }
        const bool v = ALL ? true : valid;
        const bool play = ALL ? true : (valid && g.phase == TK_PHASE_PLAY);
#ifdef TK_CARD_STAMPS
        if constexpr (ALL && NT == 3) cs_t = __builtin_amdgcn_s_memtime();
#endif
        u32 a = play ? policy_action(key, g.trick_no * 4 + g.nt, legal) : 255u;
        u64 scores = 0;
        u32 trick_info = 0;
        int res = -2;
        const u32 pos = g.trick_no * 4 + g.nt;            // cards played so far in this game
        // (trick-aligned loops: the C plane as the trick's first card finds it — what it gains until the 4th card is the trick)
        if constexpr (ALL && NT == 0) c_lead = g.C;
        const u64 *lead_plane = (ALL && NT >= 0) ? &c_lead : nullptr;
        if (play) res = apply_step<true, true>(g, a, scores, trick_info, !STD && trick != nullptr, lead_plane);
        bool fin = res == 1;
        // the play history (zgodovina, Klop.py:63 / Navadna_igra.py:127): card `pos` of the game, one byte,
        // write-only here; only the reference-layout observation (k_observe_ref) reads it
        if (HIST && hist && play && res >= 0) hist[(int64_t)pos * n + i] = (uint8_t)a;
        // (trick-aligned loops: every lane plays whole tricks — both are set once, before the loop; a per-lane
        // flag carried through a loop is a lane mask that costs three scalar instructions per update)
        if constexpr (!(ALL && NT >= 0)) {
            touched = touched || res != -2;
            seats_dirty = seats_dirty || (res >= 0 && g.nt == 0);
        }
        if (v) {
            if (STD || action_out) TK_STREAM_STORE(&action_out[row], (uint8_t)a);
            if (!STD && trick) TK_STREAM_STORE(&trick[row], (uint16_t)trick_info);
        }
        TK_SEG(0);
        // (cards 0..2 of a trick cannot end a game: no finish / renewal code in their copies)
        constexpr bool CAN_END = !(ALL && NT >= 0 && NT < 3);
        // the trick-aligned loop: queue + renewal of the finishing lanes in one exec region, every per-lane fact
        // kept in vector registers (below)
        constexpr bool FAST_RENEW = ALL && NT == 3;
        auto push_finished = [&](u64 fm, u32 slot0) __attribute__((always_inline)) {   // (lanes with fin; slot0: first free ring entry)
            u32 e = (slot0 + __builtin_amdgcn_mbcnt_hi((u32)(fm >> 32), __builtin_amdgcn_mbcnt_lo((u32)fm, 0))) & (TK_FINQ - 1);
            fq[0][e] = TK_LO(g.A); fq[1][e] = TK_HI(g.A); fq[2][e] = TK_LO(g.B); fq[3][e] = TK_HI(g.B);
            fq[4][e] = TK_LO(g.C); fq[5][e] = TK_HI(g.C); fq[6][e] = TK_LO(g.talon);
            // (one v_lshl_or per field: left alone the compiler builds a tree of shifts and v_or3, twelve instructions)
            u32 m = (g.declarer << 4) | g.contract;      TK_KEEP_VGPR(m);
            m = (g.king << 6) | m;                       TK_KEEP_VGPR(m);
            m = (g.team << 8) | m;                       TK_KEEP_VGPR(m);
            m = (g.tl << 12) | m;                        TK_KEEP_VGPR(m);
            m = (g.trick_no << 16) | m;                  TK_KEEP_VGPR(m);
            m = (g.leader << 20) | m;                    TK_KEEP_VGPR(m);
            fq[7][e] = (TK_HI(g.talon) << 24) | m;
            fq[8][e] = ((u32)ci << 16) | tid;
        };
        auto swap_in = [&]() __attribute__((always_inline)) {                   // (lanes that hold their next game's line)
            unpack_fresh(g, na.x, na.y, nb.x, nb.y);  // carries epar of the new game
            key = nkey;
            na = na2; nb = nb2; nkey = nkey2;
            nep1 = nep2;
        };
        // a game dealt here and now, for the lanes with deal_here (all 64 lanes must come along: deal_wave)
        auto deal_in_place = [&](bool deal_here, Game &gd, u64 &kd) __attribute__((always_inline)) {
            u64 pend = __ballot(deal_here);
            if (TK_RARE(pend != 0)) {
#ifdef TK_EVENT_STAMPS
                ev_deal += (u32)__popcll(pend);
#endif
                u64 dkey = 0;
                if (deal_here) dkey = game_key(seed, offset + (u64)i, cur_ep + 1);
                u64 h0 = 0, h1 = 0, h2 = 0, h3 = 0, tal = 0;
                u32 lane = __lane_id();
                while (pend) {
                    int l = __builtin_ctzll(pend);
                    pend &= pend - 1;
                    u32 klo = (u32)__builtin_amdgcn_readlane((int)(u32)dkey, l);
                    u32 khi = (u32)__builtin_amdgcn_readlane((int)(u32)(dkey >> 32), l);
                    u64 w0, w1, w2, w3, wt;
                    deal_wave(klo, khi, w0, w1, w2, w3, wt);
                    if (lane == (u32)l) { h0 = w0; h1 = w1; h2 = w2; h3 = w3; tal = wt; }
                }
                if (deal_here) {
                    u32 cc, d, k;
                    sample_setup(dkey, mix, cc, d, k);
                    setup_game(gd, h0, h1, h2, h3, tal, cc, d, k);
                    gd.epar = TK_LINE(cur_ep + 1); gd.cprev = 0;
                    if (gd.phase == TK_PHASE_EXCHANGE) bot_exchange(gd, dkey);
                    kd = dkey;
                }
            }
        };
        if constexpr (FAST_RENEW) {
            // Every lane is in play, so the lanes to renew are the finishing ones, and a lane can take its next game
            // from the line it holds exactly when that line's tag is the game's number (a tag never matches a later
            // game, and a line with the right tag IS the game: no "requested" / "dealt in place" flags here).  A
            // finishing lane without its line is rare: its game is dealt on the spot INTO the line registers, so that
            // there is one renewal path: queue entry, swap and counters of the finishing lanes in one exec region, all
            // on vector registers.  (Per-lane flags kept as lane masks cost three scalar instructions per update
            // inside a divergent region, and every compare -> mask -> select hop stalls a lone wave: tools/valu_issue.)
            u64 fm = __ballot(fin);
            if (TK_USUAL(fm != 0)) {                                      // (wave uniform)
#ifdef TK_EVENT_STAMPS
                ev_renew++;
#endif
                if (TK_RARE(fq_n >= 64)) drain_finished(64);              // room for 64 more: fewer than 64 wait now
                const u32 slot0 = fq_head + fq_n;                         // (scalar bookkeeping outside the exec region)
                fq_n += (u32)__popcll(fm);
                // (a vote on ONE compare, and-ed with the finishing lanes as a scalar: see apply_step's last lines)
                if (TK_RARE((fm & __ballot(nep1 != cur_ep + 1)) != 0)) {
                    bool lineless = fin && nep1 != cur_ep + 1;
                    // ran out of usable lines (the next ones are being re-dealt right now: the usual
                    // bookkeeping stays valid) vs a line that should have been there and is not
                    resync_v |= (lineless && consumed < allowed) ? 1u : 0u;
                    blocked_v |= lineless ? 1u : 0u;                      // (no more fetches for this lane in this launch)
                    Game d = g;
                    u64 dk = key;
                    deal_in_place(lineless, d, dk);
                    if (lineless) { pack(d, na.x, na.y, nb.x, nb.y); nkey = dk; nep1 = cur_ep + 1; }
                }
                // (who comes out of the swap without the next line: what the swap moves up is the second buffer)
                pending |= fm & __ballot(nep2 != cur_ep + 2);
                if (fin) { push_finished(fm, slot0); swap_in(); cur_ep++; consumed++; }
            }
        }
        if constexpr (CAN_END && !FAST_RENEW) {
            u64 fm = __ballot(fin);
            if (TK_USUAL(fm != 0)) {                                      // (wave uniform)
                if (TK_RARE(fq_n >= 64)) drain_finished(64);                       // room for 64 more: fewer than 64 wait now
                if (fin) push_finished(fm, fq_head + fq_n);
                fq_n += (u32)__popcll(fm);
            }
        }
        TK_SEG(1);
        if (CAN_END && !FAST_RENEW && (ALL || autoreset)) {  // (ALL implies auto-reset, and every lane was in play: done = just finished)
            bool renew = ALL ? fin : (v && g.phase == TK_PHASE_DONE);
            if (TK_USUAL(__ballot(renew) != 0)) {
#ifdef TK_EVENT_STAMPS
                ev_renew++;
#endif
                // Loops that are not trick-aligned: a third or later game of a launch (the two preloaded lines
                // are used up) is fetched on the spot: a memory round trip (~550 cycles) that nothing hides.  So
                // when one lane has to, EVERY lane that has used its lines up takes its next games' lines along —
                // finishing or not: the one wait serves them all.
                if constexpr (TOP_UP_LATE) {
                    bool lacks = spec && !blocked && !ok1 && consumed >= 1 && consumed < allowed;
                    if (TK_RARE(__ballot(renew && lacks) != 0)) {
#ifdef TK_EVENT_STAMPS
                        ev_lazy++;
#endif
                        bool lacks2 = spec && !blocked && !ok2 && consumed >= 1 && consumed + 1 < allowed;
                        fetch_lines(lacks, lacks2);
                        TK_WAIT_LOADS();
                    }
                }
                bool swap = renew && !blocked && ok1 && nep1 == cur_ep + 1;
                if (swap) {
                    swap_in();
                    ok1 = ok2 && consumed + 1 < allowed;
                    ok2 = false;
                }
                bool deal_here = renew && !swap;             // line missing, stale or being re-dealt
                // ran out of usable lines (the next ones are being re-dealt right now: the usual
                // bookkeeping stays valid) vs a line that should have been there and is not
                if (deal_here && consumed < allowed) resync = true;
                if (deal_here) blocked = true;
                deal_in_place(deal_here, g, key);
                if (renew) { cur_ep++; consumed++; seats_dirty = true; }
            }
        }
        TK_SEG(2);
        // (ALL: a finished game has been replaced just above, so every lane is in play again)
        legal = (ALL || (v && g.phase == TK_PHASE_PLAY)) ? legal_now(g) : 0;
        if (v) {
            // (ALL: res is 0 or 1 — the number itself goes into the observation's bit 62 and the done row, no selects)
            const u32 fin01 = ALL ? (u32)res : (fin ? 1u : 0u);
            TK_STREAM_STORE(&obs[row], obs_word_with<true>(g, false, legal) | ((u64)(ALL ? fin01 : ((fin || g.phase == TK_PHASE_DONE) ? 1u : 0u)) << 62));
            if (STD || done) TK_STREAM_STORE(&done[row], (uint8_t)fin01);
        }
        TK_SEG(3);
    };
    // With auto-reset a lane that is in play stays in play (a finished game is replaced within the
    // same card), so "every lane of the wave valid and in play" decided HERE holds for the whole
    // launch: one of two separate loops (a choice per card made the loop body slower than either).
    // Games are whole tricks long, so lanes that start a launch of whole tricks at a trick boundary
    // stay trick-aligned too: third loop, four specialised cards per trick.
    int64_t row = i;
    typedef std::integral_constant<int, -1> nt_any;
    if (autoreset && __ballot(valid && g.phase == TK_PHASE_PLAY) == ~0ULL) {
        if ((cards & 3) == 0 && __ballot(g.nt != 0) == 0) {
            auto tricks = [&](auto std_tag) __attribute__((always_inline)) {
                touched = true; seats_dirty = true;          // (cards >= 4: every lane plays a whole trick)
                for (int c = 0; c < cards; c += 4) {
#ifdef TK_CARD_STAMPS                       // diagnostics build (tools/card_probe.py): cycles of cards 0-2 vs the trick's 4th card
                    u64 ts_a = __builtin_amdgcn_s_memtime();
#endif
                    play_card(std::true_type{}, std::integral_constant<int, 0>{}, std_tag, row, c); row += stride;
                    play_card(std::true_type{}, std::integral_constant<int, 1>{}, std_tag, row, c + 1); row += stride;
                    play_card(std::true_type{}, std::integral_constant<int, 2>{}, std_tag, row, c + 2); row += stride;
#ifdef TK_CARD_STAMPS
                    u64 ts_b = __builtin_amdgcn_s_memtime();
#endif
                    play_card(std::true_type{}, std::integral_constant<int, 3>{}, std_tag, row, c + 3); row += stride;
#ifdef TK_CARD_STAMPS
                    u64 ts_c = __builtin_amdgcn_s_memtime();
                    cs_012 += (u32)(ts_b - ts_a); cs_3 += (u32)(ts_c - ts_b);
#endif
                }
            };
            if (action_out && done && !trick) tricks(std::true_type{});
            else tricks(std::false_type{});
        } else {
            for (int c = 0; c < cards; c++, row += stride) play_card(std::true_type{}, nt_any{}, std::false_type{}, row, c);
        }
    } else {
        for (int c = 0; c < cards; c++, row += stride) play_card(std::false_type{}, nt_any{}, std::false_type{}, row, c);
    }
    // ---- schedule the refills: after consuming, the slot must again hold episodes cur+1 .. cur+TK_AHEAD.
    // Each swap-in vacated one line (the others stay valid): the last np episodes are new;
    // a game dealt in place: all of them.
    while (fq_n) drain_finished(min(fq_n, 64u));                          // (at most two passes: fewer than 128 wait)
    __builtin_amdgcn_s_waitcnt(0xC07F);                                   // the wave's LDS adds have landed (lgkmcnt(0))
    {
        int4 sa = make_int4(sacc[0][tid], sacc[1][tid], sacc[2][tid], sacc[3][tid]);
        if (sa.x | sa.y | sa.z | sa.w) { acc.x += sa.x; acc.y += sa.y; acc.z += sa.z; acc.w += sa.w; acc_dirty = true; }
    }
    u32 np = (resync || resync_v != 0) ? (u32)TK_AHEAD : min(consumed, (u32)TK_AHEAD);
    if (valid) {
        g.cprev = np;                        // the next launch must not read those lines
        if (acc_dirty) cnt[i].score_sum = acc;
        if (consumed) { cnt[i].episode = cur_ep; gkey[i] = key; }
        if (touched || consumed || cprev0 != np) store_game(g, s01, s23, i, seats_dirty || cprev0 != np);
    }
    if (stamps) t_play = __builtin_amdgcn_s_memtime() - t_cyc0;
    if (valid && np) {
        u32 pos = atomicAdd(&push_count, np);
        push_ep[tid] = cur_ep;
        for (u32 j = 0; j < np; j++) push_list[pos + j] = (unsigned short)(j * TK_BLOCK + tid);
    }
    __syncthreads();
    u32 total = push_count;
    u64 *lst = rlist + ((int64_t)group * 2 + par) * TK_REFILL_CAP;
    for (u32 j = tid; j < total; j += TK_BLOCK) {                  // list entry: episode to deal << 32 | slot in group
        u32 en = push_list[j], t = en % TK_BLOCK;
        lst[j] = ((u64)(push_ep[t] + TK_AHEAD - en / TK_BLOCK) << 32) | t;
    }
    if (tid == 0) rcount[TK_RC(group, par)] = total;
    if (stamps && (tid & 63) == 0) {     // diagnostics only
        u64 w = (u64)i >> 6;
        stamps[3 * w + 0] = t_real0;
        stamps[3 * w + 1] = __builtin_amdgcn_s_memrealtime();
        stamps[3 * w + 2] = ((__builtin_amdgcn_s_memtime() - t_cyc0) << 32) | (t_play & 0xFFFFFFFFULL);
#ifdef TK_CARD_STAMPS
        stamps[3 * w + 0] = ((u64)cs_012 << 32) | (u64)cs_3;
        stamps[3 * w + 1] = ((u64)cs_seg[0] << 32) | (u64)cs_seg[1];
        stamps[3 * w + 2] = (stamps[3 * w + 2] & 0xFFFFFFFFULL) | ((u64)cs_seg[2] << 32);      // (cs_seg[3] = cs_3 - the others)
#endif
#ifdef TK_EVENT_STAMPS                      // (replaces the entry time stamp)
        stamps[3 * w + 0] = ((u64)ev_deal << 48) | ((u64)ev_lazy << 32) | ((u64)ev_renew << 16) | (u64)ev_early;
#endif
    }
}

// The Bot-policy kernel (tarok_krog_random; every batch size): ~165 VGPRs, nothing spills inside the card loops —
// still three waves per SIMD, and where the batch puts one wave on a SIMD (65,536 games) a scratch reload would
// be a memory round trip that nothing hides; the next games' lines are topped up early and the usual path falls
// through its branches.  (Rounds 1-2 also shipped a 128-VGPR build of it, four waves per SIMD, slower at every
// batch size — same-box medians, G steps/s, wide | narrow: 262,144 games 160.9 | 155.3, 1 M 163.9 | 146.0,
// 4 M 175.9 | 169.3 — and selectable only through an environment variable: removed in round 3.)
// HIST: also record the play history (one byte per card; tarok_create flag TAROK_HISTORY).
#define TK_PLAY_ARGS                                                                                                             \
    int64_t n, u64 seed, u64 offset, int mix, int flags, int cards, int64_t stride, u32 play_groups, u32 *epoch, u32 fan,         \
        uint8_t *__restrict__ action_out, int16_t *__restrict__ reward,                                                           \
        uint8_t *__restrict__ done, uint16_t *__restrict__ trick, u64 *__restrict__ obs, uint8_t *__restrict__ hist,              \
        ulonglong2 *__restrict__ s01, ulonglong2 *__restrict__ s23, Aux *aux, Counters *__restrict__ cnt,                         \
        u64 *__restrict__ gkey, u64 *rlist, u32 *rcount, u64 *__restrict__ stamps
template <bool HIST>
TK_KERNEL(TK_BLOCK, 168) __attribute__((amdgpu_waves_per_eu(3, 3))) void k_play_wide(TK_PLAY_ARGS) {
    TK_VGPR_TOP(168, 167);
    TkCount count = launch_count<1>(epoch, play_groups);
    if (blockIdx.x >= play_groups)
        refill_role<false>(blockIdx.x - play_groups, threadIdx.x, TK_BLOCK, seed, offset, mix, play_groups, count, epoch, fan, false, aux, rlist, rcount, nullptr);
    else {
#ifdef TK_PLAY_PRIO                          // diagnostics build: wave priority of the play role (no effect: profiles/r02_ab_lone_wave_rewrite.txt)
        __builtin_amdgcn_s_setprio(TK_PLAY_PRIO);
#endif
        play_role<HIST>(blockIdx.x, threadIdx.x, n, seed, offset, mix, flags, cards, stride, count, epoch, play_groups, fan,
                        action_out, reward, done, trick, obs, hist, s01, s23, aux, cnt, gkey, rlist, rcount, stamps);
    }
}

// A game dealt here and now by the wave (deal_wave: all 64 lanes must come along) for the lanes with deal_here:
// episode `ep` of slot i.  The rare path of a renewal: the slot's next-game line is missing or being re-dealt.
__device__ __forceinline__ void deal_in_place_wave(bool deal_here, Game &gd, u64 &kd, u64 seed, u64 offset, int64_t i, u32 ep, int mix) {
    u64 pend = __ballot(deal_here);
    if (pend == 0) return;
    u64 dkey = 0;
    if (deal_here) dkey = game_key(seed, offset + (u64)i, ep);
    u64 h0 = 0, h1 = 0, h2 = 0, h3 = 0, tal = 0;
    u32 lane = __lane_id();
    while (pend) {
        int l = __builtin_ctzll(pend);
        pend &= pend - 1;
        u32 klo = (u32)__builtin_amdgcn_readlane((int)(u32)dkey, l);
        u32 khi = (u32)__builtin_amdgcn_readlane((int)(u32)(dkey >> 32), l);
        u64 w0, w1, w2, w3, wt;
        deal_wave(klo, khi, w0, w1, w2, w3, wt);
        if (lane == (u32)l) { h0 = w0; h1 = w1; h2 = w2; h3 = w3; tal = wt; }
    }
    if (deal_here) {
        u32 cc, d, k;
        sample_setup(dkey, mix, cc, d, k);
        setup_game(gd, h0, h1, h2, h3, tal, cc, d, k);
        gd.epar = TK_LINE(ep); gd.cprev = 0;
        if (gd.phase == TK_PHASE_EXCHANGE) bot_exchange(gd, dkey);
        kd = dkey;
    }
}

// THE ONE-CARD STEP (tarok_step, tarok_step_random, the env half of tarok_policy_step): the surface an external
// policy drives, one card of every game per launch, the state through HBM on every card — the path SURVEY 8d's
// 54 B/step describe.  Same results and the same refill protocol as play_role with cards = 1 (launches of both
// kinds mix freely), but compiled on its own: no card loops, no scoring queue, no scratch — the generic kernel's
// 128-VGPR build spilled twelve dwords per lane on this path, and a spilled dword is a store to memory: 47 of
// the 82 bytes per step it wrote (profiles/r03_step_ledger.txt).
//   Reads:  play pair 16 + seat pair 16 + the card 1 (RANDOM: the game's RNG key 8 instead).
//   Writes: play pair 16 + observation word 8 (+ done 1, action 1, trick 2 where asked for);
//           the seat pair (16) only by the lanes whose trick this card completed: every 4th card.
//   A game that ENDS also touches its slot's Counters (32 B read, 32 B written), takes its successor from the
//   slot's next-game line (64 B read), writes the successor's RNG key and puts one entry on the refill list.
// Finished games are scored on DENSE lanes: about a tenth of the slots end a game on a trick's 4th card, so every
// wave would run the scoring code (both contract families, ~270 instructions) for six active lanes — measured at
// 4 M games, the finishing lanes of a mixed batch cost 116 ps each against 40 ps in an all-Klop batch, where all
// lanes end together (profiles/r03_step_durations.txt).  A lane whose game ends leaves the final state and the
// slot's score sums in a workgroup-wide LDS list (FINQ_WORDS words per entry) and goes on to the renewal; after
// the role's one barrier the first threads of the workgroup score the list, one entry per lane, and write the
// reward rows and the score sums.  Nothing a lane does after the card depends on the scores.
// Only the lanes whose game DID end load those, after the rules (the scoring list is filled meanwhile).  Rounds 1-2
// had every lane that CAN end its game with this card — 4th card of a Berac trick or of a twelfth trick, a fifth of
// the lanes of such a launch — issue them speculatively next to the state load: 5.3 B/step more read traffic at
// 4 M games, and no faster at any batch size, 65,536 games included (profiles/r03_ab_step.txt).
#define FINQ_WORDS 13
template <bool RANDOM, bool LAZY>
__device__ __forceinline__ void step_role(
    u32 group, u32 tid, bool active, u32 a_reg, bool lazy_on,
    int64_t n, u64 seed, u64 offset, int mix, int flags, TkCount count, u32 *epoch, u32 play_groups, u32 fan,
    const uint8_t *__restrict__ action_in, uint8_t *__restrict__ action_out, int16_t *__restrict__ reward,
    uint8_t *__restrict__ done, uint16_t *__restrict__ trick, u64 *__restrict__ obs, uint8_t *__restrict__ hist,
    ulonglong2 *__restrict__ s01, ulonglong2 *__restrict__ s23, Aux *aux, Counters *__restrict__ cnt,
    u64 *__restrict__ gkey, u64 *rlist, u32 *rcount, u64 *elist,
    u32 (*__restrict__ finq)[TK_BLOCK] /* LDS [FINQ_WORDS][TK_BLOCK] */) {
    __shared__ unsigned short push_list[TK_REFILL_CAP];   // (how far ahead) * TK_BLOCK + slot in group
    __shared__ u32 push_ep[TK_BLOCK];                     // the slot's episode number at the end of the launch
    __shared__ u32 push_count, fin_count, late_count;
    __shared__ u64 late_list[LAZY ? TK_BLOCK : 1];        // LAZY: entries for the stretch list (refill_role), one per slot at most
    const bool lazy = LAZY && lazy_on;                    // (LAZY: the kernel has the stretch lists at all; lazy_on: this env uses them)
    if (tid == 0) { push_count = 0; fin_count = 0; late_count = 0; }
    // (lazy) how full the two stretch lists of this group are: asked for now by every thread (one address per workgroup),
    // looked at when the launch's entries go out — no broadcast, no barrier of its own
    u32 efill0 = 0, efill1 = 0;
    if (lazy) { efill0 = tk_ld(&rcount[TK_RC(group, 2)]); efill1 = tk_ld(&rcount[TK_RC(group, 3)]); }
    __syncthreads();
    int64_t i = (int64_t)group * TK_BLOCK + tid;
    bool valid = active && i < n;
    int64_t ic = valid ? i : n - 1;
    Game g;
    load_game(g, s01, s23, ic);
    u64 key = 0;
    u32 a = 255;
    if (RANDOM) key = gkey[ic]; else a = action_in ? action_in[ic] : a_reg;
    const bool autoreset = (flags & TAROK_AUTO_RESET) != 0;
    // next-game lines this launch may take: the `cprev` farthest ahead are being re-dealt right now (play_role)
    const u32 cprev0 = valid ? g.cprev : 0u;
    const u32 allowed = TK_AHEAD - min(cprev0, (u32)TK_AHEAD);
    int4 acc = make_int4(0, 0, 0, 0);
    u32 cur_ep = 0;
    ulonglong2 na = make_ulonglong2(0, 0), nb = na;
    u64 nkey = 0;
    u32 ntag = 0;
    bool have_line = false;
    auto load_finish = [&](bool need) __attribute__((always_inline)) {
        if (need) {
            acc = cnt[i].score_sum;
            cur_ep = cnt[i].episode;
            if (autoreset && allowed > 0) {
                const AuxLine *ln = &aux[i].line[TK_LINE(g.epar + 1)];
                na = ln->n01; nb = ln->n23; nkey = ln->nkey; ntag = ln->nep;
                have_line = true;
            }
        }
    };
    launch_counted<LAZY ? 0 : 1>(epoch);     // (issued BEFORE the count is waited for: its round trip runs beside the state's)
    g.cprev = 0;
    // ---- the card: krog's body (Klop.py:47-79, Navadna_igra.py:115-141), apply_step
    const bool play = valid && g.phase == TK_PHASE_PLAY;
    const u32 pos = g.trick_no * 4 + g.nt;                // cards played so far in this game
    if (RANDOM) a = play ? policy_action(key, pos, legal_now(g)) : 255u;
    u64 scores = 0;
    u32 trick_info = 0;
    int res = -2;
    // (a finished game is scored after the barrier, from its final state, on dense lanes)
    if (play) res = apply_step<RANDOM, true>(g, a, scores, trick_info, trick != nullptr, nullptr);
    const bool fin = res == 1;
    // the play history (zgodovina, Klop.py:63 / Navadna_igra.py:127): write-only here
    if (hist && play && res >= 0) hist[(int64_t)pos * n + i] = (uint8_t)a;
    bool seats_dirty = res >= 0 && g.nt == 0;
    if (valid) {
        if (RANDOM && action_out) TK_STREAM_STORE(&action_out[i], (uint8_t)a);
        if (trick) TK_STREAM_STORE(&trick[i], (uint16_t)trick_info);
    }
    const bool renew = autoreset && valid && g.phase == TK_PHASE_DONE;     // (also a game finished by an earlier launch)
    load_finish(fin || renew);
    if (fin) {                                            // the final state goes on the scoring list (the loads above are in flight)
        u32 e = atomicAdd(&fin_count, 1u);
        finq[0][e] = TK_LO(g.A); finq[1][e] = TK_HI(g.A); finq[2][e] = TK_LO(g.B); finq[3][e] = TK_HI(g.B);
        finq[4][e] = TK_LO(g.C); finq[5][e] = TK_HI(g.C); finq[6][e] = TK_LO(g.talon);
        finq[7][e] = g.contract | (g.declarer << 4) | (g.king << 6) | (g.team << 8) | (g.tl << 12) | (g.trick_no << 16) | (g.leader << 20) |
                     (TK_HI(g.talon) << 24);
        finq[8][e] = tid;
        finq[9][e] = (u32)acc.x; finq[10][e] = (u32)acc.y; finq[11][e] = (u32)acc.z; finq[12][e] = (u32)acc.w;
    }
    // ---- auto-reset: the successor comes out of the slot's next-game line (valid iff its tag is the game's
    // number); without a usable line the wave deals the game here and the slot's lines are all refilled
    u32 consumed = 0;
    bool resync = false;
    if (__ballot(renew) != 0) {                           // (wave uniform: deal_in_place_wave needs every lane)
        bool swap = renew && have_line && ntag == cur_ep + 1;
        if (swap) { unpack_fresh(g, na.x, na.y, nb.x, nb.y); key = nkey; }
        bool deal_here = renew && !swap;
        resync = deal_here && allowed > 0;                // a line that should have been usable was not
        deal_in_place_wave(deal_here, g, key, seed, offset, i, cur_ep + 1, mix);
        if (renew) { cur_ep++; consumed = 1; seats_dirty = true; }
    }
    if (valid) {
        TK_STREAM_STORE(&obs[i], obs_word(g, fin));
        if (done) TK_STREAM_STORE(&done[i], (uint8_t)(fin ? 1 : 0));
    }
    // ---- state back; the refill list of this launch (see play_role).  LAZY: the one line a slot emptied by taking its next
    // game goes on the group's stretch list instead (refill_role: dealt in bulk, up to TK_BULK_EVERY launches later; nothing
    // of this slot is re-dealt during the next launch, so `cprev` stays 0); only a slot that dealt a game in place lists
    // its fourteen lines for the next launch, as play_role does
    const u32 np = resync ? (u32)TK_AHEAD : (lazy ? 0u : consumed);
    if (valid) {
        g.cprev = np;                        // the next launch must not read those lines
        if (consumed) { cnt[i].episode = cur_ep; gkey[i] = key; }
        if (res != -2 || consumed || cprev0 != np) store_game(g, s01, s23, i, seats_dirty || cprev0 != np);
    }
    if (valid && np) {
        u32 p0 = atomicAdd(&push_count, np);
        push_ep[tid] = cur_ep;
        for (u32 j = 0; j < np; j++) push_list[p0 + j] = (unsigned short)(j * TK_BLOCK + tid);
    }
    if (lazy && valid && consumed && !resync) late_list[atomicAdd(&late_count, 1u)] = ((u64)(cur_ep + TK_AHEAD) << 32) | tid;
    __syncthreads();
    // ---- the games that ended with this card, scored one per lane: final_scores from the final state alone
    // (Klop.py:36-45, Berac.py:33-44, Navadna_igra.py:80-113); reward row and score sums of the slot
    if (active)
        for (u32 e = tid, nf = fin_count; e < nf; e += TK_BLOCK) {
            Game f;
            f.A = TK_U64(finq[0][e], finq[1][e]); f.B = TK_U64(finq[2][e], finq[3][e]); f.C = TK_U64(finq[4][e], finq[5][e]);
            u32 m = finq[7][e], t = finq[8][e];
            f.talon = TK_U64(finq[6][e], (m >> 24) & 15u);
            f.contract = m & 15; f.declarer = (m >> 4) & 3; f.king = (m >> 6) & 3; f.team = (m >> 8) & 15;
            f.tl = (m >> 12) & 7; f.trick_no = (m >> 16) & 15; f.leader = (m >> 20) & 3;
            f.trick = 0; f.nt = 0; f.phase = TK_PHASE_DONE; f.error = 0; f.epar = 0; f.cprev = 0;
            u64 sc = final_scores(f);
            int64_t it = (int64_t)group * TK_BLOCK + t;
            if (reward) {
                u64 rs = sc;
                if ((flags & TAROK_REWARD_REF) && (f.contract == TK_BERAC || f.contract == TK_ODPRTI_BERAC)) {
                    // what rezultat_igre folds into a Berac defender's last transition (Igralec.py:434-437):
                    // -20 when the hands are empty at the end (all twelve tricks were played), else +20
                    int dv = f.trick_no >= 12 ? -20 : 20;
                    u32 d = f.declarer;
                    rs = pack_scores(d == 0 ? (int)(int16_t)(sc & 0xFFFF) : dv, d == 1 ? (int)(int16_t)((sc >> 16) & 0xFFFF) : dv,
                                     d == 2 ? (int)(int16_t)((sc >> 32) & 0xFFFF) : dv, d == 3 ? (int)(int16_t)(sc >> 48) : dv);
                }
                reinterpret_cast<u64 *>(reward)[it] = rs;
            }
            cnt[it].score_sum = make_int4((int)finq[9][e] + (int16_t)(sc & 0xFFFF), (int)finq[10][e] + (int16_t)((sc >> 16) & 0xFFFF),
                                          (int)finq[11][e] + (int16_t)((sc >> 32) & 0xFFFF), (int)finq[12][e] + (int16_t)(sc >> 48));
        }
    // ---- the lists: nothing to write in most launches (the refill roles clear the lengths they have worked off), and then
    // the launch's number is not even worked out
    const u32 total = push_count, late = lazy ? late_count : 0u;
    if ((total | late) == 0) return;
    const u32 phase = launch_phase<LAZY ? 0 : 1>(count, play_groups, fan), par = phase & 1u;    // (`count` was requested before the state: it arrived with it)
    if (total) {
        u64 *lst = rlist + ((int64_t)group * 2 + par) * TK_REFILL_CAP;
        if (active)
            for (u32 j = tid; j < total; j += TK_BLOCK) {          // list entry: episode to deal << 32 | slot in group
                u32 en = push_list[j], t = en % TK_BLOCK;
                lst[j] = ((u64)(push_ep[t] + TK_AHEAD - en / TK_BLOCK) << 32) | t;
            }
        if (tid == 0) rcount[TK_RC(group, par)] = total;
    }
    if (lazy && late) {
        const u32 eb = (phase / TK_BULK_EVERY) & 1u;       // this stretch's list; entries beyond its capacity are dropped (never:
        const u32 fill = eb ? efill1 : efill0;             // a slot ends at most four games in a stretch; a dropped line would
        u64 *el = elist + ((int64_t)group * 2 + eb) * TK_BULK_CAP;          // just stay stale)
        if (active && tid < late && fill + tid < TK_BULK_CAP) el[fill + tid] = late_list[tid];
        if (tid == 0) rcount[TK_RC(group, 2 + eb)] = min(fill + late, (u32)TK_BULK_CAP);
    }
}

// (the library is built with kernel-argument preload — the first fourteen dwords arrive in scalar registers with the
// dispatch instead of behind a scalar load: tarok_amd/_native.py.  Putting the state pointers first instead of the scalars
// measured no better at 65,536 games and 3 % worse at 262,144: profiles/r04_ab_step.txt)
#define TK_STEP_ARGS                                                                                                              \
    int64_t n, u64 seed, u64 offset, int mix, int flags, u32 play_groups, u32 *epoch, u32 fan,                                    \
        const uint8_t *__restrict__ action_in, uint8_t *__restrict__ action_out, int16_t *__restrict__ reward,                    \
        uint8_t *__restrict__ done, uint16_t *__restrict__ trick, u64 *__restrict__ obs, uint8_t *__restrict__ hist,              \
        ulonglong2 *__restrict__ s01, ulonglong2 *__restrict__ s23, Aux *aux, Counters *__restrict__ cnt,                         \
        u64 *__restrict__ gkey, u64 *rlist, u32 *rcount, u64 *elist
#ifndef TK_STEP_WAVES
#define TK_STEP_WAVES 4            // waves per SIMD the one-card kernel is compiled for (diagnostic builds: 5, 6, 8)
#endif
#if TK_STEP_WAVES == 4             // (the kernel's VGPR bucket and its last register: TK_KERNEL / TK_VGPR_TOP, tarok_device.h)
#define TK_STEP_VGPRS 128
#define TK_STEP_VTOP 127
#elif TK_STEP_WAVES == 5
#define TK_STEP_VGPRS 96
#define TK_STEP_VTOP 95
#elif TK_STEP_WAVES == 6
#define TK_STEP_VGPRS 80
#define TK_STEP_VTOP 79
#else
#define TK_STEP_VGPRS 64
#define TK_STEP_VTOP 63
#endif
// LAZY: the env deals the lines its one-card launches empty in bulk (TAROK_OPT_LAZY_REFILL, refill_role<true>); the other
// instantiation carries none of that — where the batch streams, a dozen instructions per step wave are 3 % of a launch
template <bool RANDOM, bool LAZY>
TK_KERNEL(TK_BLOCK, TK_STEP_VGPRS) __attribute__((amdgpu_waves_per_eu(TK_STEP_WAVES))) void k_step(TK_STEP_ARGS) {
    TK_VGPR_TOP(TK_STEP_VGPRS, TK_STEP_VTOP);
    TkCount count = launch_count<LAZY ? 0 : 1>(epoch, play_groups);
    __shared__ u32 finq[FINQ_WORDS][TK_BLOCK];
    if (LAZY) {
        // Kind 0 (launch_count): one workgroup per group and nothing else.  It plays the group's card, then works the
        // group's own lists off: the per-launch list of the previous launch (only slots that dealt a game in place list
        // anything there: empty in nearly every launch) and, in the first launch of a stretch, the previous stretch's
        // list — every workgroup one pass of the deal on ~100-200 lanes.  (Round 3 had 256 refill workgroups beside the
        // 256 step workgroups at 65,536 games; with nothing to do in fifteen launches of sixteen they still cost every
        // launch ~0.9 us of dispatch and wave slots: profiles/r04_ab_step.txt.)
        step_role<RANDOM, true>(blockIdx.x, threadIdx.x, true, 255u, true, n, seed, offset, mix, flags, count, epoch, play_groups, fan, action_in,
                                action_out, reward, done, trick, obs, hist, s01, s23, aux, cnt, gkey, rlist, rcount, elist, finq);
        refill_role<true>(blockIdx.x, threadIdx.x, TK_BLOCK, seed, offset, mix, play_groups, count, epoch, fan, true, aux, rlist, rcount, elist);
    } else {
        // Kind 1, the streaming batches: the refill workgroups are spread among the play workgroups — block q (fan + 1)
        // works the lists of the `fan` play groups in the blocks after it off — so that their deals (instruction bound,
        // ~2.7k per game) run UNDER the play workgroups' streaming instead of after it: at the end of the grid they were a
        // 24 us tail of every launch that follows a trick's last card at 4 M games (profiles/r03_step_durations.txt).
        u32 q = blockIdx.x / (fan + 1), r = blockIdx.x % (fan + 1);
        if (r == 0)
            refill_role<false>(q, threadIdx.x, TK_BLOCK, seed, offset, mix, play_groups, count, epoch, fan, false, aux, rlist, rcount, elist);
        else
            step_role<RANDOM, false>(blockIdx.x - q - 1, threadIdx.x, true, 255u, false, n, seed, offset, mix, flags, count, epoch, play_groups, fan,
                                     action_in, action_out, reward, done, trick, obs, hist, s01, s23, aux, cnt, gkey, rlist, rcount, elist, finq);
    }
}

// Whole games in registers: deal, setup, Bot exchange, random play to the end.
TK_KERNEL(TK_BLOCK, 96) void k_rollout(int64_t n, u64 seed, u64 offset, u32 episode, int mix,
                                                     int16_t *__restrict__ scores_out, int16_t *__restrict__ nsteps_out,
                                                     int8_t *__restrict__ seats, u64 *__restrict__ masks,
                                                     uint8_t *__restrict__ actions) {
    TK_VGPR_TOP(96, 95);
    int64_t i = (int64_t)blockIdx.x * TK_BLOCK + threadIdx.x;
    if (i >= n) return;
    u64 key = game_key(seed, offset + (u64)i, episode);
    u64 h0, h1, h2, h3, tal;
    deal_thread(key, h0, h1, h2, h3, tal);
    u32 c, d, k;
    sample_setup(key, mix, c, d, k);
    Game g;
    setup_game(g, h0, h1, h2, h3, tal, c, d, k);
    if (g.phase == TK_PHASE_EXCHANGE) bot_exchange(g, key);
    u64 scores = 0;
    int played = 0;
    // one card; NT = its position in the trick, a compile-time constant (every lane starts its game
    // at a trick boundary and plays one card per step): see play_role's trick-aligned loop
    auto card = [&](auto nt_tag, int t) __attribute__((always_inline)) {
        constexpr int NT = decltype(nt_tag)::value;
        bool live = g.phase == TK_PHASE_PLAY;
        u64 m = 0;
        u32 a = 255;
        int seat = -1;
        if (live) {
            g.nt = (u32)NT;
            m = legal_now(g);
            seat = (int)((g.leader + g.nt) & 3);
            a = policy_action(key, (u32)t, m);
            u32 ti;
            apply_step<true>(g, a, scores, ti);
            played++;
        }
        if (seats) seats[(int64_t)t * n + i] = (int8_t)seat;
        if (masks) masks[(int64_t)t * n + i] = m;
        if (actions) actions[(int64_t)t * n + i] = (uint8_t)a;
    };
    for (int t = 0; t < 48; t += 4) {
        card(std::integral_constant<int, 0>{}, t);
        card(std::integral_constant<int, 1>{}, t + 1);
        card(std::integral_constant<int, 2>{}, t + 2);
        card(std::integral_constant<int, 3>{}, t + 3);
    }
    if (scores_out) reinterpret_cast<u64 *>(scores_out)[i] = scores;
    if (nsteps_out) nsteps_out[i] = (int16_t)played;
}

// Observation features for the seat to move, 256 x bf16 per game (0.0 / 1.0), for a policy
// network (the feature set is the build's own: the reference's encoder belongs
// to its LSTM agent, Igralec.py:453-543).  Four 64-wide regions, each a 54-bit card set followed
// by 10 flag bits:
//   [  0, 64) own hand            | contract one-hot (10)
//   [ 64,128) legal cards (mozne) | declarer seat relative to the mover one-hot (4), cards on
//                                   the table one-hot (4), mover is on the declarer's team, Tri/Dve/Ena
//   [128,192) cards on the table  | called-king suit one-hot (4), trick number in binary (4), 0, 0
//   [192,256) cards already taken | game live, 0 ...
// Each thread builds the four 64-bit words of its own game; then the wave writes one game per
// iteration: lane L expands bits 4L..4L+3 into 4 bf16 and the 64 lanes store one contiguous
// 512-byte row (v_readlane broadcasts the words), so the 33 MB/step of features leave as full lines.
TK_KERNEL(TK_BLOCK, 64) void k_observe(int64_t n, const ulonglong2 *__restrict__ s01,
                                                     const ulonglong2 *__restrict__ s23, uint2 *__restrict__ out) {
    TK_VGPR_TOP(64, 63);
    int64_t i = (int64_t)blockIdx.x * TK_BLOCK + threadIdx.x;
    bool valid = i < n;
    Game g;
    load_game(g, s01, s23, valid ? i : n - 1);
    u32 seat = (g.leader + g.nt) & 3;
    bool live = g.phase == TK_PHASE_PLAY;
    u64 on_table = 0;
    for (u32 j = 0; j < g.nt; j++) on_table |= 1ULL << ((g.trick >> (6 * j)) & 63);
    u64 e0 = hand_of(g, seat) | ((u64)(1u << g.contract) << 54);
    u64 f1 = (u64)(1u << ((g.declarer - seat) & 3)) | ((u64)(1u << g.nt) << 4) | ((u64)((g.team >> seat) & 1) << 8) |
             ((u64)(has_king(g.contract) ? 1u : 0u) << 9);
    u64 e1 = (live ? legal_now(g) : 0) | (f1 << 54);
    u64 f2 = (has_king(g.contract) ? (u64)(1u << g.king) : 0) | ((u64)g.trick_no << 4);
    u64 e2 = on_table | (f2 << 54);
    u64 e3 = (g.C & ~talon_unowned(g) & ~on_table) | ((u64)(live ? 1u : 0u) << 54);
    u32 lane = __lane_id();
    u32 region = lane >> 4, shift = (lane & 15) * 4;
    int64_t wave_base = i - lane;
    for (int l = 0; l < 64; l++) {
        if (wave_base + l >= n) break;                       // wave-uniform
        u32 lo0 = (u32)__builtin_amdgcn_readlane((int)(u32)e0, l), hi0 = (u32)__builtin_amdgcn_readlane((int)(u32)(e0 >> 32), l);
        u32 lo1 = (u32)__builtin_amdgcn_readlane((int)(u32)e1, l), hi1 = (u32)__builtin_amdgcn_readlane((int)(u32)(e1 >> 32), l);
        u32 lo2 = (u32)__builtin_amdgcn_readlane((int)(u32)e2, l), hi2 = (u32)__builtin_amdgcn_readlane((int)(u32)(e2 >> 32), l);
        u32 lo3 = (u32)__builtin_amdgcn_readlane((int)(u32)e3, l), hi3 = (u32)__builtin_amdgcn_readlane((int)(u32)(e3 >> 32), l);
        u32 lo = region == 0 ? lo0 : (region == 1 ? lo1 : (region == 2 ? lo2 : lo3));
        u32 hi = region == 0 ? hi0 : (region == 1 ? hi1 : (region == 2 ? hi2 : hi3));
        u32 nib = ((shift < 32 ? lo >> shift : hi >> (shift - 32))) & 15u;
        uint2 v;
        v.x = ((nib & 1) ? 0x3F80u : 0u) | ((nib & 2) ? 0x3F800000u : 0u);
        v.y = ((nib & 4) ? 0x3F80u : 0u) | ((nib & 8) ? 0x3F800000u : 0u);
        out[(wave_base + l) * 64 + lane] = v;
    }
}

// ---------------------------------------------------------------------------
// The reference's OWN observation layout: what Nevronski_igralec builds for the
// seat to move in stanje_v_vektor_rek_navadna (Igralec.py:453-533), as one dense record of 0/1
// bytes per game (offsets TAROK_REF_* in tarok_env.h; a consumer slices views out of it):
//   opponents' history [56][3][54]  row i = the i-th card PLAYED in the game (talon entries of the
//                                   history do not take a row): one-hot card in the channel of the
//                                   opponent who played it, channels = the other seats in seat order
//                                   (igralci2index, Igralec.py:271-274)                      :508-510
//   own-hand history   [56][54]     row i of an own play = the hand vector before that card, which
//                                   starts from the hand as DEALT (zacetna_roka: the exchange is not
//                                   applied to it, discards stay set)                        :465,511-514
//   talon              [6][55]      Tri..Solo_ena: row r = one-hot of the r-th talon card, column 54 =
//                                   "in the chosen group" (:499-507); Klop: a flat 54-vector of the
//                                   talon cards gifted so far (:497-498); Berac: nothing (:487-488)
//   king one-hot [4] (:467-470), declarer-relative index one-hot [4] (:493-494, self = 3),
//   discards [54] (only in the view of the player who exchanged, :462-464), legal cards [54] (:518-519)
// plus meta = {T, network type, rows used, seat}: T = entries of the history — played cards + the
// "Talon" entry + Klop's talon gifts: the counter at :456-458 counts every entry — rounded up to the
// next multiple of 8, plus 8 when already one (:460); the reference's tensors have T rows, the
// record always 56 (rows >= rows-used are zero either way).
// Parity unpinned: the reference holds no fixture for this layout and Igralec.py cannot be imported
// (pytorch_lightning, torch_models); tested against a line-cited restatement (oracle/encoder_spec.py).
//
// Workgroup = 64 games, 256 threads.  Phase 1, one thread per game (one wave): history bytes -> LDS, trick leaders replayed from the cards (the
// winner rule of apply_step), initial hand and discards recovered from the planes, the 448 bits of
// the small fields.  Phase 2, one wave per game at a time: lane t builds row t (an exclusive
// prefix-OR over the lanes gives "own cards played before"), then the 64 lanes write the 12,544-byte
// record as 784 16-byte chunks, each computed from the row descriptors: full-line stores.
#define OR_ROWS 56
#define OR_OWN_OFF (OR_ROWS * 162)
#define OR_SMALL_OFF (OR_OWN_OFF + OR_ROWS * 54)
#define OR_REC (OR_SMALL_OFF + 448)
static_assert(OR_REC == TAROK_REF_RECORD_BYTES && OR_OWN_OFF == TAROK_REF_OWN && OR_SMALL_OFF == TAROK_REF_TALON, "record layout");

struct RefDesc { u64 h0; u64 small[7]; u32 leaders, plays, me, pad; };

// 4 bits -> 4 bytes of 0/1 (the four shifted copies do not overlap: no carries)
__device__ __forceinline__ u32 nibble_bytes(u32 x) { return ((x & 15u) * 0x00204081u) & 0x01010101u; }
__device__ __forceinline__ uint4 bits16_bytes(u32 b) {
    return make_uint4(nibble_bytes(b), nibble_bytes(b >> 4), nibble_bytes(b >> 8), nibble_bytes(b >> 12));
}
// OR `val` (WIDTH <= 64 bits) into a little-endian bit vector at the compile-time bit offset OFF
template <int OFF, int NW> __device__ __forceinline__ void bits_insert(u64 (&w)[NW], u64 val) {
    w[OFF / 64] |= val << (OFF % 64);
    if constexpr (OFF % 64 != 0 && OFF / 64 + 1 < NW) w[OFF / 64 + 1] |= val >> (64 - OFF % 64);
}
// set bit `b` (run-time index) of a bit vector kept in registers
template <int NW> __device__ __forceinline__ void bits_set(u64 (&w)[NW], u32 b) {
#pragma unroll
    for (int k = 0; k < NW; k++) w[k] |= ((b >> 6) == (u32)k) ? (1ULL << (b & 63)) : 0ULL;
}
__device__ __forceinline__ u32 ref_type(u32 c) {    // Nevronski_igralec.tip_igre_v_tip_izbire (Igralec.py:180-189), Tipi_NN values
    return c == TK_KLOP ? 0u : (has_king(c) ? 1u : ((c == TK_BERAC || c == TK_ODPRTI_BERAC) ? 3u : 2u));
}

#define OR_GAMES 64                 // games per workgroup: phase 1 on one wave, phase 2 on all four (16 games each)
TK_KERNEL(256, 64) void k_observe_ref(int64_t n, const ulonglong2 *__restrict__ s01,
                                                   const ulonglong2 *__restrict__ s23, const uint8_t *__restrict__ hist,
                                                   uint4 *__restrict__ rec, int4 *__restrict__ meta) {
    TK_VGPR_TOP(64, 63);
    __shared__ RefDesc desc[OR_GAMES];
    __shared__ uint8_t hist_s[OR_GAMES][48];
    __shared__ uint8_t rowpos_s[4][64];
    __shared__ u64 rowmask_s[4][64];
    u32 tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    int64_t base = (int64_t)blockIdx.x * OR_GAMES;
    if (tid < OR_GAMES)
    {   // ---- phase 1
        int64_t i = base + tid;
        bool valid = i < n;
        int64_t ic = valid ? i : n - 1;
        Game g;
        load_game(g, s01, s23, ic);
        bool live = valid && g.phase == TK_PHASE_PLAY;
        u32 me = (g.leader + g.nt) & 3;
        u32 plays = live ? g.trick_no * 4 + g.nt : 0u;
        bool berac = g.contract == TK_BERAC || g.contract == TK_ODPRTI_BERAC;
        u32 lead = berac ? g.declarer : 0u;                               // Berac.py:15, Klop.py:26, Navadna_igra.py:70
        u32 leaders = 0;
        u64 played_all = 0, played_me = 0;
        for (u32 t = 0; t < 12; t++) {
            if (t * 4 >= plays) break;
            leaders |= lead << (2 * t);
            u32 c[4];
#pragma unroll
            for (u32 j = 0; j < 4; j++) {
                u32 idx = t * 4 + j;
                bool has = idx < plays;
                c[j] = has ? (u32)hist[(int64_t)idx * n + ic] & 63u : 0u;
                hist_s[tid][idx] = (uint8_t)c[j];
                u64 bit = has ? (1ULL << c[j]) : 0ULL;
                played_all |= bit;
                played_me |= (((lead + j) & 3) == me) ? bit : 0ULL;
            }
            if (t * 4 + 4 <= plays) {                                     // pobere_stih (Klop.py:81-94)
                u32 w = 0, cw = c[0];
#pragma unroll
                for (u32 j = 1; j < 4; j++) {
                    u32 sw = min(cw >> 3, 4u), si = min(c[j] >> 3, 4u);
                    bool beats = (sw == si) ? (cw < c[j]) : (si == 4);
                    w = beats ? j : w;
                    cw = beats ? c[j] : cw;
                }
                lead = (lead + w) & 3;
            }
        }
        // zacetna_roka (Igralec.py:264): the hand as dealt.  hand now + own cards played = the hand after
        // the exchange = (dealt + chosen group) - discards; the discards are what lies in the exchanging
        // player's pile without ever having been played (Igralec.py:376-377)
        u64 h_after = hand_of(g, me) | played_me;
        u64 h0 = h_after, disc = 0;
        bool exch = has_exchange(g.contract);
        u32 gs = exch ? group_size(g.contract) : 1u;
        if (exch && me == g.declarer) {
            u64 grp = ids_mask(g.talon, (int)(g.tl * gs), (int)gs);
            disc = seat_cards(g, g.declarer) & g.C & ~played_all;
            h0 = (h_after & ~grp) | (disc & ~grp);
        }
        u64 small[7] = {0, 0, 0, 0, 0, 0, 0};
        if (live) {
            if (exch) {                                                   // the ("Talon", ...) entry, :499-507
#pragma unroll
                for (u32 r = 0; r < 6; r++) {
                    bits_set(small, r * 55 + (u32)((g.talon >> (6 * r)) & 63));
                    if (r / gs == g.tl) bits_set(small, r * 55 + 54);
                }
            } else if (g.contract == TK_KLOP) {                           // (None, talon card) entries, :497-498
#pragma unroll
                for (u32 r = 0; r < 6; r++)
                    if (r >= g.tl) bits_set(small, (u32)((g.talon >> (6 * r)) & 63));
            }
            if (has_king(g.contract)) bits_insert<330>(small, 1ULL << g.king);                       // :467-470
            u32 didx = g.declarer == me ? 3u : (g.declarer < me ? g.declarer : g.declarer - 1);     // :271-274, 449-451
            if (g.contract != TK_KLOP) bits_insert<334>(small, 1ULL << didx);                         // (Klop's list has no index, :526-527)
            bits_insert<338>(small, disc);
            bits_insert<392>(small, legal_now(g));
        }
        RefDesc &d = desc[tid];
        d.h0 = live ? h0 : 0;
#pragma unroll
        for (int k = 0; k < 7; k++) d.small[k] = small[k];
        d.leaders = leaders; d.plays = plays; d.me = me;
        if (valid && meta) {
            u32 entries = plays + (exch ? 1u : 0u) + (g.contract == TK_KLOP ? 6u - g.tl : 0u);       // :455-458
            u32 T = entries + (8 - entries % 8);                                                      // :460
            meta[i] = make_int4(live ? (int)T : 0, (int)ref_type(g.contract), (int)plays, (int)me);
        }
    }
    __syncthreads();
    // ---- phase 2: this wave's 16 games, one at a time (four waves per 64 games: at 65,536 games that is four
    // waves per SIMD to hide the stores' issue latency behind each other)
    for (u32 k = 0; k < OR_GAMES / 4; k++) {
        u32 gl = wave * (OR_GAMES / 4) + k;
        int64_t gidx = base + gl;
        if (gidx >= n) break;                                             // wave uniform
        const RefDesc &d = desc[gl];
        u32 me = d.me, plays = d.plays;
        u32 pos = 255;
        u64 mybit = 0;
        if (lane < plays) {
            u32 c = hist_s[gl][lane];
            u32 seat = (((d.leaders >> (2 * (lane >> 2))) & 3) + (lane & 3)) & 3;
            if (seat == me) mybit = 1ULL << c;
            else pos = (seat < me ? seat : seat - 1) * 54 + c;
        }
        u32 lo = (u32)mybit, hi = (u32)(mybit >> 32);                     // inclusive prefix OR over the rows
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            u32 l2 = (u32)__shfl_up((int)lo, o), h2 = (u32)__shfl_up((int)hi, o);
            if ((int)lane >= o) { lo |= l2; hi |= h2; }
        }
        u64 before = TK_U64(lo, hi) & ~mybit;                             // own cards played before this row
        rowpos_s[wave][lane] = (uint8_t)pos;
        rowmask_s[wave][lane] = mybit ? (d.h0 & ~before) : 0ULL;          // :512 (the hand before the card leaves it)
        __builtin_amdgcn_s_waitcnt(0xC07F);                               // lgkmcnt(0): the wave's LDS writes have landed
        __builtin_amdgcn_wave_barrier();
        for (u32 ch = lane; ch < OR_REC / 16; ch += 64) {
            u32 o = ch * 16;
            uint4 v = make_uint4(0, 0, 0, 0);
            if (o < OR_OWN_OFF) {
                u32 t0 = o / 162;
#pragma unroll
                for (u32 q = 0; q < 2; q++) {
                    u32 t = t0 + q;
                    u32 pp = rowpos_s[wave][t & 63];
                    u32 idx = t * 162 + pp - o;
                    if (t < 64 && pp < 162 && idx < 16) {
                        u32 b = 1u << ((idx & 3) * 8);
                        v.x |= (idx >> 2) == 0 ? b : 0; v.y |= (idx >> 2) == 1 ? b : 0;
                        v.z |= (idx >> 2) == 2 ? b : 0; v.w |= (idx >> 2) == 3 ? b : 0;
                    }
                }
            } else if (o < OR_SMALL_OFF) {
                u32 rel = o - OR_OWN_OFF, t0 = rel / 54, b0 = rel - t0 * 54;
                u64 m = (rowmask_s[wave][t0] >> b0) | (rowmask_s[wave][(t0 + 1) & 63] << (54 - b0));
                v = bits16_bytes((u32)m & 0xFFFFu);
            } else {
                u32 rel = o - OR_SMALL_OFF;
                v = bits16_bytes((u32)(d.small[rel >> 6] >> (rel & 63)) & 0xFFFFu);
            }
            typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
            u32x4 vv = {v.x, v.y, v.z, v.w};                         // (822 MB per call at 65,536 games: nothing re-reads it from L2)
            TK_STREAM_STORE(reinterpret_cast<u32x4 *>(rec) + gidx * (OR_REC / 16) + ch, vv);
        }
        __builtin_amdgcn_wave_barrier();                                  // the row buffers are rewritten by the next game
    }
}

// A block's per-game bit vectors (NW words each, in LDS) written as 0/1 bytes, BYTES per game (a multiple
// of 8), wave-cooperatively: a game's record leaves as consecutive 8-byte stores of neighbouring lanes.
template <int NW, int BYTES>
__device__ __forceinline__ void write_bit_records(const u64 (*words)[NW], int64_t base, int64_t n, uint2 *__restrict__ out) {
    static_assert(BYTES % 8 == 0 && BYTES <= NW * 64, "record size");
    u32 lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (u32 k = 0; k < 64; k++) {
        u32 gl = wave * 64 + k;
        int64_t gidx = base + gl;
        if (gidx >= n) break;
        for (u32 ch = lane; ch < BYTES / 8; ch += 64) {
            u32 b = (u32)(words[gl][ch >> 3] >> ((ch & 7) * 8)) & 255u;
            out[gidx * (BYTES / 8) + ch] = make_uint2(nibble_bytes(b), nibble_bytes(b >> 4));
        }
    }
}

// menjaj_talon_v_vektor (Igralec.py:535-543), the input of the exchange decision, for the games that
// wait for tarok_exchange: [roka 54 | talon (54,6) flattened card-major | igra one-hot 15 | pad 7]
// = 400 bytes of 0/1; zeros for games in any other phase.
TK_KERNEL(TK_BLOCK, 64) void k_observe_exchange_ref(int64_t n, const ulonglong2 *__restrict__ s01,
                                                                  const ulonglong2 *__restrict__ s23, uint2 *__restrict__ out) {
    TK_VGPR_TOP(64, 63);
    __shared__ u64 words[TK_BLOCK][7];
    int64_t base = (int64_t)blockIdx.x * TK_BLOCK, i = base + threadIdx.x;
    Game g;
    load_game(g, s01, s23, i < n ? i : n - 1);
    u64 w[7] = {0, 0, 0, 0, 0, 0, 0};
    if (i < n && g.phase == TK_PHASE_EXCHANGE) {
        u32 gs = group_size(g.contract);
        bits_insert<0>(w, hand_of(g, g.declarer));                        // :540 (the hand before the exchange)
#pragma unroll
        for (u32 r = 0; r < 6; r++) bits_set(w, 54 + (u32)((g.talon >> (6 * r)) & 63) * 6 + r / gs);   // :541-542
        // igra_zalozi2index (Igralec.py:717-745): Tri/Dve/Ena x suit -> 0..11, Solo_tri/dve/ena -> 12..14
        u32 idx = has_king(g.contract) ? (g.contract - 1) * 4 + g.king : 12 + (g.contract - TK_SOLO_TRI);
        bits_set(w, 378 + idx);                                           // :539
    }
#pragma unroll
    for (int k = 0; k < 7; k++) words[threadIdx.x][k] = w[k];
    __syncthreads();
    write_bit_records<7, TAROK_REF_EXCHANGE_BYTES>(words, base, n, out);
}

// The bidding input (pripavi_licitiram, Igralec.py:278-281): every seat's hand as 54 bytes of 0/1,
// [N][4][54].
TK_KERNEL(TK_BLOCK, 64) void k_observe_hands_ref(int64_t n, const ulonglong2 *__restrict__ s01,
                                                               const ulonglong2 *__restrict__ s23, uint2 *__restrict__ out) {
    TK_VGPR_TOP(64, 63);
    __shared__ u64 words[TK_BLOCK][4];
    int64_t base = (int64_t)blockIdx.x * TK_BLOCK, i = base + threadIdx.x;
    Game g;
    load_game(g, s01, s23, i < n ? i : n - 1);
    u64 w[4] = {0, 0, 0, 0};
    if (i < n) {
        bits_insert<0>(w, hand_of(g, 0)); bits_insert<54>(w, hand_of(g, 1));
        bits_insert<108>(w, hand_of(g, 2)); bits_insert<162>(w, hand_of(g, 3));
    }
#pragma unroll
    for (int k = 0; k < 4; k++) words[threadIdx.x][k] = w[k];
    __syncthreads();
    write_bit_records<4, 216>(words, base, n, out);
}

// Masked categorical sample from policy logits, one thread per game: softmax over the legal
// cards only (mask = observation word), inverse-CDF draw with the spec RNG (draw 192 + cards
// played), log-probability of the drawn card.  Replaces ~10 framework kernels per step
// (bit-expand mask, masked_fill, log_softmax, multinomial, gather) with one pass over 8 MB.
__device__ __forceinline__ float bf16_to_f32(u32 h) { return __uint_as_float(h << 16); }

TK_KERNEL(TK_BLOCK, 256) void k_sample(int64_t n, const uint4 *__restrict__ logits /* [N,64] bf16 */,
                                                    const u64 *__restrict__ obs, const u64 *__restrict__ gkey,
                                                    uint8_t *__restrict__ action, float *__restrict__ logp) {
    TK_VGPR_TOP(256, 255);
    int64_t i = (int64_t)blockIdx.x * TK_BLOCK + threadIdx.x;
    if (i >= n) return;
    u64 o = obs[i];
    u64 m = o & TAROK_OBS_MASK;
    if (!m) { action[i] = 255; if (logp) logp[i] = 0.f; return; }
    float l[56];
#pragma unroll
    for (int q = 0; q < 7; q++) {
        uint4 v = logits[i * 8 + q];
        u32 w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
        for (int k = 0; k < 4; k++) {
            l[q * 8 + 2 * k] = bf16_to_f32(w[k] & 0xFFFFu);
            l[q * 8 + 2 * k + 1] = bf16_to_f32(w[k] >> 16);
        }
    }
    float mx = -3.0e38f;
#pragma unroll
    for (int c = 0; c < 54; c++) mx = ((m >> c) & 1) ? fmaxf(mx, l[c]) : mx;
    float sum = 0.f;
#pragma unroll
    for (int c = 0; c < 54; c++) {
        float e = ((m >> c) & 1) ? __expf(l[c] - mx) : 0.f;
        l[c] = e;                      // keep exp() for the draw
        sum += e;
    }
    u32 r = rng32(gkey[i], 192u + ((u32)(o >> TAROK_OBS_STEP_SHIFT) & 63u));
    float u = ((float)(r >> 8) + 0.5f) * (1.0f / 16777216.0f) * sum;
    float acc = 0.f, pe = 0.f;
    int pickc = -1;
#pragma unroll
    for (int c = 0; c < 54; c++) {
        bool legal = (m >> c) & 1;
        acc += l[c];
        bool take = legal && pickc < 0 && acc > u;
        pe = take ? l[c] : pe;
        pickc = take ? c : pickc;
    }
    if (pickc < 0) {                   // rounding at the top end: the last legal card
        pickc = 63 - __clzll(m);
        pe = l[53];
#pragma unroll
        for (int c = 0; c < 54; c++) pe = (c == pickc) ? l[c] : pe;
    }
    action[i] = (uint8_t)pickc;
    if (logp) logp[i] = __logf(pe / sum);
}


// ---------------------------------------------------------------------------
// Learner side: a minibatch of network inputs from the rollout's 32-byte feature words
// (tarok_policy_mlp / tarok_policy_step feature_words_out): out[j] = the 256 bf16 0/1 features
// of sample index[j] (or of sample j when index is NULL).  One 16-byte chunk (one byte of a
// feature word -> 8 bf16) per thread, 32 threads per sample: the gather and the expansion in one
// pass, full 512-byte rows written per 32 lanes.
TK_KERNEL(TK_BLOCK, 64) void k_expand_features(int64_t n, const u64 *__restrict__ words /* [.,4] */,
                                                             const int64_t *__restrict__ index, uint4 *__restrict__ out) {
    TK_VGPR_TOP(64, 63);
    int64_t t = (int64_t)blockIdx.x * TK_BLOCK + threadIdx.x;
    int64_t j = t >> 5;
    if (j >= n) return;
    u32 chunk = (u32)t & 31;
    int64_t src = index ? index[j] : j;
    u32 byte = reinterpret_cast<const uint8_t *>(words + src * 4)[chunk];
    uint4 v;
    v.x = ((byte & 1) ? 0x3F80u : 0u) | ((byte & 2) ? 0x3F800000u : 0u);
    v.y = ((byte & 4) ? 0x3F80u : 0u) | ((byte & 8) ? 0x3F800000u : 0u);
    v.z = ((byte & 16) ? 0x3F80u : 0u) | ((byte & 32) ? 0x3F800000u : 0u);
    v.w = ((byte & 64) ? 0x3F80u : 0u) | ((byte & 128) ? 0x3F800000u : 0u);
    out[j * 32 + chunk] = v;
}

// ---------------------------------------------------------------------------
// Learner side of the policy step: the clipped-surrogate policy-gradient loss
// over the LEGAL cards, forward and gradient in one pass over the head outputs.  Replaces ~40
// framework elementwise kernels per minibatch (bit-expand of the mask, masked_fill, log_softmax,
// gather, exp, clamp, min, entropy, their backward kernels) with one read of [B,64] bf16 and one
// write of the same shape.
//   out [B,64] bf16: columns 0..53 card logits, column 54 the state value.
//   per sample: pi = -min(r A, clamp(r, 1-eps, 1+eps) A) with r = exp(logp[a] - logp_old),
//               v = (value - ret)^2,  H = -sum p log p over the legal cards;
//   loss = sum_i w_i (pi_i + vf v_i - ent H_i) / wsum.
// dout = d loss / d out (bf16); part[block] = {sum w pi, sum w v, sum w H, 0} (f32, unscaled).
TK_KERNEL(TK_BLOCK, 256) void k_ppo_loss(int64_t n, const uint4 *__restrict__ out, const u64 *__restrict__ words,
                                                      const int64_t *__restrict__ act, const float *__restrict__ logp_old,
                                                      const float *__restrict__ adv, const float *__restrict__ ret,
                                                      const float *__restrict__ weight, float clip, float vf_coef,
                                                      float ent_coef, const float *__restrict__ inv_wsum_p, uint4 *__restrict__ dout,
                                                      float4 *__restrict__ part) {
    TK_VGPR_TOP(256, 255);
    __shared__ float red[3][TK_BLOCK / 64];
    float inv_wsum = *inv_wsum_p;
    int64_t i = (int64_t)blockIdx.x * TK_BLOCK + threadIdx.x;
    float s_pi = 0.f, s_v = 0.f, s_h = 0.f;
    if (i < n) {
        float l[64];
#pragma unroll
        for (int q = 0; q < 8; q++) {
            uint4 v = out[i * 8 + q];
            u32 wv[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
            for (int k = 0; k < 4; k++) {
                l[q * 8 + 2 * k] = bf16_to_f32(wv[k] & 0xFFFFu);
                l[q * 8 + 2 * k + 1] = bf16_to_f32(wv[k] >> 16);
            }
        }
        u64 m = words[i] & TAROK_OBS_MASK;
        float w = weight[i];
        if (!m) { m = 1; w = 0.f; }                        // nothing to play: no contribution, finite arithmetic
        float ws = w * inv_wsum;
        u32 a = (u32)act[i];
        a = a < 54 ? a : 53u;                              // (rows without a card carry weight 0)
        float value = l[54];
        float mx = -3.0e38f;
#pragma unroll
        for (int c = 0; c < 54; c++) mx = ((m >> c) & 1) ? fmaxf(mx, l[c]) : mx;
        float sum = 0.f;
#pragma unroll
        for (int c = 0; c < 54; c++) {
            float e = ((m >> c) & 1) ? __expf(l[c] - mx) : 0.f;
            l[c] = e;
            sum += e;
        }
        float ls = __logf(sum), inv = 1.0f / sum;
        // probabilities in place; entropy; log-prob of the played card
        float H = 0.f, la = 0.f;
#pragma unroll
        for (int c = 0; c < 54; c++) {
            float p = l[c] * inv;
            float lp = p > 0.f ? __logf(l[c]) - ls : 0.f;  // log p_c (0 where p_c = 0: p log p -> 0)
            H -= p * lp;
            la = (c == (int)a) ? __logf(fmaxf(l[c], 1e-38f)) - ls : la;
            l[c] = p;
        }
        float r = __expf(la - logp_old[i]);
        float A = adv[i];
        float rc = fminf(fmaxf(r, 1.0f - clip), 1.0f + clip);
        float s1 = r * A, s2 = rc * A;
        bool first = s1 <= s2;                              // which branch of the min is active
        bool inside = r > 1.0f - clip && r < 1.0f + clip;
        float g = (first || inside) ? -A * r : 0.f;         // d pi / d logp[a]
        float dv = value - ret[i];
        s_pi = w * (-fminf(s1, s2));
        s_v = w * dv * dv;
        s_h = w * H;
        // d loss / d logit_c = ws [ g (delta_ca - p_c) + ent p_c (log p_c + H) ] on the legal cards
#pragma unroll
        for (int c = 0; c < 54; c++) {
            float p = l[c];
            float lp = p > 0.f ? __logf(p) : 0.f;
            float d = g * (((c == (int)a) ? 1.0f : 0.0f) - p) + ent_coef * p * (lp + H);
            l[c] = ((m >> c) & 1) ? ws * d : 0.f;
        }
        l[54] = ws * vf_coef * 2.0f * dv;
#pragma unroll
        for (int c = 55; c < 64; c++) l[c] = 0.f;
#pragma unroll
        for (int q = 0; q < 8; q++) {
            uint4 v;
            u32 wv[4];
#pragma unroll
            for (int k = 0; k < 4; k++) {
                __bf16 lo = (__bf16)l[q * 8 + 2 * k], hi = (__bf16)l[q * 8 + 2 * k + 1];
                wv[k] = (u32)__builtin_bit_cast(unsigned short, lo) | ((u32)__builtin_bit_cast(unsigned short, hi) << 16);
            }
            v.x = wv[0]; v.y = wv[1]; v.z = wv[2]; v.w = wv[3];
            dout[i * 8 + q] = v;
        }
    }
    // block sums of the three loss terms (fixed order: the result does not depend on the schedule)
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) {
        s_pi += __shfl_xor(s_pi, o); s_v += __shfl_xor(s_v, o); s_h += __shfl_xor(s_h, o);
    }
    if ((threadIdx.x & 63) == 0) { red[0][threadIdx.x >> 6] = s_pi; red[1][threadIdx.x >> 6] = s_v; red[2][threadIdx.x >> 6] = s_h; }
    __syncthreads();
    if (threadIdx.x == 0) {
        float a0 = 0.f, a1 = 0.f, a2 = 0.f;
        for (int k = 0; k < TK_BLOCK / 64; k++) { a0 += red[0][k]; a1 += red[1][k]; a2 += red[2][k]; }
        part[blockIdx.x] = make_float4(a0, a1, a2, 0.f);
    }
}

// ---------------------------------------------------------------------------
// Fused policy step for a learner: observation features -> MLP 256-256-256-64
// (bf16 MFMA, f32 accumulate) -> masked categorical sample, in ONE launch.  The features never
// leave the chip (built in LDS from the 32-byte packed state), the activations go LDS -> MFMA ->
// LDS, only action / log-prob / value (+ optionally the features, for the learner's update) are
// written.  Replaces tarok_observe + 4 framework GEMMs + bias/ReLU kernels + tarok_sample_policy
// (~150 MB of activation traffic per 65,536 games) — the one place in this build where work is
// GEMM shaped, so the one place that uses the matrix cores.
//
// Workgroup = 256 threads = 4 waves, 128 games, ONE activation buffer X[game][feature] in LDS
// (row stride 264 bf16 = 528 B) updated in place, 72 KB per workgroup: two workgroups share a CU
// (two waves per SIMD), so one's feature build / epilogues / sampling run under the other's MFMAs.
// Every layer is computed TRANSPOSED, D[feature][game] = W[feature][k] * X^T[k][game], with
// v_mfma_f32_32x32x16_bf16: lane l (r = l & 31, h = l >> 5) supplies A[row r][k = 8h..8h+7] = one
// 16-byte piece of the weight (pre-arranged in fragment order: a wave's load is 1 KiB contiguous)
// and B[k = 8h..8h+7][col r] = 16 contiguous bytes of game r's row in LDS; it receives
// D[(reg & 3) + 8 (reg >> 2) + 4 h][col r]: four CONSECUTIVE features of one game per register
// quad, i.e. one 8-byte LDS store of packed bf16 per quad (the untransposed product scatters 2-byte
// stores).  Layers 1-2: wave w owns features [64w, 64w+64) x all 128 games (2 x 4 tiles, 128 MFMAs,
// the weight slab streamed from L2 four k-steps ahead); layer 3 (64 outputs = 54 card logits, value
// in column 54): wave w owns games [32w, 32w+32).
#if TK_BLOCK == 256                // the MLP kernels are laid out for 256-slot groups
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;
#define PM_M 128
// the eight bits of a byte as eight bf16 0.0 / 1.0 (16 bytes): bit k of b lands on bit 8k of (b * 0x204081) (copies of b at shifts
// 0, 7, 14, 21), a byte permute puts two of those bytes into the halves of a word, times 0x3F80 — 13 instructions per byte where
// a select per bit takes 28 (the learner's kernels expand their feature words with it too)
__device__ __forceinline__ uint4 tk_expand_byte(u32 b) {
    u32 lo = __umul24(b & 15u, 0x204081u) & 0x01010101u;        // byte k = bit k
    u32 hi = __umul24(b >> 4, 0x204081u) & 0x01010101u;         // byte k = bit 4 + k
    uint4 v;
    v.x = __umul24(__builtin_amdgcn_perm(0u, lo, 0x0C010C00u), 0x3F80u);   // bytes {lo.0, 0, lo.1, 0}
    v.y = __umul24(__builtin_amdgcn_perm(0u, lo, 0x0C030C02u), 0x3F80u);
    v.z = __umul24(__builtin_amdgcn_perm(0u, hi, 0x0C010C00u), 0x3F80u);
    v.w = __umul24(__builtin_amdgcn_perm(0u, hi, 0x0C030C02u), 0x3F80u);
    return v;
}
#define PM_LD 264
#define PM_LL 68             // f32 logits row stride (272 B: float4 stores stay aligned)

// the weight fragments of a wave's first four k-steps (issued early by the caller: before the
// feature build for layer 1, before the previous layer's epilogue for layer 2)
__device__ __forceinline__ void mlp_prefetch(bf16x8 (&wq)[4][2], const __bf16 *__restrict__ w, u32 first_tile) {
    const bf16x8 *wf = reinterpret_cast<const bf16x8 *>(w) + (size_t)first_tile * 16 * 64 + __lane_id();
#pragma unroll
    for (int d = 0; d < 4; d++)
#pragma unroll
        for (int ft = 0; ft < 2; ft++) wq[d][ft] = wf[(ft * 16 + d) * 64];
    __builtin_amdgcn_sched_barrier(0);
}

// (X: the 128-game tile of this wave; wave = 0..3 within the tile; the barriers are workgroup-wide)
__device__ __forceinline__ void mlp_hidden(__bf16 *__restrict__ X, const __bf16 *__restrict__ w, const float *__restrict__ bias,
                                           bf16x8 (&wq)[4][2], u32 wave) {
    u32 lane = __lane_id(), r = lane & 31, h = lane >> 5;
    const bf16x8 *wf = reinterpret_cast<const bf16x8 *>(w) + (size_t)(wave * 2) * 16 * 64 + lane;
    f32x16 acc[2][4];
#pragma unroll
    for (int ft = 0; ft < 2; ft++)
#pragma unroll
        for (int gt = 0; gt < 4; gt++)
#pragma unroll
            for (int j = 0; j < 16; j++) acc[ft][gt][j] = 0.f;
    bf16x8 xn[4];                                      // activations one k-step ahead (wq: weights, 4 k-steps in flight)
#pragma unroll
    for (int gt = 0; gt < 4; gt++) xn[gt] = *reinterpret_cast<const bf16x8 *>(X + (32 * gt + r) * PM_LD + 8 * h);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int kk = 0; kk < 16; kk++) {
        bf16x8 x[4] = {xn[0], xn[1], xn[2], xn[3]};
        bf16x8 wc[2] = {wq[kk & 3][0], wq[kk & 3][1]};
        if (kk + 4 < 16) {
#pragma unroll
            for (int ft = 0; ft < 2; ft++) wq[kk & 3][ft] = wf[(ft * 16 + kk + 4) * 64];
        }
        if (kk < 15) {
#pragma unroll
            for (int gt = 0; gt < 4; gt++) xn[gt] = *reinterpret_cast<const bf16x8 *>(X + (32 * gt + r) * PM_LD + 16 * (kk + 1) + 8 * h);
        }
#pragma unroll
        for (int ft = 0; ft < 2; ft++)
#pragma unroll
            for (int gt = 0; gt < 4; gt++) acc[ft][gt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wc[ft], x[gt], acc[ft][gt], 0, 0, 0);
        // keep the loads issued in this k-step here: left alone, the scheduler sinks each weight
        // load to just before its use (to save registers) and every k-step waits out an L2 round trip
        __builtin_amdgcn_sched_barrier(0);
    }
    __syncthreads();                                   // every wave has read its last X fragment
#pragma unroll
    for (int ft = 0; ft < 2; ft++)
#pragma unroll
        for (int q = 0; q < 4; q++) {
            u32 f0 = wave * 64 + 32 * ft + 8 * q + 4 * h;
            float4 bv = *reinterpret_cast<const float4 *>(bias + f0);
#pragma unroll
            for (int gt = 0; gt < 4; gt++) {
                bf16x4 o;
                o[0] = (__bf16)fmaxf(acc[ft][gt][4 * q + 0] + bv.x, 0.f);
                o[1] = (__bf16)fmaxf(acc[ft][gt][4 * q + 1] + bv.y, 0.f);
                o[2] = (__bf16)fmaxf(acc[ft][gt][4 * q + 2] + bv.z, 0.f);
                o[3] = (__bf16)fmaxf(acc[ft][gt][4 * q + 3] + bv.w, 0.f);
                *reinterpret_cast<bf16x4 *>(X + (32 * gt + r) * PM_LD + f0) = o;
            }
        }
    __syncthreads();
}

// The policy step of 128 * TILES games by a workgroup of 256 * TILES threads (tile t = waves
// 4t..4t+3 and rows 128t.. of X).  TILES = 1: tarok_policy_mlp; TILES = 2: the first half of
// tarok_policy_step, which then needs the sampled cards of its 256 games in LDS (act_s).
template <int TILES>
__device__ __forceinline__ void policy_body(
    int64_t n, const ulonglong2 *__restrict__ s01, const ulonglong2 *__restrict__ s23, const u64 *__restrict__ obs,
    const u64 *__restrict__ gkey, const __bf16 *__restrict__ w1, const float *__restrict__ b1,
    const __bf16 *__restrict__ w2, const float *__restrict__ b2, const __bf16 *__restrict__ w3,
    const float *__restrict__ b3, uint8_t *__restrict__ action, float *__restrict__ logp, float *__restrict__ value,
    uint4 *__restrict__ features_out, ulonglong2 *__restrict__ feature_words_out, u64 *__restrict__ stamps,
    uint8_t *__restrict__ act_s, u32 **lds_after = nullptr /* the activation buffer: free once every thread has returned */) {
    constexpr int GAMES = PM_M * TILES;
    u64 ts[7];
#define PM_STAMP(k) if (stamps) ts[k] = __builtin_amdgcn_s_memtime();
    PM_STAMP(0)
    __shared__ __attribute__((aligned(16))) __bf16 X[GAMES * PM_LD];
    __shared__ u64 ext[GAMES][4];
    int64_t base = (int64_t)blockIdx.x * GAMES;
    u32 tid = threadIdx.x, wave4 = (tid >> 6) & 3, tile = tid >> 8;
    __bf16 *Xt = X + tile * PM_M * PM_LD;
    if (lds_after) *lds_after = reinterpret_cast<u32 *>(X);
    bf16x8 wq[4][2];
    mlp_prefetch(wq, w1, wave4 * 2);           // lands while the features are built
    // ---- the four 64-bit feature words of each game (same definition as k_observe)
    if (tid < GAMES) {
        int64_t i = base + tid < n ? base + tid : n - 1;
        Game g;
        load_game(g, s01, s23, i);
        u32 seat = (g.leader + g.nt) & 3;
        bool live = g.phase == TK_PHASE_PLAY;
        u64 on_table = 0;
        for (u32 j = 0; j < g.nt; j++) on_table |= 1ULL << ((g.trick >> (6 * j)) & 63);
        u64 f1 = (u64)(1u << ((g.declarer - seat) & 3)) | ((u64)(1u << g.nt) << 4) | ((u64)((g.team >> seat) & 1) << 8) |
                 ((u64)(has_king(g.contract) ? 1u : 0u) << 9);
        u64 f2 = (has_king(g.contract) ? (u64)(1u << g.king) : 0) | ((u64)g.trick_no << 4);
        ext[tid][0] = hand_of(g, seat) | ((u64)(1u << g.contract) << 54);
        ext[tid][1] = (live ? legal_now(g) : 0) | (f1 << 54);
        ext[tid][2] = on_table | (f2 << 54);
        ext[tid][3] = (g.C & ~talon_unowned(g) & ~on_table) | ((u64)(live ? 1u : 0u) << 54);
        if (feature_words_out && base + tid < n) {
            feature_words_out[i * 2] = make_ulonglong2(ext[tid][0], ext[tid][1]);
            feature_words_out[i * 2 + 1] = make_ulonglong2(ext[tid][2], ext[tid][3]);
        }
    }
    __syncthreads();
    PM_STAMP(6)
    // expand to bf16 0.0 / 1.0, one 16-byte chunk (8 features = one byte of a feature word) at a time.
    // Two threads per game, thread parity p takes the odd / even bytes of the game's 32: ONE 32-byte
    // LDS read per thread up front instead of a dependent byte read per chunk, and the pair's
    // 16-byte stores fall into different banks.  (A 256-entry LDS table byte -> 8 bf16 was tried:
    // no faster than the selects.)
    {
        u32 gme = tid >> 1, par = tid & 1;
        const uint4 *row = reinterpret_cast<const uint4 *>(&ext[gme][0]);
        uint4 r0 = row[0], r1 = row[1];
        u32 wd[8] = {r0.x, r0.y, r0.z, r0.w, r1.x, r1.y, r1.z, r1.w};
#pragma unroll
        for (int j = 0; j < 16; j++) {
            u32 chunk = 2 * j + par;                       // byte `chunk` of the row = byte (2j + par) & 3 of word j / 2
            u32 byte = (wd[j >> 1] >> (8 * ((2 * (j & 1)) + par))) & 255u;
            *reinterpret_cast<uint4 *>(X + gme * PM_LD + 8 * chunk) = tk_expand_byte(byte);
        }
    }
    if (features_out) {                                    // optional global copy: full 512-byte rows per 32 lanes
        __syncthreads();
#pragma unroll
        for (int it = 0; it < 16; it++) {
            u32 gme = it * (8 * TILES) + (tid >> 5), chunk = tid & 31;
            if (base + gme < n) features_out[(base + gme) * 32 + chunk] = *reinterpret_cast<const uint4 *>(X + gme * PM_LD + 8 * chunk);
        }
    }
    __syncthreads();
    PM_STAMP(1)
    mlp_hidden(Xt, w1, b1, wq, wave4);
    PM_STAMP(2)
    mlp_prefetch(wq, w2, wave4 * 2);
    mlp_hidden(Xt, w2, b2, wq, wave4);
    PM_STAMP(3)
    // ---- layer 3: 64 outputs; wave w takes games [32w, 32w+32), both 32-feature tiles; f32
    // logits to LDS [128][68] over the activation buffer
    float *L = reinterpret_cast<float *>(X);
    {
        u32 lane = __lane_id(), wave = wave4, r = lane & 31, h = lane >> 5;
        const bf16x8 *wf = reinterpret_cast<const bf16x8 *>(w3) + lane;
        f32x16 acc[2];
#pragma unroll
        for (int ft = 0; ft < 2; ft++)
#pragma unroll
            for (int j = 0; j < 16; j++) acc[ft][j] = 0.f;
        bf16x8 wq[4][2];
#pragma unroll
        for (int d = 0; d < 4; d++)
#pragma unroll
            for (int ft = 0; ft < 2; ft++) wq[d][ft] = wf[(ft * 16 + d) * 64];
        bf16x8 xn = *reinterpret_cast<const bf16x8 *>(Xt + (32 * wave + r) * PM_LD + 8 * h);
#pragma unroll
        for (int kk = 0; kk < 16; kk++) {
            bf16x8 x = xn;
            bf16x8 wc[2] = {wq[kk & 3][0], wq[kk & 3][1]};
            if (kk + 4 < 16) {
#pragma unroll
                for (int ft = 0; ft < 2; ft++) wq[kk & 3][ft] = wf[(ft * 16 + kk + 4) * 64];
            }
            if (kk < 15) xn = *reinterpret_cast<const bf16x8 *>(Xt + (32 * wave + r) * PM_LD + 16 * (kk + 1) + 8 * h);
#pragma unroll
            for (int ft = 0; ft < 2; ft++) acc[ft] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wc[ft], x, acc[ft], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
        __syncthreads();                               // all waves are done with X: the logits may overwrite it
#pragma unroll
        for (int ft = 0; ft < 2; ft++)
#pragma unroll
            for (int q = 0; q < 4; q++) {
                u32 f0 = 32 * ft + 8 * q + 4 * h;
                float4 bv = *reinterpret_cast<const float4 *>(b3 + f0);
                float4 o = make_float4(acc[ft][4 * q + 0] + bv.x, acc[ft][4 * q + 1] + bv.y, acc[ft][4 * q + 2] + bv.z, acc[ft][4 * q + 3] + bv.w);
                *reinterpret_cast<float4 *>(L + (PM_M * tile + 32 * wave + r) * PM_LL + f0) = o;
            }
    }
    __syncthreads();
    PM_STAMP(4)
    if (stamps && tid == 0) {
        for (int k = 0; k < 5; k++) stamps[blockIdx.x * 8 + k] = ts[k];
        stamps[blockIdx.x * 8 + 6] = ts[6];
    }
    // ---- masked categorical sample (same draw, same order of additions as k_sample), two lanes
    // per game: lane pair (2g, 2g+1) takes cards 0..26 / 27..53, partner values through DPP
    {
#define PM_SWAP_F(x) __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), 0xB1, 0xF, 0xF, true))   /* quad_perm [1,0,3,2] */
#define PM_SWAP_I(x) __builtin_amdgcn_update_dpp(0, (int)(x), 0xB1, 0xF, 0xF, true)
        u32 gi = tid >> 1, half = tid & 1;
        bool in_range = base + gi < n;
        int64_t i = in_range ? base + gi : n - 1;
        float l[27];
#pragma unroll
        for (int c = 0; c < 27; c++) l[c] = L[gi * PM_LL + 27 * half + c];
        float v54 = L[gi * PM_LL + 54];
        u64 o = obs[i];
        u64 m = o & TAROK_OBS_MASK;
        u32 mh = (u32)(m >> (27 * half)) & 0x7FFFFFFu;         // the legality bits of this lane's cards
        float mx = -3.0e38f;
#pragma unroll
        for (int c = 0; c < 27; c++) mx = ((mh >> c) & 1) ? fmaxf(mx, l[c]) : mx;
        mx = fmaxf(mx, PM_SWAP_F(mx));
#pragma unroll
        for (int c = 0; c < 27; c++) l[c] = ((mh >> c) & 1) ? __expf(l[c] - mx) : 0.f;
        // running sums in card order: the low half from 0, then the high half from the low half's total
        float s = 0.f;
#pragma unroll
        for (int c = 0; c < 27; c++) s += l[c];
        float s_sw = PM_SWAP_F(s);                             // (DPP reads need the source lane active: never inside a select)
        float s_low = half ? s_sw : s;                         // sum over cards 0..26 (in both lanes)
        float start = half ? s_low : 0.f;
        float sum = start;
#pragma unroll
        for (int c = 0; c < 27; c++) sum += l[c];
        float sum_sw = PM_SWAP_F(sum);
        sum = half ? sum : sum_sw;                             // total over cards 0..53, from the high lane
        u32 rr = rng32(gkey[i], 192u + ((u32)(o >> TAROK_OBS_STEP_SHIFT) & 63u));
        float u = ((float)(rr >> 8) + 0.5f) * (1.0f / 16777216.0f) * sum;
        float acc = start, pe = 0.f;
        int pickc = -1;
#pragma unroll
        for (int c = 0; c < 27; c++) {
            bool legal = (mh >> c) & 1;
            acc += l[c];
            bool take = legal && pickc < 0 && acc > u;
            pe = take ? l[c] : pe;
            pickc = take ? c : pickc;
        }
        // rounding at the top end: the last legal card (its lane supplies the probability)
        int last = m ? 63 - __clzll(m) : 0;
        float pl = 0.f;
#pragma unroll
        for (int c = 0; c < 27; c++) pl = (c + 27 * (int)half == last) ? l[c] : pl;
        int pick_o = PM_SWAP_I(pickc);
        float pe_o = PM_SWAP_F(pe), pl_o = PM_SWAP_F(pl);
        if (half == 0 && in_range) {
            int pk = pickc >= 0 ? pickc : (pick_o >= 0 ? pick_o + 27 : last);
            float pp = pickc >= 0 ? pe : (pick_o >= 0 ? pe_o : (last < 27 ? pl : pl_o));
            if (value) value[i] = v54;
            if (!m) { action[i] = 255; if (logp) logp[i] = 0.f; }
            else {
                action[i] = (uint8_t)pk;
                if (logp) logp[i] = __logf(pp / sum);
            }
            if (act_s) act_s[gi] = m ? (uint8_t)pk : (uint8_t)255;
        }
#undef PM_SWAP_F
#undef PM_SWAP_I
        if (stamps && tid == 0) stamps[blockIdx.x * 8 + 5] = __builtin_amdgcn_s_memtime();
    }
}

__global__ __launch_bounds__(TK_BLOCK, 2) TK_VGPR_BUDGET(256) void k_policy_mlp(
    int64_t n, const ulonglong2 *__restrict__ s01, const ulonglong2 *__restrict__ s23, const u64 *__restrict__ obs,
    const u64 *__restrict__ gkey, const __bf16 *__restrict__ w1, const float *__restrict__ b1,
    const __bf16 *__restrict__ w2, const float *__restrict__ b2, const __bf16 *__restrict__ w3,
    const float *__restrict__ b3, uint8_t *__restrict__ action, float *__restrict__ logp, float *__restrict__ value,
    uint4 *__restrict__ features_out, ulonglong2 *__restrict__ feature_words_out, u64 *__restrict__ stamps) {
    TK_VGPR_TOP(256, 255);
    policy_body<1>(n, s01, s23, obs, gkey, w1, b1, w2, b2, w3, b3, action, logp, value, features_out, feature_words_out, stamps,
                   nullptr);
}

// tarok_policy_step: tarok_policy_mlp and tarok_step in ONE launch.  A play workgroup (512
// threads, 256 games = the 256 slots of step group blockIdx.x) evaluates the policy for its games
// as two 128-game tiles side by side, leaves the sampled cards in LDS and then runs the step
// kernel's play role on them (threads 0..255; same refill lists, same launch-parity protocol as
// k_play); the workgroups after the play groups run the refill role.
__global__ __launch_bounds__(2 * TK_BLOCK, 2) TK_VGPR_BUDGET(256) void k_policy_step(
    int64_t n, u64 seed, u64 offset, int mix, int flags, u32 play_groups, u32 *epoch, u32 fan,
    const u64 *__restrict__ obs_in, const __bf16 *__restrict__ w1, const float *__restrict__ b1,
    const __bf16 *__restrict__ w2, const float *__restrict__ b2, const __bf16 *__restrict__ w3,
    const float *__restrict__ b3, uint8_t *__restrict__ action, float *__restrict__ logp, float *__restrict__ value,
    ulonglong2 *__restrict__ feature_words_out, int16_t *__restrict__ reward, uint8_t *__restrict__ done,
    uint16_t *__restrict__ trick, u64 *__restrict__ obs_out, uint8_t *__restrict__ hist,
    ulonglong2 *__restrict__ s01, ulonglong2 *__restrict__ s23, Aux *aux, Counters *__restrict__ cnt,
    u64 *__restrict__ gkey, u64 *rlist, u32 *rcount) {
    TK_VGPR_TOP(256, 255);
    TkCount count = launch_count<1>(epoch, play_groups);          // (in flight under the policy's first loads)
    if (blockIdx.x >= play_groups) {
        refill_role<false>(blockIdx.x - play_groups, threadIdx.x, 2 * TK_BLOCK, seed, offset, mix, play_groups, count, epoch, fan, false, aux, rlist, rcount, nullptr);
        return;
    }
    __shared__ uint8_t act_s[2 * PM_M];
    u32 *lds = nullptr;
    policy_body<2>(n, s01, s23, obs_in, gkey, w1, b1, w2, b2, w3, b3, action, logp, value, nullptr, feature_words_out, nullptr,
                   act_s, &lds);
    __syncthreads();                          // (the policy's LDS is free from here on: the step's scoring list goes there)
    u32 tid = threadIdx.x;
    step_role<false, false>(blockIdx.x, tid & (TK_BLOCK - 1), tid < TK_BLOCK, act_s[tid & (TK_BLOCK - 1)], false, n, seed, offset, mix, flags,
                           count, epoch, play_groups, fan, nullptr, nullptr, reward, done, trick, obs_out, hist, s01, s23, aux, cnt, gkey, rlist, rcount, nullptr,
                           reinterpret_cast<u32 (*)[TK_BLOCK]>(lds));
}

#endif  // TK_BLOCK == 256

// ---------------------------------------------------------------------------
// The reference agent's transition targets in its own form: what Nevronski_igralec.rezultat_stiha
// builds per trick and rezultat_igre completes (Igralec.py:387-446), for every (trick, game, seat) of a recorded
// rollout of whole tricks:  dy[54] = -70 on the cards that were not legal at the seat's decision (:392-393), on the
// card it played +Roka.vrednost_stiha(trick) if it took the trick, else minus that (:412-416: every contract runs
// this branch, the Klop / Berac branches of :394-409 compare a dict with a string), plus final_reword_factor times
// next_max (:441), where next_max is the agent's own estimate at its next decision (:351,417-418; `next_q`, the
// caller's: NULL = 0) or, on the game's last trick, the final reward (:439, the rewards of TAROK_REWARD_REF).
// Rows [T,N] as the step kernels wrote them; T a multiple of 4 and the rollout starts on a trick boundary (games are
// whole tricks long, so every slot stays trick-aligned through the auto-resets): trick b = rows 4b .. 4b+3.
// Parity unpinned for this assembly (the reference holds no fixture and Igralec.py cannot be imported); its inputs —
// legal masks, cards, trick values and winners, final scores — are pinned by the reference's recorded runs.
// Workgroup = 64 slots x 4 seats: one thread per (slot, seat) builds its row's description, then the 256 threads
// write the 64 x 4 x 54 floats of the workgroup as full lines.
#define TG_SLOTS 64
TK_KERNEL(256, 64) void k_targets_ref(int64_t n, int T, const u64 *__restrict__ obs_before, const uint8_t *__restrict__ action,
                                                    const uint16_t *__restrict__ trick, const uint8_t *__restrict__ done,
                                                    const int16_t *__restrict__ reward, const float *__restrict__ next_q, float factor,
                                                    float4 *__restrict__ dy, uint8_t *__restrict__ meta) {
    TK_VGPR_TOP(64, 63);
    __shared__ uint2 mask_s[256];
    __shared__ float val_s[256];
    __shared__ u32 card_s[256];
    const u32 tid = threadIdx.x, b = blockIdx.y;
    const int64_t i0 = (int64_t)blockIdx.x * TG_SLOTS, i = i0 + (tid >> 2);
    const u32 s = tid & 3;
    u64 legal = TK_DECK;                                      // (rows without a decision: all zeros)
    u32 card = 255, flags = 0;
    float val = 0.f;
    if (i < n) {
        int found = -1;
        for (int k = 0; k < 4; k++) {
            u64 ob = obs_before[(int64_t)(4 * b + k) * n + i];
            bool mine = ((u32)(ob >> TAROK_OBS_SEAT_SHIFT) & 3u) == s && action[(int64_t)(4 * b + k) * n + i] < 54 && found < 0;
            if (mine) { found = k; legal = ob & TAROK_OBS_MASK; card = action[(int64_t)(4 * b + k) * n + i]; }
        }
        int64_t last = (int64_t)(4 * b + 3) * n + i;
        u32 tw = trick[last];
        if (found >= 0 && (tw & 0x8000u)) {
            flags = 1;
            float tv = (float)((tw >> 4) & 0x7FFu);
            val = ((tw & 3u) == s) ? tv : -tv;                // sem_pobral (Klop.py:76-77, Navadna_igra.py:138-139)
            float next_max = 0.f;
            if (done[last]) { next_max = (float)reward[last * 4 + s]; flags |= 2; }
            else if ((int)(4 * b + 7) < T) {
                int nk = -1;
                for (int k = 0; k < 4; k++) {
                    u64 ob = obs_before[(int64_t)(4 * b + 4 + k) * n + i];
                    if (((u32)(ob >> TAROK_OBS_SEAT_SHIFT) & 3u) == s && nk < 0) nk = k;
                }
                if (nk >= 0) next_max = next_q ? next_q[(int64_t)(4 * b + 4 + nk) * n + i] : 0.f;
                else flags |= 4;
            } else flags |= 4;                                // the seat's next decision lies beyond the rollout
            val += factor * next_max;
        } else { legal = TK_DECK; card = 255; }
        meta[((int64_t)b * n + i) * 4 + s] = (uint8_t)flags;
    }
    mask_s[tid] = make_uint2(TK_LO(legal), TK_HI(legal));
    val_s[tid] = val;
    card_s[tid] = card;
    __syncthreads();
    int64_t slots = n - i0 < TG_SLOTS ? n - i0 : TG_SLOTS;
    u32 quads = (u32)(slots * 4 * 54 / 4);                    // float4 pieces of this workgroup's rows
    float4 *out = dy + ((int64_t)b * n + i0) * 54;            // (row (b, i, seat) starts at float ((b n + i) 4 + seat) 54)
    for (u32 e4 = tid; e4 < quads; e4 += 256) {
        float v[4];
#pragma unroll
        for (u32 k = 0; k < 4; k++) {
            u32 e = 4 * e4 + k, row = e / 54, c = e - 54 * row;
            uint2 m = mask_s[row];
            u32 bit = c < 32 ? (m.x >> c) & 1u : (m.y >> (c - 32)) & 1u;
            v[k] = c == card_s[row] ? val_s[row] : (bit ? 0.f : -70.f);
        }
        out[e4] = make_float4(v[0], v[1], v[2], v[3]);
    }
}

#include "tarok_learner.inc"

TK_KERNEL(TK_BLOCK, 64) void k_counters(int64_t n, const Counters *__restrict__ cnt, u32 *__restrict__ ep,
                                                      int4 *__restrict__ score_sum) {
    TK_VGPR_TOP(64, 63);
    int64_t i = (int64_t)blockIdx.x * TK_BLOCK + threadIdx.x;
    if (i >= n) return;
    if (ep) ep[i] = cnt[i].episode;
    if (score_sum) score_sum[i] = cnt[i].score_sum;
}

// canonical lanes for parity checks: H0-3, P0-3, TAL, META (tarok_env.h)
TK_KERNEL(TK_BLOCK, 64) void k_get_state(int64_t n, const ulonglong2 *__restrict__ s01,
                                                       const ulonglong2 *__restrict__ s23, u64 *__restrict__ out) {
    TK_VGPR_TOP(64, 63);
    int64_t i = (int64_t)blockIdx.x * TK_BLOCK + threadIdx.x;
    if (i >= n) return;
    Game g;
    load_game(g, s01, s23, i);
    u64 on_table = 0;
    for (u32 j = 0; j < g.nt; j++) on_table |= 1ULL << ((g.trick >> (6 * j)) & 63);
    u64 won = g.C & ~talon_unowned(g) & ~on_table;
    for (u32 s = 0; s < 4; s++) {
        out[(int64_t)s * n + i] = hand_of(g, s);
        out[(int64_t)(4 + s) * n + i] = seat_cards(g, s) & won;
    }
    out[8 * n + i] = g.talon;
    u64 m = g.trick;
    m |= (u64)g.nt << 24;
    m |= (u64)g.leader << 27;
    m |= (u64)g.trick_no << 29;
    m |= (u64)g.contract << 33;
    m |= (u64)g.declarer << 37;
    m |= (u64)(has_king(g.contract) ? g.king : 7) << 39;
    m |= (u64)g.team << 42;
    m |= (u64)(g.contract == TK_KLOP ? g.tl : 0) << 46;
    m |= (u64)(has_exchange(g.contract) ? g.tl : 7) << 49;
    m |= (u64)g.phase << 52;
    m |= (u64)g.error << 54;
    out[9 * n + i] = m;
}

// inverse of k_get_state: rebuild the packed pairs from canonical lanes (checkpoint restore,
// hand-built positions).  Cards on the table go back to whoever played them, the un-owned
// talon to where setup_game parks it.
TK_KERNEL(TK_BLOCK, 64) void k_set_state(int64_t n, const u64 *__restrict__ in,
                                                       ulonglong2 *__restrict__ s01, ulonglong2 *__restrict__ s23,
                                                       const Counters *__restrict__ cnt) {
    TK_VGPR_TOP(64, 63);
    int64_t i = (int64_t)blockIdx.x * TK_BLOCK + threadIdx.x;
    if (i >= n) return;
    u64 m = in[9 * n + i];
    Game g;
    g.trick = (u32)m & 0xFFFFFF;
    g.nt = (u32)(m >> 24) & 7; g.leader = (u32)(m >> 27) & 3; g.trick_no = (u32)(m >> 29) & 15;
    g.contract = (u32)(m >> 33) & 15; g.declarer = (u32)(m >> 37) & 3;
    u32 king = (u32)(m >> 39) & 7;
    g.king = king == 7 ? 0 : king;
    g.team = (u32)(m >> 42) & 15;
    u32 talon_left = (u32)(m >> 46) & 7, choice = (u32)(m >> 49) & 7;
    g.phase = (u32)(m >> 52) & 3; g.error = (u32)(m >> 54) & 1;
    g.epar = TK_LINE(cnt[i].episode);
    {   // lines on a refill list stay off limits for the next launch
        ulonglong2 old = s23[i];
        g.cprev = ((u32)(old.y >> 62) & 3u) | (((u32)(old.x >> 62) & 3u) << 2);
    }
    g.talon = in[8 * n + i] & ((1ULL << 36) - 1);
    g.tl = g.contract == TK_KLOP ? talon_left : ((has_exchange(g.contract) || g.contract == TK_SOLO_BREZ) ? choice : 0);
    u64 seatc[4];
    u64 piles = 0;
    for (u32 s = 0; s < 4; s++) {
        u64 h = in[(int64_t)s * n + i] & TK_DECK, p = in[(int64_t)(4 + s) * n + i] & TK_DECK;
        seatc[s] = h | p;
        piles |= p;
    }
    u64 on_table = 0;
    for (u32 j = 0; j < g.nt && j < 4; j++) {
        u64 b = 1ULL << ((g.trick >> (6 * j)) & 63);
        on_table |= b;
        seatc[(g.leader + j) & 3] |= b;
    }
    u64 unowned = talon_unowned(g);                 // depends only on contract / tl / talon ids
    u32 o = g.contract == TK_KLOP ? 0u : (u32)__builtin_ctz(~g.team & 15u);
    seatc[o & 3] |= unowned;
    g.A = seatc[1] | seatc[3];
    g.B = seatc[2] | seatc[3];
    g.C = piles | on_table | unowned;
    store_game(g, s01, s23, i);
}


// ---------------------------------------------------------------------------
// tarok_debug_refill_selftest: the refill role AS EACH STEP KERNEL COMPILES IT, fed with lists built by hand — every slot of
// every group, `per_slot` episodes each, on both parities — and every line it writes compared with a straight re-deal by a
// kernel of its own.  Round 3's corruption was a dealt-ahead game WRITTEN wrong by the refill loop of one build (a gfx950
// erratum, see TK_VGPR_TOP); no test looked at the lines themselves, and only lists of several hundred entries — four
// passes of the loop on full waves — showed it.  This is that test (tests/test_gpu_parity.py), for any build.
__global__ __launch_bounds__(TK_BLOCK) TK_VGPR_BUDGET(64) void k_dbg_fill_lists(u64 *rlist, u32 *rcount, u32 per_slot, u32 ep0, u32 order) {
    TK_VGPR_TOP(64, 63);
    u32 g = blockIdx.x;
    for (u32 idx = threadIdx.x; idx < per_slot * TK_BLOCK; idx += TK_BLOCK) {
        u32 k = idx / TK_BLOCK, t = idx % TK_BLOCK;
        if (order == 1) t = TK_BLOCK - 1 - t;
        if (order == 2) t = (t * 37u + 11u) % TK_BLOCK;
        u64 en = ((u64)(ep0 + 1 + k) << 32) | t;
        rlist[((int64_t)g * 2 + 0) * TK_REFILL_CAP + idx] = en;
        rlist[((int64_t)g * 2 + 1) * TK_REFILL_CAP + idx] = en;
    }
    if (threadIdx.x == 0) {
        rcount[TK_RC(g, 0)] = per_slot * TK_BLOCK; rcount[TK_RC(g, 1)] = per_slot * TK_BLOCK;
        rcount[TK_RC(g, 2)] = 0; rcount[TK_RC(g, 3)] = 0;
    }
}
__global__ __launch_bounds__(TK_BLOCK) TK_VGPR_BUDGET(64) void k_dbg_clear_lines(Aux *aux, int64_t n) {
    TK_VGPR_TOP(64, 63);
    int64_t i = (int64_t)blockIdx.x * TK_BLOCK + threadIdx.x;
    if (i >= n) return;
    for (int b = 0; b < TK_AHEAD; b++) {
        AuxLine *ln = &aux[i].line[b];
        ln->n01 = make_ulonglong2(0, 0); ln->n23 = make_ulonglong2(0, 0); ln->nkey = 0; ln->nep = 0xFFFFFFFFu;
    }
}
// out[0] = lines that differ; out[1 + 6 r ..] = {slot, episode, x0 read, x0 expected, y0 read, y0 expected} of the first eight
__global__ __launch_bounds__(TK_BLOCK) TK_VGPR_BUDGET(128) void k_dbg_check_lines(Aux *aux, int64_t n, u64 seed, u64 offset, int mix, u32 per_slot,
                                                                                  u32 ep0, unsigned long long *out) {
    TK_VGPR_TOP(128, 127);
    int64_t i = (int64_t)blockIdx.x * TK_BLOCK + threadIdx.x;
    if (i >= n) return;
    for (u32 k = 0; k < per_slot; k++) {
        u32 episode = ep0 + 1 + k;
        u64 key = game_key(seed, offset + (u64)i, (u64)episode);
        u64 h0, h1, h2, h3, tal;
        deal_thread(key, h0, h1, h2, h3, tal);
        u32 c, d, kk;
        sample_setup(key, mix, c, d, kk);
        Game g;
        setup_game(g, h0, h1, h2, h3, tal, c, d, kk);
        g.epar = TK_LINE(episode); g.cprev = 0;
        if (g.phase == TK_PHASE_EXCHANGE) bot_exchange(g, key);
        ulonglong2 ea, eb;
        pack(g, ea.x, ea.y, eb.x, eb.y);
        const AuxLine *ln = &aux[i].line[TK_LINE(episode)];
        ulonglong2 na = ln->n01, nb = ln->n23;
        if (ea.x != na.x || ea.y != na.y || eb.x != nb.x || eb.y != nb.y || key != ln->nkey || ln->nep != episode) {
            unsigned long long r = atomicAdd(&out[0], 1ULL);
            if (r < 8) {
                unsigned long long *o = out + 1 + 6 * r;
                o[0] = (u64)i; o[1] = episode; o[2] = na.x; o[3] = ea.x; o[4] = nb.x; o[5] = eb.x;
            }
        }
    }
}

// ---------------------------------------------------------------------------
// C ABI
// ---------------------------------------------------------------------------
extern "C" {

const char *tarok_strerror(int code) {
    switch (code) {
        case TAROK_OK: return "ok";
        case TAROK_EINVAL: return "invalid argument";
        case TAROK_EHIP: return "HIP runtime error (see tarok_last_hip_error)";
        case TAROK_ENOMEM: return "out of device memory";
        case TAROK_ENODEV: return "no usable GPU";
        default: return "unknown error";
    }
}

int tarok_abi_version(void) { return TAROK_ABI_VERSION; }
int tarok_last_hip_error(void) { return g_last_hip; }

int tarok_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) { (void)hipGetLastError(); return 0; }
    return n;
}

int tarok_create(tarok_env **out, int device, int64_t n_games, uint64_t game_offset, uint64_t seed, int mix, int flags) {
    if (!out || n_games <= 0 || n_games > (1LL << 31) || (flags & ~TAROK_HISTORY)) return TAROK_EINVAL;
    if (!(mix == TAROK_MIX_ALL || mix == TAROK_MIX_NAVADNA3 || mix == TAROK_MIX_BOT || (mix >= TAROK_MIX_FIXED && mix < TAROK_MIX_FIXED + 10))) return TAROK_EINVAL;
    if (device < 0 || device >= tarok_device_count()) return TAROK_ENODEV;
    HIPCHK(hipSetDevice(device));
    tarok_env *e = new tarok_env();
    memset(e, 0, sizeof *e);
    e->device = device; e->n = n_games; e->offset = game_offset; e->seed = seed; e->mix = mix; e->flags = flags;
    // latency-bound batches keep refill workgroups single-pass; throughput-bound ones pack them dense
    e->refill_fan = n_games >= (1 << 20) ? TK_REFILL_FAN : (n_games >= (1 << 18) ? 4 : 1);
    // Bulk deals where a launch is a handful of workgroups per CU and its refill workgroups' deals are its tail (65,536
    // games: -8 % per lock-step, 262,144: -7 %); where the batch streams the deals' instructions add to the launch
    // wherever they run, and a whole stretch of launches' worth at once cost what they cost one trick at a time (profiles/r03_ab_step.txt (e))
    e->lazy_refill = n_games < (1 << 20) ? 1 : 0;
    size_t stale_bytes = (size_t)((n_games + TK_PF_SLOTS - 1) / TK_PF_SLOTS) * TK_PF_SLOTS * sizeof(uint16_t);
    hipError_t r = hipMalloc((void **)&e->s01, (size_t)n_games * sizeof(ulonglong2));
    if (r == hipSuccess) r = hipMalloc((void **)&e->s23, (size_t)n_games * sizeof(ulonglong2));
    if (r == hipSuccess) r = hipMalloc((void **)&e->aux, (size_t)n_games * sizeof(Aux));
    if (r == hipSuccess) r = hipMalloc((void **)&e->cnt, (size_t)n_games * sizeof(Counters));
    if (r == hipSuccess) r = hipMalloc((void **)&e->nstale, stale_bytes);
    if (r == hipSuccess) r = hipMalloc((void **)&e->gkey, (size_t)n_games * sizeof(u64));
    if (r == hipSuccess && (flags & TAROK_HISTORY)) r = hipMalloc((void **)&e->hist, (size_t)n_games * 48);
    if (r == hipSuccess && e->hist) r = hipMemset(e->hist, 255, (size_t)n_games * 48);
    size_t groups = (size_t)((n_games + TK_BLOCK - 1) / TK_BLOCK);
    if (r == hipSuccess) r = hipMalloc((void **)&e->rlist, groups * 2 * TK_REFILL_CAP * sizeof(u64));
    if (r == hipSuccess) r = hipMalloc((void **)&e->rcount, TK_RC(groups, 0) * sizeof(u32));
    if (r == hipSuccess) r = hipMalloc((void **)&e->elist, groups * 2 * TK_BULK_CAP * sizeof(u64));
    if (r == hipSuccess) r = hipMalloc((void **)&e->adam_sumsq, 1024 * sizeof(float));
    if (r == hipSuccess) r = hipMalloc((void **)&e->epoch, 2 * TK_EPOCH_WORDS * sizeof(u32));
    if (r == hipSuccess) r = hipMemset(e->epoch, 0, 2 * TK_EPOCH_WORDS * sizeof(u32));
    if (r == hipSuccess) r = hipMemset(e->rcount, 0, TK_RC(groups, 0) * sizeof(u32));
    if (r == hipSuccess) r = hipMemset(e->s01, 0, (size_t)n_games * sizeof(ulonglong2));
    if (r == hipSuccess) r = hipMemset(e->s23, 0, (size_t)n_games * sizeof(ulonglong2));
    if (r == hipSuccess) r = hipMemset(e->aux, 0, (size_t)n_games * sizeof(Aux));
    if (r == hipSuccess) r = hipMemset(e->cnt, 0, (size_t)n_games * sizeof(Counters));
    if (r == hipSuccess) r = hipMemset(e->nstale, 0, stale_bytes);
    if (r == hipSuccess) r = hipMemset(e->gkey, 0, (size_t)n_games * sizeof(u64));
    if (r == hipSuccess) r = hipStreamCreateWithFlags(&e->cap_stream, hipStreamNonBlocking);
    if (r != hipSuccess) {
        g_last_hip = (int)r;
        tarok_destroy(e);
        return r == hipErrorOutOfMemory ? TAROK_ENOMEM : TAROK_EHIP;
    }
    *out = e;
    return TAROK_OK;
}

static void drop_graphs(tarok_env *e) {      // (behind a device synchronisation: nothing of theirs is queued any more)
    (void)hipDeviceSynchronize();
    for (int k = 0; k < e->n_graphs; k++) (void)hipGraphExecDestroy(e->graphs[k].exec);
    e->n_graphs = 0;
}

void tarok_destroy(tarok_env *e) {
    if (!e) return;
    (void)hipSetDevice(e->device);
    drop_graphs(e);
    if (e->cap_stream) (void)hipStreamDestroy(e->cap_stream);
    (void)hipFree(e->s01); (void)hipFree(e->s23); (void)hipFree(e->aux); (void)hipFree(e->cnt); (void)hipFree(e->nstale); (void)hipFree(e->gkey);
    (void)hipFree(e->rlist); (void)hipFree(e->rcount); (void)hipFree(e->elist); (void)hipFree(e->epoch); (void)hipFree(e->hist); (void)hipFree(e->adam_sumsq);
    delete e;
}

int64_t tarok_num_games(const tarok_env *e) { return e ? e->n : 0; }

int tarok_set_option(tarok_env *e, int option, int value) {
    if (!e) return TAROK_EINVAL;
    if (option == TAROK_OPT_REFILL_FAN && value >= 1 && value <= TK_REFILL_FAN) {
        if ((uint32_t)value != e->refill_fan && e->launched) {
            // The fan sets the step launches' grid, and the device-side launch counters count workgroups of ONE grid size
            // (launch_phase): behind launches of the old grid they would put workgroups of one launch into different
            // phases.  So the change waits for everything queued, restarts the counters and empties all four refill lists
            // of every group — their entries are only ever a hint: a line whose deal is dropped stays stale (its tag says
            // so) until its slot reaches it, deals that game in place and lists all its lines again.
            HIPCHK(hipSetDevice(e->device));
            HIPCHK(hipDeviceSynchronize());
            HIPCHK(hipMemset(e->epoch, 0, 2 * TK_EPOCH_WORDS * sizeof(u32)));
            HIPCHK(hipMemset(e->rcount, 0, TK_RC((e->n + TK_BLOCK - 1) / TK_BLOCK, 0) * sizeof(u32)));
        }
        e->refill_fan = (uint32_t)value;
    } else if (option == TAROK_OPT_LAZY_REFILL && (value == 0 || value == 1)) {
        // (safe between any two launches: with the option off every step launch empties the stretch lists unworked, so
        // there is nothing on them when it comes back on, and entries dropped when it goes off only leave stale lines)
        e->lazy_refill = (uint32_t)value;
    } else return TAROK_EINVAL;
    return TAROK_OK;      // (graphs instantiated for the old tuning stay cached under their own key: tarok_run_random)
}

static inline void launch_prefetch(tarok_env *e, hipStream_t s) {
    dim3 grid((unsigned)((e->n + TK_PF_SLOTS - 1) / TK_PF_SLOTS));
    hipLaunchKernelGGL(k_prefetch, grid, dim3(TK_BLOCK), 0, s, e->n, e->seed, e->offset, e->mix, e->aux, e->cnt, e->nstale);
}

int tarok_reset(tarok_env *e, uint32_t episode, const uint8_t *deals, const int8_t *contract, const int8_t *declarer,
                const int8_t *king_suit, const int8_t *talon_choice, const uint8_t *discards, int flags, void *stream) {
    if (!e) return TAROK_EINVAL;
    if ((talon_choice == nullptr) != (discards == nullptr)) return TAROK_EINVAL;
    if (contract && !declarer) return TAROK_EINVAL;
    HIPCHK(hipSetDevice(e->device));
    HIPCHK(hipMemsetAsync(e->rcount, 0, TK_RC((e->n + TK_BLOCK - 1) / TK_BLOCK, 0) * sizeof(u32), (hipStream_t)stream));
    hipLaunchKernelGGL(k_reset, grid_for(e->n), dim3(TK_BLOCK), 0, (hipStream_t)stream, e->n, e->seed, e->offset,
                       episode, e->mix, flags, deals, contract, declarer, king_suit, talon_choice, discards, e->s01,
                       e->s23, e->aux, e->cnt, e->nstale, e->gkey);
    launch_prefetch(e, (hipStream_t)stream);
    HIPCHK(hipGetLastError());
    return TAROK_OK;
}

int tarok_prefetch(tarok_env *e, void *stream) {
    if (!e) return TAROK_EINVAL;
    HIPCHK(hipSetDevice(e->device));
    launch_prefetch(e, (hipStream_t)stream);
    HIPCHK(hipGetLastError());
    return TAROK_OK;
}

int tarok_exchange(tarok_env *e, const int8_t *talon_choice, const uint8_t *discards, void *stream) {
    if (!e || (talon_choice == nullptr) != (discards == nullptr)) return TAROK_EINVAL;
    HIPCHK(hipSetDevice(e->device));
    hipLaunchKernelGGL(k_exchange, grid_for(e->n), dim3(TK_BLOCK), 0, (hipStream_t)stream, e->n, talon_choice,
                       discards, e->s01, e->s23, e->gkey);
    HIPCHK(hipGetLastError());
    return TAROK_OK;
}

int tarok_legal_actions(tarok_env *e, uint64_t *obs_out, int8_t *seat_out, void *stream) {
    if (!e || !obs_out) return TAROK_EINVAL;
    HIPCHK(hipSetDevice(e->device));
    hipLaunchKernelGGL(k_legal, grid_for(e->n), dim3(TK_BLOCK), 0, (hipStream_t)stream, e->n, e->s01, e->s23,
                       (u64 *)obs_out, seat_out);
    HIPCHK(hipGetLastError());
    return TAROK_OK;
}

// the diagnostics buffer for a kernel that writes `words` of it, or NULL (none registered, or too small)
static inline u64 *stamps_for(const tarok_env *e, size_t words) { return (e->stamps && e->stamps_words >= words) ? e->stamps : nullptr; }

// One play launch: play workgroups + the refill workgroups that work the previous launch's lists off.
// cards = 1 (tarok_step, tarok_step_random): the one-card kernel k_step; more cards: the Bot-policy card loops
// of k_play_wide.
static inline void launch_play(tarok_env *e, bool random, int cards, int64_t stride, const uint8_t *action_in,
                               uint8_t *action_out, int16_t *reward, uint8_t *done, uint16_t *trick, uint64_t *obs,
                               int flags, hipStream_t s) {
    u32 groups = (u32)((e->n + TK_BLOCK - 1) / TK_BLOCK);
    u32 fan = e->refill_fan;
    dim3 grid(groups + (groups + fan - 1) / fan);
    e->launched = 1;
    if (cards == 1 && e->lazy_refill) grid = dim3(groups);     // kind 0 (launch_count): the step workgroups work their own lists off
    if (cards == 1) {
#define TK_LAUNCH_STEP(R, Z)                                                                                             \
    hipLaunchKernelGGL((k_step<R, Z>), grid, dim3(TK_BLOCK), 0, s, e->n, e->seed, e->offset, e->mix, flags, groups, e->epoch, \
                       fan, action_in, action_out, reward, done, trick, (u64 *)obs, e->hist, e->s01, e->s23, e->aux, e->cnt,  \
                       e->gkey, e->rlist, e->rcount, e->elist)
        if (e->lazy_refill) { if (random) TK_LAUNCH_STEP(true, true); else TK_LAUNCH_STEP(false, true); }
        else                { if (random) TK_LAUNCH_STEP(true, false); else TK_LAUNCH_STEP(false, false); }
#undef TK_LAUNCH_STEP
        return;
    }
#define TK_LAUNCH_PLAY(H)                                                                                              \
    hipLaunchKernelGGL((k_play_wide<H>), grid, dim3(TK_BLOCK), 0, s, e->n, e->seed, e->offset, e->mix, flags, cards, stride, \
                       groups, e->epoch, fan, action_out, reward, done, trick, (u64 *)obs, e->hist, e->s01,            \
                       e->s23, e->aux, e->cnt, e->gkey, e->rlist, e->rcount, stamps)
    u64 *stamps = stamps_for(e, 3 * (size_t)((e->n + 63) / 64));
    if (e->hist) TK_LAUNCH_PLAY(true); else TK_LAUNCH_PLAY(false);
#undef TK_LAUNCH_PLAY
}

int tarok_step(tarok_env *e, const uint8_t *action, int16_t *reward_out, uint8_t *done_out, uint16_t *trick_out,
               uint64_t *obs_out, int flags, void *stream) {
    if (!e || !action || !obs_out) return TAROK_EINVAL;
    HIPCHK(hipSetDevice(e->device));
    launch_play(e, false, 1, e->n, action, nullptr, reward_out, done_out, trick_out, obs_out, flags, (hipStream_t)stream);
    HIPCHK(hipGetLastError());
    return TAROK_OK;
}

// k_policy_x4 where the batch streams (from 2^21 games; at 65,536 games its four cards in a row are 0.4 us of latency more
// per launch than the one-game kernel, at 2^20 the two are equal: profiles/r03_ab_step.txt) and the caller's arrays
// allow its 16-byte loads and 2-byte stores (torch allocations do)
static inline void launch_policy(tarok_env *e, const uint64_t *obs, uint8_t *action, hipStream_t s) {
    if (e->n >= (1 << 21) && (((uintptr_t)obs & 15) | ((uintptr_t)action & 1)) == 0)
        hipLaunchKernelGGL(k_policy_x4, dim3((unsigned)((e->n + 4 * TK_BLOCK - 1) / (4 * TK_BLOCK))), dim3(TK_BLOCK), 0, s, e->n,
                           (const u64 *)obs, e->gkey, action);
    else
        hipLaunchKernelGGL(k_policy, grid_for(e->n), dim3(TK_BLOCK), 0, s, e->n, (const u64 *)obs, e->gkey, action);
}

int tarok_policy_random(tarok_env *e, const uint64_t *obs, uint8_t *action_out, void *stream) {
    if (!e || !obs || !action_out) return TAROK_EINVAL;
    HIPCHK(hipSetDevice(e->device));
    launch_policy(e, obs, action_out, (hipStream_t)stream);
    HIPCHK(hipGetLastError());
    return TAROK_OK;
}

int tarok_step_random(tarok_env *e, uint8_t *action_out, int16_t *reward_out, uint8_t *done_out, uint16_t *trick_out,
                      uint64_t *obs_out, int flags, void *stream) {
    if (!e || !obs_out) return TAROK_EINVAL;
    HIPCHK(hipSetDevice(e->device));
    launch_play(e, true, 1, e->n, nullptr, action_out, reward_out, done_out, trick_out, obs_out, flags, (hipStream_t)stream);
    HIPCHK(hipGetLastError());
    return TAROK_OK;
}

int tarok_krog_random(tarok_env *e, int cards, int64_t stride, uint8_t *action_out, int16_t *reward_out,
                      uint8_t *done_out, uint16_t *trick_out, uint64_t *obs_out, int flags, void *stream) {
    if (!e || !obs_out || cards < 1 || cards > TAROK_MAX_CARDS_PER_LAUNCH || stride < e->n) return TAROK_EINVAL;
    HIPCHK(hipSetDevice(e->device));
    launch_play(e, true, cards, stride, nullptr, action_out, reward_out, done_out, trick_out, obs_out, flags,
                (hipStream_t)stream);
    HIPCHK(hipGetLastError());
    return TAROK_OK;
}

// cards: 0 = tarok_policy_random + tarok_step, 1 = tarok_step_random, >= 2 = tarok_krog_random
static inline void launch_one(tarok_env *e, int cards, uint8_t *action, int16_t *reward, uint8_t *done,
                              uint64_t *obs, int flags, hipStream_t s) {
    if (cards >= 1) {
        launch_play(e, true, cards, e->n, nullptr, cards >= 2 ? action : nullptr, reward, done, nullptr, obs, flags, s);
    } else {
        launch_policy(e, obs, action, s);
        launch_play(e, false, 1, e->n, action, nullptr, reward, done, nullptr, obs, flags, s);
    }
}

int tarok_run_random(tarok_env *e, int64_t n_steps, int cards_per_launch, int graph_chunk, int prefetch_every,
                     uint8_t *action, int16_t *reward_out, uint8_t *done_out, uint64_t *obs_out, int flags, void *stream) {
    if (!e || !obs_out || n_steps < 0 || graph_chunk < 0 || graph_chunk > 8192 || prefetch_every < 0) return TAROK_EINVAL;
    if (cards_per_launch < 0 || cards_per_launch > TAROK_MAX_CARDS_PER_LAUNCH) return TAROK_EINVAL;
    int unit = cards_per_launch >= 2 ? cards_per_launch : 1;          // lock-steps per launch
    if (n_steps % unit != 0 || graph_chunk % unit != 0) return TAROK_EINVAL;
    if (prefetch_every % unit != 0) return TAROK_EINVAL;
    if (graph_chunk > 0 && prefetch_every > 0 && graph_chunk % prefetch_every != 0) return TAROK_EINVAL;
    if (!(flags & TAROK_AUTO_RESET)) prefetch_every = 0;
    if (cards_per_launch == 0 && !action) return TAROK_EINVAL;
    HIPCHK(hipSetDevice(e->device));
    hipStream_t s = (hipStream_t)stream;
    int64_t left = n_steps;
    if (graph_chunk > 0 && left >= graph_chunk) {
        hipGraphExec_t gexec = nullptr;
        for (int k = 0; k < e->n_graphs && !gexec; k++) {
            const tarok_env::GraphEntry &g = e->graphs[k];
            if (g.fused == cards_per_launch && g.chunk == graph_chunk && g.flags == flags && g.prefetch == prefetch_every &&
                g.action == action && g.reward == reward_out && g.done == done_out && g.obs == obs_out && g.stamps == e->stamps &&
                g.fan == e->refill_fan && g.lazy == e->lazy_refill)
                gexec = g.exec;
        }
        if (!gexec) {
            if (e->n_graphs == TK_GRAPH_CACHE) drop_graphs(e);
            hipGraph_t graph = nullptr;
            HIPCHK(hipStreamBeginCapture(e->cap_stream, hipStreamCaptureModeRelaxed));
            for (int k = 0; k < graph_chunk; k += unit) {
                launch_one(e, cards_per_launch, action, reward_out, done_out, obs_out, flags, e->cap_stream);
                if (prefetch_every && (k + unit) % prefetch_every == 0) launch_prefetch(e, e->cap_stream);
            }
            HIPCHK(hipStreamEndCapture(e->cap_stream, &graph));
            hipError_t r = hipGraphInstantiate(&gexec, graph, nullptr, nullptr, 0);
            (void)hipGraphDestroy(graph);
            if (r != hipSuccess) { g_last_hip = (int)r; return TAROK_EHIP; }
            e->graphs[e->n_graphs++] = {gexec, cards_per_launch, graph_chunk, flags, prefetch_every, action, reward_out, done_out,
                                        obs_out, e->stamps, e->refill_fan, e->lazy_refill};
        }
        while (left >= graph_chunk) {
            HIPCHK(hipGraphLaunch(gexec, s));
            left -= graph_chunk;
        }
    }
    for (int64_t k = 0; left > 0; left -= unit, k += unit) {
        launch_one(e, cards_per_launch, action, reward_out, done_out, obs_out, flags, s);
        if (prefetch_every && (k + unit) % prefetch_every == 0) launch_prefetch(e, s);
    }
    HIPCHK(hipGetLastError());
    return TAROK_OK;
}

int tarok_rollout_random(tarok_env *e, uint32_t episode, int16_t *scores_out, int16_t *nsteps_out, int8_t *seats_out,
                         uint64_t *masks_out, uint8_t *actions_out, void *stream) {
    if (!e) return TAROK_EINVAL;
    HIPCHK(hipSetDevice(e->device));
    hipLaunchKernelGGL(k_rollout, grid_for(e->n), dim3(TK_BLOCK), 0, (hipStream_t)stream, e->n, e->seed, e->offset,
                       episode, e->mix, scores_out, nsteps_out, seats_out, (u64 *)masks_out, actions_out);
    HIPCHK(hipGetLastError());
    return TAROK_OK;
}

int tarok_debug_stamps_sized(tarok_env *e, uint64_t *stamps, int64_t n_words) {
    if (!e || n_words < 0 || (stamps && n_words == 0)) return TAROK_EINVAL;
    e->stamps = (u64 *)stamps;
    e->stamps_words = stamps ? (size_t)n_words : 0;
    return TAROK_OK;      // (graphs captured with another stamps pointer stay cached under their own key)
}

int tarok_debug_stamps(tarok_env *e, uint64_t *stamps) {      // the step kernels' size: [ceil(N/64), 3]
    if (!e) return TAROK_EINVAL;
    return tarok_debug_stamps_sized(e, stamps, stamps ? 3 * ((e->n + 63) / 64) : 0);
}

int tarok_debug_refill_selftest(tarok_env *e, int kind, int per_slot, uint32_t episode0, int order, int reps, uint64_t *report_out) {
    if (!e || kind < 0 || kind > 3 || per_slot < 1 || per_slot > TK_AHEAD || order < 0 || order > 2 || reps < 1 || !report_out) return TAROK_EINVAL;
#if TK_BLOCK != 256
    return TAROK_EINVAL;
#else
    HIPCHK(hipSetDevice(e->device));
    const int64_t n = e->n;
    const size_t rows = 4;
    void *buf = nullptr;
    unsigned long long *rep = nullptr;
    HIPCHK(hipMalloc(&buf, rows * (size_t)n * (8 + 8 + 1 + 1) + 64));
    hipError_t r = hipMalloc((void **)&rep, 49 * sizeof(unsigned long long));
    if (r == hipSuccess) r = hipMemset(rep, 0, 49 * sizeof(unsigned long long));
    if (r != hipSuccess) { (void)hipFree(buf); g_last_hip = (int)r; return TAROK_EHIP; }
    uint64_t *obs = (uint64_t *)buf;
    int16_t *reward = (int16_t *)(obs + rows * n);
    uint8_t *action = (uint8_t *)(reward + rows * n * 4), *done = action + rows * n;
    u32 groups = (u32)((n + TK_BLOCK - 1) / TK_BLOCK);
    for (int it = 0; it < reps; it++) {
        hipLaunchKernelGGL(k_dbg_clear_lines, grid_for(n), dim3(TK_BLOCK), 0, 0, e->aux, n);
        hipLaunchKernelGGL(k_dbg_fill_lists, dim3(groups), dim3(TK_BLOCK), 0, 0, e->rlist, e->rcount, (u32)per_slot, episode0, (u32)order);
        if (kind == 0) launch_play(e, true, 1, n, nullptr, action, reward, done, nullptr, obs, TAROK_AUTO_RESET, 0);            // k_step, Bot policy
        else if (kind == 1) {                                                                                                  // k_step, cards given
            hipLaunchKernelGGL(k_legal, grid_for(n), dim3(TK_BLOCK), 0, 0, n, e->s01, e->s23, (u64 *)obs, (int8_t *)nullptr);
            launch_policy(e, obs, action, 0);
            launch_play(e, false, 1, n, action, nullptr, reward, done, nullptr, obs, TAROK_AUTO_RESET, 0);
        } else launch_play(e, true, kind == 2 ? 4 : 2, n, nullptr, action, reward, done, nullptr, obs, TAROK_AUTO_RESET, 0);    // k_play_wide
        hipLaunchKernelGGL(k_dbg_check_lines, grid_for(n), dim3(TK_BLOCK), 0, 0, e->aux, n, e->seed, e->offset, e->mix, (u32)per_slot, episode0, rep);
    }
    r = hipDeviceSynchronize();
    if (r == hipSuccess) r = hipMemcpy(report_out, rep, 49 * sizeof(unsigned long long), hipMemcpyDeviceToHost);
    // the lists and lines of the env are the test's now: empty them all and deal every slot's lines afresh
    if (r == hipSuccess) r = hipMemset(e->rcount, 0, TK_RC(groups, 0) * sizeof(u32));
    if (r == hipSuccess) {
        hipLaunchKernelGGL(k_dbg_clear_lines, grid_for(n), dim3(TK_BLOCK), 0, 0, e->aux, n);
        r = hipDeviceSynchronize();
    }
    (void)hipFree(buf); (void)hipFree(rep);
    if (r != hipSuccess) { g_last_hip = (int)r; return TAROK_EHIP; }
    return TAROK_OK;
#endif
}

int tarok_observe(tarok_env *e, void *features_out, void *stream) {
    if (!e || !features_out) return TAROK_EINVAL;
    HIPCHK(hipSetDevice(e->device));
    hipLaunchKernelGGL(k_observe, grid_for(e->n), dim3(TK_BLOCK), 0, (hipStream_t)stream, e->n, e->s01, e->s23,
                       (uint2 *)features_out);
    HIPCHK(hipGetLastError());
    return TAROK_OK;
}

int tarok_observe_ref(tarok_env *e, uint8_t *record_out, int32_t *meta_out, void *stream) {
    if (!e || !record_out) return TAROK_EINVAL;
    if (!e->hist) return TAROK_EINVAL;                                   // needs a TAROK_HISTORY env
    HIPCHK(hipSetDevice(e->device));
    hipLaunchKernelGGL(k_observe_ref, dim3((unsigned)((e->n + OR_GAMES - 1) / OR_GAMES)), dim3(256), 0, (hipStream_t)stream, e->n,
                       e->s01, e->s23, e->hist, (uint4 *)record_out, (int4 *)meta_out);
    HIPCHK(hipGetLastError());
    return TAROK_OK;
}

int tarok_observe_exchange_ref(tarok_env *e, uint8_t *out, void *stream) {
    if (!e || !out) return TAROK_EINVAL;
    HIPCHK(hipSetDevice(e->device));
    hipLaunchKernelGGL(k_observe_exchange_ref, grid_for(e->n), dim3(TK_BLOCK), 0, (hipStream_t)stream, e->n, e->s01, e->s23,
                       (uint2 *)out);
    HIPCHK(hipGetLastError());
    return TAROK_OK;
}

int tarok_observe_hands_ref(tarok_env *e, uint8_t *out, void *stream) {
    if (!e || !out) return TAROK_EINVAL;
    HIPCHK(hipSetDevice(e->device));
    hipLaunchKernelGGL(k_observe_hands_ref, grid_for(e->n), dim3(TK_BLOCK), 0, (hipStream_t)stream, e->n, e->s01, e->s23,
                       (uint2 *)out);
    HIPCHK(hipGetLastError());
    return TAROK_OK;
}

int tarok_get_history(tarok_env *e, uint8_t *hist_out, void *stream) {
    if (!e || !hist_out || !e->hist) return TAROK_EINVAL;
    HIPCHK(hipSetDevice(e->device));
    HIPCHK(hipMemcpyAsync(hist_out, e->hist, (size_t)e->n * 48, hipMemcpyDeviceToDevice, (hipStream_t)stream));
    return TAROK_OK;
}

int tarok_set_history(tarok_env *e, const uint8_t *hist_in, void *stream) {
    if (!e || !hist_in || !e->hist) return TAROK_EINVAL;
    HIPCHK(hipSetDevice(e->device));
    HIPCHK(hipMemcpyAsync(e->hist, hist_in, (size_t)e->n * 48, hipMemcpyDeviceToDevice, (hipStream_t)stream));
    return TAROK_OK;
}

int tarok_sample_policy(tarok_env *e, const void *logits_bf16, const uint64_t *obs, uint8_t *action_out,
                        float *logp_out, void *stream) {
    if (!e || !logits_bf16 || !obs || !action_out) return TAROK_EINVAL;
    HIPCHK(hipSetDevice(e->device));
    hipLaunchKernelGGL(k_sample, grid_for(e->n), dim3(TK_BLOCK), 0, (hipStream_t)stream, e->n, (const uint4 *)logits_bf16,
                       (const u64 *)obs, e->gkey, action_out, logp_out);
    HIPCHK(hipGetLastError());
    return TAROK_OK;
}

int tarok_policy_mlp(tarok_env *e, const void *w1, const float *b1, const void *w2, const float *b2, const void *w3,
                     const float *b3, const uint64_t *obs, uint8_t *action_out, float *logp_out, float *value_out,
                     void *features_out, uint64_t *feature_words_out, void *stream) {
    if (!e || !w1 || !b1 || !w2 || !b2 || !w3 || !b3 || !obs || !action_out) return TAROK_EINVAL;
#if TK_BLOCK != 256
    return TAROK_EINVAL;
#else
    HIPCHK(hipSetDevice(e->device));
    dim3 grid((unsigned)((e->n + PM_M - 1) / PM_M));
    hipLaunchKernelGGL(k_policy_mlp, grid, dim3(TK_BLOCK), 0, (hipStream_t)stream, e->n, e->s01, e->s23, (const u64 *)obs,
                       e->gkey, (const __bf16 *)w1, b1, (const __bf16 *)w2, b2, (const __bf16 *)w3, b3, action_out, logp_out,
                       value_out, (uint4 *)features_out, (ulonglong2 *)feature_words_out, stamps_for(e, 8 * (size_t)grid.x));
    HIPCHK(hipGetLastError());
    return TAROK_OK;
#endif
}

int tarok_policy_step(tarok_env *e, const void *w1, const float *b1, const void *w2, const float *b2, const void *w3,
                      const float *b3, const uint64_t *obs, uint8_t *action_out, float *logp_out, float *value_out,
                      uint64_t *feature_words_out, int16_t *reward_out, uint8_t *done_out, uint16_t *trick_out,
                      uint64_t *obs_out, int flags, void *stream) {
    if (!e || !w1 || !b1 || !w2 || !b2 || !w3 || !b3 || !obs || !action_out || !obs_out) return TAROK_EINVAL;
#if TK_BLOCK != 256
    return TAROK_EINVAL;
#else
    HIPCHK(hipSetDevice(e->device));
    u32 groups = (u32)((e->n + TK_BLOCK - 1) / TK_BLOCK);
    u32 fan = e->refill_fan;
    dim3 grid(groups + (groups + fan - 1) / fan);
    e->launched = 1;
    hipLaunchKernelGGL(k_policy_step, grid, dim3(2 * TK_BLOCK), 0, (hipStream_t)stream, e->n, e->seed, e->offset, e->mix, flags,
                       groups, e->epoch, fan, (const u64 *)obs, (const __bf16 *)w1, b1, (const __bf16 *)w2, b2, (const __bf16 *)w3, b3,
                       action_out, logp_out, value_out, (ulonglong2 *)feature_words_out, reward_out, done_out, trick_out,
                       (u64 *)obs_out, e->hist, e->s01, e->s23, e->aux, e->cnt, e->gkey, e->rlist, e->rcount);
    HIPCHK(hipGetLastError());
    return TAROK_OK;
#endif
}

int tarok_expand_features(tarok_env *e, int64_t n_samples, const uint64_t *feature_words, const int64_t *index,
                          void *features_out, void *stream) {
    if (!e || n_samples <= 0 || !feature_words || !features_out) return TAROK_EINVAL;
    HIPCHK(hipSetDevice(e->device));
    hipLaunchKernelGGL(k_expand_features, grid_for(n_samples * 32), dim3(TK_BLOCK), 0, (hipStream_t)stream, n_samples,
                       (const u64 *)feature_words, index, (uint4 *)features_out);
    HIPCHK(hipGetLastError());
    return TAROK_OK;
}

int tarok_ppo_loss(tarok_env *e, int64_t n_samples, const void *out_bf16, const uint64_t *obs, const int64_t *action,
                   const float *logp_old, const float *advantage, const float *ret, const float *weight, float clip,
                   float vf_coef, float ent_coef, const float *inv_weight_sum, void *dout_bf16, float *partial_out, void *stream) {
    if (!e || n_samples <= 0 || !out_bf16 || !obs || !action || !logp_old || !advantage || !ret || !weight || !dout_bf16 ||
        !partial_out || !inv_weight_sum)
        return TAROK_EINVAL;
    HIPCHK(hipSetDevice(e->device));
    hipLaunchKernelGGL(k_ppo_loss, grid_for(n_samples), dim3(TK_BLOCK), 0, (hipStream_t)stream, n_samples,
                       (const uint4 *)out_bf16, (const u64 *)obs, action, logp_old, advantage, ret, weight, clip, vf_coef,
                       ent_coef, inv_weight_sum, (uint4 *)dout_bf16, (float4 *)partial_out);
    HIPCHK(hipGetLastError());
    return TAROK_OK;
}

int tarok_targets_ref(tarok_env *e, int T, const uint64_t *obs_before, const uint8_t *action, const uint16_t *trick,
                      const uint8_t *done, const int16_t *reward, const float *next_q, float final_reward_factor, float *dy_out,
                      uint8_t *meta_out, void *stream) {
    if (!e || T < 4 || (T & 3) || !obs_before || !action || !trick || !done || !reward || !dy_out || !meta_out) return TAROK_EINVAL;
    HIPCHK(hipSetDevice(e->device));
    dim3 grid((unsigned)((e->n + TG_SLOTS - 1) / TG_SLOTS), (unsigned)(T / 4));
    hipLaunchKernelGGL(k_targets_ref, grid, dim3(256), 0, (hipStream_t)stream, e->n, T, (const u64 *)obs_before, action, trick, done,
                       reward, next_q, final_reward_factor, (float4 *)dy_out, meta_out);
    HIPCHK(hipGetLastError());
    return TAROK_OK;
}

#if TK_BLOCK == 256
int tarok_learn_returns(tarok_env *e, int T, const uint8_t *done, const int16_t *reward, const uint64_t *obs, const float *logp,
                        const float *value, const uint8_t *action, float reward_scale, float *rec_out, float *stats_out,
                        float *scratch, void *stream) {
    if (!e || T < 1 || !done || !reward || !obs || !logp || !value || !action || !rec_out || !stats_out || !scratch) return TAROK_EINVAL;
    HIPCHK(hipSetDevice(e->device));
    dim3 grid = grid_for(e->n);
    hipLaunchKernelGGL(k_returns, grid, dim3(TK_BLOCK), 0, (hipStream_t)stream, e->n, T, done, reward, (const u64 *)obs, logp, value,
                       action, reward_scale, (float4 *)rec_out, (float4 *)scratch);
    hipLaunchKernelGGL(k_adv_stats, dim3(1), dim3(TK_BLOCK), 0, (hipStream_t)stream, (int)grid.x, (int64_t)T * e->n,
                       (const float4 *)scratch, (float4 *)stats_out);
    HIPCHK(hipGetLastError());
    return TAROK_OK;
}

int tarok_learn_chain(tarok_env *e, int64_t B, const uint64_t *feature_words, const int64_t *index, const float *rec,
                      const float *stats, float clip, float vf_coef, float ent_coef, const void *w1, const float *b1,
                      const void *w2, const float *b2, const void *w3, const float *b3, const void *w3t, const void *w2t,
                      uint64_t *Xw, void *H1, void *H2, void *dOut, void *dH2, void *dH1, float *scratch, float *terms_out,
                      float *running, void *stream) {
    if (!e || B < 1 || !feature_words || !rec || !stats || !w1 || !b1 || !w2 || !b2 || !w3 || !b3 || !w3t || !w2t || !Xw || !H1 || !H2 ||
        !dOut || !dH2 || !dH1 || !scratch || !terms_out)
        return TAROK_EINVAL;
    HIPCHK(hipSetDevice(e->device));
    LearnArgs a;
    a.B = B; a.words = (const u64 *)feature_words; a.index = index; a.rec = (const float4 *)rec; a.stats = (const float4 *)stats;
    a.clip = clip; a.vf_coef = vf_coef; a.ent_coef = ent_coef;
    a.w1 = (const __bf16 *)w1; a.w2 = (const __bf16 *)w2; a.w3 = (const __bf16 *)w3; a.w3t = (const __bf16 *)w3t; a.w2t = (const __bf16 *)w2t;
    a.b1 = b1; a.b2 = b2; a.b3 = b3;
    a.Xw = (ulonglong2 *)Xw;
    a.H1 = (uint4 *)H1; a.H2 = (uint4 *)H2; a.dH2 = (uint4 *)dH2; a.dH1 = (uint4 *)dH1; a.dOut = (uint2 *)dOut;
    a.part = (float4 *)scratch;
    unsigned blocks = (unsigned)((B + LN_M - 1) / LN_M);
    a.stamps = stamps_for(e, 8 * (size_t)blocks);       // (eight stamps per workgroup: a buffer sized for the step kernels gets none)
    hipLaunchKernelGGL(k_learn_chain, dim3(blocks), dim3(LN_CHAIN_THREADS), 0, (hipStream_t)stream, a);
    hipLaunchKernelGGL(k_learn_terms, dim3(1), dim3(TK_BLOCK), 0, (hipStream_t)stream, (int)blocks, (const float4 *)scratch,
                       (float4 *)terms_out, (float4 *)running);
    HIPCHK(hipGetLastError());
    return TAROK_OK;
}

// chunks per layer of k_learn_dw: one workgroup per CU in all.  A tile costs a workgroup about the same in every
// layer (per-tile overheads, not bytes or MFMAs, set its time at this size), a little more where both operands are
// 256 wide (layer 2) or the input is expanded from feature words (layer 1), less where the gradient is 64 wide
// (layer 3): shares 96 : 108 : 52 (same-box A/B of six settings: profiles/r03_learner_steps.txt).
static inline void learn_chunks(tarok_env *e, u32 &c2, u32 &c1, u32 &c3) {
    if (!e->n_cus) {
        hipDeviceProp_t pr;
        e->n_cus = hipGetDeviceProperties(&pr, e->device) == hipSuccess && pr.multiProcessorCount > 0 ? pr.multiProcessorCount : 256;
    }
    u32 total = (u32)e->n_cus < 8 ? 8 : (u32)e->n_cus;
    u32 s2 = 96, s1 = 108;
    if (const char *f = getenv("TAROK_DW_SHARES")) { unsigned a = 0, b = 0; if (sscanf(f, "%u,%u", &a, &b) == 2 && a >= 8 && b >= 8 && a + b <= 248) { s2 = a; s1 = b; } }   // diagnostics (A/B runs)
    c2 = total * s2 / 256; c1 = total * s1 / 256; c3 = total - c2 - c1;
}

int64_t tarok_learn_workspace_bytes(tarok_env *e) {
    if (!e) return 0;
    u32 c2, c1, c3;
    learn_chunks(e, c2, c1, c3);
    return (int64_t)sizeof(float) * ((int64_t)(c2 + c1) * 65792 + (int64_t)c3 * 16448);
}

int tarok_learn_dw(tarok_env *e, int64_t B, const uint64_t *Xw, const void *H1, const void *H2,
                   const void *dOut, const void *dH2, const void *dH1, const float *terms, void *workspace, float *grad_out,
                   void *stream) {
    if (!e || B < 1 || !Xw || !H1 || !H2 || !dOut || !dH2 || !dH1 || !terms || !workspace || !grad_out) return TAROK_EINVAL;
    if (B > TAROK_LEARN_MAX_BATCH) return TAROK_EINVAL;       // (k_learn_dw addresses its arrays through 32-bit buffer offsets: 512 B rows)
    HIPCHK(hipSetDevice(e->device));
    u32 c2, c1, c3;
    learn_chunks(e, c2, c1, c3);
    hipLaunchKernelGGL(k_learn_dw, dim3(c2 + c1 + c3), dim3(TK_BLOCK), 0, (hipStream_t)stream, B, c2, c1, c3, (const u64 *)Xw,
                       (const uint4 *)H1, (const uint4 *)H2, (const uint4 *)dOut, (const uint4 *)dH2, (const uint4 *)dH1,
                       (float *)workspace);
    hipLaunchKernelGGL(k_learn_reduce, dim3((LN_P + TK_BLOCK - 1) / TK_BLOCK), dim3(TK_BLOCK), 0, (hipStream_t)stream, c2, c1, c3,
                       (const float *)workspace, (const float4 *)terms, grad_out);
    HIPCHK(hipGetLastError());
    return TAROK_OK;
}

int tarok_learn_adam(tarok_env *e, float *param, const float *grad, float *m, float *v, int32_t *step, float lr, float beta1,
                     float beta2, float eps, float max_norm, void *w1, void *w2, void *w3, void *w3t, void *w2t, float *gnorm_out,
                     int apply, void *stream) {
    if (!e || !param || (apply && (!grad || !m || !v || !step))) return TAROK_EINVAL;
    HIPCHK(hipSetDevice(e->device));
    AdamArgs a;
    a.param = param; a.grad = const_cast<float *>(grad); a.m = m; a.v = v; a.step = step;
    a.lr = lr; a.beta1 = beta1; a.beta2 = beta2; a.eps = eps; a.max_norm = max_norm;
    a.w1 = (__bf16 *)w1; a.w2 = (__bf16 *)w2; a.w3 = (__bf16 *)w3; a.w3t = (__bf16 *)w3t; a.w2t = (__bf16 *)w2t;
    a.gnorm = gnorm_out; a.apply = apply;
    a.sumsq = e->adam_sumsq;
    if (apply) hipLaunchKernelGGL(k_learn_gnorm, dim3(LN_ADAM_BLOCKS), dim3(LN_ADAM_BLOCK), 0, (hipStream_t)stream, grad, e->adam_sumsq, step);
    hipLaunchKernelGGL(k_learn_adam, dim3(LN_ADAM_BLOCKS), dim3(LN_ADAM_BLOCK), 0, (hipStream_t)stream, a);
    HIPCHK(hipGetLastError());
    return TAROK_OK;
}
#else
int tarok_learn_returns(tarok_env *, int, const uint8_t *, const int16_t *, const uint64_t *, const float *, const float *, const uint8_t *,
                        float, float *, float *, float *, void *) { return TAROK_EINVAL; }
int tarok_learn_chain(tarok_env *, int64_t, const uint64_t *, const int64_t *, const float *, const float *, float, float, float,
                      const void *, const float *, const void *, const float *, const void *, const float *, const void *, const void *,
                      uint64_t *, void *, void *, void *, void *, void *, float *, float *, float *, void *) { return TAROK_EINVAL; }
int64_t tarok_learn_workspace_bytes(tarok_env *) { return 0; }
int tarok_learn_dw(tarok_env *, int64_t, const uint64_t *, const void *, const void *, const void *, const void *,
                   const void *, const float *, void *, float *, void *) { return TAROK_EINVAL; }
int tarok_learn_adam(tarok_env *, float *, const float *, float *, float *, int32_t *, float, float, float, float, float, void *, void *,
                     void *, void *, void *, float *, int, void *) { return TAROK_EINVAL; }
#endif

int tarok_get_state(tarok_env *e, uint64_t *lanes_out, void *stream) {
    if (!e || !lanes_out) return TAROK_EINVAL;
    HIPCHK(hipSetDevice(e->device));
    hipLaunchKernelGGL(k_get_state, grid_for(e->n), dim3(TK_BLOCK), 0, (hipStream_t)stream, e->n, e->s01, e->s23,
                       (u64 *)lanes_out);
    HIPCHK(hipGetLastError());
    return TAROK_OK;
}

int tarok_set_state(tarok_env *e, const uint64_t *lanes_in, void *stream) {
    if (!e || !lanes_in) return TAROK_EINVAL;
    HIPCHK(hipSetDevice(e->device));
    hipLaunchKernelGGL(k_set_state, grid_for(e->n), dim3(TK_BLOCK), 0, (hipStream_t)stream, e->n, (const u64 *)lanes_in,
                       e->s01, e->s23, e->cnt);
    HIPCHK(hipGetLastError());
    return TAROK_OK;
}

int tarok_get_counters(tarok_env *e, uint32_t *episode_out, int32_t *score_sum_out, void *stream) {
    if (!e) return TAROK_EINVAL;
    HIPCHK(hipSetDevice(e->device));
    hipLaunchKernelGGL(k_counters, grid_for(e->n), dim3(TK_BLOCK), 0, (hipStream_t)stream, e->n, e->cnt, episode_out,
                       (int4 *)score_sum_out);
    HIPCHK(hipGetLastError());
    return TAROK_OK;
}

}  // extern "C"
