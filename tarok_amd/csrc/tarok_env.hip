// libtarokenv: HIP kernels + C ABI (include/tarok_env.h).  gfx950 only.
//
// One thread per game, 256-thread workgroups (4 waves).  Per-game state is the
// 4 packed uint64 lanes of tarok_device.h, stored as two 16-byte SoA arrays
// (s01[g] = play pair {X0,X1}, s23[g] = seat pair {Y0,Y1}): every wave-level
// load/store moves 1 KiB contiguous.  The work is integer mask algebra + popcounts bounded by
// HBM bandwidth (no MFMA); the only cross-lane work is the wave-cooperative
// re-deal of the few games per wave that finish in a step.
#include "tarok_device.h"

#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string.h>

#include "../../include/tarok_env.h"

#define TK_BLOCK 256

// Per-slot side record, one 64-byte line: everything a FINISHING game touches
// (its score sums, episode number and the prefetched next game) sits together,
// so the sparse finish path costs one line instead of four.
struct __attribute__((aligned(64))) Aux {
    ulonglong2 n01, n23;   // the slot's NEXT game (play pair, seat pair), dealt ahead by k_prefetch;
                           // phase bits of n01.x == 0: not ready
    int4 score_sum;        // scores summed over finished games, by seat (Tarok.rezultati)
    u64 nkey;              // RNG key of that next game
    u32 episode;           // episode number of the slot's current game
    u32 pad;
};

struct tarok_env {
    int device;
    int64_t n;
    u64 offset, seed;
    int mix, flags;
    ulonglong2 *s01, *s23;   // packed state
    Aux *aux;                // finish-path record per slot
    uint8_t *nstale;         // 1 = next-game buffer empty (what k_prefetch scans); padded to 1024 slots
    u64 *gkey;               // RNG key of the slot's current game
    u64 *stamps;             // diagnostics only: per-wave {realtime start, realtime end, cycles} of the last k_step
    hipStream_t cap_stream;  // capture-only stream for tarok_run_random's graph
    // cached graph
    hipGraphExec_t gexec;
    int g_fused, g_chunk, g_flags, g_prefetch;
    void *g_action, *g_reward, *g_done, *g_obs;
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
__global__ __launch_bounds__(TK_BLOCK) void k_reset(
    int64_t n, u64 seed, u64 offset, u32 episode, int mix, int flags,
    const uint8_t *__restrict__ deals, const int8_t *__restrict__ contract, const int8_t *__restrict__ declarer,
    const int8_t *__restrict__ king, const int8_t *__restrict__ choice, const uint8_t *__restrict__ discards,
    ulonglong2 *__restrict__ s01, ulonglong2 *__restrict__ s23, Aux *__restrict__ aux,
    uint8_t *__restrict__ nstale, u64 *__restrict__ gkey) {
    int64_t i = (int64_t)blockIdx.x * TK_BLOCK + threadIdx.x;
    if (i >= n) return;
    u64 key = game_key(seed, offset + (u64)i, episode);
    u64 h0 = 0, h1 = 0, h2 = 0, h3 = 0, tal = 0;
    bool bad = false;
    aux[i].n01 = make_ulonglong2(0, 0);
    nstale[i] = 1;
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
    aux[i].episode = episode;
    if (flags & TAROK_CLEAR_COUNTERS) aux[i].score_sum = make_int4(0, 0, 0, 0);
}

// Deal the NEXT game (episode+1, synthetic contract, Bot exchange) of every slot
// whose next-game buffer is empty, so that a step that finishes a game only has
// to swap 32 bytes in.  Only a few percent of the slots are empty at a time, so
// each workgroup first compacts the empty slots of its 1024-slot tile into an
// LDS list (4 flags per thread, one LDS atomic per thread that found any) and
// then deals list entry j on thread j: the sorting-network deal runs on dense
// lanes, and waves with nothing to do leave.
#define TK_PF_SLOTS 1024
__global__ __launch_bounds__(TK_BLOCK) void k_prefetch(int64_t n, u64 seed, u64 offset, int mix,
                                                      Aux *__restrict__ aux, uint8_t *__restrict__ nstale) {
    __shared__ unsigned short list[TK_PF_SLOTS];
    __shared__ u32 count;
    int64_t base = (int64_t)blockIdx.x * TK_PF_SLOTS;
    if (threadIdx.x == 0) count = 0;
    __syncthreads();
    u32 f = reinterpret_cast<const u32 *>(nstale + base)[threadIdx.x];   // 4 slots; array is padded
    if (f) {
        u32 c = ((f & 0xFF) != 0) + ((f & 0xFF00) != 0) + ((f & 0xFF0000) != 0) + ((f >> 24) != 0);
        u32 pos = atomicAdd(&count, c);
#pragma unroll
        for (u32 k = 0; k < 4; k++)
            if ((f >> (8 * k)) & 0xFF) list[pos++] = (unsigned short)(threadIdx.x * 4 + k);
    }
    __syncthreads();
    u32 total = count;
    for (u32 j = threadIdx.x; j < total; j += TK_BLOCK) {
        int64_t i = base + list[j];
        if (i >= n) continue;
        u64 key = game_key(seed, offset + (u64)i, (u64)aux[i].episode + 1);
        u64 h0, h1, h2, h3, tal;
        deal_thread(key, h0, h1, h2, h3, tal);
        u32 c, d, k;
        sample_setup(key, mix, c, d, k);
        Game g;
        setup_game(g, h0, h1, h2, h3, tal, c, d, k);
        if (g.phase == TK_PHASE_EXCHANGE) bot_exchange(g, key);
        ulonglong2 a, b;
        pack(g, a.x, a.y, b.x, b.y);
        aux[i].n23 = b;
        aux[i].n01 = a;
        aux[i].nkey = key;
        nstale[i] = 0;
    }
}

__global__ __launch_bounds__(TK_BLOCK) void k_exchange(int64_t n, const int8_t *__restrict__ choice,
                                                      const uint8_t *__restrict__ discards,
                                                      ulonglong2 *__restrict__ s01, ulonglong2 *__restrict__ s23,
                                                      const u64 *__restrict__ gkey) {
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

__global__ __launch_bounds__(TK_BLOCK) void k_legal(int64_t n, const ulonglong2 *__restrict__ s01,
                                                   const ulonglong2 *__restrict__ s23, u64 *__restrict__ obs,
                                                   int8_t *__restrict__ seat) {
    int64_t i = (int64_t)blockIdx.x * TK_BLOCK + threadIdx.x;
    if (i >= n) return;
    Game g;
    load_game(g, s01, s23, i);
    obs[i] = obs_word(g, false);
    if (seat) seat[i] = (int8_t)((g.leader + g.nt) & 3);
}

__global__ __launch_bounds__(TK_BLOCK) void k_policy(int64_t n, const u64 *__restrict__ obs,
                                                    const u64 *__restrict__ gkey, uint8_t *__restrict__ action) {
    int64_t i = (int64_t)blockIdx.x * TK_BLOCK + threadIdx.x;
    if (i >= n) return;
    u64 o = obs[i];
    u64 m = o & TAROK_OBS_MASK;
    u32 a = 255;
    if (m) a = policy_action(gkey[i], (u32)(o >> TAROK_OBS_STEP_SHIFT) & 63u, m);
    action[i] = (uint8_t)a;
}

// One lock-step of every game (one `next(g)` per game, Tarok.py:54).
// RANDOM: the Bot policy is evaluated in the same launch instead of reading
// `action`.  Every lane of a wave stays alive to the end: the auto-reset tail
// is wave-cooperative.
template <bool RANDOM>
__global__ __launch_bounds__(TK_BLOCK) void k_step(
    int64_t n, u64 seed, u64 offset, int mix, int flags,
    const uint8_t *__restrict__ action, uint8_t *__restrict__ action_out,
    int16_t *__restrict__ reward, uint8_t *__restrict__ done, uint16_t *__restrict__ trick, u64 *__restrict__ obs,
    ulonglong2 *__restrict__ s01, ulonglong2 *__restrict__ s23, Aux *__restrict__ aux,
    uint8_t *__restrict__ nstale, u64 *__restrict__ gkey, u64 *__restrict__ stamps) {
    int64_t i = (int64_t)blockIdx.x * TK_BLOCK + threadIdx.x;
    bool valid = i < n;
    u64 t_real0 = 0, t_cyc0 = 0;
    if (stamps) { t_real0 = __builtin_amdgcn_s_memrealtime(); t_cyc0 = __builtin_amdgcn_s_memtime(); }
    int64_t ic = valid ? i : n - 1;
    Game g;
    load_game(g, s01, s23, ic);
    u64 key = 0;
    u32 a = 255;
    if (RANDOM) key = gkey[ic]; else a = action[ic];
    bool play = valid && g.phase == TK_PHASE_PLAY;
    bool autoreset = (flags & TAROK_AUTO_RESET) != 0;
    // A game can only end on the 4th card of a trick, and then only in trick 12
    // or in a Berac.  For those few lanes the loads a finish needs (score sums,
    // episode number, the prefetched next game) are issued NOW, next to the
    // state load, instead of as a second memory round trip after the rules.
    bool may_end = play && g.nt == 3 && (g.trick_no == 11 || g.contract == TK_BERAC || g.contract == TK_ODPRTI_BERAC);
    bool may_renew = autoreset && valid && (may_end || g.phase == TK_PHASE_DONE);
    int4 acc = make_int4(0, 0, 0, 0);
    ulonglong2 na = make_ulonglong2(0, 0), nb = na;
    u32 cur_ep = 0;
    u64 nkey = 0;
    if (may_end) acc = aux[i].score_sum;
    if (may_renew) { na = aux[i].n01; nb = aux[i].n23; nkey = aux[i].nkey; cur_ep = aux[i].episode; }
    if (RANDOM) {
        if (play) a = policy_action(key, g.trick_no * 4 + g.nt, legal_now(g));
        if (action_out && valid) action_out[i] = (uint8_t)a;
    }
    u64 scores = 0;
    u32 trick_info = 0;
    int res = -2;
    if (play) res = RANDOM ? apply_step<true>(g, a, scores, trick_info) : apply_step<false>(g, a, scores, trick_info);
    bool fin = res == 1;
    if (trick && valid) trick[i] = (uint16_t)trick_info;
    if (fin) {
        if (reward) reinterpret_cast<u64 *>(reward)[i] = scores;
        acc.x += (int16_t)(scores & 0xFFFF); acc.y += (int16_t)((scores >> 16) & 0xFFFF);
        acc.z += (int16_t)((scores >> 32) & 0xFFFF); acc.w += (int16_t)(scores >> 48);
        aux[i].score_sum = acc;
    }
    bool renew = false;
    if (autoreset) {
        // every finished game is replaced by the slot's next one: normally a 32-byte
        // swap from the prefetched buffer; if that is empty (tarok_prefetch not called
        // for >= 4 steps) the wave deals it cooperatively right here.
        renew = valid && g.phase == TK_PHASE_DONE;
        if (__ballot(renew)) {
            u32 nep = cur_ep + 1;
            bool swapped = false;
            if (renew) {
                if ((na.x >> 62) != 0) {
                    unpack(g, na.x, na.y, nb.x, nb.y);
                    aux[i].n01.x = 0;
                    nstale[i] = 1;
                    swapped = true;
                }
            }
            bool deal_here = renew && !swapped;
            u64 pend = __ballot(deal_here);
            if (pend) {
                if (deal_here) nkey = game_key(seed, offset + (u64)i, nep);
                u64 h0 = 0, h1 = 0, h2 = 0, h3 = 0, tal = 0;
                u32 lane = __lane_id();
                while (pend) {
                    int l = __builtin_ctzll(pend);
                    pend &= pend - 1;
                    u32 klo = (u32)__builtin_amdgcn_readlane((int)(u32)nkey, l);
                    u32 khi = (u32)__builtin_amdgcn_readlane((int)(u32)(nkey >> 32), l);
                    u64 w0, w1, w2, w3, wt;
                    deal_wave(klo, khi, w0, w1, w2, w3, wt);
                    if (lane == (u32)l) { h0 = w0; h1 = w1; h2 = w2; h3 = w3; tal = wt; }
                }
                if (deal_here) {
                    u32 c, d, k;
                    sample_setup(nkey, mix, c, d, k);
                    setup_game(g, h0, h1, h2, h3, tal, c, d, k);
                    if (g.phase == TK_PHASE_EXCHANGE) bot_exchange(g, nkey);
                }
            }
            if (renew) { aux[i].episode = nep; gkey[i] = nkey; }
        }
    }
    if (valid) {
        // A/B change only when a trick was resolved (n_in_trick wrapped to 0) or a new game came in
        if (res != -2 || renew) store_game(g, s01, s23, i, renew || (res >= 0 && g.nt == 0));
        obs[i] = obs_word(g, fin);
        if (done) done[i] = fin ? 1 : 0;
    }
    if (stamps && (threadIdx.x & 63) == 0) {     // diagnostics: never set in bench/test runs
        u64 w = ((u64)blockIdx.x * TK_BLOCK + threadIdx.x) >> 6;
        stamps[3 * w + 0] = t_real0;
        stamps[3 * w + 1] = __builtin_amdgcn_s_memrealtime();
        stamps[3 * w + 2] = __builtin_amdgcn_s_memtime() - t_cyc0;
    }
}


// `cards` cards of every game in ONE launch with the Bot policy evaluated in-kernel; cards = 4 is
// one whole trick = one pass of the reference's krog generator (Klop.py:47-79,
// Navadna_igra.py:115-141).  The packed state is read once, stays in registers while the cards
// are played, and is written once; everything a consumer of the trajectory needs is still
// written for EVERY card: row c of action/obs/done/trick/reward (rows are `stride` games apart)
// belongs to the c-th card of this launch.  Per card this moves less than the one-card kernel
// (the state traffic is shared by `cards` cards) and costs 1/cards of a launch.
//
// A finished game is replaced at once from the slot's prefetched next-game buffer, and the
// buffer is refilled before the launch ends: the ~11 % of a workgroup's slots that consumed
// theirs are compacted through LDS and dealt on dense lanes (thread j deals list entry j), so
// the sorting-network deal costs one pass of latency per launch instead of a separate kernel,
// and the next launch always finds its buffers full.  (A second finish inside the same launch
// — only possible when cards > 4 — is dealt by the wave cooperatively.)
__global__ __launch_bounds__(TK_BLOCK) void k_krog(
    int64_t n, u64 seed, u64 offset, int mix, int flags, int cards, int64_t stride,
    uint8_t *__restrict__ action_out, int16_t *__restrict__ reward, uint8_t *__restrict__ done,
    uint16_t *__restrict__ trick, u64 *__restrict__ obs,
    ulonglong2 *__restrict__ s01, ulonglong2 *__restrict__ s23, Aux *__restrict__ aux,
    uint8_t *__restrict__ nstale, u64 *__restrict__ gkey, u64 *__restrict__ stamps) {
    u64 t_real0 = 0, t_cyc0 = 0, t_play = 0;
    if (stamps) { t_real0 = __builtin_amdgcn_s_memrealtime(); t_cyc0 = __builtin_amdgcn_s_memtime(); }
    __shared__ unsigned short refill_slot[TK_BLOCK];
    __shared__ u32 refill_ep[TK_BLOCK];
    __shared__ u32 refill_count;
    if (threadIdx.x == 0) refill_count = 0;
    __syncthreads();
    int64_t i = (int64_t)blockIdx.x * TK_BLOCK + threadIdx.x;
    bool valid = i < n;
    int64_t ic = valid ? i : n - 1;
    Game g;
    load_game(g, s01, s23, ic);
    u64 key = gkey[ic];
    bool autoreset = (flags & TAROK_AUTO_RESET) != 0;
    // lanes that can reach the end of their game within this launch (Berac: any trick end;
    // the others: only in trick 12) issue their finish-path loads together with the state
    bool berac = g.contract == TK_BERAC || g.contract == TK_ODPRTI_BERAC;
    bool spec = valid && ((g.phase == TK_PHASE_PLAY && (berac || (int)(g.trick_no * 4 + g.nt) + cards >= 48)) ||
                          g.phase == TK_PHASE_DONE);
    int4 acc = make_int4(0, 0, 0, 0);
    ulonglong2 na = make_ulonglong2(0, 0), nb = na;
    u32 cur_ep = 0;
    u64 nkey = 0;
    if (spec) {
        acc = aux[i].score_sum;
        if (autoreset) { na = aux[i].n01; nb = aux[i].n23; nkey = aux[i].nkey; cur_ep = aux[i].episode; }
    }
    bool have_next = (na.x >> 62) != 0;
    bool consumed = false, renewed = false, acc_dirty = false, seats_dirty = false, touched = false;
    // the legal mask written into the observation after card c is the one the policy needs for
    // card c+1: computed once per card, carried in a register
    u64 legal = (valid && g.phase == TK_PHASE_PLAY) ? legal_now(g) : 0;
    int64_t row = i;
    for (int c = 0; c < cards; c++, row += stride) {
        bool play = valid && g.phase == TK_PHASE_PLAY;
        u32 a = 255;
        if (play) a = policy_action(key, g.trick_no * 4 + g.nt, legal);
        u64 scores = 0;
        u32 trick_info = 0;
        int res = -2;
        if (play) res = apply_step<true>(g, a, scores, trick_info);
        bool fin = res == 1;
        touched = touched || res != -2;
        seats_dirty = seats_dirty || (res >= 0 && g.nt == 0);
        if (valid) {
            if (action_out) action_out[row] = (uint8_t)a;
            if (trick) trick[row] = (uint16_t)trick_info;
        }
        if (fin) {
            if (reward) reinterpret_cast<u64 *>(reward)[row] = scores;
            acc.x += (int16_t)(scores & 0xFFFF); acc.y += (int16_t)((scores >> 16) & 0xFFFF);
            acc.z += (int16_t)((scores >> 32) & 0xFFFF); acc.w += (int16_t)(scores >> 48);
            acc_dirty = true;
        }
        if (autoreset) {
            bool renew = valid && g.phase == TK_PHASE_DONE;
            if (__ballot(renew)) {
                bool swapped = false;
                if (renew && have_next) {
                    unpack(g, na.x, na.y, nb.x, nb.y);
                    key = nkey;
                    have_next = false; consumed = true; swapped = true;
                }
                bool deal_here = renew && !swapped;
                u64 pend = __ballot(deal_here);
                if (pend) {
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
                        setup_game(g, h0, h1, h2, h3, tal, cc, d, k);
                        if (g.phase == TK_PHASE_EXCHANGE) bot_exchange(g, dkey);
                        key = dkey;
                        consumed = true;          // its buffer is empty as well: refill below
                    }
                }
                if (renew) { cur_ep++; renewed = true; seats_dirty = true; }
            }
        }
        legal = (valid && g.phase == TK_PHASE_PLAY) ? legal_now(g) : 0;
        if (valid) {
            obs[row] = obs_word_with(g, fin, legal);
            if (done) done[row] = fin ? 1 : 0;
        }
    }
    if (valid) {
        if (acc_dirty) aux[i].score_sum = acc;
        if (renewed) { aux[i].episode = cur_ep; gkey[i] = key; }
        if (touched || renewed) store_game(g, s01, s23, i, seats_dirty);
    }
    if (stamps) t_play = __builtin_amdgcn_s_memtime() - t_cyc0;
    // refill the consumed (or missing) next-game buffers of this workgroup on dense lanes
    if (consumed) {
        u32 pos = atomicAdd(&refill_count, 1u);
        refill_slot[pos] = (unsigned short)threadIdx.x;
        refill_ep[pos] = cur_ep;
    }
    __syncthreads();
    if (threadIdx.x < refill_count) {
        int64_t j = (int64_t)blockIdx.x * TK_BLOCK + refill_slot[threadIdx.x];
        u64 k2 = game_key(seed, offset + (u64)j, (u64)refill_ep[threadIdx.x] + 1);
        u64 h0, h1, h2, h3, tal;
        deal_thread(k2, h0, h1, h2, h3, tal);
        u32 cc, d, k;
        sample_setup(k2, mix, cc, d, k);
        Game ng;
        setup_game(ng, h0, h1, h2, h3, tal, cc, d, k);
        if (ng.phase == TK_PHASE_EXCHANGE) bot_exchange(ng, k2);
        ulonglong2 pa, pb;
        pack(ng, pa.x, pa.y, pb.x, pb.y);
        aux[j].n23 = pb;
        aux[j].nkey = k2;
        aux[j].n01 = pa;
    }
    if (stamps && (threadIdx.x & 63) == 0) {     // diagnostics only
        u64 w = ((u64)blockIdx.x * TK_BLOCK + threadIdx.x) >> 6;
        stamps[3 * w + 0] = t_real0;
        stamps[3 * w + 1] = __builtin_amdgcn_s_memrealtime();
        stamps[3 * w + 2] = ((__builtin_amdgcn_s_memtime() - t_cyc0) << 32) | (t_play & 0xFFFFFFFFULL);
    }
}

// Whole games in registers: deal, setup, Bot exchange, random play to the end.
__global__ __launch_bounds__(TK_BLOCK) void k_rollout(int64_t n, u64 seed, u64 offset, u32 episode, int mix,
                                                     int16_t *__restrict__ scores_out, int16_t *__restrict__ nsteps_out,
                                                     int8_t *__restrict__ seats, u64 *__restrict__ masks,
                                                     uint8_t *__restrict__ actions) {
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
    int t = 0, played = 0;
    for (; t < 48; t++) {
        bool live = g.phase == TK_PHASE_PLAY;
        u64 m = 0;
        u32 a = 255;
        int seat = -1;
        if (live) {
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
__global__ __launch_bounds__(TK_BLOCK) void k_observe(int64_t n, const ulonglong2 *__restrict__ s01,
                                                     const ulonglong2 *__restrict__ s23, uint2 *__restrict__ out) {
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

// Masked categorical sample from policy logits, one thread per game: softmax over the legal
// cards only (mask = observation word), inverse-CDF draw with the spec RNG (draw 192 + cards
// played), log-probability of the drawn card.  Replaces ~10 framework kernels per step
// (bit-expand mask, masked_fill, log_softmax, multinomial, gather) with one pass over 8 MB.
__device__ __forceinline__ float bf16_to_f32(u32 h) { return __uint_as_float(h << 16); }

__global__ __launch_bounds__(TK_BLOCK) void k_sample(int64_t n, const uint4 *__restrict__ logits /* [N,64] bf16 */,
                                                    const u64 *__restrict__ obs, const u64 *__restrict__ gkey,
                                                    uint8_t *__restrict__ action, float *__restrict__ logp) {
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

__global__ __launch_bounds__(TK_BLOCK) void k_counters(int64_t n, const Aux *__restrict__ aux, u32 *__restrict__ ep,
                                                      int4 *__restrict__ score_sum) {
    int64_t i = (int64_t)blockIdx.x * TK_BLOCK + threadIdx.x;
    if (i >= n) return;
    if (ep) ep[i] = aux[i].episode;
    if (score_sum) score_sum[i] = aux[i].score_sum;
}

// canonical lanes for parity checks: H0-3, P0-3, TAL, META (tarok_env.h)
__global__ __launch_bounds__(TK_BLOCK) void k_get_state(int64_t n, const ulonglong2 *__restrict__ s01,
                                                       const ulonglong2 *__restrict__ s23, u64 *__restrict__ out) {
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
__global__ __launch_bounds__(TK_BLOCK) void k_set_state(int64_t n, const u64 *__restrict__ in,
                                                       ulonglong2 *__restrict__ s01, ulonglong2 *__restrict__ s23) {
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
    if (!out || n_games <= 0 || n_games > (1LL << 31)) return TAROK_EINVAL;
    if (!(mix == TAROK_MIX_ALL || mix == TAROK_MIX_NAVADNA3 || mix == TAROK_MIX_BOT || (mix >= TAROK_MIX_FIXED && mix < TAROK_MIX_FIXED + 10))) return TAROK_EINVAL;
    if (device < 0 || device >= tarok_device_count()) return TAROK_ENODEV;
    HIPCHK(hipSetDevice(device));
    tarok_env *e = new tarok_env();
    memset(e, 0, sizeof *e);
    e->device = device; e->n = n_games; e->offset = game_offset; e->seed = seed; e->mix = mix; e->flags = flags;
    size_t stale_bytes = (size_t)((n_games + TK_PF_SLOTS - 1) / TK_PF_SLOTS) * TK_PF_SLOTS;
    hipError_t r = hipMalloc((void **)&e->s01, (size_t)n_games * sizeof(ulonglong2));
    if (r == hipSuccess) r = hipMalloc((void **)&e->s23, (size_t)n_games * sizeof(ulonglong2));
    if (r == hipSuccess) r = hipMalloc((void **)&e->aux, (size_t)n_games * sizeof(Aux));
    if (r == hipSuccess) r = hipMalloc((void **)&e->nstale, stale_bytes);
    if (r == hipSuccess) r = hipMalloc((void **)&e->gkey, (size_t)n_games * sizeof(u64));
    if (r == hipSuccess) r = hipMemset(e->s01, 0, (size_t)n_games * sizeof(ulonglong2));
    if (r == hipSuccess) r = hipMemset(e->s23, 0, (size_t)n_games * sizeof(ulonglong2));
    if (r == hipSuccess) r = hipMemset(e->aux, 0, (size_t)n_games * sizeof(Aux));
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

void tarok_destroy(tarok_env *e) {
    if (!e) return;
    (void)hipSetDevice(e->device);
    if (e->gexec) (void)hipGraphExecDestroy(e->gexec);
    if (e->cap_stream) (void)hipStreamDestroy(e->cap_stream);
    (void)hipFree(e->s01); (void)hipFree(e->s23); (void)hipFree(e->aux); (void)hipFree(e->nstale); (void)hipFree(e->gkey);
    delete e;
}

int64_t tarok_num_games(const tarok_env *e) { return e ? e->n : 0; }

static inline void launch_prefetch(tarok_env *e, hipStream_t s) {
    dim3 grid((unsigned)((e->n + TK_PF_SLOTS - 1) / TK_PF_SLOTS));
    hipLaunchKernelGGL(k_prefetch, grid, dim3(TK_BLOCK), 0, s, e->n, e->seed, e->offset, e->mix, e->aux, e->nstale);
}

int tarok_reset(tarok_env *e, uint32_t episode, const uint8_t *deals, const int8_t *contract, const int8_t *declarer,
                const int8_t *king_suit, const int8_t *talon_choice, const uint8_t *discards, int flags, void *stream) {
    if (!e) return TAROK_EINVAL;
    if ((talon_choice == nullptr) != (discards == nullptr)) return TAROK_EINVAL;
    if (contract && !declarer) return TAROK_EINVAL;
    HIPCHK(hipSetDevice(e->device));
    hipLaunchKernelGGL(k_reset, grid_for(e->n), dim3(TK_BLOCK), 0, (hipStream_t)stream, e->n, e->seed, e->offset,
                       episode, e->mix, flags, deals, contract, declarer, king_suit, talon_choice, discards, e->s01,
                       e->s23, e->aux, e->nstale, e->gkey);
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

static inline void launch_step(tarok_env *e, bool random, const uint8_t *action, uint8_t *action_out,
                               int16_t *reward, uint8_t *done, uint64_t *obs, int flags, hipStream_t s,
                               uint16_t *trick = nullptr) {
    if (random)
        hipLaunchKernelGGL(k_step<true>, grid_for(e->n), dim3(TK_BLOCK), 0, s, e->n, e->seed, e->offset, e->mix,
                           flags, action, action_out, reward, done, trick, (u64 *)obs, e->s01, e->s23, e->aux, e->nstale,
                           e->gkey, e->stamps);
    else
        hipLaunchKernelGGL(k_step<false>, grid_for(e->n), dim3(TK_BLOCK), 0, s, e->n, e->seed, e->offset, e->mix,
                           flags, action, action_out, reward, done, trick, (u64 *)obs, e->s01, e->s23, e->aux, e->nstale,
                           e->gkey, e->stamps);
}

int tarok_step(tarok_env *e, const uint8_t *action, int16_t *reward_out, uint8_t *done_out, uint16_t *trick_out,
               uint64_t *obs_out, int flags, void *stream) {
    if (!e || !action || !obs_out) return TAROK_EINVAL;
    HIPCHK(hipSetDevice(e->device));
    launch_step(e, false, action, nullptr, reward_out, done_out, obs_out, flags, (hipStream_t)stream, trick_out);
    HIPCHK(hipGetLastError());
    return TAROK_OK;
}

int tarok_policy_random(tarok_env *e, const uint64_t *obs, uint8_t *action_out, void *stream) {
    if (!e || !obs || !action_out) return TAROK_EINVAL;
    HIPCHK(hipSetDevice(e->device));
    hipLaunchKernelGGL(k_policy, grid_for(e->n), dim3(TK_BLOCK), 0, (hipStream_t)stream, e->n, (const u64 *)obs,
                       e->gkey, action_out);
    HIPCHK(hipGetLastError());
    return TAROK_OK;
}

int tarok_step_random(tarok_env *e, uint8_t *action_out, int16_t *reward_out, uint8_t *done_out, uint16_t *trick_out,
                      uint64_t *obs_out, int flags, void *stream) {
    if (!e || !obs_out) return TAROK_EINVAL;
    HIPCHK(hipSetDevice(e->device));
    launch_step(e, true, nullptr, action_out, reward_out, done_out, obs_out, flags, (hipStream_t)stream, trick_out);
    HIPCHK(hipGetLastError());
    return TAROK_OK;
}

static inline void launch_krog(tarok_env *e, int cards, int64_t stride, uint8_t *action_out, int16_t *reward,
                               uint8_t *done, uint16_t *trick, uint64_t *obs, int flags, hipStream_t s) {
    hipLaunchKernelGGL(k_krog, grid_for(e->n), dim3(TK_BLOCK), 0, s, e->n, e->seed, e->offset, e->mix, flags, cards,
                       stride, action_out, reward, done, trick, (u64 *)obs, e->s01, e->s23, e->aux, e->nstale, e->gkey, e->stamps);
}

int tarok_krog_random(tarok_env *e, int cards, int64_t stride, uint8_t *action_out, int16_t *reward_out,
                      uint8_t *done_out, uint16_t *trick_out, uint64_t *obs_out, int flags, void *stream) {
    if (!e || !obs_out || cards < 1 || cards > 48 || stride < e->n) return TAROK_EINVAL;
    HIPCHK(hipSetDevice(e->device));
    launch_krog(e, cards, stride, action_out, reward_out, done_out, trick_out, obs_out, flags, (hipStream_t)stream);
    HIPCHK(hipGetLastError());
    return TAROK_OK;
}

// cards: 0 = tarok_policy_random + tarok_step, 1 = tarok_step_random, >= 2 = tarok_krog_random
static inline void launch_one(tarok_env *e, int cards, uint8_t *action, int16_t *reward, uint8_t *done,
                              uint64_t *obs, int flags, hipStream_t s) {
    if (cards >= 2) {
        launch_krog(e, cards, e->n, action, reward, done, nullptr, obs, flags, s);
    } else if (cards == 1) {
        launch_step(e, true, nullptr, nullptr, reward, done, obs, flags, s);
    } else {
        hipLaunchKernelGGL(k_policy, grid_for(e->n), dim3(TK_BLOCK), 0, s, e->n, (const u64 *)obs, e->gkey, action);
        launch_step(e, false, action, nullptr, reward, done, obs, flags, s);
    }
}

int tarok_run_random(tarok_env *e, int64_t n_steps, int cards_per_launch, int graph_chunk, int prefetch_every,
                     uint8_t *action, int16_t *reward_out, uint8_t *done_out, uint64_t *obs_out, int flags, void *stream) {
    if (!e || !obs_out || n_steps < 0 || graph_chunk < 0 || graph_chunk > 4096 || prefetch_every < 0) return TAROK_EINVAL;
    if (cards_per_launch < 0 || cards_per_launch > 48) return TAROK_EINVAL;
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
        bool hit = e->gexec && e->g_fused == cards_per_launch && e->g_chunk == graph_chunk && e->g_flags == flags &&
                   e->g_prefetch == prefetch_every &&
                   e->g_action == action && e->g_reward == reward_out && e->g_done == done_out && e->g_obs == obs_out;
        if (!hit) {
            if (e->gexec) { (void)hipGraphExecDestroy(e->gexec); e->gexec = nullptr; }
            hipGraph_t graph = nullptr;
            HIPCHK(hipStreamBeginCapture(e->cap_stream, hipStreamCaptureModeRelaxed));
            for (int k = 0; k < graph_chunk; k += unit) {
                launch_one(e, cards_per_launch, action, reward_out, done_out, obs_out, flags, e->cap_stream);
                if (prefetch_every && (k + unit) % prefetch_every == 0) launch_prefetch(e, e->cap_stream);
            }
            HIPCHK(hipStreamEndCapture(e->cap_stream, &graph));
            hipError_t r = hipGraphInstantiate(&e->gexec, graph, nullptr, nullptr, 0);
            (void)hipGraphDestroy(graph);
            if (r != hipSuccess) { e->gexec = nullptr; g_last_hip = (int)r; return TAROK_EHIP; }
            e->g_fused = cards_per_launch; e->g_chunk = graph_chunk; e->g_flags = flags; e->g_prefetch = prefetch_every;
            e->g_action = action; e->g_reward = reward_out; e->g_done = done_out; e->g_obs = obs_out;
        }
        while (left >= graph_chunk) {
            HIPCHK(hipGraphLaunch(e->gexec, s));
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

int tarok_debug_stamps(tarok_env *e, uint64_t *stamps) {
    if (!e) return TAROK_EINVAL;
    e->stamps = (u64 *)stamps;
    if (e->gexec) { (void)hipGraphExecDestroy(e->gexec); e->gexec = nullptr; }
    return TAROK_OK;
}

int tarok_observe(tarok_env *e, void *features_out, void *stream) {
    if (!e || !features_out) return TAROK_EINVAL;
    HIPCHK(hipSetDevice(e->device));
    hipLaunchKernelGGL(k_observe, grid_for(e->n), dim3(TK_BLOCK), 0, (hipStream_t)stream, e->n, e->s01, e->s23,
                       (uint2 *)features_out);
    HIPCHK(hipGetLastError());
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
                       e->s01, e->s23);
    HIPCHK(hipGetLastError());
    return TAROK_OK;
}

int tarok_get_counters(tarok_env *e, uint32_t *episode_out, int32_t *score_sum_out, void *stream) {
    if (!e) return TAROK_EINVAL;
    HIPCHK(hipSetDevice(e->device));
    hipLaunchKernelGGL(k_counters, grid_for(e->n), dim3(TK_BLOCK), 0, (hipStream_t)stream, e->n, e->aux, episode_out,
                       (int4 *)score_sum_out);
    HIPCHK(hipGetLastError());
    return TAROK_OK;
}

}  // extern "C"
