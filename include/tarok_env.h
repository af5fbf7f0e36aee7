/* libtarokenv — MI355X-native vectorised Tarok card-play environment, C ABI.
 *
 * Drop-in boundary for ONE hot path of anzeA/Tarok: stepping N independent
 * 4-player games in lock-step — legal-card mask, trick resolution, and
 * Klop / Berac / Navadna_igra scoring.  It replaces what the reference does in
 *
 *     Tarok.paralel_start          Tarok.py:30-62     (N games, 48 lock-steps)
 *     Igra.razdeli + dispatch      Igra.py:38-55,65-73 (deal -> contract engine)
 *     Klop / Berac / Navadna_igra  .start()/.krog()/.mozne_karte()/.pobere_stih()
 *     Roka.prestej                 Roka.py:56-98
 *
 * The reference has no FFI: its "plugin API" is the duck-typed Igralec callback
 * set (Igralec.py:32-122).  Each entry point below names the callback /
 * generator step it stands in for; INTEGRATION.md shows the ctypes binding and
 * the Igralec-protocol adapter (tarok_amd/igralec.py) that sits on top.
 *
 * Conventions
 *   - every function returns int: 0 = OK, negative = TAROK_E*; no exceptions.
 *   - all array arguments are DEVICE pointers owned by the caller (e.g. torch
 *     tensors' data_ptr()), SoA over the env's N games; NULL where allowed.
 *   - `stream` is a hipStream_t (NULL = default stream); every call is
 *     stream-ordered and non-blocking.  One handle per GPU, one host thread per
 *     handle (the reference is single-threaded cooperative generators).
 *   - card id = suit*8 + rank-1 (suits KARA=0 SRCE=1 PIK=2 KRIZ=3), taroks
 *     32..53 (Karta.py:19-23).  Masks are 54-bit sets of card ids.
 *   - contract code = int(Tip_igre)/10 (Tip_igre.py:4-15): 0 Klop, 1 Tri, 2 Dve,
 *     3 Ena, 4 Solo_tri, 5 Solo_dve, 6 Solo_ena, 7 Berac, 8 Solo_brez,
 *     9 Odprti_berac.  Seats are positions in a game's `igralci` list.
 */
#ifndef TAROK_ENV_H
#define TAROK_ENV_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define TAROK_ABI_VERSION 5
#define TAROK_MAX_CARDS_PER_LAUNCH 192 /* tarok_krog_random / tarok_run_random: cards of every game per launch */
#define TAROK_GAMES_AHEAD 14           /* games every slot keeps dealt ahead for TAROK_AUTO_RESET (tarok_prefetch)  */

#define TAROK_OK 0
#define TAROK_EINVAL (-1) /* bad argument                                  */
#define TAROK_EHIP (-2)   /* HIP runtime error: see tarok_last_hip_error() */
#define TAROK_ENOMEM (-3)
#define TAROK_ENODEV (-4) /* no usable GPU                                 */

/* synthetic contract mixes used when tarok_reset gets contract == NULL */
#define TAROK_MIX_ALL 0      /* 1/3 Klop, 1/3 Berac (1/2 open), 1/3 Navadna+Solo over 7 types */
#define TAROK_MIX_NAVADNA3 1 /* Tri / Dve / Ena uniform                                       */
#define TAROK_MIX_BOT 2      /* a bidding round between four Bot players decides (Igra.py:75-114,  */
                             /* Igralec.py:148-156): the reference's own contract distribution    */
#define TAROK_MIX_FIXED 16   /* TAROK_MIX_FIXED + code: every game plays that contract         */

/* flags */
#define TAROK_DEFER_EXCHANGE 1 /* tarok_reset: leave Tri..Solo_ena games waiting for tarok_exchange */
#define TAROK_AUTO_RESET 2     /* step kernels: re-deal a game in the launch that finishes it        */
#define TAROK_CLEAR_COUNTERS 4 /* tarok_reset: also zero the per-slot score sums                     */
#define TAROK_REWARD_REF 8     /* step kernels: reward_out carries what rezultat_igre folds into the last       */
                               /* transition (Igralec.py:421-437): the scores, except that a Berac / Odprti_berac  */
                               /* DEFENDER gets -20 when the hands are empty at the end, else +20               */
#define TAROK_HISTORY 16       /* tarok_create: keep the play history (48 bytes per game), needed by            */
                               /* tarok_observe_ref                                                              */

/* observation word written by tarok_legal_actions / tarok_step (one u64 per game) */
#define TAROK_OBS_MASK ((1ULL << 54) - 1) /* [53:0]  legal-card mask of the seat to move  */
#define TAROK_OBS_SEAT_SHIFT 54           /* [55:54] seat to move                          */
#define TAROK_OBS_STEP_SHIFT 56           /* [61:56] cards played so far in this game      */
#define TAROK_OBS_DONE (1ULL << 62)       /* game finished (step: by this very step)       */
#define TAROK_OBS_ERROR (1ULL << 63)      /* an illegal action was rejected (sticky)       */

/* canonical state lanes of tarok_get_state: lanes_out[lane*N + g] */
#define TAROK_LANE_HAND0 0 /* 0..3  Igralec.roka[id] of seat s                     */
#define TAROK_LANE_PILE0 4 /* 4..7  Igralec.kupcek[id] of seat s                   */
#define TAROK_LANE_TALON 8 /* 6 x 6-bit ordered talon ids (Igra.py:68)             */
#define TAROK_LANE_META 9
#define TAROK_NUM_LANES 10
/* META: [23:0] trick cards 4x6 | [26:24] n_in_trick | [28:27] leader |
 * [32:29] trick_no | [36:33] contract | [38:37] declarer | [41:39] king (7 none) |
 * [45:42] team | [48:46] talon_left | [51:49] chosen group (7 none) |
 * [53:52] phase (1 exchange, 2 play, 3 done) | [54] error                      */

typedef struct tarok_env tarok_env;

const char *tarok_strerror(int code);
int tarok_abi_version(void);
int tarok_device_count(void);      /* number of visible GPUs, 0 if none (never fails) */
int tarok_last_hip_error(void);    /* last hipError_t seen by this library            */

/* One env = n_games slots on one GPU.  game_offset = global index of slot 0
 * (the deal RNG is keyed by seed, game_offset+g and the slot's episode number,
 * so any sharding over GPUs plays identical games).  Replaces building N Igra
 * objects, Tarok.py:33-35.  flags: 0, or TAROK_HISTORY (keep the play history:
 * the `zgodovina` list of Klop.py:63 / Navadna_igra.py:127 as one byte per card). */
int tarok_create(tarok_env **out, int device, int64_t n_games, uint64_t game_offset,
                 uint64_t seed, int mix, int flags);
void tarok_destroy(tarok_env *env);
int64_t tarok_num_games(const tarok_env *env);

/* Launch tuning of an env; the results never depend on it, at whatever point of a run it is changed.  Defaults from the
 * batch size at tarok_create.
 *   TAROK_OPT_REFILL_FAN   1..8 play workgroups whose refill lists one refill workgroup works off.  It sets the step
 *                               launches' grid: changing it after the env's first step launch synchronises the DEVICE,
 *                               restarts the env's launch counters and empties its refill lists (lines whose deal is
 *                               dropped are dealt by their slots when they get there) — do it between phases, not per step,
 *                               and not while a caller's graph holding this env's step launches is to be replayed
 *   TAROK_OPT_LAZY_REFILL  0/1  tarok_step / tarok_step_random: the next-game lines their slots empty are dealt in bulk
 *                               every thirty-second launch instead of in the launch after (default: on below 2^20 games);
 *                               free to change between any two launches */
#define TAROK_OPT_REFILL_FAN 2
#define TAROK_OPT_LAZY_REFILL 3
int tarok_set_option(tarok_env *env, int option, int value);

/* Deal and set up every game: Igra.razdeli (Igra.py:65-73) + the contract
 * engine constructors (Igra.py:38-55; teams Navadna_igra.py:20-30; first
 * leader Klop.py:26 / Berac.py:15 / Navadna_igra.py:70) + the talon exchange
 * (Navadna_igra.py:36-66).  Every slot's episode number is set to `episode`.
 *   deals        [N,54] u8 permutation (hand s = deals[12s:12s+12], talon = deals[48:54]);
 *                NULL = dealt on device by the spec RNG
 *   contract     [N] i8 code; NULL = sampled from the env's mix (then declarer
 *                and king_suit are sampled too)
 *   declarer     [N] i8 seat 0..3 (ignored for Klop); king_suit [N] i8 0..3 (Tri/Dve/Ena)
 *   talon_choice [N] i8 chosen talon group, discards [N,3] u8 (255 pad): what
 *                Igralec.menjaj_iz_talona returns / moves (Igralec.py:161-171);
 *                NULL = the Bot's exchange (group 0, random discardable cards)
 *                unless flags has TAROK_DEFER_EXCHANGE. */
int tarok_reset(tarok_env *env, uint32_t episode, const uint8_t *deals, const int8_t *contract,
                const int8_t *declarer, const int8_t *king_suit, const int8_t *talon_choice,
                const uint8_t *discards, int flags, void *stream);

/* menjaj_iz_talona for the games still waiting for it (Navadna_igra.py:60-66).
 * NULL arrays = the Bot's exchange.  Games not in the exchange phase are untouched. */
int tarok_exchange(tarok_env *env, const int8_t *talon_choice, const uint8_t *discards, void *stream);

/* mozne_karte for the seat to move in every game (Klop.py:96-133,
 * Navadna_igra.py:158-168) = the `mozne` handed to pripravi_igraj_karto.
 * obs_out [N] u64 observation words; seat_out [N] i8 or NULL. */
int tarok_legal_actions(tarok_env *env, uint64_t *obs_out, int8_t *seat_out, void *stream);

/* One card in every unfinished game = one `next(g)` per game at Tarok.py:54:
 * the body of krog (Klop.py:47-79, Navadna_igra.py:115-141) after igraj_karto
 * returned `action`, incl. trick resolution (pobere_stih), the Klop talon gift,
 * Berac's early end and end-of-game scoring.
 *   action     [N] u8 card id.  Not in the legal set -> the game is left
 *              unchanged and its error bit is set (reference: raises Exception,
 *              Klop.py:57-60).  Finished / waiting games ignore it.
 *   reward_out [N,4] i16 scores by seat (`pisejo`): written ONLY for games that
 *              finish in this step; may be NULL
 *   done_out   [N] u8 1 iff the game finished in this step; may be NULL
 *   trick_out  [N] u16, may be NULL: what rezultat_stiha(stih, sem_pobral) is told
 *              (Klop.py:76-77, Navadna_igra.py:138-139).  0 unless this step completed a
 *              trick; then 0x8000 | Roka.vrednost_stiha(stih) << 4 | seat that took it
 *              (Roka.py:76-95; stih = the 4 cards, 5 with Klop's talon card)
 *   obs_out    [N] u64 observation for the NEXT move (see TAROK_OBS_*)
 *   flags      TAROK_AUTO_RESET: a game that finishes is replaced at once by the slot's
 *              next game (episode+1, synthetic contract, Bot exchange, dealt ahead of
 *              time); obs_out then describes the new game and keeps TAROK_OBS_DONE set. */
int tarok_step(tarok_env *env, const uint8_t *action, int16_t *reward_out, uint8_t *done_out,
               uint16_t *trick_out, uint64_t *obs_out, int flags, void *stream);

/* Fill, synchronously, every next-game line that tarok_reset emptied (each slot keeps its next
 * FOURTEEN games dealt ahead: episode+1 .. episode+14, synthetic contract, Bot exchange).  tarok_reset
 * calls it; afterwards the step kernels keep the lines full themselves — a launch that swaps a
 * finished game's successor in puts the replacement deal on a list that extra workgroups of the
 * NEXT step launch work off while that launch plays — so callers normally never need this.  A
 * slot that finds its line missing anyway (more than fourteen games finished within two consecutive
 * launches) deals the game inside the step kernel (same result). */
int tarok_prefetch(tarok_env *env, void *stream);

/* Bot_igralec.igraj_karto (Igralec.py:158-159): uniform choice among the legal
 * cards, drawn from the spec RNG (draw 128 + cards played).  action_out[g] = 255
 * where nothing is to be played. */
int tarok_policy_random(tarok_env *env, const uint64_t *obs, uint8_t *action_out, void *stream);

/* tarok_policy_random + tarok_step fused in one launch; action_out may be NULL. */
int tarok_step_random(tarok_env *env, uint8_t *action_out, int16_t *reward_out, uint8_t *done_out,
                      uint16_t *trick_out, uint64_t *obs_out, int flags, void *stream);

/* `cards` cards of every game in one launch with the Bot policy in-kernel; cards = 4 is one trick,
 * i.e. one pass of the reference's krog generator (Klop.py:47-79, Navadna_igra.py:115-141).
 * The state is read once, kept in registers and written once, but every per-card output is
 * still written: row c (c = 0..cards-1, rows `stride` >= N games apart) of action_out [cards,stride] u8,
 * reward_out [cards,stride,4] i16 (only where done), done_out, trick_out [cards,stride] u16 and
 * obs_out [cards,stride] u64 is what tarok_step_random would have written for the c-th card.
 * action_out / reward_out / done_out / trick_out may be NULL.  Finished games are replaced at
 * once with TAROK_AUTO_RESET (from the slot's dealt-ahead lines; a game can be over after 4 cards,
 * so a slot may start several games inside one launch). */
int tarok_krog_random(tarok_env *env, int cards, int64_t stride, uint8_t *action_out, int16_t *reward_out,
                      uint8_t *done_out, uint16_t *trick_out, uint64_t *obs_out, int flags, void *stream);

/* n_steps lock-steps of the random policy, launched from C (optionally as a
 * replayed hipGraph of `graph_chunk` steps; 0 = eager launches).
 * cards_per_launch = 0: tarok_policy_random + tarok_step per step;  1: tarok_step_random;
 * c >= 2: tarok_krog_random(c) — n_steps, graph_chunk and prefetch_every must be multiples of c
 * and the buffers hold c rows of N (action [c,N], reward_out [c,N,4], done_out [c,N], obs_out [c,N]).
 * prefetch_every = k > 0: an extra tarok_prefetch after every k-th step (graph_chunk must be a
 * multiple of k); normally 0.  Any number of launches per graph: the refill-list parity lives in
 * device memory and is advanced by the launches themselves, so graph replays (this library's or a
 * caller's own capture of tarok_* calls) and eager launches may be mixed in any order.
 * Buffers otherwise as in tarok_step (action [N] u8 scratch is required for cards_per_launch = 0). */
int tarok_run_random(tarok_env *env, int64_t n_steps, int cards_per_launch, int graph_chunk,
                     int prefetch_every, uint8_t *action, int16_t *reward_out, uint8_t *done_out,
                     uint64_t *obs_out, int flags, void *stream);

/* Whole games in one launch (state never leaves registers): deal + setup + Bot
 * exchange + random play to the end, for episode `episode` of every slot.
 *   scores_out [N,4] i16, nsteps_out [N] i16 (cards played; Berac may stop early)
 *   optional step-major traces, [48,N] each, padded with -1 / 0 / 255:
 *   seats_out i8, masks_out u64, actions_out u8.
 * Does not touch the env's stepping state. */
int tarok_rollout_random(tarok_env *env, uint32_t episode, int16_t *scores_out, int16_t *nsteps_out,
                         int8_t *seats_out, uint64_t *masks_out, uint8_t *actions_out, void *stream);

/* Observation features of the seat to move for a policy network: features_out [N,256] bf16,
 * every entry 0.0 or 1.0 (feature set documented at k_observe — the build's own,
 * the reference's encoder is part of its LSTM agent, Igralec.py:453-543):
 *   [0,54) own hand, [54,64) contract one-hot;  [64,118) legal cards, [118,128) declarer
 *   relative seat (4) / cards on table count (4) / on declarer's team / contract calls a king;
 *   [128,182) cards on the table, [182,192) called-king suit (4) / trick number binary (4);
 *   [192,246) cards already taken, [246] game live. */
#define TAROK_OBS_FEATURES 256
int tarok_observe(tarok_env *env, void *features_out, void *stream);

/* ---- the reference's own observation layout (parity unpinned: the reference holds no
 * fixture for it and Igralec.py cannot be imported — tested against the line-cited restatement
 * oracle/encoder_spec.py).  Needs an env created with TAROK_HISTORY.
 *
 * tarok_observe_ref: what Nevronski_igralec.stanje_v_vektor_rek_navadna (Igralec.py:453-533) builds
 * for the seat to move of every game in play, as ONE record of 0/1 bytes per game:
 *   record_out [N, TAROK_REF_RECORD_BYTES] u8, fields at the byte offsets below (row-major):
 *     TAROK_REF_OPP      [56][3][54]  input_layer_nasprotiki: row i = the i-th card played in the game, one-hot
 *                                     card in the channel of the opponent who played it (the other three
 *                                     seats in seat order, Igralec.py:271-274); own plays leave the row empty
 *     TAROK_REF_OWN      [56][54]     roka_input: row i of an own play = the hand vector before that card; it
 *                                     starts from the hand as dealt (zacetna_roka; the exchange is not applied)
 *     TAROK_REF_TALON    [6][55]      talon_input of Tri..Solo_ena (row r: the r-th talon card, column 54: in the
 *                                     chosen group; all zero for Solo_brez); for Klop the first 54 bytes are its
 *                                     flat talon vector (the cards gifted so far); zero for Berac
 *     TAROK_REF_KING     [4]          barva_kralja one-hot (Tri/Dve/Ena)
 *     TAROK_REF_INDEX    [4]          index_tistega_ki_igra one-hot: the declarer among the other seats in seat
 *                                     order, 3 = the mover himself
 *     TAROK_REF_DISCARDS [54]         zalozil (only when the mover is the player who exchanged)
 *     TAROK_REF_LEGAL    [54]         mozne_vec; 2 bytes of padding follow
 *   meta_out [N,4] i32 (may be NULL): {T, network type (0 Klop, 1 Navadna_igra, 2 Solo, 3 Berac =
 *     Nevronski_igralec.Tipi_NN), rows used (cards played so far), seat to move}.  T is the reference's first
 *     dimension of the two history tensors: the entries of `zgodovina` (cards + the "Talon" entry + Klop's
 *     talon cards: the counter at Igralec.py:456-458 counts them all) rounded up to the next multiple of 8,
 *     plus 8 when already one (:460); T <= 56; the reference's tensors are rows [0, T) of the record's.
 *   Games not in play (finished, waiting for the exchange): an all-zero record and T = 0.
 *   Which tensors a network type consumes (Igralec.py:520-531): Navadna_igra opp, king, own, talon, index,
 *   discards, legal; Solo the same without king; Klop opp, own, talon(54), legal; Berac opp, own, index, legal. */
#define TAROK_REF_ROWS 56
#define TAROK_REF_OPP 0
#define TAROK_REF_OWN 9072
#define TAROK_REF_TALON 12096
#define TAROK_REF_KING 12426
#define TAROK_REF_INDEX 12430
#define TAROK_REF_DISCARDS 12434
#define TAROK_REF_LEGAL 12488
#define TAROK_REF_RECORD_BYTES 12544
int tarok_observe_ref(tarok_env *env, uint8_t *record_out, int32_t *meta_out, void *stream);

/* menjaj_talon_v_vektor (Igralec.py:535-543), the input of the talon-exchange decision, for every game that
 * waits for tarok_exchange: out [N, TAROK_REF_EXCHANGE_BYTES] u8 = roka [54] (the declarer's hand) | talon
 * (54,6) flattened card-major (card c lies in group i: byte 54 + 6c + i) | igra one-hot [15]
 * (igra_zalozi2index, Igralec.py:717-745: (Tri,Dve,Ena) x called suit -> 0..11, Solo_tri/dve/ena -> 12..14)
 * | 7 bytes of padding.  Zero for games in any other phase. */
#define TAROK_REF_EXCHANGE_BYTES 400
int tarok_observe_exchange_ref(tarok_env *env, uint8_t *out, void *stream);

/* The bidding input (pripavi_licitiram, Igralec.py:278-281): out [N,4,54] u8, every seat's hand one-hot. */
int tarok_observe_hands_ref(tarok_env *env, uint8_t *out, void *stream);

/* The play history of a TAROK_HISTORY env, hist [48,N] u8 (device): card p of every slot's current game
 * (entries at or beyond the cards played so far are stale).  tarok_set_state does not touch it: a
 * checkpoint of such an env is the canonical lanes plus this array. */
int tarok_get_history(tarok_env *env, uint8_t *hist_out, void *stream);
int tarok_set_history(tarok_env *env, const uint8_t *hist_in, void *stream);

/* A learned player's igraj_karto (cf. Igralec.py:344-355): sample one LEGAL card per game from
 * policy logits.  logits [N,64] bf16 (card c in column c, columns 54..63 ignored), obs [N] the
 * observation words (legal mask + cards played), softmax over the legal cards only, draw from the
 * spec RNG (draw 192 + cards played).  action_out [N] u8 (255 where nothing is to be played),
 * logp_out [N] f32 log-probability of the drawn card (may be NULL). */
int tarok_sample_policy(tarok_env *env, const void *logits_bf16, const uint64_t *obs,
                        uint8_t *action_out, float *logp_out, void *stream);

/* A whole learned-policy step in one launch: observation features (as tarok_observe) -> MLP
 * 256 -> 256 -> 256 -> 64 with ReLU (bf16 MFMA, f32 accumulate; outputs 0..53 = card logits,
 * output 54 = state value) -> masked categorical sample (as tarok_sample_policy).
 *   w1, w2 (256 outputs), w3 (64 outputs), bf16, 256 inputs each, in MFMA fragment order: the
 *   16-byte element ((o / 32) * 16 + k / 16) * 64 + ((k / 8) % 2) * 32 + o % 32 holds
 *   W[o][8 * (k / 8) .. + 7] of the [out][in] matrix (torch.nn.Linear.weight), i.e.
 *   W.view(out/32, 32, 16, 2, 8).permute(0, 2, 3, 1, 4) — a wave's weight load is then contiguous;
 *   b1, b2 [256] f32, b3 [64] f32;  obs [N] observation words;
 *   action_out [N] u8, logp_out [N] f32 (may be NULL), value_out [N] f32 (may be NULL),
 *   features_out [N,256] bf16 (may be NULL): the features, for the learner's update;
 *   feature_words_out [N,4] u64 (may be NULL): the same 256 features as bits (feature f = bit
 *   f % 64 of word f / 64) — 32 bytes per game instead of 512, for a learner that expands its
 *   minibatches itself. */
int tarok_policy_mlp(tarok_env *env, const void *w1, const float *b1, const void *w2, const float *b2,
                     const void *w3, const float *b3, const uint64_t *obs, uint8_t *action_out,
                     float *logp_out, float *value_out, void *features_out, uint64_t *feature_words_out,
                     void *stream);

/* tarok_policy_mlp followed by tarok_step(action_out, ...) in ONE launch: the workgroup that
 * evaluated the policy for 256 games plays the sampled cards at once (same results as the two
 * calls; saves a launch and its gap per lock-step).  Arguments as in those two functions; obs is
 * the observation BEFORE the step (legal masks), obs_out the one after it (they may not alias). */
int tarok_policy_step(tarok_env *env, const void *w1, const float *b1, const void *w2, const float *b2,
                      const void *w3, const float *b3, const uint64_t *obs, uint8_t *action_out,
                      float *logp_out, float *value_out, uint64_t *feature_words_out, int16_t *reward_out,
                      uint8_t *done_out, uint16_t *trick_out, uint64_t *obs_out, int flags, void *stream);

/* The learner's network input for a minibatch: features_out[j] [256] bf16 (0.0 / 1.0) = the bits
 * of feature_words[index[j]] (feature_words [M,4] u64 as written by feature_words_out; index
 * [n_samples] i64 sample numbers, or NULL for samples 0..n_samples-1).  Gather + expansion in one
 * pass. */
int tarok_expand_features(tarok_env *env, int64_t n_samples, const uint64_t *feature_words,
                          const int64_t *index, void *features_out, void *stream);

/* The learner's loss for the policy above, forward and gradient in one pass: clipped-surrogate
 * policy loss + value loss - entropy bonus over the LEGAL cards of every sample (build-owned:
 * the reference has no policy-gradient learner).
 *   out_bf16 [B,64] bf16 head outputs (0..53 card logits, 54 value), obs [B] observation words
 *   (legal mask), action [B] i64 the card played, logp_old [B] f32 its log-probability at play
 *   time, advantage / ret / weight [B] f32 (weight 0: sample ignored);
 *   loss = sum_i w_i (-min(r_i A_i, clamp(r_i, 1-clip, 1+clip) A_i) + vf (v_i - ret_i)^2 - ent H_i)
 *          * inv_weight_sum[0] (a device scalar, so that no host round trip sits between the
 *          minibatch's weight sum and this launch),   r_i = exp(logp_i[a_i] - logp_old_i), H_i the
 *          entropy of the masked softmax;
 *   dout_bf16 [B,64] bf16 = d loss / d out;  partial_out [ceil(B/256),4] f32 = per-workgroup sums of
 *   {w pi, w (v-ret)^2, w H, 0} (sum them for the loss terms). */
int tarok_ppo_loss(tarok_env *env, int64_t n_samples, const void *out_bf16, const uint64_t *obs,
                   const int64_t *action, const float *logp_old, const float *advantage, const float *ret,
                   const float *weight, float clip, float vf_coef, float ent_coef, const float *inv_weight_sum,
                   void *dout_bf16, float *partial_out, void *stream);

/* The reference agent's training targets in its own form (Nevronski_igralec.rezultat_stiha +
 * rezultat_igre, Igralec.py:387-446) for a recorded rollout of T lock-steps, T a multiple of 4, started on a trick
 * boundary (trick b = rows 4b .. 4b+3 of every slot, through the auto-resets):
 *   obs_before [T,N] u64   the observation word each card was chosen on (legal mask = `mozne`, seat to move)
 *   action [T,N] u8, trick [T,N] u16 (trick_out), done [T,N] u8, reward [T,N,4] i16 with TAROK_REWARD_REF
 *   next_q [T,N] f32 or NULL  the agent's own next_Q_max at each decision (Igralec.py:351); NULL = 0
 *   dy_out [T/4,N,4,54] f32: per (trick, game, seat) -70 on the cards that were not legal (:392-393), on the card
 *           played +/- Roka.vrednost_stiha(trick) by who took it (:412-416) + final_reward_factor * next_max (:441),
 *           next_max = next_q at the seat's next decision (:417-418) or the final reward on the game's last trick (:439)
 *   meta_out [T/4,N,4] u8: bit 0 row valid (the seat played in a completed trick), bit 1 last transition of its
 *           game, bit 2 the seat's next decision lies beyond the rollout (next_max taken as 0).
 * Parity unpinned for the assembly (no reference fixture; Igralec.py cannot be imported), pinned for its inputs. */
int tarok_targets_ref(tarok_env *env, int T, const uint64_t *obs_before, const uint8_t *action, const uint16_t *trick,
                      const uint8_t *done, const int16_t *reward, const float *next_q, float final_reward_factor,
                      float *dy_out, uint8_t *meta_out, void *stream);

/* ---- The learner's update of the policy above as a few fused launches (build-owned, like the policy itself:
 * the reference's learner is a double-Q LSTM agent under pytorch-lightning, Igralec.py:545-714).  The network
 * 256-256-256-64 lives in ONE flat f32 vector of TAROK_MLP_PARAMS entries, in torch.nn.Linear layouts:
 * W1 [256,256] | b1 [256] | W2 [256,256] | b2 [256] | W3 [64,256] | b3 [64]  (offsets TAROK_MLP_*). */
#define TAROK_MLP_W1 0
#define TAROK_MLP_B1 65536
#define TAROK_MLP_W2 65792
#define TAROK_MLP_B2 131328
#define TAROK_MLP_W3 131584
#define TAROK_MLP_B3 147968
#define TAROK_MLP_PARAMS 148032
#define TAROK_LEARN_PAD 256 /* padding rows of the activation arrays of tarok_learn_chain / tarok_learn_dw */
#define TAROK_LEARN_MAX_BATCH 4194048 /* samples per minibatch of tarok_learn_dw (its arrays are addressed through 32-bit
                                       * byte offsets: (B + TAROK_LEARN_PAD) * 512 < 2^31); larger: TAROK_EINVAL */

/* Returns of a rollout of T lock-steps (rows [T,N] as written by tarok_policy_step): every card is credited with
 * its seat's final score of the game it belongs to, times reward_scale.
 *   rec_out [T,N,4] f32: {log-probability at play time, return, value at play time, bits: card | known << 8}
 *           (known = that game ended inside the rollout; other samples carry weight 0 in the update)
 *   stats_out [4] f32: {mean, 1 / std of the advantages (return - value) over the known samples, known fraction, 0}
 *   scratch [ceil(N/256),4] f32. */
int tarok_learn_returns(tarok_env *env, int T, const uint8_t *done, const int16_t *reward, const uint64_t *obs,
                        const float *logp, const float *value, const uint8_t *action, float reward_scale,
                        float *rec_out, float *stats_out, float *scratch, void *stream);

/* Forward, loss and backward chain of one minibatch of B samples (sample j = row index[j] of feature_words [M,4] /
 * rec [M,4]; index NULL: row j): feature gather + expansion -> layers 1-3 -> the loss of tarok_ppo_loss (advantage
 * = (return - value - stats[0]) * stats[1], weight = known) -> dH2, dH1.  Weights: the bf16 fragment-order copies
 * tarok_learn_adam writes (w3t / w2t: of the transposes), biases: pointers into the flat vector.
 *   Xw [B + TAROK_LEARN_PAD, 4] u64: the samples' feature words in minibatch order (layer 1's input for tarok_learn_dw);
 *   H1 / H2 / dH2 / dH1 [B + TAROK_LEARN_PAD, 256] bf16, dOut [B + TAROK_LEARN_PAD, 64] bf16 (rows B.. are padding that
 *   tarok_learn_dw may read and ignores; values unscaled: weight w, not w / sum w);
 *   scratch [ceil(B/96),4] f32; terms_out [4] f32 = {policy loss, value loss, entropy (weighted means),
 *   1 / max(sum w, 1)}; running [4] f32 or NULL: += {the three terms, 1}. */
int tarok_learn_chain(tarok_env *env, int64_t B, const uint64_t *feature_words, const int64_t *index, const float *rec,
                      const float *stats, float clip, float vf_coef, float ent_coef, const void *w1, const float *b1,
                      const void *w2, const float *b2, const void *w3, const float *b3, const void *w3t, const void *w2t,
                      uint64_t *Xw, void *H1, void *H2, void *dOut, void *dH2, void *dH1, float *scratch, float *terms_out,
                      float *running, void *stream);

/* The weight and bias gradients of that minibatch: grad_out [TAROK_MLP_PARAMS] f32 = terms[3] * (dH^T H per layer,
 * column sums of dH), in the flat parameter order.  workspace: tarok_learn_workspace_bytes(env) bytes.
 * B <= TAROK_LEARN_MAX_BATCH. */
int64_t tarok_learn_workspace_bytes(tarok_env *env);
int tarok_learn_dw(tarok_env *env, int64_t B, const uint64_t *Xw, const void *H1,
                   const void *H2, const void *dOut, const void *dH2, const void *dH1, const float *terms,
                   void *workspace, float *grad_out, void *stream);

/* Gradient-norm clip (max_norm <= 0: none) + Adam (torch.optim.Adam semantics) on the flat vectors, then the
 * kernels' bf16 fragment-order weight copies (any may be NULL).  step: device counter, incremented.
 * apply = 0: only rebuild the copies from param.  gnorm_out [1] f32 or NULL. */
int tarok_learn_adam(tarok_env *env, float *param, const float *grad, float *m, float *v, int32_t *step, float lr,
                     float beta1, float beta2, float eps, float max_norm, void *w1, void *w2, void *w3, void *w3t,
                     void *w2t, float *gnorm_out, int apply, void *stream);

/* Diagnostics: when `stamps` (device, [ceil(N/64), 3] u64) is non-NULL every wave of the step
 * kernels records {s_memrealtime at entry, at exit, shader cycles in between}.  NULL turns it off. */
int tarok_debug_stamps(tarok_env *env, uint64_t *stamps);
/* The same with the buffer's capacity in 64-bit words stated: a kernel writes its stamps only into a buffer that
 * holds all of them — the step kernels 3 per wave (3 * ceil(N/64)), tarok_policy_mlp and tarok_learn_chain 8 per
 * workgroup (8 * ceil(N/128), 8 * ceil(B/96)).  tarok_debug_stamps registers the step kernels' size. */
int tarok_debug_stamps_sized(tarok_env *env, uint64_t *stamps, int64_t n_words);
/* Self-test of the refill role as the step kernels compile it (tests only; it takes over the env's refill lists and
 * next-game lines and leaves them empty: tarok_reset before the env is used again).  Every group's lists are filled by
 * hand — every slot, episodes episode0+1 .. episode0+per_slot (1..14), in list order `order` (0 ascending slots, 1
 * descending, 2 scattered) — `reps` step launches of `kind` (0 tarok_step_random, 1 tarok_policy_random + tarok_step,
 * 2 tarok_krog_random of 4 cards, 3 of 2 cards) work them off, and every line is compared with a re-deal by a kernel
 * of its own.  report_out (host, 49 u64): [0] lines that differ, then {slot, episode, play word read / expected, seat
 * word read / expected} of the first eight. */
int tarok_debug_refill_selftest(tarok_env *env, int kind, int per_slot, uint32_t episode0, int order, int reps, uint64_t *report_out);

/* Canonical state for parity checks / checkpoints: lanes_out [10,N] u64. */
int tarok_get_state(tarok_env *env, uint64_t *lanes_out, void *stream);
/* Restore every game from canonical lanes (the inverse of tarok_get_state): checkpoint/resume,
 * or hand-built positions.  The RNG keys / episode numbers / next-game buffers are untouched. */
int tarok_set_state(tarok_env *env, const uint64_t *lanes_in, void *stream);
/* Per-slot bookkeeping: episode_out [N] u32 (current episode number),
 * score_sum_out [N,4] i32 (scores summed over the slot's finished games =
 * Tarok.rezultati per seat, Tarok.py:59-61).  Either may be NULL. */
int tarok_get_counters(tarok_env *env, uint32_t *episode_out, int32_t *score_sum_out, void *stream);

#ifdef __cplusplus
}
#endif
#endif
