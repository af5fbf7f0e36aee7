/* TEST INFRASTRUCTURE — see tarok_oracle.h.  Scalar CPU restatement of the
 * reference's rules; every rule cites the reference file:line it follows. */
#include "tarok_oracle.h"

#include <pthread.h>
#include <stdlib.h>
#include <string.h>

#define BIT(i) (1ULL << (i))
#define DECK ((1ULL << 54) - 1)
#define TAROK_MASK (((1ULL << 22) - 1) << 32)
#define PAGAT BIT(32)
/* Roka.vrednost_stiha per-card values (Roka.py:76-95): 5 for kings and
 * tarok 1/21/22, rank-3 for suit ranks 5..7, 1 otherwise. */
#define V5 (BIT(7) | BIT(15) | BIT(23) | BIT(31) | BIT(32) | BIT(52) | BIT(53))
#define V4 (BIT(6) | BIT(14) | BIT(22) | BIT(30))
#define V3 (BIT(5) | BIT(13) | BIT(21) | BIT(29))
#define V2 (BIT(4) | BIT(12) | BIT(20) | BIT(28))

static inline int popc(uint64_t m) { return __builtin_popcountll(m); }
static inline uint64_t suit_mask(int card) { return card >= 32 ? TAROK_MASK : (0xFFULL << (8 * (card >> 3))); }
static inline int is_klop_family(int c) { return c == TO_KLOP || c == TO_BERAC || c == TO_ODPRTI_BERAC; }
static const uint8_t GROUP_SIZE[10] = {0, 3, 2, 1, 3, 2, 1, 0, 1, 0}; /* Navadna_igra.py:36-44 */
static const uint8_t N_DISCARD[10] = {0, 3, 2, 1, 3, 2, 1, 0, 0, 0};  /* Navadna_igra.py:48-57 */

/* Karta.vrednost (Karta.py:10-16): NOT the scoring value; used only by the
 * discard filter.  Applies "st > 4 -> st-3" to taroks as well. */
int to_vrednost(int card) {
    int tarok = card >= 32;
    int st = tarok ? card - 31 : (card & 7) + 1;
    if (tarok && (st == 1 || st == 21 || st == 22)) return 5;
    if (st > 4) return st - 3;
    return 1;
}

/* Roka.mozno_zalozit (Roka.py:23-27): cards with vrednost() < 5 */
uint64_t to_discardable(uint64_t hand) {
    uint64_t m = 0;
    for (int c = 0; c < 54; c++)
        if ((hand >> c) & 1 && to_vrednost(c) < 5) m |= BIT(c);
    return m;
}

/* Roka.prestej = sum over tri_po_tri groups of vrednost_stiha (Roka.py:56-98).
 * Order independent: sum(val) - 2*floor(n/3) - [n%3 != 0]. */
int to_prestej(uint64_t pile) {
    pile &= DECK;
    int n = popc(pile);
    int v = n + 4 * popc(pile & V5) + 3 * popc(pile & V4) + 2 * popc(pile & V3) + popc(pile & V2);
    return v - 2 * (n / 3) - (n % 3 != 0);
}

/* Roka.vrednost_stiha (Roka.py:76-95) of one trick given as a mask of n_cards cards */
int to_vrednost_stiha(uint64_t stih, int n_cards) {
    int v = popc(stih) + 4 * popc(stih & V5) + 3 * popc(stih & V4) + 2 * popc(stih & V3) + popc(stih & V2);
    return (n_cards == 1 || n_cards == 2) ? v - 1 : v - 2;
}

/* Navadna_igra.mozne_karte (Navadna_igra.py:158-168) */
uint64_t to_legal_navadna(uint64_t hand, int lead) {
    if (lead >= 0) {
        uint64_t s = hand & suit_mask(lead);
        if (s) return s;
        s = hand & TAROK_MASK;
        if (s) return s;
    }
    return hand;
}

/* Klop.mozne_karte (Klop.py:96-133), also used by Berac.  The over-play
 * filter (Klop.py:102,116) only gates which branch removes the pagat; every
 * branch ends as "the follow-suit set, minus pagat unless that empties it". */
uint64_t to_legal_klop(uint64_t hand, int lead) {
    uint64_t b = to_legal_navadna(hand, lead);
    uint64_t nb = b & ~PAGAT;
    return nb ? nb : b;
}

/* pobere_stih / primerjaj_karti (Klop.py:81-94 == Navadna_igra.py:143-156) */
int to_trick_winner(const uint8_t c[4]) {
    int w = 0;
    for (int i = 1; i < 4; i++) {
        int sw = c[w] >= 32 ? 4 : c[w] >> 3, si = c[i] >= 32 ? 4 : c[i] >> 3;
        if (sw == si ? c[w] < c[i] : si == 4) w = i;
    }
    return w;
}

/* Igra.razdeli (Igra.py:65-73) + engine constructors (Igra.py:38-55,
 * Navadna_igra.py:20-30, Berac.py:13-15) */
void to_new_game(to_game *g, const uint8_t perm[54], int contract, int declarer, int king) {
    memset(g, 0, sizeof *g);
    for (int s = 0; s < 4; s++)
        for (int i = 0; i < 12; i++) g->hand[s] |= BIT(perm[12 * s + i]);
    for (int i = 0; i < 6; i++) g->talon[i] = perm[48 + i];
    g->contract = (uint8_t)contract;
    g->declarer = (uint8_t)declarer;
    g->king = (contract >= TO_TRI && contract <= TO_ENA) ? (int8_t)king : -1;
    g->choice = -1;
    /* first leader: seat 0 (Klop.py:26, Navadna_igra.py:70); declarer for Berac (Berac.py:15) */
    g->leader = (contract == TO_BERAC || contract == TO_ODPRTI_BERAC) ? (uint8_t)declarer : 0;
    if (contract == TO_KLOP) {
        g->team = 0;
        g->talon_left = 6;
    } else {
        g->team = (uint8_t)(1u << declarer);
        if (g->king >= 0) /* partner = holder of the called king in the PRE-exchange hands, Navadna_igra.py:24 */
            for (int s = 0; s < 4; s++)
                if ((g->hand[s] >> (g->king * 8 + 7)) & 1) g->team |= (uint8_t)(1u << s);
    }
    g->phase = N_DISCARD[contract] ? TO_PHASE_EXCHANGE : TO_PHASE_PLAY;
}

/* odpri_talon + menjaj_iz_talona (Navadna_igra.py:36-66; player side Igralec.py:161-171) */
int to_exchange(to_game *g, int choice, const uint8_t *discards) {
    if (g->phase != TO_PHASE_EXCHANGE) return -2;
    int gs = GROUP_SIZE[g->contract], nd = N_DISCARD[g->contract];
    if (choice < 0 || choice >= 6 / gs) { g->error = 1; return -1; }
    uint64_t grp = 0, dm = 0;
    for (int i = 0; i < gs; i++) grp |= BIT(g->talon[choice * gs + i]);
    uint64_t h = g->hand[g->declarer] | grp;
    for (int i = 0; i < nd; i++) {
        if (discards[i] >= 54 || !((h >> discards[i]) & 1) || ((dm >> discards[i]) & 1)) { g->error = 1; return -1; }
        dm |= BIT(discards[i]);
    }
    g->hand[g->declarer] = h & ~dm;
    g->pile[g->declarer] |= dm;
    g->choice = (int8_t)choice;
    g->phase = TO_PHASE_PLAY;
    return 0;
}

int to_seat(const to_game *g) { return (g->leader + g->n_in_trick) & 3; }

uint64_t to_legal(const to_game *g) {
    if (g->phase != TO_PHASE_PLAY) return 0;
    uint64_t h = g->hand[to_seat(g)];
    int lead = g->n_in_trick ? g->trick[0] : -1;
    return is_klop_family(g->contract) ? to_legal_klop(h, lead) : to_legal_navadna(h, lead);
}

static uint64_t rest_talon(const to_game *g) {
    uint64_t all = 0, grp = 0;
    int gs = GROUP_SIZE[g->contract];
    for (int i = 0; i < 6; i++) all |= BIT(g->talon[i]);
    if (g->choice >= 0)
        for (int i = 0; i < gs; i++) grp |= BIT(g->talon[g->choice * gs + i]);
    return all & ~grp;
}

static void score_klop(to_game *g) { /* Klop.py:36-45 */
    int c[4], over = 0;
    for (int s = 0; s < 4; s++) { c[s] = to_prestej(g->pile[s]); over |= c[s] > 35; }
    /* any player above 35: the `< -35` re-test at Klop.py:39 is on the positive
     * count and never true, so everybody gets 0 */
    for (int s = 0; s < 4; s++) g->score[s] = over ? 0 : (int16_t)-c[s];
}

static void score_navadna(to_game *g) { /* Navadna_igra.py:80-113 */
    uint64_t t = 0;
    for (int s = 0; s < 4; s++)
        if ((g->team >> s) & 1) t |= g->pile[s];
    if (g->contract != TO_SOLO_BREZ && popc(g->team) == 1 && g->king >= 0 &&
        ((g->pile[g->declarer] >> (g->king * 8 + 7)) & 1))
        t |= rest_talon(g);                       /* Navadna_igra.py:87-92 */
    int v = to_prestej(t), d = v - 35, ad = d < 0 ? -d : d;
    int r = 5 * ((ad + 2) / 5);                   /* int(round(d/5))*5, Navadna_igra.py:103 */
    if (d < 0) r = -r;
    int c = 10 * g->contract;
    for (int s = 0; s < 4; s++)
        g->score[s] = ((g->team >> s) & 1) ? (int16_t)((v > 35 ? c : -c) + r) : 0;
}

/* krog (Klop.py:47-79, Navadna_igra.py:115-141) + the per-contract start()
 * loops (Klop.py:22-45, Berac.py:13-44, Navadna_igra.py:70-113) */
int to_step(to_game *g, int action) {
    g->last_trick = 0;
    if (g->phase != TO_PHASE_PLAY) return -2;
    uint64_t legal = to_legal(g);
    if (action < 0 || action >= 54 || !((legal >> action) & 1)) { g->error = 1; return -1; }
    int seat = to_seat(g);
    g->hand[seat] &= ~BIT(action);
    g->trick[g->n_in_trick++] = (uint8_t)action;
    if (g->n_in_trick < 4) return 0;
    int ws = (g->leader + to_trick_winner(g->trick)) & 3;
    uint64_t tm = BIT(g->trick[0]) | BIT(g->trick[1]) | BIT(g->trick[2]) | BIT(g->trick[3]);
    if (g->contract == TO_KLOP && g->talon_left > 0) tm |= BIT(g->talon[--g->talon_left]); /* Klop.py:67-71 */
    g->pile[ws] |= tm;
    g->last_trick = (uint16_t)(0x8000 | (to_vrednost_stiha(tm, popc(tm)) << 4) | ws);   /* what rezultat_stiha sees */
    g->leader = (uint8_t)ws;
    g->n_in_trick = 0;
    g->trick_no++;
    memset(g->trick, 0, 4);
    if (g->contract == TO_BERAC || g->contract == TO_ODPRTI_BERAC) {
        int v = g->contract == TO_BERAC ? 70 : 90;
        if (ws == g->declarer) { g->score[ws] = (int16_t)-v; g->phase = TO_PHASE_DONE; return 1; } /* Berac.py:33-39 */
        if (g->trick_no == 12) { g->score[g->declarer] = (int16_t)v; g->phase = TO_PHASE_DONE; return 1; }
        return 0;
    }
    if (g->trick_no < 12) return 0;
    if (g->contract == TO_KLOP) score_klop(g); else score_navadna(g);
    g->phase = TO_PHASE_DONE;
    return 1;
}

/* canonical state lanes (tarok_env.h tarok_get_state): H0-3, P0-3, TAL, META */
void to_export_lanes(const to_game *g, uint64_t l[10]) {
    for (int s = 0; s < 4; s++) { l[s] = g->hand[s]; l[4 + s] = g->pile[s]; }
    uint64_t tal = 0, m = 0;
    for (int i = 0; i < 6; i++) tal |= (uint64_t)g->talon[i] << (6 * i);
    l[8] = tal;
    for (int i = 0; i < g->n_in_trick; i++) m |= (uint64_t)g->trick[i] << (6 * i);
    m |= (uint64_t)g->n_in_trick << 24;
    m |= (uint64_t)g->leader << 27;
    m |= (uint64_t)g->trick_no << 29;
    m |= (uint64_t)g->contract << 33;
    m |= (uint64_t)g->declarer << 37;
    m |= (uint64_t)(g->king < 0 ? 7 : g->king) << 39;
    m |= (uint64_t)g->team << 42;
    m |= (uint64_t)g->talon_left << 46;
    m |= (uint64_t)(g->choice < 0 ? 7 : g->choice) << 49;
    m |= (uint64_t)g->phase << 52;
    m |= (uint64_t)g->error << 54;
    l[9] = m;
}

/* ------------------------------------------------------------------------
 * synthetic inputs — the build's own spec (oracle/tarok_spec.py)
 * ---------------------------------------------------------------------- */
static inline uint64_t mix64(uint64_t x) {
    x ^= x >> 30; x *= 0xBF58476D1CE4E5B9ULL;
    x ^= x >> 27; x *= 0x94D049BB133111EBULL;
    x ^= x >> 31;
    return x;
}

uint64_t to_game_key(uint64_t seed, uint64_t gidx, uint64_t episode) {
    uint64_t a = gidx * 0x9E3779B97F4A7C15ULL + episode * 0xD1B54A32D192ED03ULL + 0x2545F4914F6CDD1DULL;
    return mix64(seed ^ mix64(a));
}

uint32_t to_rng32(uint64_t key, uint32_t i) {
    uint32_t lo = (uint32_t)key, hi = (uint32_t)(key >> 32);
    uint32_t x = lo ^ (i * 0x9E3779B1u);
    x ^= x >> 16; x *= 0x85EBCA6Bu;
    x ^= x >> 13; x *= 0xC2B2AE35u;
    x ^= x >> 16; x ^= hi;
    x *= 0x27D4EB2Fu; x ^= x >> 15;
    return x;
}

static inline uint32_t pick(uint32_t r, uint32_t n) { return (uint32_t)(((uint64_t)r * n) >> 32); }

static inline int kth_bit(uint64_t m, int k) {
    while (k--) m &= m - 1;
    return __builtin_ctzll(m);
}

static int cmp_u32(const void *a, const void *b) {
    uint32_t x = *(const uint32_t *)a, y = *(const uint32_t *)b;
    return x < y ? -1 : x > y;
}

void to_deal_perm(uint64_t key, uint8_t perm[54]) {
    uint32_t k[54];
    for (uint32_t c = 0; c < 54; c++) k[c] = (to_rng32(key, c) & 0xFFFFFFC0u) | c;
    qsort(k, 54, sizeof k[0], cmp_u32);
    for (int i = 0; i < 54; i++) perm[i] = (uint8_t)(k[i] & 63);
}

/* One bidding round between four Bot players: Igra.licitacija (Igra.py:75-114) with
 * Bot_igralec.licitiram (Igralec.py:148-152) behind the base filter (Igralec.py:58-74).
 * Bids are int(Tip_igre): Naprej -10, Klop 0, Tri 10, Dve 20, Ena 30. */
typedef struct { uint64_t key; uint32_t calls; } bid_ctx;

static int bot_ask(bid_ctx *b, int min_igra, int obvezno /* -100 = None */, int prednost) {
    uint32_t w = pick(to_rng32(b->key, 72 + b->calls++), 6);
    int wish = w < 3 ? -10 : (int)(w - 2) * 10;               /* Igralec.py:151 */
    int ok = prednost ? wish >= min_igra : wish > min_igra;   /* Igralec.py:58-74 */
    return ok ? wish : (obvezno == -100 ? -10 : obvezno);
}

void to_bot_bidding(uint64_t key, int *contract, int *declarer) {
    bid_ctx b = {key, 0};
    unsigned still = 0;
    int top = 10;
    for (int seat = 1; seat <= 3; seat++) {                   /* Igra.py:83-87 */
        int v = bot_ask(&b, top, -100, 0);
        if (v != -10) still |= 1u << seat;
        if (v > top) top = v;
    }
    if (top == 10) {                                          /* Igra.py:89-91 */
        *declarer = 0;
        *contract = bot_ask(&b, -10, 0, 0) / 10;
        return;
    }
    int v = bot_ask(&b, top, -100, 1);                        /* Igra.py:93-96 */
    if (v != -10) still |= 1u;
    if (v > top) top = v;
    int holder = __builtin_ctz(still);                        /* min(lic), Igra.py:97 */
    static const int order[4] = {1, 2, 3, 0};                 /* seat 0 is asked last, Igra.py:102-103 */
    for (int rounds = 0; popc(still) != 1 && rounds < 8; rounds++) {
        unsigned nxt = 0;
        for (int k = 0; k < 4; k++) {
            int seat = order[k];
            if (!((still >> seat) & 1)) continue;
            v = bot_ask(&b, top, seat == holder ? top : -100, 0);
            if (v != -10) { nxt |= 1u << seat; holder = seat; top = v; }
        }
        still = nxt;
    }
    *declarer = holder;
    *contract = top / 10;
}

void to_sample_setup(uint64_t key, int mix, int *contract, int *declarer, int *king) {
    static const int nav7[7] = {TO_TRI, TO_DVE, TO_ENA, TO_SOLO_TRI, TO_SOLO_DVE, TO_SOLO_ENA, TO_SOLO_BREZ};
    int c;
    if (mix == TO_MIX_BOT) {
        to_bot_bidding(key, contract, declarer);
        *king = (*contract >= TO_TRI && *contract <= TO_ENA) ? (int)pick(to_rng32(key, 67), 4) : -1;
        return;
    }
    if (mix >= TO_MIX_FIXED) c = mix - TO_MIX_FIXED;
    else if (mix == TO_MIX_NAVADNA3) c = TO_TRI + (int)pick(to_rng32(key, 65), 3);
    else {
        uint32_t fam = pick(to_rng32(key, 64), 3), r = to_rng32(key, 65);
        if (fam == 0) c = TO_KLOP;
        else if (fam == 1) c = pick(r, 2) == 0 ? TO_BERAC : TO_ODPRTI_BERAC;
        else c = nav7[pick(r, 7)];
    }
    *contract = c;
    *declarer = c == TO_KLOP ? 0 : (int)pick(to_rng32(key, 66), 4);
    *king = (c >= TO_TRI && c <= TO_ENA) ? (int)pick(to_rng32(key, 67), 4) : -1;
}

void to_bot_discards(uint64_t key, uint64_t hand, int n, uint8_t out[3]) {
    uint64_t cand = to_discardable(hand);
    if (popc(cand) < n) cand = hand;
    for (int j = 0; j < n; j++) {
        int c = kth_bit(cand, (int)pick(to_rng32(key, 68 + (uint32_t)j), (uint32_t)popc(cand)));
        out[j] = (uint8_t)c;
        cand &= ~BIT(c);
    }
}

int to_policy_action(uint64_t key, int step, uint64_t mask) {
    return kth_bit(mask, (int)pick(to_rng32(key, 128 + (uint32_t)step), (uint32_t)popc(mask)));
}

void to_synth_game(to_game *g, uint64_t seed, uint64_t gidx, uint64_t episode, int mix) {
    uint64_t key = to_game_key(seed, gidx, episode);
    uint8_t perm[54], disc[3] = {255, 255, 255};
    int c, d, k;
    to_deal_perm(key, perm);
    to_sample_setup(key, mix, &c, &d, &k);
    to_new_game(g, perm, c, d, k);
    if (g->phase == TO_PHASE_EXCHANGE) {
        uint64_t grp = 0;
        for (int i = 0; i < GROUP_SIZE[c]; i++) grp |= BIT(g->talon[i]);
        to_bot_discards(key, g->hand[d] | grp, N_DISCARD[c], disc);
        to_exchange(g, 0, disc);
    }
}

int64_t to_rollout(uint64_t seed, uint64_t gidx0, int64_t n, uint64_t episode, int mix,
                   int16_t *nsteps, int8_t *seats, uint64_t *masks, uint8_t *actions, int16_t *scores) {
    int64_t total = 0;
    for (int64_t i = 0; i < n; i++) {
        to_game g;
        uint64_t key = to_game_key(seed, gidx0 + (uint64_t)i, episode);
        to_synth_game(&g, seed, gidx0 + (uint64_t)i, episode, mix);
        if (seats) memset(seats + i * 48, -1, 48);
        if (masks) memset(masks + i * 48, 0, 48 * sizeof(uint64_t));
        if (actions) memset(actions + i * 48, 255, 48);
        int t = 0;
        while (g.phase == TO_PHASE_PLAY) {
            uint64_t m = to_legal(&g);
            int a = to_policy_action(key, t, m);
            if (seats) seats[i * 48 + t] = (int8_t)to_seat(&g);
            if (masks) masks[i * 48 + t] = m;
            if (actions) actions[i * 48 + t] = (uint8_t)a;
            to_step(&g, a);
            t++;
        }
        if (nsteps) nsteps[i] = (int16_t)t;
        if (scores) memcpy(scores + i * 4, g.score, 4 * sizeof(int16_t));
        total += t;
    }
    return total;
}

typedef struct {
    uint64_t seed, gidx0, episode; int64_t n; int mix;
    int16_t *nsteps; int8_t *seats; uint64_t *masks; uint8_t *actions; int16_t *scores;
    int64_t total;
} mt_job;

static void *mt_run(void *p) {
    mt_job *j = (mt_job *)p;
    j->total = to_rollout(j->seed, j->gidx0, j->n, j->episode, j->mix, j->nsteps, j->seats, j->masks, j->actions, j->scores);
    return NULL;
}

int64_t to_rollout_mt(int threads, uint64_t seed, uint64_t gidx0, int64_t n, uint64_t episode, int mix,
                      int16_t *nsteps, int8_t *seats, uint64_t *masks, uint8_t *actions, int16_t *scores) {
    if (threads < 1) threads = 1;
    if (threads > 256) threads = 256;
    pthread_t th[256];
    mt_job jobs[256];
    int64_t per = (n + threads - 1) / threads, total = 0;
    int used = 0;
    for (int t = 0; t < threads; t++) {
        int64_t lo = t * per, hi = lo + per > n ? n : lo + per;
        if (lo >= hi) break;
        mt_job *j = &jobs[used];
        j->seed = seed; j->gidx0 = gidx0 + (uint64_t)lo; j->episode = episode; j->n = hi - lo; j->mix = mix;
        j->nsteps = nsteps ? nsteps + lo : NULL;
        j->seats = seats ? seats + lo * 48 : NULL;
        j->masks = masks ? masks + lo * 48 : NULL;
        j->actions = actions ? actions + lo * 48 : NULL;
        j->scores = scores ? scores + lo * 4 : NULL;
        j->total = 0;
        pthread_create(&th[used], NULL, mt_run, j);
        used++;
    }
    for (int t = 0; t < used; t++) { pthread_join(th[t], NULL); total += jobs[t].total; }
    return total;
}

/* observation word of include/tarok_env.h (TAROK_OBS_*) for the seat to move */
uint64_t to_obs_word(const to_game *g, int finished_now) {
    uint64_t o = to_legal(g);
    o |= (uint64_t)to_seat(g) << 54;
    o |= (uint64_t)(g->trick_no * 4 + g->n_in_trick) << 56;
    if (finished_now || g->phase == TO_PHASE_DONE) o |= 1ULL << 62;
    o |= (uint64_t)g->error << 63;
    return o;
}

/* n_steps lock-steps of the random policy with auto-reset (a finished game is
 * replaced at once by episode+1 of the same slot), the CPU statement of
 * tarok_run_random(..., TAROK_AUTO_RESET).  lanes_out is lane-major [10][n]. */
int64_t to_run_autoreset(uint64_t seed, uint64_t gidx0, int64_t n, int mix, uint32_t episode0, int64_t n_steps,
                         uint32_t *episode_out, int32_t *score_sum_out, uint64_t *lanes_out, uint64_t *obs_out) {
    int64_t total = 0;
    for (int64_t i = 0; i < n; i++) {
        to_game g;
        uint32_t ep = episode0;
        int32_t sum[4] = {0, 0, 0, 0};
        int fin = 0;
        to_synth_game(&g, seed, gidx0 + (uint64_t)i, ep, mix);
        uint64_t key = to_game_key(seed, gidx0 + (uint64_t)i, ep);
        for (int64_t t = 0; t < n_steps; t++) {
            int step = g.trick_no * 4 + g.n_in_trick;
            int a = to_policy_action(key, step, to_legal(&g));
            fin = to_step(&g, a) == 1;
            total++;
            if (fin) {
                for (int s = 0; s < 4; s++) sum[s] += g.score[s];
                ep++;
                to_synth_game(&g, seed, gidx0 + (uint64_t)i, ep, mix);
                key = to_game_key(seed, gidx0 + (uint64_t)i, ep);
            }
        }
        if (episode_out) episode_out[i] = ep;
        if (score_sum_out) for (int s = 0; s < 4; s++) score_sum_out[i * 4 + s] = sum[s];
        if (lanes_out) {
            uint64_t l[10];
            to_export_lanes(&g, l);
            for (int k = 0; k < 10; k++) lanes_out[(int64_t)k * n + i] = l[k];
        }
        if (obs_out) obs_out[i] = to_obs_word(&g, fin);
    }
    return total;
}
