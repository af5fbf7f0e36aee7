// The body of k_rollout (tarok_amd/csrc/tarok_env.hip) run on the CPU over the DEVICE rule code
// (tarok_amd/csrc/tarok_device.h compiled by g++ with the gfx950 builtins emulated): deal, setup,
// Bot exchange, then up to 48 cards of random play per game.  Writes seats / masks / actions /
// scores / nsteps for tests/test_device_rules_host.py to compare with the CPU oracle.
#include "hip/hip_runtime.h"
#include "../../tarok_amd/csrc/tarok_device.h"
#include <stdio.h>
#include <stdlib.h>

// Replay mode: recorded games of the reference engine (tests/golden/traces_v1.npz, flattened by the
// test into records {deal[54], contract, declarer, king, choice, discards[3], nsteps, actions[48]}):
// explicit deal -> setup_game -> apply_exchange -> apply_step<false> card by card; writes seats,
// legal masks, per-trick info, return codes and final scores for comparison with the fixture.
static int replay(const char *in, const char *out) {
    FILE *fi = fopen(in, "rb"), *fo = fopen(out, "wb");
    if (!fi || !fo) return 3;
    struct Rec { uint8_t deal[54]; int8_t contract, declarer, king, choice; uint8_t discards[3]; uint8_t nsteps; uint8_t actions[48]; } rec;
    static_assert(sizeof(Rec) == 110, "packed record");
    while (fread(&rec, sizeof rec, 1, fi) == 1) {
        u64 h[4] = {0, 0, 0, 0}, tal = 0;
        for (int k = 0; k < 48; k++) h[k / 12] |= 1ULL << rec.deal[k];
        for (int k = 0; k < 6; k++) tal |= (u64)rec.deal[48 + k] << (6 * k);
        Game g;
        setup_game(g, h[0], h[1], h[2], h[3], tal, (u32)rec.contract, (u32)rec.declarer, rec.king < 0 ? 0u : (u32)rec.king);
        g.epar = 0; g.cprev = 0;
        int8_t ok = 1;
        if (g.phase == TK_PHASE_EXCHANGE) ok = apply_exchange(g, (u32)rec.choice, rec.discards[0], rec.discards[1], rec.discards[2]) ? 1 : 0;
        int8_t seats[48]; u64 masks[48]; uint16_t tinfo[48]; int8_t rc[48];
        u64 scores = 0;
        for (int t = 0; t < 48; t++) {
            seats[t] = -1; masks[t] = 0; tinfo[t] = 0; rc[t] = -2;
            if (t < rec.nsteps && g.phase == TK_PHASE_PLAY) {
                seats[t] = (int8_t)((g.leader + g.nt) & 3);
                masks[t] = legal_now(g);
                u32 ti = 0;
                rc[t] = (int8_t)apply_step<false>(g, rec.actions[t], scores, ti);
                tinfo[t] = (uint16_t)ti;
            }
        }
        int8_t done = g.phase == TK_PHASE_DONE;
        fwrite(&ok, 1, 1, fo); fwrite(&done, 1, 1, fo); fwrite(seats, 1, 48, fo); fwrite(masks, 8, 48, fo);
        fwrite(tinfo, 2, 48, fo); fwrite(rc, 1, 48, fo); fwrite(&scores, 8, 1, fo);
    }
    fclose(fi); fclose(fo);
    return 0;
}

// direct checks of the branch-free helpers against the obvious loops
static int self_check() {
    const u32 edge[] = {0u, 1u, 0x80u, 0x8000u, 0x80000000u, 0xFF000000u, 0x7FFFFFFFu, 0xFFFFFFFFu, 0x01010101u};
    for (u32 x : edge)
        if (tk_zero_mask(x) != (x == 0 ? ~0u : 0u)) { fprintf(stderr, "tk_zero_mask(%08x)\n", x); return 6; }
    u64 r = 0x9E3779B97F4A7C15ULL;
    for (int it = 0; it < 200000; it++) {
        r ^= r << 13; r ^= r >> 7; r ^= r << 17;
        u64 m = r & TK_DECK;
        if (it & 1) { u64 r2 = r * 0xD1B54A32D192ED03ULL; m &= r2 | (r2 >> 17); m &= (r2 >> 29) | (r2 << 11); }   // sparse, like a hand
        if (it % 1000 == 0) m = TK_DECK;
        if (it % 1000 == 1) m = 1ULL << (it % 54);
        int k = 0;
        for (u32 c = 0; c < 54; c++)
            if ((m >> c) & 1) {
                if (kth_bit(m, (u32)k) != c) { fprintf(stderr, "kth_bit(%016llx, %d)\n", (unsigned long long)m, k); return 7; }
                k++;
            }
        if (tk_zero_mask((u32)m) != ((u32)m == 0 ? ~0u : 0u)) return 6;
    }
    return 0;
}

int main(int argc, char **argv) {
    if (int rc = self_check()) return rc;
    if (argc == 4 && argv[1][0] == 'r') return replay(argv[2], argv[3]);
    if (argc < 7) { fprintf(stderr, "usage: %s seed offset n episode mix out.bin\n", argv[0]); return 2; }
    u64 seed = strtoull(argv[1], 0, 10), offset = strtoull(argv[2], 0, 10);
    long n = atol(argv[3]);
    u32 episode = (u32)atol(argv[4]);
    int mix = atoi(argv[5]);
    FILE *f = fopen(argv[6], "wb");
    if (!f) return 3;
    for (long i = 0; i < n; i++) {
        u64 key = game_key(seed, offset + (u64)i, episode);
        u64 h0, h1, h2, h3, tal;
        deal_thread(key, h0, h1, h2, h3, tal);
        u32 c, d, k;
        sample_setup(key, mix, c, d, k);
        Game g;
        setup_game(g, h0, h1, h2, h3, tal, c, d, k);
        g.epar = 0; g.cprev = 0;
        if (g.phase == TK_PHASE_EXCHANGE) bot_exchange(g, key);
        u64 scores = 0, masks[48];
        int8_t seats[48];
        uint8_t actions[48];
        int16_t played = 0;
        Game gd = g;                     // the same game through the deferred-scoring form the multi-card kernel uses
        Game gl = g;                     // ... and through the trick-aligned loops' form: the trick's cards from the C plane's gain
        u64 c_lead = 0;
        {   // a line of the dealt-ahead buffer is unpacked as a fresh game: the same as the general unpack
            Game fa, fb; u64 x0, x1, y0, y1;
            Game src = g; src.epar = (u32)(i % 14);
            pack(src, x0, x1, y0, y1);
            unpack(fa, x0, x1, y0, y1); unpack_fresh(fb, x0, x1, y0, y1);
            if (fa.A != fb.A || fa.B != fb.B || fa.C != fb.C || fa.talon != fb.talon || fa.trick != fb.trick || fa.nt != fb.nt ||
                fa.leader != fb.leader || fa.trick_no != fb.trick_no || fa.phase != fb.phase || fa.contract != fb.contract ||
                fa.declarer != fb.declarer || fa.king != fb.king || fa.team != fb.team || fa.tl != fb.tl || fa.error != fb.error ||
                fa.epar != fb.epar || fa.cprev != fb.cprev) {
                fprintf(stderr, "unpack_fresh differs at game %ld\n", i);
                return 8;
            }
        }
        for (int t = 0; t < 48; t++) {
            bool live = g.phase == TK_PHASE_PLAY;
            masks[t] = 0; seats[t] = -1; actions[t] = 255;
            if (live) {
                u64 m = legal_now(g);
                masks[t] = m;
                seats[t] = (int8_t)((g.leader + g.nt) & 3);
                u32 a = policy_action(key, (u32)t, m);
                actions[t] = (uint8_t)a;
                u32 ti = 0, ti2 = 0;
                u64 untouched = 0x1234;
                int r1 = apply_step<true>(g, a, scores, ti);
                int r2 = apply_step<true, true>(gd, a, untouched, ti2, false);
                if (gl.nt == 0) c_lead = gl.C;
                u64 sl = 0; u32 ti3 = 0;
                int r3 = apply_step<true>(gl, a, sl, ti3, true, &c_lead);
                if (r3 != r1 || ti3 != ti || gl.A != g.A || gl.B != g.B || gl.C != g.C || gl.tl != g.tl || gl.leader != g.leader ||
                    gl.trick_no != g.trick_no || gl.phase != g.phase || (r1 == 1 && sl != scores)) {
                    fprintf(stderr, "lead-plane form differs at game %ld card %d\n", i, t);
                    return 9;
                }
                if (r1 != r2 || untouched != 0x1234 || ti2 != 0 || gd.phase != g.phase || (r1 == 1 && final_scores(gd) != scores)) {
                    fprintf(stderr, "deferred scoring differs at game %ld card %d\n", i, t);
                    return 5;
                }
                played++;
            }
        }
        // pack / unpack round trip of the final state must be lossless (the line bookkeeping fields included)
        g.epar = (u32)(i % 14); g.cprev = (u32)(i % 15);
        u64 x0, x1, y0, y1;
        pack(g, x0, x1, y0, y1);
        Game r;
        unpack(r, x0, x1, y0, y1);
        if (r.A != g.A || r.B != g.B || r.C != g.C || r.talon != g.talon || r.trick != g.trick || r.nt != g.nt ||
            r.leader != g.leader || r.trick_no != g.trick_no || r.phase != g.phase || r.contract != g.contract ||
            r.declarer != g.declarer || r.king != g.king || r.team != g.team || r.tl != g.tl || r.error != g.error ||
            r.epar != g.epar || r.cprev != g.cprev) {
            fprintf(stderr, "pack/unpack mismatch at game %ld\n", i);
            return 4;
        }
        fwrite(seats, 1, 48, f); fwrite(masks, 8, 48, f); fwrite(actions, 1, 48, f);
        fwrite(&scores, 8, 1, f); fwrite(&played, 2, 1, f);
    }
    fclose(f);
    return 0;
}
