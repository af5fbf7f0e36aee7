// The body of k_rollout (tarok_amd/csrc/tarok_env.hip) run on the CPU over the DEVICE rule code
// (tarok_amd/csrc/tarok_device.h compiled by g++ with the gfx950 builtins emulated): deal, setup,
// Bot exchange, then up to 48 cards of random play per game.  Writes seats / masks / actions /
// scores / nsteps for tests/test_device_rules_host.py to compare with the CPU oracle.
#include "hip/hip_runtime.h"
#include "../../tarok_amd/csrc/tarok_device.h"
#include <stdio.h>
#include <stdlib.h>

int main(int argc, char **argv) {
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
        for (int t = 0; t < 48; t++) {
            bool live = g.phase == TK_PHASE_PLAY;
            masks[t] = 0; seats[t] = -1; actions[t] = 255;
            if (live) {
                u64 m = legal_now(g);
                masks[t] = m;
                seats[t] = (int8_t)((g.leader + g.nt) & 3);
                u32 a = policy_action(key, (u32)t, m);
                actions[t] = (uint8_t)a;
                u32 ti;
                apply_step<true>(g, a, scores, ti);
                played++;
            }
        }
        // pack / unpack round trip of the final state must be lossless
        u64 x0, x1, y0, y1;
        pack(g, x0, x1, y0, y1);
        Game r;
        unpack(r, x0, x1, y0, y1);
        if (r.A != g.A || r.B != g.B || r.C != g.C || r.talon != g.talon || r.trick != g.trick || r.nt != g.nt ||
            r.leader != g.leader || r.trick_no != g.trick_no || r.phase != g.phase || r.contract != g.contract ||
            r.declarer != g.declarer || r.king != g.king || r.team != g.team || r.tl != g.tl || r.error != g.error) {
            fprintf(stderr, "pack/unpack mismatch at game %ld\n", i);
            return 4;
        }
        fwrite(seats, 1, 48, f); fwrite(masks, 8, 48, f); fwrite(actions, 1, 48, f);
        fwrite(&scores, 8, 1, f); fwrite(&played, 2, 1, f);
    }
    fclose(f);
    return 0;
}
