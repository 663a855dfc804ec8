// TEST INFRASTRUCTURE: the host-side launch planning of libgpmi355x.so (gaussian_process_amd/csrc/gpmi_plan.h, no HIP
// dependency) against brute force, built with g++ -fsanitize=address,undefined by tests/test_sanitize_cpu.py.
//   * every plan enumerates each live tile exactly once and no dead one (rectangles, lower triangles with any diagonal
//     offset, row maps with and without a host copy, ragged Tm / Tn, every supertile edge, M up to 131072);
//   * a staircase with more supertile rows than the table holds falls back to the rectangle -- it never overruns;
//   * block-width schedules cover the columns exactly; the flop counts agree with element-by-element sums.
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <random>
#include <vector>

#include "gpmi_plan.h"

using namespace gpmi;

static int g_fail = 0;
#define CHECK(cond, ...) do { if (!(cond)) { ++g_fail; fprintf(stderr, "FAIL %s:%d: ", __FILE__, __LINE__); fprintf(stderr, __VA_ARGS__); fprintf(stderr, "\n"); if (g_fail > 20) exit(1); } } while (0)

static long check_plan(int64_t Tm, int64_t Tn, int lower, int64_t diag_off, const std::vector<int32_t>* map, bool host_copy,
                       int rbt, int force_S) {
    TilePlan p;
    // canaries around the prefix table would need a wrapper struct; ASan's stack redzones do that job for `p`
    const bool ok = plan_tiles(p, Tm, Tn, lower, diag_off, map != nullptr, (map && host_copy) ? map->data() : nullptr,
                               (map && host_copy) ? (int)map->size() : 0, rbt, force_S);
    CHECK(ok, "plan_tiles refused Tm=%ld Tn=%ld", (long)Tm, (long)Tn);
    if (!ok) return 0;
    CHECK(p.S == 1 || p.S == 2 || p.S == 4 || p.S == 8, "S=%d", p.S);
    CHECK((1 << p.logS) == p.S, "logS");
    CHECK(p.nblocks % 8 == 0 && p.nblocks == ((p.nsuper + 7) / 8) * 8 * p.S * p.S, "nblocks");
    if (p.tri == 2) CHECK(p.SM <= DMA_MAX_SM && p.sprefix[p.SM] == p.nsuper, "staircase prefix");
    std::vector<unsigned char> seen((size_t)Tm * Tn, 0);
    long live_enumerated = 0;
    for (int b = 0; b < p.nblocks; ++b) {
        int ti = -1, tj = -1;
        if (!plan_block_to_tile(p, b, ti, tj)) continue;
        CHECK(ti >= 0 && ti < Tm && tj >= 0 && tj < Tn, "tile out of range b=%d -> (%d,%d)", b, ti, tj);
        if (!plan_tile_live(p, ti, tj, map ? map->data() : nullptr)) continue;
        CHECK(!seen[(size_t)ti * Tn + tj], "tile (%d,%d) enumerated twice (Tm=%ld Tn=%ld lower=%d S=%d tri=%d)", ti, tj, (long)Tm, (long)Tn, lower, p.S, p.tri);
        seen[(size_t)ti * Tn + tj] = 1;
        ++live_enumerated;
    }
    long live_brute = 0;
    for (int64_t ti = 0; ti < Tm; ++ti)
        for (int64_t tj = 0; tj < Tn; ++tj) {
            bool live = true;
            if (lower && tj * 128 > ti * 128 + 127 + diag_off) live = false;
            if (map && tj * 128 >= (*map)[(size_t)(ti / rbt)]) live = false;
            if (live) {
                ++live_brute;
                CHECK(seen[(size_t)ti * Tn + tj], "live tile (%ld,%ld) never enumerated (Tm=%ld Tn=%ld lower=%d diag=%ld S=%d tri=%d map=%d host=%d)",
                      (long)ti, (long)tj, (long)Tm, (long)Tn, lower, (long)diag_off, p.S, p.tri, map != nullptr, (int)host_copy);
            }
        }
    CHECK(live_brute == live_enumerated, "live %ld != enumerated %ld", live_brute, live_enumerated);
    return live_brute;
}

int main() {
    std::mt19937 rng(1234);
    long tiles = 0, plans = 0;
    const int forces[] = {0, 1, 2, 4, 8};
    // rectangles and triangles, ragged sizes, every supertile edge
    const int64_t dims[] = {1, 2, 3, 5, 7, 8, 9, 15, 16, 17, 31, 33, 64, 100, 127, 128, 129, 255, 512};
    for (int64_t Tm : dims)
        for (int64_t Tn : dims) {
            if (Tm * Tn > 40000) continue;
            for (int f : forces) {
                tiles += check_plan(Tm, Tn, 0, 0, nullptr, false, 1, f); ++plans;
                for (int64_t doff : {(int64_t)0, (int64_t)-128, (int64_t)128, (int64_t)-1000, (int64_t)4096, (int64_t)64}) {
                    tiles += check_plan(Tm, Tn, 1, doff, nullptr, false, 1, f); ++plans;
                }
            }
        }
    // the large square cases of the product: N = 65536 and 131072 trailing updates
    for (int64_t T : {(int64_t)496, (int64_t)512, (int64_t)1008, (int64_t)1024}) {
        tiles += check_plan(T, T, 1, 0, nullptr, false, 1, 0); ++plans;
        tiles += check_plan(T, 16, 1, 0, nullptr, false, 1, 0); ++plans;
    }
    // row maps (a rank's stacked row blocks: staircases), with and without the host copy, bands of 1 .. 16 tiles
    for (int trial = 0; trial < 400; ++trial) {
        const int rbt = 1 << (rng() % 5);
        const int bands = 1 + rng() % 40;
        const int64_t Tm = (int64_t)bands * rbt - (rng() % rbt);            // the last band may be cut short
        const int64_t Tn = 1 + rng() % 300;
        std::vector<int32_t> map((size_t)bands);
        const int shape = rng() % 3;
        for (int q = 0; q < bands; ++q) {
            if (shape == 0) map[(size_t)q] = (int32_t)std::min<int64_t>(Tn * 128, (int64_t)(q + 1) * rbt * 128);      // staircase
            else if (shape == 1) map[(size_t)q] = (int32_t)((rng() % (Tn + 1)) * 128);                                // arbitrary
            else map[(size_t)q] = (int32_t)((rng() % (Tn * 128 + 1)));                                               // not tile aligned
        }
        if (rng() % 8 == 0) map[rng() % map.size()] = 0;
        for (int f : forces) {
            if (f && ((Tm + f - 1) / f) > DMA_MAX_SM) continue;
            tiles += check_plan(Tm, Tn, 0, 0, &map, true, rbt, f); ++plans;
            tiles += check_plan(Tm, Tn, 0, 0, &map, false, rbt, f); ++plans;
        }
    }
    // a staircase with SM = DMA_MAX_SM (fits) and DMA_MAX_SM + 1 supertile rows (must fall back, not overrun)
    for (int64_t SMrows : {(int64_t)DMA_MAX_SM, (int64_t)DMA_MAX_SM + 1, (int64_t)DMA_MAX_SM * 3}) {
        const int S = 8;
        const int64_t Tm = SMrows * S, Tn = 40;
        std::vector<int32_t> map((size_t)Tm);
        for (int64_t q = 0; q < Tm; ++q) map[(size_t)q] = (int32_t)std::min<int64_t>(Tn * 128, (q % 64 + 1) * 128);
        TilePlan p;
        CHECK(plan_tiles(p, Tm, Tn, 0, 0, true, map.data(), (int)map.size(), 1, S), "plan");
        CHECK((SMrows <= DMA_MAX_SM) == (p.tri == 2), "SM=%ld tri=%d: the staircase must be used iff its table holds it", (long)SMrows, p.tri);
        tiles += check_plan(Tm, Tn, 0, 0, &map, true, 1, S); ++plans;
    }
    // the deal of blocks to the 32 shader engines (block b -> XCD b % 8, engine (b >> 3) % 4, 8 CUs each) behind the choice
    // of the supertile edge: the figure against an independent count, the chosen edge within 1 % of the best candidate,
    // and the diagonal supertiles' live tiles spread evenly over the four engines of their XCD
    for (int Tn : {8, 16, 32, 40, 64, 72, 96, 104, 112, 120, 128, 200, 256, 496}) {
        double eff_of[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
        for (int S : {8, 4, 2, 1}) {
            TilePlan p;
            CHECK(plan_tiles(p, Tn + 1, Tn, 1, 0, false, nullptr, 0, 1, S), "plan");
            long load[32], total = 0;
            for (long& l : load) l = 0;
            for (int b = 0; b < p.nblocks; ++b) {
                int ti, tj;
                if (plan_block_to_tile(p, b, ti, tj) && plan_tile_live(p, ti, tj, nullptr)) { ++load[(b & 7) + 8 * ((b >> 3) & 3)]; ++total; }
            }
            long rounds = 0;
            for (long l : load) rounds = std::max(rounds, (l + 7) / 8);
            const double eff = rounds ? (double)total / 256.0 / (double)rounds : 1.0;
            CHECK(fabs(eff - plan_xcd_efficiency(p, nullptr)) < 1e-12, "efficiency Tn=%d S=%d: %g vs %g", Tn, S, eff, plan_xcd_efficiency(p, nullptr));
            eff_of[S] = eff;
            if (p.tri == 1 && S >= 4 && Tn % S == 0) {
                // one diagonal supertile: its S (S + 1) / 2 live tiles over the 4 engines differ by at most 1
                const int s0 = 0;                                           // supertile 0 is (0, 0): blocks 8 q + 0
                int per[4] = {0, 0, 0, 0};
                for (int q = 0; q < S * S; ++q) {
                    int ti, tj;
                    const int b = ((s0 / 8) * S * S + q) * 8 + (s0 % 8);
                    if (plan_block_to_tile(p, b, ti, tj) && plan_tile_live(p, ti, tj, nullptr)) ++per[q & 3];
                }
                const int mx = std::max(std::max(per[0], per[1]), std::max(per[2], per[3]));
                const int mn = std::min(std::min(per[0], per[1]), std::min(per[2], per[3]));
                CHECK(mx - mn <= 1 && per[0] + per[1] + per[2] + per[3] == S * (S + 1) / 2, "diagonal supertile S=%d: %d %d %d %d", S, per[0], per[1], per[2], per[3]);
            }
        }
        if (Tn <= 256) {
            TilePlan p;
            CHECK(plan_tiles(p, Tn + 1, Tn, 1, 0, false, nullptr, 0, 1), "plan");
            double best = 0.0;
            for (int S : {8, 4, 2}) if (S <= Tn) best = std::max(best, eff_of[S]);
            CHECK(p.tri != 1 || p.S == 1 || eff_of[p.S] >= best - 0.01 - 1e-12, "Tn=%d: chosen S=%d (%.4f) is not within 1 %% of the best deal (%.4f)", Tn, p.S, eff_of[p.S], best);
        }
    }
    // refused arguments
    { TilePlan p; CHECK(!plan_tiles(p, 0, 4, 0, 0, false, nullptr, 0, 1), "Tm = 0 accepted");
      int32_t one = 128; CHECK(!plan_tiles(p, 4, 4, 0, 0, true, &one, 0, 1), "row_bands = 0 accepted"); }
    // block-width schedules
    for (int64_t NB : {(int64_t)128, (int64_t)512, (int64_t)1024, (int64_t)2048})
        for (int64_t ncols : {(int64_t)128, (int64_t)1024, (int64_t)12288, (int64_t)16384, (int64_t)65536, (int64_t)131072, (int64_t)65536 + 128})
            for (int ramp : {0, 1, 2, 3, 7, 2 | (5 << 4)}) {
                const std::vector<int64_t> w = plan_block_widths(NB, ncols, true, ramp);
                int64_t sum = 0;
                for (int64_t v : w) { CHECK(v > 0 && v <= NB, "width %ld", (long)v); sum += v; }
                CHECK(sum == ncols, "widths sum %ld != %ld", (long)sum, (long)ncols);
                const std::vector<int64_t> w0 = plan_block_widths(NB, ncols, false, ramp);
                for (size_t i = 0; i + 1 < w0.size(); ++i) CHECK(w0[i] == NB, "fixed schedule");
            }
    // flop counts against element-by-element sums
    for (int trial = 0; trial < 200; ++trial) {
        const int64_t M = 128 * (1 + rng() % 12), N = 128 * (1 + rng() % 12), K = 16 * (1 + rng() % 8);
        const int64_t doff = (int64_t)(rng() % 2000) - 1000, real = 1 + rng() % (M + 200);
        double alg = 0, til = 0;
        for (int64_t r = 0; r < M; ++r)
            for (int64_t c = 0; c < N; ++c)
                if (c <= r + doff && r < real) alg += 2.0 * K;
        for (int64_t ti = 0; ti < M / 128; ++ti)
            for (int64_t tj = 0; tj < N / 128; ++tj)
                if (tj * 128 <= ti * 128 + 127 + doff) til += 2.0 * 128 * 128 * K;
        CHECK(alg == plan_algorithmic_flops(M, N, K, 1, doff, real), "algorithmic flops M=%ld N=%ld doff=%ld real=%ld: %g vs %g", (long)M, (long)N, (long)doff, (long)real, alg, plan_algorithmic_flops(M, N, K, 1, doff, real));
        CHECK(til == plan_tile_flops(M, N, K, 1, doff), "tile flops");
    }
    printf("plan_check: %s (%ld plans, %ld live tiles)\n", g_fail ? "FAILED" : "ok", plans, tiles);
    return g_fail ? 1 : 0;
}
