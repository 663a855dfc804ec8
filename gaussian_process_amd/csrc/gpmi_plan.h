// Host-side planning of the launches, free of any HIP dependency: which tiles a GEMM launch enumerates and in what
// order (XCD-aware supertiles, the triangular and the staircase enumerations), which of them are live, how the block
// widths of a blocked sweep are laid out, and the algorithmic flop counts the bench line is priced on.
// The same header is compiled by hipcc into the kernels (block -> tile map) and by plain g++ under
// -fsanitize=address,undefined in tests/test_sanitize_cpu.py, where tests/sanitize/plan_check.cpp holds every plan
// against brute force (each live tile enumerated exactly once, no dead one, no table overrun).
#pragma once
#include <math.h>
#include <stdint.h>

#include <algorithm>
#include <vector>

#if defined(__HIPCC__)
#define GPMI_HD __host__ __device__ __forceinline__
#else
#define GPMI_HD inline
#endif

namespace gpmi {

constexpr int PLAN_TILE = 128;      // tile edge of the LDS-DMA GEMM
constexpr int DMA_MAX_SM = 128;     // supertile rows a staircase launch can describe (M up to 131072 at S = 8)

// The enumeration of one launch: blocks b = 0 .. nblocks - 1; block b belongs to XCD group b % 8 (the hardware deals
// workgroups round-robin over the 8 XCDs), blocks 8 w + x with equal w / S^2 form supertile (w / S^2) * 8 + x: S x S
// tiles that share A rows and B rows in that XCD's L2.
struct TilePlan {
    int Tm = 0, Tn = 0;             // tiles
    int lower = 0;                  // skip tiles entirely above {col <= row + diag_off}
    int64_t diag_off = 0;
    int row_block_tiles = 1;        // row map: 128-row tiles per band
    int S = 1, logS = 0, SM = 0, SN = 0;
    int tri = 0;                    // 0 rectangle (columns rotated by the supertile row), 1 lower triangle of supertiles, 2 staircase
    int nsuper = 0;
    int nblocks = 0;
    int sprefix[DMA_MAX_SM + 1];    // tri == 2: supertile row si holds its leftmost sprefix[si + 1] - sprefix[si] supertiles
    const int32_t* row_ncols = nullptr;   // tri == 2: the row map (host copy here; the kernels' struct holds the device copy)
};

// block number -> tile; false: the block lies outside the tile grid (padding of the enumeration)
template <class PT>
GPMI_HD bool plan_block_to_tile(const PT& p, int b, int& ti, int& tj) {
    const int xcd = b & 7;
    const int w = b >> 3;
    const int S2 = p.S * p.S;
    const int s = (w / S2) * 8 + xcd;
    if (s >= p.nsuper) return false;
    const int q = w % S2;
    int si, sj;
    if (p.tri == 2) {
        si = 0;
        while (si + 1 < p.SM && p.sprefix[si + 1] <= s) ++si;
        sj = s - p.sprefix[si];
        if (sj == p.sprefix[si + 1] - p.sprefix[si] - 1 && p.row_ncols) {
            // the LAST live supertile of a staircase row is ragged (a row block's own diagonal block: its lower triangle):
            // its live tiles first, row by row, the dead blocks behind them -- consecutive blocks go to the XCD's four
            // shader engines in turn, and as a square the engine with the supertile's first columns carried most of it
            int acc = 0;
            for (int r = 0; r < p.S; ++r) {
                const int tr = si * p.S + r;
                int w = 0;
                if (tr < p.Tm) {
                    const int64_t nc = p.row_ncols[tr / p.row_block_tiles];
                    int64_t wl = (nc + PLAN_TILE - 1) / PLAN_TILE - (int64_t)sj * p.S;
                    if (wl > p.S) wl = p.S;
                    if (wl > (int64_t)p.Tn - (int64_t)sj * p.S) wl = (int64_t)p.Tn - (int64_t)sj * p.S;
                    w = wl > 0 ? (int)wl : 0;
                }
                if (q < acc + w) {
                    ti = tr;
                    tj = sj * p.S + (q - acc);
                    return true;
                }
                acc += w;
            }
            return false;
        }
    } else if (p.tri) {
        si = (int)((sqrtf(8.f * (float)s + 1.f) - 1.f) * 0.5f);
        while ((si + 1) * (si + 2) / 2 <= s) ++si;
        while (si * (si + 1) / 2 > s) --si;
        sj = s - si * (si + 1) / 2;
    } else {
        si = s / p.SN;
        sj = s - si * p.SN;
        // Supertile s runs on XCD s % 8.  With a row map (or a lower-mode rectangle) the live supertiles of
        // a row are its leftmost ones, so a fixed column -> XCD assignment (SN % 8 == 0) gives the XCDs that
        // own the low columns up to 1.5x the work of the others (measured: 54.8 against 67.3 TF/s on a
        // triangular region).  Rotating the columns by the row index stripes the XCDs diagonally instead.
        sj += si % p.SN;
        if (sj >= p.SN) sj -= p.SN;
    }
    if (p.tri == 1 && si == sj) {
        // A supertile ON the diagonal holds S (S + 1) / 2 live tiles.  They are enumerated FIRST, in row-major order of the
        // lower triangle, the dead blocks after them: inside an XCD consecutive blocks go to its four shader engines in
        // turn (gpmi_plan.h: plan_xcd_efficiency), and with the square enumeration the engine that gets tile columns 0 and
        // 4 of every diagonal supertile carried 12 live tiles where the one with columns 3 and 7 carried 6.
        int r = (int)((sqrtf(8.f * (float)q + 1.f) - 1.f) * 0.5f);
        while ((r + 1) * (r + 2) / 2 <= q) ++r;
        while (r * (r + 1) / 2 > q) --r;
        if (r >= p.S) return false;
        ti = si * p.S + r;
        tj = sj * p.S + (q - r * (r + 1) / 2);
        return ti < p.Tm && tj < p.Tn;
    }
    ti = si * p.S + (q >> p.logS);
    tj = sj * p.S + (q & (p.S - 1));
    return ti < p.Tm && tj < p.Tn;
}

// a tile the mode of the launch leaves untouched (above the diagonal, right of its row band); row_ncols: the launch's
// row map (device pointer in a kernel, host copy in a test) or null
template <class PT>
GPMI_HD bool plan_tile_live(const PT& p, int ti, int tj, const int32_t* row_ncols) {
    if (p.lower) {
        const int64_t min_col = (int64_t)tj * PLAN_TILE;
        const int64_t max_row = (int64_t)ti * PLAN_TILE + PLAN_TILE - 1;
        if (min_col > max_row + p.diag_off) return false;
    }
    if (row_ncols) {
        if ((int64_t)tj * PLAN_TILE >= row_ncols[ti / p.row_block_tiles]) return false;
    }
    return true;
}

// How a launch of one workgroup per tile is dealt out (measured, round 4: profiles/r04_resident_cost_count.txt): block b
// goes to XCD b % 8, and inside the XCD consecutive blocks go to its four shader engines in turn -- engine ((b >> 3) % 4) --,
// each of which places them on its own 8 CUs as they become free.  Nothing is balanced ACROSS engines: the launch lasts as
// long as the engine with the most live tiles needs, in whole rounds of 8 (one CU held by another kernel costs a per-tile
// launch 12 %, two in one engine 27 %, three 40 %: 8/7, 8/6, 8/5 -- while a resident form that draws tiles from counters
// loses the CUs' share, 0.4 % each).  plan_xcd_efficiency walks a plan's blocks and returns
//     (live tiles / 256) / max over the 32 engines of ceil(live tiles of the engine / 8):
// 1.0 = every engine ends in the same round.  With 8 x 8 supertiles enumerated as squares a lower-triangular launch of 496
// tile rows sits at 0.991, of 112 rows at 0.898, of 64 at 0.838; with the diagonal supertiles' live tiles first and the edge
// chosen by this figure at 0.997 / 0.968 / 0.931.
// Plan a launch over Tm x Tn tiles.  row_ncols_host / row_bands: host copy of the row map (null: none known on the
// host); has_row_map: the launch has a device row map (with or without a host copy).  force_S != 0 pins the supertile
// edge (tests).  balance_xcds: the launch is one workgroup per tile, dealt statically to the XCDs -- pick the supertile
// edge with the deal in mind (the resident forms draw tiles from counters and take over each other's tails: they keep
// the widest supertile).  Returns false when the arguments are unusable.
inline bool plan_tiles(TilePlan& p, int64_t Tm, int64_t Tn, int lower, int64_t diag_off, bool has_row_map,
                       const int32_t* row_ncols_host, int row_bands, int row_block_tiles, int force_S = 0,
                       bool balance_xcds = true);

inline double plan_xcd_efficiency(const TilePlan& p, const int32_t* row_ncols_host, int cus_per_engine = 8) {
    long load[32];
    for (int x = 0; x < 32; ++x) load[x] = 0;
    long total = 0;
    for (int b = 0; b < p.nblocks; ++b) {
        int ti, tj;
        if (plan_block_to_tile(p, b, ti, tj) && plan_tile_live(p, ti, tj, row_ncols_host)) { ++load[b & 31]; ++total; }
    }
    long rounds = 0;
    for (int x = 0; x < 32; ++x) rounds = std::max(rounds, (load[x] + cus_per_engine - 1) / cus_per_engine);
    return rounds ? (double)total / (32.0 * cus_per_engine) / (double)rounds : 1.0;
}

inline bool plan_tiles(TilePlan& p, int64_t Tm, int64_t Tn, int lower, int64_t diag_off, bool has_row_map,
                       const int32_t* row_ncols_host, int row_bands, int row_block_tiles, int force_S,
                       bool balance_xcds) {
    if (Tm <= 0 || Tn <= 0 || Tm > (1 << 20) || Tn > (1 << 20)) return false;
    if (row_ncols_host && row_bands <= 0) return false;
    p.Tm = (int)Tm; p.Tn = (int)Tn;
    p.lower = lower; p.diag_off = diag_off;
    p.row_block_tiles = row_block_tiles > 0 ? row_block_tiles : 1;
    p.tri = (lower && diag_off == 0 && 2 * p.Tn >= p.Tm) ? 1 : 0;
    p.row_ncols = nullptr;
    const bool stairs = has_row_map && row_ncols_host && row_bands > 0 && !lower;
    // supertile edge: 8 tiles, but never wider than the launch -- a strip of Tn = 1 (a panel-internal update of 128
    // columns) enumerated in 8 x 8 supertiles is seven dead workgroups for every live one, and a dead workgroup still has
    // to be handed a CU with 96 KiB of free LDS before it can return (32768 x 128 x 128: 88 us, beside a trailing update
    // each of them waits for a tile to finish)
    int S = 8;
    while (S > 1 && (S > p.Tn || S > p.Tm)) S >>= 1;
    // mid-size triangular launches: the largest supertile edge whose deal to the XCDs is within 1 % of the best one
    // (plan_tri_xcd_efficiency); never below 2 (an edge of 1 gives up all reuse of the operands in an XCD's L2)
    // a rank's staircase of row blocks (row map with a host copy) and mid-size triangular launches: the widest edge whose
    // deal is within 1 % of the best, by walking each candidate's blocks -- for launches small enough that a supertile more
    // or less on one engine shows (up to 256 tile columns of a triangle, 32768 tiles of a staircase)
    if (!force_S && balance_xcds && S > 2 &&
        ((p.tri == 1 && !stairs && p.Tn <= 256) || (stairs && (int64_t)p.Tm * p.Tn <= 32768))) {
        double eff[4] = {0., 0., 0., 0.}, best = 0.0;
        int cand[4], nc = 0;
        for (int c = S; c >= 2 && nc < 4; c >>= 1) {
            TilePlan q;
            if (!plan_tiles(q, Tm, Tn, lower, diag_off, has_row_map, row_ncols_host, row_bands, row_block_tiles, c, false)) break;
            if (stairs && q.tri != 2) break;             // the staircase table does not hold this edge
            cand[nc] = c;
            eff[nc] = plan_xcd_efficiency(q, row_ncols_host);
            best = std::max(best, eff[nc]);
            ++nc;
        }
        for (int i = 0; i < nc; ++i)
            if (eff[i] >= best - 0.01) { force_S = cand[i]; break; }
    }
    if (force_S) S = force_S;
    for (;; S >>= 1) {
        const int SM = (p.Tm + S - 1) / S, SN = (p.Tn + S - 1) / S;
        int ns = p.tri ? SM * (SM + 1) / 2 : SM * SN;
        bool use_stairs = false;
        if (stairs && SM <= DMA_MAX_SM) {
            // live supertiles per supertile row: up to the widest band of the row
            int tot = 0;
            p.sprefix[0] = 0;
            for (int si = 0; si < SM; ++si) {
                int64_t widest = 0;
                for (int ti = si * S; ti < std::min((si + 1) * S, p.Tm); ++ti) {
                    const int band = std::min(ti / p.row_block_tiles, row_bands - 1);
                    widest = std::max<int64_t>(widest, row_ncols_host[band]);
                }
                widest = std::max<int64_t>(widest, 0);
                const int64_t live = std::min<int64_t>(SN, (widest + (int64_t)S * PLAN_TILE - 1) / ((int64_t)S * PLAN_TILE));
                tot += (int)live;
                p.sprefix[si + 1] = tot;
            }
            ns = tot;
            use_stairs = true;
        }
        if (ns >= 32 || S == 1 || force_S) {
            p.S = S; p.SM = SM; p.SN = SN; p.nsuper = ns;
            if (use_stairs) { p.tri = 2; p.row_ncols = row_ncols_host; }
            break;
        }
    }
    p.logS = (p.S == 8) ? 3 : (p.S == 4) ? 2 : (p.S == 2) ? 1 : 0;
    p.nblocks = ((p.nsuper + 7) / 8) * 8 * p.S * p.S;
    return true;
}

// Block widths of a blocked sweep over ncols columns at nominal width NB.  ramp (bit mask, only with ramp_ok): 1 ramp up
// at the start (NB/4, NB/4, NB/2), 2 half width over the last `ramp >> 4` (default 3) blocks, 4 quarter width for the
// last block.
inline std::vector<int64_t> plan_block_widths(int64_t NB, int64_t ncols, bool ramp_ok, int ramp) {
    std::vector<int64_t> w;
    if (NB <= 0 || ncols <= 0) return w;
    const bool on = ramp_ok && ramp && NB >= 1024 && ncols >= 8 * NB;
    const bool up = on && (ramp & 1), down = on && (ramp & 2);
    const int64_t tail = (ramp >> 4) ? (ramp >> 4) : 3;       // blocks at the end that run at half width
    int64_t done = 0;
    while (done < ncols) {
        int64_t nb = NB;
        const int64_t left = ncols - done;
        if (up && w.size() < 2) nb = NB / 4;
        else if (up && w.size() < 3) nb = NB / 2;
        else if (down && left <= NB && (ramp & 4)) nb = NB / 4;
        else if (down && left <= tail * NB) nb = NB / 2;
        nb = std::min(nb, ncols - done);
        w.push_back(nb);
        done += nb;
    }
    return w;
}

// flops the tiles of a launch compute (whole tiles; TN = 128 when N % 128 == 0, else 64)
inline double plan_tile_flops(int64_t M, int64_t N, int64_t K, int lower, int64_t diag_off) {
    if (M <= 0 || N <= 0 || K <= 0) return 0.;
    const int TN = (N % 128 == 0) ? 128 : 64;
    const int64_t Tm = M / 128, Tn = N / TN;
    int64_t tiles = 0;
    for (int64_t ti = 0; ti < Tm; ++ti) {
        if (!lower) { tiles += Tn; continue; }
        const int64_t lim = ti * 128 + 127 + diag_off;        // tiles tj with tj * TN <= lim
        if (lim < 0) continue;
        const int64_t cnt = lim / TN + 1;
        tiles += cnt < Tn ? cnt : Tn;
    }
    return 2.0 * (double)tiles * 128.0 * TN * (double)K;
}

// Algorithmic flops of a lower-mode update: 2 K per element on or below the diagonal (col <= row + diag_off) of the
// first `real_rows` rows -- what the Cholesky needs, as opposed to what the tiles compute (whole diagonal tiles,
// padding rows).
inline double plan_algorithmic_flops(int64_t M, int64_t N, int64_t K, int lower, int64_t diag_off, int64_t real_rows) {
    if (M <= 0 || N <= 0 || K <= 0) return 0.;
    const int64_t rows = std::min<int64_t>(M, real_rows);
    if (!lower) return 2.0 * (double)rows * (double)N * (double)K;
    double elems = 0.;                                                       // row r reaches min(N, r + diag_off + 1) columns
    const int64_t r_full = std::max<int64_t>(0, N - 1 - diag_off);          // first row that reaches all N columns
    const int64_t r_first = std::max<int64_t>(0, -diag_off);                // first row that reaches column 0
    const int64_t tri_end = std::min(rows, r_full);
    if (tri_end > r_first) {
        const double n = (double)(tri_end - r_first);
        const double first = (double)(r_first + diag_off + 1);
        elems += n * first + n * (n - 1) / 2.0;
    }
    if (rows > r_full) elems += (double)(rows - std::max(r_full, (int64_t)0)) * (double)N;
    return 2.0 * elems * (double)K;
}

}  // namespace gpmi
