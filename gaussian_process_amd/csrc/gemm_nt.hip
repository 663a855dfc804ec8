// fp64 MFMA GEMM  C (M x N) op= A (M x K) * B (N x K)^T  for gfx950.
//
// This is the Cholesky trailing update (SYRK when A == B, lower tiles only),
// the panel-internal rank-64 updates and the TRSM sweep's updates.  Both
// operands are K-contiguous row slices of the row-major factor, so the same
// staging serves A and B.
//
// Block = 256 threads = 4 wavefronts (2 x 2), tile 128 x (32*NI), K step 16.
// Each wavefront owns a 64 x (16*NI) sub-tile = 4 x NI accumulators of
// v_mfma_f64_16x16x4_f64 (4 f64 = 8 VGPRs each).
//
// LDS image of one operand tile for one K step: [kp = 0..7][row slot][2 f64],
// i.e. 16-byte k-pairs, slot = row ^ kp.  Lane l of an MFMA reads row l&15,
// k-pair 4t + (l>>4): one ds_read_b128 feeds two MFMAs (the k order inside a
// K step is permuted identically for A and B, which leaves the sum unchanged).
// With the XOR both the ds_write_b128 of the coalesced staging (8 lanes = one
// 128-byte row segment) and the ds_read_b128 fragment reads are conflict-free.
//
// Tile order: 1-D grid; block b runs on XCD group b % 8 (round-robin dispatch,
// a speed assumption only), and each group walks whole S x S super-tiles so
// that the blocks resident on one XCD share A/B row slabs in its L2.
#include <algorithm>

#include "gpmi_internal.h"
#include "gpmi_plan.h"

namespace gpmi {

typedef double d4 __attribute__((ext_vector_type(4)));
typedef double d2 __attribute__((ext_vector_type(2)));

constexpr int BK = 16;
constexpr int KP = BK / 2;  // k-pairs per K step

struct GemmDev {
    double* C;
    const double* A;
    const double* B;
    int64_t ldc, lda, ldb;
    int Tm, Tn;        // tiles in M, N
    int nchunks;       // K / 16
    int mode, lower;
    int64_t diag_off;
    int S, logS;       // super-tile edge (tiles)
    int SM, SN;        // super-tiles in M, N
    int tri;           // triangular super-tile enumeration
    int nsuper;
    const int32_t* row_ncols;
    int row_block_tiles;
    int dbg;           // timing-only ablations (gpmi_probe_gemm): results are wrong when non-zero
};

__device__ __forceinline__ bool map_tile(const GemmDev& p, int& ti, int& tj) {
    const int b = blockIdx.x;
    const int xcd = b & 7;
    const int w = b >> 3;
    const int S2 = p.S * p.S;
    const int s = (w / S2) * 8 + xcd;
    if (s >= p.nsuper) return false;
    const int q = w % S2;
    int si, sj;
    if (p.tri) {
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
    ti = si * p.S + (q >> p.logS);
    tj = sj * p.S + (q & (p.S - 1));
    return ti < p.Tm && tj < p.Tn;
}

template <int MI, int NI, bool DBG>
__global__ __launch_bounds__(256, 2) void gemm_nt_kernel(const GemmDev p) {
    const int dbg = DBG ? p.dbg : 0;   // ablation bits exist only in the probe instantiation
    constexpr int TM = 32 * MI;               // 128 (MI = 4) or 64 (MI = 2: low-latency small tiles)
    constexpr int TN = 32 * NI;
    constexpr int A_SLOTS = KP * TM;          // 16-byte slots per stage
    constexpr int B_SLOTS = KP * TN;
    constexpr int A_LD = TM * KP / 256;       // staging loads per thread
    constexpr int B_LD = TN * KP / 256;
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    d2* smem = reinterpret_cast<d2*>(smem_raw);
    // layout: [stage][A slots | B slots]
    constexpr int STAGE = A_SLOTS + B_SLOTS;

    int ti, tj;
    if (!map_tile(p, ti, tj)) return;
    if (p.lower) {
        // skip tiles entirely above {col <= row + diag_off}
        const int64_t min_col = (int64_t)tj * TN;
        const int64_t max_row = (int64_t)ti * TM + TM - 1;
        if (min_col > max_row + p.diag_off) return;
    }
    if (p.row_ncols) {
        if ((int64_t)tj * TN >= p.row_ncols[ti / p.row_block_tiles]) return;
    }

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int wr = (wave >> 1) * (16 * MI);     // wave row offset in tile
    const int wc = (wave & 1) * (16 * NI);      // wave col offset in tile
    const int fr = lane & 15;
    const int fg = lane >> 4;

    const double* Ag = p.A + ((dbg & 16) ? 0 : (int64_t)ti * TM * p.lda);
    const double* Bg = p.B + ((dbg & 16) ? 0 : (int64_t)tj * TN * p.ldb);

    // staging assignment: piece = tid + 256*i -> row = piece>>3, kp = piece&7
    const int st_kp = tid & 7;
    const int st_row = tid >> 3;                // + 32*i
    const double* a_src = Ag + (int64_t)st_row * p.lda + st_kp * 2;
    const double* b_src = Bg + (int64_t)st_row * p.ldb + st_kp * 2;

    d2 ra[A_LD], rb[B_LD];
    auto load_stage = [&](int chunk) {
        const int k0 = chunk * BK;
#pragma unroll
        for (int i = 0; i < A_LD; ++i)
            ra[i] = *reinterpret_cast<const d2*>(a_src + (int64_t)(32 * i) * p.lda + k0);
#pragma unroll
        for (int i = 0; i < B_LD; ++i)
            rb[i] = *reinterpret_cast<const d2*>(b_src + (int64_t)(32 * i) * p.ldb + k0);
    };
    auto write_stage = [&](int buf) {
        d2* sa = smem + buf * STAGE;
        d2* sb = sa + A_SLOTS;
#pragma unroll
        for (int i = 0; i < A_LD; ++i) {
            const int row = st_row + 32 * i;
            sa[st_kp * TM + (row ^ st_kp)] = ra[i];
        }
#pragma unroll
        for (int i = 0; i < B_LD; ++i) {
            const int row = st_row + 32 * i;
            sb[st_kp * TN + (row ^ st_kp)] = rb[i];
        }
    };

    d4 acc[MI][NI];
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < NI; ++j) acc[i][j] = d4{0., 0., 0., 0.};

    load_stage(0);
    write_stage(0);
    __syncthreads();

    const int nch = p.nchunks;
    for (int c = 0; c < nch; ++c) {
        const int buf = c & 1;
        // branch-free body (one scheduling region): the last step re-loads its own
        // chunk and refills a stage nobody reads any more
        if (!(dbg & 1)) load_stage(c + 1 < nch ? c + 1 : c);
        const d2* sa = smem + buf * STAGE;
        const d2* sb = sa + A_SLOTS;
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            const int kp = 4 * t + fg;
            d2 fa[MI], fb[NI];
#pragma unroll
            for (int i = 0; i < MI; ++i) fa[i] = sa[kp * TM + ((wr + 16 * i + fr) ^ kp)];
#pragma unroll
            for (int j = 0; j < NI; ++j) fb[j] = sb[kp * TN + ((wc + 16 * j + fr) ^ kp)];
#pragma unroll
            for (int i = 0; i < MI; ++i)
#pragma unroll
                for (int j = 0; j < NI; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(fa[i].x, fb[j].x, acc[i][j], 0, 0, 0);
            // the refill of the other LDS stage (last read before the previous barrier)
            // rides in the shadow of the step's last MFMAs; with a branch-free body the
            // compiler also sinks half of those MFMAs below the barrier, so the barrier
            // wait overlaps matrix work
            if (t == 1 && !(dbg & 3)) write_stage(buf ^ 1);
#pragma unroll
            for (int i = 0; i < MI; ++i)
#pragma unroll
                for (int j = 0; j < NI; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(fa[i].y, fb[j].y, acc[i][j], 0, 0, 0);
        }
        if (!(dbg & 2)) __syncthreads();
    }

    // D layout of v_mfma_f64_16x16x4_f64: col = lane&15, row = 4*v + (lane>>4)
    double* Cg = p.C + ((int64_t)ti * TM + wr) * p.ldc + (int64_t)tj * TN + wc;
    auto c_ptr = [&](int i, int j, int v) { return Cg + (int64_t)(16 * i + 4 * v + fg) * p.ldc + 16 * j + fr; };
    if ((DBG && (dbg & 8))) {
        double t = 0.;
#pragma unroll
        for (int i = 0; i < MI; ++i)
#pragma unroll
            for (int j = 0; j < NI; ++j) t += acc[i][j][0] + acc[i][j][1] + acc[i][j][2] + acc[i][j][3];
        if (t == 123.456) Cg[0] = t;
    } else if (p.mode == 0 && !(dbg & 4)) {
        // C -= acc in 16-row bands: all loads of a band are issued before its first
        // store (a load behind a possibly aliasing store would otherwise wait for it:
        // 64 serial memory round trips per lane)
#pragma unroll
        for (int i = 0; i < MI; ++i) {
            double cv[NI][4];
#pragma unroll
            for (int j = 0; j < NI; ++j)
#pragma unroll
                for (int v = 0; v < 4; ++v) cv[j][v] = *c_ptr(i, j, v);
#pragma unroll
            for (int j = 0; j < NI; ++j)
#pragma unroll
                for (int v = 0; v < 4; ++v) *c_ptr(i, j, v) = cv[j][v] - acc[i][j][v];
        }
    } else {
#pragma unroll
        for (int i = 0; i < MI; ++i)
#pragma unroll
            for (int j = 0; j < NI; ++j)
#pragma unroll
                for (int v = 0; v < 4; ++v) *c_ptr(i, j, v) = acc[i][j][v];
    }
}

static void plan(const GemmArgs& a, int TM, int TN, GemmDev& p, int& nblocks) {
    p.C = a.C; p.A = a.A; p.B = a.B;
    p.ldc = a.ldc; p.lda = a.lda; p.ldb = a.ldb;
    p.Tm = (int)(a.M / TM);
    p.Tn = (int)(a.N / TN);
    p.nchunks = (int)(a.K / BK);
    p.mode = a.mode;
    p.lower = a.lower;
    p.diag_off = a.diag_off;
    p.row_ncols = a.row_ncols;
    p.row_block_tiles = (a.row_block_tiles > 0 ? a.row_block_tiles : 1) * (128 / TM);   // in TM-row bands
    // triangular super-tile enumeration only for square tiles on the diagonal
    // (and only when the region is not a tall skinny strip, where most
    // triangular super-tiles would be empty)
    p.tri = (a.lower && TM == TN && a.diag_off == 0 && 2 * p.Tn >= p.Tm) ? 1 : 0;
    int S = 8;
    for (;; S >>= 1) {
        const int SM = (p.Tm + S - 1) / S, SN = (p.Tn + S - 1) / S;
        const int ns = p.tri ? SM * (SM + 1) / 2 : SM * SN;
        if (ns >= 32 || S == 1) {
            p.S = S; p.SM = SM; p.SN = SN; p.nsuper = ns;
            break;
        }
    }
    p.logS = (p.S == 8) ? 3 : (p.S == 4) ? 2 : (p.S == 2) ? 1 : 0;
    nblocks = ((p.nsuper + 7) / 8) * 8 * p.S * p.S;
    p.dbg = tuning().gemm_dbg;
}

bool gemm_nt_routes_dma(const GemmArgs& a) {
    // ablation bits >= 256 select the DMA kernel's ablations (low byte passed on)
    const Tuning& tn = tuning();
    if (a.b_block_off) return true;            // only the LDS-DMA kernel reads B through a block table
    return tn.gemm_use_dma && (!tn.gemm_dbg || tn.gemm_dbg >= 256) && gemm_dma_eligible(a) &&
           (a.M / 128) * (a.N / 128) >= 128;      // from half a round of tiles up (below: 64 x 64 tiles, launch_gemm_nt_small)
}

hipError_t launch_gemm_nt(hipStream_t s, const GemmArgs& a) {
    if (a.M <= 0 || a.N <= 0 || a.K <= 0) return hipSuccess;
    if (a.M % 128 || a.N % 64 || a.K % BK) return hipErrorInvalidValue;
    if (a.b_block_off && !gemm_dma_eligible(a)) return hipErrorInvalidValue;
    if (gemm_nt_routes_dma(a)) return launch_gemm_nt_dma(s, a);
    GemmDev p;
    int nblocks;
    static PerDeviceOnce once;
    const hipError_t ea = once.run([&]() -> hipError_t {
        constexpr int big = 2 * (KP * 128 + KP * 128) * 16;
        hipError_t e = hipFuncSetAttribute((const void*)gemm_nt_kernel<4, 4, false>, hipFuncAttributeMaxDynamicSharedMemorySize, big);
        if (e != hipSuccess) return e;
        return hipFuncSetAttribute((const void*)gemm_nt_kernel<4, 4, true>, hipFuncAttributeMaxDynamicSharedMemorySize, big);
    });
    if (ea != hipSuccess) return ea;
    const Tuning& tn = tuning();
    // few tiles: one 128 x 128 tile keeps a CU busy for 1.8 us per 64 of K while the rest of the
    // chip idles -- 64 x 64 tiles finish 4x sooner (panel-internal updates, diagonal blocks)
    const int64_t tiles128 = (a.M / 128) * ((a.N + 127) / 128);
    if (tn.gemm_small_tiles && tn.gemm_small_dma && !tn.gemm_dbg && tiles128 < 128 && gemm_small_eligible(a))
        return launch_gemm_nt_small(s, a);
    if (tn.gemm_small_tiles && !tn.gemm_dbg && tiles128 < 128) {
        plan(a, 64, 64, p, nblocks);
        constexpr size_t lds = 2 * (KP * 64 + KP * 64) * 16;
        hipLaunchKernelGGL((gemm_nt_kernel<2, 2, false>), dim3(nblocks), dim3(256), lds, s, p);
    } else if (a.N % 128 == 0) {
        plan(a, 128, 128, p, nblocks);
        constexpr size_t lds = 2 * (KP * 128 + KP * 128) * 16;
        if (p.dbg) hipLaunchKernelGGL((gemm_nt_kernel<4, 4, true>), dim3(nblocks), dim3(256), lds, s, p);
        else hipLaunchKernelGGL((gemm_nt_kernel<4, 4, false>), dim3(nblocks), dim3(256), lds, s, p);
    } else {
        plan(a, 128, 64, p, nblocks);
        constexpr size_t lds = 2 * (KP * 128 + KP * 64) * 16;
        hipLaunchKernelGGL((gemm_nt_kernel<4, 2, false>), dim3(nblocks), dim3(256), lds, s, p);
    }
    return hipGetLastError();
}

double gemm_nt_flops(const GemmArgs& a) {
    return plan_tile_flops(a.M, a.N, a.K, a.lower, a.diag_off);
}

// Algorithmic flops of a lower-mode update: 2 K per element on or below the diagonal
// (col <= row + diag_off) of the first `real_rows` rows -- what the Cholesky needs, as opposed to
// what the tiles compute (whole diagonal tiles, padding rows).  (gpmi_plan.h)
double gemm_nt_algorithmic_flops(const GemmArgs& a, int64_t real_rows) {
    return plan_algorithmic_flops(a.M, a.N, a.K, a.lower, a.diag_off, real_rows);
}

}  // namespace gpmi
