// fp64 MFMA GEMM  C (M x N) -= A (M x K) * B (N x K)^T, second generation:
// ONE workgroup per CU (one wavefront per SIMD) that keeps the matrix pipe fed by
// itself, so that the other half of every CU's registers and 64 KiB of its LDS stay
// free for the latency-bound panel kernels of the next Cholesky step (lookahead).
//
// Same tile and MFMA decomposition as gemm_nt.hip (128 x 128 tile, 4 waves of 4 x 4
// v_mfma_f64_16x16x4_f64 accumulators, K step 16).  What changes is the staging:
//   * global -> LDS by LDS-DMA (global_load_lds_dwordx4): no staging registers, no
//     ds_write instructions; a 3-stage ring, K step c+2 is in flight while c computes;
//   * LDS image [row][k-pair ^ (row & 7)] (16-byte slots, 128 B per row): a DMA wave
//     instruction writes 64 consecutive slots = 8 whole rows, whose 8 lanes read one
//     full 128-byte line of the operand; ds_read_b128 fragment reads are conflict-free
//     (brute-force checked over all lane groups);
//   * fragments double-buffered in registers: the reads for the next half step are
//     issued a half step ahead;
//   * ONE barrier per K step, placed in the middle of the step's 64 MFMAs: it
//     publishes stage c+1 (every wave has waited for its own DMA pieces) and retires
//     the last reads of the stage that the next DMA overwrites.
//
// Order per K step c (per wave), MFMAs in four quarters (F0.x, F0.y, F1.x, F1.y):
//   F0.x + read F1 <- stage c%3, k-pairs 4..7 | F0.y + issue DMA(c+2) -> stage (c+2)%3
//   s_waitcnt vmcnt(pieces of c+2 may fly), lgkmcnt(0) | s_barrier
//   F1.x + read F0 <- stage (c+1)%3, k-pairs 0..3 | F1.y
#include <algorithm>
#include <cstdlib>

#include "gpmi_internal.h"
#include "gpmi_plan.h"
#include <atomic>
#include <mutex>

namespace gpmi {

typedef double d4 __attribute__((ext_vector_type(4)));
typedef double d2 __attribute__((ext_vector_type(2)));

#define GPMI_LDS __attribute__((address_space(3)))
#define GPMI_GLB __attribute__((address_space(1)))
#define GPMI_CONST __attribute__((address_space(4)))

// The launch geometry (supertiles, the triangular / staircase enumerations, the block -> tile map) lives in
// gpmi_plan.h, free of HIP, where a CPU test holds it against brute force under the sanitizers.
struct GemmDmaDev {
    double* C;
    const double* A;
    const double* B;
    int64_t ldc, lda, ldb;
    int Tm, Tn;
    int nchunks;
    int lower;
    int64_t diag_off;
    const int32_t* row_ncols;
    int row_block_tiles;
    const int64_t* b_block_off;   // B as a table of row blocks (see GemmArgs), or null
    int b_block_tiles;            // 128-row tiles per block
    int S, logS, SM, SN, tri, nsuper;
    // tri == 2: staircase (row map with a host copy): supertile row si holds sprefix[si+1] - sprefix[si]
    // live supertiles, its leftmost ones; only those are enumerated
    int sprefix[DMA_MAX_SM + 1];
    unsigned long long* stamps;   // diagnostic build only (dbg & 16): per-tile phase clocks
    int dbg;   // timing-only ablations (probe instantiation): 1 no DMA in the loop, 2 no barrier/waits, 8 no epilogue
    // persistent form: work counters of this launch (one per XCD) and the number of virtual blocks
    struct PersistSlot* slot;
    int nblocks;
};

// Work counters of one persistent launch: workgroups on XCD x draw block numbers 8 w + x from ctr[x]; the last
// workgroup to finish puts the slot back to zero, so a slot is ready for the launch that draws it next.
struct PersistSlot {
    unsigned ctr[8];
    unsigned done;
    unsigned pad[7];
};

// PT: GemmDmaDev, in the generic or in the constant (kernel argument) address space
template <class PT>
__device__ __forceinline__ bool dma_block_to_tile(const PT& p, int b, int& ti, int& tj) {
    return plan_block_to_tile(p, b, ti, tj);
}

__device__ __forceinline__ bool dma_map_tile(const GemmDmaDev& p, int& ti, int& tj) {
    return dma_block_to_tile(p, blockIdx.x, ti, tj);
}

constexpr int DMA_TM = PLAN_TILE, DMA_TN = PLAN_TILE;
// a tile the mode of the launch leaves untouched (above the diagonal, right of its row band)
template <class PT>
__device__ __forceinline__ bool dma_tile_live(const PT& p, int ti, int tj) {
    return plan_tile_live(p, ti, tj, p.row_ncols);
}
constexpr int DMA_STAGE_SLOTS = (DMA_TM + DMA_TN) * 8;     // 16-byte slots per stage (A then B)
constexpr int DMA_STAGES = 3;

// MI = MFMA row tiles per wave: 4 -> 4 waves of 64 x 64 (256 threads, one wave per SIMD),
//                               2 -> 8 waves of 32 x 64 (512 threads, two waves per SIMD: the
//                               matrix pipe is fed at its full 64-cycle cadence, which a single
//                               wave does not reach on fp64 -- measured ~72 cycles per MFMA)
// TICKET: the workgroup stays and draws block numbers from the launch's per-XCD counters (PersistSlot) instead of
// computing the one block blockIdx names -- see gemm_nt_dma_ticket_kernel below.  The tile code is the same
// instruction for instruction, so a ticket launch produces the bits of a per-tile launch.
constexpr unsigned TICKET_NONE = 0xFFFFFFFFu;

// next block number for a workgroup whose blocks run on XCD group `xcd`: from its own group's counter while that lasts
// (the supertile -> L2 map of the per-tile launch), then from the other groups' (the tail of a launch: an XCD that
// runs out early takes work from the slower ones instead of idling -- the hardware's static block -> XCD deal cannot)
template <class PT>
__device__ __forceinline__ unsigned ticket_draw(const PT& p, int xcd) {
    const unsigned per = (unsigned)p.nblocks >> 3;
#pragma unroll 1
    for (int t = 0; t < 8; ++t) {
        const int x = (xcd + t) & 7;
        if (t && __hip_atomic_load(&p.slot->ctr[x], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= per) continue;
        const unsigned w = atomicAdd(&p.slot->ctr[x], 1u);
        if (w < per) return w * 8u + (unsigned)x;
    }
    return TICKET_NONE;
}

template <int MI, bool DBG, bool TICKET = false>
__device__ __forceinline__ void gemm_nt_dma_body(const GemmDmaDev& p) {
    constexpr int NWAVES = 16 / MI;                 // 4 or 8
    constexpr int DPW = 32 / NWAVES;                // DMA wave-instructions per wave per K step (8 or 4)
    constexpr int RPW = DMA_TM / NWAVES;            // operand rows a wave moves per K step (32 or 16)
    constexpr int NFR = MI + 4;                     // fragment reads per half step
    const int dbg = DBG ? p.dbg : 0;
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    const int tid = threadIdx.x;
    // ticket form: mailbox of two ints behind the ring (one barrier per tile: the writer alternates between them)
    volatile GPMI_LDS int* mbox = (volatile GPMI_LDS int*)(smem_raw + DMA_STAGES * DMA_STAGE_SLOTS * 16);
    unsigned blk = blockIdx.x, nxt_blk = TICKET_NONE;
    int par = 0;
    if constexpr (TICKET) {
        if (tid == 0) mbox[0] = (int)ticket_draw(p, (int)(blockIdx.x & 7));
        __syncthreads();
        blk = (unsigned)__builtin_amdgcn_readfirstlane(mbox[0]);
        par = 1;
    }
  for (;;) {
    int ti = 0, tj = 0;
    bool live;
    if constexpr (TICKET) {
        if (blk == TICKET_NONE) break;
        live = dma_block_to_tile(p, (int)blk, ti, tj) && dma_tile_live(p, ti, tj);
        live = __builtin_amdgcn_readfirstlane(live ? 1 : 0) != 0;
        ti = __builtin_amdgcn_readfirstlane(ti);
        tj = __builtin_amdgcn_readfirstlane(tj);
    } else {
        live = dma_map_tile(p, ti, tj) && dma_tile_live(p, ti, tj);
        if (!live) return;
    }
   if (live) {
    unsigned long long st0 = 0, st1 = 0, st2 = 0, st3 = 0;
    if (DBG && (dbg & 16)) st0 = clock64();
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = (wave >> 1) * (16 * MI);
    const int wc = (wave & 1) * 64;
    const int fr = lane & 15;
    const int fg = lane >> 4;

    // ---- DMA source pointers: wave w moves rows 32w..32w+31 of the A tile and of the B tile,
    // 8 rows (= 64 slots = 1 KiB of LDS) per instruction; lane l -> row 8i + (l>>3), k-pair (l&7)^(row&7)
    const int drow = lane >> 3;
    const int dkp = (lane & 7) ^ (drow & 7);
    const double* a_src = p.A + ((int64_t)ti * DMA_TM + RPW * wave + drow) * p.lda + dkp * 2;
    int64_t b_row0 = (int64_t)tj * DMA_TN * p.ldb;
    if (p.b_block_off) {                 // wave-uniform: one scalar load
        const int blk = tj / p.b_block_tiles;
        b_row0 = p.b_block_off[blk] + (int64_t)(tj - blk * p.b_block_tiles) * DMA_TN * p.ldb;
    }
    const double* b_src = p.B + b_row0 + (int64_t)(RPW * wave + drow) * p.ldb + dkp * 2;
    const int64_t a_step = 8 * p.lda, b_step = 8 * p.ldb;
    GPMI_LDS char* lds = (GPMI_LDS char*)smem_raw;
    const int a_dst = (RPW * wave) * 128;                        // byte offset inside a stage
    const int b_dst = DMA_TM * 128 + (RPW * wave) * 128;

    // DMA wave-instruction q (0..DPW/2-1: A rows, DPW/2..DPW-1: B rows) of K step `chunk`
    auto issue_dma_one = [&](int chunk, int q) {
        const int st = chunk % DMA_STAGES;
        GPMI_LDS char* base = lds + st * (DMA_STAGE_SLOTS * 16);
        const int k0 = chunk * 16;
        if (q < DPW / 2)
            __builtin_amdgcn_global_load_lds((const GPMI_GLB void*)(a_src + q * a_step + k0),
                                             (GPMI_LDS void*)(base + a_dst + q * 1024), 16, 0, 0);
        else
            __builtin_amdgcn_global_load_lds((const GPMI_GLB void*)(b_src + (q - DPW / 2) * b_step + k0),
                                             (GPMI_LDS void*)(base + b_dst + (q - DPW / 2) * 1024), 16, 0, 0);
    };
    auto issue_dma = [&](int chunk) {
#pragma unroll
        for (int q = 0; q < DPW; ++q) issue_dma_one(chunk, q);
    };

    // ---- fragment read offsets (16-byte slots): row*8 + (kp ^ (row&7)); rows of a fragment
    // differ from fr by multiples of 16, so row&7 == fr&7
    const int x7 = fr & 7;
    const int sl0 = fr * 8 + (fg ^ x7);            // k-pairs 0..3  (half step 0)
    const int sl1 = fr * 8 + ((4 + fg) ^ x7);      // k-pairs 4..7  (half step 1)
    const d2* smem = reinterpret_cast<const d2*>(smem_raw);
    // fragment read q (0..MI-1: A fragments, MI..MI+3: B fragments)
    auto read_one = [&](int chunk, int half, int q, d2 (&fa)[MI], d2 (&fb)[4]) {
        const d2* sa = smem + (chunk % DMA_STAGES) * DMA_STAGE_SLOTS;
        const d2* sb = sa + DMA_TM * 8;
        const int sl = half ? sl1 : sl0;
        if (q < MI) fa[q] = sa[(wr + 16 * q) * 8 + sl];
        else fb[q - MI] = sb[(wc + 16 * (q - MI)) * 8 + sl];
    };
    auto read_frags = [&](int chunk, int half, d2 (&fa)[MI], d2 (&fb)[4]) {
#pragma unroll
        for (int q = 0; q < NFR; ++q) read_one(chunk, half, q, fa, fb);
    };

    d4 acc[MI][4];
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = d4{0., 0., 0., 0.};
    const int nch = p.nchunks;
    d2 fa0[MI], fb0[4], fa1[MI], fb1[4];
    // prologue: stages 0 and 1 in flight, stage 0 landed and published, F0 loaded
    issue_dma(0);
    if (nch > 1) {
        issue_dma(1);
        if constexpr (DPW == 8) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    } else {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __builtin_amdgcn_s_barrier();
    read_frags(0, 0, fa0, fb0);
    if (DBG && (dbg & 16)) st1 = clock64();

#define GPMI_MFMA_X(FA, FB, T) acc[(T) >> 2][(T) & 3] = __builtin_amdgcn_mfma_f64_16x16x4f64(FA[(T) >> 2].x, FB[(T) & 3].x, acc[(T) >> 2][(T) & 3], 0, 0, 0)
#define GPMI_MFMA_Y(FA, FB, T) acc[(T) >> 2][(T) & 3] = __builtin_amdgcn_mfma_f64_16x16x4f64(FA[(T) >> 2].y, FB[(T) & 3].y, acc[(T) >> 2][(T) & 3], 0, 0, 0)
#define GPMI_FENCE() __builtin_amdgcn_sched_barrier(0)
    constexpr int NT = 4 * MI;                      // MFMAs per quarter step (16 or 8)
    constexpr int NSLOT = NT / 2;                   // one side instruction behind every second MFMA

    // Every non-MFMA instruction of the step sits alone behind two MFMAs (128 matrix-pipe
    // cycles of cover for an LDS-DMA issue of ~60 cycles or a ds_read of ~16):
    //   quarter 0 (F0 .x): the fragment reads F1 <- stage c, k-pairs 4..7
    //   quarter 1 (F0 .y): the DMA wave-instructions of step c+2
    //   wait + barrier (publishes stage c+1, retires the reads of the stage DMA(c+3) overwrites)
    //   quarter 2 (F1 .x): the fragment reads F0 <- stage c+1, k-pairs 0..3
    //   quarter 3 (F1 .y): matrix pipe only
    // so every fragment is read a full quarter step (>= 512 matrix-pipe cycles) before its first use
    // and neither the barrier's lgkmcnt(0) nor the next step's first MFMA waits on LDS latency.
    for (int c = 0; c < nch; ++c) {
        const bool more2 = (c + 2 < nch) && !(dbg & 1);
        const bool more1 = (c + 1 < nch);
        // quarter 0 (F0 .x): fragment reads F1 <- stage c, k-pairs 4..7 (a full quarter ahead of their use)
#pragma unroll
        for (int q = 0; q < NSLOT; ++q) {
            GPMI_MFMA_X(fa0, fb0, 2 * q);
            GPMI_MFMA_X(fa0, fb0, 2 * q + 1);
            GPMI_FENCE();
            if (q < NFR) read_one(c, 1, q, fa1, fb1);
            if (NSLOT < NFR && q + NSLOT < NFR) read_one(c, 1, q + NSLOT, fa1, fb1);
            GPMI_FENCE();
        }
        // quarter 1 (F0 .y): the DMA wave-instructions of step c+2
#pragma unroll
        for (int q = 0; q < NSLOT; ++q) {
            GPMI_MFMA_Y(fa0, fb0, 2 * q);
            GPMI_MFMA_Y(fa0, fb0, 2 * q + 1);
            GPMI_FENCE();
            if (more2 && q < DPW) issue_dma_one(c + 2, q);
            if (more2 && NSLOT < DPW && q + NSLOT < DPW) issue_dma_one(c + 2, q + NSLOT);
            GPMI_FENCE();
        }
        if (!(dbg & 2)) {
            if (more2) {
                if constexpr (DPW == 8) asm volatile("s_waitcnt vmcnt(8)\n\ts_waitcnt lgkmcnt(0)" ::: "memory");
                else asm volatile("s_waitcnt vmcnt(4)\n\ts_waitcnt lgkmcnt(0)" ::: "memory");
            } else {
                asm volatile("s_waitcnt vmcnt(0)\n\ts_waitcnt lgkmcnt(0)" ::: "memory");
            }
            __builtin_amdgcn_s_barrier();
        }
        GPMI_FENCE();
        // quarter 2 (F1 .x): fragment reads F0 <- stage c+1, k-pairs 0..3 (published by the barrier above)
#pragma unroll
        for (int q = 0; q < NSLOT; ++q) {
            GPMI_MFMA_X(fa1, fb1, 2 * q);
            GPMI_MFMA_X(fa1, fb1, 2 * q + 1);
            GPMI_FENCE();
            if (more1 && q < NFR) read_one(c + 1, 0, q, fa0, fb0);
            if (more1 && NSLOT < NFR && q + NSLOT < NFR) read_one(c + 1, 0, q + NSLOT, fa0, fb0);
            GPMI_FENCE();
        }
        // quarter 3 (F1 .y): matrix pipe only
#pragma unroll
        for (int t = 0; t < NT; ++t) GPMI_MFMA_Y(fa1, fb1, t);
        GPMI_FENCE();
    }
#undef GPMI_MFMA_X
#undef GPMI_MFMA_Y
#undef GPMI_FENCE

    if (DBG && (dbg & 16)) st2 = clock64();
    // ticket form: the successor's number is requested here, in front of the C loads, and arrives with them
    if constexpr (TICKET) {
        if (tid == 0) nxt_blk = ticket_draw(p, (int)(blockIdx.x & 7));
    }
    // epilogue: C -= acc in 16-row bands (nothing else runs on this SIMD, so it must not
    // cost a memory round trip per band)
    double* Cg = p.C + ((int64_t)ti * DMA_TM + wr) * p.ldc + (int64_t)tj * DMA_TN + wc;
    auto c_ptr = [&](int i, int j, int v) { return Cg + (int64_t)(16 * i + 4 * v + fg) * p.ldc + 16 * j + fr; };
    if (DBG && (dbg & 8)) {
        double t = 0.;
#pragma unroll
        for (int i = 0; i < MI; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) t += acc[i][j][0] + acc[i][j][1] + acc[i][j][2] + acc[i][j][3];
        if (t == 123.456) Cg[0] = t;
        return;
    }
    // up to three bands of loads in flight (the fragment registers are dead by now), so
    // the whole read-modify-write costs about one memory round trip plus issue time
    constexpr int NB3 = MI < 3 ? MI : 3;
    double cv[NB3][4][4];
#pragma unroll
    for (int i = 0; i < NB3; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int v = 0; v < 4; ++v) cv[i][j][v] = *c_ptr(i, j, v);
    if (DBG && (dbg & 16)) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        st3 = clock64();
    }
#pragma unroll
    for (int i = 0; i < MI; ++i) {
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int v = 0; v < 4; ++v) *c_ptr(i, j, v) = cv[i % NB3][j][v] - acc[i][j][v];
        if (i == 0 && MI > 3) {
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int v = 0; v < 4; ++v) cv[0][j][v] = *c_ptr(3, j, v);
        }
    }
    if (DBG && (dbg & 16)) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        const unsigned long long st4 = clock64();
        if (lane == 0 && blockIdx.x < 4096) {
            unsigned long long* o = p.stamps + ((size_t)blockIdx.x * 4 + (wave & 3)) * 4;
            o[0] = st1 - st0; o[1] = st2 - st1; o[2] = st3 - st2; o[3] = st4 - st3;
        }
    }
   }  // live
    if constexpr (!TICKET) {
        return;
    } else {
        // hand the successor's number to the workgroup.  The barrier also separates this tile's last LDS reads from
        // the next tile's first DMA writes (all of a tile's fragment reads are issued before its last in-loop barrier,
        // so nothing is pending here anyway).
        if (tid == 0) {
            if (!live) nxt_blk = ticket_draw(p, (int)(blockIdx.x & 7));
            mbox[par] = (int)nxt_blk;
        }
        __syncthreads();
        blk = (unsigned)__builtin_amdgcn_readfirstlane(mbox[par]);
        par ^= 1;
    }
  }  // tiles
    if constexpr (TICKET) {
        // the last workgroup out puts the counters back (as the persistent form below)
        if (tid == 0) {
            __threadfence();
            const unsigned old = atomicAdd(&p.slot->done, 1u);
            if (old == gridDim.x - 1) {
#pragma unroll
                for (int x = 0; x < 8; ++x) p.slot->ctr[x] = 0;
                p.slot->done = 0;
                __threadfence();
            }
        }
    }
}

template <int MI, bool DBG>
__global__ __launch_bounds__(1024 / MI, (MI == 4) ? 2 : 3) void gemm_nt_dma_kernel(const GemmDmaDev p) {
    gemm_nt_dma_body<MI, DBG>(p);
}

// The same code under its own symbol for the Cholesky trailing update (GemmArgs::role == 1), so
// that a kernel trace lists those launches apart from the in-panel and solve-sweep updates.
__global__ __launch_bounds__(512, 3) void chol_trailing_update_dma_kernel(const GemmDmaDev p) {
    gemm_nt_dma_body<2, false>(p);
}

// ---------------------------------------------------------------------------
// Ticket form of the 8-wave kernel: gridDim.x = (CUs per XCD - r) x 8 workgroups (one per CU: 96 KiB of LDS each)
// stay for the whole launch and draw their tiles from the launch's per-XCD counters.  Unlike the persistent form
// below it keeps NOTHING across tiles -- same registers as the per-tile kernel (the C tile is not prefetched), so
// the small-LDS panel kernels of the other stream still fit on a CU beside it -- and what it buys is placement:
//   * r CUs per XCD are never touched by the launch (`reserve`): a panel kernel that needs a whole CU (potrf128:
//     2 x 123 registers per SIMD lane) finds one at once instead of waiting for the update's next round boundary;
//   * an XCD whose own blocks are used up draws from the others' counters, so the tail of a launch is shared by
//     all CUs, not by the CUs of the XCD the static deal left with the most work.
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(512, 3) void gemm_nt_dma_ticket_kernel(const GemmDmaDev p) {
    gemm_nt_dma_body<2, false, true>(p);
}
__global__ __launch_bounds__(512, 3) void chol_trailing_update_ticket_kernel(const GemmDmaDev p) {
    gemm_nt_dma_body<2, false, true>(p);
}

// ---------------------------------------------------------------------------
// Persistent form of the 8-wave kernel: the K loops of consecutive tiles form ONE stream.
//
// A workgroup of the kernel above spends, per tile, ~2.5k cycles filling the ring before the first MFMA and ~14k
// cycles after the last one on the read-modify-write of C (one exposed memory round trip under load, then the
// stores): 2.8 % of a tile at K = 2048, 5.6 % at K = 1024, 11 % at K = 512 -- with one workgroup per CU (96 KiB
// of LDS) nothing else covers it.  Here gridDim.x = #CUs workgroups stay resident and draw tiles from a counter:
//   * the LDS-DMA of the NEXT tile's first two K steps is issued during the last two steps of the current tile
//     (the ring never drains, the first MFMA of a tile follows the last of its predecessor);
//   * the C tile is requested three steps before the end into registers that are free in the loop, so the
//     epilogue is 32 subtractions and 32 stores;
//   * tiles are drawn per XCD (block number 8 w + x on XCD x, w from an atomic counter) in the order the hardware
//     would have dispatched them: the supertile -> XCD map and its L2 reuse are unchanged, and a CU that runs
//     behind simply draws fewer tiles.
// The next block number is fetched inside the loop (atomic by wave 0, mailbox in LDS, three K steps per attempt); a
// workgroup that reaches the end of a tile without a resolved successor drains, resolves one synchronously and
// starts over with a prologue.  vmcnt bookkeeping: the memory counter retires in order, so a wait for "the DMA of
// step c + 1" also waits for everything issued before it -- the C loads sit between the DMA of steps nch - 1 and
// nch, which gives them two K steps to land.
// Used for launches that have the chip to themselves (see launch_gemm_nt_dma): at 216 registers per lane and two waves per
// SIMD nothing else fits on a CU beside a resident workgroup.  ROLE only separates the two kernel symbols.
// ---------------------------------------------------------------------------
template <int ROLE>
__device__ __forceinline__ void gemm_nt_dma_persist_body(const GemmDmaDev& p) {
    constexpr int MI = 2, DPW = 4, RPW = 16, NFR = 6, NT = 8, NSLOT = 4;
    constexpr int RING = DMA_STAGES * DMA_STAGE_SLOTS * 16;
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    volatile GPMI_LDS int* mbox = (volatile GPMI_LDS int*)(smem_raw + RING);
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = (wave >> 1) * (16 * MI);
    const int wc = (wave & 1) * 64;
    const int fr = lane & 15;
    const int fg = lane >> 4;
    const int drow = lane >> 3;
    const int dkp = (lane & 7) ^ (drow & 7);
    const int xcd = blockIdx.x & 7;
    const int nch = p.nchunks;
    const int64_t ldc = p.ldc;
    GPMI_LDS char* lds = (GPMI_LDS char*)smem_raw;
    const int a_dst = (RPW * wave) * 128;
    const int b_dst = DMA_TM * 128 + (RPW * wave) * 128;
    const unsigned c_lane = (unsigned)((fg * ldc + fr) * 8);       // byte offset of this lane inside a 16 x 16 C fragment

    // the launch parameters as the kernel received them (its only argument), behind a pointer the optimiser cannot
    // see through: they are wanted once per tile, and hoisted out of the K loop they would occupy two dozen scalar
    // registers for its whole length
    auto kernarg = [&]() -> const GPMI_CONST GemmDmaDev* {
        const GPMI_CONST GemmDmaDev* pp = (const GPMI_CONST GemmDmaDev*)__builtin_amdgcn_kernarg_segment_ptr();
        asm volatile("" : "+s"(pp));
        return pp;
    };
    // a wave-uniform pointer that the compiler holds in vector registers -> scalar registers
    auto uniform_ptr = [](const void* q) -> uintptr_t {
        const uintptr_t u = reinterpret_cast<uintptr_t>(q);
        const unsigned lo = (unsigned)__builtin_amdgcn_readfirstlane((int)(u & 0xffffffffu));
        const unsigned hi = (unsigned)__builtin_amdgcn_readfirstlane((int)(u >> 32));
        return ((uintptr_t)hi << 32) | lo;
    };
    // A tile as the loop wants it: the four DMA source pointers of this lane (its 8-row pieces of the A rows and of
    // the B rows of the tile, K step 0) and the wave-uniform base of this wave's part of the C tile
    struct TileRef { const double* d[DPW]; GPMI_GLB char* c; };
    // block number -> tile; false: nothing to do for it
    auto resolve = [&](unsigned b, TileRef& t) -> bool {
        const GPMI_CONST GemmDmaDev* pp = kernarg();
        int ti, tj;
        if (!dma_block_to_tile(*pp, (int)b, ti, tj)) return false;
        if (!dma_tile_live(*pp, ti, tj)) return false;
        // wave-uniform, but the map goes through a float square root (vector unit): back to scalar registers
        ti = __builtin_amdgcn_readfirstlane(ti);
        tj = __builtin_amdgcn_readfirstlane(tj);
        t.d[0] = pp->A + ((int64_t)ti * DMA_TM + RPW * wave + drow) * pp->lda + dkp * 2;
        t.d[1] = t.d[0] + 8 * pp->lda;
        int64_t b_row0 = (int64_t)tj * DMA_TN * pp->ldb;
        if (pp->b_block_off) {
            const int blk = tj / pp->b_block_tiles;
            b_row0 = pp->b_block_off[blk] + (int64_t)(tj - blk * pp->b_block_tiles) * DMA_TN * pp->ldb;
        }
        t.d[2] = pp->B + b_row0 + (int64_t)(RPW * wave + drow) * pp->ldb + dkp * 2;
        t.d[3] = t.d[2] + 8 * pp->ldb;
        double* tc = pp->C + ((int64_t)ti * DMA_TM + wr) * pp->ldc + (int64_t)tj * DMA_TN + wc;
        t.c = (GPMI_GLB char*)uniform_ptr(tc);
        return true;
    };
    // the next block number of this workgroup (one lane): from its own XCD group's counter while that lasts, then from
    // the other groups' (ticket_draw: the tail of a launch is shared by all CUs) -- TICKET_NONE when every counter is used up
    auto draw = [&]() -> unsigned {
        const GPMI_CONST GemmDmaDev* pp = kernarg();
        return ticket_draw(*pp, xcd);
    };
    auto past_end = [&](unsigned b) -> bool { return b == TICKET_NONE; };
    // a wave-uniform flag the compiler computed with vector compares -> a scalar one (scalar branches in the loop)
    auto scalar_flag = [](bool f) -> bool { return __builtin_amdgcn_readfirstlane(f ? 1 : 0) != 0; };
    // synchronous draw (start of the kernel, after a drain): false when the XCD's blocks are used up
    auto draw_sync = [&](TileRef& t) -> bool {
        for (;;) {
            if (tid == 0) mbox[0] = (int)draw();
            __syncthreads();
            const unsigned w = (unsigned)__builtin_amdgcn_readfirstlane(mbox[0]);
            __syncthreads();
            if (scalar_flag(past_end(w))) return false;
            if (scalar_flag(resolve(w, t))) return true;
        }
    };
    // DMA wave-instruction q (0..1: A rows, 2..3: B rows) of the K step at element offset k0 into ring stage st
    auto dma_one = [&](const double* const (&d)[DPW], int k0, int st, int q) {
        GPMI_LDS char* base = lds + st * (DMA_STAGE_SLOTS * 16);
        const int dst = (q < DPW / 2) ? a_dst + q * 1024 : b_dst + (q - DPW / 2) * 1024;
        __builtin_amdgcn_global_load_lds((const GPMI_GLB void*)(d[q] + k0), (GPMI_LDS void*)(base + dst), 16, 0, 0);
    };
    // element (16 i + 4 v + fg, 16 j + fr) of this wave's part of the C tile; the row pitch goes through a laundered
    // copy so that the 32 address computations stay where they are used (hoisted, they are 16 scalar registers)
    auto c_addr = [&](GPMI_GLB char* tc, int i, int j, int v) -> GPMI_GLB double* {
        int64_t pitch = ldc;
        asm volatile("" : "+s"(pitch));
        GPMI_GLB char* base = tc + ((int64_t)(16 * i + 4 * v) * pitch + 16 * j) * 8;
        return (GPMI_GLB double*)(base + c_lane);
    };
    const int x7 = fr & 7;
    const int sl0 = fr * 8 + (fg ^ x7);
    const int sl1 = fr * 8 + ((4 + fg) ^ x7);
    const d2* smem = reinterpret_cast<const d2*>(smem_raw);
    auto read_one = [&](int st, int half, int q, d2 (&fa)[MI], d2 (&fb)[4]) {
        const d2* sa = smem + st * DMA_STAGE_SLOTS;
        const d2* sb = sa + DMA_TM * 8;
        const int sl = half ? sl1 : sl0;
        if (q < MI) fa[q] = sa[(wr + 16 * q) * 8 + sl];
        else fb[q - MI] = sb[(wc + 16 * (q - MI)) * 8 + sl];
    };

#define GPMI_MFMA_X(FA, FB, T) acc[(T) >> 2][(T) & 3] = __builtin_amdgcn_mfma_f64_16x16x4f64(FA[(T) >> 2].x, FB[(T) & 3].x, acc[(T) >> 2][(T) & 3], 0, 0, 0)
#define GPMI_MFMA_Y(FA, FB, T) acc[(T) >> 2][(T) & 3] = __builtin_amdgcn_mfma_f64_16x16x4f64(FA[(T) >> 2].y, FB[(T) & 3].y, acc[(T) >> 2][(T) & 3], 0, 0, 0)
#define GPMI_FENCE() __builtin_amdgcn_sched_barrier(0)

    TileRef nxt;                                    // the tile resolved last
    const double* dsrc[DPW];                        // the DMA stream's pointers: two K steps ahead of the matrix pipe
    bool have_cur = draw_sync(nxt);
    bool exhausted = !have_cur;
    while (have_cur) {
        // ---- prologue of a stream: steps 0 and 1 of the tile in flight, step 0 landed and published, F0 loaded
        int s0 = 0, s1 = 1, s2 = 2;                 // ring stages of steps g, g + 1, g + 2
#pragma unroll
        for (int q = 0; q < DPW; ++q) dsrc[q] = nxt.d[q];
#pragma unroll
        for (int q = 0; q < DPW; ++q) dma_one(dsrc, 0, 0, q);
#pragma unroll
        for (int q = 0; q < DPW; ++q) dma_one(dsrc, 16, 1, q);
        int dk = 32;                                // K offset of the DMA stream's next step
        asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        d2 fa0[MI], fb0[4], fa1[MI], fb1[4];
#pragma unroll
        for (int q = 0; q < NFR; ++q) read_one(0, 0, q, fa0, fb0);

        bool stream = true;
        while (stream) {                            // one tile per pass
            GPMI_GLB char* c_cur = nxt.c;
            d4 acc[MI][4];
#pragma unroll
            for (int i = 0; i < MI; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[i][j] = d4{0., 0., 0., 0.};
            double cv[MI][4][4];
            bool have_nxt = false;
            int fw = 0;                             // wave 0, lane 0: the block number drawn for the successor
            const int cpre = nch - 3;               // step whose last quarter requests the C tile
            const int last_draw = nch - 6;          // no draw is started later: its answer is read two steps on
            for (int c = 0; c < nch; ++c) {
                const bool in2 = c + 2 < nch, in1 = c + 1 < nch;
                const bool more2 = in2 || have_nxt;
                const bool more1 = in1 || have_nxt;
                if (c + 2 == nch && have_nxt) {     // the DMA stream moves on to the successor
#pragma unroll
                    for (int q = 0; q < DPW; ++q) dsrc[q] = nxt.d[q];
                    dk = 0;
                }
                // quarter 0 (F0 .x): fragment reads F1 <- stage of step c, k-pairs 4..7
#pragma unroll
                for (int q = 0; q < NSLOT; ++q) {
                    GPMI_MFMA_X(fa0, fb0, 2 * q);
                    GPMI_MFMA_X(fa0, fb0, 2 * q + 1);
                    GPMI_FENCE();
                    read_one(s0, 1, q, fa1, fb1);
                    if (q + NSLOT < NFR) read_one(s0, 1, q + NSLOT, fa1, fb1);
                    GPMI_FENCE();
                }
                // quarter 1 (F0 .y): the DMA of step c + 2 -- of this tile, or of the successor's first steps
#pragma unroll
                for (int q = 0; q < NSLOT; ++q) {
                    GPMI_MFMA_Y(fa0, fb0, 2 * q);
                    GPMI_MFMA_Y(fa0, fb0, 2 * q + 1);
                    GPMI_FENCE();
                    if (more2) dma_one(dsrc, dk, s2, q);
                    GPMI_FENCE();
                }
                dk += 16;
                // the DMA of step c + 1 has landed (everything older than the newest DPW wave-instructions, or than
                // those and the 32 C loads behind it), the fragment reads of this step are in registers
                // (one test on the path every step but three of a tile takes: all waves arrive here together, and
                // what they execute between the last MFMA of quarter 1 and the barrier is matrix-pipe idle time)
                if (__builtin_expect(more2 && c != cpre + 1, 1)) {
                    asm volatile("s_waitcnt vmcnt(4)\n\ts_waitcnt lgkmcnt(0)" ::: "memory");
                } else if (c == cpre + 1) {
                    if (more2) asm volatile("s_waitcnt vmcnt(36)\n\ts_waitcnt lgkmcnt(0)" ::: "memory");
                    else asm volatile("s_waitcnt vmcnt(32)\n\ts_waitcnt lgkmcnt(0)" ::: "memory");
                } else {
                    asm volatile("s_waitcnt vmcnt(0)\n\ts_waitcnt lgkmcnt(0)" ::: "memory");
                }
                __builtin_amdgcn_s_barrier();
                GPMI_FENCE();
                // quarter 2 (F1 .x): fragment reads F0 <- stage of step c + 1, k-pairs 0..3
#pragma unroll
                for (int q = 0; q < NSLOT; ++q) {
                    GPMI_MFMA_X(fa1, fb1, 2 * q);
                    GPMI_MFMA_X(fa1, fb1, 2 * q + 1);
                    GPMI_FENCE();
                    if (more1) {
                        read_one(s1, 0, q, fa0, fb0);
                        if (q + NSLOT < NFR) read_one(s1, 0, q + NSLOT, fa0, fb0);
                    }
                    GPMI_FENCE();
                }
                // quarter 3 (F1 .y): matrix pipe only; behind it, on step nch - 3, the request of the C tile
#pragma unroll
                for (int t = 0; t < NT; ++t) GPMI_MFMA_Y(fa1, fb1, t);
                GPMI_FENCE();
                // (one copy of every MFMA: with the quarter duplicated for this step the register allocator starts
                // copying the accumulators around; the 32 loads in one burst cost the pipe ~300 cycles once per tile)
                if (__builtin_expect(c == cpre, 0)) {
#pragma unroll
                    for (int t = 0; t < NT; ++t)
#pragma unroll
                        for (int v = 0; v < 4; ++v) cv[t >> 2][t & 3][v] = *c_addr(c_cur, t >> 2, t & 3, v);
                }
                GPMI_FENCE();
                // the draw of the successor: atomic, mailbox, resolve on three consecutive steps
                if (__builtin_expect(!have_nxt && !exhausted && c >= 1 && c <= last_draw + 2, 0)) {
                    const int ph = (c - 1) % 3;
                    if (ph == 0 && c <= last_draw) {
                        if (tid == 0) fw = (int)draw();
                    } else if (ph == 1 && c <= last_draw + 1) {
                        if (tid == 0) mbox[0] = fw;
                    } else if (ph == 2) {
                        const unsigned w = (unsigned)__builtin_amdgcn_readfirstlane(mbox[0]);
                        if (scalar_flag(past_end(w))) exhausted = true;
                        else have_nxt = scalar_flag(resolve(w, nxt));
                    }
                }
                GPMI_FENCE();
                const int t3 = s0; s0 = s1; s1 = s2; s2 = t3;
            }
            // ---- epilogue: C -= acc from the registers the C tile was requested into
            if (have_nxt) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
            for (int i = 0; i < MI; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j)
#pragma unroll
                    for (int v = 0; v < 4; ++v) *c_addr(c_cur, i, j, v) = cv[i][j][v] - acc[i][j][v];
            stream = have_nxt;
        }
        have_cur = exhausted ? false : draw_sync(nxt);
        if (!have_cur) exhausted = true;
    }
#undef GPMI_MFMA_X
#undef GPMI_MFMA_Y
#undef GPMI_FENCE
    // the last workgroup out puts the counters back
    if (tid == 0) {
        __threadfence();
        PersistSlot* slot = kernarg()->slot;
        const unsigned old = atomicAdd(&slot->done, 1u);
        if (old == gridDim.x - 1) {
#pragma unroll
            for (int x = 0; x < 8; ++x) slot->ctr[x] = 0;
            slot->done = 0;
            __threadfence();
        }
    }
}

__global__ __launch_bounds__(512, 2) void gemm_nt_dma_persist_kernel(const GemmDmaDev p) {
    gemm_nt_dma_persist_body<0>(p);
}
// the same code under the trailing update's own symbol (GemmArgs::role == 1), as for the kernel above
__global__ __launch_bounds__(512, 2) void chol_trailing_update_persist_kernel(const GemmDmaDev p) {
    gemm_nt_dma_persist_body<1>(p);
}

// ---------------------------------------------------------------------------
// Small launches (panel-internal updates, diagonal blocks, small problems: fewer 64 x 64 tiles than the chip has
// room for): what they cost is the memory latency of every K step, not arithmetic -- the first-generation kernel
// took ~2 us per K step of 16 because only one step was in flight.  Same LDS image and fragment maps as above,
// 64 x 64 tiles, 4 waves of 32 x 32, and a ring of EIGHT stages filled by LDS-DMA before the first MFMA issues:
// K = 128 is one memory round trip, deeper K keeps seven steps in flight.  One barrier per K step: it publishes
// step c (every wave has waited for its own pieces) and frees the stage of step c - 1 for step c + 7.
// ---------------------------------------------------------------------------
// Ring depth: 8 stages (128 KiB) when the launch has the chip to itself; 3 stages (48 KiB) while a trailing
// update runs on the other stream (lookahead) -- its workgroups hold 96 KiB of every CU's 160 KiB of LDS, and a
// panel kernel that does not fit beside them waits for a whole tile (~130 us) instead of starting at once
// (kernel trace: 127 us per launch with the deep ring against 65 us for the 16 KiB first-generation kernel).
constexpr int SM_T = 64;                                   // tile edge
constexpr int SM_STAGE_SLOTS = (SM_T + SM_T) * 8;          // 16-byte slots per stage (A then B): 16 KiB
static thread_local int t_small_shallow = 0;
static thread_local int t_two_streams = 0;
GemmShallowScope::GemmShallowScope(bool on, bool two_streams, bool exact) : prev(t_small_shallow), prev_two(t_two_streams) {
    if (on) t_small_shallow = 1;
    else if (exact) t_small_shallow = 0;
    if (on || two_streams) t_two_streams = 1;
}
GemmShallowScope::~GemmShallowScope() { t_small_shallow = prev; t_two_streams = prev_two; }
bool gemm_shallow_active() { return t_small_shallow != 0; }
bool gemm_two_streams_active() { return t_two_streams != 0; }

struct GemmSmallDev {
    double* C;
    const double* A;
    const double* B;
    int64_t ldc, lda, ldb;
    int Tm, Tn, nchunks;
    int lower;
    int64_t diag_off;
};

template <int SM_STAGES>
__global__ __launch_bounds__(256) void gemm_nt_small_kernel(const GemmSmallDev p) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    const int ti = blockIdx.x / p.Tn, tj = blockIdx.x - ti * p.Tn;
    if (p.lower) {
        const int64_t min_col = (int64_t)tj * SM_T, max_row = (int64_t)ti * SM_T + SM_T - 1;
        if (min_col > max_row + p.diag_off) return;
    }
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = (wave >> 1) * 32, wc = (wave & 1) * 32;
    const int fr = lane & 15, fg = lane >> 4;
    // DMA: wave w moves rows 16 w .. 16 w + 15 of the A tile and of the B tile, 8 rows per instruction
    const int drow = lane >> 3;
    const int dkp = (lane & 7) ^ (drow & 7);
    const double* a_src = p.A + ((int64_t)ti * SM_T + 16 * wave + drow) * p.lda + dkp * 2;
    const double* b_src = p.B + ((int64_t)tj * SM_T + 16 * wave + drow) * p.ldb + dkp * 2;
    GPMI_LDS char* lds = (GPMI_LDS char*)smem_raw;
    auto issue = [&](int chunk) {
        GPMI_LDS char* base = lds + (chunk % SM_STAGES) * (SM_STAGE_SLOTS * 16);
        const int k0 = chunk * 16;
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            __builtin_amdgcn_global_load_lds((const GPMI_GLB void*)(a_src + (int64_t)(8 * q) * p.lda + k0),
                                             (GPMI_LDS void*)(base + (16 * wave + 8 * q) * 128), 16, 0, 0);
            __builtin_amdgcn_global_load_lds((const GPMI_GLB void*)(b_src + (int64_t)(8 * q) * p.ldb + k0),
                                             (GPMI_LDS void*)(base + SM_T * 128 + (16 * wave + 8 * q) * 128), 16, 0, 0);
        }
    };
    const int nch = p.nchunks;
    for (int c = 0; c < SM_STAGES && c < nch; ++c) issue(c);

    const int x7 = fr & 7;
    const int sl0 = fr * 8 + (fg ^ x7), sl1 = fr * 8 + ((4 + fg) ^ x7);
    const d2* smem = reinterpret_cast<const d2*>(smem_raw);
    d4 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = d4{0., 0., 0., 0.};

    for (int c = 0; c < nch; ++c) {
        // pieces of steps c + 1 .. may still fly: 4 per step in flight behind step c (at most SM_STAGES - 1 steps)
        const int behind = min(nch - 1 - c, SM_STAGES - 1 - (c > 0 ? 1 : 0));
        // s_waitcnt takes an immediate: the steady state (first step / later steps) and the drain
        if constexpr (SM_STAGES == 8) {
            if (behind >= 7) asm volatile("s_waitcnt vmcnt(28)" ::: "memory");
            else if (behind == 6) asm volatile("s_waitcnt vmcnt(24)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        } else {
            static_assert(SM_STAGES == 3, "ring depth 8 or 3");
            if (behind >= 2) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
            else if (behind == 1) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        __builtin_amdgcn_s_barrier();
        if (c > 0 && c - 1 + SM_STAGES < nch) issue(c - 1 + SM_STAGES);      // the stage step c - 1 has just left
        const d2* sa = smem + (c % SM_STAGES) * SM_STAGE_SLOTS;
        const d2* sb = sa + SM_T * 8;
        d2 fa0[2], fa1[2], fb0[2], fb1[2];
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            fa0[i] = sa[(wr + 16 * i) * 8 + sl0];
            fa1[i] = sa[(wr + 16 * i) * 8 + sl1];
            fb0[i] = sb[(wc + 16 * i) * 8 + sl0];
            fb1[i] = sb[(wc + 16 * i) * 8 + sl1];
        }
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                acc[i][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(fa0[i].x, fb0[j].x, acc[i][j], 0, 0, 0);
                acc[i][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(fa0[i].y, fb0[j].y, acc[i][j], 0, 0, 0);
                acc[i][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(fa1[i].x, fb1[j].x, acc[i][j], 0, 0, 0);
                acc[i][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(fa1[i].y, fb1[j].y, acc[i][j], 0, 0, 0);
            }
    }
    // C -= acc: all loads of the tile first, then the stores
    double* Cg = p.C + ((int64_t)ti * SM_T + wr) * p.ldc + (int64_t)tj * SM_T + wc;
    auto c_ptr = [&](int i, int j, int v) { return Cg + (int64_t)(16 * i + 4 * v + fg) * p.ldc + 16 * j + fr; };
    double cv[2][2][4];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int v = 0; v < 4; ++v) cv[i][j][v] = *c_ptr(i, j, v);
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int v = 0; v < 4; ++v) *c_ptr(i, j, v) = cv[i][j][v] - acc[i][j][v];
}

bool gemm_small_eligible(const GemmArgs& a) {
    return a.mode == 0 && a.M % SM_T == 0 && a.N % SM_T == 0 && a.K % 16 == 0 && a.K >= 16 && !a.row_ncols && !a.b_block_off;
}

hipError_t launch_gemm_nt_small(hipStream_t s, const GemmArgs& a) {
    GemmSmallDev p;
    p.C = a.C; p.A = a.A; p.B = a.B; p.ldc = a.ldc; p.lda = a.lda; p.ldb = a.ldb;
    p.Tm = (int)(a.M / SM_T); p.Tn = (int)(a.N / SM_T); p.nchunks = (int)(a.K / 16);
    p.lower = a.lower; p.diag_off = a.diag_off;
    constexpr size_t lds8 = (size_t)8 * SM_STAGE_SLOTS * 16, lds3 = (size_t)3 * SM_STAGE_SLOTS * 16;
    static PerDeviceOnce once;
    const hipError_t ea = once.run([&]() -> hipError_t {
        return hipFuncSetAttribute((const void*)gemm_nt_small_kernel<8>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds8);
    });
    if (ea != hipSuccess) return ea;
    if (t_small_shallow) hipLaunchKernelGGL(gemm_nt_small_kernel<3>, dim3((unsigned)(p.Tm * p.Tn)), dim3(256), lds3, s, p);
    else hipLaunchKernelGGL(gemm_nt_small_kernel<8>, dim3((unsigned)(p.Tm * p.Tn)), dim3(256), lds8, s, p);
    return hipGetLastError();
}

// Counter slots of the persistent launches, one pool per device: a launch takes the next slot of a ring (two launches
// that run at the same time on different streams never share one), a slot is zero when it is handed out (allocated
// zeroed, put back to zero by the last workgroup of the launch that used it).
struct PersistPool {
    static constexpr int SLOTS = 16384;
    PersistSlot* base = nullptr;
    int groups = 0;                  // workgroups of a persistent launch: one per CU, a multiple of the 8 XCDs
    std::atomic<unsigned> head{0};
    PersistSlot* next() { return base + (head.fetch_add(1, std::memory_order_relaxed) % SLOTS); }
};

static PersistPool* persist_pool() {
    static std::mutex mu;
    static PersistPool pools[16];
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 16) return nullptr;
    PersistPool& pl = pools[dev];
    std::lock_guard<std::mutex> lock(mu);
    if (!pl.base) {
        hipDeviceProp_t prop;
        if (hipGetDeviceProperties(&prop, dev) != hipSuccess) return nullptr;
        void* q = nullptr;
        if (hipMalloc(&q, sizeof(PersistSlot) * PersistPool::SLOTS) != hipSuccess) return nullptr;
        if (hipMemset(q, 0, sizeof(PersistSlot) * PersistPool::SLOTS) != hipSuccess) { (void)hipFree(q); return nullptr; }
        pl.groups = (prop.multiProcessorCount / 8) * 8;
        if (pl.groups < 8) { (void)hipFree(q); return nullptr; }
        const void* fns[] = {(const void*)gemm_nt_dma_persist_kernel, (const void*)chol_trailing_update_persist_kernel,
                             (const void*)gemm_nt_dma_ticket_kernel, (const void*)chol_trailing_update_ticket_kernel};
        for (const void* f : fns)
            if (hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize,
                                    (int)(DMA_STAGES * DMA_STAGE_SLOTS * 16 + 16)) != hipSuccess) { (void)hipFree(q); return nullptr; }
        pl.base = static_cast<PersistSlot*>(q);
    }
    return &pl;
}

bool gemm_dma_eligible(const GemmArgs& a) {
    return a.mode == 0 && a.N % 128 == 0 && a.M % 128 == 0 && a.K % 16 == 0 && a.K >= 32;
}

hipError_t launch_gemm_nt_dma(hipStream_t s, const GemmArgs& a) {
    GemmDmaDev p;
    p.C = a.C; p.A = a.A; p.B = a.B;
    p.ldc = a.ldc; p.lda = a.lda; p.ldb = a.ldb;
    p.Tm = (int)(a.M / 128); p.Tn = (int)(a.N / 128);
    p.nchunks = (int)(a.K / 16);
    p.lower = a.lower; p.diag_off = a.diag_off;
    p.row_ncols = a.row_ncols; p.row_block_tiles = a.row_block_tiles > 0 ? a.row_block_tiles : 1;
    p.b_block_off = a.b_block_off;
    p.b_block_tiles = a.b_block_off ? (int)(a.b_block_rows / 128) : 1;
    if (a.b_block_off && (a.b_block_rows <= 0 || a.b_block_rows % 128)) return hipErrorInvalidValue;
    const Tuning& tn = tuning();
    TilePlan plan;
    // the launch geometry: first as a resident form wants it (widest supertiles), which also decides whether one is used
    if (!plan_tiles(plan, p.Tm, p.Tn, a.lower, a.diag_off, a.row_ncols != nullptr, a.row_ncols_host, a.row_bands,
                    p.row_block_tiles, 0, false))
        return hipErrorInvalidValue;
    if (plan.nsuper == 0) return hipSuccess;
    const bool eight = tn.gemm_dma_waves == 8 && !(tn.gemm_dbg & 0xff);
    const bool want_ticket = eight && (tn.gemm_ticket >= 2 || (tn.gemm_ticket == 1 && a.role == 1 && gemm_two_streams_active()));
    const bool want_persist = eight && tn.gemm_persist && a.K >= 256 && !gemm_two_streams_active();
    PersistPool* pool = (want_ticket || want_persist) ? persist_pool() : nullptr;
    const bool ticket = want_ticket && pool && plan.nblocks >= pool->groups;
    const bool persist = !ticket && want_persist && pool && plan.nblocks >= 2 * pool->groups;
    // one workgroup per tile: the supertile edge is chosen with the static deal of blocks to the XCDs in mind
    if (!ticket && !persist &&
        !plan_tiles(plan, p.Tm, p.Tn, a.lower, a.diag_off, a.row_ncols != nullptr, a.row_ncols_host, a.row_bands,
                    p.row_block_tiles, 0, tn.gemm_balance != 0))
        return hipErrorInvalidValue;
    p.S = plan.S; p.logS = plan.logS; p.SM = plan.SM; p.SN = plan.SN; p.tri = plan.tri; p.nsuper = plan.nsuper;
    if (plan.tri == 2) std::copy(plan.sprefix, plan.sprefix + plan.SM + 1, p.sprefix);
    const int nblocks = plan.nblocks;
    constexpr size_t lds = (size_t)DMA_STAGES * DMA_STAGE_SLOTS * 16;
    static PerDeviceOnce once;
    const hipError_t ea = once.run([&]() -> hipError_t {
        const void* fns[] = {(const void*)gemm_nt_dma_kernel<4, false>, (const void*)gemm_nt_dma_kernel<4, true>,
                             (const void*)gemm_nt_dma_kernel<2, false>, (const void*)gemm_nt_dma_kernel<2, true>,
                             (const void*)chol_trailing_update_dma_kernel};
        for (const void* f : fns) {
            const hipError_t e = hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            if (e != hipSuccess) return e;
        }
        return hipSuccess;
    });
    if (ea != hipSuccess) return ea;
    p.dbg = tn.gemm_dbg & 0xff;
    p.stamps = tn.gemm_stamps;
    p.slot = nullptr;
    p.nblocks = nblocks;
    // ticket form (option gemm_ticket: 1 for the Cholesky's trailing updates while two streams are busy, 2 for every
    // launch of at least one round; gemm_reserve CUs per XCD stay untouched)
    if (ticket) {
        const int reserve = std::max(0, std::min(tn.gemm_reserve, pool->groups / 8 - 1));
        p.slot = pool->next();
        const int groups = pool->groups - 8 * reserve;
        constexpr size_t ldst = lds + 16;          // ring + mailbox
        if (a.role == 1) hipLaunchKernelGGL(chol_trailing_update_ticket_kernel, dim3(groups), dim3(512), ldst, s, p);
        else hipLaunchKernelGGL(gemm_nt_dma_ticket_kernel, dim3(groups), dim3(512), ldst, s, p);
        return hipGetLastError();
    }
    // persistent form: launches with at least two rounds of tiles and a K loop long enough to draw the successor in --
    // and the chip to themselves: resident workgroups (216 registers per lane, two waves per SIMD) leave no room on a
    // CU for the panel kernels of the other stream, which would then wait for the whole launch instead of a tile
    // (lookahead with both forms: N = 16384 fit + predict 39.8 against 42.9 ms).  No pool (its allocation or the opt-in
    // failed, or an unusual device): the per-tile launch below computes the same bits.
    if (persist) {
        p.slot = pool->next();
        constexpr size_t ldsp = lds + 16;          // ring + mailbox
        if (a.role == 1) hipLaunchKernelGGL(chol_trailing_update_persist_kernel, dim3(pool->groups), dim3(512), ldsp, s, p);
        else hipLaunchKernelGGL(gemm_nt_dma_persist_kernel, dim3(pool->groups), dim3(512), ldsp, s, p);
        return hipGetLastError();
    }
    if (tn.gemm_dma_waves == 8) {
        if (p.dbg) hipLaunchKernelGGL((gemm_nt_dma_kernel<2, true>), dim3(nblocks), dim3(512), lds, s, p);
        else if (a.role == 1) hipLaunchKernelGGL(chol_trailing_update_dma_kernel, dim3(nblocks), dim3(512), lds, s, p);
        else hipLaunchKernelGGL((gemm_nt_dma_kernel<2, false>), dim3(nblocks), dim3(512), lds, s, p);
    } else {
        if (p.dbg) hipLaunchKernelGGL((gemm_nt_dma_kernel<4, true>), dim3(nblocks), dim3(256), lds, s, p);
        else hipLaunchKernelGGL((gemm_nt_dma_kernel<4, false>), dim3(nblocks), dim3(256), lds, s, p);
    }
    return hipGetLastError();
}

}  // namespace gpmi
