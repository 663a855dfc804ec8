// Second-generation panel kernels: the latency chain of a block step on the matrix pipe.
//
//   potrf128_kernel   Cholesky of a 128 x 128 diagonal block by ONE workgroup (8 waves).
//   trsm128_kernel    X (m x 128) <- X * L^-T for the rows below it, 16 rows per wave, chip-wide.
//
// They replace the chains potf2_64 -> trsm_rlt64 -> rank-64 GEMM (two of each per 128 columns, all
// latency-bound launches; the substitution kernels were fp64 VALU) behind np.linalg.cholesky and the
// LU-based np.linalg.solve of the reference (GP_regression.py:138-139, 144).
//
// Everything except the 16 x 16 diagonal blocks runs on v_mfma_f64_16x16x4_f64.  The 16 x 16 blocks are
// factored AND inverted by one wave with the rows in registers (v_readlane broadcasts, no LDS, no barrier);
// with W_jj = L_jj^-1 at hand, "rows below" is a matrix product  L_ij = S_ij * W_jj^T  and so is every update.
// Only 16 x 16 inverses are ever formed (cond(L_jj) of a 16 x 16 diagonal block, never of the panel).
//
// Register layout of a 16 x 16 tile X (rows n, columns c) -- "X layout": lane (fr = lane & 15, fg = lane >> 4)
// holds x[v] = X[fr][kap(fg, v)],  kap(g, v) = 2g + (v & 1) + 8 (v >> 1): two 16-byte pieces of row fr.
// This IS the B-operand layout of the MFMA up to a permutation of the summation index (which both
// operands share), and -- with the rows of the A operand permuted by rho(g + 4v) = kap(g, v) -- also the
// layout the MFMA writes its result in.  So a tile that has just been produced feeds the next products
// straight from its accumulator registers: no shuffles, no LDS round trip for X.
//   Y[c][n] = sum_k M[c][k] X[n][k]:   acc = mfma(a_v, x[v], acc), v = 0..3,
//   a_v(lane) = M[rho(lane & 15)][kap(lane >> 4, v)],   new x[v] = acc[v].
// The A operands (W_jj, -L_kj) live in LDS as 16 rows x 128 bytes, row i' = M[rho(i')][:], 16-byte slots
// XOR-swizzled by (i' >> 1) & 7: both ds_read_b128 of a fragment are conflict-free (brute-force checked).
#include "gpmi_internal.h"
#include <utility>

namespace gpmi {

typedef double d4 __attribute__((ext_vector_type(4)));
typedef double d2 __attribute__((ext_vector_type(2)));

namespace {

constexpr int PB = 128;                 // panel block width
constexpr int NT = PB / 16;             // 16 x 16 tiles per side
constexpr int TILE_BYTES = 16 * 128;
constexpr int NTILES = NT * (NT + 1) / 2;
constexpr int POTRF_TILES = NT + 1;     // potrf128 keeps only the current step's operand tiles in LDS (18 KiB)
constexpr int TRSM_ROWS = 64;            // rows per trsm128 workgroup pass (4 waves x 16)
constexpr int SCR_LD = 18;              // doubles per scratch row (16-byte aligned rows, conflict-free row reads)

__device__ __forceinline__ int kap(int g, int v) { return 2 * g + (v & 1) + 8 * (v >> 1); }
// LDS row that holds matrix row n: rho^-1(n)
__device__ __forceinline__ int rho_inv(int n) { return ((n >> 1) & 3) + 4 * (n & 1) + 8 * (n >> 3); }

__device__ __forceinline__ double readlane_d(double v, int lane) {
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_readlane(lo, lane);
    hi = __builtin_amdgcn_readlane(hi, lane);
    return __hiloint2double(hi, lo);
}

__device__ __forceinline__ double rcp_newton(double p) {
    double y = __builtin_amdgcn_rcp(p);
    double e = fma(-p, y, 1.0);
    y = fma(y, e, y);
    e = fma(-p, y, 1.0);
    return fma(y, e, y);
}

// address (in d2 units) of slot s of row i' inside a tile
__device__ __forceinline__ int tslot(int row, int s) { return row * 8 + (s ^ ((row >> 1) & 7)); }

// A-operand fragments of a tile for this lane: a[v] = M[rho(lane & 15)][kap(lane >> 4, v)]
__device__ __forceinline__ void load_afrag(const d2* tile, int lane, double (&a)[4]) {
    const int i = lane & 15, kk = lane >> 4;
    const d2 lo = tile[tslot(i, kk)], hi = tile[tslot(i, 4 + kk)];
    a[0] = lo.x; a[1] = lo.y; a[2] = hi.x; a[3] = hi.y;
}

// publish a tile held in X layout (element (fr, kap(fg, v))) as an A operand, scaled by sgn
__device__ __forceinline__ void publish_tile(d2* tile, int lane, const double (&x)[4], double sgn) {
    const int row = rho_inv(lane & 15), fg = lane >> 4;
    tile[tslot(row, fg)] = d2{sgn * x[0], sgn * x[1]};
    tile[tslot(row, 4 + fg)] = d2{sgn * x[2], sgn * x[3]};
}

// y = M * X^T in X layout (see the file header); acc_in is the C operand
__device__ __forceinline__ void tile_mma(const double (&a)[4], const double (&b)[4], double (&c)[4]) {
    d4 acc = d4{c[0], c[1], c[2], c[3]};
#pragma unroll
    for (int v = 0; v < 4; ++v) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a[v], b[v], acc, 0, 0, 0);
    c[0] = acc[0]; c[1] = acc[1]; c[2] = acc[2]; c[3] = acc[3];
}

__device__ __forceinline__ void load_xtile(const double* p, int64_t ld, int lane, double (&x)[4]) {
    const double* r = p + (int64_t)(lane & 15) * ld + 2 * (lane >> 4);
    const d2 lo = *reinterpret_cast<const d2*>(r), hi = *reinterpret_cast<const d2*>(r + 8);
    x[0] = lo.x; x[1] = lo.y; x[2] = hi.x; x[3] = hi.y;
}
__device__ __forceinline__ void store_xtile(double* p, int64_t ld, int lane, const double (&x)[4]) {
    double* r = p + (int64_t)(lane & 15) * ld + 2 * (lane >> 4);
    *reinterpret_cast<d2*>(r) = d2{x[0], x[1]};
    *reinterpret_cast<d2*>(r + 8) = d2{x[2], x[3]};
}

// ---------------------------------------------------------------------------
// Cholesky AND inverse of a 16 x 16 block by one wave, rows in registers, no LDS, no barrier.
// Lanes 0..15: lane r holds row r of the symmetric positive definite block in v[0..15] (entries k <= r are
// read).  Lanes 16..31: lane 16 + c holds column c of the identity.  Right-looking sweep, column c:
//     p = v[c] of lane c;  v[c] *= 1/sqrt(p);  v[k] -= L[k][c] * v[c]  for k > c,  L[k][c] = v[c] of lane k.
// The same instructions run the forward substitution W = L^-1 on lanes 16..31 (both need the scalar L[k][c],
// one v_readlane pair per (c, k)), so the inverse costs no instruction of its own.  The chain pivot ->
// 1/sqrt -> scale -> next pivot is software-pipelined by hand: column c + 1's reciprocal square root runs
// stage by stage between the updates of column c (the order below is pinned with scheduling barriers; left to
// itself the compiler splits the loop in two and spills the scalars).
// Afterwards: lane r holds L[r][k] in v[k] (k <= r); lane 16 + c holds W[r][c] in v[r] (zero for r < c).
// Returns the first non-positive pivot (16 if none).
// ---------------------------------------------------------------------------
struct RsqChain {      // 1/sqrt(p) and sqrt(p) in stages (v_rsq_f64 + two coupled Newton steps + residual fix)
    double p, y, g, h, r;
    __device__ __forceinline__ void stage(int i) {
        switch (i) {
            case 0: y = __builtin_amdgcn_rsq(p); break;
            case 1: g = p * y; h = 0.5 * y; break;
            case 2: r = fma(-h, g, 0.5); break;
            case 3: g = fma(g, r, g); h = fma(h, r, h); break;
            case 4: r = fma(-h, g, 0.5); break;
            case 5: g = fma(g, r, g); h = fma(h, r, h); break;
            case 6: r = fma(-g, g, p); break;
            case 7: g = fma(r, h, g); h = h + h; break;        // g = sqrt(p), h = 1/sqrt(p)
            default: break;
        }
    }
};
constexpr int RSQ_STAGES = 8;
#define PANEL_FENCE() __builtin_amdgcn_sched_barrier(0)

__device__ __forceinline__ int factor16_packed(double (&v)[16], int lane) {
    int bad = 16;
    RsqChain ch;
    ch.p = readlane_d(v[0], 0);
#pragma unroll
    for (int i = 0; i < RSQ_STAGES; ++i) ch.stage(i);
#pragma unroll
    for (int c = 0; c < 16; ++c) {
        // a pivot that is not positive needs no special case beyond its index: 1/sqrt of a negative number is NaN, of a
        // zero +-inf and p * inf = NaN, so g and h are NaN and everything after this column is
        if (!(ch.p > 0.0) && bad == 16) bad = c;
        v[c] = (lane == c) ? ch.g : v[c] * ch.h;
        PANEL_FENCE();
        int st = 0;
        if (c < 15) {
            const double l1 = readlane_d(v[c], c + 1);
            v[c + 1] = fma(-l1, v[c], v[c + 1]);
            PANEL_FENCE();
            ch.p = readlane_d(v[c + 1], c + 1);          // next pivot: its chain overlaps the updates below
            ch.stage(st++);
            PANEL_FENCE();
        }
        // the scalars of the whole column first (a v_readlane result is not usable for ~10 cycles), then the
        // updates, one chain stage of the next pivot between every two of them
        double lk[16];
#pragma unroll
        for (int k = c + 2; k < 16; ++k) lk[k] = readlane_d(v[c], k);
        PANEL_FENCE();
#pragma unroll
        for (int k = c + 2; k < 16; ++k) {
            v[k] = fma(-lk[k], v[c], v[k]);
            if (st < RSQ_STAGES && ((k - c) & 1)) { PANEL_FENCE(); ch.stage(st++); PANEL_FENCE(); }
        }
        PANEL_FENCE();
        if (c < 15) {
#pragma unroll
            for (int i = 0; i < RSQ_STAGES; ++i)
                if (i >= st) ch.stage(i);
            PANEL_FENCE();
        }
    }
    return bad;
}

// lane 16 + c holds column c of W (v[r] = W[r][c]): write it as an A-operand tile
__device__ __forceinline__ void publish_w(double* tile, int lane, const double (&v)[16]) {
    if (lane >= 16 && lane < 32) {
        const int c = lane - 16;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = rho_inv(r);
            tile[2 * tslot(row, c >> 1) + (c & 1)] = v[r];
        }
    }
}

// Phase 1 of a potrf128 step, one copy of the code for all eight steps (everything it touches is behind
// pointers): the diagonal tile sits row-major in `scratch`; factor + invert; to memory: L_jj on and below the
// diagonal of the tile and W_jj TRANSPOSED strictly above it (tile[c][r] = W[r][c], r > c; the diagonal of W is
// 1 / diag(L)) -- the strict upper triangles of the diagonal 16 x 16 tiles are storage nothing else reads, and
// every later triangular solve with this block (trsm128, the backward solve) finds its inverse there instead
// of recomputing it; W_jj as an A operand to its LDS tile; a non-positive pivot to *info.
__device__ __noinline__ void factor_diag_tile(const double* scratch, double* Ajj, int64_t ld, double* wtile,
                                              int64_t col0, int64_t* info) {
    const int lane = threadIdx.x & 63;
    double v[16];
#pragma unroll
    for (int k = 0; k < 16; k += 2) {
        const d2 t = *reinterpret_cast<const d2*>(scratch + (lane & 15) * SCR_LD + k);
        v[k] = (lane < 16) ? t.x : ((lane - 16 == k) ? 1.0 : 0.0);
        v[k + 1] = (lane < 16) ? t.y : ((lane - 16 == k + 1) ? 1.0 : 0.0);
    }
    const int bad = factor16_packed(v, lane);
    if (bad < 16 && lane == 0) atomicMin((unsigned long long*)info, (unsigned long long)(col0 + bad));
    if (lane < 32) {
        double* row = Ajj + (int64_t)(lane & 15) * ld;
#pragma unroll
        for (int k = 0; k < 16; ++k) {
            const bool mine = (lane < 16) ? (k <= lane) : (k > lane - 16);
            if (mine) row[k] = v[k];
        }
    }
    publish_w(wtile, lane, v);
}

// Staging of a diagonal tile in trsm128: W_jj as an A operand from the factored tile G at Ljj -- W[n][k] =
// G[k][n] for k < n (stored transposed above the diagonal by factor_diag_tile), 1 / G[n][n] on the diagonal.
__device__ __forceinline__ void stage_w_tile(const double* Ljj, int64_t ldl, d2* wtile, int lane) {
    const int n = lane & 15, fg = lane >> 4;
    double x[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int k = kap(fg, q);
        double val = 0.0;
        if (k < n) val = Ljj[(int64_t)k * ldl + n];
        else if (k == n) val = rcp_newton(Ljj[(int64_t)n * ldl + n]);
        x[q] = val;
    }
    publish_tile(wtile, lane, x, 1.0);
}

}  // namespace

// ---------------------------------------------------------------------------
// potrf128: A (128 x 128, lower, leading dimension ld) <- its Cholesky factor, one workgroup of 8 waves.
// Wave i owns block row i (tiles (i, 0..i)) in registers for the whole kernel.  Step j:
//   1  wave j       factors + inverts its diagonal tile, writes L_jj to memory and W_jj to LDS
//   2  waves i > j  L_ij = S_ij W_jj^T (4 MFMAs), to memory and, negated, to LDS
//   3  waves i > j  S_ik -= L_ij L_kj^T for k = j+1..i (4 MFMAs per tile), the diagonal chain first
// Two barriers per step; a wave's tiles never leave its registers.  LDS holds only the operands of the current
// step (8 tile slots + W_jj + the scratch tile, 20 KiB).  Beside a trailing update (lookahead) the workgroup still waits
// for an empty CU -- 123 registers at two waves per SIMD do not fit next to an update workgroup's 2 x 144 -- which was
// measured to be the better deal: a 94-register variant that does fit and starts at once runs its serial 16 x 16
// factors so much slower on SIMDs shared with the update's MFMA stream that the panel stream gets longer (DESIGN.md
// section 7 item 2).
// ---------------------------------------------------------------------------
__device__ __forceinline__ void potrf128_body(double* A, int64_t ld, int64_t col_offset, int64_t* info,
                                              unsigned long long* stamps, char* smem) {
    d2* tiles = reinterpret_cast<d2*>(smem);          // slot i: -L_ij of the current step j (block row i), slot 8: W_jj
    double* scratch = reinterpret_cast<double*>(smem + POTRF_TILES * TILE_BYTES);    // 16 x SCR_LD doubles
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int fr = lane & 15, fg = lane >> 4;

    double xt[NT][4];
#pragma unroll
    for (int k = 0; k < NT; ++k) {
        if (k <= wave) load_xtile(A + (int64_t)(16 * wave) * ld + 16 * k, ld, lane, xt[k]);
        else { xt[k][0] = xt[k][1] = xt[k][2] = xt[k][3] = 0.0; }
    }

    // diagnostic (gpmi_probe_panel): stamps[6 j + q] = clock at the q-th boundary of step j, as seen by wave j
    // (phase 1 begin / end) and wave min(j + 1, 7) (after barrier, after phase 2, after barrier, after phase 3)
#define PANEL_STAMP(cond, idx) if (stamps && (cond) && lane == 0) stamps[idx] = __builtin_amdgcn_s_memtime()
    PANEL_STAMP(wave == 0, 48);
#pragma unroll
    for (int j = 0; j < NT; ++j) {
        PANEL_STAMP(wave == j, 6 * j);
        if (wave == j) {
            // X layout -> one row per lane through the scratch tile
#pragma unroll
            for (int v = 0; v < 4; ++v) scratch[fr * SCR_LD + kap(fg, v)] = xt[j][v];
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            factor_diag_tile(scratch, A + (int64_t)(16 * j) * ld + 16 * j, ld,
                             reinterpret_cast<double*>(tiles + NT * 128), col_offset + 16 * j, info);
        }
        PANEL_STAMP(wave == j, 6 * j + 1);
        __syncthreads();
        PANEL_STAMP(wave == (j < 7 ? j + 1 : 7), 6 * j + 2);
        if (wave > j) {
            double a[4], z[4] = {0., 0., 0., 0.};
            load_afrag(tiles + NT * 128, lane, a);
            tile_mma(a, xt[j], z);
#pragma unroll
            for (int v = 0; v < 4; ++v) xt[j][v] = z[v];
            publish_tile(tiles + wave * 128, lane, xt[j], -1.0);
            store_xtile(A + (int64_t)(16 * wave) * ld + 16 * j, ld, lane, xt[j]);
        }
        PANEL_STAMP(wave == (j < 7 ? j + 1 : 7), 6 * j + 3);
        __syncthreads();
        PANEL_STAMP(wave == (j < 7 ? j + 1 : 7), 6 * j + 4);
        if (wave > j) {
#pragma unroll
            for (int k = j + 1; k < NT; ++k) {
                if (k <= wave) {
                    double a[4];
                    load_afrag(tiles + k * 128, lane, a);
                    tile_mma(a, xt[j], xt[k]);
                }
            }
        }
        PANEL_STAMP(wave == (j < 7 ? j + 1 : 7), 6 * j + 5);
    }
    PANEL_STAMP(wave == 7, 49);
#undef PANEL_STAMP
}

__global__ __launch_bounds__(512) void potrf128_kernel(double* A, int64_t ld, int64_t col_offset, int64_t* info,
                                                        unsigned long long* stamps) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    potrf128_body(A, ld, col_offset, info, stamps, smem);
}

// ---------------------------------------------------------------------------
// potrf128 as a SERVER (experiment, option potrf_server): one workgroup stays on a CU for a whole factorisation and factors
// the 128 x 128 diagonal blocks it is handed through a mailbox, so that no potrf128 launch has to wait for the trailing
// update's next round boundary to find an empty CU (a kernel of 8 waves x 123 registers does not fit beside an update
// workgroup).  In the panel stream the launch of potrf128 becomes the launch of a one-wave "post" kernel that writes the
// job (agent-scope atomic stores), bumps the sequence number and WAITS for the server's answer -- the stream's order does
// the rest: the kernels behind it start when the block is factored.  Visibility: the block was written by kernels that
// finished before the post started (their end-of-kernel release); the server takes an agent-scope acquire fence before it
// reads (it has no kernel boundary of its own) and a release fence before it answers.  Every wait is bounded by wall time
// and a give-up is sticky (PotrfMail::err), so a server that never started costs ONE bound, not one per leaf.
// ---------------------------------------------------------------------------
__device__ __forceinline__ unsigned long long mail_load(const unsigned long long* p) {
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void mail_store(unsigned long long* p, unsigned long long v) {
    __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// `mode` (ablations of the experiment; results are only right for mode 1): 2 no acquire / release fences in the server,
// 4 the post does not wait for the answer, 8 the server is resident but never used (plain potrf128 launches),
// 16 the server polls once per ~0.2 ms, 32 the server answers without doing the work
__global__ __launch_bounds__(512) void potrf128_server_kernel(PotrfMail* m, unsigned long long first_seq,
                                                               unsigned long long idle_ticks, int mode) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    volatile unsigned long long* job = reinterpret_cast<volatile unsigned long long*>(smem + POTRF_TILES * TILE_BYTES + 16 * SCR_LD * 8);
    unsigned long long seen = first_seq - 1;
    // ablations 256 / 512 (with 8, the server is never used): waves 1-7 leave at once / wait for wave 0 by polling an LDS
    // word with s_sleep between reads instead of sitting at the workgroup barrier
    if ((mode & 256) && threadIdx.x >= 64) return;
    volatile unsigned long long* go = job + 6;
    if (threadIdx.x == 0) *go = 0;
    __syncthreads();
    unsigned long long round = 0;
    for (;;) {
        ++round;
        if ((mode & 512) && threadIdx.x >= 64) {
            while (*go < round) __builtin_amdgcn_s_sleep(64);
        }
        if (threadIdx.x == 0) {
            const unsigned long long t0 = wall_clock64();
            unsigned long long sq = mail_load(&m->seq_post);
            int polls = 0;
            while (sq <= seen || sq < first_seq) {
                if (mode & 16) { for (int q = 0; q < 64; ++q) __builtin_amdgcn_s_sleep(127); }      // ~one poll per 0.2 ms
                else __builtin_amdgcn_s_sleep(2);
                sq = mail_load(&m->seq_post);
                if ((++polls & 255) == 0 && wall_clock64() - t0 > idle_ticks) { sq = ~0ull; break; }
            }
            job[0] = sq;
            if (sq != ~0ull) {
                job[1] = mail_load(&m->A);
                job[2] = mail_load(&m->ld);
                job[3] = mail_load(&m->col_offset);
                job[4] = mail_load(&m->info);
                job[5] = mail_load(&m->quit);
            }
            *go = round;
        }
        __syncthreads();
        const unsigned long long sq = job[0];
        if (sq == ~0ull) {                                   // nothing came for idle_ticks: leave (a post that comes later gives up)
            if (threadIdx.x == 0) { int* e = &m->err; __hip_atomic_store(e, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
            return;
        }
        double* A = reinterpret_cast<double*>(job[1]);
        const int64_t ld = (int64_t)job[2], col = (int64_t)job[3];
        int64_t* info = reinterpret_cast<int64_t*>(job[4]);
        const bool quit = job[5] != 0;
        __syncthreads();                                      // job[] is rewritten by thread 0 in the next round
        if (!quit && !(mode & 32)) {
            if (!(mode & 2)) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
            potrf128_body(A, ld, col, info, nullptr, smem);
            if (!(mode & 2)) __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        }
        __syncthreads();
        if (threadIdx.x == 0) mail_store(&m->seq_done, sq);
        if (quit) return;
        seen = sq;
    }
}

// ablation 64: in place of the server, ONE wave that only waits for the quit message (no registers to speak of, no LDS)
__global__ __launch_bounds__(64) void potrf128_idle_kernel(PotrfMail* m, unsigned long long first_seq, unsigned long long idle_ticks) {
    if (threadIdx.x) return;
    const unsigned long long t0 = wall_clock64();
    for (;;) {
        const unsigned long long sq = mail_load(&m->seq_post);
        if (sq >= first_seq && mail_load(&m->quit)) { mail_store(&m->seq_done, sq); return; }
        for (int q = 0; q < 16; ++q) __builtin_amdgcn_s_sleep(127);
        if (wall_clock64() - t0 > 10 * idle_ticks) return;
    }
}

__global__ __launch_bounds__(64) void potrf128_post_kernel(PotrfMail* m, unsigned long long seq, double* A, int64_t ld,
                                                            int64_t col_offset, int64_t* info, int quit,
                                                            unsigned long long wait_ticks, int mode) {
    if (threadIdx.x) return;
    if (__hip_atomic_load(&m->err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) return;       // a wait has given up before: do not wait again
    mail_store(&m->A, (unsigned long long)reinterpret_cast<uintptr_t>(A));
    mail_store(&m->ld, (unsigned long long)ld);
    mail_store(&m->col_offset, (unsigned long long)col_offset);
    mail_store(&m->info, (unsigned long long)reinterpret_cast<uintptr_t>(info));
    mail_store(&m->quit, (unsigned long long)quit);
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");       // the job's fields before its number
    mail_store(&m->seq_post, seq);
    if ((mode & 4) && !quit) return;
    const unsigned long long t0 = wall_clock64();
    int polls = 0;
    while (mail_load(&m->seq_done) < seq) {
        __builtin_amdgcn_s_sleep(2);
        if ((++polls & 255) == 0 && wall_clock64() - t0 > wait_ticks) {
            __hip_atomic_store(&m->err, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            return;
        }
    }
}

// ---------------------------------------------------------------------------
// trsm128: X (m x 128) <- X * L^-T, L 128 x 128 lower (its upper triangle is not read).
// Workgroup = 4 waves, two workgroups per CU; staging: the 28 off-diagonal tiles of L go to LDS negated, the 8
// diagonal tiles contribute their stored inverses -- every workgroup does this for itself (72 KiB out of L2,
// behind the load of its first rows), then walks slabs of 64 rows, 16 rows per wave, all 128 columns of a row
// in 64 registers:
//   for j = 0..7:  X_j <- X_j W_jj^T;   X_k -= X_j L_kj^T  for k > j         (144 MFMAs per 16 rows)
// ---------------------------------------------------------------------------
// PHASE -1: the whole solve in one launch (36 operand tiles, 72 KiB of LDS).
// PHASE 0 / 1: the same solve as two launches -- columns 0..63 plus their updates of columns 64..127 (26 tiles,
// 52 KiB), then columns 64..127 (10 tiles, 20 KiB) -- for launches that run beside a trailing update
// (lookahead): its workgroups hold 96 KiB of every CU's 160 KiB, so only kernels with <= 64 KiB start at once.
template <int PHASE>
__device__ __forceinline__ constexpr int trsm_slot(int j, int k) {      // k <= j
    if (PHASE == 0) return j < 4 ? j * (j + 1) / 2 + k : 10 + (j - 4) * 4 + k;
    if (PHASE == 1) return (j - 4) * (j - 3) / 2 + (k - 4);
    return j * (j + 1) / 2 + k;
}
template <int PHASE> constexpr int trsm_tiles() { return PHASE == 0 ? 26 : PHASE == 1 ? 10 : NTILES; }

// Staging of a 128 x 128 factored diagonal block as A operands (4 waves): the off-diagonal tiles of this phase,
// negated, round-robin over the waves, and the stored inverses of the diagonal tiles.  A wave first REQUESTS all of its
// tiles (up to seven off-diagonal ones and two diagonal ones, 36 registers) and only then publishes them: one memory
// round trip instead of one per tile (the first form, a load-and-publish per tile under a per-wave test, took 12.4k
// cycles of a 12 us launch).  The tile list of a phase is a compile-time table, (j << 4) | k in launch order.
template <int PHASE> struct StageList {
    static constexpr int J0 = PHASE == 1 ? 4 : 0, J1 = PHASE == 0 ? 4 : NT;
    static constexpr int count() {
        int n = 0;
        for (int j = 1; j < NT; ++j)
            for (int k = 0; k < j; ++k)
                if (k >= J0 && k < J1) ++n;
        return n;
    }
    static constexpr int entry(int t) {
        int n = 0;
        for (int j = 1; j < NT; ++j)
            for (int k = 0; k < j; ++k)
                if (k >= J0 && k < J1) {
                    if (n == t) return (j << 4) | k;
                    ++n;
                }
        return 0;
    }
};
template <int PHASE, int... T>
__device__ __forceinline__ int stage_entry(int t, std::integer_sequence<int, T...>) {
    // a switch the compiler turns into a scalar table lookup: t is wave-uniform
    constexpr int tab[] = {StageList<PHASE>::entry(T)...};
    return tab[t];
}

template <int PHASE>
__device__ __forceinline__ void stage_l_tiles_batched(const double* L, int64_t ldl, d2* tiles, int lane, int wave,
                                                      unsigned long long* stamps) {
    constexpr int J0 = StageList<PHASE>::J0, J1 = StageList<PHASE>::J1;
    constexpr int NOFF = StageList<PHASE>::count();
    constexpr int PER = (NOFF + 3) / 4;                 // off-diagonal tiles per wave
    constexpr int NW = (J1 - J0 + 3) / 4;               // diagonal tiles per wave
    const int n = lane & 15, fg = lane >> 4;
    double x[PER][4], w[NW][4];
    int slot[PER];
    // ---- requests
#pragma unroll
    for (int i = 0; i < PER; ++i) {
        const int t = 4 * i + wave;
        const int e = stage_entry<PHASE>(t < NOFF ? t : NOFF - 1, std::make_integer_sequence<int, NOFF>{});
        const int jj = e >> 4, kk = e & 15;
        slot[i] = t < NOFF ? (PHASE == 0 ? (jj < 4 ? jj * (jj + 1) / 2 + kk : 10 + (jj - 4) * 4 + kk)
                                         : PHASE == 1 ? (jj - 4) * (jj - 3) / 2 + (kk - 4) : jj * (jj + 1) / 2 + kk)
                           : -1;
        load_xtile(L + (int64_t)(16 * jj) * ldl + 16 * kk, ldl, lane, x[i]);
    }
#pragma unroll
    for (int r = 0; r < NW; ++r) {
        const int j = J0 + wave + 4 * r;
        const double* Ljj = L + (int64_t)(16 * (j < J1 ? j : J0)) * (ldl + 1);
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int k = kap(fg, q);
            w[r][q] = (k <= n) ? Ljj[(int64_t)k * ldl + n] : 0.0;     // W^T above the diagonal, L's diagonal on it
        }
    }
    // ---- publication
#pragma unroll
    for (int i = 0; i < PER; ++i)
        if (slot[i] >= 0) publish_tile(tiles + slot[i] * 128, lane, x[i], -1.0);
    if (stamps && wave == 0 && lane == 0) stamps[57] = __builtin_amdgcn_s_memtime();
#pragma unroll
    for (int r = 0; r < NW; ++r) {
        const int j = J0 + wave + 4 * r;
        if (j < J1) {
#pragma unroll
            for (int q = 0; q < 4; ++q)
                if (kap(fg, q) == n) w[r][q] = rcp_newton(w[r][q]);        // diag(W) = 1 / diag(L)
            publish_tile(tiles + trsm_slot<PHASE>(j, j) * 128, lane, w[r], 1.0);
        }
    }
}

// The two-launch forms (PHASE 0 / 1) run beside a trailing update, where what counts is that two workgroups fit on a CU
// next to an update workgroup (<= 112 registers): they keep the one-tile-at-a-time staging (108 / 68 registers against
// 138 / 74 batched); the one-launch form, which has the CU to itself, stages batched (154 registers).
template <int PHASE>
__device__ __forceinline__ void stage_l_tiles(const double* L, int64_t ldl, d2* tiles, int lane, int wave,
                                              unsigned long long* stamps) {
    if constexpr (PHASE == -1) {
        stage_l_tiles_batched<PHASE>(L, ldl, tiles, lane, wave, stamps);
    } else {
        constexpr int J0 = PHASE == 1 ? 4 : 0, J1 = PHASE == 0 ? 4 : NT;
        int t = 0;
#pragma unroll
        for (int j = 1; j < NT; ++j)
#pragma unroll
            for (int k = 0; k < j; ++k) {
                if (k >= J0 && k < J1) {
                    if ((t & 3) == wave) {
                        double x[4];
                        load_xtile(L + (int64_t)(16 * j) * ldl + 16 * k, ldl, lane, x);
                        publish_tile(tiles + trsm_slot<PHASE>(j, k) * 128, lane, x, -1.0);
                    }
                    ++t;
                }
            }
        if (stamps && wave == 0 && lane == 0) stamps[57] = __builtin_amdgcn_s_memtime();
#pragma unroll
        for (int j = J0; j < J1; ++j)
            if (((j - J0) & 3) == wave)
                stage_w_tile(L + (int64_t)(16 * j) * ldl + 16 * j, ldl, tiles + trsm_slot<PHASE>(j, j) * 128, lane);
    }
}

template <int PHASE>
__global__ __launch_bounds__(256, 2) void trsm128_kernel(const double* L, int64_t ldl, double* X, int64_t ldx,
                                                          int64_t nslabs, unsigned long long* stamps) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    d2* tiles = reinterpret_cast<d2*>(smem);
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    constexpr int J0 = PHASE == 1 ? 4 : 0;             // first / one past last column tile this launch solves
    constexpr int J1 = PHASE == 0 ? 4 : NT;
    constexpr int T0 = PHASE == 1 ? 4 : 0;             // first column tile this launch touches
#define PANEL_STAMP(cond, idx) if (stamps && (cond) && lane == 0) stamps[idx] = __builtin_amdgcn_s_memtime()
    PANEL_STAMP(blockIdx.x == 0 && wave == 0, 56);

    // the first slab's rows are requested before L is staged: their latency hides behind the staging
    double xt[NT][4];
    int64_t slab = blockIdx.x;
    if (slab < nslabs) {
        const double* Xr = X + (slab * TRSM_ROWS + 16 * wave) * ldx;
#pragma unroll
        for (int k = T0; k < NT; ++k) load_xtile(Xr + 16 * k, ldx, lane, xt[k]);
    }
    stage_l_tiles<PHASE>(L, ldl, tiles, lane, wave, blockIdx.x == 0 ? stamps : nullptr);
    PANEL_STAMP(blockIdx.x == 0 && wave == 0, 58);
    __syncthreads();
    PANEL_STAMP(blockIdx.x == 0 && wave == 0, 59);

    for (; slab < nslabs; slab += gridDim.x) {
        double* Xr = X + (slab * TRSM_ROWS + 16 * wave) * ldx;
        if (slab != (int64_t)blockIdx.x) {
#pragma unroll
            for (int k = T0; k < NT; ++k) load_xtile(Xr + 16 * k, ldx, lane, xt[k]);
        }
        if (stamps) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        PANEL_STAMP(blockIdx.x == 0 && wave == 0 && slab == blockIdx.x, 60);
#pragma unroll
        for (int j = J0; j < J1; ++j) {
            double a[4], z[4] = {0., 0., 0., 0.};
            load_afrag(tiles + trsm_slot<PHASE>(j, j) * 128, lane, a);
            tile_mma(a, xt[j], z);
#pragma unroll
            for (int v = 0; v < 4; ++v) xt[j][v] = z[v];
#pragma unroll
            for (int k = j + 1; k < NT; ++k) {
                double b[4];
                load_afrag(tiles + trsm_slot<PHASE>(k, j) * 128, lane, b);
                tile_mma(b, xt[j], xt[k]);
            }
            __builtin_amdgcn_sched_barrier(0);      // one step's operand fragments at a time (register pressure)
        }
        PANEL_STAMP(blockIdx.x == 0 && wave == 0 && slab == blockIdx.x, 61);
#pragma unroll
        for (int k = T0; k < NT; ++k) store_xtile(Xr + 16 * k, ldx, lane, xt[k]);
    }
    PANEL_STAMP(blockIdx.x == 0 && wave == 0, 62);
#undef PANEL_STAMP
}

// ---------------------------------------------------------------------------
// vinv128: for every 128 x 128 diagonal block of a fused factor, V = L_kk^-1, stored TRANSPOSED in the block's
// upper triangle (G[c][r] = V[r][c], r > c) -- the convention of the 16 x 16 diagonal tiles (W_jj^T strictly above
// the diagonal, diag(V) = 1 / diag(L) implied) carried to the whole block: the strict block-upper tiles are storage
// nothing else reads.  One workgroup per block; wave w takes tile rows w and 4 + w of the identity through the
// trsm128 recurrence (I L^-T = V^T), skipping the column tiles left of its own (they stay zero).
// The backward solve (solve.hip) then gets x_k = V^T r_k as a matrix-vector product instead of 8 dependent
// 16 x 16 solve-and-update rounds.  Only ever used for vectors (alpha); the factorisation itself and the
// predictive solves keep to 16 x 16 inverses.
// ---------------------------------------------------------------------------
// vside (may be null): additionally V itself, row-major 128 x 128 per block with zeros above the diagonal and
// diag(V) = 1 / diag(L) in place, block b at vside + b * 128 * 128 -- the single-launch backward solve (solve.hip)
// streams it like one more block of L.  The caller zero-fills vside first (column tiles left of a wave's own are
// never written).
__global__ __launch_bounds__(256, 2) void vinv128_kernel(double* A, int64_t ld, double* vside) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    d2* tiles = reinterpret_cast<d2*>(smem);
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    double* Lb = A + (int64_t)blockIdx.x * PB * (ld + 1);
    stage_l_tiles<-1>(Lb, ld, tiles, lane, wave, nullptr);
    __syncthreads();
    for (int p = 0; p < 2; ++p) {
        const int i = 4 * p + wave;
        double xt[NT][4];
#pragma unroll
        for (int k = 0; k < NT; ++k)
#pragma unroll
            for (int v = 0; v < 4; ++v) xt[k][v] = (k == i && (lane & 15) == kap(lane >> 4, v)) ? 1.0 : 0.0;
#pragma unroll
        for (int j = 0; j < NT; ++j) {
            if (j >= i) {
                double a[4], z[4] = {0., 0., 0., 0.};
                load_afrag(tiles + trsm_slot<-1>(j, j) * 128, lane, a);
                tile_mma(a, xt[j], z);
#pragma unroll
                for (int v = 0; v < 4; ++v) xt[j][v] = z[v];
#pragma unroll
                for (int k = j + 1; k < NT; ++k) {
                    double b[4];
                    load_afrag(tiles + trsm_slot<-1>(k, j) * 128, lane, b);
                    tile_mma(b, xt[j], xt[k]);
                }
            }
            __builtin_amdgcn_sched_barrier(0);
        }
#pragma unroll
        for (int k = 1; k < NT; ++k)
            if (k > i) store_xtile(Lb + (int64_t)(16 * i) * ld + 16 * k, ld, lane, xt[k]);
        if (vside) {
            // xt[k][v] = V^T[16 i + fr][16 k + kap(fg, v)] = V[16 k + kap(fg, v)][16 i + fr]: 16 lanes (fr) write 128 contiguous bytes
            double* Vb = vside + (int64_t)blockIdx.x * PB * PB + 16 * i + (lane & 15);
#pragma unroll
            for (int k = 0; k < NT; ++k)
                if (k >= i) {
#pragma unroll
                    for (int v = 0; v < 4; ++v) Vb[(16 * k + kap(lane >> 4, v)) * PB] = xt[k][v];
                }
        }
    }
}

static hipError_t panel_mfma_attrs() {
    static PerDeviceOnce once;
    return once.run([]() -> hipError_t {
        hipError_t e = hipFuncSetAttribute((const void*)trsm128_kernel<-1>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                           NTILES * TILE_BYTES);
        if (e != hipSuccess) return e;
        return hipFuncSetAttribute((const void*)vinv128_kernel, hipFuncAttributeMaxDynamicSharedMemorySize,
                                   NTILES * TILE_BYTES);
    });
}

// the server a factorisation on this thread has started (PotrfServerScope), or null
static thread_local PotrfServerState* t_potrf_server = nullptr;

hipError_t potrf_server_start(PotrfServerState* st, hipStream_t server_stream) {
    if (!st || !st->mail) return hipErrorInvalidValue;
    hipError_t e = panel_mfma_attrs();
    if (e != hipSuccess) return e;
    st->first_seq = ++st->seq;                      // the first job of this server's life
    --st->seq;
    if (st->mode & 64)
        hipLaunchKernelGGL(potrf128_idle_kernel, dim3(1), dim3(64), 0, server_stream, st->mail, st->seq + 1,
                           (unsigned long long)(st->idle_ms * 1e5));
    else
        hipLaunchKernelGGL(potrf128_server_kernel, dim3(1), dim3(512), POTRF_TILES * TILE_BYTES + 16 * SCR_LD * 8 + 64, server_stream,
                           st->mail, st->seq + 1, (unsigned long long)(st->idle_ms * 1e5), st->mode);
    e = hipGetLastError();
    if (e == hipSuccess) t_potrf_server = st;
    return e;
}

hipError_t potrf_server_stop(PotrfServerState* st, hipStream_t s) {
    t_potrf_server = nullptr;
    if (!st || !st->mail) return hipSuccess;
    hipLaunchKernelGGL(potrf128_post_kernel, dim3(1), dim3(64), 0, s, st->mail, ++st->seq, (double*)nullptr, (int64_t)0, (int64_t)0,
                       (int64_t*)nullptr, 1, (unsigned long long)(st->wait_ms * 1e5), st->mode);
    return hipGetLastError();
}

hipError_t launch_potrf128(hipStream_t s, double* A, int64_t ld, int64_t col_offset, int64_t* info_dev) {
    if (ld % 2 || (reinterpret_cast<uintptr_t>(A) & 15)) return hipErrorInvalidValue;
    hipError_t e = panel_mfma_attrs();
    if (e != hipSuccess) return e;
    PotrfServerState* st = t_potrf_server;
    if (st && !(st->mode & 8)) {                     // hand the block to the resident workgroup and wait for it in stream order
        hipLaunchKernelGGL(potrf128_post_kernel, dim3(1), dim3(64), 0, s, st->mail, ++st->seq, A, ld, col_offset, info_dev, 0,
                           (unsigned long long)(st->wait_ms * 1e5), st->mode);
        if (!(st->mode & 4)) return hipGetLastError();
        // mode 4 (the post does not wait): the plain launch below does the work, the server does it too, unordered
    }
    hipLaunchKernelGGL(potrf128_kernel, dim3(1), dim3(512), POTRF_TILES * TILE_BYTES + 16 * SCR_LD * 8, s, A, ld, col_offset,
                       info_dev, tuning().panel_stamps);
    return hipGetLastError();
}

hipError_t launch_trsm128(hipStream_t s, const double* L, int64_t ldl, double* X, int64_t ldx, int64_t m) {
    if (m <= 0) return hipSuccess;
    if (m % TRSM_ROWS || ldl % 2 || ldx % 2 || (reinterpret_cast<uintptr_t>(L) & 15) || (reinterpret_cast<uintptr_t>(X) & 15))
        return hipErrorInvalidValue;
    hipError_t e = panel_mfma_attrs();
    if (e != hipSuccess) return e;
    const int64_t nslabs = m / TRSM_ROWS;
    const unsigned grid = (unsigned)std::min<int64_t>(nslabs, 512);     // two workgroups per CU (72 KiB of LDS each)
    if (gemm_shallow_active()) {            // beside a trailing update: two launches that fit next to its workgroups
        hipLaunchKernelGGL(trsm128_kernel<0>, dim3(grid), dim3(256), trsm_tiles<0>() * TILE_BYTES, s, L, ldl, X, ldx, nslabs,
                           tuning().panel_stamps);
        hipLaunchKernelGGL(trsm128_kernel<1>, dim3(grid), dim3(256), trsm_tiles<1>() * TILE_BYTES, s, L, ldl, X, ldx, nslabs,
                           tuning().panel_stamps);
        return hipGetLastError();
    }
    hipLaunchKernelGGL(trsm128_kernel<-1>, dim3(grid), dim3(256), NTILES * TILE_BYTES, s, L, ldl, X, ldx, nslabs,
                       tuning().panel_stamps);
    return hipGetLastError();
}

hipError_t launch_vinv128(hipStream_t s, double* A, int64_t ld, int64_t n, double* vside) {
    if (n <= 0 || n % PB || ld % 2 || (reinterpret_cast<uintptr_t>(A) & 15) || (reinterpret_cast<uintptr_t>(vside) & 15))
        return hipErrorInvalidValue;
    hipError_t e = panel_mfma_attrs();
    if (e != hipSuccess) return e;
    if (vside) {
        e = hipMemsetAsync(vside, 0, (size_t)n * PB * sizeof(double), s);
        if (e != hipSuccess) return e;
    }
    hipLaunchKernelGGL(vinv128_kernel, dim3((unsigned)(n / PB)), dim3(256), NTILES * TILE_BYTES, s, A, ld, vside);
    return hipGetLastError();
}

}  // namespace gpmi
