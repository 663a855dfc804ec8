// Internal declarations shared by the translation units of libgpmi355x.so.
// gfx950 (MI355X / CDNA4) only: 64-lane wavefronts, v_mfma_f64_16x16x4_f64.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <mutex>

namespace gpmi {

constexpr int TILE = 128;   // row / column padding granule of every matrix
constexpr int IB = 64;      // inner (register-resident) panel width

// Kernel-selection options (gpmi_set_option).  They live in the context; a C-ABI call installs its
// context's set for the calling thread (TuneScope), so two contexts driven from two threads (the lanes
// of gpmi_lml_batch) never see each other's settings.  The context-free gpmi_dev_* primitives run with
// the defaults.
struct Tuning {
    int gemm_use_dma = 1;       // LDS-DMA GEMM for launches with >= 256 tiles
    int gemm_small_tiles = 1;   // 64 x 64 tiles for launches with few tiles
    int gemm_persist = 1;       // resident workgroups that chain the K loops of consecutive tiles (launches with >= 2 rounds of tiles)
    int gemm_ticket = 0;        // ticket form of the per-tile kernel (resident workgroups, tiles drawn from counters, no state across tiles): 1 Cholesky trailing updates under lookahead, 2 every launch of at least one round
    int potrf_server = 0;       // experiment: potrf128 as a resident workgroup fed through a mailbox while a factorisation runs under lookahead
    int gemm_balance = 1;       // per-tile launches: choose the supertile edge of mid-size triangular launches by the deal of blocks to the XCDs (gpmi_plan.h: plan_tri_xcd_efficiency); 0: always the widest
    int gemm_reserve = 0;       // ticket form: CUs per XCD the launch leaves untouched (for the panel kernels of the other stream)
    int gemm_dma_waves = 8;     // 4: one wave per SIMD, 8: two waves per SIMD (32 x 64 per wave)
    int trsm_wave = 1;          // 1: wave-per-row substitution kernel for short panels, 0: lane-per-row always
    int rbf_blocks = 16384;     // persistent blocks of the register-path K build
    int trsv_vinv = 2;          // backward solve: 2 one launch, column blocks chained through the solution vector (inverted 128 x 128 diagonal blocks); 1 one launch per 128 unknowns with the same inverses; 0 the 16 x 16 rounds
    int panel_fused = 1;        // 1: fused multi-column panel kernels, 0: first-generation potf2 + substitution leaves
    int gemm_small_dma = 1;     // 1: deep-prefetch LDS-DMA kernel for launches with few tiles, 0: first-generation 64 x 64 kernel
    int gemm_dbg = 0;           // timing-only ablation bits (gpmi_probe_gemm); results are wrong when non-zero
    unsigned long long* gemm_stamps = nullptr;   // diagnostic stamp buffer (gpmi_probe_gemm variant bit 16)
    unsigned long long* panel_stamps = nullptr;  // diagnostic: s_memtime stamps of the panel kernels (gpmi_probe_panel)
};
const Tuning& tuning();         // options of the C-ABI call running on this thread
struct TuneScope {
    const Tuning* prev;
    explicit TuneScope(const Tuning* t);
    ~TuneScope();
};

// One-time, per-device opt-in (hipFuncSetAttribute for > 64 KiB of dynamic LDS): thread-safe, keyed by
// the current device, and the error is returned instead of dropped.
struct PerDeviceOnce {
    std::mutex mu;
    uint64_t done = 0;
    template <class F> hipError_t run(F fn) {
        int dev = 0;
        hipError_t e = hipGetDevice(&dev);
        if (e != hipSuccess) return e;
        std::lock_guard<std::mutex> lock(mu);
        if (dev < 64 && ((done >> dev) & 1)) return hipSuccess;
        e = fn();
        if (e == hipSuccess && dev < 64) done |= (uint64_t)1 << dev;
        return e;
    }
};

// ---- gemm_nt.hip ----------------------------------------------------------
// C (M x N) op= A (M x K) * B (N x K)^T, all row-major.  M multiple of 128,
// N multiple of 64, K multiple of 16.  mode 0: C -= A*B^T; mode 1: C = A*B^T.
// lower != 0: skip tiles lying entirely above {col <= row + diag_off}.
struct GemmArgs {
    double* C;
    const double* A;
    const double* B;
    int64_t ldc, lda, ldb;
    int64_t M, N, K;
    int mode;
    int lower;
    int64_t diag_off;
    // optional: row_ncols[ti / row_block_tiles] = number of leading columns of C that
    // the 128-row tile band ti updates (row-block cyclic storage: a rank's stacked
    // row blocks reach different distances to the right); device pointer or null
    const int32_t* row_ncols = nullptr;
    int row_block_tiles = 1;
    // optional host copy of row_ncols (row_bands entries): lets the launcher enumerate only the
    // supertiles that hold live tiles instead of the whole rectangle
    const int32_t* row_ncols_host = nullptr;
    int row_bands = 0;
    int role = 0;   // 1: Cholesky trailing update (launched under its own kernel symbol)
    // optional: B is not one matrix but a sequence of b_block_rows-row blocks (each b_block_rows x K, leading
    // dimension ldb) at B + b_block_off[i] doubles -- the panel column exactly as an all-gather leaves it, one
    // contiguous chunk per rank, read in natural block order through this table (device pointer); LDS-DMA kernel only
    const int64_t* b_block_off = nullptr;
    int64_t b_block_rows = 0;
};
hipError_t launch_gemm_nt(hipStream_t s, const GemmArgs& a);
bool gemm_nt_routes_dma(const GemmArgs& a);
double gemm_nt_algorithmic_flops(const GemmArgs& a, int64_t real_rows);   // 2K per needed element (lower: on/below the diagonal)   // true when launch_gemm_nt hands this launch to the LDS-DMA kernel
// gemm_dma.hip: one-workgroup-per-CU LDS-DMA variant (mode 0, N % 128 == 0)
bool gemm_dma_eligible(const GemmArgs& a);
hipError_t launch_gemm_nt_dma(hipStream_t s, const GemmArgs& a);
// gemm_dma.hip: latency-oriented variant for launches with few tiles (64 x 64 tiles, eight K steps in flight)
bool gemm_small_eligible(const GemmArgs& a);
bool gemm_shallow_active();
bool gemm_two_streams_active();
// while alive on this thread: `on` -- small launches use the shallow ring and trsm128 its two-launch form (they run
// beside a trailing update); `on` or `two_streams` -- the calling driver keeps two streams busy at once, so no launch
// may take the whole chip for its whole length (the persistent GEMM form stays off)
struct GemmShallowScope {
    int prev, prev_two;
    // exact: `on` == false switches the small-LDS forms OFF for the scope (default: an enclosing scope's setting stays)
    explicit GemmShallowScope(bool on, bool two_streams = false, bool exact = false);
    ~GemmShallowScope();
};
hipError_t launch_gemm_nt_small(hipStream_t s, const GemmArgs& a);
// number of tiles the launch actually computes (for flop accounting)
double gemm_nt_flops(const GemmArgs& a);

// ---- panel.hip -------------------------------------------------------------
// Cholesky of one 64x64 diagonal block in place (lower), one wavefront.
hipError_t launch_potf2_64(hipStream_t s, double* A, int64_t ld, int64_t col_offset,
                           int64_t* info_dev);
// X (m x 64) <- X * L^-T, L 64x64 lower; m multiple of 64.
hipError_t launch_trsm_rlt64(hipStream_t s, const double* L, int64_t ldl, double* X, int64_t ldx,
                             int64_t m);

// ---- panel_mfma.hip --------------------------------------------------------
// Cholesky of one 128 x 128 diagonal block in place (lower), one workgroup, MFMA updates.
hipError_t launch_potrf128(hipStream_t s, double* A, int64_t ld, int64_t col_offset, int64_t* info_dev);

// potrf128 as a resident server (experiment; panel_mfma.hip): device mailbox + host-side state of one factorisation
struct PotrfMail {
    unsigned long long seq_post, seq_done;
    unsigned long long A, ld, col_offset, info, quit;
    int err;
    int pad;
};
struct PotrfServerState {
    PotrfMail* mail = nullptr;          // device, zero-initialised once
    unsigned long long seq = 0;         // last sequence number handed out
    unsigned long long first_seq = 0;
    double idle_ms = 3000.0;            // the server leaves when nothing has come for this long
    double wait_ms = 2000.0;            // a post gives up (sticky) after this long
    int mode = 1;                       // the option's value: 1 on; further bits = timing-only ablations (panel_mfma.hip)
};
hipError_t potrf_server_start(PotrfServerState* st, hipStream_t server_stream);
hipError_t potrf_server_stop(PotrfServerState* st, hipStream_t panel_stream);

// X (m x 128) <- X * L^-T, L 128 x 128 lower; m multiple of 128; on the matrix pipe.
hipError_t launch_trsm128(hipStream_t s, const double* L, int64_t ldl, double* X, int64_t ldx, int64_t m);

// ---- rbf.hip ---------------------------------------------------------------
struct RbfArgs {
    const double* A;   // rows of the output: nA x d (row-major, device)
    const double* B;   // cols of the output: nB x d
    int64_t nA, nB, d;
    int64_t row0;      // first output row (index into A)
    int64_t nrows;     // rows to produce (multiple of 128 incl. padding)
    int64_t ncols;     // cols to produce (multiple of 128 incl. padding)
    double coef;       // -.5 * (1 / l^2)
    double sig2;       // sigma^2
    double diag_add;   // + s on global row == col (symmetric build only)
    int symmetric;     // 1: A==B, lower tiles only, identity padding
    double* out;       // nrows x ld, out(0,0) is element (row0, 0)
    int64_t ld;
    // covariance function: 0 squared-exponential (coef, sig2 above);
    // 1 linear  sum_k (a_k - c)(b_k - c), c = kp0            (GP_regression.py:22-33)
    // 2 periodic exp(-2 sin^2(pi |a-b| / p) / l^2), p = kp0, l = kp1, d == 1 (GP_regression.py:36-50)
    // 3 CO2 composite: kernel_1 + kernel_2 + kernel_3 + kernel_4 with theta_1..11 = kpv (CO2_example.py:9-94)
    int kind = 0;
    double kp0 = 0., kp1 = 0.;
    double kpv[11] = {0., 0., 0., 0., 0., 0., 0., 0., 0., 0., 0.};
    int delta_square = 0;     // kind 3: the reference's output is square -> kernel_4 adds theta_11^2 * eye (:58-59)
    int64_t delta_col0 = 0;   // ... at row == col + delta_col0 (B is a window of the column inputs starting there)
    // upper bound of |a_i - b_j|^2 over the whole launch (from the inputs' bounding boxes), or < 0
    // when unknown: lets the squared-exponential build skip its per-wave exp domain test
    double max_sq = -1.0;
};
hipError_t launch_rbf(hipStream_t s, const RbfArgs& a);

// ---- grad.hip --------------------------------------------------------------
// sum_ij (alpha_i alpha_j - K_y^-1_ij) dK_ij/dtheta over a block of rows (tune_hyperparms_regression.py:54-57)
struct GradArgs {
    const double* A;          // row inputs  nA x d (device)
    const double* B;          // col inputs  nB x d
    int64_t nA, nB, d;
    int64_t row0, nrows;      // rows row0 .. row0+nrows of the full matrix in this launch
    const double* alpha_r;    // alpha indexed by global row
    const double* alpha_c;    // alpha indexed by column
    const double* Kinv;       // (row0 + r, c) at Kinv[r*ld + c]; holds kinv_sign * K_y^-1
    int64_t ld;
    double kinv_sign;
    double coef, sig2, two_sigma, inv_l3;
    int tri;                  // 1: symmetric case, lower tiles only (needs row0 == 0, nrows == nB)
    double* partial;          // 2 doubles per block (grad_trace_blocks of them)
};
int64_t grad_trace_blocks(const GradArgs& a);
hipError_t launch_grad_trace(hipStream_t s, const GradArgs& a);
hipError_t launch_set_identity_diag(hipStream_t s, double* V, int64_t ld, int64_t n);

// ---- solve.hip -------------------------------------------------------------
// dot[i] = sum_j V[i][j]*m[j], sq[i] = sum_j V[i][j]^2, j < ncols (fixed order)
// out (n x nf row-major) = tril(L) @ Z (n x nf row-major), L n x n with leading dimension ld: reads j <= i only
hipError_t launch_tri_mul(hipStream_t s, const double* L, int64_t ld, const double* Z, int64_t n, int64_t nf, double* out);
hipError_t launch_row_dots(hipStream_t s, const double* V, int64_t ld, int64_t nrows,
                           int64_t ncols, const double* m, double* dot, double* sq);
// out[0] = sum_{i<n} log(A[i*(ld+1)]), out[1] = sum_{i<n} m[i]^2 (deterministic)
hipError_t launch_lml_reduce(hipStream_t s, const double* A, int64_t ld, const double* m,
                             int64_t n, double* out2);
// out2[0] = sum_{i<n} log(A[i*(ld+1)]) (0 if A null), out2[1] = sum_{i<nx} x[i]^2 (0 if x null)
hipError_t launch_logdiag_sumsq(hipStream_t s, const double* A, int64_t ld, int64_t n, const double* x,
                                int64_t nx, double* out2);
// backward substitution  L^T x = b  (x overwrites b); n multiple of 64
hipError_t launch_trsv_lt(hipStream_t s, const double* L, int64_t ld, double* b, int64_t n);
// the same for a factor left by the fused panel kernels (inverses in the diagonal tiles), 128 unknowns per
// launch; n multiple of 128; b is destroyed, the solution goes to xout (must not alias b)
hipError_t launch_trsv_lt_fused(hipStream_t s, const double* L, int64_t ld, double* b, double* xout, int64_t n);
// full inverses of the 128 x 128 diagonal blocks into their upper triangles, and the backward solve that uses them
// vside (optional, n * 128 doubles): V = L_kk^-1 itself, row-major 128 x 128 per block, for launch_trsv_lt_chain
hipError_t launch_vinv128(hipStream_t s, double* A, int64_t ld, int64_t n, double* vside = nullptr);
hipError_t launch_trsv_lt_vinv(hipStream_t s, const double* L, int64_t ld, double* b, double* xout, int64_t n);
// the same in ONE launch (column blocks chained through the solution vector itself); m is only read; err_dev: one int,
// set if a poll gave up
// skip / max_wait_ms: gpmi_probe_trsv_giveup only (bottom blocks left unsolved, a shorter bound on every wait; 0: 10 s)
hipError_t launch_trsv_lt_chain(hipStream_t s, const double* L, int64_t ld, const double* vside, const double* m,
                                double* xout, int64_t n, int* err_dev, int skip = 0, double max_wait_ms = 0.0);
// y[c] = sum_r A[r][c] * x[r]; scratch: ceil(nrows/64) * ncols doubles
hipError_t launch_gemv_t(hipStream_t s, const double* A, int64_t ld, int64_t nrows, int64_t ncols,
                         const double* x, double* y, double* scratch);
// out2[0..1] += the sums of the even / odd entries of part (n pairs), fixed order
hipError_t launch_sum_pairs(hipStream_t s, const double* part, int64_t n, double* out2);
// out[i] = (base ? base[i] : 0) + scale * sum_{q < count} in[q * stride + i], i < n; the sum runs in index order
hipError_t launch_sum_fixed(hipStream_t s, const double* in, int64_t count, int64_t stride, int64_t n,
                            const double* base, double scale, double* out);
// Y (rows x cols) += a * X
hipError_t launch_axpy2d(hipStream_t s, double* Y, int64_t ldy, const double* X, int64_t ldx, int64_t rows,
                         int64_t cols, double a);
// fill helpers
hipError_t launch_fill_rows(hipStream_t s, double* A, int64_t ld, int64_t nrows, int64_t ncols,
                            double value);
hipError_t launch_set_yrow(hipStream_t s, double* row, const double* y, int64_t N, int64_t ncols);
// C = diag_val*I + Kss - G on the lower tiles (post-covariance assembly)
hipError_t launch_extract(hipStream_t s, const double* A, int64_t ld, int64_t r0, int64_t r1,
                          int64_t c0, int64_t c1, double* out, int lower_only);

// ---- probes ----------------------------------------------------------------
hipError_t launch_probe_mfma(hipStream_t s, double* sink, int iters, int cus, int waves_per_simd, int nacc,
                             unsigned long long* clk);
hipError_t launch_probe_write(hipStream_t s, double* buf, int64_t n_doubles, int mode, int blocks, double* sink);

}  // namespace gpmi
