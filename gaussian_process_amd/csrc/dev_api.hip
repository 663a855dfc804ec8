// Device-pointer block primitives of the C-ABI (gpmi_dev_*): the building blocks the multi-GPU
// driver (gaussian_process_amd/dist.py) chains on its own streams.  No context, no host copies.
#include "gpmi_ctx.h"

using namespace gpmi;

extern "C" {

// ---- device-pointer block primitives (multi-GPU driver) -------------------------
int gpmi_dev_rbf_rows(void* stream, const double* X_dev, int64_t N, int64_t d, int64_t row0,
                      int64_t nrows, int64_t ncols, double sigma, double ell, double noise_var,
                      double* out_dev, int64_t ld) {
    if (!X_dev || !out_dev) return fail_arg("gpmi_dev_rbf_rows: null pointer");
    if (nrows % TILE || ncols % TILE || row0 % TILE || ld < ncols || ld % 2)
        return fail_arg("gpmi_dev_rbf_rows: sizes must be multiples of 128");
    RbfArgs r;
    r.A = r.B = X_dev; r.nA = r.nB = N; r.d = d; r.row0 = row0; r.nrows = nrows; r.ncols = ncols;
    r.coef = -.5 * (1 / (ell * ell)); r.sig2 = sigma * sigma; r.diag_add = noise_var; r.symmetric = 1;
    r.out = out_dev; r.ld = ld;
    HIP_TRY(launch_rbf((hipStream_t)stream, r));
    return GPMI_OK;
}

int gpmi_dev_rbf_cross(void* stream, const double* Xs_dev, int64_t n, const double* X_dev, int64_t N,
                       int64_t d, int64_t row0, int64_t nrows, int64_t ncols, double sigma, double ell,
                       double* out_dev, int64_t ld) {
    if (!Xs_dev || !X_dev || !out_dev) return fail_arg("gpmi_dev_rbf_cross: null pointer");
    if (nrows % TILE || ncols % TILE || ld < ncols || ld % 2)
        return fail_arg("gpmi_dev_rbf_cross: sizes must be multiples of 128");
    RbfArgs r;
    r.A = Xs_dev; r.B = X_dev; r.nA = n; r.nB = N; r.d = d; r.row0 = row0; r.nrows = nrows; r.ncols = ncols;
    r.coef = -.5 * (1 / (ell * ell)); r.sig2 = sigma * sigma; r.diag_add = 0.; r.symmetric = 0;
    r.out = out_dev; r.ld = ld;
    HIP_TRY(launch_rbf((hipStream_t)stream, r));
    return GPMI_OK;
}

// covariance parameters of the gpmi_dev_cov_* entry points, as gpmi_set_kernel / gpmi_set_kernel_params take them
static int dev_cov_args(RbfArgs& r, int kind, const double* params, int nparams, const char* who) {
    static const int want[4] = {2, 1, 2, 11};
    if (kind < 0 || kind > 3) return fail_arg("gpmi_dev_cov: kind must be 0 (rbf), 1 (linear), 2 (periodic) or 3 (CO2 composite)");
    if (!params || nparams != want[kind]) return fail_arg("gpmi_dev_cov: kinds 0 / 1 / 2 / 3 take 2 / 1 / 2 / 11 parameters");
    (void)who;
    r.kind = kind;
    if (kind == 0) {
        if (!(params[1] != 0.0)) return fail_arg("gpmi_dev_cov: ell must be non-zero");
        r.coef = -.5 * (1 / (params[1] * params[1])); r.sig2 = params[0] * params[0];
    } else if (kind == 1) {
        r.kp0 = params[0];
    } else if (kind == 2) {
        if (!(params[0] != 0.0) || !(params[1] != 0.0)) return fail_arg("gpmi_dev_cov: period and lengthscale must be non-zero");
        if (r.d != 1) return fail_arg("gpmi_dev_cov: the periodic kernel is 1-D only (GP_regression.py:48)");
        r.kp0 = params[0]; r.kp1 = params[1];
    } else {
        for (int i = 0; i < 11; ++i) r.kpv[i] = params[i];
    }
    return GPMI_OK;
}

// gpmi_dev_rbf_rows for any of the reference's covariance functions (f4 on the partitioned path): rows row0 .. of
// K(X, X) + noise_var * I, lower tiles, identity padding.  kind / params as gpmi_set_kernel (0: sigma, l; 1: c;
// 2: period, l) and gpmi_set_kernel_params (3: the 11 hyper-parameters of CO2_example.py's covariance_function, whose
// kernel_4 adds theta_11^2 on the diagonal of a square matrix).
int gpmi_dev_cov_rows(void* stream, int kind, const double* params, int nparams, const double* X_dev, int64_t N, int64_t d,
                      int64_t row0, int64_t nrows, int64_t ncols, double noise_var, double* out_dev, int64_t ld) {
    if (!X_dev || !out_dev) return fail_arg("gpmi_dev_cov_rows: null pointer");
    if (nrows % TILE || ncols % TILE || row0 % TILE || ld < ncols || ld % 2)
        return fail_arg("gpmi_dev_cov_rows: sizes must be multiples of 128");
    RbfArgs r;
    r.A = r.B = X_dev; r.nA = r.nB = N; r.d = d; r.row0 = row0; r.nrows = nrows; r.ncols = ncols;
    const int rc = dev_cov_args(r, kind, params, nparams, "gpmi_dev_cov_rows");
    if (rc) return rc;
    r.diag_add = noise_var; r.symmetric = 1; r.delta_square = 1;
    r.out = out_dev; r.ld = ld;
    HIP_TRY(launch_rbf((hipStream_t)stream, r));
    return GPMI_OK;
}

// gpmi_dev_rbf_cross likewise: out[i][j] = k(Xs[i], Xcols[j]) for a WINDOW of the column inputs that starts at input
// col0 of the full set; square != 0 says the full cross matrix is square (n == N), in which case the composite kernel's
// delta term lands on i == col0 + j (CO2_example.py:58-62).  Rows >= n and columns >= ncols_real are zero.
int gpmi_dev_cov_cross(void* stream, int kind, const double* params, int nparams, const double* Xs_dev, int64_t n,
                       const double* Xcols_dev, int64_t ncols_real, int64_t d, int64_t col0, int square, int64_t nrows,
                       int64_t ncols, double* out_dev, int64_t ld) {
    if (!Xs_dev || !Xcols_dev || !out_dev) return fail_arg("gpmi_dev_cov_cross: null pointer");
    if (nrows % TILE || ncols % TILE || ld < ncols || ld % 2 || col0 < 0)
        return fail_arg("gpmi_dev_cov_cross: sizes must be multiples of 128");
    RbfArgs r;
    r.A = Xs_dev; r.B = Xcols_dev; r.nA = n; r.nB = ncols_real > 0 ? ncols_real : 0; r.d = d; r.row0 = 0; r.nrows = nrows; r.ncols = ncols;
    const int rc = dev_cov_args(r, kind, params, nparams, "gpmi_dev_cov_cross");
    if (rc) return rc;
    r.diag_add = 0.; r.symmetric = 0; r.delta_square = square ? 1 : 0; r.delta_col0 = col0;
    r.out = out_dev; r.ld = ld;
    HIP_TRY(launch_rbf((hipStream_t)stream, r));
    return GPMI_OK;
}

int gpmi_dev_potrf_block(void* stream, double* A_dev, int64_t ld, int64_t nb, int64_t col_offset,
                         int64_t* info_dev) {
    if (!A_dev || !info_dev) return fail_arg("gpmi_dev_potrf_block: null pointer");
    if (nb <= 0 || nb % TILE || ld % 2) return fail_arg("gpmi_dev_potrf_block: nb must be a multiple of 128");
    HIP_TRY(panel_factor((hipStream_t)stream, A_dev, ld, nb, nb, col_offset, info_dev));
    return GPMI_OK;
}

int gpmi_dev_trsm_block(void* stream, const double* L_dev, int64_t ldl, double* X_dev, int64_t ldx,
                        int64_t m, int64_t nb) {
    if (!L_dev || !X_dev) return fail_arg("gpmi_dev_trsm_block: null pointer");
    if (m < 0 || m % TILE || nb <= 0 || nb % IB || ldl % 2 || ldx % 2)
        return fail_arg("gpmi_dev_trsm_block: m must be a multiple of 128, nb of 64");
    if (m == 0) return GPMI_OK;
    HIP_TRY(trsm_block((hipStream_t)stream, L_dev, ldl, X_dev, ldx, m, nb));
    return GPMI_OK;
}

int gpmi_dev_gemm_nt(void* stream, double* C_dev, int64_t ldc, const double* A_dev, int64_t lda,
                     const double* B_dev, int64_t ldb, int64_t M, int64_t N, int64_t K, int lower,
                     int64_t diag_off) {
    if (!C_dev || !A_dev || !B_dev) return fail_arg("gpmi_dev_gemm_nt: null pointer");
    if (M < 0 || N < 0 || K < 0 || M % TILE || N % IB || K % 16 || ldc % 2 || lda % 2 || ldb % 2)
        return fail_arg("gpmi_dev_gemm_nt: M%128, N%64, K%16 must be 0");
    GemmArgs g;
    g.C = C_dev; g.A = A_dev; g.B = B_dev; g.ldc = ldc; g.lda = lda; g.ldb = ldb;
    g.M = M; g.N = N; g.K = K; g.mode = 0; g.lower = lower; g.diag_off = diag_off;
    HIP_TRY(launch_gemm_nt((hipStream_t)stream, g));
    return GPMI_OK;
}

int gpmi_dev_gemm_nt_rowmap(void* stream, double* C_dev, int64_t ldc, const double* A_dev, int64_t lda,
                            const double* B_dev, int64_t ldb, int64_t M, int64_t N, int64_t K,
                            const int32_t* row_ncols_dev, int64_t row_block_rows) {
    if (!C_dev || !A_dev || !B_dev || !row_ncols_dev) return fail_arg("gpmi_dev_gemm_nt_rowmap: null pointer");
    if (M < 0 || N < 0 || K < 0 || M % TILE || N % IB || K % 16 || ldc % 2 || lda % 2 || ldb % 2 ||
        row_block_rows <= 0 || row_block_rows % TILE)
        return fail_arg("gpmi_dev_gemm_nt_rowmap: M%128, N%64, K%16, row_block_rows%128 must be 0");
    GemmArgs g;
    g.C = C_dev; g.A = A_dev; g.B = B_dev; g.ldc = ldc; g.lda = lda; g.ldb = ldb;
    g.M = M; g.N = N; g.K = K; g.mode = 0; g.lower = 0; g.diag_off = 0;
    g.row_ncols = row_ncols_dev; g.row_block_tiles = (int)(row_block_rows / TILE);
    HIP_TRY(launch_gemm_nt((hipStream_t)stream, g));
    return GPMI_OK;
}

// The same with a host copy of the row map (row_bands entries): the launcher then enumerates only the
// supertiles that hold live tiles (a rectangle half full of skipped workgroups runs 15 % slower).
int gpmi_dev_gemm_nt_rowmap_host(void* stream, double* C_dev, int64_t ldc, const double* A_dev, int64_t lda,
                                 const double* B_dev, int64_t ldb, int64_t M, int64_t N, int64_t K,
                                 const int32_t* row_ncols_dev, const int32_t* row_ncols_host, int64_t row_bands,
                                 int64_t row_block_rows) {
    if (!C_dev || !A_dev || !B_dev || !row_ncols_dev || !row_ncols_host) return fail_arg("gpmi_dev_gemm_nt_rowmap_host: null pointer");
    if (M < 0 || N < 0 || K < 0 || M % TILE || N % IB || K % 16 || ldc % 2 || lda % 2 || ldb % 2 ||
        row_block_rows <= 0 || row_block_rows % TILE || row_bands * row_block_rows < M)
        return fail_arg("gpmi_dev_gemm_nt_rowmap_host: M%128, N%64, K%16, row_block_rows%128 must be 0 and the map must cover M");
    GemmArgs g;
    g.C = C_dev; g.A = A_dev; g.B = B_dev; g.ldc = ldc; g.lda = lda; g.ldb = ldb;
    g.M = M; g.N = N; g.K = K; g.mode = 0; g.lower = 0; g.diag_off = 0;
    g.row_ncols = row_ncols_dev; g.row_block_tiles = (int)(row_block_rows / TILE);
    g.row_ncols_host = row_ncols_host; g.row_bands = (int)row_bands;
    HIP_TRY(launch_gemm_nt((hipStream_t)stream, g));
    return GPMI_OK;
}

// The row-map update with B given as a table of row blocks: block i of B (b_block_rows x K, leading dimension
// ldb) starts at B_dev + b_block_off_dev[i] doubles.  This is how the multi-rank driver reads the panel column
// straight out of the all-gather's receive buffer (one contiguous chunk per rank) in natural block order,
// with no re-ordering copy.  row_ncols_dev / row_ncols_host may both be NULL (plain rectangle).
int gpmi_dev_gemm_nt_blocks(void* stream, double* C_dev, int64_t ldc, const double* A_dev, int64_t lda,
                            const double* B_dev, int64_t ldb, const int64_t* b_block_off_dev, int64_t b_block_rows,
                            int64_t M, int64_t N, int64_t K, const int32_t* row_ncols_dev,
                            const int32_t* row_ncols_host, int64_t row_bands, int64_t row_block_rows) {
    if (!C_dev || !A_dev || !B_dev || !b_block_off_dev) return fail_arg("gpmi_dev_gemm_nt_blocks: null pointer");
    if (M < 0 || N < 0 || K < 0 || M % TILE || N % TILE || K % 16 || K < 32 || ldc % 2 || lda % 2 || ldb % 2 ||
        b_block_rows <= 0 || b_block_rows % TILE)
        return fail_arg("gpmi_dev_gemm_nt_blocks: M%128, N%128, K%16 (K >= 32), b_block_rows%128 must be 0");
    if ((row_ncols_dev != nullptr) != (row_ncols_host != nullptr))
        return fail_arg("gpmi_dev_gemm_nt_blocks: the row map needs both its device and its host copy");
    if (row_ncols_dev && (row_block_rows <= 0 || row_block_rows % TILE || row_bands * row_block_rows < M))
        return fail_arg("gpmi_dev_gemm_nt_blocks: row_block_rows%128 must be 0 and the map must cover M");
    GemmArgs g;
    g.C = C_dev; g.A = A_dev; g.B = B_dev; g.ldc = ldc; g.lda = lda; g.ldb = ldb;
    g.M = M; g.N = N; g.K = K; g.mode = 0; g.lower = 0; g.diag_off = 0;
    g.b_block_off = b_block_off_dev; g.b_block_rows = b_block_rows;
    if (row_ncols_dev) {
        g.row_ncols = row_ncols_dev; g.row_block_tiles = (int)(row_block_rows / TILE);
        g.row_ncols_host = row_ncols_host; g.row_bands = (int)row_bands;
    }
    HIP_TRY(launch_gemm_nt((hipStream_t)stream, g));
    return GPMI_OK;
}

int gpmi_dev_logdiag_sumsq(void* stream, const double* A_dev, int64_t ld, int64_t n, const double* x_dev,
                           int64_t nx, double* out2_dev) {
    if (!out2_dev) return fail_arg("gpmi_dev_logdiag_sumsq: null output");
    HIP_TRY(launch_logdiag_sumsq((hipStream_t)stream, A_dev, ld, n, x_dev, nx, out2_dev));
    return GPMI_OK;
}

int gpmi_dev_gemv_t(void* stream, const double* A_dev, int64_t ld, int64_t nrows, int64_t ncols,
                    const double* x_dev, double* y_dev, double* scratch_dev) {
    if (!y_dev || !scratch_dev || (nrows > 0 && (!A_dev || !x_dev))) return fail_arg("gpmi_dev_gemv_t: null pointer");
    if (nrows < 0 || ncols < 0) return fail_arg("gpmi_dev_gemv_t: negative size");
    HIP_TRY(launch_gemv_t((hipStream_t)stream, A_dev, ld, nrows, ncols, x_dev, y_dev, scratch_dev));
    return GPMI_OK;
}

int gpmi_dev_trsv_lt(void* stream, const double* L_dev, int64_t ld, double* b_dev, int64_t n) {
    if (!L_dev || !b_dev) return fail_arg("gpmi_dev_trsv_lt: null pointer");
    if (n <= 0 || n % IB) return fail_arg("gpmi_dev_trsv_lt: n must be a positive multiple of 64");
    HIP_TRY(launch_trsv_lt((hipStream_t)stream, L_dev, ld, b_dev, n));
    return GPMI_OK;
}

// backward substitution with a block as gpmi_dev_potrf_block leaves it (inverses in the diagonal tiles): 128 unknowns
// per launch instead of 64 per pair of launches.  b is destroyed, the solution goes to x_dev (n doubles, no alias).
int gpmi_dev_trsv_lt_fused(void* stream, const double* L_dev, int64_t ld, double* b_dev, double* x_dev, int64_t n) {
    if (!L_dev || !b_dev || !x_dev || b_dev == x_dev) return fail_arg("gpmi_dev_trsv_lt_fused: null or aliased pointer");
    if (n <= 0 || n % TILE || ld % 2) return fail_arg("gpmi_dev_trsv_lt_fused: n must be a positive multiple of 128, ld even");
    HIP_TRY(launch_trsv_lt_fused((hipStream_t)stream, L_dev, ld, b_dev, x_dev, n));
    return GPMI_OK;
}

// the same through the full inverses of the 128 x 128 diagonal blocks (launch_vinv128 writes them into the blocks' upper
// triangles when invert != 0; later calls on the same factored block pass 0): one product per 128 unknowns
int gpmi_dev_trsv_lt_vinv(void* stream, double* L_dev, int64_t ld, double* b_dev, double* x_dev, int64_t n, int invert) {
    if (!L_dev || !b_dev || !x_dev || b_dev == x_dev) return fail_arg("gpmi_dev_trsv_lt_vinv: null or aliased pointer");
    if (n <= 0 || n % TILE || ld % 2) return fail_arg("gpmi_dev_trsv_lt_vinv: n must be a positive multiple of 128, ld even");
    if (invert) HIP_TRY(launch_vinv128((hipStream_t)stream, L_dev, ld, n));
    HIP_TRY(launch_trsv_lt_vinv((hipStream_t)stream, L_dev, ld, b_dev, x_dev, n));
    return GPMI_OK;
}

int gpmi_dev_trsv_lt_chain(void* stream, double* L_dev, int64_t ld, double* vside_dev, const double* m_dev, double* x_dev,
                           int64_t n, int invert, int* err_dev) {
    if (!L_dev || !vside_dev || !m_dev || !x_dev || !err_dev || m_dev == x_dev)
        return fail_arg("gpmi_dev_trsv_lt_chain: null or aliased pointer");
    if (n <= 0 || n % TILE || ld % 2) return fail_arg("gpmi_dev_trsv_lt_chain: n must be a positive multiple of 128, ld even");
    if (invert) HIP_TRY(launch_vinv128((hipStream_t)stream, L_dev, ld, n, vside_dev));
    HIP_TRY(launch_trsv_lt_chain((hipStream_t)stream, L_dev, ld, vside_dev, m_dev, x_dev, n, err_dev));
    return GPMI_OK;
}

// Tell the block primitives called from this thread that they run beside a trailing update on another stream
// (the multi-rank driver's lookahead): the panel kernels then use their small-LDS forms (two-launch trsm128, shallow
// ring for small GEMMs), which fit on a CU next to an update workgroup and start at once.  Results are the
// same bits either way.  0 switches back.
int gpmi_dev_set_concurrent(int on) {
    static thread_local GemmShallowScope* scope = nullptr;
    if (on && !scope) scope = new GemmShallowScope(true);
    if (!on && scope) { delete scope; scope = nullptr; }
    return GPMI_OK;
}

// One row chunk of the gradient trace (tune_hyperparms_regression.py:43-57) on device pointers:
//   out2[0] += sum_ij W_ij dK_ij/dl,  out2[1] += sum_ij W_ij dK_ij/dsigma,  W_ij = alpha_r[i] alpha_c[j] - kinv_sign * Kinv[i - row0][j]
// over rows row0 .. row0 + nrows and ALL N columns (dK recomputed from X).  The multi-rank driver calls it once per
// row block with its own partial of -K_y^-1 (and alpha only on one rank), so nothing N x N is ever summed across
// ranks.  partial_dev: workspace of 2 * ceil(nrows / 128) * ceil(N / 128) doubles; out2 is accumulated
// (fixed order) on the stream.
int gpmi_dev_grad_trace(void* stream, const double* X_dev, int64_t N, int64_t d, int64_t row0, int64_t nrows,
                        const double* alpha_r_dev, const double* alpha_c_dev, const double* Kinv_dev, int64_t ld,
                        double kinv_sign, double sigma, double ell, double* partial_dev, double* out2_dev) {
    if (!X_dev || !alpha_r_dev || !alpha_c_dev || !Kinv_dev || !partial_dev || !out2_dev)
        return fail_arg("gpmi_dev_grad_trace: null pointer");
    if (N <= 0 || d <= 0 || row0 < 0 || nrows <= 0 || row0 + nrows > N || ld < N || !(ell != 0.0))
        return fail_arg("gpmi_dev_grad_trace: bad dimensions");
    GradArgs g;
    g.A = g.B = X_dev; g.nA = g.nB = N; g.d = d; g.row0 = row0; g.nrows = nrows;
    g.alpha_r = alpha_r_dev; g.alpha_c = alpha_c_dev;
    g.Kinv = Kinv_dev; g.ld = ld; g.kinv_sign = kinv_sign;
    g.coef = -.5 * (1 / (ell * ell)); g.sig2 = sigma * sigma; g.two_sigma = 2 * sigma;
    g.inv_l3 = 1.0 / (ell * ell * ell);
    g.tri = 0;
    g.partial = partial_dev;
    HIP_TRY(launch_grad_trace((hipStream_t)stream, g));
    HIP_TRY(launch_sum_pairs((hipStream_t)stream, partial_dev, grad_trace_blocks(g), out2_dev));
    return GPMI_OK;
}

// Fixed-order sum of `count` contributions (stride doubles apart) onto an optional base vector:
//   out[i] = (base ? base[i] : 0) + scale * (in[i] + in[stride + i] + ... ), i < n, added in index order.
// The partitioned path's reductions over gathered per-rank partials (the backward solve's right-hand side
// m_k - sum_r part_r, GP_regression.py:140; the log-determinant pieces of tune_hyperparms_regression.py:312):
// same bits on every rank, no dependence on a library's reduction tree.  out may alias base.
int gpmi_dev_sum_fixed(void* stream, const double* in_dev, int64_t count, int64_t stride, int64_t n,
                       const double* base_dev, double scale, double* out_dev) {
    if (!out_dev || (count > 0 && !in_dev)) return fail_arg("gpmi_dev_sum_fixed: null pointer");
    if (count < 0 || n < 0 || stride < 0) return fail_arg("gpmi_dev_sum_fixed: negative size");
    HIP_TRY(launch_sum_fixed((hipStream_t)stream, in_dev, count, stride, n, base_dev, scale, out_dev));
    return GPMI_OK;
}

// Y (rows x cols, ldy) += a * X (rows x cols, ldx) -- the assembly K_ss + jitter * I - v^T v of the posterior
// covariance on the partitioned path (GP_regression.py:154: the all-reduced -v^T v added onto the covariance rows)
int gpmi_dev_axpy2d(void* stream, double* Y_dev, int64_t ldy, const double* X_dev, int64_t ldx, int64_t rows,
                    int64_t cols, double a) {
    if (!Y_dev || !X_dev) return fail_arg("gpmi_dev_axpy2d: null pointer");
    if (rows < 0 || cols < 0 || ldy < cols || ldx < cols) return fail_arg("gpmi_dev_axpy2d: bad dimensions");
    HIP_TRY(launch_axpy2d((hipStream_t)stream, Y_dev, ldy, X_dev, ldx, rows, cols, a));
    return GPMI_OK;
}

int gpmi_dev_row_dots(void* stream, const double* V_dev, int64_t ld, int64_t nrows, int64_t ncols,
                      const double* m_dev, double* dot_out_dev, double* sq_out_dev) {
    if (!V_dev || !m_dev) return fail_arg("gpmi_dev_row_dots: null pointer");
    if (ncols % 2 || ld % 2) return fail_arg("gpmi_dev_row_dots: ncols and ld must be even");
    HIP_TRY(launch_row_dots((hipStream_t)stream, V_dev, ld, nrows, ncols, m_dev, dot_out_dev, sq_out_dev));
    return GPMI_OK;
}

}  // extern "C"
