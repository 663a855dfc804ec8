/*
 * gpmi.h -- C-ABI of libgpmi355x.so, the MI355X (gfx950) implementation of the
 * GP-regression hot path of happyjin/Gaussian_process.
 *
 * The reference has no FFI/plugin interface: its boundary is a set of plain
 * Python functions other scripts import.  Each entry point
 * below names the reference statement(s) it replaces; the Python shim in
 * gaussian_process_amd/ binds them with ctypes and keeps the reference's
 * function signatures.  All matrices are float64, row-major, caller-owned host
 * buffers unless the name says `_dev` (device pointers for the multi-GPU
 * driver).  No torch types cross this boundary.
 *
 * Every function returns an int status:
 *   GPMI_OK            0
 *   GPMI_ERR_NOT_PD    1  Cholesky met a non-positive pivot (reference:
 *                         numpy.linalg.LinAlgError from np.linalg.cholesky,
 *                         GP_regression.py:138,154); *bad_pivot = 1-based index
 *   GPMI_ERR_BAD_ARG   2  -> ValueError in the shim
 *   GPMI_ERR_RUNTIME   3  HIP runtime failure -> RuntimeError; text in
 *                         gpmi_last_error()
 * Nothing aborts the process.  A context is not thread-safe (one per thread).
 */
#ifndef GPMI_H
#define GPMI_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define GPMI_OK 0
#define GPMI_ERR_NOT_PD 1
#define GPMI_ERR_BAD_ARG 2
#define GPMI_ERR_RUNTIME 3

#define GPMI_ABI_VERSION 1

/* stage timer slots filled by gpmi_get_timers (milliseconds, hipEvent-timed on
 * the context's compute stream; 0 when the stage did not run in the last call) */
enum {
    GPMI_T_KBUILD = 0,    /* a1+a2: K(X,X)+s*I lower tiles                */
    GPMI_T_CHOL = 1,      /* a3 (+a4 folded in): blocked Cholesky, total   */
    GPMI_T_CHOL_PANEL = 2,/*    of which: diagonal block + panel TRSM      */
    GPMI_T_CHOL_TRAIL = 3,/*    of which: trailing SYRK/GEMM updates       */
    GPMI_T_LML = 4,       /* a10: reductions for the log-marginal-likelihood */
    GPMI_T_KS = 5,        /* a1 for K(X*,X) (transposed K_s)               */
    GPMI_T_SOLVE_V = 6,   /* a7: v = L^-1 K_s (TRSM sweep), total          */
    GPMI_T_MEANVAR = 7,   /* a6+a8: mean / variance reductions             */
    GPMI_T_ALPHA = 8,     /* a5: backward solve L^T alpha = m              */
    GPMI_T_POSTCHOL = 9,  /* f1: v^T v, K** + jitter*I - v^T v, its Cholesky */
    GPMI_T_TRAIL_LAUNCHES = 10, /* number of trailing-update launches in last fit */
    GPMI_T_TRAIL_FLOPS = 11,    /* algorithmic flops of those launches: 2K per element on or below the diagonal, real rows + the y row */
    GPMI_T_GRAD = 12,     /* f2: L^-T, K_y^-1 and the fused gradient trace */
    GPMI_T_COUNT = 16
};

typedef struct gpmi_ctx gpmi_ctx;

int gpmi_abi_version(void);
/* text of the last GPMI_ERR_RUNTIME / GPMI_ERR_BAD_ARG on this thread */
const char* gpmi_last_error(void);
int gpmi_device_count(int* count);

/* one context = one GPU (device ordinal) + its streams and workspaces */
int gpmi_ctx_create(int device, gpmi_ctx** out);
int gpmi_ctx_destroy(gpmi_ctx* ctx);
/* tuning knobs; unknown names -> GPMI_ERR_BAD_ARG.
 * per context:  "nb" (outer Cholesky block, 0 = by size), "ld_pad" (doubles added to leading dimensions),
 *               "timing" (0/1: hipEvent stage timers), "lookahead" (0/1), "la_min" (columns from which lookahead is used,
 *               default 12288), "one_pass_form" (gpmi_fit_predict_resident: 0 by size, 1 the test rows ride inside the
 *               panel and update launches, 2 they follow on a stream of their own), "lanes" (factorisations in flight in
 *               gpmi_lml_batch, 0 = by size), "ramp" (bit mask, default 0: 1 block widths ramp up at the start of the
 *               sweep, 2 half width over the last blocks, 4 quarter width for the last one, bits 4.. = how many blocks
 *               count as "last" (0: three), when "nb" is automatic.  Until round 2 any non-zero value meant "up and
 *               down": that is 3 now; other bits or a negative value are refused),
 *               "shallow_min" (under lookahead, panels with fewer columns left than this use the one-launch panel
 *               kernels: the update they would run beside is over long before they are; default 6144, 0 = never);
 *               kernel selection (for measurements; also per context -- the lanes of gpmi_lml_batch inherit them):
 *               "panel_fused" (0/1: 128-column MFMA panel kernels / first-generation 64-column leaves),
 *               "gemm_dma" (0/1), "gemm_dma_waves" (4/8), "gemm_small_tiles" (0/1), "gemm_small_dma" (0/1),
 *               "gemm_persist" (0/1: resident workgroups for update GEMMs that have the chip to themselves),
 *               "trsv_vinv" (backward solve: 2 ONE launch, column blocks chained through the solution vector, with the
 *               inverted 128 x 128 diagonal blocks -- the default; 1 one launch per block with the same inverses; 0 the
 *               16 x 16 rounds),
 *               "trsm_wave" (0/1), "rbf_blocks" (persistent blocks of the K build).
 * The context-free gpmi_dev_* primitives run with the defaults. */
int gpmi_set_option(gpmi_ctx* ctx, const char* name, int64_t value);

/* RBF_kernel(a, b, sigma, l)                         GP_regression.py:8-19
 *   out[i*M + j] = sigma^2 * exp(-.5 * (1/l^2) * sum_k (a[i,k]-b[j,k])^2)
 * a: N x d, b: M x d, out: N x M. */
int gpmi_rbf(gpmi_ctx* ctx, const double* a, int64_t N, const double* b, int64_t M,
             int64_t d, double sigma, double ell, double* out);

/* The reference's other covariance functions, same layout as gpmi_rbf:
 *   kind 0: RBF_kernel (p0 = sigma, p1 = l)                               GP_regression.py:8-19
 *   kind 1: lin_kernel(a, b, c): np.dot(a - c, b.T - c), p0 = c            GP_regression.py:22-33
 *   kind 2: per_kernel(a, b, (p, l)): exp(-2 sin(pi|a-b|/p)^2 / l^2), d = 1, p0 = p, p1 = l   :36-50 */
int gpmi_cov(gpmi_ctx* ctx, int kind, const double* a, int64_t N, const double* b, int64_t M,
             int64_t d, double p0, double p1, double* out);
/* Covariance function used by gpmi_factorize / gpmi_predict / gpmi_post_chol from now on
 * (kernel_choice of prediction(), GP_regression.py:125-136).  kind 0 (default) takes sigma and l
 * from gpmi_factorize; kinds 1, 2 take (p0, p1) as above and ignore them. */
int gpmi_set_kernel(gpmi_ctx* ctx, int kind, double p0, double p1);
/* The composite covariance of the CO2 example (second half):
 *   kind 3: covariance_function(a, b, hyperparms) = kernel_1 + kernel_2 + kernel_3 + kernel_4,
 *           hyperparms = theta_1..theta_11, any d                            CO2_example.py:9-94
 * kernel_4 adds theta_11^2 on row == col whenever the matrix is square (N == M), as the reference
 * does (:58-59).  The *_params forms take the parameters as an array; kinds 0-2 accept
 * nparams == 2 with (p0, p1) as above. */
int gpmi_cov_params(gpmi_ctx* ctx, int kind, const double* a, int64_t N, const double* b, int64_t M,
                    int64_t d, const double* params, int nparams, double* out);
int gpmi_set_kernel_params(gpmi_ctx* ctx, int kind, const double* params, int nparams);

/* Copy the training set to the device (X: N x d, y: N).  Replaces nothing in
 * the reference (it has no device); separates PCIe from the timed path. */
int gpmi_set_train(gpmi_ctx* ctx, const double* X, int64_t N, int64_t d, const double* y);

/* K = RBF_kernel(X,X,sigma,l); L = cholesky(K + s*I); m = solve(L, y)
 *                    GP_regression.py:126,138-139; tune_hyperparms_regression.py:306-308
 * and the log-marginal-likelihood of tune_hyperparms_regression.py:312:
 *   lml = -.5*y^T alpha - sum(log(diag L)) - N/2*log(2*pi)   (y^T alpha = m^T m)
 * Leaves L and m resident in the context.  lml/bad_pivot may be NULL. */
int gpmi_factorize(gpmi_ctx* ctx, double sigma, double ell, double noise_var,
                   double* lml, int64_t* bad_pivot);

/* gpmi_set_train + gpmi_factorize in one call (host buffers in). */
int gpmi_fit(gpmi_ctx* ctx, const double* X, int64_t N, int64_t d, const double* y,
             double sigma, double ell, double noise_var, double* lml, int64_t* bad_pivot);

/* alpha = solve(L.T, m)                               GP_regression.py:140
 * out: N doubles. */
int gpmi_get_alpha(gpmi_ctx* ctx, double* alpha_out);
/* m = solve(L, y) (GP_regression.py:139) and diag(L); each N doubles. */
int gpmi_get_m(gpmi_ctx* ctx, double* m_out);
int gpmi_get_diag(gpmi_ctx* ctx, double* diag_out);
/* rows [r0,r1) x cols [c0,c1) of the resident factor, lower part, zeros above
 * the diagonal (np.linalg.cholesky convention); for tests and small N. */
int gpmi_get_factor_block(gpmi_ctx* ctx, int64_t r0, int64_t r1, int64_t c0, int64_t c1,
                          double* out);

/* Copy test inputs to the device (Xs: n x d, d as in set_train). */
int gpmi_set_test(gpmi_ctx* ctx, const double* Xs, int64_t n);

/* K_s = RBF_kernel(X, Xs); mu = K_s.T @ alpha; v = solve(L, K_s);
 * var = diag(K_ss) - sum(v**2, 0); sd = sqrt(var)     GP_regression.py:127,143-148
 * mu, out2: n doubles each; out2 = sd if want_sd else var.  A negative var
 * gives NaN sd, as np.sqrt does in the reference (no clamp).  Either output
 * may be NULL.  v stays resident (transposed, n x N) for gpmi_post_chol. */
int gpmi_predict_resident(gpmi_ctx* ctx, double* mu, double* out2, int want_sd);
int gpmi_predict(gpmi_ctx* ctx, const double* Xs, int64_t n, double* mu, double* out2,
                 int want_sd);

/* prediction() in one pass: gpmi_factorize + gpmi_predict_resident for a training and a test set that are both
 * resident (gpmi_set_train, gpmi_set_test), GP_regression.py:109-156 (a1-a8).  K(X*, X) rides below the y row through
 * the Cholesky, so a7 (v = L^-1 K_s) has no launches of its own.  lml / bad_pivot as gpmi_factorize, mu / out2 / want_sd
 * as gpmi_predict_resident; alpha, post_chol and lml_grad work on the result as after the two calls.  Results agree with
 * the two-call form to rounding (not bit for bit: the two sweeps' block widths differ). */
int gpmi_fit_predict_resident(gpmi_ctx* ctx, double sigma, double ell, double noise_var, double* lml, int64_t* bad_pivot,
                              double* mu, double* out2, int want_sd);

/* ... and with the posterior-sample factor of GP_regression.py:154 in the same pass: one Cholesky of the augmented matrix
 *   [[K + sI, .], [K(X*, X), K_ss + jitter I]]   (N + n columns, the y rows below): its last n columns are
 *   L_ = cholesky(K_ss + jitter I - v.T @ v), the Schur complement the trailing updates leave there.  Arguments as
 * gpmi_fit_predict_resident plus jitter and L_out (n x n row-major, zeros above the diagonal; may be NULL -- gpmi_post_chol
 * with the same jitter then only downloads it).  GPMI_ERR_NOT_PD for a pivot of K + sI (:138) or of the posterior covariance
 * (:154; bad_pivot then counts from the posterior's own first column).  LML, alpha, mean and variance as the other forms to
 * rounding (the leading dimension and the block boundaries differ); L_ as gpmi_post_chol to rounding. */
int gpmi_fit_predict_sample_resident(gpmi_ctx* ctx, double sigma, double ell, double noise_var, double jitter, double* lml,
                                     int64_t* bad_pivot, double* mu, double* out2, int want_sd, double* L_out);

/* L_ = cholesky(K_ss + jitter*I - v.T @ v)            GP_regression.py:154
 * for the test set of the last predict; L_out: n x n row-major, zeros above the
 * diagonal.   */
int gpmi_post_chol(gpmi_ctx* ctx, double jitter, double* L_out, int64_t* bad_pivot);

/* L_ @ Z for f_post = mu + L_ @ normals (GP_regression.py:155) without bringing L_ (n x n) to the host: Z (n x num_fun
 * row-major; the caller's normals, drawn on the host so that np.random's order stays the reference's) goes up, LZ_out
 * (n x num_fun) comes down.  L_ is the factor gpmi_post_chol(jitter) returns -- the resident one when it rode through
 * gpmi_fit_predict_sample_resident or an earlier call formed it for this jitter, else formed now (status and bad_pivot as
 * gpmi_post_chol).  Row sums are added in a fixed order (bitwise reproducible); against np.dot they differ by rounding. */
int gpmi_post_sample(gpmi_ctx* ctx, double jitter, const double* Z, int64_t num_fun, double* LZ_out, int64_t* bad_pivot);

/* Gradient of the log marginal likelihood at the resident factorisation:
 *   d_ell   = .5 * trace((alpha alpha^T - K_y^-1) @ l_grad),     l_grad     = sigma^2 exp(-.5 sqdist/l^2) sqdist/l^3
 *                                                     tune_hyperparms_regression.py:54-57
 *   d_sigma = .5 * trace((alpha alpha^T - K_y^-1) @ sigma_grad), sigma_grad = 2 sigma exp(-.5 sqdist/l^2)
 *                                                     tune_hyperparms_regression.py:46-51 (commented out there)
 * with K_y^-1 = inv(L.T) @ inv(L) (:144) formed on the device from the resident L.
 * Squared-exponential kernel only. */
int gpmi_lml_grad(gpmi_ctx* ctx, double* d_ell, double* d_sigma);
/* The same two traces from the arguments gradient_ascent(a, b, sigma, l, alpha, K_y) receives
 * (tune_hyperparms_regression.py:31): a, b: N x d; alpha: N; K_y_inv: N x N row-major (host).
 * One fused N^2 pass instead of the reference's two N x N products (:55). */
int gpmi_grad_trace(gpmi_ctx* ctx, const double* a, const double* b, int64_t N, int64_t d, double sigma,
                    double ell, const double* alpha, const double* K_y_inv, double* d_ell, double* d_sigma);

/* compute_mar_likelihood for T hyper-parameter triples on one training set
 *                    tune_hyperparms_regression.py:292-313 called in the loops at :368-369,:385-386
 * triples: T x 3 = (ell, sigma_f, noise_var).  lml_out: T doubles (NaN where the
 * factorisation failed), status_out: T ints (GPMI_OK / GPMI_ERR_NOT_PD), may be
 * NULL.  Uses the training set of gpmi_set_train.  Up to "lanes" factorisations are in flight at once
 * (extra lanes = internal contexts with their own streams and workspaces; by size: 2 up to N = 32768,
 * else 1): the latency-bound panel steps of one fill with the MFMA work of another.
 * Each triple is factored by the same launches whatever the lane count, so the results are bitwise
 * independent of it; the factor left resident is that of the last triple. */
int gpmi_lml_batch(gpmi_ctx* ctx, const double* triples, int64_t T, double* lml_out,
                   int* status_out);

int gpmi_get_timers(gpmi_ctx* ctx, double* stage_ms, int count);
/* block the host until everything queued on the context has finished */
int gpmi_sync(gpmi_ctx* ctx);

/* ---- micro-benchmarks used by bench.py to re-read chip peaks on the box ---- */
/* fp64 MFMA issue rate: returns achieved TFLOP/s of a register-only
 * v_mfma_f64_16x16x4_f64 loop over the whole chip. */
int gpmi_probe_mfma_f64(gpmi_ctx* ctx, double* tflops);
/* same loop with one workgroup of 4 * blocks_per_cu waves on every CU (blocks_per_cu = waves per SIMD, 1 .. 4) and nacc
 * (4, 8, 16) independent accumulators per wave; out[0] = TFLOP/s, out[1] = shader clock in
 * GHz held during the loop, out[2] = shader cycles per MFMA per SIMD */
int gpmi_probe_mfma_f64_ex(gpmi_ctx* ctx, int blocks_per_cu, int nacc, int iters, double* out);
/* one launch shape of the trailing-update GEMM on scratch buffers; variant = timing-only
 * ablation bits (0 = the production kernel); out[0] = TFLOP/s, out[1] = ms per launch */
int gpmi_probe_gemm(gpmi_ctx* ctx, int64_t M, int64_t N, int64_t K, int lower, int variant, int reps,
                    double* out);
/* one resident workgroup that does nothing (threads, lds_bytes of untouched LDS), asleep for `milliseconds` on a stream of its
 * own (high_priority != 0: highest stream priority); returns at once -- time gpmi_probe_gemm while it is resident.
 * poll_sleep > 0: instead of sleeping, thread 0 re-reads a device flag with agent-scope atomic loads, s_sleep(poll_sleep)
 * between two reads (how a flag-chained resident kernel waits); fences != 0: plus an agent-scope acquire + release every ~30 us */
int gpmi_probe_resident(gpmi_ctx* ctx, int high_priority, int lds_bytes, int threads, double milliseconds, int poll_sleep,
                        int fences);
/* streaming-store bandwidth (GB/s) over `bytes` of device memory */
int gpmi_probe_hbm_write(gpmi_ctx* ctx, int64_t bytes, double* gbps);
/* streaming bandwidth with a chosen access form: mode 0 grid-stride 16-byte stores, 1 the same
 * non-temporal, 2 one contiguous span per workgroup, 3 span + non-temporal, 4 16-byte loads */
int gpmi_probe_hbm_ex(gpmi_ctx* ctx, int64_t bytes, int mode, int blocks, double* gbps);
/* what the device reports about itself: out[0] compute units, [1] shader clock kHz, [2] memory clock kHz,
 * [3] memory bus width (bits), [4] global memory bytes, [5] L2 bytes, [6] LDS per workgroup bytes, [7] wavefront size */
int gpmi_device_info(gpmi_ctx* ctx, double* out, int count);
/* the panel kernels alone on scratch data: kind 0 = Cholesky of one 128 x 128 block (potrf128), kind 1 = X L^-T on
 * m rows (trsm128); *out_us = microseconds per launch; stamps_out (64 entries or NULL) = in-kernel clock stamps
 * of one instrumented launch (layout: csrc/panel_mfma.hip) */
int gpmi_probe_panel(gpmi_ctx* ctx, int kind, int64_t m, int reps, double* out_us, uint64_t* stamps_out);
/* diagnostic: gpmi_probe_gemm while the resident potrf128 server (option potrf_server, an experiment) sits on a CU with
 * the given mode bits and is never used; 0: no server */
int gpmi_probe_gemm_beside_server(gpmi_ctx* ctx, int64_t M, int64_t N, int64_t K, int lower, int variant, int reps, int mode,
                                  double* out);
/* diagnostic: n_high streams at the highest priority + n_norm at the default one, one sleeping one-wave kernel of
 * `milliseconds` on each; *wall_ms = time until all are done (about `milliseconds` when every stream has a hardware
 * queue of its own, a multiple when streams share one) */
int gpmi_probe_stream_overlap(gpmi_ctx* ctx, int n_high, int n_norm, double milliseconds, double* wall_ms);
/* diagnostic: `count` one-wave kernels of sleep_us microseconds each on a stream of their own (high_priority != 0: the
 * device's highest priority), back to back; kind 0 sleep only, 1 + an agent-scope release / acquire fence pair, 2 + an
 * agent-scope atomic store.  Returns at once: time gpmi_probe_gemm meanwhile to see what a second stream's kernel
 * boundaries cost a long-running GEMM. */
int gpmi_probe_launch_storm(gpmi_ctx* ctx, int high_priority, int count, double sleep_us, int kind);
/* diagnostic: the give-up path of the one-launch backward solve.  An n x n identity system whose bottom block is
 * deliberately never solved: every wait runs into its bound (wait_ms here, 10 s in the product), the kernel must set
 * its error word (*err_out = 1), leave NaN in the entries it waited for (x_out, n doubles) and RETURN
 * (*elapsed_ms: about wait_ms per dependent block in flight). */
int gpmi_probe_trsv_giveup(gpmi_ctx* ctx, int64_t n, double wait_ms, int* err_out, double* elapsed_ms, double* x_out);

/* ---------------------------------------------------------------------------
 * Device-pointer block primitives for the multi-GPU (row-block cyclic) driver
 * in gaussian_process_amd/dist.py.  `stream` is a hipStream_t passed as void*
 * (torch.cuda.current_stream().cuda_stream).  All leading dimensions in
 * doubles.  Sizes must be multiples of 64 (panel width) / 128 (rows).
 * ------------------------------------------------------------------------- */
/* rows [row0,row0+nrows) of K(X,X)+s*I, columns [0, row0+nrows), into out
 * (nrows x ld); columns beyond N and rows beyond N are the identity padding. */
int gpmi_dev_rbf_rows(void* stream, const double* X_dev, int64_t N, int64_t d,
                      int64_t row0, int64_t nrows, int64_t ncols, double sigma, double ell,
                      double noise_var, double* out_dev, int64_t ld);
/* rows [row0,row0+nrows) of K(Xs,X): out[i][j] = k(Xs[row0+i], X[j]), j < ncols */
int gpmi_dev_rbf_cross(void* stream, const double* Xs_dev, int64_t n, const double* X_dev,
                       int64_t N, int64_t d, int64_t row0, int64_t nrows, int64_t ncols,
                       double sigma, double ell, double* out_dev, int64_t ld);
/* The same two builds for every covariance function the reference's prediction() serves (GP_regression.py:125-136:
 * 'rbf' / 'lin' / 'per') and for CO2_example.py:66-90's composite -- f4 on the row-block partitioned path.
 * kind / params as gpmi_set_kernel (0: sigma, l; 1: c; 2: period, l -- 1-D inputs) and gpmi_set_kernel_params (3: the
 * 11 hyper-parameters; kernel_4 adds theta_11^2 on the diagonal of a square matrix).  gpmi_dev_cov_cross takes a WINDOW
 * of the column inputs that starts at input col0 of the full set; square != 0: the full cross matrix is square (n == N),
 * so the composite kernel's delta term lands on row == col0 + column (CO2_example.py:58-62). */
int gpmi_dev_cov_rows(void* stream, int kind, const double* params, int nparams, const double* X_dev, int64_t N, int64_t d,
                      int64_t row0, int64_t nrows, int64_t ncols, double noise_var, double* out_dev, int64_t ld);
int gpmi_dev_cov_cross(void* stream, int kind, const double* params, int nparams, const double* Xs_dev, int64_t n,
                       const double* Xcols_dev, int64_t ncols_real, int64_t d, int64_t col0, int square, int64_t nrows,
                       int64_t ncols, double* out_dev, int64_t ld);
/* in-place Cholesky of the nb x nb diagonal block (nb multiple of 128);
 * info_dev: int64 on the device, atomically min-ed with col_offset + failing
 * column (initialise to INT64_MAX).  On return the lower triangle holds L; the strict upper triangles of
 * the 16 x 16 tiles on the diagonal hold the transposed inverses of those tiles (storage nothing else reads),
 * which gpmi_dev_trsm_block uses. */
int gpmi_dev_potrf_block(void* stream, double* A_dev, int64_t ld, int64_t nb,
                         int64_t col_offset, int64_t* info_dev);
/* X (m x nb, ldx) <- X * L^-T with L the nb x nb lower factor (ldl) AS LEFT BY gpmi_dev_potrf_block / the
 * factorisation (nb multiple of 128: the 16 x 16 diagonal tiles carry their inverses above the diagonal; a
 * copy of the block must be a copy of the whole nb x nb square) */
int gpmi_dev_trsm_block(void* stream, const double* L_dev, int64_t ldl, double* X_dev,
                        int64_t ldx, int64_t m, int64_t nb);
/* C (M x N, ldc) -= A (M x K, lda) * B (N x K, ldb)^T.  lower != 0: only tiles
 * that intersect {col <= row + diag_off} are touched. */
int gpmi_dev_gemm_nt(void* stream, double* C_dev, int64_t ldc, const double* A_dev, int64_t lda,
                     const double* B_dev, int64_t ldb, int64_t M, int64_t N, int64_t K,
                     int lower, int64_t diag_off);
/* same, for a rank's STACKED row blocks (row-block cyclic storage): the 128-row tile
 * bands of row block q (row_block_rows rows each) update only the leading
 * row_ncols_dev[q] columns of C (int32 on the device) */
int gpmi_dev_gemm_nt_rowmap(void* stream, double* C_dev, int64_t ldc, const double* A_dev, int64_t lda,
                            const double* B_dev, int64_t ldb, int64_t M, int64_t N, int64_t K,
                            const int32_t* row_ncols_dev, int64_t row_block_rows);
/* the same with a host copy of the row map (row_bands entries covering M): only the supertiles that hold
 * live tiles are launched */
int gpmi_dev_gemm_nt_rowmap_host(void* stream, double* C_dev, int64_t ldc, const double* A_dev, int64_t lda,
                                 const double* B_dev, int64_t ldb, int64_t M, int64_t N, int64_t K,
                                 const int32_t* row_ncols_dev, const int32_t* row_ncols_host, int64_t row_bands,
                                 int64_t row_block_rows);
/* C (M x N) -= A (M x K) * B^T with B given as a TABLE of row blocks: block i (b_block_rows x K, leading dimension
 * ldb) starts at B_dev + b_block_off_dev[i] doubles (device array, ceil(N / b_block_rows) entries).  The
 * multi-rank driver reads the panel column this way straight from the all-gather's receive buffer (one
 * contiguous chunk per rank) in natural block order -- no re-ordering copy.  The row map is optional
 * (row_ncols_dev and row_ncols_host both NULL: plain rectangle). */
int gpmi_dev_gemm_nt_blocks(void* stream, double* C_dev, int64_t ldc, const double* A_dev, int64_t lda,
                            const double* B_dev, int64_t ldb, const int64_t* b_block_off_dev, int64_t b_block_rows,
                            int64_t M, int64_t N, int64_t K, const int32_t* row_ncols_dev,
                            const int32_t* row_ncols_host, int64_t row_bands, int64_t row_block_rows);
/* out2[0] = sum_{i<n} log(A[i][i]) (skipped if A_dev is NULL), out2[1] = sum_{i<nx} x[i]^2
 * (skipped if x_dev is NULL): the per-rank pieces of the log-marginal-likelihood */
int gpmi_dev_logdiag_sumsq(void* stream, const double* A_dev, int64_t ld, int64_t n, const double* x_dev,
                           int64_t nx, double* out2_dev);
/* y[c] = sum_r A[r][c] * x[r] for a row-major nrows x ncols block (fixed summation order);
 * scratch: ceil(nrows/64) * ncols doubles (ncols % 2 == 0 and ld % 2 == 0 take the 16-byte-load path, which uses
 * ceil(nrows/128) * ncols of them).  Piece of the distributed backward solve
 * (GP_regression.py:140): a rank's contribution L_jk^T alpha_j of its rows below block k. */
int gpmi_dev_gemv_t(void* stream, const double* A_dev, int64_t ld, int64_t nrows, int64_t ncols,
                    const double* x_dev, double* y_dev, double* scratch_dev);
/* backward substitution L^T x = b on an n x n lower block (x overwrites b), n % 64 == 0 */
int gpmi_dev_trsv_lt(void* stream, const double* L_dev, int64_t ld, double* b_dev, int64_t n);
/* the same for a block as gpmi_dev_potrf_block leaves it (inverses in its diagonal tiles), n % 128 == 0: 128 unknowns
 * per launch; b_dev is destroyed, the solution goes to x_dev (n doubles, must not alias b_dev) */
int gpmi_dev_trsv_lt_fused(void* stream, const double* L_dev, int64_t ld, double* b_dev, double* x_dev, int64_t n);
/* the same through the inverted 128 x 128 diagonal blocks: invert != 0 first writes L_kk^-T of every 128 x 128 diagonal
 * block into that block's upper triangle (storage nothing else reads; one launch), then -- and on every later call with
 * invert == 0 on the same factored block -- each step is one matrix-vector product with it.  a5 of the multi-rank driver
 * (GP_regression.py:140). */
int gpmi_dev_trsv_lt_vinv(void* stream, double* L_dev, int64_t ld, double* b_dev, double* x_dev, int64_t n, int invert);
/* the same in ONE launch: the 128-column blocks are chained through the solution vector itself (x_dev is filled with a
 * "not yet" NaN pattern, a block's consumers poll the entries they need), so no launch gap and no chip-wide update sits
 * between two blocks.  vside_dev: n * 128 doubles owned by the caller; invert != 0 first fills it with the inverses of the
 * 128 x 128 diagonal blocks (row-major, one launch; also written into the blocks' upper triangles as above), later calls
 * on the same factored block pass 0 and the same buffer.  m_dev is only read; x_dev must not alias it.  err_dev: one int
 * the kernel sets to 1 if a wait gave up (non-finite factor); zero it before the call.  L_dev and vside_dev must be
 * 16-byte aligned (the kernel reads both with 16-byte loads; an odd column offset of a view is refused).  The chain
 * advances only while the queue makes progress: workgroup b waits for workgroups < b, which the hardware dispatches
 * first; a wait is bounded by wall time (10 s), so a co-tenant that holds the card for a while reads as a slow solve,
 * never as a wrong one.  a5 (GP_regression.py:140). */
int gpmi_dev_trsv_lt_chain(void* stream, double* L_dev, int64_t ld, double* vside_dev, const double* m_dev, double* x_dev,
                           int64_t n, int invert, int* err_dev);
/* on != 0: the block primitives called from this thread run beside a trailing update on another stream (lookahead)
 * and use their small-LDS forms, which fit on a CU next to an update workgroup; same results.  0 switches back. */
int gpmi_dev_set_concurrent(int on);
/* kernel-selection options (the gpmi_set_option names that choose between kernel forms: "gemm_ticket", "gemm_reserve",
 * "gemm_persist", "panel_prio", ...) for the context-free block primitives called from THIS thread; same results
 * whatever the choice.  The multi-rank driver switches its large update launches to the ticket form with it. */
int gpmi_dev_set_option(const char* name, int64_t value);
/* f2 on device pointers, one row chunk of the gradient trace (tune_hyperparms_regression.py:43-57):
 *   out2[0] += sum_ij W_ij dK_ij/dl,  out2[1] += sum_ij W_ij dK_ij/dsigma,
 *   W_ij = alpha_r[i] alpha_c[j] - kinv_sign * Kinv[(i - row0) * ld + j],  rows row0 .. row0 + nrows, all N columns.
 * partial_dev: 2 * ceil(nrows / 128) * ceil(N / 128) doubles of workspace.  The multi-rank driver feeds it
 * its own partial of -K_y^-1 row block by row block (DistGP.lml_grad). */
int gpmi_dev_grad_trace(void* stream, const double* X_dev, int64_t N, int64_t d, int64_t row0, int64_t nrows,
                        const double* alpha_r_dev, const double* alpha_c_dev, const double* Kinv_dev, int64_t ld,
                        double kinv_sign, double sigma, double ell, double* partial_dev, double* out2_dev);
/* out[i] = sum_j V[i][j]*m[j] ; out2[i] = sum_j V[i][j]^2  (partial sums over
 * the columns this rank owns), i < nrows, j < ncols */
int gpmi_dev_row_dots(void* stream, const double* V_dev, int64_t ld, int64_t nrows,
                      int64_t ncols, const double* m_dev, double* dot_out_dev,
                      double* sq_out_dev);

/* out[i] = (base_dev ? base_dev[i] : 0) + scale * (in[i] + in[stride + i] + ... + in[(count - 1) * stride + i]),
 * i < n, the contributions added one after the other in index order (no reduction tree): the partitioned path's sums
 * over gathered per-rank partials -- the right-hand side m_k - sum_r part_r of the distributed backward solve
 * (GP_regression.py:140), the log-determinant pieces (tune_hyperparms_regression.py:312) -- come out with the same
 * bits on every rank.  out_dev may alias base_dev. */
int gpmi_dev_sum_fixed(void* stream, const double* in_dev, int64_t count, int64_t stride, int64_t n,
                       const double* base_dev, double scale, double* out_dev);
/* Y (rows x cols, ldy) += a * X (rows x cols, ldx): K_ss + jitter * I - v^T v on the partitioned path
 * (GP_regression.py:154), the all-reduced -v^T v added onto the covariance rows */
int gpmi_dev_axpy2d(void* stream, double* Y_dev, int64_t ldy, const double* X_dev, int64_t ldx, int64_t rows,
                    int64_t cols, double a);

/* ---------------------------------------------------------------------------
 * RCCL behind the C-ABI: the collectives of the row-block partitioned path (broadcast of a
 * factored diagonal block, all-gather of a panel column over xGMI, the small all-reduces) issued on the CALLER'S stream,
 * straight into librccl -- opened at run time (dlopen), never linked, so the library loads on hosts without it and binds
 * the copy of RCCL a process already carries (PyTorch ships one next to its HIP runtime).  One communicator must not be
 * used from two streams at once; the multi-rank driver keeps one per stream (gaussian_process_amd/dist.py: RcclComm).
 * All sizes in bytes except gpmi_comm_all_reduce.
 * ------------------------------------------------------------------------- */
typedef struct gpmi_comm gpmi_comm;
/* optional: the librccl to bind (absolute path); without it the first call looks for a copy already in the process,
 * then for librccl.so.1 on the loader's path ($GPMI_RCCL_LIB overrides) */
int gpmi_comm_load(const char* librccl_path);
/* which library was bound (out: cap bytes) and its ncclGetVersion (may be NULL) */
int gpmi_comm_library(char* out, int64_t cap, int* version);
/* rank 0: a fresh ncclUniqueId (128 bytes) to hand to every rank by any side channel */
int gpmi_comm_unique_id(char* id128);
/* collective over the `size` ranks that hold the same id: ncclCommInitRank on `device` */
int gpmi_comm_create(const char* id128, int rank, int size, int device, gpmi_comm** out);
int gpmi_comm_destroy(gpmi_comm* comm);
/* every rank's buf_dev (nbytes) <- root's */
int gpmi_comm_broadcast(gpmi_comm* comm, void* stream, void* buf_dev, int64_t nbytes, int root);
/* recv_dev (size * nbytes_per_rank) <- every rank's send_dev (nbytes_per_rank), in rank order */
int gpmi_comm_all_gather(gpmi_comm* comm, void* stream, const void* send_dev, void* recv_dev, int64_t nbytes_per_rank);
/* in place over `count` elements; dtype 0 float64, 1 int64; op 0 sum, 1 min, 2 max */
int gpmi_comm_all_reduce(gpmi_comm* comm, void* stream, void* buf_dev, int64_t count, int dtype, int op);

#ifdef __cplusplus
}
#endif
#endif /* GPMI_H */
