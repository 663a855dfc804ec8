// C-ABI of libgpmi355x.so (see include/gpmi.h) and the single-GPU drivers:
// blocked right-looking Cholesky, the TRSM sweep for v = L^-1 K_s, and the
// glue that keeps the reference's call surface (GP_regression.py:109-156,
// tune_hyperparms_regression.py:292-313) reachable through plain C.
//
// Data layout in HBM (one context):
//   A   (Np + 128) x ldA   row-major, ldA = Np + ld_pad.  Rows/cols < N hold
//       K + s*I (lower tiles only) and, after gpmi_factorize, L.  Np = N rounded
//       up to 128; the padding is the identity, so every block operation works
//       on whole tiles.  Row Np carries y: the factorisation treats it as one
//       more row of the panel, so when it finishes that row holds
//       m = L^-1 y (forward substitution folded into the sweep, no extra pass).
//   V   n_p x ldV          row-major, row i = K(x*_i, X) then (L^-1 K_s)[:, i]
//       (v transposed: both GEMM operands of the sweep stay K-contiguous).
//   P   n_p x ldP          posterior covariance / its factor (gpmi_post_chol).
#include "gpmi_ctx.h"

using namespace gpmi;

namespace gpmi {
thread_local std::string g_err;

// options of the context-free gpmi_dev_* primitives called from this thread (gpmi_dev_set_option); the defaults otherwise
static thread_local Tuning t_dev_tuning;
static thread_local const Tuning* t_tuning = nullptr;
const Tuning& tuning() { return t_tuning ? *t_tuning : t_dev_tuning; }
TuneScope::TuneScope(const Tuning* t) : prev(t_tuning) { t_tuning = t; }
TuneScope::~TuneScope() { t_tuning = prev; }
}

extern "C" {

int gpmi_abi_version(void) { return GPMI_ABI_VERSION; }
const char* gpmi_last_error(void) { return g_err.c_str(); }

int gpmi_device_count(int* count) {
    if (!count) return fail_arg("gpmi_device_count: null pointer");
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess) { *count = 0; return fail_runtime(e, "hipGetDeviceCount"); }
    *count = n;
    return GPMI_OK;
}

int gpmi_ctx_create(int device, gpmi_ctx** out) {
    if (!out) return fail_arg("gpmi_ctx_create: null out pointer");
    *out = nullptr;
    int n = 0;
    HIP_TRY(hipGetDeviceCount(&n));
    if (device < 0 || device >= n) return fail_arg("gpmi_ctx_create: no such device");
    HIP_TRY(hipSetDevice(device));
    gpmi_ctx* c = new gpmi_ctx();
    c->device = device;
    int prio_lo = 0, prio_hi = 0;
    (void)hipDeviceGetStreamPriorityRange(&prio_lo, &prio_hi);
    hipError_t e = hipStreamCreateWithPriority(&c->stream, hipStreamNonBlocking, prio_lo);
    if (e != hipSuccess) { delete c; return fail_runtime(e, "hipStreamCreate"); }
    e = hipStreamCreateWithPriority(&c->pstream, hipStreamNonBlocking, prio_hi);
    if (e != hipSuccess) { (void)hipStreamDestroy(c->stream); delete c; return fail_runtime(e, "hipStreamCreate"); }
    const char* env;
    if ((env = getenv("GPMI_NB"))) c->nb = std::max<int64_t>(128, atoll(env) / 128 * 128);
    if ((env = getenv("GPMI_LD_PAD"))) c->ld_pad = std::max<int64_t>(0, atoll(env) / 2 * 2);
    if ((env = getenv("GPMI_LOOKAHEAD"))) c->lookahead = atoi(env) ? 1 : 0;
    *out = c;
    return GPMI_OK;
}

int gpmi_ctx_destroy(gpmi_ctx* c) {
    if (!c) return GPMI_OK;
    for (gpmi_ctx* l : c->lane_ctx) (void)gpmi_ctx_destroy(l);
    c->lane_ctx.clear();
    (void)hipSetDevice(c->device);
    (void)hipStreamSynchronize(c->stream);
    for (DevBuf* b : {&c->X, &c->y, &c->A, &c->info, &c->red, &c->Xs, &c->V, &c->P, &c->vec, &c->dense,
                       &c->U, &c->Kn, &c->gpart, &c->cov_a, &c->cov_b, &c->cov_out, &c->flag, &c->vside, &c->pmail})
        b->release();
    for (auto e : c->ev_pool) (void)hipEventDestroy(e);
    (void)hipStreamDestroy(c->stream);
    (void)hipStreamDestroy(c->pstream);
    if (c->sstream) (void)hipStreamDestroy(c->sstream);
    if (c->vstream) (void)hipStreamDestroy(c->vstream);
    delete c;
    return GPMI_OK;
}

// kernel-selection options (gpmi_internal.h: Tuning) by name; -1: no such option, else a status
static int set_tuning_option(Tuning& t, const char* name, int64_t value) {
    if (!strcmp(name, "gemm_small_tiles")) {
        t.gemm_small_tiles = value ? 1 : 0;
    } else if (!strcmp(name, "trsm_wave")) {
        t.trsm_wave = value ? 1 : 0;
    } else if (!strcmp(name, "gemm_small_dma")) {
        t.gemm_small_dma = value ? 1 : 0;
    } else if (!strcmp(name, "gemm_persist")) {
        t.gemm_persist = value ? 1 : 0;
    } else if (!strcmp(name, "gemm_ticket")) {
        if (value < 0 || value > 2) return fail_arg("gemm_ticket must be 0 (off), 1 (trailing updates under lookahead) or 2 (every launch)");
        t.gemm_ticket = (int)value;
    } else if (!strcmp(name, "potrf_server")) {
        if (value < 0 || value > 1023) return fail_arg("potrf_server: 0 off, 1 on, bits 2..512 timing-only ablations");
        t.potrf_server = (int)value;
    } else if (!strcmp(name, "gemm_balance")) {
        t.gemm_balance = value ? 1 : 0;
    } else if (!strcmp(name, "gemm_reserve")) {
        if (value < 0 || value > 24) return fail_arg("gemm_reserve must be in 0..24 (CUs per XCD)");
        t.gemm_reserve = (int)value;
    } else if (!strcmp(name, "trsv_vinv")) {
        if (value < 0 || value > 2) return fail_arg("trsv_vinv must be 0 (16 x 16 rounds), 1 (one launch per block) or 2 (one launch)");
        t.trsv_vinv = (int)value;
    } else if (!strcmp(name, "panel_fused")) {
        t.panel_fused = value ? 1 : 0;
    } else if (!strcmp(name, "rbf_blocks")) {
        if (value < 1 || value > (1 << 24)) return fail_arg("rbf_blocks must be in 1..2^24");
        t.rbf_blocks = (int)value;
    } else if (!strcmp(name, "gemm_dma_waves")) {
        if (value != 4 && value != 8) return fail_arg("gemm_dma_waves must be 4 or 8");
        t.gemm_dma_waves = (int)value;
    } else if (!strcmp(name, "gemm_dma")) {
        t.gemm_use_dma = value ? 1 : 0;
    } else {
        return -1;
    }
    return GPMI_OK;
}

int gpmi_set_option(gpmi_ctx* c, const char* name, int64_t value) {
    if (!c || !name) return fail_arg("gpmi_set_option: null argument");
    if (!strcmp(name, "nb")) {
        if (value != 0 && (value < 128 || value % 128)) return fail_arg("nb must be 0 (auto) or a positive multiple of 128");
        c->nb = value;
    } else if (!strcmp(name, "ld_pad")) {
        if (value < 0 || value % 2) return fail_arg("ld_pad must be even and >= 0");
        c->ld_pad = value;
        c->have_factor = c->have_v = false;
    } else if (!strcmp(name, "timing")) {
        c->timing = value ? 1 : 0;
    } else if (!strcmp(name, "lookahead")) {
        c->lookahead = value ? 1 : 0;
    } else if (!strcmp(name, "la_min")) {
        if (value < 0) return fail_arg("la_min must be >= 0");
        c->la_min = value;
    } else if (!strcmp(name, "shallow_min")) {
        if (value < 0) return fail_arg("shallow_min must be >= 0");
        c->shallow_min = value;
    } else if (!strcmp(name, "one_pass_form")) {
        if (value < 0 || value > 2) return fail_arg("one_pass_form must be 0 (by size), 1 (rows ride) or 2 (rows follow)");
        c->one_pass_form = (int)value;
    } else if (!strcmp(name, "lanes")) {
        if (value < 0 || value > 8) return fail_arg("lanes must be 0 (by size) .. 8");
        c->lanes = (int)value;
    } else if (!strcmp(name, "ramp")) {
        // bit mask: 1 ramp up, 2 half width over the last blocks, 4 quarter width for the last block, bits 4.. = how many
        // blocks count as "last" (0: three)
        if (value < 0 || (value & 8) || (value >> 4) > 64) return fail_arg("ramp: bits 0-2 and a tail count of at most 64 in bits 4..");
        c->ramp = (int)value;
    } else {
        const int rc = set_tuning_option(c->tune, name, value);
        if (rc < 0) return fail_arg("gpmi_set_option: unknown option");
        return rc;
    }
    return GPMI_OK;
}

// The same kernel-selection options for the context-free gpmi_dev_* block primitives called from THIS thread (the
// multi-rank driver): e.g. "gemm_ticket" 2 around its large update launches.  Thread-local; the defaults otherwise.
int gpmi_dev_set_option(const char* name, int64_t value) {
    if (!name) return fail_arg("gpmi_dev_set_option: null argument");
    const int rc = set_tuning_option(t_dev_tuning, name, value);
    if (rc < 0) return fail_arg("gpmi_dev_set_option: unknown option");
    return rc;
}

int gpmi_set_kernel(gpmi_ctx* c, int kind, double p0, double p1) {
    if (!c) return fail_arg("gpmi_set_kernel: null context");
    if (kind < 0 || kind > 2) return fail_arg("gpmi_set_kernel: kind must be 0 (rbf), 1 (linear) or 2 (periodic)");
    if (kind == 2 && (!(p0 != 0.0) || !(p1 != 0.0))) return fail_arg("gpmi_set_kernel: period and lengthscale must be non-zero");
    c->kind = kind; c->kp0 = p0; c->kp1 = p1;
    c->have_factor = c->have_v = false;
    return GPMI_OK;
}

int gpmi_set_kernel_params(gpmi_ctx* c, int kind, const double* params, int nparams) {
    if (!c || !params) return fail_arg("gpmi_set_kernel_params: null argument");
    if (kind != 3) {
        if (kind < 0 || kind > 2 || nparams != 2) return fail_arg("gpmi_set_kernel_params: kinds 0-2 take 2 parameters");
        return gpmi_set_kernel(c, kind, params[0], params[1]);
    }
    if (nparams != 11) return fail_arg("gpmi_set_kernel_params: the CO2 composite kernel takes 11 hyper-parameters (CO2_example.py:86-89)");
    if (!(params[1] != 0.0) || !(params[3] != 0.0) || !(params[4] != 0.0) || !(params[6] != 0.0) || !(params[7] != 0.0) ||
        !(params[9] != 0.0))
        return fail_arg("gpmi_set_kernel_params: theta_2, 4, 5, 7, 8, 10 divide and must be non-zero");
    c->kind = 3;
    for (int i = 0; i < 11; ++i) c->kpv[i] = params[i];
    c->have_factor = c->have_v = false;
    return GPMI_OK;
}

int gpmi_sync(gpmi_ctx* c) {
    if (!c) return fail_arg("gpmi_sync: null context");
    HIP_TRY(hipSetDevice(c->device));
    HIP_TRY(hipStreamSynchronize(c->stream));
    return GPMI_OK;
}

static int cov_impl(gpmi_ctx* c, int kind, const double* a, int64_t N, const double* b, int64_t M, int64_t d,
                    double p0, double p1, const double* kpv, double* out);

int gpmi_cov(gpmi_ctx* c, int kind, const double* a, int64_t N, const double* b, int64_t M, int64_t d,
             double p0, double p1, double* out) {
    if (kind < 0 || kind > 2) return fail_arg("gpmi_cov: kind must be 0, 1 or 2");
    return cov_impl(c, kind, a, N, b, M, d, p0, p1, nullptr, out);
}

int gpmi_cov_params(gpmi_ctx* c, int kind, const double* a, int64_t N, const double* b, int64_t M, int64_t d,
                    const double* params, int nparams, double* out) {
    if (!params) return fail_arg("gpmi_cov_params: null parameters");
    if (kind != 3) {
        if (kind < 0 || kind > 2 || nparams != 2) return fail_arg("gpmi_cov_params: kinds 0-2 take 2 parameters");
        return cov_impl(c, kind, a, N, b, M, d, params[0], params[1], nullptr, out);
    }
    if (nparams != 11) return fail_arg("gpmi_cov_params: the CO2 composite kernel takes 11 hyper-parameters");
    if (!(params[1] != 0.0) || !(params[3] != 0.0) || !(params[4] != 0.0) || !(params[6] != 0.0) || !(params[7] != 0.0) ||
        !(params[9] != 0.0))
        return fail_arg("gpmi_cov_params: theta_2, 4, 5, 7, 8, 10 divide and must be non-zero");
    return cov_impl(c, 3, a, N, b, M, d, 0., 0., params, out);
}

int gpmi_rbf(gpmi_ctx* c, const double* a, int64_t N, const double* b, int64_t M, int64_t d,
             double sigma, double ell, double* out) {
    return gpmi_cov(c, 0, a, N, b, M, d, sigma, ell, out);
}

static int cov_impl(gpmi_ctx* c, int kind, const double* a, int64_t N, const double* b, int64_t M, int64_t d,
                    double p0, double p1, const double* kpv, double* out) {
    const double sigma = p0, ell = p1;
    if (!c || !a || !b || !out) return fail_arg("gpmi_rbf: null argument");
    if (N < 0 || M < 0 || d <= 0) return fail_arg("gpmi_rbf: bad dimensions");
    if (kind == 0 && !(ell != 0.0)) return fail_arg("gpmi_rbf: ell must be non-zero");
    if (kind == 2 && (d != 1 || !(p0 != 0.0) || !(p1 != 0.0)))
        return fail_arg("gpmi_cov: the periodic kernel is 1-D with non-zero period and lengthscale");
    if (N == 0 || M == 0) return GPMI_OK;
    HIP_TRY(hipSetDevice(c->device));
    TuneScope tune_scope(&c->tune);
    hipStream_t s = c->stream;
    DevBuf &da = c->cov_a, &db = c->cov_b, &dout = c->cov_out;   // pooled: grow-only, freed with the context
    const int64_t Mp = round_up(M, TILE), ld = Mp + 32;
    const int64_t chunk = std::max<int64_t>(TILE, std::min<int64_t>(round_up(N, TILE),
                          ((int64_t)1 << 30) / (ld * 8) / TILE * TILE));
    int rc = GPMI_OK;
    hipError_t e;
    double max_sq = -1.0;
    if (kind == 0) {
        Box ba, bb;
        ba.assign(a, N, d);
        bb.assign(b, M, d);
        max_sq = box_max_sq(ba, bb);
    }
    do {
        if ((e = da.ensure((size_t)N * d * 8)) != hipSuccess) { rc = fail_runtime(e, "hipMalloc a"); break; }
        if ((e = db.ensure((size_t)M * d * 8)) != hipSuccess) { rc = fail_runtime(e, "hipMalloc b"); break; }
        if ((e = dout.ensure((size_t)chunk * ld * 8)) != hipSuccess) { rc = fail_runtime(e, "hipMalloc out"); break; }
        if ((e = hipMemcpyAsync(da.p, a, (size_t)N * d * 8, hipMemcpyHostToDevice, s)) != hipSuccess ||
            (e = hipMemcpyAsync(db.p, b, (size_t)M * d * 8, hipMemcpyHostToDevice, s)) != hipSuccess) {
            rc = fail_runtime(e, "hipMemcpy H2D"); break;
        }
        for (int64_t r0 = 0; r0 < N && rc == GPMI_OK; r0 += chunk) {
            const int64_t rows = std::min(chunk, N - r0);
            RbfArgs r;
            r.A = da.as<double>(); r.B = db.as<double>();
            r.nA = N; r.nB = M; r.d = d; r.row0 = r0; r.nrows = round_up(rows, TILE); r.ncols = Mp;
            r.coef = (kind == 0) ? -.5 * (1 / (ell * ell)) : 0.; r.sig2 = sigma * sigma; r.diag_add = 0.; r.symmetric = 0;
            r.kind = kind; r.kp0 = p0; r.kp1 = p1;
            if (kpv) for (int i = 0; i < 11; ++i) r.kpv[i] = kpv[i];
            r.delta_square = (N == M) ? 1 : 0;
            r.max_sq = max_sq;
            r.out = dout.as<double>(); r.ld = ld;
            if ((e = launch_rbf(s, r)) != hipSuccess) { rc = fail_runtime(e, "rbf kernel"); break; }
            if ((e = hipMemcpy2DAsync(out + r0 * M, (size_t)M * 8, dout.p, (size_t)ld * 8, (size_t)M * 8,
                                      (size_t)rows, hipMemcpyDeviceToHost, s)) != hipSuccess ||
                (e = hipStreamSynchronize(s)) != hipSuccess) {
                rc = fail_runtime(e, "rbf D2H"); break;
            }
        }
    } while (0);
    (void)hipStreamSynchronize(s);
    return rc;
}

int gpmi_set_train(gpmi_ctx* c, const double* X, int64_t N, int64_t d, const double* y) {
    if (!c || !X || !y) return fail_arg("gpmi_set_train: null argument");
    if (N <= 0 || d <= 0) return fail_arg("gpmi_set_train: N and d must be positive");
    HIP_TRY(hipSetDevice(c->device));
    c->have_train = c->have_factor = c->have_v = c->have_test = false;
    HIP_TRY(c->X.ensure((size_t)N * d * 8));
    HIP_TRY(c->y.ensure((size_t)N * 8));
    HIP_TRY(hipMemcpyAsync(c->X.p, X, (size_t)N * d * 8, hipMemcpyHostToDevice, c->stream));
    HIP_TRY(hipMemcpyAsync(c->y.p, y, (size_t)N * 8, hipMemcpyHostToDevice, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    c->N = N; c->d = d;
    c->boxX.assign(X, N, d);
    c->have_train = true;
    return GPMI_OK;
}

int gpmi_factorize(gpmi_ctx* c, double sigma, double ell, double noise_var, double* lml,
                   int64_t* bad_pivot) {
    if (!c) return fail_arg("gpmi_factorize: null context");
    HIP_TRY(hipSetDevice(c->device));
    TuneScope tune_scope(&c->tune);
    return factorize_impl(c, sigma, ell, noise_var, lml, bad_pivot);
}

int gpmi_fit(gpmi_ctx* c, const double* X, int64_t N, int64_t d, const double* y, double sigma,
             double ell, double noise_var, double* lml, int64_t* bad_pivot) {
    int rc = gpmi_set_train(c, X, N, d, y);
    if (rc) return rc;
    return gpmi_factorize(c, sigma, ell, noise_var, lml, bad_pivot);
}

int gpmi_get_m(gpmi_ctx* c, double* m_out) {
    if (!c || !m_out) return fail_arg("gpmi_get_m: null argument");
    if (!c->have_factor) return fail_arg("gpmi_get_m: no factorisation resident");
    HIP_TRY(hipSetDevice(c->device));
    HIP_TRY(hipMemcpyAsync(m_out, c->m_row(), (size_t)c->N * 8,
                           hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    return GPMI_OK;
}

int gpmi_get_diag(gpmi_ctx* c, double* diag_out) {
    if (!c || !diag_out) return fail_arg("gpmi_get_diag: null argument");
    if (!c->have_factor) return fail_arg("gpmi_get_diag: no factorisation resident");
    HIP_TRY(hipSetDevice(c->device));
    HIP_TRY(hipMemcpy2DAsync(diag_out, 8, c->A.p, (size_t)(c->ldA + 1) * 8, 8, (size_t)c->N,
                             hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    return GPMI_OK;
}

int gpmi_get_factor_block(gpmi_ctx* c, int64_t r0, int64_t r1, int64_t c0, int64_t c1, double* out) {
    if (!c || !out) return fail_arg("gpmi_get_factor_block: null argument");
    if (!c->have_factor) return fail_arg("gpmi_get_factor_block: no factorisation resident");
    if (r0 < 0 || c0 < 0 || r1 > c->N || c1 > c->N || r0 > r1 || c0 > c1)
        return fail_arg("gpmi_get_factor_block: block out of range");
    if (r0 == r1 || c0 == c1) return GPMI_OK;
    HIP_TRY(hipSetDevice(c->device));
    const size_t bytes = (size_t)(r1 - r0) * (c1 - c0) * 8;
    HIP_TRY(c->dense.ensure(bytes));
    HIP_TRY(launch_extract(c->stream, c->A.as<double>(), c->ldA, r0, r1, c0, c1, c->dense.as<double>(), 1));
    HIP_TRY(hipMemcpyAsync(out, c->dense.p, bytes, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    return GPMI_OK;
}

// L^T x = b on the resident fused factor (a5): the first call after a factorisation inverts the 128 x 128 diagonal
// blocks into their upper triangles (one launch, all blocks at once), every call then runs one product per block
static hipError_t backward_solve_fused(gpmi_ctx* c, double* b, double* xout) {
    double* A = c->A.as<double>();
    const int mode = tuning().trsv_vinv;
    if (!mode) return launch_trsv_lt_fused(c->stream, A, c->ldA, b, xout, c->Np);
    hipError_t e;
    if (mode >= 2 && (e = c->vside.ensure((size_t)c->Np * 128 * 8)) != hipSuccess) return e;
    if (!c->have_vinv || (mode >= 2 && !c->have_vside)) {
        e = launch_vinv128(c->stream, A, c->ldA, c->Np, mode >= 2 ? c->vside.as<double>() : nullptr);
        if (e != hipSuccess) return e;
        c->have_vinv = true;
        c->have_vside = mode >= 2;
    }
    if (mode >= 2) {
        if ((e = c->flag.ensure(64)) != hipSuccess) return e;
        if ((e = hipMemsetAsync(c->flag.p, 0, 64, c->stream)) != hipSuccess) return e;
        return launch_trsv_lt_chain(c->stream, A, c->ldA, c->vside.as<double>(), b, xout, c->Np, c->flag.as<int>());
    }
    return launch_trsv_lt_vinv(c->stream, A, c->ldA, b, xout, c->Np);
}

int gpmi_get_alpha(gpmi_ctx* c, double* alpha_out) {
    if (!c || !alpha_out) return fail_arg("gpmi_get_alpha: null argument");
    if (!c->have_factor) return fail_arg("gpmi_get_alpha: no factorisation resident");
    HIP_TRY(hipSetDevice(c->device));
    TuneScope tune_scope(&c->tune);
    hipStream_t s = c->stream;
    c->timers_reset({GPMI_T_ALPHA});
    HIP_TRY(c->vec.ensure((size_t)std::max(c->Np, c->np_) * 4 * 8));
    double* x = c->vec.as<double>();
    // padded tail of m is zero (identity padding), so the padded system stays consistent
    HIP_TRY(hipMemcpyAsync(x, c->m_row(), (size_t)c->Np * 8,
                           hipMemcpyDeviceToDevice, s));
    size_t sp = c->span_begin(GPMI_T_ALPHA);
    if (c->factor_fused) {
        HIP_TRY(backward_solve_fused(c, x, x + c->Np));
        x += c->Np;
    } else {
        HIP_TRY(launch_trsv_lt(s, c->A.as<double>(), c->ldA, x, c->Np));
    }
    c->span_end(sp);
    HIP_TRY(hipMemcpyAsync(alpha_out, x, (size_t)c->N * 8, hipMemcpyDeviceToHost, s));
    int gave_up = 0;
    if (c->factor_fused && tuning().trsv_vinv >= 2)
        HIP_TRY(hipMemcpyAsync(&gave_up, c->flag.p, sizeof(int), hipMemcpyDeviceToHost, s));
    HIP_TRY(hipStreamSynchronize(s));
    c->timers_collect();
    if (gave_up) return fail_runtime(hipErrorUnknown, "gpmi_get_alpha: the single-launch backward solve gave up waiting for a block");
    return GPMI_OK;
}

int gpmi_set_test(gpmi_ctx* c, const double* Xs, int64_t n) {
    if (!c || !Xs) return fail_arg("gpmi_set_test: null argument");
    if (!c->have_train) return fail_arg("gpmi_set_test: set the training set first");
    if (n <= 0) return fail_arg("gpmi_set_test: n must be positive");
    HIP_TRY(hipSetDevice(c->device));
    c->have_test = c->have_v = false;
    HIP_TRY(c->Xs.ensure((size_t)n * c->d * 8));
    HIP_TRY(hipMemcpyAsync(c->Xs.p, Xs, (size_t)n * c->d * 8, hipMemcpyHostToDevice, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    c->hXs.assign(Xs, Xs + (size_t)n * c->d);
    c->boxXs.assign(Xs, n, c->d);
    c->n = n;
    c->np_ = round_up(n, TILE);
    c->have_test = true;
    return GPMI_OK;
}

int gpmi_predict_resident(gpmi_ctx* c, double* mu, double* out2, int want_sd) {
    if (!c) return fail_arg("gpmi_predict: null context");
    if (!c->have_factor) return fail_arg("gpmi_predict: no factorisation resident (call gpmi_factorize)");
    if (!c->have_test) return fail_arg("gpmi_predict: no test set (call gpmi_set_test)");
    HIP_TRY(hipSetDevice(c->device));
    Tuning tn = c->tune;
    tn.panel_fused = c->factor_fused;      // solve with the kind of leaves that produced the resident factor
    TuneScope tune_scope(&tn);
    hipStream_t s = c->stream;
    c->timers_reset({GPMI_T_KS, GPMI_T_SOLVE_V, GPMI_T_MEANVAR});
    c->have_v = false;
    c->v_in_A = false;
    c->ldV = c->Np + c->ld_pad;
    HIP_TRY(c->V.ensure((size_t)c->np_ * c->ldV * 8));
    HIP_TRY(c->vec.ensure((size_t)std::max(c->Np, c->np_) * 4 * 8));
    double* V = c->V.as<double>();

    size_t sp = c->span_begin(GPMI_T_KS);
    RbfArgs r;
    r.A = c->Xs.as<double>(); r.B = c->X.as<double>();
    r.nA = c->n; r.nB = c->N; r.d = c->d; r.row0 = 0; r.nrows = c->np_; r.ncols = c->Np;
    set_kernel_args(c, r);
    r.diag_add = 0.; r.symmetric = 0;
    r.delta_square = (c->n == c->N) ? 1 : 0;   // kernel_4's delta is eye whenever the matrix is square (CO2_example.py:58)
    r.max_sq = box_max_sq(c->boxXs, c->boxX);
    r.out = V; r.ld = c->ldV;
    HIP_TRY(launch_rbf(s, r));
    c->span_end(sp);

    sp = c->span_begin(GPMI_T_SOLVE_V);
    HIP_TRY(solve_sweep(c, V, c->ldV, c->np_));
    c->span_end(sp);

    sp = c->span_begin(GPMI_T_MEANVAR);
    double* dot = c->vec.as<double>();
    double* sq = dot + c->np_;
    HIP_TRY(launch_row_dots(s, V, c->ldV, c->np_, c->Np, c->m_row(), dot, sq));
    c->span_end(sp);

    std::vector<double> h(2 * (size_t)c->np_);
    HIP_TRY(hipMemcpyAsync(h.data(), dot, h.size() * 8, hipMemcpyDeviceToHost, s));
    HIP_TRY(hipStreamSynchronize(s));
    c->timers_collect();
    c->have_v = true;
    ++c->v_gen;
    meanvar_to_host(c, h, mu, out2, want_sd);
    return GPMI_OK;
}

// prediction() in one pass (a11 = a1 .. a8 for a training AND a test set known up front, GP_regression.py:109-156): the
// rows K(X*, X) are appended below the y row, so the panel solves and trailing updates of the Cholesky turn them into
// v^T = K_s^T L^-T on the way (a7 costs no launches of its own and fills the chip while the last, short updates leave
// it idle), and mean / variance are read off behind the LML.  Same results as gpmi_factorize + gpmi_predict_resident to
// rounding (the block widths of the two sweeps can differ).
int gpmi_fit_predict_resident(gpmi_ctx* c, double sigma, double ell, double noise_var, double* lml, int64_t* bad_pivot,
                              double* mu, double* out2, int want_sd) {
    if (!c) return fail_arg("gpmi_fit_predict: null context");
    if (!c->have_test) return fail_arg("gpmi_fit_predict: no test set (call gpmi_set_test)");
    HIP_TRY(hipSetDevice(c->device));
    TuneScope tune_scope(&c->tune);
    return factorize_impl(c, sigma, ell, noise_var, lml, bad_pivot, true, mu, out2, want_sd);
}

// prediction() WITH its posterior-sample factor in one pass (a11 + f1, GP_regression.py:109-156): one Cholesky of the augmented
//     [[K + sI, .], [K(X*, X), K_ss + jitter I]]        (N + n columns; the y rows ride below)
// -- its first N columns are L, the test rows' first N columns v^T, and its last n columns cholesky(K_ss + jitter I - v^T v):
// the Schur complement the trailing updates leave there IS the posterior covariance.  Same flops as the three separate steps
// ((N + n)^3 / 3), no SYRK launch and no second factorisation.  L_out (n x n row-major, zeros above the diagonal) may be
// null: gpmi_post_chol with the same jitter then only downloads it.
int gpmi_fit_predict_sample_resident(gpmi_ctx* c, double sigma, double ell, double noise_var, double jitter, double* lml,
                                     int64_t* bad_pivot, double* mu, double* out2, int want_sd, double* L_out) {
    if (!c) return fail_arg("gpmi_fit_predict_sample: null context");
    if (!c->have_test) return fail_arg("gpmi_fit_predict_sample: no test set (call gpmi_set_test)");
    HIP_TRY(hipSetDevice(c->device));
    TuneScope tune_scope(&c->tune);
    int rc = factorize_impl(c, sigma, ell, noise_var, lml, bad_pivot, true, mu, out2, want_sd, true, jitter);
    if (rc || !L_out) return rc;
    return gpmi_post_chol(c, jitter, L_out, bad_pivot);
}

int gpmi_predict(gpmi_ctx* c, const double* Xs, int64_t n, double* mu, double* out2, int want_sd) {
    int rc = gpmi_set_test(c, Xs, n);
    if (rc) return rc;
    return gpmi_predict_resident(c, mu, out2, want_sd);
}

// f2 -- gradient of the log marginal likelihood at the resident factorisation:
// 0.5 * tr((alpha alpha^T - K_y^-1) dK/dtheta)  (tune_hyperparms_regression.py:43-57; the reference
// builds K_y^-1 = inv(L.T) inv(L) at :144 and two N x N products).  Here: U = L^-T by the TRSM
// sweep on the identity (N^3/3), -K_y^-1 = -U U^T by one MFMA GEMM per row block over the
// non-zero column range (N^3/3), then one fused pass for the trace (grad.hip).
int gpmi_lml_grad(gpmi_ctx* c, double* d_ell, double* d_sigma) {
    if (!c || !d_ell || !d_sigma) return fail_arg("gpmi_lml_grad: null argument");
    if (!c->have_factor) return fail_arg("gpmi_lml_grad: no factorisation resident (call gpmi_factorize)");
    if (c->kind != 0) return fail_arg("gpmi_lml_grad: squared-exponential kernel only (tune_hyperparms_regression.py:54)");
    HIP_TRY(hipSetDevice(c->device));
    Tuning tn = c->tune;
    tn.panel_fused = c->factor_fused;
    TuneScope tune_scope(&tn);
    hipStream_t s = c->stream;
    const int64_t Np = c->Np, ld = c->ldA;
    c->timers_reset({GPMI_T_GRAD});
    HIP_TRY(c->U.ensure((size_t)Np * ld * 8));
    HIP_TRY(c->Kn.ensure((size_t)Np * ld * 8));
    HIP_TRY(c->vec.ensure((size_t)std::max(c->Np, c->np_) * 4 * 8));
    size_t sp = c->span_begin(GPMI_T_GRAD);
    // alpha = L^-T m (a5)
    double* alpha = c->vec.as<double>();
    HIP_TRY(hipMemcpyAsync(alpha, c->m_row(), (size_t)Np * 8, hipMemcpyDeviceToDevice, s));
    if (c->factor_fused) {
        HIP_TRY(backward_solve_fused(c, alpha, alpha + Np));
        alpha += Np;
    } else {
        HIP_TRY(launch_trsv_lt(s, c->A.as<double>(), ld, alpha, Np));
    }
    // U = I * L^-T
    double* U = c->U.as<double>();
    HIP_TRY(launch_fill_rows(s, U, ld, Np, Np, 0.0));
    HIP_TRY(launch_set_identity_diag(s, U, ld, Np));
    HIP_TRY(solve_sweep(c, U, ld, Np, true));
    // Kn = -U U^T, lower tiles: row block i needs columns >= its first row only
    double* Kn = c->Kn.as<double>();
    HIP_TRY(launch_fill_rows(s, Kn, ld, Np, Np, 0.0));
    const int64_t NB = c->block(Np);
    for (int64_t r0 = 0; r0 < Np; r0 += NB) {
        const int64_t nb = std::min<int64_t>(NB, Np - r0);
        GemmArgs g;
        g.C = Kn + r0 * ld; g.A = U + r0 * ld + r0; g.B = U + r0;
        g.ldc = g.lda = g.ldb = ld;
        g.M = nb; g.N = r0 + nb; g.K = Np - r0;
        g.mode = 0; g.lower = 1; g.diag_off = r0;
        HIP_TRY(launch_gemm_nt(s, g));
    }
    GradArgs a;
    a.A = a.B = c->X.as<double>(); a.nA = a.nB = c->N; a.d = c->d;
    a.row0 = 0; a.nrows = c->N;
    a.alpha_r = a.alpha_c = alpha;
    a.Kinv = Kn; a.ld = ld; a.kinv_sign = -1.0;
    a.coef = c->coef; a.sig2 = c->sig2; a.two_sigma = 2 * c->sigma;
    a.inv_l3 = 1.0 / (c->ell * c->ell * c->ell);
    a.tri = 1;
    const int64_t nblk = grad_trace_blocks(a);
    HIP_TRY(c->gpart.ensure((size_t)nblk * 16));
    a.partial = c->gpart.as<double>();
    HIP_TRY(launch_grad_trace(s, a));
    c->span_end(sp);
    std::vector<double> part((size_t)nblk * 2);
    HIP_TRY(hipMemcpyAsync(part.data(), a.partial, part.size() * 8, hipMemcpyDeviceToHost, s));
    int gave_up = 0;      // the one-launch backward solve's "a poll gave up" word, as gpmi_get_alpha reads it
    if (c->factor_fused && tuning().trsv_vinv >= 2)
        HIP_TRY(hipMemcpyAsync(&gave_up, c->flag.p, sizeof(int), hipMemcpyDeviceToHost, s));
    HIP_TRY(hipStreamSynchronize(s));
    c->timers_collect();
    if (gave_up) return fail_runtime(hipErrorUnknown, "gpmi_lml_grad: the single-launch backward solve gave up waiting for a block");
    double sl = 0.0, ss = 0.0;
    for (int64_t b = 0; b < nblk; ++b) { sl += part[2 * b]; ss += part[2 * b + 1]; }   // fixed order
    *d_ell = .5 * sl;
    *d_sigma = .5 * ss;
    return GPMI_OK;
}

// The same trace from caller-supplied alpha and K_y^-1 (host, N x N row-major): the arguments the
// reference's gradient_ascent(a, b, sigma, l, alpha, K_y) receives (tune_hyperparms_regression.py:31).
int gpmi_grad_trace(gpmi_ctx* c, const double* a_in, const double* b_in, int64_t N, int64_t d, double sigma,
                    double ell, const double* alpha_in, const double* Kinv_in, double* d_ell, double* d_sigma) {
    if (!c || !a_in || !b_in || !alpha_in || !Kinv_in || !d_ell || !d_sigma) return fail_arg("gpmi_grad_trace: null argument");
    if (N <= 0 || d <= 0) return fail_arg("gpmi_grad_trace: N and d must be positive");
    if (!(ell != 0.0)) return fail_arg("gpmi_grad_trace: ell must be non-zero");
    HIP_TRY(hipSetDevice(c->device));
    hipStream_t s = c->stream;
    DevBuf da, db, dal, dk, dp;
    const int64_t chunk = std::max<int64_t>(TILE, std::min<int64_t>(round_up(N, TILE), ((int64_t)1 << 30) / (N * 8) / TILE * TILE));
    int rc = GPMI_OK;
    hipError_t e = hipSuccess;
    double sl = 0.0, ss = 0.0;
    do {
        if ((e = da.ensure((size_t)N * d * 8)) != hipSuccess || (e = db.ensure((size_t)N * d * 8)) != hipSuccess ||
            (e = dal.ensure((size_t)N * 8)) != hipSuccess || (e = dk.ensure((size_t)chunk * N * 8)) != hipSuccess) {
            rc = fail_runtime(e, "hipMalloc"); break;
        }
        if ((e = hipMemcpyAsync(da.p, a_in, (size_t)N * d * 8, hipMemcpyHostToDevice, s)) != hipSuccess ||
            (e = hipMemcpyAsync(db.p, b_in, (size_t)N * d * 8, hipMemcpyHostToDevice, s)) != hipSuccess ||
            (e = hipMemcpyAsync(dal.p, alpha_in, (size_t)N * 8, hipMemcpyHostToDevice, s)) != hipSuccess) {
            rc = fail_runtime(e, "hipMemcpy H2D"); break;
        }
        for (int64_t r0 = 0; r0 < N && rc == GPMI_OK; r0 += chunk) {
            const int64_t rows = std::min(chunk, N - r0);
            if ((e = hipMemcpyAsync(dk.p, Kinv_in + r0 * N, (size_t)rows * N * 8, hipMemcpyHostToDevice, s)) != hipSuccess) {
                rc = fail_runtime(e, "hipMemcpy H2D"); break;
            }
            GradArgs g;
            g.A = da.as<double>(); g.B = db.as<double>(); g.nA = g.nB = N; g.d = d;
            g.row0 = r0; g.nrows = rows;
            g.alpha_r = g.alpha_c = dal.as<double>();
            g.Kinv = dk.as<double>(); g.ld = N; g.kinv_sign = 1.0;
            g.coef = -.5 * (1 / (ell * ell)); g.sig2 = sigma * sigma; g.two_sigma = 2 * sigma;
            g.inv_l3 = 1.0 / (ell * ell * ell);
            g.tri = 0;
            const int64_t nblk = grad_trace_blocks(g);
            if ((e = dp.ensure((size_t)nblk * 16)) != hipSuccess) { rc = fail_runtime(e, "hipMalloc"); break; }
            g.partial = dp.as<double>();
            std::vector<double> part((size_t)nblk * 2);
            if ((e = launch_grad_trace(s, g)) != hipSuccess ||
                (e = hipMemcpyAsync(part.data(), g.partial, part.size() * 8, hipMemcpyDeviceToHost, s)) != hipSuccess ||
                (e = hipStreamSynchronize(s)) != hipSuccess) {
                rc = fail_runtime(e, "gradient trace"); break;
            }
            for (int64_t b = 0; b < nblk; ++b) { sl += part[2 * b]; ss += part[2 * b + 1]; }
        }
    } while (0);
    (void)hipStreamSynchronize(s);
    da.release(); db.release(); dal.release(); dk.release(); dp.release();
    if (rc == GPMI_OK) { *d_ell = .5 * sl; *d_sigma = .5 * ss; }
    return rc;
}

// the posterior-sample factor on the device: where cholesky(K_ss + jitter I - v^T v) of the resident test set sits (factor, ld)
// -- behind L when it rode through the augmented factorisation (gpmi_fit_predict_sample_resident), else formed now in P
// (K_ss build, v^T v by one MFMA SYRK, the same Cholesky) unless P already holds it for this jitter
static int post_factor_device(gpmi_ctx* c, double jitter, const double** factor, int64_t* ld, int64_t* bad_pivot) {
    if (!c->have_v) return fail_arg("gpmi_post_chol: run gpmi_predict first");
    hipStream_t s = c->stream;
    const int64_t np_ = c->np_, n = c->n;
    if (bad_pivot) *bad_pivot = 0;
    if (c->post_in_A && c->v_in_A && jitter == c->post_jitter) {
        *factor = c->A.as<double>() + c->Np * c->ldA + c->Np;
        *ld = c->ldA;
        return GPMI_OK;
    }
    if (c->post_in_P && jitter == c->post_jitter_P && c->post_gen_P == c->v_gen) {
        *factor = c->P.as<double>();
        *ld = c->ldP;
        return GPMI_OK;
    }
    c->post_in_P = false;
    c->timers_reset({GPMI_T_POSTCHOL});
    c->ldP = np_ + 32;
    HIP_TRY(c->P.ensure((size_t)np_ * c->ldP * 8));
    double* P = c->P.as<double>();
    const int64_t big = std::numeric_limits<int64_t>::max();
    HIP_TRY(hipMemcpyAsync(c->info.p, &big, sizeof big, hipMemcpyHostToDevice, s));
    size_t sp = c->span_begin(GPMI_T_POSTCHOL);
    RbfArgs r;   // K_ss + jitter*I, lower tiles (GP_regression.py:128,154)
    r.A = r.B = c->Xs.as<double>();
    r.nA = r.nB = n; r.d = c->d; r.row0 = 0; r.nrows = np_; r.ncols = np_;
    set_kernel_args(c, r);
    r.diag_add = jitter; r.symmetric = 1; r.delta_square = 1;
    r.max_sq = box_max_sq(c->boxXs, c->boxXs);
    r.out = P; r.ld = c->ldP;
    HIP_TRY(launch_rbf(s, r));
    GemmArgs g;  // P -= v^T v  (rows of V are the columns of v)
    g.C = P; g.A = g.B = c->v_rows();
    g.ldc = c->ldP; g.lda = g.ldb = c->ldV;
    g.M = g.N = np_; g.K = c->Np; g.mode = 0; g.lower = 1; g.diag_off = 0;
    HIP_TRY(launch_gemm_nt(s, g));
    HIP_TRY(cholesky_inplace(c, P, c->ldP, np_, np_, c->info.as<int64_t>(), false));
    c->span_end(sp);
    int64_t info;
    HIP_TRY(hipMemcpyAsync(&info, c->info.p, sizeof info, hipMemcpyDeviceToHost, s));
    HIP_TRY(hipStreamSynchronize(s));
    c->timers_collect();
    if (info != big && info < n) {
        if (bad_pivot) *bad_pivot = info + 1;
        g_err = "Matrix is not positive definite";
        return GPMI_ERR_NOT_PD;
    }
    c->post_in_P = true;
    c->post_jitter_P = jitter;
    c->post_gen_P = c->v_gen;
    *factor = P;
    *ld = c->ldP;
    return GPMI_OK;
}

int gpmi_post_chol(gpmi_ctx* c, double jitter, double* L_out, int64_t* bad_pivot) {
    if (!c || !L_out) return fail_arg("gpmi_post_chol: null argument");
    HIP_TRY(hipSetDevice(c->device));
    TuneScope tune_scope(&c->tune);
    const double* F = nullptr;
    int64_t ld = 0;
    int rc = post_factor_device(c, jitter, &F, &ld, bad_pivot);
    if (rc) return rc;
    hipStream_t s = c->stream;
    const int64_t n = c->n;
    HIP_TRY(c->dense.ensure((size_t)n * n * 8));
    HIP_TRY(launch_extract(s, F, ld, 0, n, 0, n, c->dense.as<double>(), 1));
    HIP_TRY(hipMemcpyAsync(L_out, c->dense.p, (size_t)n * n * 8, hipMemcpyDeviceToHost, s));
    HIP_TRY(hipStreamSynchronize(s));
    return GPMI_OK;
}

// L_ @ Z for the posterior samples f_post = mu + L_ @ normals (GP_regression.py:155) without bringing L_ to the host: Z
// (n x num_fun row-major, the caller's normals -- drawn on the host so that np.random's order is the reference's) goes up,
// the product comes down.  The factor is the one gpmi_post_chol(jitter) would return.
int gpmi_post_sample(gpmi_ctx* c, double jitter, const double* Z, int64_t num_fun, double* LZ_out, int64_t* bad_pivot) {
    if (!c || !Z || !LZ_out) return fail_arg("gpmi_post_sample: null argument");
    if (num_fun <= 0 || num_fun > (1 << 20)) return fail_arg("gpmi_post_sample: num_fun must be in 1 .. 2^20");
    HIP_TRY(hipSetDevice(c->device));
    TuneScope tune_scope(&c->tune);
    const double* F = nullptr;
    int64_t ld = 0;
    int rc = post_factor_device(c, jitter, &F, &ld, bad_pivot);
    if (rc) return rc;
    hipStream_t s = c->stream;
    const int64_t n = c->n;
    const size_t bytes = (size_t)n * (size_t)num_fun * 8;
    HIP_TRY(c->dense.ensure(2 * bytes));
    double* Zd = c->dense.as<double>();
    double* Od = Zd + (size_t)n * (size_t)num_fun;
    HIP_TRY(hipMemcpyAsync(Zd, Z, bytes, hipMemcpyHostToDevice, s));
    HIP_TRY(launch_tri_mul(s, F, ld, Zd, n, num_fun, Od));
    HIP_TRY(hipMemcpyAsync(LZ_out, Od, bytes, hipMemcpyDeviceToHost, s));
    HIP_TRY(hipStreamSynchronize(s));
    return GPMI_OK;
}

// Lane l of a batch: its own context (streams, A, workspaces) on the same device, with the parent's
// training set copied device to device.  Small and mid-size factorisations leave most of the chip idle
// in their latency-bound panel steps; two in flight fill it (measured per triple, 1 -> 2 lanes: N = 512
// 0.45 -> 0.24 ms, N = 2048 1.74 -> 0.93 ms, N = 8192 10.5 -> 6.9 ms, N = 16384 39.5 -> 32.1 ms,
// N = 32768 213 -> 201 ms).
static int lane_prepare(gpmi_ctx* c, gpmi_ctx* l) {
    l->nb = c->nb; l->ld_pad = c->ld_pad; l->lookahead = c->lookahead; l->la_min = c->la_min; l->ramp = c->ramp; l->timing = c->timing;
    l->tune = c->tune;
    l->kind = c->kind; l->kp0 = c->kp0; l->kp1 = c->kp1;
    for (int i = 0; i < 11; ++i) l->kpv[i] = c->kpv[i];
    l->N = c->N; l->d = c->d; l->boxX = c->boxX;
    HIP_TRY(l->X.ensure((size_t)c->N * c->d * 8));
    HIP_TRY(l->y.ensure((size_t)c->N * 8));
    HIP_TRY(hipMemcpyAsync(l->X.p, c->X.p, (size_t)c->N * c->d * 8, hipMemcpyDeviceToDevice, l->stream));
    HIP_TRY(hipMemcpyAsync(l->y.p, c->y.p, (size_t)c->N * 8, hipMemcpyDeviceToDevice, l->stream));
    HIP_TRY(hipStreamSynchronize(l->stream));
    l->have_train = true;
    l->have_factor = l->have_v = l->have_test = false;
    return GPMI_OK;
}

int gpmi_lml_batch(gpmi_ctx* c, const double* triples, int64_t T, double* lml_out, int* status_out) {
    if (!c || !triples || !lml_out) return fail_arg("gpmi_lml_batch: null argument");
    if (T < 0) return fail_arg("gpmi_lml_batch: T < 0");
    if (!c->have_train) return fail_arg("gpmi_lml_batch: no training set (call gpmi_set_train)");
    HIP_TRY(hipSetDevice(c->device));
    const int64_t Np = round_up(c->N, TILE);
    // lanes by size, measured with 24 triples per call (round 4, profiles/r04_lml_batch_lanes_small.txt; ms per triple with
    // 2 / 3 / 4 / 5 / 6 lanes): N = 512 0.159 / 0.112 / 0.092 / 0.161 / 0.163, 2048 0.572 / 0.408 / 0.322 / 0.468 / 0.505,
    // 8192 5.21 / 4.46 / 4.13 / 4.85 / 4.75 -- four single-stream lanes below the lookahead threshold (round 2's "two, more
    // do not help" was measured with four hardware queues; _lib.py now asks for eight); 12288 12.6 / 12.4 / 12.1 / 11.8 /
    // 11.8 and 16384 26.5 / 25.3 / 26.2 / 24.8 / 24.8 -- five there; N = 32768 (profiles/r04_cfg5_lanes.txt) 0.1934 / 0.1865 /
    // 0.1849 / 0.1855 s with 1 / 2 / 3 / 4 -- three
    int L = c->lanes ? c->lanes : (Np > 32768 ? 1 : Np >= 24576 ? 3 : Np >= 12288 ? 5 : 4);
    L = (int)std::min<int64_t>(L, std::max<int64_t>(T, 1));
    // lanes beside each other fill the chip themselves: below 12288 columns every lane runs on ONE stream (a lane's
    // lookahead would add a second stream per lane and loses: N = 8192, 3 lanes 4.46 ms per triple without, 5.45 with;
    // profiles/r04_la_min_batch.txt), whatever the threshold of this context's single fits
    struct LaMinScope {
        gpmi_ctx* c; int64_t keep;
        ~LaMinScope() { c->la_min = keep; }
    } la_scope{c, c->la_min};
    if (L > 1) c->la_min = std::max<int64_t>(c->la_min, 12288);
    while ((int)c->lane_ctx.size() < L - 1) {
        gpmi_ctx* l = nullptr;
        int rc = gpmi_ctx_create(c->device, &l);
        if (rc) return rc;
        c->lane_ctx.push_back(l);
    }
    for (int j = 0; j < L - 1; ++j) {
        int rc = lane_prepare(c, c->lane_ctx[(size_t)j]);
        if (rc) return rc;
    }
    // residue class r of the triple index runs on lane r; the parent context takes the class of the last
    // triple, so the factor left resident afterwards is that of triples[T-1], as with one lane
    const int parent_class = (int)((T - 1 + L) % L);
    std::vector<gpmi_ctx*> lane((size_t)L);
    for (int r = 0, nx = 0; r < L; ++r) lane[(size_t)r] = (r == parent_class) ? c : c->lane_ctx[(size_t)nx++];
    std::vector<int> lane_rc((size_t)L, GPMI_OK);
    std::vector<std::string> lane_err((size_t)L);
    std::vector<std::vector<double>> lane_ms((size_t)L, std::vector<double>(GPMI_T_COUNT, 0.0));
    auto work = [&](int r) {
        gpmi_ctx* l = lane[(size_t)r];
        TuneScope tune_scope(&l->tune);         // this lane's thread runs with this lane's options
        GemmShallowScope shares_chip(false, L > 1);   // another lane's kernels run beside this one's: no launch takes the whole chip
        if (hipSetDevice(l->device) != hipSuccess) { lane_rc[(size_t)r] = GPMI_ERR_RUNTIME; lane_err[(size_t)r] = "hipSetDevice"; return; }
        for (int64_t t = r; t < T; t += L) {
            const double ell = triples[3 * t], sigma = triples[3 * t + 1], s2 = triples[3 * t + 2];
            double lml = 0.;
            int64_t bad = 0;
            const int rc = factorize_impl(l, sigma, ell, s2, &lml, &bad);
            if (rc == GPMI_ERR_RUNTIME || rc == GPMI_ERR_BAD_ARG) {
                lane_rc[(size_t)r] = rc;
                lane_err[(size_t)r] = g_err;       // this thread's message
                return;
            }
            lml_out[t] = lml;
            if (status_out) status_out[t] = rc;
            for (int i = 0; i < GPMI_T_COUNT; ++i) lane_ms[(size_t)r][(size_t)i] += l->stage_ms[i];
        }
    };
    if (L == 1) {
        work(0);
    } else {
        std::vector<std::thread> th;
        for (int r = 0; r < L; ++r) th.emplace_back(work, r);
        for (auto& t : th) t.join();
    }
    for (int r = 0; r < L; ++r)
        if (lane_rc[(size_t)r] != GPMI_OK) {
            g_err = lane_err[(size_t)r];
            return lane_rc[(size_t)r];
        }
    for (int i = 0; i < GPMI_T_COUNT; ++i) {
        c->stage_ms[i] = 0.0;
        for (int r = 0; r < L; ++r) c->stage_ms[i] += lane_ms[(size_t)r][(size_t)i];
    }
    return GPMI_OK;
}

int gpmi_get_timers(gpmi_ctx* c, double* stage_ms, int count) {
    if (!c || !stage_ms) return fail_arg("gpmi_get_timers: null argument");
    for (int i = 0; i < count; ++i) stage_ms[i] = i < GPMI_T_COUNT ? c->stage_ms[i] : 0.;
    return GPMI_OK;
}

}  // extern "C"
