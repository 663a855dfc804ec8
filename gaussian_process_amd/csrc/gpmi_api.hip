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
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <limits>
#include <string>
#include <thread>
#include <vector>

#include "../../include/gpmi.h"
#include "gpmi_internal.h"

using namespace gpmi;

namespace {

thread_local std::string g_err;

int fail_runtime(hipError_t e, const char* what) {
    char buf[512];
    snprintf(buf, sizeof buf, "%s: %s (hipError %d)", what, hipGetErrorString(e), (int)e);
    g_err = buf;
    return GPMI_ERR_RUNTIME;
}
int fail_arg(const char* what) {
    g_err = what;
    return GPMI_ERR_BAD_ARG;
}

#define HIP_TRY(expr)                                             \
    do {                                                          \
        hipError_t _e = (expr);                                   \
        if (_e != hipSuccess) return fail_runtime(_e, #expr);     \
    } while (0)

inline int64_t round_up(int64_t x, int64_t m) { return (x + m - 1) / m * m; }

struct DevBuf {
    void* p = nullptr;
    size_t cap = 0;
    hipError_t ensure(size_t bytes) {
        if (bytes <= cap) return hipSuccess;
        if (p) { hipError_t e = hipFree(p); if (e != hipSuccess) return e; p = nullptr; cap = 0; }
        hipError_t e = hipMalloc(&p, bytes);
        if (e == hipSuccess) cap = bytes;
        return e;
    }
    void release() { if (p) (void)hipFree(p); p = nullptr; cap = 0; }
    template <class T> T* as() const { return reinterpret_cast<T*>(p); }
};

struct TimedSpan { hipEvent_t a, b; int slot; };

// Per-dimension bounding box of a point set (host side, at upload time).  Two boxes bound every
// squared distance of a kernel-matrix launch, which lets the squared-exponential build drop its
// per-wave exp domain test (RbfArgs::max_sq).  Non-finite inputs make the box invalid.
struct Box {
    std::vector<double> lo, hi;
    bool valid = false;
    void assign(const double* X, int64_t n, int64_t d) {
        lo.assign((size_t)d, std::numeric_limits<double>::infinity());
        hi.assign((size_t)d, -std::numeric_limits<double>::infinity());
        bool finite = n > 0;
        for (int64_t i = 0; i < n; ++i)
            for (int64_t k = 0; k < d; ++k) {
                const double v = X[i * d + k];
                finite &= std::isfinite(v);
                lo[(size_t)k] = std::min(lo[(size_t)k], v);
                hi[(size_t)k] = std::max(hi[(size_t)k], v);
            }
        valid = finite;
    }
};
double box_max_sq(const Box& a, const Box& b) {
    if (!a.valid || !b.valid || a.lo.size() != b.lo.size()) return -1.0;
    double s = 0.0;
    for (size_t k = 0; k < a.lo.size(); ++k) {
        const double w = std::max(a.hi[k] - b.lo[k], b.hi[k] - a.lo[k]);
        s += w * w;
    }
    return std::isfinite(s) ? s : -1.0;
}

}  // namespace

struct gpmi_ctx {
    int device = 0;
    hipStream_t stream = nullptr;    // main stream: K build, trailing updates, reductions
    hipStream_t pstream = nullptr;   // high-priority stream: panel factorisations (lookahead)
    // options
    int64_t nb = 0;         // outer block width of the Cholesky (multiple of 128); 0 = by size
    int64_t block(int64_t ncols) const { return nb ? nb : (ncols >= 49152 ? 2048 : ncols >= 24576 ? 1024 : 512); }
    int64_t ld_pad = 544;   // doubles added to every leading dimension
    int timing = 1;
    int lookahead = 1;      // factor panel k+1 while the rest of trailing update k runs
    int lanes = 0;          // gpmi_lml_batch: factorisations in flight (0 = by size)
    std::vector<gpmi_ctx*> lane_ctx;   // the extra lanes (own streams and workspaces), created on demand
    int ramp = 0;           // block widths ramp up at the start and down at the end of the sweep (measured: 0.4 % slower at N = 65536, off)
    // training set / factor
    int64_t N = 0, d = 0, Np = 0, ldA = 0, Mp = 0;
    bool have_train = false, have_factor = false;
    double sig2 = 1.0, coef = -0.5;
    int kind = 0;            // covariance function: 0 rbf, 1 linear, 2 periodic, 3 CO2 composite (gpmi_set_kernel*)
    double kp0 = 0., kp1 = 0.;
    double kpv[11] = {0., 0., 0., 0., 0., 0., 0., 0., 0., 0., 0.};
    DevBuf X, y, A, info, red;
    // test set
    int64_t n = 0, np_ = 0, ldV = 0, ldP = 0;
    bool have_test = false, have_v = false;
    std::vector<double> hXs; // host copy of the test inputs (diag(K_ss) of the linear kernel)
    Box boxX, boxXs;         // bounding boxes of the training / test inputs
    DevBuf Xs, V, P, vec, dense;
    DevBuf U, Kn, gpart;     // f2: L^-T, -(K+sI)^-1, per-tile partial sums of the gradient trace
    double sigma = 1.0, ell = 1.0;   // hyper-parameters of the resident factorisation
    // timers
    std::vector<hipEvent_t> ev_pool;
    size_t ev_used = 0;
    std::vector<TimedSpan> spans;
    double stage_ms[GPMI_T_COUNT] = {0};

    hipEvent_t new_event() {
        if (ev_used == ev_pool.size()) {
            hipEvent_t e;
            (void)hipEventCreate(&e);
            ev_pool.push_back(e);
        }
        return ev_pool[ev_used++];
    }
    size_t span_begin(int slot, hipStream_t st = nullptr) {
        if (!timing) return 0;
        TimedSpan s{new_event(), new_event(), slot};
        (void)hipEventRecord(s.a, st ? st : stream);
        spans.push_back(s);
        return spans.size() - 1;
    }
    void span_end(size_t idx, hipStream_t st = nullptr) {
        if (!timing) return;
        (void)hipEventRecord(spans[idx].b, st ? st : stream);
    }
    // make stream `waiter` wait for everything queued so far on `signaller`
    hipError_t order(hipStream_t signaller, hipStream_t waiter) {
        hipEvent_t e = new_event();
        hipError_t r = hipEventRecord(e, signaller);
        if (r != hipSuccess) return r;
        return hipStreamWaitEvent(waiter, e, 0);
    }
    void timers_reset(std::initializer_list<int> slots) {
        for (int s : slots) stage_ms[s] = 0.;
    }
    // call after the stream has been synchronised
    void timers_collect() {
        for (auto& s : spans) {
            float ms = 0.f;
            if (hipEventElapsedTime(&ms, s.a, s.b) == hipSuccess) stage_ms[s.slot] += ms;
        }
        spans.clear();
        ev_used = 0;
    }
};

namespace {

// ---------------------------------------------------------------------------
// Panel factorisation: the nb-wide block column whose diagonal block starts at
// A (global column col_offset), `mrows` rows tall (mrows >= nb, multiple of
// 128).  Recursive halving down to 64 columns:
//   factor the left half (all rows), update the right half with ONE MFMA GEMM of
//   depth = width of the left half, factor the right half.
// The leaves are the 64 x 64 potf2 and the substitution TRSM of every row below
// it.  Same flops and the same number of launches as a flat right-looking sweep
// in 64-column steps, but half of the update flops run at depth >= nb/4 instead of
// 64, and the panel is streamed 2.3x less often.
// ---------------------------------------------------------------------------
hipError_t panel_rec(hipStream_t s, double* A, int64_t ld, int64_t mrows, int64_t off, int64_t w,
                     int64_t col_offset, int64_t* info) {
    hipError_t e;
    if (w <= IB) {
        double* Ajj = A + off * ld + off;
        if ((e = launch_potf2_64(s, Ajj, ld, col_offset + off, info)) != hipSuccess) return e;
        const int64_t below = mrows - off - IB;
        if (below > 0) return launch_trsm_rlt64(s, Ajj, ld, A + (off + IB) * ld + off, ld, below);
        return hipSuccess;
    }
    const int64_t h = (w / 2) / IB * IB;           // left width (multiple of 64, >= 64)
    if ((e = panel_rec(s, A, ld, mrows, off, h, col_offset, info)) != hipSuccess) return e;
    {
        // right half -= (rows of the left half) * (its own rows of the left half)^T, lower part.
        // Rows start at the 128-aligned row at or above off+h: the extra 64 rows (when off+h is
        // not a multiple of 128) lie above the diagonal of the updated columns and are never read.
        const int64_t c0 = off + h;
        const int64_t r0 = c0 / TILE * TILE;
        GemmArgs g;
        g.C = A + r0 * ld + c0;
        g.A = A + r0 * ld + off;
        g.B = A + c0 * ld + off;
        g.ldc = g.lda = g.ldb = ld;
        g.M = mrows - r0; g.N = w - h; g.K = h;
        g.mode = 0; g.lower = 1; g.diag_off = r0 - c0;
        if ((e = launch_gemm_nt(s, g)) != hipSuccess) return e;
    }
    return panel_rec(s, A, ld, mrows, off + h, w - h, col_offset, info);
}

hipError_t panel_factor(hipStream_t s, double* A, int64_t ld, int64_t nb, int64_t mrows,
                        int64_t col_offset, int64_t* info) {
    return panel_rec(s, A, ld, mrows, 0, nb, col_offset, info);
}

// X (m x nb) <- X * L^-T, L nb x nb lower; m multiple of 128, nb multiple of 64.
// Same recursion: X1 <- X1 L11^-T;  X2 <- (X2 - X1 L21^T) L22^-T.
hipError_t trsm_rec(hipStream_t s, const double* L, int64_t ldl, double* X, int64_t ldx, int64_t m,
                    int64_t off, int64_t w) {
    hipError_t e;
    if (w <= IB) return launch_trsm_rlt64(s, L + off * ldl + off, ldl, X + off, ldx, m);
    const int64_t h = (w / 2) / IB * IB;
    if ((e = trsm_rec(s, L, ldl, X, ldx, m, off, h)) != hipSuccess) return e;
    GemmArgs g;
    g.C = X + off + h;
    g.A = X + off;
    g.B = L + (off + h) * ldl + off;
    g.ldc = g.lda = ldx; g.ldb = ldl;
    g.M = m; g.N = w - h; g.K = h;
    g.mode = 0; g.lower = 0; g.diag_off = 0;
    if ((e = launch_gemm_nt(s, g)) != hipSuccess) return e;
    return trsm_rec(s, L, ldl, X, ldx, m, off + h, w - h);
}

hipError_t trsm_block(hipStream_t s, const double* L, int64_t ldl, double* X, int64_t ldx,
                      int64_t m, int64_t nb) {
    return trsm_rec(s, L, ldl, X, ldx, m, 0, nb);
}

// Block widths of the sweep.  With a fixed width NB the first panel (NB columns x all rows) runs
// before there is any trailing update to hide it behind, and the last few panels are longer than
// the updates they overlap.  Option "ramp" lets the widths ramp up (NB/4, NB/4, NB/2, then NB)
// and down again over the last columns.  Measured at N = 65536: the exposed panel time drops by
// 7 ms but the narrower first updates cost 14 ms, so it is off by default.
std::vector<int64_t> block_schedule(const gpmi_ctx* c, int64_t ncols) {
    const int64_t NB = c->block(ncols);
    std::vector<int64_t> w;
    const bool ramp = c->nb == 0 && c->ramp && NB >= 1024 && ncols >= 8 * NB;
    int64_t done = 0;
    while (done < ncols) {
        int64_t nb = NB;
        if (ramp) {
            const int64_t left = ncols - done;
            if (w.size() < 2) nb = NB / 4;
            else if (w.size() < 3) nb = NB / 2;
            else if (left <= NB) nb = NB / 4;
            else if (left <= 3 * NB) nb = NB / 2;
        }
        nb = std::min(nb, ncols - done);
        w.push_back(nb);
        done += nb;
    }
    return w;
}

// In-place blocked right-looking Cholesky of the leading ncols x ncols block of
// A; rows ncols..nrows-1 are carried along (they end up multiplied by L^-T).
//
// With lookahead the trailing update of step k is split in two launches on the
// main stream: (a) the next block column only, (b) the rest.  The panel stream
// (high priority) factors panel k+1 as soon as (a) is done, i.e. concurrently
// with (b), whose tiles it neither reads nor writes.  Dependencies:
//   panel k  ->  (a)_k, (b)_k          (main waits on the panel event)
//   (a)_k    ->  panel k+1             (panel stream waits on the column event)
//   (b)_k    ->  (a)_{k+1}, (b)_{k+1}  (same stream)
hipError_t cholesky_inplace(gpmi_ctx* c, double* A, int64_t ld, int64_t ncols, int64_t nrows,
                            int64_t* info, bool account) {
    hipError_t e;
    hipStream_t sm = c->stream;
    const std::vector<int64_t> widths = block_schedule(c, ncols);
    const int64_t NB = c->block(ncols);
    // below ~12k columns the two-stream choreography costs more than the panel it hides
    const bool la = c->lookahead && c->pstream && ncols > NB && ncols >= 12288;
    hipStream_t sp_ = la ? c->pstream : sm;
    const int slot_p = account ? GPMI_T_CHOL_PANEL : GPMI_T_COUNT - 1;
    const int slot_t = account ? GPMI_T_CHOL_TRAIL : GPMI_T_COUNT - 1;
    if (la && (e = c->order(sm, sp_)) != hipSuccess) return e;   // panel 0 after the K build
    auto trail = [&](int64_t r0, int64_t c0, int64_t k, int64_t nb, int64_t ncol_upd) -> hipError_t {
        // C = A[r0.., c0..c0+ncol_upd) -= A[r0.., k..k+nb) * A[c0.., k..k+nb)^T, lower part
        GemmArgs g;
        g.C = A + r0 * ld + c0;
        g.A = A + r0 * ld + k;
        g.B = A + c0 * ld + k;
        g.ldc = g.lda = g.ldb = ld;
        g.M = nrows - r0; g.N = ncol_upd; g.K = nb;
        g.mode = 0; g.lower = 1; g.diag_off = r0 - c0;
        g.role = 1;
        // the roofline figures are those of the LDS-DMA kernel: the last, small updates that
        // run on the first-generation kernel are timed into the scratch slot
        const bool dma = gemm_nt_routes_dma(g);
        size_t sp = c->span_begin(dma ? slot_t : GPMI_T_COUNT - 1, sm);
        hipError_t er = launch_gemm_nt(sm, g);
        c->span_end(sp, sm);
        if (account && dma) {
            c->stage_ms[GPMI_T_TRAIL_LAUNCHES] += 1.0;
            // algorithmic: the lower triangle of the real rows plus the one row that carries y
            c->stage_ms[GPMI_T_TRAIL_FLOPS] += gemm_nt_algorithmic_flops(g, ncols - r0 + 1);
        }
        return er;
    };
    int64_t k = 0;
    for (size_t step = 0; step < widths.size(); ++step) {
        const int64_t nb = widths[step];
        size_t sp = c->span_begin(slot_p, sp_);
        e = panel_factor(sp_, A + k * ld + k, ld, nb, nrows - k, k, info);
        c->span_end(sp, sp_);
        if (e != hipSuccess) return e;
        if (la && (e = c->order(sp_, sm)) != hipSuccess) return e;
        const int64_t r0 = k + nb;
        k = r0;
        if (r0 >= ncols) continue;
        if (!la) {
            if ((e = trail(r0, r0, r0 - nb, nb, ncols - r0)) != hipSuccess) return e;
            continue;
        }
        const int64_t nbn = widths[step + 1];
        if ((e = trail(r0, r0, r0 - nb, nb, nbn)) != hipSuccess) return e;           // (a) next block column
        if ((e = c->order(sm, sp_)) != hipSuccess) return e;
        if (r0 + nbn < ncols &&
            (e = trail(r0 + nbn, r0 + nbn, r0 - nb, nb, ncols - r0 - nbn)) != hipSuccess) return e;  // (b) rest
    }
    return hipSuccess;
}

void set_kernel_args(const gpmi_ctx* c, RbfArgs& r) {
    r.coef = c->coef; r.sig2 = c->sig2;
    r.kind = c->kind; r.kp0 = c->kp0; r.kp1 = c->kp1;
    for (int i = 0; i < 11; ++i) r.kpv[i] = c->kpv[i];
}

int ensure_train_buffers(gpmi_ctx* c) {
    c->Np = round_up(c->N, TILE);
    c->ldA = c->Np + c->ld_pad;
    c->Mp = c->Np + TILE;
    HIP_TRY(c->A.ensure((size_t)c->Mp * c->ldA * sizeof(double)));
    HIP_TRY(c->info.ensure(sizeof(int64_t)));
    HIP_TRY(c->red.ensure(16 * sizeof(double)));
    return GPMI_OK;
}

// K build + Cholesky (+ forward solve through the y row) + LML on the stream
int factorize_impl(gpmi_ctx* c, double sigma, double ell, double noise_var, double* lml,
                   int64_t* bad_pivot) {
    if (!c->have_train) return fail_arg("gpmi_factorize: no training set (call gpmi_set_train)");
    if (c->kind == 0 && (!(ell != 0.0) || std::isnan(ell) || std::isnan(sigma)))
        return fail_arg("gpmi_factorize: ell must be non-zero and hyper-parameters finite");
    if (std::isnan(noise_var)) return fail_arg("gpmi_factorize: noise_var is NaN");
    if (c->kind == 2 && c->d != 1) return fail_arg("gpmi_factorize: the periodic kernel is 1-D only (GP_regression.py:48)");
    int rc = ensure_train_buffers(c);
    if (rc) return rc;
    hipStream_t s = c->stream;
    c->have_factor = false;
    c->have_v = false;
    c->timers_reset({GPMI_T_KBUILD, GPMI_T_CHOL, GPMI_T_CHOL_PANEL, GPMI_T_CHOL_TRAIL, GPMI_T_LML,
                     GPMI_T_TRAIL_LAUNCHES, GPMI_T_TRAIL_FLOPS});
    c->sig2 = sigma * sigma;
    c->coef = -.5 * (1 / (ell * ell));      // GP_regression.py:19 evaluation order
    c->sigma = sigma; c->ell = ell;
    double* A = c->A.as<double>();
    const int64_t big = std::numeric_limits<int64_t>::max();
    HIP_TRY(hipMemcpyAsync(c->info.p, &big, sizeof big, hipMemcpyHostToDevice, s));

    size_t sp = c->span_begin(GPMI_T_KBUILD);
    RbfArgs r;
    r.A = r.B = c->X.as<double>();
    r.nA = r.nB = c->N; r.d = c->d; r.row0 = 0; r.nrows = c->Np; r.ncols = c->Np;
    set_kernel_args(c, r);
    r.diag_add = noise_var; r.symmetric = 1; r.delta_square = 1;
    r.max_sq = box_max_sq(c->boxX, c->boxX);
    r.out = A; r.ld = c->ldA;
    HIP_TRY(launch_rbf(s, r));
    // the augmented rows: y then zeros
    HIP_TRY(launch_fill_rows(s, A + c->Np * c->ldA, c->ldA, TILE, c->Np, 0.0));
    HIP_TRY(launch_set_yrow(s, A + c->Np * c->ldA, c->y.as<double>(), c->N, c->Np));
    c->span_end(sp);

    sp = c->span_begin(GPMI_T_CHOL);
    HIP_TRY(cholesky_inplace(c, A, c->ldA, c->Np, c->Mp, c->info.as<int64_t>(), true));
    c->span_end(sp);

    sp = c->span_begin(GPMI_T_LML);
    HIP_TRY(launch_lml_reduce(s, A, c->ldA, A + c->Np * c->ldA, c->N, c->red.as<double>()));
    c->span_end(sp);

    double red[2];
    int64_t info;
    HIP_TRY(hipMemcpyAsync(red, c->red.p, sizeof red, hipMemcpyDeviceToHost, s));
    HIP_TRY(hipMemcpyAsync(&info, c->info.p, sizeof info, hipMemcpyDeviceToHost, s));
    HIP_TRY(hipStreamSynchronize(s));
    c->timers_collect();
    if (info != big && info < c->N) {
        if (bad_pivot) *bad_pivot = info + 1;
        if (lml) *lml = std::numeric_limits<double>::quiet_NaN();
        g_err = "Matrix is not positive definite";
        return GPMI_ERR_NOT_PD;
    }
    if (bad_pivot) *bad_pivot = 0;
    // tune_hyperparms_regression.py:312, with y^T alpha = m^T m
    if (lml) *lml = -.5 * red[1] - red[0] - (double)c->N / 2.0 * std::log(2 * M_PI);
    c->have_factor = true;
    return GPMI_OK;
}

// v^T = K_s^T L^-T: right-looking sweep over the block columns of L, with the
// same lookahead split as the Cholesky (the triangular solve of block column
// k+1 overlaps the update of the columns beyond it).
// tri: V starts as the identity (m == Np), so at step k only rows < k + nb are non-zero in
// block column k -- the sweep then costs Np^3/3 and leaves the upper triangular L^-T.
hipError_t solve_sweep(gpmi_ctx* c, double* V, int64_t ldv, int64_t m, bool tri = false) {
    hipError_t e;
    hipStream_t sm = c->stream;
    const double* A = c->A.as<double>();
    const int64_t ld = c->ldA, Np = c->Np;
    const int64_t NB = c->block(Np);
    const bool la = c->lookahead && c->pstream && Np > NB && Np >= 12288;
    hipStream_t sp_ = la ? c->pstream : sm;
    if (la && (e = c->order(sm, sp_)) != hipSuccess) return e;
    auto update = [&](int64_t c0, int64_t k, int64_t nb, int64_t ncol_upd) -> hipError_t {
        GemmArgs g;   // V[:, c0..c0+ncol_upd) -= V[:, k..k+nb) * L[c0.., k..k+nb)^T
        g.C = V + c0; g.A = V + k; g.B = A + c0 * ld + k;
        g.ldc = g.lda = ldv; g.ldb = ld;
        g.M = tri ? std::min(m, k + nb) : m; g.N = ncol_upd; g.K = nb;
        g.mode = 0; g.lower = 0; g.diag_off = 0;
        return launch_gemm_nt(sm, g);
    };
    for (int64_t k = 0; k < Np; k += NB) {
        const int64_t nb = std::min<int64_t>(NB, Np - k);
        if ((e = trsm_block(sp_, A + k * ld + k, ld, V + k, ldv, tri ? std::min(m, k + nb) : m, nb)) != hipSuccess) return e;
        if (la && (e = c->order(sp_, sm)) != hipSuccess) return e;
        const int64_t r0 = k + nb;
        if (r0 >= Np) continue;
        if (!la) {
            if ((e = update(r0, k, nb, Np - r0)) != hipSuccess) return e;
            continue;
        }
        const int64_t nbn = std::min<int64_t>(NB, Np - r0);
        if ((e = update(r0, k, nb, nbn)) != hipSuccess) return e;
        if ((e = c->order(sm, sp_)) != hipSuccess) return e;
        if (r0 + nbn < Np && (e = update(r0 + nbn, k, nb, Np - r0 - nbn)) != hipSuccess) return e;
    }
    return hipSuccess;
}

}  // namespace

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
                       &c->U, &c->Kn, &c->gpart})
        b->release();
    for (auto e : c->ev_pool) (void)hipEventDestroy(e);
    (void)hipStreamDestroy(c->stream);
    (void)hipStreamDestroy(c->pstream);
    delete c;
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
    } else if (!strcmp(name, "lanes")) {
        if (value < 0 || value > 8) return fail_arg("lanes must be 0 (by size) .. 8");
        c->lanes = (int)value;
    } else if (!strcmp(name, "ramp")) {
        c->ramp = value ? 1 : 0;
    } else if (!strcmp(name, "gemm_small_tiles")) {
        g_gemm_small_tiles = value ? 1 : 0;
    } else if (!strcmp(name, "trsm_wave")) {
        g_trsm_wave = value ? 1 : 0;
    } else if (!strcmp(name, "rbf_blocks")) {
        if (value < 1 || value > (1 << 24)) return fail_arg("rbf_blocks must be in 1..2^24");
        g_rbf_blocks = (int)value;
    } else if (!strcmp(name, "gemm_dma_waves")) {
        if (value != 4 && value != 8) return fail_arg("gemm_dma_waves must be 4 or 8");
        g_gemm_dma_waves = (int)value;
    } else if (!strcmp(name, "gemm_dma")) {
        g_gemm_use_dma = value ? 1 : 0;
    } else {
        return fail_arg("gpmi_set_option: unknown option");
    }
    return GPMI_OK;
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
    hipStream_t s = c->stream;
    DevBuf da, db, dout;
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
    da.release(); db.release(); dout.release();
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
    HIP_TRY(hipMemcpyAsync(m_out, c->A.as<double>() + c->Np * c->ldA, (size_t)c->N * 8,
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

int gpmi_get_alpha(gpmi_ctx* c, double* alpha_out) {
    if (!c || !alpha_out) return fail_arg("gpmi_get_alpha: null argument");
    if (!c->have_factor) return fail_arg("gpmi_get_alpha: no factorisation resident");
    HIP_TRY(hipSetDevice(c->device));
    hipStream_t s = c->stream;
    c->timers_reset({GPMI_T_ALPHA});
    HIP_TRY(c->vec.ensure((size_t)std::max(c->Np, c->np_) * 4 * 8));
    double* x = c->vec.as<double>();
    // padded tail of m is zero (identity padding), so the padded system stays consistent
    HIP_TRY(hipMemcpyAsync(x, c->A.as<double>() + c->Np * c->ldA, (size_t)c->Np * 8,
                           hipMemcpyDeviceToDevice, s));
    size_t sp = c->span_begin(GPMI_T_ALPHA);
    HIP_TRY(launch_trsv_lt(s, c->A.as<double>(), c->ldA, x, c->Np));
    c->span_end(sp);
    HIP_TRY(hipMemcpyAsync(alpha_out, x, (size_t)c->N * 8, hipMemcpyDeviceToHost, s));
    HIP_TRY(hipStreamSynchronize(s));
    c->timers_collect();
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
    hipStream_t s = c->stream;
    c->timers_reset({GPMI_T_KS, GPMI_T_SOLVE_V, GPMI_T_MEANVAR});
    c->have_v = false;
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
    HIP_TRY(launch_row_dots(s, V, c->ldV, c->np_, c->Np, c->A.as<double>() + c->Np * c->ldA, dot, sq));
    c->span_end(sp);

    std::vector<double> h(2 * (size_t)c->np_);
    HIP_TRY(hipMemcpyAsync(h.data(), dot, h.size() * 8, hipMemcpyDeviceToHost, s));
    HIP_TRY(hipStreamSynchronize(s));
    c->timers_collect();
    c->have_v = true;
    for (int64_t i = 0; i < c->n; ++i) {
        if (mu) mu[i] = h[i];
        if (out2) {
            double kss = c->sig2;                          // diag(K_ss) == sigma^2 exactly for the RBF (GP_regression.py:147)
            if (c->kind == 2) kss = 1.0;                   // periodic: exp(0)
            else if (c->kind == 3) {                       // composite at sqdist 0, square K_ss: every factor is 1
                const double* th = c->kpv;
                kss = ((th[0] * th[0] + th[2] * th[2]) + th[5] * th[5]) + (th[8] * th[8] + th[10] * th[10]);
            }
            else if (c->kind == 1) {                       // linear: (x - c).(x - c)
                kss = 0.0;
                for (int64_t k = 0; k < c->d; ++k) {
                    const double e = c->hXs[(size_t)i * c->d + k] - c->kp0;
                    kss = kss + e * e;
                }
            }
            const double var = kss - h[c->np_ + i];
            out2[i] = want_sd ? std::sqrt(var) : var;      // sqrt(<0) -> NaN like np.sqrt (:148)
        }
    }
    return GPMI_OK;
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
    hipStream_t s = c->stream;
    const int64_t Np = c->Np, ld = c->ldA;
    c->timers_reset({GPMI_T_GRAD});
    HIP_TRY(c->U.ensure((size_t)Np * ld * 8));
    HIP_TRY(c->Kn.ensure((size_t)Np * ld * 8));
    HIP_TRY(c->vec.ensure((size_t)std::max(c->Np, c->np_) * 4 * 8));
    size_t sp = c->span_begin(GPMI_T_GRAD);
    // alpha = L^-T m (a5)
    double* alpha = c->vec.as<double>();
    HIP_TRY(hipMemcpyAsync(alpha, c->A.as<double>() + Np * ld, (size_t)Np * 8, hipMemcpyDeviceToDevice, s));
    HIP_TRY(launch_trsv_lt(s, c->A.as<double>(), ld, alpha, Np));
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
    HIP_TRY(hipStreamSynchronize(s));
    c->timers_collect();
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

int gpmi_post_chol(gpmi_ctx* c, double jitter, double* L_out, int64_t* bad_pivot) {
    if (!c || !L_out) return fail_arg("gpmi_post_chol: null argument");
    if (!c->have_v) return fail_arg("gpmi_post_chol: run gpmi_predict first");
    HIP_TRY(hipSetDevice(c->device));
    hipStream_t s = c->stream;
    c->timers_reset({GPMI_T_POSTCHOL});
    const int64_t np_ = c->np_, n = c->n;
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
    g.C = P; g.A = g.B = c->V.as<double>();
    g.ldc = c->ldP; g.lda = g.ldb = c->ldV;
    g.M = g.N = np_; g.K = c->Np; g.mode = 0; g.lower = 1; g.diag_off = 0;
    HIP_TRY(launch_gemm_nt(s, g));
    HIP_TRY(cholesky_inplace(c, P, c->ldP, np_, np_, c->info.as<int64_t>(), false));
    c->span_end(sp);
    HIP_TRY(c->dense.ensure((size_t)n * n * 8));
    HIP_TRY(launch_extract(s, P, c->ldP, 0, n, 0, n, c->dense.as<double>(), 1));
    int64_t info;
    HIP_TRY(hipMemcpyAsync(L_out, c->dense.p, (size_t)n * n * 8, hipMemcpyDeviceToHost, s));
    HIP_TRY(hipMemcpyAsync(&info, c->info.p, sizeof info, hipMemcpyDeviceToHost, s));
    HIP_TRY(hipStreamSynchronize(s));
    c->timers_collect();
    if (info != big && info < n) {
        if (bad_pivot) *bad_pivot = info + 1;
        g_err = "Matrix is not positive definite";
        return GPMI_ERR_NOT_PD;
    }
    if (bad_pivot) *bad_pivot = 0;
    return GPMI_OK;
}

// Lane l of a batch: its own context (streams, A, workspaces) on the same device, with the parent's
// training set copied device to device.  Small and mid-size factorisations leave most of the chip idle
// in their latency-bound panel steps; two in flight fill it (measured per triple, 1 -> 2 lanes: N = 512
// 0.45 -> 0.24 ms, N = 2048 1.74 -> 0.93 ms, N = 8192 10.5 -> 6.9 ms, N = 16384 39.5 -> 32.1 ms,
// N = 32768 213 -> 201 ms).
static int lane_prepare(gpmi_ctx* c, gpmi_ctx* l) {
    l->nb = c->nb; l->ld_pad = c->ld_pad; l->lookahead = c->lookahead; l->ramp = c->ramp; l->timing = c->timing;
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
    // two lanes = four streams = the four hardware queues a process gets; more lanes share queues and
    // serialise again (measured, 12 triples: N = 512 0.45 / 0.24 / 0.38 / 0.31 ms per triple with 1 / 2 / 3 / 4 lanes)
    int L = c->lanes ? c->lanes : (Np <= 32768 ? 2 : 1);
    L = (int)std::min<int64_t>(L, std::max<int64_t>(T, 1));
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

// out[0] = TFLOP/s, out[1] = shader clock (GHz) held during the loop,
// out[2] = shader cycles per MFMA per SIMD
int gpmi_probe_mfma_f64_ex(gpmi_ctx* c, int blocks_per_cu, int nacc, int iters, double* out) {
    if (!c || !out) return fail_arg("gpmi_probe_mfma_f64_ex: null argument");
    if (blocks_per_cu < 1 || blocks_per_cu > 8 || iters < 1) return fail_arg("gpmi_probe_mfma_f64_ex: bad argument");
    if (nacc != 4 && nacc != 8 && nacc != 16) return fail_arg("gpmi_probe_mfma_f64_ex: nacc must be 4, 8 or 16");
    HIP_TRY(hipSetDevice(c->device));
    HIP_TRY(c->red.ensure(16 * 8));
    hipStream_t s = c->stream;
    hipDeviceProp_t prop;
    HIP_TRY(hipGetDeviceProperties(&prop, c->device));
    const int blocks = prop.multiProcessorCount * blocks_per_cu;
    double* sink = c->red.as<double>();
    unsigned long long* clk = reinterpret_cast<unsigned long long*>(sink + 8);
    HIP_TRY(launch_probe_mfma(s, sink, 64, blocks, nacc, clk));   // warm-up
    hipEvent_t a, b;
    HIP_TRY(hipEventCreate(&a));
    HIP_TRY(hipEventCreate(&b));
    HIP_TRY(hipEventRecord(a, s));
    HIP_TRY(launch_probe_mfma(s, sink, iters, blocks, nacc, clk));
    HIP_TRY(hipEventRecord(b, s));
    HIP_TRY(hipEventSynchronize(b));
    float ms = 0.f;
    HIP_TRY(hipEventElapsedTime(&ms, a, b));
    (void)hipEventDestroy(a); (void)hipEventDestroy(b);
    unsigned long long h[2];
    HIP_TRY(hipMemcpy(h, clk, sizeof h, hipMemcpyDeviceToHost));
    const double n_mfma_wave = (double)iters * nacc;
    out[0] = (double)blocks * 4 * n_mfma_wave * 2048.0 / (ms * 1e-3) / 1e12;
    out[1] = h[1] ? (double)h[0] / ((double)h[1] * 10.0) : 0.;      // s_memrealtime ticks at 100 MHz
    out[2] = (double)h[0] / (n_mfma_wave * blocks_per_cu);           // waves per SIMD = blocks per CU
    return GPMI_OK;
}

int gpmi_probe_mfma_f64(gpmi_ctx* c, double* tflops) {
    if (!tflops) return fail_arg("gpmi_probe_mfma_f64: null argument");
    double out[3];
    int rc = gpmi_probe_mfma_f64_ex(c, 2, 16, 2048, out);
    if (rc == GPMI_OK) *tflops = out[0];
    return rc;
}

// Timing of one GEMM launch shape on scratch buffers (results discarded).
// variant: ablation bits (1: no global loads in the K loop, 2: no LDS writes / barriers,
// 4: epilogue without the C read, 8: no epilogue).  out[0] = TFLOP/s over computed tiles,
// out[1] = ms per launch.
int gpmi_probe_gemm(gpmi_ctx* c, int64_t M, int64_t N, int64_t K, int lower, int variant, int reps,
                    double* out) {
    if (!c || !out) return fail_arg("gpmi_probe_gemm: null argument");
    if (M <= 0 || N <= 0 || K <= 0 || M % TILE || N % IB || K % 16 || reps < 1)
        return fail_arg("gpmi_probe_gemm: M%128, N%64, K%16 must be 0");
    HIP_TRY(hipSetDevice(c->device));
    hipStream_t s = c->stream;
    const int64_t ldc = N + c->ld_pad, ldk = K + c->ld_pad;
    DevBuf C, A, B;
    int rc = GPMI_OK;
    hipError_t e;
    hipEvent_t ea = nullptr, eb = nullptr;
    do {
        if ((e = C.ensure((size_t)M * ldc * 8)) != hipSuccess || (e = A.ensure((size_t)M * ldk * 8)) != hipSuccess ||
            (e = B.ensure((size_t)N * ldk * 8)) != hipSuccess) { rc = fail_runtime(e, "hipMalloc"); break; }
        (void)hipMemsetAsync(C.p, 0, (size_t)M * ldc * 8, s);
        (void)launch_fill_rows(s, A.as<double>(), ldk, M, K, 0.001);
        (void)launch_fill_rows(s, B.as<double>(), ldk, N, K, -0.002);
        GemmArgs g;
        g.C = C.as<double>(); g.A = A.as<double>(); g.B = B.as<double>();
        g.ldc = ldc; g.lda = g.ldb = ldk; g.M = M; g.N = N; g.K = K; g.mode = 0; g.lower = lower; g.diag_off = 0;
        DevBuf stamps;
        if (variant & 16) {
            if ((e = stamps.ensure(4096 * 16 * 8)) != hipSuccess) { rc = fail_runtime(e, "hipMalloc"); break; }
            (void)hipMemsetAsync(stamps.p, 0, 4096 * 16 * 8, s);
            g_gemm_stamps = stamps.as<unsigned long long>();
        }
        g_gemm_dbg = variant;
        e = launch_gemm_nt(s, g);
        (void)hipEventCreate(&ea); (void)hipEventCreate(&eb);
        (void)hipEventRecord(ea, s);
        for (int r = 0; r < reps && e == hipSuccess; ++r) e = launch_gemm_nt(s, g);
        (void)hipEventRecord(eb, s);
        hipError_t e2 = hipEventSynchronize(eb);
        g_gemm_dbg = 0;
        if (e != hipSuccess) { rc = fail_runtime(e, "gemm launch"); break; }
        if (e2 != hipSuccess) { rc = fail_runtime(e2, "gemm sync"); break; }
        float ms = 0.f;
        (void)hipEventElapsedTime(&ms, ea, eb);
        out[1] = ms / reps;
        out[0] = gemm_nt_flops(g) / (out[1] * 1e-3) / 1e12;
        if (variant & 16) {
            std::vector<unsigned long long> h(4096 * 16);
            (void)hipMemcpy(h.data(), stamps.p, h.size() * 8, hipMemcpyDeviceToHost);
            double sum[4] = {0, 0, 0, 0};
            int cnt = 0;
            for (size_t i = 0; i < h.size(); i += 4)
                if (h[i + 1]) { for (int q = 0; q < 4; ++q) sum[q] += (double)h[i + q]; ++cnt; }
            if (cnt) fprintf(stderr, "[gemm stamps] waves %d: prologue %.0f  loop %.0f  epilogue-loads %.0f  epilogue-stores %.0f cycles\n",
                             cnt, sum[0] / cnt, sum[1] / cnt, sum[2] / cnt, sum[3] / cnt);
            g_gemm_stamps = nullptr;
            stamps.release();
        }
    } while (0);
    g_gemm_dbg = 0;
    if (ea) (void)hipEventDestroy(ea);
    if (eb) (void)hipEventDestroy(eb);
    (void)hipStreamSynchronize(s);
    C.release(); A.release(); B.release();
    return rc;
}

int gpmi_probe_hbm_ex(gpmi_ctx* c, int64_t bytes, int mode, int blocks, double* gbps) {
    if (!c || !gbps || bytes < 4096 || blocks < 1 || mode < 0 || mode > 6) return fail_arg("gpmi_probe_hbm_ex: bad argument");
    HIP_TRY(hipSetDevice(c->device));
    HIP_TRY(c->red.ensure(16 * 8));
    hipStream_t s = c->stream;
    DevBuf buf;
    HIP_TRY(buf.ensure((size_t)bytes));
    hipError_t e = launch_probe_write(s, buf.as<double>(), bytes / 8, mode == 4 ? 0 : mode, blocks, c->red.as<double>());
    hipEvent_t a, b;
    (void)hipEventCreate(&a); (void)hipEventCreate(&b);
    (void)hipEventRecord(a, s);
    const int reps = 3;
    for (int r = 0; r < reps && e == hipSuccess; ++r)
        e = launch_probe_write(s, buf.as<double>(), bytes / 8, mode, blocks, c->red.as<double>());
    (void)hipEventRecord(b, s);
    hipError_t e2 = hipEventSynchronize(b);
    float ms = 0.f;
    (void)hipEventElapsedTime(&ms, a, b);
    (void)hipEventDestroy(a); (void)hipEventDestroy(b);
    buf.release();
    if (e != hipSuccess) return fail_runtime(e, "probe kernel");
    if (e2 != hipSuccess) return fail_runtime(e2, "probe sync");
    *gbps = (double)bytes * reps / (ms * 1e-3) / 1e9;
    return GPMI_OK;
}

int gpmi_probe_hbm_write(gpmi_ctx* c, int64_t bytes, double* gbps) {
    return gpmi_probe_hbm_ex(c, bytes, 0, 2048, gbps);
}

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

int gpmi_dev_row_dots(void* stream, const double* V_dev, int64_t ld, int64_t nrows, int64_t ncols,
                      const double* m_dev, double* dot_out_dev, double* sq_out_dev) {
    if (!V_dev || !m_dev) return fail_arg("gpmi_dev_row_dots: null pointer");
    if (ncols % 2 || ld % 2) return fail_arg("gpmi_dev_row_dots: ncols and ld must be even");
    HIP_TRY(launch_row_dots((hipStream_t)stream, V_dev, ld, nrows, ncols, m_dev, dot_out_dev, sq_out_dev));
    return GPMI_OK;
}

}  // extern "C"
