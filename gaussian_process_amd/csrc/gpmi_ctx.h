// Internal to libgpmi355x.so: the context object behind the opaque gpmi_ctx handle, the error
// helpers every translation unit of the C-ABI shares, and the single-GPU drivers (driver.hip).
#pragma once
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

namespace gpmi {

extern thread_local std::string g_err;      // text behind gpmi_last_error() (gpmi_api.hip)

inline int fail_runtime(hipError_t e, const char* what) {
    char buf[512];
    snprintf(buf, sizeof buf, "%s: %s (hipError %d)", what, hipGetErrorString(e), (int)e);
    g_err = buf;
    return GPMI_ERR_RUNTIME;
}
inline int fail_arg(const char* what) {
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
inline double box_max_sq(const Box& a, const Box& b) {
    if (!a.valid || !b.valid || a.lo.size() != b.lo.size()) return -1.0;
    double s = 0.0;
    for (size_t k = 0; k < a.lo.size(); ++k) {
        const double w = std::max(a.hi[k] - b.lo[k], b.hi[k] - a.lo[k]);
        s += w * w;
    }
    return std::isfinite(s) ? s : -1.0;
}

}  // namespace gpmi

// the handle type of include/gpmi.h lives at global scope
using gpmi::Box;
using gpmi::DevBuf;
using gpmi::TimedSpan;

struct gpmi_ctx {
    int device = 0;
    hipStream_t stream = nullptr;    // main stream: K build, trailing updates, reductions
    hipStream_t pstream = nullptr;   // high-priority stream: panel factorisations (lookahead)
    hipStream_t vstream = nullptr;   // one-pass prediction, form "follow": the test set's sweep, one block column behind the Cholesky
    hipStream_t sstream = nullptr;   // experiment (option potrf_server): the stream the resident potrf128 workgroup lives on
    gpmi::PotrfServerState pserver;  // its mailbox and sequence numbers
    DevBuf pmail;
    // options
    int64_t nb = 0;         // outer block width of the Cholesky (multiple of 128); 0 = by size
    int64_t block(int64_t ncols) const { return nb ? nb : (ncols >= 32768 ? 2048 : ncols >= 12288 ? 1024 : 512); }
    int64_t ld_pad = 544;   // doubles added to every leading dimension
    int timing = 1;
    int lookahead = 1;      // factor panel k+1 while the rest of trailing update k runs
    int64_t la_min = 6144;  // ... from this many columns on (below, the two-stream choreography costs more than the panel it hides:
                            // profiles/r04_la_min_sweep.txt -- one pass / two calls / fit alone at N = 4096: 3.06 -> 2.96 / 3.82 -> 3.86 /
                            // 2.56 -> 2.59 ms, 6144: 5.41 -> 5.20 / 6.69 -> 6.50 / 4.67 -> 4.48, 10240: 13.8 -> 12.6 / 15.9 -> 14.7; same bits)
    int64_t shallow_min = 6144; // under lookahead, panels with fewer columns left than this use the one-launch panel kernels (0: never)
    int one_pass_form = 0;  // gpmi_fit_predict_resident: 1 the test set's rows ride in the panel and update launches, 2 they follow on a
                            // stream of their own (panel k done -> their solve against L_kk -> their update), 0 = by size
    int lanes = 0;          // gpmi_lml_batch: factorisations in flight (0 = by size)
    std::vector<gpmi_ctx*> lane_ctx;   // the extra lanes (own streams and workspaces), created on demand
    int ramp = 0;           // block-width schedule, bit mask: 1 ramp up at the start, 2 half width over the last blocks (count in bits 4.., default 3),
                            // 4 quarter width for the last block (measured: ramp up 0.4 % slower at N = 65536; ramp down within noise at N = 16384 / 32768: off)
    gpmi::Tuning tune;      // kernel-selection options of this context (installed per call: gpmi::TuneScope)
    // training set / factor
    int64_t N = 0, d = 0, Np = 0, ldA = 0, Mp = 0;
    bool have_train = false, have_factor = false;
    // The resident factor's 128 x 128 diagonal blocks carry their full inverses (launch_vinv128; set by the first backward
    // solve).  OWNERSHIP: from then on the strict block-upper 16 x 16 tiles of every diagonal block of A hold L_kk^-T, not
    // zeros and not K: nothing but the backward-solve kernels may read them (every other consumer of a diagonal block masks
    // to the lower triangle; gpmi_get_factor_block zeroes the upper triangle on the way out).
    bool have_vinv = false;
    int factor_fused = 1;    // the resident factor came from the fused panel kernels: its diagonal 16 x 16 tiles carry
                             // their inverses above the diagonal, which trsm128 reads (panel_mfma.hip)
    double sig2 = 1.0, coef = -0.5;
    int kind = 0;            // covariance function: 0 rbf, 1 linear, 2 periodic, 3 CO2 composite (gpmi_set_kernel*)
    double kp0 = 0., kp1 = 0.;
    double kpv[11] = {0., 0., 0., 0., 0., 0., 0., 0., 0., 0., 0.};
    DevBuf X, y, A, info, red;
    // test set
    int64_t n = 0, np_ = 0, ldV = 0, ldP = 0;
    bool have_test = false, have_v = false;
    bool v_in_A = false;     // v^T is resident in rows of A (gpmi_fit_predict_resident: from row v_row0), not in V
    int64_t v_row0 = 0;
    double* v_rows() { return v_in_A ? A.as<double>() + v_row0 * ldA : V.as<double>(); }
    // the y rows: row Np of A -- or, when the posterior factor rides as well (gpmi_fit_predict_sample_resident), behind the
    // test rows, whose own n_p columns then follow the training columns
    int64_t yrow = 0;
    double* m_row() { return A.as<double>() + yrow * ldA; }
    bool post_in_A = false;  // A[Np.., Np..] holds cholesky(K_ss + post_jitter I - v^T v) of the resident test set
    double post_jitter = 0.0;
    bool post_in_P = false;  // ... or P does (gpmi_post_chol / gpmi_post_sample on a resident v), for post_jitter_P and the
    double post_jitter_P = 0.0;   // v of generation post_gen_P (v_gen counts every (re)computation of v)
    uint64_t v_gen = 0, post_gen_P = 0;
    std::vector<double> hXs; // host copy of the test inputs (diag(K_ss) of the linear kernel)
    Box boxX, boxXs;         // bounding boxes of the training / test inputs
    DevBuf Xs, V, P, vec, dense;
    DevBuf flag;             // one int the single-launch backward solve sets if a poll gave up
    DevBuf vside;            // Np x 128: the inverses of the 128 x 128 diagonal blocks, row-major (launch_vinv128's side buffer)
    bool have_vside = false; // vside matches the resident factor
    DevBuf cov_a, cov_b, cov_out;   // gpmi_rbf / gpmi_cov staging, kept across calls (the BO loops call them hundreds of times)
    DevBuf U, Kn, gpart;     // f2: L^-T, -(K+sI)^-1, per-tile partial sums of the gradient trace
    double sigma = 1.0, ell = 1.0;   // hyper-parameters of the resident factorisation
    // timers
    std::vector<hipEvent_t> ev_pool;
    size_t ev_used = 0;
    std::vector<TimedSpan> spans;
    double stage_ms[GPMI_T_COUNT] = {0};

    // nullptr (and ev_error set) if the runtime cannot create another event
    hipError_t ev_error = hipSuccess;
    hipEvent_t new_event() {
        if (ev_used == ev_pool.size()) {
            hipEvent_t e = nullptr;
            const hipError_t r = hipEventCreateWithFlags(&e, hipEventDefault);
            if (r != hipSuccess) { ev_error = r; return nullptr; }
            ev_pool.push_back(e);
        }
        return ev_pool[ev_used++];
    }
    static constexpr size_t NO_SPAN = (size_t)-1;
    // a span that cannot get its events is dropped (timers are diagnostics); ordering events are not optional
    size_t span_begin(int slot, hipStream_t st = nullptr) {
        if (!timing) return NO_SPAN;
        TimedSpan s{new_event(), new_event(), slot};
        if (!s.a || !s.b) return NO_SPAN;
        if (hipEventRecord(s.a, st ? st : stream) != hipSuccess) return NO_SPAN;
        spans.push_back(s);
        return spans.size() - 1;
    }
    void span_end(size_t idx, hipStream_t st = nullptr) {
        if (!timing || idx == NO_SPAN) return;
        if (hipEventRecord(spans[idx].b, st ? st : stream) != hipSuccess) spans[idx].slot = GPMI_T_COUNT - 1;
    }
    // make stream `waiter` wait for everything queued so far on `signaller`
    hipError_t order(hipStream_t signaller, hipStream_t waiter) {
        hipEvent_t e = new_event();
        if (!e) return ev_error;
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


namespace gpmi {

// driver.hip
hipError_t panel_factor(hipStream_t s, double* A, int64_t ld, int64_t nb, int64_t mrows, int64_t col_offset,
                        int64_t* info);
hipError_t trsm_block(hipStream_t s, const double* L, int64_t ldl, double* X, int64_t ldx, int64_t m, int64_t nb);
// rows that follow the factorisation on a stream of their own: V (m x ncols, leading dimension ldv) <- V L^-T
struct SweepFollower {
    double* V = nullptr;
    int64_t ldv = 0, m = 0;
    hipStream_t vs = nullptr;
};
hipError_t cholesky_inplace(gpmi_ctx* c, double* A, int64_t ld, int64_t ncols, int64_t nrows, int64_t* info,
                            bool account, int64_t carried_rows = 0, const SweepFollower* follow = nullptr);
hipError_t solve_sweep(gpmi_ctx* c, double* V, int64_t ldv, int64_t m, bool tri = false);
void set_kernel_args(const gpmi_ctx* c, RbfArgs& r);
int ensure_train_buffers(gpmi_ctx* c, int64_t test_rows = 0, bool test_cols = false);
// with_test: the test set's rows K(X*, X) ride below the y rows (they come out as v^T = K_s^T L^-T, a7 inside a3) and
// mean / variance (a6, a8) are read off them behind the LML
// with_post (needs with_test): the test rows also get their own columns K_ss + jitter I, i.e. ONE Cholesky of
//   [[K + sI, .], [K_s^T, K_ss + jitter I]]  (N + n columns)  --  its last n columns are cholesky(K_ss + jitter I - v^T v), f1
int factorize_impl(gpmi_ctx* c, double sigma, double ell, double noise_var, double* lml, int64_t* bad_pivot,
                   bool with_test = false, double* mu = nullptr, double* out2 = nullptr, int want_sd = 1,
                   bool with_post = false, double jitter = 0.0);
void meanvar_to_host(gpmi_ctx* c, const std::vector<double>& h, double* mu, double* out2, int want_sd);

}  // namespace gpmi
