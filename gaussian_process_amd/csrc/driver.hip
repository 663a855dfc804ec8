// Single-GPU drivers behind the C-ABI: recursive panel factorisation, the recursive triangular
// solve of a block, the blocked right-looking Cholesky with lookahead, the TRSM sweep for
// v = L^-1 K_s (and for L^-T), and the fit (K build + Cholesky + LML) they add up to.
// Data layout: see gpmi_api.hip.
#include "gpmi_ctx.h"
#include "gpmi_plan.h"

namespace gpmi {

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
    // leaves: 128 columns on the matrix pipe (potrf128 + trsm128, panel_mfma.hip), or -- option
    // "panel_fused" 0, and widths that are not a multiple of 128 -- the first-generation 64-column pair
    const bool fused = tuning().panel_fused && w % 128 == 0 && off % 128 == 0;
    const int64_t leaf = fused ? 128 : IB;
    if (w <= leaf) {
        double* Ajj = A + off * ld + off;
        const int64_t below = mrows - off - leaf;
        if (fused) {
            if ((e = launch_potrf128(s, Ajj, ld, col_offset + off, info)) != hipSuccess) return e;
            if (below > 0) return launch_trsm128(s, Ajj, ld, A + (off + leaf) * ld + off, ld, below);
            return hipSuccess;
        }
        if ((e = launch_potf2_64(s, Ajj, ld, col_offset + off, info)) != hipSuccess) return e;
        if (below > 0) return launch_trsm_rlt64(s, Ajj, ld, A + (off + IB) * ld + off, ld, below);
        return hipSuccess;
    }
    const int64_t h = (w / 2) / leaf * leaf;       // left width (multiple of the leaf width, >= one leaf)
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
    const bool fused = tuning().panel_fused && w % 128 == 0 && off % 128 == 0 && m % 128 == 0;
    const int64_t leaf = fused ? 128 : IB;
    if (w <= leaf) {
        if (fused) return launch_trsm128(s, L + off * ldl + off, ldl, X + off, ldx, m);
        return launch_trsm_rlt64(s, L + off * ldl + off, ldl, X + off, ldx, m);
    }
    const int64_t h = (w / 2) / leaf * leaf;
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
    return plan_block_widths(c->block(ncols), ncols, c->nb == 0, c->ramp);      // gpmi_plan.h
}

// In-place blocked right-looking Cholesky of the leading ncols x ncols block of
// A; rows ncols..nrows-1 are carried along (they end up multiplied by L^-T).
//
// With lookahead the trailing update of step k is split in two launches: (a) the next
// block column only and (b) the rest.  From 49152 columns up both run on the main stream, the panel
// stream (high priority) factors panel k+1 behind (a), concurrently with (b).  Below that (a) itself goes
// on the panel stream, right behind panel k and in front of panel k+1, (b) on the main stream.  (a) and (b) write different columns
// and both only read panel k, so they start together: the tail of (a) -- a launch whose last
// tiles leave most CUs idle -- and the whole latency-bound panel k+1 run under (b).
// Dependencies:
//   panel k  ->  (a)_k                 (same stream)
//   panel k  ->  (b)_k                 (main waits on the panel event)
//   (a)_k    ->  panel k+1             (same stream)
//   (b)_k    ->  (a)_{k+1}             (panel stream waits on the event behind (b)_k)
//   (b)_k    ->  (b)_{k+1}             (same stream)
hipError_t cholesky_inplace(gpmi_ctx* c, double* A, int64_t ld, int64_t ncols, int64_t nrows,
                            int64_t* info, bool account, int64_t carried_rows, const SweepFollower* follow) {
    hipError_t e;
    hipStream_t sm = c->stream;
    const std::vector<int64_t> widths = block_schedule(c, ncols);
    const int64_t NB = c->block(ncols);
    // below ~12k columns the two-stream choreography costs more than the panel it hides
    const bool la = c->lookahead && c->pstream && ncols > NB && ncols >= c->la_min;
    hipStream_t sp_ = la ? c->pstream : sm;
    // panel kernels beside the trailing update use their small-LDS forms -- while there IS a trailing update of some length
    // to run beside: once the columns right of the panel are fewer than c->shallow_min, part (b) of a step is over long
    // before the panel chain is, and the one-launch forms (which then find empty CUs) are the shorter chain
    GemmShallowScope shallow(la || follow != nullptr);
    // (a) on the panel stream pays at mid sizes (N = 16384: -4 %, 32768: -1 %), where a launch's tail and the
    // panel chain are a visible share of a step; at the headline size it is worth 0.4 % and would put two
    // trailing-update launches in flight at once, which makes "time per launch" (the roofline figure) ambiguous
    const bool col_on_panel = la && ncols < 49152;
    const int slot_p = account ? GPMI_T_CHOL_PANEL : GPMI_T_COUNT - 1;
    const int slot_t = account ? GPMI_T_CHOL_TRAIL : GPMI_T_COUNT - 1;
    // experiment (option potrf_server): the potrf128 of every leaf goes to ONE resident workgroup on a stream of its own
    // (panel_mfma.hip), so that it never waits for an empty CU beside the update; the scope stops the server on every
    // way out of this function
    struct ServerScope {
        gpmi_ctx* c; hipStream_t sp; bool on = false;
        ~ServerScope() { if (on) (void)potrf_server_stop(&c->pserver, sp); }
    } server{c, sp_};
    if (la && tuning().potrf_server && tuning().panel_fused) {
        if (!c->pmail.p) {
            if ((e = c->pmail.ensure(sizeof(PotrfMail))) != hipSuccess) return e;
            if ((e = hipMemset(c->pmail.p, 0, sizeof(PotrfMail))) != hipSuccess) return e;
            c->pserver.mail = c->pmail.as<PotrfMail>();
        }
        if (!c->sstream) {
            int lo = 0, hi = 0;
            (void)hipDeviceGetStreamPriorityRange(&lo, &hi);
            // ablation 128: the server's stream at the default priority
            if ((e = hipStreamCreateWithPriority(&c->sstream, hipStreamNonBlocking, (tuning().potrf_server & 128) ? lo : hi)) != hipSuccess) return e;
        }
        c->pserver.mode = tuning().potrf_server;
        if ((e = potrf_server_start(&c->pserver, c->sstream)) != hipSuccess) return e;
        server.on = true;
    }
    if (la && (e = c->order(sm, sp_)) != hipSuccess) return e;   // panel 0 after the K build
    // `counted`: the launch enters the roofline figures (GPMI_T_CHOL_TRAIL, its own kernel symbol).  Under lookahead
    // only (b) does: (a) runs at the same time on the other stream, so summing both durations would count that
    // time twice; (a) is launched under the generic symbol and timed with the panel it belongs to.
    auto trail = [&](hipStream_t st, bool counted, int64_t r0, int64_t c0, int64_t k, int64_t nb, int64_t ncol_upd) -> hipError_t {
        // C = A[r0.., c0..c0+ncol_upd) -= A[r0.., k..k+nb) * A[c0.., k..k+nb)^T, lower part
        GemmArgs g;
        g.C = A + r0 * ld + c0;
        g.A = A + r0 * ld + k;
        g.B = A + c0 * ld + k;
        g.ldc = g.lda = g.ldb = ld;
        g.M = nrows - r0; g.N = ncol_upd; g.K = nb;
        g.mode = 0; g.lower = 1; g.diag_off = r0 - c0;
        g.role = counted ? 1 : 0;
        // the roofline figures are those of the LDS-DMA kernel: the last, small updates that
        // run on the first-generation kernel are timed into the scratch slot
        const bool dma = gemm_nt_routes_dma(g);
        size_t sp = c->span_begin(counted ? (dma ? slot_t : GPMI_T_COUNT - 1) : slot_p, st);
        hipError_t er = launch_gemm_nt(st, g);
        c->span_end(sp, st);
        if (account && dma && counted) {
            c->stage_ms[GPMI_T_TRAIL_LAUNCHES] += 1.0;
            // algorithmic: the lower triangle of the real rows plus the one row that carries y (plus the test set's rows
            // when they ride along: all of them reach every column, so only their number counts)
            c->stage_ms[GPMI_T_TRAIL_FLOPS] += gemm_nt_algorithmic_flops(g, ncols - r0 + 1 + carried_rows);
        }
        return er;
    };
    int64_t k = 0;
    hipEvent_t ev_b = nullptr;                    // behind the last (b) on the main stream
    for (size_t step = 0; step < widths.size(); ++step) {
        const int64_t nb = widths[step];
        size_t sp = c->span_begin(slot_p, sp_);
        {
            GemmShallowScope panel_forms(la && ncols - k >= c->shallow_min, la, true);
            e = panel_factor(sp_, A + k * ld + k, ld, nb, nrows - k, k, info);
        }
        c->span_end(sp, sp_);
        if (e != hipSuccess) return e;
        if (la && (e = c->order(sp_, sm)) != hipSuccess) return e;
        const int64_t r0 = k + nb;
        if (follow) {
            // block column k of L is final: the following rows' solve against L_kk and their update by it, behind their own
            // step k-1 on their stream -- nothing of the factorisation waits for them
            if ((e = c->order(sp_, follow->vs)) != hipSuccess) return e;
            {
                // as in solve_sweep: the small-LDS solve forms only for enough rows to matter beside the updates
                GemmShallowScope solve_forms(follow->m >= 2048, true, true);
                if ((e = trsm_block(follow->vs, A + k * ld + k, ld, follow->V + k, follow->ldv, follow->m, nb)) != hipSuccess) return e;
            }
            if (r0 < ncols) {
                GemmArgs g;   // V[:, r0..) -= V[:, k..k+nb) * L[r0.., k..k+nb)^T
                g.C = follow->V + r0; g.A = follow->V + k; g.B = A + r0 * ld + k;
                g.ldc = g.lda = follow->ldv; g.ldb = ld;
                g.M = follow->m; g.N = ncols - r0; g.K = nb;
                g.mode = 0; g.lower = 0; g.diag_off = 0;
                if ((e = launch_gemm_nt(follow->vs, g)) != hipSuccess) return e;
            }
        }
        k = r0;
        if (r0 >= ncols) continue;
        if (!la) {
            if ((e = trail(sm, true, r0, r0, r0 - nb, nb, ncols - r0)) != hipSuccess) return e;
            continue;
        }
        const int64_t nbn = widths[step + 1];
        if (!col_on_panel) {
            // large problems: (a) then (b) on the main stream, panel k+1 behind (a)
            if ((e = trail(sm, true, r0, r0, r0 - nb, nb, nbn)) != hipSuccess) return e;
            if ((e = c->order(sm, sp_)) != hipSuccess) return e;
            if (r0 + nbn < ncols &&
                (e = trail(sm, true, r0 + nbn, r0 + nbn, r0 - nb, nb, ncols - r0 - nbn)) != hipSuccess) return e;
            continue;
        }
        // (a) next block column, behind panel k on the panel stream and behind (b) of the step before
        if (ev_b && (e = hipStreamWaitEvent(sp_, ev_b, 0)) != hipSuccess) return e;
        if ((e = trail(sp_, false, r0, r0, r0 - nb, nb, nbn)) != hipSuccess) return e;
        if (r0 + nbn < ncols) {                                                          // (b) rest
            if ((e = trail(sm, true, r0 + nbn, r0 + nbn, r0 - nb, nb, ncols - r0 - nbn)) != hipSuccess) return e;
            ev_b = c->new_event();
            if (!ev_b) return c->ev_error;
            if ((e = hipEventRecord(ev_b, sm)) != hipSuccess) return e;
        }
    }
    return hipSuccess;
}

void set_kernel_args(const gpmi_ctx* c, RbfArgs& r) {
    r.coef = c->coef; r.sig2 = c->sig2;
    r.kind = c->kind; r.kp0 = c->kp0; r.kp1 = c->kp1;
    for (int i = 0; i < 11; ++i) r.kpv[i] = c->kpv[i];
}

int ensure_train_buffers(gpmi_ctx* c, int64_t test_rows, bool test_cols) {
    c->Np = round_up(c->N, TILE);
    c->ldA = c->Np + (test_cols ? test_rows : 0) + c->ld_pad;
    c->Mp = c->Np + TILE + test_rows;
    c->yrow = test_cols ? c->Np + test_rows : c->Np;         // with their own columns the test rows come BEFORE the y rows
    c->post_in_A = false;
    HIP_TRY(c->A.ensure((size_t)c->Mp * c->ldA * sizeof(double)));
    HIP_TRY(c->info.ensure(sizeof(int64_t)));
    HIP_TRY(c->red.ensure(16 * sizeof(double)));
    return GPMI_OK;
}

// mean and variance (or standard deviation) from the row dots of v^T: h = [v_i . m for i < n_p | v_i . v_i for i < n_p]
void meanvar_to_host(gpmi_ctx* c, const std::vector<double>& h, double* mu, double* out2, int want_sd) {
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
}

// K build + Cholesky (+ forward solve through the y row) + LML on the stream
int factorize_impl(gpmi_ctx* c, double sigma, double ell, double noise_var, double* lml,
                   int64_t* bad_pivot, bool with_test, double* mu, double* out2, int want_sd, bool with_post, double jitter) {
    if (!c->have_train) return fail_arg("gpmi_factorize: no training set (call gpmi_set_train)");
    if (c->kind == 0 && (!(ell != 0.0) || std::isnan(ell) || std::isnan(sigma)))
        return fail_arg("gpmi_factorize: ell must be non-zero and hyper-parameters finite");
    if (std::isnan(noise_var)) return fail_arg("gpmi_factorize: noise_var is NaN");
    if (c->kind == 2 && c->d != 1) return fail_arg("gpmi_factorize: the periodic kernel is 1-D only (GP_regression.py:48)");
    // the test set's rows: inside the panel and update launches (1) or one block column behind on their own stream (2)
    int form = 0;
    if (with_test) {
        // measured (profiles/r04_one_pass_ab.txt, ms for two calls / ride / follow): N = 2048 1.69 / 1.32 / 1.49, 8192 10.5 / 8.94 /
        // 8.74, 16384 37.5 / 34.9 / 34.9, 32768 262.7 / 257.9 / 256.7, 65536 1702 / 1698 / 1710 -- the chip is work-conserving
        // either way, and riding costs no launches: the default at every size
        form = c->one_pass_form ? c->one_pass_form : 1;
        if (with_post) form = 3;                   // the augmented matrix: the rows ride (there is no other way) with columns of their own
        if (form == 2 && !c->vstream) {
            int lo = 0, hi = 0;
            (void)hipDeviceGetStreamPriorityRange(&lo, &hi);
            HIP_TRY(hipStreamCreateWithPriority(&c->vstream, hipStreamNonBlocking, lo));
        }
    }
    if (with_post && std::isnan(jitter)) return fail_arg("gpmi_fit_predict_sample: jitter is NaN");
    int rc = ensure_train_buffers(c, with_test ? c->np_ : 0, form == 3);
    if (rc) return rc;
    hipStream_t s = c->stream;
    c->have_factor = false;
    c->v_in_A = false;
    c->have_vinv = false;
    c->have_vside = false;
    c->have_v = false;
    c->timers_reset({GPMI_T_KBUILD, GPMI_T_CHOL, GPMI_T_CHOL_PANEL, GPMI_T_CHOL_TRAIL, GPMI_T_LML,
                     GPMI_T_TRAIL_LAUNCHES, GPMI_T_TRAIL_FLOPS});
    if (with_test) c->timers_reset({GPMI_T_KS, GPMI_T_SOLVE_V, GPMI_T_MEANVAR});
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
    c->span_end(sp);                  // GPMI_T_KBUILD is the kernel-matrix build (a1 + a2) alone
    // the augmented rows: y then zeros (a4 rides in the factorisation)
    const int64_t ncols = c->Np + (form == 3 ? c->np_ : 0);   // columns of the factorisation
    HIP_TRY(launch_fill_rows(s, c->m_row(), c->ldA, TILE, ncols, 0.0));
    HIP_TRY(launch_set_yrow(s, c->m_row(), c->y.as<double>(), c->N, c->Np));
    c->v_row0 = form == 3 ? c->Np : c->Np + TILE;
    double* Vr = A + c->v_row0 * c->ldA;
    if (with_test) {                  // K(X*, X) below the y rows (a1 for K_s, GP_regression.py:127): they leave as v^T
        sp = c->span_begin(GPMI_T_KS);
        RbfArgs t;
        t.A = c->Xs.as<double>(); t.B = c->X.as<double>();
        t.nA = c->n; t.nB = c->N; t.d = c->d; t.row0 = 0; t.nrows = c->np_; t.ncols = c->Np;
        set_kernel_args(c, t);
        t.diag_add = 0.; t.symmetric = 0;
        t.delta_square = (c->n == c->N) ? 1 : 0;
        t.max_sq = box_max_sq(c->boxXs, c->boxX);
        t.out = Vr; t.ld = c->ldA;
        HIP_TRY(launch_rbf(s, t));
        if (form == 3) {              // K_ss + jitter I in the rows' own columns, lower tiles (GP_regression.py:128, 154)
            RbfArgs q;
            q.A = q.B = c->Xs.as<double>();
            q.nA = q.nB = c->n; q.d = c->d; q.row0 = 0; q.nrows = c->np_; q.ncols = c->np_;
            set_kernel_args(c, q);
            q.diag_add = jitter; q.symmetric = 1; q.delta_square = 1;
            q.max_sq = box_max_sq(c->boxXs, c->boxXs);
            q.out = Vr + c->Np; q.ld = c->ldA;
            HIP_TRY(launch_rbf(s, q));
        }
        c->span_end(sp);
    }

    sp = c->span_begin(GPMI_T_CHOL);
    if (form == 2) {
        SweepFollower f;
        f.V = Vr; f.ldv = c->ldA; f.m = c->np_; f.vs = c->vstream;
        HIP_TRY(c->order(s, f.vs));                     // K(X*, X) is in place
        size_t spv = c->span_begin(GPMI_T_SOLVE_V, f.vs);
        HIP_TRY(cholesky_inplace(c, A, c->ldA, c->Np, c->Np + TILE, c->info.as<int64_t>(), true, 0, &f));
        c->span_end(spv, f.vs);
        HIP_TRY(c->order(f.vs, s));
    } else if (form == 3) {
        HIP_TRY(cholesky_inplace(c, A, c->ldA, ncols, c->Mp, c->info.as<int64_t>(), true, 0));
    } else {
        HIP_TRY(cholesky_inplace(c, A, c->ldA, c->Np, c->Mp, c->info.as<int64_t>(), true, with_test ? c->n : 0));
    }
    c->span_end(sp);

    sp = c->span_begin(GPMI_T_LML);
    HIP_TRY(launch_lml_reduce(s, A, c->ldA, c->m_row(), c->N, c->red.as<double>()));
    c->span_end(sp);

    std::vector<double> h;
    if (with_test) {                  // a6 + a8 off the finished rows: mean_i = v_i . m, sum of squares for the variance
        HIP_TRY(c->vec.ensure((size_t)std::max(c->Np, c->np_) * 4 * 8));
        sp = c->span_begin(GPMI_T_MEANVAR);
        double* dot = c->vec.as<double>();
        HIP_TRY(launch_row_dots(s, Vr, c->ldA, c->np_, c->Np, c->m_row(), dot, dot + c->np_));
        c->span_end(sp);
        h.resize(2 * (size_t)c->np_);
        HIP_TRY(hipMemcpyAsync(h.data(), dot, h.size() * 8, hipMemcpyDeviceToHost, s));
    }
    double red[2];
    int64_t info;
    HIP_TRY(hipMemcpyAsync(red, c->red.p, sizeof red, hipMemcpyDeviceToHost, s));
    HIP_TRY(hipMemcpyAsync(&info, c->info.p, sizeof info, hipMemcpyDeviceToHost, s));
    HIP_TRY(hipStreamSynchronize(s));
    c->timers_collect();
    if (c->pserver.mail) {                  // the resident potrf128 server (experiment): a wait that gave up is an error, not a result
        HIP_TRY(hipStreamSynchronize(c->pstream));
        if (c->sstream) HIP_TRY(hipStreamSynchronize(c->sstream));
        int perr = 0;
        HIP_TRY(hipMemcpy(&perr, &c->pserver.mail->err, sizeof(int), hipMemcpyDeviceToHost));
        if (perr) {
            (void)hipMemset(&c->pserver.mail->err, 0, sizeof(int));
            return fail_runtime(hipErrorUnknown, "gpmi_factorize: the resident potrf128 server did not answer (option potrf_server)");
        }
    }
    if (info != big && (info < c->N || (form == 3 && info >= c->Np && info < c->Np + c->n))) {
        // a pivot of K + sI (GP_regression.py:138) or, in the augmented form, of the posterior covariance (:154; counted from
        // its own first column, as gpmi_post_chol reports it)
        if (bad_pivot) *bad_pivot = info < c->N ? info + 1 : info - c->Np + 1;
        if (lml) *lml = std::numeric_limits<double>::quiet_NaN();
        g_err = "Matrix is not positive definite";
        return GPMI_ERR_NOT_PD;
    }
    if (bad_pivot) *bad_pivot = 0;
    // tune_hyperparms_regression.py:312, with y^T alpha = m^T m
    if (lml) *lml = -.5 * red[1] - red[0] - (double)c->N / 2.0 * std::log(2 * M_PI);
    c->have_factor = true;
    c->factor_fused = tuning().panel_fused;
    if (with_test) {
        c->v_in_A = true;
        c->ldV = c->ldA;
        c->have_v = true;
        ++c->v_gen;
        c->post_in_A = form == 3;
        c->post_jitter = jitter;
        meanvar_to_host(c, h, mu, out2, want_sd);
    }
    return GPMI_OK;
}

// v^T = K_s^T L^-T: right-looking sweep over the block columns of L, with the
// same lookahead split as the Cholesky (the triangular solve of block column
// k+1 overlaps the update of the columns beyond it).
// tri: V starts as the identity (m == Np), so at step k only rows < k + nb are non-zero in
// block column k -- the sweep then costs Np^3/3 and leaves the upper triangular L^-T.
hipError_t solve_sweep(gpmi_ctx* c, double* V, int64_t ldv, int64_t m, bool tri) {
    hipError_t e;
    hipStream_t sm = c->stream;
    const double* A = c->A.as<double>();
    const int64_t ld = c->ldA, Np = c->Np;
    const int64_t NB = c->block(Np);
    const bool la = c->lookahead && c->pstream && Np > NB && Np >= c->la_min;
    hipStream_t sp_ = la ? c->pstream : sm;
    // small-LDS panel forms beside the update only for sweeps with enough rows to keep the chip busy: for a few
    // hundred test points the chain of launches is what counts, and the one-launch trsm128 is shorter
    // (N = 16384, n = 1024: 7.96 against 8.47 ms)
    GemmShallowScope shallow(la && m >= 2048, la);
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

}  // namespace gpmi
