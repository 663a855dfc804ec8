// Squared-exponential kernel-matrix build (reference: RBF_kernel,
// GP_regression.py:8-19, and the "+ s * np.eye(N)" of :138 fused into the
// diagonal).
//
//   out[i][j] = sig2 * exp(coef * sum_k (a[i,k] - b[j,k])^2)      coef = -.5*(1/l^2)
//
// The per-element arithmetic follows the reference exactly: difference, square
// (one rounding each, no FMA contraction) and the sum over k in the order
// NumPy's add.reduce uses for the middle axis of the reference's (N,d,M)
// temporary (pairwise_sum: sequential for d < 8; eight interleaved partial sums
// r[k mod 8], the tree ((r0+r1)+(r2+r3))+((r4+r5)+(r6+r7)) and a sequential tail
// for d <= 128; recursive halving above -- checked bit-for-bit against NumPy
// 2.2.6 in tests/test_oracle_vs_golden.py), then one multiplication by coef,
// exp, one multiplication by sigma^2.  Only exp() itself may differ from
// NumPy's (both are < 1 ulp).
//
// HBM-write bound: a block produces a 128 x 128 tile; a wavefront writes whole
// 1-KiB row segments (64 lanes x 16 B).  The x rows of both tile edges are
// staged once in LDS; for the common d (1..8, 16) the b values of a thread's two
// columns live in registers and the a values are wave-wide LDS broadcasts.
#include <cmath>

#include "gpmi_internal.h"

namespace gpmi {

typedef double d2 __attribute__((ext_vector_type(2)));

constexpr int RT = 128;        // tile edge
constexpr int LDS_MAXD = 32;   // largest d staged in LDS; above: rbf_naive_kernel

struct RbfDev {
    const double* __restrict__ A;
    const double* __restrict__ B;
    int64_t nA, nB;
    int d;
    int64_t row0;
    int Tm, Tn;
    double coef, sig2, diag_add;
    int symmetric;
    int tri;               // triangular tile enumeration (symmetric, row0 == 0)
    double* __restrict__ out;
    int64_t ld;
    int kind;
    double kp0, kp1;
    double kpv[11];        // kind 3: theta_1 .. theta_11 of the CO2 composite kernel
    int delta_square;      // kind 3: the output is square, so kernel_4 adds theta_11^2 on row == col
    int64_t delta_col0;    // ... where "col" counts from this offset (B is a window of the column inputs: multi-rank predict)
    int nocheck;           // host-proved: every exp argument of this launch lies in [-700, 0]
    int strip;             // rbf_regs_kernel: row tiles per work item
    int nitems;            // rbf_regs_kernel: strips x column tiles
};

#pragma clang fp contract(off)

// sum_k (a(k) - b(k))^2 in NumPy's pairwise order, n <= 128 (see file header)
template <class FA, class FB>
__device__ __forceinline__ double sq_pw_small(FA a, FB b, int n) {
    if (n < 8) {
        double res = 0.0;
        for (int k = 0; k < n; ++k) { const double e = a(k) - b(k); res = res + e * e; }
        return res;
    }
    double r[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) { const double e = a(j) - b(j); r[j] = e * e; }
    int i = 8;
    for (; i < n - (n % 8); i += 8) {
#pragma unroll
        for (int j = 0; j < 8; ++j) { const double e = a(i + j) - b(i + j); r[j] = r[j] + e * e; }
    }
    double res = ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]));
    for (; i < n; ++i) { const double e = a(i) - b(i); res = res + e * e; }
    return res;
}

// exp(x) for the kernel's arguments (x <= 0): Cody-Waite reduction x = n*ln2 + r with the
// round-to-nearest n taken from the low bits of x*log2(e) + 1.5*2^52, a degree-11 minimax
// polynomial in r (|r| <= ln2/2; 1 + r + r^2 P(r) fitted for relative error by
// scripts/exp_poly_fit.py: 3.6e-18, 1.1e-17 with the coefficients rounded to double), and 2^n
// applied by adding n to the exponent field -- full-rate fp64 FMAs and one integer op, no
// v_rndne/v_cvt/v_ldexp.  The two leading coefficients are exactly 1, so exp(0) == 1 and the
// diagonal of K is sigma^2 exactly, as in NumPy.
// Valid while the result is a normal number; the caller falls back to the library exp
// for the whole wave if any lane is outside [-700, 0] (or NaN).  Error < 1 ulp (checked
// against NumPy at the 3-ulp parity tolerance of the K tests).
__device__ __forceinline__ double exp_neg_fast(double x) {
    const double MAGIC = 6755399441055744.0;                    // 1.5 * 2^52
    const double t = fma(x, 1.4426950408889634074, MAGIC);
    const double n = t - MAGIC;
    double r = fma(-n, 6.93147180369123816490e-01, x);          // ln2_hi (low bits zero: exact)
    r = fma(-n, 1.90821492927058770002e-10, r);                 // ln2_lo
    double q = 0x1.ad7f3c1cdbf13p-26;                           // c11
    q = fma(q, r, 0x1.28ad9b87c947cp-22);                       // c10
    q = fma(q, r, 0x1.71df25b4b9501p-19);                       // c9
    q = fma(q, r, 0x1.a01999e260c97p-16);                       // c8
    q = fma(q, r, 0x1.a01a012a0e822p-13);                       // c7
    q = fma(q, r, 0x1.6c16c18438b14p-10);                       // c6
    q = fma(q, r, 0x1.1111111127d10p-7);                        // c5
    q = fma(q, r, 0x1.555555555083ep-5);                        // c4
    q = fma(q, r, 0x1.55555555554f9p-3);                        // c3
    q = fma(q, r, 0x1.000000000000ap-1);                        // c2
    q = fma(q, r, 1.0);
    q = fma(q, r, 1.0);
    const int ni = __double2loint(t);                           // n in the low word of t (two's complement)
    return __hiloint2double(__double2hiint(q) + (ni << 20), __double2loint(q));
}

// exp for a whole wavefront: fast path unless some lane leaves its domain
__device__ __forceinline__ void exp_pair(double x0, double x1, double& e0, double& e1) {
    // x = coef * (sum of squares) with coef < 0 is never positive; NaN fails the compare
    const bool ok = (x0 >= -700.0) & (x1 >= -700.0);
    if (__builtin_amdgcn_ballot_w64(!ok) == 0) {                // compare mask straight into SGPRs
        e0 = exp_neg_fast(x0);
        e1 = exp_neg_fast(x1);
    } else {
        e0 = exp(x0);
        e1 = exp(x1);
    }
}

// exp(x) for x <= 0 with a 4096-entry table of 2^(j/4096) in LDS (32 KiB per workgroup, filled from a
// device-resident copy at kernel start; the host computes the entries in extended precision).
//   x = (4096 k + j) ln2 / 4096 + r,  |r| <= ln2 / 8192:   exp(x) = 2^k * T[j] * (1 + r + r^2/2 + r^3/6)
// (the cubic's truncation error is r^4 / 24 < 2.2e-18 relative; a 2048-entry table leaves 0.3 ulp there).  13 vector instructions against the 17 of
// the polynomial-only form above (the K build is bound by its instruction count):
// one multiply-add for k and j at once, a two-term Cody-Waite reduction, two instructions each for the table
// address and for the exponent, three for the cubic, one for T + T q.  T[0] == 1 and r == 0 at x == 0, so
// exp(0) == 1 exactly and the diagonal of K stays sigma^2.  Error: 0.5 ulp of the table entry + 0.5 ulp of the
// final fused multiply-add (measured: scripts/exp_accuracy.py).  Same domain as exp_neg_fast.
constexpr int EXP_TAB = 4096;
constexpr int EXP_TAB_LOG = 12;
__device__ double g_exp2_tab[EXP_TAB];

__device__ __forceinline__ double exp_neg_tab(double x, const double* __restrict__ tab) {
    const double MAGIC = 6755399441055744.0 / (double)EXP_TAB;        // 1.5 * 2^52 / 4096: one unit in the last place = 1/4096
    const double t = fma(x, 1.4426950408889634074, MAGIC);            // low word: round(4096 x log2 e) = 4096 k + j
    const double n = t - MAGIC;                                       // (4096 k + j) / 4096
    double r = fma(-n, 6.93147180369123816490e-01, x);                // ln2_hi (trailing zero bits: exact product while |x| < 700)
    r = fma(-n, 1.90821492927058770002e-10, r);                       // ln2_lo
    const int m = __double2loint(t);
    const double T = tab[m & (EXP_TAB - 1)];
    double q = fma(r, 1.0 / 6.0, 0.5);
    q = fma(q, r, 1.0);
    q = q * r;                                                        // r + r^2/2 + r^3/6
    const double e = fma(T, q, T);
    // 2^k into the exponent field: ((m & ~4095) << 8) + hi(e) as v_and + v_lshl_add_u32 -- left to itself the compiler
    // canonicalises to shift, mask, add (three instructions; the K build is bound by its instruction count)
    static_assert(20 - EXP_TAB_LOG == 8, "shift below is written out");
    unsigned hi;
    asm("v_lshl_add_u32 %0, %1, 8, %2" : "=v"(hi) : "v"((unsigned)m & ~(unsigned)(EXP_TAB - 1)), "v"((unsigned)__double2hiint(e)));
    return __hiloint2double((int)hi, __double2loint(e));
}

__device__ __forceinline__ void exp_pair_tab(double x0, double x1, double& e0, double& e1, const double* tab) {
    const bool ok = (x0 >= -700.0) & (x1 >= -700.0);
    if (__builtin_amdgcn_ballot_w64(!ok) == 0) {
        e0 = exp_neg_tab(x0, tab);
        e1 = exp_neg_tab(x1, tab);
    } else {
        e0 = exp(x0);
        e1 = exp(x1);
    }
}

template <int D, class FA, class FB>
__device__ __forceinline__ double sq_pw_static(FA a, FB b) {
    if constexpr (D < 8) {
        double res = 0.0;
#pragma unroll
        for (int k = 0; k < D; ++k) { const double e = a(k) - b(k); res = res + e * e; }
        return res;
    } else {
        double r[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) { const double e = a(j) - b(j); r[j] = e * e; }
        constexpr int NB = D - (D % 8);
#pragma unroll
        for (int i = 8; i < NB; i += 8) {
#pragma unroll
            for (int j = 0; j < 8; ++j) { const double e = a(i + j) - b(i + j); r[j] = r[j] + e * e; }
        }
        double res = ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]));
#pragma unroll
        for (int i = NB; i < D; ++i) { const double e = a(i) - b(i); res = res + e * e; }
        return res;
    }
}

__device__ double sq_pw_global(const double* a, const double* b, int n) {
    if (n <= 128) return sq_pw_small([&](int k) { return a[k]; }, [&](int k) { return b[k]; }, n);
    int n2 = n / 2;
    n2 -= n2 % 8;
    return sq_pw_global(a, b, n2) + sq_pw_global(a + n2, b + n2, n - n2);
}

__device__ __forceinline__ bool rbf_map_tile(const RbfDev& p, int& ti, int& tj) {
    if (p.tri) {
        const int s = blockIdx.x;
        ti = (int)((sqrtf(8.f * (float)s + 1.f) - 1.f) * 0.5f);
        while ((ti + 1) * (ti + 2) / 2 <= s) ++ti;
        while (ti * (ti + 1) / 2 > s) --ti;
        tj = s - ti * (ti + 1) / 2;
    } else {
        ti = blockIdx.x / p.Tn;
        tj = blockIdx.x - ti * p.Tn;
    }
    // symmetric build: skip tiles above the diagonal
    return !(p.symmetric && (int64_t)tj * RT > p.row0 + (int64_t)ti * RT + RT - 1);
}

__device__ __forceinline__ void cov_store(const RbfDev& p, int64_t gr, int64_t gc, double v0, double v1,
                                          double* dst);

__device__ __forceinline__ void rbf_finish(const RbfDev& p, int64_t gr, int64_t gc, double s0, double s1,
                                           double* dst) {
    double e0, e1;
    exp_pair(p.coef * s0, p.coef * s1, e0, e1);
    cov_store(p, gr, gc, p.sig2 * e0, p.sig2 * e1, dst);
}

// diagonal term, identity / zero padding, 16-byte store of a column pair
__device__ __forceinline__ void cov_store(const RbfDev& p, int64_t gr, int64_t gc, double v0, double v1,
                                          double* dst) {
    if (p.symmetric) {
        if (gr == gc) v0 = v0 + p.diag_add;
        if (gr == gc + 1) v1 = v1 + p.diag_add;
        // identity padding beyond the real matrix
        if (gr >= p.nA || gc >= p.nB) v0 = (gr == gc) ? 1.0 : 0.0;
        if (gr >= p.nA || gc + 1 >= p.nB) v1 = (gr == gc + 1) ? 1.0 : 0.0;
    } else {
        if (gr >= p.nA || gc >= p.nB) v0 = 0.0;
        if (gr >= p.nA || gc + 1 >= p.nB) v1 = 0.0;
    }
    *reinterpret_cast<d2*>(dst) = d2{v0, v1};
}

template <class FA, class FB>
__device__ __forceinline__ double cov_other_acc(const RbfDev& p, FA a, FB b, bool diag) {
    if (p.kind == 1) {                         // lin_kernel: np.dot(a - c, b.T - c)
        double s = 0.0;
        for (int k = 0; k < p.d; ++k) s = s + (a(k) - p.kp0) * (b(k) - p.kp0);
        return s;
    }
    if (p.kind == 3) {
        const double* th = p.kpv;              // th[i] = theta_(i+1)
        const double sq = sq_pw_small(a, b, p.d);                                    // CO2_example.py:76 / :83 (d <= 128)
        const double r = sqrt(sq);                                                   // :77 / :85
        const double k1 = (th[0] * th[0]) * exp(-.5 * sq / (th[1] * th[1]));         // :17
        const double first = -.5 * sq / (th[3] * th[3]);                             // :30
        const double sn = sin(M_PI * r) / th[4];
        const double second = -2 * (sn * sn);                                        // :31
        const double k2 = (th[2] * th[2]) * exp(first + second);                     // :32
        const double item = 1 + .5 * sq / (th[7] * (th[6] * th[6]));                 // :44
        const double k3 = (th[5] * th[5]) * (1.0 / pow(item, th[7]));                // :45-46
        const double delta = (p.delta_square && diag) ? 1.0 : 0.0;                   // :58-62
        const double k4 = (th[8] * th[8]) * exp(-.5 * sq / (th[9] * th[9])) + (th[10] * th[10]) * delta;   // :63-64
        return ((k1 + k2) + k3) + k4;                                                // :86-89
    }
    // per_kernel (1-D): exp(-2 * sin(pi * |a-b| / p)**2 / l**2), evaluated in the reference's order
    const double t = fabs(a(0) - b(0));
    const double sn = sin(M_PI * t / p.kp0);
    return exp(-2.0 * (sn * sn) / (p.kp1 * p.kp1));
}

__device__ __forceinline__ double cov_other(const RbfDev& p, const double* a, const double* b, bool diag) {
    if (p.kind == 3 && p.d > 128) {            // recursive-halving order above 128 terms
        const double* th = p.kpv;
        const double sq = sq_pw_global(a, b, p.d);
        const double r = sqrt(sq);
        const double k1 = (th[0] * th[0]) * exp(-.5 * sq / (th[1] * th[1]));
        const double first = -.5 * sq / (th[3] * th[3]);
        const double sn = sin(M_PI * r) / th[4];
        const double second = -2 * (sn * sn);
        const double k2 = (th[2] * th[2]) * exp(first + second);
        const double item = 1 + .5 * sq / (th[7] * (th[6] * th[6]));
        const double k3 = (th[5] * th[5]) * (1.0 / pow(item, th[7]));
        const double delta = (p.delta_square && diag) ? 1.0 : 0.0;
        const double k4 = (th[8] * th[8]) * exp(-.5 * sq / (th[9] * th[9])) + (th[10] * th[10]) * delta;
        return ((k1 + k2) + k3) + k4;
    }
    return cov_other_acc(p, [&](int k) { return a[k]; }, [&](int k) { return b[k]; }, diag);
}

// D > 0: d == D at compile time (b columns in registers); D == 0: runtime d <= LDS_MAXD
template <int D>
__global__ __launch_bounds__(256) void rbf_kernel(const RbfDev p) {
    extern __shared__ __attribute__((aligned(16))) double lds[];
    const int d = D ? D : p.d;
    double* As = lds;               // [RT][d]   row-major: broadcast reads
    double* Bs = lds + RT * d;      // [d][RT]   k-major: conflict-free column reads
    int ti, tj;
    if (!rbf_map_tile(p, ti, tj)) return;
    const int64_t grow0 = p.row0 + (int64_t)ti * RT;
    const int64_t gcol0 = (int64_t)tj * RT;
    const int tid = threadIdx.x;
    for (int e = tid; e < RT * d; e += 256) {
        const int r = e / d, k = e - r * d;
        const int64_t ga = grow0 + r, gb = gcol0 + r;
        As[r * d + k] = (ga < p.nA) ? p.A[ga * d + k] : 0.0;
        Bs[k * RT + r] = (gb < p.nB) ? p.B[gb * d + k] : 0.0;
    }
    __syncthreads();
    const int cp = tid & 63;       // column pair: cols 2cp, 2cp+1
    const int rg = tid >> 6;       // row group: rows 32*rg .. +32
    const int64_t gc = gcol0 + 2 * cp;
    double* out0 = p.out + ((int64_t)ti * RT + 32 * rg) * p.ld + gc;
    if constexpr (D > 0) {
        double b0[D], b1[D];
#pragma unroll
        for (int k = 0; k < D; ++k) {
            const d2 bv = *reinterpret_cast<const d2*>(&Bs[k * RT + 2 * cp]);
            b0[k] = bv.x;
            b1[k] = bv.y;
        }
#pragma unroll 2
        for (int r = 0; r < 32; ++r) {
            const double* ar = &As[(32 * rg + r) * D];
            double av[D];
#pragma unroll
            for (int k = 0; k < D; ++k) av[k] = ar[k];
            const double s0 = sq_pw_static<D>([&](int k) { return av[k]; }, [&](int k) { return b0[k]; });
            const double s1 = sq_pw_static<D>([&](int k) { return av[k]; }, [&](int k) { return b1[k]; });
            rbf_finish(p, grow0 + 32 * rg + r, gc, s0, s1, out0 + (int64_t)r * p.ld);
        }
    } else {
        if (p.kind != 0) {
            // the reference's other covariance functions through the same tile: X rows of both tile edges in LDS
            // (a: broadcast reads, b: conflict-free column reads), one column pair x 32 rows per thread
            for (int r = 0; r < 32; ++r) {
                const int64_t gr = grow0 + 32 * rg + r;
                const double* ar = &As[(32 * rg + r) * d];
                const double* bc = &Bs[2 * cp];
                double v0 = 0., v1 = 0.;
                if (gr < p.nA) {
                    if (gc < p.nB) v0 = cov_other_acc(p, [&](int k) { return ar[k]; }, [&](int k) { return bc[k * RT]; }, gr == gc + p.delta_col0);
                    if (gc + 1 < p.nB) v1 = cov_other_acc(p, [&](int k) { return ar[k]; }, [&](int k) { return bc[k * RT + 1]; }, gr == gc + 1 + p.delta_col0);
                }
                cov_store(p, gr, gc, v0, v1, out0 + (int64_t)r * p.ld);
            }
            return;
        }
        for (int r = 0; r < 32; ++r) {
            const double* ar = &As[(32 * rg + r) * d];
            const double* bc = &Bs[2 * cp];
            const double s0 = sq_pw_small([&](int k) { return ar[k]; }, [&](int k) { return bc[k * RT]; }, d);
            const double s1 = sq_pw_small([&](int k) { return ar[k]; }, [&](int k) { return bc[k * RT + 1]; }, d);
            rbf_finish(p, grow0 + 32 * rg + r, gc, s0, s1, out0 + (int64_t)r * p.ld);
        }
    }
}

// Fast path for compile-time d (1..8, 16): no LDS, no barrier.  The two b columns of a
// thread are 2*D contiguous doubles of X (coalesced 16-byte loads), the a row of a wave
// is wave-uniform and comes through the scalar cache (s_load -> SGPR operands of the
// VALU).  A block owns one 128-column tile and walks down a strip of row tiles with its b
// columns resident in registers; blocks adjacent in blockIdx.x write adjacent 1-KiB
// segments of the same rows.  Interior tiles (all rows and columns inside the matrix, not
// on the diagonal) skip every padding / diagonal test.
template <int D>
__device__ __forceinline__ void rbf_load_cols(const RbfDev& p, const double* __restrict__ Bp, int64_t gc,
                                              double (&b0)[D], double (&b1)[D]) {
    if (gc + 1 < p.nB) {
        const double* bp = Bp + gc * D;
        d2 t[D];
#pragma unroll
        for (int k = 0; k < D; ++k) t[k] = *reinterpret_cast<const d2*>(bp + 2 * k);
#pragma unroll
        for (int k = 0; k < 2 * D; ++k) {
            const double v = (k & 1) ? t[k >> 1].y : t[k >> 1].x;
            if (k < D) b0[k] = v; else b1[k - D] = v;
        }
    } else {
#pragma unroll
        for (int k = 0; k < D; ++k) {
            b0[k] = (gc < p.nB) ? Bp[gc * D + k] : 0.0;
            b1[k] = 0.0;
        }
    }
    // pin the columns here so the wait for their loads sits before the row loops; a wait inside
    // a loop would also wait for the inline-asm stores, which the compiler does not count
#pragma unroll
    for (int k = 0; k < D; ++k) {
        asm volatile("" : "+v"(b0[k]));
        asm volatile("" : "+v"(b1[k]));
    }
}

template <int D, bool EDGE, bool CHECK = true, bool UNIT = false>
__device__ __forceinline__ void rbf_tile_rows(const RbfDev& p, const double* __restrict__ Ap,
                                              double* __restrict__ outp, int64_t grow0, int64_t gcol0, int ti,
                                              const double (&b0)[D], const double (&b1)[D], const double* tab) {
    const int cp = threadIdx.x & 63;
    const int rg = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    if constexpr (EDGE) {
        const int64_t gc = gcol0 + 2 * cp;
        double* out0 = outp + ((int64_t)ti * RT + 32 * rg) * p.ld + gc;
#pragma unroll 2
        for (int r = 0; r < 32; ++r) {
            const int64_t gr = grow0 + 32 * rg + r;                 // wave-uniform
            const double* ar = Ap + ((gr < p.nA) ? gr : 0) * D;
            double av[D];
#pragma unroll
            for (int k = 0; k < D; ++k) av[k] = ar[k];               // scalar loads (uniform address)
            const double s0 = sq_pw_static<D>([&](int k) { return av[k]; }, [&](int k) { return b0[k]; });
            const double s1 = sq_pw_static<D>([&](int k) { return av[k]; }, [&](int k) { return b1[k]; });
            double e0, e1;
            exp_pair_tab(p.coef * s0, p.coef * s1, e0, e1, tab);
            cov_store(p, gr, gc, p.sig2 * e0, p.sig2 * e1, out0 + (int64_t)r * p.ld);
        }
    } else {
        // wave-uniform row base + constant 32-bit lane offset: the store takes the scalar-base form
        double* urow = outp + ((int64_t)ti * RT + 32 * rg) * p.ld + gcol0;
        const unsigned lane_off = 16u * (unsigned)cp;             // bytes
        const double* arow = Ap + (grow0 + 32 * rg) * D;
#pragma unroll 2
        for (int r = 0; r < 32; ++r) {
            const double* ar = arow + r * D;
            double av[D];
#pragma unroll
            for (int k = 0; k < D; ++k) av[k] = ar[k];               // scalar loads (uniform address)
            const double s0 = sq_pw_static<D>([&](int k) { return av[k]; }, [&](int k) { return b0[k]; });
            const double s1 = sq_pw_static<D>([&](int k) { return av[k]; }, [&](int k) { return b1[k]; });
            double e0, e1;
            if constexpr (CHECK) {
                exp_pair_tab(p.coef * s0, p.coef * s1, e0, e1, tab);
            } else {
                e0 = exp_neg_tab(p.coef * s0, tab);
                e1 = exp_neg_tab(p.coef * s1, tab);
            }
            if constexpr (!UNIT) { e0 = p.sig2 * e0; e1 = p.sig2 * e1; }   // sigma^2 == 1: the product is exact
            // global_store with SGPR base: the per-row address costs no vector instruction
            const d2 ev = d2{e0, e1};
            asm volatile("global_store_dwordx4 %0, %1, %2" ::"v"(lane_off), "v"(ev), "s"(urow + (int64_t)r * p.ld) : "memory");
        }
    }
}

// Work item = (strip of p.strip row tiles, column tile), column tile fastest.  Blocks are
// persistent: block b takes items b, b + gridDim.x, ... (a bare store stream in this pattern
// runs 10 % faster from persistent blocks than from one short block per item).
template <int D>
__global__ __launch_bounds__(256) void rbf_regs_kernel(const double* __restrict__ Ap, const double* __restrict__ Bp,
                                                        double* __restrict__ outp, const RbfDev p) {
    __shared__ __attribute__((aligned(16))) double tab[EXP_TAB];
    for (int i = 2 * threadIdx.x; i < EXP_TAB; i += 512)
        *reinterpret_cast<d2*>(tab + i) = *reinterpret_cast<const d2*>(g_exp2_tab + i);
    __syncthreads();
    const bool unit = p.sig2 == 1.0;
    const int row_tile0 = (int)(p.row0 / RT);
    for (int item = blockIdx.x; item < p.nitems; item += gridDim.x) {
        const int sy = item / p.Tn;
        const int tj = item - sy * p.Tn;
        int ta = sy * p.strip;
        const int tb = min(ta + p.strip, p.Tm);
        // symmetric build: only tiles that reach the diagonal or lie below it (row0 is a tile multiple)
        if (p.symmetric) ta = max(ta, tj - row_tile0);
        if (ta >= tb) continue;
        const int64_t gcol0 = (int64_t)tj * RT;
        double b0[D], b1[D];
        rbf_load_cols<D>(p, Bp, gcol0 + 2 * (threadIdx.x & 63), b0, b1);
        for (int ti = ta; ti < tb; ++ti) {
            const int64_t grow0 = p.row0 + (int64_t)ti * RT;
            const bool inside = grow0 + RT <= p.nA && gcol0 + RT <= p.nB;
            const bool on_diag = p.symmetric && gcol0 + RT > grow0;      // touches global row == col
            if (inside && !on_diag) {
                // launch-uniform specialisations of the interior loop: no per-wave domain test when the
                // host has bounded the arguments, no sigma^2 multiply when it is 1 (the reference's default)
                if (p.nocheck) {
                    if (unit) rbf_tile_rows<D, false, false, true>(p, Ap, outp, grow0, gcol0, ti, b0, b1, tab);
                    else rbf_tile_rows<D, false, false, false>(p, Ap, outp, grow0, gcol0, ti, b0, b1, tab);
                } else {
                    if (unit) rbf_tile_rows<D, false, true, true>(p, Ap, outp, grow0, gcol0, ti, b0, b1, tab);
                    else rbf_tile_rows<D, false, true, false>(p, Ap, outp, grow0, gcol0, ti, b0, b1, tab);
                }
            } else {
                rbf_tile_rows<D, true>(p, Ap, outp, grow0, gcol0, ti, b0, b1, tab);
            }
        }
    }
}

// any d: operands straight from global memory (L2-resident), one column pair per thread
__global__ __launch_bounds__(256) void rbf_naive_kernel(const RbfDev p) {
    int ti, tj;
    if (!rbf_map_tile(p, ti, tj)) return;
    const int64_t grow0 = p.row0 + (int64_t)ti * RT;
    const int64_t gcol0 = (int64_t)tj * RT;
    const int cp = threadIdx.x & 63, rg = threadIdx.x >> 6;
    const int64_t gc = gcol0 + 2 * cp;
    const int d = p.d;
    for (int r = 0; r < 32; ++r) {
        const int64_t gr = grow0 + 32 * rg + r;
        double s0 = 0., s1 = 0.;
        if (gr < p.nA) {
            if (gc < p.nB) s0 = sq_pw_global(p.A + gr * d, p.B + gc * d, d);
            if (gc + 1 < p.nB) s1 = sq_pw_global(p.A + gr * d, p.B + (gc + 1) * d, d);
        }
        rbf_finish(p, gr, gc, s0, s1, p.out + ((int64_t)ti * RT + 32 * rg + r) * p.ld + gc);
    }
}

// The reference's other covariance functions.  Up to d = 32 they run through
// the LDS-staged tile kernel above (rbf_kernel<0>, kind != 0); cov_other_kernel below is the any-d fallback,
// one column pair per thread straight from global memory.
//   kind 1  lin_kernel                                   GP_regression.py:22-33
//   kind 2  per_kernel (1-D)                             GP_regression.py:36-50
//   kind 3  covariance_function = kernel_1 + kernel_2 + kernel_3 + kernel_4 (RBF + decaying
//           periodic + rational quadratic + noise), any d             CO2_example.py:9-94
__global__ __launch_bounds__(256) void cov_other_kernel(const RbfDev p) {
    int ti, tj;
    if (!rbf_map_tile(p, ti, tj)) return;
    const int64_t grow0 = p.row0 + (int64_t)ti * RT;
    const int64_t gcol0 = (int64_t)tj * RT;
    const int cp = threadIdx.x & 63, rg = threadIdx.x >> 6;
    const int64_t gc = gcol0 + 2 * cp;
    const int d = p.d;
    for (int r = 0; r < 32; ++r) {
        const int64_t gr = grow0 + 32 * rg + r;
        double v0 = 0., v1 = 0.;
        if (gr < p.nA) {
            if (gc < p.nB) v0 = cov_other(p, p.A + gr * d, p.B + gc * d, gr == gc + p.delta_col0);
            if (gc + 1 < p.nB) v1 = cov_other(p, p.A + gr * d, p.B + (gc + 1) * d, gr == gc + 1 + p.delta_col0);
        }
        cov_store(p, gr, gc, v0, v1, p.out + ((int64_t)ti * RT + 32 * rg + r) * p.ld + gc);
    }
}

// 2^(j / EXP_TAB), j = 0 .. EXP_TAB - 1 (EXP_TAB = 4096), rounded to double from an extended-precision evaluation,
// uploaded once per device
static hipError_t exp_table_ready() {
    static PerDeviceOnce once;
    return once.run([]() -> hipError_t {
        static double h[EXP_TAB];
        for (int j = 0; j < EXP_TAB; ++j) h[j] = (double)exp2l((long double)j / (long double)EXP_TAB);
        return hipMemcpyToSymbol(HIP_SYMBOL(g_exp2_tab), h, sizeof h, 0, hipMemcpyHostToDevice);
    });
}

hipError_t launch_rbf(hipStream_t s, const RbfArgs& a) {
    if (a.nrows <= 0 || a.ncols <= 0) return hipSuccess;
    if (a.nrows % RT || a.ncols % RT || a.d <= 0) return hipErrorInvalidValue;
    if (a.symmetric && a.row0 % RT) return hipErrorInvalidValue;
    RbfDev p;
    p.A = a.A; p.B = a.B; p.nA = a.nA; p.nB = a.nB; p.d = (int)a.d; p.row0 = a.row0;
    p.Tm = (int)(a.nrows / RT); p.Tn = (int)(a.ncols / RT);
    p.coef = a.coef; p.sig2 = a.sig2; p.diag_add = a.diag_add; p.symmetric = a.symmetric;
    p.out = a.out; p.ld = a.ld;
    p.kind = a.kind; p.kp0 = a.kp0; p.kp1 = a.kp1;
    for (int i = 0; i < 11; ++i) p.kpv[i] = a.kpv[i];
    p.delta_square = a.delta_square;
    p.delta_col0 = a.delta_col0;
    // coef <= 0 and max_sq bounds every squared distance of this launch (NaN / unknown fail the test)
    p.nocheck = (a.max_sq >= 0.0 && a.coef <= 0.0 && a.coef * a.max_sq * 1.000001 >= -690.0) ? 1 : 0;
    if (a.kind < 0 || a.kind > 3 || (a.kind == 2 && a.d != 1)) return hipErrorInvalidValue;
    p.tri = (a.symmetric && a.row0 == 0 && p.Tm == p.Tn) ? 1 : 0;
    const int64_t nblk = p.tri ? (int64_t)p.Tm * (p.Tm + 1) / 2 : (int64_t)p.Tm * p.Tn;
    dim3 grid((unsigned)nblk), block(256);
    const size_t lds = (size_t)2 * RT * a.d * sizeof(double);
    if (a.kind != 0) {
        // lin / per / CO2 composite: the LDS-staged tile kernel (d <= 32), straight from global memory above that
        if (a.d <= LDS_MAXD) hipLaunchKernelGGL(rbf_kernel<0>, grid, block, lds, s, p);
        else hipLaunchKernelGGL(cov_other_kernel, grid, block, 0, s, p);
        return hipGetLastError();
    }
    {
        const hipError_t et = exp_table_ready();
        if (et != hipSuccess) return et;
    }
    // big builds: 4 row tiles per block (b columns loaded once, 4x fewer block launches)
    p.strip = (nblk >= 4096) ? 4 : 1;
    p.nitems = p.Tn * ((p.Tm + p.strip - 1) / p.strip);
    const dim3 sgrid((unsigned)std::min<int64_t>(p.nitems, tuning().rbf_blocks));
#define RBF_CASE(DD) case DD: hipLaunchKernelGGL(rbf_regs_kernel<DD>, sgrid, block, 0, s, p.A, p.B, p.out, p); break
    if (a.d > LDS_MAXD) {
        hipLaunchKernelGGL(rbf_naive_kernel, grid, block, 0, s, p);
    } else {
        switch (a.d) {
            RBF_CASE(1); RBF_CASE(2); RBF_CASE(3); RBF_CASE(4); RBF_CASE(5); RBF_CASE(6); RBF_CASE(7);
            RBF_CASE(8); RBF_CASE(16);
            default: hipLaunchKernelGGL(rbf_kernel<0>, grid, block, lds, s, p); break;
        }
    }
#undef RBF_CASE
    return hipGetLastError();
}

}  // namespace gpmi
