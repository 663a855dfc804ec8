// Panel kernels of the blocked Cholesky: the 64 x 64 diagonal-block
// factorisation and the triangular solve of the rows below it.
//
// Both run the textbook recurrences fully unrolled with the matrix in registers
// (every array index a compile-time constant, no scratch).  They replace the inner loops of LAPACK
// dpotrf / the LU-based np.linalg.solve the reference calls at
// GP_regression.py:138-139; their share of the flops is O(N^2 * 64), the
// O(N^3) part runs in gemm_nt.hip.
#include "gpmi_internal.h"

namespace gpmi {

typedef double d2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ double readlane_f64(double v, int lane) {
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_readlane(lo, lane);
    hi = __builtin_amdgcn_readlane(hi, lane);
    return __hiloint2double(hi, lo);
}

// ---------------------------------------------------------------------------
// potf2_64: right-looking unblocked Cholesky of a 64 x 64 block by one workgroup
// of 256 threads, ONE barrier per column.
// Thread (ty, tx) = (tid >> 4, tid & 15) keeps the 4 x 4 elements
// A[ty + 16 i][tx + 16 jj] in registers.  Step j: every thread reads the current
// (unscaled) column j from LDS -- c_r = A[r][j] -- and the pivot p = c_j, and
// applies   A[r][c] -= c_r * c_c / p   (= l_r * l_c with l = c / sqrt(p));
// the owners of column j+1 then publish their updated column for the next step.
// 1/p is a v_rcp_f64 refined by two Newton steps, evaluated redundantly by all
// threads (no second barrier for a broadcast).  The division by sqrt(p) is
// deferred: the columns stay unscaled through the sweep and one final pass applies
// 1/sqrt(p_c) (v_rsq_f64 + coupled Newton steps, 64 in parallel).  A non-positive (or NaN) pivot records col_offset + j in *info
// (atomic min) and poisons the block with NaN.
// ---------------------------------------------------------------------------
__device__ __forceinline__ void sqrt_and_rsqrt(double p, double& s, double& rinv) {
    const double y = __builtin_amdgcn_rsq(p);   // ~2^-26 relative
    double g = p * y, h = 0.5 * y;
    double r = fma(-h, g, 0.5);
    g = fma(g, r, g);
    h = fma(h, r, h);
    r = fma(-h, g, 0.5);
    g = fma(g, r, g);
    h = fma(h, r, h);
    const double d = fma(-g, g, p);             // residual p - g^2
    s = fma(d, h, g);
    rinv = h + h;
}

__device__ __forceinline__ double fast_rcp(double p) {
    double y = __builtin_amdgcn_rcp(p);
    double e = fma(-p, y, 1.0);
    y = fma(y, e, y);
    e = fma(-p, y, 1.0);
    return fma(y, e, y);
}

template <int J>
struct Potf2Step {
    static __device__ __forceinline__ void run(double (&a)[4][4], int tx, int ty, int64_t col_offset,
                                               int64_t* info, double* colbuf) {
        __syncthreads();                         // column J (unscaled) is in colbuf[J & 1]
        const double* cb = colbuf + (J & 1) * 64;
        const double piv = cb[J];
        if (!(piv > 0.0) && threadIdx.x == 0)
            atomicMin((unsigned long long*)info, (unsigned long long)(col_offset + J));
        double cr[4], cc[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) { cr[i] = cb[ty + 16 * i]; cc[i] = cb[tx + 16 * i]; }
        const double ip = (piv > 0.0) ? fast_rcp(piv) : __builtin_nan("");
#pragma unroll
        for (int jj = 0; jj < 4; ++jj) {
            if (16 * jj + 15 > J) {              // tile column still has columns > J
                const double w = cc[jj] * ip;
                const bool live = (tx + 16 * jj) > J;
#pragma unroll
                for (int i = jj; i < 4; ++i)     // lower tiles only (i >= jj)
                    if (live) a[i][jj] = fma(-cr[i], w, a[i][jj]);
            }
        }
        // owners of column J+1 publish it (rows above J+1 are never read)
        if constexpr (J < 63) {
            constexpr int NT = (J + 1) >> 4;
            if (tx == ((J + 1) & 15)) {
                double* nb = colbuf + ((J + 1) & 1) * 64;
#pragma unroll
                for (int i = NT; i < 4; ++i) nb[ty + 16 * i] = a[i][NT];
            }
        }
        // the scaling by 1/sqrt(pivot) is deferred to the end of the kernel: its 15-deep
        // dependent chain would otherwise sit in front of every barrier
        if (threadIdx.x == 0) colbuf[128 + J] = piv;
        Potf2Step<J + 1>::run(a, tx, ty, col_offset, info, colbuf);
    }
};
template <>
struct Potf2Step<64> {
    static __device__ __forceinline__ void run(double (&)[4][4], int, int, int64_t, int64_t*, double*) {}
};

__global__ __launch_bounds__(256) void potf2_64_kernel(double* A, int64_t ld, int64_t col_offset,
                                                        int64_t* info) {
    __shared__ __attribute__((aligned(16))) double colbuf[128 + 64 + 128];   // columns | pivots | sqrt, 1/sqrt
    const int tx = threadIdx.x & 15, ty = threadIdx.x >> 4;
    double a[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int jj = 0; jj < 4; ++jj)
            a[i][jj] = (i >= jj) ? A[(int64_t)(ty + 16 * i) * ld + tx + 16 * jj] : 0.0;
    if (tx == 0) {
#pragma unroll
        for (int i = 0; i < 4; ++i) colbuf[ty + 16 * i] = a[i][0];
    }
    Potf2Step<0>::run(a, tx, ty, col_offset, info, colbuf);
    // L[r][c] = (unscaled column entry) / sqrt(pivot_c), L[c][c] = sqrt(pivot_c): 64 parallel
    // square roots, then one scaling pass
    __syncthreads();
    if (threadIdx.x < 64) {
        double sq, rinv;
        sqrt_and_rsqrt(colbuf[128 + threadIdx.x], sq, rinv);
        colbuf[192 + threadIdx.x] = sq;
        colbuf[256 + threadIdx.x] = rinv;
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int jj = 0; jj <= i; ++jj) {
            const int r = ty + 16 * i, c = tx + 16 * jj;
            if (c < r) A[(int64_t)r * ld + c] = a[i][jj] * colbuf[256 + c];
            else if (c == r) A[(int64_t)r * ld + c] = colbuf[192 + c];
        }
}

hipError_t launch_potf2_64(hipStream_t s, double* A, int64_t ld, int64_t col_offset,
                           int64_t* info_dev) {
    hipLaunchKernelGGL(potf2_64_kernel, dim3(1), dim3(256), 0, s, A, ld, col_offset, info_dev);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------
// trsm_rlt64: X (m x 64) <- X * L^-T by substitution along each row, right-looking:
//   x[c] *= 1/L[c][c];  then  x[c'] -= x[c] * L[c'][c]  for every c' > c
// (independent FMAs, so the dependent chain is 64 multiply+FMA steps, not 2016).
// One lane per row of X; L^T sits in LDS (column c of L contiguous) and is
// read as wave-wide 16-byte broadcasts.  One true division per column per block.
// ---------------------------------------------------------------------------
constexpr int TRSM_THREADS = 128;

template <int C>
struct TrsmStep {
    static __device__ __forceinline__ void run(double (&x)[64], const double* Lt, const double* rd) {
        const double xc = x[C] * rd[C];
        x[C] = xc;
        if constexpr (C < 63) {
            constexpr int C0 = (C + 1) & ~1;
            const double* col = Lt + C * 64;     // col[c'] = L[c'][C]
#pragma unroll
            for (int c = C0; c < 64; c += 2) {
                const d2 lv = *reinterpret_cast<const d2*>(col + c);
                if (c > C) x[c] = fma(-xc, lv.x, x[c]);
                x[c + 1] = fma(-xc, lv.y, x[c + 1]);
            }
        }
        TrsmStep<C + 1>::run(x, Lt, rd);
    }
};
template <>
struct TrsmStep<64> {
    static __device__ __forceinline__ void run(double (&)[64], const double*, const double*) {}
};

__global__ __launch_bounds__(TRSM_THREADS) void trsm_rlt64_kernel(const double* L, int64_t ldl,
                                                                   double* X, int64_t ldx, int64_t m) {
    __shared__ __attribute__((aligned(16))) double Lt[64 * 64];
    __shared__ double rd[64];
    const int tid = threadIdx.x;
    // stage L transposed: Lt[c][r] = L[r][c]; global reads coalesced along c
    for (int p = tid; p < 64 * 64; p += TRSM_THREADS) {
        const int r = p >> 6, c = p & 63;
        Lt[c * 64 + r] = (c <= r) ? L[(int64_t)r * ldl + c] : 0.0;
    }
    if (tid < 64) rd[tid] = 1.0 / L[(int64_t)tid * ldl + tid];
    __syncthreads();
    const int64_t row = (int64_t)blockIdx.x * TRSM_THREADS + tid;
    if (row >= m) return;
    double* xr = X + row * ldx;
    double x[64];
#pragma unroll
    for (int c = 0; c < 64; c += 2) {
        const d2 v = *reinterpret_cast<const d2*>(xr + c);
        x[c] = v.x;
        x[c + 1] = v.y;
    }
    TrsmStep<0>::run(x, Lt, rd);
#pragma unroll
    for (int c = 0; c < 64; c += 2) *reinterpret_cast<d2*>(xr + c) = d2{x[c], x[c + 1]};
}

// ---------------------------------------------------------------------------
// trsm_rlt64, wave-per-row form: lane c holds x[c] of a row; step C broadcasts
// x[C] / L[C][C] with v_readlane and every lane c' > C applies
// x[c'] -= x[C] * L[c'][C] (column C of L contiguous in LDS: conflict-free).
// The dependent chain per row is 64 x (readlane, mul, fma) ~ 2k cycles instead of
// 2016 FMAs per lane, rows are read and written as whole 512-byte lines, and a
// wave interleaves TRSM_WR independent rows for latency cover.  This is what makes
// the diagonal-block factorisation (few rows, launch after launch) short; for the
// tall panels it costs about the same as the lane-per-row form.
// ---------------------------------------------------------------------------
constexpr int TRSM_WR = 4;      // rows per wave
constexpr int TRSM_WT = 256;    // threads per block (4 waves -> 16 rows per block)

__global__ __launch_bounds__(TRSM_WT) void trsm_rlt64_wave_kernel(const double* L, int64_t ldl, double* X,
                                                                   int64_t ldx, int64_t m) {
    __shared__ __attribute__((aligned(16))) double Lt[64 * 64];   // Lt[c][r] = L[r][c]
    __shared__ double rd[64];
    const int tid = threadIdx.x;
    for (int p = tid; p < 64 * 64; p += TRSM_WT) {
        const int r = p >> 6, c = p & 63;
        Lt[c * 64 + r] = (c <= r) ? L[(int64_t)r * ldl + c] : 0.0;
    }
    if (tid < 64) rd[tid] = 1.0 / L[(int64_t)tid * ldl + tid];
    __syncthreads();
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int64_t row0 = ((int64_t)blockIdx.x * (TRSM_WT / 64) + wave) * TRSM_WR;
    if (row0 >= m) return;
    double x[TRSM_WR];
#pragma unroll
    for (int i = 0; i < TRSM_WR; ++i) x[i] = (row0 + i < m) ? X[(row0 + i) * ldx + lane] : 0.0;
#pragma unroll 4
    for (int C = 0; C < 64; ++C) {
        const double lc = Lt[C * 64 + lane];       // L[lane][C] (zero above the diagonal)
        const double rdc = rd[C];
#pragma unroll
        for (int i = 0; i < TRSM_WR; ++i) {
            const double xc = readlane_f64(x[i], C) * rdc;
            x[i] = (lane == C) ? xc : fma(-xc, (lane > C) ? lc : 0.0, x[i]);
        }
    }
#pragma unroll
    for (int i = 0; i < TRSM_WR; ++i)
        if (row0 + i < m) X[(row0 + i) * ldx + lane] = x[i];
}

hipError_t launch_trsm_rlt64(hipStream_t s, const double* L, int64_t ldl, double* X, int64_t ldx,
                             int64_t m) {
    if (m <= 0) return hipSuccess;
    // short panels (diagonal blocks, small problems, a rank's share): the wave-per-row kernel,
    // whose latency is ~10x lower; tall panels run hidden under the trailing update, where the
    // lane-per-row kernel disturbs the concurrent MFMA stream less
    if (tuning().trsm_wave && m <= 16384) {
        const int rows_per_block = (TRSM_WT / 64) * TRSM_WR;
        const int blocks = (int)((m + rows_per_block - 1) / rows_per_block);
        hipLaunchKernelGGL(trsm_rlt64_wave_kernel, dim3(blocks), dim3(TRSM_WT), 0, s, L, ldl, X, ldx, m);
    } else {
        const int blocks = (int)((m + TRSM_THREADS - 1) / TRSM_THREADS);
        hipLaunchKernelGGL(trsm_rlt64_kernel, dim3(blocks), dim3(TRSM_THREADS), 0, s, L, ldl, X, ldx, m);
    }
    return hipGetLastError();
}

}  // namespace gpmi
