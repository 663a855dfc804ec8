// Panel kernels of the blocked Cholesky: the 64 x 64 diagonal-block
// factorisation and the triangular solve of the rows below it.
//
// Both keep one matrix ROW per lane in registers (64 f64 = 128 VGPRs) and run
// the textbook recurrences fully unrolled, so that every array index is a
// compile-time constant (no scratch).  They replace the inner loops of LAPACK
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
// potf2_64: right-looking unblocked Cholesky of a 64 x 64 block, one wavefront.
// Lane r owns row r.  Step j: pivot = A[j][j] broadcast with v_readlane,
// column j scaled, then every lane updates its row with the (broadcast)
// entries of column j.  A non-positive (or NaN) pivot records
// col_offset + j in *info (atomic min) and poisons the block with NaN.
// ---------------------------------------------------------------------------
template <int J>
struct Potf2Step {
    static __device__ __forceinline__ void run(double (&a)[64], int lane, int64_t col_offset,
                                               int64_t* info) {
        const double piv = readlane_f64(a[J], J);
        if (!(piv > 0.0)) {
            if (lane == 0) atomicMin((unsigned long long*)info, (unsigned long long)(col_offset + J));
        }
        const double s = sqrt(piv);     // NaN for a negative pivot: poisons what follows
        const double l = a[J] / s;      // lanes < J hold upper-triangle garbage, never stored
        a[J] = (lane == J) ? s : l;
#pragma unroll
        for (int c = J + 1; c < 64; ++c) {
            const double lc = readlane_f64(l, c);
            a[c] = fma(-l, lc, a[c]);
        }
        Potf2Step<J + 1>::run(a, lane, col_offset, info);
    }
};
template <>
struct Potf2Step<64> {
    static __device__ __forceinline__ void run(double (&)[64], int, int64_t, int64_t*) {}
};

__global__ __launch_bounds__(64) void potf2_64_kernel(double* A, int64_t ld, int64_t col_offset,
                                                       int64_t* info) {
    const int lane = threadIdx.x;
    double a[64];
    double* row = A + (int64_t)lane * ld;
#pragma unroll
    for (int c = 0; c < 64; c += 2) {
        const d2 v = *reinterpret_cast<const d2*>(row + c);
        a[c] = v.x;
        a[c + 1] = v.y;
    }
    Potf2Step<0>::run(a, lane, col_offset, info);
    // store the lower part of the row (columns <= lane); pairs straddling the
    // diagonal keep the old upper element
#pragma unroll
    for (int c = 0; c < 64; c += 2) {
        if (c + 1 <= lane) {
            *reinterpret_cast<d2*>(row + c) = d2{a[c], a[c + 1]};
        } else if (c == lane) {
            row[c] = a[c];
        }
    }
}

hipError_t launch_potf2_64(hipStream_t s, double* A, int64_t ld, int64_t col_offset,
                           int64_t* info_dev) {
    hipLaunchKernelGGL(potf2_64_kernel, dim3(1), dim3(64), 0, s, A, ld, col_offset, info_dev);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------
// trsm_rlt64: X (m x 64) <- X * L^-T by forward substitution along each row:
//   x[c] = (x[c] - sum_{k<c} x[k] * L[c][k]) / L[c][c]
// One lane per row of X; L sits in LDS and is read as wave-wide broadcasts.
// The division is a multiplication with 1/L[c][c] (one true division per
// column per block).
// ---------------------------------------------------------------------------
constexpr int TRSM_THREADS = 128;

template <int C>
struct TrsmStep {
    static __device__ __forceinline__ void run(double (&x)[64], const double* Ls, const double* rd) {
        double s = x[C];
#pragma unroll
        for (int k = 0; k < C; ++k) s = fma(-x[k], Ls[C * 64 + k], s);
        x[C] = s * rd[C];
        TrsmStep<C + 1>::run(x, Ls, rd);
    }
};
template <>
struct TrsmStep<64> {
    static __device__ __forceinline__ void run(double (&)[64], const double*, const double*) {}
};

__global__ __launch_bounds__(TRSM_THREADS) void trsm_rlt64_kernel(const double* L, int64_t ldl,
                                                                   double* X, int64_t ldx, int64_t m) {
    __shared__ __attribute__((aligned(16))) double Ls[64 * 64];
    __shared__ double rd[64];
    const int tid = threadIdx.x;
    // stage L (row-major 64 x 64), coalesced 16-byte pieces
    for (int p = tid; p < 64 * 32; p += TRSM_THREADS) {
        const int r = p >> 5, c2 = (p & 31) * 2;
        const d2 v = *reinterpret_cast<const d2*>(L + (int64_t)r * ldl + c2);
        *reinterpret_cast<d2*>(&Ls[r * 64 + c2]) = v;
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
    TrsmStep<0>::run(x, Ls, rd);
#pragma unroll
    for (int c = 0; c < 64; c += 2) *reinterpret_cast<d2*>(xr + c) = d2{x[c], x[c + 1]};
}

hipError_t launch_trsm_rlt64(hipStream_t s, const double* L, int64_t ldl, double* X, int64_t ldx,
                             int64_t m) {
    if (m <= 0) return hipSuccess;
    const int blocks = (int)((m + TRSM_THREADS - 1) / TRSM_THREADS);
    hipLaunchKernelGGL(trsm_rlt64_kernel, dim3(blocks), dim3(TRSM_THREADS), 0, s, L, ldl, X, ldx, m);
    return hipGetLastError();
}

}  // namespace gpmi
