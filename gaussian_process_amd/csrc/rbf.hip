// Squared-exponential kernel-matrix build (reference: RBF_kernel,
// GP_regression.py:8-19, and the "+ s * np.eye(N)" of :138 fused into the
// diagonal).
//
//   out[i][j] = sig2 * exp(coef * sum_k (a[i,k] - b[j,k])^2)      coef = -.5*(1/l^2)
//
// The per-element arithmetic follows the reference exactly: difference, square
// and a sequential (k = 0..d-1) sum with one rounding per operation (no FMA
// contraction), one multiplication by coef, exp, one multiplication by sigma^2.
// Only exp() itself may differ from NumPy's (both are < 1 ulp).
//
// HBM-write bound: a block produces a 128 x 128 tile; a wavefront writes whole
// 1-KiB row segments (64 lanes x 16 B).  The x rows of both tile edges are
// staged once in LDS (k-chunks of 16); the b values of a thread's two columns
// live in registers, the a values are wave-wide LDS broadcasts.
#include "gpmi_internal.h"

namespace gpmi {

typedef double d2 __attribute__((ext_vector_type(2)));

constexpr int RT = 128;    // tile edge
constexpr int DK = 16;     // k-chunk held in LDS / registers

struct RbfDev {
    const double* A;
    const double* B;
    int64_t nA, nB;
    int d;
    int64_t row0;
    int Tm, Tn;
    double coef, sig2, diag_add;
    int symmetric;
    int tri;               // triangular tile enumeration (symmetric, row0 == 0)
    double* out;
    int64_t ld;
};

#pragma clang fp contract(off)
template <int DC>   // DC: compile-time chunk length (1..16), 0 = runtime
__global__ __launch_bounds__(256) void rbf_kernel(const RbfDev p) {
    __shared__ __attribute__((aligned(16))) double As[RT * DK];
    __shared__ __attribute__((aligned(16))) double Bs[RT * DK];
    int ti, tj;
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
    const int64_t grow0 = p.row0 + (int64_t)ti * RT;   // global first row of tile
    const int64_t gcol0 = (int64_t)tj * RT;
    if (p.symmetric && gcol0 > grow0 + RT - 1) return;  // tile above the diagonal

    const int tid = threadIdx.x;
    const int cp = tid & 63;       // column pair: cols 2cp, 2cp+1
    const int rg = tid >> 6;       // row group: rows 32*rg .. +32
    double acc[32][2];
#pragma unroll
    for (int r = 0; r < 32; ++r) acc[r][0] = acc[r][1] = 0.0;

    const int d = p.d;
    for (int k0 = 0; k0 < d; k0 += DK) {
        const int dk = DC ? DC : ((d - k0) < DK ? (d - k0) : DK);
        if (k0) __syncthreads();
        // stage rows of A and B for this k-chunk: [row][DK]
        for (int e = tid; e < RT * dk; e += 256) {
            const int r = e / dk, k = e - r * dk;
            const int64_t ga = grow0 + r, gb = gcol0 + r;
            As[r * DK + k] = (ga < p.nA) ? p.A[ga * d + k0 + k] : 0.0;
            Bs[k * RT + r] = (gb < p.nB) ? p.B[gb * d + k0 + k] : 0.0;   // k-major: conflict-free column reads
        }
        __syncthreads();
        double b0[DK], b1[DK];
#pragma unroll
        for (int k = 0; k < DK; ++k) {
            if (k < dk) {
                const d2 bv = *reinterpret_cast<const d2*>(&Bs[k * RT + 2 * cp]);
                b0[k] = bv.x;
                b1[k] = bv.y;
            }
        }
#pragma unroll
        for (int r = 0; r < 32; ++r) {
            const double* ar = &As[(32 * rg + r) * DK];
            double s0 = acc[r][0], s1 = acc[r][1];
#pragma unroll
            for (int k = 0; k < DK; ++k) {
                if (k < dk) {
                    const double a = ar[k];
                    const double e0 = a - b0[k];
                    const double e1 = a - b1[k];
                    s0 = s0 + e0 * e0;
                    s1 = s1 + e1 * e1;
                }
            }
            acc[r][0] = s0;
            acc[r][1] = s1;
        }
    }

    const int64_t gc = gcol0 + 2 * cp;
#pragma unroll
    for (int r = 0; r < 32; ++r) {
        const int64_t gr = grow0 + 32 * rg + r;
        double v0 = p.sig2 * exp(p.coef * acc[r][0]);
        double v1 = p.sig2 * exp(p.coef * acc[r][1]);
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
        double* dst = p.out + ((int64_t)ti * RT + 32 * rg + r) * p.ld + gc;
        *reinterpret_cast<d2*>(dst) = d2{v0, v1};
    }
}

hipError_t launch_rbf(hipStream_t s, const RbfArgs& a) {
    if (a.nrows <= 0 || a.ncols <= 0) return hipSuccess;
    if (a.nrows % RT || a.ncols % RT || a.d <= 0) return hipErrorInvalidValue;
    RbfDev p;
    p.A = a.A; p.B = a.B; p.nA = a.nA; p.nB = a.nB; p.d = (int)a.d; p.row0 = a.row0;
    p.Tm = (int)(a.nrows / RT); p.Tn = (int)(a.ncols / RT);
    p.coef = a.coef; p.sig2 = a.sig2; p.diag_add = a.diag_add; p.symmetric = a.symmetric;
    p.out = a.out; p.ld = a.ld;
    p.tri = (a.symmetric && a.row0 == 0 && p.Tm == p.Tn) ? 1 : 0;
    const int64_t nblk = p.tri ? (int64_t)p.Tm * (p.Tm + 1) / 2 : (int64_t)p.Tm * p.Tn;
    dim3 grid((unsigned)nblk), block(256);
    switch (a.d) {
        case 1: hipLaunchKernelGGL(rbf_kernel<1>, grid, block, 0, s, p); break;
        case 2: hipLaunchKernelGGL(rbf_kernel<2>, grid, block, 0, s, p); break;
        case 3: hipLaunchKernelGGL(rbf_kernel<3>, grid, block, 0, s, p); break;
        case 4: hipLaunchKernelGGL(rbf_kernel<4>, grid, block, 0, s, p); break;
        case 8: hipLaunchKernelGGL(rbf_kernel<8>, grid, block, 0, s, p); break;
        case 16: hipLaunchKernelGGL(rbf_kernel<16>, grid, block, 0, s, p); break;
        default: hipLaunchKernelGGL(rbf_kernel<0>, grid, block, 0, s, p); break;
    }
    return hipGetLastError();
}

}  // namespace gpmi
