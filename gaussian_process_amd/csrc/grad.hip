// grad.hip -- gradient of the log marginal likelihood w.r.t. the squared-exponential
// hyper-parameters.
//
// Reference: tune_hyperparms_regression.py:31-64 (gradient_ascent):
//     l_grad   = sigma**2 * exp(-.5*sqdist/l**2) * (sqdist/l**3)            (:54)
//     l_matrix = dot(dot(alpha, alpha.T) - K_y, l_grad); l_var = .5*trace   (:55-57)
// and the commented-out sigma twin (:46-52, sigma_grad = 2*sigma*exp(...)).
// The reference forms two N x N products to read off a trace; trace(W D) with
// D symmetric is sum_ij W_ij D_ij, so one fused pass over the matrix does it:
// D_ij is recomputed from X (as in the K build), W_ij = alpha_i alpha_j - Kinv_ij
// reads Kinv once.  HBM-read bound: 8 B per element.
//
// One 128 x 128 tile per block, a column pair x 32 rows per thread, X tiles in LDS
// (d <= 32) or straight from global memory; per-block partial sums, reduced in a
// fixed order (bitwise reproducible); the host adds the per-tile partials in order.
#include "gpmi_internal.h"

namespace gpmi {

namespace {

constexpr int RT = 128;
constexpr int GRAD_MAXD = 32;

struct GradDev {
    const double* A;        // rows: nA x d
    const double* B;        // cols: nB x d
    int64_t nA, nB;
    int d;
    int64_t row0;           // first global row of this launch (index into A and alpha_r)
    int Tm, Tn;
    const double* alpha_r;  // length nA
    const double* alpha_c;  // length nB
    const double* Kinv;     // element (row0 + r, c) at Kinv[r * ld + c]
    int64_t ld;
    double kinv_sign;       // Kinv holds sign * K_y^-1 (the LAUUM product is accumulated negated)
    double coef, sig2, two_sigma, inv_l3;
    int tri;                // 1: lower tiles only (triangular enumeration), strictly-lower elements count twice
    double* partial;        // 2 doubles per block
};

template <bool LDS>
__global__ __launch_bounds__(256) void grad_trace_kernel(const GradDev p) {
    extern __shared__ __attribute__((aligned(16))) double lds[];
    __shared__ double red[2][256];
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
    const int d = p.d;
    const int64_t grow0 = p.row0 + (int64_t)ti * RT;
    const int64_t gcol0 = (int64_t)tj * RT;
    const int tid = threadIdx.x;
    double* As = lds;               // [RT][d]
    double* Bs = lds + RT * d;      // [d][RT]
    if (LDS) {
        for (int e = tid; e < RT * d; e += 256) {
            const int r = e / d, k = e - r * d;
            const int64_t ga = grow0 + r, gb = gcol0 + r;
            As[r * d + k] = (ga < p.nA) ? p.A[ga * d + k] : 0.0;
            Bs[k * RT + r] = (gb < p.nB) ? p.B[gb * d + k] : 0.0;
        }
        __syncthreads();
    }
    const int cp = tid & 63, rg = tid >> 6;
    const int64_t gc = gcol0 + 2 * cp;
    const double ac0 = (gc < p.nB) ? p.alpha_c[gc] : 0.0;
    const double ac1 = (gc + 1 < p.nB) ? p.alpha_c[gc + 1] : 0.0;
    double acc_l = 0.0, acc_s = 0.0;
    for (int r = 0; r < 32; ++r) {
        const int lr = 32 * rg + r;
        const int64_t gr = grow0 + lr;
        if (gr >= p.nA) break;                                   // wave-uniform
        double s0 = 0.0, s1 = 0.0;
        if (LDS) {
            const double* ar = &As[lr * d];
            const double* bc = &Bs[2 * cp];
            for (int k = 0; k < d; ++k) {
                const double a = ar[k];
                const double e0 = a - bc[k * RT], e1 = a - bc[k * RT + 1];
                s0 = fma(e0, e0, s0);
                s1 = fma(e1, e1, s1);
            }
        } else {
            const double* ar = p.A + gr * d;
            const double* b0 = p.B + ((gc < p.nB) ? gc : 0) * d;
            const double* b1 = p.B + ((gc + 1 < p.nB) ? gc + 1 : 0) * d;
            for (int k = 0; k < d; ++k) {
                const double a = ar[k];
                const double e0 = a - b0[k], e1 = a - b1[k];
                s0 = fma(e0, e0, s0);
                s1 = fma(e1, e1, s1);
            }
        }
        const double ar_ = p.alpha_r[gr];
        const double* kp = p.Kinv + ((int64_t)ti * RT + lr) * p.ld + gc;
        const double k0 = (gc < p.nB) ? kp[0] : 0.0;
        const double k1 = (gc + 1 < p.nB) ? kp[1] : 0.0;
        const double x0 = exp(p.coef * s0), x1 = exp(p.coef * s1);
        double w0 = (gc < p.nB) ? fma(ar_, ac0, -p.kinv_sign * k0) : 0.0;
        double w1 = (gc + 1 < p.nB) ? fma(ar_, ac1, -p.kinv_sign * k1) : 0.0;
        if (p.tri) {
            // symmetric case: only elements on or below the diagonal are valid in Kinv (the GEMM
            // that produced it skips whatever lies above); strictly-lower elements count twice
            w0 *= (gc < gr) ? 2.0 : (gc == gr) ? 1.0 : 0.0;
            w1 *= (gc + 1 < gr) ? 2.0 : (gc + 1 == gr) ? 1.0 : 0.0;
        }
        // dK/dl = sigma^2 exp(.) sq / l^3 ; dK/dsigma = 2 sigma exp(.)
        acc_l += w0 * (p.sig2 * x0 * (s0 * p.inv_l3)) + w1 * (p.sig2 * x1 * (s1 * p.inv_l3));
        acc_s += w0 * (p.two_sigma * x0) + w1 * (p.two_sigma * x1);
    }
    red[0][tid] = acc_l;
    red[1][tid] = acc_s;
    __syncthreads();
    for (int off = 128; off > 0; off >>= 1) {
        if (tid < off) {
            red[0][tid] += red[0][tid + off];
            red[1][tid] += red[1][tid + off];
        }
        __syncthreads();
    }
    if (tid == 0) {
        p.partial[2 * (size_t)blockIdx.x] = red[0][0];
        p.partial[2 * (size_t)blockIdx.x + 1] = red[1][0];
    }
}

__global__ void set_identity_kernel(double* V, int64_t ld, int64_t n) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) V[i * ld + i] = 1.0;
}

}  // namespace

hipError_t launch_set_identity_diag(hipStream_t s, double* V, int64_t ld, int64_t n) {
    if (n <= 0) return hipSuccess;
    hipLaunchKernelGGL(set_identity_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, V, ld, n);
    return hipGetLastError();
}

int64_t grad_trace_blocks(const GradArgs& a) {
    const int64_t Tm = (a.nrows + RT - 1) / RT, Tn = (a.nB + RT - 1) / RT;
    return a.tri ? Tm * (Tm + 1) / 2 : Tm * Tn;
}

hipError_t launch_grad_trace(hipStream_t s, const GradArgs& a) {
    if (a.nrows <= 0 || a.nB <= 0) return hipSuccess;
    GradDev p;
    p.A = a.A; p.B = a.B; p.nA = a.nA; p.nB = a.nB; p.d = (int)a.d; p.row0 = a.row0;
    p.Tm = (int)((a.nrows + RT - 1) / RT); p.Tn = (int)((a.nB + RT - 1) / RT);
    p.alpha_r = a.alpha_r; p.alpha_c = a.alpha_c;
    p.Kinv = a.Kinv; p.ld = a.ld; p.kinv_sign = a.kinv_sign;
    p.coef = a.coef; p.sig2 = a.sig2; p.two_sigma = a.two_sigma; p.inv_l3 = a.inv_l3;
    p.tri = a.tri; p.partial = a.partial;
    const dim3 grid((unsigned)grad_trace_blocks(a)), block(256);
    if (a.d <= GRAD_MAXD) {
        const size_t lds = (size_t)2 * RT * a.d * sizeof(double);
        hipLaunchKernelGGL(grad_trace_kernel<true>, grid, block, lds, s, p);
    } else {
        hipLaunchKernelGGL(grad_trace_kernel<false>, grid, block, 0, s, p);
    }
    return hipGetLastError();
}

}  // namespace gpmi
