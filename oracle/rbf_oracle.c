/* CPU oracle, C part -- TEST INFRASTRUCTURE, NOT PRODUCT (see gp_oracle.py).
 *
 * rbf_oracle(): the arithmetic of RBF_kernel (/root/reference GP_regression.py:18-19)
 * restated element by element so that mid-size checks and the CPU baseline do
 * not need the reference's (N,d,M) broadcast temporary:
 *     sqdist = sum_k (a[i,k]-b[j,k])^2   in the order of NumPy's add.reduce over the
 *              middle axis (pairwise_sum of numpy/_core/src/umath/loops_utils.h.src:
 *              sequential for d < 8; 8 interleaved partial sums, a fixed tree and a
 *              sequential tail for d <= 128; recursive halving above), one rounding
 *              per operation -- verified bit-for-bit against NumPy 2.2.6
 *     out    = sig2 * exp(coef * sqdist)   coef = -.5*(1/l^2) computed by the caller
 * Build: gcc -O2 -fopenmp -ffp-contract=off -shared -fPIC (oracle/Makefile).
 * Pinned by tests/test_oracle_vs_golden.py against the reference's own outputs.
 */
#include <math.h>
#include <stdint.h>

static double sq_pairwise(const double* a, const double* b, int64_t n)
{
    if (n < 8) {
        double res = 0.0;
        for (int64_t k = 0; k < n; ++k) { const double e = a[k] - b[k]; res = res + e * e; }
        return res;
    } else if (n <= 128) {
        double r[8];
        for (int j = 0; j < 8; ++j) { const double e = a[j] - b[j]; r[j] = e * e; }
        int64_t i = 8;
        for (; i < n - (n % 8); i += 8)
            for (int j = 0; j < 8; ++j) { const double e = a[i + j] - b[i + j]; r[j] = r[j] + e * e; }
        double res = ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]));
        for (; i < n; ++i) { const double e = a[i] - b[i]; res = res + e * e; }
        return res;
    } else {
        int64_t n2 = n / 2;
        n2 -= n2 % 8;
        return sq_pairwise(a, b, n2) + sq_pairwise(a + n2, b + n2, n - n2);
    }
}

void rbf_oracle(const double* a, int64_t N, const double* b, int64_t M, int64_t d,
                double coef, double sig2, double diag_add, double* out, int64_t ld)
{
#pragma omp parallel for schedule(static)
    for (int64_t i = 0; i < N; ++i) {
        const double* ai = a + i * d;
        double* o = out + i * ld;
        for (int64_t j = 0; j < M; ++j) {
            const double* bj = b + j * d;
            const double s = sq_pairwise(ai, bj, d);
            o[j] = sig2 * exp(coef * s);
        }
        if (diag_add != 0.0 && i < M) o[i] += diag_add;
    }
}
