/* Driver for the CPU-only sanitizer job (tests/test_sanitize_cpu.py): oracle/rbf_oracle.c under
 * gcc -fsanitize=address,undefined on edge shapes -- N = 1, ragged M, every branch of the pairwise summation
 * (d < 8, 8 <= d <= 128 with and without a tail, d > 128), a leading dimension larger than M, the diagonal term.
 * Exit code 0 and an empty stderr are the pass criteria; a handful of values are checked against closed forms. */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <stdint.h>

void rbf_oracle(const double* a, int64_t N, const double* b, int64_t M, int64_t d, double coef, double sig2,
                double diag_add, double* out, int64_t ld);

static int check(int64_t N, int64_t M, int64_t d, int64_t ld) {
    double* a = malloc(sizeof(double) * (size_t)(N * d));
    double* b = malloc(sizeof(double) * (size_t)(M * d));
    double* out = malloc(sizeof(double) * (size_t)(N * ld));
    if (!a || !b || !out) return 2;
    for (int64_t i = 0; i < N * d; ++i) a[i] = sin(0.37 * (double)i);
    for (int64_t i = 0; i < M * d; ++i) b[i] = cos(0.11 * (double)i);
    for (int64_t i = 0; i < N * ld; ++i) out[i] = -7.0;
    rbf_oracle(a, N, b, M, d, -.125, 2.25, 5e-4, out, ld);
    int bad = 0;
    for (int64_t i = 0; i < N && !bad; ++i) {
        for (int64_t j = 0; j < M; ++j) {
            double s = 0.0;
            for (int64_t k = 0; k < d; ++k) { const double e = a[i * d + k] - b[j * d + k]; s += e * e; }
            const double want = 2.25 * exp(-.125 * s) + (i == j ? 5e-4 : 0.0);
            if (fabs(out[i * ld + j] - want) > 1e-12 * (1.0 + fabs(want))) { bad = 1; break; }
        }
        for (int64_t j = M; j < ld; ++j)
            if (out[i * ld + j] != -7.0) { bad = 1; break; }     /* nothing beyond column M is touched */
    }
    free(a); free(b); free(out);
    return bad;
}

int main(void) {
    const int64_t ds[] = {1, 3, 7, 8, 9, 16, 17, 127, 128, 129, 200, 300};
    int fails = 0;
    for (unsigned t = 0; t < sizeof ds / sizeof ds[0]; ++t) {
        fails += check(1, 1, ds[t], 1);
        fails += check(5, 3, ds[t], 3);
        fails += check(3, 7, ds[t], 9);
        fails += check(33, 33, ds[t], 40);
    }
    if (fails) { fprintf(stderr, "rbf_oracle_driver: %d case(s) failed\n", fails); return 1; }
    printf("rbf_oracle_driver: ok\n");
    return 0;
}
