"""Minimax polynomial for the K build's exp (csrc/rbf.hip, exp_neg_fast).

exp(r) ~ 1 + r + r^2 * P(r),  P of degree DEG-2, |r| <= A (ln2/2 plus a margin for the rounding of n),
minimising the RELATIVE error of exp -- Remez exchange on g(r) = (exp(r)-1-r)/r^2 with weight
r^2/exp(r), 60-digit arithmetic (mpmath).  The first two coefficients stay exactly 1 so that
exp(0) = 1 (the diagonal of K is sigma^2 exactly, as in NumPy).

Prints the coefficients as C hex-float / decimal literals and the achieved error.
usage: python scripts/exp_poly_fit.py [total degree, default 11]
"""
import sys
from mpmath import mp, mpf, exp, matrix, lu_solve, cos, pi, log

mp.dps = 60
DEG = int(sys.argv[1]) if len(sys.argv) > 1 else 11
NP = DEG - 1                      # free coefficients c2..cDEG  -> P has NP terms
A = log(2) / 2 * mpf("1.0005")


def g(x):
    if abs(x) < mpf("1e-8"):
        return mpf(1) / 2 + x / 6 + x * x / 24
    return (exp(x) - 1 - x) / (x * x)


def W(x):
    return x * x / exp(x)


def P(c, x):
    s = mpf(0)
    for cj in reversed(c):
        s = s * x + cj
    return s


def werr(c, x):
    return W(x) * (P(c, x) - g(x))


# initial reference: Chebyshev extrema, none at zero
ref = [A * cos(pi * mpf(k) / NP) for k in range(NP + 1)][::-1]
ref = [x if abs(x) > A / 50 else A / 50 for x in ref]
for it in range(30):
    M = matrix(NP + 1, NP + 1)
    b = matrix(NP + 1, 1)
    for i, x in enumerate(ref):
        for j in range(NP):
            M[i, j] = x ** j
        M[i, NP] = (-1) ** i / W(x)
        b[i] = g(x)
    sol = lu_solve(M, b)
    c = [sol[j] for j in range(NP)]
    E = sol[NP]
    # extrema of the weighted error between sign changes of P - g
    grid = [-A + 2 * A * mpf(k) / 6000 for k in range(6001)]
    vals = [werr(c, x) for x in grid]
    raw = [P(c, x) - g(x) for x in grid]
    segs, start = [], 0
    for k in range(1, len(grid)):
        if (raw[k] > 0) != (raw[k - 1] > 0):
            segs.append((start, k)); start = k
    segs.append((start, len(grid)))
    new = []
    for a0, a1 in segs:
        kbest = max(range(a0, a1), key=lambda k: abs(vals[k]))
        new.append(grid[kbest])
    emax = max(abs(v) for v in vals)
    print("iter %d  |E| %.4e  max weighted err %.4e  segments %d" % (it, abs(E), emax, len(segs)), file=sys.stderr)
    if len(new) != NP + 1:
        # keep the NP+1 largest while preserving order (rare; happens when a zero crossing is missed)
        new = sorted(sorted(new, key=lambda x: -abs(werr(c, x)))[:NP + 1])
    if abs(emax - abs(E)) < abs(E) * mpf("1e-6"):
        break
    ref = new

coef = [mpf(1), mpf(1)] + c
print("// minimax degree %d on |r| <= %.6f, relative error %.3e (Taylor of the same degree: %.3e)" % (
    DEG, float(A), float(emax), float(A ** (DEG + 1) / mp.factorial(DEG + 1))))
for k, ck in enumerate(coef):
    d = float(ck)
    print("c%-2d = %-24s // %s   (1/%d! = %.17g)" % (k, d.hex(), repr(d), k, float(1 / mp.factorial(k))))

# error of the double-rounded coefficients, evaluated exactly (no Horner rounding), relative to exp
cd = [mpf(float(ck)) for ck in coef]
worst = mpf(0)
for k in range(4001):
    x = -A + 2 * A * mpf(k) / 4000
    s = mpf(0)
    for cj in reversed(cd):
        s = s * x + cj
    worst = max(worst, abs(s / exp(x) - 1))
print("// with coefficients rounded to double: relative error %.3e = %.3f ulp(1.0)" % (float(worst), float(worst / mpf(2) ** -52)))
