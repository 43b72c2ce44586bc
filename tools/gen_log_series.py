"""Coefficients for the long tier of se3_log_fast (csrc/tolg_lie.h), computed in 60-digit arithmetic (mpmath):
 * f(y) = asin(sqrt y) / sqrt y on [0, Y]: interpolation at the Chebyshev nodes of [0, Y], converted to powers of y --
   a polynomial of the same length as the Taylor series that served y < 1/16 covers y < 1/4 (a deviation of 60 degrees
   instead of 29): the singularity at y = 1 sits 14 half-widths away in the Chebyshev variable instead of 16 radii away
   in the Taylor one, but the interpolant spends its degrees of freedom on the interval, not on the disc.
 * L(t2) = 1/t^2 - cot(t/2)/(2t) = sum |B_{2k+2}| / (2k+2)! t^2k: Taylor coefficients (radius (2 pi)^2).
Prints C arrays and the measured maximum relative errors.   usage: python tools/gen_log_series.py [Y] [degree]"""
import sys

import mpmath as mp

mp.mp.dps = 60


def f(y):
    if y == 0:
        return mp.mpf(1)
    s = mp.sqrt(y)
    return mp.asin(s) / s


def cheb_fit_monomial(Y, deg):
    n = deg + 1
    nodes = [(mp.cos(mp.pi * (2 * k + 1) / (2 * n)) + 1) / 2 * Y for k in range(n)]
    V = mp.matrix(n, n)
    rhs = mp.matrix(n, 1)
    for i, y in enumerate(nodes):
        for j in range(n):
            V[i, j] = y ** j
        rhs[i] = f(y)
    c = mp.lu_solve(V, rhs)
    return [c[j] for j in range(n)]


def horner_double(c, y):
    """Horner in IEEE double (what the kernel does), result as mpf"""
    r = float(c[-1])
    y = float(y)
    for k in range(len(c) - 2, -1, -1):
        r = r * y + float(c[k])  # (fma would round once; this bounds it from above)
    return mp.mpf(r)


def main():
    Y = mp.mpf(sys.argv[1]) if len(sys.argv) > 1 else mp.mpf(1) / 4
    deg = int(sys.argv[2]) if len(sys.argv) > 2 else 16
    c = cheb_fit_monomial(Y, deg)
    worst_exact, worst_dbl = mp.mpf(0), mp.mpf(0)
    for k in range(4001):
        y = Y * k / 4000
        p = sum(cj * y ** j for j, cj in enumerate(c))
        worst_exact = max(worst_exact, abs(p / f(y) - 1))
        worst_dbl = max(worst_dbl, abs(horner_double(c, y) / f(mp.mpf(float(y))) - 1))
    print("// asin(sqrt y)/sqrt y on [0, %s], degree %d: max rel error %s (exact coefficients), %s (double Horner)"
          % (mp.nstr(Y, 5), deg, mp.nstr(worst_exact, 3), mp.nstr(worst_dbl, 3)))
    print("const double AL[%d] = {%s};" % (len(c), ", ".join(mp.nstr(x, 17) for x in c)))
    taylor = [mp.factorial(2 * k) / (4 ** k * mp.factorial(k) ** 2 * (2 * k + 1)) for k in range(6)]
    print("// Taylor, first six:", ", ".join(mp.nstr(x, 17) for x in taylor))
    L = [abs(mp.bernoulli(2 * k + 2)) / mp.factorial(2 * k + 2) for k in range(13)]
    print("const double L[%d] = {%s};" % (len(L), ", ".join(mp.nstr(x, 17) for x in L)))
    for T2 in (mp.mpf("1.21"), mp.mpf("1.1")):
        for n in (11, 12, 13):
            worst = mp.mpf(0)
            for k in range(1, 401):
                t2 = T2 * k / 400
                t = mp.sqrt(t2)
                ref = 1 / t2 - mp.cot(t / 2) / (2 * t)
                worst = max(worst, abs(sum(L[j] * t2 ** j for j in range(n)) / ref - 1))
            print("// L: %d terms on t2 <= %s: truncation %s" % (n, mp.nstr(T2, 4), mp.nstr(worst, 3)))


if __name__ == "__main__":
    main()


def constrained(Y, NS, dg):
    """p(y) = Taylor_{<NS}(y) + y^NS * q(y), q = Chebyshev interpolant (degree dg) of (f - Taylor_{<NS}) / y^NS on [0, Y]:
    the short tier (the first NS Taylor coefficients) stays a prefix of the array, as horner2 wants it."""
    a = [mp.factorial(2 * k) / (4 ** k * mp.factorial(k) ** 2 * (2 * k + 1)) for k in range(NS + 60)]

    def g(y):  # (f - sum_{k<NS} a_k y^k) / y^NS, by its own series near 0 (no cancellation)
        if y < mp.mpf("0.05"):
            return sum(a[NS + j] * y ** j for j in range(60))
        return (f(y) - sum(a[k] * y ** k for k in range(NS))) / y ** NS

    n = dg + 1
    nodes = [(mp.cos(mp.pi * (2 * k + 1) / (2 * n)) + 1) / 2 * Y for k in range(n)]
    V = mp.matrix(n, n); rhs = mp.matrix(n, 1)
    for i, y in enumerate(nodes):
        for j in range(n):
            V[i, j] = y ** j
        rhs[i] = g(y)
    q = mp.lu_solve(V, rhs)
    c = a[:NS] + [q[j] for j in range(n)]
    worst_exact, worst_dbl = mp.mpf(0), mp.mpf(0)
    for k in range(4001):
        y = Y * k / 4000
        p = sum(cj * y ** j for j, cj in enumerate(c))
        worst_exact = max(worst_exact, abs(p / f(y) - 1))
        worst_dbl = max(worst_dbl, abs(horner_double(c, y) / f(mp.mpf(float(y))) - 1))
    print("// constrained: first %d Taylor, then degree-%d fit on [0, %s]: %d coefficients, max rel error %s (exact), %s (double Horner)"
          % (NS, dg, mp.nstr(Y, 5), len(c), mp.nstr(worst_exact, 3), mp.nstr(worst_dbl, 3)))
    print("const double A[%d] = {%s};" % (len(c), ", ".join(mp.nstr(x, 17) for x in c)))


if __name__ == "__main__" and len(sys.argv) > 3:
    constrained(mp.mpf(sys.argv[1]), 6, int(sys.argv[3]))
