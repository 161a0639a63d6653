#!/usr/bin/env python3
"""Coefficients of the fp64 logarithm of the MI epilogue (ldw_epi.h fast_log_ratio): q(z) = (atanh(s)/s - 1)/z, z = s^2,
as a near-minimax polynomial (Chebyshev-node interpolation, mpmath 60 digits) on [0, Z].  Prints the coefficients as C
hex-float literals and the largest error of log = 2s(1 + z q(z)) over the interval."""
import sys
import mpmath as mp
mp.mp.dps = 60
SMAX = mp.mpf(sys.argv[1]) if len(sys.argv) > 1 else mp.mpf('0.2006')
Z = SMAX * SMAX
def q(z):
    if z == 0:
        return mp.mpf(1) / 3
    s = mp.sqrt(z)
    return (mp.atanh(s) / s - 1) / z
for deg in (5, 6, 7):
    n = deg + 1
    nodes = [Z / 2 * (1 + mp.cos(mp.pi * (2 * k + 1) / (2 * n))) for k in range(n)]
    A = mp.matrix(n, n)
    b = mp.matrix(n, 1)
    for i, x in enumerate(nodes):
        for j in range(n):
            A[i, j] = x ** j
        b[i] = q(x)
    c = mp.lu_solve(A, b)
    cf = [float(c[j]) for j in range(n)]   # rounded to fp64: what the kernel uses
    worst = 0
    for t in range(2001):
        z = Z * t / 2000
        p = sum(mp.mpf(cf[j]) * z ** j for j in range(n))
        s = mp.sqrt(z)
        err = abs(2 * s * z * (p - q(z)))     # absolute error of the logarithm
        worst = max(worst, err)
    print("degree", deg, "max abs error of log", mp.nstr(worst, 3))
    if deg == 6:
        for j in range(n):
            print("   c%d = %s   // %.17g" % (j, cf[j].hex(), cf[j]))
