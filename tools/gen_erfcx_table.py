"""Piecewise polynomials of erfcx(y) = exp(y^2) erfc(y) for the fp64 pair body (csrc/srpot.h, kErfcxTab): intervals of width W
on [0, Y), degree D each, in the local variable u = y - centre; Chebyshev interpolation at D + 1 nodes converted to monomials.
Prints the table as a C initialiser and the maximal relative error against scipy.special.erfcx on a fine grid.
    python tools/gen_erfcx_table.py [W] [Y] [D]"""
import sys

import numpy as np
from numpy.polynomial import chebyshev as C
from scipy.special import erfcx

W = float(sys.argv[1]) if len(sys.argv) > 1 else 0.5
Y = float(sys.argv[2]) if len(sys.argv) > 2 else 6.5
D = int(sys.argv[3]) if len(sys.argv) > 3 else 10
n_int = int(round(Y / W))
rows, worst = [], 0.0
for j in range(n_int):
    c0 = (j + 0.5) * W
    h = 0.5 * W
    k = np.arange(D + 1)
    nodes = np.cos(np.pi * (k + 0.5) / (D + 1))
    cheb = C.chebfit(nodes, erfcx(c0 + h * nodes), D)
    mono_x = C.cheb2poly(cheb)  # in x = u / h
    mono_u = mono_x / h ** np.arange(D + 1)
    rows.append(mono_u)
    u = np.linspace(-h, h, 4001)
    p = np.zeros_like(u)
    for a in mono_u[::-1]:
        p = p * u + a
    err = np.max(np.abs(p / erfcx(c0 + u) - 1))
    worst = max(worst, err)
print(f"// erfcx(y), y in [0, {Y}): {n_int} intervals of width {W}, degree {D} in u = y - centre; max relative error {worst:.2e}")
print(f"static constexpr int kErfcxIntervals = {n_int}, kErfcxTerms = {D + 1};")
print(f"static constexpr double kErfcxWidthInv = {1.0 / W!r}, kErfcxWidth = {W!r}, kErfcxEnd = {Y!r};")
print("#define MIPME_ERFCX_TAB \\")
for j, r in enumerate(rows):
    body = ", ".join(f"{v:.17e}" for v in r)
    print(f"  {body}{',' if j + 1 < n_int else ''} \\")
print("")
