"""Winograd / Toom-Cook F(m x m, 5x5) transform matrices, m in {2, 4} (exact rationals -> float64).

y = A^T [ (G g G^T) .* (B^T d B) ] A computes an m x m block of a 5x5 *correlation* from an (m+4)^2 input block:
36 multiplications per 2x2 outputs (9 per output) or 64 per 4x4 outputs (4 per output) instead of 25.  Derived from the polynomial evaluation points (0, 1, -1, 2, -1/2, inf) by
transposing Toom-Cook multiplication: A^T = E_2^T, G = E_5, B^T = (V^-1)^T with E_k the evaluation matrix of
degree-(k-1) polynomials at the points and V = E_6.  The point set was picked for fp32 accuracy
(1024-channel reduction: 4e-6 abs error vs 1.4e-5 for (0, +-1, +-2)).
"""
from fractions import Fraction as Fr

import numpy as np

POINTS = {2: (Fr(0), Fr(1), Fr(-1), Fr(2), Fr(-1, 2)),
          4: (Fr(0), Fr(1), Fr(-1), Fr(2), Fr(-1, 2), Fr(1, 2), Fr(-2))}


def toom_cook(m, r, pts=None):
    pts = POINTS[m] if pts is None else pts
    n = m + r - 1
    assert len(pts) == n - 1

    def ev(k):
        rows = [[Fr(p) ** i for i in range(k)] for p in pts]
        rows.append([Fr(0)] * (k - 1) + [Fr(1)])
        return rows

    a = [row[:] + [Fr(int(i == j)) for j in range(n)] for i, row in enumerate(ev(n))]
    for c in range(n):  # exact Gauss-Jordan inverse
        piv = next(i for i in range(c, n) if a[i][c] != 0)
        a[c], a[piv] = a[piv], a[c]
        inv = 1 / a[c][c]
        a[c] = [x * inv for x in a[c]]
        for i in range(n):
            if i != c and a[i][c] != 0:
                f = a[i][c]
                a[i] = [x - f * y for x, y in zip(a[i], a[c])]
    vinv = [row[n:] for row in a]
    em = ev(m)
    at = [[em[j][i] for j in range(n)] for i in range(m)]
    bt = [[vinv[j][i] for j in range(n)] for i in range(n)]
    f = lambda mat: np.array([[float(x) for x in row] for row in mat], np.float64)  # noqa: E731
    return f(at), f(ev(r)), f(bt)


# F(6x6, 5x5): 100 products per 36 outputs (2.78 per output instead of 4); nine finite points searched for fp32 accuracy at a
# 1024-channel reduction: 3.0e-5 relative with (0, +-1, +-4/3, +-5/2, +-2/5) against 1.2e-3 for (0, +-1, +-2, +-1/2, +-3)
POINTS[6] = (Fr(0), Fr(1), Fr(-1), Fr(4, 3), Fr(-4, 3), Fr(5, 2), Fr(-5, 2), Fr(2, 5), Fr(-2, 5))
MATS = {m: toom_cook(m, 5) for m in (2, 4, 6)}   # m -> (A^T (m, m+4), G (m+4, 5), B^T (m+4, m+4))
# F(4x4, 3x3) for the encoder's stride-1 3x3 convs: 36 multiplies per 16 outputs (2.25 instead of 9 per output);
# points (0, 1, -1, 2, -1/2, inf): 3e-6 relative fp32 error at a 512-channel reduction (9e-6 for the usual (0, +-1, +-2))
MATS3 = {4: toom_cook(4, 3, (Fr(0), Fr(1), Fr(-1), Fr(2), Fr(-1, 2))),
         # F(6x6, 3x3): 64 products per 36 outputs (1.78 instead of 2.25 per output), the 8 x 8 tile of F(4x4, 5x5) with its point set:
         # 7.8e-6 relative fp32 error at a 256-channel reduction
         6: toom_cook(6, 3, (Fr(0), Fr(1), Fr(-1), Fr(2), Fr(-2), Fr(1, 2), Fr(-1, 2)))}


def mats(m, r):
    """(A^T (m, n), G (n, r), B^T (n, n)) with n = m + r - 1."""
    return MATS[m] if r == 5 else MATS3[m]


def transform_weights(wt, m):
    """[cout, cin, r, r] float64 -> U [n^2, cout, cin] float64 with U[a*n+b] = (G g G^T)[a][b], n = m + r - 1."""
    r = wt.shape[2]
    g = mats(m, r)[1]
    # one GEMM: U[(a,b), (o,c)] = sum_{r,s} G[a,r] G[b,s] g[o,c,r,s] = kron(G, G) @ W2^T
    n = m + r - 1
    u = np.kron(g, g) @ np.ascontiguousarray(wt.reshape(wt.shape[0] * wt.shape[1], r * r).T)   # [n*n, o*c] float64
    return u.reshape(n * n, wt.shape[0], wt.shape[1])
