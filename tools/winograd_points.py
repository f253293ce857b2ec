"""Rounding error of f32 Winograd convolution as a function of the interpolation points (dev tool, CPU, numpy; round 4).

Cook-Toom matrices for F(m,3) on arbitrary finite points + infinity (wincnn scaling: G carries 1 / prod(a_j - a_l)), an f32
simulation of the product's arithmetic (U = G g G^T in f64 rounded once; B^T d B, the channel sum and A^T M A in f32) on one random
3x3 layer, error against an f64 direct convolution relative to max|ref|.  Compares the tall form F(4,3) x F(2,3) the product used
since round 2, the square form F(4,3) x F(4,3) on the textbook points and a search over point sets.

    python tools/winograd_points.py [--search]
"""
import itertools
import sys

import numpy as np
from numpy.polynomial import polynomial as P


def cook_toom(m, r, pts):
    n = m + r - 1
    a = list(pts)
    assert len(a) == n - 1
    f = [np.prod([a[j] - a[l] for l in range(n - 1) if l != j]) for j in range(n - 1)]
    AT, G, BT = np.zeros((m, n)), np.zeros((n, r)), np.zeros((n, n))
    for j in range(n - 1):
        for i in range(m):
            AT[i, j] = a[j] ** i
        for k in range(r):
            G[j, k] = a[j] ** k / f[j]
        c = np.array([1.0])
        for l in range(n - 1):
            if l != j:
                c = P.polymul(c, [-a[l], 1.0])
        BT[j, :len(c)] = c
    AT[m - 1, n - 1] = 1.0
    G[n - 1, r - 1] = 1.0
    c = np.array([1.0])
    for l in range(n - 1):
        c = P.polymul(c, [-a[l], 1.0])
    BT[n - 1, :len(c)] = c
    return AT, G, BT


class Layer:
    def __init__(self, C=128, Co=16, H=8, W=16, seed=1):
        rng = np.random.default_rng(seed)
        self.C, self.Co, self.H, self.W = C, Co, H, W
        x = np.maximum(rng.normal(size=(C, H + 2, W + 2)), 0)   # post-ReLU activations, zero border = the convolution's padding
        x[:, 0, :] = 0; x[:, -1, :] = 0; x[:, :, 0] = 0; x[:, :, -1] = 0
        self.x = x
        self.g = rng.normal(size=(Co, C, 3, 3)) * (2.0 / (C * 9)) ** 0.5
        cols = np.stack([x[:, i:i + H, j:j + W] for i in range(3) for j in range(3)], 0)
        self.ref = np.einsum('kchw,ock->ohw', cols, self.g.reshape(Co, C, 9), optimize=True)
        A = cols.transpose(2, 3, 0, 1).reshape(H * W, 9 * C).astype(np.float32)
        B = self.g.reshape(Co, C, 9).transpose(0, 2, 1).reshape(Co, 9 * C).astype(np.float32)
        self.direct_f32 = (A @ B.T).T.reshape(Co, H, W)
        self.scale = np.abs(self.ref).max()

    def err(self, out):
        return np.abs(out.astype(np.float64) - self.ref).max() / self.scale

    def wino(self, mh, ph, mw, pw):
        f = np.float32
        ATh, Gh, BTh = cook_toom(mh, 3, ph)
        ATw, Gw, BTw = cook_toom(mw, 3, pw)
        nh, nw = mh + 2, mw + 2
        U = np.einsum('ik,ockl,jl->ijoc', Gh, self.g, Gw).astype(f)
        BTh_, BTw_, ATh_, ATw_ = BTh.astype(f), BTw.astype(f), ATh.astype(f), ATw.astype(f)
        out = np.zeros((self.Co, self.H, self.W), dtype=f)
        xf = self.x.astype(f)
        for th in range(0, self.H, mh):
            for tw in range(0, self.W, mw):
                d = xf[:, th:th + nh, tw:tw + nw]
                V = np.einsum('ik,ckl->cil', BTh_, d).astype(f)
                V = np.einsum('cil,jl->cij', V, BTw_).astype(f)
                M = np.einsum('cij,ijoc->oij', V, U).astype(f)
                Y = np.einsum('ai,oij->oaj', ATh_, M).astype(f)
                Y = np.einsum('oaj,bj->oab', Y, ATw_).astype(f)
                out[:, th:th + mh, tw:tw + mw] = Y
        return self.err(out)


def main():
    L = Layer()
    std, f23 = [0, 1, -1, 2, -2], [0, 1, -1]
    print(f"layer C={L.C} Cout={L.Co} {L.H}x{L.W}, errors relative to max|f64 result|")
    print(f"direct f32                                   {L.err(L.direct_f32):.3e}")
    print(f"tall F(4,3)xF(2,3), points 0 +-1 +-2         {L.wino(4, std, 2, f23):.3e}   <- the product since round 2")
    print(f"square F(4,3)^2,    points 0 +-1 +-2         {L.wino(4, std, 4, std):.3e}")
    for name, pts in (("0 +-3/2 +-2/3", [0, 1.5, -1.5, 2 / 3, -2 / 3]), ("0 +-1/2 +-2", [0, 0.5, -0.5, 2, -2]), ("0 +-1 2 -1/2", [0, 1, -1, 2, -0.5]),
                      ("0 +-1 +-1/2", [0, 1, -1, 0.5, -0.5])):
        print(f"square F(4,3)^2,    points {name:<18}{L.wino(4, pts, 4, pts):.3e}   tall with them on H: {L.wino(4, pts, 2, f23):.3e}")
    if "--search" in sys.argv:
        cands = [1, -1, 0.5, -0.5, 2, -2, 1.5, -1.5, 2 / 3, -2 / 3, 0.75, -0.75, 4 / 3, -4 / 3]
        res = sorted((L.wino(4, [0] + list(c), 4, [0] + list(c)), [0] + list(c)) for c in itertools.combinations(cands, 4))
        for e2, pts in res[:10]:
            print(f"search: square {e2:.3e}  {[round(p, 4) for p in pts]}")


if __name__ == "__main__":
    main()
