#!/usr/bin/env python
"""CPU: would a Winograd F(2x2,5x5) convolution (2.78x fewer MFMA products than the direct 5x5 convolution
of the pair trunk, reference network.py:57-63) hold the parity bar?  Emulates, in numpy, the direct and the
Winograd convolution with float32 products and with the shipped split-f16 products (weights pre-scaled by a
power of two, as conv_f16.h does), against a float64 direct convolution on trunk-like activations.

    python tools/winograd_numerics.py            # prints the table kept in profiles/r03_winograd_numerics.txt
"""
import numpy as np
from fractions import Fraction
np.set_printoptions(linewidth=200, precision=5, suppress=True)

def cook_toom(m, r, pts):
    n = m + r - 1
    assert len(pts) == n - 1
    AT = np.zeros((m, n)); G = np.zeros((n, r))
    for j, a in enumerate(pts):
        N = np.prod([a - b for l, b in enumerate(pts) if l != j])
        for i in range(m): AT[i, j] = a ** i
        for k in range(r): G[j, k] = a ** k / N
    AT[m - 1, n - 1] = 1.0; G[n - 1, r - 1] = 1.0
    # solve for BT: sum_j AT[i,j] G[j,k] BT[j,l] = delta(l == i+k)
    rows = []; rhs = []
    for i in range(m):
        for k in range(r):
            for l in range(n):
                row = np.zeros(n * n)
                for j in range(n): row[j * n + l] = AT[i, j] * G[j, k]
                rows.append(row); rhs.append(1.0 if l == i + k else 0.0)
    sol, res, rk, sv = np.linalg.lstsq(np.array(rows), np.array(rhs), rcond=None)
    BT = sol.reshape(n, n)
    return AT, G, BT

def split16(x):
    hi = x.astype(np.float16)
    lo = (x - hi.astype(np.float32)).astype(np.float16)
    return hi.astype(np.float32), lo.astype(np.float32)

def mm3(a, b):   # a (.., K) x b (K, N): split-f16 products, f32 accumulate; b (weights) scaled into [512, 1024)
    sc = np.float32(2.0 ** (10 - np.frexp(np.abs(b).max())[1]))
    ah, al = split16(a); bh, bl = split16(b * sc)
    return (ah @ bh + (ah @ bl + al @ bh)) / sc

def direct(x, w, mm, dt):
    # x [C][H+4][W+4] padded, w [O][C][5][5] -> [O][H][W]
    C, Hp, Wp = x.shape; H, W = Hp - 4, Wp - 4; O = w.shape[0]
    cols = np.empty((H * W, C * 25), dtype=dt)
    idx = 0
    for c in range(C):
        for u in range(5):
            for v in range(5):
                cols[:, idx] = x[c, u:u + H, v:v + W].reshape(-1); idx += 1
    return mm(cols, w.reshape(O, -1).T.astype(dt)).T.reshape(O, H, W)

def winograd(x, w, mats, mm, dt, wscale=None):
    AT, G, BT = [a.astype(dt) for a in mats]
    m = AT.shape[0]; n = BT.shape[0]
    C, Hp, Wp = x.shape; H, W = Hp - 4, Wp - 4; O = w.shape[0]
    th, tw = H // m, W // m
    # weight transform U[n][n][C][O]
    U = np.einsum('ik,ockl,jl->ijco', G, w.astype(dt), G).astype(dt)
    # input tiles d [th][tw][C][n][n]
    V = np.empty((n, n, th * tw, C), dtype=dt)
    tiles = np.empty((th, tw, C, n, n), dtype=dt)
    for a in range(th):
        for b in range(tw):
            tiles[a, b] = x[:, a * m:a * m + n, b * m:b * m + n]
    V = np.einsum('ik,abckl,jl->ijabc', BT, tiles, BT).astype(dt).reshape(n, n, th * tw, C)
    M = np.empty((n, n, th * tw, O), dtype=dt)
    for i in range(n):
        for j in range(n):
            M[i, j] = mm(V[i, j], U[i, j])
    Y = np.einsum('ik,klto,jl->toij', AT, M, AT).astype(dt)   # [t][O][m][m]
    Y = Y.reshape(th, tw, O, m, m).transpose(2, 0, 3, 1, 4).reshape(O, H, W)
    return Y

rng = np.random.default_rng(0)
C, O, H = 128, 32, 32
x = rng.standard_normal((C, H + 4, H + 4))
x = np.where(x > 0, x, np.expm1(x))          # ELU-like activations
bound = np.sqrt(6.0 / (C * 25 + 512 * 25))
w = (2 * rng.random((O, C, 5, 5)) - 1) * bound
ref = direct(x, w, lambda a, b: a @ b, np.float64)
scale = np.abs(ref).max(); rms = np.sqrt((ref ** 2).mean())
def report(name, y):
    e = np.abs(y.astype(np.float64) - ref)
    print("%-42s max err %.3e  rms err %.3e   (rel to rms out %.3e)" % (name, e.max(), np.sqrt((e**2).mean()), np.sqrt((e**2).mean()) / rms))
x32 = x.astype(np.float32); w32 = w.astype(np.float32)
report("direct f32", direct(x32, w32, lambda a, b: a @ b, np.float32))
report("direct f16x3", direct(x32, w32, mm3, np.float32))
for name, pts in (("F(2,5) pts 0,1,-1,2,-2", [0, 1, -1, 2, -2]),
                  ("F(2,5) pts 0,1,-1,1/2,-1/2", [0, 1, -1, .5, -.5]),
                  ("F(2,5) pts 0,1,-1,2,-1/2", [0, 1, -1, 2, -.5])):
    mats = cook_toom(2, 5, pts)
    chk = winograd(x, w, mats, lambda a, b: a @ b, np.float64)
    assert np.abs(chk - ref).max() < 1e-9 * scale, np.abs(chk - ref).max()
    report(name + " f32", winograd(x32, w32, mats, lambda a, b: a @ b, np.float32))
    report(name + " f16x3", winograd(x32, w32, mats, mm3, np.float32))
    print("   max |BT| %.2f  max |G| %.3f  max|AT| %.1f" % (np.abs(mats[2]).max(), np.abs(mats[1]).max(), np.abs(mats[0]).max()))
