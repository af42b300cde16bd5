"""Accuracy gate for Winograd F(4x4, 3x3) -- CPU emulation, no GPU needed.

    python tools/study/wino_f4_accuracy.py [quick]

Emulates in float32, with the accumulation order of the fp32 matrix-core kernels (sequential over input channels in steps of 4, one fp32
accumulator per output), three ways of computing a 3x3 / stride 1 / pad 1 convolution and measures each against float64:
  direct      sum over (tap, channel) in the igemm kernels' order (channel fastest), K = 9 Cin accumulation steps
  F(2x2,3x3)  what conv_wino.hip runs: V = B^T d B, U = G g G^T (float64, rounded once: the filter transform runs once per step), 16 position GEMMs,
              Y = A^T M A
  F(4x4,3x3)  the same with 6x6 tiles, for several interpolation point sets (Lavin's 0, +-1, +-2, inf; 0, +-1, +-1/2, inf; mixed sets from the
              error-analysis literature)
Error = max |y - y64| / max |y64| (the bar of tests/test_gpu_winograd.py: 2e-5; the direct tiles sit at 0.7-3.4e-6, F(2x2) at 2-9e-7 on the GPU).
The gate of the verdict: adopt F(4x4) only where its error is at or below the direct tile's on the shapes of tests/test_gpu_winograd.py.
"""
import sys
from fractions import Fraction as Fr
import numpy as np


def cook_toom(points, m, r):
    """A^T (m x n), G (n x r), B^T (n x n) for F(m, r) with the finite `points` + infinity, n = m + r - 1 (wincnn's construction, exact rationals)"""
    n = m + r - 1
    assert len(points) == n - 1
    p = [Fr(x) for x in points]
    AT = [[(p[j] ** i if j < n - 1 else (Fr(1) if i == m - 1 else Fr(0))) for j in range(n)] for i in range(m)]
    N = [Fr(1)] * (n - 1)
    for j in range(n - 1):
        for l in range(n - 1):
            if l != j:
                N[j] *= (p[j] - p[l])
    G = [[p[j] ** k / N[j] for k in range(r)] for j in range(n - 1)] + [[Fr(0)] * (r - 1) + [Fr(1)]]
    # B^T from the identity  sum_j AT[i][j] G[j][k] BT[j][q] = [q == i + k]  (solved in float64, then snapped to rationals with small denominators)
    rows, rhs = [], []
    for i in range(m):
        for k in range(r):
            for q in range(n):
                row = np.zeros(n * n)
                for j in range(n):
                    row[j * n + q] = float(AT[i][j] * G[j][k])
                rows.append(row); rhs.append(1.0 if q == i + k else 0.0)
    sol, res, rank, _ = np.linalg.lstsq(np.array(rows), np.array(rhs), rcond=None)
    BT = sol.reshape(n, n)
    assert np.abs(np.array(rows) @ sol - np.array(rhs)).max() < 1e-9, "no exact B^T for these points"
    f = lambda M: np.array([[float(v) for v in row] for row in M])
    return f(AT), f(G), BT


def conv_ref64(x, w):
    n, h, wd, c = x.shape
    co = w.shape[0]
    xp = np.zeros((n, h + 2, wd + 2, c)); xp[:, 1:-1, 1:-1] = x
    y = np.zeros((n, h, wd, co))
    for r in range(3):
        for s in range(3):
            y += np.einsum("nhwc,oc->nhwo", xp[:, r:r + h, s:s + wd], w[:, :, r, s].astype(np.float64))
    return y


def seq_gemm32(a, b):
    """a [M, K] x b [N, K] -> [M, N] in float32 with one accumulator per output, K walked in steps of 4 (v_mfma_f32_16x16x4 / 32x32x2 pairs)"""
    M, K = a.shape
    acc = np.zeros((M, b.shape[0]), np.float32)
    for k in range(0, K, 4):
        acc = (acc + a[:, k:k + 4] @ b[:, k:k + 4].T).astype(np.float32)
    return acc


def conv_direct32(x, w):
    n, h, wd, c = x.shape
    co = w.shape[0]
    xp = np.zeros((n, h + 2, wd + 2, c), np.float32); xp[:, 1:-1, 1:-1] = x
    a = np.concatenate([xp[:, r:r + h, s:s + wd].reshape(-1, c) for r in range(3) for s in range(3)], axis=1)      # k = (r, s, ci)
    b = np.concatenate([w[:, :, r, s] for r in range(3) for s in range(3)], axis=1).astype(np.float32)
    return seq_gemm32(a, b).reshape(n, h, wd, co)


def conv_wino32(x, w, m, AT, G, BT):
    """F(m x m, 3 x 3) in float32: input / output transforms in float32 (as the kernel's VALU would), U from float64 rounded once"""
    n, h, wd, c = x.shape
    co = w.shape[0]
    t = m + 2
    th, tw = -(-h // m), -(-wd // m)
    xp = np.zeros((n, th * m + 2, tw * m + 2, c), np.float32); xp[:, 1:h + 1, 1:wd + 1] = x
    U = np.einsum("ij,ocjk,lk->iloc", G, w.astype(np.float64), G).astype(np.float32)                # [t][t][co][c]
    BT32, AT32 = BT.astype(np.float32), AT.astype(np.float32)
    tiles = np.stack([xp[:, i * m:i * m + t, j * m:j * m + t] for i in range(th) for j in range(tw)], axis=1)     # [n][T][t][t][c]
    # B^T d B with float32 rounding after each 1-D pass
    V = np.einsum("ij,ntjkc->ntikc", BT32, tiles).astype(np.float32)
    V = np.einsum("ntikc,lk->ntilc", V, BT32).astype(np.float32)
    Mx = np.zeros((n, th * tw, t, t, co), np.float32)
    for i in range(t):
        for j in range(t):
            Mx[:, :, i, j] = seq_gemm32(V[:, :, i, j].reshape(-1, c), U[i, j]).reshape(n, th * tw, co)
    Y = np.einsum("ij,ntjko->ntiko", AT32, Mx).astype(np.float32)
    Y = np.einsum("ntiko,lk->ntilo", Y, AT32).astype(np.float32)
    y = np.zeros((n, th * m, tw * m, co), np.float32)
    for i in range(th):
        for j in range(tw):
            y[:, i * m:(i + 1) * m, j * m:(j + 1) * m] = Y[:, i * tw + j]
    return y[:, :h, :wd]


def main():
    quick = len(sys.argv) > 1 and sys.argv[1] == "quick"
    shapes = [(2, 56, 48, 32), (2, 28, 64, 32), (2, 56, 128, 32), (2, 28, 256, 32), (2, 14, 512, 32)]
    if quick:
        shapes = [(1, 28, 64, 16), (1, 14, 256, 16)]
    F2 = cook_toom([0, 1, -1], 2, 3)
    sets = {"F4 0,+-1,+-2,inf (Lavin)": [0, 1, -1, 2, -2], "F4 0,+-1,+-1/2,inf": [0, 1, -1, Fr(1, 2), Fr(-1, 2)],
            "F4 0,+-1,1/2,-2,inf": [0, 1, -1, Fr(1, 2), -2], "F4 0,+-1/2,+-3/2... (0,+-1/2,+-1 scaled): 0,+-1/2,+-2": [0, Fr(1, 2), Fr(-1, 2), 2, -2],
            "F4 0,1,-1,2,-1/2,inf": [0, 1, -1, 2, Fr(-1, 2)]}
    F4 = {k: cook_toom(v, 4, 3) for k, v in sets.items()}
    print("shape (n, hw, cin, cout): error = max|y - y64| / max|y64|")
    for (n, hw, cin, co) in shapes:
        g = np.random.default_rng(cin * 7 + hw)
        x = g.standard_normal((n, hw, hw, cin)).astype(np.float32)
        w = (g.standard_normal((co, cin, 3, 3)) / (3 * cin ** 0.5)).astype(np.float32)
        y64 = conv_ref64(x.astype(np.float64), w)
        sc = np.abs(y64).max()
        e = lambda y: float(np.abs(y - y64).max() / sc)
        row = {"direct": e(conv_direct32(x, w)), "F(2x2,3x3)": e(conv_wino32(x, w, 2, *F2))}
        for k, mats in F4.items():
            row[k] = e(conv_wino32(x, w, 4, *mats))
        print(f"  ({n}, {hw}, {cin}, {co}): " + "; ".join(f"{k} {v:.2e}" for k, v in row.items()), flush=True)


if __name__ == "__main__":
    main()
