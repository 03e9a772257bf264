"""Lane-level numpy model of the micro-blocked 32 x 32 factor routines of qr_factor32.h (one wave, v_mfma_f64_16x16x4_f64).

Written BEFORE the HIP code: there is no GPU in the development container, so the data flow (which register of which accumulator tile
is which 4-row strip, which operand role transposes, what has to be masked) is debugged here against numpy.linalg, then transliterated.

MFMA model (the layout every kernel of this library relies on, e.g. cq_tile in qr_panel_cqr.hip):
    lane = l15 + 16 * l4;   A operand: A[i = l15][k = l4];   B operand: B[k = l4][j = l15];   D register r: D[i = 4 r + l4][j = l15]
Consequence used throughout: register r of a 16 x 16 accumulator tile is the 4 x 16 strip of rows 4r .. 4r+3 in B-operand layout, and the
SAME register read as an A operand is that strip transposed.
"""
import numpy as np

LANES = np.arange(64)
L15 = LANES & 15
L4 = LANES >> 4


def mfma(a, b, c):
    """a, b: (64,) operands; c: (4, 64) accumulator -> new (4, 64) accumulator"""
    A = np.zeros((16, 4)); B = np.zeros((4, 16))
    A[L15, L4] = a
    B[L4, L15] = b
    D = A @ B
    out = c.copy()
    for r in range(4):
        out[r] += D[4 * r + L4, L15]
    return out


def tile_load(M, i0, j0):
    """16 x 16 block of M at (i0, j0) into accumulator layout (4, 64)"""
    t = np.zeros((4, 64))
    for r in range(4):
        t[r] = M[i0 + 4 * r + L4, j0 + L15]
    return t


def readlane(v, lane):
    return v[lane]


def chol4_inv(d):
    """d: 4 x 4 symmetric (upper read).  Returns M = U^-T (lower) with d = U^T U, and ok"""
    ok = True
    u = np.zeros((4, 4)); inv = np.zeros(4)
    w = np.array(d, dtype=float)
    for k in range(4):
        p = w[k, k]
        ok = ok and (p > 0)
        inv[k] = 1.0 / np.sqrt(p) if p > 0 else 0.0
        for j in range(k, 4):
            u[k, j] = w[k, j] * inv[k]
        for i in range(k + 1, 4):
            for j in range(i, 4):
                w[i, j] -= u[k, i] * u[k, j]
    # M = U^-T: forward substitution by rows, M lower triangular
    M = np.zeros((4, 4))
    for i in range(4):
        M[i, i] = inv[i]
        for j in range(i):
            s = 0.0
            for k in range(j, i):
                s += u[k, i] * M[k, j]
            M[i, j] = -s * inv[i]
    return M, ok


def chol4_u(d):
    w = np.array(d, dtype=float); u = np.zeros((4, 4))
    for k in range(4):
        inv = 1.0 / np.sqrt(w[k, k])
        for j in range(k, 4):
            u[k, j] = w[k, j] * inv
        for i in range(k + 1, 4):
            for j in range(i, 4):
                w[i, j] -= u[k, i] * u[k, j]
    return u


def chol32_aug(G, udiag=False):
    """G 32 x 32 SPD -> R (upper, G = R^T R), X = R^-T (lower), ok.  One wave; tiles T00, T01, T11 and aug X00, X10, X11."""
    T = {(0, 0): tile_load(G, 0, 0), (0, 1): tile_load(G, 0, 16), (1, 1): tile_load(G, 16, 16)}
    I = np.eye(32)
    X = {(0, 0): tile_load(I, 0, 0), (1, 0): tile_load(I, 16, 0), (1, 1): tile_load(I, 16, 16)}
    Rout = np.zeros((32, 32)); Xout = np.zeros((32, 32))
    ok = True
    zero = np.zeros((4, 64))
    for s in range(8):
        K, ti, r = 4 * s, s // 4, s % 4
        dreg = T[(ti, ti)][r]
        d = np.zeros((4, 4))
        for i in range(4):
            for j in range(i, 4):
                d[i, j] = readlane(dreg, 16 * i + 4 * r + j)
        M, okk = chol4_inv(d)
        ok = ok and okk
        U4 = np.linalg.inv(M).T if False else None
        Amat = np.where((L15 < 4) & (L4 <= L15), M[np.minimum(L15, 3), L4], 0.0)
        S = {}
        for b in range(ti, 2):
            z = mfma(Amat, T[(ti, b)][r], zero)[0]
            col = 16 * b + L15; row = K + L4
            S[b] = np.where(col >= row, z, 0.0)
            if udiag and b == ti:
                # the pivot block of R from the uniform factorisation (exactly upper triangular, consistent with M = U4^-T)
                U4 = chol4_u(d)
                inblk = (col >= K) & (col < K + 4)
                S[b] = np.where(inblk, U4[L4, np.clip(col - K, 0, 3)], S[b])
                S[b] = np.where(col >= row, S[b], 0.0)
            Rout[row, col] = S[b]
        Y = {}
        for b in range(0, ti + 1):
            Y[b] = mfma(Amat, X[(ti, b)][r], zero)[0]
            Xout[K + L4, 16 * b + L15] = Y[b]
        for a in range(ti, 2):
            for b in range(a, 2):
                T[(a, b)] = mfma(-S[a], S[b], T[(a, b)])
            for b in range(0, ti + 1):
                X[(a, b)] = mfma(-S[a], Y[b], X[(a, b)])
    return Rout, Xout, ok


def lu4_signed(w, r2):
    """modified LU of the 4 x 4 block w - diag(S) r2 = L U with S_i = -sign of the current (i, i) entry.
    Returns L (unit lower), U (upper), S, Linv = L^-1, UinvT = U^-T (lower)"""
    w = np.array(w, dtype=float)
    S = np.zeros(4)
    for i in range(4):
        x = w[i, i]
        S[i] = -1.0 if x >= 0 else 1.0
        for j in range(i, 4):
            w[i, j] -= S[i] * r2[i, j]
        inv = 1.0 / w[i, i]
        for jp in range(i + 1, 4):
            l = w[jp, i] * inv
            w[jp, i] = l
            for c in range(i + 1, 4):
                w[jp, c] -= l * w[i, c]
    Lm = np.tril(w, -1) + np.eye(4)
    U = np.triu(w)
    Linv = np.linalg.inv(Lm)          # (in the kernel: 6 entries by forward substitution)
    UinvT = np.linalg.inv(U).T        # (in the kernel: 10 entries)
    return Lm, U, S, Linv, UinvT


def lu32_aug(W, R2):
    """W - S R2 = L1 U' (Householder sign choice).  Returns LU (L1 strictly below, U' on/above), S, L1inv (unit lower), UinvT (= U'^-T, lower)"""
    Wt = {(a, b): tile_load(W, 16 * a, 16 * b) for a in range(2) for b in range(2)}
    WT = W.T.copy()
    Vt = {(a, b): tile_load(WT, 16 * a, 16 * b) for a in range(2) for b in range(2)}       # Vt[a][b][i][j] = W(16 b + j, 16 a + i)
    R2t = {(a, b): tile_load(np.triu(R2), 16 * a, 16 * b) for a in range(2) for b in range(2)}
    I = np.eye(32)
    XL = {(0, 0): tile_load(I, 0, 0), (1, 0): tile_load(I, 16, 0), (1, 1): tile_load(I, 16, 16)}
    XU = {(0, 0): tile_load(I, 0, 0), (1, 0): tile_load(I, 16, 0), (1, 1): tile_load(I, 16, 16)}
    LU = np.zeros((32, 32)); Sv = np.zeros(32); Linv = np.zeros((32, 32)); UinvT = np.zeros((32, 32))
    zero = np.zeros((4, 64))
    for s in range(8):
        K, ti, r = 4 * s, s // 4, s % 4
        wreg, rreg = Wt[(ti, ti)][r], R2t[(ti, ti)][r]
        w = np.zeros((4, 4)); r2 = np.zeros((4, 4))
        for i in range(4):
            for j in range(4):
                w[i, j] = readlane(wreg, 16 * i + 4 * r + j)
                if j >= i:
                    r2[i, j] = readlane(rreg, 16 * i + 4 * r + j)
        L11, U11, S4, Li, UiT = lu4_signed(w, r2)
        Sv[K:K + 4] = S4
        AL = np.where((L15 < 4) & (L4 <= L15), Li[np.minimum(L15, 3), L4], 0.0)
        AU = np.where((L15 < 4) & (L4 <= L15), UiT[np.minimum(L15, 3), L4], 0.0)
        Ssel = S4[L4]
        U = {}; Lt = {}
        for b in range(ti, 2):
            zb = Wt[(ti, b)][r] - Ssel * R2t[(ti, b)][r]
            z = mfma(AL, zb, zero)[0]
            col = 16 * b + L15; row = K + L4
            U[b] = np.where(col >= row, z, 0.0)
            m = col >= row
            LU[row[m], col[m]] = U[b][m]
            z = mfma(AU, Vt[(ti, b)][r], zero)[0]          # z[lane (k, l15)] = L(16 b + l15, K + k)
            rowi = 16 * b + L15
            Lt[b] = np.where(rowi >= K + 4, z, 0.0)
            m = rowi >= K + 4
            LU[rowi[m], (K + L4)[m]] = Lt[b][m]
        for j in range(4):
            for i in range(j):
                LU[K + j, K + i] = L11[j, i]
        YL = {}; YU = {}
        for b in range(0, ti + 1):
            YL[b] = mfma(AL, XL[(ti, b)][r], zero)[0]
            Linv[K + L4, 16 * b + L15] = YL[b]
            YU[b] = mfma(AU, XU[(ti, b)][r], zero)[0]
            UinvT[K + L4, 16 * b + L15] = YU[b]
        for a in range(ti, 2):
            for b in range(ti, 2):
                Wt[(a, b)] = mfma(-Lt[a], U[b], Wt[(a, b)])
                Vt[(a, b)] = mfma(-U[a], Lt[b], Vt[(a, b)])
            for b in range(0, ti + 1):
                XL[(a, b)] = mfma(-Lt[a], YL[b], XL[(a, b)])
                XU[(a, b)] = mfma(-U[a], YU[b], XU[(a, b)])
    return LU, Sv, Linv, UinvT


def ref_lu_signed(W, R2):
    w = W.copy(); S = np.zeros(32)
    for i in range(32):
        S[i] = -1.0 if w[i, i] >= 0 else 1.0
        w[i, i:] -= S[i] * R2[i, i:]
        w[i + 1:, i] /= w[i, i]
        w[i + 1:, i + 1:] -= np.outer(w[i + 1:, i], w[i, i + 1:])
    return w, S


if __name__ == "__main__":
    rng = np.random.default_rng(5)
    A = rng.standard_normal((200, 32))
    G = A.T @ A
    R, X, ok = chol32_aug(G)
    Rref = np.linalg.cholesky(G).T
    print("chol ok", ok, "|R - Rref|", np.abs(R - Rref).max(), "|X - R^-T|", np.abs(X - np.linalg.inv(Rref).T).max(),
          "lower(R)", np.abs(np.tril(R, -1)).max(), "upper(X)", np.abs(np.triu(X, 1)).max())
    Q, _ = np.linalg.qr(rng.standard_normal((300, 32)))
    E = 1e-3 * rng.standard_normal((32, 32)); G2 = np.eye(32) + (E + E.T) / 2
    R2 = np.linalg.cholesky(G2).T
    W = Q[:32] @ np.linalg.inv(R2) @ R2        # any 32 x 32: the LU does not care where W comes from
    W = Q[:32]
    LU, S, Li, UiT = lu32_aug(W, R2)
    LUr, Sr = ref_lu_signed(W, R2)
    L1 = np.tril(LUr, -1) + np.eye(32); U1 = np.triu(LUr)
    print("lu  |LU - ref|", np.abs(LU - LUr).max(), "S equal", np.array_equal(S, Sr), "|Linv - L^-1|", np.abs(Li - np.linalg.inv(L1)).max(),
          "|UinvT - U^-T|", np.abs(UiT - np.linalg.inv(U1).T).max(), "resid", np.abs(L1 @ U1 - (W - S[:, None] * np.triu(R2))).max())
