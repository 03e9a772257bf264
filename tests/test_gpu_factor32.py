"""-m gpu: the 32 x 32 small-factor core on the matrix cores (csrc/qr_factor32.h) against numpy -- Cholesky with the inverse factor and the
modified LU of the Householder reconstruction (W - S R2 = L1 U', signs as Householder chooses them, reference qr.c:141-151) with both inverse
factors.  One wave per matrix, through the development entry points qrd_dbg_chol32 / qrd_dbg_lu32; variant 0 is the register recurrence the
routine replaces and must agree with it too."""
import ctypes as C
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def L(qr):
    """The development entry points live in the LAB library only (make lab: libmi355xqr_lab.so = the product sources + qr_factor32_dbg.hip
    with -DQR_LAB); the core they exercise, qr_factor32.h, is the one the product's panel kernels include."""
    assert os.path.exists(qr.LAB_LIB_PATH), f"{qr.LAB_LIB_PATH} is missing: make -C cuda-qr_amd lab (__graft_entry__.build() does)"
    lib = C.CDLL(qr.LAB_LIB_PATH)
    qr.check(lib.qrd_init(), "qrd_init")
    lib.qrd_dbg_chol32.restype = C.c_int
    lib.qrd_dbg_chol32.argtypes = [C.c_void_p] * 6 + [C.c_int] * 3
    lib.qrd_dbg_lu32.restype = C.c_int
    lib.qrd_dbg_lu32.argtypes = [C.c_void_p] * 8 + [C.c_int] * 3
    return lib


def _t(a):
    t = torch.from_numpy(np.ascontiguousarray(a)).cuda()
    torch.cuda.synchronize()
    return t


def run_chol(L, G, variant):
    n = G.shape[0]
    dG = _t(G)
    dR, dX = torch.zeros_like(dG), torch.zeros_like(dG)
    ok = torch.zeros(n, dtype=torch.int32, device="cuda")
    tk = torch.zeros(n, dtype=torch.int64, device="cuda")
    torch.cuda.synchronize()
    assert L.qrd_dbg_chol32(None, dG.data_ptr(), dR.data_ptr(), dX.data_ptr(), ok.data_ptr(), tk.data_ptr(), n, variant, 1) == 0
    torch.cuda.synchronize()
    return dR.cpu().numpy(), dX.cpu().numpy(), ok.cpu().numpy()


def spd_batch(rng, n, cond):
    out = np.empty((n, 32, 32))
    for q in range(n):
        U, _ = np.linalg.qr(rng.standard_normal((32, 32)))
        s = np.logspace(0, -np.log10(cond), 32) * 10.0 ** rng.uniform(-3, 3)
        out[q] = (U * s) @ U.T
        out[q] = (out[q] + out[q].T) / 2
    return out


@pytest.mark.parametrize("variant", [0, 1])
@pytest.mark.parametrize("cond", [1e1, 1e6, 1e12])
def test_chol32_matches_numpy(L, variant, cond):
    rng = np.random.default_rng(int(np.log10(cond)) + variant)
    G = spd_batch(rng, 64, cond)
    R, X, ok = run_chol(L, G, variant)
    assert ok.all()
    for q in range(G.shape[0]):
        Rref = np.linalg.cholesky(G[q]).T
        assert np.array_equal(np.tril(R[q], -1), np.zeros((32, 32))) and np.array_equal(np.triu(X[q], 1), np.zeros((32, 32)))
        # backward error of the factor and of the inverse (forward errors grow with the condition number, for numpy's factor too)
        assert np.abs(R[q].T @ R[q] - G[q]).max() <= 1e-13 * np.abs(G[q]).max()
        assert np.abs(X[q] @ R[q].T - np.eye(32)).max() <= 1e-10 * max(1.0, cond * 1e-6)          # X = R^-T
        if cond <= 1e6:
            assert np.abs(R[q] - Rref).max() <= 1e-9 * np.abs(Rref).max()


def test_chol32_refuses_indefinite_and_nan(L):
    rng = np.random.default_rng(1)
    G = spd_batch(rng, 8, 10.0)
    G[1, 5, 5] = -1.0
    G[2, 3, 3] = 0.0; G[2, 3, :] = 0.0; G[2, :, 3] = 0.0
    G[3, 0, 0] = np.nan
    G[4, 31, 31] = -G[4, 31, 31]
    G[5] = np.ones((32, 32))                      # rank one: the second pivot is exactly zero
    for variant in (0, 1):
        _, _, ok = run_chol(L, G, variant)
        assert list(ok) == [1, 0, 0, 0, 0, 0, 1, 1], (variant, ok)


def ref_lu_signed(W, R2):
    w = W.copy()
    S = np.zeros(32)
    for i in range(32):
        S[i] = -1.0 if w[i, i] >= 0 else 1.0
        w[i, i:] -= S[i] * R2[i, i:]
        w[i + 1:, i] /= w[i, i]
        w[i + 1:, i + 1:] -= np.outer(w[i + 1:, i], w[i, i + 1:])
    return w, S


@pytest.mark.parametrize("variant", [0, 1])
@pytest.mark.parametrize("e", [0.0, 1e-10, 1e-3])
def test_lu32_matches_the_reference_recurrence(L, variant, e):
    """W = the top block of an orthonormal 300 x 32 basis (what the reconstruction factors), R2 = chol(I + E)"""
    rng = np.random.default_rng(7 + variant)
    n = 48
    W = np.empty((n, 32, 32)); R2 = np.empty((n, 32, 32))
    for q in range(n):
        Q, _ = np.linalg.qr(rng.standard_normal((300, 32)))
        W[q] = Q[:32]
        E = e * rng.standard_normal((32, 32))
        R2[q] = np.linalg.cholesky(np.eye(32) + (E + E.T) / 2).T
    dW, dR2 = _t(W), _t(R2)
    dLU, dLi, dUit = torch.zeros_like(dW), torch.zeros_like(dW), torch.zeros_like(dW)
    dS = torch.zeros((n, 32), dtype=torch.float64, device="cuda")
    tk = torch.zeros(n, dtype=torch.int64, device="cuda")
    torch.cuda.synchronize()
    assert L.qrd_dbg_lu32(None, dW.data_ptr(), dR2.data_ptr(), dLU.data_ptr(), dS.data_ptr(), dLi.data_ptr(), dUit.data_ptr(), tk.data_ptr(),
                          n, variant, 1) == 0
    torch.cuda.synchronize()
    LU, S, Li, Uit = dLU.cpu().numpy(), dS.cpu().numpy(), dLi.cpu().numpy(), dUit.cpu().numpy()
    for q in range(n):
        LUr, Sr = ref_lu_signed(W[q], R2[q])
        assert np.array_equal(S[q], Sr)
        L1 = np.tril(LU[q], -1) + np.eye(32); U1 = np.triu(LU[q])
        assert np.abs(L1 @ U1 - (W[q] - Sr[:, None] * R2[q])).max() <= 1e-14
        assert np.abs(LU[q] - LUr).max() <= 1e-12 * max(1.0, np.abs(LUr).max())
        assert np.abs(np.diag(U1)).min() >= np.diag(R2[q]).min() * (1 - 1e-12)          # |pivot| >= R2(i, i): the sign choice
        assert np.array_equal(np.triu(Li[q], 1), np.zeros((32, 32))) and np.array_equal(np.diag(Li[q]), np.ones(32))
        assert np.abs(Li[q] @ L1 - np.eye(32)).max() <= 1e-12
        assert np.array_equal(np.triu(Uit[q], 1), np.zeros((32, 32)))
        assert np.abs(Uit[q].T @ U1 - np.eye(32)).max() <= 1e-12
