"""-m gpu: the one-launch outer panel (qr_panel_fused.hip) against numpy -- a whole panel of up to 256 columns (8 leaves of 32:
CholeskyQR2 + Householder reconstruction per leaf, the in-panel products and updates, all inside ONE launch whose workgroups exchange
32 x 32 partial matrices through write-through slabs).  Checked as a compact-WY panel: explicit V and in-place tails agree, V is unit
lower trapezoidal, every leaf's T block is what V and tau imply, the Gram blocks are V_prev^T V_l, and (I - V T V^T)^T P = [R; 0]
with T merged from the leaves' blocks and the Gram blocks the way the host's merge tree does it."""
import ctypes as C

import numpy as np
import pytest
import torch

from gpu_util import dev, host, zeros

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def q(qr):
    qr.check(qr.lib.qrd_init(), "qrd_init")
    f = qr.lib.qrd_panel_fused
    f.restype = C.c_int
    f.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_int,
                  C.c_void_p, C.POINTER(C.c_uint), C.c_void_p]
    fr = qr.lib.qrd_panel_fused_rows                         # rows per row workgroup given: 128 (round 6) / 256 (round 4-5) / 0 = the library's choice
    fr.restype = C.c_int
    fr.argtypes = f.argtypes + [C.c_int]
    qr.lib.qrd_panel_fused_ws_doubles.restype = C.c_size_t
    return qr


class Ws:
    """one exchange workspace + epoch counter, as a plan keeps them"""

    def __init__(self, q):
        self.buf = torch.zeros(int(q.lib.qrd_panel_fused_ws_doubles()), dtype=torch.float64, device="cuda")
        self.epoch = C.c_uint(0)
        self.status = torch.zeros(4, dtype=torch.int32, device="cuda")
        torch.cuda.synchronize()


def merged_t(V, Tdiag, wh):
    """T of the whole panel from the leaves' diagonal blocks: T(0:c, c:c+32) = -T(0:c, 0:c) (V(:, 0:c)^T V(:, c:c+32)) T_l"""
    T = np.zeros((wh, wh))
    for c in range(0, wh, 32):
        T[c:c + 32, c:c + 32] = np.triu(Tdiag[c:c + 32, c:c + 32])
        if c:
            T[:c, c:c + 32] = -T[:c, :c] @ (V[:, :c].T @ V[:, c:c + 32]) @ T[c:c + 32, c:c + 32]
    return T


def run_panel(q, ws, P, lda=None, ldv=None, rows=0):
    mk, wh = P.shape
    lda = lda or mk
    ldv = ldv or mk
    buf = np.full((lda, wh), 7.0)
    buf[:mk] = P
    dA = dev(buf)
    dV = zeros(ldv, wh)
    dT, dtau, dG = dev(np.full((wh, wh), np.nan)), zeros(wh, 1), dev(np.full((wh, wh), np.nan))
    ws.status.zero_()
    torch.cuda.synchronize()
    rc = q.lib.qrd_panel_fused_rows(None, dA.data_ptr(), lda, mk, wh, dtau.data_ptr(), dT.data_ptr(), wh, dV.data_ptr(), ldv, dG.data_ptr(), wh,
                                    ws.buf.data_ptr(), C.byref(ws.epoch), ws.status.data_ptr(), rows)
    assert rc == 0, rc
    q.check(q.lib.qrd_device_sync(), "sync")
    st = ws.status.cpu().numpy()
    out = host(dA)
    assert np.array_equal(out[mk:], buf[mk:]), "rows below the panel are never written"
    return out[:mk], host(dV)[:mk], host(dT), host(dtau)[:, 0], host(dG), st


def check_panel(P, out, V, Tdiag, tau, G, tol=1e-12):
    mk, wh = P.shape
    assert np.isfinite(out).all() and np.isfinite(V).all() and np.isfinite(tau).all()
    assert np.array_equal(np.triu(V[:wh], 1), np.zeros((wh, wh))) and np.array_equal(np.diag(V[:wh]), np.ones(wh))
    assert np.array_equal(np.tril(V, -1), np.tril(out, -1)), "explicit V and in-place tails must agree"
    for c in range(0, wh, 32):
        Tl = Tdiag[c:c + 32, c:c + 32]
        assert np.isfinite(Tl).all() and np.array_equal(np.tril(Tl, -1), np.zeros((32, 32)))
        assert np.array_equal(np.diag(Tl), tau[c:c + 32])
        Vl = V[:, c:c + 32]
        # T_l^-1 + T_l^-T = V_l^T V_l  (compact-WY identity)
        Ti = np.linalg.inv(Tl)
        assert np.abs(Ti + Ti.T - Vl.T @ Vl).max() < 1e-11 * max(1.0, np.abs(Ti).max())
        if c:
            Gref = V[:, :c].T @ Vl
            assert np.abs(G[:c, c:c + 32] - Gref).max() < tol * np.sqrt(mk) * max(1.0, np.abs(Gref).max())
    T = merged_t(V, Tdiag, wh)
    if wh == 64:            # round 6: a 64-column launch that takes the Gram block along merges its two T blocks itself
        assert np.abs(Tdiag[:32, 32:] - T[:32, 32:]).max() < 1e-12 * max(1.0, np.abs(T).max()), "in-launch T merge"
    QtP = P - V @ (T.T @ (V.T @ P))
    scale = np.abs(P).max()
    assert np.abs(np.tril(QtP, -1)).max() < 1e-11 * np.sqrt(mk) * scale
    assert np.abs(np.triu(QtP[:wh]) - np.triu(out[:wh])).max() < 1e-11 * np.sqrt(mk) * scale
    R = np.triu(out[:wh])
    Rref = np.linalg.qr(P, mode="r")
    S = np.sign(np.diag(R)) * np.sign(np.diag(Rref))
    assert np.linalg.norm(S[:, None] * R - Rref) / np.linalg.norm(Rref) < 1e-12


@pytest.mark.parametrize("rows", [128, 256])
@pytest.mark.parametrize("mk,wh", [(256, 32), (256, 64), (512, 64), (1024, 128), (1000, 256), (2048, 256), (4096, 64), (4096, 256),
                                   (8192, 256), (8192, 128), (5000, 96), (260, 256), (3968, 256), (132, 128)])
def test_panel_fused_well_conditioned(q, mk, wh, rows):
    ws = Ws(q)
    P = np.random.default_rng(mk + wh).random((mk, wh))
    out, V, T, tau, G, st = run_panel(q, ws, P, lda=mk + 6, ldv=mk + 2, rows=rows)
    assert st[1] == 0, "a wait timed out"
    assert st[0] == 0, "a leaf of a well-conditioned panel took the Householder route"
    check_panel(P, out, V, T, tau, G)


@pytest.mark.parametrize("mk,wh", [(12288, 256), (16384, 256), (16384, 128), (16380, 64), (10000, 96), (8196, 256)])
def test_panel_fused_beyond_8192_rows(q, mk, wh):
    """End of round 6: up to 64 row workgroups of 256 rows = 16384 rows in one launch (the limit of 8192 was round 4's workspace)."""
    ws = Ws(q)
    P = np.random.default_rng(mk + wh).random((mk, wh))
    out, V, T, tau, G, st = run_panel(q, ws, P, lda=mk + 6, ldv=mk + 2, rows=0)
    assert st[1] == 0, "a wait timed out"
    assert st[0] == 0, "a leaf of a well-conditioned panel took the Householder route"
    check_panel(P, out, V, T, tau, G)
    P2 = P.copy(); P2[:, 40] = P2[:, 3]                     # a dependent column: the Householder-route leaf at this height
    out, V, T, tau, G, st = run_panel(q, ws, P2, rows=0)
    assert st[1] == 0 and st[0] >= 1
    _check_wy_only(P2, out, V, T, tau)


def test_panel_fused_repeated_launches_share_a_workspace(q):
    """the epoch words are never reset: several panels of different heights through one workspace, results bitwise reproducible"""
    ws = Ws(q)
    rng = np.random.default_rng(5)
    first = {}
    for rep in range(3):
        for mk, wh, rows in [(2048, 128, 128), (4096, 256, 256), (512, 32, 0), (4096, 256, 128)]:      # both row splits through the same epoch words
            P = np.random.default_rng(mk).random((mk, wh))
            out, V, T, tau, G, st = run_panel(q, ws, P, rows=rows)
            assert st[0] == 0 and st[1] == 0
            if rep == 0:
                check_panel(P, out, V, T, tau, G)
                first[(mk, wh, rows)] = out
            else:
                assert np.array_equal(out, first[(mk, wh, rows)])


def test_panel_fused_declines_what_it_cannot_take(q):
    ws = Ws(q)
    d = zeros(17000, 64)
    args = lambda mk, wh, lda: (None, d.data_ptr(), lda, mk, wh, d.data_ptr(), d.data_ptr(), wh, d.data_ptr(), lda, d.data_ptr(), wh,
                                ws.buf.data_ptr(), C.byref(ws.epoch), ws.status.data_ptr())
    assert q.lib.qrd_panel_fused(*args(16388, 64, 17000)) == -7    # more than 16384 rows
    assert q.lib.qrd_panel_fused(*args(1026, 64, 17000)) == -7      # rows not a multiple of 4
    assert q.lib.qrd_panel_fused(*args(1024, 48, 17000)) == -7      # not whole leaves
    assert q.lib.qrd_panel_fused(*args(1024, 64, 16999)) == -7      # odd leading dimension


def _check_wy_only(P, out, V, Tdiag, tau, tol=1e-11):
    """compact-WY consistency without comparing R with LAPACK (rank-deficient panels: R is not unique)"""
    mk, wh = P.shape
    assert np.isfinite(out).all() and np.isfinite(V).all() and np.isfinite(tau).all()
    assert all(np.isfinite(Tdiag[c:c + 32, c:c + 32]).all() for c in range(0, wh, 32))
    assert np.array_equal(np.triu(V[:wh], 1), np.zeros((wh, wh))) and np.array_equal(np.diag(V[:wh]), np.ones(wh))
    assert np.array_equal(np.tril(V, -1), np.tril(out, -1))
    T = merged_t(V, Tdiag, wh)
    if wh == 64:
        assert np.abs(Tdiag[:32, 32:] - T[:32, 32:]).max() < 1e-11 * max(1.0, np.abs(T).max()), "in-launch T merge (Householder-route leaves included)"
    QtP = P - V @ (T.T @ (V.T @ P))
    cs = np.maximum(np.abs(P).max(axis=0), 1e-300)
    assert (np.abs(np.tril(QtP, -1)) / cs).max() < tol * np.sqrt(mk)
    assert (np.abs(np.triu(QtP[:wh]) - np.triu(out[:wh])) / cs).max() < tol * np.sqrt(mk)
    H = np.eye(mk) - V @ T @ V.T if mk <= 2048 else None
    if H is not None:
        assert np.abs(H.T @ H - np.eye(mk)).max() < 1e-12


@pytest.mark.parametrize("rows", [128, 256])
@pytest.mark.parametrize("kind", ["zero_column", "dependent", "cond1e10", "two_bad_leaves", "all_zero"])
@pytest.mark.parametrize("mk,wh", [(2048, 128), (512, 64), (4096, 256)])
def test_panel_fused_householder_route(q, kind, mk, wh, rows):
    """leaves the CholeskyQR2 route must refuse (zero / dependent columns, cond 1e10): the launch switches to its in-kernel Householder
    route for exactly those leaves (status counts them) and the panel is a valid compact-WY panel all the same"""
    ws = Ws(q)
    rng = np.random.default_rng(mk + wh + len(kind))
    P = rng.random((mk, wh))
    bad = 1
    if kind == "zero_column":
        P[:, 5] = 0.0
    elif kind == "dependent":
        P[:, wh - 3] = P[:, wh - 20]                       # last leaf
    elif kind == "cond1e10":
        U, _ = np.linalg.qr(rng.standard_normal((mk, 32))); Vr, _ = np.linalg.qr(rng.standard_normal((32, 32)))
        P[:, 32:64] = (U * np.logspace(0, -10, 32)) @ Vr.T
    elif kind == "two_bad_leaves":
        P[:, 7] = P[:, 3]
        P[:, 40] = 0.0
        bad = 2
    else:
        P[:, :32] = 0.0
    out, V, T, tau, G, st = run_panel(q, ws, P, rows=rows)
    assert st[1] == 0
    assert st[0] == bad, st
    _check_wy_only(P, out, V, T, tau, tol=1e-10 if kind == "cond1e10" else 1e-11)
    # a second launch through the same workspace (the epoch words moved by the extra exchanges) still works
    P2 = rng.random((mk, wh))
    out, V, T, tau, G, st = run_panel(q, ws, P2, rows=rows)
    assert st[0] == 0 and st[1] == 0
    check_panel(P2, out, V, T, tau, G)
