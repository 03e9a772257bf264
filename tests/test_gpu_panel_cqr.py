"""-m gpu: the full-width tall panel (qr_panel_cqr.hip) against numpy -- CholeskyQR2 + Householder reconstruction of a whole panel of up
to 128 columns in six passes.  Checked as a compact-WY panel: V unit lower trapezoidal and equal to the in-place tails, T upper
triangular with tau on its diagonal and T^-1 + T^-T = V^T V, (I - V T V^T)^T P = [R; 0], R equal to LAPACK's up to the signs of its
rows; ill-conditioned panels are refused with A untouched; and the whole route inside qr_geqrf_dev / the TSQR shard."""
import ctypes as C

import numpy as np
import pytest
import torch

from gpu_util import dev, host, zeros

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def q(qr):
    qr.check(qr.lib.qrd_init(), "qrd_init")
    L = qr.lib
    L.qrd_panel_cqr_ws_doubles.restype = C.c_size_t
    L.qrd_panel_cqr_g1.restype = C.c_void_p
    L.qrd_panel_cqr_g1.argtypes = [C.c_void_p]
    L.qrd_panel_cqr_g2.restype = C.c_void_p
    L.qrd_panel_cqr_g2.argtypes = [C.c_void_p]
    L.qrd_panel_cqr_ok.argtypes = [C.c_int, C.c_int]
    L.qrd_panel_cqr_stage1.restype = C.c_int
    L.qrd_panel_cqr_stage1.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]
    L.qrd_panel_cqr_stage2.restype = C.c_int
    L.qrd_panel_cqr_stage2.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_int,
                                       C.c_void_p, C.c_void_p]
    L.qrd_panel_cqr.restype = C.c_int
    L.qrd_panel_cqr.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]
    L.qrd_panel_cqr_q.restype = C.c_int
    L.qrd_panel_cqr_q.argtypes = L.qrd_panel_cqr.argtypes + [C.c_void_p, C.c_int, C.c_void_p, C.c_uint]
    L.qrd_gemm_tn.restype = C.c_int
    L.qrd_gemm_tn.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_double, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_double, C.c_void_p,
                              C.c_int, C.c_void_p, C.c_size_t, C.c_void_p, C.c_int]
    return qr


def run_panel(q, P, lda=None, ldv=None, two_stage=False, qbuf=False, v0=0.0):
    """qbuf: Q in a buffer of its own (what the library's schedule does); the workspace is POISONED with NaN either way (the plan's is
    never zeroed), v0: what Vw holds before the call"""
    mk, w = P.shape
    lda, ldv = lda or mk, ldv or mk
    buf = np.full((lda, w), 7.0)
    buf[:mk] = P
    dA, dV = dev(buf), dev(np.full((ldv, w), v0))
    dT, dtau = dev(np.full((w, w), np.nan)), zeros(w, 1)
    ws = torch.full((int(q.lib.qrd_panel_cqr_ws_doubles()),), float("nan"), dtype=torch.float64, device="cuda")
    status = torch.zeros(4, dtype=torch.int32, device="cuda")
    dQ = dev(np.full((ldv, w), np.nan)) if qbuf else None
    cap = 1 << 22
    slabs = torch.zeros(cap, dtype=torch.float64, device="cuda")
    torch.cuda.synchronize()
    L = q.lib
    if qbuf:
        assert L.qrd_panel_cqr_q(None, dA.data_ptr(), lda, mk, w, dtau.data_ptr(), dT.data_ptr(), w, dV.data_ptr(), ldv, ws.data_ptr(), status.data_ptr(),
                                 dQ.data_ptr(), ldv, None, 0) == 0
    elif not two_stage:
        assert L.qrd_panel_cqr(None, dA.data_ptr(), lda, mk, w, dtau.data_ptr(), dT.data_ptr(), w, dV.data_ptr(), ldv, ws.data_ptr(), status.data_ptr()) == 0
    else:          # the development form: Gram matrices from the library's general product
        g1, g2 = L.qrd_panel_cqr_g1(ws.data_ptr()), L.qrd_panel_cqr_g2(ws.data_ptr())
        assert L.qrd_gemm_tn(None, w, w, mk, 1.0, dA.data_ptr(), lda, dA.data_ptr(), lda, 0.0, g1, 128, slabs.data_ptr(), cap, None, 0) == 0
        assert L.qrd_panel_cqr_stage1(None, dA.data_ptr(), lda, mk, w, dV.data_ptr(), ldv, ws.data_ptr(), status.data_ptr()) == 0
        assert L.qrd_gemm_tn(None, w, w, mk, 1.0, dV.data_ptr(), ldv, dV.data_ptr(), ldv, 0.0, g2, 128, slabs.data_ptr(), cap, None, 0) == 0
        assert L.qrd_panel_cqr_stage2(None, dA.data_ptr(), lda, mk, w, dtau.data_ptr(), dT.data_ptr(), w, dV.data_ptr(), ldv, ws.data_ptr(),
                                      status.data_ptr()) == 0
    q.check(L.qrd_device_sync(), "sync")
    out = host(dA)
    assert np.array_equal(out[mk:], buf[mk:]), "rows below the panel are never written"
    return out[:mk], host(dV)[:mk], host(dT), host(dtau)[:, 0], status.cpu().numpy()


def check_panel(P, out, V, T, tau):
    mk, w = P.shape
    assert np.isfinite(out).all() and np.isfinite(V).all() and np.isfinite(T).all() and np.isfinite(tau).all()
    assert np.array_equal(np.triu(V[:w], 1), np.zeros((w, w))) and np.array_equal(np.diag(V[:w]), np.ones(w))
    assert np.array_equal(np.tril(V, -1), np.tril(out, -1)), "explicit V and in-place tails must agree"
    assert np.array_equal(np.tril(T, -1), np.zeros((w, w))) and np.array_equal(np.diag(T), tau)
    Ti = np.linalg.inv(T)
    assert np.abs(Ti + Ti.T - V.T @ V).max() < 1e-11 * max(1.0, np.abs(Ti).max())          # compact-WY identity
    QtP = P - V @ (T.T @ (V.T @ P))
    scale = np.abs(P).max()
    assert np.abs(np.tril(QtP, -1)).max() < 1e-11 * np.sqrt(mk) * scale
    assert np.abs(np.triu(QtP[:w]) - np.triu(out[:w])).max() < 1e-11 * np.sqrt(mk) * scale
    R = np.triu(out[:w])
    Rref = np.linalg.qr(P, mode="r")
    S = np.sign(np.diag(R)) * np.sign(np.diag(Rref))
    assert np.linalg.norm(S[:, None] * R - Rref) / np.linalg.norm(Rref) < 1e-12
    # Householder's sign convention (reference qr.c:141-151): the diagonal of R has the sign opposite to the pivot it annihilates
    assert R[0, 0] * P[0, 0] < 0


@pytest.mark.parametrize("mk,w", [(4096, 128), (4096, 64), (5000, 96), (1000, 32), (20000, 128), (300, 128), (32768, 128), (40003, 64)])
def test_panel_cqr_well_conditioned(q, mk, w):
    P = np.random.default_rng(mk + w).random((mk, w))
    out, V, T, tau, st = run_panel(q, P, lda=mk + 6, ldv=mk + 2, qbuf=(mk % 3 == 0))
    assert st[0] == 0, "the guard refused a well-conditioned panel"
    check_panel(P, out, V, T, tau)


def test_panel_cqr_two_stage_form_with_external_gram_matrices(q):
    P = np.random.default_rng(77).random((9000, 128))
    out, V, T, tau, st = run_panel(q, P, two_stage=True)
    assert st[0] == 0
    check_panel(P, out, V, T, tau)


def test_panel_cqr_is_bitwise_reproducible(q):
    P = np.random.default_rng(78).random((50000, 96))
    a = run_panel(q, P)
    b = run_panel(q, P)
    assert all(np.array_equal(x, y) for x, y in zip(a[:4], b[:4]))


def test_panel_cqr_moderately_conditioned_takes_the_second_cholesky(q):
    """cond ~ 3e5: |Q^T Q - I| after one pass is far above 1e-9 (second Cholesky instead of the first-order factor), still below 1/64"""
    rng = np.random.default_rng(3)
    mk, w = 8192, 128
    U, _ = np.linalg.qr(rng.standard_normal((mk, w)))
    W, _ = np.linalg.qr(rng.standard_normal((w, w)))
    P = (U * np.logspace(0, -5.5, w)) @ W.T
    out, V, T, tau, st = run_panel(q, P)
    assert st[0] == 0
    check_panel(P, out, V, T, tau)


@pytest.mark.parametrize("kind", ["rank_deficient", "cond_1e9", "nan"])
def test_panel_cqr_refuses_and_leaves_the_panel_untouched(q, kind):
    rng = np.random.default_rng(11)
    mk, w = 4096, 128
    P = rng.random((mk, w))
    if kind == "rank_deficient":
        P[:, 77] = P[:, 3] + P[:, 5]
    elif kind == "cond_1e9":
        U, _ = np.linalg.qr(rng.standard_normal((mk, w)))
        W, _ = np.linalg.qr(rng.standard_normal((w, w)))
        P = (U * np.logspace(0, -9, w)) @ W.T
    else:
        P[100, 7] = np.nan
    out, V, T, tau, st = run_panel(q, P)
    assert st[0] == 1 and st[1] == 1
    assert np.array_equal(out, P, equal_nan=True), "a refused panel must be left exactly as it was"
    # with Q in a buffer of its own (the library's schedule) the explicit V is untouched as well: the leaf chain that takes the panel over
    # relies on the zeros above each leaf's diagonal block (round-4 advisor finding: Q used to be parked in Vw)
    out, V, T, tau, st = run_panel(q, P, qbuf=True, v0=3.25)
    assert st[0] == 1
    assert np.array_equal(out, P, equal_nan=True) and np.array_equal(V, np.full_like(V, 3.25))


@pytest.mark.parametrize("m,n,nb", [(458752, 256, 128), (400001, 128, 128)])
def test_geqrf_takes_the_full_width_route_on_tall_shapes(qr, m, n, nb):
    """qr_geqrf_dev on tall shapes (128-column panels of at least MI355XQR_CQR_MIN_ROWS = 196608 rows: the full-width route) against LAPACK"""
    A = np.random.default_rng(m + n).random((m, n))
    p = qr.Plan(m, n, nb, 32)
    dA, dtau, dQ = dev(A), zeros(n, 1), zeros(m, n)
    p.geqrf(dA, m, n, m, dtau)
    p.sync()
    F = host(dA)
    p.applyq(dA, m, n, m, dtau, dQ, n, m, True)
    p.sync()
    Q = host(dQ)
    p.close()
    R = np.triu(F[:n])
    Rref = np.linalg.qr(A, mode="r")
    S = np.sign(np.diag(R)) * np.sign(np.diag(Rref))
    assert np.linalg.norm(S[:, None] * R - Rref) / np.linalg.norm(Rref) < 1e-12
    assert np.linalg.norm(A - Q @ R) / np.linalg.norm(A) < 1e-13
    assert np.linalg.norm(Q.T @ Q - np.eye(n)) < 1e-12


def test_parked_panels_give_the_same_bits_as_v_written_twice(qr):
    """Parked full-width panels (tall single-stream plans: V written once, into the caller's array, R of the top block restored from
    the panel workspace behind the update) against MI355XQR_CQR_PARK=0 (child process on the lab library: a measurement knob, read once): the factored array, tau
    and the thin Q must be IDENTICAL -- the update reads the same V values from another place, nothing else changes."""
    import subprocess, sys, os
    m, n, nb = 98304, 384, 128                     # two parked panels and a last one that is not (nothing follows it)
    p = qr.Plan(m, n, nb, 32)
    dA, dtau, dQ = zeros(m, n), zeros(n, 1), zeros(m, n)
    p.fill_uniform(dA, m, m, n, seed=31)
    p.sync()
    A0 = host(dA)
    p.geqrf(dA, m, n, m, dtau)
    p.sync()
    F, tau = host(dA), host(dtau)[:, 0]
    assert p.route_stats()["tall_panels"] == 3 and p.route_stats()["tall_panels_refused"] == 0
    p.applyq(dA, m, n, m, dtau, dQ, n, m, True)
    p.sync()
    Q = host(dQ)
    p.close()
    R = np.triu(F[:n])
    assert np.linalg.norm(A0 - Q @ R) / np.linalg.norm(A0) < 1e-13 and np.linalg.norm(Q.T @ Q - np.eye(n)) < 1e-12
    code = ("import sys, numpy as np, torch; sys.path.insert(0, %r); import cuda_qr_amd as q\n"
            "m, n = 98304, 384\n"
            "p = q.Plan(m, n, 128, 32)\n"
            "A = torch.zeros((n, m), dtype=torch.float64, device='cuda'); tau = torch.zeros((1, n), dtype=torch.float64, device='cuda'); torch.cuda.synchronize()\n"
            "p.fill_uniform(A, m, m, n, seed=31); p.sync(); p.geqrf(A, m, n, m, tau); p.sync()\n"
            "np.save(sys.argv[1], A.cpu().numpy().T); np.save(sys.argv[2], tau.cpu().numpy().ravel())\n") % os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    subprocess.run([sys.executable, "-c", code, "/tmp/park_off_F.npy", "/tmp/park_off_tau.npy"], check=True,
                   env=dict(os.environ, CUDA_QR_AMD_LIB="lab", MI355XQR_CQR_PARK="0"), timeout=300)      # a measurement knob: lab library
    assert np.array_equal(np.load("/tmp/park_off_F.npy"), F) and np.array_equal(np.load("/tmp/park_off_tau.npy"), tau)


def _ill_conditioned(kind, m, n, seed):
    rng = np.random.default_rng(seed)
    A = rng.random((m, n))
    if kind == "cond_1e9":
        U, _ = np.linalg.qr(rng.standard_normal((m, n)))           # every 128-column panel of condition 1e9 on its own
        for c in range(0, n, 128):
            W, _ = np.linalg.qr(rng.standard_normal((128, 128)))
            A[:, c:c + 128] = (U[:, c:c + 128] * np.logspace(0, -9, 128)) @ W.T
    elif kind == "duplicated_column":
        A[:, 200] = A[:, 131]            # inside the second 128-column panel
        A[:, 17] = A[:, 3]               # and the first
    elif kind == "zero_column":
        A[:, 150] = 0.0
    return A


@pytest.mark.parametrize("kind", ["cond_1e9", "duplicated_column", "zero_column"])
def test_geqrf_with_refused_tall_panels_is_householder_grade(qr, kind):
    """qr_geqrf_dev on a tall shape whose 128-column panels the full-width route REFUSES (cond > 1e7, dependent or zero columns): the leaf
    chain takes the untouched panel over and the result is as good as for any other input.  The plan is used for a well-conditioned
    matrix FIRST, so that every workspace holds stale data (round-4 advisor finding: the fall-back ran on a V buffer polluted with Q)."""
    m, n, nb = 200000, 256, 128
    p = qr.Plan(m, n, nb, 32)
    dtau, dQ = zeros(n, 1), zeros(m, n)
    dA = dev(np.random.default_rng(1).random((m, n)))
    p.geqrf(dA, m, n, m, dtau)
    p.sync()
    assert p.route_stats()["tall_panels"] == 2 and p.route_stats()["tall_panels_refused"] == 0
    A = _ill_conditioned(kind, m, n, 5)
    dA = dev(A)
    p.geqrf(dA, m, n, m, dtau)
    p.sync()
    st = p.route_stats()
    assert st["tall_panels"] == 4 and st["tall_panels_refused"] >= 1, st
    F = host(dA)
    p.applyq(dA, m, n, m, dtau, dQ, n, m, True)
    p.sync()
    Q = host(dQ)
    p.close()
    R = np.triu(F[:n])
    assert np.isfinite(R).all() and np.isfinite(Q).all()
    assert np.linalg.norm(A - Q @ R) / np.linalg.norm(A) < 1e-13
    assert np.linalg.norm(Q.T @ Q - np.eye(n)) < 1e-12
    if kind != "cond_1e9":
        # R is determined by A up to its first dependent column (behind it the arbitrary direction Householder picks for the dependent
        # column enters every later row of R); that column's diagonal entry is zero to rounding
        jd = 17 if kind == "duplicated_column" else 150
        Rref = np.linalg.qr(A[:, :jd], mode="r")
        assert np.allclose(np.abs(np.diag(R)[:jd]), np.abs(np.diag(Rref)), rtol=1e-10)
        assert abs(R[jd, jd]) < 1e-10 * np.abs(np.diag(R)[:jd]).max()


def test_latch_mode_leaves_a_refused_parked_panel_untouched(qr):
    """latch mode, one 128-column panel (the last panel parks: V once, R restored from the workspace behind it): when the device-side
    guard refuses it, every launch of the panel -- the restore included -- returns at once and the array is bit for bit what it was"""
    m, n, nb = 32768, 128, 128
    p = qr.Plan(m, n, nb, 32)
    p.set_guard_mode(True)
    dtau = zeros(n, 1)
    good = np.random.default_rng(5).standard_normal((m, n))
    dG = dev(good)
    p.geqrf(dG, m, n, m, dtau)              # leaves an R of its own in the panel workspace
    p.sync()
    assert p.route_stats()["tall_panels"] >= 1
    A = np.random.default_rng(9).random((m, n))
    A[:, 17] = A[:, 3]
    dA = dev(A)
    p.geqrf(dA, m, n, m, dtau)
    with pytest.raises(qr.QRError, match="-106"):
        p.sync()
    back = host(dA)
    p.close()
    assert np.array_equal(back, A)


def test_latch_mode_reports_a_refused_panel_at_sync_and_never_blocks(qr):
    """latch mode (qr_plan_set_guard_mode): qr_geqrf_dev is stream-ordered -- four tall factorisations are queued before the first has
    finished -- and a refusal comes back from qr_plan_sync as QR_E_REFUSED (-106), once"""
    import time
    m, n, nb = 262144, 256, 128
    p = qr.Plan(m, n, nb, 32)
    p.set_guard_mode(True)
    dtau = zeros(n, 1)
    bufs = [dev(np.zeros((m, n))) for _ in range(4)]
    for b in bufs:
        p.fill_uniform(b, m, m, n, seed=12)
    p.geqrf(bufs[0], m, n, m, dtau)          # warm-up
    p.sync()
    for b in bufs:
        p.fill_uniform(b, m, m, n, seed=12)
    p.sync()
    t0 = time.perf_counter()
    for b in bufs:
        p.geqrf(b, m, n, m, dtau)
    t_issue = time.perf_counter() - t0
    p.sync()
    t_all = time.perf_counter() - t0
    assert t_issue < 0.5 * t_all, (t_issue, t_all)           # the host ran ahead of the device
    assert p.route_stats()["tall_panels_refused"] == 0
    A = _ill_conditioned("duplicated_column", m, n, 9)
    dA = dev(A)
    p.geqrf(dA, m, n, m, dtau)
    with pytest.raises(qr.QRError, match="-106"):
        p.sync()
    p.sync()                                                # cleared by the read
    assert p.route_stats()["tall_panels_refused"] >= 1
    p.set_guard_mode(False)                                 # the documented recovery: a fresh copy, default mode
    dA = dev(A)
    dQ = zeros(m, n)
    p.geqrf(dA, m, n, m, dtau)
    p.sync()
    R = np.triu(host(dA)[:n])
    p.applyq(dA, m, n, m, dtau, dQ, n, m, True)
    p.sync()
    Q = host(dQ)
    p.close()
    assert np.linalg.norm(A - Q @ R) / np.linalg.norm(A) < 1e-13 and np.linalg.norm(Q.T @ Q - np.eye(n)) < 1e-12


def test_bench_cond_input_refuses_every_tall_panel(qr):
    """bench.py --workload tsqr --cond 1e9 (the price of a refusal): its input -- every 128-column panel = the panel's
    first column + uniform noise / cond -- must make the device-side guard refuse EVERY full-width panel, and the leaf chain that takes
    them over must still deliver a Householder-grade factorisation."""
    m, n, nb, cond = 65536, 256, 128, 1e9
    p = qr.Plan(m, n, nb, 32)
    dA, dtau, dQ = zeros(m, n), zeros(n, 1), zeros(m, n)
    p.fill_uniform(dA, m, m, n, seed=12)
    p.sync()
    for c in range(0, n, 128):                      # bench.py: condition()
        dA[c + 1:c + 128] = dA[c:c + 1] + dA[c + 1:c + 128] / cond
    torch.cuda.synchronize()
    A = host(dA)
    p.geqrf(dA, m, n, m, dtau)
    p.sync()
    st = p.route_stats()
    assert st["tall_panels"] == 2 and st["tall_panels_refused"] == 2, st
    # round 6: a refused panel is retried preconditioned (shifted CholeskyQR3) before the leaf chain; cond 1e9 is well inside its range
    assert st["tall_panels_retried"] == 2 and st["tall_panels_retry_accepted"] == 2, st
    R = np.triu(host(dA)[:n])
    p.applyq(dA, m, n, m, dtau, dQ, n, m, True)
    p.sync()
    Q = host(dQ)
    p.close()
    assert np.isfinite(R).all() and np.isfinite(Q).all()
    assert np.linalg.norm(A - Q @ R) / np.linalg.norm(A) < 1e-13
    assert np.linalg.norm(Q.T @ Q - np.eye(n)) < 1e-12


@pytest.mark.parametrize("cond", [1e7, 1e8, 1e9, 1e10, 1e11, 1e13])
def test_refused_panel_retry_is_householder_grade_across_conditions(qr, cond):
    """The preconditioned retry of a refused full-width panel (shifted CholeskyQR3: R0 = chol(A^T A + s I), the three-pass pipeline on
    A R0^-1, R = S R2 R1 R0) on panels whose singular values fall geometrically from 1 to 1 / cond, mixed over all columns: whichever
    route ends up factoring the panel -- first attempt (cond <~ 1e7), retry (up to ~1e10), Householder leaf chain (beyond) -- the result
    must be Householder-grade: backward error and orthogonality at round-off, R equal to LAPACK's where R is well determined."""
    m, n = 32768, 128
    rng = np.random.default_rng(int(np.log10(cond)))
    U, _ = np.linalg.qr(rng.standard_normal((m, n)))
    W, _ = np.linalg.qr(rng.standard_normal((n, n)))
    A = (U * np.logspace(0, -np.log10(cond), n)) @ W.T
    p = qr.Plan(m, n, 128, 32)
    dA, dtau, dQ = dev(A), zeros(n, 1), zeros(m, n)
    p.geqrf(dA, m, n, m, dtau)
    p.sync()
    st = p.route_stats()
    R = np.triu(host(dA)[:n])
    p.applyq(dA, m, n, m, dtau, dQ, n, m, True)
    p.sync()
    Q = host(dQ)
    p.close()
    assert st["tall_panels"] == 1, st
    if cond >= 1e8:
        assert st["tall_panels_refused"] == 1 and st["tall_panels_retried"] == 1, st
    if 1e8 <= cond <= 1e9:
        assert st["tall_panels_retry_accepted"] == 1, st                 # inside the retry's range: no leaf chain
    assert np.isfinite(R).all() and np.isfinite(Q).all()
    assert np.linalg.norm(A - Q @ R) / np.linalg.norm(A) < 1e-14 * 4, (cond, st)
    assert np.linalg.norm(Q.T @ Q - np.eye(n)) < 2e-13, (cond, st)
    from oracle import oracle as O                                       # the checker: sign normalisation only
    Rl = O.sign_normalise(np.linalg.qr(A, mode="r"))
    assert np.linalg.norm(O.sign_normalise(R) - Rl) / np.linalg.norm(Rl) < 1e-11, (cond, st)
