"""-m gpu: each hand-written HIP kernel against a numpy restatement of the same operation
(fp64; tolerance stated per test).  Calls go through the library's launch layer."""
import numpy as np
import pytest
import torch

from gpu_util import dev, host, rel, zeros

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def q(qr):
    qr.check(qr.lib.qrd_init(), "qrd_init")
    return qr


def _sync(q):
    q.check(q.lib.qrd_device_sync(), "sync")


NN_SHAPES = [(16, 16, 4), (128, 128, 16), (256, 384, 128), (130, 70, 33), (1000, 40, 32), (37, 300, 7),
             (4096, 128, 128), (2048, 2048, 64), (6, 4, 2), (515, 515, 515), (12800, 32, 32), (64, 1, 64)]


@pytest.mark.parametrize("M,N,K", NN_SHAPES)
def test_gemm_nn(q, M, N, K):
    """C = beta*C + alpha*A*B on f64 MFMA tiles; asymmetric random data catches any row/col swap."""
    rng = np.random.default_rng(M * 7 + N * 3 + K)
    A, B, C0 = rng.standard_normal((M, K)), rng.standard_normal((K, N)), rng.standard_normal((M, N))
    # (0, 1) and (0.5, 1): beta = 1 only takes the "C enters through the accumulators" shortcut for alpha = +-1
    for alpha, beta in ((1.0, 0.0), (-1.0, 1.0), (0.5, -2.0), (0.0, 1.0), (0.5, 1.0), (1.0, 1.0)):
        dA, dB, dC = dev(A), dev(B), dev(C0)
        torch.cuda.synchronize()
        q.check(q.lib.qrd_gemm_nn(None, M, N, K, alpha, dA.data_ptr(), M, dB.data_ptr(), K, beta, dC.data_ptr(), M))
        _sync(q)
        ref = alpha * (A @ B) + beta * C0
        assert rel(host(dC), ref) < 1e-13, (M, N, K, alpha, beta)


@pytest.mark.parametrize("M,N,K,lda,ldbt,ldc", [(128, 128, 16, 128, 128, 128), (384, 256, 64, 384, 256, 384), (1024, 640, 256, 1030, 700, 1100),
                                               (2048, 1920, 128, 2048, 2048, 2048), (256, 4096, 512, 300, 4096, 258)])
@pytest.mark.parametrize("sign", [-1, 1])
def test_gemm_nt_update(q, M, N, K, lda, ldbt, ldc, sign):
    """Second-generation trailing update C -+= A Bt^T (W stored transposed, direct-to-LDS tile loads, permuted MFMA rows):
    asymmetric random data, every XCD tile-order variant."""
    rng = np.random.default_rng(M + N + K)
    A, Bt, C0 = rng.standard_normal((lda, K)), rng.standard_normal((ldbt, K)), rng.standard_normal((ldc, N))
    ref = C0.copy()
    ref[:M] += sign * (A[:M] @ Bt[:N].T)
    for gm in (0, 8, 3):
        dA, dB, dC = dev(A), dev(Bt), dev(C0)
        torch.cuda.synchronize()
        q.check(q.lib.qrd_gemm_nt(None, M, N, K, sign, dA.data_ptr(), lda, dB.data_ptr(), ldbt, dC.data_ptr(), ldc, gm, None))
        _sync(q)
        out = host(dC)
        assert rel(out[:M], ref[:M]) < 1e-13 and np.array_equal(out[M:], C0[M:]), (gm,)


@pytest.mark.parametrize("M,N,K,lda", [(192, 64, 64, 256), (4032, 3968, 64, 4096), (4032, 4032, 64, 4096), (1984, 192, 256, 2048), (2048, 192, 64, 2050),
                                       (320, 320, 32, 384), (200, 128, 64, 256), (4030, 1984, 256, 4096), (130, 64, 128, 256), (3850, 3840, 256, 3968)])
def test_gemm_nt4_half_tiles(q, M, N, K, lda):
    """Round 6: the four-workgroup update kernel on any even M (ragged bottom row tile: gemm_nt4_kernel<.., RAG>) and N = 64 (mod 128):
    the trailing matrices of every other 64-column outer step, matrices whose height is not a multiple of 128.  A's rows beyond M (read by the tile loader, never used) are NaN; C's rows beyond M
    are never written."""
    rng = np.random.default_rng(M + N + K)
    ldbt, ldc = N + 2, lda + 2
    A, Bt, C0 = rng.standard_normal((lda, K)), rng.standard_normal((ldbt, K)), rng.standard_normal((ldc, N))
    A[M:] = np.nan
    ref = C0.copy()
    ref[:M] -= A[:M] @ Bt[:N].T
    assert q.lib.qrd_gemm_nt4_ok(M, N, K, None, lda, None, ldbt, None, ldc) == 1
    for gm in (0, 8):
        dA, dB, dC = dev(A), dev(Bt), dev(C0)
        torch.cuda.synchronize()
        q.check(q.lib.qrd_gemm_nt(None, M, N, K, -1, dA.data_ptr(), lda, dB.data_ptr(), ldbt, dC.data_ptr(), ldc, gm, None))
        _sync(q)
        out = host(dC)
        assert rel(out[:M], ref[:M]) < 1e-13 and np.array_equal(out[M:], C0[M:]), (gm,)
    # a half tile whose 128 rows do not fit A's leading dimension is declined
    assert q.lib.qrd_gemm_nt4_ok(192, 64, 64, None, 192, None, ldbt, None, ldc) == 0
    assert q.lib.qrd_gemm_nt(None, 192, 64, 64, -1, dA.data_ptr(), 192, dB.data_ptr(), ldbt, dC.data_ptr(), ldc, 0, None) != 0


def test_gemm_nt_rejects_ragged_shapes(q):
    d = zeros(256, 256)
    assert q.lib.qrd_gemm_nt(None, 200, 128, 16, -1, d.data_ptr(), 256, d.data_ptr(), 256, d.data_ptr(), 256, 0, None) != 0
    assert q.lib.qrd_gemm_nt(None, 128, 128, 8, -1, d.data_ptr(), 256, d.data_ptr(), 256, d.data_ptr(), 256, 0, None) != 0
    assert q.lib.qrd_gemm_nt(None, 128, 128, 16, -1, d.data_ptr(), 255, d.data_ptr(), 256, d.data_ptr(), 256, 0, None) != 0


def test_gemm_nn_with_leading_dimensions(q):
    rng = np.random.default_rng(5)
    M, N, K, lda, ldb, ldc = 200, 96, 48, 260, 50, 333     # ldc odd -> scalar path for C only
    A, B, C0 = rng.standard_normal((lda, K)), rng.standard_normal((ldb, N)), rng.standard_normal((ldc, N))
    dA, dB, dC = dev(A), dev(B), dev(C0)
    torch.cuda.synchronize()
    q.check(q.lib.qrd_gemm_nn(None, M, N, K, -1.0, dA.data_ptr(), lda, dB.data_ptr(), ldb, 1.0, dC.data_ptr(), ldc))
    _sync(q)
    out = host(dC)
    ref = C0.copy()
    ref[:M] -= A[:M] @ B[:K]
    assert rel(out, ref) < 1e-13
    assert np.array_equal(out[M:], C0[M:])             # rows beyond M untouched


TN_SHAPES = [(16, 16, 64), (128, 128, 4096), (128, 1000, 5000), (32, 96, 16384), (32, 224, 777), (64, 64, 100),
             (128, 16384, 1024), (5, 3, 11), (256, 256, 3000), (130, 67, 129), (1, 1, 1), (32, 20, 100000),
             (512, 300, 50000), (300, 512, 40000), (257, 33, 9000),
             (512, 1024, 8200), (384, 1000, 10248)]     # K % 16 != 0 above 4 GFLOP: whole k-tiles on the fast path + an accumulated guarded remainder (round 6)


@pytest.mark.parametrize("M,N,K", TN_SHAPES)
def test_gemm_tn_splitk(q, M, N, K):
    """C = A^T B with the long K dimension split over workgroups + deterministic slab reduction."""
    rng = np.random.default_rng(M + 13 * N + K)
    A, B = rng.standard_normal((K, M)), rng.standard_normal((K, N))
    dA, dB, dC = dev(A), dev(B), zeros(M, N)
    cap = 4 << 20
    slabs = torch.empty(cap, dtype=torch.float64, device="cuda")
    torch.cuda.synchronize()
    outs = []
    for _ in range(2):
        q.check(q.lib.qrd_gemm_tn(None, M, N, K, 1.0, dA.data_ptr(), K, dB.data_ptr(), K, 0.0, dC.data_ptr(), M,
                                  slabs.data_ptr(), cap, None, 0))
        _sync(q)
        outs.append(host(dC))
    ref = A.T @ B
    assert np.abs(outs[0] - ref).max() < 1e-13 * np.sqrt(K) * max(1.0, np.abs(ref).max())
    assert np.array_equal(outs[0], outs[1]), "split-K reduction must be bitwise reproducible"


def test_gemm_tn_without_slabs_and_beta(q):
    rng = np.random.default_rng(9)
    M, N, K = 96, 80, 500
    A, B, C0 = rng.standard_normal((K, M)), rng.standard_normal((K, N)), rng.standard_normal((M, N))
    dA, dB, dC = dev(A), dev(B), dev(C0)
    torch.cuda.synchronize()
    q.check(q.lib.qrd_gemm_tn(None, M, N, K, 2.0, dA.data_ptr(), K, dB.data_ptr(), K, -1.0, dC.data_ptr(), M,
                              None, 0, None, 0))
    _sync(q)
    assert rel(host(dC), 2.0 * A.T @ B - C0) < 1e-13


@pytest.mark.parametrize("w,N,K", [(32, 96, 4000), (16, 224, 333), (32, 7, 64), (8, 100, 20000)])
def test_gemm_tn_fused_Tt(q, w, N, K):
    """leaf-level fold: C = T^T (A^T B) with T upper triangular (w <= 32)."""
    rng = np.random.default_rng(w + N + K)
    A, B = rng.standard_normal((K, w)), rng.standard_normal((K, N))
    T = np.triu(rng.standard_normal((w, w)))
    ldt = 40
    Tpad = np.zeros((ldt, w)); Tpad[:w] = T
    dA, dB, dT, dC = dev(A), dev(B), dev(Tpad), zeros(w, N)
    cap = 1 << 20
    slabs = torch.empty(cap, dtype=torch.float64, device="cuda")
    torch.cuda.synchronize()
    q.check(q.lib.qrd_gemm_tn(None, w, N, K, 1.0, dA.data_ptr(), K, dB.data_ptr(), K, 0.0, dC.data_ptr(), w,
                              slabs.data_ptr(), cap, dT.data_ptr(), ldt))
    _sync(q)
    ref = T.T @ (A.T @ B)
    assert np.abs(host(dC) - ref).max() < 1e-12 * np.sqrt(K) * max(1.0, np.abs(ref).max())


TSQR_SHAPES = [(32, 32), (33, 32), (512, 32), (513, 32), (1000, 32), (4096, 32), (8192, 32), (8193, 32), (16384, 32),
               (70000, 32), (200000, 32), (262144, 16), (300, 7), (5, 5), (2, 1), (600, 1), (5000, 24), (20000, 8)]


@pytest.mark.parametrize("mk,w", TSQR_SHAPES)
def test_panel_tsqr_householder_reconstruction(q, oracle, mk, w):
    """Leaf panel by intra-GPU TSQR + Householder reconstruction: the output must be an ordinary compact-WY
    panel -- unit-lower V in place below R, tau = diag(T) = 2/(v^T v), T upper triangular with
    (I - V T V^T)^T P = [R; 0] -- and R must equal LAPACK's up to row signs."""
    rng = np.random.default_rng(mk * 3 + w)
    P = rng.random((mk, w))
    ld, ldv, ldt = mk + 6, mk + 2, w + 3
    buf = np.full((ld, w), 7.0); buf[:mk] = P
    dP = dev(buf)
    dtau, dT, dV = zeros(w, 1), dev(np.full((ldt, w), np.nan)), dev(np.full((ldv, w), np.nan))
    ws = torch.zeros(int(q.lib.qrd_panel_ws_size(mk)), dtype=torch.float64, device="cuda")
    torch.cuda.synchronize()
    q.check(q.lib.qrd_panel_tsqr(None, dP.data_ptr(), ld, mk, w, dtau.data_ptr(), dT.data_ptr(), ldt, dV.data_ptr(),
                                 ldv, ws.data_ptr(), mk))
    _sync(q)
    out, tau, T, V = host(dP), host(dtau)[:, 0], host(dT)[:w], host(dV)[:mk]
    assert np.array_equal(out[mk:], buf[mk:])                                   # padding rows untouched
    assert np.isfinite(out).all() and np.isfinite(T).all() and np.isfinite(V).all()
    R = np.triu(out[:w])
    assert np.array_equal(np.triu(V[:w], 1), np.zeros((w, w))) and np.array_equal(np.diag(V[:w]), np.ones(w))
    assert np.array_equal(np.tril(V, -1), np.tril(out[:mk], -1)), "explicit V and in-place tails must agree"
    assert np.abs(np.tril(T, -1)).max(initial=0.0) == 0.0
    assert np.abs(np.diag(T) - tau).max() == 0.0
    vv = (V * V).sum(axis=0)
    live = vv > 1.0                    # an empty tail (last column of a square block) gives tau = 0, H = I
    assert np.abs(tau[live] - 2.0 / vv[live]).max(initial=0.0) < 1e-12 and np.all(tau[~live] == 0.0)
    QtP = P - V @ (T.T @ (V.T @ P))
    tol = 2e-13 * np.sqrt(mk) * max(1.0, np.abs(R).max())
    assert np.abs(QtP[:w] - R).max() < tol and np.abs(QtP[w:]).max(initial=0.0) < tol
    ref = oracle.sign_normalise(np.linalg.qr(P, mode="r"))
    assert np.linalg.norm(oracle.sign_normalise(R) - ref) / np.linalg.norm(ref) < 1e-13 * max(1.0, np.sqrt(mk) / 30)
    if mk <= 4096:
        H = np.eye(mk) - V @ T @ V.T
        assert np.abs(H.T @ H - np.eye(mk)).max() < 1e-12


def test_panel_tsqr_zero_and_dependent_columns(q):
    mk, w = 3000, 16
    P = np.random.default_rng(1).random((mk, w))
    P[:, 3] = 0.0
    P[:, 9] = P[:, 2]
    dP, dtau, dT, dV = dev(P), zeros(w, 1), zeros(w, w), zeros(mk, w)
    ws = torch.zeros(int(q.lib.qrd_panel_ws_size(mk)), dtype=torch.float64, device="cuda")
    torch.cuda.synchronize()
    q.check(q.lib.qrd_panel_tsqr(None, dP.data_ptr(), mk, mk, w, dtau.data_ptr(), dT.data_ptr(), w, dV.data_ptr(), mk,
                                 ws.data_ptr(), mk))
    _sync(q)
    out, T, V = host(dP), host(dT), host(dV)
    assert np.isfinite(out).all() and np.isfinite(T).all()
    QtP = P - V @ (T.T @ (V.T @ P))
    assert np.abs(np.tril(QtP, -1)).max() < 1e-11 and np.abs(np.triu(QtP[:w]) - np.triu(out[:w])).max() < 1e-11


def _cholqr_leaf(q, P, ld=None, ldv=None, ldt=None):
    mk, w = P.shape
    ld, ldv, ldt = ld or mk, ldv or mk, ldt or w
    buf = np.full((ld, w), 7.0); buf[:mk] = P
    dP = dev(buf)
    dtau, dT, dV = zeros(w, 1), dev(np.full((ldt, w), np.nan)), dev(np.full((ldv, w), np.nan))
    ws = torch.zeros(int(q.lib.qrd_panel_ws_size(mk)), dtype=torch.float64, device="cuda")
    cws = torch.zeros(4 * 32 * 32 + 16, dtype=torch.float64, device="cuda")
    slabs = torch.zeros(1 << 22, dtype=torch.float64, device="cuda")
    torch.cuda.synchronize()
    q.check(q.lib.qrd_panel_cholqr(None, dP.data_ptr(), ld, mk, w, dtau.data_ptr(), dT.data_ptr(), ldt, dV.data_ptr(), ldv,
                                   ws.data_ptr(), mk, cws.data_ptr(), slabs.data_ptr(), 1 << 22, 0))
    _sync(q)
    guard = int(cws[4 * 32 * 32:].view(torch.int32)[0].item())
    return host(dP), host(dtau)[:, 0], host(dT)[:w], host(dV)[:mk], guard, buf


def _check_compact_wy(oracle, P, out, tau, T, V, rtol=1e-13):
    mk, w = P.shape
    assert np.isfinite(out).all() and np.isfinite(T).all() and np.isfinite(V).all()
    R = np.triu(out[:w])
    assert np.array_equal(np.triu(V[:w], 1), np.zeros((w, w))) and np.array_equal(np.diag(V[:w]), np.ones(w))
    assert np.array_equal(np.tril(V, -1), np.tril(out[:mk], -1)), "explicit V and in-place tails must agree"
    assert np.abs(np.tril(T, -1)).max(initial=0.0) == 0.0 and np.abs(np.diag(T) - tau).max() == 0.0
    QtP = P - V @ (T.T @ (V.T @ P))
    tol = 2e-13 * np.sqrt(mk) * max(1.0, np.abs(R).max())
    assert np.abs(QtP[:w] - R).max() < tol and np.abs(QtP[w:]).max(initial=0.0) < tol
    ref = oracle.sign_normalise(np.linalg.qr(P, mode="r"))
    assert np.linalg.norm(oracle.sign_normalise(R) - ref) / np.linalg.norm(ref) < rtol * max(1.0, np.sqrt(mk) / 30)
    if mk <= 4096:
        H = np.eye(mk) - V @ T @ V.T
        assert np.abs(H.T @ H - np.eye(mk)).max() < 1e-12


@pytest.mark.parametrize("mk", [513, 1000, 4096, 8192, 16384, 70000, 262144])
def test_panel_cholqr2_fast_path(q, oracle, mk):
    """CholeskyQR2 + Householder reconstruction leaf on a well-conditioned panel: the guard word must read 0 (the
    Householder-TSQR launches behind it were no-ops) and the output must be the same kind of compact-WY panel."""
    w = 32
    P = np.random.default_rng(mk).random((mk, w))
    out, tau, T, V, guard, buf = _cholqr_leaf(q, P, ld=mk + 6, ldv=mk + 2, ldt=w + 3)
    assert guard == 0
    assert np.array_equal(out[mk:], buf[mk:])
    _check_compact_wy(oracle, P, out, tau, T, V)
    vv = (V * V).sum(axis=0)
    assert np.abs(tau - 2.0 / vv).max() < 1e-12


@pytest.mark.parametrize("kind", ["zero_column", "dependent", "cond1e10", "graded", "nan_free_tiny"])
def test_panel_cholqr2_guard_falls_back_to_householder(q, oracle, kind):
    """Panels CholeskyQR2 must refuse (zero / dependent columns, cond 1e10): the guard word stays 1 and the Householder
    TSQR launches do the work.  Column grading and a tiny overall scale do not hurt CholeskyQR (it is invariant to column
    scaling); whichever route is taken there, the accuracy must be that of the Householder path."""
    mk, w = 6000, 32
    rng = np.random.default_rng(7)
    P = rng.random((mk, w))
    rtol = 1e-13
    if kind == "zero_column":
        P[:, 5] = 0.0
    elif kind == "dependent":
        P[:, 9] = P[:, 2]
    elif kind == "cond1e10":
        U, _ = np.linalg.qr(rng.standard_normal((mk, w))); Vr, _ = np.linalg.qr(rng.standard_normal((w, w)))
        P = (U * np.logspace(0, -10, w)) @ Vr.T
        rtol = 1e-5                                    # forward error of R ~ cond * eps
    elif kind == "graded":
        P = P * np.logspace(0, -12, w)                 # column scaling: badly conditioned Gram matrix
        rtol = 1e-12
    else:
        P = P * 1e-160                                 # Gram matrix underflows to zero
        rtol = 1e-12
    out, tau, T, V, guard, _ = _cholqr_leaf(q, P)
    if kind in ("zero_column", "dependent", "cond1e10"):
        assert guard == 1
    assert np.isfinite(out).all() and np.isfinite(T).all() and np.isfinite(V).all()
    QtP = P - V @ (T.T @ (V.T @ P))
    cs = np.maximum(np.abs(P).max(axis=0), 1e-300)          # column-wise scale (graded columns)
    assert (np.abs(np.tril(QtP, -1)) / cs).max() < 1e-11 * np.sqrt(mk)
    assert (np.abs(np.triu(QtP[:w]) - np.triu(out[:w])) / cs).max() < 1e-11 * np.sqrt(mk)
    if kind in ("cond1e10", "graded", "nan_free_tiny"):
        ref = oracle.sign_normalise(np.linalg.qr(P, mode="r"))
        R = np.triu(out[:w])
        assert np.linalg.norm(oracle.sign_normalise(R) - ref) / np.linalg.norm(ref) < rtol


@pytest.mark.parametrize("mk", [514, 700, 1024, 4098, 8192, 8194, 16384, 16386, 40000, 65536, 70002, 262144, 300000])
def test_panel_guard_route_all_heights(q, oracle, mk):
    """A leaf the CholeskyQR2 route must refuse (two equal columns) at heights on both sides of every switch of the guard route:
    one block / several blocks, 512- and 1024-row blocks, the one-launch cooperative form of short leaves (<= 16 blocks) and of tall
    ones (persistent workgroups, one upper tree level up to 262144 rows, two above), odd sizes.  The Householder route's result must
    be a valid compact-WY panel."""
    w = 32
    P = np.random.default_rng(mk).random((mk, w))
    P[:, 17] = P[:, 4]
    out, tau, T, V, guard, _ = _cholqr_leaf(q, P)
    # (a tall leaf whose height is not a multiple of 4 has no streaming CholeskyQR2 form since round 5: it takes the Householder route
    # directly, without a refusal to record)
    direct = mk > 32 * 512 and mk % 4 != 0
    assert guard == (0 if direct else 1)
    assert np.isfinite(out).all() and np.isfinite(T).all() and np.isfinite(V).all() and np.isfinite(tau).all()
    QtP = P - V @ (T.T @ (V.T @ P))
    assert np.abs(np.tril(QtP, -1)).max() < 1e-11 * np.sqrt(mk)
    assert np.abs(np.triu(QtP[:w]) - np.triu(out[:w])).max() < 1e-11 * np.sqrt(mk)
    assert np.abs(np.diag(T) - tau).max() == 0.0 and np.abs(np.tril(T, -1)).max() == 0.0


def test_panel_cholqr2_moderate_condition_stays_accurate(q, oracle):
    """cond ~ 1e3: inside the range where the fast path is taken; orthogonality and residual must be at Householder level."""
    mk, w = 20000, 32
    rng = np.random.default_rng(11)
    U, _ = np.linalg.qr(rng.standard_normal((mk, w))); Vr, _ = np.linalg.qr(rng.standard_normal((w, w)))
    P = (U * np.logspace(0, -3, w)) @ Vr.T
    out, tau, T, V, guard, _ = _cholqr_leaf(q, P)
    R = np.triu(out[:w])
    QtP = P - V @ (T.T @ (V.T @ P))
    assert np.abs(QtP[:w] - R).max() < 1e-13 * np.sqrt(mk) and np.abs(QtP[w:]).max() < 1e-13 * np.sqrt(mk)
    ref = oracle.sign_normalise(np.linalg.qr(P, mode="r"))
    assert np.linalg.norm(oracle.sign_normalise(R) - ref) / np.linalg.norm(ref) < 1e-11


def test_leaf_panel_zero_column_gives_tau_zero(q):
    """Deviation from the reference stated in include/mi355x_qr.h: zero tail -> tau = 0 (reference: NaN).  The Householder leaf."""
    mk, w = 300, 8
    P = np.random.default_rng(1).random((mk, w))
    P[:, 3] = 0.0
    P[4:, 5] = 0.0
    dP, dtau, dT, dV = dev(P), zeros(w, 1), zeros(w, w), zeros(mk, w)
    ws = torch.zeros(int(q.lib.qrd_panel_ws_size(mk)), dtype=torch.float64, device="cuda")
    torch.cuda.synchronize()
    q.check(q.lib.qrd_panel_tsqr(None, dP.data_ptr(), mk, mk, w, dtau.data_ptr(), dT.data_ptr(), w, dV.data_ptr(),
                                 mk, ws.data_ptr(), mk))
    _sync(q)
    out, tau = host(dP), host(dtau)[:, 0]
    assert np.isfinite(out).all() and np.isfinite(tau).all()
    V, T = host(dV), host(dT)
    QtP = P - V @ (T.T @ (V.T @ P))
    assert np.abs(np.tril(QtP, -1)).max() < 1e-12


@pytest.mark.parametrize("nbp,ib,build_diag", [(128, 32, 1), (128, 32, 0), (256, 32, 1), (96, 16, 1), (64, 32, 0),
                                               (40, 32, 1), (32, 32, 1), (8, 8, 1), (72, 8, 0)])
def test_larft(q, oracle, nbp, ib, build_diag):
    rng = np.random.default_rng(nbp + ib)
    mk = 3 * nbp + 17
    V = np.tril(rng.standard_normal((mk, nbp)), -1) * 0.3
    V[np.arange(nbp), np.arange(nbp)] = 1.0
    tau = 2.0 / (V * V).sum(axis=0)                    # genuine Householder scalars
    G = V.T @ V
    Tref = oracle.np_larft(V, tau)
    T0 = np.full((nbp, nbp), np.nan)
    for cb in range(0, nbp, ib):                       # diagonal blocks as the leaf kernels would leave them
        wb = min(ib, nbp - cb)
        T0[cb:cb + wb, cb:cb + wb] = Tref[cb:cb + wb, cb:cb + wb] if not build_diag else np.nan
    T0[np.tril_indices(nbp, -1)] = np.where(np.isnan(T0[np.tril_indices(nbp, -1)]), 0.0, T0[np.tril_indices(nbp, -1)])
    for cb in range(0, nbp, ib):                       # sub-diagonal blocks are zero in the plan's T buffer
        T0[cb + ib:, cb:cb + ib] = 0.0
    dG, dtau, dT, dTt, dX = dev(G), dev(tau[:, None]), dev(T0), zeros(nbp, nbp), zeros(nbp, nbp)
    torch.cuda.synchronize()
    q.check(q.lib.qrd_larft(None, nbp, ib, dG.data_ptr(), nbp, dtau.data_ptr(), dT.data_ptr(), nbp, dTt.data_ptr(),
                            build_diag, dX.data_ptr(), nbp))
    _sync(q)
    T = host(dT)
    assert np.abs(T - Tref).max() < 1e-12 * max(1.0, np.abs(Tref).max())
    assert np.array_equal(host(dTt), T.T)
    H = np.eye(mk) - V @ T @ V.T
    assert np.abs(H.T @ H - np.eye(mk)).max() < 1e-11


def test_fill_uniform_matches_host_hash(q):
    p = q.Plan(64, 32)
    rows, cols, ld, off, total = 1000, 7, 1003, 12345, 99999
    d = zeros(ld, cols)
    torch.cuda.synchronize()
    p.fill_uniform(d, ld, rows, cols, row_off=off, total_rows=total, seed=12)
    p.sync()
    out = host(d)
    assert np.array_equal(out[:rows], q.uniform_matrix_host(rows, cols, off, total, 12))
    assert np.array_equal(out[rows:], np.zeros((ld - rows, cols)))
    p.close()


def test_diffnorm(q):
    p = q.Plan(64, 32)
    rng = np.random.default_rng(0)
    X, Y = rng.standard_normal((777, 13)), rng.standard_normal((777, 13))
    dX, dY = dev(X), dev(Y)
    torch.cuda.synchronize()
    a, b = p.diffnorm(dX, 777, 777, 13, dY=dY, ldy=777)
    assert abs(a - ((X - Y) ** 2).sum()) < 1e-9 and abs(b - (Y ** 2).sum()) < 1e-9
    G = q.uniform_matrix_host(777, 13, 5, 1000, 12)
    a, b = p.diffnorm(dX, 777, 777, 13, row_off=5, total_rows=1000, seed=12)
    assert abs(a - ((X - G) ** 2).sum()) < 1e-9 and abs(b - (G ** 2).sum()) < 1e-9
    a, b = p.diffnorm(dX, 777, 13, 13, mode=1)
    assert abs(a - ((X[:13] - np.eye(13)) ** 2).sum()) < 1e-9 and abs(b - 13.0) < 1e-12
    p.close()


@pytest.mark.parametrize("mk,N,gy", [(64, 32, 0), (512, 96, 0), (1000, 64, 0), (4100, 224, 0), (7168, 224, 2), (7168, 96, 1),
                                     (33000, 96, 1), (131072, 64, 1)])
def test_leaf_update_gram(q, mk, N, gy):
    """Fused in-panel update of the leaf chain (qr.c:215-235): A_rest -= V W on the matrix cores and, from the same accumulators,
    the partial Gram matrices of the NEXT leaf's 32 columns (rows below the current leaf's 32 x 32 diagonal block) -- against
    numpy; the padding rows of A_rest (ld > mk) must stay untouched."""
    import ctypes as C
    rng = np.random.default_rng(mk + N)
    V, W, A0 = rng.standard_normal((mk, 32)), rng.standard_normal((32, N)), rng.standard_normal((mk, N))
    ld = mk + 6
    buf = np.full((ld, N), 7.0); buf[:mk] = A0
    dV, dW, dA = dev(V), dev(W), dev(buf)
    cap = 1 << 20
    slabs = torch.full((cap,), np.nan, dtype=torch.float64, device="cuda")
    nslab = C.c_int(0)
    torch.cuda.synchronize()
    q.check(q.lib.qrd_leaf_update_gram(None, mk, N, dV.data_ptr(), mk, dW.data_ptr(), dA.data_ptr(), ld, slabs.data_ptr(), cap, gy,
                                       C.byref(nslab)))
    _sync(q)
    ref = A0 - V @ W
    out = host(dA)
    assert rel(out[:mk], ref) < 1e-14
    assert np.array_equal(out[mk:], np.full((ld - mk, N), 7.0))
    assert nslab.value == (mk + 511) // 512
    G = slabs[: nslab.value * 1024].cpu().numpy().reshape(nslab.value, 32, 32).sum(axis=0).T      # slabs are column-major 32 x 32
    Gref = ref[32:, :32].T @ ref[32:, :32]
    assert rel(G, Gref) < 1e-13
    # shapes the kernel does not take are refused, not mangled (the host then launches the plain product)
    assert q.lib.qrd_leaf_update_gram(None, mk + 2, N, dV.data_ptr(), mk, dW.data_ptr(), dA.data_ptr(), ld, None, 0, 0, None) == -7
    assert q.lib.qrd_leaf_update_gram(None, mk, N + 16, dV.data_ptr(), mk, dW.data_ptr(), dA.data_ptr(), ld, None, 0, 0, None) == -7


EP_CASES = [(1024, 224, 0), (4096, 96, 96), (8192, 0, 224), (8192, 32, 0), (6144, 160, 64), (16384, 192, 32), (70000 // 16 * 16, 96, 96),
            (262144, 64, 32), (131072, 64, 0), (2048, 448, 0)]


@pytest.mark.parametrize("kind", ["ok", "dependent"])
@pytest.mark.parametrize("mk,n1,n2", EP_CASES)
def test_panel_cholqr_early_product(q, oracle, mk, n1, n2, kind):
    """qrd_panel_cholqr_ep: the leaf AND its two long-K in-panel products in one call -- the products run on Q in the launch of the
    one-workgroup reconstruction (hr3_ep_kernel) and are folded to  W = T^T V^T A_rest  and  G = V_prev^T V  by slab_reduce_ep_kernel.
    Checked against the same quantities formed in numpy from the call's own V and T, on the fast route ("ok") and on the guard route
    ("dependent": two equal columns, the Householder TSQR forms V and the product is redone from it -- inside the one-launch guard
    route for short leaves, as a separate launch for tall ones)."""
    import ctypes as C
    w = 32
    rng = np.random.default_rng(mk + n1 + 3 * n2)
    P = rng.random((mk, w))
    if kind == "dependent":
        P[:, 21] = P[:, 3]
    B1 = rng.standard_normal((mk, max(n1, 1)))
    B2 = rng.standard_normal((mk, max(n2, 1))) / np.sqrt(mk)
    ld = mk + 2
    panel = np.full((ld, w + max(n1, 1)), 3.0)
    panel[:mk, :w] = P
    panel[:mk, w:] = B1
    dA, dB2 = dev(panel), dev(B2)
    dtau, dT, dV = zeros(w, 1), dev(np.full((w, w), np.nan)), dev(np.full((mk, w), np.nan))
    dW, dG = dev(np.full((w, max(n1, 1)), np.nan)), dev(np.full((max(n2, 1) + 5, w), np.nan))
    ws = torch.zeros(int(q.lib.qrd_panel_ws_size(mk)), dtype=torch.float64, device="cuda")
    cws = torch.zeros(10 * 32 * 32 + 16, dtype=torch.float64, device="cuda")
    slabs = torch.zeros(1 << 22, dtype=torch.float64, device="cuda")
    eps = torch.zeros(1 << 20, dtype=torch.float64, device="cuda")
    torch.cuda.synchronize()
    did = C.c_int(-1)
    f = q.lib.qrd_panel_cholqr_ep
    f.restype = C.c_int
    f.argtypes = [C.c_void_p] * 2 + [C.c_int] * 3 + [C.c_void_p] * 2 + [C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p,
                                                                         C.c_size_t, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_int,
                                                                         C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_size_t, C.POINTER(C.c_int)]
    q.check(f(None, dA.data_ptr(), ld, mk, w, dtau.data_ptr(), dT.data_ptr(), w, dV.data_ptr(), mk, ws.data_ptr(), mk, cws.data_ptr(),
              slabs.data_ptr(), 1 << 22, 0, n1, dA.data_ptr() + 8 * ld * w, ld, n2, dB2.data_ptr(), mk, dW.data_ptr(), w, dG.data_ptr(),
              max(n2, 1) + 5, eps.data_ptr(), 1 << 20, C.byref(did)))
    _sync(q)
    guard = int(cws[4 * 32 * 32:].view(torch.int32)[0].item())
    # the fused launch is only used while the product's operands stay below ~128 MB (a tall leaf streams faster as a launch of its own)
    expect_ep = 8 * mk * (32 + n1 + n2) <= 128e6
    assert did.value == (1 if expect_ep else 0) and guard == (1 if kind == "dependent" else 0)
    out, tau, T, V = host(dA)[:mk, :w], host(dtau)[:, 0], host(dT), host(dV)
    assert np.isfinite(V).all() and np.isfinite(T).all()
    assert np.array_equal(np.triu(V[:w], 1), np.zeros((w, w))) and np.array_equal(np.diag(V[:w]), np.ones(w))
    assert np.array_equal(np.tril(V, -1), np.tril(out, -1)), "explicit V and in-place tails must agree"
    QtP = P - V @ (T.T @ (V.T @ P))
    assert np.abs(np.tril(QtP, -1)).max() < 1e-11 * np.sqrt(mk) and np.abs(np.triu(QtP[:w]) - np.triu(out[:w])).max() < 1e-11 * np.sqrt(mk)
    assert np.array_equal(host(dA)[:mk, w:], B1), "the rest of the panel is read, never written"
    if not expect_ep:
        return
    if n1:
        Wref = T.T @ (V.T @ B1)
        assert np.abs(host(dW) - Wref).max() < 1e-12 * np.sqrt(mk) * max(1.0, np.abs(Wref).max())
    if n2:
        Gref = B2.T @ V
        G = host(dG)
        assert np.abs(G[:n2] - Gref).max() < 1e-12 * np.sqrt(mk) * max(1.0, np.abs(Gref).max())
        assert np.isnan(G[n2:]).all(), "rows of the Gram buffer beyond N2 stay untouched"


def test_panel_cholqr_early_product_declines_odd_heights(q):
    """heights that are not whole k-tiles: the leaf is factored, the products are left to the caller (did = 0)"""
    import ctypes as C
    mk, w, n1 = 1000, 32, 64
    rng = np.random.default_rng(1)
    panel = rng.random((mk, w + n1))
    dA = dev(panel)
    dtau, dT, dV, dW = zeros(w, 1), zeros(w, w), zeros(mk, w), dev(np.full((w, n1), np.nan))
    ws = torch.zeros(int(q.lib.qrd_panel_ws_size(mk)), dtype=torch.float64, device="cuda")
    cws = torch.zeros(10 * 32 * 32 + 16, dtype=torch.float64, device="cuda")
    slabs = torch.zeros(1 << 20, dtype=torch.float64, device="cuda")
    eps = torch.zeros(1 << 20, dtype=torch.float64, device="cuda")
    torch.cuda.synchronize()
    did = C.c_int(-1)
    f = q.lib.qrd_panel_cholqr_ep
    q.check(f(None, dA.data_ptr(), mk, mk, w, dtau.data_ptr(), dT.data_ptr(), w, dV.data_ptr(), mk, ws.data_ptr(), mk, cws.data_ptr(),
              slabs.data_ptr(), 1 << 20, 0, n1, dA.data_ptr() + 8 * mk * w, mk, 0, None, mk, dW.data_ptr(), w, None, w, eps.data_ptr(), 1 << 20,
              C.byref(did)))
    _sync(q)
    assert did.value == 0 and np.isnan(host(dW)).all()
    V, T = host(dV), host(dT)
    P = panel[:, :w]
    assert np.abs(np.tril(P - V @ (T.T @ (V.T @ P)), -1)).max() < 1e-11


@pytest.mark.parametrize("M,N,K", [(1280, 256, 4096), (1300, 512, 2064), (128, 256, 512), (3000, 256, 20000), (640, 128, 1024), (2048, 256, 1000)])
@pytest.mark.parametrize("entry", ["qrd_gemm_tn_update", "qrd_gemm_tn_update_wide"])
def test_gemm_tn_update_wide_tiles(q, M, N, K, entry):
    """The wide product of the trailing update, Wt = A2^T (V T): the default 128 x 128 kernel (qrd_gemm_tn_update) and the 128 x 256
    workgroup tiles of gemm_tn_wide_kernel (qrd_gemm_tn_update_wide: A2 read half as often; the same rate, so not the default) on the
    interior with the ragged rows on the guarded 128 x 128 kernel and the same K slices; split-K slabs summed in a fixed order --
    against numpy, and bitwise reproducible run to run.  N = 128 and K not a multiple of 16 take the 128 x 128 kernel throughout."""
    rng = np.random.default_rng(M + N + K)
    A, B = rng.standard_normal((K, M)), rng.standard_normal((K, N))
    dA, dB = dev(A), dev(B)
    slabs = torch.zeros(1 << 24, dtype=torch.float64, device="cuda")
    outs = []
    for rep in range(2):
        dC = dev(np.full((M, N), np.nan))
        torch.cuda.synchronize()
        q.check(getattr(q.lib, entry)(None, M, N, K, 1.0, dA.data_ptr(), K, dB.data_ptr(), K, 0.0, dC.data_ptr(), M, slabs.data_ptr(), 1 << 24))
        _sync(q)
        outs.append(host(dC))
    ref = A.T @ B
    assert np.abs(outs[0] - ref).max() < 1e-12 * np.sqrt(K) * max(1.0, np.abs(ref).max())
    assert np.array_equal(outs[0], outs[1])


@pytest.mark.parametrize("K,N1,N2", [(4096, 96, 64), (8192, 224, 0), (2048, 0, 96), (100352, 96, 96), (131072, 32, 0), (262144, 64, 128), (98304, 0, 224),
                                     (120000, 160, 64)])
def test_gemm_tn_dual_short_and_tall(q, K, N1, N2):
    """The leaf's two long-K in-panel products in one call, [W | G^T] = T^T V^T [A_rest | V_prev] (qrd_gemm_tn_dual: split-K 32 x 32
    tiles, K slices dealt to the XCDs, T^T folded into the slab reduce) at short and tall leaf heights -- against numpy, bitwise
    reproducible."""
    rng = np.random.default_rng(K + N1 + 7 * N2)
    V = rng.standard_normal((K, 32)) / np.sqrt(K)
    B1 = rng.standard_normal((K, max(N1, 1)))
    B2 = rng.standard_normal((K, max(N2, 1)))
    T = np.triu(rng.standard_normal((32, 32)))
    dV, dB1, dB2, dT = dev(V), dev(B1), dev(B2), dev(T)
    cap = 1 << 23
    slabs = torch.zeros(cap, dtype=torch.float64, device="cuda")
    outs = []
    for rep in range(2):
        dW, dG = dev(np.full((32, max(N1, 1)), np.nan)), dev(np.full((max(N2, 1) + 3, 32), np.nan))
        torch.cuda.synchronize()
        q.check(q.lib.qrd_gemm_tn_dual(None, N1, N2, K, dV.data_ptr(), K, dB1.data_ptr(), K, dB2.data_ptr(), K, dT.data_ptr(), 32, dW.data_ptr(), 32,
                                       dG.data_ptr(), max(N2, 1) + 3, slabs.data_ptr(), cap))
        _sync(q)
        outs.append((host(dW), host(dG)))
    if N1:
        ref = T.T @ (V.T @ B1)
        assert np.abs(outs[0][0] - ref).max() < 1e-12 * max(1.0, np.abs(ref).max()) * np.sqrt(K)
        assert np.array_equal(outs[0][0], outs[1][0])
    if N2:
        ref = B2.T @ V
        assert np.abs(outs[0][1][:N2] - ref).max() < 1e-12 * max(1.0, np.abs(ref).max()) * np.sqrt(K)
        assert np.array_equal(outs[0][1][:N2], outs[1][1][:N2])
        assert np.isnan(outs[0][1][N2:]).all()


@pytest.mark.parametrize("kw,nc,lds", [(256, 256, (256, 256, 256, 256)), (128, 16, (200, 130, 128, 160)), (64, 48, (64, 64, 70, 64)), (32, 32, (40, 32, 32, 32)),
                                       (96, 256, (96, 100, 96, 96)), (224, 64, (256, 256, 224, 224))])
def test_trsm_gt_applies_t_transposed_without_the_merged_t(q, kw, nc, lds):
    """qrd_trsm_gt (round 6): W = T^T Y from the panel's Gram matrix G = V^T V and the leaves' own 32 x 32 T blocks -- for a compact-WY panel
    T^-1 = striu(G) + diag(G) / 2, so W is a block forward substitution over the leaves and the merge tree (qrd_larft) is not needed.
    Checked against T^T Y with the full T formed in numpy; W written over Y as well."""
    rng = np.random.default_rng(kw + nc)
    V = np.tril(rng.standard_normal((3 * kw, kw)), -1)[:, :]      # unit lower trapezoidal reflectors, tails of moderate size
    V[np.arange(kw), np.arange(kw)] = 1.0
    V[kw:] *= 0.3
    G = V.T @ V
    Tinv = np.triu(G, 1) + np.diag(np.diag(G) / 2.0)
    T = np.linalg.inv(Tinv)
    Tin = np.full((kw, kw), np.nan)                                # only the leaves' diagonal blocks may be read
    for c in range(0, kw, 32):
        Tin[c:c + 32, c:c + 32] = np.triu(T[c:c + 32, c:c + 32])
    Y = rng.standard_normal((kw, nc))
    ldg, ldt, ldy, ldw = lds
    pad = lambda M, ld: np.vstack([M, np.full((ld - M.shape[0], M.shape[1]), np.nan)])
    dG, dT, dY, dW = dev(pad(G, ldg)), dev(pad(Tin, ldt)), dev(pad(Y, ldy)), dev(np.full((ldw, nc), np.nan))
    assert q.lib.qrd_trsm_gt(None, kw, nc, dG.data_ptr(), ldg, dT.data_ptr(), ldt, dY.data_ptr(), ldy, dW.data_ptr(), ldw) == 0
    q.check(q.lib.qrd_device_sync(), "sync")
    Wref = T.T @ Y
    W = host(dW)[:kw]
    assert np.isfinite(W).all()
    assert rel(W, Wref) < 1e-13
    if ldy == ldw or True:                                         # in place: W over Y
        assert q.lib.qrd_trsm_gt(None, kw, nc, dG.data_ptr(), ldg, dT.data_ptr(), ldt, dY.data_ptr(), ldy, dY.data_ptr(), ldy) == 0
        q.check(q.lib.qrd_device_sync(), "sync")
        assert rel(host(dY)[:kw], Wref) < 1e-13
    # shapes it does not take are declined, not mangled
    assert q.lib.qrd_trsm_gt(None, 48, 16, dG.data_ptr(), ldg, dT.data_ptr(), ldt, dY.data_ptr(), ldy, dW.data_ptr(), ldw) == -7
    assert q.lib.qrd_trsm_gt(None, 32, 24, dG.data_ptr(), ldg, dT.data_ptr(), ldt, dY.data_ptr(), ldy, dW.data_ptr(), ldw) == -7
