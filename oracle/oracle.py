"""oracle/oracle.py -- TEST INFRASTRUCTURE ONLY.

ctypes access to
  * liboracle.so            our CPU restatement of the reference qr.c (oracle/mmqr_oracle.c)
  * _ref/libqrref_*.so      the REAL reference qr.c compiled from /root/reference (oracle/Makefile),
                            present only if it was built in the dev container (it travels to the
                            GPU box as a prebuilt, git-ignored binary).
plus a numpy mirror of the blocked compact-WY algorithm the HIP path implements
(geqr2 / larft / larfb / orgqr shaped), used to unit-test individual kernels.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
The product (cuda-qr_amd/) never does.
"""
import ctypes as C
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
_libc = C.CDLL(None)
_libc.free.argtypes = [C.c_void_p]
_libc.fflush.argtypes = [C.c_void_p]


def build(quiet=True):
    """Compile liboracle.so (always) and _ref/ (only where /root/reference exists)."""
    out = subprocess.run(["make", "-C", HERE, "all"], capture_output=True, text=True)
    if out.returncode != 0:
        raise RuntimeError("oracle build failed:\n" + out.stdout + out.stderr)
    if not quiet:
        print(out.stdout)


_oracle = None


def lib():
    global _oracle
    if _oracle is None:
        path = os.path.join(HERE, "liboracle.so")
        if not os.path.exists(path):
            build()
        _oracle = C.CDLL(path)
    return _oracle


def _ct(dtype):
    dtype = np.dtype(dtype)
    if dtype == np.float64:
        return C.c_double, "_d"
    if dtype == np.float32:
        return C.c_float, "_f"
    raise TypeError(dtype)


def _ptr(a, ct):
    return a.ctypes.data_as(C.POINTER(ct))


# ----------------------------------------------------------------------------------------------
# restatement (port)
# ----------------------------------------------------------------------------------------------
def panel_dims(m, n, PR, PC):
    rp, cp = C.c_int(), C.c_int()
    lib().oracle_panel_dims_d(m, n, PR, PC, C.byref(rp), C.byref(cp))
    return rp.value, cp.value


def check_shape(m, n, PR, PC):
    """The constraints the reference silently assumes."""
    if not (m >= n and m >= PR and n % PC == 0 and (m - PR) % (PR - PC) == 0 and PR > PC):
        raise ValueError(f"reference MMQR cannot factor {m}x{n} with a {PR}x{PC} window")


def fill_rand(m, n, seed=12, dtype=np.float64):
    """The reference generator, qr.c:468-474 (column-major linear order)."""
    ct, suf = _ct(dtype)
    A = np.empty((m, n), dtype=dtype, order="F")
    f = getattr(lib(), "oracle_fill_rand" + suf)
    f.argtypes = [C.POINTER(ct), C.c_size_t, C.c_uint]
    f(_ptr(A, ct), m * n, seed)
    return A


def mmqr(A, PR, PC):
    """Restated MMQR.  Returns (factored copy of A, tau array, number of windows)."""
    A = np.array(A, order="F", copy=True)
    m, n = A.shape
    check_shape(m, n, PR, PC)
    ct, suf = _ct(A.dtype)
    rp, cp = panel_dims(m, n, PR, PC)
    tau = np.zeros(rp * cp * PC, dtype=A.dtype)
    f = getattr(lib(), "oracle_mmqr" + suf)
    f.argtypes = [C.POINTER(ct), C.POINTER(ct), C.c_int, C.c_int, C.c_int, C.c_int]
    f.restype = C.c_int
    windows = f(_ptr(A, ct), _ptr(tau, ct), m, n, PR, PC)
    return A, tau, windows


def explicit_qr(F, tau, PR, PC, faithful=False):
    """Q (m x m), R (m x n) from restated-MMQR output."""
    F = np.asfortranarray(F)
    m, n = F.shape
    ct, suf = _ct(F.dtype)
    Q = np.empty((m, m), dtype=F.dtype, order="F")
    R = np.empty((m, n), dtype=F.dtype, order="F")
    tau = np.ascontiguousarray(tau, dtype=F.dtype)
    f = getattr(lib(), "oracle_explicit_qr" + suf)
    f.argtypes = [C.POINTER(ct)] * 4 + [C.c_int] * 5
    f.restype = None
    f(_ptr(F, ct), _ptr(tau, ct), _ptr(Q, ct), _ptr(R, ct), m, n, PR, PC, int(bool(faithful)))
    return Q, R


def qr(A, PR=64, PC=8):
    """Convenience: full reference-path QR (restatement) -> Q (m x m), R (m x n)."""
    F, tau, _ = mmqr(A, PR, PC)
    return explicit_qr(F, tau, PR, PC, faithful=False)


# ----------------------------------------------------------------------------------------------
# the real reference (oracle/_ref)
# ----------------------------------------------------------------------------------------------
def ref_path(dtype, PR, PC):
    tag = "f64" if np.dtype(dtype) == np.float64 else "f32"
    p = os.path.join(HERE, "_ref", f"libqrref_{tag}_{PR}x{PC}.so")
    return p if os.path.exists(p) else None


class _Quiet:
    """The reference prints from inside mmqr/explicitQR (qr.c:57-303); send fd 1 to /dev/null."""

    def __enter__(self):
        _libc.fflush(None)
        self.saved = os.dup(1)
        self.null = os.open(os.devnull, os.O_WRONLY)
        os.dup2(self.null, 1)

    def __exit__(self, *exc):
        _libc.fflush(None)
        os.dup2(self.saved, 1)
        os.close(self.saved)
        os.close(self.null)


def ref_mmqr(A, PR, PC):
    """Run the REAL reference mmqr (qr.c:55).  Returns (factored A, tau) or raises if not built."""
    A = np.array(A, order="F", copy=True)
    m, n = A.shape
    check_shape(m, n, PR, PC)
    path = ref_path(A.dtype, PR, PC)
    if path is None:
        raise FileNotFoundError(f"oracle/_ref build for {A.dtype} {PR}x{PC} not present")
    ct, _ = _ct(A.dtype)
    ref = C.CDLL(path)
    ref.mmqr.argtypes = [C.POINTER(ct), C.POINTER(C.POINTER(ct)), C.c_int, C.c_int]
    ref.mmqr.restype = None
    rp, cp = panel_dims(m, n, PR, PC)
    tptr = C.POINTER(ct)()
    with _Quiet():
        ref.mmqr(_ptr(A, ct), C.byref(tptr), m, n)
    tau = np.ctypeslib.as_array(tptr, shape=(rp * cp * PC,)).copy()
    _libc.free(C.cast(tptr, C.c_void_p))
    return A, tau


def ref_explicit_qr(F, tau, PR, PC):
    """Run the REAL reference explicitQR (qr.c:330).  O(m^3) per reflector: small sizes only."""
    F = np.asfortranarray(F)
    m, n = F.shape
    path = ref_path(F.dtype, PR, PC)
    if path is None:
        raise FileNotFoundError("oracle/_ref not present")
    ct, _ = _ct(F.dtype)
    ref = C.CDLL(path)
    ref.explicitQR.argtypes = [C.POINTER(ct)] * 4 + [C.c_int, C.c_int]
    ref.explicitQR.restype = None
    Q = np.empty((m, m), dtype=F.dtype, order="F")
    R = np.empty((m, n), dtype=F.dtype, order="F")
    tau = np.ascontiguousarray(tau, dtype=F.dtype)
    with _Quiet():
        ref.explicitQR(_ptr(F, ct), _ptr(tau, ct), _ptr(Q, ct), _ptr(R, ct), m, n)
    return Q, R


# ----------------------------------------------------------------------------------------------
# implementation-independent comparison helpers (SURVEY section 8c "Parity definition")
# ----------------------------------------------------------------------------------------------
def sign_normalise(R):
    """Upper triangle with rows flipped so that diag(R) >= 0: the implementation-independent R
    (accepts a factored matrix whose sub-diagonal holds reflectors)."""
    R = np.triu(np.array(R[: R.shape[1], :], dtype=np.float64))
    s = np.where(np.diag(R) < 0, -1.0, 1.0)
    return s[:, None] * R


def flops(m, n):
    """Householder QR factorisation flop count used for every GFLOP/s figure."""
    return 2.0 * m * n * n - 2.0 * n ** 3 / 3.0


# ----------------------------------------------------------------------------------------------
# numpy mirror of the blocked compact-WY algorithm (what the HIP kernels implement)
# ----------------------------------------------------------------------------------------------
def np_geqr2(P):
    """Unblocked Householder QR of a panel, LAPACK dlarfg convention (tau = 0 for a zero tail).
    Returns (factored panel: R on/above diagonal, v below with implicit 1), tau."""
    P = np.array(P, dtype=np.float64, order="F", copy=True)
    m, w = P.shape
    tau = np.zeros(w)
    for j in range(min(m, w)):
        alpha = P[j, j]
        sigma = float(P[j + 1:, j] @ P[j + 1:, j])
        if sigma == 0.0:
            tau[j] = 0.0
            continue
        beta = -np.copysign(np.sqrt(alpha * alpha + sigma), alpha)
        tau[j] = (beta - alpha) / beta
        P[j + 1:, j] /= (alpha - beta)
        P[j, j] = beta
        if j + 1 < w:
            v = np.concatenate(([1.0], P[j + 1:, j]))
            s = v @ P[j:, j + 1:]
            P[j:, j + 1:] -= tau[j] * np.outer(v, s)
    return P, tau


def np_unit_lower(P):
    """Explicit V (unit lower trapezoid) from a factored panel."""
    m, w = P.shape
    V = np.tril(P, -1)
    V[np.arange(min(m, w)), np.arange(min(m, w))] = 1.0
    return V


def np_larft(V, tau):
    """Forward columnwise T: I - V T V^T = H_0 H_1 ... H_{w-1}."""
    w = V.shape[1]
    T = np.zeros((w, w))
    G = V.T @ V
    for j in range(w):
        T[j, j] = tau[j]
        if j:
            T[:j, j] = -tau[j] * (T[:j, :j] @ G[:j, j])
    return T


def np_geqrf(A, nb=32):
    """Blocked right-looking Householder QR.  Returns factored A (LAPACK layout), tau, list of T."""
    A = np.array(A, dtype=np.float64, order="F", copy=True)
    m, n = A.shape
    kmax = min(m, n)
    tau = np.zeros(kmax)
    Ts = []
    for k in range(0, kmax, nb):
        w = min(nb, kmax - k)
        P, t = np_geqr2(A[k:, k:k + w])
        A[k:, k:k + w] = P
        tau[k:k + w] = t
        V = np_unit_lower(P)
        T = np_larft(V, t)
        Ts.append(T)
        if k + w < n:
            Wm = (V @ T).T @ A[k:, k + w:]        # T^T V^T A2
            A[k:, k + w:] -= V @ Wm
    return A, tau, Ts


def np_orgqr(F, tau, ncols, nb=32):
    """Explicit Q[:, :ncols] from LAPACK-layout factors (backward block accumulation)."""
    m, n = F.shape
    kmax = min(m, n)
    Q = np.eye(m, ncols)
    starts = list(range(0, kmax, nb))
    for k in reversed(starts):
        w = min(nb, kmax - k)
        V = np_unit_lower(F[k:, k:k + w])
        T = np_larft(V, tau[k:k + w])
        Wm = (V @ T.T).T @ Q[k:, k:]              # T V^T Q
        Q[k:, k:] -= V @ Wm
    return Q


def np_tsqr(shards, nb=32):
    """TSQR over row shards: returns (R_final sign-free, list of thin Q shards)."""
    locs = [np_geqrf(S, nb) for S in shards]
    n = shards[0].shape[1]
    Rs = [np.triu(F[:n, :]) for F, _, _ in locs]
    stack = np.vstack(Rs)
    Fs, ts, _ = np_geqrf(stack, nb)
    R = np.triu(Fs[:n, :])
    Qt = np_orgqr(Fs, ts, n, nb)
    Qs = []
    for p, (F, t, _) in enumerate(locs):
        Ql = np_orgqr(F, t, n, nb)
        Qs.append(Ql @ Qt[p * n:(p + 1) * n, :])
    return R, Qs
