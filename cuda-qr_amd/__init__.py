"""cuda-qr_amd: thin ctypes binding over the C-ABI library libmi355xqr.so (include/mi355x_qr.h).

This Python layer is plumbing for tests and bench.py -- the product is the C library (C host layer
+ hand-written gfx950 HIP kernels).  There is NO CPU fallback: importing works anywhere the shared
library loads, but every compute entry point needs an MI355X and fails loudly otherwise.

The directory name has a hyphen (it mirrors the reference repo's name), so import it through the
root-level shim:  `import cuda_qr_amd`.
"""
import ctypes as C
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
# CUDA_QR_AMD_LIB=lab (this binding's own switch, not a library knob): load libmi355xqr_lab.so -- the same sources built with -DQR_LAB
# (`make -C cuda-qr_amd lab`), in which the measurement knobs are environment variables and the development entry points of the 32 x 32
# factor core exist.  For devtools/ scripts and the tests that force a schedule branch on a small matrix; everything else (bench.py,
# smoke(), the parity tests) runs the product library.
_which = os.environ.get("CUDA_QR_AMD_LIB", "")
LAB = _which == "lab" or _which.endswith(".so")        # (a path: an experimental build of the lab flavour, devtools/ A/Bs of two kernel versions)
LAB_LIB_PATH = os.path.join(HERE, "libmi355xqr_lab.so")
LIB_PATH = (os.path.join(HERE, _which) if _which.endswith(".so") else LAB_LIB_PATH) if LAB else os.path.join(HERE, "libmi355xqr.so")
HEADER = os.path.join(os.path.dirname(HERE), "include", "mi355x_qr.h")

if not os.path.exists(LIB_PATH):
    raise ImportError(
        f"{LIB_PATH} is missing: build it with `make -C {HERE} all lab` (or python -c 'import __graft_entry__ as g; "
        "g.build()').  The HIP extension is mandatory; there is no CPU fallback.")

# Load order matters when torch is used in the same process (every caller in this repo: device buffers are torch tensors): the torch
# wheel bundles its own libamdhip64, and whichever HIP runtime is mapped first serves both -- with the system one mapped first
# torch.cuda later reports "No HIP GPUs are available".  So torch goes first when it is installed (plumbing only: nothing below uses it).
try:
    import torch  # noqa: F401
except Exception:      # a host without torch: the binding works on raw device pointers
    pass

lib = C.CDLL(LIB_PATH)

QR_PROF_CLASSES = 4
PROF_NAMES = ("update_nn", "vta_tn", "panel", "vt_misc")


class QRError(RuntimeError):
    pass


class Profile(C.Structure):
    _fields_ = [("ms", C.c_double * QR_PROF_CLASSES), ("flops", C.c_double * QR_PROF_CLASSES),
                ("bytes", C.c_double * QR_PROF_CLASSES), ("launches", C.c_longlong * QR_PROF_CLASSES)]


_dp = C.POINTER(C.c_double)
_vp = C.c_void_p


def _sig(name, restype, *argtypes):
    f = getattr(lib, name)
    f.restype = restype
    f.argtypes = list(argtypes)
    return f


_sig("getPanelDims", None, C.c_int, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int))
_sig("mmqr_status", C.c_int, _dp, C.POINTER(_dp), C.c_int, C.c_int)
_sig("mmqr", None, _dp, C.POINTER(_dp), C.c_int, C.c_int)
_sig("explicitQR_status", C.c_int, _dp, _dp, _dp, _dp, C.c_int, C.c_int)
_sig("explicitQR", None, _dp, _dp, _dp, _dp, C.c_int, C.c_int)
_sig("dgemm_status", C.c_int, _dp, _dp, _dp, C.c_int, C.c_int, C.c_int)
_sig("dgemm", None, _dp, _dp, _dp, C.c_int, C.c_int, C.c_int)
_sig("identity", None, _dp, C.c_int)
_sig("printMat", None, _dp, C.c_int, C.c_int)
_sig("getPanelDims_legacy", None, C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int))
_sig("mmqr_legacy_status", C.c_int, _dp, C.POINTER(_dp), C.c_int, C.c_int, C.c_int, C.c_int)
_sig("explicitQR_legacy_status", C.c_int, _dp, _dp, _dp, _dp, C.c_int, C.c_int, C.c_int, C.c_int)
_sig("qr_strerror", C.c_char_p, C.c_int)
_sig("qr_set_block_size", C.c_int, C.c_int, C.c_int)
_sig("qr_get_block_size", None, C.POINTER(C.c_int), C.POINTER(C.c_int))
_sig("qr_default_block_size", C.c_int, C.c_int, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int))
_sig("qr_thin", C.c_int, _dp, C.c_int, C.c_int, _dp, _dp, C.c_int, C.c_int)
_fp = C.POINTER(C.c_float)
_sig("mmqr_f32_status", C.c_int, _fp, C.POINTER(_fp), C.c_int, C.c_int)
_sig("explicitQR_f32_status", C.c_int, _fp, _fp, _fp, _fp, C.c_int, C.c_int)
_sig("qr_thin_mgpu", C.c_int, _dp, C.c_int, C.c_int, _dp, _dp, C.c_int, C.c_int)
_sig("qr_release_cached_plans", C.c_int)
_sig("qr_tsqr_unique_id", C.c_int, _vp)
_sig("qr_tsqr_plan_create", C.c_int, C.POINTER(_vp), _vp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int)
_sig("qr_tsqr_plan_create_comm", C.c_int, C.POINTER(_vp), _vp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int)
_sig("qr_tsqr_plan_destroy", C.c_int, _vp)
_sig("qr_tsqr_factor_dev", C.c_int, _vp, _vp, C.c_int, _vp)
_sig("qr_tsqr_formq_dev", C.c_int, _vp, _vp, C.c_int, _vp, C.c_int)
_sig("qr_tsqr_local_dev", C.c_int, _vp, _vp, C.c_int)
_sig("qr_tsqr_exchange_buffers", C.c_int, _vp, C.POINTER(_vp), C.POINTER(_vp))
_sig("qr_tsqr_stacked_dev", C.c_int, _vp, _vp)
_sig("qr_tsqr_is_pipelined", C.c_int, _vp)
_sig("qr_tsqr_set_schedule", C.c_int, _vp, C.c_int)
_sig("qr_tsqr_gather_stats", C.c_int, _vp, C.POINTER(C.c_double))
_sig("qr_tsqr_factor_virtual_dev", C.c_int, C.POINTER(_vp), C.c_int, C.POINTER(_vp), C.c_int, C.POINTER(_vp))
_sig("qr_tsqr_factor_selfgather_dev", C.c_int, _vp, _vp, C.c_int, _vp)
_sig("qr_tsqr_sync", C.c_int, _vp)
_sig("qr_tsqr_stream", _vp, _vp)
_sig("qr_tsqr_comm_ranks", C.c_int, _vp, C.POINTER(C.c_int))
_sig("qr_tsqr_local_plan", _vp, _vp)
_sig("qr_tsqr_stacked_plan", _vp, _vp)
_sig("qr_plan_create", C.c_int, C.POINTER(_vp), C.c_int, C.c_int, C.c_int, C.c_int)
_sig("qr_plan_destroy", C.c_int, _vp)
_sig("qr_geqrf_dev", C.c_int, _vp, _vp, C.c_int, C.c_int, C.c_int, _vp)
_sig("qr_applyq_dev", C.c_int, _vp, _vp, C.c_int, C.c_int, C.c_int, _vp, _vp, C.c_int, C.c_int, C.c_int)
_sig("qr_extract_r_dev", C.c_int, _vp, _vp, C.c_int, C.c_int, C.c_int, _vp, C.c_int, C.c_int)
_sig("qr_gemm_dev", C.c_int, _vp, C.c_char, C.c_int, C.c_int, C.c_int, C.c_double, _vp, C.c_int, _vp, C.c_int,
     C.c_double, _vp, C.c_int)
_sig("qr_fill_uniform_dev", C.c_int, _vp, _vp, C.c_int, C.c_longlong, C.c_int, C.c_longlong, C.c_longlong,
     C.c_ulonglong)
_sig("qr_uniform_at", C.c_double, C.c_ulonglong, C.c_ulonglong)
_sig("qr_diffnorm_dev", C.c_int, _vp, _vp, C.c_int, _vp, C.c_int, C.c_longlong, C.c_int, C.c_longlong,
     C.c_longlong, C.c_ulonglong, C.c_int, _dp)
_sig("qr_device_malloc", C.c_int, C.POINTER(_vp), C.c_size_t)
_sig("qr_device_free", C.c_int, _vp)
_sig("qr_copy_to_device", C.c_int, _vp, _vp, C.c_size_t)
_sig("qr_copy_to_host", C.c_int, _vp, _vp, C.c_size_t)
_sig("qr_plan_sync", C.c_int, _vp)
_sig("qr_plan_stream", _vp, _vp)
_sig("qr_plan_set_guard_mode", C.c_int, _vp, C.c_int)
_sig("qr_plan_route_stats", C.c_int, _vp, C.POINTER(C.c_longlong))
_sig("qr_plan_retry_stats", C.c_int, _vp, C.POINTER(C.c_longlong))
_sig("qr_plan_info", C.c_int, _vp, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int))
_sig("qr_plan_update_cus", C.c_int, _vp)
_sig("qr_plan_set_profile", C.c_int, _vp, C.c_int)
_sig("qr_plan_pause_profile", C.c_int, _vp, C.c_int)
_sig("qr_plan_get_profile", C.c_int, _vp, C.POINTER(Profile))
_sig("qr_plan_get_profile_records", C.c_int, _vp, C.c_int, C.POINTER(C.c_int), _dp, _dp)
_sig("qr_device_info", C.c_int, C.c_char_p, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_size_t))
_sig("qr_probe_mfma_f64_tflops", C.c_int, _dp)
_sig("qr_probe_copy_gbps", C.c_int, _dp)
# internal launch layer (kernel unit tests only)
_sig("qrd_gemm_nn", C.c_int, _vp, C.c_int, C.c_int, C.c_int, C.c_double, _vp, C.c_int, _vp, C.c_int, C.c_double,
     _vp, C.c_int)
_sig("qrd_gemm_tn", C.c_int, _vp, C.c_int, C.c_int, C.c_int, C.c_double, _vp, C.c_int, _vp, C.c_int, C.c_double,
     _vp, C.c_int, _vp, C.c_size_t, _vp, C.c_int)
_sig("qrd_gemm_tn_update", C.c_int, _vp, C.c_int, C.c_int, C.c_int, C.c_double, _vp, C.c_int, _vp, C.c_int, C.c_double,
     _vp, C.c_int, _vp, C.c_size_t)
_sig("qrd_gemm_tn_update_wide", C.c_int, _vp, C.c_int, C.c_int, C.c_int, C.c_double, _vp, C.c_int, _vp, C.c_int, C.c_double,
     _vp, C.c_int, _vp, C.c_size_t)
_sig("qrd_gemm_tn_dual", C.c_int, _vp, C.c_int, C.c_int, C.c_int, _vp, C.c_int, _vp, C.c_int, _vp, C.c_int, _vp, C.c_int, _vp, C.c_int, _vp, C.c_int,
     _vp, C.c_size_t)
_sig("qrd_trsm_gt", C.c_int, _vp, C.c_int, C.c_int, _vp, C.c_int, _vp, C.c_int, _vp, C.c_int, _vp, C.c_int)
_sig("qrd_larft", C.c_int, _vp, C.c_int, C.c_int, _vp, C.c_int, _vp, _vp, C.c_int, _vp, C.c_int, _vp, C.c_int)
_sig("qrd_panel_ws_size", C.c_size_t, C.c_int)
_sig("qrd_panel_tsqr", C.c_int, _vp, _vp, C.c_int, C.c_int, C.c_int, _vp, _vp, C.c_int, _vp, C.c_int, _vp, C.c_int)
_sig("qrd_panel_cholqr", C.c_int, _vp, _vp, C.c_int, C.c_int, C.c_int, _vp, _vp, C.c_int, _vp, C.c_int, _vp, C.c_int, _vp, _vp, C.c_size_t, C.c_int)
_sig("qrd_leaf_update_gram", C.c_int, _vp, C.c_int, C.c_int, _vp, C.c_int, _vp, _vp, C.c_int, _vp, C.c_size_t, C.c_int, C.POINTER(C.c_int))
_sig("qrd_gemm_nt", C.c_int, _vp, C.c_int, C.c_int, C.c_int, C.c_int, _vp, C.c_int, _vp, C.c_int, _vp, C.c_int, C.c_int, _vp)
if hasattr(lib, "qrd_gemm_nt4_ok"):        # (absent from libraries of earlier commits loaded for A/B runs through CUDA_QR_AMD_LIB)
    _sig("qrd_gemm_nt4_ok", C.c_int, C.c_int, C.c_int, C.c_int, _vp, C.c_int, _vp, C.c_int, _vp, C.c_int)
_sig("qrd_copy_block", C.c_int, _vp, _vp, C.c_int, _vp, C.c_int, C.c_int, C.c_int)
_sig("qrd_init", C.c_int)
_sig("qrd_device_sync", C.c_int)


def strerror(rc):
    return lib.qr_strerror(rc).decode()


def check(rc, what=""):
    if rc != 0:
        raise QRError(f"{what or 'mi355xqr'} failed: {strerror(rc)} ({rc})")


def exported_symbols():
    """Names declared in include/mi355x_qr.h (parsed), for the 'library exports its header' test."""
    import re
    txt = open(HEADER).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    names = re.findall(r"^\s*(?:const\s+)?(?:int|void|double|char\*|void\*|const char\*|qr_plan\*)\s*\*?\s*(\w+)\s*\(", txt, flags=re.M)
    return sorted(set(names))


# ------------------------------------------------------------------------------------------------
# host-pointer drop-in calls (numpy in / numpy out)
# ------------------------------------------------------------------------------------------------
def _f(a):
    return np.asfortranarray(a, dtype=np.float64)


def _p(a):
    return a.ctypes.data_as(_dp)


def get_panel_dims(m, n):
    rp, cp = C.c_int(), C.c_int()
    lib.getPanelDims(m, n, C.byref(rp), C.byref(cp))
    return rp.value, cp.value


def get_block_size():
    nb, ib = C.c_int(), C.c_int()
    lib.qr_get_block_size(C.byref(nb), C.byref(ib))
    return nb.value, ib.value


def default_block_size(m, n):
    """(nb, ib) an m x n problem really gets from mmqr / a default plan (shape-dependent; sizes mmqr's tau)."""
    nb, ib = C.c_int(), C.c_int()
    check(lib.qr_default_block_size(m, n, C.byref(nb), C.byref(ib)), "qr_default_block_size")
    return nb.value, ib.value


def tau_len(m, n):
    """entries of the tau array mmqr mallocs for an m x n problem: rowPanels * colPanels * nb (qr.c:61 sizing rule)."""
    rp, cp = get_panel_dims(m, n)
    return rp * cp * default_block_size(m, n)[0]


def set_block_size(nb, ib):
    check(lib.qr_set_block_size(nb, ib), "qr_set_block_size")


_libc = C.CDLL(None)
_libc.free.argtypes = [C.c_void_p]


def mmqr(A):
    """reference mmqr (qr.c:55): returns (factored copy, tau array of colPanels*nb entries)."""
    F = np.array(A, dtype=np.float64, order="F", copy=True)
    m, n = F.shape
    tptr = _dp()
    check(lib.mmqr_status(_p(F), C.byref(tptr), m, n), "mmqr")
    tau = np.ctypeslib.as_array(tptr, shape=(tau_len(m, n),)).copy()
    _libc.free(C.cast(tptr, C.c_void_p))
    return F, tau


def explicit_qr(F, tau):
    """reference explicitQR (qr.c:330): Q (m x m), R (m x n)."""
    F = _f(F)
    m, n = F.shape
    tau = np.ascontiguousarray(tau, dtype=np.float64)
    if tau.size < n:
        raise QRError(f"explicit_qr: tau has {tau.size} entries, the factorisation of {n} columns has {n}")
    Q = np.empty((m, m), order="F")
    R = np.empty((m, n), order="F")
    check(lib.explicitQR_status(_p(F), _p(tau), _p(Q), _p(R), m, n), "explicitQR")
    return Q, R


def mmqr_legacy(A, PR, PC):
    """The reference's sliding-window MMQR itself (legacy-layout shim): (factored copy, window-indexed tau) for window PR x PC."""
    F = np.array(A, dtype=np.float64, order="F", copy=True)
    m, n = F.shape
    tptr = _dp()
    check(lib.mmqr_legacy_status(_p(F), C.byref(tptr), m, n, PR, PC), "mmqr_legacy")
    rp, cp = C.c_int(), C.c_int()
    lib.getPanelDims_legacy(m, n, PR, PC, C.byref(rp), C.byref(cp))
    tau = np.ctypeslib.as_array(tptr, shape=(rp.value * cp.value * PC,)).copy()
    _libc.free(C.cast(tptr, C.c_void_p))
    return F, tau


def explicit_qr_legacy(F, tau, PR, PC):
    F = _f(F)
    m, n = F.shape
    tau = np.ascontiguousarray(tau, dtype=np.float64)
    Q = np.empty((m, m), order="F")
    R = np.empty((m, n), order="F")
    check(lib.explicitQR_legacy_status(_p(F), _p(tau), _p(Q), _p(R), m, n, PR, PC), "explicitQR_legacy")
    return Q, R


def dgemm(A, B):
    """reference dgemm (qr.c:443): C (k x n) = A (k x m) B (m x n)."""
    A, B = _f(A), _f(B)
    k, m = A.shape
    m2, n = B.shape
    assert m == m2
    Cm = np.empty((k, n), order="F")
    check(lib.dgemm_status(_p(A), _p(B), _p(Cm), k, m, n), "dgemm")
    return Cm


def identity(m):
    A = np.empty((m, m), order="F")
    lib.identity(_p(A), m)
    return A


def qr_thin(A, nb=0, nshards=1):
    A = _f(A)
    m, n = A.shape
    Q = np.empty((m, n), order="F")
    R = np.empty((n, n), order="F")
    check(lib.qr_thin(_p(A), m, n, _p(Q), _p(R), nb, nshards), "qr_thin")
    return Q, R


def mmqr_f32(A):
    """Float instantiation of mmqr (reference Scalar = float, qr.c:11): returns (factored float matrix, float tau)."""
    F = np.asfortranarray(A, dtype=np.float32).copy(order="F")
    m, n = F.shape
    tau = _fp()
    check(lib.mmqr_f32_status(F.ctypes.data_as(_fp), C.byref(tau), m, n), "mmqr_f32")
    t = np.ctypeslib.as_array(tau, shape=(tau_len(m, n),)).copy()
    _libc.free(C.cast(tau, C.c_void_p))
    return F, t


def explicit_qr_f32(F, tau):
    F = np.asfortranarray(F, dtype=np.float32)
    m, n = F.shape
    t = np.ascontiguousarray(tau, dtype=np.float32)
    if t.size < n:
        raise QRError(f"explicit_qr_f32: tau has {t.size} entries, the factorisation of {n} columns has {n}")
    Q = np.empty((m, m), dtype=np.float32, order="F")
    R = np.empty((m, n), dtype=np.float32, order="F")
    check(lib.explicitQR_f32_status(F.ctypes.data_as(_fp), t.ctypes.data_as(_fp), Q.ctypes.data_as(_fp), R.ctypes.data_as(_fp), m, n),
          "explicitQR_f32")
    return Q, R


def qr_thin_mgpu(A, nb=0, ngpu=1):
    """Thin QR over `ngpu` devices of this node through the C-level TSQR entry (one host thread per GPU, one RCCL all-gather)."""
    A = _f(A)
    m, n = A.shape
    Q = np.empty((m, n), order="F")
    R = np.empty((n, n), order="F")
    check(lib.qr_thin_mgpu(_p(A), m, n, _p(Q), _p(R), nb, ngpu), "qr_thin_mgpu")
    return Q, R


def release_cached_plans():
    check(lib.qr_release_cached_plans(), "qr_release_cached_plans")


def device_info():
    name = C.create_string_buffer(64)
    cus, clk, mem = C.c_int(), C.c_int(), C.c_size_t()
    check(lib.qr_device_info(name, 64, C.byref(cus), C.byref(clk), C.byref(mem)), "qr_device_info")
    return {"arch": name.value.decode(), "compute_units": cus.value, "clock_khz": clk.value, "hbm_bytes": mem.value}


def probe_mfma_f64_tflops():
    """{mfma_tflops, mfma_clock_ghz, valu_tflops}: sustained fp64 rates measured on this device."""
    v = (C.c_double * 3)()
    check(lib.qr_probe_mfma_f64_tflops(v), "probe")
    return {"mfma_f64_tflops": v[0], "mfma_clock_ghz": v[1], "valu_f64_tflops": v[2]}


def probe_copy_gbps():
    v = C.c_double()
    check(lib.qr_probe_copy_gbps(C.byref(v)), "probe")
    return v.value


# ------------------------------------------------------------------------------------------------
# device-resident API.  Device buffers are anything with .data_ptr() (torch tensors) or raw ints.
# A column-major m x n matrix with leading dimension ld is a flat float64 buffer of ld*n elements;
# `colmajor(m, n)` makes one as a torch tensor of shape (n, m) whose .T is the matrix.
# ------------------------------------------------------------------------------------------------
def _dptr(x):
    if x is None:
        return None
    if isinstance(x, int):
        return x
    return x.data_ptr()


def colmajor(m, n, device="cuda"):
    import torch
    return torch.empty((n, m), dtype=torch.float64, device=device)


def to_device_colmajor(A, device="cuda"):
    """numpy (m x n) -> torch (n, m) buffer holding A column-major."""
    import torch
    t = torch.from_numpy(np.ascontiguousarray(np.asarray(A, dtype=np.float64).T)).to(device)
    torch.cuda.synchronize()      # the copy ran on torch's stream; the plan's streams are not ordered with it
    return t


def from_device_colmajor(t):
    return np.asfortranarray(t.detach().cpu().numpy().T)


class Plan:
    """qr_plan wrapper.  The plan's HIP stream is independent of torch's current stream: call
    torch.cuda.synchronize() (or Plan.sync) at the hand-over points."""

    def __init__(self, m, n, nb=0, ib=0, borrowed=None):
        self.h = None
        self.owned = borrowed is None
        if borrowed is not None:            # a qr_plan that lives inside another object (TsqrPlan): used, never destroyed here
            self.h, self.m, self.n = _vp(borrowed), m, n
            return
        h = _vp()
        check(lib.qr_plan_create(C.byref(h), m, n, nb, ib), "qr_plan_create")
        self.h, self.m, self.n = h, m, n

    def close(self):
        if self.h and self.owned:
            lib.qr_plan_destroy(self.h)
        self.h = None

    __del__ = close

    def sync(self):
        check(lib.qr_plan_sync(self.h), "qr_plan_sync")

    def set_guard_mode(self, latch):
        """latch = True: qr_geqrf_dev never waits for the device; a refused full-width tall panel is reported by sync() (QR_E_REFUSED)"""
        check(lib.qr_plan_set_guard_mode(self.h, int(bool(latch))), "qr_plan_set_guard_mode")

    def route_stats(self):
        out = (C.c_longlong * 4)()
        check(lib.qr_plan_route_stats(self.h, out), "qr_plan_route_stats")
        rt = (C.c_longlong * 2)()
        check(lib.qr_plan_retry_stats(self.h, rt), "qr_plan_retry_stats")
        return {"tall_panels": out[0], "tall_panels_refused": out[1], "fused_leaf_fallbacks": out[2], "fused_stalls": out[3],
                "tall_panels_retried": rt[0], "tall_panels_retry_accepted": rt[1]}

    @property
    def stream(self):
        return lib.qr_plan_stream(self.h)

    def info(self):
        nb, ib, la = C.c_int(), C.c_int(), C.c_int()
        check(lib.qr_plan_info(self.h, C.byref(nb), C.byref(ib), C.byref(la)), "qr_plan_info")
        return nb.value, ib.value, bool(la.value)

    @property
    def nb(self):
        return self.info()[0]

    def geqrf(self, dA, m, n, lda, dtau):
        check(lib.qr_geqrf_dev(self.h, _dptr(dA), m, n, lda, _dptr(dtau)), "qr_geqrf_dev")

    def applyq(self, dA, m, n, lda, dtau, dC, ccols, ldc, identity_start):
        check(lib.qr_applyq_dev(self.h, _dptr(dA), m, n, lda, _dptr(dtau), _dptr(dC), ccols, ldc,
                                int(identity_start)), "qr_applyq_dev")

    def extract_r(self, dA, m, n, lda, dR, rrows, ldr):
        check(lib.qr_extract_r_dev(self.h, _dptr(dA), m, n, lda, _dptr(dR), rrows, ldr), "qr_extract_r_dev")

    def gemm(self, trans, M, N, K, alpha, dA, lda, dB, ldb, beta, dC, ldc):
        check(lib.qr_gemm_dev(self.h, trans.encode(), M, N, K, alpha, _dptr(dA), lda, _dptr(dB), ldb, beta,
                              _dptr(dC), ldc), "qr_gemm_dev")

    def fill_uniform(self, dA, lda, rows, cols, row_off=0, total_rows=None, seed=12):
        check(lib.qr_fill_uniform_dev(self.h, _dptr(dA), lda, rows, cols, row_off,
                                      rows if total_rows is None else total_rows, seed), "qr_fill_uniform_dev")

    def diffnorm(self, dX, ldx, rows, cols, dY=None, ldy=0, row_off=0, total_rows=None, seed=12, mode=0):
        out = (C.c_double * 2)()
        check(lib.qr_diffnorm_dev(self.h, _dptr(dX), ldx, _dptr(dY), ldy, rows, cols, row_off,
                                  rows if total_rows is None else total_rows, seed, mode, out), "qr_diffnorm_dev")
        return out[0], out[1]

    def update_cus(self):
        return lib.qr_plan_update_cus(self.h)

    def set_profile(self, on):
        check(lib.qr_plan_set_profile(self.h, int(on)), "qr_plan_set_profile")

    def pause_profile(self, pause):
        check(lib.qr_plan_pause_profile(self.h, int(bool(pause))), "qr_plan_pause_profile")

    def get_profile_records(self, max_records=65536):
        """[(class, start_ms, end_ms)] of the last profiled run, in issue order (call before get_profile)."""
        cls = (C.c_int * max_records)()
        t0 = (C.c_double * max_records)()
        t1 = (C.c_double * max_records)()
        n = lib.qr_plan_get_profile_records(self.h, max_records, cls, t0, t1)
        if n < 0:
            check(n, "qr_plan_get_profile_records")
        return [(cls[i], t0[i], t1[i]) for i in range(n)]

    def get_profile(self):
        pr = Profile()
        check(lib.qr_plan_get_profile(self.h, C.byref(pr)), "qr_plan_get_profile")
        return {PROF_NAMES[c]: {"ms": pr.ms[c], "flops": pr.flops[c], "bytes": pr.bytes[c],
                                "launches": pr.launches[c]} for c in range(QR_PROF_CLASSES)}


class TsqrPlan:
    """qr_tsqr_plan wrapper: the device-resident TSQR step of one rank (include/mi355x_qr.h).  `unique_id`: the 128 bytes
    rank 0 got from tsqr_unique_id(), carried to every rank by the caller; None with nranks = 1; comm="external" builds a
    plan without a communicator (the caller exchanges the R factors itself: local / exchange_buffers / stacked)."""

    def __init__(self, m_local, n, nranks=1, rank=0, nb=0, unique_id=None, comm=None):
        self.h = None
        h = _vp()
        if comm == "external":
            check(lib.qr_tsqr_plan_create_comm(C.byref(h), None, nranks, rank, m_local, n, nb), "qr_tsqr_plan_create_comm")
        else:
            buf = None
            if nranks > 1:
                assert unique_id is not None and len(unique_id) == 128
                buf = C.create_string_buffer(bytes(unique_id), 128)
            check(lib.qr_tsqr_plan_create(C.byref(h), buf, nranks, rank, m_local, n, nb), "qr_tsqr_plan_create")
        self.h, self.m, self.n, self.nranks, self.rank = h, m_local, n, nranks, rank
        self.local = Plan(m_local, n, borrowed=lib.qr_tsqr_local_plan(h))
        sp = lib.qr_tsqr_stacked_plan(h)
        self.stacked = Plan(nranks * n, n, borrowed=sp) if sp else None

    def close(self):
        if self.h and lib is not None:      # (at interpreter shutdown the module's globals may already be gone)
            lib.qr_tsqr_plan_destroy(self.h)
            self.h = None

    __del__ = close

    def factor(self, dA, lda, dR):
        check(lib.qr_tsqr_factor_dev(self.h, _dptr(dA), lda, _dptr(dR)), "qr_tsqr_factor_dev")

    def formq(self, dA, lda, dQ, ldq):
        check(lib.qr_tsqr_formq_dev(self.h, _dptr(dA), lda, _dptr(dQ), ldq), "qr_tsqr_formq_dev")

    def factor_selfgather(self, dA, lda, dR):
        check(lib.qr_tsqr_factor_selfgather_dev(self.h, _dptr(dA), lda, _dptr(dR)), "qr_tsqr_factor_selfgather_dev")

    def set_schedule(self, mode):
        """0 = one collective, 1 = panel-pipelined, 2 = the library's rule; every rank must make the same call"""
        check(lib.qr_tsqr_set_schedule(self.h, int(mode)), "qr_tsqr_set_schedule")

    def is_pipelined(self):
        return bool(lib.qr_tsqr_is_pipelined(self.h))

    def gather_stats(self):
        """the exchange of the last pipelined factor(): per-gather intervals from the stacked stream's events (include/mi355x_qr.h)"""
        out = (C.c_double * 5)()
        check(lib.qr_tsqr_gather_stats(self.h, out), "qr_tsqr_gather_stats")
        return {"gather_ms": out[0], "gather_max_ms": out[1], "call_ms": out[2], "pipelined": bool(out[3]), "fell_back": bool(out[4])}

    def local_factor(self, dA, lda):
        check(lib.qr_tsqr_local_dev(self.h, _dptr(dA), lda), "qr_tsqr_local_dev")

    def exchange_buffers(self):
        s, r = _vp(), _vp()
        check(lib.qr_tsqr_exchange_buffers(self.h, C.byref(s), C.byref(r)), "qr_tsqr_exchange_buffers")
        return s.value, r.value

    def stacked_factor(self, dR):
        check(lib.qr_tsqr_stacked_dev(self.h, _dptr(dR)), "qr_tsqr_stacked_dev")

    def sync(self):
        check(lib.qr_tsqr_sync(self.h), "qr_tsqr_sync")

    def comm_ranks(self):
        n = C.c_int()
        check(lib.qr_tsqr_comm_ranks(self.h, C.byref(n)), "qr_tsqr_comm_ranks")
        return n.value


def tsqr_factor_virtual(plans, shards, lda, Rs):
    """qr_tsqr_factor_virtual_dev: the panel-pipelined schedule over P TsqrPlans of one device (virtual ranks)."""
    P = len(plans)
    hs = (_vp * P)(*[p.h for p in plans])
    As = (_vp * P)(*[_dptr(a) for a in shards])
    Rp = (_vp * P)(*[_dptr(r) for r in Rs])
    check(lib.qr_tsqr_factor_virtual_dev(hs, P, As, lda, Rp), "qr_tsqr_factor_virtual_dev")


def tsqr_unique_id():
    buf = C.create_string_buffer(128)
    check(lib.qr_tsqr_unique_id(buf), "qr_tsqr_unique_id")
    return bytes(buf.raw)


def uniform_at(seed, idx):
    return lib.qr_uniform_at(seed, idx)


def uniform_matrix_host(rows, cols, row_off=0, total_rows=None, seed=12):
    """Host evaluation of the device generator (same hash): numpy (rows x cols)."""
    total = rows if total_rows is None else total_rows
    c = np.arange(cols, dtype=np.uint64)[None, :]
    i = (np.arange(rows, dtype=np.uint64) + np.uint64(row_off))[:, None]
    idx = c * np.uint64(total) + i
    with np.errstate(over="ignore"):
        z = np.uint64(seed) * np.uint64(0x9E3779B97F4A7C15) + (idx + np.uint64(1)) * np.uint64(0xD1B54A32D192ED03)
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        z = z ^ (z >> np.uint64(31))
    return (z >> np.uint64(11)).astype(np.float64) * (1.0 / 9007199254740992.0)


def flops(m, n):
    """Householder QR factorisation flops, 2mn^2 - 2n^3/3."""
    return 2.0 * m * n * n - 2.0 * n ** 3 / 3.0
