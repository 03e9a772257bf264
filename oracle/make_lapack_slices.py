#!/usr/bin/env python3
"""Generates tests/golden/lapack_16384_slices.npz: slices of the sign-normalised R factor of BASELINE config C3 (16384 x 16384,
the device generator's uniform[0,1) matrix, seed 12) computed ONCE on the CPU by LAPACK dgeqrf (scipy), so that the full-size GPU
test has an R to compare with instead of properties only.  ~2 minutes and ~6 GB on 8 cores; the output is ~0.7 MB.

    python3 oracle/make_lapack_slices.py [n]        (n = 16384)

TEST INFRASTRUCTURE: nothing under cuda-qr_amd/ imports this; the generator it calls is the host evaluation of the counter hash
(cuda_qr_amd.uniform_matrix_host), which tests/test_gpu_kernels.py pins to the device generator bit for bit."""
import os
import sys
import time

import numpy as np
import scipy.linalg

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
    import cuda_qr_amd as qr
    t0 = time.time()
    A = np.asfortranarray(qr.uniform_matrix_host(n, n, seed=12))
    fro = float(np.linalg.norm(A))
    print(f"generated {n}x{n} in {time.time() - t0:.1f} s, |A|_F = {fro!r}", flush=True)
    t0 = time.time()
    (R,) = scipy.linalg.qr(A, mode="r", overwrite_a=True, check_finite=False)
    print(f"dgeqrf in {time.time() - t0:.1f} s", flush=True)
    s = np.where(np.diag(R) < 0, -1.0, 1.0)
    R = np.triu(R) * s[:, None]                      # diag >= 0: the implementation-independent R
    h = n // 2
    out = {
        "n": np.int64(n), "seed": np.int64(12), "fro_A": np.float64(fro),
        "diag": np.diag(R).copy(),
        "rownorm": np.sqrt((R * R).sum(axis=1)),
        "top_right": R[:32, n - 256:].copy(),                       # rows 0..31, last 256 columns
        "middle": R[h:h + 48, h:h + 512].copy(),                    # rows h..h+47, columns h..h+511
        "bottom_right": R[n - 128:, n - 128:].copy(),               # the last 128 x 128 triangle
        "col_last": R[:, n - 1].copy(),                             # the last column: depends on every reflector
    }
    path = os.path.join(ROOT, "tests", "golden", f"lapack_{n}_slices.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()
