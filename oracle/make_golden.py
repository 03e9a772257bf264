#!/usr/bin/env python3
"""oracle/make_golden.py -- regenerate tests/golden/*.npz from the REAL reference.

Run in the dev container only (needs oracle/_ref/, i.e. /root/reference/qr.c compiled by
oracle/Makefile).  Every array written here is an OUTPUT of the reference's own mmqr/explicitQR
(qr.c:55, qr.c:330) on the reference's own input generator (srand(12); rand()/RAND_MAX,
qr.c:468-474); no reference source text is stored.  The fixtures are what pins
oracle/mmqr_oracle.c (bitwise) and the HIP path (sign-normalised R, residual, orthogonality)
on machines where /root/reference does not exist.
"""
import hashlib
import json
import os
import sys

import numpy as np

sys.path[0] = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
from oracle import oracle as O  # noqa: E402

OUT = os.path.join(os.path.dirname(O.HERE), "tests", "golden")
ONLY_MISSING = "--only-missing" in sys.argv      # keep existing fixtures, generate the new ones (the 2024^2 case takes ~3 min)


def sha(a):
    return hashlib.sha256(np.asfortranarray(a).tobytes(order="F")).hexdigest()


def metrics(A, Q, R):
    A64, Q64, R64 = (x.astype(np.float64) for x in (A, Q, R))
    return (np.linalg.norm(A64 - Q64 @ R64) / np.linalg.norm(A64),
            np.linalg.norm(Q64.T @ Q64 - np.eye(Q64.shape[0])))


def case(m, n, PR, PC, dtype, store_full, with_q, partial=False):
    tag = "f64" if dtype == np.float64 else "f32"
    name = f"ref_{m}x{n}_{tag}_{PR}x{PC}"
    if ONLY_MISSING and os.path.exists(os.path.join(OUT, name + ".npz")):
        print(name, "kept")
        return name
    A = O.fill_rand(m, n, 12, dtype)
    F, tau = O.ref_mmqr(A, PR, PC)
    d = dict(m=m, n=n, PR=PR, PC=PC, seed=12,
             A00=A[0, 0], A10=A[1, 0], normA=np.linalg.norm(A.astype(np.float64)),
             sha_A=sha(A), sha_F=sha(F), sha_tau=sha(tau),
             Rn=O.sign_normalise(F.astype(np.float64)),
             diagR=np.diag(F[:n, :]).copy())
    if store_full:
        d.update(A=A, F=F, tau=tau)
    if with_q:
        if m <= 128:
            Q, R = O.ref_explicit_qr(F, tau, PR, PC)         # the reference's own O(m^3) builder
            d["q_by"] = "reference explicitQR"
        else:
            Q, R = O.explicit_qr(F, tau, PR, PC, faithful=False)   # too slow in the reference (314 s)
            d["q_by"] = "oracle fast builder on reference factors"
        resid, orth = metrics(A, Q, R)
        d.update(resid=resid, orth=orth)
        if store_full:
            d.update(Q=Q, R=R)
    # store only the upper triangle for the big ones; `partial`: only slices of it (leading rows, trailing
    # columns, the diagonal and the row norms) plus a digest of the whole triangle, to keep the fixture small
    if partial:
        Rn = d.pop("Rn")
        d.update(Rn_rows_head=Rn[:32, :].copy(), Rn_cols_tail=Rn[:, -64:].copy(), Rn_diag=np.diag(Rn).copy(),
                 Rn_rownorm=np.linalg.norm(Rn, axis=1), Rn_fro=np.linalg.norm(Rn), sha_Rn=sha(Rn))
    elif not store_full:
        iu = np.triu_indices(n)
        d["Rn_triu"] = d.pop("Rn")[iu]
    np.savez_compressed(os.path.join(OUT, name + ".npz"), **d)
    print(name, {k: (v if np.ndim(v) == 0 else np.shape(v)) for k, v in d.items()
                 if k in ("resid", "orth", "normA", "Rn", "Rn_triu", "q_by")})
    return name


def main():
    O.build()
    os.makedirs(OUT, exist_ok=True)
    names = [
        case(6, 4, 4, 2, np.float32, True, True),      # the reference's own self-test shape, as committed
        case(6, 4, 4, 2, np.float64, True, True),
        case(128, 32, 4, 2, np.float64, True, True),
        case(120, 32, 64, 8, np.float64, True, True),
        case(64, 20, 16, 4, np.float64, True, True),
        case(512, 128, 64, 8, np.float64, False, True),  # C1 (BASELINE.json configs[0])
        case(512, 128, 4, 2, np.float64, False, False),  # C1 with the window as committed
        # multi-panel cases for the blocked GPU path (5 / 10 outer panels at nb = 128 / 64; the sample bench.py times
        # as cpu_baseline) and a square one (16 outer panels at nb = 128): these reach the trailing update, the
        # look-ahead schedule and the CholeskyQR2 leaf, which the single-panel cases above cannot
        case(1184, 640, 64, 8, np.float64, False, False),
        case(2024, 2024, 64, 8, np.float64, False, False, partial=True),
    ]
    with open(os.path.join(OUT, "MANIFEST.json"), "w") as f:
        json.dump({"generated_by": "oracle/make_golden.py", "source": "reference qr.c via oracle/_ref",
                   "files": [n + ".npz" for n in names]}, f, indent=1)


if __name__ == "__main__":
    main()
