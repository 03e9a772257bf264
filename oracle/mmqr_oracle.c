/* mmqr_oracle.c -- TEST INFRASTRUCTURE ONLY; never linked into or called by the product library.
 *
 * CPU restatement ("port") of the reference brian-kelley/CUDA-QR qr.c host path:
 *   getPanelDims qr.c:47-53, mmqr qr.c:55-313, identity qr.c:316-324,
 *   explicitQR qr.c:330-438, dgemm qr.c:443-459, input generator qr.c:468-474,
 *   residual qr.c:505-515.
 * Differences from the reference, none of them arithmetic: the window shape PR x PC is a run-time
 * argument (the reference hard-codes it, qr.c:12-13), tau is caller-allocated, nothing is printed
 * and nothing is malloc'ed per window.
 *
 * PARITY PINNING: tests/test_oracle.py checks oracle_mmqr_{d,f} bitwise (factored matrix and
 * tau) against (a) golden fixtures in tests/golden/ that were produced by the REAL reference
 * compiled from /root/reference/qr.c (oracle/make_golden.py), and (b) the real reference itself
 * (oracle/_ref/libqrref_*.so) whenever those prebuilt libraries are present.
 *
 * Users: tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg only.
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>

#define REAL double
#define SUF(x) x##_d
#include "mmqr_oracle_body.inc"
#undef REAL
#undef SUF

#define REAL float
#define SUF(x) x##_f
#include "mmqr_oracle_body.inc"
#undef REAL
#undef SUF
