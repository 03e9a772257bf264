/* selfcheck.c -- a C caller of the drop-in library, shaped like the reference's own self-test
 * (reference qr.c:461-523: random m x n matrix from srand(12), mmqr, explicitQR, dgemm(Q,R), unnormalised
 * Frobenius residual printed with "%.9g").  Written against include/mi355x_qr.h only; it links the same six
 * symbols a reference caller links (qr.c:15-18,47,55) and nothing else.
 * usage: selfcheck [m n]      (default 6 4 = PR + (PR-PC), 2*PC of the committed reference)
 */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>

#include "mi355x_qr.h"

int main(int argc, char** argv)
{
    int m = 6, n = 4;
    if (argc >= 3) { m = atoi(argv[1]); n = atoi(argv[2]); }
    if (m < n || n < 1) { fprintf(stderr, "need m >= n >= 1\n"); return 2; }
    double* A = malloc(sizeof(double) * (size_t) m * n);
    double* RV = malloc(sizeof(double) * (size_t) m * n);
    srand(12);
    for (size_t i = 0; i < (size_t) m * n; i++) RV[i] = A[i] = (double) rand() / RAND_MAX;

    double* tau = NULL;
    mmqr(RV, &tau, m, n);
    if (!tau) { fprintf(stderr, "mmqr failed\n"); return 1; }
    int rowPanels, colPanels;
    getPanelDims(m, n, &rowPanels, &colPanels);
    printf("panels: %d x %d\n", rowPanels, colPanels);

    double* Q = malloc(sizeof(double) * (size_t) m * m);
    double* R = malloc(sizeof(double) * (size_t) m * n);
    double* QR = malloc(sizeof(double) * (size_t) m * n);
    explicitQR(RV, tau, Q, R, m, n);
    dgemm(Q, R, QR, m, m, n);
    double err = 0, nrm = 0;
    for (size_t i = 0; i < (size_t) m * n; i++) { err += (QR[i] - A[i]) * (QR[i] - A[i]); nrm += A[i] * A[i]; }
    if (m <= 8) { printMat(R, m, n); }
    printf("L2 norm of residual QR-A: %.9g\n", sqrt(err));
    printf("relative residual: %.9g\n", sqrt(err / nrm));
    free(A); free(RV); free(Q); free(R); free(QR); free(tau);
    return 0;
}
