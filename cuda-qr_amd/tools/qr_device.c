/* qr_device.c -- command-line timing harness in the shape of the reference's GPU binary
 * (reference qr.cu:709-806: `./qr_device m n`, srand(12) input, `trials` = 3 timed calls of the host-pointer
 * mmqr, average seconds printed as " MMQR ran QR on MxN matrix in T s (avg over 3)").
 *
 * Like the reference's timed region (qr.cu:776-789, which wraps cudaMalloc + H2D + kernels + D2H), the first
 * figure times the whole host-pointer call.  The second figure is the same factorisation with the matrix already
 * resident in HBM (what bench.py reports).  The reference silently rounds m and n to fit its window ladder
 * (qr.cu:722-734); this library takes any m >= n, so the sizes are used as given.
 *
 * `./qr_device m n --compare` adds the vendor line the reference prints under ENABLE_MAGMA (qr.cu:790-806, "MAGMA ran QR on ..."):
 * rocSOLVER's dgeqrf on the same matrix, resident in HBM.  rocSOLVER / rocBLAS are dlopen()ed by THIS tool only when the flag is
 * given: the library never links or loads them.
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>
#include <dlfcn.h>

#include "mi355x_qr.h"

#define TRIALS 3

static double now(void)
{
    struct timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return ts.tv_sec + 1e-9 * ts.tv_nsec;
}

/* the comparator (qr.cu:555-565 magmaQR, timed at qr.cu:790-806): rocsolver_dgeqrf(handle, m, n, dA, lda, dtau), column-major in place */
static int vendor_line(const double* A, double* dA, double* dtau, int m, int n, double flops)
{
    void* hb = dlopen("librocblas.so.5", RTLD_NOW | RTLD_GLOBAL);
    if (!hb) hb = dlopen("librocblas.so", RTLD_NOW | RTLD_GLOBAL);
    void* hs = dlopen("librocsolver.so.0", RTLD_NOW | RTLD_GLOBAL);
    if (!hs) hs = dlopen("librocsolver.so", RTLD_NOW | RTLD_GLOBAL);
    void* hh = dlopen("libamdhip64.so", RTLD_NOW | RTLD_GLOBAL);
    if (!hb || !hs || !hh) { printf("rocSOLVER not available on this machine (%s)\n", dlerror()); return 0; }
    int (*create)(void**) = (int (*)(void**)) dlsym(hb, "rocblas_create_handle");
    int (*destroy)(void*) = (int (*)(void*)) dlsym(hb, "rocblas_destroy_handle");
    int (*geqrf)(void*, int, int, double*, int, double*) = (int (*)(void*, int, int, double*, int, double*)) dlsym(hs, "rocsolver_dgeqrf");
    int (*devsync)(void) = (int (*)(void)) dlsym(hh, "hipDeviceSynchronize");
    void* handle = NULL;
    if (!create || !destroy || !geqrf || !devsync || create(&handle)) { printf("rocSOLVER not usable on this machine\n"); return 0; }
    double el = 0.0;
    for (int t = -1; t < TRIALS; t++) {             /* t = -1: untimed first call (workspace, kernel loading) */
        if (qr_copy_to_device(dA, A, sizeof(double) * (size_t) m * n)) return 1;
        devsync();
        const double t0 = now();
        if (geqrf(handle, m, n, dA, m, dtau) || devsync()) { fprintf(stderr, "rocsolver_dgeqrf failed\n"); destroy(handle); return 1; }
        if (t >= 0) el += now() - t0;
    }
    destroy(handle);
    printf("rocSOLVER ran QR on %dx%d matrix in %f s (avg over %d)   [matrix resident in HBM, %.1f GFLOP/s fp64]\n",
           m, n, el / TRIALS, TRIALS, flops / (el / TRIALS) / 1e9);
    return 0;
}

int main(int argc, char** argv)
{
    if (argc < 3) { puts("Usage: ./qr_device m n [--compare]"); return 1; }
    const int compare = argc > 3 && strcmp(argv[3], "--compare") == 0;
    const int m = atoi(argv[1]), n = atoi(argv[2]);
    if (m < 1 || n < 1 || m < n) { fprintf(stderr, "need m >= n >= 1\n"); return 1; }
    printf("Exact problem size: %dx%d\n", m, n);
    char arch[64]; int cus = 0, khz = 0; size_t hbm = 0;
    if (qr_device_info(arch, sizeof arch, &cus, &khz, &hbm)) { fprintf(stderr, "no HIP device\n"); return 1; }
    printf("Testing mmqr on \"%s\" (%d CUs, %.1f GiB HBM)\n", arch, cus, hbm / 1073741824.0);

    const size_t cnt = (size_t) m * n;
    double* A = malloc(sizeof(double) * cnt);
    double* RV = malloc(sizeof(double) * cnt);
    srand(12);
    for (size_t i = 0; i < cnt; i++) RV[i] = A[i] = (double) rand() / RAND_MAX;      /* qr.cu:765-771 */

    double* tau = NULL;
    mmqr(RV, &tau, m, n);                       /* untimed: first call pays library/context initialisation */
    free(tau);
    double el = 0.0;
    for (int t = 0; t < TRIALS; t++) {
        memcpy(RV, A, sizeof(double) * cnt);    /* refresh, untimed like qr.cu:786-787 */
        const double t0 = now();
        if (mmqr_status(RV, &tau, m, n)) { fprintf(stderr, "mmqr failed\n"); return 1; }
        el += now() - t0;
        free(tau);
    }
    const double flops = 2.0 * m * (double) n * n - 2.0 * (double) n * n * n / 3.0;
    printf(" MMQR ran QR on %dx%d matrix in %f s (avg over %d)   [host pointers: alloc + H2D + QR + D2H, %.1f GFLOP/s fp64]\n",
           m, n, el / TRIALS, TRIALS, flops / (el / TRIALS) / 1e9);

    /* the same factorisation with the matrix already resident in HBM (plan API; what bench.py reports) */
    qr_plan* p = NULL;
    double *dA = NULL, *dtau = NULL;
    if (qr_plan_create(&p, m, n, 0, 0) || qr_device_malloc((void**) &dA, sizeof(double) * cnt) ||
        qr_device_malloc((void**) &dtau, sizeof(double) * n)) { fprintf(stderr, "device setup failed\n"); return 1; }
    el = 0.0;
    for (int t = -1; t < TRIALS; t++) {
        if (qr_copy_to_device(dA, A, sizeof(double) * cnt)) { fprintf(stderr, "copy failed\n"); return 1; }
        const double t0 = now();
        if (qr_geqrf_dev(p, dA, m, n, m, dtau) || qr_plan_sync(p)) { fprintf(stderr, "qr_geqrf_dev failed\n"); return 1; }
        if (t >= 0) el += now() - t0;
    }
    printf(" MMQR ran QR on %dx%d matrix in %f s (avg over %d)   [matrix resident in HBM, %.1f GFLOP/s fp64]\n",
           m, n, el / TRIALS, TRIALS, flops / (el / TRIALS) / 1e9);
    if (compare && vendor_line(A, dA, dtau, m, n, flops)) return 1;
    qr_device_free(dA); qr_device_free(dtau);
    qr_plan_destroy(p);
    free(A); free(RV);
    return 0;
}
