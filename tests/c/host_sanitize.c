/* host_sanitize.c -- TEST-ONLY driver of the C host layer (cuda-qr_amd/csrc/qr_host.c) over the stub device layer
 * (qrd_stub.c), built with -fsanitize=address,undefined or -fsanitize=thread by `make -C cuda-qr_amd asan|tsan`.
 * Exercises what the sanitizers can see without a GPU: the schedule's workspace indexing (the stub bounds-checks every operand
 * block), plan life cycle and leak-freedom, the plan cache of the host-pointer entry points under concurrent callers, the
 * per-device threads + barrier protocol of qr_thin_mgpu, and the event protocol of the TSQR plan. */
#define _POSIX_C_SOURCE 200809L
#include <pthread.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "../../include/mi355x_qr.h"

long qrd_stub_launches(void);
long qrd_stub_stalls(void);
int qrd_stub_live_allocations(void);

#define OK(x) do { int rc_ = (x); if (rc_) { fprintf(stderr, "%s:%d: %s -> %d (%s)\n", __FILE__, __LINE__, #x, rc_, qr_strerror(rc_)); exit(1); } } while (0)

static void factor_once(int pm, int pn, int nb, int m, int n, int with_q, int profile)
{
    qr_plan* p = NULL;
    double *dA = NULL, *dtau = NULL, *dQ = NULL, *dR = NULL;
    OK(qr_plan_create(&p, pm, pn, nb, 0));
    OK(qr_device_malloc((void**) &dA, sizeof(double) * (size_t) m * n));
    OK(qr_device_malloc((void**) &dtau, sizeof(double) * n));
    if (profile) OK(qr_plan_set_profile(p, 1));
    OK(qr_geqrf_dev(p, dA, m, n, m, dtau));
    OK(qr_geqrf_dev(p, dA, m, n, m, dtau));
    if (profile) { qr_profile pr; OK(qr_plan_get_profile(p, &pr)); }
    if (with_q) {
        OK(qr_device_malloc((void**) &dQ, sizeof(double) * (size_t) m * n));
        OK(qr_device_malloc((void**) &dR, sizeof(double) * (size_t) n * n));
        OK(qr_extract_r_dev(p, dA, m, n, m, dR, n, n));
        OK(qr_applyq_dev(p, dA, m, n, m, dtau, dQ, n, m, 1));
        OK(qr_applyq_dev(p, dA, m, n, m, dtau, dQ, n, m, 0));
    }
    OK(qr_plan_sync(p));
    OK(qr_device_free(dA)); OK(qr_device_free(dtau)); OK(qr_device_free(dQ)); OK(qr_device_free(dR));
    OK(qr_plan_destroy(p));
}

/* one rank of a multi-rank TSQR over the stub's thread communicators: four steps, the joint fall-back decision before the third */
int qrd_comm_init_all(void** comms, int n, const int* devs);
int qrd_comm_destroy(void* comm);
typedef struct { void* comm; int rank, P, expect_fallback, bad; } tsqr_rank_arg;
static void* tsqr_rank(void* arg)
{
    tsqr_rank_arg* a = (tsqr_rank_arg*) arg;
    const int ml = 8192, n = 256;
    qr_tsqr_plan* t = NULL;
    double *dA = NULL, *dR = NULL, st[5];
    OK(qr_tsqr_plan_create_comm(&t, a->comm, a->P, a->rank, ml, n, 64));
    OK(qr_device_malloc((void**) &dA, sizeof(double) * (size_t) ml * n));
    OK(qr_device_malloc((void**) &dR, sizeof(double) * (size_t) n * n));
    if (!qr_tsqr_is_pipelined(t)) a->bad = 1;
    for (int it = 0; it < 4; ++it) OK(qr_tsqr_factor_dev(t, dA, ml, dR));
    OK(qr_tsqr_gather_stats(t, st));
    if ((int) st[4] != a->expect_fallback || qr_tsqr_is_pipelined(t) == a->expect_fallback) a->bad = 2;
    if (!a->expect_fallback && !(st[0] > 0.0 && st[1] > 0.0 && st[2] > st[0])) a->bad = 3;
    OK(qr_tsqr_plan_destroy(t));
    OK(qr_device_free(dA)); OK(qr_device_free(dR));
    return NULL;
}
static int tsqr_ranks_run(int expect_fallback)
{
    enum { P = 3 };
    void* comms[P];
    const int devs[P] = {0, 1, 2};
    pthread_t th[P];
    tsqr_rank_arg args[P];
    OK(qrd_comm_init_all(comms, P, devs));
    for (int r = 0; r < P; ++r) {
        args[r] = (tsqr_rank_arg){comms[r], r, P, expect_fallback, 0};
        pthread_create(&th[r], NULL, tsqr_rank, &args[r]);
    }
    int bad = 0;
    for (int r = 0; r < P; ++r) { pthread_join(th[r], NULL); if (args[r].bad) bad = args[r].bad; }
    for (int r = P - 1; r >= 0; --r) qrd_comm_destroy(comms[r]);
    return bad;
}

static void* host_caller(void* arg)
{
    const int id = (int) (long) arg;
    const int shapes[3][2] = {{512, 128}, {300, 77}, {1100, 1030}};
    for (int it = 0; it < 6; ++it) {
        const int m = shapes[(id + it) % 3][0], n = shapes[(id + it) % 3][1];
        double* A = (double*) calloc((size_t) m * n, sizeof(double));
        double* Q = (double*) calloc((size_t) m * m, sizeof(double));
        double* R = (double*) calloc((size_t) m * n, sizeof(double));
        double* tau = NULL;
        OK(mmqr_status(A, &tau, m, n));
        int nb = 0, rp = 0, cp = 0;
        OK(qr_default_block_size(m, n, &nb, NULL));
        getPanelDims(m, n, &rp, &cp);
        tau[(size_t) rp * cp * nb - 1] = 0.0;                /* the last entry of the documented length is ours to touch */
        OK(explicitQR_status(A, tau, Q, R, m, n));
        free(tau); free(A); free(Q); free(R);
    }
    return NULL;
}

int main(void)
{
    /* 1. schedules: single stream, look-ahead on shared CUs, look-ahead on a CU partition with the panel stream's share,
     *    two-level panels, ragged widths, sub-size problems on a bigger plan (incl. the tall-skinny shortcut mix), graph replay */
    setenv("MI355XQR_LOOKAHEAD", "0", 1);
    factor_once(1000, 333, 128, 1000, 333, 1, 0);
    factor_once(4096, 1024, 512, 4096, 1000, 1, 1);
    factor_once(70000, 96, 32, 70000, 96, 1, 0);
    factor_once(16384, 2048, 256, 4096, 2048, 0, 0);
    factor_once(65536, 256, 128, 65536, 256, 1, 0);       /* tall panels at full width (qr_panel_cqr): accepted at 65536 rows, refused by the stub's guard at 65408 */
    setenv("MI355XQR_LOOKAHEAD", "1", 1);
    factor_once(3000, 2100, 128, 3000, 2100, 1, 1);
    factor_once(2304, 2304, 512, 2304, 2304, 0, 0);
    factor_once(16384, 2048, 256, 8192, 2048, 0, 0);
    setenv("MI355XQR_SPLIT", "64", 1);
    setenv("MI355XQR_BALANCE", "14,44,0.05,0.05", 1);
    factor_once(6144, 4096, 256, 6144, 4096, 0, 1);
    factor_once(6144, 4096, 128, 6000, 3900, 0, 0);
    factor_once(5120, 5120, 512, 5120, 5120, 0, 0);
    setenv("MI355XQR_NEXT", "update", 1);
    factor_once(6144, 4096, 256, 6144, 4096, 0, 0);
    unsetenv("MI355XQR_NEXT"); unsetenv("MI355XQR_SPLIT"); unsetenv("MI355XQR_BALANCE");
    setenv("MI355XQR_LOOKAHEAD", "0", 1);
    setenv("MI355XQR_GRAPH", "1", 1);
    factor_once(2000, 500, 128, 2000, 500, 0, 0);
    unsetenv("MI355XQR_GRAPH"); unsetenv("MI355XQR_LOOKAHEAD");
    setenv("MI355XQR_PANEL", "tsqr", 1);
    factor_once(9000, 200, 128, 9000, 200, 1, 0);
    unsetenv("MI355XQR_PANEL");
    /* 1b. guard modes of the full-width panel.  The stub's guard refuses heights with bit 12 set: of a 65536 x 256 problem, panel 0
     *     (65536 rows) is accepted and panel 1 (65408) refused.  Poll mode hands the refused panel to the leaf chain (status 0); latch
     *     mode reports QR_E_REFUSED at the sync, also when the mode is switched in between (the pending refusal must not be lost), and
     *     a plan that has reported is clean again */
    {
        qr_plan* p = NULL;
        double *dA = NULL, *dtau = NULL;
        long long st[4];
        const int m = 65536, n = 256;
        OK(qr_plan_create(&p, m, n, 128, 0));
        OK(qr_device_malloc((void**) &dA, sizeof(double) * (size_t) m * n));
        OK(qr_device_malloc((void**) &dtau, sizeof(double) * n));
        OK(qr_geqrf_dev(p, dA, m, n, m, dtau));                               /* poll (default) */
        OK(qr_plan_sync(p));
        OK(qr_plan_route_stats(p, st));
        if (st[0] != 2 || st[1] != 1) { fprintf(stderr, "poll mode: %lld tall panels, %lld refused\n", st[0], st[1]); return 20; }
        OK(qr_plan_set_guard_mode(p, 1));
        OK(qr_geqrf_dev(p, dA, m, n, m, dtau));
        if (qr_plan_sync(p) != QR_E_REFUSED) return 21;
        OK(qr_plan_sync(p));                                                  /* reported once, then clean */
        OK(qr_geqrf_dev(p, dA, m, n, m, dtau));
        if (qr_plan_set_guard_mode(p, 0) != QR_E_REFUSED) return 22;          /* the switch drains and reports what was latched */
        OK(qr_plan_sync(p));
        OK(qr_geqrf_dev(p, dA, m, n, m, dtau));                               /* poll again */
        OK(qr_plan_sync(p));
        OK(qr_plan_route_stats(p, st));
        if (st[0] != 8 || st[1] != 4) { fprintf(stderr, "after the mode switches: %lld tall panels, %lld refused\n", st[0], st[1]); return 23; }
        OK(qr_device_free(dA)); OK(qr_device_free(dtau));
        OK(qr_plan_destroy(p));
    }
    /* 1b'. the preconditioned retry of a refused panel (poll mode): the stub's guard accepts a retried panel unless bit 7 of its height is
     *      set -- 69632 rows: panel 0 refused, retried, accepted; panel 1 (69504) accepted at once.  (65408 above: refused twice -> leaf chain) */
    {
        qr_plan* p = NULL;
        double *dA = NULL, *dtau = NULL;
        long long st[4], rt[2];
        const int m = 69632, n = 256;
        OK(qr_plan_create(&p, m, n, 128, 0));
        OK(qr_device_malloc((void**) &dA, sizeof(double) * (size_t) m * n));
        OK(qr_device_malloc((void**) &dtau, sizeof(double) * n));
        OK(qr_geqrf_dev(p, dA, m, n, m, dtau));
        OK(qr_plan_sync(p));
        OK(qr_plan_route_stats(p, st));
        OK(qr_plan_retry_stats(p, rt));
        if (st[0] != 2 || st[1] != 1 || rt[0] != 1 || rt[1] != 1) {
            fprintf(stderr, "retry: %lld tall panels, %lld refused, %lld retried, %lld accepted then\n", st[0], st[1], rt[0], rt[1]);
            return 25;
        }
        OK(qr_device_free(dA)); OK(qr_device_free(dtau));
        OK(qr_plan_destroy(p));
    }
    /* 1c. a stalled one-launch panel under the host-pointer entry points: mmqr_status and qr_thin must see QR_E_STALL at their plan
     *     sync and factor again with the route off (status 0 for the caller), never return rc = 0 over the stalled result */
    {
        const int m = 4096, n = 256;
        double* A = (double*) calloc((size_t) m * n, sizeof(double));
        double* Q = (double*) calloc((size_t) m * n, sizeof(double));
        double* R = (double*) calloc((size_t) n * n, sizeof(double));
        double* tau = NULL;
        const long s0 = qrd_stub_stalls();
        setenv("QRD_STUB_STALL_ONCE", "1", 1);
        OK(mmqr_status(A, &tau, m, n));
        free(tau);
        setenv("QRD_STUB_STALL_ONCE", "1", 1);
        OK(qr_thin(A, m, n, Q, R, 0, 1));
        setenv("QRD_STUB_STALL_ONCE", "1", 1);
        OK(qr_thin(A, 2 * m, n / 2, Q, R, 0, 2));
        if (qrd_stub_stalls() != s0 + 3) { fprintf(stderr, "stall injection: %ld of 3 fired\n", qrd_stub_stalls() - s0); return 24; }
        OK(qr_release_cached_plans());
        free(A); free(Q); free(R);
    }
    /* argument errors */
    { qr_plan* p = NULL; if (qr_plan_create(&p, 10, 20, 0, 0) != QR_E_ARG || qr_plan_create(&p, 64, 64, 100, 32) != QR_E_ARG) return 2; }

    /* 2. host-pointer entry points from four threads at once: the plan cache (QR_CACHE_SLOTS = 4, three shapes) */
    pthread_t th[4];
    for (long i = 0; i < 4; ++i) pthread_create(&th[i], NULL, host_caller, (void*) i);
    for (int i = 0; i < 4; ++i) pthread_join(th[i], NULL);
    OK(qr_release_cached_plans());

    /* 3. thin QR: virtual shards on one device, and one host thread per (stub) device with the all-gather in between */
    {
        const int m = 4096, n = 64;
        double* A = (double*) calloc((size_t) m * n, sizeof(double));
        double* Q = (double*) calloc((size_t) m * n, sizeof(double));
        double* R = (double*) calloc((size_t) n * n, sizeof(double));
        OK(qr_thin(A, m, n, Q, R, 0, 1));
        OK(qr_thin(A, m, n, Q, R, 32, 3));
        OK(qr_thin_mgpu(A, m, n, Q, R, 0, 1));
        OK(qr_thin_mgpu(A, m, n, Q, R, 0, 4));
        OK(qr_thin_mgpu(A, 4093, n, Q, R, 0, 3));                               /* ragged last shard */
        if (qr_thin_mgpu(A, m, n, Q, R, 0, 5) != QR_E_ARG) return 3;          /* more than the 4 stub devices */
        free(A); free(Q); free(R);
        /* n a multiple of the block size: the per-device threads run the panel-pipelined exchange (4 all-gathers each) */
        const int m2 = 16384, n2 = 256;
        A = (double*) calloc((size_t) m2 * n2, sizeof(double));
        Q = (double*) calloc((size_t) m2 * n2, sizeof(double));
        R = (double*) calloc((size_t) n2 * n2, sizeof(double));
        OK(qr_thin_mgpu(A, m2, n2, Q, R, 64, 4));
        free(A); free(Q); free(R);
    }

    /* 4. TSQR plan: external transport, back-to-back steps (event protocol between the local and the stacked plan), thin Q */
    {
        const int ml = 8192, n = 128, P = 4;
        qr_tsqr_plan* t = NULL;
        double *dA = NULL, *dR = NULL, *dQ = NULL, *send = NULL, *recv = NULL;
        OK(qr_tsqr_plan_create_comm(&t, NULL, P, 1, ml, n, 0));
        OK(qr_device_malloc((void**) &dA, sizeof(double) * (size_t) ml * n));
        OK(qr_device_malloc((void**) &dQ, sizeof(double) * (size_t) ml * n));
        OK(qr_device_malloc((void**) &dR, sizeof(double) * (size_t) n * n));
        OK(qr_tsqr_exchange_buffers(t, &send, &recv));
        if (!send || !recv) return 4;
        if (qr_tsqr_factor_dev(t, dA, ml, dR) != QR_E_ARG) return 5;          /* no communicator: the all-in-one call is refused */
        for (int it = 0; it < 3; ++it) {
            OK(qr_tsqr_local_dev(t, dA, ml));
            OK(qr_tsqr_stacked_dev(t, dR));
        }
        OK(qr_tsqr_formq_dev(t, dA, ml, dQ, ml));
        int nr = 0;
        OK(qr_tsqr_comm_ranks(t, &nr));
        OK(qr_tsqr_sync(t));
        OK(qr_tsqr_plan_destroy(t));
        OK(qr_tsqr_plan_create(&t, NULL, 1, 0, ml, n, 0));
        OK(qr_tsqr_factor_dev(t, dA, ml, dR));
        OK(qr_tsqr_formq_dev(t, dA, ml, dQ, ml));
        OK(qr_tsqr_plan_destroy(t));
        OK(qr_device_free(dA)); OK(qr_device_free(dQ)); OK(qr_device_free(dR));
    }
    /* 4b. panel-pipelined TSQR: virtual ranks (P plans, one thread) and the self-gather form, twice (send blocks reused) */
    {
        enum { P = 3 };
        const int ml = 16384, n = 384;
        qr_tsqr_plan* tps[P];
        double *dA[P], *dR[P];
        for (int r = 0; r < P; ++r) {
            OK(qr_tsqr_plan_create_comm(&tps[r], NULL, P, r, ml, n, 128));
            if (!qr_tsqr_is_pipelined(tps[r])) return 9;
            OK(qr_device_malloc((void**) &dA[r], sizeof(double) * (size_t) ml * n));
            OK(qr_device_malloc((void**) &dR[r], sizeof(double) * (size_t) n * n));
        }
        OK(qr_tsqr_factor_virtual_dev(tps, P, dA, ml, dR));
        OK(qr_tsqr_factor_virtual_dev(tps, P, dA, ml, dR));
        OK(qr_tsqr_factor_selfgather_dev(tps[1], dA[1], ml, dR[1]));
        OK(qr_tsqr_factor_selfgather_dev(tps[1], dA[1], ml, dR[1]));
        OK(qr_tsqr_formq_dev(tps[1], dA[1], ml, dA[0], ml));
        for (int r = 0; r < P; ++r) { OK(qr_tsqr_plan_destroy(tps[r])); OK(qr_device_free(dA[r])); OK(qr_device_free(dR[r])); }
    }

    /* 4c. three ranks (threads) over the stub communicators: the gathers are timed, and with MI355XQR_TSQR_PIPE unset the ranks
     * decide together before their third step -- keep the pipelined exchange when the gathers are short, one collective when slow */
    {
        unsetenv("MI355XQR_TSQR_PIPE");
        if (tsqr_ranks_run(0)) return 10;
        setenv("QRD_STUB_GATHER_MS", "50", 1);
        if (tsqr_ranks_run(1)) return 11;
        unsetenv("QRD_STUB_GATHER_MS");
        setenv("MI355XQR_TSQR_PIPE", "1", 1);                              /* forced on: no decision, slow gathers or not */
        setenv("QRD_STUB_GATHER_MS", "50", 1);
        if (tsqr_ranks_run(0)) return 12;
        unsetenv("QRD_STUB_GATHER_MS");
        unsetenv("MI355XQR_TSQR_PIPE");
    }

    /* 5. legacy-layout shim: argument checks and buffer sizes of the sliding-window path */
    {
        const int m = 512, n = 128;
        double* A = (double*) calloc((size_t) m * n, sizeof(double));
        double* Q = (double*) calloc((size_t) m * m, sizeof(double));
        double* R = (double*) calloc((size_t) m * n, sizeof(double));
        double* tau = NULL;
        OK(mmqr_legacy_status(A, &tau, m, n, 64, 8));
        int rp = 0, cp = 0;
        getPanelDims_legacy(m, n, 64, 8, &rp, &cp);
        if (rp != 9 || cp != 16) return 7;                                  /* C1 @PR64/PC8 is a 9 x 16 grid */
        tau[(size_t) rp * cp * 8 - 1] = 0.0;
        OK(explicitQR_legacy_status(A, tau, Q, R, m, n, 64, 8));
        free(tau);
        if (mmqr_legacy_status(A, &tau, m, n, 64, 7) != QR_E_ARG || mmqr_legacy_status(A, &tau, 500, n, 64, 8) != QR_E_ARG) return 8;
        free(A); free(Q); free(R);
    }
    if (qrd_stub_live_allocations() != 0) {
        fprintf(stderr, "device-memory leak: %d allocations still live\n", qrd_stub_live_allocations());
        return 6;
    }
    printf("host layer sanitize run ok: %ld operand blocks checked\n", qrd_stub_launches());
    return 0;
}
