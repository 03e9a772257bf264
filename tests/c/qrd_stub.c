/* qrd_stub.c -- TEST-ONLY stand-in for the HIP launch layer (cuda-qr_amd/csrc/qr_device.h), so that the C host layer
 * (qr_host.c: schedule, plan cache, TSQR plan, per-device threads) can run under AddressSanitizer / UBSan / ThreadSanitizer on a
 * box without a GPU (SURVEY section 5 "race detection / sanitizers"; GPU sanitizers are not available on the pool).
 *
 * It computes NOTHING: "device memory" is host calloc, copies are memcpy, every kernel launch only CHECKS that the operand blocks
 * it was given lie inside live allocations (so an out-of-range workspace index in the schedule aborts here with a message) and
 * returns.  It is never linked into libmi355xqr.so and nothing under cuda-qr_amd/ refers to it. */
#define _POSIX_C_SOURCE 200809L
#include <pthread.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "../../cuda-qr_amd/csrc/qr_device.h"

static long g_stub_stalls = 0;      /* injected stalls of the one-launch panel (QRD_STUB_STALL_ONCE) */

#define MAXA 4096
static struct { char* p; size_t n; } g_alloc[MAXA];
static int g_nalloc;
static pthread_mutex_t g_mu = PTHREAD_MUTEX_INITIALIZER;
static __thread int t_dev = 0;
static long g_launches;

static void die(const char* what, const void* p, size_t bytes)
{
    fprintf(stderr, "qrd_stub: %s: block %p + %zu bytes is not inside a live device allocation\n", what, p, bytes);
    abort();
}

/* column-major block rows x cols (doubles) with leading dimension ld at p: inside ONE live allocation? */
static void chk(const char* what, const void* p, long ld, long rows, long cols)
{
    if (rows <= 0 || cols <= 0) return;
    if (!p) die(what, p, 0);
    if (ld < rows) { fprintf(stderr, "qrd_stub: %s: ld %ld < rows %ld\n", what, ld, rows); abort(); }
    const size_t bytes = sizeof(double) * ((size_t) ld * (size_t) (cols - 1) + (size_t) rows);
    const char* c = (const char*) p;
    int ok = 0;
    pthread_mutex_lock(&g_mu);
    for (int i = 0; i < g_nalloc && !ok; ++i)
        if (c >= g_alloc[i].p && c + bytes <= g_alloc[i].p + g_alloc[i].n) ok = 1;
    ++g_launches;
    pthread_mutex_unlock(&g_mu);
    if (!ok) die(what, p, bytes);
}
static void chkb(const char* what, const void* p, size_t bytes) { if (bytes) chk(what, p, (long) ((bytes + 7) / 8), (long) ((bytes + 7) / 8), 1); }

long qrd_stub_launches(void) { return g_launches; }
int qrd_stub_live_allocations(void) { return g_nalloc; }

void qrd_range_push(const char* name) { (void) name; }
void qrd_range_pop(void) { }
int qrd_init(void) { return 0; }
int qrd_gemm2_init(void) { return 0; }
int qrd_panel_tsqr_init(void) { return 0; }
int qrd_leaf_fused_init(void) { return 0; }
int qrd_panel_fused_init(void) { return 0; }

int qrd_malloc(void** p, size_t bytes)
{
    if (!p) return 1;
    const size_t n = (bytes + 7) & ~(size_t) 7;
    char* q = (char*) calloc(1, n ? n : 8);
    if (!q) return 2;
    pthread_mutex_lock(&g_mu);
    if (g_nalloc == MAXA) { pthread_mutex_unlock(&g_mu); free(q); return 2; }
    g_alloc[g_nalloc].p = q; g_alloc[g_nalloc].n = n ? n : 8; ++g_nalloc;
    pthread_mutex_unlock(&g_mu);
    *p = q;
    return 0;
}
int qrd_free(void* p)
{
    if (!p) return 0;
    pthread_mutex_lock(&g_mu);
    int found = 0;
    for (int i = 0; i < g_nalloc; ++i)
        if (g_alloc[i].p == (char*) p) { g_alloc[i] = g_alloc[--g_nalloc]; found = 1; break; }
    pthread_mutex_unlock(&g_mu);
    if (!found) { fprintf(stderr, "qrd_stub: qrd_free of %p, which is not a live allocation (double free?)\n", p); abort(); }
    free(p);
    return 0;
}
int qrd_memset(void* s, void* p, int v, size_t bytes) { (void) s; chkb("memset", p, bytes); memset(p, v, bytes); return 0; }
int qrd_h2d(void* s, void* d, const void* h, size_t bytes) { (void) s; chkb("h2d", d, bytes); memcpy(d, h, bytes); return 0; }
int qrd_d2h(void* s, void* h, const void* d, size_t bytes) { (void) s; chkb("d2h", d, bytes); memcpy(h, d, bytes); return 0; }
int qrd_d2d(void* s, void* dst, const void* src, size_t bytes) { (void) s; chkb("d2d dst", dst, bytes); chkb("d2d src", src, bytes); memmove(dst, src, bytes); return 0; }
int qrd_h2d_2d(void* s, void* d, size_t dp, const void* h, size_t hp, size_t w, size_t hgt)
{
    (void) s;
    for (size_t r = 0; r < hgt; ++r) { chkb("h2d_2d", (char*) d + r * dp, w); memcpy((char*) d + r * dp, (const char*) h + r * hp, w); }
    return 0;
}
int qrd_d2h_2d(void* s, void* h, size_t hp, const void* d, size_t dp, size_t w, size_t hgt)
{
    (void) s;
    for (size_t r = 0; r < hgt; ++r) { chkb("d2h_2d", (const char*) d + r * dp, w); memcpy((char*) h + r * hp, (const char*) d + r * dp, w); }
    return 0;
}

typedef struct stub_stream { int cus; } stub_stream;
int qrd_stream_create(void** s, int hp) { (void) hp; stub_stream* x = calloc(1, sizeof *x); if (!x) return 2; x->cus = 256; *s = x; return 0; }
int qrd_stream_create_cumask(void** s, int first, int count) { (void) first; stub_stream* x = calloc(1, sizeof *x); if (!x) return 2; x->cus = count; *s = x; return 0; }
int qrd_stream_destroy(void* s) { free(s); return 0; }
int qrd_stream_cus(void* s) { return s ? ((stub_stream*) s)->cus : 256; }
int qrd_stream_cus_coresident(void* s) { const int c = qrd_stream_cus(s); return c >= 32 ? c / 32 * 32 : c / 2; }
int qrd_stream_sync(void* s) { (void) s; return 0; }
int qrd_device_sync(void) { return 0; }
int qrd_capture_begin(void* s) { (void) s; return 0; }
int qrd_capture_end(void* s, void** exec) { (void) s; *exec = calloc(1, 8); return *exec ? 0 : 2; }
int qrd_graph_launch(void* exec, void* s) { (void) s; return exec ? 0 : 1; }
int qrd_graph_destroy(void* exec) { free(exec); return 0; }
int qrd_event_create(void** e) { *e = calloc(1, 8); return *e ? 0 : 2; }
int qrd_event_create_notiming(void** e) { return qrd_event_create(e); }
int qrd_event_create_timing(void** e) { return qrd_event_create(e); }
int qrd_event_destroy(void* e) { free(e); return 0; }
/* a record stamps the event with a process-wide tick (0.01 "ms" apart, or QRD_STUB_GATHER_MS after a collective: the joint fall-back
 * decision of the pipelined exchange can be driven from a test), so that elapsed times are ordered like the records */
static __thread double g_stub_tick = 0.0, g_stub_after_gather = 0.0;      /* per host thread = per rank */
int qrd_event_record(void* e, void* s)
{
    (void) s;
    if (!e) return 1;
    g_stub_tick += 0.01 + g_stub_after_gather;
    g_stub_after_gather = 0.0;
    *(double*) e = g_stub_tick;
    return 0;
}
int qrd_event_sync(void* e) { return e ? 0 : 1; }
int qrd_stream_wait_event(void* s, void* e) { (void) s; return e ? 0 : 1; }
int qrd_event_elapsed_ms(void* a, void* b, float* ms) { if (!a || !b) return 1; *ms = (float) (*(double*) b - *(double*) a); return 0; }
int qrd_device_count(int* n) { const char* e = getenv("QRD_STUB_NDEV"); *n = e ? atoi(e) : 4; return 0; }
int qrd_set_device(int d) { t_dev = d; return 0; }
int qrd_get_device(int* d) { *d = t_dev; return 0; }
int qrd_host_register(void* p, size_t b) { (void) p; (void) b; return 0; }
int qrd_host_unregister(void* p) { (void) p; return 0; }
const char* qrd_error_string(int e) { (void) e; return "stub device layer error"; }
int qrd_device_info(char* name, int len, int* cus, int* khz, size_t* mem)
{
    if (name && len > 0) snprintf(name, (size_t) len, "stub");
    if (cus) *cus = 256;
    if (khz) *khz = 2400000;
    if (mem) *mem = (size_t) 1 << 36;
    return 0;
}
int qrd_probe_mfma_f64(double* o) { o[0] = o[1] = o[2] = 1.0; return 0; }
int qrd_probe_copy(double* g) { *g = 1.0; return 0; }

/* ---- kernel launches: operand checks only ---- */
int qrd_gemm_nn(void* s, int M, int N, int K, double al, const double* A, int lda, const double* B, int ldb, double be, double* C, int ldc)
{ (void) s; (void) al; (void) be; chk("gemm_nn A", A, lda, M, K); chk("gemm_nn B", B, ldb, K, N); chk("gemm_nn C", C, ldc, M, N); return 0; }
int qrd_gemm_nn_update(void* s, int M, int N, int K, double al, const double* A, int lda, const double* B, int ldb, double be, double* C, int ldc)
{ return qrd_gemm_nn(s, M, N, K, al, A, lda, B, ldb, be, C, ldc); }
int qrd_gemm_nn_update2(void* s, int M, int N, int K, double al, const double* A, int lda, const double* B, int ldb, double be, double* C, int ldc)
{ return qrd_gemm_nn(s, M, N, K, al, A, lda, B, ldb, be, C, ldc); }
int qrd_gemm_tn(void* s, int M, int N, int K, double al, const double* A, int lda, const double* B, int ldb, double be, double* C, int ldc,
                double* slabs, size_t cap, const double* Tm, int ldt)
{
    (void) s; (void) al; (void) be;
    chk("gemm_tn A", A, lda, K, M); chk("gemm_tn B", B, ldb, K, N); chk("gemm_tn C", C, ldc, M, N);
    if (slabs) chkb("gemm_tn slabs", slabs, cap * sizeof(double));
    if (Tm) chk("gemm_tn T", Tm, ldt, M, M);
    return 0;
}
int qrd_gemm_tn_update(void* s, int M, int N, int K, double al, const double* A, int lda, const double* B, int ldb, double be, double* C, int ldc,
                       double* slabs, size_t cap)
{ return qrd_gemm_tn(s, M, N, K, al, A, lda, B, ldb, be, C, ldc, slabs, cap, NULL, 0); }
int qrd_gemm_tn_update_wide(void* s, int M, int N, int K, double al, const double* A, int lda, const double* B, int ldb, double be, double* C, int ldc,
                            double* slabs, size_t cap)
{ return qrd_gemm_tn(s, M, N, K, al, A, lda, B, ldb, be, C, ldc, slabs, cap, NULL, 0); }
int qrd_gemm_tn_dual(void* s, int N1, int N2, int K, const double* A, int lda, const double* B1, int ldb1, const double* B2, int ldb2,
                     const double* Tm, int ldt, double* W, int ldw, double* G2, int ldg, double* slabs, size_t cap)
{
    (void) s;
    chk("tn_dual A", A, lda, K, 32);
    if (N1 > 0) { chk("tn_dual B1", B1, ldb1, K, N1); chk("tn_dual W", W, ldw, 32, N1); }
    if (N2 > 0) { chk("tn_dual B2", B2, ldb2, K, N2); chk("tn_dual G2", G2, ldg, N2, 32); }
    chk("tn_dual T", Tm, ldt, 32, 32);
    chkb("tn_dual slabs", slabs, cap * sizeof(double));
    return (K & 64) ? -7 : 0;          /* both outcomes of the fused launch are exercised */
}
int qrd_gemm_nt_ok(int M, int N, int K, const double* A, int lda, const double* Bt, int ldbt, const double* C, int ldc)
{ (void) A; (void) Bt; (void) C; (void) lda; (void) ldbt; (void) ldc; return M % 128 == 0 && N % 128 == 0 && K % 16 == 0; }
int qrd_gemm_nt4_ok(int M, int N, int K, const double* A, int lda, const double* Bt, int ldbt, const double* C, int ldc)
{ (void) A; (void) Bt; (void) C; (void) ldbt; (void) ldc; return M >= 128 && N >= 64 && N % 64 == 0 && K >= 32 && K % 16 == 0 && (M % 128 == 0 || (M % 2 == 0 && (M + 127) / 128 * 128 <= lda)); }
int qrd_gemm_nt(void* s, int M, int N, int K, int sign, const double* A, int lda, const double* Bt, int ldbt, double* C, int ldc, int gm,
                unsigned long long* st)
{ (void) s; (void) sign; (void) gm; (void) st; chk("gemm_nt A", A, lda, M, K); chk("gemm_nt Bt", Bt, ldbt, N, K); chk("gemm_nt C", C, ldc, M, N); return 0; }
static void leaf_chk(const char* w, double* P, int ld, int mk, int wd, double* tau, double* T, int ldt, double* Vw, int ldv)
{ chk(w, P, ld, mk, wd); chk(w, tau, wd, wd, 1); chk(w, T, ldt, wd, wd); chk(w, Vw, ldv, mk, wd); }
size_t qrd_panel_ws_size(int m) { return (size_t) 80 * (size_t) (m > 0 ? m : 1) + 65536; }
int qrd_panel_tsqr(void* s, double* P, int ld, int mk, int w, double* tau, double* T, int ldt, double* Vw, int ldv, double* ws, int mcap)
{ (void) s; leaf_chk("panel_tsqr", P, ld, mk, w, tau, T, ldt, Vw, ldv); chkb("panel ws", ws, sizeof(double) * qrd_panel_ws_size(mcap)); return 0; }
int qrd_panel_cholqr(void* s, double* P, int ld, int mk, int w, double* tau, double* T, int ldt, double* Vw, int ldv, double* ws, int mcap,
                     double* cws, double* slabs, size_t cap, int gn)
{
    (void) gn;
    qrd_panel_tsqr(s, P, ld, mk, w, tau, T, ldt, Vw, ldv, ws, mcap);
    chkb("cholqr ws", cws, sizeof(double) * QRD_CHOLQR_WS); chkb("cholqr slabs", slabs, cap * sizeof(double));
    return 0;
}
int qrd_panel_cholqr_ep(void* s, double* P, int ld, int mk, int w, double* tau, double* T, int ldt, double* Vw, int ldv, double* ws, int mcap,
                        double* cws, double* slabs, size_t cap, int gn, int N1, const double* B1, int ldb1, int N2, const double* B2, int ldb2,
                        double* W, int ldw, double* G2, int ldg, double* eps, size_t ecap, int* did)
{
    qrd_panel_cholqr(s, P, ld, mk, w, tau, T, ldt, Vw, ldv, ws, mcap, cws, slabs, cap, gn);
    if (N1 > 0) { chk("ep B1", B1, ldb1, mk, N1); chk("ep W", W, ldw, 32, N1); }
    if (N2 > 0) { chk("ep B2", B2, ldb2, mk, N2); chk("ep G2", G2, ldg, N2, 32); }
    chkb("ep slabs", eps, ecap * sizeof(double));
    if (ecap < (size_t) 32 * (size_t) (N1 + N2)) { fprintf(stderr, "qrd_stub: early-product slab buffer too small\n"); abort(); }
    *did = (mk & 64) ? 0 : 1;          /* both outcomes are exercised */
    return 0;
}
/* the full-width tall panel: same shape rule as the real layer; panels whose height has bit 12 set are REFUSED by the (stub) guard, so
 * that the host's fall-back to the leaf chain on the untouched panel is exercised as well */
size_t qrd_panel_cqr_ws_doubles(void) { return 12 * 128 * 128 + 128 + 64 + 256 * 36 * 256; }
int qrd_panel_cqr_q(void* s, double* A, int lda, int mk, int w, double* tau, double* T, int ldt, double* Vw, int ldv, double* ws, int* status,
                    double* Qb, int ldq, unsigned* hflag, unsigned seq)
{
    (void) s;
    if (!qrd_panel_cqr_ok(mk, w)) return -7;
    leaf_chk("panel_cqr", A, lda, mk, w, tau, T, ldt, Vw, ldv);
    if (Qb) chk("panel_cqr Q", Qb, ldq, mk, w);
    chkb("cqr ws", ws, sizeof(double) * qrd_panel_cqr_ws_doubles()); chkb("cqr status", status, 4 * sizeof(int));
    status[0] = 0;
    if (mk & 4096) { status[0] = 1; if (!hflag) status[1] += 1; }
    if (hflag) __atomic_store_n(hflag, 2u * seq + (unsigned) status[0], __ATOMIC_RELEASE);
    return 0;
}
int qrd_panel_cqr_p(void* s, double* A, int lda, int mk, int w, double* tau, double* T, int ldt, double* Vw, int ldv, double* ws, int* status,
                    double* Qb, int ldq, unsigned* hflag, unsigned seq, int park)
{
    if (park && (!Qb || Qb == A)) return -7;
    return qrd_panel_cqr_q(s, A, lda, mk, w, tau, T, ldt, Vw, ldv, ws, status, Qb, ldq, hflag, seq);
}
/* the stub's guard accepts a retried panel unless bit 7 of its height is set too (65408 = 0xFF80: refused twice; 61440 + ...: see host_sanitize.c) */
int qrd_panel_cqr_retry(void* s, double* A, int lda, int mk, int w, double* tau, double* T, int ldt, double* Vw, int ldv, double* ws, int* status,
                        double* Qb, int ldq, unsigned* hflag, unsigned seq, int park)
{
    (void) s; (void) park;
    if (!Qb || Qb == A || Qb == Vw) return -7;
    leaf_chk("cqr retry", A, lda, mk, w, tau, T, ldt, Vw, ldv);
    chk("cqr retry Q", Qb, ldq, mk, w);
    chkb("cqr ws", ws, sizeof(double) * qrd_panel_cqr_ws_doubles()); chkb("cqr status", status, 4 * sizeof(int));
    status[0] = (mk & 128) ? 1 : 0;
    if (hflag) __atomic_store_n(hflag, 2u * seq + (unsigned) status[0], __ATOMIC_RELEASE);
    return 0;
}
int qrd_panel_cqr_restore_r(void* s, double* A, int lda, int w, const double* ws, const int* status)
{ (void) s; (void) status; chk("cqr restore A", A, lda, w, w); chkb("cqr ws", ws, sizeof(double) * qrd_panel_cqr_ws_doubles()); return 0; }
int qrd_panel_cqr_r_block(void* s, const double* ws, int w, double* D, int ldd)
{ (void) s; chk("cqr r block", D, ldd, w, w); chkb("cqr ws", ws, sizeof(double) * qrd_panel_cqr_ws_doubles()); return 0; }
int qrd_panel_cqr(void* s, double* A, int lda, int mk, int w, double* tau, double* T, int ldt, double* Vw, int ldv, double* ws, int* status)
{
    return qrd_panel_cqr_q(s, A, lda, mk, w, tau, T, ldt, Vw, ldv, ws, status, NULL, 0, NULL, 0u);
}
int qrd_host_word_alloc(unsigned** host, unsigned** dev)
{
    unsigned* w = (unsigned*) calloc(16, sizeof(unsigned));
    if (!w) return 2;
    *host = w; *dev = w;
    return 0;
}
int qrd_host_word_free(unsigned* host) { free(host); return 0; }
int qrd_trsm_gt(void* s, int kw, int nc, const double* G, int ldg, const double* T, int ldt, const double* Y, int ldy, double* W, int ldw)
{
    (void) s;
    if (kw < 32 || kw > 256 || kw % 32 || nc < 16 || nc % 16) return -7;
    chk("trsm G", G, ldg, kw, kw); chk("trsm T", T, ldt, kw, kw); chk("trsm Y", Y, ldy, kw, nc); chk("trsm W", W, ldw, kw, nc);
    return 0;
}
int qrd_transpose(void* s, int rows, int cols, const double* S, int lds, double* D, int ldd)
{
    (void) s;
    chk("transpose S", S, lds, rows, cols); chk("transpose D", D, ldd, cols, rows);
    return 0;
}
int qrd_panel_cqr_init(void) { return 0; }
int qrd_panel_cqr_ok(int mk, int w) { return w >= 32 && w <= 128 && w % 32 == 0 && mk >= 2 * w; }
double* qrd_panel_cqr_g1(double* ws) { return ws; }
double* qrd_panel_cqr_g2(double* ws) { return ws + 128 * 128; }
int qrd_panel_cqr_stage1(void* s, const double* A, int lda, int mk, int w, double* Vw, int ldv, double* ws, int* status)
{
    (void) s;
    chk("cqr stage1 A", A, lda, mk, w); chk("cqr stage1 Vw", Vw, ldv, mk, w);
    chkb("cqr ws", ws, sizeof(double) * qrd_panel_cqr_ws_doubles()); chkb("cqr status", status, 4 * sizeof(int));
    if (mk & 4096) status[0] = 1;
    return 0;
}
int qrd_panel_cqr_stage2(void* s, double* A, int lda, int mk, int w, double* tau, double* T, int ldt, double* Vw, int ldv, double* ws, int* status)
{
    (void) s;
    leaf_chk("cqr stage2", A, lda, mk, w, tau, T, ldt, Vw, ldv);
    chkb("cqr ws", ws, sizeof(double) * qrd_panel_cqr_ws_doubles()); chkb("cqr status", status, 4 * sizeof(int));
    return 0;
}
size_t qrd_panel_fused_ws_doubles(void) { return 700000; }
/* same shape rules as the real launch layer (whole leaves, <= 256 columns, <= 32 x 256 rows, rows a multiple of 4, aligned operands);
 * heights with bit 9 set are declined so that both routes of factor_panel are exercised */
int qrd_panel_fused_merges_t(int wh, int with_gram) { return wh == 64 && with_gram; }
int qrd_panel_fused_ok(void* s, const double* A, int lda, int mk, int wh, const double* Vw, int ldv)
{
    (void) s;
    if (wh < 32 || wh > 256 || wh % 32 || mk < wh || mk % 4 || mk > 8192 || (mk & 512)) return 0;
    return !(((uintptr_t) A & 15) || ((uintptr_t) Vw & 15) || lda % 2 || ldv % 2);
}
int qrd_panel_fused(void* s, double* A, int lda, int mk, int wh, double* tau, double* T, int ldt, double* Vw, int ldv, double* G, int ldg,
                    double* ws, unsigned* epoch, int* status)
{
    if (!qrd_panel_fused_ok(s, A, lda, mk, wh, Vw, ldv)) return -7;
    leaf_chk("panel_fused", A, lda, mk, wh, tau, T, ldt, Vw, ldv);
    if (G) chk("panel_fused G", G, ldg, wh, wh);           /* NULL: the caller forms V^T V itself */
    chkb("panel_fused ws", ws, sizeof(double) * qrd_panel_fused_ws_doubles());
    chkb("panel_fused status", status, 4 * sizeof(int));
    *epoch += 1024u;
    /* QRD_STUB_STALL_ONCE=1 (host_sanitize.c): the next one-launch panel reports a timed-out hand-off, once -- the host-pointer entry
     * points must notice (QR_E_STALL at their plan sync) and factor again with the route off */
    if (getenv("QRD_STUB_STALL_ONCE")) { status[1] = 1; unsetenv("QRD_STUB_STALL_ONCE"); ++g_stub_stalls; }
    return 0;
}
int qrd_panel_fused_rows(void* s, double* A, int lda, int mk, int wh, double* tau, double* T, int ldt, double* Vw, int ldv, double* G, int ldg,
                         double* ws, unsigned* epoch, int* status, int rows)
{ (void) rows; return qrd_panel_fused(s, A, lda, mk, wh, tau, T, ldt, Vw, ldv, G, ldg, ws, epoch, status); }
long qrd_stub_stalls(void) { return g_stub_stalls; }
int qrd_slab_reduce(void* s, int M, int N, int ns, const double* slabs, int lds, size_t stride, double* out, int ldo)
{ (void) s; chk("slab_reduce in", slabs, lds, M, N); (void) ns; (void) stride; chk("slab_reduce out", out, ldo, M, N); return 0; }
int qrd_leaf_update_gram(void* s, int mk, int N, const double* V, int ldv, const double* W, double* C, int ldc, double* gs, size_t cap, int gy,
                         int* nslab)
{
    (void) s; (void) gy;
    if (mk < 1024) return -7;
    chk("lug V", V, ldv, mk, 32); chk("lug W", W, 32, 32, N); chk("lug C", C, ldc, mk, N);
    if (gs) chkb("lug slabs", gs, cap * sizeof(double));
    if (nslab) *nslab = gs ? 8 : 0;
    return 0;
}
int qrd_larft(void* s, int nbp, int ib, const double* G, int ldg, const double* tau, double* T, int ldt, double* Tt, int bd, double* X, int ldx)
{
    (void) s; (void) ib; (void) bd;
    chk("larft G", G, ldg, nbp, nbp); chk("larft tau", tau, nbp, nbp, 1); chk("larft T", T, ldt, nbp, nbp); chk("larft X", X, ldx, nbp, nbp);
    if (Tt) chk("larft Tt", Tt, ldt, nbp, nbp);
    return 0;
}
/* The pure DATA MOVERS move data (still no arithmetic: the factorisations above them are no-ops on the values), so that a test can
 * follow a tagged R factor through pack -> exchange -> stack -> extract (tests/test_tsqr_cplan_gloo.py: the C qr_tsqr_plan over gloo).
 * Every copy_block is also logged (source, destination, shape): the stacking order of the gathered factors is read off the log. */
#define STUB_COPYLOG 256
static __thread struct { const double* S; double* D; int lds, ldd, r, c; } t_copylog[STUB_COPYLOG];
static __thread long t_ncopy = 0;
long qrd_stub_copy_count(void) { return t_ncopy; }
int qrd_stub_copy_entry(long i, const double** S, double** D, int* lds, int* ldd, int* r, int* c)
{
    if (i < 0 || i >= t_ncopy || t_ncopy - i > STUB_COPYLOG) return 1;
    *S = t_copylog[i % STUB_COPYLOG].S; *D = t_copylog[i % STUB_COPYLOG].D; *lds = t_copylog[i % STUB_COPYLOG].lds;
    *ldd = t_copylog[i % STUB_COPYLOG].ldd; *r = t_copylog[i % STUB_COPYLOG].r; *c = t_copylog[i % STUB_COPYLOG].c;
    return 0;
}
int qrd_zero_block(void* s, double* A, int ld, int r, int c)
{
    (void) s; chk("zero_block", A, ld, r, c);
    for (int j = 0; j < c; ++j) memset(A + (size_t) j * ld, 0, sizeof(double) * (size_t) r);
    return 0;
}
int qrd_extract_v(void* s, const double* P, int ld, int mk, int w, double* V, int ldv) { (void) s; chk("extract_v P", P, ld, mk, w); chk("extract_v V", V, ldv, mk, w); return 0; }
int qrd_extract_r(void* s, const double* A, int lda, int m, int n, double* R, int ldr, int rr)
{
    (void) s; chk("extract_r A", A, lda, m, n); chk("extract_r R", R, ldr, rr, n);
    for (int c = 0; c < n; ++c)
        for (int i = 0; i < rr; ++i) R[(size_t) c * ldr + i] = (i <= c && i < m) ? A[(size_t) c * lda + i] : 0.0;
    return 0;
}
int qrd_extract_r_block(void* s, const double* A, int lda, int k, int w, double* R, int ldr, int rr)
{
    (void) s; chk("extract_r_block A", A + (size_t) k * lda, lda, k + w, w); chk("extract_r_block R", R, ldr, rr, w);
    for (int j = 0; j < w; ++j)
        for (int i = 0; i < rr; ++i) R[(size_t) j * ldr + i] = (i <= k + j) ? A[(size_t) (k + j) * lda + i] : 0.0;
    return 0;
}
int qrd_set_identity(void* s, double* C, int ld, int r, int c, int ro)
{
    (void) s; chk("set_identity", C, ld, r, c);
    for (int j = 0; j < c; ++j)
        for (int i = 0; i < r; ++i) C[(size_t) j * ld + i] = (i + ro == j) ? 1.0 : 0.0;
    return 0;
}
int qrd_copy_blocks(void* s, const double* S, int lds, size_t ss, double* D, int ldd, size_t ds, int r, int c, int batch)
{
    (void) s;
    for (int q = 0; q < batch; ++q) {
        chk("copy_blocks S", S + q * ss, lds, r, c); chk("copy_blocks D", D + q * ds, ldd, r, c);
        for (int j = 0; j < c; ++j) memmove(D + q * ds + (size_t) j * ldd, S + q * ss + (size_t) j * lds, sizeof(double) * (size_t) r);
    }
    return 0;
}
int qrd_copy_block(void* s, const double* S, int lds, double* D, int ldd, int r, int c)
{
    (void) s; chk("copy_block S", S, lds, r, c); chk("copy_block D", D, ldd, r, c);
    for (int j = 0; j < c; ++j) memmove(D + (size_t) j * ldd, S + (size_t) j * lds, sizeof(double) * (size_t) r);
    t_copylog[t_ncopy % STUB_COPYLOG].S = S; t_copylog[t_ncopy % STUB_COPYLOG].D = D; t_copylog[t_ncopy % STUB_COPYLOG].lds = lds;
    t_copylog[t_ncopy % STUB_COPYLOG].ldd = ldd; t_copylog[t_ncopy % STUB_COPYLOG].r = r; t_copylog[t_ncopy % STUB_COPYLOG].c = c;
    ++t_ncopy;
    return 0;
}
int qrd_fill_uniform(void* s, double* A, int ld, long long rows, int cols, long long ro, long long tr, unsigned long long seed)
{ (void) s; (void) ro; (void) tr; (void) seed; chk("fill", A, ld, (long) rows, cols); return 0; }
double qrd_hash_uniform_host(unsigned long long seed, unsigned long long idx) { (void) seed; (void) idx; return 0.5; }
int qrd_diff_norm(void* s, const double* X, int ldx, const double* Y, int ldy, long long rows, int cols, long long ro, long long tr,
                  unsigned long long seed, int si, double* out)
{ (void) s; (void) ro; (void) tr; (void) seed; (void) si; chk("diffnorm X", X, ldx, (long) rows, cols); if (Y) chk("diffnorm Y", Y, ldy, (long) rows, cols); out[0] = 0.0; out[1] = 1.0; return 0; }

size_t qrd_legacy_ws_size(int m, int PR, int PC) { return (size_t) (1 + (m - PR) / (PR - PC)) * 2 * (size_t) PC * 64; }
int qrd_legacy_shape_ok(int m, int n, int PR, int PC)
{ return PR >= 2 && PR <= 64 && (PC == 2 || PC == 4 || PC == 8 || PC == 16) && PC < PR && m >= PR && n >= PC && n % PC == 0 && n <= m && (m - PR) % (PR - PC) == 0; }
int qrd_legacy_panel(void* s, double* A, int m, int n, int PR, int PC, int rp, int pc, int pcCount, double* tau, double* wy)
{
    (void) s; (void) pc;
    if (pcCount < 0 || pcCount >= n / PC) { fprintf(stderr, "qrd_stub: legacy panel index out of range\n"); abort(); }
    chk("legacy A", A, m, m, n);
    chkb("legacy tau all", tau, sizeof(double) * (size_t) rp * (size_t) (n / PC) * PC); chkb("legacy wy", wy, sizeof(double) * qrd_legacy_ws_size(m, PR, PC));
    return 0;
}
int qrd_legacy_formq(void* s, const double* A, const double* tau, int m, int n, int PR, int PC, int rp, double* Q)
{ (void) s; (void) PR; chk("legacy A", A, m, m, n); chkb("legacy tau", tau, sizeof(double) * (size_t) rp * (size_t) (n / PC) * PC); chk("legacy Q", Q, m, m, m); return 0; }

/* ---- "RCCL": a thread-level all-gather (one thread per device, the way qr_thin_mgpu drives it) ---- */
typedef struct stub_world { int n; pthread_barrier_t bar; const double* send[64]; } stub_world;
typedef struct stub_comm { stub_world* w; int rank; } stub_comm;
int qrd_comm_init_all(void** comms, int n, const int* devs)
{
    (void) devs;
    stub_world* w = calloc(1, sizeof *w);
    if (!w) return QRD_E_RCCL;
    w->n = n;
    pthread_barrier_init(&w->bar, NULL, (unsigned) n);
    for (int i = 0; i < n; ++i) { stub_comm* c = calloc(1, sizeof *c); c->w = w; c->rank = i; comms[i] = c; }
    return 0;
}
int qrd_comm_unique_id(void* id) { memset(id, 7, QRD_UNIQUE_ID_BYTES); return 0; }
int qrd_comm_init_rank(void** comm, int n, const void* id, int rank) { (void) comm; (void) n; (void) id; (void) rank; return QRD_E_NORCCL; }
int qrd_comm_count(void* comm, int* n) { *n = ((stub_comm*) comm)->w->n; return 0; }
const char* qrd_rccl_error_string(int r) { (void) r; return "stub rccl error"; }
int qrd_comm_destroy(void* comm)
{
    stub_comm* c = (stub_comm*) comm;
    if (!c) return 0;
    if (c->rank == 0) { pthread_barrier_destroy(&c->w->bar); free(c->w); }
    free(c);
    return 0;
}
int qrd_allgather_f64(void* comm, void* stream, const double* send, double* recv, size_t count)
{
    (void) stream;
    stub_comm* c = (stub_comm*) comm;
    chkb("allgather send", send, count * sizeof(double));
    chkb("allgather recv", recv, count * sizeof(double) * (size_t) c->w->n);
    c->w->send[c->rank] = send;
    pthread_barrier_wait(&c->w->bar);
    for (int q = 0; q < c->w->n; ++q) memcpy(recv + (size_t) q * count, c->w->send[q], count * sizeof(double));
    pthread_barrier_wait(&c->w->bar);
    { const char* e = getenv("QRD_STUB_GATHER_MS"); if (e) g_stub_after_gather = atof(e); }
    return 0;
}
