/* qr_device.h -- internal C interface between the C host layer (qr_host.c) and the HIP launch
 * layer (qr_kernels.hip).  Plain C types only: the host layer never includes a HIP header.
 * All functions return 0 on success or a hipError_t / negative library code. */
#ifndef QR_DEVICE_H
#define QR_DEVICE_H
#include <stddef.h>
#ifdef __cplusplus
extern "C" {
#endif

int qrd_init(void);
int qrd_gemm_nn(void* stream, int M, int N, int K, double alpha, const double* A, int lda, const double* B,
                int ldb, double beta, double* C, int ldc);
int qrd_gemm_tn(void* stream, int M, int N, int K, double alpha, const double* A, int lda, const double* B,
                int ldb, double beta, double* C, int ldc, double* slabs, size_t slab_cap, const double* Tm,
                int ldt);
int qrd_gemm_nn_update(void* stream, int M, int N, int K, double alpha, const double* A, int lda, const double* B,
                       int ldb, double beta, double* C, int ldc);
int qrd_gemm_nn_update2(void* stream, int M, int N, int K, double alpha, const double* A, int lda, const double* B,
                        int ldb, double beta, double* C, int ldc);
int qrd_gemm_tn_dual(void* stream, int N1, int N2, int K, const double* A, int lda, const double* B1, int ldb1, const double* B2,
                     int ldb2, const double* Tm, int ldt, double* W, int ldw, double* G2, int ldg, double* slabs, size_t slab_cap);
int qrd_gemm_tn_update(void* stream, int M, int N, int K, double alpha, const double* A, int lda, const double* B,
                       int ldb, double beta, double* C, int ldc, double* slabs, size_t slab_cap);
int qrd_gemm_tn_update_wide(void* stream, int M, int N, int K, double alpha, const double* A, int lda, const double* B,
                            int ldb, double beta, double* C, int ldc, double* slabs, size_t slab_cap);
/* second-generation wide update (qr_gemm_nt.hip): W kept transposed, direct-to-LDS tile loads */
int qrd_gemm2_init(void);
int qrd_gemm_nt_ok(int M, int N, int K, const double* A, int lda, const double* Bt, int ldbt, const double* C, int ldc);
/* ... to the four-workgroup kernel of the trailing update: also N = 64 (mod 128), and M = 64 (mod 128) where A is readable to the next multiple of 128 rows */
int qrd_gemm_nt4_ok(int M, int N, int K, const double* A, int lda, const double* Bt, int ldbt, const double* C, int ldc);
int qrd_gemm_nt(void* stream, int M, int N, int K, int sign, const double* A, int lda, const double* Bt, int ldbt,
                double* C, int ldc, int gm, unsigned long long* stamps);
size_t qrd_panel_ws_size(int m);
int qrd_panel_tsqr_init(void);
int qrd_panel_tsqr(void* stream, double* P, int ld, int mk, int w, double* tau, double* T, int ldt, double* Vw, int ldv,
                   double* ws, int m_cap);
int qrd_slab_reduce(void* stream, int M, int N, int nslab, const double* slabs, int lds, size_t stride, double* out, int ldo);
#define QRD_CHOLQR_WS (10 * 32 * 32 + 16)      /* G1, G2, R1, M, guard words; then L1 and the four fold matrices of the early product */
/* gram_nslab > 0: `slabs` already holds that many 32 x 32 partial Gram matrices of this leaf (qrd_leaf_update_gram) */
int qrd_panel_cholqr(void* stream, double* P, int ld, int mk, int w, double* tau, double* T, int ldt, double* Vw, int ldv,
                     double* ws, int m_cap, double* cws, double* slabs, size_t slab_cap, int gram_nslab);
/* the leaf and its in-panel products in one call: the long-K product runs in the same launch as the one-workgroup reconstruction */
int qrd_panel_cholqr_ep(void* stream, double* P, int ld, int mk, int w, double* tau, double* T, int ldt, double* Vw, int ldv,
                        double* ws, int m_cap, double* cws, double* slabs, size_t slab_cap, int gram_nslab,
                        int N1, const double* B1, int ldb1, int N2, const double* B2, int ldb2, double* W, int ldw, double* G2, int ldg,
                        double* ep_slabs, size_t ep_slab_cap, int* did);
/* fused kernels of the leaf chain (qr_leaf_fused.hip) */
int qrd_leaf_fused_init(void);
int qrd_leaf_update_gram(void* stream, int mk, int N, const double* V, int ldv, const double* W, double* C, int ldc, double* gslabs,
                         size_t gslab_cap, int gy, int* nslab);
int qrd_gemm_nn_batch(void* stream, int M, int N, int K, double alpha, const double* A, int lda, size_t sA,
                      const double* B, int ldb, size_t sB, double beta, double* C, int ldc, size_t sC, int batch);
int qrd_larft(void* stream, int nbp, int ib, const double* G, int ldg, const double* tau, double* T, int ldt,
              double* Tt, int build_diag, double* X, int ldx);
int qrd_zero_block(void* stream, double* A, int ld, int rows, int cols);
/* W = T^T Y from the panel's Gram matrix G = V^T V and the leaves' 32 x 32 T blocks on the diagonal of T, no merged T needed (forward
 * substitution over the leaves); -7: shape not taken (kw % 32, nc % 16, kw > 256) */
int qrd_trsm_gt(void* stream, int kw, int nc, const double* G, int ldg, const double* T, int ldt, const double* Y, int ldy, double* W, int ldw);
int qrd_transpose(void* stream, int rows, int cols, const double* S, int lds, double* D, int ldd);   /* D (cols x rows) = S^T */
int qrd_extract_v(void* stream, const double* P, int ld, int mk, int w, double* V, int ldv);
int qrd_extract_r(void* stream, const double* A, int lda, int m, int n, double* R, int ldr, int rrows);
int qrd_extract_r_block(void* stream, const double* A, int lda, int k, int w, double* R, int ldr, int rrows);
int qrd_set_identity(void* stream, double* C, int ld, int rows, int cols, int row_off);
int qrd_copy_block(void* stream, const double* S, int lds, double* D, int ldd, int rows, int cols);
int qrd_copy_blocks(void* stream, const double* S, int lds, size_t sstride, double* D, int ldd, size_t dstride, int rows, int cols, int batch);
int qrd_fill_uniform(void* stream, double* A, int ld, long long rows, int cols, long long row_off,
                     long long total_rows, unsigned long long seed);
double qrd_hash_uniform_host(unsigned long long seed, unsigned long long idx);
int qrd_diff_norm(void* stream, const double* X, int ldx, const double* Y, int ldy, long long rows, int cols,
                  long long row_off, long long total_rows, unsigned long long seed, int sub_identity, double* out);

/* a TALL panel (<= 128 columns) at its full width: CholeskyQR2 + Householder reconstruction in three passes over the panel
 * (qr_panel_cqr.hip): V -> Vw and below the diagonal of A, R, T (complete), tau.  status: 4 device ints, zeroed by the call;
 * status[0] = 1 afterwards: the guard refused the panel and A is untouched.  -7: shape not taken (qrd_panel_cqr_ok).
 * stage1 / stage2 + g1 / g2: the same in two halves with the Gram matrices supplied by the caller (development checks) */
size_t qrd_panel_cqr_ws_doubles(void);
int qrd_panel_cqr_init(void);
int qrd_panel_cqr_ok(int mk, int w);
int qrd_panel_cqr(void* stream, double* A, int lda, int mk, int w, double* tau, double* T, int ldt, double* Vw, int ldv, double* ws, int* status);
/* the same with Q in a buffer of its own (Qb: mk x w, ld ldq): a refused panel then leaves Vw untouched as well; status[1] counts refused
 * panels (sticky, never reset by the call); hflag (NULL: none): device address of a host word that receives 2 * seq + refused as soon as
 * the verdict exists -- two thirds into the panel, so the host can decide what comes next while the last pass is still running */
int qrd_panel_cqr_q(void* stream, double* A, int lda, int mk, int w, double* tau, double* T, int ldt, double* Vw, int ldv, double* ws, int* status,
                    double* Qb, int ldq, unsigned* hflag, unsigned seq);
/* parked form (park != 0): V written once, into A (top block: unit lower, zeros above); R stays in ws until qrd_panel_cqr_restore_r;
 * qrd_panel_cqr_r_block: R (upper, zeros below) into a w x w block of its own */
int qrd_panel_cqr_p(void* stream, double* A, int lda, int mk, int w, double* tau, double* T, int ldt, double* Vw, int ldv, double* ws, int* status,
                    double* Qb, int ldq, unsigned* hflag, unsigned seq, int park);
/* the retry of a panel that call has just refused, preconditioned (shifted CholeskyQR3): same arguments, Qb a buffer of its own */
int qrd_panel_cqr_retry(void* stream, double* A, int lda, int mk, int w, double* tau, double* T, int ldt, double* Vw, int ldv, double* ws, int* status,
                    double* Qb, int ldq, unsigned* hflag, unsigned seq, int park);
int qrd_panel_cqr_restore_r(void* stream, double* A, int lda, int w, const double* ws, const int* status);   /* status[0] != 0 (refused panel): A stays untouched */
int qrd_panel_cqr_r_block(void* stream, const double* ws, int w, double* D, int ldd);
double* qrd_panel_cqr_g1(double* ws);
double* qrd_panel_cqr_g2(double* ws);
int qrd_panel_cqr_stage1(void* stream, const double* A, int lda, int mk, int w, double* Vw, int ldv, double* ws, int* status);
int qrd_panel_cqr_stage2(void* stream, double* A, int lda, int mk, int w, double* tau, double* T, int ldt, double* Vw, int ldv, double* ws,
                         int* status);
/* a whole outer panel (<= 256 columns, <= 8192 rows) in ONE launch (qr_panel_fused.hip) */
size_t qrd_panel_fused_ws_doubles(void);
int qrd_panel_fused_init(void);
int qrd_panel_fused_merges_t(int wh, int with_gram);   /* 1: a launch with these arguments leaves the panel's complete T (no merge tree behind it) */
int qrd_panel_fused_ok(void* stream, const double* A, int lda, int mk, int wh, const double* Vw, int ldv);
int qrd_panel_fused(void* stream, double* A, int lda, int mk, int wh, double* tau, double* T, int ldt, double* Vw, int ldv,
                    double* G, int ldg, double* ws, unsigned* epoch, int* status);
/* the same with the rows per row workgroup given (0 = the library's choice, 128, 256): kernel unit tests */
int qrd_panel_fused_rows(void* stream, double* A, int lda, int mk, int wh, double* tau, double* T, int ldt, double* Vw, int ldv,
                    double* G, int ldg, double* ws, unsigned* epoch, int* status, int want_rows);

/* the reference's sliding-window schedule on the device (qr_legacy.hip): legacy-layout shim */
size_t qrd_legacy_ws_size(int m, int PR, int PC);
int qrd_legacy_shape_ok(int m, int n, int PR, int PC);
int qrd_legacy_panel(void* stream, double* A, int m, int n, int PR, int PC, int rowPanels, int pc, int pcCount, double* tau, double* wy);
int qrd_legacy_formq(void* stream, const double* A, const double* tau, int m, int n, int PR, int PC, int rowPanels, double* Q);

/* RCCL glue (qr_comm.hip): librccl is dlopen()ed on first use, never linked */
#define QRD_E_NORCCL (-120)   /* librccl.so could not be loaded */
#define QRD_E_RCCL   (-130)   /* an RCCL call failed: -130 - ncclResult_t */
#define QRD_UNIQUE_ID_BYTES 128
int qrd_comm_init_all(void** comms, int n, const int* devs);
int qrd_comm_unique_id(void* id);
int qrd_comm_init_rank(void** comm, int nranks, const void* id, int rank);
int qrd_comm_count(void* comm, int* n);
const char* qrd_rccl_error_string(int r);
int qrd_comm_destroy(void* comm);
int qrd_allgather_f64(void* comm, void* stream, const double* send, double* recv, size_t count);

/* optional roctx ranges around the host-side issue of a step's phases (MI355XQR_ROCTX=1; rocprofv3 --marker-trace) */
void qrd_range_push(const char* name);
void qrd_range_pop(void);

int qrd_malloc(void** p, size_t bytes);
int qrd_free(void* p);
int qrd_memset(void* stream, void* p, int v, size_t bytes);
int qrd_h2d(void* stream, void* d, const void* h, size_t bytes);
int qrd_d2h(void* stream, void* h, const void* d, size_t bytes);
int qrd_d2d(void* stream, void* dst, const void* src, size_t bytes);
int qrd_h2d_2d(void* stream, void* d, size_t dpitch, const void* h, size_t hpitch, size_t width, size_t height);
int qrd_d2h_2d(void* stream, void* h, size_t hpitch, const void* d, size_t dpitch, size_t width, size_t height);
int qrd_stream_create(void** s, int high_priority);
int qrd_stream_create_cumask(void** s, int first, int count);
int qrd_stream_destroy(void* s);
int qrd_capture_begin(void* s);
int qrd_capture_end(void* s, void** exec);
int qrd_graph_launch(void* exec, void* s);
int qrd_graph_destroy(void* exec);
int qrd_stream_sync(void* s);
int qrd_device_sync(void);
int qrd_event_create(void** e);
int qrd_event_create_timing(void** e);          /* timing brackets only: recorded without a system-scope fence */
int qrd_event_create_notiming(void** e);
int qrd_event_destroy(void* e);
int qrd_event_record(void* e, void* s);
int qrd_event_sync(void* e);
int qrd_stream_wait_event(void* s, void* e);
int qrd_event_elapsed_ms(void* a, void* b, float* ms);
int qrd_device_count(int* n);
int qrd_set_device(int d);
int qrd_get_device(int* d);
int qrd_stream_cus(void* s);
int qrd_stream_cus_coresident(void* s);   /* ... that a launch of workgroups waiting for each other may count on (whole multiples of 32 of a mask) */
/* one 32-bit word of pinned host memory mapped into the device (kernels publish small verdicts into it with system scope) */
int qrd_host_word_alloc(unsigned** host, unsigned** dev);
int qrd_host_word_free(unsigned* host);
int qrd_host_register(void* p, size_t bytes);
int qrd_host_unregister(void* p);
const char* qrd_error_string(int e);
int qrd_device_info(char* name, int name_len, int* cus, int* clock_khz, size_t* mem_bytes);
int qrd_probe_mfma_f64(double* out3);
int qrd_probe_copy(double* gbps);

#define QRD_LEAFW 32

#ifdef __cplusplus
}
#endif
#endif
