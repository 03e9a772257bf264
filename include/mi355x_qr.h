/* mi355x_qr.h -- public C ABI of libmi355xqr.so: MI355X-native (gfx950) fp64 blocked Householder QR.
 *
 * The library is a drop-in for the host entry points of brian-kelley/CUDA-QR's qr.c built with
 * Scalar = double ("double* A, m, n -> Q, R"), plus a device-resident API for callers that keep the
 * matrix in HBM (bench, TSQR over several GPUs).  Plain pointers and sizes only; no HIP, C++ or torch
 * types cross this boundary.  All matrices are column-major with leading dimension = row count unless
 * an ld argument says otherwise (reference layout, qr.c:35,85).
 *
 * Every entry point cites the reference interface it replaces as file:line into the reference repo.
 */
#ifndef MI355X_QR_H
#define MI355X_QR_H
#include <stddef.h>
#ifdef __cplusplus
extern "C" {
#endif

/* ---------------------------------------------------------------------------------------------
 * 1. Drop-in symbols (same names, argument order and meaning as the reference, Scalar = double)
 * ------------------------------------------------------------------------------------------- */

/* replaces qr.c:47-53 getPanelDims.  The reference's sliding PR x PC window does not exist here: a
 * column panel of width nb (the library block size, default 128) is factored over its full height,
 * so rowPanels = 1 and colPanels = ceil(n/nb).  tau therefore holds rowPanels*colPanels*nb >= n
 * entries, which keeps the reference caller's tau[i*rowPanels + j] indexing (qr.c:483-490) in bounds. */
void getPanelDims(int m, int n, int* rowPanels, int* colPanels);

/* replaces qr.c:55-313 mmqr (and its GPU twin qr.cu:475-553).  In-place Householder QR of the m x n
 * (m >= n) host matrix `mat`: on return the upper triangle holds R and the sub-diagonal of column j
 * holds the tail of reflector v_j (v_j(j) = 1 implicit).  *tau is malloc'ed here (caller frees, as
 * qr.c:61,521) with colPanels*nb entries: tau[j] = Householder scalar of column j, zero padded.
 * tau is an opaque array consumed by this library's own explicitQR: the reference's
 * window-indexed layout (qr.c:300-304) depends on its compile-time PR, PC and is not reproduced.
 * Convention: LAPACK dlarfg (H = I - tau v v^T, R diagonal = -sign(x0)*||x||, same as qr.c:144-158),
 * except that an exactly-zero column tail gives tau = 0 where the reference yields NaN (qr.c:152). */
void mmqr(double* mat, double** tau, int m, int n);

/* replaces qr.c:330-438 explicitQR.  From mmqr's output builds R (m x n: upper triangle of A, zero
 * below, qr.c:334-343) and the dense m x m orthogonal Q = H_0 H_1 ... H_{n-1} (the reference forms it
 * with an m^3 product per reflector, qr.c:415-429; here: blocked backward accumulation on MFMA tiles).
 * Q and R are caller-allocated (qr.c:492-493). */
void explicitQR(double* A, double* tau, double* Q, double* R, int m, int n);

/* replaces qr.c:443-459 dgemm: C (k x n) = A (k x m) * B (m x n), column-major. Runs on the GPU. */
void dgemm(double* A, double* B, double* C, int k, int m, int n);

/* replaces qr.c:316-324 identity: A = I(m). */
void identity(double* A, int m);

/* replaces qr.c:21-33 printMat (row-by-row debug print to stdout, "%9f "). */
void printMat(double* mat, int m, int n);

/* The reference returns nothing and exits/asserts on error (qr.c:465, qr.cu:467-471).  The void
 * shims above print a diagnostic to stderr and return; these variants return a status instead
 * (0 = ok, >0 = hipError_t, <0 = QR_E_*).  The library never calls exit(). */
int mmqr_status(double* mat, double** tau, int m, int n);
int explicitQR_status(double* A, double* tau, double* Q, double* R, int m, int n);
int dgemm_status(double* A, double* B, double* C, int k, int m, int n);

/* Float instantiation (the reference as committed has Scalar = float, qr.c:11): mmqr / explicitQR on float
 * arrays with the same layouts and ownership rules.  The arithmetic is fp64 on the device (inputs widened, outputs rounded
 * once), so the results are at least as accurate as a float build of the reference.  The reference's window-indexed tau
 * layout (qr.c:300-304) is not what these return (it describes the reflectors of its sliding-window algorithm and cannot be
 * derived from a different reflector set): mmqr_legacy_status below runs that algorithm for callers who need it. */
void mmqr_f32(float* mat, float** tau, int m, int n);
void explicitQR_f32(float* A, float* tau, float* Q, float* R, int m, int n);
int mmqr_f32_status(float* mat, float** tau, int m, int n);
int explicitQR_f32_status(float* A, float* tau, float* Q, float* R, int m, int n);

/* Legacy-layout shim: the reference's sliding-window MMQR ITSELF on the device, with its compile-time window
 * PR x PC (qr.c:12-13) as arguments, for callers that consume the raw factored form -- the reflector tails exactly where qr.c:242-248
 * leaves them and tau[(rowPanels * pcCount + prCount) * PC + i] (qr.c:300-304; read by the reference's main, qr.c:483-490).
 * mmqr_legacy_status = qr.c:55-313, explicitQR_legacy_status = qr.c:330-438, getPanelDims_legacy = qr.c:47-53 for a given window.
 * Results equal the reference's to rounding (sums in wave-reduction order; tests: 1e-12 of the matrix scale), NaN for a zero column
 * like the reference (qr.c:152).  Shapes: what the reference's loops assume -- n % PC == 0, (m - PR) % (PR - PC) == 0, m >= PR --
 * with PR <= 64 and PC in {2, 4, 8, 16}; anything else is QR_E_ARG.  Two launches per column panel: a compatibility path, not a fast one. */
void getPanelDims_legacy(int m, int n, int PR, int PC, int* rowPanels, int* colPanels);
int mmqr_legacy_status(double* mat, double** tau, int m, int n, int PR, int PC);
int explicitQR_legacy_status(double* A, double* tau, double* Q, double* R, int m, int n, int PR, int PC);

#define QR_E_ARG      (-101)  /* bad argument (null pointer, m < n, non-positive size) */
#define QR_E_ALLOC    (-102)  /* host allocation failed */
#define QR_E_NODEVICE (-103)  /* no HIP device visible: there is NO CPU fallback */
#define QR_E_INTERNAL (-104)
#define QR_E_STALL    (-105)  /* qr_plan_sync: a hand-off between the workgroups of a one-launch panel timed out; the factorisation is invalid */
#define QR_E_REFUSED  (-106)  /* qr_plan_sync in latch mode (qr_plan_set_guard_mode): a full-width tall panel was refused; result invalid */
const char* qr_strerror(int status);

/* Block sizes used by the drop-in entry points (outer compact-WY block nb: multiple of ib, <= 512, above 256 a multiple of 256;
 * leaf width ib <= 32).  Defaults 128 / 32; when nothing was set explicitly the outer block follows the shape: 256 where the two-stream
 * look-ahead schedule is used (n >= 2048, m n >= 8 M, not too tall) and from 8192 columns on, 64 for small square-ish problems
 * (n >= 512, m <= 3 n), 256 for taller ones of at most 8192 rows -- qr_default_block_size / getPanelDims(m, n, ..) report what that shape will really get;
 * env MI355XQR_NB / MI355XQR_IB override the defaults.
 * Threading: the library may be used from one host thread per GPU (each thread with its own current device and its own
 * plans; a plan belongs to one thread at a time).  Process-wide state (these defaults, per-device kernel attributes, the
 * plan cache of the host-pointer entry points) is guarded internally. */
int qr_set_block_size(int nb, int ib);
void qr_get_block_size(int* nb, int* ib);
/* the block sizes an m x n problem REALLY gets from mmqr / a plan created with nb = 0, ib = 0 (shape-dependent, see above):
 * mmqr's tau holds ceil(n / nb) * nb entries of this nb -- size buffers from here, not from qr_get_block_size */
int qr_default_block_size(int m, int n, int* nb, int* ib);

/* Thin QR for shapes whose m x m Q cannot exist (no reference counterpart: the
 * reference's explicitQR is m x m only).  A (m x n, host) is not modified; Q is m x n, R is n x n
 * (upper triangular, diag of any sign).  nshards > 1 factors `nshards` contiguous row blocks
 * independently and combines their R factors (TSQR on one device; the multi-GPU form of the same
 * steps is driven through the device API below with an RCCL all-gather between steps 1 and 2). */
int qr_thin(const double* A, int m, int n, double* Q, double* R, int nb, int nshards);

/* The same over `ngpu` REAL devices of this node ("qr_thin(..., int nb, int ngpu) ... owns its own threads /
 * RCCL communicator").  Device d (0 <= d < ngpu) factors the contiguous row block d of A on its own host thread; the R factors
 * travel in ONE ncclAllGather (librccl is dlopen()ed on the first call with ngpu > 1, never linked), every device factors
 * the stacked (ngpu*n) x n matrix redundantly and forms its rows of Q.  ngpu = 1 needs no communicator and gives exactly
 * qr_thin(..., nshards = 1).  Returns QR_E_ARG when ngpu exceeds the visible devices or a shard would have fewer than n rows.
 * Q is m x n, R is n x n (host memory).  No reference counterpart (the reference is single-device, qr.cu:711,737). */
int qr_thin_mgpu(const double* A, int m, int n, double* Q, double* R, int nb, int ngpu);

/* Device-resident TSQR step, one rank per GPU (one process or one host thread each), for callers whose row shard already lives
 * in HBM -- the per-GPU step of BASELINE configs C4 / C5.  No reference counterpart (single-device, qr.cu:711,737).
 *   rank 0:      qr_tsqr_unique_id(id)            128 bytes; carry them to every rank (MPI_Bcast, a key-value store, a file)
 *   every rank:  hipSetDevice(local gpu); qr_tsqr_plan_create(&tp, id, nranks, rank, m_local, n, nb)     (collective)
 *   per matrix:  qr_tsqr_factor_dev(tp, dA_shard, lda, dR)      local QR -> ONE ncclAllGather of the n x n R factors ->
 *                                                               redundant QR of the stacked (nranks n) x n matrix -> dR (n x n, ld n)
 *                qr_tsqr_formq_dev(tp, dA_shard, lda, dQ, ldq)  optional: this rank's m_local x n rows of the thin Q
 * Stream-ordered: no stream is drained inside a step; in the default guard mode (qr_plan_set_guard_mode below) the host thread waits
 * once per full-width panel of a shard above 16384 rows for that panel's verdict word while its last pass still runs -- latch mode on
 * qr_tsqr_local_plan(tp) removes even that.  The stacked factorisation of one call overlaps the local factorisation
 * of the next (independent matrices).  qr_tsqr_sync() before results are read on another stream.  librccl.so is dlopen()ed
 * on first use.  nranks = 1 needs no id (NULL) and no communicator.  qr_tsqr_plan_create_comm takes an ncclComm_t the caller
 * already owns (as void*; NULL = the caller exchanges the factors itself: qr_tsqr_local_dev, copy through
 * qr_tsqr_exchange_buffers -- send: this rank's n*n doubles, recv: nranks*n*n in rank order --, qr_tsqr_stacked_dev). */
typedef struct qr_tsqr_plan qr_tsqr_plan;
#define QR_TSQR_UNIQUE_ID_BYTES 128
int qr_tsqr_unique_id(void* id128);
int qr_tsqr_plan_create(qr_tsqr_plan** tp, const void* id128, int nranks, int rank, int m_local, int n, int nb);
int qr_tsqr_plan_create_comm(qr_tsqr_plan** tp, void* nccl_comm, int nranks, int rank, int m_local, int n, int nb);
int qr_tsqr_plan_destroy(qr_tsqr_plan* tp);
int qr_tsqr_factor_dev(qr_tsqr_plan* tp, double* dA_shard, int lda, double* dR);
int qr_tsqr_formq_dev(qr_tsqr_plan* tp, const double* dA_shard, int lda, double* dQ, int ldq);
int qr_tsqr_local_dev(qr_tsqr_plan* tp, double* dA_shard, int lda);
int qr_tsqr_exchange_buffers(qr_tsqr_plan* tp, double** send, double** recv);
int qr_tsqr_stacked_dev(qr_tsqr_plan* tp, double* dR);
/* How the exchange is scheduled.  For single-stream (tall-skinny) shapes whose n is a multiple of the block size, qr_tsqr_factor_dev
 * goes block column by block column ("panel-pipelined", qr_tsqr_is_pipelined() = 1): block column k of a rank's R is final as soon as
 * local panel k is factored, so it is gathered (n / nb small ncclAllGathers instead of one) and the stacked matrix is factored
 * left-looking on a second stream WHILE the local factorisation continues -- only the last block column's share of the stacked QR
 * (~0.4 ms of 1.2 at 8 x 512 columns) is added to the latency of the step.  MI355XQR_TSQR_PIPE=0: one collective after the local QR.
 * qr_tsqr_factor_virtual_dev: the same schedule over P plans of ONE device from one thread (plans from
 * qr_tsqr_plan_create_comm(.., NULL, ..); the gather is device copies) -- tests and single-device bring-up.
 * qr_tsqr_factor_selfgather_dev: one rank's complete step with its own factor copied into every rank slot: the launches, streams and
 * events of a real rank minus the network (latency measurements on one GPU). */
int qr_tsqr_is_pipelined(qr_tsqr_plan* tp);
/* The schedule chosen by the caller instead: mode 0 = one collective after the local factorisation, 1 = panel-pipelined (QR_E_ARG when the
 * plan's shape cannot run it), 2 = back to the library's rule.  COLLECTIVE in the sense that every rank must make the same call between the
 * same two factorisations (the two forms issue different collectives); drains the plan.  bench.py --gpus N uses it to time both forms. */
int qr_tsqr_set_schedule(qr_tsqr_plan* tp, int mode);
/* The exchange of the last pipelined qr_tsqr_factor_dev, from events on the stacked plan's stream (drains the plan's streams):
 * out5[0] = sum over the block columns of [stacked stream past its wait for the local panel -> gather done] in ms, out5[1] = the longest
 * of them, out5[2] = the whole call, out5[3] = 1 pipelined / 0 one collective, out5[4] = 1 when the ranks fell back together.
 * With MI355XQR_TSQR_PIPE unset (or "auto") every rank contributes {out5[0], out5[2]} of its second call to one more all-gather before
 * its third and all apply the same rule: slowest rank's gathers > half of the fastest rank's step -> one collective from then on
 * (MI355XQR_TSQR_PIPE=1 keeps the pipelined form, =0 never uses it). */
int qr_tsqr_gather_stats(qr_tsqr_plan* tp, double* out5);
int qr_tsqr_factor_virtual_dev(qr_tsqr_plan** tps, int nranks, double** dA_shards, int lda, double** dR);
int qr_tsqr_factor_selfgather_dev(qr_tsqr_plan* tp, double* dA_shard, int lda, double* dR);
int qr_tsqr_sync(qr_tsqr_plan* tp);
void* qr_tsqr_stream(qr_tsqr_plan* tp);                 /* hipStream_t of the local step and the collective */
int qr_tsqr_comm_ranks(qr_tsqr_plan* tp, int* nranks);  /* ranks as RCCL itself counts them */

/* The host-pointer entry points (mmqr, explicitQR) keep their last few plans and device buffers, keyed by (device, m, n, nb),
 * so that repeated calls on same-sized matrices -- what the reference's harness does, qr.cu:776-789 -- do not pay ~10 ms of
 * allocation and stream creation each time.  This frees them (idle ones); MI355XQR_PLAN_CACHE=0 disables the cache. */
int qr_release_cached_plans(void);

/* ---------------------------------------------------------------------------------------------
 * 2. Device-resident API (all d* pointers are device memory of the current HIP device)
 * ------------------------------------------------------------------------------------------- */
typedef struct qr_plan qr_plan;

/* Workspace + streams for factoring matrices up to m x n with outer block nb and leaf width ib
 * (nb = 0 / ib = 0 select the library defaults).  A plan whose height is not a multiple of 16 (from 512 rows on, up to 4 GiB of
 * matrix) also holds an m x n buffer: qr_geqrf_dev at the plan's full height factors a copy of the caller's matrix with zero rows
 * appended -- same R, tau and V -- because an odd height or leading dimension keeps every kernel off its aligned path (x3). */
int qr_plan_create(qr_plan** plan, int m, int n, int nb, int ib);
int qr_plan_destroy(qr_plan* plan);

/* In-place blocked Householder QR of dA (m x n, lda), m <= plan m, n <= plan n; dtau: n doubles.
 * Device-side counterpart of mmqr (qr.c:55) with input already resident in HBM; asynchronous on the
 * plan's stream -- call qr_plan_sync before reading results on another stream.  The plan's streams are
 * non-blocking streams of their own: work queued on ANOTHER stream that writes a buffer handed to the plan
 * (a memset, a fill, a framework's allocation-time zeroing) must have completed -- synchronise that stream,
 * or make qr_plan_stream() wait on an event -- before the call, or it may land on top of the plan's output. */
int qr_geqrf_dev(qr_plan* plan, double* dA, int m, int n, int lda, double* dtau);

/* dC (m x ccols, ldc) <- Q * dC where Q = H_0..H_{n-1} comes from qr_geqrf_dev's factors.
 * identity_start != 0 first sets dC = I(m, ccols) and skips the structurally-zero part, i.e. forms
 * the leading ccols columns of Q (ccols = n: thin Q; ccols = m: the reference's m x m Q, qr.c:330). */
int qr_applyq_dev(qr_plan* plan, const double* dA, int m, int n, int lda, const double* dtau, double* dC,
                  int ccols, int ldc, int identity_start);

/* dR (rrows x n, ldr) = upper triangle of the factored dA, zero elsewhere (qr.c:334-343). */
int qr_extract_r_dev(qr_plan* plan, const double* dA, int m, int n, int lda, double* dR, int rrows, int ldr);

/* C = beta*C + alpha*op(A)*B with op = none ('N') or transpose ('T'); MFMA f64 tiles. */
int qr_gemm_dev(qr_plan* plan, char transa, int M, int N, int K, double alpha, const double* dA, int lda,
                const double* dB, int ldb, double beta, double* dC, int ldc);

/* Synthetic input: uniform[0,1) by a counter-based hash of the global element index, so any row
 * shard (rows [row_off, row_off+rows) of a total_rows x cols matrix) of the same seed is the same
 * data regardless of how many GPUs hold it.  Same distribution as the reference's
 * generator (qr.c:468-474), which is serial glibc rand() and kept for the small CPU-parity cases. */
int qr_fill_uniform_dev(qr_plan* plan, double* dA, int lda, long long rows, int cols, long long row_off,
                        long long total_rows, unsigned long long seed);
double qr_uniform_at(unsigned long long seed, unsigned long long linear_index);

/* sums[0] = ||X - Y||_F^2, sums[1] = ||Y||_F^2 over an rows x cols block.  Y is dY (ldy) if non-null,
 * else the generator above (row_off/total_rows/seed), else (mode 1) the identity.  Synchronous. */
int qr_diffnorm_dev(qr_plan* plan, const double* dX, int ldx, const double* dY, int ldy, long long rows,
                    int cols, long long row_off, long long total_rows, unsigned long long seed, int mode,
                    double* sums);

/* Device memory helpers so that a caller in any host language can use the device API without binding HIP itself
 * (the buffers are ordinary hipMalloc memory of the current device; copies are synchronous). */
int qr_device_malloc(void** dptr, size_t bytes);
int qr_device_free(void* dptr);
int qr_copy_to_device(void* dst, const void* src, size_t bytes);
int qr_copy_to_host(void* dst, const void* src, size_t bytes);

/* the two qr_plans inside a TSQR plan (profiling, fills, norms on the same streams); stacked: NULL when nranks = 1 */
qr_plan* qr_tsqr_local_plan(qr_tsqr_plan* tp);
qr_plan* qr_tsqr_stacked_plan(qr_tsqr_plan* tp);

/* Waits for everything queued on the plan's streams and reads the status words the panel kernels left on the device: QR_E_STALL /
 * QR_E_REFUSED (above) report a factorisation issued since the last call that must not be used; both are cleared by the call. */
int qr_plan_sync(qr_plan* plan);
/* What happens when the device-side guard refuses a full-width tall panel (>= MI355XQR_CQR_MIN_ROWS rows x 128 columns: CholeskyQR2 at
 * panel width needs cond(panel) < ~1e7 and full rank; no reference counterpart, the reference factors column by column, qr.c:109-235):
 *   latch = 0 (default): the panel is handed to the Householder-guarded leaf chain, the result is as good as for any other input.  The
 *              host thread reads the verdict from a host word the deciding kernel writes while the panel's last pass still runs: the GPU
 *              does not idle, but qr_geqrf_dev waits for that word once per tall panel (it is not capturable into a graph);
 *   latch = 1: qr_geqrf_dev never waits for the device (fully stream-ordered).  A refused panel makes the factorisation INVALID (the
 *              matrix is overwritten with garbage from that panel on); qr_plan_sync / qr_tsqr_sync return QR_E_REFUSED and the caller
 *              factors a fresh copy with latch = 0.  For pipelines that own their inputs and check a residual anyway.
 * MI355XQR_GUARD=latch selects latch = 1 for every plan a caller of the DEVICE API creates; the host-pointer entry points (mmqr, explicitQR,
 * qr_thin, qr_thin_mgpu) block anyway and always use latch = 0 on their own plans.
 * The call drains the plan and returns qr_plan_sync's status: a refusal latched before the switch is reported here, not lost. */
int qr_plan_set_guard_mode(qr_plan* plan, int latch);
/* out4: full-width tall panels issued, of them refused, leaves of one-launch panels that took their Householder route, one-launch
 * panels whose hand-off stalled -- since the plan was created (refusals in latch mode and the last two are counted at qr_plan_sync) */
int qr_plan_route_stats(qr_plan* plan, long long* out4);
/* out2: refused full-width panels that were retried PRECONDITIONED (shifted CholeskyQR3: R0 = chol(A^T A + s I), the same three-pass pipeline
 * on A R0^-1; default guard mode only), and how many of those the guard then accepted -- the others (rank deficient, cond > ~1e10) went
 * to the Householder leaf chain.  No reference counterpart (the reference factors column by column, qr.c:109-235). */
int qr_plan_retry_stats(qr_plan* plan, long long* out2);
void* qr_plan_stream(qr_plan* plan);          /* the hipStream_t work is queued on */
int qr_plan_update_cus(qr_plan* plan);        /* compute units the wide trailing update runs on (its share of the CU partition) */

/* Per-kernel-class timing with HIP events recorded on the plan's stream inside the timed region.
 * class 0 = trailing update A2 -= V*W (gemm_nn), 1 = W = (V T)^T A2 (gemm_tn + slab reduce),
 * 2 = panel factorisation (leaf kernels + in-panel updates + T), 3 = V*T and misc. */
#define QR_PROF_CLASSES 4
typedef struct qr_profile {
    double ms[QR_PROF_CLASSES];      /* summed event-to-event time */
    double flops[QR_PROF_CLASSES];   /* algorithmic flops issued */
    double bytes[QR_PROF_CLASSES];   /* algorithmic HBM bytes (compulsory traffic) */
    long long launches[QR_PROF_CLASSES];
} qr_profile;
/* on = 0: off; 1: every class; 2 * mask: only the classes whose bit is set in mask (bit c = class c; bits 4, 5 = the look-ahead
 * update / the panel stream's share of a wide update).  A record costs two event packets on its stream: profile what you read. */
int qr_plan_set_profile(qr_plan* plan, int on);
int qr_plan_pause_profile(qr_plan* plan, int pause);       /* stop / resume recording, keeping the records made so far */
/* what the plan was built with: outer / leaf block size, 1 = two-stream look-ahead schedule (any pointer may be NULL) */
int qr_plan_info(qr_plan* plan, int* nb, int* ib, int* lookahead);
int qr_plan_get_profile(qr_plan* plan, qr_profile* out);   /* synchronises, sums, resets */
/* The individual records behind the sums, in issue order (call before qr_plan_get_profile): class as above, plus 4 = the
 * look-ahead update of the next panel's columns and 5 = the panel stream's share of a wide update (both summed into class 3
 * by qr_plan_get_profile); start / end in ms since the first record began.  Returns the number of records written (<= max). */
int qr_plan_get_profile_records(qr_plan* plan, int max, int* cls, double* t0_ms, double* t1_ms);

/* Device facts + micro-probes used by bench.py / DESIGN.md (measured, not datasheet). */
int qr_device_info(char* arch, int arch_len, int* compute_units, int* clock_khz, size_t* hbm_bytes);
/* out3[0] = sustained back-to-back v_mfma_f64_16x16x4_f64 TFLOP/s (best over 1/2/4 workgroups per CU),
 * out3[1] = in-kernel shader clock (GHz) during that run, out3[2] = f64 VALU FMA TFLOP/s */
int qr_probe_mfma_f64_tflops(double* out3);
int qr_probe_copy_gbps(double* gbps);

#ifdef __cplusplus
}
#endif
#endif
