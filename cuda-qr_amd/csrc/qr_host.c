/* qr_host.c -- C host layer of libmi355xqr.so.
 *
 * Plain C99 (the reference's host language, qr.c); it owns the algorithm schedule -- which panel,
 * which leaf, which contraction, on which stream -- and reaches the GPU only through the extern "C"
 * launch layer declared in qr_device.h.  No HIP headers here, no exit(), no CPU compute fallback:
 * without a HIP device every entry point fails with QR_E_NODEVICE.
 *
 * Algorithm (replaces the reference's sliding-window "MMQR", qr.c:55-313, on purpose -- see DESIGN.md):
 * right-looking blocked Householder QR with compact-WY accumulation.
 *   for each outer panel k (nb columns, full remaining height mk = m-k):
 *     for each leaf (ib <= 32 columns): leaf_panel  -> v, tau, R, leaf T        [qr.c:109-235]
 *                                       W_l = T_l^T V_l^T A_rest ; A_rest -= V_l W_l   (rest of panel)
 *     G = V^T V ; T = larft(G, leaf T's) ; VT = V T                             [qr.c:170-213]
 *     W = VT^T A2 ; A2 -= V W                                                   [qr.c:255-293]
 */
#define _POSIX_C_SOURCE 200809L      /* strtok_r, pthread under -std=c99 */
#include <pthread.h>
#include <sched.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

#include "../../include/mi355x_qr.h"
#include "qr_device.h"

#define CHECK(x) do { int rc_ = (x); if (rc_) return rc_; } while (0)

#define QR_MAX_PAIRS 4
#define QR_MAX_NB 512           /* outer block (the K of the wide update) */
#define QR_HALF 256             /* outer blocks wider than this are factored half by half (two-level panel, factor_panel) */
#define QR_DEFAULT_SPLIT "64"
/* large square problems are update-bound for most of their flops: 32 CUs for the panel chain, 224 for the wide update, and the
 * look-ahead update N(s) on the update stream in the chain-bound phase (it would crawl on 32 CUs).  Measured at 16384^2: 125.6 ms
 * against 131.8 with 64 / 192 (the update GEMMs run at 50 instead of 43 TFLOP/s per launch); at 8192^2 and below the chain
 * dominated and 64 CUs were faster (round 6, with the one-launch panel on 128-row workgroups: 8192^2 23.5 -> 23.0 ms with 32, 6144^2
 * equal, 16384 x 8192 46.3 against 47.7 and 8192 x 4096 9.4 against 10.1 still want 64: profiles/r06_cu_split_by_shape.txt).  A mask bit i is compute unit i/8 of XCC i%8 (profiles/r02_probe_cumask.txt): a contiguous
 * range of 32 bits = 4 CUs of every XCD, so both streams stay balanced over the XCDs; a 48 / 208 split was measured 15 % slower. */
#define QR_DEFAULT_SPLIT_BIG "32"
struct qr_plan {
    int m, n, nb, ib, ldv, ldt;
    int use_graph;              /* 1: qr_geqrf_dev is captured into a hipGraph once per argument set and replayed */
    void* graph_exec;
    double *g_dA, *g_dtau; int g_m, g_n, g_lda;
    int lookahead;              /* 1: panel k+1 is factored on `stream` while `stream_u` updates the rest */
    void* stream;               /* stream the next launch of the critical path goes to: s_main outside qr_geqrf_dev,
                                 * the current phase's panel stream inside it */
    void* stream_u;             /* wide trailing-update stream of the current phase */
    void* s_main;               /* the plan's public stream: all compute units */
    /* CU partition phases: while more than until[i] of the columns remain, the panel chain owns cus[i] compute
     * units (s_pair[i][0]) and the wide update the rest (s_pair[i][1]).  npairs = 0: no partition. */
    int npairs, pair_cur;
    void* s_pair[QR_MAX_PAIRS][2];
    int pair_shared_u[QR_MAX_PAIRS];   /* phase i's update stream is phase i-1's (not owned: never destroyed / synchronised twice) */
    double pair_until[QR_MAX_PAIRS];
    void* ev_hop[2];
    void* ev_extra[2];          /* panel-stream share of wide update s finished */
    double *We, *Ye;            /* its W buffer and raw V^T A2 */
    double *Ye2;                /* raw V^T A2 of a wide update that applies T to the small product (tall-skinny plans) */
    int m_user, n_user;         /* the shape the plan was asked for; m x n (>= it) is what it factors: see padA */
    double* pad_tau;            /* n scalars of the padded factorisation (the caller's dtau has n_user) */
    int pad_failed;             /* the lazy allocation of padA failed once: do not try again */
    double* padA;               /* (m x n, ld m) heights that are not multiples of 16: qr_geqrf_dev factors a copy with zero rows appended */
    double *Yn;                 /* raw V^T A_next of the look-ahead update */
    double bal_rp, bal_ru, bal_tc0, bal_tc1;   /* load-balance model (TFLOP/s, ms); bal_rp = 0: off */
    double bal_tail_tc;         /* chain time where the next panel is ONE launch (see chain_ms); 0: the linear model everywhere */
    double bal_tc0_base, bal_tc1_base; int bal_auto;   /* bal_auto: no MI355XQR_BALANCE override -- rates follow the phase's partition */
    void* ev_half[2];           /* W_a(s): the wide update has finished the columns of panel s+1 that N(s) left out (its second half) */
    void* ev_next[2];           /* look-ahead update N(s) of the next panel's columns finished (when it runs on the update stream) */
    int next_on_update;         /* 1: N(s) runs on the update stream's CUs, ahead of W(s); 0: on the panel stream; 2: on the panel
                                 * stream while the factorisation is update-bound, on the update stream once it is chain-bound */
    void* ev_panel[2];          /* panel set s ready (V, T, VT) */
    void* t_wait;               /* see apply_small_t */
    void* ev_v[2];              /* V of panel set s complete (its T merge still running): the long-K product of N(s) may start */
    void* v_ready;              /* event factor_panel records before the T merge of a one-level panel (NULL: none) */
    int defer_hint, t_deferred; /* defer_hint (set by the caller for ONE factor_panel call): behind a one-launch panel leave the Gram matrix and the
                                 * T merge to the caller (deferred_t_merge, on the update stream: 224 compute units instead of the panel stream's 32);
                                 * t_deferred: factor_panel did so */
    void* ev_wide[2];           /* wide update that read panel set s finished */
    double *Vw, *VT, *T;        /* current panel set (aliases of set[cur]) */
    double *Vw2[2], *VT2[2], *T2[2];
    int vt_formed[2];           /* VT2[e] = Vw2[e] * T2[e] of the panel NOW in set e exists (cleared when a panel is factored into the set,
                                 * set where the V*T product is issued): a slice of the wide update that needs it forms it on demand */
    double *W, *Wn, *Tt, *G, *X, *slabs, *slabs_u, *panel_ws;
    double* slabs_ep;           /* split-K slabs of the leaf's early product: written while the reconstruction still reads p->slabs */
    size_t slab_ep_cap;
    int panel_tsqr;             /* 1: Householder-TSQR leaf alone (MI355XQR_PANEL=tsqr); 3: CholeskyQR2 + Householder reconstruction, guarded by (1) */
    double* chol_ws;
    double* cq_ws; int* cq_status;   /* small-factor workspace and guard words of the full-width tall panel (qr_panel_cqr.hip); NULL: not used */
    /* "parked" full-width panels (round 5): on single-stream plans whose next update applies T to the small product (tall shapes) the panel's
     * V is written once, into the caller's array -- top block included, as the unit lower triangle -- and the update reads it from there;
     * R of the top block waits in the workspace and is put back right after that update (cq_unpark).  park_hint: set by the caller of
     * factor_panel for ONE call (it knows what follows the panel); cq_parked: the panel in cq_park_top is in that state now */
    int park_hint, cq_parked, cq_park_lda, cq_park_w;
    double* cq_park_top;
    unsigned *cq_hword, *cq_hword_dev;   /* host word (mapped into the device) that receives a tall panel's verdict as soon as it exists */
    unsigned cq_seq;            /* sequence number of the last tall panel issued */
    int guard_latch;            /* 0: a refused tall panel is handed to the leaf chain (the host reads the verdict while the panel's last pass
                                 * runs); 1: nothing is read inside qr_geqrf_dev, a refusal is reported by qr_plan_sync (QR_E_REFUSED) */
    int cq_dirty, pf_dirty;     /* tall panels in LATCH mode / one-launch panels have been issued since the status words were last read (a tall
                                 * panel in poll mode has its verdict read at once: nothing is left pending) */
    int fused_off;              /* 1: never the one-launch panel (set by the host-pointer entry points after a stalled hand-off, QR_E_STALL) */
    long long n_cqr, n_cqr_refused, n_pf_leaf_fallback, n_pf_stall;   /* qr_plan_route_stats */
    long long n_cqr_retried, n_cqr_retry_ok;                          /* qr_plan_retry_stats: refused panels retried preconditioned / accepted then */
    double* pf_ws;              /* exchange workspace of the one-launch panel (qr_panel_fused.hip); NULL: not used */
    unsigned pf_epoch;          /* its epoch counter: the workspace's epoch words never exceed it */
    int* pf_status;             /* device: [0] leaves that took the Householder route inside a one-launch panel, [1] a wait timed out */
    size_t slab_cap, w_cap;
    /* profiling */
    int prof_on, prof_mask, prof_count, prof_cap, prof_open, prof_paused;
    void** prof_ev;             /* 2 events per record */
    void* prof_stream;          /* stream of the open record */
    int* prof_cls;
    double *prof_flops, *prof_bytes;
};

/* ---------------------------------------------------------------------------------------------- */
/* Process-wide state, all of it behind g_lock: the default block sizes and which devices have had their kernel
 * attributes set.  Everything else lives in a qr_plan, which belongs to one host thread at a time ("callable
 * from one host thread per GPU"). */
#define QR_MAX_DEVICES 64
static pthread_mutex_t g_lock = PTHREAD_MUTEX_INITIALIZER;
static int g_nb = 0, g_ib = 0;
static int g_nb_explicit = 0;      /* MI355XQR_NB or qr_set_block_size chose nb: no automatic 256 for large square problems */
static unsigned char g_dev_inited[QR_MAX_DEVICES];

/* Tuning knobs of the schedule (MI355XQR_* environment variables): read ONCE per process under pthread_once --
 * qr_thin_mgpu drives qr_geqrf_dev from one host thread per device, so nothing here may be a lazily written function static. */
/* constants that used to be knobs: swept in rounds 2 and 3 (profiles/r02_session2_ab_measurements.txt, r03_split_sweeps.txt), flat or
 * worse away from these values */
#define QR_FUSE_NN_MIN_ROWS 20000   /* the fused in-panel update + next leaf's Gram launch: tall leaves only */
#define QR_TFOLD_MAX 128            /* T^T folded into the slab reduce up to this panel width */
#define QR_EARLY_W1 4096            /* columns of the first slice of an early look-ahead step's wide update */
#define QR_TAIL_TC_MS 0.0           /* chain_ms: flat chain time (ms per 256 columns) behind a one-launch panel; 0 = linear model only */

typedef struct qr_knobs {
    int fuse_nn;                                            /* MI355XQR_FUSE_NN */
    int split_t;                                            /* MI355XQR_SPLIT_T */
    int early_next;                                         /* MI355XQR_EARLY_NEXT */
    int plan_cache;                                         /* MI355XQR_PLAN_CACHE */
    int early_product;                                      /* MI355XQR_EP: the leaf's in-panel product in the launch of its reconstruction */
    int fused_panel;                                        /* MI355XQR_FUSED_PANEL: a whole outer panel (<= 16384 rows) in one launch */
    int fused_min_rows;                                     /* MI355XQR_FUSED_MIN_ROWS: ... from this many rows on */
    int tsqr_halves_rows;                                   /* MI355XQR_TSQR_HALVES (lab): stacked matrices shorter than this get their last block column in two halves */
    int cqr_min_rows;                                       /* MI355XQR_CQR_MIN_ROWS: 128-column panels of at least this many rows at full width (0 = never).  Round 4:
                                                             * 196608 (its one-workgroup kernels cost 274 us per panel); round 5 (181 us, profiles/r05_cqr_crossover.txt): everything
                                                             * the one-launch panel cannot take (> 8192 rows then; end of round 6: > 16384 rows, where it wins again -- 16384 x 512
                                                             * 1.24 -> 1.14 ms, 12288 x 2048 5.89 -> 5.66 single-stream) on single-stream plans -- 32768 x 256 0.73 against 0.82 ms,
                                                             * 65536 x 256 0.93 against 0.99, 131072 x 256 1.28 against 1.35; below 8192 rows the one-launch panel wins (4096 x 512
                                                             * 1.04 against 1.20), and on the CU-masked panel stream of a look-ahead plan the leaf chain does (16384 x 2048 8.0
                                                             * against 8.7): there the round-4 threshold stays */
    int tall_nt;                                            /* MI355XQR_TALL_NT: the update of a tall block through gemm_nt (W transposed first) */
    int cqr_park;                                           /* MI355XQR_CQR_PARK: full-width panels of tall single-stream plans write V once (into A) */
    int cqr_min_rows_la;                                    /* MI355XQR_CQR_MIN_ROWS_LA (lab): the same threshold on look-ahead plans (QR_CQR_MIN_ROWS_LOOKAHEAD) */
    int cqr_retry;                                          /* MI355XQR_CQR_RETRY (lab): a refused full-width panel is retried preconditioned (shifted CholeskyQR3) before the leaf chain */
    int trsm_next;                                          /* MI355XQR_TRSM (lab): the look-ahead update behind a deferred merge without the merged T (qrd_trsm_gt) */
    int defer_t;                                            /* MI355XQR_DEFER_T (lab): a one-launch panel's Gram matrix + T merge on the update stream in the chain-bound phase */
    int fused_gram;                                         /* MI355XQR_FUSED_GRAM (lab): widest panel whose Gram blocks come out of the one-launch panel itself */
} qr_knobs;
static qr_knobs g_knobs;
static pthread_once_t g_knobs_once = PTHREAD_ONCE_INIT;

static int env_int(const char* name, int dflt)
{
    const char* e = getenv(name);
    return e ? atoi(e) : dflt;
}

/* Two classes of environment variables.  The twelve PUBLIC ones (NB, IB, PANEL, GUARD, LOOKAHEAD, SPLIT,
 * CQR_MIN_ROWS, FUSED_MIN_ROWS, TSQR_PIPE, TSQR_RESERVE_CUS, PLAN_CACHE, ROCTX) are read in every build.  The measurement knobs -- settled
 * A/Bs and overrides of the schedule's own choices -- exist only in the LAB build (`make lab`: libmi355xqr_lab.so, -DQR_LAB), which the
 * scripts under devtools/ and the schedule-variant tests load; in the product library they are the constants below and no stray
 * variable can change what a factorisation does. */
#ifdef QR_LAB
#define lab_getenv(name) getenv(name)
#else
#define lab_getenv(name) ((const char*) NULL)
#endif
static int lab_env_int(const char* name, int dflt)
{
    const char* e = lab_getenv(name);
    (void) name;
    return e ? atoi(e) : dflt;
}

static void knobs_init(void)
{
    qr_knobs* k = &g_knobs;
    k->fuse_nn = lab_env_int("MI355XQR_FUSE_NN", 1) != 0;
    k->split_t = lab_env_int("MI355XQR_SPLIT_T", 1) != 0;
    k->early_next = lab_env_int("MI355XQR_EARLY_NEXT", 1) != 0;
    k->plan_cache = env_int("MI355XQR_PLAN_CACHE", 1) != 0;
    k->early_product = lab_env_int("MI355XQR_EP", 1) != 0;
    k->fused_panel = lab_env_int("MI355XQR_FUSED_PANEL", 1) != 0;
    k->fused_min_rows = env_int("MI355XQR_FUSED_MIN_ROWS", 256);      /* round 4-5: 3072 (below, the launch chain was as fast).  Round 6, with 128-row workgroups: every
                                                                       * height gains -- 16384^2 118.1 -> 116.4 ms, 8192^2 25.3 -> 24.4, 4096^2 (nb 64) 11.2 -> 10.6, 2048^2 4.55 -> 4.08,
                                                                       * 1024^2 2.31 -> 2.04; 128 and 32 measure like 256 (profiles/r06_fused_min_rows.txt) */
    k->tsqr_halves_rows = lab_env_int("MI355XQR_TSQR_HALVES", 0);     /* round 5: 3072 (short stacked matrices went leaf by leaf); with one-launch stacked panels from 256 rows on the
                                                                       * split costs more than it hides at every size: C4 rank of 4 / of 2 1.12 / 1.44 -> 1.07 / 1.37 ms (profiles/r06_tsqr_rank_step_latency.txt) */
    k->fused_gram = lab_env_int("MI355XQR_FUSED_GRAM", 128);   /* widest panel whose Gram blocks V_prev^T V_l come out of the one-launch panel itself (its in-panel product
                                                                * takes the columns of V_prev along) instead of one launch pair behind it.  At 256 columns the extra product
                                                                * columns cost what the launch pair does (profiles/r04_fused_ab.txt); at 64-128 they are a tile or six */
    k->cqr_min_rows = env_int("MI355XQR_CQR_MIN_ROWS", 16385);          /* = everything the one-launch panel cannot take, see the struct comment */
    k->cqr_min_rows_la = lab_env_int("MI355XQR_CQR_MIN_ROWS_LA", 196608);   /* = QR_CQR_MIN_ROWS_LOOKAHEAD: see plan_cqr_min_rows */
    k->tall_nt = lab_env_int("MI355XQR_TALL_NT", 1) != 0;
    k->cqr_park = lab_env_int("MI355XQR_CQR_PARK", 1) != 0;
    k->defer_t = lab_env_int("MI355XQR_DEFER_T", 1) != 0;
    k->cqr_retry = lab_env_int("MI355XQR_CQR_RETRY", 1) != 0;
    k->trsm_next = lab_env_int("MI355XQR_TRSM", 1) != 0;
}

static const qr_knobs* knobs(void)
{
    pthread_once(&g_knobs_once, knobs_init);
    return &g_knobs;
}

static void defaults_from_env_locked(void)
{
    if (g_nb == 0) {
        const char* e = getenv("MI355XQR_NB");
        g_nb = e ? atoi(e) : 128;
        g_nb_explicit = e != NULL;
        e = getenv("MI355XQR_IB");
        g_ib = e ? atoi(e) : 32;
        if (g_ib < 1 || g_ib > QRD_LEAFW) g_ib = 32;
        if (g_nb < g_ib || g_nb > QR_MAX_NB || g_nb % g_ib || (g_nb > QR_HALF && g_nb % QR_HALF)) g_nb = 128;
    }
}

/* Does the look-ahead schedule (two streams, CU partition) pay for an m x n problem at block size nb?  Measured rule (round 6, after the
 * one-launch panel moved to 128-row workgroups: profiles/r06_block_size_lookahead_rule.txt, 40 shapes x 3 block sizes x on / off):
 *  - there has to be a wide update worth a stream of its own: n >= 2048 and m n >= 8 M (3072^2 6.31 -> 5.95 ms, 4096 x 2048 3.98 -> 3.80;
 *    2560^2 and below: the single-stream schedule at nb = 64 wins), never for m >= 16 n (panel and update both HBM-bound there);
 *  - the panel stream has 32 or 64 compute units: a panel the one-launch kernel takes (<= 16384 rows; on 32 CUs <= 7936) fits them, a
 *    taller one is a chain of launches that wants the whole chip, and the overlap only pays for it when the update is long enough --
 *    m <= n^2 / 800 at nb = 256 (24576 x 4096 25.4 -> 24.1 but 23.3 single-stream at nb = 128; 65536 x 8192 168.2 -> 164.3), m <= 2 n at
 *    nb = 128 (32768 x 8192: 90.1 single-stream against 94.3).  (The one-launch panel stopped at 8192 rows when this rule was first
 *    measured, and so did its first clause; with 16384: 12288 x 2048 6.43 single-stream -> 5.25 two-stream, 16384 x 2048 6.95 -> 5.89,
 *    12288 x 4096 14.8 -> 12.45, 16384 x 4096 18.1 -> 15.1: profiles/r06_panel_fused_16384_rows.txt);
 *  - narrow blocks (nb < 128): the round-2 rule, 18 M elements (4096^2 at nb = 64: 9.50 single-stream against 9.67). */
static int lookahead_pays(long long m, long long n, int nb)
{
    if (n < 2048 || m >= 16 * n) return 0;
    if (nb < 128) return m * n >= 18000000LL;
    if (m * n < 8000000LL) return 0;
    if (m <= 16384) return 1;
    return nb >= 256 ? 800 * m <= n * n : m <= 2 * n;
}

/* the block sizes a plan for an m x n problem gets when the caller passes nb = 0 / ib = 0 */
static void default_blocks(int m, int n, int* nb, int* ib)
{
    pthread_mutex_lock(&g_lock);
    defaults_from_env_locked();
    int b = g_nb;
    /* (same measurement as lookahead_pays.)  Wherever the look-ahead schedule pays it pays most at nb = 256: K = 256 lifts the update
     * GEMMs and halves the number of panel tails; so do problems of 8192 columns and more whatever their height.  Small square-ish
     * problems run single-stream and are all panel: 64-column panels are two leaves whose T is merged inside the panel's one launch,
     * and the update GEMMs are too small for K to matter (2048^2: 3.65 ms at nb 64, 3.91 at 128, 4.04 at 256; 1024^2 1.83 / 1.98 / 2.08;
     * 2048 x 1024 1.67 / 1.82 / 1.82; from m = 4 n on 128 is as good or better).  Tall shapes keep 128 (16384 x 2048: 6.95 against 7.78). */
    if (!g_nb_explicit) {
        /* (two-stream shapes of more than 8192 rows and fewer than 3072 columns: 128 is 2-4 % ahead -- 12288 x 2048 5.25 against 5.40 ms) */
        if (256 % g_ib == 0 && (n >= 8192 || lookahead_pays(m, n, 256))) b = (n < 3072 && m > 8192 && lookahead_pays(m, n, 128)) ? g_nb : 256;
        else if (64 % g_ib == 0 && n >= 512 && (long long) m <= 3LL * n) b = 64;
        /* taller than that but still one-launch panels all the way (m <= 8192): fewer, wider panels -- 8192 x 512 0.99 -> 0.95 ms,
         * 4096 x 512 0.87 -> 0.83, 8192 x 1024 2.20 -> 2.10; beyond 8192 rows the 128-column full-width panel route wants 128
         * (65536 x 512: 1.89 against 2.59 at 256 and 2.54 at 64) */
        else if (256 % g_ib == 0 && n >= 512 && m <= 8192) b = 256;
    }
    if (nb) *nb = b;
    if (ib) *ib = g_ib;
    pthread_mutex_unlock(&g_lock);
}

int qr_set_block_size(int nb, int ib)
{
    if (ib < 1 || ib > QRD_LEAFW || nb < ib || nb > QR_MAX_NB || nb % ib || (nb > QR_HALF && nb % QR_HALF)) return QR_E_ARG;
    pthread_mutex_lock(&g_lock);
    g_nb = nb; g_ib = ib; g_nb_explicit = 1;
    pthread_mutex_unlock(&g_lock);
    return 0;
}

void qr_get_block_size(int* nb, int* ib)
{
    pthread_mutex_lock(&g_lock);
    defaults_from_env_locked();
    if (nb) *nb = g_nb;
    if (ib) *ib = g_ib;
    pthread_mutex_unlock(&g_lock);
}

/* the block sizes mmqr / a plan created with nb = 0, ib = 0 really use for an m x n problem (getPanelDims reports the grid of these) */
int qr_default_block_size(int m, int n, int* nb, int* ib)
{
    if (m < 1 || n < 1 || m < n) return QR_E_ARG;
    default_blocks(m, n, nb, ib);
    return 0;
}

/* kernel attributes (dynamic-LDS caps) are per device: initialise the CURRENT device of the calling thread once */
static int ensure_device(void)
{
    int n = 0, dev = 0;
    if (qrd_device_count(&n) != 0 || n < 1) return QR_E_NODEVICE;
    CHECK(qrd_get_device(&dev));
    if (dev < 0 || dev >= QR_MAX_DEVICES) return QR_E_INTERNAL;
    pthread_mutex_lock(&g_lock);
    int rc = 0;
    if (!g_dev_inited[dev]) {
        rc = qrd_init();
        if (!rc) g_dev_inited[dev] = 1;
    }
    pthread_mutex_unlock(&g_lock);
    return rc;
}

const char* qr_strerror(int status)
{
    switch (status) {
    case 0: return "success";
    case QR_E_ARG: return "invalid argument";
    case QR_E_ALLOC: return "host allocation failed";
    case QR_E_NODEVICE: return "no HIP device (this library has no CPU fallback)";
    case QR_E_INTERNAL: return "internal error";
    case QR_E_STALL: return "a hand-off inside a one-launch panel timed out (its workgroups were not co-resident?): the factorisation is invalid";
    case QR_E_REFUSED: return "latch mode: the guard refused a full-width tall panel (ill-conditioned or rank-deficient): the factorisation is invalid";
    case QRD_E_NORCCL: return "librccl.so could not be loaded (multi-GPU entry points need RCCL)";
    default:
        if (status > 0) return qrd_error_string(status);
        if (status <= QRD_E_RCCL) return qrd_rccl_error_string(QRD_E_RCCL - status);
        return "kernel launch argument error";
    }
}

static int imin(int a, int b) { return a < b ? a : b; }

static inline void cpu_relax(void)
{
#if defined(__x86_64__) || defined(__i386__)
    __builtin_ia32_pause();
#elif defined(__aarch64__)
    __asm__ __volatile__("yield");
#endif
}

/* rows from which a 128-column panel of this plan takes the full-width route (knob comment above).  On look-ahead plans: round 4's
 * threshold.  Re-measured in round 6 (profiles/NOTES.md, r6_run27.sh): at nb 128 the full-width route on the 32 / 64-CU panel stream is
 * 25-30 % faster per tall panel than the leaf chain (16384 x 8192 51.3 -> 48.5 ms, 16384^2 128.3 -> 127.2) -- but those shapes run at
 * nb 256 (45.9 / 114-116 ms), whose 256-column panels the 128-column route does not take; factoring them as two 128-column halves is
 * worth an estimated 0.7 ms at C3 (the update-bound phase returns 12 % of what a panel saves) and was not built. */
#define QR_CQR_MIN_ROWS_LOOKAHEAD 196608
static int plan_cqr_min_rows(const qr_plan* p)
{
    const int r = knobs()->cqr_min_rows;
    const int la = knobs()->cqr_min_rows_la;
    return (p->lookahead && r > 0 && r < la) ? la : r;
}

/* ---------------------------------------------------------------------------------------------- */
/* tsqr_local: the local factorisation of a multi-GPU TSQR plan.  Every choice that shapes the EXCHANGE (block size, look-ahead) must
 * then come out the same on every rank, whatever its shard height: ranks of unequal height (m % ngpu != 0) used to straddle the
 * thresholds below and disagree on the number and size of the collectives.  Such plans take the block size of a very tall m x n
 * problem and the single-stream schedule, always. */
static int plan_create_impl(qr_plan** out, int m, int n, int nb, int ib, int tsqr_local);

int qr_plan_create(qr_plan** out, int m, int n, int nb, int ib) { return plan_create_impl(out, m, n, nb, ib, 0); }

static int plan_create_impl(qr_plan** out, int m, int n, int nb, int ib, int tsqr_local)
{
    if (!out || m < 1 || n < 1 || m < n) return QR_E_ARG;
    CHECK(ensure_device());
    /* Heights that are not multiples of 16: the plan is built for the next multiple, and qr_geqrf_dev on the full height factors a copy
     * of the caller's matrix with zero rows appended (R, tau and the first m rows of V are the caller's matrix's -- see mmqr_status;
     * 5001^2 through the caller's own buffer: 34 ms, 5000^2: 11).  Costs one more m x n buffer; not for the local plans of a multi-GPU
     * step (their shards are the caller's to size) and not above 4 GiB of matrix. */
    const int m_user = m, n_user = n;
    /* ... and widths that are not whole 32-column leaves get columns appended: unit vectors e_{m_user + j}, in zero rows appended for them.
     * Column j of R, V and tau depends on columns 0 .. j only (Cholesky, the triangular solves and the reconstruction are all
     * column-recursive), so the caller's n columns come out as without them -- and the appended ones stay exactly orthogonal to them
     * (every reflector has zeros in their rows) -- while the last panel, ragged otherwise (leaf by leaf, its updates on the generic
     * kernels), takes the whole-leaf routes (262144 x 500: 5.8 ms against 4.6 for 262144 x 512; profiles/r06_odd_sizes.txt) */
    /* (not for tall-skinny shapes, m >= 16 n: the two copies of the matrix cost what the ragged panel does -- 100000 x 300 1.55 -> 1.68 ms
     * with them, 262144 x 500 5.8 -> 5.6; from 10000 x 1000 (2.79 -> 2.65) to 1000^2 (2.05 -> 1.85) and 8192 x 8191 (25.3 -> 24.2) they pay) */
    if (!tsqr_local && m >= 512 && n >= 64 && n % 32 != 0 && (long long) m < 16LL * n && m <= 2147483647 - 64 &&
        ((double) m + 48.0) * ((n + 31) & ~31) * 8.0 <= 4294967296.0) {
        n = (n + 31) & ~31;
        m = (m + (n - n_user) + 15) & ~15;
    }
    else if (!tsqr_local && m >= 512 && m % 16 != 0 && (double) m * n * 8.0 <= 4294967296.0 && m <= 2147483647 - 16) m = (m + 15) & ~15;
    {
        int dnb, dib;
        default_blocks(tsqr_local ? (1 << 30) : m, n, &dnb, &dib);
        if (ib <= 0) ib = dib;
        if (nb <= 0) nb = (dnb % ib == 0) ? dnb : 128;
    }
    if (ib > QRD_LEAFW || nb < ib || nb > QR_MAX_NB || nb % ib || (nb > QR_HALF && nb % QR_HALF)) return QR_E_ARG;
    qr_plan* p = (qr_plan*) calloc(1, sizeof(qr_plan));
    if (!p) return QR_E_ALLOC;
    p->m = m; p->n = n; p->nb = nb; p->ib = ib; p->m_user = m_user; p->n_user = n_user;
    p->ldv = (m + 127) & ~127;       /* (a multiple of 128: the update kernel's tile loader reads V to the end of the last row tile, gemm_nt4_kernel<.., RAG>) */
    p->ldt = nb;
    const char* la = getenv("MI355XQR_LOOKAHEAD");
    /* Look-ahead (two streams, CU partition): where lookahead_pays() says so; MI355XQR_LOOKAHEAD = 0 / 1 decides instead.  (History of the
     * rule: profiles/r02_session2_ab_measurements.txt section 12, profiles/r06_lookahead_threshold.txt, profiles/r06_block_size_lookahead_rule.txt) */
    p->lookahead = tsqr_local ? 0 : (la ? atoi(la) != 0 : lookahead_pays(m, n, nb));
    /* MI355XQR_GRAPH=1: the single-stream schedule captured once per argument set and replayed (no measured gain: the cost of a leaf is
     * the device-side kernel boundary, not the host launch).  Never with look-ahead: capturing the CU-masked two-stream schedule
     * crashes inside the runtime (round 3: segmentation fault in hipStreamEndCapture), so the knob is ignored there. */
    const char* gr = lab_getenv("MI355XQR_GRAPH");
    p->use_graph = (gr ? atoi(gr) != 0 : 0) && !p->lookahead;
    /* MI355XQR_SPLIT = "c0:f0,c1:f1,...,ck": the panel chain runs on its own c_i compute units and the wide update
     * on the other 256-c_i while more than the fraction f_i of the columns is still to be factored (last entry: to
     * the end), so a leaf kernel never queues behind resident GEMM workgroups and the split follows the shrinking
     * trailing matrix.  "0" = no partition (shared CUs, stream priority only); a single number c is the one-phase
     * form.  Default: partition when there is a wide update worth overlapping with (n >= 2048);
     * tall-skinny problems are all panel, so they keep the whole chip on one stream set. */
    /* (a multi-rank TSQR's local plan takes a normal-priority stream: the stacked plan's high-priority stream, which carries the exchange
     * and the short stacked panels everyone is waiting for, goes first whenever both have work queued) */
    /* MI355XQR_TSQR_RESERVE_CUS=c (multi-rank local plans only; default 0 = off): the local factorisation's stream is masked to all but c
     * compute units, so that RCCL's kernels -- a handful of workgroups -- never wait for a slot behind a chip-filling update.  Stream priority
     * alone orders the QUEUES, it does not free a compute unit.  c is rounded up to a multiple of 32 = one CU of every shader engine of
     * every XCD (a mask bit is CU i/8 of XCC i%8, and CU j of an XCC sits in its shader engine j%4): the dispatcher deals the workgroups of a
     * launch evenly over the shader engines whatever the mask says, so with 16 reserved (two engines of each XCD one CU short) the
     * one-workgroup-per-CU passes of the full-width panel ran a second round on those engines and took twice as long
     * (local QR of a 262144 x 512 shard 4.7 -> 7.0 ms; with 32: see profiles/NOTES.md). */
    int rc = 0;
    {
        int reserve = tsqr_local ? env_int("MI355XQR_TSQR_RESERVE_CUS", 0) : 0;
        if (reserve > 0) reserve = (reserve + 31) / 32 * 32;
        int cus = 256;
        qrd_device_info(NULL, 0, &cus, NULL, NULL);
        if (reserve > 0 && reserve < cus) rc = qrd_stream_create_cumask(&p->s_main, reserve, cus - reserve);
        else rc = qrd_stream_create(&p->s_main, !tsqr_local);
    }
    p->stream = p->s_main;
    p->pair_cur = -1;
    if (!rc && p->lookahead) {
        const char* sp = getenv("MI355XQR_SPLIT");
        char spec[128];
        if (sp) snprintf(spec, sizeof spec, "%s", sp);
        else snprintf(spec, sizeof spec, "%s", n >= 2048 ? (((m >= 10240 && n >= 10240) || (n >= 8192 && 4LL * m <= 5LL * n)) ? QR_DEFAULT_SPLIT_BIG : QR_DEFAULT_SPLIT) : "0");
        int cus = 256;
        qrd_device_info(NULL, 0, &cus, NULL, NULL);
        char* save = NULL;
        for (char* tok = strtok_r(spec, ",", &save); tok && !rc && p->npairs < QR_MAX_PAIRS; tok = strtok_r(NULL, ",", &save)) {
            const int c = atoi(tok);
            const char* colon = strchr(tok, ':');
            if ((tok[0] == 'U' || tok[0] == 'u') && p->npairs > 0) {
                /* "U": from here on the panel chain runs on an UNMASKED stream (any compute unit that is free -- in the chain-bound
                 * phase the update stream idles half of the time) while the wide update keeps the previous phase's masked stream */
                const int i = p->npairs++;
                p->pair_until[i] = 0.0;
                rc = qrd_stream_create(&p->s_pair[i][0], 1);
                p->s_pair[i][1] = p->s_pair[i - 1][1];
                p->pair_shared_u[i] = 1;
                continue;
            }
            if (c <= 0 || c >= cus) {           /* "0" or a bad entry: no partition; drop what was already created */
                for (int i = 0; i < p->npairs; ++i)
                    for (int j = 0; j < 2; ++j) { if (!(j == 1 && p->pair_shared_u[i])) qrd_stream_destroy(p->s_pair[i][j]); p->s_pair[i][j] = NULL; }
                p->npairs = 0;
                break;
            }
            const int i = p->npairs++;
            p->pair_until[i] = colon ? atof(colon + 1) : 0.0;
            rc = qrd_stream_create_cumask(&p->s_pair[i][0], 0, c);
            if (!rc) rc = qrd_stream_create_cumask(&p->s_pair[i][1], c, cus - c);
        }
        if (p->npairs) p->pair_until[p->npairs - 1] = 0.0;
    }
    if (!rc && p->npairs == 0) rc = qrd_stream_create(&p->stream_u, 0);
    for (int e = 0; e < 2 && !rc; ++e) rc = qrd_event_create_notiming(&p->ev_hop[e]);
    for (int e = 0; e < 2 && !rc; ++e) rc = qrd_event_create_notiming(&p->ev_extra[e]);
    for (int e = 0; e < 2 && !rc; ++e) rc = qrd_event_create_notiming(&p->ev_next[e]);
    for (int e = 0; e < 2 && !rc; ++e) rc = qrd_event_create_notiming(&p->ev_half[e]);
    {
        /* MI355XQR_NEXT=panel|update|auto: which stream applies panel s to the columns of panel s+1 (the look-ahead update
         * N(s)).  On the update stream it is a 60 us job for 190+ CUs instead of a 180 us one for the panel stream's few --
         * but while the update stream is busy back to back it would only delay W(s); auto (default) switches with the phase. */
        const char* nx = lab_getenv("MI355XQR_NEXT");
        p->next_on_update = !nx ? 2 : (strcmp(nx, "update") == 0 ? 1 : (strcmp(nx, "panel") == 0 ? 0 : 2));   /* 2 = by phase */
        if (!nx && p->npairs && qrd_stream_cus(p->s_pair[0][0]) <= 32) p->next_on_update = 1;
    }
    {
        /* MI355XQR_BALANCE = "Rp,Ru,tc0,tc1" (TFLOP/s on the panel CUs, on the update CUs; next-panel chain time
         * tc0 + tc1*mk/16384 ms at nb = 256); "0" = the panel stream takes no share of the wide update */
        const char* b = lab_getenv("MI355XQR_BALANCE");
        p->bal_rp = 14.0; p->bal_ru = 44.0; p->bal_tc0 = 1.1; p->bal_tc1 = 0.6;
        if (p->npairs) {            /* measured ~0.22 TFLOP/s per CU for the K = 256 update GEMMs on either side of the partition */
            int cus = 256;
            qrd_device_info(NULL, 0, &cus, NULL, NULL);
            const int pc = qrd_stream_cus(p->s_pair[0][0]);
            p->bal_rp = 0.22 * pc; p->bal_ru = 0.23 * (cus - pc);
        }
        if (b) {
            p->bal_rp = 0.0;
            sscanf(b, "%lf,%lf,%lf,%lf", &p->bal_rp, &p->bal_ru, &p->bal_tc0, &p->bal_tc1);
        }
        {
            const char* tt = lab_getenv("MI355XQR_TAILTC");          /* lab: ms; "0" = off */
            p->bal_tail_tc = tt ? atof(tt) : QR_TAIL_TC_MS;
        }
        p->bal_auto = (b == NULL) && p->npairs > 1;
        p->bal_tc0_base = p->bal_tc0; p->bal_tc1_base = p->bal_tc1;
    }
    for (int e = 0; e < 2 && !rc; ++e) rc = qrd_event_create_notiming(&p->ev_v[e]);
    for (int e = 0; e < 2 && !rc; ++e) {
        rc = qrd_event_create_notiming(&p->ev_panel[e]);
        if (!rc) rc = qrd_event_create_notiming(&p->ev_wide[e]);
    }
    size_t cap = (size_t) 256 * nb * (size_t) n;
    if (cap > ((size_t) 16 << 20)) cap = (size_t) 16 << 20;
    if (cap < ((size_t) 1 << 16)) cap = (size_t) 1 << 16;
    p->slab_cap = cap;
    p->w_cap = (size_t) nb * n;
    for (int e = 0; e < 2 && !rc; ++e) {
        rc = qrd_malloc((void**) &p->Vw2[e], sizeof(double) * (size_t) p->ldv * nb);
        if (!rc) rc = qrd_malloc((void**) &p->VT2[e], sizeof(double) * (size_t) p->ldv * nb);
        if (!rc) rc = qrd_malloc((void**) &p->T2[e], sizeof(double) * (size_t) nb * nb);
        if (!rc) rc = qrd_memset(p->stream, p->T2[e], 0, sizeof(double) * (size_t) nb * nb);
        if (!rc) rc = qrd_memset(p->stream, p->Vw2[e], 0, sizeof(double) * (size_t) p->ldv * nb);
    }
    p->Vw = p->Vw2[0]; p->VT = p->VT2[0]; p->T = p->T2[0];
    if (!rc) rc = qrd_malloc((void**) &p->W, sizeof(double) * p->w_cap);
    if (!rc) rc = qrd_malloc((void**) &p->Wn, sizeof(double) * (size_t) nb * nb);
    if (!rc && p->npairs && p->bal_rp > 0.0) rc = qrd_malloc((void**) &p->We, sizeof(double) * p->w_cap);
    if (!rc && p->We) rc = qrd_malloc((void**) &p->Ye, sizeof(double) * p->w_cap);
    if (!rc) rc = qrd_malloc((void**) &p->Yn, sizeof(double) * (size_t) nb * nb);
    if (!rc && (size_t) n * 8 <= (size_t) m) rc = qrd_malloc((void**) &p->Ye2, sizeof(double) * p->w_cap);
    if (!rc) rc = qrd_malloc((void**) &p->slabs_u, sizeof(double) * p->slab_cap);
    if (!rc) rc = qrd_malloc((void**) &p->Tt, sizeof(double) * (size_t) nb * nb);
    if (!rc) rc = qrd_malloc((void**) &p->G, sizeof(double) * (size_t) nb * nb);
    if (!rc) rc = qrd_malloc((void**) &p->X, sizeof(double) * (size_t) nb * nb);
    if (!rc) rc = qrd_malloc((void**) &p->slabs, sizeof(double) * p->slab_cap);
    {
        const char* pa = getenv("MI355XQR_PANEL");
        p->panel_tsqr = (pa && strcmp(pa, "tsqr") == 0) ? 1 : 3;
    }
    if (!rc) rc = qrd_malloc((void**) &p->panel_ws, sizeof(double) * qrd_panel_ws_size(m));
    if (!rc && p->panel_tsqr == 3) rc = qrd_malloc((void**) &p->chol_ws, sizeof(double) * QRD_CHOLQR_WS);
    if (!rc && p->panel_tsqr == 3 && knobs()->early_product) {
        /* 32 x (nb - 32) outputs per K slice, up to 128 slices */
        p->slab_ep_cap = (size_t) 32 * (size_t) (nb > 64 ? nb : 64) * 128;
        rc = qrd_malloc((void**) &p->slabs_ep, sizeof(double) * p->slab_ep_cap);
    }
    if (!rc && p->panel_tsqr == 3 && knobs()->fused_panel) {
        if (knobs()->cqr_min_rows > 0 && m >= plan_cqr_min_rows(p)) {
            rc = qrd_malloc((void**) &p->cq_ws, sizeof(double) * qrd_panel_cqr_ws_doubles());
            if (!rc) rc = qrd_malloc((void**) &p->cq_status, 4 * sizeof(int));
            if (!rc) rc = qrd_memset(p->stream, p->cq_status, 0, 4 * sizeof(int));
            if (!rc) rc = qrd_host_word_alloc(&p->cq_hword, &p->cq_hword_dev);
            {
                const char* gm = getenv("MI355XQR_GUARD");      /* latch | host (default) */
                p->guard_latch = gm != NULL && strcmp(gm, "latch") == 0;
            }
        }
        if (!rc) rc = qrd_malloc((void**) &p->pf_ws, sizeof(double) * qrd_panel_fused_ws_doubles());
        if (!rc) rc = qrd_memset(p->stream, p->pf_ws, 0, sizeof(double) * qrd_panel_fused_ws_doubles());
        if (!rc) rc = qrd_malloc((void**) &p->pf_status, 4 * sizeof(int));
        if (!rc) rc = qrd_memset(p->stream, p->pf_status, 0, 4 * sizeof(int));
    }
    if (!rc && (p->m != p->m_user || p->n != p->n_user)) rc = qrd_malloc((void**) &p->padA, sizeof(double) * (size_t) p->m * p->n);
    if (!rc && p->n != p->n_user) rc = qrd_malloc((void**) &p->pad_tau, sizeof(double) * (size_t) p->n);
    if (rc) { qr_plan_destroy(p); return rc; }
    *out = p;
    return 0;
}

int qr_plan_destroy(qr_plan* p)
{
    if (!p) return 0;
    qr_plan_sync(p);
    qrd_graph_destroy(p->graph_exec);
    for (int e = 0; e < 2; ++e) {
        if (p->ev_panel[e]) qrd_event_destroy(p->ev_panel[e]);
        if (p->ev_v[e]) qrd_event_destroy(p->ev_v[e]);
        if (p->ev_wide[e]) qrd_event_destroy(p->ev_wide[e]);
        qrd_free(p->Vw2[e]); qrd_free(p->VT2[e]); qrd_free(p->T2[e]);
    }
    qrd_free(p->Wn); qrd_free(p->slabs_u); qrd_free(p->We); qrd_free(p->Ye); qrd_free(p->Yn); qrd_free(p->Ye2);
    for (int i = 0; i < p->npairs; ++i)
        for (int j = 0; j < 2; ++j)
            if (p->s_pair[i][j] && !(j == 1 && p->pair_shared_u[i])) qrd_stream_destroy(p->s_pair[i][j]);
    if (p->npairs == 0 && p->stream_u) qrd_stream_destroy(p->stream_u);
    for (int e = 0; e < 2; ++e) {
        if (p->ev_hop[e]) qrd_event_destroy(p->ev_hop[e]);
        if (p->ev_extra[e]) qrd_event_destroy(p->ev_extra[e]);
        if (p->ev_next[e]) qrd_event_destroy(p->ev_next[e]);
        if (p->ev_half[e]) qrd_event_destroy(p->ev_half[e]);
    }
    for (int i = 0; i < 2 * p->prof_cap; ++i)
        if (p->prof_ev && p->prof_ev[i]) qrd_event_destroy(p->prof_ev[i]);
    free(p->prof_ev); free(p->prof_cls); free(p->prof_flops); free(p->prof_bytes);
    qrd_free(p->W); qrd_free(p->Tt); qrd_free(p->G); qrd_free(p->X);
    qrd_free(p->slabs); qrd_free(p->panel_ws); qrd_free(p->chol_ws); qrd_free(p->slabs_ep);
    qrd_free(p->pf_ws); qrd_free(p->pf_status); qrd_free(p->cq_ws); qrd_free(p->cq_status); qrd_free(p->padA); qrd_free(p->pad_tau);
    qrd_host_word_free(p->cq_hword);
    if (p->s_main) qrd_stream_destroy(p->s_main);
    free(p);
    return 0;
}

/* The status words the panel kernels leave on the device, read where the host waits anyway: a hand-off between the workgroups of a
 * one-launch panel that timed out (the launch ended with garbage: QR_E_STALL), and -- latch mode only -- tall panels the guard refused
 * (QR_E_REFUSED: the factorisation behind them is invalid).  Both are cleared by the read. */
static int plan_read_status(qr_plan* p)
{
    int rc = 0;
    if (p->pf_dirty && p->pf_status) {
        int st[4] = {0, 0, 0, 0};
        CHECK(qrd_d2h(p->s_main, st, p->pf_status, sizeof st));
        CHECK(qrd_stream_sync(p->s_main));
        p->pf_dirty = 0;
        if (st[0] || st[1]) {
            CHECK(qrd_memset(p->s_main, p->pf_status, 0, sizeof st));
            CHECK(qrd_stream_sync(p->s_main));
        }
        p->n_pf_leaf_fallback += st[0];
        if (st[1]) { p->n_pf_stall += 1; rc = QR_E_STALL; }
    }
    if (p->cq_dirty && p->cq_status) {       /* (keyed on what was ISSUED, not on the current mode: qr_plan_set_guard_mode may have switched since) */
        int st[4] = {0, 0, 0, 0};
        CHECK(qrd_d2h(p->s_main, st, p->cq_status, sizeof st));
        CHECK(qrd_stream_sync(p->s_main));
        p->cq_dirty = 0;
        if (st[1]) {
            CHECK(qrd_memset(p->s_main, p->cq_status, 0, sizeof st));
            CHECK(qrd_stream_sync(p->s_main));
            p->n_cqr_refused += st[1];
            if (!rc) rc = QR_E_REFUSED;
        }
    }
    return rc;
}

int qr_plan_sync(qr_plan* p)
{
    if (!p) return QR_E_ARG;
    for (int i = 0; i < p->npairs; ++i)
        for (int j = 0; j < 2; ++j)
            if (p->s_pair[i][j]) CHECK(qrd_stream_sync(p->s_pair[i][j]));
    if (p->npairs == 0 && p->stream_u) CHECK(qrd_stream_sync(p->stream_u));
    if (p->s_main) CHECK(qrd_stream_sync(p->s_main));
    return plan_read_status(p);
}

/* Drains the plan first: panels issued in the old mode have their status read (and reported: the return value is qr_plan_sync's) before
 * the mode changes -- a refusal latched earlier is neither lost nor left to surface under the other mode's rules. */
int qr_plan_set_guard_mode(qr_plan* p, int latch)
{
    if (!p || (latch != 0 && latch != 1)) return QR_E_ARG;
    const int rc = qr_plan_sync(p);
    p->guard_latch = latch;
    return rc;
}

int qr_plan_route_stats(qr_plan* p, long long* out4)
{
    if (!p || !out4) return QR_E_ARG;
    out4[0] = p->n_cqr; out4[1] = p->n_cqr_refused; out4[2] = p->n_pf_leaf_fallback; out4[3] = p->n_pf_stall;
    return 0;
}
int qr_plan_retry_stats(qr_plan* p, long long* out2)
{
    if (!p || !out2) return QR_E_ARG;
    out2[0] = p->n_cqr_retried; out2[1] = p->n_cqr_retry_ok;
    return 0;
}
void* qr_plan_stream(qr_plan* p) { return p ? p->s_main : NULL; }

int qr_plan_info(qr_plan* p, int* nb, int* ib, int* lookahead)
{
    if (!p) return QR_E_ARG;
    if (nb) *nb = p->nb;
    if (ib) *ib = p->ib;
    if (lookahead) *lookahead = p->lookahead;
    return 0;
}

/* compute units the wide trailing update runs on (the update stream's share of the CU partition, or the whole device) */
int qr_plan_update_cus(qr_plan* p)
{
    if (!p) return QR_E_ARG;
    if (p->npairs) return qrd_stream_cus(p->s_pair[0][1]);
    int cus = 0;
    if (qrd_device_info(NULL, 0, &cus, NULL, NULL)) return QR_E_INTERNAL;
    return cus;
}

static int ensure_w(qr_plan* p, size_t elems)
{
    if (elems <= p->w_cap) return 0;
    CHECK(qr_plan_sync(p));
    qrd_graph_destroy(p->graph_exec);       /* a captured factorisation holds the old W pointer */
    p->graph_exec = NULL; p->g_dA = NULL;
    qrd_free(p->W);
    p->W = NULL; p->w_cap = 0;
    CHECK(qrd_malloc((void**) &p->W, sizeof(double) * elems));
    p->w_cap = elems;
    return 0;
}

/* ---- profiling (HIP events on the plan's stream) ---------------------------------------------- */
int qr_plan_set_profile(qr_plan* p, int on)
{
    if (!p) return QR_E_ARG;
    /* on = 1: every class; on = 2 * mask (even): only the classes whose bit is set in mask (bit c = class c of the header, bits 4 / 5 =
     * look-ahead update / panel stream's share) -- each record is two event packets on a stream, ~4 us of queue time apiece, and
     * the bench's timed region only needs the dominant kernel's */
    p->prof_on = on ? 1 : 0;
    p->prof_mask = (on & 1) || !on ? 0x3f : ((on >> 1) & 0x3f);
    p->prof_count = 0;
    p->prof_open = 0;
    p->prof_paused = 0;
    return 0;
}

/* suspend / resume recording without dropping what has been recorded (bench.py brackets a SAMPLE of the timed steps: a record is two
 * event packets on a stream, and 91 bracketed launches per 16384^2 step cost the step 0.5 ms) */
int qr_plan_pause_profile(qr_plan* p, int pause)
{
    if (!p) return QR_E_ARG;
    p->prof_paused = pause != 0;
    return 0;
}

static int prof_begin_on(qr_plan* p, int cls, void* stream)
{
    if (!p->prof_on || p->prof_paused || !(p->prof_mask & (1 << cls))) return 0;
    if (p->prof_count == p->prof_cap) {
        const int ncap = p->prof_cap ? 2 * p->prof_cap : 1024;
        void** ev = (void**) realloc(p->prof_ev, sizeof(void*) * 2 * ncap);
        if (!ev) return QR_E_ALLOC;
        p->prof_ev = ev;
        int* cl = (int*) realloc(p->prof_cls, sizeof(int) * ncap);
        if (!cl) return QR_E_ALLOC;
        p->prof_cls = cl;
        double* fl = (double*) realloc(p->prof_flops, sizeof(double) * ncap);
        if (!fl) return QR_E_ALLOC;
        p->prof_flops = fl;
        double* by = (double*) realloc(p->prof_bytes, sizeof(double) * ncap);
        if (!by) return QR_E_ALLOC;
        p->prof_bytes = by;
        for (int i = 2 * p->prof_cap; i < 2 * ncap; ++i) p->prof_ev[i] = NULL;
        p->prof_cap = ncap;
    }
    const int r = p->prof_count;
    for (int e = 0; e < 2; ++e)
        if (!p->prof_ev[2 * r + e]) CHECK(qrd_event_create_timing(&p->prof_ev[2 * r + e]));
    p->prof_cls[r] = cls;
    p->prof_open = 1;
    p->prof_stream = stream;
    return qrd_event_record(p->prof_ev[2 * r], stream);
}

static int prof_begin(qr_plan* p, int cls) { return prof_begin_on(p, cls, p->stream); }

static int prof_end(qr_plan* p, double flops, double bytes)
{
    if (!p->prof_on || !p->prof_open) return 0;
    const int r = p->prof_count++;
    p->prof_flops[r] = flops;
    p->prof_bytes[r] = bytes;
    p->prof_open = 0;
    return qrd_event_record(p->prof_ev[2 * r + 1], p->prof_stream);
}

int qr_plan_get_profile(qr_plan* p, qr_profile* out)
{
    if (!p || !out) return QR_E_ARG;
    memset(out, 0, sizeof(*out));
    CHECK(qr_plan_sync(p));
    for (int r = 0; r < p->prof_count; ++r) {
        float ms = 0.f;
        CHECK(qrd_event_elapsed_ms(p->prof_ev[2 * r], p->prof_ev[2 * r + 1], &ms));
        const int c = p->prof_cls[r] < QR_PROF_CLASSES ? p->prof_cls[r] : 3;     /* look-ahead / balance updates count as misc */
        out->ms[c] += ms;
        out->flops[c] += p->prof_flops[r];
        out->bytes[c] += p->prof_bytes[r];
        out->launches[c] += 1;
    }
    p->prof_count = 0;
    return 0;
}

/* raw records of the last profiled run, in issue order: internal class (0..3 as in the header; 4 = look-ahead update N(s),
 * 5 = the panel stream's share E(s) of a wide update), start and end in ms since the first record's start.  Does not reset. */
int qr_plan_get_profile_records(qr_plan* p, int max, int* cls, double* t0_ms, double* t1_ms)
{
    if (!p || max < 0) return QR_E_ARG;
    if (qr_plan_sync(p)) return QR_E_INTERNAL;
    int n = p->prof_count < max ? p->prof_count : max;
    for (int r = 0; r < n; ++r) {
        float a = 0.f, b = 0.f;
        if (qrd_event_elapsed_ms(p->prof_ev[0], p->prof_ev[2 * r], &a) || qrd_event_elapsed_ms(p->prof_ev[0], p->prof_ev[2 * r + 1], &b))
            return QR_E_INTERNAL;
        cls[r] = p->prof_cls[r]; t0_ms[r] = a; t1_ms[r] = b;
    }
    return n;
}

/* ---- thin wrappers --------------------------------------------------------------------------- */
static int tn(qr_plan* p, int M, int N, int K, const double* A, int lda, const double* B, int ldb, double* C,
              int ldc, const double* Tm)
{
    return qrd_gemm_tn(p->stream, M, N, K, 1.0, A, lda, B, ldb, 0.0, C, ldc, p->slabs, p->slab_cap, Tm, p->ldt);
}

int qr_gemm_dev(qr_plan* p, char transa, int M, int N, int K, double alpha, const double* dA, int lda,
                const double* dB, int ldb, double beta, double* dC, int ldc)
{
    if (!p || !dA || !dB || !dC) return QR_E_ARG;
    if (transa == 'N' || transa == 'n') return qrd_gemm_nn(p->stream, M, N, K, alpha, dA, lda, dB, ldb, beta, dC, ldc);
    if (transa == 'T' || transa == 't')
        return qrd_gemm_tn(p->stream, M, N, K, alpha, dA, lda, dB, ldb, beta, dC, ldc, p->slabs, p->slab_cap, NULL, 0);
    return QR_E_ARG;
}

/* ---- factorisation --------------------------------------------------------------------------- */
/* One outer panel: columns [k, k+wout) over rows [k, m).  Leaves V (explicit, unit lower trapezoid)
 * in p->Vw and the panel's compact-WY T in p->T (only if want_t). */
static int apply_small_t(qr_plan* p, void* stream, const double* V, int ldv, const double* T, int ldt, int mk, int kw, double* A2,
                         int lda, int nc, double* Wbuf, double* Ybuf, double* slabs);
static int apply_vw(qr_plan* p, void* stream, const double* V, int ldv, int mk, int kw, double* A2, int lda, int nc, double* Wbuf, double* Ybuf);

/* One outer panel: columns [k, k+wout) over rows [k, m).  Leaves V (explicit, unit lower trapezoid) in p->Vw and the panel's
 * compact-WY T in p->T (only if want_t).
 * Two-level panel: an outer block wider than QR_HALF is factored half by half -- the leaves of a half update only the rest of
 * THAT half (K = 32), and the next half receives the whole previous part in one block update with K = 256 ... (the in-panel
 * traffic of a 512-wide block would otherwise be 4x that of a 256-wide one).  The wide trailing update then runs with
 * K = wout = 512: half the C traffic per flop of K = 256, which is what bounds it.
 * half_ready: event the second half's columns must wait for (the wide update of the previous panel reaches them on the
 * update stream while the first half is being factored), or NULL. */
static int factor_panel_inner(qr_plan* p, double* dA, int m, int lda, int k, int wout, double* dtau, int want_t, void* half_ready);

static int factor_panel(qr_plan* p, double* dA, int m, int lda, int k, int wout, double* dtau, int want_t, void* half_ready)
{
    qrd_range_push("mi355xqr panel");                 /* host-side issue range (MI355XQR_ROCTX=1) */
    const int rc = factor_panel_inner(p, dA, m, lda, k, wout, dtau, want_t, half_ready);
    qrd_range_pop();
    return rc;
}

/* A tall half (mkh x wh, wh <= 128) at its full width: CholeskyQR2 + Householder reconstruction in three passes instead of the leaf chain's
 * twelve (qr_panel_cqr.hip); T of the half comes out complete.  Q lives in Qh (the set's V*T buffer, free while its panel is being
 * factored), so a refused panel leaves A AND Vh untouched.  Returns 1 when the guard refused the panel (the caller runs the leaf chain),
 * 0 when done.
 * The verdict exists two thirds into the panel (after the 128 x 128 reconstruction, before the last pass over the panel): the kernel
 * that forms it publishes it into a host word, and the host reads THAT -- with the last pass still queued and running -- instead of
 * draining the stream as round 4 did: whatever comes next is queued ~0.2 ms before the stream needs it, the GPU never idles.  The host
 * thread does wait for that word; a caller that must not block inside qr_geqrf_dev sets the latch mode (qr_plan_set_guard_mode), in
 * which nothing is read here and a refusal is reported by qr_plan_sync. */
static int panel_cqr_half(qr_plan* p, double* Ah, int lda, int mkh, int wh, double* tauh, double* Th, int ldt, double* Vh, int ldv, double* Qh, int park)
{
    const unsigned seq = (++p->cq_seq) & 0x3fffffffu;
    const int latch = p->guard_latch || !p->cq_hword;
    CHECK(qrd_panel_cqr_p(p->stream, Ah, lda, mkh, wh, tauh, Th, ldt, Vh, ldv, p->cq_ws, p->cq_status, Qh, ldv, latch ? NULL : p->cq_hword_dev, seq, park));
    p->n_cqr += 1;
    if (latch) { p->cq_dirty = 1; return 0; }
    struct timespec t0, t1;
    clock_gettime(CLOCK_MONOTONIC, &t0);
    unsigned cur = seq;
    int retried = 0;
    for (unsigned spins = 0;; ++spins) {
        const unsigned v = __atomic_load_n(p->cq_hword, __ATOMIC_ACQUIRE);
        if ((v >> 1) == cur) {
            if (!(v & 1u)) { if (retried) p->n_cqr_retry_ok += 1; return 0; }
            if (!retried) p->n_cqr_refused += 1;
            /* Refused (cond > ~1e7, or not of full rank).  Once more, PRECONDITIONED (shifted CholeskyQR3, qrd_panel_cqr_retry: R0 from the
             * shifted Gram matrix this attempt left in the workspace, the same pipeline on A R0^-1) -- queued, like everything here, while
             * the refused attempt's remaining launches return at once; only a panel refused twice goes to the leaf chain */
            if (retried || !knobs()->cqr_retry || Qh == Vh || Qh == Ah) return 1;
            retried = 1;
            cur = (++p->cq_seq) & 0x3fffffffu;
            CHECK(qrd_panel_cqr_retry(p->stream, Ah, lda, mkh, wh, tauh, Th, ldt, Vh, ldv, p->cq_ws, p->cq_status, Qh, ldv, p->cq_hword_dev, cur, park));
            p->n_cqr_retried += 1;
            clock_gettime(CLOCK_MONOTONIC, &t0);
            continue;
        }
        cpu_relax();                         /* the verdict is ~0.3 ms away: do not hammer the line the device is about to write */
        if ((spins & 1023u) == 1023u) {
            clock_gettime(CLOCK_MONOTONIC, &t1);
            const double waited = (t1.tv_sec - t0.tv_sec) + 1e-9 * (t1.tv_nsec - t0.tv_nsec);
            if (waited > 5.0) break;
            if (waited > 0.002) sched_yield();       /* far beyond a panel's time (a queue full of other work in front): let the core go */
        }
    }
    /* the word never came (a fault in front of it?): drain the stream -- an error there is the caller's answer -- and read the device word */
    CHECK(qrd_stream_sync(p->stream));
    int st[4] = {0, 0, 0, 0};
    CHECK(qrd_d2h(p->stream, st, p->cq_status, sizeof st));
    CHECK(qrd_stream_sync(p->stream));
    if (st[0] && !retried) p->n_cqr_refused += 1;          /* (a retried panel's first refusal has been counted) */
    if (!st[0] && retried) p->n_cqr_retry_ok += 1;
    return st[0] != 0;
}

/* R of a parked panel back into the top block of the caller's array (after the updates that read V from there) */
static int cq_unpark(qr_plan* p)
{
    if (!p->cq_parked) return 0;
    p->cq_parked = 0;
    return qrd_panel_cqr_restore_r(p->stream, p->cq_park_top, p->cq_park_lda, p->cq_park_w, p->cq_ws, p->cq_status);
}

static int factor_panel_inner(qr_plan* p, double* dA, int m, int lda, int k, int wout, double* dtau, int want_t, void* half_ready)
{
    const int park_hint = p->park_hint;
    p->park_hint = 0;
    p->t_deferred = 0;
    CHECK(cq_unpark(p));                                      /* (never pending here: every caller that hints unparks behind its update) */
    const int mk = m - k, ib = p->ib, ldv = p->ldv, ldt = p->ldt, nb = p->nb;
    double* Ak = dA + (size_t) k * lda + k;
    /* No per-panel zeroing of V's strictly upper part: column j of a panel only ever receives rows >= ib * floor(j / ib) (every
     * leaf route writes from its own diagonal block down, whatever the panel), so the zeros written once at plan creation stay
     * (5 us per panel on the critical chain otherwise). */
    const int nhalf = (wout + QR_HALF - 1) / QR_HALF;
    p->vt_formed[p->Vw == p->Vw2[1]] = 0;                     /* this set's V*T (if any) belongs to an older panel */
    const qr_knobs* const kn = knobs();
    const int fuse_gram = 1;
    /* MI355XQR_FUSE_NN=0: in-panel update and the next leaf's Gram matrix as separate launches (gemm_nn + gram32_kernel);
     * QR_FUSE_NN_MIN_ROWS: leaf heights the fused launch is used for (tall leaves only -- it saves a pass over
     * the next leaf, 262144 x 512: 7.15 -> 7.08 ms; on the short leaves of square problems its 128 matrix-core instructions per
     * wave sit on 10-14 compute units and the launch takes 18 us where gemm_nn + gram32 take 14: 8192^2 32.4 -> 33.0 ms);
     * on tall leaves every workgroup walks all columns (V read once) */
    const int fuse_nn = kn->fuse_nn, fuse_nn_min = QR_FUSE_NN_MIN_ROWS, fuse_nn_max = 0, fuse_nn_gy_tall = 1;
    for (int h = 0; h < nhalf; ++h) {
        const int c0 = h * QR_HALF, wh = imin(QR_HALF, wout - c0), cend = c0 + wh;
        const int need_t = want_t || h + 1 < nhalf;          /* the next half's block update needs T of everything before it */
        /* the Gram blocks V(:, c0:c)^T V_l that T needs are collected leaf by leaf, in the same launch as the leaf's in-panel
         * product (qrd_gemm_tn_dual); gram_done stays 1 while every leaf of the half could do that */
        int gram_done = fuse_gram && need_t;
        int gram_nslab = 0;          /* > 0: p->slabs holds that many partial Gram matrices of the coming leaf (left by the previous leaf's update) */
        if (h > 0) {
            /* A(:, c0:cend) <- (I - V T V^T)^T A(:, c0:cend) with the c0 reflectors factored so far */
            if (half_ready) CHECK(qrd_stream_wait_event(p->stream, half_ready));
            CHECK(apply_small_t(p, p->stream, p->Vw, ldv, p->T, ldt, mk, c0, Ak + (size_t) c0 * lda, lda, wh, p->Wn, p->Yn, p->slabs));
        }
        /* the whole half in ONE launch (qr_panel_fused.hip: every leaf, its in-panel product and update; leaf T blocks, tau, V and
         * the Gram blocks for the merge below come out exactly as from the leaf loop) where the panel is short enough */
        int cqr_done = 0, t_merged = 0;
        if (p->cq_ws && !p->use_graph && ib == 32 && kn->cqr_min_rows > 0 && mk - c0 >= plan_cqr_min_rows(p) && (wh == 128 || (wh == wout && wh == 96 && nhalf == 1)) && qrd_panel_cqr_ok(mk - c0, wh)) {   /* (96: the whole leaves of a ragged last panel; at 32 the route's one-workgroup kernels cost more than the leaf: 100000 x 300 1.55 -> 1.64 ms) */
            /* (2: the last panel -- V once, R in place at once, nothing parked) */
            const int park = (park_hint && kn->cqr_park && nhalf == 1 && !p->lookahead) ? (want_t ? 1 : 2) : 0;
            const int rc = panel_cqr_half(p, Ak + (size_t) c0 * lda + c0, lda, mk - c0, wh, dtau + k + c0, p->T + (size_t) c0 * ldt + c0, ldt,
                                          p->Vw + (size_t) c0 * ldv + c0, ldv, p->VT + (size_t) c0 * ldv + c0, park);
            if (rc < 0) return rc;
            cqr_done = rc == 0;
            if (cqr_done && park == 1) { p->cq_parked = 1; p->cq_park_top = Ak; p->cq_park_lda = lda; p->cq_park_w = wh; }
        }
        const int fused_half = !cqr_done && p->pf_ws && !p->fused_off && !p->use_graph && ib == 32 && mk - c0 >= kn->fused_min_rows &&
                               qrd_panel_fused_ok(p->stream, Ak + (size_t) c0 * lda + c0, lda, mk - c0, wh, p->Vw + (size_t) c0 * ldv + c0, ldv);
        /* the Gram blocks out of the launch itself: always at 64 columns (one tile of product more, and the launch then merges T as well);
         * up to MI355XQR_FUSED_GRAM columns from 2048 rows on (1024^2 at nb 128: 1.98 against 2.04 ms with them, profiles/r06_fused_gram.txt) */
        const int fused_gram_here = wh <= 64 ? kn->fused_gram >= 64 : (wh <= kn->fused_gram && mk - c0 >= 2048);
        if (fused_half) {
            CHECK(qrd_panel_fused(p->stream, Ak + (size_t) c0 * lda + c0, lda, mk - c0, wh, dtau + k + c0, p->T + (size_t) c0 * ldt + c0, ldt,
                                  p->Vw + (size_t) c0 * ldv + c0, ldv, (need_t && fused_gram_here) ? p->G + (size_t) c0 * nb + c0 : NULL, nb, p->pf_ws,
                                  &p->pf_epoch, p->pf_status));
            p->pf_dirty = 1;
            gram_done = need_t && fused_gram_here;
            t_merged = gram_done && qrd_panel_fused_merges_t(wh, 1);       /* (64-column panels: the launch has merged its two T blocks itself) */
        }
        for (int c = c0; c < cend && !fused_half && !cqr_done; c += ib) {
            const int w = imin(ib, cend - c), mkl = mk - c;
            double* P = Ak + (size_t) c * lda + c;
            double* Vl = p->Vw + (size_t) c * ldv + c;
            double* Tl = p->T + (size_t) c * ldt + c;
            /* leaf algorithm (MI355XQR_PANEL = cholqr [default] | tsqr): CholeskyQR2 + Householder reconstruction (4 short launches)
             * guarded by the Householder TSQR leaf (4-6 launches, no-ops unless the guard trips); tsqr = the Householder TSQR leaf alone */
            const int nrest = cend - (c + w), nprev = gram_done ? c - c0 : 0;
            double* Arest = P + (size_t) w * lda;
            int fused = 0;
            if (p->panel_tsqr == 3 && p->slabs_ep && gram_done && w == 32 && nrest + nprev > 0) {
                /* the leaf and its two long-K products (W = T_l^T V_l^T A_rest into Wn; the Gram block V_prev^T V_l into G) in one
                 * call: the products run in the launch of the leaf's one-workgroup reconstruction (qr_panel_tsqr.hip, hr3_ep_kernel) */
                CHECK(qrd_panel_cholqr_ep(p->stream, P, lda, mkl, w, dtau + k + c, Tl, ldt, Vl, ldv, p->panel_ws, p->m, p->chol_ws,
                                          p->slabs, p->slab_cap, gram_nslab, nrest, Arest, lda, nprev, p->Vw + (size_t) c0 * ldv + c, ldv,
                                          p->Wn, w, p->G + (size_t) c * nb + c0, nb, p->slabs_ep, p->slab_ep_cap, &fused));
            } else if (p->panel_tsqr == 3)
                CHECK(qrd_panel_cholqr(p->stream, P, lda, mkl, w, dtau + k + c, Tl, ldt, Vl, ldv, p->panel_ws, p->m, p->chol_ws,
                                       p->slabs, p->slab_cap, gram_nslab));
            else
                CHECK(qrd_panel_tsqr(p->stream, P, lda, mkl, w, dtau + k + c, Tl, ldt, Vl, ldv, p->panel_ws, p->m));
            if (!fused && gram_done && w == 32 && nrest + nprev > 0) {
                /* Wn, not W: with look-ahead the wide update on stream_u owns p->W while this panel runs */
                const int rc = qrd_gemm_tn_dual(p->stream, nrest, nprev, mkl, Vl, ldv, Arest, lda, p->Vw + (size_t) c0 * ldv + c, ldv, Tl, ldt,
                                                p->Wn, w, p->G + (size_t) c * nb + c0, nb, p->slabs, p->slab_cap);
                if (rc == 0) fused = 1;
                else if (rc != -7) return rc;
            }
            if (!fused) {
                if (nprev > 0 || (c > c0 && w != 32)) gram_done = 0;     /* this leaf's Gram column is missing: whole Gram at the end */
                if (nrest > 0) CHECK(tn(p, w, nrest, mkl, Vl, ldv, Arest, lda, p->Wn, w, Tl));  /* T_l^T V_l^T A_rest */
            }
            gram_nslab = 0;
            if (nrest > 0) {
                /* A_rest -= V_l W; with the CholeskyQR2 leaf the same launch leaves the Gram matrix of the NEXT leaf's columns in
                 * p->slabs (qr_leaf_fused.hip: one launch less per leaf on the critical chain) */
                int rc = -7;
                if (fuse_nn && w == 32 && mkl >= fuse_nn_min && (fuse_nn_max <= 0 || mkl <= fuse_nn_max)) {
                    const int want_gram = p->panel_tsqr == 3 && nrest >= 32;
                    rc = qrd_leaf_update_gram(p->stream, mkl, nrest, Vl, ldv, p->Wn, Arest, lda, want_gram ? p->slabs : NULL, p->slab_cap / 2,
                                              mkl > 32768 ? fuse_nn_gy_tall : 0, &gram_nslab);
                    if (rc != 0 && rc != -7) return rc;
                }
                if (rc == -7) CHECK(qrd_gemm_nn(p->stream, mkl, nrest, w, -1.0, Vl, ldv, p->Wn, w, 1.0, Arest, lda));
            }
        }
        if (!need_t) continue;
        if (p->v_ready && nhalf == 1) CHECK(qrd_event_record(p->v_ready, p->stream));   /* V done; what follows only builds T */
        double* Vh = p->Vw + (size_t) c0 * ldv + c0;          /* this half's V: rows from c0 (zero above) */
        double* Thh = p->T + (size_t) c0 * ldt + c0;
        if (wh > ib && !cqr_done && p->defer_hint && fused_half && nhalf == 1 && !gram_done) {
            p->t_deferred = 1;                                /* V, tau and the leaves' T blocks are complete: the caller merges (deferred_t_merge) */
            continue;
        }
        if (wh > ib && !cqr_done) {
            double* Ghh = p->G + (size_t) c0 * nb + c0;
            if (!gram_done) CHECK(tn(p, wh, wh, mk - c0, Vh, ldv, Vh, ldv, Ghh, nb, NULL));       /* Gram of the half */
            /* V*T is not formed here: the look-ahead update applies T to the small product V^T A_next instead, and the
             * wide update builds V*T itself on its own stream (update_cols), off the critical path */
            if (!t_merged) CHECK(qrd_larft(p->stream, wh, ib, Ghh, nb, dtau + k + c0, Thh, ldt, NULL, 0, p->X, nb));
        }
        if (h > 0) {
            /* T(0:c0, c0:cend) = -T(0:c0, 0:c0) (V(:, 0:c0)^T V(:, c0:cend)) T(c0:cend, c0:cend) */
            double* G12 = p->G + (size_t) c0 * nb;
            CHECK(tn(p, c0, wh, mk - c0, p->Vw + c0, ldv, Vh, ldv, G12, nb, NULL));
            CHECK(qrd_gemm_nn(p->stream, c0, wh, wh, 1.0, G12, nb, Thh, ldt, 0.0, p->X, c0));
            CHECK(qrd_gemm_nn(p->stream, c0, wh, c0, -1.0, p->T, ldt, p->X, c0, 0.0, p->T + (size_t) c0 * ldt, ldt));
        }
    }
    return 0;
}

static void use_set(qr_plan* p, int e) { p->Vw = p->Vw2[e]; p->VT = p->VT2[e]; p->T = p->T2[e]; }

/* A2 (mk x nc) <- (I - V T V^T)^T A2 for V: mk x kw, T: kw x kw upper, WITHOUT forming V T:  Y = V^T A2 (long-K product),
 * W = T^T Y (small), A2 -= V W.  Used where V*T does not exist yet: look-ahead update, mid-panel update, the panel stream's
 * share of a wide update.  Ybuf, Wbuf: kw * nc doubles each. */
static int apply_small_t(qr_plan* p, void* stream, const double* V, int ldv, const double* T, int ldt, int mk, int kw, double* A2,
                         int lda, int nc, double* Wbuf, double* Ybuf, double* slabs)
{
    /* p->t_wait (set by the caller around ONE call): event after which T is complete -- V was complete before this call, so the
     * long-K product Y = V^T A2 runs under the T merge of the panel stream and only W = T^T Y waits */
    void* const t_wait = p->t_wait;
    p->t_wait = NULL;
    /* narrow panels: T^T is folded into the reduction of the split-K slabs (one thread per entry of W walks a column of T:
     * fine for kw <= 128, 32 KB of T per output column; at kw = 256 every column's workgroup would pull 256 KB through L2) */
    const int fold_max = QR_TFOLD_MAX;
    if (kw <= fold_max && slabs != NULL) {
        if (t_wait) CHECK(qrd_stream_wait_event(stream, t_wait));
        CHECK(qrd_gemm_tn(stream, kw, nc, mk, 1.0, V, ldv, A2, lda, 0.0, Wbuf, kw, slabs, p->slab_cap, T, ldt));
    } else {
        CHECK(qrd_gemm_tn(stream, kw, nc, mk, 1.0, V, ldv, A2, lda, 0.0, Ybuf, kw, slabs, p->slab_cap, NULL, 0));
        if (t_wait) CHECK(qrd_stream_wait_event(stream, t_wait));
        CHECK(qrd_gemm_tn(stream, kw, nc, kw, 1.0, T, ldt, Ybuf, kw, 0.0, Wbuf, kw, NULL, 0, NULL, 0));
    }
    return apply_vw(p, stream, V, ldv, mk, kw, A2, lda, nc, Wbuf, Ybuf);
}

/* A2 (mk x nc) -= V (mk x kw) W (kw x nc): the last step of every small-T application.  Ybuf: kw * nc doubles of scratch (free by now) or NULL. */
static int apply_vw(qr_plan* p, void* stream, const double* V, int ldv, int mk, int kw, double* A2, int lda, int nc, double* Wbuf, double* Ybuf)
{
    (void) p;
    /* Round 5: the update of a tall block through the trailing-update kernel of the square case (qr_gemm_nt.hip: four workgroups per CU,
     * operands HBM -> LDS directly, tiles dealt to the XCDs so that the column tiles of a row block share V in one L2): it wants W
     * transposed (nc x kw, row-fast), one more tiny launch.  MI355XQR_TALL_NT=0: the 8-wave NN kernel as before. */
    if (knobs()->tall_nt && Ybuf && mk >= 16384 && kw >= 32 && kw % 16 == 0 && nc >= 128) {
        /* (round 6: any width -- whole 64-column tiles through the kernel, the last nc % 64 columns through the generic one -- and any even
         * height, gemm_nt4_kernel<.., RAG>: 262144 x 500 took 5.75 ms where 262144 x 512 takes 4.53, nearly all of it here) */
        const int r = nc % 64, ni = nc - r;
        if (ni >= 128 && qrd_gemm_nt4_ok(mk, ni, kw, V, ldv, Ybuf, ni, A2, lda)) {
            if (r > 0) CHECK(qrd_gemm_nn(stream, mk, r, kw, -1.0, V, ldv, Wbuf + (size_t) ni * kw, kw, 1.0, A2 + (size_t) ni * lda, lda));
            CHECK(qrd_transpose(stream, kw, ni, Wbuf, kw, Ybuf, ni));      /* Ybuf (free by now) <- W^T */
            return qrd_gemm_nt(stream, mk, ni, kw, -1, V, ldv, Ybuf, ni, A2, lda, -1, NULL);
        }
    }
    /* tall products go to the 8-wave kernel (4 waves per SIMD keep the C traffic of a K <= 256 update flowing);
     * it hands anything it cannot take (ragged, unaligned) to the generic path itself */
    const int use_w8 = 1;
    /* ... as long as its 128 x 128 tiles give every compute unit of the stream one: the look-ahead update of a late panel
     * (mk <= 8192, 256 columns: 128 tiles for 192-224 CUs) runs 16 serial K steps on half the chip -- 64 x 64 tiles then
     * (8192^2: 29.7 -> 29.2 ms) */
    if (use_w8 && mk >= 2048 && nc >= 128 && (long long) (mk / 128) * (nc / 128) >= qrd_stream_cus(stream))
        return qrd_gemm_nn_update2(stream, mk, nc, kw, -1.0, V, ldv, Wbuf, kw, 1.0, A2, lda);
    return qrd_gemm_nn(stream, mk, nc, kw, -1.0, V, ldv, Wbuf, kw, 1.0, A2, lda);
}

/* Apply panel set e's block reflector to columns [c0, c0+nc):  A2 -= V (T^T (V^T A2)).
 * profile 1 = the wide update on the update stream: V*T is formed first when form_vt is set (class 3) and W = (V T)^T A2 is
 *             one long-K product (class 1), then A2 -= V W (class 0); kernels under their own profiler names;
 * profile 0 = look-ahead update of the next panel (class 4), 2 = the share of the wide update done on the panel CUs
 *             (class 5): Y = V^T A2, W = T^T Y (a small product), A2 -= V W -- no V*T needed.
 * Ybuf: nc*wout doubles of scratch for profile 0 / 2. */
static int update_cols_inner(qr_plan* p, void* stream, int e, double* dA, int lda, int k, int mk, int wout, int c0, int nc,
                             double* Wbuf, double* Ybuf, double* slabs, int profile, int form_vt);

static int update_cols(qr_plan* p, void* stream, int e, double* dA, int lda, int k, int mk, int wout, int c0, int nc,
                       double* Wbuf, double* Ybuf, double* slabs, int profile, int form_vt)
{
    qrd_range_push(profile == 1 ? "mi355xqr wide update" : (profile == 0 ? "mi355xqr look-ahead update" : "mi355xqr panel-stream share"));
    const int rc = update_cols_inner(p, stream, e, dA, lda, k, mk, wout, c0, nc, Wbuf, Ybuf, slabs, profile, form_vt);
    qrd_range_pop();
    return rc;
}

static int update_cols_inner(qr_plan* p, void* stream, int e, double* dA, int lda, int k, int mk, int wout, int c0, int nc,
                             double* Wbuf, double* Ybuf, double* slabs, int profile, int form_vt)
{
    double* A2 = dA + (size_t) c0 * lda + k;
    const int ldv = p->ldv;
    if (nc <= 0) return 0;
    if (profile == 1 && (size_t) nc * 8 <= (size_t) mk && (Ybuf || p->Ye2)) {
        /* tall-skinny: forming V*T (mk x wout, 16 mk wout bytes, 2 mk wout^2 flops) would cost more than the few columns it is
         * applied to; apply T to the small product instead (262144 x 512: 1.1 GB less HBM traffic per factorisation) */
        CHECK(prof_begin_on(p, 3, stream));
        /* a parked panel (cq_parked): its V is the panel itself in the caller's array */
        const double* Vp = p->cq_parked ? dA + (size_t) k * lda + k : p->Vw2[e];
        CHECK(apply_small_t(p, stream, Vp, p->cq_parked ? lda : ldv, p->T2[e], p->ldt, mk, wout, A2, lda, nc, Wbuf, Ybuf ? Ybuf : p->Ye2, slabs));
        CHECK(prof_end(p, 4.0 * mk * (double) nc * wout, 24.0 * mk * (double) nc + 16.0 * mk * wout));
        return 0;
    }
    /* (round 6: small wide updates -- 512^2 ... 4096^2 at nb 64 / 128 -- through the T-folded reduction instead, three launches and no V*T:
     * within 0.5 % at every threshold; these updates are not launch-bound.  profiles/NOTES.md) */
    if (profile == 1 && nc % 64 != 0 && nc >= 192 && wout >= 64) {
        /* ragged width: the last nc % 64 columns through the generic kernels, the rest as whole 64-column tiles of the update kernel
         * (whose bottom row tile may be ragged: gemm_nt4_kernel<.., RAG>) -- where that part would take the kernel at all */
        const int r = nc % 64, ni = nc - r;
        if ((wout >= 128 || (long long) mk * ni >= 8388608LL) && qrd_gemm_nt4_ok(mk, ni, wout, p->Vw2[e], ldv, Wbuf, ni, A2, lda)) {
            CHECK(update_cols_inner(p, stream, e, dA, lda, k, mk, wout, c0 + ni, r, Wbuf, Ybuf, slabs, profile, form_vt));
            return update_cols_inner(p, stream, e, dA, lda, k, mk, wout, c0, ni, Wbuf, Ybuf, slabs, profile, 0);
        }
    }
    /* (half tiles -- mk or nc = 64 mod 128: every other step at nb 64, every step of a matrix whose height is 64 mod 128 -- go to the
     * four-workgroup kernel too (round 6): always at K >= 128 (4032^2 at nb 128 23.8 -> 19.4 ms, 4160^2 at nb 256 18.0 -> 15.9), at K = 64 from
     * 8 M elements of trailing matrix on (4096^2 at nb 64 9.59 -> 9.09, 8192^2 73.7 -> 68.5; below, the generic kernels' 64 x 64 tiles fill
     * the chip better on a product that is HBM-bound anyway: 2048^2 3.68 against 3.82 through this route; not at K = 32, where the generic
     * kernels win outright: 16384^2 at nb 32 296 against 324 ms).  profiles/r06_nt4_half_tiles.txt) */
    if (profile == 1 && (qrd_gemm_nt_ok(mk, nc, wout, p->Vw2[e], ldv, Wbuf, nc, A2, lda) ||
                         ((wout >= 128 || (wout >= 64 && (long long) mk * nc >= 8388608LL)) && qrd_gemm_nt4_ok(mk, nc, wout, p->Vw2[e], ldv, Wbuf, nc, A2, lda)))) {
        /* second-generation wide update: W kept transposed (Wt = A2^T (V T), nc x wout), so that both operands of
         * A2 -= V Wt^T are row-fast and go HBM -> LDS directly (qr_gemm_nt.hip) */
        if (form_vt || !p->vt_formed[e]) {     /* on demand: an earlier slice of this update may have taken the tall-skinny shortcut above */
            CHECK(prof_begin_on(p, 3, stream));
            CHECK(qrd_gemm_nn(stream, mk, wout, wout, 1.0, p->Vw2[e], ldv, p->T2[e], p->ldt, 0.0, p->VT2[e], ldv));
            CHECK(prof_end(p, 2.0 * mk * (double) wout * wout, 16.0 * mk * wout));
            p->vt_formed[e] = 1;
        }
        const int cw = nc;       /* (walking A2 in cache-sized column chunks was measured and removed: DESIGN section 8, round 2) */
        for (int c = 0; c < nc; c += cw) {
            const int w1 = imin(cw, nc - c);
            double* A2c = A2 + (size_t) c * lda;
            CHECK(prof_begin_on(p, 1, stream));
            CHECK(qrd_gemm_tn_update(stream, w1, wout, mk, 1.0, A2c, lda, p->VT2[e], ldv, 0.0, Wbuf, w1, slabs, p->slab_cap));
            CHECK(prof_end(p, 2.0 * mk * (double) w1 * wout, 8.0 * mk * ((double) w1 + wout)));
            CHECK(prof_begin_on(p, 0, stream));
            CHECK(qrd_gemm_nt(stream, mk, w1, wout, -1, p->Vw2[e], ldv, Wbuf, w1, A2c, lda, -1, NULL));
            CHECK(prof_end(p, 2.0 * mk * (double) w1 * wout, 16.0 * mk * (double) w1 + 8.0 * mk * wout));
        }
        return 0;
    }
    if (profile == 1) {
        const int tagged = wout >= 128 && nc >= 128;
        if (form_vt || !p->vt_formed[e]) {     /* on demand: an earlier slice of this update may have taken the tall-skinny shortcut above */
            CHECK(prof_begin_on(p, 3, stream));
            CHECK(qrd_gemm_nn(stream, mk, wout, wout, 1.0, p->Vw2[e], ldv, p->T2[e], p->ldt, 0.0, p->VT2[e], ldv));
            CHECK(prof_end(p, 2.0 * mk * (double) wout * wout, 16.0 * mk * wout));
            p->vt_formed[e] = 1;
        }
        CHECK(prof_begin_on(p, 1, stream));
        if (tagged)
            CHECK(qrd_gemm_tn_update(stream, wout, nc, mk, 1.0, p->VT2[e], ldv, A2, lda, 0.0, Wbuf, wout, slabs, p->slab_cap));
        else
            CHECK(qrd_gemm_tn(stream, wout, nc, mk, 1.0, p->VT2[e], ldv, A2, lda, 0.0, Wbuf, wout, slabs, p->slab_cap, NULL, 0));
        CHECK(prof_end(p, 2.0 * mk * (double) nc * wout, 8.0 * mk * ((double) nc + wout)));
        CHECK(prof_begin_on(p, 0, stream));
        if (tagged)
            CHECK(qrd_gemm_nn_update(stream, mk, nc, wout, -1.0, p->Vw2[e], ldv, Wbuf, wout, 1.0, A2, lda));
        else
            CHECK(qrd_gemm_nn(stream, mk, nc, wout, -1.0, p->Vw2[e], ldv, Wbuf, wout, 1.0, A2, lda));
        CHECK(prof_end(p, 2.0 * mk * (double) nc * wout, 16.0 * mk * (double) nc + 8.0 * mk * wout));
        return 0;
    }
    CHECK(prof_begin_on(p, profile ? 5 : 4, stream));
    CHECK(apply_small_t(p, stream, p->Vw2[e], ldv, p->T2[e], p->ldt, mk, wout, A2, lda, nc, Wbuf, Ybuf, slabs));
    CHECK(prof_end(p, 4.0 * mk * (double) nc * wout, 24.0 * mk * (double) nc + 16.0 * mk * wout));
    return 0;
}

/* Columns of the wide update that the panel stream takes over after it has factored the next panel, so that both
 * streams finish step s together:  tc + F x / Rp = F (nwide - x) / Ru  with F = 4 mk nb flops per column, tc the
 * modelled time of the next panel chain, Rp / Ru the GEMM rates of the panel CUs / the update CUs. */
/* modelled time (ms) of the chain the panel stream runs before it can take a share of / while the update stream runs W(s): the next
 * panel (mk - wout rows) plus its look-ahead update.  Linear in the height where the panel is a launch chain on the masked stream;
 * FLAT where the next panel is one launch (<= 8192 rows: qr_panel_fused.hip, 0.58-0.65 ms at 256 columns whatever the height, round 5
 * gantt) -- the linear model, fitted to the leaf chain, said 1.4 ms there and declared the factorisation chain-bound nine steps early */
static double chain_ms(const qr_plan* p, int mk, int wout)
{
    if (p->bal_tail_tc > 0.0 && p->pf_ws && !p->fused_off && mk - wout <= 8192 && mk - wout >= knobs()->fused_min_rows && wout <= QR_HALF)
        return p->bal_tail_tc * (double) wout / 256.0;
    return (p->bal_tc0 + p->bal_tc1 * (double) mk / 16384.0) * (double) wout / 256.0;
}

static int balance_cols(const qr_plan* p, int mk, int wout, int nwide)
{
    if (!p->npairs || p->bal_rp <= 0.0 || nwide <= 0) return 0;
    const double F = 4.0 * mk * (double) wout * 1e-9;                  /* GFLOP per column */
    /* (round 6, run 32: a chain time of its own for this decision -- the 64-CU panel stream idles 0.4-0.85 ms per step once its tall panels are
     * one launch -- gains nothing: a larger share costs the update stream what it saves; profiles/NOTES.md) */
    const double tc = chain_ms(p, mk, wout);                           /* ms */
    const double x = (F * nwide / p->bal_ru - tc) / (F * (1.0 / p->bal_rp + 1.0 / p->bal_ru));   /* rates in GFLOP/ms = TFLOP/s */
    int xi = (int) x;
    xi -= xi % 128;
    if (xi < 128) return 0;
    return xi > nwide ? nwide : xi;
}

/* 1 when the wide update W(s) is predicted to outlast the next panel's chain (same model as balance_cols): the update
 * stream is then busy back to back and the look-ahead update N(s) is better off on the panel stream, concurrent with the
 * start of W(s); in the chain-bound phase the update stream has nothing else to do and runs N(s) 3x faster than the panel
 * stream's few CUs would. */
static int update_bound(const qr_plan* p, int mk, int wout, int nwide)
{
    if (!p->npairs || p->bal_ru <= 0.0 || nwide <= 0) return 0;
    const double F = 4.0 * mk * (double) wout * 1e-9;
    const double tc = chain_ms(p, mk, wout);
    return F * nwide / p->bal_ru > tc;
}

/* The T merge a one-launch panel left to its caller (factor_panel with defer_hint): G = V^T V and the merge tree, for panel set e, on
 * `stream`.  In the chain-bound phase of a look-ahead factorisation the critical chain is  panel -> Gram -> merge -> look-ahead update ->
 * next panel;  on the panel stream's 32 compute units the Gram product of a 4096 x 256 V takes 85-90 us (its 0.4 GFLOP are matrix-core
 * time there), on the idle update stream's 224 it takes 22, and the hop between the streams in front of the look-ahead update goes too
 * (profiles/r06_c3_tail_kernels.txt). */
static int deferred_gram(qr_plan* p, void* stream, int e, int mk, int wout, double* slabs)
{
#ifdef QR_TRACE_DEFER
    fprintf(stderr, "deferred_t_merge: mk %d wout %d on %s\n", mk, wout, stream == p->stream_u ? "update stream" : "panel stream");
#endif
    return qrd_gemm_tn(stream, wout, wout, mk, 1.0, p->Vw2[e], p->ldv, p->Vw2[e], p->ldv, 0.0, p->G, p->nb, slabs, p->slab_cap, NULL, 0);
}
static int deferred_tree(qr_plan* p, void* stream, int e, int wout, const double* tau_k)
{
    return qrd_larft(stream, wout, p->ib, p->G, p->nb, tau_k, p->T2[e], p->ldt, NULL, 0, p->X, p->nb);
}
static int deferred_t_merge(qr_plan* p, void* stream, int e, int mk, int wout, const double* tau_k, double* slabs)
{
    CHECK(deferred_gram(p, stream, e, mk, wout, slabs));
    return deferred_tree(p, stream, e, wout, tau_k);
}

/* The look-ahead update N(s) of a panel whose T merge was deferred, WITHOUT the merged T (round 6): A_next -= V W with W = T^T (V^T A_next)
 * from the forward substitution over the leaves (qrd_trsm_gt: the panel's Gram matrix and the leaves' own T blocks suffice), so that the
 * merge tree -- six dependent launches at 256 columns, 35 us -- runs BEHIND N(s), off the chain P(s) -> N(s) -> P(s+1).  Returns 1
 * when the shape is not whole leaves x whole 16-column tiles (the caller merges first and applies T as usual). */
static int lookahead_update_no_t(qr_plan* p, void* stream, int e, double* dA, int lda, int k, int mk, int wout, int nc, double* slabs)
{
    double* A2 = dA + (size_t) (k + wout) * lda + k;
    if (wout % 32 || wout > 256 || nc % 16 || nc < 16 || p->ib != 32) return 1;
    CHECK(prof_begin_on(p, 4, stream));
    CHECK(qrd_gemm_tn(stream, wout, nc, mk, 1.0, p->Vw2[e], p->ldv, A2, lda, 0.0, p->Yn, wout, slabs, p->slab_cap, NULL, 0));
    CHECK(qrd_trsm_gt(stream, wout, nc, p->G, p->nb, p->T2[e], p->ldt, p->Yn, wout, p->Wn, wout));
    CHECK(apply_vw(p, stream, p->Vw2[e], p->ldv, mk, wout, A2, lda, nc, p->Wn, p->Yn));
    CHECK(prof_end(p, 4.0 * mk * (double) nc * wout, 24.0 * mk * (double) nc + 16.0 * mk * wout));
    return 0;
}

static int geqrf_issue(qr_plan* p, double* dA, int m, int n, int lda, double* dtau);

int qr_geqrf_dev(qr_plan* p, double* dA, int m, int n, int lda, double* dtau)
{
    if (!p || !dA || !dtau || n < 1 || m < n || m > p->m || n > p->n || lda < m) return QR_E_ARG;
    if (!p->padA && m == p->m_user && n == p->n_user && m >= 512 && !p->use_graph && !p->pad_failed && ((lda & 1) || ((uintptr_t) dA & 15)) &&
        (double) p->m * p->n * 8.0 <= 4294967296.0) {
        /* an odd leading dimension or a misaligned array under an otherwise aligned height: the same copy, allocated on first need */
        if (qrd_malloc((void**) &p->padA, sizeof(double) * (size_t) p->m * p->n)) { p->padA = NULL; p->pad_failed = 1; }
    }
    if (p->padA && m == p->m_user && n == p->n_user && !p->use_graph) {
        /* zero rows / unit columns appended (plan_create_impl): copy in, factor the padded shape, copy the caller's m x n back -- all on the plan's stream */
        const int mp = p->m, np = p->n;
        CHECK(qrd_memset(p->stream, p->padA, 0, sizeof(double) * (size_t) mp * np));
        CHECK(qrd_copy_block(p->stream, dA, lda, p->padA, mp, m, n));
        if (np > n) CHECK(qrd_set_identity(p->stream, p->padA + (size_t) n * mp + m, mp, np - n, np - n, 0));   /* e_{m + j} in appended column j (rows m .. are zero everywhere else) */
        CHECK(geqrf_issue(p, p->padA, mp, np, mp, np > n ? p->pad_tau : dtau));
        if (np > n) CHECK(qrd_d2d(p->stream, dtau, p->pad_tau, sizeof(double) * (size_t) n));
        return qrd_copy_block(p->stream, p->padA, mp, dA, lda, m, n);
    }
    if (!p->use_graph || p->prof_on) return geqrf_issue(p, dA, m, n, lda, dtau);
    if (!(p->graph_exec && p->g_dA == dA && p->g_dtau == dtau && p->g_m == m && p->g_n == n && p->g_lda == lda)) {
        CHECK(qr_plan_sync(p));
        qrd_graph_destroy(p->graph_exec);
        p->graph_exec = NULL;
        CHECK(qrd_capture_begin(p->stream));
        int rc = geqrf_issue(p, dA, m, n, lda, dtau);
        void* exec = NULL;
        int rc2 = qrd_capture_end(p->stream, &exec);
        if (rc || rc2) { qrd_graph_destroy(exec); return rc ? rc : rc2; }
        p->graph_exec = exec;
        p->g_dA = dA; p->g_dtau = dtau; p->g_m = m; p->g_n = n; p->g_lda = lda;
    }
    return qrd_graph_launch(p->graph_exec, p->stream);
}

/* CU-partition phase for a step with `remaining` of n columns still to factor */
static int phase_of(const qr_plan* p, int remaining, int n)
{
    int i = 0;
    while (i + 1 < p->npairs && (double) remaining <= p->pair_until[i] * n) ++i;
    return p->npairs ? i : -1;
}

/* Move the critical path (and the wide-update stream) to phase i's streams; -1 = back to the public stream.
 * Everything queued so far on the old streams is ordered before anything queued later on the new ones. */
static int enter_phase(qr_plan* p, int i)
{
    if (p->npairs == 0 || i == p->pair_cur) return 0;
    void* ns = i < 0 ? p->s_main : p->s_pair[i][0];
    void* nu = i < 0 ? NULL : p->s_pair[i][1];
    CHECK(qrd_event_record(p->ev_hop[0], p->stream));
    CHECK(qrd_stream_wait_event(ns, p->ev_hop[0]));
    if (p->stream_u && p->stream_u != nu) {
        CHECK(qrd_event_record(p->ev_hop[1], p->stream_u));
        CHECK(qrd_stream_wait_event(nu ? nu : ns, p->ev_hop[1]));
    }
    p->stream = ns; p->stream_u = nu; p->pair_cur = i;
    if (i >= 0 && p->bal_auto) {       /* the load-balance model follows the phase's partition (chain time: shorter on more CUs) */
        int cus = 256;
        qrd_device_info(NULL, 0, &cus, NULL, NULL);
        const int pc = qrd_stream_cus(ns);
        const int uc = nu ? qrd_stream_cus(nu) : cus;
        p->bal_rp = 0.22 * (pc < cus ? pc : 32);
        p->bal_ru = 0.23 * uc;
        const double f = pc <= 32 ? 1.0 : (pc <= 64 ? 0.8 : 0.7);
        p->bal_tc0 = p->bal_tc0_base * f; p->bal_tc1 = p->bal_tc1_base * f;
    }
    return 0;
}

static int geqrf_issue_inner(qr_plan* p, double* dA, int m, int n, int lda, double* dtau);

/* a failure in the middle of the schedule must not leave the plan pointing at a phase's masked streams */
static int geqrf_issue(qr_plan* p, double* dA, int m, int n, int lda, double* dtau)
{
    const int rc = geqrf_issue_inner(p, dA, m, n, lda, dtau);
    if (rc) {
        p->stream = p->s_main;
        p->stream_u = p->npairs ? NULL : p->stream_u;
        p->pair_cur = -1;
        use_set(p, 0);
    }
    return rc;
}

static int geqrf_issue_inner(qr_plan* p, double* dA, int m, int n, int lda, double* dtau)
{
    const int nb = p->nb;
    p->cq_parked = 0;             /* (a factorisation that ended in an error may have left a panel parked in ANOTHER array: never restore into that) */
    p->park_hint = 0;
    p->defer_hint = 0;
    p->t_deferred = 0;
    if (!p->lookahead) {
        use_set(p, 0);
        for (int k = 0, wout = 0; k < n; k += wout) {
            wout = imin(nb, n - k);
            /* a ragged last panel (width not a multiple of the leaf width): its whole leaves first, as a panel of their own -- they keep the
             * one-launch / full-width routes, which take whole leaves only -- and the few columns left over as one more (65536 x 500: the
             * 116-column last panel went leaf by leaf, 2.34 ms against 1.92 for 65536 x 512; profiles/r06_odd_sizes.txt) */
            if (p->ib == 32 && wout > 32 && wout % 32 != 0) wout -= wout % 32;
            const int mk = m - k, nt = n - (k + wout);
            CHECK(prof_begin(p, 2));
            /* (what follows decides whether a full-width panel may park its R: only the update that applies T to the small product reads V
             * through a pointer of its own -- and behind the last panel nothing reads V at all: R goes back at once) */
            p->park_hint = nt == 0 || ((size_t) nt * 8 <= (size_t) mk && p->Ye2 != NULL);
            CHECK(factor_panel(p, dA, m, lda, k, wout, dtau, nt > 0, NULL));
            CHECK(prof_end(p, 2.0 * mk * (double) wout * wout, 16.0 * mk * wout));
            if (nt > 0) CHECK(update_cols(p, p->stream, 0, dA, lda, k, mk, wout, k + wout, nt, p->W, NULL, p->slabs, 1, 1));
            CHECK(cq_unpark(p));
        }
        return 0;
    }
    /* Look-ahead (depth 1).  Step s: P(s) factor panel s, N(s) update the FIRST HALF (<= 256 columns) of the next panel,
     * W(s) update everything to the right of that, in two pieces: W_a(s) the second half of the next panel (two-level
     * panels only; ev_half when done), then W_b(s) the rest.  Critical chain P(s) -> N(s) -> P(s+1) runs on p->stream
     * (or, for N(s) in the chain-bound phase, on the update stream); W(s) runs on p->stream_u concurrently with P(s+1).
     *   N(s)  needs P(s) (stream order / ev_panel) and W(s-1) (ev_wide[(s-1)&1] / stream order);
     *   W(s)  needs P(s) (ev_panel[s&1]) and W(s-1) (stream order);
     *   P(s+1)'s second half needs W_a(s) (ev_half[s&1]);
     *   P(s+2) overwrites panel set s&1, which W(s) reads: ordered through N(s+1)'s wait on ev_wide[s&1]. */
    int wide_pending[2] = {0, 0}, extra_pending = 0;
    int v_recorded[2] = {0, 0};     /* ev_v[e] was recorded inside factor_panel for the panel now in set e */
    int t_deferred[2] = {0, 0};     /* the panel now in set e left its T merge to this loop (deferred_t_merge) */
    const int split_t = knobs()->split_t;
    /* Early look-ahead update (update-bound phase, one-level panels): N(s+1) is issued on the panel stream right after P(s+1),
     * ahead of the panel stream's share E(s) of the wide update, instead of on the update stream between W(s) and W(s+1) --
     * there it waited for ALL of W(s) and then ran alone on the chip for 0.16-0.18 ms per step (C3: 33 such steps).  For
     * that, W(s) starts with the columns of panel s+2 (W1(s), event ev_half[s&1]) and E(s) takes the LAST columns of the wide
     * range instead of the first.  early_done: N of the coming step has been issued already. */
    const int early_env = knobs()->early_next, early_w1 = QR_EARLY_W1;
    int early_done = 0;
    /* P(0) has nothing to overlap with: it runs on the plan's public stream -- every compute unit -- and the CU partition starts behind it
     * (round 6; on the panel stream's 32 CUs the first panel of C3 took 1.29 ms with the other 224 idle) */
    {
        const int w0 = imin(nb, n);
        use_set(p, 0);
        CHECK(prof_begin(p, 2));
        p->v_ready = (split_t && p->npairs > 0 && w0 <= QR_HALF && n > w0) ? p->ev_v[0] : NULL;
        v_recorded[0] = p->v_ready != NULL;
        CHECK(factor_panel(p, dA, m, lda, 0, w0, dtau, n > w0, NULL));
        p->v_ready = NULL;
        CHECK(prof_end(p, 2.0 * m * (double) w0 * w0, 16.0 * m * w0));
        CHECK(qrd_event_record(p->ev_panel[0], p->stream));
    }
    CHECK(enter_phase(p, phase_of(p, n, n)));
    int s = 0;
    for (int k = 0; k < n; k += nb, ++s) {
        const int e = s & 1, wout = imin(nb, n - k), mk = m - k, nt = n - (k + wout);
        if (nt <= 0) break;
        const int wnext = imin(nb, nt), nwide = nt - wnext;
        const int nfirst = imin(QR_HALF, wnext), nhalf2 = wnext - nfirst;    /* N(s) covers nfirst columns, W_a(s) the other nhalf2 */
        const int phase_now = phase_of(p, nt, n);
        const int n_early = early_done;                                      /* N(s) went out in the previous iteration */
        early_done = 0;
        CHECK(enter_phase(p, phase_now));
        int extra = p->We ? balance_cols(p, mk, wout, nwide) : 0;
        const int k1 = k + wout, mk1 = m - k1, nt1 = n - (k1 + wnext);                                        /* P(s+1) */
        const int wnext2 = nt1 > 0 ? imin(nb, nt1) : 0;                      /* width of panel s+2 */
        /* N(s+1) early?  Only while the wide update outlasts the panel chain (now and in the coming step), with the CU
         * partition on, one-level panels, and no change of partition phase in between (the panel stream must stay the same) */
        const int early_next = early_env && p->npairs > 0 && p->stream_u != NULL && nhalf2 == 0 && wnext2 > 0 && wnext2 <= QR_HALF &&
                               nwide > wnext2 && update_bound(p, mk, wout, nwide) &&
                               update_bound(p, mk1, wnext, nt1 - wnext2) && phase_of(p, nt1, n) == phase_now;
        if (early_next && extra > nwide - wnext2) { extra = nwide - wnext2; extra -= extra % 128; }
        const int n_on_u = !n_early && p->stream_u != NULL && p->npairs > 0 &&
                           (p->next_on_update == 1 || (p->next_on_update == 2 && !update_bound(p, mk, wout, nwide)));
        if (t_deferred[e] && !n_on_u) {
            /* (never with the defaults: the deferral below predicts n_on_u with the same rule.  A knob that moves N(s) elsewhere: merge on
             * the panel stream now, and let ev_panel say "T complete" again -- nobody has waited for it yet) */
            CHECK(deferred_t_merge(p, p->stream, e, mk, wout, dtau + k, p->slabs));
            CHECK(qrd_event_record(p->ev_panel[e], p->stream));
            t_deferred[e] = 0;
        }
        if (n_early) {
            /* nothing: N(s) sits on the panel stream behind P(s) */
        } else if (n_on_u && t_deferred[e]) {
            /* chain-bound phase, P(s) was one launch: its Gram matrix and T merge run HERE, on the update stream (idle by now), then N(s) */
            CHECK(qrd_stream_wait_event(p->stream_u, v_recorded[e] ? p->ev_v[e] : p->ev_panel[e]));
            if (extra_pending) CHECK(qrd_stream_wait_event(p->stream_u, p->ev_extra[e ^ 1]));
            CHECK(prof_begin_on(p, 3, p->stream_u));          /* (class 3, misc: a class-2 record is "one panel" to the readers of the records) */
            CHECK(deferred_gram(p, p->stream_u, e, mk, wout, p->slabs_u));
            CHECK(prof_end(p, 2.0 * mk * (double) wout * wout, 8.0 * mk * wout));
            t_deferred[e] = 0;
            /* N(s) from the Gram matrix and the leaves' T blocks, the merge tree behind it (MI355XQR_TRSM=0 in the lab build: tree first) */
            int need_tree_first = !knobs()->trsm_next;
            if (!need_tree_first) {
                const int rc = lookahead_update_no_t(p, p->stream_u, e, dA, lda, k, mk, wout, nfirst, p->slabs_u);
                if (rc < 0 || rc > 1) return rc;
                need_tree_first = rc == 1;
            }
            if (need_tree_first) {
                CHECK(deferred_tree(p, p->stream_u, e, wout, dtau + k));
                CHECK(update_cols(p, p->stream_u, e, dA, lda, k, mk, wout, k + wout, nfirst, p->Wn, p->Yn, p->slabs_u, 0, 0));
            }
            CHECK(qrd_event_record(p->ev_next[e], p->stream_u));
            CHECK(qrd_stream_wait_event(p->stream, p->ev_next[e]));
            if (!need_tree_first) CHECK(deferred_tree(p, p->stream_u, e, wout, dtau + k));      /* W(s), next on this stream, needs the merged T */
            wide_pending[e ^ 1] = 0;
        } else if (n_on_u) {
            /* N(s) on the update stream: behind W(s-1) by stream order, after P(s) (ev_panel) and E(s-1) (ev_extra) */
            if (v_recorded[e]) {        /* V^T A_next may start as soon as V is complete; T^T (.) waits for the T merge */
                CHECK(qrd_stream_wait_event(p->stream_u, p->ev_v[e]));
                p->t_wait = p->ev_panel[e];
            } else
                CHECK(qrd_stream_wait_event(p->stream_u, p->ev_panel[e]));
            if (extra_pending) CHECK(qrd_stream_wait_event(p->stream_u, p->ev_extra[e ^ 1]));
            CHECK(update_cols(p, p->stream_u, e, dA, lda, k, mk, wout, k + wout, nfirst, p->Wn, p->Yn, p->slabs_u, 0, 0));
            p->t_wait = NULL;
            CHECK(qrd_event_record(p->ev_next[e], p->stream_u));
            CHECK(qrd_stream_wait_event(p->stream, p->ev_next[e]));
            wide_pending[e ^ 1] = 0;                 /* W(s-1) is ordered before N(s), hence before everything the panel stream does next */
        } else {
            if (s > 0 && wide_pending[e ^ 1]) {
                CHECK(qrd_stream_wait_event(p->stream, p->ev_wide[e ^ 1]));
                wide_pending[e ^ 1] = 0;
            }
            CHECK(update_cols(p, p->stream, e, dA, lda, k, mk, wout, k + wout, nfirst, p->Wn, p->Yn, p->slabs, 0, 0));   /* N(s) */
        }
        const int wide_cols = nhalf2 + nwide - extra;
        const int cw = k + wout + wnext;                                     /* first column of the wide range */
        if (wide_cols > 0) {                                                                                  /* W(s) */
            if (!n_on_u) {
                CHECK(qrd_stream_wait_event(p->stream_u, p->ev_panel[e]));
                if (extra_pending) CHECK(qrd_stream_wait_event(p->stream_u, p->ev_extra[e ^ 1]));   /* E(s-1) wrote columns W(s) reads */
            }
            int formed = 0;
            if (nhalf2 > 0) {                                                                                 /* W_a(s) */
                CHECK(update_cols(p, p->stream_u, e, dA, lda, k, mk, wout, k + wout + nfirst, nhalf2, p->W, NULL, p->slabs_u, 1, 1));
                CHECK(qrd_event_record(p->ev_half[e], p->stream_u));
                formed = 1;
            }
            if (early_next) {
                /* W1(s): a first slice that holds the columns of panel s+2, so that N(s+1) need not wait for the rest; E(s) at the
                 * far end.  The slice is ~4096 columns, not just the 256 N(s+1) needs: a 256-column launch pair fills half the
                 * update stream's workgroup slots (0.14 ms for 0.07 ms of work, every step), and N(s+1) has P(s+1)'s 1.4 ms to spare
                 * (16384^2: 1024 / 2048 / 4096 / 6144 / 8192 columns: 125.1 / 124.9 / 124.3 / 131.0 / 143.7 ms) */
                int w1 = nwide - extra;
                {
                    /* ... but W1(s) must be over well before P(s+1) is (N(s+1) waits for it): at most 0.7 of the panel's measured
                     * time 0.62 + 0.9 mk / 16384 ms (nb = 256) at the update stream's rate, and never more than early_w1 columns */
                    const int one_launch = p->bal_tail_tc > 0.0 && p->pf_ws && !p->fused_off && mk1 <= 8192 && mk1 >= knobs()->fused_min_rows;
                    const double pest = (one_launch ? 0.62 : 0.62 + 0.9 * (double) mk1 / 16384.0) * (double) wnext / 256.0;   /* ms */
                    const double fcol = 4.0 * mk * (double) wout * 1e-9;                                         /* GFLOP per column */
                    int cap = (int) (0.7 * pest * p->bal_ru / fcol);
                    cap -= cap % 128;
                    if (cap > early_w1) cap = early_w1;
                    if (cap < wnext2) cap = wnext2;
                    if (w1 > cap + 1024) w1 = cap;
                }
                CHECK(update_cols(p, p->stream_u, e, dA, lda, k, mk, wout, cw, w1, p->W, NULL, p->slabs_u, 1, 1));
                CHECK(qrd_event_record(p->ev_half[e], p->stream_u));
                if (nwide - w1 - extra > 0)
                    CHECK(update_cols(p, p->stream_u, e, dA, lda, k, mk, wout, cw + w1, nwide - w1 - extra, p->W, NULL,
                                      p->slabs_u, 1, 0));
            } else if (nwide - extra > 0)                                                                     /* W_b(s) */
                CHECK(update_cols(p, p->stream_u, e, dA, lda, k, mk, wout, cw + extra, nwide - extra, p->W,
                                  NULL, p->slabs_u, 1, !formed));
            CHECK(qrd_event_record(p->ev_wide[e], p->stream_u));
            wide_pending[e] = 1;
        }
        extra_pending = 0;
        if (wide_pending[e ^ 1]) {           /* P(s+1) overwrites panel set e^1, which W(s-1) reads (only still pending after an early N(s)) */
            CHECK(qrd_stream_wait_event(p->stream, p->ev_wide[e ^ 1]));
            wide_pending[e ^ 1] = 0;
        }
        use_set(p, e ^ 1);
        CHECK(prof_begin(p, 2));
        p->v_ready = (split_t && p->npairs > 0 && wnext <= QR_HALF && nt1 > 0) ? p->ev_v[e ^ 1] : NULL;
        v_recorded[e ^ 1] = p->v_ready != NULL;
        {
            /* Leave P(s+1)'s Gram matrix and T merge to the update stream?  Only where that stream will be idle when P(s+1) ends -- W(s),
             * issued above, shorter than the one-launch panel (0.45 ms per 256 columns; the update's modelled rate is optimistic in the
             * tail, hence the margin) -- and N(s+1) is going to run there anyway (the n_on_u rule of the next iteration, evaluated now) */
            const int nwide1 = nt1 - wnext2;
            const double w_ms = 4.0 * mk * (double) wout * (double) wide_cols * 1e-9 / (p->bal_ru > 0.0 ? p->bal_ru : 50.0);
            const int n_on_u1 = !early_next && p->stream_u != NULL && p->npairs > 0 &&
                                (p->next_on_update == 1 || (p->next_on_update == 2 && !update_bound(p, mk1, wnext, nwide1)));
            p->defer_hint = knobs()->defer_t && n_on_u1 && nt1 > 0 && wnext <= QR_HALF && w_ms < 0.75 * 0.45 * (double) wnext / 256.0 &&
                            phase_of(p, nt1, n) == phase_now;
        }
        CHECK(factor_panel(p, dA, m, lda, k1, wnext, dtau, nt1 > 0, nhalf2 > 0 ? p->ev_half[e] : NULL));
        p->defer_hint = 0;
        t_deferred[e ^ 1] = p->t_deferred;
        p->t_deferred = 0;
        p->v_ready = NULL;
        CHECK(prof_end(p, 2.0 * mk1 * (double) wnext * wnext, 16.0 * mk1 * wnext));
        CHECK(qrd_event_record(p->ev_panel[e ^ 1], p->stream));
        if (early_next) {                                                                                     /* N(s+1), early */
            CHECK(qrd_stream_wait_event(p->stream, p->ev_half[e]));                                           /* W1(s) */
            CHECK(update_cols(p, p->stream, e ^ 1, dA, lda, k1, mk1, wnext, k1 + wnext, imin(QR_HALF, wnext2), p->Wn, p->Yn, p->slabs, 0, 0));
            early_done = 1;
        }
        if (extra > 0) {                                                                                      /* E(s) */
            CHECK(update_cols(p, p->stream, e, dA, lda, k, mk, wout, early_next ? cw + nwide - extra : cw, extra, p->We, p->Ye,
                              p->slabs, 2, 0));
            CHECK(qrd_event_record(p->ev_extra[e], p->stream));
            extra_pending = 1;
        }
    }
    for (int e = 0; e < 2; ++e)
        if (wide_pending[e]) CHECK(qrd_stream_wait_event(p->stream, p->ev_wide[e]));
    use_set(p, 0);
    return enter_phase(p, -1);
}

int qr_applyq_dev(qr_plan* p, const double* dA, int m, int n, int lda, const double* dtau, double* dC, int ccols,
                  int ldc, int identity_start)
{
    if (!p || !dA || !dtau || !dC || n < 1 || m < n || m > p->m || n > p->n || ccols < 1 || ldc < m) return QR_E_ARG;
    const int nb = p->nb, ib = p->ib, ldv = p->ldv, ldt = p->ldt;
    CHECK(ensure_w(p, (size_t) nb * ccols));
    use_set(p, 0);
    if (identity_start) CHECK(qrd_set_identity(p->stream, dC, ldc, m, ccols, 0));
    const int npan = (n + nb - 1) / nb;
    for (int pi = npan - 1; pi >= 0; --pi) {
        const int k = pi * nb, wout = imin(nb, n - k), mk = m - k;
        const int c0 = identity_start ? k : 0, nc = ccols - c0;
        if (nc <= 0) continue;
        const double* Ak = dA + (size_t) k * lda + k;
        CHECK(qrd_extract_v(p->stream, Ak, lda, mk, wout, p->Vw, ldv));
        CHECK(tn(p, wout, wout, mk, p->Vw, ldv, p->Vw, ldv, p->G, nb, NULL));
        CHECK(qrd_larft(p->stream, wout, ib, p->G, nb, dtau + k, p->T, ldt, p->Tt, 1, p->X, nb));
        CHECK(qrd_gemm_nn(p->stream, mk, wout, wout, 1.0, p->Vw, ldv, p->Tt, ldt, 0.0, p->VT, ldv));   /* V T^T */
        double* Cs = dC + (size_t) c0 * ldc + k;
        CHECK(tn(p, wout, nc, mk, p->VT, ldv, Cs, ldc, p->W, wout, NULL));                              /* T V^T C */
        CHECK(qrd_gemm_nn(p->stream, mk, nc, wout, -1.0, p->Vw, ldv, p->W, wout, 1.0, Cs, ldc));
    }
    return 0;
}

int qr_extract_r_dev(qr_plan* p, const double* dA, int m, int n, int lda, double* dR, int rrows, int ldr)
{
    if (!p || !dA || !dR || rrows < 1 || ldr < rrows) return QR_E_ARG;
    return qrd_extract_r(p->stream, dA, lda, m, n, dR, ldr, rrows);
}

int qr_fill_uniform_dev(qr_plan* p, double* dA, int lda, long long rows, int cols, long long row_off,
                        long long total_rows, unsigned long long seed)
{
    if (!p || !dA) return QR_E_ARG;
    return qrd_fill_uniform(p->stream, dA, lda, rows, cols, row_off, total_rows, seed);
}

double qr_uniform_at(unsigned long long seed, unsigned long long idx) { return qrd_hash_uniform_host(seed, idx); }

int qr_diffnorm_dev(qr_plan* p, const double* dX, int ldx, const double* dY, int ldy, long long rows, int cols,
                    long long row_off, long long total_rows, unsigned long long seed, int mode, double* sums)
{
    if (!p || !dX || !sums) return QR_E_ARG;
    return qrd_diff_norm(p->stream, dX, ldx, dY, ldy, rows, cols, row_off, total_rows, seed, mode == 1, sums);
}

int qr_device_malloc(void** dptr, size_t bytes) { CHECK(ensure_device()); return dptr ? qrd_malloc(dptr, bytes) : QR_E_ARG; }
int qr_device_free(void* dptr) { return qrd_free(dptr); }
int qr_copy_to_device(void* dst, const void* src, size_t bytes)
{
    CHECK(ensure_device());
    CHECK(qrd_h2d(NULL, dst, src, bytes));
    return qrd_stream_sync(NULL);
}
int qr_copy_to_host(void* dst, const void* src, size_t bytes)
{
    CHECK(ensure_device());
    CHECK(qrd_d2h(NULL, dst, src, bytes));
    return qrd_stream_sync(NULL);
}

int qr_device_info(char* arch, int arch_len, int* cus, int* clock_khz, size_t* hbm)
{
    CHECK(ensure_device());
    return qrd_device_info(arch, arch_len, cus, clock_khz, hbm);
}
int qr_probe_mfma_f64_tflops(double* t3) { CHECK(ensure_device()); return qrd_probe_mfma_f64(t3); }
int qr_probe_copy_gbps(double* g) { CHECK(ensure_device()); return qrd_probe_copy(g); }

/* ---------------------------------------------------------------------------------------------- *
 * Host-pointer entry points (drop-in for the reference's qr.c)
 * ---------------------------------------------------------------------------------------------- */
void getPanelDims(int m, int n, int* rowPanels, int* colPanels)
{
    int nb = 128;
    default_blocks(m, n, &nb, NULL);          /* the block size mmqr will really use for this shape */
    if (colPanels) *colPanels = n / nb + (n % nb != 0);
    if (rowPanels) *rowPanels = 1;
}

/* ---- cached plans for the host-pointer entry points -------------------------------------------------------------
 * The reference's callers invoke mmqr / explicitQR back to back on same-sized matrices (qr.cu:776-789 times three calls in
 * a row).  Creating a plan is ~25 hipMallocs, 2-6 streams and a dozen events -- about 10 ms, i.e. 60x the factorisation of
 * a 256 x 64 matrix -- so the drop-in entry points keep their last few plans (and the device copies of A / tau / Q / R)
 * keyed by (device, m, n, nb).  A slot is used by one call at a time; a concurrent call of the same shape from another
 * thread simply builds a private plan.  Shapes above QR_CACHE_MAX_ELEMS are never cached (their set-up cost is noise next
 * to the work, and the cache would pin gigabytes).  MI355XQR_PLAN_CACHE=0 turns it off; qr_release_cached_plans() empties it. */
#define QR_CACHE_SLOTS 4
#define QR_CACHE_MAX_ELEMS ((size_t) 1 << 26)
typedef struct host_slot {
    int used, busy, cached, dev, m, n, nb;
    unsigned long long stamp;
    qr_plan* p;
    double *dA, *dtau, *dQ, *dR;
    size_t q_cap, r_cap;            /* doubles */
} host_slot;
static host_slot g_slots[QR_CACHE_SLOTS];
static unsigned long long g_stamp = 0;

/* frees what the slot holds; a reserved cache slot keeps its busy flag (it is handed back through g_lock only) */
static void slot_free_contents(host_slot* sl, int keep_reserved)
{
    qrd_free(sl->dA); qrd_free(sl->dtau); qrd_free(sl->dQ); qrd_free(sl->dR);
    qr_plan_destroy(sl->p);
    sl->p = NULL; sl->dA = sl->dtau = sl->dQ = sl->dR = NULL;
    sl->q_cap = sl->r_cap = 0;
    sl->m = sl->n = sl->nb = 0;
    if (keep_reserved) return;
    pthread_mutex_lock(&g_lock);
    sl->used = 0; sl->cached = 0; sl->busy = 0;
    pthread_mutex_unlock(&g_lock);
}

static int slot_fill(host_slot* sl, int dev, int m, int n, int nb)
{
    int rc = qr_plan_create(&sl->p, m, n, nb, 0);
    if (!rc) rc = qrd_malloc((void**) &sl->dA, sizeof(double) * (size_t) m * n);
    if (!rc) rc = qrd_malloc((void**) &sl->dtau, sizeof(double) * (size_t) n);
    sl->dev = dev; sl->m = m; sl->n = n; sl->nb = nb;
    return rc;
}

/* returns a slot whose plan fits (m, n) exactly: a cached one, or `priv` filled as a private one-shot slot */
static int slot_acquire(int m, int n, host_slot* priv, host_slot** out)
{
    CHECK(ensure_device());
    int dev = 0, nb = 128;
    CHECK(qrd_get_device(&dev));
    default_blocks(m, n, &nb, NULL);
    const int cacheable = knobs()->plan_cache && (size_t) m * n <= QR_CACHE_MAX_ELEMS;
    host_slot* sl = NULL;
    int reuse = 0;
    if (cacheable) {
        pthread_mutex_lock(&g_lock);
        for (int i = 0; i < QR_CACHE_SLOTS && !sl; ++i)
            if (g_slots[i].used && !g_slots[i].busy && g_slots[i].dev == dev && g_slots[i].m == m && g_slots[i].n == n && g_slots[i].nb == nb) {
                sl = &g_slots[i];
                reuse = 1;
            }
        if (!sl) {                       /* a free slot, else the least recently used idle one */
            for (int i = 0; i < QR_CACHE_SLOTS; ++i) {
                if (g_slots[i].busy) continue;
                if (!g_slots[i].used) { sl = &g_slots[i]; break; }
                if (!sl || g_slots[i].stamp < sl->stamp) sl = &g_slots[i];
            }
        }
        if (sl) sl->busy = 1;            /* reserved: nobody else touches it until slot_release */
        pthread_mutex_unlock(&g_lock);
    }
    if (sl && reuse) { *out = sl; return 0; }
    if (sl) {                            /* rebuild the reserved slot outside the lock (busy stays 1 throughout) */
        if (sl->used) slot_free_contents(sl, 1);
        sl->cached = 1;
        const int rc = slot_fill(sl, dev, m, n, nb);
        if (rc) {
            slot_free_contents(sl, 0);   /* the slot is free again */
            return rc;
        }
        pthread_mutex_lock(&g_lock);
        sl->used = 1;
        pthread_mutex_unlock(&g_lock);
        *out = sl;
        return 0;
    }
    memset(priv, 0, sizeof(*priv));
    const int rc = slot_fill(priv, dev, m, n, nb);
    if (rc) { slot_free_contents(priv, 1); return rc; }
    *out = priv;
    return 0;
}

static void slot_release(host_slot* sl)
{
    if (!sl) return;
    if (!sl->cached) { slot_free_contents(sl, 1); return; }
    pthread_mutex_lock(&g_lock);
    sl->stamp = ++g_stamp;
    sl->busy = 0;
    pthread_mutex_unlock(&g_lock);
}

static int slot_need(double** buf, size_t* cap, size_t elems)
{
    if (*cap >= elems) return 0;
    qrd_free(*buf);
    *buf = NULL; *cap = 0;
    CHECK(qrd_malloc((void**) buf, sizeof(double) * elems));
    *cap = elems;
    return 0;
}

int qr_release_cached_plans(void)
{
    for (int i = 0; i < QR_CACHE_SLOTS; ++i) {
        pthread_mutex_lock(&g_lock);
        const int take = g_slots[i].used && !g_slots[i].busy;
        if (take) g_slots[i].busy = 1;
        pthread_mutex_unlock(&g_lock);
        if (take) slot_free_contents(&g_slots[i], 0);
    }
    return 0;
}

int mmqr_status(double* mat, double** tau, int m, int n)
{
    if (!mat || !tau || n < 1 || m < n) return QR_E_ARG;
    /* Heights that are not multiples of 16 are factored with zero rows appended, in a device buffer of its own leading dimension: the
     * reflectors get zeros there and R, tau and the first m rows of V are those of the unpadded matrix (in exact arithmetic: the same
     * sums with zeros added), while an odd height or leading dimension keeps every kernel off its aligned path -- the one-launch panel
     * wants mk % 4 == 0, vector loads an even leading dimension: 5001^2 34.0 ms against 10.9 for 5000^2, 8191^2 60.9 against 22.9
     * (profiles/r06_odd_sizes.txt).  Callers of the device API own their buffers and pad (or not) themselves. */
    const int mp = (m >= 512 && m % 16 != 0) ? (int) (((long long) m + 15) & ~15LL) : m;
    host_slot priv, *sl = NULL;
    CHECK(slot_acquire(mp, n, &priv, &sl));
    qr_plan* p = sl->p;
    int nb_u = p->nb;
    if (mp != m) default_blocks(m, n, &nb_u, NULL);                           /* tau is sized by the block size the CALLER's shape gets (getPanelDims) */
    const size_t ntau = (size_t) ((n + nb_u - 1) / nb_u) * nb_u;              /* rowPanels * colPanels * nb, qr.c:61 sizing rule */
    double* htau = (double*) calloc(ntau, sizeof(double));                    /* zero-filled like qr.c:61-62 */
    if (!htau) { slot_release(sl); return QR_E_ALLOC; }
    const size_t bytes = sizeof(double) * (size_t) m * n;
    int rc = 0;
    /* host-pointer entry points block anyway: their plans always POLL the guard's verdict (a refused tall panel goes to the leaf chain
     * at once), whatever MI355XQR_GUARD says -- the latch mode is for callers of the device API that must not block */
    const int latch0 = p->guard_latch;
    p->guard_latch = 0;
    for (int attempt = 0; attempt < 2; ++attempt) {
        if (mp == m) rc = qrd_h2d(p->stream, sl->dA, mat, bytes);
        else {
            rc = qrd_memset(p->stream, sl->dA, 0, sizeof(double) * (size_t) mp * n);
            if (!rc) rc = qrd_h2d_2d(p->stream, sl->dA, sizeof(double) * mp, mat, sizeof(double) * m, sizeof(double) * m, n);
        }
        if (!rc) rc = qr_geqrf_dev(p, sl->dA, mp, n, mp, sl->dtau);
        /* the status words of the panel kernels, before the result replaces the caller's matrix: after a stalled one-launch panel the
         * matrix is factored again from the host copy with that route off */
        const int rs = qr_plan_sync(p);
        if (!rc) rc = rs;
        if (rc == QR_E_STALL && !p->fused_off) { p->fused_off = 1; continue; }
        break;
    }
    p->guard_latch = latch0;
    if (!rc) rc = mp == m ? qrd_d2h(p->stream, mat, sl->dA, bytes)
                          : qrd_d2h_2d(p->stream, mat, sizeof(double) * m, sl->dA, sizeof(double) * mp, sizeof(double) * m, n);
    if (!rc) rc = qrd_d2h(p->stream, htau, sl->dtau, sizeof(double) * n);
    if (!rc) rc = qrd_stream_sync(p->stream);
    else qr_plan_sync(p);
    slot_release(sl);
    if (rc) { free(htau); return rc; }
    *tau = htau;
    return 0;
}

int explicitQR_status(double* A, double* tau, double* Q, double* R, int m, int n)
{
    if (!A || !tau || !Q || !R || n < 1 || m < n) return QR_E_ARG;
    host_slot priv, *sl = NULL;
    CHECK(slot_acquire(m, n, &priv, &sl));
    qr_plan* p = sl->p;
    const size_t abytes = sizeof(double) * (size_t) m * n, qbytes = sizeof(double) * (size_t) m * m;
    int rc = slot_need(&sl->dQ, &sl->q_cap, (size_t) m * m);
    if (!rc) rc = slot_need(&sl->dR, &sl->r_cap, (size_t) m * n);
    if (!rc) rc = qrd_h2d(p->stream, sl->dA, A, abytes);
    if (!rc) rc = qrd_h2d(p->stream, sl->dtau, tau, sizeof(double) * n);
    if (!rc) rc = qr_extract_r_dev(p, sl->dA, m, n, m, sl->dR, m, m);
    if (!rc) rc = qr_applyq_dev(p, sl->dA, m, n, m, sl->dtau, sl->dQ, m, m, 1);
    if (!rc) rc = qrd_d2h(p->stream, R, sl->dR, abytes);
    if (!rc) rc = qrd_d2h(p->stream, Q, sl->dQ, qbytes);
    if (!rc) rc = qrd_stream_sync(p->stream);
    else qr_plan_sync(p);
    slot_release(sl);
    return rc;
}

int dgemm_status(double* A, double* B, double* C, int k, int m, int n)
{
    if (!A || !B || !C || k < 1 || m < 1 || n < 1) return QR_E_ARG;
    CHECK(ensure_device());
    void* s = NULL;
    double *dA = NULL, *dB = NULL, *dC = NULL;
    int rc = qrd_stream_create(&s, 0);
    if (!rc) rc = qrd_malloc((void**) &dA, sizeof(double) * (size_t) k * m);
    if (!rc) rc = qrd_malloc((void**) &dB, sizeof(double) * (size_t) m * n);
    if (!rc) rc = qrd_malloc((void**) &dC, sizeof(double) * (size_t) k * n);
    if (!rc) rc = qrd_h2d(s, dA, A, sizeof(double) * (size_t) k * m);
    if (!rc) rc = qrd_h2d(s, dB, B, sizeof(double) * (size_t) m * n);
    if (!rc) rc = qrd_gemm_nn(s, k, n, m, 1.0, dA, k, dB, m, 0.0, dC, k);
    if (!rc) rc = qrd_d2h(s, C, dC, sizeof(double) * (size_t) k * n);
    if (!rc) rc = qrd_stream_sync(s);
    qrd_free(dA); qrd_free(dB); qrd_free(dC);
    if (s) qrd_stream_destroy(s);
    return rc;
}

/* ---- float instantiation of the drop-in symbols (the reference as committed has Scalar = float, qr.c:11) ----
 * Same semantics on float arrays.  The arithmetic is the fp64 path: inputs are widened, results rounded once on the way out, so
 * a float caller gets results at least as accurate as a float build of the reference (whose 6 x 4 self-test prints a residual
 * of 3.8e-07, qr.c:505-515).  fp32 MFMA kernels would double the rate; not built (no consumer). */
static double* widen(const float* x, size_t n)
{
    double* d = (double*) malloc(sizeof(double) * (n ? n : 1));
    if (d) for (size_t i = 0; i < n; ++i) d[i] = (double) x[i];
    return d;
}

int mmqr_f32_status(float* mat, float** tau, int m, int n)
{
    if (!mat || !tau || n < 1 || m < n) return QR_E_ARG;
    const size_t cnt = (size_t) m * n;
    double* d = widen(mat, cnt);
    if (!d) return QR_E_ALLOC;
    double* dt = NULL;
    int rc = mmqr_status(d, &dt, m, n);
    if (!rc) {
        int rp, cp, nb = 128;
        getPanelDims(m, n, &rp, &cp);
        default_blocks(m, n, &nb, NULL);
        const size_t nt = (size_t) rp * cp * nb;
        float* ft = (float*) calloc(nt, sizeof(float));
        if (!ft) rc = QR_E_ALLOC;
        else {
            for (size_t i = 0; i < nt; ++i) ft[i] = (float) dt[i];
            for (size_t i = 0; i < cnt; ++i) mat[i] = (float) d[i];
            *tau = ft;
        }
    }
    free(dt); free(d);
    return rc;
}

int explicitQR_f32_status(float* A, float* tau, float* Q, float* R, int m, int n)
{
    if (!A || !tau || !Q || !R || n < 1 || m < n) return QR_E_ARG;
    const size_t an = (size_t) m * n, qn = (size_t) m * m;
    double *dA = widen(A, an), *dt = widen(tau, (size_t) n);
    double *dQ = (double*) malloc(sizeof(double) * qn), *dR = (double*) malloc(sizeof(double) * an);
    int rc = (dA && dt && dQ && dR) ? explicitQR_status(dA, dt, dQ, dR, m, n) : QR_E_ALLOC;
    if (!rc) {
        for (size_t i = 0; i < qn; ++i) Q[i] = (float) dQ[i];
        for (size_t i = 0; i < an; ++i) R[i] = (float) dR[i];
    }
    free(dA); free(dt); free(dQ); free(dR);
    return rc;
}

/* ---- legacy-layout shim: the reference's sliding-window MMQR itself, window shape as arguments ---------------
 * For callers that read the RAW factored form: reflector tails where the reference leaves them and the window-indexed tau array
 * tau[(rowPanels * pcCount + prCount) * PC + i] (qr.c:300-304; main's debug print reads it, qr.c:483-490).  That form describes the
 * reference's own reflector set for a given PR x PC (compile-time macros there, qr.c:12-13) and cannot be derived from the blocked
 * factorisation, so the device runs the reference's schedule (qr_legacy.hip): one launch pair per column panel.  The same shape
 * constraints as the reference's loops silently assume (n % PC == 0, (m - PR) % (PR - PC) == 0), checked here. */
void getPanelDims_legacy(int m, int n, int PR, int PC, int* rowPanels, int* colPanels)
{
    if (PC < 1 || PR <= PC) { if (rowPanels) *rowPanels = 0; if (colPanels) *colPanels = 0; return; }
    if (colPanels) *colPanels = n / PC + (n % PC != 0);                       /* qr.c:47-53 */
    if (rowPanels) {
        *rowPanels = 1;
        if (m > PR) { const int span = m - PR, step = PR - PC; *rowPanels += span / step + (span % step != 0); }
    }
}

int mmqr_legacy_status(double* mat, double** tau, int m, int n, int PR, int PC)
{
    if (!mat || !tau || n < 1 || m < n || !qrd_legacy_shape_ok(m, n, PR, PC)) return QR_E_ARG;
    CHECK(ensure_device());
    int rp = 0, cp = 0;
    getPanelDims_legacy(m, n, PR, PC, &rp, &cp);
    const size_t ntau = (size_t) rp * cp * PC, bytes = sizeof(double) * (size_t) m * n;
    double* htau = (double*) calloc(ntau, sizeof(double));                     /* zero-filled like qr.c:61-62 */
    if (!htau) return QR_E_ALLOC;
    void* s = NULL;
    double *dA = NULL, *dtau = NULL, *dwy = NULL;
    int rc = qrd_stream_create(&s, 0);
    if (!rc) rc = qrd_malloc((void**) &dA, bytes);
    if (!rc) rc = qrd_malloc((void**) &dtau, sizeof(double) * ntau);
    if (!rc) rc = qrd_malloc((void**) &dwy, sizeof(double) * qrd_legacy_ws_size(m, PR, PC));
    if (!rc) rc = qrd_memset(s, dtau, 0, sizeof(double) * ntau);
    if (!rc) rc = qrd_h2d(s, dA, mat, bytes);
    int pcCount = 0;
    for (int pc = 0; pc < n && !rc; pc += PC, ++pcCount)                       /* qr.c:68: column panels left -> right */
        rc = qrd_legacy_panel(s, dA, m, n, PR, PC, rp, pc, pcCount, dtau, dwy);
    if (!rc) rc = qrd_d2h(s, mat, dA, bytes);
    if (!rc) rc = qrd_d2h(s, htau, dtau, sizeof(double) * ntau);
    if (s) { const int rs = qrd_stream_sync(s); if (!rc) rc = rs; }
    qrd_free(dA); qrd_free(dtau); qrd_free(dwy);
    if (s) qrd_stream_destroy(s);
    if (rc) { free(htau); return rc; }
    *tau = htau;
    return 0;
}

/* R (m x n: upper triangle of A, qr.c:334-343) and the dense m x m Q = product of the reflectors in the reference's order
 * (qr.c:353-438) from mmqr_legacy_status's output */
int explicitQR_legacy_status(double* A, double* tau, double* Q, double* R, int m, int n, int PR, int PC)
{
    if (!A || !tau || !Q || !R || n < 1 || m < n || !qrd_legacy_shape_ok(m, n, PR, PC)) return QR_E_ARG;
    CHECK(ensure_device());
    int rp = 0, cp = 0;
    getPanelDims_legacy(m, n, PR, PC, &rp, &cp);
    const size_t ntau = (size_t) rp * cp * PC, abytes = sizeof(double) * (size_t) m * n, qbytes = sizeof(double) * (size_t) m * m;
    void* s = NULL;
    double *dA = NULL, *dtau = NULL, *dQ = NULL, *dR = NULL;
    int rc = qrd_stream_create(&s, 0);
    if (!rc) rc = qrd_malloc((void**) &dA, abytes);
    if (!rc) rc = qrd_malloc((void**) &dtau, sizeof(double) * ntau);
    if (!rc) rc = qrd_malloc((void**) &dQ, qbytes);
    if (!rc) rc = qrd_malloc((void**) &dR, abytes);
    if (!rc) rc = qrd_h2d(s, dA, A, abytes);
    if (!rc) rc = qrd_h2d(s, dtau, tau, sizeof(double) * ntau);
    if (!rc) rc = qrd_extract_r(s, dA, m, m, n, dR, m, m);
    if (!rc) rc = qrd_legacy_formq(s, dA, dtau, m, n, PR, PC, rp, dQ);
    if (!rc) rc = qrd_d2h(s, R, dR, abytes);
    if (!rc) rc = qrd_d2h(s, Q, dQ, qbytes);
    if (s) { const int rs = qrd_stream_sync(s); if (!rc) rc = rs; }
    qrd_free(dA); qrd_free(dtau); qrd_free(dQ); qrd_free(dR);
    if (s) qrd_stream_destroy(s);
    return rc;
}

static void complain(const char* fn, int rc)
{
    if (rc) fprintf(stderr, "mi355xqr: %s failed: %s (%d)\n", fn, qr_strerror(rc), rc);
}

void mmqr(double* mat, double** tau, int m, int n) { complain("mmqr", mmqr_status(mat, tau, m, n)); }
void explicitQR(double* A, double* tau, double* Q, double* R, int m, int n)
{ complain("explicitQR", explicitQR_status(A, tau, Q, R, m, n)); }
void dgemm(double* A, double* B, double* C, int k, int m, int n) { complain("dgemm", dgemm_status(A, B, C, k, m, n)); }
void mmqr_f32(float* mat, float** tau, int m, int n) { complain("mmqr_f32", mmqr_f32_status(mat, tau, m, n)); }
void explicitQR_f32(float* A, float* tau, float* Q, float* R, int m, int n)
{ complain("explicitQR_f32", explicitQR_f32_status(A, tau, Q, R, m, n)); }

void identity(double* A, int m)
{
    if (!A || m < 1) return;
    memset(A, 0, sizeof(double) * (size_t) m * m);
    for (int i = 0; i < m; ++i) A[(size_t) i * m + i] = 1.0;
}

void printMat(double* mat, int m, int n)
{
    if (!mat) return;
    printf("Matrix %d x %d, row by row:\n", m, n);
    for (int r = 0; r < m; ++r) {
        for (int c = 0; c < n; ++c) printf("%9f ", mat[(size_t) c * m + r]);
        putchar('\n');
    }
    putchar('\n');
}

/* ---- thin QR / single-device TSQR over row shards -------------------------------------------- */
int qr_thin(const double* A, int m, int n, double* Q, double* R, int nb, int nshards)
{
    if (!A || !Q || !R || n < 1 || m < n || nshards < 1) return QR_E_ARG;
    const int ms = (m + nshards - 1) / nshards;
    if (nshards > 1 && (m - (nshards - 1) * ms) < n) return QR_E_ARG;   /* every shard needs >= n rows */
    qr_plan *p = NULL, *p2 = NULL;
    double *dA = NULL, *dQ = NULL, *dtau = NULL, *dR = NULL, *dS = NULL, *dQt = NULL, *dtau2 = NULL;
    /* one shard: heights that are not multiples of 16 get zero rows appended on the device (see mmqr_status); md = the device-side height */
    const int md = (nshards == 1 && m >= 512 && m % 16 != 0) ? (int) (((long long) m + 15) & ~15LL) : m;
    const size_t abytes = sizeof(double) * (size_t) md * n;
    const int sm = nshards * n;
    int rc = qr_plan_create(&p, nshards > 1 ? ms : md, n, nb, 0);
    if (!rc) rc = qrd_malloc((void**) &dA, abytes);
    if (!rc) rc = qrd_malloc((void**) &dQ, abytes);
    if (!rc) rc = qrd_malloc((void**) &dtau, sizeof(double) * (size_t) n * nshards);
    if (!rc) rc = qrd_malloc((void**) &dR, sizeof(double) * (size_t) n * n);
    if (!rc && nshards > 1) {
        rc = qr_plan_create(&p2, sm, n, nb, 0);
        if (!rc) rc = qrd_malloc((void**) &dS, sizeof(double) * (size_t) sm * n);
        if (!rc) rc = qrd_malloc((void**) &dQt, sizeof(double) * (size_t) sm * n);
        if (!rc) rc = qrd_malloc((void**) &dtau2, sizeof(double) * n);
    }
    /* a host-pointer entry point: its plans poll the guard's verdict (see mmqr_status), and the status words of the panel kernels are
     * read (qr_plan_sync, both plans) BEFORE anything is copied back -- after a stalled one-launch panel everything is redone from the
     * caller's matrix with that route off; any other status is the caller's answer, never a silent rc = 0 over garbage */
    if (p) p->guard_latch = 0;
    if (p2) p2->guard_latch = 0;
    for (int attempt = 0; attempt < 2 && !rc; ++attempt) {
        if (md == m) rc = qrd_h2d(p->stream, dA, A, abytes);
        else {
            rc = qrd_memset(p->stream, dA, 0, abytes);
            if (!rc) rc = qrd_h2d_2d(p->stream, dA, sizeof(double) * md, A, sizeof(double) * m, sizeof(double) * m, n);
        }
        if (!rc && nshards == 1) {
            rc = qr_geqrf_dev(p, dA, md, n, md, dtau);
            if (!rc) rc = qr_extract_r_dev(p, dA, md, n, md, dR, n, n);
            if (!rc) rc = qr_applyq_dev(p, dA, md, n, md, dtau, dQ, n, md, 1);      /* the appended rows of the thin Q are zero */
        } else if (!rc) {
            for (int s = 0; s < nshards && !rc; ++s) {            /* step 1: independent local QRs */
                const int r0 = s * ms, rows = imin(ms, m - r0);
                rc = qr_geqrf_dev(p, dA + r0, rows, n, m, dtau + (size_t) s * n);
                if (!rc) rc = qr_extract_r_dev(p, dA + r0, rows, n, m, dS + (size_t) s * n, n, sm);
            }
            if (!rc) rc = qr_plan_sync(p);
            /* step 2: QR of the stacked R factors (what every rank does after the all-gather) */
            if (!rc) rc = qr_geqrf_dev(p2, dS, sm, n, sm, dtau2);
            if (!rc) rc = qr_extract_r_dev(p2, dS, sm, n, sm, dR, n, n);
            if (!rc) rc = qr_applyq_dev(p2, dS, sm, n, sm, dtau2, dQt, n, sm, 1);
            if (!rc) rc = qr_plan_sync(p2);
            /* step 3: Q_s = Q_local_s * [Qtree_s ; 0] */
            if (!rc) rc = qrd_memset(p->stream, dQ, 0, abytes);
            for (int s = 0; s < nshards && !rc; ++s) {
                const int r0 = s * ms, rows = imin(ms, m - r0);
                rc = qrd_copy_block(p->stream, dQt + (size_t) s * n, sm, dQ + r0, m, n, n);
                if (!rc) rc = qr_applyq_dev(p, dA + r0, rows, n, m, dtau + (size_t) s * n, dQ + r0, n, m, 0);
            }
        }
        {
            const int rs = qr_plan_sync(p), rs2 = p2 ? qr_plan_sync(p2) : 0;
            if (!rc) rc = rs ? rs : rs2;
        }
        if (rc == QR_E_STALL && !p->fused_off) {
            p->fused_off = 1;
            if (p2) p2->fused_off = 1;
            rc = 0;
            continue;
        }
        break;
    }
    if (!rc) rc = md == m ? qrd_d2h(p->stream, Q, dQ, abytes)
                          : qrd_d2h_2d(p->stream, Q, sizeof(double) * m, dQ, sizeof(double) * md, sizeof(double) * m, n);
    if (!rc) rc = qrd_d2h(p->stream, R, dR, sizeof(double) * (size_t) n * n);
    if (!rc) rc = qrd_stream_sync(p->stream);
    qrd_free(dA); qrd_free(dQ); qrd_free(dtau); qrd_free(dR); qrd_free(dS); qrd_free(dQt); qrd_free(dtau2);
    qr_plan_destroy(p); qr_plan_destroy(p2);
    return rc;
}

/* ---- device-resident TSQR step over one communicator (north_star: "host code stays in C") -------------------
 * One qr_tsqr_plan per rank (= per GPU: one process or one host thread each).  A factorisation of the rank's row shard:
 *   1. local qr_geqrf_dev of the shard -> R_p (n x n) packed into the plan's send buffer
 *   2. ONE ncclAllGather of the R factors on the plan's stream (RCCL over xGMI; latency-bound: n = 512 is 2 MiB per rank)
 *   3. redundant QR of the stacked (P n) x n matrix on every rank -> the final R (identical bits everywhere)
 *   4. (qr_tsqr_formq_dev) Q_p = Q_local_p [Qtree_p; 0]
 * Stream-ordered, with ONE exception: in the default guard mode the host thread waits for the verdict word of each full-width panel of
 * the local factorisation (shards above 8192 rows: once per 128 columns, while that panel's last pass still runs -- the GPU does not
 * idle, see panel_cqr_half); qr_plan_set_guard_mode(qr_tsqr_local_plan(tp), 1) removes that wait at the price of QR_E_REFUSED on
 * ill-conditioned input.  No stream is ever drained inside a step.  The stacked factorisation runs on its own plan's streams
 * behind an event, and the next step's exchange waits for it through another, so in a sequence of independent factorisations the
 * stacked QR of step i runs under the local QR of step i+1 without the caller doing anything; qr_tsqr_sync() drains both.
 * No reference counterpart: the reference is single-device (qr.cu:711,737). */
#define QR_TSQR_MAXPAN 16
struct qr_tsqr_plan {
    int nranks, rank, m_local, n, nb, sm;
    qr_plan *p, *p2;                /* local shard, stacked R factors (NULL when nranks == 1) */
    void* comm; int own_comm;       /* ncclComm_t as void*; own_comm: created here from a unique id, destroyed with the plan */
    double *dtau, *dtau2, *dRp, *dRall, *dS, *dQt;
    void *ev_gathered, *ev_stacked; /* stack matrix filled (local stream) / stacked factorisation has consumed it (stack stream) */
    int stacked_pending;            /* ev_stacked has been recorded and not yet waited for by the local stream */
    void* ev_q; int q_pending;      /* the local stream has copied this rank's block out of dQt (qr_tsqr_formq_dev): the next call's tree Q waits */
    int local_done;                 /* qr_tsqr_local_dev ran and the stacked step has not yet consumed its R factor */
    /* panel-pipelined form (see qr_tsqr_factor_dev): the exchange and the stacked QR go block column by block column */
    int pipe_ok, npan;              /* usable: single-stream local plan, nb | n, same nb in both plans */
    int pan_k[QR_TSQR_MAXPAN + 1];  /* block column boundaries: full blocks of nb, the LAST block in two halves (what is left exposed after
                                     * the local QR has ended is the stacked factorisation of the last block column: make it short) */
    double *dsend;                  /* npan blocks of n x nb: block column k of this rank's R, zero below its trapezoid */
    double *drecv;                  /* nranks blocks of n x nb: one gathered block column (reused panel after panel, stream order) */
    double *Vst, *Tst;              /* explicit V (ldv2 x nb) and T (ldt x nb) of every stacked panel: later panels apply them */
    void *ev_pan[QR_TSQR_MAXPAN];   /* local panel k factored, its block column of R packed (local stream) */
    void *ev_sent[QR_TSQR_MAXPAN];  /* the exchange has consumed send block k (stacked stream; with timing: end of gather k) */
    void *ev_g0[QR_TSQR_MAXPAN];    /* the stacked stream is past its wait for local panel k: gather k starts (timing) */
    void *ev_t0, *ev_t1;            /* first launch of a pipelined call (local stream) / its last (stacked stream) */
    int sent_pending[QR_TSQR_MAXPAN];
    /* what the exchange costs on THIS node is not known before the first multi-rank run: the gathers are timed (qr_tsqr_gather_stats),
     * and with MI355XQR_TSQR_PIPE unset the ranks decide TOGETHER, once, whether to keep the pipelined form (tsqr_decide) */
    int pipe_auto, pipe_calls, pipe_fell_back, stats_valid;
    int pipe_able, pipe_env_on, pipe_env_auto;   /* the plan can run the pipelined form at all / what the environment asked for (qr_tsqr_set_schedule) */
    double *dstat;                  /* 2 + 2 * nranks doubles: this rank's {gather sum, step} ms, then every rank's */
};
#define QR_TSQR_DECIDE_CALL 2       /* the decision is taken before this pipelined call (0-based): call 0 pays RCCL's lazy set-up, call 1 is warm */

int qr_tsqr_unique_id(void* id128)
{
    if (!id128) return QR_E_ARG;
    return qrd_comm_unique_id(id128);
}

static int tsqr_plan_build(qr_tsqr_plan** out, void* comm, int own_comm, int nranks, int rank, int m_local, int n, int nb)
{
    qr_tsqr_plan* t = (qr_tsqr_plan*) calloc(1, sizeof *t);
    if (!t) { if (own_comm) qrd_comm_destroy(comm); return QR_E_ALLOC; }
    t->nranks = nranks; t->rank = rank; t->m_local = m_local; t->n = n; t->nb = nb; t->sm = nranks * n;
    t->comm = comm; t->own_comm = own_comm;
    const size_t nn = (size_t) n * n;
    int rc = plan_create_impl(&t->p, m_local, n, nb, 0, nranks > 1);
    if (!rc) rc = qrd_malloc((void**) &t->dtau, sizeof(double) * n);
    if (!rc) rc = qrd_malloc((void**) &t->dRp, sizeof(double) * nn);
    if (!rc && nranks > 1) {
        rc = qr_plan_create(&t->p2, t->sm, n, t->p->nb, t->p->ib);     /* the local plan's blocks: the pipelined exchange walks both panel by panel */
        if (!rc) rc = qrd_malloc((void**) &t->dtau2, sizeof(double) * n);
        if (!rc) rc = qrd_malloc((void**) &t->dRall, sizeof(double) * nn * nranks);
        if (!rc) rc = qrd_malloc((void**) &t->dS, sizeof(double) * (size_t) t->sm * n);
        if (!rc) rc = qrd_event_create_notiming(&t->ev_gathered);
        if (!rc) rc = qrd_event_create_notiming(&t->ev_stacked);
        if (!rc) rc = qrd_event_create_notiming(&t->ev_q);
        if (!rc) rc = qrd_malloc((void**) &t->dQt, sizeof(double) * (size_t) t->sm * n);      /* not on first use: no hipMalloc beside running collectives */
        /* panel-pipelined exchange: MI355XQR_TSQR_PIPE=0 keeps the one-collective form */
        const int pnb = rc ? 0 : t->p->nb;
        t->pipe_auto = getenv("MI355XQR_TSQR_PIPE") == NULL || strcmp(getenv("MI355XQR_TSQR_PIPE"), "auto") == 0;
        t->pipe_env_auto = t->pipe_auto;
        t->pipe_env_on = t->pipe_auto || env_int("MI355XQR_TSQR_PIPE", 1) != 0;
        /* (the pipelined form's buffers and events exist whenever the shape allows it, whatever the environment says: qr_tsqr_set_schedule can
         * switch between the two forms) */
        if (!rc && !t->p->lookahead && !t->p2->lookahead && pnb == t->p2->nb && n % pnb == 0 &&
            n / pnb >= 2 && n / pnb + 1 <= QR_TSQR_MAXPAN) {
            t->npan = n / pnb;
            for (int k = 0; k <= t->npan; ++k) t->pan_k[k] = k * pnb;
            /* the last block column in two halves -- what stays exposed after the local QR has ended is the stacked factorisation of the
             * last block, and the first half's can run under the local second half -- while the stacked matrix is short (its panels go
             * leaf by leaf); from 3072 stacked rows a stacked panel is ONE launch (qr_panel_fused.hip) and two of them with an update in
             * between cost more than they hide (C5 rank, 4096 stacked rows: 0.83 -> 0.65 ms exposed without the split; C4 ranks, 1024 /
             * 512 rows: 0.29 / 0.40 with it against 0.32 / 0.44).  Halves of 32 columns lose (65536 x 256, nb 64: 0.39 against 0.32) */
            if (pnb >= 128 && (pnb / 2) % t->p->ib == 0 && t->sm < knobs()->tsqr_halves_rows) {
                t->pan_k[t->npan] = n - pnb / 2;
                t->pan_k[++t->npan] = n;
            }
            rc = qrd_malloc((void**) &t->dsend, sizeof(double) * (size_t) n * n);
            if (!rc) rc = qrd_memset(t->p2->s_main, t->dsend, 0, sizeof(double) * (size_t) n * n);
            if (!rc) rc = qrd_malloc((void**) &t->drecv, sizeof(double) * (size_t) nranks * n * pnb);
            if (!rc) rc = qrd_malloc((void**) &t->Vst, sizeof(double) * (size_t) t->npan * t->p2->ldv * pnb);
            if (!rc) rc = qrd_malloc((void**) &t->Tst, sizeof(double) * (size_t) t->npan * t->p2->ldt * pnb);
            if (!rc) rc = qrd_memset(t->p2->s_main, t->Vst, 0, sizeof(double) * (size_t) t->npan * t->p2->ldv * pnb);
            if (!rc) rc = qrd_memset(t->p2->s_main, t->Tst, 0, sizeof(double) * (size_t) t->npan * t->p2->ldt * pnb);
            for (int k = 0; k < t->npan && !rc; ++k) {
                rc = qrd_event_create_notiming(&t->ev_pan[k]);
                if (!rc) rc = qrd_event_create(&t->ev_sent[k]);
                if (!rc) rc = qrd_event_create(&t->ev_g0[k]);
            }
            if (!rc) rc = qrd_event_create(&t->ev_t0);
            if (!rc) rc = qrd_event_create(&t->ev_t1);
            if (!rc) rc = qrd_malloc((void**) &t->dstat, sizeof(double) * (size_t) (2 + 2 * nranks));
            if (!rc) rc = qrd_stream_sync(t->p2->s_main);
            t->pipe_able = !rc;
            t->pipe_ok = t->pipe_able && t->pipe_env_on;
        }
    }
    if (rc) { qr_tsqr_plan_destroy(t); return rc; }
    *out = t;
    return 0;
}

int qr_tsqr_plan_create(qr_tsqr_plan** out, const void* id128, int nranks, int rank, int m_local, int n, int nb)
{
    if (!out || nranks < 1 || rank < 0 || rank >= nranks || n < 1 || m_local < n || (nranks > 1 && !id128)) return QR_E_ARG;
    if ((long long) nranks * n > 0x7fffffffLL / 2) return QR_E_ARG;
    CHECK(ensure_device());
    void* comm = NULL;
    if (nranks > 1) CHECK(qrd_comm_init_rank(&comm, nranks, id128, rank));
    return tsqr_plan_build(out, comm, nranks > 1, nranks, rank, m_local, n, nb);
}

/* the same over a communicator the caller already owns (an ncclComm_t, passed as void*): it is used, never destroyed */
int qr_tsqr_plan_create_comm(qr_tsqr_plan** out, void* nccl_comm, int nranks, int rank, int m_local, int n, int nb)
{
    if (!out || nranks < 1 || rank < 0 || rank >= nranks || n < 1 || m_local < n) return QR_E_ARG;
    if ((long long) nranks * n > 0x7fffffffLL / 2) return QR_E_ARG;
    CHECK(ensure_device());
    return tsqr_plan_build(out, nccl_comm, 0, nranks, rank, m_local, n, nb);
}

int qr_tsqr_plan_destroy(qr_tsqr_plan* t)
{
    if (!t) return 0;
    qr_tsqr_sync(t);
    if (t->ev_gathered) qrd_event_destroy(t->ev_gathered);
    if (t->ev_stacked) qrd_event_destroy(t->ev_stacked);
    if (t->ev_q) qrd_event_destroy(t->ev_q);
    qrd_free(t->dtau); qrd_free(t->dtau2); qrd_free(t->dRp); qrd_free(t->dRall); qrd_free(t->dS); qrd_free(t->dQt);
    qrd_free(t->dsend); qrd_free(t->drecv); qrd_free(t->Vst); qrd_free(t->Tst); qrd_free(t->dstat);
    for (int k = 0; k < QR_TSQR_MAXPAN; ++k) {
        if (t->ev_pan[k]) qrd_event_destroy(t->ev_pan[k]);
        if (t->ev_sent[k]) qrd_event_destroy(t->ev_sent[k]);
        if (t->ev_g0[k]) qrd_event_destroy(t->ev_g0[k]);
    }
    if (t->ev_t0) qrd_event_destroy(t->ev_t0);
    if (t->ev_t1) qrd_event_destroy(t->ev_t1);
    qr_plan_destroy(t->p); qr_plan_destroy(t->p2);
    if (t->own_comm && t->comm) qrd_comm_destroy(t->comm);
    free(t);
    return 0;
}

int qr_tsqr_sync(qr_tsqr_plan* t)
{
    if (!t) return QR_E_ARG;
    /* BOTH plans, always: a soft status of the local plan (QR_E_REFUSED / QR_E_STALL) must not leave the stacked plan's streams -- and
     * the collectives of the invalid step -- running, nor its status words unread, when the caller reacts by factoring again */
    const int rc = t->p ? qr_plan_sync(t->p) : 0;
    const int rc2 = t->p2 ? qr_plan_sync(t->p2) : 0;
    return rc ? rc : rc2;
}

void* qr_tsqr_stream(qr_tsqr_plan* t) { return t && t->p ? t->p->s_main : NULL; }
qr_plan* qr_tsqr_local_plan(qr_tsqr_plan* t) { return t ? t->p : NULL; }
qr_plan* qr_tsqr_stacked_plan(qr_tsqr_plan* t) { return t ? t->p2 : NULL; }

/* ranks of the communicator as RCCL itself counts them (1 without a communicator) */
int qr_tsqr_comm_ranks(qr_tsqr_plan* t, int* nranks)
{
    if (!t || !nranks) return QR_E_ARG;
    *nranks = 1;
    if (t->nranks == 1 || !t->comm) return 0;
    return qrd_comm_count(t->comm, nranks);
}

/* step 1: local factorisation of the shard, R_p packed into the send buffer */
int qr_tsqr_local_dev(qr_tsqr_plan* t, double* dA, int lda)
{
    if (!t || !dA || lda < t->m_local) return QR_E_ARG;
    qr_plan* p = t->p;
    CHECK(qr_geqrf_dev(p, dA, t->m_local, t->n, lda, t->dtau));
    CHECK(qr_extract_r_dev(p, dA, t->m_local, t->n, lda, t->dRp, t->n, t->n));
    t->local_done = 1;
    return 0;
}

/* for transports other than RCCL (bring-up with more ranks than GPUs): after qr_tsqr_local_dev + qr_tsqr_sync, `*send` holds
 * this rank's n x n R factor (column-major, ld n); the caller fills `*recv` with all ranks' factors in rank order
 * (nranks * n * n doubles) and calls qr_tsqr_stacked_dev. */
int qr_tsqr_exchange_buffers(qr_tsqr_plan* t, double** send, double** recv)
{
    if (!t) return QR_E_ARG;
    if (send) *send = t->dRp;
    if (recv) *recv = t->dRall;
    return 0;
}

/* steps 3: stack the gathered factors, factor, dR (n x n, ld n) = final R */
int qr_tsqr_stacked_dev(qr_tsqr_plan* t, double* dR)
{
    if (!t || !dR) return QR_E_ARG;
    const int n = t->n, P = t->nranks, sm = t->sm;
    const size_t nn = (size_t) n * n;
    qr_plan* p = t->p;
    if (P == 1) { t->local_done = 0; return qrd_d2d(p->s_main, dR, t->dRp, sizeof(double) * nn); }
    /* the previous stacked factorisation may still be reading dS on the other plan's streams */
    if (t->stacked_pending) { CHECK(qrd_stream_wait_event(p->s_main, t->ev_stacked)); t->stacked_pending = 0; }
    for (int q = 0; q < P; ++q)
        CHECK(qrd_copy_block(p->s_main, t->dRall + (size_t) q * nn, n, t->dS + (size_t) q * n, sm, n, n));
    CHECK(qrd_event_record(t->ev_gathered, p->s_main));
    qr_plan* p2 = t->p2;
    CHECK(qrd_stream_wait_event(p2->s_main, t->ev_gathered));
    CHECK(qr_geqrf_dev(p2, t->dS, sm, n, sm, t->dtau2));
    CHECK(qr_extract_r_dev(p2, t->dS, sm, n, sm, dR, n, n));
    CHECK(qrd_event_record(t->ev_stacked, p2->s_main));
    t->stacked_pending = 1;
    t->local_done = 0;
    return 0;
}

/* ---- panel-pipelined form -----------------------------------------------------------------------------------------------
 * The stacked (P n) x n factorisation is a chain of n / 32 dependent leaves (~1.2 ms at 8 x 512 columns whatever the chip does) and,
 * as ONE step behind ONE all-gather, all of it is added to the latency of a factorisation.  But block column k of a rank's R is final as
 * soon as local panel k is factored -- long before the local factorisation ends.  So the exchange and the stacked QR go block column
 * by block column, LEFT-looking, on the stacked plan's stream while the local stream carries on with its trailing update:
 *     local stream :  P_0  [pack R(:, 0)]  U_0   P_1  [pack R(:, 1)]  U_1   ...                       (single-stream tall-skinny schedule)
 *     stacked stream:        gather_0 -> S_0           gather_1 -> apply S_0's reflectors -> S_1  ...  (S_k: factor block column k)
 * Only the last block column's gather + apply + factor is exposed (~0.4 ms at 8 x 512), n / nb small all-gathers (512 KB per rank at
 * n = 512, nb = 128) instead of one.  Every rank issues the same collectives in the same order.  The result is an ordinary LAPACK-
 * layout factorisation of the stacked matrix (qr_tsqr_formq_dev does not care how it was scheduled). */
static int tsqr_local_panel(qr_tsqr_plan* t, double* dA, int lda, int pi)
{
    qr_plan* p = t->p;
    const int m = t->m_local, n = t->n, k = t->pan_k[pi], wout = t->pan_k[pi + 1] - k, mk = m - k, nt = n - (k + wout);
    use_set(p, 0);
    if (pi == 0) p->cq_parked = 0;                            /* (see geqrf_issue_inner) */
    CHECK(prof_begin(p, 2));
    p->park_hint = nt == 0 || ((size_t) nt * 8 <= (size_t) mk && p->Ye2 != NULL);
    CHECK(factor_panel(p, dA, m, lda, k, wout, t->dtau, nt > 0, NULL));
    CHECK(prof_end(p, 2.0 * mk * (double) wout * wout, 16.0 * mk * wout));
    /* block column k of R is final (its rows above the panel were finished by the earlier trailing updates): pack it */
    if (t->sent_pending[pi]) { CHECK(qrd_stream_wait_event(p->stream, t->ev_sent[pi])); t->sent_pending[pi] = 0; }
    CHECK(qrd_extract_r_block(p->stream, dA, lda, k, wout, t->dsend + (size_t) k * n, n, n));
    /* (a parked panel's diagonal block of R is still in the panel workspace: the array holds V there until the update has run) */
    if (p->cq_parked) CHECK(qrd_panel_cqr_r_block(p->stream, p->cq_ws, wout, t->dsend + (size_t) k * n + k, n));
    CHECK(qrd_event_record(t->ev_pan[pi], p->stream));
    if (nt > 0) CHECK(update_cols(p, p->stream, 0, dA, lda, k, mk, wout, k + wout, nt, p->W, NULL, p->slabs, 1, 1));
    CHECK(cq_unpark(p));
    return 0;
}

/* drecv holds every rank's block column pi: stack it, apply the reflectors of the stacked panels before it, factor it */
static int tsqr_stacked_panel(qr_tsqr_plan* t, int pi)
{
    qr_plan* p2 = t->p2;
    const int n = t->n, P = t->nranks, sm = t->sm, nb = p2->nb, k = t->pan_k[pi], wout = t->pan_k[pi + 1] - k;
    void* s = p2->s_main;
    /* rank q's block column: n x wout, packed (stride n * wout) -> rows q n .. of the stacked block column; one launch for all ranks */
    CHECK(qrd_copy_blocks(s, t->drecv, n, (size_t) n * wout, t->dS + (size_t) k * sm, sm, (size_t) n, n, wout, P));
    for (int j = 0; j < pi; ++j) {      /* (I - V_j T_j V_j^T)^T on rows pan_k[j] .. of the new block column */
        const int kj = t->pan_k[j], wj = t->pan_k[j + 1] - kj;
        CHECK(apply_small_t(p2, s, t->Vst + (size_t) j * p2->ldv * nb, p2->ldv, t->Tst + (size_t) j * p2->ldt * nb, p2->ldt, sm - kj, wj,
                            t->dS + (size_t) k * sm + (size_t) kj, sm, wout, p2->Wn, p2->Yn, p2->slabs));
    }
    p2->Vw = t->Vst + (size_t) pi * p2->ldv * nb;      /* this panel's V and T stay: the later block columns need them */
    p2->T = t->Tst + (size_t) pi * p2->ldt * nb;
    CHECK(prof_begin(p2, 2));
    const int rc = factor_panel(p2, t->dS, sm, sm, k, wout, t->dtau2, pi + 1 < t->npan, NULL);
    if (!rc) CHECK(prof_end(p2, 2.0 * (sm - k) * (double) wout * wout, 16.0 * (sm - k) * wout));
    use_set(p2, 0);
    return rc;
}

/* the last pipelined call's gathers, from its events (both streams drained by the caller): sum and longest of ev_g0 -> ev_sent, and
 * the whole call ev_t0 -> ev_t1.  A gather's interval starts when the stacked stream is past its wait for the LOCAL panel, so it holds
 * the collective itself plus the wait for the slowest rank's panel -- what pipelining has to hide */
static int tsqr_read_stats(qr_tsqr_plan* t, double* sum_ms, double* max_ms, double* step_ms)
{
    *sum_ms = *max_ms = *step_ms = 0.0;
    if (!t->stats_valid) return 0;
    for (int pi = 0; pi < t->npan; ++pi) {
        float ms = 0.0f;
        CHECK(qrd_event_elapsed_ms(t->ev_g0[pi], t->ev_sent[pi], &ms));
        *sum_ms += ms;
        if (ms > *max_ms) *max_ms = ms;
    }
    float st = 0.0f;
    CHECK(qrd_event_elapsed_ms(t->ev_t0, t->ev_t1, &st));
    *step_ms = st;
    return 0;
}

/* out[0] = sum of the gathers' intervals (ms), out[1] = the longest, out[2] = the whole call, out[3] = 1 pipelined / 0 one collective,
 * out[4] = 1 when the ranks' joint decision (MI355XQR_TSQR_PIPE unset) fell back to one collective.  Drains the plan's streams. */
int qr_tsqr_gather_stats(qr_tsqr_plan* t, double* out5)
{
    if (!t || !out5) return QR_E_ARG;
    CHECK(qr_tsqr_sync(t));
    out5[3] = (t->pipe_ok && t->nranks > 1) ? 1.0 : 0.0;
    out5[4] = t->pipe_fell_back ? 1.0 : 0.0;
    return tsqr_read_stats(t, &out5[0], &out5[1], &out5[2]);
}

/* MI355XQR_TSQR_PIPE unset: before pipelined call QR_TSQR_DECIDE_CALL every rank puts {sum of its gathers, its step} of the previous
 * (warm) call into ONE more all-gather, and every rank applies the same rule to the same 2 P numbers: if the slowest rank's gathers
 * took more than half of the fastest rank's step, n / nb small collectives cannot hide behind the local factorisation on this node,
 * and all ranks go back to the one-collective form from this call on -- together, because a rank that kept the pipelined order
 * would wait for ever in a collective the others never issue.  (A fixed call index, no timing-dependent branch before it.) */
static int tsqr_decide(qr_tsqr_plan* t, int self_gather)
{
    const int P = t->nranks;
    double mine[2], step_ms = 0.0, mx = 0.0;
    CHECK(qr_tsqr_sync(t));
    CHECK(tsqr_read_stats(t, &mine[0], &mx, &step_ms));
    mine[1] = step_ms;
    double all[2 * 64];
    if (P > 64) return 0;
    if (self_gather || !t->comm) {
        for (int q = 0; q < P; ++q) { all[2 * q] = mine[0]; all[2 * q + 1] = mine[1]; }
    } else {
        void* s2 = t->p2->s_main;
        CHECK(qrd_h2d(s2, t->dstat, mine, sizeof mine));
        CHECK(qrd_allgather_f64(t->comm, s2, t->dstat, t->dstat + 2, 2));
        CHECK(qrd_d2h(s2, all, t->dstat + 2, sizeof(double) * 2 * (size_t) P));
        CHECK(qrd_stream_sync(s2));
    }
    double gmax = 0.0, smin = 1e300;
    for (int q = 0; q < P; ++q) {
        if (all[2 * q] > gmax) gmax = all[2 * q];
        if (all[2 * q + 1] < smin) smin = all[2 * q + 1];
    }
    if (gmax > 0.5 * smin) { t->pipe_ok = 0; t->pipe_fell_back = 1; t->stats_valid = 0; }
    return 0;
}

static int tsqr_factor_pipelined(qr_tsqr_plan* t, double* dA, int lda, double* dR, int self_gather)
{
    const int n = t->n;
    void* s2 = t->p2->s_main;
    ++t->pipe_calls;
    CHECK(qrd_event_record(t->ev_t0, t->p->s_main));
    for (int pi = 0; pi < t->npan; ++pi) {
        CHECK(tsqr_local_panel(t, dA, lda, pi));
        CHECK(qrd_stream_wait_event(s2, t->ev_pan[pi]));
        CHECK(qrd_event_record(t->ev_g0[pi], s2));
        /* exactly the block column's own n x wout doubles (the half blocks at the end used to send a full n x nb from their start: the
         * tail of that range is the next block's, which the local stream may be repacking -- discarded by the receivers, but an
         * unordered read all the same) */
        const size_t cnt = (size_t) n * (size_t) (t->pan_k[pi + 1] - t->pan_k[pi]);
        if (self_gather) {
            for (int q = 0; q < t->nranks; ++q)
                CHECK(qrd_d2d(s2, t->drecv + (size_t) q * cnt, t->dsend + (size_t) t->pan_k[pi] * n, sizeof(double) * cnt));
        } else
            CHECK(qrd_allgather_f64(t->comm, s2, t->dsend + (size_t) t->pan_k[pi] * n, t->drecv, cnt));
        CHECK(qrd_event_record(t->ev_sent[pi], s2));
        t->sent_pending[pi] = 1;
        CHECK(tsqr_stacked_panel(t, pi));
    }
    CHECK(qr_extract_r_dev(t->p2, t->dS, t->sm, n, t->sm, dR, n, n));
    CHECK(qrd_event_record(t->ev_stacked, s2));
    CHECK(qrd_event_record(t->ev_t1, s2));
    t->stats_valid = 1;
    t->stacked_pending = 1;
    t->local_done = 0;
    return 0;
}

/* The same schedule over P plans of ONE device driven from one thread -- "virtual ranks": the gather is P stream-ordered device
 * copies per rank.  For tests and single-device bring-up of the pipelined form (no communicator needed: plans from
 * qr_tsqr_plan_create_comm(.., NULL, ..)); every plan must be of the same shape.  dA[r], dR[r]: rank r's shard and its copy of R. */
int qr_tsqr_factor_virtual_dev(qr_tsqr_plan** tps, int P, double** dA, int lda, double** dR)
{
    if (!tps || !dA || !dR || P < 2) return QR_E_ARG;
    for (int r = 0; r < P; ++r)
        if (!tps[r] || !dA[r] || !dR[r] || tps[r]->nranks != P || tps[r]->rank != r || !tps[r]->pipe_ok || tps[r]->n != tps[0]->n ||
            tps[r]->npan != tps[0]->npan || tps[r]->p->nb != tps[0]->p->nb || lda < tps[r]->m_local)
            return QR_E_ARG;                 /* (shards of unequal height are fine: the schedule of the exchange does not depend on them) */
    const int n = tps[0]->n;
    {
        /* P ranks on ONE device are 2 P streams, each of which may hold a one-launch panel (up to 33 -- 65 between 4097 and 16384 rows --
         * co-resident workgroups, a compute unit each): beyond the chip's compute units two such launches can each sit on part of the chip and wait for workgroups the other one
         * keeps out -- until the hand-off times out (QR_E_STALL).  A real rank has the chip to itself (2 streams); here the launch chain is
         * used instead once the streams could crowd each other. */
        int cus = 256;
        qrd_device_info(NULL, 0, &cus, NULL, NULL);
        /* (round 6: a one-launch panel of more than 4096 rows may be dealt out in 128-row workgroups: up to 65 of them) */
        int tallest = tps[0]->sm;
        for (int r = 0; r < P; ++r) if (tps[r]->m_local > tallest) tallest = tps[r]->m_local;
        const int per_launch = (tallest > 4096 && tallest <= 16384) ? 65 : 33;
        if (2 * P * per_launch > cus)
            for (int r = 0; r < P; ++r) { tps[r]->p->fused_off = 1; if (tps[r]->p2) tps[r]->p2->fused_off = 1; }
    }
    for (int pi = 0; pi < tps[0]->npan; ++pi) {
        for (int r = 0; r < P; ++r) CHECK(tsqr_local_panel(tps[r], dA[r], lda, pi));
        for (int r = 0; r < P; ++r) {
            void* s2 = tps[r]->p2->s_main;
            const size_t cnt = (size_t) n * (size_t) (tps[r]->pan_k[pi + 1] - tps[r]->pan_k[pi]);
            for (int q = 0; q < P; ++q) {
                CHECK(qrd_stream_wait_event(s2, tps[q]->ev_pan[pi]));
                CHECK(qrd_d2d(s2, tps[r]->drecv + (size_t) q * cnt, tps[q]->dsend + (size_t) tps[q]->pan_k[pi] * n, sizeof(double) * cnt));
            }
            CHECK(tsqr_stacked_panel(tps[r], pi));
        }
    }
    for (int r = 0; r < P; ++r) {
        CHECK(qr_extract_r_dev(tps[r]->p2, tps[r]->dS, tps[r]->sm, n, tps[r]->sm, dR[r], n, n));
        tps[r]->local_done = 0;
    }
    /* a bring-up driver: drained here, so that no send block is repacked while another rank's stream still reads it */
    for (int r = 0; r < P; ++r) CHECK(qr_tsqr_sync(tps[r]));
    return 0;
}

/* One rank's complete step with the collective replaced by copies of its OWN factor into every rank slot (R of [A; A; ...; A]): the
 * same launches, streams and events as a real rank of an nranks-GPU run -- which also factors the full stacked matrix redundantly --
 * minus the network.  For measuring the latency of the step on one GPU (bench.py, tsqr_model_1gpu) and for stream-order tests. */
int qr_tsqr_factor_selfgather_dev(qr_tsqr_plan* t, double* dA, int lda, double* dR)
{
    if (!t || !dA || !dR || lda < t->m_local || t->nranks < 2) return QR_E_ARG;
    /* (a diagnostic path: it neither takes part in the ranks' joint decision nor counts towards its call index -- a rank that made an
     * extra self-gather call used to reach the decision collective one call before the others) */
    if (t->pipe_ok) {
        const int calls = t->pipe_calls;
        const int rc = tsqr_factor_pipelined(t, dA, lda, dR, 1);
        t->pipe_calls = calls;
        return rc;
    }
    CHECK(qr_tsqr_local_dev(t, dA, lda));
    for (int q = 0; q < t->nranks; ++q)
        CHECK(qrd_d2d(t->p->s_main, t->dRall + (size_t) q * t->n * t->n, t->dRp, sizeof(double) * (size_t) t->n * t->n));
    return qr_tsqr_stacked_dev(t, dR);
}

/* 1 when qr_tsqr_factor_dev runs the panel-pipelined form for this plan */
int qr_tsqr_is_pipelined(qr_tsqr_plan* t) { return t && t->pipe_ok && t->nranks > 1; }

/* Exchange schedule, set by the caller instead of by the timing rule: 0 = one collective after the local factorisation, 1 = panel-pipelined
 * (QR_E_ARG when the plan's shape cannot run it), 2 = back to the library's choice (MI355XQR_TSQR_PIPE / the joint decision).  Every rank of
 * the communicator must make the same call between the same two factorisations (the schedules issue different collectives); the plan's
 * streams are drained first. */
int qr_tsqr_set_schedule(qr_tsqr_plan* t, int mode)
{
    if (!t || mode < 0 || mode > 2) return QR_E_ARG;
    if (t->nranks < 2) return 0;
    CHECK(qr_tsqr_sync(t));
    if (mode == 1 && !t->pipe_able) return QR_E_ARG;
    if (mode == 2) { t->pipe_ok = t->pipe_able && t->pipe_env_on; t->pipe_auto = t->pipe_env_auto; t->pipe_calls = 0; t->pipe_fell_back = 0; }
    else { t->pipe_ok = mode; t->pipe_auto = 0; }
    t->stats_valid = 0;
    return 0;
}

/* steps 1-3 with the RCCL exchange in between; asynchronous (qr_tsqr_sync before dR is read on another stream) */
int qr_tsqr_factor_dev(qr_tsqr_plan* t, double* dA, int lda, double* dR)
{
    if (!t || !dA || !dR || lda < t->m_local) return QR_E_ARG;
    if (t->nranks > 1 && !t->comm) return QR_E_ARG;      /* plan made for an external transport: use local / exchange_buffers / stacked */
    if (t->nranks > 1 && t->pipe_ok && t->pipe_auto && t->pipe_calls == QR_TSQR_DECIDE_CALL) CHECK(tsqr_decide(t, 0));
    if (t->nranks > 1 && t->pipe_ok) return tsqr_factor_pipelined(t, dA, lda, dR, 0);
    CHECK(qr_tsqr_local_dev(t, dA, lda));
    if (t->nranks > 1)
        /* dRall is read by the copies of the previous step on this same stream: stream order is enough */
        CHECK(qrd_allgather_f64(t->comm, t->p->s_main, t->dRp, t->dRall, (size_t) t->n * t->n));
    return qr_tsqr_stacked_dev(t, dR);
}

/* step 4 after a factorisation of dA: dQ (m_local x n, ldq) = this rank's rows of the thin Q */
int qr_tsqr_formq_dev(qr_tsqr_plan* t, const double* dA, int lda, double* dQ, int ldq)
{
    if (!t || !dA || !dQ || lda < t->m_local || ldq < t->m_local) return QR_E_ARG;
    const int n = t->n, rows = t->m_local, sm = t->sm;
    qr_plan* p = t->p;
    if (t->nranks == 1) return qr_applyq_dev(p, dA, rows, n, lda, t->dtau, dQ, n, ldq, 1);
    qr_plan* p2 = t->p2;
    if (t->ev_q && t->q_pending) { CHECK(qrd_stream_wait_event(p2->s_main, t->ev_q)); t->q_pending = 0; }   /* the previous call's copy out of dQt */
    CHECK(qr_applyq_dev(p2, t->dS, sm, n, sm, t->dtau2, t->dQt, n, sm, 1));          /* the tree's Q, (P n) x n */
    CHECK(qrd_event_record(t->ev_stacked, p2->s_main));
    CHECK(qrd_stream_wait_event(p->s_main, t->ev_stacked));
    t->stacked_pending = 0;
    CHECK(qrd_zero_block(p->s_main, dQ, ldq, rows, n));
    CHECK(qrd_copy_block(p->s_main, t->dQt + (size_t) t->rank * n, sm, dQ, ldq, n, n));
    CHECK(qrd_event_record(t->ev_q, p->s_main));
    t->q_pending = 1;
    return qr_applyq_dev(p, dA, rows, n, lda, t->dtau, dQ, n, ldq, 0);
}

/* ---- thin QR over the GPUs of one node, host pointers: a thin user of the TSQR plan above -------------------------------------
 * One host thread per device (created here); device d owns the contiguous row block d of A: H2D of its rows, qr_tsqr_factor_dev,
 * qr_tsqr_formq_dev, D2H of its rows of Q.  Communicators from ONE ncclCommInitAll in the calling thread. */
typedef struct mg_ctx {
    int rank, ngpu, dev, m, n, nb, r0, rows, rc;
    const double* A;
    double *Q, *R;
    void* comm;
    pthread_barrier_t* bar;
    int* any_fail;
} mg_ctx;

static int mg_run(mg_ctx* c)
{
    const int n = c->n, rows = c->rows;
    const size_t nn = (size_t) n * n;
    qr_tsqr_plan* t = NULL;
    double *dA = NULL, *dQ = NULL, *dR = NULL;
    int rc = qrd_set_device(c->dev);
    if (!rc) rc = qr_tsqr_plan_create_comm(&t, c->comm, c->ngpu, c->rank, rows, n, c->nb);
    if (!rc) {                  /* host-pointer entry point: poll the guard (see mmqr_status); the final qr_tsqr_sync reports QR_E_STALL, if any */
        t->p->guard_latch = 0;
        if (t->p2) t->p2->guard_latch = 0;
    }
    if (!rc) rc = qrd_malloc((void**) &dA, sizeof(double) * (size_t) rows * n);
    if (!rc) rc = qrd_malloc((void**) &dQ, sizeof(double) * (size_t) rows * n);
    if (!rc) rc = qrd_malloc((void**) &dR, sizeof(double) * nn);
    void* s = t ? qr_tsqr_stream(t) : NULL;
    /* rows [r0, r0 + rows) of the column-major host matrix (ld m) -> its own column-major array (ld rows) */
    if (!rc) rc = qrd_h2d_2d(s, dA, sizeof(double) * rows, c->A + c->r0, sizeof(double) * c->m, sizeof(double) * rows, n);
    /* every thread reaches the collective or none does: a rank that failed before it would leave the others hanging */
    if (rc) __atomic_store_n(c->any_fail, 1, __ATOMIC_SEQ_CST);
    pthread_barrier_wait(c->bar);
    if (__atomic_load_n(c->any_fail, __ATOMIC_SEQ_CST)) { if (!rc) rc = QR_E_INTERNAL; goto done; }
    rc = qr_tsqr_factor_dev(t, dA, rows, dR);
    if (!rc) rc = qr_tsqr_formq_dev(t, dA, rows, dQ, rows);
    if (!rc) rc = qrd_d2h_2d(s, c->Q + c->r0, sizeof(double) * c->m, dQ, sizeof(double) * rows, sizeof(double) * rows, n);
    if (!rc && c->rank == 0) {
        rc = qr_plan_sync(t->p2 ? t->p2 : t->p);            /* dR is written on the stacked plan's stream */
        if (!rc) rc = qrd_d2h(s, c->R, dR, sizeof(double) * nn);
    }
    if (!rc) rc = qr_tsqr_sync(t);
done:
    qrd_free(dA); qrd_free(dQ); qrd_free(dR);
    qr_tsqr_plan_destroy(t);
    return rc;
}

static void* mg_worker(void* arg)
{
    mg_ctx* c = (mg_ctx*) arg;
    c->rc = mg_run(c);
    return NULL;
}

int qr_thin_mgpu(const double* A, int m, int n, double* Q, double* R, int nb, int ngpu)
{
    if (!A || !Q || !R || n < 1 || m < n || ngpu < 1 || ngpu > QR_MAX_DEVICES) return QR_E_ARG;
    int ndev = 0;
    if (qrd_device_count(&ndev) != 0 || ndev < 1) return QR_E_NODEVICE;
    if (ngpu > ndev) return QR_E_ARG;                       /* more shards than visible devices */
    if (ngpu == 1) return qr_thin(A, m, n, Q, R, nb, 1);    /* no thread, no communicator; qr_thin's status handling (retry after a stall) */
    const int ms = (m + ngpu - 1) / ngpu;
    if (m - (ngpu - 1) * ms < n) return QR_E_ARG;           /* every shard needs at least n rows */
    int prev = 0;
    qrd_get_device(&prev);
    void* comms[QR_MAX_DEVICES];
    int devs[QR_MAX_DEVICES];
    memset(comms, 0, sizeof comms);
    for (int d = 0; d < ngpu; ++d) devs[d] = d;
    if (ngpu > 1) CHECK(qrd_comm_init_all(comms, ngpu, devs));
    pthread_barrier_t bar;
    if (pthread_barrier_init(&bar, NULL, (unsigned) ngpu)) {
        for (int d = 0; d < ngpu; ++d) qrd_comm_destroy(comms[d]);
        return QR_E_INTERNAL;
    }
    mg_ctx ctx[QR_MAX_DEVICES];
    pthread_t th[QR_MAX_DEVICES];
    int started = 0, any_fail = 0, rc = 0;
    for (int d = 0; d < ngpu; ++d) {
        mg_ctx* c = &ctx[d];
        memset(c, 0, sizeof *c);
        c->rank = d; c->ngpu = ngpu; c->dev = devs[d]; c->m = m; c->n = n; c->nb = nb;
        c->r0 = d * ms; c->rows = imin(ms, m - d * ms);
        c->A = A; c->Q = Q; c->R = R; c->comm = comms[d]; c->bar = &bar; c->any_fail = &any_fail;
    }
    {
        for (int d = 0; d < ngpu; ++d) {
            if (pthread_create(&th[d], NULL, mg_worker, &ctx[d])) break;
            ++started;
        }
        if (started < ngpu) {
            /* the running workers would wait for the missing ones at the barrier: make them bail out, then stand in */
            __atomic_store_n(&any_fail, 1, __ATOMIC_SEQ_CST);
            for (int d = started; d < ngpu; ++d) pthread_barrier_wait(&bar);
            rc = QR_E_INTERNAL;
        }
        for (int d = 0; d < started; ++d) {
            pthread_join(th[d], NULL);
            if (!rc && ctx[d].rc) rc = ctx[d].rc;
        }
    }
    for (int d = 0; d < ngpu; ++d) qrd_comm_destroy(comms[d]);
    pthread_barrier_destroy(&bar);
    qrd_set_device(prev);
    return rc;
}
