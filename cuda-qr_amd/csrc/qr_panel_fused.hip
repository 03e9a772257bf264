// qr_panel_fused.hip -- a whole outer panel (up to 256 columns = 8 leaves of 32) in ONE launch.
//
// Replaces, for panels of up to 16384 rows (8192 until the end of round 6), the per-leaf launch sequence of qr_panel_tsqr.hip (gram32 -> cholq3 -> hr3_ep -> final4 ->
// reduce' -> update: six dependent launches per 32 columns, each starting cold from L2 / HBM) -- the role of the reference's serial
// one-block panel kernel (panelHouseholderKernel, qr.cu:60-333, launched at qr.cu:518; host form qr.c:109-235) plus its in-panel apply
// (qr.c:215-235).  Same mathematics per leaf (CholeskyQR2 + Householder reconstruction, qr_leaf_math.h), different execution:
//
//   * <= 32 workgroups, all co-resident, each OWNS 256 rows of the panel for the whole launch.  The current leaf's 32 columns of those
//     rows stay in registers from one phase to the next (a -> q -> v -> the next leaf's a out of the update's accumulators): the leaf is
//     read once and written once instead of three times and twice.
//   * what the workgroups must agree on -- the 32 x 32 Gram matrices G1 = A^T A, G2 = Q^T Q, the top block of Q, the in-panel
//     product Z = Q^T [A_rest | V_prev] -- crosses workgroups through small write-through (sc1) slabs in global memory and one epoch
//     word per workgroup (MI355X_MICROARCH.md, "Valid forms": sc1 stores, every storing wave's vmcnt(0), workgroup barrier, one lane's
//     sc1 flag store; the consumer polls with sc1 loads, joins a barrier, reads with sc1 loads).  No agent-scope fence anywhere: a hop
//     costs ~1.5-2 us, the price of the kernel boundary it replaces, but nothing restarts cold behind it.
//   * every workgroup sums the partial matrices in the same order and runs the one-wave recurrences (Cholesky, modified LU, triangular
//     inverses) REDUNDANTLY on three service waves, so the small factors never have to be broadcast: all workgroups hold the same bits.
//   * the four row waves (64 rows each, one per SIMD) do everything that is row-parallel on the matrix cores in ONE register layout
//     ("L_row": lane l15 = row group, registers = the row's columns congruent to l4 mod 4; transposed products keep it):
//         q = a R1^-1,  v = q U'^-1,  C -= v W;
//     products that contract over rows (Gram matrices, Z) read a [column][row] image of the rows in LDS.
//   * the leaf's long-K product runs on Q while the service wave is in the modified LU (the "early product" of hr3_ep_kernel), is
//     corrected by the owner of the top block (z -= B^T x_top), reduced column-slice-wise by the workgroups (reduce-scatter), folded
//     (y = U'^-T z, W = T^T y / G = y^T) and gathered again; the update then takes W straight from the gathered slab.
//
// Output: exactly what the leaf loop of factor_panel (qr_host.c) leaves behind -- R and the reflector tails in A, explicit unit-lower
// V in Vw, tau, the leaves' T blocks on the diagonal of T, and the Gram blocks V_prev^T V_l in G for the T merge tree (qrd_larft).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>
#include "qr_device.h"
#include "qr_common.h"
#include "qr_leaf_math.h"
#include "qr_factor32.h"

#define PF_THREADS 256            /* four waves, one per SIMD */
// RT = rows per lane (template parameter of everything row-parallel below): 4 -> a workgroup owns 256 rows (round 4/5), 2 -> 128 rows.
// The streaming work of a leaf (deferred update + in-panel product over the rest of the panel) is matrix-core time on ONE compute unit per
// row workgroup -- 43 of a 73 us leaf at 256 columns -- so a panel short enough to be dealt out in 128-row pieces over the stream's compute
// units (mk <= 128 (cus - 1)) is: half the streaming per workgroup, the same hops (round 6).
#define PF_ROWS_OF(RT) (64 * (RT))        /* rows per workgroup */
#define PF_LDQ_OF(RT) (64 * (RT) + 4)     /* row stride of the [column][row] image (doubles) */
#define PF_MAXWG 64
#define PF_ZCOLS 224              /* columns of Z: wh - 32 <= 224 */
#define PF_SPIN_LIMIT (1u << 22)

// workspace layout (doubles); the first 2 KB are the epoch words, one per workgroup, 64 bytes apart
#define PF_FAC_WORD (16 * PF_MAXWG)   /* the factor workgroup's epoch word (unsigned index) */
#define PF_OFF_X1 1024                    /* (the epoch words: 16 * (PF_MAXWG + 1) unsigned) */
#define PF_OFF_X2 (PF_OFF_X1 + 2 * PF_MAXWG * 1024)
#define PF_OFF_QT (PF_OFF_X2 + 2 * PF_MAXWG * 1024)
#define PF_OFF_X3 (PF_OFF_QT + 2 * 1024)
#define PF_ZSLAB (32 * PF_ZCOLS + 768)   /* one workgroup's product slab: Z (32 x 224) and the partial Gram matrix of the NEXT leaf's columns */
#define PF_OFF_X4 (PF_OFF_X3 + 2 * (PF_MAXWG + 1) * PF_ZSLAB)
#define PF_OFF_XF (PF_OFF_X4 + 2 * 32 * PF_ZCOLS)
#define PF_OFF_F1 (PF_OFF_XF + 2 * PF_MAXWG * 128)      /* factor -> rows: R1^-1 (1024) + status (8), per parity */
#define PF_F1_SZ 1032
#define PF_OFF_F2 (PF_OFF_F1 + 2 * PF_F1_SZ)              /* factor -> rows, in two hand-offs: U'^-1, L1\\U', S R2 (3 x 1024; + status at 5 x 1024), then T, R (2 x 1024) */
#define PF_F2_SZ (5 * 1024 + 8)
#define PF_OFF_XG (PF_OFF_F2 + 2 * PF_F2_SZ)              /* summed Gram matrix of the next leaf's columns before their update (768) */
#define PF_OFF_XT (PF_OFF_XG + 2 * 1024)                 /* their top 32 rows before the update (32 x 32) */
#define PF_WS_DOUBLES (PF_OFF_XT + 2 * 1024)

// LDS carve-up (doubles).  The 32 x 32 factors come FIRST: their addresses are compile-time constants, and below 64 KB they fit the
// immediate offset of a ds_read -- above it hipcc materialises one scalar register per address (~900 of them, spilled, in the unrolled
// recurrences).  The row image, addressed per lane anyway, comes last.
#define PF_M33 (32 * 33)
#define PF_SM_BS 0                               /* L1 \ U' */
#define PF_SM_R2 (PF_SM_BS + PF_M33)
#define PF_SM_GS (PF_SM_R2 + PF_M33)             /* sum of the partial Gram matrices; later L1^-1 */
#define PF_SM_R1 (PF_SM_GS + PF_M33)
#define PF_SM_WS (PF_SM_R1 + PF_M33)             /* R1^-1; later U */
#define PF_SM_SS (PF_SM_WS + PF_M33)             /* S (32), 1 / diag R2 (32) */
#define PF_SM_SCR (PF_SM_SS + 64)                /* 4 waves x 128 (+ 256 for the factor workgroup's 16 x 16 intermediate) */
#define PF_SM_FLAGS (PF_SM_SCR + 4 * 128 + 256)  /* ints */
#define PF_SM_PART (PF_SM_FLAGS + 8)             /* (16 ints) */             /* 4 x 768 per-wave Gram partials; later U'^-1 and T */
#define PF_SM_IMG (PF_SM_PART + 3072)            /* [column][row] image of the workgroup's rows; later -W */
#define PF_SM_IMG_DOUBLES(RT) ((32 * PF_LDQ_OF(RT)) > (4 * PF_M33) ? (32 * PF_LDQ_OF(RT)) : (4 * PF_M33))   /* (the factor workgroup keeps four 32 x 32 matrices there) */
#define PF_SM_DOUBLES(RT) (PF_SM_IMG + PF_SM_IMG_DOUBLES(RT))

struct PfArgs {
    double* A; int lda;          // panel origin: mk rows x wh columns, factored in place
    int mk, wh;
    double* Vw; int ldv;         // explicit V, same origin
    double* T; int ldt;          // wh x wh: the leaves' T blocks go on the diagonal
    double* tau;
    double* G; int ldg;          // Gram blocks for the T merge: G(j', c + i) = V(:, j')^T V(:, c + i), j' < c (NULL: not wanted --
                                 // the in-panel product then covers A_rest only and the caller forms V^T V in one launch afterwards)
    double* ws;                  // PF_WS_DOUBLES
    unsigned epoch0;             // epoch words hold values <= epoch0 when the launch starts
    int* status;                 // [0] += leaves that took the Householder route; [1] = 1 when a wait timed out
    int merge64;                 // wh == 64 with G given: the launch also merges the two leaves' T blocks, T(0:32, 32:64) = -T_0 (V_0^T V_1) T_1
                                 // (the factor workgroup, behind the last fold: the host's two-launch merge tree is not needed)
    long long* stamps;           // development builds (-DPF_STAMPS): 32 phase stamps per leaf of workgroup 0 (100 MHz clock)
};

typedef double (*pf_m33)[33];

#ifdef PF_STAMPS
#define PF_STAMP(k) do { if (g == 0 && threadIdx.x == 0 && P.stamps) P.stamps[(c >> 5) * 32 + (k)] = (long long) __builtin_amdgcn_s_memrealtime(); } while (0)
#define PF_STAMP_S(k) do { if (g == 0 && threadIdx.x == 0 && P.stamps) P.stamps[(c >> 5) * 32 + (k)] = (long long) __builtin_amdgcn_s_memrealtime(); } while (0)
#else
#define PF_STAMP(k) do { } while (0)
#define PF_STAMP_S(k) do { } while (0)
#endif

__device__ __forceinline__ double pf_ld(const double* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void pf_st(double* p, double v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
// 16-byte write-through store (the compiler has no builtin for it; pf_publish drains these with its own s_waitcnt).  Per byte, 8-byte
// sc1 stores cost 2.7x as much on the fabric (MI355X_MICROARCH.md): the 57 KB product slabs went out in 18 us that way
__device__ __forceinline__ void pf_st2(double* p, double a, double b)
{
    const v2d v = (v2d){a, b};
    // The hazard recognizer does not look into the asm.  In front: the data usually come straight out of an MFMA (matrix-core write ->
    // vector-memory read of the same registers: up to 18 wait states for the f64 16x16x4; without them the store sent stale registers
    // and every panel wider than one leaf was wrong).  Behind: a VALU write to the data registers of a > 64-bit store needs one.
    asm volatile("s_nop 15\n\ts_nop 7\n\tglobal_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" : : "v"(p), "v"(v) : "memory");
}
// where element (i, j) of a 32 x ncols product slab lives: accumulator order, so that a wave stores whole 1 KB rows --
// [16-column tile jt][half ti][register pair][lane][2]
__device__ __forceinline__ int pf_zidx(int i, int j)
{
    const int jt = j >> 4, l15 = j & 15, ti = i >> 4, l4 = i & 3, rr = (i >> 2) & 3;
    return ((((jt * 2 + ti) * 2 + (rr >> 1)) * 64 + (l4 * 16 + l15)) << 1) + (rr & 1);
}
__device__ __forceinline__ v4d pf_mfma(double a, double b, v4d c) { return __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0); }

// every thread of the workgroup: the sc1 stores issued so far are drained, then ONE lane raises this workgroup's epoch word
__device__ __forceinline__ void pf_publish(unsigned* flags, int word, unsigned val)
{
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) __hip_atomic_store(flags + word, val, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// every thread: returns when the epoch words of the first n row workgroups (n > 0; lane w of wave 0 polls word w) or, n = 0, of the
// factor workgroup have reached val.  A wait that does not complete in ~1 s marks the launch dead -- every later wait returns at
// once, the launch ends with garbage and status[1] = 1 instead of hanging
__device__ __forceinline__ void pf_wait(const unsigned* flags, int n, unsigned val, int* dead)
{
    if (threadIdx.x < 64) {
        const int lane = threadIdx.x;
        if (*dead == 0) {
            const unsigned* w = (n > 0) ? flags + 16 * min(lane, n - 1) : flags + PF_FAC_WORD;
            unsigned spins = 0;
            for (;;) {
                const unsigned f = __hip_atomic_load(w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (__all((int) (f - val) >= 0)) break;
                __builtin_amdgcn_s_sleep(1);
                if (++spins > PF_SPIN_LIMIT) { if (lane == 0) *dead = 1; break; }
            }
        }
    }
    __syncthreads();
}

// this lane's four rows x eight columns of a leaf (L_row layout): a[t][ks] = A(r4 + t, col0 + 4 ks + l4)
template <int RT>
__device__ __forceinline__ void pf_load_rows(double (&a)[RT][8], const double* __restrict__ A, int lda, int col0, int r4c, int l4)
{
#pragma unroll
    for (int ks = 0; ks < 8; ++ks) {
        const double* p = A + (size_t) (col0 + 4 * ks + l4) * lda + r4c;
#pragma unroll
        for (int t = 0; t < RT; t += 2) {
            const v2d lo = *reinterpret_cast<const v2d*>(p + t);
            a[t][ks] = lo[0]; a[t + 1][ks] = lo[1];
        }
    }
}

template <int RT>
__device__ __forceinline__ void pf_store_rows(const double (&a)[RT][8], double* __restrict__ A, int lda, int col0, int r4, int l4)
{
#pragma unroll
    for (int ks = 0; ks < 8; ++ks) {
        double* p = A + (size_t) (col0 + 4 * ks + l4) * lda + r4;
#pragma unroll
        for (int t = 0; t < RT; t += 2) *reinterpret_cast<v2d*>(p + t) = (v2d){a[t][ks], a[t + 1][ks]};
    }
}

// [column][row] image of the workgroup's 64 RT rows: img[col * PF_LDQ_OF(RT) + row - wgrow0]
template <int RT>
__device__ __forceinline__ void pf_image_write(double* img, const double (&a)[RT][8], int wave, int l15, int l4)
{
#pragma unroll
    for (int ks = 0; ks < 8; ++ks) {
        double* p = img + (4 * ks + l4) * PF_LDQ_OF(RT) + wave * (16 * RT) + RT * l15;
#pragma unroll
        for (int t = 0; t < RT; t += 2) *reinterpret_cast<v2d*>(p + t) = (v2d){a[t][ks], a[t + 1][ks]};
    }
}

template <int RT>
__device__ __forceinline__ void pf_image_read(const double* img, double (&a)[RT][8], int wave, int l15, int l4)
{
#pragma unroll
    for (int ks = 0; ks < 8; ++ks) {
        const double* p = img + (4 * ks + l4) * PF_LDQ_OF(RT) + wave * (16 * RT) + RT * l15;
#pragma unroll
        for (int t = 0; t < RT; t += 2) {
            const v2d lo = *reinterpret_cast<const v2d*>(p + t);
            a[t][ks] = lo[0]; a[t + 1][ks] = lo[1];
        }
    }
}

// x <- x M for the lane's rows, M (32 x 32, upper triangular) in LDS as Mm[k][c]; transposed product, so the result lands in the
// layout of the input:  D[i = column][j = row] = sum_k M(k, i) x(row, k)
template <int RT>
__device__ __forceinline__ void pf_rows_times_upper(double (&a)[RT][8], pf_m33 Mm, int l15, int l4)
{
    double aw[2][8];
#pragma unroll
    for (int ti = 0; ti < 2; ++ti)
#pragma unroll
        for (int ks = 0; ks < 8; ++ks) aw[ti][ks] = Mm[4 * ks + l4][16 * ti + l15];
#pragma unroll
    for (int t = 0; t < RT; ++t) {
        v4d acc[2];
#pragma unroll
        for (int ti = 0; ti < 2; ++ti) {
            acc[ti] = (v4d){0.0, 0.0, 0.0, 0.0};
#pragma unroll
            for (int ks = 0; ks < 8; ++ks) {
                if (ti == 0 && ks >= 4) continue;            // upper triangular: rows k >= 16 of the first 16 columns are zero
                acc[ti] = pf_mfma(aw[ti][ks], a[t][ks], acc[ti]);
            }
        }
#pragma unroll
        for (int ti = 0; ti < 2; ++ti)
#pragma unroll
            for (int rr = 0; rr < 4; ++rr) a[t][4 * ti + rr] = acc[ti][rr];
    }
}

// Gram matrix of this row wave's 16 RT rows from the image: tiles (0,0), (0,1), (1,1) -> part[wave][(tile * 4 + rr) * 64 + lane]
template <int RT>
__device__ __forceinline__ void pf_gram_wave(const double* img, double* part, int wave, int lane, int l15, int l4)
{
    v4d acc[3];
#pragma unroll
    for (int i = 0; i < 3; ++i) acc[i] = (v4d){0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int sg = 0; sg < RT; ++sg) {
        const double* p0 = img + l15 * PF_LDQ_OF(RT) + wave * (16 * RT) + 16 * sg + 4 * l4;
        const double* p1 = p0 + 16 * PF_LDQ_OF(RT);
        const v2d a0 = *reinterpret_cast<const v2d*>(p0), a1 = *reinterpret_cast<const v2d*>(p0 + 2);
        const v2d b0 = *reinterpret_cast<const v2d*>(p1), b1 = *reinterpret_cast<const v2d*>(p1 + 2);
        const double f0[4] = {a0[0], a0[1], a1[0], a1[1]}, f1[4] = {b0[0], b0[1], b1[0], b1[1]};
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            acc[0] = pf_mfma(f0[s], f0[s], acc[0]);
            acc[1] = pf_mfma(f0[s], f1[s], acc[1]);
            acc[2] = pf_mfma(f1[s], f1[s], acc[2]);
        }
    }
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int rr = 0; rr < 4; ++rr) part[wave * 768 + (i * 4 + rr) * 64 + lane] = acc[i][rr];
}

// all threads: sum the four waves' partials and write this workgroup's partial Gram matrix to X with sc1 -- the three computed
// tiles in accumulator order (768 doubles: whole 256-byte rows per wave instruction)
__device__ __forceinline__ void pf_gram_publish(const double* part, double* __restrict__ X)
{
    for (int e = threadIdx.x; e < 768; e += PF_THREADS) pf_st(X + e, (part[e] + part[768 + e]) + (part[1536 + e] + part[2304 + e]));
}

__device__ __forceinline__ void pf_gram_entry(int e, double s, pf_m33 Gs, int* gflags, bool check)
{
    const int tile = e >> 8, rr = (e >> 6) & 3, ln = e & 63, p = (ln >> 4) + 4 * rr, q = ln & 15;
    const int i = (tile == 2) ? 16 + p : p, j = (tile == 0) ? q : 16 + q;
    Gs[j][i] = s;
    if (tile == 1) Gs[i][j] = s;
    if (check) {
        const double d = s - (i == j ? 1.0 : 0.0);
        if (!(fabs(d) <= QRD_GUARD_THR)) gflags[1] = 1;       // also catches NaN
        if (!(fabs(d) <= QRD_CHOL1_THR)) gflags[2] = 1;
    }
}

// 16-byte sc1 loads (no builtin; per byte they cost ~0.6 of 8-byte ones): issued by pf_ld2_issue, usable after pf_ld2_wait16 on the
// same registers (the "+v" operands keep the compiler from touching them before the wait)
__device__ __forceinline__ void pf_ld2_issue(v2d& v, const double* p)
{
    asm volatile("global_load_dwordx4 %0, %1, off sc1" : "=v"(v) : "v"(p) : "memory");
}
__device__ __forceinline__ void pf_ld2_wait16(v2d (&v)[16])
{
    asm volatile("s_waitcnt vmcnt(0)"
                 : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]), "+v"(v[4]), "+v"(v[5]), "+v"(v[6]), "+v"(v[7]), "+v"(v[8]), "+v"(v[9]),
                   "+v"(v[10]), "+v"(v[11]), "+v"(v[12]), "+v"(v[13]), "+v"(v[14]), "+v"(v[15])
                 :
                 : "memory");
}

// all threads: Gs[j][i] = sum over the workgroups (in index order) of their partial G(i, j).  A thread owns the entry pairs 2 tid and
// (tid < 128) 2 (tid + 256) and has the 16-byte loads of 16 workgroups for both in flight at once.  (NOT 32, round 6: with 64 asm loads in
// flight the register allocator parks some of their destination registers in AGPRs BEFORE the s_waitcnt -- to the compiler an asm output is
// defined when the statement ends -- and the sums read registers the loads had not reached yet: every leaf failed its guard.)
__device__ __forceinline__ void pf_gram_sum(const double* __restrict__ X, int nwg, pf_m33 Gs, int* gflags, bool check)
{
    const int e0 = 2 * threadIdx.x, e1 = 2 * (threadIdx.x + 256);
    const bool h1 = e1 < 768;
    const int e1c = h1 ? e1 : e0;
    v2d s0 = (v2d){0.0, 0.0}, s1 = s0;
    for (int w0 = 0; w0 < nwg; w0 += 16) {
        v2d v0[16], v1[16];
#pragma unroll
        for (int u = 0; u < 16; ++u) {
            const double* xw = X + (size_t) min(w0 + u, nwg - 1) * 1024;
            pf_ld2_issue(v0[u], xw + e0);
            pf_ld2_issue(v1[u], xw + e1c);
        }
        pf_ld2_wait16(v0);
        pf_ld2_wait16(v1);
#pragma unroll
        for (int u = 0; u < 16; ++u)
            if (w0 + u < nwg) { s0 += v0[u]; s1 += v1[u]; }
    }
    pf_gram_entry(e0, s0[0], Gs, gflags, check);
    pf_gram_entry(e0 + 1, s0[1], Gs, gflags, check);
    if (h1) {
        pf_gram_entry(e1, s1[0], Gs, gflags, check);
        pf_gram_entry(e1 + 1, s1[1], Gs, gflags, check);
    }
}

// ---- hand-offs between waves of ONE workgroup through an LDS word (no workgroup barrier: the other role keeps running) ----------
__device__ __forceinline__ void pf_lds_signal(int* w, int val)
{
    if ((threadIdx.x & 63) == 0) __hip_atomic_store(w, val, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
}
__device__ __forceinline__ void pf_lds_await(int* w, int val)
{
    unsigned spins = 0;
    while (__hip_atomic_load(w, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) < val) {
        __builtin_amdgcn_s_sleep(1);
        if (++spins > PF_SPIN_LIMIT) break;               // never in a correct run; keeps a broken one from hanging
    }
}

// 16 consecutive doubles of LDS (8-byte aligned) into registers, all reads in flight together: the compiler, left alone, sinks every
// LDS read of the unrolled recurrences to its use (read -> wait -> FMA, ~100 cycles per term)
__device__ __forceinline__ void pf_lds_row16(const double* p, double (&r)[16])
{
    v2d a0, a1, a2, a3, a4, a5, a6, a7;
    const unsigned addr = (unsigned) reinterpret_cast<uintptr_t>(p);
    asm volatile("ds_read2_b64 %0, %8 offset0:0 offset1:1\n\t"
                 "ds_read2_b64 %1, %8 offset0:2 offset1:3\n\t"
                 "ds_read2_b64 %2, %8 offset0:4 offset1:5\n\t"
                 "ds_read2_b64 %3, %8 offset0:6 offset1:7\n\t"
                 "ds_read2_b64 %4, %8 offset0:8 offset1:9\n\t"
                 "ds_read2_b64 %5, %8 offset0:10 offset1:11\n\t"
                 "ds_read2_b64 %6, %8 offset0:12 offset1:13\n\t"
                 "ds_read2_b64 %7, %8 offset0:14 offset1:15\n\t"
                 "s_waitcnt lgkmcnt(0)"
                 : "=&v"(a0), "=&v"(a1), "=&v"(a2), "=&v"(a3), "=&v"(a4), "=&v"(a5), "=&v"(a6), "=&v"(a7)
                 : "v"(addr)
                 : "memory");
    r[0] = a0[0]; r[1] = a0[1]; r[2] = a1[0]; r[3] = a1[1]; r[4] = a2[0]; r[5] = a2[1]; r[6] = a3[0]; r[7] = a3[1];
    r[8] = a4[0]; r[9] = a4[1]; r[10] = a5[0]; r[11] = a5[1]; r[12] = a6[0]; r[13] = a6[1]; r[14] = a7[0]; r[15] = a7[1];
}

// p[k * 33], k = 0 .. 31 (a column of a padded 32 x 32 matrix; p differs per lane), all 32 reads in flight together
__device__ __forceinline__ void pf_lds_col32(const double* p, double (&r)[32])
{
    const unsigned addr = (unsigned) reinterpret_cast<uintptr_t>(p);
    asm volatile("ds_read_b64 %0, %16\n\t"
                 "ds_read_b64 %1, %16 offset:264\n\t"  "ds_read_b64 %2, %16 offset:528\n\t"  "ds_read_b64 %3, %16 offset:792\n\t"
                 "ds_read_b64 %4, %16 offset:1056\n\t" "ds_read_b64 %5, %16 offset:1320\n\t" "ds_read_b64 %6, %16 offset:1584\n\t"
                 "ds_read_b64 %7, %16 offset:1848\n\t" "ds_read_b64 %8, %16 offset:2112\n\t" "ds_read_b64 %9, %16 offset:2376\n\t"
                 "ds_read_b64 %10, %16 offset:2640\n\t" "ds_read_b64 %11, %16 offset:2904\n\t" "ds_read_b64 %12, %16 offset:3168\n\t"
                 "ds_read_b64 %13, %16 offset:3432\n\t" "ds_read_b64 %14, %16 offset:3696\n\t" "ds_read_b64 %15, %16 offset:3960\n\t"
                 "s_waitcnt lgkmcnt(0)"
                 : "=&v"(r[0]), "=&v"(r[1]), "=&v"(r[2]), "=&v"(r[3]), "=&v"(r[4]), "=&v"(r[5]), "=&v"(r[6]), "=&v"(r[7]), "=&v"(r[8]),
                   "=&v"(r[9]), "=&v"(r[10]), "=&v"(r[11]), "=&v"(r[12]), "=&v"(r[13]), "=&v"(r[14]), "=&v"(r[15])
                 : "v"(addr)
                 : "memory");
    asm volatile("ds_read_b64 %0, %16 offset:4224\n\t"
                 "ds_read_b64 %1, %16 offset:4488\n\t"  "ds_read_b64 %2, %16 offset:4752\n\t"  "ds_read_b64 %3, %16 offset:5016\n\t"
                 "ds_read_b64 %4, %16 offset:5280\n\t" "ds_read_b64 %5, %16 offset:5544\n\t" "ds_read_b64 %6, %16 offset:5808\n\t"
                 "ds_read_b64 %7, %16 offset:6072\n\t" "ds_read_b64 %8, %16 offset:6336\n\t" "ds_read_b64 %9, %16 offset:6600\n\t"
                 "ds_read_b64 %10, %16 offset:6864\n\t" "ds_read_b64 %11, %16 offset:7128\n\t" "ds_read_b64 %12, %16 offset:7392\n\t"
                 "ds_read_b64 %13, %16 offset:7656\n\t" "ds_read_b64 %14, %16 offset:7920\n\t" "ds_read_b64 %15, %16 offset:8184\n\t"
                 "s_waitcnt lgkmcnt(0)"
                 : "=&v"(r[16]), "=&v"(r[17]), "=&v"(r[18]), "=&v"(r[19]), "=&v"(r[20]), "=&v"(r[21]), "=&v"(r[22]), "=&v"(r[23]), "=&v"(r[24]),
                   "=&v"(r[25]), "=&v"(r[26]), "=&v"(r[27]), "=&v"(r[28]), "=&v"(r[29]), "=&v"(r[30]), "=&v"(r[31])
                 : "v"(addr)
                 : "memory");
}

// inverse of the 16 x 16 upper-triangular diagonal block of Um at offset o, by columns (lane j < 16 = column j), back substitution:
// x(i) = (delta(i, j) - sum_{k > i} U(i, k) x(k)) / U(i, i).  A step's row of U arrives as one batch of LDS reads.
template <int I> struct PfUpperInv16 {
    static __device__ __forceinline__ void run(double (&x)[16], pf_m33 Um, int o, int j)
    {
        double row[16];
        pf_lds_row16(&Um[o + I][o], row);
        double acc = (I == j) ? 1.0 : 0.0;
#pragma unroll
        for (int k = I + 1; k < 16; ++k) acc -= row[k] * x[k];
        x[I] = acc * rcp_newton(row[I]);
        if constexpr (I > 0) PfUpperInv16<I - 1>::run(x, Um, o, j);
    }
};

// LDS views.  Row workgroups use: img, part (later U'^-1 and T), Ws (R1^-1), Bs (L1 \ U': owner of the top block), R1s (R: owner),
// R2s (S R2: owner), scr.  The factor workgroup uses all the 32 x 32 factors and keeps R in the (otherwise unused) image region.
struct PfLds {
    double* img; double* part;
    pf_m33 Uinv, Ts, Gs, Ls, R1s, Ws, Us, R2s, Bs, Rm;
    double *Ss, *r2inv, *scr;
    int* gflags;     // [0] first Cholesky ok, [1] |G2 - I| > 1/64, [2] > 1e-9, [3] second Cholesky failed, [4] dead (a wait timed out),
                     // [5] leaves whose LU (B, L1, U') is in LDS, [6] leaves whose inverse of U'(16:32, 16:32) is in LDS,
                     // [8 + w] leaves whose first hand-off wave w of the factor workgroup has drained (w = 1 .. 3)
};

__device__ __forceinline__ PfLds pf_lds(double* sm)
{
    PfLds L;
    L.img = sm + PF_SM_IMG;
    L.part = sm + PF_SM_PART;
    L.Uinv = reinterpret_cast<pf_m33>(sm + PF_SM_PART);
    L.Ts = reinterpret_cast<pf_m33>(sm + PF_SM_PART + PF_M33);
    L.Gs = reinterpret_cast<pf_m33>(sm + PF_SM_GS);
    L.Ls = L.Gs;
    L.R1s = reinterpret_cast<pf_m33>(sm + PF_SM_R1);
    L.Ws = reinterpret_cast<pf_m33>(sm + PF_SM_WS);
    L.Us = L.Ws;
    L.R2s = reinterpret_cast<pf_m33>(sm + PF_SM_R2);
    L.Bs = reinterpret_cast<pf_m33>(sm + PF_SM_BS);
    L.Rm = reinterpret_cast<pf_m33>(sm + PF_SM_IMG);
    L.Ss = sm + PF_SM_SS;
    L.r2inv = L.Ss + 32;
    L.scr = sm + PF_SM_SCR;
    L.gflags = reinterpret_cast<int*>(sm + PF_SM_FLAGS);
    return L;
}

struct PfLeaf {                      // per-leaf constants
    int c, nrest, ncols, gown;
    double *X1, *X2, *QT, *X3, *X4, *F1, *F2, *XG, *XT;
};

template <int RT>
__device__ __forceinline__ PfLeaf pf_leaf(const PfArgs& P, int c)
{
    PfLeaf f;
    const int li = c >> 5, par = li & 1;
    f.c = c; f.nrest = P.wh - c - 32; f.ncols = P.G ? P.wh - 32 : f.nrest; f.gown = c / PF_ROWS_OF(RT);
    f.X1 = P.ws + PF_OFF_X1 + (size_t) par * PF_MAXWG * 1024;
    f.X2 = P.ws + PF_OFF_X2 + (size_t) par * PF_MAXWG * 1024;
    f.QT = P.ws + PF_OFF_QT + (size_t) par * 1024;
    f.X3 = P.ws + PF_OFF_X3 + (size_t) par * (PF_MAXWG + 1) * PF_ZSLAB;
    f.XG = P.ws + PF_OFF_XG + (size_t) par * 1024;
    f.XT = P.ws + PF_OFF_XT + (size_t) par * 1024;
    f.X4 = P.ws + PF_OFF_X4 + (size_t) par * 32 * PF_ZCOLS;
    f.F1 = P.ws + PF_OFF_F1 + (size_t) par * PF_F1_SZ;
    f.F2 = P.ws + PF_OFF_F2 + (size_t) par * PF_F2_SZ;
    return f;
}

// all threads: a 32 x 32 matrix between LDS (padded rows) and a dense slab in global memory (sc1 both ways)
__device__ __forceinline__ void pf_m33_out(pf_m33 M, double* __restrict__ slab)
{
    for (int e = threadIdx.x; e < 1024; e += PF_THREADS) pf_st(slab + e, M[e >> 5][e & 31]);
}
// N consecutive dense 32 x 32 matrices of a slab into the LDS matrices M[0 .. N-1]: ALL loads are issued before the first LDS
// store (written as load / store pairs, the sc1 loads were not moved across the stores: four serial round trips per matrix, 10 us for
// the five matrices the owner of the top block takes)
template <int N>
__device__ __forceinline__ void pf_m33_in(pf_m33 (&M)[N], const double* __restrict__ slab)
{
    double v[N][1024 / PF_THREADS];
#pragma unroll
    for (int n = 0; n < N; ++n)
#pragma unroll
        for (int q = 0; q < 1024 / PF_THREADS; ++q) v[n][q] = pf_ld(slab + n * 1024 + threadIdx.x + q * PF_THREADS);
#pragma unroll
    for (int n = 0; n < N; ++n)
#pragma unroll
        for (int q = 0; q < 1024 / PF_THREADS; ++q) {
            const int e = threadIdx.x + q * PF_THREADS;
            M[n][e >> 5][e & 31] = v[n][q];
        }
}

// reduce-scatter of Z (row workgroups, four waves): this workgroup's columns j = g, g + nrow, ... of Z, two per wave at a time
// (half-wave = column):  z = sum of the partials + the owner's correction,  y = U'^-T z,  W(:, j) = T^T y -> X4  or  G(j', c + :) = y
__device__ __forceinline__ void pf_fold(const PfArgs& P, const PfLeaf& f, const PfLds& L, int g, int nwg, int wave, int lane, bool nocorr)
{
    double* s1 = L.scr + wave * 128;
    const int h = lane >> 5, i = lane & 31;
    for (int k = 2 * wave + h; ; k += 8) {
        const int j = g + k * nwg;
        const bool have = j < f.ncols;
        if (!__any(have)) break;
        if (have) {
            double z = 0.0;
            const int zi = pf_zidx(i, j);
            const double corr = nocorr ? 0.0 : pf_ld(f.X3 + (size_t) nwg * PF_ZSLAB + zi);   // the top-block owner's -B^T x_top
            for (int u0 = 0; u0 < nwg; u0 += 32) {            // (32 loads in flight per batch; one batch up to 32 row workgroups)
                double v[32];
#pragma unroll
                for (int u = 0; u < 32; ++u) v[u] = pf_ld(f.X3 + (size_t) min(u0 + u, nwg - 1) * PF_ZSLAB + zi);
#pragma unroll
                for (int u = 0; u < 32; ++u)
                    if (u0 + u < nwg) z += v[u];
            }
            s1[h * 32 + i] = z + corr;
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        double y = 0.0;
        double mcol[32], vec[32];
        if (have) {
            pf_lds_col32(&L.Uinv[0][i], mcol);                                            // U'^-1 is stored with its zeros
            pf_lds_row16(s1 + h * 32, *reinterpret_cast<double (*)[16]>(&vec[0]));
            pf_lds_row16(s1 + h * 32 + 16, *reinterpret_cast<double (*)[16]>(&vec[16]));
#pragma unroll
            for (int kk = 0; kk < 32; ++kk) y += mcol[kk] * vec[kk];
            s1[64 + h * 32 + i] = y;
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        if (have) {
            if (j >= f.nrest) {
                pf_st(P.G + (size_t) (f.c + i) * P.ldg + (j - f.nrest), y);     // (write-through: with merge64 the factor workgroup reads the block in this launch)
            } else {
                double wv = 0.0;
                pf_lds_col32(&L.Ts[0][i], mcol);                                          // T is stored with its zeros
                pf_lds_row16(s1 + 64 + h * 32, *reinterpret_cast<double (*)[16]>(&vec[0]));
                pf_lds_row16(s1 + 64 + h * 32 + 16, *reinterpret_cast<double (*)[16]>(&vec[16]));
#pragma unroll
                for (int cc = 0; cc < 32; ++cc) wv += mcol[cc] * vec[cc];
                pf_st(f.X4 + j * 32 + i, -wv);                                           // -W: the updates then ADD V (-W) with no operation on a loaded value
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
    }
    // ... and this workgroup's slice of G' = A_next^T A_next (the next leaf's columns BEFORE this leaf's update, rows >= c): entries
    // g, g + nwg, ... of the 768; the factor workgroup turns the sum into the next leaf's G1 without another exchange (pf_factor_wg)
    if (f.nrest > 0) {
        for (int e = g + nwg * (int) threadIdx.x; e < 768; e += nwg * PF_THREADS) {
            double z = 0.0;
            for (int u0 = 0; u0 < nwg; u0 += 32) {
                double v[32];
#pragma unroll
                for (int u = 0; u < 32; ++u) v[u] = pf_ld(f.X3 + (size_t) min(u0 + u, nwg - 1) * PF_ZSLAB + 32 * PF_ZCOLS + e);
#pragma unroll
                for (int u = 0; u < 32; ++u)
                    if (u0 + u < nwg) z += v[u];
            }
            pf_st(f.XG + e, z);
        }
    }
}

// Row waves: A_rest(:, 32 g .. ) -= V W for this lane's rows and the column groups g1 - 1 down to g0 (32 columns each) of the leaf at
// column c: V in vr (L_row layout), W straight from the gathered slab X4 (sc1 loads, the next group's requested a group ahead, as is
// the next 64 x 16 piece of A_rest).  keep: the LAST group's result (group g0) is returned in vr instead of V (it is the next leaf's a).
template <int RT>
__device__ __forceinline__ void pf_update_groups(double (&vr)[RT][8], const double* __restrict__ X4, double* __restrict__ A, int lda, int c,
                                                 int g0, int g1, bool keep, int r4, int r4c, bool act, int l15, int l4)
{
    if (g1 <= g0) return;
    double aw[2][8], awn[2][8];
    v4d ca[RT], cbn[RT];
    auto wload = [&](double (&w)[2][8], int jg) {
#pragma unroll
        for (int ti = 0; ti < 2; ++ti)
#pragma unroll
            for (int ks = 0; ks < 8; ++ks) w[ti][ks] = pf_ld(X4 + (32 * jg + 16 * ti + l15) * 32 + 4 * ks + l4);      // X4 holds -W
    };
    auto cptr = [&](int jg, int ti) { return A + (size_t) (c + 32 + 32 * jg + 16 * ti + l4) * lda; };
    auto cload = [&](v4d (&cc)[RT], const double* cp) {
#pragma unroll
        for (int rr = 0; rr < 4; ++rr) {
            const double* q = cp + (size_t) (4 * rr) * lda + r4c;
#pragma unroll
            for (int t = 0; t < RT; t += 2) {
                const v2d lo = *reinterpret_cast<const v2d*>(q + t);
                cc[t][rr] = lo[0]; cc[t + 1][rr] = lo[1];
            }
        }
    };
    auto cstore = [&](const v4d (&cc)[RT], double* cp) {
        if (act) {
#pragma unroll
            for (int rr = 0; rr < 4; ++rr) {
                double* q = cp + (size_t) (4 * rr) * lda + r4;
#pragma unroll
                for (int t = 0; t < RT; t += 2) *reinterpret_cast<v2d*>(q + t) = (v2d){cc[t][rr], cc[t + 1][rr]};
            }
        }
    };
    auto mma = [&](v4d (&cc)[RT], const double (&w)[8]) {
#pragma unroll
        for (int t = 0; t < RT; ++t)
#pragma unroll
            for (int ks = 0; ks < 8; ++ks) cc[t] = pf_mfma(w[ks], vr[t][ks], cc[t]);
    };
    wload(aw, g1 - 1);
    cload(ca, cptr(g1 - 1, 0));
    for (int jg = g1 - 1; jg > g0; --jg) {
        double* const cp0 = cptr(jg, 0);
        double* const cp1 = cptr(jg, 1);
        cload(cbn, cp1);
        wload(awn, jg - 1);
        mma(ca, aw[0]);
        cstore(ca, cp0);
        cload(ca, cptr(jg - 1, 0));
        mma(cbn, aw[1]);
        cstore(cbn, cp1);
#pragma unroll
        for (int ti = 0; ti < 2; ++ti)
#pragma unroll
            for (int ks = 0; ks < 8; ++ks) aw[ti][ks] = awn[ti][ks];
    }
    {
        double* const cp0 = cptr(g0, 0);
        double* const cp1 = cptr(g0, 1);
        cload(cbn, cp1);
        mma(ca, aw[0]);
        cstore(ca, cp0);
        mma(cbn, aw[1]);
        cstore(cbn, cp1);
        if (keep) {
#pragma unroll
            for (int t = 0; t < RT; ++t)
#pragma unroll
                for (int rr = 0; rr < 4; ++rr) { vr[t][rr] = ca[t][rr]; vr[t][4 + rr] = cbn[t][rr]; }
        }
    }
}

// The next leaf's 32 columns (group 0), in two steps around the wait for W: the 64 x 32 piece of A_rest is requested before the wait.
template <int RT>
__device__ __forceinline__ void pf_update0_load(v4d (&ca)[RT], v4d (&cb)[RT], const double* __restrict__ A, int lda, int c, int r4c, int l4)
{
#pragma unroll
    for (int ti = 0; ti < 2; ++ti) {
        const double* cp = A + (size_t) (c + 32 + 16 * ti + l4) * lda;
#pragma unroll
        for (int rr = 0; rr < 4; ++rr) {
            const double* q = cp + (size_t) (4 * rr) * lda + r4c;
            v4d (&cc)[RT] = ti ? cb : ca;
#pragma unroll
            for (int t = 0; t < RT; t += 2) {
                const v2d lo = *reinterpret_cast<const v2d*>(q + t);
                cc[t][rr] = lo[0]; cc[t + 1][rr] = lo[1];
            }
        }
    }
}
template <int RT>
__device__ __forceinline__ void pf_update0_finish(double (&vr)[RT][8], v4d (&ca)[RT], v4d (&cb)[RT], const double* __restrict__ X4, double* __restrict__ A,
                                                  int lda, int c, int r4, bool act, int l15, int l4)
{
    double aw[2][8];
#pragma unroll
    for (int ti = 0; ti < 2; ++ti)
#pragma unroll
        for (int ks = 0; ks < 8; ++ks) aw[ti][ks] = pf_ld(X4 + (16 * ti + l15) * 32 + 4 * ks + l4);                 // X4 holds -W
#pragma unroll
    for (int t = 0; t < RT; ++t)
#pragma unroll
        for (int ks = 0; ks < 8; ++ks) { ca[t] = pf_mfma(aw[0][ks], vr[t][ks], ca[t]); cb[t] = pf_mfma(aw[1][ks], vr[t][ks], cb[t]); }
    if (act) {
#pragma unroll
        for (int ti = 0; ti < 2; ++ti) {
            double* cp = A + (size_t) (c + 32 + 16 * ti + l4) * lda;
            v4d (&cc)[RT] = ti ? cb : ca;
#pragma unroll
            for (int rr = 0; rr < 4; ++rr) {
                double* q = cp + (size_t) (4 * rr) * lda + r4;
#pragma unroll
                for (int t = 0; t < RT; t += 2) *reinterpret_cast<v2d*>(q + t) = (v2d){cc[t][rr], cc[t + 1][rr]};
            }
        }
    }
#pragma unroll
    for (int t = 0; t < RT; ++t)
#pragma unroll
        for (int rr = 0; rr < 4; ++rr) { vr[t][rr] = ca[t][rr]; vr[t][4 + rr] = cb[t][rr]; }
}

// ---------------------------------------------------------------------------------------------------------------------------------
// The Householder route of ONE leaf inside the launch (the guard of the CholeskyQR2 route: a non-positive Cholesky pivot, or
// Q^T Q too far from I -- zero, dependent or badly conditioned columns).  Column by column, the classical way (qr.c:109-235): the row
// workgroups exchange 32 partial dot products per column (x_J^T x_c for all c: the norm, every v^T a, and the Gram column that T needs,
// as in house_step of qr_panel_tsqr.hip) and every workgroup forms tau, beta and the update coefficients redundantly from the same
// sums.  32 exchanges of ~2.5 us: slow, rare, and the output has the same form as the fast route's (unit-lower V below R, tau, T).
// On entry nothing of the leaf has been written; on exit: ar = this lane's rows of V (top block: unit-lower L1), V / R stored, T and tau
// stored (owner), Ts = T and Uinv = I in LDS.  er0: the row workgroups' epoch value before the first column exchange.
// ---------------------------------------------------------------------------------------------------------------------------------
struct PfTau { double tau[32]; };

template <int RT, int J>
__device__ __forceinline__ void pf_house_col(double (&a)[RT][8], const PfArgs& P, const PfLeaf& f, const PfLds& L, unsigned* flags, unsigned er0,
                                             int g, int nrow, int r4, int l15, int l4, int wave)
{
    constexpr int KJ = J >> 2, LJ = J & 3;
    const int tid = threadIdx.x, c = f.c;
    double* red = L.part;                     // [4][32] per-wave sums
    double* sc = L.part + 128;                // [32] s_c, then [3] tau, beta, 1/u
    double* XF = P.ws + PF_OFF_XF + (size_t) (J & 1) * PF_MAXWG * 128;
    // x_J of this lane's four rows: held by the lanes with l4 == J % 4
    double xj[RT], p[8];
#pragma unroll
    for (int t = 0; t < RT; ++t) {
        const double v = __shfl(a[t][KJ], l15 + 16 * LJ);
        xj[t] = (r4 + t > c + J) ? v : 0.0;                 // rows strictly below the pivot row (rows beyond mk hold zeros)
    }
#pragma unroll
    for (int ks = 0; ks < 8; ++ks) {
        if constexpr (RT == 4) p[ks] = (xj[0] * a[0][ks] + xj[1] * a[1][ks]) + (xj[2] * a[2][ks] + xj[3] * a[3][ks]);
        else p[ks] = xj[0] * a[0][ks] + xj[1] * a[1][ks];
    }
#pragma unroll
    for (int ks = 0; ks < 8; ++ks) {                        // sum over the 16 lanes of this l4 (the 64 rows of the wave)
        p[ks] += __shfl_xor(p[ks], 1); p[ks] += __shfl_xor(p[ks], 2); p[ks] += __shfl_xor(p[ks], 4); p[ks] += __shfl_xor(p[ks], 8);
    }
    if (l15 == 0) {
#pragma unroll
        for (int ks = 0; ks < 8; ++ks) red[wave * 32 + 4 * ks + l4] = p[ks];
    }
    if (g == f.gown && r4 == c + (J & ~(RT - 1))) {         // the pivot row: row c + J = r4 + J % RT of these four lanes (one per l4)
#pragma unroll
        for (int ks = 0; ks < 8; ++ks) pf_st(XF + (size_t) g * 128 + 32 + 4 * ks + l4, a[J & (RT - 1)][ks]);
    }
    __syncthreads();
    if (tid < 32) pf_st(XF + (size_t) g * 128 + tid, (red[tid] + red[32 + tid]) + (red[64 + tid] + red[96 + tid]));
    pf_publish(flags, 16 * g, er0 + 1u + (unsigned) J);
    pf_wait(flags, nrow, er0 + 1u + (unsigned) J, &L.gflags[4]);
    if (tid < 32) {
        double d = 0.0;
        for (int w0 = 0; w0 < nrow; w0 += 8) {
            double v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = pf_ld(XF + (size_t) min(w0 + u, nrow - 1) * 128 + tid);
#pragma unroll
            for (int u = 0; u < 8; ++u)
                if (w0 + u < nrow) d += v[u];
        }
        const double rowv = pf_ld(XF + (size_t) f.gown * 128 + 32 + tid);
        const double alpha = __shfl(rowv, J), sigma = __shfl(d, J);
        double t, b, iu;
        if (sigma == 0.0) { t = 0.0; b = alpha; iu = 0.0; }
        else {
            const double nrm = sqrt(alpha * alpha + sigma);
            b = -copysign(nrm, alpha);
            t = (b - alpha) / b;
            iu = 1.0 / (alpha - b);
        }
        const double sv = rowv + d * iu;
        sc[tid] = sv;
        if (tid < J) L.Bs[J][tid] = sv;                       // Z(c', J) = v_c'^T v_J (Bs is free on this route)
        if (tid == 0) { sc[32] = t; sc[33] = b; sc[34] = iu; L.Ss[J] = t; }
    }
    __syncthreads();
    const double tj = sc[32], beta = sc[33], iu = sc[34];
#pragma unroll
    for (int t = 0; t < RT; ++t) {
        const int row = r4 + t;
        const bool below = row > c + J && row < P.mk, piv = row == c + J;
        const double vi = below ? xj[t] * iu : (piv ? 1.0 : 0.0);
        const double coef = tj * vi;
#pragma unroll
        for (int ks = 0; ks < 8; ++ks) {
            const int col = 4 * ks + l4;
            if (ks > KJ || (ks == KJ && l4 > LJ)) a[t][ks] -= coef * sc[col];
            else if (ks == KJ && l4 == LJ) a[t][ks] = below ? vi : (piv ? beta : a[t][ks]);
        }
    }
    __syncthreads();                                          // sc / red are rewritten by the next column
    if constexpr (J + 1 < 32) pf_house_col<RT, J + 1>(a, P, f, L, flags, er0, g, nrow, r4, l15, l4, wave);
}

template <int RT>
__device__ __forceinline__ void pf_householder_leaf(double (&ar)[RT][8], const PfArgs& P, const PfLeaf& f, const PfLds& L, unsigned* flags,
                                                    unsigned er0, int g, int nrow, int r4, int r4c, bool act, bool toprow, int l15, int l4,
                                                    int wave)
{
    const int tid = threadIdx.x, c = f.c;
    pf_load_rows<RT>(ar, P.A, P.lda, c, r4c, l4);             // the leaf as the previous update left it
    if (!act) {
#pragma unroll
        for (int t = 0; t < RT; ++t)
#pragma unroll
            for (int ks = 0; ks < 8; ++ks) ar[t][ks] = 0.0;
    }
    __syncthreads();
    pf_house_col<RT, 0>(ar, P, f, L, flags, er0, g, nrow, r4, l15, l4, wave);
    // T from tau and the Gram columns (row p of T depends on row p only: thread p), U'^-1 := I
    {
        PfTau th;
#pragma unroll
        for (int q = 0; q < 32; ++q) th.tau[q] = L.Ss[q];
        __syncthreads();
        build_t_rows(L.Ts, L.Bs, th, 32, tid);
        for (int e = tid; e < 1024; e += PF_THREADS) L.Uinv[e >> 5][e & 31] = ((e >> 5) == (e & 31)) ? 1.0 : 0.0;
    }
    __syncthreads();
    if (g == f.gown) {
#pragma unroll
        for (int q = 0; q < 1024 / PF_THREADS; ++q) {
            const int el = tid + q * PF_THREADS, i = el & 31, cc = el >> 5;
            P.T[(size_t) (c + cc) * P.ldt + c + i] = L.Ts[i][cc];
            if (i == cc) P.tau[c + i] = L.Ss[i];
        }
        if (P.merge64) pf_m33_out(L.Ts, f.F2 + 3 * 1024);    // where the factor workgroup's T would be: it merges the panel's T from there
    }
    if (toprow) {                                             // R on and above the diagonal, reflector tails below it: LAPACK's in-place form
#pragma unroll
        for (int t = 0; t < RT; ++t)
#pragma unroll
            for (int ks = 0; ks < 8; ++ks) {
                const int rr_ = r4 + t - c, col = 4 * ks + l4;
                const double v = ar[t][ks];
                P.A[(size_t) (c + col) * P.lda + c + rr_] = v;
                const double l1 = (col < rr_) ? v : (col == rr_ ? 1.0 : 0.0);
                P.Vw[(size_t) (c + col) * P.ldv + c + rr_] = l1;
                ar[t][ks] = l1;
            }
    } else if (act) {
        pf_store_rows<RT>(ar, P.Vw, P.ldv, c, r4, l4);
        pf_store_rows<RT>(ar, P.A, P.lda, c, r4, l4);
    }
}

// ---------------------------------------------------------------------------------------------------------------------------------
// A ROW workgroup (four waves, 64 rows each).  It never runs a recurrence: the 32 x 32 factors come from the factor workgroup, which
// has a compute unit to itself -- the first version ran them on three extra waves of every row workgroup, and beside a row wave's f64
// MFMAs the f64 VALU chain of the LU took 2.5 x as long (24.6 us instead of 9.6: the two share the SIMD's double-precision datapath),
// so that nothing could be hidden behind anything.  Epoch protocol per leaf (PfLeaf): rows publish + 1 (G1 partial), + 2 (G2 partial,
// Q_top), + 3 (Z partials; the owner of the top block also its correction, R, L1, T, tau), + 4 (W slices); the factor workgroup
// publishes + 1 (R1^-1), + 2 (U'^-1, L1 \ U', S R2 and the leaf's verdict: as soon as the LU is done) and + 3 (T, R: round 6 -- they
// used to travel with + 2, 4.8 us later than V needs U'^-1).
// ---------------------------------------------------------------------------------------------------------------------------------
template <int RT>
__device__ __forceinline__ void pf_row_wg(const PfArgs& P, double* sm, int g, int nrow)
{
    constexpr int PF_ROWS = PF_ROWS_OF(RT), PF_LDQ = PF_LDQ_OF(RT);
    const PfLds L = pf_lds(sm);
    const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // (scalar: "this wave owns that tile" tests become uniform branches, not exec-mask blocks)
    const int mk = P.mk, wh = P.wh, lda = P.lda, ldv = P.ldv;
    double* const A = P.A;
    double* const Vw = P.Vw;
    unsigned* const flags = reinterpret_cast<unsigned*>(P.ws);
    const int wgrow0 = g * PF_ROWS;
    unsigned er = P.epoch0, ef = P.epoch0;                 // epoch values of the row workgroups / the factor workgroup before this leaf
    bool g1_derived = false;                               // the factor workgroup derives this leaf's G1 from the previous leaf's G' and R12
    double ar[RT][8];
    if (tid == 0) L.gflags[4] = 0;
    {
        const int l15 = tid & 15, l4 = (tid & 63) >> 4;
        pf_load_rows<RT>(ar, A, lda, 0, min(wgrow0 + wave * (16 * RT) + RT * l15, mk - RT), l4);
    }
    for (int c = 0; c < wh; c += 32) {
        const PfLeaf f = pf_leaf<RT>(P, c);
        int lane = tid & 63;                                  // opaque once per leaf: keeps the lane-dependent addresses and selects
        asm volatile("" : "+v"(lane));                        // of the unrolled bodies below from being hoisted out of this loop
        const int l15 = lane & 15, l4 = lane >> 4;
        const int r4 = wgrow0 + wave * (16 * RT) + RT * l15; // this lane's RT rows r4 .. r4 + RT - 1
        const int r4c = min(r4, mk - RT);
        const bool act = r4 >= c && r4 < mk;
        const bool own = g == f.gown;
        const bool toprow = own && r4 >= c && r4 < c + 32;
        PF_STAMP(0);
        // ---- G1 = A^T A: image of the leaf's rows, partial Gram per wave, workgroup partial -> X1
        if (!act) {
#pragma unroll
            for (int t = 0; t < RT; ++t)
#pragma unroll
                for (int ks = 0; ks < 8; ++ks) ar[t][ks] = 0.0;
        }
        pf_image_write<RT>(L.img, ar, wave, l15, l4);
        __syncthreads();
        if (!g1_derived) {                                    // first leaf of the panel, or the one after a Householder-route leaf
            pf_gram_wave<RT>(L.img, L.part, wave, lane, l15, l4);
            __syncthreads();
            pf_gram_publish(L.part, f.X1 + (size_t) g * 1024);
        }
        pf_publish(flags, 16 * g, er + 1);                    // (an empty publish when G1 is derived: the epoch count stays uniform)
        PF_STAMP(1);
        pf_wait(flags, 0, ef + 1, &L.gflags[4]);
        PF_STAMP(3);
        { pf_m33 m1[1] = {L.Ws}; pf_m33_in<1>(m1, f.F1); }    // R1^-1
        __syncthreads();
        // ---- Q = A R1^-1 (registers), its image, the top block of Q -> QT, partial G2 = Q^T Q -> X2
        pf_rows_times_upper<RT>(ar, L.Ws, l15, l4);
        pf_image_write<RT>(L.img, ar, wave, l15, l4);
        if (toprow) {
#pragma unroll
            for (int ks = 0; ks < 8; ++ks)
#pragma unroll
                for (int t = 0; t < RT; ++t) pf_st(f.QT + (4 * ks + l4) * 32 + (r4 + t - c), ar[t][ks]);
        }
        __syncthreads();
        pf_gram_wave<RT>(L.img, L.part, wave, lane, l15, l4);
        __syncthreads();
        pf_gram_publish(L.part, f.X2 + (size_t) g * 1024);
        pf_publish(flags, 16 * g, er + 2);
        PF_STAMP(4);
        // ---- (the factor workgroup: G2, modified LU, triangular inverses -- ~17 us.)  The PREVIOUS leaf's update of everything beyond
        // this leaf's columns happens HERE (round 5; until then it sat in front of the wait for R1^-1, which the matrix-core Cholesky now
        // publishes 2.4 us into the leaf: the 9-24 us of this update had become part of the critical chain).  This leaf only needed its
        // own 32 columns (done at the end of the previous pass); the product below needs the rest.  V of the previous leaf comes back
        // from Vw (this lane's own rows); this leaf's q is parked in its LDS image meanwhile
        if (c > 0 && f.nrest > 0) {
            const int cp_ = c - 32;
            const bool actp = r4 >= cp_ && r4 < mk;
            pf_load_rows<RT>(ar, Vw, ldv, cp_, r4c, l4);
            if (!actp) {
#pragma unroll
                for (int t = 0; t < RT; ++t)
#pragma unroll
                    for (int ks = 0; ks < 8; ++ks) ar[t][ks] = 0.0;
            }
            pf_update_groups<RT>(ar, P.ws + PF_OFF_X4 + (size_t) (((cp_ >> 5) & 1)) * 32 * PF_ZCOLS, A, lda, cp_, 1, (wh - cp_ - 32) / 32, false,
                             r4, r4c, actp, l15, l4);
            pf_image_read<RT>(L.img, ar, wave, l15, l4);
            __syncthreads();                                  // the product's waves read rows other waves have just updated
        }
        PF_STAMP(2);
        // ---- (the factor workgroup: G2, modified LU, triangular inverses.)  Meanwhile  Z = Q^T [A_rest | V_prev]  for this workgroup's
        // rows, 16-column tiles dealt to the waves.  A tile's 256 rows go by in four chunks of 64; the rows of the next chunk (or of
        // the next tile) are requested before the 32 matrix-core instructions of the current one, and a tile is published as soon as
        // it is complete (16-byte write-through stores in accumulator order)
        const int ntile = f.ncols / 16;
        auto product = [&]() {
            const int nval = (ntile > wave) ? (ntile - wave + 3) / 4 : 0;
            double* X3g = f.X3 + (size_t) g * PF_ZSLAB;
            double xb[2][4][4];
            auto xptr = [&](int slot) {
                const int j = 16 * (wave + 4 * slot) + l15;
                return (j < f.nrest) ? A + (size_t) (c + 32 + j) * lda : Vw + (size_t) (j - f.nrest) * ldv;
            };
            auto xload = [&](double (&x)[4][4], const double* xp, int ch) {
#pragma unroll
                for (int s4 = 0; s4 < 4; ++s4) {
                    const int row = wgrow0 + 16 * (4 * ch + s4) + 4 * l4, rowc = min(row, mk - 4);
                    const v2d lo = *reinterpret_cast<const v2d*>(xp + rowc), hi = *reinterpret_cast<const v2d*>(xp + rowc + 2);
                    // no mask here: a select on a loaded value is a use of it, and the wait for the load would land in front of the MFMAs
                    // the load is meant to run under.  Z needs none (Q is zero outside the leaf's rows, the operands there are finite);
                    // the Gram tiles mask their operands where they are used (gmma)
                    x[s4][0] = lo[0]; x[s4][1] = lo[1]; x[s4][2] = hi[0]; x[s4][3] = hi[1];
                }
            };
            // G' = A_next^T A_next (the next leaf's columns as they are NOW, rows >= c), one 16 x 16 tile per wave beside its product:
            // wave 0 tile (0,0), wave 1 tile (1,1) from the operand registers they hold anyway, wave 2 tile (0,1) in a pass of its own.
            // With R12 = the top 32 rows of those columns after this leaf's update, the next leaf's Gram matrix is G' - R12^T R12
            // (the update is orthogonal on rows >= c): its all-to-all exchange and 200 KB sum disappear (pf_factor_wg)
            const bool want_g = f.nrest > 0;
            v4d gacc = (v4d){0.0, 0.0, 0.0, 0.0};
            auto gmma = [&](const double (&xa)[4][4], const double (&xc)[4][4], int ch) {
#pragma unroll
                for (int s4 = 0; s4 < 4; ++s4) {
                    const int row = wgrow0 + 16 * (4 * ch + s4) + 4 * l4;
                    const bool on = row >= c && row < mk;            // rows of this leaf only
#pragma unroll
                    for (int kk = 0; kk < 4; ++kk) gacc = pf_mfma(on ? xa[s4][kk] : 0.0, xc[s4][kk], gacc);
                }
            };
            auto mma = [&](const double (&x)[4][4], int ch, v4d& acc0, v4d& acc1) {
#pragma unroll
                for (int s4 = 0; s4 < 4; ++s4) {
                    const double* q0 = L.img + l15 * PF_LDQ + 16 * (4 * ch + s4) + 4 * l4;
                    const double* q1 = q0 + 16 * PF_LDQ;
                    const v2d qa0 = *reinterpret_cast<const v2d*>(q0), qa1 = *reinterpret_cast<const v2d*>(q0 + 2);
                    const v2d qb0 = *reinterpret_cast<const v2d*>(q1), qb1 = *reinterpret_cast<const v2d*>(q1 + 2);
                    acc0 = pf_mfma(qa0[0], x[s4][0], acc0); acc1 = pf_mfma(qb0[0], x[s4][0], acc1);
                    acc0 = pf_mfma(qa0[1], x[s4][1], acc0); acc1 = pf_mfma(qb0[1], x[s4][1], acc1);
                    acc0 = pf_mfma(qa1[0], x[s4][2], acc0); acc1 = pf_mfma(qb1[0], x[s4][2], acc1);
                    acc0 = pf_mfma(qa1[1], x[s4][3], acc0); acc1 = pf_mfma(qb1[1], x[s4][3], acc1);
                }
            };
            static_assert(RT == 2 || RT == 4, "the chunk pipeline below alternates two operand buffers: an even number of 64-row chunks per tile");
            if (nval > 0) xload(xb[0], xptr(0), 0);
            for (int slot = 0; slot < nval; ++slot) {
                v4d acc0 = (v4d){0.0, 0.0, 0.0, 0.0}, acc1 = acc0;
                const double* xp = xptr(slot);
                const bool gdiag = want_g && slot == 0 && wave < 2;       // wave-uniform
#pragma unroll
                for (int ch = 0; ch < RT; ++ch) {                        // the next chunk (or the next tile's first) is requested before this chunk's MFMAs
                    if (ch + 1 < RT) xload(xb[(ch + 1) & 1], xp, ch + 1);
                    else if (slot + 1 < nval) xload(xb[0], xptr(slot + 1), 0);
                    mma(xb[ch & 1], ch, acc0, acc1);
                    if (gdiag) gmma(xb[ch & 1], xb[ch & 1], ch);
                }
                double* zp = X3g + (wave + 4 * slot) * 512 + 2 * lane;          // pf_zidx layout
                pf_st2(zp, acc0[0], acc0[1]); pf_st2(zp + 128, acc0[2], acc0[3]);
                pf_st2(zp + 256, acc1[0], acc1[1]); pf_st2(zp + 384, acc1[2], acc1[3]);
            }
            if (want_g && wave == 2) {                           // tile (0, 1): both 16-column tiles of the next leaf, chunk by chunk
                double xc[4][4];
                const double* x0 = A + (size_t) (c + 32 + l15) * lda;
                const double* x1 = x0 + (size_t) 16 * lda;
#pragma unroll
                for (int ch = 0; ch < RT; ++ch) { xload(xb[0], x0, ch); xload(xc, x1, ch); gmma(xb[0], xc, ch); }
            }
            if (want_g && wave < 3) {                            // accumulator order, as the Gram partials: tile 0, 2, 1 for wave 0, 1, 2
                const int tile = (wave == 0) ? 0 : (wave == 1 ? 2 : 1);
#pragma unroll
                for (int rr = 0; rr < 4; ++rr) pf_st(X3g + 32 * PF_ZCOLS + (tile * 4 + rr) * 64 + lane, gacc[rr]);
            }
        };
        product();
        // the owner of the top block: x_top (its own rows c .. c + 31 of [A_rest | V_prev], final since the deferred update) for the
        // correction below -- requested now, so that only 16 MFMAs per tile remain once the factors are there
        double xt[4][8];
        if (own && ntile > 0) {
#pragma unroll
            for (int slot = 0; slot < 4; ++slot) {
                const int j = 16 * min(wave + 4 * slot, ntile - 1) + l15;
                const double* xp = ((j < f.nrest) ? A + (size_t) (c + 32 + j) * lda : Vw + (size_t) (j - f.nrest) * ldv) + c;
#pragma unroll
                for (int ks = 0; ks < 8; ++ks) xt[slot][ks] = xp[4 * ks + l4];
            }
            if (f.nrest > 0 && wave < 2) {                       // tiles 0, 1 = the next leaf's columns: their rows c .. c + 31 -> XT
#pragma unroll
                for (int ks = 0; ks < 8; ++ks) pf_st(f.XT + (16 * wave + l15) * 32 + 4 * ks + l4, xt[0][ks]);
            }
        }
        PF_STAMP(5);
        pf_wait(flags, 0, ef + 2, &L.gflags[4]);
        PF_STAMP(6);
        const bool fb = pf_ld(f.F2 + 5 * 1024) != 0.0;         // the factor workgroup refused the leaf (workgroup-, launch-uniform)
        if (fb) {
            pf_householder_leaf<RT>(ar, P, f, L, flags, er + 2u, g, nrow, r4, r4c, act, toprow, l15, l4, wave);
            pf_image_write<RT>(L.img, ar, wave, l15, l4);       // the product again, from V (Q never existed)
            __syncthreads();
            product();
        }
      if (!fb) {
        // ---- the factors: U'^-1 and T for everyone (they alias the Gram partials, published long ago); the owner of the top block
        // also takes L1 \ U', R = S R2 R1 and S R2
        if (own) { pf_m33 m3[3] = {L.Uinv, L.Bs, L.R2s}; pf_m33_in<3>(m3, f.F2); }
        else { pf_m33 m1[1] = {L.Uinv}; pf_m33_in<1>(m1, f.F2); }
        __syncthreads();
        // ---- V = Q U'^-1; the top block's rows become L1
        pf_rows_times_upper<RT>(ar, L.Uinv, l15, l4);
        if (toprow) {
#pragma unroll
            for (int t = 0; t < RT; ++t)
#pragma unroll
                for (int ks = 0; ks < 8; ++ks) {
                    const int rr_ = r4 + t - c, col = 4 * ks + l4;
                    ar[t][ks] = (col < rr_) ? L.Bs[rr_][col] : (col == rr_ ? 1.0 : 0.0);
                }
        }
        if (own && ntile > 0) {
            // Q = V U' + [B; 0] with B = S R2, so V^T x = U'^-T (Q^T x - B^T x_top): the correction -B^T X_top goes out as one more
            // partial of Z (slot nrow)
            double* X3c = f.X3 + (size_t) nrow * PF_ZSLAB;
            double ba[2][8];
#pragma unroll
            for (int ti = 0; ti < 2; ++ti)
#pragma unroll
                for (int ks = 0; ks < 8; ++ks) ba[ti][ks] = -L.R2s[4 * ks + l4][16 * ti + l15];
#pragma unroll
            for (int slot = 0; slot < 4; ++slot) {
                const int jt = wave + 4 * slot;
                if (jt < ntile) {
                    v4d acc0 = (v4d){0.0, 0.0, 0.0, 0.0}, acc1 = acc0;
#pragma unroll
                    for (int ks = 0; ks < 8; ++ks) {
                        acc0 = pf_mfma(ba[0][ks], xt[slot][ks], acc0);
                        acc1 = pf_mfma(ba[1][ks], xt[slot][ks], acc1);
                    }
                    double* zp = X3c + jt * 512 + 2 * lane;
                    pf_st2(zp, acc0[0], acc0[1]); pf_st2(zp + 128, acc0[2], acc0[3]);
                    pf_st2(zp + 256, acc1[0], acc1[1]); pf_st2(zp + 384, acc1[2], acc1[3]);
                }
            }
        }
      }
        const unsigned xe = fb ? 32u : 0u;                      // the Householder route's 32 column exchanges
        pf_publish(flags, 16 * g, er + 3 + xe);
        PF_STAMP(7);
        // V goes to memory behind the publish (nobody else reads these rows; the publish's drain would wait for the 128 KB)
        if (!fb && act && !toprow) {
            pf_store_rows<RT>(ar, Vw, ldv, c, r4, l4);
            pf_store_rows<RT>(ar, A, lda, c, r4, l4);
        }
        if (!fb) {
            // the second hand-off of the factor workgroup (T; the owner also takes R): it left ~6 us behind the first, and the
            // correction, the publish and the stores above took as long
            pf_wait(flags, 0, ef + 3, &L.gflags[4]);
            if (own) { pf_m33 m2[2] = {L.Ts, L.R1s}; pf_m33_in<2>(m2, f.F2 + 3 * 1024); }
            else { pf_m33 m1[1] = {L.Ts}; pf_m33_in<1>(m1, f.F2 + 3 * 1024); }
            __syncthreads();
            PF_STAMP(12);
        }
        if (own && !fb) {
            // the top block: R above the diagonal of A, L1 below it and (unit lower) in Vw; T and tau -- behind the publish: nobody
            // else waits for these.  Written by the workgroup that owns these rows: it reads them back later (x_top of the
            // corrections, V of the deferred update) as its own stores
#pragma unroll
            for (int q = 0; q < 1024 / PF_THREADS; ++q) {
                const int el = tid + q * PF_THREADS, i = el & 31, cc = el >> 5;
                const double tv = L.Ts[i][cc];
                P.T[(size_t) (c + cc) * P.ldt + c + i] = tv;
                if (i == cc) P.tau[c + i] = tv;
                A[(size_t) (c + cc) * lda + c + i] = (cc >= i) ? L.R1s[i][cc] : L.Bs[i][cc];
                Vw[(size_t) (c + cc) * ldv + c + i] = (cc < i) ? L.Bs[i][cc] : (cc == i ? 1.0 : 0.0);
            }
        }
        pf_wait(flags, nrow, er + 3 + xe, &L.gflags[4]);
        PF_STAMP(8);
        pf_fold(P, f, L, g, nrow, wave, lane, fb);
        pf_publish(flags, 16 * g, er + 4 + xe);
        PF_STAMP(9);
        v4d uca[RT], ucb[RT];
        if (f.nrest > 0) pf_update0_load<RT>(uca, ucb, A, lda, c, r4c, l4);
        pf_wait(flags, nrow, er + 4 + xe, &L.gflags[4]);
        PF_STAMP(10);
        // the next leaf's 32 columns are updated now (the result stays in registers as its a); the other columns of A_rest wait for
        // the next pass's Cholesky window (above)
        if (f.nrest > 0) pf_update0_finish<RT>(ar, uca, ucb, f.X4, A, lda, c, r4, act, l15, l4);
        PF_STAMP(11);
        er += 4u + (fb ? 32u : 0u);
        ef += 3u;
        g1_derived = !fb && f.nrest > 0;
    }
    if (g == 0 && tid == 0 && L.gflags[4]) P.status[1] = 1;
}

// ---------------------------------------------------------------------------------------------------------------------------------
// The FACTOR workgroup (the launch's last workgroup; four waves on a compute unit of its own): sums the partial Gram matrices,
// runs the one-wave recurrences with nothing beside them on their SIMDs, and publishes the 32 x 32 factors for the row workgroups.
//   wave 0: Cholesky of G1 with R1^-1 on its upper lanes; modified LU with L1^-1 on its upper lanes; U = U' R2^-1
//   wave 1, 2: the two 16 x 16 diagonal blocks of U'^-1 (wave 1 then the off-diagonal block on the matrix cores)
//   waves 0 - 2: one 16 x 16 tile each of T = -U S L1^-T and R = S R2 R1 on the matrix cores
// ---------------------------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ void pf_factor_wg(const PfArgs& P, double* sm, int nrow)
{
    const PfLds L = pf_lds(sm);
    const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int g = 0;                                      // (stamps: this workgroup reports under its own stamp numbers)
    unsigned* const flags = reinterpret_cast<unsigned*>(P.ws);
    int nfallback = 0;
    unsigned er = P.epoch0, ef = P.epoch0;
    bool g1_derived = false;
    if (tid == 0) { L.gflags[4] = 0; L.gflags[5] = 0; L.gflags[6] = 0; L.gflags[8] = 0; L.gflags[9] = 0; L.gflags[10] = 0; L.gflags[11] = 0; }
    for (int c = 0; c < P.wh; c += 32) {
        const PfLeaf f = pf_leaf<4>(P, c);                  // (only gown depends on the row split, and this workgroup does not use it)
        // the lane index is made opaque once per leaf: otherwise every lane-dependent constant of the unrolled recurrences below
        // (identity columns, `lane == K` selects, ...) is hoisted out of this loop and kept in registers across it -- ~400 of them
        int lane = tid & 63;
        asm volatile("" : "+v"(lane));
        const int rc = lane & 31;
        const int seq = (c >> 5) + 1;
        if (tid == 0) { L.gflags[0] = 1; L.gflags[1] = 0; L.gflags[2] = 0; L.gflags[3] = 0; }
        if (!g1_derived) {
            pf_wait(flags, nrow, er + 1, &L.gflags[4]);
            PF_STAMP_S(16);
            pf_gram_sum(f.X1, nrow, L.Gs, L.gflags, false);
            __syncthreads();
        }
        PF_STAMP_S(17);
        // R1 = chol(G1) and R1^-1 on one wave (the identity columns ride on the wave's upper half)
        if (wave == 0) {
            // (round 5: on the matrix cores, micro-blocked by 4 -- qr_factor32.h; the register recurrence CholAugStep took 8 us here)
            int e2 = 0;                                        // even power-of-two scaling, as in cholq3_kernel
            {
                const double d = L.Gs[0][0];
                if (d > 0.0 && d < 1.7e308) { (void) frexp(d, &e2); e2 &= ~1; }
            }
            const double rs = ldexp(1.0, e2 / 2), wsc = ldexp(1.0, -(e2 / 2));
            const bool ok = chol32_mfma(lane, [&](int i, int j) { return (L.Gs[j][i] * wsc) * wsc; },
                                        [&](int i, int j, double v) { L.R1s[i][j] = v * rs; },                     // R1(i, j)
                                        [&](int i, int j, double v) { L.Ws[j][i] = v * wsc; });                    // R1^-1(j, i) = R1^-T(i, j)
            if (lane == 0 && !ok) L.gflags[0] = 0;
        }
        __syncthreads();
        PF_STAMP_S(18);
        pf_m33_out(L.Ws, f.F1);
        if (tid == 0) pf_st(f.F1 + 1024, (double) L.gflags[0]);
        pf_publish(flags, PF_FAC_WORD, ef + 1);
        PF_STAMP_S(19);
        pf_wait(flags, nrow, er + 2, &L.gflags[4]);
        PF_STAMP_S(20);
        // Q_top (published with the G2 partials) is requested NOW and parked in LDS while the partials are summed: the modified LU then
        // takes its 32 operand values per lane from LDS instead of opening with a round trip to memory (9.5 us in situ against 7.6 alone)
        pf_m33 Qts = reinterpret_cast<pf_m33>(L.img + PF_M33);          // (At's place: free until this leaf's G1 derivation, far behind the LU)
        double qtv[1024 / PF_THREADS];
#pragma unroll
        for (int q = 0; q < 1024 / PF_THREADS; ++q) qtv[q] = pf_ld(f.QT + tid + q * PF_THREADS);
        pf_gram_sum(f.X2, nrow, L.Gs, L.gflags, true);
#pragma unroll
        for (int q = 0; q < 1024 / PF_THREADS; ++q) {
            const int e = tid + q * PF_THREADS;
            Qts[e >> 5][e & 31] = qtv[q];                               // Qts[j][i] = Q_top(i, j)
        }
        __syncthreads();
        PF_STAMP_S(21);
        // R2 = chol(G2) (to first order when G2 - I is tiny) and the modified LU  Q_top - S R2 = L1 U' on wave 0.  Its upper 32 lanes
        // carry the columns of the identity through the same row operations: they end as L1^-1, for nothing.  Then:
        //   wave 0: U = U' R2^-1 -- with the first-order R2, R2^-1 = 2 I - R2 to ~1e-17 and U = 2 U' - U' R2 is three tiles on the
        //           matrix cores; the general R2 (leaves of condition > ~1e4) takes the row solve
        //   wave 2: inverse of the lower diagonal block of U'          wave 1: inverse of the upper one, then the off-diagonal block
        //           -X11 U'12 X22 on the matrix cores  (two 16-step recurrences + 8 MFMAs instead of one 32-step recurrence)
        if (wave == 0) {
            const bool refused = L.gflags[0] == 0 || L.gflags[1] != 0;
            const bool first_order = L.gflags[2] == 0;
            bool ok = true;
            {
                // R2 -> R2s (column `lane` per lane): to first order when G2 = I + E with |E| <= 1e-9 (R2 = I + striu(E) + diag(E) / 2 up to
                // terms of size 32 |E|^2), else the second Cholesky as a register recurrence (leaves of condition > ~1e4)
                double gg[PW];
                double dinv = 1.0;
#pragma unroll
                for (int i = 0; i < PW; ++i) gg[i] = L.Gs[rc][i];
                if (!first_order) {
                    Chol3Step<0>::run(gg, lane, ok, dinv);
                } else {
#pragma unroll
                    for (int i = 0; i < PW; ++i) gg[i] = (i < rc) ? gg[i] : (i == rc ? 1.0 + 0.5 * (gg[i] - 1.0) : 0.0);
                }
                if (lane < PW) {
#pragma unroll
                    for (int k = 0; k < PW; ++k) L.R2s[k][lane] = gg[k];
                    L.r2inv[lane] = dinv;                      // (read by the row solve of the general case only)
                }
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            }
            if (ok && !refused) {
                // modified LU  Q_top - S R2 = L1 U'  with L1^-1 and U'^-1, all on the matrix cores (qr_factor32.h; round 4: a register
                // recurrence here, 10 us, and two 16-step recurrences + 8 MFMAs on waves 1 and 2 behind it for U'^-1, 3.5 us).
                // Ls aliases Gs: G2 is not needed any more
                lu32_mfma(lane, [&](int i, int j) { return Qts[j][i]; }, [&](int i, int j) { return L.R2s[i][j]; },
                          [&](int i, int j, double v) { L.Bs[i][j] = v; }, [&](int i, double v) { L.Ss[i] = v; },
                          [&](int i, int j, double v) { L.Ls[i][j] = v; }, [&](int i, int j, double v) { L.Uinv[j][i] = v; });
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            } else if (lane == 0) L.gflags[3] = 1;
            PF_STAMP_S(22);
            pf_lds_signal(&L.gflags[5], seq);
            __builtin_amdgcn_wave_barrier();
            // U -> Us (aliases R1^-1, which is published)
            if (first_order) {
                const int l15 = lane & 15, l4 = lane >> 4;
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
#pragma unroll
                for (int tile = 0; tile < 3; ++tile) {
                    const int ti = (tile == 2) ? 1 : 0, tc = (tile == 0) ? 0 : 1;
                    v4d acc = (v4d){0.0, 0.0, 0.0, 0.0};
#pragma unroll
                    for (int ks = 0; ks < 8; ++ks) {
                        const int k = 4 * ks + l4, i = 16 * ti + l15;
                        const double up = (k >= i) ? L.Bs[i][k] : 0.0;
                        acc = pf_mfma(-up, L.R2s[k][16 * tc + l15], acc);
                    }
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int i = 16 * ti + l4 + 4 * r, cc = 16 * tc + l15;
                        L.Us[i][cc] = (cc >= i) ? 2.0 * L.Bs[i][cc] + acc[r] : 0.0;
                    }
                }
                if (lane < 16)
#pragma unroll
                    for (int r = 0; r < 16; ++r) L.Us[16 + r][lane] = 0.0;
            } else {
                double u[PW];
#pragma unroll
                for (int cc = 0; cc < PW; ++cc) u[cc] = (cc >= rc) ? L.Bs[rc][cc] : 0.0;
                RowSolve<0>::run(u, L.R2s, L.r2inv);
                if (lane < PW) {
#pragma unroll
                    for (int cc = 0; cc < PW; ++cc) L.Us[lane][cc] = (cc >= lane) ? u[cc] : 0.0;
                }
            }
        } else {
            // Round 6: the FIRST hand-off of the leaf's factors -- U'^-1 (V = q U'^-1), L1 \ U' (the top block's rows of V) and S R2 (the
            // correction of the product), with the leaf's verdict -- goes out from waves 1-3 as soon as the LU is done, while wave 0
            // goes on to U, T and R: the row workgroups used to wait for those too (4.8 us per leaf), and need T only at the fold, ~6 us
            // later.  No workgroup barrier here (wave 0 is busy): each wave drains its own stores and raises an LDS word, wave 3 collects
            // the three and stores the epoch word.
            pf_lds_await(&L.gflags[5], seq);
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
            const int t3 = tid - 64;                             // 0 .. 191
            for (int e = t3; e < 1024; e += 192) {
                const int i = e >> 5, cc = e & 31;
                pf_st(f.F2 + e, L.Uinv[i][cc]);
                pf_st(f.F2 + 1024 + e, L.Bs[i][cc]);
                pf_st(f.F2 + 2 * 1024 + e, L.Ss[i] * L.R2s[i][cc]);      // S R2 replaces R2 for the row workgroups' correction
            }
            if (t3 == 0) pf_st(f.F2 + 5 * 1024, (L.gflags[0] == 0 || L.gflags[1] != 0 || L.gflags[3] != 0) ? 1.0 : 0.0);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            pf_lds_signal(&L.gflags[8 + wave], seq);
            if (wave == 3) {
                pf_lds_await(&L.gflags[9], seq);
                pf_lds_await(&L.gflags[10], seq);
                if (lane == 0) __hip_atomic_store(flags + PF_FAC_WORD, ef + 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#ifdef PF_STAMPS
                if (lane == 0 && P.stamps) P.stamps[(c >> 5) * 32 + 25] = (long long) __builtin_amdgcn_s_memrealtime();
#endif
            }
        }
        __syncthreads();
        PF_STAMP_S(23);
        // T = -U S L1^-T and R = S R2 R1 on the matrix cores, one 16 x 16 tile of each per wave (tiles (0,0), (0,1), (1,1))
        if (wave <= 2) {
            const int l15 = lane & 15, l4 = lane >> 4;
            const int ti = (wave == 2) ? 1 : 0, tc = (wave == 0) ? 0 : 1;
            v4d tt = (v4d){0.0, 0.0, 0.0, 0.0}, rt = tt;
            const double si = L.Ss[16 * ti + l15];
#pragma unroll
            for (int ks = 0; ks < 8; ++ks) {
                const int k = 4 * ks + l4;
                tt = pf_mfma(-L.Us[16 * ti + l15][k] * L.Ss[k], L.Ls[16 * tc + l15][k], tt);
                rt = pf_mfma(si * L.R2s[16 * ti + l15][k], L.R1s[k][16 * tc + l15], rt);
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int i = 16 * ti + l4 + 4 * r, cc = 16 * tc + l15;
                L.Ts[i][cc] = (cc >= i) ? tt[r] : 0.0;
                L.Rm[i][cc] = (cc >= i) ? rt[r] : 0.0;
            }
            if (wave == 0) {                                 // tile (1, 0): zeros
#pragma unroll
                for (int r = 0; r < 4; ++r) { L.Ts[16 + l4 + 4 * r][l15] = 0.0; L.Rm[16 + l4 + 4 * r][l15] = 0.0; }
            }
        }
        __syncthreads();
        // the SECOND hand-off: T (the fold's W = T^T y; the owner's T block and tau) and R (the owner's top block)
        const bool fb = L.gflags[0] == 0 || L.gflags[1] != 0 || L.gflags[3] != 0;
        if (!fb) {                                               // (a refused leaf: its T comes from the Householder route of the row workgroups)
            pf_m33_out(L.Ts, f.F2 + 3 * 1024);
            pf_m33_out(L.Rm, f.F2 + 4 * 1024);
        }
        if (fb) ++nfallback;
        pf_publish(flags, PF_FAC_WORD, ef + 3);
        PF_STAMP_S(24);
        // ---- the next leaf's G1 without an exchange: G1 = G' - R12^T R12 with G' = A_next^T A_next before this leaf's update (summed
        // slice-wise by the row workgroups with their fold) and R12 = A_next(top 32 rows) - L1 W(:, next 32 columns), the rows this
        // leaf's update leaves above the next leaf: the update is an orthogonal transformation of rows >= c, so the Gram matrix of the
        // rows >= c + 32 is what is left.  (An error of G1 only costs the first CholeskyQR pass orthogonality, which the second pass
        // measures from the real Q; cancellation -- a next leaf nearly inside this leaf's span -- ends at the guard like any other
        // badly conditioned leaf.)  Not behind a Householder-route leaf: its L1 lives in the row workgroups.
        g1_derived = !fb && f.nrest > 0;
        if (g1_derived) {
            pf_wait(flags, nrow, er + 4, &L.gflags[4]);          // all W slices and G' slices are published
            pf_m33 At = reinterpret_cast<pf_m33>(L.img + PF_M33), Wn = reinterpret_cast<pf_m33>(L.img + 2 * PF_M33),
                   R12 = reinterpret_cast<pf_m33>(L.img + 3 * PF_M33);
            {
                double ge[3], at[4], wn[4];
#pragma unroll
                for (int q = 0; q < 3; ++q) ge[q] = pf_ld(f.XG + tid + 256 * q);
#pragma unroll
                for (int q = 0; q < 4; ++q) { at[q] = pf_ld(f.XT + tid + 256 * q); wn[q] = pf_ld(f.X4 + tid + 256 * q); }
#pragma unroll
                for (int q = 0; q < 3; ++q) pf_gram_entry(tid + 256 * q, ge[q], L.Gs, L.gflags, false);
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int e = tid + 256 * q, col = e >> 5, row = e & 31;
                    At[row][col] = at[q];                      // XT[col * 32 + row]
                    Wn[row][col] = -wn[q];                     // X4[j * 32 + i] = -W(i, j)
                }
            }
            __syncthreads();
            {
                const int l15 = lane & 15, l4 = lane >> 4, ti = wave & 1, tj = wave >> 1;
                v4d acc;
#pragma unroll
                for (int r = 0; r < 4; ++r) acc[r] = At[16 * ti + l4 + 4 * r][16 * tj + l15];
#pragma unroll
                for (int ks = 0; ks < 8; ++ks) {
                    const int k = 4 * ks + l4, i = 16 * ti + l15;
                    const double l1 = (k < i) ? L.Bs[i][k] : (k == i ? 1.0 : 0.0);
                    acc = pf_mfma(-l1, Wn[k][16 * tj + l15], acc);
                }
#pragma unroll
                for (int r = 0; r < 4; ++r) R12[16 * ti + l4 + 4 * r][16 * tj + l15] = acc[r];
            }
            __syncthreads();
            if (wave < 3) {
                const int l15 = lane & 15, l4 = lane >> 4, ti = (wave == 2) ? 1 : 0, tc = (wave == 0) ? 0 : 1;
                v4d acc;
#pragma unroll
                for (int r = 0; r < 4; ++r) acc[r] = L.Gs[16 * tc + l15][16 * ti + l4 + 4 * r];      // Gs[j][i] = G(i, j)
#pragma unroll
                for (int ks = 0; ks < 8; ++ks) {
                    const int k = 4 * ks + l4;
                    acc = pf_mfma(-R12[k][16 * ti + l15], R12[k][16 * tc + l15], acc);
                }
                __builtin_amdgcn_wave_barrier();
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int i = 16 * ti + l4 + 4 * r, j = 16 * tc + l15;
                    L.Gs[j][i] = acc[r];
                    if (wave == 1) L.Gs[i][j] = acc[r];
                }
            }
            __syncthreads();
        }
        if (P.merge64 && c == 32) {
            // The panel's T in this launch (64-column panels: BASELINE's C2 block size): T(0:32, 32:64) = -T_0 (V_0^T V_1) T_1.  The Gram block
            // comes from the row workgroups' fold of this leaf (write-through stores, complete with their W slices), the leaves' T blocks
            // from the two hand-off slabs (leaf parity 0 / 1) -- the factor workgroup's own, or the owner's after a Householder-route leaf
            pf_wait(flags, nrow, er + 4u + (fb ? 32u : 0u), &L.gflags[4]);
            pf_m33 T0 = reinterpret_cast<pf_m33>(L.img + PF_M33), T1 = reinterpret_cast<pf_m33>(L.img + 2 * PF_M33),
                   Gm = reinterpret_cast<pf_m33>(L.img + 3 * PF_M33), Xm = L.Rm;
            {
                double t0[4], t1[4], gv[4];
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int e = tid + 256 * q;
                    t0[q] = pf_ld(P.ws + PF_OFF_F2 + 3 * 1024 + e);
                    t1[q] = pf_ld(P.ws + PF_OFF_F2 + PF_F2_SZ + 3 * 1024 + e);
                    gv[q] = pf_ld(P.G + (size_t) (32 + (e >> 5)) * P.ldg + (e & 31));      // G(j' = e & 31, 32 + i), i = e >> 5
                }
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int e = tid + 256 * q;
                    T0[e >> 5][e & 31] = t0[q];
                    T1[e >> 5][e & 31] = t1[q];
                    Gm[e & 31][e >> 5] = gv[q];
                }
            }
            __syncthreads();
            const int l15 = lane & 15, l4 = lane >> 4, ti = wave & 1, tj = wave >> 1;
            {
                v4d acc = (v4d){0.0, 0.0, 0.0, 0.0};
#pragma unroll
                for (int ks = 0; ks < 8; ++ks) {
                    const int k = 4 * ks + l4;
                    acc = pf_mfma(Gm[16 * ti + l15][k], T1[k][16 * tj + l15], acc);
                }
#pragma unroll
                for (int r = 0; r < 4; ++r) Xm[16 * ti + l4 + 4 * r][16 * tj + l15] = acc[r];
            }
            __syncthreads();
            {
                v4d acc = (v4d){0.0, 0.0, 0.0, 0.0};
#pragma unroll
                for (int ks = 0; ks < 8; ++ks) {
                    const int k = 4 * ks + l4;
                    acc = pf_mfma(-T0[16 * ti + l15][k], Xm[k][16 * tj + l15], acc);
                }
#pragma unroll
                for (int r = 0; r < 4; ++r) P.T[(size_t) (32 + 16 * tj + l15) * P.ldt + 16 * ti + l4 + 4 * r] = acc[r];
            }
        }
        er += 4u + (fb ? 32u : 0u);                          // the Householder route's 32 column exchanges among the row workgroups
        ef += 3u;
    }
    if (tid == 0) {
        if (nfallback) atomicAdd(P.status, nfallback);
        if (L.gflags[4]) P.status[1] = 1;
    }
}

template <int RT>
__global__ __launch_bounds__(PF_THREADS) void panel_fused_kernel(PfArgs P)
{
    extern __shared__ __attribute__((aligned(16))) double sm[];
    const int nrow = (int) gridDim.x - 1;
    if ((int) blockIdx.x < nrow) pf_row_wg<RT>(P, sm, blockIdx.x, nrow);
    else pf_factor_wg(P, sm, nrow);
}

static long long* g_pf_stamps = nullptr;      // development only (qrd_panel_fused_set_stamps)

extern "C" {

void qrd_panel_fused_set_stamps(long long* d) { g_pf_stamps = d; }

size_t qrd_panel_fused_ws_doubles(void) { return (size_t) PF_WS_DOUBLES; }

int qrd_panel_fused_init(void)
{
    int rc = (int) hipFuncSetAttribute(reinterpret_cast<const void*>(panel_fused_kernel<4>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                       (int) (PF_SM_DOUBLES(4) * sizeof(double)));
    rc |= (int) hipFuncSetAttribute(reinterpret_cast<const void*>(panel_fused_kernel<2>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                    (int) (PF_SM_DOUBLES(2) * sizeof(double)));
    return rc;
}

// Rows per row workgroup for a panel of mk rows x wh columns on this stream: 128 or 256; 0: the one-launch panel cannot take it (more
// than 16384 rows, or more row workgroups than the stream has compute units beside the factor workgroup's: all must be co-resident).
// 128 halves a workgroup's streaming work (deferred update + in-panel product: matrix-core time on its one compute unit) and doubles
// the partials the hand-offs sum; measured, us per panel, 256 -> 128 rows (profiles/r06_panel_fused_perf.txt): 2048 x 256 477 -> 359,
// 4096 x 256 443 -> 359, 8192 x 256 440 -> 405, 2048 x 128 181 -> 156, 4096 x 128 175 -> 167, 8192 x 128 185 -> 194, 4096 x 64 86 -> 85,
// 2048 x 64 86 -> 77.  So: 128 up to 32 row workgroups (4096 rows), and beyond that for panels wider than 128 columns, where the
// streaming dominates.  want: 0 = this rule, 128 / 256 = that form if it fits (unit tests; MI355XQR_PF_ROWS in the lab build).
static int pf_rows_for(void* stream, int mk, int wh, int want = 0)
{
    static const int env = QRD_LAB_ENV_INT("MI355XQR_PF_ROWS", 0);
    const int forced = want ? want : env;
    int cus = qrd_stream_cus_coresident(stream) - 1;
    if (cus > PF_MAXWG) cus = PF_MAXWG;
    // up to PF_MAXWG row workgroups of 256 rows: 16384 rows where the stream has 65 compute units (rounds 4-6: 8192, the 32 x 256 rows of round
    // 4's workspace -- the limit outlived it by a round; 16384 x 256 is 473 us in one launch where the leaf chain took 1.0-1.4 ms:
    // profiles/r06_panel_fused_16384_rows.txt).  MI355XQR_PF_MAX_ROWS (lab build): the old limit for A/B runs
    static const int max_rows = QRD_LAB_ENV_INT("MI355XQR_PF_MAX_ROWS", PF_MAXWG * 256);
    if (mk > max_rows || mk > PF_MAXWG * 256) return 0;
    const int n128 = (mk + 127) / 128, n256 = (mk + 255) / 256;
    const bool fit128 = n128 <= cus, fit256 = n256 <= cus;
    if (forced == 128 && fit128) return 128;
    if (forced == 256 && fit256) return 256;
    if (fit128 && (n128 <= 32 || wh > 128 || !fit256)) return 128;
    return fit256 ? 256 : 0;
}

// 1 when a launch with these arguments leaves the panel's COMPLETE T behind (the leaves' blocks and the merge): 64-column panels whose Gram
// block is taken along.  The caller then skips its merge tree (qrd_larft).
int qrd_panel_fused_merges_t(int wh, int with_gram) { return wh == 64 && with_gram; }

// 1 when the one-launch panel can take this (half-)panel: whole 32-column leaves, at most 256 columns, at most 16384 rows and a free
// compute unit per row workgroup on the stream, vector-aligned operands
int qrd_panel_fused_ok(void* stream, const double* A, int lda, int mk, int wh, const double* Vw, int ldv)
{
    if (wh < 32 || wh > 256 || wh % 32 || mk < wh || mk % 4) return 0;
    if ((reinterpret_cast<uintptr_t>(A) & 15) || (reinterpret_cast<uintptr_t>(Vw) & 15) || lda % 2 || ldv % 2) return 0;
    return pf_rows_for(stream, mk, wh) != 0;
}

// *epoch: the caller's epoch counter for this workspace (starts at 0 with a zeroed workspace); advanced by the launch.
// rows: 0 = the library's choice (pf_rows_for); 256 = the 256-row workgroups whatever the stream offers (kernel unit tests)
int qrd_panel_fused_rows(void* stream, double* A, int lda, int mk, int wh, double* tau, double* T, int ldt, double* Vw, int ldv,
                         double* G, int ldg, double* ws, unsigned* epoch, int* status, int want_rows)
{
    if (want_rows != 0 && want_rows != 128 && want_rows != 256) return -7;
    if (!qrd_panel_fused_ok(stream, A, lda, mk, wh, Vw, ldv) || !ws || !epoch || !status) return -7;
    PfArgs a;
    a.A = A; a.lda = lda; a.mk = mk; a.wh = wh; a.Vw = Vw; a.ldv = ldv; a.T = T; a.ldt = ldt; a.tau = tau; a.G = G; a.ldg = ldg;
    a.ws = ws; a.epoch0 = *epoch; a.status = status; a.stamps = g_pf_stamps;
    a.merge64 = qrd_panel_fused_merges_t(wh, G != nullptr);
    *epoch += 1024u;
    const int rows = pf_rows_for(stream, mk, wh, want_rows), nrow = (mk + rows - 1) / rows;
    if (rows == 128)
        hipLaunchKernelGGL(panel_fused_kernel<2>, dim3(nrow + 1), dim3(PF_THREADS), PF_SM_DOUBLES(2) * sizeof(double), (hipStream_t) stream, a);
    else
        hipLaunchKernelGGL(panel_fused_kernel<4>, dim3(nrow + 1), dim3(PF_THREADS), PF_SM_DOUBLES(4) * sizeof(double), (hipStream_t) stream, a);
    return (int) hipGetLastError();
}

int qrd_panel_fused(void* stream, double* A, int lda, int mk, int wh, double* tau, double* T, int ldt, double* Vw, int ldv,
                    double* G, int ldg, double* ws, unsigned* epoch, int* status)
{
    return qrd_panel_fused_rows(stream, A, lda, mk, wh, tau, T, ldt, Vw, ldv, G, ldg, ws, epoch, status, 0);
}

}   // extern "C"
