// qr_panel_fused.hip -- a whole outer panel (up to 256 columns = 8 leaves of 32) in ONE launch.
//
// Replaces, for panels of up to 8192 rows, the per-leaf launch sequence of qr_panel_tsqr.hip (gram32 -> cholq3 -> hr3_ep -> final4 ->
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

#define PF_THREADS 448            /* 4 row waves + 3 service waves */
#define PF_ROWS 256               /* rows per workgroup */
#define PF_LDQ 260                /* row stride of the [column][row] image (doubles) */
#define PF_MAXWG 32
#define PF_ZCOLS 224              /* columns of Z: wh - 32 <= 224 */
#define PF_SPIN_LIMIT (1u << 22)

// workspace layout (doubles); the first 2 KB are the epoch words, one per workgroup, 64 bytes apart
#define PF_OFF_X1 256
#define PF_OFF_X2 (PF_OFF_X1 + 2 * PF_MAXWG * 1024)
#define PF_OFF_QT (PF_OFF_X2 + 2 * PF_MAXWG * 1024)
#define PF_OFF_X3 (PF_OFF_QT + 2 * 1024)
#define PF_OFF_X4 (PF_OFF_X3 + 2 * (PF_MAXWG + 1) * 32 * PF_ZCOLS)
#define PF_OFF_XF (PF_OFF_X4 + 2 * 32 * PF_ZCOLS)
#define PF_WS_DOUBLES (PF_OFF_XF + 2 * PF_MAXWG * 128)

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
#define PF_SM_SCR (PF_SM_SS + 64)                /* 7 waves x 128 */
#define PF_SM_FLAGS (PF_SM_SCR + 7 * 128)        /* ints */
#define PF_SM_PART (PF_SM_FLAGS + 8)             /* (16 ints) */             /* 4 x 768 per-wave Gram partials; later U'^-1 and T */
#define PF_SM_IMG (PF_SM_PART + 3072)            /* [column][row] image of the workgroup's rows; later -W */
#define PF_SM_DOUBLES (PF_SM_IMG + 32 * PF_LDQ)

struct PfArgs {
    double* A; int lda;          // panel origin: mk rows x wh columns, factored in place
    int mk, wh;
    double* Vw; int ldv;         // explicit V, same origin
    double* T; int ldt;          // wh x wh: the leaves' T blocks go on the diagonal
    double* tau;
    double* G; int ldg;          // Gram blocks for the T merge: G(j', c + i) = V(:, j')^T V(:, c + i), j' < c
    double* ws;                  // PF_WS_DOUBLES
    unsigned epoch0;             // epoch words hold values <= epoch0 when the launch starts
    int* status;                 // [0] += leaves that took the Householder route; [1] = 1 when a wait timed out
    long long* stamps;           // development builds (-DPF_STAMPS): 32 phase stamps per leaf of workgroup 0 (100 MHz clock)
};

typedef double (*pf_m33)[33];

#ifdef PF_STAMPS
#define PF_STAMP(k) do { if (g == 0 && threadIdx.x == 0 && P.stamps) P.stamps[(c >> 5) * 32 + (k)] = (long long) __builtin_amdgcn_s_memrealtime(); } while (0)
#define PF_STAMP_S(k) do { if (g == 0 && threadIdx.x == 256 && P.stamps) P.stamps[(c >> 5) * 32 + (k)] = (long long) __builtin_amdgcn_s_memrealtime(); } while (0)
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
__device__ __forceinline__ void pf_publish(unsigned* flags, int g, unsigned val)
{
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) __hip_atomic_store(flags + 16 * g, val, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// every thread: returns when all nwg epoch words have reached val (lane w of wave 0 polls word w); a wait that does not complete
// in ~1 s marks the launch dead -- every later wait returns at once, the launch ends with garbage and status[1] = 1 instead of hanging
__device__ __forceinline__ void pf_wait(const unsigned* flags, int nwg, unsigned val, int* dead)
{
    if (threadIdx.x < 64) {
        const int lane = threadIdx.x;
        if (*dead == 0) {
            unsigned spins = 0;
            for (;;) {
                const unsigned f = (lane < nwg) ? __hip_atomic_load(flags + 16 * lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : val;
                if (__all((int) (f - val) >= 0)) break;
                __builtin_amdgcn_s_sleep(2);
                if (++spins > PF_SPIN_LIMIT) { if (lane == 0) *dead = 1; break; }
            }
        }
    }
    __syncthreads();
}

// this lane's four rows x eight columns of a leaf (L_row layout): a[t][ks] = A(r4 + t, col0 + 4 ks + l4)
__device__ __forceinline__ void pf_load_rows(double (&a)[4][8], const double* __restrict__ A, int lda, int col0, int r4c, int l4)
{
#pragma unroll
    for (int ks = 0; ks < 8; ++ks) {
        const double* p = A + (size_t) (col0 + 4 * ks + l4) * lda + r4c;
        const v2d lo = *reinterpret_cast<const v2d*>(p), hi = *reinterpret_cast<const v2d*>(p + 2);
        a[0][ks] = lo[0]; a[1][ks] = lo[1]; a[2][ks] = hi[0]; a[3][ks] = hi[1];
    }
}

__device__ __forceinline__ void pf_store_rows(const double (&a)[4][8], double* __restrict__ A, int lda, int col0, int r4, int l4)
{
#pragma unroll
    for (int ks = 0; ks < 8; ++ks) {
        double* p = A + (size_t) (col0 + 4 * ks + l4) * lda + r4;
        *reinterpret_cast<v2d*>(p) = (v2d){a[0][ks], a[1][ks]};
        *reinterpret_cast<v2d*>(p + 2) = (v2d){a[2][ks], a[3][ks]};
    }
}

// [column][row] image of the workgroup's 256 rows: img[col * PF_LDQ + row - wgrow0]
__device__ __forceinline__ void pf_image_write(double* img, const double (&a)[4][8], int wave, int l15, int l4)
{
#pragma unroll
    for (int ks = 0; ks < 8; ++ks) {
        double* p = img + (4 * ks + l4) * PF_LDQ + wave * 64 + 4 * l15;
        *reinterpret_cast<v2d*>(p) = (v2d){a[0][ks], a[1][ks]};
        *reinterpret_cast<v2d*>(p + 2) = (v2d){a[2][ks], a[3][ks]};
    }
}

__device__ __forceinline__ void pf_image_read(const double* img, double (&a)[4][8], int wave, int l15, int l4)
{
#pragma unroll
    for (int ks = 0; ks < 8; ++ks) {
        const double* p = img + (4 * ks + l4) * PF_LDQ + wave * 64 + 4 * l15;
        const v2d lo = *reinterpret_cast<const v2d*>(p), hi = *reinterpret_cast<const v2d*>(p + 2);
        a[0][ks] = lo[0]; a[1][ks] = lo[1]; a[2][ks] = hi[0]; a[3][ks] = hi[1];
    }
}

// x <- x M for the lane's rows, M (32 x 32, upper triangular) in LDS as Mm[k][c]; transposed product, so the result lands in the
// layout of the input:  D[i = column][j = row] = sum_k M(k, i) x(row, k)
__device__ __forceinline__ void pf_rows_times_upper(double (&a)[4][8], pf_m33 Mm, int l15, int l4)
{
    double aw[2][8];
#pragma unroll
    for (int ti = 0; ti < 2; ++ti)
#pragma unroll
        for (int ks = 0; ks < 8; ++ks) aw[ti][ks] = Mm[4 * ks + l4][16 * ti + l15];
#pragma unroll
    for (int t = 0; t < 4; ++t) {
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

// Gram matrix of this row wave's 64 rows from the image: tiles (0,0), (0,1), (1,1) -> part[wave][(tile * 4 + rr) * 64 + lane]
__device__ __forceinline__ void pf_gram_wave(const double* img, double* part, int wave, int lane, int l15, int l4)
{
    v4d acc[3];
#pragma unroll
    for (int i = 0; i < 3; ++i) acc[i] = (v4d){0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int sg = 0; sg < 4; ++sg) {
        const double* p0 = img + l15 * PF_LDQ + wave * 64 + 16 * sg + 4 * l4;
        const double* p1 = p0 + 16 * PF_LDQ;
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

// all threads: Gs[j][i] = sum over the workgroups (in index order) of their partial G(i, j).  A thread owns entries tid and tid + 448 and
// has the loads of 16 workgroups for both in flight at once: two round trips at 32 workgroups (three rounds of four were 5.7 us)
__device__ __forceinline__ void pf_gram_sum(const double* __restrict__ X, int nwg, pf_m33 Gs, int* gflags, bool check)
{
    const int e0 = threadIdx.x, e1 = threadIdx.x + PF_THREADS;
    const bool h1 = e1 < 768;
    const int e1c = h1 ? e1 : e0;
    double s0 = 0.0, s1 = 0.0;
    for (int w0 = 0; w0 < nwg; w0 += 16) {
        double v0[16], v1[16];
#pragma unroll
        for (int u = 0; u < 16; ++u) {
            const double* xw = X + (size_t) min(w0 + u, nwg - 1) * 1024;
            v0[u] = pf_ld(xw + e0);
            v1[u] = pf_ld(xw + e1c);
        }
#pragma unroll
        for (int u = 0; u < 16; ++u)
            if (w0 + u < nwg) { s0 += v0[u]; s1 += v1[u]; }
    }
    pf_gram_entry(e0, s0, Gs, gflags, check);
    if (h1) pf_gram_entry(e1, s1, Gs, gflags, check);
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

// LDS views shared by the two roles
struct PfLds {
    double* img; double* part;
    pf_m33 Uinv, Ts, Gs, Ls, R1s, Ws, Us, R2s, Bs;
    double *Ss, *r2inv, *scr;
    int* gflags;     // [0] first Cholesky ok, [1] |G2 - I| > 1/64, [2] > 1e-9, [3] second Cholesky failed, [4] dead (a wait timed out),
                     // [5] leaves whose LU (B, L1, U') is in LDS, [6] leaves whose inverse of U'(16:32, 16:32) is in LDS
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
    L.Ss = sm + PF_SM_SS;
    L.r2inv = L.Ss + 32;
    L.scr = sm + PF_SM_SCR;
    L.gflags = reinterpret_cast<int*>(sm + PF_SM_FLAGS);
    return L;
}

struct PfLeaf {                      // per-leaf constants
    int c, nrest, ncols, gown;
    double *X1, *X2, *QT, *X3, *X4;
};

__device__ __forceinline__ PfLeaf pf_leaf(const PfArgs& P, int c)
{
    PfLeaf f;
    const int par = (c >> 5) & 1;
    f.c = c; f.nrest = P.wh - c - 32; f.ncols = P.wh - 32; f.gown = c / PF_ROWS;
    f.X1 = P.ws + PF_OFF_X1 + (size_t) par * PF_MAXWG * 1024;
    f.X2 = P.ws + PF_OFF_X2 + (size_t) par * PF_MAXWG * 1024;
    f.QT = P.ws + PF_OFF_QT + (size_t) par * 1024;
    f.X3 = P.ws + PF_OFF_X3 + (size_t) par * (PF_MAXWG + 1) * 32 * PF_ZCOLS;
    f.X4 = P.ws + PF_OFF_X4 + (size_t) par * 32 * PF_ZCOLS;
    return f;
}

// reduce-scatter of Z (all seven waves): this workgroup's columns j = g, g + nwg, ... of Z, two per wave at a time (half-wave =
// column):  z = sum of the partials,  y = U'^-T z,  W(:, j) = T^T y -> X4  or  G(j', c + :) = y
__device__ __forceinline__ void pf_fold(const PfArgs& P, const PfLeaf& f, const PfLds& L, int g, int nwg, int wave, int lane)
{
    double* s1 = L.scr + wave * 128;
    const int h = lane >> 5, i = lane & 31;
    for (int k = 2 * wave + h; ; k += 14) {
        const int j = g + k * nwg;
        const bool have = j < f.ncols;
        if (!__any(have)) break;
        if (have) {
            double z = 0.0;
            const int zi = pf_zidx(i, j);
            const double corr = pf_ld(f.X3 + (size_t) nwg * 32 * PF_ZCOLS + zi);             // the top-block owner's -B^T x_top
            {
                double v[PF_MAXWG];
#pragma unroll
                for (int u = 0; u < PF_MAXWG; ++u) v[u] = pf_ld(f.X3 + (size_t) min(u, nwg - 1) * 32 * PF_ZCOLS + zi);
#pragma unroll
                for (int u = 0; u < PF_MAXWG; ++u)
                    if (u < nwg) z += v[u];
            }
            s1[h * 32 + i] = z + corr;
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        double y = 0.0;
        if (have) {
#pragma unroll
            for (int kk = 0; kk < 32; ++kk) y += L.Uinv[kk][i] * s1[h * 32 + kk];        // U'^-1 is stored with its zeros
            s1[64 + h * 32 + i] = y;
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        if (have) {
            if (j >= f.nrest) {
                P.G[(size_t) (f.c + i) * P.ldg + (j - f.nrest)] = y;
            } else {
                double wv = 0.0;
#pragma unroll
                for (int cc = 0; cc < 32; ++cc) wv += L.Ts[cc][i] * s1[64 + h * 32 + cc];  // T is stored with its zeros
                pf_st(f.X4 + j * 32 + i, wv);
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
    }
}

// Row waves: A_rest(:, 32 g .. ) -= V W for this lane's rows and the column groups g1 - 1 down to g0 (32 columns each) of the leaf at
// column c: V in vr (L_row layout), W straight from the gathered slab X4 (sc1 loads, the next group's requested a group ahead, as is
// the next 64 x 16 piece of A_rest).  keep: the LAST group's result (group g0) is returned in vr instead of V (it is the next leaf's a).
__device__ __forceinline__ void pf_update_groups(double (&vr)[4][8], const double* __restrict__ X4, double* __restrict__ A, int lda, int c,
                                                 int g0, int g1, bool keep, int r4, int r4c, bool act, int l15, int l4)
{
    if (g1 <= g0) return;
    double aw[2][8], awn[2][8];
    v4d ca[4], cbn[4];
    auto wload = [&](double (&w)[2][8], int jg) {
#pragma unroll
        for (int ti = 0; ti < 2; ++ti)
#pragma unroll
            for (int ks = 0; ks < 8; ++ks) w[ti][ks] = -pf_ld(X4 + (32 * jg + 16 * ti + l15) * 32 + 4 * ks + l4);
    };
    auto cptr = [&](int jg, int ti) { return A + (size_t) (c + 32 + 32 * jg + 16 * ti + l4) * lda; };
    auto cload = [&](v4d (&cc)[4], const double* cp) {
#pragma unroll
        for (int rr = 0; rr < 4; ++rr) {
            const double* q = cp + (size_t) (4 * rr) * lda + r4c;
            const v2d lo = *reinterpret_cast<const v2d*>(q), hi = *reinterpret_cast<const v2d*>(q + 2);
            cc[0][rr] = lo[0]; cc[1][rr] = lo[1]; cc[2][rr] = hi[0]; cc[3][rr] = hi[1];
        }
    };
    auto cstore = [&](const v4d (&cc)[4], double* cp) {
        if (act) {
#pragma unroll
            for (int rr = 0; rr < 4; ++rr) {
                double* q = cp + (size_t) (4 * rr) * lda + r4;
                *reinterpret_cast<v2d*>(q) = (v2d){cc[0][rr], cc[1][rr]};
                *reinterpret_cast<v2d*>(q + 2) = (v2d){cc[2][rr], cc[3][rr]};
            }
        }
    };
    auto mma = [&](v4d (&cc)[4], const double (&w)[8]) {
#pragma unroll
        for (int t = 0; t < 4; ++t)
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
            for (int t = 0; t < 4; ++t)
#pragma unroll
                for (int rr = 0; rr < 4; ++rr) { vr[t][rr] = ca[t][rr]; vr[t][4 + rr] = cbn[t][rr]; }
        }
    }
}

// The two roles run the same sequence of workgroup barriers (numbered per leaf); a role with nothing to do in a phase just joins.
//
//   #1  image of a                 #2  per-wave G1          #3  G1 partial published     #4  all epoch words seen
//   #5  G1 summed                  #6  R1, R1^-1            #7  Q, image, Q_top          #8  per-wave G2
//   #9  G2 partial published       #10 all seen             #11 G2 summed, guard         (#12: none -- LU, then U and U'^-1, run beside
//   #13 Z published | factors ready                          the row waves' product; hand-offs through LDS words)                       #14 T, outputs | V           #15 all Z seen
//   #16 W slices published         #17 all seen
__device__ __forceinline__ void pf_rows(const PfArgs& P, double* sm, int g, int nwg)
{
    const PfLds L = pf_lds(sm);
    const int tid = threadIdx.x, wave = tid >> 6;
    const int mk = P.mk, wh = P.wh, lda = P.lda, ldv = P.ldv;
    double* const A = P.A;
    double* const Vw = P.Vw;
    unsigned* const flags = reinterpret_cast<unsigned*>(P.ws);
    unsigned ep = P.epoch0;
    const int wgrow0 = g * PF_ROWS;
    double ar[4][8];
    int nfallback = 0;
    if (tid == 0) { L.gflags[4] = 0; L.gflags[5] = 0; L.gflags[6] = 0; }
    {
        const int l15 = tid & 15, l4 = (tid & 63) >> 4;
        pf_load_rows(ar, A, lda, 0, min(wgrow0 + wave * 64 + 4 * l15, mk - 4), l4);
    }
    for (int c = 0; c < wh; c += 32) {
        const PfLeaf f = pf_leaf(P, c);
        int lane = tid & 63;                                  // opaque once per leaf: keeps the lane-dependent addresses and selects
        asm volatile("" : "+v"(lane));                        // of the unrolled bodies below from being hoisted out of this loop
        const int l15 = lane & 15, l4 = lane >> 4;
        const int r4 = wgrow0 + wave * 64 + 4 * l15;         // this lane's four rows r4 .. r4 + 3
        const int r4c = min(r4, mk - 4);
        const bool act = r4 >= c && r4 < mk;
        const bool toprow = g == f.gown && r4 >= c && r4 < c + 32;
        PF_STAMP(0);
        if (tid == 0) { L.gflags[0] = 1; L.gflags[1] = 0; L.gflags[2] = 0; L.gflags[3] = 0; }
        if (!act) {
#pragma unroll
            for (int t = 0; t < 4; ++t)
#pragma unroll
                for (int ks = 0; ks < 8; ++ks) ar[t][ks] = 0.0;
        }
        pf_image_write(L.img, ar, wave, l15, l4);
        __syncthreads();                                                             // #1
        PF_STAMP(1);
        pf_gram_wave(L.img, L.part, wave, lane, l15, l4);
        __syncthreads();                                                             // #2
        PF_STAMP(2);
        pf_gram_publish(L.part, f.X1 + (size_t) g * 1024);
        pf_publish(flags, g, ++ep);                                                  // #3
        PF_STAMP(3);
        pf_wait(flags, nwg, ep, &L.gflags[4]);                                       // #4
        PF_STAMP(4);
        pf_gram_sum(f.X1, nwg, L.Gs, L.gflags, false);
        PF_STAMP(20);
        __syncthreads();                                                             // #5
        PF_STAMP(5);
        // (service wave 0 is in the Cholesky.)  The PREVIOUS leaf's update of everything beyond this leaf's columns happens here, off
        // the critical chain: this leaf only needed its own 32 columns (done at the end of the previous pass, below).  V of the
        // previous leaf comes back from Vw (this lane's own rows); this leaf's a is parked in its LDS image meanwhile
        if (c > 0 && f.nrest > 0) {
            const int cp_ = c - 32;
            const bool actp = r4 >= cp_ && r4 < mk;
            pf_load_rows(ar, Vw, ldv, cp_, r4c, l4);
            if (!actp) {
#pragma unroll
                for (int t = 0; t < 4; ++t)
#pragma unroll
                    for (int ks = 0; ks < 8; ++ks) ar[t][ks] = 0.0;
            }
            pf_update_groups(ar, P.ws + PF_OFF_X4 + (size_t) (((cp_ >> 5) & 1)) * 32 * PF_ZCOLS, A, lda, cp_, 1, (wh - cp_ - 32) / 32, false,
                             r4, r4c, actp, l15, l4);
            pf_image_read(L.img, ar, wave, l15, l4);
        }
        __syncthreads();                                                             // #6  (service wave 0: Cholesky)
        PF_STAMP(6);
        // Q = A R1^-1 (registers), its image, the top block of Q -> QT
        pf_rows_times_upper(ar, L.Ws, l15, l4);
        pf_image_write(L.img, ar, wave, l15, l4);
        if (toprow) {
#pragma unroll
            for (int ks = 0; ks < 8; ++ks)
#pragma unroll
                for (int t = 0; t < 4; ++t) pf_st(f.QT + (4 * ks + l4) * 32 + (r4 + t - c), ar[t][ks]);
        }
        __syncthreads();                                                             // #7
        PF_STAMP(7);
        pf_gram_wave(L.img, L.part, wave, lane, l15, l4);
        __syncthreads();                                                             // #8
        PF_STAMP(8);
        pf_gram_publish(L.part, f.X2 + (size_t) g * 1024);
        pf_publish(flags, g, ++ep);                                                  // #9
        PF_STAMP(9);
        pf_wait(flags, nwg, ep, &L.gflags[4]);                                       // #10
        PF_STAMP(10);
        pf_gram_sum(f.X2, nwg, L.Gs, L.gflags, true);
        __syncthreads();                                                             // #11
        PF_STAMP(11);
        // Z = Q^T [A_rest | V_prev] for this workgroup's rows, 16-column tiles dealt to the waves (service wave 0: modified LU).
        // A tile's 256 rows go by in four chunks of 64; the rows of the next chunk (or of the next tile) are requested before the
        // 32 matrix-core instructions of the current one (the first version waited for every chunk: 9 us per tile instead of 3.4),
        // and a tile is published as soon as it is complete
        const int ntile = f.ncols / 16;
        {
            const int nval = (ntile > wave) ? (ntile - wave + 3) / 4 : 0;
            double* X3g = f.X3 + (size_t) g * 32 * PF_ZCOLS;
            double xb[2][4][4];
            auto xptr = [&](int slot) {
                const int j = 16 * (wave + 4 * slot) + l15;
                return (j < f.nrest) ? A + (size_t) (c + 32 + j) * lda : Vw + (size_t) (j - f.nrest) * ldv;
            };
            auto xload = [&](double (&x)[4][4], const double* xp, int ch) {
#pragma unroll
                for (int s4 = 0; s4 < 4; ++s4) {
                    const int rowc = min(wgrow0 + 16 * (4 * ch + s4) + 4 * l4, mk - 4);
                    const v2d lo = *reinterpret_cast<const v2d*>(xp + rowc), hi = *reinterpret_cast<const v2d*>(xp + rowc + 2);
                    x[s4][0] = lo[0]; x[s4][1] = lo[1]; x[s4][2] = hi[0]; x[s4][3] = hi[1];
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
            if (nval > 0) xload(xb[0], xptr(0), 0);
            for (int slot = 0; slot < nval; ++slot) {
                v4d acc0 = (v4d){0.0, 0.0, 0.0, 0.0}, acc1 = acc0;
                const double* xp = xptr(slot);
                xload(xb[1], xp, 1); mma(xb[0], 0, acc0, acc1);
                xload(xb[0], xp, 2); mma(xb[1], 1, acc0, acc1);
                xload(xb[1], xp, 3); mma(xb[0], 2, acc0, acc1);
                if (slot + 1 < nval) xload(xb[0], xptr(slot + 1), 0);
                mma(xb[1], 3, acc0, acc1);
                double* zp = X3g + (wave + 4 * slot) * 512 + 2 * lane;          // pf_zidx layout
                pf_st2(zp, acc0[0], acc0[1]); pf_st2(zp + 128, acc0[2], acc0[3]);
                pf_st2(zp + 256, acc1[0], acc1[1]); pf_st2(zp + 384, acc1[2], acc1[3]);
            }
        }
        PF_STAMP(12);
        pf_publish(flags, g, ++ep);                                                  // #13
        PF_STAMP(13);
        if (L.gflags[0] == 0 || L.gflags[1] != 0 || L.gflags[3] != 0) ++nfallback;  // (Householder route: below; until then garbage)
        // V = Q U'^-1; the top block's rows become L1 (their copy in global memory comes from the service waves)
        pf_rows_times_upper(ar, L.Uinv, l15, l4);
        if (toprow) {
#pragma unroll
            for (int t = 0; t < 4; ++t)
#pragma unroll
                for (int ks = 0; ks < 8; ++ks) {
                    const int rr_ = r4 + t - c, col = 4 * ks + l4;
                    ar[t][ks] = (col < rr_) ? L.Bs[rr_][col] : (col == rr_ ? 1.0 : 0.0);
                }
        } else if (act) {
            pf_store_rows(ar, Vw, ldv, c, r4, l4);
            pf_store_rows(ar, A, lda, c, r4, l4);
        }
        __syncthreads();                                                             // #14
        PF_STAMP(14);
        pf_wait(flags, nwg, ep, &L.gflags[4]);                                       // #15
        PF_STAMP(15);
        pf_fold(P, f, L, g, nwg, wave, lane);
        PF_STAMP(21);
        pf_publish(flags, g, ++ep);                                                  // #16
        PF_STAMP(16);
        pf_wait(flags, nwg, ep, &L.gflags[4]);                                       // #17
        PF_STAMP(17);
        // the next leaf's 32 columns are updated now (the result stays in registers as its a); the other columns of A_rest wait for
        // the next pass's Cholesky window (above)
        if (f.nrest > 0) pf_update_groups(ar, f.X4, A, lda, c, 0, 1, true, r4, r4c, act, l15, l4);
        PF_STAMP(18);
    }
    if (g == 0 && tid == 0) {
        if (nfallback) atomicAdd(P.status, nfallback);
        if (L.gflags[4]) P.status[1] = 1;
    }
}

__device__ __forceinline__ void pf_service(const PfArgs& P, double* sm, int g, int nwg)
{
    const PfLds L = pf_lds(sm);
    const int tid = threadIdx.x, wave = tid >> 6, sw = wave - 4;
    // the recurrences on these waves ARE the critical chain; the row waves they share SIMDs with were dispatched first, and at equal
    // priority the older wave wins the vector-issue arbitration (MI355X_MICROARCH.md, two waves per SIMD): beside a row wave in its
    // product the LU took 2.5x as long
    __builtin_amdgcn_s_setprio(3);
    for (int c = 0; c < P.wh; c += 32) {
        const PfLeaf f = pf_leaf(P, c);
        // the lane index is made opaque once per leaf: otherwise every lane-dependent constant of the unrolled recurrences below
        // (identity columns, `lane == K` selects, ...) is hoisted out of this loop and kept in registers across it -- ~400 of them
        int lane = tid & 63;
        asm volatile("" : "+v"(lane));
        const int rc = lane & 31;
        __syncthreads();                                                             // #1
        __syncthreads();                                                             // #2
        pf_gram_publish(L.part, f.X1 + (size_t) g * 1024);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();                                                             // #3
        __syncthreads();                                                             // #4
        pf_gram_sum(f.X1, nwg, L.Gs, L.gflags, false);
        __syncthreads();                                                             // #5
        // R1 = chol(G1) and R1^-1 on one wave (the identity columns ride on the wave's upper half)
        if (sw == 0) {
            double gg[PW];
#pragma unroll
            for (int i = 0; i < PW; ++i) gg[i] = (lane < PW) ? L.Gs[rc][i] : (i == rc ? 1.0 : 0.0);
            int e2 = 0;                                        // even power-of-two scaling, as in cholq3_kernel
            {
                const double d = readlane_f64(gg[0], 0);
                if (d > 0.0 && d < 1.7e308) { (void) frexp(d, &e2); e2 &= ~1; }
            }
            const double rs = ldexp(1.0, e2 / 2), wsc = ldexp(1.0, -(e2 / 2));
            if (lane < PW) {
#pragma unroll
                for (int i = 0; i < PW; ++i) gg[i] = (gg[i] * wsc) * wsc;
            }
            bool ok = true;
            CholAugStep<0>::run(gg, lane, ok);
            if (lane >= PW) {
#pragma unroll
                for (int k = 0; k < PW; ++k) L.Ws[rc][k] = (k >= rc) ? gg[k] * wsc : 0.0;       // row rc of R1^-1
            } else {
#pragma unroll
                for (int k = 0; k < PW; ++k) L.R1s[k][rc] = (k <= rc) ? gg[k] * rs : 0.0;       // column rc of R1
            }
            if (lane == 0 && !ok) L.gflags[0] = 0;
        }
        __syncthreads();                                                             // #6
        __syncthreads();                                                             // #7
        __syncthreads();                                                             // #8
        pf_gram_publish(L.part, f.X2 + (size_t) g * 1024);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();                                                             // #9
        __syncthreads();                                                             // #10
        pf_gram_sum(f.X2, nwg, L.Gs, L.gflags, true);
        __syncthreads();                                                             // #11
        PF_STAMP_S(23);
        // R2 = chol(G2) (to first order when G2 - I is tiny) and the modified LU  Q_top - S R2 = L1 U' on service wave 0.  Its upper
        // 32 lanes carry the columns of the identity through the same row operations: they end as L1^-1, for nothing.
        // Then (no workgroup barrier in between: the row waves are in their long product):
        //   wave 0: U = U' R2^-1 -- with the first-order R2, R2^-1 = 2 I - R2 to ~1e-17 and U = 2 U' - U' R2 is three tiles on the
        //           matrix cores; the general R2 (leaves of condition > ~1e4) takes the row solve
        //   wave 2: inverse of the lower diagonal block of U'          wave 1: inverse of the upper one, then the off-diagonal block
        //           -X11 U'12 X22 on the matrix cores  (two 16-step recurrences + 8 MFMAs instead of one 32-step recurrence)
        const int seq = (c >> 5) + 1;
        if (sw == 0) {
            const bool refused = L.gflags[0] == 0 || L.gflags[1] != 0;
            const bool first_order = L.gflags[2] == 0;
            double gg[PW], b[PW];
            if (lane < PW) {
#pragma unroll
                for (int r = 0; r < PW; ++r) b[r] = pf_ld(f.QT + rc * 32 + r);
#pragma unroll
                for (int i = 0; i < PW; ++i) gg[i] = L.Gs[rc][i];
            } else {
#pragma unroll
                for (int r = 0; r < PW; ++r) { b[r] = (r == rc) ? 1.0 : 0.0; gg[r] = 0.0; }
            }
            bool ok = true;
            double dinv = 1.0, sgn = 1.0;
            if (!first_order) {
                Chol3Step<0>::run(gg, lane, ok, dinv);
            } else if (lane < PW) {
#pragma unroll
                for (int i = 0; i < PW; ++i) gg[i] = (i < rc) ? gg[i] : (i == rc ? 1.0 + 0.5 * (gg[i] - 1.0) : 0.0);
                double d = 1.0;
#pragma unroll
                for (int i = 0; i < PW; ++i) d = (i == rc) ? gg[i] : d;
                dinv = 1.0 / d;
            }
            if (ok && !refused) {
                Hr3Lu<0>::run(b, gg, lane, sgn);
                if (lane < PW) {
#pragma unroll
                    for (int k = 0; k < PW; ++k) { L.R2s[k][lane] = gg[k]; L.Bs[k][lane] = b[k]; }
                    L.r2inv[lane] = dinv;
                    L.Ss[lane] = sgn;
                } else {                                   // Ls aliases Gs: this wave has G2 in registers since the start of the LU
#pragma unroll
                    for (int k = 0; k < PW; ++k) L.Ls[k][rc] = (k >= rc) ? b[k] : 0.0;           // Ls[i][j] = L1^-1(i, j)
                }
            } else if (lane == 0) L.gflags[3] = 1;
            PF_STAMP_S(24);
            pf_lds_signal(&L.gflags[5], seq);
            __builtin_amdgcn_wave_barrier();
            // U -> Us (aliases R1^-1: dead since the row waves formed Q)
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
            pf_lds_await(&L.gflags[5], seq);
            const int o = (sw == 2) ? 16 : 0, j = lane & 15;
            double x[16];
            PfUpperInv16<15>::run(x, L.Bs, o, j);
            if (lane < 16) {                               // Uinv aliases the Gram partials: dead since G2 was published
#pragma unroll
                for (int i = 0; i < 16; ++i) L.Uinv[o + i][o + lane] = (i <= lane) ? x[i] : 0.0;  // Uinv[k][c] = U'^-1(k, c)
                if (sw == 2)
#pragma unroll
                    for (int i = 0; i < 16; ++i) L.Uinv[16 + i][lane] = 0.0;
            }
            if (sw == 2) {
                pf_lds_signal(&L.gflags[6], seq);
            } else {
                pf_lds_await(&L.gflags[6], seq);
                // X12 = -X11 (U'12 X22): two 16 x 16 x 16 products; the intermediate goes through this wave's scratch
                const int l15 = lane & 15, l4 = lane >> 4;
                double* tmp = L.scr + wave * 128;          // 16 x 16 is 256 doubles: use the scratch of waves 5 and 6 (adjacent)
                v4d acc = (v4d){0.0, 0.0, 0.0, 0.0};
#pragma unroll
                for (int ks = 0; ks < 4; ++ks) {           // P = U'12 X22:  P(i, c) = sum_k U'(i, 16 + k) X22(k, c)
                    const int k = 4 * ks + l4;
                    acc = pf_mfma(L.Bs[l15][16 + k], L.Uinv[16 + k][16 + l15], acc);
                }
#pragma unroll
                for (int r = 0; r < 4; ++r) tmp[(l4 + 4 * r) * 16 + l15] = acc[r];            // P(i = l4 + 4 r, c = l15)
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
                acc = (v4d){0.0, 0.0, 0.0, 0.0};
#pragma unroll
                for (int ks = 0; ks < 4; ++ks) {           // X12(i, c) = -sum_k X11(i, k) P(k, c)
                    const int k = 4 * ks + l4;
                    acc = pf_mfma(-L.Uinv[l15][k], tmp[k * 16 + l15], acc);
                }
#pragma unroll
                for (int r = 0; r < 4; ++r) L.Uinv[l4 + 4 * r][16 + l15] = acc[r];
            }
        }
        PF_STAMP_S(25);
        // Q = V U' + [B; 0] with B = S R2, so V^T x = U'^-T (Q^T x - B^T x_top): the owner of the top block (its rows are x_top)
        // publishes the correction -B^T X_top as one more partial of Z (slot nwg) -- on the service waves, which are done with the
        // factors well before the row waves are with their product
        if (g == f.gown && L.gflags[3] == 0) {
            const int l15 = lane & 15, l4 = lane >> 4;
            const int ntile = f.ncols / 16;
            if (sw == 0) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
            double* X3c = f.X3 + (size_t) nwg * 32 * PF_ZCOLS;
            double ba[2][8];
#pragma unroll
            for (int ti = 0; ti < 2; ++ti)
#pragma unroll
                for (int ks = 0; ks < 8; ++ks) ba[ti][ks] = -L.Ss[4 * ks + l4] * L.R2s[4 * ks + l4][16 * ti + l15];
            for (int jt = sw; jt < ntile; jt += 3) {
                const int j = 16 * jt + l15;
                const double* xp = ((j < f.nrest) ? P.A + (size_t) (c + 32 + j) * P.lda : P.Vw + (size_t) (j - f.nrest) * P.ldv) + c;
                double xt[8];
#pragma unroll
                for (int ks = 0; ks < 8; ++ks) xt[ks] = xp[4 * ks + l4];
                v4d acc0 = (v4d){0.0, 0.0, 0.0, 0.0}, acc1 = acc0;
#pragma unroll
                for (int ks = 0; ks < 8; ++ks) {
                    acc0 = pf_mfma(ba[0][ks], xt[ks], acc0);
                    acc1 = pf_mfma(ba[1][ks], xt[ks], acc1);
                }
                double* zp = X3c + jt * 512 + 2 * lane;
                pf_st2(zp, acc0[0], acc0[1]); pf_st2(zp + 128, acc0[2], acc0[3]);
                pf_st2(zp + 256, acc1[0], acc1[1]); pf_st2(zp + 384, acc1[2], acc1[3]);
            }
        }
        PF_STAMP_S(26);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();                                                             // #13
        // T = -U S L1^-T (every workgroup keeps it in LDS) and R = S R2 R1 on the matrix cores, one 16 x 16 tile of each per wave
        // (tiles (0,0), (0,1), (1,1); the first version's scalar loops over LDS took 14 us); the owner of the top block writes
        // R, L1, T and tau to global memory
        {
            const int l15 = lane & 15, l4 = lane >> 4;
            const int ti = (sw == 2) ? 1 : 0, tc = (sw == 0) ? 0 : 1;
            const bool own = g == f.gown;
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
                const double tv = (cc >= i) ? tt[r] : 0.0;
                L.Ts[i][cc] = tv;
                if (own) {
                    P.T[(size_t) (c + cc) * P.ldt + c + i] = tv;
                    if (i == cc) P.tau[c + i] = tv;
                    P.A[(size_t) (c + cc) * P.lda + c + i] = (cc >= i) ? rt[r] : L.Bs[i][cc];
                    P.Vw[(size_t) (c + cc) * P.ldv + c + i] = (cc < i) ? L.Bs[i][cc] : (cc == i ? 1.0 : 0.0);
                }
            }
            if (sw == 0) {                                   // tile (1, 0): zeros of T, L1
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int i = 16 + l4 + 4 * r, cc = l15;
                    L.Ts[i][cc] = 0.0;
                    if (own) {
                        P.T[(size_t) (c + cc) * P.ldt + c + i] = 0.0;
                        P.A[(size_t) (c + cc) * P.lda + c + i] = L.Bs[i][cc];
                        P.Vw[(size_t) (c + cc) * P.ldv + c + i] = L.Bs[i][cc];
                    }
                }
            }
        }
        __syncthreads();                                                             // #14
        __syncthreads();                                                             // #15
        pf_fold(P, f, L, g, nwg, wave, lane);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();                                                             // #16
        __syncthreads();                                                             // #17
    }
}

__global__ __launch_bounds__(PF_THREADS) void panel_fused_kernel(PfArgs P)
{
    extern __shared__ __attribute__((aligned(16))) double sm[];
    // whole waves take one role or the other: the barriers in the two bodies pair up one to one
    if (__builtin_amdgcn_readfirstlane((int) (threadIdx.x >> 6)) < 4) pf_rows(P, sm, blockIdx.x, gridDim.x);
    else pf_service(P, sm, blockIdx.x, gridDim.x);
}

static long long* g_pf_stamps = nullptr;      // development only (qrd_panel_fused_set_stamps)

extern "C" {

void qrd_panel_fused_set_stamps(long long* d) { g_pf_stamps = d; }

size_t qrd_panel_fused_ws_doubles(void) { return (size_t) PF_WS_DOUBLES; }

int qrd_panel_fused_init(void)
{
    return (int) hipFuncSetAttribute(reinterpret_cast<const void*>(panel_fused_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                     (int) (PF_SM_DOUBLES * sizeof(double)));
}

// 1 when the one-launch panel can take this (half-)panel: whole 32-column leaves, at most 256 columns, at most 32 x 256 rows and as many
// free compute units on the stream, vector-aligned operands
int qrd_panel_fused_ok(void* stream, const double* A, int lda, int mk, int wh, const double* Vw, int ldv)
{
    if (wh < 32 || wh > 256 || wh % 32 || mk < wh || mk % 4) return 0;
    if ((reinterpret_cast<uintptr_t>(A) & 15) || (reinterpret_cast<uintptr_t>(Vw) & 15) || lda % 2 || ldv % 2) return 0;
    const int nwg = (mk + PF_ROWS - 1) / PF_ROWS;
    int cus = qrd_stream_cus(stream);
    if (cus > PF_MAXWG) cus = PF_MAXWG;
    return nwg <= cus;
}

// *epoch: the caller's epoch counter for this workspace (starts at 0 with a zeroed workspace); advanced by the launch
int qrd_panel_fused(void* stream, double* A, int lda, int mk, int wh, double* tau, double* T, int ldt, double* Vw, int ldv,
                    double* G, int ldg, double* ws, unsigned* epoch, int* status)
{
    if (!qrd_panel_fused_ok(stream, A, lda, mk, wh, Vw, ldv) || !ws || !epoch || !status) return -7;
    PfArgs a;
    a.A = A; a.lda = lda; a.mk = mk; a.wh = wh; a.Vw = Vw; a.ldv = ldv; a.T = T; a.ldt = ldt; a.tau = tau; a.G = G; a.ldg = ldg;
    a.ws = ws; a.epoch0 = *epoch; a.status = status; a.stamps = g_pf_stamps;
    *epoch += 1024u;
    const int nwg = (mk + PF_ROWS - 1) / PF_ROWS;
    hipLaunchKernelGGL(panel_fused_kernel, dim3(nwg), dim3(PF_THREADS), PF_SM_DOUBLES * sizeof(double), (hipStream_t) stream, a);
    return (int) hipGetLastError();
}

}   // extern "C"
