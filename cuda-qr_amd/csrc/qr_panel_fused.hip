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
#define PF_OFF_X4 (PF_OFF_X3 + 2 * PF_MAXWG * 32 * PF_ZCOLS)
#define PF_OFF_XF (PF_OFF_X4 + 2 * 32 * PF_ZCOLS)
#define PF_WS_DOUBLES (PF_OFF_XF + 2 * PF_MAXWG * 128)

// LDS carve-up (doubles)
#define PF_M33 (32 * 33)
#define PF_SM_IMG 0
#define PF_SM_PART (32 * PF_LDQ)                 /* 4 x 768 per-wave Gram partials; later U'^-1 and T */
#define PF_SM_GS (PF_SM_PART + 3072)             /* sum of the partial Gram matrices; later L1^-1 */
#define PF_SM_R1 (PF_SM_GS + PF_M33)
#define PF_SM_WS (PF_SM_R1 + PF_M33)             /* R1^-1; later U */
#define PF_SM_R2 (PF_SM_WS + PF_M33)
#define PF_SM_BS (PF_SM_R2 + PF_M33)             /* L1 \ U' */
#define PF_SM_SS (PF_SM_BS + PF_M33)             /* S (32), 1 / diag R2 (32) */
#define PF_SM_SCR (PF_SM_SS + 64)                /* 7 waves x 128 */
#define PF_SM_FLAGS (PF_SM_SCR + 7 * 128)        /* ints */
#define PF_SM_DOUBLES (PF_SM_FLAGS + 8)

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
};

typedef double (*pf_m33)[33];

__device__ __forceinline__ double pf_ld(const double* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void pf_st(double* p, double v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
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

// all threads: sum the four waves' partials and write this workgroup's 32 x 32 partial Gram matrix (dense, column-major) to X with sc1
__device__ __forceinline__ void pf_gram_publish(const double* part, double* __restrict__ X)
{
    for (int e = threadIdx.x; e < 768; e += PF_THREADS) {
        const double s = (part[e] + part[768 + e]) + (part[1536 + e] + part[2304 + e]);
        const int tile = e >> 8, rr = (e >> 6) & 3, ln = e & 63, p = (ln >> 4) + 4 * rr, q = ln & 15;
        const int i = (tile == 2) ? 16 + p : p, j = (tile == 0) ? q : 16 + q;
        pf_st(X + j * 32 + i, s);
        if (tile == 1) pf_st(X + i * 32 + j, s);
    }
}

// all threads: Gs[j][i] = sum over the workgroups (in index order) of their partial G(i, j)
__device__ __forceinline__ void pf_gram_sum(const double* __restrict__ X, int nwg, pf_m33 Gs, int* gflags, bool check)
{
    for (int e = threadIdx.x; e < 1024; e += PF_THREADS) {
        double s = 0.0;
        for (int w0 = 0; w0 < nwg; w0 += 8) {
            double v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = pf_ld(X + (size_t) min(w0 + u, nwg - 1) * 1024 + e);
#pragma unroll
            for (int u = 0; u < 8; ++u)
                if (w0 + u < nwg) s += v[u];
        }
        const int i = e & 31, j = e >> 5;
        Gs[j][i] = s;
        if (check) {
            const double d = s - (i == j ? 1.0 : 0.0);
            if (!(fabs(d) <= QRD_GUARD_THR)) gflags[1] = 1;       // also catches NaN
            if (!(fabs(d) <= QRD_CHOL1_THR)) gflags[2] = 1;
        }
    }
}

// LDS views shared by the two roles
struct PfLds {
    double* img; double* part;
    pf_m33 Uinv, Ts, Gs, Ls, R1s, Ws, Us, R2s, Bs;
    double *Ss, *r2inv, *scr;
    int* gflags;     // [0] first Cholesky ok, [1] |G2 - I| > 1/64, [2] > 1e-9, [3] second Cholesky failed, [4] dead (a wait timed out)
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
    f.X3 = P.ws + PF_OFF_X3 + (size_t) par * PF_MAXWG * 32 * PF_ZCOLS;
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
            for (int w0 = 0; w0 < nwg; w0 += 8) {
                double v[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) v[u] = pf_ld(f.X3 + (size_t) min(w0 + u, nwg - 1) * 32 * PF_ZCOLS + j * 32 + i);
#pragma unroll
                for (int u = 0; u < 8; ++u)
                    if (w0 + u < nwg) z += v[u];
            }
            s1[h * 32 + i] = z;
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        double y = 0.0;
        if (have) {
            for (int kk = 0; kk <= i; ++kk) y += L.Uinv[kk][i] * s1[h * 32 + kk];
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
                for (int cc = 0; cc <= i; ++cc) wv += L.Ts[cc][i] * s1[64 + h * 32 + cc];
                pf_st(f.X4 + j * 32 + i, wv);
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
    }
}

// The two roles run the same sequence of workgroup barriers (numbered per leaf); a role with nothing to do in a phase just joins.
//
//   #1  image of a                 #2  per-wave G1          #3  G1 partial published     #4  all epoch words seen
//   #5  G1 summed                  #6  R1, R1^-1            #7  Q, image, Q_top          #8  per-wave G2
//   #9  G2 partial published       #10 all seen             #11 G2 summed, guard         #12 LU | product
//   #13 Z published | U, L1^-1, U'^-1                       #14 T, outputs | V           #15 all Z seen
//   #16 W slices published         #17 all seen             #18 update
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
    if (tid == 0) L.gflags[4] = 0;
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
        if (tid == 0) { L.gflags[0] = 1; L.gflags[1] = 0; L.gflags[2] = 0; L.gflags[3] = 0; }
        if (!act) {
#pragma unroll
            for (int t = 0; t < 4; ++t)
#pragma unroll
                for (int ks = 0; ks < 8; ++ks) ar[t][ks] = 0.0;
        }
        pf_image_write(L.img, ar, wave, l15, l4);
        __syncthreads();                                                             // #1
        pf_gram_wave(L.img, L.part, wave, lane, l15, l4);
        __syncthreads();                                                             // #2
        pf_gram_publish(L.part, f.X1 + (size_t) g * 1024);
        pf_publish(flags, g, ++ep);                                                  // #3
        pf_wait(flags, nwg, ep, &L.gflags[4]);                                       // #4
        pf_gram_sum(f.X1, nwg, L.Gs, L.gflags, false);
        __syncthreads();                                                             // #5
        __syncthreads();                                                             // #6  (service wave 0: Cholesky)
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
        pf_gram_wave(L.img, L.part, wave, lane, l15, l4);
        __syncthreads();                                                             // #8
        pf_gram_publish(L.part, f.X2 + (size_t) g * 1024);
        pf_publish(flags, g, ++ep);                                                  // #9
        pf_wait(flags, nwg, ep, &L.gflags[4]);                                       // #10
        pf_gram_sum(f.X2, nwg, L.Gs, L.gflags, true);
        __syncthreads();                                                             // #11
        // Z = Q^T [A_rest | V_prev] for this workgroup's rows, 16-column tiles dealt to the waves (service wave 0: modified LU)
        v4d zt[4][2];
        const int ntile = f.ncols / 16;
#pragma unroll
        for (int slot = 0; slot < 4; ++slot) {
            zt[slot][0] = (v4d){0.0, 0.0, 0.0, 0.0};
            zt[slot][1] = (v4d){0.0, 0.0, 0.0, 0.0};
            const int jt = wave + 4 * slot;
            if (jt < ntile) {
                const int j = 16 * jt + l15;
                const double* xp = (j < f.nrest) ? A + (size_t) (c + 32 + j) * lda : Vw + (size_t) (j - f.nrest) * ldv;
                v4d acc0 = (v4d){0.0, 0.0, 0.0, 0.0}, acc1 = acc0;
#pragma unroll 4
                for (int sg = 0; sg < 16; ++sg) {
                    const int rowc = min(wgrow0 + 16 * sg + 4 * l4, mk - 4);
                    const v2d xa = *reinterpret_cast<const v2d*>(xp + rowc), xb = *reinterpret_cast<const v2d*>(xp + rowc + 2);
                    const double* q0 = L.img + l15 * PF_LDQ + 16 * sg + 4 * l4;
                    const double* q1 = q0 + 16 * PF_LDQ;
                    const v2d qa0 = *reinterpret_cast<const v2d*>(q0), qa1 = *reinterpret_cast<const v2d*>(q0 + 2);
                    const v2d qb0 = *reinterpret_cast<const v2d*>(q1), qb1 = *reinterpret_cast<const v2d*>(q1 + 2);
                    acc0 = pf_mfma(qa0[0], xa[0], acc0); acc1 = pf_mfma(qb0[0], xa[0], acc1);
                    acc0 = pf_mfma(qa0[1], xa[1], acc0); acc1 = pf_mfma(qb0[1], xa[1], acc1);
                    acc0 = pf_mfma(qa1[0], xb[0], acc0); acc1 = pf_mfma(qb1[0], xb[0], acc1);
                    acc0 = pf_mfma(qa1[1], xb[1], acc0); acc1 = pf_mfma(qb1[1], xb[1], acc1);
                }
                zt[slot][0] = acc0; zt[slot][1] = acc1;
            }
        }
        __syncthreads();                                                             // #12
        if (L.gflags[0] == 0 || L.gflags[1] != 0 || L.gflags[3] != 0) ++nfallback;  // (Householder route: below; until then garbage)
        // the owner of the top block corrects its partial (z -= B^T x_top, B = S R2); everyone publishes Z -> X3
        if (g == f.gown) {
            double ba[2][8];
#pragma unroll
            for (int ti = 0; ti < 2; ++ti)
#pragma unroll
                for (int ks = 0; ks < 8; ++ks) ba[ti][ks] = -L.Ss[4 * ks + l4] * L.R2s[4 * ks + l4][16 * ti + l15];
#pragma unroll
            for (int slot = 0; slot < 4; ++slot) {
                const int jt = wave + 4 * slot;
                if (jt < ntile) {
                    const int j = 16 * jt + l15;
                    const double* xp = ((j < f.nrest) ? A + (size_t) (c + 32 + j) * lda : Vw + (size_t) (j - f.nrest) * ldv) + c;
                    double xt[8];
#pragma unroll
                    for (int ks = 0; ks < 8; ++ks) xt[ks] = xp[4 * ks + l4];
#pragma unroll
                    for (int ks = 0; ks < 8; ++ks) {
                        zt[slot][0] = pf_mfma(ba[0][ks], xt[ks], zt[slot][0]);
                        zt[slot][1] = pf_mfma(ba[1][ks], xt[ks], zt[slot][1]);
                    }
                }
            }
        }
        {
            double* X3g = f.X3 + (size_t) g * 32 * PF_ZCOLS;
#pragma unroll
            for (int slot = 0; slot < 4; ++slot) {
                const int jt = wave + 4 * slot;
                if (jt < ntile) {
#pragma unroll
                    for (int ti = 0; ti < 2; ++ti)
#pragma unroll
                        for (int rr = 0; rr < 4; ++rr) pf_st(X3g + (16 * jt + l15) * 32 + 16 * ti + 4 * rr + l4, zt[slot][ti][rr]);
                }
            }
        }
        pf_publish(flags, g, ++ep);                                                  // #13
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
        pf_wait(flags, nwg, ep, &L.gflags[4]);                                       // #15
        pf_fold(P, f, L, g, nwg, wave, lane);
        pf_publish(flags, g, ++ep);                                                  // #16
        pf_wait(flags, nwg, ep, &L.gflags[4]);                                       // #17
        // in-panel update  A_rest -= V W  for this lane's rows, 32 columns at a time; the next leaf's columns last: they stay in
        // registers as the next leaf's a
        for (int jg = f.nrest / 32 - 1; jg >= 0; --jg) {
            const int j0 = 32 * jg;
            double aw[2][8];
#pragma unroll
            for (int ti = 0; ti < 2; ++ti)
#pragma unroll
                for (int ks = 0; ks < 8; ++ks) aw[ti][ks] = -pf_ld(f.X4 + (j0 + 16 * ti + l15) * 32 + 4 * ks + l4);
            double keep[4][4];
#pragma unroll
            for (int ti = 0; ti < 2; ++ti) {
                v4d acc[4];
                double* cp = A + (size_t) (c + 32 + j0 + 16 * ti + l4) * lda;
#pragma unroll
                for (int rr = 0; rr < 4; ++rr) {
                    const double* p = cp + (size_t) (4 * rr) * lda + r4c;
                    const v2d lo = *reinterpret_cast<const v2d*>(p), hi = *reinterpret_cast<const v2d*>(p + 2);
                    acc[0][rr] = lo[0]; acc[1][rr] = lo[1]; acc[2][rr] = hi[0]; acc[3][rr] = hi[1];
                }
#pragma unroll
                for (int t = 0; t < 4; ++t)
#pragma unroll
                    for (int ks = 0; ks < 8; ++ks) acc[t] = pf_mfma(aw[ti][ks], ar[t][ks], acc[t]);
                if (act) {
#pragma unroll
                    for (int rr = 0; rr < 4; ++rr) {
                        double* p = cp + (size_t) (4 * rr) * lda + r4;
                        *reinterpret_cast<v2d*>(p) = (v2d){acc[0][rr], acc[1][rr]};
                        *reinterpret_cast<v2d*>(p + 2) = (v2d){acc[2][rr], acc[3][rr]};
                    }
                }
                if (jg == 0) {
                    if (ti == 0) {
#pragma unroll
                        for (int t = 0; t < 4; ++t)
#pragma unroll
                            for (int rr = 0; rr < 4; ++rr) keep[t][rr] = acc[t][rr];
                    } else {
#pragma unroll
                        for (int t = 0; t < 4; ++t)
#pragma unroll
                            for (int rr = 0; rr < 4; ++rr) { ar[t][rr] = keep[t][rr]; ar[t][4 + rr] = acc[t][rr]; }
                    }
                }
            }
        }
        __syncthreads();                                                             // #18
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
        // R2 = chol(G2) (to first order when G2 - I is tiny) and the modified LU  Q_top - S R2 = L1 U'
        if (sw == 0) {
            const bool refused = L.gflags[0] == 0 || L.gflags[1] != 0;
            double gg[PW], b[PW];
#pragma unroll
            for (int r = 0; r < PW; ++r) b[r] = pf_ld(f.QT + rc * 32 + r);
#pragma unroll
            for (int i = 0; i < PW; ++i) gg[i] = L.Gs[rc][i];
            bool ok = true;
            double dinv = 1.0, sgn = 1.0;
            if (L.gflags[2]) {
                Chol3Step<0>::run(gg, rc, ok, dinv);
            } else {
#pragma unroll
                for (int i = 0; i < PW; ++i) gg[i] = (i < rc) ? gg[i] : (i == rc ? 1.0 + 0.5 * (gg[i] - 1.0) : 0.0);
                double d = 1.0;
#pragma unroll
                for (int i = 0; i < PW; ++i) d = (i == rc) ? gg[i] : d;
                dinv = 1.0 / d;
            }
            if (ok && !refused) {
                Hr3Lu<0>::run(b, gg, rc, sgn);
                if (lane < PW) {
#pragma unroll
                    for (int k = 0; k < PW; ++k) { L.R2s[k][lane] = gg[k]; L.Bs[k][lane] = b[k]; }
                    L.r2inv[lane] = dinv;
                    L.Ss[lane] = sgn;
                }
            } else if (lane == 0) L.gflags[3] = 1;
        }
        __syncthreads();                                                             // #12
        // U = U' R2^-1 (rows), L1^-1 (columns), U'^-1 (columns): one wave each
        if (sw == 0) {
            double u[PW];
#pragma unroll
            for (int cc = 0; cc < PW; ++cc) u[cc] = (cc >= rc) ? L.Bs[rc][cc] : 0.0;
            RowSolve<0>::run(u, L.R2s, L.r2inv);
            if (lane < PW) {                               // Us aliases Ws (R1^-1): dead since the row waves formed Q
#pragma unroll
                for (int cc = 0; cc < PW; ++cc) L.Us[lane][cc] = (cc >= lane) ? u[cc] : 0.0;
            }
        } else if (sw == 1) {
            double x[PW];
            UnitLowerInv<0>::run(x, L.Bs, rc);
            if (lane < PW) {                               // Ls aliases Gs: dead since service wave 0 took G2 into registers
#pragma unroll
                for (int i = 0; i < PW; ++i) L.Ls[i][lane] = (i >= lane) ? x[i] : 0.0;          // Ls[i][j] = L1^-1(i, j)
            }
        } else {
            double x[PW];
            UpperInv<PW - 1>::run(x, L.Bs, rcp_newton(L.Bs[rc][rc]), rc);
            if (lane < PW) {                               // Uinv aliases the Gram partials: dead since G2 was published
#pragma unroll
                for (int i = 0; i < PW; ++i) L.Uinv[i][lane] = (i <= lane) ? x[i] : 0.0;        // Uinv[k][c] = U'^-1(k, c)
            }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();                                                             // #13
        // T = -U S L1^-T (every workgroup keeps it in LDS); the owner of the top block writes R, L1, T, tau
        for (int el = tid - 256; el < 1024; el += PF_THREADS - 256) {
            const int i = el & 31, cc = el >> 5;
            double acc = 0.0;
            if (cc >= i)
                for (int k = i; k <= cc; ++k) acc -= L.Us[i][k] * L.Ss[k] * L.Ls[cc][k];
            L.Ts[i][cc] = acc;
            if (g == f.gown) {
                P.T[(size_t) (c + cc) * P.ldt + c + i] = acc;
                if (i == cc) P.tau[c + i] = acc;
                double r = L.Bs[i][cc];                      // strictly lower: L1
                if (cc >= i) {
                    r = 0.0;
                    for (int k = i; k <= cc; ++k) r += L.R2s[i][k] * L.R1s[k][cc];
                    r *= L.Ss[i];
                }
                P.A[(size_t) (c + cc) * P.lda + c + i] = r;
                P.Vw[(size_t) (c + cc) * P.ldv + c + i] = (cc < i) ? L.Bs[i][cc] : (cc == i ? 1.0 : 0.0);
            }
        }
        __syncthreads();                                                             // #14
        __syncthreads();                                                             // #15
        pf_fold(P, f, L, g, nwg, wave, lane);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();                                                             // #16
        __syncthreads();                                                             // #17
        __syncthreads();                                                             // #18
    }
}

__global__ __launch_bounds__(PF_THREADS) void panel_fused_kernel(PfArgs P)
{
    extern __shared__ __attribute__((aligned(16))) double sm[];
    // whole waves take one role or the other: the barriers in the two bodies pair up one to one
    if (__builtin_amdgcn_readfirstlane((int) (threadIdx.x >> 6)) < 4) pf_rows(P, sm, blockIdx.x, gridDim.x);
    else pf_service(P, sm, blockIdx.x, gridDim.x);
}

extern "C" {

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
    a.ws = ws; a.epoch0 = *epoch; a.status = status;
    *epoch += 1024u;
    const int nwg = (mk + PF_ROWS - 1) / PF_ROWS;
    hipLaunchKernelGGL(panel_fused_kernel, dim3(nwg), dim3(PF_THREADS), PF_SM_DOUBLES * sizeof(double), (hipStream_t) stream, a);
    return (int) hipGetLastError();
}

}   // extern "C"
