// qr_panel_cqr.hip -- a TALL panel (mk >> w, w <= 128 columns) factored at its full width by CholeskyQR2 + Householder reconstruction.
//
// The 32-column leaf chain (qr_panel_tsqr.hip) passes ~12 times over a tall panel: every leaf reads and writes its own columns 6 times
// and, after it, the rest of the panel 3 times (in-panel product + update).  At full width the same mathematics needs six passes:
//     G1 = A^T A                      (gemm_tn, read A)
//     R1 = chol(G1), R1^-1            (one workgroup, cqr_chol_kernel)
//     Q  = A R1^-1                    (cqr_rows_kernel<0>: read A, write Q into Vw)
//     G2 = Q^T Q                      (gemm_tn, read Q)
//     R2 = chol(G2) [first order when |G2 - I| <= 1e-9], modified LU  Q_top - S R2 = L1 U',  U'^-1,
//     T = -U' R2^-1 S L1^-T,  R = S R2 R1            (one workgroup, cqr_lu_kernel)
//     V  = (Q - [S R2; 0]) U'^-1     (cqr_rows_kernel<1>: read Q, write V into Vw and below the diagonal of A)
// and T of the whole panel comes out of the reconstruction (no V^T V Gram pass, no merge of leaf T blocks).  It is the panel-level
// form of what the leaves do (reference qr.c:109-235 factors a panel column by column; the factors V, tau, R it returns are the
// Householder ones, which the reconstruction reproduces: Ballard et al., "Reconstructing Householder vectors from TSQR").
// The guard is the leaves' guard: a failed Cholesky or |G2 - I| > 1/64 sets status[0] = 1, nothing of A has been touched by then
// (Q lives in Vw) and the last two kernels return at once; the host sees the flag and runs the leaf chain on the untouched panel.
#include <hip/hip_runtime.h>
#include "qr_device.h"
#include "qr_common.h"
#include "qr_leaf_math.h"

namespace {
typedef double v4d __attribute__((ext_vector_type(4)));

constexpr int CQ_W = 128;                 // widest panel
constexpr int CQ_LD = 129;                // row stride of the LDS matrix
constexpr int CQ_T = 256;                 // threads of the one-workgroup kernels: a 16 x 16 grid, 8 x 8 elements each (one wave per SIMD:
                                          // sixteen waves of 4 x 4 elements spent 1 us per elimination step on predicates and selects)
constexpr int CQ_E = 8;                   // elements per thread and dimension
// workspace (doubles), all matrices CQ_W x CQ_W
constexpr int CQ_G1 = 0;                  // Gram matrices, column-major ld CQ_W (gemm_tn output)
constexpr int CQ_G2 = 1 * CQ_W * CQ_W;
constexpr int CQ_R1 = 2 * CQ_W * CQ_W;    // R1, row-major [k][j]
constexpr int CQ_R1I = 3 * CQ_W * CQ_W;   // R1^-1, row-major
constexpr int CQ_UI = 4 * CQ_W * CQ_W;    // U'^-1, row-major
constexpr int CQ_R2 = 5 * CQ_W * CQ_W;    // R2 row-major
constexpr int CQ_LU = 6 * CQ_W * CQ_W;    // L1 \ U' row-major
constexpr int CQ_X1 = 7 * CQ_W * CQ_W;    // scratch operands, row-major
constexpr int CQ_X2 = 8 * CQ_W * CQ_W;
constexpr int CQ_RR = 9 * CQ_W * CQ_W;    // R = S R2 R1 row-major
constexpr int CQ_TT = 10 * CQ_W * CQ_W;   // T row-major
constexpr int CQ_SV = 11 * CQ_W * CQ_W;   // S (CQ_W doubles)
constexpr int CQ_ST = 11 * CQ_W * CQ_W + CQ_W;   // 64 phase stamps of the one-workgroup kernels (CQ_STAMPS builds)
constexpr int CQ_WS = CQ_ST + 64;

__device__ __forceinline__ double cq_ld(const double* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void cq_st(double* p, double v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

#ifdef CQ_STAMPS
#define CQ_STAMP(n) do { if (threadIdx.x == 0) reinterpret_cast<unsigned long long*>(ws + CQ_ST)[n] = wall_clock64(); } while (0)
#else
#define CQ_STAMP(n) do { } while (0)
#endif
// barrier after which this workgroup's own global (workspace) stores can be read back by any of its threads
__device__ __forceinline__ void cq_sync_global() { __threadfence(); __syncthreads(); }

// LDS of the one-workgroup kernels
struct CqLds {
    double* M;            // [CQ_LD][CQ_LD]: an upper-triangular matrix on and above the diagonal, its inverse X transposed strictly
                          // below it (X(i, j) at M[j + 1][i])
    double* row;          // [2][CQ_W] pivot row of the current / next elimination step
    double* col;          // [2][CQ_W] pivot column
    double* sv;           // [CQ_W] signs
    double* sb;           // [32][33] block scratch of the inverse
    double* red;          // [CQ_T / 64] reduction scratch
    int* flag;            // [4]
};
__device__ __forceinline__ CqLds cq_lds(double* sm)
{
    CqLds L;
    L.M = sm;
    L.row = L.M + CQ_LD * CQ_LD;
    L.col = L.row + 2 * CQ_W;
    L.sv = L.col + 2 * CQ_W;
    L.sb = L.sv + CQ_W;
    L.red = L.sb + 32 * 33;
    L.flag = reinterpret_cast<int*>(L.red + CQ_T / 64);
    return L;
}
constexpr size_t CQ_LDS_BYTES = sizeof(double) * (CQ_LD * CQ_LD + 4 * CQ_W + CQ_W + 32 * 33 + CQ_T / 64) + 64;

// thread (ti, tj) of the 16 x 16 grid owns elements (ti + 16 a, tj + 16 b), a, b = 0 .. 7
#define CQ_FOR_TILE for (int a = 0; a < CQ_E; ++a) for (int b = 0; b < CQ_E; ++b)

// right-looking Cholesky of the symmetric matrix whose upper triangle sits in the register tiles: R (upper) -> L.M, row by row.
// One barrier per column: the pivot row of step k + 1 is published by its owners at the end of step k.  Blocks of 16 x 16 elements
// (one per thread and (a, b)) that lie entirely above the pivot row or below the diagonal are skipped by uniform branches.
__device__ __forceinline__ bool cq_chol(double (&t)[CQ_E][CQ_E], const CqLds& L, int w, int ti, int tj, int tid)
{
    bool ok = true;
    const int nbk = w >> 4;
    if (ti == 0)
#pragma unroll
        for (int b = 0; b < CQ_E; ++b) L.row[tj + 16 * b] = t[0][b];
    for (int k = 0; k < w; ++k) {
        __syncthreads();
        const double* rk = L.row + (k & 1) * CQ_W;
        const double p = rk[k];
        ok = ok && (p > 0.0);                                 // false for NaN as well
        const double inv = rcp_newton(p);
        if (tid < w) L.M[k * CQ_LD + tid] = (tid >= k) ? rk[tid] * rsqrt_newton(p) : 0.0;
        const int a0 = k >> 4;                                // blocks a < a0 lie above the pivot row
        double rj[CQ_E];
#pragma unroll
        for (int b = 0; b < CQ_E; ++b) rj[b] = (b >= a0 && b < nbk) ? rk[tj + 16 * b] : 0.0;
#pragma unroll
        for (int a = 0; a < CQ_E; ++a) {
            if (a >= a0 && a < nbk) {
                const int i = ti + 16 * a;
                const double li = rk[i] * inv;
#pragma unroll
                for (int b = 0; b < CQ_E; ++b) {
                    if (b >= a && b < nbk) {
                        const int j = tj + 16 * b;
                        if (i > k && j >= i) t[a][b] -= li * rj[b];
                    }
                }
            }
        }
        const int kn = k + 1;
        if (kn < w && ti == (kn & 15)) {
            double* rn = L.row + (kn & 1) * CQ_W;
#pragma unroll
            for (int a = 0; a < CQ_E; ++a)
                if (a == (kn >> 4))
#pragma unroll
                    for (int b = 0; b < CQ_E; ++b) rn[tj + 16 * b] = t[a][b];
        }
    }
    __syncthreads();
    return ok;
}

// one 16 x 16 tile of a product on the matrix cores: sum over k0 <= k < k1 (multiples of 4) of a(i0 + p, k) b(k, j0 + q).
// Accumulator register r of a lane: row i0 + (lane >> 4) + 4 r, column j0 + (lane & 15)
template <class FA, class FB>
__device__ __forceinline__ v4d cq_tile(FA a, FB b, int i0, int j0, int k0, int k1, int lane)
{
    const int l15 = lane & 15, l4 = lane >> 4;
    v4d acc = (v4d){0.0, 0.0, 0.0, 0.0};
    for (int k = k0; k < k1; k += 4) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a(i0 + l15, k + l4), b(k + l4, j0 + l15), acc, 0, 0, 0);
    return acc;
}

// inverse of the upper-triangular matrix in L.M (rows / columns < w, w a multiple of 32); the off-diagonal blocks of the matrix are
// DESTROYED.  Diagonal 32 x 32 blocks by back substitution (one wave each, a column per lane); then, as for a 2 x 2 block matrix,
// X12 = -X11 (R12 X22) first inside each half of 64 columns and then between the halves, the products on the matrix cores with the
// intermediate R12 X22 parked in R12's place.  X(i, j) goes to L.M[j + 1][i] (strictly below the diagonal: stride 129 keeps a
// column of X on distinct banks)
__device__ __forceinline__ void cq_offdiag(const CqLds& L, int r0, int nr, int c0, int nc, int tid)
{
    // X(r0 : r0 + nr, c0 : c0 + nc) = -X(r0 .., r0 ..) (R(r0 .., c0 ..) X(c0 .., c0 ..)), nr, nc multiples of 16, c0 = r0 + nr
    const int wave = tid >> 6, lane = tid & 63, l15 = lane & 15, l4 = lane >> 4;
    const int ntc = nc >> 4, nt = (nr >> 4) * ntc;           // at most 16 tiles: up to four per wave
    auto R = [&](int i, int k) { return L.M[i * CQ_LD + k]; };
    auto X = [&](int k, int j) { return (k <= j) ? L.M[(j + 1) * CQ_LD + k] : 0.0; };
    v4d acc[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int tl = wave + 4 * q;
        if (tl < nt) acc[q] = cq_tile(R, X, r0 + 16 * (tl / ntc), c0 + 16 * (tl % ntc), c0, c0 + 16 * (tl % ntc) + 16, lane);   // X(c0 .., c0 ..) upper: k <= j
    }
    __syncthreads();
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int tl = wave + 4 * q, i0 = r0 + 16 * (tl / ntc), j0 = c0 + 16 * (tl % ntc);
        if (tl < nt)
#pragma unroll
            for (int r = 0; r < 4; ++r) L.M[(i0 + l4 + 4 * r) * CQ_LD + j0 + l15] = acc[q][r];
    }
    __syncthreads();
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int tl = wave + 4 * q, i0 = r0 + 16 * (tl / ntc), j0 = c0 + 16 * (tl % ntc);
        if (tl < nt) acc[q] = cq_tile(X, R, i0, j0, i0, r0 + nr, lane);                  // X(r0 .., r0 ..) upper: k >= i
    }
    __syncthreads();
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int tl = wave + 4 * q, i0 = r0 + 16 * (tl / ntc), j0 = c0 + 16 * (tl % ntc);
        if (tl < nt)
#pragma unroll
            for (int r = 0; r < 4; ++r) L.M[(j0 + l15 + 1) * CQ_LD + i0 + l4 + 4 * r] = -acc[q][r];
    }
    __syncthreads();
}
__device__ __forceinline__ void cq_upper_inv(const CqLds& L, int w, int ti, int tj, int tid)
{
    (void) ti; (void) tj;
    const int nblk = w >> 5, wave = tid >> 6, lane = tid & 63;
    // the strictly lower part must read as zero where X has not been written
    for (int e = tid; e < (w + 1) * w; e += CQ_T) {
        const int r = e / w, c = e - r * w;
        if (c < r) L.M[r * CQ_LD + c] = 0.0;
    }
    __syncthreads();
    if (wave < nblk && lane < 32) {
        const int o = 32 * wave, j = lane;
        double* xj = L.M + (o + j + 1) * CQ_LD + o;          // x(i) = X(o + i, o + j)
        for (int i = 31; i >= 0; --i) {
            double acc = (i == j) ? 1.0 : 0.0;
            for (int k = i + 1; k < 32; ++k) acc -= L.M[(o + i) * CQ_LD + o + k] * ((k <= j) ? xj[k] : 0.0);
            const double x = acc * rcp_newton(L.M[(o + i) * CQ_LD + o + i]);
            if (i <= j) xj[i] = x;
        }
    }
    __syncthreads();
    if (w >= 64) cq_offdiag(L, 0, 32, 32, 32, tid);
    if (w == 128) cq_offdiag(L, 64, 32, 96, 32, tid);
    if (w > 64) cq_offdiag(L, 0, 64, 64, w - 64, tid);
}

// the inverse out of L.M into a row-major global matrix (zero below the diagonal)
__device__ __forceinline__ void cq_inv_out(const CqLds& L, double* X, int w, int tid)
{
    for (int e = tid; e < w * w; e += CQ_T) {
        const int i = e / w, j = e - i * w;
        cq_st(X + i * CQ_W + j, (j >= i) ? L.M[(j + 1) * CQ_LD + i] : 0.0);
    }
}

// C = A B for upper-triangular A (in L.M, on and above the diagonal) and upper-triangular B (row-major, global), on the matrix cores:
// wave v owns tile rows v and v + 4 and all eight tile columns; c[q] (accumulator layout of cq_tile) is tile (v + 4 (q >> 3), q & 7)
__device__ __forceinline__ void cq_upper_product(v4d (&c)[16], const CqLds& L, const double* B, int w, int tid)
{
    const int wave = tid >> 6, lane = tid & 63;
    auto A = [&](int i, int k) { return (k >= i) ? L.M[i * CQ_LD + k] : 0.0; };
    auto Bg = [&](int k, int j) { return cq_ld(B + k * CQ_W + j); };
#pragma unroll
    for (int q = 0; q < 16; ++q) {
        const int tr = wave + 4 * (q >> 3), tc = q & 7;
        c[q] = (v4d){0.0, 0.0, 0.0, 0.0};
        if (tc >= tr && 16 * tc < w && 16 * tr < w) c[q] = cq_tile(A, Bg, 16 * tr, 16 * tc, 16 * tr, 16 * tc + 16, lane);
    }
}
// visit the elements of the wave's product tiles: f(i, j, value)
template <class F>
__device__ __forceinline__ void cq_product_visit(const v4d (&c)[16], int w, int tid, F f)
{
    const int wave = tid >> 6, lane = tid & 63, l15 = lane & 15, l4 = lane >> 4;
#pragma unroll
    for (int q = 0; q < 16; ++q) {
        const int tr = wave + 4 * (q >> 3), tc = q & 7;
        if (16 * tc < w && 16 * tr < w)
#pragma unroll
            for (int r = 0; r < 4; ++r) f(16 * tr + l4 + 4 * r, 16 * tc + l15, c[q][r]);
    }
}

__device__ __forceinline__ void cq_tile_to_lds_upper(const double (&t)[CQ_E][CQ_E], const CqLds& L, int w, int ti, int tj)
{
#pragma unroll
    CQ_FOR_TILE {
        const int i = ti + 16 * a, j = tj + 16 * b;
        if (i < w && j < w && j >= i) L.M[i * CQ_LD + j] = t[a][b];
    }
}

// ---------------------------------------------------------------------------------------------------------------------------------
// R1 = chol(G1), R1^-1.  G: column-major ld CQ_W (upper triangle read).  status[0] |= 1 on a non-positive pivot.
// ---------------------------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(CQ_T) void cqr_chol_kernel(double* ws, int w, int* status)
{
    extern __shared__ double sm[];
    const CqLds L = cq_lds(sm);
    const int tid = threadIdx.x, tj = tid & 15, ti = tid >> 4;
    double t[CQ_E][CQ_E];
#pragma unroll
    CQ_FOR_TILE {
        const int i = ti + 16 * a, j = tj + 16 * b;
        t[a][b] = (i < w && j < w && j >= i) ? ws[CQ_G1 + i + CQ_W * j] : 0.0;
    }
    CQ_STAMP(0);
    const bool ok = cq_chol(t, L, w, ti, tj, tid);
    CQ_STAMP(1);
    if (!ok) { if (tid == 0) status[0] = 1; return; }         // uniform: every thread read the same pivots
    for (int e = tid; e < w * w; e += CQ_T) {
        const int i = e / w, j = e - i * w;
        cq_st(ws + CQ_R1 + i * CQ_W + j, (j >= i) ? L.M[i * CQ_LD + j] : 0.0);
    }
    CQ_STAMP(2);
    cq_upper_inv(L, w, ti, tj, tid);
    CQ_STAMP(3);
    cq_inv_out(L, ws + CQ_R1I, w, tid);
    CQ_STAMP(4);
}

// ---------------------------------------------------------------------------------------------------------------------------------
// Everything between the two streaming passes, on one workgroup (see the header).  Q_top: the first w rows of Q in Vw.
// Outputs: Vw top block <- Q_top - S R2; ws: U'^-1, L1 \ U', R, T, S; status[0] |= 1 when the panel is refused.
// ---------------------------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(CQ_T) void cqr_lu_kernel(double* ws, int w, double* Vw, int ldv, int* status)
{
    extern __shared__ double sm[];
    const CqLds L = cq_lds(sm);
    const int tid = threadIdx.x, tj = tid & 15, ti = tid >> 4;
    if (status[0]) return;                                    // the first Cholesky failed
    double t[CQ_E][CQ_E];
    CQ_STAMP(8);
    // ---- G2: its distance from I decides between the first-order factor, the Cholesky and the refusal
    double dmax = 0.0;
    bool nan = false;
#pragma unroll
    CQ_FOR_TILE {
        const int i = ti + 16 * a, j = tj + 16 * b;
        const bool in = i < w && j < w && j >= i;
        t[a][b] = in ? ws[CQ_G2 + i + CQ_W * j] : 0.0;
        const double d = in ? fabs(t[a][b] - (i == j ? 1.0 : 0.0)) : 0.0;
        nan = nan || !(d == d);
        dmax = fmax(dmax, d);
    }
    if (nan) dmax = 1e300;
    for (int o = 32; o > 0; o >>= 1) dmax = fmax(dmax, __shfl_xor(dmax, o));
    if ((tid & 63) == 0) L.red[tid >> 6] = dmax;
    __syncthreads();
    dmax = 0.0;
    for (int q = 0; q < CQ_T / 64; ++q) dmax = fmax(dmax, L.red[q]);
    if (!(dmax <= QRD_GUARD_THR)) { if (tid == 0) status[0] = 1; return; }
    const bool first_order = dmax <= QRD_CHOL1_THR;
    if (first_order) {
        // G2 = I + E, |E| <= 1e-9: R2 = I + triu(E, 1) + diag(E) / 2 to ~1e-18
        for (int e = tid; e < w * w; e += CQ_T) L.M[(e / w) * CQ_LD + (e % w)] = 0.0;
        __syncthreads();
#pragma unroll
        CQ_FOR_TILE {
            const int i = ti + 16 * a, j = tj + 16 * b;
            if (i < w && j < w && j >= i) L.M[i * CQ_LD + j] = (i == j) ? 1.0 + 0.5 * (t[a][b] - 1.0) : t[a][b];
        }
        __syncthreads();
    } else {
        const bool ok = cq_chol(t, L, w, ti, tj, tid);
        if (!ok) { if (tid == 0) status[0] = 1; return; }
    }
    for (int e = tid; e < w * w; e += CQ_T) {
        const int i = e / w, j = e - i * w;
        cq_st(ws + CQ_R2 + i * CQ_W + j, (j >= i) ? L.M[i * CQ_LD + j] : 0.0);
    }
    CQ_STAMP(9);
    // ---- modified LU of Q_top - S R2 (R2 in L.M), the sign of every pivot chosen as Householder would (reference qr.c:141-151)
#pragma unroll
    CQ_FOR_TILE {
        const int i = ti + 16 * a, j = tj + 16 * b;
        t[a][b] = (i < w && j < w) ? Vw[i + (size_t) ldv * j] : 0.0;
    }
    if (ti == 0)
#pragma unroll
        for (int b = 0; b < CQ_E; ++b) L.row[tj + 16 * b] = t[0][b];
    if (tj == 0)
#pragma unroll
        for (int a = 0; a < CQ_E; ++a) L.col[ti + 16 * a] = t[a][0];
    const int nbk = w >> 4;
    for (int I = 0; I < w; ++I) {
        __syncthreads();
        const double* rI = L.row + (I & 1) * CQ_W;
        const double* cI = L.col + (I & 1) * CQ_W;
        const double x = rI[I];
        const double S = (x >= 0.0) ? -1.0 : 1.0;
        const double piv = x - S * L.M[I * CQ_LD + I];        // |piv| >= R2(I, I) > 0
        const double inv = rcp_newton(piv);
        if (tid == 0) L.sv[I] = S;
        const int a0 = I >> 4;                                // blocks a < a0 (b < a0) lie above (left of) the pivot
        double uj[CQ_E];
#pragma unroll
        for (int b = 0; b < CQ_E; ++b) {
            const int j = tj + 16 * b;
            uj[b] = (b >= a0 && b < nbk && j >= I) ? rI[j] - S * L.M[I * CQ_LD + j] : 0.0;
        }
#pragma unroll
        for (int a = 0; a < CQ_E; ++a) {
            if (a >= a0 && a < nbk) {
                const int i = ti + 16 * a;
                const double li = cI[i] * inv;
#pragma unroll
                for (int b = 0; b < CQ_E; ++b) {
                    if (b >= a0 && b < nbk) {
                        const int j = tj + 16 * b;
                        if (i == I) { if (j >= I) t[a][b] = uj[b]; }
                        else if (i > I) {
                            if (j == I) t[a][b] = li;
                            else if (j > I) t[a][b] -= li * uj[b];
                        }
                    }
                }
            }
        }
        const int In = I + 1;
        if (In < w) {
            if (ti == (In & 15)) {
                double* rn = L.row + (In & 1) * CQ_W;
#pragma unroll
                for (int a = 0; a < CQ_E; ++a)
                    if (a == (In >> 4))
#pragma unroll
                        for (int b = 0; b < CQ_E; ++b) rn[tj + 16 * b] = t[a][b];
            }
            if (tj == (In & 15)) {
                double* cn = L.col + (In & 1) * CQ_W;
#pragma unroll
                for (int b = 0; b < CQ_E; ++b)
                    if (b == (In >> 4))
#pragma unroll
                        for (int a = 0; a < CQ_E; ++a) cn[ti + 16 * a] = t[a][b];
            }
        }
    }
    __syncthreads();
    CQ_STAMP(10);
    // L1 \ U' and S out; Q_top - S R2 back into Vw (the last pass multiplies it by U'^-1 like every other row: it becomes L1)
#pragma unroll
    CQ_FOR_TILE {
        const int i = ti + 16 * a, j = tj + 16 * b;
        if (i < w && j < w) {
            cq_st(ws + CQ_LU + i * CQ_W + j, t[a][b]);
            if (j >= i) Vw[i + (size_t) ldv * j] -= L.sv[i] * L.M[i * CQ_LD + j];
        }
    }
    if (tid < w) cq_st(ws + CQ_SV + tid, L.sv[tid]);
    CQ_STAMP(11);
    // ---- R = S R2 R1 (R2 still in L.M)
    v4d c[16];
    cq_upper_product(c, L, ws + CQ_R1, w, tid);
    cq_product_visit(c, w, tid, [&](int i, int j, double v) { cq_st(ws + CQ_RR + i * CQ_W + j, (j >= i) ? L.sv[i] * v : 0.0); });
    CQ_STAMP(12);
    // ---- R2^-1 -> X1 (first order: 2 I - R2)
    if (first_order) {
        for (int e = tid; e < w * w; e += CQ_T) {
            const int i = e / w, j = e - i * w;
            cq_st(ws + CQ_X1 + i * CQ_W + j, (j > i) ? -L.M[i * CQ_LD + j] : (j == i ? 2.0 - L.M[i * CQ_LD + j] : 0.0));
        }
        cq_sync_global();
    } else {
        __syncthreads();                                      // (the product above still reads R2's off-diagonal blocks)
        cq_upper_inv(L, w, ti, tj, tid);
        cq_inv_out(L, ws + CQ_X1, w, tid);
        cq_sync_global();
    }
    CQ_STAMP(13);
    // ---- U' -> L.M; U = U' R2^-1 -> X2; U'^-1 -> UI
    cq_tile_to_lds_upper(t, L, w, ti, tj);
    __syncthreads();
    cq_upper_product(c, L, ws + CQ_X1, w, tid);
    cq_product_visit(c, w, tid, [&](int i, int j, double v) { cq_st(ws + CQ_X2 + i * CQ_W + j, (j >= i) ? v : 0.0); });
    __syncthreads();
    CQ_STAMP(14);
    cq_upper_inv(L, w, ti, tj, tid);
    CQ_STAMP(15);
    cq_inv_out(L, ws + CQ_UI, w, tid);
    __syncthreads();
    CQ_STAMP(16);
    // ---- L1^-T: the inverse of the unit upper-triangular L1^T, rows scaled by S on the way out: X1 = S L1^-T
#pragma unroll
    CQ_FOR_TILE {
        const int i = ti + 16 * a, j = tj + 16 * b;
        if (i < w && j < w && i >= j) L.M[j * CQ_LD + i] = (i == j) ? 1.0 : t[a][b];         // (L1^T)(j, i) = L1(i, j)
    }
    __syncthreads();
    cq_upper_inv(L, w, ti, tj, tid);
    for (int e = tid; e < w * w; e += CQ_T) {
        const int i = e / w, j = e - i * w;
        cq_st(ws + CQ_X1 + i * CQ_W + j, (j >= i) ? L.sv[i] * L.M[(j + 1) * CQ_LD + i] : 0.0);
    }
    cq_sync_global();
    CQ_STAMP(17);
    // ---- T = -U (S L1^-T): U (X2) -> L.M
    for (int e = tid; e < w * w; e += CQ_T) {
        const int i = e / w, j = e - i * w;
        if (j >= i) L.M[i * CQ_LD + j] = cq_ld(ws + CQ_X2 + i * CQ_W + j);
    }
    __syncthreads();
    cq_upper_product(c, L, ws + CQ_X1, w, tid);
    cq_product_visit(c, w, tid, [&](int i, int j, double v) { cq_st(ws + CQ_TT + i * CQ_W + j, (j >= i) ? -v : 0.0); });
    CQ_STAMP(18);
}

// ---------------------------------------------------------------------------------------------------------------------------------
// The streaming passes: dst rows = src rows times an upper-triangular w x w matrix X (row-major in ws).  Matrix cores with the
// product transposed -- D(col, row) = sum_k X(k, col) src(row, k) -- so that the accumulator registers of a lane are four columns of
// 16 CONSECUTIVE ROWS: loads and stores are both whole 128-byte lines of a column.  A wave takes 16 rows at a time: the 16 x w row
// block sits in 32 operand registers, column tile jt needs only k < 16 (jt + 1).
//   MODE 0: Q = A R1^-1 (src A, dst Vw).    MODE 1: V = Q' U'^-1 (src Vw, dst Vw and A); returns at once when status[0] is set.
// ---------------------------------------------------------------------------------------------------------------------------------
constexpr int CR_THREADS = 256;
template <int MODE>
__global__ __launch_bounds__(CR_THREADS) void cqr_rows_kernel(const double* __restrict__ X, int w, int mk, const double* src, int lds_,
                                                                double* dst, int ldd, double* dst2, int ldd2, const int* status)
{
    extern __shared__ double sm[];                           // X^T? no: X row-major [k][j], stride CQ_LD
    if (MODE == 1 && status[0]) return;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l15 = lane & 15, l4 = lane >> 4;
    for (int e = tid; e < w * w; e += CR_THREADS) {
        const int k = e / w, j = e - k * w;
        sm[k * CQ_LD + j] = X[k * CQ_W + j];
    }
    __syncthreads();
    const int ntile = (mk + 15) >> 4, nct = w >> 4;
    for (int tile = blockIdx.x * (CR_THREADS / 64) + wave; tile < ntile; tile += gridDim.x * (CR_THREADS / 64)) {
        const int r0 = 16 * tile, row = r0 + l15;
        const bool rin = row < mk;
        const double* sp = src + (rin ? row : mk - 1);
        double q[32];
#pragma unroll
        for (int ks = 0; ks < 32; ++ks) q[ks] = (4 * ks + l4 < w) ? sp[(size_t) (4 * ks + l4) * lds_] : 0.0;
        if (!rin) {
#pragma unroll
            for (int ks = 0; ks < 32; ++ks) q[ks] = 0.0;
        }
#pragma unroll
        for (int jt = 0; jt < 8; ++jt) {
            if (jt < nct) {
                v4d acc = (v4d){0.0, 0.0, 0.0, 0.0};
#pragma unroll
                for (int ks = 0; ks < 4 * (jt + 1); ++ks)
                    acc = __builtin_amdgcn_mfma_f64_16x16x4f64(sm[(4 * ks + l4) * CQ_LD + 16 * jt + l15], q[ks], acc, 0, 0, 0);
                if (rin) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int colj = 16 * jt + l4 + 4 * r;
                        dst[row + (size_t) colj * ldd] = acc[r];
                        if (MODE == 1) dst2[row + (size_t) colj * ldd2] = acc[r];
                    }
                }
            }
        }
    }
}

// the top block after the last pass: A <- R on and above the diagonal, L1 below; Vw <- unit lower L1; T, tau
__global__ __launch_bounds__(256) void cqr_top_kernel(const double* ws, int w, double* A, int lda, double* Vw, int ldv, double* T, int ldt,
                                                       double* tau, const int* status)
{
    if (status[0]) return;
    for (int e = threadIdx.x + blockIdx.x * blockDim.x; e < w * w; e += blockDim.x * gridDim.x) {
        const int j = e / w, i = e - j * w;                   // consecutive threads: consecutive rows of a column
        const double lu = ws[CQ_LU + i * CQ_W + j];
        A[i + (size_t) lda * j] = (j >= i) ? ws[CQ_RR + i * CQ_W + j] : lu;
        Vw[i + (size_t) ldv * j] = (j < i) ? lu : (j == i ? 1.0 : 0.0);
        const double tv = ws[CQ_TT + i * CQ_W + j];
        T[i + (size_t) ldt * j] = tv;
        if (i == j) tau[i] = tv;
    }
}
}   // namespace

extern "C" {

size_t qrd_panel_cqr_ws_doubles(void) { return (size_t) CQ_WS; }

int qrd_panel_cqr_init(void)
{
    hipError_t e = hipFuncSetAttribute((const void*) cqr_chol_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int) CQ_LDS_BYTES);
    if (e == hipSuccess) e = hipFuncSetAttribute((const void*) cqr_lu_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int) CQ_LDS_BYTES);
    const int rows_lds = (int) (sizeof(double) * CQ_W * CQ_LD);
    if (e == hipSuccess) e = hipFuncSetAttribute((const void*) cqr_rows_kernel<0>, hipFuncAttributeMaxDynamicSharedMemorySize, rows_lds);
    if (e == hipSuccess) e = hipFuncSetAttribute((const void*) cqr_rows_kernel<1>, hipFuncAttributeMaxDynamicSharedMemorySize, rows_lds);
    return (int) e;
}

// shapes this route takes: w a multiple of 32 up to 128, at least 2 w rows
int qrd_panel_cqr_ok(int mk, int w) { return w >= 32 && w <= CQ_W && (w & 31) == 0 && mk >= 2 * w; }

// stage 1 (after G1 = A^T A has been put into ws + CQ_G1, column-major ld 128): R1, R1^-1, Q = A R1^-1 -> Vw
int qrd_panel_cqr_stage1(void* stream, const double* A, int lda, int mk, int w, double* Vw, int ldv, double* ws, int* status)
{
    hipStream_t s = (hipStream_t) stream;
    hipLaunchKernelGGL(cqr_chol_kernel, dim3(1), dim3(CQ_T), CQ_LDS_BYTES, s, ws, w, status);
    const int ntile = (mk + 15) / 16, grid = ((ntile + 3) / 4 < 256) ? (ntile + 3) / 4 : 256;
    hipLaunchKernelGGL(cqr_rows_kernel<0>, dim3(grid), dim3(CR_THREADS), sizeof(double) * CQ_W * CQ_LD, s, ws + CQ_R1I, w, mk, A, lda, Vw, ldv,
                       (double*) nullptr, 0, status);
    return (int) hipGetLastError();
}

// stage 2 (after G2 = Q^T Q has been put into ws + CQ_G2): the small factors, V -> Vw and A, the top block, T, tau
int qrd_panel_cqr_stage2(void* stream, double* A, int lda, int mk, int w, double* tau, double* T, int ldt, double* Vw, int ldv, double* ws,
                         int* status)
{
    hipStream_t s = (hipStream_t) stream;
    hipLaunchKernelGGL(cqr_lu_kernel, dim3(1), dim3(CQ_T), CQ_LDS_BYTES, s, ws, w, Vw, ldv, status);
    const int ntile = (mk + 15) / 16, grid = ((ntile + 3) / 4 < 256) ? (ntile + 3) / 4 : 256;
    hipLaunchKernelGGL(cqr_rows_kernel<1>, dim3(grid), dim3(CR_THREADS), sizeof(double) * CQ_W * CQ_LD, s, ws + CQ_UI, w, mk, Vw, ldv, Vw, ldv, A, lda,
                       status);
    hipLaunchKernelGGL(cqr_top_kernel, dim3((w * w + 255) / 256), dim3(256), 0, s, ws, w, A, lda, Vw, ldv, T, ldt, tau, status);
    return (int) hipGetLastError();
}

double* qrd_panel_cqr_g1(double* ws) { return ws + CQ_G1; }
double* qrd_panel_cqr_g2(double* ws) { return ws + CQ_G2; }

}   // extern "C"
