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
#include "qr_factor32.h"

namespace {
typedef double v4d __attribute__((ext_vector_type(4)));
// The wave index is taken through readfirstlane everywhere below: as `tid >> 6` it is a VECTOR value to the compiler, every
// "this wave owns that tile" test became an exec-mask block of its own and each matrix-core instruction sat behind its private
// `ds_read; s_waitcnt lgkmcnt(0)` -- the Gram pass ran at 11.6 us per 64-row block where the instructions account for 4


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
constexpr int CQ_X3 = 11 * CQ_W * CQ_W;
constexpr int CQ_SV = 12 * CQ_W * CQ_W;   // S (CQ_W doubles)
constexpr int CQ_ST = 12 * CQ_W * CQ_W + CQ_W;   // 64 phase stamps of the one-workgroup kernels (CQ_STAMPS builds)
constexpr int CQ_FO = CQ_ST + 63;                  // 1.0: R2 of this panel is the first-order factor (its inverse is 2 I - R2: nobody stores it)
constexpr int CQ_SL = CQ_ST + 64;                  // Gram partials of the streaming passes: 256 workgroups x 36 tiles x 256 doubles (19 MB)
constexpr int CQ_R0 = CQ_SL + 256 * 36 * 256;    // R0 = chol(G1 + s I), row-major: the preconditioner of a retried panel (qrd_panel_cqr_retry)
constexpr int CQ_WS = CQ_R0 + CQ_W * CQ_W;

// Workspace traffic of the one-workgroup kernels: plain stores and loads.  Every workspace matrix is written ONCE per launch and read
// only after cq_sync_global() (so no line of it can be in this compute unit's cache before it is written); agent-scope atomic stores
// were measured at ~0.3 us EACH here (the compiler waits for every one of them: 64 per thread = 20 us per matrix written)
__device__ __forceinline__ void cq_st(double* p, double v) { *p = v; }

#ifdef CQ_STAMPS
#define CQ_STAMP(n) do { if (threadIdx.x == 0) reinterpret_cast<unsigned long long*>(ws + CQ_ST)[n] = wall_clock64(); } while (0)
#define CQ_STAMP_L(n) do { if (threadIdx.x == 0 && L.wsdbg) reinterpret_cast<unsigned long long*>(L.wsdbg + CQ_ST)[n] = wall_clock64(); } while (0)
#else
#define CQ_STAMP(n) do { } while (0)
#define CQ_STAMP_L(n) do { } while (0)
#endif
// barrier after which this workgroup's own global (workspace) stores can be read back by any of its threads
__device__ __forceinline__ void cq_sync_global() { __threadfence(); __syncthreads(); }

// Elementwise pass over the w x 128 index space (e -> (e >> 7, e & 127)) with the loads of 32 elements per thread requested together:
// a rolled loop with a dependent global load per iteration costs the memory latency (~0.4 us) EVERY iteration -- 64 iterations = 25 us
template <class FL, class FS>
__device__ __forceinline__ void cq_elems(int w, int tid, FL load, FS store)
{
    const int n = w * CQ_W;
    for (int base = tid; base < n; base += 32 * CQ_T) {
        double v[32];
#pragma unroll
        for (int u = 0; u < 32; ++u) { const int e = base + u * CQ_T; v[u] = load((e < n ? e : tid) >> 7, (e < n ? e : tid) & (CQ_W - 1)); }
#pragma unroll
        for (int u = 0; u < 32; ++u) { const int e = base + u * CQ_T; if (e < n) store(e >> 7, e & (CQ_W - 1), v[u]); }
    }
}

// LDS of the one-workgroup kernels
struct CqLds {
    double* M;            // [CQ_LD][CQ_LD]: the working matrix; for the inverses an upper-triangular matrix on and above the diagonal and
                          // its inverse X transposed strictly below it (X(i, j) at M[j + 1][i])
    double* sb1;          // [32][33] block scratch: L11^-1 (R11^-T) of the current diagonal block
    double* sb2;          // [32][33] block scratch: U'11^-1
    double* sv;           // [CQ_W] signs
    double* red;          // [CQ_T / 64] reduction scratch
    int* flag;            // [4]
    double* wsdbg;        // workspace (phase stamps of CQ_STAMPS builds)
    double* dinv;         // global: per 32-block, U'11^-1 (row-major 32 x 32) then L11^-1; NULL: not kept
    double* dui;          // global: the last pass's mixed matrix, whose diagonal blocks are U'11^-1; NULL: not kept
};
__device__ __forceinline__ CqLds cq_lds(double* sm)
{
    CqLds L;
    L.M = sm;
    L.sb1 = L.M + CQ_LD * CQ_LD;
    L.sb2 = L.sb1 + 32 * 33;
    L.sv = L.sb2 + 32 * 33;
    L.red = L.sv + CQ_W;
    L.flag = reinterpret_cast<int*>(L.red + CQ_T / 64);
    L.wsdbg = nullptr;
    L.dinv = nullptr;
    L.dui = nullptr;
    return L;
}
constexpr size_t CQ_LDS_BYTES = sizeof(double) * (CQ_LD * CQ_LD + 2 * 32 * 33 + CQ_W + CQ_T / 64) + 64;

// ---- 16 x 16 tiles of the block steps' products, all contracting over one 32-column block (8 matrix-core steps per tile) -------------
// Two lessons shaped this (round 5, stamps in profiles/r05_cq_stamps.txt): (1) as a chain of  ds_read, ds_read, wait, MFMA  a tile cost
// ~190 cycles per step where the MFMA takes 64: the 16 operand values of a tile are requested together, and the NEXT tile's before this
// tile's eight MFMAs; (2) these kernels run every instruction ONCE: unrolled over a dozen tiles per wave the LU kernel was 41 000
// instructions (~250 KB against a 64 KB instruction cache) and ran at the speed of instruction fetch -- the tile lists are walked by ROLLED
// loops (two tiles per trip: the operand sets ping-pong), and the diagonal-block routine is inlined once per kernel.
struct CqOps { double a[8], b[8]; v4d c; };
template <bool SUB, class FA, class FB, class FC>
__device__ __forceinline__ void cq_ops_load(CqOps& o, FA a, FB b, FC c, int i0, int j0, int k0, int l15, int l4)
{
#pragma unroll
    for (int s = 0; s < 8; ++s) { o.a[s] = a(i0 + l15, k0 + 4 * s + l4); o.b[s] = b(k0 + 4 * s + l4, j0 + l15); }
    if (SUB)
#pragma unroll
        for (int r = 0; r < 4; ++r) o.c[r] = c(i0 + l4 + 4 * r, j0 + l15);
}
template <bool SUB, class FS>
__device__ __forceinline__ void cq_ops_mma_store(const CqOps& o, FS store, int i0, int j0, int l15, int l4)
{
    v4d acc = SUB ? o.c : (v4d){0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int s = 0; s < 8; ++s) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(o.a[s], o.b[s], acc, 0, 0, SUB ? 1 : 0);      // SUB: acc - a b
#pragma unroll
    for (int r = 0; r < 4; ++r) store(i0 + l4 + 4 * r, j0 + l15, acc[r]);
}
// Tile lists: mode 0 = all tiles of a grid with ntc tile columns (row-major), 1 = the tiles on and above the tile diagonal of an
// ntc x ntc grid, 2 / 3 = modes 0 / 1 of an ntc x ntc grid WITHOUT its leading 2 x 2 tiles.  Walked with scalar increments.
__device__ __forceinline__ int cq_tile_count(int nt, int ntc, int mode)
{
    return mode == 0 ? nt : (mode == 1 ? ntc * (ntc + 1) / 2 : (mode == 2 ? ntc * ntc - 4 : ntc * (ntc + 1) / 2 - 3));
}
struct CqTileIt {
    int tr, tc, ntc, mode;
    __device__ __forceinline__ int row_first(int r) const { return mode == 0 ? 0 : (mode == 1 ? r : (mode == 2 ? (r < 2 ? 2 : 0) : (r < 2 ? 2 : r))); }
    __device__ __forceinline__ void norm() { while (tc >= ntc && tr < 64) { ++tr; tc = row_first(tr); } }     // (rows 0, 1 of modes 2 / 3 are empty when ntc == 2)
    __device__ __forceinline__ void init(int ntc_, int mode_, int skip) { ntc = ntc_; mode = mode_; tr = 0; tc = row_first(0); norm(); step(skip); }
    __device__ __forceinline__ void step(int k) { for (int q = 0; q < k; ++q) { ++tc; norm(); } }
};
// tiles n = first, first + stride, ... of the list at (r0, c0): out(i, j) = [c(i, j) -] sum_k a(i, k) b(k, j), k = k0 .. k0 + 31, each
// tile stored as soon as it is complete -- the outputs must not overlap the operands of tiles this wave (or a concurrently running one)
// still has to compute.  Rolled loop, two tiles per trip.
template <bool SUB, class FA, class FB, class FC, class FS>
__device__ __forceinline__ void cq_tiles_walk(int nt, int ntc, int mode, int r0, int c0, int k0, int first, int stride, int lane, FA a, FB b, FC c, FS store)
{
    const int l15 = lane & 15, l4 = lane >> 4, total = cq_tile_count(nt, ntc, mode);
    if (first >= total) return;
    CqOps o0, o1;
    CqTileIt it;
    it.init(ntc, mode, first);
    int n = first, i0 = r0 + 16 * it.tr, j0 = c0 + 16 * it.tc;
    cq_ops_load<SUB>(o0, a, b, c, i0, j0, k0, l15, l4);
#pragma nounroll
    for (;;) {
        int i1 = 0, j1 = 0;
        const bool more1 = n + stride < total;
        if (more1) { it.step(stride); i1 = r0 + 16 * it.tr; j1 = c0 + 16 * it.tc; cq_ops_load<SUB>(o1, a, b, c, i1, j1, k0, l15, l4); }
        cq_ops_mma_store<SUB>(o0, store, i0, j0, l15, l4);
        if (!more1) break;
        n += stride;
        const bool more0 = n + stride < total;
        if (more0) { it.step(stride); i0 = r0 + 16 * it.tr; j0 = c0 + 16 * it.tc; cq_ops_load<SUB>(o0, a, b, c, i0, j0, k0, l15, l4); }
        cq_ops_mma_store<SUB>(o1, store, i1, j1, l15, l4);
        if (!more0) break;
        n += stride;
    }
}
// In-place solves of a block step: two tiles that share an operand and whose outputs overlap each other's inputs are computed together
// and stored afterwards, by ONE wave -- no workgroup barrier between compute and store.
//   COLS: out(r0 .. r0+31, jc .. jc+15) = a(r0 .. r0+31, k) x b(k, jc ..)   (the two row tiles of a column: R12 = R11^-T G12, U'12 = L11^-1 W12)
//  !COLS: out(ir .. ir+15, c0 .. c0+31) = a(ir .., k) x b(k, c0 .. c0+31)    (the two column tiles of a row: L21 = W21 U'11^-1)
template <bool COLS, class FA, class FB, class FS>
__device__ __forceinline__ void cq_tile_pair(int r0, int c0, int k0, int lane, FA a, FB b, FS store)
{
    const int l15 = lane & 15, l4 = lane >> 4;
    double x[8], y0[8], y1[8];
#pragma unroll
    for (int s = 0; s < 8; ++s) {
        const int k = k0 + 4 * s + l4;
        if (COLS) { x[s] = b(k, c0 + l15); y0[s] = a(r0 + l15, k); y1[s] = a(r0 + 16 + l15, k); }
        else { x[s] = a(r0 + l15, k); y0[s] = b(k, c0 + l15); y1[s] = b(k, c0 + 16 + l15); }
    }
    v4d acc0 = (v4d){0.0, 0.0, 0.0, 0.0}, acc1 = acc0;
#pragma unroll
    for (int s = 0; s < 8; ++s) {
        acc0 = __builtin_amdgcn_mfma_f64_16x16x4f64(COLS ? y0[s] : x[s], COLS ? x[s] : y0[s], acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f64_16x16x4f64(COLS ? y1[s] : x[s], COLS ? x[s] : y1[s], acc1, 0, 0, 0);
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        store(r0 + l4 + 4 * r, c0 + l15, acc0[r]);
        if (COLS) store(r0 + 16 + l4 + 4 * r, c0 + l15, acc1[r]);
        else store(r0 + l4 + 4 * r, c0 + 16 + l15, acc1[r]);
    }
}
__device__ __forceinline__ void cq_wave_sync_lds()
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// Blocked right-looking Cholesky of the symmetric matrix whose upper triangle is in L.M: R (upper) in place.  Per block of 32 columns:
// the diagonal block on wave 0 on the matrix cores (chol32_mfma: R11 in place, R11^-T -> sb1 and, as the inverse's diagonal block
// R11^-1, into the slots strictly below the diagonal: X(c, i) at M[i + 1][c]), then R12 = R11^-T G12 and G22 -= R12^T R12 on the matrix
// cores.  Look-ahead inside the workgroup (round 5): wave 0 updates only the NEXT diagonal block and factors it while waves 1 - 3 finish
// the previous step's update of the rest of G22, so a block step costs max(diagonal block, trailing update) instead of their sum.
// Returns false on a non-positive pivot (uniform).
__device__ __forceinline__ bool cq_chol_blocked(const CqLds& L, int w, int tid)
{
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
    auto Mx = [&](int i, int j) { return L.M[i * CQ_LD + j]; };
    auto Mt = [&](int i, int k) { return L.M[k * CQ_LD + i]; };
    auto St = [&](int i, int j, double v) { L.M[i * CQ_LD + j] = v; };
    if (tid == 0) L.flag[0] = 1;
    __syncthreads();
#pragma nounroll
    for (int o = 0; o < w; o += 32) {
        // wave 0: the diagonal block o (its three tiles were updated by this wave at the end of the previous trip);
        // waves 1 - 3: the rest of the previous step's trailing update
        if (wave == 0) {
            int ln = lane;
            asm volatile("" : "+v"(ln));                       // keeps the unrolled steps' lane constants inside this trip
            double* const Mo = L.M + o * CQ_LD + o;
            const bool ok = chol32_mfma(ln, [&](int i, int j) { return (j >= i) ? Mo[i * CQ_LD + j] : Mo[j * CQ_LD + i]; },
                                        [&](int i, int j, double v) { if (j >= i) Mo[i * CQ_LD + j] = v; },
                                        [&](int i, int j, double v) { L.sb1[i * 33 + j] = v; if (j <= i) Mo[(i + 1) * CQ_LD + j] = v; });
            if (!ok) L.flag[0] = 0;
        } else if (o > 0) {
            cq_tiles_walk<true>(0, (w - o) >> 4, 3, o, o, o - 32, wave - 1, 3, lane, Mt, Mx, Mx, St);
        }
        __syncthreads();
        const int rest = w - o - 32, ntc = rest >> 4;
        if (rest <= 0) break;
        // R12 = R11^-T G12 in place: a wave takes whole columns of tiles
        auto Li = [&](int i, int k) { return L.sb1[(i - o) * 33 + (k - o)]; };
        for (int tcol = wave; tcol < ntc; tcol += 4) cq_tile_pair<true>(o, o + 32 + 16 * tcol, o, lane, Li, Mx, St);
        __syncthreads();
        // wave 0: the next diagonal block's three upper tiles now (it factors that block at the top of the next trip)
        if (wave == 0) { cq_tiles_walk<true>(3, 2, 1, o + 32, o + 32, o, 0, 1, lane, Mt, Mx, Mx, St); cq_wave_sync_lds(); }
    }
    return L.flag[0] != 0;
}

// Blocked modified LU of W - S R2 = L1 U' (W in L.M, whole; R2 row-major in global memory, zero below its diagonal), the sign of every
// pivot chosen as Householder would (reference qr.c:141-151).  Per block of 32 columns: the diagonal block on wave 0 on the matrix cores
// (lu32_mfma: L11 \ U'11 in place, S, L11^-1 -> sb1, U'11^-1 -> sb2, and both into the workspace: they are the diagonal blocks of the
// inverses cqr_post_kernel builds), then U'12 = L11^-1 (W12 - S R2_12), L21 = W21 U'11^-1, W22 -= L21 U'12 on the matrix cores --
// with the same look-ahead as the Cholesky: wave 0 updates and factors the next diagonal block while waves 1 - 3 update the rest.
// R2g: R2 row-major (ld CQ_W) -- or, first_order, the Gram matrix G2 = I + E itself (both triangles, as cqr_gram_reduce_kernel leaves it), of
// which R2 = I + striu(E) + diag(E) / 2 is read off on the fly: the launch then needs no R2 of its own in global memory before the LU
__device__ __forceinline__ void cq_lu_blocked(const CqLds& L, int w, const double* R2g, bool first_order, int tid)
{
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
    auto r2_at = [&](const double* blk, int i, int j) {        // element (i, j) of a DIAGONAL block
        const double g = blk[i * CQ_W + j];
        return (j > i) ? g : (j == i ? (first_order ? 1.0 + 0.5 * (g - 1.0) : g) : 0.0);
    };
    auto Mx = [&](int i, int j) { return L.M[i * CQ_LD + j]; };
    auto St = [&](int i, int j, double v) { L.M[i * CQ_LD + j] = v; };
    v4d r2t[3];                                                // wave 0: R2's diagonal block of the block it factors next, requested a phase early
    if (wave == 0) {
        lu32_load_r2(lane, [&](int i, int j) { return r2_at(R2g, i, j); }, r2t);
    }
    CQ_STAMP_L(24);
#pragma nounroll
    for (int o = 0; o < w; o += 32) {
        if (wave == 0) {
            int ln = lane;
            asm volatile("" : "+v"(ln));
            double* const Mo = L.M + o * CQ_LD + o;
            lu32_mfma_r2(ln, [&](int i, int j) { return Mo[i * CQ_LD + j]; }, r2t,
                         [&](int i, int j, double v) { Mo[i * CQ_LD + j] = v; }, [&](int i, double v) { L.sv[o + i] = v; },
                         [&](int i, int j, double v) { L.sb1[i * 33 + j] = v; },          // L11^-1(i, j)
                         [&](int i, int j, double v) { L.sb2[j * 33 + i] = v; });         // U'11^-1(j, i)
            if (o + 32 < w) {
                const double* const R2n = R2g + (o + 32) * CQ_W + o + 32;
                lu32_load_r2(lane, [&](int i, int j) { return r2_at(R2n, i, j); }, r2t);      // lands under the next phases
            }
        } else if (o > 0) {
            cq_tiles_walk<true>(0, (w - o) >> 4, 2, o, o, o - 32, wave - 1, 3, lane, Mx, Mx, Mx, St);
        }
        __syncthreads();
        if (o == 0) CQ_STAMP_L(25);
        if (L.dinv) {                                          // the block's inverses for cqr_post_kernel: U'11^-1 then L11^-1, row-major (stores only)
            double* const dinv = L.dinv + (o >> 5) * 2048;
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int e = tid + u * CQ_T, i = (e >> 5) & 31, j = e & 31;
                dinv[e] = (e < 1024) ? L.sb2[i * 33 + j] : L.sb1[i * 33 + j];
                if (L.dui && e < 1024) L.dui[(o + i) * CQ_W + o + j] = (j < i) ? 0.0 : L.sb2[i * 33 + j];   // the last pass's diagonal block
            }
        }
        const int rest = w - o - 32, ntc = rest >> 4;
        if (rest <= 0) break;
        {   // W12 -= S R2_12, the R2 values of a thread requested together (element e: row o + (e >> 7), column o + 32 + (e & 127))
            double r2[16];
#pragma unroll
            for (int u = 0; u < 16; ++u) {
                const int e = tid + u * CQ_T, jj = e & (CQ_W - 1);
                r2[u] = R2g[(o + (e >> 7)) * CQ_W + (jj < rest ? o + 32 + jj : 0)];
            }
#pragma unroll
            for (int u = 0; u < 16; ++u) {
                const int e = tid + u * CQ_T, k = o + (e >> 7), jj = e & (CQ_W - 1);
                if (jj < rest) L.M[k * CQ_LD + o + 32 + jj] -= L.sv[k] * r2[u];
            }
            __syncthreads();
        }
        // U'12 = L11^-1 W12 (whole tile columns per wave) and L21 = W21 U'11^-1 (whole tile rows per wave), both in place; they touch
        // disjoint parts of the matrix, so no barrier between them
        auto Li = [&](int i, int k) { return L.sb1[(i - o) * 33 + (k - o)]; };
        auto Ui = [&](int k, int j) { return L.sb2[(k - o) * 33 + (j - o)]; };
        for (int t = wave; t < ntc; t += 4) cq_tile_pair<true>(o, o + 32 + 16 * t, o, lane, Li, Mx, St);
        for (int t = wave; t < ntc; t += 4) cq_tile_pair<false>(o + 32 + 16 * t, o, o, lane, Mx, Ui, St);
        __syncthreads();
        if (o == 0) CQ_STAMP_L(26);
        // wave 0: the next diagonal block (four tiles) now; it factors that block at the top of the next trip
        if (wave == 0) { cq_tiles_walk<true>(4, 2, 0, o + 32, o + 32, o, 0, 1, lane, Mx, Mx, Mx, St); cq_wave_sync_lds(); }
        if (o == 0) CQ_STAMP_L(27);
    }
    __syncthreads();
}

// inverse of the upper-triangular matrix in L.M (rows / columns < w, w a multiple of 32); the off-diagonal blocks of the matrix are
// DESTROYED.  Diagonal 32 x 32 blocks by back substitution (one wave each, a column per lane); then, as for a 2 x 2 block matrix,
// X12 = -X11 (R12 X22) first inside each half of 64 columns and then between the halves, the products on the matrix cores with the
// intermediate R12 X22 parked in R12's place.  X(i, j) goes to L.M[j + 1][i] (strictly below the diagonal: stride 129 keeps a
// column of X on distinct banks)
template <int NTR>
__device__ __forceinline__ void cq_offdiag(const CqLds& L, int r0, int c0, int nc, int tid)
{
    // X(r0 : r0 + nr, c0 : c0 + nc) = -X(r0 .., r0 ..) (R(r0 .., c0 ..) X(c0 .., c0 ..)), nr = 16 NTR, nc a multiple of 16 up to 64, c0 = r0 + nr.
    // A wave owns a tile COLUMN (its B fragment is shared by the NTR tile rows, whose accumulators interleave on the matrix core: a
    // chain of dependent f64 MFMAs issues one every ~70 cycles, independent ones every 64); static tile-row count, operands of the next
    // k-step requested before this one's instructions
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63, l15 = lane & 15, l4 = lane >> 4;
    constexpr int nr = 16 * NTR;
    const int ntc = nc >> 4;
    const bool on = wave < ntc;
    const int j0 = c0 + 16 * wave, j = j0 + l15;
    v4d acc[NTR];
#pragma unroll
    for (int q = 0; q < NTR; ++q) acc[q] = (v4d){0.0, 0.0, 0.0, 0.0};
    if (on) {
        // P = R(r0 .., c0 ..) X(c0 .., c0 ..): X upper triangular, k <= j
        double a[NTR], an[NTR], b, bn = 0.0;
        auto ld = [&](int k, double (&av)[NTR], double& bv) {
            const int kk = k + l4;
            bv = (kk <= j) ? L.M[(j + 1) * CQ_LD + kk] : 0.0;
#pragma unroll
            for (int q = 0; q < NTR; ++q) av[q] = L.M[(r0 + 16 * q + l15) * CQ_LD + kk];
        };
        ld(c0, a, b);
        for (int k = c0; k < j0 + 16; k += 4) {
            if (k + 4 < j0 + 16) ld(k + 4, an, bn);
#pragma unroll
            for (int q = 0; q < NTR; ++q) acc[q] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[q], b, acc[q], 0, 0, 0);
            b = bn;
#pragma unroll
            for (int q = 0; q < NTR; ++q) a[q] = an[q];
        }
    }
    __syncthreads();
    if (on) {
#pragma unroll
        for (int q = 0; q < NTR; ++q)
#pragma unroll
            for (int r = 0; r < 4; ++r) L.M[(r0 + 16 * q + l4 + 4 * r) * CQ_LD + j0 + l15] = acc[q][r];
    }
    __syncthreads();
#pragma unroll
    for (int q = 0; q < NTR; ++q) acc[q] = (v4d){0.0, 0.0, 0.0, 0.0};
    if (on) {
        // X12 = -X(r0 .., r0 ..) P: X upper triangular, k >= i (tile row q starts at k = r0 + 16 q)
#pragma unroll
        for (int ks = 0; ks < 4 * NTR; ++ks) {
            const int kk = r0 + 4 * ks + l4;
            const double b = L.M[kk * CQ_LD + j0 + l15];
#pragma unroll
            for (int q = 0; q < NTR; ++q)
                if (4 * ks >= 16 * q) {
                    const int i = r0 + 16 * q + l15;
                    acc[q] = __builtin_amdgcn_mfma_f64_16x16x4f64((i <= kk) ? L.M[(kk + 1) * CQ_LD + i] : 0.0, b, acc[q], 0, 0, 0);
                }
        }
    }
    __syncthreads();
    if (on) {
#pragma unroll
        for (int q = 0; q < NTR; ++q)
#pragma unroll
            for (int r = 0; r < 4; ++r) L.M[(j0 + l15 + 1) * CQ_LD + r0 + 16 * q + l4 + 4 * r] = -acc[q][r];
    }
    __syncthreads();
}
__device__ __forceinline__ void cq_upper_inv(const CqLds& L, int w, int have_diag, int, int tid)
{
    // have_diag: the diagonal 32 x 32 blocks of the inverse are in their slots already (left by cq_chol_blocked).  Nothing below the
    // diagonal needs to be zero beforehand: every read of an X slot is masked to the part that has been written
    const int nblk = w >> 5, wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
    __syncthreads();
    CQ_STAMP_L(20);
    if (have_diag) goto offdiag;
    if (wave < nblk) {
        // column j of the block's inverse in registers (static indices: the loops unroll); row i of R is a broadcast read
        const int o = 32 * wave, j = lane & 31;
        const double dinv = rcp_newton(L.M[(o + j) * CQ_LD + o + j]);
        // (dot-product form: a chain of 496 dependent FMAs, 7.5 us; the column (axpy) form with a residual vector in registers was
        // measured at 17 us -- 64 more live registers spill in this kernel)
        double x[32];
#pragma unroll
        for (int i = 31; i >= 0; --i) {
            double acc = (i == j) ? 1.0 : 0.0;
#pragma unroll
            for (int k = i + 1; k < 32; ++k) acc -= L.M[(o + i) * CQ_LD + o + k] * x[k];
            x[i] = (i <= j) ? acc * readlane_f64(dinv, i) : 0.0;
        }
        if (lane < 32) {
            double* xj = L.M + (o + j + 1) * CQ_LD + o;          // x(i) = X(o + i, o + j)
#pragma unroll
            for (int i = 0; i < 32; ++i)
                if (i <= j) xj[i] = x[i];
        }
    }
    __syncthreads();
offdiag:
    CQ_STAMP_L(21);
    if (w >= 64) cq_offdiag<2>(L, 0, 32, 32, tid);
    if (w == 128) cq_offdiag<2>(L, 64, 96, 32, tid);
    CQ_STAMP_L(22);
    if (w > 64) cq_offdiag<4>(L, 0, 64, w - 64, tid);
    CQ_STAMP_L(23);
}

// the inverse out of L.M into a row-major global matrix (zero below the diagonal)
__device__ __forceinline__ void cq_inv_out(const CqLds& L, double* X, int w, int tid)
{
    for (int e = tid; e < w * CQ_W; e += CQ_T) {
        const int i = e >> 7, j = e & (CQ_W - 1);
        if (j >= w) continue;
        cq_st(X + i * CQ_W + j, (j >= i) ? L.M[(j + 1) * CQ_LD + i] : 0.0);
    }
}

// C = A diag(d) B for upper-triangular A (L.M, on and above the diagonal) and upper-triangular B (transposed strictly below the
// diagonal of L.M, where the inverses leave their result: B(k, j) at M[j + 1][k]); d = NULL: no scaling.  A wave owns two tile ROWS
// (tr and 7 - tr: nine tiles on and above the diagonal each at w = 128): the A fragment of a k-step is shared by the row's tiles and
// their accumulators interleave on the matrix core.  f(i, j, value) for every element on and above the tile diagonal.
// (Round 5: a rolled k loop with a run-time tile row -- a third of the code -- was measured and is SLOWER: cqr_post_kernel 55 -> 92 us;
// the unrolled form below stays.)
template <int TR, bool FULL, class F>
__device__ __forceinline__ void cq_product_row(const CqLds& L, const double* d, int nt, int l15, int l4, F f)
{
    if (!FULL && TR >= nt) return;
    // tile row TR: tiles (TR, TR .. 7), k from 16 TR; tile q needs k < 16 q + 16.  Static ranges, the next k-step's operands requested
    // before this one's matrix-core instructions
    v4d acc[8];
#pragma unroll
    for (int q = TR; q < 8; ++q) acc[q] = (v4d){0.0, 0.0, 0.0, 0.0};
    const int i = 16 * TR + l15;
    double a, an = 0.0, b[8], bn[8];
    auto ld = [&](int k, double& av, double (&bv)[8]) {
        const int kk = k + l4;
        av = (kk >= i) ? L.M[i * CQ_LD + kk] * (d ? d[kk] : 1.0) : 0.0;
#pragma unroll
        for (int q = TR; q < 8; ++q)
            if (k < 16 * q + 16) { const int j = 16 * q + l15; bv[q] = (kk <= j) ? L.M[(j + 1) * CQ_LD + kk] : 0.0; }
    };
    ld(16 * TR, a, b);
#pragma unroll
    for (int k = 16 * TR; k < 128; k += 4) {
        if (!FULL && k >= 16 * nt) break;
        if (k + 4 < 128) ld(k + 4, an, bn);
#pragma unroll
        for (int q = TR; q < 8; ++q)
            if (k < 16 * q + 16 && (FULL || q < nt)) acc[q] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b[q], acc[q], 0, 0, 0);
        a = an;
#pragma unroll
        for (int q = TR; q < 8; ++q) b[q] = bn[q];
    }
#pragma unroll
    for (int q = TR; q < 8; ++q)
        if (FULL || q < nt)
#pragma unroll
            for (int r = 0; r < 4; ++r) f(16 * TR + l4 + 4 * r, 16 * q + l15, acc[q][r]);
}
template <bool FULL, class F>
__device__ __forceinline__ void cq_product_rows(const CqLds& L, const double* d, int nt, int wave, int l15, int l4, F f)
{
    switch (wave) {
    case 0: cq_product_row<0, FULL>(L, d, nt, l15, l4, f); cq_product_row<7, FULL>(L, d, nt, l15, l4, f); break;
    case 1: cq_product_row<1, FULL>(L, d, nt, l15, l4, f); cq_product_row<6, FULL>(L, d, nt, l15, l4, f); break;
    case 2: cq_product_row<2, FULL>(L, d, nt, l15, l4, f); cq_product_row<5, FULL>(L, d, nt, l15, l4, f); break;
    default: cq_product_row<3, FULL>(L, d, nt, l15, l4, f); cq_product_row<4, FULL>(L, d, nt, l15, l4, f); break;
    }
}
template <class F>
__device__ __forceinline__ void cq_upper_product(const CqLds& L, const double* d, int w, int tid, F f)
{
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63, l15 = lane & 15, l4 = lane >> 4, nt = w >> 4;
    if (nt == 8) cq_product_rows<true>(L, d, nt, wave, l15, l4, f);
    else cq_product_rows<false>(L, d, nt, wave, l15, l4, f);
}

// a row-major global upper-triangular matrix into the transposed slots strictly below the diagonal of L.M (operand B of cq_upper_product)
__device__ __forceinline__ void cq_load_lowerT(const CqLds& L, const double* G, int w, int tid)
{
    cq_elems(w, tid, [&](int k, int j) { return G[k * CQ_W + j]; }, [&](int k, int j, double v) { if (j < w && j >= k) L.M[(j + 1) * CQ_LD + k] = v; });
}

// a row-major global matrix (upper triangle) into L.M
__device__ __forceinline__ void cq_load_upper(const CqLds& L, const double* G, int w, int tid)
{
    cq_elems(w, tid, [&](int i, int j) { return G[i * CQ_W + j]; }, [&](int i, int j, double v) { if (j < w && j >= i) L.M[i * CQ_LD + j] = v; });
}

// ---------------------------------------------------------------------------------------------------------------------------------
// R1 = chol(G1), R1^-1.  G: column-major ld CQ_W, symmetric (read along its rows).  status[0] |= 1 on a non-positive pivot.
// ---------------------------------------------------------------------------------------------------------------------------------
// shift_scale > 0: the SHIFTED factorisation that opens the retry of a refused panel (qrd_panel_cqr_retry; shifted CholeskyQR3, Fukaya et
// al.): R0 = chol(G1 + s I), s = shift_scale * trace(G1) with shift_scale = 11 (m n + n (n + 1)) u -- positive definite whatever the
// condition of the panel -- kept in CQ_R0 as well, and the refusal word of the first attempt is cleared (this launch is the retry's first).
__global__ __launch_bounds__(CQ_T) void cqr_chol_kernel(double* ws, int w, int* status, double shift_scale)
{
    extern __shared__ double sm[];
    CqLds L = cq_lds(sm);
#ifdef CQ_STAMPS
    L.wsdbg = ws;
#endif
    const int tid = threadIdx.x;
    // G(i, j) = G(j, i): consecutive lanes read consecutive addresses
    cq_elems(w, tid, [&](int i, int j) { return ws[CQ_G1 + j + CQ_W * i]; }, [&](int i, int j, double v) { if (j < w && j >= i) L.M[i * CQ_LD + j] = v; });
    if (shift_scale > 0.0) {
        if (tid == 0) status[0] = 0;
        __syncthreads();
        double tr = 0.0;                                      // every thread the same sum, in the same order
        for (int i = 0; i < w; ++i) tr += L.M[i * CQ_LD + i];
        __syncthreads();
        if (tid < w) L.M[tid * CQ_LD + tid] += shift_scale * tr;      // (a NaN / Inf trace ends at the pivot test below like any other bad panel)
        __syncthreads();
    }
    CQ_STAMP(0);
    const bool ok = cq_chol_blocked(L, w, tid);
    CQ_STAMP(1);
    if (!ok) { if (tid == 0) status[0] = 1; return; }
#pragma unroll 8
    for (int e = tid; e < w * CQ_W; e += CQ_T) {
        const int i = e >> 7, j = e & (CQ_W - 1);
        if (j < w) {
            const double r = (j >= i) ? L.M[i * CQ_LD + j] : 0.0;
            cq_st(ws + CQ_R1 + i * CQ_W + j, r);
            if (shift_scale > 0.0) cq_st(ws + CQ_R0 + i * CQ_W + j, r);
        }
    }
    CQ_STAMP(2);
    // what pass 2 multiplies by (cqr_stream_body: block back substitution): R1's off-diagonal 32 x 32 blocks and the inverses of its
    // diagonal blocks, which the Cholesky's augmented columns left in the slots below the diagonal (X(i, j) at M[j + 1][i]) -- the
    // full inverse (two more levels of products, 14 us + 4 us to write it) is not formed any more
#pragma unroll 8
    for (int e = tid; e < w * CQ_W; e += CQ_T) {
        const int i = e >> 7, j = e & (CQ_W - 1);
        if (j >= w) continue;
        cq_st(ws + CQ_R1I + i * CQ_W + j, (j < i) ? 0.0 : (((i ^ j) & ~31) == 0 ? L.M[(j + 1) * CQ_LD + i] : L.M[i * CQ_LD + j]));
    }
    CQ_STAMP(4);
}

// ---------------------------------------------------------------------------------------------------------------------------------
// Retried panel: R1 <- R1 R0 in the workspace (R1: the first Cholesky factor of the PRECONDITIONED panel Q0 = A R0^-1, just written by
// cqr_chol_kernel; R0 from CQ_R0), so that the rider of the last pass, R = S R2 R1, comes out as the R of A = Q (S R2 R1 R0) with nothing
// else of the pipeline knowing about the preconditioner.  The mixed matrix the next pass solves with (CQ_R1I) is R1's and stays.
// ---------------------------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(CQ_T) void cqr_rmul_kernel(double* ws, int w, const int* status)
{
    extern __shared__ double sm[];
    CqLds L = cq_lds(sm);
    const int tid = threadIdx.x;
    if (status[0]) return;
    cq_load_upper(L, ws + CQ_R1, w, tid);
    cq_load_lowerT(L, ws + CQ_R0, w, tid);
    __syncthreads();
    cq_upper_product(L, nullptr, w, tid, [&](int i, int j, double v) { cq_st(ws + CQ_R1 + i * CQ_W + j, (j >= i) ? v : 0.0); });
}

// ---------------------------------------------------------------------------------------------------------------------------------
// Everything between the two streaming passes, on one workgroup (see the header).  Q_top: the first w rows of Q in Vw.
// Outputs: Vw top block <- Q_top - S R2; ws: U'^-1, L1 \ U', R, T, S; status[0] |= 1 when the panel is refused.
// ---------------------------------------------------------------------------------------------------------------------------------
// (the body returns true -- uniformly -- when the panel is refused; cqr_lu_kernel publishes the verdict)
__device__ __forceinline__ bool cq_lu_body(double* sm, double* ws, int w, double* Vw, int ldv, const int* status)
{
    CqLds L = cq_lds(sm);
    L.dinv = ws + CQ_X3;
    L.dui = ws + CQ_UI;
#ifdef CQ_STAMPS
    L.wsdbg = ws;
#endif
    const int tid = threadIdx.x;
    if (status[0]) return true;                               // the first Cholesky failed
    CQ_STAMP(8);
    // Q_top (w x w, the first rows of Q) is requested now -- 64 values per thread, consecutive lanes = consecutive rows of a column -- and
    // goes into L.M once G2 has been dealt with: one global round trip instead of two in a row
    double qt[64];
#pragma unroll
    for (int u = 0; u < 64; ++u) {
        const int e = tid + u * CQ_T, j = e >> 7, i = e & (CQ_W - 1);
        qt[u] = Vw[(i < w ? i : w - 1) + (size_t) ldv * (j < w ? j : w - 1)];
    }
    // ---- G2 -> L.M (upper); its distance from I decides between the first-order factor, the Cholesky and the refusal
    double dmax = 0.0;
    cq_elems(w, tid, [&](int i, int j) { return ws[CQ_G2 + j + CQ_W * i]; }, [&](int i, int j, double g) {
        if (j < w && j >= i) {
            L.M[i * CQ_LD + j] = g;
            const double d = fabs(g - (i == j ? 1.0 : 0.0));
            dmax = (d == d) ? fmax(dmax, d) : 1e300;
        }
    });
    CQ_STAMP(28);
    for (int o = 32; o > 0; o >>= 1) dmax = fmax(dmax, __shfl_xor(dmax, o));
    if ((tid & 63) == 0) L.red[tid >> 6] = dmax;
    __syncthreads();
    dmax = 0.0;
    for (int q = 0; q < CQ_T / 64; ++q) dmax = fmax(dmax, L.red[q]);
    if (!(dmax <= QRD_GUARD_THR)) return true;
    const bool first_order = dmax <= QRD_CHOL1_THR;
    if (first_order) {
        // G2 = I + E, |E| <= 1e-9: R2 = I + triu(E, 1) + diag(E) / 2 to ~1e-18
        if (tid < w) L.M[tid * CQ_LD + tid] = 1.0 + 0.5 * (L.M[tid * CQ_LD + tid] - 1.0);
        __syncthreads();
    } else {
        const bool ok = cq_chol_blocked(L, w, tid);
        if (!ok) return true;
    }
    CQ_STAMP(29);
    // R2 -> global (cqr_post_kernel's products read it from there; this launch's LU too unless R2 is first order: it then reads G2
    // and nothing waits for these stores); R2^-1 -> X1 (first order: it is 2 I - R2, which the T rider reads off R2 itself)
#pragma unroll 8
    for (int e = tid; e < w * CQ_W; e += CQ_T) {
        const int i = e >> 7, j = e & (CQ_W - 1);
        if (j >= w) continue;
        const double r = (j >= i) ? L.M[i * CQ_LD + j] : 0.0;
        cq_st(ws + CQ_R2 + i * CQ_W + j, r);
    }
    if (tid == 0) cq_st(ws + CQ_FO, first_order ? 1.0 : 0.0);
    CQ_STAMP(30);
    if (!first_order) {
        __syncthreads();
        cq_upper_inv(L, w, 1, 0, tid);
        cq_inv_out(L, ws + CQ_X1, w, tid);
        cq_sync_global();
    } else __syncthreads();
    CQ_STAMP(9);
    // ---- W = Q_top -> L.M (whole), blocked modified LU
#pragma unroll
    for (int u = 0; u < 64; ++u) {
        const int e = tid + u * CQ_T, j = e >> 7, i = e & (CQ_W - 1);
        if (i < w && j < w) L.M[i * CQ_LD + j] = qt[u];
    }
    __syncthreads();
    cq_lu_blocked(L, w, first_order ? ws + CQ_G2 : ws + CQ_R2, first_order, tid);
    CQ_STAMP(10);
    // L1 \ U' and S out.  (Rounds 4 wrote Q_top - S R2 back into the Q buffer here so that the last pass would turn it into L1: dead
    // work -- cqr_top_kernel overwrites the top block of A and Vw behind that pass anyway.)
#pragma unroll 8
    for (int e = tid; e < w * CQ_W; e += CQ_T) {
        const int i = e >> 7, j = e & (CQ_W - 1);
        if (j < w) {
            const double v = L.M[i * CQ_LD + j];
            cq_st(ws + CQ_LU + i * CQ_W + j, v);
            // what the last pass multiplies by: U''s off-diagonal 32 x 32 blocks here, the inverses of its diagonal blocks from the
            // blocked LU's diagonal steps -- nobody forms U'^-1 any more
            if (((i ^ j) & ~31) != 0) cq_st(ws + CQ_UI + i * CQ_W + j, (j < i) ? 0.0 : v);
        }
    }
    if (tid < w) cq_st(ws + CQ_SV + tid, L.sv[tid]);
    CQ_STAMP(11);
    return false;
}

// status[0]: this panel was refused (the kernels behind this one return at once, A is untouched); status[1]: panels refused in LATCH
// mode (hflag == NULL) since the host last cleared the word (sticky: reported at qr_plan_sync; a refusal the polling host reads from
// hflag is dealt with at once and must not surface again after a later switch of modes); hflag (optional, host memory mapped into the
// device): 2 * seq + refused, written with system scope as soon as the verdict exists -- the host reads it while the panel's last pass
// is still running, so its decision (the leaf chain, or nothing) is queued long before the stream gets there
__global__ __launch_bounds__(CQ_T) void cqr_lu_kernel(double* ws, int w, double* Vw, int ldv, int* status, unsigned* hflag, unsigned seq)
{
    extern __shared__ double sm[];
    const bool refused = cq_lu_body(sm, ws, w, Vw, ldv, status);
    if (threadIdx.x == 0) {
        if (refused) { status[0] = 1; if (!hflag) status[1] += 1; }
        if (hflag) {
            __threadfence_system();
            __hip_atomic_store(hflag, 2u * seq + (refused ? 1u : 0u), __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
        }
    }
}

// the diagonal 32 x 32 blocks of an inverse into their slots (X(i, j) at M[j + 1][i], i <= j) from the blocks the LU kernel left in the
// workspace: per block U'11^-1 (row-major) then L11^-1;  lower = true: the inverse wanted is (L1^T)^-1, whose block is (L11^-1)^T
__device__ __forceinline__ void cq_diag_slots(const CqLds& L, const double* dinv, int w, int tid, bool lower)
{
    for (int e = tid; e < (w >> 5) * 1024; e += CQ_T) {
        const int blk = e >> 10, i = (e >> 5) & 31, j = e & 31, o = 32 * blk;          // element (i, j) of the block, i <= j kept
        if (i > j) continue;
        const double v = lower ? dinv[blk * 2048 + 1024 + j * 32 + i] : dinv[blk * 2048 + i * 32 + j];
        L.M[(o + j + 1) * CQ_LD + o + i] = v;
    }
}

// What follows the LU, on TWO workgroups side by side (four independent pieces were 120 us in a row on one):
//   workgroup 0: U'^-1 -> UI (the operand of the V pass), then R = S R2 R1
//   workgroup 1: U = U' R2^-1, the inverse of L1^T, T = -U S L1^-T
// Operands from the workspace (written by cqr_lu_kernel / cqr_chol_kernel: an earlier launch); S from ws + CQ_SV.
// (round 5, later: NOTHING stands between the LU and the last pass any more -- that pass solves with U' block by block, cqr_stream_body --
// and the two pieces below ride in its launch as its first two workgroups, cqr_vpass_kernel: 56 us of one-workgroup work off the chain)
// Where the riders put the panel's top block (end of round 5: cqr_top_kernel, a launch of its own behind the last pass, did this from the
// workspace; the streaming workgroups of the pass then leave the first w rows alone).  mode 0: nowhere (cqr_top_kernel follows);
// 1: A <- R on and above the diagonal, L1 below, Vw <- the unit lower L1; 2 (parked): A <- the unit lower L1, zeros above.  T, tau always.
struct CqTop { double* A; int lda; double* Vw; int ldv; double* T; int ldt; double* tau; int mode; };

__device__ __forceinline__ void cq_post_r(const CqLds& L, double* ws, int w, int tid, const CqTop& top)
{
    if (top.mode) {
        // L1 (from the LU kernel's L1 \ U') into the top block: consecutive threads, consecutive rows of a column
        for (int e = tid; e < w * w; e += CQ_T) {
            const int j = e / w, i = e - j * w;
            const double lu = ws[CQ_LU + i * CQ_W + j], unit = (j < i) ? lu : (j == i ? 1.0 : 0.0);
            if (top.mode == 2) top.A[i + (size_t) top.lda * j] = unit;
            else {
                if (j < i) top.A[i + (size_t) top.lda * j] = lu;
                top.Vw[i + (size_t) top.ldv * j] = unit;
            }
        }
    }
    // ---- R = S R2 R1: R2 -> upper triangle, R1 -> below the diagonal
    cq_load_upper(L, ws + CQ_R2, w, tid);
    cq_load_lowerT(L, ws + CQ_R1, w, tid);
    __syncthreads();
    cq_upper_product(L, nullptr, w, tid, [&](int i, int j, double v) {
        const double r = (j >= i) ? L.sv[i] * v : 0.0;
        cq_st(ws + CQ_RR + i * CQ_W + j, r);
        if (top.mode == 1 && j >= i) top.A[i + (size_t) top.lda * j] = r;
    });
}
__device__ __forceinline__ void cq_post_t(const CqLds& L, double* ws, int w, int tid, const CqTop& top)
{
    if (top.mode) {
        // (the product below stores the tiles on and above the tile diagonal; the others are zero)
        for (int e = tid; e < w * w; e += CQ_T) {
            const int j = e / w, i = e - j * w;
            if ((i >> 4) > (j >> 4)) top.T[i + (size_t) top.ldt * j] = 0.0;
        }
    }
    // ---- U = U' R2^-1 -> X2 (U' above the diagonal, R2^-1 transposed below it)
    cq_load_upper(L, ws + CQ_LU, w, tid);
    if (ws[CQ_FO] != 0.0)                                     // first-order R2: R2^-1 = 2 I - R2
        cq_elems(w, tid, [&](int k, int j) { return ws[CQ_R2 + k * CQ_W + j]; },
                 [&](int k, int j, double v) { if (j < w && j >= k) L.M[(j + 1) * CQ_LD + k] = (j > k) ? -v : 2.0 - v; });
    else cq_load_lowerT(L, ws + CQ_X1, w, tid);
    __syncthreads();
    cq_upper_product(L, nullptr, w, tid, [&](int i, int j, double v) { cq_st(ws + CQ_X2 + i * CQ_W + j, (j >= i) ? v : 0.0); });
    __syncthreads();
    // ---- L1^-T: the inverse of the unit upper-triangular L1^T stays below the diagonal of L.M
    // (L1^T)(j, i) = L1(i, j): row i of LU read along j
    cq_elems(w, tid, [&](int i, int j) { return ws[CQ_LU + i * CQ_W + j]; }, [&](int i, int j, double v) { if (j < w && i >= j) L.M[j * CQ_LD + i] = (i == j) ? 1.0 : v; });
    cq_diag_slots(L, ws + CQ_X3, w, tid, true);
    __syncthreads();
    cq_upper_inv(L, w, 1, 0, tid);
    // ---- T = -U S L1^-T: U (X2) -> the upper triangle of L.M, S folded into U's columns
    cq_sync_global();
    cq_load_upper(L, ws + CQ_X2, w, tid);
    __syncthreads();
    cq_upper_product(L, L.sv, w, tid, [&](int i, int j, double v) {
        const double tv = (j >= i) ? -v : 0.0;
        cq_st(ws + CQ_TT + i * CQ_W + j, tv);
        if (top.mode) {
            top.T[i + (size_t) top.ldt * j] = tv;
            if (i == j) top.tau[i] = tv;
        }
    });
}
// both pieces in one launch (two workgroups): the stage-by-stage entry points of devtools/tools_cqr_debug.py
__global__ __launch_bounds__(CQ_T) void cqr_post_kernel(double* ws, int w, const int* status)
{
    extern __shared__ double sm[];
    const CqLds L = cq_lds(sm);
    const int tid = threadIdx.x;
    if (status[0]) return;
    if (tid < w) L.sv[tid] = ws[CQ_SV + tid];
    const CqTop none{};
    if (blockIdx.x == 0) { __syncthreads(); cq_post_r(L, ws, w, tid, none); }
    else cq_post_t(L, ws, w, tid, none);
}

// ---------------------------------------------------------------------------------------------------------------------------------
// The streaming passes, one kernel template:
//   MULT: dst rows = src rows times an upper-triangular w x w matrix X (row-major in ws).  Matrix cores with the product transposed
//         -- D(col, row) = sum_k X(k, col) src(row, k) -- so that the accumulator registers of a lane are four columns of 16
//         CONSECUTIVE ROWS: loads and stores are both whole 128-byte lines of a column.  A wave takes 16 rows at a time: the 16 x w
//         row block sits in 32 operand registers, column tile jt needs only k < 16 (jt + 1).  X sits in LDS as its 36 upper
//         16 x 16 blocks (72 KB instead of 132).
//   GRAM: the Gram matrix of the rows this wave has produced (MULT) or read (!MULT) accumulates in 36 accumulator tiles: the 16 x w
//         block goes through a wave-private LDS tile to change from "lane = row" to "lane = column", then 144 matrix-core
//         instructions per block -- the separate V^T V pass over the panel (gemm_tn: 176 us at 262144 x 128) disappears.  Per-workgroup
//         partials go to `slabs`, summed by cqr_gram_reduce_kernel in a fixed order (deterministic).
//   pass 1: !MULT, GRAM  (G1 = A^T A)     pass 2: MULT, GRAM  (Q = A R1^-1 -> Vw, G2 = Q^T Q)     pass 3: MULT, DST2  (V -> Vw and A)
// The next block's rows are requested before the current block's matrix-core work.
// ---------------------------------------------------------------------------------------------------------------------------------
// Workgroup barrier that orders LDS only.  __syncthreads() is a release/acquire fence on EVERY address space: hipcc puts
// `s_waitcnt vmcnt(0)` in front of the barrier, i.e. the next block's global loads -- requested early precisely to run under this block's
// matrix-core work -- had to land before the barrier: memory time and matrix-core time simply added up (Gram pass 132 us = 66 + 66).
// (An inline-asm `s_waitcnt lgkmcnt(0); s_barrier` does not help: the compiler puts its own vmcnt(0) in front of inline asm that
// clobbers memory.  The fence builtin with an address-space argument compiles to exactly lgkmcnt(0) + s_barrier.)
__device__ __forceinline__ void cs_lds_barrier()
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
}
constexpr int CS_THREADS = 256;
constexpr int CS_QLD = 130;                                    // row stride of the workgroup's block in LDS (accumulator-order writes conflict-free, operand reads 2-way)
constexpr int CS_XC = 36 * 256;                                // doubles of the compact X
constexpr int CS_NWG = 256;                                    // workgroups of the passes that hold X in LDS (one per compute unit)
constexpr int CS_NWG_GRAM = 256;                               // workgroups (= partials) of the Gram-only pass
constexpr size_t CS_LDS_BYTES = sizeof(double) * (CS_XC + 4 * 16 * CS_QLD);
constexpr size_t CS_LDS_GRAM = sizeof(double) * (4 * 16 * CS_QLD);      // the Gram-only pass
constexpr size_t CQ_VP_LDS_BYTES = CQ_LDS_BYTES > CS_LDS_BYTES ? CQ_LDS_BYTES : CS_LDS_BYTES;   // cqr_vpass_kernel: one-workgroup riders + the streaming workgroups
__device__ __forceinline__ int cs_blk(int kb, int jb) { return (jb * (jb + 1) / 2 + kb) * 256; }
// the 144 X operands of a 16-row block's solve by 32-column blocks, in the order the MFMAs use them.  Block b (start cs_s_off(b)): for the
// destination half h = 0, 1, every earlier 16-column tile it (0 .. 2 b - 1), sub-step r: X tile (it, 2 b + h), sub-step r -- 16 b operands
// (half 0's chain first: it is complete, and no wait state is left to pad, when the diagonal block's products need it); then the
// diagonal block's three tiles (2b, 2b), (2b, 2b+1), (2b+1, 2b+1), four sub-steps each.  cs_s_dec: tile row | tile column << 4 | r << 8.
__host__ __device__ constexpr int cs_s_off(int b) { return 8 * b * (b - 1) + 12 * b; }
__host__ __device__ constexpr int cs_s_dec(int n)
{
    int b = 0;
    while (b < 3 && n >= cs_s_off(b + 1)) ++b;
    const int m = n - cs_s_off(b);
    if (m < 16 * b) return ((m % (8 * b)) >> 2) | ((2 * b + m / (8 * b)) << 4) | ((m & 3) << 8);
    const int d = m - 16 * b;
    return d < 4 ? ((2 * b) | ((2 * b) << 4) | (d << 8)) : (d < 8 ? ((2 * b) | ((2 * b + 1) << 4) | ((d - 4) << 8)) : ((2 * b + 1) | ((2 * b + 1) << 4) | ((d - 8) << 8)));
}
template <int I> struct cs_int { static constexpr int value = I; };
struct cs_true { static constexpr bool value = true; };
struct cs_false { static constexpr bool value = false; };

// tile row TR of the Gram matrix of the workgroup's 64 x w block in LDS: tiles (TR, TR .. 7), 16 k-steps of 4 rows; the B operands of a
// k-step are read together, unconditionally (the block's LDS rows are 130 doubles whatever w), then the matrix-core instructions
// mid(i), i = IBASE .. IBASE + 15: called once per k-step (PIPE) -- the caller issues one global load of the next block there: a burst of
// 32 loads in front of the Gram instructions fills the vector-memory queue and the wave waits AT a load, the MFMAs behind it too
template <int TR, bool FULL, bool PIPE, int IBASE, class FM>
__device__ __forceinline__ void cs_gram_row(v4d (&g)[8], const double* Qall, int nct, int l15, int l4, FM mid)
{
    if (!FULL && TR >= nct) {
        if (PIPE)
#pragma unroll
            for (int i = 0; i < 16; ++i) mid(IBASE + i);
        return;
    }
    const double* qr = Qall + l4 * CS_QLD + l15;
    if (PIPE) {
        // one wave per SIMD: nothing else hides the LDS latency -- fully unrolled, the next k-step's operands requested before this
        // one's matrix-core instructions
        double b[8], bn[8];
#pragma unroll
        for (int t = TR; t < 8; ++t) b[t] = qr[16 * t];
#pragma unroll
        for (int k = 0; k < 64; k += 4) {
            if (k + 4 < 64)
#pragma unroll
                for (int t = TR; t < 8; ++t) bn[t] = qr[(k + 4) * CS_QLD + 16 * t];
#pragma unroll
            for (int t = TR; t < 8; ++t)
                if (FULL || t < nct) g[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(b[TR], b[t], g[t], 0, 0, 0);
            mid(IBASE + (k >> 2));
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int t = TR; t < 8; ++t) b[t] = bn[t];
        }
    } else {
        // two waves per SIMD (the Gram-only pass): the other wave covers the latency, the registers go to the accumulators
#pragma unroll 2
        for (int k = 0; k < 64; k += 4) {
            double b[8];
#pragma unroll
            for (int t = TR; t < 8; ++t) b[t] = qr[k * CS_QLD + 16 * t];
#pragma unroll
            for (int t = TR; t < 8; ++t)
                if (FULL || t < nct) g[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(b[TR], b[t], g[t], 0, 0, 0);
        }
    }
}
template <bool FULL, bool PIPE, class FM>
__device__ __forceinline__ void cs_gram_rows(v4d (&g0)[8], v4d (&g1)[8], const double* Qall, int nct, int wave, int l15, int l4, FM mid)
{
    switch (wave) {                                           // (scalar: static tile ranges; FULL = 128 columns: no predicate around a matrix-core instruction)
    case 0: cs_gram_row<0, FULL, PIPE, 0>(g0, Qall, nct, l15, l4, mid); cs_gram_row<7, FULL, PIPE, 16>(g1, Qall, nct, l15, l4, mid); break;
    case 1: cs_gram_row<1, FULL, PIPE, 0>(g0, Qall, nct, l15, l4, mid); cs_gram_row<6, FULL, PIPE, 16>(g1, Qall, nct, l15, l4, mid); break;
    case 2: cs_gram_row<2, FULL, PIPE, 0>(g0, Qall, nct, l15, l4, mid); cs_gram_row<5, FULL, PIPE, 16>(g1, Qall, nct, l15, l4, mid); break;
    default: cs_gram_row<3, FULL, PIPE, 0>(g0, Qall, nct, l15, l4, mid); cs_gram_row<4, FULL, PIPE, 16>(g1, Qall, nct, l15, l4, mid); break;
    }
}

template <bool MULT, bool GRAM, bool DST2>
__device__ __forceinline__ void cqr_stream_body(double* sm, const double* __restrict__ X, int w, int mk, const double* src, int lds_, double* dst,
                                                int ldd, double* dst2, int ldd2, double* slabs, const int* status, const int bid, const int nbid, const int skip = 0)
{
    if (MULT && status && status[0]) return;                  // refused (pass 2: by the first Cholesky): nothing is written
    double* Xc = sm;
    double* Qall = sm + (MULT ? CS_XC : 0);                   // the workgroup's 64 x w block, rows of wave v at 16 v
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), l15 = lane & 15, l4 = lane >> 4;
    double* Qt = Qall + wave * 16 * CS_QLD;
    const int nct = w >> 4;
    if (MULT) {
        for (int e = tid; e < w * CQ_W; e += CS_THREADS) {
            const int k = e >> 7, j = e & (CQ_W - 1);
            if (j < w && j >= (k & ~15)) Xc[cs_blk(k >> 4, j >> 4) + (k & 15) * 16 + (j & 15)] = X[k * CQ_W + j];
        }
        __syncthreads();
    }
    // GRAM: a wave owns the tile rows `wave` and 7 - wave of the Gram matrix (nine tiles on and above the diagonal at w = 128) over ALL
    // 64 rows of the workgroup's block: 72 accumulator registers instead of the 288 of a whole Gram matrix per wave
    v4d g0[8], g1[8];
    if (GRAM)
#pragma unroll
        for (int t = 0; t < 8; ++t) { g0[t] = (v4d){0.0, 0.0, 0.0, 0.0}; g1[t] = g0[t]; }
    const int ntile = (mk + 15) >> 4, nblk = (ntile + 3) >> 2;
    double q[32], qn[32];
    // unconditional, clamped loads: no select may consume a loaded register before its use (see cqr_gram_kernel).  Rows beyond the panel
    // then hold copies of its last row: their products are never stored and their rows of the LDS block are zeroed (`rin`)
    auto load = [&](double (&x)[32], int t) {
        const int row = 16 * t + l15;
        const double* sp = src + (row < mk ? row : mk - 1);
#pragma unroll
        for (int ks = 0; ks < 32; ++ks) x[ks] = sp[(size_t) (4 * ks + l4 < w ? 4 * ks + l4 : 0) * lds_];
    };
    // Tile order: tile = 4 blk + wave, blocks strided over the workgroups (the four waves of a workgroup on one 64-row block: the Gram
    // phase needs that, and the chip then reads ONE moving window of the panel).  The pass without a Gram phase has independent waves and
    // deals its LAST, partial round out by tiles, wave-major (tile = base + wave nbid + bid): the round is then short by tiles, not by
    // whole blocks -- which is what lets cqr_vpass_kernel give two of the 256 compute units to its riders (254 streaming workgroups:
    // 16.13 rounds; by blocks 32 workgroups would run a 17th block, +6 %; by tiles +1.5 %)
    const int tstride = 4 * nbid;
    const int tend = GRAM ? 4 * nblk : ntile;                 // (GRAM: every wave of a workgroup makes the same number of trips: barriers inside)
    const int nfull = GRAM ? 0x7fffffff / (tstride + 1) : ntile / tstride;
    auto tile_at = [&](int k) { return k < nfull ? k * tstride + 4 * bid + wave : (k == nfull ? nfull * tstride + wave * nbid + bid : tend); };
    int kround = 0, tile = tile_at(0);
    if (tile < tend) load(q, tile);
    for (int tnext; tile < tend; tile = tnext) {
        tnext = tile_at(++kround);
        // the next block's rows are requested before this block's matrix-core work: with GRAM into q itself once the block has gone to
        // LDS / through the product (the Gram instructions cover the latency, and a second buffer would cost the second wave per SIMD)
        const int row = 16 * tile + l15;
        const bool rin = row < mk;
        if (MULT) {
            // (one column tile at a time: with the k-step outermost -- eight interleaved accumulator chains -- the stores came in one burst
            // at the end and the pass was 10 % slower; tiles in pairs (jt, 7 - jt), two chains: 271 against 240 us.)
            // Round 5: the vector-memory instructions are dealt out BETWEEN the matrix-core instructions, in a pinned order: the stores
            // of tile jt - 1 one per MFMA from the third MFMA of tile jt on (its results are then two MFMAs old: no wait states to pad),
            // and -- the pass without a Gram phase -- the next block's loads one per four MFMAs.  In a burst they fill the CU's
            // vector-memory queue, the wave waits AT the instruction and the MFMAs behind it wait too (in-order issue).
            const int tn = tnext, rown = 16 * tn + l15;
            const double* spn = src + (rown < mk ? rown : mk - 1);
            auto store_reg = [&](const v4d& acc, int jt, int r) __attribute__((always_inline)) {
                const int colj = 16 * jt + l4 + 4 * r;
                if (rin && row >= skip) {                      // (skip: the first rows belong to the riders of cqr_vpass_kernel)
                    dst[row + (size_t) colj * ldd] = acc[r];
                    if (DST2) dst2[row + (size_t) colj * ldd2] = acc[r];
                }
                if (GRAM) Qt[l15 * CS_QLD + colj] = rin ? acc[r] : 0.0;
            };
            // The multiplication by the inverse of the upper-triangular factor (R1 in pass 2, U' in pass 3) as a BLOCK back substitution
            // over 32-column blocks (end of round 5): X holds the factor's off-diagonal 32 x 32 blocks and the INVERSES of its diagonal
            // blocks -- which the matrix-core diagonal steps of the Cholesky / the LU leave behind anyway -- so that nobody has to form the
            // full inverse between the factorisation and this pass (cqr_ui_kernel, 30 us, and 18 us of cqr_chol_kernel were exactly that):
            //     T_b = q_b - sum_{i < b} V_i X_ib   (accumulators initialised with the block's source registers: the accumulator layout of
            //                                         the transposed product IS the operand layout, register r of tile t <-> q[4 t + r])
            //     V_b = T_b X_bb
            // 16 b + 12 MFMAs for block b: 144 per 16 rows, as for the product with the explicit inverse.  The X operand of MFMA n (n counts
            // them in order, cs_s_dec) is read from LDS four MFMAs ahead into a ring of eight registers: with the order pinned MFMA by MFMA
            // the compiler cannot hoist the read itself, and a read in front of its own MFMA costs the LDS latency per instruction.
            // fullw: 128 columns, no test around an instruction.
            auto product = [&](auto fullw) {
                constexpr bool FW = decltype(fullw)::value;
                auto xop = [&](int n) __attribute__((always_inline)) { return Xc[cs_blk(cs_s_dec(n) & 15, (cs_s_dec(n) >> 4) & 15) + (4 * (cs_s_dec(n) >> 8) + l4) * 16 + l15]; };
                double xr[8];
#pragma unroll
                for (int n = 0; n < 4; ++n) xr[n] = xop(n);
                v4d V[8];
                // behind MFMA n: one store of the previous block's two tiles (MFMAs 2 .. 9 of a block); in the pass without a Gram phase
                // one load of the next 16 rows per four MFMAs.  (n is spelled out from the loop indices: as a running counter it stayed a
                // run-time value, and every array it indexes -- the ring, V, q -- went to scratch: 1.2 ms per pass)
                auto behind = [&](int b, int n, bool on) __attribute__((always_inline)) {      // (not inlined, it takes n at run time and every array it captures lives in scratch)
                    const int m = n - cs_s_off(b);
                    if (on && b > 0 && m >= 2 && m < 10) store_reg(V[2 * (b - 1) + ((m - 2) >> 2)], 2 * (b - 1) + ((m - 2) >> 2), (m - 2) & 3);
                    if (!GRAM && (n & 3) == 3 && (n >> 2) < 32) {
                        const int i = n >> 2;
                        qn[i] = spn[(size_t) (4 * i + l4 < w ? 4 * i + l4 : 0) * lds_];
                    }
                    __builtin_amdgcn_sched_barrier(0);
                };
                // (one instantiation per block, b a compile-time constant: as a loop over b the compiler kept the loop -- dynamic register
                // indexing, the operand table decoded with scalar instructions at run time, every array in scratch: 1.2 ms per pass)
                auto block = [&](auto bc) __attribute__((always_inline)) {
                    constexpr int b = decltype(bc)::value;
                    const bool on = FW || 2 * b < nct;
                    v4d T0, T1;
#pragma unroll
                    for (int r = 0; r < 4; ++r) { T0[r] = q[8 * b + r]; T1[r] = q[8 * b + 4 + r]; }
#pragma unroll
                    for (int h = 0; h < 2; ++h)
#pragma unroll
                        for (int it = 0; it < 2 * b; ++it)
#pragma unroll
                            for (int r = 0; r < 4; ++r) {
                                const int n = cs_s_off(b) + 8 * b * h + 4 * it + r;
                                if (n + 4 < 144) xr[(n + 4) & 7] = xop(n + 4);
                                if (on) {
                                    if (h) T1 = __builtin_amdgcn_mfma_f64_16x16x4f64(xr[n & 7], V[it][r], T1, 0, 0, 1);      // T - x v
                                    else T0 = __builtin_amdgcn_mfma_f64_16x16x4f64(xr[n & 7], V[it][r], T0, 0, 0, 1);
                                }
                                behind(b, n, on);
                            }
                    v4d o0 = (v4d){0.0, 0.0, 0.0, 0.0}, o1 = o0;
#pragma unroll
                    for (int d = 0; d < 12; ++d) {
                        const int n = cs_s_off(b) + 16 * b + d;
                        if (n + 4 < 144) xr[(n + 4) & 7] = xop(n + 4);
                        if (on) {
                            if (d < 4) o0 = __builtin_amdgcn_mfma_f64_16x16x4f64(xr[n & 7], T0[d], o0, 0, 0, 0);
                            else if (d < 8) o1 = __builtin_amdgcn_mfma_f64_16x16x4f64(xr[n & 7], T0[d - 4], o1, 0, 0, 0);
                            else o1 = __builtin_amdgcn_mfma_f64_16x16x4f64(xr[n & 7], T1[d - 8], o1, 0, 0, 0);
                        }
                        behind(b, n, on);
                    }
                    if (on) { V[2 * b] = o0; V[2 * b + 1] = o1; }
                    // the last active block's two tiles go out at once (the others ride in the next block).  (Indexed by b, not by nct: a
                    // run-time index into V sends the whole array to scratch)
                    if (b == 3 ? on : (!FW && on && 2 * (b + 1) == nct)) {
#pragma unroll
                        for (int r = 0; r < 4; ++r) { store_reg(o0, 2 * b, r); store_reg(o1, 2 * b + 1, r); }
                    }
                };
                block(cs_int<0>{}); block(cs_int<1>{}); block(cs_int<2>{}); block(cs_int<3>{});
            };
            if (nct == 8) product(cs_true{}); else product(cs_false{});
        } else if (GRAM) {
#pragma unroll
            for (int ks = 0; ks < 32; ++ks)
                if (4 * ks < w) Qt[l15 * CS_QLD + 4 * ks + l4] = rin ? q[ks] : 0.0;
        }
        if (GRAM) {
            // the next block's rows: one load per k-step of the Gram instructions (32 k-steps per wave and block)
            const int tn = tnext, rown = 16 * tn + l15;
            const double* spn = src + (rown < mk ? rown : mk - 1);
            auto load1 = [&](int ks) { q[ks] = spn[(size_t) (4 * ks + l4 < w ? 4 * ks + l4 : 0) * lds_]; };
            cs_lds_barrier();
            if (nct == 8) cs_gram_rows<true, true>(g0, g1, Qall, nct, wave, l15, l4, load1);
            else cs_gram_rows<false, true>(g0, g1, Qall, nct, wave, l15, l4, load1);
            cs_lds_barrier();                                 // the block is free for the next one's writes
        }
        if (!GRAM)
#pragma unroll
            for (int ks = 0; ks < 32; ++ks) q[ks] = qn[ks];
    }
    if (GRAM) {
        double* out = slabs + (size_t) bid * 36 * 256;
#pragma unroll
        for (int half = 0; half < 2; ++half) {
            const int tr = half ? 7 - wave : wave;
#pragma unroll
            for (int t = 0; t < 8; ++t)
                if (tr < nct && t >= tr && t < nct)
#pragma unroll
                    for (int r = 0; r < 4; ++r) out[(t * (t + 1) / 2 + tr) * 256 + r * 64 + lane] = half ? g1[t][r] : g0[t][r];
        }
    }
}

template <bool MULT, bool GRAM, bool DST2>
__global__ __launch_bounds__(CS_THREADS) void cqr_stream_kernel(const double* __restrict__ X, int w, int mk, const double* src, int lds_, double* dst,
                                                                 int ldd, double* dst2, int ldd2, double* slabs, const int* status)
{
    extern __shared__ double sm[];
    cqr_stream_body<MULT, GRAM, DST2>(sm, X, w, mk, src, lds_, dst, ldd, dst2, ldd2, slabs, status, blockIdx.x, gridDim.x);
}
// The last pass (V = (Q - [S R2; 0]) U'^-1 -> Vw and A) with T and R riding along: workgroups 0 and 1 run cq_post_t / cq_post_r (one-workgroup
// work that nothing in this launch waits for: cqr_top_kernel, behind it, reads T and R), the others stream.  LDS: the larger of the two.
// DST2 = false: V goes to ONE destination -- the "parked" form (qrd_panel_cqr_p): the caller's array takes the whole of V, its top block
// included (cqr_top_kernel<true>), the updates that follow read V from there, and R -- which stays in the workspace -- is put back
// by cqr_restore_r_kernel afterwards: one 8 mk w byte write of the pass less.
template <bool DST2>
__global__ __launch_bounds__(CS_THREADS) void cqr_vpass_kernel(double* ws, int w, int mk, const double* src, int lds_, double* dst, int ldd, double* dst2,
                                                                int ldd2, const int* status, const CqTop top)
{
    extern __shared__ double sm[];
    if (status[0]) return;
    if (blockIdx.x < 2) {
        const CqLds L = cq_lds(sm);
        const int tid = threadIdx.x;
        if (tid < w) L.sv[tid] = ws[CQ_SV + tid];
        if (blockIdx.x == 0) cq_post_t(L, ws, w, tid, top);
        else { __syncthreads(); cq_post_r(L, ws, w, tid, top); }
        return;
    }
    cqr_stream_body<true, false, DST2>(sm, ws + CQ_UI, w, mk, src, lds_, dst, ldd, dst2, ldd2, nullptr, status, blockIdx.x - 2, gridDim.x - 2, top.mode ? w : 0);
}
// The Gram-only pass, G1 = A^T A (33 KB of LDS).  Tried and no faster: two waves per SIMD (132 us against 128), four workgroups per CU
// (191 = 191 before the scalar wave index), two LDS block buffers with one barrier per block and the transposition of block i + 1
// behind the matrix-core instructions of block i (143-145 us: memory time plus matrix-core time again, although the ISA has the waits
// for the prefetched block behind the MFMAs)
// (status: the panel's refusal word starts at zero -- cleared here, by the panel's first launch, instead of by a memset node in front of it)
// (reset = 0: the Gram pass of a RETRIED panel -- the shifted Cholesky in front of it has cleared the word, and a failure of that launch
// must stand: this pass then returns at once like every other launch of the panel)
__global__ __launch_bounds__(CS_THREADS) void cqr_gram_kernel(int w, int mk, const double* src, int lds_, double* slabs, int* status, int reset)
{
    extern __shared__ double sm[];
    if (reset) { if (blockIdx.x == 0 && threadIdx.x == 0) status[0] = 0; }
    else if (status[0]) return;
    cqr_stream_body<false, true, false>(sm, nullptr, w, mk, src, lds_, nullptr, 0, nullptr, 0, slabs, nullptr, blockIdx.x, gridDim.x);
}

// G (column-major ld CQ_W, both triangles) = sum over the workgroup partials, in slab order.  Tile t = (ti <= tj), accumulator
// register r of lane l: element (16 ti + (l >> 4) + 4 r, 16 tj + (l & 15)).  Grid: (tiles, 8 chunks of 32 elements).
__global__ __launch_bounds__(256) void cqr_gram_reduce_kernel(const double* __restrict__ slabs, int nslab, double* G)
{
    __shared__ double part[8][33];
    const int t = blockIdx.x, chunk = blockIdx.y, tid = threadIdx.x, el = tid & 31, grp = tid >> 5;
    const int e = chunk * 32 + el;
    // fixed order per group; the (up to) 32 partials of a thread are requested together: one after the other, each add waited for its own
    // load and the launch was 32 memory latencies long (11.7 us for 19 MB)
    double s = 0.0;
    for (int q0 = grp; q0 < nslab; q0 += 8 * 32) {
        double x[32];
#pragma unroll
        for (int u = 0; u < 32; ++u) { const int q = q0 + 8 * u; x[u] = slabs[(size_t) (q < nslab ? q : grp) * 36 * 256 + t * 256 + e]; }
#pragma unroll
        for (int u = 0; u < 32; ++u) if (q0 + 8 * u < nslab) s += x[u];
    }
    part[grp][el] = s;
    __syncthreads();
    if (grp == 0) {
        double v = 0.0;
#pragma unroll
        for (int q2 = 0; q2 < 8; ++q2) v += part[q2][el];
        int tj = 0;
        while ((tj + 1) * (tj + 2) / 2 <= t) ++tj;
        const int ti = t - tj * (tj + 1) / 2;
        const int r = e >> 6, l = e & 63, i = 16 * ti + (l >> 4) + 4 * r, j = 16 * tj + (l & 15);
        G[i + CQ_W * j] = v;
        G[j + CQ_W * i] = v;
    }
}

// the top block after the last pass: A <- R on and above the diagonal, L1 below; Vw <- unit lower L1; T, tau.
// PARK: A <- the unit lower L1 itself (zero above its diagonal: the whole of V now sits in A, ready to be a GEMM operand); Vw untouched;
// R waits in the workspace for cqr_restore_r_kernel
template <bool PARK>
__global__ __launch_bounds__(256) void cqr_top_kernel(const double* ws, int w, double* A, int lda, double* Vw, int ldv, double* T, int ldt,
                                                       double* tau, const int* status)
{
    if (status[0]) return;
    for (int e = threadIdx.x + blockIdx.x * blockDim.x; e < w * w; e += blockDim.x * gridDim.x) {
        const int j = e / w, i = e - j * w;                   // consecutive threads: consecutive rows of a column
        const double lu = ws[CQ_LU + i * CQ_W + j];
        if (PARK) A[i + (size_t) lda * j] = (j < i) ? lu : (j == i ? 1.0 : 0.0);
        else {
            A[i + (size_t) lda * j] = (j >= i) ? ws[CQ_RR + i * CQ_W + j] : lu;
            Vw[i + (size_t) ldv * j] = (j < i) ? lu : (j == i ? 1.0 : 0.0);
        }
        const double tv = (j >= i) ? ws[CQ_TT + i * CQ_W + j] : 0.0;      // (tiles below the tile diagonal are never written)
        T[i + (size_t) ldt * j] = tv;
        if (i == j) tau[i] = tv;
    }
}
// R (upper triangle, from the workspace) into a column-major w x w block: the top block of A after a parked panel's updates (lower part
// kept: L1), or -- zero_below -- a block of its own (the multi-GPU step packs R before the update has run)
// (status: the panel's refusal word -- a refused panel left the array untouched, and so does this launch; NULL: not consulted)
__global__ __launch_bounds__(256) void cqr_restore_r_kernel(const double* ws, int w, double* D, int ldd, int zero_below, const int* status)
{
    if (status && status[0]) return;
    for (int e = threadIdx.x + blockIdx.x * blockDim.x; e < w * w; e += blockDim.x * gridDim.x) {
        const int j = e / w, i = e - j * w;
        if (j >= i) D[i + (size_t) ldd * j] = ws[CQ_RR + i * CQ_W + j];
        else if (zero_below) D[i + (size_t) ldd * j] = 0.0;
    }
}
}   // namespace

extern "C" {

size_t qrd_panel_cqr_ws_doubles(void) { return (size_t) CQ_WS; }

int qrd_panel_cqr_init(void)
{
    hipError_t e = hipFuncSetAttribute((const void*) cqr_chol_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int) CQ_LDS_BYTES);
    if (e == hipSuccess) e = hipFuncSetAttribute((const void*) cqr_lu_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int) CQ_LDS_BYTES);
    if (e == hipSuccess) e = hipFuncSetAttribute((const void*) cqr_post_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int) CQ_LDS_BYTES);
    if (e == hipSuccess) e = hipFuncSetAttribute((const void*) cqr_rmul_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int) CQ_LDS_BYTES);
    if (e == hipSuccess) e = hipFuncSetAttribute((const void*) cqr_vpass_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int) CQ_VP_LDS_BYTES);
    if (e == hipSuccess) e = hipFuncSetAttribute((const void*) cqr_vpass_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int) CQ_VP_LDS_BYTES);
    if (e == hipSuccess) e = hipFuncSetAttribute((const void*) cqr_gram_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int) CS_LDS_GRAM);
    if (e == hipSuccess) e = hipFuncSetAttribute((const void*) cqr_stream_kernel<true, true, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int) CS_LDS_BYTES);
    if (e == hipSuccess) e = hipFuncSetAttribute((const void*) cqr_stream_kernel<true, false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int) CS_LDS_BYTES);
    if (e == hipSuccess) e = hipFuncSetAttribute((const void*) cqr_stream_kernel<true, false, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int) CS_LDS_BYTES);
    return (int) e;
}

// shapes this route takes: w a multiple of 32 up to 128, at least 2 w rows
int qrd_panel_cqr_ok(int mk, int w) { return w >= 32 && w <= CQ_W && (w & 31) == 0 && mk >= 2 * w; }

static int cs_grid(int mk, int cap = CS_NWG) { const int g = (mk + 63) / 64; return g < cap ? g : cap; }

// The whole panel: six launches + two small reductions on `stream`.  status (device, 4 ints): [0] is zeroed here and is 1 afterwards when the
// guard refused the panel -- A is then untouched, and so is Vw when Q has a buffer of its own (Qb: mk x w, ld ldq; NULL = Q lives in
// Vw, whose contents are then garbage after a refusal); [1] counts refused panels (never reset here).  hflag / seq: see cqr_lu_kernel.
static int panel_cqr_impl(void* stream, double* A, int lda, int mk, int w, double* tau, double* T, int ldt, double* Vw, int ldv, double* ws, int* status,
                          double* Qb, int ldq, unsigned* hflag, unsigned seq, int park, int retry = 0);
int qrd_panel_cqr_q(void* stream, double* A, int lda, int mk, int w, double* tau, double* T, int ldt, double* Vw, int ldv, double* ws, int* status,
                    double* Qb, int ldq, unsigned* hflag, unsigned seq)
{
    return panel_cqr_impl(stream, A, lda, mk, w, tau, T, ldt, Vw, ldv, ws, status, Qb, ldq, hflag, seq, 0);
}
// park != 0 ("parked" form; Qb must be a buffer other than A): V is written ONCE, into A -- all of it, the top block as the unit lower
// triangle with zeros above -- and Vw is not touched; R stays in the workspace until qrd_panel_cqr_restore_r puts it into the top block
// (after the updates that use A's panel as V; before the next panel of this workspace).  qrd_panel_cqr_r_block: R as a block of its own.
// park == 2: V once, into A, and R into the top block at once (the last panel of a factorisation: no update reads V).
int qrd_panel_cqr_p(void* stream, double* A, int lda, int mk, int w, double* tau, double* T, int ldt, double* Vw, int ldv, double* ws, int* status,
                    double* Qb, int ldq, unsigned* hflag, unsigned seq, int park)
{
    if (park && (!Qb || Qb == A)) return -7;
    return panel_cqr_impl(stream, A, lda, mk, w, tau, T, ldt, Vw, ldv, ws, status, Qb, ldq, hflag, seq, park);
}
int qrd_panel_cqr_restore_r(void* stream, double* A, int lda, int w, const double* ws, const int* status)
{
    hipLaunchKernelGGL(cqr_restore_r_kernel, dim3((w * w + 255) / 256), dim3(256), 0, (hipStream_t) stream, ws, w, A, lda, 0, status);
    return (int) hipGetLastError();
}
int qrd_panel_cqr_r_block(void* stream, const double* ws, int w, double* D, int ldd)
{
    hipLaunchKernelGGL(cqr_restore_r_kernel, dim3((w * w + 255) / 256), dim3(256), 0, (hipStream_t) stream, ws, w, D, ldd, 1, (const int*) nullptr);
    return (int) hipGetLastError();
}
// The retry of a panel the guard has just refused (same arguments as the refused call; Qb a buffer other than A and Vw; the workspace
// still holds that call's G1): SHIFTED CholeskyQR3.  R0 = chol(G1 + s I) always exists; Q0 = A R0^-1 (one more pass over the panel, into
// Qb) has condition <= ~1 / sqrt(11 m n u) ~ 5e3 unless the panel is numerically rank deficient; the ordinary pipeline then factors Q0
// in place of A (Gram pass, Cholesky, Q + G2 pass, reconstruction, last pass -- with its own guard) and R comes out as S R2 R1 R0
// (cqr_rmul_kernel).  A refused panel thus costs one failed attempt + 1.3 panels instead of four guarded leaf chains (2.2 ms -> ~1.0 ms at
// 262144 x 128, profiles/r06_guard_price.txt); a panel that is refused AGAIN (rank deficient, cond > ~1e10) goes to the Householder leaf
// chain as before.  A is untouched until the last pass, as in the first attempt.
int qrd_panel_cqr_retry(void* stream, double* A, int lda, int mk, int w, double* tau, double* T, int ldt, double* Vw, int ldv, double* ws, int* status,
                        double* Qb, int ldq, unsigned* hflag, unsigned seq, int park)
{
    if (!Qb || Qb == A || Qb == Vw) return -7;
    return panel_cqr_impl(stream, A, lda, mk, w, tau, T, ldt, Vw, ldv, ws, status, Qb, ldq, hflag, seq, park, 1);
}
static int panel_cqr_impl(void* stream, double* A, int lda, int mk, int w, double* tau, double* T, int ldt, double* Vw, int ldv, double* ws, int* status,
                          double* Qb, int ldq, unsigned* hflag, unsigned seq, int park, int retry)
{
    if (!qrd_panel_cqr_ok(mk, w)) return -7;
    if (!Qb) { Qb = Vw; ldq = ldv; }
    hipStream_t s = (hipStream_t) stream;
    // one persistent workgroup per compute unit OF THIS STREAM (a CU-masked stream -- MI355XQR_TSQR_RESERVE_CUS, a panel partition -- would
    // otherwise run the workgroups beyond its mask as a second round behind the first: twice the pass)
    int cap = qrd_stream_cus(stream);
    if (cap <= 0 || cap > CS_NWG) cap = CS_NWG;
    const int grid = cs_grid(mk, cap), ggrid = cs_grid(mk, cap), ntl = (w >> 4) * ((w >> 4) + 1) / 2;
    const double* src = A;                                    // what the pipeline factors: the panel, or (retry) its preconditioned image in Qb
    int lsrc = lda;
    if (retry) {
        const double u = 1.1102230246251565e-16;
        const double shift_scale = 11.0 * ((double) mk * w + (double) w * (w + 1)) * u;
        hipLaunchKernelGGL(cqr_chol_kernel, dim3(1), dim3(CQ_T), CQ_LDS_BYTES, s, ws, w, status, shift_scale);
        hipLaunchKernelGGL((cqr_stream_kernel<true, false, false>), dim3(grid), dim3(CS_THREADS), CS_LDS_BYTES, s, ws + CQ_R1I, w, mk, (const double*) A, lda,
                           Qb, ldq, (double*) nullptr, 0, (double*) nullptr, status);
        src = Qb; lsrc = ldq;
    }
    hipLaunchKernelGGL(cqr_gram_kernel, dim3(ggrid), dim3(CS_THREADS), CS_LDS_GRAM, s, w, mk, src, lsrc, ws + CQ_SL, status, retry ? 0 : 1);
    hipLaunchKernelGGL(cqr_gram_reduce_kernel, dim3(ntl, 8), dim3(256), 0, s, ws + CQ_SL, ggrid, ws + CQ_G1);
    hipLaunchKernelGGL(cqr_chol_kernel, dim3(1), dim3(CQ_T), CQ_LDS_BYTES, s, ws, w, status, 0.0);
    if (retry) hipLaunchKernelGGL(cqr_rmul_kernel, dim3(1), dim3(CQ_T), CQ_LDS_BYTES, s, ws, w, (const int*) status);
    hipLaunchKernelGGL((cqr_stream_kernel<true, true, false>), dim3(grid), dim3(CS_THREADS), CS_LDS_BYTES, s, ws + CQ_R1I, w, mk, src, lsrc, Qb, ldq,
                       (double*) nullptr, 0, ws + CQ_SL, status);
    hipLaunchKernelGGL(cqr_gram_reduce_kernel, dim3(ntl, 8), dim3(256), 0, s, ws + CQ_SL, grid, ws + CQ_G2);
    hipLaunchKernelGGL(cqr_lu_kernel, dim3(1), dim3(CQ_T), CQ_LDS_BYTES, s, ws, w, Qb, ldq, status, hflag, seq);
    const int vgrid = grid + 2 <= cap ? grid + 2 : (cap >= 3 ? cap : 3);      // riders included: never more workgroups than compute units (and at least one streaming workgroup)
    // the riders of the last pass also place the top block, T and tau (cqr_top_kernel: stage-by-stage entry points only)
    if (park) {
        // park == 1: A <- the unit lower L1, R parked in the workspace; park == 2: nobody reads V behind this panel -- R into the top block
        // at once (Vw gets its w x w corner and is otherwise stale), nothing to restore
        const CqTop top{A, lda, Vw, ldv, T, ldt, tau, park == 2 ? 1 : 2};
        hipLaunchKernelGGL(cqr_vpass_kernel<false>, dim3(vgrid), dim3(CS_THREADS), CQ_VP_LDS_BYTES, s, ws, w, mk, (const double*) Qb, ldq, A, lda,
                           (double*) nullptr, 0, (const int*) status, top);
    } else {
        const CqTop top{A, lda, Vw, ldv, T, ldt, tau, 1};
        hipLaunchKernelGGL(cqr_vpass_kernel<true>, dim3(vgrid), dim3(CS_THREADS), CQ_VP_LDS_BYTES, s, ws, w, mk, (const double*) Qb, ldq, Vw, ldv, A, lda,
                           (const int*) status, top);
    }
    return (int) hipGetLastError();
}

int qrd_panel_cqr(void* stream, double* A, int lda, int mk, int w, double* tau, double* T, int ldt, double* Vw, int ldv, double* ws, int* status)
{
    return qrd_panel_cqr_q(stream, A, lda, mk, w, tau, T, ldt, Vw, ldv, ws, status, nullptr, 0, nullptr, 0u);
}

// the two halves with the Gram matrices supplied by the caller (G1 in qrd_panel_cqr_g1(ws) before stage 1, G2 = Q^T Q in
// qrd_panel_cqr_g2(ws) before stage 2; column-major ld 128): kept for the stage-by-stage checks of devtools/tools_cqr_debug.py
int qrd_panel_cqr_stage1(void* stream, const double* A, int lda, int mk, int w, double* Vw, int ldv, double* ws, int* status)
{
    hipStream_t s = (hipStream_t) stream;
    hipLaunchKernelGGL(cqr_chol_kernel, dim3(1), dim3(CQ_T), CQ_LDS_BYTES, s, ws, w, status, 0.0);
    hipLaunchKernelGGL((cqr_stream_kernel<true, false, false>), dim3(cs_grid(mk)), dim3(CS_THREADS), CS_LDS_BYTES, s, ws + CQ_R1I, w, mk, A, lda, Vw, ldv,
                       (double*) nullptr, 0, (double*) nullptr, status);
    return (int) hipGetLastError();
}

int qrd_panel_cqr_stage2(void* stream, double* A, int lda, int mk, int w, double* tau, double* T, int ldt, double* Vw, int ldv, double* ws,
                         int* status)
{
    hipStream_t s = (hipStream_t) stream;
    hipLaunchKernelGGL(cqr_lu_kernel, dim3(1), dim3(CQ_T), CQ_LDS_BYTES, s, ws, w, Vw, ldv, status, (unsigned*) nullptr, 0u);
    hipLaunchKernelGGL(cqr_post_kernel, dim3(2), dim3(CQ_T), CQ_LDS_BYTES, s, ws, w, (const int*) status);
    hipLaunchKernelGGL((cqr_stream_kernel<true, false, true>), dim3(cs_grid(mk)), dim3(CS_THREADS), CS_LDS_BYTES, s, ws + CQ_UI, w, mk, Vw, ldv, Vw, ldv, A, lda,
                       (double*) nullptr, status);
    hipLaunchKernelGGL(cqr_top_kernel<false>, dim3((w * w + 255) / 256), dim3(256), 0, s, ws, w, A, lda, Vw, ldv, T, ldt, tau, status);
    return (int) hipGetLastError();
}

double* qrd_panel_cqr_g1(double* ws) { return ws + CQ_G1; }
double* qrd_panel_cqr_g2(double* ws) { return ws + CQ_G2; }

}   // extern "C"
