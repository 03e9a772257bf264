// qr_gemm_tile.h -- the LDS-staged MFMA f64 tile machinery shared by the GEMM kernels (qr_kernels.hip) and the fused leaf
// kernels (qr_panel_tsqr.hip: the leaf's long-K product runs in the same launch as its one-workgroup reconstruction).
//
// MFMA f64 16x16x4 lane maps (cdna_hip_programming.md section 3):
//   A-operand: lane l holds Aop[p = l&15][k = l>>4];  B-operand: lane l holds Bop[k = l>>4][q = l&15]
//   D: lane l, reg r holds D[p = (l>>4) + 4r][q = l&15].
// The COLUMN index of the (column-major) output is always on p and the ROW index on q, so that 16 consecutive lanes touch 16
// consecutive rows of one output column (128 contiguous bytes).
#ifndef QR_GEMM_TILE_H
#define QR_GEMM_TILE_H
#include "qr_common.h"

// ------------------------------------------------------------------------------------------------
// GEMM tiles.  Block = 256 threads = 4 waves arranged 2 (rows) x 2 (cols); each wave owns
// (16*TI) x (16*TJ) of the output as TI*TJ MFMA accumulators; block tile = (32*TI) x (32*TJ), BK = 16.
// LDS images (double units):
//   "row-fast"  image [BK][32*TI + 16]  (NN left operand: k-major, rows contiguous; +16 pad makes the two
//                k-rows a half-wave reads land on disjoint bank halves -> conflict-free ds_read_b64)
//   "k-fast"    image [cols][BK + 2]    (operands whose k runs contiguously in memory; the stride 18
//                = 2*odd spreads 16 columns x 2 k's over all 32 eight-byte banks -> conflict-free)
// ------------------------------------------------------------------------------------------------
#define BK 16
#define LDKF (BK + 2)

// Tile loaders.  `fast` is block-uniform (whole tile in range, 16-byte aligned): the fast path is
// straight-line 16-byte loads.  The edge path uses clamped addresses + selects, never a branch per
// element: hipcc waits (s_waitcnt vmcnt(0)) inside every divergent branch that consumes a load, which
// would turn one tile fetch into dozens of serial HBM round trips.
template <int TR>   // TR = tile extent / 32 (rows of the row-fast image)
__device__ __forceinline__ void load_rowfast(v2d (&reg)[TR], const double* __restrict__ A, int lda,
                                             int i0, int k0, int M, int kend, bool fast, int tid)
{
    constexpr int HALF = 16 * TR;          // double2 per column
    if (fast) {
#pragma unroll
        for (int q = 0; q < TR; ++q) {
            const int idx = tid + 256 * q;
            reg[q] = *reinterpret_cast<const v2d*>(A + (size_t) (k0 + idx / HALF) * lda + i0 + 2 * (idx % HALF));
        }
    } else {
#pragma unroll
        for (int q = 0; q < TR; ++q) {
            const int idx = tid + 256 * q;
            const int i = i0 + 2 * (idx % HALF), k = k0 + idx / HALF;
            const double* col = A + (size_t) min(k, kend - 1) * lda;
            const double a = col[min(i, M - 1)], b = col[min(i + 1, M - 1)];
            reg[q] = (v2d){(k < kend && i < M) ? a : 0.0, (k < kend && i + 1 < M) ? b : 0.0};
        }
    }
}

template <int TR>
__device__ __forceinline__ void store_rowfast(const v2d (&reg)[TR], double* __restrict__ S, int tid)
{
    constexpr int HALF = 16 * TR, LD = 32 * TR + 16;
#pragma unroll
    for (int q = 0; q < TR; ++q) {
        const int idx = tid + 256 * q;
        const int col = idx / HALF, r2 = idx % HALF;
        *reinterpret_cast<v2d*>(S + col * LD + 2 * r2) = reg[q];
    }
}

template <int TC>   // TC = tile extent / 32 (columns of the k-fast image)
__device__ __forceinline__ void load_kfast(v2d (&reg)[TC], const double* __restrict__ B, int ldb,
                                           int j0, int k0, int N, int kend, bool fast, int tid)
{
    if (fast) {
#pragma unroll
        for (int q = 0; q < TC; ++q) {
            const int idx = tid + 256 * q;
            reg[q] = *reinterpret_cast<const v2d*>(B + (size_t) (j0 + idx / 8) * ldb + k0 + 2 * (idx % 8));
        }
    } else {
#pragma unroll
        for (int q = 0; q < TC; ++q) {
            const int idx = tid + 256 * q;
            const int j = j0 + idx / 8, k = k0 + 2 * (idx % 8);
            const double* col = B + (size_t) min(j, N - 1) * ldb;
            const double a = col[min(k, kend - 1)], b = col[min(k + 1, kend - 1)];
            reg[q] = (v2d){(j < N && k < kend) ? a : 0.0, (j < N && k + 1 < kend) ? b : 0.0};
        }
    }
}

template <int TC>
__device__ __forceinline__ void store_kfast(const v2d (&reg)[TC], double* __restrict__ S, int tid)
{
#pragma unroll
    for (int q = 0; q < TC; ++q) {
        const int idx = tid + 256 * q;
        *reinterpret_cast<v2d*>(S + (idx / 8) * LDKF + 2 * (idx % 8)) = reg[q];
    }
}

// ---- shared GEMM pieces -----------------------------------------------------------------------
// One BK-deep step of the wave tile from LDS.  AROW: the row operand comes from a row-fast image
// (NN kernel), else from a k-fast image (TN kernel).  The column operand image is always k-fast.
// PRIO: s_setprio around the MFMA burst.  Measured on the C3 update shapes: +3 % for the TN kernel, -2 % for the NN kernel,
// so only the k-fast (TN) instantiation raises its priority.
template <int TI, int TJ, bool AROW, bool PRIO = !AROW>
__device__ __forceinline__ void mma_tile(v4d (&acc)[TJ][TI], const double* __restrict__ as,
                                         const double* __restrict__ bs, int wi, int wj, int l15, int l4)
{
    constexpr int LA = 32 * TI + 16;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
        const int kk = 4 * ks + l4;
        double rowv[TI], colv[TJ];
#pragma unroll
        for (int b = 0; b < TI; ++b)
            rowv[b] = AROW ? as[kk * LA + wi * 16 * TI + 16 * b + l15] : as[(wi * 16 * TI + 16 * b + l15) * LDKF + kk];
#pragma unroll
        for (int a = 0; a < TJ; ++a) colv[a] = bs[(wj * 16 * TJ + 16 * a + l15) * LDKF + kk];
        if (PRIO) __builtin_amdgcn_s_setprio(1);      // keep the MFMA pipe for the wave that has its fragments
#pragma unroll
        for (int a = 0; a < TJ; ++a)
#pragma unroll
            for (int b = 0; b < TI; ++b)
                acc[a][b] = __builtin_amdgcn_mfma_f64_16x16x4f64(colv[a], rowv[b], acc[a][b], 0, 0, 0);
        if (PRIO) __builtin_amdgcn_s_setprio(0);
    }
}

// Double-buffered K loop over [kbeg, kend).  FAST is a compile-time copy of the block-uniform "every
// tile of this block is fully in range and 16-byte aligned" flag, so the hot instantiation has no
// branch (and therefore no compiler-inserted wait) between issuing the next tile's global loads and
// starting this tile's MFMAs: the loads fly under 64 MFMAs (~4k cycles) and are only waited for at
// the ds_write that follows them.
template <int TI, int TJ, bool AROW, bool FAST>
__device__ __forceinline__ void gemm_kloop(v4d (&acc)[TJ][TI], const double* __restrict__ A, int lda,
                                           const double* __restrict__ B, int ldb, int i0, int j0, int M, int N,
                                           int kbeg, int kend, double* __restrict__ As, double* __restrict__ Bs,
                                           int tid, int wi, int wj, int l15, int l4)
{
    constexpr int BM = 32 * TI, BN = 32 * TJ;
    constexpr int ASZ = AROW ? BK * (BM + 16) : BM * LDKF, BSZ = BN * LDKF;
    v2d ra[TI], rb[TJ];
    const int nk = (kend - kbeg + BK - 1) / BK;
    if (nk > 0) {
        if (AROW) load_rowfast<TI>(ra, A, lda, i0, kbeg, M, kend, FAST, tid);
        else load_kfast<TI>(ra, A, lda, i0, kbeg, M, kend, FAST, tid);
        load_kfast<TJ>(rb, B, ldb, j0, kbeg, N, kend, FAST, tid);
        if (AROW) store_rowfast<TI>(ra, As, tid); else store_kfast<TI>(ra, As, tid);
        store_kfast<TJ>(rb, Bs, tid);
    }
    __syncthreads();
    for (int kt = 0; kt < nk; ++kt) {
        const int buf = kt & 1;
        if (kt + 1 < nk) {
            const int k0 = kbeg + (kt + 1) * BK;
            if (AROW) load_rowfast<TI>(ra, A, lda, i0, k0, M, kend, FAST, tid);
            else load_kfast<TI>(ra, A, lda, i0, k0, M, kend, FAST, tid);
            load_kfast<TJ>(rb, B, ldb, j0, k0, N, kend, FAST, tid);
        }
        mma_tile<TI, TJ, AROW>(acc, As + buf * ASZ, Bs + buf * BSZ, wi, wj, l15, l4);
        if (kt + 1 < nk) {
            if (AROW) store_rowfast<TI>(ra, As + (buf ^ 1) * ASZ, tid); else store_kfast<TI>(ra, As + (buf ^ 1) * ASZ, tid);
            store_kfast<TJ>(rb, Bs + (buf ^ 1) * BSZ, tid);
        }
        __syncthreads();
    }
}

// ---- K loop with the issue order spelled out -------------------------------------------------------------------------------------
// The plain loop above, as hipcc schedules it (ISA): the 8 ds_read_b64 of a 4-deep MFMA step are issued after the previous step's 16
// MFMAs and waited for before the next 16; the staging registers go to LDS after the tile's last MFMA, then the barrier, then the
// first reads of the next tile.  Reading the fragments one step ahead but still in clumps (sched_barrier-fenced groups) changed nothing
// (65.8 -> 66.2 TFLOP/s isolated); what helps is below.
template <int TI, int TJ, bool AROW>
__device__ __forceinline__ void load_frags(double (&rowv)[TI], double (&colv)[TJ], const double* __restrict__ as,
                                           const double* __restrict__ bs, int ks, int wi, int wj, int l15, int l4)
{
    constexpr int LA = 32 * TI + 16;
    const int kk = 4 * ks + l4;
#pragma unroll
    for (int b = 0; b < TI; ++b)
        rowv[b] = AROW ? as[kk * LA + wi * 16 * TI + 16 * b + l15] : as[(wi * 16 * TI + 16 * b + l15) * LDKF + kk];
#pragma unroll
    for (int a = 0; a < TJ; ++a) colv[a] = bs[(wj * 16 * TJ + 16 * a + l15) * LDKF + kk];
}

// MFMAs t in [T0, T1) of a step, t = a * TI + b
template <int TI, int TJ, int T0, int T1>
__device__ __forceinline__ void mfma_range(v4d (&acc)[TJ][TI], const double (&rowv)[TI], const double (&colv)[TJ])
{
#pragma unroll
    for (int t = T0; t < T1; ++t)
        acc[t / TI][t % TI] = __builtin_amdgcn_mfma_f64_16x16x4f64(colv[t / TI], rowv[t % TI], acc[t / TI][t % TI], 0, 0, 0);
}

// PMC of the plain loop at one wave per SIMD (profiles/r03_tn_issue_order.txt): the MFMA pipe is busy 78 % of the time; the wave spends
// 8 % in s_waitcnt and ~14 % issuing its LDS / VMEM / VALU instructions in clumps between the 16-MFMA bursts, where nothing executes on
// the matrix pipe.  Here every memory instruction of a K tile is issued right behind ONE MFMA (64 cycles of shadow each) and
// sched_barrier(0) after each pair keeps hipcc from clumping them again:
//   MFMA  0- 7: the 8 global loads of the next tile          MFMA  8-15: the 8 fragment reads of step 1
//   MFMA 16-23: fragment reads of step 2                     MFMA 32-39: fragment reads of step 3
//   MFMA 40-47: the 8 LDS writes of the next tile            MFMA 55   : barrier
//   MFMA 56-63: fragment reads of the next tile's step 0
// Whole tiles only (the FAST instantiation of the 4 x 4 wave tile, k-fast operands).
template <int TI, int TJ>
__device__ __forceinline__ void frag_one(double (&rowv)[TI], double (&colv)[TJ], const double* __restrict__ as, const double* __restrict__ bs,
                                         int q, int kk, int wi, int wj, int l15)
{
    // order c0, r0, r1, r2, r3, c1, c2, c3: what the first MFMAs of a step need comes first
    if (q == 0) colv[0] = bs[(wj * 16 * TJ + l15) * LDKF + kk];
    else if (q <= TI) rowv[q - 1] = as[(wi * 16 * TI + 16 * (q - 1) + l15) * LDKF + kk];
    else colv[q - TI] = bs[(wj * 16 * TJ + 16 * (q - TI) + l15) * LDKF + kk];
}

template <int TI, int TJ>
__device__ __forceinline__ void mfma_one(v4d (&acc)[TJ][TI], const double (&rowv)[TI], const double (&colv)[TJ], int t)
{
    acc[t / TI][t % TI] = __builtin_amdgcn_mfma_f64_16x16x4f64(colv[t / TI], rowv[t % TI], acc[t / TI][t % TI], 0, 0, 0);
}

template <int TI, int TJ>
__device__ __forceinline__ void gemm_kloop_il(v4d (&acc)[TJ][TI], const double* __restrict__ A, int lda,
                                              const double* __restrict__ B, int ldb, int i0, int j0, int M, int N,
                                              int kbeg, int kend, double* __restrict__ As, double* __restrict__ Bs,
                                              int tid, int wi, int wj, int l15, int l4)
{
    static_assert(TI == 4 && TJ == 4, "issue pattern written for the 4 x 4 wave tile");
    constexpr int BM = 32 * TI, BN = 32 * TJ, NT = TI * TJ;
    constexpr int ASZ = BM * LDKF, BSZ = BN * LDKF;
    v2d ra[TI], rb[TJ];
    const int nk = (kend - kbeg) / BK;
    if (nk <= 0) { __syncthreads(); return; }
    load_kfast<TI>(ra, A, lda, i0, kbeg, M, kend, true, tid);
    load_kfast<TJ>(rb, B, ldb, j0, kbeg, N, kend, true, tid);
    store_kfast<TI>(ra, As, tid);
    store_kfast<TJ>(rb, Bs, tid);
    __syncthreads();
    double r0[TI], c0[TJ], r1[TI], c1[TJ];
    load_frags<TI, TJ, false>(r0, c0, As, Bs, 0, wi, wj, l15, l4);
    // this thread's pieces of the two operand tiles: (column idx / 8, k pair idx % 8), idx = tid + 256 q
    const double* ga = A + (size_t) (i0 + tid / 8) * lda + 2 * (tid % 8);
    const double* gb = B + (size_t) (j0 + tid / 8) * ldb + 2 * (tid % 8);
    const size_t sa = (size_t) 32 * lda, sb = (size_t) 32 * ldb;
    const int lw = (tid / 8) * LDKF + 2 * (tid % 8);
#define QR_SB __builtin_amdgcn_sched_barrier(0)
    for (int kt = 0; kt + 1 < nk; ++kt) {
        const int buf = kt & 1;
        const double* as = As + buf * ASZ;
        const double* bs = Bs + buf * BSZ;
        double* asn = As + (buf ^ 1) * ASZ;
        double* bsn = Bs + (buf ^ 1) * BSZ;
        const int k0 = kbeg + (kt + 1) * BK;
#pragma unroll
        for (int t = 0; t < NT; ++t) {                       // step 0
            mfma_one<TI, TJ>(acc, r0, c0, t);
            if (t < 4) ra[t] = *reinterpret_cast<const v2d*>(ga + t * sa + k0);
            else if (t < 8) rb[t - 4] = *reinterpret_cast<const v2d*>(gb + (t - 4) * sb + k0);
            else frag_one<TI, TJ>(r1, c1, as, bs, t - 8, 4 + l4, wi, wj, l15);
            QR_SB;
        }
#pragma unroll
        for (int t = 0; t < NT; ++t) {                       // step 1
            mfma_one<TI, TJ>(acc, r1, c1, t);
            if (t < 8) frag_one<TI, TJ>(r0, c0, as, bs, t, 8 + l4, wi, wj, l15);
            QR_SB;
        }
#pragma unroll
        for (int t = 0; t < NT; ++t) {                       // step 2
            mfma_one<TI, TJ>(acc, r0, c0, t);
            if (t < 8) frag_one<TI, TJ>(r1, c1, as, bs, t, 12 + l4, wi, wj, l15);
            else if (t < 12) *reinterpret_cast<v2d*>(asn + lw + (t - 8) * 32 * LDKF) = ra[t - 8];
            else *reinterpret_cast<v2d*>(bsn + lw + (t - 12) * 32 * LDKF) = rb[t - 12];
            QR_SB;
        }
#pragma unroll
        for (int t = 0; t < NT / 2; ++t) { mfma_one<TI, TJ>(acc, r1, c1, t); QR_SB; }      // step 3, first half
        __syncthreads();
        QR_SB;
#pragma unroll
        for (int t = NT / 2; t < NT; ++t) {                  // step 3, second half, over the next tile's first fragment reads
            mfma_one<TI, TJ>(acc, r1, c1, t);
            // r0 / c0 are free: their last readers (step 2) were issued a burst ago
            frag_one<TI, TJ>(r0, c0, asn, bsn, t - NT / 2, l4, wi, wj, l15);
            QR_SB;
        }
    }
#undef QR_SB
    {   // last tile: nothing to prefetch
        const int buf = (nk - 1) & 1;
        const double* as = As + buf * ASZ;
        const double* bs = Bs + buf * BSZ;
        load_frags<TI, TJ, false>(r1, c1, as, bs, 1, wi, wj, l15, l4);
        mfma_range<TI, TJ, 0, NT>(acc, r0, c0);
        load_frags<TI, TJ, false>(r0, c0, as, bs, 2, wi, wj, l15, l4);
        mfma_range<TI, TJ, 0, NT>(acc, r1, c1);
        load_frags<TI, TJ, false>(r1, c1, as, bs, 3, wi, wj, l15, l4);
        mfma_range<TI, TJ, 0, NT>(acc, r0, c0);
        mfma_range<TI, TJ, 0, NT>(acc, r1, c1);
        __syncthreads();
    }
}

// (Global loads two tiles ahead -- a second staging register set, 256 VGPRs and 84 B of scratch -- were measured at 59.5 against 68.2
// TFLOP/s on the isolated product and removed.)

// C = alpha*acc (+ beta*C on the generic path).  STORE_ONLY: no load sits between the stores (a load
// there makes every store wait for the previous one: vmcnt is in-order and counts stores).
template <int TI, int TJ, bool STORE_ONLY>
__device__ __forceinline__ void gemm_epilogue(const v4d (&acc)[TJ][TI], double* __restrict__ C, int ldc, int M, int N,
                                              int i0, int j0, double alpha, double beta, int wi, int wj, int l15, int l4)
{
#pragma unroll
    for (int a = 0; a < TJ; ++a)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int j = j0 + wj * 16 * TJ + 16 * a + l4 + 4 * r;
#pragma unroll
            for (int b = 0; b < TI; ++b) {
                const int i = i0 + wi * 16 * TI + 16 * b + l15;
                if (i < M && j < N) {
                    double* cp = C + (size_t) j * ldc + i;
                    double v = alpha * acc[a][b][r];
                    if (!STORE_ONLY) v += beta * (*cp);
                    *cp = v;
                }
            }
        }
}


// ------------------------------------------------------------------------------------------------
// The leaf's long-K in-panel product Z = A^T [B1 | B2] (A: the leaf's 32 columns) on 32 x (32 TJ) tiles: shared by the fused launch of
// the leaf's reconstruction (qr_panel_tsqr.hip, hr3_ep_kernel) and the stand-alone product of tall leaves (qr_kernels.hip).
// ------------------------------------------------------------------------------------------------
struct EpArgs {
    int N1, N2, K, kchunk, tiles, ksplit;    // Z is 32 x (N1 + N2); K rows in slices of kchunk; tiles = (N1 + N2) / 32 (redo tiling)
    int ftiles;                               // column tiles of the fused launch: ceil((N1 + N2) / (32 TJ))
    const double* Q; int ldq;                 // the leaf's 32 columns (Q before / V after the reconstruction), K rows
    const double* B1; int ldb1;               // A_rest: K x N1
    const double* B2; int ldb2;               // V_prev: K x N2 (rows from the leaf's top row)
    double* slabs; size_t slab_stride;        // slab z at slabs + z * slab_stride, each 32 x (N1 + N2), ld 32
};

#define EP_FUSED_SMEM_BYTES(TJ) (sizeof(double) * 2 * 32 * (1 + (TJ)) * LDKF)

// The fused launch's product item: a 32 x (32 TJ) tile of Z over one K slice, on the first 256 threads of a 512-thread workgroup
// (the launch is sized by the reconstruction: ~205 VGPRs, one workgroup per compute unit, so the product has ONE 4-wave
// workgroup per CU where the separate launch had up to eight: wider tiles -- TJ accumulators per wave between two barriers -- and TWO
// k-tiles of global loads in flight make up for the missing occupancy).  The 32-column blocks of a tile may sit on either side of the
// A_rest | V_prev boundary (N1 is a multiple of 32); blocks past the last column read block 0 of the tile and are never stored.
template <int TJ>
__device__ __forceinline__ void ep_fused_item(const EpArgs& e, int tile, int z)
{
    extern __shared__ __attribute__((aligned(16))) double ep_smem[];
    constexpr int ASZ = 32 * LDKF, BSZ = 32 * TJ * LDKF;
    double* As = ep_smem;                    // [2][32][LDKF]
    double* Bsm = ep_smem + 2 * ASZ;         // [2][32 TJ][LDKF]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wi = wave & 1, wj = wave >> 1, l15 = lane & 15, l4 = lane >> 4;
    const int j0 = tile * 32 * TJ, N = e.N1 + e.N2;
    const int kbeg = z * e.kchunk, kend = min(e.K, kbeg + e.kchunk);
    const int nk = (kend - kbeg + BK - 1) / BK;
    // this thread's column of every 32-column block, and of the Q tile: 8 threads per column, 2 doubles each per k-tile
    const double* bp[TJ];
#pragma unroll
    for (int q = 0; q < TJ; ++q) {
        int j = j0 + 32 * q + (tid >> 3);
        if (j >= N) j = j0 + (tid >> 3);
        bp[q] = (j >= e.N1 ? e.B2 + (size_t) (j - e.N1) * e.ldb2 : e.B1 + (size_t) j * e.ldb1) + 2 * (tid & 7);
    }
    const double* qp = e.Q + (size_t) (tid >> 3) * e.ldq + 2 * (tid & 7);
    v2d ra[2], rb[2][TJ];
    auto gload = [&](int st, int k0) {
        ra[st] = *reinterpret_cast<const v2d*>(qp + k0);
#pragma unroll
        for (int q = 0; q < TJ; ++q) rb[st][q] = *reinterpret_cast<const v2d*>(bp[q] + k0);
    };
    auto sstore = [&](int st, int buf) {
        *reinterpret_cast<v2d*>(As + buf * ASZ + (tid >> 3) * LDKF + 2 * (tid & 7)) = ra[st];
#pragma unroll
        for (int q = 0; q < TJ; ++q) *reinterpret_cast<v2d*>(Bsm + buf * BSZ + (32 * q + (tid >> 3)) * LDKF + 2 * (tid & 7)) = rb[st][q];
    };
    v4d acc[TJ][1];
#pragma unroll
    for (int a = 0; a < TJ; ++a) acc[a][0] = (v4d){0.0, 0.0, 0.0, 0.0};
    if (nk > 0) gload(0, kbeg);
    if (nk > 1) gload(1, kbeg + BK);
    if (nk > 0) sstore(0, 0);
    __syncthreads();
    // k-tile kt: LDS buffer kt & 1 holds it, register stage (kt + 1) & 1 holds tile kt + 1, stage kt & 1 is free for tile kt + 2
    auto step = [&](int kt, int st) {          // st = kt & 1 as a literal after unrolling
        if (kt + 2 < nk) gload(st, kbeg + (kt + 2) * BK);
        mma_tile<1, TJ, false>(acc, As + st * ASZ, Bsm + st * BSZ, wi, wj, l15, l4);
        if (kt + 1 < nk) sstore(st ^ 1, st ^ 1);
        __syncthreads();
    };
    int kt = 0;
    for (; kt + 1 < nk; kt += 2) { step(kt, 0); step(kt + 1, 1); }
    if (kt < nk) step(kt, 0);
    gemm_epilogue<1, TJ, true>(acc, e.slabs + (size_t) z * e.slab_stride, 32, 32, N, 0, j0, 1.0, 0.0, wi, wj, l15, l4);
}


#endif
