// qr_kernels.hip -- hand-written gfx950 (CDNA4 / MI355X) kernels for fp64 blocked Householder QR,
// plus the extern "C" launchers the C host layer (qr_host.c) calls.  No vendor BLAS/solver calls.
//
// What replaces what in the reference (brian-kelley/CUDA-QR):
//   (the panel: qr_panel_fused.hip / qr_panel_cqr.hip / qr_panel_tsqr.hip <- panelHouseholderKernel, qr.cu:60-333 / qr.c:109-235)
//   gemm_tn / gemm_nn     <- trailingUpdateKernel (qr.cu:335-465) / qr.c:255-293: the compact-WY update
//                            W = (V T)^T A2 ; A2 -= V W as two dense contractions on
//                            v_mfma_f64_16x16x4_f64 tiles staged through LDS.
//   larft_kernel          <- the WY accumulation qr.c:170-213 (compact-WY T instead of W = Y*T).
//
// MFMA f64 16x16x4 lane maps (cdna_hip_programming.md section 3):
//   A-operand: lane l holds Aop[p = l&15][k = l>>4];  B-operand: lane l holds Bop[k = l>>4][q = l&15]
//   D: lane l, reg r holds D[p = (l>>4) + 4r][q = l&15].
// We always put the COLUMN index of the (column-major) output on p and the ROW index on q, so that
// 16 consecutive lanes touch 16 consecutive rows of one output column (128 contiguous bytes).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string.h>
#include <stdlib.h>
#include <utility>
#include <mutex>
#include "qr_device.h"
#include "qr_common.h"
#include "qr_gemm_tile.h"

#define HIPCHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) return (int) e_; } while (0)

// C = beta*C + alpha*A*B     A: M x K (lda), B: K x N (ldb), C: M x N (ldc), all column-major.
// Used for: trailing update A2 -= V*W (K = nb), VT = V*T, T merges, Q*R products, Q_local*Q_tree.
// TAG only gives the wide trailing-update launches their own kernel name in profiler output (TAG = 1).
template <int TI, int TJ, bool FAST>
__device__ __forceinline__ void gemm_nn_body(int M, int N, int K, double alpha, const double* __restrict__ A, int lda,
                                             const double* __restrict__ B, int ldb, double beta,
                                             double* __restrict__ C, int ldc)
{
    constexpr int BM = 32 * TI, BN = 32 * TJ, LA = BM + 16;
    extern __shared__ __attribute__((aligned(16))) double smem[];
    double* As = smem;                      // [2][BK][LA]
    double* Bs = smem + 2 * BK * LA;        // [2][BN][LDKF]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wi = wave & 1, wj = wave >> 1;
    const int i0 = blockIdx.x * BM, j0 = blockIdx.y * BN;
    const int l15 = lane & 15, l4 = lane >> 4;

    // beta == 1 with alpha = +-1 (the trailing update C -= V*W): C enters through the accumulators, acc = C/alpha
    // (exact), so the epilogue is stores only.  Any other alpha takes the generic beta epilogue: C/alpha would round C
    // twice, overflow for tiny |alpha| and produce 0 * inf for alpha = 0.
    const bool cinit = (beta == 1.0) && (alpha == 1.0 || alpha == -1.0);
    const double inv_alpha = 1.0 / alpha;
    v4d acc[TJ][TI];
    if (cinit) {          // one uniform branch, straight-line body: all TI*TJ*4 loads in flight together
#pragma unroll
        for (int a = 0; a < TJ; ++a)
#pragma unroll
            for (int b = 0; b < TI; ++b) {
                const int i = i0 + wi * 16 * TI + 16 * b + l15;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int j = j0 + wj * 16 * TJ + 16 * a + l4 + 4 * r;
                    const double cv = C[(size_t) min(j, N - 1) * ldc + min(i, M - 1)];    // clamped, unconditional
                    acc[a][b][r] = (i < M && j < N) ? cv * inv_alpha : 0.0;
                }
            }
    } else {
#pragma unroll
        for (int a = 0; a < TJ; ++a)
#pragma unroll
            for (int b = 0; b < TI; ++b) acc[a][b] = (v4d){0.0, 0.0, 0.0, 0.0};
    }

    // FAST (host-checked): every tile of this launch is fully in range, K % BK == 0, operands 16-B aligned
    gemm_kloop<TI, TJ, true, FAST>(acc, A, lda, B, ldb, i0, j0, M, N, 0, K, As, Bs, tid, wi, wj, l15, l4);

    if (cinit || beta == 0.0) gemm_epilogue<TI, TJ, true>(acc, C, ldc, M, N, i0, j0, alpha, beta, wi, wj, l15, l4);
    else gemm_epilogue<TI, TJ, false>(acc, C, ldc, M, N, i0, j0, alpha, beta, wi, wj, l15, l4);
}

template <int TI, int TJ, bool FAST, int TAG = 0>
__global__ __launch_bounds__(256, 2) void gemm_nn_kernel(int M, int N, int K, double alpha,
                                                         const double* __restrict__ A, int lda,
                                                         const double* __restrict__ B, int ldb,
                                                         double beta, double* __restrict__ C, int ldc)
{
    gemm_nn_body<TI, TJ, FAST>(M, N, K, alpha, A, lda, B, ldb, beta, C, ldc);
}

// blockIdx.z = batch index: the same small product on operands a fixed stride apart (the T-merge tree)
template <int TI, int TJ, bool FAST>
__global__ __launch_bounds__(256, 2) void gemm_nn_batch_kernel(int M, int N, int K, double alpha,
                                                               const double* __restrict__ A, int lda, size_t sA,
                                                               const double* __restrict__ B, int ldb, size_t sB,
                                                               double beta, double* __restrict__ C, int ldc, size_t sC)
{
    const size_t z = blockIdx.z;
    gemm_nn_body<TI, TJ, FAST>(M, N, K, alpha, A + z * sA, lda, B + z * sB, ldb, beta, C + z * sC, ldc);
}

// ------------------------------------------------------------------------------------------------
// Wide trailing update, 8-wave form: the same 128 x 128 block tile and LDS images as gemm_nn_kernel<4,4>, but 512 threads
// = 8 waves arranged 2 (rows) x 4 (cols), each wave a 64 x 32 tile (8 accumulators, 64 VGPRs).  Two workgroups per CU
// then put 4 waves on every SIMD instead of 2: with K = nb = 256 a tile is only 16 K-steps between a 128 KiB read and a
// 128 KiB write of C, and two waves per SIMD left the MFMA pipe idle 35 % of the time (rocprofv3 MfmaUtil 65 %).
// Tile-aligned interior only (M % 128 == N % 128 == K % 16 == 0 parts, 16-byte aligned operands), C += alpha*A*B.
// ------------------------------------------------------------------------------------------------
// TAG only names the launch for profilers: 0 = the wide trailing update (the kernel bench.py's roofline is quoted on),
// 1 = every other use (look-ahead / mid-panel updates), so that rocprofv3's per-kernel average is the wide update's alone.
template <int TAG>
__global__ __launch_bounds__(512, 4) void gemm_nn_w8_kernel(int M, int N, int K, double alpha,
                                                            const double* __restrict__ A, int lda,
                                                            const double* __restrict__ B, int ldb,
                                                            double* __restrict__ C, int ldc)
{
    constexpr int TI = 4, TJ = 2, BM = 128, BN = 128, LA = BM + 16, ASZ = BK * LA, BSZ = BN * LDKF;
    extern __shared__ __attribute__((aligned(16))) double smem[];
    double* As = smem;                      // [2][BK][LA]
    double* Bs = smem + 2 * ASZ;            // [2][BN][LDKF]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wi = wave & 1, wj = wave >> 1;
    int bx = blockIdx.x, by = blockIdx.y;
    if (TAG == 1 && gridDim.y <= 8 && (gridDim.x & 7) == 0) {
        // Tall updates with a handful of column tiles (the tall-skinny factorisation: 128 x 128 of V against 1 - 4 column tiles of W): as
        // launched, the column tiles of one row block are gridDim.x workgroups apart -- different XCDs, or the same one long after the
        // block's tile of V has left its L2 -- and V came from HBM once per column tile (rocprofv3: x1.4 the algorithmic bytes).
        // Workgroups are dealt round-robin to the 8 XCDs: XCD x takes the row blocks r = x (mod 8) and walks their column tiles back to back.
        const int id = blockIdx.x + gridDim.x * blockIdx.y, x8 = id & 7, q = id >> 3;
        by = q % (int) gridDim.y;
        bx = 8 * (q / (int) gridDim.y) + x8;
    }
    const int i0 = bx * BM, j0 = by * BN;
    const int l15 = lane & 15, l4 = lane >> 4;
    const double inv_alpha = 1.0 / alpha;
    v4d acc[TJ][TI];
#pragma unroll
    for (int a = 0; a < TJ; ++a)
#pragma unroll
        for (int b = 0; b < TI; ++b) {
            const int i = i0 + wi * 16 * TI + 16 * b + l15;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int j = j0 + wj * 16 * TJ + 16 * a + l4 + 4 * r;
                acc[a][b][r] = C[(size_t) j * ldc + i] * inv_alpha;
            }
        }
    // tile loaders: A tile BK x BM row-fast (64 double2 per column), B tile BN x BK k-fast (8 double2 per column); 2 each
    v2d ra[2], rb[2];
    const int nk = K / BK;
    auto gload = [&](int k0) {
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const int idx = tid + 512 * q;
            ra[q] = *reinterpret_cast<const v2d*>(A + (size_t) (k0 + idx / 64) * lda + i0 + 2 * (idx % 64));
            rb[q] = *reinterpret_cast<const v2d*>(B + (size_t) (j0 + idx / 8) * ldb + k0 + 2 * (idx % 8));
        }
    };
    auto sstore = [&](int buf) {
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const int idx = tid + 512 * q;
            *reinterpret_cast<v2d*>(As + buf * ASZ + (idx / 64) * LA + 2 * (idx % 64)) = ra[q];
            *reinterpret_cast<v2d*>(Bs + buf * BSZ + (idx / 8) * LDKF + 2 * (idx % 8)) = rb[q];
        }
    };
    gload(0);
    sstore(0);
    __syncthreads();
    for (int kt = 0; kt < nk; ++kt) {
        const int buf = kt & 1;
        if (kt + 1 < nk) gload((kt + 1) * BK);
        mma_tile<TI, TJ, true, false>(acc, As + buf * ASZ, Bs + buf * BSZ, wi, wj, l15, l4);
        if (kt + 1 < nk) sstore(buf ^ 1);
        __syncthreads();
    }
    gemm_epilogue<TI, TJ, true>(acc, C, ldc, M, N, i0, j0, alpha, 1.0, wi, wj, l15, l4);
}

// C = alpha * A^T * B (+ beta*C when not split)   A: K x M (lda), B: K x N (ldb), C: M x N (ldc).
// K is the long dimension (panel height): gridDim.z K-slices each write their own slab
// (slab z at C + z*slab_stride, ld = ldc) and slab_reduce_kernel sums them in a fixed order
// (deterministic; no float atomics).  Used for W = (V T)^T A2, Gram = V^T V, Q^T Q.
// MIXED (round 6; with FAST): launched over the WHOLE ragged problem, ceil(M / BM) x ceil(N / BN) tiles -- interior tiles take the fast K
// loop, edge tiles (a workgroup-uniform test) the guarded one.  Before, the ragged strips were launches of their own behind the interior:
// a row of a few edge tiles, each walking a K slice as long as the interior's, with the chip idle around them (the wide product of a
// 16448^2 factorisation: 37 TFLOP/s against 55 on the aligned 16384^2).
template <int TI, int TJ, bool FAST, int TAG = 0, bool MIXED = false>
__global__ __launch_bounds__(256, 2) void gemm_tn_kernel(int M, int N, int K, int kchunk, double alpha,
                                                         const double* __restrict__ A, int lda,
                                                         const double* __restrict__ B, int ldb,
                                                         double beta, double* __restrict__ C, int ldc,
                                                         size_t slab_stride)
{
    constexpr int BM = 32 * TI, BN = 32 * TJ;
    extern __shared__ __attribute__((aligned(16))) double smem[];
    double* As = smem;                        // [2][BM][LDKF]
    double* Bs = smem + 2 * BM * LDKF;        // [2][BN][LDKF]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wi = wave & 1, wj = wave >> 1;
    int by = blockIdx.y, bz = blockIdx.z;
    if (gridDim.x == 1 && gridDim.y > 1 && gridDim.y <= 8 && gridDim.z >= 8) {
        // One row of output tiles against a long K split into slices (the tall-skinny update's V^T A2): the column tiles of a K slice all read
        // the same slice of A, but round-robin dealing puts them on different XCDs.  Slices z < 8 floor(gz / 8) are regrouped so that XCD x
        // takes the slices z = x (mod 8) and walks their column tiles back to back (the slice of A comes from HBM once, not gridDim.y times).
        const int id = blockIdx.y + gridDim.y * blockIdx.z, zfull = (int) (gridDim.z & ~7u);
        if (id < (int) gridDim.y * zfull) {
            const int x8 = id & 7, q = id >> 3;
            by = q % (int) gridDim.y;
            bz = 8 * (q / (int) gridDim.y) + x8;
        }
    }
    const int i0 = blockIdx.x * BM, j0 = by * BN;
    const int l15 = lane & 15, l4 = lane >> 4;
    const int kbeg = bz * kchunk;
    const int kend = min(K, kbeg + kchunk);
    C += (size_t) bz * slab_stride;

    v4d acc[TJ][TI];
#pragma unroll
    for (int a = 0; a < TJ; ++a)
#pragma unroll
        for (int b = 0; b < TI; ++b) acc[a][b] = (v4d){0.0, 0.0, 0.0, 0.0};

    // TAG 0 / 1 (1: the wide product of the trailing update under its own profiler name): K loop with the issue order spelled out where
    // the instantiation allows it (whole 128 x 128 tiles); TAG 3 / 4: the same two with the plain double-buffered loop (MI355XQR_KPIPE=0)
    if constexpr (MIXED) {
        if (i0 + BM > M || j0 + BN > N) gemm_kloop<TI, TJ, false, false>(acc, A, lda, B, ldb, i0, j0, M, N, kbeg, kend, As, Bs, tid, wi, wj, l15, l4);
        else if constexpr (TAG <= 1 && TI == 4 && TJ == 4) gemm_kloop_il<TI, TJ>(acc, A, lda, B, ldb, i0, j0, M, N, kbeg, kend, As, Bs, tid, wi, wj, l15, l4);
        else gemm_kloop<TI, TJ, false, true>(acc, A, lda, B, ldb, i0, j0, M, N, kbeg, kend, As, Bs, tid, wi, wj, l15, l4);
    }
    else if constexpr (TAG <= 1 && FAST && TI == 4 && TJ == 4) gemm_kloop_il<TI, TJ>(acc, A, lda, B, ldb, i0, j0, M, N, kbeg, kend, As, Bs, tid, wi, wj, l15, l4);
    else gemm_kloop<TI, TJ, false, FAST>(acc, A, lda, B, ldb, i0, j0, M, N, kbeg, kend, As, Bs, tid, wi, wj, l15, l4);

    if (beta == 0.0) gemm_epilogue<TI, TJ, true>(acc, C, ldc, M, N, i0, j0, alpha, beta, wi, wj, l15, l4);
    else gemm_epilogue<TI, TJ, false>(acc, C, ldc, M, N, i0, j0, alpha, beta, wi, wj, l15, l4);
}

// ------------------------------------------------------------------------------------------------
// Wide TN product of the trailing update, Wt = A2^T (V T):  M = columns of A2 (thousands), N = nb (256 / 512), K = panel height.
// The 128 x 128 kernel above gives every 128 columns of V T their own workgroup, so A2 -- the big operand, 8 mk nt bytes -- is read
// N / 128 times (rocprofv3: x2.3 the algorithmic bytes at nb = 256, x4 at 512).  Here a workgroup owns 128 columns of A2 against
// 256 columns of V T: 512 threads = 8 waves as 2 x 4, wave tile 64 x 64 (16 accumulators), so A2 is read N / 256 times; the V T
// slice of a K chunk is shared through L2 by the workgroups that walk the same chunk.  LDS 108 KB (two stages): one workgroup,
// two waves per SIMD, per compute unit -- the same occupancy as two workgroups of the 4-wave kernel.
// Interior only: M % 128 == 0, N % 256 == 0, K % 16 == 0, 16-byte aligned operands; slab z of the split-K at C + z * slab_stride.
// ------------------------------------------------------------------------------------------------
template <int TAG>
__global__ __launch_bounds__(512) void gemm_tn_wide_kernel(int K, int kchunk, const double* __restrict__ A, int lda,
                                                           const double* __restrict__ B, int ldb, double* __restrict__ C, int ldc,
                                                           size_t slab_stride)
{
    constexpr int BM = 128, BN = 256, ASZ = BM * LDKF, BSZ = BN * LDKF;
    extern __shared__ __attribute__((aligned(16))) double smem[];
    double* As = smem;                       // [2][BM][LDKF]
    double* Bs = smem + 2 * ASZ;             // [2][BN][LDKF]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wi = wave & 1, wj = wave >> 1, l15 = lane & 15, l4 = lane >> 4;
    const int i0 = blockIdx.x * BM, j0 = blockIdx.y * BN;
    const int kbeg = blockIdx.z * kchunk, kend = min(K, kbeg + kchunk);
    const int nk = (kend - kbeg) / BK;
    C += (size_t) blockIdx.z * slab_stride;
    // 8 threads per column, one double2 each per k-tile: 2 columns of A2 and 4 of V T per thread
    const double* ap[2];
    const double* bp[4];
#pragma unroll
    for (int q = 0; q < 2; ++q) ap[q] = A + (size_t) (i0 + ((tid + 512 * q) >> 3)) * lda + 2 * (tid & 7);
#pragma unroll
    for (int q = 0; q < 4; ++q) bp[q] = B + (size_t) (j0 + ((tid + 512 * q) >> 3)) * ldb + 2 * (tid & 7);
    v2d ra[2], rb[4];
    auto gload = [&](int k0) {
#pragma unroll
        for (int q = 0; q < 2; ++q) ra[q] = *reinterpret_cast<const v2d*>(ap[q] + k0);
#pragma unroll
        for (int q = 0; q < 4; ++q) rb[q] = *reinterpret_cast<const v2d*>(bp[q] + k0);
    };
    auto sstore = [&](int buf) {
#pragma unroll
        for (int q = 0; q < 2; ++q) *reinterpret_cast<v2d*>(As + buf * ASZ + ((tid + 512 * q) >> 3) * LDKF + 2 * (tid & 7)) = ra[q];
#pragma unroll
        for (int q = 0; q < 4; ++q) *reinterpret_cast<v2d*>(Bs + buf * BSZ + ((tid + 512 * q) >> 3) * LDKF + 2 * (tid & 7)) = rb[q];
    };
    v4d acc[4][4];
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b) acc[a][b] = (v4d){0.0, 0.0, 0.0, 0.0};
    if (nk > 0) { gload(kbeg); sstore(0); }
    __syncthreads();
    for (int kt = 0; kt < nk; ++kt) {
        const int buf = kt & 1;
        if (kt + 1 < nk) gload(kbeg + (kt + 1) * BK);
        const double* as = As + buf * ASZ + (wi * 64 + l15) * LDKF;
        const double* bs = Bs + buf * BSZ + (wj * 64 + l15) * LDKF;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            const int kk = 4 * ks + l4;
            double rowv[4], colv[4];
#pragma unroll
            for (int b = 0; b < 4; ++b) rowv[b] = as[16 * b * LDKF + kk];
#pragma unroll
            for (int a = 0; a < 4; ++a) colv[a] = bs[16 * a * LDKF + kk];
            __builtin_amdgcn_s_setprio(1);
#pragma unroll
            for (int a = 0; a < 4; ++a)
#pragma unroll
                for (int b = 0; b < 4; ++b) acc[a][b] = __builtin_amdgcn_mfma_f64_16x16x4f64(colv[a], rowv[b], acc[a][b], 0, 0, 0);
            __builtin_amdgcn_s_setprio(0);
        }
        if (kt + 1 < nk) sstore(buf ^ 1);
        __syncthreads();
    }
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            double* cp = C + (size_t) (j0 + wj * 64 + 16 * a + l4 + 4 * r) * ldc + i0 + wi * 64 + l15;
#pragma unroll
            for (int b = 0; b < 4; ++b) cp[16 * b] = acc[a][b][r];
        }
}

// The leaf's two long-K products in one launch:  [W | G^T] = V_l^T [A_rest | V_prev]  (32 x (N1 + N2), K = leaf height).
// Column tiles j0 < N1 read B1 (the rest of the panel), the others B2 (the reflectors of the panel's earlier leaves): the Gram
// blocks the panel's T needs are thus collected leaf by leaf, and the panel-wide Gram product after the last leaf goes away.
// Workgroup -> (column tile, K slice): workgroups are dealt round-robin over the 8 XCDs, and every column tile of a K slice reads the
// same 32-column slice of A -- so the slices are dealt to the XCDs (slice z on XCD z % 8, all its tiles there): A's slice then comes
// from HBM once per slice instead of once per XCD that happens to hold one of its tiles (a tall leaf: 6 tiles -> up to 6 x 67 MB).
// The grid is tiles * 8 * ceil(ksplit / 8) workgroups; those whose slice does not exist leave at once.
__global__ __launch_bounds__(256, 2) void gemm_tn_dual_kernel(int N1, int N2, int K, int kchunk, int tiles, int ksplit,
                                                              const double* __restrict__ A, int lda,
                                                              const double* __restrict__ B1, int ldb1, const double* __restrict__ B2,
                                                              int ldb2, double* __restrict__ C, size_t slab_stride)
{
    extern __shared__ __attribute__((aligned(16))) double smem[];
    double* As = smem;
    double* Bs = smem + 2 * 32 * LDKF;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wi = wave & 1, wj = wave >> 1;
    const int slot = blockIdx.x >> 3, z = (int) (blockIdx.x & 7) + 8 * (slot / tiles);
    if (z >= ksplit) return;
    const int j0 = (slot % tiles) * 32;
    const int l15 = lane & 15, l4 = lane >> 4;
    const int kbeg = z * kchunk;
    const int kend = min(K, kbeg + kchunk);
    C += (size_t) z * slab_stride;
    const bool second = j0 >= N1;
    const double* B = second ? B2 + (size_t) (j0 - N1) * ldb2 : B1 + (size_t) j0 * ldb1;
    const int ldb = second ? ldb2 : ldb1;
    v4d acc[1][1];
    acc[0][0] = (v4d){0.0, 0.0, 0.0, 0.0};
    gemm_kloop<1, 1, false, true>(acc, A, lda, B, ldb, 0, 0, 32, 32, kbeg, kend, As, Bs, tid, wi, wj, l15, l4);
    gemm_epilogue<1, 1, true>(acc, C, 32, 32, N1 + N2, 0, j0, 1.0, 0.0, wi, wj, l15, l4);
}

// out(:,j) = beta*out(:,j) + Tm^T * sum_z slab_z(:,j)   (Tm optional, upper triangular M x M, M <= 256)
// One 256-thread block per output column.  The slab index is spread over 256/Mp thread groups (Mp = M rounded up
// to 32), each summing its slabs in a fixed order, then the groups are added in a fixed order: deterministic and
// short dependency chains even for hundreds of slabs.
__global__ __launch_bounds__(256) void slab_reduce_kernel(int M, int N, int nslab, const double* __restrict__ slabs, int lds,
                                                          size_t slab_stride, const double* __restrict__ Tm, int ldt,
                                                          double beta, double* __restrict__ out, int ldo, int chunk,
                                                          int nsplit, double* __restrict__ out2, int ldo2)
{
    // columns j >= nsplit (out2 != nullptr): raw sums, stored TRANSPOSED:  out2(j - nsplit, i)  (gemm_tn_dual_kernel's Gram part)
    __shared__ double red[256];
    __shared__ double col[256];
    const int j = blockIdx.x, tid = threadIdx.x;
    // rows [i0, i0 + Mc) of the column: one `chunk`-row piece per blockIdx.y (chunk = 256, or 32 / 64 for reductions over many
    // slabs: the 256 threads then split the slab index 8 / 4 ways instead of walking hundreds of slabs one after the other; no Tm then)
    const int i0 = blockIdx.y * chunk, Mc = min(chunk, M - i0);
    const int Mp = (Mc + 31) & ~31, nz = 256 / Mp;
    const int i = tid % Mp, zp = tid / Mp;
    double s = 0.0;
    if (i < Mc && zp < nz) {
        const double* p = slabs + (size_t) j * lds + i0 + i;
        double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;
        int z = zp;
        for (; z + 3 * nz < nslab; z += 4 * nz) {
            s0 += p[(size_t) z * slab_stride];
            s1 += p[(size_t) (z + nz) * slab_stride];
            s2 += p[(size_t) (z + 2 * nz) * slab_stride];
            s3 += p[(size_t) (z + 3 * nz) * slab_stride];
        }
        for (; z < nslab; z += nz) s0 += p[(size_t) z * slab_stride];
        s = (s0 + s1) + (s2 + s3);
    }
    red[tid] = s;
    __syncthreads();
    if (tid < Mp) {
        double t = 0.0;
        for (int g = 0; g < nz; ++g) t += red[g * Mp + tid];
        col[tid] = t;
    }
    __syncthreads();
    if (out2 && j >= nsplit) {
        if (tid < Mc) out2[(size_t) (i0 + tid) * ldo2 + (j - nsplit)] = col[tid];
        return;
    }
    if (tid < Mc) {
        double v = col[tid];
        if (Tm) {
            double t = 0.0;
            for (int p2 = 0; p2 <= tid; ++p2) t += Tm[(size_t) tid * ldt + p2] * col[p2];   // (T^T)(i,p) = T(p,i)
            v = t;
        }
        double* o = out + (size_t) j * ldo + i0 + tid;
        *o = (beta != 0.0) ? beta * (*o) + v : v;
    }
}

// ------------------------------------------------------------------------------------------------
// Compact-WY T, the counterpart of the reference's W accumulation (qr.c:170-213):
// I - V T V^T = H_0 H_1 ... H_{nbp-1}.
// leaf_t_kernel: diagonal blocks (one block of threads per leaf) from the Gram entries
//   G(q,j) = v_q^T v_j and tau by the column recurrence T(0:j,j) = -tau_j T(0:j,0:j) G(0:j,j);
//   row p of a block depends only on row p, so one thread per row needs no synchronisation.
// The off-diagonal blocks T(0:cb, cb:cb+wb) = -T(0:cb,0:cb) (G(0:cb,cb:cb+wb) T_bb) are two small
// MFMA GEMMs per leaf (qrd_larft below).
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(64) void leaf_t_kernel(int nbp, int ib, const double* __restrict__ G, int ldg,
                                                   const double* __restrict__ tau, double* __restrict__ T, int ldt)
{
    __shared__ double sg[LEAFW][LEAFW + 1];
    __shared__ double st[LEAFW];
    const int cb = blockIdx.x * ib, wb = min(ib, nbp - cb), pl = threadIdx.x;
    const double* Gb = G + (size_t) cb * ldg + cb;
    double* Tb = T + (size_t) cb * ldt + cb;
    for (int e = pl; e < wb * wb; e += 64) {
        const int q = e % wb, jj = e / wb;
        sg[jj][q] = (q < jj) ? Gb[(size_t) jj * ldg + q] : 0.0;
    }
    if (pl < wb) st[pl] = tau[cb + pl];
    __syncthreads();
    if (pl >= wb) return;
    double trow[LEAFW];
#pragma unroll
    for (int q = 0; q < LEAFW; ++q) trow[q] = 0.0;
#pragma unroll
    for (int jj = 0; jj < LEAFW; ++jj) {
        if (jj < wb) {
            const double tj = st[jj];
            double sacc = 0.0;
#pragma unroll
            for (int q = 0; q < jj; ++q) sacc += trow[q] * sg[jj][q];      // trow[q] == 0 for q < pl
            trow[jj] = (pl == jj) ? tj : ((pl < jj) ? -tj * sacc : 0.0);
        }
    }
#pragma unroll
    for (int q = 0; q < LEAFW; ++q)
        if (q < wb) Tb[(size_t) q * ldt + pl] = trow[q];
}

__global__ void transpose_kernel(int n, const double* __restrict__ S, int lds, double* __restrict__ D, int ldd)
{
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e < n * n) { const int p = e % n, c = e / n; D[(size_t) p * ldd + c] = S[(size_t) c * lds + p]; }
}

// D (cols x rows, ldd) = S (rows x cols, lds)^T: the small W of a tall update into the row-fast form gemm_nt wants
__global__ void transpose_rect_kernel(int rows, int cols, const double* __restrict__ S, int lds, double* __restrict__ D, int ldd)
{
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e < rows * cols) { const int i = e % rows, j = e / rows; D[(size_t) i * ldd + j] = S[(size_t) j * lds + i]; }
}

// ------------------------------------------------------------------------------------------------
// small utility kernels
// ------------------------------------------------------------------------------------------------
__global__ void zero_block_kernel(double* A, int ld, int rows, int cols)
{
    const size_t e = (size_t) blockIdx.x * blockDim.x + threadIdx.x;
    if (e < (size_t) rows * cols) A[(e / rows) * ld + (e % rows)] = 0.0;
}

// explicit unit-lower-trapezoidal V (mk x w) from a factored panel (reflector tails below the diagonal)
__global__ void extract_v_kernel(const double* __restrict__ P, int ld, int mk, int w, double* __restrict__ V, int ldv)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x, c = blockIdx.y;
    if (i < mk && c < w) V[(size_t) c * ldv + i] = (i > c) ? P[(size_t) c * ld + i] : (i == c ? 1.0 : 0.0);
}

// R (rrows x n, ldr) = upper triangle of the factored matrix, zero elsewhere (reference qr.c:334-343)
__global__ void extract_r_kernel(const double* __restrict__ A, int lda, int m, int n, double* __restrict__ R, int ldr, int rrows)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x, c = blockIdx.y;
    if (i < rrows && c < n) R[(size_t) c * ldr + i] = (i <= c && i < m) ? A[(size_t) c * lda + i] : 0.0;
}

// block column [k, k + w) of R: out (rrows x w, ldr) = rows 0 .. k + j of column k + j of the factored matrix, zero below
__global__ void extract_r_block_kernel(const double* __restrict__ A, int lda, int k, int w, double* __restrict__ R, int ldr, int rrows)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x, j = blockIdx.y;
    if (i < rrows && j < w) R[(size_t) j * ldr + i] = (i <= k + j) ? A[(size_t) (k + j) * lda + i] : 0.0;
}

// C(i,c) = (i + row_off == c) ? 1 : 0      (reference identity(), qr.c:316-324)
__global__ void set_identity_kernel(double* __restrict__ C, int ld, int rows, int cols, int row_off)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x, c = blockIdx.y;
    if (i < rows && c < cols) C[(size_t) c * ld + i] = (i + row_off == c) ? 1.0 : 0.0;
}

__global__ void copy_block_kernel(const double* __restrict__ S, int lds, double* __restrict__ D, int ldd, int rows, int cols)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x, c = blockIdx.y;
    if (i < rows && c < cols) D[(size_t) c * ldd + i] = S[(size_t) c * lds + i];
}

// `batch` blocks in one launch: block q at S + q * sstride -> D + q * dstride (the P rank blocks of a stacked block column: P launches of
// ~6 us each sat on the exposed tail of every multi-GPU step)
__global__ void copy_blocks_kernel(const double* __restrict__ S, int lds, size_t sstride, double* __restrict__ D, int ldd, size_t dstride, int rows, int cols)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x, c = blockIdx.y;
    const size_t q = blockIdx.z;
    if (i < rows && c < cols) D[q * dstride + (size_t) c * ldd + i] = S[q * sstride + (size_t) c * lds + i];
}

// Counter-based uniform[0,1) generator: element (global row gi, column c) of a total_rows x cols
// matrix depends only on (seed, c*total_rows + gi): shard-count independent, so 1/2/4/8-GPU
// runs factor the same matrix.  splitmix64 finaliser; 53 random mantissa bits.
__host__ __device__ __forceinline__ double hash_uniform(uint64_t seed, uint64_t idx)
{
    uint64_t z = seed * 0x9E3779B97F4A7C15ull + (idx + 1) * 0xD1B54A32D192ED03ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    z ^= z >> 31;
    return (double) (z >> 11) * (1.0 / 9007199254740992.0);
}

__global__ void fill_uniform_kernel(double* __restrict__ A, int ld, long long rows, int cols, long long row_off,
                                    long long total_rows, uint64_t seed)
{
    const long long i = (long long) blockIdx.x * blockDim.x + threadIdx.x;
    const int c = blockIdx.y;
    if (i < rows && c < cols)
        A[(size_t) c * ld + i] = hash_uniform(seed, (uint64_t) c * (uint64_t) total_rows + (uint64_t) (i + row_off));
}

// partial sums for ||X - Y||_F^2 and ||Y||_F^2 where Y is either a stored matrix (Y != nullptr) or the
// hash generator above (Y == nullptr).  One (sumdiff, sumref) pair per block, summed on the host in
// block order (deterministic).
__global__ __launch_bounds__(256) void diff_norm_kernel(const double* __restrict__ X, int ldx,
                                                        const double* __restrict__ Y, int ldy,
                                                        long long rows, int cols, long long row_off,
                                                        long long total_rows, uint64_t seed, int sub_identity,
                                                        double* __restrict__ partials)
{
    __shared__ double sd[4], sr[4];
    double d = 0.0, rsum = 0.0;
    const int c = blockIdx.y;
    for (long long i = (long long) blockIdx.x * blockDim.x + threadIdx.x; i < rows; i += (long long) gridDim.x * blockDim.x) {
        const double x = X[(size_t) c * ldx + i];
        double y;
        if (sub_identity) y = (i == c) ? 1.0 : 0.0;
        else if (Y) y = Y[(size_t) c * ldy + i];
        else y = hash_uniform(seed, (uint64_t) c * (uint64_t) total_rows + (uint64_t) (i + row_off));
        d += (x - y) * (x - y);
        rsum += y * y;
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) { d += __shfl_xor(d, off); rsum += __shfl_xor(rsum, off); }
    if ((threadIdx.x & 63) == 0) { sd[threadIdx.x >> 6] = d; sr[threadIdx.x >> 6] = rsum; }
    __syncthreads();
    if (threadIdx.x == 0) {
        const size_t b = (size_t) blockIdx.y * gridDim.x + blockIdx.x;
        partials[2 * b] = sd[0] + sd[1] + sd[2] + sd[3];
        partials[2 * b + 1] = sr[0] + sr[1] + sr[2] + sr[3];
    }
}

// MFMA fp64 issue-rate probe: each wave runs `iters` x NACC independent accumulators back to back and
// stamps shader clock (s_memtime) and the 100 MHz wall counter (s_memrealtime) around the loop, so the
// sustained in-kernel clock is known next to the rate (DVFS: MI355X_MICROARCH.md 'DVFS give-back').
// a 4 x 4 grid of accumulators fed by 4 A and 4 B fragments (the register pattern of mma_tile), 16 independent
// v_mfma_f64_16x16x4_f64 per trip
__global__ __launch_bounds__(256) void mfma_grid_kernel(double* out, unsigned long long* stamps, int iters, double seed)
{
    v4d acc[4][4];
    double a[4], b[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        a[i] = seed + threadIdx.x * 1e-3 + i; b[i] = 1.0 - threadIdx.x * 1e-4 - i;
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = (v4d){0.0, 0.0, 0.0, 0.0};
    }
    const unsigned long long c0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j)       // inline asm: the builtin makes hipcc shuttle the accumulators VGPR<->AGPR every trip
                asm volatile("v_mfma_f64_16x16x4_f64 %0, %1, %2, %0" : "+v"(acc[i][j]) : "v"(a[i]), "v"(b[j]));
    }
    double s = 0.0;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) s += acc[i][j][0] + acc[i][j][1] + acc[i][j][2] + acc[i][j][3];
    const unsigned long long c1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    out[(size_t) blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0 && stamps) { stamps[2 * blockIdx.x] = c1 - c0; stamps[2 * blockIdx.x + 1] = r1 - r0; }
}

// f64 VALU FMA probe (the vector pipe has the same datasheet rate as the matrix pipe on CDNA4)
__global__ __launch_bounds__(256) void valu_peak_kernel(double* out, int iters, double seed)
{
    double a[16];
#pragma unroll
    for (int q = 0; q < 16; ++q) a[q] = seed + q + threadIdx.x * 1e-3;
    const double m = 1.0 - 1e-9, c = 1e-9;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int q = 0; q < 16; ++q) a[q] = a[q] * m + c;
    }
    double s = 0.0;
#pragma unroll
    for (int q = 0; q < 16; ++q) s += a[q];
    out[(size_t) blockIdx.x * blockDim.x + threadIdx.x] = s;
}

__global__ void stream_copy_kernel(const v2d* __restrict__ src, v2d* __restrict__ dst, size_t n2)
{
    for (size_t i = (size_t) blockIdx.x * blockDim.x + threadIdx.x; i < n2; i += (size_t) gridDim.x * blockDim.x)
        dst[i] = src[i];
}

// ================================================================================================
// extern "C" launch layer (declared in qr_device.h; the only thing qr_host.c sees of HIP)
// ================================================================================================
static inline int vec_ok(const void* p, int ld) { return (((uintptr_t) p) % 16 == 0) && (ld % 2 == 0); }

// Launch helpers.  The FAST instantiation runs on the largest tile-aligned interior of the problem; the
// ragged right / bottom strips (and everything, when operands are not 16-byte aligned or K % 16 != 0)
// go to the guarded instantiation as separate launches.
template <int TI, int TJ, bool FAST, int TAG = 0>
static int launch_nn1(hipStream_t s, int M, int N, int K, double alpha, const double* A, int lda,
                      const double* B, int ldb, double beta, double* C, int ldc)
{
    constexpr int BM = 32 * TI, BN = 32 * TJ;
    if (M <= 0 || N <= 0) return 0;
    const size_t shm = sizeof(double) * (2 * BK * (BM + 16) + 2 * BN * LDKF);
    dim3 grid((M + BM - 1) / BM, (N + BN - 1) / BN);
    hipLaunchKernelGGL((gemm_nn_kernel<TI, TJ, FAST, TAG>), grid, dim3(256), shm, s, M, N, K, alpha, A, lda, B, ldb, beta,
                       C, ldc);
    return (int) hipGetLastError();
}

template <int TI, int TJ, int TAG = 0>
static int launch_nn(hipStream_t s, int M, int N, int K, double alpha, const double* A, int lda,
                     const double* B, int ldb, double beta, double* C, int ldc)
{
    constexpr int BM = 32 * TI, BN = 32 * TJ;
    const bool al = vec_ok(A, lda) && vec_ok(B, ldb) && (K % BK) == 0;
    const int Mi = al ? (M / BM) * BM : 0, Ni = al ? (N / BN) * BN : 0;
    if (!(Mi > 0 && Ni > 0)) return launch_nn1<TI, TJ, false>(s, M, N, K, alpha, A, lda, B, ldb, beta, C, ldc);
    // small ragged products sit on the panel's critical path, where a second (edge) launch costs more than guarded loads
    // (round 6 re-measured the 0.4 GFLOP threshold: at 0.06 GFLOP 2000^2 loses 8 %, 4096^2 at nb 64 3 %; and an edge-tiles-inside-the-fast-grid
    // form like gemm_tn_kernel<.., MIXED> spills here -- 158 VGPRs -- and loses to both.  profiles/NOTES.md)
    if ((Mi < M || Ni < N) && (double) M * N * K < 4e8)
        return launch_nn1<TI, TJ, false>(s, M, N, K, alpha, A, lda, B, ldb, beta, C, ldc);
    int rc = launch_nn1<TI, TJ, true, TAG>(s, Mi, Ni, K, alpha, A, lda, B, ldb, beta, C, ldc);
    if (!rc && Ni < N)      /* right strip: all rows, columns [Ni, N) */
        rc = launch_nn1<TI, TJ, false>(s, M, N - Ni, K, alpha, A, lda, B + (size_t) Ni * ldb, ldb, beta,
                                       C + (size_t) Ni * ldc, ldc);
    if (!rc && Mi < M)      /* bottom strip: rows [Mi, M), columns [0, Ni) */
        rc = launch_nn1<TI, TJ, false>(s, M - Mi, Ni, K, alpha, A + Mi, lda, B, ldb, beta, C + Mi, ldc);
    return rc;
}

template <int TI, int TJ, bool FAST, int TAG = 0>
static int launch_tn1(hipStream_t s, int M, int N, int K, int ksplit, int kchunk, double alpha, const double* A,
                      int lda, const double* B, int ldb, double beta, double* C, int ldc, size_t slab_stride)
{
    constexpr int BM = 32 * TI, BN = 32 * TJ;
    if (M <= 0 || N <= 0) return 0;
    const size_t shm = sizeof(double) * (2 * BM * LDKF + 2 * BN * LDKF);
    dim3 grid((M + BM - 1) / BM, (N + BN - 1) / BN, ksplit);
    hipLaunchKernelGGL((gemm_tn_kernel<TI, TJ, FAST, TAG>), grid, dim3(256), shm, s, M, N, K, kchunk, alpha, A, lda, B, ldb,
                       beta, C, ldc, slab_stride);
    return (int) hipGetLastError();
}

template <int TI, int TJ, int TAG = 0>
static int launch_tn(hipStream_t s, int M, int N, int K, int ksplit, int kchunk, double alpha, const double* A,
                     int lda, const double* B, int ldb, double beta, double* C, int ldc, size_t slab_stride)
{
    constexpr int BM = 32 * TI, BN = 32 * TJ;
    const bool al = vec_ok(A, lda) && vec_ok(B, ldb) && (K % BK) == 0;
    const int Mi = al ? (M / BM) * BM : 0, Ni = al ? (N / BN) * BN : 0;
    if (!(Mi > 0 && Ni > 0))
        return launch_tn1<TI, TJ, false>(s, M, N, K, ksplit, kchunk, alpha, A, lda, B, ldb, beta, C, ldc, slab_stride);
    if constexpr (TI == 4 && TJ == 4 && TAG <= 1) {
        if (Mi < M || Ni < N) {         // ragged: one launch over all tiles, the edge tiles guarded (gemm_tn_kernel<.., MIXED>)
            const size_t shm = sizeof(double) * (2 * BM * LDKF + 2 * BN * LDKF);
            dim3 grid((M + BM - 1) / BM, (N + BN - 1) / BN, ksplit);
            hipLaunchKernelGGL((gemm_tn_kernel<TI, TJ, true, TAG, true>), grid, dim3(256), shm, s, M, N, K, kchunk, alpha, A, lda, B, ldb,
                               beta, C, ldc, slab_stride);
            return (int) hipGetLastError();
        }
    }
    int rc = launch_tn1<TI, TJ, true, TAG>(s, Mi, Ni, K, ksplit, kchunk, alpha, A, lda, B, ldb, beta, C, ldc, slab_stride);
    if (!rc && Ni < N)
        rc = launch_tn1<TI, TJ, false>(s, M, N - Ni, K, ksplit, kchunk, alpha, A, lda, B + (size_t) Ni * ldb, ldb, beta,
                                       C + (size_t) Ni * ldc, ldc, slab_stride);
    if (!rc && Mi < M)
        rc = launch_tn1<TI, TJ, false>(s, M - Mi, Ni, K, ksplit, kchunk, alpha, A + (size_t) Mi * lda, lda, B, ldb, beta,
                                       C + Mi, ldc, slab_stride);
    return rc;
}

// ------------------------------------------------------------------------------------------------
// W = T^T Y WITHOUT the merged T (round 6): for a compact-WY panel of nblk leaves of 32 columns, T^-1 = striu(G) + diag(1 / tau) with
// G = V^T V, so W solves the block lower-triangular system  (T^-T) W = Y  by forward substitution over the leaves,
//     W_i = T_ii^T (Y_i - sum_{j < i} G_ij W_j),        G_ij = V_i^T V_j (the block BELOW the diagonal), T_ii the leaf's own 32 x 32 T,
// which needs the panel's Gram matrix and the leaves' T blocks only: the merge tree (qrd_larft: six dependent launches at 256 columns)
// leaves the critical chain of a look-ahead step.  One workgroup per 16 columns of Y; wave w keeps the running right-hand sides of the
// leaves i = w, w + 4 in accumulator tiles (register r of a tile IS the B operand of k-step r: nothing moves between lanes); W_j goes
// round through LDS.  kw a multiple of 32 up to 256, nc a multiple of 16.  W may be Y.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void trsm_gt_kernel(int kw, const double* __restrict__ G, int ldg, const double* __restrict__ T, int ldt,
                                                      const double* Y, int ldy, double* W, int ldw)
{
    __shared__ double Ws[2][32][17];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l15 = lane & 15, l4 = lane >> 4;
    const int nblk = kw >> 5, c0 = blockIdx.x * 16;
    v4d acc[2][2];                                            // [own leaf slot s: leaf i = wave + 4 s][16-row tile t]
    double tf[2][2][8];                                       // the own leaves' T^T fragments (A operands of W_j = T_jj^T acc_j), requested up front
#pragma unroll
    for (int s = 0; s < 2; ++s) {
        const int i = wave + 4 * s, ic = i < nblk ? i : 0;
#pragma unroll
        for (int t = 0; t < 2; ++t) {
#pragma unroll
            for (int r = 0; r < 4; ++r) acc[s][t][r] = Y[(size_t) (c0 + l15) * ldy + 32 * ic + 16 * t + 4 * r + l4];
            const double* tc = T + (size_t) (32 * ic + 16 * t + l15) * ldt + 32 * ic;
#pragma unroll
            for (int ks = 0; ks < 8; ++ks) tf[s][t][ks] = tc[4 * ks + l4];
        }
    }
    // The Gram blocks G_ij of this wave's leaves i > j are requested ONE STEP AHEAD (they do not depend on W_j), and the barrier of a step is
    // an LDS-only one: __syncthreads() also drains the vector-memory counter, i.e. waits for the blocks just requested -- a memory latency
    // per step, 23-28 us per launch of eight steps
    double gf[2][2][2][8];                                    // [buffer][slot][tile][k-step]
    auto gload = [&](double (&g)[2][2][8], int j) __attribute__((always_inline)) {
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            const int i = wave + 4 * s, jc = j < nblk ? j : nblk - 1, ic = (i > jc && i < nblk) ? i : jc;   // (clamped: an inactive slot reads a valid block and ignores it)
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                const double* gr = G + (size_t) (32 * jc) * ldg + 32 * ic + 16 * t + l15;
#pragma unroll
                for (int ks = 0; ks < 8; ++ks) g[s][t][ks] = gr[(size_t) (4 * ks + l4) * ldg];
            }
        }
    };
    auto step = [&](const double (&g)[2][2][8], double (&gn)[2][2][8], int j) __attribute__((always_inline)) {
        gload(gn, j + 1);
        __builtin_amdgcn_sched_barrier(0);
        const int ow = j & 3, os = j >> 2;                    // the wave / slot that owns leaf j
        if (wave == ow) {
            // W_j = T_jj^T acc_j: A operand (T^T)(row 16 t' + l15, k) = T(32 j + k, 32 j + 16 t' + l15), B operand: the accumulator registers
            v4d o[2];
#pragma unroll
            for (int tp = 0; tp < 2; ++tp) {
                o[tp] = (v4d){0.0, 0.0, 0.0, 0.0};
#pragma unroll
                for (int ks = 0; ks < 8; ++ks) {
                    const int k = 4 * ks + l4;
                    const double tv = os ? tf[1][tp][ks] : tf[0][tp][ks];
                    const double a = (k <= 16 * tp + l15) ? tv : 0.0;             // T_jj is upper triangular
                    const double b = os ? acc[1][ks >> 2][ks & 3] : acc[0][ks >> 2][ks & 3];
                    o[tp] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, o[tp], 0, 0, 0);
                }
            }
#pragma unroll
            for (int tp = 0; tp < 2; ++tp)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int row = 16 * tp + 4 * r + l4;
                    Ws[j & 1][row][l15] = o[tp][r];
                    W[(size_t) (c0 + l15) * ldw + 32 * j + row] = o[tp][r];
                }
        }
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");          // W_j is in LDS for everybody; nothing waits for global memory here
        // every wave: its leaves i > j take  acc_i -= G_ij W_j
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            const int i = wave + 4 * s;
            if (i > j && i < nblk) {
#pragma unroll
                for (int t = 0; t < 2; ++t)
#pragma unroll
                    for (int ks = 0; ks < 8; ++ks)
                        acc[s][t] = __builtin_amdgcn_mfma_f64_16x16x4f64(-g[s][t][ks], Ws[j & 1][4 * ks + l4][l15], acc[s][t], 0, 0, 0);
            }
        }
        // (no second barrier: W_{j+1} goes to the other LDS buffer, and buffer j & 1 is rewritten at step j + 2, behind the next barrier)
    };
    gload(gf[0], 0);
    for (int j = 0; j < nblk; j += 2) {                      // (two steps per trip: the operand buffers alternate by name, not by a run-time index)
        step(gf[0], gf[1], j);
        if (j + 1 < nblk) step(gf[1], gf[0], j + 1);
    }
}

template <typename K>
static int allow_lds(K kern, size_t bytes)
{
    return (int) hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int) bytes);
}

extern "C" {

// Raise the dynamic-LDS cap of the big-tile kernels (they use 72 KiB; gfx950 has 160 KiB per CU).
int qrd_init(void)
{
    int rc = 0;
    rc |= allow_lds(gemm_nn_kernel<4, 4, true>, sizeof(double) * (2 * BK * (128 + 16) + 2 * 128 * LDKF));
    rc |= allow_lds(gemm_nn_kernel<4, 4, true, 1>, sizeof(double) * (2 * BK * (128 + 16) + 2 * 128 * LDKF));
    rc |= allow_lds(gemm_nn_w8_kernel<0>, sizeof(double) * (2 * BK * (128 + 16) + 2 * 128 * LDKF));
    rc |= allow_lds(gemm_nn_w8_kernel<1>, sizeof(double) * (2 * BK * (128 + 16) + 2 * 128 * LDKF));
    rc |= allow_lds(gemm_tn_kernel<4, 4, true, 1>, sizeof(double) * (4 * 128 * LDKF));
    rc |= allow_lds(gemm_tn_kernel<4, 4, true, 3>, sizeof(double) * (4 * 128 * LDKF));
    rc |= allow_lds(gemm_tn_kernel<4, 4, true, 4>, sizeof(double) * (4 * 128 * LDKF));
    rc |= allow_lds(gemm_tn_wide_kernel<1>, sizeof(double) * (2 * (128 + 256) * LDKF));
    rc |= allow_lds(gemm_nn_kernel<4, 4, false>, sizeof(double) * (2 * BK * (128 + 16) + 2 * 128 * LDKF));
    rc |= allow_lds(gemm_tn_kernel<4, 4, true>, sizeof(double) * (4 * 128 * LDKF));
    rc |= allow_lds(gemm_tn_kernel<4, 4, true, 0, true>, sizeof(double) * (4 * 128 * LDKF));
    rc |= allow_lds(gemm_tn_kernel<4, 4, true, 1, true>, sizeof(double) * (4 * 128 * LDKF));
    rc |= allow_lds(gemm_tn_kernel<4, 4, false>, sizeof(double) * (4 * 128 * LDKF));
    rc |= qrd_gemm2_init();
    rc |= qrd_panel_tsqr_init();
    rc |= qrd_leaf_fused_init();
    rc |= qrd_panel_fused_init();
    rc |= qrd_panel_cqr_init();
    return rc;
}

// the wide trailing update A2 -= V*W: always the 128x128 tile, its own kernel name (TAG = 1) for the profiler
// (the 8-wave kernel serves the wide update, see gemm_nn_w8_kernel; the knob that selected the 4-wave one was folded in round 4)
static int nn_waves(void)
{
    static const int v = 8;
    return v;
}

static int gemm_nn_update_impl(void* stream, int M, int N, int K, double alpha, const double* A, int lda, const double* B,
                               int ldb, double beta, double* C, int ldc, int tag);

int qrd_gemm_nn_update(void* stream, int M, int N, int K, double alpha, const double* A, int lda, const double* B,
                       int ldb, double beta, double* C, int ldc)
{
    return gemm_nn_update_impl(stream, M, N, K, alpha, A, lda, B, ldb, beta, C, ldc, 0);
}

// the same 8-wave kernel for tall C -= A*B products outside the wide update (look-ahead / mid-panel updates)
int qrd_gemm_nn_update2(void* stream, int M, int N, int K, double alpha, const double* A, int lda, const double* B,
                        int ldb, double beta, double* C, int ldc)
{
    return gemm_nn_update_impl(stream, M, N, K, alpha, A, lda, B, ldb, beta, C, ldc, 1);
}

static int gemm_nn_update_impl(void* stream, int M, int N, int K, double alpha, const double* A, int lda, const double* B,
                               int ldb, double beta, double* C, int ldc, int tag)
{
    if (M <= 0 || N <= 0 || K <= 0) return 0;
    hipStream_t s = (hipStream_t) stream;
    const bool al = vec_ok(A, lda) && vec_ok(B, ldb) && (K % BK) == 0;
    const int Mi = (M / 128) * 128, Ni = (N / 128) * 128;
    if (nn_waves() != 8 || beta != 1.0 || (alpha != 1.0 && alpha != -1.0) || !al || Mi == 0 || Ni == 0)
        return launch_nn<4, 4, 1>(s, M, N, K, alpha, A, lda, B, ldb, beta, C, ldc);
    const size_t shm = sizeof(double) * (2 * BK * (128 + 16) + 2 * 128 * LDKF);
    if (tag == 0)
        hipLaunchKernelGGL(gemm_nn_w8_kernel<0>, dim3(Mi / 128, Ni / 128), dim3(512), shm, s, Mi, Ni, K, alpha, A, lda, B, ldb, C, ldc);
    else
        hipLaunchKernelGGL(gemm_nn_w8_kernel<1>, dim3(Mi / 128, Ni / 128), dim3(512), shm, s, Mi, Ni, K, alpha, A, lda, B, ldb, C, ldc);
    int rc = (int) hipGetLastError();
    if (!rc && Ni < N)      /* right strip: all rows, columns [Ni, N) */
        rc = launch_nn1<4, 4, false>(s, M, N - Ni, K, alpha, A, lda, B + (size_t) Ni * ldb, ldb, beta, C + (size_t) Ni * ldc, ldc);
    if (!rc && Mi < M)      /* bottom strip: rows [Mi, M), columns [0, Ni) */
        rc = launch_nn1<4, 4, false>(s, M - Mi, Ni, K, alpha, A + Mi, lda, B, ldb, beta, C + Mi, ldc);
    return rc;
}

int qrd_gemm_nn(void* stream, int M, int N, int K, double alpha, const double* A, int lda, const double* B,
                int ldb, double beta, double* C, int ldc)
{
    hipStream_t s = (hipStream_t) stream;
    if (M <= 0 || N <= 0) return 0;
    if (K <= 0) {   // C = beta*C
        if (beta == 1.0) return 0;
        return -1;
    }
    if (alpha == 0.0 && beta == 1.0) return 0;      // C unchanged (and no 0 * inf from the C/alpha shortcut)
    // tile choice: big square tiles when the grid still fills the chip, smaller otherwise
    const long long t44 = (long long) ((M + 127) / 128) * ((N + 127) / 128);
    {
        static const int tall_tile = 44;
        if (tall_tile == 22 && M >= 65536 && N <= 512 && N > 32) return launch_nn<2, 2>(s, M, N, K, alpha, A, lda, B, ldb, beta, C, ldc);
    }
    if (N > 64 && M > 64 && t44 >= 192) return launch_nn<4, 4>(s, M, N, K, alpha, A, lda, B, ldb, beta, C, ldc);
    if (N <= 32) {
        if (M >= 128 * 96) return launch_nn<4, 1>(s, M, N, K, alpha, A, lda, B, ldb, beta, C, ldc);
        return launch_nn<1, 1>(s, M, N, K, alpha, A, lda, B, ldb, beta, C, ldc);
    }
    const long long t22 = (long long) ((M + 63) / 64) * ((N + 63) / 64);
    if (t22 >= 192) return launch_nn<2, 2>(s, M, N, K, alpha, A, lda, B, ldb, beta, C, ldc);
    return launch_nn<1, 1>(s, M, N, K, alpha, A, lda, B, ldb, beta, C, ldc);
}

// C = alpha*A^T*B + beta*C with split-K through `slabs` (capacity slab_cap doubles).
// Tm (optional, M <= 256): C = beta*C + Tm^T * (alpha*A^T*B)   [leaf-level T^T fold].
static int gemm_tn_impl(void* stream, int M, int N, int K, double alpha, const double* A, int lda, const double* B,
                        int ldb, double beta, double* C, int ldc, double* slabs, size_t slab_cap, const double* Tm,
                        int ldt, int tag);

// compute units behind a stream (CU-masked streams register themselves; anything else: the whole device)
// (one table for the process, any device, any host thread: guarded by g_stream_mutex)
#define QRD_MAX_MASKED_STREAMS 256
static struct { hipStream_t s; int cus; } g_stream_cus[QRD_MAX_MASKED_STREAMS];
static int g_nstream_cus = 0;
static std::mutex g_stream_mutex;
static int stream_cus(hipStream_t s)
{
    {
        std::lock_guard<std::mutex> lock(g_stream_mutex);
        for (int i = 0; i < g_nstream_cus; ++i)
            if (g_stream_cus[i].s == s) return g_stream_cus[i].cus;
    }
    int dev = 0, cus = 0;
    if (hipGetDevice(&dev) == hipSuccess && hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && cus > 0)
        return cus;
    return 256;
}


int qrd_gemm_tn(void* stream, int M, int N, int K, double alpha, const double* A, int lda, const double* B,
                int ldb, double beta, double* C, int ldc, double* slabs, size_t slab_cap, const double* Tm,
                int ldt)
{
    return gemm_tn_impl(stream, M, N, K, alpha, A, lda, B, ldb, beta, C, ldc, slabs, slab_cap, Tm, ldt, 0);
}

// [W | G2^T] = A^T [B1 | B2] for a 32-column A (the leaf's reflectors):  W (32 x N1, ld ldw) = Tm^T A^T B1,  G2 (N2 x 32, ld ldg) =
// B2^T A.  One product launch + one reduce launch.  Returns -7 when the shapes do not fit the fast kernel (caller falls back to
// two ordinary products): N1, N2 multiples of 32, K a multiple of the k-tile, 16-byte aligned operands.
int qrd_gemm_tn_dual(void* stream, int N1, int N2, int K, const double* A, int lda, const double* B1, int ldb1, const double* B2,
                     int ldb2, const double* Tm, int ldt, double* W, int ldw, double* G2, int ldg, double* slabs, size_t slab_cap)
{
    hipStream_t s = (hipStream_t) stream;
    const int N = N1 + N2;
    if (N <= 0) return 0;
    if (N1 % 32 || N2 % 32 || K % BK || K < BK || !vec_ok(A, lda) || (N1 > 0 && !vec_ok(B1, ldb1)) || (N2 > 0 && !vec_ok(B2, ldb2)))
        return -7;
    const size_t per = (size_t) 32 * N;
    if (!slabs || slab_cap < per) return -7;
    // workgroup slots per compute unit: the 32 x 32 tile kernel is latency-bound per workgroup (18 KB of LDS, 4 waves), so more
    // and shorter K slices than the 2 per CU of the big-tile products pay (8192^2: 34.4 ms at 2 slots / >= 8 k-tiles per slice,
    // 33.3 ms at 8 slots / >= 4 k-tiles)
    static const int spc = 8;
    static const int kmin_tiles = 4;
    // (round 3: a streaming form of this product for tall leaves -- both operands straight from global memory in MFMA operand layout,
    // 32 B per lane, all outputs of a column group in accumulators, no LDS / barrier in the loop -- was built and measured at 100-109 us
    // per launch on a 262144-row leaf against 78 us here: 128-byte-per-column accesses from 8 waves per CU stream worse than this
    // kernel's LDS-staged tiles from 32 waves per CU; and the 32 x 64 tiles of the fused launch (ep_fused_item) as a launch of their own:
    // 87 us.  Both removed; profiles/r03_tn_stream_negative.txt.)
    const int tiles = N / 32, slots = spc * stream_cus(s);
    long long kmax = (K + kmin_tiles * BK - 1) / (kmin_tiles * BK);
    if (kmax > 256) kmax = 256;
    if ((size_t) kmax * per > slab_cap) kmax = (long long) (slab_cap / per);
    if (kmax < 1) kmax = 1;
    // same cost model as gemm_tn_impl: rounds * (K rows per slice + fixed) + reduce
    const double row_us = 2.0 * 32 * 32 / 0.113e6, red_rows = (double) per * 8.0 / 2.0e6 / row_us;
    int ksplit = 1;
    double best = 1e300;
    for (long long k = 1; k <= kmax; ++k) {
        const long long rounds = ((long long) tiles * k + slots - 1) / slots;
        const double cost = (double) rounds * ((double) K / (double) k + 96.0) + (double) k * red_rows + 30.0;
        if (cost < best) { best = cost; ksplit = (int) k; }
    }
    int kchunk = ((K + ksplit - 1) / ksplit + BK - 1) / BK * BK;
    ksplit = (K + kchunk - 1) / kchunk;
    const size_t shm = sizeof(double) * (4 * 32 * LDKF);
    hipLaunchKernelGGL(gemm_tn_dual_kernel, dim3(tiles * 8 * ((ksplit + 7) / 8)), dim3(256), shm, s, N1, N2, K, kchunk, tiles, ksplit, A, lda,
                       B1, ldb1, B2, ldb2, slabs, per);
    hipLaunchKernelGGL(slab_reduce_kernel, dim3(N, 1), dim3(256), 0, s, 32, N, ksplit, slabs, 32, per, Tm, ldt, 0.0, W, ldw, 256, N1,
                       G2, ldg);
    return (int) hipGetLastError();
}

// the wide W = (V T)^T A2 of the trailing update: same kernel under its own profiler name (TAG = 1)
// out (M x N, ldo) = sum of nslab slabs (slab z at slabs + z*stride, ld lds), fixed order
int qrd_slab_reduce(void* stream, int M, int N, int nslab, const double* slabs, int lds, size_t stride, double* out, int ldo)
{
    hipLaunchKernelGGL(slab_reduce_kernel, dim3(N, (M + 255) / 256), dim3(256), 0, (hipStream_t) stream, M, N, nslab, slabs, lds, stride,
                       (const double*) nullptr, 0, 0.0, out, ldo, 256, N, (double*) nullptr, 0);
    return (int) hipGetLastError();
}

int qrd_gemm_tn_update(void* stream, int M, int N, int K, double alpha, const double* A, int lda, const double* B,
                       int ldb, double beta, double* C, int ldc, double* slabs, size_t slab_cap)
{
    return gemm_tn_impl(stream, M, N, K, alpha, A, lda, B, ldb, beta, C, ldc, slabs, slab_cap, nullptr, 0, 1);
}

// the wide-tile form of the same product, whatever MI355XQR_TN_WIDE says (kernel tests, A/B measurements)
int qrd_gemm_tn_update_wide(void* stream, int M, int N, int K, double alpha, const double* A, int lda, const double* B,
                            int ldb, double beta, double* C, int ldc, double* slabs, size_t slab_cap)
{
    return gemm_tn_impl(stream, M, N, K, alpha, A, lda, B, ldb, beta, C, ldc, slabs, slab_cap, nullptr, 0, 2);
}

static int gemm_tn_impl(void* stream, int M, int N, int K, double alpha, const double* A, int lda, const double* B,
                        int ldb, double beta, double* C, int ldc, double* slabs, size_t slab_cap, const double* Tm,
                        int ldt, int tag)
{
    hipStream_t s = (hipStream_t) stream;
    if (M <= 0 || N <= 0) return 0;
    if (Tm && M > 256) return -2;
    // K not a multiple of the k-tile (matrix heights that are not multiples of 16): the fast kernels cannot take it, and a guarded K loop over
    // the whole product costs a third of its rate (8200^2: the wide product at 38 TFLOP/s against 55).  Large products are split instead: the
    // whole k-tiles through the fast path, the last K % 16 rows as a tiny guarded product accumulated behind it (round 6: 10000^2 45.0 -> 40.0 ms,
    // the wide product 32.7 -> 45.7 TFLOP/s; from 4 GFLOP on -- 8200 x 2056 at nb 128, 2 GFLOP per product, loses 3 % to the extra launch)
    if (K % BK != 0 && K >= 16 * BK && Tm == nullptr && vec_ok(A, lda) && vec_ok(B, ldb) && (double) M * N * K >= 4e9) {
        const int Kf = K - K % BK;
        const int rc0 = gemm_tn_impl(stream, M, N, Kf, alpha, A, lda, B, ldb, beta, C, ldc, slabs, slab_cap, nullptr, 0, tag);
        if (rc0) return rc0;
        return gemm_tn_impl(stream, M, N, K - Kf, alpha, A + Kf, lda, B + Kf, ldb, 1.0, C, ldc, nullptr, 0, nullptr, 0, 0);
    }
    int ti, tj;
    const bool shortk = (K <= 512 && Tm == nullptr);     // e.g. W = T^T Y: one K slice, small tiles for parallelism
    if (M <= 32) { ti = 1; tj = (N >= 128 && N % 128 == 0) ? 4 : 1; }   // keep the FAST (unguarded) instantiation
    else if (shortk && (long long) ((M + 31) / 32) * ((N + 31) / 32) <= 1024) { ti = 1; tj = 1; }
    else if (M <= 64 || N <= 64 || (shortk && (long long) ((M + 63) / 64) * ((N + 63) / 64) <= 2048)) { ti = 2; tj = 2; }
    else { ti = 4; tj = 4; }
    {
        static const int tall_tile = 44;
        if (ti == 4 && tall_tile == 22 && K >= 65536 && (long long) M * N <= 512 * 512) { ti = 2; tj = 2; }
    }
    // the wide product of the trailing update on 128 x 256 workgroup tiles (gemm_tn_wide_kernel: A2 read N / 256 times instead of
    // N / 128) -- measured in round 3: half the HBM traffic, the same rate (16384^2: 50.7 against 51.4 TFLOP/s in situ; the product
    // is bound on the matrix-core side, not by bytes), so the 128 x 128 kernel stays the default; MI355XQR_TN_WIDE=1 or tag 2 select it
    static const int wide_ok = QRD_LAB_ENV_INT("MI355XQR_TN_WIDE", 0) != 0;
    const bool wide = (tag == 2 || (tag == 1 && wide_ok)) && ti == 4 && Tm == nullptr && alpha == 1.0 && beta == 0.0 && N % 256 == 0 && M >= 128 &&
                      K % BK == 0 && vec_ok(A, lda) && vec_ok(B, ldb) && slabs != nullptr;
    const int BM = 32 * ti, BN = wide ? 256 : 32 * tj;
    const long long tiles = (long long) ((M + BM - 1) / BM) * ((N + BN - 1) / BN);
    const size_t per = (size_t) M * N;
    // K split: minimise  rounds(k) * (K/k + fixed) + reduce(k)  over k, where rounds = ceil(tiles*k / slots), slots =
    // 2 workgroups per compute unit OF THE STREAM (a CU-masked stream has fewer than the device), `fixed` the per-workgroup
    // overhead in K rows and reduce(k) the slab traffic of slab_reduce_kernel.  (The old rule aimed at 512 workgroups
    // whatever the stream: 624 workgroups on the 192-CU update stream = 1.6 rounds.)
    long long kmax = (K + 8 * BK - 1) / (8 * BK);             // each K slice at least 8 k-tiles long
    static const int kcap = 256;
    if (kmax > kcap) kmax = kcap;
    if (slabs == nullptr || slab_cap < per) kmax = 1;
    else if ((size_t) kmax * per > slab_cap) kmax = (long long) (slab_cap / per);
    if (kmax < 1 || shortk) kmax = 1;
    // K loop of the 128 x 128 products: MI355XQR_KPIPE=1 (default) issue order spelled out, one memory instruction behind every MFMA
    // (gemm_kloop_il); 0 = the plain double-buffered loop.  Isolated, 15872 x 256 x 16128: 65.8 -> 68.9 TFLOP/s; one workgroup per CU:
    // 58.4 -> 68.1.  In situ at 16384^2: 51.5 -> 54.3 TFLOP/s (profiles/r03_tn_issue_order.txt)
    static const int kpipe = QRD_LAB_ENV_INT("MI355XQR_KPIPE", 1);
    const int slots = (wide ? 1 : 2) * stream_cus(s);
    const double row_us = 2.0 * BM * BN / (wide ? 0.226e6 : 0.113e6);   // one K row of one tile on one workgroup slot (wide: the whole CU)
    const double red_rows = (double) per * 8.0 / 2.0e6 / row_us;   // reading one slab of the output at ~2 TB/s, in K rows
    int ksplit = 1;
    double best = 1e300;
    for (long long k = 1; k <= kmax; ++k) {
        const long long wgs = tiles * k, rounds = (wgs + slots - 1) / slots;
        const double cost = (double) rounds * ((double) K / (double) k + 96.0) + (k > 1 ? (double) k * red_rows + 30.0 : 0.0);
        if (cost < best) { best = cost; ksplit = (int) k; }
    }
    int kchunk = ((K + ksplit - 1) / ksplit + BK - 1) / BK * BK;
    ksplit = (K + kchunk - 1) / kchunk;
    if (ksplit < 1) ksplit = 1;
    const bool direct = (ksplit == 1 && Tm == nullptr);
    double* dst = direct ? C : slabs;
    const int ldd = direct ? ldc : M;
    const double b2 = direct ? beta : 0.0;
    if (!direct && (slabs == nullptr || slab_cap < per)) return -3;
    int rc;
    if (ti == 1 && tj == 4) rc = launch_tn<1, 4>(s, M, N, K, ksplit, kchunk, alpha, A, lda, B, ldb, b2, dst, ldd, per);
    else if (ti == 1) rc = launch_tn<1, 1>(s, M, N, K, ksplit, kchunk, alpha, A, lda, B, ldb, b2, dst, ldd, per);
    else if (ti == 2) rc = launch_tn<2, 2>(s, M, N, K, ksplit, kchunk, alpha, A, lda, B, ldb, b2, dst, ldd, per);
    else if (wide) {
        // interior rows (multiples of 128) on the wide kernel, the ragged rest on the guarded 128 x 128 kernel, same K slices
        const int Mi = (M / 128) * 128;
        hipLaunchKernelGGL(gemm_tn_wide_kernel<1>, dim3(Mi / 128, N / 256, ksplit), dim3(512), sizeof(double) * (2 * (128 + 256) * LDKF), s,
                           K, kchunk, A, lda, B, ldb, dst, ldd, per);
        rc = (int) hipGetLastError();
        if (!rc && Mi < M)
            rc = launch_tn1<4, 4, false>(s, M - Mi, N, K, ksplit, kchunk, alpha, A + (size_t) Mi * lda, lda, B, ldb, b2, dst + Mi, ldd, per);
    }
    else if (tag >= 1 && kpipe) rc = launch_tn<4, 4, 1>(s, M, N, K, ksplit, kchunk, alpha, A, lda, B, ldb, b2, dst, ldd, per);
    else if (tag >= 1) rc = launch_tn<4, 4, 4>(s, M, N, K, ksplit, kchunk, alpha, A, lda, B, ldb, b2, dst, ldd, per);
    else if (kpipe) rc = launch_tn<4, 4>(s, M, N, K, ksplit, kchunk, alpha, A, lda, B, ldb, b2, dst, ldd, per);
    else rc = launch_tn<4, 4, 3>(s, M, N, K, ksplit, kchunk, alpha, A, lda, B, ldb, b2, dst, ldd, per);
    if (rc) return rc;
    if (!direct) {
        // many slabs (tall-skinny products): shorter row pieces per workgroup, so that more threads share the slab index
        int piece = 256;
        if (Tm == nullptr && ksplit >= 32 && M >= 64) piece = ksplit >= 128 ? 32 : 64;
        hipLaunchKernelGGL(slab_reduce_kernel, dim3(N, (M + piece - 1) / piece), dim3(256), 0, s, M, N, ksplit, slabs, M, per, Tm, ldt,
                           beta, C, ldc, piece, N, (double*) nullptr, 0);
        rc = (int) hipGetLastError();
    }
    return rc;
}

int qrd_gemm_nn_batch(void* stream, int M, int N, int K, double alpha, const double* A, int lda, size_t sA,
                      const double* B, int ldb, size_t sB, double beta, double* C, int ldc, size_t sC, int batch)
{
    if (M <= 0 || N <= 0 || K <= 0 || batch <= 0) return 0;
    const size_t shm = sizeof(double) * (2 * BK * (32 + 16) + 2 * 32 * LDKF);
    dim3 grid((M + 31) / 32, (N + 31) / 32, batch);
    const bool fast = vec_ok(A, lda) && vec_ok(B, ldb) && (sA % 2) == 0 && (sB % 2) == 0 && (K % BK) == 0 && (M % 32) == 0 &&
                      (N % 32) == 0;
    if (fast)
        hipLaunchKernelGGL((gemm_nn_batch_kernel<1, 1, true>), grid, dim3(256), shm, (hipStream_t) stream, M, N, K, alpha, A,
                           lda, sA, B, ldb, sB, beta, C, ldc, sC);
    else
        hipLaunchKernelGGL((gemm_nn_batch_kernel<1, 1, false>), grid, dim3(256), shm, (hipStream_t) stream, M, N, K, alpha, A,
                           lda, sA, B, ldb, sB, beta, C, ldc, sC);
    return (int) hipGetLastError();
}

// Outer-panel T (nbp x nbp, leaves of width ib) from G = V^T V.  build_diag: diagonal blocks are rebuilt
// from G and tau (otherwise the leaf kernels already left them in T).  X: scratch of nbp*nbp/2 doubles.
// Blocks are merged pairwise up a binary tree,  T = [T1, -T1 (V1^T V2) T2; 0, T2],  all pairs of a level in
// two batched launches: 2*log2(nbp/ib) dependent launches instead of 2*(nbp/ib - 1).
int qrd_larft(void* stream, int nbp, int ib, const double* G, int ldg, const double* tau, double* T, int ldt,
              double* Tt, int build_diag, double* X, int ldx)
{
    (void) ldx;
    hipStream_t s = (hipStream_t) stream;
    if (ib > LEAFW || nbp < 1) return -5;
    if (build_diag) {
        hipLaunchKernelGGL(leaf_t_kernel, dim3((nbp + ib - 1) / ib), dim3(64), 0, s, nbp, ib, G, ldg, tau, T, ldt);
        int rc = (int) hipGetLastError();
        if (rc) return rc;
    }
    for (int sz = ib; sz < nbp; sz *= 2) {
        const int nf = nbp / (2 * sz), rem = nbp - 2 * sz * nf;
        if (nf > 0) {        // full pairs: left [o, o+sz), right [o+sz, o+2sz), o = 2*sz*p
            int rc = qrd_gemm_nn_batch(stream, sz, sz, sz, 1.0, G + (size_t) sz * ldg, ldg, (size_t) 2 * sz * (ldg + 1),
                                       T + (size_t) sz * ldt + sz, ldt, (size_t) 2 * sz * (ldt + 1), 0.0, X, sz,
                                       (size_t) sz * sz, nf);
            if (rc) return rc;
            rc = qrd_gemm_nn_batch(stream, sz, sz, sz, -1.0, T, ldt, (size_t) 2 * sz * (ldt + 1), X, sz, (size_t) sz * sz, 0.0,
                                   T + (size_t) sz * ldt, ldt, (size_t) 2 * sz * (ldt + 1), nf);
            if (rc) return rc;
        }
        if (rem > sz) {      // ragged last pair: left sz, right rem - sz
            const int o = 2 * sz * nf, w2 = rem - sz;
            double* Xr = X + (size_t) nf * sz * sz;
            int rc = qrd_gemm_nn_batch(stream, sz, w2, w2, 1.0, G + (size_t) (o + sz) * ldg + o, ldg, 0,
                                       T + (size_t) (o + sz) * ldt + o + sz, ldt, 0, 0.0, Xr, sz, 0, 1);
            if (rc) return rc;
            rc = qrd_gemm_nn_batch(stream, sz, w2, sz, -1.0, T + (size_t) o * ldt + o, ldt, 0, Xr, sz, 0, 0.0,
                                   T + (size_t) (o + sz) * ldt + o, ldt, 0, 1);
            if (rc) return rc;
        }
    }
    if (Tt) {
        hipLaunchKernelGGL(transpose_kernel, dim3((nbp * nbp + 255) / 256), dim3(256), 0, s, nbp, T, ldt, Tt, ldt);
        return (int) hipGetLastError();
    }
    return 0;
}

// W (kw x nc, ldw) = T^T Y for the panel whose Gram matrix is G and whose leaves' T blocks sit on the diagonal of T (trsm_gt_kernel): -7 when
// the shape is not whole leaves x whole 16-column tiles (the caller then merges T and multiplies)
int qrd_trsm_gt(void* stream, int kw, int nc, const double* G, int ldg, const double* T, int ldt, const double* Y, int ldy, double* W, int ldw)
{
    if (kw < 32 || kw > 256 || kw % 32 || nc < 16 || nc % 16) return -7;
    hipLaunchKernelGGL(trsm_gt_kernel, dim3(nc / 16), dim3(256), 0, (hipStream_t) stream, kw, G, ldg, T, ldt, Y, ldy, W, ldw);
    return (int) hipGetLastError();
}

int qrd_transpose(void* stream, int rows, int cols, const double* S, int lds, double* D, int ldd)
{
    if (rows <= 0 || cols <= 0) return 0;
    hipLaunchKernelGGL(transpose_rect_kernel, dim3((unsigned) (((size_t) rows * cols + 255) / 256)), dim3(256), 0, (hipStream_t) stream, rows, cols, S, lds, D, ldd);
    return (int) hipGetLastError();
}

int qrd_zero_block(void* stream, double* A, int ld, int rows, int cols)
{
    if (rows <= 0 || cols <= 0) return 0;
    const size_t n = (size_t) rows * cols;
    hipLaunchKernelGGL(zero_block_kernel, dim3((unsigned) ((n + 255) / 256)), dim3(256), 0, (hipStream_t) stream, A,
                       ld, rows, cols);
    return (int) hipGetLastError();
}

int qrd_extract_v(void* stream, const double* P, int ld, int mk, int w, double* V, int ldv)
{
    hipLaunchKernelGGL(extract_v_kernel, dim3((mk + 255) / 256, w), dim3(256), 0, (hipStream_t) stream, P, ld, mk, w,
                       V, ldv);
    return (int) hipGetLastError();
}

int qrd_extract_r(void* stream, const double* A, int lda, int m, int n, double* R, int ldr, int rrows)
{
    hipLaunchKernelGGL(extract_r_kernel, dim3((rrows + 255) / 256, n), dim3(256), 0, (hipStream_t) stream, A, lda, m,
                       n, R, ldr, rrows);
    return (int) hipGetLastError();
}

int qrd_extract_r_block(void* stream, const double* A, int lda, int k, int w, double* R, int ldr, int rrows)
{
    if (w <= 0 || rrows <= 0) return 0;
    hipLaunchKernelGGL(extract_r_block_kernel, dim3((rrows + 255) / 256, w), dim3(256), 0, (hipStream_t) stream, A, lda, k, w, R, ldr, rrows);
    return (int) hipGetLastError();
}

int qrd_set_identity(void* stream, double* C, int ld, int rows, int cols, int row_off)
{
    if (rows <= 0 || cols <= 0) return 0;
    hipLaunchKernelGGL(set_identity_kernel, dim3((rows + 255) / 256, cols), dim3(256), 0, (hipStream_t) stream, C, ld,
                       rows, cols, row_off);
    return (int) hipGetLastError();
}

int qrd_copy_block(void* stream, const double* S, int lds, double* D, int ldd, int rows, int cols)
{
    if (rows <= 0 || cols <= 0) return 0;
    hipLaunchKernelGGL(copy_block_kernel, dim3((rows + 255) / 256, cols), dim3(256), 0, (hipStream_t) stream, S, lds,
                       D, ldd, rows, cols);
    return (int) hipGetLastError();
}

int qrd_copy_blocks(void* stream, const double* S, int lds, size_t sstride, double* D, int ldd, size_t dstride, int rows, int cols, int batch)
{
    if (rows <= 0 || cols <= 0 || batch <= 0) return 0;
    hipLaunchKernelGGL(copy_blocks_kernel, dim3((rows + 255) / 256, cols, batch), dim3(256), 0, (hipStream_t) stream, S, lds, sstride, D, ldd, dstride,
                       rows, cols);
    return (int) hipGetLastError();
}

int qrd_fill_uniform(void* stream, double* A, int ld, long long rows, int cols, long long row_off,
                     long long total_rows, unsigned long long seed)
{
    if (rows <= 0 || cols <= 0) return 0;
    hipLaunchKernelGGL(fill_uniform_kernel, dim3((unsigned) ((rows + 255) / 256), cols), dim3(256), 0,
                       (hipStream_t) stream, A, ld, rows, cols, row_off, total_rows, (uint64_t) seed);
    return (int) hipGetLastError();
}

double qrd_hash_uniform_host(unsigned long long seed, unsigned long long idx) { return hash_uniform(seed, idx); }

// out[0] = sum (X - Y)^2, out[1] = sum Y^2  (Y stored, generated from the hash, or the identity)
int qrd_diff_norm(void* stream, const double* X, int ldx, const double* Y, int ldy, long long rows, int cols,
                  long long row_off, long long total_rows, unsigned long long seed, int sub_identity, double* out)
{
    hipStream_t s = (hipStream_t) stream;
    int gx = (int) ((rows + 255) / 256);
    if (gx > 64) gx = 64;
    if (gx < 1) gx = 1;
    const size_t nb = (size_t) gx * cols;
    double* dpart = nullptr;
    HIPCHK(hipMalloc(&dpart, sizeof(double) * 2 * nb));
    hipLaunchKernelGGL(diff_norm_kernel, dim3(gx, cols), dim3(256), 0, s, X, ldx, Y, ldy, rows, cols, row_off,
                       total_rows, (uint64_t) seed, sub_identity, dpart);
    double* h = (double*) malloc(sizeof(double) * 2 * nb);
    hipError_t e = hipMemcpyAsync(h, dpart, sizeof(double) * 2 * nb, hipMemcpyDeviceToHost, s);
    if (e == hipSuccess) e = hipStreamSynchronize(s);
    double a = 0.0, b = 0.0;
    if (e == hipSuccess)
        for (size_t i = 0; i < nb; ++i) { a += h[2 * i]; b += h[2 * i + 1]; }
    free(h);
    hipFree(dpart);
    out[0] = a; out[1] = b;
    return (int) e;
}

// --- memory / stream / event plumbing (so that qr_host.c stays plain C with no HIP headers) ---
int qrd_malloc(void** p, size_t bytes) { return (int) hipMalloc(p, bytes ? bytes : 16); }
int qrd_free(void* p) { return (int) hipFree(p); }
int qrd_memset(void* stream, void* p, int v, size_t bytes) { return (int) hipMemsetAsync(p, v, bytes, (hipStream_t) stream); }
int qrd_h2d(void* stream, void* d, const void* h, size_t bytes) { return (int) hipMemcpyAsync(d, h, bytes, hipMemcpyHostToDevice, (hipStream_t) stream); }
int qrd_d2h(void* stream, void* h, const void* d, size_t bytes) { return (int) hipMemcpyAsync(h, d, bytes, hipMemcpyDeviceToHost, (hipStream_t) stream); }
int qrd_d2d(void* stream, void* dst, const void* src, size_t bytes) { return (int) hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToDevice, (hipStream_t) stream); }
int qrd_h2d_2d(void* stream, void* d, size_t dpitch, const void* h, size_t hpitch, size_t width, size_t height)
{ return (int) hipMemcpy2DAsync(d, dpitch, h, hpitch, width, height, hipMemcpyHostToDevice, (hipStream_t) stream); }
int qrd_d2h_2d(void* stream, void* h, size_t hpitch, const void* d, size_t dpitch, size_t width, size_t height)
{ return (int) hipMemcpy2DAsync(h, hpitch, d, dpitch, width, height, hipMemcpyDeviceToHost, (hipStream_t) stream); }
int qrd_stream_create(void** s, int high_priority)
{
    hipStream_t st = nullptr;
    int lo = 0, hi = 0;
    hipError_t e;
    *s = nullptr;
    if (high_priority && hipDeviceGetStreamPriorityRange(&lo, &hi) == hipSuccess)
        e = hipStreamCreateWithPriority(&st, hipStreamNonBlocking, hi);
    else
        e = hipStreamCreateWithFlags(&st, hipStreamNonBlocking);
    if (e == hipSuccess) *s = (void*) st;       // the output stays NULL on failure: nothing to destroy
    return (int) e;
}
// Stream restricted to CUs [first, first+count) of the device (count = 0: no restriction).  Used to give
// the latency-critical panel chain its own compute units while the wide update saturates the rest.
int qrd_stream_create_cumask(void** s, int first, int count)
{
    hipStream_t st = nullptr;
    *s = nullptr;
    if (count <= 0) { hipError_t e = hipStreamCreateWithFlags(&st, hipStreamNonBlocking); if (e == hipSuccess) *s = (void*) st; return (int) e; }
    uint32_t mask[16] = {0};
    for (int c = first; c < first + count && c < 512; ++c) mask[c >> 5] |= 1u << (c & 31);
    int dev = 0, cus = 0;
    HIPCHK(hipGetDevice(&dev)); HIPCHK(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev));
    const uint32_t words = (uint32_t) ((cus + 31) / 32);
    hipError_t e = hipExtStreamCreateWithCUMask(&st, words, mask);
    if (e != hipSuccess) return (int) e;
    *s = (void*) st;
    std::lock_guard<std::mutex> lock(g_stream_mutex);
    if (g_nstream_cus < QRD_MAX_MASKED_STREAMS) { g_stream_cus[g_nstream_cus].s = st; g_stream_cus[g_nstream_cus].cus = count; ++g_nstream_cus; }
    return 0;
}
// Which (XCC, SE, CU) the workgroups of a stream land on: out[b] = XCC_ID | HW_ID << 8 (development probe for the CU-mask layout)
__global__ void mask_probe_kernel(unsigned* __restrict__ out, int spin)
{
    const unsigned xcc = __builtin_amdgcn_s_getreg((20) | (0 << 6) | ((4 - 1) << 11));        // HW_REG_XCC_ID[3:0]
    const unsigned hw = __builtin_amdgcn_s_getreg((4) | (0 << 6) | ((32 - 1) << 11));          // HW_REG_HW_ID
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    while (__builtin_amdgcn_s_memtime() - t0 < (unsigned long long) spin) { }
    if (threadIdx.x == 0) out[blockIdx.x] = (xcc & 0xf) | (hw << 8);
}

// mask words given explicitly (development probe); n workgroups, result per workgroup in out (host array)
int qrd_probe_cumask(const unsigned* mask_words, int nwords, int nwg, unsigned* out_host)
{
    hipStream_t st = nullptr;
    hipError_t e = hipExtStreamCreateWithCUMask(&st, (uint32_t) nwords, mask_words);
    if (e != hipSuccess) return (int) e;
    unsigned* d = nullptr;
    HIPCHK(hipMalloc(&d, sizeof(unsigned) * nwg));
    hipLaunchKernelGGL(mask_probe_kernel, dim3(nwg), dim3(64), 0, st, d, 20000);
    HIPCHK(hipStreamSynchronize(st));
    HIPCHK(hipMemcpy(out_host, d, sizeof(unsigned) * nwg, hipMemcpyDeviceToHost));
    hipFree(d);
    hipStreamDestroy(st);
    return 0;
}

int qrd_stream_destroy(void* s)
{
    if (!s) return 0;
    {
        std::lock_guard<std::mutex> lock(g_stream_mutex);
        for (int i = 0; i < g_nstream_cus; ++i)
            if (g_stream_cus[i].s == (hipStream_t) s) { g_stream_cus[i] = g_stream_cus[--g_nstream_cus]; break; }
    }
    return (int) hipStreamDestroy((hipStream_t) s);
}
// compute units behind a stream created here (CU-masked: its mask; else the device)
int qrd_stream_cus(void* s) { return stream_cus((hipStream_t) s); }
// ... of which a launch whose workgroups wait for each other may count on, one workgroup per compute unit: the dispatcher deals the workgroups
// of a launch evenly over the four shader engines of every XCD whatever the mask says, so what counts is the engine with the fewest unmasked
// CUs.  Masks made here are ranges [first, first + count) of bits, bit i = CU i/8 of XCC i%8, CU j of an XCC in engine j%4: that is
// count/32 per engine and XCD.  (MI355XQR_SPLIT=48: 25 co-resident workgroups of a one-launch panel on "48 compute units" never all started --
// the panel's hand-off timed out, status -105, profiles/r06_cu_split_by_shape.txt.)
int qrd_stream_cus_coresident(void* s)
{
    const int c = stream_cus((hipStream_t) s);
    return c >= 32 ? c / 32 * 32 : c / 2;
}
// hipGraph capture of a whole factorisation (thousands of dependent launches replayed by one call)
int qrd_capture_begin(void* s) { return (int) hipStreamBeginCapture((hipStream_t) s, hipStreamCaptureModeRelaxed); }
int qrd_capture_end(void* s, void** exec)
{
    hipGraph_t g = nullptr;
    hipError_t e = hipStreamEndCapture((hipStream_t) s, &g);
    if (e != hipSuccess) return (int) e;
    hipGraphExec_t x = nullptr;
    e = hipGraphInstantiate(&x, g, nullptr, nullptr, 0);
    hipGraphDestroy(g);
    *exec = (void*) x;
    return (int) e;
}
int qrd_graph_launch(void* exec, void* s) { return (int) hipGraphLaunch((hipGraphExec_t) exec, (hipStream_t) s); }
int qrd_graph_destroy(void* exec) { return exec ? (int) hipGraphExecDestroy((hipGraphExec_t) exec) : 0; }
int qrd_stream_sync(void* s) { return (int) hipStreamSynchronize((hipStream_t) s); }
int qrd_device_sync(void) { return (int) hipDeviceSynchronize(); }
int qrd_event_create(void** e) { hipEvent_t ev = nullptr; hipError_t r = hipEventCreate(&ev); *e = (r == hipSuccess) ? (void*) ev : nullptr; return (int) r; }
// an event that only brackets a launch for timing: no system-scope fence (cache write-back) when it is recorded -- the default event's
// fence sat inside every profiled interval (the bench's HIP-event figure of the update kernel read 3 % above rocprofv3's)
int qrd_event_create_timing(void** e)
{
    hipEvent_t ev = nullptr;
    hipError_t r = hipEventCreateWithFlags(&ev, hipEventDisableSystemFence);
    if (r != hipSuccess) r = hipEventCreate(&ev);
    *e = (r == hipSuccess) ? (void*) ev : nullptr;
    return (int) r;
}
int qrd_event_create_notiming(void** e)
{
    hipEvent_t ev = nullptr;
    hipError_t r = hipEventCreateWithFlags(&ev, hipEventDisableTiming);
    *e = (r == hipSuccess) ? (void*) ev : nullptr;
    return (int) r;
}
int qrd_event_destroy(void* e) { return (int) hipEventDestroy((hipEvent_t) e); }
int qrd_event_record(void* e, void* s) { return (int) hipEventRecord((hipEvent_t) e, (hipStream_t) s); }
int qrd_event_sync(void* e) { return (int) hipEventSynchronize((hipEvent_t) e); }
int qrd_stream_wait_event(void* s, void* e) { return (int) hipStreamWaitEvent((hipStream_t) s, (hipEvent_t) e, 0); }
int qrd_event_elapsed_ms(void* a, void* b, float* ms) { return (int) hipEventElapsedTime(ms, (hipEvent_t) a, (hipEvent_t) b); }
int qrd_device_count(int* n) { return (int) hipGetDeviceCount(n); }
int qrd_set_device(int d) { return (int) hipSetDevice(d); }
int qrd_get_device(int* d) { return (int) hipGetDevice(d); }
int qrd_host_word_alloc(unsigned** host, unsigned** dev)
{
    void* h = nullptr;
    void* d = nullptr;
    // coherent (fine-grained) on purpose: the device's store must become visible to the polling host thread while the stream still runs,
    // whatever HIP_HOST_COHERENT says
    hipError_t e = hipHostMalloc(&h, 64, hipHostMallocMapped | hipHostMallocCoherent);
    if (e == hipSuccess) e = hipHostGetDevicePointer(&d, h, 0);
    if (e != hipSuccess) { if (h) hipHostFree(h); return (int) e; }
    *reinterpret_cast<volatile unsigned*>(h) = 0u;
    *host = (unsigned*) h;
    *dev = (unsigned*) d;
    return 0;
}
int qrd_host_word_free(unsigned* host) { return host ? (int) hipHostFree(host) : 0; }
int qrd_host_register(void* p, size_t bytes) { return (int) hipHostRegister(p, bytes, hipHostRegisterDefault); }
int qrd_host_unregister(void* p) { return (int) hipHostUnregister(p); }
const char* qrd_error_string(int e) { return hipGetErrorString((hipError_t) e); }
int qrd_device_info(char* name, int name_len, int* cus, int* clock_khz, size_t* mem_bytes)
{
    int dev = 0;
    HIPCHK(hipGetDevice(&dev));
    if (!name && !clock_khz && !mem_bytes) {        // cheap path (plan creation only wants the CU count)
        if (cus) HIPCHK(hipDeviceGetAttribute(cus, hipDeviceAttributeMultiprocessorCount, dev));
        return 0;
    }
    hipDeviceProp_t p;
    HIPCHK(hipGetDeviceProperties(&p, dev));
    if (name && name_len > 0) { strncpy(name, p.gcnArchName, (size_t) name_len - 1); name[name_len - 1] = 0; }
    if (cus) *cus = p.multiProcessorCount;
    if (clock_khz) *clock_khz = p.clockRate;
    if (mem_bytes) *mem_bytes = p.totalGlobalMem;
    return 0;
}

// microbenchmarks (DESIGN.md "measured peaks").  out[0] = best TFLOP/s of back-to-back f64 MFMA over
// {1,2,4 workgroups per CU} x {4,8 accumulators}; out[1] = in-kernel shader clock (GHz) of that best run;
// out[2] = f64 VALU FMA TFLOP/s.
}   // extern "C" (templates need C++ linkage)
static int probe_one(int blocks, int iters, double* tflops, double* ghz)
{
    double* out; unsigned long long* st;
    HIPCHK(hipMalloc(&out, sizeof(double) * blocks * 256));
    HIPCHK(hipMalloc(&st, sizeof(unsigned long long) * 2 * blocks));
    hipEvent_t a, b; HIPCHK(hipEventCreate(&a)); HIPCHK(hipEventCreate(&b));
    hipLaunchKernelGGL(mfma_grid_kernel, dim3(blocks), dim3(256), 0, 0, out, st, iters / 8, 0.5);   // warm
    HIPCHK(hipEventRecord(a, 0));
    hipLaunchKernelGGL(mfma_grid_kernel, dim3(blocks), dim3(256), 0, 0, out, st, iters, 0.5);
    HIPCHK(hipEventRecord(b, 0)); HIPCHK(hipEventSynchronize(b));
    float ms = 0; HIPCHK(hipEventElapsedTime(&ms, a, b));
    *tflops = (double) blocks * 4 * iters * 16 * 2048.0 / (ms * 1e-3) / 1e12;
    unsigned long long h[2];
    HIPCHK(hipMemcpy(h, st, sizeof(h), hipMemcpyDeviceToHost));
    *ghz = h[1] ? (double) h[0] / (double) h[1] * 0.1 : 0.0;      // s_memtime ticks per 10 ns of s_memrealtime
    HIPCHK(hipEventDestroy(a)); HIPCHK(hipEventDestroy(b)); HIPCHK(hipFree(out)); HIPCHK(hipFree(st));
    return 0;
}
extern "C" {

int qrd_probe_mfma_f64(double* out3)
{
    hipDeviceProp_t p; int dev = 0;
    HIPCHK(hipGetDevice(&dev)); HIPCHK(hipGetDeviceProperties(&p, dev));
    const int cus = p.multiProcessorCount;
    double best = 0.0, best_ghz = 0.0;
    for (int bpc = 1; bpc <= 4; bpc *= 2) {
        double t, g;
        int rc = probe_one(cus * bpc, 4000 / bpc, &t, &g);
        if (rc) return rc;
        if (t > best) { best = t; best_ghz = g; }
    }
    out3[0] = best; out3[1] = best_ghz;
    {
        const int blocks = cus * 8, iters = 20000;
        double* o; HIPCHK(hipMalloc(&o, sizeof(double) * blocks * 256));
        hipEvent_t a, b; HIPCHK(hipEventCreate(&a)); HIPCHK(hipEventCreate(&b));
        hipLaunchKernelGGL(valu_peak_kernel, dim3(blocks), dim3(256), 0, 0, o, 1000, 0.5);
        HIPCHK(hipEventRecord(a, 0));
        hipLaunchKernelGGL(valu_peak_kernel, dim3(blocks), dim3(256), 0, 0, o, iters, 0.5);
        HIPCHK(hipEventRecord(b, 0)); HIPCHK(hipEventSynchronize(b));
        float ms = 0; HIPCHK(hipEventElapsedTime(&ms, a, b));
        out3[2] = (double) blocks * 256 * iters * 16 * 2.0 / (ms * 1e-3) / 1e12;
        hipEventDestroy(a); hipEventDestroy(b); hipFree(o);
    }
    return 0;
}

// one probe point: `blocks` workgroups of 4 waves; out2 = {TFLOP/s, s_memtime GHz}
int qrd_probe_mfma_f64_point(int blocks, int iters, double* out2)
{
    return probe_one(blocks, iters, &out2[0], &out2[1]);
}

int qrd_probe_copy(double* gbps)
{
    const size_t bytes = (size_t) 1 << 30;
    v2d *s, *d; HIPCHK(hipMalloc(&s, bytes)); HIPCHK(hipMalloc(&d, bytes));
    HIPCHK(hipMemset(s, 1, bytes));
    hipEvent_t a, b; HIPCHK(hipEventCreate(&a)); HIPCHK(hipEventCreate(&b));
    hipLaunchKernelGGL(stream_copy_kernel, dim3(2048), dim3(256), 0, 0, s, d, bytes / 16);
    HIPCHK(hipEventRecord(a, 0));
    for (int r = 0; r < 5; ++r) hipLaunchKernelGGL(stream_copy_kernel, dim3(2048), dim3(256), 0, 0, s, d, bytes / 16);
    HIPCHK(hipEventRecord(b, 0)); HIPCHK(hipEventSynchronize(b));
    float ms = 0; HIPCHK(hipEventElapsedTime(&ms, a, b));
    *gbps = 5.0 * 2.0 * bytes / (ms * 1e-3) / 1e9;
    hipEventDestroy(a); hipEventDestroy(b); hipFree(s); hipFree(d);
    return 0;
}

}   // extern "C"
