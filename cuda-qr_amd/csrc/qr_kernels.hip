// qr_kernels.hip -- hand-written gfx950 (CDNA4 / MI355X) kernels for fp64 blocked Householder QR,
// plus the extern "C" launchers the C host layer (qr_host.c) calls.  No vendor BLAS/solver calls.
//
// What replaces what in the reference (brian-kelley/CUDA-QR):
//   leaf_step_kernel      <- panelHouseholderKernel (qr.cu:60-333) / qr.c:109-235: Householder vector,
//                            ||x||_2, reflector apply inside the panel.  Here: rows live one-per-thread
//                            in registers, all loads coalesced down columns, norms + all v^T a dot
//                            products come from ONE fused wave-shuffle reduction per column.
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
#include "qr_device.h"

typedef double v4d __attribute__((ext_vector_type(4)));
typedef double v2d __attribute__((ext_vector_type(2)));

#define HIPCHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) return (int) e_; } while (0)

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

template <int TR>   // TR = tile extent / 32 (rows of the row-fast image)
__device__ __forceinline__ void load_rowfast(v2d (&reg)[TR], const double* __restrict__ A, int lda,
                                             int i0, int k0, int M, int kend, bool vec, int tid)
{
    constexpr int HALF = 16 * TR;          // double2 per column
#pragma unroll
    for (int q = 0; q < TR; ++q) {
        const int idx = tid + 256 * q;
        const int col = idx / HALF, r2 = idx % HALF;
        const int i = i0 + 2 * r2, k = k0 + col;
        v2d v = {0.0, 0.0};
        if (vec && i + 1 < M && k < kend) {
            v = *reinterpret_cast<const v2d*>(A + (size_t) k * lda + i);
        } else if (k < kend) {
            if (i < M) v.x = A[(size_t) k * lda + i];
            if (i + 1 < M) v.y = A[(size_t) k * lda + i + 1];
        }
        reg[q] = v;
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
                                           int j0, int k0, int N, int kend, bool vec, int tid)
{
#pragma unroll
    for (int q = 0; q < TC; ++q) {
        const int idx = tid + 256 * q;
        const int j = j0 + idx / 8, k = k0 + 2 * (idx % 8);
        v2d v = {0.0, 0.0};
        if (j < N) {
            if (vec && k + 1 < kend) {
                v = *reinterpret_cast<const v2d*>(B + (size_t) j * ldb + k);
            } else {
                if (k < kend) v.x = B[(size_t) j * ldb + k];
                if (k + 1 < kend) v.y = B[(size_t) j * ldb + k + 1];
            }
        }
        reg[q] = v;
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

// C = beta*C + alpha*A*B     A: M x K (lda), B: K x N (ldb), C: M x N (ldc), all column-major.
// Used for: trailing update A2 -= V*W (K = nb), VT = V*T, Q*R products, Q_local*Q_tree.
template <int TI, int TJ>
__global__ __launch_bounds__(256, 2) void gemm_nn_kernel(int M, int N, int K, double alpha,
                                                         const double* __restrict__ A, int lda,
                                                         const double* __restrict__ B, int ldb,
                                                         double beta, double* __restrict__ C, int ldc,
                                                         int vecA, int vecB)
{
    constexpr int BM = 32 * TI, BN = 32 * TJ, LA = BM + 16;
    extern __shared__ __attribute__((aligned(16))) double smem[];
    double* As = smem;                      // [2][BK][LA]
    double* Bs = smem + 2 * BK * LA;        // [2][BN][LDKF]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wi = wave & 1, wj = wave >> 1;
    const int i0 = blockIdx.x * BM, j0 = blockIdx.y * BN;
    const int l15 = lane & 15, l4 = lane >> 4;

    v4d acc[TJ][TI];
#pragma unroll
    for (int a = 0; a < TJ; ++a)
#pragma unroll
        for (int b = 0; b < TI; ++b) acc[a][b] = (v4d){0.0, 0.0, 0.0, 0.0};

    v2d ra[TI], rb[TJ];
    const int nk = (K + BK - 1) / BK;
    load_rowfast<TI>(ra, A, lda, i0, 0, M, K, vecA != 0, tid);
    load_kfast<TJ>(rb, B, ldb, j0, 0, N, K, vecB != 0, tid);
    store_rowfast<TI>(ra, As, tid);
    store_kfast<TJ>(rb, Bs, tid);
    __syncthreads();
    for (int kt = 0; kt < nk; ++kt) {
        const int buf = kt & 1;
        if (kt + 1 < nk) {
            load_rowfast<TI>(ra, A, lda, i0, (kt + 1) * BK, M, K, vecA != 0, tid);
            load_kfast<TJ>(rb, B, ldb, j0, (kt + 1) * BK, N, K, vecB != 0, tid);
        }
        const double* as = As + buf * BK * LA;
        const double* bs = Bs + buf * BN * LDKF;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            const int kk = 4 * ks + l4;
            double rowv[TI], colv[TJ];
#pragma unroll
            for (int b = 0; b < TI; ++b) rowv[b] = as[kk * LA + wi * 16 * TI + 16 * b + l15];
#pragma unroll
            for (int a = 0; a < TJ; ++a) colv[a] = bs[(wj * 16 * TJ + 16 * a + l15) * LDKF + kk];
#pragma unroll
            for (int a = 0; a < TJ; ++a)
#pragma unroll
                for (int b = 0; b < TI; ++b)
                    acc[a][b] = __builtin_amdgcn_mfma_f64_16x16x4f64(colv[a], rowv[b], acc[a][b], 0, 0, 0);
        }
        if (kt + 1 < nk) {
            store_rowfast<TI>(ra, As + (buf ^ 1) * BK * LA, tid);
            store_kfast<TJ>(rb, Bs + (buf ^ 1) * BN * LDKF, tid);
        }
        __syncthreads();
    }
#pragma unroll
    for (int a = 0; a < TJ; ++a)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int j = j0 + wj * 16 * TJ + 16 * a + l4 + 4 * r;
            if (j < N) {
#pragma unroll
                for (int b = 0; b < TI; ++b) {
                    const int i = i0 + wi * 16 * TI + 16 * b + l15;
                    if (i < M) {
                        double* cp = C + (size_t) j * ldc + i;
                        double v = alpha * acc[a][b][r];
                        if (beta != 0.0) v += beta * (*cp);
                        *cp = v;
                    }
                }
            }
        }
}

// C = alpha * A^T * B (+ beta*C when not split)   A: K x M (lda), B: K x N (ldb), C: M x N (ldc).
// K is the long dimension (panel height): gridDim.z K-slices each write their own slab
// (slab z at C + z*slab_stride, ld = ldc) and slab_reduce_kernel sums them in a fixed order
// (deterministic; no float atomics).  Used for W = (V T)^T A2, Gram = V^T V, Q^T Q.
template <int TI, int TJ>
__global__ __launch_bounds__(256, 2) void gemm_tn_kernel(int M, int N, int K, int kchunk, double alpha,
                                                         const double* __restrict__ A, int lda,
                                                         const double* __restrict__ B, int ldb,
                                                         double beta, double* __restrict__ C, int ldc,
                                                         size_t slab_stride, int vecA, int vecB)
{
    constexpr int BM = 32 * TI, BN = 32 * TJ;
    extern __shared__ __attribute__((aligned(16))) double smem[];
    double* As = smem;                        // [2][BM][LDKF]
    double* Bs = smem + 2 * BM * LDKF;        // [2][BN][LDKF]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wi = wave & 1, wj = wave >> 1;
    const int i0 = blockIdx.x * BM, j0 = blockIdx.y * BN;
    const int l15 = lane & 15, l4 = lane >> 4;
    const int kbeg = blockIdx.z * kchunk;
    const int kend = min(K, kbeg + kchunk);
    C += (size_t) blockIdx.z * slab_stride;

    v4d acc[TJ][TI];
#pragma unroll
    for (int a = 0; a < TJ; ++a)
#pragma unroll
        for (int b = 0; b < TI; ++b) acc[a][b] = (v4d){0.0, 0.0, 0.0, 0.0};

    v2d ra[TI], rb[TJ];
    const int nk = (kend - kbeg + BK - 1) / BK;
    if (nk > 0) {
        load_kfast<TI>(ra, A, lda, i0, kbeg, M, kend, vecA != 0, tid);
        load_kfast<TJ>(rb, B, ldb, j0, kbeg, N, kend, vecB != 0, tid);
        store_kfast<TI>(ra, As, tid);
        store_kfast<TJ>(rb, Bs, tid);
    }
    __syncthreads();
    for (int kt = 0; kt < nk; ++kt) {
        const int buf = kt & 1;
        if (kt + 1 < nk) {
            load_kfast<TI>(ra, A, lda, i0, kbeg + (kt + 1) * BK, M, kend, vecA != 0, tid);
            load_kfast<TJ>(rb, B, ldb, j0, kbeg + (kt + 1) * BK, N, kend, vecB != 0, tid);
        }
        const double* as = As + buf * BM * LDKF;
        const double* bs = Bs + buf * BN * LDKF;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            const int kk = 4 * ks + l4;
            double rowv[TI], colv[TJ];
#pragma unroll
            for (int b = 0; b < TI; ++b) rowv[b] = as[(wi * 16 * TI + 16 * b + l15) * LDKF + kk];
#pragma unroll
            for (int a = 0; a < TJ; ++a) colv[a] = bs[(wj * 16 * TJ + 16 * a + l15) * LDKF + kk];
#pragma unroll
            for (int a = 0; a < TJ; ++a)
#pragma unroll
                for (int b = 0; b < TI; ++b)
                    acc[a][b] = __builtin_amdgcn_mfma_f64_16x16x4f64(colv[a], rowv[b], acc[a][b], 0, 0, 0);
        }
        if (kt + 1 < nk) {
            store_kfast<TI>(ra, As + (buf ^ 1) * BM * LDKF, tid);
            store_kfast<TJ>(rb, Bs + (buf ^ 1) * BN * LDKF, tid);
        }
        __syncthreads();
    }
#pragma unroll
    for (int a = 0; a < TJ; ++a)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int j = j0 + wj * 16 * TJ + 16 * a + l4 + 4 * r;
            if (j < N) {
#pragma unroll
                for (int b = 0; b < TI; ++b) {
                    const int i = i0 + wi * 16 * TI + 16 * b + l15;
                    if (i < M) {
                        double* cp = C + (size_t) j * ldc + i;
                        double v = alpha * acc[a][b][r];
                        if (beta != 0.0) v += beta * (*cp);
                        *cp = v;
                    }
                }
            }
        }
}

// out(:,j) = beta*out(:,j) + Tm^T * sum_z slab_z(:,j)   (Tm optional, upper triangular M x M, M <= 256)
// One block per output column; blockDim.x >= M.
__global__ void slab_reduce_kernel(int M, int N, int nslab, const double* __restrict__ slabs, int lds,
                                   size_t slab_stride, const double* __restrict__ Tm, int ldt,
                                   double beta, double* __restrict__ out, int ldo)
{
    __shared__ double col[256];
    const int j = blockIdx.x, i = threadIdx.x;
    if (j >= N) return;
    double s = 0.0;
    if (i < M) {
        const double* p = slabs + (size_t) j * lds + i;
        for (int z = 0; z < nslab; ++z) s += p[(size_t) z * slab_stride];
    }
    if (Tm) {
        col[i] = s;
        __syncthreads();
        if (i < M) {
            double t = 0.0;
            for (int p = 0; p <= i; ++p) t += Tm[(size_t) i * ldt + p] * col[p];   // (T^T)(i,p) = T(p,i)
            s = t;
        }
    }
    if (i < M) {
        double* o = out + (size_t) j * ldo + i;
        *o = (beta != 0.0) ? beta * (*o) + s : s;
    }
}

// ------------------------------------------------------------------------------------------------
// Leaf panel factorisation: one launch per column (the kernel boundary is the grid-wide
// dependency of Householder QR: ~1.5 us, cheaper than an in-kernel grid barrier on 8 XCDs).
// Launch j (j = -1 .. w-1):
//   1. every block sums the per-block partial dot products left by launch j-1
//        d[c] = sum_{i>j} P(i,j) P(i,c)   (c = 0..w-1: c==j -> ||x||^2 tail, c>j -> x^T a_c, c<j -> v_c^T x)
//      in a fixed order (bitwise identical in all blocks, deterministic run to run);
//   2. forms beta, tau, 1/u (LAPACK dlarfg convention; tau = 0 when the tail is exactly zero -- the
//      reference divides by norm = 0 there and produces NaN, qr.c:152) and s_c = v^T a_c;
//   3. block 0 appends column j of the leaf's T (T(0:j,j) = -tau T(0:j,0:j) V^T v);
//   4. every thread scales its row of v, applies the reflector to its row of the remaining
//      columns (registers only), stores, and accumulates the dot products of column j+1;
//   5. wave-shuffle + LDS reduction of the w partial sums -> part_out[block][c].
// Rows are one-per-thread: every global access is 64 consecutive doubles of one column (coalesced).
// ------------------------------------------------------------------------------------------------
#define LEAFW 32

__global__ __launch_bounds__(256) void leaf_step_kernel(double* __restrict__ P, int ld, int mk, int w,
                                                        int j, int rpt,
                                                        const double* __restrict__ part_in, int nblk_in,
                                                        const double* __restrict__ row_in,
                                                        double* __restrict__ part_out,
                                                        double* __restrict__ row_out,
                                                        double* __restrict__ tau, double* __restrict__ T,
                                                        int ldt, double* __restrict__ Vw, int ldv)
{
    __shared__ double s_red[8][LEAFW];
    __shared__ double s_d[LEAFW];
    __shared__ double s_s[LEAFW];
    __shared__ double s_scal[3];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;

    double tj = 0.0, beta = 0.0, inv_u = 0.0;
    if (j >= 0) {
        {
            const int c = tid & 31, part = tid >> 5;
            double sum = 0.0;
            for (int g = part; g < nblk_in; g += 8) sum += part_in[g * LEAFW + c];
            s_red[part][c] = sum;
        }
        __syncthreads();
        if (tid < LEAFW) {
            double d = 0.0;
#pragma unroll
            for (int p = 0; p < 8; ++p) d += s_red[p][tid];
            s_d[tid] = d;
        }
        __syncthreads();
        if (tid < LEAFW) {
            const double alpha = row_in[j];
            const double sigma = s_d[j];
            double t, b, iu;
            if (sigma == 0.0) { t = 0.0; b = alpha; iu = 0.0; }
            else {
                const double nrm = sqrt(alpha * alpha + sigma);
                b = -copysign(nrm, alpha);
                t = (b - alpha) / b;
                iu = 1.0 / (alpha - b);
            }
            s_s[tid] = (tid < w) ? row_in[tid] + s_d[tid] * iu : 0.0;
            if (tid == 0) { s_scal[0] = t; s_scal[1] = b; s_scal[2] = iu; }
        }
        __syncthreads();
        tj = s_scal[0]; beta = s_scal[1]; inv_u = s_scal[2];
        if (blockIdx.x == 0 && tid < w) {
            if (tid < j) {
                double t = 0.0;
                for (int q = tid; q < j; ++q) t += T[(size_t) q * ldt + tid] * s_s[q];
                T[(size_t) j * ldt + tid] = -tj * t;
            } else if (tid == j) {
                T[(size_t) j * ldt + j] = tj;
                tau[j] = tj;
            } else {
                T[(size_t) j * ldt + tid] = 0.0;
            }
        }
    }

    const int jn = j + 1;
    double acc[LEAFW];
#pragma unroll
    for (int c = 0; c < LEAFW; ++c) acc[c] = 0.0;

    for (int r = 0; r < rpt; ++r) {
        const int i = (blockIdx.x * rpt + r) * 256 + tid;
        if (i >= mk) continue;
        double x[LEAFW];
#pragma unroll
        for (int c = 0; c < LEAFW; ++c) x[c] = (c < w) ? P[(size_t) c * ld + i] : 0.0;
        if (j >= 0) {
            if (i > j) {
                double xj = 0.0;
#pragma unroll
                for (int c = 0; c < LEAFW; ++c) if (c == j) xj = x[c];
                const double vi = xj * inv_u;
                const double tv = tj * vi;
#pragma unroll
                for (int c = 0; c < LEAFW; ++c) {
                    if (c == j) { x[c] = vi; P[(size_t) c * ld + i] = vi; Vw[(size_t) c * ldv + i] = vi; }
                    else if (c > j && c < w) { x[c] -= tv * s_s[c]; P[(size_t) c * ld + i] = x[c]; }
                }
            } else if (i == j) {
#pragma unroll
                for (int c = 0; c < LEAFW; ++c) {
                    if (c == j) { x[c] = beta; P[(size_t) c * ld + i] = beta; Vw[(size_t) c * ldv + i] = 1.0; }
                    else if (c > j && c < w) { x[c] -= tj * s_s[c]; P[(size_t) c * ld + i] = x[c]; }
                }
            } else {
                Vw[(size_t) j * ldv + i] = 0.0;
            }
        }
        if (jn < w) {
            if (i == jn) {
#pragma unroll
                for (int c = 0; c < LEAFW; ++c) if (c < w) row_out[c] = x[c];
            } else if (i > jn) {
                double xn = 0.0;
#pragma unroll
                for (int c = 0; c < LEAFW; ++c) if (c == jn) xn = x[c];
#pragma unroll
                for (int c = 0; c < LEAFW; ++c) acc[c] += xn * x[c];
            }
        }
    }
    if (jn < w) {
#pragma unroll
        for (int c = 0; c < LEAFW; ++c) {
            double v = acc[c];
#pragma unroll
            for (int off = 32; off >= 1; off >>= 1) v += __shfl_xor(v, off);
            if (lane == 0) s_red[wave][c] = v;
        }
        __syncthreads();
        if (tid < LEAFW)
            part_out[blockIdx.x * LEAFW + tid] = s_red[0][tid] + s_red[1][tid] + s_red[2][tid] + s_red[3][tid];
    }
}

// ------------------------------------------------------------------------------------------------
// Compact-WY T for an outer panel of nbp columns made of leaves of width ib, from the Gram matrix
// G = V^T V (one gemm_tn) -- the compact-WY counterpart of the reference's W accumulation
// (qr.c:170-213).  I - V T V^T = H_0 H_1 ... H_{nbp-1}.
//   build_diag != 0 : diagonal blocks are (re)built from G and tau by the column recurrence
//                     T(0:j,j) = -tau_j T(0:j,0:j) G(0:j,j); row p of a block depends only on row p,
//                     so one thread per row needs no synchronisation at all.
//   then block column b is merged: T(0:cb, cb:cb+wb) = -T(0:cb,0:cb) * (G(0:cb, cb:cb+wb) * T_bb).
// Also writes Tt = T^T (used when applying Q instead of Q^T).  Single block of 256 threads.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void larft_kernel(int nbp, int ib, const double* __restrict__ G, int ldg,
                                                    const double* __restrict__ tau, double* __restrict__ T,
                                                    int ldt, double* __restrict__ Tt, int build_diag)
{
    extern __shared__ __attribute__((aligned(16))) double X[];   // [cb][LEAFW]
    const int tid = threadIdx.x;
    if (build_diag) {
        for (int p = tid; p < nbp; p += 256) {
            const int cb = (p / ib) * ib, wb = min(ib, nbp - cb), pl = p - cb;
            double trow[LEAFW];
#pragma unroll
            for (int q = 0; q < LEAFW; ++q) trow[q] = 0.0;
#pragma unroll
            for (int jj = 0; jj < LEAFW; ++jj) {
                if (jj < wb) {
                    const double tj = tau[cb + jj];
                    if (pl == jj) trow[jj] = tj;
                    else if (pl < jj) {
                        double s = 0.0;
#pragma unroll
                        for (int q = 0; q < LEAFW; ++q)
                            if (q >= pl && q < jj) s += trow[q] * G[(size_t) (cb + jj) * ldg + cb + q];
                        trow[jj] = -tj * s;
                    }
                }
            }
#pragma unroll
            for (int q = 0; q < LEAFW; ++q)
                if (q < wb) T[(size_t) (cb + q) * ldt + p] = trow[q];
        }
        __syncthreads();
    }
    // zero the strictly-lower block part (leaf kernels only write their own diagonal blocks)
    for (int e = tid; e < nbp * nbp; e += 256) {
        const int p = e % nbp, c = e / nbp;
        if (p / ib > c / ib) T[(size_t) c * ldt + p] = 0.0;
    }
    __syncthreads();
    for (int cb = ib; cb < nbp; cb += ib) {
        const int wb = min(ib, nbp - cb);
        // X(q,c) = sum_{r<=c} G(q, cb+r) * T_bb(r, c)
        for (int e = tid; e < cb * wb; e += 256) {
            const int q = e % cb, c = e / cb;
            double s = 0.0;
            for (int r = 0; r <= c; ++r) s += G[(size_t) (cb + r) * ldg + q] * T[(size_t) (cb + c) * ldt + cb + r];
            X[q * LEAFW + c] = s;
        }
        __syncthreads();
        // T(p, cb+c) = - sum_{q>=p} T(p,q) X(q,c)
        for (int p = tid; p < cb; p += 256) {
            double a[LEAFW];
#pragma unroll
            for (int c = 0; c < LEAFW; ++c) a[c] = 0.0;
            for (int q = p; q < cb; ++q) {
                const double t = T[(size_t) q * ldt + p];
                const double* xr = X + q * LEAFW;
#pragma unroll
                for (int c = 0; c < LEAFW; ++c) a[c] += t * xr[c];
            }
#pragma unroll
            for (int c = 0; c < LEAFW; ++c)
                if (c < wb) T[(size_t) (cb + c) * ldt + p] = -a[c];
        }
        __syncthreads();
    }
    if (Tt)
        for (int e = tid; e < nbp * nbp; e += 256) {
            const int p = e % nbp, c = e / nbp;
            Tt[(size_t) p * ldt + c] = T[(size_t) c * ldt + p];
        }
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

// MFMA fp64 issue-rate probe: each wave runs `iters` x 8 independent accumulators back to back.
__global__ __launch_bounds__(256) void mfma_peak_kernel(double* out, int iters, double seed)
{
    v4d acc[8];
#pragma unroll
    for (int a = 0; a < 8; ++a) acc[a] = (v4d){0.0, 0.0, 0.0, 0.0};
    double x = seed + threadIdx.x * 1e-3, y = 1.0 - threadIdx.x * 1e-4;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int a = 0; a < 8; ++a) acc[a] = __builtin_amdgcn_mfma_f64_16x16x4f64(x, y, acc[a], 0, 0, 0);
    }
    double s = 0.0;
#pragma unroll
    for (int a = 0; a < 8; ++a) s += acc[a][0] + acc[a][1] + acc[a][2] + acc[a][3];
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

template <int TI, int TJ>
static int launch_nn(hipStream_t s, int M, int N, int K, double alpha, const double* A, int lda,
                     const double* B, int ldb, double beta, double* C, int ldc)
{
    constexpr int BM = 32 * TI, BN = 32 * TJ;
    const size_t shm = sizeof(double) * (2 * BK * (BM + 16) + 2 * BN * LDKF);
    dim3 grid((M + BM - 1) / BM, (N + BN - 1) / BN);
    hipLaunchKernelGGL((gemm_nn_kernel<TI, TJ>), grid, dim3(256), shm, s, M, N, K, alpha, A, lda, B, ldb, beta,
                       C, ldc, vec_ok(A, lda), vec_ok(B, ldb));
    return (int) hipGetLastError();
}

template <int TI, int TJ>
static int launch_tn(hipStream_t s, int M, int N, int K, int ksplit, int kchunk, double alpha, const double* A,
                     int lda, const double* B, int ldb, double beta, double* C, int ldc, size_t slab_stride)
{
    constexpr int BM = 32 * TI, BN = 32 * TJ;
    const size_t shm = sizeof(double) * (2 * BM * LDKF + 2 * BN * LDKF);
    dim3 grid((M + BM - 1) / BM, (N + BN - 1) / BN, ksplit);
    hipLaunchKernelGGL((gemm_tn_kernel<TI, TJ>), grid, dim3(256), shm, s, M, N, K, kchunk, alpha, A, lda, B, ldb,
                       beta, C, ldc, slab_stride, vec_ok(A, lda), vec_ok(B, ldb));
    return (int) hipGetLastError();
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
    rc |= allow_lds(gemm_nn_kernel<4, 4>, sizeof(double) * (2 * BK * (128 + 16) + 2 * 128 * LDKF));
    rc |= allow_lds(gemm_tn_kernel<4, 4>, sizeof(double) * (4 * 128 * LDKF));
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
    // tile choice: big square tiles when the grid still fills the chip, smaller otherwise
    const long long t44 = (long long) ((M + 127) / 128) * ((N + 127) / 128);
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
int qrd_gemm_tn(void* stream, int M, int N, int K, double alpha, const double* A, int lda, const double* B,
                int ldb, double beta, double* C, int ldc, double* slabs, size_t slab_cap, const double* Tm,
                int ldt)
{
    hipStream_t s = (hipStream_t) stream;
    if (M <= 0 || N <= 0) return 0;
    if (Tm && M > 256) return -2;
    int ti, tj;
    if (M <= 32) { ti = 1; tj = (N > 64) ? 4 : 1; }
    else if (M <= 64 || N <= 64) { ti = 2; tj = 2; }
    else { ti = 4; tj = 4; }
    const int BM = 32 * ti, BN = 32 * tj;
    const long long tiles = (long long) ((M + BM - 1) / BM) * ((N + BN - 1) / BN);
    // aim at ~2 blocks per CU; each K slice at least 8 k-tiles long
    long long want = (512 + tiles - 1) / tiles;
    long long maxk = (K + 8 * BK - 1) / (8 * BK);
    if (want > maxk) want = maxk;
    if (want < 1) want = 1;
    const size_t per = (size_t) M * N;
    if (slabs == nullptr || slab_cap < per) want = 1;
    else if ((size_t) want * per > slab_cap) want = (long long) (slab_cap / per);
    if (want > 1024) want = 1024;
    int ksplit = (int) want;
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
    else rc = launch_tn<4, 4>(s, M, N, K, ksplit, kchunk, alpha, A, lda, B, ldb, b2, dst, ldd, per);
    if (rc) return rc;
    if (!direct) {
        const int threads = (M + 63) / 64 * 64;
        hipLaunchKernelGGL(slab_reduce_kernel, dim3(N), dim3(threads), 0, s, M, N, ksplit, slabs, M, per, Tm, ldt,
                           beta, C, ldc);
        rc = (int) hipGetLastError();
    }
    return rc;
}

// One leaf: factor the mk x w panel at P (ld) in place; tau[0..w), T (w x w at T, ldt), explicit V into Vw.
// scratch: 2*(256*LEAFW + LEAFW) doubles.
int qrd_leaf_panel(void* stream, double* P, int ld, int mk, int w, double* tau, double* T, int ldt, double* Vw,
                   int ldv, double* scratch)
{
    hipStream_t s = (hipStream_t) stream;
    if (w < 1 || w > LEAFW || mk < w) return -4;
    int rpt = (mk + 256 * 256 - 1) / (256 * 256);
    if (rpt < 1) rpt = 1;
    const int nblk = (mk + 256 * rpt - 1) / (256 * rpt);
    double* part[2] = {scratch, scratch + 256 * LEAFW + LEAFW};
    double* rowb[2] = {part[0] + 256 * LEAFW, part[1] + 256 * LEAFW};
    for (int j = -1; j < w; ++j) {
        const int in = (j + 2) & 1, out = (j + 1) & 1;
        hipLaunchKernelGGL(leaf_step_kernel, dim3(nblk), dim3(256), 0, s, P, ld, mk, w, j, rpt, part[in], nblk,
                           rowb[in], part[out], rowb[out], tau, T, ldt, Vw, ldv);
    }
    return (int) hipGetLastError();
}

int qrd_larft(void* stream, int nbp, int ib, const double* G, int ldg, const double* tau, double* T, int ldt,
              double* Tt, int build_diag)
{
    if (ib > LEAFW || nbp < 1) return -5;
    const size_t shm = sizeof(double) * (size_t) (nbp > ib ? nbp : ib) * LEAFW;
    hipLaunchKernelGGL(larft_kernel, dim3(1), dim3(256), shm, (hipStream_t) stream, nbp, ib, G, ldg, tau, T, ldt,
                       Tt, build_diag);
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
    hipStream_t st;
    int lo = 0, hi = 0;
    hipError_t e;
    if (high_priority && hipDeviceGetStreamPriorityRange(&lo, &hi) == hipSuccess)
        e = hipStreamCreateWithPriority(&st, hipStreamNonBlocking, hi);
    else
        e = hipStreamCreateWithFlags(&st, hipStreamNonBlocking);
    *s = (void*) st;
    return (int) e;
}
int qrd_stream_destroy(void* s) { return (int) hipStreamDestroy((hipStream_t) s); }
int qrd_stream_sync(void* s) { return (int) hipStreamSynchronize((hipStream_t) s); }
int qrd_device_sync(void) { return (int) hipDeviceSynchronize(); }
int qrd_event_create(void** e) { hipEvent_t ev; hipError_t r = hipEventCreate(&ev); *e = (void*) ev; return (int) r; }
int qrd_event_create_notiming(void** e) { hipEvent_t ev; hipError_t r = hipEventCreateWithFlags(&ev, hipEventDisableTiming); *e = (void*) ev; return (int) r; }
int qrd_event_destroy(void* e) { return (int) hipEventDestroy((hipEvent_t) e); }
int qrd_event_record(void* e, void* s) { return (int) hipEventRecord((hipEvent_t) e, (hipStream_t) s); }
int qrd_event_sync(void* e) { return (int) hipEventSynchronize((hipEvent_t) e); }
int qrd_stream_wait_event(void* s, void* e) { return (int) hipStreamWaitEvent((hipStream_t) s, (hipEvent_t) e, 0); }
int qrd_event_elapsed_ms(void* a, void* b, float* ms) { return (int) hipEventElapsedTime(ms, (hipEvent_t) a, (hipEvent_t) b); }
int qrd_device_count(int* n) { return (int) hipGetDeviceCount(n); }
int qrd_set_device(int d) { return (int) hipSetDevice(d); }
const char* qrd_error_string(int e) { return hipGetErrorString((hipError_t) e); }
int qrd_device_info(char* name, int name_len, int* cus, int* clock_khz, size_t* mem_bytes)
{
    hipDeviceProp_t p;
    int dev = 0;
    HIPCHK(hipGetDevice(&dev));
    HIPCHK(hipGetDeviceProperties(&p, dev));
    if (name && name_len > 0) { strncpy(name, p.gcnArchName, (size_t) name_len - 1); name[name_len - 1] = 0; }
    if (cus) *cus = p.multiProcessorCount;
    if (clock_khz) *clock_khz = p.clockRate;
    if (mem_bytes) *mem_bytes = p.totalGlobalMem;
    return 0;
}

// microbenchmarks (DESIGN.md "measured peaks"): returns TFLOP/s of back-to-back f64 MFMA and GB/s of a 16-B copy
int qrd_probe_mfma_f64(double* tflops)
{
    int cus = 256;
    hipDeviceProp_t p; int dev = 0;
    HIPCHK(hipGetDevice(&dev)); HIPCHK(hipGetDeviceProperties(&p, dev)); cus = p.multiProcessorCount;
    const int blocks = cus * 2, iters = 4000;
    double* out; HIPCHK(hipMalloc(&out, sizeof(double) * blocks * 256));
    hipEvent_t a, b; HIPCHK(hipEventCreate(&a)); HIPCHK(hipEventCreate(&b));
    hipLaunchKernelGGL(mfma_peak_kernel, dim3(blocks), dim3(256), 0, 0, out, 100, 0.5);
    HIPCHK(hipEventRecord(a, 0));
    hipLaunchKernelGGL(mfma_peak_kernel, dim3(blocks), dim3(256), 0, 0, out, iters, 0.5);
    HIPCHK(hipEventRecord(b, 0)); HIPCHK(hipEventSynchronize(b));
    float ms = 0; HIPCHK(hipEventElapsedTime(&ms, a, b));
    const double fl = (double) blocks * 4 * iters * 8 * 2048.0;
    *tflops = fl / (ms * 1e-3) / 1e12;
    hipEventDestroy(a); hipEventDestroy(b); hipFree(out);
    return 0;
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
