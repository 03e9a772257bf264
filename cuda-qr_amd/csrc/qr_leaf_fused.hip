// qr_leaf_fused.hip -- fused kernels of the leaf chain (in-panel block update of the blocked Householder QR, qr.c:215-235).
//
// Every launch between two leaves of an outer panel sits on the factorisation's critical chain and costs ~5 us before it does
// any work, so the chain is shortened by giving one launch the work of two:
//
//   leaf_update_gram_kernel:   A_rest -= V_l W          (the in-panel update with the leaf's 32 reflectors, K = 32)
//                              G      = A_next^T A_next (Gram matrix of the NEXT leaf's 32 columns, rows below the current
//                                                        leaf's diagonal block) -- what gram32_kernel did in a launch of its own
//
// Layout trick shared by these kernels (v_mfma_f64_16x16x4_f64: A operand lane l = A[p = l&15][k = l>>4], B operand lane l =
// B[k = l>>4][q = l&15], D reg r of lane l = D[p = (l>>4) + 4r][q = l&15]).  Matrix ROWS go on p, a wave owns 64 consecutive
// rows as four INTERLEAVED 16-row tiles: row index p of tile t is physical row base + 4 p + t.  Then
//   * an A-operand fragment of tile t (row p = l15, reflector k = 4 ks + l4) is one of four consecutive doubles of column k:
//     one 32-byte access per lane brings the fragments of all four tiles (a wave instruction = 4 columns x 512 B);
//   * D reg r of tile t in lane (l15, l4) is row base + 4 l4 + 16 r + t of column l15: for a fixed r the four tiles are again four
//     consecutive doubles, so C is read and written 32 B per lane (16 columns x 128 B per wave instruction);
//   * and D reg r IS an operand fragment for a product that contracts over rows (k slot l4 of step r = row 4 l4 + 16 r + t):
//     the updated columns go straight from the accumulators into the Gram product, no LDS transpose, no barrier.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>
#include "qr_device.h"
#include "qr_common.h"

#define LF_PT 512
#define LF_PW 32
#define LF_MAXSLAB 1024

__device__ __forceinline__ void lf_load4(const double* __restrict__ p, double (&d)[4])
{
    const v2d a = *reinterpret_cast<const v2d*>(p), b = *reinterpret_cast<const v2d*>(p + 2);
    d[0] = a[0]; d[1] = a[1]; d[2] = b[0]; d[3] = b[1];
}

__device__ __forceinline__ void lf_store4(double* __restrict__ p, double a, double b, double c, double d)
{
    *reinterpret_cast<v2d*>(p) = (v2d){a, b};
    *reinterpret_cast<v2d*>(p + 2) = (v2d){c, d};
}

// C (mk x 32 npairs, ldc) -= V (mk x 32, ldv) W (32 x 32 npairs, ld 32); gslabs != nullptr: slab blockIdx.x (32 x 32, ld 32) =
// C(r, 0:32)^T C(r, 0:32) over this workgroup's rows r >= 32 of the UPDATED columns.  mk % 4 == 0; V, C 16-byte aligned with even
// leading dimensions.  Workgroup (x, y): row blocks of 512 rows x, x + gridDim.x, ..., column pairs y, y + gridDim.y, ...  (a pair = 32 columns).
__global__ __launch_bounds__(LF_PT) void leaf_update_gram_kernel(int mk, int npairs, const double* __restrict__ V, int ldv,
                                                                 const double* __restrict__ W, double* __restrict__ C, int ldc,
                                                                 double* __restrict__ gslabs)
{
    extern __shared__ __attribute__((aligned(16))) double lf_red[];          // [8 waves][32 * 32] when gslabs
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l15 = lane & 15, l4 = lane >> 4;
    const bool gram = gslabs != nullptr && blockIdx.y == 0;
    v4d g[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) g[i][j] = (v4d){0.0, 0.0, 0.0, 0.0};
    // row blocks blockIdx.x, blockIdx.x + gridDim.x, ... (very tall leaves: at most LF_MAXSLAB workgroups, i.e. partial Grams)
    for (int base = blockIdx.x * LF_PT + wave * 64; base < mk; base += gridDim.x * LF_PT) {
        double xa[8][4];                                                     // V(base + 4 l15 + t, 4 ks + l4)
        {
            const int rowA = base + 4 * l15;
            const bool va = rowA < mk;
            const double* pa = V + (va ? rowA : 0);
#pragma unroll
            for (int ks = 0; ks < 8; ++ks) {
                lf_load4(pa + (size_t) (4 * ks + l4) * ldv, xa[ks]);
                if (!va) { xa[ks][0] = 0.0; xa[ks][1] = 0.0; xa[ks][2] = 0.0; xa[ks][3] = 0.0; }
            }
        }
        for (int pr = blockIdx.y; pr < npairs; pr += gridDim.y) {
            v4d cc[2][4];                                                    // [tile of the pair][t], reg rr: row base + 4 l4 + 16 rr + t
            double wb[2][8];
            double* pc[2];
#pragma unroll
            for (int tl = 0; tl < 2; ++tl) {
                const int col = 32 * pr + 16 * tl + l15;
#pragma unroll
                for (int ks = 0; ks < 8; ++ks) wb[tl][ks] = W[(size_t) col * LF_PW + 4 * ks + l4];
                pc[tl] = C + (size_t) col * ldc;
#pragma unroll
                for (int rr = 0; rr < 4; ++rr) {
                    const int row0 = base + 4 * l4 + 16 * rr;
                    const bool vr = row0 < mk;
                    double tmp[4];
                    lf_load4(pc[tl] + (vr ? row0 : 0), tmp);
#pragma unroll
                    for (int t = 0; t < 4; ++t) cc[tl][t][rr] = vr ? tmp[t] : 0.0;
                }
            }
#pragma unroll
            for (int tl = 0; tl < 2; ++tl)
#pragma unroll
                for (int t = 0; t < 4; ++t)
#pragma unroll
                    for (int ks = 0; ks < 8; ++ks)
                        cc[tl][t] = __builtin_amdgcn_mfma_f64_16x16x4f64(xa[ks][t], wb[tl][ks], cc[tl][t], 0, 0, 1);   // neg A: C -= V W
#pragma unroll
            for (int tl = 0; tl < 2; ++tl)
#pragma unroll
                for (int rr = 0; rr < 4; ++rr) {
                    const int row0 = base + 4 * l4 + 16 * rr;
                    if (row0 < mk) lf_store4(pc[tl] + row0, cc[tl][0][rr], cc[tl][1][rr], cc[tl][2][rr], cc[tl][3][rr]);
                }
            if (gram && pr == 0) {
                // rows below the current leaf's 32 x 32 diagonal block only: in the wave that owns rows 0..63, registers rr = 0, 1
                // (rows 4 l4 + 16 rr + t < 32) stay out of the product
                const bool top = (base == 0);
#pragma unroll
                for (int t = 0; t < 4; ++t)
#pragma unroll
                    for (int rr = 0; rr < 4; ++rr) {
                        double f[2];
#pragma unroll
                        for (int ti = 0; ti < 2; ++ti) f[ti] = (rr < 2 && top) ? 0.0 : cc[ti][t][rr];
#pragma unroll
                        for (int ti = 0; ti < 2; ++ti)
#pragma unroll
                            for (int tj = 0; tj < 2; ++tj)
                                g[ti][tj] = __builtin_amdgcn_mfma_f64_16x16x4f64(f[ti], f[tj], g[ti][tj], 0, 0, 0);
                    }
            }
        }
    }
    if (gram) {
        // D reg r of lane (l15, l4) of tile (ti, tj) = G(16 ti + l4 + 4 r, 16 tj + l15); fixed summation order over the waves
        double* red = lf_red + wave * LF_PW * LF_PW;
#pragma unroll
        for (int ti = 0; ti < 2; ++ti)
#pragma unroll
            for (int tj = 0; tj < 2; ++tj)
#pragma unroll
                for (int r = 0; r < 4; ++r) red[(16 * tj + l15) * LF_PW + 16 * ti + l4 + 4 * r] = g[ti][tj][r];
        __syncthreads();
        for (int e = tid; e < LF_PW * LF_PW; e += LF_PT) {
            double s = 0.0;
#pragma unroll
            for (int v = 0; v < LF_PT / 64; ++v) s += lf_red[v * LF_PW * LF_PW + e];
            gslabs[(size_t) blockIdx.x * LF_PW * LF_PW + e] = s;
        }
    }
}

static inline bool lf_al16(const void* p, int ld) { return (((uintptr_t) p) & 15) == 0 && (ld & 1) == 0; }

extern "C" {

int qrd_leaf_fused_init(void)
{
    return (int) hipFuncSetAttribute(reinterpret_cast<const void*>(leaf_update_gram_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                     (int) (8 * LF_PW * LF_PW * sizeof(double)));
}

// A_rest (mk x N, ldc) -= V (mk x 32, ldv) W (32 x N, ld 32), N a multiple of 32.  gslabs != NULL: also *nslab partial Gram matrices
// (32 x 32 each, ld 32, summed = A_rest(32:mk, 0:32)^T A_rest(32:mk, 0:32) AFTER the update) for the next leaf's CholeskyQR2.
// gy: column pairs side by side (<= 0: all of them).  Returns -7 when the shapes / alignment do not fit (caller: plain product).
int qrd_leaf_update_gram(void* stream, int mk, int N, const double* V, int ldv, const double* W, double* C, int ldc, double* gslabs,
                         size_t gslab_cap, int gy, int* nslab)
{
    if (nslab) *nslab = 0;
    if (mk < 64 || (mk & 3) || N < 32 || (N & 31) || !lf_al16(V, ldv) || !lf_al16(C, ldc)) return -7;
    int nblk = (mk + LF_PT - 1) / LF_PT;
    const int npairs = N / 32;
    if (nblk > LF_MAXSLAB) nblk = LF_MAXSLAB;
    if (gslabs && (size_t) nblk * LF_PW * LF_PW > gslab_cap) gslabs = nullptr;
    if (gy <= 0 || gy > npairs) gy = npairs;
    const size_t shm = gslabs ? 8 * LF_PW * LF_PW * sizeof(double) : 0;
    hipLaunchKernelGGL(leaf_update_gram_kernel, dim3(nblk, gy), dim3(LF_PT), shm, (hipStream_t) stream, mk, npairs, V, ldv, W, C, ldc, gslabs);
    if (nslab && gslabs) *nslab = nblk;
    return (int) hipGetLastError();
}

}   // extern "C"
