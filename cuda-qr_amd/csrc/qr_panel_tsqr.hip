// qr_panel_tsqr.hip -- leaf panel factorisation as an intra-GPU TSQR with Householder reconstruction.
//
// Replaces the reference's one-block serial panel kernel (panelHouseholderKernel, qr.cu:60-333; host loop
// qr.c:109-235) for a leaf of w <= 32 columns over mk rows.  Instead of one grid-wide dependency per
// Householder column (one kernel launch per column: 33 launches and ~49 passes over the leaf), the leaf is
// cut into row blocks of <= 512 rows; each workgroup holds its block in registers (one row per thread) and
// runs all w Householder columns with workgroup-local synchronisation only:
//
//   F  tsqr_factor_kernel   every block: local Householder QR -> R_b (w x w), local reflectors
//      (repeated on the stacked R_b while the stack is taller than 512 rows)
//   T  tsqr_top_kernel      one block: QR of the last stack -> R~, and its explicit Q applied to [I;0]
//   A  tsqr_apply_kernel    down the tree: Q1_b = Q_local_b [C_b ; 0]  -> explicit Q1 (mk x w)
//   H1 hr_top_kernel        Householder reconstruction (Ballard/Demmel/Grigori/Jacquelin/Nguyen/Solomonik):
//                           sign matrix S, LU(Q1_top - S) = L1 U, T = -U S L1^-T, R = S R~, tau = diag(T)
//   H2 hr_rows_kernel       V(w:mk, :) = Q1(w:mk, :) U^-1   (row-parallel)
//
// The result is an ordinary compact-WY panel (unit-lower V in place below R, tau, T), identical in form to what
// the per-column kernels produce, so everything downstream is unchanged; I - V T V^T has first w columns Q1 S.
// 4 launches (mk <= 8192), 6 (mk <= 131072), 2 passes over the leaf.  mk <= 512: panel_single_kernel, 1 launch.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>
#include "qr_device.h"
#include "qr_common.h"
#include "qr_gemm_tile.h"
#include "qr_leaf_math.h"
#include "qr_factor32.h"

#define PT 512            // threads per workgroup = rows per block
#define PWAVES (PT / 64)

template <int NW>
struct PanelSharedT {
    double part[NW][PW];
    double row[PW];
    double s[PW];
    double tau[PW];
    double scal[4];
};
typedef PanelSharedT<PWAVES> PanelShared;

// rows [start, start+rows) of block b: even split when chunk == 0 (level 1), fixed chunks otherwise
__device__ __forceinline__ void block_range(int rows_total, int chunk, int b, int nb, int& start, int& rows)
{
    if (chunk > 0) {
        start = b * chunk;
        rows = min(chunk, rows_total - start);
    } else {
        start = (int) ((long long) b * rows_total / nb);
        rows = (int) ((long long) (b + 1) * rows_total / nb) - start;
    }
}

// One Householder column on a workgroup-resident block (row r of the block in x[], one row per thread).
// Same arithmetic as leaf_step_kernel (dlarfg convention; tau = 0 for an exactly-zero tail; qr.c:144-167 for the
// reference's form).  ZCAP: also record Z(c, J) = v_c^T v_J for c < J (needed only when T is built from Z).
template <int J, bool ZCAP, int NT, int RPT>
__device__ __forceinline__ void house_step(double (&x)[RPT][PW], int r0, int rows, int w, PanelSharedT<NT / 64>& sh,
                                           double (*Z)[PW + 1], int tid, int lane, int wave)
{
    if (J >= w) return;                              // wave-uniform
    {
        double prod[PW];
#pragma unroll
        for (int c = 0; c < PW; ++c) prod[c] = 0.0;
#pragma unroll
        for (int q = 0; q < RPT; ++q) {
            const int r = r0 + NT * q;
            const double xj = ((r > J) && (r < rows)) ? x[q][J] : 0.0;
#pragma unroll
            for (int c = 0; c < PW; ++c) prod[c] += xj * x[q][c];
        }
        if (tid == J) {                              // local row J is row 0 of thread J
#pragma unroll
            for (int c = 0; c < PW; ++c) sh.row[c] = x[0][c];
        }
        const double v = wave_reduce32(prod, lane);
        if ((lane & 1) == 0) sh.part[wave][lane >> 1] = v;
    }
    __syncthreads();
    if (tid < PW) {
        double d = 0.0;
#pragma unroll
        for (int p = 0; p < NT / 64; ++p) d += sh.part[p][tid];
        const double alpha = sh.row[J];
        const double sigma = __shfl(d, J);
        double t, b, iu;
        if (sigma == 0.0) { t = 0.0; b = alpha; iu = 0.0; }
        else {
            const double nrm = sqrt(alpha * alpha + sigma);
            b = -copysign(nrm, alpha);
            t = (b - alpha) / b;
            iu = 1.0 / (alpha - b);
        }
        const double sc = sh.row[tid] + d * iu;
        sh.s[tid] = sc;
        if (ZCAP && tid < J) Z[J][tid] = sc;
        if (tid == 0) { sh.scal[0] = t; sh.scal[1] = b; sh.scal[2] = iu; sh.tau[J] = t; }
    }
    __syncthreads();
    const double tj = sh.scal[0], beta = sh.scal[1], iu = sh.scal[2];
#pragma unroll
    for (int q = 0; q < RPT; ++q) {
        const int r = r0 + NT * q;
        const bool below = (r > J) && (r < rows), diag = (r == J);
        const double vi = below ? x[q][J] * iu : (diag ? 1.0 : 0.0);
        const double coef = tj * vi;
        x[q][J] = below ? vi : (diag ? beta : x[q][J]);
#pragma unroll
        for (int c = J + 1; c < PW; ++c) x[q][c] -= coef * sh.s[c];
    }
}

template <int J, bool ZCAP, int NT, int RPT>
__device__ __forceinline__ void factor_all(double (&x)[RPT][PW], int r0, int rows, int w, PanelSharedT<NT / 64>& sh,
                                           double (*Z)[PW + 1], int tid, int lane, int wave)
{
    house_step<J, ZCAP, NT, RPT>(x, r0, rows, w, sh, Z, tid, lane, wave);
    if constexpr (J + 1 < PW) factor_all<J + 1, ZCAP, NT, RPT>(x, r0, rows, w, sh, Z, tid, lane, wave);
}

// Compact-WY application of a block's Q to [Cin; 0]:  out(r,:) = Cin(r,:) - V(r,:) M,  M = T (V1^T Cin)
// (V1 = unit-lower top w x w of the block).  No reductions over rows: M is w x w and every row is independent,
// so the tree is walked down at GEMM speed instead of one workgroup-wide reduction per reflector.
// Ml must hold V1^T Cin on entry in Wl... see callers; here: given M in LDS, compute this thread's row.
__device__ __forceinline__ void wy_row(const double (&x)[PW], double (&out)[PW], int r, int w, double (*Ml)[PW + 1])
{
#pragma unroll
    for (int k = 0; k < PW; ++k) {
        if (k < w) {
            const double vk = (k < r) ? x[k] : (k == r ? 1.0 : 0.0);
#pragma unroll
            for (int q = 0; q < PW; ++q) out[q] -= vk * Ml[k][q];
        }
        __builtin_amdgcn_sched_barrier(0);        // keep the 1024 LDS reads from being hoisted (VGPR blow-up)
    }
}

// M = T (V1^T C) for one block without ever forming T:  T^-1 = striu(Z) + diag(1/tau)  (Z(i,k) = v_i^T v_k), so
// M solves  T^-1 M = V1^T C  by back substitution; column q of M lives in the registers of thread q (no
// synchronisation inside the solve).  tau_i = 0 (H_i = I) makes row i of T zero: M(i,:) = 0.
// LDS operands: V1 (unit-lower top of the block), Cl (w x w input), Zl[k][i] = Z(i,k) for i < k, tl[i] = tau_i.
template <int NT>
__device__ __forceinline__ void small_m(double (*V1)[PW + 1], double (*Cl)[PW + 1], double (*Zl)[PW + 1], const double* tl,
                                        double (*Wl)[PW + 1], double (*Ml)[PW + 1], int w, int tid)
{
    for (int e = tid; e < PW * PW; e += NT) {           // Wl = V1^T Cl   (V1 unit lower: V1(k, i) for k >= i)
        const int i = e / PW, q = e % PW;
        double acc = 0.0;
        for (int k = i; k < w; ++k) acc += V1[k][i] * Cl[k][q];
        Wl[i][q] = acc;
    }
    __syncthreads();
    if (tid < PW) {
        double mcol[PW];
#pragma unroll
        for (int i = PW - 1; i >= 0; --i) {
            double acc = 0.0;
            if (i < w) {
                acc = Wl[i][tid];
#pragma unroll
                for (int k = i + 1; k < PW; ++k) acc -= Zl[k][i] * mcol[k];      // Zl[k][i] = 0 for k >= w
                acc *= tl[i];                                                     // (1/tau_i)^-1; 0 when tau_i = 0
            }
            mcol[i] = acc;
            __builtin_amdgcn_sched_barrier(0);
        }
#pragma unroll
        for (int i = 0; i < PW; ++i) Ml[i][tid] = mcol[i];
    }
    __syncthreads();
}

__device__ __forceinline__ void load_block(double (&x)[PW], const double* __restrict__ src, int ld, int start, int r,
                                           int rows, int w)
{
    const double* p = src + start + min(r, rows - 1);
#pragma unroll
    for (int c = 0; c < PW; ++c) {
        const double v = (c < w) ? p[(size_t) c * ld] : 0.0;     // c < w is wave-uniform; the row index is clamped
        x[c] = (r < rows) ? v : 0.0;
    }
}

// F: local QR of every block of `src` (rows_total x w).  Leaves the factored block (R on top, reflector tails
// below) in Vloc, its tau in tauloc[b*PW..], the Gram entries Z(i,k) = v_i^T v_k (the strict upper triangle of T^-1)
// in Tloc[b*PW*PW + k*PW + i], and its R (w x w, zeros below the diagonal) in rows [b*w, b*w+w) of Rstack.
template <int NT, int RPT>
__device__ __forceinline__ void tsqr_factor_body(int b, int nblk, const double* __restrict__ src, int lds, int rows_total, int chunk,
                                                 int w, double* __restrict__ Vloc, int ldv,
                                                 double* __restrict__ tauloc, double* __restrict__ Tloc,
                                                 double* __restrict__ Rstack, int ldr)
{
    __shared__ PanelSharedT<NT / 64> sh;
    __shared__ double Z[PW][PW + 1];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    int start, rows;
    block_range(rows_total, chunk, b, nblk, start, rows);
    double x[RPT][PW];
#pragma unroll
    for (int q = 0; q < RPT; ++q) load_block(x[q], src, lds, start, tid + NT * q, rows, w);
    factor_all<0, true, NT, RPT>(x, tid, rows, w, sh, Z, tid, lane, wave);
    __syncthreads();
#pragma unroll
    for (int q = 0; q < RPT; ++q) {
        const int r = tid + NT * q;
        if (r < rows) {
#pragma unroll
            for (int c = 0; c < PW; ++c)
                if (c < w) Vloc[(size_t) c * ldv + start + r] = x[q][c];
        }
    }
    if (tid < w) {
        tauloc[b * PW + tid] = sh.tau[tid];
#pragma unroll
        for (int c = 0; c < PW; ++c)
            if (c < w) Rstack[(size_t) c * ldr + b * w + tid] = (c >= tid && tid < rows) ? x[0][c] : 0.0;
    }
    for (int e = tid; e < PW * PW; e += NT) {           // Zloc[b][k][i] = Z(i, k) = v_i^T v_k (i < k < w), else 0
        const int i = e % PW, k = e / PW;
        Tloc[(size_t) b * PW * PW + e] = (i < k && k < w) ? Z[k][i] : 0.0;
    }
}

template <int NT, int RPT>
__global__ __launch_bounds__(NT) void tsqr_factor_kernel(const double* __restrict__ src, int lds, int rows_total, int chunk,
                                                         int w, double* __restrict__ Vloc, int ldv,
                                                         double* __restrict__ tauloc, double* __restrict__ Tloc,
                                                         double* __restrict__ Rstack, int ldr, const int* __restrict__ guard)
{
    if (guard && *guard == 0) return;        // the CholeskyQR2 leaf succeeded: this fallback launch is a no-op
    tsqr_factor_body<NT, RPT>(blockIdx.x, gridDim.x, src, lds, rows_total, chunk, w, Vloc, ldv, tauloc, Tloc, Rstack, ldr);
}

// T: the last stack (rows <= 512): R~ -> Rt (ld PW), explicit Q_top [I;0] -> Cout (rows x w) in compact-WY form:
// Q_top [I;0] = [I;0] - V (T V1^T)
template <int NT, int RPT>
__device__ __forceinline__ void tsqr_top_body(const double* __restrict__ stack, int lds, int rows, int w,
                                              double* __restrict__ Rt, double* __restrict__ Cout, int ldc)
{
    __shared__ PanelSharedT<NT / 64> sh;
    __shared__ double Z[PW][PW + 1];
    __shared__ double V1[PW][PW + 1], Cl[PW][PW + 1], Wl[PW][PW + 1], Ml[PW][PW + 1];
    __shared__ double tl[PW];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    double x[RPT][PW];
#pragma unroll
    for (int q = 0; q < RPT; ++q) load_block(x[q], stack, lds, 0, tid + NT * q, rows, w);
    factor_all<0, true, NT, RPT>(x, tid, rows, w, sh, Z, tid, lane, wave);
    __syncthreads();
    if (tid < PW) {
        tl[tid] = (tid < w) ? sh.tau[tid] : 0.0;
#pragma unroll
        for (int c = 0; c < PW; ++c) {
            if (tid < w && c < w) Rt[c * PW + tid] = (c >= tid) ? x[0][c] : 0.0;
            V1[tid][c] = (tid < w && c < w) ? ((c < tid) ? x[0][c] : (c == tid ? 1.0 : 0.0)) : 0.0;
            Cl[tid][c] = (tid == c && tid < w) ? 1.0 : 0.0;
            if (!(c < tid && tid < w)) Z[tid][c] = 0.0;          // keep only Z(k = tid, i = c < k)
        }
    }
    __syncthreads();
    small_m<NT>(V1, Cl, Z, tl, Wl, Ml, w, tid);
#pragma unroll
    for (int q = 0; q < RPT; ++q) {
        const int r = tid + NT * q;
        double out[PW];
#pragma unroll
        for (int c = 0; c < PW; ++c) out[c] = (r == c && r < w) ? 1.0 : 0.0;
        wy_row(x[q], out, r, w, Ml);
        if (r < rows) {
#pragma unroll
            for (int c = 0; c < PW; ++c)
                if (c < w) Cout[(size_t) c * ldc + r] = out[c];
        }
    }
}

template <int NT, int RPT>
__global__ __launch_bounds__(NT) void tsqr_top_kernel(const double* __restrict__ stack, int lds, int rows, int w,
                                                      double* __restrict__ Rt, double* __restrict__ Cout, int ldc, const int* __restrict__ guard)
{
    if (guard && *guard == 0) return;        // the CholeskyQR2 leaf succeeded: this fallback launch is a no-op
    tsqr_top_body<NT, RPT>(stack, lds, rows, w, Rt, Cout, ldc);
}

// A: Cout(block rows, :) = Q_local_b [Cin_b ; 0] = [Cin_b; 0] - V_b (T_b V_b1^T Cin_b),
// Cin_b = rows [b*w, b*w+w) of the parent's output
// the five 32 x 33 LDS arrays (+ one vector) the walk-down bodies work in: declared by the calling kernel, so that kernels which run
// several bodies one after the other (the one-launch guard routes) hold ONE set
struct WalkLds {
    double (*Zl)[PW + 1]; double (*V1)[PW + 1]; double (*Cl)[PW + 1]; double (*Wl)[PW + 1]; double (*Ml)[PW + 1];
    double* tl;
};
#define WALK_LDS_DECL(name)                                                                                              \
    __shared__ double name##_a[5][PW][PW + 1];                                                                           \
    __shared__ double name##_t[PW];                                                                                      \
    const WalkLds name = {name##_a[0], name##_a[1], name##_a[2], name##_a[3], name##_a[4], name##_t}

template <int NT>
__device__ __forceinline__ void tsqr_apply_body(int b, int nblk, const WalkLds& L, const double* __restrict__ Vloc, int ldv,
                                                const double* __restrict__ tauloc,
                                                const double* __restrict__ Tloc, int rows_total, int chunk, int w,
                                                const double* __restrict__ Cin, int ldci,
                                                double* __restrict__ Cout, int ldco)
{
    double (*Zl)[PW + 1] = L.Zl; double (*V1)[PW + 1] = L.V1; double (*Cl)[PW + 1] = L.Cl; double (*Wl)[PW + 1] = L.Wl;
    double (*Ml)[PW + 1] = L.Ml; double* tl = L.tl;
    const int tid = threadIdx.x;
    int start, rows;
    block_range(rows_total, chunk, b, nblk, start, rows);
    double x[PW];
    load_block(x, Vloc, ldv, start, tid, rows, w);
    for (int e = tid; e < PW * PW; e += NT) {
        const int i = e % PW, c = e / PW;
        Zl[c][i] = Tloc[(size_t) b * PW * PW + e];                           // Zl[k][i] = Z(i, k)
        Cl[i][c] = (i < w && c < w) ? Cin[(size_t) c * ldci + b * w + i] : 0.0;
    }
    if (tid < PW) {
        tl[tid] = (tid < w) ? tauloc[b * PW + tid] : 0.0;
#pragma unroll
        for (int c = 0; c < PW; ++c)
            V1[tid][c] = (tid < w && c < w) ? ((c < tid) ? x[c] : (c == tid ? 1.0 : 0.0)) : 0.0;
    }
    __syncthreads();
    small_m<NT>(V1, Cl, Zl, tl, Wl, Ml, w, tid);
    double out[PW];
#pragma unroll
    for (int q = 0; q < PW; ++q) out[q] = (tid < w) ? Cl[min(tid, PW - 1)][q] : 0.0;
    wy_row(x, out, tid, w, Ml);
    if (tid < rows) {
#pragma unroll
        for (int q = 0; q < PW; ++q)
            if (q < w) Cout[(size_t) q * ldco + start + tid] = out[q];
    }
}

template <int NT>
__global__ __launch_bounds__(NT) void tsqr_apply_kernel(const double* __restrict__ Vloc, int ldv,
                                                        const double* __restrict__ tauloc,
                                                        const double* __restrict__ Tloc, int rows_total, int chunk, int w,
                                                        const double* __restrict__ Cin, int ldci,
                                                        double* __restrict__ Cout, int ldco, const int* __restrict__ guard)
{
    if (guard && *guard == 0) return;        // the CholeskyQR2 leaf succeeded: this fallback launch is a no-op
    WALK_LDS_DECL(L);
    tsqr_apply_body<NT>(blockIdx.x, gridDim.x, L, Vloc, ldv, tauloc, Tloc, rows_total, chunk, w, Cin, ldci, Cout, ldco);
}

// H1: Householder reconstruction on the top w x w block of Q1 = Q_local_0 [C_0; 0].  One 16-wave workgroup; lane
// r < 32 of wave g owns row r of the columns c = g + 16 q (q < 2).  Rows are broadcast inside a wave with
// v_readlane (the row a step needs always lives in a lane of the same wave), the multipliers come from LDS, so
// only the LU needs a barrier per step; the three substitutions run without any synchronisation:
//   M_0 = T_0 V_01^T C_0 by back substitution with T_0^-1 = striu(Z_0) + diag(1/tau_0);  Q1_top = C_0 - V_01 M_0
//   modified LU without pivoting:  S_j = -sign(pivot), pivot -= S_j (|pivot| >= 1 afterwards),  Q1_top - S = L1 U
//   T = -U S L1^-T  as the forward substitution  L1 X = -S U^T,  X = T^T ;   Uinv = U^-1
// Outputs: R = S R~ and L1 into the top of the panel, tau = diag(T), T, the unit-lower top of Vw, Umat = U^-1.
#define HQ 2
#define HG (PW / HQ)      /* waves in hr_top_kernel: wave g owns columns g + HG q */

template <int I, int HGx = HG> struct HrStep {
    // back substitution step i (descending): row I of M is final; eliminate it from rows < I
    static __device__ __forceinline__ void msolve(double (&e)[PW / HGx], int r, int w, const double* t0, double (*Zs)[PW + 1])
    {
        if (I < w) {
            const double ti = t0[I];
            const double z = (r < I) ? Zs[r][I] : 0.0;
#pragma unroll
            for (int q = 0; q < PW / HGx; ++q) {
                e[q] = (r == I) ? e[q] * ti : e[q];
                e[q] -= z * readlane_f64(e[q], I);
            }
        }
        if constexpr (I > 0) HrStep<I - 1, HGx>::msolve(e, r, w, t0, Zs);
    }
    // Q1_top(r, :) -= V1(r, k) M(k, :), k ascending
    static __device__ __forceinline__ void q1top(double (&b)[PW / HGx], const double (&e)[PW / HGx], int r, int w, double (*V1)[PW + 1])
    {
        if (I < w) {
            const double v = (I <= r) ? V1[r][I] : 0.0;
#pragma unroll
            for (int q = 0; q < PW / HGx; ++q) b[q] -= v * readlane_f64(e[q], I);
        }
        if constexpr (I + 1 < PW) HrStep<I + 1, HGx>::q1top(b, e, r, w, V1);
    }
    // modified LU step j = I (ascending); column j multipliers go through LDS (one barrier per step)
    static __device__ __forceinline__ void lu(double (&b)[PW / HGx], int r, int g, int w, double (*colb)[PW], double* Ss)
    {
        if (I < w) {
            constexpr int pp = I & 1, qj = I / HGx, gj = I % HGx;
            if (g == gj && r < PW) colb[pp][r] = b[qj];
            __syncthreads();
            const double p = colb[pp][I];
            const double S = (p >= 0.0) ? -1.0 : 1.0;
            const double piv = p - S;
            const double l = (r > I && r < w) ? colb[pp][r] / piv : 0.0;
#pragma unroll
            for (int q = 0; q < PW / HGx; ++q) {
                const int c = g + HGx * q;
                const double u = readlane_f64(b[q], I);
                if (c > I) b[q] -= l * u;
            }
            if (g == gj) {
                b[qj] = (r > I) ? l : ((r == I) ? piv : b[qj]);
                if (r == I) Ss[I] = S;
            }
        }
        if constexpr (I + 1 < PW) HrStep<I + 1, HGx>::lu(b, r, g, w, colb, Ss);
    }
    // forward substitution L1 X = RHS, step k = I (ascending): row I of X is final
    static __device__ __forceinline__ void xsolve(double (&x)[PW / HGx], int r, int w, double (*Bs)[PW + 1])
    {
        if (I < w) {
            const double l = (r > I && r < w) ? Bs[r][I] : 0.0;
#pragma unroll
            for (int q = 0; q < PW / HGx; ++q) x[q] -= l * readlane_f64(x[q], I);
        }
        if constexpr (I + 1 < PW) HrStep<I + 1, HGx>::xsolve(x, r, w, Bs);
    }
    // back substitution U Uinv = I, step k = I (descending)
    static __device__ __forceinline__ void uinv(double (&ui)[PW / HGx], int r, int w, double (*Bs)[PW + 1])
    {
        if (I < w) {
            const double d = 1.0 / Bs[I][I];
            const double u = (r < I) ? Bs[r][I] : 0.0;
#pragma unroll
            for (int q = 0; q < PW / HGx; ++q) {
                ui[q] = (r == I) ? ui[q] * d : ui[q];
                ui[q] -= u * readlane_f64(ui[q], I);
            }
        }
        if constexpr (I > 0) HrStep<I - 1, HGx>::uinv(ui, r, w, Bs);
    }
};

template <int HGx>
__device__ __forceinline__ void hr_top_body(const double* __restrict__ Vloc, int ldvl, const double* __restrict__ tau0,
                                                     const double* __restrict__ Z0, const double* __restrict__ C0, int ldc0,
                                                     const double* __restrict__ Rt, int w, double* __restrict__ A, int lda,
                                                     double* __restrict__ tau, double* __restrict__ T, int ldt,
                                                     double* __restrict__ Vw, int ldv, double* __restrict__ Umat)
{
    __shared__ double Bs[PW][PW + 1], V1[PW][PW + 1], Cs[PW][PW + 1];
    __shared__ double colb[2][PW];
    __shared__ double Ss[PW], t0[PW];
    const int lane = threadIdx.x & 63, g = threadIdx.x >> 6;
    const int r = lane;                          // lanes >= 32 carry zeros and write nothing
    const bool ra = r < w;
    for (int el = threadIdx.x; el < PW * PW; el += 64 * HGx) {
        const int i = el % PW, c = el / PW;
        const bool in = (i < w && c < w);
        const double v = in ? Vloc[(size_t) c * ldvl + i] : 0.0;
        V1[i][c] = in ? ((c < i) ? v : (c == i ? 1.0 : 0.0)) : 0.0;
        Cs[i][c] = in ? C0[(size_t) c * ldc0 + i] : 0.0;
        Bs[i][c] = (in && i < c) ? Z0[c * PW + i] : 0.0;          // Z(i, c), i < c
    }
    if (threadIdx.x < PW) t0[threadIdx.x] = (threadIdx.x < w) ? tau0[threadIdx.x] : 0.0;
    __syncthreads();
    double e[PW / HGx], b[PW / HGx];
#pragma unroll
    for (int q = 0; q < PW / HGx; ++q) {               // W = V1^T C0
        const int c = g + HGx * q;
        double acc = 0.0;
        if (r < PW)
            for (int k = r; k < w; ++k) acc += V1[k][r] * Cs[k][c];
        e[q] = acc;
        b[q] = (ra && c < w) ? Cs[r][c] : 0.0;
    }
    HrStep<PW - 1, HGx>::msolve(e, r, w, t0, Bs);     // e := M_0 rows
    HrStep<0, HGx>::q1top(b, e, ra ? r : -1, w, V1);  // b := Q1_top rows (inactive lanes: no update)
    __syncthreads();                             // everyone is done with Bs as Z
    HrStep<0, HGx>::lu(b, r, g, w, colb, Ss);
    __syncthreads();
    if (r < PW) {
#pragma unroll
        for (int q = 0; q < PW / HGx; ++q) Bs[r][g + HGx * q] = b[q];
    }
    __syncthreads();
    double x[PW / HGx], ui[PW / HGx];
#pragma unroll
    for (int q = 0; q < PW / HGx; ++q) {
        const int c = g + HGx * q;
        x[q] = (ra && c <= r) ? -Ss[r] * Bs[c][r] : 0.0;          // -S U^T
        ui[q] = (ra && r == c) ? 1.0 : 0.0;
    }
    HrStep<0, HGx>::xsolve(x, r, w, Bs);
    HrStep<PW - 1, HGx>::uinv(ui, r, w, Bs);
    if (!ra) return;
#pragma unroll
    for (int q = 0; q < PW / HGx; ++q) {
        const int c = g + HGx * q;
        if (c < w) {
            T[(size_t) r * ldt + c] = x[q];                                        // T(c, r) = X(r, c)
            A[(size_t) c * lda + r] = (c >= r) ? Ss[r] * Rt[c * PW + r] : b[q];
            Vw[(size_t) c * ldv + r] = (c < r) ? b[q] : (c == r ? 1.0 : 0.0);
            Umat[c * PW + r] = (c >= r) ? ui[q] : 0.0;                             // Umat := U^-1 (ld PW)
            if (c == r) tau[r] = x[q];
        }
    }
}

__global__ __launch_bounds__(64 * HG) void hr_top_kernel(const double* __restrict__ Vloc, int ldvl, const double* __restrict__ tau0,
                                                     const double* __restrict__ Z0, const double* __restrict__ C0, int ldc0,
                                                     const double* __restrict__ Rt, int w, double* __restrict__ A, int lda,
                                                     double* __restrict__ tau, double* __restrict__ T, int ldt,
                                                     double* __restrict__ Vw, int ldv, double* __restrict__ Umat, const int* __restrict__ guard)
{
    if (guard && *guard == 0) return;        // the CholeskyQR2 leaf succeeded: this fallback launch is a no-op
    hr_top_body<HG>(Vloc, ldvl, tau0, Z0, C0, ldc0, Rt, w, A, lda, tau, T, ldt, Vw, ldv, Umat);
}

// A1 + H2 fused (level 1, last launch of the leaf): V(rows, :) = Q1(rows, :) U^-1 with Q1 = [C_b; 0] - V_b M_b, i.e.
// V(r, :) = [C_b U^-1; 0](r, :) - V_b(r, :) (M_b U^-1), written straight into the panel and Vw for global rows >= w
// (the top w rows were written by hr_top_kernel).  No explicit Q1 round trip through memory.
__device__ __forceinline__ void tsqr_final_body(int bid, const WalkLds& L, double (*Ui)[PW + 1], const double* __restrict__ Vloc, int ldvl,
                                                const double* __restrict__ tauloc, const double* __restrict__ Tloc,
                                                int rows_total, int nblk, int halves, int w,
                                                const double* __restrict__ Cin, int ldci,
                                                const double* __restrict__ Umat, double* __restrict__ A, int lda,
                                                double* __restrict__ Vw, int ldv)
{
    double (*Zl)[PW + 1] = L.Zl; double (*V1)[PW + 1] = L.V1; double (*Cl)[PW + 1] = L.Cl; double (*Wl)[PW + 1] = L.Wl;
    double (*Ml)[PW + 1] = L.Ml; double* tl = L.tl;
    const int tid = threadIdx.x, b = bid / halves, h = bid % halves;
    int bstart, brows;
    block_range(rows_total, 0, b, nblk, bstart, brows);             // the level-0 block
    const int start = bstart + h * PT, rows = min(PT, brows - h * PT);   // this workgroup's rows of it
    double x[PW];
    load_block(x, Vloc, ldvl, start, tid, max(rows, 1), w);
    for (int e = tid; e < PW * PW; e += PT) {
        const int i = e % PW, c = e / PW;
        const bool in = (i < w && c < w);
        Zl[c][i] = Tloc[(size_t) b * PW * PW + e];                           // Zl[k][i] = Z(i, k)
        Cl[i][c] = in ? Cin[(size_t) c * ldci + b * w + i] : 0.0;
        Ui[i][c] = (i <= c && c < w) ? Umat[c * PW + i] : 0.0;
        const double v = in ? Vloc[(size_t) c * ldvl + bstart + i] : 0.0;    // unit-lower top of the block
        V1[i][c] = in ? ((c < i) ? v : (c == i ? 1.0 : 0.0)) : 0.0;
    }
    if (tid < PW) tl[tid] = (tid < w) ? tauloc[b * PW + tid] : 0.0;
    __syncthreads();
    small_m<PT>(V1, Cl, Zl, tl, Wl, Ml, w, tid);
    for (int e = tid; e < PW * PW; e += PT) {           // Wl = Ml Ui ;  Zl (reused) = Cl Ui     (Ui upper triangular)
        const int i = e / PW, q = e % PW;
        double a1 = 0.0, a2 = 0.0;
        for (int k = 0; k <= q && k < w; ++k) { a1 += Ml[i][k] * Ui[k][q]; a2 += Cl[i][k] * Ui[k][q]; }
        Wl[i][q] = a1;
        Zl[i][q] = a2;
    }
    __syncthreads();
    const int rb = h * PT + tid;                        // row index inside the level-0 block
    double out[PW];
#pragma unroll
    for (int q = 0; q < PW; ++q) out[q] = (rb < w) ? Zl[min(rb, PW - 1)][q] : 0.0;
    wy_row(x, out, rb, w, Wl);
    if (tid < rows && start + tid >= w) {
        const size_t rg = (size_t) start + tid;
#pragma unroll
        for (int q = 0; q < PW; ++q)
            if (q < w) { A[(size_t) q * lda + rg] = out[q]; Vw[(size_t) q * ldv + rg] = out[q]; }
    }
}

__global__ __launch_bounds__(PT) void tsqr_final_kernel(const double* __restrict__ Vloc, int ldvl,
                                                        const double* __restrict__ tauloc, const double* __restrict__ Tloc,
                                                        int rows_total, int nblk, int halves, int w,
                                                        const double* __restrict__ Cin, int ldci,
                                                        const double* __restrict__ Umat, double* __restrict__ A, int lda,
                                                        double* __restrict__ Vw, int ldv, const int* __restrict__ guard)
{
    if (guard && *guard == 0) return;        // the CholeskyQR2 leaf succeeded: this fallback launch is a no-op
    WALK_LDS_DECL(L);
    __shared__ double Ui[PW][PW + 1];
    tsqr_final_body(blockIdx.x, L, Ui, Vloc, ldvl, tauloc, Tloc, rows_total, nblk, halves, w, Cin, ldci, Umat, A, lda, Vw, ldv);
}

// ---------------------------------------------------------------------------------------------------------
// The whole Householder-TSQR guard route of a SHORT leaf (one tree level: mk <= 16 blocks of 1024 rows) as ONE launch.
// Behind a CholeskyQR2 leaf the guard route is almost never needed, and as four separate launches it cost four dependent
// kernel boundaries (~5 us each, 20 us of a ~140 us leaf) just to find the guard word clear.  Here the four phases
// (factor | top | reconstruction | final) are separated by software grid barriers instead: nblk <= 16 workgroups, one per
// compute unit at most, all co-resident; a monotone arrival counter with agent-scope release / acquire around it
// (MI355X_MICROARCH.md, "barrier-counter": ~7 us each, paid only when the guard trips).  The counter is zeroed by the leaf's
// reconstruction kernel (hr3_kernel, which always runs before this launch), so nothing has to be reset here.  Spins are
// bounded: a barrier that is not completed after ~1 s poisons tau with NaN and lets the launch end instead of hanging.
// ---------------------------------------------------------------------------------------------------------
__device__ __forceinline__ bool coop_barrier(unsigned* cnt, unsigned target)
{
    __shared__ int timed_out;
    __syncthreads();                                              // every wave has waited for its own stores (vmcnt(0))
    if (threadIdx.x == 0) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");        // write back what this workgroup produced
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __hip_atomic_fetch_add(cnt, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        unsigned spins = 0;
        int to = 0;
        while (__hip_atomic_load(cnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
            __builtin_amdgcn_s_sleep(16);
            if (++spins > (1u << 20)) { to = 1; break; }
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");        // drop this CU's stale lines before anyone reads
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        timed_out = to;
    }
    __syncthreads();
    return timed_out == 0;
}

__device__ __forceinline__ void tsqr_coop_body(const double* P, int ld, int mk, int w, int nblk, int halves,
                                               double* __restrict__ Vloc1, double* __restrict__ taus, double* __restrict__ Ts,
                                               double* __restrict__ stack, double* __restrict__ Rt, double* __restrict__ Ctop,
                                               double* __restrict__ Umat, double* A /* == P */, int lda,
                                               double* __restrict__ tau, double* __restrict__ T, int ldt,
                                               double* __restrict__ Vw, int ldv, unsigned* __restrict__ bar)
{
    const int b = blockIdx.x, G = nblk, rows_stack = nblk * w;      // participants: workgroups 0 .. nblk-1 (the launch may hold more)
    bool ok = true;
    WALK_LDS_DECL(L);
    __shared__ double Ui[PW][PW + 1];
    // F: local Householder QR of this workgroup's row block (<= 1024 rows: two rows per thread)
    tsqr_factor_body<PT, 2>(b, nblk, P, ld, mk, 0, w, Vloc1, mk, taus, Ts, stack, rows_stack);
    ok = coop_barrier(bar, (unsigned) G) && ok;
    // T: QR of the stacked R factors and its explicit Q [I; 0]
    if (b == 0) tsqr_top_body<PT, 1>(stack, rows_stack, rows_stack, w, Rt, Ctop, rows_stack);
    ok = coop_barrier(bar, 2u * (unsigned) G) && ok;
    // H: Householder reconstruction of the top block
    if (b == 0) hr_top_body<PT / 64>(Vloc1, mk, taus, Ts, Ctop, rows_stack, Rt, w, A, lda, tau, T, ldt, Vw, ldv, Umat);
    ok = coop_barrier(bar, 3u * (unsigned) G) && ok;
    // Fin: every block writes its rows of V
    for (int h = 0; h < halves; ++h) {
        tsqr_final_body(b * halves + h, L, Ui, Vloc1, mk, taus, Ts, mk, nblk, halves, w, Ctop, rows_stack, Umat, A, lda, Vw, ldv);
        __syncthreads();
    }
    if (!ok && b == 0 && threadIdx.x < w) tau[threadIdx.x] = __builtin_nan("");
}

__global__ __launch_bounds__(PT) void tsqr_coop_kernel(const double* P, int ld, int mk, int w, int nblk, int halves,
                                                       double* __restrict__ Vloc1, double* __restrict__ taus, double* __restrict__ Ts,
                                                       double* __restrict__ stack, double* __restrict__ Rt, double* __restrict__ Ctop,
                                                       double* __restrict__ Umat, double* A /* == P */, int lda,
                                                       double* __restrict__ tau, double* __restrict__ T, int ldt,
                                                       double* __restrict__ Vw, int ldv, unsigned* __restrict__ bar,
                                                       const int* __restrict__ guard)
{
    if (*guard == 0) return;                 // the CholeskyQR2 leaf succeeded: nothing to do, one kernel boundary paid
    tsqr_coop_body(P, ld, mk, w, nblk, halves, Vloc1, taus, Ts, stack, Rt, Ctop, Umat, A, lda, tau, T, ldt, Vw, ldv, bar);
}

// ---------------------------------------------------------------------------------------------------------
// The same for TALL leaves (two or three tree levels, up to 2^21 rows): the six guard launches -- level-0 factor, upper-level
// factor(s), top, walk down, reconstruction, final -- as ONE launch of G persistent workgroups (G = min(level-0 blocks, compute units
// of the stream): one per CU, co-resident) that walk their blocks grid-stride between software grid barriers.  Six launches that found
// the guard word clear cost 28 us per 32-column leaf: 0.46 ms of a 6.9 ms 262144 x 512 factorisation, a quarter of a 65536-row leaf.
// ---------------------------------------------------------------------------------------------------------
#define TALL_MAXL 3              /* upper levels 1, 2: leaves of up to 4 M rows; the level loops are unrolled (constant indices into the by-value tree) */
struct TallTree {
    int L;                          // upper levels 1 .. L (0: the level-0 R factors go straight to the top)
    int nblk0, halves;              // level-0 blocks (1024 rows: halves = 2) and final-phase workgroup items per block
    int rows[TALL_MAXL], nblk[TALL_MAXL], chunk[TALL_MAXL];
    long long off[TALL_MAXL], tau[TALL_MAXL];   // level l's input stack / reflectors / coefficients at stacks|Vup|Cup + off, taus at + tau
    long long top_off; int top_rows;            // the last stack
};

__device__ __forceinline__ void tsqr_coop_tall_body(const TallTree& t, int G, const double* P, int ld, int mk, int w,
                                                    double* __restrict__ Vloc1, double* __restrict__ Vup, double* __restrict__ taus,
                                                    double* __restrict__ Ts, double* __restrict__ stacks, double* __restrict__ Cup,
                                                    double* __restrict__ Rt, double* __restrict__ Umat, double* A, int lda,
                                                    double* __restrict__ tau, double* __restrict__ T, int ldt,
                                                    double* __restrict__ Vw, int ldv, unsigned* __restrict__ bar, unsigned& tgt)
{
    const int wg = blockIdx.x;
    bool ok = true;
    WALK_LDS_DECL(L);
    __shared__ double Ui[PW][PW + 1];
    // level 0: local Householder QR of the 1024-row blocks
    for (int b = wg; b < t.nblk0; b += G) {
        tsqr_factor_body<PT, 2>(b, t.nblk0, P, ld, mk, 0, w, Vloc1, mk, taus, Ts, stacks, t.nblk0 * w);
        __syncthreads();
    }
    tgt += (unsigned) G; ok = coop_barrier(bar, tgt) && ok;
    // up the tree: QR of the stacked R factors, 512 rows (16 factors) per block
#pragma unroll
    for (int l = 1; l < TALL_MAXL; ++l) {
        if (l > t.L) break;
        for (int b = wg; b < t.nblk[l]; b += G) {
            tsqr_factor_body<PT, 2>(b, t.nblk[l], stacks + t.off[l], t.rows[l], t.rows[l], t.chunk[l], w, Vup + t.off[l], t.rows[l],
                                    taus + t.tau[l], Ts + t.tau[l] * PW, stacks + t.off[l] + (long long) t.rows[l] * PW, t.nblk[l] * w);
            __syncthreads();
        }
        tgt += (unsigned) G; ok = coop_barrier(bar, tgt) && ok;
    }
    // top: factor + explicit Q of the last stack
    if (wg == 0) tsqr_top_body<PT, 1>(stacks + t.top_off, t.top_rows, t.top_rows, w, Rt, Cup + t.top_off, t.top_rows);
    tgt += (unsigned) G; ok = coop_barrier(bar, tgt) && ok;
    // down the tree
    const double* Cin = Cup + t.top_off;
    int ldci = t.top_rows;
#pragma unroll
    for (int l = TALL_MAXL - 1; l >= 1; --l) {
        if (l > t.L) continue;
        for (int b = wg; b < t.nblk[l]; b += G) {
            tsqr_apply_body<PT>(b, t.nblk[l], L, Vup + t.off[l], t.rows[l], taus + t.tau[l], Ts + t.tau[l] * PW, t.rows[l], t.chunk[l], w,
                                Cin, ldci, Cup + t.off[l], t.rows[l]);
            __syncthreads();
        }
        tgt += (unsigned) G; ok = coop_barrier(bar, tgt) && ok;
        Cin = Cup + t.off[l];
        ldci = t.rows[l];
    }
    // Householder reconstruction of the top block, then every block writes its rows of V
    if (wg == 0) hr_top_body<PT / 64>(Vloc1, mk, taus, Ts, Cin, ldci, Rt, w, A, lda, tau, T, ldt, Vw, ldv, Umat);
    tgt += (unsigned) G; ok = coop_barrier(bar, tgt) && ok;
    for (int hb = wg; hb < t.nblk0 * t.halves; hb += G) {
        tsqr_final_body(hb, L, Ui, Vloc1, mk, taus, Ts, mk, t.nblk0, t.halves, w, Cin, ldci, Umat, A, lda, Vw, ldv);
        __syncthreads();
    }
    if (!ok && wg == 0 && threadIdx.x < w) tau[threadIdx.x] = __builtin_nan("");
}

// mk <= 512: the whole leaf in one workgroup -- V, R, tau and T in a single launch.
__global__ __launch_bounds__(PT) void panel_single_kernel(double* __restrict__ P, int ld, int mk, int w,
                                                          double* __restrict__ tau, double* __restrict__ T, int ldt,
                                                          double* __restrict__ Vw, int ldv)
{
    __shared__ PanelShared sh;
    __shared__ double Z[PW][PW + 1];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    double x1[1][PW];
    load_block(x1[0], P, ld, 0, tid, mk, w);
    factor_all<0, true, PT, 1>(x1, tid, mk, w, sh, Z, tid, lane, wave);
    double (&x)[PW] = x1[0];
    __syncthreads();
    if (tid < mk) {
#pragma unroll
        for (int c = 0; c < PW; ++c)
            if (c < w) {
                P[(size_t) c * ld + tid] = x[c];
                Vw[(size_t) c * ldv + tid] = (tid > c) ? x[c] : (tid == c ? 1.0 : 0.0);
            }
    }
    __shared__ double Tl[PW][PW + 1];
    build_t_rows(Tl, Z, sh, w, tid);
    if (tid < w) {
#pragma unroll
        for (int q = 0; q < PW; ++q)
            if (q < w) T[(size_t) q * ldt + tid] = Tl[tid][q];
        tau[tid] = sh.tau[tid];
    }
}


// =========================================================================================================
// CholeskyQR2 + Householder reconstruction leaf (fast path; the Householder TSQR kernels above are its guard):
//   G1 = A^T A (split-K MFMA product)            R1 = chol(G1),  Q = A R1^-1  -> Vw          cholq_kernel
//   G2 = Q^T Q                                   refuse unless max|G2 - I| <= 1/64 (then a second pass brings Q to
//   R2 = chol(G2), Q1 = Q R2^-1, R~ = R2 R1      working-precision orthogonality: Yamamoto et al., CholeskyQR2)
//   Householder reconstruction of Q1 exactly as in hr_top_kernel: LU(Q1_top - S) = L1 U, T = -U S L1^-T,
//   R = S R~, V = Q1 U^-1 = Q (R2^-1 U^-1)                                                   hr2_kernel, final2_kernel
// A zero, dependent or badly conditioned leaf (cond above ~1e4..1e7) makes a pivot non-positive or G2 far from I; the
// guard word then stays 1, nothing of the leaf has been overwritten, and the Householder TSQR launches that follow do
// the work.  The result has the same form either way: unit-lower V below R, tau, T.
// =========================================================================================================


// right-looking Cholesky G = R^T R on one wave: lane j (< 32) holds column j in g[]; on exit g[k] = R(k, j), k <= j
template <int K> struct CholStep {
    static __device__ __forceinline__ void run(double (&g)[PW], int lane, bool& ok)
    {
        const double p = readlane_f64(g[K], K);
        ok = ok && (p > 0.0);                       // false for NaN as well
        const double inv = rsqrt_newton(p);
        const double rk = (lane >= K) ? g[K] * inv : 0.0;
        g[K] = rk;
#pragma unroll
        for (int i = K + 1; i < PW; ++i) g[i] -= readlane_f64(rk, i) * rk;
        if constexpr (K + 1 < PW) CholStep<K + 1>::run(g, lane, ok);
    }
};

// wave 0: Gs (symmetric, identity outside w x w) -> Rs (upper, zero below), returns ok (wave-uniform)
__device__ __forceinline__ bool chol_wave(const double* __restrict__ G, int w, int lane, double (*Rs)[PW + 1])
{
    double g[PW];
    const int j = lane & (PW - 1);
#pragma unroll
    for (int i = 0; i < PW; ++i) g[i] = (i < w && j < w) ? G[(size_t) j * PW + i] : (i == j ? 1.0 : 0.0);
    bool ok = true;
    CholStep<0>::run(g, j, ok);
    if (lane < PW) {
#pragma unroll
        for (int k = 0; k < PW; ++k) Rs[k][lane] = (k <= lane) ? g[k] : 0.0;
    }
    return ok;
}

// wave 0: X = R^-1 (upper) by back substitution, lane j computes column j; Rs in LDS
__device__ __forceinline__ void triu_inv_wave(double (*Rs)[PW + 1], int lane, double (*Xs)[PW + 1])
{
    double x[PW];
#pragma unroll
    for (int i = PW - 1; i >= 0; --i) {
        double acc = (i == lane) ? 1.0 : 0.0;
#pragma unroll
        for (int k = i + 1; k < PW; ++k) acc -= Rs[i][k] * x[k];
        x[i] = (i <= lane) ? acc / Rs[i][i] : 0.0;
        __builtin_amdgcn_sched_barrier(0);
    }
    if (lane < PW) {
#pragma unroll
        for (int i = 0; i < PW; ++i) Xs[i][lane] = x[i];
    }
}

// G_z = A_z^T A_z for the K slice z of a tall mk x 32 leaf (A = the leaf itself or its Q): a streaming kernel instead of the
// general split-K TN product with A = B -- the slice is loaded ONCE (the TN kernel fetches it as both operands), 512
// contiguous bytes per column per step (128 there), and 16 MFMAs per wave between two barriers (4 there).  Slab z (32 x 32,
// ld 32) goes to slabs + z*1024 for slab_reduce_kernel.  Rows past mk read as zero.  16-byte aligned A, even ld and mk (else the general TN product).
#define GKB 64
#define GLD (GKB + 2)
__global__ __launch_bounds__(256) void gram32_kernel(const double* __restrict__ A, int ld, int mk, int rows_per,
                                                     double* __restrict__ slabs)
{
    __shared__ __attribute__((aligned(16))) double img[2][PW * GLD];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wi = wave & 1, wj = wave >> 1, l15 = lane & 15, l4 = lane >> 4;
    const int kbeg = blockIdx.x * rows_per, kend = min(mk, kbeg + rows_per);
    const int nk = (kend - kbeg + GKB - 1) / GKB;
    v2d reg[4];
    auto gload = [&](int k0) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int idx = tid + 256 * q, col = idx >> 5, k = k0 + 2 * (idx & 31);
            const v2d v = *reinterpret_cast<const v2d*>(A + (size_t) col * ld + min(k, mk - 2));
            reg[q] = (v2d){(k < kend) ? v[0] : 0.0, (k + 1 < kend) ? v[1] : 0.0};
        }
    };
    auto sstore = [&](int buf) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int idx = tid + 256 * q;
            *reinterpret_cast<v2d*>(&img[buf][(idx >> 5) * GLD + 2 * (idx & 31)]) = reg[q];
        }
    };
    v4d acc = (v4d){0.0, 0.0, 0.0, 0.0};
    if (nk > 0) { gload(kbeg); sstore(0); }
    __syncthreads();
    for (int kt = 0; kt < nk; ++kt) {
        const int buf = kt & 1;
        if (kt + 1 < nk) gload(kbeg + (kt + 1) * GKB);
        const double* ia = &img[buf][(16 * wi + l15) * GLD];
        const double* ib = &img[buf][(16 * wj + l15) * GLD];
#pragma unroll
        for (int ks = 0; ks < GKB / 4; ++ks)
            acc = __builtin_amdgcn_mfma_f64_16x16x4f64(ib[4 * ks + l4], ia[4 * ks + l4], acc, 0, 0, 0);
        if (kt + 1 < nk) sstore(buf ^ 1);
        __syncthreads();
    }
    // D reg r of lane l holds [p = l4 + 4r][q = l15] with p indexing the first operand (ib: column 16wj + p of G) and q the
    // second (ia: row 16wi + q): G(16wi + l15, 16wj + l4 + 4r)
    double* out = slabs + (size_t) blockIdx.x * PW * PW;
#pragma unroll
    for (int r = 0; r < 4; ++r) out[(16 * wj + l4 + 4 * r) * PW + 16 * wi + l15] = acc[r];
}

// G (32 x 32, ld 32) = A^T A for a tall mk x 32 block: gram32_kernel slabs + slab_reduce (through qrd_slab_reduce32)
static int gram32(hipStream_t s, const double* A, int ld, int mk, double* G, double* slabs, size_t slab_cap)
{
    const bool ok = ((reinterpret_cast<uintptr_t>(A) & 15) == 0) && (ld % 2 == 0) && (mk % 2 == 0) && mk >= 2;
    if (!ok) return qrd_gemm_tn((void*) s, PW, PW, mk, 1.0, A, ld, A, ld, 0.0, G, PW, slabs, slab_cap, nullptr, 0);
    int nslab = (mk + 127) / 128;                      // 128 rows (2 steps) per workgroup, at most 512 workgroups: the short
                                                       // leaves of square problems are latency-bound, the tall ones bandwidth-bound
    if (nslab > 512) nslab = 512;
    if ((size_t) nslab * PW * PW > slab_cap) nslab = (int) (slab_cap / (PW * PW));
    if (nslab < 1) return -3;
    int rows_per = (mk + nslab - 1) / nslab;
    rows_per = (rows_per + GKB - 1) / GKB * GKB;
    nslab = (mk + rows_per - 1) / rows_per;
    hipLaunchKernelGGL(gram32_kernel, dim3(nslab), dim3(256), 0, s, A, ld, mk, rows_per, slabs);
    return qrd_slab_reduce(s, PW, PW, nslab, slabs, PW, (size_t) PW * PW, G, PW);
}

// =========================================================================================================
// Second-generation CholeskyQR2 leaf: the same mathematics in 4 launches instead of 7 (plus the guard launches).
//   gram32_kernel          G1 slabs (at most CQ2_MAXSLAB of them, so that a consumer can sum them itself)
//   cholq3_kernel          (short leaves) every workgroup: G1 = sum of slabs, R1 = chol(G1) (one wave), its rows of Q = A R1^-1 -> Vw,
//                          AND its share of G2 = Q^T Q on the MFMA pipe (rows staged through LDS) -> slab2[b];
//   chol1 + cholq4_tall    (tall leaves) the same with the Cholesky in a launch of its own and a streaming pass behind it
//   hr3_kernel             one workgroup: G2 = sum of slab2 (or a pre-reduced G2), guard, R2 = chol(G2), then the modified LU
//                          of the reconstruction run directly on Q_top:  with U' = U R2,
//                              LU(Q_top - S R2) = L1 U'     (same L1, same S as LU(Q_top R2^-1 - S) = L1 U: R2 is upper
//                          triangular with a positive diagonal, so Schur complements only get multiplied by R2's trailing
//                          block; row j of S R2 is added when step j fixes S_j = -sign of the current (j, j) entry),
//                          so neither R2^-1 nor Q1_top = Q_top R2^-1 is ever formed.  Then, concurrently on different
//                          waves, U = U' R2^-1 (row solves) and L1^-1 (column solves), T = -U S L1^-T as a product,
//                          R = S R2 R1.
//   final3_kernel          rows >= w:  V = Q U'^-1 as the row-parallel triangular solve v U' = q  (V = Q1 U^-1 =
//                          Q R2^-1 U^-1 = Q (U R2)^-1)
// The old route needed R2^-1, U^-1 and their product M (three more 32-step recurrences) plus two slab_reduce launches and a
// second gram32 launch with its own pass over Q.
// =========================================================================================================
// elements 2t, 2t+1 of the sum of `nslab` dense 32 x 32 slabs (1024 doubles apart), slabs added in index order; 16 loads
// are issued before the first add (the previous two-at-a-time loop paid one L2 round trip per pair: ~10 us for 14 slabs)
template <int BATCH = 16>
__device__ __forceinline__ v2d slab_sum2(const double* __restrict__ slabs, int nslab, int t)
{
    const v2d* p = reinterpret_cast<const v2d*>(slabs) + t;
    v2d acc = (v2d){0.0, 0.0};
    for (int z0 = 0; z0 < nslab; z0 += BATCH) {
        v2d v[BATCH];
#pragma unroll
        for (int u = 0; u < BATCH; ++u) v[u] = p[(size_t) min(z0 + u, nslab - 1) * (PW * PW / 2)];
#pragma unroll
        for (int u = 0; u < BATCH; ++u)
            if (z0 + u < nslab) acc += v[u];
    }
    return acc;
}

// Phase stamps of the leaf kernels (development builds only: make STAMPS=1; devtools/tools_leaf_stamps.py reads them):
// thread 0 of workgroup 0 records the 100 MHz wall clock at phase boundaries.
#ifdef QRD_STAMPS
__device__ long long qrd_dbg_stamps[64];
// stamps stay in scalar registers until the kernel's last line (a store per stamp in the middle of the unrolled recurrences
// cost 3.5 KB of scratch per thread)
#define STAMP_DECL long long qst_[8] = {0, 0, 0, 0, 0, 0, 0, 0}
#define STAMP(k) qst_[(k) & 7] = (long long) __builtin_amdgcn_s_memrealtime()
#define STAMP_FLUSH(base, n) do { if (threadIdx.x == 0 && blockIdx.x == 0) for (int q_ = 0; q_ < (n); ++q_) qrd_dbg_stamps[(base) + q_] = qst_[q_]; } while (0)
#else
#define STAMP_DECL do { } while (0)
#define STAMP(k) do { } while (0)
#define STAMP_FLUSH(base, n) do { } while (0)
#endif

#define CQ2_MAXSLAB 32
#define CQ2_LDS_DOUBLES(HALF) (PW * ((HALF) + 2) + 2 * PW * (PW + 1) + PW + 8)

// ---------------------------------------------------------------------------------------------------------
// Third-generation leaf: the two row solves (q = a R1^-1 and v = q U'^-1 as per-thread triangular solves) were 7 us each and
// bound by LDS bandwidth -- every thread reads all 528 entries of the triangle as broadcast reads, 8 waves x 496 x 512 B per
// workgroup.  Here they are products with the explicit inverses on the matrix cores, the row operands going from global memory
// straight into the MFMA operand layout (no LDS):
//   * R1^-1 costs nothing: the Cholesky wave has 32 idle lanes, which run the same elimination on the columns of the identity
//     (G = R^T R, so the row operations that turn G into R turn I into R^-T);
//   * U'^-1 is a third 32-step recurrence in hr3_kernel<true>, on a wave that only copied data before.
// Transposed products, so that the result lands column-fast for the stores:  q^T = (R1^-1)^T a^T : A operand (i = column, lane
// l15; k, lane l4) = R1^-1(k, i), B operand (k; j = row, lane l15) = a(row, k), D reg rr of lane (l4, l15) = q(row l15, column
// l4 + 4 rr): a wave's store instruction covers 4 columns x 16 consecutive rows.
// ---------------------------------------------------------------------------------------------------------

template <int NT, int HALF>
__global__ __launch_bounds__(NT) void cholq3_kernel(const double* __restrict__ P, int ld, int mk,
                                                    const double* __restrict__ gslabs, int nslab, double* __restrict__ R1,
                                                    double* __restrict__ Vw, int ldv, double* __restrict__ slab2,
                                                    int* __restrict__ guard)
{
    extern __shared__ __attribute__((aligned(16))) double cq_smem[];
    double* Qs = cq_smem;                                                    // [PW][(HALF + 2)]; later NT/64 partial Grams [PW*PW]
    double (*Gs)[PW + 1] = reinterpret_cast<double (*)[PW + 1]>(cq_smem + PW * (HALF + 2));
    double (*Ws)[PW + 1] = reinterpret_cast<double (*)[PW + 1]>(cq_smem + PW * (HALF + 2) + PW * (PW + 1));   // Ws[k][c] = R1^-1(k, c)
    int* okf = reinterpret_cast<int*>(cq_smem + PW * (HALF + 2) + 2 * PW * (PW + 1) + PW);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, b = blockIdx.x;
    const int l15 = lane & 15, l4 = lane >> 4;
    const int wrow0 = b * NT + wave * 64;                                    // this wave's 64 rows: four 16-row tiles
    double bq[4][8];                                                         // B operands: a(row0 + 16 t + l15, 4 ks + l4)
    STAMP_DECL;
    STAMP(0);
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        const int row = wrow0 + 16 * t + l15;
        const double* p = P + min(row, mk - 1);
#pragma unroll
        for (int ks = 0; ks < 8; ++ks) bq[t][ks] = p[(size_t) (4 * ks + l4) * ld];          // in flight under the Gram sum
    }
    for (int t2 = tid; t2 < PW * PW / 2; t2 += NT) {
        const v2d gsum = slab_sum2(gslabs, nslab, t2);
        const int e = 2 * t2;
        Gs[e / PW][e % PW] = gsum[0];                                         // Gs[j][i] = G(i, j)
        Gs[e / PW][e % PW + 1] = gsum[1];
    }
    __syncthreads();
    STAMP(1);
    if (tid < 64) {
        // Bring G to O(1) by an even power of two first (exact): next to the underflow threshold the Gram part of the factorisation would
        // round to multiples of 4.9e-324 while the identity part stays in the normal range, and R1 and "its" inverse would stop
        // being inverses of each other (a panel scaled by 1e-160 lost 8 digits).  G' = 4^-h G = R'^T R', R = 2^h R', R^-1 = 2^-h R'^-1.
        int e2 = 0;
        {
            const double d = Gs[0][0];
            if (d > 0.0 && d < 1.7e308) { (void) frexp(d, &e2); e2 &= ~1; }
        }
        const double rs = ldexp(1.0, e2 / 2), ws = ldexp(1.0, -(e2 / 2));       // 4^-h itself may not be representable (G ~ 1e-317)
        // (round 5: on the matrix cores, qr_factor32.h; the register recurrence CholAugStep took 7-8 us of every workgroup here)
        const bool wr = b == 0;
        const bool ok = chol32_mfma(lane, [&](int i, int j) { return (Gs[j][i] * ws) * ws; },
                                    [&](int i, int j, double v) { if (wr) R1[j * PW + i] = v * rs; },             // R1 column-major
                                    [&](int i, int j, double v) { Ws[j][i] = v * ws; });                           // Ws[j][k] = R1^-1(j, k)
        if (lane == 0) *okf = ok ? 1 : 0;
    }
    __syncthreads();
    STAMP(2);
    const bool ok = *okf != 0;
    if (b == 0 && tid == 0) *guard = ok ? 0 : 1;
    if (!ok) return;                                                          // workgroup-uniform
    v4d q[4][2];
    {
        double aw[2][8];                                                      // A operands: R1^-1(4 ks + l4, 16 ti + l15)
#pragma unroll
        for (int ti = 0; ti < 2; ++ti)
#pragma unroll
            for (int ks = 0; ks < 8; ++ks) aw[ti][ks] = Ws[4 * ks + l4][16 * ti + l15];
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const bool live = wrow0 + 16 * t + l15 < mk;
#pragma unroll
            for (int ti = 0; ti < 2; ++ti) {
                v4d acc = (v4d){0.0, 0.0, 0.0, 0.0};
#pragma unroll
                for (int ks = 0; ks < 8; ++ks) {
                    if (ti == 0 && ks >= 4) continue;          // R1^-1 is upper triangular: rows k >= 16 of its first 16 columns are zero
                    acc = __builtin_amdgcn_mfma_f64_16x16x4f64(aw[ti][ks], live ? bq[t][ks] : 0.0, acc, 0, 0, 0);
                }
                q[t][ti] = acc;
            }
        }
    }
    STAMP(3);
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        const int row = wrow0 + 16 * t + l15;
        if (row < mk) {
#pragma unroll
            for (int ti = 0; ti < 2; ++ti)
#pragma unroll
                for (int rr = 0; rr < 4; ++rr) Vw[(size_t) (16 * ti + l4 + 4 * rr) * ldv + row] = q[t][ti][rr];
        }
    }
    STAMP(4);
    // ---- this workgroup's share of G2 = Q^T Q: rows staged HALF at a time as a [column][row] image
    v4d acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int jj = 0; jj < 2; ++jj) acc[i][jj] = (v4d){0.0, 0.0, 0.0, 0.0};
    const int myhalf = (wave * 64) / HALF;
#pragma unroll
    for (int h = 0; h < NT / HALF; ++h) {
        if (myhalf == h) {
#pragma unroll
            for (int t = 0; t < 4; ++t)
#pragma unroll
                for (int ti = 0; ti < 2; ++ti)
#pragma unroll
                    for (int rr = 0; rr < 4; ++rr)
                        Qs[(16 * ti + l4 + 4 * rr) * (HALF + 2) + ((wave * 64 + 16 * t + l15) & (HALF - 1))] = q[t][ti][rr];
        }
        __syncthreads();
        const double* base = Qs + 32 * wave + l4;
#pragma unroll
        for (int ks = 0; ks < 8; ++ks) {
            const double f0 = base[l15 * (HALF + 2) + 4 * ks], f1 = base[(16 + l15) * (HALF + 2) + 4 * ks];
            acc[0][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(f0, f0, acc[0][0], 0, 0, 0);
            acc[0][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(f0, f1, acc[0][1], 0, 0, 0);   // tile (1, 0) is its transpose (bitwise)
            acc[1][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(f1, f1, acc[1][1], 0, 0, 0);
        }
        __syncthreads();
    }
    STAMP(5);
    double* red = Qs + wave * PW * PW;
#pragma unroll
    for (int ti = 0; ti < 2; ++ti)
#pragma unroll
        for (int tj = ti; tj < 2; ++tj)
#pragma unroll
            for (int rr = 0; rr < 4; ++rr) {
                red[(16 * tj + l15) * PW + 16 * ti + l4 + 4 * rr] = acc[ti][tj][rr];
                if (ti != tj) red[(16 * ti + l4 + 4 * rr) * PW + 16 * tj + l15] = acc[ti][tj][rr];     // G2 is symmetric
            }
    __syncthreads();
    for (int e = tid; e < PW * PW; e += NT) {
        double sum = 0.0;
#pragma unroll
        for (int v = 0; v < NT / 64; ++v) sum += Qs[v * PW * PW + e];
        slab2[(size_t) b * PW * PW + e] = sum;
    }
    STAMP(6);
    STAMP_FLUSH(0, 7);
}

// ---------------------------------------------------------------------------------------------------------
// Tall leaves, streaming form (round 2, end): a kernel that factors G1 itself spends 7.8 us of every workgroup in a one-wave Cholesky with nothing
// in flight, solves q R1 = a on the vector ALUs with R1 broadcast from LDS, and stages Q through LDS for G2.  Here the Cholesky
// (with R1^-1, CholAugStep) runs ONCE in a one-wave kernel, and the pass over the leaf is pure streaming on the matrix cores:
// rows on the MFMA row index, each wave's 64 rows as four interleaved 16-row tiles (physical row 4 p + t: every global access is 32
// contiguous bytes per lane, qr_leaf_fused.hip), q = a R1^-1 with R1^-1 as the B operand, and the accumulators of q feed G2 = q^T q
// directly -- no LDS, no barrier in the loop.  Workgroups walk row blocks grid-stride and leave one partial G2 each.
// ---------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(64) void chol1_kernel(const double* __restrict__ G1, double* __restrict__ R1, double* __restrict__ Rinv,
                                                   int* __restrict__ guard)
{
    const int lane = threadIdx.x;
    int e2 = 0;                                            // power-of-two scaling as in cholq3_kernel
    {
        const double d = G1[0];
        if (d > 0.0 && d < 1.7e308) { (void) frexp(d, &e2); e2 &= ~1; }
    }
    const double rs = ldexp(1.0, e2 / 2), ws = ldexp(1.0, -(e2 / 2));
    // (round 5: on the matrix cores, qr_factor32.h)
    const bool ok = chol32_mfma(lane, [&](int i, int j) { return (G1[j * PW + i] * ws) * ws; },
                                [&](int i, int j, double v) { R1[j * PW + i] = v * rs; },                         // R1(i, j), column-major
                                [&](int i, int j, double v) { Rinv[i * PW + j] = v * ws; });                       // R1^-1(j, i) = R1^-T(i, j)
    if (lane == 0) *guard = ok ? 0 : 1;
}


// Q (mk x 32 -> Vw) = P R1^-1 and this workgroup's partial G2 = Q^T Q (slab2[blockIdx.x], 32 x 32, ld 32).  mk % 4 == 0, P / Vw 16-byte
// aligned with even leading dimensions.  Rinv: R1^-1 column-major ld 32.  512 threads; row blocks of 512 rows, grid-stride.
__global__ __launch_bounds__(PT) void cholq4_tall_kernel(const double* __restrict__ P, int ld, int mk, const double* __restrict__ Rinv,
                                                         double* __restrict__ Vw, int ldv, double* __restrict__ slab2,
                                                         const int* __restrict__ guard)
{
    extern __shared__ __attribute__((aligned(16))) double tq_red[];          // [8 waves][32 * 32]
    if (*guard != 0) return;                                                 // the Cholesky refused the leaf: nothing is written
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l15 = lane & 15, l4 = lane >> 4;
    double rb[2][8];                                                         // B operand: R1^-1(4 ks + l4, 16 ti + l15)
#pragma unroll
    for (int ti = 0; ti < 2; ++ti)
#pragma unroll
        for (int ks = 0; ks < 8; ++ks) rb[ti][ks] = Rinv[(16 * ti + l15) * PW + 4 * ks + l4];
    v4d g[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int jj = 0; jj < 2; ++jj) g[i][jj] = (v4d){0.0, 0.0, 0.0, 0.0};
    for (int base = blockIdx.x * PT + wave * 64; base < mk; base += gridDim.x * PT) {
        double xa[8][4];                                                     // a(base + 4 l15 + t, 4 ks + l4)
        {
            const int rowA = base + 4 * l15;
            const bool va = rowA < mk;
            const double* pa = P + (va ? rowA : 0);
#pragma unroll
            for (int ks = 0; ks < 8; ++ks) {
                tq_load4(pa + (size_t) (4 * ks + l4) * ld, xa[ks]);
                if (!va) { xa[ks][0] = 0.0; xa[ks][1] = 0.0; xa[ks][2] = 0.0; xa[ks][3] = 0.0; }
            }
        }
        v4d q[2][4];                                                         // [column tile][t], reg rr: row base + 4 l4 + 16 rr + t
#pragma unroll
        for (int ti = 0; ti < 2; ++ti)
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                v4d acc = (v4d){0.0, 0.0, 0.0, 0.0};
#pragma unroll
                for (int ks = 0; ks < 8; ++ks) {
                    if (ti == 0 && ks >= 4) continue;                        // R1^-1 upper triangular
                    acc = __builtin_amdgcn_mfma_f64_16x16x4f64(xa[ks][t], rb[ti][ks], acc, 0, 0, 0);
                }
                q[ti][t] = acc;
            }
#pragma unroll
        for (int ti = 0; ti < 2; ++ti) {
            double* pc = Vw + (size_t) (16 * ti + l15) * ldv;
#pragma unroll
            for (int rr = 0; rr < 4; ++rr) {
                const int row0 = base + 4 * l4 + 16 * rr;
                if (row0 < mk) {
                    *reinterpret_cast<v2d*>(pc + row0) = (v2d){q[ti][0][rr], q[ti][1][rr]};
                    *reinterpret_cast<v2d*>(pc + row0 + 2) = (v2d){q[ti][2][rr], q[ti][3][rr]};
                }
            }
        }
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int rr = 0; rr < 4; ++rr) {
                g[0][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(q[0][t][rr], q[0][t][rr], g[0][0], 0, 0, 0);
                g[0][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(q[0][t][rr], q[1][t][rr], g[0][1], 0, 0, 0);
                g[1][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(q[1][t][rr], q[1][t][rr], g[1][1], 0, 0, 0);
            }
    }
    // D reg r of lane (l15, l4) of tile (ti, tj) = G2(16 ti + l4 + 4 r, 16 tj + l15); tile (1, 0) mirrored; fixed order over the waves
    double* red = tq_red + wave * PW * PW;
#pragma unroll
    for (int ti = 0; ti < 2; ++ti)
#pragma unroll
        for (int tj = ti; tj < 2; ++tj)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                red[(16 * tj + l15) * PW + 16 * ti + l4 + 4 * r] = g[ti][tj][r];
                if (ti != tj) red[(16 * ti + l4 + 4 * r) * PW + 16 * tj + l15] = g[ti][tj][r];
            }
    __syncthreads();
    for (int e = tid; e < PW * PW; e += PT) {
        double sum = 0.0;
#pragma unroll
        for (int v = 0; v < PT / 64; ++v) sum += tq_red[v * PW * PW + e];
        slab2[(size_t) blockIdx.x * PW * PW + e] = sum;
    }
}

// rows >= 32 of V = Q U'^-1 (Winv: U'^-1, column-major, ld PW), written to Vw and to A.  No LDS, no barrier.
// rows_wg: rows per workgroup -- PT (every wave 64 rows) or PT / 2 (waves 0..3 only: twice the workgroups, i.e. compute units, for
// the same 64 matrix-core instructions per wave; two waves per SIMD took 3.7 us of the launch's 11, one takes 1.9)
// L1s != nullptr (early-product leaf): the top 32 rows of Vw still hold Q_top -- the unit-lower L1 that belongs there comes from L1s
__device__ __forceinline__ void final4_body(double* __restrict__ Vw, int ldv, int mk, const double* __restrict__ Winv,
                                            double* __restrict__ A, int lda, int rows_wg = PT, const double* __restrict__ L1s = nullptr)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, l15 = lane & 15, l4 = lane >> 4;
    if (wave * 64 >= rows_wg) return;
    const int wrow0 = blockIdx.x * rows_wg + wave * 64;
    if (wrow0 >= mk) return;
    double bq[4][8], aw[2][8];
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        const double* p = Vw + min(wrow0 + 16 * t + l15, mk - 1);
#pragma unroll
        for (int ks = 0; ks < 8; ++ks) bq[t][ks] = p[(size_t) (4 * ks + l4) * ldv];
    }
#pragma unroll
    for (int ti = 0; ti < 2; ++ti)
#pragma unroll
        for (int ks = 0; ks < 8; ++ks) aw[ti][ks] = Winv[(16 * ti + l15) * PW + 4 * ks + l4];
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        const int row = wrow0 + 16 * t + l15;
        v4d acc[2];
#pragma unroll
        for (int ti = 0; ti < 2; ++ti) {
            acc[ti] = (v4d){0.0, 0.0, 0.0, 0.0};
#pragma unroll
            for (int ks = 0; ks < 8; ++ks) {
                if (ti == 0 && ks >= 4) continue;              // U'^-1 is upper triangular: nothing below row 15 in its first 16 columns
                acc[ti] = __builtin_amdgcn_mfma_f64_16x16x4f64(aw[ti][ks], bq[t][ks], acc[ti], 0, 0, 0);
            }
        }
        if (row >= PW && row < mk) {
#pragma unroll
            for (int ti = 0; ti < 2; ++ti)
#pragma unroll
                for (int rr = 0; rr < 4; ++rr) {
                    const size_t c = (size_t) (16 * ti + l4 + 4 * rr);
                    Vw[c * ldv + row] = acc[ti][rr];
                    A[c * lda + row] = acc[ti][rr];
                }
        } else if (L1s && row < PW) {
#pragma unroll
            for (int ti = 0; ti < 2; ++ti)
#pragma unroll
                for (int rr = 0; rr < 4; ++rr) {
                    const size_t c = (size_t) (16 * ti + l4 + 4 * rr);
                    Vw[c * ldv + row] = L1s[c * PW + row];
                }
        }
    }
}

__global__ __launch_bounds__(PT) void final4_kernel(double* __restrict__ Vw, int ldv, int mk, const double* __restrict__ Winv,
                                                    double* __restrict__ A, int lda, const int* __restrict__ guard,
                                                    const double* __restrict__ L1s)
{
    if (*guard != 0) return;
    final4_body(Vw, ldv, mk, Winv, A, lda, PT, L1s);
}

// Cholesky + modified LU of the reconstruction on ONE wave, no LDS and no barrier: lane c (< 32) holds column c of both
// matrices in registers -- g[k] = R2(k, c) and b[r] = (Q_top - S R2)(r, c) being eliminated -- so a row of the pivot step is one
// register per lane and only the multiplier column (the registers of lane I) is broadcast, with v_readlane.  The 8-wave
// version with one workgroup barrier per step took 16 us of the kernel's 41; this is the same arithmetic in ~5 us.
// (See the header of this section for why the LU runs on Q_top - S R2 instead of Q_top R2^-1 - S.)
// On exit: b = L1 below the diagonal, U' = U R2 on and above it; sgn = S_c in lane c.
#define H3G 8

// UINV: Uout receives U'^-1 (third-generation leaf: final4 multiplies by it on the matrix cores) instead of U' (final3 solves with it)
// EP ("early product", see hr3_ep_kernel): the unit-lower top block L1 of V goes to epw (Vw's top rows still hold Q_top, which the
// product workgroups of the same launch are reading), followed by the two 32 x 32 matrices slab_reduce_ep_kernel folds in besides T:
//   epw + 0*1024: L1 (unit lower)        + 1*1024: U'^-1 (upper)        + 2*1024: B = S R2 (upper)        (all column-major, ld 32)
// -- written by waves that were storing anyway, so the reconstruction's own chain is not a step longer than without EP
template <bool UINV, bool EP>
__device__ __forceinline__ void hr3_body(const double* __restrict__ G2s, int nslab, const double* __restrict__ R1,
                                         double* __restrict__ Vw, int ldv, double* __restrict__ A, int lda,
                                         double* __restrict__ tau, double* __restrict__ T, int ldt,
                                         double* __restrict__ Uout, int* __restrict__ guard, unsigned* __restrict__ bar,
                                         double* __restrict__ epw)
{
    __shared__ double Bs[PW][PW + 1], R1s[PW][PW + 1], R2s[PW][PW + 1], Us[PW][PW + 1], Ls[PW][PW + 1], Gs[PW][PW + 1];

    __shared__ double Ss[PW], r2inv[PW];
    __shared__ int flags[3];
    const int tid = threadIdx.x, lane = tid & 63, g = tid >> 6;
    const int rc = lane & (PW - 1);
    STAMP_DECL;
    STAMP(16);
    if (tid == 0) *bar = 0u;                               // arrival counter of the one-launch guard route that follows (tsqr_coop_kernel)
    // the guard word is TESTED only after the loads below have been issued (behind the second barrier): an early return here
    // would put one more global round trip -- 2-4 us next to a running update GEMM -- in front of every other load
    const int gword = *guard;
    if (tid < 3) flags[tid] = 0;
    // Q_top -> Us (free until U is formed), two consecutive entries of a column per thread, in flight under the sum of the slabs below
    const v2d qtop = *reinterpret_cast<const v2d*>(Vw + (size_t) ((2 * tid) / PW) * ldv + (2 * tid) % PW);
    __syncthreads();
    {
        const v2d gsum = slab_sum2(G2s, nslab, tid);       // elements 2 tid, 2 tid + 1 of G2 (column-major, ld 32)
        const int e = 2 * tid, i = e % PW, c = e / PW;
        Gs[c][i] = gsum[0];                                // Gs[j][i] = G2(i, j)
        Gs[c][i + 1] = gsum[1];
        const double d0 = gsum[0] - (i == c ? 1.0 : 0.0), d1 = gsum[1] - (i + 1 == c ? 1.0 : 0.0);
        if (!(fabs(d0) <= QRD_GUARD_THR) || !(fabs(d1) <= QRD_GUARD_THR)) flags[0] = 1;     // also catches NaN
        if (!(fabs(d0) <= QRD_CHOL1_THR) || !(fabs(d1) <= QRD_CHOL1_THR)) flags[2] = 1;     // G2 - I too large for the first-order factor
        R1s[i][c] = (i <= c) ? R1[e] : 0.0;
        R1s[i + 1][c] = (i + 1 <= c) ? R1[e + 1] : 0.0;
        Us[i][c] = qtop[0];
        Us[i + 1][c] = qtop[1];
    }
    __syncthreads();
    STAMP(17);
    if (gword != 0) return;                                // pass 1 already refused
    if (flags[0]) { if (tid == 0) *guard = 1; return; }
    const bool first_order = flags[2] == 0;
    if (g == 0) {
        // Round 5: everything 32-step on this wave runs on the matrix cores (qr_factor32.h).  R2 = chol(G2) -- to first order when
        // G2 = I + E with max|E| <= 1e-9 (after one CholeskyQR pass E ~ cond(leaf)^2 eps: R2 = I + striu(E) + diag(E)/2 up to terms of
        // size n |E|^2 <= 3e-17, below the rounding of R2 itself), else chol32_mfma, which also leaves R2^-1 (in Gs) -- then the modified
        // LU with L1^-1 and U'^-1 (lu32_mfma): rounds 2-4 ran three more 32-step recurrences on waves 0-2 behind the LU for U, L1^-1, U'^-1.
        bool ok = true;
        if (first_order) {
            double gg[PW];
#pragma unroll
            for (int i = 0; i < PW; ++i) gg[i] = Gs[rc][i];
            if (lane < PW) {
#pragma unroll
                for (int i = 0; i < PW; ++i) R2s[i][lane] = (i < lane) ? gg[i] : (i == lane ? 1.0 + 0.5 * (gg[i] - 1.0) : 0.0);
            }
        } else {
            ok = chol32_mfma(lane, [&](int i, int j) { return Gs[j][i]; }, [&](int i, int j, double v) { R2s[i][j] = v; },
                             [&](int i, int j, double v) { Gs[j][i] = v; });                       // Gs[k][c] <- R2^-1(k, c) = R2^-T(c, k)
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        STAMP(18);
        if (ok) {
            lu32_mfma(lane, [&](int i, int j) { return Us[i][j]; }, [&](int i, int j) { return R2s[i][j]; },
                      [&](int i, int j, double v) { Bs[i][j] = v; }, [&](int i, double v) { Ss[i] = v; },
                      [&](int i, int j, double v) { Ls[i][j] = v; },                               // L1^-1(i, j)
                      [&](int i, int j, double v) {                                                // v = U'^-1(j, i): column-major outputs
                          if (UINV) Uout[i * PW + j] = v;
                          if (EP) epw[1 * PW * PW + i * PW + j] = v;
                      });
        } else if (lane == 0) flags[1] = 1;
    }
    __syncthreads();
    STAMP(19);
    if (flags[1]) { if (tid == 0) *guard = 1; return; }
    // wave 0: U = U' R2^-1 on the matrix cores (R2^-1 = 2 I - R2 to first order, else from Gs) -> Us; the other waves: R = S (R2 R1) and L1
    // -> A, unit-lower L1 -> Vw, U' -> Uout
    if (g == 0) {
        const int l15 = lane & 15, l4 = lane >> 4;
#pragma unroll
        for (int tile = 0; tile < 3; ++tile) {
            const int ti = (tile == 2) ? 1 : 0, tc = (tile == 0) ? 0 : 1;
            v4d acc = (v4d){0.0, 0.0, 0.0, 0.0};
#pragma unroll
            for (int ks = 0; ks < 8; ++ks) {
                const int k = 4 * ks + l4, i = 16 * ti + l15, cc = 16 * tc + l15;
                const double up = (k >= i) ? Bs[i][k] : 0.0;
                const double ri = first_order ? ((k == cc) ? 2.0 - R2s[k][cc] : -R2s[k][cc]) : Gs[k][cc];
                acc = f32_mma(up, (k <= cc) ? ri : 0.0, acc);
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int i = 16 * ti + l4 + 4 * r, cc = 16 * tc + l15;
                Us[i][cc] = (cc >= i) ? acc[r] : 0.0;
            }
        }
        if (lane < 16)
#pragma unroll
            for (int r = 0; r < 16; ++r) Us[16 + r][lane] = 0.0;
    } else {
        for (int el = tid - 64; el < PW * PW; el += 64 * H3G - 64) {
            const int i = el % PW, c = el / PW;
            if (c >= i) {
                double acc = 0.0;
                for (int k = i; k <= c; ++k) acc += R2s[i][k] * R1s[k][c];
                A[(size_t) c * lda + i] = Ss[i] * acc;
                if (!UINV) Uout[el] = Bs[i][c];
            } else {
                A[(size_t) c * lda + i] = Bs[i][c];
                if (!UINV) Uout[el] = 0.0;
            }
            const double l1 = (c < i) ? Bs[i][c] : (c == i ? 1.0 : 0.0);
            if (EP) {
                epw[el] = l1;                           // Vw's top rows are still being read as Q_top: the last leaf launch copies this
                epw[2 * PW * PW + el] = (c >= i) ? Ss[i] * R2s[i][c] : 0.0;                      // B = S R2 (upper triangular)
            } else Vw[(size_t) c * ldv + i] = l1;
        }
    }
    __syncthreads();
    STAMP(20);
    // T = -U S L1^-T:  T(i, c) = -sum_{k = i .. c} U(i, k) S_k L1^-1(c, k)      (upper triangular; T(i, i) = -U(i, i) S_i = tau_i)
    for (int el = tid; el < PW * PW; el += 64 * H3G) {
        const int i = el % PW, c = el / PW;
        double acc = 0.0;
        if (c >= i)
            for (int k = i; k <= c; ++k) acc -= Us[i][k] * Ss[k] * Ls[c][k];
        T[(size_t) c * ldt + i] = acc;
        if (i == c) tau[i] = acc;
    }
    STAMP(21);
    STAMP_FLUSH(16, 6);
}

template <bool UINV>
__global__ __launch_bounds__(64 * H3G, 2) void hr3_kernel(const double* __restrict__ G2s, int nslab, const double* __restrict__ R1,
                                                      double* __restrict__ Vw, int ldv, double* __restrict__ A, int lda,
                                                      double* __restrict__ tau, double* __restrict__ T, int ldt,
                                                      double* __restrict__ Uout, int* __restrict__ guard, unsigned* __restrict__ bar)
{
    hr3_body<UINV, false>(G2s, nslab, R1, Vw, ldv, A, lda, tau, T, ldt, Uout, guard, bar, nullptr);
}

// ---------------------------------------------------------------------------------------------------------
// "Early product": the leaf's long-K in-panel product in the SAME launch as its one-workgroup reconstruction.
// The reconstruction (hr3: 25-30 us on ONE workgroup, the rest of the chip idle) and the product V_l^T [A_rest | V_prev] that
// followed it (18 us on a short leaf, 80 us on a 262144-row one) were two links of every leaf's dependent chain.  But
//     V = (Q - [S R2; 0]) U'^-1        (rows >= 32: Q U'^-1; top block: L1)
// so  V^T X = U'^-T (Q^T X - (S R2)^T X_top): the long-K part Q^T X needs only Q, which exists BEFORE the reconstruction.  Workgroup
// 0 of this launch runs hr3; workgroups 1.. compute split-K slabs of Z = Q^T [A_rest | V_prev] (32 x 32 tiles on the MFMA pipe, the
// tile loop of the GEMM kernels); slab_reduce_ep_kernel then sums the slabs and folds in the 32 x 32 factors (one workgroup per column,
// three 32-term products each -- cheap there, a serial step each in the one reconstruction workgroup):
//     y = U'^-T (z - B^T x_top),   B = S R2       z = a column of Z, x_top = the top 32 rows of that column of [A_rest | V_prev]
//     W(:, j) = T^T y   (columns of A_rest)       G(j, :) = y^T   (columns of V_prev)
// If the reconstruction refuses the leaf (guard word set), the slabs are garbage; the Householder-TSQR fallback then forms V and
// the product is redone from V with (U'^-1, B) := (I, 0), so the reduce launch is the same either way.
// ---------------------------------------------------------------------------------------------------------
// one (32-column tile, K-slice) item of the product on the first 256 threads of the workgroup: the guard routes' redo
__device__ __forceinline__ void ep_product_item(const EpArgs& e, int item, const double* __restrict__ Qp)
{
    extern __shared__ __attribute__((aligned(16))) double ep_smem[];
    double* As = ep_smem;
    double* Bsm = ep_smem + 2 * 32 * LDKF;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wi = wave & 1, wj = wave >> 1, l15 = lane & 15, l4 = lane >> 4;
    const int tile = item % e.tiles, z = item / e.tiles;
    const int j0 = tile * 32;
    const int kbeg = z * e.kchunk, kend = min(e.K, kbeg + e.kchunk);
    const bool second = j0 >= e.N1;
    const double* B = second ? e.B2 + (size_t) (j0 - e.N1) * e.ldb2 : e.B1 + (size_t) j0 * e.ldb1;
    const int ldb = second ? e.ldb2 : e.ldb1;
    v4d acc[1][1];
    acc[0][0] = (v4d){0.0, 0.0, 0.0, 0.0};
    gemm_kloop<1, 1, false, true>(acc, Qp, e.ldq, B, ldb, 0, 0, 32, 32, kbeg, kend, As, Bsm, tid, wi, wj, l15, l4);
    gemm_epilogue<1, 1, true>(acc, e.slabs + (size_t) z * e.slab_stride, 32, 32, e.N1 + e.N2, 0, j0, 1.0, 0.0, wi, wj, l15, l4);
}

#define EP_SMEM_BYTES (sizeof(double) * 4 * 32 * LDKF)
template <bool UINV, int TJ>
__global__ __launch_bounds__(64 * H3G, 2) void hr3_ep_kernel(const double* __restrict__ G2s, int nslab, const double* __restrict__ R1,
                                                         double* __restrict__ Vw, int ldv, double* __restrict__ A, int lda,
                                                         double* __restrict__ tau, double* __restrict__ T, int ldt,
                                                         double* __restrict__ Uout, int* __restrict__ guard, unsigned* __restrict__ bar,
                                                         double* __restrict__ epw, EpArgs e)
{
    if (blockIdx.x == 0) {
        hr3_body<UINV, true>(G2s, nslab, R1, Vw, ldv, A, lda, tau, T, ldt, Uout, guard, bar, epw);
        return;
    }
    if (threadIdx.x >= 256) return;          // the product tile is a 4-wave job (whole waves leave: the barriers below count the rest)
    if (*guard != 0) return;                 // pass 1 refused the leaf: there is no Q
    // K slices are dealt to the XCDs (workgroup b runs on XCD b % 8; all column tiles of a slice share its 32-column slice of Q in
    // that XCD's L2, see gemm_tn_dual_kernel) -- to XCDs 1..7 only: XCD 0 has the reconstruction, and on a CU-masked panel stream
    // (4 compute units per XCD, one workgroup of this launch per CU) a product workgroup there would queue behind it
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
    if (xcd == 0 || slot == 0) return;
    const int z = xcd - 1 + 7 * ((slot - 1) / e.ftiles);
    if (z >= e.ksplit) return;
    ep_fused_item<TJ>(e, (slot - 1) % e.ftiles, z);
}

// the fold matrices of the fallback: slabs recomputed from the final V, so y = z:  U'^-1 := I, B := 0
__device__ __forceinline__ void ep_fallback_folds(double* __restrict__ epw)
{
    for (int el = threadIdx.x; el < PW * PW; el += blockDim.x) {
        epw[1 * PW * PW + el] = (el % PW == el / PW) ? 1.0 : 0.0;
        epw[2 * PW * PW + el] = 0.0;
    }
}

// fallback launch (leaves whose guard route runs as separate launches): no-op unless the guard tripped; then the product
// again, from V (complete in Vw by now), and the fallback's fold matrices
__global__ __launch_bounds__(256, 2) void ep_redo_kernel(const int* __restrict__ guard, double* __restrict__ epw, EpArgs e)
{
    if (*guard == 0) return;
    if (blockIdx.x == 0) { ep_fallback_folds(epw); return; }
    ep_product_item(e, (int) blockIdx.x - 1, e.Q);
}

// W (32 x N1, ldw) and the Gram block G2 (N2 x 32, ldg) from the product slabs, T, and the factors in epw (see hr3_ep_kernel).
// One 256-thread workgroup per column j of Z: thread (i = tid & 31, zp = tid >> 5); the slab index is spread over the 8 groups zp,
// every sum is formed in a fixed order.  The three folds are 32-term products: 4 terms per thread, 8 partials per output.
__global__ __launch_bounds__(256) void slab_reduce_ep_kernel(int N1, int N2, int nslab, const double* __restrict__ slabs, size_t slab_stride,
                                                             const double* __restrict__ epw, const double* __restrict__ T, int ldt,
                                                             const double* __restrict__ B1, int ldb1,
                                                             const double* __restrict__ B2, int ldb2, double* __restrict__ W, int ldw,
                                                             double* __restrict__ G2, int ldg)
{
    __shared__ double red[256];
    __shared__ double vec[PW], top[PW];
    const int j = blockIdx.x, tid = threadIdx.x, i = tid & 31, zp = tid >> 5;
    const bool second = j >= N1;
    // the factor entries this thread needs, issued before the slab walk (they come from L2 under it):
    //   fb: B(r = zp + 8q, k = i)     fu: U'^-1(k = zp + 8q, c = i)     ft: T(c = zp + 8q, i)
    double fb[4], fu[4], ft[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int r = zp + 8 * q;
        fb[q] = epw[2 * PW * PW + i * PW + r];
        fu[q] = epw[1 * PW * PW + i * PW + r];
        ft[q] = (!second && r <= i) ? T[(size_t) i * ldt + r] : 0.0;
    }
    if (tid < PW) top[tid] = second ? B2[(size_t) (j - N1) * ldb2 + tid] : B1[(size_t) j * ldb1 + tid];
    {
        const double* p = slabs + (size_t) j * 32 + i;
        double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;
        int z = zp;
        for (; z + 24 < nslab; z += 32) {
            s0 += p[(size_t) z * slab_stride];
            s1 += p[(size_t) (z + 8) * slab_stride];
            s2 += p[(size_t) (z + 16) * slab_stride];
            s3 += p[(size_t) (z + 24) * slab_stride];
        }
        for (; z < nslab; z += 8) s0 += p[(size_t) z * slab_stride];
        red[tid] = (s0 + s1) + (s2 + s3);
    }
    __syncthreads();
    // z(k = i) - sum_r B(r, k) top(r)
    double part = 0.0;
#pragma unroll
    for (int q = 0; q < 4; ++q) part -= fb[q] * top[zp + 8 * q];
    if (zp == 0) {
        double t = 0.0;
#pragma unroll
        for (int g = 0; g < 8; ++g) t += red[g * 32 + i];
        part += t;
    }
    __syncthreads();
    red[tid] = part;
    __syncthreads();
    if (tid < PW) {
        double t = 0.0;
#pragma unroll
        for (int g = 0; g < 8; ++g) t += red[g * 32 + tid];
        vec[tid] = t;
    }
    __syncthreads();
    // y(c = i) = sum_k U'^-1(k, c) vec(k)
    part = 0.0;
#pragma unroll
    for (int q = 0; q < 4; ++q) part += fu[q] * vec[zp + 8 * q];
    red[tid] = part;
    __syncthreads();
    double y = 0.0;
    if (tid < PW) {
#pragma unroll
        for (int g = 0; g < 8; ++g) y += red[g * 32 + tid];
        if (second) { G2[(size_t) tid * ldg + (j - N1)] = y; }
        else vec[tid] = y;
    }
    if (second) return;
    __syncthreads();
    // W(i, j) = sum_c T(c, i) y(c)
    part = 0.0;
#pragma unroll
    for (int q = 0; q < 4; ++q) part += ft[q] * vec[zp + 8 * q];
    __syncthreads();
    red[tid] = part;
    __syncthreads();
    if (tid < PW) {
        double t = 0.0;
#pragma unroll
        for (int g = 0; g < 8; ++g) t += red[g * 32 + tid];
        W[(size_t) j * ldw + tid] = t;
    }
}

// rows >= w of V: v U' = q by forward substitution over the columns (U' upper triangular), one row per thread
// final3_load: this thread's row of Q into a[], U' into LDS (ends with a workgroup barrier); final3_finish: solve and store.
// (Issuing these loads before the guard word is tested was tried: the fused kernel then spills -- 100 B of scratch -- and
// 8192^2 ran 3 % slower.)
template <bool FULL>
__device__ __forceinline__ void final3_load(double (&a)[PW], double (*Usm)[PW + 1], double* uinv, const double* __restrict__ Vw, int ldv,
                                            int mk, int w, const double* __restrict__ Um)
{
    const int tid = threadIdx.x, r = blockIdx.x * PT + tid;
    {
        const double* p = Vw + min(r, mk - 1);
#pragma unroll
        for (int c = 0; c < PW; ++c) a[c] = (FULL || c < w) ? p[(size_t) c * ldv] : 0.0;
    }
    for (int el = tid; el < PW * PW; el += PT) {
        const int i = el % PW, c = el / PW;
        const double u = (i < w && c < w && i <= c) ? Um[c * PW + i] : (i == c ? 1.0 : 0.0);
        Usm[i][c] = u;
        if (i == c) uinv[i] = 1.0 / u;
    }
    __syncthreads();
}

template <bool FULL>
__device__ __forceinline__ void final3_finish(double (&a)[PW], double (*Usm)[PW + 1], const double* uinv, double* __restrict__ Vw, int ldv,
                                              int mk, int w, double* __restrict__ A, int lda, const double* __restrict__ L1s = nullptr)
{
    const int tid = threadIdx.x, r = blockIdx.x * PT + tid;
    if (r >= mk) return;
    if (r < w) {                              // early-product leaf: the unit-lower top block of V comes from L1s (see final4_body)
        if (L1s) {
#pragma unroll
            for (int c = 0; c < PW; ++c) Vw[(size_t) c * ldv + r] = L1s[c * PW + r];
        }
        return;
    }
    double v[PW];
#pragma unroll
    for (int k = 0; k < PW; ++k) {
        v[k] = a[k] * uinv[k];
#pragma unroll
        for (int c = k + 1; c < PW; ++c) a[c] -= v[k] * Usm[k][c];
        __builtin_amdgcn_sched_barrier(0);
    }
#pragma unroll
    for (int c = 0; c < PW; ++c)
        if (FULL || c < w) { Vw[(size_t) c * ldv + r] = v[c]; A[(size_t) c * lda + r] = v[c]; }
}

template <bool FULL>
__global__ __launch_bounds__(PT) void final3_kernel(double* __restrict__ Vw, int ldv, int mk, int w, const double* __restrict__ Um,
                                                    double* __restrict__ A, int lda, const int* __restrict__ guard,
                                                    const double* __restrict__ L1s)
{
    __shared__ double Usm[PW][PW + 1];
    __shared__ double uinv[PW];
    if (*guard != 0) return;
    double a[PW];
    final3_load<FULL>(a, Usm, uinv, Vw, ldv, mk, w, Um);
    final3_finish<FULL>(a, Usm, uinv, Vw, ldv, mk, w, A, lda, L1s);
}

// Short leaves (<= 16 row blocks): the last launch of the CholeskyQR2 leaf and the one-launch guard route share a grid
// (one 512-row block per workgroup), so they are ONE launch -- the guard word picks the body.  Saves the kernel boundary
// (~4.7 us per leaf, 3-4 % of an 8192^2 factorisation) that the guard route cost when it had nothing to do.
// ep != 0 (early-product leaf, see hr3_ep_kernel): on the guard route the product slabs are recomputed from the final V by the
// same workgroups behind one more grid barrier, and the fold matrices become (T, 0, I, 0)
__device__ __forceinline__ void coop_ep_redo(int nblk, unsigned* __restrict__ bar, double* __restrict__ epw, const EpArgs& e)
{
    coop_barrier(bar, 4u * (unsigned) nblk);                 // V is complete in every workgroup's rows
    if (blockIdx.x == 0) ep_fallback_folds(epw);
    if (threadIdx.x >= 256) return;
    for (int item = blockIdx.x; item < e.tiles * e.ksplit; item += nblk) ep_product_item(e, item, e.Q);
}

__global__ __launch_bounds__(PT) void final3_coop_kernel(const double* __restrict__ Um, int nblk, int halves, int mk, int w,
                                                         double* __restrict__ Vloc1, double* __restrict__ taus, double* __restrict__ Ts,
                                                         double* __restrict__ stack, double* __restrict__ Rt, double* __restrict__ Ctop,
                                                         double* __restrict__ Umat, double* A, int lda,
                                                         double* __restrict__ tau, double* __restrict__ T, int ldt,
                                                         double* __restrict__ Vw, int ldv, unsigned* __restrict__ bar,
                                                         const int* __restrict__ guard, int ep, double* __restrict__ epw, EpArgs e)
{
    __shared__ double Usm[PW][PW + 1];
    __shared__ double uinv[PW];
    if (*guard != 0) {
        tsqr_coop_body(A, lda, mk, w, nblk, halves, Vloc1, taus, Ts, stack, Rt, Ctop, Umat, A, lda, tau, T, ldt, Vw, ldv, bar);
        if (ep) coop_ep_redo(nblk, bar, epw, e);
        return;
    }
    double a[PW];
    final3_load<true>(a, Usm, uinv, Vw, ldv, mk, w, Um);
    final3_finish<true>(a, Usm, uinv, Vw, ldv, mk, w, A, lda, ep ? epw : nullptr);
}

// tall leaves: the last CholeskyQR2 launch (final3) and the one-launch tall guard route (tsqr_coop_tall_body) share a grid the same
// way; the first G workgroups are the guard route's participants
__global__ __launch_bounds__(PT) void final3_coop_tall_kernel(const double* __restrict__ Um, TallTree t, int G, int mk, int w,
                                                              double* __restrict__ Vloc1, double* __restrict__ Vup, double* __restrict__ taus,
                                                              double* __restrict__ Ts, double* __restrict__ stacks, double* __restrict__ Cup,
                                                              double* __restrict__ Rt, double* __restrict__ Umat, double* A, int lda,
                                                              double* __restrict__ tau, double* __restrict__ T, int ldt,
                                                              double* __restrict__ Vw, int ldv, unsigned* __restrict__ bar,
                                                              const int* __restrict__ guard, int ep, double* __restrict__ epw, EpArgs e)
{
    __shared__ double Usm[PW][PW + 1];
    __shared__ double uinv[PW];
    if (*guard != 0) {
        if ((int) blockIdx.x >= G) return;
        unsigned tgt = 0u;
        tsqr_coop_tall_body(t, G, A, lda, mk, w, Vloc1, Vup, taus, Ts, stacks, Cup, Rt, Umat, A, lda, tau, T, ldt, Vw, ldv, bar, tgt);
        if (ep) {
            tgt += (unsigned) G;
            coop_barrier(bar, tgt);                            // V is complete in every workgroup's rows
            if (blockIdx.x == 0) ep_fallback_folds(epw);
            if (threadIdx.x >= 256) return;
            for (int item = blockIdx.x; item < e.tiles * e.ksplit; item += G) ep_product_item(e, item, e.Q);
        }
        return;
    }
    double a[PW];
    final3_load<true>(a, Usm, uinv, Vw, ldv, mk, w, Um);
    final3_finish<true>(a, Usm, uinv, Vw, ldv, mk, w, A, lda, ep ? epw : nullptr);
}

// ... and on its own, behind a separate final3 launch: fused, final3 inherits this kernel's footprint (256 VGPRs, 140 KB of LDS: one
// workgroup per compute unit), which costs a 262144-row leaf's final3 more bandwidth (36 -> ~55 us) than the saved launch is worth;
// leaves whose final3 has no more workgroups than the stream has compute units lose nothing and take the fused form
__global__ __launch_bounds__(PT) void tsqr_coop_tall_kernel(TallTree t, int G, int mk, int w,
                                                            double* __restrict__ Vloc1, double* __restrict__ Vup, double* __restrict__ taus,
                                                            double* __restrict__ Ts, double* __restrict__ stacks, double* __restrict__ Cup,
                                                            double* __restrict__ Rt, double* __restrict__ Umat, double* A, int lda,
                                                            double* __restrict__ tau, double* __restrict__ T, int ldt,
                                                            double* __restrict__ Vw, int ldv, unsigned* __restrict__ bar,
                                                            const int* __restrict__ guard, int ep, double* __restrict__ epw, EpArgs e)
{
    if (*guard == 0) return;
    unsigned tgt = 0u;
    tsqr_coop_tall_body(t, G, A, lda, mk, w, Vloc1, Vup, taus, Ts, stacks, Cup, Rt, Umat, A, lda, tau, T, ldt, Vw, ldv, bar, tgt);
    if (ep) {
        tgt += (unsigned) G;
        coop_barrier(bar, tgt);
        if (blockIdx.x == 0) ep_fallback_folds(epw);
        if (threadIdx.x >= 256) return;
        for (int item = blockIdx.x; item < e.tiles * e.ksplit; item += G) ep_product_item(e, item, e.Q);
    }
}

// the same for the third-generation leaf (Um = U'^-1, final4_body)
__global__ __launch_bounds__(PT) void final4_coop_kernel(const double* __restrict__ Um, int nblk, int halves, int mk, int w,
                                                         double* __restrict__ Vloc1, double* __restrict__ taus, double* __restrict__ Ts,
                                                         double* __restrict__ stack, double* __restrict__ Rt, double* __restrict__ Ctop,
                                                         double* __restrict__ Umat, double* A, int lda,
                                                         double* __restrict__ tau, double* __restrict__ T, int ldt,
                                                         double* __restrict__ Vw, int ldv, unsigned* __restrict__ bar,
                                                         const int* __restrict__ guard, int rows_wg, int ep, double* __restrict__ epw, EpArgs e)
{
    if (*guard != 0) {
        // guard route: the first nblk workgroups (dispatched first), one 512-row block each; any others have nothing to do
        if ((int) blockIdx.x < nblk) {
            tsqr_coop_body(A, lda, mk, w, nblk, halves, Vloc1, taus, Ts, stack, Rt, Ctop, Umat, A, lda, tau, T, ldt, Vw, ldv, bar);
            if (ep) coop_ep_redo(nblk, bar, epw, e);
        }
        return;
    }
    final4_body(Vw, ldv, mk, Um, A, lda, rows_wg, ep ? epw : nullptr);
}

// ---------------------------------------------------------------------------------------------------------
// launchers: the workgroup is sized to the tallest block of the launch (64..512 threads) -- a 64-row top stack
// runs on one wave with no cross-wave reduction instead of eight mostly-idle ones
static int nt_for(int rows) { return rows <= 64 ? 64 : (rows <= 128 ? 128 : (rows <= 256 ? 256 : 512)); }

// 512- and 1024-row blocks on 256 threads, one wave per SIMD (not 512 threads, two per SIMD): per Householder step a SIMD then
// runs ONE transposing wave reduction instead of two and the cross-wave sum has 4 partials instead of 8
static int leaf_waves(void)
{
    static const int v = 4;
    return v;
}

#define LAUNCH_F(NT, RPT) hipLaunchKernelGGL((tsqr_factor_kernel<NT, RPT>), dim3(nblk), dim3(NT), 0, s, src, lds, rows_total, chunk, w, Vloc, ldv, tauloc, Tloc, Rstack, ldr, guard)
static void launch_factor(hipStream_t s, int nblk, int maxrows, const double* src, int lds, int rows_total, int chunk, int w,
                          double* Vloc, int ldv, double* tauloc, double* Tloc, double* Rstack, int ldr, const int* guard)
{
    const bool w4 = leaf_waves() == 4;
    if (maxrows > PT) {      // 1024-row blocks, one tree level less for tall leaves
        LAUNCH_F(512, 2);        // (256 threads x 4 rows needs 360 VGPRs and measured slower)
        return;
    }
    switch (nt_for(maxrows)) {
    case 64: LAUNCH_F(64, 1); break;
    case 128: LAUNCH_F(128, 1); break;
    case 256: LAUNCH_F(256, 1); break;
    default: if (w4) LAUNCH_F(256, 2); else LAUNCH_F(512, 1); break;
    }
}
#undef LAUNCH_F

#define LAUNCH_T(NT, RPT) hipLaunchKernelGGL((tsqr_top_kernel<NT, RPT>), dim3(1), dim3(NT), 0, s, stack, lds, rows, w, Rt, Cout, ldc, guard)
static void launch_top(hipStream_t s, const double* stack, int lds, int rows, int w, double* Rt, double* Cout, int ldc, const int* guard)
{
    switch (nt_for(rows)) {
    case 64: LAUNCH_T(64, 1); break;
    case 128: LAUNCH_T(128, 1); break;
    case 256: LAUNCH_T(256, 1); break;
    default: LAUNCH_T(512, 1); break;      // (256 threads x 2 rows measured 6% slower here, unlike in the factor kernel)
    }
}
#undef LAUNCH_T


static void launch_apply(hipStream_t s, int nblk, int maxrows, const double* Vloc, int ldv, const double* tauloc, const double* Tloc,
                         int rows_total, int chunk, int w, const double* Cin, int ldci, double* Cout, int ldco, const int* guard)
{
    switch (nt_for(maxrows)) {
    case 64: hipLaunchKernelGGL(tsqr_apply_kernel<64>, dim3(nblk), dim3(64), 0, s, Vloc, ldv, tauloc, Tloc, rows_total, chunk, w, Cin, ldci, Cout, ldco, guard); break;
    case 128: hipLaunchKernelGGL(tsqr_apply_kernel<128>, dim3(nblk), dim3(128), 0, s, Vloc, ldv, tauloc, Tloc, rows_total, chunk, w, Cin, ldci, Cout, ldco, guard); break;
    case 256: hipLaunchKernelGGL(tsqr_apply_kernel<256>, dim3(nblk), dim3(256), 0, s, Vloc, ldv, tauloc, Tloc, rows_total, chunk, w, Cin, ldci, Cout, ldco, guard); break;
    default: hipLaunchKernelGGL(tsqr_apply_kernel<512>, dim3(nblk), dim3(512), 0, s, Vloc, ldv, tauloc, Tloc, rows_total, chunk, w, Cin, ldci, Cout, ldco, guard); break;
    }
}

extern "C" {

// workspace (doubles) for leaves of up to m rows
size_t qrd_panel_ws_size(int m)
{
    const size_t nblk1 = (size_t) (m + PT - 1) / PT;
    const size_t up = 2 * nblk1 * PW + 4 * PT;                 // all upper-level stacks together (geometric)
    return 2 * (size_t) m * PW      /* Vloc1, Q1 */
         + 3 * up * PW              /* stacks, upper Vloc, upper C */
         + (nblk1 + up / PW + 64) * (PW + PW * PW)   /* tau and T per block, all levels */
         + 2 * PW * PW + PW + 64;   /* Rt, Umat + reciprocal diagonal */
}

// MI355XQR_COOP=0: the guard route always as separate launches (default 1: one cooperative launch for short leaves)
static int coop_enabled(void)
{
    static const int v = QRD_LAB_ENV_INT("MI355XQR_COOP", 1) != 0;
    return v;
}

// ep / epw: early-product leaf (hr3_ep_kernel has run): the last CholeskyQR2 launch also restores the top block of Vw from epw,
// and the guard route ends with the product's redo (inside the one-launch form, as ep_redo_kernel otherwise)
static int panel_tsqr_impl(void* stream, double* P, int ld, int mk, int w, double* tau, double* T, int ldt, double* Vw, int ldv,
                           double* ws, int m_cap, const int* guard, unsigned* bar = nullptr, const double* final3_u = nullptr,
                           int gen = 2, const EpArgs* ep = nullptr, double* epw = nullptr)
{
    hipStream_t s = (hipStream_t) stream;
    if (w < 1 || w > PW || mk < w || mk > m_cap) return -4;
    if (mk <= PT) {
        hipLaunchKernelGGL(panel_single_kernel, dim3(1), dim3(PT), 0, s, P, ld, mk, w, tau, T, ldt, Vw, ldv);
        return (int) hipGetLastError();
    }
    // carve the workspace
    const size_t nblk1cap = (size_t) (m_cap + PT - 1) / PT;
    const size_t up = 2 * nblk1cap * PW + 4 * PT;
    double* Vloc1 = ws;                        // ld = mk
    double* Q1 = Vloc1 + (size_t) m_cap * PW;  // ld = mk
    double* stacks = Q1 + (size_t) m_cap * PW;
    double* Vup = stacks + up * PW;
    double* Cup = Vup + up * PW;
    double* taus = Cup + up * PW;
    double* Ts = taus + (nblk1cap + up / PW + 64) * PW;
    double* Rt = Ts + (nblk1cap + up / PW + 64) * PW * PW;
    double* Umat = Rt + PW * PW;

    // ---- up the tree
    const int MAXL = 8;
    int lv_rows[MAXL], lv_nblk[MAXL], lv_chunk[MAXL];
    size_t lv_off[MAXL], lv_tau[MAXL];        // offsets of level l's input stack / Vloc / C (l >= 1) and tau
    int L = 0;
    // level-0 blocks: 512 rows (one row per thread) up to 8192 rows -- 16 blocks, a two-level tree --, 1024 rows (two rows
    // per thread) above, which keeps the tree at two levels up to 16384 rows and three up to 262144
    const int brows0 = (mk <= 16 * PT) ? PT : 2 * PT;
    lv_rows[0] = mk; lv_chunk[0] = 0; lv_nblk[0] = (mk + brows0 - 1) / brows0; lv_off[0] = 0; lv_tau[0] = 0;
    // final3_u: the caller's last CholeskyQR2 launch (final3_kernel<true> with this U') is still to be issued -- fused with the
    // one-launch guard route where that exists (same grid), on its own otherwise
    static const int fuse = 1;
    const bool coop = guard && bar && coop_enabled() && lv_nblk[0] * w <= PT;
    const EpArgs ep0 = ep ? *ep : EpArgs{};
    const size_t ep_shm = ep ? EP_SMEM_BYTES : 0;
    // the product's redo for guard routes that run as separate launches: one more launch that returns at once unless the guard tripped
    auto redo = [&]() {
        if (ep)
            hipLaunchKernelGGL(ep_redo_kernel, dim3(1 + ep->tiles * ep->ksplit), dim3(256), EP_SMEM_BYTES, s, guard, epw, ep0);
        return (int) hipGetLastError();
    };
    if (final3_u && coop && fuse && brows0 == PT && w == PW) {
        if (gen == 3) {
            static const int half_wg = 1;
            const int rows_wg = half_wg ? PT / 2 : PT;
            hipLaunchKernelGGL(final4_coop_kernel, dim3(half_wg ? (mk + rows_wg - 1) / rows_wg : lv_nblk[0]), dim3(PT), ep_shm, s, final3_u, lv_nblk[0],
                               brows0 / PT, mk, w, Vloc1, taus, Ts, stacks, Rt, Cup, Umat, P, ld, tau, T, ldt, Vw, ldv, bar, guard, rows_wg,
                               ep ? 1 : 0, epw, ep0);
        }
        else
            hipLaunchKernelGGL(final3_coop_kernel, dim3(lv_nblk[0]), dim3(PT), ep_shm, s, final3_u, lv_nblk[0], brows0 / PT, mk, w, Vloc1, taus, Ts,
                               stacks, Rt, Cup, Umat, P, ld, tau, T, ldt, Vw, ldv, bar, guard, ep ? 1 : 0, epw, ep0);
        return (int) hipGetLastError();
    }
    // tall leaf behind a CholeskyQR2 attempt (second-generation kernels): final3 and the whole multi-level guard route in one launch
    static const int tall_coop = 1;
    if (final3_u && gen == 2 && guard && bar && !coop && coop_enabled() && tall_coop && fuse && w == PW && brows0 == 2 * PT) {
        TallTree t{};
        t.nblk0 = lv_nblk[0]; t.halves = brows0 / PT;
        long long off = 0, toff = (long long) lv_nblk[0] * PW;
        int cur_rows = lv_nblk[0] * w, Lv = 0;
        const int gchunk = (PT / w) * w;
        bool fits = true;
        while (cur_rows > PT) {
            if (Lv + 2 >= TALL_MAXL) { fits = false; break; }
            ++Lv;
            t.rows[Lv] = cur_rows; t.chunk[Lv] = gchunk; t.nblk[Lv] = (cur_rows + gchunk - 1) / gchunk;
            t.off[Lv] = off; t.tau[Lv] = toff;
            off += (long long) cur_rows * PW; toff += (long long) t.nblk[Lv] * PW;
            cur_rows = t.nblk[Lv] * w;
        }
        if (fits) {
            t.L = Lv; t.top_off = off; t.top_rows = cur_rows;
            // participants: the launch is a no-op nearly always, and what a no-op costs is dispatching its workgroups (140 KB of LDS
            // each) -- 64 are gone in ~5 us; on the guard route they walk their blocks grid-stride
            static const int gcap = 64;
            int G = qrd_stream_cus_coresident(stream);
            if (G > gcap) G = gcap;
            if (G > t.nblk0) G = t.nblk0;
            if (G < 1) G = 1;
            const int fblocks = (mk + PT - 1) / PT;
            // (final3 as its own launch measured equal within noise at 65536 / 131072 rows: 1.04 / 1.46 ms against 1.03 / 1.43 fused)
            static const int fuse_tall = 1;
            if (fuse_tall && fblocks <= qrd_stream_cus_coresident(stream))
                hipLaunchKernelGGL(final3_coop_tall_kernel, dim3(fblocks > G ? fblocks : G), dim3(PT), ep_shm, s, final3_u, t, G, mk, w, Vloc1, Vup,
                                   taus, Ts, stacks, Cup, Rt, Umat, P, ld, tau, T, ldt, Vw, ldv, bar, guard, ep ? 1 : 0, epw, ep0);
            else {
                hipLaunchKernelGGL(final3_kernel<true>, dim3(fblocks), dim3(PT), 0, s, Vw, ldv, mk, w, final3_u, P, ld, guard, ep ? epw : nullptr);
                hipLaunchKernelGGL(tsqr_coop_tall_kernel, dim3(G), dim3(PT), ep_shm, s, t, G, mk, w, Vloc1, Vup, taus, Ts, stacks, Cup, Rt, Umat,
                                   P, ld, tau, T, ldt, Vw, ldv, bar, guard, ep ? 1 : 0, epw, ep0);
            }
            return (int) hipGetLastError();
        }
    }
    if (final3_u && gen == 3)
        hipLaunchKernelGGL(final4_kernel, dim3((mk + PT - 1) / PT), dim3(PT), 0, s, Vw, ldv, mk, final3_u, P, ld, guard, ep ? epw : nullptr);
    else if (final3_u)
        hipLaunchKernelGGL(final3_kernel<true>, dim3((mk + PT - 1) / PT), dim3(PT), 0, s, Vw, ldv, mk, w, final3_u, P, ld, guard, ep ? epw : nullptr);
    if (coop) {
        // short leaf behind a CholeskyQR2 attempt: the whole guard route in one launch (grid barriers inside)
        hipLaunchKernelGGL(tsqr_coop_kernel, dim3(lv_nblk[0]), dim3(PT), 0, s, P, ld, mk, w, lv_nblk[0], brows0 / PT, Vloc1, taus, Ts, stacks,
                           Rt, Cup, Umat, P, ld, tau, T, ldt, Vw, ldv, bar, guard);
        return redo();
    }
    size_t off = 0, toff = (size_t) lv_nblk[0] * PW;
    launch_factor(s, lv_nblk[0], (mk + lv_nblk[0] - 1) / lv_nblk[0], P, ld, mk, 0, w, Vloc1, mk, taus, Ts, stacks, lv_nblk[0] * w, guard);
    int cur_rows = lv_nblk[0] * w;
    const int gchunk = (PT / w) * w;
    while (cur_rows > PT) {
        if (L + 2 >= MAXL) return -6;
        ++L;
        lv_rows[L] = cur_rows; lv_chunk[L] = gchunk; lv_nblk[L] = (cur_rows + gchunk - 1) / gchunk;
        lv_off[L] = off; lv_tau[L] = toff;
        const size_t next_off = off + (size_t) cur_rows * PW;
        launch_factor(s, lv_nblk[L], cur_rows < gchunk ? cur_rows : gchunk, stacks + off, cur_rows, cur_rows, gchunk, w,
                      Vup + off, cur_rows, taus + toff, Ts + toff * PW, stacks + next_off, lv_nblk[L] * w, guard);
        off = next_off; toff += (size_t) lv_nblk[L] * PW;
        cur_rows = lv_nblk[L] * w;
    }
    // ---- top: factor + explicit Q of the last stack; its output is the coefficient input of the level below
    launch_top(s, stacks + off, cur_rows, cur_rows, w, Rt, Cup + off, cur_rows, guard);
    // ---- down the tree
    const double* Cin = Cup + off;
    int ldci = cur_rows;
    for (int l = L; l >= 1; --l) {
        launch_apply(s, lv_nblk[l], lv_rows[l] < lv_chunk[l] ? lv_rows[l] : lv_chunk[l], Vup + lv_off[l], lv_rows[l], taus + lv_tau[l],
                     Ts + lv_tau[l] * PW, lv_rows[l], lv_chunk[l], w, Cin, ldci, Cup + lv_off[l], lv_rows[l], guard);
        Cin = Cup + lv_off[l];
        ldci = lv_rows[l];
    }
    // ---- Householder reconstruction on the top block, then every level-1 block writes its rows of V directly
    hipLaunchKernelGGL(hr_top_kernel, dim3(1), dim3(1024), 0, s, Vloc1, mk, taus, Ts, Cin, ldci, Rt, w, P, ld, tau, T, ldt, Vw,
                       ldv, Umat, guard);
    const int halves = brows0 / PT;
    hipLaunchKernelGGL(tsqr_final_kernel, dim3(lv_nblk[0] * halves), dim3(PT), 0, s, Vloc1, mk, taus, Ts, mk, lv_nblk[0], halves, w,
                       Cin, ldci, Umat, P, ld, Vw, ldv, guard);
    (void) Q1;
    return redo();
}

int qrd_panel_tsqr(void* stream, double* P, int ld, int mk, int w, double* tau, double* T, int ldt, double* Vw, int ldv,
                   double* ws, int m_cap)
{
    return panel_tsqr_impl(stream, P, ld, mk, w, tau, T, ldt, Vw, ldv, ws, m_cap, nullptr);
}

// Leaf = CholeskyQR2 + Householder reconstruction, guarded: 7 short launches, then the Householder-TSQR leaf above as
// launches that return at once unless the guard word says the Cholesky route was refused for this leaf.
// cws: QRD_CHOLQR_WS doubles (G1, G2, R1, M, guard word).
// MI355XQR_LEAF=2 selects the second-generation launch sequence (4 launches, row solves on
// the vector ALUs), default 3 (4 launches, row solves as matrix-core products with explicit triangular inverses; short leaves)
#ifdef QRD_STAMPS
int qrd_dbg_read_stamps(long long* out)
{
    return (int) hipMemcpyFromSymbol(out, HIP_SYMBOL(qrd_dbg_stamps), sizeof(long long) * 64, 0, hipMemcpyDeviceToHost);
}
#endif

int qrd_panel_tsqr_init(void)
{
    int rc = 0;
    rc |= (int) hipFuncSetAttribute(reinterpret_cast<const void*>(cholq3_kernel<512, 256>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                    (int) (CQ2_LDS_DOUBLES(256) * sizeof(double)));
    rc |= (int) hipFuncSetAttribute(reinterpret_cast<const void*>(cholq4_tall_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                    (int) (8 * PW * PW * sizeof(double)));
    rc |= (int) hipFuncSetAttribute(reinterpret_cast<const void*>(cholq3_kernel<256, 128>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                    (int) (CQ2_LDS_DOUBLES(128) * sizeof(double)));
    // early-product launches: ~59 KB of static LDS (reconstruction) + the product's 18 KB tile images
    rc |= (int) hipFuncSetAttribute(reinterpret_cast<const void*>(hr3_ep_kernel<true, 1>), hipFuncAttributeMaxDynamicSharedMemorySize, (int) EP_FUSED_SMEM_BYTES(1));
    rc |= (int) hipFuncSetAttribute(reinterpret_cast<const void*>(hr3_ep_kernel<true, 2>), hipFuncAttributeMaxDynamicSharedMemorySize, (int) EP_FUSED_SMEM_BYTES(2));
    rc |= (int) hipFuncSetAttribute(reinterpret_cast<const void*>(hr3_ep_kernel<true, 4>), hipFuncAttributeMaxDynamicSharedMemorySize, (int) EP_FUSED_SMEM_BYTES(4));
    rc |= (int) hipFuncSetAttribute(reinterpret_cast<const void*>(hr3_ep_kernel<false, 1>), hipFuncAttributeMaxDynamicSharedMemorySize, (int) EP_FUSED_SMEM_BYTES(1));
    rc |= (int) hipFuncSetAttribute(reinterpret_cast<const void*>(hr3_ep_kernel<false, 2>), hipFuncAttributeMaxDynamicSharedMemorySize, (int) EP_FUSED_SMEM_BYTES(2));
    rc |= (int) hipFuncSetAttribute(reinterpret_cast<const void*>(hr3_ep_kernel<false, 4>), hipFuncAttributeMaxDynamicSharedMemorySize, (int) EP_FUSED_SMEM_BYTES(4));
    rc |= (int) hipFuncSetAttribute(reinterpret_cast<const void*>(final4_coop_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int) EP_SMEM_BYTES);
    rc |= (int) hipFuncSetAttribute(reinterpret_cast<const void*>(final3_coop_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int) EP_SMEM_BYTES);
    rc |= (int) hipFuncSetAttribute(reinterpret_cast<const void*>(final3_coop_tall_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int) EP_SMEM_BYTES);
    rc |= (int) hipFuncSetAttribute(reinterpret_cast<const void*>(tsqr_coop_tall_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int) EP_SMEM_BYTES);
    return rc;
}

// early-product request of the host (qrd_panel_cholqr_ep): where the product goes and what it needs
struct EpHost {
    int N1, N2;
    const double* B1; int ldb1;
    const double* B2; int ldb2;
    double* W; int ldw;
    double* G2; int ldg;
    double* slabs; size_t slab_cap;
    int done;                       // out: 1 = W and G2 are (will be, in stream order) complete
};

static inline bool ep_vec_ok(const void* p, int ld) { return ((reinterpret_cast<uintptr_t>(p) & 15) == 0) && (ld % 2 == 0); }

static int panel_cholqr_impl(void* stream, double* P, int ld, int mk, int w, double* tau, double* T, int ldt, double* Vw, int ldv,
                             double* ws, int m_cap, double* cws, double* slabs, size_t slab_cap, int gram_nslab, EpHost* eph)
{
    hipStream_t s = (hipStream_t) stream;
    if (w < 1 || w > PW || mk < w || mk > m_cap) return -4;
    // one workgroup covers the leaf / ragged last leaf (the per-column predicates a narrow leaf needs cost the
    // Cholesky kernels 3.5 KB of scratch per thread): Householder path directly
    if (mk <= PT || w < PW) return panel_tsqr_impl(stream, P, ld, mk, w, tau, T, ldt, Vw, ldv, ws, m_cap, nullptr);
    // early product: same shape rules as the separate product launch (32-column tiles, whole k-tiles, 16-byte aligned operands)
    EpArgs ep{};
    double* epw = cws + 5 * PW * PW;
    bool use_ep = false;
    int ep_tj = 1;
    if (eph) {
        const int N = eph->N1 + eph->N2;
        const size_t per = (size_t) 32 * N;
        // The fused launch has one 4-wave product workgroup per compute unit (~2 TB/s on a tall leaf) where the separate launch streams
        // at 3.3 TB/s; it wins while the product is shorter than that difference plus the reconstruction it hides (~28 us): up to
        // ~128 MB of operands (65536 x 256: 1.17 -> 1.11 ms; 262144 x 512: 6.96 -> 7.09, hence not there)
        static const long long ep_max_mb = 128;
        if (N > 0 && 8LL * mk * (32 + N) <= ep_max_mb * 1000000LL &&
            eph->N1 % 32 == 0 && eph->N2 % 32 == 0 && mk % BK == 0 && mk >= BK && ep_vec_ok(Vw, ldv) &&
            (eph->N1 == 0 || ep_vec_ok(eph->B1, eph->ldb1)) && (eph->N2 == 0 || ep_vec_ok(eph->B2, eph->ldb2)) && eph->slabs &&
            eph->slab_cap >= per) {
            // Fused launch: ONE 4-wave product workgroup per compute unit (the launch is sized by the reconstruction), so: the widest
            // tile that still gives every CU of the stream a workgroup, and as many K slices as there are workgroup slots (one round),
            // each at least 4 k-tiles long; the slabs are summed by one reduce workgroup per column, 8 slabs at a time
            const int tiles = N / 32;
            int per_xcd = qrd_stream_cus(stream) / 8;            // one product workgroup per compute unit, XCDs 1..7
            if (per_xcd < 1) per_xcd = 1;
            long long kcap = (mk + 4 * BK - 1) / (4 * BK);
            if (kcap > 128) kcap = 128;
            if ((size_t) kcap * per > eph->slab_cap) kcap = (long long) (eph->slab_cap / per);
            if (kcap < 1) kcap = 1;
            ep_tj = N >= 128 ? 4 : (N >= 64 ? 2 : 1);
            while (ep_tj > 1 && ((N + 32 * ep_tj - 1) / (32 * ep_tj) > per_xcd ||
                                 (long long) ((N + 32 * ep_tj - 1) / (32 * ep_tj)) * kcap < (3LL * 7 * per_xcd) / 4)) ep_tj >>= 1;
            const int ftiles = (N + 32 * ep_tj - 1) / (32 * ep_tj);
            long long ks = 7LL * (per_xcd / ftiles > 0 ? per_xcd / ftiles : 1);      // slices per XCD * 7 XCDs: one round of workgroups
            if (ks > kcap) ks = kcap;
            int ksplit = (int) ks;
            int kchunk = ((mk + ksplit - 1) / ksplit + BK - 1) / BK * BK;
            ksplit = (mk + kchunk - 1) / kchunk;
            ep.N1 = eph->N1; ep.N2 = eph->N2; ep.K = mk; ep.kchunk = kchunk; ep.tiles = tiles; ep.ksplit = ksplit; ep.ftiles = ftiles;
            ep.Q = Vw; ep.ldq = ldv; ep.B1 = eph->B1; ep.ldb1 = eph->ldb1; ep.B2 = eph->B2; ep.ldb2 = eph->ldb2;
            ep.slabs = eph->slabs; ep.slab_stride = per;
            use_ep = true;
        }
    }
    double *G1 = cws, *G2 = cws + PW * PW, *R1 = cws + 2 * PW * PW, *Mm = cws + 3 * PW * PW;
    int* guard = (int*) (cws + 4 * PW * PW);
    unsigned* bar = (unsigned*) (guard + 2);             // zeroed by hr3_kernel: second-generation leaf only
    bool have_bar = false;
    const int nblk = (mk + PT - 1) / PT;
    const bool al = ((reinterpret_cast<uintptr_t>(P) & 15) == 0) && (ld % 2 == 0) && (mk % 2 == 0) &&
                    ((reinterpret_cast<uintptr_t>(Vw) & 15) == 0) && (ldv % 2 == 0);
    int rc;
    // third generation (matrix-core row solves) for short leaves only: on tall leaves, which stream 67 MB per pass, its operand
    // loads (4 columns x 128 B per instruction instead of 1 x 512 B) cost more than the solves did (262144 x 512: 7.9 vs 7.6 ms).
    // A tall leaf the streaming form cannot take (rows not a multiple of 4, fewer than 32 columns) goes to the Householder route.
    const int gen = nblk > CQ2_MAXSLAB ? 2 : 3;
    const bool tall_ok = (mk & 3) == 0 && w == PW;
    if (al && (gen == 3 || tall_ok) && slab_cap >= (size_t) (CQ2_MAXSLAB + 2 * nblk) * PW * PW) {
        double* slab2 = slabs + (size_t) CQ2_MAXSLAB * PW * PW;
        int nblk2 = nblk;                                   // workgroups of the Q pass = slabs of G2
        if (nblk <= CQ2_MAXSLAB) {
            // short leaf (square problems): at most CQ2_MAXSLAB Gram slabs of >= 128 rows, summed by their consumers
            static const int gslab_max = CQ2_MAXSLAB;
            int rows_per = ((mk + gslab_max - 1) / gslab_max + GKB - 1) / GKB * GKB;
            if (rows_per < 2 * GKB) rows_per = 2 * GKB;
            int nslab = (mk + rows_per - 1) / rows_per;
            if (gram_nslab > 0 && gram_nslab <= CQ2_MAXSLAB) nslab = gram_nslab;      // left by the previous leaf's in-panel update
            else hipLaunchKernelGGL(gram32_kernel, dim3(nslab), dim3(256), 0, s, P, ld, mk, rows_per, slabs);
            // 256-row workgroups where that keeps the partial Grams of G2 within CQ2_MAXSLAB (mk <= 8192, the chain-bound part of a
            // square factorisation): the kernel is bound by its 128 matrix-core instructions per wave -- at two waves per SIMD 7.4 us
            // of its 25 -- and half-size workgroups put them on twice the compute units
            static const int half_wg = 1;
            if (half_wg && (mk + 255) / 256 <= CQ2_MAXSLAB) {
                nblk2 = (mk + 255) / 256;
                hipLaunchKernelGGL((cholq3_kernel<256, 128>), dim3(nblk2), dim3(256), CQ2_LDS_DOUBLES(128) * sizeof(double), s, P, ld, mk,
                                   slabs, nslab, R1, Vw, ldv, slab2, guard);
            } else
                hipLaunchKernelGGL((cholq3_kernel<512, 256>), dim3(nblk), dim3(512), CQ2_LDS_DOUBLES(256) * sizeof(double), s, P, ld, mk,
                                   slabs, nslab, R1, Vw, ldv, slab2, guard);
        } else {
            // tall leaf: the Gram pass needs hundreds of workgroups to reach HBM bandwidth, so its slabs go through a reduce launch;
            // then the streaming form: one-wave Cholesky (R1, R1^-1 -> Mm, which hr3 overwrites later) and the matrix-core pass
            static const int gq_max = 512;
            int gq = (mk + PT - 1) / PT;
            if (gq > gq_max) gq = gq_max;
            nblk2 = gq;
            if (gram_nslab > 0 && (size_t) gram_nslab * PW * PW <= slab_cap - (size_t) nblk2 * PW * PW)
                rc = qrd_slab_reduce(s, PW, PW, gram_nslab, slabs, PW, (size_t) PW * PW, G1, PW);   // partial Grams left by the previous leaf's update
            else
                rc = gram32(s, P, ld, mk, G1, slabs, slab_cap - (size_t) nblk2 * PW * PW);      // the tail of the buffer holds slab2
            if (rc) return rc;
            slab2 = slabs + (slab_cap - (size_t) nblk2 * PW * PW);
            hipLaunchKernelGGL(chol1_kernel, dim3(1), dim3(64), 0, s, G1, R1, Mm, guard);
            hipLaunchKernelGGL(cholq4_tall_kernel, dim3(gq), dim3(PT), 8 * PW * PW * sizeof(double), s, P, ld, mk, Mm, Vw, ldv, slab2, guard);
        }
        const double* g2src = slab2;
        int g2n = nblk2;
        if (nblk2 > 2 * CQ2_MAXSLAB) {   // tall leaf: hundreds of partial Grams are summed by a grid, not by the one reconstruction workgroup
            rc = qrd_slab_reduce(s, PW, PW, nblk2, slab2, PW, (size_t) PW * PW, G2, PW);
            if (rc) return rc;
            g2src = G2; g2n = 1;
        }
        if (use_ep) {
            // reconstruction (workgroup 0) and the leaf's long-K product on Q (the others) in ONE launch
            // 8 workgroups per "slot": slot 0 = the reconstruction (+ 7 that leave at once), then ftiles slots per group of 7 slices
            const dim3 grid(8 * (1 + ep.ftiles * ((ep.ksplit + 6) / 7)));
#define LAUNCH_HR3_EP(UI, TJ) hipLaunchKernelGGL((hr3_ep_kernel<UI, TJ>), grid, dim3(64 * H3G), EP_FUSED_SMEM_BYTES(TJ), s, g2src, g2n, R1, Vw, ldv, \
                                                 P, ld, tau, T, ldt, Mm, guard, bar, epw, ep)
            if (gen == 3) { if (ep_tj == 4) LAUNCH_HR3_EP(true, 4); else if (ep_tj == 2) LAUNCH_HR3_EP(true, 2); else LAUNCH_HR3_EP(true, 1); }
            else { if (ep_tj == 4) LAUNCH_HR3_EP(false, 4); else if (ep_tj == 2) LAUNCH_HR3_EP(false, 2); else LAUNCH_HR3_EP(false, 1); }
#undef LAUNCH_HR3_EP
        } else if (gen == 3)
            hipLaunchKernelGGL(hr3_kernel<true>, dim3(1), dim3(64 * H3G), 0, s, g2src, g2n, R1, Vw, ldv, P, ld, tau, T, ldt, Mm, guard, bar);
        else
            hipLaunchKernelGGL(hr3_kernel<false>, dim3(1), dim3(64 * H3G), 0, s, g2src, g2n, R1, Vw, ldv, P, ld, tau, T, ldt, Mm, guard, bar);
        have_bar = true;                                    // final3_kernel<true>: issued by panel_tsqr_impl, fused with the guard route where it can
    } else
        return panel_tsqr_impl(stream, P, ld, mk, w, tau, T, ldt, Vw, ldv, ws, m_cap, nullptr);     // unaligned operands: Householder route directly
    rc = (int) hipGetLastError();
    if (rc) return rc;
    rc = panel_tsqr_impl(stream, P, ld, mk, w, tau, T, ldt, Vw, ldv, ws, m_cap, guard, have_bar ? bar : nullptr, have_bar ? Mm : nullptr,
                         gen, use_ep ? &ep : nullptr, epw);
    if (rc || !use_ep) return rc;
    hipLaunchKernelGGL(slab_reduce_ep_kernel, dim3(ep.N1 + ep.N2), dim3(256), 0, s, ep.N1, ep.N2, ep.ksplit, ep.slabs, ep.slab_stride, epw,
                       T, ldt, ep.B1, ep.ldb1, ep.B2, ep.ldb2, eph->W, eph->ldw, eph->G2, eph->ldg);
    eph->done = 1;
    return (int) hipGetLastError();
}

int qrd_panel_cholqr(void* stream, double* P, int ld, int mk, int w, double* tau, double* T, int ldt, double* Vw, int ldv,
                     double* ws, int m_cap, double* cws, double* slabs, size_t slab_cap, int gram_nslab)
{
    return panel_cholqr_impl(stream, P, ld, mk, w, tau, T, ldt, Vw, ldv, ws, m_cap, cws, slabs, slab_cap, gram_nslab, nullptr);
}

// The leaf AND its in-panel products in one call ("early product", hr3_ep_kernel):  W (32 x N1, ldw) = T_l^T V_l^T B1 and
// G2 (N2 x 32, ldg) = B2^T V_l, with B1 = the rest of the panel (mk x N1) and B2 = the panel's earlier reflectors (mk x N2, rows from
// this leaf's top row).  ep_slabs: split-K workspace of the product, NOT aliasing `slabs`.  *did = 1: W and G2 are done (in stream
// order); *did = 0: the leaf is factored but the shapes did not fit -- the caller issues the products itself (qrd_gemm_tn_dual).
int qrd_panel_cholqr_ep(void* stream, double* P, int ld, int mk, int w, double* tau, double* T, int ldt, double* Vw, int ldv,
                        double* ws, int m_cap, double* cws, double* slabs, size_t slab_cap, int gram_nslab,
                        int N1, const double* B1, int ldb1, int N2, const double* B2, int ldb2, double* W, int ldw, double* G2, int ldg,
                        double* ep_slabs, size_t ep_slab_cap, int* did)
{
    EpHost h{N1, N2, B1, ldb1, B2, ldb2, W, ldw, G2, ldg, ep_slabs, ep_slab_cap, 0};
    const int rc = panel_cholqr_impl(stream, P, ld, mk, w, tau, T, ldt, Vw, ldv, ws, m_cap, cws, slabs, slab_cap, gram_nslab, &h);
    if (did) *did = h.done;
    return rc;
}

}   // extern "C"
