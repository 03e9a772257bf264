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
#include "qr_device.h"
#include "qr_common.h"

#define PW LEAFW          // max leaf width
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

// T (w x w upper triangular, into Tl[PW][PW+1] in LDS) of the block's reflectors from the captured Gram
// columns Z(q, j) = v_q^T v_j and tau:  T(0:j,j) = -tau_j T(0:j,0:j) Z(0:j,j).  Row p depends only on row p:
// thread tid < PW computes row tid with no synchronisation.
template <class SH>
__device__ __forceinline__ void build_t_rows(double (*Tl)[PW + 1], double (*Z)[PW + 1], const SH& sh, int w, int tid)
{
    if (tid < PW) {
        double trow[PW];
#pragma unroll
        for (int q = 0; q < PW; ++q) trow[q] = 0.0;
#pragma unroll
        for (int jj = 0; jj < PW; ++jj) {
            if (jj < w) {
                const double tj = sh.tau[jj];
                double sacc = 0.0;
#pragma unroll
                for (int q = 0; q < jj; ++q) sacc += trow[q] * Z[jj][q];
                trow[jj] = (tid == jj) ? tj : ((tid < jj) ? -tj * sacc : 0.0);
            }
        }
#pragma unroll
        for (int q = 0; q < PW; ++q) Tl[tid][q] = (tid < w && q < w) ? trow[q] : 0.0;
    }
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
__global__ __launch_bounds__(NT) void tsqr_factor_kernel(const double* __restrict__ src, int lds, int rows_total, int chunk,
                                                         int w, double* __restrict__ Vloc, int ldv,
                                                         double* __restrict__ tauloc, double* __restrict__ Tloc,
                                                         double* __restrict__ Rstack, int ldr, const int* __restrict__ guard)
{
    if (guard && *guard == 0) return;        // the CholeskyQR2 leaf succeeded: this fallback launch is a no-op
    __shared__ PanelSharedT<NT / 64> sh;
    __shared__ double Z[PW][PW + 1];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, b = blockIdx.x;
    int start, rows;
    block_range(rows_total, chunk, b, gridDim.x, start, rows);
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

// T: the last stack (rows <= 512): R~ -> Rt (ld PW), explicit Q_top [I;0] -> Cout (rows x w) in compact-WY form:
// Q_top [I;0] = [I;0] - V (T V1^T)
template <int NT, int RPT>
__global__ __launch_bounds__(NT) void tsqr_top_kernel(const double* __restrict__ stack, int lds, int rows, int w,
                                                      double* __restrict__ Rt, double* __restrict__ Cout, int ldc, const int* __restrict__ guard)
{
    if (guard && *guard == 0) return;        // the CholeskyQR2 leaf succeeded: this fallback launch is a no-op
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

// A: Cout(block rows, :) = Q_local_b [Cin_b ; 0] = [Cin_b; 0] - V_b (T_b V_b1^T Cin_b),
// Cin_b = rows [b*w, b*w+w) of the parent's output
template <int NT>
__global__ __launch_bounds__(NT) void tsqr_apply_kernel(const double* __restrict__ Vloc, int ldv,
                                                        const double* __restrict__ tauloc,
                                                        const double* __restrict__ Tloc, int rows_total, int chunk, int w,
                                                        const double* __restrict__ Cin, int ldci,
                                                        double* __restrict__ Cout, int ldco, const int* __restrict__ guard)
{
    if (guard && *guard == 0) return;        // the CholeskyQR2 leaf succeeded: this fallback launch is a no-op
    __shared__ double Zl[PW][PW + 1], V1[PW][PW + 1], Cl[PW][PW + 1], Wl[PW][PW + 1], Ml[PW][PW + 1];
    __shared__ double tl[PW];
    const int tid = threadIdx.x, b = blockIdx.x;
    int start, rows;
    block_range(rows_total, chunk, b, gridDim.x, start, rows);
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
__device__ __forceinline__ double readlane_f64(double v, int lane)
{
    return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), lane), __builtin_amdgcn_readlane(__double2loint(v), lane));
}

template <int I> struct HrStep {
    // back substitution step i (descending): row I of M is final; eliminate it from rows < I
    static __device__ __forceinline__ void msolve(double (&e)[HQ], int r, int w, const double* t0, double (*Zs)[PW + 1])
    {
        if (I < w) {
            const double ti = t0[I];
            const double z = (r < I) ? Zs[r][I] : 0.0;
#pragma unroll
            for (int q = 0; q < HQ; ++q) {
                e[q] = (r == I) ? e[q] * ti : e[q];
                e[q] -= z * readlane_f64(e[q], I);
            }
        }
        if constexpr (I > 0) HrStep<I - 1>::msolve(e, r, w, t0, Zs);
    }
    // Q1_top(r, :) -= V1(r, k) M(k, :), k ascending
    static __device__ __forceinline__ void q1top(double (&b)[HQ], const double (&e)[HQ], int r, int w, double (*V1)[PW + 1])
    {
        if (I < w) {
            const double v = (I <= r) ? V1[r][I] : 0.0;
#pragma unroll
            for (int q = 0; q < HQ; ++q) b[q] -= v * readlane_f64(e[q], I);
        }
        if constexpr (I + 1 < PW) HrStep<I + 1>::q1top(b, e, r, w, V1);
    }
    // modified LU step j = I (ascending); column j multipliers go through LDS (one barrier per step)
    static __device__ __forceinline__ void lu(double (&b)[HQ], int r, int g, int w, double (*colb)[PW], double* Ss)
    {
        if (I < w) {
            constexpr int pp = I & 1, qj = I / HG, gj = I % HG;
            if (g == gj && r < PW) colb[pp][r] = b[qj];
            __syncthreads();
            const double p = colb[pp][I];
            const double S = (p >= 0.0) ? -1.0 : 1.0;
            const double piv = p - S;
            const double l = (r > I && r < w) ? colb[pp][r] / piv : 0.0;
#pragma unroll
            for (int q = 0; q < HQ; ++q) {
                const int c = g + HG * q;
                const double u = readlane_f64(b[q], I);
                if (c > I) b[q] -= l * u;
            }
            if (g == gj) {
                b[qj] = (r > I) ? l : ((r == I) ? piv : b[qj]);
                if (r == I) Ss[I] = S;
            }
        }
        if constexpr (I + 1 < PW) HrStep<I + 1>::lu(b, r, g, w, colb, Ss);
    }
    // forward substitution L1 X = RHS, step k = I (ascending): row I of X is final
    static __device__ __forceinline__ void xsolve(double (&x)[HQ], int r, int w, double (*Bs)[PW + 1])
    {
        if (I < w) {
            const double l = (r > I && r < w) ? Bs[r][I] : 0.0;
#pragma unroll
            for (int q = 0; q < HQ; ++q) x[q] -= l * readlane_f64(x[q], I);
        }
        if constexpr (I + 1 < PW) HrStep<I + 1>::xsolve(x, r, w, Bs);
    }
    // back substitution U Uinv = I, step k = I (descending)
    static __device__ __forceinline__ void uinv(double (&ui)[HQ], int r, int w, double (*Bs)[PW + 1])
    {
        if (I < w) {
            const double d = 1.0 / Bs[I][I];
            const double u = (r < I) ? Bs[r][I] : 0.0;
#pragma unroll
            for (int q = 0; q < HQ; ++q) {
                ui[q] = (r == I) ? ui[q] * d : ui[q];
                ui[q] -= u * readlane_f64(ui[q], I);
            }
        }
        if constexpr (I > 0) HrStep<I - 1>::uinv(ui, r, w, Bs);
    }
};

__global__ __launch_bounds__(64 * HG) void hr_top_kernel(const double* __restrict__ Vloc, int ldvl, const double* __restrict__ tau0,
                                                     const double* __restrict__ Z0, const double* __restrict__ C0, int ldc0,
                                                     const double* __restrict__ Rt, int w, double* __restrict__ A, int lda,
                                                     double* __restrict__ tau, double* __restrict__ T, int ldt,
                                                     double* __restrict__ Vw, int ldv, double* __restrict__ Umat, const int* __restrict__ guard)
{
    if (guard && *guard == 0) return;        // the CholeskyQR2 leaf succeeded: this fallback launch is a no-op
    __shared__ double Bs[PW][PW + 1], V1[PW][PW + 1], Cs[PW][PW + 1];
    __shared__ double colb[2][PW];
    __shared__ double Ss[PW], t0[PW];
    const int lane = threadIdx.x & 63, g = threadIdx.x >> 6;
    const int r = lane;                          // lanes >= 32 carry zeros and write nothing
    const bool ra = r < w;
    for (int el = threadIdx.x; el < PW * PW; el += 64 * HG) {
        const int i = el % PW, c = el / PW;
        const bool in = (i < w && c < w);
        const double v = in ? Vloc[(size_t) c * ldvl + i] : 0.0;
        V1[i][c] = in ? ((c < i) ? v : (c == i ? 1.0 : 0.0)) : 0.0;
        Cs[i][c] = in ? C0[(size_t) c * ldc0 + i] : 0.0;
        Bs[i][c] = (in && i < c) ? Z0[c * PW + i] : 0.0;          // Z(i, c), i < c
    }
    if (threadIdx.x < PW) t0[threadIdx.x] = (threadIdx.x < w) ? tau0[threadIdx.x] : 0.0;
    __syncthreads();
    double e[HQ], b[HQ];
#pragma unroll
    for (int q = 0; q < HQ; ++q) {               // W = V1^T C0
        const int c = g + HG * q;
        double acc = 0.0;
        if (r < PW)
            for (int k = r; k < w; ++k) acc += V1[k][r] * Cs[k][c];
        e[q] = acc;
        b[q] = (ra && c < w) ? Cs[r][c] : 0.0;
    }
    HrStep<PW - 1>::msolve(e, r, w, t0, Bs);     // e := M_0 rows
    HrStep<0>::q1top(b, e, ra ? r : -1, w, V1);  // b := Q1_top rows (inactive lanes: no update)
    __syncthreads();                             // everyone is done with Bs as Z
    HrStep<0>::lu(b, r, g, w, colb, Ss);
    __syncthreads();
    if (r < PW) {
#pragma unroll
        for (int q = 0; q < HQ; ++q) Bs[r][g + HG * q] = b[q];
    }
    __syncthreads();
    double x[HQ], ui[HQ];
#pragma unroll
    for (int q = 0; q < HQ; ++q) {
        const int c = g + HG * q;
        x[q] = (ra && c <= r) ? -Ss[r] * Bs[c][r] : 0.0;          // -S U^T
        ui[q] = (ra && r == c) ? 1.0 : 0.0;
    }
    HrStep<0>::xsolve(x, r, w, Bs);
    HrStep<PW - 1>::uinv(ui, r, w, Bs);
    if (!ra) return;
#pragma unroll
    for (int q = 0; q < HQ; ++q) {
        const int c = g + HG * q;
        if (c < w) {
            T[(size_t) r * ldt + c] = x[q];                                        // T(c, r) = X(r, c)
            A[(size_t) c * lda + r] = (c >= r) ? Ss[r] * Rt[c * PW + r] : b[q];
            Vw[(size_t) c * ldv + r] = (c < r) ? b[q] : (c == r ? 1.0 : 0.0);
            Umat[c * PW + r] = (c >= r) ? ui[q] : 0.0;                             // Umat := U^-1 (ld PW)
            if (c == r) tau[r] = x[q];
        }
    }
}

// A1 + H2 fused (level 1, last launch of the leaf): V(rows, :) = Q1(rows, :) U^-1 with Q1 = [C_b; 0] - V_b M_b, i.e.
// V(r, :) = [C_b U^-1; 0](r, :) - V_b(r, :) (M_b U^-1), written straight into the panel and Vw for global rows >= w
// (the top w rows were written by hr_top_kernel).  No explicit Q1 round trip through memory.
__global__ __launch_bounds__(PT) void tsqr_final_kernel(const double* __restrict__ Vloc, int ldvl,
                                                        const double* __restrict__ tauloc, const double* __restrict__ Tloc,
                                                        int rows_total, int nblk, int halves, int w,
                                                        const double* __restrict__ Cin, int ldci,
                                                        const double* __restrict__ Umat, double* __restrict__ A, int lda,
                                                        double* __restrict__ Vw, int ldv, const int* __restrict__ guard)
{
    if (guard && *guard == 0) return;        // the CholeskyQR2 leaf succeeded: this fallback launch is a no-op
    __shared__ double Zl[PW][PW + 1], V1[PW][PW + 1], Cl[PW][PW + 1], Wl[PW][PW + 1], Ml[PW][PW + 1], Ui[PW][PW + 1];
    __shared__ double tl[PW];
    const int tid = threadIdx.x, b = blockIdx.x / halves, h = blockIdx.x % halves;
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
#define QRD_GUARD_THR (1.0 / 64.0)

// right-looking Cholesky G = R^T R on one wave: lane j (< 32) holds column j in g[]; on exit g[k] = R(k, j), k <= j
template <int K> struct CholStep {
    static __device__ __forceinline__ void run(double (&g)[PW], int lane, bool& ok)
    {
        const double p = readlane_f64(g[K], K);
        ok = ok && (p > 0.0);                       // false for NaN as well
        const double inv = 1.0 / sqrt(p);
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

// K2: every workgroup factors G1 itself (one wave, ~2 us) and turns its 512 rows of A into rows of Q = A R1^-1 (-> Vw)
// FULL = (w == 32): no per-column predicates.  (With `if (c < w)` around each store LLVM sinks the arithmetic of column c
// into that column's block, i.e. reorders the solve column-major, keeping all 270 LDS loads live: 3.5 KB of scratch.)
template <bool FULL>
__global__ __launch_bounds__(PT) void cholq_kernel(const double* __restrict__ P, int ld, int mk, int w,
                                                   const double* __restrict__ G1, double* __restrict__ R1,
                                                   double* __restrict__ Vw, int ldv, int* __restrict__ guard)
{
    __shared__ double Rs[PW][PW + 1];
    __shared__ double rinv[PW];
    __shared__ int okf;
    const int tid = threadIdx.x, lane = tid & 63, b = blockIdx.x;
    const int r = b * PT + tid;
    double a[PW];
    {
        const double* p = P + min(r, mk - 1);
#pragma unroll
        for (int c = 0; c < PW; ++c) a[c] = (FULL || c < w) ? p[(size_t) c * ld] : 0.0;      // issued before the factorisation
    }
    if (tid < 64) {
        const bool ok = chol_wave(G1, w, lane, Rs);
        if (lane == 0) okf = ok ? 1 : 0;
    }
    __syncthreads();
    if (tid < PW) rinv[tid] = 1.0 / Rs[tid][tid];
    const bool ok = okf != 0;
    if (b == 0) {
        if (tid == 0) *guard = ok ? 0 : 1;
        if (ok && tid < PW) {
#pragma unroll
            for (int k = 0; k < PW; ++k) R1[tid * PW + k] = Rs[k][tid];          // column tid
        }
    }
    if (!ok) return;
    __syncthreads();
    if (r >= mk) return;
    double q[PW];
#pragma unroll
    for (int k = 0; k < PW; ++k) {                                              // q R1 = a, column by column
        q[k] = a[k] * rinv[k];
#pragma unroll
        for (int c = k + 1; c < PW; ++c) a[c] -= q[k] * Rs[k][c];
        __builtin_amdgcn_sched_barrier(0);
    }
#pragma unroll
    for (int c = 0; c < PW; ++c)
        if (FULL || c < w) Vw[(size_t) c * ldv + r] = q[c];
}

// K4: second Cholesky pass + Householder reconstruction of the top block, one 16-wave workgroup (layout of hr_top_kernel:
// lane r < 32 of wave g owns row r of the columns c = g + 16 q)
__global__ __launch_bounds__(64 * HG) void hr2_kernel(const double* __restrict__ G2, const double* __restrict__ R1,
                                                  double* __restrict__ Vw, int ldv, int w, double* __restrict__ A, int lda,
                                                  double* __restrict__ tau, double* __restrict__ T, int ldt,
                                                  double* __restrict__ Mout, int* __restrict__ guard)
{
    __shared__ double Bs[PW][PW + 1], R1s[PW][PW + 1], R2s[PW][PW + 1], R2i[PW][PW + 1], Qs[PW][PW + 1];
    __shared__ double colb[2][PW];
    __shared__ double Ss[PW];
    __shared__ int flags[2];
    const int lane = threadIdx.x & 63, g = threadIdx.x >> 6;
    const int r = lane;
    const bool ra = r < w;
    if (*guard != 0) return;                               // pass 1 already refused
    if (threadIdx.x < 2) flags[threadIdx.x] = 0;
    __syncthreads();
    for (int el = threadIdx.x; el < PW * PW; el += 64 * HG) {
        const int i = el % PW, c = el / PW;
        const bool in = (i < w && c < w);
        R1s[i][c] = (in && i <= c) ? R1[c * PW + i] : 0.0;
        Qs[i][c] = in ? Vw[(size_t) c * ldv + i] : 0.0;
        if (in) {
            const double d = G2[(size_t) c * PW + i] - (i == c ? 1.0 : 0.0);
            if (!(fabs(d) <= QRD_GUARD_THR)) flags[0] = 1;  // also catches NaN
        }
    }
    __syncthreads();
    if (flags[0]) { if (threadIdx.x == 0) *guard = 1; return; }
    if (g == 0) {
        const bool ok = chol_wave(G2, w, lane, R2s);
        if (!ok && lane == 0) flags[1] = 1;
    }
    __syncthreads();
    if (flags[1]) { if (threadIdx.x == 0) *guard = 1; return; }
    if (g == 0) triu_inv_wave(R2s, lane, R2i);
    __syncthreads();
    double b[HQ], rt[HQ];
#pragma unroll
    for (int q = 0; q < HQ; ++q) {
        const int c = g + HG * q;
        double acc = 0.0, acr = 0.0;
        if (ra && c < w) {
            for (int k = 0; k <= c; ++k) acc += Qs[r][k] * R2i[k][c];             // Q1_top = Q_top R2^-1
            for (int k = r; k <= c; ++k) acr += R2s[r][k] * R1s[k][c];            // R~ = R2 R1
        }
        b[q] = acc; rt[q] = acr;
    }
    HrStep<0>::lu(b, r, g, w, colb, Ss);
    __syncthreads();
    if (r < PW) {
#pragma unroll
        for (int q = 0; q < HQ; ++q) Bs[r][g + HG * q] = b[q];
    }
    __syncthreads();
    double x[HQ], ui[HQ];
#pragma unroll
    for (int q = 0; q < HQ; ++q) {
        const int c = g + HG * q;
        x[q] = (ra && c <= r) ? -Ss[r] * Bs[c][r] : 0.0;          // -S U^T
        ui[q] = (ra && r == c) ? 1.0 : 0.0;
    }
    HrStep<0>::xsolve(x, r, w, Bs);
    HrStep<PW - 1>::uinv(ui, r, w, Bs);
    __syncthreads();                                              // everyone is done with Qs
    if (r < PW) {
#pragma unroll
        for (int q = 0; q < HQ; ++q) Qs[r][g + HG * q] = (ra && g + HG * q >= r) ? ui[q] : 0.0;     // Qs := U^-1
    }
    __syncthreads();
    if (!ra) return;
#pragma unroll
    for (int q = 0; q < HQ; ++q) {
        const int c = g + HG * q;
        if (c < w) {
            double m = 0.0;
            for (int k = r; k <= c; ++k) m += R2i[r][k] * Qs[k][c];            // M = R2^-1 U^-1 (upper)
            Mout[c * PW + r] = (c >= r) ? m : 0.0;
            T[(size_t) r * ldt + c] = x[q];                                        // T(c, r) = X(r, c)
            A[(size_t) c * lda + r] = (c >= r) ? Ss[r] * rt[q] : b[q];
            Vw[(size_t) c * ldv + r] = (c < r) ? b[q] : (c == r ? 1.0 : 0.0);
            if (c == r) tau[r] = x[q];
        }
    }
}

// K5: rows >= w of V = Q M, written into Vw and below the diagonal block of the panel
template <bool FULL>
__global__ __launch_bounds__(PT) void final2_kernel(double* __restrict__ Vw, int ldv, int mk, int w, const double* __restrict__ Mm,
                                                    double* __restrict__ A, int lda, const int* __restrict__ guard)
{
    __shared__ double Ms[PW][PW + 1];
    if (*guard != 0) return;
    const int tid = threadIdx.x, r = blockIdx.x * PT + tid;
    for (int el = tid; el < PW * PW; el += PT) {
        const int i = el % PW, c = el / PW;
        Ms[i][c] = (i < w && c < w && i <= c) ? Mm[c * PW + i] : 0.0;
    }
    double q[PW], v[PW];
    {
        const double* p = Vw + min(r, mk - 1);
#pragma unroll
        for (int c = 0; c < PW; ++c) { q[c] = (FULL || c < w) ? p[(size_t) c * ldv] : 0.0; v[c] = 0.0; }
    }
    __syncthreads();
    if (r < w || r >= mk) return;
#pragma unroll
    for (int k = 0; k < PW; ++k) {
#pragma unroll
        for (int c = k; c < PW; ++c) v[c] += q[k] * Ms[k][c];
        __builtin_amdgcn_sched_barrier(0);
    }
#pragma unroll
    for (int c = 0; c < PW; ++c)
        if (FULL || c < w) { Vw[(size_t) c * ldv + r] = v[c]; A[(size_t) c * lda + r] = v[c]; }
}

// ---------------------------------------------------------------------------------------------------------
// launchers: the workgroup is sized to the tallest block of the launch (64..512 threads) -- a 64-row top stack
// runs on one wave with no cross-wave reduction instead of eight mostly-idle ones
static int nt_for(int rows) { return rows <= 64 ? 64 : (rows <= 128 ? 128 : (rows <= 256 ? 256 : 512)); }

// MI355XQR_LEAF_WAVES=8: 512-thread workgroups (two waves per SIMD) for the 512- and 1024-row blocks; default 4: the same
// rows on 256 threads, one wave per SIMD -- per Householder step a SIMD then runs ONE transposing wave reduction instead
// of two and the cross-wave sum has 4 partials instead of 8
static int leaf_waves(void)
{
    static int v = 0;
    if (!v) { const char* e = getenv("MI355XQR_LEAF_WAVES"); v = (e && atoi(e) == 8) ? 8 : 4; }
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

static int panel_tsqr_impl(void* stream, double* P, int ld, int mk, int w, double* tau, double* T, int ldt, double* Vw, int ldv,
                           double* ws, int m_cap, const int* guard)
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
    return (int) hipGetLastError();
}

int qrd_panel_tsqr(void* stream, double* P, int ld, int mk, int w, double* tau, double* T, int ldt, double* Vw, int ldv,
                   double* ws, int m_cap)
{
    return panel_tsqr_impl(stream, P, ld, mk, w, tau, T, ldt, Vw, ldv, ws, m_cap, nullptr);
}

// Leaf = CholeskyQR2 + Householder reconstruction, guarded: 7 short launches, then the Householder-TSQR leaf above as
// launches that return at once unless the guard word says the Cholesky route was refused for this leaf.
// cws: QRD_CHOLQR_WS doubles (G1, G2, R1, M, guard word).
int qrd_panel_cholqr(void* stream, double* P, int ld, int mk, int w, double* tau, double* T, int ldt, double* Vw, int ldv,
                     double* ws, int m_cap, double* cws, double* slabs, size_t slab_cap)
{
    hipStream_t s = (hipStream_t) stream;
    if (w < 1 || w > PW || mk < w || mk > m_cap) return -4;
    // one workgroup covers the leaf / ragged last leaf (the per-column predicates a narrow leaf needs cost the
    // Cholesky kernels 3.5 KB of scratch per thread): Householder path directly
    if (mk <= PT || w < PW) return panel_tsqr_impl(stream, P, ld, mk, w, tau, T, ldt, Vw, ldv, ws, m_cap, nullptr);
    double *G1 = cws, *G2 = cws + PW * PW, *R1 = cws + 2 * PW * PW, *Mm = cws + 3 * PW * PW;
    int* guard = (int*) (cws + 4 * PW * PW);
    const int nblk = (mk + PT - 1) / PT;
    int rc = gram32(s, P, ld, mk, G1, slabs, slab_cap);
    if (rc) return rc;
    hipLaunchKernelGGL(cholq_kernel<true>, dim3(nblk), dim3(PT), 0, s, P, ld, mk, w, G1, R1, Vw, ldv, guard);
    rc = gram32(s, Vw, ldv, mk, G2, slabs, slab_cap);
    if (rc) return rc;
    hipLaunchKernelGGL(hr2_kernel, dim3(1), dim3(64 * HG), 0, s, G2, R1, Vw, ldv, w, P, ld, tau, T, ldt, Mm, guard);
    hipLaunchKernelGGL(final2_kernel<true>, dim3(nblk), dim3(PT), 0, s, Vw, ldv, mk, w, Mm, P, ld, guard);
    rc = (int) hipGetLastError();
    if (rc) return rc;
    return panel_tsqr_impl(stream, P, ld, mk, w, tau, T, ldt, Vw, ldv, ws, m_cap, guard);
}

}   // extern "C"
