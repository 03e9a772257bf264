// qr_leaf_math.h -- the one-wave 32 x 32 recurrences of the CholeskyQR2 + Householder-reconstruction leaf (Cholesky with the
// inverse factor on the idle half of the wave, the modified LU of the reconstruction, the three triangular solves), shared by the
// per-leaf launch sequence (qr_panel_tsqr.hip) and the one-launch panel (qr_panel_fused.hip).  Reference: the serial panel of
// qr.cu:60-333 / qr.c:109-235, whose per-column norm -> tau -> apply chain these replace.
#ifndef QR_LEAF_MATH_H
#define QR_LEAF_MATH_H
#include "qr_common.h"

#ifndef PW
#define PW LEAFW          // max leaf width
#endif

__device__ __forceinline__ double readlane_f64(double v, int lane)
{
    return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), lane), __builtin_amdgcn_readlane(__double2loint(v), lane));
}

#define QRD_GUARD_THR (1.0 / 64.0)
#define QRD_CHOL1_THR 1e-9      /* hr3_kernel: below this max|Q^T Q - I| the second Cholesky factor is taken to first order */

// 1/sqrt(p) from the hardware estimate (v_rsq_f64, ~2^-26 relative) and two Newton steps y <- y (3/2 - p/2 y^2): full double
// accuracy to an ulp or two in ~10 dependent instructions.  The IEEE sqrt + division this replaces is ~60 instructions
// deep and sat 32 times on the one-wave critical path of every Cholesky (4 of its 7 us).  p <= 0 / NaN is caught by the caller.
__device__ __forceinline__ double rsqrt_newton(double p)
{
    double y = __builtin_amdgcn_rsq(p);
    const double h = 0.5 * p;
    y = y * (1.5 - h * y * y);
    y = y * (1.5 - h * y * y);
    return y;
}
__device__ __forceinline__ double rcp_newton(double p)
{
    double y = __builtin_amdgcn_rcp(p);
    y = y * (2.0 - p * y);
    y = y * (2.0 - p * y);
    return y;
}

template <int K> struct CholAugStep {
    static __device__ __forceinline__ void run(double (&g)[PW], int lane, bool& ok)
    {
        const double p = readlane_f64(g[K], K);
        ok = ok && (p > 0.0);                       // false for NaN as well
        const double inv = rsqrt_newton(p);
        const double rk = (lane >= K) ? g[K] * inv : 0.0;          // lanes >= 32 (identity columns) always pass
        g[K] = rk;
#pragma unroll
        for (int i = K + 1; i < PW; ++i) g[i] -= readlane_f64(rk, i) * rk;
        if constexpr (K + 1 < PW) CholAugStep<K + 1>::run(g, lane, ok);
    }
};

__device__ __forceinline__ void tq_load4(const double* __restrict__ p, double (&d)[4])
{
    const v2d a = *reinterpret_cast<const v2d*>(p), b = *reinterpret_cast<const v2d*>(p + 2);
    d[0] = a[0]; d[1] = a[1]; d[2] = b[0]; d[3] = b[1];
}

template <int K> struct Chol3Step {
    static __device__ __forceinline__ void run(double (&g)[PW], int lane, bool& ok, double& dinv)
    {
        const double p = readlane_f64(g[K], K);
        ok = ok && (p > 0.0);                       // false for NaN as well
        const double inv = rsqrt_newton(p);
        if (lane == K) dinv = inv;                  // 1 / R2(K, K)
        const double rk = (lane >= K) ? g[K] * inv : 0.0;
        g[K] = rk;
#pragma unroll
        for (int i = K + 1; i < PW; ++i) g[i] -= readlane_f64(rk, i) * rk;
        if constexpr (K + 1 < PW) Chol3Step<K + 1>::run(g, lane, ok, dinv);
    }
};
template <int I> struct Hr3Lu {
    static __device__ __forceinline__ void run(double (&b)[PW], const double (&g)[PW], int lane, double& sgn)
    {
        const double x = readlane_f64(b[I], I);                 // current (I, I) entry, S_I not yet applied
        const double S = (x >= 0.0) ? -1.0 : 1.0;
        b[I] -= S * g[I];                                        // row I of S R2 (g[I] = R2(I, lane) is zero left of the diagonal)
        const double piv = readlane_f64(b[I], I);                // x - S R2(I, I): |piv| >= R2(I, I) > 0
        const double inv = rcp_newton(piv);
        if (lane == I) sgn = S;
        const double scale = (lane == I) ? inv : 1.0;
        const double u = (lane > I) ? b[I] : 0.0;
#pragma unroll
        for (int r = I + 1; r < PW; ++r) {
            b[r] *= scale;                                       // lane I: multiplier l_r
            b[r] -= readlane_f64(b[r], I) * u;                   // lanes right of I: a(r, c) -= l_r u_c
        }
        if constexpr (I + 1 < PW) Hr3Lu<I + 1>::run(b, g, lane, sgn);
    }
};

// upper-triangular row solve  u R = u'  for lane = row (rows beyond 31 compute garbage that is never stored):
// u(c) = (u'(c) - sum_{k<c} u(k) R(k, c)) / R(c, c)
template <int C> struct RowSolve {
    static __device__ __forceinline__ void run(double (&u)[PW], double (*Rm)[PW + 1], const double* rinv)
    {
        double acc = u[C];
#pragma unroll
        for (int k = 0; k < C; ++k) acc -= u[k] * Rm[k][C];
        u[C] = acc * rinv[C];
        if constexpr (C + 1 < PW) RowSolve<C + 1>::run(u, Rm, rinv);
    }
};
// unit-lower column solve  L x = e_lane : x(i) = delta(i, lane) - sum_{k<i} L(i, k) x(k)
template <int I> struct UnitLowerInv {
    static __device__ __forceinline__ void run(double (&x)[PW], double (*Lm)[PW + 1], int lane)
    {
        double acc = (I == lane) ? 1.0 : 0.0;
#pragma unroll
        for (int k = 0; k < I; ++k) acc -= Lm[I][k] * x[k];
        x[I] = acc;
        if constexpr (I + 1 < PW) UnitLowerInv<I + 1>::run(x, Lm, lane);
    }
};

// back substitution  U' X = I  by columns: lane j computes column j of X = U'^-1 (zero below the diagonal);  Um[i][k] = U'(i, k)
template <int I> struct UpperInv {
    static __device__ __forceinline__ void run(double (&x)[PW], double (*Um)[PW + 1], double dinv, int lane)
    {
        double acc = (I == lane) ? 1.0 : 0.0;
#pragma unroll
        for (int k = I + 1; k < PW; ++k) acc -= Um[I][k] * x[k];
        x[I] = acc * readlane_f64(dinv, I);                    // dinv: 1 / U'(lane, lane) in lane `lane`
        if constexpr (I > 0) UpperInv<I - 1>::run(x, Um, dinv, lane);
    }
};

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


#endif
