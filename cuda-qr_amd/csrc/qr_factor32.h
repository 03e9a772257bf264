// qr_factor32.h -- the 32 x 32 small-factor core on the matrix cores (round 5): Cholesky with the inverse factor, and the modified LU of the
// Householder reconstruction with both inverse factors, each on ONE wave, micro-blocked by 4 columns.
//
// What they replace: the register recurrences CholAugStep / Chol3Step / Hr3Lu of qr_leaf_math.h -- a rank-1 update per pivot, sum over
// K of (31 - K) x (2 v_readlane + 1 FMA) ~ 1500 dependent-ish VALU instructions per factor, 8-11 us each, twice per 32-column leaf and four
// times per 128-column panel.  Here a pivot BLOCK of 4 is factored by all lanes redundantly from 10 (16) broadcast values, and everything
// else is v_mfma_f64_16x16x4_f64, whose K = 4 is exactly one micro-block:
//   * the 32 x 32 matrix lives in 16 x 16 accumulator tiles; register r of a tile is the 4 x 16 strip of rows 4r .. 4r+3 in B-operand layout
//     (lane = column + 16 * row-in-strip), and the SAME register read as the A operand is the strip transposed.  So
//       strip solve      R(K:K+4, :) = U4^-T G(K:K+4, :)        one MFMA per 16 columns, A = U4^-T padded to 16 x 4, result = register 0
//       trailing update  G(i, j) -= sum_k R(K+k, i) R(K+k, j)   one MFMA per tile, A = B = the solved strip registers (negate modifier)
//     no LDS, no shuffle, no transposition anywhere in the loop;
//   * the inverse factor rides along as the identity block of the augmented matrix [G | I] -> [R | R^-T], like the upper lanes did in
//     CholAugStep; the LU carries W and W^T side by side (column strips of W are row strips of W^T) and two identity blocks:
//     [W | I] -> [U' | L1^-1] and [W^T | I] -> [L1^T | U'^-T], so that U'^-1 -- a third 32-step recurrence on another wave until now -- is
//     free as well.
// Critical path per micro-block: 10-16 v_readlane, a 4 x 4 factorisation (4 rsqrt / rcp + two Newton steps each), one solve MFMA, one
// update MFMA; the other 4-16 MFMAs of the micro-block run under the next block's scalar chain.
// Reference: the serial panel kernel qr.cu:60-333 / qr.c:109-235 (norm -> tau -> apply per column), whose small dense factors these are
// in the CholeskyQR2 + Householder-reconstruction form.  Lane-level numpy model: devtools/sim_mfma_factor.py.
#ifndef QR_FACTOR32_H
#define QR_FACTOR32_H
#include "qr_common.h"
#include "qr_leaf_math.h"

__device__ __forceinline__ v4d f32_mma(double a, double b, v4d c) { return __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0); }   // c + a b
__device__ __forceinline__ v4d f32_mms(double a, double b, v4d c) { return __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 1); }   // c - a b

// lower-triangular 4 x 4 matrix (uniform values) as the A operand of a strip solve: A[i = l15][k = l4] = m(i, k), zero elsewhere
struct F32Low4 { double m00, m10, m11, m20, m21, m22, m30, m31, m32, m33; };
__device__ __forceinline__ double f32_low4_operand(const F32Low4& m, int lane)
{
    double a = 0.0;
    a = (lane == 0) ? m.m00 : a;
    a = (lane == 1) ? m.m10 : a;
    a = (lane == 2) ? m.m20 : a;
    a = (lane == 3) ? m.m30 : a;
    a = (lane == 17) ? m.m11 : a;
    a = (lane == 18) ? m.m21 : a;
    a = (lane == 19) ? m.m31 : a;
    a = (lane == 34) ? m.m22 : a;
    a = (lane == 35) ? m.m32 : a;
    a = (lane == 51) ? m.m33 : a;
    return a;
}

// d = U^T U (upper 4 x 4 read): m = U^-T.  false on a non-positive (or NaN) pivot
__device__ __forceinline__ bool f32_chol4(double d00, double d01, double d02, double d03, double d11, double d12, double d13, double d22,
                                          double d23, double d33, F32Low4& m)
{
    bool ok = d00 > 0.0;
    const double i0 = rsqrt_newton(d00);
    const double u01 = d01 * i0, u02 = d02 * i0, u03 = d03 * i0;
    const double p1 = d11 - u01 * u01;
    ok = ok && (p1 > 0.0);
    const double i1 = rsqrt_newton(p1);
    const double u12 = (d12 - u01 * u02) * i1, u13 = (d13 - u01 * u03) * i1;
    const double p2 = (d22 - u02 * u02) - u12 * u12;
    ok = ok && (p2 > 0.0);
    const double i2 = rsqrt_newton(p2);
    const double u23 = ((d23 - u02 * u03) - u12 * u13) * i2;
    const double p3 = ((d33 - u03 * u03) - u13 * u13) - u23 * u23;
    ok = ok && (p3 > 0.0);
    const double i3 = rsqrt_newton(p3);
    m.m00 = i0; m.m11 = i1; m.m22 = i2; m.m33 = i3;
    m.m10 = -(u01 * i0) * i1;
    m.m21 = -(u12 * i1) * i2;
    m.m32 = -(u23 * i2) * i3;
    m.m20 = -(u02 * i0 + u12 * m.m10) * i2;
    m.m31 = -(u13 * i1 + u23 * m.m21) * i3;
    m.m30 = -((u03 * i0 + u13 * m.m10) + u23 * m.m20) * i3;
    return ok;
}

// ---------------------------------------------------------------------------------------------------------------------------------
// Cholesky G = R^T R of a symmetric 32 x 32 matrix with X = R^-T, one wave.
//   loadG(i, j)        G(i, j); called for every (i, j) of the tiles (0,0), (0,1), (1,1): entries below the diagonal must be FINITE (they are
//                      multiplied by zeros), e.g. the mirrored value
//   storeR(i, j, v)    R(i, j) for every j < 32 (v = 0 left of the diagonal)
//   storeX(i, j, v)    X(i, j) = R^-T(i, j) for every j < 32 (v = 0 right of the diagonal)
// Each (i, j) is stored by exactly one lane.  Returns false (uniform) on a non-positive or NaN pivot; the outputs are then garbage.
// ---------------------------------------------------------------------------------------------------------------------------------
template <int S, class SR, class SX>
struct F32CholStep {
    static __device__ __forceinline__ void run(v4d (&T)[3], v4d (&X)[3], int lane, bool& ok, SR& storeR, SX& storeX)
    {
        constexpr int K = 4 * S, ti = S / 4, r = S % 4;
        constexpr int tdiag = ti ? 2 : 0;                                    // T index of tile (ti, ti); tiles: 0 = (0,0), 1 = (0,1), 2 = (1,1)
        const int l15 = lane & 15, l4 = lane >> 4;
        const double dg = T[tdiag][r];
        F32Low4 m;
        const bool okk = f32_chol4(readlane_f64(dg, 4 * r), readlane_f64(dg, 4 * r + 1), readlane_f64(dg, 4 * r + 2), readlane_f64(dg, 4 * r + 3),
                                   readlane_f64(dg, 16 + 4 * r + 1), readlane_f64(dg, 16 + 4 * r + 2), readlane_f64(dg, 16 + 4 * r + 3),
                                   readlane_f64(dg, 32 + 4 * r + 2), readlane_f64(dg, 32 + 4 * r + 3), readlane_f64(dg, 48 + 4 * r + 3), m);
        ok = ok && okk;
        const double am = f32_low4_operand(m, lane);
        const v4d zero = (v4d){0.0, 0.0, 0.0, 0.0};
        const int row = K + l4;
        // the solved strips: s1 = columns 16 .. 31, s0 = columns 0 .. 15 (first tile row only); masked to the upper triangle
        double s0 = 0.0, s1;
        if constexpr (ti == 0) {
            s0 = f32_mma(am, T[0][r], zero)[0];
            s0 = (l15 >= row) ? s0 : 0.0;
            // (the tile that holds the next pivot block first: the rest runs under the next block's scalar chain)
            T[0] = f32_mms(s0, s0, T[0]);
            s1 = f32_mma(am, T[1][r], zero)[0];
            T[1] = f32_mms(s0, s1, T[1]);
            T[2] = f32_mms(s1, s1, T[2]);
            storeR(row, l15, s0);
            storeR(row, 16 + l15, s1);
            const double y0 = f32_mma(am, X[0][r], zero)[0];
            X[0] = f32_mms(s0, y0, X[0]);
            X[1] = f32_mms(s1, y0, X[1]);
            storeX(row, l15, y0);
            storeX(row, 16 + l15, 0.0);
        } else {
            s1 = f32_mma(am, T[2][r], zero)[0];
            s1 = (16 + l15 >= row) ? s1 : 0.0;
            T[2] = f32_mms(s1, s1, T[2]);
            storeR(row, l15, 0.0);
            storeR(row, 16 + l15, s1);
            const double y0 = f32_mma(am, X[1][r], zero)[0], y1 = f32_mma(am, X[2][r], zero)[0];
            X[1] = f32_mms(s1, y0, X[1]);
            X[2] = f32_mms(s1, y1, X[2]);
            storeX(row, l15, y0);
            storeX(row, 16 + l15, y1);
        }
        if constexpr (S + 1 < 8) F32CholStep<S + 1, SR, SX>::run(T, X, lane, ok, storeR, storeX);
    }
};

template <class LG, class SR, class SX>
__device__ __forceinline__ bool chol32_mfma(int lane, LG loadG, SR storeR, SX storeX)
{
    const int l15 = lane & 15, l4 = lane >> 4;
    v4d T[3], X[3];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int i = 4 * r + l4;
        T[0][r] = loadG(i, l15);
        T[1][r] = loadG(i, 16 + l15);
        T[2][r] = loadG(16 + i, 16 + l15);
        X[0][r] = (i == l15) ? 1.0 : 0.0;
        X[1][r] = 0.0;
        X[2][r] = (i == l15) ? 1.0 : 0.0;
    }
    bool ok = true;
    F32CholStep<0, SR, SX>::run(T, X, lane, ok, storeR, storeX);
    return ok;
}

// ---------------------------------------------------------------------------------------------------------------------------------
// Modified LU of the Householder reconstruction, W - S R2 = L1 U' with S_i = -sign of the pivot position's current entry (the sign
// Householder's reflector would give R, reference qr.c:141-151), one wave.  R2 upper triangular with positive diagonal.
//   loadW(i, j)   W(i, j), any (i, j) < 32;   loadR2(i, j)   R2(i, j), called with any (i, j): must return 0 for j < i
//   storeLU(i, j, v)   L1(i, j) for j < i (unit diagonal not stored), U'(i, j) for j >= i       (every (i, j) exactly once)
//   storeS(i, v)       S_i
//   storeLi(i, j, v)   L1^-1(i, j), every j < 32 (v = 0 right of the diagonal, 1 on it)
//   storeUit(i, j, v)  U'^-T(i, j) = U'^-1(j, i), every j < 32 (v = 0 right of the diagonal)
// ---------------------------------------------------------------------------------------------------------------------------------
struct F32Lu4 {
    double s0, s1, s2, s3;            // signs
    double l10, l20, l21, l30, l31, l32;
    F32Low4 li, uit;                  // L^-1 (unit diagonal), U^-T
};
__device__ __forceinline__ void f32_lu4(double (&w)[4][4], const double (&r2)[4][4], F32Lu4& o)
{
    double inv[4], S[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        S[i] = (w[i][i] >= 0.0) ? -1.0 : 1.0;
#pragma unroll
        for (int j = i; j < 4; ++j) w[i][j] -= S[i] * r2[i][j];
        inv[i] = rcp_newton(w[i][i]);                                        // |pivot| >= R2(i, i) > 0
#pragma unroll
        for (int jp = i + 1; jp < 4; ++jp) {
            const double l = w[jp][i] * inv[i];
            w[jp][i] = l;
#pragma unroll
            for (int c = i + 1; c < 4; ++c) w[jp][c] -= l * w[i][c];
        }
    }
    o.s0 = S[0]; o.s1 = S[1]; o.s2 = S[2]; o.s3 = S[3];
    o.l10 = w[1][0]; o.l20 = w[2][0]; o.l21 = w[2][1]; o.l30 = w[3][0]; o.l31 = w[3][1]; o.l32 = w[3][2];
    // L^-1 (unit lower): x(i, j) = -sum_{k = j}^{i - 1} l(i, k) x(k, j)
    o.li.m00 = 1.0; o.li.m11 = 1.0; o.li.m22 = 1.0; o.li.m33 = 1.0;
    o.li.m10 = -o.l10;
    o.li.m21 = -o.l21;
    o.li.m32 = -o.l32;
    o.li.m20 = -(o.l20 + o.l21 * o.li.m10);
    o.li.m31 = -(o.l31 + o.l32 * o.li.m21);
    o.li.m30 = -((o.l30 + o.l31 * o.li.m10) + o.l32 * o.li.m20);
    // U^-T (lower): m(i, j) = -(sum_{k = j}^{i - 1} u(k, i) m(k, j)) / u(i, i)
    o.uit.m00 = inv[0]; o.uit.m11 = inv[1]; o.uit.m22 = inv[2]; o.uit.m33 = inv[3];
    o.uit.m10 = -(w[0][1] * inv[0]) * inv[1];
    o.uit.m21 = -(w[1][2] * inv[1]) * inv[2];
    o.uit.m32 = -(w[2][3] * inv[2]) * inv[3];
    o.uit.m20 = -(w[0][2] * inv[0] + w[1][2] * o.uit.m10) * inv[2];
    o.uit.m31 = -(w[1][3] * inv[1] + w[2][3] * o.uit.m21) * inv[3];
    o.uit.m30 = -((w[0][3] * inv[0] + w[1][3] * o.uit.m10) + w[2][3] * o.uit.m20) * inv[3];
}

// tiles: Wt[2 a + b] = W(16 a .., 16 b ..); Vt[2 a + b] = W^T(16 a .., 16 b ..); R2t: 0 = (0,0), 1 = (0,1), 2 = (1,1); XL / XU: 0 = (0,0), 1 = (1,0), 2 = (1,1)
template <int S, class SLU, class SS, class SLI, class SUI>
struct F32LuStep {
    static __device__ __forceinline__ void run(v4d (&Wt)[4], v4d (&Vt)[4], const v4d (&R2t)[3], v4d (&XL)[3], v4d (&XU)[3], int lane, SLU& storeLU,
                                               SS& storeS, SLI& storeLi, SUI& storeUit)
    {
        constexpr int K = 4 * S, ti = S / 4, r = S % 4;
        constexpr int wd = ti ? 3 : 0, rd = ti ? 2 : 0;
        const int l15 = lane & 15, l4 = lane >> 4;
        const double wg = Wt[wd][r], rg = R2t[rd][r];
        double w[4][4], r2[4][4];
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                w[i][j] = readlane_f64(wg, 16 * i + 4 * r + j);
                r2[i][j] = (j >= i) ? readlane_f64(rg, 16 * i + 4 * r + j) : 0.0;
            }
        F32Lu4 f;
        f32_lu4(w, r2, f);
        const double al = f32_low4_operand(f.li, lane), au = f32_low4_operand(f.uit, lane);
        const double ssel = (l4 == 0) ? f.s0 : ((l4 == 1) ? f.s1 : ((l4 == 2) ? f.s2 : f.s3));
        if (l15 == 0) storeS(K + l4, ssel);
        {   // strictly lower part of the pivot block of L1: lane (l15 = row, l4 = column)
            double lb = 0.0;
            lb = (lane == 1) ? f.l10 : lb;
            lb = (lane == 2) ? f.l20 : lb;
            lb = (lane == 3) ? f.l30 : lb;
            lb = (lane == 18) ? f.l21 : lb;
            lb = (lane == 19) ? f.l31 : lb;
            lb = (lane == 35) ? f.l32 : lb;
            if (l15 < 4 && l4 < l15) storeLU(K + l15, K + l4, lb);
        }
        const v4d zero = (v4d){0.0, 0.0, 0.0, 0.0};
        const int row = K + l4;
        if constexpr (ti == 0) {
            // W side: U' strips (columns 0..15 / 16..31), W^T side: L1 strips (rows 0..15 / 16..31 of L1, as lt[lane (k, l15)] = L1(16 b + l15, K + k))
            double u0 = f32_mma(al, Wt[0][r] - ssel * R2t[0][r], zero)[0];
            u0 = (l15 >= row) ? u0 : 0.0;
            double lt0 = f32_mma(au, Vt[0][r], zero)[0];
            lt0 = (l15 >= K + 4) ? lt0 : 0.0;
            Wt[0] = f32_mms(lt0, u0, Wt[0]);
            Vt[0] = f32_mms(u0, lt0, Vt[0]);
            const double u1 = f32_mma(al, Wt[1][r] - ssel * R2t[1][r], zero)[0];
            const double lt1 = f32_mma(au, Vt[1][r], zero)[0];          // rows 16 .. 31 of L1: all below the pivot block
            Wt[1] = f32_mms(lt0, u1, Wt[1]);
            Wt[2] = f32_mms(lt1, u0, Wt[2]);
            Wt[3] = f32_mms(lt1, u1, Wt[3]);
            Vt[1] = f32_mms(u0, lt1, Vt[1]);
            Vt[2] = f32_mms(u1, lt0, Vt[2]);
            Vt[3] = f32_mms(u1, lt1, Vt[3]);
            if (l15 >= row) storeLU(row, l15, u0);
            storeLU(row, 16 + l15, u1);
            if (l15 >= K + 4) storeLU(l15, K + l4, lt0);
            storeLU(16 + l15, K + l4, lt1);
            const double yl = f32_mma(al, XL[0][r], zero)[0], yu = f32_mma(au, XU[0][r], zero)[0];
            XL[0] = f32_mms(lt0, yl, XL[0]);
            XL[1] = f32_mms(lt1, yl, XL[1]);
            XU[0] = f32_mms(u0, yu, XU[0]);
            XU[1] = f32_mms(u1, yu, XU[1]);
            storeLi(row, l15, yl);
            storeLi(row, 16 + l15, 0.0);
            storeUit(row, l15, yu);
            storeUit(row, 16 + l15, 0.0);
        } else {
            double u1 = f32_mma(al, Wt[3][r] - ssel * R2t[2][r], zero)[0];
            u1 = (16 + l15 >= row) ? u1 : 0.0;
            double lt1 = f32_mma(au, Vt[3][r], zero)[0];
            lt1 = (16 + l15 >= K + 4) ? lt1 : 0.0;
            Wt[3] = f32_mms(lt1, u1, Wt[3]);
            Vt[3] = f32_mms(u1, lt1, Vt[3]);
            if (16 + l15 >= row) storeLU(row, 16 + l15, u1);
            if (16 + l15 >= K + 4) storeLU(16 + l15, K + l4, lt1);
            const double yl0 = f32_mma(al, XL[1][r], zero)[0], yl1 = f32_mma(al, XL[2][r], zero)[0];
            const double yu0 = f32_mma(au, XU[1][r], zero)[0], yu1 = f32_mma(au, XU[2][r], zero)[0];
            XL[1] = f32_mms(lt1, yl0, XL[1]);
            XL[2] = f32_mms(lt1, yl1, XL[2]);
            XU[1] = f32_mms(u1, yu0, XU[1]);
            XU[2] = f32_mms(u1, yu1, XU[2]);
            storeLi(row, l15, yl0);
            storeLi(row, 16 + l15, yl1);
            storeUit(row, l15, yu0);
            storeUit(row, 16 + l15, yu1);
        }
        if constexpr (S + 1 < 8) F32LuStep<S + 1, SLU, SS, SLI, SUI>::run(Wt, Vt, R2t, XL, XU, lane, storeLU, storeS, storeLi, storeUit);
    }
};

// R2 in operand tiles (requested early by callers whose R2 comes from global memory: the loads then run under other work)
template <class LR>
__device__ __forceinline__ void lu32_load_r2(int lane, LR loadR2, v4d (&R2t)[3])
{
    const int l15 = lane & 15, l4 = lane >> 4;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int i = 4 * r + l4;
        R2t[0][r] = loadR2(i, l15);
        R2t[1][r] = loadR2(i, 16 + l15);
        R2t[2][r] = loadR2(16 + i, 16 + l15);
    }
}

template <class LW, class SLU, class SS, class SLI, class SUI>
__device__ __forceinline__ void lu32_mfma_r2(int lane, LW loadW, const v4d (&R2t)[3], SLU storeLU, SS storeS, SLI storeLi, SUI storeUit)
{
    const int l15 = lane & 15, l4 = lane >> 4;
    v4d Wt[4], Vt[4], XL[3], XU[3];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int i = 4 * r + l4;
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int b = 0; b < 2; ++b) {
                Wt[2 * a + b][r] = loadW(16 * a + i, 16 * b + l15);
                Vt[2 * a + b][r] = loadW(16 * b + l15, 16 * a + i);
            }
        XL[0][r] = (i == l15) ? 1.0 : 0.0;
        XL[1][r] = 0.0;
        XL[2][r] = (i == l15) ? 1.0 : 0.0;
        XU[0][r] = XL[0][r]; XU[1][r] = 0.0; XU[2][r] = XL[2][r];
    }
    F32LuStep<0, SLU, SS, SLI, SUI>::run(Wt, Vt, R2t, XL, XU, lane, storeLU, storeS, storeLi, storeUit);
}

template <class LW, class LR, class SLU, class SS, class SLI, class SUI>
__device__ __forceinline__ void lu32_mfma(int lane, LW loadW, LR loadR2, SLU storeLU, SS storeS, SLI storeLi, SUI storeUit)
{
    v4d R2t[3];
    lu32_load_r2(lane, loadR2, R2t);
    lu32_mfma_r2(lane, loadW, R2t, storeLU, storeS, storeLi, storeUit);
}

#endif
