// qr_factor32_dbg.hip -- development / test entry points of the 32 x 32 small-factor core (qr_factor32.h): one wave per matrix, LDS in and
// out like the kernels that use the routines, with the register recurrences they replace (qr_leaf_math.h) as variant 0 for timing.
// Used by tests/test_gpu_factor32.py (parity against numpy) and devtools/tools_factor32.py (profiles/r05_leaf_phase_stamps.txt); not
// on any product path.
#include <hip/hip_runtime.h>
#include "qr_device.h"
#include "qr_common.h"
#include "qr_leaf_math.h"
#include "qr_factor32.h"

namespace {

// all matrices row-major 32 x 32: element (i, j) at [32 i + j]
__global__ __launch_bounds__(64) void f32_chol_test_kernel(const double* __restrict__ G, double* __restrict__ R, double* __restrict__ X,
                                                           int* __restrict__ okout, unsigned long long* __restrict__ ticks, int variant, int reps)
{
    __shared__ double Gs[32][33], Rs[32][33], Xs[32][33];
    int lane = threadIdx.x;
    const double* g = G + (size_t) blockIdx.x * 1024;
    for (int e = lane; e < 1024; e += 64) { Gs[e >> 5][e & 31] = g[e]; Rs[e >> 5][e & 31] = 777.0; Xs[e >> 5][e & 31] = 777.0; }      // (poisoned: every entry must be written)
    __syncthreads();
    bool ok = true;
    const unsigned long long t0 = wall_clock64();
    for (int rep = 0; rep < reps; ++rep) {
        asm volatile("" : "+v"(lane));
        if (variant == 0) {
            const int rc = lane & 31;
            double gg[PW];
#pragma unroll
            for (int i = 0; i < PW; ++i) gg[i] = (lane < PW) ? Gs[rc][i] : (i == rc ? 1.0 : 0.0);
            bool okk = true;
            CholAugStep<0>::run(gg, lane, okk);
            ok = ok && okk;
            if (lane < PW) {
#pragma unroll
                for (int k = 0; k < PW; ++k) Rs[k][rc] = (k <= rc) ? gg[k] : 0.0;
            } else {
#pragma unroll
                for (int k = 0; k < PW; ++k) Xs[k][rc] = (k >= rc) ? gg[k] : 0.0;          // X(k, rc) = R^-T(k, rc)
            }
        } else {
            const bool okk = chol32_mfma(lane, [&](int i, int j) { return (j >= i) ? Gs[i][j] : Gs[j][i]; },
                                         [&](int i, int j, double v) { Rs[i][j] = v; }, [&](int i, int j, double v) { Xs[i][j] = v; });
            ok = ok && okk;
        }
        __syncthreads();
    }
    const unsigned long long t1 = wall_clock64();
    for (int e = lane; e < 1024; e += 64) {
        R[(size_t) blockIdx.x * 1024 + e] = Rs[e >> 5][e & 31];
        X[(size_t) blockIdx.x * 1024 + e] = Xs[e >> 5][e & 31];
    }
    if (lane == 0) { okout[blockIdx.x] = ok ? 1 : 0; ticks[blockIdx.x] = t1 - t0; }
}

// variant 0: Hr3Lu (L1^-1 on the upper lanes) followed by the U'^-1 recurrence (in the kernels: on a second wave, after the LU)
__global__ __launch_bounds__(64) void f32_lu_test_kernel(const double* __restrict__ W, const double* __restrict__ R2, double* __restrict__ LU,
                                                         double* __restrict__ S, double* __restrict__ Li, double* __restrict__ Uit,
                                                         unsigned long long* __restrict__ ticks, int variant, int reps)
{
    __shared__ double Ws[32][33], R2s[32][33], Bs[32][33], Ls[32][33], Us[32][33], Ss[32];
    int lane = threadIdx.x;
    for (int e = lane; e < 1024; e += 64) {
        Ws[e >> 5][e & 31] = W[(size_t) blockIdx.x * 1024 + e];
        R2s[e >> 5][e & 31] = ((e & 31) >= (e >> 5)) ? R2[(size_t) blockIdx.x * 1024 + e] : 0.0;
        Bs[e >> 5][e & 31] = 777.0; Ls[e >> 5][e & 31] = 777.0; Us[e >> 5][e & 31] = 777.0;
    }
    __syncthreads();
    const unsigned long long t0 = wall_clock64();
    for (int rep = 0; rep < reps; ++rep) {
        asm volatile("" : "+v"(lane));
        if (variant == 0) {
            const int rc = lane & 31;
            double b[PW], gg[PW];
#pragma unroll
            for (int r = 0; r < PW; ++r) {
                b[r] = (lane < PW) ? Ws[r][rc] : (r == rc ? 1.0 : 0.0);
                gg[r] = (lane < PW) ? R2s[r][rc] : 0.0;
            }
            double sgn = 1.0;
            Hr3Lu<0>::run(b, gg, lane, sgn);
            if (lane < PW) {
#pragma unroll
                for (int k = 0; k < PW; ++k) Bs[k][rc] = b[k];
                Ss[rc] = sgn;
            } else {
#pragma unroll
                for (int k = 0; k < PW; ++k) Ls[k][rc] = (k > rc) ? b[k] : (k == rc ? 1.0 : 0.0);
            }
            __syncthreads();
            double x[PW];
            UpperInv<PW - 1>::run(x, Bs, rcp_newton(Bs[rc][rc]), rc);
            if (lane < PW) {
#pragma unroll
                for (int i = 0; i < PW; ++i) Us[rc][i] = (i <= rc) ? x[i] : 0.0;            // Us[j][i] = U'^-1(i, j) = U'^-T(j, i)
            }
        } else {
            lu32_mfma(lane, [&](int i, int j) { return Ws[i][j]; }, [&](int i, int j) { return R2s[i][j]; },
                      [&](int i, int j, double v) { Bs[i][j] = v; }, [&](int i, double v) { Ss[i] = v; },
                      [&](int i, int j, double v) { Ls[i][j] = v; }, [&](int i, int j, double v) { Us[i][j] = v; });
        }
        __syncthreads();
    }
    const unsigned long long t1 = wall_clock64();
    for (int e = lane; e < 1024; e += 64) {
        LU[(size_t) blockIdx.x * 1024 + e] = Bs[e >> 5][e & 31];
        Li[(size_t) blockIdx.x * 1024 + e] = Ls[e >> 5][e & 31];
        Uit[(size_t) blockIdx.x * 1024 + e] = Us[e >> 5][e & 31];
    }
    if (lane < 32) S[(size_t) blockIdx.x * 32 + lane] = Ss[lane];
    if (lane == 0) ticks[blockIdx.x] = t1 - t0;
}

}   // namespace

extern "C" {

// ticks: per matrix, 100 MHz wall-clock ticks of `reps` back-to-back factorisations (LDS -> LDS)
int qrd_dbg_chol32(void* stream, const double* G, double* R, double* X, int* ok, unsigned long long* ticks, int nmat, int variant, int reps)
{
    if (nmat < 1 || reps < 1) return -1;
    hipLaunchKernelGGL(f32_chol_test_kernel, dim3(nmat), dim3(64), 0, (hipStream_t) stream, G, R, X, ok, ticks, variant, reps);
    return (int) hipGetLastError();
}

int qrd_dbg_lu32(void* stream, const double* W, const double* R2, double* LU, double* S, double* Li, double* Uit, unsigned long long* ticks,
                 int nmat, int variant, int reps)
{
    if (nmat < 1 || reps < 1) return -1;
    hipLaunchKernelGGL(f32_lu_test_kernel, dim3(nmat), dim3(64), 0, (hipStream_t) stream, W, R2, LU, S, Li, Uit, ticks, variant, reps);
    return (int) hipGetLastError();
}

}   // extern "C"
