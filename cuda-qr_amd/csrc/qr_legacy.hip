// qr_legacy.hip -- the reference's own sliding-window "MMQR" (qr.c:55-313) on the device, for callers that consume the RAW factored
// form: the window-indexed tau array (qr.c:300-304) and the reflector tails where the reference leaves them.  Those depend on the
// reference's compile-time window PR x PC (qr.c:12-13) and describe ITS reflector set -- they cannot be derived from the blocked
// factorisation of qr_geqrf_dev -- so this shim runs the reference's schedule itself ("legacy-layout shim").
// Low value by design (the only consumer in the reference is main's debug print, qr.c:483-490): written for clarity, not speed.
//
// Not a port of qr.cu (one block, 2 launches per window, Y W^T rebuilt entry by entry): per COLUMN PANEL there are two launches --
//   legacy_panel_kernel<PC>     one wave walks the panel's windows bottom -> top (they overlap by PC rows: a serial chain, qr.c:73),
//                               lane = window row, the window's PC columns in registers; Householder norm and every v^T a by wave
//                               shuffles; leaves R / reflector tails in the matrix, tau in the reference's index, and each
//                               window's WY pair (Y, W of qr.c:170-213) in a workspace
//   legacy_trailing_kernel<PC>  one wave per trailing column walks the same windows: a <- a + Y (W^T a) (qr.c:255-293 without the
//                               PR^2 PC rebuild per column); the PC rows two consecutive windows share travel between lanes, so a
//                               column is read and written exactly once per window with no read-after-write through memory
// and the explicit m x m Q of qr.c:330-438 is ONE launch: row i of Q only ever combines with itself (Q <- Q H), so a thread owns a row
// and walks all reflectors in the reference's order (legacy_formq_kernel).
// Arithmetic: the reference's formulas with sums in wave-reduction order -- equal to the reference to rounding (tests: 1e-12 of the
// matrix scale on factored matrix, tau and Q), not bitwise.  Window shapes: PR <= 64, PC in {2, 4, 8, 16}, PC < PR.
#include <hip/hip_runtime.h>
#include "qr_device.h"

namespace {

__device__ __forceinline__ double wave_sum(double v)
{
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}

struct LegacyDims { int m, n, PR, PC, rowPanels; };

// windows of panel pc: pr = m - PR, m - PR - (PR - PC), ... while pr + PR > pc && pr >= 0   (qr.c:73)
__device__ __host__ inline int legacy_windows(int m, int PR, int PC, int pc)
{
    int cnt = 0;
    for (int pr = m - PR; pr + PR > pc && pr >= 0; pr -= PR - PC) ++cnt;
    return cnt;
}

template <int PC>
__global__ __launch_bounds__(64) void legacy_panel_kernel(double* __restrict__ A, LegacyDims d, int pc, int pcCount, double* __restrict__ tau,
                                                          double* __restrict__ wy)
{
    const int lane = threadIdx.x, m = d.m, PR = d.PR;
    const bool live = lane < PR;
    const int step = PR - PC;
    int prCount = 0;
    double x[PC];
#pragma unroll
    for (int c = 0; c < PC; ++c) x[c] = 0.0;
    for (int pr = m - PR; pr + PR > pc && pr >= 0; pr -= step, ++prCount) {
        double W[PC], Y[PC];
#pragma unroll
        for (int c = 0; c < PC; ++c) {
            // gather (qr.c:81-87): the bottom PC rows of this window are the top PC rows of the window below, which this wave has just
            // produced -- they travel between lanes instead of through memory
            const double carried = __shfl(x[c], lane >= step ? lane - step : 0, 64);
            const double fresh = (live && (prCount == 0 || lane < step)) ? A[(size_t) (pc + c) * m + pr + lane] : 0.0;
            x[c] = (prCount > 0 && lane >= step && live) ? carried : fresh;
            W[c] = 0.0; Y[c] = 0.0;                                        // qr.c:100-107
        }
        const bool bottom = (pr == m - PR), top = (pr <= pc);             // qr.c:109-111
#pragma unroll
        for (int c = 0; c < PC; ++c) {
            const int vstart = top ? pc - pr + c : c;                      // the 4-case table of qr.c:117-141
            const int vend = bottom ? PR : PR - PC + c + 1;
            const bool in = lane >= vstart && lane < vend;
            // Householder vector (qr.c:144-167): norm, sign, u, tau, x / u
            const double xs = in ? x[c] : 0.0;
            const double norm = sqrt(wave_sum(xs * xs));
            const double x0 = __shfl(x[c], vstart, 64);
            const double sign = (x0 < 0.0) ? -1.0 : 1.0;
            const double u = x0 + sign * norm;
            const double t = sign * u / norm;                              // NaN for a zero column, like the reference (qr.c:152)
            const double v = (lane == vstart) ? 1.0 : (in ? x[c] / u : 0.0);
            if (lane == vstart) x[c] = -sign * norm;
            else if (in) x[c] = v;
            // next W column: z = -t v - t W (Y^T v)   (qr.c:170-202)
            double z = -t * v;
#pragma unroll
            for (int k = 0; k < PC; ++k)
                if (k < c) z -= t * W[k] * wave_sum(Y[k] * v);
            W[c] = z;                                                      // qr.c:204-207
            Y[c] = v;                                                      // qr.c:210-213
            // apply H to the window's remaining columns: a <- a - t v (v^T a)   (qr.c:215-235)
#pragma unroll
            for (int ac = 0; ac < PC; ++ac)
                if (ac > c) {
                    const double dot = wave_sum(v * x[ac]);
                    x[ac] -= t * v * dot;
                }
            if (lane == 0) tau[((size_t) d.rowPanels * pcCount + prCount) * PC + c] = t;        // qr.c:300-304
        }
        double* wyw = wy + (size_t) prCount * 2 * PC * 64;
#pragma unroll
        for (int c = 0; c < PC; ++c) {
            if (live) A[(size_t) (pc + c) * m + pr + lane] = x[c];         // scatter, qr.c:242-248
            wyw[c * 64 + lane] = Y[c];
            wyw[(PC + c) * 64 + lane] = W[c];
        }
    }
}

template <int PC>
__global__ __launch_bounds__(256) void legacy_trailing_kernel(double* __restrict__ A, LegacyDims d, int pc, const double* __restrict__ wy)
{
    const int lane = threadIdx.x & 63, m = d.m, PR = d.PR;
    const int col = pc + PC + (int) (blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6));
    if (col >= d.n) return;
    double* a = A + (size_t) col * m;
    const bool live = lane < PR;
    const int step = PR - PC;
    double x = 0.0;
    int w = 0;
    for (int pr = m - PR; pr + PR > pc && pr >= 0; pr -= step, ++w) {
        // this window's rows of the column: the bottom PC of them are the previous (lower) window's top PC, still in registers
        const double carried = __shfl(x, lane >= step ? lane - step : 0, 64);
        const double fresh = (live && (w == 0 || lane < step)) ? a[pr + lane] : 0.0;
        x = (w > 0 && lane >= step && live) ? carried : fresh;
        const double* wyw = wy + (size_t) w * 2 * PC * 64;
        double s[PC];
#pragma unroll
        for (int k = 0; k < PC; ++k) s[k] = wave_sum(wyw[(PC + k) * 64 + lane] * x);          // W^T a
#pragma unroll
        for (int k = 0; k < PC; ++k) x += wyw[k * 64 + lane] * s[k];                          // a += Y (W^T a)
        // rows that no later window touches again: everything but the top PC rows, unless this is the panel's last window
        const bool last = !(pr - step + PR > pc && pr - step >= 0);
        if (live && (last || lane >= PC)) a[pr + lane] = x;
    }
}

// Q (m x m) <- I, then Q <- Q H for every reflector in the reference's order (qr.c:353-438); thread i owns row i of Q
__global__ __launch_bounds__(256) void legacy_formq_kernel(const double* __restrict__ A, const double* __restrict__ tau, LegacyDims d,
                                                           double* __restrict__ Q)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x, m = d.m, PR = d.PR, PC = d.PC;
    if (i >= m) return;
    for (int j = 0; j < m; ++j) Q[(size_t) j * m + i] = (i == j) ? 1.0 : 0.0;
    int pcCount = 0;
    for (int pc = 0; pc < d.n; pc += PC, ++pcCount) {
        int prCount = 0;
        for (int pr = m - PR; pr + PR > pc && pr >= 0; pr -= PR - PC, ++prCount) {
            const bool bottom = (pr == m - PR), top = (pr <= pc);
            for (int c = 0; c < PC && c + pc < d.n; ++c) {
                const double t = tau[((size_t) d.rowPanels * pcCount + prCount) * PC + c];
                const int j0 = pr + (top ? pc - pr + c : c), j1 = pr + (bottom ? PR : PR - PC + c + 1);
                const double* vcol = A + (size_t) (pc + c) * m;
                double qv = Q[(size_t) j0 * m + i];                                        // v(j0) = 1
                for (int j = j0 + 1; j < j1; ++j) qv += Q[(size_t) j * m + i] * vcol[j];
                qv *= t;
                Q[(size_t) j0 * m + i] -= qv;
                for (int j = j0 + 1; j < j1; ++j) Q[(size_t) j * m + i] -= qv * vcol[j];
            }
        }
    }
}

}   // namespace

extern "C" {

// doubles of workspace for the WY pairs of one column panel
size_t qrd_legacy_ws_size(int m, int PR, int PC)
{
    return (size_t) legacy_windows(m, PR, PC, 0) * 2 * (size_t) PC * 64;
}

int qrd_legacy_shape_ok(int m, int n, int PR, int PC)
{
    return PR >= 2 && PR <= 64 && (PC == 2 || PC == 4 || PC == 8 || PC == 16) && PC < PR && m >= PR && n >= PC && n % PC == 0 && n <= m &&
           (m - PR) % (PR - PC) == 0;
}

// one column panel: windows bottom -> top, then the trailing columns
int qrd_legacy_panel(void* stream, double* A, int m, int n, int PR, int PC, int rowPanels, int pc, int pcCount, double* tau, double* wy)
{
    hipStream_t s = (hipStream_t) stream;
    if (!qrd_legacy_shape_ok(m, n, PR, PC)) return -4;
    const LegacyDims d{m, n, PR, PC, rowPanels};
    const int ncols = n - (pc + PC), grid = (ncols + 3) / 4;
    switch (PC) {
#define LEGACY_CASE(P)                                                                                                    \
    case P:                                                                                                               \
        hipLaunchKernelGGL(legacy_panel_kernel<P>, dim3(1), dim3(64), 0, s, A, d, pc, pcCount, tau, wy);                  \
        if (ncols > 0) hipLaunchKernelGGL(legacy_trailing_kernel<P>, dim3(grid), dim3(256), 0, s, A, d, pc, wy);          \
        break
        LEGACY_CASE(2); LEGACY_CASE(4); LEGACY_CASE(8); LEGACY_CASE(16);
#undef LEGACY_CASE
    default: return -4;
    }
    return (int) hipGetLastError();
}

int qrd_legacy_formq(void* stream, const double* A, const double* tau, int m, int n, int PR, int PC, int rowPanels, double* Q)
{
    if (!qrd_legacy_shape_ok(m, n, PR, PC)) return -4;
    const LegacyDims d{m, n, PR, PC, rowPanels};
    hipLaunchKernelGGL(legacy_formq_kernel, dim3((m + 255) / 256), dim3(256), 0, (hipStream_t) stream, A, tau, d, Q);
    return (int) hipGetLastError();
}

}   // extern "C"
