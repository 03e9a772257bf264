// probe: does hipExtAnyOrderLaunch let two kernels of ONE stream overlap on gfx950?  (hip_ext.h says "not supported on GFX9xx")
// A = long one-workgroup spin (~60 us), B = many-workgroup spin (~60 us per workgroup).  Serial: ~120 us; overlapped: ~60 us.
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <cstdio>
#include <chrono>
__global__ void spin(long long cycles, int* sink)
{
    const long long t0 = __builtin_amdgcn_s_memrealtime();
    while (__builtin_amdgcn_s_memrealtime() - t0 < cycles) { __builtin_amdgcn_s_sleep(4); }
    if (sink && threadIdx.x == 1000) *sink = 1;
}
int main()
{
    hipStream_t s, sm;
    hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
    int* d; hipMalloc(&d, 4);
    const long long cyc = 6000;     // s_memrealtime runs at 100 MHz: 60 us
    uint32_t mask[8] = {0xffffffffu, 0, 0, 0, 0, 0, 0, 0};
    hipExtStreamCreateWithCUMask(&sm, 8, mask);
    for (int masked = 0; masked < 2; ++masked) {
        hipStream_t st = masked ? sm : s;
        for (int mode = 0; mode < 3; ++mode) {
            double best = 1e9;
            for (int rep = 0; rep < 5; ++rep) {
                hipStreamSynchronize(st);
                auto t0 = std::chrono::steady_clock::now();
                for (int it = 0; it < 10; ++it) {
                    hipLaunchKernelGGL(spin, dim3(1), dim3(64), 0, st, 100LL, d);          // predecessor
                    hipLaunchKernelGGL(spin, dim3(24), dim3(256), 0, st, cyc, d);           // "product"
                    if (mode == 0) hipLaunchKernelGGL(spin, dim3(1), dim3(512), 0, st, cyc, d);
                    else if (mode == 1) hipExtLaunchKernelGGL(spin, dim3(1), dim3(512), 0, st, nullptr, nullptr, hipExtAnyOrderLaunch, cyc, d);
                    else { }                                                                    // product alone
                    hipLaunchKernelGGL(spin, dim3(1), dim3(64), 0, st, 100LL, d);          // successor
                }
                hipStreamSynchronize(st);
                const double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / 10;
                if (us < best) best = us;
            }
            printf("%s stream, %s: %.1f us per (pred, product 60us, recon 60us, succ) group\n", masked ? "CU-masked" : "plain",
                   mode == 0 ? "ordered   " : mode == 1 ? "any-order " : "no recon  ", best);
        }
    }
    return 0;
}
