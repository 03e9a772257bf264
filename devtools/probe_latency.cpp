// probe_latency.cpp -- dependent-issue latencies of the instructions the small-factor kernels are made of (one wave, s_memtime cycles).
// hipcc --offload-arch=gfx950 -O3 devtools/probe_latency.cpp -o devtools/probe_latency.bin ; run on the GPU box
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double v4d __attribute__((ext_vector_type(4)));
#define N 256
__device__ __forceinline__ double rdl(double v, int lane)
{
    return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), lane), __builtin_amdgcn_readlane(__double2loint(v), lane));
}
__device__ __forceinline__ long long tick()
{
    long long t;
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t) :: "memory");
    return t;
}
#define PIN(v) asm volatile("" : "+v"(v))
__global__ __launch_bounds__(64) void probe(double* out, long long* cyc, double seed)
{
    const int lane = threadIdx.x;
    double x = seed + lane * 1e-3, y = 1.0 + seed, z = 0.5;
    long long t0, t1;
    // 1. dependent fma chain
    t0 = tick();
    PIN(x);
#pragma unroll
    for (int i = 0; i < N; ++i) x = __builtin_fma(x, y, z);
    PIN(x);
    t1 = tick();
    if (lane == 0) cyc[0] = t1 - t0;
    // 2. dependent rsq chain
    double r = 1.5 + x * 1e-300;
    t0 = tick();
    PIN(r);
#pragma unroll
    for (int i = 0; i < N; ++i) r = __builtin_amdgcn_rsq(r) + 1.0;
    PIN(r);
    t1 = tick();
    if (lane == 0) cyc[1] = t1 - t0;          // rsq + add
    // 3. dependent rcp chain
    double q = 1.5 + r * 1e-300;
    t0 = tick();
    PIN(q);
#pragma unroll
    for (int i = 0; i < N; ++i) q = __builtin_amdgcn_rcp(q) + 1.0;
    PIN(q);
    t1 = tick();
    if (lane == 0) cyc[2] = t1 - t0;          // rcp + add
    // 4. readlane -> valu -> readlane chain
    double w = q;
    t0 = tick();
    PIN(w);
#pragma unroll
    for (int i = 0; i < N; ++i) w = rdl(w, (i * 7) & 63) * y + (double) lane;
    PIN(w);
    t1 = tick();
    if (lane == 0) cyc[3] = t1 - t0;          // 2 readlane + fma
    // 5. dependent MFMA chain (accumulator dependence)
    v4d acc = (v4d){w, 0.0, 0.0, 0.0};
    t0 = tick();
    { double pv = acc[0]; PIN(pv); acc[0] = pv; }
#pragma unroll
    for (int i = 0; i < N; ++i) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(y, z, acc, 0, 0, 0);
    { double pv = acc[0]; PIN(pv); acc[0] = pv; }
    t1 = tick();
    if (lane == 0) cyc[4] = t1 - t0;
    // 6. MFMA -> VALU -> MFMA (operand dependence through a VALU op)
    double a = acc[0] * 1e-300 + 1.0;
    t0 = tick();
    PIN(a);
#pragma unroll
    for (int i = 0; i < N; ++i) {
        v4d d = __builtin_amdgcn_mfma_f64_16x16x4f64(a, z, (v4d){0.0, 0.0, 0.0, 0.0}, 0, 0, 0);
        a = d[0] * 1e-3 + 1.0;
    }
    PIN(a);
    t1 = tick();
    if (lane == 0) cyc[5] = t1 - t0;
    // 7. independent MFMAs (4 accumulators)
    v4d b0 = acc, b1 = acc, b2 = acc, b3 = acc;
    t0 = tick();
    { double pv = b0[0]; PIN(pv); b0[0] = pv; b1[0] = pv + 1.0; b2[0] = pv + 2.0; b3[0] = pv + 3.0; }
#pragma unroll
    for (int i = 0; i < N / 4; ++i) {
        b0 = __builtin_amdgcn_mfma_f64_16x16x4f64(y, z, b0, 0, 0, 0);
        b1 = __builtin_amdgcn_mfma_f64_16x16x4f64(y, z, b1, 0, 0, 0);
        b2 = __builtin_amdgcn_mfma_f64_16x16x4f64(y, z, b2, 0, 0, 0);
        b3 = __builtin_amdgcn_mfma_f64_16x16x4f64(y, z, b3, 0, 0, 0);
    }
    { double pv = b0[0] + b1[0] + b2[0] + b3[0]; PIN(pv); b0[0] = pv; }
    t1 = tick();
    if (lane == 0) cyc[6] = t1 - t0;
    // 8. independent fmas (8 chains): issue rate
    double c[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) c[k] = a + k;
    t0 = tick();
    { double pv = c[0]; PIN(pv); c[0] = pv; }
#pragma unroll
    for (int i = 0; i < N / 8; ++i)
#pragma unroll
        for (int k = 0; k < 8; ++k) c[k] = __builtin_fma(c[k], y, z);
    { double pv = c[0] + c[1] + c[2] + c[3] + c[4] + c[5] + c[6] + c[7]; PIN(pv); c[0] = pv; }
    t1 = tick();
    if (lane == 0) cyc[7] = t1 - t0;
    // 9. cndmask chain (f64 select)
    double s = a;
    t0 = tick();
    PIN(s);
#pragma unroll
    for (int i = 0; i < N; ++i) s = (lane == (i & 63)) ? c[i & 7] : s;
    PIN(s);
    t1 = tick();
    if (lane == 0) cyc[8] = t1 - t0;
    // 10. LDS write -> read round trip (dependent)
    __shared__ double sh[64];
    double u = s;
    t0 = tick();
    PIN(u);
#pragma unroll 16
    for (int i = 0; i < N; ++i) { sh[lane] = u; __builtin_amdgcn_s_waitcnt(0xc07f); u = sh[(lane + 1) & 63] + 1.0; }
    PIN(u);
    t1 = tick();
    if (lane == 0) cyc[9] = t1 - t0;
    // 11. the shader clock itself: s_memtime ticks per 100 MHz wall-clock tick over a long dependent chain (one wave on an idle chip)
    {
        double f = u;
        const long long w0 = wall_clock64();
        t0 = tick();
        PIN(f);
        for (int rep = 0; rep < 200; ++rep)
#pragma unroll
            for (int i = 0; i < N; ++i) f = __builtin_fma(f, y, z);
        PIN(f);
        t1 = tick();
        const long long w1 = wall_clock64();
        if (lane == 0) { cyc[10] = t1 - t0; cyc[11] = w1 - w0; }
        u += f;
    }
    double tot = x + r + q + w + acc[0] + a + b0[0] + b1[1] + b2[2] + b3[3] + s + u;
#pragma unroll
    for (int k = 0; k < 8; ++k) tot += c[k];
    out[lane] = tot;
}
int main()
{
    double* out; long long* cyc;
    hipMalloc(&out, 64 * 8); hipMalloc(&cyc, 16 * 8);
    for (int rep = 0; rep < 3; ++rep) hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, out, cyc, 0.25);
    hipDeviceSynchronize();
    long long h[16]; hipMemcpy(h, cyc, sizeof h, hipMemcpyDeviceToHost);
    const char* names[10] = {"dependent v_fma_f64", "dependent v_rsq_f64 + add", "dependent v_rcp_f64 + add", "readlane x2 -> fma (dependent)",
                             "dependent MFMA f64 16x16x4 (accumulator)", "MFMA -> VALU -> MFMA operand", "independent MFMA (4 accumulators)",
                             "independent fma (8 chains): issue", "dependent f64 select (2 cndmask)", "LDS write -> wait -> read -> wait + add"};
    for (int i = 0; i < 10; ++i) printf("%-48s %8.1f cycles per iteration (s_memtime/readcyclecounter units)\n", names[i], (double) h[i] / (i == 6 ? N : (i == 7 ? N : N)));
    printf("s_memtime ticks %lld over %lld wall ticks (100 MHz): s_memtime runs at %.1f MHz; dependent fma = %.2f s_memtime ticks = %.2f ns\n", h[10], h[11],
           100.0 * (double) h[10] / (double) h[11], (double) h[10] / (200.0 * N), 10.0 * (double) h[11] / (200.0 * N));
    return 0;
}
