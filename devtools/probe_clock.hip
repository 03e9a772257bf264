// probe_clock.hip -- shader clock next to a running workload: one wave per sample reads s_memtime (shader-clock counter) and
// s_memrealtime (100 MHz) before and after sleeping, on a stream of its own; the caller runs its workload on other streams.
//   hipcc --offload-arch=gfx950 -shared -fPIC -o devtools/probe_clock.so devtools/probe_clock.hip
#include <hip/hip_runtime.h>
__global__ void clock_sample_kernel(unsigned long long* out, int sleeps)
{
    const unsigned long long c0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int i = 0; i < sleeps; ++i) __builtin_amdgcn_s_sleep(127);
    const unsigned long long c1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    if (threadIdx.x == 0) { out[0] = c1 - c0; out[1] = r1 - r0; out[2] = r0; }
}
extern "C" int clock_probe_launch(void** stream_out, void** dev_out, int nsamples, int sleeps)
{
    hipStream_t st;
    if (hipStreamCreateWithFlags(&st, hipStreamNonBlocking) != hipSuccess) return 1;
    unsigned long long* d = nullptr;
    if (hipMalloc(&d, sizeof(unsigned long long) * 3 * nsamples) != hipSuccess) return 2;
    for (int i = 0; i < nsamples; ++i) hipLaunchKernelGGL(clock_sample_kernel, dim3(1), dim3(64), 0, st, d + 3 * i, sleeps);
    *stream_out = st; *dev_out = d;
    return (int) hipGetLastError();
}
extern "C" int clock_probe_collect(void* stream, void* dev, int nsamples, unsigned long long* host_out)
{
    if (hipStreamSynchronize((hipStream_t) stream) != hipSuccess) return 1;
    if (hipMemcpy(host_out, dev, sizeof(unsigned long long) * 3 * nsamples, hipMemcpyDeviceToHost) != hipSuccess) return 2;
    hipFree(dev); hipStreamDestroy((hipStream_t) stream);
    return 0;
}
