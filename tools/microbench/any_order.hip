// Does hipExtAnyOrderLaunch let two kernels of ONE stream run side by side on gfx950?  (hip_ext.h says "not supported on GFX9xx".)
// Two single-workgroup kernels that each spin ~200 us: serialised = ~400 us, overlapped = ~200 us.
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <cstdio>
__global__ void spin(long long cycles, int* out)
{
    const long long t0 = wall_clock64();
    while (wall_clock64() - t0 < cycles) {}
    if (out) out[0] = 1;
}
int main()
{
    hipStream_t st;
    hipStreamCreateWithFlags(&st, hipStreamNonBlocking);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    const long long cyc = 20000;   // wall_clock64 ticks at 100 MHz: 200 us
    for (int mode = 0; mode < 2; mode++) {
        for (int rep = 0; rep < 3; rep++) {
            hipEventRecord(e0, st);
            hipLaunchKernelGGL(spin, dim3(1), dim3(64), 0, st, cyc, (int*)nullptr);
            if (mode == 0) hipLaunchKernelGGL(spin, dim3(1), dim3(64), 0, st, cyc, (int*)nullptr);
            else hipExtLaunchKernelGGL(spin, dim3(1), dim3(64), 0, st, nullptr, nullptr, hipExtAnyOrderLaunch, cyc, (int*)nullptr);
            hipEventRecord(e1, st);
            hipStreamSynchronize(st);
            float ms = 0;
            hipEventElapsedTime(&ms, e0, e1);
            printf("%s: two 200-us kernels in one stream took %.1f us\n", mode ? "second launch hipExtAnyOrderLaunch" : "plain launches", ms * 1e3);
        }
    }
    return 0;
}
