// micro-benchmark: latency of a dependent chain of wave-wide sums (DPP) on one wavefront
#include <hip/hip_runtime.h>
#include <cstdio>
__device__ __forceinline__ float wave_sum_dpp(float v)
{
#define A(ctrl, rmask) v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), ctrl, rmask, 0xf, false))
    A(0xB1, 0xf); A(0x4E, 0xf); A(0x141, 0xf); A(0x140, 0xf); A(0x142, 0xa); A(0x143, 0xc);
#undef A
    return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 63));
}
__device__ __forceinline__ float wave_sum_rl(float v)   // 4 DPP steps to row sums, then 4 readlanes + 3 adds
{
#define A(ctrl) v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), ctrl, 0xf, 0xf, false))
    A(0xB1); A(0x4E); A(0x141); A(0x140);
#undef A
    const float a = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 0));
    const float b = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 16));
    const float c = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 32));
    const float d = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 48));
    return (a + b) + (c + d);
}
__device__ __forceinline__ float wave_sum_shfl(float v)
{
    for (int o = 32; o >= 1; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
template<int MODE> __global__ __launch_bounds__(64) void k(float* out, int n)
{
    float v = threadIdx.x * 0.001f + 1.0f, acc = 0.f;
    for (int i = 0; i < n; i++) {
        float s = MODE == 0 ? wave_sum_dpp(v) : (MODE == 1 ? wave_sum_rl(v) : wave_sum_shfl(v));
        v = v * 0.5f + s * 1e-3f;   // dependent chain
        acc += s;
    }
    out[blockIdx.x * 64 + threadIdx.x] = acc + v;
}
template<int MODE> void run(const char* name)
{
    float* out; hipMalloc(&out, 1024 * 64 * 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int n = 2000;
    for (int grid : {1, 256, 1024}) {
        hipLaunchKernelGGL(k<MODE>, dim3(grid), dim3(64), 0, 0, out, n); hipDeviceSynchronize();
        hipEventRecord(e0); hipLaunchKernelGGL(k<MODE>, dim3(grid), dim3(64), 0, 0, out, n); hipEventRecord(e1); hipDeviceSynchronize();
        float ms; hipEventElapsedTime(&ms, e0, e1);
        printf("%-28s grid %4d: %7.1f ns per reduction\n", name, grid, ms * 1e6 / n);
    }
    hipFree(out);
}
int main() { run<0>("6 DPP + readlane"); run<1>("4 DPP + 4 readlane + adds"); run<2>("__shfl_xor butterfly"); return 0; }
