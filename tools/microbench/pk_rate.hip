// Issue rate of v_fma_f32 against v_pk_fma_f32 on gfx950, per SIMD, at 1 / 2 / 4 wavefronts per SIMD, with 1 / 4 / 8 independent
// accumulator chains per wavefront.  Question behind step2_kernel (two line-search step sizes per lane as float2v): does a packed fp32
// instruction issue in the slot of a plain one (then two items per lane halve the VALU time of an issue-bound kernel), and how many
// wavefronts per SIMD does it take to get there?  Prints shader cycles per instruction per SIMD (wall_clock of the whole launch / instructions
// issued on one SIMD) -- plain fp32 at full rate is 4.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x2 __attribute__((ext_vector_type(2)));
constexpr int REP = 1 << 16;

template<int CH, bool PK> __global__ __launch_bounds__(1024) void rate_kernel(float* out, long long* cyc, float seed)
{
    f32x2 acc[CH], m = {seed, seed * 1.0001f}, a = {1e-3f, 2e-3f};
    for (int c = 0; c < CH; c++) acc[c] = f32x2{(float)(threadIdx.x + c), (float)c};
    const long long t0 = clock64();
    for (int r = 0; r < REP / CH; r++) {
#pragma unroll
        for (int c = 0; c < CH; c++) {
            if (PK) acc[c] = __builtin_elementwise_fma(acc[c], m, a);
            else acc[c].x = __builtin_fmaf(acc[c].x, m.x, a.x);
        }
    }
    const long long t1 = clock64();
    float s = 0.f;
    for (int c = 0; c < CH; c++) s += acc[c].x + acc[c].y;
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)] = t1 - t0;
}

template<int CH, bool PK> static void run(int waves_per_simd, float* d_out, long long* d_cyc)
{
    const int T = 256 * waves_per_simd;   // one workgroup per CU: 4 SIMDs x waves_per_simd wavefronts
    hipLaunchKernelGGL((rate_kernel<CH, PK>), dim3(256), dim3(T), 0, 0, d_out, d_cyc, 0.999f);
    hipDeviceSynchronize();
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0, 0);
    hipLaunchKernelGGL((rate_kernel<CH, PK>), dim3(256), dim3(T), 0, 0, d_out, d_cyc, 0.999f);
    hipEventRecord(e1, 0);
    hipDeviceSynchronize();
    float ms = 0.f;
    hipEventElapsedTime(&ms, e0, e1);
    const double tflops = 256.0 * T * (double)REP * (PK ? 4.0 : 2.0) / (ms * 1e-3) * 1e-12;   // whole chip, from the launch's wall time
    std::vector<long long> c(256 * (T / 64));
    hipMemcpy(c.data(), d_cyc, c.size() * sizeof(long long), hipMemcpyDeviceToHost);
    double mean = 0;
    for (auto v : c) mean += (double)v;
    mean /= (double)c.size();
        printf("%s chains %d waves/SIMD %d: %.3f ticks per wavefront-instruction, %.3f per SIMD-instruction, %.1f TFLOP/s on the chip\n", PK ? "v_pk_fma_f32" : "v_fma_f32   ", CH,
           waves_per_simd, mean / REP, mean / REP / waves_per_simd, tflops);
}

int main()
{
    float* d_out; long long* d_cyc;
    hipMalloc(&d_out, 256 * 1024 * sizeof(float));
    hipMalloc(&d_cyc, 256 * 16 * sizeof(long long));
    for (int w : {1, 2, 4}) {
        run<1, false>(w, d_out, d_cyc); run<1, true>(w, d_out, d_cyc);
        run<4, false>(w, d_out, d_cyc); run<4, true>(w, d_out, d_cyc);
        run<8, false>(w, d_out, d_cyc); run<8, true>(w, d_out, d_cyc);
    }
    return 0;
}
