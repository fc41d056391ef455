// Shader clock and single-wavefront issue rates on the box: s_memtime (shader clock) against the 100 MHz wall clock, for a lone wavefront
// and for a full chip of them; cycles per dependent v_fma_f32, per independent v_pk_fma_f32, per LDS write->barrier->read round trip.
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/clock_rate tools/microbench/clock_rate.hip && /tmp/clock_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x2 __attribute__((ext_vector_type(2)));
__global__ void chain(float* out, long long* t, int n, int mode)
{
    __shared__ float lds[256];
    float a = threadIdx.x * 1e-3f, b = 1.0001f, c = 1e-4f;
    f32x2 p0 = {a, b}, p1 = {b, a}, p2 = {c, a}, p3 = {a, c}, q = {1.0001f, 0.9999f}, r = {1e-5f, 2e-5f};
    const long long w0 = wall_clock64(), c0 = clock64();
    if (mode == 0) {
        for (int i = 0; i < n; i++) {
#pragma unroll
            for (int u = 0; u < 64; u++) a = __builtin_fmaf(a, b, c);   // dependent chain
        }
    } else if (mode == 1) {
        for (int i = 0; i < n; i++) {
#pragma unroll
            for (int u = 0; u < 16; u++) {                                // 4 independent packed chains
                p0 = __builtin_elementwise_fma(p0, q, r);
                p1 = __builtin_elementwise_fma(p1, q, r);
                p2 = __builtin_elementwise_fma(p2, q, r);
                p3 = __builtin_elementwise_fma(p3, q, r);
            }
        }
        a = p0.x + p0.y + p1.x + p1.y + p2.x + p2.y + p3.x + p3.y;
    } else {
        for (int i = 0; i < n; i++) {
#pragma unroll
            for (int u = 0; u < 8; u++) {                                 // LDS write -> barrier -> read of a neighbour's value
                lds[threadIdx.x] = a;
                __syncthreads();
                a += lds[(threadIdx.x + 1) & (blockDim.x - 1)];
                __syncthreads();
            }
        }
    }
    const long long c1 = clock64(), w1 = wall_clock64();
    if (threadIdx.x == 0) { t[2 * blockIdx.x] = c1 - c0; t[2 * blockIdx.x + 1] = w1 - w0; }
    out[blockIdx.x * blockDim.x + threadIdx.x] = a;
}
int main()
{
    int wrate = 0;
    hipDeviceGetAttribute(&wrate, hipDeviceAttributeWallClockRate, 0);
    printf("wall clock rate %d kHz\n", wrate);
    float* out; long long* t;
    hipMalloc(&out, 4096 * 256 * 4); hipMalloc(&t, 4096 * 16);
    const char* names[3] = {"dependent v_fma_f32 (64 per loop trip)", "4 independent v_pk_fma_f32 chains (64 per trip)", "LDS write->barrier->read->barrier (8 per trip)"};
    const int per[3] = {64, 64, 8};
    for (int mode = 0; mode < 3; mode++)
        for (int blocks : {1, 1024, 2048}) {
            const int n = mode == 2 ? 2000 : 4000;
            for (int rep = 0; rep < 2; rep++) {
                hipLaunchKernelGGL(chain, dim3(blocks), dim3(128), 0, 0, out, t, n, mode);
                hipDeviceSynchronize();
            }
            std::vector<long long> h(2 * blocks);
            hipMemcpy(h.data(), t, 16 * blocks, hipMemcpyDeviceToHost);
            const double cyc = (double)h[0], wall_s = (double)h[1] / (wrate * 1e3);
            printf("%-52s blocks %4d x 128 thr: shader clock %.0f MHz, %.2f cycles = %.2f ns per op\n", names[mode], blocks, cyc / wall_s / 1e6,
                   cyc / ((double)n * per[mode]), wall_s * 1e9 / ((double)n * per[mode]));
        }
    return 0;
}
