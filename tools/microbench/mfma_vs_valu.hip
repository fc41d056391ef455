// What would MFMA buy the fused PCG prologue?  Its GEMM-shaped work is phi = A Q^-1, theta = phi A^T + ... : 12 x 12 x 12 products, one per
// knot, 16 knots per wavefront (4 lanes per knot, 3 rows per lane).  Two ways to form ONE such product for all 16 knots of a wavefront:
//   VALU  (what kernels.hpp does): every lane accumulates its 3 rows x 12 columns with packed FMAs, the right operand's rows broadcast
//         from LDS: 216 v_pk_fma_f32 + 36 ds_read_b128 per lane, all 16 knots at once;
//   MFMA  v_mfma_f32_16x16x4_f32: the whole wavefront forms ONE knot's product per issue chain (3 issues for K = 12, the 12 x 12 operands
//         padded to the 16 x 16 tile: 56 % of the tile is zeros), 48 issues for the 16 knots, operands fetched from LDS in the
//         instruction's layout (one value per lane and issue), results left in 4 registers per knot in the MFMA's own layout.
// Prints shader cycles per 16-knot product, one wavefront per SIMD (how the prologue runs when a trajectory is alone) and two (crowded).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
constexpr int REP = 64;

__global__ __launch_bounds__(512) void valu_kernel(float* out, long long* cyc)
{
    __shared__ __attribute__((aligned(16))) float B[16][12 * 12];   // right operands of the wavefront's 16 knots
    const int lane = threadIdx.x & 63, knot = lane >> 2;
    for (int i = threadIdx.x; i < 16 * 144; i += blockDim.x) (&B[0][0])[i] = 0.001f * (float)(i % 97);
    float a[3][12], acc[3][12];
    for (int u = 0; u < 3; u++)
        for (int j = 0; j < 12; j++) { a[u][j] = 0.01f * (float)(lane + u + j); acc[u][j] = 0.f; }
    __syncthreads();
    const long long t0 = clock64();
    for (int r = 0; r < REP; r++) {
#pragma unroll
        for (int j = 0; j < 12; j++) {
            const float4* row = reinterpret_cast<const float4*>(&B[knot][j * 12]);
#pragma unroll
            for (int c = 0; c < 3; c++) {
                const float4 v = row[c];
#pragma unroll
                for (int u = 0; u < 3; u++) {
                    f32x2 lo = {acc[u][4 * c], acc[u][4 * c + 1]}, hi = {acc[u][4 * c + 2], acc[u][4 * c + 3]};
                    lo = __builtin_elementwise_fma(f32x2{a[u][j], a[u][j]}, f32x2{v.x, v.y}, lo);
                    hi = __builtin_elementwise_fma(f32x2{a[u][j], a[u][j]}, f32x2{v.z, v.w}, hi);
                    acc[u][4 * c] = lo.x; acc[u][4 * c + 1] = lo.y; acc[u][4 * c + 2] = hi.x; acc[u][4 * c + 3] = hi.y;
                }
            }
        }
#pragma unroll
        for (int u = 0; u < 3; u++) a[u][0] += acc[u][0] * 1e-9f;   // a dependence from one product to the next, as in the prologue's chain
    }
    const long long t1 = clock64();
    float s = 0.f;
    for (int u = 0; u < 3; u++)
        for (int j = 0; j < 12; j++) s += acc[u][j];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (lane == 0) cyc[blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)] = t1 - t0;
}

__global__ __launch_bounds__(512) void mfma_kernel(float* out, long long* cyc)
{
    __shared__ __attribute__((aligned(16))) float A[16][16 * 12], B[16][12 * 16];   // per knot: A rows padded to 16, B columns padded to 16
    const int lane = threadIdx.x & 63;
    for (int i = threadIdx.x; i < 16 * 192; i += blockDim.x) { (&A[0][0])[i] = 0.001f * (float)(i % 89); (&B[0][0])[i] = 0.002f * (float)(i % 83); }
    __syncthreads();
    f32x4 acc[16];
    for (int k = 0; k < 16; k++) acc[k] = f32x4{0.f, 0.f, 0.f, 0.f};
    float bump = 0.f;
    const long long t0 = clock64();
    for (int r = 0; r < REP; r++) {
#pragma unroll
        for (int k = 0; k < 16; k++) {
#pragma unroll
            for (int kk = 0; kk < 3; kk++) {
                // operand layout of v_mfma_f32_16x16x4_f32: lane l supplies A[l % 16][4 kk + l / 16] and B[4 kk + l / 16][l % 16]
                const float av = A[k][(lane & 15) * 12 + 4 * kk + (lane >> 4)] + bump;
                const float bv = B[k][(4 * kk + (lane >> 4)) * 16 + (lane & 15)];
                acc[k] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, bv, acc[k], 0, 0, 0);
            }
        }
        bump += acc[15].x * 1e-9f;   // the same product-to-product dependence
    }
    const long long t1 = clock64();
    float s = 0.f;
    for (int k = 0; k < 16; k++) s += acc[k].x + acc[k].y + acc[k].z + acc[k].w;
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (lane == 0) cyc[blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)] = t1 - t0;
}

int main()
{
    float* out; long long* cyc;
    hipMalloc(&out, 512 * 4); hipMalloc(&cyc, 8 * 8);
    for (int waves = 4; waves <= 8; waves += 4) {   // one workgroup on one CU: 4 wavefronts = one per SIMD, 8 = two per SIMD
        for (int which = 0; which < 2; which++) {
            std::vector<long long> h(8);
            for (int rep = 0; rep < 3; rep++) {
                if (which == 0) hipLaunchKernelGGL(valu_kernel, dim3(1), dim3(64 * waves), 0, 0, out, cyc);
                else hipLaunchKernelGGL(mfma_kernel, dim3(1), dim3(64 * waves), 0, 0, out, cyc);
                hipDeviceSynchronize();
            }
            hipMemcpy(h.data(), cyc, waves * 8, hipMemcpyDeviceToHost);
            long long mx = 0;
            for (int w = 0; w < waves; w++) mx = h[w] > mx ? h[w] : mx;
            printf("%s, %d wavefront(s) per SIMD: %.0f shader cycles per 12x12x12 product of 16 knots (slowest wavefront)\n", which ? "MFMA 16x16x4 f32" : "VALU packed FMA  ",
                   waves / 4, (double)mx / REP);
        }
    }
    return 0;
}
