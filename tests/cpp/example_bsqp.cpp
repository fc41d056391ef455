// The reference's C++ usage pattern (examples/bsqp.cu:7-77: construct BSQP<T, B>, device buffers for the reference window, x_s and
// the warm start, ProblemInputs, solve, copy the iterates back) against include/bsqp.hpp + libgato_hip.so, with hip* in place of
// cuda* and NON-ZERO costs (the reference's example passes zeros, which makes Q singular).  Writes the iterates and the result of
// BSQP::sim_forward (bsqp.cuh:91) to argv[1] so tests/test_cpp_api.py can compare them bit for bit with the Python path.
#include <hip/hip_runtime.h>

#include <cstdio>
#include <vector>

#include "bsqp.hpp"

#ifdef GATO_DOUBLE
typedef double T;   // the reference's USE_DOUBLES build: link with -lgato_hip_f64
#else
typedef float T;
#endif
#define CHECK(x)                                                                   \
    do {                                                                           \
        hipError_t e_ = (x);                                                       \
        if (e_ != hipSuccess) { std::fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 2; } \
    } while (0)

int main(int argc, char** argv)
{
    constexpr uint32_t BatchSize = 16, N = 16, NX = 12, NU = 6;
    constexpr uint32_t TRAJ = (NX + NU) * (N - 1) + NX;
    const T dt = 0.03f;
    BSQP<T, BatchSize> bsqp(dt, 4u, 1e-3f, 100u, 1e-4f, 1.0f, 10.0f, /*q*/ 2.0f, /*qd*/ 1e-2f, /*u*/ 2e-6f, /*N*/ 50.0f, /*q_lim*/ 0.01f, 0.0f, 0.0f,
                            /*rho*/ 0.01f, GATO_PLANT_INDY7, (int)N);

    // reference window: a slow line in the workspace, different per trajectory
    std::vector<T> ref(6 * N * BatchSize, 0.f);
    for (uint32_t b = 0; b < BatchSize; b++)
        for (uint32_t k = 0; k < N; k++) {
            ref[(b * N + k) * 6 + 0] = 0.30f + 0.005f * k + 0.01f * b;
            ref[(b * N + k) * 6 + 1] = 0.35f - 0.002f * k;
            ref[(b * N + k) * 6 + 2] = 0.80f - 0.004f * k;
        }
    const T x0[NX] = {-1.0f, -0.1f, 0.8f, -0.1f, 0.5f, 0.01f, 0, 0, 0, 0, 0, 0};
    std::vector<T> xs(NX * BatchSize), xu((size_t)TRAJ * BatchSize, 0.f);
    for (uint32_t b = 0; b < BatchSize; b++) {
        for (uint32_t i = 0; i < NX; i++) xs[b * NX + i] = x0[i] + (i < 6 ? 0.01f * b : 0.f);
        for (uint32_t k = 0; k < N; k++)
            for (uint32_t i = 0; i < NX; i++) xu[(size_t)b * TRAJ + k * (NX + NU) + i] = xs[b * NX + i];
    }
    std::vector<T> fext(6 * BatchSize, 0.f);
    for (uint32_t b = 0; b < BatchSize; b++) fext[6 * b + 2] = 0.5f * b;
    bsqp.set_f_ext_batch(fext.data());

    T *d_ref, *d_xs, *d_xu, *d_xkp1, *d_xk, *d_uk;
    CHECK(hipMalloc(&d_ref, ref.size() * sizeof(T)));
    CHECK(hipMalloc(&d_xs, xs.size() * sizeof(T)));
    CHECK(hipMalloc(&d_xu, xu.size() * sizeof(T)));
    CHECK(hipMalloc(&d_xkp1, NX * BatchSize * sizeof(T)));
    CHECK(hipMalloc(&d_xk, NX * sizeof(T)));
    CHECK(hipMalloc(&d_uk, NU * sizeof(T)));
    CHECK(hipMemcpy(d_ref, ref.data(), ref.size() * sizeof(T), hipMemcpyHostToDevice));
    CHECK(hipMemcpy(d_xs, xs.data(), xs.size() * sizeof(T), hipMemcpyHostToDevice));
    CHECK(hipMemcpy(d_xu, xu.data(), xu.size() * sizeof(T), hipMemcpyHostToDevice));

    ProblemInputs<T, BatchSize> inputs;
    inputs.timestep = dt;
    inputs.d_x_s_batch = d_xs;
    inputs.d_reference_traj_batch = d_ref;
    inputs.d_GRiD_mem = nullptr;
    SQPStats<T, BatchSize> stats = bsqp.solve(d_xu, inputs);

    std::vector<T> h_xu(xu.size()), h_next(NX * BatchSize), merit(BatchSize);
    CHECK(hipMemcpy(h_xu.data(), d_xu, h_xu.size() * sizeof(T), hipMemcpyDeviceToHost));
    bsqp.copy_final_merit_to_host(merit.data());
    const T uk[NU] = {1.f, -2.f, 0.5f, 0.1f, -0.1f, 0.05f};
    CHECK(hipMemcpy(d_xk, x0, NX * sizeof(T), hipMemcpyHostToDevice));
    CHECK(hipMemcpy(d_uk, uk, NU * sizeof(T), hipMemcpyHostToDevice));
    bsqp.sim_forward(d_xkp1, d_xk, d_uk, 0.01f);
    CHECK(hipDeviceSynchronize());
    CHECK(hipMemcpy(h_next.data(), d_xkp1, h_next.size() * sizeof(T), hipMemcpyDeviceToHost));

    std::printf("XU Traj: %.6f, %.6f, %.6f, %.6f\n", (double)h_xu[0], (double)h_xu[1], (double)h_xu[2], (double)h_xu[3]);
    std::printf("solve_time_us %.1f sqp_iterations %d line_searches %zu pcg_records %zu\n", stats.solve_time_us, stats.sqp_iterations[0],
                stats.line_search_stats.size(), stats.pcg_stats.size());
    if (argc > 1) {
        FILE* f = std::fopen(argv[1], "wb");
        if (!f) return 3;
        std::fwrite(h_xu.data(), sizeof(T), h_xu.size(), f);
        std::fwrite(merit.data(), sizeof(T), merit.size(), f);
        std::fwrite(h_next.data(), sizeof(T), h_next.size(), f);
        const double meta[4] = {stats.solve_time_us, (double)stats.sqp_iterations[0], (double)stats.line_search_stats.size(),
                                (double)stats.line_search_stats.back().step_size[0]};
        std::fwrite(meta, sizeof(double), 4, f);
        std::fclose(f);
    }
    (void)hipFree(d_ref); (void)hipFree(d_xs); (void)hipFree(d_xu); (void)hipFree(d_xkp1); (void)hipFree(d_xk); (void)hipFree(d_uk);
    return 0;
}
