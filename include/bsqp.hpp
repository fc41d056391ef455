// bsqp.hpp -- the reference's C++ solver API `template<typename T, uint32_t BatchSize> class BSQP` (gato/bsqp/bsqp.cuh:20-197) and its
// companion structs (gato/types.cuh:13-59) as a header-only wrapper over the C ABI of libgato_hip.so, so that the reference's
// C++ example (examples/bsqp.cu:7-77) compiles against this library after its cuda* -> hip* renames.  T = float with libgato_hip.so (the reference's
// default `typedef float T`, settings.h:7-11), T = double with -DGATO_DOUBLE and libgato_hip_f64.so (its USE_DOUBLES).
//
// Plant and horizon are template/ctor arguments here instead of -D defines: BSQP<float, 16> solver(GATO_PLANT_INDY7, 16, dt, ...).
// If GATO_PLANT and KNOT_POINTS macros are defined (the reference's build convention) they are the defaults.
#pragma once
#include <type_traits>
#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <stdexcept>
#include <string>
#include <vector>

#include "gato_abi.h"

#ifndef GATO_PLANT
#if defined(PLANT_IIWA14)
#define GATO_PLANT GATO_PLANT_IIWA14
#else
#define GATO_PLANT GATO_PLANT_INDY7
#endif
#endif
#ifndef KNOT_POINTS
#define KNOT_POINTS 32
#endif

// gato/settings.h:7-11: the library-wide real type `T` (namespace sqp, pulled in by `using namespace sqp` like bsqp.cuh:18 does) -- float,
// or double under the reference's USE_DOUBLES convention, which selects the float64 library here
#if defined(USE_DOUBLES) && !defined(GATO_DOUBLE)
#error "USE_DOUBLES needs the float64 ABI: compile with -DGATO_DOUBLE (before gato_abi.h) and link libgato_hip_f64.so"
#endif
namespace sqp {
typedef gato_real T;
}
using namespace sqp;

// gato/utils/cuda.cuh:7-19: the example's error check around runtime calls (hip* after the example's cuda* -> hip* renames; the error type
// is whatever the wrapped call returns, so this header needs no HIP include of its own)
#ifndef gpuErrchk
#ifndef NDEBUG
#define gpuErrchk(ans)                                                                                           \
    {                                                                                                            \
        const auto gato_err_ = (ans);                                                                            \
        if (static_cast<int>(gato_err_) != 0) {                                                                  \
            std::fprintf(stderr, "GPUassert: error %d %s %d\n", static_cast<int>(gato_err_), __FILE__, __LINE__); \
            std::exit(static_cast<int>(gato_err_));                                                              \
        }                                                                                                        \
    }
#else
#define gpuErrchk(ans) ans
#endif
#endif

template<typename T, uint32_t BatchSize>
struct ProblemInputs {  // gato/types.cuh:13-19
    T timestep;
    T* d_x_s_batch;             // STATE_SIZE * batch_size (device)
    T* d_reference_traj_batch;  // 6 * KNOT_POINTS * batch_size (device)
    void* d_GRiD_mem;           // unused: the robot tables are compile-time constants of the kernels
};

template<uint32_t BatchSize>
struct PCGStats {  // gato/types.cuh:23-31
    double solve_time_us = 0;
    std::vector<int> num_iterations = std::vector<int>(BatchSize, 0);
    std::vector<int> converged = std::vector<int>(BatchSize, 0);
};

template<typename T, uint32_t BatchSize>
struct LineSearchStats {  // gato/types.cuh:35-42
    std::vector<T> min_merit = std::vector<T>(BatchSize, 0);
    std::vector<T> step_size = std::vector<T>(BatchSize, 0);
};

template<typename T, uint32_t BatchSize>
struct SQPStats {  // gato/types.cuh:46-59
    double solve_time_us = 0;
    std::vector<int> sqp_iterations = std::vector<int>(BatchSize, 0);
    std::vector<int> kkt_converged = std::vector<int>(BatchSize, 0);
    std::vector<PCGStats<BatchSize>> pcg_stats;
    std::vector<LineSearchStats<T, BatchSize>> line_search_stats;
};

// gato/types.cuh:63-81: pointer bundles the reference's class keeps for its own kernels.  No caller of the path touches them (the device
// buffers are the solver's, behind the C ABI, in a COMPACT layout: DESIGN.md section 2); declared so that code naming the types compiles.
template<typename T, uint32_t BatchSize>
struct KKTSystem {
    T *d_Q_batch, *d_R_batch, *d_q_batch, *d_r_batch, *d_A_batch, *d_B_batch, *d_c_batch;
};
template<typename T, uint32_t BatchSize>
struct SchurSystem {
    T *d_S_batch, *d_P_inv_batch, *d_gamma_batch;
};

template<typename T, uint32_t BatchSize>
class BSQP {
    static_assert(std::is_same<T, gato_real>::value, "T must be the library's real type: float with libgato_hip.so, double with -DGATO_DOUBLE and libgato_hip_f64.so");

  public:
    BSQP(int plant = GATO_PLANT, int knot_points = KNOT_POINTS)
    {
        GatoParams p;
        gato_default_params(&p);
        init(plant, knot_points, p);
    }
    BSQP(T dt, uint32_t max_sqp_iters, T kkt_tol, uint32_t max_pcg_iters, T pcg_tol, T solve_ratio, T mu, T q_cost, T qd_cost, T u_cost, T N_cost,
         T q_lim_cost, T vel_lim_cost, T ctrl_lim_cost, T rho, int plant = GATO_PLANT, int knot_points = KNOT_POINTS)
    {
        GatoParams p{dt, max_sqp_iters, kkt_tol, max_pcg_iters, pcg_tol, solve_ratio, mu, q_cost, qd_cost, u_cost, N_cost, q_lim_cost, vel_lim_cost, ctrl_lim_cost, rho};
        init(plant, knot_points, p);
    }
    ~BSQP() { gato_destroy(s_); }
    BSQP(const BSQP&) = delete;
    BSQP& operator=(const BSQP&) = delete;

    void set_f_ext_batch(T* h) { chk(gato_set_f_ext_batch(s_, h)); }
    void set_rho_penalty_batch(const T* h, bool set_as_reset_default = true) { chk(gato_set_rho_penalty_batch(s_, h, set_as_reset_default)); }
    void set_drho_batch(const T* h, bool set_as_reset_default = true) { chk(gato_set_drho_batch(s_, h, set_as_reset_default)); }
    void set_mu_batch(const T* h) { chk(gato_set_mu_batch(s_, h)); }
    void set_pcg_tol_batch(const T* h) { chk(gato_set_pcg_tol_batch(s_, h)); }
    // extension: per-trajectory cost weights, h[BatchSize][7] = q, qd, u, N, q_lim, vel_lim, ctrl_lim
    void set_cost_weights_batch(const T* h) { chk(gato_set_cost_weights_batch(s_, h)); }
    void reset_dual() { chk(gato_reset_dual(s_)); }
    void reset_rho() { chk(gato_reset_rho(s_)); }
    void set_rho_adaptation(bool enabled) { chk(gato_set_rho_adaptation(s_, enabled)); }
    void copy_final_merit_to_host(T* h_out) { chk(gato_get_final_merit(s_, h_out)); }
    void copy_initial_merit0_to_host(T* h_out) { chk(gato_get_initial_merit(s_, h_out)); }
    // bsqp.cuh:91 -- one integrator step of the shared (x_k, u_k) under the B stored wrenches, device pointers, default stream
    void sim_forward(T* d_xkp1_batch, T* d_xk, T* d_uk, T dt) { chk(gato_sim_forward_device(s_, d_xkp1_batch, d_xk, d_uk, dt, nullptr)); }

    // device-pointer solve, blocking like the reference's (its loop synchronises on the pageable D2H copies, bsqp.cuh:137,184)
    SQPStats<T, BatchSize> solve(T* d_xu_traj_batch, ProblemInputs<T, BatchSize> inputs)
    {
        // solve_time_us: host wall clock around the device-synchronised SQP loop, like bsqp.cuh:109,185,190
        const auto t0 = std::chrono::high_resolution_clock::now();
        chk(gato_solve_device(s_, d_xu_traj_batch, inputs.timestep, inputs.d_x_s_batch, inputs.d_reference_traj_batch, nullptr));
        chk(gato_synchronize(s_));
        const auto t1 = std::chrono::high_resolution_clock::now();
        SQPStats<T, BatchSize> st;
        st.solve_time_us = std::chrono::duration<double, std::micro>(t1 - t0).count();
        uint32_t iters = 0, ls = 0;
        chk(gato_get_counts(s_, &iters, &ls));
        chk(gato_get_sqp_iters(s_, st.sqp_iterations.data()));
        chk(gato_get_kkt_converged(s_, st.kkt_converged.data()));
        std::vector<int32_t> pcg((size_t)(iters ? iters : 1) * BatchSize);
        std::vector<T> mm((size_t)(ls ? ls : 1) * BatchSize), ss((size_t)(ls ? ls : 1) * BatchSize);
        chk(gato_get_pcg_iters(s_, pcg.data()));
        chk(gato_get_ls_min_merit(s_, mm.data()));
        chk(gato_get_ls_step_size(s_, ss.data()));
        for (uint32_t i = 0; i < iters; i++) {
            PCGStats<BatchSize> ps;
            for (uint32_t b = 0; b < BatchSize; b++) ps.num_iterations[b] = pcg[(size_t)i * BatchSize + b];
            st.pcg_stats.push_back(ps);
        }
        for (uint32_t i = 0; i < ls; i++) {
            LineSearchStats<T, BatchSize> l;
            for (uint32_t b = 0; b < BatchSize; b++) {
                l.min_merit[b] = mm[(size_t)i * BatchSize + b];
                l.step_size[b] = ss[(size_t)i * BatchSize + b];
            }
            st.line_search_stats.push_back(l);
        }
        return st;
    }

    GatoSolver* handle() { return s_; }

  private:
    void init(int plant, int knot_points, const GatoParams& p)
    {
        // the library linked must carry this translation unit's real type (libgato_hip.so: float, libgato_hip_f64.so with -DGATO_DOUBLE: double)
        if (gato_abi_version() != GATO_ABI_VERSION) throw std::runtime_error("libgato_hip: the linked library has another ABI version than this header");
        if (gato_abi_real_size() != (int)sizeof(T)) throw std::runtime_error("libgato_hip: the linked library's real type is not sizeof(T) -- -DGATO_DOUBLE goes with libgato_hip_f64.so");
        chk(gato_create(plant, knot_points, (int)BatchSize, &p, &s_));
    }
    static void chk(int rc)
    {
        if (rc != GATO_OK) throw std::runtime_error(std::string("libgato_hip: ") + gato_last_error());
    }
    GatoSolver* s_ = nullptr;
};
