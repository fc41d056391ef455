// pyext.cpp -- the compiled Python binding of the MI355X batched SQP solver: pybind11 over the C ABI of libgato_hip.so.
//
// Replaces python/bindings.cu (PyBSQP<T, BatchSize>, :10-220, and the module registration :222-266).  The reference compiles one
// extension module per (plant, KNOT_POINTS) and one class per batch size; plant, horizon and batch are run-time arguments of the C ABI,
// so ONE extension (`gato_amd._gato_ext`) carries one class `BSQP(plant, knot_points, batch_size[, 15 scalars])`, and the
// `bsqpN{N}_{plant}` modules (gato_amd/bsqp/_module_factory.py) register `BSQP_{B}_float` as subclasses that fix the three.
// Same method names, argument meaning and result dict as the reference; errors of the C ABI surface as RuntimeError (the reference
// ignores CUDA errors in Release builds).  The GIL is released while the device works.
#include <pybind11/numpy.h>
#include <pybind11/pybind11.h>
#include <pybind11/stl.h>

#include <cstring>
#include <stdexcept>
#include <string>
#include <vector>

// -DGATO_DOUBLE: the USE_DOUBLES build of the binding (python/bindings.cu:244-252) -- module _gato_ext_f64 over libgato_hip_f64.so; every
// `float` below (numpy dtypes included) is the library's real type
#ifdef GATO_DOUBLE
#define float double
#define GATO_EXT_NAME _gato_ext_f64
#else
#define GATO_EXT_NAME _gato_ext
#endif
#include "../../include/gato_abi.h"

namespace py = pybind11;
using farray = py::array_t<float, py::array::c_style | py::array::forcecast>;

static void chk(int rc)
{
    if (rc != GATO_OK) throw std::runtime_error(std::string("libgato_hip: status ") + std::to_string(rc) + ": " + gato_last_error());
}
static int plant_id(const std::string& plant)
{
    if (plant == "indy7") return GATO_PLANT_INDY7;
    if (plant == "iiwa14") return GATO_PLANT_IIWA14;
    throw py::value_error("unknown plant '" + plant + "' (indy7 | iiwa14)");
}
static const float* need(const farray& a, size_t count, const char* what)
{
    if ((size_t)a.size() != count) throw py::value_error(std::string(what) + ": expected " + std::to_string(count) + " floats, got " + std::to_string(a.size()));
    return a.data();
}

class PyBSQP {
  public:
    PyBSQP(const std::string& plant, int knot_points, int batch) { init(plant, knot_points, batch, nullptr); }
    PyBSQP(const std::string& plant, int knot_points, int batch, float dt, uint32_t max_sqp_iters, float kkt_tol, uint32_t max_pcg_iters, float pcg_tol,
           float solve_ratio, float mu, float q_cost, float qd_cost, float u_cost, float N_cost, float q_lim_cost, float vel_lim_cost,
           float ctrl_lim_cost, float rho)
    {
        GatoParams p{dt, max_sqp_iters, kkt_tol, max_pcg_iters, pcg_tol, solve_ratio, mu, q_cost, qd_cost, u_cost, N_cost, q_lim_cost, vel_lim_cost, ctrl_lim_cost, rho};
        init(plant, knot_points, batch, &p);
    }
    ~PyBSQP() { gato_destroy(s_); }
    PyBSQP(const PyBSQP&) = delete;
    PyBSQP& operator=(const PyBSQP&) = delete;

    // PyBSQP::solve, bindings.cu:68-148: host arrays in, result dict out (same keys, dtypes and shapes)
    py::dict solve(farray xu_traj_batch, float timestep, farray x_s_batch, farray reference_traj_batch)
    {
        const size_t B = B_, T = traj_;
        const float* xu_in = need(xu_traj_batch, B * T, "xu_traj_batch");
        const float* xs = need(x_s_batch, B * nx_, "x_s_batch");
        const float* ref = need(reference_traj_batch, B * 6 * N_, "reference_traj_batch");
        py::array_t<float> XU({(py::ssize_t)B, (py::ssize_t)T});
        std::memcpy(XU.mutable_data(), xu_in, B * T * sizeof(float));
        double t_us = 0.0;
        uint32_t iters = 0, ls = 0;
        py::array_t<int32_t> sqp_iters((py::ssize_t)B), conv((py::ssize_t)B);
        py::array_t<float> fm((py::ssize_t)B), im((py::ssize_t)B);
        std::vector<int32_t> pcg;
        std::vector<float> mm, ss;
        {
            py::gil_scoped_release nogil;
            chk(gato_solve(s_, XU.mutable_data(), timestep, xs, ref, &t_us));
            chk(gato_get_counts(s_, &iters, &ls));
            chk(gato_get_sqp_iters(s_, sqp_iters.mutable_data()));
            chk(gato_get_kkt_converged(s_, conv.mutable_data()));
            chk(gato_get_final_merit(s_, fm.mutable_data()));
            chk(gato_get_initial_merit(s_, im.mutable_data()));
            pcg.resize((size_t)(iters ? iters : 1) * B);
            mm.resize((size_t)(ls ? ls : 1) * B);
            ss.resize((size_t)(ls ? ls : 1) * B);
            chk(gato_get_pcg_iters(s_, pcg.data()));
            chk(gato_get_ls_min_merit(s_, mm.data()));
            chk(gato_get_ls_step_size(s_, ss.data()));
        }
        py::dict r;
        r["XU"] = XU;
        r["sqp_time_us"] = t_us;
        r["sqp_iters"] = sqp_iters;
        r["kkt_converged"] = conv;
        r["final_merit"] = fm;
        r["initial_merit"] = im;
        r["ls_num_iters"] = (int)ls;
        // per-iteration statistics as (line searches, B); the reference keeps one more PCG record than line searches when the
        // solve_ratio exit fires and truncates it here too (bsqp.cuh:139,165 vs bindings.cu:111-128)
        py::array_t<float> pt((py::ssize_t)ls);
        std::memset(pt.mutable_data(), 0, ls * sizeof(float));  // pcg_times_us is never filled by the reference either (bsqp.cuh:138)
        py::array_t<int32_t> pi({(py::ssize_t)ls, (py::ssize_t)B});
        py::array_t<float> lm({(py::ssize_t)ls, (py::ssize_t)B}), lst({(py::ssize_t)ls, (py::ssize_t)B});
        if (ls) {
            std::memcpy(pi.mutable_data(), pcg.data(), (size_t)ls * B * sizeof(int32_t));
            std::memcpy(lm.mutable_data(), mm.data(), (size_t)ls * B * sizeof(float));
            std::memcpy(lst.mutable_data(), ss.data(), (size_t)ls * B * sizeof(float));
        }
        r["pcg_times_us"] = pt;
        r["pcg_iters"] = pi;
        r["ls_min_merit"] = lm;
        r["ls_step_size"] = lst;
        // beyond the reference's keys: every executed iteration's PCG counts (incl. the one the early exit cut off)
        py::array_t<int32_t> pa({(py::ssize_t)iters, (py::ssize_t)B});
        if (iters) std::memcpy(pa.mutable_data(), pcg.data(), (size_t)iters * B * sizeof(int32_t));
        r["pcg_iters_all"] = pa;
        r["iters_done"] = (int)iters;
        return r;
    }

    void set_f_ext_batch(farray a) { chk(gato_set_f_ext_batch(s_, need(a, (size_t)B_ * 6, "f_ext_batch"))); }
    void set_rho_penalty_batch(farray a, bool set_as_reset_default) { chk(gato_set_rho_penalty_batch(s_, need(a, B_, "rho_batch"), set_as_reset_default)); }
    void set_drho_batch(farray a, bool set_as_reset_default) { chk(gato_set_drho_batch(s_, need(a, B_, "drho_batch"), set_as_reset_default)); }
    void set_mu_batch(farray a) { chk(gato_set_mu_batch(s_, need(a, B_, "mu_batch"))); }
    void set_pcg_tol_batch(farray a) { chk(gato_set_pcg_tol_batch(s_, need(a, B_, "pcg_tol_batch"))); }
    void set_cost_weights_batch(farray a) { chk(gato_set_cost_weights_batch(s_, need(a, (size_t)B_ * 7, "cost_weights_batch"))); }
    void reset_dual() { chk(gato_reset_dual(s_)); }
    void reset_rho() { chk(gato_reset_rho(s_)); }
    void set_rho_adaptation(bool enabled) { chk(gato_set_rho_adaptation(s_, enabled)); }
    // extension: "pcg" (reference) | "direct" (block-tridiagonal LU sweep on S), SURVEY.md 8(f)4
    void set_linear_solver(const std::string& mode)
    {
        if (mode != "pcg" && mode != "direct") throw py::value_error("linear solver: 'pcg' or 'direct'");
        chk(gato_set_linear_solver(s_, mode == "direct" ? GATO_LINSOLVE_DIRECT : GATO_LINSOLVE_PCG));
    }

    // PyBSQP::sim_forward, bindings.cu:180-194
    py::array_t<float> sim_forward(farray xk, farray uk, float dt)
    {
        const float* x = need(xk, nx_, "xk");
        const float* u = need(uk, nu_, "uk");
        py::array_t<float> out({(py::ssize_t)B_, (py::ssize_t)nx_});
        {
            py::gil_scoped_release nogil;
            chk(gato_sim_forward(s_, out.mutable_data(), x, u, dt));
        }
        return out;
    }
    // not in the reference's class: the facade's ee_pos goes through pinocchio there (interface.py:212-214)
    py::array_t<float> ee_pos(farray q)
    {
        if (q.size() % nq_) throw py::value_error("q: expected a multiple of nq floats");
        const int n = (int)(q.size() / nq_);
        py::array_t<float> out({(py::ssize_t)n, (py::ssize_t)3});
        chk(gato_ee_pos(s_, q.data(), n, out.mutable_data()));
        return out;
    }
    // MPC hypothesis selection in one call (mpc_controller.py:294-309): sim_forward of (x_last, u_last) under the B wrenches, the
    // distance of every outcome to the measured state and the arg-min, all on the device
    py::tuple select_best(farray x_last, farray u_last, farray x_meas, float dt)
    {
        const float* xl = need(x_last, nx_, "x_last");
        const float* ul = need(u_last, nu_, "u_last");
        const float* xm = need(x_meas, nx_, "x_meas");
        py::array_t<float> err((py::ssize_t)B_);
        int best = 0;
        {
            py::gil_scoped_release nogil;
            chk(gato_select_best(s_, xl, ul, xm, dt, &best, err.mutable_data()));
        }
        return py::make_tuple(best, err);
    }

    // the MPC loop's plant simulator (common.py:49-91 `rk4`): len(u_seq) RK4 steps of the library's own forward dynamics
    py::array_t<float> plant_rk4(farray x, farray u_seq, farray f_ext6, float sim_dt)
    {
        need(x, nx_, "x");
        need(f_ext6, 6, "f_ext6");
        if (u_seq.size() % nu_) throw py::value_error("u_seq: expected [nsteps][nu] floats");
        const int nsteps = (int)(u_seq.size() / nu_);
        py::array_t<float> out((py::ssize_t)nx_);
        std::memcpy(out.mutable_data(), x.data(), nx_ * sizeof(float));
        {
            py::gil_scoped_release nogil;
            chk(gato_plant_rk4(s_, out.mutable_data(), u_seq.data(), nsteps, f_ext6.data(), sim_dt));
        }
        return out;
    }

    // the same plant with a swinging payload pend11 = [quat xyzw | w | mass, length, damping, inertia]: (new state, new pend11)
    py::tuple plant_payload_rk4(farray x, farray pend11, farray u_seq, farray f_ext6, float sim_dt)
    {
        need(x, nx_, "x");
        need(pend11, 11, "pend11");
        need(f_ext6, 6, "f_ext6");
        if (u_seq.size() % nu_) throw py::value_error("u_seq: expected [nsteps][nu] floats");
        const int nsteps = (int)(u_seq.size() / nu_);
        py::array_t<float> out((py::ssize_t)nx_), pend((py::ssize_t)11);
        std::memcpy(out.mutable_data(), x.data(), nx_ * sizeof(float));
        std::memcpy(pend.mutable_data(), pend11.data(), 11 * sizeof(float));
        {
            py::gil_scoped_release nogil;
            chk(gato_plant_payload_rk4(s_, out.mutable_data(), pend.mutable_data(), u_seq.data(), nsteps, f_ext6.data(), sim_dt));
        }
        return py::make_tuple(out, pend);
    }

    // ---- MPC session (gato_mpc_*, include/gato_abi.h): one call per MPC step, the loop's state stays on the device
    void mpc_begin(farray x0) { chk(gato_mpc_begin(s_, need(x0, nx_, "x0"))); }
    void mpc_set_payload(py::object pend11)
    {
        if (pend11.is_none()) { chk(gato_mpc_set_payload(s_, nullptr)); return; }
        farray p = pend11.cast<farray>();
        chk(gato_mpc_set_payload(s_, need(p, 11, "pend11")));
    }
    py::array_t<float> mpc_payload()
    {
        py::array_t<float> out((py::ssize_t)7);
        chk(gato_mpc_get_payload(s_, out.mutable_data()));
        return out;
    }
    py::dict mpc_step(bool advance, bool plan, int plant_steps, float sim_dt, double steps_per_knot, py::object plant_wrench, py::object ref_window,
                      py::object hyp_world, bool select, float select_dt, bool time_solve_wall)
    {
        GatoMpcStep io;
        std::memset(&io, 0, sizeof(io));
        io.struct_size = (uint32_t)sizeof(io);
        io.phases = (advance ? GATO_MPC_ADVANCE : 0) | (plan ? GATO_MPC_PLAN : 0) | (time_solve_wall ? GATO_MPC_TIME_SOLVE : 0);
        io.plant_steps = plant_steps;
        io.sim_dt = sim_dt;
        io.steps_per_knot = steps_per_knot;
        farray fw, rw, hw;   // keep the converted arrays alive for the call
        if (!plant_wrench.is_none()) {
            fw = plant_wrench.cast<farray>();
            std::memcpy(io.plant_wrench, need(fw, 6, "plant_wrench"), 6 * sizeof(float));
        }
        if (!ref_window.is_none()) {
            rw = ref_window.cast<farray>();
            io.ref_window = need(rw, (size_t)6 * N_, "ref_window");
        }
        if (!hyp_world.is_none()) {
            hw = hyp_world.cast<farray>();
            io.hyp_world = need(hw, (size_t)6 * B_, "hyp_world");
        }
        io.select = select ? 1 : 0;
        io.select_dt = select_dt;
        py::array_t<float> err((py::ssize_t)B_);
        io.errors = err.mutable_data();
        {
            py::gil_scoped_release nogil;
            chk(gato_mpc_step(s_, &io));
        }
        py::array_t<float> x((py::ssize_t)nx_), ee((py::ssize_t)3);
        std::memcpy(x.mutable_data(), io.x, nx_ * sizeof(float));
        std::memcpy(ee.mutable_data(), io.ee, 3 * sizeof(float));
        py::dict r;
        r["x"] = x;
        r["ee"] = ee;
        r["best"] = (int)io.best;
        r["solve_us"] = io.solve_us;
        r["solve_wall_us"] = io.solve_wall_us;
        r["plant_us"] = io.plant_us;
        r["errors"] = err;
        return r;
    }
    py::array_t<float> mpc_best()
    {
        py::array_t<float> out((py::ssize_t)traj_);
        chk(gato_mpc_get_best(s_, out.mutable_data()));
        return out;
    }
    // the statistics of the last solve without another solve (the MPC session solves inside gato_mpc_step)
    py::dict last_stats()
    {
        uint32_t iters = 0, ls = 0;
        chk(gato_get_counts(s_, &iters, &ls));
        py::array_t<int32_t> sqp_iters((py::ssize_t)B_);
        chk(gato_get_sqp_iters(s_, sqp_iters.mutable_data()));
        std::vector<int32_t> pcg((size_t)(iters ? iters : 1) * B_);
        chk(gato_get_pcg_iters(s_, pcg.data()));
        py::array_t<int32_t> pa({(py::ssize_t)iters, (py::ssize_t)B_});
        if (iters) std::memcpy(pa.mutable_data(), pcg.data(), (size_t)iters * B_ * sizeof(int32_t));
        py::dict r;
        r["sqp_iters"] = sqp_iters;
        r["pcg_iters_all"] = pa;
        r["iters_done"] = (int)iters;
        r["ls_num_iters"] = (int)ls;
        return r;
    }

    int knot_points() const { return N_; }
    int batch_size() const { return B_; }
    std::string plant() const { return plant_; }
    uintptr_t handle() const { return reinterpret_cast<uintptr_t>(s_); }

  private:
    void init(const std::string& plant, int knot_points, int batch, const GatoParams* params)
    {
        GatoParams p;
        gato_default_params(&p);
        if (params) p = *params;
        const int pid = plant_id(plant);
        if (gato_dims(pid, knot_points, &nq_, &nx_, &nu_, &traj_) != GATO_OK) throw py::value_error(gato_last_error());
        const int rc = gato_create(pid, knot_points, batch, &p, &s_);
        if (rc == GATO_ERR_INVALID) throw py::value_error(gato_last_error());
        chk(rc);
        plant_ = plant; N_ = knot_points; B_ = batch;
    }
    GatoSolver* s_ = nullptr;
    std::string plant_;
    int N_ = 0, B_ = 0, nq_ = 0, nx_ = 0, nu_ = 0, traj_ = 0;
};

PYBIND11_MODULE(GATO_EXT_NAME, m)
{
    m.doc() = "MI355X-native batched SQP solver (pybind11 over the C ABI of libgato_hip.so); replaces python/bindings.cu";
    m.attr("version") = gato_version();
    if (gato_abi_version() != GATO_ABI_VERSION)
        throw std::runtime_error("libgato_hip: ABI version " + std::to_string(gato_abi_version()) + ", this module was compiled against " + std::to_string(GATO_ABI_VERSION));
    if (gato_abi_real_size() != (int)sizeof(float))   // `float` is this module's real type (see the top of the file)
        throw std::runtime_error("the library this extension is linked to carries another real type (-DGATO_DOUBLE goes with libgato_hip_f64.so)");
    // world placements (R [nq,3,3], p [nq,3], float64) of the joint frames from the library's kinematic tables -- pinocchio's
    // data.oMi[1..nq] in MPC_GATO.transform_force_to_gato_frame (mpc_controller.py:311-338); host-only
    m.def("fk_placements", [](const std::string& plant, farray q) {
        const int pid = plant_id(plant);
        int nq = 0;
        chk(gato_dims(pid, 8, &nq, nullptr, nullptr, nullptr));
        need(q, nq, "q");
        std::vector<double> buf((size_t)nq * 12);
        chk(gato_fk_placements(pid, q.data(), buf.data()));
        py::array_t<double> R({(py::ssize_t)nq, (py::ssize_t)3, (py::ssize_t)3}), p({(py::ssize_t)nq, (py::ssize_t)3});
        for (int k = 0; k < nq; k++) {
            std::memcpy(R.mutable_data() + 9 * k, buf.data() + 12 * k, 9 * sizeof(double));
            std::memcpy(p.mutable_data() + 3 * k, buf.data() + 12 * k + 9, 3 * sizeof(double));
        }
        return py::make_tuple(R, p);
    });
    py::class_<PyBSQP>(m, "BSQP", py::module_local())
        .def(py::init<const std::string&, int, int>(), py::arg("plant"), py::arg("knot_points"), py::arg("batch_size"))
        .def(py::init<const std::string&, int, int, float, uint32_t, float, uint32_t, float, float, float, float, float, float, float, float, float, float, float>(),
             py::arg("plant"), py::arg("knot_points"), py::arg("batch_size"), py::arg("dt"), py::arg("max_sqp_iters"), py::arg("kkt_tol"),
             py::arg("max_pcg_iters"), py::arg("pcg_tol"), py::arg("solve_ratio"), py::arg("mu"), py::arg("q_cost"), py::arg("qd_cost"), py::arg("u_cost"),
             py::arg("N_cost"), py::arg("q_lim_cost"), py::arg("vel_lim_cost"), py::arg("ctrl_lim_cost"), py::arg("rho"))
        .def("solve", &PyBSQP::solve)
        .def("reset_dual", &PyBSQP::reset_dual)
        .def("set_f_ext_batch", &PyBSQP::set_f_ext_batch)
        .def("set_rho_penalty_batch", &PyBSQP::set_rho_penalty_batch, py::arg("rho_batch"), py::arg("set_as_reset_default") = true)
        .def("set_drho_batch", &PyBSQP::set_drho_batch, py::arg("drho_batch"), py::arg("set_as_reset_default") = true)
        .def("set_mu_batch", &PyBSQP::set_mu_batch)
        .def("set_pcg_tol_batch", &PyBSQP::set_pcg_tol_batch)
        .def("set_cost_weights_batch", &PyBSQP::set_cost_weights_batch)
        .def("sim_forward", &PyBSQP::sim_forward)
        .def("reset_rho", &PyBSQP::reset_rho)
        .def("set_rho_adaptation", &PyBSQP::set_rho_adaptation)
        .def("set_linear_solver", &PyBSQP::set_linear_solver)
        .def("ee_pos", &PyBSQP::ee_pos)
        .def("select_best", &PyBSQP::select_best)
        .def("plant_rk4", &PyBSQP::plant_rk4)
        .def("plant_payload_rk4", &PyBSQP::plant_payload_rk4)
        .def("mpc_set_payload", &PyBSQP::mpc_set_payload)
        .def("mpc_payload", &PyBSQP::mpc_payload)
        .def("mpc_begin", &PyBSQP::mpc_begin)
        .def("mpc_step", &PyBSQP::mpc_step, py::arg("advance"), py::arg("plan"), py::arg("plant_steps"), py::arg("sim_dt"), py::arg("steps_per_knot"),
             py::arg("plant_wrench"), py::arg("ref_window"), py::arg("hyp_world"), py::arg("select"), py::arg("select_dt"), py::arg("time_solve_wall") = false)
        .def("mpc_best", &PyBSQP::mpc_best)
        .def("last_stats", &PyBSQP::last_stats)
        .def_property_readonly("knot_points", &PyBSQP::knot_points)
        .def_property_readonly("batch_size", &PyBSQP::batch_size)
        .def_property_readonly("plant", &PyBSQP::plant)
        .def_property_readonly("handle", &PyBSQP::handle);
}
