/*
 * gato_abi.h -- C ABI of libgato_hip.so, the MI355X-native batched SQP solver.
 *
 * This is the drop-in boundary for the reference's batched-SQP path: every entry point below replaces one member of the
 * reference's `BSQP<T,BatchSize>` C++ class (gato/bsqp/bsqp.cuh) / its pybind11 wrapper `PyBSQP<T,B>` (python/bindings.cu).
 * Plain pointers and sizes only, no exceptions across the boundary: every function returns GATO_OK (0) or a negative status,
 * and gato_last_error() gives the text.  Plant and horizon are run-time parameters here (the reference bakes them in at
 * compile time: -DPLANT_* -DKNOT_POINTS, CMakeLists.txt:57-83); the batch size is run-time too (template parameter there).
 *
 * Real type: `gato_real` = float in libgato_hip.so (the reference's default `typedef float T`, gato/settings.h:7-11); compiled with
 * -DGATO_DOUBLE -- the reference's USE_DOUBLES -- the same sources give libgato_hip_f64.so with the same entry points on double
 * buffers.  VALIDATION ONLY, as in the reference (python/bindings.cu:244-252 registers double classes up to batch 128 only): the kernels'
 * register budgets are sized for 4-byte reals, the float64 build spills and runs several times slower (C3: 104 vs 5.6 ms per solve); it
 * exists as the arbiter of the fp32 path (it equals the float64 oracle to 1e-9).  A client must be compiled with the same -DGATO_DOUBLE as
 * the library it links: gato_abi_real_size() returns the library's sizeof(gato_real) for a start-up check (include/bsqp.hpp makes it).
 *
 * Layouts are the reference's (gato/utils/linalg.cuh:545-672), all gato_real, C-contiguous:
 *   xu    [B][TRAJ]      TRAJ = (nx+nu) N - nu, knot = [x_k (nx), u_k (nu)], last knot x only
 *   x_s   [B][nx]        nx = 2 nq, nu = nq  (indy7 nq = 6, iiwa14 nq = 7)
 *   ref   [B][N][6]      end-effector reference, xyz + 3 unused
 *   f_ext [B][6]         wrench on the last link, link-local frame
 *
 * Threading: one host thread per solver handle; a handle is bound to the HIP device that was current at gato_create.
 */
#ifndef GATO_ABI_H
#define GATO_ABI_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define GATO_OK 0
#define GATO_ERR_INVALID (-1)   /* bad argument (unsupported plant / N / B, null pointer) */
#define GATO_ERR_HIP (-2)       /* a HIP runtime call failed; see gato_last_error() */
#define GATO_ERR_NO_DEVICE (-3) /* no gfx950-capable device visible */

#ifdef GATO_DOUBLE
typedef double gato_real;
#else
typedef float gato_real;
#endif

#define GATO_PLANT_INDY7 0
#define GATO_PLANT_IIWA14 1

typedef struct GatoSolver GatoSolver;

/* The 15 scalars of BSQP's second constructor, in its order (gato/bsqp/bsqp.cuh:43-45; python/bindings.cu:35-56). */
typedef struct GatoParams {
    gato_real dt;
    uint32_t max_sqp_iters;
    gato_real kkt_tol; /* accepted and unused, as in the reference (bsqp.cuh:153) */
    uint32_t max_pcg_iters;
    gato_real pcg_tol;
    gato_real solve_ratio;
    gato_real mu;
    gato_real q_cost, qd_cost, u_cost, N_cost, q_lim_cost, vel_lim_cost, ctrl_lim_cost;
    gato_real rho;
} GatoParams;

/* Fills *p with the defaults of BSQP's first constructor (bsqp.cuh:24-27). */
void gato_default_params(GatoParams* p);

/* Sizes for a plant / horizon: nq, nx, nu, TRAJ.  Returns GATO_ERR_INVALID for an unknown plant. */
int gato_dims(int plant, int knot_points, int* nq, int* nx, int* nu, int* traj_size);

/* BSQP<T,B>::BSQP(...) + allocateMemory() (bsqp.cuh:43-59, 200-256).  knot_points: power of two in [4, 256] (the reference builds
 * 8..128, CMakeLists.txt:46); batch >= 1 (the reference registers 1..1024, bindings.cu:254-264). */
int gato_create(int plant, int knot_points, int batch, const GatoParams* params, GatoSolver** out);
/* ~BSQP() (bsqp.cuh:61) */
int gato_destroy(GatoSolver* s);

/* PyBSQP::solve (python/bindings.cu:68-148) with host buffers: H2D of xu, x_s, ref; BSQP::solve; D2H of xu.
 * xu is updated in place.  sqp_time_us (may be NULL) receives the host wall time of the device-synchronised SQP loop with the
 * reference's meaning (bsqp.cuh:109,185,190: copies excluded). */
int gato_solve(GatoSolver* s, gato_real* xu, gato_real timestep, const gato_real* x_s, const gato_real* ref, double* sqp_time_us);

/* BSQP::solve (bsqp.cuh:103-197) on device pointers, enqueued on `stream` (a hipStream_t, NULL = default stream) WITHOUT a host
 * synchronisation: the caller synchronises the stream before reading d_xu or calling the gato_get_* functions. */
int gato_solve_device(GatoSolver* s, gato_real* d_xu, gato_real timestep, const gato_real* d_x_s, const gato_real* d_ref, void* stream);

/* Waits for the solve in flight on the stream of the last gato_solve_device call (no-op when there is none). */
int gato_synchronize(GatoSolver* s);

/* Stream-ordered variants for callers that keep everything on one HIP stream (bench.py, the multi-GPU layer): the resets of
 * bsqp.cuh:81-87 as device-side copies enqueued on `stream`, and the final merits copied device-to-device into d_out [B]. */
int gato_reset_async(GatoSolver* s, int reset_dual, int reset_rho, void* stream);
int gato_copy_final_merit_device(GatoSolver* s, gato_real* d_out, void* stream);

/* Statistics of the last solve = the fields of SQPStats / the result dict of PyBSQP::solve (gato/types.cuh:23-59,
 * python/bindings.cu:96-145).  Synchronises the solver's last stream.
 *   iters_done    outer iterations executed (== every entry of sqp_iters)
 *   ls_num_iters  line searches executed (one less than iters_done when the solve_ratio exit fired) */
int gato_get_counts(GatoSolver* s, uint32_t* iters_done, uint32_t* ls_num_iters);
int gato_get_sqp_iters(GatoSolver* s, int32_t* out /* [B] */);
int gato_get_kkt_converged(GatoSolver* s, int32_t* out /* [B] */);
int gato_get_final_merit(GatoSolver* s, gato_real* out /* [B] */);   /* BSQP::copy_final_merit_to_host, bsqp.cuh:93-96 */
int gato_get_initial_merit(GatoSolver* s, gato_real* out /* [B] */); /* BSQP::copy_initial_merit0_to_host, bsqp.cuh:98-101 */
int gato_get_pcg_iters(GatoSolver* s, int32_t* out /* [iters_done][B] */);
int gato_get_ls_min_merit(GatoSolver* s, gato_real* out /* [ls_num_iters][B] */);
int gato_get_ls_step_size(GatoSolver* s, gato_real* out /* [ls_num_iters][B] */);

/* Setters, host arrays of length B (6 B for the wrench): bsqp.cuh:63-89. */
int gato_set_f_ext_batch(GatoSolver* s, const gato_real* f_ext);
int gato_set_rho_penalty_batch(GatoSolver* s, const gato_real* rho, int set_as_reset_default);
int gato_set_drho_batch(GatoSolver* s, const gato_real* drho, int set_as_reset_default);
int gato_set_mu_batch(GatoSolver* s, const gato_real* mu);
int gato_set_pcg_tol_batch(GatoSolver* s, const gato_real* pcg_tol);
/* EXTENSION beyond the reference API ("per-trajectory cost weights"): w[B][7] = q_cost, qd_cost, u_cost, N_cost,
 * q_lim_cost, vel_lim_cost, ctrl_lim_cost of each trajectory; generalises the scalar weights of the constructor (bsqp.cuh:344-350),
 * which every trajectory has until this is called.  A hyper-parameter sweep (gato_hparam_batch.ipynb) then is ONE batch. */
int gato_set_cost_weights_batch(GatoSolver* s, const gato_real* w);
int gato_reset_dual(GatoSolver* s);
int gato_reset_rho(GatoSolver* s);
int gato_set_rho_adaptation(GatoSolver* s, int enabled);
/* EXTENSION (the north-star's "block-tridiagonal Riccati/Schur solve"): how S lambda = gamma is solved in every SQP
 * iteration.  GATO_LINSOLVE_PCG (default) is the reference's warm-started, stair-preconditioned PCG (gato/bsqp/kernels/pcg.cuh);
 * GATO_LINSOLVE_DIRECT is a block LU sweep over the block-tridiagonal system (no preconditioner, no iteration count, lambda exact to
 * fp32 rounding; pcg_iters reports 1 and no trajectory is ever flagged converged by the "0 PCG iterations" rule). */
/* Replay the launch sequence of gato_solve (host-buffer form) as a hipGraph captured on first use (re-captured when dt, the iteration
 * count or a mode switch changes).  Off by default: measured on MI355X the kernels of a solve already run back to back, so the replay
 * changes the solve time by less than the run-to-run noise; results are bit-identical either way. */
int gato_set_graph_mode(GatoSolver* s, int enabled);
#define GATO_LINSOLVE_PCG 0
#define GATO_LINSOLVE_DIRECT 1
int gato_set_linear_solver(GatoSolver* s, int mode);

/* BSQP::sim_forward / PyBSQP::sim_forward (bsqp.cuh:91, bindings.cu:180-194): one integrator step of the SHARED (xk, uk) under the
 * B stored wrench hypotheses; xkp1 is [B][nx] on the host. */
int gato_sim_forward(GatoSolver* s, gato_real* xkp1, const gato_real* xk, const gato_real* uk, gato_real dt);
/* BSQP::sim_forward(T* d_xkp1_batch, T* d_xk, T* d_uk, T dt) itself (bsqp.cuh:91, kernel sim.cuh:14-49): device pointers
 * (d_xkp1 [B][nx], d_xk [nx], d_uk [nu]), enqueued on `stream` without a host synchronisation. */
int gato_sim_forward_device(GatoSolver* s, gato_real* d_xkp1, const gato_real* d_xk, const gato_real* d_uk, gato_real dt, void* stream);

/* Hypothesis selection of the MPC loop, MPC_GATO.evaluate_best_trajectory (python/bsqp/mpc_controller.py:294-309), in one launch:
 * sim_forward of the shared (x_last, u_last) under the B stored wrenches, errors[b] = |x_next_b - x_meas|_2 and *best = the first
 * arg-min (np.argmin).  errors ([B], host) may be NULL.  The _device form takes device pointers, is enqueued on `stream` and does not
 * synchronise; d_best is one int32. */
int gato_select_best(GatoSolver* s, const gato_real* x_last, const gato_real* u_last, const gato_real* x_meas, gato_real dt, int* best, gato_real* errors);
int gato_select_best_device(GatoSolver* s, const gato_real* d_x_last, const gato_real* d_u_last, const gato_real* d_x_meas, gato_real dt, int32_t* d_best,
                            gato_real* d_errors, void* stream);

/* ---- MPC session: the closed loop of python/bsqp/mpc_controller.py:196-242 with its state resident on the device ------------------
 * The reference's MPC step is host numpy between four solver calls (rk4 plant in pinocchio, window slide, reset_rho, solve, sim_forward +
 * argmin, broadcast of the best row).  Here the measured state x, the best trajectory and the batch iterates live on the device and ONE
 * call per MPC step enqueues everything on the solver's stream: plant RK4 over the measured interval (controls read from the best
 * trajectory by knot index) -> every row := best trajectory with first state := x, x_s := x, reference window and wrench hypotheses
 * (world frame -> last joint frame, mpc_controller.py:311-338) -> reset_rho -> BSQP::solve -> hypothesis selection (:294-309) -> best row.
 * The host sends the reference window and the hypotheses and reads back {x, end effector, best index, selection errors, solve time}. */
#define GATO_MPC_ADVANCE 1 /* the plant: `plant_steps` RK4 steps of size sim_dt from the session's state */
#define GATO_MPC_PLAN 2    /* prepare + reset_rho + solve + selection + take the best row */
#define GATO_MPC_TIME_SOLVE 4   /* with PLAN: also fill solve_wall_us (two extra host waits around the solve) */
typedef struct GatoMpcStep {
    uint32_t struct_size;    /* in: sizeof(GatoMpcStep) as the CLIENT was compiled; the library refuses any other value (a header that gained a field
                              * must not make the library write past an older client's struct) */
    /* in */
    int32_t phases;          /* GATO_MPC_ADVANCE | GATO_MPC_PLAN (a goal-driven loop decides between the two; figure-8 tracking does both at once) */
    int32_t plant_steps;     /* RK4 steps of the plant (mpc_controller.py:199-218: int(interval / sim_dt), + 1 when the remainders add up) */
    gato_real sim_dt;
    double steps_per_knot;   /* dt / sim_dt in double: step i is driven by the control of knot min(int(i / steps_per_knot), N - 1) */
    gato_real plant_wrench[6];   /* the disturbance actually acting on the last link, spatial [angular; linear], link frame */
    const gato_real* ref_window; /* [N][6] host: the reference of the N knots (PLAN) */
    const gato_real* hyp_world;  /* [B][6] host: world-frame wrench hypotheses (linear, angular), or NULL: the stored wrenches stay (PLAN) */
    int32_t select;          /* != 0: hypothesis selection over the batch (mpc_controller.py:294-309); 0: row 0 is the best */
    gato_real select_dt;     /* the interval the selection integrates over */
    /* out */
    gato_real x[16];         /* the session's state after ADVANCE (nx used) */
    gato_real ee[3];         /* end-effector position at x */
    int32_t best;            /* selected row */
    double solve_us;         /* device time of the SQP solve (hipEvents around its launches); 0 without PLAN */
    gato_real* errors;       /* [B] host or NULL: the selection's per-hypothesis errors */
    double solve_wall_us;    /* out, only with GATO_MPC_TIME_SOLVE in `phases`: HOST wall clock around the solve alone -- the stream is drained, the clock started,
                              * the solve enqueued, the stream drained again, the clock stopped: the reference's `sqp_time_us` (bsqp.cuh:109,185: host
                              * clock around the loop, device-synchronised), measured INSIDE the session on the session's own states.  The two extra
                              * waits make the step slower: a measurement mode (tools/mpc_heatmap.py --solve-wall), 0 otherwise */
    double plant_us;         /* out: device time of the plant simulation (hipEvents around its launch); 0 without ADVANCE.  SIMULATING the plant
                              * is not controller latency: a loop that feeds measured latency back (mpc_controller.py:234-236 charges the time
                              * around solver.solve only) subtracts this from the wall time of the call */
} GatoMpcStep;
/* state := x0 ([nx] host); every row and the best trajectory := warm start (x0 over the knots, zero controls: common.py:93-99); reset_dual */
int gato_mpc_begin(GatoSolver* s, const gato_real* x0);
int gato_mpc_step(GatoSolver* s, GatoMpcStep* io);
/* the session's best trajectory ([TRAJ] host) */
int gato_mpc_get_best(GatoSolver* s, gato_real* xu_best);

/* The plant of the closed MPC loop (python/bsqp/common.py:49-91 `rk4` over pinocchio's aba, stepped at 1 kHz by
 * mpc_controller.py:199-218), on the library's own forward dynamics: nsteps RK4 steps of size sim_dt from x ([nx], host, updated in
 * place) with control u_seq[step] ([nsteps][nu], host) under the constant spatial wrench f_ext6 = [angular; linear] acting on the
 * last link, expressed in that link's frame. */
int gato_plant_rk4(GatoSolver* s, gato_real* x, const gato_real* u_seq, int nsteps, const gato_real* f_ext6, gato_real sim_dt);
/* The plant carrying a swinging payload: MPC_GATO(pendulum_config=...) (python/bsqp/mpc_controller.py:44-60; _add_pendulum_to_model :340-360:
 * a pin.JointModelSpherical at the last joint frame, a bob of `mass` at (0, 0, -length) with inertia `inertia` x 1 about its centre -- 0.001
 * there; the loop drives the joint with -damping x its velocity, :472-478).  Only the SIMULATED arm carries it, the solver's model never does.
 *   pend11 = [quat x y z w | relative angular velocity (3, pendulum frame) | mass, length, damping, inertia]
 * (pinocchio's configuration / velocity layout of the spherical joint); the first seven are the state, updated in place.
 * gato_plant_payload_rk4 is gato_plant_rk4 with the payload; gato_mpc_set_payload gives the payload to the session's plant (NULL takes it
 * away), gato_mpc_get_payload reads its state. */
int gato_plant_payload_rk4(GatoSolver* s, gato_real* x, gato_real* pend11, const gato_real* u_seq, int nsteps, const gato_real* f_ext6, gato_real sim_dt);
int gato_mpc_set_payload(GatoSolver* s, const gato_real* pend11);
int gato_mpc_get_payload(GatoSolver* s, gato_real* pend7);
/* World placements of the nq joint frames (pinocchio's data.oMi[1..nq] in mpc_controller.py:311-338) from the library's own
 * kinematic tables: out[k] = {R row-major (9 doubles), p (3 doubles)}.  Host-only, no device needed. */
int gato_fk_placements(int plant, const gato_real* q, double* out);

/* End-effector positions [n][3] of n joint configurations [n][nq] (host arrays): what interface.BSQP.ee_pos obtains from
 * pinocchio in the reference (python/bsqp/interface.py:212-214), computed with the solver's own kinematics. */
int gato_ee_pos(GatoSolver* s, const gato_real* q, int n, gato_real* out);

/* ---- one batch sharded over the GPUs of a node -------------------------------------------------------------------
 * Rank r owns rows [r B, (r + 1) B) of a batch of world_size x B trajectories, with its own handle on its own device.  The only thing that
 * couples trajectories is the solved count of the exit rule (bsqp.cuh:165): with a communicator every SQP iteration carries ONE 4-byte
 * ncclAllReduce of that count on the solve's stream, so every rank takes the exit of the WHOLE batch in the same iteration, for any
 * solve_ratio (and a shard whose rows have all converged keeps stepping them while others have not, as the unsharded solver would).
 * RCCL is opened with dlopen here, never linked: a single-GPU host does not need it.
 *   gato_comm_unique_id   ncclGetUniqueId: 128 bytes rank 0 hands to the others (any transport)
 *   gato_comm_init        ncclCommInitRank on the solver's device; collective over all ranks; global_batch = world_size x B
 *                         = gato_comm_init_rank + gato_comm_confirm, for a caller with no side channel between the ranks
 *   gato_comm_init_rank   step 1: ncclCommInitRank alone -- NO collective is issued on the new communicator, so the ranks can compare their return
 *                         codes over whatever transport carried the id and ALL drop (gato_comm_destroy) an initialisation that failed on any of them
 *                         (gato_amd/sharding.py:connect does; a rank that fails here cannot leave its peers inside a collective)
 *   gato_comm_confirm     step 2, collective on the new communicator: the ranks agree on the solved-count mode (below); sharded solves on the handle
 *                         return GATO_ERR_INVALID until it has succeeded.  A disagreement fails on every rank alike and every rank drops its
 *                         communicator; any other failure (allocation, HIP, RCCL) is the failing rank's own -- its message is kept, its communicator
 *                         dropped, and the peers must gato_comm_destroy theirs
 *   gato_gather_results   ncclAllGather of `count` reals per rank on `stream`: the one data-path collective of a solve (packed iterates + merits)
 *   gato_comm_available   0 when librccl can be opened and has every entry point used here; no RCCL call is made (a side-effect-free probe: every
 *                         rank calls it before ANY rank enters gato_comm_init, so that the ranks fail together or not at all) */
int gato_comm_available(void);
int gato_comm_unique_id(char* out128);
int gato_comm_init(GatoSolver* s, const char* id128, int world_size, int rank, int64_t global_batch);
int gato_comm_init_rank(GatoSolver* s, const char* id128, int world_size, int rank, int64_t global_batch);
int gato_comm_confirm(GatoSolver* s);
int gato_comm_destroy(GatoSolver* s);
int gato_gather_results(GatoSolver* s, const gato_real* d_local, gato_real* d_all, uint64_t count, void* stream);
/* How a sharded solve learns the whole batch's solved count (the exit rule of bsqp.cuh:165 is the only coupling between the shards):
 *   GATO_COUNT_DEFERRED (default)   the solve runs speculatively as if the rule never fired, every rank counting its own rows; ONE
 *       all-reduce of the [max_sqp_iters] count vector at its end, published to the host by the device itself (pinned memory, no copy engine).
 *       Only if some iteration's whole-batch count reached batch x solve_ratio -- never on workloads whose trajectories do not converge -- the
 *       snapshot taken at the start (xu, lambda, rho, drho) is restored and the solve re-run with the per-iteration reduction: the results are
 *       those of the unsharded solver either way, bit for bit.
 *       WHEN the host looks (round 6): gato_solve_device returns as soon as the solve, the reduction and the publication are enqueued, with the
 *       verdict PENDING; it is taken -- the one host wait of a sharded solve, and the replay if the rule fired -- by the NEXT entry point on the
 *       handle that reads or changes the solver's state or the solve's results: gato_solve_device, gato_reset_async, gato_copy_final_merit_device,
 *       gato_synchronize, every gato_get_* / gato_set_* / gato_reset_* / gato_debug_*, gato_comm_*, gato_destroy.  Until then d_xu holds the speculative
 *       iterates: a caller that reads d_xu with its OWN kernels or copies calls one of those first (gato_synchronize also drains the stream;
 *       gato_copy_final_merit_device only takes the verdict).  gato_gather_results does NOT take it -- it touches nothing of the solver's -- which
 *       is what the split is for: between gato_solve_device and the next entry point the host is free, and bench.py enqueues the PREVIOUS solve's
 *       gather there, with the device busy (the verdict taken inside gato_solve_device left the device idle for every host call between two
 *       solves: 87 us per 1.7 ms solve at C4, profiles/r06_scaling_prediction.json).  gato_solve and gato_mpc_step take the verdict themselves.
 *       The wait itself is a spin on the sequence number in pinned memory (one host core per rank is busy while its device finishes the solve; the stream
 *       is queried every ~1e6 spins, so a device fault ends it with an error instead of a hang).
 *   GATO_COUNT_PER_ITERATION        one 4-byte all-reduce between the PCG launch and the step launch of every SQP iteration, no host wait
 *       (round 3's form; also what a hipGraph capture of a sharded solve uses).
 * A solve on a stream that is being CAPTURED (gato_solve_device under hipStreamBeginCapture) always counts per iteration: the deferred form's
 * host wait would invalidate the capture.  After a replay the next 8 sharded solves count per iteration (doubling up to 1024 while replays keep
 * coming): a batch whose exit rule fires on every solve pays the speculative pass once in a while, not every time.
 * Every rank of a communicator must be in the same mode: gato_comm_init checks it (and fails on every rank alike), and with a communicator
 * gato_set_solved_count_mode is COLLECTIVE -- every rank calls it, with the same mode; when the ranks turn out to disagree (or the call fails on
 * this rank) the handle keeps the mode it had, so mixed modes are never left behind.
 * gato_get_shard_stats: speculative solves run so far, and how many of them had to be replayed.
 * gato_get_solved_count_state: the mode the handle is in, the sharded solves that counted per iteration so far (because of the mode, a stream
 * capture, or the back-off after a replay -- none of them appears in gato_get_shard_stats), and how many more the back-off will take that way. */
#define GATO_COUNT_PER_ITERATION 0
#define GATO_COUNT_DEFERRED 1
int gato_set_solved_count_mode(GatoSolver* s, int mode);
int gato_get_shard_stats(GatoSolver* s, uint64_t* deferred_solves, uint64_t* replays);
int gato_get_solved_count_state(GatoSolver* s, int* mode, uint64_t* per_iteration_solves, uint32_t* per_iteration_left);
/* TEST HOOK: a shard of a global_batch-trajectory batch WITHOUT a communicator: the other shards' solved counts per SQP iteration are given
 * (global_batch = 0 ends it).  Lets a 1-GPU box check the sharded exit rule against the unsharded solve. */
int gato_debug_set_remote_solved(GatoSolver* s, const uint32_t* per_iter, int n, int64_t global_batch);

/* Debug / test access to a device buffer by name ("xu" = the solver's own copy used by gato_solve / gato_debug_stage):
 * "xu","D","Qq","Qd","Rd","q","r","c","Qqi","Qdi","Rdi","S","Pinv","gamma","lambda","dz","merit","merit_cur","rho","drho", "step".
 * Copies `count` floats to `out`; returns the buffer length in floats through *len when out == NULL.  "S" and "Pinv" are presented in
 * the reference's layout [k][row][left | main | right] (linalg.cuh:663-666) whatever the device's internal one is. */
int gato_debug_read(GatoSolver* s, const char* name, gato_real* out, uint64_t count, uint64_t* len);
int gato_debug_write(GatoSolver* s, const char* name, const gato_real* in, uint64_t count);
/* Runs ONE stage of an SQP iteration on device buffers previously filled (tests drive the stages one at a time):
 * stage: 0 merit(8 alphas) 1 kkt 2 schur(+stair) 3 pcg 4 dz 5 line-search 6 merit(1, dz ignored) 7 direct solve (instead of 3) */
int gato_debug_stage(GatoSolver* s, int stage, gato_real* xu, gato_real timestep, const gato_real* x_s, const gato_real* ref, gato_real* out);

/* Per-stage device time of the last gato_solve / gato_solve_device call when profiling was enabled (hipEvents around each
 * kernel family): out[7] = {merit, kkt, schur, pcg, dz, line_search, total} in microseconds, summed over the iterations. */
int gato_set_profiling(GatoSolver* s, int enabled);
int gato_get_stage_times_us(GatoSolver* s, double* out7);

const char* gato_last_error(void);
const char* gato_version(void);
/* sha256[:16] of the kernel sources THIS binary was built from (gato_amd/csrc/Makefile: -DGATO_SRC_HASH from tools/source_hash.py; "unknown" for a
 * build that bypassed the Makefile).  bench.py reports it next to the hash of the tree it runs from: a stale library cannot pass for a fresh one. */
const char* gato_source_hash(void);
/* Bumped whenever a struct or an entry point of this header changes shape; every binding compares it with the GATO_ABI_VERSION it was written
 * against at load time (include/bsqp.hpp, gato_amd/csrc/pyext.cpp, gato_amd/_lib.py) and refuses a library of another version. */
#define GATO_ABI_VERSION 6
int gato_abi_version(void);
/* sizeof(gato_real) of the LIBRARY: 4 for libgato_hip.so, 8 for libgato_hip_f64.so (a client compiled with the other setting must not call it further) */
int gato_abi_real_size(void);

#ifdef __cplusplus
}
#endif
#endif /* GATO_ABI_H */
