// solver.hip -- host driver + C ABI (include/gato_abi.h) of the MI355X-native batched SQP solver.
//
// Replaces, for the reference's hot path, the host class BSQP<T,B> (gato/bsqp/bsqp.cuh) and what PyBSQP<T,B> does around it
// (python/bindings.cu).  One solve = a fixed sequence of kernel launches on one HIP stream with NO host round trip: the
// convergence bookkeeping and the solve_ratio early exit of bsqp.cuh:137-176 run on the device (Ctrl / num_solved).
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>   // types and enums only: the library is bound at run time (gato_comm_init), a single-GPU host does not need it

#include <dlfcn.h>

#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <type_traits>
#include <vector>

#include "real.hpp"   // GATO_DOUBLE: `float` is the real type from here on (the ABI below included)
#include "../../include/gato_abi.h"
#include "kernels.hpp"

using namespace gato;

static thread_local std::string g_err;
static int fail(int code, const std::string& msg)
{
    g_err = msg;
    return code;
}
#define HIPCHK(expr)                                                                                                       \
    do {                                                                                                                   \
        hipError_t e_ = (expr);                                                                                            \
        if (e_ != hipSuccess) return fail(GATO_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(e_));                \
    } while (0)

enum Stage { ST_MERIT = 0, ST_KKT, ST_SCHUR, ST_PCG, ST_DZ, ST_LS, ST_COUNT };

// A handle is bound to the HIP device that was current at gato_create: every ABI entry makes that device current for its
// duration and restores the caller's afterwards (a host that drives several GPUs from one thread switches devices between calls).
struct DeviceGuard {
    int prev = -1, want = -1;
    explicit DeviceGuard(int dev) : want(dev)
    {
        if (hipGetDevice(&prev) != hipSuccess) prev = -1;
        if (prev != want) (void)hipSetDevice(want);
    }
    ~DeviceGuard()
    {
        if (prev >= 0 && prev != want) (void)hipSetDevice(prev);
    }
};
#define GUARD(s) DeviceGuard guard_((s)->device)

struct GatoSolver {
    int plant, N, B, nq, nx, nu, traj, vecp, brow;
    int device;
    GatoParams p;
    Costs cw;          // the solver's scalar cost weights (defaults of every trajectory)
    float* d_costw;    // [B][8] per-trajectory weights the kernels read
    int adapt_rho;
    float* zero_slab;   // dz | pcg_iters | converged | ctrl | num_solved: zeroed by one memset at the start of a solve
    size_t zero_words;
    int fuse_schur;  // Schur complement formed inside the PCG kernel (GATO_SCHUR_FUSED, default 1)
    int fuse_step;   // dz + merit + line search in one launch (GATO_STEP_FUSED, default 1); both are read when the solver is created
    int schur_rowlane;  // stand-alone Schur kernel with one row per lane (GATO_SCHUR_ROWLANE; default: nq odd)
    // the PCG launch plan, decided ONCE when the solver is created (plan_pcg): which register-resident kernel runs (0 = the
    // streaming pcg_kernel), whether it forms the stair off-diagonals itself (then schur2_kernel is not launched) and whether the
    // Schur complement is formed inside it.  Tuning overrides GATO_PCG_VARIANT / GATO_PCG_FOLD are read there, never in the solve loop.
    int pcg_choice, pcg_fold, pcg_fused, pcg_pair;
    int cus;   // compute units of the solver's device
    int pcg_rounds;   // > 1: the PCG workgroups are scheduled hardest-first (Buffers::order), see plan_pcg
    int32_t* d_order;
    int merit_in_step_forced;   // GATO_MERIT_IN_STEP = 0 / 1, else -1
    int linear_solver;  // 0: PCG (the reference's solver, pcg.cuh), 1: direct block-tridiagonal sweep (gato_set_linear_solver)
    int direct_cr_forced;   // GATO_DIRECT_CR = 0 / 1 (read ONCE, when the solver is created), else -1: direct_uses_cr decides
    bool cr_granted;        // the cyclic-reduction kernels' LDS has been granted on THIS handle's device (gato_set_linear_solver)
    // optional hipGraph replay of the host-buffer solve (gato_set_graph_mode): the fixed launch sequence of one solve captured once per
    // (dt, iteration count, mode switches) on the solver's own stream -- the buffers of gato_solve are the solver's own, so the kernel
    // arguments never change between solves
    int graph_mode;
    hipStream_t own_stream;
    hipGraphExec_t graph_exec;
    float graph_dt; uint32_t graph_iters; int graph_adapt, graph_lin;
    float *d_sim_x, *d_sim_u, *d_sim_out;  // staging of sim_forward ([nx], [nu], [B][nx]), allocated with the solver
    float *d_sel_xm, *d_sel_err;           // gato_select_best: measured state [nx], per-hypothesis error [B]
    int32_t* d_sel_best;                   // [0] arg-min, [1] completion counter of the selection kernel
    float *d_ee_q, *d_ee_out;              // staging of ee_pos, grown on demand
    size_t ee_cap;
    float* d_plant;                        // staging of gato_plant_rk4 (x | wrench | control sequence), grown on demand
    // MPC session (gato_mpc_*): state, previous state, best trajectory, reference window, world-frame hypotheses, plant wrench, record
    float *d_mpc_x, *d_mpc_xlast, *d_mpc_best, *d_mpc_refw, *d_mpc_hyp, *d_mpc_fw, *d_mpc_rec, *d_mpc_err;
    float* d_mpc_pend;                     // the plant's swinging payload [quat | w | mass, length, damping, inertia] (gato_mpc_set_payload)
    bool mpc_payload = false;
    float *h_mpc_in = nullptr, *h_mpc_out = nullptr;   // pinned mirrors of [wrench | window | hypotheses] and [record | errors]: one copy each way per step
    hipEvent_t mpc_ev0, mpc_ev1, mpc_ev2, mpc_ev3;   // around the solve | around the plant kernel
    bool mpc_begun = false;
    // sharded batch (gato_comm_init): this solver holds rows [rank B, (rank + 1) B) of a batch of global_batch trajectories
    void* comm = nullptr;              // ncclComm_t
    int world = 1, rank = 0;
    long global_batch = 0;             // 0: not sharded
    uint32_t* d_ns_local = nullptr;    // this rank's own counts (in the per-solve slab)
    uint32_t* d_ns_remote = nullptr;   // test hook: the other shards' solved counts per iteration (gato_debug_set_remote_solved)
    // how a sharded solve learns the whole batch's solved count (gato_set_solved_count_mode):
    //   deferred (default): the solve runs SPECULATIVELY as if the exit rule never fired, counting its own rows; ONE reduction of the count
    //   vector at its end; only if some iteration's whole-batch count reached the threshold (never on workloads where nothing converges) the
    //   snapshot taken at the start is restored and the solve re-run with ...   per-iteration: ... one 4-byte reduction per SQP iteration
    int deferred_count = 1;
    float thresh_override = -1.f;      // >= 0: what exit_threshold() returns (the speculative run: +inf)
    float *d_snap_xu = nullptr, *d_snap_lambda = nullptr, *d_snap_rho = nullptr;   // snapshot of what a solve changes for good: xu | lambda | rho, drho
    uint32_t* h_counts = nullptr;      // pinned, host-coherent: [max_iters reduced counts | sequence number], WRITTEN BY THE DEVICE (publish_counts_kernel) behind
                                       // the reduction; the host spins on the sequence number: no copy engine, no interrupt between the reduction and the verdict
    uint32_t* d_h_counts = nullptr;    // the device's address of h_counts
    uint32_t verdict_seq = 0;
    // a speculative solve whose verdict has not been taken yet (round 6).  gato_solve_device returns as soon as the solve, the reduction and the
    // publication are ENQUEUED; the verdict is taken -- and the exact replay run, if the rule fired -- at the next entry point that reads or changes the
    // handle's state or the solve's results (settle()).  Between the two the host is free: bench.py enqueues the PREVIOUS solve's gather there instead
    // of with the device idle behind a host wait (profiles/r06_scaling_prediction.json: 87 -> see DESIGN section 5)
    struct { bool active = false; float* d_xu = nullptr; float dt = 0; const float* d_xs = nullptr; const float* d_ref = nullptr; hipStream_t st = nullptr; uint32_t iters = 0; } pend;
    uint64_t n_replays = 0, n_deferred = 0;   // statistics: speculative solves run / of those replayed (gato_get_shard_stats)
    uint64_t n_periter = 0;            // sharded solves that shared the count per SQP iteration instead: the mode, a capture, or the back-off after a replay
    bool comm_confirmed = false;       // the ranks have agreed on the count mode over THIS communicator (gato_comm_confirm): sharded solves refuse before
    // after a replay the next `periter_left` sharded solves count per iteration (a batch whose exit rule fires -- an MPC loop near convergence,
    // solve_ratio < 1 -- would otherwise pay a speculative pass plus a replay on every solve); the back-off doubles while replays keep coming
    // (8 .. 1024) and returns to 8 with the first speculative solve that stands.  The verdict comes from the all-reduced counts: every rank
    // takes the same decisions, so the ranks never disagree on which collectives a solve issues
    uint32_t periter_left = 0, replay_backoff = 8;
    uint32_t* d_agree = nullptr;       // [2]: the cross-rank agreement on the count mode (gato_comm_init / gato_set_solved_count_mode)
    size_t plant_cap;
    uint32_t max_iters_alloc;
    Buffers bf;
    float *d_xu_own, *d_xs_own, *d_ref_own, *d_merit_init0, *d_drho_init, *d_rho_init;
    std::vector<float> h_rho_init, h_drho_init;
    hipStream_t last_stream;
    bool last_stream_valid = false;  // a solve has been enqueued on last_stream (it may be the caller's stream)
    std::vector<void*> allocs;
    // profiling
    int profiling;
    std::vector<hipEvent_t> events;
    std::vector<int> event_stage;  // stage that ENDS at event i (event 0 = start)
    double stage_us[ST_COUNT + 1];
};

template<typename T> static int dalloc(GatoSolver* s, T** p, size_t count, bool zero = true)
{
    void* d = nullptr;
    size_t bytes = (count ? count : 1) * sizeof(T);
    HIPCHK(hipMalloc(&d, bytes));
    if (zero) HIPCHK(hipMemset(d, 0, bytes));
    s->allocs.push_back(d);
    *p = (T*)d;
    return GATO_OK;
}

extern "C" void gato_default_params(GatoParams* p)
{
    // BSQP() default constructor, bsqp.cuh:24-27
    p->dt = 0.01f; p->max_sqp_iters = 5; p->kkt_tol = 0.0001f; p->max_pcg_iters = 100; p->pcg_tol = 1e-5f; p->solve_ratio = 1.0f;
    p->mu = 10.0f; p->q_cost = 1.0f; p->qd_cost = 1e-3f; p->u_cost = 1e-6f; p->N_cost = 50.0f; p->q_lim_cost = 1e-3f;
    p->vel_lim_cost = 0.0f; p->ctrl_lim_cost = 0.0f; p->rho = 1e-3f;
}

extern "C" int gato_dims(int plant, int N, int* nq, int* nx, int* nu, int* traj)
{
    int q = plant == GATO_PLANT_INDY7 ? 6 : (plant == GATO_PLANT_IIWA14 ? 7 : 0);
    if (!q) return fail(GATO_ERR_INVALID, "unknown plant");
    if (nq) *nq = q;
    if (nx) *nx = 2 * q;
    if (nu) *nu = q;
    if (traj) *traj = 3 * q * N - q;
    return GATO_OK;
}

static int settle(GatoSolver* s);
static int plan_pcg_dispatch(GatoSolver* s);
static int sync_last(GatoSolver* s);
static int create_impl(GatoSolver* s, int plant, int N, int B, const GatoParams* params)
{
    s->plant = plant; s->N = N; s->B = B;
    gato_dims(plant, N, &s->nq, &s->nx, &s->nu, &s->traj);
    s->vecp = (N + 2) * s->nx;
    s->brow = 3 * s->nx * s->nx;
    s->p = *params;
    s->cw = Costs{params->q_cost, params->qd_cost, params->u_cost, params->N_cost, params->q_lim_cost, params->vel_lim_cost, params->ctrl_lim_cost};
    s->adapt_rho = 1;
    s->linear_solver = 0;
    s->deferred_count = getenv("GATO_SOLVED_COUNT") ? (strcmp(getenv("GATO_SOLVED_COUNT"), "periter") != 0 ? 1 : 0) : 1;
    s->direct_cr_forced = getenv("GATO_DIRECT_CR") ? (atoi(getenv("GATO_DIRECT_CR")) != 0 ? 1 : 0) : -1;
    s->cr_granted = false;
    s->graph_mode = getenv("GATO_GRAPH") ? atoi(getenv("GATO_GRAPH")) : 0;
    s->own_stream = nullptr;
    s->graph_exec = nullptr;
    s->fuse_schur = getenv("GATO_SCHUR_FUSED") ? atoi(getenv("GATO_SCHUR_FUSED")) : 1;
    s->fuse_step = getenv("GATO_STEP_FUSED") ? atoi(getenv("GATO_STEP_FUSED")) : 1;
    s->merit_in_step_forced = getenv("GATO_MERIT_IN_STEP") ? atoi(getenv("GATO_MERIT_IN_STEP")) : -1;
    s->schur_rowlane = getenv("GATO_SCHUR_ROWLANE") ? atoi(getenv("GATO_SCHUR_ROWLANE")) : (s->nq % 2);
    s->profiling = 0;
    s->last_stream = nullptr;
    memset(s->stage_us, 0, sizeof(s->stage_us));
    HIPCHK(hipGetDevice(&s->device));
    // the stream of the host-buffer entry points: a BLOCKING stream, so the few null-stream operations of the setters / resets (hipMemset,
    // hipMemcpy) stay ordered with it, while the streams of different handles do not order with each other
    HIPCHK(hipStreamCreateWithFlags(&s->own_stream, hipStreamDefault));
    const int nq = s->nq, nx = s->nx, nu = s->nu;
    const size_t BN = (size_t)B * N;
    s->max_iters_alloc = params->max_sqp_iters ? params->max_sqp_iters : 1;
    Buffers& bf = s->bf;
    memset(&bf, 0, sizeof(bf));
    int rc;
#define DA(ptr, n) if ((rc = dalloc(s, &(ptr), (n))) != GATO_OK) return rc;
    DA(bf.lambda, (size_t)B * s->vecp);
    DA(bf.rho, B); DA(bf.drho, B); DA(bf.mu, B); DA(bf.pcg_tol, B); DA(bf.f_ext, 6 * (size_t)B);
    { float* cwp = nullptr; DA(cwp, 8 * (size_t)B); s->d_costw = cwp; bf.costw = cwp; }
    DA(bf.D, BN * 3 * nq * nq); DA(bf.Qq, BN * nq * nq); DA(bf.Qd, BN * nq); DA(bf.Rd, BN * nu);
    DA(bf.q, BN * nx); DA(bf.r, BN * nu); DA(bf.c, BN * nx);
    DA(bf.Qqi, BN * nq * nq); DA(bf.Qdi, BN * nq); DA(bf.Rdi, BN * nu);
    DA(bf.S, BN * s->brow); DA(bf.Pinv, BN * s->brow); DA(bf.gamma, (size_t)B * s->vecp);  // zero padding blocks are relied upon
    {
        // everything a solve zeroes first (bsqp.cuh:112-114 + the device-side loop control) lives in ONE slab: one memset per solve
        auto up = [](size_t n) { return (n + 63) & ~(size_t)63; };  // 256-byte granules, in 4-byte words
        const size_t o_dz = 0, o_pi = up((size_t)B * s->traj), o_cv = o_pi + up(B), o_ct = o_cv + up(B), o_ns = o_ct + up(sizeof(Ctrl) / 4);
        const size_t o_nsl = o_ns + up(s->max_iters_alloc);   // this rank's own solved counts (sharded batch)
        s->zero_words = o_nsl + up(s->max_iters_alloc);
        float* slab = nullptr;
        DA(slab, s->zero_words);
        s->zero_slab = slab;
        bf.dz = slab + o_dz;
        bf.pcg_iters = reinterpret_cast<uint32_t*>(slab + o_pi);
        bf.converged = reinterpret_cast<int32_t*>(slab + o_cv);
        bf.ctrl = reinterpret_cast<Ctrl*>(slab + o_ct);
        bf.num_solved = reinterpret_cast<uint32_t*>(slab + o_ns);
        bf.num_solved_w = bf.num_solved;   // one GPU: the count the exit rule reads is the one the PCG kernels add to
        s->d_ns_local = reinterpret_cast<uint32_t*>(slab + o_nsl);
    }
    DA(bf.merit, (size_t)B * NUM_ALPHAS); DA(bf.merit_cur, B); DA(bf.step, B); DA(s->d_order, B);
    DA(bf.st_pcg_iters, (size_t)s->max_iters_alloc * B); DA(bf.st_min_merit, (size_t)s->max_iters_alloc * B);
    DA(bf.st_step, (size_t)s->max_iters_alloc * B);
    DA(s->d_xu_own, (size_t)B * s->traj); DA(s->d_xs_own, (size_t)B * nx); DA(s->d_ref_own, (size_t)B * 6 * N);
    DA(s->d_merit_init0, B); DA(s->d_drho_init, B); DA(s->d_rho_init, B);
    DA(s->d_sim_x, nx); DA(s->d_sim_u, nu); DA(s->d_sim_out, (size_t)B * nx);
    DA(s->d_sel_xm, nx); DA(s->d_sel_err, B); DA(s->d_sel_best, 2);
    DA(s->d_mpc_x, nx); DA(s->d_mpc_xlast, nx); DA(s->d_mpc_best, s->traj); DA(s->d_mpc_pend, 12);
    DA(s->d_mpc_fw, 8 + 6 * (size_t)N + 6 * (size_t)B);   // [wrench (6, padded to 8) | reference window | world-frame hypotheses], one H2D copy
    s->d_mpc_refw = s->d_mpc_fw + 8; s->d_mpc_hyp = s->d_mpc_refw + 6 * (size_t)N;
    DA(s->d_mpc_rec, 20 + (size_t)B);                         // [record (nx + 4, padded to 20) | selection errors], one D2H copy
    s->d_mpc_err = s->d_mpc_rec + 20;
#undef DA
    s->d_ee_q = s->d_ee_out = nullptr;
    s->ee_cap = 0;
    s->d_plant = nullptr;
    s->plant_cap = 0;
    s->mpc_ev0 = s->mpc_ev1 = s->mpc_ev2 = s->mpc_ev3 = nullptr;
    // per-trajectory defaults (bsqp.cuh:48-58)
    s->h_rho_init.assign(B, params->rho);
    s->h_drho_init.assign(B, 1.0f);
    std::vector<float> mu(B, params->mu), tol(B, params->pcg_tol);
    HIPCHK(hipMemcpy(bf.rho, s->h_rho_init.data(), B * sizeof(float), hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(bf.drho, s->h_drho_init.data(), B * sizeof(float), hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(s->d_drho_init, s->h_drho_init.data(), B * sizeof(float), hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(s->d_rho_init, s->h_rho_init.data(), B * sizeof(float), hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(bf.mu, mu.data(), B * sizeof(float), hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(bf.pcg_tol, tol.data(), B * sizeof(float), hipMemcpyHostToDevice));
    {
        std::vector<int32_t> ident(B);
        for (int b = 0; b < B; b++) ident[b] = b;
        HIPCHK(hipMemcpy(s->d_order, ident.data(), B * sizeof(int32_t), hipMemcpyHostToDevice));
        bf.order = s->d_order;
    }
    {
        std::vector<float> w(8 * (size_t)B, 0.f);
        for (int b = 0; b < B; b++) {
            float* r = &w[8 * (size_t)b];
            r[0] = s->cw.q_cost; r[1] = s->cw.qd_cost; r[2] = s->cw.u_cost; r[3] = s->cw.N_cost; r[4] = s->cw.q_lim_cost;
            r[5] = s->cw.vel_lim_cost; r[6] = s->cw.ctrl_lim_cost;
        }
        HIPCHK(hipMemcpy(s->d_costw, w.data(), w.size() * sizeof(float), hipMemcpyHostToDevice));
    }
    HIPCHK(hipDeviceSynchronize());
    return plan_pcg_dispatch(s);
}

extern "C" int gato_create(int plant, int N, int B, const GatoParams* params, GatoSolver** out)
{
    if (!out || !params) return fail(GATO_ERR_INVALID, "null argument");
    if (plant != GATO_PLANT_INDY7 && plant != GATO_PLANT_IIWA14) return fail(GATO_ERR_INVALID, "unknown plant");
    if (N < 4 || N > 256 || (N & (N - 1))) return fail(GATO_ERR_INVALID, "knot_points must be a power of two in [4, 256]");
    if (B < 1) return fail(GATO_ERR_INVALID, "batch must be >= 1");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0) return fail(GATO_ERR_NO_DEVICE, "no HIP device visible");
    GatoSolver* s = new GatoSolver();
    const int rc = create_impl(s, plant, N, B, params);
    if (rc != GATO_OK) {
        const std::string msg = g_err;  // gato_destroy must not clobber the reason
        gato_destroy(s);
        g_err = msg;
        return rc;
    }
    *out = s;
    return GATO_OK;
}

extern "C" int gato_destroy(GatoSolver* s)
{
    if (!s) return GATO_OK;
    GUARD(s);
    s->pend.active = false;   // a pending verdict dies with the handle (its stream is drained next; nobody will read the results)
    if (s->last_stream_valid) (void)hipStreamSynchronize(s->last_stream);
    if (s->comm) (void)gato_comm_destroy(s);
    if (s->d_ee_q) (void)hipFree(s->d_ee_q);
    if (s->d_ee_out) (void)hipFree(s->d_ee_out);
    if (s->d_plant) (void)hipFree(s->d_plant);
    if (s->graph_exec) (void)hipGraphExecDestroy(s->graph_exec);
    if (s->h_counts) (void)hipHostFree(s->h_counts);
    if (s->h_mpc_in) (void)hipHostFree(s->h_mpc_in);
    if (s->h_mpc_out) (void)hipHostFree(s->h_mpc_out);
    if (s->mpc_ev0) (void)hipEventDestroy(s->mpc_ev0);
    if (s->mpc_ev1) (void)hipEventDestroy(s->mpc_ev1);
    if (s->mpc_ev2) (void)hipEventDestroy(s->mpc_ev2);
    if (s->mpc_ev3) (void)hipEventDestroy(s->mpc_ev3);
    if (s->own_stream) (void)hipStreamDestroy(s->own_stream);
    for (void* p : s->allocs) (void)hipFree(p);
    for (hipEvent_t e : s->events) (void)hipEventDestroy(e);
    delete s;
    return GATO_OK;
}

// ---- launches ---------------------------------------------------------------------------------------------------------
static inline int cdiv(long a, long b) { return (int)((a + b - 1) / b); }
// num_solved >= BatchSize * solve_ratio (bsqp.cuh:165), BatchSize = the WHOLE batch when it is sharded over ranks
static inline float exit_threshold_exact(const GatoSolver* s) { return (float)(s->global_batch > 0 ? s->global_batch : (long)s->B) * s->p.solve_ratio; }
static inline float exit_threshold(const GatoSolver* s) { return s->thresh_override >= 0.f ? s->thresh_override : exit_threshold_exact(s); }

// out2 / zero: only for the first launch of a solve (merit of the initial iterate): second copy of the merits, slab to clear
template<class M> static void launch_merit(GatoSolver* s, hipStream_t st, int na, float dt, int use_dz, int sqp_iter, float* out, float* out2 = nullptr,
                                           float* zero = nullptr, size_t zero_words = 0)
{
    const float thresh = exit_threshold(s);
    const long n = (long)s->B * na * s->N;
    if (na == 1)
        hipLaunchKernelGGL((merit_kernel<M, 1>), dim3(cdiv(n, 256)), dim3(256), 0, st, s->bf, s->N, s->B, dt, use_dz, sqp_iter, thresh, out, out2,
                           reinterpret_cast<real4*>(zero), (uint32_t)(zero_words / 4));
    else
        hipLaunchKernelGGL((merit_kernel<M, NUM_ALPHAS>), dim3(cdiv(n, 256)), dim3(256), 0, st, s->bf, s->N, s->B, dt, use_dz, sqp_iter,
                           thresh, out, (float*)nullptr, (real4*)nullptr, 0u);
}
template<class M> static void launch_kkt(GatoSolver* s, hipStream_t st, float dt, int sqp_iter, int row0 = 0, bool clear_slab = false)
{
    // The same task split for EVERY batch size: it changes the generated code (and with it the last bit of D), so choosing it by
    // batch size would make a trajectory's iterates depend on how many neighbours it has.
    constexpr int NT = kkt_tasks<M>();
    hipLaunchKernelGGL((kkt_kernel<M>), dim3(cdiv((long)s->B * s->N, 64)), dim3(64 * NT), (size_t)64 * 3 * M::NQ * M::NQ * sizeof(float), st, s->bf, s->N,
                       s->B, dt, sqp_iter, exit_threshold(s), row0, reinterpret_cast<real4*>(clear_slab ? s->zero_slab : nullptr),
                       clear_slab ? (uint32_t)(s->zero_words / 4) : 0u);
}
// ---- the PCG launch plan ------------------------------------------------------------------------------------------------
// Register-resident kernels pcgc_kernel<M, RPT, MAXT, FOLD[, FUSE]>, tried in the order below; choice ids:
//   2: <3 rows>   3: <2 rows>   1: <6 rows>   0: streaming pcg_kernel   (round 1 also measured <2 rows, 3 waves/SIMD>, <1 row> and
//   <3 rows, 512 threads>: 172 / 213 / 151 us per launch at C2, profiles/r01d_pcg_variants.txt -- removed)
//   7: pcgs_kernel, symmetric half storage, 4 N threads (long horizons: the system stays on the CU where the others would stream it)
// Measured at indy7 N=32 B=1024 (profiles/r01d_pcg_variants.txt): 3 rows/thread 151 us, 2 rows 172 us, 1 row 213 us, 6 rows (one wave
// per trajectory) 217 us per launch.  3 rows per thread first: 256 registers without spills = two wavefronts per SIMD = four 2-wave
// trajectories per CU, so all 1024 trajectories of C2 are resident at once.
template<int NX, int RPT, int FORCE_WPS> struct PcgcShape {
    static constexpr int REGS = RPT * 6 * NX + 48;  // matrix rows + working set, per lane
    static constexpr int WPS = FORCE_WPS ? FORCE_WPS : (REGS > 256 ? 1 : (REGS > 168 ? 2 : (REGS > 128 ? 3 : 4)));
    static constexpr int MAXT = WPS * 256;          // threads per block that still leave REGS registers per lane
    static int threads(int rows) { return (((rows + RPT - 1) / RPT + 63) / 64) * 64; }
};
// the two LDS vectors of the register-resident kernels keep their blocks at a stride of nx rounded up to a multiple of 4 (pcgc_kernel: VS)
static size_t pcg_vec_lds(const GatoSolver* s) { return (size_t)(2 * (s->N + 2) * pcg_vec_stride(s->nx) + 36) * sizeof(float); }
static size_t pcg_fold_lds(const GatoSolver* s) { return (size_t)2 * s->N * s->nx * s->nx * sizeof(float); }

// Dynamic LDS beyond the 64 KB default has to be asked for, per function and device; the status is checked (a refused request
// would otherwise turn into a failed launch reported iterations later).
static bool grant_lds(const void* fn, size_t bytes)
{
    if (bytes <= 64 * 1024) return true;
    if (bytes > 160 * 1024) return false;
    return hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes) == hipSuccess;
}

template<class M, int RPT, int FORCE_WPS = 0> static bool pcgc_fits(const GatoSolver* s)
{
    constexpr int NX = 2 * M::NQ;
    if constexpr (NX % RPT != 0) {
        return false;
    } else {
        using Sh = PcgcShape<NX, RPT, FORCE_WPS>;
        return Sh::threads(s->N * s->nx) <= Sh::MAXT;
    }
}
template<class M, int RPT, int FORCE_WPS = 0> static bool pcgc_grant_fold(const GatoSolver* s)
{
    constexpr int NX = 2 * M::NQ;
    if constexpr (NX % RPT != 0) {
        return false;
    } else {
        using Sh = PcgcShape<NX, RPT, FORCE_WPS>;
        return grant_lds(reinterpret_cast<const void*>(&pcgc_kernel<M, RPT, Sh::MAXT, true>), pcg_vec_lds(s) + pcg_fold_lds(s)) &&
               grant_lds(reinterpret_cast<const void*>(&pcgc_kernel<M, RPT, Sh::MAXT, true, false, false, true>), pcg_vec_lds(s) + pcg_fold_lds(s));
    }
}
template<class M> static size_t pcg_fused_lds(const GatoSolver* s)
{
    // LDS behind the vectors: the two fold buffers [N][nx][nx], later reused to park 3 x nx/4 real4 per thread (the larger for N < 8,
    // where the workgroup is padded to one wavefront)
    constexpr int NX = 2 * M::NQ;
    const int T = PcgcShape<NX, 3, 0>::threads(s->N * s->nx);
    const size_t fold = pcg_fold_lds(s), park = (size_t)3 * (NX / 4) * T * 4 * sizeof(float);
    return pcg_vec_lds(s) + (fold > park ? fold : park);
}

// pcgs_kernel: 4 N threads; LDS = the two vectors + partial sums + row partials [N nx] + transposed partials [(N+1) 2 nx] + four parked rows
// of every thread's P^-1 block (real4 [nx][T])
static size_t pcgs_lds(const GatoSolver* s)
{
    const size_t T = 4 * (size_t)s->N, nx = s->nx;
    const size_t vecl = (size_t)(s->N + 2) * ((nx + 3) & ~(size_t)3);   // the LDS vectors' block stride (pcgs_kernel: VS)
    return ((size_t)2 * vecl + 36 + (size_t)s->N * nx + (size_t)(s->N + 1) * 2 * nx) * sizeof(float) + (4 * nx / 4) * T * sizeof(real4);
}
template<class M> static bool pcgs_grant(const GatoSolver* s)
{
    if (4 * s->N <= 256)
        return grant_lds(reinterpret_cast<const void*>(&pcgs_kernel<M, 256, false>), pcgs_lds(s)) &&
               grant_lds(reinterpret_cast<const void*>(&pcgs_kernel<M, 256, true>), pcgs_lds(s));
    return grant_lds(reinterpret_cast<const void*>(&pcgs_kernel<M, 512, false>), pcgs_lds(s)) &&
           grant_lds(reinterpret_cast<const void*>(&pcgs_kernel<M, 512, true>), pcgs_lds(s));
}
// fold: the kernel forms the stair off-diagonals itself (solve path); otherwise it reads the complete P^-1 (stage tests, GATO_PCG_FOLD=0)
template<class M> static void launch_pcgs(GatoSolver* s, hipStream_t st, int sqp_iter, bool fold)
{
    const int T = 4 * s->N;
    if (T <= 256) {
        if (fold) hipLaunchKernelGGL((pcgs_kernel<M, 256, true>), dim3(s->B), dim3(T), pcgs_lds(s), st, s->bf, s->N, s->B, s->p.max_pcg_iters, sqp_iter);
        else hipLaunchKernelGGL((pcgs_kernel<M, 256, false>), dim3(s->B), dim3(T), pcgs_lds(s), st, s->bf, s->N, s->B, s->p.max_pcg_iters, sqp_iter);
    } else {
        if (fold) hipLaunchKernelGGL((pcgs_kernel<M, 512, true>), dim3(s->B), dim3(T), pcgs_lds(s), st, s->bf, s->N, s->B, s->p.max_pcg_iters, sqp_iter);
        else hipLaunchKernelGGL((pcgs_kernel<M, 512, false>), dim3(s->B), dim3(T), pcgs_lds(s), st, s->bf, s->N, s->B, s->p.max_pcg_iters, sqp_iter);
    }
}

static bool step_fused(const GatoSolver* s);
template<class M> static int plan_pcg(GatoSolver* s)
{
    if (hipDeviceGetAttribute(&s->cus, hipDeviceAttributeMultiprocessorCount, s->device) != hipSuccess) s->cus = 0;
    constexpr int NX = 2 * M::NQ;
    const char* e = getenv("GATO_PCG_VARIANT");  // test / tuning override: 0 streaming, 1 RPT=6, 2 RPT=3, 3 RPT=2, 7 symmetric storage
    const int v = e ? atoi(e) : 100;
    const char* f = getenv("GATO_PCG_FOLD");
    const bool fold_wanted = !(f && atoi(f) == 0);
    int choice = 0;
    if ((v == 100 || v == 2) && pcgc_fits<M, 3>(s)) choice = 2;
    else if ((v == 100 || v == 3) && pcgc_fits<M, 2>(s)) choice = 3;
    else if ((v == 100 || v == 1) && pcgc_fits<M, 6>(s)) choice = 1;
    if (v == 4 && pcgc_fits<M, 1, 4>(s)) choice = 4;   // one row per thread at four wavefronts per SIMD (measured at C5; never the default)
    // symmetric half storage: asked for (7), or the default where no full-storage kernel holds the system (iiwa14 N = 128)
    // ... and ahead of the 6-rows-per-thread form, whose 432 matrix registers per lane live in AGPRs (indy7 N = 128: 296 vs 341 us per
    // one-iteration solve at B = 1, 1.49 vs 1.96 ms at B = 1024)
    if ((v == 7 || (v == 100 && (choice == 0 || choice == 1))) && s->N >= 16 && 4 * s->N <= 512 && pcgs_grant<M>(s)) choice = 7;
    s->pcg_choice = choice;
    // the kernel that will run forms the stair off-diagonals itself when the two fold buffers + the vectors fit one CU's LDS and the
    // runtime grants them; otherwise schur2_kernel is launched (launch_schur) and the kernel reads the complete P^-1
    bool fold = fold_wanted && choice != 0 && (choice == 7 || pcg_vec_lds(s) + pcg_fold_lds(s) <= 150 * 1024);  // pcgs folds inside its own LDS
    if (fold) {
        switch (choice) {
            case 2: fold = pcgc_grant_fold<M, 3>(s); break;
            case 3: fold = pcgc_grant_fold<M, 2>(s); break;
            case 1: fold = pcgc_grant_fold<M, 6>(s); break;
            case 4: fold = pcgc_grant_fold<M, 1, 4>(s); break;
            case 7: break;   // granted with the kernel (pcgs_grant)
        }
    }
    s->pcg_fold = fold ? 1 : 0;
    // Schur complement formed inside the PCG kernel (pcgc_kernel<.., FUSE>): nx = 12 (3 rows per thread = the rows of one lane of a
    // 4-lane Schur group), the trajectory in <= 256 threads, stair fold available.  GATO_SCHUR_FUSED=0 keeps the launches apart.
    bool fused = false;
    if constexpr (NX == 12) {
        fused = s->fuse_schur && fold && choice == 2 && PcgcShape<NX, 3, 0>::threads(s->N * s->nx) <= 256;
        if (fused) fused = grant_lds(reinterpret_cast<const void*>(&pcgc_kernel<M, 3, 256, true, true>), pcg_fused_lds<M>(s)) &&
                           grant_lds(reinterpret_cast<const void*>(&pcgc_kernel<M, 3, 256, true, true, false, true>), pcg_fused_lds<M>(s)) &&
                           grant_lds(reinterpret_cast<const void*>(&pcgc_kernel<M, 3, 128, true, true, false, true>), pcg_fused_lds<M>(s));
    }
    s->pcg_fused = fused ? 1 : 0;
    // Pair form of the fused kernel: two lanes per row group (twice the threads, half of the columns each; same bits).  Rounds 2-5 chose it wherever
    // the whole batch is resident in it (B <= 512 at N = 32): halving a wavefront's share of the multiply-adds shortened the iteration (0.79 vs 0.97 us).
    // Round 6 took ~300 cycles of exposed LDS latency and several branches out of the SINGLE-LANE loop (kernels.hpp: GATO_PCG_TAIL): it now iterates in
    // 0.711 us alone against the pair form's 0.714 (profiles/r06_pcg_rate.txt) and is ahead by ~1 % on whole solves at every B <= 512 (its prologue is the
    // lighter one).  So the pair form is OPT-IN now: GATO_PCG_PAIR = 1 (the two stay bit-identical: tests/test_gpu_parity.py).
    bool pair = false;
    if constexpr (NX == 12) {
        const int T2 = 2 * PcgcShape<NX, 3, 0>::threads(s->N * s->nx);
        const int cus = s->cus;
        const char* pe = getenv("GATO_PCG_PAIR");
        const bool want = pe ? atoi(pe) != 0 : false;
        (void)cus;
        pair = fused && want && T2 <= 256 && T2 >= 64 &&
               grant_lds(reinterpret_cast<const void*>(&pcgc_kernel<M, 3, 256, true, true, true>), pcg_fused_lds<M>(s)) &&
               grant_lds(reinterpret_cast<const void*>(&pcgc_kernel<M, 3, 256, true, true, true, true>), pcg_fused_lds<M>(s));
    }
    s->pcg_pair = pair ? 1 : 0;
    // Hardest-first scheduling of the PCG launch (Buffers::order, maintained by one extra workgroup of the step launch): wherever a CU
    // hosts more than one trajectory, in rounds (iiwa14 N = 64 B = 512: one 7-wavefront workgroup at a time per CU) or side by side
    // (C2: four).  Workgroups are dealt to the CUs in index order, so sorted by difficulty every CU gets one trajectory of each
    // difficulty class: the long ones start first and finish alone at the lone-trajectory rate instead of sharing a SIMD with another
    // long one.  Measured: C5 shard 4.72 -> 4.18 ms per solve, C2 1.81 -> 1.76 ms.  Results do not depend on the order.
    {
        const char* oe = getenv("GATO_PCG_ORDER");
        const bool can = step_fused(s) && (choice == 1 || choice == 2 || choice == 3 || choice == 4);
        s->pcg_rounds = can && (oe ? atoi(oe) != 0 : s->B > s->cus) ? 2 : 1;
    }
    return GATO_OK;
}
static int plan_pcg_dispatch(GatoSolver* s) { return s->plant == GATO_PLANT_INDY7 ? plan_pcg<Indy7>(s) : plan_pcg<Iiwa14>(s); }

// row0_done: the Q_0 rows were formed by the assembly kernel's cost task (launch_kkt with row0 = 1: every solve path)
template<class M> static void launch_schur(GatoSolver* s, hipStream_t st, float dt, bool force_stair = false, bool no_stair = false, bool row0_done = false)
{
    // lanes per (b,k): 4 with rows 3l..3l+2 (indy7); nq odd (iiwa14) divides evenly only into 2 x 7 rows -- heavier on registers
    // (AGPR moves, a few spills) but still ahead of a lane-per-knot kernel pair: 185 vs 238 us per launch at C3
    const long probs = (long)s->B * s->N;
    constexpr int LPP = (M::NQ % 2 == 0) ? 4 : 2;
    // right blocks (the transposes of the next block row's left blocks) are scattered 4-byte column writes: skipped when nobody reads them
    // -- the symmetric-storage PCG kernel and the direct sweep work from the left blocks (stage tests ask for the complete matrices)
    const int wr = (force_stair || !(s->pcg_choice == 7 || no_stair)) ? 1 : 0;
    if (s->schur_rowlane) {  // one row per lane, 16 lanes per knot: the default for nq odd (GATO_SCHUR_ROWLANE)
        if (row0_done) hipLaunchKernelGGL((schur1_kernel<M, false>), dim3(cdiv(probs * 16, 256), 1), dim3(256), 0, st, s->bf, s->N, s->B, dt, wr);
        else hipLaunchKernelGGL((schur1_kernel<M, true>), dim3(cdiv(probs * 16, 256), 2), dim3(256), 0, st, s->bf, s->N, s->B, dt, wr);
    } else
        hipLaunchKernelGGL((schurq_kernel<M, LPP>), dim3(cdiv(probs * LPP, 256), 2), dim3(256), 0, st, s->bf, s->N, s->B, dt, wr);
    if (!no_stair && (force_stair || !s->pcg_fold))
        hipLaunchKernelGGL((schur2_kernel<M>), dim3(cdiv(probs, 64)), dim3(64), 0, st, s->bf, s->N, s->B, wr);
}

template<class M, int RPT, int FORCE_WPS = 0> static void launch_pcgc(GatoSolver* s, hipStream_t st, int sqp_iter, int write_p)
{
    constexpr int NX = 2 * M::NQ;
    if constexpr (NX % RPT == 0) {
        using Sh = PcgcShape<NX, RPT, FORCE_WPS>;
        const int T = Sh::threads(s->N * s->nx);
        const bool full = T * RPT == s->N * s->nx;   // every thread owns rows: the mask-free form of the fold kernel (the solve path)
        if (s->pcg_fold && full)
            hipLaunchKernelGGL((pcgc_kernel<M, RPT, Sh::MAXT, true, false, false, true>), dim3(s->B), dim3(T), pcg_vec_lds(s) + pcg_fold_lds(s), st,
                               s->bf, s->N, s->B, s->p.max_pcg_iters, sqp_iter, write_p, 0.f);
        else if (s->pcg_fold)
            hipLaunchKernelGGL((pcgc_kernel<M, RPT, Sh::MAXT, true>), dim3(s->B), dim3(T), pcg_vec_lds(s) + pcg_fold_lds(s), st, s->bf, s->N, s->B,
                               s->p.max_pcg_iters, sqp_iter, write_p, 0.f);
        else
            hipLaunchKernelGGL((pcgc_kernel<M, RPT, Sh::MAXT, false>), dim3(s->B), dim3(T), pcg_vec_lds(s), st, s->bf, s->N, s->B,
                               s->p.max_pcg_iters, sqp_iter, 0, 0.f);
    }
}

template<class M> static void launch_pcg_fused(GatoSolver* s, hipStream_t st, float dt, int sqp_iter)
{
    constexpr int NX = 2 * M::NQ;
    if constexpr (NX == 12) {
        const int T = PcgcShape<NX, 3, 0>::threads(s->N * s->nx);
        const bool full = T * 3 == s->N * s->nx;   // every thread owns a row group (N a multiple of 16)
        if (s->pcg_pair) {
            if (full)
                hipLaunchKernelGGL((pcgc_kernel<M, 3, 256, true, true, true, true>), dim3(s->B), dim3(2 * T), pcg_fused_lds<M>(s), st, s->bf, s->N, s->B,
                                   s->p.max_pcg_iters, sqp_iter, 0, dt);
            else
                hipLaunchKernelGGL((pcgc_kernel<M, 3, 256, true, true, true>), dim3(s->B), dim3(2 * T), pcg_fused_lds<M>(s), st, s->bf, s->N, s->B,
                                   s->p.max_pcg_iters, sqp_iter, 0, dt);
        } else {
            if (full && T == 128)   // N = 32 (C2): exactly two wavefronts -- the instantiation whose reductions read two partials, not four (block_sum<.., TWO>)
                hipLaunchKernelGGL((pcgc_kernel<M, 3, 128, true, true, false, true>), dim3(s->B), dim3(T), pcg_fused_lds<M>(s), st, s->bf, s->N, s->B,
                                   s->p.max_pcg_iters, sqp_iter, 0, dt);
            else if (full)
                hipLaunchKernelGGL((pcgc_kernel<M, 3, 256, true, true, false, true>), dim3(s->B), dim3(T), pcg_fused_lds<M>(s), st, s->bf, s->N, s->B,
                                   s->p.max_pcg_iters, sqp_iter, 0, dt);
            else
                hipLaunchKernelGGL((pcgc_kernel<M, 3, 256, true, true>), dim3(s->B), dim3(T), pcg_fused_lds<M>(s), st, s->bf, s->N, s->B,
                                   s->p.max_pcg_iters, sqp_iter, 0, dt);
        }
    }
}

template<class M> static void launch_pcg(GatoSolver* s, hipStream_t st, int sqp_iter, int write_p = 0)
{
    const int rows = s->N * s->nx;
    const size_t lds = pcg_vec_lds(s);
    switch (s->pcg_choice) {
        case 2: launch_pcgc<M, 3>(s, st, sqp_iter, write_p); return;
        case 3: launch_pcgc<M, 2>(s, st, sqp_iter, write_p); return;
        case 1: launch_pcgc<M, 6>(s, st, sqp_iter, write_p); return;
        case 4: launch_pcgc<M, 1, 4>(s, st, sqp_iter, write_p); return;
        case 7: launch_pcgs<M>(s, st, sqp_iter, s->pcg_fold != 0); return;
        default: break;
    }
    const int T1 = ((rows + 63) / 64) * 64;
    if (T1 <= 512) {
        hipLaunchKernelGGL((pcg_kernel<M, 1, false, 512>), dim3(s->B), dim3(T1), lds, st, s->bf, s->N, s->B, s->p.max_pcg_iters, sqp_iter);
    } else if (T1 <= 1024) {
        hipLaunchKernelGGL((pcg_kernel<M, 1, false, 1024>), dim3(s->B), dim3(T1), lds, st, s->bf, s->N, s->B, s->p.max_pcg_iters, sqp_iter);
    } else {
        const int T2 = (((rows + 1) / 2 + 63) / 64) * 64;
        if (T2 <= 768) {
            hipLaunchKernelGGL((pcg_kernel<M, 2, false, 768>), dim3(s->B), dim3(T2), lds, st, s->bf, s->N, s->B, s->p.max_pcg_iters, sqp_iter);
        } else if (T2 <= 1024) {
            hipLaunchKernelGGL((pcg_kernel<M, 2, true, 1024>), dim3(s->B), dim3(T2), lds, st, s->bf, s->N, s->B, s->p.max_pcg_iters, sqp_iter);
        } else {
            const int T4 = (((rows + 3) / 4 + 63) / 64) * 64;  // N = 256
            hipLaunchKernelGGL((pcg_kernel<M, 4, true, 1024>), dim3(s->B), dim3(T4), lds, st, s->bf, s->N, s->B, s->p.max_pcg_iters, sqp_iter);
        }
    }
}
// opt-in direct solve of S lambda = gamma.  Two kernels: the sweep (one wavefront per trajectory, N dependent block eliminations) and block
// cyclic reduction (one workgroup of up to 16 wavefronts per trajectory, log2(N) levels of independent eliminations, the kept rows'
// products on the matrix cores).  Measured on MI355X, one-iteration solves (tools/exp/direct_time.py -> DESIGN.md 5d): the reduction wins
// wherever its workgroups are resident at once (indy7 N = 128, B = 1: 78 vs 310 us for the linear solve, PCG 236) and from N = 64 on at
// every batch size (N = 128, B = 1024: 365 vs 446 us); the sweep keeps short horizons at full batches (N = 32, B = 1024: 104 vs 119 us).
// GATO_DIRECT_CR = 0 / 1 forces the choice.
static bool direct_uses_cr(const GatoSolver* s)
{
    if (s->direct_cr_forced >= 0) return s->direct_cr_forced != 0;
    if (s->N < 8) return false;
    if (s->N >= 64) return true;
    const long waves = (long)s->B * (s->N >= 32 ? 16 : (s->N >= 16 ? 8 : 4));
    return waves <= (long)s->cus * 16;   // every workgroup resident at once, four wavefronts per SIMD
}
// the LDS tier of btd_cr_kernel: [L | D | D^-1 | C][16][nx^2] + g [16][nx]
template<class M> static constexpr size_t cr_lds() { return (size_t)(4 * 16 * (2 * M::NQ) * (2 * M::NQ) + 16 * (2 * M::NQ)) * sizeof(float); }
// The kernel's static tiles + this tier exceed the 64 KB a kernel gets unasked, so the tier is asked for -- per FUNCTION AND DEVICE
// (hipFuncSetAttribute acts on the current device): once per handle, on the handle's device, with the status checked, when the direct mode
// is selected (gato_set_linear_solver).  Round 3 granted it once per process behind a function-local static and ignored the status: a host
// that drives two GPUs from one process (DeviceGuard) got a failed launch on the second device, reported an iteration later.
template<class M> static int grant_direct(GatoSolver* s)
{
    if (s->cr_granted) return GATO_OK;
    auto grant = [](const void* fn, size_t bytes) { return hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes) == hipSuccess; };
    bool ok = grant(reinterpret_cast<const void*>(&btd_cr_kernel<M, 4>), cr_lds<M>()) && grant(reinterpret_cast<const void*>(&btd_cr_kernel<M, 8>), cr_lds<M>());
    if (!kDouble) ok = ok && grant(reinterpret_cast<const void*>(&btd_cr_kernel<M, 16>), cr_lds<M>());   // (never launched by the float64 build)
    if (!ok) {
        (void)hipGetLastError();
        return fail(GATO_ERR_HIP, "the direct solver's cyclic-reduction kernel was refused its LDS on this device (hipFuncSetAttribute)");
    }
    s->cr_granted = true;
    return GATO_OK;
}
template<class M, int WAVES> static void launch_cr(GatoSolver* s, hipStream_t st, int sqp_iter)
{
    hipLaunchKernelGGL((btd_cr_kernel<M, WAVES>), dim3(s->B), dim3(64 * WAVES), cr_lds<M>(), st, s->bf, s->N, s->B, sqp_iter);
}
template<class M> static void launch_direct(GatoSolver* s, hipStream_t st, int sqp_iter)
{
    if (direct_uses_cr(s)) {
        if (s->N >= 32 && !kDouble) launch_cr<M, 16>(s, st, sqp_iter);   // (float64 build: 8 wavefronts, its LDS tier is twice the size)
        else if (s->N >= 16) launch_cr<M, 8>(s, st, sqp_iter);
        else launch_cr<M, 4>(s, st, sqp_iter);
        return;
    }
    hipLaunchKernelGGL((btd_direct_kernel<M>), dim3(s->B), dim3(64), 0, st, s->bf, s->N, s->B, sqp_iter);
}
template<class M> static void launch_dz(GatoSolver* s, hipStream_t st, float dt, int sqp_iter)
{
    hipLaunchKernelGGL((dz_kernel<M>), dim3(cdiv((long)s->B * s->N, 256)), dim3(256), 0, st, s->bf, s->N, s->B, dt, sqp_iter);
}
// dz + merit(8 alphas) + line search in one launch: a workgroup of 8 N lanes per trajectory
static bool step_fused(const GatoSolver* s)
{
    return s->fuse_step && NUM_ALPHAS * s->N <= 1024;
}
// last: this is the final iteration of the solve -- the line search also puts drho back to its default (bsqp.cuh:189)
// first: the first step launch of a solve also forms the merit of the current iterate ((NUM_ALPHAS + 1) N lanes; see merit_in_step)
template<class M> static void launch_step(GatoSolver* s, hipStream_t st, float dt, int sqp_iter, int last, bool first = false)
{
    const float thresh = exit_threshold(s);
    size_t lds = (size_t)(((s->traj + 3) & ~3) + 12 + 16 + s->N * s->nx) * sizeof(float);   // the step, 8 + 1 merits, 16 wavefront partials, dz_rows' residuals
    const int T = (NUM_ALPHAS + (first ? 1 : 0)) * s->N;
    float* init0 = first ? s->d_merit_init0 : nullptr;
    const int extra = s->pcg_rounds > 1 ? 1 : 0;   // one more workgroup re-orders the trajectories for the next PCG launch
    if (extra && lds < 512 * sizeof(int)) lds = 512 * sizeof(int);
    if (T <= 512)
        hipLaunchKernelGGL((step_kernel<M, 512>), dim3(s->B + extra), dim3(T), lds, st, s->bf, s->N, s->B, dt, sqp_iter, thresh, s->adapt_rho,
                           (const float*)s->d_drho_init, last, init0);
    else
        hipLaunchKernelGGL((step_kernel<M, 1024>), dim3(s->B + extra), dim3(T), lds, st, s->bf, s->N, s->B, dt, sqp_iter, thresh, s->adapt_rho,
                           (const float*)s->d_drho_init, last, init0);
}
// The initial merit of a solve (bsqp.cuh:116-118) rides in the first step launch when that launch exists and has the lanes to spare:
// no merit launch ahead of the loop (8-11 us per solve: 10 % of a one-iteration MPC solve at B = 1); the first assembly launch
// clears the per-solve slab then.
// Only while every workgroup of that launch is still resident at once (the step kernel runs 3-4 wavefronts per SIMD, >= 12 per CU,
// tests/test_kernel_resources.py): at C2 the ninth N lanes push 1024 workgroups of 5 wavefronts past the chip (+10 us), while the
// merit launch they replace costs 11 us of a 1.8 ms solve.  Either way the same device code forms the value: same bits.
static bool merit_in_step(const GatoSolver* s, uint32_t iters)
{
    const int T = (NUM_ALPHAS + 1) * s->N;
    if (!(iters > 0 && step_fused(s) && T <= 1024)) return false;
    if (s->merit_in_step_forced >= 0) return s->merit_in_step_forced != 0;   // GATO_MERIT_IN_STEP
    return (long)s->B * ((T + 63) / 64) <= (long)s->cus * 12;
}
static void launch_ls(GatoSolver* s, hipStream_t st, int sqp_iter, int last)
{
    const float thresh = exit_threshold(s);
    hipLaunchKernelGGL(line_search_kernel, dim3(s->B), dim3(128), 0, st, s->bf, s->traj, s->B, s->adapt_rho, sqp_iter, thresh,
                       (const float*)s->d_drho_init, last);
}

static void mark(GatoSolver* s, hipStream_t st, int stage, size_t& ei)
{
    if (!s->profiling) return;
    if (ei >= s->events.size()) {
        hipEvent_t e;
        (void)hipEventCreate(&e);
        s->events.push_back(e);
        s->event_stage.push_back(stage);
    }
    s->event_stage[ei] = stage;
    (void)hipEventRecord(s->events[ei], st);
    ei++;
}

// ---- sharded batch: RCCL bound at run time -------------------------------------------------------------------------------------------
// The solved count is the only thing that couples the trajectories (bsqp.cuh:165); on a batch sharded over the GPUs of a node every rank
// counts its own rows (Buffers::num_solved_w) and ONE 4-byte ncclAllReduce per SQP iteration, enqueued on the solve's stream between the PCG
// launch and the step launch, gives every rank the whole batch's count (Buffers::num_solved): the same exit in the same iteration on every
// rank, for any solve_ratio, with no host round trip.  librccl is opened with dlopen when a communicator is asked for.
struct Rccl {
    void* lib = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*AllReduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*AllGather)(const void*, void*, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
    const char* (*GetErrorString)(ncclResult_t) = nullptr;
};
static Rccl g_rccl;
static int rccl_load()
{
    if (g_rccl.lib) return GATO_OK;
    // a process that already carries an RCCL (PyTorch ships its own) gets THAT one: same soname, and the one HIP runtime both sit on
    const char* names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
    void* h = nullptr;
    for (const char* n : names)
        if ((h = dlopen(n, RTLD_NOW | RTLD_GLOBAL)) != nullptr) break;
    if (!h) return fail(GATO_ERR_INVALID, std::string("librccl.so is not available: ") + dlerror());
#define SYM(field, name)                                                                              \
    *reinterpret_cast<void**>(&g_rccl.field) = dlsym(h, name);                                        \
    if (!g_rccl.field) return fail(GATO_ERR_INVALID, std::string("librccl.so has no symbol ") + name)
    SYM(GetUniqueId, "ncclGetUniqueId");
    SYM(CommInitRank, "ncclCommInitRank");
    SYM(CommDestroy, "ncclCommDestroy");
    SYM(AllReduce, "ncclAllReduce");
    SYM(AllGather, "ncclAllGather");
    SYM(GetErrorString, "ncclGetErrorString");
#undef SYM
    g_rccl.lib = h;
    return GATO_OK;
}
#define NCCLCHK(expr)                                                                                                       \
    do {                                                                                                                    \
        ncclResult_t r_ = (expr);                                                                                           \
        if (r_ != ncclSuccess) return fail(GATO_ERR_HIP, std::string(#expr) + ": " + g_rccl.GetErrorString(r_));            \
    } while (0)

// after the PCG launch of iteration `it`: Buffers::num_solved[it] := the whole batch's count
static int reduce_solved(GatoSolver* s, hipStream_t st, int it)
{
    if (s->comm) {
        NCCLCHK(g_rccl.AllReduce(s->d_ns_local + it, s->bf.num_solved + it, 1, ncclUint32, ncclSum, (ncclComm_t)s->comm, st));
    } else if (s->d_ns_remote) {
        hipLaunchKernelGGL(add_remote_solved_kernel, dim3(1), dim3(64), 0, st, s->bf.num_solved, (const uint32_t*)s->d_ns_local, (const uint32_t*)s->d_ns_remote, it);
    } else {
        return fail(GATO_ERR_INVALID, "sharded solver without a communicator");
    }
    return GATO_OK;
}
// A captured solve (gato_set_graph_mode) has the sharding state baked in: the exit threshold, the array the exit rule reads, the
// per-iteration collective / add_remote_solved nodes and the communicator itself.  Whatever changes any of them drops the graph; the next
// gato_solve captures again (a stale graph would apply the unsharded rule without an error, or launch on a destroyed communicator).
static void drop_graph(GatoSolver* s)
{
    if (s->graph_exec) { (void)hipGraphExecDestroy(s->graph_exec); s->graph_exec = nullptr; }
}
// the deferred form: after the LAST iteration of a speculative solve, the whole count vector in one reduction
static int reduce_solved_all(GatoSolver* s, hipStream_t st, int iters)
{
    if (s->comm) {
        NCCLCHK(g_rccl.AllReduce(s->d_ns_local, s->bf.num_solved, (size_t)iters, ncclUint32, ncclSum, (ncclComm_t)s->comm, st));
    } else if (s->d_ns_remote) {
        hipLaunchKernelGGL(add_remote_solved_kernel, dim3(1), dim3(64), 0, st, s->bf.num_solved, (const uint32_t*)s->d_ns_local, (const uint32_t*)s->d_ns_remote, -iters);
    } else {
        return fail(GATO_ERR_INVALID, "sharded solver without a communicator");
    }
    return GATO_OK;
}
static void set_sharded(GatoSolver* s, long global_batch)
{
    drop_graph(s);
    s->global_batch = global_batch;
    s->bf.num_solved_w = global_batch > 0 ? s->d_ns_local : s->bf.num_solved;
}

// Every rank of a communicator must count the same way (a rank in the deferred mode issues ONE reduction per solve, a rank in the per-iteration
// mode one per SQP iteration: mixed, the collectives never match and the job hangs).  max over the ranks of {mode, 1 - mode}: both 1 = disagreement.
static int agree_on_count_mode(GatoSolver* s, bool* disagree = nullptr)
{
    if (disagree) *disagree = false;
    if (!s->comm) return GATO_OK;
    if (!s->d_agree) {
        int rc = dalloc(s, &s->d_agree, 2);
        if (rc != GATO_OK) return rc;
    }
    const uint32_t mine[2] = {(uint32_t)(s->deferred_count ? 1 : 0), (uint32_t)(s->deferred_count ? 0 : 1)};
    uint32_t all[2] = {0, 0};
    hipStream_t st = s->own_stream;
    HIPCHK(hipMemcpyAsync(s->d_agree, mine, sizeof(mine), hipMemcpyHostToDevice, st));
    NCCLCHK(g_rccl.AllReduce(s->d_agree, s->d_agree, 2, ncclUint32, ncclMax, (ncclComm_t)s->comm, st));
    HIPCHK(hipMemcpyAsync(all, s->d_agree, sizeof(all), hipMemcpyDeviceToHost, st));
    HIPCHK(hipStreamSynchronize(st));
    if (all[0] && all[1]) {
        if (disagree) *disagree = true;
        return fail(GATO_ERR_INVALID, "the ranks of this communicator disagree on the solved-count mode (GATO_SOLVED_COUNT / gato_set_solved_count_mode): set the same mode on every rank");
    }
    return GATO_OK;
}
// librccl can be opened and has every entry point this library binds -- no RCCL call is made (gato_comm_unique_id would start a bootstrap listener)
extern "C" int gato_comm_available(void) { return rccl_load(); }
extern "C" int gato_comm_unique_id(char* out128)
{
    if (!out128) return fail(GATO_ERR_INVALID, "null argument");
    int rc = rccl_load();
    if (rc) return rc;
    ncclUniqueId id;
    NCCLCHK(g_rccl.GetUniqueId(&id));
    static_assert(sizeof(id) == 128, "ncclUniqueId is 128 bytes");
    memcpy(out128, &id, sizeof(id));
    return GATO_OK;
}
// step 1 of 2: ncclCommInitRank alone.  No collective is issued on the new communicator, so a caller with a side channel (gato_amd.sharding.connect:
// torch.distributed) can compare the ranks' return codes first and let every rank drop an initialisation that failed on ANY of them
extern "C" int gato_comm_init_rank(GatoSolver* s, const char* id128, int world_size, int rank, int64_t global_batch)
{
    if (!s || !id128) return fail(GATO_ERR_INVALID, "null argument");
    if (world_size < 1 || rank < 0 || rank >= world_size) return fail(GATO_ERR_INVALID, "rank must lie in [0, world_size)");
    if (global_batch != (int64_t)world_size * s->B) return fail(GATO_ERR_INVALID, "global_batch must be world_size x the solver's batch (equal shards)");
    if (s->comm) return fail(GATO_ERR_INVALID, "the solver already has a communicator");
    GUARD(s);
    int rc = rccl_load();
    if (rc) return rc;
    rc = sync_last(s);
    if (rc) return rc;
    ncclUniqueId id;
    memcpy(&id, id128, sizeof(id));
    ncclComm_t c = nullptr;
    if (!s->d_agree && (rc = dalloc(s, &s->d_agree, 2)) != GATO_OK) return rc;   // step 2's only allocation, made HERE: what can still fail there is HIP / RCCL itself
    NCCLCHK(g_rccl.CommInitRank(&c, world_size, id, rank));
    s->comm = c;
    s->comm_confirmed = false;
    s->world = world_size;
    s->rank = rank;
    set_sharded(s, (long)global_batch);
    s->periter_left = 0;
    s->replay_backoff = 8;
    return GATO_OK;
}
// step 2 of 2, COLLECTIVE on the new communicator: the ranks agree on the solved-count mode; sharded solves refuse until this has succeeded.
// A disagreement is seen by every rank alike (the reduced pair is the same everywhere): all of them drop the communicator.  Any OTHER failure
// (allocation, HIP, RCCL) is this rank's alone: its own message is kept, the communicator is dropped HERE, and the peers -- who may have
// passed -- are left with a communicator whose partner is gone: they must gato_comm_destroy (sharding.connect compares the ranks' verdicts)
extern "C" int gato_comm_confirm(GatoSolver* s)
{
    if (!s) return fail(GATO_ERR_INVALID, "null solver");
    if (!s->comm) return fail(GATO_ERR_INVALID, "the solver has no communicator (gato_comm_init_rank)");
    GUARD(s);
    bool disagree = false;
    const int rc = agree_on_count_mode(s, &disagree);
    if (rc != GATO_OK) {
        const std::string why = g_err;
        (void)gato_comm_destroy(s);
        return fail(rc, disagree ? "the ranks disagree on the solved-count mode (GATO_SOLVED_COUNT / gato_set_solved_count_mode before gato_comm_init): no communicator on any rank"
                                 : "this rank failed while the ranks agreed on the solved-count mode (its communicator is dropped; the peers must gato_comm_destroy theirs): " + why);
    }
    s->comm_confirmed = true;
    return GATO_OK;
}
extern "C" int gato_comm_init(GatoSolver* s, const char* id128, int world_size, int rank, int64_t global_batch)
{
    const int rc = gato_comm_init_rank(s, id128, world_size, rank, global_batch);
    return rc != GATO_OK ? rc : gato_comm_confirm(s);
}
extern "C" int gato_comm_destroy(GatoSolver* s)
{
    if (!s) return fail(GATO_ERR_INVALID, "null solver");
    GUARD(s);
    int rc = sync_last(s);
    if (rc) return rc;
    drop_graph(s);
    if (s->comm) {
        NCCLCHK(g_rccl.CommDestroy((ncclComm_t)s->comm));
        s->comm = nullptr;
        s->comm_confirmed = false;
    }
    s->world = 1;
    s->rank = 0;
    if (!s->d_ns_remote) set_sharded(s, 0);
    return GATO_OK;
}
// the ONE collective of a solve's data path: every rank's packed results [count reals] -> [world][count] on every rank
extern "C" int gato_gather_results(GatoSolver* s, const float* d_local, float* d_all, uint64_t count, void* stream)
{
    if (!s || !d_local || !d_all) return fail(GATO_ERR_INVALID, "null argument");
    if (!s->comm) return fail(GATO_ERR_INVALID, "the solver has no communicator (gato_comm_init)");
    GUARD(s);
    NCCLCHK(g_rccl.AllGather(d_local, d_all, (size_t)count, kDouble ? ncclFloat64 : ncclFloat32, (ncclComm_t)s->comm, (hipStream_t)stream));
    return GATO_OK;
}
// TEST HOOK: a shard without a communicator -- the other shards' solved counts per SQP iteration are given (n <= max_sqp_iters entries,
// the rest 0); global_batch = 0 returns the solver to the unsharded rule
extern "C" int gato_debug_set_remote_solved(GatoSolver* s, const uint32_t* per_iter, int n, int64_t global_batch)
{
    if (!s) return fail(GATO_ERR_INVALID, "null solver");
    if (s->comm) return fail(GATO_ERR_INVALID, "the solver has a communicator");
    GUARD(s);
    int rc = sync_last(s);
    if (rc) return rc;
    if (global_batch <= 0) {
        set_sharded(s, 0);
        return GATO_OK;
    }
    if (!per_iter || n < 0 || (uint32_t)n > s->max_iters_alloc || global_batch < s->B) return fail(GATO_ERR_INVALID, "bad argument");
    if (!s->d_ns_remote) {
        rc = dalloc(s, &s->d_ns_remote, s->max_iters_alloc);
        if (rc) return rc;
    }
    std::vector<uint32_t> h(s->max_iters_alloc, 0u);
    for (int i = 0; i < n; i++) h[i] = per_iter[i];
    HIPCHK(hipMemcpy(s->d_ns_remote, h.data(), h.size() * sizeof(uint32_t), hipMemcpyHostToDevice));
    set_sharded(s, (long)global_batch);
    return GATO_OK;
}

// the launch sequence of one solve on `st`.  per_iter_count: a sharded solve shares the solved count after every PCG launch (the exact,
// fully asynchronous form); otherwise the kernels read whatever exit_threshold() says (the unsharded rule, or +inf in a speculative run)
template<class M> static int enqueue_solve(GatoSolver* s, float dt, hipStream_t st, uint32_t iters, bool per_iter_count)
{
    Buffers& bf = s->bf;
    const int B = s->B;
    size_t ei = 0;
    mark(s, st, -1, ei);
    // bsqp.cuh:112-118: the merit of the initial iterate (kept twice: running merit and merit_initial0) and the clearing of dz,
    // pcg_iters, converged, ctrl, num_solved (the slab is a multiple of 64 words) -- inside the first step / first assembly launch
    // where the plan allows it, otherwise ONE launch ahead of the loop
    const bool ride = merit_in_step(s, iters);
    if (!ride) {
        launch_merit<M>(s, st, 1, dt, 0, -1, bf.merit_cur, s->d_merit_init0, s->zero_slab, s->zero_words);
        mark(s, st, ST_MERIT, ei);
    }
    for (uint32_t it = 0; it < iters; it++) {
        const bool direct = s->linear_solver == 1;
        const bool fused = s->pcg_fused != 0 && !direct;
        const bool row0 = fused || s->schur_rowlane;   // the Q_0 rows by the assembly kernel's cost task (always, where the Schur kernel allows it)
        launch_kkt<M>(s, st, dt, (int)it, row0 ? 1 : 0, ride && it == 0);
        mark(s, st, ST_KKT, ei);
        if (direct) {
            launch_schur<M>(s, st, dt, false, true, row0);   // S and gamma only: no preconditioner in this mode
            mark(s, st, ST_SCHUR, ei);
            launch_direct<M>(s, st, (int)it);
        } else if (fused) {
            launch_pcg_fused<M>(s, st, dt, (int)it);
        } else {
            launch_schur<M>(s, st, dt, false, false, row0);
            mark(s, st, ST_SCHUR, ei);
            launch_pcg<M>(s, st, (int)it);
        }
        if (per_iter_count) {
            const int rc_r = reduce_solved(s, st, (int)it);
            if (rc_r != GATO_OK) return rc_r;
        }
        mark(s, st, ST_PCG, ei);
        if (step_fused(s)) {
            launch_step<M>(s, st, dt, (int)it, it + 1 == iters, ride && it == 0);
            mark(s, st, ST_MERIT, ei);
        } else {
            launch_dz<M>(s, st, dt, (int)it);
            mark(s, st, ST_DZ, ei);
            launch_merit<M>(s, st, NUM_ALPHAS, dt, 1, (int)it, bf.merit);
            mark(s, st, ST_MERIT, ei);
            launch_ls(s, st, (int)it, it + 1 == iters);
            mark(s, st, ST_LS, ei);
        }
    }
    // The merit of the returned xu (bsqp.cuh:180-182) needs no launch: merit_cur already holds it -- the line search stores the merit
    // of the step it accepts, evaluated at fma(alpha, dz, xu), which is exactly what it then writes to xu
    // (tests/test_gpu_parity.py::test_final_merit_is_the_merit_of_the_returned_iterates recomputes it).
    // bsqp.cuh:189 (drho back to its default; rho is NOT reset) is done by the kernel that ends the loop -- the last line search or the
    // solve_ratio break -- so only a solve without iterations needs the copy
    if (iters == 0) HIPCHK(hipMemcpyAsync(bf.drho, s->d_drho_init, B * sizeof(float), hipMemcpyDeviceToDevice, st));
    HIPCHK(hipGetLastError());
    return GATO_OK;
}

// async_only: the caller cannot take a host synchronisation inside the solve (stream capture): a sharded solve then shares the count per iteration
template<class M> static int solve_impl(GatoSolver* s, float* d_xu, float dt, const float* d_xs, const float* d_ref, hipStream_t st, bool async_only = false, bool lazy_ok = false)
{
    Buffers& bf = s->bf;
    const int B = s->B;
    bf.xu = d_xu; bf.x_s = d_xs; bf.ref = d_ref;
    s->last_stream = st;
    s->last_stream_valid = true;
    const uint32_t iters = s->p.max_sqp_iters <= s->max_iters_alloc ? s->p.max_sqp_iters : s->max_iters_alloc;
    const bool sharded = s->global_batch > 0;
    if (!sharded) return enqueue_solve<M>(s, dt, st, iters, false);
    if (s->comm && !s->comm_confirmed)
        return fail(GATO_ERR_INVALID, "the communicator is not confirmed: gato_comm_confirm (the ranks' agreement on the solved-count mode) has not succeeded on this handle");
    bool deferred = s->deferred_count && !async_only && iters > 0;
    if (deferred) {
        // a caller's stream capture (gato_solve_device under hipStreamBeginCapture) cannot take the host wait of the deferred form: it would
        // invalidate the capture.  Such a solve shares the count per iteration, the fully asynchronous form
        hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
        if (hipStreamIsCapturing(st, &cs) != hipSuccess || cs != hipStreamCaptureStatusNone) { (void)hipGetLastError(); deferred = false; }
    }
    if (deferred && s->periter_left > 0) { s->periter_left--; deferred = false; }
    if (!deferred) { s->n_periter++; return enqueue_solve<M>(s, dt, st, iters, true); }
    // ---- deferred: the solved count is the ONLY coupling between the shards (bsqp.cuh:165) and on every workload where trajectories do not
    // converge it never changes anything -- ten small-message all-reduces on the critical path of every solve for nothing.  So: snapshot what a
    // solve changes for good, run it as if the rule never fired (own rows counted), reduce the whole count vector ONCE, let the host look at it.
    {   // each of the four is allocated once; a failure leaves the others for the next attempt and this solve returns the error
        int rc;
        if (!s->d_snap_xu && (rc = dalloc(s, &s->d_snap_xu, (size_t)B * s->traj, false)) != GATO_OK) return rc;
        if (!s->d_snap_lambda && (rc = dalloc(s, &s->d_snap_lambda, (size_t)B * s->vecp, false)) != GATO_OK) return rc;
        if (!s->d_snap_rho && (rc = dalloc(s, &s->d_snap_rho, 2 * (size_t)B, false)) != GATO_OK) return rc;
        if (!s->h_counts) {
            void *hc = nullptr, *dc = nullptr;
            HIPCHK(hipHostMalloc(&hc, (s->max_iters_alloc + 1) * sizeof(uint32_t), hipHostMallocMapped | hipHostMallocCoherent));
            if (hipHostGetDevicePointer(&dc, hc, 0) != hipSuccess) { (void)hipHostFree(hc); return fail(GATO_ERR_HIP, "hipHostGetDevicePointer (the verdict's pinned counts)"); }
            memset(hc, 0, (s->max_iters_alloc + 1) * sizeof(uint32_t));
            s->h_counts = static_cast<uint32_t*>(hc);
            s->d_h_counts = static_cast<uint32_t*>(dc);
        }
    }
    const size_t bx = (size_t)B * s->traj * sizeof(float), bl = (size_t)B * s->vecp * sizeof(float);
    hipLaunchKernelGGL(snapshot_kernel, dim3(s->cus > 0 ? s->cus * 4 : 256), dim3(256), 0, st, s->d_snap_xu, (const float*)d_xu, (uint32_t)(bx / sizeof(float)),
                       s->d_snap_lambda, (const float*)bf.lambda, (uint32_t)(bl / sizeof(float)), s->d_snap_rho, (const float*)bf.rho, (const float*)bf.drho, (uint32_t)B);
    s->thresh_override = INFINITY;
    int rc = enqueue_solve<M>(s, dt, st, iters, false);
    s->thresh_override = -1.f;
    if (rc != GATO_OK) return rc;
    rc = reduce_solved_all(s, st, (int)iters);
    if (rc != GATO_OK) return rc;
    // the reduced vector and, behind a system-scope fence, a fresh sequence number straight into host memory
    s->verdict_seq++;
    if (s->verdict_seq == 0) s->verdict_seq = 1;
    hipLaunchKernelGGL(publish_counts_kernel, dim3(1), dim3(64), 0, st, s->d_h_counts, (const uint32_t*)bf.num_solved, iters, s->max_iters_alloc, s->verdict_seq);
    HIPCHK(hipGetLastError());
    s->pend.active = true;
    s->pend.d_xu = d_xu; s->pend.dt = dt; s->pend.d_xs = d_xs; s->pend.d_ref = d_ref; s->pend.st = st; s->pend.iters = iters;
    // lazy: only gato_solve_device leaves the verdict to the next entry point; every other caller (gato_solve, the MPC session) goes on to use
    // the results on this stream at once
    return lazy_ok ? GATO_OK : settle(s);
}

// The verdict of the pending speculative solve, and its exact replay if the exit rule fired.  Called at the top of every entry point that reads or
// changes what the solve reads or writes (through sync_last, and by the stream-ordered ones: gato_solve_device, gato_reset_async,
// gato_copy_final_merit_device, the *_device helpers).  NOT by gato_gather_results: it touches nothing of the solver's (the caller orders it).
template<class M> static int settle_impl(GatoSolver* s)
{
    Buffers& bf = s->bf;
    const int B = s->B;
    const uint32_t iters = s->pend.iters;
    hipStream_t st = s->pend.st;
    s->pend.active = false;
    // the one host wait of a deferred solve: spin on the sequence number the device writes behind the reduction (a hipStreamSynchronize here took the
    // interrupt path: tens of microseconds with the device idle); the stream is queried now and then so that a device fault ends the wait
    volatile uint32_t* seq = s->h_counts + s->max_iters_alloc;
    for (uint64_t spins = 0; *seq != s->verdict_seq; spins++) {
        if ((spins & 0xfffff) == 0xfffff) {
            const hipError_t q = hipStreamQuery(st);
            if (q == hipSuccess) {   // everything on the stream has run: the number must be there (give the write a moment to land), else something is wrong
                HIPCHK(hipStreamSynchronize(st));
                if (*seq != s->verdict_seq) return fail(GATO_ERR_HIP, "the deferred solve finished without publishing its solved counts");
                break;
            }
            if (q != hipErrorNotReady) return fail(GATO_ERR_HIP, std::string("waiting for the verdict of a deferred solve: ") + hipGetErrorString(q));
        }
        __builtin_ia32_pause();
    }
    __atomic_thread_fence(__ATOMIC_ACQUIRE);
    s->n_deferred++;
    const float thresh = exit_threshold_exact(s);
    bool fired = false;
    for (uint32_t i = 0; i < iters; i++) fired = fired || (float)s->h_counts[i] >= thresh;
    if (!fired) { s->replay_backoff = 8; return GATO_OK; }
    // some iteration's whole-batch count reached the threshold: the speculative run went past the exit.  Back to the snapshot and again, exactly.
    s->n_replays++;
    s->periter_left = s->replay_backoff;
    s->replay_backoff = s->replay_backoff < 1024 ? s->replay_backoff * 2 : 1024;
    const size_t bx = (size_t)B * s->traj * sizeof(float), bl = (size_t)B * s->vecp * sizeof(float), bb = (size_t)B * sizeof(float);
    bf.xu = s->pend.d_xu; bf.x_s = s->pend.d_xs; bf.ref = s->pend.d_ref;
    HIPCHK(hipMemcpyAsync(s->pend.d_xu, s->d_snap_xu, bx, hipMemcpyDeviceToDevice, st));
    HIPCHK(hipMemcpyAsync(bf.lambda, s->d_snap_lambda, bl, hipMemcpyDeviceToDevice, st));
    HIPCHK(hipMemcpyAsync(bf.rho, s->d_snap_rho, bb, hipMemcpyDeviceToDevice, st));
    HIPCHK(hipMemcpyAsync(bf.drho, s->d_snap_rho + B, bb, hipMemcpyDeviceToDevice, st));
    return enqueue_solve<M>(s, s->pend.dt, st, iters, true);   // (counted in n_replays, not in n_periter: that one counts solves that never ran speculatively)
}
static int settle(GatoSolver* s)
{
    if (!s->pend.active) return GATO_OK;
    return s->plant == GATO_PLANT_INDY7 ? settle_impl<Indy7>(s) : settle_impl<Iiwa14>(s);
}

static int solve_dispatch(GatoSolver* s, float* d_xu, float dt, const float* d_xs, const float* d_ref, hipStream_t st, bool async_only = false, bool lazy_ok = false)
{
    int rc = settle(s);   // a pending solve's verdict (and replay) before the next one starts from its results
    if (rc != GATO_OK) return rc;
    return s->plant == GATO_PLANT_INDY7 ? solve_impl<Indy7>(s, d_xu, dt, d_xs, d_ref, st, async_only, lazy_ok) : solve_impl<Iiwa14>(s, d_xu, dt, d_xs, d_ref, st, async_only, lazy_ok);
}

static void collect_profile(GatoSolver* s)
{
    if (!s->profiling || s->events.size() < 2) return;
    memset(s->stage_us, 0, sizeof(s->stage_us));
    for (size_t i = 1; i < s->events.size(); i++) {
        f32_t ms = 0;
        if (hipEventElapsedTime(&ms, s->events[i - 1], s->events[i]) == hipSuccess && s->event_stage[i] >= 0) s->stage_us[s->event_stage[i]] += ms * 1e3;
    }
    f32_t ms = 0;
    if (hipEventElapsedTime(&ms, s->events.front(), s->events.back()) == hipSuccess) s->stage_us[ST_COUNT] = ms * 1e3;
}

extern "C" int gato_solve_device(GatoSolver* s, float* d_xu, float dt, const float* d_xs, const float* d_ref, void* stream)
{
    if (!s || !d_xu || !d_xs || !d_ref) return fail(GATO_ERR_INVALID, "null argument");
    GUARD(s);
    // a sharded solve in the deferred count mode returns with its verdict PENDING (taken by the next entry point on this handle, settle()); profiling
    // wants the stage events of a finished solve
    return solve_dispatch(s, d_xu, dt, d_xs, d_ref, (hipStream_t)stream, false, !s->profiling);
}

extern "C" int gato_solve(GatoSolver* s, float* xu, float dt, const float* x_s, const float* ref, double* sqp_time_us)
{
    if (!s || !xu || !x_s || !ref) return fail(GATO_ERR_INVALID, "null argument");
    GUARD(s);
    const size_t nxu = (size_t)s->B * s->traj * sizeof(float);
    // Everything of this call -- copies in, the launch sequence, copy out -- is ordered on the solver's OWN stream and the host waits on
    // that stream only: handles that share a device (several MPC solvers, two ranks on one GPU) do not serialise on each other the way a
    // device-wide synchronisation made them (the reference blocks the whole device: default stream + cudaDeviceSynchronize, bsqp.cuh:184)
    hipStream_t st = s->own_stream;
    int rc = sync_last(s);   // a solve enqueued on a caller's stream (gato_solve_device) may still be using the solver's state
    if (rc) return rc;
    HIPCHK(hipMemcpyAsync(s->d_xu_own, xu, nxu, hipMemcpyHostToDevice, st));
    HIPCHK(hipMemcpyAsync(s->d_xs_own, x_s, (size_t)s->B * s->nx * sizeof(float), hipMemcpyHostToDevice, st));
    HIPCHK(hipMemcpyAsync(s->d_ref_own, ref, (size_t)s->B * 6 * s->N * sizeof(float), hipMemcpyHostToDevice, st));
    HIPCHK(hipStreamSynchronize(st));
    const bool graph = s->graph_mode && !s->profiling;
    if (graph) {
        // (re)capture when something a kernel argument depends on changed; otherwise the instantiated graph is replayed as it is
        const uint32_t iters = s->p.max_sqp_iters <= s->max_iters_alloc ? s->p.max_sqp_iters : s->max_iters_alloc;
        if (!s->graph_exec || s->graph_dt != dt || s->graph_iters != iters || s->graph_adapt != s->adapt_rho || s->graph_lin != s->linear_solver) {
            if (s->graph_exec) { (void)hipGraphExecDestroy(s->graph_exec); s->graph_exec = nullptr; }
            hipGraph_t g = nullptr;
            HIPCHK(hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal));
            const int rc_c = solve_dispatch(s, s->d_xu_own, dt, s->d_xs_own, s->d_ref_own, st, true);   // no host wait inside a capture
            const hipError_t e_c = hipStreamEndCapture(st, &g);
            if (rc_c != GATO_OK) { if (g) (void)hipGraphDestroy(g); return rc_c; }
            if (e_c != hipSuccess) return fail(GATO_ERR_HIP, std::string("hipStreamEndCapture: ") + hipGetErrorString(e_c));
            const hipError_t e_i = hipGraphInstantiate(&s->graph_exec, g, nullptr, nullptr, 0);
            (void)hipGraphDestroy(g);
            if (e_i != hipSuccess) { s->graph_exec = nullptr; return fail(GATO_ERR_HIP, std::string("hipGraphInstantiate: ") + hipGetErrorString(e_i)); }
            s->graph_dt = dt; s->graph_iters = iters; s->graph_adapt = s->adapt_rho; s->graph_lin = s->linear_solver;
        }
        s->last_stream = st;
        s->last_stream_valid = true;
    }
    auto t0 = std::chrono::high_resolution_clock::now();
    if (graph) {
        HIPCHK(hipGraphLaunch(s->graph_exec, st));
    } else {
        rc = solve_dispatch(s, s->d_xu_own, dt, s->d_xs_own, s->d_ref_own, st);
        if (rc != GATO_OK) return rc;
    }
    HIPCHK(hipStreamSynchronize(st));
    auto t1 = std::chrono::high_resolution_clock::now();
    if (sqp_time_us) *sqp_time_us = std::chrono::duration<double, std::micro>(t1 - t0).count();
    collect_profile(s);
    HIPCHK(hipMemcpyAsync(xu, s->d_xu_own, nxu, hipMemcpyDeviceToHost, st));
    HIPCHK(hipStreamSynchronize(st));
    return GATO_OK;
}

// ---- statistics -------------------------------------------------------------------------------------------------------
static int sync_last(GatoSolver* s)
{
    {   // a deferred sharded solve whose verdict is still out: take it (and replay, if the rule fired) before anything looks at the state
        const int rc = settle(s);
        if (rc != GATO_OK) return rc;
    }
    // a solve enqueued on the caller's stream (gato_solve_device) may still be in flight: everything that reads or overwrites solver
    // state from the host waits for it first
    if (s->last_stream_valid) HIPCHK(hipStreamSynchronize(s->last_stream));
    return GATO_OK;
}
extern "C" int gato_synchronize(GatoSolver* s)
{
    if (!s) return fail(GATO_ERR_INVALID, "null solver");
    GUARD(s);
    return sync_last(s);
}
extern "C" int gato_get_counts(GatoSolver* s, uint32_t* iters_done, uint32_t* ls)
{
    if (!s) return fail(GATO_ERR_INVALID, "null solver");
    GUARD(s);
    int rc = sync_last(s);
    if (rc) return rc;
    Ctrl c;
    HIPCHK(hipMemcpy(&c, s->bf.ctrl, sizeof(Ctrl), hipMemcpyDeviceToHost));
    if (iters_done) *iters_done = c.iters_done;
    if (ls) *ls = c.ls_done;
    return GATO_OK;
}
extern "C" int gato_get_sqp_iters(GatoSolver* s, int32_t* out)
{
    if (!s || !out) return fail(GATO_ERR_INVALID, "null argument");
    uint32_t it = 0;
    int rc = gato_get_counts(s, &it, nullptr);
    if (rc) return rc;
    for (int b = 0; b < s->B; b++) out[b] = (int32_t)it;  // every trajectory counts every executed iteration (bsqp.cuh:153-162)
    return GATO_OK;
}
// one device array of the last solve -> host (synchronises the solver's stream first)
template<typename T> static int read_back(GatoSolver* s, T* out, const T* d_src, size_t count)
{
    if (!s || !out) return fail(GATO_ERR_INVALID, "null argument");
    GUARD(s);
    int rc = sync_last(s);
    if (rc) return rc;
    if (count) HIPCHK(hipMemcpy(out, d_src, count * sizeof(T), hipMemcpyDeviceToHost));
    return GATO_OK;
}
extern "C" int gato_get_kkt_converged(GatoSolver* s, int32_t* out) { return read_back(s, out, s ? s->bf.converged : nullptr, s ? s->B : 0); }
extern "C" int gato_get_final_merit(GatoSolver* s, float* out) { return read_back(s, out, s ? s->bf.merit_cur : nullptr, s ? s->B : 0); }
extern "C" int gato_get_initial_merit(GatoSolver* s, float* out) { return read_back(s, out, s ? s->d_merit_init0 : nullptr, s ? s->B : 0); }
extern "C" int gato_get_pcg_iters(GatoSolver* s, int32_t* out)
{
    if (!s || !out) return fail(GATO_ERR_INVALID, "null argument");
    uint32_t it = 0;
    int rc = gato_get_counts(s, &it, nullptr);
    if (rc) return rc;
    return read_back(s, out, s->bf.st_pcg_iters, (size_t)it * s->B);
}
extern "C" int gato_get_ls_min_merit(GatoSolver* s, float* out)
{
    if (!s || !out) return fail(GATO_ERR_INVALID, "null argument");
    uint32_t ls = 0;
    int rc = gato_get_counts(s, nullptr, &ls);
    if (rc) return rc;
    return read_back(s, out, s->bf.st_min_merit, (size_t)ls * s->B);
}
extern "C" int gato_get_ls_step_size(GatoSolver* s, float* out)
{
    if (!s || !out) return fail(GATO_ERR_INVALID, "null argument");
    uint32_t ls = 0;
    int rc = gato_get_counts(s, nullptr, &ls);
    if (rc) return rc;
    return read_back(s, out, s->bf.st_step, (size_t)ls * s->B);
}

// ---- setters (bsqp.cuh:63-89) -------------------------------------------------------------------------------------------
// host array -> per-trajectory device array, after the solve in flight (if any) is done with it
static int write_param(GatoSolver* s, float* d_dst, const float* v, size_t count, float* d_default = nullptr, std::vector<float>* h_default = nullptr)
{
    if (!s || !v) return fail(GATO_ERR_INVALID, "null argument");
    GUARD(s);
    int rc = sync_last(s);
    if (rc) return rc;
    if (d_default) {
        h_default->assign(v, v + count);
        HIPCHK(hipMemcpy(d_default, v, count * sizeof(float), hipMemcpyHostToDevice));
    }
    HIPCHK(hipMemcpy(d_dst, v, count * sizeof(float), hipMemcpyHostToDevice));
    return GATO_OK;
}
extern "C" int gato_set_f_ext_batch(GatoSolver* s, const float* v) { return write_param(s, s ? s->bf.f_ext : nullptr, v, s ? 6 * (size_t)s->B : 0); }
extern "C" int gato_set_rho_penalty_batch(GatoSolver* s, const float* v, int as_default)
{
    if (!s) return fail(GATO_ERR_INVALID, "null solver");
    return write_param(s, s->bf.rho, v, s->B, as_default ? s->d_rho_init : nullptr, &s->h_rho_init);
}
extern "C" int gato_set_drho_batch(GatoSolver* s, const float* v, int as_default)
{
    if (!s) return fail(GATO_ERR_INVALID, "null solver");
    return write_param(s, s->bf.drho, v, s->B, as_default ? s->d_drho_init : nullptr, &s->h_drho_init);
}
extern "C" int gato_set_mu_batch(GatoSolver* s, const float* v) { return write_param(s, s ? s->bf.mu : nullptr, v, s ? s->B : 0); }
extern "C" int gato_set_pcg_tol_batch(GatoSolver* s, const float* v) { return write_param(s, s ? s->bf.pcg_tol : nullptr, v, s ? s->B : 0); }
extern "C" int gato_set_cost_weights_batch(GatoSolver* s, const float* w)
{
    if (!s || !w) return fail(GATO_ERR_INVALID, "null argument");
    std::vector<float> p(8 * (size_t)s->B, 0.f);
    for (int b = 0; b < s->B; b++)
        for (int i = 0; i < 7; i++) p[8 * (size_t)b + i] = w[7 * (size_t)b + i];
    return write_param(s, s->d_costw, p.data(), p.size());
}
extern "C" int gato_reset_dual(GatoSolver* s)
{
    if (!s) return fail(GATO_ERR_INVALID, "null solver");
    GUARD(s);
    int rc = sync_last(s);
    if (rc) return rc;
    HIPCHK(hipMemset(s->bf.lambda, 0, (size_t)s->B * s->vecp * sizeof(float)));
    HIPCHK(hipStreamSynchronize(nullptr));   // a device memset is asynchronous to the host: done before a solve on ANY stream can follow
    return GATO_OK;
}
extern "C" int gato_reset_rho(GatoSolver* s)
{
    if (!s) return fail(GATO_ERR_INVALID, "null solver");
    GUARD(s);
    int rc = sync_last(s);
    if (rc) return rc;
    HIPCHK(hipMemcpy(s->bf.rho, s->h_rho_init.data(), s->B * sizeof(float), hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(s->bf.drho, s->h_drho_init.data(), s->B * sizeof(float), hipMemcpyHostToDevice));
    return GATO_OK;
}
extern "C" int gato_reset_async(GatoSolver* s, int dual, int rho, void* stream)
{
    if (!s) return fail(GATO_ERR_INVALID, "null solver");
    GUARD(s);
    hipStream_t st = (hipStream_t)stream;
    if (!dual && !rho) return GATO_OK;
    {
        const int rc = settle(s);   // the reset must come behind a pending solve's replay, not between its speculative run and the replay
        if (rc != GATO_OK) return rc;
    }
    // one launch for both
    const uint32_t n = dual ? (uint32_t)((size_t)s->B * s->vecp) : 0u;
    size_t blocks = ((size_t)n / 4 + 255) / 256;
    if (blocks < 1) blocks = 1;
    if (blocks > (size_t)s->cus * 8) blocks = (size_t)s->cus * 8;
    hipLaunchKernelGGL(reset_kernel, dim3((unsigned)blocks), dim3(256), 0, st, s->bf.lambda, n, rho ? s->bf.rho : (float*)nullptr,
                       (const float*)s->d_rho_init, s->bf.drho, (const float*)s->d_drho_init, s->B);
    HIPCHK(hipGetLastError());
    return GATO_OK;
}
extern "C" int gato_copy_final_merit_device(GatoSolver* s, float* d_out, void* stream)
{
    if (!s || !d_out) return fail(GATO_ERR_INVALID, "null argument");
    GUARD(s);
    {
        const int rc = settle(s);   // the merits of the solve that stands
        if (rc != GATO_OK) return rc;
    }
    HIPCHK(hipMemcpyAsync(d_out, s->bf.merit_cur, s->B * sizeof(float), hipMemcpyDeviceToDevice, (hipStream_t)stream));
    return GATO_OK;
}
extern "C" int gato_set_solved_count_mode(GatoSolver* s, int mode)
{
    if (!s) return fail(GATO_ERR_INVALID, "null solver");
    if (mode != GATO_COUNT_DEFERRED && mode != GATO_COUNT_PER_ITERATION) return fail(GATO_ERR_INVALID, "unknown solved-count mode (0 = per iteration, 1 = deferred)");
    GUARD(s);
    int rc = sync_last(s);
    if (rc) return rc;
    const int before = s->deferred_count;
    s->deferred_count = mode == GATO_COUNT_DEFERRED ? 1 : 0;
    // with a communicator this call is COLLECTIVE: every rank makes it, with the same mode.  On a disagreement every rank sees the same verdict
    // and every rank goes back to the mode it had -- which gato_comm_confirm had found equal on all of them: mixed modes (mismatched collectives:
    // a hang in the next sharded solve) cannot be left behind.  Any other failure is this rank's alone and also restores its mode.
    rc = agree_on_count_mode(s);
    if (rc != GATO_OK) { s->deferred_count = before; return rc; }
    s->periter_left = 0;
    s->replay_backoff = 8;
    return GATO_OK;
}
extern "C" int gato_get_shard_stats(GatoSolver* s, uint64_t* deferred_solves, uint64_t* replays)
{
    if (!s) return fail(GATO_ERR_INVALID, "null solver");
    if (deferred_solves) *deferred_solves = s->n_deferred;
    if (replays) *replays = s->n_replays;
    return GATO_OK;
}
extern "C" int gato_get_solved_count_state(GatoSolver* s, int* mode, uint64_t* per_iteration_solves, uint32_t* per_iteration_left)
{
    if (!s) return fail(GATO_ERR_INVALID, "null solver");
    if (mode) *mode = s->deferred_count ? GATO_COUNT_DEFERRED : GATO_COUNT_PER_ITERATION;
    if (per_iteration_solves) *per_iteration_solves = s->n_periter;
    if (per_iteration_left) *per_iteration_left = s->periter_left;
    return GATO_OK;
}
extern "C" int gato_set_graph_mode(GatoSolver* s, int enabled)
{
    if (!s) return fail(GATO_ERR_INVALID, "null solver");
    s->graph_mode = enabled ? 1 : 0;
    return GATO_OK;
}
extern "C" int gato_set_linear_solver(GatoSolver* s, int mode)
{
    if (!s) return fail(GATO_ERR_INVALID, "null solver");
    if (mode != GATO_LINSOLVE_PCG && mode != GATO_LINSOLVE_DIRECT) return fail(GATO_ERR_INVALID, "unknown linear solver (0 = PCG, 1 = direct)");
    if (mode == GATO_LINSOLVE_DIRECT) {
        GUARD(s);   // the grant is per device: the handle's
        const int rc = s->plant == GATO_PLANT_INDY7 ? grant_direct<Indy7>(s) : grant_direct<Iiwa14>(s);
        if (rc != GATO_OK) return rc;   // the solver stays on the linear solver it had
    }
    s->linear_solver = mode;
    return GATO_OK;
}
extern "C" int gato_set_rho_adaptation(GatoSolver* s, int enabled)
{
    if (!s) return fail(GATO_ERR_INVALID, "null solver");
    s->adapt_rho = enabled ? 1 : 0;
    return GATO_OK;
}

// ---- sim_forward / ee_pos -----------------------------------------------------------------------------------------------
// BSQP::sim_forward(T* d_xkp1_batch, T* d_xk, T* d_uk, T dt) (bsqp.cuh:91): device pointers, enqueued on `stream`, no synchronisation
extern "C" int gato_sim_forward_device(GatoSolver* s, float* d_xkp1, const float* d_xk, const float* d_uk, float dt, void* stream)
{
    if (!s || !d_xkp1 || !d_xk || !d_uk) return fail(GATO_ERR_INVALID, "null argument");
    GUARD(s);
    hipStream_t st = (hipStream_t)stream;
    if (s->plant == GATO_PLANT_INDY7)
        hipLaunchKernelGGL((sim_forward_kernel<Indy7>), dim3(cdiv(s->B, 256)), dim3(256), 0, st, d_xkp1, d_xk, d_uk, s->bf.f_ext, s->B, dt);
    else
        hipLaunchKernelGGL((sim_forward_kernel<Iiwa14>), dim3(cdiv(s->B, 256)), dim3(256), 0, st, d_xkp1, d_xk, d_uk, s->bf.f_ext, s->B, dt);
    HIPCHK(hipGetLastError());
    return GATO_OK;
}
// PyBSQP::sim_forward (bindings.cu:180-194): host arrays through the staging buffers the solver owns (no per-call allocation)
extern "C" int gato_sim_forward(GatoSolver* s, float* xkp1, const float* xk, const float* uk, float dt)
{
    if (!s || !xkp1 || !xk || !uk) return fail(GATO_ERR_INVALID, "null argument");
    GUARD(s);
    int rc = sync_last(s);  // the wrenches may have been set behind a solve that still reads them
    if (rc) return rc;
    HIPCHK(hipMemcpy(s->d_sim_x, xk, s->nx * sizeof(float), hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(s->d_sim_u, uk, s->nu * sizeof(float), hipMemcpyHostToDevice));
    rc = gato_sim_forward_device(s, s->d_sim_out, s->d_sim_x, s->d_sim_u, dt, nullptr);
    if (rc) return rc;
    HIPCHK(hipMemcpy(xkp1, s->d_sim_out, (size_t)s->B * s->nx * sizeof(float), hipMemcpyDeviceToHost));  // blocking on the null stream
    return GATO_OK;
}

// evaluate_best_trajectory of the MPC loop (mpc_controller.py:294-309) in one launch: x_next under every stored wrench hypothesis,
// err_b = |x_next_b - x_meas|_2 (fp32 on the device; the reference forms it in float64 on the host), best = the first arg-min.
extern "C" int gato_select_best_device(GatoSolver* s, const float* d_x_last, const float* d_u_last, const float* d_x_meas, float dt, int32_t* d_best,
                                       float* d_err, void* stream)
{
    if (!s || !d_x_last || !d_u_last || !d_x_meas || !d_best || !d_err) return fail(GATO_ERR_INVALID, "null argument");
    GUARD(s);
    hipStream_t st = (hipStream_t)stream;
    // one completion counter per handle, cleared on the call's stream ahead of every launch (a launch that faulted, or two calls in flight,
    // cannot leave a count behind); x_next goes to the solver-owned d_sim_out: the _device entry is single-stream per handle
    uint32_t* cnt = reinterpret_cast<uint32_t*>(s->d_sel_best + 1);
    HIPCHK(hipMemsetAsync(cnt, 0, sizeof(uint32_t), st));
    if (s->plant == GATO_PLANT_INDY7)
        hipLaunchKernelGGL((select_best_kernel<Indy7>), dim3(cdiv(s->B, 256)), dim3(256), 0, st, s->d_sim_out, d_err, d_best, cnt, d_x_last, d_u_last, d_x_meas,
                           s->bf.f_ext, s->B, dt);
    else
        hipLaunchKernelGGL((select_best_kernel<Iiwa14>), dim3(cdiv(s->B, 256)), dim3(256), 0, st, s->d_sim_out, d_err, d_best, cnt, d_x_last, d_u_last, d_x_meas,
                           s->bf.f_ext, s->B, dt);
    HIPCHK(hipGetLastError());
    return GATO_OK;
}
extern "C" int gato_select_best(GatoSolver* s, const float* x_last, const float* u_last, const float* x_meas, float dt, int* best, float* errors)
{
    if (!s || !x_last || !u_last || !x_meas || !best) return fail(GATO_ERR_INVALID, "null argument");
    GUARD(s);
    int rc = sync_last(s);
    if (rc) return rc;
    HIPCHK(hipMemcpy(s->d_sim_x, x_last, s->nx * sizeof(float), hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(s->d_sim_u, u_last, s->nu * sizeof(float), hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(s->d_sel_xm, x_meas, s->nx * sizeof(float), hipMemcpyHostToDevice));
    rc = gato_select_best_device(s, s->d_sim_x, s->d_sim_u, s->d_sel_xm, dt, s->d_sel_best, s->d_sel_err, nullptr);
    if (rc) return rc;
    int32_t b = 0;
    HIPCHK(hipMemcpy(&b, s->d_sel_best, sizeof(int32_t), hipMemcpyDeviceToHost));
    *best = (int)b;
    if (errors) HIPCHK(hipMemcpy(errors, s->d_sel_err, (size_t)s->B * sizeof(float), hipMemcpyDeviceToHost));
    return GATO_OK;
}

// The plant simulator of the MPC loop (python/bsqp/common.py:49-91 `rk4`, called 1 kHz from mpc_controller.py:199-218): nsteps RK4
// steps of the library's own forward dynamics from x (host, [nx], updated in place) with control u_seq[step] (host, [nsteps][nu]) and a
// constant spatial wrench [angular; linear] on the last link in its local frame.
extern "C" int gato_plant_rk4(GatoSolver* s, float* x, const float* u_seq, int nsteps, const float* f_ext6, float sim_dt)
{
    if (!s || !x || !f_ext6 || nsteps < 0 || (nsteps > 0 && !u_seq)) return fail(GATO_ERR_INVALID, "bad argument");
    if (nsteps == 0) return GATO_OK;
    GUARD(s);
    const size_t need = (size_t)s->nx + 6 + (size_t)nsteps * s->nu;
    if (need > s->plant_cap) {
        if (s->d_plant) (void)hipFree(s->d_plant);
        s->d_plant = nullptr;
        s->plant_cap = 0;
        HIPCHK(hipMalloc((void**)&s->d_plant, (need + 64 * s->nu) * sizeof(float)));
        s->plant_cap = need + 64 * s->nu;
    }
    float *d_x = s->d_plant, *d_f = d_x + s->nx, *d_u = d_f + 6;
    std::vector<float> h(need);
    memcpy(h.data(), x, s->nx * sizeof(float));
    memcpy(h.data() + s->nx, f_ext6, 6 * sizeof(float));
    memcpy(h.data() + s->nx + 6, u_seq, (size_t)nsteps * s->nu * sizeof(float));
    HIPCHK(hipMemcpy(d_x, h.data(), need * sizeof(float), hipMemcpyHostToDevice));
    if (s->plant == GATO_PLANT_INDY7) hipLaunchKernelGGL((plant_rk4_kernel<Indy7>), dim3(1), dim3(64), 0, nullptr, d_x, d_u, d_f, nsteps, sim_dt, 1);
    else hipLaunchKernelGGL((plant_rk4_kernel<Iiwa14>), dim3(1), dim3(64), 0, nullptr, d_x, d_u, d_f, nsteps, sim_dt, 1);
    HIPCHK(hipGetLastError());
    HIPCHK(hipMemcpy(x, d_x, s->nx * sizeof(float), hipMemcpyDeviceToHost));
    return GATO_OK;
}

// The same plant carrying a swinging payload (MPC_GATO's pendulum_config: mpc_controller.py:44-60, 340-360, 472-478; kernels.hpp
// payload_dynamics): pend11 = [quat x y z w | w (3) | mass, length, damping, inertia], the first seven updated in place.
extern "C" int gato_plant_payload_rk4(GatoSolver* s, float* x, float* pend11, const float* u_seq, int nsteps, const float* f_ext6, float sim_dt)
{
    if (!s || !x || !pend11 || !f_ext6 || nsteps < 0 || (nsteps > 0 && !u_seq)) return fail(GATO_ERR_INVALID, "bad argument");
    if (!(pend11[7] > 0) || !(pend11[8] > 0) || !(pend11[10] > 0)) return fail(GATO_ERR_INVALID, "payload mass, length and inertia must be positive");
    if (nsteps == 0) return GATO_OK;
    GUARD(s);
    const size_t need = (size_t)s->nx + 6 + 12 + (size_t)nsteps * s->nu;
    if (need > s->plant_cap) {
        if (s->d_plant) (void)hipFree(s->d_plant);
        s->d_plant = nullptr;
        s->plant_cap = 0;
        HIPCHK(hipMalloc((void**)&s->d_plant, (need + 64 * s->nu) * sizeof(float)));
        s->plant_cap = need + 64 * s->nu;
    }
    float *d_x = s->d_plant, *d_f = d_x + s->nx, *d_p = d_f + 6, *d_u = d_p + 12;
    std::vector<float> h(need);
    memcpy(h.data(), x, s->nx * sizeof(float));
    memcpy(h.data() + s->nx, f_ext6, 6 * sizeof(float));
    memcpy(h.data() + s->nx + 6, pend11, 11 * sizeof(float));
    memcpy(h.data() + s->nx + 18, u_seq, (size_t)nsteps * s->nu * sizeof(float));
    HIPCHK(hipMemcpy(d_x, h.data(), need * sizeof(float), hipMemcpyHostToDevice));
    if (s->plant == GATO_PLANT_INDY7) hipLaunchKernelGGL((plant_rk4_kernel<Indy7>), dim3(1), dim3(64), 0, nullptr, d_x, d_u, d_f, nsteps, sim_dt, 1, d_p);
    else hipLaunchKernelGGL((plant_rk4_kernel<Iiwa14>), dim3(1), dim3(64), 0, nullptr, d_x, d_u, d_f, nsteps, sim_dt, 1, d_p);
    HIPCHK(hipGetLastError());
    HIPCHK(hipMemcpy(h.data(), d_x, ((size_t)s->nx + 6 + 7) * sizeof(float), hipMemcpyDeviceToHost));
    memcpy(x, h.data(), s->nx * sizeof(float));
    memcpy(pend11, h.data() + s->nx + 6, 7 * sizeof(float));
    return GATO_OK;
}

// World placements of the joint frames (pinocchio's data.oMi[1..nq], which MPC_GATO.transform_force_to_gato_frame reads,
// mpc_controller.py:311-338) from the library's own kinematic tables: oMi_k = oMi_{k-1} [R0_k Rz(q_k) | r_k], R0_k = E0_k^T.
// Host code, float64, no device: out[k] = {R row-major (9), p (3)}.
template<class M> static void fk_placements(const float* q, double* out)
{
    double R[3][3] = {{1, 0, 0}, {0, 1, 0}, {0, 0, 1}}, p[3] = {0, 0, 0};
    for (int k = 0; k < M::NQ; k++) {
        const double c = cos((double)q[k]), sn = sin((double)q[k]);
        double L[3][3], Rn[3][3], pn[3];  // L = E0^T Rz(q)
        for (int i = 0; i < 3; i++) {
            const double a = M::E0[k][0][i], b = M::E0[k][1][i], cc = M::E0[k][2][i];  // row i of E0^T
            L[i][0] = a * c + b * sn;
            L[i][1] = -a * sn + b * c;
            L[i][2] = cc;
        }
        for (int i = 0; i < 3; i++) {
            pn[i] = p[i] + R[i][0] * M::R[k][0] + R[i][1] * M::R[k][1] + R[i][2] * M::R[k][2];
            for (int j = 0; j < 3; j++) Rn[i][j] = R[i][0] * L[0][j] + R[i][1] * L[1][j] + R[i][2] * L[2][j];
        }
        for (int i = 0; i < 3; i++) {
            p[i] = pn[i];
            for (int j = 0; j < 3; j++) { R[i][j] = Rn[i][j]; out[12 * k + 3 * i + j] = Rn[i][j]; }
            out[12 * k + 9 + i] = pn[i];
        }
    }
}
extern "C" int gato_fk_placements(int plant, const float* q, double* out)
{
    if (!q || !out) return fail(GATO_ERR_INVALID, "null argument");
    if (plant == GATO_PLANT_INDY7) fk_placements<Indy7>(q, out);
    else if (plant == GATO_PLANT_IIWA14) fk_placements<Iiwa14>(q, out);
    else return fail(GATO_ERR_INVALID, "unknown plant");
    return GATO_OK;
}

extern "C" int gato_ee_pos(GatoSolver* s, const float* q, int n, float* out)
{
    if (!s || !q || !out || n < 0) return fail(GATO_ERR_INVALID, "bad argument");
    if (n == 0) return GATO_OK;
    GUARD(s);
    if ((size_t)n > s->ee_cap) {  // staging grows to the largest request seen (the MPC loop asks for one configuration per step)
        if (s->d_ee_q) (void)hipFree(s->d_ee_q);
        if (s->d_ee_out) (void)hipFree(s->d_ee_out);
        s->d_ee_q = s->d_ee_out = nullptr;
        s->ee_cap = 0;
        HIPCHK(hipMalloc((void**)&s->d_ee_q, (size_t)n * s->nq * sizeof(float)));
        HIPCHK(hipMalloc((void**)&s->d_ee_out, (size_t)n * 3 * sizeof(float)));
        s->ee_cap = (size_t)n;
    }
    HIPCHK(hipMemcpy(s->d_ee_q, q, (size_t)n * s->nq * sizeof(float), hipMemcpyHostToDevice));
    if (s->plant == GATO_PLANT_INDY7) hipLaunchKernelGGL((ee_pos_kernel<Indy7>), dim3(cdiv(n, 256)), dim3(256), 0, nullptr, s->d_ee_out, s->d_ee_q, n);
    else hipLaunchKernelGGL((ee_pos_kernel<Iiwa14>), dim3(cdiv(n, 256)), dim3(256), 0, nullptr, s->d_ee_out, s->d_ee_q, n);
    HIPCHK(hipGetLastError());
    HIPCHK(hipMemcpy(out, s->d_ee_out, (size_t)n * 3 * sizeof(float), hipMemcpyDeviceToHost));
    return GATO_OK;
}

// ---- MPC session (include/gato_abi.h) -------------------------------------------------------------------------------------------------
template<class M> static int mpc_begin_impl(GatoSolver* s, const float* x0)
{
    hipStream_t st = s->own_stream;
    HIPCHK(hipMemcpyAsync(s->d_sel_xm, x0, s->nx * sizeof(float), hipMemcpyHostToDevice, st));   // staging for x0
    hipLaunchKernelGGL((mpc_warm_kernel<M>), dim3(s->B + 1), dim3(256), 0, st, s->d_xu_own, s->d_mpc_best, s->d_mpc_x, s->d_mpc_xlast, (const float*)s->d_sel_xm, s->traj, s->B);
    HIPCHK(hipMemsetAsync(s->bf.lambda, 0, (size_t)s->B * s->vecp * sizeof(float), st));         // reset_dual (mpc_controller.py:172)
    HIPCHK(hipGetLastError());
    HIPCHK(hipStreamSynchronize(st));
    return GATO_OK;
}
extern "C" int gato_mpc_begin(GatoSolver* s, const float* x0)
{
    if (!s || !x0) return fail(GATO_ERR_INVALID, "null argument");
    GUARD(s);
    int rc = sync_last(s);
    if (rc) return rc;
    if (!s->mpc_ev0) {
        HIPCHK(hipEventCreate(&s->mpc_ev0)); HIPCHK(hipEventCreate(&s->mpc_ev1));
        HIPCHK(hipEventCreate(&s->mpc_ev2)); HIPCHK(hipEventCreate(&s->mpc_ev3));
    }
    if (!s->h_mpc_in) {
        HIPCHK(hipHostMalloc((void**)&s->h_mpc_in, (8 + 6 * (size_t)s->N + 6 * (size_t)s->B) * sizeof(float), hipHostMallocDefault));
        HIPCHK(hipHostMalloc((void**)&s->h_mpc_out, (20 + (size_t)s->B) * sizeof(float), hipHostMallocDefault));
    }
    rc = s->plant == GATO_PLANT_INDY7 ? mpc_begin_impl<Indy7>(s, x0) : mpc_begin_impl<Iiwa14>(s, x0);
    if (rc == GATO_OK) s->mpc_begun = true;
    return rc;
}
template<class M> static int mpc_step_impl(GatoSolver* s, GatoMpcStep* io)
{
    hipStream_t st = s->own_stream;
    // ADVANCE with plant_steps == 0 still runs the plant launch: it is what takes x_last := x (mpc_controller.py:196-197, every loop iteration)
    const bool advance = (io->phases & GATO_MPC_ADVANCE) != 0, plan = (io->phases & GATO_MPC_PLAN) != 0;
    if (advance && io->plant_steps > 0 && !(io->steps_per_knot > 0.0)) return fail(GATO_ERR_INVALID, "steps_per_knot must be positive");
    if (plan && !io->ref_window) return fail(GATO_ERR_INVALID, "ref_window is required for GATO_MPC_PLAN");
    // ONE host-to-device copy from pinned memory: [wrench | reference window | hypotheses] (only as far as this step needs)
    {
        const size_t nw = 6 * (size_t)s->N, nh = 6 * (size_t)s->B;
        for (int i = 0; i < 6; i++) s->h_mpc_in[i] = advance ? io->plant_wrench[i] : (float)0;
        size_t n = 8;
        if (plan) {
            memcpy(s->h_mpc_in + 8, io->ref_window, nw * sizeof(float));
            n += nw;
            if (io->hyp_world) { memcpy(s->h_mpc_in + 8 + nw, io->hyp_world, nh * sizeof(float)); n += nh; }
        }
        if (advance || plan) HIPCHK(hipMemcpyAsync(s->d_mpc_fw, s->h_mpc_in, n * sizeof(float), hipMemcpyHostToDevice, st));
    }
    if (advance) {
        HIPCHK(hipEventRecord(s->mpc_ev2, st));
        hipLaunchKernelGGL((mpc_plant_kernel<M>), dim3(1), dim3(64), 0, st, s->d_mpc_x, s->d_mpc_xlast, (const float*)s->d_mpc_best, (const float*)s->d_mpc_fw,
                           (int)io->plant_steps, io->sim_dt, io->plant_steps > 0 ? io->steps_per_knot : 1.0, s->N, s->mpc_payload ? s->d_mpc_pend : (float*)nullptr);
        HIPCHK(hipEventRecord(s->mpc_ev3, st));
    }
    const bool selecting = plan && io->select && s->B > 1;
    if (plan) {
        hipLaunchKernelGGL((mpc_prepare_kernel<M>), dim3(s->B), dim3(256), 0, st, s->d_xu_own, s->d_xs_own, s->d_ref_own, s->bf.f_ext, (const float*)s->d_mpc_best,
                           (const float*)s->d_mpc_x, (const float*)s->d_mpc_refw, io->hyp_world ? (const float*)s->d_mpc_hyp : (const float*)nullptr, s->N, s->traj);
        // reset_rho ahead of every solve (mpc_controller.py:229)
        HIPCHK(hipMemcpyAsync(s->bf.rho, s->d_rho_init, s->B * sizeof(float), hipMemcpyDeviceToDevice, st));
        HIPCHK(hipMemcpyAsync(s->bf.drho, s->d_drho_init, s->B * sizeof(float), hipMemcpyDeviceToDevice, st));
        // GATO_MPC_TIME_SOLVE: the reference's sqp_time_us (bsqp.cuh:109,185) on this solve -- host clock, device-synchronised on both sides
        const bool wall = (io->phases & GATO_MPC_TIME_SOLVE) != 0;
        std::chrono::high_resolution_clock::time_point w0;
        if (wall) { HIPCHK(hipStreamSynchronize(st)); w0 = std::chrono::high_resolution_clock::now(); }
        HIPCHK(hipEventRecord(s->mpc_ev0, st));
        const int rc = solve_impl<M>(s, s->d_xu_own, s->p.dt, s->d_xs_own, s->d_ref_own, st);
        if (rc != GATO_OK) return rc;
        HIPCHK(hipEventRecord(s->mpc_ev1, st));
        if (wall) {
            HIPCHK(hipStreamSynchronize(st));
            io->solve_wall_us = std::chrono::duration<double, std::micro>(std::chrono::high_resolution_clock::now() - w0).count();
        }
        if (selecting) {
            // evaluate_best_trajectory (mpc_controller.py:294-309): the PREVIOUS state and the previous best trajectory's first control under
            // every hypothesis against the state just measured
            uint32_t* cnt = reinterpret_cast<uint32_t*>(s->d_sel_best + 1);
            HIPCHK(hipMemsetAsync(cnt, 0, sizeof(uint32_t), st));
            hipLaunchKernelGGL((select_best_kernel<M>), dim3(cdiv(s->B, 256)), dim3(256), 0, st, s->d_sim_out, s->d_mpc_err, s->d_sel_best, cnt,
                               (const float*)s->d_mpc_xlast, (const float*)(s->d_mpc_best + s->nx), (const float*)s->d_mpc_x, (const float*)s->bf.f_ext, s->B,
                               io->select_dt);
        }
    }
    // best row (after a plan) + the record, one launch; ONE device-to-host copy into pinned memory
    hipLaunchKernelGGL((mpc_finish_kernel<M>), dim3(plan ? cdiv(s->traj, 256) : 1), dim3(256), 0, st, s->d_mpc_best, (const float*)s->d_xu_own,
                       selecting ? (const int32_t*)s->d_sel_best : (const int32_t*)nullptr, s->traj, s->B, plan ? 1 : 0, s->d_mpc_rec, (const float*)s->d_mpc_x);
    HIPCHK(hipGetLastError());
    const bool want_err = selecting && io->errors;
    HIPCHK(hipMemcpyAsync(s->h_mpc_out, s->d_mpc_rec, (20 + (want_err ? (size_t)s->B : 0)) * sizeof(float), hipMemcpyDeviceToHost, st));
    HIPCHK(hipStreamSynchronize(st));
    const float* rec = s->h_mpc_out;
    for (int i = 0; i < s->nx; i++) io->x[i] = rec[i];
    for (int i = 0; i < 3; i++) io->ee[i] = rec[s->nx + i];
    io->best = plan ? (int32_t)rec[s->nx + 3] : 0;
    io->solve_us = 0.0;
    io->plant_us = 0.0;
    if (!(plan && (io->phases & GATO_MPC_TIME_SOLVE))) io->solve_wall_us = 0.0;
    if (advance) {
        f32_t ms = 0;
        HIPCHK(hipEventElapsedTime(&ms, s->mpc_ev2, s->mpc_ev3));
        io->plant_us = (double)ms * 1e3;
    }
    if (plan) {
        f32_t ms = 0;
        HIPCHK(hipEventElapsedTime(&ms, s->mpc_ev0, s->mpc_ev1));
        io->solve_us = (double)ms * 1e3;
    }
    if (io->errors)
        for (int b = 0; b < s->B; b++) io->errors[b] = want_err ? rec[20 + b] : (float)0;
    return GATO_OK;
}
extern "C" int gato_mpc_step(GatoSolver* s, GatoMpcStep* io)
{
    if (!s || !io) return fail(GATO_ERR_INVALID, "null argument");
    if (io->struct_size != sizeof(GatoMpcStep)) return fail(GATO_ERR_INVALID, "GatoMpcStep::struct_size is not sizeof(GatoMpcStep) of this library: the client was compiled against another gato_abi.h");
    if (!s->mpc_begun) return fail(GATO_ERR_INVALID, "gato_mpc_begin has not been called on this solver");
    if (!(io->phases & (GATO_MPC_ADVANCE | GATO_MPC_PLAN))) return fail(GATO_ERR_INVALID, "phases: GATO_MPC_ADVANCE and / or GATO_MPC_PLAN");
    if (io->plant_steps < 0) return fail(GATO_ERR_INVALID, "plant_steps must not be negative");
    GUARD(s);
    int rc = sync_last(s);
    if (rc) return rc;
    return s->plant == GATO_PLANT_INDY7 ? mpc_step_impl<Indy7>(s, io) : mpc_step_impl<Iiwa14>(s, io);
}
// the session's plant carries a payload from now on (pend11 as in gato_plant_payload_rk4), or none (NULL)
extern "C" int gato_mpc_set_payload(GatoSolver* s, const float* pend11)
{
    if (!s) return fail(GATO_ERR_INVALID, "null argument");
    GUARD(s);
    int rc = sync_last(s);
    if (rc) return rc;
    if (!pend11) { s->mpc_payload = false; return GATO_OK; }
    if (!(pend11[7] > 0) || !(pend11[8] > 0) || !(pend11[10] > 0)) return fail(GATO_ERR_INVALID, "payload mass, length and inertia must be positive");
    HIPCHK(hipMemcpyAsync(s->d_mpc_pend, pend11, 11 * sizeof(float), hipMemcpyHostToDevice, s->own_stream));
    HIPCHK(hipStreamSynchronize(s->own_stream));
    s->mpc_payload = true;
    return GATO_OK;
}
extern "C" int gato_mpc_get_payload(GatoSolver* s, float* pend7)
{
    if (!s || !pend7) return fail(GATO_ERR_INVALID, "null argument");
    GUARD(s);
    if (!s->mpc_payload) return fail(GATO_ERR_INVALID, "the session's plant carries no payload (gato_mpc_set_payload)");
    HIPCHK(hipMemcpyAsync(pend7, s->d_mpc_pend, 7 * sizeof(float), hipMemcpyDeviceToHost, s->own_stream));
    HIPCHK(hipStreamSynchronize(s->own_stream));
    return GATO_OK;
}
extern "C" int gato_mpc_get_best(GatoSolver* s, float* xu_best)
{
    if (!s || !xu_best) return fail(GATO_ERR_INVALID, "null argument");
    if (!s->mpc_begun) return fail(GATO_ERR_INVALID, "gato_mpc_begin has not been called on this solver");
    GUARD(s);
    int rc = sync_last(s);
    if (rc) return rc;
    HIPCHK(hipMemcpy(xu_best, s->d_mpc_best, (size_t)s->traj * sizeof(float), hipMemcpyDeviceToHost));
    return GATO_OK;
}

// ---- debug / test access ------------------------------------------------------------------------------------------------
static float* find_buf(GatoSolver* s, const char* name, uint64_t* len)
{
    const size_t BN = (size_t)s->B * s->N;
    const int nq = s->nq, nx = s->nx, nu = s->nu;
    struct E { const char* n; float* p; uint64_t l; };
    const E tab[] = {
        {"D", s->bf.D, BN * 3 * nq * nq}, {"Qq", s->bf.Qq, BN * nq * nq}, {"Qd", s->bf.Qd, BN * nq}, {"Rd", s->bf.Rd, BN * nu},
        {"q", s->bf.q, BN * nx}, {"r", s->bf.r, BN * nu}, {"c", s->bf.c, BN * nx}, {"Qqi", s->bf.Qqi, BN * nq * nq}, {"Qdi", s->bf.Qdi, BN * nq},
        {"Rdi", s->bf.Rdi, BN * nu}, {"S", s->bf.S, BN * s->brow}, {"Pinv", s->bf.Pinv, BN * s->brow},
        {"gamma", s->bf.gamma, (uint64_t)s->B * s->vecp}, {"lambda", s->bf.lambda, (uint64_t)s->B * s->vecp},
        {"dz", s->bf.dz, (uint64_t)s->B * s->traj}, {"merit", s->bf.merit, (uint64_t)s->B * NUM_ALPHAS}, {"merit_cur", s->bf.merit_cur, (uint64_t)s->B},
        {"rho", s->bf.rho, (uint64_t)s->B}, {"drho", s->bf.drho, (uint64_t)s->B}, {"step", s->bf.step, (uint64_t)s->B},
        {"mu", s->bf.mu, (uint64_t)s->B}, {"pcg_tol", s->bf.pcg_tol, (uint64_t)s->B}, {"f_ext", s->bf.f_ext, (uint64_t)s->B * 6},
        {"xu", s->d_xu_own, (uint64_t)s->B * s->traj},  // the solver's own copy (gato_solve / gato_debug_stage), not a caller's device buffer
    };
    for (const E& e : tab)
        if (!strcmp(e.n, name)) { *len = e.l; return e.p; }
    return nullptr;
}
extern "C" int gato_debug_read(GatoSolver* s, const char* name, float* out, uint64_t count, uint64_t* len)
{
    if (!s || !name) return fail(GATO_ERR_INVALID, "null argument");
    GUARD(s);
    uint64_t l = 0;
    float* p = find_buf(s, name, &l);
    if (!p) {
        if (!strcmp(name, "pcg_iters") || !strcmp(name, "converged") || !strcmp(name, "order")) {  // integer buffers, returned as floats
            std::vector<int32_t> t(s->B);
            HIPCHK(hipDeviceSynchronize());
            HIPCHK(hipMemcpy(t.data(), !strcmp(name, "pcg_iters") ? (void*)s->bf.pcg_iters : (!strcmp(name, "order") ? (void*)s->d_order : (void*)s->bf.converged), s->B * sizeof(int32_t),
                             hipMemcpyDeviceToHost));
            if (len) *len = s->B;
            if (out) for (uint64_t i = 0; i < count && i < (uint64_t)s->B; i++) out[i] = (float)t[i];
            return GATO_OK;
        }
        return fail(GATO_ERR_INVALID, std::string("unknown buffer ") + name);
    }
    if (len) *len = l;
    if (!out) return GATO_OK;
    if (count > l) return fail(GATO_ERR_INVALID, "count exceeds buffer length");
    HIPCHK(hipDeviceSynchronize());
    if (!strcmp(name, "S") || !strcmp(name, "Pinv")) {
        // the device keeps these block-major ([k][left | main | right][row][col]); the debug view is the reference's layout
        // ([k][row][left | main | right], linalg.cuh:663-666), whole block rows only
        const size_t nx = s->nx, blk = nx * nx, brow = 3 * blk;
        if (count % brow) return fail(GATO_ERR_INVALID, "S / Pinv are read in whole block rows (3 nx^2 floats)");
        std::vector<float> t(count);
        HIPCHK(hipMemcpy(t.data(), p, count * sizeof(float), hipMemcpyDeviceToHost));
        for (size_t r = 0; r < count / brow; r++)
            for (size_t j = 0; j < 3; j++)
                for (size_t y = 0; y < nx; y++)
                    memcpy(out + r * brow + y * 3 * nx + j * nx, t.data() + r * brow + j * blk + y * nx, nx * sizeof(float));
        return GATO_OK;
    }
    HIPCHK(hipMemcpy(out, p, count * sizeof(float), hipMemcpyDeviceToHost));
    return GATO_OK;
}
extern "C" int gato_debug_write(GatoSolver* s, const char* name, const float* in, uint64_t count)
{
    if (!s || !name || !in) return fail(GATO_ERR_INVALID, "null argument");
    GUARD(s);
    uint64_t l = 0;
    float* p = find_buf(s, name, &l);
    if (!p) {
        if (!strcmp(name, "converged")) {
            std::vector<int32_t> t(s->B, 0);
            for (uint64_t i = 0; i < count && i < (uint64_t)s->B; i++) t[i] = (int32_t)in[i];
            HIPCHK(hipMemcpy(s->bf.converged, t.data(), s->B * sizeof(int32_t), hipMemcpyHostToDevice));
            return GATO_OK;
        }
        return fail(GATO_ERR_INVALID, std::string("unknown buffer ") + name);
    }
    if (count > l) return fail(GATO_ERR_INVALID, "count exceeds buffer length");
    if (!strcmp(name, "S") || !strcmp(name, "Pinv")) {   // reference layout in, block-major on the device (see gato_debug_read)
        const size_t nx = s->nx, blk = nx * nx, brow = 3 * blk;
        if (count % brow) return fail(GATO_ERR_INVALID, "S / Pinv are written in whole block rows (3 nx^2 floats)");
        std::vector<float> t(count);
        for (size_t r = 0; r < count / brow; r++)
            for (size_t j = 0; j < 3; j++)
                for (size_t y = 0; y < nx; y++)
                    memcpy(t.data() + r * brow + j * blk + y * nx, in + r * brow + y * 3 * nx + j * nx, nx * sizeof(float));
        HIPCHK(hipMemcpy(p, t.data(), count * sizeof(float), hipMemcpyHostToDevice));
        return GATO_OK;
    }
    HIPCHK(hipMemcpy(p, in, count * sizeof(float), hipMemcpyHostToDevice));
    return GATO_OK;
}

template<class M> static int stage_impl(GatoSolver* s, int stage, float dt, float* out)
{
    Buffers& bf = s->bf;
    hipStream_t st = nullptr;
    HIPCHK(hipMemsetAsync(bf.ctrl, 0, sizeof(Ctrl), st));
    HIPCHK(hipMemsetAsync(bf.num_solved, 0, s->max_iters_alloc * sizeof(uint32_t), st));
    HIPCHK(hipMemsetAsync(s->d_ns_local, 0, s->max_iters_alloc * sizeof(uint32_t), st));
    switch (stage) {
        case 0: launch_merit<M>(s, st, NUM_ALPHAS, dt, 1, 0, bf.merit); break;
        case 1: launch_kkt<M>(s, st, dt, 0); break;
        case 2: launch_schur<M>(s, st, dt, true); break;   // stage tests read the complete P^-1
        case 3: launch_pcg<M>(s, st, 0, 1); break;
        case 7: launch_direct<M>(s, st, 0); break;
        case 4: launch_dz<M>(s, st, dt, 0); break;
        case 5: launch_ls(s, st, 0, 0); break;
        case 6: launch_merit<M>(s, st, 1, dt, 0, -1, bf.merit_cur); break;
        default: return fail(GATO_ERR_INVALID, "unknown stage");
    }
    HIPCHK(hipGetLastError());
    HIPCHK(hipDeviceSynchronize());
    (void)out;
    return GATO_OK;
}
extern "C" int gato_debug_stage(GatoSolver* s, int stage, float* xu, float dt, const float* x_s, const float* ref, float* out)
{
    if (!s || !xu || !x_s || !ref) return fail(GATO_ERR_INVALID, "null argument");
    GUARD(s);
    const size_t nxu = (size_t)s->B * s->traj * sizeof(float);
    HIPCHK(hipMemcpy(s->d_xu_own, xu, nxu, hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(s->d_xs_own, x_s, (size_t)s->B * s->nx * sizeof(float), hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(s->d_ref_own, ref, (size_t)s->B * 6 * s->N * sizeof(float), hipMemcpyHostToDevice));
    s->bf.xu = s->d_xu_own; s->bf.x_s = s->d_xs_own; s->bf.ref = s->d_ref_own;
    s->last_stream = nullptr;
    s->last_stream_valid = true;
    int rc = s->plant == GATO_PLANT_INDY7 ? stage_impl<Indy7>(s, stage, dt, out) : stage_impl<Iiwa14>(s, stage, dt, out);
    if (rc) return rc;
    HIPCHK(hipMemcpy(xu, s->d_xu_own, nxu, hipMemcpyDeviceToHost));
    return GATO_OK;
}

extern "C" int gato_set_profiling(GatoSolver* s, int enabled)
{
    if (!s) return fail(GATO_ERR_INVALID, "null solver");
    s->profiling = enabled ? 1 : 0;
    return GATO_OK;
}
extern "C" int gato_get_stage_times_us(GatoSolver* s, double* out7)
{
    if (!s || !out7) return fail(GATO_ERR_INVALID, "null argument");
    GUARD(s);
    int rc = sync_last(s);
    if (rc) return rc;
    collect_profile(s);
    for (int i = 0; i <= ST_COUNT; i++) out7[i] = s->stage_us[i];
    return GATO_OK;
}

extern "C" const char* gato_last_error(void) { return g_err.c_str(); }
// GATO_SRC_HASH: sha256 (first 16 hex digits) of kernels.hpp | rbd.hpp | solver.hip | robot_models.hpp as the Makefile saw them when THIS binary was
// built (tools/source_hash.py, the same function bench.py applies to the tree it runs from): a stale .so says so on the bench line
#ifndef GATO_SRC_HASH
#define GATO_SRC_HASH "unknown"
#endif
extern "C" const char* gato_version(void) { return "gato_amd 0.1.0 (gfx950) src " GATO_SRC_HASH; }
extern "C" const char* gato_source_hash(void) { return GATO_SRC_HASH; }
extern "C" int gato_abi_version(void) { return GATO_ABI_VERSION; }
extern "C" int gato_abi_real_size(void) { return (int)sizeof(float); }   // `float` is the real type here (real.hpp)
static_assert(sizeof(GatoParams) == (kDouble ? 15 * 8 : 15 * 4), "GatoParams: 13 reals + 2 uint32 (padded to reals in the float64 build)");
static_assert(sizeof(((GatoMpcStep*)nullptr)->x) == 16 * sizeof(float), "GatoMpcStep carries the library's real type");
