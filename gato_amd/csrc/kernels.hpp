// kernels.hpp -- CDNA4 (gfx950) kernels of the batched SQP iteration.  Included once per plant by solver.hip.
//
// Mapping (MI355X-first, not the reference's block-per-knot scheme) -- three launches per SQP iteration:
//   * kkt_kernel: ONE LANE per (trajectory, knot) with the rigid-body recursion in registers (rbd.hpp); the work of 64 knots is
//     split by TASK over the wavefronts of a workgroup (pairs of derivative columns | cost blocks); D leaves through LDS, coalesced;
//   * pcgc_kernel<.., FUSE>: one workgroup per trajectory, three rows of the block-tridiagonal system per thread.  The Schur
//     complement is FORMED here by 4-lane groups (DPP pivot broadcast) and stays in registers together with the stair preconditioner
//     for the whole PCG solve: S and P^-1 never reach global memory (the reference re-reads both on every PCG iteration,
//     pcg.cuh:100,119); vectors are exchanged through LDS, dot products are DPP + one LDS slot per wavefront; its PAIR form gives a
//     row group to two lanes (half of the columns each, same bits) where the whole batch is resident that way; pcgs_kernel keeps
//     long horizons (iiwa14 / indy7 N = 128) on the CU in symmetric half storage;
//   * step_kernel: one workgroup per trajectory: dz -> merit at the 8 step sizes (lane = (alpha, knot), wave-butterfly sums:
//     deterministic, no float atomics) -> line search -> xu, rho; the first one of a solve also forms the initial merit;
//   * stand-alone forms of every stage (schurq/schur1/schur2, pcg(c)_kernel, dz, merit<8>, line_search) serve the stage tests, iiwa14
//     (nx = 14) and N > 64;
//   * the SQP loop has no host round trip: convergence counting and the solve_ratio early exit run on the device (Ctrl).
//
// Global layouts are trajectory-major like the reference's (linalg.cuh:545-672) so xu / x_s / ref / lambda / gamma are byte-compatible
// with its buffers.  S / P^-1, where they are materialised, are BLOCK-major: [b][k][left | main | right][row][col] -- a block is nx^2
// contiguous floats, so the lanes that own the rows of one block read and write one contiguous run (the reference's row-major block rows
// [row][left | main | right] put a block's rows 3 nx floats apart: schur2 moved 4x its bytes; gato_debug_read converts for the tests).
// KKT blocks are stored COMPACT:
//   D    [b][k][3 nq^2]   = [dqdd/dq | dqdd/dqd | M^-1] col-major nq x 3nq   (A_k, B_k are functions of D and dt: A_elem/B_elem)
//   Qq   [b][k][nq^2], Qd [b][k][nq]   Q_k = blkdiag(Qq, diag(Qd))   (the cost Hessian has no other non-zeros, indy7_plant.cuh:375-408)
//   Rd   [b][k][nu]                     R_k = diag(Rd)
//   q [b][k][nx], r [b][k][nu], c [b][k][nx]; Qqi/Qdi/Rdi hold the inverses (separate buffers: no RAW hazard)
// `float` is the solver's REAL type: real.hpp turns it into double for the float64 build (-DGATO_DOUBLE, the reference's USE_DOUBLES).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <type_traits>

#include "real.hpp"
#include "rbd.hpp"

namespace gato {

typedef float f32x2 __attribute__((ext_vector_type(2)));   // a pair of reals: one v_pk_fma_f32 per fma in the fp32 build

// The LDS vectors of the register-resident PCG kernels keep their blocks at a stride of nx rounded up to a multiple of 4 (16-byte aligned
// windows for nx = 14); GATO_PCG_VSTRIDE=0 at build time keeps the dense layout (the A/B switch of the measurement in DESIGN.md section 6)
#ifndef GATO_PCG_VSTRIDE
#define GATO_PCG_VSTRIDE 1
#endif
#ifndef GATO_PCG_VSKEW
#define GATO_PCG_VSKEW 4   // floats added to a block stride that is a multiple of 16 (64 bytes): see pcg_vec_stride
#endif
#ifndef GATO_STEP_DZ_ROWS
#define GATO_STEP_DZ_ROWS 1
#endif
#ifndef GATO_SCHUR1_GJ_LDS
#define GATO_SCHUR1_GJ_LDS 1
#endif
#ifndef GATO_SCHUR1_STAGE
#define GATO_SCHUR1_STAGE 1
#endif
#ifndef GATO_PCGS_VSTRIDE
#define GATO_PCGS_VSTRIDE 0   // the same layout in pcgs_kernel (symmetric half storage): MEASURED SLOWER, off -- the two floats of padding a window's last
                              // ds_read_b128 brings along cost that kernel 20 more bytes of scratch inside its loop (256 registers, 36 -> 56 bytes): C3 577 vs 397 us per launch
#endif
// Block stride (floats) of the LDS vectors of the register-resident PCG kernels.  nx % 4 == 0: dense.  Otherwise nx rounded up to whole 16-byte
// chunks -- and, where that is a multiple of 16 floats (nx = 14 -> 16), GATO_PCG_VSKEW floats more: the lanes of a wavefront that own
// consecutive block rows read the same chunk of consecutive blocks in ONE ds_read_b128, and at a 64-byte stride blocks k and k + 4 (k + 2 on 32
// banks) start on the same bank: 41 % of C5's LDS cycles were bank conflicts at stride 16 (SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE, 3.5 % at
// the dense 14); at 20 the nine block rows of a wavefront start on nine different 4-bank slots.
constexpr int pcg_vec_stride(int nx)
{
    const int lw = (nx + 3) & ~3;
    return nx % 4 == 0 ? nx : (lw % 16 == 0 ? lw + GATO_PCG_VSKEW : lw);
}
constexpr int NUM_ALPHAS = 8;        // settings.h:16
constexpr float RHO_FACTOR = 1.2f;   // settings.h:20
constexpr float RHO_MIN = 1e-8f;     // settings.h:21
constexpr float RHO_MAX = 10.0f;     // settings.h:22
constexpr float RHO_INIT = 1e-3f;    // settings.h:18

// Cost weights of ONE trajectory.  The reference has one scalar set per solver (bsqp.cuh:344-350); here they live per trajectory
// (Buffers::costw, [B][8] floats, filled from the solver's scalars unless gato_set_cost_weights_batch overrides them) so that a
// hyper-parameter sweep is one batch.
struct Costs {
    float q_cost, qd_cost, u_cost, N_cost, q_lim_cost, vel_lim_cost, ctrl_lim_cost;
};

// device-side SQP loop control (replaces the host loop of bsqp.cuh:137-176)
struct Ctrl {
    uint32_t done;        // set once the solve_ratio exit fired: every later kernel of this solve is a no-op
    uint32_t iters_done;  // outer iterations executed (kkt..dz ran)
    uint32_t ls_done;     // line searches executed
    uint32_t pad;
};

struct Buffers {
    // problem (borrowed for the solve)
    float* xu; const float* x_s; const float* ref;
    // persistent per-trajectory state
    float *lambda, *rho, *drho, *mu, *pcg_tol, *f_ext;
    const float* costw;  // [B][8]: q, qd, u, N, q_lim, vel_lim, ctrl_lim cost weights, one padding float
    // KKT (compact) + Schur
    float *D, *Qq, *Qd, *Rd, *q, *r, *c, *Qqi, *Qdi, *Rdi, *S, *Pinv, *gamma, *dz;
    float *merit, *merit_cur, *step;
    int32_t* converged; uint32_t* pcg_iters;
    const int32_t* order;  // a permutation of the trajectories: workgroup i of pcgc_kernel solves order[i] (identity unless the launch runs in
                           // several rounds, then hardest-first by the previous iteration's PCG counts: order_by_pcg_iters)
    // per-iteration stats [max_iters][B]
    int32_t* st_pcg_iters; float *st_min_merit, *st_step;
    Ctrl* ctrl;
    uint32_t* num_solved;  // [max_iters]: trajectories counted as solved after outer iteration i (bsqp.cuh:142-163) -- over the WHOLE batch: what
                           // the exit rule reads.  On a sharded batch (gato_comm_init) the sum over the ranks of ...
    uint32_t* num_solved_w;  // ... this rank's own count, which the PCG kernels add to (the same array on a single GPU)
};

#define GATO_DEV_EARLY __device__ __forceinline__
GATO_DEV_EARLY Costs load_costs(const Buffers& bf, int b)
{
    const real4 lo = reinterpret_cast<const real4*>(bf.costw)[2 * b], hi = reinterpret_cast<const real4*>(bf.costw)[2 * b + 1];
    return Costs{lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z};
}

// ---- integrator pieces (integrator.cuh:34-37, 143-184), INTEGRATOR_TYPE 2 ----------------------------------------------
GATO_DEV float half_dt_sq(float dt) { return (float)(0.5 * (double)dt * (double)dt); }

template<int NQ> GATO_DEV float A_elem(const float* D, int r, int c, float dt, float h2)
{
    // D[c*NQ + r'] = d qdd_r' / d x_c, c < 2 NQ
    const float d = D[c * NQ + (r % NQ)];
    float val = (r == c) ? 1.0f : 0.0f;
    if (r < NQ) {
        if (c >= NQ && r == c - NQ) val += dt;
        val += h2 * d;
    } else {
        val += dt * d;
    }
    return val;
}
template<int NQ> GATO_DEV float B_elem(const float* D, int r, int c, float dt, float h2)
{
    const float d = D[2 * NQ * NQ + c * NQ + (r % NQ)];
    return (r < NQ) ? h2 * d : dt * d;
}

// ---- barrier functions (indy7_plant.cuh:130-148, iiwa14_plant.cuh:104-155) -------------------------------------------------
GATO_DEV float joint_barrier(float q, float lo, float hi)
{
    float dmin = q - lo, dmax = hi - q;
    dmin = (dmin <= 1e-10) ? (float)1e-10 : dmin;
    dmax = (dmax <= 1e-10) ? (float)1e-10 : dmax;
    return -logf(dmin) - logf(dmax);
}
template<int MODE> GATO_DEV float joint_barrier_grad(float q, float lo, float hi)
{
    float dmin = q - lo, dmax = hi - q;
    if constexpr (MODE == 0) {
        dmin = (dmin <= 1e-6) ? (float)1e-6 : dmin;
        dmax = (dmax <= 1e-6) ? (float)1e-6 : dmax;
        return (-1 / dmin) + (1 / dmax);
    } else {
        const float eps = 1e-6f;
        if (dmin >= 0.f) { if (dmin < eps) dmin = eps; } else { if (dmin > -eps) dmin = -eps; }
        if (dmax >= 0.f) { if (dmax < eps) dmax = eps; } else { if (dmax > -eps) dmax = -eps; }
        return (-1.0f / dmin) + (1.0f / dmax);
    }
}
GATO_DEV float joint_barrier_hess(float q, float lo, float hi)
{
    float dmin = q - lo, dmax = hi - q;
    const float eps = 1e-6f;
    float amin = dmin >= 0.f ? dmin : -dmin, amax = dmax >= 0.f ? dmax : -dmax;
    if (amin < eps) amin = eps;
    if (amax < eps) amax = eps;
    return 1.0f / (amin * amin) + 1.0f / (amax * amax);
}

// GLOBAL memory only: 16-byte accesses at 4-byte alignment (legal for global_load/store_dwordx4 on gfx950; LDS needs natural
// alignment, so the ALIGN-aware helpers below stay in charge there).  nq = 7 gives every per-knot block an odd float count, and
// with the aligned helpers all of iiwa14's block traffic was single-dword instructions.
struct __attribute__((packed, aligned(4))) F4U { float x, y, z, w; };
template<int CNT> GATO_DEV void gload_vec(float* dst, const float* __restrict__ src)
{
#pragma unroll
    for (int i = 0; i < CNT / 4; i++) {
        const F4U v = reinterpret_cast<const F4U*>(src)[i];
        dst[4 * i] = v.x; dst[4 * i + 1] = v.y; dst[4 * i + 2] = v.z; dst[4 * i + 3] = v.w;
    }
#pragma unroll
    for (int i = CNT / 4 * 4; i < CNT; i++) dst[i] = src[i];
}
template<int CNT> GATO_DEV void gstore_vec(float* __restrict__ dst, const float* src)
{
#pragma unroll
    for (int i = 0; i < CNT / 4; i++) reinterpret_cast<F4U*>(dst)[i] = F4U{src[4 * i], src[4 * i + 1], src[4 * i + 2], src[4 * i + 3]};
#pragma unroll
    for (int i = CNT / 4 * 4; i < CNT; i++) dst[i] = src[i];
}

// vectorised private<->global copies.  ALIGN = a number of floats that divides every offset the pointer can take (and the
// allocation itself is 256-byte aligned), so the widest access that is BOTH naturally aligned and divides CNT is used.
template<int CNT, int ALIGN> constexpr int vec_width()
{
    return (CNT % 4 == 0 && ALIGN % 4 == 0) ? 4 : ((CNT % 2 == 0 && ALIGN % 2 == 0) ? 2 : 1);
}
template<int CNT, int ALIGN> GATO_DEV void store_vec(float* __restrict__ dst, const float* src)
{
    constexpr int W = vec_width<CNT, ALIGN>();
    if constexpr (W == 4) {
#pragma unroll
        for (int i = 0; i < CNT / 4; i++) reinterpret_cast<real4*>(dst)[i] = make_real4(src[4 * i], src[4 * i + 1], src[4 * i + 2], src[4 * i + 3]);
    } else if constexpr (W == 2) {
#pragma unroll
        for (int i = 0; i < CNT / 2; i++) reinterpret_cast<real2*>(dst)[i] = make_real2(src[2 * i], src[2 * i + 1]);
    } else {
#pragma unroll
        for (int i = 0; i < CNT; i++) dst[i] = src[i];
    }
}
template<int CNT, int ALIGN> GATO_DEV void load_vec(float* dst, const float* __restrict__ src)
{
    constexpr int W = vec_width<CNT, ALIGN>();
    if constexpr (W == 4) {
#pragma unroll
        for (int i = 0; i < CNT / 4; i++) {
            const real4 v = reinterpret_cast<const real4*>(src)[i];
            dst[4 * i] = v.x; dst[4 * i + 1] = v.y; dst[4 * i + 2] = v.z; dst[4 * i + 3] = v.w;
        }
    } else if constexpr (W == 2) {
#pragma unroll
        for (int i = 0; i < CNT / 2; i++) {
            const real2 v = reinterpret_cast<const real2*>(src)[i];
            dst[2 * i] = v.x; dst[2 * i + 1] = v.y;
        }
    } else {
#pragma unroll
        for (int i = 0; i < CNT; i++) dst[i] = src[i];
    }
}

// sum over aligned groups of `seg` consecutive threads (seg = power of two, <= blockDim); every thread gets the group's sum
GATO_DEV float seg_sum(float v, int seg, float* lds_part)
{
    const int w = seg < 64 ? seg : 64;
    for (int off = w >> 1; off >= 1; off >>= 1) v += __shfl_xor(v, off, 64);
    if (seg > 64) {
        const int wave = threadIdx.x >> 6, per = seg >> 6;
        if ((threadIdx.x & 63) == 0) lds_part[wave] = v;
        __syncthreads();
        const int base = (wave / per) * per;
        float s = 0.f;
        for (int i = 0; i < per; i++) s += lds_part[base + i];
        v = s;
    }
    return v;
}

// =========================================================================================================================
// merit: M_b(alpha) = sum_k cost_k + mu_b (sum_k |x_{k+1} - f(x_k,u_k)|_1 + |x_0 - x_s|_1) at xu + alpha dz   (merit.cuh:17-92)
// one lane per (b, alpha, k); grid-stride free: thread g -> k = g % N, a = (g / N) % NA, b = g / (N NA)
// =========================================================================================================================
// one lane's term of the merit function: knot k of trajectory b at xu + alpha dz (dzb: the trajectory's step, global or LDS)
template<class M>
GATO_DEV float merit_term(const Buffers& bf, const Costs& cw, int N, int b, int k, float alpha, int use_dz, const float* dzb, float dt)
{
    constexpr int NQ = M::NQ, NX = 2 * NQ, NU = NQ, KS = NX + NU;
    const int traj = KS * N - NU;
    const float* xu = bf.xu + (size_t)b * traj + (size_t)k * KS;
    const float* dz = dzb + (size_t)k * KS;
    const bool last = (k == N - 1);
    // [x_k, u_k, x_{k+1}] + alpha [dz...]; the last knot has x only.  Loads are UNCONDITIONAL (a branch around a load makes hipcc wait for
    // each one separately): last-knot lanes read the previous knot's tail instead, then zero it.  Order of work = order of register
    // need: q first (kinematics, then M^-1 with nothing else live), then qd and u (costs, bias forces), x_{k+1} last.
    constexpr int AL = (KS % 2 == 0) ? 2 : 1;  // knot offsets are multiples of KS floats, TRAJ = KS N - NU has the parity of KS for NU | KS
    const float* xt = last ? xu - KS : xu;
    const float* dzt = last ? dz - KS : dz;
    float sq[NQ];
    {
        load_vec<NQ, AL>(sq, xu);
        if (use_dz) {
            float t[NQ];
            load_vec<NQ, AL>(t, dz);
#pragma unroll
            for (int i = 0; i < NQ; i++) sq[i] += alpha * t[i];
        }
    }
    RBD<M> d;
    d.set_q(sq);
    float cost = 0.f;
    {
        // ---- end-effector and joint-limit terms of the cost (plant::trackingcost, indy7_plant.cuh:266-318)
        const float* ref = bf.ref + (size_t)b * 6 * N + 6 * k;
        float e[3];
        d.ee_pos(e);
        const float w = last ? cw.N_cost : cw.q_cost;
#pragma unroll
        for (int i = 0; i < NQ; i++) cost += cw.q_lim_cost * joint_barrier(sq[i], M::Q_LIM[i][0], M::Q_LIM[i][1]);
#pragma unroll
        for (int i = 0; i < 3; i++) {
            const float err = e[i] - ref[i];
            cost += (float)(0.5 * (double)w * (double)err * (double)err);
        }
    }
    typename RBD<M>::MinvT Mi;
    if (opaque_true()) d.minv(Mi);  // its own basic block, entered with q, sin/cos and one accumulator live
    float s[KS];  // only [NQ, KS) is used: qd_k, u_k at the trial point
    {
        load_vec<NQ, 1>(s + NQ, xu + NQ);
        load_vec<NU, AL>(s + NX, xt + NX);
        if (use_dz) {
            float t[KS];
            load_vec<NQ, 1>(t + NQ, dz + NQ);
            load_vec<NU, AL>(t + NX, dzt + NX);
#pragma unroll
            for (int i = NQ; i < KS; i++) s[i] += alpha * t[i];
        }
#pragma unroll
        for (int i = NX; i < KS; i++) s[i] = last ? 0.f : s[i];
    }
#pragma unroll
    for (int i = 0; i < NQ; i++) {
        const float err = s[NQ + i];
        float t = 0.5f * cw.qd_cost * err * err;
        if (cw.vel_lim_cost != 0.f) t += cw.vel_lim_cost * joint_barrier(err, M::V_LIM[i][0], M::V_LIM[i][1]);
        cost += t;
    }
    if (!last) {
#pragma unroll
        for (int i = 0; i < NU; i++) {
            const float err = s[NX + i];
            float t = 0.5f * cw.u_cost * err * err;
            if (cw.ctrl_lim_cost != 0.f) t += cw.ctrl_lim_cost * joint_barrier(err, M::U_LIM[i][0], M::U_LIM[i][1]);
            cost += t;
        }
    }
    // ---- constraint violation
    float con = 0.f;
    if (!last) {
        float fe[6], qdd[NQ], f[NQ][6];
#pragma unroll
        for (int i = 0; i < 6; i++) fe[i] = bf.f_ext[6 * b + i];
        d.rnea_lean(s + NQ, fe, f);
        RBD<M>::fd_finish(Mi, s + NX, f, qdd);
        float xn[NX];
        load_vec<NX, AL>(xn, xt + KS);
        if (use_dz) {
            float t[NX];
            load_vec<NX, AL>(t, dzt + KS);
#pragma unroll
            for (int i = 0; i < NX; i++) xn[i] += alpha * t[i];
        }
#pragma unroll
        for (int i = 0; i < NQ; i++) {
            const float qdn = s[NQ + i] + dt * qdd[i];
            const float qn = (float)((double)(sq[i] + dt * s[NQ + i]) + 0.5 * (double)qdd[i] * (double)dt * (double)dt);
            con += fabsf(xn[i] - qn);
            con += fabsf(xn[NQ + i] - qdn);
        }
    } else {
        const float* x0 = bf.xu + (size_t)b * traj;
        const float* dz0 = dzb;
#pragma unroll
        for (int i = 0; i < NX; i++) {
            float v = x0[i];
            if (use_dz) v += alpha * dz0[i];
            con += fabsf(v - bf.x_s[(size_t)b * NX + i]);
        }
    }
    return cost + bf.mu[b] * con;
}

template<class M, int NA>
__global__ __launch_bounds__(256) void merit_kernel(Buffers bf, int N, int B, float dt, int use_dz, int sqp_iter, float thresh,
                                                    float* __restrict__ out, float* __restrict__ out2, real4* __restrict__ zero4, uint32_t zero_n4)
{
    constexpr int NQ = M::NQ, NX = 2 * NQ, NU = NQ, KS = NX + NU;
    __shared__ float part[4];
    if (sqp_iter >= 0) {
        if (bf.ctrl->done) return;
        if ((float)bf.num_solved[sqp_iter] >= thresh) return;  // loop breaks before the line search (bsqp.cuh:165)
    }
    const int g = blockIdx.x * blockDim.x + threadIdx.x;
    // the first launch of a solve also clears what bsqp.cuh:112-114 memsets (dz, PCG counts, convergence flags) and the device-side
    // loop control: nothing reads them before the next launch, and a fill launch of its own costs 2-4 us per solve
    for (uint32_t i = g; i < zero_n4; i += gridDim.x * blockDim.x) zero4[i] = make_real4(0.f, 0.f, 0.f, 0.f);
    const int k = g % N, ai = (g / N) % NA;
    int b = g / (N * NA);
    const bool live = b < B;
    if (!live) b = B - 1;
    const float alpha = (float)(1.0 / (double)(1 << ai));
    float m = merit_term<M>(bf, load_costs(bf, b), N, b, k, alpha, use_dz, bf.dz + (size_t)b * (KS * N - NU), dt);
    m = seg_sum(m, N, part);
    if (live && k == 0) {
        out[b * NA + ai] = m;
        if (NA == 1 && out2) out2[b] = m;  // merit_initial0 (bsqp.cuh:116-118) without a device-to-device copy
    }
}

// =========================================================================================================================
// KKT assembly (setup_kkt.cuh:15-108): one lane per (b,k).  k <= N-2: D_k, c_{k+1}, cost blocks of knot k; k == N-2 also the
// terminal blocks (at x_{N-2} against ref_{N-1}, with q_cost -- SURVEY A.1/A.2); k == N-1: c_0 = x_0 - x_s.
// =========================================================================================================================
template<class M>
GATO_DEV void cost_blocks(const RBD<M>& d, const Costs& cw, const float* x, const float* u, const float* ref, float* Qq, float* Qd, float* qv,
                          float* Rd, float* rv)
{
    constexpr int NQ = M::NQ, MODE = M::BARRIER_MODE;
    float e[3], Jc[NQ][3], g[NQ], bq[NQ];
    d.ee_jac(e, Jc);
    const float w = cw.q_cost;
    const float e0 = e[0] - ref[0], e1 = e[1] - ref[1], e2 = e[2] - ref[2];
#pragma unroll
    for (int i = 0; i < NQ; i++) {
        g[i] = (Jc[i][0] * e0 + Jc[i][1] * e1 + Jc[i][2] * e2);
        bq[i] = joint_barrier_grad<MODE>(x[i], M::Q_LIM[i][0], M::Q_LIM[i][1]);
        float t = g[i] * w;
        t += cw.q_lim_cost * bq[i];
        qv[i] = t;
        float t2 = cw.qd_cost * x[NQ + i];
        float dd = cw.qd_cost;
        if (cw.vel_lim_cost != 0.f) {
            const float bv = joint_barrier_grad<MODE>(x[NQ + i], M::V_LIM[i][0], M::V_LIM[i][1]);
            t2 += cw.vel_lim_cost * bv;
            if constexpr (MODE == 0) dd += cw.vel_lim_cost * bv * bv;
            else dd += cw.vel_lim_cost * joint_barrier_hess(x[NQ + i], M::V_LIM[i][0], M::V_LIM[i][1]);
        }
        qv[NQ + i] = t2;
        Qd[i] = dd;
    }
#pragma unroll
    for (int i = 0; i < NQ; i++)
#pragma unroll
        for (int j = 0; j < NQ; j++) {
            float val = (g[i] * g[j]) * w;
            if constexpr (MODE == 0) {
                val += cw.q_lim_cost * bq[i] * bq[j];
            } else {
                if (i == j) val += cw.q_lim_cost * joint_barrier_hess(x[i], M::Q_LIM[i][0], M::Q_LIM[i][1]);
            }
            Qq[i * NQ + j] = val;
        }
    if (Rd) {
#pragma unroll
        for (int i = 0; i < NQ; i++) {
            float t = cw.u_cost * u[i];
            float dd = cw.u_cost;
            if (cw.ctrl_lim_cost != 0.f) {
                const float bu = joint_barrier_grad<MODE>(u[i], M::U_LIM[i][0], M::U_LIM[i][1]);
                t += cw.ctrl_lim_cost * bu;
                if constexpr (MODE == 0) dd += cw.ctrl_lim_cost * bu * bu;
                else dd += cw.ctrl_lim_cost * joint_barrier_hess(u[i], M::U_LIM[i][0], M::U_LIM[i][1]);
            }
            rv[i] = t;
            Rd[i] = dd;
        }
    }
}


// Gauss-Jordan inverse without pivoting, the arithmetic of block::invertMatrix (linalg.cuh:364-519) on [V | I]:
//   THREE = true : a / p * row   (2-/3-matrix overloads, used for Q_k, Q_{k+1}, R_k)
//   THREE = false: a * (1/p) * row (1-matrix overload, used for theta_k and Q_0)
// IN PLACE: at pivot p column p of V has become e_p in the augmented scheme and column p of the identity side is about to be
// produced, so that column is stored in V's slot.  Every element sees exactly the operations (and operands) of the reference's
// (n+1)-column sliding window -- columns outside the window are untouched there because their pivot-row entry is an exact 0 --
// so the result is bit-identical to the augmented form at half the registers.  Mat is col-major n x n and returns the inverse.
template<int n, bool THREE> GATO_DEV void gj_inverse(float* Mat)
{
#pragma unroll
    for (int p = 0; p < n; p++) {
        float colv[n];
#pragma unroll
        for (int i = 0; i < n; i++) colv[i] = Mat[p * n + i];
        const float pv = colv[p];
        const float pvInv = 1.0f / pv;
#pragma unroll
        for (int c = 0; c < n; c++) {
            const float rowv = (c == p) ? 1.0f : Mat[c * n + p];
#pragma unroll
            for (int r = 0; r < n; r++) {
                const float x = (c == p) ? ((r == p) ? 1.0f : 0.0f) : Mat[c * n + r];
                float y;
                if constexpr (THREE) {
                    y = (r == p) ? x / pv : x - colv[r] / pv * rowv;
                } else {
                    y = (r == p) ? x * pvInv : x - colv[r] * pvInv * rowv;
                }
                Mat[c * n + r] = y;
            }
        }
    }
}

// One derivative-column task of the assembly kernel (see kkt_kernel): forward dynamics, then columns JA and JB of
// [d qdd / d q | d qdd / d qd | M^-1]; the task with column 0 also stores the defect c_{k+1}.  The common prefix (M^-1, two RNEA
// passes, 3.7 k instructions) is recomputed per task.
// NOT inlined on purpose: with every task inlined into one kernel the code object grows past the +-128 KB reach of s_cbranch
// (325 KB for iiwa14 with 7 tasks) and the compiler's long-branch relaxation produced wrong results on gfx950 / ROCm 7.2 (columns of
// D corrupted when f_ext != 0; caught by tests/test_gpu_parity.py).  As real functions every task body stays below 64 KB and is
// reached through s_swappc.  not_tail_called: a call marked `tail` disables the no-callee-saved-registers optimisation and the task
// would open with ~110 scratch stores of caller registers nobody needs.
template<class M, int JA, int JB, int JC = -1>
__device__ __noinline__ __attribute__((not_tail_called)) void kkt_columns(float* __restrict__ D, float* __restrict__ c_out, const float* __restrict__ xu,
                                         const float* __restrict__ f_ext, float dt)
{
    constexpr int NQ = M::NQ, NX = 2 * NQ, NU = NQ, KS = NX + NU;
    // the knot's [x_k u_k x_{k+1}] and the wrench are fetched HERE, once, into registers: handed over through private memory they
    // were re-read (flat_load + s_waitcnt 0) a dozen times along the chain
    float x[KS + NX], fe[6];
#pragma unroll
    for (int i = 0; i < KS + NX; i++) x[i] = xu[i];
#pragma unroll
    for (int i = 0; i < 6; i++) fe[i] = f_ext[i];
    RBD<M> d;
    d.set_q(x);
    d.template fd_grad_columns<JA, JB, JC>(
        x + NQ, x + NX, fe,
        [&](int J, const float* cq, const float* cd, const float* cm) {
            store_vec<NQ, NQ>(D + J * NQ, cq);
            store_vec<NQ, NQ>(D + NQ * NQ + J * NQ, cd);
            store_vec<NQ, NQ>(D + 2 * NQ * NQ + J * NQ, cm);
        },
        [&](const float* qdd) {
            if constexpr (JA == 0) {
                float c[NX];
#pragma unroll
                for (int i = 0; i < NQ; i++) {
                    const float qdn = x[NQ + i] + dt * qdd[i];
                    const float qn = (float)((double)(x[i] + dt * x[NQ + i]) + 0.5 * (double)qdd[i] * (double)dt * (double)dt);
                    c[i] = x[KS + i] - qn;
                    c[NQ + i] = x[KS + NQ + i] - qdn;
                }
                store_vec<NX, NX>(c_out, c);
            }
        });
}

// Cost blocks of one knot and their inverses (Q + rho I_q)^-1, R^-1 -- the arithmetic of the 3-matrix Gauss-Jordan of
// schur_linsys.cuh:96.  `terminal`: the lane produces the blocks of knot N-1 from x_{N-2} against ref_{N-1} and
// has no R block.  One uniform instruction stream for both kinds of lane; only the R stores are predicated.
template<class M>
GATO_DEV void schur_row0_regs(float* Qq, const float* Qd, const float* q0, const float* c0, float rho, float* S, float* P, float* gam);

// row0 (only ever true in the k = 0 lane): also form the Q_0 rows of S, P^-1 and gamma_0 here, from the blocks in registers -- the
// fused Schur + PCG kernel has no lane to spare for them
template<class M>
GATO_DEV void kkt_costs(const Buffers& bf, const Costs& cw, const float* x, const float* ref, size_t bk, float rho, bool terminal, bool row0,
                        const float* x_s, float* S0, float* P0, float* gam0)
{
    constexpr int NQ = M::NQ, NX = 2 * NQ, NU = NQ;
    RBD<M> d;
    d.set_q(x);
    float Qq[NQ * NQ], Qd[NQ], qv[NX], Rd[NU], rv[NU];
    cost_blocks<M>(d, cw, x, x + NX, ref, Qq, Qd, qv, Rd, rv);
    if (row0) {
        float Q0[NQ * NQ], c0[NX];
#pragma unroll
        for (int i = 0; i < NQ * NQ; i++) Q0[i] = Qq[i];
#pragma unroll
        for (int i = 0; i < NX; i++) c0[i] = x[i] - x_s[i];
        schur_row0_regs<M>(Q0, Qd, qv, c0, rho, S0, P0, gam0);
    }
    store_vec<NQ * NQ, NQ * NQ>(bf.Qq + bk * NQ * NQ, Qq);
    store_vec<NQ, NQ>(bf.Qd + bk * NQ, Qd);
    store_vec<NX, NX>(bf.q + bk * NX, qv);
    if (!terminal) {
        store_vec<NU, NU>(bf.Rd + bk * NU, Rd);
        store_vec<NU, NU>(bf.r + bk * NU, rv);
    }
#pragma unroll
    for (int i = 0; i < NQ; i++) {
        Qq[i * NQ + i] += rho;        // first nx/2 diagonal entries only (linalg.cuh:84-96)
        Qd[i] = 1.0f / Qd[i];
        Rd[i] = 1.0f / Rd[i];
    }
    gj_inverse<NQ, true>(Qq);
    store_vec<NQ * NQ, NQ * NQ>(bf.Qqi + bk * NQ * NQ, Qq);
    store_vec<NQ, NQ>(bf.Qdi + bk * NQ, Qd);
    if (!terminal) store_vec<NU, NU>(bf.Rdi + bk * NU, Rd);
}

// Tasks: one wavefront of 64 (b,k) lanes each, the same split for every batch size (results do not depend on B).  Every column task pays
// the common prefix (M^-1, two RNEA passes: ~3.7 k instructions) before its derivative columns (~1.4 k each, falling with J):
//   NQ even (indy7):  task g < NQ/2: columns g and NQ-1-g (a long and a short one); task NQ/2: the cost blocks            -- 4 wavefronts
//   NQ = 7 (iiwa14):  columns {0,5,6}, {1,3}, {2,4} (6.8 k / 7.1 k / 6.5 k instructions); task 3: the cost blocks            -- 4 wavefronts
//     (was {0,6} {1,5} {2,4} {3} + costs = 5 wavefronts: at two wavefronts per SIMD a CU holds 8, so one 5-wavefront workgroup at a time and
//     the 512 workgroups of C3 / C5 in two rounds; with 4, two fit and all are resident at once)
//   task 0 also stores the defect c_{k+1}; the cost task's lane k = N-1 produces the terminal blocks from x_{N-2} and c_0 = x_0 - x_s.
// __launch_bounds__(64, 2): two wavefronts per SIMD.  A lone wavefront issues one dependent VALU instruction per ~10 cycles; the
// second one fills the gaps, which pays for the few spills the 256-register budget costs.
template<class M> constexpr int kkt_tasks() { return (M::NQ == 7 ? 3 : (M::NQ + 1) / 2) + 1; }
template<class M, int g> GATO_DEV void kkt_dispatch(int task, const Buffers& bf, const float* xu, const float* fe, float* Dout, size_t bk, float dt)
{
    constexpr int NQ = M::NQ, NX = 2 * NQ;
    if constexpr (g < kkt_tasks<M>() - 1) {
        if (task == g) {
            if constexpr (NQ == 7) {
                if constexpr (g == 0) kkt_columns<M, 0, 5, 6>(Dout, bf.c + (bk + 1) * NX, xu, fe, dt);
                else if constexpr (g == 1) kkt_columns<M, 1, 3>(Dout, bf.c + (bk + 1) * NX, xu, fe, dt);
                else kkt_columns<M, 2, 4>(Dout, bf.c + (bk + 1) * NX, xu, fe, dt);
            } else {
                kkt_columns<M, g, NQ - 1 - g>(Dout, bf.c + (bk + 1) * NX, xu, fe, dt);
            }
        } else {
            kkt_dispatch<M, g + 1>(task, bf, xu, fe, Dout, bk, dt);
        }
    }
}

// One workgroup = the NT tasks (wavefronts) of 64 consecutive knots.  The column tasks leave their pieces of D in LDS ([64][3 nq^2]);
// after a barrier the whole workgroup copies the 64 blocks -- contiguous in global memory -- out with 16-byte stores of consecutive
// lanes.  Written directly, every lane's 24-byte column pieces were separate 8-byte requests to different cache lines (54 per knot);
// the request rate of those stores, not the arithmetic, bounded this kernel.
// the assembly of the 64 consecutive knots [64 wg, 64 wg + 64) by the NT task-wavefronts of a workgroup (the body of kkt_kernel)
template<class M>
GATO_DEV void kkt_body(const Buffers& bf, int N, int B, float dt, int row0, int wg, float* ldsD)
{
    constexpr int NQ = M::NQ, NX = 2 * NQ, NU = NQ, KS = NX + NU, NT = kkt_tasks<M>(), ND = 3 * NQ * NQ;
    const int lane = threadIdx.x & 63;
    const int task = threadIdx.x >> 6;  // wave-uniform
    const long total = (long)B * N;
    const long g = (long)wg * 64 + lane;
    const bool valid = g < total;       // lanes past the batch shadow knot (0,0) and store nothing
    const int k = valid ? (int)(g % N) : 0, b = valid ? (int)(g / N) : 0;
    const bool last = (k == N - 1);
    const int traj = KS * N - NU;
    const int kx = last ? N - 2 : k;  // the terminal lane reads knot N-2
    const float* xu = bf.xu + (size_t)b * traj + (size_t)kx * KS;
    const size_t bk = (size_t)b * N + k;
    float* Dl = ldsD + lane * ND;
    if (task != NT - 1) {
        if (valid && !last) {
            kkt_dispatch<M, 0>(task, bf, xu, bf.f_ext + 6 * b, Dl, bk, dt);
        } else if (task == 0) {
            for (int i = 0; i < ND; i++) Dl[i] = 0.f;  // the last knot has no dynamics block: its slot stays zero
        }
    } else if (valid) {
        float x[KS + NX];
#pragma unroll
        for (int i = 0; i < KS + NX; i++) x[i] = xu[i];
        constexpr int BROW = 3 * NX * NX;
        kkt_costs<M>(bf, load_costs(bf, b), x, bf.ref + (size_t)b * 6 * N + 6 * k, bk, bf.rho[b], last, row0 && k == 0, bf.x_s + (size_t)b * NX,
                     bf.S + (size_t)b * N * BROW, bf.Pinv + (size_t)b * N * BROW, bf.gamma + (size_t)b * (N + 2) * NX);
        if (last) {
            float c0[NX];
            const float* x0 = bf.xu + (size_t)b * traj;
#pragma unroll
            for (int i = 0; i < NX; i++) c0[i] = x0[i] - bf.x_s[(size_t)b * NX + i];
            store_vec<NX, NX>(bf.c + (size_t)b * N * NX, c0);
        }
    }
    __syncthreads();
    const long first = (long)wg * 64;
    const int cnt = (int)((total - first) < 64 ? (total - first) : 64);
    const int nfl = cnt * ND, n4 = nfl / 4;
    float* gD = bf.D + (size_t)first * ND;  // 64 ND floats per workgroup: 16-byte aligned
    for (int i = threadIdx.x; i < n4; i += blockDim.x) reinterpret_cast<real4*>(gD)[i] = reinterpret_cast<const real4*>(ldsD)[i];
    for (int i = 4 * n4 + threadIdx.x; i < nfl; i += blockDim.x) gD[i] = ldsD[i];
}

template<class M>
__global__ __launch_bounds__(64 * kkt_tasks<M>(), 2) void kkt_kernel(Buffers bf, int N, int B, float dt, int sqp_iter, float thresh, int row0,
                                                                     real4* __restrict__ zero4, uint32_t zero_n4)
{
    extern __shared__ __attribute__((aligned(16))) float ldsD[];
    if (zero_n4) {
        // the FIRST launch of a solve whose initial merit is formed by the first step launch (solver.hip:solve_impl): it clears what
        // bsqp.cuh:112-114 memsets (dz, PCG counts, convergence flags) and the device-side loop control -- nothing in this kernel
        // reads them at iteration 0, the next launch does -- so a solve needs neither a fill nor a merit launch ahead of its loop
        for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < zero_n4; i += gridDim.x * blockDim.x) zero4[i] = make_real4(0.f, 0.f, 0.f, 0.f);
    } else {
        if (bf.ctrl->done) return;
        if (sqp_iter > 0 && (float)bf.num_solved[sqp_iter - 1] >= thresh) {
            // the previous iteration ended the loop (bsqp.cuh:165): raise `done` for everything that follows; whether a workgroup sees
            // the flag or the count, it leaves
            if (blockIdx.x == 0 && threadIdx.x == 0) bf.ctrl->done = 1;
            return;
        }
    }
    kkt_body<M>(bf, N, B, dt, row0, (int)blockIdx.x, ldsD);
}

// =========================================================================================================================
// Schur complement formation (schur_linsys.cuh:14-211)
// =========================================================================================================================
// the Q_0 row (schur_linsys.cuh:166-210) of trajectory b, by one lane
template<class M>
GATO_DEV void schur_row0_regs(float* Qq, const float* Qd, const float* q0, const float* c0, float rho, float* S, float* P, float* gam)
{
    constexpr int NQ = M::NQ, NX = 2 * NQ, BLK = NX * NX;
    {
        // the Q_0 row (schur_linsys.cuh:166-210): P^-1 row 0 = -(Q_0 + rho I_q), S row 0 = -(Q_0 + rho I_q)^-1, gamma_0 = c_0 - Q_0^-1 q_0
        float Qi[NQ * NQ];
#pragma unroll
        for (int i = 0; i < NQ; i++) Qq[i * NQ + i] += rho;
#pragma unroll
        for (int y = 0; y < NX; y++) {
            float row[NX];
#pragma unroll
            for (int x = 0; x < NX; x++) {
                float v = 0.f;
                if (y < NQ && x < NQ) v = -Qq[x * NQ + y];
                else if (x == y) v = -Qd[y - NQ];
                row[x] = v;
            }
            store_vec<NX, NX>(P + BLK + y * NX, row);
        }
        gj_inverse<NQ, false>(Qq);
#pragma unroll
        for (int i = 0; i < NQ * NQ; i++) Qi[i] = Qq[i];
        float di[NQ];
#pragma unroll
        for (int i = 0; i < NQ; i++) di[i] = 1.0f * (1.0f / Qd[i]);
#pragma unroll
        for (int y = 0; y < NX; y++) {
            float row[NX];
#pragma unroll
            for (int x = 0; x < NX; x++) {
                float v = 0.f;
                if (y < NQ && x < NQ) v = -Qi[x * NQ + y];
                else if (x == y) v = -di[y - NQ];
                row[x] = v;
            }
            store_vec<NX, NX>(S + BLK + y * NX, row);
        }
        float g0[NX];
#pragma unroll
        for (int y = 0; y < NX; y++) {
            float s = 0.f;
            if (y < NQ) {
#pragma unroll
                for (int j = 0; j < NQ; j++) s += Qi[j * NQ + y] * q0[j];
            } else {
                s = di[y - NQ] * q0[y];
            }
            g0[y] = c0[y] + (-s);
        }
        store_vec<NX, NX>(gam + NX, g0);
    }
}

template<class M>
GATO_DEV void schur_row0(const Buffers& bf, int N, int b, float rho, float* S, float* P, float* gam)
{
    constexpr int NQ = M::NQ, NX = 2 * NQ;
    const size_t b0 = (size_t)b * N;
    float Qq[NQ * NQ], Qd[NQ], q0[NX], c0[NX];
    load_vec<NQ * NQ, NQ * NQ>(Qq, bf.Qq + b0 * NQ * NQ);
    load_vec<NQ, NQ>(Qd, bf.Qd + b0 * NQ);
    load_vec<NX, NX>(q0, bf.q + b0 * NX);
    load_vec<NX, NX>(c0, bf.c + b0 * NX);
    schur_row0_regs<M>(Qq, Qd, q0, c0, rho, S, P, gam);
}

// ---- cooperative form: LPP lanes per (b,k) ------------------------------------------------------------------------------------
// Lane l of a group owns the RW = nx / LPP rows y0 = l RW .. of phi_k, theta_k and, after the Gauss-Jordan sweep, of
// (theta_k + rho I_q)^-1: 4 x the wavefronts of the lane-per-knot form at a quarter of the chain each, ~250 registers, and theta
// never travels through memory between the two halves (schur_kernel + pinv_kernel wrote and re-read it).  All lanes of a group load
// the same D_k, Q^-1 (identical addresses: one request per group); their own rows are picked out of the register copy with
// v_cndmask, never by dynamic indexing.  The pivot row of each elimination step comes from its owner through DPP quad_perm.
// LPP = 4 needs nq = 2 RW (indy7: rows 3l..3l+2), LPP = 2 needs nq = RW (iiwa14): a lane's rows never straddle the q | qd halves.
template<int LPP, int O> GATO_DEV float group_bcast(float v)  // lane O of every group of LPP consecutive lanes
{
    constexpr int ctrl = (LPP == 4) ? (O * 0x55) : (O | (O << 2) | ((2 + O) << 4) | ((2 + O) << 6));
    return dpp_get<ctrl>(v);
}
template<int LPP, int RW, int NX, int P> GATO_DEV void gj_coop_step(float (*W)[NX], int l, float rho_unused)
{
    // pivot P: owner lane O = P / RW, its local row I = P % RW
    constexpr int O = P / RW, I = P % RW;
    float prow[NX];
#pragma unroll
    for (int c = 0; c < NX; c++) prow[c] = group_bcast<LPP, O>(W[I][c]);
    const float pvInv = 1.0f / prow[P];
    const bool owner = (l == O);
#pragma unroll
    for (int i = 0; i < RW; i++) {
        const float f = W[i][P] * pvInv;  // colv[r] * pvInv
        // columns in pairs: away from the pivot column the update x - f prow[c] is one packed FMA for two columns (the same fma each)
#pragma unroll
        for (int c0 = 0; c0 < NX; c0 += 2) {
            float xv[2], yv[2];
            if (NX % 2 == 0 && c0 != (P & ~1)) {
                const f32x2 x2 = {W[i][c0], W[i][c0 + 1]};
                const f32x2 y2 = __builtin_elementwise_fma(f32x2{-f, -f}, f32x2{prow[c0], prow[c0 + 1]}, x2);
                xv[0] = x2.x; xv[1] = x2.y;
                yv[0] = y2.x; yv[1] = y2.y;
            } else {
#pragma unroll
                for (int e = 0; e < 2; e++) {
                    const int c = c0 + e;
                    if (c >= NX) break;
                    if (c == P) {
                        yv[e] = 0.f - f;  // x = 0, rowv = 1
                        xv[e] = 1.0f;
                    } else {
                        xv[e] = W[i][c];
                        yv[e] = xv[e] - f * prow[c];
                    }
                }
            }
#pragma unroll
            for (int e = 0; e < 2; e++) {
                const int c = c0 + e;
                if (c >= NX) break;
                float y = yv[e];
                if (i == I) {
                    const float piv = xv[e] * pvInv;  // the pivot row itself (only in the owner lane)
                    y = owner ? piv : y;
                }
                W[i][c] = y;
            }
        }
    }
    if constexpr (P + 1 < NX) gj_coop_step<LPP, RW, NX, P + 1>(W, l, rho_unused);
}

// Rows y0 = l RW .. of phi_k, theta_k and gamma_{k+1} in registers (lane l of the group that works on knot k <= N-2)
template<class M, int LPP>
GATO_DEV void schur_coop_rows(const Buffers& bf, int N, int b, int k, int l, float dt, float (*phi)[2 * M::NQ], float (*th)[2 * M::NQ], float* gg)
{
    constexpr int NQ = M::NQ, NX = 2 * NQ, NU = NQ, RW = NX / LPP;
    static_assert(NX % LPP == 0 && (RW == NQ || 2 * RW == NQ), "a lane's rows must stay inside one half of the state");
    const size_t bk = (size_t)b * N + k;
    const float h2 = half_dt_sq(dt);
    const int y0 = l * RW;
    const bool upper = y0 < NQ;                // rows in the q half
    const bool odd = (RW != NQ) && (l & 1);    // y0 % NQ == RW
    const float coef = upper ? h2 : dt;

    {
        float Dm[3 * NQ * NQ], ri[NU];
        load_vec<3 * NQ * NQ, 3 * NQ * NQ>(Dm, bf.D + bk * 3 * NQ * NQ);
        load_vec<NU, NU>(ri, bf.Rdi + bk * NU);
        // own rows of A_k and B_k (row r: r % nq is i or RW + i, picked by lane parity)
        float Ar[RW][NX], Br[RW][NU];
#pragma unroll
        for (int i = 0; i < RW; i++) {
#pragma unroll
            for (int c = 0; c < NX; c++) {
                float d = Dm[c * NQ + i];
                if constexpr (RW != NQ) d = odd ? Dm[c * NQ + RW + i] : d;
                // A_elem with a lane-dependent row: delta_rc, + dt at (r, r + nq) in the q half, then + coef * d
                float v0 = 0.f;
                if (c < NQ) {
                    if (c == i) v0 = (upper && !odd) ? 1.0f : 0.f;
                    if (RW != NQ && c == RW + i) v0 = (upper && odd) ? 1.0f : 0.f;
                } else {
                    if (c - NQ == i) v0 = !odd ? (upper ? dt : 1.0f) : 0.f;
                    if (RW != NQ && c - NQ == RW + i) v0 = odd ? (upper ? dt : 1.0f) : 0.f;
                }
                Ar[i][c] = v0 + coef * d;
            }
#pragma unroll
            for (int c = 0; c < NU; c++) {
                float d = Dm[2 * NQ * NQ + c * NQ + i];
                if constexpr (RW != NQ) d = odd ? Dm[2 * NQ * NQ + c * NQ + RW + i] : d;
                Br[i][c] = coef * d;
            }
        }
        if (opaque_true()) {
            float Qi[NQ * NQ], di[NQ];
            load_vec<NQ * NQ, NQ * NQ>(Qi, bf.Qqi + bk * NQ * NQ);
            load_vec<NQ, NQ>(di, bf.Qdi + bk * NQ);
#pragma unroll
            for (int i = 0; i < RW; i++)
#pragma unroll
                for (int c = 0; c < NQ; c++) {
                    float sacc = 0.f;
#pragma unroll
                    for (int j = 0; j < NQ; j++) sacc += Ar[i][j] * Qi[c * NQ + j];
                    phi[i][c] = sacc;
                    phi[i][NQ + c] = Ar[i][NQ + c] * di[c];
                }
        }
        // own rows of Q_{k+1}^-1: tq (q half, zero in qd-half lanes) and the single diagonal entry td (qd half)
        float tq[RW][NQ], td[RW];
        if (opaque_true()) {
            float Qi1[NQ * NQ], di1[NQ];
            load_vec<NQ * NQ, NQ * NQ>(Qi1, bf.Qqi + (bk + 1) * NQ * NQ);
            load_vec<NQ, NQ>(di1, bf.Qdi + (bk + 1) * NQ);
#pragma unroll
            for (int i = 0; i < RW; i++) {
#pragma unroll
                for (int x = 0; x < NQ; x++) {
                    float v = Qi1[x * NQ + i];
                    if constexpr (RW != NQ) v = odd ? Qi1[x * NQ + RW + i] : v;
                    tq[i][x] = upper ? v : 0.f;
                }
                float dv = di1[i];
                if constexpr (RW != NQ) dv = odd ? di1[RW + i] : dv;
                td[i] = upper ? 0.f : dv;
            }
        }
        // theta rows: Q_{k+1}^-1 + phi A^T + (B R^-1) B^T, one column x at a time (A, B rows of x rebuilt from D)
        if (opaque_true()) {
            float Bri[RW][NU];
#pragma unroll
            for (int i = 0; i < RW; i++)
#pragma unroll
                for (int j = 0; j < NU; j++) Bri[i][j] = Br[i][j] * ri[j];
#pragma unroll
            for (int x = 0; x < NX; x++) {
                float Ax[NX], Bx[NU];
#pragma unroll
                for (int j = 0; j < NX; j++) Ax[j] = A_elem<NQ>(Dm, x, j, dt, h2);
#pragma unroll
                for (int j = 0; j < NU; j++) Bx[j] = B_elem<NQ>(Dm, x, j, dt, h2);
#pragma unroll
                for (int i = 0; i < RW; i++) {
                    float sacc = 0.f, s2 = 0.f;
#pragma unroll
                    for (int j = 0; j < NX; j++) sacc += phi[i][j] * Ax[j];
#pragma unroll
                    for (int j = 0; j < NU; j++) s2 += Bri[i][j] * Bx[j];
                    float t = 0.f;
                    if (x < NQ) {
                        t = tq[i][x];
                    } else {
                        if (x - NQ == i) t = !odd ? td[i] : 0.f;
                        if (RW != NQ && x - NQ == RW + i) t = odd ? td[i] : 0.f;
                    }
                    t += sacc;
                    t += s2;
                    th[i][x] = t;
                }
            }
            // gamma_{k+1} rows
            float qk[NX], qk1[NX], rk[NU], ck1[NX];
            load_vec<NX, NX>(qk, bf.q + bk * NX);
            load_vec<NX, NX>(qk1, bf.q + (bk + 1) * NX);
            load_vec<NU, NU>(rk, bf.r + bk * NU);
            load_vec<NX, NX>(ck1, bf.c + (bk + 1) * NX);
#pragma unroll
            for (int i = 0; i < RW; i++) {
                // own entries of c_{k+1} and q_{k+1} (row y0 + i: one of LPP candidates)
                float cy = ck1[i], qy = qk1[i];
#pragma unroll
                for (int m = 1; m < LPP; m++) {
                    cy = (l == m) ? ck1[m * RW + i] : cy;
                    qy = (l == m) ? qk1[m * RW + i] : qy;
                }
                float g1 = -1.0f * cy;
                float sq = 0.f;
#pragma unroll
                for (int j = 0; j < NQ; j++) sq += tq[i][j] * qk1[j];
                const float sd = td[i] * qy;
                g1 += upper ? sq : sd;
                float sacc = 0.f;
#pragma unroll
                for (int j = 0; j < NX; j++) sacc += phi[i][j] * qk[j];
                g1 += -sacc;
                sacc = 0.f;
#pragma unroll
                for (int j = 0; j < NU; j++) sacc += Bri[i][j] * rk[j];
                g1 += -sacc;
                gg[i] = -1.0f * g1;
            }
        }
    }
}
// th <- (theta_k + rho I_q)^-1 across the group (P^-1 row k+1 main = -th, schur_linsys.cuh:150-164)
template<class M, int LPP>
GATO_DEV void schur_coop_pinv(float (*th)[2 * M::NQ], int l, float rho)
{
    constexpr int NQ = M::NQ, NX = 2 * NQ, RW = NX / LPP;
    const bool upper = l * RW < NQ;
    const bool odd = (RW != NQ) && (l & 1);
#pragma unroll
    for (int i = 0; i < RW; i++) {
        // + rho on the first nq diagonal entries: row y0 + i, column y0 + i
        if constexpr (RW == NQ) {
            th[i][i] += upper ? rho : 0.f;
        } else {
            th[i][i] += (upper && !odd) ? rho : 0.f;
            th[i][RW + i] += (upper && odd) ? rho : 0.f;
        }
    }
    gj_coop_step<LPP, RW, NX, 0>(th, l, 0.f);
}

template<class M, int LPP>
__global__ __launch_bounds__(256) void schurq_kernel(Buffers bf, int N, int B, float dt, int write_right)
{
    constexpr int NQ = M::NQ, NX = 2 * NQ, NU = NQ, RW = NX / LPP, BLK = NX * NX, BROW = 3 * NX * NX;
    if (bf.ctrl->done) return;
    if (blockIdx.y == 1) {  // the Q_0 rows: one lane per trajectory
        const int b = blockIdx.x * blockDim.x + threadIdx.x;
        if (b >= B) return;
        schur_row0<M>(bf, N, b, bf.rho[b], bf.S + (size_t)b * N * BROW, bf.Pinv + (size_t)b * N * BROW, bf.gamma + (size_t)b * (N + 2) * NX);
        return;
    }
    const int g = blockIdx.x * blockDim.x + threadIdx.x;
    const int l = threadIdx.x % LPP;
    const int pidx = g / LPP;
    const int k = pidx % N, b = pidx / N;
    if (b >= B || k == N - 1) return;  // whole groups are in or out
    const int y0 = l * RW;
    float phi[RW][NX], th[RW][NX], gg[RW];
    schur_coop_rows<M, LPP>(bf, N, b, k, l, dt, phi, th, gg);
    float* S = bf.S + (size_t)b * N * BROW;
    float* Sk = S + (size_t)k * BROW;
    float* Sk1 = S + (size_t)(k + 1) * BROW;
    if (opaque_true()) {
        // S row k+1: [phi | -theta] (own rows, 16-byte stores); S row k right block = phi^T: the lane's RW rows of phi are RW
        // consecutive entries of every transposed row
#pragma unroll
        for (int i = 0; i < RW; i++) {
            float row[NX];
#pragma unroll
            for (int x = 0; x < NX; x++) row[x] = -th[i][x];
            store_vec<NX, NX>(Sk1 + (size_t)(y0 + i) * NX, phi[i]);
            store_vec<NX, NX>(Sk1 + BLK + (size_t)(y0 + i) * NX, row);
        }
        if (write_right) {  // the symmetric-storage PCG kernel forms the right blocks' products from the left blocks: no need to store them
#pragma unroll
            for (int x = 0; x < NX; x++) {
                float* dst = Sk + 2 * BLK + (size_t)x * NX + y0;
#pragma unroll
                for (int i = 0; i < RW; i++) dst[i] = phi[i][x];
            }
        }
        float* gam = bf.gamma + (size_t)b * (N + 2) * NX + (size_t)(k + 2) * NX + y0;
#pragma unroll
        for (int i = 0; i < RW; i++) gam[i] = gg[i];
    }
    if (opaque_true()) {
        schur_coop_pinv<M, LPP>(th, l, bf.rho[b]);
        float* Pk1 = bf.Pinv + ((size_t)b * N + k + 1) * BROW;
#pragma unroll
        for (int i = 0; i < RW; i++) {
            float row[NX];
#pragma unroll
            for (int x = 0; x < NX; x++) row[x] = -th[i][x];
            store_vec<NX, NX>(Pk1 + BLK + (size_t)(y0 + i) * NX, row);
        }
    }
}

// ---- one ROW per lane: groups of 16 lanes (nx of them active) per (b,k) ----------------------------------------------------------
// For nx = 14 (iiwa14) the 4-lane split does not exist and 2 lanes x 7 rows need ~400 registers.  Here lane l < nx owns row l of
// phi_k, theta_k and (theta_k + rho I_q)^-1, 16 x the wavefronts of a lane-per-knot kernel.  Everything a knot's rows read -- D_k, the
// inverses of knots k and k+1, q, r, c -- is fetched ONCE by the group into an LDS record with independent, coalesced loads (one
// memory latency; the first version chained six dependent fetch phases at two wavefronts per SIMD and spent 80 us per launch on
// them) and the rows are formed from LDS: no 147-register copy of D, 4+ wavefronts per SIMD.  The arithmetic and its order are
// unchanged.  The pivot row of each elimination step is broadcast inside the group with ds_bpermute (__shfl, width 16).
// ROW0: the launch also forms the Q_0 rows (grid.y = 2; stage tests) -- in a solve the assembly kernel's cost task forms them, and the
// lane-per-trajectory code (a 7 x 7 Gauss-Jordan in registers) does not set this kernel's register count.
template<class M, bool ROW0>
__global__ __launch_bounds__(256) void schur1_kernel(Buffers bf, int N, int B, float dt, int write_right)
{
    constexpr int NQ = M::NQ, NX = 2 * NQ, NU = NQ, BLK = NX * NX, BROW = 3 * NX * NX, ND = 3 * NQ * NQ, NQQ = NQ * NQ;
    static_assert(NX <= 16, "one group of 16 lanes per knot");
    // The inputs of the workgroup's 16 consecutive knots (and of the knot after them: Q^-1, q and c of knot k+1) are staged in LDS FIELD
    // by field: each field of 16 (17) consecutive knots is one contiguous, 16-byte aligned run in global memory, copied by the whole
    // workgroup with 16-byte loads (per knot and lane-group it was ~20 four-byte loads per lane).  A knot's piece of a field starts
    // at an odd stride (147, 49, 7) or a small even one (14): the 16 groups spread over the banks.
    __shared__ __attribute__((aligned(16))) float sD[16 * ND];
    __shared__ __attribute__((aligned(16))) float sQi[17 * NQQ + 3];
    __shared__ __attribute__((aligned(16))) float sDi[17 * NQ + 1];
    __shared__ __attribute__((aligned(16))) float sRi[16 * NU];
    __shared__ __attribute__((aligned(16))) float sQ[17 * NX + 2];
    __shared__ __attribute__((aligned(16))) float sR[16 * NU];
    __shared__ __attribute__((aligned(16))) float sC[17 * NX + 2];
    // A_k and B_k of the workgroup's 16 knots, formed ONCE (every lane its own row) and read by the theta columns as 16-byte vectors: row x
    // = [A_k[x][0..nx) | pad to 16 | B_k[x][0..nu) | pad to 8]; the group stride (+4) keeps the four groups of a wavefront on different banks
    constexpr int ABR = 24, ABG = NX * ABR + 4;
    static_assert(NX <= 16 && NU <= 8, "row layout of sAB");
    __shared__ __attribute__((aligned(16))) float sAB[16 * ABG];
    if (bf.ctrl->done) return;
    if constexpr (ROW0) {
        if (blockIdx.y == 1) {  // the Q_0 rows: one lane per trajectory
            const int b = blockIdx.x * blockDim.x + threadIdx.x;
            if (b >= B) return;
            schur_row0<M>(bf, N, b, bf.rho[b], bf.S + (size_t)b * N * BROW, bf.Pinv + (size_t)b * N * BROW, bf.gamma + (size_t)b * (N + 2) * NX);
            return;
        }
    }
    const int g = blockIdx.x * blockDim.x + threadIdx.x;
    const int l = threadIdx.x & 15;
    const int pidx = g >> 4;
    const int k = pidx % N, b = pidx / N;
    const bool live = b < B && k < N - 1;   // whole groups are in or out; they still reach the barrier
    {
        const long total = (long)B * N;
        const long k0 = (long)blockIdx.x * 16;                       // first knot of the workgroup (flat index b N + k)
        auto stage = [&](float* dst, const float* field, int per, int knots) {
            long n = (k0 + knots <= total ? (long)knots : total - k0) * per;   // floats available
            if (n < 0) n = 0;
            const float* src = field + k0 * per;                     // k0 is a multiple of 16: 16-byte aligned for every `per`
            const int n4 = (int)(n / 4);
            for (int i = threadIdx.x; i < n4; i += blockDim.x) reinterpret_cast<real4*>(dst)[i] = reinterpret_cast<const real4*>(src)[i];
            for (int i = 4 * n4 + threadIdx.x; i < (int)n; i += blockDim.x) dst[i] = src[i];
        };
        stage(sD, bf.D, ND, 16);
        stage(sQi, bf.Qqi, NQQ, 17);
        stage(sDi, bf.Qdi, NQ, 17);
        stage(sRi, bf.Rdi, NU, 16);
        stage(sQ, bf.q, NX, 17);
        stage(sR, bf.r, NU, 16);
        stage(sC, bf.c, NX, 17);
    }
    const int grp = threadIdx.x >> 4;
    __syncthreads();
    // rows of nx = 14 floats are 56-byte pieces: with GATO_SCHUR1_STAGE the three blocks a knot writes go through the wavefront's own part of
    // sAB (free once the theta columns are formed) and leave as whole 784-byte blocks in 16-byte stores.  Every lane then stays to the end
    // (the copy-out is by wavefront); dead groups compute on whatever LDS holds and store nothing.
    constexpr bool STAGE = GATO_SCHUR1_STAGE && (NX % 4 != 0) && (BLK % 4 == 0) && (NX % 2 == 0);
    if constexpr (!STAGE) {
        if (!live) return;
    }
    const bool act = live && l < NX;
    const int y = l < NX ? l : 0;      // the 16 - nx spare lanes shadow row 0 and store nothing
    const bool upper = y < NQ;         // row in the q half
    const int rm = upper ? y : y - NQ;
    const float h2 = half_dt_sq(dt);
    const float coef = upper ? h2 : dt;
    const float* Dm = sD + grp * ND;

    float Ar[NX], Bri[NU], phi[NX], th[NX], gg;
    float* ABg = sAB + grp * ABG;
    {
        // own row of A_k (A_elem with a lane-dependent row) and of B_k, left in LDS for the group's theta columns; then B_k R_k^-1
        float ab[ABR];
#pragma unroll
        for (int c = 0; c < NX; c++) {
            const float d = Dm[c * NQ + rm];
            float v0 = (c == y) ? 1.0f : 0.f;
            if (c >= NQ) v0 = (upper && c - NQ == y) ? dt : v0;
            Ar[c] = v0 + coef * d;
            ab[c] = Ar[c];
        }
#pragma unroll
        for (int c = NX; c < 16; c++) ab[c] = 0.f;
#pragma unroll
        for (int c = 0; c < NU; c++) {
            const float bu = coef * Dm[2 * NQQ + c * NQ + rm];   // B_elem(y, c)
            ab[16 + c] = bu;
            Bri[c] = bu * sRi[grp * NU + c];
        }
#pragma unroll
        for (int c = 16 + NU; c < ABR; c++) ab[c] = 0.f;
        if (act) {
#pragma unroll
            for (int c = 0; c < ABR; c += 4) *reinterpret_cast<real4*>(ABg + y * ABR + c) = make_real4(ab[c], ab[c + 1], ab[c + 2], ab[c + 3]);
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");   // the rows are read by the other lanes of the group (same wavefront)
    }
    {
#pragma unroll
        for (int c = 0; c < NQ; c++) {
            float sacc = 0.f;
#pragma unroll
            for (int j = 0; j < NQ; j++) sacc += Ar[j] * sQi[grp * NQQ + c * NQ + j];
            phi[c] = sacc;
            phi[NQ + c] = Ar[NQ + c] * sDi[grp * NQ + c];
        }
    }
    // own row of Q_{k+1}^-1: tq (q half, zero in qd-half lanes) and the single diagonal entry td (qd half)
    float tq[NQ], td;
    {
#pragma unroll
        for (int x = 0; x < NQ; x++) {
            const float v = sQi[(grp + 1) * NQQ + x * NQ + rm];
            tq[x] = upper ? v : 0.f;
        }
        const float dv = sDi[(grp + 1) * NQ + rm];
        td = upper ? 0.f : dv;
    }
    {
#pragma unroll
        for (int x = 0; x < NX; x++) {
            // row x of A_k and B_k: six 16-byte LDS reads, the same for every lane of the group (a broadcast).  The values are the ones
            // A_elem / B_elem form (same expression, same bits); the sums run in the same order.
            float arow[ABR];
#pragma unroll
            for (int c = 0; c < ABR; c += 4) {
                const real4 v = *reinterpret_cast<const real4*>(ABg + x * ABR + c);
                arow[c] = v.x; arow[c + 1] = v.y; arow[c + 2] = v.z; arow[c + 3] = v.w;
            }
            float sacc = 0.f, s2 = 0.f;
#pragma unroll
            for (int j = 0; j < NX; j++) sacc += phi[j] * arow[j];
#pragma unroll
            for (int j = 0; j < NU; j++) s2 += Bri[j] * arow[16 + j];
            float t;
            if (x < NQ) t = tq[x];
            else t = (x - NQ == rm) ? td : 0.f;
            t += sacc;
            t += s2;
            // the column is finished HERE: left to itself the compiler sinks the fourteen sums to the LDS stores below and keeps all 336 loaded
            // A/B entries alive until then (256 VGPR + 132 AGPR instead of 98)
            if constexpr (STAGE) asm volatile("" : "+v"(t));
            th[x] = t;
        }
    }
    {
        const float cy = sC[(grp + 1) * NX + y], qy = sQ[(grp + 1) * NX + y];
        float g1 = -1.0f * cy;
        float sq = 0.f;
#pragma unroll
        for (int j = 0; j < NQ; j++) sq += tq[j] * sQ[(grp + 1) * NX + j];
        const float sd = td * qy;
        g1 += upper ? sq : sd;
        float sacc = 0.f;
#pragma unroll
        for (int j = 0; j < NX; j++) sacc += phi[j] * sQ[grp * NX + j];
        g1 += -sacc;
        sacc = 0.f;
#pragma unroll
        for (int j = 0; j < NU; j++) sacc += Bri[j] * sR[grp * NU + j];
        g1 += -sacc;
        gg = -1.0f * g1;
    }
    float* S = bf.S + (size_t)b * N * BROW;
    // staged copy-out: the wavefront's four knots are consecutive block rows (flat index b N + k + 1)
    float* wst = sAB + (grp & ~3) * ABG;
    const unsigned long long livem = __ballot(live);
    auto put_row = [&](const float* rowv, float sgn) {
        if (l < NX) {
            float* d = wst + (grp & 3) * BLK + y * NX;
#pragma unroll
            for (int c = 0; c < NX; c += 2) *reinterpret_cast<real2*>(d + c) = make_real2(sgn * rowv[c], sgn * rowv[c + 1]);
        }
    };
    auto copy_out = [&](float* field, int bt) {
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        const long row1 = (long)blockIdx.x * 16 + (grp & ~3) + 1;
        for (int i = threadIdx.x & 63; i < BLK; i += 64) {       // 4 blocks x BLK/4 vectors
            const int gq = i / (BLK / 4), j = i - gq * (BLK / 4);
            if ((livem >> (gq * 16)) & 1)
                reinterpret_cast<real4*>(field + (size_t)(row1 + gq) * BROW + bt * BLK)[j] = reinterpret_cast<const real4*>(wst + gq * BLK)[j];
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    };
    if constexpr (STAGE) {
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");   // the group's reads of its A/B rows are done
        put_row(phi, 1.0f);
        copy_out(bf.S, 0);
        put_row(th, -1.0f);
        copy_out(bf.S, 1);
    }
    if (act) {
        float* Sk = S + (size_t)k * BROW;
        float* Sk1 = S + (size_t)(k + 1) * BROW;
        if constexpr (!STAGE) {
            float row[NX];
#pragma unroll
            for (int x = 0; x < NX; x++) row[x] = -th[x];
            gstore_vec<NX>(Sk1 + (size_t)y * NX, phi);          // the group's 14 rows of a block: one contiguous run
            gstore_vec<NX>(Sk1 + BLK + (size_t)y * NX, row);
        }
        if (write_right) {
#pragma unroll
            for (int x = 0; x < NX; x++) Sk[2 * BLK + (size_t)x * NX + y] = phi[x];  // right block of row k = phi^T
        }
        bf.gamma[(size_t)b * (N + 2) * NX + (size_t)(k + 2) * NX + y] = gg;
    }
    // (theta_k + rho I_q)^-1 by Gauss-Jordan across the group (schur_linsys.cuh:150-164)
    {
        const float rho = bf.rho[live ? b : 0];
#pragma unroll
        for (int c = 0; c < NQ; c++) th[c] += (c == y) ? rho : 0.f;  // first nq diagonal entries (only rows y < nq have c == y there)
#pragma unroll
        for (int p = 0; p < NX; p++) {
            float prow[NX];
            const bool owner = (y == p);
            if constexpr (STAGE && GATO_SCHUR1_GJ_LDS) {
                // the pivot row through 16 floats of the wavefront's LDS slice (free between the staged copies) instead of fourteen ds_bpermute:
                // its owner writes it (four stores), everybody reads it back as a broadcast (four 16-byte reads); LDS operations of one
                // wavefront execute in order, the same values arrive
                float* slot = wst + (grp & 3) * 16;
                if (owner) {
#pragma unroll
                    for (int c = 0; c < NX; c += 2) *reinterpret_cast<real2*>(slot + c) = make_real2(th[c], th[c + 1]);
                }
                __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
#pragma unroll
                for (int c = 0; c < NX; c += 2) {
                    const real2 v = *reinterpret_cast<const real2*>(slot + c);
                    prow[c] = v.x; prow[c + 1] = v.y;
                }
                __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            } else {
#pragma unroll
                for (int c = 0; c < NX; c++) prow[c] = __shfl(th[c], p, 16);
            }
            const float pvInv = 1.0f / prow[p];
            const float f = th[p] * pvInv;
#pragma unroll
            for (int c = 0; c < NX; c++) {
                float x, yv;
                if (c == p) {
                    yv = 0.f - f;
                    x = 1.0f;
                } else {
                    x = th[c];
                    yv = x - f * prow[c];
                }
                const float piv = x * pvInv;
                th[c] = owner ? piv : yv;
            }
        }
        if constexpr (STAGE) {
            put_row(th, -1.0f);
            copy_out(bf.Pinv, 1);
        } else if (act) {
            float* Pk1 = bf.Pinv + ((size_t)b * N + k + 1) * BROW;
            float row[NX];
#pragma unroll
            for (int x = 0; x < NX; x++) row[x] = -th[x];
            gstore_vec<NX>(Pk1 + BLK + (size_t)y * NX, row);
        }
    }
}

// stair preconditioner off-diagonals (formSchurSystemBatchedKernel2, schur_linsys.cuh:213-260): for k <= N-2
//   res = Pm_{k+1} (phi_k Pm_k);  P^-1 row k+1 left = -res, row k right = -res^T   (Pm = the STORED, sign-carrying diagonals)
template<class M>
__global__ __launch_bounds__(64) void schur2_kernel(Buffers bf, int N, int B, int write_right)
{
    constexpr int NQ = M::NQ, NX = 2 * NQ, BLK = NX * NX, BROW = 3 * NX * NX;
    if (bf.ctrl->done) return;
    const int g = blockIdx.x * blockDim.x + threadIdx.x;
    const int k = g % N, b = g / N;
    if (b >= B || k >= N - 1) return;
    const float* S = bf.S + (size_t)b * N * BROW;
    float* P = bf.Pinv + (size_t)b * N * BROW;
    const float* Pk = P + (size_t)k * BROW;
    float* Pk1 = P + (size_t)(k + 1) * BROW;
    const float* Sk1 = S + (size_t)(k + 1) * BROW;
    float scr[NX][NX];  // phi_k Pm_k
    {
        float tkm1[NX][NX];
#pragma unroll
        for (int y = 0; y < NX; y++) load_vec<NX, NX>(tkm1[y], Pk + BLK + y * NX);
#pragma unroll
        for (int y = 0; y < NX; y++) {
            float ph[NX];
            load_vec<NX, NX>(ph, Sk1 + y * NX);
#pragma unroll
            for (int x = 0; x < NX; x++) {
                float s = 0.f;
#pragma unroll
                for (int j = 0; j < NX; j++) s += ph[j] * tkm1[j][x];
                scr[y][x] = s;
            }
        }
    }
    float* Pkw = P + (size_t)k * BROW;
#pragma unroll
    for (int y = 0; y < NX; y++) {
        float tk[NX], res[NX];
        load_vec<NX, NX>(tk, Pk1 + BLK + y * NX);
#pragma unroll
        for (int x = 0; x < NX; x++) {
            float s = 0.f;
#pragma unroll
            for (int j = 0; j < NX; j++) s += tk[j] * scr[j][x];
            res[x] = -s;
        }
        store_vec<NX, NX>(Pk1 + y * NX, res);  // left of row k+1, row y
        if (write_right) {
#pragma unroll
            for (int x = 0; x < NX; x++) Pkw[2 * BLK + x * NX + y] = res[x];  // right of row k: (row x, col y) = -res[y][x]
        }
    }
}

// =========================================================================================================================
// PCG on the block-tridiagonal Schur system S lambda = gamma, preconditioner P^-1 (pcg.cuh:14-148).
// One workgroup per trajectory; thread t owns rows t, t + T (RPT <= 2) with their S / P^-1 rows in registers.
// LDS: two padded vectors of (N+2) nx floats + reduction partials.
// =========================================================================================================================
// The two quotients of a PCG iteration (alpha = rho / pAp, beta = rho' / rho) as numerator x v_rcp_f32(denominator): 2 instructions
// instead of the 12 of an IEEE division, on the per-wavefront instruction chain that bounds the launch.  v_rcp_f32 is good to 1 ulp;
// the reference is built with -use_fast_math (CMakeLists.txt:22), whose division is the 2-ulp __fdividef.
GATO_DEV float pcg_div(float a, float b) { return a * fast_rcp(b); }

template<int NXT> GATO_DEV float row_dot(const float* __restrict__ row, const float* __restrict__ win)
{
    // win: LDS window of 3 nx floats starting at the left-neighbour block (16-byte aligned when nx % 4 == 0).
    // Four independent accumulators: the FMA chain no longer serialises behind the LDS reads.
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    if constexpr ((3 * NXT) % 4 == 0 && NXT % 4 == 0) {
#pragma unroll
        for (int c = 0; c < 3 * NXT / 4; c++) {
            const real4 v = reinterpret_cast<const real4*>(win)[c];
            s0 += row[4 * c] * v.x;
            s1 += row[4 * c + 1] * v.y;
            s2 += row[4 * c + 2] * v.z;
            s3 += row[4 * c + 3] * v.w;
        }
    } else {
#pragma unroll
        for (int c = 0; c < 3 * NXT / 2; c++) {
            const real2 v = reinterpret_cast<const real2*>(win)[c];
            if (c & 1) { s2 += row[2 * c] * v.x; s3 += row[2 * c + 1] * v.y; }
            else { s0 += row[2 * c] * v.x; s1 += row[2 * c + 1] * v.y; }
        }
    }
    return (s0 + s1) + (s2 + s3);
}

// wave64 sum with DPP row operations (no LDS crossbar traffic, unlike ds_bpermute-based __shfl): quad butterflies, half-row and
// row mirrors, then row_bcast15 / row_bcast31 accumulate the four 16-lane rows into lane 63, which is broadcast back.
GATO_DEV float wave_sum(float v)
{
#define GATO_DPP_ADD(ctrl) v += dpp_get<ctrl>(v)
    GATO_DPP_ADD(0xB1);   // quad_perm [1,0,3,2]
    GATO_DPP_ADD(0x4E);   // quad_perm [2,3,0,1]
    GATO_DPP_ADD(0x141);  // row_half_mirror
    GATO_DPP_ADD(0x140);  // row_mirror: every lane holds its row's sum
#undef GATO_DPP_ADD
    // the four row sums through the scalar unit (independent v_readlane's) instead of two more dependent DPP steps:
    // 52 vs 59 ns per reduction in isolation (tools/microbench/wave_sum.hip)
    const float r0 = lane_read(v, 0);
    const float r1 = lane_read(v, 16);
    const float r2 = lane_read(v, 32);
    const float r3 = lane_read(v, 48);
    return (r3 + r2) + (r1 + r0);   // the association of the row_bcast:15 / row_bcast:31 chain it replaces: same bits
}

// block-wide sum; `part` has 16 slots (unused ones zeroed once by the caller), read back with four 16-byte LDS loads
// PARTS = 16-byte groups that can hold a wavefront's partial (<= 4 PARTS wavefronts in the workgroup): the skipped ones are exact
// zeros, so every PARTS gives the same bits.
// tx: the thread's index inside the group of wavefronts that sums (the workgroup)
// TWO: the summing group is exactly two wavefronts (C2's fused kernel, instantiated with MAXT = 128): their partials by one 8-byte read and one add.
// The other slots are exact zeros, so (a.x + a.y) + (0 + 0) has the same bits -- two dependent adds and half the read off the chain behind every
// reduction barrier.  A COMPILE-TIME switch: the same thing behind a wavefront-uniform run-time branch measured 0.75 % SLOWER than not having it (the
// branch sits on that same chain), the instantiation +0.8 % (profiles/r06_c2_two_form_abc.json: three builds round-robin on one box, same bits)
// shadow: work of the caller that does not depend on the sum, issued between the partial's store and the barrier -- i.e. while the store completes
// (s_waitcnt lgkmcnt(0) sits in front of s_barrier).  The PCG loops put the x update there: sunk by the compiler it sat on the dependent chain in front of
// the next store (C2 +0.4 %, same bits: profiles/r06_c2_chain5.json)
struct NoShadow { GATO_DEV void operator()() const {} };
template<int PARTS = 4, bool TWO = false, class F = NoShadow> GATO_DEV float block_sum(float v, float* part, unsigned tx = threadIdx.x, F shadow = F())
{
    v = wave_sum(v);
    // EVERY lane stores the wavefront's sum (the same value to the same address: one LDS write, no conflict) instead of lane 0 under an exec mask: the
    // s_and_saveexec / s_cbranch_execz pair and the three adds it guarded sat on the dependent chain in front of the barrier -- C2 +1.9 %, same bits
    // (round 6, profiles/r06_c2_micro_variants.json)
    part[tx >> 6] = v;
    asm volatile("" ::: "memory");
    shadow();   // work that does not depend on the sum: issued while the store completes, in front of the barrier's wait
    __syncthreads();
    if constexpr (TWO) {
        const real2 a2 = reinterpret_cast<const real2*>(part)[0];
        return a2.x + a2.y;
    }
    const real4 a = reinterpret_cast<const real4*>(part)[0];
    float r = (a.x + a.y) + (a.z + a.w);
    if constexpr (PARTS == 1) return r;
    const real4 b = reinterpret_cast<const real4*>(part)[1];
    r = r + ((b.x + b.y) + (b.z + b.w));
    if constexpr (PARTS == 2) return r;
    const real4 c = reinterpret_cast<const real4*>(part)[2], d = reinterpret_cast<const real4*>(part)[3];
    return r + (((c.x + c.y) + (c.z + c.w)) + ((d.x + d.y) + (d.z + d.w)));
}

// row `y` of block row `k` of a block-major matrix as the 3 nx floats [left | main | right] the row dots work on
template<int NX> GATO_DEV void load_btd_row(float* dst, const float* __restrict__ Mb, int k, int y)
{
    constexpr int BLK = NX * NX;
    const float* base = Mb + (size_t)k * 3 * BLK + (size_t)y * NX;
    load_vec<NX, NX>(dst, base);
    load_vec<NX, NX>(dst + NX, base + BLK);
    load_vec<NX, NX>(dst + 2 * NX, base + 2 * BLK);
}

// The tail of a PCG iteration (pcg.cuh:127-141) and the SHAPE of its loop (round 6).  beta = rho' / rho and p = z + beta p are formed BEFORE the exit
// test is branched on -- the quotient's v_rcp_f32 does not depend on rho' and issues while the partial sums are still in flight, the FMAs issue beside the
// compare -- p is updated IN PLACE by one three-address v_fma_f32 per entry (left to itself the compiler takes the two-address v_fmac_f32 into z's register
// and copies it back: four v_mov_b32 on the dependent chain), and the two ways out of the loop -- the test and the iteration cap -- are ONE branch.  p is
// dead on the way out.  Same arithmetic, same bits.  The loop is `if (max_iters > 0) for (;;) { iters++; ... GATO_PCG_TAIL(...) }`: in that single-exit
// form the compiler issues all nine 16-byte window reads of a product up front (255 registers) where the two-exit loop staged them 4 + 2 + 2 + 1 behind
// four waits (246 registers) -- ~300 cycles per iteration: C2's PCG launch 119.5 -> 104.5 us, the headline +9.8 % (profiles/r06_c2_chain4.json).
GATO_DEV void pcg_p_update(float& p, float beta, float z)
{
    if constexpr (sizeof(float) == 4) asm volatile("v_fma_f32 %0, %1, %0, %2" : "+v"(p) : "v"(beta), "v"(z));
    else { p = z + beta * p; asm volatile("" : "+v"(p)); }
}
#define GATO_PCG_TAIL(R, pv, zv, rho, rho_new, exit_thresh, iters, max_iters)                   \
    {                                                                                            \
        const float beta_ = pcg_div(rho_new, rho);                                               \
        _Pragma("unroll") for (int u_ = 0; u_ < R; u_++) pcg_p_update(pv[u_], beta_, zv[u_]);    \
        if ((fabsf(rho_new) < (exit_thresh)) | ((iters) == (max_iters))) break;                  \
        rho = rho_new;                                                                           \
    }

// RPT rows per thread; STREAM = false keeps the thread's S / P^-1 rows in registers, true re-reads them from global memory
// (L2 / Infinity Cache) for systems that do not fit one CU's register file (iiwa14 N = 128: 602 KB); MAXT = launch bound.
template<class M, int RPT, bool STREAM, int MAXT>
__global__ __launch_bounds__(MAXT) void pcg_kernel(Buffers bf, int N, int B, uint32_t max_iters, int sqp_iter)
{
    constexpr int NQ = M::NQ, NX = 2 * NQ, BR = 3 * NX, BROW = 3 * NX * NX;
    extern __shared__ __attribute__((aligned(16))) float lds[];
    if (bf.ctrl->done) return;
    const int b = blockIdx.x;
    const int T = blockDim.x;
    const int nrows = N * NX, vecp = (N + 2) * NX;
    float* va = lds;               // padded vector A (x, then p)
    float* vb = lds + vecp;        // padded vector B (r)
    float* partA = vb + ((vecp + 3) & ~3);  // [16], 16-byte aligned
    float* partB = partA + 16;               // [16]
    if (threadIdx.x < 32) partA[threadIdx.x] = 0.f;  // unused wave slots stay zero (published by the first barrier below)
    const float abs_tol = 1e-6f;
    uint32_t iters = 0;
    const bool skip = bf.converged[b] != 0;  // pcg.cuh:29-32

    if (!skip) {
        const float eps = bf.pcg_tol[b];
        const float* S = bf.S + (size_t)b * N * BROW;
        const float* P = bf.Pinv + (size_t)b * N * BROW;
        const float* gam = bf.gamma + (size_t)b * vecp;
        float* lam = bf.lambda + (size_t)b * vecp;

        constexpr int RB = STREAM ? 1 : BR;
        float Srow[RPT][RB], Prow[RPT][RB], xv[RPT], rv[RPT], pv[RPT], zv[RPT];
        int row[RPT];
        bool have[RPT];
        int rk[RPT], ry[RPT];   // block row and row inside it
#pragma unroll
        for (int u = 0; u < RPT; u++) {
            row[u] = threadIdx.x + u * T;
            have[u] = row[u] < nrows;
            const int rr = have[u] ? row[u] : 0;
            rk[u] = rr / NX;
            ry[u] = rr - rk[u] * NX;
            if constexpr (!STREAM) {
                load_btd_row<NX>(Srow[u], S, rk[u], ry[u]);
                load_btd_row<NX>(Prow[u], P, rk[u], ry[u]);
            }
            xv[u] = have[u] ? lam[NX + rr] : 0.f;
        }
        auto sdot = [&](int u, const float* win) -> float {
            if constexpr (STREAM) {
                float tmp[BR];
                load_btd_row<NX>(tmp, S, rk[u], ry[u]);
                return row_dot<NX>(tmp, win);
            } else {
                return row_dot<NX>(Srow[u], win);
            }
        };
        auto pdot = [&](int u, const float* win) -> float {
            if constexpr (STREAM) {
                float tmp[BR];
                load_btd_row<NX>(tmp, P, rk[u], ry[u]);
                return row_dot<NX>(tmp, win);
            } else {
                return row_dot<NX>(Prow[u], win);
            }
        };
        // zero the padding blocks of both vectors
        for (int i = threadIdx.x; i < NX; i += T) {
            va[i] = 0.f; vb[i] = 0.f;
            va[vecp - NX + i] = 0.f; vb[vecp - NX + i] = 0.f;
        }
#pragma unroll
        for (int u = 0; u < RPT; u++)
            if (have[u]) va[NX + row[u]] = xv[u];
        __syncthreads();
        // r = gamma - S x
#pragma unroll
        for (int u = 0; u < RPT; u++) {
            if (have[u]) {
                const int kb = row[u] / NX;
                rv[u] = gam[NX + row[u]] - sdot(u, va + kb * NX);
                vb[NX + row[u]] = rv[u];
            } else {
                rv[u] = 0.f;
            }
        }
        __syncthreads();
        float loc = 0.f;
#pragma unroll
        for (int u = 0; u < RPT; u++) {
            if (have[u]) {
                const int kb = row[u] / NX;
                zv[u] = pdot(u, vb + kb * NX);
            } else {
                zv[u] = 0.f;
            }
            pv[u] = zv[u];
            loc += rv[u] * zv[u];
        }
        float rho = block_sum(loc, partA);
        if (!(fabsf(rho) < abs_tol)) {
            const float rho_init = fabsf(rho);
            if (max_iters > 0) for (;;) {
                iters++;
#pragma unroll
                for (int u = 0; u < RPT; u++)
                    if (have[u]) va[NX + row[u]] = pv[u];
                __syncthreads();
                float Ap[RPT];
                loc = 0.f;
#pragma unroll
                for (int u = 0; u < RPT; u++) {
                    Ap[u] = have[u] ? sdot(u, va + (row[u] / NX) * NX) : 0.f;
                    loc += pv[u] * Ap[u];
                }
                const float pAp = block_sum(loc, partB);
                const float alpha = pcg_div(rho, pAp);
#pragma unroll
                for (int u = 0; u < RPT; u++) {
                    xv[u] += alpha * pv[u];
                    rv[u] -= alpha * Ap[u];
                    if (have[u]) vb[NX + row[u]] = rv[u];
                }
                __syncthreads();
                loc = 0.f;
#pragma unroll
                for (int u = 0; u < RPT; u++) {
                    zv[u] = have[u] ? pdot(u, vb + (row[u] / NX) * NX) : 0.f;
                    loc += rv[u] * zv[u];
                }
                const float rho_new = block_sum(loc, partA);
                GATO_PCG_TAIL(RPT, pv, zv, rho, rho_new, abs_tol + eps * rho_init, iters, max_iters)
            }
#pragma unroll
            for (int u = 0; u < RPT; u++)
                if (have[u]) lam[NX + row[u]] = xv[u];
        }
    }
    if (threadIdx.x == 0) {
        bf.pcg_iters[b] = iters;
        bf.st_pcg_iters[(size_t)sqp_iter * B + b] = (int32_t)iters;
        int conv = skip ? 1 : 0;
        if (iters == 0) { conv = 1; bf.converged[b] = 1; }  // bsqp.cuh:153-156 (kkt_tol is unused there)
        if (conv) atomicAdd(&bf.num_solved_w[sqp_iter], 1u);
    }
}

// ---- PCG, register-resident, contiguous rows ------------------------------------------------------------------------------
// Thread t owns the RPT CONSECUTIVE rows [t*RPT, (t+1)*RPT) of the system; NX % RPT == 0, so they lie in ONE block row and share
// the 3 nx-wide window of the input vector: one LDS read of the window feeds RPT rows (LDS traffic / RPT) and the RPT dot products
// give the FMA stream its ILP.  With RPT = 6 an indy7 N = 32 trajectory is ONE wavefront (64 lanes x 6 rows): 432 matrix registers
// per lane, no cross-wave barrier at all, and four trajectories co-resident per CU (one per SIMD).
// Packed FP32: each row keeps an (even, odd) pair of partial sums and advances it with v_pk_fma_f32 -- two FMAs per issued
// instruction; the matrix rows already sit in consecutive registers and the window arrives as 16-byte LDS reads, so no packing
// moves are needed.  The pair is added once at the end.
// Association of a row's 3 nx products (every register-resident PCG form shares it, so that they give the same bits): the row is cut
// into two halves of 3 nx / 2 columns; each half accumulates its even and its odd columns in sequence (one packed FMA chain), and
// the row sum is (even_lo + even_hi) + (odd_lo + odd_hi) -- the pair form of pcgc_kernel gives one half to each lane of a pair.
template<int NXT, int RPT, bool PACKED = true> GATO_DEV void rows_dot(const float (*rows)[3 * NXT], const float* __restrict__ win, float* acc)
{
    constexpr int HP = (3 * NXT) / 4;   // float pairs per half (the halves split between pairs for every even nx)
    static_assert((3 * NXT) % 4 == 0 || NXT % 4 != 0, "halves are whole pairs");
    if constexpr (!PACKED) {
        // scalar form of the same association
        float e[2][RPT], o[2][RPT];
#pragma unroll
        for (int u = 0; u < RPT; u++) e[0][u] = o[0][u] = e[1][u] = o[1][u] = 0.f;
#pragma unroll
        for (int c = 0; c < 3 * NXT / 2; c++) {
            const real2 v = reinterpret_cast<const real2*>(win)[c];
            const int hf = (NXT % 4 == 0 && c >= HP) ? 1 : 0;
#pragma unroll
            for (int u = 0; u < RPT; u++) {
                e[hf][u] = __builtin_fmaf(rows[u][2 * c], v.x, e[hf][u]);
                o[hf][u] = __builtin_fmaf(rows[u][2 * c + 1], v.y, o[hf][u]);
            }
        }
#pragma unroll
        for (int u = 0; u < RPT; u++) acc[u] = (NXT % 4 == 0) ? (e[0][u] + e[1][u]) + (o[0][u] + o[1][u]) : e[0][u] + o[0][u];
        return;
    }
    if constexpr (NXT % 4 == 0) {
        f32x2 a2[2][RPT];
#pragma unroll
        for (int u = 0; u < RPT; u++) a2[0][u] = a2[1][u] = f32x2{0.f, 0.f};
#pragma unroll
        for (int c = 0; c < 3 * NXT / 4; c++) {
            const real4 v = reinterpret_cast<const real4*>(win)[c];
            const f32x2 lo = {v.x, v.y}, hi = {v.z, v.w};
#pragma unroll
            for (int u = 0; u < RPT; u++) {
                f32x2& s0 = a2[(2 * c >= HP) ? 1 : 0][u];
                s0 = __builtin_elementwise_fma(f32x2{rows[u][4 * c], rows[u][4 * c + 1]}, lo, s0);
                f32x2& s1 = a2[(2 * c + 1 >= HP) ? 1 : 0][u];
                s1 = __builtin_elementwise_fma(f32x2{rows[u][4 * c + 2], rows[u][4 * c + 3]}, hi, s1);
            }
        }
#pragma unroll
        for (int u = 0; u < RPT; u++) {
            const f32x2 t = a2[0][u] + a2[1][u];   // one packed add: (even_lo + even_hi, odd_lo + odd_hi)
            acc[u] = t.x + t.y;
        }
    } else {
        f32x2 a2[RPT];
#pragma unroll
        for (int u = 0; u < RPT; u++) a2[u] = f32x2{0.f, 0.f};
#pragma unroll
        for (int c = 0; c < 3 * NXT / 2; c++) {
            const real2 v = reinterpret_cast<const real2*>(win)[c];
            const f32x2 w = {v.x, v.y};
#pragma unroll
            for (int u = 0; u < RPT; u++) a2[u] = __builtin_elementwise_fma(f32x2{rows[u][2 * c], rows[u][2 * c + 1]}, w, a2[u]);
        }
#pragma unroll
        for (int u = 0; u < RPT; u++) acc[u] = a2[u].x + a2[u].y;
    }
}

// nx % 4 != 0 (iiwa14, nx = 14): the LDS vectors keep every block at a stride of VS = 16 floats, so a row's window is three 16-byte
// aligned blocks read with ds_read_b128 (4 per block: 12 loads per product) instead of 21 eight-byte reads that hipcc pairs into
// ds_read2_b64 (half the LDS rate, twice the instructions).  Same association as rows_dot's nx % 4 != 0 form: ONE (even, odd) pair of
// partial sums per row, advanced over the 3 nx / 2 column pairs in sequence -- the same bits.
template<int NXT, int RPT, int VS> GATO_DEV void rows_dot_strided(const float (*rows)[3 * NXT], const float* __restrict__ win, float* acc)
{
    static_assert(NXT % 2 == 0 && VS % 4 == 0 && VS >= NXT, "blocks of whole pairs at a 16-byte stride");
    constexpr int LW = (NXT + 3) & ~3;   // floats fetched per block (whole 16-byte chunks; the stride may be larger: pcg_vec_stride)
    f32x2 a2[RPT];
#pragma unroll
    for (int u = 0; u < RPT; u++) a2[u] = f32x2{0.f, 0.f};
#pragma unroll
    for (int blk = 0; blk < 3; blk++) {
        float w[LW];
#pragma unroll
        for (int c = 0; c < LW / 4; c++) {
            const real4 v = reinterpret_cast<const real4*>(win + blk * VS)[c];
            w[4 * c] = v.x; w[4 * c + 1] = v.y; w[4 * c + 2] = v.z; w[4 * c + 3] = v.w;
        }
#pragma unroll
        for (int c = 0; c < NXT / 2; c++) {
            const f32x2 wv = {w[2 * c], w[2 * c + 1]};
#pragma unroll
            for (int u = 0; u < RPT; u++) a2[u] = __builtin_elementwise_fma(f32x2{rows[u][blk * NXT + 2 * c], rows[u][blk * NXT + 2 * c + 1]}, wv, a2[u]);
        }
    }
#pragma unroll
    for (int u = 0; u < RPT; u++) acc[u] = a2[u].x + a2[u].y;
}

// Same dot products with the rows' RIGHT block (the last nx entries) parked in LDS as real4 [chunk][thread] (conflict-free): the
// iteration loop then needs nx x RPT fewer registers.  Only for nx % 4 == 0.
template<int NXT, int RPT> GATO_DEV void rows_dot_parked(const float (*rows)[3 * NXT], const float* __restrict__ win, const real4* park, int T,
                                                         float* acc)
{
    static_assert(NXT % 4 == 0, "16-byte chunks");
    constexpr int CH = NXT / 4;  // chunks per block
    constexpr int HP = (3 * NXT) / 4;   // float pairs per half: rows_dot's association
    f32x2 a2[2][RPT];
#pragma unroll
    for (int u = 0; u < RPT; u++) a2[0][u] = a2[1][u] = f32x2{0.f, 0.f};
#pragma unroll
    for (int c = 0; c < 2 * CH; c++) {
        const real4 v = reinterpret_cast<const real4*>(win)[c];
        const f32x2 lo = {v.x, v.y}, hi = {v.z, v.w};
#pragma unroll
        for (int u = 0; u < RPT; u++) {
            f32x2& s0 = a2[(2 * c >= HP) ? 1 : 0][u];
            s0 = __builtin_elementwise_fma(f32x2{rows[u][4 * c], rows[u][4 * c + 1]}, lo, s0);
            f32x2& s1 = a2[(2 * c + 1 >= HP) ? 1 : 0][u];
            s1 = __builtin_elementwise_fma(f32x2{rows[u][4 * c + 2], rows[u][4 * c + 3]}, hi, s1);
        }
    }
#pragma unroll
    for (int c = 0; c < CH; c++) {
        const real4 v = reinterpret_cast<const real4*>(win)[2 * CH + c];
        const f32x2 lo = {v.x, v.y}, hi = {v.z, v.w};
#pragma unroll
        for (int u = 0; u < RPT; u++) {
            const real4 m = park[(u * CH + c) * T];
            a2[1][u] = __builtin_elementwise_fma(f32x2{m.x, m.y}, lo, a2[1][u]);   // the right block lies in the upper half (2 nx >= 3 nx / 2)
            a2[1][u] = __builtin_elementwise_fma(f32x2{m.z, m.w}, hi, a2[1][u]);
        }
    }
#pragma unroll
    for (int u = 0; u < RPT; u++) {
        const f32x2 t = a2[0][u] + a2[1][u];
        acc[u] = t.x + t.y;
    }
}

// FOLD: the stair off-diagonals of P^-1 (formSchurSystemBatchedKernel2, schur_linsys.cuh:213-260) are formed HERE, from the stored
// diagonal blocks, instead of by a kernel of their own: the workgroup holds every block row of the trajectory, so
// left_k = -Pm_k (phi_{k-1} Pm_{k-1}) and right_k = left_{k+1}^T go through two LDS buffers of N nx^2 floats and straight into the
// threads' P^-1 rows -- they never touch HBM (write_p != 0 stores them for the stage tests).
// FUSE (with FOLD, RPT = nx / 4): the block rows of S, the diagonal of P^-1 and gamma are FORMED here by the cooperative Schur code
// (schur_coop_rows / schur_coop_pinv: thread t of the PCG layout is lane t % 4 of the group of knot t / 4 - 1) instead of being
// read back: S and P^-1 never exist in global memory, one launch and ~150 MB of traffic per iteration less.  Block row 0 (the
// Q_0 rows) comes from the assembly kernel's cost task.
// ---- helpers of the PAIR form of pcgc_kernel: a row group's 3 nx columns split over two lanes ----------------------------------
// A pair is lanes (t, t ^ 4): the two quads of an 8-lane group run the same 4-lane Schur group through the prologue.
// the partner lane's value: row_shl:4 into banks 0 and 2, row_shr:4 into banks 1 and 3
GATO_DEV float pair_partner(float p)
{
    const float a = dpp_mov<0x104, 0xf, 0x5>((float)0, p);
    return dpp_mov<0x114, 0xf, 0xA>(a, p);
}
// rows x (the lane's half window, in registers), joined with the partner's half in rows_dot's association
template<int HC, int RPT> GATO_DEV void rows_dot_half(const float (*rows)[HC], const float* w, float* acc)
{
    static_assert(HC % 2 == 0, "whole pairs");
    f32x2 a2[RPT];
#pragma unroll
    for (int u = 0; u < RPT; u++) a2[u] = f32x2{0.f, 0.f};
#pragma unroll
    for (int c = 0; c < HC / 2; c++) {
        const f32x2 v = {w[2 * c], w[2 * c + 1]};
#pragma unroll
        for (int u = 0; u < RPT; u++) a2[u] = __builtin_elementwise_fma(f32x2{rows[u][2 * c], rows[u][2 * c + 1]}, v, a2[u]);
    }
#pragma unroll
    for (int u = 0; u < RPT; u++) {
        const f32x2 t = a2[u] + f32x2{pair_partner(a2[u].x), pair_partner(a2[u].y)};
        acc[u] = t.x + t.y;
    }
}
template<int HC> GATO_DEV void load_half_window(float* w, const float* __restrict__ win)   // 8-byte aligned LDS window
{
#pragma unroll
    for (int c = 0; c < HC / 2; c++) {
        const real2 v = reinterpret_cast<const real2*>(win)[c];
        w[2 * c] = v.x;
        w[2 * c + 1] = v.y;
    }
}
// sum over the wavefront of a value that both lanes of every pair hold, each pair counted once: wave_sum without the butterfly that
// would add a lane to its partner (row_half_mirror adds the two quads of an 8-lane group) -- the same tree over the same row groups
GATO_DEV float wave_sum_pairs(float v)
{
#define GATO_DPP_ADD(ctrl) v += dpp_get<ctrl>(v)
    GATO_DPP_ADD(0xB1);   // quad_perm [1,0,3,2]
    GATO_DPP_ADD(0x4E);   // quad_perm [2,3,0,1]
    GATO_DPP_ADD(0x140);  // row_mirror
#undef GATO_DPP_ADD
    const float r0 = lane_read(v, 0);
    const float r1 = lane_read(v, 16);
    const float r2 = lane_read(v, 32);
    const float r3 = lane_read(v, 48);
    return (r3 + r2) + (r1 + r0);
}
GATO_DEV float read_parts(const float* part)   // <= 4 wavefront partials
{
    const real4 a = reinterpret_cast<const real4*>(part)[0];
    return (a.x + a.y) + (a.z + a.w);
}

// acc[u][x] += a[u] * row[x] for the thread's RPT rows, as packed FMAs over column pairs (each element still sees exactly one fma per
// term, in the same order: the bits of the scalar loop at half the instructions)
template<int RPT, int NXT> GATO_DEV void rows_axpy(float (*acc)[NXT], const float* a, const float* __restrict__ row)
{
    if constexpr (NXT % 2 == 0) {
#pragma unroll
        for (int x = 0; x < NXT; x += 2) {
            const f32x2 r2 = {row[x], row[x + 1]};
#pragma unroll
            for (int u = 0; u < RPT; u++) {
                const f32x2 t = __builtin_elementwise_fma(f32x2{a[u], a[u]}, r2, f32x2{acc[u][x], acc[u][x + 1]});
                acc[u][x] = t.x;
                acc[u][x + 1] = t.y;
            }
        }
    } else {
#pragma unroll
        for (int u = 0; u < RPT; u++)
#pragma unroll
            for (int x = 0; x < NXT; x++) acc[u][x] += a[u] * row[x];
    }
}

// FULL: every thread owns a row group (threads x RPT == N nx: N a multiple of 16 in the fused forms) -- the masks of idle lanes and the
// exec-mask juggling around the LDS stores drop out of the iteration (14 of ~305 instructions)
// The body of pcgc_kernel as a device function: `b` the trajectory, tx / TT the thread's index in / the size of the group of wavefronts that
// works on it, `lds` that group's own LDS.  The launched kernel passes its workgroup.  (The lock-step form for two trajectories per workgroup
// that the persistent loop of round 4 used is kept as tools/microbench/sqp_pair.patch.)
template<class M, int RPT, int MAXT, bool FOLD, bool FUSE, bool PAIR, bool FULL>
GATO_DEV void pcgc_body(const Buffers& bf, int N, int B, int b, unsigned tx, unsigned TT, float* lds, uint32_t max_iters, int sqp_iter, int write_p, float dt)
{
    constexpr int NQ = M::NQ, NX = 2 * NQ, BR = 3 * NX, BROW = 3 * NX * NX;
    constexpr int PARTS = (FUSE || MAXT <= 256) ? 1 : (MAXT <= 512 ? 2 : 4);  // FUSE is only launched with <= 256 threads
    constexpr bool TWO = MAXT == 128 && !PAIR;   // launched with exactly 128 threads (solver.hip: launch_pcg_fused at N = 32): block_sum's two-wavefront form
    static_assert(!PAIR || (FUSE && FOLD), "the pair form exists for the fused kernel");
    constexpr int LA2 = (NX % 4 == 0) ? 4 : 2;  // alignment (floats) of an nx-float row in the LDS buffers
    static_assert(NX % RPT == 0, "rows of one thread must share a block row");
    // block stride of the two LDS vectors: nx, or the next multiple of 4 where nx % 4 != 0 (16-byte aligned windows, rows_dot_strided);
    constexpr int VS = (NX % 4 == 0 || !GATO_PCG_VSTRIDE || PAIR) ? NX : pcg_vec_stride(NX);
    const int nrows = N * NX, vecp = (N + 2) * NX, vecl = (N + 2) * VS;
    float* va = lds;
    float* vb = lds + vecl;
    float* partA = vb + ((vecl + 3) & ~3);
    float* partB = partA + 16;
    if (tx < 32) partA[tx] = 0.f;
    const float abs_tol = 1e-6f;
    uint32_t iters = 0;
    const bool skip = bf.converged[b] != 0;  // pcg.cuh:29-32

    if (!skip) {
        const float eps = bf.pcg_tol[b];
        // PAIR: lanes t and t ^ 4 (the two quads of an 8-lane group) run the SAME row group through the prologue (tid is the lane's
        // index in the single-lane form) and then split its 3 nx columns for the iteration
        const int tid = PAIR ? (int)(((tx >> 3) << 2) | (tx & 3)) : (int)tx;
        const int r0 = tid * RPT;
        const bool have = FULL || r0 < nrows;  // threads * RPT >= nrows; whole threads are in or out
        const int rr = have ? r0 : 0;
        const int kb = rr / NX;
        const float* gam = bf.gamma + (size_t)b * vecp;
        float* lam = bf.lambda + (size_t)b * vecp;
        float Srow[RPT][BR], Prow[RPT][BR], xv[RPT], rv[RPT], pv[RPT], zv[RPT], gv[RPT];
        if constexpr (FUSE) {
            static_assert(FOLD && RPT * 4 == NX, "the PCG row layout must coincide with the 4-lane Schur groups");
            const int l = tx & 3;
            {
                // every thread runs the group code (block row 0's and idle threads on knot 0: a valid, discarded computation) -- no
                // branch, so no merged live ranges for the register allocator to spill across the iteration loop
                float phi[RPT][NX], th[RPT][NX], gg[RPT];
                schur_coop_rows<M, 4>(bf, N, b, (have && kb >= 1) ? kb - 1 : 0, l, dt, phi, th, gg);
#pragma unroll
                for (int u = 0; u < RPT; u++) {
#pragma unroll
                    for (int x = 0; x < NX; x++) {
                        Srow[u][x] = phi[u][x];
                        Srow[u][NX + x] = -th[u][x];
                    }
                    gv[u] = gg[u];
                }
                if (opaque_true()) {
                    schur_coop_pinv<M, 4>(th, l, bf.rho[b]);
#pragma unroll
                    for (int u = 0; u < RPT; u++)
#pragma unroll
                        for (int x = 0; x < NX; x++) Prow[u][NX + x] = -th[u][x];
                }
            }
            if (opaque_true()) {
                // block row 0 (the Q_0 rows, written by the assembly kernel's cost task): rows 3l.. are read by every thread (one
                // cached request per row) and kept by the first group only
                constexpr int BLK = NX * NX;
                const float* S = bf.S + (size_t)b * N * BROW + BLK + (size_t)(RPT * l) * NX;      // main block of block row 0
                const float* P = bf.Pinv + (size_t)b * N * BROW + BLK + (size_t)(RPT * l) * NX;
                const bool first = have && kb == 0;
                float g0[RPT];
                load_vec<RPT, RPT>(g0, gam + NX + RPT * l);
#pragma unroll
                for (int u = 0; u < RPT; u++) {
                    float sm[NX], pm[NX];
                    load_vec<NX, NX>(sm, S + u * NX);
                    load_vec<NX, NX>(pm, P + u * NX);
#pragma unroll
                    for (int x = 0; x < NX; x++) {
                        Srow[u][x] = first ? 0.f : Srow[u][x];
                        Srow[u][NX + x] = first ? sm[x] : Srow[u][NX + x];
                        Prow[u][NX + x] = first ? pm[x] : Prow[u][NX + x];
                    }
                    gv[u] = first ? g0[u] : gv[u];
                }
            }
            load_vec<RPT, RPT>(xv, lam + NX + rr);
            // right blocks = the next block row's left block transposed, through LDS
            float* bufA = partB + 16;
            const int i0 = rr - kb * NX;
            if (have) {
#pragma unroll
                for (int u = 0; u < RPT; u++) store_vec<NX, LA2>(bufA + (kb * NX + i0 + u) * NX, &Srow[u][0]);
            }
            __syncthreads();
#pragma unroll
            for (int u = 0; u < RPT; u++)
#pragma unroll
                for (int x = 0; x < NX; x++) Srow[u][2 * NX + x] = (have && kb + 1 < N) ? bufA[((kb + 1) * NX + x) * NX + i0 + u] : 0.f;
            __syncthreads();  // bufA is reused by the stair fold below
        } else {
            const float* S = bf.S + (size_t)b * N * BROW;
            const float* P = bf.Pinv + (size_t)b * N * BROW;
            const int y0 = rr - kb * NX;   // the thread's first row inside block row kb
#pragma unroll
            for (int u = 0; u < RPT; u++) {
                load_btd_row<NX>(Srow[u], S, kb, y0 + u);
                if constexpr (FOLD) load_vec<NX, NX>(&Prow[u][NX], P + (size_t)kb * BROW + NX * NX + (size_t)(y0 + u) * NX);  // off-diagonals are formed below
                else load_btd_row<NX>(Prow[u], P, kb, y0 + u);
            }
            load_vec<RPT, RPT>(xv, lam + NX + rr);
            load_vec<RPT, RPT>(gv, gam + NX + rr);
        }
        if constexpr (FOLD) {
            float* bufA = partB + 16;              // [N][NX][NX]: Pm blocks, later the left blocks
            float* bufB = bufA + N * NX * NX;      // [N][NX][NX]: phi_{k-1} Pm_{k-1}
            const int i0 = rr - kb * NX;
            if (have) {
#pragma unroll
                for (int u = 0; u < RPT; u++) store_vec<NX, LA2>(bufA + (kb * NX + i0 + u) * NX, &Prow[u][NX]);
            }
            __syncthreads();
            constexpr int LA = (NX % 4 == 0) ? 4 : 2;  // LDS rows of nx floats: 16-byte aligned when nx % 4 == 0
            if (have && kb >= 1) {
                // every operand row is read from LDS once and feeds the thread's RPT rows (the sums run over jj in the same order)
                const float* Pm1 = bufA + (kb - 1) * NX * NX;
                float scr[RPT][NX];
#pragma unroll
                for (int u = 0; u < RPT; u++)
#pragma unroll
                    for (int x = 0; x < NX; x++) scr[u][x] = 0.f;
#pragma unroll
                for (int jj = 0; jj < NX; jj++) {
                    float prow_[NX];
                    load_vec<NX, LA>(prow_, Pm1 + jj * NX);
                    float a_[RPT];
#pragma unroll
                    for (int u = 0; u < RPT; u++) a_[u] = Srow[u][jj];
                    rows_axpy<RPT, NX>(scr, a_, prow_);
                }
#pragma unroll
                for (int u = 0; u < RPT; u++) store_vec<NX, LA>(bufB + (kb * NX + i0 + u) * NX, scr[u]);
            }
            __syncthreads();
            if (have) {
                const float* sc = bufB + kb * NX * NX;
                float res[RPT][NX];
#pragma unroll
                for (int u = 0; u < RPT; u++)
#pragma unroll
                    for (int x = 0; x < NX; x++) res[u][x] = 0.f;
                if (kb >= 1) {
#pragma unroll
                    for (int jj = 0; jj < NX; jj++) {
                        float srow_[NX];
                        load_vec<NX, LA>(srow_, sc + jj * NX);
                        float a_[RPT];
#pragma unroll
                        for (int u = 0; u < RPT; u++) a_[u] = Prow[u][NX + jj];
                        rows_axpy<RPT, NX>(res, a_, srow_);
                    }
#pragma unroll
                    for (int u = 0; u < RPT; u++)
#pragma unroll
                        for (int x = 0; x < NX; x++) res[u][x] = -res[u][x];
                }
#pragma unroll
                for (int u = 0; u < RPT; u++) {
#pragma unroll
                    for (int x = 0; x < NX; x++) Prow[u][x] = res[u][x];
                    store_vec<NX, LA>(bufA + (kb * NX + i0 + u) * NX, res[u]);
                }
            }
            __syncthreads();
            if (have) {
#pragma unroll
                for (int u = 0; u < RPT; u++) {
#pragma unroll
                    for (int x = 0; x < NX; x++)
                        Prow[u][2 * NX + x] = (kb + 1 < N) ? bufA[((kb + 1) * NX + x) * NX + i0 + u] : 0.f;  // right = left_{k+1}^T
                }
                if (write_p) {
                    float* Pg = bf.Pinv + (size_t)b * N * BROW + (size_t)kb * BROW + (size_t)i0 * NX;
#pragma unroll
                    for (int u = 0; u < RPT; u++) {
                        store_vec<NX, 2>(Pg + u * NX, &Prow[u][0]);                      // left block
                        store_vec<NX, 2>(Pg + 2 * NX * NX + u * NX, &Prow[u][2 * NX]);   // right block
                    }
                }
            }
        }
        // PARK: the right blocks of the P^-1 rows move to the LDS the fold no longer needs
        constexpr bool PARK = FUSE && FOLD && (NX % 4 == 0) && !PAIR;
        const real4* park = nullptr;
        if constexpr (PARK) {
            real4* pk = reinterpret_cast<real4*>(partB + 16) + tx;
            __syncthreads();  // the fold's last readers of bufA are done
#pragma unroll
            for (int u = 0; u < RPT; u++)
#pragma unroll
                for (int c = 0; c < NX / 4; c++)
                    pk[(u * (NX / 4) + c) * TT] = make_real4(Prow[u][2 * NX + 4 * c], Prow[u][2 * NX + 4 * c + 1], Prow[u][2 * NX + 4 * c + 2],
                                                                      Prow[u][2 * NX + 4 * c + 3]);
            park = pk;
        }
        if constexpr (PAIR) {
            // The launch lasts as long as its slowest trajectory iterates (duration = a + 0.97 us x max iterations at C2, crowded or not,
            // tools/exp_pcg_rate.py), and an iteration of the single-lane form is a dependent chain of 108 packed FMAs and 29 LDS reads
            // issued in register-starved batches.  Here lane h of a pair keeps columns [HC h, HC h + HC) of [left | main | right] of
            // its S and P^-1 rows: per product 27 packed FMAs behind 9 ds_read_b64 that are all in flight at once; two DPP moves
            // fetch the partner's half sum, and the wavefront sum leaves out the butterfly that would add a lane to its partner.
            constexpr int HC = BR / 2;
            static_assert(BR % 4 == 0, "halves are whole real2 pairs");
            const int h = (tx >> 2) & 1;
            const bool owner = have && h == 0;   // the lane of the pair that publishes the pair's vector entries
            float Sh[RPT][HC], Ph[RPT][HC];
#pragma unroll
            for (int u = 0; u < RPT; u++)
#pragma unroll
                for (int c = 0; c < HC; c++) {
                    Sh[u][c] = h ? Srow[u][HC + c] : Srow[u][c];
                    Ph[u][c] = h ? Prow[u][HC + c] : Prow[u][c];
                }
            const float* wa = va + kb * NX + HC * h;
            const float* wb = vb + kb * NX + HC * h;
            float* oa = va + NX + rr;
            float* ob = vb + NX + rr;
            const int wv = tx >> 6;
            for (int i = tx; i < NX; i += TT) {
                va[i] = 0.f; vb[i] = 0.f;
                va[vecp - NX + i] = 0.f; vb[vecp - NX + i] = 0.f;
            }
            if (owner) store_vec<RPT, RPT>(oa, xv);
            __syncthreads();
            float w[HC], acc[RPT];
            load_half_window<HC>(w, wa);
            rows_dot_half<HC, RPT>(Sh, w, acc);  // r = gamma - S x
#pragma unroll
            for (int u = 0; u < RPT; u++) rv[u] = have ? gv[u] - acc[u] : 0.f;
            if (owner) store_vec<RPT, RPT>(ob, rv);
            __syncthreads();
            load_half_window<HC>(w, wb);
            rows_dot_half<HC, RPT>(Ph, w, acc);  // z = p = P^-1 r
            float loc = 0.f;
#pragma unroll
            for (int u = 0; u < RPT; u++) {
                zv[u] = have ? acc[u] : 0.f;
                pv[u] = zv[u];
                loc += rv[u] * zv[u];
            }
            loc = wave_sum_pairs(loc);
            partA[wv] = loc;   // every lane, the same value: see block_sum
            __syncthreads();
            float rho = read_parts(partA);
            if (!(fabsf(rho) < abs_tol)) {
                const float rho_init = fabsf(rho);
                if (max_iters > 0) for (;;) {
                    iters++;
                    if (owner) store_vec<RPT, RPT>(oa, pv);
                    __syncthreads();
                    load_half_window<HC>(w, wa);
                    rows_dot_half<HC, RPT>(Sh, w, acc);  // A p
                    loc = 0.f;
#pragma unroll
                    for (int u = 0; u < RPT; u++) {
                        if (!have) acc[u] = 0.f;
                        loc += pv[u] * acc[u];
                    }
                    loc = wave_sum_pairs(loc);
                    partB[wv] = loc;
                    __syncthreads();
                    const float pAp = read_parts(partB);
                    const float alpha = pcg_div(rho, pAp);
#pragma unroll
                    for (int u = 0; u < RPT; u++) {
                        xv[u] += alpha * pv[u];
                        rv[u] -= alpha * acc[u];
                    }
                    if (owner) store_vec<RPT, RPT>(ob, rv);
                    __syncthreads();
                    load_half_window<HC>(w, wb);
                    rows_dot_half<HC, RPT>(Ph, w, acc);  // z = P^-1 r
                    loc = 0.f;
#pragma unroll
                    for (int u = 0; u < RPT; u++) {
                        zv[u] = have ? acc[u] : 0.f;
                        loc += rv[u] * zv[u];
                    }
                    loc = wave_sum_pairs(loc);
                    partA[wv] = loc;   // every lane, the same value: see block_sum
                    __syncthreads();
                    const float rho_new = read_parts(partA);
                    GATO_PCG_TAIL(RPT, pv, zv, rho, rho_new, abs_tol + eps * rho_init, iters, max_iters)
                }
                if (owner) store_vec<RPT, RPT>(lam + NX + rr, xv);
            }
        } else {
        const float* wa = va + kb * VS;
        const float* wb = vb + kb * VS;
        float* oa = va + (kb + 1) * VS + (rr - kb * NX);
        float* ob = vb + (kb + 1) * VS + (rr - kb * NX);
        for (int i = tx; i < VS; i += TT) {
            va[i] = 0.f; vb[i] = 0.f;
            va[vecl - VS + i] = 0.f; vb[vecl - VS + i] = 0.f;
        }
        // S x, S p and P^-1 r of the thread's rows against the window [block kb-1 | kb | kb+1] of an LDS vector
        auto sdot = [&](const float* win, float* out) {
            if constexpr (VS != NX) rows_dot_strided<NX, RPT, VS>(Srow, win, out);
            else rows_dot<NX, RPT>(Srow, win, out);
        };
        auto pdot = [&](const float* win, float* out) {
            if constexpr (PARK) rows_dot_parked<NX, RPT>(Prow, win, park, TT, out);
            else if constexpr (VS != NX) rows_dot_strided<NX, RPT, VS>(Prow, win, out);
            else rows_dot<NX, RPT>(Prow, win, out);
        };
        if (have) store_vec<RPT, RPT>(oa, xv);
        __syncthreads();
        float acc[RPT];
        sdot(wa, acc);  // r = gamma - S x
#pragma unroll
        for (int u = 0; u < RPT; u++) rv[u] = have ? gv[u] - acc[u] : 0.f;
        if (have) store_vec<RPT, RPT>(ob, rv);
        __syncthreads();
        pdot(wb, acc);  // z = p = P^-1 r
        float loc = 0.f;
#pragma unroll
        for (int u = 0; u < RPT; u++) {
            zv[u] = have ? acc[u] : 0.f;
            pv[u] = zv[u];
            loc += rv[u] * zv[u];
        }
        float rho = block_sum<PARTS, TWO>(loc, partA, tx);
        const bool entered = !(fabsf(rho) < abs_tol);
        if (entered) {
            const float rho_init = fabsf(rho);
            if (max_iters > 0) for (;;) {
                iters++;
                if (have) store_vec<RPT, RPT>(oa, pv);
                __syncthreads();
                sdot(wa, acc);  // A p
                loc = 0.f;
#pragma unroll
                for (int u = 0; u < RPT; u++) {
                    if (!have) acc[u] = 0.f;
                    loc += pv[u] * acc[u];
                }
                const float pAp = block_sum<PARTS, TWO>(loc, partB, tx);
                const float alpha = pcg_div(rho, pAp);
#pragma unroll
                for (int u = 0; u < RPT; u++) rv[u] -= alpha * acc[u];
                if (have) store_vec<RPT, RPT>(ob, rv);
                __syncthreads();
                pdot(wb, acc);  // z = P^-1 r
                loc = 0.f;
#pragma unroll
                for (int u = 0; u < RPT; u++) {
                    zv[u] = have ? acc[u] : 0.f;
                    loc += rv[u] * zv[u];
                }
                // the x update rides in the shadow of the partial sum's store (it needs alpha and the OLD p: both still live here)
                auto xup = [&]() {
                    float al = alpha;
                    asm volatile("" : "+v"(al));
#pragma unroll
                    for (int u = 0; u < RPT; u++) { xv[u] += al * pv[u]; asm volatile("" : "+v"(xv[u])); }
                };
                const float rho_new = block_sum<PARTS, TWO>(loc, partA, tx, xup);
                GATO_PCG_TAIL(RPT, pv, zv, rho, rho_new, abs_tol + eps * rho_init, iters, max_iters)
            }
            if (have) store_vec<RPT, RPT>(lam + NX + rr, xv);
        }
        }
    }
    if (tx == 0) {
        bf.pcg_iters[b] = iters;
        bf.st_pcg_iters[(size_t)sqp_iter * B + b] = (int32_t)iters;
        int conv = skip ? 1 : 0;
        if (iters == 0) { conv = 1; bf.converged[b] = 1; }  // bsqp.cuh:153-156 (kkt_tol is unused there)
        if (conv) atomicAdd(&bf.num_solved_w[sqp_iter], 1u);
    }
}

template<class M, int RPT, int MAXT, bool FOLD, bool FUSE = false, bool PAIR = false, bool FULL = false>
__global__ __launch_bounds__(MAXT) void pcgc_kernel(Buffers bf, int N, int B, uint32_t max_iters, int sqp_iter, int write_p, float dt)
{
    extern __shared__ __attribute__((aligned(16))) float lds[];
    if (bf.ctrl->done) return;
    const int b = bf.order[blockIdx.x];   // hardest-first where the launch runs in rounds (solver.hip: pcg_rounds), else the identity
    pcgc_body<M, RPT, MAXT, FOLD, FUSE, PAIR, FULL>(bf, N, B, b, threadIdx.x, blockDim.x, lds, max_iters, sqp_iter, write_p, dt);
}

// ---- PCG, register-resident in SYMMETRIC HALF STORAGE (long horizons: iiwa14 N = 128 is 602 KB of S and P^-1 per trajectory) ------
// S and P^-1 are symmetric block-tridiagonal with right_k = left_{k+1}^T EXACTLY (schur1 stores phi^T by transposition,
// schur_linsys.cuh:130-146; schur2 writes -res and -res^T from one `res`, :243-258), so the right blocks need no storage: a row's
// product is  left_k v_{k-1} + main_k v_k + (left_{k+1}^T v_{k+1}),  and the last term is formed by the threads that hold left_{k+1}
// as a TRANSPOSED accumulate.  A trajectory then needs 2 x N nx x 2 nx floats = 401 KB (iiwa14 N = 128): one CU's register file holds
// it (512 KB), one workgroup per trajectory, B = 256 = one trajectory per CU -- nothing is re-read from L2 / MALL during the
// iteration (the streaming kernel moved 602 KB per PCG iteration and trajectory and was bound by exactly that: 2.5 ms per launch).
//
// Workgroup = 4 N threads in two ROLES, split by wavefront (role = t >= 2N), u = t mod 2N, block row k = u / 2, row half h = u % 2:
//   left role : rows k nx + h nx/2 .. of left_k (S and P^-1): the row dot with v_{k-1} AND the transposed accumulate left_k^T v_k;
//   main role : the same rows of main_k: the row dot with v_k; owns the vector entries x, r, p of its rows and does the PCG scalar work.
// Per matrix-vector product: vector -> LDS | barrier | both roles work, the left role leaves its row partials and transposed partials in
// LDS, every wavefront its share of v^T Mx v | barrier | the main role adds  main + left + (two halves of the transposed partial of
// block k+1), everybody sums the wavefront shares: 4 barriers per PCG iteration.  Four rows of every thread's
// P^-1 block live in LDS (real4 [nx][T], conflict-free) to stay under 256 registers.  Sums associate as (main + left) + t0 + t1; the
// reference sums a row's 3 nx terms in sequence (linalg.cuh:174-260) -- the same fp32 freedom the other PCG kernels take (even / odd
// pairs).  Needs the COMPLETE P^-1 in global memory (schur2_kernel) and N >= 16 (whole wavefronts).
// One thread's HR x NX block times the window w (row dots) and -- TR -- its transpose times the
// thread's own HR vector entries vo (tp[j] = sum_i Mt[i][j] vo[i]).  The first NP rows of the block are parked in LDS as real4 chunks
// [c][T] and streamed through four registers at a time; rows NP.. sit in Mt[0 .. HR-NP).
template<int NX, int HR, int NP, bool TR>
GATO_DEV void half_block(const float (*Mt)[NX], const real4* park, int T, const float* w, const float* vo, float* acc, float* tp)
{
    // packed FP32 (v_pk_fma_f32): row dots keep an (even, odd) pair of partial sums per row, the transposed accumulate advances two
    // adjacent columns with the row's vector entry in both halves
    f32x2 a2[HR], t2[TR ? NX / 2 : 1];
#pragma unroll
    for (int i = 0; i < HR; i++) a2[i] = f32x2{0.f, 0.f};
    if constexpr (TR) {
#pragma unroll
        for (int j = 0; j < NX / 2; j++) t2[j] = f32x2{0.f, 0.f};
    }
    if constexpr (NP > 0) {
#pragma unroll
        for (int c = 0; c < (NP * NX) / 4; c++) {
            const real4 m4 = park[c * T];
            const f32x2 m[2] = {f32x2{m4.x, m4.y}, f32x2{m4.z, m4.w}};
#pragma unroll
            for (int q = 0; q < 2; q++) {
                const int idx = 4 * c + 2 * q, i = idx / NX, j = idx % NX;   // NX even: a pair never straddles two rows
                a2[i] = __builtin_elementwise_fma(m[q], f32x2{w[j], w[j + 1]}, a2[i]);
                if constexpr (TR) t2[j / 2] = __builtin_elementwise_fma(m[q], f32x2{vo[i], vo[i]}, t2[j / 2]);
            }
        }
    }
#pragma unroll
    for (int i = NP; i < HR; i++) {
#pragma unroll
        for (int j = 0; j < NX; j += 2) {
            const f32x2 m = {Mt[i - NP][j], Mt[i - NP][j + 1]};
            a2[i] = __builtin_elementwise_fma(m, f32x2{w[j], w[j + 1]}, a2[i]);
            if constexpr (TR) t2[j / 2] = __builtin_elementwise_fma(m, f32x2{vo[i], vo[i]}, t2[j / 2]);
        }
    }
#pragma unroll
    for (int i = 0; i < HR; i++) acc[i] = a2[i].x + a2[i].y;
    if constexpr (TR) {
#pragma unroll
        for (int j = 0; j < NX / 2; j++) { tp[2 * j] = t2[j].x; tp[2 * j + 1] = t2[j].y; }
    }
}

// FOLD: the stair off-diagonals of P^-1 (formSchurSystemBatchedKernel2, schur_linsys.cuh:213-260: left_k = -Pm_k (phi_{k-1} Pm_{k-1}) from the
// STORED diagonal blocks Pm) are formed HERE instead of by schur2_kernel: the main role publishes its rows of Pm in LDS, the left role
// -- which holds the rows of phi_{k-1} as its S block -- forms its rows of phi_{k-1} Pm_{k-1}, the two left threads of a block row
// exchange them through LDS and each forms its rows of left_k.  Two passes over the block rows (the not-yet-used parking area holds
// N/2 + 1 blocks of Pm and N/2 blocks of the intermediate product); the sums run in schur2_kernel's order.
template<class M, int MAXT, bool FOLD>
__global__ __launch_bounds__(MAXT, 2) void pcgs_kernel(Buffers bf, int N, int B, uint32_t max_iters, int sqp_iter)
{
    constexpr int NQ = M::NQ, NX = 2 * NQ, HR = NX / 2, BR = 3 * NX, BROW = 3 * NX * NX;
    constexpr int NP = 4;                      // rows of the thread's P^-1 block parked in LDS
    constexpr int PF4 = (NP * NX) / 4;         // as real4 chunks
    static_assert(NX % 2 == 0 && (NP * NX) % 4 == 0, "layout");
    constexpr int PARTS = MAXT <= 256 ? 1 : 2;
    extern __shared__ __attribute__((aligned(16))) float lds[];
    if (bf.ctrl->done) return;
    const int b = blockIdx.x, t = threadIdx.x, T = blockDim.x;  // T = 4 N, a multiple of 64
    // the two LDS vectors keep their blocks at a stride of VS floats = nx rounded up to a multiple of 4 (pcgc_body: the same layout): a thread's
    // window is one 16-byte aligned block, four ds_read_b128 instead of seven 8-byte reads
    constexpr int VS = (NX % 4 == 0 || !GATO_PCGS_VSTRIDE) ? NX : ((NX + 3) & ~3);
    const int vecp = (N + 2) * NX, vecl = (N + 2) * VS;
    float* va = lds;
    float* vb = va + vecl;
    float* partA = vb + ((vecl + 3) & ~3);
    float* partB = partA + 16;
    float* rowbuf = partB + 16;                // [N nx]: the left-block part of every row's product
    float* tbuf = rowbuf + N * NX;             // [N + 1][2][nx]: transposed partials of block k, by row half; block N stays zero
    real4* park = reinterpret_cast<real4*>(tbuf + (N + 1) * 2 * NX) + t;  // [PF4][T]
    if (t < 32) partA[t] = 0.f;  // partA and partB are adjacent: slots of wavefronts the workgroup does not have stay zero
    const float abs_tol = 1e-6f;
    uint32_t iters = 0;
    const bool skip = bf.converged[b] != 0;    // pcg.cuh:29-32

    if (!skip) {
        const float eps = bf.pcg_tol[b];
        const bool mainrole = t >= (T >> 1);
        // a SIMD hosts one left-role and one main-role wavefront and the left one carries twice the multiply-adds of every product: served first,
        // it is not held to every second issue slot while the main one still runs (launch 403.5 -> 397 us at C3; priorities 2 and 3 alike)
        if (!mainrole) __builtin_amdgcn_s_setprio(2);
        const int u = mainrole ? t - (T >> 1) : t;
        const int k = u >> 1, h = u & 1;
        const int r0 = k * NX + h * HR;
        const float* gam = bf.gamma + (size_t)b * vecp;
        float* lam = bf.lambda + (size_t)b * vecp;
        float Sm[HR][NX], Pm[HR - NP][NX];
        {
            // rows h HR .. of the left or the main block of block row k: HR nx contiguous floats in the block-major storage
            const size_t boff = (size_t)b * N * BROW + (size_t)k * BROW + (mainrole ? NX * NX : 0) + (size_t)h * HR * NX;
            const float* Sg = bf.S + boff;
            const float* Pg = bf.Pinv + boff;
            float Pf[HR][NX];   // the thread's block of P^-1 before its first NP rows are parked
#pragma unroll
            for (int i = 0; i < HR; i++) {
                load_vec<NX, 2>(Sm[i], Sg + i * NX);
                if constexpr (FOLD) {
                    if (mainrole) load_vec<NX, 2>(Pf[i], Pg + i * NX);
                    else {
#pragma unroll
                        for (int c = 0; c < NX; c++) Pf[i][c] = 0.f;
                    }
                } else {
                    load_vec<NX, 2>(Pf[i], Pg + i * NX);
                }
            }
            if constexpr (FOLD) {
                const int NB = N >> 1;                                  // block rows per pass
                float* bufP = reinterpret_cast<float*>(park - t);       // [NB + 1][NX][NX]: Pm of block rows first-1 .. first+NB-1
                float* bufX = bufP + (NB + 1) * NX * NX;                // [NB][NX][NX]: phi_{k-1} Pm_{k-1} of block rows first .. first+NB-1
                for (int ps = 0; ps < 2; ps++) {
                    const int first = ps * NB;
                    const int slot = k - (first - 1);                   // this block row's slot in bufP
                    if (mainrole && slot >= 0 && slot <= NB) {
#pragma unroll
                        for (int i = 0; i < HR; i++) store_vec<NX, 2>(bufP + (slot * NX + h * HR + i) * NX, Pf[i]);
                    }
                    __syncthreads();
                    const bool mine = !mainrole && k >= first && k < first + NB && k >= 1;
                    if (mine) {
                        const float* Pk1 = bufP + (slot - 1) * NX * NX;  // Pm_{k-1}
#pragma unroll
                        for (int i = 0; i < HR; i++) if (opaque_true()) {   // a basic block per row: the scheduler must not hoist every row's LDS loads
                            float x[NX];
#pragma unroll
                            for (int c = 0; c < NX; c++) x[c] = 0.f;
#pragma unroll
                            for (int j = 0; j < NX; j++) {
                                float row[NX];
                                load_vec<NX, 2>(row, Pk1 + j * NX);
#pragma unroll
                                for (int c = 0; c < NX; c++) x[c] += Sm[i][j] * row[c];
                            }
                            store_vec<NX, 2>(bufX + ((k - first) * NX + h * HR + i) * NX, x);
                        }
                    }
                    __syncthreads();
                    if (mine) {
                        const float* Xk = bufX + (k - first) * NX * NX;
#pragma unroll
                        for (int i = 0; i < HR; i++) if (opaque_true()) {
                            float tk[NX], res[NX];
                            load_vec<NX, 2>(tk, bufP + (slot * NX + h * HR + i) * NX);   // row of Pm_k
#pragma unroll
                            for (int c = 0; c < NX; c++) res[c] = 0.f;
#pragma unroll
                            for (int j = 0; j < NX; j++) {
                                float row[NX];
                                load_vec<NX, 2>(row, Xk + j * NX);
#pragma unroll
                                for (int c = 0; c < NX; c++) res[c] += tk[j] * row[c];
                            }
#pragma unroll
                            for (int c = 0; c < NX; c++) Pf[i][c] = -res[c];
                        }
                    }
                    __syncthreads();   // bufP / bufX are free for the next pass, then for the parked rows
                }
            }
#pragma unroll
            for (int c = 0; c < PF4; c++)
                park[c * T] = make_real4((&Pf[0][0])[4 * c], (&Pf[0][0])[4 * c + 1], (&Pf[0][0])[4 * c + 2], (&Pf[0][0])[4 * c + 3]);
#pragma unroll
            for (int i = NP; i < HR; i++)
#pragma unroll
                for (int c = 0; c < NX; c++) Pm[i - NP][c] = Pf[i][c];
        }
        float xv[HR], rv[HR], pv[HR], gv[HR];
#pragma unroll
        for (int i = 0; i < HR; i++) {
            xv[i] = mainrole ? lam[NX + r0 + i] : 0.f;
            gv[i] = mainrole ? gam[NX + r0 + i] : 0.f;
        }
        for (int i = t; i < VS; i += T) {
            va[i] = 0.f; vb[i] = 0.f;
            va[vecl - VS + i] = 0.f; vb[vecl - VS + i] = 0.f;
        }
        for (int i = t; i < 2 * NX; i += T) tbuf[N * 2 * NX + i] = 0.f;
        const float* wina = va + (k + (mainrole ? 1 : 0)) * VS;   // the role's window: block k-1 (left) or k (main) of the padded vector
        const float* winb = vb + (k + (mainrole ? 1 : 0)) * VS;
        const int own = (k + 1) * VS + h * HR;                     // the thread's rows in the padded vectors

        // out[i] = row r0 + i of Mx v for the main role; `vec` is the LDS copy of v (published by the caller's barrier); ISP: Mx = P^-1.
        // The roles are whole wavefronts, so each takes ONE branch per product and runs straight-line code inside it.
        // The product's barrier also carries the reduction of v^T Mx v, formed from the stored blocks only:
        //     v^T Mx v = sum_k v_k^T main_k v_k + 2 v_k^T left_k v_{k-1}        (right_k = left_{k+1}^T exactly)
        // -- the main role contributes its rows' v_k (main_k v_k), the left role 2 v_k (left_k v_{k-1}); the wavefront partials are written
        // before the barrier and summed after it (4 barriers per PCG iteration instead of 6).  `vown`: the main role's rows of v.
        auto matvec = [&](const float* vec, const float* win, auto isp, float* out, const float* vown, float* part) -> float {
            constexpr bool ISP = decltype(isp)::value;
            float acc[HR], w[VS];
            if constexpr (VS != NX) load_vec<VS, 4>(w, win);   // (the block's two padding floats ride along, unused)
            else load_vec<NX, 2>(w, win);
            float dotc = 0.f;
            if (!mainrole) {
                float tp[NX], vo[HR];
#pragma unroll
                for (int i = 0; i < HR; i++) vo[i] = vec[own + i];
                if constexpr (ISP) half_block<NX, HR, NP, true>(Pm, park, T, w, vo, acc, tp);
                else half_block<NX, HR, 0, true>(Sm, nullptr, T, w, vo, acc, tp);
                store_vec<NX, 2>(tbuf + (k * 2 + h) * NX, tp);
                store_vec<HR, 1>(rowbuf + r0, acc);
#pragma unroll
                for (int i = 0; i < HR; i++) dotc = __builtin_fmaf(acc[i], vo[i], dotc);
                dotc = 2.0f * dotc;
            } else {
                if constexpr (ISP) half_block<NX, HR, NP, false>(Pm, park, T, w, nullptr, acc, nullptr);
                else half_block<NX, HR, 0, false>(Sm, nullptr, T, w, nullptr, acc, nullptr);
#pragma unroll
                for (int i = 0; i < HR; i++) dotc = __builtin_fmaf(acc[i], vown[i], dotc);
            }
            dotc = wave_sum(dotc);
            part[t >> 6] = dotc;   // every lane, the same value (block_sum: no exec-mask branch in front of the barrier)
            __syncthreads();
            if (mainrole) {
                const float* t0 = tbuf + ((k + 1) * 2 + 0) * NX + h * HR;
                const float* t1 = tbuf + ((k + 1) * 2 + 1) * NX + h * HR;
#pragma unroll
                for (int i = 0; i < HR; i++) out[i] = ((acc[i] + rowbuf[r0 + i]) + t0[i]) + t1[i];
            } else {
#pragma unroll
                for (int i = 0; i < HR; i++) out[i] = 0.f;
            }
            const real4 pa = reinterpret_cast<const real4*>(part)[0];
            float tot = (pa.x + pa.y) + (pa.z + pa.w);
            if constexpr (PARTS == 2) {
                const real4 pb = reinterpret_cast<const real4*>(part)[1];
                tot = tot + ((pb.x + pb.y) + (pb.z + pb.w));
            }
            return tot;
        };
        using yes = std::true_type;
        using no = std::false_type;

        if (mainrole) store_vec<HR, 1>(va + own, xv);
        __syncthreads();
        float acc[HR], zv[HR];
        (void)matvec(va, wina, no{}, acc, xv, partB);  // r = gamma - S x
#pragma unroll
        for (int i = 0; i < HR; i++) rv[i] = gv[i] - acc[i];   // both are 0 in the left role
        if (mainrole) store_vec<HR, 1>(vb + own, rv);
        __syncthreads();
        float rho = matvec(vb, winb, yes{}, zv, rv, partA);  // z = p = P^-1 r, rho = r^T z
#pragma unroll
        for (int i = 0; i < HR; i++) pv[i] = zv[i];
        if (!(fabsf(rho) < abs_tol)) {
            const float rho_init = fabsf(rho);
            if (max_iters > 0) for (;;) {
                iters++;
                if (mainrole) store_vec<HR, 1>(va + own, pv);
                __syncthreads();
                const float pAp = matvec(va, wina, no{}, acc, pv, partB);  // A p and p^T A p
                const float alpha = pcg_div(rho, pAp);
#pragma unroll
                for (int i = 0; i < HR; i++) {
                    xv[i] += alpha * pv[i];
                    rv[i] -= alpha * acc[i];
                }
                if (mainrole) store_vec<HR, 1>(vb + own, rv);
                __syncthreads();
                const float rho_new = matvec(vb, winb, yes{}, zv, rv, partA);  // z = P^-1 r and r^T z
                GATO_PCG_TAIL(HR, pv, zv, rho, rho_new, abs_tol + eps * rho_init, iters, max_iters)
            }
            if (mainrole) {
#pragma unroll
                for (int i = 0; i < HR; i++) lam[NX + r0 + i] = xv[i];
            }
        }
    }
    if (threadIdx.x == 0) {
        bf.pcg_iters[b] = iters;
        bf.st_pcg_iters[(size_t)sqp_iter * B + b] = (int32_t)iters;
        int conv = skip ? 1 : 0;
        if (iters == 0) { conv = 1; bf.converged[b] = 1; }  // bsqp.cuh:153-156 (kkt_tol is unused there)
        if (conv) atomicAdd(&bf.num_solved_w[sqp_iter], 1u);
    }
}

// =========================================================================================================================
// DIRECT solve of the block-tridiagonal Schur system S lambda = gamma (opt-in: gato_set_linear_solver; the north-star's "block-
// tridiagonal Riccati/Schur solve").  The reference only has PCG; this mode replaces pcg.cuh:14-148 by a block
// LU sweep without pivoting (S is symmetric negative definite: -S is the Schur complement of the regularised KKT system):
//     D_0 = main_0,  g_0 = gamma_0;      W_k = left_k D_{k-1}^-1,  D_k = main_k - W_k left_k^T,  g_k = gamma_k - W_k g_{k-1}   (k = 1 .. N-1)
//     lambda_{N-1} = D_{N-1}^-1 g_{N-1};  lambda_k = D_k^-1 (g_k - left_{k+1}^T lambda_{k+1})                                     (k = N-2 .. 0)
// (right_k = left_{k+1}^T exactly, see pcgs_kernel).  The recursion is serial in k -- this is the Riccati depth -- so the lever is
// the length of one step: ONE WAVEFRONT per trajectory, lane = (row r = lane / 4, column group cq = lane % 4) holds CW = ceil(nx / 4)
// entries of row r of the running blocks (nx = 14 is padded to 16 with identity rows).  Operand blocks meet in LDS (stride 17: no bank
// conflicts; the workgroup is one wavefront, so its barriers cost no wait and only order the lanes' LDS accesses); D^-1 by Gauss-Jordan
// without pivoting (the scheme of block::invertMatrix, linalg.cuh:364-519, with v_rcp_f32 pivots) -- the pivot element through v_readlane,
// the pivot row through ds_bpermute and the row multiplier through a DPP quad broadcast; the next block row's operands are fetched
// from global memory while the current one is eliminated.  D_k^-1 is kept for the back substitution in the P^-1 buffer's main blocks
// (the preconditioner is not formed in this mode).  Cost: N serial steps, no iteration count -- the solve time no longer depends on
// the conditioning (rho), and lambda is exact to fp32 rounding instead of PCG's exit tolerance.
// =========================================================================================================================
template<int PC> GATO_DEV float quad_bcast(float v)  // lane PC of every quad
{
    return dpp_get<PC * 0x55>(v);
}
GATO_DEV float quad_sum(float v)
{
    v += dpp_get<0xB1>(v);  // quad_perm [1,0,3,2]
    v += dpp_get<0x4E>(v);  // quad_perm [2,3,0,1]
    return v;
}
// one Gauss-Jordan pivot of the distributed block: D[i] = entry (r, cq CW + i); P is a compile-time constant.  In place: column P of
// the block becomes column P of the inverse (entry (r, P) is consumed as the row multiplier before it is overwritten).  1 / pivot is
// v_rcp_f32 (1 ulp) -- this mode is not the reference's arithmetic, and the serial chain is what it pays for.
template<int NX, int CW, int P> GATO_DEV void gj_quad_step(float* D, int r, int cq, int baddr)
{
    constexpr int PC = P / CW, PI = P % CW;   // column group and local index of column P
    float prow[CW];
#pragma unroll
    for (int i = 0; i < CW; i++)              // row P, this lane's columns: lane 4 P + cq (ds_bpermute takes a byte address)
        prow[i] = lane_permute(baddr + 16 * P, D[i]);
    const float ppiv = lane_read(D[PI], P * 4 + PC);  // entry (P, P)
    const float pvInv = fast_rcp(ppiv);
    const float f = quad_bcast<PC>(D[PI]) * pvInv;                                                                    // entry (r, P) / pivot
    const bool owner = (r == P);
#pragma unroll
    for (int i = 0; i < CW; i++) D[i] = owner ? D[i] * pvInv : __builtin_fmaf(-f, prow[i], D[i]);
    D[PI] = (cq == PC) ? (owner ? pvInv : -f) : D[PI];
    if constexpr (P + 1 < NX) gj_quad_step<NX, CW, P + 1>(D, r, cq, baddr);
}

template<class M>
__global__ __launch_bounds__(64) void btd_direct_kernel(Buffers bf, int N, int B, int sqp_iter)
{
    constexpr int NQ = M::NQ, NX = 2 * NQ, BR = 3 * NX, BROW = 3 * NX * NX;
    constexpr int CW = (NX + 3) / 4, LD = 17;
    __shared__ float sD[16 * LD], sL[16 * LD], sW[16 * LD], sv[32];
    if (bf.ctrl->done) return;
    const int b = blockIdx.x, lane = threadIdx.x;
    const int r = lane >> 2, cq = lane & 3;
    const int c0 = cq * CW;
    const bool rowok = r < NX;
    const float* S = bf.S + (size_t)b * N * BROW;
    float* Dinv = bf.Pinv + (size_t)b * N * BROW;
    const float* gam = bf.gamma + (size_t)b * (N + 2) * NX;
    float* lam = bf.lambda + (size_t)b * (N + 2) * NX;
    const bool skip = bf.converged[b] != 0;

    if (!skip) {
        // entry (r, c0 + i) of a block of S (col offset `off`: 0 left, NX main); the padding of a 14 x 14 block to 16 x 16 is the identity
        auto load_slice = [&](const float* base, int off, float* out, float diag) {
#pragma unroll
            for (int i = 0; i < CW; i++) {
                const int c = c0 + i;
                out[i] = (rowok && c < NX) ? base[(size_t)off * NX + (size_t)r * NX + c] : ((r == c) ? diag : 0.f);   // off: 0 left block, NX main block
            }
        };
        float Di[CW];          // D_{k-1}^-1, this lane's entries
        float g_prev = 0.f;    // g_{k-1}[r]
        float Mn[CW], Ln[CW], Lrow[NX], gn;   // block row k's operands, fetched one step ahead
        load_slice(S, NX, Mn, 1.0f);
        gn = rowok ? gam[NX + r] : 0.f;
#pragma unroll
        for (int i = 0; i < CW; i++) Ln[i] = 0.f;
#pragma unroll
        for (int j = 0; j < NX; j++) Lrow[j] = 0.f;
        for (int k = 0; k < N; k++) {
            float D[CW], Lc[CW], Lr[NX], g = gn;
#pragma unroll
            for (int i = 0; i < CW; i++) { D[i] = Mn[i]; Lc[i] = Ln[i]; }
#pragma unroll
            for (int j = 0; j < NX; j++) Lr[j] = Lrow[j];
            if (k + 1 < N) {   // prefetch block row k+1 (its latency hides behind this step's elimination)
                const float* Sn = S + (size_t)(k + 1) * BROW;
                load_slice(Sn, NX, Mn, 1.0f);
                load_slice(Sn, 0, Ln, 0.f);
#pragma unroll
                for (int j = 0; j < NX; j++) Lrow[j] = rowok ? Sn[(size_t)r * NX + j] : 0.f;
                gn = rowok ? gam[(k + 2) * NX + r] : 0.f;
            }
            if (k > 0) {
#pragma unroll
                for (int i = 0; i < CW; i++) {
                    sD[r * LD + c0 + i] = Di[i];
                    sL[r * LD + c0 + i] = Lc[i];
                }
                if (cq == 0) sv[r] = g_prev;
                __syncthreads();
                // W(r, own columns) = left_k(r, :) D_{k-1}^-1(:, own columns)
                float W[CW];
#pragma unroll
                for (int i = 0; i < CW; i++) W[i] = 0.f;
#pragma unroll
                for (int j = 0; j < NX; j++) {
#pragma unroll
                    for (int i = 0; i < CW; i++) W[i] = __builtin_fmaf(Lr[j], sD[j * LD + c0 + i], W[i]);
                }
#pragma unroll
                for (int i = 0; i < CW; i++) sW[r * LD + c0 + i] = W[i];
                __syncthreads();
                // D_k(r, own columns) = main_k - W(r, :) left_k(own columns, :)^T ;  g_k(r) = gamma_k(r) - W(r, :) g_{k-1}
                float Wr[NX];
#pragma unroll
                for (int j = 0; j < NX; j++) Wr[j] = sW[r * LD + j];
                float wg = 0.f;
#pragma unroll
                for (int j = 0; j < NX; j++) wg = __builtin_fmaf(Wr[j], sv[j], wg);
                g = g - wg;
#pragma unroll
                for (int i = 0; i < CW; i++) {
                    float acc = 0.f;
#pragma unroll
                    for (int j = 0; j < NX; j++) acc = __builtin_fmaf(Wr[j], sL[(c0 + i) * LD + j], acc);
                    D[i] = D[i] - acc;
                }
                __syncthreads();   // sD / sL / sW / sv are free for the next block row
            }
            gj_quad_step<NX, CW, 0>(D, r, cq, cq * 4);
#pragma unroll
            for (int i = 0; i < CW; i++) Di[i] = D[i];
            g_prev = g;
            if (rowok) {
#pragma unroll
                for (int i = 0; i < CW; i++)
                    if (c0 + i < NX) Dinv[(size_t)k * BROW + NX * NX + (size_t)r * NX + c0 + i] = Di[i];
                if (cq == 0) lam[(k + 1) * NX + r] = g;   // g_k parks in lambda until the back substitution overwrites it
            }
        }
        // back substitution; Di / g_prev hold block N-1.  Rows are split over the quad for the transposed product: group cq takes rows cq, cq+4, ...
        float lnext = 0.f;
        for (int k = N - 1; k >= 0; k--) {
            float rhs = g_prev;
            if (k < N - 1) {
                const float* Ln1 = S + (size_t)(k + 1) * BROW;   // left_{k+1}: entry (i, r)
                if (cq == 0) sv[r] = lnext;
                __syncthreads();
                float acc = 0.f;
#pragma unroll
                for (int q = 0; q < 4; q++) {
                    const int i = cq + 4 * q;
                    if (i < NX && rowok) acc = __builtin_fmaf(Ln1[(size_t)i * NX + r], sv[i], acc);
                }
                rhs = rhs - quad_sum(acc);
            }
            if (cq == 0) sv[16 + r] = rhs;
            __syncthreads();
            float part = 0.f;
#pragma unroll
            for (int i = 0; i < CW; i++) part = __builtin_fmaf(Di[i], (c0 + i < NX) ? sv[16 + c0 + i] : 0.f, part);
            const float l = quad_sum(part);
            lnext = l;
            __syncthreads();
            if (k > 0) {  // block k-1: D^-1 from the P^-1 buffer, g_{k-1} from lambda's slot k
                load_slice(Dinv + (size_t)(k - 1) * BROW, NX, Di, 1.0f);
                g_prev = rowok ? lam[k * NX + r] : 0.f;
            }
            if (rowok && cq == 0) lam[(k + 1) * NX + r] = l;
        }
    }
    if (lane == 0) {
        // statistics: one "iteration"; a trajectory is never declared converged by the PCG rule (0 iterations, bsqp.cuh:153) in this mode
        const uint32_t it = skip ? 0u : 1u;
        bf.pcg_iters[b] = it;
        bf.st_pcg_iters[(size_t)sqp_iter * B + b] = (int32_t)it;
        if (skip) atomicAdd(&bf.num_solved_w[sqp_iter], 1u);
    }
}

// ---- direct solve, parallel over the knots: block cyclic reduction (round 3) ------------------------------------------------------------
// The sweep above is a chain of N dependent block eliminations (N x ~2.4 us at a lone wavefront's issue rate); PCG at a long horizon is
// 110-130 iterations of 1.9 us.  Neither uses more than a fraction of one CU when the batch is small -- the MPC case (B = 1..8), where the
// reference's published N = 128 times were still ahead.  Cyclic reduction has log2(N) levels: at stride s every second active block row
// j = s-1 + 2 s u is eliminated, its neighbours i = j +- s keep
//     D_i <- D_i - L_i D_{i-s}^-1 L_i^T - L_{i+s}^T D_{i+s}^-1 L_{i+s},   g_i <- g_i - L_i D_{i-s}^-1 g_{i-s} - L_{i+s}^T D_{i+s}^-1 g_{i+s},
//     L_i <- - L_i D_{i-s}^-1 L_{i-s}                                      (L_i: the coupling of row i to its LEFT active neighbour)
// and the system stays symmetric block-tridiagonal over the kept rows (right_i = left_{i+2s}^T), so only left and main blocks exist.  All
// eliminations of a level are independent: one workgroup of WAVES wavefronts per trajectory deals them to its wavefronts (the 14 x 14 /
// 12 x 12 algebra of one block row in the sweep's lane = (row, column group) layout, operands through a per-wavefront LDS tile), two
// workgroup barriers per level.  After log2(N) levels one block row is left; back substitution runs the levels in reverse:
//     x_j = D_j^-1 (g_j - L_j x_{j-s} - C_j^T x_{j+s}),   C_j = the coupling block L_{j+s} as it was when j was eliminated.
// -S is symmetric positive definite, every reduced block is a Schur complement of it: no pivoting needed, like the sweep.
// Storage: S's left / main blocks are overwritten (the system is re-formed every SQP iteration); the P^-1 buffer (unused in this mode)
// keeps D_j^-1 (left slot) and C_j (main slot) of the eliminated rows; g lives in lambda until x overwrites it.
template<class M, int WAVES>
__global__ __launch_bounds__(64 * WAVES) void btd_cr_kernel(Buffers bf, int N, int B, int sqp_iter)
{
    constexpr int NQ = M::NQ, NX = 2 * NQ, BLK = NX * NX, BROW = 3 * NX * NX;
    constexpr int CW = (NX + 3) / 4, LD = 20, TOP = 16;   // TOP: active rows from which on the reduction runs out of LDS
    // per-wavefront operand tiles: 16 rows of LD floats; column c sits at P(c) = 4 (c / CW) + c % CW, so that a lane's CW columns are ONE
    // aligned 16-byte LDS access (the product loops were 56 ds_read_b32 per 14 x 14 product otherwise) and a whole row is four of them
    auto P = [](int c) { return 4 * (c / CW) + (c % CW); };
#ifdef GATO_DOUBLE
    constexpr bool MFMA = false;   // the validation build keeps the packed-FMA products (and half the wavefronts: its reals are twice as wide)
    constexpr int NT = 2, TSZ = 16 * LD;
#else
    // fp32: the kept rows' five 14 x 14 x 14 (12 x 12 x 12) products run on the matrix cores.  Here ONE wavefront forms ONE product at a
    // time -- the shape v_mfma_f32_16x16x4_f32 is made for (77 % / 56 % of the tile filled, 4 issues of 32 cycles per product against
    // ~80 vector instructions) -- unlike the PCG prologue, where a wavefront's lanes work on 16 knots side by side.
    constexpr bool MFMA = true;
    constexpr int NT = 4, TSZ = 16 * 17;
#endif
    __shared__ __attribute__((aligned(32))) float tiles[WAVES][NT][TSZ];
    __shared__ float vecs[WAVES][32];
    extern __shared__ __attribute__((aligned(16))) float top[];   // [L | D | D^-1 | C][TOP][BLK] + g [TOP][NX]: the last log2(TOP) levels
    if (bf.ctrl->done) return;
    const int b = blockIdx.x, lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int r = lane >> 2, cq = lane & 3;
    const int c0 = cq * CW;
    const bool rowok = r < NX;
    float* S = bf.S + (size_t)b * N * BROW;
    float* W = bf.Pinv + (size_t)b * N * BROW;
    const float* gam = bf.gamma + (size_t)b * (N + 2) * NX;
    float* g = bf.lambda + (size_t)b * (N + 2) * NX + NX;   // g_k, later x_k, at g + k NX
    const bool skip = bf.converged[b] != 0;

    if (!skip) {
        float* tA = tiles[wv][0];
        float* tB = tiles[wv][1];
        float* tv = vecs[wv];
        // Lanes of ONE wavefront hand values to each other through its LDS tiles: a wavefront-scope fence between the stores and the loads
        // (an s_waitcnt, no barrier).  Without it the loads were seen returning the tile's previous contents (measured: g of the first kept row).
        auto lds_handoff = [&]() { __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); };
        // Two tiers of storage.  While more than TOP block rows are active they live in global memory (L2): 2 N nx^2 floats do not fit
        // LDS, and with >= one elimination per wavefront the four wavefronts of a SIMD hide each other's load latency.  From TOP active
        // rows on there are fewer eliminations than wavefronts and every level is ONE task deep: its operands' latency is the level's
        // duration, so the active rows move to LDS (4 x 16 blocks) and the last log2(TOP) levels and their back substitution run from there.
        const int s_top = N > TOP ? N / TOP : 1;              // stride from which on the active rows are in LDS
        int sh = 0;
        while ((1 << sh) < s_top) sh++;
        struct Tier { float *L, *D, *Wd, *Wc, *g; int bs, shift; };
        const Tier glob{S, S + BLK, W, W + BLK, g, BROW, 0};
        const Tier lds{top, top + TOP * BLK, top + 2 * TOP * BLK, top + 3 * TOP * BLK, top + 4 * TOP * BLK, BLK, sh};
        auto at = [&](const Tier& T, int i) { return ((i + 1) >> T.shift) - 1; };   // position of block row i in its tier

        // entry (r, c0 + i) of a row-major NX x NX block; the padding to 16 x 16 is `diag` on the diagonal, 0 elsewhere
        auto load_slice = [&](const float* blk, float* out, float diag) {
#pragma unroll
            for (int i = 0; i < CW; i++) {
                const int c = c0 + i;
                out[i] = (rowok && c < NX) ? blk[(size_t)r * NX + c] : ((r == c) ? diag : 0.f);
            }
        };
        auto store_slice = [&](float* blk, const float* in) {
#pragma unroll
            for (int i = 0; i < CW; i++)
                if (rowok && c0 + i < NX) blk[(size_t)r * NX + c0 + i] = in[i];
        };
        auto tile_put = [&](float* t, const float* sl) {
            lds_handoff();   // the tile's previous readers are done
            reinterpret_cast<real4*>(t + r * LD)[cq] = make_real4(sl[0], sl[1], sl[2], CW > 3 ? sl[CW - 1] : 0.f);
            lds_handoff();
        };
        auto tile_row = [&](const float* t, float* row) {      // row r of the tile
            float w[16];
#pragma unroll
            for (int q = 0; q < 4; q++) {
                const real4 v = reinterpret_cast<const real4*>(t + r * LD)[q];
                w[4 * q] = v.x; w[4 * q + 1] = v.y; w[4 * q + 2] = v.z; w[4 * q + 3] = v.w;
            }
#pragma unroll
            for (int j = 0; j < NX; j++) row[j] = w[P(j)];
        };
        auto tile_col = [&](const float* t, float* col) {      // column r of the tile = row r of its transpose
            const int pr = 4 * (r / CW) + (r % CW);
#pragma unroll
            for (int j = 0; j < NX; j++) col[j] = t[j * LD + pr];
        };
        auto row_x_tile = [&](const float* row, const float* t, float* out) {    // (row . B)(own columns)
            float o[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int j = 0; j < NX; j++) {
                const real4 v = reinterpret_cast<const real4*>(t + j * LD)[cq];
                o[0] = __builtin_fmaf(row[j], v.x, o[0]);
                o[1] = __builtin_fmaf(row[j], v.y, o[1]);
                o[2] = __builtin_fmaf(row[j], v.z, o[2]);
                o[3] = __builtin_fmaf(row[j], v.w, o[3]);
            }
#pragma unroll
            for (int i = 0; i < CW; i++) out[i] = o[i];
        };
        auto row_x_tileT = [&](const float* row, const float* t, float* out) {   // (row . B^T)(own columns)
#pragma unroll
            for (int i = 0; i < CW; i++) {
                float w[16];
#pragma unroll
                for (int q = 0; q < 4; q++) {
                    const real4 v = reinterpret_cast<const real4*>(t + (c0 + i) * LD)[q];
                    w[4 * q] = v.x; w[4 * q + 1] = v.y; w[4 * q + 2] = v.z; w[4 * q + 3] = v.w;
                }
                float a = 0.f;
#pragma unroll
                for (int j = 0; j < NX; j++) a = __builtin_fmaf(row[j], w[P(j)], a);
                out[i] = a;
            }
        };
        auto row_dot_vec = [&](const float* row, const float* gv) {               // row . g (through the wavefront's vector slot)
            lds_handoff();
            if (cq == 0) tv[r] = rowok ? gv[r] : 0.f;
            lds_handoff();
            float a = 0.f;
#pragma unroll
            for (int j = 0; j < NX; j++) a = __builtin_fmaf(row[j], tv[j], a);
            return a;
        };

        for (int i = threadIdx.x; i < N * NX; i += blockDim.x) g[i] = gam[NX + i];
        __syncthreads();

        // ---- reduction ----
        for (int s = 1; s < N; s <<= 1) {
            if (s == s_top) {   // the active rows (every s-th) move to LDS
                const int na = N / s;
                for (int e = threadIdx.x; e < na * BLK; e += blockDim.x) {
                    const int u = e / BLK, o = e - u * BLK;
                    const int i = s * (u + 1) - 1;
                    lds.L[e] = S[(size_t)i * BROW + o];
                    lds.D[e] = S[(size_t)i * BROW + BLK + o];
                }
                for (int e = threadIdx.x; e < na * NX; e += blockDim.x) {
                    const int u = e / NX, o = e - u * NX;
                    lds.g[e] = g[(size_t)(s * (u + 1) - 1) * NX + o];
                }
                __syncthreads();
            }
            const Tier& T = s >= s_top ? lds : glob;
            const int cnt = N / (2 * s);
            for (int t = wv; t < cnt; t += WAVES) {            // the eliminated rows' inverses
                const int j = at(T, s - 1 + 2 * s * t);
                float D[CW];
                load_slice(T.D + (size_t)j * T.bs, D, 1.0f);
                gj_quad_step<NX, CW, 0>(D, r, cq, cq * 4);
                store_slice(T.Wd + (size_t)j * T.bs, D);
            }
            __syncthreads();
            for (int t = wv; t < cnt; t += WAVES) {            // the kept rows take both neighbours in
                const int ri = 2 * s - 1 + 2 * s * t;
                const bool has_ll = ri - 2 * s >= 0, has_r = ri + s < N;     // wavefront-uniform
                const int i = at(T, ri), jm = at(T, ri - s), jp = at(T, has_r ? ri + s : ri);
                float* Li_p = T.L + (size_t)i * T.bs;
                float* Di_p = T.D + (size_t)i * T.bs;
#ifndef GATO_DOUBLE
                {
                    typedef float mf4 __attribute__((ext_vector_type(4)));
                    constexpr int LM = 17;
                    const int mn = lane & 15, mg = lane >> 4;   // MFMA coordinates: operand row / column, k group = output row group
                    float* T0 = tiles[wv][0];
                    float* T1 = tiles[wv][1];
                    float* T2 = tiles[wv][2];
                    float* T3 = tiles[wv][3];
                    // a block in the MFMA's accumulator layout: v[q] = entry (4 mg + q, mn); zero outside NX x NX
                    auto gload = [&](const float* blk, float* v) {
#pragma unroll
                        for (int q = 0; q < 4; q++) v[q] = (4 * mg + q < NX && mn < NX) ? blk[(size_t)(4 * mg + q) * NX + mn] : 0.f;
                    };
                    auto gstore = [&](float* blk, const float* v) {
#pragma unroll
                        for (int q = 0; q < 4; q++)
                            if (4 * mg + q < NX && mn < NX) blk[(size_t)(4 * mg + q) * NX + mn] = v[q];
                    };
                    auto tput = [&](float* t, const float* v) {
                        lds_handoff();   // the tile's previous readers are done
#pragma unroll
                        for (int q = 0; q < 4; q++) t[(4 * mg + q) * LM + mn] = v[q];
                        lds_handoff();
                    };
                    // c += op(X) op(Y), X and Y 16 x 16 tiles: four issues over k; operand layout a = A[mn][4 kk + mg], b = B[4 kk + mg][mn]
                    auto mm = [&](const float* X, bool xT, const float* Y, bool yT, mf4 c) {
#pragma unroll
                        for (int kk = 0; kk < 4; kk++) {
                            const int kx = 4 * kk + mg;
                            const float av = xT ? X[kx * LM + mn] : X[mn * LM + kx];
                            const float bv = yT ? Y[mn * LM + kx] : Y[kx * LM + mn];
                            c = __builtin_amdgcn_mfma_f32_16x16x4f32(av, bv, c, 0, 0, 0);
                        }
                        return c;
                    };
                    // rows of a tile times a vector, in the (r, cq) layout: sum over the lane's own columns, then over the quad
                    auto tile_dot = [&](const float* t, float gval) {
                        lds_handoff();
                        if (cq == 0) tv[r] = gval;
                        lds_handoff();
                        float part = 0.f;
#pragma unroll
                        for (int q = 0; q < CW; q++) part = __builtin_fmaf(t[r * LM + c0 + q], (c0 + q < NX) ? tv[c0 + q] : 0.f, part);
                        return quad_sum(part);
                    };
                    // every operand is fetched before the algebra starts (6 x 4 registers in this layout): one memory latency per task
                    float li[4], dm[4], di[4], ljm[4], ljp[4], dp[4];
                    gload(Li_p, li);
                    gload(T.Wd + (size_t)jm * T.bs, dm);
                    gload(Di_p, di);
                    gload(T.L + (size_t)jm * T.bs, ljm);               // (meaningless and unused when !has_ll)
                    gload(T.L + (size_t)jp * T.bs, ljp);               // (jp == i when !has_r: unused)
                    gload(T.Wd + (size_t)jp * T.bs, dp);
                    float gi = rowok ? T.g[(size_t)i * NX + r] : 0.f;
                    const float gm = rowok ? T.g[(size_t)jm * NX + r] : 0.f;
                    const float gp = rowok ? T.g[(size_t)jp * NX + r] : 0.f;
                    const mf4 zero = {0.f, 0.f, 0.f, 0.f};
                    tput(T0, li);
                    tput(T1, dm);
                    const mf4 wr = mm(T0, false, T1, false, zero);     // Wr = L_i D_jm^-1
                    float w4[4] = {wr.x, wr.y, wr.z, wr.w};
                    tput(T2, w4);
                    mf4 ad = mm(T2, false, T0, true, zero);            // Wr L_i^T
                    gi -= tile_dot(T2, gm);
                    tput(T1, ljm);                                     // (the reads of D_jm^-1 are done: LDS is in order per wavefront)
                    const mf4 nl = mm(T2, false, T1, false, zero);     // Wr L_jm
                    float newL[4] = {has_ll ? -nl.x : 0.f, has_ll ? -nl.y : 0.f, has_ll ? -nl.z : 0.f, has_ll ? -nl.w : 0.f};
                    if (has_r) {
                        tput(T0, ljp);
                        tput(T1, dp);
                        const mf4 wl = mm(T0, true, T1, false, zero);  // Wl = L_jp^T D_jp^-1
                        float v4[4] = {wl.x, wl.y, wl.z, wl.w};
                        tput(T3, v4);
                        ad = mm(T3, false, T0, false, ad);             // + Wl L_jp
                        gi -= tile_dot(T3, gp);
                    }
                    di[0] -= ad.x; di[1] -= ad.y; di[2] -= ad.z; di[3] -= ad.w;
                    gstore(T.Wc + (size_t)jm * T.bs, li);              // C_jm: what couples jm to its right neighbour, as of now
                    gstore(Li_p, newL);
                    gstore(Di_p, di);
                    if (rowok && cq == 0) T.g[(size_t)i * NX + r] = gi;
                    continue;
                }
#endif
                float Di[CW], Li[CW], tmp[CW], acc[CW], newL[CW], row[NX], wrow[NX];
                load_slice(Di_p, Di, 1.0f);
                load_slice(Li_p, Li, 0.f);
                float gi = rowok ? T.g[(size_t)i * NX + r] : 0.f;
                // left neighbour jm:  Wr = L_i D_jm^-1
                tile_put(tA, Li);
                tile_row(tA, row);                                           // row r of L_i
                load_slice(T.Wd + (size_t)jm * T.bs, tmp, 0.f);
                tile_put(tB, tmp);
                row_x_tile(row, tB, acc);                                    // Wr(r, own columns)
                tile_put(tB, acc);
                tile_row(tB, wrow);                                          // Wr(r, :)
                row_x_tileT(wrow, tA, acc);                                  // Wr L_i^T
#pragma unroll
                for (int q = 0; q < CW; q++) Di[q] -= acc[q];
                gi -= row_dot_vec(wrow, T.g + (size_t)jm * NX);
                if (has_ll) {                                                // L_i <- - Wr L_jm  (the first active row has no left neighbour: 0)
                    load_slice(T.L + (size_t)jm * T.bs, tmp, 0.f);
                    tile_put(tB, tmp);
                    row_x_tile(wrow, tB, newL);
#pragma unroll
                    for (int q = 0; q < CW; q++) newL[q] = -newL[q];
                } else {
#pragma unroll
                    for (int q = 0; q < CW; q++) newL[q] = 0.f;
                }
                if (has_r) {                                                 // right neighbour jp:  Wl = L_jp^T D_jp^-1
                    load_slice(T.L + (size_t)jp * T.bs, tmp, 0.f);
                    tile_put(tA, tmp);                                       // L_jp
                    tile_col(tA, row);                                       // row r of L_jp^T
                    load_slice(T.Wd + (size_t)jp * T.bs, tmp, 0.f);
                    tile_put(tB, tmp);
                    row_x_tile(row, tB, acc);                                // Wl(r, own columns)
                    tile_put(tB, acc);
                    tile_row(tB, wrow);
                    row_x_tile(wrow, tA, acc);                               // Wl L_jp
#pragma unroll
                    for (int q = 0; q < CW; q++) Di[q] -= acc[q];
                    gi -= row_dot_vec(wrow, T.g + (size_t)jp * NX);
                }
                store_slice(T.Wc + (size_t)jm * T.bs, Li);                   // C_jm: what couples jm to its right neighbour, as of now
                store_slice(Li_p, newL);
                store_slice(Di_p, Di);
                if (rowok && cq == 0) T.g[(size_t)i * NX + r] = gi;
            }
            __syncthreads();
        }
        // ---- the last block row (always in the LDS tier), then back substitution ----
        if (wv == 0) {
            const int e = at(lds, N - 1);
            float D[CW];
            load_slice(lds.D + (size_t)e * BLK, D, 1.0f);
            gj_quad_step<NX, CW, 0>(D, r, cq, cq * 4);
            if (cq == 0) tv[r] = rowok ? lds.g[(size_t)e * NX + r] : 0.f;
            lds_handoff();
            float part = 0.f;
#pragma unroll
            for (int i = 0; i < CW; i++) part = __builtin_fmaf(D[i], (c0 + i < NX) ? tv[c0 + i] : 0.f, part);
            const float x = quad_sum(part);
            if (rowok && cq == 0) lds.g[(size_t)e * NX + r] = x;
        }
        __syncthreads();
        for (int s = N >> 1; s >= 1; s >>= 1) {
            if (s < s_top && 2 * s >= s_top) {   // leaving the LDS tier: its rows' solutions go where the global tier looks for x
                const int na = N / s_top;
                for (int e = threadIdx.x; e < na * NX; e += blockDim.x) {
                    const int u = e / NX, o = e - u * NX;
                    g[(size_t)(s_top * (u + 1) - 1) * NX + o] = lds.g[e];
                }
                __syncthreads();
            }
            const Tier& T = s >= s_top ? lds : glob;
            const int cnt = N / (2 * s);
            for (int t = wv; t < cnt; t += WAVES) {
                const int rj = s - 1 + 2 * s * t;                            // rj + s <= N - 1 always
                const bool has_l = rj - s >= 0;
                const int j = at(T, rj), jm = at(T, has_l ? rj - s : rj), jp = at(T, rj + s);
                float rhs = rowok ? T.g[(size_t)j * NX + r] : 0.f;
                float tmp[CW];
                if (has_l) {                                                 // - L_j x_jm
                    load_slice(T.L + (size_t)j * T.bs, tmp, 0.f);
                    lds_handoff();
                    if (cq == 0) tv[r] = rowok ? T.g[(size_t)jm * NX + r] : 0.f;
                    lds_handoff();
                    float part = 0.f;
#pragma unroll
                    for (int i = 0; i < CW; i++) part = __builtin_fmaf(tmp[i], (c0 + i < NX) ? tv[c0 + i] : 0.f, part);
                    rhs -= quad_sum(part);
                }
                {                                                            // - C_j^T x_jp: the quad splits the rows of C_j
                    const float* C = T.Wc + (size_t)j * T.bs;
                    if (cq == 0) tv[16 + r] = rowok ? T.g[(size_t)jp * NX + r] : 0.f;
                    lds_handoff();
                    float part = 0.f;
#pragma unroll
                    for (int q = 0; q < 4; q++) {
                        const int i = cq + 4 * q;
                        if (i < NX && rowok) part = __builtin_fmaf(C[(size_t)i * NX + r], tv[16 + i], part);
                    }
                    rhs -= quad_sum(part);
                }
                load_slice(T.Wd + (size_t)j * T.bs, tmp, 0.f);               // D_j^-1
                lds_handoff();
                if (cq == 0) tv[r] = rhs;
                lds_handoff();
                float part = 0.f;
#pragma unroll
                for (int i = 0; i < CW; i++) part = __builtin_fmaf(tmp[i], (c0 + i < NX) ? tv[c0 + i] : 0.f, part);
                const float x = quad_sum(part);
                if (rowok && cq == 0) T.g[(size_t)j * NX + r] = x;
            }
            __syncthreads();
        }
        if (s_top == 1) {   // short horizons never left LDS: the whole solution is there
            for (int e = threadIdx.x; e < N * NX; e += blockDim.x) g[e] = lds.g[e];
        }
    }
    if (threadIdx.x == 0) {
        const uint32_t it = skip ? 0u : 1u;   // statistics as in the sweep: one "iteration", never converged by the 0-iterations rule
        bf.pcg_iters[b] = it;
        bf.st_pcg_iters[(size_t)sqp_iter * B + b] = (int32_t)it;
        if (skip) atomicAdd(&bf.num_solved_w[sqp_iter], 1u);
    }
}

// =========================================================================================================================
// dz recovery (computeDzBatchedKernel, schur_linsys.cuh:316-431), one lane per (b,k); q, r are overwritten by the KKT residuals
// =========================================================================================================================
// dz of knot k of trajectory b (computeDz, kkt.cuh), also left in `mirror` (the trajectory's step in LDS) when given
// PART: 0 = state and control rows of knot k, 1 = state row only, 2 = control row only (two lanes share a knot in the step kernel:
// each then holds only its half of D and the kernel's register count is the larger half, not the sum)
template<class M, int PART = 0>
GATO_DEV void dz_knot(const Buffers& bf, int N, int b, int k, float dt, float* mirror)
{
    constexpr int NQ = M::NQ, NX = 2 * NQ, NU = NQ, KS = NX + NU;
    const size_t bk = (size_t)b * N + k;
    const int traj = KS * N - NU;
    const float* lam = bf.lambda + (size_t)b * (N + 2) * NX;
    float* dz = bf.dz + (size_t)b * traj + (size_t)k * KS;
    const float h2 = half_dt_sq(dt);
    float lk1[NX], Dm[3 * NQ * NQ];  // only the part of D a PART reads is loaded (and allocated)
    const bool inner = k < N - 1;
    // unconditional loads (valid memory for the last knot too: lambda's zero padding block, an unused D slot)
    load_vec<NX, NX>(lk1, lam + (size_t)(k + 2) * NX);
    if constexpr (PART != 2) load_vec<2 * NQ * NQ, 3 * NQ * NQ>(Dm, bf.D + bk * 3 * NQ * NQ);
    if constexpr (PART != 1) load_vec<NQ * NQ, NQ * NQ>(Dm + 2 * NQ * NQ, bf.D + bk * 3 * NQ * NQ + 2 * NQ * NQ);
    if constexpr (PART != 2) {  // state row
        float lk[NX], qk[NX], res[NX], out[NX];
        load_vec<NX, NX>(lk, lam + (size_t)(k + 1) * NX);
        load_vec<NX, NX>(qk, bf.q + bk * NX);
#pragma unroll
        for (int x = 0; x < NX; x++) {
            float s = 0.f;
            if (inner) {
#pragma unroll
                for (int j = 0; j < NX; j++) s += lk1[j] * A_elem<NQ>(Dm, j, x, dt, h2);
                s = -s;
            }
            const float scr = s + lk[x];
            res[x] = qk[x] - scr;
        }
        float Qi[NQ * NQ], di[NQ];
        if (opaque_true()) {  // fetched only now: D's 2 nq^2 registers are free again
            load_vec<NQ * NQ, NQ * NQ>(Qi, bf.Qqi + bk * NQ * NQ);
            load_vec<NQ, NQ>(di, bf.Qdi + bk * NQ);
        }
#pragma unroll
        for (int y = 0; y < NX; y++) {
            float s = 0.f;
            if (y < NQ) {
#pragma unroll
                for (int j = 0; j < NQ; j++) s += Qi[j * NQ + y] * res[j];
            } else {
                s = di[y - NQ] * res[y];
            }
            out[y] = -1.0f * s;
        }
        if (inner) {
            store_vec<NX, 2>(dz, out);
        } else {
#pragma unroll
            for (int i = 0; i < NX; i++) dz[i] = out[i];  // last knot: dz + k*KS is only 8-byte aligned in general
        }
        if (mirror) {
#pragma unroll
            for (int i = 0; i < NX; i++) mirror[(size_t)k * KS + i] = out[i];
        }
        store_vec<NX, NX>(bf.q + bk * NX, res);
    }
    if constexpr (PART != 1) {  // control row
        float* rk = bf.r + bk * NU;
        if (!inner) {
#pragma unroll
            for (int i = 0; i < NU; i++) rk[i] = 0.f;
            return;
        }
        float rr[NU], ri[NU], su[NU], out[NU];
        load_vec<NU, NU>(rr, rk);
        load_vec<NU, NU>(ri, bf.Rdi + bk * NU);
#pragma unroll
        for (int x = 0; x < NU; x++) {
            float s = 0.f;
#pragma unroll
            for (int j = 0; j < NX; j++) s += lk1[j] * B_elem<NQ>(Dm, j, x, dt, h2);
            su[x] = rr[x] - (-s);
            out[x] = -1.0f * (ri[x] * su[x]);
        }
#pragma unroll
        for (int i = 0; i < NU; i++) dz[NX + i] = out[i];
        if (mirror) {
#pragma unroll
            for (int i = 0; i < NU; i++) mirror[(size_t)k * KS + NX + i] = out[i];
        }
        store_vec<NU, NU>(rk, su);
    }
}

template<class M>
__global__ __launch_bounds__(256) void dz_kernel(Buffers bf, int N, int B, float dt, int sqp_iter)
{
    if (bf.ctrl->done) return;
    const int g = blockIdx.x * blockDim.x + threadIdx.x;
    if (g == 0) bf.ctrl->iters_done = sqp_iter + 1;
    const int k = g % N, b = g / N;
    if (b >= B) return;
    dz_knot<M>(bf, N, b, k, dt, nullptr);
}

// =========================================================================================================================
// line search + trajectory update + rho adaptation (line_search.cuh:13-98): one workgroup per trajectory
// =========================================================================================================================
// the line search of trajectory b by its workgroup: mer = the 8 merits, dz = the step (global or LDS)
// `drho_reset` (non-null in the LAST iteration of a solve): drho goes back to its default once the search has used it (bsqp.cuh:189)
// cur_lds: the merit of the current iterate when the caller has just formed it (first step launch of a solve) instead of
// bf.merit_cur[b], which thread 0 then initialises
GATO_DEV void line_search_block(const Buffers& bf, int b, int B, int traj, const float* mer, const float* dz, int adapt_rho, int sqp_iter,
                                const float* drho_reset, const float* cur_lds = nullptr)
{
    float best = 1e38f;
    uint32_t idx = 0;
#pragma unroll
    for (uint32_t i = 0; i < NUM_ALPHAS; i++) {
        const float m = mer[i];
        if (m < best) { best = m; idx = i; }  // first minimum: ties keep the larger alpha
    }
    const float cur = cur_lds ? *cur_lds : bf.merit_cur[b];
    const bool success = best < cur;
    __syncthreads();  // everyone has read merit_cur before thread 0 overwrites it
    if (threadIdx.x == 0) {
        if (cur_lds) bf.merit_cur[b] = cur;
        if (adapt_rho) {
            const float dr = bf.drho[b];
            const float mult = success ? fminf(dr / RHO_FACTOR, 1 / RHO_FACTOR) : fmaxf(dr * RHO_FACTOR, RHO_FACTOR);
            bf.drho[b] = mult;
            float r = fmaxf(bf.rho[b] * mult, RHO_MIN);
            r = fminf(r, RHO_MAX);
            bf.rho[b] = r;
        }
        // line_search.cuh:77-79: a failed search resets rho > RHO_MAX to RHO_INIT -- dead after the clamp above, live when
        // adaptation is off and the caller's rho_batch exceeds RHO_MAX
        if (!success && bf.rho[b] > RHO_MAX) bf.rho[b] = RHO_INIT;
        float step = -1.f;
        if (success) {
            step = (float)(1.0 / (double)(1 << idx));
            bf.merit_cur[b] = best;
        }
        bf.step[b] = step;
        bf.st_step[(size_t)sqp_iter * B + b] = step;
        bf.st_min_merit[(size_t)sqp_iter * B + b] = success ? best : cur;
        if (b == 0) bf.ctrl->ls_done = sqp_iter + 1;
        if (drho_reset) bf.drho[b] = drho_reset[b];
    }
    if (success) {
        const float step = (float)(1.0 / (double)(1 << idx));
        float* x = bf.xu + (size_t)b * traj;
        for (int i = threadIdx.x; i < traj; i += blockDim.x) x[i] += step * dz[i];
    }
}
// drho_init: the per-trajectory default drho; the solve's exit paths -- the last iteration's search, or the solve_ratio break before
// a search -- put drho back to it (bsqp.cuh:189) so that no copy has to follow the loop
__global__ __launch_bounds__(128) void line_search_kernel(Buffers bf, int traj, int B, int adapt_rho, int sqp_iter, float thresh,
                                                          const float* __restrict__ drho_init, int last_iter)
{
    if (bf.ctrl->done) return;
    const int b = blockIdx.x;
    if ((float)bf.num_solved[sqp_iter] >= thresh) {
        if (blockIdx.x == 0 && threadIdx.x == 0) bf.ctrl->done = 1;  // every block takes the same branch; later kernels see done
        if (threadIdx.x == 0) bf.drho[b] = drho_init[b];
        return;
    }
    line_search_block(bf, b, B, traj, bf.merit + (size_t)b * NUM_ALPHAS, bf.dz + (size_t)b * traj, adapt_rho, sqp_iter, last_iter ? drho_init : nullptr);
}

// dz + merit at the 8 step sizes + line search in ONE launch, a workgroup of 8 N lanes per trajectory (N <= 64): the step never
// leaves the CU between the three (LDS), two launches and their cache write-back / invalidate are gone.  Lane t < N forms dz_t, then
// lane t is (alpha index t / N, knot t % N) of the merit evaluation, then all lanes apply the chosen step.
// MAXT = 512 serves N <= 64 (the 8 merits of a trajectory are wave-level sums); MAXT = 1024 (N = 128: 128 registers per lane) sums a
// merit over two wavefronts through LDS in the order of the stand-alone merit kernel.
// One workgroup: order[] <- the trajectories sorted by this iteration's PCG iteration count, largest first (counting sort over 256 bins;
// ties in the arrival order of the atomics -- any order is a valid schedule, the results do not depend on it).  Where a PCG launch
// cannot hold every trajectory at once (iiwa14 N = 64, B = 512: one 7-wavefront workgroup per CU, two rounds) the long-running
// trajectories -- the same ones from iteration to iteration -- start first instead of waiting for a slot.
// key(i) in [0, 255]: larger = harder = earlier in order[]
template<class KeyFn> GATO_DEV void order_by(const Buffers& bf, int B, int* scratch /* 512 ints of LDS */, KeyFn key)
{
    int* hist = scratch;
    int* off = scratch + 256;
    int* order = const_cast<int*>(bf.order);
    const int t = threadIdx.x, T = blockDim.x;
    for (int i = t; i < 256; i += T) hist[i] = 0;
    __syncthreads();
    for (int i = t; i < B; i += T) atomicAdd(&hist[255 - key(i)], 1);
    __syncthreads();
    for (int i = t; i < 256; i += T) {
        int sum = 0;
        for (int j = 0; j < i; j++) sum += hist[j];
        off[i] = sum;
    }
    __syncthreads();
    for (int i = t; i < B; i += T) order[atomicAdd(&off[255 - key(i)], 1)] = i;
}
GATO_DEV void order_by_pcg_iters(const Buffers& bf, int B, int* scratch /* 512 ints of LDS */)
{
    order_by(bf, B, scratch, [&](int i) { return min((int)bf.pcg_iters[i], 255); });
}

// the step of trajectory b by a group of 8 N (first step of a solve: 9 N) threads, thread index t (the body of step_kernel).  exit_now: the
// solve_ratio rule ended the loop in this iteration (bsqp.cuh:165): dz is still formed, the line search is not
// dz of ONE trajectory with one lane per ROW (the step launch: TT = 8 N or 9 N lanes).  dz_knot gives a knot's state rows to one lane and its
// control rows to another -- 2 N of the workgroup's lanes worked through ~180 loads and ~500 dependent instructions each while the others
// waited at the barrier: 6 of the step launch's 25.8 us at C2, 11.6 of 36 at C5 (timed with the phases cut out one by one).  Here a lane
// forms one entry of the stage-1 residual (state row x: q_x - ((-(A_k^T lambda_{k+2})_x) + lambda_{k+1,x}); control row: r_x + (B_k^T
// lambda_{k+2})_x, schur_linsys.cuh:316-431) from ~20 loads -- consecutive lanes read consecutive columns of D -- and finishes every row
// that needs nothing else (Q's qd half is diagonal, R is diagonal); the q half (Q_qq^-1 times the knot's nq residuals) follows after one
// barrier from an LDS copy of the residuals.  EVERY load of both stages is issued up front: one memory round trip.  Each entry is the same
// sequence of multiply-adds as in dz_knot (A_elem / B_elem with a lane-dependent column: selects instead of folded constants): same bits.
template<class M>
GATO_DEV void dz_rows(const Buffers& bf, int N, int b, int t, int TT, float dt, float* mirror, float* resb)
{
    constexpr int NQ = M::NQ, NX = 2 * NQ, NU = NQ, KS = NX + NU;
    static_assert(NX <= 2 * NUM_ALPHAS && NU <= NUM_ALPHAS, "two state rows and one control row per lane at TT >= NUM_ALPHAS N");
    const int traj = KS * N - NU;
    const float* lam = bf.lambda + (size_t)b * (N + 2) * NX;
    float* dzg = bf.dz + (size_t)b * traj;
    const float h2 = half_dt_sq(dt);
    const size_t b0 = (size_t)b * N;
    // ---- loads: two state rows, one control row, one q-half row of stage 2 (indices clamped: loads are unconditional, stores are not)
    int sk[2], sx[2];
    bool son[2];
    float sl1[2][NX], sD[2][NQ], slk[2], sq[2], sdi[2];
#pragma unroll
    for (int rep = 0; rep < 2; rep++) {
        const int i = t + rep * TT;
        son[rep] = i < N * NX;
        const int ic = son[rep] ? i : 0;
        sk[rep] = ic / NX;
        sx[rep] = ic - sk[rep] * NX;
        const size_t bk = b0 + sk[rep];
#pragma unroll
        for (int j = 0; j < NX; j++) sl1[rep][j] = lam[(size_t)(sk[rep] + 2) * NX + j];
#pragma unroll
        for (int j = 0; j < NQ; j++) sD[rep][j] = bf.D[bk * 3 * NQ * NQ + sx[rep] * NQ + j];   // column x of [dqdd/dq | dqdd/dqd]: what A_elem(., x) reads
        slk[rep] = lam[(size_t)(sk[rep] + 1) * NX + sx[rep]];
        sq[rep] = bf.q[bk * NX + sx[rep]];
        sdi[rep] = bf.Qdi[bk * NQ + (sx[rep] >= NQ ? sx[rep] - NQ : 0)];
    }
    const bool con = t < N * NU;
    const int ck = con ? t / NU : 0, cx = con ? t - ck * NU : 0;
    float cl1[NX], cD[NQ], cr, cri;
    {
        const size_t bk = b0 + ck;
#pragma unroll
        for (int j = 0; j < NX; j++) cl1[j] = lam[(size_t)(ck + 2) * NX + j];
#pragma unroll
        for (int j = 0; j < NQ; j++) cD[j] = bf.D[bk * 3 * NQ * NQ + 2 * NQ * NQ + cx * NQ + j];   // column x of M^-1: what B_elem(., x) reads
        cr = bf.r[bk * NU + cx];
        cri = bf.Rdi[bk * NU + cx];
    }
    const bool qon = t < N * NQ;
    const int qk = qon ? t / NQ : 0, qy = qon ? t - qk * NQ : 0;
    float qi[NQ];
#pragma unroll
    for (int j = 0; j < NQ; j++) qi[j] = bf.Qqi[(b0 + qk) * NQ * NQ + j * NQ + qy];
    // ---- stage 1
#pragma unroll
    for (int rep = 0; rep < 2; rep++) {
        const int k = sk[rep], x = sx[rep];
        const bool inner = k < N - 1;
        float s = 0.f;
#pragma unroll
        for (int j = 0; j < NX; j++) {
            // A_elem<NQ>(D, j, x, dt, h2) on the fetched column
            const float d = sD[rep][j % NQ];
            float val = (j == x) ? 1.0f : 0.0f;
            if (j < NQ) {
                if (x >= NQ && j == x - NQ) val += dt;
                val += h2 * d;
            } else {
                val += dt * d;
            }
            s += sl1[rep][j] * val;
        }
        s = inner ? -s : 0.f;
        const float scr = s + slk[rep];
        const float res = sq[rep] - scr;
        if (son[rep]) {
            resb[k * NX + x] = res;
            bf.q[(b0 + k) * NX + x] = res;
            if (x >= NQ) {
                const float out = -1.0f * (sdi[rep] * res);
                dzg[(size_t)k * KS + x] = out;
                mirror[(size_t)k * KS + x] = out;
            }
        }
    }
    if (con) {
        const bool inner = ck < N - 1;
        float s = 0.f;
#pragma unroll
        for (int j = 0; j < NX; j++) {
            const float d = cD[j % NQ];
            s += cl1[j] * ((j < NQ) ? h2 * d : dt * d);   // B_elem<NQ>(D, j, x, dt, h2)
        }
        const float su = cr - (-s);
        const float out = -1.0f * (cri * su);
        float* rk = bf.r + (b0 + ck) * NU;
        if (inner) {
            dzg[(size_t)ck * KS + NX + cx] = out;
            mirror[(size_t)ck * KS + NX + cx] = out;
            rk[cx] = su;
        } else {
            rk[cx] = 0.f;
        }
    }
    __syncthreads();
    // ---- stage 2: the q half, Q_qq^-1 times the knot's residuals
    if (qon) {
        float s = 0.f;
#pragma unroll
        for (int j = 0; j < NQ; j++) s += qi[j] * resb[qk * NX + j];
        const float out = -1.0f * s;
        dzg[(size_t)qk * KS + qy] = out;
        mirror[(size_t)qk * KS + qy] = out;
    }
}

template<class M>
GATO_DEV void step_body(const Buffers& bf, int N, int B, int b, int t, int TT, float dt, int sqp_iter, bool exit_now, int adapt_rho, const float* __restrict__ drho_init,
                        int last_iter, float* __restrict__ merit_init0, float* lds)
{
    constexpr int NQ = M::NQ, NX = 2 * NQ, NU = NQ, KS = NX + NU;
    const int traj = KS * N - NU;
    float* dzs = lds;
    float* mer = lds + ((traj + 3) & ~3);   // NUM_ALPHAS merits, the current merit at [NUM_ALPHAS], wavefront partials from [12]
    if (b == 0 && t == 0) bf.ctrl->iters_done = sqp_iter + 1;
    const Costs cw = load_costs(bf, b);  // fetched now, used after the dz phase: the latency hides behind it
#if GATO_STEP_DZ_ROWS
    dz_rows<M>(bf, N, b, t, TT, dt, dzs, mer + 28);              // one lane per row; its stage-1 residuals [N nx] live behind the merits
#else
    if (t < N) dz_knot<M, 1>(bf, N, b, t, dt, dzs);              // state rows
    else if (t < 2 * N) dz_knot<M, 2>(bf, N, b, t - N, dt, dzs);  // control rows
#endif
    const int k = t % N, ai = t / N;
    // The loop breaks before the line search (bsqp.cuh:165); every workgroup takes the same branch.  `done` is raised by the NEXT
    // launch (kkt_kernel): set here it could stop a workgroup of this very launch before its dz.
    if (exit_now) {
        if (merit_init0) {   // the solve ends in its first iteration: its initial (= final) merit is still owed
            if (ai == NUM_ALPHAS) {
                float m = merit_term<M>(bf, cw, N, b, k, 0.f, 0, dzs, dt);
                m = seg_sum(m, N, mer + 12);
                if (k == 0) { bf.merit_cur[b] = m; merit_init0[b] = m; }
            }
        }
        if (t == 0) bf.drho[b] = drho_init[b];  // the solve ends here (bsqp.cuh:165,189)
        return;
    }
    __syncthreads();
    {
        const bool cand = ai < NUM_ALPHAS;
        const float alpha = cand ? (float)(1.0 / (double)(1 << ai)) : 0.f;
        float m = merit_term<M>(bf, cw, N, b, k, alpha, cand ? 1 : 0, dzs, dt);
        m = seg_sum(m, N, mer + 12);  // N <= 64: inside one wavefront; N = 128: two wavefront partials per merit through LDS
        if (k == 0) {
            mer[ai] = m;
            if (cand) bf.merit[(size_t)b * NUM_ALPHAS + ai] = m;
            else merit_init0[b] = m;
        }
    }
    __syncthreads();
    line_search_block(bf, b, B, traj, mer, dzs, adapt_rho, sqp_iter, last_iter ? drho_init : nullptr, merit_init0 ? mer + NUM_ALPHAS : nullptr);
}

template<class M, int MAXT>
__global__ __launch_bounds__(MAXT, MAXT == 512 ? 4 : 1) void step_kernel(Buffers bf, int N, int B, float dt, int sqp_iter, float thresh, int adapt_rho,
                                                    const float* __restrict__ drho_init, int last_iter, float* __restrict__ merit_init0)
{
    // merit_init0 != nullptr: the first step launch of a solve, (NUM_ALPHAS + 1) N lanes -- the extra N lanes form the merit of the
    // CURRENT iterate (bsqp.cuh:116-118: the line search's reference value and the `initial_merit` statistic) with the code and the
    // sum tree of merit_kernel<M, 1>, so a solve has no merit launch ahead of its loop
    extern __shared__ __attribute__((aligned(16))) float lds[];
    if (bf.ctrl->done) return;
    const int b = blockIdx.x, t = threadIdx.x;
    if (b == B) {   // one workgroup more than trajectories: the PCG launch plan runs in rounds and wants them hardest-first next time
        order_by_pcg_iters(bf, B, reinterpret_cast<int*>(lds));
        return;
    }
    step_body<M>(bf, N, B, b, t, (int)blockDim.x, dt, sqp_iter, (float)bf.num_solved[sqp_iter] >= thresh, adapt_rho, drho_init, last_iter, merit_init0, lds);
}

// =========================================================================================================================
// forward simulation of ONE shared (x_k, u_k) under B wrench hypotheses (sim.cuh:14-49): one lane per hypothesis
// =========================================================================================================================
template<class M>
__global__ __launch_bounds__(256) void sim_forward_kernel(float* __restrict__ xkp1, const float* __restrict__ xk, const float* __restrict__ uk,
                                                          const float* __restrict__ f_ext, int B, float dt)
{
    constexpr int NQ = M::NQ, NX = 2 * NQ;
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    float x[NX], u[NQ], fe[6], qdd[NQ];
#pragma unroll
    for (int i = 0; i < NX; i++) x[i] = xk[i];
#pragma unroll
    for (int i = 0; i < NQ; i++) u[i] = uk[i];
#pragma unroll
    for (int i = 0; i < 6; i++) fe[i] = f_ext[6 * b + i];
    RBD<M> d;
    d.set_q(x);
    d.forward_dynamics(x + NQ, u, fe, qdd);
#pragma unroll
    for (int i = 0; i < NQ; i++) {
        xkp1[(size_t)b * NX + NQ + i] = x[NQ + i] + dt * qdd[i];
        xkp1[(size_t)b * NX + i] = (float)((double)(x[i] + dt * x[NQ + i]) + 0.5 * (double)qdd[i] * (double)dt * (double)dt);
    }
}

// Hypothesis selection of the MPC loop in ONE launch (mpc_controller.py:294-309: sim_forward, per-hypothesis distance to the measured
// state, arg-min): lane b integrates the shared (x_last, u_last) under wrench b, leaves x_next_b and err_b = |x_next_b - x_meas|_2;
// the last workgroup to finish (device counter) takes the first minimum over the batch like np.argmin.
template<class M>
__global__ __launch_bounds__(256) void select_best_kernel(float* __restrict__ xkp1, float* __restrict__ err, int* __restrict__ best,
                                                          uint32_t* __restrict__ count, const float* __restrict__ xk, const float* __restrict__ uk,
                                                          const float* __restrict__ xmeas, const float* __restrict__ f_ext, int B, float dt)
{
    constexpr int NQ = M::NQ, NX = 2 * NQ;
    __shared__ float s_e[256];
    __shared__ int s_i[256];
    __shared__ bool s_last;
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b < B) {
        float x[NX], u[NQ], fe[6], qdd[NQ];
#pragma unroll
        for (int i = 0; i < NX; i++) x[i] = xk[i];
#pragma unroll
        for (int i = 0; i < NQ; i++) u[i] = uk[i];
#pragma unroll
        for (int i = 0; i < 6; i++) fe[i] = f_ext[6 * b + i];
        RBD<M> d;
        d.set_q(x);
        d.forward_dynamics(x + NQ, u, fe, qdd);
        float e2 = 0.f;
#pragma unroll
        for (int i = 0; i < NQ; i++) {
            const float qdn = x[NQ + i] + dt * qdd[i];
            const float qn = (float)((double)(x[i] + dt * x[NQ + i]) + 0.5 * (double)qdd[i] * (double)dt * (double)dt);
            xkp1[(size_t)b * NX + NQ + i] = qdn;
            xkp1[(size_t)b * NX + i] = qn;
            const float dq = qn - xmeas[i], dv = qdn - xmeas[NQ + i];
            e2 += dq * dq;
            e2 += dv * dv;
        }
        err[b] = sqrtf(e2);
    }
    __threadfence();
    __syncthreads();
    if (threadIdx.x == 0) s_last = (atomicAdd(count, 1u) == gridDim.x - 1);
    __syncthreads();
    if (!s_last) return;
    __threadfence();
    float be = 3.4e38f;
    int bi = 0x7fffffff;
    for (int i = threadIdx.x; i < B; i += blockDim.x) {
        const float e = __builtin_nontemporal_load(err + i);
        if (e < be) { be = e; bi = i; }   // ascending i per thread: the first minimum of its stride
    }
    s_e[threadIdx.x] = be;
    s_i[threadIdx.x] = bi;
    __syncthreads();
    for (int off = 128; off >= 1; off >>= 1) {
        if ((int)threadIdx.x < off) {
            const float e2 = s_e[threadIdx.x + off];
            const int i2 = s_i[threadIdx.x + off];
            if (e2 < s_e[threadIdx.x] || (e2 == s_e[threadIdx.x] && i2 < s_i[threadIdx.x])) { s_e[threadIdx.x] = e2; s_i[threadIdx.x] = i2; }
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        *best = s_i[0] == 0x7fffffff ? 0 : s_i[0];
        *count = 0;  // ready for the next call
    }
}

// one RK4 step of size h of the arm's forward dynamics under the control u and the wrench fe (common.py:49-91 `rk4`)
template<class M> GATO_DEV void plant_rk4_step(float* q, float* v, const float* u, const float* fe, float h)
{
    constexpr int NQ = M::NQ;
    const float hh = 0.5f * h;
    float k1v[NQ], k2v[NQ], k3v[NQ], k4v[NQ], k2q[NQ], k3q[NQ], k4q[NQ], qs[NQ];
    {
        RBD<M> d;
        d.set_q(q);
        d.forward_dynamics(v, u, fe, k1v);
    }
#pragma unroll
    for (int i = 0; i < NQ; i++) { qs[i] = q[i] + v[i] * hh; k2q[i] = v[i] + k1v[i] * hh; }
    {
        RBD<M> d;
        d.set_q(qs);
        d.forward_dynamics(k2q, u, fe, k2v);
    }
#pragma unroll
    for (int i = 0; i < NQ; i++) { qs[i] = q[i] + k2q[i] * hh; k3q[i] = v[i] + k2v[i] * hh; }
    {
        RBD<M> d;
        d.set_q(qs);
        d.forward_dynamics(k3q, u, fe, k3v);
    }
#pragma unroll
    for (int i = 0; i < NQ; i++) { qs[i] = q[i] + k3q[i] * h; k4q[i] = v[i] + k3v[i] * h; }
    {
        RBD<M> d;
        d.set_q(qs);
        d.forward_dynamics(k4q, u, fe, k4v);
    }
#pragma unroll
    for (int i = 0; i < NQ; i++) {
        const float avg = (v[i] + 2.f * k2q[i] + 2.f * k3q[i] + k4q[i]) / 6.f;
        const float vn = v[i] + (h / 6.f) * (k1v[i] + 2.f * k2v[i] + 2.f * k3v[i] + k4v[i]);
        q[i] = q[i] + avg * h;
        v[i] = vn;
    }
}

// ---- the swinging payload of the MPC plant ------------------------------------------------------------------------------------------
// MPC_GATO(pendulum_config=...) (mpc_controller.py:44-60, 340-360): the SIMULATED arm -- never the solver's model -- carries a pendulum: a
// spherical joint at the last joint frame (placement identity), a bob of `mass` at (0, 0, -length) of the joint frame with inertia
// `inertia` x 1 about its own centre (0.001 there), joint torque -damping x (relative angular velocity) (:472-478).  State beside the arm's:
// the joint rotation as a unit quaternion (x, y, z, w: pendulum -> last-link coordinates, pinocchio's JointModelSpherical) and the relative
// angular velocity in pendulum coordinates.
// The joint is eliminated the articulated-body way (Featherstone, RBDA ch. 7, leaf step), in pendulum coordinates with c = (0, 0, -l):
//   I_p = [[D, m c~], [m c~^T, m 1]], D = diag(ic + m l^2, ic + m l^2, ic);  S = [1; 0]  =>  U = I_p S, S^T U = D,
//   I_a = I_p - U D^-1 U^T = blkdiag(0, diag(mu, mu, m)), mu = m ic / (ic + m l^2)   (the bob resists only along the rod, and a little across),
//   p_a = p_A + I_a c_J + U D^-1 (tau - S^T p_A),  p_A = v_p x* I_p v_p,  c_J = v_p x S w.
// The arm then is the arm with I_a added to its last link (RBD::minv's payload argument) and p_a + I_a a_L as one more wrench on it, and
// afterwards  w' = D^-1 (tau - S^T p_A - U^T (X_p a_L + c_J)).  tests/pendulum_ref.py holds the same system un-eliminated in float64.
struct Payload {
    float mass, length, damping, inertia;
};
GATO_DEV void cross3(const float* a, const float* b, float* o)
{
    o[0] = a[1] * b[2] - a[2] * b[1];
    o[1] = a[2] * b[0] - a[0] * b[2];
    o[2] = a[0] * b[1] - a[1] * b[0];
}
// E = R(quat)^T: last-link coordinates -> pendulum coordinates
GATO_DEV void quat_to_E(const float* qt, float (*E)[3])
{
    const float x = qt[0], y = qt[1], z = qt[2], w = qt[3];
    E[0][0] = 1.f - 2.f * (y * y + z * z); E[1][0] = 2.f * (x * y - z * w);       E[2][0] = 2.f * (x * z + y * w);
    E[0][1] = 2.f * (x * y + z * w);       E[1][1] = 1.f - 2.f * (x * x + z * z); E[2][1] = 2.f * (y * z - x * w);
    E[0][2] = 2.f * (x * z - y * w);       E[1][2] = 2.f * (y * z + x * w);       E[2][2] = 1.f - 2.f * (x * x + y * y);
}
// pin.integrate on the spherical joint: quat (x) exp(w h), renormalised
GATO_DEV void quat_integrate(const float* qt, const float* w, float h, float* out)
{
    const float tx = w[0] * h, ty = w[1] * h, tz = w[2] * h;
    const float t2 = tx * tx + ty * ty + tz * tz;
    float sv, cw;   // exp(theta) = (sv theta, cw)
    if (t2 < 1e-8f) {
        sv = 0.5f - t2 / 48.f;
        cw = 1.f - t2 / 8.f;
    } else {
        const float t = sqrtf(t2);
        float sn, cs;
        sincosf(0.5f * t, &sn, &cs);
        sv = sn / t;
        cw = cs;
    }
    const float bx = sv * tx, by = sv * ty, bz = sv * tz;
    const float ax = qt[0], ay = qt[1], az = qt[2], aw = qt[3];
    float o[4];
    o[0] = aw * bx + cw * ax + (ay * bz - az * by);
    o[1] = aw * by + cw * ay + (az * bx - ax * bz);
    o[2] = aw * bz + cw * az + (ax * by - ay * bx);
    o[3] = aw * cw - (ax * bx + ay * by + az * bz);
    const float n = 1.f / sqrtf(o[0] * o[0] + o[1] * o[1] + o[2] * o[2] + o[3] * o[3]);
#pragma unroll
    for (int i = 0; i < 4; i++) out[i] = o[i] * n;
}
// accelerations of the arm (qdd) and of the pendulum (wd) at (q, quat), (qd, w), control u, wrench fe on the last link, joint torque taup
template<class M>
GATO_DEV void payload_dynamics(const float* q, const float* qd, const float* quat, const float* w, const float* u, const float* fe, const float* taup,
                               const Payload& pp, float* qdd, float* wd)
{
    constexpr int NQ = M::NQ, L = NQ - 1;
    RBD<M> d;
    d.set_q(q);
    float v[NQ][6], a[NQ][6], f[NQ][6];
    d.template rnea_fwd<0>(qd, nullptr, v, a);
    float E[3][3];
    quat_to_E(quat, E);
    const float m = pp.mass, hz = -pp.mass * pp.length;                       // h = m c = (0, 0, hz)
    const float Dx = pp.inertia + pp.mass * pp.length * pp.length, Dz = pp.inertia;
    const float lam[3] = {pp.mass * pp.inertia / Dx, pp.mass * pp.inertia / Dx, pp.mass};
    auto rot = [&](const float* x, float* o) {                                 // o = E x
#pragma unroll
        for (int r = 0; r < 3; r++) o[r] = E[r][0] * x[0] + E[r][1] * x[1] + E[r][2] * x[2];
    };
    auto rotT = [&](const float* x, float* o) {                                // o = E^T x
#pragma unroll
        for (int r = 0; r < 3; r++) o[r] = E[0][r] * x[0] + E[1][r] * x[1] + E[2][r] * x[2];
    };
    auto Ip_top = [&](const float* aw, const float* al, float* o) {            // S^T I_p [aw; al] = D aw + h x al
        o[0] = Dx * aw[0] - hz * al[1];
        o[1] = Dx * aw[1] + hz * al[0];
        o[2] = Dz * aw[2];
    };
    float vp[6], cj[6];
    rot(v[L], vp);
    rot(v[L] + 3, vp + 3);
    cross3(vp, w, cj);            // (E w_L + w) x w = (E w_L) x w
    cross3(vp + 3, w, cj + 3);
#pragma unroll
    for (int i = 0; i < 3; i++) vp[i] += w[i];
    float Iv[6], pA[6];
    Ip_top(vp, vp + 3, Iv);
    Iv[3] = m * vp[3] + hz * vp[1];   // m v - h x w
    Iv[4] = m * vp[4] - hz * vp[0];
    Iv[5] = m * vp[5];
    RBD<M>::fxv(vp, Iv, pA);
    float uj[3], y[3];
#pragma unroll
    for (int i = 0; i < 3; i++) uj[i] = taup[i] - pA[i];
    y[0] = uj[0] / Dx; y[1] = uj[1] / Dx; y[2] = uj[2] / Dz;
    // p_a: moment = taup (the joint carries nothing else), force = p_A + Lambda c_J + m c~^T y,  c~^T y = (0, 0, l) x y
    float pa[3], paL[6], lamL[3][3], fx[6];
    pa[0] = pA[3] + lam[0] * cj[3] + hz * y[1];
    pa[1] = pA[4] + lam[1] * cj[4] - hz * y[0];
    pa[2] = pA[5] + lam[2] * cj[5];
    rotT(taup, paL);
    rotT(pa, paL + 3);
#pragma unroll
    for (int r = 0; r < 3; r++)
#pragma unroll
        for (int c = 0; c < 3; c++) lamL[r][c] = E[0][r] * lam[0] * E[0][c] + E[1][r] * lam[1] * E[1][c] + E[2][r] * lam[2] * E[2][c];
#pragma unroll
    for (int r = 0; r < 3; r++) {
        fx[r] = fe[r] - paL[r];
        fx[3 + r] = fe[3 + r] - paL[3 + r] - (lamL[r][0] * a[L][3] + lamL[r][1] * a[L][4] + lamL[r][2] * a[L][5]);
    }
    d.template rnea_force<0>(v, a, f, fx);
    d.template rnea_bwd<NQ - 1>(f);
    {
        typename RBD<M>::MinvT Mi;
        d.minv(Mi, lamL);
        RBD<M>::fd_finish(Mi, u, f, qdd);
    }
    d.template rnea_fwd<0>(qd, qdd, v, a);
    float ap[6], t[3];
    rot(a[L], ap);
    rot(a[L] + 3, ap + 3);
#pragma unroll
    for (int i = 0; i < 6; i++) ap[i] += cj[i];
    Ip_top(ap, ap + 3, t);
    wd[0] = (uj[0] - t[0]) / Dx;
    wd[1] = (uj[1] - t[1]) / Dx;
    wd[2] = (uj[2] - t[2]) / Dz;
}
// one step of common.py:49-91 `rk4` on the arm + payload model: pin.integrate is q + v h on the revolute joints and quat (x) exp(w h) on the
// spherical one; the damping torque is formed once per step from the step's initial velocity, as the loop that calls rk4 does (mpc_controller.py:472-478)
template<class M> GATO_DEV void payload_rk4_step(float* q, float* v, float* pend, const float* u, const float* fe, const Payload& pp, float h)
{
    constexpr int NQ = M::NQ;
    const float hh = 0.5f * h;
    float* quat = pend;
    float* w = pend + 4;
    const float taup[3] = {-pp.damping * w[0], -pp.damping * w[1], -pp.damping * w[2]};
    float k1v[NQ], k2v[NQ], k3v[NQ], k4v[NQ], k2q[NQ], k3q[NQ], k4q[NQ], qs[NQ];
    float k1w[3], k2w[3], k3w[3], k4w[3], k2o[3], k3o[3], k4o[3], qts[4];
    payload_dynamics<M>(q, v, quat, w, u, fe, taup, pp, k1v, k1w);
#pragma unroll
    for (int i = 0; i < NQ; i++) { qs[i] = q[i] + v[i] * hh; k2q[i] = v[i] + k1v[i] * hh; }
    quat_integrate(quat, w, hh, qts);
#pragma unroll
    for (int i = 0; i < 3; i++) k2o[i] = w[i] + k1w[i] * hh;
    payload_dynamics<M>(qs, k2q, qts, k2o, u, fe, taup, pp, k2v, k2w);
#pragma unroll
    for (int i = 0; i < NQ; i++) { qs[i] = q[i] + k2q[i] * hh; k3q[i] = v[i] + k2v[i] * hh; }
    quat_integrate(quat, k2o, hh, qts);
#pragma unroll
    for (int i = 0; i < 3; i++) k3o[i] = w[i] + k2w[i] * hh;
    payload_dynamics<M>(qs, k3q, qts, k3o, u, fe, taup, pp, k3v, k3w);
#pragma unroll
    for (int i = 0; i < NQ; i++) { qs[i] = q[i] + k3q[i] * h; k4q[i] = v[i] + k3v[i] * h; }
    quat_integrate(quat, k3o, h, qts);
#pragma unroll
    for (int i = 0; i < 3; i++) k4o[i] = w[i] + k3w[i] * h;
    payload_dynamics<M>(qs, k4q, qts, k4o, u, fe, taup, pp, k4v, k4w);
#pragma unroll
    for (int i = 0; i < NQ; i++) {
        const float avg = (v[i] + 2.f * k2q[i] + 2.f * k3q[i] + k4q[i]) / 6.f;
        const float vn = v[i] + (h / 6.f) * (k1v[i] + 2.f * k2v[i] + 2.f * k3v[i] + k4v[i]);
        q[i] = q[i] + avg * h;
        v[i] = vn;
    }
    float avo[3];
#pragma unroll
    for (int i = 0; i < 3; i++) avo[i] = (w[i] + 2.f * k2o[i] + 2.f * k3o[i] + k4o[i]) / 6.f;
    quat_integrate(quat, avo, h, qts);
#pragma unroll
    for (int i = 0; i < 3; i++) w[i] = w[i] + (h / 6.f) * (k1w[i] + 2.f * k2w[i] + 2.f * k3w[i] + k4w[i]);
#pragma unroll
    for (int i = 0; i < 4; i++) quat[i] = qts[i];
}

// The plant of the closed MPC loop: `nsteps` RK4 steps of the arm's forward dynamics under a constant wrench on the last link, one
// control vector per step -- what python/bsqp/common.py:49-91 (`rk4`: k1..k4 from the articulated-body forward dynamics, revolute
// joints so pin.integrate is q + v h) does with pinocchio between two solves of MPC_GATO.run_mpc_fig8 (mpc_controller.py:199-218).
// Here the dynamics are the library's own (rbd.hpp, the tables the solver optimises with): one lane per plant instance.
template<class M>
__global__ __launch_bounds__(64) void plant_rk4_kernel(float* __restrict__ x_io, const float* __restrict__ u_seq, const float* __restrict__ f_ext,
                                                       int nsteps, float h, int nplants, float* __restrict__ pend = nullptr)
{
    constexpr int NQ = M::NQ, NX = 2 * NQ;
    const int p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= nplants) return;
    float q[NQ], v[NQ], fe[6];
#pragma unroll
    for (int i = 0; i < NQ; i++) { q[i] = x_io[(size_t)p * NX + i]; v[i] = x_io[(size_t)p * NX + NQ + i]; }
#pragma unroll
    for (int i = 0; i < 6; i++) fe[i] = f_ext[6 * p + i];
    if (pend) {   // pend[p]: [quat (4) | w (3) | mass, length, damping, inertia]: the arm carries a swinging payload
        float ps[7];
#pragma unroll
        for (int i = 0; i < 7; i++) ps[i] = pend[11 * p + i];
        const Payload pp = {pend[11 * p + 7], pend[11 * p + 8], pend[11 * p + 9], pend[11 * p + 10]};
        for (int s = 0; s < nsteps; s++) {
            float u[NQ];
#pragma unroll
            for (int i = 0; i < NQ; i++) u[i] = u_seq[((size_t)p * nsteps + s) * NQ + i];
            payload_rk4_step<M>(q, v, ps, u, fe, pp, h);
        }
#pragma unroll
        for (int i = 0; i < 7; i++) pend[11 * p + i] = ps[i];
    } else {
        for (int s = 0; s < nsteps; s++) {
            float u[NQ];
#pragma unroll
            for (int i = 0; i < NQ; i++) u[i] = u_seq[((size_t)p * nsteps + s) * NQ + i];
            plant_rk4_step<M>(q, v, u, fe, h);
        }
    }
#pragma unroll
    for (int i = 0; i < NQ; i++) { x_io[(size_t)p * NX + i] = q[i]; x_io[(size_t)p * NX + NQ + i] = v[i]; }
}

// end-effector positions of a batch of configurations (the facade's ee_pos; the reference uses pinocchio FK, interface.py:212-214)
template<class M>
__global__ __launch_bounds__(256) void ee_pos_kernel(float* __restrict__ out, const float* __restrict__ q, int n)
{
    constexpr int NQ = M::NQ;
    const int g = blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= n) return;
    float qq[NQ], e[3];
#pragma unroll
    for (int i = 0; i < NQ; i++) qq[i] = q[(size_t)g * NQ + i];
    RBD<M> d;
    d.set_q(qq);
    d.ee_pos(e);
    out[3 * g] = e[0]; out[3 * g + 1] = e[1]; out[3 * g + 2] = e[2];
}

// =========================================================================================================================
// MPC session (gato_mpc_begin / gato_mpc_step): the closed loop of python/bsqp/mpc_controller.py:196-242 with the measured state, the
// best trajectory and the batch iterates RESIDENT on the device -- one host call per MPC step enqueues, on one stream:
//   mpc_plant_kernel     the plant over the measured interval (RK4, the controls read from the best trajectory by knot index)
//   mpc_prepare_kernel   every row := best trajectory with its first state := the measured one; x_s, the reference window and the
//                        wrench hypotheses (world frame -> last joint frame, mpc_controller.py:311-338) broadcast / transformed per row
//   the SQP solve        (solver.hip:solve_impl, unchanged)
//   select_best_kernel   hypothesis selection (mpc_controller.py:294-309), then mpc_finish_kernel takes the winner and writes the record the host reads
// =========================================================================================================================
// rows := warm start (x0 repeated over the knots, zero controls: common.py:93-99), best := the same, state := x0
template<class M>
__global__ __launch_bounds__(256) void mpc_warm_kernel(float* __restrict__ xu, float* __restrict__ xu_best, float* __restrict__ x, float* __restrict__ x_last,
                                                       const float* __restrict__ x0, int traj, int B)
{
    constexpr int NX = 2 * M::NQ, KS = 3 * M::NQ;
    const int b = blockIdx.x;   // b == B: the best-trajectory buffer and the state
    float* dst = b < B ? xu + (size_t)b * traj : xu_best;
    for (int i = threadIdx.x; i < traj; i += blockDim.x) {
        const int o = i % KS;
        dst[i] = o < NX ? x0[o] : 0.f;
    }
    // x_last := x0 as well: a selection issued before any plant step compares against the state the session started from
    // (mpc_controller.py:196 takes x_last = x_curr at the top of every loop iteration), never against uninitialised memory
    if (b == B && threadIdx.x < NX) { x[threadIdx.x] = x0[threadIdx.x]; x_last[threadIdx.x] = x0[threadIdx.x]; }
}

// x_last := x (also when nsteps == 0: mpc_controller.py:196 takes x_last = x_curr at the top of EVERY loop iteration, so a step whose latency
// rounds to no plant step scores the hypotheses against the current state, not a stale one); x := the plant after nsteps RK4 steps of size h
// under the wrench fe6, step i driven by the control of knot
// min(int(i / steps_per_knot), N - 1) of the best trajectory (mpc_controller.py:199-218; the quotient in double like the Python it mirrors)
template<class M>
__global__ __launch_bounds__(64) void mpc_plant_kernel(float* __restrict__ x, float* __restrict__ x_last, const float* __restrict__ xu_best,
                                                       const float* __restrict__ fe6, int nsteps, float h, double steps_per_knot, int N,
                                                       float* __restrict__ pend)
{
    constexpr int NQ = M::NQ, NX = 2 * NQ, KS = 3 * NQ;
    if (blockIdx.x != 0 || threadIdx.x != 0) return;
    float q[NQ], v[NQ], fe[6], ps[7];
    Payload pp = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int i = 0; i < NQ; i++) { q[i] = x[i]; v[i] = x[NQ + i]; x_last[i] = q[i]; x_last[NQ + i] = v[i]; }
#pragma unroll
    for (int i = 0; i < 6; i++) fe[i] = fe6[i];
    if (pend) {   // the session's payload: [quat | w | mass, length, damping, inertia] (gato_mpc_set_payload)
#pragma unroll
        for (int i = 0; i < 7; i++) ps[i] = pend[i];
        pp = {pend[7], pend[8], pend[9], pend[10]};
    }
    for (int s = 0; s < nsteps; s++) {
        int k = (int)((double)s / steps_per_knot);
        k = k < N - 1 ? k : N - 1;
        float u[NQ];
#pragma unroll
        for (int i = 0; i < NQ; i++) u[i] = xu_best[(size_t)k * KS + NX + i];
        if (pend) payload_rk4_step<M>(q, v, ps, u, fe, pp, h);
        else plant_rk4_step<M>(q, v, u, fe, h);
    }
#pragma unroll
    for (int i = 0; i < NQ; i++) { x[i] = q[i]; x[NQ + i] = v[i]; }
    if (pend) {
#pragma unroll
        for (int i = 0; i < 7; i++) pend[i] = ps[i];
    }
}

// MPC_GATO.transform_force_to_gato_frame (mpc_controller.py:311-338) on the library's own kinematics: the world-frame wrench
// (linear fw[0:3], angular fw[3:6]) expressed in the last joint's frame -- SE3.actInv with that joint's world placement -- and then
// actInv of the joint's placement in its parent.  R_ee^T w = E_{n-1} ... E_0 w with E_k = Ez(q_k) E0_k (rbd.hpp), p_ee = the end effector,
// the local placement is (E_{n-1}^T, r_{n-1}).  Returned as [linear, angular], the order the reference hands to set_f_ext_batch.
template<class M, int K = 0> GATO_DEV void world_to_last_frame(const RBD<M>& d, float* w)
{
    if constexpr (K < M::NQ) {
        float t[3];
        d.template Emul<K>(w, t);
        w[0] = t[0]; w[1] = t[1]; w[2] = t[2];
        world_to_last_frame<M, K + 1>(d, w);
    }
}
template<class M> GATO_DEV void force_to_gato_frame(const float* q, const float* fw, float* out)
{
    constexpr int L = M::NQ - 1;
    RBD<M> d;
    d.set_q(q);
    float pe[3];
    d.ee_pos(pe);
    float lin[3] = {fw[0], fw[1], fw[2]};
    float ang[3] = {fw[3] - (pe[1] * fw[2] - pe[2] * fw[1]), fw[4] - (pe[2] * fw[0] - pe[0] * fw[2]), fw[5] - (pe[0] * fw[1] - pe[1] * fw[0])};
    world_to_last_frame<M>(d, lin);
    world_to_last_frame<M>(d, ang);
    float rl[3], t[3], lin2[3], ang2[3];
    d.template rcross<L>(lin, rl);                       // r_L x lin
    t[0] = ang[0] - rl[0]; t[1] = ang[1] - rl[1]; t[2] = ang[2] - rl[2];
    d.template Emul<L>(lin, lin2);
    d.template Emul<L>(t, ang2);
    out[0] = lin2[0]; out[1] = lin2[1]; out[2] = lin2[2];
    out[3] = ang2[0]; out[4] = ang2[1]; out[5] = ang2[2];
}

// one workgroup per row b: xu_b := best trajectory, its first state := x; x_s_b := x; ref_b := the window; f_ext_b := hypothesis b in the
// last joint's frame (hyp_world == nullptr: the stored wrenches stay)
template<class M>
__global__ __launch_bounds__(256) void mpc_prepare_kernel(float* __restrict__ xu, float* __restrict__ x_s, float* __restrict__ ref, float* __restrict__ f_ext,
                                                          const float* __restrict__ xu_best, const float* __restrict__ x, const float* __restrict__ refw,
                                                          const float* __restrict__ hyp_world, int N, int traj)
{
    constexpr int NQ = M::NQ, NX = 2 * NQ;
    const int b = blockIdx.x;
    for (int i = threadIdx.x; i < traj; i += blockDim.x) xu[(size_t)b * traj + i] = i < NX ? x[i] : xu_best[i];
    for (int i = threadIdx.x; i < 6 * N; i += blockDim.x) ref[(size_t)b * 6 * N + i] = refw[i];
    if (threadIdx.x < NX) x_s[(size_t)b * NX + threadIdx.x] = x[threadIdx.x];
    if (hyp_world != nullptr && threadIdx.x == 64) {   // a lane of the second wavefront: beside the copies, not behind them
        float q[NQ], fw[6], o[6];
#pragma unroll
        for (int i = 0; i < NQ; i++) q[i] = x[i];
#pragma unroll
        for (int i = 0; i < 6; i++) fw[i] = hyp_world[6 * b + i];
        force_to_gato_frame<M>(q, fw, o);
#pragma unroll
        for (int i = 0; i < 6; i++) f_ext[6 * b + i] = o[i];
    }
}

// sharded batch without a communicator (tests: gato_debug_set_remote_solved): the other shards' solved count of this iteration is given
// reset_dual() and reset_rho() of a stream-ordered caller (gato_reset_async) in ONE launch: lambda := 0 (n reals), rho / drho := their
// reset values.  Three runtime memset / copy kernels of ~5 us each sat between two solves of a loop that resets before every solve.
__global__ __launch_bounds__(256) void reset_kernel(float* __restrict__ lambda, uint32_t n, float* __restrict__ rho, const float* __restrict__ rho_init,
                                                    float* __restrict__ drho, const float* __restrict__ drho_init, int B)
{
    const uint32_t g = blockIdx.x * blockDim.x + threadIdx.x, T = gridDim.x * blockDim.x, n4 = n / 4;
    for (uint32_t i = g; i < n4; i += T) reinterpret_cast<real4*>(lambda)[i] = make_real4(0.f, 0.f, 0.f, 0.f);
    for (uint32_t i = 4 * n4 + g; i < n; i += T) lambda[i] = 0.f;
    if (rho)
        for (uint32_t i = g; i < (uint32_t)B; i += T) { rho[i] = rho_init[i]; drho[i] = drho_init[i]; }
}

// it >= 0: entry `it`; it < 0: the first -it entries (the deferred form: the whole count vector after the last iteration)
__global__ void add_remote_solved_kernel(uint32_t* __restrict__ global, const uint32_t* __restrict__ local, const uint32_t* __restrict__ remote, int it)
{
    if (blockIdx.x != 0) return;
    if (it >= 0) {
        if (threadIdx.x == 0) global[it] = local[it] + remote[it];
    } else {
        for (int i = threadIdx.x; i < -it; i += blockDim.x) global[i] = local[i] + remote[i];
    }
}

// what a solve changes for good -- the iterates, the duals, rho and drho -- copied aside in ONE launch at the start of a speculative (deferred-
// count) sharded solve, so that it can be re-run exactly if some iteration's whole-batch count turns out to have reached the exit threshold
__global__ __launch_bounds__(256) void snapshot_kernel(float* __restrict__ sx, const float* __restrict__ xu, uint32_t nxu, float* __restrict__ sl,
                                                       const float* __restrict__ lambda, uint32_t nl, float* __restrict__ sr, const float* __restrict__ rho,
                                                       const float* __restrict__ drho, uint32_t B)
{
    const uint32_t g = blockIdx.x * blockDim.x + threadIdx.x, T = gridDim.x * blockDim.x;
    // both arrays start 256-byte aligned (hipMalloc / torch allocations); tails by the float
    for (uint32_t i = g; i < nxu / 4; i += T) reinterpret_cast<real4*>(sx)[i] = reinterpret_cast<const real4*>(xu)[i];
    for (uint32_t i = 4 * (nxu / 4) + g; i < nxu; i += T) sx[i] = xu[i];
    for (uint32_t i = g; i < nl / 4; i += T) reinterpret_cast<real4*>(sl)[i] = reinterpret_cast<const real4*>(lambda)[i];
    for (uint32_t i = 4 * (nl / 4) + g; i < nl; i += T) sl[i] = lambda[i];
    for (uint32_t i = g; i < B; i += T) { sr[i] = rho[i]; sr[B + i] = drho[i]; }
}

// the verdict of a deferred sharded solve, handed to the host without a copy engine or an interrupt: the reduced per-iteration counts written straight
// into host-coherent pinned memory and, behind a system-scope release, the sequence number the host spins on (solver.hip: settle_impl)
__global__ __launch_bounds__(64) void publish_counts_kernel(uint32_t* __restrict__ host_counts, const uint32_t* __restrict__ counts, uint32_t n, uint32_t seq_at, uint32_t seq)
{
    for (uint32_t i = threadIdx.x; i < n; i += blockDim.x) __hip_atomic_store(&host_counts[i], counts[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    __threadfence_system();
    __syncthreads();
    if (threadIdx.x == 0) __hip_atomic_store(&host_counts[seq_at], seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}

// the end of a session step in ONE launch: best trajectory := row *best of the batch (the winner of select_best_kernel; row 0 without a
// selection; skipped when take == 0: an advance-only step), and the packed record the host reads back: [x (nx) | ee (3) | best]
template<class M>
__global__ __launch_bounds__(256) void mpc_finish_kernel(float* __restrict__ xu_best, const float* __restrict__ xu, const int32_t* __restrict__ best, int traj, int B,
                                                         int take, float* __restrict__ rec, const float* __restrict__ x)
{
    constexpr int NQ = M::NQ, NX = 2 * NQ;
    int w = (take && best) ? *best : 0;
    w = (w < 0 || w >= B) ? 0 : w;
    if (take)
        for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < traj; i += gridDim.x * blockDim.x) xu_best[i] = xu[(size_t)w * traj + i];
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        float q[NQ], e[3];
#pragma unroll
        for (int i = 0; i < NQ; i++) q[i] = x[i];
        RBD<M> d;
        d.set_q(q);
        d.ee_pos(e);
#pragma unroll
        for (int i = 0; i < NX; i++) rec[i] = x[i];
        rec[NX] = e[0]; rec[NX + 1] = e[1]; rec[NX + 2] = e[2];
        rec[NX + 3] = (float)w;
    }
}

}  // namespace gato
