/*
 * gato_oracle.c -- CPU restatement of the A2R-Lab/GATO batched-SQP hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  This file is the parity oracle and the host-core baseline of the repository.  Only
 * tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load it; the product (gato_amd/) never does.
 *
 * PARITY STATUS: "parity unpinned by reference execution".  The reference is CUDA-only (nvcc / CUDA runtime are not in
 * this image) and carries no tests, golden vectors or fixtures for this path, so the oracle cannot be
 * checked against outputs of the reference itself.  What pins it instead (tests/test_oracle_*.py):
 *   - the rigid-body constant tables equal the literals of the reference's init_XImats / load_update_* (test_robot_tables);
 *   - the dynamics obey the identities the reference's algorithms imply (M * Minv = 1 with M from RNEA columns, analytic
 *     gradients == central differences of this file's own forward dynamics, FK Jacobian == central differences);
 *   - the Python workload generators equal the importable Python half of the reference (the .npz files under tests/golden).
 * Every function cites the reference file:line it restates (paths relative to the reference root).
 *
 * Arithmetic: IEEE fp32 like the reference's `typedef float T` (gato/settings.h:7-11), with the double-typed literals of
 * the reference kept where they promote an expression (integrator.cuh:37,145), no fast-math.
 * Layouts: exactly the reference's trajectory-major batch layouts (gato/utils/linalg.cuh:545-672).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include "robot_tables.h"

#define NQMAX ORC_MAX_NQ
#define NXMAX (2 * NQMAX)
#define NUM_ALPHAS 8           /* gato/settings.h:16 */
#define RHO_FACTOR 1.2f        /* gato/settings.h:20 */
#define RHO_MIN 1e-8f          /* gato/settings.h:21 */
#define RHO_MAX 10.0f          /* gato/settings.h:22 */
#define RHO_INIT 1e-3f         /* gato/settings.h:18 */
#define GRAVITY 9.81f          /* gato/dynamics/indy7/indy7_plant.cuh:25-28 */

typedef struct {
    float dt;
    uint32_t max_sqp_iters;
    float kkt_tol;
    uint32_t max_pcg_iters;
    float pcg_tol, solve_ratio, mu, q_cost, qd_cost, u_cost, N_cost, q_lim_cost, vel_lim_cost, ctrl_lim_cost, rho;
} OrcParams;

typedef struct {
    const OrcModel* model;
    int nq, nx, nu, N, B;
    int traj, vecp, brow; /* TRAJ_SIZE, VEC_SIZE_PADDED, BLOCK_ROW_SIZE (gato/constants.h:21-24) */
    OrcParams p;
    int adapt_rho;
    /* persistent per-trajectory state (gato/bsqp/bsqp.cuh:299-327) */
    float *lambda, *rho, *drho, *rho_init, *drho_init, *mu, *pcg_tol, *f_ext;
    /* cost weights per trajectory, [B][7] = q, qd, u, N, q_lim, vel_lim, ctrl_lim (the reference has one scalar set per solver,
     * bsqp.cuh:344-350; SURVEY.md 8(f)3 generalises it so that a hyper-parameter sweep is one batch) */
    float* costw;
    /* KKT + Schur buffers (gato/types.cuh:63-81) */
    float *Q, *R, *q, *r, *A, *Bm, *c, *Qinv, *Rinv, *S, *Pinv, *gamma, *dz;
    float *merit, *merit_cur, *merit_init0, *step;
    float* pcg_work; /* [B][5 * vecp] */
    int32_t* converged;
    uint32_t* pcg_iters;
    /* stats of the last solve */
    uint32_t iters_done, ls_done;
    uint32_t (*shard_reduce)(uint32_t local_count, uint32_t sqp_iter, void* ctx); /* NULL: the batch is whole */
    void* shard_ctx;
    long shard_global_batch;
    int32_t* st_pcg_iters;  /* [max_sqp_iters][B] */
    float *st_min_merit, *st_step; /* [max_sqp_iters][B] */
    float *st_merits, *st_merit_before; /* [max_sqp_iters][B][NUM_ALPHAS], [max_sqp_iters][B]: what every line search chose from (tests) */
    uint32_t* sqp_iters;    /* [B] */
    int32_t* kkt_converged; /* [B] */
    int nthreads;
} Orc;

/* ------------------------------------------------------------------------------------------------------------------
 * spatial algebra helpers (gato/dynamics/indy7/indy7_grid.cuh:108-168, 336-402, 858-887)
 * ---------------------------------------------------------------------------------------------------------------- */
static void matvec6(float* out, const float* Mcol, const float* v) /* out = M v, M col-major 6x6 */
{
    for (int r = 0; r < 6; r++) {
        float s = 0.f;
        for (int c = 0; c < 6; c++) s += Mcol[6 * c + r] * v[c];
        out[r] = s;
    }
}
static void matTvec6(float* out, const float* Mcol, const float* v) /* out = M^T v */
{
    for (int r = 0; r < 6; r++) {
        float s = 0.f;
        for (int c = 0; c < 6; c++) s += Mcol[6 * r + c] * v[c];
        out[r] = s;
    }
}
/* column 2 of the motion cross-product matrix: v x S for S = e_z (indy7_grid.cuh:336-344) */
static void mx2(float* o, const float* v)
{
    o[0] = v[1]; o[1] = -v[0]; o[2] = 0.f; o[3] = v[4]; o[4] = -v[3]; o[5] = 0.f;
}
/* force cross product  v x* f  (indy7_grid.cuh:858-866) */
static void fx_times_v(float* o, const float* fx, const float* t)
{
    o[0] = -fx[2] * t[1] + fx[1] * t[2] - fx[5] * t[4] + fx[4] * t[5];
    o[1] = fx[2] * t[0] - fx[0] * t[2] + fx[5] * t[3] - fx[3] * t[5];
    o[2] = -fx[1] * t[0] + fx[0] * t[1] - fx[4] * t[3] + fx[3] * t[4];
    o[3] = -fx[2] * t[4] + fx[1] * t[5];
    o[4] = fx[2] * t[3] - fx[0] * t[5];
    o[5] = -fx[1] * t[3] + fx[0] * t[4];
}

/* X_k(q) as the reference's load_update_XImats_helpers builds it (indy7_grid.cuh:1597-1682): col-major 6x6,
 * upper-left = lower-right = E = Ez(q) E0, lower-left = -E r~ . */
static void build_X(const OrcModel* m, const float* q, float X[][36])
{
    for (int k = 0; k < m->nq; k++) {
        float s = sinf(q[k]), c = cosf(q[k]);
        const float* E0 = m->E0[k];
        const float* r = m->r[k];
        float E[3][3], L[3][3];
        for (int j = 0; j < 3; j++) {
            E[0][j] = c * E0[j] + s * E0[3 + j];
            E[1][j] = -s * E0[j] + c * E0[3 + j];
            E[2][j] = E0[6 + j];
        }
        float K[3][3] = {{0, -r[2], r[1]}, {r[2], 0, -r[0]}, {-r[1], r[0], 0}};
        for (int i = 0; i < 3; i++)
            for (int j = 0; j < 3; j++) {
                float a = 0.f;
                for (int t = 0; t < 3; t++) a += E[i][t] * K[t][j];
                L[i][j] = -a;
            }
        float* Xk = X[k];
        memset(Xk, 0, 36 * sizeof(float));
        for (int i = 0; i < 3; i++)
            for (int j = 0; j < 3; j++) {
                Xk[6 * j + i] = E[i][j];
                Xk[6 * (j + 3) + (i + 3)] = E[i][j];
                Xk[6 * j + (i + 3)] = L[i][j];
            }
    }
}

/* RNEA with the external wrench on the last link.  use_qdd = 0: inverse_dynamics_inner (indy7_fext.cuh:16-212), which also
 * extracts c; use_qdd = 1: inverse_dynamics_inner_vaf (indy7_fext.cuh:216-405).  v,a,f are [nq][6]; f comes back ACCUMULATED
 * (f_{k-1} += X_k^T f_k), which is what the gradient pass consumes. */
static void rnea(const OrcModel* m, float X[][36], const float* qd, const float* qdd, int use_qdd, const float* f_ext,
                 float v[][6], float a[][6], float f[][6], float* c)
{
    int nq = m->nq;
    for (int r = 0; r < 6; r++) {
        v[0][r] = 0.f;
        a[0][r] = X[0][30 + r] * GRAVITY; /* a_0 = X_0[:,5] * g */
    }
    v[0][2] += qd[0];
    if (use_qdd) a[0][2] += qdd[0];
    for (int k = 1; k < nq; k++) {
        matvec6(v[k], X[k], v[k - 1]);
        matvec6(a[k], X[k], a[k - 1]);
        v[k][2] += qd[k];
        if (use_qdd) a[k][2] += qdd[k];
        /* a += mx2(v) * qd  (mx2_peq_scaled, indy7_grid.cuh:396-402) */
        a[k][0] += v[k][1] * qd[k];
        a[k][1] += -v[k][0] * qd[k];
        a[k][3] += v[k][4] * qd[k];
        a[k][4] += -v[k][3] * qd[k];
    }
    for (int k = 0; k < nq; k++) {
        float Iv[6], t[6];
        matvec6(f[k], m->I[k], a[k]);
        matvec6(Iv, m->I[k], v[k]);
        fx_times_v(t, v[k], Iv);
        for (int r = 0; r < 6; r++) f[k][r] += t[r];
        if (k == nq - 1)
            for (int r = 0; r < 6; r++) f[k][r] -= f_ext[r]; /* indy7_fext.cuh:134-144 */
    }
    for (int k = nq - 1; k >= 1; k--) {
        float t[6];
        matTvec6(t, X[k], f[k]);
        for (int r = 0; r < 6; r++) f[k - 1][r] += t[r];
    }
    if (c)
        for (int k = 0; k < nq; k++) c[k] = f[k][2];
}

/* direct_minv_inner (indy7_grid.cuh:2918-3308): Carpentier's direct M^-1, upper triangle only, Minv[col*nq + row], row<=col. */
static void direct_minv(const OrcModel* m, float X[][36], float* Minv)
{
    int nq = m->nq;
    float IA[NQMAX][36], F[NQMAX][NQMAX][6], U[NQMAX][6], Dinv[NQMAX];
    memset(F, 0, sizeof(F));
    for (int i = 0; i < nq * nq; i++) Minv[i] = 0.f;
    for (int k = 0; k < nq; k++) memcpy(IA[k], m->I[k], 36 * sizeof(float));
    for (int k = nq - 1; k >= 0; k--) {
        for (int r = 0; r < 6; r++) U[k][r] = IA[k][12 + r]; /* U = IA S */
        Dinv[k] = 1.0f / U[k][2];
        Minv[k * nq + k] = Dinv[k];
        for (int j = k; j < nq; j++) {
            Minv[j * nq + k] -= Dinv[k] * F[k][j][2];
            if (k > 0)
                for (int r = 0; r < 6; r++) F[k][j][r] += U[k][r] * Minv[j * nq + k];
        }
        if (k > 0) {
            float Ia[36], T[36];
            for (int i = 0; i < 36; i++) Ia[i] = IA[k][i] - (U[k][i % 6] * Dinv[k] * U[k][i / 6]);
            for (int j = k; j < nq; j++) matTvec6(F[k - 1][j], X[k], F[k][j]); /* F_parent = X^T F */
            for (int col = 0; col < 6; col++) matTvec6(&T[6 * col], X[k], &Ia[6 * col]); /* T = X^T Ia */
            for (int col = 0; col < 6; col++)
                for (int row = 0; row < 6; row++) {
                    float s = 0.f;
                    for (int t = 0; t < 6; t++) s += T[6 * t + row] * X[k][6 * col + t];
                    IA[k - 1][6 * col + row] += s; /* IA_parent += (X^T Ia) X */
                }
        }
    }
    /* forward pass (indy7_grid.cuh:3202-3307) */
    for (int j = 0; j < nq; j++)
        for (int r = 0; r < 6; r++) F[0][j][r] = (r == 2) ? Minv[j * nq + 0] : 0.f;
    for (int k = 1; k < nq; k++) {
        for (int j = k; j < nq; j++) {
            matvec6(F[k][j], X[k], F[k - 1][j]);
            float d = 0.f;
            for (int r = 0; r < 6; r++) d += F[k][j][r] * U[k][r];
            Minv[j * nq + k] -= Dinv[k] * d;
            F[k][j][2] += Minv[j * nq + k];
        }
    }
}

static float minv_sym(const float* Minv, int nq, int row, int col)
{
    return (row <= col) ? Minv[col * nq + row] : Minv[row * nq + col]; /* indy7_grid.cuh:3328 */
}

/* inverse_dynamics_gradient_inner (indy7_grid.cuh:3373-3774): dc_du = [dc/dq | dc/dqd], each nq x nq col-major. */
static void rnea_gradient(const OrcModel* m, float X[][36], const float* qd, float v[][6], float a[][6], float f[][6], float* dc_du)
{
    int nq = m->nq;
    float Iv[NQMAX][6], MxXv[NQMAX][6], MxXa[NQMAX][6], Mxv[NQMAX][6], Mxf[NQMAX][6], FxvI[NQMAX][36], XTmxf[NQMAX][6];
    static const float S2[6] = {0, 0, 1, 0, 0, 0};
    for (int k = 0; k < nq; k++) {
        float Xv[6], Xa[6];
        matvec6(Iv[k], m->I[k], v[k]);
        if (k == 0) {
            for (int r = 0; r < 6; r++) { Xv[r] = 0.f; Xa[r] = X[0][30 + r] * GRAVITY; }
        } else {
            matvec6(Xv, X[k], v[k - 1]);
            matvec6(Xa, X[k], a[k - 1]);
        }
        mx2(MxXv[k], Xv); mx2(MxXa[k], Xa); mx2(Mxv[k], v[k]); mx2(Mxf[k], f[k]);
        for (int col = 0; col < 6; col++) fx_times_v(&FxvI[k][6 * col], v[k], &m->I[k][6 * col]);
        matTvec6(XTmxf[k], X[k], Mxf[k]);
        for (int r = 0; r < 6; r++) XTmxf[k][r] = -XTmxf[k][r];
    }
    for (int which = 0; which < 2; which++) { /* 0: d/dq, 1: d/dqd */
        float dv[NQMAX][NQMAX][6], da[NQMAX][NQMAX][6], df[NQMAX][NQMAX][6];
        memset(df, 0, sizeof(df));
        for (int i = 0; i < nq; i++)
            for (int j = 0; j <= i; j++) {
                if (j == i) {
                    for (int r = 0; r < 6; r++) dv[i][j][r] = which == 0 ? (i == 0 ? 0.f : MxXv[i][r]) : S2[r];
                } else {
                    matvec6(dv[i][j], X[i], dv[i - 1][j]);
                }
            }
        for (int i = 0; i < nq; i++)
            for (int j = 0; j <= i; j++) {
                mx2(da[i][j], dv[i][j]);
                for (int r = 0; r < 6; r++) da[i][j][r] *= qd[i];
                if (j == i)
                    for (int r = 0; r < 6; r++) da[i][j][r] += which == 0 ? MxXa[i][r] : Mxv[i][r];
            }
        for (int i = 1; i < nq; i++)
            for (int j = 0; j < i; j++) {
                float t[6];
                matvec6(t, X[i], da[i - 1][j]);
                for (int r = 0; r < 6; r++) da[i][j][r] += t[r];
            }
        for (int i = 0; i < nq; i++)
            for (int j = 0; j <= i; j++) {
                float t1[6], t2[6];
                fx_times_v(df[i][j], dv[i][j], Iv[i]);
                matvec6(t1, m->I[i], da[i][j]);
                matvec6(t2, FxvI[i], dv[i][j]);
                for (int r = 0; r < 6; r++) df[i][j][r] += t1[r] + t2[r];
            }
        for (int i = nq - 1; i >= 1; i--)
            for (int j = 0; j < nq; j++) {
                float t[6];
                matTvec6(t, X[i], df[i][j]);
                for (int r = 0; r < 6; r++) df[i - 1][j][r] += t[r] + ((which == 0 && j == i) ? XTmxf[i][r] : 0.f);
            }
        for (int i = 0; i < nq; i++)
            for (int j = 0; j < nq; j++) dc_du[which * nq * nq + j * nq + i] = df[i][j][2];
    }
}

/* plant::forwardDynamics(..., d_f_ext) (indy7_plant.cuh:163-173 -> forward_dynamics_inner indy7_fext.cuh:408-414) */
static void forward_dynamics(const OrcModel* m, const float* q, const float* qd, const float* u, const float* f_ext, float* qdd)
{
    int nq = m->nq;
    float X[NQMAX][36], Minv[NQMAX * NQMAX], c[NQMAX], v[NQMAX][6], a[NQMAX][6], f[NQMAX][6];
    build_X(m, q, X);
    direct_minv(m, X, Minv);
    rnea(m, X, qd, NULL, 0, f_ext, v, a, f, c);
    for (int row = 0; row < nq; row++) { /* forward_dynamics_finish, indy7_grid.cuh:3322-3334 */
        float val = 0.f;
        for (int col = 0; col < nq; col++) val += minv_sym(Minv, nq, row, col) * (u[col] - c[col]);
        qdd[row] = val;
    }
}

/* plant::forwardDynamicsAndGradient(..., d_f_ext) (indy7_plant.cuh:220-259): dqdd = [dqdd/dq | dqdd/dqd | Minv], nq x 3nq col-major */
static void forward_dynamics_and_gradient(const OrcModel* m, const float* q, const float* qd, const float* u, const float* f_ext,
                                          float* qdd, float* dqdd)
{
    int nq = m->nq;
    float X[NQMAX][36], Minv[NQMAX * NQMAX], c[NQMAX], v[NQMAX][6], a[NQMAX][6], f[NQMAX][6], dc_du[2 * NQMAX * NQMAX];
    build_X(m, q, X);
    direct_minv(m, X, Minv);
    rnea(m, X, qd, NULL, 0, f_ext, v, a, f, c);
    for (int row = 0; row < nq; row++) {
        float val = 0.f;
        for (int col = 0; col < nq; col++) val += minv_sym(Minv, nq, row, col) * (u[col] - c[col]);
        qdd[row] = val;
    }
    rnea(m, X, qd, qdd, 1, f_ext, v, a, f, NULL);
    rnea_gradient(m, X, qd, v, a, f, dc_du);
    for (int ind = 0; ind < 2 * nq * nq; ind++) {
        int row = ind % nq, off = ind - row;
        float val = 0.f;
        for (int col = 0; col < nq; col++) val += minv_sym(Minv, nq, row, col) * dc_du[off + col];
        dqdd[ind] = -val;
        if (ind < nq * nq) dqdd[ind + 2 * nq * nq] = minv_sym(Minv, nq, row, ind / nq);
    }
}

/* end_effector_positions_inner / _gradient_inner (indy7_grid.cuh:1834-1901, 1933-2025): xyz of the chained homogeneous
 * transforms and its Jacobian columns, J[3*j + xyz].  Full 4x4 products, right to left, like the reference. */
static void build_Xhom(const OrcModel* m, const float* q, float Xh[][16], float dXh[][16])
{
    for (int k = 0; k < m->nq; k++) {
        float s = sinf(q[k]), c = cosf(q[k]);
        const float* E0 = m->E0[k];
        memset(Xh[k], 0, 16 * sizeof(float));
        memset(dXh[k], 0, 16 * sizeof(float));
        for (int j = 0; j < 3; j++) {
            /* R = E^T : R[j][0] = E[0][j], ... ; col-major 4x4 index = 4*col + row */
            Xh[k][4 * 0 + j] = c * E0[j] + s * E0[3 + j];
            Xh[k][4 * 1 + j] = -s * E0[j] + c * E0[3 + j];
            Xh[k][4 * 2 + j] = E0[6 + j];
            dXh[k][4 * 0 + j] = -s * E0[j] + c * E0[3 + j];
            dXh[k][4 * 1 + j] = -c * E0[j] - s * E0[3 + j];
            Xh[k][12 + j] = m->r[k][j];
        }
        Xh[k][15] = 1.f;
    }
}
static void matmul4(float* out, const float* A, const float* Bm)
{
    for (int col = 0; col < 4; col++)
        for (int row = 0; row < 4; row++) {
            float s = 0.f;
            for (int t = 0; t < 4; t++) s += A[4 * t + row] * Bm[4 * col + t];
            out[4 * col + row] = s;
        }
}
static void ee_pos(const OrcModel* m, const float* q, float* e, float* J /* may be NULL; [3*nq] */)
{
    int nq = m->nq;
    float Xh[NQMAX][16], dXh[NQMAX][16], T[16], T2[16];
    build_Xhom(m, q, Xh, dXh);
    memcpy(T, Xh[nq - 1], sizeof(T));
    for (int k = nq - 2; k >= 0; k--) { matmul4(T2, Xh[k], T); memcpy(T, T2, sizeof(T)); }
    e[0] = T[12]; e[1] = T[13]; e[2] = T[14];
    if (!J) return;
    for (int j = 0; j < nq; j++) {
        memcpy(T, (j == nq - 1) ? dXh[nq - 1] : Xh[nq - 1], sizeof(T));
        for (int k = nq - 2; k >= 0; k--) { matmul4(T2, (k == j) ? dXh[k] : Xh[k], T); memcpy(T, T2, sizeof(T)); }
        J[3 * j + 0] = T[12]; J[3 * j + 1] = T[13]; J[3 * j + 2] = T[14];
    }
}

/* ------------------------------------------------------------------------------------------------------------------
 * cost (gato/dynamics/indy7/indy7_plant.cuh:130-148, 266-447; iiwa14_plant.cuh:104-155, 339-450)
 * ---------------------------------------------------------------------------------------------------------------- */
static float joint_barrier(float q, float lo, float hi) /* indy7_plant.cuh:130-138 (same in iiwa14_plant.cuh) */
{
    float dmin = q - lo, dmax = hi - q;
    dmin = (dmin <= 1e-10) ? (float)1e-10 : dmin;
    dmax = (dmax <= 1e-10) ? (float)1e-10 : dmax;
    return -logf(dmin) - logf(dmax);
}
static float joint_barrier_grad(int mode, float q, float lo, float hi)
{
    float dmin = q - lo, dmax = hi - q;
    if (mode == 0) { /* indy7_plant.cuh:140-148 */
        dmin = (dmin <= 1e-6) ? (float)1e-6 : dmin;
        dmax = (dmax <= 1e-6) ? (float)1e-6 : dmax;
        return (-1 / dmin) + (1 / dmax);
    }
    const float eps = 1e-6f; /* iiwa14_plant.cuh:114-135 */
    if (dmin >= 0.f) { if (dmin < eps) dmin = eps; } else { if (dmin > -eps) dmin = -eps; }
    if (dmax >= 0.f) { if (dmax < eps) dmax = eps; } else { if (dmax > -eps) dmax = -eps; }
    return (-1.0f / dmin) + (1.0f / dmax);
}
static float joint_barrier_hess(float q, float lo, float hi) /* iiwa14_plant.cuh:141-155 */
{
    float dmin = q - lo, dmax = hi - q;
    const float eps = 1e-6f;
    float amin = dmin >= 0.f ? dmin : -dmin, amax = dmax >= 0.f ? dmax : -dmax;
    if (amin < eps) amin = eps;
    if (amax < eps) amax = eps;
    return 1.0f / (amin * amin) + 1.0f / (amax * amax);
}

/* plant::trackingcost (indy7_plant.cuh:266-318): xu = [q, qd, (u)], has_u = (knot < N-1), terminal selects N_cost */
typedef struct { float q_cost, qd_cost, u_cost, N_cost, q_lim_cost, vel_lim_cost, ctrl_lim_cost; } OrcCosts;

static float tracking_cost(const Orc* o, const OrcCosts* p, const float* xu, const float* ref, int has_u, int terminal)
{
    const OrcModel* m = o->model;
    int nq = o->nq;
    float e[3], cost = 0.f, terms[3 * NQMAX + 3];
    int n = 0;
    ee_pos(m, xu, e, NULL);
    for (int i = 0; i < nq; i++) {
        float err = xu[i + nq];
        float t = 0.5f * p->qd_cost * err * err;
        t += p->q_lim_cost * joint_barrier(xu[i], m->q_lim[i][0], m->q_lim[i][1]);
        t += p->vel_lim_cost * joint_barrier(xu[i + nq], m->v_lim[i][0], m->v_lim[i][1]);
        terms[n++] = t;
    }
    if (has_u)
        for (int i = 0; i < nq; i++) {
            float err = xu[2 * nq + i];
            float t = 0.5f * p->u_cost * err * err;
            t += p->ctrl_lim_cost * joint_barrier(err, m->u_lim[i][0], m->u_lim[i][1]);
            terms[n++] = t;
        }
    for (int i = 0; i < 3; i++) {
        float err = e[i] - ref[i];
        terms[n++] = (float)(0.5 * (double)(terminal ? p->N_cost : p->q_cost) * (double)err * (double)err); /* `0.5 * N_cost * err * err`: double literal */
    }
    for (int i = 0; i < n; i++) cost += terms[i]; /* block::reduce, linalg.cuh:329-353 (tree order not restated) */
    return cost;
}

/* plant::trackingCostGradientAndHessian<computeR> (indy7_plant.cuh:325-421, iiwa14_plant.cuh:339-424).  Q col-major nx x nx.
 * The `blockIdx.x == KNOT_POINTS-1` selector of the reference is never true where this runs: weight = q_cost. */
static void tracking_cost_grad_hess(const Orc* o, const OrcCosts* p, const float* xu, const float* ref, float* Q, float* qv, float* R, float* rv)
{
    const OrcModel* m = o->model;
    int nq = o->nq, nx = o->nx;
    int mode = m->barrier_mode;
    float e[3], J[3 * NQMAX], g[NQMAX];
    ee_pos(m, xu, e, J);
    float w = p->q_cost;
    for (int i = 0; i < nq; i++) {
        g[i] = (J[3 * i + 0] * (e[0] - ref[0]) + J[3 * i + 1] * (e[1] - ref[1]) + J[3 * i + 2] * (e[2] - ref[2]));
        qv[i] = g[i] * w;
        qv[i] += p->q_lim_cost * joint_barrier_grad(mode, xu[i], m->q_lim[i][0], m->q_lim[i][1]);
        qv[nq + i] = p->qd_cost * xu[nq + i];
        qv[nq + i] += p->vel_lim_cost * joint_barrier_grad(mode, xu[nq + i], m->v_lim[i][0], m->v_lim[i][1]);
    }
    for (int i = 0; i < nx; i++)
        for (int j = 0; j < nx; j++) {
            float val;
            if (i < nq && j < nq) {
                val = (g[i] * g[j]) * w;
                if (mode == 0) {
                    float bi = joint_barrier_grad(0, xu[i], m->q_lim[i][0], m->q_lim[i][1]);
                    float bj = joint_barrier_grad(0, xu[j], m->q_lim[j][0], m->q_lim[j][1]);
                    val += p->q_lim_cost * bi * bj;
                } else if (i == j) {
                    val += p->q_lim_cost * joint_barrier_hess(xu[i], m->q_lim[i][0], m->q_lim[i][1]);
                }
            } else {
                val = (i == j) ? p->qd_cost : 0.f;
                if (i == j) {
                    if (mode == 0) {
                        float b = joint_barrier_grad(0, xu[i], m->v_lim[i - nq][0], m->v_lim[i - nq][1]);
                        val += p->vel_lim_cost * b * b;
                    } else {
                        val += p->vel_lim_cost * joint_barrier_hess(xu[i], m->v_lim[i - nq][0], m->v_lim[i - nq][1]);
                    }
                }
            }
            Q[i * nx + j] = val;
        }
    if (R) {
        for (int i = 0; i < nq; i++) {
            float uu = xu[nx + i];
            rv[i] = p->u_cost * uu;
            rv[i] += p->ctrl_lim_cost * joint_barrier_grad(mode, uu, m->u_lim[i][0], m->u_lim[i][1]);
            for (int j = 0; j < nq; j++) {
                float val = (i == j) ? p->u_cost : 0.f;
                if (i == j) {
                    if (mode == 0) {
                        float b = joint_barrier_grad(0, uu, m->u_lim[i][0], m->u_lim[i][1]);
                        val += p->ctrl_lim_cost * b * b;
                    } else {
                        val += p->ctrl_lim_cost * joint_barrier_hess(uu, m->u_lim[i][0], m->u_lim[i][1]);
                    }
                }
                R[i * nq + j] = val;
            }
        }
    }
}

/* ------------------------------------------------------------------------------------------------------------------
 * integrator (gato/dynamics/integrator.cuh), INTEGRATOR_TYPE = 2, ANGLE_WRAP = false
 * ---------------------------------------------------------------------------------------------------------------- */
static void integrate(int nq, float* xn, const float* q, const float* qd, const float* qdd, float dt) /* integrator.cuh:34-37 */
{
    for (int i = 0; i < nq; i++) {
        xn[nq + i] = qd[i] + dt * qdd[i];
        xn[i] = (float)((double)(q[i] + dt * qd[i]) + 0.5 * (double)qdd[i] * (double)dt * (double)dt);
    }
}
/* integrator_gradient_inner (integrator.cuh:143-184) */
static void integrator_gradient(int nq, float* A, float* Bm, const float* dqdd, float dt)
{
    int nx = 2 * nq;
    const float dt_sq_half = (float)(0.5 * (double)dt * (double)dt);
    for (int i = 0; i < nx * nx; i++) {
        int c = i / nx, r = i % nx, rd = r % nq;
        float d = dqdd[c * nq + rd];
        float val = (r == c) ? 1.0f : 0.0f;
        if (r < nq) {
            if (c >= nq && r == (c - nq)) val += dt;
            val += dt_sq_half * d;
        } else {
            val += dt * d;
        }
        A[i] = val;
    }
    for (int i = 0; i < nx * nq; i++) {
        int c = i / nx, r = i % nx, rd = r % nq;
        float d = dqdd[nx * nq + c * nq + rd];
        Bm[i] = (r < nq) ? dt_sq_half * d : dt * d;
    }
}

/* ------------------------------------------------------------------------------------------------------------------
 * layout helpers (gato/utils/linalg.cuh:545-672)
 * ---------------------------------------------------------------------------------------------------------------- */
#define XU(o, b, k) ((o)->xu_ptr + (size_t)(b) * (o)->traj + (size_t)(k) * ((o)->nx + (o)->nu))

/* setupKKTSystemBatchedKernel (gato/bsqp/kernels/setup_kkt.cuh:15-108) for one trajectory */
static void setup_kkt_one(Orc* o, int b, const float* xu, const float* x_s, const float* ref, float dt)
{
    int nq = o->nq, nx = o->nx, nu = o->nu, N = o->N;
    const float* fe = o->f_ext + 6 * b;
    float* Q = o->Q + (size_t)b * nx * nx * N;
    float* R = o->R + (size_t)b * nu * nu * N;
    float* qv = o->q + (size_t)b * nx * N;
    float* rv = o->r + (size_t)b * nu * N;
    float* A = o->A + (size_t)b * nx * nx * N;
    float* Bm = o->Bm + (size_t)b * nx * nu * N;
    float* c = o->c + (size_t)b * nx * N;
    const float* xub = xu + (size_t)b * o->traj;
    const float* refb = ref + (size_t)b * 6 * N;
    for (int k = 0; k < N - 1; k++) {
        const float* xk = xub + k * (nx + nu);
        const float* xn = xk + nx + nu;
        float qdd[NQMAX], dqdd[3 * NQMAX * NQMAX], xnew[NXMAX];
        forward_dynamics_and_gradient(o->model, xk, xk + nq, xk + nx, fe, qdd, dqdd);
        integrate(nq, xnew, xk, xk + nq, qdd, dt);
        for (int i = 0; i < nx; i++) c[(k + 1) * nx + i] = xn[i] - xnew[i]; /* integrator_error_inner, integrator.cuh:48-62 */
        integrator_gradient(nq, A + k * nx * nx, Bm + k * nx * nu, dqdd, dt);
        tracking_cost_grad_hess(o, (const OrcCosts*)(o->costw + 7 * (size_t)b), xk, refb + 6 * k, Q + k * nx * nx, qv + k * nx, R + k * nu * nu, rv + k * nu);
        if (k == N - 2) {
            /* _lastblock: terminal blocks at x_{N-2} against ref_{N-1} (indy7_plant.cuh:423-447) */
            tracking_cost_grad_hess(o, (const OrcCosts*)(o->costw + 7 * (size_t)b), xk, refb + 6 * (k + 1), Q + (k + 1) * nx * nx, qv + (k + 1) * nx, NULL, NULL);
            for (int i = 0; i < nx; i++) c[i] = xub[i] - x_s[b * nx + i]; /* setup_kkt.cuh:92-95 */
        }
    }
}

/* Gauss-Jordan inverse without pivoting on [V | I] (block::invertMatrix, gato/utils/linalg.cuh:364-519); the arithmetic of
 * the 2-/3-matrix form (`a / p * row`) is used for Q_k, Q_k+1, R_k and the 1-matrix form (`a * (1/p) * row`) for theta, as
 * the reference does.  V and Vinv col-major n x n, V is destroyed. */
static void gj_inverse(int n, float* V, float* Vinv, int one_matrix_form)
{
    float aug[NXMAX * 2 * NXMAX];
    for (int i = 0; i < n * n; i++) aug[i] = V[i];
    for (int i = 0; i < n * n; i++) aug[n * n + i] = ((i / n) == (i % n)) ? 1.f : 0.f;
    for (int p = 0; p < n; p++) {
        float colv[NXMAX], rowv[NXMAX + 1];
        for (int i = 0; i < n; i++) colv[i] = aug[p * n + i];
        for (int j = 0; j <= n; j++) rowv[j] = aug[(p + j) * n + p];
        float pv = colv[p], pvInv = 1.0f / pv;
        for (int j = 0; j <= n; j++)
            for (int r = 0; r < n; r++) {
                float* x = &aug[(p + j) * n + r];
                if (one_matrix_form) {
                    if (r == p) *x *= pvInv; else *x -= colv[r] * pvInv * rowv[j];
                } else {
                    if (r == p) *x /= pv; else *x -= colv[r] / pv * rowv[j];
                }
            }
    }
    for (int i = 0; i < n * n; i++) Vinv[i] = aug[n * n + i];
}

static void add_rho(int nx, float* M, float rho) /* block::addScaledIdentity: first nx/2 diagonal entries only (linalg.cuh:84-96) */
{
    for (int i = 0; i < nx / 2; i++) M[i * nx + i] += rho;
}

/* formSchurSystemBatchedKernel1/2 (gato/bsqp/kernels/schur_linsys.cuh:14-260) for one trajectory.  Reads the un-inverted,
 * un-regularised Q,R of setup_kkt (the intended semantics of SURVEY.md A.16); inverses go to Qinv/Rinv. */
static void form_schur_one(Orc* o, int b)
{
    int nx = o->nx, nu = o->nu, N = o->N, br = 3 * nx;
    float rho = o->rho[b];
    const float* Q = o->Q + (size_t)b * nx * nx * N;
    const float* R = o->R + (size_t)b * nu * nu * N;
    const float* qv = o->q + (size_t)b * nx * N;
    const float* rv = o->r + (size_t)b * nu * N;
    const float* A = o->A + (size_t)b * nx * nx * N;
    const float* Bm = o->Bm + (size_t)b * nx * nu * N;
    const float* c = o->c + (size_t)b * nx * N;
    float* Qinv = o->Qinv + (size_t)b * nx * nx * N;
    float* Rinv = o->Rinv + (size_t)b * nu * nu * N;
    float* S = o->S + (size_t)b * o->brow * N;
    float* P = o->Pinv + (size_t)b * o->brow * N;
    float* gam = o->gamma + (size_t)b * o->vecp;
    for (int k = 0; k < N - 1; k++) {
        float Qk[NXMAX * NXMAX], Qk1[NXMAX * NXMAX], Rk[NQMAX * NQMAX], Qki[NXMAX * NXMAX], Qk1i[NXMAX * NXMAX], Rki[NQMAX * NQMAX];
        float phi[NXMAX * NXMAX], BR[NXMAX * NQMAX], theta[NXMAX * NXMAX], thinv[NXMAX * NXMAX], g[NXMAX];
        const float *Ak = A + k * nx * nx, *Bk = Bm + k * nx * nu;
        memcpy(Qk, Q + k * nx * nx, nx * nx * sizeof(float));
        memcpy(Qk1, Q + (k + 1) * nx * nx, nx * nx * sizeof(float));
        memcpy(Rk, R + k * nu * nu, nu * nu * sizeof(float));
        add_rho(nx, Qk, rho);
        add_rho(nx, Qk1, rho);
        gj_inverse(nx, Qk, Qki, 0);
        gj_inverse(nx, Qk1, Qk1i, 0);
        gj_inverse(nu, Rk, Rki, 0);
        memcpy(Qinv + k * nx * nx, Qki, nx * nx * sizeof(float));
        memcpy(Rinv + k * nu * nu, Rki, nu * nu * sizeof(float));
        if (k == N - 2) memcpy(Qinv + (k + 1) * nx * nx, Qk1i, nx * nx * sizeof(float));
        for (int i = 0; i < nx * nx; i++) { /* phi = A Qinv */
            int y = i % nx, x = i / nx;
            float s = 0.f;
            for (int j = 0; j < nx; j++) s += Ak[j * nx + y] * Qki[x * nx + j];
            phi[i] = s;
        }
        for (int i = 0; i < nx * nu; i++) { /* B Rinv */
            int y = i % nx, x = i / nx;
            float s = 0.f;
            for (int j = 0; j < nu; j++) s += Bk[j * nx + y] * Rki[x * nu + j];
            BR[i] = s;
        }
        for (int i = 0; i < nx * nx; i++) { /* theta = Qk1inv + phi A^T + BR B^T */
            int y = i % nx, x = i / nx;
            float s = 0.f, s2 = 0.f;
            for (int j = 0; j < nx; j++) s += phi[j * nx + y] * Ak[j * nx + x];
            for (int j = 0; j < nu; j++) s2 += BR[j * nx + y] * Bk[j * nx + x];
            theta[i] = Qk1i[i];
            theta[i] += s;
            theta[i] += s2;
        }
        for (int y = 0; y < nx; y++) { /* gamma (schur_linsys.cuh:81,121-128) */
            float gg = -1.0f * c[(k + 1) * nx + y];
            float s = 0.f;
            for (int j = 0; j < nx; j++) s += Qk1i[j * nx + y] * qv[(k + 1) * nx + j];
            gg += s;
            s = 0.f;
            for (int j = 0; j < nx; j++) s += phi[j * nx + y] * qv[k * nx + j];
            gg += -s;
            s = 0.f;
            for (int j = 0; j < nu; j++) s += BR[j * nx + y] * rv[k * nu + j];
            gg += -s;
            g[y] = gg;
            gam[(k + 2) * nx + y] = -1.0f * gg;
        }
        /* S block rows (schur_linsys.cuh:136-147): row k right = phi^T, row k+1 left = phi, row k+1 main = -theta */
        for (int y = 0; y < nx; y++)
            for (int x = 0; x < nx; x++) {
                S[k * o->brow + y * br + 2 * nx + x] = phi[y * nx + x];
                S[(k + 1) * o->brow + y * br + x] = phi[x * nx + y];
                S[(k + 1) * o->brow + y * br + nx + x] = -theta[x * nx + y];
            }
        add_rho(nx, theta, rho);
        gj_inverse(nx, theta, thinv, 1);
        for (int y = 0; y < nx; y++)
            for (int x = 0; x < nx; x++) P[(k + 1) * o->brow + y * br + nx + x] = -thinv[x * nx + y];
        (void)g;
    }
    { /* last block: the Q_0 row (schur_linsys.cuh:166-210) */
        float Q0[NXMAX * NXMAX], Q0i[NXMAX * NXMAX];
        memcpy(Q0, Q, nx * nx * sizeof(float));
        add_rho(nx, Q0, rho);
        for (int y = 0; y < nx; y++)
            for (int x = 0; x < nx; x++) P[y * br + nx + x] = -Q0[x * nx + y];
        gj_inverse(nx, Q0, Q0i, 1);
        for (int y = 0; y < nx; y++)
            for (int x = 0; x < nx; x++) S[y * br + nx + x] = -Q0i[x * nx + y];
        for (int y = 0; y < nx; y++) {
            float s = 0.f;
            for (int j = 0; j < nx; j++) s += Q0i[j * nx + y] * qv[j];
            gam[nx + y] = c[y] + (-s);
        }
    }
    /* kernel 2: stair off-diagonals of P^-1 = -theta_k^-1 phi_k theta_{k-1}^-1 from the STORED diagonals (schur_linsys.cuh:213-260) */
    for (int k = 0; k < N - 1; k++) {
        float tk[NXMAX * NXMAX], tkm1[NXMAX * NXMAX], ph[NXMAX * NXMAX], scr[NXMAX * NXMAX], res[NXMAX * NXMAX];
        for (int y = 0; y < nx; y++)
            for (int x = 0; x < nx; x++) {
                tk[x * nx + y] = P[(k + 1) * o->brow + y * br + nx + x];
                tkm1[x * nx + y] = P[k * o->brow + y * br + nx + x];
                ph[x * nx + y] = S[(k + 1) * o->brow + y * br + x];
            }
        for (int i = 0; i < nx * nx; i++) {
            int y = i % nx, x = i / nx;
            float s = 0.f;
            for (int j = 0; j < nx; j++) s += ph[j * nx + y] * tkm1[x * nx + j];
            scr[i] = s;
        }
        for (int i = 0; i < nx * nx; i++) {
            int y = i % nx, x = i / nx;
            float s = 0.f;
            for (int j = 0; j < nx; j++) s += tk[j * nx + y] * scr[x * nx + j];
            res[i] = s;
        }
        for (int y = 0; y < nx; y++)
            for (int x = 0; x < nx; x++) {
                P[k * o->brow + y * br + 2 * nx + x] = -res[y * nx + x];      /* right = left^T */
                P[(k + 1) * o->brow + y * br + x] = -res[x * nx + y];          /* left */
            }
    }
}

/* block::btdMatrixVectorProduct (linalg.cuh:174-221): out[(row+1)*nx + i] = sum_c M[row][i][c] * vec[row*nx + c] (padded vectors) */
static void btd_matvec(int N, int nx, float* out, const float* M, const float* vec)
{
    int br = 3 * nx;
    for (int row = 0; row < N; row++)
        for (int i = 0; i < nx; i++) {
            float s = 0.f;
            const float* mrow = M + (size_t)row * br * nx + i * br;
            for (int c = 0; c < br; c++) s += mrow[c] * vec[row * nx + c];
            out[(row + 1) * nx + i] = s;
        }
}
static float dotv(int n, const float* a, const float* b)
{
    float s = 0.f;
    for (int i = 0; i < n; i++) s += a[i] * b[i];
    return s;
}

/* solvePCGBatchedKernel (gato/bsqp/kernels/pcg.cuh:14-148) for one trajectory */
static void pcg_one(Orc* o, int b)
{
    int nx = o->nx, N = o->N, n = o->vecp;
    const float abs_tol = 1e-6f;
    if (o->converged[b]) { o->pcg_iters[b] = 0; return; }
    const float* S = o->S + (size_t)b * o->brow * N;
    const float* P = o->Pinv + (size_t)b * o->brow * N;
    const float* bvec = o->gamma + (size_t)b * n;
    float* xg = o->lambda + (size_t)b * n;
    float eps = o->pcg_tol[b];
    float* w = o->pcg_work + (size_t)b * 5 * n; /* block::zeroSharedMemory, pcg.cuh:37 */
    memset(w, 0, (size_t)5 * n * sizeof(float));
    float *Ap = w, *x = w + n, *r = w + 2 * n, *z = w + 3 * n, *pv = w + 4 * n;
    memcpy(x, xg, n * sizeof(float));
    btd_matvec(N, nx, r, S, x);
    for (int i = 0; i < n; i++) r[i] = bvec[i] - r[i];
    btd_matvec(N, nx, z, P, r);
    memcpy(pv, z, n * sizeof(float));
    float rho = dotv(n, r, z);
    uint32_t iters = 0;
    if (fabsf(rho) < abs_tol) { o->pcg_iters[b] = 0; return; }
    float rho_init = fabsf(rho);
    for (uint32_t i = 0; i < o->p.max_pcg_iters; i++) {
        iters++;
        btd_matvec(N, nx, Ap, S, pv);
        float alpha = rho / dotv(n, pv, Ap);
        for (int j = 0; j < n; j++) { x[j] += alpha * pv[j]; r[j] -= alpha * Ap[j]; }
        btd_matvec(N, nx, z, P, r);
        float rho_new = dotv(n, r, z);
        if (fabsf(rho_new) < (abs_tol + eps * rho_init)) break;
        float beta = rho_new / rho;
        rho = rho_new;
        for (int j = 0; j < n; j++) pv[j] = z[j] + beta * pv[j];
    }
    o->pcg_iters[b] = iters;
    memcpy(xg, x, n * sizeof(float));
}

/* computeDzBatchedKernel (gato/bsqp/kernels/schur_linsys.cuh:316-431) for one trajectory; q,r become the KKT residuals */
static void compute_dz_one(Orc* o, int b)
{
    int nx = o->nx, nu = o->nu, N = o->N;
    const float* lam = o->lambda + (size_t)b * o->vecp;
    float* dz = o->dz + (size_t)b * o->traj;
    for (int k = 0; k < N; k++) {
        const float* Qi = o->Qinv + ((size_t)b * N + k) * nx * nx;
        const float* Ak = o->A + ((size_t)b * N + k) * nx * nx;
        float* qk = o->q + ((size_t)b * N + k) * nx;
        float scr[NXMAX], res[NXMAX];
        for (int x = 0; x < nx; x++) {
            float s = 0.f;
            if (k < N - 1) {
                for (int j = 0; j < nx; j++) s += lam[(k + 2) * nx + j] * Ak[x * nx + j];
                s = -s;
            }
            scr[x] = s + lam[(k + 1) * nx + x];
        }
        for (int x = 0; x < nx; x++) res[x] = qk[x] - scr[x];
        for (int y = 0; y < nx; y++) {
            float s = 0.f;
            for (int j = 0; j < nx; j++) s += Qi[j * nx + y] * res[j];
            dz[k * (nx + nu) + y] = -1.0f * s;
        }
        for (int x = 0; x < nx; x++) qk[x] = res[x];
        float* rk = o->r + ((size_t)b * N + k) * nu;
        if (k == N - 1) {
            for (int i = 0; i < nu; i++) rk[i] = 0.f;
            continue;
        }
        const float* Ri = o->Rinv + ((size_t)b * N + k) * nu * nu;
        const float* Bk = o->Bm + ((size_t)b * N + k) * nx * nu;
        float su[NQMAX], ru[NQMAX];
        for (int x = 0; x < nu; x++) {
            float s = 0.f;
            for (int j = 0; j < nx; j++) s += lam[(k + 2) * nx + j] * Bk[x * nx + j];
            su[x] = rk[x] - (-s);
        }
        for (int y = 0; y < nu; y++) {
            float s = 0.f;
            for (int j = 0; j < nu; j++) s += Ri[j * nu + y] * su[j];
            ru[y] = s;
            dz[k * (nx + nu) + nx + y] = -1.0f * s;
        }
        (void)ru;
        for (int x = 0; x < nu; x++) rk[x] = su[x];
    }
}

/* computeMeritBatchedKernel (gato/bsqp/kernels/merit.cuh:17-92): merit[b*na + a] */
static void merit_one(Orc* o, int b, int na, float* merit, const float* xu, const float* x_s, const float* ref, float dt, int zero_dz)
{
    int nq = o->nq, nx = o->nx, nu = o->nu, N = o->N;
    const float* fe = o->f_ext + 6 * b;
    const float* xub = xu + (size_t)b * o->traj;
    const float* dzb = o->dz + (size_t)b * o->traj;
    float mu = o->mu[b];
    for (int ai = 0; ai < na; ai++) {
        float alpha = (float)(1.0 / (double)(1 << ai));
        float total = 0.f;
        for (int k = 0; k < N; k++) {
            float s[3 * NXMAX];
            int cnt = (k == N - 1) ? nx : (2 * nx + nu);
            for (int i = 0; i < cnt; i++) s[i] = xub[k * (nx + nu) + i] + alpha * (zero_dz ? 0.f : dzb[k * (nx + nu) + i]);
            float cost = tracking_cost(o, (const OrcCosts*)(o->costw + 7 * (size_t)b), s, ref + (size_t)b * 6 * N + 6 * k, k < N - 1, k == N - 1);
            float con = 0.f;
            if (k < N - 1) { /* compute_integrator_error, integrator.cuh:211-233 */
                float qdd[NQMAX], xn[NXMAX];
                forward_dynamics(o->model, s, s + nq, s + nx, fe, qdd);
                integrate(nq, xn, s, s + nq, qdd, dt);
                for (int i = 0; i < nq; i++) con += fabsf(s[nx + nu + i] - xn[i]);
                for (int i = 0; i < nq; i++) con += fabsf(s[nx + nu + nq + i] - xn[nq + i]);
            } else {
                for (int i = 0; i < nx; i++)
                    con += fabsf(xub[i] + alpha * (zero_dz ? 0.f : dzb[i]) - x_s[b * nx + i]); /* merit.cuh:72-84 */
            }
            total += cost + mu * con; /* atomicAdd order not restated (merit.cuh:88-91) */
        }
        merit[b * na + ai] = total;
    }
}

/* lineSearchAndUpdateBatchedKernel (gato/bsqp/kernels/line_search.cuh:13-98) for one trajectory */
static void line_search_one(Orc* o, int b, float* xu)
{
    float* mer = o->merit + (size_t)b * NUM_ALPHAS;
    float best = 1e38f;
    uint32_t idx = 0;
    for (uint32_t i = 0; i < NUM_ALPHAS; i++) {
        if (mer[i] < best) { best = mer[i]; idx = i; }
        mer[i] = 0.f;
    }
    int success = best < o->merit_cur[b];
    if (o->adapt_rho) {
        float mult = success ? fminf(o->drho[b] / RHO_FACTOR, 1 / RHO_FACTOR) : fmaxf(o->drho[b] * RHO_FACTOR, RHO_FACTOR);
        o->drho[b] = mult;
        o->rho[b] = fmaxf(o->rho[b] * mult, RHO_MIN);
        o->rho[b] = fminf(o->rho[b], RHO_MAX);
    }
    if (!success) {
        if (o->rho[b] > RHO_MAX) o->rho[b] = RHO_INIT; /* line_search.cuh:77-79: live only with adaptation off and rho_batch > RHO_MAX */
        o->step[b] = -1.f;
    } else {
        float step = (float)(1.0 / (double)(1 << idx));
        o->merit_cur[b] = best;
        o->step[b] = step;
        float* x = xu + (size_t)b * o->traj;
        const float* dz = o->dz + (size_t)b * o->traj;
        for (int i = 0; i < o->traj; i++) x[i] += step * dz[i];
    }
}

/* ------------------------------------------------------------------------------------------------------------------
 * public C interface (ctypes): oracle/oracle.py
 * ---------------------------------------------------------------------------------------------------------------- */
#define ALLOCF(n) ((float*)calloc((size_t)(n), sizeof(float)))

Orc* orc_create(int plant, int N, int B, const OrcParams* p)
{
    Orc* o = (Orc*)calloc(1, sizeof(Orc));
    o->model = plant == 0 ? &ORC_MODEL_INDY7 : &ORC_MODEL_IIWA14;
    o->nq = o->model->nq; o->nx = 2 * o->nq; o->nu = o->nq; o->N = N; o->B = B;
    o->traj = (o->nx + o->nu) * N - o->nu;
    o->vecp = (N + 2) * o->nx;
    o->brow = 3 * o->nx * o->nx;
    o->p = *p;
    o->adapt_rho = 1;
    o->nthreads = 1;
    int nx = o->nx, nu = o->nu;
    size_t BN = (size_t)B * N;
    o->lambda = ALLOCF((size_t)B * o->vecp);
    o->rho = ALLOCF(B); o->drho = ALLOCF(B); o->rho_init = ALLOCF(B); o->drho_init = ALLOCF(B);
    o->mu = ALLOCF(B); o->pcg_tol = ALLOCF(B); o->f_ext = ALLOCF(6 * (size_t)B);
    o->costw = ALLOCF(7 * (size_t)B);
    for (int b = 0; b < B; b++) { /* bsqp.cuh:48-57 */
        o->drho[b] = o->drho_init[b] = 1.0f;
        o->rho[b] = o->rho_init[b] = p->rho;
        o->mu[b] = p->mu;
        o->pcg_tol[b] = p->pcg_tol;
        float* w = o->costw + 7 * (size_t)b;
        w[0] = p->q_cost; w[1] = p->qd_cost; w[2] = p->u_cost; w[3] = p->N_cost; w[4] = p->q_lim_cost; w[5] = p->vel_lim_cost; w[6] = p->ctrl_lim_cost;
    }
    o->Q = ALLOCF(BN * nx * nx); o->R = ALLOCF(BN * nu * nu); o->q = ALLOCF(BN * nx); o->r = ALLOCF(BN * nu);
    o->A = ALLOCF(BN * nx * nx); o->Bm = ALLOCF(BN * nx * nu); o->c = ALLOCF(BN * nx);
    o->Qinv = ALLOCF(BN * nx * nx); o->Rinv = ALLOCF(BN * nu * nu);
    o->S = ALLOCF(BN * o->brow); o->Pinv = ALLOCF(BN * o->brow); o->gamma = ALLOCF((size_t)B * o->vecp);
    o->dz = ALLOCF((size_t)B * o->traj);
    o->pcg_work = ALLOCF((size_t)B * 5 * o->vecp);
    o->merit = ALLOCF((size_t)B * NUM_ALPHAS); o->merit_cur = ALLOCF(B); o->merit_init0 = ALLOCF(B); o->step = ALLOCF(B);
    o->converged = (int32_t*)calloc(B, sizeof(int32_t));
    o->pcg_iters = (uint32_t*)calloc(B, sizeof(uint32_t));
    size_t mi = p->max_sqp_iters ? p->max_sqp_iters : 1;
    o->st_pcg_iters = (int32_t*)calloc(mi * B, sizeof(int32_t));
    o->st_min_merit = ALLOCF(mi * B); o->st_step = ALLOCF(mi * B);
    o->st_merits = ALLOCF(mi * B * NUM_ALPHAS); o->st_merit_before = ALLOCF(mi * B);
    o->sqp_iters = (uint32_t*)calloc(B, sizeof(uint32_t));
    o->kkt_converged = (int32_t*)calloc(B, sizeof(int32_t));
    return o;
}

void orc_destroy(Orc* o)
{
    if (!o) return;
    float* fl[] = {o->lambda, o->rho, o->drho, o->rho_init, o->drho_init, o->mu, o->pcg_tol, o->f_ext, o->Q, o->R, o->q, o->r, o->A, o->Bm,
                   o->c, o->Qinv, o->Rinv, o->S, o->Pinv, o->gamma, o->dz, o->merit, o->merit_cur, o->merit_init0, o->step, o->st_min_merit, o->st_step, o->st_merits, o->st_merit_before, o->pcg_work, o->costw};
    for (size_t i = 0; i < sizeof(fl) / sizeof(fl[0]); i++) free(fl[i]);
    free(o->converged); free(o->pcg_iters); free(o->st_pcg_iters); free(o->sqp_iters); free(o->kkt_converged);
    free(o);
}

void orc_set_threads(Orc* o, int n) { o->nthreads = n > 0 ? n : 1; }
void orc_set_f_ext(Orc* o, const float* v) { memcpy(o->f_ext, v, 6 * (size_t)o->B * sizeof(float)); }
void orc_set_rho(Orc* o, const float* v, int as_default)
{
    if (as_default) memcpy(o->rho_init, v, o->B * sizeof(float));
    memcpy(o->rho, v, o->B * sizeof(float));
}
void orc_set_drho(Orc* o, const float* v, int as_default)
{
    if (as_default) memcpy(o->drho_init, v, o->B * sizeof(float));
    memcpy(o->drho, v, o->B * sizeof(float));
}
void orc_set_mu(Orc* o, const float* v) { memcpy(o->mu, v, o->B * sizeof(float)); }
/* [B][7] = q, qd, u, N, q_lim, vel_lim, ctrl_lim cost weights per trajectory */
void orc_set_cost_weights(Orc* o, const float* v) { memcpy(o->costw, v, 7 * (size_t)o->B * sizeof(float)); }
void orc_set_pcg_tol(Orc* o, const float* v) { memcpy(o->pcg_tol, v, o->B * sizeof(float)); }
void orc_reset_dual(Orc* o) { memset(o->lambda, 0, (size_t)o->B * o->vecp * sizeof(float)); }
void orc_reset_rho(Orc* o)
{
    memcpy(o->rho, o->rho_init, o->B * sizeof(float));
    memcpy(o->drho, o->drho_init, o->B * sizeof(float));
}
void orc_set_rho_adaptation(Orc* o, int e) { o->adapt_rho = e; }

/* stage entry points (each loops the batch, OpenMP over trajectories when built with -fopenmp) */
void orc_setup_kkt(Orc* o, const float* xu, const float* x_s, const float* ref, float dt)
{
#pragma omp parallel for schedule(dynamic, 1) num_threads(o->nthreads)
    for (int b = 0; b < o->B; b++) setup_kkt_one(o, b, xu, x_s, ref, dt);
}
void orc_form_schur(Orc* o)
{
#pragma omp parallel for schedule(dynamic, 1) num_threads(o->nthreads)
    for (int b = 0; b < o->B; b++) form_schur_one(o, b);
}
void orc_pcg(Orc* o)
{
#pragma omp parallel for schedule(dynamic, 1) num_threads(o->nthreads)
    for (int b = 0; b < o->B; b++) pcg_one(o, b);
}
void orc_compute_dz(Orc* o)
{
#pragma omp parallel for schedule(dynamic, 1) num_threads(o->nthreads)
    for (int b = 0; b < o->B; b++) compute_dz_one(o, b);
}
void orc_merit(Orc* o, int na, float* merit, const float* xu, const float* x_s, const float* ref, float dt, int zero_dz)
{
#pragma omp parallel for schedule(dynamic, 1) num_threads(o->nthreads)
    for (int b = 0; b < o->B; b++) merit_one(o, b, na, merit, xu, x_s, ref, dt, zero_dz);
}
void orc_line_search(Orc* o, float* xu)
{
#pragma omp parallel for schedule(dynamic, 1) num_threads(o->nthreads)
    for (int b = 0; b < o->B; b++) line_search_one(o, b, xu);
}

/* BSQP::solve (gato/bsqp/bsqp.cuh:103-197).  xu is updated in place.  Returns the number of outer iterations executed. */
uint32_t orc_solve(Orc* o, float* xu, float dt, const float* x_s, const float* ref)
{
    int B = o->B;
    memset(o->dz, 0, (size_t)B * o->traj * sizeof(float));
    memset(o->pcg_iters, 0, B * sizeof(uint32_t));
    memset(o->converged, 0, B * sizeof(int32_t));
    memset(o->sqp_iters, 0, B * sizeof(uint32_t));
    memset(o->kkt_converged, 0, B * sizeof(int32_t));
    orc_merit(o, 1, o->merit_cur, xu, x_s, ref, dt, 1);
    memcpy(o->merit_init0, o->merit_cur, B * sizeof(float));
    o->iters_done = 0; o->ls_done = 0;
    for (uint32_t it = 0; it < o->p.max_sqp_iters; it++) {
        orc_setup_kkt(o, xu, x_s, ref, dt);
        orc_form_schur(o);
        orc_pcg(o);
        orc_compute_dz(o);
        o->iters_done = it + 1;
        uint32_t num_solved = 0;
        for (int b = 0; b < B; b++) {
            o->st_pcg_iters[(size_t)it * B + b] = (int32_t)o->pcg_iters[b];
            if (o->pcg_iters[b] == 0) o->kkt_converged[b] = 1; /* bsqp.cuh:153-156; kkt_tol is unused */
            o->sqp_iters[b] += 1;
            if (o->kkt_converged[b]) num_solved++;
        }
        /* bsqp.cuh:165.  A SHARD of a larger batch (orc_set_shard: the oracle standing in for one rank's solver in the multi-process tests)
         * applies the rule to the whole batch: the count is summed over the ranks by the caller's reduction */
        float batch = (float)B;
        if (o->shard_reduce) {
            num_solved = o->shard_reduce(num_solved, it, o->shard_ctx);
            batch = (float)o->shard_global_batch;
        }
        if ((float)num_solved >= batch * o->p.solve_ratio) break;
        memcpy(o->converged, o->kkt_converged, B * sizeof(int32_t)); /* bsqp.cuh:167 */
        orc_merit(o, NUM_ALPHAS, o->merit, xu, x_s, ref, dt, 0);
        memcpy(o->st_merits + (size_t)it * B * NUM_ALPHAS, o->merit, (size_t)B * NUM_ALPHAS * sizeof(float));
        memcpy(o->st_merit_before + (size_t)it * B, o->merit_cur, B * sizeof(float));
        orc_line_search(o, xu);
        memcpy(o->st_min_merit + (size_t)it * B, o->merit_cur, B * sizeof(float));
        memcpy(o->st_step + (size_t)it * B, o->step, B * sizeof(float));
        o->ls_done = it + 1;
    }
    orc_merit(o, 1, o->merit_cur, xu, x_s, ref, dt, 1); /* final merit with dz = 0 (bsqp.cuh:180-182) */
    memcpy(o->drho, o->drho_init, B * sizeof(float));    /* bsqp.cuh:189; rho is NOT reset */
    return o->iters_done;
}

/* simForwardBatchedKernel (gato/bsqp/kernels/sim.cuh:14-49): one shared (x_k,u_k), B wrench hypotheses */
void orc_sim_forward(Orc* o, float* xkp1, const float* xk, const float* uk, float dt)
{
    int nq = o->nq, nx = o->nx;
    for (int b = 0; b < o->B; b++) {
        float qdd[NQMAX];
        forward_dynamics(o->model, xk, xk + nq, uk, o->f_ext + 6 * b, qdd);
        integrate(nq, xkp1 + (size_t)b * nx, xk, xk + nq, qdd, dt);
    }
}

/* raw accessors for stage dumps */
float* orc_buf(Orc* o, const char* name)
{
#define M(n, f) if (!strcmp(name, n)) return o->f
    M("Q", Q); M("R", R); M("q", q); M("r", r); M("A", A); M("B", Bm); M("c", c); M("Qinv", Qinv); M("Rinv", Rinv);
    M("S", S); M("Pinv", Pinv); M("gamma", gamma); M("lambda", lambda); M("dz", dz); M("merit", merit);
    M("merit_cur", merit_cur); M("merit_init0", merit_init0); M("step", step); M("rho", rho); M("drho", drho);
    M("st_min_merit", st_min_merit); M("st_step", st_step); M("st_merits", st_merits); M("st_merit_before", st_merit_before);
#undef M
    return NULL;
}
int32_t* orc_ibuf(Orc* o, const char* name)
{
    if (!strcmp(name, "st_pcg_iters")) return o->st_pcg_iters;
    if (!strcmp(name, "kkt_converged")) return o->kkt_converged;
    if (!strcmp(name, "sqp_iters")) return (int32_t*)o->sqp_iters;
    if (!strcmp(name, "pcg_iters")) return (int32_t*)o->pcg_iters;
    if (!strcmp(name, "converged")) return o->converged;
    return NULL;
}
/* multi-process tests: this solver holds a shard of a batch of global_batch trajectories; reduce(local solved count, iteration) -> the sum over the ranks */
void orc_set_shard(Orc* o, uint32_t (*reduce)(uint32_t, uint32_t, void*), void* ctx, long global_batch)
{
    o->shard_reduce = reduce; o->shard_ctx = ctx; o->shard_global_batch = global_batch;
}
uint32_t orc_iters_done(Orc* o) { return o->iters_done; }
uint32_t orc_ls_done(Orc* o) { return o->ls_done; }

/* unit-level entry points for the dynamics / kinematics (tests/test_oracle_dynamics.py) */
void orc_fd(int plant, const float* q, const float* qd, const float* u, const float* f_ext, float* qdd)
{
    forward_dynamics(plant == 0 ? &ORC_MODEL_INDY7 : &ORC_MODEL_IIWA14, q, qd, u, f_ext, qdd);
}
void orc_fd_grad(int plant, const float* q, const float* qd, const float* u, const float* f_ext, float* qdd, float* dqdd)
{
    forward_dynamics_and_gradient(plant == 0 ? &ORC_MODEL_INDY7 : &ORC_MODEL_IIWA14, q, qd, u, f_ext, qdd, dqdd);
}
void orc_rnea(int plant, const float* q, const float* qd, const float* qdd, const float* f_ext, float* c)
{
    const OrcModel* m = plant == 0 ? &ORC_MODEL_INDY7 : &ORC_MODEL_IIWA14;
    float X[NQMAX][36], v[NQMAX][6], a[NQMAX][6], f[NQMAX][6];
    build_X(m, q, X);
    rnea(m, X, qd, qdd, 1, f_ext, v, a, f, c);
}
void orc_minv(int plant, const float* q, float* Minv_full)
{
    const OrcModel* m = plant == 0 ? &ORC_MODEL_INDY7 : &ORC_MODEL_IIWA14;
    float X[NQMAX][36], Mi[NQMAX * NQMAX];
    build_X(m, q, X);
    direct_minv(m, X, Mi);
    for (int r = 0; r < m->nq; r++)
        for (int c = 0; c < m->nq; c++) Minv_full[c * m->nq + r] = minv_sym(Mi, m->nq, r, c);
}
void orc_ee(int plant, const float* q, float* e, float* J)
{
    ee_pos(plant == 0 ? &ORC_MODEL_INDY7 : &ORC_MODEL_IIWA14, q, e, J);
}
void orc_gj_inverse(int n, float* V, float* Vinv, int one_matrix_form) { gj_inverse(n, V, Vinv, one_matrix_form); }
