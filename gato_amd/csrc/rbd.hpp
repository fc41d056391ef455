// rbd.hpp -- per-lane rigid-body dynamics for serial revolute-z chains, CDNA4 (gfx950).
//
// One lane evaluates one (trajectory, knot) problem start to finish; the batch x knot dimension is spread over the 64 lanes of
// a wavefront, so there is no cross-lane traffic and no barrier anywhere in the dynamics (the reference spends ~100
// __syncthreads per evaluation on one 128..352-thread block per knot, setup_kkt.cuh:15-108 / indy7_grid.cuh).
// Every loop over joints is fully unrolled and the plant tables (gato::Indy7 / gato::Iiwa14, robot_models.hpp) are
// compile-time constants: the joint-origin rotations E0 are signed permutations and the spatial inertias are sparse, so the
// `cmad` helper below lets the compiler drop every structural zero and +-1 multiply.
//
// What is computed follows the reference's generated GRiD code (same recursions, same outputs):
//   RNEA with wrench      indy7_fext.cuh:16-212, 216-405        -> rnea()
//   direct M^-1           indy7_grid.cuh:2918-3308              -> minv()
//   d(RNEA)/d(q,qd)       indy7_grid.cuh:3373-3774              -> rnea_grad()   (column-at-a-time instead of the compact table)
//   FK position/Jacobian  indy7_grid.cuh:1834-1901, 1933-2025   -> ee_pos(), ee_jac()
// but on 3x3 rotation + translation form  X = [E 0; -E r~ E]  instead of dense 6x6 products.
#pragma once
#include <hip/hip_runtime.h>

#include "robot_models.hpp"

namespace gato {

#define GATO_DEV __device__ __forceinline__

// acc (+)= coef * x with coef a compile-time constant after inlining: zeros vanish, +-1 become add/sub, first term initialises.
GATO_DEV void cmad(float& acc, bool& started, const float coef, const float x)
{
    if (coef == 0.0f) return;
    const float t = (coef == 1.0f) ? x : ((coef == -1.0f) ? -x : coef * x);
    if (!started) {
        acc = t;
        started = true;
    } else {
        acc += t;
    }
}

// A wave-uniform `true` the optimiser cannot see through: `if (opaque_true()) { phase }` gives `phase` its own basic block.
GATO_DEV bool opaque_true()
{
    int one;
    asm volatile("s_mov_b32 %0, 1" : "=s"(one));
    return one != 0;
}

template<class M>
struct RBD {
    static constexpr int NQ = M::NQ;
    static constexpr float G = 9.81f;  // indy7_plant.cuh:25-28

    float sn[NQ], cs[NQ];

    GATO_DEV void set_q(const float* q)
    {
#pragma unroll
        for (int k = 0; k < NQ; k++) {
            sincosf(q[k], &sn[k], &cs[k]);
        }
    }

    // ---- 3-vector pieces of X_k = [E 0; -E r~ E], E = Ez(q_k) E0_k ----
    template<int K> GATO_DEV void E0mul(const float* v, float* o) const
    {
#pragma unroll
        for (int r = 0; r < 3; r++) {
            float a = 0.f;
            bool st = false;
#pragma unroll
            for (int c = 0; c < 3; c++) cmad(a, st, M::E0[K][r][c], v[c]);
            o[r] = a;
        }
    }
    template<int K> GATO_DEV void E0Tmul(const float* v, float* o) const
    {
#pragma unroll
        for (int r = 0; r < 3; r++) {
            float a = 0.f;
            bool st = false;
#pragma unroll
            for (int c = 0; c < 3; c++) cmad(a, st, M::E0[K][c][r], v[c]);
            o[r] = a;
        }
    }
    template<int K> GATO_DEV void Emul(const float* v, float* o) const  // o = Ez(q) E0 v
    {
        float w[3];
        E0mul<K>(v, w);
        o[0] = cs[K] * w[0] + sn[K] * w[1];
        o[1] = cs[K] * w[1] - sn[K] * w[0];
        o[2] = w[2];
    }
    template<int K> GATO_DEV void ETmul(const float* u, float* o) const  // o = E0^T Ez(q)^T u
    {
        float w[3];
        w[0] = cs[K] * u[0] - sn[K] * u[1];
        w[1] = sn[K] * u[0] + cs[K] * u[1];
        w[2] = u[2];
        E0Tmul<K>(w, o);
    }
    template<int K> GATO_DEV void rcross(const float* w, float* o) const  // o = r_K x w
    {
        {
            float a = 0.f; bool st = false;
            cmad(a, st, M::R[K][1], w[2]); cmad(a, st, -M::R[K][2], w[1]);
            o[0] = a;
        }
        {
            float a = 0.f; bool st = false;
            cmad(a, st, M::R[K][2], w[0]); cmad(a, st, -M::R[K][0], w[2]);
            o[1] = a;
        }
        {
            float a = 0.f; bool st = false;
            cmad(a, st, M::R[K][0], w[1]); cmad(a, st, -M::R[K][1], w[0]);
            o[2] = a;
        }
    }
    // o = X_K v  (motion vector [angular; linear])
    template<int K> GATO_DEV void X(const float* v, float* o) const
    {
        float t[3], rw[3];
        rcross<K>(v, rw);
        t[0] = v[3] - rw[0]; t[1] = v[4] - rw[1]; t[2] = v[5] - rw[2];
        Emul<K>(v, o);
        Emul<K>(t, o + 3);
    }
    // o = X_K^T f  (force vector [moment; force])
    template<int K> GATO_DEV void XT(const float* f, float* o) const
    {
        float n[3], rl[3];
        ETmul<K>(f + 3, o + 3);
        ETmul<K>(f, n);
        rcross<K>(o + 3, rl);
        o[0] = n[0] + rl[0]; o[1] = n[1] + rl[1]; o[2] = n[2] + rl[2];
    }
    // o = I_K v
    template<int K> GATO_DEV static void Imul(const float* v, float* o)
    {
#pragma unroll
        for (int r = 0; r < 6; r++) {
            float a = 0.f;
            bool st = false;
#pragma unroll
            for (int c = 0; c < 6; c++) cmad(a, st, M::I[K][r][c], v[c]);
            o[r] = a;
        }
    }
    // v x S for S = e_z: column 2 of the motion cross-product matrix (indy7_grid.cuh:336-344)
    GATO_DEV static void mx2(const float* v, float* o)
    {
        o[0] = v[1]; o[1] = -v[0]; o[2] = 0.f; o[3] = v[4]; o[4] = -v[3]; o[5] = 0.f;
    }
    // o = v x* f (indy7_grid.cuh:858-866)
    GATO_DEV static void fxv(const float* v, const float* t, float* o)
    {
        o[0] = -v[2] * t[1] + v[1] * t[2] - v[5] * t[4] + v[4] * t[5];
        o[1] = v[2] * t[0] - v[0] * t[2] + v[5] * t[3] - v[3] * t[5];
        o[2] = -v[1] * t[0] + v[0] * t[1] - v[4] * t[3] + v[3] * t[4];
        o[3] = -v[2] * t[4] + v[1] * t[5];
        o[4] = v[2] * t[3] - v[0] * t[5];
        o[5] = -v[1] * t[3] + v[0] * t[4];
    }

    // ---- RNEA ----------------------------------------------------------------------------------------------------
    // v,a,f: [NQ][6]; f returns ACCUMULATED over the subtree; c[k] = f[k][2].  qdd == nullptr <=> zero accelerations.
    template<int K> GATO_DEV void rnea_fwd(const float* qd, const float* qdd, float (*v)[6], float (*a)[6]) const
    {
        if constexpr (K == 0) {
            // v_0 = S qd_0, a_0 = X_0 [0,0,0,0,0,g] (+ S qdd_0)
            const float g6[6] = {0.f, 0.f, 0.f, 0.f, 0.f, G};
#pragma unroll
            for (int r = 0; r < 6; r++) v[0][r] = 0.f;
            v[0][2] = qd[0];
            X<0>(g6, a[0]);
            if (qdd) a[0][2] += qdd[0];
        } else {
            X<K>(v[K - 1], v[K]);
            X<K>(a[K - 1], a[K]);
            v[K][2] += qd[K];
            if (qdd) a[K][2] += qdd[K];
            a[K][0] += v[K][1] * qd[K];
            a[K][1] -= v[K][0] * qd[K];
            a[K][3] += v[K][4] * qd[K];
            a[K][4] -= v[K][3] * qd[K];
        }
        if constexpr (K + 1 < NQ) rnea_fwd<K + 1>(qd, qdd, v, a);
    }
    template<int K> GATO_DEV void rnea_force(const float (*v)[6], const float (*a)[6], float (*f)[6], const float* fext) const
    {
        float Iv[6], t[6];
        Imul<K>(a[K], f[K]);
        Imul<K>(v[K], Iv);
        fxv(v[K], Iv, t);
#pragma unroll
        for (int r = 0; r < 6; r++) f[K][r] += t[r];
        if constexpr (K == NQ - 1) {
#pragma unroll
            for (int r = 0; r < 6; r++) f[K][r] -= fext[r];  // indy7_fext.cuh:134-144
        }
        if constexpr (K + 1 < NQ) rnea_force<K + 1>(v, a, f, fext);
    }
    template<int K> GATO_DEV void rnea_bwd(float (*f)[6]) const
    {
        if constexpr (K >= 1) {
            float t[6];
            XT<K>(f[K], t);
#pragma unroll
            for (int r = 0; r < 6; r++) f[K - 1][r] += t[r];
            rnea_bwd<K - 1>(f);
        }
    }
    // Bias forces only (f accumulated over the subtrees, c[k] = f[k][2]) at zero accelerations: v_k, a_k live for ONE body -- each
    // body's force is formed as soon as its velocity and acceleration exist (same operations and order per value as rnea()), so the
    // forward-dynamics-only paths (merit, sim) carry 36 + 12 floats through the recursion instead of 108.
    template<int K> GATO_DEV void rnea_lean_fwd(const float* qd, const float* vp, const float* ap, float (*f)[6], const float* fext) const
    {
        float v[6], a[6];
        if constexpr (K == 0) {
            const float g6[6] = {0.f, 0.f, 0.f, 0.f, 0.f, G};
#pragma unroll
            for (int r = 0; r < 6; r++) v[r] = 0.f;
            v[2] = qd[0];
            X<0>(g6, a);
        } else {
            X<K>(vp, v);
            X<K>(ap, a);
            v[2] += qd[K];
            a[0] += v[1] * qd[K];
            a[1] -= v[0] * qd[K];
            a[3] += v[4] * qd[K];
            a[4] -= v[3] * qd[K];
        }
        {
            float Iv[6], t[6];
            Imul<K>(a, f[K]);
            Imul<K>(v, Iv);
            fxv(v, Iv, t);
#pragma unroll
            for (int r = 0; r < 6; r++) f[K][r] += t[r];
            if constexpr (K == NQ - 1) {
#pragma unroll
                for (int r = 0; r < 6; r++) f[K][r] -= fext[r];
            }
        }
        if constexpr (K + 1 < NQ) rnea_lean_fwd<K + 1>(qd, v, a, f, fext);
    }
    GATO_DEV void rnea_lean(const float* qd, const float* fext, float (*f)[6]) const
    {
        rnea_lean_fwd<0>(qd, nullptr, nullptr, f, fext);
        rnea_bwd<NQ - 1>(f);
    }
    GATO_DEV void rnea(const float* qd, const float* qdd, const float* fext, float (*v)[6], float (*a)[6], float (*f)[6]) const
    {
        rnea_fwd<0>(qd, qdd, v, a);
        rnea_force<0>(v, a, f, fext);
        rnea_bwd<NQ - 1>(f);
    }

    // ---- direct M^-1 (Carpentier) -------------------------------------------------------------------------------
    // Minv[c][r] valid for r <= c (upper triangle, like the reference); sym() folds the index.
    struct MinvT {
        float m[NQ][NQ];  // m[col][row]
        GATO_DEV float sym(int r, int c) const { return r <= c ? m[c][r] : m[r][c]; }
    };
    template<int K> GATO_DEV void minv_bwd(float* IA, float (*F)[6], float (*U)[6], float* Dinv, MinvT& Mi) const
    {
        // IA: articulated inertia of body K, col-major 6x6 (IA[6*c + r]); F[j] = column j of F_K (j >= K)
#pragma unroll
        for (int r = 0; r < 6; r++) U[K][r] = IA[12 + r];
        Dinv[K] = 1.0f / U[K][2];
        Mi.m[K][K] = Dinv[K];
#pragma unroll
        for (int j = K; j < NQ; j++) {
            if (j > K) Mi.m[j][K] = -Dinv[K] * F[j][2];  // F[K] column K is still zero
            if constexpr (K > 0) {
                if (j > K) {
#pragma unroll
                    for (int r = 0; r < 6; r++) F[j][r] += U[K][r] * Mi.m[j][K];
                } else {
#pragma unroll
                    for (int r = 0; r < 6; r++) F[j][r] = U[K][r] * Mi.m[j][K];
                }
            }
        }
        if constexpr (K > 0) {
            // F_parent[:, j] = X^T F[:, j]
#pragma unroll
            for (int j = K; j < NQ; j++) {
                float t[6];
                XT<K>(F[j], t);
#pragma unroll
                for (int r = 0; r < 6; r++) F[j][r] = t[r];
            }
            // IA_parent = I_{K-1} + X^T (IA - U Dinv U^T) X
            float T[36];
#pragma unroll
            for (int c = 0; c < 6; c++) {
                float col[6];
#pragma unroll
                for (int r = 0; r < 6; r++) col[r] = IA[6 * c + r] - U[K][r] * Dinv[K] * U[K][c];
                XT<K>(col, &T[6 * c]);  // T = X^T Ia, column c
            }
            // IA_parent = (X^T T^T)^T with T^T's columns = T's rows; the result is symmetric, so fill it column by column
#pragma unroll
            for (int r = 0; r < 6; r++) {
                float row[6], o[6];
#pragma unroll
                for (int c = 0; c < 6; c++) row[c] = T[6 * c + r];
                XT<K>(row, o);  // o[c'] = (T X)[r][c']
#pragma unroll
                for (int c = 0; c < 6; c++) IA[6 * c + r] = M::I[K - 1][r][c] + o[c];
            }
            minv_bwd<K - 1>(IA, F, U, Dinv, Mi);
        }
    }
    template<int K> GATO_DEV void minv_fwd(float (*F)[6], const float (*U)[6], const float* Dinv, MinvT& Mi) const
    {
        if constexpr (K == 0) {
#pragma unroll
            for (int j = 0; j < NQ; j++) {
#pragma unroll
                for (int r = 0; r < 6; r++) F[j][r] = 0.f;
                F[j][2] = Mi.m[j][0];
            }
        } else {
#pragma unroll
            for (int j = K; j < NQ; j++) {
                float t[6];
                X<K>(F[j], t);
                float d = 0.f;
#pragma unroll
                for (int r = 0; r < 6; r++) d += t[r] * U[K][r];
                Mi.m[j][K] -= Dinv[K] * d;
                t[2] += Mi.m[j][K];
#pragma unroll
                for (int r = 0; r < 6; r++) F[j][r] = t[r];
            }
        }
        if constexpr (K + 1 < NQ) minv_fwd<K + 1>(F, U, Dinv, Mi);
    }
    // payload != nullptr: a body hanging from the last link through a joint that has been eliminated articulated-body fashion adds its
    // articulated inertia blkdiag(0, payload) (3x3, force per linear acceleration, last-link frame) to the last link's (payload_dynamics, kernels.hpp)
    GATO_DEV void minv(MinvT& Mi, const float (*payload)[3] = nullptr) const
    {
        float IA[36], F[NQ][6], U[NQ][6], Dinv[NQ];
#pragma unroll
        for (int c = 0; c < 6; c++)
#pragma unroll
            for (int r = 0; r < 6; r++) IA[6 * c + r] = M::I[NQ - 1][r][c];
        if (payload) {
#pragma unroll
            for (int c = 0; c < 3; c++)
#pragma unroll
                for (int r = 0; r < 3; r++) IA[6 * (3 + c) + 3 + r] += payload[r][c];
        }
#pragma unroll
        for (int j = 0; j < NQ; j++)
#pragma unroll
            for (int r = 0; r < 6; r++) F[j][r] = 0.f;
        minv_bwd<NQ - 1>(IA, F, U, Dinv, Mi);
        minv_fwd<0>(F, U, Dinv, Mi);
    }

    // qdd = Minv (u - c)   (forward_dynamics_finish, indy7_grid.cuh:3322-3334)
    GATO_DEV static void fd_finish(const MinvT& Mi, const float* u, const float (*f)[6], float* qdd)
    {
        float tau[NQ];
#pragma unroll
        for (int k = 0; k < NQ; k++) tau[k] = u[k] - f[k][2];
#pragma unroll
        for (int r = 0; r < NQ; r++) {
            float s = 0.f;
#pragma unroll
            for (int c = 0; c < NQ; c++) s += Mi.sym(r, c) * tau[c];
            qdd[r] = s;
        }
    }

    // forward dynamics only (merit / sim paths): plant::forwardDynamics(..., d_f_ext), indy7_plant.cuh:163-173
    GATO_DEV void forward_dynamics(const float* qd, const float* u, const float* fext, float* qdd) const
    {
        MinvT Mi;
        if (opaque_true()) minv(Mi);  // its own basic block: M^-1's 110 live values are gone before the recursion below starts
        float f[NQ][6];
        rnea_lean(qd, fext, f);
        fd_finish(Mi, u, f, qdd);
    }

    // ---- d(RNEA)/d(q_J) and d(RNEA)/d(qd_J), one derivative column at a time --------------------------------------
    // Same recursions as inverse_dynamics_gradient_inner (indy7_grid.cuh:3373-3774); dc_dq[i], dc_dqd[i] = d c_i / d (q_J, qd_J).
    // QD = false: derivatives w.r.t. q_J (dv, da, df hold d/dq_J);  QD = true: w.r.t. qd_J.  The two passes share nothing but v, a, f,
    // so they run one after the other: together they kept 440 registers live, apart 270 (hipcc 7.2, gfx950).
    template<int J, int I2, bool QD> GATO_DEV void grad_fwd(const float* qd, const float (*v)[6], const float (*a)[6], float* dv, float* da,
                                                            float (*df)[6]) const
    {
        // on entry (I2 > J): dv, da hold the derivatives of body I2-1; on exit those of body I2
        if constexpr (I2 == J) {
            if constexpr (QD) {
#pragma unroll
                for (int r = 0; r < 6; r++) dv[r] = 0.f;
                dv[2] = 1.f;
                // da/dqd_J = mx2(S) qd_J + mx2(v_J) = mx2(v_J)   (mx2(S) = 0)
                mx2(v[J], da);
            } else if constexpr (J == 0) {
                const float g6[6] = {0.f, 0.f, 0.f, 0.f, 0.f, G};
                float Xa[6];
                X<0>(g6, Xa);
#pragma unroll
                for (int r = 0; r < 6; r++) dv[r] = 0.f;
                mx2(Xa, da);
            } else {
                float Xv[6], Xa[6], t[6];
                X<J>(v[J - 1], Xv);
                X<J>(a[J - 1], Xa);
                mx2(Xv, dv);
                mx2(dv, t);
                mx2(Xa, da);
#pragma unroll
                for (int r = 0; r < 6; r++) da[r] += t[r] * qd[J];
            }
        } else {
            float t[6], u6[6];
            X<I2>(dv, t);
#pragma unroll
            for (int r = 0; r < 6; r++) dv[r] = t[r];
            X<I2>(da, t);
            mx2(dv, u6);
#pragma unroll
            for (int r = 0; r < 6; r++) da[r] = t[r] + u6[r] * qd[I2];
        }
        // df = I da + dv x* (I v) + v x* (I dv)
        {
            float Ida[6], Idv[6], t1[6], t2[6], Ivl[6];
            Imul<I2>(v[I2], Ivl);  // recomputed here instead of kept for all bodies: fewer live registers, same value
            Imul<I2>(da, Ida);
            Imul<I2>(dv, Idv);
            fxv(dv, Ivl, t1);
            fxv(v[I2], Idv, t2);
#pragma unroll
            for (int r = 0; r < 6; r++) df[I2][r] = Ida[r] + t1[r] + t2[r];
        }
        if constexpr (I2 + 1 < NQ) grad_fwd<J, I2 + 1, QD>(qd, v, a, dv, da, df);
    }
    template<int J, int I2, bool QD> GATO_DEV void grad_bwd(const float (*f)[6], float (*df)[6], float* dc) const
    {
        // I2 runs NQ-1 .. 0; df[I2] is complete when visited.  Bodies below J only receive the propagated part.
        dc[I2] = df[I2][2];
        if constexpr (I2 >= 1) {
            float t[6];
            XT<I2>(df[I2], t);
            if constexpr (I2 == J && !QD) {
                // + d(X_J^T)/dq_J f_J = -X_J^T mx2(f_J)
                float mf[6], t2[6];
                mx2(f[J], mf);
                XT<J>(mf, t2);
#pragma unroll
                for (int r = 0; r < 6; r++) t[r] -= t2[r];
            }
            if constexpr (I2 - 1 >= J) {
#pragma unroll
                for (int r = 0; r < 6; r++) df[I2 - 1][r] += t[r];
            } else {
#pragma unroll
                for (int r = 0; r < 6; r++) df[I2 - 1][r] = t[r];
            }
            grad_bwd<J, I2 - 1, QD>(f, df, dc);
        }
    }
    // dcq[i] = d c_i / d q_J, dcd[i] = d c_i / d qd_J (v, a, f from rnea() at the solved qdd): one derivative column
    template<int J> GATO_DEV void rnea_grad_col(const float* qd, const float (*v)[6], const float (*a)[6], const float (*f)[6], float* dcq,
                                                float* dcd) const
    {
        // Each pass in its own basic block (a branch the compiler cannot fold): instruction selection and scheduling work per block,
        // so the two independent passes are not interleaved for ILP (which kept 440 registers live instead of 270).
        {
            float dv[6], da[6], df[NQ][6];
            if (opaque_true()) grad_fwd<J, J, false>(qd, v, a, dv, da, df);
            if (opaque_true()) grad_bwd<J, NQ - 1, false>(f, df, dcq);
        }
        {
            float dv[6], da[6], df[NQ][6];
            if (opaque_true()) grad_fwd<J, J, true>(qd, v, a, dv, da, df);
            if (opaque_true()) grad_bwd<J, NQ - 1, true>(f, df, dcd);
        }
    }
    // column J of [dqdd/dq | dqdd/dqd | M^-1]: the three nq-vectors D[J], D[nq+J], D[2nq+J] of the compact KKT storage
    // Columns of [dqdd/dq | dqdd/dqd | M^-1] after ONE evaluation of the common prefix (M^-1, RNEA, qdd, RNEA at qdd):
    // `emit(J, colq, cold, colm)` receives each column, `after_qdd(qdd)` runs as soon as the accelerations are known (the defect
    // c_{k+1} is formed there by the task that owns column 0).
    template<int J, class E> GATO_DEV void grad_column(const float* qd, const float (*v)[6], const float (*a)[6], const float (*f)[6],
                                                       const MinvT& Mi, E&& emit) const
    {
        float dcq[NQ], dcd[NQ];
        rnea_grad_col<J>(qd, v, a, f, dcq, dcd);
        if (opaque_true()) {
            float colq[NQ], cold[NQ], colm[NQ];
#pragma unroll
            for (int r = 0; r < NQ; r++) {
                float s1 = 0.f, s2 = 0.f;
#pragma unroll
                for (int c = 0; c < NQ; c++) {
                    s1 += Mi.sym(r, c) * dcq[c];
                    s2 += Mi.sym(r, c) * dcd[c];
                }
                colq[r] = -s1;
                cold[r] = -s2;
                colm[r] = Mi.sym(r, J);
            }
            emit(J, colq, cold, colm);
        }
    }
    // columns JA and JB (JB == JA: one column) after one evaluation of the common prefix
    template<int JA, int JB, int JC, class E, class F> GATO_DEV void fd_grad_columns(const float* qd, const float* u, const float* fext, E&& emit,
                                                                                      F&& after_qdd) const
    {
        // phases in basic blocks of their own (see rnea_grad_col)
        MinvT Mi;
        if (opaque_true()) minv(Mi);
        float v[NQ][6], a[NQ][6], f[NQ][6], qdd[NQ];
        if (opaque_true()) {
            rnea_lean(qd, fext, f);  // bias forces only: one body of v, a live
            fd_finish(Mi, u, f, qdd);
            after_qdd(qdd);
        }
        if (opaque_true()) rnea(qd, qdd, fext, v, a, f);
        grad_column<JA>(qd, v, a, f, Mi, emit);
        if constexpr (JB != JA) grad_column<JB>(qd, v, a, f, Mi, emit);
        if constexpr (JC >= 0) grad_column<JC>(qd, v, a, f, Mi, emit);
    }
    // ---- forward kinematics: e = origin of the last joint frame, Jc[j] = d e / d q_j ------------------------------
    // p_{n-1} = r_{n-1}; p_i = r_i + R_i p_{i+1},  R_i = E_i^T      (chain of Xhom products, indy7_grid.cuh:1834-1901)
    template<int K> GATO_DEV void fk_chain(float* p) const
    {
        if constexpr (K >= 0) {
            float t[3];
            ETmul<K>(p, t);
            p[0] = M::R[K][0] + t[0]; p[1] = M::R[K][1] + t[1]; p[2] = M::R[K][2] + t[2];
            fk_chain<K - 1>(p);
        }
    }
    GATO_DEV void ee_pos(float* e) const
    {
        e[0] = M::R[NQ - 1][0]; e[1] = M::R[NQ - 1][1]; e[2] = M::R[NQ - 1][2];
        fk_chain<NQ - 2>(e);
    }
    template<int K> GATO_DEV void rot_chain(float* d) const  // d <- R_K d, K..0
    {
        if constexpr (K >= 0) {
            float t[3];
            ETmul<K>(d, t);
            d[0] = t[0]; d[1] = t[1]; d[2] = t[2];
            rot_chain<K - 1>(d);
        }
    }
    // tails: pt[j] = position of the EE origin expressed in frame j's CHILD side, i.e. p_{j+1} above (pt[NQ-1] = 0)
    template<int K> GATO_DEV void fk_tails(float (*pt)[3], float* p) const
    {
        if constexpr (K >= 0) {
            pt[K][0] = p[0]; pt[K][1] = p[1]; pt[K][2] = p[2];  // p_{K+1}
            float t[3];
            ETmul<K>(p, t);
            p[0] = M::R[K][0] + t[0]; p[1] = M::R[K][1] + t[1]; p[2] = M::R[K][2] + t[2];
            fk_tails<K - 1>(pt, p);
        }
    }
    template<int J> GATO_DEV void jac_cols(const float (*pt)[3], float (*Jc)[3]) const
    {
        // d/dq_J [R_J p_{J+1}] = E0_J^T dEz^T(q_J) p_{J+1};   dEz^T u = (-s u0 - c u1, c u0 - s u1, 0)
        float w[3], d[3];
        w[0] = -sn[J] * pt[J][0] - cs[J] * pt[J][1];
        w[1] = cs[J] * pt[J][0] - sn[J] * pt[J][1];
        w[2] = 0.f;
        E0Tmul<J>(w, d);
        rot_chain<J - 1>(d);
        Jc[J][0] = d[0]; Jc[J][1] = d[1]; Jc[J][2] = d[2];
        if constexpr (J + 1 < NQ) jac_cols<J + 1>(pt, Jc);
    }
    GATO_DEV void ee_jac(float* e, float (*Jc)[3]) const
    {
        float pt[NQ][3];
        float p[3] = {0.f, 0.f, 0.f};  // p_NQ: the EE is the origin of the last frame
        fk_tails<NQ - 1>(pt, p);
        e[0] = p[0]; e[1] = p[1]; e[2] = p[2];
        jac_cols<0>(pt, Jc);
    }
};

}  // namespace gato
