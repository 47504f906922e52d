// mpc_forces_qp.h -- FORCES-mode solve: one SQP step = one stage-structured QP, per instance.
//
// What `solver.solve(problem)` of ForcesproOptimizer (MPC_Planner/optimizer.py:196-245, 326) asks of FORCESPRO: SQP with ONE
// quadratic programme per call (`sqp_nlp.maxqps = 1`), BFGS Hessian initialised to 2.5 I (never updated within a single QP),
// Hessian regularisation 5e-6.  The generated solver is a closed binary, so the QP is solved here by a primal-dual interior
// point method (Mehrotra predictor-corrector) whose Newton systems are eliminated stage by stage (Riccati recursion with
// the dense RK4 sensitivities A_k (5x5), B_k (5x2)):
//
//    min_dz  sum_k grad f_k(zbar_k)' dz_k + 1/2 dz_k' H_k dz_k        (H_k diagonal: forces_hessian_diag)
//    s.t.    x_1 = xinit,   x_{k+1} = c(zbar_k) + C_k dz_k,   lb <= zbar + dz <= ub,   hl <= h(zbar_k) + J_k dz_k <= hu
//
// (z = [deltaDot, aLong, x, y, delta, v, psi]; stage functions: forces_stage_functions below = FORCESNLPsolver_model.c, row a11.)
//
// Mapping: ONE THREAD PER (instance, stage).  Everything that is local to a stage -- building the QP, residuals, condensing
// the inequality rows into the stage Hessian, slack / multiplier steps, step lengths, updates -- runs for all stages at once;
// only the Riccati recursion itself is a chain over the stages, walked by handing the turn from stage to stage inside the
// workgroup (cost-to-go and step travel through the workspace rows of the neighbouring stage).  Per-instance sums / minima over
// the stages are combined in stage order, so the CPU driver below (forces_qp_instance: the same phase functions called stage
// after stage, used by the emulation harness) and the GPU kernel (k_forces_qp in mpcgpu.hip) produce the same numbers.
// The workspace is [stage][row][Bp]: the threads of one stage read consecutive doubles per row.
// Shared by the HIP kernel (mpcgpu.hip) and the CPU emulation harness (tests/emu); oracle: oracle/forces_qp_numpy.py.
#pragma once
#include <math.h>
#include <stdint.h>

#include "mpc_stage_math.h"

namespace mpc {

struct ForcesQpArgs {
    int32_t B, Bp, N, max_it;
    double dt, l, wb, rho, tol, tol_mu;             // converged: residuals <= tol and complementarity gap <= tol_mu
    double hd[7], hdN[7];                           // diagonal QP Hessian of a stage / of the last stage (see forces_hessian_diag)
    double Q[5], R[2], Pt[5];
    double lb[7], ub[7], hl[10], hu[10];          // +-1e300 and beyond = absent
    const double* zbar;                           // [B, N, 7]  problem["x0"]
    const double* params;                         // [B, N, 10] problem["all_parameters"]
    const double* xinit;                          // [B, 5]     problem["xinit"]
    double* z_out;                                // [B, N, 7]  output x01..xN
    int32_t* iters;                               // [B]        info.it
    int32_t* status;                              // [B]        exitflag: 1 solved, 0 iteration limit, -6 NaN, -7 infeasible QP (FORCESNLPsolver.h:68-106)
    double* kkt;                                  // [B]        final residual (max of dual / primal / equality / complementarity)
    double* ws;                                   // [FQP_ROWS * N][Bp]
    uint32_t ws_bytes;                            // its size (device: buffer addressing, below 4 GiB)
};

// per-stage workspace rows
enum FqpRow {
    FQ_C = 0,                      // 35  RK4 Jacobian C = [B A] (5x7, row-major)
    FQ_E = FQ_C + 35,              // 5   c(zbar_k) - zbar_{k+1}[2:7]
    FQ_G = FQ_E + 5,               // 7   grad f
    FQ_D = FQ_G + 7,               // 34  right-hand sides d of G dz <= d: [7 lower bounds | 7 upper bounds | 10 h upper | 10 h lower]
    FQ_W = FQ_D + 34,              // 7   dz
    FQ_S = FQ_W + 7,               // 34  slacks
    FQ_LAM = FQ_S + 34,            // 34  inequality multipliers
    FQ_PI = FQ_LAM + 34,           // 5   multipliers of the stage's equality block
    FQ_K = FQ_PI + 5,              // 10  feedback gain (2x5)
    FQ_HUI = FQ_K + 10,            // 3   inverse of Huu (symmetric 2x2)
    FQ_HXU = FQ_HUI + 3,           // 10  Phi_ux (2x5) of the condensed stage Hessian
    FQ_PHU = FQ_HXU + 10,          // 3   Phi_uu (symmetric 2x2)
    FQ_PE = FQ_PHU + 3,            // 5   P_{k+1} eps_k
    FQ_P = FQ_PE + 5,              // 15  cost-to-go matrix (symmetric 5x5, upper)
    FQ_DW = FQ_P + 15,             // 7   Newton step (affine, then final)
    FQ_DS = FQ_DW + 7,             // 34
    FQ_DL = FQ_DS + 34,            // 34
    FQ_DPI = FQ_DL + 34,           // 5   new equality multipliers of the step
    FQ_PV = FQ_DPI + 5,            // 5   cost-to-go vector p_k (handed to stage k - 1 in the backward recursion)
    FQ_DXIN = FQ_PV + 5,           // 5   dx_k as handed over by stage k - 1 in the forward recursion
    FQ_JHS = FQ_DXIN + 5,          // 30  Jacobian of h, nonzeros only: friction row (columns 1, 4, 5), nine distance rows (columns 2, 3, 6)
    FQ_ROWS = FQ_JHS + 30
};
// the rows the recursions hand from stage to stage, and the stage's RK4 Jacobian, live in LDS on the device: [row][thread],
// the neighbouring stage of the same instance is IB threads away (on the host they are workspace rows like the others)
// (round 4: the Jacobian of h -- 30 doubles per stage, constant during a solve and visited by every pass over the inequality rows, ~6 per
//  interior-point iteration -- lives in LDS too: it was a fifth of the kernel's HBM traffic)
enum FqLocal { FL_C = 0, FL_P = 35, FL_PV = 50, FL_DXIN = 55, FL_JHS = 60, FL_ROWS = 90 };
constexpr int FQ_MI = 34;

// Diagonal Hessian of the QP.  mode 0 (default): the exact Hessian of the reference's least-squares cost, 2 diag(R, Q) per
// stage and 2 diag(0, P) at the last one (Gauss-Newton SQP: the constraint curvature is dropped), plus the reference's
// regularisation 5e-6.  mode 1: the literal FORCESPRO options of optimizer.py:234-236, `bfgs_init = 2.5 I` -- with one QP
// per call and a guess that is never refreshed (optimizer.py:264-274) the BFGS matrix would stay at its initial value;
// that Hessian under-weights the tracking terms by two orders of magnitude and the closed loop runs away at full throttle,
// which the recorded forcespro runs do not do -- the binary evidently does something else, so the default is mode 0.
inline void forces_hessian_diag(int mode, const double* Q, const double* R, const double* Pt, double* hd, double* hdN) {
    const double reg = 5e-6;
    for (int i = 0; i < 7; ++i) {
        if (mode == 1) { hd[i] = 2.5 + reg; hdN[i] = 2.5 + reg; continue; }
        hd[i] = (i < 2 ? 2.0 * R[i] : 2.0 * Q[i - 2]) + reg;
        hdN[i] = (i < 2 ? 0.0 : 2.0 * Pt[i - 2]) + reg;
    }
}

MPC_HD bool fq_fin_lo(double v) { return v > -1e300; }
MPC_HD bool fq_fin_hi(double v) { return v < 1e300; }
MPC_HD int fq_sidx(int i, int j) { return (i <= j) ? i * 5 - i * (i - 1) / 2 + (j - i) : j * 5 - j * (j - 1) / 2 + (i - j); }

// is inequality row q of stage k present?  rows: [0,7) z_i >= lb_i, [7,14) z_i <= ub_i, [14,24) h_j <= hu_j, [24,34) h_j >= hl_j
MPC_HD bool fq_row_on(const ForcesQpArgs& A, int k, int q) {
    if (q < 7) return fq_fin_lo(A.lb[q]) && !(k == 0 && q >= 2);
    if (q < 14) return fq_fin_hi(A.ub[q - 7]) && !(k == 0 && q - 7 >= 2);
    if (q < 24) return fq_fin_hi(A.hu[q - 14]);
    return (q - 24) > 0 && fq_fin_lo(A.hl[q - 24]);              // the lower bound 0 of the friction row (a sum of squares) is vacuous
}

// FORCES-mode stage functions (row a11), one (z, p) pair: cost gradient, RK4 step + Jacobian, inequalities + Jacobian
// (device: every loop over a fixed range is unrolled, so that the small arrays of the phases are registers with literal indices -- the
//  RK4 sensitivities below are mostly structural zeros and ones that fold away -- instead of dynamically indexed scratch arrays)
#if defined(__HIP_DEVICE_COMPILE__)
#define FQ_UNROLL _Pragma("unroll")
#else
#define FQ_UNROLL
#endif
MPC_HD void forces_ode_eval(const double* x, const double* u, double l, double* f, double* F4, double& F42, double& F43) {
    double sn, cs;
    mpc_sincos(x[4], sn, cs);
    const double td = mpc_tan(x[2]);
    f[0] = x[3] * cs; f[1] = x[3] * sn; f[2] = u[0]; f[3] = u[1]; f[4] = x[3] / l * td;
    F4[0] = cs; F4[1] = -x[3] * sn; F4[2] = sn; F4[3] = x[3] * cs;
    F42 = x[3] / l * (1.0 + td * td);
    F43 = td / l;
}
MPC_HD void forces_stage_functions(const ForcesQpArgs& A, const double* z, const double* p, bool terminal, double& fval, double* gf,
                                   double* c, double* jc /*35*/, double* h /*10*/, double* jh /*70*/) {
    const double* w = terminal ? A.Pt : A.Q;
    const double r[5] = {z[2] - p[0], z[3] - p[1], z[4], z[5] - p[2], z[6] - p[3]};
    fval = 0.0;
    gf[0] = gf[1] = 0.0;
    FQ_UNROLL
    for (int i = 0; i < 5; ++i) { fval += w[i] * r[i] * r[i]; gf[2 + i] = 2.0 * w[i] * r[i]; }
    if (!terminal) {
        fval += A.R[0] * z[0] * z[0] + A.R[1] * z[1] * z[1];
        gf[0] = 2.0 * A.R[0] * z[0];
        gf[1] = 2.0 * A.R[1] * z[1];
    }
    if (!terminal && c != nullptr) {
        const double* u = z;
        const double* x = z + 2;
        double xs[5], kk[5], acc[5], dk[35], dacc[35], dxs[35];
        FQ_UNROLL
        for (int i = 0; i < 5; ++i) {
            xs[i] = x[i];
            acc[i] = 0.0;
            FQ_UNROLL
            for (int j = 0; j < 7; ++j) { dxs[i * 7 + j] = (j == i + 2) ? 1.0 : 0.0; dacc[i * 7 + j] = 0.0; dk[i * 7 + j] = 0.0; }
        }
        const double aw[4] = {0.0, 0.5, 0.5, 1.0}, bw[4] = {1.0, 2.0, 2.0, 1.0};
        FQ_UNROLL
        for (int s = 0; s < 4; ++s) {
            if (s > 0) {
                FQ_UNROLL
                for (int i = 0; i < 5; ++i) {
                    xs[i] = x[i] + aw[s] * A.dt * kk[i];
                    FQ_UNROLL
                    for (int j = 0; j < 7; ++j) dxs[i * 7 + j] = ((j == i + 2) ? 1.0 : 0.0) + aw[s] * A.dt * dk[i * 7 + j];
                }
            }
            double F4[4], F42, F43;
            forces_ode_eval(xs, u, A.l, kk, F4, F42, F43);
            FQ_UNROLL
            for (int j = 0; j < 7; ++j) {
                dk[0 * 7 + j] = F4[0] * dxs[3 * 7 + j] + F4[1] * dxs[4 * 7 + j];
                dk[1 * 7 + j] = F4[2] * dxs[3 * 7 + j] + F4[3] * dxs[4 * 7 + j];
                dk[2 * 7 + j] = (j == 0) ? 1.0 : 0.0;
                dk[3 * 7 + j] = (j == 1) ? 1.0 : 0.0;
                dk[4 * 7 + j] = F42 * dxs[2 * 7 + j] + F43 * dxs[3 * 7 + j];
            }
            FQ_UNROLL
            for (int i = 0; i < 5; ++i) {
                acc[i] += bw[s] * kk[i];
                FQ_UNROLL
                for (int j = 0; j < 7; ++j) dacc[i * 7 + j] += bw[s] * dk[i * 7 + j];
            }
        }
        FQ_UNROLL
        for (int i = 0; i < 5; ++i) {
            c[i] = x[i] + A.dt / 6.0 * acc[i];
            FQ_UNROLL
            for (int j = 0; j < 7; ++j) jc[i * 7 + j] = ((j == i + 2) ? 1.0 : 0.0) + A.dt / 6.0 * dacc[i * 7 + j];
        }
    }
    if (h != nullptr) {
        FQ_UNROLL
        for (int i = 0; i < 70; ++i) jh[i] = 0.0;
        const double td = mpc_tan(z[4]);
        const double q = z[5] * z[5] * td / A.wb;                 // v * psi_dot
        h[0] = z[1] * z[1] + q * q;
        jh[1] = 2.0 * z[1];
        jh[4] = 2.0 * q * z[5] * z[5] * (1.0 + td * td) / A.wb;
        jh[5] = 2.0 * q * 2.0 * z[5] * td / A.wb;
        double sn, cs;
        mpc_sincos(z[6], sn, cs);
        FQ_UNROLL
        for (int e = 0; e < 3; ++e) {
            const double sg = (e == 0) ? 0.0 : (e == 1 ? 1.0 : -1.0);
            const double ex = z[2] + sg * A.rho * cs, ey = z[3] + sg * A.rho * sn;
            FQ_UNROLL
            for (int j = 0; j < 3; ++j) {
                const double dx = ex - p[4 + 2 * j], dy = ey - p[5 + 2 * j];
                const int row = 1 + 3 * e + j;
                h[row] = dx * dx + dy * dy;
                jh[row * 7 + 2] = 2.0 * dx;
                jh[row * 7 + 3] = 2.0 * dy;
                jh[row * 7 + 6] = 2.0 * dx * (-sg * A.rho * sn) + 2.0 * dy * (sg * A.rho * cs);
            }
        }
    }
}

// Workspace element (stage k_, row row_) of the thread's instance.  Device: a BUFFER access -- per-thread byte offset of (stage, instance)
// in the vector offset, the row's offset row_ * Bp * 8 (wave-uniform; row_ is a literal after unrolling) in the scalar offset.  With
// plain 64-bit pointers the compiler precomputed one address per row (224 pairs of registers), spilled them all to scratch before the
// iteration loop and reloaded one before every access: 448 scratch stores and a scratch load in front of each of ~1 200 global accesses.
#if defined(__HIP_DEVICE_COMPILE__)
// (the descriptor is built ONCE, by the kernel, into the thread state: rebuilt at each of the ~400 access sites it cost three s_mov and an
//  s_and per site, a quarter of the kernel's scalar instructions.  Rows are Bp * 8 bytes apart: no two 8-byte stores are neighbours, so
//  the shared descriptor gives the compiler nothing to merge into the unguarded 16-byte form, see ws_store2)
struct FqWsRef {
    const __amdgpu_buffer_rsrc_t& rs;
    uint32_t voff, soff;
    __device__ __forceinline__ operator double() const {
        return __builtin_bit_cast(double, __builtin_amdgcn_raw_buffer_load_b64(rs, (int)voff, mpc_uni(soff), 0));
    }
    __device__ __forceinline__ double operator=(double x) const {
        __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(mpc_v2u, x), rs, (int)voff, mpc_uni(soff), 0);
        return x;
    }
    __device__ __forceinline__ double operator=(const FqWsRef& o) const { return (*this = (double)o); }
};
#define FQW(k_, row_) FqWsRef{c.rs, ((uint32_t)(k_) * (uint32_t)FQ_ROWS * (uint32_t)A.Bp + (uint32_t)c.b) * 8u, (uint32_t)(row_) * (uint32_t)A.Bp * 8u}
#else
#define FQW(k_, row_) A.ws[((size_t)(k_) * FQ_ROWS + (size_t)(row_)) * (size_t)A.Bp + (size_t)c.b]
#endif
// FQL(dk, GRP, i): entry i of row group GRP (C, P, PV, DXIN) of stage k + dk of the same instance
#if defined(__HIP_DEVICE_COMPILE__)
#define FQL(dk_, grp_, i_) c.loc[(FL_##grp_ + (i_)) * c.T + c.t + (dk_) * c.IB]
#else
#define FQL(dk_, grp_, i_) FQW(c.k + (dk_), FQ_##grp_ + (i_))
#endif

// per-(instance, stage) thread state that lives in registers across the phases of one solve
struct FqCtx {
    int b, k;
    bool valid;          // k < N and b < B
    bool run;            // the instance is still iterating (the same for all stages of an instance)
    int it, status;
    double gscale, kkt;
    double rho[7], Phi[28], eps[5];      // what newton_prep hands to the stage's turn in the backward recursion
    // what a turn of the recursions leaves for the later turns of the same stage -- the corrector's backward turn and both forward turns
    // reuse the predictor's factorisation.  In the thread's registers: a turn is ONE stage's lanes working while everybody else waits
    // at the barrier, so a workspace round trip inside it (~2 us to the L2 and back) was most of the turn, 40 times per iteration.
    double hi[3], Puu[3], Kk[10], Pxu[10], pe[5], kf[2], dx0[5];
    double* loc;         // device: LDS rows [FL_ROWS][T]
    int t, T, IB;        // device: thread index, threads per workgroup, instances per workgroup
#if defined(__HIP_DEVICE_COMPILE__)
    __amdgpu_buffer_rsrc_t rs;           // buffer descriptor of the workspace
#endif
};
struct FqRed { double a, b, c, d; };      // partial of one stage / combined value of one instance (meaning per phase)

// Visits the inequality rows of stage k that are present: f(q, n, c0, c1, c2, v0, v1, v2) with the n (1 or 3) nonzeros of row q of
// G_k -- columns c0 < c1 < c2, values v*.  Bound rows have one nonzero (-1 / +1 on their variable); the rows of h three
// (friction: columns 1, 4, 5; squared distances: columns 2, 3, 6), negated for the lower-bound form.  The column numbers are
// literals at every call of f, so register arrays indexed by them stay in registers.  Row order: bounds of variable 0, 1, ..., 6
// (lower, upper), friction (upper, lower), distances 1..9 (upper, lower).
template <class F>
MPC_HD void fq_each_row(const ForcesQpArgs& A, const FqCtx& c, F f) {
    const int k = c.k;
    FQ_UNROLL
    for (int i = 0; i < 7; ++i) {
        if (fq_row_on(A, k, i)) f(i, 1, i, 0, 0, -1.0, 0.0, 0.0);
        if (fq_row_on(A, k, 7 + i)) f(7 + i, 1, i, 0, 0, 1.0, 0.0, 0.0);
    }
    {
        const double a = FQL(0, JHS, 0), b2 = FQL(0, JHS, 1), d = FQL(0, JHS, 2);
        if (fq_row_on(A, k, 14)) f(14, 3, 1, 4, 5, a, b2, d);
        if (fq_row_on(A, k, 24)) f(24, 3, 1, 4, 5, -a, -b2, -d);
    }
    FQ_UNROLL
    for (int j = 1; j < 10; ++j) {
        const double a = FQL(0, JHS, 3 * j), b2 = FQL(0, JHS, 3 * j + 1), d = FQL(0, JHS, 3 * j + 2);
        if (fq_row_on(A, k, 14 + j)) f(14 + j, 3, 2, 3, 6, a, b2, d);
        if (fq_row_on(A, k, 24 + j)) f(24 + j, 3, 2, 3, 6, -a, -b2, -d);
    }
}
#define FQ_ROW_ARGS int q, int n, int c0, int c1, int c2, double v0, double v1, double v2
#define FQ_PHI(i_, j_) ((i_) * 7 - (i_) * ((i_) - 1) / 2 + ((j_) - (i_)))

// ---- phase: build the QP of stage k at zbar.  partial.a = largest cost-gradient entry of the stage
MPC_HD void fq_build(const ForcesQpArgs& A, FqCtx& c, FqRed& part) {
    part = FqRed{1.0, 0.0, 0.0, 0.0};
    if (!c.valid) return;
    const int N = A.N, k = c.k;
    double z[7], p[10], gf[7], cc[5], jc[35], h[10], jh[70], fv;
    FQ_UNROLL
    for (int i = 0; i < 7; ++i) z[i] = A.zbar[((size_t)c.b * N + k) * 7 + i];
    FQ_UNROLL
    for (int i = 0; i < 10; ++i) p[i] = A.params[((size_t)c.b * N + k) * 10 + i];
    const bool term = (k == N - 1);
    forces_stage_functions(A, z, p, term, fv, gf, term ? nullptr : cc, jc, h, jh);
    double gs = 1.0;
    FQ_UNROLL
    for (int i = 0; i < 7; ++i) { FQW(k, FQ_G + i) = gf[i]; FQW(k, FQ_W + i) = 0.0; gs = fmax(gs, fabs(gf[i])); }
    FQL(0, JHS, 0) = jh[1]; FQL(0, JHS, 1) = jh[4]; FQL(0, JHS, 2) = jh[5];
    FQ_UNROLL
    for (int j = 1; j < 10; ++j) { FQL(0, JHS, 3 * j) = jh[j * 7 + 2]; FQL(0, JHS, 3 * j + 1) = jh[j * 7 + 3]; FQL(0, JHS, 3 * j + 2) = jh[j * 7 + 6]; }
    if (!term) {
        FQ_UNROLL
        for (int i = 0; i < 35; ++i) FQL(0, C, i) = jc[i];
        FQ_UNROLL
        for (int i = 0; i < 5; ++i) FQW(k, FQ_E + i) = cc[i] - (double)A.zbar[((size_t)c.b * N + k + 1) * 7 + 2 + i];
    }
    FQ_UNROLL
    for (int i = 0; i < 5; ++i) FQW(k, FQ_PI + i) = 0.0;
    FQ_UNROLL
    for (int q = 0; q < FQ_MI; ++q) {
        double d = 0.0;
        if (q < 7) d = z[q] - A.lb[q];
        else if (q < 14) d = A.ub[q - 7] - z[q - 7];
        else if (q < 24) d = A.hu[q - 14] - h[q - 14];
        else d = h[q - 24] - A.hl[q - 24];
        const bool on = fq_row_on(A, k, q);
        FQW(k, FQ_D + q) = on ? d : 0.0;
        FQW(k, FQ_S + q) = on ? fmax(d, 1.0) : 1.0;
        FQW(k, FQ_LAM + q) = on ? mpc_rcp(fmax(d, 1.0)) : 0.0;          // centred start: s * lam = 1 on every row
    }
    part.a = gs;
}

// ---- phase: residuals of stage k.  partial: a = sum s lam, b = rows, c = max primal / equality residual, d = max dual residual
MPC_HD void fq_residual(const ForcesQpArgs& A, const FqCtx& c, FqRed& part) {
    part = FqRed{0.0, 0.0, 0.0, 0.0};
    if (!(c.valid && c.run)) return;
    const int N = A.N, k = c.k;
    double w[7], rd[7], mu = 0.0, rmax = 0.0, rdmax = 0.0;
    int M = 0;
    const double* hd = (k == N - 1) ? A.hdN : A.hd;
    FQ_UNROLL
    for (int i = 0; i < 7; ++i) { w[i] = FQW(k, FQ_W + i); rd[i] = hd[i] * w[i] + (double)FQW(k, FQ_G + i); }
    FQ_UNROLL
    for (int i = 0; i < 5; ++i) rd[2 + i] += (double)FQW(k, FQ_PI + i);
    if (k < N - 1)
        FQ_UNROLL
        for (int j = 0; j < 7; ++j) {
            double t = 0.0;
            FQ_UNROLL
            for (int r = 0; r < 5; ++r) t += (double)FQL(0, C, r * 7 + j) * (double)FQW(k + 1, FQ_PI + r);
            rd[j] -= t;
        }
    fq_each_row(A, c, [&](FQ_ROW_ARGS) {
        const double s = FQW(k, FQ_S + q), lam = FQW(k, FQ_LAM + q);
        double gw = v0 * w[c0];
        rd[c0] += lam * v0;
        if (n == 3) { gw += v1 * w[c1] + v2 * w[c2]; rd[c1] += lam * v1; rd[c2] += lam * v2; }
        rmax = fmax(rmax, fabs(gw + s - (double)FQW(k, FQ_D + q)));
        mu += s * lam;
        ++M;
    });
    FQ_UNROLL
    for (int i = 0; i < 7; ++i) rdmax = fmax(rdmax, fabs(rd[i]));
    FQ_UNROLL
    for (int i = 0; i < 5; ++i) {
        double re;
        if (k == 0) re = w[2 + i] - ((double)A.xinit[(size_t)c.b * 5 + i] - (double)A.zbar[((size_t)c.b * N) * 7 + 2 + i]);
        else {
            re = w[2 + i] - (double)FQW(k - 1, FQ_E + i);
            FQ_UNROLL
            for (int j = 0; j < 7; ++j) re -= (double)FQL(-1, C, i * 7 + j) * (double)FQW(k - 1, FQ_W + j);
        }
        rmax = fmax(rmax, fabs(re));
    }
    part = FqRed{mu, (double)M, rmax, rdmax};
}
MPC_HD void fq_residual_combine(FqRed& a, const FqRed& p) { a.a += p.a; a.b += p.b; a.c = fmax(a.c, p.c); a.d = fmax(a.d, p.d); }

// termination test on the combined residuals (every thread of the instance takes the same decision); returns mu
MPC_HD double fq_decide(const ForcesQpArgs& A, FqCtx& c, const FqRed& tot) {
    if (!(c.valid && c.run)) return 0.0;
    const double M = tot.b, rmax = tot.c, rdmax = tot.d;
    const double mu = M > 0 ? tot.a / M : 0.0;
    c.kkt = fmax(fmax(rmax, rdmax / c.gscale), mu);
    if (!(rmax == rmax) || !(rdmax == rdmax) || !(mu == mu) || c.kkt > 1e300) { c.kkt = NAN; c.status = -6; c.run = false; return mu; }
    // Stage-wise elimination condenses the inequality rows into the stage Hessians with weights lam/s, which grow like
    // 1/mu for active rows; past mu ~ 1e-7 the Riccati recursion loses the small entries next to them.  The targets are
    // therefore the accuracy an SQP step needs (FORCESPRO is called with tolerances of 1e-1, optimizer.py:272-273):
    // primal / equality residuals <= tol (metres, m/s, rad), dual residual <= tol relative to the largest cost gradient.
    if (rmax <= A.tol && rdmax <= A.tol * c.gscale && mu <= A.tol_mu) { c.status = 1; c.run = false; return mu; }
    if (mu > 1e6) { c.status = -7; c.run = false; return mu; }                   // multipliers diverge: the linearised constraints are inconsistent
    if (c.it >= A.max_it) { c.status = 0; c.run = false; return mu; }
    return mu;
}

// ---- phase (all stages at once): the stage-local part of a Newton step for the complementarity target rc (per row:
// rc_q = s lam [+ ds_aff dl_aff - sigma mu]): condensed gradient rho, condensed Hessian Phi (predictor only; the corrector
// reuses the factorisation), eps_k = e_k - w_{k+1}[2:7] + C_k w_k
MPC_HD void fq_newton_prep(const ForcesQpArgs& A, FqCtx& c, bool corr, double sigma_mu) {
    if (!(c.valid && c.run)) return;
    const int N = A.N, k = c.k;
    double w[7];
    FQ_UNROLL
    for (int i = 0; i < 7; ++i) w[i] = FQW(k, FQ_W + i);
    const double* hd = (k == N - 1) ? A.hdN : A.hd;
    FQ_UNROLL
    for (int i = 0; i < 7; ++i) c.rho[i] = hd[i] * w[i] + (double)FQW(k, FQ_G + i);
    if (!corr) {
        FQ_UNROLL
        for (int i = 0; i < 28; ++i) c.Phi[i] = 0.0;
        FQ_UNROLL
        for (int i = 0; i < 7; ++i) c.Phi[i * 7 - i * (i - 1) / 2] = hd[i];
    }
    fq_each_row(A, c, [&](FQ_ROW_ARGS) {
        const double s = FQW(k, FQ_S + q), lam = FQW(k, FQ_LAM + q), d = FQW(k, FQ_D + q);
        double gw = v0 * w[c0];
        if (n == 3) gw += v1 * w[c1] + v2 * w[c2];
        const double rp = gw + s - d;
        double rc = s * lam;
        if (corr) rc += (double)FQW(k, FQ_DS + q) * (double)FQW(k, FQ_DL + q) - sigma_mu;
        const double is = mpc_rcp(s), D = lam * is, cf = lam + D * rp - rc * is;          // (s > 0: one reciprocal per row)
        c.rho[c0] += cf * v0;
        if (n == 3) { c.rho[c1] += cf * v1; c.rho[c2] += cf * v2; }
        if (!corr) {
            c.Phi[FQ_PHI(c0, c0)] += D * v0 * v0;
            if (n == 3) {
                c.Phi[FQ_PHI(c0, c1)] += D * v0 * v1; c.Phi[FQ_PHI(c0, c2)] += D * v0 * v2;
                c.Phi[FQ_PHI(c1, c1)] += D * v1 * v1; c.Phi[FQ_PHI(c1, c2)] += D * v1 * v2;
                c.Phi[FQ_PHI(c2, c2)] += D * v2 * v2;
            }
        }
    });
    FQ_UNROLL
    for (int i = 0; i < 5; ++i) {
        c.eps[i] = 0.0;
        // (stage 0: the start of the forward sweep, dx_0 = xinit - x_0 of the current iterate)
        c.dx0[i] = (k == 0) ? -(w[2 + i] - ((double)A.xinit[(size_t)c.b * 5 + i] - (double)A.zbar[((size_t)c.b * N) * 7 + 2 + i])) : 0.0;
    }
    if (k < N - 1)
        FQ_UNROLL
        for (int i = 0; i < 5; ++i) {
            double t = (double)FQW(k, FQ_E + i) - (double)FQW(k + 1, FQ_W + 2 + i);
            FQ_UNROLL
            for (int j = 0; j < 7; ++j) t += (double)FQL(0, C, i * 7 + j) * w[j];
            c.eps[i] = t;
        }
}

// ---- the stage's turn in the backward recursion (stages N-1, N-2, ..., 0 one after the other): cost-to-go P_{k+1}, p_{k+1}
// come from the workspace rows of stage k + 1, this stage leaves P_k, p_k, the gains and k_ff in its own
MPC_HD void fq_newton_back(const ForcesQpArgs& A, FqCtx& c, bool corr) {
    if (!(c.valid && c.run)) return;
    const int N = A.N, k = c.k;
    const double* rho = c.rho;
    const double* Phi = c.Phi;
    const double* eps = c.eps;
    double Pp[15], pv[5];
    FQ_UNROLL
    for (int i = 0; i < 15; ++i) Pp[i] = (k < N - 1) ? (double)FQL(1, P, i) : 0.0;
    FQ_UNROLL
    for (int i = 0; i < 5; ++i) pv[i] = (k < N - 1) ? (double)FQL(1, PV, i) : 0.0;
    // Q-function blocks.  Stage N-1 has no successor
    double Huu[3], Hux[10], hu[2], hx[5];
    if (!corr) {
        Huu[0] = Phi[0]; Huu[1] = Phi[1]; Huu[2] = Phi[7];
        FQ_UNROLL
        for (int j = 0; j < 5; ++j) { Hux[j] = Phi[2 + j]; Hux[5 + j] = Phi[7 + 1 + j]; }
    }
    hu[0] = rho[0]; hu[1] = rho[1];
    FQ_UNROLL
    for (int i = 0; i < 5; ++i) hx[i] = rho[2 + i];
    double Cm[35];
    if (k < N - 1) {
        double q5[5];
        FQ_UNROLL
        for (int i = 0; i < 35; ++i) Cm[i] = FQL(0, C, i);
        if (!corr) {
            double Pe[5];
            FQ_UNROLL
            for (int i = 0; i < 5; ++i) {
                double t = 0.0;
                FQ_UNROLL
                for (int j = 0; j < 5; ++j) t += Pp[fq_sidx(i, j)] * eps[j];
                Pe[i] = t;
                c.pe[i] = t;
            }
            // the u-rows of C' (P C), column by column: m = column j of P C (5 numbers live at a time, not the 5 x 7 product), then
            // rows 0, 1 of C' against it
            FQ_UNROLL
            for (int j = 0; j < 7; ++j) {
                double m[5];
                FQ_UNROLL
                for (int i = 0; i < 5; ++i) {
                    double t = 0.0;
                    FQ_UNROLL
                    for (int r = 0; r < 5; ++r) t += Pp[fq_sidx(i, r)] * Cm[r * 7 + j];
                    m[i] = t;
                }
                FQ_UNROLL
                for (int i = 0; i < 2; ++i) {
                    if (j < i) continue;
                    double t = 0.0;
                    FQ_UNROLL
                    for (int r = 0; r < 5; ++r) t += Cm[r * 7 + i] * m[r];
                    if (j < 2) Huu[i + j] += t;
                    else Hux[i * 5 + (j - 2)] += t;
                }
            }
            FQ_UNROLL
            for (int i = 0; i < 5; ++i) q5[i] = Pe[i] + pv[i];
        } else {
            FQ_UNROLL
            for (int i = 0; i < 5; ++i) q5[i] = c.pe[i] + pv[i];
        }
        FQ_UNROLL
        for (int j = 0; j < 7; ++j) {
            double t = 0.0;
            FQ_UNROLL
            for (int r = 0; r < 5; ++r) t += Cm[r * 7 + j] * q5[r];
            if (j < 2) hu[j] += t; else hx[j - 2] += t;
        }
    }
    double hi[3], Kk[10], Puu[3], Pxu[10];
    if (!corr) {
        Puu[0] = Phi[0]; Puu[1] = Phi[1]; Puu[2] = Phi[7];
        FQ_UNROLL
        for (int j = 0; j < 5; ++j) { Pxu[j] = Phi[2 + j]; Pxu[5 + j] = Phi[8 + j]; }
        const double det = Huu[0] * Huu[2] - Huu[1] * Huu[1];
        const double idet = 1.0 / det;
        hi[0] = Huu[2] * idet; hi[1] = -Huu[1] * idet; hi[2] = Huu[0] * idet;
        FQ_UNROLL
        for (int j = 0; j < 5; ++j) {
            Kk[j] = -(hi[0] * Hux[j] + hi[1] * Hux[5 + j]);
            Kk[5 + j] = -(hi[1] * Hux[j] + hi[2] * Hux[5 + j]);
        }
        FQ_UNROLL
        for (int i = 0; i < 3; ++i) { c.hi[i] = hi[i]; c.Puu[i] = Puu[i]; }
        FQ_UNROLL
        for (int i = 0; i < 10; ++i) { c.Kk[i] = Kk[i]; c.Pxu[i] = Pxu[i]; }
    } else {
        FQ_UNROLL
        for (int i = 0; i < 3; ++i) { hi[i] = c.hi[i]; Puu[i] = c.Puu[i]; }
        FQ_UNROLL
        for (int i = 0; i < 10; ++i) { Kk[i] = c.Kk[i]; Pxu[i] = c.Pxu[i]; }
    }
    const double kf0 = -(hi[0] * hu[0] + hi[1] * hu[1]), kf1 = -(hi[1] * hu[0] + hi[2] * hu[1]);
    // Cost-to-go in the symmetric ("Joseph") form: with u = K x + kff and A_cl = A + B K,
    //    P_k = [K; I]' Phi [K; I] + A_cl' P+ A_cl,   p_k = rho_x + K' rho_u + (Phi_xu + K' Phi_uu) kff + A_cl' (P+ (B kff + eps) + p+)
    // -- sums of terms of one sign.  The textbook form Hxx - Hxu Huu^-1 Hux subtracts two numbers of size lam/s (1e10 and
    // more for active rows late in the iteration) and loses the O(1) entries that decide the step.
    double Acl[25], t5[5] = {0, 0, 0, 0, 0};
    if (k < N - 1) {
        double off[5];
        FQ_UNROLL
        for (int i = 0; i < 5; ++i) {
            off[i] = eps[i] + Cm[i * 7 + 0] * kf0 + Cm[i * 7 + 1] * kf1;
            FQ_UNROLL
            for (int j = 0; j < 5; ++j) Acl[i * 5 + j] = Cm[i * 7 + 2 + j] + Cm[i * 7 + 0] * Kk[j] + Cm[i * 7 + 1] * Kk[5 + j];
        }
        FQ_UNROLL
        for (int i = 0; i < 5; ++i) {
            double t = pv[i];
            FQ_UNROLL
            for (int j = 0; j < 5; ++j) t += Pp[fq_sidx(i, j)] * off[j];
            t5[i] = t;
        }
    }
    double pn[5];
    FQ_UNROLL
    for (int i = 0; i < 5; ++i) {
        double t = rho[2 + i] + Kk[i] * rho[0] + Kk[5 + i] * rho[1];
        t += (Pxu[i] + Kk[i] * Puu[0] + Kk[5 + i] * Puu[1]) * kf0 + (Pxu[5 + i] + Kk[i] * Puu[1] + Kk[5 + i] * Puu[2]) * kf1;
        if (k < N - 1)
            FQ_UNROLL
            for (int r = 0; r < 5; ++r) t += Acl[r * 5 + i] * t5[r];
        pn[i] = t;
    }
    if (!corr) {
        // column by column: pa = column j of P+ A_cl (5 numbers live at a time), entries (i <= j) of the new cost-to-go go straight out
        FQ_UNROLL
        for (int j = 0; j < 5; ++j) {
            double pa[5] = {0, 0, 0, 0, 0};
            if (k < N - 1) {
                FQ_UNROLL
                for (int i = 0; i < 5; ++i) {
                    double t = 0.0;
                    FQ_UNROLL
                    for (int r = 0; r < 5; ++r) t += Pp[fq_sidx(i, r)] * Acl[r * 5 + j];
                    pa[i] = t;
                }
            }
            FQ_UNROLL
            for (int i = 0; i <= j; ++i) {
                double t = Phi[(i + 2) * 7 - (i + 2) * (i + 1) / 2 + (j - i)];
                t += Pxu[i] * Kk[j] + Pxu[5 + i] * Kk[5 + j] + Kk[i] * Pxu[j] + Kk[5 + i] * Pxu[5 + j];
                t += Kk[i] * (Puu[0] * Kk[j] + Puu[1] * Kk[5 + j]) + Kk[5 + i] * (Puu[1] * Kk[j] + Puu[2] * Kk[5 + j]);
                if (k < N - 1)
                    FQ_UNROLL
                    for (int r = 0; r < 5; ++r) t += Acl[r * 5 + i] * pa[r];
                FQL(0, P, fq_sidx(i, j)) = t;
            }
        }
    }
    // the step of u_k is finished in the forward sweep; keep kff and p_k
    c.kf[0] = kf0;
    c.kf[1] = kf1;
    FQ_UNROLL
    for (int i = 0; i < 5; ++i) FQL(0, PV, i) = pn[i];
}

// ---- the stage's turn in the forward recursion (stages 0, 1, ..., N-1): dx_k arrives in the stage's own DW rows (written by
// stage k - 1), leaves dw_k, the new equality multipliers, and dx_{k+1} in the rows of stage k + 1
MPC_HD void fq_newton_fwd(const ForcesQpArgs& A, FqCtx& c) {
    if (!(c.valid && c.run)) return;
    const int N = A.N, k = c.k;
    double dx[5];
    FQ_UNROLL
    for (int i = 0; i < 5; ++i) dx[i] = (k == 0) ? c.dx0[i] : (double)FQL(0, DXIN, i);
    double du[2] = {c.kf[0], c.kf[1]};
    FQ_UNROLL
    for (int j = 0; j < 5; ++j) { du[0] += c.Kk[j] * dx[j]; du[1] += c.Kk[5 + j] * dx[j]; }
    const double dw[7] = {du[0], du[1], dx[0], dx[1], dx[2], dx[3], dx[4]};
    if (k < N - 1) {
        FQ_UNROLL
        for (int i = 0; i < 5; ++i) {
            // dx_{k+1} = C dw + eps,  eps = e_k - w_{k+1}[2:7] + C w_k  (newton_prep's: first, so that the next stage's turn can start)
            double t = c.eps[i];
            FQ_UNROLL
            for (int j = 0; j < 7; ++j) t += (double)FQL(0, C, i * 7 + j) * dw[j];
            FQL(1, DXIN, i) = t;
        }
    }
    FQ_UNROLL
    for (int i = 0; i < 7; ++i) FQW(k, FQ_DW + i) = dw[i];
    // new equality multipliers of block k: -(P_k dx_k + p_k)
    FQ_UNROLL
    for (int i = 0; i < 5; ++i) {
        double t = FQL(0, PV, i);
        FQ_UNROLL
        for (int j = 0; j < 5; ++j) t += (double)FQL(0, P, fq_sidx(i, j)) * dx[j];
        FQW(k, FQ_DPI + i) = -t;
    }
}

// ---- phase (all stages at once): slack and multiplier steps of the rows of stage k
MPC_HD void fq_newton_rows(const ForcesQpArgs& A, const FqCtx& c, bool corr, double sigma_mu) {
    if (!(c.valid && c.run)) return;
    const int k = c.k;
    double w[7], dw[7];
    FQ_UNROLL
    for (int i = 0; i < 7; ++i) { w[i] = FQW(k, FQ_W + i); dw[i] = FQW(k, FQ_DW + i); }
    fq_each_row(A, c, [&](FQ_ROW_ARGS) {
        const double s = FQW(k, FQ_S + q), lam = FQW(k, FQ_LAM + q), d = FQW(k, FQ_D + q);
        double gw = v0 * w[c0], gdw = v0 * dw[c0];
        if (n == 3) { gw += v1 * w[c1] + v2 * w[c2]; gdw += v1 * dw[c1] + v2 * dw[c2]; }
        const double rp = gw + s - d;
        double rc = s * lam;
        if (corr) rc += (double)FQW(k, FQ_DS + q) * (double)FQW(k, FQ_DL + q) - sigma_mu;
        const double ds = -rp - gdw;
        const double dl = -(rc + lam * ds) * mpc_rcp(s);
        FQW(k, FQ_DS + q) = ds;
        FQW(k, FQ_DL + q) = dl;
    });
}

// ---- phase: largest step to the boundary over the rows of stage k; partial.a = primal, .b = dual (both <= 1), scaled by `frac`
MPC_HD void fq_steplen(const ForcesQpArgs& A, const FqCtx& c, double frac, FqRed& part) {
    part = FqRed{1.0, 1.0, 0.0, 0.0};
    if (!(c.valid && c.run)) return;
    const int k = c.k;
    double a_p = 1.0, a_d = 1.0;
    FQ_UNROLL
    for (int q = 0; q < FQ_MI; ++q) {
        if (!fq_row_on(A, k, q)) continue;
        const double s = FQW(k, FQ_S + q), lam = FQW(k, FQ_LAM + q), ds = FQW(k, FQ_DS + q), dl = FQW(k, FQ_DL + q);
        if (ds < 0) a_p = fmin(a_p, -frac * s * mpc_rcp(ds));          // (normal, non-zero denominators: estimate + two Newton steps)
        if (dl < 0) a_d = fmin(a_d, -frac * lam * mpc_rcp(dl));
    }
    part.a = a_p;
    part.b = a_d;
}
MPC_HD void fq_steplen_combine(FqRed& a, const FqRed& p) { a.a = fmin(a.a, p.a); a.b = fmin(a.b, p.b); }

// ---- phase: complementarity after the affine step of length a_aff; partial.a = sum over the rows of stage k
MPC_HD void fq_mu_aff(const ForcesQpArgs& A, const FqCtx& c, double a_aff, FqRed& part) {
    part = FqRed{0.0, 0.0, 0.0, 0.0};
    if (!(c.valid && c.run)) return;
    const int k = c.k;
    double m = 0.0;
    FQ_UNROLL
    for (int q = 0; q < FQ_MI; ++q) {
        if (!fq_row_on(A, k, q)) continue;
        m += ((double)FQW(k, FQ_S + q) + a_aff * (double)FQW(k, FQ_DS + q)) * ((double)FQW(k, FQ_LAM + q) + a_aff * (double)FQW(k, FQ_DL + q));
    }
    part.a = m;
}
MPC_HD void fq_sum_combine(FqRed& a, const FqRed& p) { a.a += p.a; }
MPC_HD void fq_max_combine(FqRed& a, const FqRed& p) { a.a = fmax(a.a, p.a); }

// ---- phase: take the step
MPC_HD void fq_update(const ForcesQpArgs& A, FqCtx& c, double a_p, double a_d) {
    if (!(c.valid && c.run)) return;
    const int k = c.k;
    FQ_UNROLL
    for (int i = 0; i < 7; ++i) FQW(k, FQ_W + i) = (double)FQW(k, FQ_W + i) + a_p * (double)FQW(k, FQ_DW + i);
    FQ_UNROLL
    for (int i = 0; i < 5; ++i) FQW(k, FQ_PI + i) = (double)FQW(k, FQ_PI + i) + a_d * ((double)FQW(k, FQ_DPI + i) - (double)FQW(k, FQ_PI + i));
    FQ_UNROLL
    for (int q = 0; q < FQ_MI; ++q) {
        if (!fq_row_on(A, k, q)) continue;
        FQW(k, FQ_S + q) = (double)FQW(k, FQ_S + q) + a_p * (double)FQW(k, FQ_DS + q);
        FQW(k, FQ_LAM + q) = (double)FQW(k, FQ_LAM + q) + a_d * (double)FQW(k, FQ_DL + q);
    }
    ++c.it;
}

// ---- phase: output rows of stage k (the stage-0 thread also reports the instance)
MPC_HD void fq_output(const ForcesQpArgs& A, const FqCtx& c) {
    if (!c.valid) return;
    const int N = A.N, k = c.k;
    FQ_UNROLL
    for (int i = 0; i < 7; ++i) A.z_out[((size_t)c.b * N + k) * 7 + i] = (double)A.zbar[((size_t)c.b * N + k) * 7 + i] + (double)FQW(k, FQ_W + i);
    if (k == 0) {
        if (A.iters) A.iters[c.b] = c.it;
        if (A.status) A.status[c.b] = c.status;
        if (A.kkt) A.kkt[c.b] = c.kkt;
    }
}

#undef FQW
#undef FQL
#undef FQ_UNROLL
#undef FQ_ROW_ARGS
#undef FQ_PHI

#if !defined(__HIP_DEVICE_COMPILE__)
// One SQP step of instance b on the CPU (emulation harness): the phase functions above, stage after stage, with the per-instance
// reductions taken in stage order -- the order the kernel's reductions use as well.
inline void forces_qp_instance(const ForcesQpArgs& A, int b) {
    const int N = A.N;
    FqCtx* cs = new FqCtx[N];
    FqRed part, tot;
    for (int k = 0; k < N; ++k) { cs[k] = FqCtx{}; cs[k].b = b; cs[k].k = k; cs[k].valid = true; cs[k].run = true; cs[k].it = 0; cs[k].status = 0; cs[k].kkt = INFINITY; }
    tot = FqRed{1.0, 0, 0, 0};
    for (int k = 0; k < N; ++k) { fq_build(A, cs[k], part); fq_max_combine(tot, part); }
    for (int k = 0; k < N; ++k) cs[k].gscale = tot.a;
    for (;;) {
        tot = FqRed{0, 0, 0, 0};
        for (int k = 0; k < N; ++k) { fq_residual(A, cs[k], part); fq_residual_combine(tot, part); }
        double mu = 0.0;
        const double n_rows = tot.b;
        for (int k = 0; k < N; ++k) mu = fq_decide(A, cs[k], tot);
        if (!cs[0].run) break;
        for (int corr = 0; corr < 2; ++corr) {
            double sigma_mu = 0.0;
            if (corr) {
                tot = FqRed{1.0, 1.0, 0, 0};
                for (int k = 0; k < N; ++k) { fq_steplen(A, cs[k], 1.0, part); fq_steplen_combine(tot, part); }
                const double a_aff = fmin(tot.a, tot.b);
                tot = FqRed{0, 0, 0, 0};
                for (int k = 0; k < N; ++k) { fq_mu_aff(A, cs[k], a_aff, part); fq_sum_combine(tot, part); }
                const double mu_aff = tot.a / n_rows, sg = mu_aff / mu;
                sigma_mu = sg * sg * sg * mu;
            }
            for (int k = 0; k < N; ++k) fq_newton_prep(A, cs[k], corr != 0, sigma_mu);
            for (int k = N - 1; k >= 0; --k) fq_newton_back(A, cs[k], corr != 0);
            for (int k = 0; k < N; ++k) fq_newton_fwd(A, cs[k]);
            for (int k = 0; k < N; ++k) fq_newton_rows(A, cs[k], corr != 0, sigma_mu);
        }
        tot = FqRed{1.0, 1.0, 0, 0};
        for (int k = 0; k < N; ++k) { fq_steplen(A, cs[k], 0.995, part); fq_steplen_combine(tot, part); }
        for (int k = 0; k < N; ++k) fq_update(A, cs[k], tot.a, tot.b);
    }
    for (int k = 0; k < N; ++k) fq_output(A, cs[k]);
    delete[] cs;
}
#endif

}  // namespace mpc
