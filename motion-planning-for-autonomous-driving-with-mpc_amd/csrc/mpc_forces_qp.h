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
// One instance per thread; the workspace is [row][Bp] so that a wavefront reads 64 consecutive doubles per row.
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
};

// per-stage workspace rows
enum FqpRow {
    FQ_C = 0,                      // 35  RK4 Jacobian C = [B A] (5x7, row-major)
    FQ_E = FQ_C + 35,              // 5   c(zbar_k) - zbar_{k+1}[2:7]
    FQ_G = FQ_E + 5,               // 7   grad f
    FQ_JH = FQ_G + 7,              // 70  Jacobian of h (10x7)
    FQ_D = FQ_JH + 70,             // 34  right-hand sides d of G dz <= d: [7 lower bounds | 7 upper bounds | 10 h upper | 10 h lower]
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
    FQ_ROWS = FQ_DPI + 5
};
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
MPC_HD void forces_ode_eval(const double* x, const double* u, double l, double* f, double* F4, double& F42, double& F43) {
    const double sn = sin(x[4]), cs = cos(x[4]), td = tan(x[2]);
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
        for (int i = 0; i < 5; ++i) {
            xs[i] = x[i];
            acc[i] = 0.0;
            for (int j = 0; j < 7; ++j) { dxs[i * 7 + j] = (j == i + 2) ? 1.0 : 0.0; dacc[i * 7 + j] = 0.0; dk[i * 7 + j] = 0.0; }
        }
        const double aw[4] = {0.0, 0.5, 0.5, 1.0}, bw[4] = {1.0, 2.0, 2.0, 1.0};
        for (int s = 0; s < 4; ++s) {
            if (s > 0) {
                for (int i = 0; i < 5; ++i) {
                    xs[i] = x[i] + aw[s] * A.dt * kk[i];
                    for (int j = 0; j < 7; ++j) dxs[i * 7 + j] = ((j == i + 2) ? 1.0 : 0.0) + aw[s] * A.dt * dk[i * 7 + j];
                }
            }
            double F4[4], F42, F43;
            forces_ode_eval(xs, u, A.l, kk, F4, F42, F43);
            for (int j = 0; j < 7; ++j) {
                dk[0 * 7 + j] = F4[0] * dxs[3 * 7 + j] + F4[1] * dxs[4 * 7 + j];
                dk[1 * 7 + j] = F4[2] * dxs[3 * 7 + j] + F4[3] * dxs[4 * 7 + j];
                dk[2 * 7 + j] = (j == 0) ? 1.0 : 0.0;
                dk[3 * 7 + j] = (j == 1) ? 1.0 : 0.0;
                dk[4 * 7 + j] = F42 * dxs[2 * 7 + j] + F43 * dxs[3 * 7 + j];
            }
            for (int i = 0; i < 5; ++i) {
                acc[i] += bw[s] * kk[i];
                for (int j = 0; j < 7; ++j) dacc[i * 7 + j] += bw[s] * dk[i * 7 + j];
            }
        }
        for (int i = 0; i < 5; ++i) {
            c[i] = x[i] + A.dt / 6.0 * acc[i];
            for (int j = 0; j < 7; ++j) jc[i * 7 + j] = ((j == i + 2) ? 1.0 : 0.0) + A.dt / 6.0 * dacc[i * 7 + j];
        }
    }
    if (h != nullptr) {
        for (int i = 0; i < 70; ++i) jh[i] = 0.0;
        const double td = tan(z[4]);
        const double q = z[5] * z[5] * td / A.wb;                 // v * psi_dot
        h[0] = z[1] * z[1] + q * q;
        jh[1] = 2.0 * z[1];
        jh[4] = 2.0 * q * z[5] * z[5] * (1.0 + td * td) / A.wb;
        jh[5] = 2.0 * q * 2.0 * z[5] * td / A.wb;
        const double sn = sin(z[6]), cs = cos(z[6]);
        for (int e = 0; e < 3; ++e) {
            const double sg = (e == 0) ? 0.0 : (e == 1 ? 1.0 : -1.0);
            const double ex = z[2] + sg * A.rho * cs, ey = z[3] + sg * A.rho * sn;
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

#define FQW(k_, row_) A.ws[((size_t)(k_) * FQ_ROWS + (size_t)(row_)) * (size_t)A.Bp + (size_t)b]

// (G_k v)_q for a vector v (7) held in registers
MPC_HD double fq_gdot(const ForcesQpArgs& A, int b, int k, int q, const double* v) {
    if (q < 7) return -v[q];
    if (q < 14) return v[q - 7];
    const int j = (q < 24) ? q - 14 : q - 24;
    double s = 0.0;
    for (int i = 0; i < 7; ++i) s += (double)FQW(k, FQ_JH + j * 7 + i) * v[i];
    return (q < 24) ? s : -s;
}
// out += coef * G_k' e_q
MPC_HD void fq_gt_axpy(const ForcesQpArgs& A, int b, int k, int q, double coef, double* out) {
    if (q < 7) { out[q] -= coef; return; }
    if (q < 14) { out[q - 7] += coef; return; }
    const int j = (q < 24) ? q - 14 : q - 24;
    const double sg = (q < 24) ? coef : -coef;
    for (int i = 0; i < 7; ++i) out[i] += sg * (double)FQW(k, FQ_JH + j * 7 + i);
}

// Newton step for complementarity target rc (per row: rc_q = s lam [+ ds_aff dl_aff - sigma mu]); `factor`: also (re)build
// the matrix part of the Riccati recursion.  corr: 0 = predictor (rc = s lam), 1 = corrector (uses FQ_DS / FQ_DL of the predictor)
MPC_HD void fq_newton(const ForcesQpArgs& A, int b, bool corr, double sigma_mu) {
    const int N = A.N;
    double P[15], pv[5];
    for (int i = 0; i < 15; ++i) P[i] = 0.0;
    for (int i = 0; i < 5; ++i) pv[i] = 0.0;
    // ---------------- backward
    for (int k = N - 1; k >= 0; --k) {
        double w[7], rho[7], Phi[28];                       // Phi: symmetric 7x7, upper, index i*7 - i(i-1)/2 + (j-i)
        for (int i = 0; i < 7; ++i) w[i] = FQW(k, FQ_W + i);
        const double* hd = (k == N - 1) ? A.hdN : A.hd;
        for (int i = 0; i < 7; ++i) rho[i] = hd[i] * w[i] + (double)FQW(k, FQ_G + i);
        if (!corr) {
            for (int i = 0; i < 28; ++i) Phi[i] = 0.0;
            for (int i = 0; i < 7; ++i) Phi[i * 7 - i * (i - 1) / 2] = hd[i];
        }
        for (int q = 0; q < FQ_MI; ++q) {
            if (!fq_row_on(A, k, q)) continue;
            const double s = FQW(k, FQ_S + q), lam = FQW(k, FQ_LAM + q), d = FQW(k, FQ_D + q);
            const double rp = fq_gdot(A, b, k, q, w) + s - d;
            double rc = s * lam;
            if (corr) rc += (double)FQW(k, FQ_DS + q) * (double)FQW(k, FQ_DL + q) - sigma_mu;
            const double D = lam / s;
            fq_gt_axpy(A, b, k, q, lam + D * rp - rc / s, rho);
            if (!corr) {
                double gq[7] = {0, 0, 0, 0, 0, 0, 0};
                fq_gt_axpy(A, b, k, q, 1.0, gq);
                for (int i = 0; i < 7; ++i) {
                    if (gq[i] == 0.0) continue;
                    for (int j = i; j < 7; ++j) Phi[i * 7 - i * (i - 1) / 2 + (j - i)] += D * gq[i] * gq[j];
                }
            }
        }
        // Q-function blocks.  Stage N-1 has no successor; otherwise eps_k = -r_e(k+1)
        double Huu[3], Hux[10], Hxx[15], hu[2], hx[5];
        if (!corr) {
            Huu[0] = Phi[0]; Huu[1] = Phi[1]; Huu[2] = Phi[7];
            for (int j = 0; j < 5; ++j) { Hux[j] = Phi[2 + j]; Hux[5 + j] = Phi[7 + 1 + j]; }
            for (int i = 0; i < 5; ++i)
                for (int j = i; j < 5; ++j) Hxx[fq_sidx(i, j)] = Phi[(i + 2) * 7 - (i + 2) * (i + 1) / 2 + (j - i)];
        }
        hu[0] = rho[0]; hu[1] = rho[1];
        for (int i = 0; i < 5; ++i) hx[i] = rho[2 + i];
        double Cm[35], eps[5] = {0, 0, 0, 0, 0};
        if (k < N - 1) {
            double q5[5];
            for (int i = 0; i < 35; ++i) Cm[i] = FQW(k, FQ_C + i);
            // eps = -(w_{k+1}[2:7] - C w_k - e_k)
            for (int i = 0; i < 5; ++i) {
                double t = (double)FQW(k, FQ_E + i) - (double)FQW(k + 1, FQ_W + 2 + i);
                for (int j = 0; j < 7; ++j) t += Cm[i * 7 + j] * w[j];
                eps[i] = t;
            }
            if (!corr) {
                double Pe[5];
                for (int i = 0; i < 5; ++i) {
                    double t = 0.0;
                    for (int j = 0; j < 5; ++j) t += P[fq_sidx(i, j)] * eps[j];
                    Pe[i] = t;
                    FQW(k, FQ_PE + i) = t;
                }
                // M = P C (5x7), then C' M
                double M[35];
                for (int i = 0; i < 5; ++i)
                    for (int j = 0; j < 7; ++j) {
                        double t = 0.0;
                        for (int r = 0; r < 5; ++r) t += P[fq_sidx(i, r)] * Cm[r * 7 + j];
                        M[i * 7 + j] = t;
                    }
                for (int i = 0; i < 7; ++i)
                    for (int j = i; j < 7; ++j) {
                        double t = 0.0;
                        for (int r = 0; r < 5; ++r) t += Cm[r * 7 + i] * M[r * 7 + j];
                        if (i < 2 && j < 2) Huu[i + j] += t;
                        else if (i < 2) Hux[i * 5 + (j - 2)] += t;
                        else Hxx[fq_sidx(i - 2, j - 2)] += t;
                    }
                for (int i = 0; i < 5; ++i) q5[i] = Pe[i] + pv[i];
            } else {
                for (int i = 0; i < 5; ++i) q5[i] = (double)FQW(k, FQ_PE + i) + pv[i];
            }
            for (int j = 0; j < 7; ++j) {
                double t = 0.0;
                for (int r = 0; r < 5; ++r) t += Cm[r * 7 + j] * q5[r];
                if (j < 2) hu[j] += t; else hx[j - 2] += t;
            }
        }
        double hi[3], Kk[10], Puu[3], Pxu[10];
        if (!corr) {
            Puu[0] = Phi[0]; Puu[1] = Phi[1]; Puu[2] = Phi[7];
            for (int j = 0; j < 5; ++j) { Pxu[j] = Phi[2 + j]; Pxu[5 + j] = Phi[8 + j]; }
            const double det = Huu[0] * Huu[2] - Huu[1] * Huu[1];
            hi[0] = Huu[2] / det; hi[1] = -Huu[1] / det; hi[2] = Huu[0] / det;
            for (int j = 0; j < 5; ++j) {
                Kk[j] = -(hi[0] * Hux[j] + hi[1] * Hux[5 + j]);
                Kk[5 + j] = -(hi[1] * Hux[j] + hi[2] * Hux[5 + j]);
            }
            for (int i = 0; i < 3; ++i) { FQW(k, FQ_HUI + i) = hi[i]; FQW(k, FQ_PHU + i) = Puu[i]; }
            for (int i = 0; i < 10; ++i) { FQW(k, FQ_K + i) = Kk[i]; FQW(k, FQ_HXU + i) = Pxu[i]; }
        } else {
            for (int i = 0; i < 3; ++i) { hi[i] = FQW(k, FQ_HUI + i); Puu[i] = FQW(k, FQ_PHU + i); }
            for (int i = 0; i < 10; ++i) { Kk[i] = FQW(k, FQ_K + i); Pxu[i] = FQW(k, FQ_HXU + i); }
        }
        const double kf0 = -(hi[0] * hu[0] + hi[1] * hu[1]), kf1 = -(hi[1] * hu[0] + hi[2] * hu[1]);
        // Cost-to-go in the symmetric ("Joseph") form: with u = K x + kff and A_cl = A + B K,
        //    P_k = [K; I]' Phi [K; I] + A_cl' P+ A_cl,   p_k = rho_x + K' rho_u + (Phi_xu + K' Phi_uu) kff + A_cl' (P+ (B kff + eps) + p+)
        // -- sums of terms of one sign.  The textbook form Hxx - Hxu Huu^-1 Hux subtracts two numbers of size lam/s (1e10 and
        // more for active rows late in the iteration) and loses the O(1) entries that decide the step.
        double Acl[25], t5[5] = {0, 0, 0, 0, 0}, Pp[15];
        if (k < N - 1) {
            for (int i = 0; i < 15; ++i) Pp[i] = corr ? (double)FQW(k + 1, FQ_P + i) : P[i];
            double off[5];
            for (int i = 0; i < 5; ++i) {
                off[i] = eps[i] + Cm[i * 7 + 0] * kf0 + Cm[i * 7 + 1] * kf1;
                for (int j = 0; j < 5; ++j) Acl[i * 5 + j] = Cm[i * 7 + 2 + j] + Cm[i * 7 + 0] * Kk[j] + Cm[i * 7 + 1] * Kk[5 + j];
            }
            for (int i = 0; i < 5; ++i) {
                double t = pv[i];
                for (int j = 0; j < 5; ++j) t += Pp[fq_sidx(i, j)] * off[j];
                t5[i] = t;
            }
        }
        double pn[5];
        for (int i = 0; i < 5; ++i) {
            double t = rho[2 + i] + Kk[i] * rho[0] + Kk[5 + i] * rho[1];
            t += (Pxu[i] + Kk[i] * Puu[0] + Kk[5 + i] * Puu[1]) * kf0 + (Pxu[5 + i] + Kk[i] * Puu[1] + Kk[5 + i] * Puu[2]) * kf1;
            if (k < N - 1)
                for (int r = 0; r < 5; ++r) t += Acl[r * 5 + i] * t5[r];
            pn[i] = t;
        }
        if (!corr) {
            double PA[25];                                  // P+ A_cl
            if (k < N - 1)
                for (int i = 0; i < 5; ++i)
                    for (int j = 0; j < 5; ++j) {
                        double t = 0.0;
                        for (int r = 0; r < 5; ++r) t += Pp[fq_sidx(i, r)] * Acl[r * 5 + j];
                        PA[i * 5 + j] = t;
                    }
            for (int i = 0; i < 5; ++i)
                for (int j = i; j < 5; ++j) {
                    double t = Phi[(i + 2) * 7 - (i + 2) * (i + 1) / 2 + (j - i)];
                    t += Pxu[i] * Kk[j] + Pxu[5 + i] * Kk[5 + j] + Kk[i] * Pxu[j] + Kk[5 + i] * Pxu[5 + j];
                    t += Kk[i] * (Puu[0] * Kk[j] + Puu[1] * Kk[5 + j]) + Kk[5 + i] * (Puu[1] * Kk[j] + Puu[2] * Kk[5 + j]);
                    if (k < N - 1)
                        for (int r = 0; r < 5; ++r) t += Acl[r * 5 + i] * PA[r * 5 + j];
                    P[fq_sidx(i, j)] = t;
                }
            for (int i = 0; i < 15; ++i) FQW(k, FQ_P + i) = P[i];
        }
        for (int i = 0; i < 5; ++i) pv[i] = pn[i];
        // the step of u_k is finished in the forward sweep; keep kff and p_k (in the DW / DPI rows for now)
        FQW(k, FQ_DW + 0) = kf0;
        FQW(k, FQ_DW + 1) = kf1;
        for (int i = 0; i < 5; ++i) FQW(k, FQ_DPI + i) = pv[i];
    }
    // ---------------- forward
    double dx[5];
    for (int i = 0; i < 5; ++i) dx[i] = -((double)FQW(0, FQ_W + 2 + i) - ((double)A.xinit[(size_t)b * 5 + i] - (double)A.zbar[((size_t)b * N) * 7 + 2 + i]));
    for (int k = 0; k < N; ++k) {
        double du[2] = {FQW(k, FQ_DW + 0), FQW(k, FQ_DW + 1)};
        for (int j = 0; j < 5; ++j) { du[0] += (double)FQW(k, FQ_K + j) * dx[j]; du[1] += (double)FQW(k, FQ_K + 5 + j) * dx[j]; }
        double dw[7] = {du[0], du[1], dx[0], dx[1], dx[2], dx[3], dx[4]};
        for (int i = 0; i < 7; ++i) FQW(k, FQ_DW + i) = dw[i];
        // new equality multipliers of block k: -(P_k dx_k + p_k)
        for (int i = 0; i < 5; ++i) {
            double t = FQW(k, FQ_DPI + i);
            for (int j = 0; j < 5; ++j) t += (double)FQW(k, FQ_P + fq_sidx(i, j)) * dx[j];
            FQW(k, FQ_DPI + i) = -t;
        }
        double w[7];
        for (int i = 0; i < 7; ++i) w[i] = FQW(k, FQ_W + i);
        for (int q = 0; q < FQ_MI; ++q) {
            if (!fq_row_on(A, k, q)) continue;
            const double s = FQW(k, FQ_S + q), lam = FQW(k, FQ_LAM + q), d = FQW(k, FQ_D + q);
            const double rp = fq_gdot(A, b, k, q, w) + s - d;
            double rc = s * lam;
            if (corr) rc += (double)FQW(k, FQ_DS + q) * (double)FQW(k, FQ_DL + q) - sigma_mu;
            const double ds = -rp - fq_gdot(A, b, k, q, dw);
            const double dl = -(rc + lam * ds) / s;
            FQW(k, FQ_DS + q) = ds;
            FQW(k, FQ_DL + q) = dl;
        }
        if (k < N - 1) {
            double dn[5];
            for (int i = 0; i < 5; ++i) {
                // dx_{k+1} = C dw + eps,  eps = e_k - w_{k+1}[2:7] + C w_k
                double t = (double)FQW(k, FQ_E + i) - (double)FQW(k + 1, FQ_W + 2 + i);
                for (int j = 0; j < 7; ++j) t += (double)FQW(k, FQ_C + i * 7 + j) * (w[j] + dw[j]);
                dn[i] = t;
            }
            for (int i = 0; i < 5; ++i) dx[i] = dn[i];
        }
    }
}

// one SQP step of instance b
MPC_HD void forces_qp_instance(const ForcesQpArgs& A, int b) {
    const int N = A.N;
    // ---------------- build the QP at zbar
    double gscale = 1.0;
    for (int k = 0; k < N; ++k) {
        double z[7], p[10], gf[7], c[5], jc[35], h[10], jh[70], fv;
        for (int i = 0; i < 7; ++i) z[i] = A.zbar[((size_t)b * N + k) * 7 + i];
        for (int i = 0; i < 10; ++i) p[i] = A.params[((size_t)b * N + k) * 10 + i];
        const bool term = (k == N - 1);
        forces_stage_functions(A, z, p, term, fv, gf, term ? nullptr : c, jc, h, jh);
        for (int i = 0; i < 7; ++i) { FQW(k, FQ_G + i) = gf[i]; FQW(k, FQ_W + i) = 0.0; gscale = fmax(gscale, fabs(gf[i])); }
        for (int i = 0; i < 70; ++i) FQW(k, FQ_JH + i) = jh[i];
        if (!term) {
            for (int i = 0; i < 35; ++i) FQW(k, FQ_C + i) = jc[i];
            for (int i = 0; i < 5; ++i) FQW(k, FQ_E + i) = c[i] - (double)A.zbar[((size_t)b * N + k + 1) * 7 + 2 + i];
        }
        for (int i = 0; i < 5; ++i) FQW(k, FQ_PI + i) = 0.0;
        for (int q = 0; q < FQ_MI; ++q) {
            double d = 0.0;
            if (q < 7) d = z[q] - A.lb[q];
            else if (q < 14) d = A.ub[q - 7] - z[q - 7];
            else if (q < 24) d = A.hu[q - 14] - h[q - 14];
            else d = h[q - 24] - A.hl[q - 24];
            const bool on = fq_row_on(A, k, q);
            FQW(k, FQ_D + q) = on ? d : 0.0;
            FQW(k, FQ_S + q) = on ? fmax(d, 1.0) : 1.0;
            FQW(k, FQ_LAM + q) = on ? 1.0 / fmax(d, 1.0) : 0.0;          // centred start: s * lam = 1 on every row
        }
    }
    // ---------------- Mehrotra predictor-corrector
    int it = 0, status = 0;
    double kkt = INFINITY;
    for (;; ++it) {
        // residual norms
        double mu = 0.0, rmax = 0.0, rdmax = 0.0;      // rmax: primal + equality residuals, rdmax: dual residual
        int M = 0;
        for (int k = 0; k < N; ++k) {
            double w[7], rd[7];
            const double* hd = (k == N - 1) ? A.hdN : A.hd;
            for (int i = 0; i < 7; ++i) { w[i] = FQW(k, FQ_W + i); rd[i] = hd[i] * w[i] + (double)FQW(k, FQ_G + i); }
            for (int i = 0; i < 5; ++i) rd[2 + i] += (double)FQW(k, FQ_PI + i);
            if (k < N - 1)
                for (int j = 0; j < 7; ++j) {
                    double t = 0.0;
                    for (int r = 0; r < 5; ++r) t += (double)FQW(k, FQ_C + r * 7 + j) * (double)FQW(k + 1, FQ_PI + r);
                    rd[j] -= t;
                }
            for (int q = 0; q < FQ_MI; ++q) {
                if (!fq_row_on(A, k, q)) continue;
                const double s = FQW(k, FQ_S + q), lam = FQW(k, FQ_LAM + q);
                fq_gt_axpy(A, b, k, q, lam, rd);
                rmax = fmax(rmax, fabs(fq_gdot(A, b, k, q, w) + s - (double)FQW(k, FQ_D + q)));
                mu += s * lam;
                ++M;
            }
            for (int i = 0; i < 7; ++i) rdmax = fmax(rdmax, fabs(rd[i]));
            for (int i = 0; i < 5; ++i) {
                double re;
                if (k == 0) re = w[2 + i] - ((double)A.xinit[(size_t)b * 5 + i] - (double)A.zbar[((size_t)b * N) * 7 + 2 + i]);
                else {
                    re = w[2 + i] - (double)FQW(k - 1, FQ_E + i);
                    for (int j = 0; j < 7; ++j) re -= (double)FQW(k - 1, FQ_C + i * 7 + j) * (double)FQW(k - 1, FQ_W + j);
                }
                rmax = fmax(rmax, fabs(re));
            }
        }
        mu = M > 0 ? mu / M : 0.0;
        kkt = fmax(fmax(rmax, rdmax / gscale), mu);
        if (!(rmax == rmax) || !(rdmax == rdmax) || !(mu == mu) || kkt > 1e300) { kkt = NAN; status = -6; break; }
        // Stage-wise elimination condenses the inequality rows into the stage Hessians with weights lam/s, which grow like
        // 1/mu for active rows; past mu ~ 1e-7 the Riccati recursion loses the small entries next to them.  The targets are
        // therefore the accuracy an SQP step needs (FORCESPRO is called with tolerances of 1e-1, optimizer.py:272-273):
        // primal / equality residuals <= tol (metres, m/s, rad), dual residual <= tol relative to the largest cost gradient.
        if (rmax <= A.tol && rdmax <= A.tol * gscale && mu <= A.tol_mu) { status = 1; break; }
        if (mu > 1e6) { status = -7; break; }                   // multipliers diverge: the linearised constraints are inconsistent
        if (it >= A.max_it) { status = 0; break; }
        // predictor
        fq_newton(A, b, false, 0.0);
        double a_aff = 1.0;
        for (int k = 0; k < N; ++k)
            for (int q = 0; q < FQ_MI; ++q) {
                if (!fq_row_on(A, k, q)) continue;
                const double s = FQW(k, FQ_S + q), lam = FQW(k, FQ_LAM + q), ds = FQW(k, FQ_DS + q), dl = FQW(k, FQ_DL + q);
                if (ds < 0) a_aff = fmin(a_aff, -s / ds);
                if (dl < 0) a_aff = fmin(a_aff, -lam / dl);
            }
        double mu_aff = 0.0;
        for (int k = 0; k < N; ++k)
            for (int q = 0; q < FQ_MI; ++q) {
                if (!fq_row_on(A, k, q)) continue;
                mu_aff += ((double)FQW(k, FQ_S + q) + a_aff * (double)FQW(k, FQ_DS + q)) * ((double)FQW(k, FQ_LAM + q) + a_aff * (double)FQW(k, FQ_DL + q));
            }
        mu_aff /= M;
        const double sg = mu_aff / mu, sigma = sg * sg * sg;
        // corrector
        fq_newton(A, b, true, sigma * mu);
        double a_p = 1.0, a_d = 1.0;
        for (int k = 0; k < N; ++k)
            for (int q = 0; q < FQ_MI; ++q) {
                if (!fq_row_on(A, k, q)) continue;
                const double s = FQW(k, FQ_S + q), lam = FQW(k, FQ_LAM + q), ds = FQW(k, FQ_DS + q), dl = FQW(k, FQ_DL + q);
                if (ds < 0) a_p = fmin(a_p, -0.995 * s / ds);
                if (dl < 0) a_d = fmin(a_d, -0.995 * lam / dl);
            }
        for (int k = 0; k < N; ++k) {
            for (int i = 0; i < 7; ++i) FQW(k, FQ_W + i) = (double)FQW(k, FQ_W + i) + a_p * (double)FQW(k, FQ_DW + i);
            for (int i = 0; i < 5; ++i) FQW(k, FQ_PI + i) = (double)FQW(k, FQ_PI + i) + a_d * ((double)FQW(k, FQ_DPI + i) - (double)FQW(k, FQ_PI + i));
            for (int q = 0; q < FQ_MI; ++q) {
                if (!fq_row_on(A, k, q)) continue;
                FQW(k, FQ_S + q) = (double)FQW(k, FQ_S + q) + a_p * (double)FQW(k, FQ_DS + q);
                FQW(k, FQ_LAM + q) = (double)FQW(k, FQ_LAM + q) + a_d * (double)FQW(k, FQ_DL + q);
            }
        }
    }
    for (int k = 0; k < N; ++k)
        for (int i = 0; i < 7; ++i) A.z_out[((size_t)b * N + k) * 7 + i] = (double)A.zbar[((size_t)b * N + k) * 7 + i] + (double)FQW(k, FQ_W + i);
    if (A.iters) A.iters[b] = it;
    if (A.status) A.status[b] = status;
    if (A.kkt) A.kkt[b] = kkt;
}

#undef FQW

}  // namespace mpc
