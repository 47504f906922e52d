/*
 * ORACLE (test infrastructure, NOT product code).
 *
 * Plain-C, single-thread-per-instance CPU restatement of the hot path of the reference:
 *   the NLP of MPC_Planner/optimizer.py:373-558 (CasadiOptimizer) solved the way
 *   `sol(x0=..., p=..., lbg=..., lbx=..., ubg=..., ubx=...)` (optimizer.py:607) solves it --
 *   IPOPT's primal-dual interior-point filter line-search method (Waechter & Biegler 2006), restated with a
 *   stage-structured (Riccati) solve of the condensed KKT system in place of MUMPS.
 *
 * PARITY UNPINNED: CasADi (>=3.5.1, requirments:5) / IPOPT are third-party, absent from /root/reference and not
 * installable here; the optimum is pinned instead against scipy (SLSQP + trust-constr) golden vectors, the
 * plant step against the reference's recorded trajectories, and the ODE against the reference's
 * CasADi-generated FORCESNLPsolver_model.c (oracle/_ref).
 *
 * Only tests/, bench.py's cpu_baseline leg and __graft_entry__.smoke() may load this library.
 */
#ifndef MPC_ORACLE_H
#define MPC_ORACLE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MPCO_NX_MAX 6
#define MPCO_NU 2
#define MPCO_N_MAX 128

enum { MPCO_CONVERGED = 1, MPCO_MAXITER = 0, MPCO_NAN = -6, MPCO_NOPROGRESS = -7 };

typedef struct mpco_desc {
    int32_t N;               /* horizon (optimizer.py:520)                                             */
    int32_t nx;              /* 5 (reference) or 6 (5 + decoupled progress state)                      */
    int32_t obst_mult;       /* 3: every distinct circle distance is appended three times (:395-403)   */
    int32_t max_iter;        /* 100 (optimizer.py:556)                                                  */
    int32_t fixed_iters;     /* >0: run exactly this many iterations (benchmark mode)                  */
    int32_t reserved;        /* 1 = keep the stage-0 friction row as a row (no presolve into a bound on a_0); 2 = as IPOPT sees it: a row
                              * whose slack also carries the lower bound lbg[0] = 0 with its log barrier                          */
    double dt;               /* scenario.dt = 0.1                                                       */
    double wheelbase;        /* 2.5789128  (configuration.py:362-363)                                   */
    double friction_div;     /* 2.578      (optimizer.py:378)                                           */
    double ego_offset;       /* 0.75       (configuration.py:80-91 with L=4.508)                        */
    double Q[MPCO_NX_MAX];   /* diag weights on (x,y,delta,v,psi[,s])                                   */
    double R[MPCO_NU];       /* diag weights on (deltaDot, a)                                           */
    double obst[6];          /* obstacle circle centres (x0,y0,x1,y1,x2,y2), optimizer.py:60-64        */
    double fric_lo, fric_hi; /* lbg[0], ubg[0]                                                          */
    double obst_lo, obst_hi; /* lbg/ubg of the obstacle rows (r_ego + r_obs, +inf)                      */
    double tol;              /* 1e-8 (IPOPT default)                                                    */
} mpco_desc;

/* lbx/ubx: per-variable bounds in the reference's decision-vector order [u_0..u_{N-1} | x_0..x_N]
 * (optimizer.py:470-491, :550); +-inf means absent.  x0 / p / x_out: same order, length n_w.          */
int mpco_solve(const mpco_desc* d, const double* lbx, const double* ubx,
               const double* x0, const double* p,
               double* x_out, int32_t* status, int32_t* iters, double* kkt, double* obj);

/* B instances, row-major [B, n_w]; nthreads <= 1 runs serially, otherwise OpenMP over instances.      */
int mpco_solve_batch(const mpco_desc* d, const double* lbx, const double* ubx, int32_t B,
                     const double* x0, const double* p,
                     double* x_out, int32_t* status, int32_t* iters, double* kkt, int32_t nthreads);

/* NLP pieces exposed for cross-checks against oracle/nlp_numpy.py and the HIP stage kernels */
void mpco_ode(const mpco_desc* d, const double* x, const double* u, double* f);
void mpco_plant_step_euler(const mpco_desc* d, const double* x, const double* u, double* xn);
void mpco_plant_step_rk4(const mpco_desc* d, const double* x, const double* u, double* xn);
double mpco_objective(const mpco_desc* d, const double* w, const double* p);
void mpco_constraints(const mpco_desc* d, const double* w, const double* p, double* g /* n_g */);

/* same as mpco_solve, additionally records per-iteration scalars:
 * trace[it*8 + {0:mu, 1:theta, 2:phi, 3:alpha, 4:alpha_z, 5:delta_w, 6:E0, 7:n_trials}], trace_cap rows.  */
int mpco_solve_trace(const mpco_desc* d, const double* lbx, const double* ubx,
                     const double* x0, const double* p,
                     double* x_out, int32_t* status, int32_t* iters, double* kkt, double* obj,
                     double* trace, int32_t trace_cap);

int mpco_version(void);

#ifdef __cplusplus
}
#endif
#endif
