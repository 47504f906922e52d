/*
 * ORACLE (test infrastructure, NOT product code) -- see mpc_oracle.h for the scope statement.
 *
 * What is restated, and from where:
 *   decision vector  w = [u_0..u_{N-1} | x_0..x_N]              optimizer.py:550  (column-major reshape)
 *   parameter vector p = [U_ref (unused) | X_ref]                optimizer.py:552
 *   cost             sum_{i<N} (x_i - r_{i+1})' Q (x_i - r_{i+1}) + u_i' R u_i     optimizer.py:507-511
 *                    (the terminal P term on line 510 is a dead expression)
 *   g rows           [ friction | x_0 - r_0 | Euler defects | 9 obstacle rows per stage ]  optimizer.py:378-403
 *   bounds           optimizer.py:421-491
 *   ODE              configuration.py:353-368
 *   circle geometry  configuration.py:69-93
 *   solver           ca.nlpsol('ipopt') with max_iter=100, tol=1e-8   optimizer.py:556-558, called at :607
 *
 * Interior-point method (IPOPT's published algorithm, Waechter & Biegler 2006; constants are IPOPT defaults):
 *   - slack per inequality row, log barrier on finite bounds of variables and slacks, bounds relaxed by 1e-8
 *   - the three identical copies of each obstacle row keep identical slacks/multipliers for ever (the Newton
 *     system is symmetric in them), so they are carried once with multiplicity `obst_mult`
 *   - a lower bound <= 0 on the friction row |y| is implied by the absolute value and is NOT given a barrier
 *     (with a barrier the row has a spurious stationary point at a_0 -> 0 with a diverging multiplier;
 *     the set of KKT points of the NLP is unchanged)  -- deviation from a literal IPOPT run, see DESIGN.md
 *   - start point pushed inside (kappa_1 = kappa_2 = 1e-2), z = 1, equality multipliers 0 (no LS estimate)
 *   - gradient-based objective scaling (max gradient 100)
 *   - monotone mu, kappa_eps = 10, kappa_mu = 0.2, theta_mu = 1.5, tau_min = 0.99
 *   - condensed KKT system solved by a Riccati recursion over the stages; inertia correction = retry with
 *     delta_w * I until every 2x2 input block of the recursion is positive definite
 *   - filter line search (switching + Armijo conditions), no second-order correction, no restoration phase
 *     (line-search failure => status NOPROGRESS)
 *   - termination on the scaled KKT error E_0 <= tol
 */
#include "mpc_oracle.h"

#include <float.h>
#include <math.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define NX MPCO_NX_MAX
#define NU MPCO_NU
#define NS (MPCO_N_MAX + 1)
#define FILTER_MAX 32

/* IPOPT default constants */
static const double MU_INIT = 0.1, KAPPA_EPS = 10.0, KAPPA_MU = 0.2, THETA_MU = 1.5, TAU_MIN = 0.99;
static const double KAPPA_1 = 1e-2, KAPPA_2 = 1e-2, BOUND_RELAX = 1e-8, S_MAX = 100.0, KAPPA_SIGMA = 1e10;
static const double GAMMA_THETA = 1e-5, GAMMA_PHI = 1e-8, LS_DELTA = 1.0, S_THETA = 1.1, S_PHI = 2.3;
static const double ETA_PHI = 1e-8, GAMMA_ALPHA = 0.05;
static const double DW_MIN = 1e-20, DW_0 = 1e-4, DW_MAX = 1e40, KW_MINUS = 1.0 / 3.0, KW_PLUS = 8.0, KW_PLUS_BAR = 100.0;
static const double SCALING_MAX_GRAD = 100.0;
static const double ROLLOUT_FACTOR = 10.0;
/* constraint violations (1-norm) below this are round-off: two orders under the 1e-8 feasibility tolerance.  The
 * theta-comparisons of the filter line search clamp at this floor, otherwise noise in theta ~ 1e-12 rejects
 * the last Newton steps that only reduce the dual infeasibility. */
static const double THETA_FLOOR = 1e-10;

typedef struct {
    double u[NS][NU], x[NS][NX];
    double zlu[NS][NU], zuu[NS][NU], zlx[NS][NX], zux[NS][NX];
    double so[NS][3], nuo[NS][3], zlo[NS][3], zuo[NS][3];
    double lam[NS][NX];
    double sf, nuf, zlf, zuf;
} iterate_t;

typedef struct {
    double fcost;              /* unscaled objective */
    double c[NS][NX];          /* c[0] = x_0 - r_0, c[k+1] = x_{k+1} - x_k - dt f(x_k,u_k) */
    double dobs[NS][3];        /* circle distances */
    double dfric;              /* |y| */
    /* derivatives (only when requested) */
    double fx[NS][NX][NX];     /* d f / d x (dense for clarity) */
    double jo[NS][3][3];       /* d dist_j / d (sx, sy, psi) */
    double ho[NS][3][3][3];    /* Hessians of dist_j wrt (sx, sy, psi) */
    double gfr[3];             /* d |y| / d (a_0, delta_0, v_0) */
    double hfr[3][3];
} eval_t;

typedef struct {
    int N, nx;
    double lbu[NS][NU], ubu[NS][NU], lbx[NS][NX], ubx[NS][NX];
    int has_fl, has_fu, has_ol, has_ou;
    int fric_row;              /* 0: the stage-0 friction row was presolved into bounds on a_0 */
    double fl, fu, ol, ou;
    const double* r;           /* X_ref, [N+1][nx] */
    const mpco_desc* d;
} prob_t;

typedef struct {
    iterate_t it, step, trial;
    eval_t ev, evt;
    double Hxx[NS][NX][NX], Huu[NS][NU][NU], Hux0[NU][NX], gx[NS][NX], gu[NS][NU];
    double P[NS][NX][NX], pv[NS][NX], K[NS][NU][NX], kff[NS][NU];
    double gradfx[NS][NX], gradfu[NS][NU];    /* scaled objective gradient */
    double gbx[NS][NX], gbu[NS][NU], gbo[NS][3], gbf;   /* barrier gradients */
    double sgo[NS][3], sgf;                              /* per-copy sigma of the slacks */
} work_t;

static inline int fin(double v) { return isfinite(v); }
/* IPOPT's Compare_le: lhs <= rhs up to 10 machine epsilons of a reference magnitude (round-off safeguard
 * of the acceptance tests, IpUtils.cpp) */
static inline int cmp_le(double lhs, double rhs, double base) { return lhs - rhs <= 10.0 * DBL_EPSILON * fabs(base); }

/* ------------------------------------------------------------------------------------------------ model */
void mpco_ode(const mpco_desc* d, const double* x, const double* u, double* f) {
    f[0] = x[3] * cos(x[4]);
    f[1] = x[3] * sin(x[4]);
    f[2] = u[0];
    f[3] = u[1];
    f[4] = x[3] / d->wheelbase * tan(x[2]);
    if (d->nx == 6) f[5] = x[3];
}

void mpco_plant_step_euler(const mpco_desc* d, const double* x, const double* u, double* xn) {
    double f[NX];
    mpco_ode(d, x, u, f);
    for (int i = 0; i < d->nx; ++i) xn[i] = x[i] + d->dt * f[i];   /* optimizer.py:649-650 */
}

void mpco_plant_step_rk4(const mpco_desc* d, const double* x, const double* u, double* xn) {
    /* classic RK4, one step of length dt (forcespro.nlp.integrators.RK4, optimizer.py:97-98) */
    double k1[NX], k2[NX], k3[NX], k4[NX], t[NX];
    const int nx = d->nx;
    const double h = d->dt;
    mpco_ode(d, x, u, k1);
    for (int i = 0; i < nx; ++i) t[i] = x[i] + 0.5 * h * k1[i];
    mpco_ode(d, t, u, k2);
    for (int i = 0; i < nx; ++i) t[i] = x[i] + 0.5 * h * k2[i];
    mpco_ode(d, t, u, k3);
    for (int i = 0; i < nx; ++i) t[i] = x[i] + h * k3[i];
    mpco_ode(d, t, u, k4);
    for (int i = 0; i < nx; ++i) xn[i] = x[i] + h / 6.0 * (k1[i] + 2 * k2[i] + 2 * k3[i] + k4[i]);
}

static void ode_jac(const mpco_desc* d, const double* x, double fx[NX][NX]) {
    memset(fx, 0, sizeof(double) * NX * NX);
    const double c = cos(x[4]), s = sin(x[4]), cd = cos(x[2]);
    fx[0][3] = c;
    fx[0][4] = -x[3] * s;
    fx[1][3] = s;
    fx[1][4] = x[3] * c;
    fx[4][2] = x[3] / (d->wheelbase * cd * cd);
    fx[4][3] = tan(x[2]) / d->wheelbase;
    if (d->nx == 6) fx[5][3] = 1.0;
}

/* sum_r lam[r] * Hessian_x f_r, added into H with factor `scale` */
static void ode_hess_add(const mpco_desc* d, const double* x, const double* lam, double scale, double H[NX][NX]) {
    const double c = cos(x[4]), s = sin(x[4]), v = x[3];
    const double cd = cos(x[2]), td = tan(x[2]), l = d->wheelbase;
    const double h34 = lam[0] * (-s) + lam[1] * c;
    const double h44 = lam[0] * (-v * c) + lam[1] * (-v * s);
    const double h23 = lam[4] / (l * cd * cd);
    const double h22 = lam[4] * v * 2.0 * td / (l * cd * cd);
    H[3][4] += scale * h34;
    H[4][3] += scale * h34;
    H[4][4] += scale * h44;
    H[2][3] += scale * h23;
    H[3][2] += scale * h23;
    H[2][2] += scale * h22;
}

static void obstacle_eval(const mpco_desc* d, const double* x, double dist[3], double J[3][3], double H[3][3][3], int derivs) {
    static const double sgn[3] = {0.0, 1.0, -1.0};
    const double rho = d->ego_offset, c = cos(x[4]), s = sin(x[4]);
    for (int j = 0; j < 3; ++j) {
        const double cx = x[0] + sgn[j] * rho * c - d->obst[2 * j];
        const double cy = x[1] + sgn[j] * rho * s - d->obst[2 * j + 1];
        const double r = sqrt(cx * cx + cy * cy);
        dist[j] = r;
        if (!derivs) continue;
        const double ex = cx / r, ey = cy / r;
        const double tx = -sgn[j] * rho * s, ty = sgn[j] * rho * c;
        J[j][0] = ex;
        J[j][1] = ey;
        J[j][2] = ex * tx + ey * ty;
        const double m00 = (1 - ex * ex) / r, m01 = -ex * ey / r, m11 = (1 - ey * ey) / r;
        const double mt0 = m00 * tx + m01 * ty, mt1 = m01 * tx + m11 * ty;
        const double nxx = -sgn[j] * rho * c, nyy = -sgn[j] * rho * s;
        H[j][0][0] = m00; H[j][0][1] = m01; H[j][1][0] = m01; H[j][1][1] = m11;
        H[j][0][2] = mt0; H[j][2][0] = mt0; H[j][1][2] = mt1; H[j][2][1] = mt1;
        H[j][2][2] = tx * mt0 + ty * mt1 + ex * nxx + ey * nyy;
    }
}

/* |y|, y = a^2 + v*(tan(delta)*v/kappa); gradient/Hessian wrt (a, delta, v); sign(0) = 0 */
static double friction_eval(const mpco_desc* d, const double* u0, const double* x0, double g[3], double H[3][3], int derivs) {
    const double kap = d->friction_div, a = u0[1], dl = x0[2], v = x0[3];
    const double td = tan(dl), cd = cos(dl), cd2 = cd * cd;
    const double y = a * a + v * (td * v / kap);
    if (derivs) {
        const double sg = (y > 0) - (y < 0);
        g[0] = sg * 2 * a;
        g[1] = sg * v * v / (kap * cd2);
        g[2] = sg * 2 * v * td / kap;
        memset(H, 0, sizeof(double) * 9);
        H[0][0] = sg * 2.0;
        H[1][1] = sg * 2 * v * v * td / (kap * cd2);
        H[1][2] = H[2][1] = sg * 2 * v / (kap * cd2);
        H[2][2] = sg * 2 * td / kap;
    }
    return fabs(y);
}

static void eval_point(const prob_t* pb, const iterate_t* it, eval_t* ev, int derivs) {
    const mpco_desc* d = pb->d;
    const int N = pb->N, nx = pb->nx;
    double fc = 0.0, f[NX];
    for (int i = 0; i < nx; ++i) ev->c[0][i] = it->x[0][i] - pb->r[i];
    for (int k = 0; k < N; ++k) {
        const double* rk1 = pb->r + (size_t)(k + 1) * nx;
        for (int i = 0; i < nx; ++i) {
            const double e = it->x[k][i] - rk1[i];
            fc += d->Q[i] * e * e;
        }
        fc += d->R[0] * it->u[k][0] * it->u[k][0] + d->R[1] * it->u[k][1] * it->u[k][1];
        mpco_ode(d, it->x[k], it->u[k], f);
        for (int i = 0; i < nx; ++i) ev->c[k + 1][i] = it->x[k + 1][i] - (f[i] * d->dt + it->x[k][i]);
        if (derivs) ode_jac(d, it->x[k], ev->fx[k]);
    }
    for (int k = 0; k <= N; ++k) obstacle_eval(d, it->x[k], ev->dobs[k], ev->jo[k], ev->ho[k], derivs);
    if (pb->fric_row) {
        ev->dfric = friction_eval(d, it->u[0], it->x[0], ev->gfr, ev->hfr, derivs);
    } else {
        ev->dfric = 0.0;
        memset(ev->gfr, 0, sizeof(ev->gfr));
        memset(ev->hfr, 0, sizeof(ev->hfr));
    }
    ev->fcost = fc;
}

double mpco_objective(const mpco_desc* d, const double* w, const double* p) {
    const int N = d->N, nx = d->nx;
    const double* U = w;
    const double* X = w + 2 * N;
    const double* R = p + 2 * N;
    double fc = 0;
    for (int k = 0; k < N; ++k) {
        for (int i = 0; i < nx; ++i) {
            const double e = X[k * nx + i] - R[(k + 1) * nx + i];
            fc += d->Q[i] * e * e;
        }
        fc += d->R[0] * U[2 * k] * U[2 * k] + d->R[1] * U[2 * k + 1] * U[2 * k + 1];
    }
    return fc;
}

void mpco_constraints(const mpco_desc* d, const double* w, const double* p, double* g) {
    const int N = d->N, nx = d->nx;
    const double* U = w;
    const double* X = w + 2 * N;
    const double* R = p + 2 * N;
    double f[NX], dist[3];
    g[0] = friction_eval(d, U, X, NULL, NULL, 0);
    for (int i = 0; i < nx; ++i) g[1 + i] = X[i] - R[i];
    for (int k = 0; k < N; ++k) {
        mpco_ode(d, X + k * nx, U + 2 * k, f);
        for (int i = 0; i < nx; ++i) g[1 + nx * (k + 1) + i] = X[(k + 1) * nx + i] - (f[i] * d->dt + X[k * nx + i]);
    }
    for (int k = 0; k <= N; ++k) {
        obstacle_eval(d, X + k * nx, dist, NULL, NULL, 0);
        for (int j = 0; j < 3; ++j)
            for (int rep = 0; rep < 3; ++rep) g[1 + nx * (N + 1) + 9 * k + 3 * j + rep] = dist[j];
    }
}

/* ------------------------------------------------------------------------------------------------ helpers */
static double push_in(double v, double lo, double hi) {
    const int hl = fin(lo), hu = fin(hi);
    if (hl && hu) {
        const double pl = fmin(KAPPA_1 * fmax(1.0, fabs(lo)), KAPPA_2 * (hi - lo));
        const double pu = fmin(KAPPA_1 * fmax(1.0, fabs(hi)), KAPPA_2 * (hi - lo));
        v = fmax(v, lo + pl);
        v = fmin(v, hi - pu);
    } else if (hl) {
        v = fmax(v, lo + KAPPA_1 * fmax(1.0, fabs(lo)));
    } else if (hu) {
        v = fmin(v, hi - KAPPA_1 * fmax(1.0, fabs(hi)));
    }
    return v;
}
static double relax_lo(double lo) { return fin(lo) ? lo - BOUND_RELAX * fmax(1.0, fabs(lo)) : lo; }
static double relax_hi(double hi) { return fin(hi) ? hi + BOUND_RELAX * fmax(1.0, fabs(hi)) : hi; }

typedef struct {
    double dual_inf, prim_inf, cmin, cmax, sum_mult, sum_z;
    int n_mult, n_z;
} kkt_t;

/* one bound pair: accumulates barrier quantities */
#define FOR_BOUND(val, lo, hi, zl, zu, BODY_LO, BODY_HI) \
    do {                                                  \
        if (fin(lo)) { const double gap = (val) - (lo); const double z = (zl); BODY_LO; } \
        if (fin(hi)) { const double gap = (hi) - (val); const double z = (zu); BODY_HI; } \
    } while (0)

static void acc_compl(kkt_t* k, double gap, double z, int mult) {
    const double c = gap * z;
    if (c < k->cmin) k->cmin = c;
    if (c > k->cmax) k->cmax = c;
    k->sum_z += mult * z;
    k->n_z += mult;
}

static double barrier_phi(const prob_t* pb, const iterate_t* it, double fcost, double df, double mu) {
    const int N = pb->N, nx = pb->nx, m = pb->d->obst_mult;
    double s = 0.0;
    int bad = 0;
#define LOGGAP(g, w) do { if ((g) <= 0) bad = 1; else s += (w) * log(g); } while (0)
    for (int k = 0; k <= N; ++k) {
        if (k < N)
            for (int i = 0; i < NU; ++i) {
                if (fin(pb->lbu[k][i])) LOGGAP(it->u[k][i] - pb->lbu[k][i], 1);
                if (fin(pb->ubu[k][i])) LOGGAP(pb->ubu[k][i] - it->u[k][i], 1);
            }
        for (int i = 0; i < nx; ++i) {
            if (fin(pb->lbx[k][i])) LOGGAP(it->x[k][i] - pb->lbx[k][i], 1);
            if (fin(pb->ubx[k][i])) LOGGAP(pb->ubx[k][i] - it->x[k][i], 1);
        }
        for (int j = 0; j < 3; ++j) {
            if (pb->has_ol) LOGGAP(it->so[k][j] - pb->ol, m);
            if (pb->has_ou) LOGGAP(pb->ou - it->so[k][j], m);
        }
    }
    if (pb->has_fl) LOGGAP(it->sf - pb->fl, 1);
    if (pb->has_fu) LOGGAP(pb->fu - it->sf, 1);
#undef LOGGAP
    if (bad) return INFINITY;
    return df * fcost - mu * s;
}

static double theta_of(const prob_t* pb, const iterate_t* it, const eval_t* ev) {
    const int N = pb->N, nx = pb->nx, m = pb->d->obst_mult;
    double th = 0;
    for (int k = 0; k <= N; ++k) {
        for (int i = 0; i < nx; ++i) th += fabs(ev->c[k][i]);
        for (int j = 0; j < 3; ++j) th += m * fabs(ev->dobs[k][j] - it->so[k][j]);
    }
    th += fabs(ev->dfric - it->sf);
    return th;
}

/* ------------------------------------------------------------------------------------------------ solver */
static int solve_impl(const mpco_desc* d, const double* lbx_in, const double* ubx_in, const double* x0, const double* p,
                      double* x_out, int32_t* status_out, int32_t* iters_out, double* kkt_out, double* obj_out,
                      double* trace, int trace_cap, work_t* W_ext) {
    const int N = d->N, nx = d->nx, m = d->obst_mult;
    if (N < 1 || N > MPCO_N_MAX || (nx != 5 && nx != 6)) return -1;
    work_t* W = W_ext ? W_ext : (work_t*)malloc(sizeof(work_t));
    if (!W) return -2;
    memset(W, 0, sizeof(work_t));
    prob_t pbs, *pb = &pbs;
    memset(pb, 0, sizeof(*pb));
    pb->N = N; pb->nx = nx; pb->d = d; pb->r = p + 2 * N;
    for (int k = 0; k <= N; ++k) {
        if (k < N)
            for (int i = 0; i < NU; ++i) {
                pb->lbu[k][i] = relax_lo(lbx_in[2 * k + i]);
                pb->ubu[k][i] = relax_hi(ubx_in[2 * k + i]);
            }
        for (int i = 0; i < nx; ++i) {
            pb->lbx[k][i] = relax_lo(lbx_in[2 * N + nx * k + i]);
            pb->ubx[k][i] = relax_hi(ubx_in[2 * N + nx * k + i]);
        }
    }
    /* |y| >= lo <= 0 is vacuous -- unless the row is asked for as IPOPT sees it (reserved == 2: a slack with both bounds, log barrier on
     * the lower one too; optimizer.py:378, 424-425) */
    pb->has_fl = fin(d->fric_lo) && (d->fric_lo > 0.0 || d->reserved == 2);
    pb->has_fu = fin(d->fric_hi);
    pb->fl = relax_lo(d->fric_lo); pb->fu = relax_hi(d->fric_hi);
    pb->fric_row = 1;
    if (!d->reserved && !pb->has_fl && pb->has_fu) {
        /* presolve: x_0 is pinned to r_0 by the equality rows, so |a_0^2 + c| <= fu with c = v_0^2 tan(delta_0)/kappa
         * evaluated at r_0 is the simple bound a_0^2 <= fu - c (valid when also -fu - c <= 0, i.e. the lower branch
         * of the absolute value cannot bind).  The row has zero gradient at the usual warm start a_0 = 0, which
         * an SQP/IPM linearisation cannot see; the bound form is exact and has the same KKT points. */
        const double cf = pb->r[3] * (tan(pb->r[2]) * pb->r[3] / d->friction_div);
        const double Rhi = pb->fu - cf, Rlo = -pb->fu - cf;
        if (Rhi > 0.0 && Rlo <= 0.0) {
            const double amax = sqrt(Rhi);
            pb->lbu[0][1] = fmax(pb->lbu[0][1], -amax);
            pb->ubu[0][1] = fmin(pb->ubu[0][1], amax);
            pb->fric_row = 0;
            pb->has_fu = 0;
        }
    }
    pb->has_ol = fin(d->obst_lo); pb->has_ou = fin(d->obst_hi);
    pb->ol = relax_lo(d->obst_lo); pb->ou = relax_hi(d->obst_hi);
    const double fsl = pb->has_fl ? pb->fl : -INFINITY, fsu = pb->has_fu ? pb->fu : INFINITY;
    const double osl = pb->has_ol ? pb->ol : -INFINITY, osu = pb->has_ou ? pb->ou : INFINITY;

    iterate_t* it = &W->it;
    eval_t* ev = &W->ev;
    /* start-point safeguard: the caller's state guess is replaced by a forward rollout of the (bound-projected)
     * control guess when its dynamics defect is more than ROLLOUT_FACTOR x worse (the reference's first MPC
     * step passes a transposed state guess, SURVEY.md App. C-6, from which an interior-point method jams) */
    double* xs = (double*)malloc(sizeof(double) * (size_t)(N + 1) * NX);
    {
        double xg[NX], xr[NX], f[NX], fr[NX], u[NU], th_g = 0.0, th_r = 0.0;
        double* roll = (double*)malloc(sizeof(double) * (size_t)(N + 1) * NX);
        for (int i = 0; i < nx; ++i) {
            xg[i] = push_in(x0[2 * N + i], pb->lbx[0][i], pb->ubx[0][i]);
            xr[i] = push_in(pb->r[i], pb->lbx[0][i], pb->ubx[0][i]);
            th_g += fabs(xg[i] - pb->r[i]);
            th_r += fabs(xr[i] - pb->r[i]);
            roll[i] = xr[i];
        }
        for (int k = 0; k < N; ++k) {
            for (int i = 0; i < NU; ++i) u[i] = push_in(x0[2 * k + i], pb->lbu[k][i], pb->ubu[k][i]);
            mpco_ode(d, xg, u, f);
            mpco_ode(d, xr, u, fr);
            for (int i = 0; i < nx; ++i) {
                const double gn = push_in(x0[2 * N + nx * (k + 1) + i], pb->lbx[k + 1][i], pb->ubx[k + 1][i]);
                th_g += fabs(gn - (f[i] * d->dt + xg[i]));
                xg[i] = gn;
                const double rraw = fr[i] * d->dt + xr[i];
                const double rn = push_in(rraw, pb->lbx[k + 1][i], pb->ubx[k + 1][i]);
                th_r += fabs(rn - rraw);
                xr[i] = rn;
                roll[(size_t)(k + 1) * NX + i] = rn;
            }
        }
        const int use = !(th_g <= ROLLOUT_FACTOR * fmax(1.0, th_r));
        for (int k = 0; k <= N; ++k)
            for (int i = 0; i < nx; ++i) xs[(size_t)k * NX + i] = use ? roll[(size_t)k * NX + i] : x0[2 * N + nx * k + i];
        free(roll);
    }
    /* objective scaling at the start point */
    double gmax = 0.0;
    for (int k = 0; k < N; ++k) {
        for (int i = 0; i < nx; ++i) gmax = fmax(gmax, fabs(2 * d->Q[i] * (xs[(size_t)k * NX + i] - pb->r[(k + 1) * nx + i])));
        for (int i = 0; i < NU; ++i) gmax = fmax(gmax, fabs(2 * d->R[i] * x0[2 * k + i]));
    }
    const double df = gmax > SCALING_MAX_GRAD ? SCALING_MAX_GRAD / gmax : 1.0;
    /* start point */
    for (int k = 0; k <= N; ++k) {
        if (k < N)
            for (int i = 0; i < NU; ++i) {
                it->u[k][i] = push_in(x0[2 * k + i], pb->lbu[k][i], pb->ubu[k][i]);
                it->zlu[k][i] = fin(pb->lbu[k][i]) ? 1.0 : 0.0;
                it->zuu[k][i] = fin(pb->ubu[k][i]) ? 1.0 : 0.0;
            }
        for (int i = 0; i < nx; ++i) {
            it->x[k][i] = push_in(xs[(size_t)k * NX + i], pb->lbx[k][i], pb->ubx[k][i]);
            it->zlx[k][i] = fin(pb->lbx[k][i]) ? 1.0 : 0.0;
            it->zux[k][i] = fin(pb->ubx[k][i]) ? 1.0 : 0.0;
        }
    }
    free(xs);
    eval_point(pb, it, ev, 0);
    for (int k = 0; k <= N; ++k)
        for (int j = 0; j < 3; ++j) {
            it->so[k][j] = push_in(ev->dobs[k][j], osl, osu);
            it->zlo[k][j] = pb->has_ol ? 1.0 : 0.0;
            it->zuo[k][j] = pb->has_ou ? 1.0 : 0.0;
        }
    it->sf = push_in(ev->dfric, fsl, fsu);
    it->zlf = pb->has_fl ? 1.0 : 0.0;
    it->zuf = pb->has_fu ? 1.0 : 0.0;

    double mu = MU_INIT, tau = fmax(TAU_MIN, 1 - mu);
    double filt_th[FILTER_MAX], filt_ph[FILTER_MAX];
    int nfilt = 0;
    double delta_last = 0.0, theta_max = 0, theta_min = 0;
    int have_theta0 = 0, iter = 0, status = MPCO_MAXITER, conv_seen = 0;
    double E0 = NAN;
    const int iter_cap = d->fixed_iters > 0 ? d->fixed_iters : d->max_iter;

    for (;;) {
        eval_point(pb, it, ev, 1);
        /* ---------------- residuals / KKT error */
        kkt_t kk;
        kk.dual_inf = kk.prim_inf = kk.sum_mult = kk.sum_z = 0;
        kk.cmin = INFINITY; kk.cmax = -INFINITY; kk.n_mult = kk.n_z = 0;
        int nanflag = !fin(ev->fcost);
        for (int k = 0; k <= N; ++k) {
            double rx[NX], ru[NU] = {0, 0};
            for (int i = 0; i < nx; ++i) {
                double g = (k < N) ? df * 2 * d->Q[i] * (it->x[k][i] - pb->r[(k + 1) * nx + i]) : 0.0;
                W->gradfx[k][i] = g;
                g += it->lam[k][i];
                if (k < N) {
                    g -= it->lam[k + 1][i];
                    for (int r = 0; r < nx; ++r) g -= d->dt * ev->fx[k][r][i] * it->lam[k + 1][r];
                }
                rx[i] = g - it->zlx[k][i] + it->zux[k][i];
            }
            static const int oi[3] = {0, 1, 4};
            for (int j = 0; j < 3; ++j)
                for (int a = 0; a < 3; ++a) rx[oi[a]] += m * it->nuo[k][j] * ev->jo[k][j][a];
            if (k < N) {
                for (int i = 0; i < NU; ++i) {
                    W->gradfu[k][i] = df * 2 * d->R[i] * it->u[k][i];
                    /* d c_{k+1} / d u = -dt * Fu, Fu = [e_delta e_v] */
                    ru[i] = W->gradfu[k][i] - d->dt * it->lam[k + 1][2 + i] - it->zlu[k][i] + it->zuu[k][i];
                }
            }
            if (k == 0) {
                ru[1] += it->nuf * ev->gfr[0];
                rx[2] += it->nuf * ev->gfr[1];
                rx[3] += it->nuf * ev->gfr[2];
            }
            for (int i = 0; i < nx; ++i) {
                kk.dual_inf = fmax(kk.dual_inf, fabs(rx[i]));
                if (!fin(rx[i])) nanflag = 1;
                kk.prim_inf = fmax(kk.prim_inf, fabs(ev->c[k][i]));
                kk.sum_mult += fabs(it->lam[k][i]);
                FOR_BOUND(it->x[k][i], pb->lbx[k][i], pb->ubx[k][i], it->zlx[k][i], it->zux[k][i],
                          acc_compl(&kk, gap, z, 1), acc_compl(&kk, gap, z, 1));
            }
            kk.n_mult += nx;
            if (k < N)
                for (int i = 0; i < NU; ++i) {
                    kk.dual_inf = fmax(kk.dual_inf, fabs(ru[i]));
                    FOR_BOUND(it->u[k][i], pb->lbu[k][i], pb->ubu[k][i], it->zlu[k][i], it->zuu[k][i],
                              acc_compl(&kk, gap, z, 1), acc_compl(&kk, gap, z, 1));
                }
            for (int j = 0; j < 3; ++j) {
                kk.dual_inf = fmax(kk.dual_inf, fabs(-it->nuo[k][j] - it->zlo[k][j] + it->zuo[k][j]));
                kk.prim_inf = fmax(kk.prim_inf, fabs(ev->dobs[k][j] - it->so[k][j]));
                kk.sum_mult += m * fabs(it->nuo[k][j]);
                kk.n_mult += m;
                FOR_BOUND(it->so[k][j], osl, osu, it->zlo[k][j], it->zuo[k][j],
                          acc_compl(&kk, gap, z, m), acc_compl(&kk, gap, z, m));
            }
        }
        kk.dual_inf = fmax(kk.dual_inf, fabs(-it->nuf - it->zlf + it->zuf));
        kk.prim_inf = fmax(kk.prim_inf, fabs(ev->dfric - it->sf));
        kk.sum_mult += fabs(it->nuf);
        kk.n_mult += pb->fric_row;
        FOR_BOUND(it->sf, fsl, fsu, it->zlf, it->zuf, acc_compl(&kk, gap, z, 1), acc_compl(&kk, gap, z, 1));
        const int nden = kk.n_mult + kk.n_z;
        const double s_d = fmax(S_MAX, (kk.sum_mult + kk.sum_z) / (nden > 0 ? nden : 1)) / S_MAX;
        const double s_c = fmax(S_MAX, kk.sum_z / (kk.n_z > 0 ? kk.n_z : 1)) / S_MAX;
#define EMU(mu_) fmax(fmax(kk.dual_inf / s_d, kk.prim_inf), (kk.n_z ? fmax(kk.cmax - (mu_), (mu_) - kk.cmin) : 0.0) / s_c)
        E0 = EMU(0.0);
        if (nanflag || !fin(E0)) { status = MPCO_NAN; break; }
        if (d->fixed_iters <= 0 && E0 <= d->tol) { status = MPCO_CONVERGED; break; }
        if (d->fixed_iters > 0 && E0 <= d->tol) conv_seen = 1;   /* benchmark mode: keep stepping, accept what comes */
        if (iter >= iter_cap) { status = d->fixed_iters > 0 ? MPCO_CONVERGED : MPCO_MAXITER; break; }
        /* ---------------- barrier parameter */
        int mu_changed = 0;
        while (EMU(mu) <= KAPPA_EPS * mu) {
            const double nm = fmax(d->tol / 10.0, fmin(KAPPA_MU * mu, pow(mu, THETA_MU)));
            if (nm == mu) break;
            mu = nm;
            tau = fmax(TAU_MIN, 1 - mu);
            mu_changed = 1;
        }
        if (mu_changed) nfilt = 0;
        /* ---------------- condensed stage blocks */
        for (int k = 0; k <= N; ++k) {
            double (*H)[NX] = W->Hxx[k];
            memset(H, 0, sizeof(double) * NX * NX);
            if (k < N) {
                for (int i = 0; i < nx; ++i) H[i][i] = df * 2 * d->Q[i];
                ode_hess_add(d, it->x[k], it->lam[k + 1], -d->dt, H);
            }
            static const int oi[3] = {0, 1, 4};
            for (int i = 0; i < nx; ++i) {
                double sg = 0, gb = 0;
                FOR_BOUND(it->x[k][i], pb->lbx[k][i], pb->ubx[k][i], it->zlx[k][i], it->zux[k][i],
                          (sg += z / gap, gb -= mu / gap), (sg += z / gap, gb += mu / gap));
                H[i][i] += sg;
                W->gbx[k][i] = gb;
                W->gx[k][i] = W->gradfx[k][i] + gb;
            }
            for (int j = 0; j < 3; ++j) {
                double sg = 0, gb = 0;
                FOR_BOUND(it->so[k][j], osl, osu, it->zlo[k][j], it->zuo[k][j],
                          (sg += z / gap, gb -= mu / gap), (sg += z / gap, gb += mu / gap));
                W->sgo[k][j] = sg;
                W->gbo[k][j] = gb;
                const double rs = ev->dobs[k][j] - it->so[k][j];
                const double coef = m * (gb + sg * rs);
                for (int a = 0; a < 3; ++a) {
                    W->gx[k][oi[a]] += ev->jo[k][j][a] * coef;
                    for (int b = 0; b < 3; ++b)
                        H[oi[a]][oi[b]] += m * (it->nuo[k][j] * ev->ho[k][j][a][b] + sg * ev->jo[k][j][a] * ev->jo[k][j][b]);
                }
            }
            if (k < N) {
                for (int i = 0; i < NU; ++i) {
                    double sg = 0, gb = 0;
                    FOR_BOUND(it->u[k][i], pb->lbu[k][i], pb->ubu[k][i], it->zlu[k][i], it->zuu[k][i],
                              (sg += z / gap, gb -= mu / gap), (sg += z / gap, gb += mu / gap));
                    W->Huu[k][i][i] = df * 2 * d->R[i] + sg;
                    W->Huu[k][i][1 - i] = 0.0;
                    W->gbu[k][i] = gb;
                    W->gu[k][i] = W->gradfu[k][i] + gb;
                }
            }
            if (k == 0) {
                double sg = 0, gb = 0;
                FOR_BOUND(it->sf, fsl, fsu, it->zlf, it->zuf, (sg += z / gap, gb -= mu / gap), (sg += z / gap, gb += mu / gap));
                W->sgf = sg;
                W->gbf = gb;
                const double coef = gb + sg * (ev->dfric - it->sf);
                const double* g = ev->gfr;
                memset(W->Hux0, 0, sizeof(W->Hux0));
                W->gu[0][1] += g[0] * coef;
                W->gx[0][2] += g[1] * coef;
                W->gx[0][3] += g[2] * coef;
                W->Huu[0][1][1] += it->nuf * ev->hfr[0][0] + sg * g[0] * g[0];
                W->Hux0[1][2] = sg * g[0] * g[1];
                W->Hux0[1][3] = sg * g[0] * g[2];
                H[2][2] += it->nuf * ev->hfr[1][1] + sg * g[1] * g[1];
                H[2][3] += it->nuf * ev->hfr[1][2] + sg * g[1] * g[2];
                H[3][2] += it->nuf * ev->hfr[2][1] + sg * g[2] * g[1];
                H[3][3] += it->nuf * ev->hfr[2][2] + sg * g[2] * g[2];
            }
        }
        /* ---------------- Riccati with inertia correction */
        double delta = 0.0;
        int ok = 0;
        for (;;) {
            ok = 1;
            for (int i = 0; i < nx; ++i) {
                for (int j = 0; j < nx; ++j) W->P[N][i][j] = W->Hxx[N][i][j];
                W->P[N][i][i] += delta;
                W->pv[N][i] = W->gx[N][i];
            }
            for (int k = N - 1; k >= 0 && ok; --k) {
                double A[NX][NX], PA[NX][NX], h[NX], G[NU][NX], L[NU][NU], l[NU];
                for (int i = 0; i < nx; ++i)
                    for (int j = 0; j < nx; ++j) A[i][j] = (i == j) + d->dt * ev->fx[k][i][j];
                /* b_k = -c_{k+1};  h = p+ + P+ b */
                for (int i = 0; i < nx; ++i) {
                    double s = W->pv[k + 1][i];
                    for (int j = 0; j < nx; ++j) s -= W->P[k + 1][i][j] * ev->c[k + 1][j];
                    h[i] = s;
                }
                for (int i = 0; i < nx; ++i)
                    for (int j = 0; j < nx; ++j) {
                        double s = 0;
                        for (int r = 0; r < nx; ++r) s += W->P[k + 1][i][r] * A[r][j];
                        PA[i][j] = s;
                    }
                /* B = dt * [e_2 e_3]  =>  B' M = dt * rows (2,3) of M */
                for (int a = 0; a < NU; ++a) {
                    for (int j = 0; j < nx; ++j) G[a][j] = d->dt * PA[2 + a][j] + (k == 0 ? W->Hux0[a][j] : 0.0);
                    for (int b = 0; b < NU; ++b) L[a][b] = W->Huu[k][a][b] + d->dt * d->dt * W->P[k + 1][2 + a][2 + b];
                    L[a][a] += delta;
                    l[a] = W->gu[k][a] + d->dt * h[2 + a];
                }
                /* Lam = Ruu + B'P+B is dominated by a rank-one term w g g' once a circle row is active (w = z / s up to 1e13): L00 L11 and L01^2
                 * agree to ten digits, and so do the two products of every entry of adj(Lam) G.  Differences of products with the rounding
                 * error of the subtracted product carried along (Kahan); without it the collision-avoidance family has instances that wander
                 * at a KKT error of 1e-8 ... 1e-4 for dozens of iterations (profiles/r05_ca_lottery.txt). */
                const double l01sq = L[0][1] * L[1][0];
                const double det = fma(L[0][0], L[1][1], -l01sq) - fma(L[0][1], L[1][0], -l01sq);
                if (!(L[0][0] > 0.0) || !(det > 0.0)) { ok = 0; break; }
#define DOP(a, b, c, d) (fma((a), (b), -((c) * (d))) - fma((c), (d), -((c) * (d))))      /* a b - c d */
                for (int j = 0; j < nx; ++j) {
                    W->K[k][0][j] = -DOP(L[1][1], G[0][j], L[0][1], G[1][j]) / det;
                    W->K[k][1][j] = -DOP(L[0][0], G[1][j], L[0][1], G[0][j]) / det;
                }
                W->kff[k][0] = -DOP(L[1][1], l[0], L[0][1], l[1]) / det;
                W->kff[k][1] = -DOP(L[0][0], l[1], L[0][1], l[0]) / det;
#undef DOP
                for (int i = 0; i < nx; ++i) {
                    for (int j = 0; j < nx; ++j) {
                        double s = W->Hxx[k][i][j] + (i == j ? delta : 0.0);
                        for (int r = 0; r < nx; ++r) s += A[r][i] * PA[r][j];
                        s += G[0][i] * W->K[k][0][j] + G[1][i] * W->K[k][1][j];
                        W->P[k][i][j] = s;
                    }
                    double s = W->gx[k][i];
                    for (int r = 0; r < nx; ++r) s += A[r][i] * h[r];
                    s += G[0][i] * W->kff[k][0] + G[1][i] * W->kff[k][1];
                    W->pv[k][i] = s;
                }
                /* keep P symmetric against round-off */
                for (int i = 0; i < nx; ++i)
                    for (int j = i + 1; j < nx; ++j) {
                        const double s = 0.5 * (W->P[k][i][j] + W->P[k][j][i]);
                        W->P[k][i][j] = W->P[k][j][i] = s;
                    }
            }
            if (ok) break;
            if (delta == 0.0) delta = (delta_last == 0.0) ? DW_0 : fmax(DW_MIN, KW_MINUS * delta_last);
            else delta *= (delta_last == 0.0) ? KW_PLUS_BAR : KW_PLUS;
            if (delta > DW_MAX) break;
        }
        if (!ok) { status = MPCO_NOPROGRESS; break; }
        if (delta > 0.0) delta_last = delta;
        /* ---------------- forward sweep + dual steps */
        iterate_t* st = &W->step;
        for (int i = 0; i < nx; ++i) st->x[0][i] = -ev->c[0][i];
        for (int k = 0; k <= N; ++k) {
            if (k < N) {
                for (int a = 0; a < NU; ++a) {
                    double s = W->kff[k][a];
                    for (int j = 0; j < nx; ++j) s += W->K[k][a][j] * st->x[k][j];
                    st->u[k][a] = s;
                }
                for (int i = 0; i < nx; ++i) {
                    double s = st->x[k][i] - ev->c[k + 1][i];
                    for (int j = 0; j < nx; ++j) s += d->dt * ev->fx[k][i][j] * st->x[k][j];
                    st->x[k + 1][i] = s;
                }
                st->x[k + 1][2] += d->dt * st->u[k][0];
                st->x[k + 1][3] += d->dt * st->u[k][1];
            }
            for (int i = 0; i < nx; ++i) {
                double s = W->pv[k][i];
                for (int j = 0; j < nx; ++j) s += W->P[k][i][j] * st->x[k][j];
                st->lam[k][i] = -s - it->lam[k][i];
            }
        }
        double a_pr = 1.0, a_du = 1.0, dphi = 0.0;
#define FTB_PR(gap, dgap) do { if ((dgap) < 0) a_pr = fmin(a_pr, -tau * (gap) / (dgap)); } while (0)
#define FTB_DU(z, dz) do { if ((dz) < 0) a_du = fmin(a_du, -tau * (z) / (dz)); } while (0)
        static const int oi[3] = {0, 1, 4};
        for (int k = 0; k <= N; ++k) {
            for (int i = 0; i < nx; ++i) {
                const double dv = st->x[k][i];
                dphi += (W->gradfx[k][i] + W->gbx[k][i]) * dv;
                st->zlx[k][i] = st->zux[k][i] = 0.0;
                if (fin(pb->lbx[k][i])) {
                    const double gap = it->x[k][i] - pb->lbx[k][i], z = it->zlx[k][i];
                    st->zlx[k][i] = mu / gap - z - z / gap * dv;
                    FTB_PR(gap, dv); FTB_DU(z, st->zlx[k][i]);
                }
                if (fin(pb->ubx[k][i])) {
                    const double gap = pb->ubx[k][i] - it->x[k][i], z = it->zux[k][i];
                    st->zux[k][i] = mu / gap - z + z / gap * dv;
                    FTB_PR(gap, -dv); FTB_DU(z, st->zux[k][i]);
                }
            }
            if (k < N)
                for (int i = 0; i < NU; ++i) {
                    const double dv = st->u[k][i];
                    dphi += (W->gradfu[k][i] + W->gbu[k][i]) * dv;
                    st->zlu[k][i] = st->zuu[k][i] = 0.0;
                    if (fin(pb->lbu[k][i])) {
                        const double gap = it->u[k][i] - pb->lbu[k][i], z = it->zlu[k][i];
                        st->zlu[k][i] = mu / gap - z - z / gap * dv;
                        FTB_PR(gap, dv); FTB_DU(z, st->zlu[k][i]);
                    }
                    if (fin(pb->ubu[k][i])) {
                        const double gap = pb->ubu[k][i] - it->u[k][i], z = it->zuu[k][i];
                        st->zuu[k][i] = mu / gap - z + z / gap * dv;
                        FTB_PR(gap, -dv); FTB_DU(z, st->zuu[k][i]);
                    }
                }
            for (int j = 0; j < 3; ++j) {
                double ds = ev->dobs[k][j] - it->so[k][j];
                for (int a = 0; a < 3; ++a) ds += ev->jo[k][j][a] * st->x[k][oi[a]];
                st->so[k][j] = ds;
                st->nuo[k][j] = W->gbo[k][j] - it->nuo[k][j] + W->sgo[k][j] * ds;
                dphi += m * W->gbo[k][j] * ds;
                st->zlo[k][j] = st->zuo[k][j] = 0.0;
                if (pb->has_ol) {
                    const double gap = it->so[k][j] - pb->ol, z = it->zlo[k][j];
                    st->zlo[k][j] = mu / gap - z - z / gap * ds;
                    FTB_PR(gap, ds); FTB_DU(z, st->zlo[k][j]);
                }
                if (pb->has_ou) {
                    const double gap = pb->ou - it->so[k][j], z = it->zuo[k][j];
                    st->zuo[k][j] = mu / gap - z + z / gap * ds;
                    FTB_PR(gap, -ds); FTB_DU(z, st->zuo[k][j]);
                }
            }
        }
        {
            double ds = ev->dfric - it->sf + ev->gfr[0] * st->u[0][1] + ev->gfr[1] * st->x[0][2] + ev->gfr[2] * st->x[0][3];
            st->sf = ds;
            st->nuf = W->gbf - it->nuf + W->sgf * ds;
            dphi += W->gbf * ds;
            st->zlf = st->zuf = 0.0;
            if (pb->has_fl) {
                const double gap = it->sf - pb->fl, z = it->zlf;
                st->zlf = mu / gap - z - z / gap * ds;
                FTB_PR(gap, ds); FTB_DU(z, st->zlf);
            }
            if (pb->has_fu) {
                const double gap = pb->fu - it->sf, z = it->zuf;
                st->zuf = mu / gap - z + z / gap * ds;
                FTB_PR(gap, -ds); FTB_DU(z, st->zuf);
            }
        }
        /* ---------------- filter line search */
        const double theta = theta_of(pb, it, ev);
        if (!have_theta0) {
            have_theta0 = 1;
            theta_max = 1e4 * fmax(1.0, theta);
            theta_min = 1e-4 * fmax(1.0, theta);
        }
        const double phi = barrier_phi(pb, it, ev->fcost, df, mu);
        double a_min;
        if (dphi < 0 && theta <= theta_min)
            a_min = fmin(fmin(GAMMA_THETA, GAMMA_PHI * theta / (-dphi)), LS_DELTA * pow(theta, S_THETA) / pow(-dphi, S_PHI));
        else if (dphi < 0)
            a_min = fmin(GAMMA_THETA, GAMMA_PHI * theta / (-dphi));
        else
            a_min = GAMMA_THETA;
        a_min *= GAMMA_ALPHA;
        if (conv_seen) a_min = 0.0;
        double alpha = a_pr;
        int accepted = 0, ftype = 0, ntrial = 0;
        iterate_t* tr = &W->trial;
        while (alpha >= a_min) {
            for (int k = 0; k <= N; ++k) {
                for (int i = 0; i < nx; ++i) tr->x[k][i] = it->x[k][i] + alpha * st->x[k][i];
                if (k < N) for (int i = 0; i < NU; ++i) tr->u[k][i] = it->u[k][i] + alpha * st->u[k][i];
                for (int j = 0; j < 3; ++j) tr->so[k][j] = it->so[k][j] + alpha * st->so[k][j];
            }
            tr->sf = it->sf + alpha * st->sf;
            eval_point(pb, tr, &W->evt, 0);
            const double th_t = theta_of(pb, tr, &W->evt);
            const double ph_t = barrier_phi(pb, tr, W->evt.fcost, df, mu);
            ++ntrial;
            int good = fin(th_t) && fin(ph_t) && th_t <= theta_max;
            for (int q = 0; q < nfilt && good; ++q)
                if (!(cmp_le(fmax(th_t, THETA_FLOOR), fmax(filt_th[q], THETA_FLOOR), filt_th[q]) || cmp_le(ph_t, filt_ph[q], filt_ph[q]))) good = 0;
            if (good && conv_seen) {
                accepted = 1; ftype = 1;
            } else if (good) {
                const int sw = theta <= theta_min && dphi < 0 && alpha * pow(-dphi, S_PHI) > LS_DELTA * pow(theta, S_THETA);
                if (sw) {
                    if (cmp_le(ph_t - phi, ETA_PHI * alpha * dphi, phi)) { accepted = 1; ftype = 1; }
                } else if (cmp_le(fmax(th_t, THETA_FLOOR), fmax((1 - GAMMA_THETA) * theta, THETA_FLOOR), theta) ||
                           cmp_le(ph_t - phi, -GAMMA_PHI * theta, phi)) {
                    accepted = 1;
                }
            }
            if (accepted) break;
            alpha *= 0.5;
            if (ntrial >= 64) break;
        }
        if (trace && iter < trace_cap) {
            double* t = trace + (size_t)iter * 8;
            t[0] = mu; t[1] = theta; t[2] = phi; t[3] = accepted ? alpha : 0.0; t[4] = a_du; t[5] = delta; t[6] = E0; t[7] = ntrial;
        }
        if (!accepted) { status = MPCO_NOPROGRESS; break; }
        if (!ftype) {
            if (nfilt == FILTER_MAX) {
                memmove(filt_th, filt_th + 1, sizeof(double) * (FILTER_MAX - 1));
                memmove(filt_ph, filt_ph + 1, sizeof(double) * (FILTER_MAX - 1));
                --nfilt;
            }
            filt_th[nfilt] = (1 - GAMMA_THETA) * theta;
            filt_ph[nfilt] = phi - GAMMA_PHI * theta;
            ++nfilt;
        }
        /* ---------------- update + multiplier reset */
#define ZRESET(z, gap) do { const double lo_ = mu / (KAPPA_SIGMA * (gap)), hi_ = KAPPA_SIGMA * mu / (gap); \
                            if ((z) < lo_) (z) = lo_;                                                        \
                            if ((z) > hi_) (z) = hi_; } while (0)
        for (int k = 0; k <= N; ++k) {
            for (int i = 0; i < nx; ++i) {
                it->x[k][i] = tr->x[k][i];
                it->lam[k][i] += alpha * st->lam[k][i];
                if (fin(pb->lbx[k][i])) { it->zlx[k][i] += a_du * st->zlx[k][i]; ZRESET(it->zlx[k][i], it->x[k][i] - pb->lbx[k][i]); }
                if (fin(pb->ubx[k][i])) { it->zux[k][i] += a_du * st->zux[k][i]; ZRESET(it->zux[k][i], pb->ubx[k][i] - it->x[k][i]); }
            }
            if (k < N)
                for (int i = 0; i < NU; ++i) {
                    it->u[k][i] = tr->u[k][i];
                    if (fin(pb->lbu[k][i])) { it->zlu[k][i] += a_du * st->zlu[k][i]; ZRESET(it->zlu[k][i], it->u[k][i] - pb->lbu[k][i]); }
                    if (fin(pb->ubu[k][i])) { it->zuu[k][i] += a_du * st->zuu[k][i]; ZRESET(it->zuu[k][i], pb->ubu[k][i] - it->u[k][i]); }
                }
            for (int j = 0; j < 3; ++j) {
                it->so[k][j] = tr->so[k][j];
                it->nuo[k][j] += alpha * st->nuo[k][j];
                if (pb->has_ol) { it->zlo[k][j] += a_du * st->zlo[k][j]; ZRESET(it->zlo[k][j], it->so[k][j] - pb->ol); }
                if (pb->has_ou) { it->zuo[k][j] += a_du * st->zuo[k][j]; ZRESET(it->zuo[k][j], pb->ou - it->so[k][j]); }
            }
        }
        it->sf = tr->sf;
        it->nuf += alpha * st->nuf;
        if (pb->has_fl) { it->zlf += a_du * st->zlf; ZRESET(it->zlf, it->sf - pb->fl); }
        if (pb->has_fu) { it->zuf += a_du * st->zuf; ZRESET(it->zuf, pb->fu - it->sf); }
        ++iter;
    }
    for (int k = 0; k <= N; ++k) {
        if (k < N) for (int i = 0; i < NU; ++i) x_out[2 * k + i] = it->u[k][i];
        for (int i = 0; i < nx; ++i) x_out[2 * N + nx * k + i] = it->x[k][i];
    }
    if (status_out) *status_out = status;
    if (iters_out) *iters_out = iter;
    if (kkt_out) *kkt_out = E0;
    if (obj_out) *obj_out = ev->fcost;
    if (!W_ext) free(W);
    return 0;
}

int mpco_solve(const mpco_desc* d, const double* lbx, const double* ubx, const double* x0, const double* p,
               double* x_out, int32_t* status, int32_t* iters, double* kkt, double* obj) {
    return solve_impl(d, lbx, ubx, x0, p, x_out, status, iters, kkt, obj, NULL, 0, NULL);
}

int mpco_solve_trace(const mpco_desc* d, const double* lbx, const double* ubx, const double* x0, const double* p,
                     double* x_out, int32_t* status, int32_t* iters, double* kkt, double* obj, double* trace, int32_t trace_cap) {
    return solve_impl(d, lbx, ubx, x0, p, x_out, status, iters, kkt, obj, trace, trace_cap, NULL);
}

int mpco_solve_batch(const mpco_desc* d, const double* lbx, const double* ubx, int32_t B, const double* x0, const double* p,
                     double* x_out, int32_t* status, int32_t* iters, double* kkt, int32_t nthreads) {
    const size_t nw = (size_t)2 * d->N + (size_t)d->nx * (d->N + 1);
    int rc = 0;
#ifdef _OPENMP
    if (nthreads > 1) omp_set_num_threads(nthreads);
#pragma omp parallel if (nthreads > 1)
#endif
    {
        /* one workspace per thread, kept for the thread's lifetime: a malloc of this size is an mmap, and a team of 128 threads mapping and
         * unmapping (and faulting in) its workspaces on every call made every other batch 15 x slower than its neighbours (6.5 / 94 ms) */
        static _Thread_local work_t* tls_W = NULL;
        if (!tls_W) tls_W = (work_t*)malloc(sizeof(work_t));
        /* (never freed: it lives as long as the thread.  A failed malloc leaves NULL: solve_impl then allocates -- and checks -- a workspace of its
         *  own per instance.  solve_impl zeroes the whole workspace first thing, so an instance never sees what the previous one left in it.) */
        work_t* W = tls_W;
#ifdef _OPENMP
#pragma omp for schedule(dynamic, 4)
#endif
        for (int b = 0; b < B; ++b) {
            const int r = solve_impl(d, lbx, ubx, x0 + b * nw, p + b * nw, x_out + b * nw, status ? status + b : NULL,
                                     iters ? iters + b : NULL, kkt ? kkt + b : NULL, NULL, NULL, 0, W);
            if (r != 0) rc = r;
        }
    }
    return rc;
}

int mpco_version(void) { return 1; }
