"""
ORACLE (test infrastructure, not product code) -- dense-linear-algebra interior-point solver.

Restates the published IPOPT algorithm (Waechter & Biegler, "On the implementation of an interior-point
filter line-search algorithm for large-scale nonlinear programming", Math. Prog. 106, 2006) as far as the
reference uses it: `ca.nlpsol('solver','ipopt', ...)` with max_iter=100, tol=1e-8 (default), monotone mu
(optimizer.py:556-558, called at optimizer.py:607).  CasADi >= 3.5.1 (requirments:5, unpinned) bundles IPOPT
3.12/3.13 + MUMPS; none of it is present in the build container, so this restatement is

    PARITY UNPINNED against IPOPT itself; pinned instead against two independent scipy solvers
    (tests/golden/make_golden.py) on the same NLP.

This module is the *literal* variant: every g-row with lbg < ubg gets its own slack (so each obstacle distance
appears three times, as in optimizer.py:395-403), and the KKT system is solved densely with numpy.  The C
oracle (oracle/mpc_oracle.c) and the HIP kernels use the stage-structured Riccati solve and a weight-3
compression of the duplicated rows; tests check that all three produce the same iterates/optimum.

Algorithm summary (paper section numbers in brackets):
  * slacks for inequality rows, d(w) - s = 0, log barrier on all finite bounds of (w, s)            [3.4? / eq.(3)]
  * bounds relaxed by 1e-8*max(1,|b|) (bound_relax_factor)                                            [3.5]
  * starting point pushed into the interior with kappa_1 = kappa_2 = 1e-2, bound multipliers = 1     [3.6]
    (equality multipliers start at 0: IPOPT's least-squares estimate is skipped -- documented deviation)
  * objective scaled so that |grad f(w0)|_inf <= 100 (nlp_scaling_method = gradient-based)           [3.8]
  * monotone barrier update mu <- max(tol/10, min(0.2 mu, mu^1.5)), kappa_eps = 10, tau = max(.99,1-mu) [eq.(7),(8)]
  * primal-dual Newton step on the condensed system, inertia correction by delta_w*I                 [3.1]
  * fraction-to-boundary rule, separate primal / dual step lengths                                   [eq.(15)]
  * filter line search with switching + Armijo conditions (no second-order correction)               [2.3]
  * if the line search hits alpha_min: accept a step that reduces the primal-dual error, else give up
    (IPOPT would enter feasibility restoration; never needed on the MPC instances here)              [3.3]
  * bound-multiplier reset with kappa_Sigma = 1e10                                                   [eq.(16)]
  * termination on the scaled optimality error E_0 <= tol                                            [eq.(5),(6)]
"""
from __future__ import annotations

import numpy as np

OPTS = dict(
    tol=1e-8, max_iter=100, mu_init=0.1, kappa_eps=10.0, kappa_mu=0.2, theta_mu=1.5, tau_min=0.99,
    kappa_1=1e-2, kappa_2=1e-2, bound_relax=1e-8, s_max=100.0, kappa_sigma=1e10,
    gamma_theta=1e-5, gamma_phi=1e-8, delta=1.0, s_theta=1.1, s_phi=2.3, eta_phi=1e-8, gamma_alpha=0.05,
    delta_w_min=1e-20, delta_w_0=1e-4, delta_w_max=1e40, kappa_w_minus=1.0 / 3.0, kappa_w_plus=8.0,
    kappa_w_plus_bar=100.0, scaling_max_gradient=100.0, fixed_iters=0,
)

STATUS_CONVERGED = 1
STATUS_MAXITER = 0
STATUS_NAN = -6
STATUS_NOPROGRESS = -7

_EPS10 = 10.0 * np.finfo(np.float64).eps
THETA_FLOOR = 1e-10     # constraint violation below this is round-off (see oracle/mpc_oracle.c)


def _cmp_le(lhs, rhs, base):
    """IPOPT's Compare_le (IpUtils.cpp): lhs <= rhs up to 10 eps of a reference magnitude"""
    return lhs - rhs <= _EPS10 * abs(base)


def _push(x, lo, hi, k1, k2):
    """IPOPT section 3.6: project the start point into the interior of its bounds."""
    x = x.copy()
    both = np.isfinite(lo) & np.isfinite(hi)
    only_lo = np.isfinite(lo) & ~np.isfinite(hi)
    only_hi = ~np.isfinite(lo) & np.isfinite(hi)
    pl = np.where(both, np.minimum(k1 * np.maximum(1, np.abs(lo)), k2 * (hi - lo)), k1 * np.maximum(1, np.abs(lo)))
    pu = np.where(both, np.minimum(k1 * np.maximum(1, np.abs(hi)), k2 * (hi - lo)), k1 * np.maximum(1, np.abs(hi)))
    m = both | only_lo
    x[m] = np.maximum(x[m], (lo + pl)[m])
    m = both | only_hi
    x[m] = np.minimum(x[m], (hi - pu)[m])
    return x


class DenseIPM:
    def __init__(self, nlp, **opts):
        self.nlp = nlp
        self.o = dict(OPTS)
        self.o.update(opts)

    def solve(self, x0, p, lbg=None, ubg=None, lbx=None, ubx=None, trace=None):
        o, nlp = self.o, self.nlp
        b = nlp.bounds()
        lbg = b[0] if lbg is None else np.asarray(lbg, float)
        ubg = b[1] if ubg is None else np.asarray(ubg, float)
        lbx = b[2] if lbx is None else np.asarray(lbx, float)
        ubx = b[3] if ubx is None else np.asarray(ubx, float)
        eq = np.where(lbg == ubg)[0]
        iq = np.where(lbg != ubg)[0]
        n, me, mi = nlp.n_w, len(eq), len(iq)
        rel = o["bound_relax"]

        def relax(lo, hi):
            lo = np.where(np.isfinite(lo), lo - rel * np.maximum(1, np.abs(lo)), lo)
            hi = np.where(np.isfinite(hi), hi + rel * np.maximum(1, np.abs(hi)), hi)
            return lo, hi
        xl, xu = relax(lbx, ubx)
        sl, su = relax(lbg[iq], ubg[iq])
        # variable vector v = (w, s); bounds
        vl = np.concatenate([xl, sl])
        vu = np.concatenate([xu, su])
        hasl, hasu = np.isfinite(vl), np.isfinite(vu)
        nb = int(hasl.sum() + hasu.sum())

        w = np.asarray(x0, float).ravel().copy()
        p = np.asarray(p, float).ravel()
        # objective scaling at the user's start point
        g0 = nlp.grad(w, p)
        gmax = np.abs(g0).max()
        df = 1.0
        if gmax > o["scaling_max_gradient"]:
            df = o["scaling_max_gradient"] / gmax
        w = _push(w, xl, xu, o["kappa_1"], o["kappa_2"])
        gv = nlp.g(w, p)
        s = _push(gv[iq], sl, su, o["kappa_1"], o["kappa_2"])
        v = np.concatenate([w, s])
        zl = np.where(hasl, 1.0, 0.0)
        zu = np.where(hasu, 1.0, 0.0)
        lam = np.zeros(me)          # multipliers of equality rows (g_eq - lbg)
        nu = np.zeros(mi)           # multipliers of (g_iq - s)
        mu = o["mu_init"]
        tau = max(o["tau_min"], 1 - mu)
        filt = []
        delta_last = 0.0
        status = STATUS_MAXITER
        theta0 = None
        it = 0

        def evaluate(v):
            w_, s_ = v[:n], v[n:]
            gv_ = nlp.g(w_, p)
            c_ = gv_[eq] - lbg[eq]
            d_ = gv_[iq] - s_
            return gv_, c_, d_

        def barrier_obj(v, fval, mu_):
            gl = v[hasl] - vl[hasl]
            gu = vu[hasu] - v[hasu]
            if (gl <= 0).any() or (gu <= 0).any():
                return np.inf
            return df * fval - mu_ * (np.log(gl).sum() + np.log(gu).sum())

        while True:
            w, s = v[:n], v[n:]
            fval = nlp.f(w, p)
            grad = df * nlp.grad(w, p)
            J = nlp.jac(w, p)
            Je, Ji = J[eq], J[iq]
            gv, c, d = evaluate(v)
            # full-space multipliers for the Hessian
            lam_full = np.zeros(nlp.n_g)
            lam_full[eq] = lam
            lam_full[iq] = nu
            # residuals
            rw = grad + Je.T @ lam + Ji.T @ nu - zl[:n] + zu[:n]
            rs = -nu - zl[n:] + zu[n:]
            gapl = np.where(hasl, v - vl, 1.0)
            gapu = np.where(hasu, vu - v, 1.0)
            compl_l = (gapl * zl)[hasl]
            compl_u = (gapu * zu)[hasu]
            allc = np.concatenate([compl_l, compl_u])
            zsum = zl[hasl].sum() + zu[hasu].sum()
            s_d = max(o["s_max"], (np.abs(lam).sum() + np.abs(nu).sum() + zsum) / max(1, me + mi + nb)) / o["s_max"]
            s_c = max(o["s_max"], zsum / max(1, nb)) / o["s_max"]
            dual_inf = max(np.abs(rw).max(), np.abs(rs).max() if mi else 0.0)
            prim_inf = max(np.abs(c).max() if me else 0.0, np.abs(d).max() if mi else 0.0)

            def E(mu_):
                cm = max(allc.max() - mu_, mu_ - allc.min()) if nb else 0.0
                return max(dual_inf / s_d, prim_inf, cm / s_c)
            if trace is not None:
                trace.append(dict(it=it, mu=mu, f=fval, dual_inf=dual_inf, prim_inf=prim_inf, E0=E(0.0), w=w.copy()))
            if not np.isfinite(fval) or not np.isfinite(rw).all():
                status = STATUS_NAN
                break
            if o["fixed_iters"] == 0 and E(0.0) <= o["tol"]:
                status = STATUS_CONVERGED
                break
            if it >= (o["fixed_iters"] or o["max_iter"]):
                status = STATUS_MAXITER if o["fixed_iters"] == 0 else STATUS_CONVERGED
                break
            # barrier update
            changed = False
            while E(mu) <= o["kappa_eps"] * mu:
                new_mu = max(o["tol"] / 10.0, min(o["kappa_mu"] * mu, mu ** o["theta_mu"]))
                if new_mu == mu:
                    break
                mu = new_mu
                tau = max(o["tau_min"], 1 - mu)
                changed = True
            if changed:
                filt = []
            # condensed KKT system
            sig = np.where(hasl, zl / gapl, 0.0) + np.where(hasu, zu / gapu, 0.0)
            gbar = np.where(hasl, -mu / gapl, 0.0) + np.where(hasu, mu / gapu, 0.0)   # barrier gradient
            H = nlp.hess_lag(w, p, df, lam_full)
            sig_w, sig_s = sig[:n], sig[n:]
            gb_w, gb_s = gbar[:n], gbar[n:]
            Hc = H + np.diag(sig_w) + Ji.T @ (sig_s[:, None] * Ji)
            rhs_w = -(grad + gb_w + Je.T @ lam + Ji.T @ (gb_s + sig_s * d))
            rhs_c = -c
            delta = 0.0
            ok = False
            trial = 0
            while True:
                K = np.block([[Hc + delta * np.eye(n), Je.T], [Je, np.zeros((me, me))]])
                ev = np.linalg.eigvalsh(K)
                npos, nneg = int((ev > 0).sum()), int((ev < 0).sum())
                if npos == n and nneg == me:
                    ok = True
                    break
                if delta == 0.0:
                    delta = o["delta_w_0"] if delta_last == 0.0 else max(o["delta_w_min"], o["kappa_w_minus"] * delta_last)
                else:
                    delta *= o["kappa_w_plus_bar"] if delta_last == 0.0 else o["kappa_w_plus"]
                trial += 1
                if delta > o["delta_w_max"]:
                    break
            if not ok:
                status = STATUS_NOPROGRESS
                break
            if delta > 0:
                delta_last = delta
            sol = np.linalg.solve(K, np.concatenate([rhs_w, rhs_c]))
            dw, dlam = sol[:n], sol[n:]
            ds = Ji @ dw + d
            dnu = gb_s - nu + sig_s * ds
            dv = np.concatenate([dw, ds])
            dzl = np.where(hasl, mu / gapl - zl - (zl / gapl) * dv, 0.0)
            dzu = np.where(hasu, mu / gapu - zu + (zu / gapu) * dv, 0.0)
            # fraction to the boundary

            def ftb(val, dval, mask):
                m = mask & (dval < 0)
                if not m.any():
                    return 1.0
                return min(1.0, float((-tau * val[m] / dval[m]).min()))
            a_max = min(ftb(gapl, dv, hasl), ftb(gapu, -dv, hasu))
            a_z = min(ftb(zl, dzl, hasl), ftb(zu, dzu, hasu))
            # filter line search
            theta = np.abs(c).sum() + np.abs(d).sum()
            if theta0 is None:
                theta0 = theta
                theta_max = 1e4 * max(1.0, theta0)
                theta_min = 1e-4 * max(1.0, theta0)
            phi = barrier_obj(v, fval, mu)
            dphi = float((grad + gb_w) @ dw + gb_s @ ds)
            if dphi < 0 and theta <= theta_min:
                a_min = min(o["gamma_theta"], o["gamma_phi"] * theta / (-dphi),
                            o["delta"] * theta ** o["s_theta"] / (-dphi) ** o["s_phi"])
            elif dphi < 0:
                a_min = min(o["gamma_theta"], o["gamma_phi"] * theta / (-dphi))
            else:
                a_min = o["gamma_theta"]
            a_min *= o["gamma_alpha"]
            alpha = a_max
            accepted = False
            ftype = False
            nls = 0
            while alpha >= a_min:
                vt = v + alpha * dv
                _, ct, dt_ = evaluate(vt)
                th_t = np.abs(ct).sum() + np.abs(dt_).sum()
                ph_t = barrier_obj(vt, nlp.f(vt[:n], p), mu)
                nls += 1
                good = np.isfinite(th_t) and np.isfinite(ph_t) and th_t <= theta_max
                if good:
                    for (tf, pf) in filt:
                        if not (_cmp_le(max(th_t, THETA_FLOOR), max(tf, THETA_FLOOR), tf) or _cmp_le(ph_t, pf, pf)):
                            good = False
                            break
                if good:
                    switching = (theta <= theta_min and dphi < 0 and
                                 alpha * (-dphi) ** o["s_phi"] > o["delta"] * theta ** o["s_theta"])
                    if switching:
                        if _cmp_le(ph_t - phi, o["eta_phi"] * alpha * dphi, phi):
                            accepted, ftype = True, True
                    else:
                        if (_cmp_le(max(th_t, THETA_FLOOR), max((1 - o["gamma_theta"]) * theta, THETA_FLOOR), theta)
                                or _cmp_le(ph_t - phi, -o["gamma_phi"] * theta, phi)):
                            accepted = True
                if accepted:
                    break
                alpha *= 0.5
            if not accepted:
                # fallback replacing IPOPT's restoration phase: largest tried step that lowers the
                # primal-dual error; otherwise give up
                status = STATUS_NOPROGRESS
                break
            if not ftype:
                filt.append(((1 - o["gamma_theta"]) * theta, phi - o["gamma_phi"] * theta))
            v = v + alpha * dv
            lam = lam + alpha * dlam
            nu = nu + alpha * dnu
            zl = zl + a_z * dzl
            zu = zu + a_z * dzu
            gapl = np.where(hasl, v - vl, 1.0)
            gapu = np.where(hasu, vu - v, 1.0)
            ks = o["kappa_sigma"]
            zl = np.where(hasl, np.clip(zl, mu / (ks * gapl), ks * mu / gapl), 0.0)
            zu = np.where(hasu, np.clip(zu, mu / (ks * gapu), ks * mu / gapu), 0.0)
            it += 1
            if trace is not None:
                trace[-1].update(alpha=alpha, a_z=a_z, delta=delta, nls=nls, ftype=ftype, theta=theta, phi=phi)
        return dict(x=v[:n].copy(), status=status, iters=it, f=nlp.f(v[:n], p), lam=lam, nu=nu, mu=mu,
                    kkt=E(0.0) if np.isfinite(fval) else np.nan, s=v[n:].copy(), df=df)
