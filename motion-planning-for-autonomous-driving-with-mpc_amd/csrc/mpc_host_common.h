// mpc_host_common.h -- host-side problem bookkeeping shared by the HIP library (mpcgpu.hip) and the CPU
// emulation harness used by tests (tests/emu/emu.cpp): descriptor validation, parsing of the reference's
// lbx/ubx/lbg/ubg lists (optimizer.py:413-491), IPOPT bound relaxation, workspace layout.
#pragma once
#include <cmath>
#include <cstddef>
#include <cstdint>
#include <cstdio>
#include <string>
#include <vector>

#include "../../include/mpcgpu.h"
#include "mpc_stage_math.h"

namespace mpc {

constexpr double BOUND_RELAX = 1e-8;   // IPOPT bound_relax_factor

struct HostProblem {
    mpc_problem_desc desc{};
    std::vector<double> LB, UB;         // [(N+1)*NZ] relaxed, +-inf = absent
    int has_fl = 0, has_fu = 0, has_ol = 0, has_ou = 0;
    double fl = 0, fu = 0, ol = 0, ou = 0;
    double ol_raw = 0;                  // lower bound of the obstacle rows as the caller gave it (before relaxation)
    double fl_raw = 0;                  // lbg[0] as the caller gave it
    // option friction_lb.  0 ("nlp", the default): the lower bound lbg[0] <= 0 of the stage-0 friction row |a_0^2 + c| is implied by
    // the absolute value and gets no barrier (and the row is presolved into a bound on a_0 wherever that is exact).  1 ("ipopt"): the
    // row as IPOPT sees it (optimizer.py:378, 424-425) -- a slack with BOTH bounds, log barrier on the lower one too: the kink of |.|
    // at a_0^2 = -c becomes a wall the slack cannot cross, and a solve can end at it (tests/test_recorded_residuals.py: what the
    // recorded CasADi runs show at ZAM_Over-1_1 steps 4 and 13)
    int fric_literal = 0;
    int n_mult = 0, n_z = 0;
    bool bounds_set = false;
    int NZ() const { return desc.nx + 2; }
    size_t n_w() const { return (size_t)2 * desc.N + (size_t)desc.nx * (desc.N + 1); }
    size_t n_g() const { return (size_t)1 + (size_t)desc.nx * (desc.N + 1) + (size_t)9 * (desc.N + 1); }
};

// (option friction_lb changed after the bounds were set)
inline void apply_fric_literal(HostProblem& hp, int literal) {
    hp.fric_literal = literal ? 1 : 0;
    if (hp.bounds_set) hp.has_fl = std::isfinite(hp.fl_raw) && (hp.fl_raw > 0.0 || hp.fric_literal);
}
inline double relax_lo(double lo) { return std::isfinite(lo) ? lo - BOUND_RELAX * std::fmax(1.0, std::fabs(lo)) : -INFINITY; }
inline double relax_hi(double hi) { return std::isfinite(hi) ? hi + BOUND_RELAX * std::fmax(1.0, std::fabs(hi)) : INFINITY; }

inline int validate_desc(const mpc_problem_desc& d, std::string& err) {
    char buf[256];
    if (d.N < 1 || d.N > 127) { snprintf(buf, sizeof buf, "N=%d outside [1,127]", d.N); err = buf; return MPC_ERR_INVALID; }
    if (d.nx != 5 && d.nx != 6) { snprintf(buf, sizeof buf, "nx=%d (supported: 5, 6)", d.nx); err = buf; return MPC_ERR_INVALID; }
    if (d.nu != 2) { err = "nu must be 2"; return MPC_ERR_INVALID; }
    if (d.formulation != MPC_FORM_CASADI_EULER) { err = "unsupported formulation"; return MPC_ERR_INVALID; }
    if (!(d.dt > 0) || !(d.wheelbase > 0) || !(d.friction_div != 0) || !(d.tol > 0)) { err = "dt, wheelbase, tol must be > 0"; return MPC_ERR_INVALID; }
    if (d.obst_mult < 1 || d.max_iter < 1 || d.fixed_iters < 0) { err = "obst_mult>=1, max_iter>=1, fixed_iters>=0 required"; return MPC_ERR_INVALID; }
    if (d.max_iter > 1024 || d.fixed_iters > 1024) { err = "max_iter / fixed_iters above 1024 (the convergence-poll table)"; return MPC_ERR_INVALID; }
    return MPC_OK;
}

// reference defaults of inequal_constraints() for the vehicle-2 limits (optimizer.py:37-46, 421-491)
inline void default_bounds(const mpc_problem_desc& d, std::vector<double>& lbx, std::vector<double>& ubx,
                           std::vector<double>& lbg, std::vector<double>& ubg) {
    const int N = d.N, nx = d.nx;
    lbx.clear(); ubx.clear(); lbg.clear(); ubg.clear();
    for (int k = 0; k < N; ++k) { lbx.push_back(-0.4); ubx.push_back(0.4); lbx.push_back(-INFINITY); ubx.push_back(11.5); }
    for (int k = 0; k <= N; ++k) {
        const double lo[6] = {-INFINITY, -INFINITY, -1.066, 0.0, -INFINITY, -INFINITY};
        const double hi[6] = {INFINITY, INFINITY, 1.066, 50.8, INFINITY, INFINITY};
        for (int i = 0; i < nx; ++i) { lbx.push_back(lo[i]); ubx.push_back(hi[i]); }
    }
    lbg.push_back(0.0); ubg.push_back(11.5);
    for (int i = 0; i < nx * (N + 1); ++i) { lbg.push_back(0.0); ubg.push_back(0.0); }
    // r_ego + r_obstacle with a zero-size dummy obstacle: 1.2000000000000002 (configuration.py:40-66)
    for (int i = 0; i < 9 * (N + 1); ++i) { lbg.push_back(1.2000000000000002); ubg.push_back(INFINITY); }
}

inline int set_bounds(HostProblem& hp, const double* lbx, const double* ubx, const double* lbg, const double* ubg, std::string& err) {
    const mpc_problem_desc& d = hp.desc;
    const int N = d.N, nx = d.nx, NZ = nx + 2;
    std::vector<double> dlbx, dubx, dlbg, dubg;
    if (!lbx && !ubx && !lbg && !ubg) {
        default_bounds(d, dlbx, dubx, dlbg, dubg);
        lbx = dlbx.data(); ubx = dubx.data(); lbg = dlbg.data(); ubg = dubg.data();
    } else if (!lbx || !ubx || !lbg || !ubg) {
        err = "lbx, ubx, lbg, ubg must be all given or all NULL";
        return MPC_ERR_INVALID;
    }
    char buf[256];
    hp.LB.assign((size_t)(N + 1) * NZ, -INFINITY);
    hp.UB.assign((size_t)(N + 1) * NZ, INFINITY);
    int nb = 0;
    for (int k = 0; k <= N; ++k)
        for (int i = 0; i < NZ; ++i) {
            if (i < 2 && k == N) continue;
            const size_t src = (i < 2) ? (size_t)2 * k + i : (size_t)2 * N + (size_t)nx * k + (i - 2);
            const double lo = lbx[src], hi = ubx[src];
            if (std::isnan(lo) || std::isnan(hi) || lo > hi) { snprintf(buf, sizeof buf, "lbx/ubx[%zu]: invalid pair", src); err = buf; return MPC_ERR_BOUNDS; }
            if (lo == hi) { snprintf(buf, sizeof buf, "lbx == ubx at %zu: fixed variables are not supported", src); err = buf; return MPC_ERR_BOUNDS; }
            hp.LB[(size_t)k * NZ + i] = relax_lo(lo);
            hp.UB[(size_t)k * NZ + i] = relax_hi(hi);
            if (!(k == 0 && i == 1)) nb += (int)std::isfinite(lo) + (int)std::isfinite(hi);   // a_0: counted per instance
        }
    // g rows: [friction | nx(N+1) equalities | 9(N+1) obstacle rows]  (optimizer.py:421-469)
    const double flo = lbg[0], fhi = ubg[0];
    if (std::isnan(flo) || std::isnan(fhi) || !(flo < fhi)) { err = "friction row needs lbg[0] < ubg[0]"; return MPC_ERR_BOUNDS; }
    for (int i = 0; i < nx * (N + 1); ++i)
        if (!(lbg[1 + i] == 0.0 && ubg[1 + i] == 0.0)) { snprintf(buf, sizeof buf, "g row %d must be an equality with lbg = ubg = 0", 1 + i); err = buf; return MPC_ERR_BOUNDS; }
    const size_t o0 = (size_t)1 + (size_t)nx * (N + 1);
    const double olo = lbg[o0], ohi = ubg[o0];
    for (int i = 0; i < 9 * (N + 1); ++i)
        if (!(lbg[o0 + i] == olo && ubg[o0 + i] == ohi)) { err = "obstacle rows must share one [lbg, ubg] pair"; return MPC_ERR_BOUNDS; }
    if (std::isnan(olo) || std::isnan(ohi) || !(olo < ohi)) { err = "obstacle rows need lbg < ubg"; return MPC_ERR_BOUNDS; }
    hp.fl_raw = flo;
    hp.has_fl = std::isfinite(flo) && (flo > 0.0 || hp.fric_literal);     // |y| >= lo with lo <= 0 is implied by the absolute value (unless asked for literally)
    hp.has_fu = std::isfinite(fhi);
    hp.fl = relax_lo(flo); hp.fu = relax_hi(fhi);
    hp.has_ol = std::isfinite(olo); hp.has_ou = std::isfinite(ohi);
    hp.ol = relax_lo(olo); hp.ou = relax_hi(ohi);
    hp.ol_raw = olo;
    const int m = d.obst_mult;
    // per-instance parts (friction row or its presolved bound on a_0) are added in phase_finish()
    hp.n_mult = nx * (N + 1) + 3 * m * (N + 1);
    hp.n_z = nb + 3 * m * (N + 1) * (hp.has_ol + hp.has_ou);
    hp.bounds_set = true;
    return MPC_OK;
}

// ---- workspace: one allocation of doubles + one of int32, tile-major [tile of 64 instances][row][64 lanes] --------
// WsLayout members are ROW offsets inside a tile (all even: rows are stored in pairs, see mpc_prow in mpc_stage_math.h);
// element(row, b) = (b >> 6) * tile_elems + mpc_prow(row) + 2 * (b & 63).
struct WsLayout {
    size_t Z, ZL, ZU, SO, NUO, ZLO, ZUO, LAM, REF, DZ, PK, KK, BLK, ROLL, SC, FILT, OBST;
    size_t MZ, MZL, MZU, MSO, MNUO, MZLO, MZUO, MLAM, MREF;
    size_t MBLK, MPK, MDZ;       // ELEMENT offsets of the instance-major mailbox arrays ([instance][stage][rows], behind the tiles)
    size_t rows, irows;          // rows per tile (double / int32 workspace)
    size_t tile_elems, itile_elems, ntiles;
    size_t total, itotal;        // elements to allocate
    size_t elem(size_t row0, size_t row, size_t b) const { return (b >> 6) * tile_elems + mpc_prow((uint32_t)(row0 + row)) + (b & 63) * 2; }
    size_t ielem(size_t row, size_t b) const { return (b >> 6) * itile_elems + mpc_prow((uint32_t)row) + (b & 63) * 2; }
};

// mailbox: with the instance-major copies k_solve_wg works on (about as large as the tile-major section: a handle that can never run
// that kernel -- long horizons, a fixed iteration count, hybrid switched off -- does without them and fits twice the rows below 4 GiB)
inline WsLayout ws_layout(int N, int nx, size_t Bp, bool mailbox = true) {
    const size_t NS = (size_t)nx * (nx + 1) / 2, S = N + 1;
    auto ev = [](size_t r) { return (r + 1) & ~(size_t)1; };                          // MPC_EV: whole row pairs per stage
    const size_t NZ = ev(nx + 2), XS = ev(nx), NBLK = ev((nx + 5) + 10 + 2 * nx), NPK = ev(NS + nx), NKK = 2 * nx + 2;   // Dim<NX>::NBLK: sparse H (NX + 5 entries)
    WsLayout w{};
    size_t off = 0;
    auto take = [&](size_t rows) { const size_t o = off; off += ev(rows); return o; };
    w.Z = take(S * NZ); w.ZL = take(S * NZ); w.ZU = take(S * NZ);
    w.SO = take(S * 4); w.NUO = take(S * 4); w.ZLO = take(S * 4); w.ZUO = take(S * 4);
    w.LAM = take(S * XS); w.REF = take(S * XS); w.DZ = take(S * NZ);
    w.PK = take(S * NPK); w.KK = take((size_t)N * NKK); w.BLK = take(S * NBLK); w.ROLL = take(S * XS);
    w.SC = take(SC_COUNT); w.FILT = take(2 * FILTER_MAX); w.OBST = take(6);
    w.rows = off;
    w.irows = ev(IS_COUNT);
    w.tile_elems = w.rows * 64;
    w.itile_elems = w.irows * 64;
    w.ntiles = Bp / 64;
    w.total = w.ntiles * w.tile_elems;
    const size_t mb = mailbox ? Bp * S : 0;
    w.MBLK = w.total; w.total += mb * NBLK;
    w.MPK = w.total; w.total += mb * NPK;
    w.MDZ = w.total; w.total += mb * NZ;
    w.MZ = w.total; w.total += mb * NZ;
    w.MZL = w.total; w.total += mb * NZ;
    w.MZU = w.total; w.total += mb * NZ;
    w.MSO = w.total; w.total += mb * 4;
    w.MNUO = w.total; w.total += mb * 4;
    w.MZLO = w.total; w.total += mb * 4;
    w.MZUO = w.total; w.total += mb * 4;
    w.MLAM = w.total; w.total += mb * XS;
    w.MREF = w.total; w.total += mb * XS;
    w.itotal = w.ntiles * w.itile_elems;
    return w;
}

inline int pick_bx(int N, int max_threads) {
    int bx = 64;
    while (bx > 4 && bx * (N + 1) > max_threads) bx >>= 1;      // N <= 127 (validate_desc): 4 * 128 = 512 threads at most
    return bx;
}

inline void fill_params(Params& P, const HostProblem& hp, int B, size_t Bp, int bx, double* base, int32_t* ibase,
                        const double* dLB, const double* dUB, bool mailbox = true) {
    const mpc_problem_desc& d = hp.desc;
    const WsLayout w = ws_layout(d.N, d.nx, Bp, mailbox);
    P.B = B; P.Bp = (int32_t)Bp; P.N = d.N; P.nx = d.nx; P.bx = bx;
    P.obst_mult = d.obst_mult; P.max_iter = d.max_iter; P.fixed_iters = d.fixed_iters;
    P.has_fl = hp.has_fl; P.has_fu = hp.has_fu; P.has_ol = hp.has_ol; P.has_ou = hp.has_ou; P.per_inst_obst = 0;
    P.dt = d.dt; P.wheelbase = d.wheelbase; P.friction_div = d.friction_div; P.ego_offset = d.ego_offset; P.tol = d.tol;
    for (int i = 0; i < 6; ++i) P.Q[i] = (i < d.nx) ? d.Q[i] : 0.0;
    P.R[0] = d.R[0]; P.R[1] = d.R[1];
    for (int i = 0; i < 6; ++i) P.obst[i] = d.obstacle[i];
    P.fl = hp.fl; P.fu = hp.fu; P.ol = hp.ol; P.ou = hp.ou;
    P.inv_S = (uint32_t)((0x100000000ull + (uint64_t)d.N) / (uint64_t)(d.N + 1));
    P.run_counter = nullptr;
    P.dec_s = 0;                      // set after the masks below
    P.tile_mask = nullptr;
    P.lo_mask = P.hi_mask = 0;
    P.dense_mask = 0xFFFFu;
    for (size_t q = 0; q < hp.LB.size(); ++q) {
        const int i = (int)(q % (size_t)hp.NZ()), k = (int)(q / (size_t)hp.NZ());
        if (hp.LB[q] > -1e300) P.lo_mask |= 1u << i;
        if (hp.UB[q] < 1e300) P.hi_mask |= 1u << i;
        if (!(i < 2 && k == d.N)) {                                  // (the inputs of the terminal stage do not exist)
            if (!(hp.LB[q] > -1e300)) P.dense_mask &= ~(1u << i);
            if (!(hp.UB[q] < 1e300)) P.dense_mask &= ~(0x100u << i);
        }
    }
    P.dec_s = (d.nx == 6 && d.Q[5] == 0.0 && !((P.lo_mask | P.hi_mask) & (1u << 7))) ? 1 : 0;
    P.x0 = nullptr; P.p = nullptr; P.LB = dLB; P.UB = dUB;
    // every array pointer addresses its first row inside tile 0
    P.Z = base + w.Z * 64; P.ZL = base + w.ZL * 64; P.ZU = base + w.ZU * 64;
    P.SO = base + w.SO * 64; P.NUO = base + w.NUO * 64; P.ZLO = base + w.ZLO * 64; P.ZUO = base + w.ZUO * 64;
    P.LAM = base + w.LAM * 64; P.REF = base + w.REF * 64; P.DZ = base + w.DZ * 64; P.PK = base + w.PK * 64; P.KK = base + w.KK * 64;
    P.BLK = base + w.BLK * 64; P.ROLL = base + w.ROLL * 64; P.SC = base + w.SC * 64;
    P.FILT = base + w.FILT * 64; P.OBST = base + w.OBST * 64;
    P.MBLK = base + w.MBLK; P.MPK = base + w.MPK; P.MDZ = base + w.MDZ;
    P.MZ = base + w.MZ; P.MZL = base + w.MZL; P.MZU = base + w.MZU; P.MSO = base + w.MSO; P.MNUO = base + w.MNUO; P.MZLO = base + w.MZLO;
    P.MZUO = base + w.MZUO; P.MLAM = base + w.MLAM; P.MREF = base + w.MREF;
    P.tile_elems = (uint32_t)w.tile_elems; P.itile_elems = (uint32_t)w.itile_elems;
    P.ISC = ibase;
    P.WS = base; P.IWS = ibase;
    P.ws_bytes = (uint32_t)(w.total * sizeof(double)); P.iws_bytes = (uint32_t)(w.itotal * sizeof(int32_t));
    P.x_out = nullptr; P.status_out = nullptr; P.iters_out = nullptr; P.kkt_out = nullptr;
    P.emit = 0; P.fail_count = nullptr; P.fin_ctl = nullptr; P.fin_host = nullptr;
    P.DBG = nullptr;
    P.tile0 = 0;
}

inline void default_desc(mpc_problem_desc* d, int32_t N, int32_t nx) {
    *d = mpc_problem_desc{};
    d->N = N; d->nx = nx; d->nu = 2; d->formulation = MPC_FORM_CASADI_EULER;
    d->max_iter = 100; d->fixed_iters = 0; d->obst_mult = 3; d->device = 0;
    d->dt = 0.1; d->wheelbase = 2.5789128; d->friction_div = 2.578; d->ego_offset = 0.75; d->tol = 1e-8;
    const double Q[5] = {2.3, 2.3, 500.0, 0.1, 10.0};          // config_LF_ZAM_Over-1_1.yaml:20-24
    for (int i = 0; i < 5; ++i) d->Q[i] = Q[i];
    d->R[0] = 2.0; d->R[1] = 0.2;                               // config_LF_ZAM_Over-1_1.yaml:25-26
    const double Pt[5] = {80.0, 80.0, 100.0, 0.1, 100.0};       // config_LF_ZAM_Over-1_1.yaml:27-31
    for (int i = 0; i < 5; ++i) d->P[i] = Pt[i];
    for (int j = 0; j < 3; ++j) { d->obstacle[2 * j] = -100.0; d->obstacle[2 * j + 1] = 0.0; }   // configuration.py:471-483
}

}  // namespace mpc
