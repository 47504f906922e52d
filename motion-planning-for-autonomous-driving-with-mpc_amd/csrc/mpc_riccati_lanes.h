// mpc_riccati_lanes.h -- the Riccati factor/solve of the RESIDENT solve path (k_resident in mpcgpu.hip): the recursion of one instance
// spread over the lanes of a wavefront, all of its data in LDS (product code; the same source is stepped lane by lane by the CPU
// emulation harness, tests/emu).
//
// Why a second formulation.  riccati_tile (mpcgpu.hip) runs one instance per lane: 64 instances per wavefront, a 31-stage chain of
// ~2 100 ticks per stage whatever the number of live lanes.  In the resident path a workgroup owns `bx` (8) instances for the whole
// solve and keeps their stage blocks in LDS; here a wavefront gives every instance a group of 8 lanes and lane r of the group owns ROW r
// of the augmented cost-to-go [P_k | p_k] (NX + 1 numbers in registers).  One backward stage is then
//     phase A   (own row only)   Y[r][:] = P+[r][:] (I + dtF),  h[r] = p+[r] - P+[r][:] c        -> exchange area (LDS)
//     phase B   (all rows)       P_k[r][:] = H[r][:] + Y[r][:] + sum_a coef_a[r] Y[a][:]          <- exchange area
// with coef = ((dtF)' entries of row r | dt K[:,r]): the products (dtF)'Y and G'K of the scalar recursion written as ONE combination of the
// rows of Y, and K[:,r] = -Lam^-1 G[:,r] computed by every lane for its own column.  ~150 instructions per stage for 8 instances, one
// LDS round trip, against ~370 instructions and 38 memory operations of the one-instance-per-lane step.
//
// The sweep only produces the SEQUENCE P_N ... P_0 (upper triangle + p_k per stage, into the stage's LDS slot).  Gains are not stored by
// it: once the sweep has succeeded, the stage threads (one per (instance, stage), all stages at once) recompute G, Lam^-1, K, k_ff of
// their stage from P_{k+1} with the expressions of ric_matrix_step / ric_vector_step (lane_gain), write the closed-loop rows the forward
// sweep needs over the consumed part of their slot, and the forward sweep is one LDS round trip per stage (lane_forward_step).
//
// Reference: the linear solve inside IPOPT's step computation for the NLP of MPC_Planner/optimizer.py:373-558 (the reference hands it
// to MUMPS); recursion and inertia-correction schedule as in riccati_instance (mpc_stage_math.h).
#pragma once
#include "mpc_stage_math.h"

namespace mpc {

// exchange area of the backward sweep, per (instance group, row): Y row (NX), h, P+[r][2], P+[r][3]
template <int NX> struct Xch { static constexpr int ROW = (int)MPC_EV(NX + 3); static constexpr int GROUP = 8 * ROW; };

template <int NX>
struct LaneRic {
    int g, r;                 // instance column of the workgroup, row of the cost-to-go
    bool row;                 // r < NX and the instance exists
    bool need;                // the instance still needs a (or another) backward sweep
    bool ok;                  // every Lam of the sweep so far was positive definite
    bool sym;                 // symmetrise P_k every stage (instances that ever needed an inertia correction, see ric_matrix_step)
    bool failed;
    double delta, delta_last, hux0, hux1;
    double Pr[NX], pr;        // row r of P+ and p+[r]
    int hoff[NX];             // slot offset of H[r][j] (or Slot::ZERO)
    // carried from phase A to phase B of a stage
    double Y[NX], h;
};

template <int NX>
MPC_HD void lane_setup(LaneRic<NX>& L, int lane, int bx, bool inst_active, mpc_lds_cptr inst) {
    using D = Dim<NX>;
    using S = Slot<NX>;
    L.g = lane >> 3;
    L.r = lane & 7;
    L.row = L.r < NX && L.g < bx;
    L.need = L.g < bx && inst_active;
    L.ok = true;
    L.failed = false;
    L.delta = 0.0;
    L.delta_last = L.g < bx ? inst[IR_DLAST] : 0.0;
    L.hux0 = L.g < bx ? inst[IR_HUX0] : 0.0;
    L.hux1 = L.g < bx ? inst[IR_HUX1] : 0.0;
    // Every lane computes its whole row, so P_k[r][j] and P_k[j][r] come from different expressions and differ in the last bits; with
    // active circle rows the cost-to-go carries weights of 1/mu and an unsymmetric P_k fed back into the recursion costs iterations
    // (collision avoidance, 512 instances: mean 27.0 / slowest 97 without the symmetrisation round trip, 25.0 / 68 with it; the
    // streaming path, symmetric by construction, needs 25.8 / 75).  Always on; the flag stays so that a cheaper rule can be tried.
    L.sym = true;
#pragma unroll
    for (int j = 0; j < NX; ++j) {
        const int i0 = L.r < j ? L.r : j, i1 = L.r < j ? j : L.r;
        int hr = -1;
        // (Dim::hrow with run-time arguments; rows beyond NX point at the zero pad)
        if (i1 < NX) hr = i0 == i1 ? i0 : (i0 == 0 && i1 == 1) ? NX : (i0 == 0 && i1 == 4) ? NX + 1 : (i0 == 1 && i1 == 4) ? NX + 2
                        : (i0 == 2 && i1 == 3) ? NX + 3 : (i0 == 3 && i1 == 4) ? NX + 4 : -1;
        L.hoff[j] = hr >= 0 ? S::H + hr : S::ZERO;
    }
#pragma unroll
    for (int j = 0; j < NX; ++j) { L.Pr[j] = 0.0; L.Y[j] = 0.0; }
    L.pr = 0.0;
    L.h = 0.0;
    (void)sizeof(D);
}

// start of a sweep: delta_w of this attempt is in L.delta; terminal stage P_N = H_N + delta I, p_N = g_N
template <int NX>
MPC_HD void lane_terminal(LaneRic<NX>& L, mpc_lds_ptr slotN) {
    using D = Dim<NX>;
    using S = Slot<NX>;
    L.ok = true;
    if (!L.row) return;
#pragma unroll
    for (int j = 0; j < NX; ++j) L.Pr[j] = slotN[L.hoff[j]] + ((j == L.r) ? L.delta : 0.0);
    L.pr = slotN[S::GX + L.r];
    if (L.need) {
#pragma unroll
        for (int j = 0; j < NX; ++j)
            if (j >= L.r) slotN[S::PK + L.r * NX - L.r * (L.r - 1) / 2 + (j - L.r)] = L.Pr[j];
        slotN[S::PK + D::NS + L.r] = L.pr;
    }
}

// phase A of backward stage k: own row only.  `slot` = record of (instance g, stage k), `xg` = exchange area of group g
template <int NX>
MPC_HD void lane_bwd_A(const Params& P, LaneRic<NX>& L, mpc_lds_cptr slot, mpc_lds_ptr xg) {
    using S = Slot<NX>;
    using X = Xch<NX>;
    if (!L.row) return;
    const double dt = P.dt;
    const double a03 = slot[S::A + 0], a04 = slot[S::A + 1], a13 = slot[S::A + 2], a14 = slot[S::A + 3], a42 = slot[S::A + 4], a43 = slot[S::A + 5];
    // W = P+ (dtF), columns delta, v, psi (ric_matrix_step: same expressions, row r only)
    const double w0 = L.Pr[4] * a42;
    double w1 = L.Pr[0] * a03 + L.Pr[1] * a13 + L.Pr[4] * a43;
    if (NX == 6) w1 += L.Pr[NX - 1] * dt;
    const double w2 = L.Pr[0] * a04 + L.Pr[1] * a14;
#pragma unroll
    for (int j = 0; j < NX; ++j) L.Y[j] = L.Pr[j];
    L.Y[2] += w0; L.Y[3] += w1; L.Y[4] += w2;
    // h = p+ - P+ c_{k+1} (ric_vector_step)
    double t = L.pr;
#pragma unroll
    for (int j = 0; j < NX; ++j) t -= L.Pr[j] * slot[S::CN + j];
    L.h = t;
    mpc_lds_ptr x = xg + L.r * X::ROW;
#pragma unroll
    for (int j = 0; j < NX; ++j) x[j] = L.Y[j];
    x[NX] = L.h;
    x[NX + 1] = L.Pr[2];
    x[NX + 2] = L.Pr[3];
}

// phase B of backward stage k: P_k row r from all rows of Y; writes the upper-triangle part of the row and p_k[r] into the slot
template <int NX>
MPC_HD void lane_bwd_B(const Params& P, LaneRic<NX>& L, int k, mpc_lds_ptr slot, mpc_lds_cptr xg) {
    using D = Dim<NX>;
    using S = Slot<NX>;
    using X = Xch<NX>;
    if (!L.row) return;
    const double dt = P.dt;
    const int r = L.r;
    const double a03 = slot[S::A + 0], a04 = slot[S::A + 1], a13 = slot[S::A + 2], a14 = slot[S::A + 3], a42 = slot[S::A + 4], a43 = slot[S::A + 5];
    // Lam = Ruu + B'P+B (+ delta_w), its inverse: every lane of the group computes the same bits
    const double P22 = xg[2 * X::ROW + NX + 1], P23 = xg[2 * X::ROW + NX + 2], P33 = xg[3 * X::ROW + NX + 2];
    const double L00 = slot[S::RUU] + dt * dt * P22 + L.delta;
    const double L01 = dt * dt * P23;
    const double L11 = slot[S::RUU + 1] + dt * dt * P33 + L.delta;
    const double det = L00 * L11 - L01 * L01;
    if (!((L00 > 0.0) && (det > 0.0))) L.ok = false;
    const double idet = 1.0 / det;
    const double i00 = L11 * idet, i01 = -L01 * idet, i11 = L00 * idet;
    // G[:, r] = dt Y[2..3][r] (+ Hux at stage 0), K[:, r] = -Lam^-1 G[:, r]
    double g0 = dt * xg[2 * X::ROW + r], g1 = dt * xg[3 * X::ROW + r];
    if (k == 0) { if (r == 2) g1 += L.hux0; if (r == 3) g1 += L.hux1; }
    const double k0 = -(i00 * g0 + i01 * g1), k1 = -(i01 * g0 + i11 * g1);
    // coefficients of the rows of Y in row r of  (dtF)'Y + G'K:  rows 0, 1, 4, 5 from (dtF)' (column r of dtF), rows 2, 3 from dt K[:, r]
    const double e0 = (r == 3) ? a03 : (r == 4) ? a04 : 0.0;
    const double e1 = (r == 3) ? a13 : (r == 4) ? a14 : 0.0;
    const double e4 = (r == 2) ? a42 : (r == 3) ? a43 : 0.0;
    const double e5 = (NX == 6 && r == 3) ? dt : 0.0;
    const double c2 = dt * k0, c3 = dt * k1;
    double Pn[NX], pn;
#pragma unroll
    for (int j = 0; j <= NX; ++j) {
        double t = (j < NX) ? slot[L.hoff[j < NX ? j : 0]] : slot[S::GX + r];
        t += (j < NX) ? L.Y[j < NX ? j : 0] : L.h;
        t += e0 * xg[0 * X::ROW + j];
        t += e1 * xg[1 * X::ROW + j];
        t += c2 * xg[2 * X::ROW + j];
        t += c3 * xg[3 * X::ROW + j];
        t += e4 * xg[4 * X::ROW + j];
        if (NX == 6) t += e5 * xg[5 * X::ROW + j];
        if (j == r) t += L.delta;
        if (j < NX) Pn[j < NX ? j : 0] = t; else pn = t;
    }
    // what G carries besides dt Y[2..3]: Hux at stage 0 (columns delta, v of G1) and gu in the vector column
    if (k == 0) { Pn[2] += k1 * L.hux0; Pn[3] += k1 * L.hux1; }
    pn += k0 * slot[S::GU] + k1 * slot[S::GU + 1];
#pragma unroll
    for (int j = 0; j < NX; ++j) L.Pr[j] = Pn[j];
    L.pr = pn;
    (void)sizeof(D);
}

// symmetrisation round trip (only for wavefronts with an instance that asks for it): full row out, column in, mean of the two
template <int NX>
MPC_HD void lane_sym_out(const LaneRic<NX>& L, mpc_lds_ptr xg) {
    using X = Xch<NX>;
    if (!L.row) return;
#pragma unroll
    for (int j = 0; j < NX; ++j) xg[L.r * X::ROW + j] = L.Pr[j];
}
template <int NX>
MPC_HD void lane_sym_in(LaneRic<NX>& L, mpc_lds_cptr xg) {
    using X = Xch<NX>;
    if (!L.row || !L.sym) return;
#pragma unroll
    for (int j = 0; j < NX; ++j) L.Pr[j] = 0.5 * (L.Pr[j] + xg[j * X::ROW + L.r]);
}
// the stage's result into its slot (upper triangle of row r, p_k[r])
template <int NX>
MPC_HD void lane_bwd_store(const LaneRic<NX>& L, mpc_lds_ptr slot) {
    using D = Dim<NX>;
    using S = Slot<NX>;
    if (!(L.row && L.need && L.ok)) return;
#pragma unroll
    for (int j = 0; j < NX; ++j)
        if (j >= L.r) slot[S::PK + L.r * NX - L.r * (L.r - 1) / 2 + (j - L.r)] = L.Pr[j];
    slot[S::PK + D::NS + L.r] = L.pr;
}

// end of a sweep: IPOPT's inertia-correction schedule (Waechter & Biegler section 3.1), as riccati_instance
template <int NX>
MPC_HD void lane_sweep_decide(LaneRic<NX>& L) {
    if (!L.need) return;
    if (L.ok) { L.need = false; return; }
    if (L.delta == 0.0) L.delta = (L.delta_last == 0.0) ? DW_0 : fmax(DW_MIN, KW_MINUS * L.delta_last);
    else L.delta *= (L.delta_last == 0.0) ? KW_PLUS_BAR : KW_PLUS;
    L.sym = true;
    if (L.delta > DW_MAX) { L.need = false; L.failed = true; }
}
// IR_RSTAT: 1 = the instance is iterating (written by its k = 0 stage thread before the sweep), -7 = no admissible delta_w
template <int NX>
MPC_HD void lane_sweep_finish(const LaneRic<NX>& L, bool inst_active, mpc_lds_ptr inst) {
    if (L.r != 0 || !inst_active) return;
    if (L.failed) { inst[IR_RSTAT] = -7.0; return; }
    if (L.delta > 0.0) inst[IR_DLAST] = L.delta;
    inst[IR_DELTA] = L.delta;
}

// ---- gain phase: stage thread (instance, k < N), all stages at once ------------------------------------------------------------
// G, Lam^-1, K, k_ff of stage k from P_{k+1} (slot of stage k+1) and the stage's own inputs: the expressions of ric_matrix_step /
// ric_vector_step.  Leaves the gains in registers (K0, K1, kf: the caller forms du_k from them) and the rows of the forward sweep
// (dt K, dt k_ff) in the slot, over inputs the backward sweep has consumed.
template <int NX>
MPC_HD void lane_gain(const Params& P, int k, double delta, double hux0, double hux1, mpc_lds_ptr slot, mpc_lds_cptr slot_next,
                      double* K0, double* K1, double& kf0, double& kf1) {
    using D = Dim<NX>;
    using S = Slot<NX>;
    const double dt = P.dt;
    const double a03 = slot[S::A + 0], a04 = slot[S::A + 1], a13 = slot[S::A + 2], a14 = slot[S::A + 3], a42 = slot[S::A + 4], a43 = slot[S::A + 5];
    const double ruu0 = slot[S::RUU], ruu1 = slot[S::RUU + 1], gu0 = slot[S::GU], gu1 = slot[S::GU + 1];
    double Ps[D::NS], pv[NX], cn[NX];
#pragma unroll
    for (int i = 0; i < D::NS; ++i) Ps[i] = slot_next[S::PK + i];
#pragma unroll
    for (int i = 0; i < NX; ++i) { pv[i] = slot_next[S::PK + D::NS + i]; cn[i] = slot[S::CN + i]; }
    // rows delta, v of W = P+ (dtF)
    double W[2][3];
#pragma unroll
    for (int q = 0; q < 2; ++q) {
        const int i = 2 + q;
        const double pi0 = sym<NX>(Ps, i, 0), pi1 = sym<NX>(Ps, i, 1), pi4 = sym<NX>(Ps, i, 4);
        W[q][0] = pi4 * a42;
        double t = pi0 * a03 + pi1 * a13 + pi4 * a43;
        if (NX == 6) t += sym<NX>(Ps, i, 5) * dt;
        W[q][1] = t;
        W[q][2] = pi0 * a04 + pi1 * a14;
    }
    double G0[NX], G1[NX];
#pragma unroll
    for (int j = 0; j < NX; ++j) {
        double g0 = sym<NX>(Ps, 2, j), g1 = sym<NX>(Ps, 3, j);
        if (j >= 2 && j <= 4) { g0 += W[0][j - 2]; g1 += W[1][j - 2]; }
        G0[j] = dt * g0;
        G1[j] = dt * g1;
    }
    if (k == 0) { G1[2] += hux0; G1[3] += hux1; }
    const double L00 = ruu0 + dt * dt * sym<NX>(Ps, 2, 2) + delta;
    const double L01 = dt * dt * sym<NX>(Ps, 2, 3);
    const double L11 = ruu1 + dt * dt * sym<NX>(Ps, 3, 3) + delta;
    const double idet = 1.0 / (L00 * L11 - L01 * L01);
    const double i00 = L11 * idet, i01 = -L01 * idet, i11 = L00 * idet;
#pragma unroll
    for (int j = 0; j < NX; ++j) {
        K0[j] = -(i00 * G0[j] + i01 * G1[j]);
        K1[j] = -(i01 * G0[j] + i11 * G1[j]);
    }
    // k_ff = -Lam^-1 (gu + B'h),  h = p+ - P+ c  (rows delta, v only)
    double h2 = pv[2], h3 = pv[3];
#pragma unroll
    for (int j = 0; j < NX; ++j) { h2 -= sym<NX>(Ps, 2, j) * cn[j]; h3 -= sym<NX>(Ps, 3, j) * cn[j]; }
    const double l0 = gu0 + dt * h2, l1 = gu1 + dt * h3;
    kf0 = -(i00 * l0 + i01 * l1);
    kf1 = -(i01 * l0 + i11 * l1);
#pragma unroll
    for (int j = 0; j < NX; ++j) { slot[S::DK0 + j] = dt * K0[j]; slot[S::DK1 + j] = dt * K1[j]; }
    slot[S::DKF] = dt * kf0;
    slot[S::DKF + 1] = dt * kf1;
}

// ---- forward sweep: lane r of group g owns component r of dx -------------------------------------------------------------------
// dx_{k+1} = (A + B K) dx_k + B k_ff - c_{k+1}: rows delta, v read their closed-loop row (dt K, dt k_ff) from the slot, the others the
// entries of dtF.  One LDS round trip per stage (dx_k in, dx_{k+1}[r] out).
template <int NX>
MPC_HD double lane_forward_step(const Params& P, int r, mpc_lds_cptr slot) {
    using S = Slot<NX>;
    double dx[NX];
#pragma unroll
    for (int j = 0; j < NX; ++j) dx[j] = slot[S::DX + j];
    double t = dx[r < NX ? r : 0] - slot[S::CN + (r < NX ? r : 0)];
    if (r == 0) t += slot[S::A + 0] * dx[3] + slot[S::A + 1] * dx[4];
    if (r == 1) t += slot[S::A + 2] * dx[3] + slot[S::A + 3] * dx[4];
    if (r == 4) t += slot[S::A + 4] * dx[2] + slot[S::A + 5] * dx[3];
    if (NX == 6 && r == 5) t += P.dt * dx[3];
    if (r == 2 || r == 3) {
        const int o = (r == 2) ? S::DK0 : S::DK1;
        double du = slot[S::DKF + (r - 2)];
#pragma unroll
        for (int j = 0; j < NX; ++j) du += slot[o + j] * dx[j];
        t += du;
    }
    return t;
}

// ---- stage-thread side of the resident path ------------------------------------------------------------------------------------
// (one thread per (instance, stage) as in stage_block; these replace phase_load_scalars / phase_preload of the streaming path: the
//  iterate and the per-instance scalars never leave the registers, the step comes from the LDS slots)

// before a Riccati sweep: the k = 0 thread of every instance says whether the instance is iterating
template <int NX>
MPC_HD void res_announce(Ctx<NX>& c) {
    if (c.valid && c.k == 0) c.inst[IR_RSTAT] = (c.status == ST_RUNNING) ? 1.0 : 0.0;
}
// after a successful backward sweep: gains of the thread's stage (all stages at once)
template <int NX>
MPC_HD void res_gain(const Params& P, Ctx<NX>& c, mpc_lds_cptr slot_next) {
    if (!c.valid || c.k >= P.N) return;
    if (c.status != ST_RUNNING || !(c.inst[IR_RSTAT] > 0.0)) return;
    lane_gain<NX>(P, c.k, c.inst[IR_DELTA], c.inst[IR_HUX0], c.inst[IR_HUX1], c.slot, slot_next, c.K0, c.K1, c.kf0, c.kf1);
}
// after the forward sweep: the Newton step of the thread's stage, then what phase_premath computes from it
template <int NX>
MPC_HD void res_round_begin(const Params& P, Ctx<NX>& c, mpc_lds_cptr slot_next) {
    using D = Dim<NX>;
    using S = Slot<NX>;
    c.active = false;
    if (!c.valid) return;
    if (c.status == ST_RUNNING && c.inst[IR_RSTAT] < 0.0) c.status = -7;      // (riccati_instance: IS_STATUS = -7)
    c.active = c.status == ST_RUNNING;
    if (!c.active) return;
    if (!c.haveth0) {                          // first iteration: theta_max / theta_min from theta(w_0) (phase_load_scalars)
        c.thmax = 1e4 * fmax(1.0, c.theta);
        c.thmin = 1e-4 * fmax(1.0, c.theta);
    }
    const int N = P.N, k = c.k;
    double du0 = 0.0, du1 = 0.0;
#pragma unroll
    for (int i = 0; i < NX; ++i) {
        c.dz[2 + i] = c.slot[S::DX + i];
        c.dxn[i] = (k < N) ? slot_next[S::DX + i] : 0.0;
    }
    if (k < N) {                               // du_k = K dx_k + k_ff (riccati_forward_step)
        du0 = c.kf0; du1 = c.kf1;
#pragma unroll
        for (int j = 0; j < NX; ++j) { du0 += c.K0[j] * c.dz[2 + j]; du1 += c.K1[j] * c.dz[2 + j]; }
    }
    c.dz[0] = du0;
    c.dz[1] = du1;
    PreTmp<NX> tmp;
#pragma unroll
    for (int i = 0; i < D::NPK; ++i) tmp.pk[i] = c.slot[S::PK + i];
#pragma unroll
    for (int i = 0; i < NX; ++i) tmp.lam[i] = c.lam[i];
    phase_premath<NX>(P, c, tmp);
}
// end of the solve: what k_egest reads (iterate, status, iteration count, KKT error) goes to the workspace
template <int NX>
MPC_HD void res_store(const Params& P, Ctx<NX>& c) {
    constexpr int NZ = NX + 2;
    if (!c.valid) return;
    ws_store_rows<NZ>(MPC_ROWS(MPC_K(P.Z, NZ, 0, e)), c.z);
    if (c.k == 0) {
        MPC_S(P.ISC, IS_STATUS) = c.status;
        MPC_S(P.ISC, IS_ITERS) = c.iters;
        MPC_S(P.SC, SC_E0) = c.e0;
        MPC_S(P.SC, SC_MU) = c.mu;
    }
}

}  // namespace mpc
