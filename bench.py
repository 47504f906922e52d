#!/usr/bin/env python3
"""
bench.py -- MPC steps/s of the batched NLP solve (the path of MPC_Planner/optimizer.py:607) on MI355X.

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
           bench.py --gpus N --steps K --warmup W

One "step" = one pass of the hot path over one batch: B = 4096 independent instances of the N = 30, nx = 6,
nu = 2 kinematic-bicycle lane-following NLP (BASELINE.json `metric`), synthetic references generated as
SURVEY.md section 8(d) prescribes, inputs and outputs resident in HBM (mpc_solve_batch_dev).  Every instance is
solved to the reference's IPOPT tolerance (tol 1e-8, max_iter 100): `value` counts converged NLP solves per
second, whole job.  With N > 1 each rank solves its own B instances on its own GPU (weak scaling, instances are
independent: no data-path collective) and the result rows are all-gathered over RCCL once per step.

Prints ONE JSON line on rank 0 (contract of the build driver) with two extra objects:
  roofline      dominant kernel's algorithmic bytes per launch / its mean launch duration (HIP events on the
                solve stream, measured in a second, profiled pass over the same K steps) against the 8 TB/s HBM peak
  cpu_baseline  the oracle (oracle/mpc_oracle.c, "port") on the host cores, bounded sample of the same workload
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

N_HORIZON, NX, NU, BATCH = 30, 6, 2, 4096
HBM_PEAK_GBS = 8000.0           # MI355X_MICROARCH.md: HBM3E 8 TB/s spec (6.3 TB/s achievable)


def algorithmic_bytes(N, nx, nu=2):
    """bytes per instance-iteration and per MPC step (I/O part)."""
    nz = nx + nu
    K = nz * (nz + 1) // 2 + nx * nz + nz + nx          # stage Hessian (sym) + [A B] + gradient + defect
    m = 16                                               # 7 bound sides + 9 obstacle rows
    I = nz + nx + 2 * m                                  # iterate: primal + equality multipliers + slack/dual pairs
    n_w = nu * N + nx * (N + 1)
    b_iter = 8 * 2 * (N + 1) * (K + I)
    b_io = 8 * 3 * n_w
    b_riccati = 8 * (N + 1) * (K + nz + nx)              # read the stage block once, write step + multipliers
    b_stage = b_iter - b_riccati                         # write the stage block, read/write the iterate, read the step
    return dict(K=K, I=I, b_iter=b_iter, b_io=b_io, b_riccati=b_riccati, b_stage=b_stage)


def pmc_traffic(kernel):
    """HBM bytes per launch of `kernel` from the committed rocprofv3 PMC summary (separate --pmc passes over this very
    command, tools/pmc_run.sh + tools/pmc_summary.py; counters cannot be read from inside the timed process)."""
    path = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    if not os.path.exists(path):
        return None
    try:
        return json.load(open(path))[kernel]["hbm_bytes_per_launch_mean"]
    except (KeyError, ValueError):
        return None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=BATCH)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist
    import mpc_amd
    from mpc_amd import sharding
    from oracle.nlp_numpy import NLPConfig, synthetic_batch

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the HIP path has no CPU fallback")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
    assert world == args.gpus or world == 1, "launch with torch.distributed.run --nproc-per-node N for --gpus N"

    B = args.batch
    cfg = NLPConfig(N=N_HORIZON, nx=NX)                 # ZAM_Over-1_1 lane-following weights, dummy obstacle
    x0, p = synthetic_batch(cfg, B, start=rank * B)     # per-instance rng(20240929 + global index)
    d_x0, d_p = torch.from_numpy(x0).to(dev), torch.from_numpy(p).to(dev)
    d_out = torch.empty_like(d_x0)
    d_st = torch.empty(B, dtype=torch.int32, device=dev)
    d_it = torch.empty(B, dtype=torch.int32, device=dev)
    d_kkt = torch.empty(B, dtype=torch.float64, device=dev)
    stream = torch.cuda.current_stream(dev).cuda_stream

    def make(fixed):
        return mpc_amd.BatchedMPCSolver(cfg.N, cfg.nx, Q=cfg.Qdiag, R=cfg.R, obstacle_centers=cfg.obstacle_centers,
                                        fixed_iters=fixed, device=local_rank)

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize(dev)

    gathered = [torch.empty_like(d_out) for _ in range(world)] if world > 1 else None

    def step(solver):
        solver.solve_device(B, d_x0.data_ptr(), d_p.data_ptr(), d_out.data_ptr(), d_st.data_ptr(), d_it.data_ptr(),
                            d_kkt.data_ptr(), stream=stream)
        if world > 1:                                    # the only exchange of the path: final gather of the rows
            dist.all_gather(gathered, d_out)

    def timed(solver, steps, warmup):
        for _ in range(warmup):
            step(solver)
        barrier()
        t0 = time.perf_counter()
        for _ in range(steps):
            step(solver)
        barrier()
        dt = time.perf_counter() - t0
        return sharding.max_over_ranks(dt, device=dev) if world > 1 else dt

    solver = make(0)
    dt = timed(solver, args.steps, args.warmup)
    ms_per_step = dt / args.steps * 1e3
    value = world * B * args.steps / dt
    st, it, kkt = d_st.cpu().numpy(), d_it.cpu().numpy(), d_kkt.cpu().numpy()
    converged = float((st == 1).mean())
    mean_it, max_it = float(it.mean()), int(it.max())

    # ---- roofline: second pass over the same K steps with HIP events around every kernel launch
    ab = algorithmic_bytes(cfg.N, cfg.nx)
    solver.set_profiling(True)
    ric_ms = ric_n = stg_ms = stg_n = pipe_ms = pipe_n = 0.0
    pipe_stats = None
    torch.cuda.synchronize(dev)
    tp0 = time.perf_counter()
    for _ in range(args.steps):
        step(solver)
        pr = solver.get_profile()
        ric_ms += pr["riccati_ms"]; ric_n += pr["riccati_launches"]
        stg_ms += pr["stage_ms"]; stg_n += pr["stage_launches"]
        pp = solver.get_pipeline_profile()
        if pp["ran"]:                                    # all iterations in ONE persistent launch (k_pipeline)
            pipe_ms += pp["ms"]; pipe_n += 1
            pipe_stats = {k: (float(v) if isinstance(v, (float, np.floating)) else v) for k, v in pp.items()}
    torch.cuda.synchronize(dev)
    prof_ms_per_step = (time.perf_counter() - tp0) / args.steps * 1e3
    solver.set_profiling(False)
    inst_iters = float(it.sum())                         # instance-iterations actually performed per step
    kern = {}
    for name, ms, n, bpi in (("k_riccati", ric_ms, ric_n, ab["b_riccati"]), ("k_stage", stg_ms, stg_n, ab["b_stage"]),
                             ("k_pipeline", pipe_ms, pipe_n, ab["b_riccati"] + ab["b_stage"])):
        if n == 0:
            continue
        n_per_step = n / args.steps
        avg_us = ms / max(n, 1) * 1e3
        bytes_per_launch = bpi * inst_iters / max(n_per_step, 1)      # active instances per launch x bytes per instance
        kern[name] = dict(avg_us=avg_us, launches_per_step=n_per_step, bytes_per_launch=bytes_per_launch,
                          gbs=bytes_per_launch / (avg_us * 1e-6) / 1e9, total_ms_per_step=ms / args.steps)
    dom = max(kern, key=lambda k: kern[k]["total_ms_per_step"])
    roofline = dict(bound="hbm", kernel=dom, achieved=kern[dom]["gbs"], peak=HBM_PEAK_GBS, unit="GB/s",
                    frac=kern[dom]["gbs"] / HBM_PEAK_GBS, traffic=pmc_traffic(dom),
                    avg_launch_us=kern[dom]["avg_us"], launches_per_step=kern[dom]["launches_per_step"],
                    algorithmic_bytes_per_launch=kern[dom]["bytes_per_launch"],
                    other_kernel={k: v for k, v in kern.items() if k != dom},
                    whole_step=dict(bytes_per_mpc_step=ab["b_io"] + mean_it * ab["b_iter"],
                                    gbs=value / world * (ab["b_io"] + mean_it * ab["b_iter"]) / 1e9,
                                    frac=value / world * (ab["b_io"] + mean_it * ab["b_iter"]) / 1e9 / HBM_PEAK_GBS),
                    profiled_ms_per_step=prof_ms_per_step, pipeline=pipe_stats)

    # ---- deterministic-work variant (exactly 20 iterations per instance, no early exit)
    fsolver = make(20)
    dtf = timed(fsolver, args.steps, args.warmup)
    fixed20 = dict(value=world * B * args.steps / dtf, ms_per_step=dtf / args.steps * 1e3,
                   hbm_frac=B * args.steps / dtf * (ab["b_io"] + 20 * ab["b_iter"]) / 1e9 / HBM_PEAK_GBS)

    # ---- CPU baseline: the oracle on the host cores (rank 0, single-GPU run only)
    cpu_baseline = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        from oracle.binding import OracleSolver
        osol = OracleSolver(cfg)
        avail = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
        xs, ps = synthetic_batch(cfg, 4096)
        best = None
        for cores in sorted({min(avail, c) for c in (16, 32, 64, 128, avail)}):   # OpenMP over instances; keep the best count
            osol.solve_batch(xs[:256], ps[:256], nthreads=cores)          # warm the thread pool
            t0 = time.perf_counter()
            ro = osol.solve_batch(xs, ps, nthreads=cores)
            t = time.perf_counter() - t0
            if best is None or t < best[1]:
                best = (cores, t)
        cores = best[0]
        reps = max(1, int(round(3.0 / best[1])))                            # ~3 s wall on the chosen thread count
        t0 = time.perf_counter()
        for _ in range(reps):
            ro = osol.solve_batch(xs, ps, nthreads=cores)
        t_all = time.perf_counter() - t0
        t0 = time.perf_counter()
        osol.solve_batch(xs[:1024], ps[:1024], nthreads=1)
        t_one = time.perf_counter() - t0
        model = "unknown"
        try:
            with open("/proc/cpuinfo") as fh:
                model = next(l.split(":", 1)[1].strip() for l in fh if l.startswith("model name"))
        except Exception:
            pass
        cpu_baseline = dict(value=reps * len(xs) / t_all, unit="MPC steps/s", cores=cores, kind="port",
                            sample=f"{reps} x 4096 instances of the same workload (N=30, nx=6), oracle/mpc_oracle.c, "
                                   f"OpenMP over instances, all converged={bool((ro['status'] == 1).all())}",
                            single_thread_value=1024 / t_one, cpu_model=model, host_cpus=avail,
                            published_casadi_ipopt="25.1 steps/s (N=10, 1 instance, unknown CPU; BASELINE.md section 1)")

    # ---- the paths around the solve, single-GPU run only, a few hundred milliseconds in total
    other_paths = None
    if rank == 0 and world == 1:
        other_paths = {}
        try:
            s5 = mpc_amd.BatchedMPCSolver(N_HORIZON, 5, Q=cfg.Qdiag[:5], R=cfg.R, obstacle_centers=cfg.obstacle_centers, device=local_rank)
            s5.set_bounds()                                                      # the reference's default limits
            L, Bc = 60, B
            k = np.arange(L)
            path = np.stack([k * 1.5 * np.cos(0.1), k * 1.5 * np.sin(0.1)], axis=1)
            rng = np.random.default_rng(0)
            init = np.tile([0.0, 0.0, 0.0, 15.0, 0.1], (Bc, 1))
            init[:, 1] += rng.uniform(-0.5, 0.5, Bc)
            init[:, 3] *= rng.uniform(0.9, 1.1, Bc)
            P_, O_ = np.tile(path, (Bc, 1, 1)), np.full((Bc, L), 0.1)
            s5.closed_loop(init[:64], P_[:64], O_[:64], np.full(64, 15.0), L)
            t0 = time.perf_counter()
            _, _, st_ = s5.closed_loop(init, P_, O_, np.full(Bc, 15.0), L)
            tcl = time.perf_counter() - t0
            other_paths["closed_loop"] = dict(ego_steps_per_s=Bc * L / tcl, ms_per_step_of_batch=tcl / L * 1e3, batch=Bc, steps=L, horizon=N_HORIZON,
                                              converged_frac=float((st_ == 1).mean()),
                                              note="mpc_closed_loop_batch, nx=5, host buffers in/out once per call (row f1)")
            sf = mpc_amd.BatchedMPCSolver(10, 5, Q=(2.0, 2.0, 50.0, 0.1, 5.0), R=(2.0, 0.2), P=(4.0, 4.0, 100.0, 0.2, 10.0), device=local_rank)
            zi = np.array([0.0, 0.0, 29.9948, -1.1501, 0.0, 19.0, 0.03495])
            zb = np.tile(zi, (B, 10, 1))
            kk = np.arange(1, 11)
            par = np.tile(np.hstack([np.stack([zi[2] + kk * 2 * np.cos(0.03495), zi[3] + kk * 2 * np.sin(0.03495)], 1), np.full((10, 1), 20.0),
                                     np.full((10, 1), 0.03495), np.tile([-100.0, 0, -100, 0, -100, 0], (10, 1))]), (B, 1, 1))
            lbf = np.array([-0.4, -11.5, -np.inf, -np.inf, -1.066, 0.0, -np.inf])
            ubf = np.array([0.4, 11.5, np.inf, np.inf, 1.066, 50.8, np.inf])
            hlf, huf = np.concatenate(([0.0], np.full(9, 1.44))), np.concatenate(([11.5 ** 2], np.full(9, np.inf)))
            sf.forces_solve(zb[:64], zb[:64, 0, 2:], par[:64], lbf, ubf, hlf, huf)
            t0 = time.perf_counter()
            _, fl_, it_, _ = sf.forces_solve(zb, zb[:, 0, 2:], par, lbf, ubf, hlf, huf)
            tf = time.perf_counter() - t0
            other_paths["forces_sqp_step"] = dict(solves_per_s=B / tf, ms_per_batch=tf * 1e3, batch=B, horizon=10, solved_frac=float((fl_ == 1).mean()),
                                                  mean_qp_iterations=float(it_.mean()),
                                                  note="mpc_forces_solve_batch, host buffers incl. PCIe and per-call allocation (row f3)")
        except Exception as e:      # never let the side measurements take the bench line down
            other_paths["error"] = repr(e)

    if rank == 0:
        out = dict(metric="MPC steps/sec (N=30, nx=6 nu=2) at batch=4096", value=value, unit="MPC steps/s", n_gpus=world,
                   steps=args.steps, warmup=args.warmup, ms_per_step=ms_per_step, higher_is_better=True, scaling="weak",
                   vs_baseline=None, dtype="f64", data="synthetic",
                   config=dict(workload="N=30 nx=6 nu=2 kinematic-bicycle lane-following (ZAM_Over-1_1 LF weights, dummy "
                                        "obstacle), batch=%d per GPU, solved to tol 1e-8 (max_iter 100)" % B,
                               batch_per_gpu=B, horizon=cfg.N, nx=cfg.nx, nu=2, parallelism="independent instances x%d" % world,
                               mode="converged", gpu=torch.cuda.get_device_name(dev)),
                   converged_frac=converged, mean_iters=mean_it, max_iters=max_it, kkt_max=float(kkt.max()),
                   fixed20=fixed20, roofline=roofline, cpu_baseline=cpu_baseline, other_paths=other_paths)
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
