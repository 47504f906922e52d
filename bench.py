#!/usr/bin/env python3
"""
bench.py -- MPC steps/s of the batched NLP solve (the path of MPC_Planner/optimizer.py:607) on MI355X.

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
           bench.py --gpus N --steps K --warmup W

One "step" = one pass of the hot path over one batch: B = 4096 independent instances of the N = 30, nx = 6,
nu = 2 kinematic-bicycle lane-following NLP (BASELINE.json `metric`), synthetic references generated as
SURVEY.md section 8(d) prescribes (tools/workloads.py), inputs and outputs resident in HBM (mpc_solve_batch_dev).
Every instance is solved to the reference's IPOPT tolerance (tol 1e-8, max_iter 100): `value` counts converged NLP
solves per second, whole job.  With N > 1 each rank solves its own B instances on its own GPU (weak scaling, instances
are independent: no data-path collective) and ONE packed block per rank -- the result rows of the step with their status and
iteration counts -- is all-gathered over RCCL, on RCCL's own stream while the next batch is being solved (`--gather overlap`, the
default; the last gathers are waited for inside the timed region) or inside the step (`--gather sync`); the collective alone is timed
too (`gather.gather_ms_alone`), and the statistics of the line are reduced over the ranks.

`--gpus N` without a torch.distributed launcher (WORLD_SIZE unset) starts the N ranks itself and FAILS when fewer than N
devices are visible; it never reports a smaller n_gpus than it was asked for.

`--workload mixed` runs BASELINE configuration 5 instead (mixed scenario sweep: 4096 rows per GPU dealt over the four
problem families, one handle and one stream per family -- solved concurrently, collision avoidance first --, one packed
all-gather at the end).

Prints ONE JSON line on rank 0 (contract of the build driver; compact: < 2000 characters, arrays where the long form has
objects -- the long form goes to stderr as `[bench detail] {...}` and to `--detail FILE`) with extra objects:
  roofline      the iteration loop of a solve = ONE launch of k_pipeline (tiles with many instances iterating) followed by ONE launch
                of k_solve_wg (the stragglers, one wavefront per two instances): algorithmic bytes of the instance-iterations
                each performed / its mean launch duration (HIP events on the solve stream, a second, profiled pass over the
                same K steps); `achieved` = bytes of both / duration of both, against the 8 TB/s HBM peak and against the
                library's own streaming copy kernel; `kernels` has the two launches one by one; `traffic` = HBM bytes per
                launch pair from rocprofv3 PMC passes run by this very process (FETCH_SIZE / WRITE_SIZE, gfx950 correction)
  configs       BASELINE.json configurations 2, 3, 4 and one shard of 5 under the same clock (a few batches each), one array per
                configuration: [id, batch, ms per batch, steps/s, converged fraction, mean iterations, max iterations,
                roofline fraction of the iteration loop, instances that took the second chance]
  cpu_baseline  the oracle (oracle/mpc_oracle.c, "port") on the host cores, bounded sample of the same workload;
                plus a run-time probe for CasADi/IPOPT (the reference's own solver) on the host
"""
import argparse
import csv
import glob
import json
import os
import shutil
import socket
import subprocess
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))

N_HORIZON, NX, NU, BATCH = 30, 6, 2, 4096
HBM_PEAK_GBS = 8000.0           # MI355X_MICROARCH.md: HBM3E 8 TB/s spec (6.3 TB/s achievable)
JSON_OUT = sys.stdout            # main() swaps it for a private copy of file descriptor 1
HOST_CPUS = 1                    # main() counts the host cpus before the OpenMP runtime pins the main thread
PUBLISHED_CASADI = "25.1 steps/s (N=10, 1 instance, unknown CPU, graph rebuilt every step; BASELINE.md section 1: 36-41 ms per step)"


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


def committed_traffic(kernel):
    path = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    try:
        return json.load(open(path))[kernel]["hbm_bytes_per_launch_mean"]
    except (OSError, KeyError, ValueError):
        return None


def measure_traffic(kernel_tags, timeout=240, child="mpc", fallback="k_pipeline"):
    """HBM bytes per solve of the kernels of the iteration loop (mean per launch, summed over the kernels), measured NOW: two rocprofv3 passes (`--pmc FETCH_SIZE`, `--pmc WRITE_SIZE`,
    kernel trace only -- counters get their own runs, MI355X_MICROARCH.md) over `bench.py --pmc-child`, which performs three
    converged-mode solves of the headline batch and nothing else (`--pmc-child forces`: three FORCES-mode SQP steps of the
    `other_paths` batch).  bytes = (2 x FETCH_SIZE + WRITE_SIZE) KiB x 1024: on gfx950
    FETCH_SIZE counts 64 B per 128-byte request (the guide's correction; re-checked with tools/ubench/ldpat.hip).
    Returns (bytes per launch | None, source text)."""
    exe = shutil.which("rocprofv3")
    if exe is None:
        return committed_traffic(fallback), "rocprofv3 not found; committed profiles/pmc_traffic.json"
    out = tempfile.mkdtemp(prefix="bench_pmc_", dir="/tmp")
    vals = {}
    env = dict(os.environ, TMPDIR="/tmp")
    for ctr in ("FETCH_SIZE", "WRITE_SIZE"):
        d = os.path.join(out, ctr)
        cmd = [exe, "--kernel-trace", "--pmc", ctr, "-d", d, "-o", "p", "--output-format", "csv", "--", sys.executable,
               os.path.join(ROOT, "bench.py"), "--pmc-child", child]
        try:
            subprocess.run(cmd, cwd="/tmp", env=env, timeout=timeout, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, check=True)
        except (subprocess.SubprocessError, OSError) as e:
            shutil.rmtree(out, ignore_errors=True)
            return committed_traffic(fallback), "rocprofv3 pass failed (%s); committed profiles/pmc_traffic.json" % type(e).__name__
        per_kernel = {}
        for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
            with open(f) as fh:
                for row in csv.DictReader(fh):
                    for tag in kernel_tags:
                        if tag in row["Kernel_Name"] and row["Counter_Name"] == ctr:
                            per_kernel.setdefault(tag, []).append(float(row["Counter_Value"]))
        if kernel_tags[0] not in per_kernel:
            shutil.rmtree(out, ignore_errors=True)
            return committed_traffic(fallback), "no %s rows for %s; committed profiles/pmc_traffic.json" % (ctr, kernel_tags[0])
        # (every dispatch is one row per XCD-summed counter; per solve: one launch of each kernel)
        vals[ctr] = sum(sum(v) / len(v) for v in per_kernel.values())
    shutil.rmtree(out, ignore_errors=True)
    return (2.0 * vals["FETCH_SIZE"] + vals["WRITE_SIZE"]) * 1024.0, \
        "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE passes run by this process (2*FETCH_SIZE + WRITE_SIZE, KiB)"


def copy_bandwidth(solver):
    """achievable HBM bandwidth of a kernel that only moves data: the library's own streaming copy (16 bytes per lane and access,
    1 GiB in, 1 GiB out; bytes read + written / time), GB/s -- MI355X_MICROARCH.md quotes 6.29 TB/s for a float4 copy"""
    return solver.measure_copy_bandwidth(1 << 30, 5)


def casadi_probe(fam, x0, p, wl):
    """probe `import casadi` on this host; when it is there, time nlpsol('ipopt') on the restated NLP
    (optimizer.py:373-558), solver built once and rebuilt per step (optimizer.py:605); when not, say so."""
    try:
        import casadi as ca
    except Exception as e:                               # expected here: not installed, no network
        return dict(available=False, note="import casadi failed on this host (%s): the reference's own CasADi/IPOPT path cannot be timed; "
                                          "its only published figure is %s" % (type(e).__name__, PUBLISHED_CASADI))
    try:
        N, nx, dt = fam.N, 5, fam.dt
        lbx, ubx, lbg, ubg = wl.bounds(wl.Family("probe", N, 5, fam.Q, fam.R))

        def build():
            U, X = ca.SX.sym("U", 2, N), ca.SX.sym("X", nx, N + 1)
            Xr = ca.SX.sym("Xr", nx, N + 1)
            Ur = ca.SX.sym("Ur", 2, N)
            Q, R = ca.diag(ca.DM(list(fam.Q))), ca.diag(ca.DM(list(fam.R)))
            f = lambda x, u: ca.vertcat(x[3] * ca.cos(x[4]), x[3] * ca.sin(x[4]), u[0], u[1], x[3] / 2.5789128 * ca.tan(x[2]))   # noqa: E731
            J = 0
            g = [ca.sqrt((U[1, 0] ** 2 + X[3, 0] * (ca.tan(X[2, 0]) * X[3, 0] / 2.578)) ** 2), X[:, 0] - Xr[:, 0]]
            for i in range(N):
                e = X[:, i] - Xr[:, i + 1]
                J = J + ca.mtimes([e.T, Q, e]) + ca.mtimes([U[:, i].T, R, U[:, i]])
                g.append(X[:, i + 1] - (X[:, i] + dt * f(X[:, i], U[:, i])))
            for i in range(N + 1):
                for j, sg in ((0, 0.0), (1, 1.0), (2, -1.0)):
                    cx, cy = X[0, i] + sg * 0.75 * ca.cos(X[4, i]) + 100.0, X[1, i] + sg * 0.75 * ca.sin(X[4, i])
                    g += [ca.sqrt(cx ** 2 + cy ** 2)] * 3
            nlp = dict(f=J, x=ca.vertcat(ca.reshape(U, -1, 1), ca.reshape(X, -1, 1)), p=ca.vertcat(ca.reshape(Ur, -1, 1), ca.reshape(Xr, -1, 1)),
                       g=ca.vertcat(*g))
            return ca.nlpsol("solver", "ipopt", nlp, {"ipopt.max_iter": 100, "ipopt.print_level": 0, "print_time": 0,
                                                      "ipopt.acceptable_tol": 1e-8, "ipopt.acceptable_obj_change_tol": 1e-6})
        keep = np.r_[0:2 * N, [2 * N + 6 * k + i for k in range(N + 1) for i in range(5)]] if x0.shape[1] != 2 * N + 5 * (N + 1) else slice(None)
        xs, ps = x0[:8, keep], p[:8, keep]
        sol = build()
        t0 = time.perf_counter()
        for b in range(len(xs)):
            sol(x0=xs[b], p=ps[b], lbg=lbg, ubg=ubg, lbx=lbx, ubx=ubx)
        t_once = (time.perf_counter() - t0) / len(xs)
        t0 = time.perf_counter()
        for b in range(2):
            build()(x0=xs[b], p=ps[b], lbg=lbg, ubg=ubg, lbx=lbx, ubx=ubx)
        t_rebuild = (time.perf_counter() - t0) / 2
        return dict(available=True, version=ca.__version__, steps_per_s_solver_built_once=1.0 / t_once, steps_per_s_rebuilt_per_step=1.0 / t_rebuild,
                    threads=1, sample="%d instances of the N=%d nx=5 part of the same workload" % (len(xs), N))
    except Exception as e:
        return dict(available=True, error=repr(e), note="casadi imported but the timing failed; published figure: " + PUBLISHED_CASADI)


def spawn_ranks(args):
    """--gpus N without a launcher: become `torch.distributed.run` with N ranks -- or fail; never measure fewer GPUs than asked"""
    import torch
    have = torch.cuda.device_count() if torch.cuda.is_available() else 0
    if have < args.gpus:
        raise SystemExit("bench.py --gpus %d: only %d GPU(s) visible on this host -- refusing to report a %d-GPU figure from fewer devices"
                         % (args.gpus, have, args.gpus))
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus), "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    os.execv(sys.executable, cmd)


def forces_setup(B, local_rank=0):
    """the FORCES-mode SQP step of `other_paths` (row f3): solver, a closure that runs one batch on device-resident buffers, the flag /
    iteration rows it fills, and the bound vectors of the lane-following formulation"""
    import torch
    import mpc_amd
    sf = mpc_amd.BatchedMPCSolver(10, 5, Q=(2.0, 2.0, 50.0, 0.1, 5.0), R=(2.0, 0.2), P=(4.0, 4.0, 100.0, 0.2, 10.0), device=local_rank)
    zi = np.array([0.0, 0.0, 29.9948, -1.1501, 0.0, 19.0, 0.03495])
    zb = np.tile(zi, (B, 10, 1))
    kk = np.arange(1, 11)
    par = np.tile(np.hstack([np.stack([zi[2] + kk * 2 * np.cos(0.03495), zi[3] + kk * 2 * np.sin(0.03495)], 1), np.full((10, 1), 20.0),
                             np.full((10, 1), 0.03495), np.tile([-100.0, 0, -100, 0, -100, 0], (10, 1))]), (B, 1, 1))
    lbf = np.array([-0.4, -11.5, -np.inf, -np.inf, -1.066, 0.0, -np.inf])
    ubf = np.array([0.4, 11.5, np.inf, np.inf, 1.066, 50.8, np.inf])
    hlf, huf = np.concatenate(([0.0], np.full(9, 1.44))), np.concatenate(([11.5 ** 2], np.full(9, np.inf)))
    dev = torch.device("cuda", local_rank)
    d_z, d_xi, d_par = torch.from_numpy(zb).to(dev), torch.from_numpy(np.ascontiguousarray(zb[:, 0, 2:])).to(dev), torch.from_numpy(par).to(dev)
    d_o = torch.empty_like(d_z)
    d_fl = torch.empty(B, dtype=torch.int32, device=dev)
    d_it = torch.empty(B, dtype=torch.int32, device=dev)
    d_rs = torch.empty(B, dtype=torch.float64, device=dev)

    def fstep():
        sf.forces_solve_device(B, d_z.data_ptr(), d_xi.data_ptr(), d_par.data_ptr(), lbf, ubf, hlf, huf, d_o.data_ptr(), d_fl.data_ptr(),
                               d_it.data_ptr(), d_rs.data_ptr())
    fstep.keep = (d_z, d_xi, d_par, d_o, d_rs)
    return sf, fstep, d_fl, d_it, lbf, ubf, hlf, huf


def pmc_child(kind="mpc"):
    """the command the PMC passes profile: three converged-mode solves of the headline batch, nothing else"""
    import torch
    import mpc_amd  # noqa: F401
    import workloads as wl
    if kind == "forces":
        _, fstep, *_ = forces_setup(BATCH)
        for _ in range(3):
            fstep()
        torch.cuda.synchronize()
        return
    fam = wl.FAMILIES["zamlf_n30_nx6"]
    x0, p = wl.batch(fam, BATCH)
    dev = torch.device("cuda", 0)
    d_x0, d_p = torch.from_numpy(x0).to(dev), torch.from_numpy(p).to(dev)
    d_out = torch.empty_like(d_x0)
    s = wl.make_solver(fam)
    for _ in range(3):
        s.solve_device(BATCH, d_x0.data_ptr(), d_p.data_ptr(), d_out.data_ptr())
    torch.cuda.synchronize()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=BATCH)
    ap.add_argument("--workload", choices=("headline", "mixed"), default="headline")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-traffic", action="store_true", help="skip the rocprofv3 PMC passes (roofline.traffic falls back to the committed figure)")
    ap.add_argument("--gather", choices=("overlap", "sync"), default="overlap",
                    help="multi-GPU: the all-gather of a step's result rows runs on RCCL's stream under the next solve (overlap, default) or "
                         "is waited for inside the step (sync) -- the fallback should the collective starve behind the persistent kernel")
    ap.add_argument("--detail", default=None, help="write the full (long) result object to this file; it goes to stderr anyway")
    ap.add_argument("--headline-only", action="store_true",
                    help="only the timed converged-mode steps and their roofline pass (no fixed-20 variant, configurations 2 - 5, side paths): "
                         "under `rocprofv3 --kernel-trace --stats` the averages of k_pipeline<6> / k_solve_wg<6> are then those of the bench line")
    ap.add_argument("--pmc-child", nargs="?", const="mpc", default=None, help=argparse.SUPPRESS)
    args = ap.parse_args()
    if args.pmc_child:
        return pmc_child(args.pmc_child)
    if args.gpus < 1:
        raise SystemExit("--gpus must be >= 1")
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        spawn_ranks(args)                                # does not return

    # stdout belongs to the ONE JSON line: RCCL prints its banner and warnings to file descriptor 1 (NCCL_DEBUG=VERSION is exported on
    # the GPU boxes) -- from here on descriptor 1 is stderr for everybody, and the line goes out through a copy of the original
    global JSON_OUT
    sys.stdout.flush()
    JSON_OUT = os.fdopen(os.dup(1), "w")
    os.dup2(2, 1)

    # (the cpu_baseline leg: OpenMP threads of the oracle pinned to cores, close to each other -- an unbound run moved 2.7x between boxes)
    global HOST_CPUS
    HOST_CPUS = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)     # (before the OpenMP runtime binds this thread)
    if args.gpus == 1 and int(os.environ.get("WORLD_SIZE", "1")) == 1:
        # (single-rank runs only -- the one place the cpu_baseline leg runs: with a binding policy set, the OpenMP runtime also pins the
        #  MAIN thread to the first place, and the main threads of eight ranks would share one core)
        os.environ.setdefault("OMP_PROC_BIND", "close")
        os.environ.setdefault("OMP_PLACES", "cores")
    import torch
    import torch.distributed as dist
    import mpc_amd
    from mpc_amd import sharding
    import workloads as wl

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit("bench.py --gpus %d was launched with WORLD_SIZE=%d: the two must agree" % (args.gpus, world))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the HIP path has no CPU fallback")
    if torch.cuda.device_count() <= local_rank:
        raise SystemExit("rank %d: local rank %d but only %d GPU(s) visible" % (rank, local_rank, torch.cuda.device_count()))
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    # (MPC_BENCH_FORCE_GATHER=1: a single rank goes through the collective code of the multi-GPU path -- a one-GPU box can exercise it)
    gather = world > 1 or os.environ.get("MPC_BENCH_FORCE_GATHER") == "1"
    if gather:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
    stream = torch.cuda.current_stream(dev).cuda_stream

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize(dev)

    def timed(step_fn, steps, warmup, finish=None):
        """contract timing: W untimed steps, then exactly K steps between two barrier + synchronize pairs, max over ranks;
        the per-step host times (every converged-mode call ends with a stream synchronisation) give the median beside it.
        `finish`: waits for what the steps left in flight (the gathers of the last two steps), inside the timed region"""
        for _ in range(warmup):
            step_fn()
        if finish:
            finish()
        barrier()
        per = []
        t0 = time.perf_counter()
        for _ in range(steps):
            ts = time.perf_counter()
            step_fn()
            per.append(time.perf_counter() - ts)
        if finish:
            finish()
        barrier()
        dt = time.perf_counter() - t0
        return (sharding.max_over_ranks(dt, device=dev) if world > 1 else dt), per

    if args.workload == "mixed":
        return run_mixed(args, torch, dist, mpc_amd, sharding, wl, world, rank, local_rank, dev, stream, timed)

    B = args.batch
    fam = wl.FAMILIES["zamlf_n30_nx6"]                  # ZAM_Over-1_1 lane-following weights, dummy obstacle, N = 30, nx = 6
    assert (fam.N, fam.nx) == (N_HORIZON, NX)
    x0, p = wl.batch(fam, B, start=rank * B)            # per-instance rng(20240929 + global index)
    d_x0, d_p = torch.from_numpy(x0).to(dev), torch.from_numpy(p).to(dev)
    d_out = torch.empty_like(d_x0)
    d_st = torch.empty(B, dtype=torch.int32, device=dev)
    d_it = torch.empty(B, dtype=torch.int32, device=dev)
    d_kkt = torch.empty(B, dtype=torch.float64, device=dev)
    # The only exchange of the path is the final gather of the result rows (north_star).  It runs on RCCL's own stream while the next
    # batch is being solved: two output buffers, the gather of step i is waited for before step i + 2 writes its buffer again, and
    # the last ones inside the timed region (`drain`).  The solver's persistent kernel fills every CU, so the collective's workgroups
    # find room in the straggler phase of the following solve, where most CUs idle.
    # What travels is ONE block per rank: the rows with their status and iteration count behind them (sharding.pack_rows).
    d_outs = [d_out, torch.empty_like(d_out)] if gather else [d_out]
    PW = sharding.packed_width(d_out.shape[1])
    packed = [torch.empty((B, PW), dtype=torch.float64, device=dev) for _ in d_outs] if gather else None
    gathered = [torch.empty((world * B, PW), dtype=torch.float64, device=dev) for _ in d_outs] if gather else None
    works = [None] * len(d_outs)
    n_step = [0]
    gather_wait = [0.0, 0]                                # seconds the host waited for gathers inside steps (sync mode: all of them), count

    def step(solver):
        j = n_step[0] % len(d_outs)
        n_step[0] += 1
        if works[j] is not None:
            works[j].wait()
            works[j] = None
        solver.solve_device(B, d_x0.data_ptr(), d_p.data_ptr(), d_outs[j].data_ptr(), d_st.data_ptr(), d_it.data_ptr(),
                            d_kkt.data_ptr(), stream=stream)
        if gather:
            sharding.pack_rows(d_outs[j], d_st, d_it, out=packed[j])                          # (three strided copies on the solve stream)
            works[j] = sharding.gather_packed(packed[j], gathered[j], async_op=True)         # (one kernel: no list of outputs to copy into)
            if args.gather == "sync":
                tw = time.perf_counter()
                works[j].wait()
                torch.cuda.current_stream(dev).synchronize()
                gather_wait[0] += time.perf_counter() - tw
                gather_wait[1] += 1
                works[j] = None

    def drain():
        for j in range(len(works)):
            if works[j] is not None:
                works[j].wait()
                works[j] = None

    solver = wl.make_solver(fam, device=local_rank)
    dt, per = timed(lambda: step(solver), args.steps, args.warmup, finish=drain)
    ms_per_step = dt / args.steps * 1e3
    value = world * B * args.steps / dt
    med = float(np.median(per))
    st, it, kkt = d_st.cpu().numpy(), d_it.cpu().numpy(), d_kkt.cpu().numpy()
    # (over ALL ranks' rows; the roofline pass below stays this rank's own launch statistics)
    allst = sharding.solve_stats_over_ranks(st, it, device=dev) if world > 1 else dict(converged_frac=float((st == 1).mean()), mean_iters=float(it.mean()), max_iters=int(it.max()))
    converged, mean_it_all, max_it = allst["converged_frac"], allst["mean_iters"], allst["max_iters"]
    mean_it = float(it.mean())
    kkt_max = sharding.max_over_ranks(float(kkt.max()), device=dev) if world > 1 else float(kkt.max())
    # the collective on its own: K gathers with nothing else on the machine, and a check of what arrived
    gather_info = None
    if gather:
        drain()
        barrier()
        tg = time.perf_counter()
        for _ in range(args.steps):
            sharding.gather_packed(packed[0], gathered[0])
        barrier()
        g_alone = (time.perf_counter() - tg) / args.steps
        gx, gst, git = sharding.unpack_rows(gathered[0][rank * B:(rank + 1) * B], d_out.shape[1])
        ok = bool(torch.equal(gx, d_outs[0]) or torch.equal(gx, d_outs[-1])) and bool(torch.equal(gst, d_st)) and bool(torch.equal(git, d_it))
        gather_info = dict(mode=args.gather, gather_ms_alone=(sharding.max_over_ranks(g_alone, device=dev) if world > 1 else g_alone) * 1e3,
                           bytes_per_rank=B * PW * 8, own_rows_round_trip=ok,
                           pipeline_abandoned=bool(solver.get_option("pipe_aborts")),       # (a persistent launch starved by the collective's workgroups would show here)
                           wait_ms_in_step=(gather_wait[0] / gather_wait[1] * 1e3) if gather_wait[1] else None)

    # ---- roofline: second pass over the same K steps with HIP events around every kernel launch
    ab = algorithmic_bytes(fam.N, fam.nx)
    solver.set_profiling(True)
    ric_ms = ric_n = stg_ms = stg_n = pipe_ms = pipe_n = wg_ms = wg_n = wg_it = 0.0
    pipe_stats = wg_stats = None
    torch.cuda.synchronize(dev)
    tp0 = time.perf_counter()
    for _ in range(args.steps):
        step(solver)
        pr = solver.get_profile()
        ric_ms += pr["riccati_ms"]; ric_n += pr["riccati_launches"]
        stg_ms += pr["stage_ms"]; stg_n += pr["stage_launches"]
        pp = solver.get_pipeline_profile()
        if pp["ran"]:                                    # the iterations of the tiles with many instances: ONE persistent launch (k_pipeline)
            pipe_ms += pp["ms"]; pipe_n += 1
            pipe_stats = {k: (float(v) if isinstance(v, (float, np.floating)) else v) for k, v in pp.items()}
        rp = solver.get_resident_profile()
        if rp["ran"]:                                    # the stragglers: ONE launch of k_solve_wg behind it
            wg_ms += rp["ms"]; wg_n += 1; wg_it += rp["instance_iterations"]
            wg_stats = {k: (float(v) if isinstance(v, (float, np.floating)) else v) for k, v in rp.items()}
    drain()
    torch.cuda.synchronize(dev)
    prof_ms_per_step = (time.perf_counter() - tp0) / args.steps * 1e3
    # ... and a third pass with ONE pair of events around the whole iteration loop (k_pipeline + k_solve_wg back to back, no marker between
    # them: the marker of the pass above costs the stream a few microseconds a production solve does not pay).  This span is the launch duration
    # of the roofline; the per-kernel split comes from the pass above.
    span_ms = span_n = 0.0
    if pipe_n and wg_n:
        solver.set_profiling(2)
        for _ in range(args.steps):
            step(solver)
            pp = solver.get_pipeline_profile()
            if pp["ran"] and solver.get_resident_profile()["ran"]:
                span_ms += pp["ms"]; span_n += 1
        drain()
        torch.cuda.synchronize(dev)
    solver.set_profiling(False)
    inst_iters = float(it.sum())                         # instance-iterations actually performed per step
    wg_iters = wg_it / max(wg_n, 1)                      # ... of which in k_solve_wg
    kern = {}
    for name, ms, n, bpi, iters in (("k_riccati", ric_ms, ric_n, ab["b_riccati"], inst_iters), ("k_stage", stg_ms, stg_n, ab["b_stage"], inst_iters),
                                    ("k_pipeline", pipe_ms, pipe_n, ab["b_iter"], inst_iters - wg_iters), ("k_solve_wg", wg_ms, wg_n, ab["b_iter"], wg_iters)):
        if n == 0:
            continue
        n_per_step = n / args.steps
        avg_us = ms / max(n, 1) * 1e3
        bytes_per_launch = bpi * iters / max(n_per_step, 1)           # instance-iterations of the launch x bytes per instance-iteration
        kern[name] = dict(avg_us=avg_us, launches_per_step=n_per_step, instance_iterations_per_launch=iters / max(n_per_step, 1),
                          bytes_per_launch=bytes_per_launch, gbs=bytes_per_launch / (avg_us * 1e-6) / 1e9, total_ms_per_step=ms / args.steps)
    loop = [k for k in ("k_pipeline", "k_solve_wg") if k in kern]
    if not loop:                                         # one launch per kernel and iteration (pipeline switched off): the heavier of the two
        loop = [max(kern, key=lambda k: kern[k]["total_ms_per_step"])]
    loop_bytes = sum(kern[k]["bytes_per_launch"] * kern[k]["launches_per_step"] for k in loop)
    loop_us_events_sum = sum(kern[k]["avg_us"] * kern[k]["launches_per_step"] for k in loop)
    loop_us = (span_ms / span_n * 1e3) if (span_n and len(loop) == 2) else loop_us_events_sum
    loop_gbs = loop_bytes / (loop_us * 1e-6) / 1e9
    copy_gbs = copy_bandwidth(solver) if rank == 0 else None
    traffic, traffic_src = None, "not measured (multi-GPU run or --no-traffic)"
    tags = {"k_pipeline": "k_pipeline<6", "k_solve_wg": "k_solve_wg<6", "k_stage": "k_stage<6, false", "k_riccati": "k_riccati<6"}
    if rank == 0 and world == 1 and not args.no_traffic:
        traffic, traffic_src = measure_traffic([tags[k] for k in loop])
    elif rank == 0:
        traffic, traffic_src = committed_traffic("+".join(loop)), "committed profiles/pmc_traffic.json"
    # bound: the roof the fraction is taken against (algorithmic bytes against the HBM peak).  limiter: what the counters say holds
    # the kernels back -- neither roof: one wavefront per SIMD, ~half of its cycles waiting on its own dependent instruction stream
    # (profiles/*_pmc_summary.txt: SQ_WAIT_ANY / SQ_WAVE_CYCLES ~ 0.5, VALU busy ~ 0.23, HBM traffic = 1.02 x the algorithmic bytes)
    roofline = dict(bound="hbm", limiter="issue/latency", kernel="+".join(loop), achieved=loop_gbs, peak=HBM_PEAK_GBS, unit="GB/s",
                    frac=loop_gbs / HBM_PEAK_GBS, traffic=traffic, traffic_source=traffic_src,
                    measured_copy_bw_gbs=copy_gbs, frac_vs_measured_copy_bw=(loop_gbs / copy_gbs) if copy_gbs else None,
                    copy_bw_note="the library's own 16-byte streaming copy (mpc_measure_copy_bandwidth), read + write bytes; the guide's float4 copy: 6290 GB/s",
                    avg_launch_us=loop_us, avg_launch_us_sum_of_kernel_spans=loop_us_events_sum, launches_per_step=len(loop), algorithmic_bytes_per_launch=loop_bytes,
                    note="a solve's iteration loop is one launch of each kernel listed, back to back on one stream; achieved = algorithmic "
                         "bytes of the instance-iterations performed (%d B each) / the duration of the loop, ONE pair of HIP events around both launches "
                         "(avg_launch_us; with a third event between the two kernels -- the per-kernel split -- the two spans add up to "
                         "avg_launch_us_sum_of_kernel_spans)" % ab["b_iter"],
                    kernels={k: dict(v, frac=v["gbs"] / HBM_PEAK_GBS) for k, v in kern.items()},
                    whole_step=dict(bytes_per_mpc_step=ab["b_io"] + mean_it * ab["b_iter"],
                                    gbs=value / world * (ab["b_io"] + mean_it * ab["b_iter"]) / 1e9,
                                    frac=value / world * (ab["b_io"] + mean_it * ab["b_iter"]) / 1e9 / HBM_PEAK_GBS),
                    profiled_ms_per_step=prof_ms_per_step, pipeline=pipe_stats, solve_wg=wg_stats)

    # ---- deterministic-work variant (exactly 20 iterations per instance, no early exit).  NOT the headline: an
    # instance that has reached the tolerance keeps iterating (steps accepted as they come), so only mean_iters / 20 of the
    # instance-iterations below are iterations a solve needs; the rest is deterministic filler that shows what the pipeline moves.
    fixed20 = None
    fsolver = None if args.headline_only else wl.make_solver(fam, device=local_rank, fixed_iters=20)
    if fsolver is not None:
        dtf, _ = timed(lambda: step(fsolver), args.steps, args.warmup, finish=drain)
    if fsolver is not None:
        fixed20 = dict(value=world * B * args.steps / dtf, ms_per_step=dtf / args.steps * 1e3,
                   hbm_frac_all_iterations=B * args.steps / dtf * (ab["b_io"] + 20 * ab["b_iter"]) / 1e9 / HBM_PEAK_GBS,
                   useful_iteration_share=mean_it / 20.0,
                   hbm_frac_useful_iterations=B * args.steps / dtf * (ab["b_io"] + mean_it * ab["b_iter"]) / 1e9 / HBM_PEAK_GBS,
                   note="20 iterations per instance regardless of convergence; %.1f of them are needed on average -- not headline credit" % mean_it)

    # ---- CPU baseline: the oracle on the host cores (rank 0, single-GPU run only)
    cpu_baseline = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        from oracle.binding import OracleSolver
        from oracle.nlp_numpy import NLPConfig
        osol = OracleSolver(NLPConfig(N=fam.N, nx=fam.nx, Q=fam.Q, R=fam.R))
        avail = HOST_CPUS
        xs, ps = x0[:4096], p[:4096]
        # SUSTAINED throughput per thread count: a warm batch (the team is created and placed, the output arrays are touched), then back-to-back
        # batches into the same arrays for ~0.8 s of wall clock (at least 5).  Why not the best single batch: the box's container runs under a
        # CPU-time quota (cgroup cpu.max, reported below) -- a team of 128 threads finishes one batch in 6.5 ms and is then throttled for the rest
        # of the 100 ms accounting period, every other batch takes 94 ms (tools/cpu_baseline_probe.py); round 4 took the minimum of two single
        # batches for the sweep and got 9 x the rate the same team sustained.
        def sustained(cores, seconds, min_batches=5):
            out = osol.solve_batch(xs, ps, nthreads=cores)
            ts = []
            t_start = time.perf_counter()
            while len(ts) < min_batches or time.perf_counter() - t_start < seconds:
                t0 = time.perf_counter()
                osol.solve_batch(xs, ps, nthreads=cores, out=out)
                ts.append(time.perf_counter() - t0)
            return len(ts) * len(xs) / (time.perf_counter() - t_start), float(np.median(ts)), len(ts), out
        sweep, sweep_median = {}, {}
        for cores in sorted({min(avail, c) for c in (8, 16, 32, 64, 128)}):   # (SMT siblings -- all of `avail` -- only add contention: 20 k steps/s at 256 threads)
            rate, med_t, _, _ = sustained(cores, 0.8)
            sweep[cores], sweep_median[cores] = rate, med_t
        cores = max(sweep, key=sweep.get)
        value_cpu, med_t, reps, ro = sustained(cores, 3.0)                  # ~3 s on the chosen thread count: the reported value
        agree = max(value_cpu, sweep[cores]) / max(1e-9, min(value_cpu, sweep[cores]))
        t0 = time.perf_counter()
        osol.solve_batch(xs[:1024], ps[:1024], nthreads=1)
        t_one = time.perf_counter() - t0
        n32 = min(32, avail)
        quota = None
        try:
            with open("/sys/fs/cgroup/cpu.max") as fh:
                q = fh.read().split()
                quota = None if q[0] == "max" else float(q[0]) / float(q[1])
        except Exception:
            pass
        model = "unknown"
        try:
            with open("/proc/cpuinfo") as fh:
                model = next(l.split(":", 1)[1].strip() for l in fh if l.startswith("model name"))
        except Exception:
            pass
        cpu_baseline = dict(value=value_cpu, unit="MPC steps/s", cores=cores, kind="port",
                            sample=f"{reps} x {len(xs)} instances of the same workload (N=30, nx=6) back to back, oracle/mpc_oracle.c, "
                                   f"OpenMP over instances, all converged={bool((ro['status'] == 1).all())}",
                            sample_short=f"{reps}x{len(xs)} instances back to back (sustained), oracle C port, OpenMP bound to cores",
                            single_thread_value=1024 / t_one, bound_32_threads_value=sweep.get(n32), bound_threads=n32,
                            omp_proc_bind=os.environ.get("OMP_PROC_BIND"), cpu_model=model, host_cpus=avail, cgroup_cpu_quota=quota,
                            median_batch_ms=med_t * 1e3, agrees_with_sweep_within=agree, stable=bool(agree <= 1.3),
                            sweep_steps_per_s={str(c): r for c, r in sorted(sweep.items())},
                            sweep_median_batch_ms={str(c): t * 1e3 for c, t in sorted(sweep_median.items())},
                            casadi_ipopt=casadi_probe(fam, x0, p, wl), published_casadi_ipopt=PUBLISHED_CASADI)

    # ---- BASELINE.json configurations 2 - 5 under the same clock (single-GPU run only; a few batches each, ~1 s in total)
    configs = None
    if rank == 0 and world == 1 and not args.headline_only:
        configs = other_configs(torch, wl, local_rank, dev, stream)

    # ---- the paths around the solve, single-GPU run only, a few hundred milliseconds in total
    other_paths = None
    if rank == 0 and world == 1 and not args.headline_only:
        other_paths = side_paths(torch, mpc_amd, fam, B, local_rank, traffic=not args.no_traffic)

    if rank == 0:
        out = dict(metric="MPC steps/sec (N=30, nx=6 nu=2) at batch=4096", value=value, unit="MPC steps/s", n_gpus=world,
                   steps=args.steps, warmup=args.warmup, ms_per_step=ms_per_step, higher_is_better=True, scaling="weak",
                   vs_baseline=None, dtype="f64", data="synthetic",
                   config=dict(workload="N=30 nx=6 nu=2 kinematic-bicycle lane-following (ZAM_Over-1_1 LF weights, dummy "
                                        "obstacle), batch=%d per GPU, solved to tol 1e-8 (max_iter 100)" % B,
                               batch_per_gpu=B, horizon=fam.N, nx=fam.nx, nu=2, parallelism="independent instances x%d" % world,
                               mode="converged", gpu=torch.cuda.get_device_name(dev)),
                   ms_per_step_median=med * 1e3, value_median_batch=world * B / med if world == 1 else None,
                   converged_frac=converged, mean_iters=mean_it_all, max_iters=max_it, kkt_max=kkt_max, gather=gather_info,
                   fixed20=fixed20, roofline=roofline, cpu_baseline=cpu_baseline, configs=configs, other_paths=other_paths)
        emit(out, args)
    if gather:
        dist.destroy_process_group()


def _g(x, n=5):
    """a float with n significant digits (None / ints / strings as they are)"""
    if isinstance(x, bool) or x is None or isinstance(x, (int, str)):
        return x
    try:
        return float(("%%.%dg" % n) % float(x))
    except (TypeError, ValueError):
        return x


def compact_line(out):
    """the ONE line of stdout: every contract key, the roofline and cpu_baseline objects and the other configurations, short enough for a
    2000-character tail (the driver keeps that much) -- the long form with every note goes to stderr / --detail"""
    r, cb = out["roofline"], out.get("cpu_baseline")
    line = dict(metric=out["metric"], value=_g(out["value"], 6), unit=out["unit"], n_gpus=out["n_gpus"], steps=out["steps"], warmup=out["warmup"],
                ms_per_step=_g(out["ms_per_step"]), higher_is_better=True, scaling=out["scaling"], vs_baseline=None, dtype=out["dtype"], data=out["data"],
                config=dict(workload="N=30 nx=6 nu=2 bicycle lane-following, batch=%d/GPU, tol 1e-8" % out["config"]["batch_per_gpu"], mode="converged",
                            gpu=out["config"]["gpu"]),
                converged_frac=_g(out["converged_frac"]), mean_iters=_g(out["mean_iters"]), max_iters=out["max_iters"],
                roofline=dict(bound=r["bound"], kernel=r["kernel"], achieved=_g(r["achieved"]), peak=r["peak"], unit=r["unit"], frac=_g(r["frac"], 4),
                              traffic=_g(r["traffic"]), avg_launch_us=_g(r["avg_launch_us"]), bytes_per_launch=_g(r["algorithmic_bytes_per_launch"]),
                              copy_bw=_g(r.get("measured_copy_bw_gbs"), 4), whole_step=_g(r["whole_step"]["frac"], 4), limiter=r["limiter"],
                              kernels={k: [_g(v["avg_us"], 4), _g(v["gbs"], 4), _g(v["frac"], 3)] for k, v in r["kernels"].items()}))
    if cb:
        line["cpu_baseline"] = dict(value=_g(cb["value"]), unit=cb["unit"], cores=cb["cores"], kind=cb["kind"], sample=cb["sample_short"],
                                    one_thread=_g(cb["single_thread_value"], 4), sweep={k: _g(v, 3) for k, v in cb["sweep_steps_per_s"].items()},
                                    stable=cb["stable"], cpu_quota=_g(cb.get("cgroup_cpu_quota"), 3))
    else:
        line["cpu_baseline"] = None
    if out.get("configs"):
        # [configuration, batch, ms per batch, steps/s, converged, mean iters, max iters, roofline frac of the loop, rescued]
        line["configs"] = [[c["config"].split(":")[0], c["batch"], _g(c["ms_per_batch"], 4), _g(c["steps_per_s"], 4), _g(c["converged_frac"], 4),
                            _g(c["mean_iters"], 4), c["max_iters"], _g(c["roofline_frac"], 3), c["rescued"]] if "error" not in c else ["error"]
                           for c in out["configs"]]
    if out.get("fixed20"):
        line["fixed20"] = [_g(out["fixed20"]["value"]), _g(out["fixed20"]["ms_per_step"], 4)]
    op = out.get("other_paths") or {}
    side = {}
    if "closed_loop" in op:
        side["closed_loop_ego_steps_s"] = _g(op["closed_loop"]["ego_steps_per_s"], 4)
    if "forces_sqp_step" in op:
        f = op["forces_sqp_step"]
        side["forces_sqp"] = [_g(f["solves_per_s"], 4), _g(f["roofline"]["frac"], 3), _g(f["roofline"]["traffic"], 4), _g(f["roofline"]["algorithmic_bytes_per_launch"], 4)]
    if "forces_closed_loop" in op:
        side["forces_loop_ego_steps_s"] = _g(op["forces_closed_loop"].get("ego_steps_per_s"), 4)
    if side:
        line["other_paths"] = side
    if out.get("gather"):
        line["gather"] = {k: _g(v, 4) for k, v in out["gather"].items()}
    return line


def emit(out, args):
    long_form = json.dumps(out)
    print("[bench detail] " + long_form, file=sys.stderr, flush=True)
    if getattr(args, "detail", None):
        with open(args.detail, "w") as fh:
            fh.write(long_form + "\n")
    print(json.dumps(compact_line(out), separators=(",", ":")), file=JSON_OUT, flush=True)


def other_configs(torch, wl, local_rank, dev, stream):
    """BASELINE.json configs[1..4]: ms per batch, steps/s, converged fraction, iterations and the roofline fraction of the iteration loop
    (algorithmic bytes of the instance-iterations performed / duration of the launches of the loop), device-resident buffers"""
    out = []

    def run(label, fams_rows, reps=5):
        """fams_rows: [(family, x0, p)] solved back to back per batch (one handle per family)"""
        items = []
        for fam, x0, p in fams_rows:
            B = len(x0)
            s = wl.make_solver(fam, device=local_rank)
            d = dict(s=s, fam=fam, B=B, x0=torch.from_numpy(x0).to(dev), p=torch.from_numpy(p).to(dev))
            d["out"] = torch.empty_like(d["x0"])
            d["st"] = torch.empty(B, dtype=torch.int32, device=dev)
            d["it"] = torch.empty(B, dtype=torch.int32, device=dev)
            d["kkt"] = torch.empty(B, dtype=torch.float64, device=dev)
            items.append(d)

        # several handles (configuration 5): concurrently, one host thread and one stream each, the collision-avoidance family first
        if len(items) > 1:
            from concurrent.futures import ThreadPoolExecutor
            items.sort(key=lambda d: 0 if "ca" in d["fam"].name else 1)
            pool = ThreadPoolExecutor(max_workers=len(items))
            for d in items:
                d["stream"] = torch.cuda.Stream(device=dev)

        def solve_one(d):
            torch.cuda.set_device(dev)
            d["s"].solve_device(d["B"], d["x0"].data_ptr(), d["p"].data_ptr(), d["out"].data_ptr(), d["st"].data_ptr(), d["it"].data_ptr(),
                                d["kkt"].data_ptr(), stream=d["stream"].cuda_stream if "stream" in d else stream)

        def once():
            if len(items) > 1:
                list(pool.map(solve_one, items))
            else:
                solve_one(items[0])
        once()
        torch.cuda.synchronize(dev)
        t0 = time.perf_counter()
        for _ in range(reps):
            once()
        torch.cuda.synchronize(dev)
        dt = (time.perf_counter() - t0) / reps
        loop_ms = loop_bytes = 0.0
        whole_call = False
        for d in items:                              # one profiled batch: launch durations of the iteration loop
            d["s"].set_profiling(True)
        if len(items) > 1:                           # (launch durations of the loop: the handles one after the other, nothing else on the machine)
            for d in items:
                solve_one(d)
        else:
            once()
        torch.cuda.synchronize(dev)
        paths = []
        for d in items:
            ab = algorithmic_bytes(d["fam"].N, d["fam"].nx)
            pr, pp, rp = d["s"].get_profile(), d["s"].get_pipeline_profile(), d["s"].get_resident_profile()
            # (which kernels served the family's last solve -- a pipeline launch that had to be abandoned, e.g. two persistent launches starving each
            #  other, leaves the handle on one launch per kernel for good: that must not go unnoticed)
            paths.append("%s: %s%s" % (d["fam"].name, "+".join((["k_pipeline"] if pp["ran"] else []) + (["k_solve_wg"] if rp["ran"] else [])) or "one launch per kernel",
                                       " (PIPELINE ABANDONED %d x)" % d["s"].get_option("pipe_aborts") if d["s"].get_option("pipe_aborts") else ""))
            loop_ms += pp["ms"] + rp["ms"] + pr["riccati_ms"] + pr["stage_ms"]
            loop_bytes += float(d["it"].sum().item()) * ab["b_iter"]
            d["s"].set_profiling(False)
            if d["s"].last_rescued() > 0:                # a second chance ran (inside the launch, or as further solves behind the call): the iteration counts
                whole_call = True                        # are accumulated over its levels, so the fraction is taken over the whole call
        Bt = sum(d["B"] for d in items)
        st = torch.cat([d["st"] for d in items]).cpu().numpy()
        it = torch.cat([d["it"] for d in items]).cpu().numpy()
        out.append(dict(config=label, batch=Bt, ms_per_batch=dt * 1e3, steps_per_s=Bt / dt, converged_frac=float((st == 1).mean()),
                        mean_iters=float(it.mean()), max_iters=int(it.max()), loop_launch_ms=None if whole_call else loop_ms,
                        # (algorithmic bytes of the instance-iterations performed / duration of the loop's launches; with a second chance: / the whole call)
                        roofline_frac=(loop_bytes / dt / 1e9 / HBM_PEAK_GBS) if whole_call else ((loop_bytes / (loop_ms * 1e-3) / 1e9 / HBM_PEAK_GBS) if loop_ms > 0 else None),
                        roofline_basis="whole call" if whole_call else "loop launches",
                        rescued=int(sum(d["s"].last_rescued() for d in items)), paths=paths))
    try:
        f2, f3, f4 = wl.FAMILIES["zamlf_n30_nx6"], wl.FAMILIES["zamca_n30_nx5"], wl.FAMILIES["usalf_n50_nx5"]
        run("2: N=30 nx=6 lane following, batch=256", [(f2,) + tuple(wl.batch(f2, 256))])
        run("3: ZAM_Over-1_1 collision avoidance (cold starts through the obstacle), N=30, batch=1024", [(f3,) + tuple(wl.batch(f3, 1024))])
        run("4: USA_Lanker weights, N=50, batch=4096", [(f4,) + tuple(wl.batch(f4, 4096))])
        shard = wl.mixed_shard(0, 8)
        run("5: mixed scenario sweep, shard 0 of 8 (4096 rows over the four families, one handle and one stream each, concurrently)",
            [(wl.FAMILIES[name], x0, p) for name, (rows, x0, p) in shard.items()], reps=3)
    except Exception as e:                               # the headline must not die of a side measurement
        out.append(dict(error=repr(e)))
    return out


def side_paths(torch, mpc_amd, fam, B, local_rank, traffic=True):
    other_paths = {}
    try:
        s5 = mpc_amd.BatchedMPCSolver(N_HORIZON, 5, Q=fam.Q, R=fam.R, device=local_rank)
        s5.set_bounds()                                                      # the reference's default limits
        L, Bc = 60, B
        k = np.arange(L)
        path = np.stack([k * 1.5 * np.cos(0.1), k * 1.5 * np.sin(0.1)], axis=1)
        rng = np.random.default_rng(0)
        init = np.tile([0.0, 0.0, 0.0, 15.0, 0.1], (Bc, 1))
        init[:, 1] += rng.uniform(-0.5, 0.5, Bc)
        init[:, 3] *= rng.uniform(0.9, 1.1, Bc)
        P_, O_ = np.tile(path, (Bc, 1, 1)), np.full((Bc, L), 0.1)
        s5.closed_loop(init, P_, O_, np.full(Bc, 15.0), L)        # warm-up at the full size (allocations)
        t0 = time.perf_counter()
        _, _, st_ = s5.closed_loop(init, P_, O_, np.full(Bc, 15.0), L)
        tcl = time.perf_counter() - t0
        other_paths["closed_loop"] = dict(ego_steps_per_s=Bc * L / tcl, ms_per_step_of_batch=tcl / L * 1e3, batch=Bc, steps=L, horizon=N_HORIZON,
                                          converged_frac=float((st_ == 1).mean()),
                                          note="mpc_closed_loop_batch, nx=5, host buffers in/out once per call (row f1)")
        sf, fstep, d_fl, d_it, lbf, ubf, hlf, huf = forces_setup(B, local_rank)
        dev = torch.device("cuda", local_rank)
        fstep()
        torch.cuda.synchronize(dev)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5):
            fstep()
        e1.record()
        e1.synchronize()
        tf = e0.elapsed_time(e1) * 1e-3 / 5
        fl_, it_ = d_fl.cpu().numpy(), d_it.cpu().numpy()
        # bytes a QP iteration has to move per instance: stage blocks (Hessian 7x7 sym, [A B] 5x7, gradient 7, defect 5, 10 rows of
        # inequality Jacobian 10x7 + residuals) written and read once, iterate + multipliers + slacks read and written
        nq = 10 * (28 + 35 + 7 + 5 + 70 + 10)
        ni = 10 * (7 + 5 + 2 * (14 + 10) * 2)
        b_it = 8 * 2 * (nq + ni)
        f_traffic, f_src = None, "not measured (--no-traffic)"
        if traffic and B == BATCH:
            f_traffic, f_src = measure_traffic(["k_forces_qp", "k_forces_stage"], child="forces", fallback="k_forces_qp+k_forces_stage")
        other_paths["forces_sqp_step"] = dict(solves_per_s=B / tf, ms_per_batch=tf * 1e3, batch=B, horizon=10, solved_frac=float((fl_ == 1).mean()),
                                              mean_qp_iterations=float(it_.mean()),
                                              roofline=dict(bound="hbm", achieved=B * float(it_.mean()) * b_it / tf / 1e9, peak=HBM_PEAK_GBS, unit="GB/s",
                                                            frac=B * float(it_.mean()) * b_it / tf / 1e9 / HBM_PEAK_GBS, traffic=f_traffic, traffic_source=f_src,
                                                            algorithmic_bytes_per_launch=B * float(it_.mean()) * b_it,
                                                            algorithmic_bytes_per_instance_iteration=b_it),
                                              note="mpc_forces_solve_batch_dev, device-resident buffers, HIP events (row f3)")
        Lf = 30
        kf = np.arange(Lf)
        pathf = np.stack([29.9948 + kf * 2.0 * np.cos(0.03495), -1.1501 + kf * 2.0 * np.sin(0.03495)], 1)
        initf = np.tile([29.9948, -1.1501, 0.0, 20.0, 0.03495], (B, 1))
        initf[:, 1] += np.random.default_rng(2).uniform(-0.3, 0.3, B)
        argsf = (np.tile(pathf, (B, 1, 1)), np.full((B, Lf), 0.03495), np.full(B, 20.0), Lf, lbf, ubf, hlf, huf)
        sf.forces_closed_loop(initf, *argsf)                     # warm-up at the full size: the grow-only scratch of the loop is allocated here
        t0 = time.perf_counter()
        _, _, flf = sf.forces_closed_loop(initf, *argsf)
        tfl = time.perf_counter() - t0
        other_paths["forces_closed_loop"] = dict(ego_steps_per_s=B * Lf / tfl, ms_per_step_of_batch=tfl / Lf * 1e3, batch=B, steps=Lf, horizon=10,
                                                 solved_frac=float((flf == 1).mean()),
                                                 note="mpc_forces_closed_loop_batch, host buffers in/out once per call (rows f1 + f3)")
    except Exception as e:      # never let the side measurements take the bench line down
        other_paths["error"] = repr(e)
    return other_paths


def run_mixed(args, torch, dist, mpc_amd, sharding, wl, world, rank, local_rank, dev, stream, timed):
    """BASELINE configuration 5: mixed scenario sweep, 4096 rows per GPU (32 768 over 8), one handle per family on every rank.  The
    four solves of a step run CONCURRENTLY -- one host thread and one HIP stream per handle, the collision-avoidance family first: its
    long tail is a latency chain of a few wavefronts, under which the lane-following families fill the machine -- then ONE all-gather
    of the padded result rows with their status and iteration counts (sharding.pack_rows)."""
    from concurrent.futures import ThreadPoolExecutor
    per_gpu = wl.MIXED_SHARD
    shard = wl.mixed_shard(rank, world, total=per_gpu * world)
    W = wl.MIXED_ROW_WIDTH
    PW = sharding.packed_width(W)
    d_res = torch.zeros(per_gpu, PW, dtype=torch.float64, device=dev)         # result rows in global order, padded to one width, + status, iters
    gathered = torch.empty((world * per_gpu, PW), dtype=torch.float64, device=dev) if world > 1 else None
    parts = []
    order = sorted(wl.MIXED_ORDER, key=lambda n: 0 if "ca" in n else 1)       # collision avoidance first
    for name in order:
        rows, x0, p = shard[name]
        fam = wl.FAMILIES[name]
        nB = len(rows)
        parts.append(dict(fam=fam, B=nB, solver=wl.make_solver(fam, device=local_rank), x0=torch.from_numpy(x0).to(dev), p=torch.from_numpy(p).to(dev),
                          out=torch.empty(nB, fam.n_w, dtype=torch.float64, device=dev), st=torch.empty(nB, dtype=torch.int32, device=dev),
                          it=torch.empty(nB, dtype=torch.int32, device=dev), local=torch.from_numpy(rows - rank * per_gpu).to(dev),
                          stream=torch.cuda.Stream(device=dev)))
    serial = os.environ.get("MPC_BENCH_MIXED_SERIAL") == "1"                   # (A/B: the four solves back to back on one stream)
    pool = ThreadPoolExecutor(max_workers=len(parts))

    def solve_part(q):
        torch.cuda.set_device(dev)
        q["solver"].solve_device(q["B"], q["x0"].data_ptr(), q["p"].data_ptr(), q["out"].data_ptr(), q["st"].data_ptr(), q["it"].data_ptr(), 0,
                                 stream=stream if serial else q["stream"].cuda_stream)

    def step():
        if serial:
            for q in parts:
                solve_part(q)
        else:
            list(pool.map(solve_part, parts))                                # (every solve_device returns after its own stream has drained)
        for q in parts:
            n_w = q["fam"].n_w
            d_res[q["local"], :n_w] = q["out"]
            d_res[q["local"], W] = q["st"].double()
            d_res[q["local"], W + 1] = q["it"].double()
        if world > 1:
            sharding.gather_packed(d_res, gathered)

    dt, per = timed(step, args.steps, args.warmup)
    value = world * per_gpu * args.steps / dt
    fams = {}
    for q in parts:
        st, it = q["st"].cpu().numpy(), q["it"].cpu().numpy()
        a = sharding.solve_stats_over_ranks(st, it, device=dev) if world > 1 else dict(converged_frac=float((st == 1).mean()), mean_iters=float(it.mean()), max_iters=int(it.max()))
        fams[q["fam"].name] = [q["B"], _g(a["converged_frac"], 4), _g(a["mean_iters"], 4), a["max_iters"], int(q["solver"].last_rescued())]
    if rank == 0:
        out = dict(metric="MPC steps/sec, mixed scenario sweep (BASELINE configuration 5)", value=_g(value, 6), unit="MPC steps/s", n_gpus=world,
                   steps=args.steps, warmup=args.warmup, ms_per_step=_g(dt / args.steps * 1e3), ms_per_step_median=_g(float(np.median(per)) * 1e3),
                   higher_is_better=True, scaling="weak", vs_baseline=None, dtype="f64", data="synthetic",
                   config=dict(workload="mixed scenario sweep: %d rows per GPU dealt row by row over %s, one handle per family, tol 1e-8" % (per_gpu, ", ".join(wl.MIXED_ORDER)),
                               rows_total=per_gpu * world, parallelism="contiguous shards x%d; four handles on four streams; one packed all-gather per step" % world,
                               handles="serial" if serial else "concurrent", gpu=torch.cuda.get_device_name(dev)),
                   families=fams, families_cols="rows/GPU, converged, mean iters, max iters, rescued (rank 0)")
        print(json.dumps(out, separators=(",", ":")), file=JSON_OUT, flush=True)
    pool.shutdown()
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
