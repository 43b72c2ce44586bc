#!/usr/bin/env python3
"""bench.py -- BASELINE.json metric: DDP (MS-iLQR) iterations/s at batch x horizon = 4096 x 200,
SE3 exact tracking, on N GPUs of one node (one process per GPU, weak scaling: 4096 trajectories
per GPU, no collective inside the solve; one RCCL all_gather of costs / controls afterwards).

A "step" is one batch-iteration: every one of the B*N knot-iterations of the batch advanced once
(backward Riccati sweep + closed-loop rollout + re-linearisation).  Inputs are
resident in HBM before the timed region.  W warm-up steps, then R timed regions of exactly K steps
each (repeated timed regions, median reported), every region bracketed by
barrier + synchronize pairs and reduced with MAX over ranks; `value` = N_gpus * K / median region.

Launch: `python bench.py --gpus N` spawns N rank processes itself (fresh children, created before
anything touches a GPU); under `python -m torch.distributed.run --nproc-per-node N bench.py --gpus N`
the ranks already exist (RANK / LOCAL_RANK / WORLD_SIZE in the environment) and are used as they are.
The reference's counterpart of this fan-out is joblib.Parallel over independent initial conditions
(visualization/perturb_all_compute.py:240-250).
"""
import argparse
import json
import os
import socket
import statistics
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8 TB/s spec
HBM_ACHIEVABLE_GBS = 6290.0    # same guide: measured float4 copy
FP64_VALU_PEAK_TFLOPS = 78.6   # vendor fp64 vector peak
ALG_BYTES_PER_KNOT_ITER = 448  # read+write of (q 4x4, xi 6, u 6) in fp64
ALG_FLOPS_PER_KNOT_ITER = 25e3  # dense count, SURVEY.md §8d (secondary figure)
METRIC = "DDP iterations/sec at batch x horizon = 4096 x 200 (SE3 tracking)"


# ---------------------------------------------------------------------------------------------------
# launcher (no torch import here: the parent of spawned ranks must never initialise a GPU)
# ---------------------------------------------------------------------------------------------------
def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--repeats", type=int, default=5, help="timed regions of --steps steps each (median reported)")
    ap.add_argument("--batch", type=int, default=4096)
    ap.add_argument("--horizon", type=int, default=200)
    ap.add_argument("--schedule", choices=["auto", "split"], default="auto")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=15.0, help="target wall time of the CPU baseline sample")
    ap.add_argument("--dry-run", action="store_true",
                    help="launcher / collective plumbing only (gloo on CPU, no solver): used by the CPU tests")
    return ap.parse_args(argv)


def launch_plan(gpus, env):
    """What this process is: ('single', 0, 1) one rank, no rendezvous; ('worker', rank, world) a rank started by
    a launcher; ('spawn', 0, gpus) the parent that has to start `gpus` ranks; ('error', msg, 0)."""
    if gpus < 1:
        return ("error", "--gpus must be >= 1", 0)
    if "WORLD_SIZE" in env:
        world = int(env["WORLD_SIZE"])
        if world != gpus:
            return ("error", "--gpus %d but WORLD_SIZE=%d: launch one rank per GPU" % (gpus, world), 0)
        return ("worker" if world > 1 else "single", int(env.get("RANK", "0")), world)
    return ("spawn", 0, gpus) if gpus > 1 else ("single", 0, 1)


def rank_env(rank, world, port, base=None):
    """Environment of rank `rank` of `world` on this node: one device each, rendezvous on 127.0.0.1."""
    env = dict(os.environ if base is None else base)
    env.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), LOCAL_WORLD_SIZE=str(world),
               MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    return env


def free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def spawn_ranks(world, argv):
    """Start `world` fresh child processes of this script (rank r -> device r) and wait for them.  Rank 0's
    stdout (the JSON line) passes through; returns the worst exit code."""
    port = free_port()
    procs = [subprocess.Popen([sys.executable, os.path.abspath(__file__)] + list(argv), env=rank_env(r, world, port))
             for r in range(world)]
    rc = 0
    for p in procs:
        rc = max(rc, abs(p.wait()))
    return rc


# ---------------------------------------------------------------------------------------------------
# measurement
# ---------------------------------------------------------------------------------------------------
def measured_traffic(kernel="k_backward"):
    """HBM bytes per launch of the dominant kernel from the newest committed rocprofv3 PMC passes
    (profiles/*_hbm_traffic_pmc.json: FETCH_SIZE and WRITE_SIZE collected in separate --pmc runs of this
    same command, FETCH_SIZE doubled as MI355X_MICROARCH.md §HBM prescribes).  PMC counters cannot be read
    from inside the timed process, so this is the figure of the profiled run, or None."""
    import glob
    best = None
    for f in glob.glob(os.path.join(ROOT, "profiles", "*_hbm_traffic_pmc.json")):
        try:
            d = json.load(open(f))
        except Exception:
            continue
        key = (d.get("captured", ""), os.path.basename(f))  # newest capture; files without a stamp sort first
        if best is None or key > best[0]:
            best = (key, d, f)
    if best is None:
        return None, None
    for name, v in best[1].get("kernels", {}).items():
        if kernel in name:
            return v.get("hbm_bytes_per_launch_fetch_doubled"), os.path.basename(best[2])
    return None, None


def host_cpu_share():
    """Threads worth starting on this host: the affinity mask, capped by the cgroup CPU quota when there is one
    (a GPU box exposes all its hardware threads but grants a share of them)."""
    try:
        n = len(os.sched_getaffinity(0))
    except Exception:
        n = os.cpu_count() or 1
    quota = None
    try:  # cgroup v2
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()
        if q != "max":
            quota = float(q) / float(per)
    except Exception:
        try:  # cgroup v1
            q = float(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            per = float(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0:
                quota = q / per
        except Exception:
            pass
    if quota:
        n = max(1, min(n, int(quota + 0.5)))
    return n


def cpu_baseline(prob, x0_q, x0_xi, us0, seconds):
    """The CPU oracle (oracle/tolg_oracle.c: the parity-checked port of the reference algorithm) on this
    host's cores: OpenMP over trajectories, one workspace per thread.  Bounded sample of the same workload:
    all B trajectories, as many iterations as fit in about `seconds`; thread count and iteration count come
    from short calibration probes, so the sample stays bounded whatever the host's real CPU share is."""
    from oracle import bridge as ob
    op = ob.OracleProblem(prob.kind, prob.J, prob.dt, prob.Q, prob.R, prob.P, prob.q_ref, prob.xi_ref)
    B = x0_q.shape[0]
    # one thread, 4 trajectories x 20 iterations: the single-core rate (what one reference fit corresponds to)
    nb1, it1 = min(4, B), 20
    t0 = time.perf_counter()
    ob.fit_batch(op, x0_q[:nb1], x0_xi[:nb1], us0[:nb1], mode="ms", max_iter=it1, threads=1)
    r1 = nb1 * it1 / (time.perf_counter() - t0)                      # trajectory-iterations/s on one core
    # thread count: the visible CPUs are not necessarily the CPU share of this job (a GPU box shows 256 hardware
    # threads and grants fewer): calibrate a few counts on a 512-trajectory, 2-iteration probe and keep the fastest
    share = host_cpu_share()
    nbc = min(B, 512)
    best = (0.0, 1)
    for th in sorted({min(share, c) for c in (8, 16, 32, 64, 128)} | {share}):
        t0 = time.perf_counter()
        ob.fit_batch(op, x0_q[:nbc], x0_xi[:nbc], us0[:nbc], mode="ms", max_iter=2, threads=th)
        best = max(best, (nbc * 2 / (time.perf_counter() - t0), th))
    cores = best[1]
    per_iter = B / best[0]
    iters = int(max(3, min(200, seconds / per_iter)))
    t0 = time.perf_counter()
    r = ob.fit_batch(op, x0_q, x0_xi, us0, mode="ms", max_iter=iters, threads=cores)
    dt = time.perf_counter() - t0
    used = r["threads"]
    return {"value": iters / dt, "unit": "batch-iterations/s", "cores": used, "kind": "port",
            "sample": "all %d trajectories x N=%d, %d iterations (%.1f s), OpenMP over trajectories on %d threads"
                      % (B, prob.N, iters, dt, used),
            "trajectory_iterations_per_s": B * iters / dt,
            "single_core_trajectory_iterations_per_s": r1,
            "parallel_efficiency": (B * iters / dt) / (r1 * used)}


def run_dry(rank, world):
    """Plumbing check on CPU (gloo): rendezvous, barrier, MAX-reduced region time, the final gather through
    sharding.gather_results, and the JSON line -- everything of the multi-rank path except the solver."""
    import torch
    import torch.distributed as dist
    from trajectory_optimization_matrix_lie_groups_amd import sharding
    if world > 1:
        dist.init_process_group("gloo", rank=rank, world_size=world)
        dist.barrier()
    t = torch.tensor([0.001 * (rank + 1)], dtype=torch.float64)
    if world > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    Bg = 4 * world + 1                                    # uneven shards on purpose
    lo, hi = sharding.shard_bounds(Bg, world, rank)
    local = torch.arange(lo, hi, dtype=torch.float64).reshape(-1, 1) * torch.ones(1, 3, dtype=torch.float64)
    full = sharding.gather_results(local, Bg)
    ok = bool(torch.equal(full[:, 0], torch.arange(Bg, dtype=torch.float64)))
    if rank == 0:
        print(json.dumps({"metric": METRIC, "dry_run": True, "n_gpus": world, "region_s_max": float(t.item()),
                          "gather_ok": ok, "gathered_rows": int(full.shape[0])}), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    return 0 if ok else 1


def run_rank(args, rank, world):
    import torch
    import torch.distributed as dist
    from trajectory_optimization_matrix_lie_groups_amd import BatchedTrackingILQR, sharding, workloads

    local_rank = int(os.environ.get("LOCAL_RANK", str(rank)))
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        dist.init_process_group("nccl", device_id=dev)

    B, N, K, W, R = args.batch, args.horizon, args.steps, args.warmup, max(1, args.repeats)
    # each rank owns an independent shard of the (weak-scaled) batch: different seeded perturbations
    prob, x0_q, x0_xi, us0 = workloads.se3_tracking(B, N=N, seed=workloads.SEED + rank)
    solver = BatchedTrackingILQR(prob, B, device=dev)
    x0_q_d = torch.as_tensor(x0_q, device=dev); x0_xi_d = torch.as_tensor(x0_xi, device=dev)
    us0_d = torch.as_tensor(us0, device=dev)

    def barrier():
        torch.cuda.synchronize(dev)
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize(dev)

    total = W + R * K
    res = solver.solve_begin(x0_q_d, x0_xi_d, us0_d, mode="ms", n_iterations=total, tol_grad_norm=0.0, tol_d_norm=0.0,
                             schedule=args.schedule)
    solver.solve_iterate(W)
    regions, kern = [], []
    for _ in range(R):
        solver.enable_timing(True)
        barrier()
        t0 = time.perf_counter()
        solver.solve_iterate(K)
        barrier()
        t1 = time.perf_counter()
        ms_b, ms_r, ms_l, n_b = solver.kernel_time(reset=True)
        solver.enable_timing(False)
        el = torch.tensor([t1 - t0], dtype=torch.float64, device=dev)
        if world > 1:
            dist.all_reduce(el, op=dist.ReduceOp.MAX)
        regions.append(float(el.item()))
        kern.append((ms_b / max(n_b, 1), ms_r / max(n_b, 1), ms_l / max(n_b, 1)))
    res = solver.solve_end()
    torch.cuda.synchronize(dev)
    # final gather of costs / controls over RCCL (outside the timed solve, reported separately)
    gather_ms, gather_err = None, None
    if world > 1:
        try:
            Jf = res.J_hist[:, total - 1].contiguous().reshape(-1, 1)
            torch.cuda.synchronize(dev)
            g0 = time.perf_counter()
            Jall = sharding.gather_results(Jf, B * world)
            Uall = sharding.gather_results(res.us, B * world)
            torch.cuda.synchronize(dev)
            gather_ms = (time.perf_counter() - g0) * 1e3
            assert Jall.shape[0] == B * world and Uall.shape[0] == B * world
        except Exception as e:  # the timed figure above stands on its own; say what happened to the gather
            gather_err = "%s: %s" % (type(e).__name__, e)
    finite = bool(torch.isfinite(res.J_hist[:, :total]).all().item())
    clean = bool((res.status == 0).all().item())

    if rank == 0:
        med = statistics.median(regions)
        imed = min(range(R), key=lambda i: abs(regions[i] - med))
        kb, kr, kl = kern[imed]
        value = world * K / med
        ms_step = med / K * 1e3
        alg_bytes = ALG_BYTES_PER_KNOT_ITER * B * N
        dominant, t_dom = max((("k_backward", kb), ("k_rollout_lin" if kl == 0.0 else "k_rollout", kr),
                               ("k_linearize", kl)), key=lambda kv: kv[1])
        achieved = alg_bytes / (t_dom * 1e-3) / 1e9 if t_dom > 0 else None
        step_gbs = alg_bytes / (ms_step * 1e-3) / 1e9
        traffic, traffic_src = measured_traffic(dominant) if (B == 4096 and N == 200) else (None, None)
        line = {
            "metric": METRIC,
            "value": value, "unit": "batch-iterations/s", "n_gpus": world, "steps": K, "warmup": W,
            "ms_per_step": ms_step, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": "SE3 exact tracking, MS-iLQR (line_search=False, rollout=nonlinear), "
                                   "B=%d trajectories per GPU x N=%d knots, path_se3_generate_sine_2, "
                                   "seeded perturbed initial states" % (B, N),
                       "batch_per_gpu": B, "horizon": N, "global_batch": B * world, "schedule": args.schedule,
                       "trajectory_iterations_per_s": value * B, "all_finite": finite, "all_status_ok": clean,
                       "timed_regions": {"repeats": R, "steps_each": K, "reported": "median",
                                         "ms_per_step": [r / K * 1e3 for r in regions],
                                         "min_ms_per_step": min(regions) / K * 1e3,
                                         "max_ms_per_step": max(regions) / K * 1e3},
                       "kernel_ms_per_step": {"backward": kb, "rollout": kr, "linearize": kl},
                       "final_gather_ms": gather_ms, **({"final_gather_error": gather_err} if gather_err else {})},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": (achieved / HBM_PEAK_GBS) if achieved else None, "traffic": traffic,
                         "traffic_source": traffic_src,
                         "kernel": dominant, "algorithmic_bytes_per_launch": alg_bytes, "kernel_avg_ms": t_dom,
                         # SURVEY §8d's own formula: 448 B x B x N per batch-iteration over the WHOLE step
                         "achieved_step": step_gbs, "frac_step": step_gbs / HBM_PEAK_GBS,
                         "peak_achievable": HBM_ACHIEVABLE_GBS,
                         "frac_achievable": (achieved / HBM_ACHIEVABLE_GBS) if achieved else None,
                         "frac_step_achievable": step_gbs / HBM_ACHIEVABLE_GBS,
                         "fp64_frac_step": ALG_FLOPS_PER_KNOT_ITER * B * N / (ms_step * 1e-3) / 1e12 / FP64_VALU_PEAK_TFLOPS,
                         "note": "fp64 VALU / dependent-issue bound by construction (SURVEY §8d): fp64_frac_step = "
                                 "25 kflop per knot-iteration (dense count) over the whole step against %.1f TFLOP/s"
                                 % FP64_VALU_PEAK_TFLOPS},
        }
        if not args.no_cpu_baseline and world == 1:
            line["cpu_baseline"] = cpu_baseline(prob, x0_q, x0_xi, us0, args.cpu_seconds)
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    return 0


def main(argv=None):
    argv = sys.argv[1:] if argv is None else argv
    args = parse_args(argv)
    kind, rank, world = launch_plan(args.gpus, os.environ)
    if kind == "error":
        print("bench.py: " + rank, file=sys.stderr)
        return 2
    if kind == "spawn":
        return spawn_ranks(world, argv)
    if args.dry_run:
        return run_dry(rank, world)
    return run_rank(args, rank, world)


if __name__ == "__main__":
    sys.exit(main())
