#!/usr/bin/env python3
"""bench.py -- BASELINE.json metric: DDP (MS-iLQR) iterations/s at batch x horizon = 4096 x 200,
SE3 exact tracking, on N GPUs of one node (one process per GPU, no collective inside the solve; one
RCCL all_gather of costs / controls afterwards).

A "step" is one batch-iteration: every one of the B*N knot-iterations of the batch advanced once
(backward Riccati sweep + closed-loop rollout + re-linearisation).  Inputs are
resident in HBM before the timed region.  W warm-up steps, then R timed regions of exactly K steps
each (repeated timed regions, median reported), every region bracketed by
barrier + synchronize pairs and reduced with MAX over ranks.

Scaling (`--scaling`): "weak" (default, what the driver's `--gpus N` measures) gives every rank its own
4096 trajectories, `value` = N * K / median region (in 4096-batch-iterations/s); "strong" splits ONE
global batch (`--batch`, 4096) contiguously over the ranks, `value` = K / median region -- the fixed
4096 x 200 problem of the metric name on 1/2/4/8 GPUs.  The JSON line says which.

Secondary lines (never the headline): `--mode ss`, `--line-search`, `--workload drone400` (BASELINE config 5:
drone racing, 8192 trajectories, N = 400; defaults to strong scaling: 1024 per GPU on 8), `--horizon`.

Launch: `python bench.py --gpus N` spawns N rank processes itself (fresh children, created before
anything touches a GPU) and supervises them: the first rank that fails takes the others down with it,
and the whole job has a deadline; under `python -m torch.distributed.run --nproc-per-node N bench.py
--gpus N` the ranks already exist (RANK / LOCAL_RANK / WORLD_SIZE in the environment) and are used as
they are.  The reference's counterpart of this fan-out is joblib.Parallel over independent initial
conditions (visualization/perturb_all_compute.py:240-250).
"""
import argparse
import hashlib
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
ALG_FLOPS_PER_KNOT_ITER = 25e3  # dense count, SURVEY.md §8d (secondary figure)
METRIC = "DDP iterations/sec at batch x horizon = 4096 x 200 (SE3 tracking)"
RENDEZVOUS_TIMEOUT_S = 180     # init_process_group / collectives: a missing rank fails the others, it does not hang them


def alg_bytes_per_knot_iter(m):
    """read + write of (q 4x4, xi 6, u m) in fp64: 448 B for m = 6, 416 B for m = 4."""
    return 2 * 8 * (16 + 6 + m)


# ---------------------------------------------------------------------------------------------------
# launcher (no torch import here: the parent of spawned ranks must never initialise a GPU)
# ---------------------------------------------------------------------------------------------------
def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--repeats", type=int, default=200,
                    help="timed regions of --steps steps each (median reported); the default keeps the GPU busy for ~2.5 s at 4096 x 200")
    ap.add_argument("--fresh-regions", type=int, default=8,
                    help="headline only: additional regions that time iterations W..W+K of a FRESH solve each (the early-iteration "
                         "regime; reported as config.fresh_solve, never as `value`); 0 = off")
    ap.add_argument("--batch", type=int, default=None, help="trajectories per GPU (weak) or in all (strong); default 4096 (se3) / 8192 (drone400)")
    ap.add_argument("--horizon", type=int, default=None, help="default 200 (se3) / 400 (drone400)")
    ap.add_argument("--workload", choices=["se3", "drone400", "so3", "al1024", "pendulum"], default="se3",
                    help="se3: the metric's workload; drone400 / so3 / al1024: BASELINE configs 5 / 2 / 4 at their stated sizes "
                         "(secondary lines; so3 defaults to the SS solver its script uses, al1024 times inner MS iterations with "
                         "the AL terms of the first outer iteration attached; pendulum: Pendulum3dDyanmics swing-up, 80 knots -- the "
                         "model with a state-dependent input matrix, on the general backward sweep)")
    ap.add_argument("--rollout", choices=["nonlinear", "linear"], default="nonlinear",
                    help="linear: the reference constructors' own default (traopt_controller.py:1837-1838, :2359-2363), which every "
                         "script overrides; secondary line")
    ap.add_argument("--inertia", choices=["diag", "dense"], default="diag",
                    help="dense: full 3x3 inertia blocks (the general backward sweep k_backward instead of k_backward3); secondary line")
    ap.add_argument("--scaling", choices=["weak", "strong"], default=None, help="default weak (se3) / strong (drone400)")
    ap.add_argument("--mode", choices=["ms", "ss"], default=None, help="default ms (ss for --workload so3)")
    ap.add_argument("--line-search", action="store_true")
    ap.add_argument("--schedule", choices=["auto", "split"], default="auto")
    ap.add_argument("--r-scale", type=float, default=None,
                    help="input weight R = r I of the se3 / drone400 workloads (default 1e-5 / 1e-3); not the metric's workload when set")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=15.0, help="target wall time of the CPU baseline sample")
    ap.add_argument("--allow-lib-override", action="store_true",
                    help="accept TOLG_HIP_LIB (another build of the C ABI, A/B timing); refused otherwise")
    ap.add_argument("--deadline", type=float, default=1500.0, help="seconds after which the launcher kills its ranks")
    ap.add_argument("--rccl-selftest", action="store_true",
                    help="one rank only: take the multi-rank code path anyway (RCCL process group of size 1, barriers, the MAX "
                         "all-reduce, the final gather on the device) -- what a one-GPU box can verify of --gpus N")
    ap.add_argument("--dry-run", action="store_true",
                    help="launcher / collective plumbing only (gloo on CPU, no solver): used by the CPU tests")
    ap.add_argument("--dry-run-fail-rank", type=int, default=-1,
                    help="(tests) with --dry-run: this rank exits 3 before the rendezvous")
    a = ap.parse_args(argv)
    if a.batch is None:
        a.batch = {"se3": 4096, "drone400": 8192, "so3": 1, "al1024": 1024, "pendulum": 4096}[a.workload]
    if a.horizon is None:
        a.horizon = {"se3": 200, "drone400": 400, "so3": 100, "al1024": 200, "pendulum": 80}[a.workload]
    if a.scaling is None:
        a.scaling = "strong" if a.workload == "drone400" else "weak"
    if a.mode is None:
        a.mode = "ss" if a.workload == "so3" else "ms"
    return a


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
    """A port nobody listens on right now.  (Rank 0 binds it a moment later; a collision in between shows up as a
    failed rank 0, which the supervisor turns into a prompt non-zero exit -- not a hang.)"""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def supervise(procs, deadline_s, poll_s=0.2):
    """Wait for the rank processes.  The first non-zero exit (or the deadline) terminates the others -- ranks blocked
    in a rendezvous or a collective whose peer died would otherwise wait forever -- and becomes the return code."""
    t_end = time.monotonic() + deadline_s
    rc = 0
    alive = list(procs)
    while alive:
        for p in list(alive):
            r = p.poll()
            if r is None:
                continue
            alive.remove(p)
            if r != 0 and rc == 0:
                rc = abs(r) or 1
        if alive and (rc != 0 or time.monotonic() > t_end):
            if rc == 0:
                rc = 124  # deadline
            for p in alive:
                p.terminate()
            t_kill = time.monotonic() + 10.0
            while any(p.poll() is None for p in alive) and time.monotonic() < t_kill:
                time.sleep(poll_s)
            for p in alive:
                if p.poll() is None:
                    p.kill()
            for p in alive:
                p.wait()
            break
        if alive:
            time.sleep(poll_s)
    return rc


def spawn_ranks(world, argv, deadline_s):
    """Start `world` fresh child processes of this script (rank r -> device r) and supervise them.  Rank 0's
    stdout (the JSON line) passes through; returns the first failure's exit code (0 if every rank succeeded)."""
    port = free_port()
    procs = [subprocess.Popen([sys.executable, os.path.abspath(__file__)] + list(argv), env=rank_env(r, world, port))
             for r in range(world)]
    return supervise(procs, deadline_s)


# ---------------------------------------------------------------------------------------------------
# what ran: library path, version, source / build identity
# ---------------------------------------------------------------------------------------------------
def _sha16(paths):
    h = hashlib.sha256()
    for p in paths:
        with open(p, "rb") as f:
            h.update(f.read())
    return h.hexdigest()[:16]


def library_identity(allow_override):
    """Path and version string of the C-ABI library this process uses, the hash of the sources it should have been
    built from, and the git head recorded at build time (the GPU box has no .git).  TOLG_HIP_LIB is refused unless
    the caller asked for it: a benchmark line must not silently come from another build."""
    from trajectory_optimization_matrix_lie_groups_amd import _build, _capi
    override = os.environ.get("TOLG_HIP_LIB")
    if override and not allow_override:
        raise SystemExit("bench.py: TOLG_HIP_LIB=%s is set; pass --allow-lib-override to time another build" % override)
    lib = _capi.load()
    info = {"lib_path": os.path.relpath(_build.lib_path(), ROOT) if not override else override,
            "lib_override": bool(override),
            "tolg_version": lib.tolg_version().decode(),
            "lib_sha256_16": _sha16([_build.lib_path()])}
    bi = _build.build_info()
    info["git_head"] = bi.get("git_head")
    info["source_sha256_16"] = _build.source_hash()
    info["lib_built_from_these_sources"] = (bi.get("source_sha256_16") == info["source_sha256_16"]) if not override else None
    return info


# ---------------------------------------------------------------------------------------------------
# measurement
# ---------------------------------------------------------------------------------------------------
def fp64_issue_fraction(kernel, t_ms):
    """Fraction of the chip's fp64 vector ISSUE rate the kernel ran at: fp64 instructions it executed (SQ_INSTS_VALU_{FMA,
    MUL,ADD,TRANS}_F64 of the newest committed profiles/*_sq_mix.json, per wave x waves per launch) over what 1 024 SIMDs
    issue in its measured time at one wave-instruction per four cycles of the 2.4 GHz peak clock.  None without a profile."""
    import glob
    loaded = []
    for f in glob.glob(os.path.join(ROOT, "profiles", "*_sq_mix.json")):
        try:
            d = json.load(open(f))
        except Exception:
            continue
        loaded.append(((d.get("captured", ""), os.path.basename(f)), f, d))   # newest capture; files without a stamp sort first
    for _key, f, d in sorted(loaded, reverse=True):
        name, v = _pick_kernel(d.get("kernels", {}), kernel)
        if name and "SQ_WAVES" in v and t_ms > 0:
            n = sum(v.get("SQ_INSTS_VALU_%s_F64" % k, 0.0) for k in ("FMA", "MUL", "ADD", "TRANS"))   # summed over the waves of a launch
            peak = 1024 * 2.4e9 / 4.0
            return {"fp64_instructions_per_launch": n, "issue_rate": n / (t_ms * 1e-3), "peak_issue_rate": peak,
                    "frac": n / (t_ms * 1e-3) / peak, "kernel": name.split("::")[-1], "source": os.path.basename(f)}
    return None


def _pick_kernel(ks, kernel):
    """The profile entry of `kernel`.  The backward sweep is two kernels since round 4 -- k_backward3<.., true> (the fast sweep:
    every launch of a converged solve) and <.., false> (the full kernel: the first sweep of a solve, and the mostly empty redo
    launch behind every fast one); the headline's dominant kernel is the fast one."""
    cands = [(n, v) for n, v in ks.items() if kernel in n]
    if kernel == "k_backward3":
        fast = [(n, v) for n, v in cands if n.rstrip().endswith("true>")]
        cands = fast or cands
    return cands[0] if cands else (None, None)


def executed_fp64_fraction(ms_step):
    """fp64 flops the step's launches EXECUTED (SQ_INSTS_VALU_{FMA x 2, MUL, ADD, TRANS}_F64 of the newest committed
    profiles/*_sq_mix.json, wave instructions x 64 lanes, idle lanes included, k_backward3 + k_rollout_lin per launch) over the
    step time, as a fraction of the fp64 vector peak.  The nominal 25 kflop per knot-iteration of SURVEY 8d counts a dense
    formulation; this counts what ran.  None without a profile."""
    import glob
    loaded = []
    for f in glob.glob(os.path.join(ROOT, "profiles", "*_sq_mix.json")):
        try:
            d = json.load(open(f))
        except Exception:
            continue
        loaded.append(((d.get("captured", ""), os.path.basename(f)), f, d))
    for _key, f, d in sorted(loaded, reverse=True):
        ks = d.get("kernels", {})
        flops, seen = 0.0, []
        for kern in ("k_backward3", "k_rollout_lin"):
            name, v = _pick_kernel(ks, kern)
            if name and "SQ_INSTS_VALU_FMA_F64" in v:
                flops += 64.0 * (2.0 * v["SQ_INSTS_VALU_FMA_F64"] + v.get("SQ_INSTS_VALU_MUL_F64", 0.0) + v.get("SQ_INSTS_VALU_ADD_F64", 0.0)
                                 + v.get("SQ_INSTS_VALU_TRANS_F64", 0.0))
                seen.append(name.split("::")[-1])
        if len(seen) == 2 and ms_step > 0:
            return {"flops_per_step": flops, "frac": flops / (ms_step * 1e-3) / 1e12 / FP64_VALU_PEAK_TFLOPS, "kernels": seen,
                    "source": os.path.basename(f)}
    return None


def fresh_series(fresh, K, W, active, units, ms_step_headline):
    """The second timed series of the headline run: every region = iterations W..W+K of a fresh solve."""
    plain = [r for r, ins, _k in fresh if not ins] or [r for r, _ins, _k in fresh]
    med = statistics.median(plain)
    kk = [k for _r, ins, k in fresh if ins and k is not None]
    out = {"what": "each region = iterations %d..%d of a FRESH solve of the same batch (GPU warm): the regime a caller who solves a "
                   "problem once is in; `value` above is the converged regime of one long solve" % (W, W + K),
           "regions": len(fresh), "ms_per_step": [r / K * 1e3 for r, _i, _k in fresh], "with_kernel_events": [i for _r, i, _k in fresh],
           "median_ms_per_step": med / K * 1e3, "value": units * K / med, "ratio_to_headline_ms_per_step": (med / K * 1e3) / ms_step_headline,
           "active_fraction_at_region_end": active}
    if kk:
        out["kernel_ms_per_step"] = {"backward": statistics.median(k[0] for k in kk), "rollout": statistics.median(k[1] for k in kk),
                                     "linearize": statistics.median(k[2] for k in kk)}
    return out


def measured_traffic(kernel):
    """HBM bytes per launch of `kernel` from the newest committed rocprofv3 PMC passes
    (profiles/*_hbm_traffic_pmc.json: FETCH_SIZE and WRITE_SIZE collected in separate --pmc runs of the default
    command, FETCH_SIZE doubled as MI355X_MICROARCH.md §HBM prescribes).  PMC counters cannot be read from inside
    the timed process, so this is the figure of the profiled run, or None."""
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
    name, v = _pick_kernel(best[1].get("kernels", {}), kernel)
    if name:
        return v.get("hbm_bytes_per_launch_fetch_doubled"), os.path.basename(best[2])
    return None, os.path.basename(best[2])


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


def cpu_baseline(prob, x0_q, x0_xi, us0, seconds, mode="ms", line_search=False, rollout="nonlinear"):
    """The CPU oracle (oracle/tolg_oracle.c: the parity-checked port of the reference algorithm) on this
    host's cores: OpenMP over trajectories, one workspace per thread.  Bounded sample of the same workload:
    all B trajectories, as many iterations as fit in about `seconds`; thread count and iteration count come
    from short calibration probes, so the sample stays bounded whatever the host's real CPU share is."""
    from oracle import bridge as ob
    op = ob.OracleProblem(prob.kind, prob.J, prob.dt, prob.Q, prob.R, prob.P, prob.q_ref, prob.xi_ref,
                          pend_mass=getattr(prob, "pend_mass", 0.0), pend_length=getattr(prob, "pend_length", 0.0))
    kw = dict(mode=mode, line_search=line_search) if line_search else dict(mode=mode)
    if rollout != "nonlinear":
        kw["rollout"] = rollout
    B = x0_q.shape[0]
    # one thread, 4 trajectories x 20 iterations: the single-core rate (what one reference fit corresponds to)
    nb1, it1 = min(4, B), 20
    t0 = time.perf_counter()
    ob.fit_batch(op, x0_q[:nb1], x0_xi[:nb1], us0[:nb1], max_iter=it1, threads=1, **kw)
    r1 = nb1 * it1 / (time.perf_counter() - t0)                      # trajectory-iterations/s on one core
    # thread count: the visible CPUs are not necessarily the CPU share of this job (a GPU box shows 256 hardware
    # threads and grants fewer): calibrate a few counts on a 512-trajectory, 2-iteration probe and keep the fastest
    share = host_cpu_share()
    nbc = min(B, 512)
    best = (0.0, 1)
    for th in sorted({min(share, c) for c in (8, 16, 32, 64, 128)} | {share}):
        t0 = time.perf_counter()
        ob.fit_batch(op, x0_q[:nbc], x0_xi[:nbc], us0[:nbc], max_iter=2, threads=th, **kw)
        best = max(best, (nbc * 2 / (time.perf_counter() - t0), th))
    cores = best[1]
    per_iter = B / best[0]
    iters = int(max(3, min(200, seconds / per_iter)))
    t0 = time.perf_counter()
    r = ob.fit_batch(op, x0_q, x0_xi, us0, max_iter=iters, threads=cores, **kw)
    dt = time.perf_counter() - t0
    used = r["threads"]
    return {"value": iters / dt, "unit": "batch-iterations/s", "cores": used, "kind": "port",
            "sample": "all %d trajectories x N=%d, %d iterations (%.1f s), OpenMP over trajectories on %d threads"
                      % (B, prob.N, iters, dt, used),
            "trajectory_iterations_per_s": B * iters / dt,
            "single_core_trajectory_iterations_per_s": r1,
            "parallel_efficiency": (B * iters / dt) / (r1 * used),
            # the reference itself cannot run here (manifpy / jax absent, and no reference file travels): its own recorded
            # timing, for the record -- BASELINE.md section 2, row 2
            "reference_as_shipped": {
                "ms_per_knot_iteration": 1.9, "trajectory_iterations_per_s": 3.3 * 150 / prob.N,
                "batch_iterations_per_s_one_process": 3.3 * 150 / prob.N / B,
                "hardware": "unknown", "threads": 1,
                "source": "baseline_applications.ipynb:170-183: DroneDynamics N = 150, 28 iterations in 8.44 s = 3.3 iterations/s of ONE "
                          "trajectory, scaled to this horizon by the knot count; recorded by the reference's authors, not measured here"}}


def run_dry(args, rank, world):
    """Plumbing check on CPU (gloo): rendezvous, barrier, MAX-reduced region time, the batch partition of both
    scaling modes, the final gather through sharding.gather_results, and the JSON line -- everything of the
    multi-rank path except the solver."""
    import datetime
    import torch
    import torch.distributed as dist
    from trajectory_optimization_matrix_lie_groups_amd import sharding
    if rank == args.dry_run_fail_rank:
        sys.exit(3)  # a rank that dies before the rendezvous: the launcher must not wait for the others forever
    if world > 1:
        dist.init_process_group("gloo", rank=rank, world_size=world, timeout=datetime.timedelta(seconds=RENDEZVOUS_TIMEOUT_S))
        dist.barrier()
    t = torch.tensor([0.001 * (rank + 1)], dtype=torch.float64)
    if world > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    B_local, B_global = local_and_global_batch(args, rank, world)
    Bg = 4 * world + 1                                    # uneven shards on purpose
    lo, hi = sharding.shard_bounds(Bg, world, rank)
    local = torch.arange(lo, hi, dtype=torch.float64).reshape(-1, 1) * torch.ones(1, 3, dtype=torch.float64)
    full = sharding.gather_results(local, Bg)
    ok = bool(torch.equal(full[:, 0], torch.arange(Bg, dtype=torch.float64)))
    sizes = torch.tensor([float(B_local)], dtype=torch.float64)
    if world > 1:
        dist.all_reduce(sizes, op=dist.ReduceOp.SUM)
    if rank == 0:
        print(json.dumps({"metric": METRIC, "dry_run": True, "n_gpus": world, "region_s_max": float(t.item()),
                          "gather_ok": ok, "gathered_rows": int(full.shape[0]), "scaling": args.scaling,
                          "workload": args.workload, "batch_rank0": B_local, "global_batch": B_global,
                          "sum_of_rank_batches": int(sizes.item())}), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    return 0 if ok else 1


def local_and_global_batch(args, rank, world):
    """Trajectories this rank owns / in the whole job.  weak: --batch per rank; strong: --batch in all, contiguous
    shards (sharding.shard_bounds)."""
    if args.scaling == "weak":
        return args.batch, args.batch * world
    from trajectory_optimization_matrix_lie_groups_amd import sharding
    lo, hi = sharding.shard_bounds(args.batch, world, rank)
    return hi - lo, args.batch


def run_rank(args, rank, world):
    import datetime
    import torch
    import torch.distributed as dist
    from trajectory_optimization_matrix_lie_groups_amd import BatchedTrackingILQR, sharding, workloads

    ident = library_identity(args.allow_lib_override)
    local_rank = int(os.environ.get("LOCAL_RANK", str(rank)))
    multi = world > 1 or args.rccl_selftest   # every `if multi` below is the multi-rank path
    # stdout carries ONE JSON line and nothing else.  RCCL prints a version banner with printf when its first communicator
    # comes up (five lines on this image): in the multi-rank path the process's stdout descriptor is therefore pointed at
    # stderr for the whole run, and rank 0 writes its line to the descriptor that was stdout.
    real_stdout = None
    if multi:
        sys.stdout.flush()
        real_stdout = os.dup(1)
        os.dup2(2, 1)
    if multi:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    if args.rccl_selftest and world == 1:
        os.environ.setdefault("MASTER_PORT", str(free_port()))
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if multi:
        dist.init_process_group("nccl", device_id=dev, timeout=datetime.timedelta(seconds=RENDEZVOUS_TIMEOUT_S))

    N, K, W, R = args.horizon, args.steps, args.warmup, max(1, args.repeats)
    B, B_global = local_and_global_batch(args, rank, world)
    if B < 1:
        raise SystemExit("bench.py: --batch %d leaves rank %d without a trajectory" % (args.batch, rank))
    # drone400: R = 1e-3 I.  With the R = 1e-5 of benchmark_drone_racing_tracking.py:208 (a 150-knot problem) accept-always MS
    # diverges on the 400-knot problem for nearly every member -- in the oracle as on the GPU, tests/test_gpu_configs.py --
    # and a diverged trajectory stops doing work: tools/drone_spread_survival.py, profiles/r03_drone_spread_survival.txt
    drone = lambda B_, N=400, seed=workloads.SEED: workloads.drone_tracking(B_, N=N, seed=seed, R_scale=args.r_scale or 1e-3)
    se3 = (lambda B_, N=200, seed=workloads.SEED: workloads.se3_tracking(B_, N=N, seed=seed, R_scale=args.r_scale)) if args.r_scale \
        else workloads.se3_tracking
    def pend(B_, N=80, seed=workloads.SEED):
        p_, q_, xi_, u_ = workloads.pendulum_swingup(B_, seed=seed)
        if N != p_.N:
            raise SystemExit("bench.py: the stored pendulum swing-up path has %d knots" % p_.N)
        return p_, q_, xi_, u_
    make = {"se3": se3, "drone400": drone, "so3": workloads.so3_tracking, "pendulum": pend,
            "al1024": lambda B_, N=200, seed=workloads.SEED: workloads.al_tracking(B_, N=N, seed=seed)[:4]}[args.workload]
    if args.scaling == "weak":
        # each rank owns an independent shard of the weak-scaled batch: different seeded perturbations
        prob, x0_q, x0_xi, us0 = make(B, N=N, seed=workloads.SEED + rank)
    else:
        # one global batch (the same on every rank count), this rank's contiguous shard of it
        prob, gq, gxi, gus = make(B_global, N=N)
        lo, hi = sharding.shard_bounds(B_global, world, rank)
        x0_q, x0_xi, us0 = gq[lo:hi], gxi[lo:hi], gus[lo:hi]
    m = us0.shape[2]
    if args.inertia == "dense":
        # the same problem with full inertia blocks (a rotated body frame's inertia): J = blkdiag(Ib + A A^T, Jv) keeps the
        # structure the reference's G assumes; the backward sweep then reads I + H dt from the records (k_backward)
        import numpy as np
        from trajectory_optimization_matrix_lie_groups_amd import TrackingProblem
        Jd = np.array(prob.J, dtype=float).copy()
        A_ = np.array([[0.10, -0.05, 0.02], [0.03, 0.12, -0.04], [-0.02, 0.06, 0.09]])
        Jd[:3, :3] += A_ @ A_.T
        if prob.kind == "se3":
            Jd[3:, 3:] += 0.5 * (A_ @ A_.T)
        prob = TrackingProblem(prob.kind, Jd, prob.dt, prob.Q, prob.R, prob.P, prob.q_ref, prob.xi_ref)
    solver = BatchedTrackingILQR(prob, B, device=dev)
    x0_q_d = torch.as_tensor(x0_q, device=dev); x0_xi_d = torch.as_tensor(x0_xi, device=dev)
    us0_d = torch.as_tensor(us0, device=dev)
    if args.workload == "al1024":
        # ALConstrainedCost + InputConstraint(-10, 10) with the multipliers / penalties of the first outer iteration
        # (traopt_controller.py:3218-3240: lmbd = 0, Imu = mu0 I): the timed steps are inner MS iterations of that solve
        lam_d = torch.zeros(B, N, 2 * m, dtype=torch.float64, device=dev)
        imu_d = torch.full((B, N, 2 * m), 1e-2, dtype=torch.float64, device=dev)
        solver.set_al([-10.0] * m, [10.0] * m, lam_d, imu_d)

    def barrier():
        torch.cuda.synchronize(dev)
        if multi:
            dist.barrier()
        torch.cuda.synchronize(dev)

    headline = args.mode == "ms" and not args.line_search
    regions, kern = [], []
    fresh, fresh_active = [], None
    begin_kw = dict(mode=args.mode, tol_grad_norm=0.0, tol_d_norm=0.0, schedule=args.schedule, line_search=args.line_search,
                    rollout=args.rollout)

    # Per-kernel durations need an event pair per launch, and timing costs the step 11 us of its 600 (tools/event_cost.py:
    # 607 against 596 us at 4096 x 200 with the pair attached to the dispatch; 614 with marker packets around it).  So the regions alternate: even ones carry the
    # events (-> kernel_ms_per_step, the roofline's kernel figures), odd ones run the library as a caller runs it (-> value,
    # ms_per_step).  Both series are in the line.  With one region only (--repeats 1: the profiling runs) it carries the events.
    instrumented = []

    def timed_region():
        instr = len(regions) % 2 == 0
        solver.enable_timing(instr)
        barrier()
        t0 = time.perf_counter()
        solver.solve_iterate(K)
        barrier()
        t1 = time.perf_counter()
        if instr:
            ms_b, ms_r, ms_l, n_b = solver.kernel_time(reset=True)
            kern.append((ms_b / max(n_b, 1), ms_r / max(n_b, 1), ms_l / max(n_b, 1)))
        else:
            kern.append(None)
        solver.enable_timing(False)
        el = torch.tensor([t1 - t0], dtype=torch.float64, device=dev)
        if multi:
            dist.all_reduce(el, op=dist.ReduceOp.MAX)
        regions.append(float(el.item()))
        instrumented.append(instr)

    if headline:
        # accept-always MS with tolerances 0: every trajectory does full work in every iteration of one long solve
        total = W + R * K
        solver.solve_begin(x0_q_d, x0_xi_d, us0_d, n_iterations=total, **begin_kw)
        solver.solve_iterate(W)
        for _ in range(R):
            timed_region()
        res = solver.solve_end()
        # The series above is ONE long solve: all but its first region time iterations of a converged solve, whose knots sit in
        # the short tier of every series (tolg_lie.h) -- the cheapest regime.  A caller who solves a problem once spends its
        # time in the FIRST iterations: each region of this second series is iterations W..W+K of a fresh solve (the protocol
        # of the line-search lines), on the GPU the first series has warmed up.  Reported beside `value`, never as it.
        n_head = len(regions)
        for _ in range(max(0, args.fresh_regions)):
            solver.solve_begin(x0_q_d, x0_xi_d, us0_d, n_iterations=W + K, **begin_kw)
            solver.solve_iterate(W)
            timed_region()
            fres = solver.solve_end()
            fresh_active = float((fres.iters == W + K).double().mean().item())
        fresh = [(regions[i], instrumented[i], kern[i]) for i in range(n_head, len(regions))]
        del regions[n_head:], instrumented[n_head:], kern[n_head:]
    else:
        # line-search modes stop trajectories that find no descent: every region is iterations W .. W+K of a fresh
        # solve, and the line reports how many trajectories were still being solved at the end of it
        total = W + K
        for _ in range(R):
            solver.solve_begin(x0_q_d, x0_xi_d, us0_d, n_iterations=total, **begin_kw)
            solver.solve_iterate(W)
            timed_region()
            res = solver.solve_end()
    torch.cuda.synchronize(dev)
    active_end = float((res.iters == total).double().mean().item())
    # iterations the trajectories of this rank actually ran inside the last timed region (a stopped trajectory runs none): the
    # line-search lines report the rate over THAT work beside `value`, which counts every trajectory of the batch as iterated
    if headline:
        worked = None
    else:
        worked = float((res.iters.clamp(min=W, max=total) - W).double().sum().item())
    # final gather of costs / controls over RCCL (outside the timed solve, reported separately)
    gather_ms, gather_err = None, None
    if multi:
        try:
            last = (res.iters.clamp(min=1) - 1).long().reshape(-1, 1)
            Jf = torch.gather(res.J_hist, 1, last).contiguous()
            torch.cuda.synchronize(dev)
            g0 = time.perf_counter()
            Jall = sharding.gather_results(Jf, B_global) if args.scaling == "strong" else gather_even(Jf, world)
            Uall = sharding.gather_results(res.us, B_global) if args.scaling == "strong" else gather_even(res.us, world)
            torch.cuda.synchronize(dev)
            gather_ms = (time.perf_counter() - g0) * 1e3
            assert Jall.shape[0] == B_global and Uall.shape[0] == B_global
        except Exception as e:  # the timed figure above stands on its own; say what happened to the gather
            gather_err = "%s: %s" % (type(e).__name__, e)
    finite = bool(torch.isfinite(res.J_hist[:, :total]).all().item()) if headline else None
    clean = bool((res.status == 0).all().item())
    # accept-always solves with tolerances 0 must keep every trajectory working to the end: a stopped trajectory (diverged,
    # status != 0) does no work, and a rate measured over it is not the workload's rate
    invalid = None
    if headline and (active_end < 1.0 or not clean or not finite):
        invalid = ("only %.1f %% of the trajectories were still being solved at the end of the timed regions (status ok: %s, "
                   "finite: %s): the figure is NOT a valid rate for this workload" % (100 * active_end, clean, finite))
        if rank == 0:
            print("bench.py: WARNING: " + invalid, file=sys.stderr)

    if rank == 0:
        plain = [r for r, ins in zip(regions, instrumented) if not ins]
        with_ev = [r for r, ins in zip(regions, instrumented) if ins]
        med = statistics.median(plain if plain else with_ev)
        med_ev = statistics.median(with_ev)
        imed = min((i for i in range(R) if instrumented[i]), key=lambda i: abs(regions[i] - med_ev))
        kb, kr, kl = kern[imed]
        # weak: every rank advanced its own --batch trajectories K times; strong: the one global batch K times
        value = (world if args.scaling == "weak" else 1) * K / med
        ms_step = med / K * 1e3
        alg_bytes = alg_bytes_per_knot_iter(m) * B * N          # per launch on this rank (what its kernels process)
        dominant, t_dom = max((("k_backward3", kb), ("k_rollout_lin" if kl == 0.0 else "k_rollout", kr),
                               ("k_linearize", kl)), key=lambda kv: kv[1])
        dom_gbs = alg_bytes / (t_dom * 1e-3) / 1e9 if t_dom > 0 else None
        step_gbs = alg_bytes / (ms_step * 1e-3) / 1e9
        std_cfg = (args.workload == "se3" and B == 4096 and N == 200 and headline and args.schedule == "auto" and not args.r_scale
                   and args.rollout == "nonlinear" and args.inertia == "diag")
        traffic, traffic_src = measured_traffic(dominant) if std_cfg else (None, None)
        measured_gbs = traffic / (t_dom * 1e-3) / 1e9 if (traffic and t_dom > 0) else None
        what = {"se3": "SE3 exact tracking", "drone400": "drone racing tracking (BASELINE config 5; R = 1e-3 I)",
                "so3": "SO3 exact tracking (BASELINE config 2)",
                "pendulum": "Pendulum3dDyanmics swing-up tracking (state-dependent input matrix)",
                "al1024": "SE3 AL-DDP MS with input box constraints, inner iterations of the first outer iteration "
                          "(BASELINE config 4)"}[args.workload]
        algo = ("MS-iLQR" if args.mode == "ms" else "SS-iLQR") + (
            " (line_search=%s, rollout=%s)" % ("True" if args.line_search else "False", args.rollout) if args.mode == "ms"
            else " (13-alpha backtracking%s)" % (", rollout=linear" if args.rollout == "linear" else ""))
        if args.inertia == "dense":
            what += " [dense inertia blocks]"
        if args.r_scale:
            what += " [R = %g I]" % args.r_scale
        metric = METRIC if std_cfg else (
            "DDP iterations/sec at batch x horizon = %d x %d (%s, %s)" % (B_global if args.scaling == "strong" else B, N, what, algo))
        line = {
            "metric": metric,
            "value": value, "unit": "batch-iterations/s", "n_gpus": world, "steps": K, "warmup": W,
            "ms_per_step": ms_step, "higher_is_better": True, "scaling": args.scaling,
            "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": "%s, %s, %s, seeded perturbed initial states" % (
                           what, algo, ("B=%d trajectories per GPU" % B) if args.scaling == "weak" else
                           ("one global batch of %d split over %d GPU(s): %d on rank 0" % (B_global, world, B))) +
                           " x N=%d knots" % N,
                       "batch_per_gpu": B, "horizon": N, "global_batch": B_global, "schedule": args.schedule,
                       "mode": args.mode, "line_search": bool(args.line_search), "rollout": args.rollout, "inertia": args.inertia,
                       "value_counts": "one unit = every trajectory of a %d-trajectory batch advanced by one iteration"
                                       % (B if args.scaling == "weak" else B_global),
                       "trajectory_iterations_per_s": value * (B if args.scaling == "weak" else B_global),
                       "active_fraction_at_region_end": active_end, "all_finite": finite, "all_status_ok": clean,
                       **({"active_normalised": {
                           "trajectory_iterations_run_in_last_region_rank0": worked,
                           "share_of_batch_x_steps": worked / (B * K),
                           "trajectory_iterations_per_s_rank0": worked / (regions[-1] if regions else float("nan")),
                           "note": "value x batch counts a stopped trajectory as iterated; this is the last region's rate over the "
                                   "iterations its trajectories really ran (rank 0's shard)"}} if worked is not None else {}),
                       **({"invalid": invalid} if invalid else {}),
                       "timed_regions": {"repeats": R, "steps_each": K,
                                         "reported": ("median of the regions WITHOUT per-kernel events (odd ones); kernel_ms_per_step "
                                                      "from the median region WITH them (even ones)") if plain else
                                                     "the only region(s) carry the per-kernel events",
                                         "ms_per_step": [r / K * 1e3 for r in regions],
                                         "with_kernel_events": instrumented,
                                         "median_ms_per_step_with_kernel_events": med_ev / K * 1e3,
                                         "min_ms_per_step": min(regions) / K * 1e3,
                                         "max_ms_per_step": max(regions) / K * 1e3},
                       **({"fresh_solve": fresh_series(fresh, K, W, fresh_active, world if args.scaling == "weak" else 1, ms_step)} if fresh else {}),
                       "kernel_ms_per_step": {"backward": kb, "rollout": kr, "linearize": kl},
                       "kernel_ms_per_step_note": "a HIP event pair per launch (attached to the dispatch of the two hot kernels), measured in the "
                                                  "regions that carry them, whose median step is "
                                                  "timed_regions.median_ms_per_step_with_kernel_events, not ms_per_step: timing "
                                                  "lengthens the step by ~11 us",
                       **({"rccl_selftest": "process group of one rank: barrier, all_reduce(MAX), all_gather ran on the device"}
                          if (args.rccl_selftest and world == 1) else {}),
                       "final_gather_ms": gather_ms, **({"final_gather_error": gather_err} if gather_err else {}),
                       **ident},
            # `frac` is SURVEY §8d's own formula: algorithmic bytes of one batch-iteration over the WHOLE step.
            # Per-kernel figures carry their own keys; `traffic` is what the PMC counters of the profiled run saw for
            # the dominant kernel, `measured_*` what that traffic means against the achievable 6.29 TB/s.
            "roofline": {"bound": "hbm", "achieved": step_gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": step_gbs / HBM_PEAK_GBS, "traffic": traffic, "traffic_source": traffic_src,
                         "algorithmic_bytes_per_step": alg_bytes,
                         "frac_achievable": step_gbs / HBM_ACHIEVABLE_GBS, "peak_achievable": HBM_ACHIEVABLE_GBS,
                         "dominant_kernel": dominant, "dominant_kernel_avg_ms": t_dom,
                         "achieved_dominant_kernel": dom_gbs,
                         "frac_dominant_kernel": (dom_gbs / HBM_PEAK_GBS) if dom_gbs else None,
                         "measured_achieved_dominant_kernel": measured_gbs,
                         "measured_frac_of_achievable_dominant_kernel": (measured_gbs / HBM_ACHIEVABLE_GBS) if measured_gbs else None,
                         "fp64_issue_dominant_kernel": fp64_issue_fraction(dominant, t_dom) if std_cfg else None,
                         "fp64_frac_step": ALG_FLOPS_PER_KNOT_ITER * B * N / (ms_step * 1e-3) / 1e12 / FP64_VALU_PEAK_TFLOPS,
                         "fp64_frac_step_executed": executed_fp64_fraction(ms_step) if std_cfg else None,
                         "note": "bound named as BASELINE.json stipulates (HBM); by the numbers the step moves its "
                                 "algorithmic bytes at frac of the HBM peak while the dominant kernel's MEASURED traffic "
                                 "runs at measured_frac_of_achievable of what the memory system delivers and the dense "
                                 "25 kflop per knot-iteration at fp64_frac_step of the fp64 vector peak (nominal count; "
                                 "fp64_frac_step_executed prices the fp64 instructions the two launches actually executed, idle "
                                 "lanes included): an fp64 issue / latency-bound sweep that also moves ~6x its algorithmic bytes "
                                 "(DESIGN.md §5)"},
        }
        if not args.no_cpu_baseline and world == 1 and args.workload != "al1024":  # (the oracle's batch driver has no AL terms)
            line["cpu_baseline"] = cpu_baseline(prob, x0_q, x0_xi, us0, args.cpu_seconds, args.mode, args.line_search, args.rollout)
        if real_stdout is not None:
            sys.stdout.flush()
            os.write(real_stdout, (json.dumps(line) + "\n").encode())
        else:
            print(json.dumps(line), flush=True)
    if multi:
        dist.barrier()
        dist.destroy_process_group()
    return 0


def gather_even(local, world):
    """all_gather of equal shards (weak scaling: every rank holds --batch rows)."""
    import torch
    import torch.distributed as dist
    outs = [torch.empty_like(local) for _ in range(world)]
    dist.all_gather(outs, local.contiguous())
    return torch.cat(outs, dim=0)


def main(argv=None):
    argv = sys.argv[1:] if argv is None else argv
    args = parse_args(argv)
    kind, rank, world = launch_plan(args.gpus, os.environ)
    if kind == "error":
        print("bench.py: " + rank, file=sys.stderr)
        return 2
    if kind == "spawn":
        return spawn_ranks(world, argv, args.deadline)
    if args.dry_run:
        return run_dry(args, rank, world)
    return run_rank(args, rank, world)


if __name__ == "__main__":
    sys.exit(main())
