#!/usr/bin/env python3
"""bench.py -- BASELINE.json metric: DDP (MS-iLQR) iterations/s at batch x horizon = 4096 x 200,
SE3 exact tracking, on N GPUs of one node (one process per GPU, weak scaling: 4096 trajectories
per GPU, no collective inside the solve; one RCCL all_gather of costs/controls afterwards).

A "step" is one batch-iteration: every one of the B*N knot-iterations of the batch advanced once
(backward Riccati sweep + closed-loop rollout + re-linearisation).  Inputs are
resident in HBM before the timed region; W warm-up steps, then exactly K steps timed between
barrier + synchronize pairs, max over ranks.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8 TB/s spec
FP64_VALU_PEAK_TFLOPS = 78.6   # vendor fp64 vector peak
ALG_BYTES_PER_KNOT_ITER = 448  # read+write of (q 4x4, xi 6, u 6) in fp64
ALG_FLOPS_PER_KNOT_ITER = 25e3  # dense count, SURVEY.md §8d (secondary figure)


def measured_traffic(kernel="k_backward"):
    """HBM bytes per launch of the dominant kernel from the committed rocprofv3 PMC passes
    (profiles/*_hbm_traffic_pmc.json: FETCH_SIZE and WRITE_SIZE collected in separate --pmc runs of this
    same command, FETCH_SIZE doubled as MI355X_MICROARCH.md §HBM prescribes).  PMC counters cannot be read
    from inside the timed process, so this is the figure of the profiled run, or None."""
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "*_hbm_traffic_pmc.json")))
    if not files:
        return None, None
    try:
        d = json.load(open(files[-1]))
        for name, v in d["kernels"].items():
            if kernel in name:
                return v["hbm_bytes_per_launch_fetch_doubled"], os.path.basename(files[-1])
    except Exception:
        pass
    return None, None


def cpu_baseline(prob, x0_q, x0_xi, us0, iters):
    """The CPU oracle (oracle/tolg_oracle.c: the parity-checked port of the reference algorithm),
    OpenMP over trajectories on this host's cores, same workload, `iters` iterations."""
    from oracle import bridge as ob
    op = ob.OracleProblem(prob.kind, prob.J, prob.dt, prob.Q, prob.R, prob.P, prob.q_ref, prob.xi_ref)
    cores = os.cpu_count() or 1
    try:
        cores = len(os.sched_getaffinity(0))
    except Exception:
        pass
    os.environ["OMP_NUM_THREADS"] = str(cores)
    B = x0_q.shape[0]
    t0 = time.perf_counter()
    ob.fit_batch(op, x0_q, x0_xi, us0, mode="ms", max_iter=iters, tol_grad=0.0, tol_defect=0.0)
    dt = time.perf_counter() - t0
    # the same solver on ONE core, one trajectory (what the reference's per-trajectory fit corresponds to)
    t1 = time.perf_counter()
    ob.fit(op, x0_q[0], x0_xi[0], us0[0], mode="ms", max_iter=iters, tol_grad=0.0, tol_defect=0.0)
    dt1 = time.perf_counter() - t1
    # one oracle "iteration" includes the same phases; the first linearisation is amortised like the GPU side's
    return {"value": iters / dt, "unit": "batch-iterations/s", "cores": cores, "kind": "port",
            "sample": "full workload: %d trajectories x N=%d, %d iterations, OpenMP over trajectories (%.1f s)"
                      % (B, prob.N, iters, dt),
            "trajectory_iterations_per_s": B * iters / dt,
            "single_core_trajectory_iterations_per_s": iters / dt1}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=4096)
    ap.add_argument("--horizon", type=int, default=200)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-iters", type=int, default=40)
    args = ap.parse_args()

    import torch
    import torch.distributed as dist
    from trajectory_optimization_matrix_lie_groups_amd import BatchedTrackingILQR, workloads

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        dist.init_process_group("nccl", device_id=dev)

    B, N, K, W = args.batch, args.horizon, args.steps, args.warmup
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

    res = solver.solve_begin(x0_q_d, x0_xi_d, us0_d, mode="ms", n_iterations=W + K, tol_grad_norm=0.0, tol_d_norm=0.0)
    solver.solve_iterate(W)
    solver.enable_timing(True)
    barrier()
    t0 = time.perf_counter()
    solver.solve_iterate(K)
    barrier()
    t1 = time.perf_counter()
    ms_b, ms_r, ms_l, n_b = solver.kernel_time(reset=True)
    solver.enable_timing(False)
    res = solver.solve_end()
    torch.cuda.synchronize(dev)
    elapsed = t1 - t0
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    # final gather of costs / controls over RCCL (outside the timed solve, reported separately)
    gather_ms, gather_err = None, None
    if world > 1:
        try:
            Jf = res.J_hist[:, W + K - 1].contiguous()
            outJ = [torch.empty_like(Jf) for _ in range(world)]
            outU = [torch.empty_like(res.us) for _ in range(world)]
            torch.cuda.synchronize(dev)
            g0 = time.perf_counter()
            dist.all_gather(outJ, Jf)
            dist.all_gather(outU, res.us)
            torch.cuda.synchronize(dev)
            gather_ms = (time.perf_counter() - g0) * 1e3
        except Exception as e:  # the timed figure above stands on its own; say what happened to the gather
            gather_err = "%s: %s" % (type(e).__name__, e)
    finite = bool(torch.isfinite(res.J_hist[:, : W + K]).all().item())

    if rank == 0:
        value = world * K / elapsed
        t_bwd = ms_b / max(n_b, 1) * 1e-3
        alg_bytes = ALG_BYTES_PER_KNOT_ITER * B * N
        achieved = alg_bytes / t_bwd / 1e9 if t_bwd > 0 else None
        traffic, traffic_src = measured_traffic() if (B == 4096 and N == 200) else (None, None)
        line = {
            "metric": "DDP iterations/sec at batch x horizon = 4096 x 200 (SE3 tracking)",
            "value": value, "unit": "batch-iterations/s", "n_gpus": world, "steps": K, "warmup": W,
            "ms_per_step": elapsed / K * 1e3, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": "SE3 exact tracking, MS-iLQR (line_search=False, rollout=nonlinear), "
                                   "B=%d trajectories per GPU x N=%d knots, path_se3_generate_sine_2, "
                                   "seeded perturbed initial states" % (B, N),
                       "batch_per_gpu": B, "horizon": N, "global_batch": B * world,
                       "trajectory_iterations_per_s": value * B, "all_finite": finite,
                       "kernel_ms_per_step": {"backward": ms_b / max(n_b, 1), "rollout": ms_r / max(n_b, 1),
                                              "linearize": ms_l / max(n_b, 1)},
                       "final_gather_ms": gather_ms, **({"final_gather_error": gather_err} if gather_err else {})},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": (achieved / HBM_PEAK_GBS) if achieved else None, "traffic": traffic,
                         "traffic_source": traffic_src,
                         "kernel": "k_backward", "algorithmic_bytes_per_launch": alg_bytes,
                         "kernel_avg_ms": t_bwd * 1e3,
                         "note": "fp64 VALU-bound by construction (SURVEY §8d): whole-step fp64 fraction = "
                                 "%.3f of %.1f TFLOP/s at 25 kflop per knot-iteration"
                                 % (ALG_FLOPS_PER_KNOT_ITER * B * N * (K / elapsed) / 1e12 / FP64_VALU_PEAK_TFLOPS,
                                    FP64_VALU_PEAK_TFLOPS)},
        }
        if not args.no_cpu_baseline and world == 1:
            line["cpu_baseline"] = cpu_baseline(prob, x0_q, x0_xi, us0, args.cpu_iters)
        print(json.dumps(line))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
