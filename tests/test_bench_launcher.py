"""bench.py's rank launcher and multi-rank plumbing, on CPU (gloo, world size 2): rank -> device map and
environment, the refusal of a --gpus / WORLD_SIZE mismatch, and -- through `--dry-run`, which runs everything
of the multi-rank path except the solver -- rendezvous, MAX-reduced region time, the final gather through
sharding.gather_results and `n_gpus` in the JSON line; both scaling modes and both workloads through the same
dry run; and the supervisor: a rank that dies before the rendezvous ends the job promptly instead of hanging it."""
import json
import os
import subprocess
import sys
import time

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def test_launch_plan():
    assert bench.launch_plan(1, {}) == ("single", 0, 1)
    assert bench.launch_plan(4, {}) == ("spawn", 0, 4)
    assert bench.launch_plan(4, {"WORLD_SIZE": "4", "RANK": "3"}) == ("worker", 3, 4)
    assert bench.launch_plan(1, {"WORLD_SIZE": "1", "RANK": "0"}) == ("single", 0, 1)
    assert bench.launch_plan(8, {"WORLD_SIZE": "2", "RANK": "0"})[0] == "error"
    assert bench.launch_plan(0, {})[0] == "error"


def test_rank_env_maps_one_device_per_rank():
    for r in range(4):
        e = bench.rank_env(r, 4, 29511, base={"PATH": "/usr/bin"})
        assert e["RANK"] == str(r) and e["LOCAL_RANK"] == str(r) and e["WORLD_SIZE"] == "4"
        assert e["MASTER_ADDR"] == "127.0.0.1" and e["MASTER_PORT"] == "29511"
        assert e["HSA_ENABLE_IPC_MODE_LEGACY"] == "0" and e["PATH"] == "/usr/bin"


def _clean_env():
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    return env


def test_spawned_world2_dry_run_reports_two_ranks():
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--dry-run"], env=_clean_env(),
                         capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1  # rank 0 alone prints
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["gather_ok"] is True and d["gathered_rows"] == 9
    assert abs(d["region_s_max"] - 0.002) < 1e-12  # MAX over ranks of (rank + 1) ms


def test_world_size_mismatch_is_refused():
    env = _clean_env()
    env.update(WORLD_SIZE="4", RANK="0")
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--dry-run"], env=env,
                         capture_output=True, text=True, timeout=120)
    assert out.returncode == 2 and "WORLD_SIZE" in out.stderr


def _dry(*flags, timeout=300):
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--dry-run", *flags], env=_clean_env(),
                         capture_output=True, text=True, timeout=timeout)
    return out


def test_dry_run_strong_scaling_splits_one_global_batch():
    out = _dry("--gpus", "2", "--scaling", "strong", "--batch", "4097")
    assert out.returncode == 0, out.stderr[-2000:]
    d = json.loads([ln for ln in out.stdout.splitlines() if ln.startswith("{")][0])
    assert d["scaling"] == "strong" and d["global_batch"] == 4097 and d["batch_rank0"] == 2049
    assert d["sum_of_rank_batches"] == 4097  # contiguous shards cover the global batch exactly once


def test_dry_run_weak_scaling_and_drone_defaults():
    out = _dry("--gpus", "2")
    d = json.loads([ln for ln in out.stdout.splitlines() if ln.startswith("{")][0])
    assert d["scaling"] == "weak" and d["batch_rank0"] == 4096 and d["global_batch"] == 8192 and d["sum_of_rank_batches"] == 8192
    out = _dry("--gpus", "2", "--workload", "drone400")
    d = json.loads([ln for ln in out.stdout.splitlines() if ln.startswith("{")][0])
    # BASELINE config 5: 8192 trajectories in all, strong scaling by default (1024 per GPU on 8)
    assert d["workload"] == "drone400" and d["scaling"] == "strong" and d["global_batch"] == 8192 and d["batch_rank0"] == 4096
    a = bench.parse_args(["--workload", "drone400"])
    assert (a.batch, a.horizon, a.scaling) == (8192, 400, "strong")
    a = bench.parse_args([])
    assert (a.batch, a.horizon, a.scaling, a.repeats, a.mode) == (4096, 200, "weak", 200, "ms")
    # BASELINE configs 2 and 4 at their stated sizes (secondary lines); the SO3 script runs the SS solver
    a = bench.parse_args(["--workload", "so3"])
    assert (a.batch, a.horizon, a.scaling, a.mode) == (1, 100, "weak", "ss")
    a = bench.parse_args(["--workload", "so3", "--mode", "ms", "--batch", "256"])
    assert (a.batch, a.mode) == (256, "ms")
    a = bench.parse_args(["--workload", "al1024"])
    assert (a.batch, a.horizon, a.scaling, a.mode) == (1024, 200, "weak", "ms")


def test_rank_dying_before_rendezvous_ends_the_job_promptly():
    """Rank 1 exits 3 before init_process_group: rank 0 would wait in the rendezvous; the supervisor must take it
    down and return the failure, well inside the rendezvous timeout."""
    t0 = time.monotonic()
    out = _dry("--gpus", "2", "--dry-run-fail-rank", "1", timeout=120)
    dt = time.monotonic() - t0
    assert out.returncode == 3, (out.returncode, out.stderr[-1000:])
    assert dt < 60, dt
    assert not [ln for ln in out.stdout.splitlines() if ln.startswith("{")]  # no result line from a failed job


def test_supervisor_deadline_kills_stuck_ranks():
    procs = [subprocess.Popen([sys.executable, "-c", "import time; time.sleep(300)"]) for _ in range(2)]
    t0 = time.monotonic()
    rc = bench.supervise(procs, deadline_s=1.0)
    assert rc == 124 and time.monotonic() - t0 < 30
    assert all(p.poll() is not None for p in procs)


def test_lib_override_is_refused_without_the_flag(tmp_path):
    env = _clean_env()
    env["TOLG_HIP_LIB"] = str(tmp_path / "other.so")
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "1"], env=env, capture_output=True,
                         text=True, timeout=300)
    assert out.returncode != 0 and "--allow-lib-override" in (out.stderr + out.stdout)


def test_profile_lookups_pick_the_fast_sweep_and_fresh_series_shape():
    """bench.py's helpers that read the committed profiles: since round 4 the backward sweep is two kernels and the headline's
    dominant one is the fast sweep (`k_backward3<.., true>`); the executed-fp64 figure sums it and the fused launch; the fresh-solve
    series reports the median of the regions without per-kernel events."""
    ks = {"void tolg::k_backward3<6, false, false, false>": {"SQ_WAVES": 1024.0, "SQ_INSTS_VALU_FMA_F64": 1.0},
          "void tolg::k_backward3<6, false, false, true>": {"SQ_WAVES": 1024.0, "SQ_INSTS_VALU_FMA_F64": 2.0},
          "void tolg::k_rollout_lin<6>": {"SQ_WAVES": 1024.0, "SQ_INSTS_VALU_FMA_F64": 3.0}}
    name, v = bench._pick_kernel(ks, "k_backward3")
    assert name.endswith("true>") and v["SQ_INSTS_VALU_FMA_F64"] == 2.0
    assert bench._pick_kernel(ks, "k_rollout_lin")[0].endswith("k_rollout_lin<6>")
    assert bench._pick_kernel({"void tolg::k_backward3<6, false, false>": {}}, "k_backward3")[0].endswith("false>")   # round-3 profiles
    assert bench._pick_kernel(ks, "k_nothing") == (None, None)
    fr = bench.fresh_series([(0.012, True, (0.30, 0.25, 0.0)), (0.011, False, None), (0.013, True, (0.31, 0.24, 0.0)), (0.0115, False, None)],
                            20, 5, 1.0, 1, 0.54)
    assert fr["regions"] == 4 and fr["median_ms_per_step"] == pytest.approx(0.5625) and fr["value"] == pytest.approx(20 / 0.01125)
    assert fr["kernel_ms_per_step"]["backward"] == pytest.approx(0.305) and fr["ratio_to_headline_ms_per_step"] == pytest.approx(0.5625 / 0.54)
    ex = bench.executed_fp64_fraction(0.54)   # from the committed profiles/*_sq_mix.json
    assert ex is None or (0.1 < ex["frac"] < 1.0 and len(ex["kernels"]) == 2)
