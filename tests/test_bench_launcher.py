"""bench.py's rank launcher and multi-rank plumbing, on CPU (gloo, world size 2): rank -> device map and
environment, the refusal of a --gpus / WORLD_SIZE mismatch, and -- through `--dry-run`, which runs everything
of the multi-rank path except the solver -- rendezvous, MAX-reduced region time, the final gather through
sharding.gather_results and `n_gpus` in the JSON line."""
import json
import os
import subprocess
import sys

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
