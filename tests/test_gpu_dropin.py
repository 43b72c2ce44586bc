"""Drop-in, caller-shaped: examples/benchmark_tracking_dropin.py has the shape of the reference's benchmark
scripts (benchmark_drone_racing_tracking.py:48-216 / benchmark_SE3_tracking.py:168-217: imports under the
reference's names, SE3(position, quaternion).transform(), the 15- and 12-argument callbacks, MS then SS fit,
result dict).  Run end to end on the GPU on the Drone problem of the reference's recorded notebook run and
compared with that record (tests/golden/drone_n150_log.json); plus the C-ABI entry points that only a caller
loop exercises (solve_peek, active count / early exit, the in-flight guards)."""
import importlib.util
import json
import os
import warnings

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from trajectory_optimization_matrix_lie_groups_amd import BatchedTrackingILQR, TrackingProblem, results_io, workloads  # noqa: E402

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _script():
    spec = importlib.util.spec_from_file_location("benchmark_tracking_dropin",
                                                  os.path.join(ROOT, "examples", "benchmark_tracking_dropin.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def test_caller_shaped_script_reproduces_the_recorded_drone_run(golden_dir, tmp_path, capsys):
    log = json.load(open(os.path.join(golden_dir, "drone_n150_log.json")))
    out = str(tmp_path / "results_drone.npz")
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")  # the SS run ends with the reference's "Couldn't find descent direction"
        data = _script().main(os.path.join(golden_dir, "drone_n150_problem.npz"), "drone", save_to=out)
    printed = capsys.readouterr().out
    its_ms = [it for it in log["ms"]["iterations"] if "J_new" in it]
    ms, ss = data["ms_se3"], data["ss_se3"]
    # multiple shooting: the 28 recorded iterations, through the callback-filled lists
    assert len(ms["J_hist"]) == 28 == len(ms["grad_hist"]) and len(ms["defect_hist"]) == 29
    for k, it in enumerate(its_ms):
        assert ms["J_hist"][k] == pytest.approx(it["J_new"], rel=1e-11)
        assert float(ms["grad_hist"][k]) == pytest.approx(it["grad"], rel=1e-7, abs=2e-14)
    assert ms["defect_hist"][0] == pytest.approx(its_ms[0]["defect_lin"], rel=1e-12)
    assert ms["max_dyn_err"] < 1e-9
    # single shooting: 9 iterations, the last one a failed line search
    its_ss = log["ss"]["iterations"]
    assert len(ss["J_hist"]) == 9
    for k, it in enumerate(its_ss):
        assert ss["J_hist"][k] == pytest.approx(it["cb_J"], rel=1e-11)
    assert printed.count("Iteration") == 28 + 9 and "failed" in printed
    # the result file round-trips in the reference's dict layout
    back = results_io.load_results(out)
    assert set(back) == {"prob", "ms_se3", "ss_se3"}
    assert back["ms_se3"]["J_hist"] == [float(v) for v in ms["J_hist"]]
    assert len(back["ms_se3"]["xs"]) == 151 and back["ms_se3"]["xs"][3][0].shape == (4, 4)
    np.testing.assert_array_equal(back["ss_se3"]["us"], ss["us"])
    np.testing.assert_array_equal(back["prob"]["x0"][0], data["prob"]["x0"][0])
    assert back["prob"]["dt"] == data["prob"]["dt"]


def test_solve_peek_and_active_count_between_slices():
    B = 24
    prob, x0_q, x0_xi, us0 = workloads.se3_tracking(B, N=60, R_scale=1e-3)
    s = BatchedTrackingILQR(prob, B)
    ref = s.fit_batch(x0_q, x0_xi, us0, mode="ms", n_iterations=60, tol_grad_norm=1e-7, check_every=0)
    ref_us, ref_iters = ref.us.clone(), ref.iters.clone()
    res = s.solve_begin(x0_q, x0_xi, us0, mode="ms", n_iterations=60, tol_grad_norm=1e-7)
    assert s.active_count() == B
    counts = []
    for _ in range(12):
        s.solve_iterate(5)
        peek = s.solve_peek()  # does not end the solve
        counts.append(s.active_count())
        assert int(peek.iters.max()) <= 5 * len(counts)
    assert counts[-1] == 0 and sorted(counts, reverse=True) == counts and counts[0] > 0
    # in flight: the unit-parity entry points that would overwrite the workspace are refused
    with pytest.raises(RuntimeError, match="bad argument"):
        s.linearize_backward(np.repeat(prob.q_ref[None], B, 0), np.repeat(prob.xi_ref[None], B, 0), us0)
    with pytest.raises(RuntimeError, match="bad argument"):
        s.rollout(B)
    end = s.solve_end()
    assert torch.equal(end.us, ref_us) and torch.equal(end.iters, ref_iters)
    # the sliced fit with the early exit stops issuing launches but returns the same result
    early = s.fit_batch(x0_q, x0_xi, us0, mode="ms", n_iterations=60, tol_grad_norm=1e-7, check_every=4)
    assert torch.equal(early.us, ref_us) and torch.equal(early.iters, ref_iters)
    # ... one slice past the first count of zero at most (tolg_solve_iterate_until looks one slice behind)
    done_at = int(ref_iters.max())
    assert done_at < 52 and s.iterations_issued <= (done_at + 3) // 4 * 4 + 8 and s.iterations_issued % 4 == 0
    # a solve that nothing can end is one slice, and the one-call entry point takes the same option
    free = s.fit_batch(x0_q, x0_xi, us0, mode="ms", n_iterations=12, tol_grad_norm=0.0, tol_d_norm=0.0, check_every=4)
    assert s.iterations_issued == 12 and int(free.iters.min()) == 12
    one = s.solve_batch_one_call(x0_q, x0_xi, us0, mode="ms", n_iterations=60, tol_grad_norm=1e-7, check_every=4)
    assert torch.equal(one.us, ref_us) and torch.equal(one.iters, ref_iters)
