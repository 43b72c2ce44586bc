"""Result-file writer / reader in the layout of the reference's save_results_pickle / load_results_pickle
(benchmark_SE3_tracking.py:272-345), as a NumPy archive."""
import numpy as np

from trajectory_optimization_matrix_lie_groups_amd import results_io


def test_round_trip_keeps_the_reference_layout(tmp_path):
    rng = np.random.default_rng(0)
    N, m = 7, 4
    xs = [[np.eye(4) + 0.01 * rng.normal(size=(4, 4)), rng.normal(size=6)] for _ in range(N + 1)]
    data = {
        "prob": {"J": np.eye(6), "dt": 0.004, "q_ref": rng.normal(size=(N + 1, 4, 4)), "xi_ref": rng.normal(size=(N + 1, 6)),
                 "x0": xs[0], "Q": np.eye(12), "P": 1.5 * np.eye(12), "R": 1e-5 * np.eye(m)},
        "ms_se3": {"xs": xs, "us": rng.normal(size=(N, m)), "J_hist": [3.0, 2.0, 1.5], "grad_hist": [np.float64(0.3), 0.1, 0.01],
                   "defect_hist": [5.0, 1e-14, 2e-14, 1e-14]},
        "ss_se3": {"xs": xs, "us": rng.normal(size=(N, m)), "J_hist": [3.0], "grad_hist": [0.5, 0.2]},
    }
    f = results_io.save_results(str(tmp_path / "r.npz"), data)
    back = results_io.load_results(f)
    assert set(back) == set(data)
    assert back["prob"]["dt"] == 0.004 and isinstance(back["prob"]["x0"], list) and len(back["prob"]["x0"]) == 2
    np.testing.assert_array_equal(back["prob"]["x0"][1], xs[0][1])
    assert len(back["ms_se3"]["xs"]) == N + 1
    for a, b in zip(back["ms_se3"]["xs"], xs):
        np.testing.assert_array_equal(a[0], b[0]); np.testing.assert_array_equal(a[1], b[1])
    assert back["ms_se3"]["defect_hist"] == data["ms_se3"]["defect_hist"]
    assert back["ss_se3"]["grad_hist"] == [0.5, 0.2] and "defect_hist" not in back["ss_se3"]
    np.testing.assert_array_equal(back["ms_se3"]["us"], data["ms_se3"]["us"])
    # nothing executable in the file
    with np.load(f, allow_pickle=False) as z:
        assert all(z[k].dtype == np.float64 for k in z.files)
