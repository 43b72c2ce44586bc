"""The rollouts of a line-search stage in two wavefronts per sixteen quads (k_rollout_ls2, round 4): one wave carries the twist
chain (Log of the deviation -> control -> next twist), a second one the pose chain (Exp of the twist -> next pose), one step
ahead, handing twist and pose over through an LDS ring; the merit search's factors (traopt_controller.py:2713-2716) are formed
off the chain.  `TOLG_LS_ONEWAVE=1` (read at every stage) keeps the one-wave forms (K3 for the first try, k_rollout_ls): the
two must take the same decisions -- step size per iteration, exit code, iteration count -- and agree to rounding (each wave
gates its own series evaluation; the one-wave step gates Log and Exp together), and both with the oracle.  Covered: the merit
search (alpha < 1 steps with the factors) and the backtracking search, stages on the flags (first try) and on compacted lists,
lists shorter than a wave and batches that are not a multiple of 16, the drone (m = 4, gravity: the twist chain reads the pose
it was handed), a dense inertia block, trajectories that find no step."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import bridge as ob  # noqa: E402  (test infrastructure)
from trajectory_optimization_matrix_lie_groups_amd import BatchedTrackingILQR, TrackingProblem, workloads  # noqa: E402


def _rel(a, b):
    a = np.nan_to_num(np.asarray(a)); b = np.nan_to_num(np.asarray(b))
    return np.abs(a - b).max() / max(np.abs(b).max(), 1e-300)


def _fit(prob, x0_q, x0_xi, us0, K, onewave, **kw):
    old = os.environ.pop("TOLG_LS_ONEWAVE", None)
    try:
        if onewave:
            os.environ["TOLG_LS_ONEWAVE"] = "1"
        r = BatchedTrackingILQR(prob, x0_q.shape[0]).fit_batch(x0_q, x0_xi, us0, n_iterations=K, tol_grad_norm=0.0, tol_d_norm=0.0, **kw)
        torch.cuda.synchronize()
        return r
    finally:
        os.environ.pop("TOLG_LS_ONEWAVE", None)
        if old is not None:
            os.environ["TOLG_LS_ONEWAVE"] = old


def _same(ra, rb, tol=1e-10):
    """Same decisions, costs to rounding.  Where the two forms stop at different iterations the solve must be AT its floor there
    (the search then compares costs that differ in the last digits): the last cost decrease both saw is below 1e-11 of the cost."""
    ia, ib = ra.iters.cpu().numpy(), rb.iters.cpu().numpy()
    sa, sb = ra.status.cpu().numpy(), rb.status.cpu().numpy()
    Ja, Jb = ra.J_hist.cpu().numpy(), rb.J_hist.cpu().numpy()
    aa, ab = ra.alpha_hist.cpu().numpy(), rb.alpha_hist.cpu().numpy()
    n = np.minimum(ia, ib)
    K = Ja.shape[1]
    mask = np.arange(K)[None, :] < n[:, None]
    assert np.array_equal(np.isnan(Ja) & mask, np.isnan(Jb) & mask)
    assert _rel(np.where(mask, Ja, 0.0), np.where(mask, Jb, 0.0)) < tol
    # the step sizes of the common iterations that still moved the cost (at the floor the accepted step size is a coin flip too)
    fork = (ia != ib) | (sa != sb)
    with np.errstate(all="ignore"):
        moved = np.ones_like(mask)
        moved[:, 1:] = np.abs(Ja[:, 1:] - Ja[:, :-1]) > 1e-10 * np.abs(Ja[:, 1:])
    m2 = mask & moved & ~(fork[:, None] & (np.arange(K)[None, :] >= (n - 1)[:, None]))
    assert np.array_equal(np.nan_to_num(np.where(m2, aa, 0.0)), np.nan_to_num(np.where(m2, ab, 0.0)))
    for b in np.nonzero(fork)[0]:
        k = int(n[b])
        assert k >= 2, (b, ia[b], ib[b])
        assert abs(Ja[b, k - 1] - Ja[b, k - 2]) <= 1e-11 * abs(Ja[b, k - 1]), (b, ia[b], ib[b], Ja[b, :k])
    assert fork.mean() <= 0.25
    ok = ~fork
    assert _rel(ra.us.cpu().numpy()[ok], rb.us.cpu().numpy()[ok]) < 1e-8
    assert _rel(ra.xs_q.cpu().numpy()[ok], rb.xs_q.cpu().numpy()[ok]) < 1e-8


def _oracle(prob, x0_q, x0_xi, us0, K, mode):
    op = ob.OracleProblem(prob.kind, prob.J, prob.dt, prob.Q, prob.R, prob.P, prob.q_ref, prob.xi_ref)
    return ob.fit_batch(op, x0_q, x0_xi, us0, mode=mode, max_iter=K, line_search=True)


@pytest.mark.parametrize("kind,mode,B,N,K", [("se3", "ms", 37, 45, 8), ("se3", "ss", 21, 30, 8), ("drone", "ms", 10, 60, 2),
                                              ("drone", "ss", 18, 40, 6), ("se3", "ms", 100, 25, 12)])
def test_two_wave_rollouts_agree_with_the_one_wave_forms(kind, mode, B, N, K):
    make = {"se3": workloads.se3_tracking, "drone": workloads.drone_tracking}[kind]
    prob, x0_q, x0_xi, us0 = make(B, N=N)
    ra = _fit(prob, x0_q, x0_xi, us0, K, False, mode=mode, line_search=True)
    rb = _fit(prob, x0_q, x0_xi, us0, K, True, mode=mode, line_search=True)
    _same(ra, rb)
    o = _oracle(prob, x0_q, x0_xi, us0, K, mode)
    n = int(min(o["iters"].min(), ra.iters.min().item()))
    assert n >= 1
    assert _rel(ra.J_hist.cpu().numpy()[:, :n], o["J_hist"][:, :n]) < 1e-9
    same_exit = (ra.iters.cpu().numpy() == o["iters"]) & (ra.status.cpu().numpy() == o["status"])
    assert same_exit.mean() > 0.9


def test_mixed_outcomes_in_one_batch():
    """Wild initial controls on every third trajectory: their searches go deep into the step-size list or end without a step while
    their neighbours accept the first one -- the later stages run on compacted lists of a few quads."""
    B, N, K = 48, 30, 10
    prob, x0_q, x0_xi, us0 = workloads.se3_tracking(B, N=N)
    rng = np.random.default_rng(5)
    us0 = us0.copy()
    us0[::3] += 40.0 * rng.standard_normal(us0[::3].shape)
    for mode in ("ms", "ss"):
        ra = _fit(prob, x0_q, x0_xi, us0, K, False, mode=mode, line_search=True)
        rb = _fit(prob, x0_q, x0_xi, us0, K, True, mode=mode, line_search=True)
        _same(ra, rb, tol=1e-9)
        al = np.nan_to_num(ra.alpha_hist.cpu().numpy(), nan=1.0)
        assert (al < 1.0).any() and (al == 1.0).any()   # both kinds of search did occur
        o = _oracle(prob, x0_q, x0_xi, us0, K, mode)
        Jg, Jo = ra.J_hist.cpu().numpy(), o["J_hist"]
        n = np.minimum(o["iters"], ra.iters.cpu().numpy())
        mask = np.arange(K)[None, :] < n[:, None]
        with np.errstate(all="ignore"):
            d = np.where(mask, np.abs(Jg - Jo) / np.abs(Jo), 0.0)
        d = np.where(np.isfinite(d), d, 0.0)
        # (a wild trajectory may take another branch of a search at rounding level; the tame two thirds must agree)
        tame = np.ones(B, bool); tame[::3] = False
        assert d[tame].max() < 1e-9


def test_dense_inertia():
    B, N, K = 20, 30, 6
    prob, x0_q, x0_xi, us0 = workloads.se3_tracking(B, N=N)
    # full inertia blocks (a rotated body frame's inertia), block-diagonal as the reference's G assumes
    A = np.array([[0.10, -0.05, 0.02], [0.03, 0.12, -0.04], [-0.02, 0.06, 0.09]])
    Jd = np.array(prob.J, dtype=float).copy()
    Jd[:3, :3] += A @ A.T
    Jd[3:, 3:] += 0.5 * (A @ A.T)
    pd = TrackingProblem("se3", Jd, prob.dt, prob.Q, prob.R, prob.P, prob.q_ref, prob.xi_ref)
    ra = _fit(pd, x0_q, x0_xi, us0, K, False, mode="ms", line_search=True)
    rb = _fit(pd, x0_q, x0_xi, us0, K, True, mode="ms", line_search=True)
    _same(ra, rb)
    o = _oracle(pd, x0_q, x0_xi, us0, K, "ms")
    n = int(min(o["iters"].min(), ra.iters.min().item()))
    assert _rel(ra.J_hist.cpu().numpy()[:, :n], o["J_hist"][:, :n]) < 1e-9


@pytest.mark.parametrize("mode", ["ms", "ss"])
def test_pendulum_stages_in_two_waves(mode):
    """Pendulum3dDyanmics (k_rollout_ls2<6, ., 1, PK = 1>: the twist chain reads the pose it was handed for the gravity torque):
    two-wave against one-wave stage kernels (the oracle comparison of these solves is tests/test_pendulum.py's)."""
    B, K = 11, 10
    prob, x0_q, x0_xi, us0 = workloads.pendulum_swingup(B, xi0_scale=1.0)
    ra = _fit(prob, x0_q, x0_xi, us0, K, False, mode=mode, line_search=True)
    rb = _fit(prob, x0_q, x0_xi, us0, K, True, mode=mode, line_search=True)
    _same(ra, rb)
