"""Randomised parity cases (tools/parity_fuzz.py): model kind, inertia structure, solver mode, line search, rollout form,
batch, horizon, weights over four decades, time step and the spread of the initial data drawn at random, every trajectory
that stays in a regime where two fp64 implementations can agree compared with the oracle.  Sixty fixed seeds here; the
tool runs any number (300 at the end of round 3: profiles/r03_s3_parity_fuzz.txt)."""
import os
import sys

import pytest

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("first", [2000, 2020, 2040])
def test_random_cases_match_oracle(first):
    import parity_fuzz as pf
    compared = 0
    for seed in range(first, first + 20):
        cfg, worst_j, worst_u, notes, stats = pf.one(seed)
        assert not notes, (seed, cfg, notes)
        assert worst_j <= pf.TOL_J and worst_u <= pf.TOL_U, (seed, cfg, worst_j, worst_u)
        compared += sum(v for k, v in stats.items() if k in (0, 1, 2, 3)) - stats.get("wild", 0)
    assert compared >= 40  # most trajectories of twenty cases are comparable


@pytest.mark.parametrize("seed", [50312, 50349, 40265])
def test_seeds_the_campaigns_flagged(seed):
    """Seeds 50312 / 50349 (round 4, final campaign): SO(3) accept-always solves sixteen orders of magnitude into a divergence, in
    which the antisymmetric part of V outgrew the four knots between two symmetrisations of the backward sweep and V_SS came out
    indefinite where the reference's (symmetrised at every knot, traopt_controller.py:3004) is positive definite -- the fast sweep
    now hands such a group to the full kernel, which symmetrises at every knot (csrc/tolg_backward3.h).  Seed 40265: a search at
    its cost floor under mu = 3.4e10 whose last decision is a coin flip between fp64, fp64 and long double: the exit codes may
    differ there, the costs may not."""
    import parity_fuzz as pf
    cfg, worst_j, worst_u, notes, stats = pf.one(seed)
    assert worst_j <= pf.TOL_J and worst_u <= pf.TOL_U, (seed, cfg, worst_j, worst_u)
    if seed != 40265:
        assert not notes, (seed, cfg, notes)
