"""The series forms of csrc/tolg_lie.h (se3_exp_fast, se3_log_fast, so3_coef_fast, ljacinv_coef_fast, so3_exp_fast) against
a long-double reference, with arguments on both sides of every tier / domain boundary sharing a wavefront ("mixed-lane
gates"): a wave whose gate says "some lane needs the long tier / the closed form" must still give every lane the value its
own argument asks for, and a shared gate (the one lin_knot builds) must not shorten a series for a lane that needs it.
Through the C ABI (tolg_selftest_series)."""
import ctypes as C
import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from trajectory_optimization_matrix_lie_groups_amd import _capi  # noqa: E402

pytestmark = pytest.mark.gpu
LD = np.longdouble
EPS = 1e-10  # manif's small-angle switch (TOLG_EPS)


def _series(coefs, x):
    r = LD(0)
    for c in reversed(coefs):
        r = r * x + c
    return r


def _fact(n):
    f = LD(1)
    for k in range(2, n + 1):
        f *= k
    return f


def _coef_ref(th2):
    """a = (1-cos t)/t^2, b = (t-sin t)/t^3, c2 = (t^2+2cos t-2)/(2t^4), c3 = (2t-3sin t+t cos t)/(2t^5): Taylor series in
    long double below 1 rad (30 terms), closed forms above."""
    th2 = LD(th2)
    if th2 < 1.0:
        a = _series([LD((-1) ** k) / _fact(2 * k + 2) for k in range(30)], th2)
        b = _series([LD((-1) ** k) / _fact(2 * k + 3) for k in range(30)], th2)
        c2 = _series([LD((-1) ** k) / _fact(2 * k + 4) for k in range(30)], th2)
        c3 = _series([LD((-1) ** k) * (k + 1) / _fact(2 * k + 5) for k in range(30)], th2)
    else:
        t = np.sqrt(th2)
        s, c = np.sin(t), np.cos(t)
        a = (1 - c) / th2; b = (t - s) / (th2 * t)
        c2 = (th2 + 2 * c - 2) / (2 * th2 * th2); c3 = (2 * t - 3 * s + t * c) / (2 * th2 * th2 * t)
    return a, b, c2, c3


def _ljinv_ref(th2):
    """coefficient of W^2 in V^-1: 1/t^2 - (1+cos t)/(2 t sin t)"""
    th2 = LD(th2)
    if th2 < 0.3:
        # sum |B_{2k+2}| / (2k+2)! t^2k
        L = [LD(1) / 12, LD(1) / 720, LD(1) / 30240, LD(1) / 1209600, LD(1) / 47900160, LD(691) / 1307674368000,
             LD(1) / 74724249600, LD(3617) / 10670622842880000, LD(43867) / 5109094217170944000,
             LD(174611) / 802857662698291200000, LD(77683) / 14101100039391805440000]
        return _series(L, th2)
    t = np.sqrt(th2)
    return 1 / th2 - (1 + np.cos(t)) / (2 * t * np.sin(t))


def _skew(w):
    return np.array([[0, -w[2], w[1]], [w[2], 0, -w[0]], [-w[1], w[0], 0]], dtype=LD)


def _args():
    """Angles on both sides of every threshold of tolg_lie.h, log-spaced fill, each with a few axes; shuffled so that a
    wavefront of 64 consecutive argument sets mixes small / long / out-of-domain lanes."""
    rng = np.random.default_rng(7)
    edges = [1e-10, 0.01, 0.04, 1.0, 1.21]                       # th2 thresholds: eps, ljinv_small, exp/coef_small, exp_dom, coef/ljinv_dom
    edges += [(2 * np.arcsin(np.sqrt(y))) ** 2 for y in (1e-3, 0.0625, 0.25)]  # the Log's y thresholds as angles (+ the round-2 one)
    th2 = []
    for e in edges:
        th2 += [e * (1 - 1e-6), e * (1 + 1e-6), e * 0.9, e * 1.1]
    th2 += list(np.logspace(-14, np.log10(1.7), 420))
    th2 = np.array(th2)
    n = (len(th2) + 63) // 64 * 64
    th2 = np.r_[th2, rng.uniform(1e-4, 0.05, n - len(th2))]
    rows = []
    for mode in (0.0, 1.0):
        perm = rng.permutation(n)
        for t2 in th2[perm]:
            ax = rng.normal(size=3); ax /= np.linalg.norm(ax)
            w = ax * np.sqrt(t2)
            v = rng.uniform(-1, 1, 3)
            th2s = float(rng.choice([1e-12, 0.005, 0.0399, 0.0401, 0.2, 0.99, 1.2]))  # step rotation: all tiers
            rows.append(np.r_[w, v, th2s, mode])
    return np.array(rows)


def _rel(a, b, floor=1e-300):
    return float(abs(LD(a) - b) / max(abs(b), floor))


def test_series_forms_match_long_double_reference_with_mixed_lane_gates():
    lib = _capi.load()
    A = _args()
    n = A.shape[0]
    assert n % 64 == 0
    dev = torch.device("cuda")
    d_a = torch.as_tensor(A, device=dev).contiguous()
    d_o = torch.zeros((n, 24), dtype=torch.float64, device=dev)
    rc = lib.tolg_selftest_series(n, C.c_void_p(d_a.data_ptr()), C.c_void_p(d_o.data_ptr()), None)
    assert rc == 0
    torch.cuda.synchronize()
    O = d_o.cpu().numpy()
    worst = {}

    def chk(name, got, ref, tol, floor=1e-300):
        e = _rel(got, ref, floor)
        worst[name] = max(worst.get(name, 0.0), e)
        assert e <= tol, (name, got, float(ref), e)

    for a, o in zip(A, O):
        w, v, th2s, shared = a[:3].astype(LD), a[3:6].astype(LD), a[6], a[7] != 0
        th2 = float(np.dot(a[:3], a[:3]))
        # --- Exp: quaternion (sin(t/2)/t w, cos(t/2)), translation V(w) v
        t = np.sqrt(LD(th2))
        ca, cb, _, _ = _coef_ref(th2)
        so = np.sin(t / 2) / t if th2 > 0 else LD(0.5)
        W = _skew(w)
        if th2 > EPS:
            q_ref = np.r_[so * w, np.cos(t / 2)]
            t_ref = v + ca * (W @ v) + cb * (W @ (W @ v))
        else:  # manif's small-angle Exp: q = normalise(w/2, 1), V = I + W/2
            q_ref = np.r_[w / 2, LD(1)]; q_ref = q_ref / np.sqrt(np.dot(q_ref, q_ref))
            t_ref = v + LD(0.5) * (W @ v)
        for k in range(4):
            chk("exp.q", o[6 + k], q_ref[k], 4e-16, floor=1.0)
        for k in range(3):
            chk("exp.t", o[10 + k], t_ref[k], 1e-15, floor=1.0)
        # --- Log(Exp(w, v)) = (w, v) up to the conditioning of the round trip (angles < 1.3 rad)
        for k in range(3):
            # (|q_v|^2 <= 1e-10, i.e. angle^2 <= 4e-10, takes manif's small-angle Log 2 q_v and V^-1 = I - W/2)
            chk("log.w", o[13 + k], w[k], 2e-15 if th2 > 5e-10 else 1e-9, floor=max(float(t), 1e-8))
            chk("log.v", o[16 + k], v[k], 3e-14 if th2 > 5e-10 else 1e-9, floor=1.0)
        # --- coefficient series at the rotation's own angle (own gates) or at the Log's angle (shared gate)
        th2k = float(np.dot(o[13:16], o[13:16])) if shared else th2
        ra, rb, rc2, rc3 = _coef_ref(th2k)
        if th2k > EPS:
            chk("coef.a", o[0], ra, 5e-16); chk("coef.b", o[1], rb, 5e-16)
            chk("coef.c2", o[3], rc2, 6e-15 if th2k > 1.0 else 5e-16); chk("coef.c3", o[4], rc3, 3e-13 if th2k > 1.0 else 5e-16)
            chk("ljinv", o[5], _ljinv_ref(th2k), 2e-14 if th2k > 0.26 else 5e-16)
        else:
            assert o[0] == 0.5 and o[1] == 0.0 and o[5] == 0.0
        chk("coef.c1", o[2], rb if th2k > EPS else LD(1) / 6, 5e-16 if th2k > EPS else 1e-10)
        # --- step rotation about the same axis, angle^2 = th2s
        ts = np.sqrt(LD(th2s))
        ax = (w / t) if th2 > 0 else w
        qs = np.r_[np.sin(ts / 2) * ax, np.cos(ts / 2)] if th2s > EPS else np.r_[ts * ax / 2, LD(1)]
        if th2 > 0:
            for k in range(4):
                chk("so3exp.q", o[19 + k], qs[k], 5e-16 if th2s > EPS else 1e-12, floor=1.0)
            chk("coef.a(step)", o[23], _coef_ref(th2s)[0] if th2s > EPS else LD(0.5), 5e-16)
    print({k: "%.1e" % v for k, v in worst.items()})
