"""Host side of the batched solver: PyTorch-ROCm tensors in, C ABI (include/tolg.h) underneath.

PyTorch is plumbing here (device memory, streams, torch.distributed); every flop of the hot path
runs in the hand-written HIP kernels of csrc/tolg_kernels.hip.
"""
import ctypes as C
from dataclasses import dataclass, field
from typing import Optional

import numpy as np
import torch

from . import _capi

_KIND = {"se3": _capi.DYN_SE3, "rigidbody": _capi.DYN_RIGIDBODY, "drone": _capi.DYN_DRONE, "so3": _capi.DYN_SO3,
         "pendulum3d": _capi.DYN_PENDULUM3D}


@dataclass
class TrackingProblem:
    """One (dynamics, cost) pair shared by the batch.

    Mirrors the constructor arguments of SE3Dynamics / RigidBodyDynamics / DroneDynamics
    (reference traoptlibrary/traopt_dynamics.py:633, :906, :1214) and
    SE3TrackingQuadraticGaussNewtonCost (traoptlibrary/traopt_cost.py:587)."""
    kind: str
    J: np.ndarray
    dt: float
    Q: np.ndarray
    R: np.ndarray
    P: np.ndarray
    q_ref: np.ndarray   # (N+1, 4, 4)
    xi_ref: np.ndarray  # (N+1, 6)
    pend_mass: float = 0.0    # Pendulum3dDyanmics m, length (traopt_dynamics.py:425); other kinds ignore them
    pend_length: float = 0.0

    @property
    def N(self):
        return int(np.asarray(self.q_ref).shape[0]) - 1

    @property
    def m(self):
        return 4 if self.kind == "drone" else 6


def embed_so3(J3, dt, Q6, R3, P6, R_ref, w_ref) -> TrackingProblem:
    """SO(3) tracking problem in the SE(3) layout of the C ABI (include/tolg.h, TOLG_DYN_SO3):
    zero translation / linear velocity / inputs 3..5, J = blkdiag(J_so3, I3), R = blkdiag(R_so3, I3)."""
    R_ref = np.asarray(R_ref, float); w_ref = np.asarray(w_ref, float)
    n = R_ref.shape[0]
    J = np.eye(6); J[:3, :3] = J3
    Q = np.zeros((12, 12)); Q[:3, :3] = np.asarray(Q6)[:3, :3]; Q[6:9, 6:9] = np.asarray(Q6)[3:, 3:]
    P = np.zeros((12, 12)); P[:3, :3] = np.asarray(P6)[:3, :3]; P[6:9, 6:9] = np.asarray(P6)[3:, 3:]
    R = np.eye(6); R[:3, :3] = R3
    q_ref = np.tile(np.eye(4), (n, 1, 1)); q_ref[:, :3, :3] = R_ref
    xi_ref = np.zeros((n, 6)); xi_ref[:, :3] = w_ref
    return TrackingProblem("so3", J, float(dt), Q, R, P, q_ref, xi_ref)


def embed_pendulum3d(J3, mass, length, dt, Q6, R3, P6, R_ref, w_ref) -> TrackingProblem:
    """Pendulum3dDyanmics (reference traoptlibrary/traopt_dynamics.py:421-626) with the SO3 tracking cost, in
    the same embedding as embed_so3 (include/tolg.h, TOLG_DYN_PENDULUM3D); the pivot acceleration is u[0:3]."""
    p = embed_so3(J3, dt, Q6, R3, P6, R_ref, w_ref)
    p.kind, p.pend_mass, p.pend_length = "pendulum3d", float(mass), float(length)
    return p


@dataclass
class FitResult:
    xs_q: torch.Tensor      # [B, N+1, 4, 4]
    xs_xi: torch.Tensor     # [B, N+1, 6]
    us: torch.Tensor        # [B, N, m]
    J_hist: torch.Tensor    # [B, max_iter]
    grad_hist: torch.Tensor  # [B, max_iter+1]
    defect_hist: torch.Tensor  # [B, max_iter+1]
    alpha_hist: torch.Tensor
    mu_hist: torch.Tensor
    iters: torch.Tensor     # [B] int32
    status: torch.Tensor    # [B] int32
    converged: torch.Tensor  # [B] int32
    extra: dict = field(default_factory=dict)


def _ptr(t: Optional[torch.Tensor]):
    return C.c_void_p(0 if t is None else t.data_ptr())


class BatchedTrackingILQR:
    """Batched iLQR_Tracking_SE3_MS / iLQR_Tracking_SE3 on one GPU (one process per GPU).

    The object owns a device workspace tensor (allocated once, here) and an opaque C handle; the
    solve calls allocate nothing inside the library."""

    def __init__(self, problem: TrackingProblem, max_batch: int, device=None):
        if not torch.cuda.is_available():
            raise RuntimeError("no GPU visible: the batched solver is HIP-only (there is no CPU fallback)")
        self.lib = _capi.load()
        self.device = torch.device(device if device is not None else "cuda:%d" % torch.cuda.current_device())
        self.problem = problem
        self.N, self.m = problem.N, problem.m
        self.max_batch = int(max_batch)
        p = _capi.Problem()
        p.kind, p.m, p.N, p.dt = _KIND[problem.kind], self.m, self.N, float(problem.dt)
        p.pend_mass, p.pend_length = float(problem.pend_mass), float(problem.pend_length)
        p.J[:] = list(np.asarray(problem.J, dtype=np.float64).reshape(36))
        p.Q[:] = list(np.asarray(problem.Q, dtype=np.float64).reshape(144))
        p.P[:] = list(np.asarray(problem.P, dtype=np.float64).reshape(144))
        Rm = np.zeros(36)
        Rm[: self.m * self.m] = np.asarray(problem.R, dtype=np.float64).reshape(-1)
        p.R[:] = list(Rm)
        self._p = p
        nbytes = self.lib.tolg_workspace_bytes(C.byref(p), self.max_batch)
        if nbytes == 0:
            raise ValueError("invalid problem description")
        self.workspace_bytes = int(nbytes)
        with torch.cuda.device(self.device):
            self._ws = torch.empty(nbytes + 256, dtype=torch.uint8, device=self.device)
            off = (-self._ws.data_ptr()) % 256
            self._ws_ptr = self._ws.data_ptr() + off
            self._q_ref = torch.as_tensor(np.ascontiguousarray(problem.q_ref, dtype=np.float64).reshape(self.N + 1, 16),
                                          device=self.device)
            self._xi_ref = torch.as_tensor(np.ascontiguousarray(problem.xi_ref, dtype=np.float64).reshape(self.N + 1, 6),
                                           device=self.device)
            h = C.c_void_p()
            rc = self.lib.tolg_create(C.byref(p), _ptr(self._q_ref), _ptr(self._xi_ref), self.max_batch,
                                      C.c_void_p(self._ws_ptr), C.c_size_t(nbytes), self._stream(), C.byref(h))
        _capi.check(rc, "tolg_create")
        self._h = h

    def __del__(self):
        h = getattr(self, "_h", None)
        if h:
            try:
                torch.cuda.synchronize(self.device)
            except Exception:
                pass
            self.lib.tolg_destroy(h)
            self._h = None

    def _stream(self):
        return C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)

    def _dev(self, a, shape):
        t = torch.as_tensor(a, dtype=torch.float64, device=self.device)
        return t.reshape(shape).contiguous()

    # ------------------------------------------------------------------------------------------
    def _alloc_result(self, B, K, histories=True):
        f64 = dict(dtype=torch.float64, device=self.device)
        nanf = lambda *shape: torch.full(shape, float("nan"), **f64) if histories else None  # noqa: E731
        return FitResult(
            xs_q=torch.empty(B, self.N + 1, 4, 4, **f64), xs_xi=torch.empty(B, self.N + 1, 6, **f64),
            us=torch.empty(B, self.N, self.m, **f64), J_hist=nanf(B, K), grad_hist=nanf(B, K + 1),
            defect_hist=nanf(B, K + 1), alpha_hist=nanf(B, K), mu_hist=nanf(B, K),
            iters=torch.zeros(B, dtype=torch.int32, device=self.device),
            status=torch.zeros(B, dtype=torch.int32, device=self.device),
            converged=torch.zeros(B, dtype=torch.int32, device=self.device))

    def solve_begin(self, x0_q, x0_xi, us_init=None, mode="ms", n_iterations=100, tol_grad_norm=1e-6,
                    tol_d_norm=1e-6, line_search=False, rollout="nonlinear", max_reg=1e10, histories=True,
                    out: Optional[FitResult] = None, schedule="auto") -> FitResult:
        """_initial_guess + first _linearization; leaves the batch resident in HBM.
        schedule: "auto" (rollout and re-linearisation fused in one launch where the mode allows it) or
        "split" (separate launches); launch structure only, same algorithm."""
        x0_q = self._dev(x0_q, (-1, 16))
        B = x0_q.shape[0]
        x0_xi = self._dev(x0_xi, (B, 6))
        if us_init is None:
            us_init = torch.zeros(B, self.N, self.m, dtype=torch.float64, device=self.device)
        us_init = self._dev(us_init, (B, self.N, self.m))
        K = int(n_iterations)
        if out is None:
            out = self._alloc_result(B, K, histories)
        o = _capi.Options(_capi.MODE_MS if mode == "ms" else _capi.MODE_SS, K, int(bool(line_search)),
                          int(rollout == "linear"), float(tol_grad_norm), float(tol_d_norm),
                          float(max_reg if max_reg else 0.0),
                          {"auto": _capi.SCHED_AUTO, "split": _capi.SCHED_SPLIT}[schedule], 0)
        with torch.cuda.device(self.device):
            rc = self.lib.tolg_solve_begin(self._h, C.byref(o), B, _ptr(x0_q), _ptr(x0_xi), _ptr(us_init),
                                           _ptr(out.J_hist), _ptr(out.grad_hist), _ptr(out.defect_hist),
                                           _ptr(out.alpha_hist), _ptr(out.mu_hist), self._stream())
        _capi.check(rc, "tolg_solve_begin")
        self._inflight = (out, (x0_q, x0_xi, us_init))  # keep the inputs alive until the stream has used them
        return out

    def solve_iterate(self, n_iter):
        """n_iter passes of the iteration body (backward sweep, rollout, re-linearisation)."""
        with torch.cuda.device(self.device):
            rc = self.lib.tolg_solve_iterate(self._h, int(n_iter), self._stream())
        _capi.check(rc, "tolg_solve_iterate")

    def solve_iterate_until(self, n_iter, check_every) -> int:
        """Up to n_iter iterations in slices of check_every, stopping once no trajectory is iterating any more
        (tolg_solve_iterate_until: the read-back of one slice overlaps the next).  Returns the iterations queued."""
        n = C.c_int32(0)
        with torch.cuda.device(self.device):
            rc = self.lib.tolg_solve_iterate_until(self._h, int(n_iter), int(check_every), C.byref(n), self._stream())
        _capi.check(rc, "tolg_solve_iterate_until")
        return int(n.value)

    def active_count(self) -> int:
        """Trajectories of the solve in flight that are still being iterated (one small kernel + a host read)."""
        if getattr(self, "_active_buf", None) is None:
            self._active_buf = torch.zeros(1, dtype=torch.int32, device=self.device)
        with torch.cuda.device(self.device):
            rc = self.lib.tolg_solve_active_count(self._h, _ptr(self._active_buf), self._stream())
        _capi.check(rc, "tolg_solve_active_count")
        return int(self._active_buf.item())

    def solve_peek(self) -> FitResult:
        """Export the trajectories in flight (xs, us, iters, status) without ending the solve."""
        out = self._inflight[0]
        with torch.cuda.device(self.device):
            rc = self.lib.tolg_solve_peek(self._h, _ptr(out.xs_q), _ptr(out.xs_xi), _ptr(out.us), _ptr(out.iters),
                                          _ptr(out.status), _ptr(out.converged), self._stream())
        _capi.check(rc, "tolg_solve_peek")
        return out

    def solve_end(self) -> FitResult:
        out = self._inflight[0]
        with torch.cuda.device(self.device):
            rc = self.lib.tolg_solve_end(self._h, _ptr(out.xs_q), _ptr(out.xs_xi), _ptr(out.us), _ptr(out.iters),
                                         _ptr(out.status), _ptr(out.converged), self._stream())
        _capi.check(rc, "tolg_solve_end")
        return out

    def fit_batch(self, x0_q, x0_xi, us_init=None, mode="ms", n_iterations=100, tol_grad_norm=1e-6,
                  tol_d_norm=1e-6, line_search=False, rollout="nonlinear", max_reg=1e10,
                  histories=True, out: Optional[FitResult] = None, schedule="auto", check_every=16) -> FitResult:
        """B independent fits (the reference's joblib fan-out, visualization/perturb_all_compute.py:240).
        Inputs may be numpy arrays or tensors already on the device; outputs are device tensors.
        The iterations are issued in slices of `check_every`; behind each slice the number of trajectories still
        iterating is read back (overlapped with the next slice) and the loop stops when it reaches zero (the early
        exit of traopt_controller.py:2528-2532 for the whole batch).  check_every=0, or tolerances of zero, issue
        all n_iterations without a host read.  `iterations_issued` keeps how many were queued."""
        self.solve_begin(x0_q, x0_xi, us_init, mode, n_iterations, tol_grad_norm, tol_d_norm, line_search, rollout,
                         max_reg, histories, out, schedule)
        self.iterations_issued = self.solve_iterate_until(int(n_iterations), int(check_every or 0))
        return self.solve_end()

    def solve_batch_one_call(self, x0_q, x0_xi, us_init=None, mode="ms", n_iterations=100, tol_grad_norm=1e-6,
                             tol_d_norm=1e-6, line_search=False, rollout="nonlinear", max_reg=1e10, schedule="auto",
                             check_every=0) -> FitResult:
        """The same fit through the single entry point tolg_solve_batch (what a C / C++ caller binds): begin,
        iterations (tolg_options.check_every: 0 = all of them, never synchronising), end in one call."""
        x0_q = self._dev(x0_q, (-1, 16))
        B = x0_q.shape[0]
        x0_xi = self._dev(x0_xi, (B, 6))
        if us_init is None:
            us_init = torch.zeros((B, self.N, self.m), dtype=torch.float64, device=self.device)
        us_init = self._dev(us_init, (B, self.N, self.m))
        K = int(n_iterations)
        out = self._alloc_result(B, K, True)
        o = _capi.Options(_capi.MODE_MS if mode == "ms" else _capi.MODE_SS, K, int(bool(line_search)),
                          int(rollout == "linear"), float(tol_grad_norm), float(tol_d_norm),
                          float(max_reg if max_reg else 0.0),
                          {"auto": _capi.SCHED_AUTO, "split": _capi.SCHED_SPLIT}[schedule], int(check_every))
        with torch.cuda.device(self.device):
            rc = self.lib.tolg_solve_batch(self._h, C.byref(o), B, _ptr(x0_q), _ptr(x0_xi), _ptr(us_init), _ptr(out.xs_q),
                                           _ptr(out.xs_xi), _ptr(out.us), _ptr(out.J_hist), _ptr(out.grad_hist),
                                           _ptr(out.defect_hist), _ptr(out.alpha_hist), _ptr(out.mu_hist), _ptr(out.iters),
                                           _ptr(out.status), _ptr(out.converged), self._stream())
        _capi.check(rc, "tolg_solve_batch")
        torch.cuda.current_stream(self.device).synchronize()  # the inputs above must outlive the queued work
        return out

    # ------------------------------------------------------------------------------------------
    def set_al(self, lb=None, ub=None, lam=None, imu=None):
        """Attach (or detach with lb=None) the augmented-Lagrangian box input constraint terms
        (ALConstrainedCost + InputConstraint).  lam, imu: device tensors [B, N, 2m]."""
        if lb is None:
            self._al = None
            rc = self.lib.tolg_set_al(self._h, None, None, None, None)
        else:
            lb = self._dev(lb, (self.m,)); ub = self._dev(ub, (self.m,))
            self._al = (lb, ub, lam, imu)  # keep alive
            rc = self.lib.tolg_set_al(self._h, _ptr(lb), _ptr(ub), _ptr(lam), _ptr(imu))
        _capi.check(rc, "tolg_set_al")

    def al_fit_batch(self, x0_q, x0_xi, us_init, lb, ub, n_al_iters=100, n_ilqr_iters=200, tol_grad_norm=1e-6,
                     tol_d_norm=1e-6, tol_constr=1e-2, mu0=1e-2, mu_scale=10.0, mu_max=1e8, line_search=False,
                     on_outer=None):
        """AL_iLQR_Tracking_SE3_MS.fit (reference traoptlibrary/traopt_controller.py:3218-3267) for B
        independent problems: every outer iteration re-solves from (x0, us_init) -- no warm start, as in
        the reference -- then updates multipliers on the device.  Returns (FitResult, info dict)."""
        x0_q = self._dev(x0_q, (-1, 16))
        B = x0_q.shape[0]
        f64 = dict(dtype=torch.float64, device=self.device)
        lam = torch.zeros(B, self.N, 2 * self.m, **f64)
        imu = torch.full((B, self.N, 2 * self.m), float(mu0), **f64)
        mu = torch.full((B,), float(mu0), **f64)
        maxviol = torch.zeros(B, **f64)
        alconv = torch.zeros(B, dtype=torch.int32, device=self.device)
        lb_d = self._dev(lb, (self.m,)); ub_d = self._dev(ub, (self.m,))
        self.set_al(lb_d, ub_d, lam, imu)
        outer = 0
        res = None
        final = None
        try:
            for outer in range(int(n_al_iters)):
                res = self.fit_batch(x0_q, x0_xi, us_init, mode="ms", n_iterations=n_ilqr_iters,
                                     tol_grad_norm=tol_grad_norm, tol_d_norm=tol_d_norm, line_search=line_search,
                                     rollout="nonlinear")
                if final is None:
                    final = res
                else:  # problems that had already converged keep the result of their converging solve
                    keep = alconv.bool()
                    for name in ("xs_q", "xs_xi", "us", "J_hist", "grad_hist", "defect_hist", "alpha_hist",
                                 "mu_hist", "iters", "status", "converged"):
                        new, old = getattr(res, name), getattr(final, name)
                        if new is not None:
                            new[keep] = old[keep]
                    final = res
                if on_outer is not None:  # before the multiplier update, like on_iteration_al (:3253-3259)
                    on_outer(outer, final, lam, imu, mu)
                with torch.cuda.device(self.device):
                    rc = self.lib.tolg_al_update(self._h, B, _ptr(final.us), _ptr(lb_d), _ptr(ub_d), _ptr(lam), _ptr(imu),
                                                 _ptr(mu), float(mu_scale), float(mu_max), float(tol_constr),
                                                 _ptr(maxviol), _ptr(alconv), self._stream())
                _capi.check(rc, "tolg_al_update")
                if bool(alconv.all().item()):
                    break
        finally:
            self.set_al(None)
        return final, dict(lmbd=lam, Imu=imu, mu=mu, max_violation=maxviol, al_converged=alconv, outer_iterations=outer + 1)

    # ------------------------------------------------------------------------------------------
    def linearize_backward(self, xs_q, xs_xi, us, ms=True, mu=1.0, delta=2.0, max_reg=1e10):
        """One _linearization + _backward_pass (+ gradient norm) on given trajectories."""
        xs_q = self._dev(xs_q, (-1, self.N + 1, 16))
        B = xs_q.shape[0]
        xs_xi = self._dev(xs_xi, (B, self.N + 1, 6))
        us = self._dev(us, (B, self.N, self.m))
        f64 = dict(dtype=torch.float64, device=self.device)
        md = torch.empty(B, 2, **f64)
        md[:, 0] = mu
        md[:, 1] = delta
        r = dict(Fx=torch.empty(B, self.N, 12, 12, **f64), d=torch.empty(B, self.N, 12, **f64),
                 lx=torch.empty(B, self.N + 1, 12, **f64), lxx11=torch.empty(B, self.N + 1, 6, 6, **f64),
                 k=torch.empty(B, self.N, self.m, **f64), K=torch.empty(B, self.N, self.m, 12, **f64),
                 J=torch.empty(B, **f64), dnorm=torch.empty(B, **f64), grad=torch.empty(B, **f64), mu_delta=md)
        with torch.cuda.device(self.device):
            rc = self.lib.tolg_linearize_backward(self._h, int(ms), float(max_reg), B, _ptr(xs_q), _ptr(xs_xi), _ptr(us),
                                                  _ptr(md), _ptr(r["Fx"]), _ptr(r["d"]), _ptr(r["lx"]), _ptr(r["lxx11"]),
                                                  _ptr(r["k"]), _ptr(r["K"]), _ptr(r["J"]), _ptr(r["dnorm"]),
                                                  _ptr(r["grad"]), self._stream())
        _capi.check(rc, "tolg_linearize_backward")
        return r

    def eval_knot(self, i, x_q, x_xi, u=None):
        """Per-knot plugin quantities (f, f_x, f_u, l, l_x, l_xx, l_u, l_uu, err) for n states at knot i."""
        x_q = self._dev(x_q, (-1, 16))
        n = x_q.shape[0]
        x_xi = self._dev(x_xi, (n, 6))
        term = int(i) == self.N
        u_d = None if (u is None or term) else self._dev(u, (n, self.m))
        f64 = dict(dtype=torch.float64, device=self.device)
        r = dict(l=torch.zeros(n, **f64), lx=torch.zeros(n, 12, **f64), lxx=torch.zeros(n, 12, 12, **f64),
                 err=torch.zeros(n, 12, **f64))
        if not term:
            r.update(f_q=torch.zeros(n, 4, 4, **f64), f_xi=torch.zeros(n, 6, **f64), Fx=torch.zeros(n, 12, 12, **f64),
                     Fu=torch.zeros(n, 12, self.m, **f64), lu=torch.zeros(n, self.m, **f64),
                     luu=torch.zeros(n, self.m, self.m, **f64))
        g = lambda k: _ptr(r.get(k))  # noqa: E731
        with torch.cuda.device(self.device):
            rc = self.lib.tolg_eval_knot(self._h, int(i), n, _ptr(x_q), _ptr(x_xi), _ptr(u_d), g("f_q"), g("f_xi"),
                                         g("Fx"), g("Fu"), g("l"), g("lx"), g("lxx"), g("lu"), g("luu"), g("err"),
                                         self._stream())
        _capi.check(rc, "tolg_eval_knot")
        return r

    def rollout(self, B, alpha=1.0, ms=True, rollout="nonlinear"):
        """Closed-loop rollout with the gains of the preceding linearize_backward call."""
        f64 = dict(dtype=torch.float64, device=self.device)
        xs_q = torch.empty(B, self.N + 1, 4, 4, **f64)
        xs_xi = torch.empty(B, self.N + 1, 6, **f64)
        us = torch.empty(B, self.N, self.m, **f64)
        with torch.cuda.device(self.device):
            rc = self.lib.tolg_rollout(self._h, int(ms), int(rollout == "linear"), float(alpha), B, _ptr(xs_q),
                                       _ptr(xs_xi), _ptr(us), self._stream())
        _capi.check(rc, "tolg_rollout")
        return xs_q, xs_xi, us

    def expected_change(self, B, form="auto"):
        """_expected_cost_change of the linear alpha = 1 rollout (traopt_controller.py:2550-2552, :2756-2769) with the
        records and gains of the preceding linearize_backward(ms=True) call.  form: "statement" (the kernel that walks
        the reference's statements), "ring" (the affine recursion alone; flag marks what it hands back, NaN there),
        "auto" (ring + hand-back: what a solve runs).  Returns (ecc [B, 2], flag [B])."""
        ecc = torch.empty(B, 2, dtype=torch.float64, device=self.device)
        flag = torch.zeros(B, dtype=torch.int32, device=self.device)
        with torch.cuda.device(self.device):
            rc = self.lib.tolg_expected_change(self._h, {"statement": 0, "ring": 1, "auto": 2}[form], B, _ptr(ecc),
                                               _ptr(flag), self._stream())
        _capi.check(rc, "tolg_expected_change")
        return ecc, flag

    # ------------------------------------------------------------------------------------------
    def enable_timing(self, on=True):
        self.lib.tolg_enable_timing(self._h, int(on))

    def kernel_time(self, reset=True):
        """(ms in backward sweeps, ms in rollouts, ms in linearisation, number of backward launches)
        measured with HIP events on the launch stream."""
        a, b, c, n = C.c_double(), C.c_double(), C.c_double(), C.c_int64()
        self.lib.tolg_kernel_time(self._h, int(reset), C.byref(a), C.byref(b), C.byref(c), C.byref(n))
        return a.value, b.value, c.value, n.value
