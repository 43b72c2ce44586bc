"""Generic plugin path of the SE(3) controllers: user-defined dynamics / cost plugins, per-knot callbacks.

The fused HIP path exists for the closed-form classes only (SURVEY.md §8b "What calls it").  A controller that
is handed anything else -- a subclass with overridden methods, a hand-written BaseDynamics / BaseCost -- runs
here: the same algorithm as the reference's iLQR_Tracking_SE3_MS / iLQR_Tracking_SE3 loops
(traopt_controller.py:2443-2639 with `_linearization` :2823-2910, `_backward_pass` / `_Q` :2912-3068,
`_gradient_wrt_control` :3070-3093, `_rollout` :2641-2740; single shooting :1880-2013, :2030-2082, :2323-2349)
driven through the plugin methods f, f_x, f_u, l, l_x, l_u, l_xx, l_ux, l_uu, one knot at a time, on the host.
It is the slow, fully general fallback, not a second implementation of the hot path: the arithmetic is the
homogeneous-coordinate sweep of the Euclidean iLQR class (one congruence per knot with the defect in the last
column of the transition), the group operations go through the manif interface of traopt_utilis.
"""
import warnings

import numpy as np

from .traopt_utilis import SE32manifSE3, is_pos_def, manifSE32SE3, manifse32se3, se32manifse3

_MSG_MAXREG = "exceeded max regularization term"  # traopt_controller.py:2984
_MSG_NODESCENT = "Couldn't find descent direction, regularization and line search step exhausted"  # :2632


def _log(T):
    """Log of a 4x4 pose as a twist in the library's [omega, v] order."""
    return manifse32se3(SE32manifSE3(T).log())


def _exp(tau):
    return manifSE32SE3(se32manifse3(tau).exp())


def _inv(T):
    return manifSE32SE3(SE32manifSE3(T).inverse())


def _project(T):
    """Every pose the reference touches passes through a unit quaternion (traopt_utilis.py:331-354)."""
    return manifSE32SE3(SE32manifSE3(T))


def _dev(xa, xb):
    """State deviation [Log(qa^-1 qb); xi_b - xi_a] (traopt_controller.py:2680-2687)."""
    return np.r_[_log(_inv(xa[0]) @ xb[0]), np.asarray(xb[1], float) - np.asarray(xa[1], float)]


class GenericLieILQR:
    """mode 'ms' / 'ss'; the public controllers keep the reference's signatures and delegate here."""

    def __init__(self, dynamics, cost, N, mode, max_reg=1e10, line_search=False, rollout="nonlinear"):
        self.dynamics, self.cost, self.N, self.mode = dynamics, cost, int(N), mode
        self.n, self.m = 12, int(dynamics.action_size)
        self.max_reg, self.linear = max_reg, rollout == "linear"
        self.merit = mode == "ms" and bool(line_search)   # merit-function search (traopt_controller.py:2549-2590)
        self.mu, self.delta = 1.0, 2.0

    # ---- one pass over the plugins ---------------------------------------------------------------
    def _expand(self, xs, us):
        n, m, N, ms = self.n, self.m, self.N, self.mode == "ms"
        G = np.zeros((N, n + 1, n + m + 1)); H = np.zeros((N, n + m + 1, n + m + 1)); F = []
        J = 0.0
        for i in range(N):
            x, u = xs[i], us[i]
            fq, fxi = self.dynamics.f(x, u, i)
            F.append((np.asarray(fq, float), np.asarray(fxi, float)))
            G[i, :n, :n] = self.dynamics.f_x(x, u, i)
            G[i, :n, n:n + m] = self.dynamics.f_u(x, u, i)
            if ms:  # defect d = [Log(x_{i+1}^-1 f_q); f_xi - xi_{i+1}] (:2882-2888)
                G[i, :n, n + m] = np.r_[_log(_inv(xs[i + 1][0]) @ F[i][0]), F[i][1] - xs[i + 1][1]]
            G[i, n, n + m] = 1.0
            H[i, :n, :n] = self.cost.l_xx(x, u, i)
            H[i, n:n + m, :n] = self.cost.l_ux(x, u, i)
            H[i, :n, n:n + m] = H[i, n:n + m, :n].T
            H[i, n:n + m, n:n + m] = self.cost.l_uu(x, u, i)
            g = np.r_[self.cost.l_x(x, u, i), self.cost.l_u(x, u, i)]
            H[i, :n + m, n + m] = H[i, n + m, :n + m] = g
            J += float(self.cost.l(x, u, i))
        xN = xs[N]
        HN = np.zeros((n + 1, n + 1))
        HN[:n, :n] = self.cost.l_xx(xN, None, N, terminal=True)
        HN[:n, n] = HN[n, :n] = self.cost.l_x(xN, None, N, terminal=True)
        J += float(self.cost.l(xN, None, N, terminal=True))
        dnorm = float(np.linalg.norm(G[:, :n, n + m])) if ms else 0.0
        return dict(G=G, H=H, HN=HN, F=F, J=J, dnorm=dnorm)

    def _sweep(self, e):
        """Gains and the gradient norm; mu / delta persist across knots and iterations (:2977-2991)."""
        n, m, N, ms = self.n, self.m, self.N, self.mode == "ms"
        V = e["HN"].copy()
        p = e["HN"][:n, n].copy()
        k = np.zeros((N, m)); K = np.zeros((N, m, n))
        gsum, warned = 0.0, False
        for i in range(N - 1, -1, -1):
            G, Hi = e["G"][i], e["H"][i]
            fu = G[:n, n:n + m]
            Q0 = Hi + G.T @ V @ G
            reg = fu.T @ G[:n, :n + m]
            while True:
                Q = Q0.copy()
                Q[n:n + m, :n + m] += self.mu * reg
                Q[:n, n:n + m] = Q[n:n + m, :n].T
                Quu = Q[n:n + m, n:n + m]
                if not is_pos_def(Quu + Quu.T):
                    self.delta = max(1.0, self.delta) * 2.0
                    self.mu = max(1e-6, self.mu * self.delta)
                    if self.max_reg and self.mu >= self.max_reg:
                        warnings.warn(_MSG_MAXREG)
                        warned = True
                        break
                    continue
                self.delta = min(1.0, self.delta) / 2.0
                self.mu *= self.delta
                if self.mu <= 1e-6:
                    self.mu = 0.0
                break
            Qua = np.concatenate([Q[n:n + m, :n], Q[n:n + m, n + m:]], axis=1)  # [Q_ux | Q_u]
            if ms:
                gsum += float(np.linalg.norm(Qua[:, n]))                        # l_u + F_u^T (V_x + V_xx d) (:3090)
            else:
                gsum += float(np.linalg.norm(Hi[n:n + m, n + m] + fu.T @ p))     # adjoint recursion (:2323-2349)
                p = Hi[:n, n + m] + G[:n, :n].T @ p
            Kk = -np.linalg.solve(Quu, Qua)
            K[i], k[i] = Kk[:, :n], Kk[:, n]
            Qaa = np.empty((n + 1, n + 1))
            Qaa[:n, :n] = Q[:n, :n]
            Qaa[:n, n] = Qaa[n, :n] = Q[:n, n + m]
            Qaa[n, n] = Q[n + m, n + m]
            cross = Kk.T @ Qua
            V = Qaa + Kk.T @ Quu @ Kk + cross + cross.T
            V[:n, :n] = 0.5 * (V[:n, :n] + V[:n, :n].T)
        return k, K, gsum / N, warned

    def _rollout(self, xs, us, k, K, e, alpha, linear=None, errs=None):
        """`errs` (a dict) receives the deviations dx [N+1, n], du [N, m] of the new path from the nominal one."""
        n, m, N, ms = self.n, self.m, self.N, self.mode == "ms"
        linear = self.linear if linear is None else linear
        xs_new = [[np.array(xs[0][0], float), np.array(xs[0][1], float)]]
        us_new = np.zeros_like(us)
        dxs, dus = np.zeros((N + 1, n)), np.zeros((N, m))
        for i in range(N):
            dx = _dev(xs[i], xs_new[i])
            du = alpha * k[i] + K[i] @ dx
            dxs[i], dus[i] = dx, du
            us_new[i] = us[i] + du
            G = e["G"][i]
            d = G[:n, n + m]
            if linear:        # :2720-2726: x_{i+1} (+) (F_x dx + F_u du + alpha d)
                lin = G[:n, :n] @ dx + G[:n, n:n + m] @ du + alpha * d
                base = xs[i + 1]
                q = _project(np.asarray(base[0], float) @ _exp(lin[:6]))
                xi = np.asarray(base[1], float) + lin[6:]
            else:
                fq, fxi = self.dynamics.f(xs_new[i], us_new[i], i)
                if ms:        # :2713-2718
                    q = _project(np.asarray(xs[i + 1][0], float) @ _exp(alpha * d[:6]) @ _inv(e["F"][i][0]) @ np.asarray(fq, float))
                    xi = np.asarray(xs[i + 1][1], float) + np.asarray(fxi, float) - e["F"][i][1] + alpha * d[6:]
                else:         # :2073-2080
                    q, xi = np.asarray(fq, float), np.asarray(fxi, float)
            xs_new.append([q, xi])
        if errs is not None:
            dxs[N] = _dev(xs[N], xs_new[N])
            errs["dx"], errs["du"] = dxs, dus
        return xs_new, us_new

    def _expected_cost_change(self, e, dxs, dus):
        """(first order, second order) of the quadratic cost model along the deviations (:2756-2769)."""
        n, m, N = self.n, self.m, self.N
        c1 = c2 = 0.0
        for i in range(N):
            z = np.r_[dxs[i], dus[i]]
            Hi = e["H"][i]
            c1 += float(Hi[:n + m, n + m] @ z)
            c2 += float(z @ Hi[:n + m, :n + m] @ z)
        c1 += float(e["HN"][:n, n] @ dxs[N])
        c2 += float(dxs[N] @ e["HN"][:n, :n] @ dxs[N])
        return c1, c2

    def _cost_and_defect(self, xs, us):
        J = sum(float(self.cost.l(xs[i], us[i], i)) for i in range(self.N))
        J += float(self.cost.l(xs[self.N], None, self.N, terminal=True))
        d2 = 0.0
        if self.mode == "ms":
            for i in range(self.N):
                fq, fxi = self.dynamics.f(xs[i], us[i], i)
                r = np.r_[_log(_inv(xs[i + 1][0]) @ np.asarray(fq, float)), np.asarray(fxi, float) - xs[i + 1][1]]
                d2 += float(r @ r)
        return J, float(np.sqrt(d2))

    # ---- fit ---------------------------------------------------------------------------------------
    def fit(self, x0, us_init, q_ref, xi_ref, n_iterations, tol_grad_norm, tol_d_norm, on_iteration,
            append_grad_on_convergence=False):
        N, ms = self.N, self.mode == "ms"
        self.mu, self.delta = 1.0, 2.0
        us = np.array(us_init, dtype=float).reshape(N, self.m)
        x0 = [np.asarray(x0[0], float), np.asarray(x0[1], float)]
        if ms:   # _initial_guess (:3123-3136)
            xs = [x0] + [[np.asarray(q_ref[i], float), np.asarray(xi_ref[i], float)] for i in range(1, N + 1)]
        else:    # _init_rollout (:2015-2028)
            xs = [x0]
            for i in range(N):
                fq, fxi = self.dynamics.f(xs[i], us[i], i)
                xs.append([np.asarray(fq, float), np.asarray(fxi, float)])
        J_hist, xs_hist, us_hist, grad_hist, defect_hist = [], [list(xs)], [us.copy()], [], []
        merit = self.merit
        alphas = (list(1.1 ** (-np.arange(20) ** 2)) if merit else [1.0]) if ms else list(1.1 ** (-np.arange(13) ** 2))
        d_weight_prev = 10.0   # _defect_mu0 (:2406-2410: rho 0.5, gamma 0.05, mu_min = mu0, kappa 1e-12)
        e = self._expand(xs, us)
        if ms:
            defect_hist.append(e["dnorm"])
        for it in range(int(n_iterations)):
            k, K, grad, _ = self._sweep(e)
            if not ms:
                grad_hist.append(grad)
            if grad < tol_grad_norm and (not ms or e["dnorm"] < tol_d_norm):
                if ms and append_grad_on_convergence:
                    grad_hist.append(grad)
                break
            accepted, J_opt, dn, alpha = False, e["J"], e["dnorm"], 1.0
            if merit:
                # expected change of the quadratic model along the full linear step, defect weight, merit (:2549-2558)
                errs = {}
                self._rollout(xs, us, k, K, e, 1.0, linear=True, errs=errs)
                c1, c2 = self._expected_cost_change(e, errs["dx"], errs["du"])
                d_norm = e["dnorm"]
                d_weight = d_weight_prev if d_norm < 1e-12 else max(10.0, 10.0 + abs(c1 + 0.5 * c2) / ((1 - 0.5) * d_norm))
                d_weight_prev = d_weight
                merit0 = J_opt + d_weight * d_norm
            for alpha in alphas:
                xs_try, us_try = self._rollout(xs, us, k, K, e, alpha)
                J_try, d_try = self._cost_and_defect(xs_try, us_try)
                if merit:                    # Armijo test on the merit function (:2560-2590)
                    dn = d_try               # the callback sees the last candidate's defect norm, accepted or not
                    J_exp = alpha * c1 + 0.5 * alpha ** 2 * c2
                    if (J_try + d_weight * d_try) - merit0 < 0.05 * (J_exp - alpha * d_weight * d_norm):
                        xs, us, J_opt, accepted = xs_try, us_try, J_try, True
                        break
                elif ms or J_try < J_opt:    # MS without line search accepts every step (:2593-2612)
                    xs, us, J_opt, dn, accepted = xs_try, us_try, J_try, d_try, True
                    break
            if accepted:
                e = self._expand(xs, us)
            if on_iteration:
                if ms:
                    on_iteration(it, xs, us, J_opt, accepted, False, dn, grad, alpha, self.mu, J_hist, xs_hist, us_hist,
                                 grad_hist, defect_hist)
                else:
                    on_iteration(it, xs, us, J_opt, accepted, False, grad, alpha, self.mu, J_hist, xs_hist, us_hist)
            if not accepted:
                warnings.warn(_MSG_NODESCENT)
                break
            if not np.isfinite(J_opt):
                break
        return xs, us, J_hist, xs_hist, us_hist, grad_hist, defect_hist
