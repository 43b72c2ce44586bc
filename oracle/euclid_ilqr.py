"""TEST INFRASTRUCTURE (oracle) -- CPU restatement of the reference's Euclidean iLQR / DDP
(traoptlibrary/traopt_controller.py:42-520) for BASELINE.json config 1 (main_ddp.py, cart-pole swing-up).

Only tests/ may import this.  Parity unpinned: the reference holds no recorded output of main_ddp.py and its
AutoDiff classes need jax, which is absent; this file restates the algorithm line by line in NumPy with
derivatives obtained independently of the product (complex-step first derivatives, central differences of
those for the DDP tensors, closed-form quadratic cost), so that the torch.func plumbing is checked against
something that shares no code with it.
"""
import numpy as np

MC, MP, LEN, G = 1.0, 1.0, 1.0, 9.8  # main_ddp.py:41-44
R_U = 200.0                          # main_ddp.py:74
Q_X = np.diag([100.0, 100.0, 10000.0, 100.0])  # main_ddp.py:77, :83


def cartpole_f(x, u):
    """main_ddp.py:38-55 (works on complex arrays for the complex-step derivative)."""
    x1, x2, x3, x4 = x
    u = u[0]
    s, c = np.sin(x3), np.cos(x3)
    dx2 = 1 / (MC + MP * s ** 2) * (u + MP * s * (LEN * x4 ** 2 + G * c))
    dx4 = 1 / (LEN * MC + LEN * MP * s ** 2) * (-u * c - MP * LEN * x4 ** 2 * c * s - (MC + MP) * G * s)
    return np.array([x2, dx2, x4, dx4])


def fd_rk4(x, u, dt):
    """main_ddp.py:59-67"""
    s1 = cartpole_f(x, u)
    s2 = cartpole_f(x + dt / 2 * s1, u)
    s3 = cartpole_f(x + dt / 2 * s2, u)
    s4 = cartpole_f(x + dt * s3, u)
    return x + dt / 6 * (s1 + 2 * s2 + 2 * s3 + s4)


def jac(x, u, dt, h=1e-30):
    n, m = x.size, u.size
    Fx = np.empty((n, n)); Fu = np.empty((n, m))
    for k in range(n):
        xc = x.astype(complex); xc[k] += 1j * h
        Fx[:, k] = fd_rk4(xc, u.astype(complex), dt).imag / h
    for k in range(m):
        uc = u.astype(complex); uc[k] += 1j * h
        Fu[:, k] = fd_rk4(x.astype(complex), uc, dt).imag / h
    return Fx, Fu


def hess(x, u, dt, h=1e-5):
    """F_xx[a,b,c] = d2 f_a / dx_b dx_c, F_ux[a,b,c] = d2 f_a / du_b dx_c, F_uu (traopt_dynamics.py:166-168)."""
    n, m = x.size, u.size
    Fxx = np.empty((n, n, n)); Fux = np.empty((n, m, n)); Fuu = np.empty((n, m, m))
    for c in range(n):
        d = np.zeros(n); d[c] = h
        ap, bp = jac(x + d, u, dt); am, bm = jac(x - d, u, dt)
        Fxx[:, :, c] = (ap - am) / (2 * h)
        Fux[:, :, c] = (bp - bm) / (2 * h)
    for c in range(m):
        d = np.zeros(m); d[c] = h
        _, bp = jac(x, u + d, dt); _, bm = jac(x, u - d, dt)
        Fuu[:, :, c] = (bp - bm) / (2 * h)
    return Fxx, Fux, Fuu


def fit(x0, x_goal, us_init, dt, n_iterations=100, tol_J=1e-6, tol_grad_norm=1e-3, hessians=False, max_reg=1e10):
    """iLQR.fit (traopt_controller.py:84-221) with the cart-pole plugins of main_ddp.py inlined."""
    N, m = us_init.shape
    n = x0.size
    mu, mu_min, delta_0 = 1.0, 1e-6, 2.0
    delta = delta_0
    alphas = 1.1 ** (-np.arange(10) ** 2)
    us = us_init.copy()

    def cost(xs, us):
        dx = xs - x_goal
        return 0.5 * R_U * np.sum(us[:, 0] ** 2) + 0.5 * np.einsum("ia,ab,ib->", dx[:-1], Q_X, dx[:-1]) \
            + 0.5 * dx[-1] @ Q_X @ dx[-1]

    hist = dict(J=[], grad=[], alpha=[], mu=[], accepted=[])
    changed, converged = True, False
    for it in range(n_iterations):
        accepted = False
        if changed:
            xs = np.empty((N + 1, n)); xs[0] = x0
            for i in range(N):
                xs[i + 1] = fd_rk4(xs[i], us[i], dt)
            F = [jac(xs[i], us[i], dt) for i in range(N)]
            H = [hess(xs[i], us[i], dt) for i in range(N)] if hessians else None
            L_x = (xs - x_goal) @ Q_X
            L_u = R_U * us
            J_opt = cost(xs, us)
            changed = False
        # _backward_pass (:341-417) with _Q (:419-492)
        V_x, V_xx = L_x[-1], Q_X
        k = np.empty((N, m)); K = np.empty((N, m, n))
        for i in range(N - 1, -1, -1):
            fx, fu = F[i]
            Q_x = L_x[i] + fx.T @ V_x
            Q_u = L_u[i] + fu.T @ V_x
            Q_xx = Q_X + fx.T @ V_xx @ fx
            reg = mu * np.eye(n)
            Q_ux = fu.T @ (V_xx + reg) @ fx
            Q_uu = R_U * np.eye(m) + fu.T @ (V_xx + reg) @ fu
            if hessians:
                fxx, fux, fuu = H[i]
                Q_xx = Q_xx + np.tensordot(V_x, fxx, axes=1)
                Q_ux = Q_ux + np.tensordot(V_x, fux, axes=1)
                Q_uu = Q_uu + np.tensordot(V_x, fuu, axes=1)
            k[i] = -np.linalg.solve(Q_uu, Q_u)
            K[i] = -np.linalg.solve(Q_uu, Q_ux)
            V_x = Q_x + K[i].T @ Q_uu @ k[i] + K[i].T @ Q_u + Q_ux.T @ k[i]
            V_xx = Q_xx + K[i].T @ Q_uu @ K[i] + K[i].T @ Q_ux + Q_ux.T @ K[i]
            V_xx = 0.5 * (V_xx + V_xx.T)
        # _gradient_wrt_control (:494-520)
        p = L_x[N]; gsum = 0.0
        for t in range(N - 1, -1, -1):
            fx, fu = F[t]
            gsum += np.linalg.norm(L_u[t] + fu.T @ p)
            p = L_x[t] + fx.T @ p
        grad = gsum / N
        for alpha in alphas:
            xs_new = np.empty_like(xs); us_new = np.empty_like(us); xs_new[0] = xs[0]
            for i in range(N):
                us_new[i] = us[i] + alpha * k[i] + K[i] @ (xs_new[i] - xs[i])
                xs_new[i + 1] = fd_rk4(xs_new[i], us_new[i], dt)
            J_new = cost(xs_new, us_new)
            if grad < tol_grad_norm:
                converged = True; accepted = True
                break
            if J_new < J_opt:
                if abs((J_opt - J_new) / J_opt) < tol_J:
                    converged = True
                J_opt, xs, us, changed = J_new, xs_new, us_new, True
                delta = min(1.0, delta) / delta_0
                mu *= delta
                if mu <= mu_min:
                    mu = 0.0
                accepted = True
                break
        if not accepted:
            delta = max(1.0, delta) * delta_0
            mu = max(mu_min, mu * delta)
            if max_reg and mu >= max_reg:
                break
        hist["J"].append(J_opt); hist["grad"].append(grad); hist["alpha"].append(alpha); hist["mu"].append(mu)
        hist["accepted"].append(accepted)
        if converged:
            break
    return xs, us, hist
