/*
 * tolg.h -- C ABI of libtolg_hip.so: batched tracking-iLQR on SE(3) for MI355X (gfx950).
 *
 * This is the drop-in boundary for the hot path of
 * chenghuailin/trajectory_optimization_matrix_lie_groups.  Each entry point names the reference
 * interface it replaces (paths relative to the reference repository root).  The reference is pure
 * Python (no FFI of its own); the binding a maintainer adds is the ctypes stub shown in
 * INTEGRATION.md -- exactly what trajectory_optimization_matrix_lie_groups_amd/_capi.py does.
 *
 * Conventions
 *   - every pointer named d_* is DEVICE memory owned by the caller (PyTorch-ROCm tensors);
 *     the library never allocates, frees or synchronises; all work is enqueued on `stream`
 *     (a hipStream_t passed as void*).
 *   - poses are 4x4 row-major homogeneous matrices, twists are [omega, v]
 *     (traoptlibrary/traopt_utilis.py:43-92), fp64 throughout.
 *   - return value: 0 ok, <0 argument / launch error (TOLG_E_*); per-trajectory outcomes are
 *     written to d_status[B] (TOLG_ST_*).  No exception crosses this boundary.
 */
#ifndef TOLG_H
#define TOLG_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* TOLG_DYN_SO3: SO3Dynamics / SO3TrackingQuadraticGaussNewtonCost / iLQR_Tracking_SO3{,_MS}
 * (traoptlibrary/traopt_dynamics.py:275-418, traopt_cost.py:280-564, traopt_controller.py:526-1824)
 * carried in the SE(3) layout: poses are 4x4 with zero translation, twists [omega, 0], m = 6 with
 * u[3:6] = 0, J = blkdiag(J_so3, I3), R = blkdiag(R_so3, I3), Q/P with zero rows for the unused
 * coordinates.  The rotational sub-problem decouples exactly; Jacobian and cost follow the SO3
 * classes (no swapped-twist quirk, terminal l and l_x weighted with Q). */
/* TOLG_DYN_PENDULUM3D: Pendulum3dDyanmics (traoptlibrary/traopt_dynamics.py:421-626) under the SO3 cost and
 * controllers, same embedding as TOLG_DYN_SO3; the pivot acceleration u in R^3 is u[0:3].  Its F_u
 * depends on the state (J^-1 skew(m rho) R^T dt), the lower-left block of F_x on the input. */
enum { TOLG_DYN_SE3 = 0, TOLG_DYN_RIGIDBODY = 1, TOLG_DYN_DRONE = 2, TOLG_DYN_SO3 = 3, TOLG_DYN_PENDULUM3D = 4 };
enum { TOLG_MODE_MS = 0, TOLG_MODE_SS = 1 };
enum { TOLG_E_ARG = -1, TOLG_E_WORKSPACE = -2, TOLG_E_LAUNCH = -3, TOLG_E_SINGULAR = -4 };
/* TOLG_ST_INTERNAL: a wavefront of the fused rollout gave up waiting for its producer (never expected; it
 * replaces a GPU hang by a visible status). */
enum { TOLG_ST_OK = 0, TOLG_ST_MAXREG = 1, TOLG_ST_NODESCENT = 2, TOLG_ST_NONFINITE = 3, TOLG_ST_INTERNAL = 4 };
/* tolg_options.schedule: how one accept-always MS iteration (line_search = 0, rollout = 'nonlinear') is
 * issued.  AUTO: the rollout and the re-linearisation of the new trajectory share one launch (k_rollout_lin);
 * SPLIT: separate rollout and linearisation launches (the only form for every other mode).  Same results
 * either way up to the summation order inside lin_knot (identical code). */
enum { TOLG_SCHED_AUTO = 0, TOLG_SCHED_SPLIT = 1 };

/* Problem = one (dynamics, cost) pair shared by the whole batch.
 * Replaces the constructor arguments of SE3Dynamics / RigidBodyDynamics / DroneDynamics
 * (traoptlibrary/traopt_dynamics.py:633-690, :906-970, :1214-1278) and of
 * SE3TrackingQuadraticGaussNewtonCost (traoptlibrary/traopt_cost.py:587-622). */
typedef struct {
  int32_t kind;   /* TOLG_DYN_* */
  int32_t m;      /* action size: 6 (SE3, RigidBody) or 4 (Drone) */
  int32_t N;      /* horizon */
  int32_t reserved;
  double dt;
  double J[36];   /* inertia diag(I_b, mass*I3); any SPD 6x6 is accepted */
  double Q[144];  /* stage weights, 12x12 (only the two 6x6 diagonal blocks are read, as in
                     traopt_cost.py:697,702) */
  double P[144];  /* terminal weights */
  double R[36];   /* m x m row-major */
  double pend_mass;   /* Pendulum3dDyanmics m      (traopt_dynamics.py:425; other kinds: ignored) */
  double pend_length; /* Pendulum3dDyanmics length (traopt_dynamics.py:425) */
} tolg_problem;

/* Replaces the keyword arguments of iLQR_Tracking_SE3_MS.__init__/fit and
 * iLQR_Tracking_SE3.__init__/fit (traoptlibrary/traopt_controller.py:2359-2363, :2443-2445,
 * :1837-1838, :1880-1881). */
typedef struct {
  int32_t mode;            /* TOLG_MODE_MS | TOLG_MODE_SS */
  int32_t max_iter;        /* n_iterations */
  int32_t line_search;     /* MS: merit-function search (:2549-2590); SS always backtracks */
  int32_t rollout_linear;  /* rollout == 'linear' (the reference constructors' default, :1837-1838, :2359-2363): on the device the
                              linear rollout is an affine recursion in the deviation, linear in the step size -- one sweep serves
                              every candidate of a line search (DESIGN.md section 4 "Linear rollouts") */
  double tol_grad;         /* tol_grad_norm */
  double tol_defect;       /* tol_d_norm (MS) */
  double max_reg;          /* max_reg (1e10) */
  int32_t schedule;        /* TOLG_SCHED_* (no reference counterpart: launch structure only) */
  int32_t check_every;     /* tolg_solve_batch: 0 = enqueue all max_iter iterations and never synchronise; k > 0 = issue
                              them k at a time and stop once every trajectory has finished (tolg_solve_iterate_until) */
} tolg_options;

typedef struct tolg_handle_s* tolg_handle_t;

/* Bytes of device workspace tolg_create needs for batches up to max_batch and max_iter
 * iterations. */
size_t tolg_workspace_bytes(const tolg_problem* prob, int32_t max_batch);

/* Build a solver instance on caller-provided device workspace.  d_q_ref [(N+1)][16],
 * d_xi_ref [(N+1)][6] are read once (the q_ref "manifisation" of traopt_cost.py:614).
 * The handle itself is a small host object; destroy frees only that. */
int tolg_create(const tolg_problem* prob, const double* d_q_ref, const double* d_xi_ref,
                int32_t max_batch, void* d_workspace, size_t workspace_bytes, void* stream,
                tolg_handle_t* out);
void tolg_destroy(tolg_handle_t h);

/* fit for a batch of B independent trajectories -- replaces B calls of
 * iLQR_Tracking_SE3_MS.fit (traoptlibrary/traopt_controller.py:2443-2639) or
 * iLQR_Tracking_SE3.fit (:1880-2013), i.e. the joblib fan-out of
 * visualization/perturb_all_compute.py:240-250.
 *   in : d_x0_q [B][16], d_x0_xi [B][6], d_us_init [B][N][m]
 *   out: d_xs_q [B][N+1][16], d_xs_xi [B][N+1][6], d_us [B][N][m]
 *        d_J_hist [B][max_iter]        cost after iteration k (what on_iteration appends)
 *        d_grad_hist [B][max_iter+1]   gradient norm evaluated in iteration k
 *        d_defect_hist [B][max_iter+1] MS: [0] initial defect, [k+1] after iteration k
 *        d_alpha_hist [B][max_iter], d_mu_hist [B][max_iter]
 *        d_iters [B] callbacks made, d_status [B] TOLG_ST_*, d_converged [B]
 * History buffers may be NULL.  Entries past d_iters[b] are left untouched. */
int tolg_solve_batch(tolg_handle_t h, const tolg_options* opt, int32_t B, const double* d_x0_q,
                     const double* d_x0_xi, const double* d_us_init, double* d_xs_q, double* d_xs_xi,
                     double* d_us, double* d_J_hist, double* d_grad_hist, double* d_defect_hist,
                     double* d_alpha_hist, double* d_mu_hist, int32_t* d_iters, int32_t* d_status,
                     int32_t* d_converged, void* stream);

/* The same solve split in three, so a caller (bench.py, a receding-horizon loop) can issue the
 * iterations in slices with the batch resident in HBM: begin = _initial_guess + first
 * _linearization (traopt_controller.py:2486-2507); iterate = n_iter passes of the loop body
 * (:2522-2626); end = unpack to the reference's 4x4 layout.  tolg_solve_batch == begin +
 * iterate(max_iter) + end. */
int tolg_solve_begin(tolg_handle_t h, const tolg_options* opt, int32_t B, const double* d_x0_q,
                     const double* d_x0_xi, const double* d_us_init, double* d_J_hist, double* d_grad_hist,
                     double* d_defect_hist, double* d_alpha_hist, double* d_mu_hist, void* stream);
int tolg_solve_iterate(tolg_handle_t h, int32_t n_iter, void* stream);
int tolg_solve_end(tolg_handle_t h, double* d_xs_q, double* d_xs_xi, double* d_us, int32_t* d_iters,
                   int32_t* d_status, int32_t* d_converged, void* stream);
/* Up to n_iter iterations, issued check_every at a time; stops when no trajectory of the batch is iterating any more
 * -- the early exit of traopt_controller.py:2528-2532 (:1937-1942 single shooting) for the batch as a whole.  Unlike
 * every other entry point this one waits on the stream: after queueing slice s it waits for the count read back
 * behind slice s - 1, so the device never idles and a finished batch costs at most one more slice of launches
 * (finished trajectories are masked, their workgroups exit at once).  check_every = 0, or a solve nothing can end
 * (multiple shooting with tol_grad = 0 and no line search), is tolg_solve_iterate(n_iter).  *n_issued (may be NULL):
 * iterations queued. */
int tolg_solve_iterate_until(tolg_handle_t h, int32_t n_iter, int32_t check_every, int32_t* n_issued, void* stream);
/* Number of trajectories of the solve in flight that are still being iterated (not converged, not stopped by
 * a status), written to d_count[0] on `stream`.  Apart from tolg_solve_iterate_until the library never synchronises: with
 * check_every = 0 tolg_solve_batch enqueues max_iter iterations (finished trajectories are masked, finished workgroups
 * exit at once); a caller
 * that wants the early exit of traopt_controller.py:2528-2532 issues the iterations in slices and reads this
 * count between them, or calls tolg_solve_iterate_until, which pipelines exactly that. */
int tolg_solve_active_count(tolg_handle_t h, int32_t* d_count, void* stream);
/* Same export without ending the solve: what the per-iteration on_iteration callback of
 * traoptlibrary/traopt_controller.py:2621-2626 needs (current xs, us) when a caller wants it. */
int tolg_solve_peek(tolg_handle_t h, double* d_xs_q, double* d_xs_xi, double* d_us, int32_t* d_iters,
                    int32_t* d_status, int32_t* d_converged, void* stream);

/* Augmented-Lagrangian box input constraint lb <= u <= ub -- replaces ALConstrainedCost wrapping the
 * tracking cost with an InputConstraint (traoptlibrary/traopt_cost.py:1173-1320,
 * traoptlibrary/traopt_constraints.py:66-169).  d_lb/d_ub [m]; d_lambda, d_imu [B][N][2m] (multipliers
 * and the diagonal of I_mu for g = [lb - u; u - ub]; the terminal knot has g = 0).  The buffers stay
 * caller-owned and are read by every later solve on this handle; d_lb = NULL switches AL off. */
int tolg_set_al(tolg_handle_t h, const double* d_lb, const double* d_ub, const double* d_lambda,
                const double* d_imu);

/* One outer iteration of AL_iLQR_Tracking_SE3_MS (traoptlibrary/traopt_controller.py:3242-3250,
 * :3270-3290) for B independent problems on the controls d_us [B][N][m] of the inner solve:
 * d_maxviol[b] = max_k g_k over all knots; if it is below tol_constr the problem is marked in
 * d_al_converged[b] and left alone (now and in later calls); otherwise
 * lambda <- max(0, lambda + I_mu g), mu <- min(mu_scale*mu, mu_max),
 * I_mu <- (g < 0 and lambda_new == 0) ? 0 : mu_new. */
int tolg_al_update(tolg_handle_t h, int32_t B, const double* d_us, const double* d_lb, const double* d_ub,
                   double* d_lambda, double* d_imu, double* d_mu, double mu_scale, double mu_max,
                   double tol_constr, double* d_maxviol, int32_t* d_al_converged, void* stream);

/* One linearisation + backward pass on given trajectories (unit-parity entry point): replaces
 * iLQR_Tracking_SE3_MS._linearization + _backward_pass + _gradient_wrt_control
 * (traoptlibrary/traopt_controller.py:2823-3093; ms = 0: the SS variants :2098-2349).
 *   in : d_xs_q [B][N+1][16], d_xs_xi [B][N+1][6], d_us [B][N][m], mu/delta in d_mu_delta [B][2]
 *   out: d_Fx [B][N][12][12], d_d [B][N][12], d_lx [B][N+1][12], d_lxx11 [B][N+1][6][6],
 *        d_k [B][N][m], d_K [B][N][m][12], d_J [B], d_dnorm [B], d_grad [B],
 *        d_mu_delta updated.  Any output may be NULL. */
int tolg_linearize_backward(tolg_handle_t h, int32_t ms, double max_reg, int32_t B, const double* d_xs_q,
                            const double* d_xs_xi, const double* d_us, double* d_mu_delta, double* d_Fx,
                            double* d_d, double* d_lx, double* d_lxx11, double* d_k, double* d_K,
                            double* d_J, double* d_dnorm, double* d_grad, void* stream);

/* The reference's per-knot plugin methods for n states at knot i (i == N: terminal) -- replaces
 * dynamics.f / f_x / f_u (traoptlibrary/traopt_dynamics.py:789-850, :1403-1482) and cost.l / l_x / l_u /
 * l_xx / l_uu / _err (traoptlibrary/traopt_cost.py:659-867; with tolg_set_al active: ALConstrainedCost).
 *   in : d_x_q [n][16], d_x_xi [n][6], d_u [n][m] (ignored at the terminal knot)
 *   out: d_f_q [n][16], d_f_xi [n][6], d_Fx [n][12][12], d_Fu [n][12][m], d_l [n], d_lx [n][12],
 *        d_lxx [n][12][12], d_lu [n][m], d_luu [n][m][m], d_err [n][12] = [Log(x x_ref^-1); xi - xi_ref].
 * Any output may be NULL.  Uses the handle's workspace: not to be called during a solve in flight. */
int tolg_eval_knot(tolg_handle_t h, int32_t i, int32_t n, const double* d_x_q, const double* d_x_xi,
                   const double* d_u, double* d_f_q, double* d_f_xi, double* d_Fx, double* d_Fu, double* d_l,
                   double* d_lx, double* d_lxx, double* d_lu, double* d_luu, double* d_err, void* stream);

/* One closed-loop rollout with the gains left by the last tolg_linearize_backward on the same
 * trajectories -- replaces iLQR_Tracking_SE3_MS._rollout (:2641-2740) / iLQR_Tracking_SE3._rollout
 * (:2030-2082).  out: d_xs_q_new, d_xs_xi_new, d_us_new (same shapes as the inputs). */
int tolg_rollout(tolg_handle_t h, int32_t ms, int32_t rollout_linear, double alpha, int32_t B,
                 double* d_xs_q_new, double* d_xs_xi_new, double* d_us_new, void* stream);

/* Unit-parity entry point for the merit search's preparation: the linear alpha = 1 rollout and
 * _expected_cost_change (traopt_controller.py:2550-2552, :2730-2737, :2756-2769) on the trajectory, records and
 * gains tolg_linearize_backward(ms = 1) left in the workspace.  form 0: the statement-by-statement kernel
 * (k_expected_change); 1: the ring form alone (k_expected_change_ring; d_flag[b] = 1 marks the trajectories it
 * hands back, whose d_ecc entries are NaN); 2: the ring form with the hand-back behind it -- what a solve runs.
 * out: d_ecc [B][2] (first-order, second-order term), d_flag [B] (may be NULL). */
int tolg_expected_change(tolg_handle_t h, int32_t form, int32_t B, double* d_ecc, int32_t* d_flag, void* stream);

/* Timing hook for bench.py: HIP-event time (ms) and launch count of the dominant kernel
 * (backward sweep) accumulated since the last call with reset != 0.  Synchronises the recorded
 * events only.  Timing is not free: an event pair per launch lengthens an accept-always iteration
 * (two launches) by ~11 us of 600 on an MI355X -- leave it off where the rate matters (default). */
int tolg_kernel_time(tolg_handle_t h, int32_t reset, double* ms_backward, double* ms_rollout,
                     double* ms_linearize, int64_t* launches);
void tolg_enable_timing(tolg_handle_t h, int32_t on);

const char* tolg_version(void);

/* Diagnostic (no reference counterpart): the series forms of the Lie primitives (csrc/tolg_lie.h: se3_exp_fast,
 * se3_log_fast, so3_coef_fast, ljacinv_coef_fast, so3_exp_fast) evaluated one lane per argument set, under the
 * wave-shared gates their callers build -- so that arguments on both sides of every tier / domain boundary share a
 * wavefront -- for the parity test of those forms against a long-double reference (tests/test_gpu_series.py).
 *   in : d_args [n][8] = rotation vector w (3), translation part v (3), th2 of a second (step) rotation, mode
 *        (0: every function under its own gate, 1: one gate for all, built as lin_knot builds it)
 *   out: d_out [n][24] = so3_coef_fast a b c1 c2 c3 | ljacinv_coef_fast | se3_exp_fast q(4) t(3) |
 *        se3_log_fast(se3_exp(w, v)) w(3) v(3) | so3_exp_fast(w) q(4) | so3_coef_fast(th2_step).a
 * n must be a multiple of 64. */
int tolg_selftest_series(int32_t n, const double* d_args, double* d_out, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* TOLG_H */
