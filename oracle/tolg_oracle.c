/*
 * tolg_oracle.c -- CPU restatement (the ORACLE) of the reference's SE3 / RigidBody / Drone
 * tracking-iLQR hot path.  TEST INFRASTRUCTURE ONLY: nothing in the product package may link,
 * import or call this file.  Allowed users: tests/, __graft_entry__.smoke(), bench.py's
 * cpu_baseline leg.
 *
 * Parity status: PINNED for DroneDynamics MS-iLQR (29 iterations) and SS-iLQR (9 iterations incl.
 * the 13-alpha failed line search) against the reference's own per-iteration prints stored in
 * /root/reference/baseline_applications.ipynb (tests/golden/drone_n150_log.json; every printed
 * J / gradient reproduced, see tests/test_oracle_golden.py).  SE3Dynamics shares every code path
 * with DroneDynamics except gravity / Pu (flags below) and has no recorded run of its own that a
 * sanctioned loader can open (the *.pkl results are refused by numpy.load/torch.load, see
 * DESIGN.md); branches never exercised by a golden (non-PD regularisation loop,
 * line_search=True, rollout='linear', AL terms) are "parity unpinned".
 *
 * Every function cites the reference file:line it restates (paths relative to /root/reference).
 * manifpy (C++ `manif`, not vendored, version unpinned by the reference) is restated from its
 * published formulas (Sola, Deray, Atchuthan, "A micro Lie theory for state estimation in
 * robotics"; Barfoot, "State Estimation for Robotics" eq. 7.86) and checked formula-free against
 * scipy expm/logm in tests/test_oracle_lie.py.
 *
 * Conventions: twist order [omega, v] (traoptlibrary/traopt_utilis.py:43-92); 4x4 row-major;
 * quaternion (x, y, z, w) as scipy/manif store it.
 */
#include <math.h>
#include <omp.h>
#include <stdlib.h>
#include <string.h>

#define MANIF_EPS 1e-10 /* manif Constants<double>::eps */

/* ------------------------------------------------------------------------------------------ */
/* small dense helpers                                                                         */
/* ------------------------------------------------------------------------------------------ */
static void skew3(const double w[3], double S[9]) { /* traopt_utilis.py:13-24 */
  S[0] = 0;     S[1] = -w[2]; S[2] = w[1];
  S[3] = w[2];  S[4] = 0;     S[5] = -w[0];
  S[6] = -w[1]; S[7] = w[0];  S[8] = 0;
}
static void mat3_mul(const double A[9], const double B[9], double C[9]) {
  double T[9];
  for (int i = 0; i < 3; i++)
    for (int j = 0; j < 3; j++) {
      double s = 0;
      for (int k = 0; k < 3; k++) s += A[3 * i + k] * B[3 * k + j];
      T[3 * i + j] = s;
    }
  memcpy(C, T, sizeof T);
}
static void mat3_vec(const double A[9], const double v[3], double o[3]) {
  double t[3];
  for (int i = 0; i < 3; i++) t[i] = A[3 * i] * v[0] + A[3 * i + 1] * v[1] + A[3 * i + 2] * v[2];
  o[0] = t[0]; o[1] = t[1]; o[2] = t[2];
}
static void mat3_T(const double A[9], double B[9]) {
  double T[9];
  for (int i = 0; i < 3; i++)
    for (int j = 0; j < 3; j++) T[3 * i + j] = A[3 * j + i];
  memcpy(B, T, sizeof T);
}
/* C (n x p) = A (n x k) * B (k x p), row-major, no aliasing */
static void mm(int n, int k, int p, const double *A, const double *B, double *C) {
  for (int i = 0; i < n; i++)
    for (int j = 0; j < p; j++) {
      double s = 0;
      for (int l = 0; l < k; l++) s += A[i * k + l] * B[l * p + j];
      C[i * p + j] = s;
    }
}
/* C (k x p) = A^T (A is n x k) * B (n x p) */
static void mtm(int n, int k, int p, const double *A, const double *B, double *C) {
  for (int i = 0; i < k; i++)
    for (int j = 0; j < p; j++) {
      double s = 0;
      for (int l = 0; l < n; l++) s += A[l * k + i] * B[l * p + j];
      C[i * p + j] = s;
    }
}

/* ------------------------------------------------------------------------------------------ */
/* SO(3) / SE(3) primitives (manif semantics)                                                   */
/* ------------------------------------------------------------------------------------------ */
typedef struct { double q[4]; double t[3]; } se3_t; /* unit quaternion xyzw + translation */

/* scipy.spatial.transform.Rotation.from_matrix(...).as_quat()  (traopt_utilis.py:167-181, :331-342) */
static void quat_from_rotm(const double R[9], double q[4]) {
  double dec[4] = {R[0], R[4], R[8], R[0] + R[4] + R[8]};
  int c = 0;
  for (int i = 1; i < 4; i++) if (dec[i] > dec[c]) c = i;
  if (c != 3) {
    int i = c, j = (i + 1) % 3, k = (j + 1) % 3;
    q[i] = 1 - dec[3] + 2 * R[3 * i + i];
    q[j] = R[3 * j + i] + R[3 * i + j];
    q[k] = R[3 * k + i] + R[3 * i + k];
    q[3] = R[3 * k + j] - R[3 * j + k];
  } else {
    q[0] = R[7] - R[5];
    q[1] = R[2] - R[6];
    q[2] = R[3] - R[1];
    q[3] = 1 + dec[3];
  }
  double n = sqrt(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
  for (int i = 0; i < 4; i++) q[i] /= n;
}
/* Eigen::Quaternion::toRotationMatrix (manif SE3::transform / rotation) */
static void rotm_from_quat(const double q[4], double R[9]) {
  double x = q[0], y = q[1], z = q[2], w = q[3];
  double tx = 2 * x, ty = 2 * y, tz = 2 * z;
  double twx = tx * w, twy = ty * w, twz = tz * w;
  double txx = tx * x, txy = ty * x, txz = tz * x;
  double tyy = ty * y, tyz = tz * y, tzz = tz * z;
  R[0] = 1 - (tyy + tzz); R[1] = txy - twz;       R[2] = txz + twy;
  R[3] = txy + twz;       R[4] = 1 - (txx + tzz); R[5] = tyz - twx;
  R[6] = txz - twy;       R[7] = tyz + twx;       R[8] = 1 - (txx + tyy);
}
static void quat_mul(const double a[4], const double b[4], double o[4]) {
  double ax = a[0], ay = a[1], az = a[2], aw = a[3], bx = b[0], by = b[1], bz = b[2], bw = b[3];
  o[0] = aw * bx + ax * bw + ay * bz - az * by;
  o[1] = aw * by + ay * bw + az * bx - ax * bz;
  o[2] = aw * bz + az * bw + ax * by - ay * bx;
  o[3] = aw * bw - ax * bx - ay * by - az * bz;
}
/* SE32manifSE3 (traopt_utilis.py:331-342): matrix -> scipy quaternion -> manif SE3 */
static void se3_from_matrix(const double M[16], se3_t *X) {
  double R[9] = {M[0], M[1], M[2], M[4], M[5], M[6], M[8], M[9], M[10]};
  quat_from_rotm(R, X->q);
  X->t[0] = M[3]; X->t[1] = M[7]; X->t[2] = M[11];
}
/* manifSE32SE3 (traopt_utilis.py:344-354): SE3.transform() */
static void se3_to_matrix(const se3_t *X, double M[16]) {
  double R[9];
  rotm_from_quat(X->q, R);
  M[0] = R[0]; M[1] = R[1]; M[2] = R[2];  M[3] = X->t[0];
  M[4] = R[3]; M[5] = R[4]; M[6] = R[5];  M[7] = X->t[1];
  M[8] = R[6]; M[9] = R[7]; M[10] = R[8]; M[11] = X->t[2];
  M[12] = 0; M[13] = 0; M[14] = 0; M[15] = 1;
}
/* manif SO3Tangent::exp */
static void so3_exp(const double w[3], double q[4]) {
  double th2 = w[0] * w[0] + w[1] * w[1] + w[2] * w[2];
  if (th2 > MANIF_EPS) {
    double th = sqrt(th2), s = sin(0.5 * th) / th;
    q[0] = s * w[0]; q[1] = s * w[1]; q[2] = s * w[2]; q[3] = cos(0.5 * th);
  } else {
    q[0] = 0.5 * w[0]; q[1] = 0.5 * w[1]; q[2] = 0.5 * w[2]; q[3] = 1.0;
    double n = sqrt(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + 1.0);
    for (int i = 0; i < 4; i++) q[i] /= n;
  }
}
/* manif SO3::log */
static void so3_log(const double q[4], double w[3]) {
  double s2 = q[0] * q[0] + q[1] * q[1] + q[2] * q[2], c;
  if (s2 > MANIF_EPS) {
    double s = sqrt(s2), cw = q[3];
    double two = 2.0 * ((cw < 0.0) ? atan2(-s, -cw) : atan2(s, cw));
    c = two / s;
  } else {
    c = 2.0;
  }
  w[0] = c * q[0]; w[1] = c * q[1]; w[2] = c * q[2];
}
/* manif SO3Tangent::ljac = V(w); rjac = ljac(-w) */
static void so3_ljac(const double w[3], double J[9]) {
  double W[9], W2[9];
  skew3(w, W);
  double th2 = w[0] * w[0] + w[1] * w[1] + w[2] * w[2];
  memset(J, 0, 9 * sizeof(double));
  J[0] = J[4] = J[8] = 1.0;
  if (th2 <= MANIF_EPS) {
    for (int i = 0; i < 9; i++) J[i] += 0.5 * W[i];
    return;
  }
  double th = sqrt(th2), a = (1 - cos(th)) / th2, b = (th - sin(th)) / (th2 * th);
  mat3_mul(W, W, W2);
  for (int i = 0; i < 9; i++) J[i] += a * W[i] + b * W2[i];
}
/* manif SO3Tangent::ljacinv; rjacinv = ljacinv(-w) */
static void so3_ljacinv(const double w[3], double J[9]) {
  double W[9], W2[9];
  skew3(w, W);
  double th2 = w[0] * w[0] + w[1] * w[1] + w[2] * w[2];
  memset(J, 0, 9 * sizeof(double));
  J[0] = J[4] = J[8] = 1.0;
  if (th2 <= MANIF_EPS) {
    for (int i = 0; i < 9; i++) J[i] -= 0.5 * W[i];
    return;
  }
  double th = sqrt(th2);
  double c = 1.0 / th2 - (1 + cos(th)) / (2 * th * sin(th));
  mat3_mul(W, W, W2);
  for (int i = 0; i < 9; i++) J[i] += -0.5 * W[i] + c * W2[i];
}
/* manif SE3Tangent::exp: (V(w) v, Exp(w)) */
static void se3_exp(const double tau[6], se3_t *X) {
  double V[9];
  so3_exp(tau, X->q);
  so3_ljac(tau, V);
  mat3_vec(V, tau + 3, X->t);
}
/* manif SE3::log: (w = Log(R), V(w)^-1 t) */
static void se3_log(const se3_t *X, double tau[6]) {
  double Vi[9];
  so3_log(X->q, tau);
  so3_ljacinv(tau, Vi);
  mat3_vec(Vi, X->t, tau + 3);
}
static void se3_compose(const se3_t *A, const se3_t *B, se3_t *C) {
  double R[9], t[3], q[4];
  rotm_from_quat(A->q, R);
  mat3_vec(R, B->t, t);
  quat_mul(A->q, B->q, q);
  for (int i = 0; i < 3; i++) C->t[i] = A->t[i] + t[i];
  memcpy(C->q, q, sizeof q);
}
static void se3_inverse(const se3_t *A, se3_t *B) {
  double R[9], Rt[9], t[3];
  rotm_from_quat(A->q, R);
  mat3_T(R, Rt);
  mat3_vec(Rt, A->t, t);
  B->q[0] = -A->q[0]; B->q[1] = -A->q[1]; B->q[2] = -A->q[2]; B->q[3] = A->q[3];
  B->t[0] = -t[0]; B->t[1] = -t[1]; B->t[2] = -t[2];
}
/* Ad(X) in [w,v] order = Jmnf2J(manif adj): [[R,0],[[t]x R, R]] (traopt_utilis.py:387-399) */
static void se3_adj(const se3_t *X, double Ad[36]) {
  double R[9], T[9], TR[9];
  rotm_from_quat(X->q, R);
  skew3(X->t, T);
  mat3_mul(T, R, TR);
  memset(Ad, 0, 36 * sizeof(double));
  for (int i = 0; i < 3; i++)
    for (int j = 0; j < 3; j++) {
      Ad[6 * i + j] = R[3 * i + j];
      Ad[6 * (i + 3) + j] = TR[3 * i + j];
      Ad[6 * (i + 3) + j + 3] = R[3 * i + j];
    }
}
/* Barfoot eq. 7.86 Q(rho, theta) (what manif SE3Tangent::fillQ evaluates) */
static void se3_Q(const double rho[3], const double th[3], double Q[9]) {
  double P[9], W[9];
  skew3(rho, P);
  skew3(th, W);
  double t2 = th[0] * th[0] + th[1] * th[1] + th[2] * th[2];
  double c1, c2, c3;
  if (t2 <= MANIF_EPS) {
    c1 = 1.0 / 6 - t2 / 120; c2 = 1.0 / 24 - t2 / 720; c3 = 1.0 / 120 - t2 / 2520;
  } else {
    double t = sqrt(t2), s = sin(t), c = cos(t);
    c1 = (t - s) / (t2 * t);
    c2 = (t2 + 2 * c - 2) / (2 * t2 * t2);
    c3 = (2 * t - 3 * s + t * c) / (2 * t2 * t2 * t);
  }
  double WP[9], PW[9], WPW[9], WWP[9], PWW[9], WPWW[9], WWPW[9];
  mat3_mul(W, P, WP);
  mat3_mul(P, W, PW);
  mat3_mul(WP, W, WPW);
  mat3_mul(W, WP, WWP);
  mat3_mul(PW, W, PWW);
  mat3_mul(WPW, W, WPWW);
  mat3_mul(W, WPW, WWPW);
  for (int i = 0; i < 9; i++)
    Q[i] = 0.5 * P[i] + c1 * (WP[i] + PW[i] + WPW[i]) + c2 * (WWP[i] + PWW[i] - 3 * WPW[i]) +
           c3 * (WPWW[i] + WWPW[i]);
}
/* Jl_SE3(tau) in [w,v] order: [[Jl(w),0],[Q(v,w),Jl(w)]] */
static void se3_ljac(const double tau[6], double J[36]) {
  double Jl[9], Q[9];
  so3_ljac(tau, Jl);
  se3_Q(tau + 3, tau, Q);
  memset(J, 0, 36 * sizeof(double));
  for (int i = 0; i < 3; i++)
    for (int j = 0; j < 3; j++) {
      J[6 * i + j] = Jl[3 * i + j];
      J[6 * (i + 3) + j] = Q[3 * i + j];
      J[6 * (i + 3) + j + 3] = Jl[3 * i + j];
    }
}
static void se3_rjac(const double tau[6], double J[36]) {
  double m[6];
  for (int i = 0; i < 6; i++) m[i] = -tau[i];
  se3_ljac(m, J);
}
/* Jr_SE3(tau)^-1 = [[Jr^-1,0],[-Jr^-1 Qr Jr^-1, Jr^-1]], Qr = Q(-v,-w) */
static void se3_rjacinv(const double tau[6], double J[36]) {
  double m[6], Ji[9], Q[9], T[9], B[9];
  for (int i = 0; i < 6; i++) m[i] = -tau[i];
  so3_ljacinv(m, Ji); /* rjacinv(w) = ljacinv(-w) */
  se3_Q(m + 3, m, Q);
  mat3_mul(Ji, Q, T);
  mat3_mul(T, Ji, B);
  memset(J, 0, 36 * sizeof(double));
  for (int i = 0; i < 3; i++)
    for (int j = 0; j < 3; j++) {
      J[6 * i + j] = Ji[3 * i + j];
      J[6 * (i + 3) + j] = -B[3 * i + j];
      J[6 * (i + 3) + j + 3] = Ji[3 * i + j];
    }
}
/* manif rminus: A - B = Log(B^-1 A)  (operator- used at traopt_controller.py:2683, :2804, :2883) */
static void se3_rminus(const se3_t *A, const se3_t *B, double tau[6]) {
  se3_t Bi, C;
  se3_inverse(B, &Bi);
  se3_compose(&Bi, A, &C);
  se3_log(&C, tau);
}
/* manif lminus(A,B) = Log(A B^-1), J_A = Jr^-1(e) Ad(B)  (traopt_cost.py:668, :778, :826) */
static void se3_lminus(const se3_t *A, const se3_t *B, double e[6], double *J_A /* 36 or NULL */) {
  se3_t Bi, C;
  se3_inverse(B, &Bi);
  se3_compose(A, &Bi, &C);
  se3_log(&C, e);
  if (J_A) {
    double Ji[36], Ad[36];
    se3_rjacinv(e, Ji);
    se3_adj(B, Ad);
    mm(6, 6, 6, Ji, Ad, J_A);
  }
}
/* adjoint / coadjoint (traopt_utilis.py:75-92) */
static void ad6(const double xi[6], double A[36]) {
  double W[9], V[9];
  skew3(xi, W);
  skew3(xi + 3, V);
  memset(A, 0, 36 * sizeof(double));
  for (int i = 0; i < 3; i++)
    for (int j = 0; j < 3; j++) {
      A[6 * i + j] = W[3 * i + j];
      A[6 * (i + 3) + j] = V[3 * i + j];
      A[6 * (i + 3) + j + 3] = W[3 * i + j];
    }
}
static void coad6(const double xi[6], double A[36]) {
  double T[36];
  ad6(xi, T);
  for (int i = 0; i < 6; i++)
    for (int j = 0; j < 6; j++) A[6 * i + j] = T[6 * j + i];
}

/* ------------------------------------------------------------------------------------------ */
/* problem description                                                                         */
/* ------------------------------------------------------------------------------------------ */
/* TOLG_DYN_SO3: SO3Dynamics + SO3TrackingQuadraticGaussNewtonCost + iLQR_Tracking_SO3{,_MS} carried in the
 * SE(3) containers with translation, linear velocity and the last three inputs identically zero
 * (J = blkdiag(J_so3, I3)): the rotational sub-problem decouples exactly, so costs, gradients, defects
 * and iterates are those of the SO(3) solver.  Pinned against the reference's SO3 run in
 * tests/golden/so3_n249_log.json. */
/* TOLG_DYN_PENDULUM3D: Pendulum3dDyanmics (traopt_dynamics.py:421-626) with the SO3 tracking cost and the
 * SO3 controllers, in the same embedding as TOLG_DYN_SO3; the pivot acceleration u in R^3 rides in
 * u[0:3].  pend_mass / pend_length are the constructor's m and length.  No artefact of the reference
 * pins this model (its only golden is a pickle no safe loader reads): parity unpinned, checked by
 * finite differences and the closed forms in tests/test_pendulum.py. */
enum { TOLG_DYN_SE3 = 0, TOLG_DYN_RIGIDBODY = 1, TOLG_DYN_DRONE = 2, TOLG_DYN_SO3 = 3, TOLG_DYN_PENDULUM3D = 4 };
#define IS_SO3(k) ((k) == TOLG_DYN_SO3 || (k) == TOLG_DYN_PENDULUM3D)

typedef struct {
  int kind;          /* SE3Dynamics / RigidBodyDynamics / DroneDynamics */
  int m;             /* action size: 6, 6, 4 */
  int N;             /* horizon */
  double dt;
  double J[36];      /* inertia diag(Ib, m I3) (any SPD 6x6 accepted) */
  double Q[144], P[144];
  double R[36];      /* m x m row-major */
  const double *q_ref;  /* (N+1) x 16 */
  const double *xi_ref; /* (N+1) x 6 */
  /* augmented-Lagrangian box input constraint (traopt_cost.py:1173-1320,
   * traopt_constraints.py:66-169); al_on = 0 disables */
  int al_on;
  const double *al_lb, *al_ub; /* m */
  const double *al_lambda;     /* N x 2m */
  const double *al_imu;        /* N x 2m (diagonal of I_mu) */
  double pend_mass, pend_length; /* Pendulum3dDyanmics m, length (other kinds: ignored) */
} tolg_problem;

typedef struct {
  double Jinv[36], Ib[9], mass, grav;
  double Pu[36];  /* 6 x m */
  double Bt[72];  /* 12 x m: F_u / dt */
  se3_t *qref;    /* N+1 manif-ised references (traopt_cost.py:614) */
} dyn_cache;

static int inv6(const double A[36], double Ai[36]) { /* np.linalg.inv */
  double M[6][12];
  for (int i = 0; i < 6; i++)
    for (int j = 0; j < 6; j++) { M[i][j] = A[6 * i + j]; M[i][j + 6] = (i == j); }
  for (int c = 0; c < 6; c++) {
    int p = c;
    for (int r = c + 1; r < 6; r++) if (fabs(M[r][c]) > fabs(M[p][c])) p = r;
    if (M[p][c] == 0) return -1;
    if (p != c) for (int j = 0; j < 12; j++) { double t = M[c][j]; M[c][j] = M[p][j]; M[p][j] = t; }
    double d = M[c][c];
    for (int j = 0; j < 12; j++) M[c][j] /= d;
    for (int r = 0; r < 6; r++) if (r != c) {
      double f = M[r][c];
      if (f != 0) for (int j = 0; j < 12; j++) M[r][j] -= f * M[c][j];
    }
  }
  for (int i = 0; i < 6; i++) for (int j = 0; j < 6; j++) Ai[6 * i + j] = M[i][j + 6];
  return 0;
}

/* SE3Dynamics.__init__ (traopt_dynamics.py:633-690); DroneDynamics.__init__ (:1214-1278);
 * RigidBodyDynamics.__init__ (:906-970) */
static int dyn_init(const tolg_problem *p, dyn_cache *c) {
  int m = p->m;
  if (inv6(p->J, c->Jinv)) return -1;
  for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) c->Ib[3 * i + j] = p->J[6 * i + j];
  c->mass = p->J[6 * 4 + 4];
  c->grav = (p->kind == TOLG_DYN_SE3 || p->kind == TOLG_DYN_SO3) ? 0.0 : 9.8; /* :1245, pendulum :466 */
  memset(c->Pu, 0, sizeof c->Pu);
  if (p->kind == TOLG_DYN_DRONE) {
    if (m != 4) return -2;
    c->Pu[0 * 4 + 0] = 1; c->Pu[1 * 4 + 1] = 1; c->Pu[2 * 4 + 2] = 1; c->Pu[5 * 4 + 3] = 1;
  } else {
    if (m != 6) return -2;
    for (int i = 0; i < 6; i++) c->Pu[i * 6 + i] = 1;
    /* the pendulum's input acts through skew(m rho) R^T u only (traopt_dynamics.py:539-540) */
    if (p->kind == TOLG_DYN_PENDULUM3D) for (int i = 0; i < 3; i++) c->Pu[i * 6 + i] = 0;
  }
  memset(c->Bt, 0, sizeof c->Bt);
  mm(6, 6, m, c->Jinv, c->Pu, c->Bt + 6 * m);
  c->qref = (se3_t *)malloc(sizeof(se3_t) * (size_t)(p->N + 1));
  for (int i = 0; i <= p->N; i++) se3_from_matrix(p->q_ref + 16 * i, &c->qref[i]);
  return 0;
}
static void dyn_free(dyn_cache *c) { free(c->qref); c->qref = NULL; }

/* fd_euler: SE3Dynamics (traopt_dynamics.py:763-787), DroneDynamics (:1373-1401),
 * RigidBodyDynamics (:1049-1077) */
static void dyn_f(const tolg_problem *p, const dyn_cache *c, const double q[16], const double xi[6],
                  const double *u, double qn[16], double xin[6]) {
  se3_t X, E, Xn;
  double tau[6], Jxi[6], co[36], rhs[6], acc[6];
  se3_from_matrix(q, &X);
  for (int i = 0; i < 6; i++) tau[i] = xi[i] * p->dt;
  se3_exp(tau, &E);
  se3_compose(&X, &E, &Xn);
  se3_to_matrix(&Xn, qn);
  mm(6, 6, 1, p->J, xi, Jxi);
  coad6(xi, co);
  mm(6, 6, 1, co, Jxi, rhs);
  if (p->kind == TOLG_DYN_PENDULUM3D) {
    /* Pendulum3dDyanmics.fd_euler (traopt_dynamics.py:531-552): rho = l/2 (0,0,-1),
     * g_term = skew(m g rho) R^T (0,0,-1), M = skew(m rho) R^T u, xi+ = xi + J^-1(ad^T J xi + g_term + M) dt */
    double R[9], rtd[3], rtu[3], Sg[9], Sm[9], gt[3], Mt[3];
    const double rho[3] = {0, 0, -p->pend_length / 2};
    double mgr[3], mr[3];
    for (int i = 0; i < 3; i++) { mgr[i] = p->pend_mass * c->grav * rho[i]; mr[i] = p->pend_mass * rho[i]; }
    rotm_from_quat(X.q, R);
    for (int i = 0; i < 3; i++) {
      rtd[i] = -R[3 * 2 + i];
      rtu[i] = R[3 * 0 + i] * u[0] + R[3 * 1 + i] * u[1] + R[3 * 2 + i] * u[2];
    }
    skew3(mgr, Sg); skew3(mr, Sm);
    mat3_vec(Sg, rtd, gt); mat3_vec(Sm, rtu, Mt);
    for (int i = 0; i < 3; i++) rhs[i] = rhs[i] + gt[i] + Mt[i];
  } else if (c->grav != 0.0) { /* g_acc = [0; m g R^T (0,0,-1)] */
    double R[9];
    rotm_from_quat(X.q, R);
    for (int i = 0; i < 3; i++) rhs[3 + i] += c->mass * c->grav * (-R[3 * 2 + i]);
  }
  for (int i = 0; i < 6; i++)
    for (int j = 0; j < p->m; j++) rhs[i] += c->Pu[i * p->m + j] * u[j];
  mm(6, 6, 1, c->Jinv, rhs, acc);
  for (int i = 0; i < 6; i++) xin[i] = xi[i] + acc[i] * p->dt;
}

/* f_x: SE3Dynamics (traopt_dynamics.py:802-837), DroneDynamics (:1416-1469),
 * RigidBodyDynamics (:1092-1145).  Literal quirks kept: coadjoint of the manif-ordered twist
 * [v, w] (:832, :1464); gravity Jacobian without m*g (:1445-1458). */
static void dyn_fx(const tolg_problem *p, const dyn_cache *c, const double q[16], const double xi[6],
                   const double *u, double Fx[144]) {
  se3_t X, E, Ei;
  double tau[6], Jqq[36], Jqxi[36];
  se3_from_matrix(q, &X);
  for (int i = 0; i < 6; i++) tau[i] = xi[i] * p->dt;
  se3_exp(tau, &E);
  se3_inverse(&E, &Ei);
  se3_adj(&Ei, Jqq);   /* rplus: J_X = Ad(Exp(tau))^-1 */
  se3_rjac(tau, Jqxi); /* rplus: J_tau = Jr(tau) */
  double G[36], SIw[9], Sv[9], Ibw[3];
  mat3_vec(c->Ib, xi, Ibw);
  skew3(Ibw, SIw);
  skew3(xi + 3, Sv);
  memset(G, 0, sizeof G);
  for (int i = 0; i < 3; i++)
    for (int j = 0; j < 3; j++) {
      G[6 * i + j] = SIw[3 * i + j];
      G[6 * i + j + 3] = c->mass * Sv[3 * i + j];
      G[6 * (i + 3) + j] = c->mass * Sv[3 * i + j];
    }
  double sw[6] = {xi[3], xi[4], xi[5], xi[0], xi[1], xi[2]}; /* manif coeffs [v, w] */
  double co[36], coJ[36], H[36];
  if (IS_SO3(p->kind)) {
    /* SO3Dynamics.f_x (the pendulum's H is the same expression, :566-567) (traopt_dynamics.py:385-400): G = skew(J w), H = J^-1 (smallAdj(w)^T J + G) with
     * smallAdj(w) = skew(w) -- no swapped-twist quirk on SO(3); the unused v block stays the identity */
    double Sw[9], JJ[9], Ji3[9], T3[9], H3[9];
    skew3(xi, Sw);
    for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) { JJ[3 * i + j] = p->J[6 * i + j]; Ji3[3 * i + j] = c->Jinv[6 * i + j]; }
    double SwT[9];
    mat3_T(Sw, SwT);
    mat3_mul(SwT, JJ, T3);
    for (int i = 0; i < 9; i++) T3[i] += SIw[i];
    mat3_mul(Ji3, T3, H3);
    memset(H, 0, sizeof H);
    for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) H[6 * i + j] = H3[3 * i + j];
  } else {
    coad6(sw, co);
    mm(6, 6, 6, co, p->J, coJ);
    for (int i = 0; i < 36; i++) coJ[i] += G[i];
    mm(6, 6, 6, c->Jinv, coJ, H);
  }
  double L[36];
  memset(L, 0, sizeof L);
  if (p->kind == TOLG_DYN_PENDULUM3D) {
    /* Pendulum3dDyanmics.f_x (traopt_dynamics.py:574-588): manif gives J_inv = -Ad(q) = -R and, for
     * q^-1.act(v), J_wrt_q^-1 = -R^T skew(v), so J_act J_inv = R^T skew(v) R;
     * L1 = skew(m g rho) J J_inv (v = down), L2 = skew(m rho) J J_inv (v = u), L = J^-1 (L1 + L2) */
    double R[9], Rt[9], Sv[9], T1[9], JJ[9], Sg[9], Sm[9], L1[9], L2[9], Ji3[9], L3[9];
    const double down[3] = {0, 0, -1.0};
    const double rho[3] = {0, 0, -p->pend_length / 2};
    double mgr[3], mr[3], uu[3] = {u ? u[0] : 0.0, u ? u[1] : 0.0, u ? u[2] : 0.0};
    for (int i = 0; i < 3; i++) { mgr[i] = p->pend_mass * c->grav * rho[i]; mr[i] = p->pend_mass * rho[i]; }
    rotm_from_quat(X.q, R); mat3_T(R, Rt);
    skew3(mgr, Sg); skew3(mr, Sm);
    skew3(down, Sv); mat3_mul(Rt, Sv, T1); mat3_mul(T1, R, JJ); mat3_mul(Sg, JJ, L1);
    skew3(uu, Sv); mat3_mul(Rt, Sv, T1); mat3_mul(T1, R, JJ); mat3_mul(Sm, JJ, L2);
    for (int i = 0; i < 9; i++) L1[i] += L2[i];
    for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) Ji3[3 * i + j] = c->Jinv[6 * i + j];
    mat3_mul(Ji3, L1, L3);
    for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) L[6 * i + j] = L3[3 * i + j] * p->dt;
  } else if (c->grav != 0.0) { /* J_v_R = skew(R^T (0,0,-1)) */
    double R[9], rte[3], S[9], Jxiq[36];
    rotm_from_quat(X.q, R);
    for (int i = 0; i < 3; i++) rte[i] = -R[3 * 2 + i];
    skew3(rte, S);
    memset(Jxiq, 0, sizeof Jxiq);
    for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) Jxiq[6 * (i + 3) + j] = S[3 * i + j];
    mm(6, 6, 6, c->Jinv, Jxiq, L);
    for (int i = 0; i < 36; i++) L[i] *= p->dt;
  }
  for (int i = 0; i < 6; i++)
    for (int j = 0; j < 6; j++) {
      Fx[12 * i + j] = Jqq[6 * i + j];
      Fx[12 * i + j + 6] = Jqxi[6 * i + j] * p->dt;
      Fx[12 * (i + 6) + j] = L[6 * i + j];
      Fx[12 * (i + 6) + j + 6] = (i == j) + H[6 * i + j] * p->dt;
    }
}
/* f_u = Bt * dt (traopt_dynamics.py:839-850, :1471-1482) */
static void dyn_fu(const tolg_problem *p, const dyn_cache *c, const double q[16], double Fu[72]) {
  for (int i = 0; i < 12 * p->m; i++) Fu[i] = c->Bt[i] * p->dt;
  if (p->kind == TOLG_DYN_PENDULUM3D) {
    /* Pendulum3dDyanmics.f_u (traopt_dynamics.py:596-609): bt = J^-1 skew(m rho) J_act_wrt_v with
     * J_act_wrt_v = R^T (the rotation of q^-1); state dependent */
    se3_t X;
    double R[9], Rt[9], Sm[9], T1[9], Ji3[9], bt[9];
    const double mr[3] = {0, 0, -p->pend_mass * p->pend_length / 2};
    se3_from_matrix(q, &X);
    rotm_from_quat(X.q, R); mat3_T(R, Rt);
    skew3(mr, Sm); mat3_mul(Sm, Rt, T1);
    for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) Ji3[3 * i + j] = c->Jinv[6 * i + j];
    mat3_mul(Ji3, T1, bt);
    for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) Fu[(6 + i) * p->m + j] = bt[3 * i + j] * p->dt;
  }
}

/* ------------------------------------------------------------------------------------------ */
/* cost: SE3TrackingQuadraticGaussNewtonCost (traopt_cost.py:570-867) + ALConstrainedCost       */
/* ------------------------------------------------------------------------------------------ */
/* g = [lb-u; u-ub] (traopt_constraints.py:130-133); terminal -> zeros */
static double al_terms(const tolg_problem *p, int i, const double *u, double *lu, double *luu) {
  int m = p->m;
  double val = 0;
  for (int j = 0; j < 2 * m; j++) {
    int k = j % m;
    double g = (j < m) ? (p->al_lb[k] - u[k]) : (u[k] - p->al_ub[k]);
    double gu = (j < m) ? -1.0 : 1.0;
    double lam = p->al_lambda[(size_t)i * 2 * m + j], imu = p->al_imu[(size_t)i * 2 * m + j];
    val += lam * g + 0.5 * g * imu * g;        /* traopt_cost.py:1219-1224 */
    if (lu) lu[k] += gu * (lam + imu * g);     /* :1262-1266 */
    if (luu) luu[k * m + k] += gu * imu * gu;  /* :1302-1306 */
  }
  return val;
}
/* _l / _l_terminal (traopt_cost.py:675-738) */
static double cost_l(const tolg_problem *p, const dyn_cache *c, const double q[16], const double xi[6],
                     const double *u, int i, int terminal) {
  se3_t X;
  double e[6], ve[6], s = 0;
  /* SO3 cost: _l_terminal weighs with Q, not P (traopt_cost.py:434-438) */
  const double *W = (terminal && !IS_SO3(p->kind)) ? p->P : p->Q;
  se3_from_matrix(q, &X);
  se3_lminus(&X, &c->qref[i], e, NULL);
  for (int a = 0; a < 6; a++) ve[a] = xi[a] - p->xi_ref[6 * i + a];
  for (int a = 0; a < 6; a++)
    for (int b = 0; b < 6; b++) s += e[a] * W[12 * a + b] * e[b];
  for (int a = 0; a < 6; a++)
    for (int b = 0; b < 6; b++) s += ve[a] * W[12 * (a + 6) + b + 6] * ve[b];
  if (!terminal) {
    for (int a = 0; a < p->m; a++)
      for (int b = 0; b < p->m; b++) s += u[a] * p->R[a * p->m + b] * u[b];
    if (p->al_on) s += al_terms(p, i, u, NULL, NULL);
  }
  return s;
}
/* l_x (traopt_cost.py:758-790) and l_xx (:806-839) share J_e = Jmnf2J(lminus Jacobian) */
static void cost_lx_lxx(const tolg_problem *p, const dyn_cache *c, const double q[16], const double xi[6],
                        int i, int terminal, double lx[12], double lxx[144]) {
  se3_t X;
  double e[6], Je[36], W1[36], WJ[36], We[6];
  const double *W = terminal ? p->P : p->Q;
  /* SO3 cost: l_x always uses Q (traopt_cost.py:480-483), only l_xx switches to P (:530-531) */
  const double *Wg = (IS_SO3(p->kind)) ? p->Q : W;
  se3_from_matrix(q, &X);
  se3_lminus(&X, &c->qref[i], e, Je);
  for (int a = 0; a < 6; a++) for (int b = 0; b < 6; b++) W1[6 * a + b] = Wg[12 * a + b];
  mm(6, 6, 1, W1, e, We);
  for (int a = 0; a < 6; a++) {
    double s = 0;
    for (int b = 0; b < 6; b++) s += 2 * Je[6 * b + a] * We[b];
    lx[a] = s;
  }
  for (int a = 0; a < 6; a++) {
    double s = 0;
    for (int b = 0; b < 6; b++) s += 2 * Wg[12 * (a + 6) + b + 6] * (xi[b] - p->xi_ref[6 * i + b]);
    lx[a + 6] = s;
  }
  for (int a = 0; a < 6; a++) for (int b = 0; b < 6; b++) W1[6 * a + b] = W[12 * a + b];
  mm(6, 6, 6, W1, Je, WJ);
  memset(lxx, 0, 144 * sizeof(double));
  for (int a = 0; a < 6; a++)
    for (int b = 0; b < 6; b++) {
      double s = 0;
      for (int k = 0; k < 6; k++) s += 2 * Je[6 * k + a] * WJ[6 * k + b];
      lxx[12 * a + b] = s;
      lxx[12 * (a + 6) + b + 6] = 2 * W[12 * (a + 6) + b + 6];
    }
}
/* l_u = 2 R u (:792-804), l_uu = 2 R (:855-867) (+ AL terms) */
static void cost_lu_luu(const tolg_problem *p, const double *u, int i, double *lu, double *luu) {
  int m = p->m;
  for (int a = 0; a < m; a++) {
    double s = 0;
    for (int b = 0; b < m; b++) { s += 2 * p->R[a * m + b] * u[b]; luu[a * m + b] = 2 * p->R[a * m + b]; }
    lu[a] = s;
  }
  if (p->al_on) al_terms(p, i, u, lu, luu);
}

/* ------------------------------------------------------------------------------------------ */
/* linear algebra used by the controllers                                                      */
/* ------------------------------------------------------------------------------------------ */
/* is_pos_def (traopt_utilis.py:320-329): symmetric + np.linalg.cholesky succeeds */
static int is_pos_def(int n, const double *A) {
  double L[36];
  for (int i = 0; i < n; i++) for (int j = 0; j < n; j++) if (A[i * n + j] != A[j * n + i]) return 0;
  for (int j = 0; j < n; j++) {
    double d = A[j * n + j];
    for (int k = 0; k < j; k++) d -= L[j * n + k] * L[j * n + k];
    if (!(d > 0.0)) return 0;
    d = sqrt(d);
    L[j * n + j] = d;
    for (int i = j + 1; i < n; i++) {
      double s = A[i * n + j];
      for (int k = 0; k < j; k++) s -= L[i * n + k] * L[j * n + k];
      L[i * n + j] = s / d;
    }
  }
  return 1;
}
/* np.linalg.solve (LAPACK gesv: LU with partial pivoting); B is n x r, overwritten with X */
static int lu_solve(int n, int r, const double *A, double *B) {
  double M[36];
  memcpy(M, A, sizeof(double) * (size_t)(n * n));
  for (int c = 0; c < n; c++) {
    int pv = c;
    for (int i = c + 1; i < n; i++) if (fabs(M[i * n + c]) > fabs(M[pv * n + c])) pv = i;
    if (M[pv * n + c] == 0.0) return -1;
    if (pv != c) {
      for (int j = 0; j < n; j++) { double t = M[c * n + j]; M[c * n + j] = M[pv * n + j]; M[pv * n + j] = t; }
      for (int j = 0; j < r; j++) { double t = B[c * r + j]; B[c * r + j] = B[pv * r + j]; B[pv * r + j] = t; }
    }
    for (int i = c + 1; i < n; i++) {
      double f = M[i * n + c] / M[c * n + c];
      M[i * n + c] = f;
      for (int j = c + 1; j < n; j++) M[i * n + j] -= f * M[c * n + j];
      for (int j = 0; j < r; j++) B[i * r + j] -= f * B[c * r + j];
    }
  }
  for (int i = n - 1; i >= 0; i--)
    for (int j = 0; j < r; j++) {
      double s = B[i * r + j];
      for (int k = i + 1; k < n; k++) s -= M[i * n + k] * B[k * r + j];
      B[i * r + j] = s / M[i * n + i];
    }
  return 0;
}

/* ------------------------------------------------------------------------------------------ */
/* controller workspace                                                                        */
/* ------------------------------------------------------------------------------------------ */
typedef struct {
  int N, m;
  double *d, *Fx, *Fu, *L, *Lx, *Lu, *Lxx, *Luu, *k, *K, *Vx, *Vxx;
  double *xq, *xxi, *us;       /* current trajectory */
  double *nq, *nxi, *nus;      /* candidate trajectory */
  double *xerr, *uerr;         /* rollout deviations (line_search=True) */
  double mu, delta;
} ws_t;

static void ws_alloc(ws_t *w, int N, int m) {
  w->N = N; w->m = m;
  size_t n = (size_t)N;
  w->d = calloc(n * 12, sizeof(double)); w->Fx = calloc(n * 144, sizeof(double)); w->Fu = calloc(n * 12 * m, sizeof(double));
  w->L = calloc(n + 1, sizeof(double)); w->Lx = calloc((n + 1) * 12, sizeof(double)); w->Lu = calloc(n * m, sizeof(double));
  w->Lxx = calloc((n + 1) * 144, sizeof(double)); w->Luu = calloc(n * m * m, sizeof(double));
  w->k = calloc(n * m, sizeof(double)); w->K = calloc(n * m * 12, sizeof(double));
  w->Vx = calloc((n + 1) * 12, sizeof(double)); w->Vxx = calloc((n + 1) * 144, sizeof(double));
  w->xq = calloc((n + 1) * 16, sizeof(double)); w->xxi = calloc((n + 1) * 6, sizeof(double)); w->us = calloc(n * m, sizeof(double));
  w->nq = calloc((n + 1) * 16, sizeof(double)); w->nxi = calloc((n + 1) * 6, sizeof(double)); w->nus = calloc(n * m, sizeof(double));
  w->xerr = calloc((n + 1) * 12, sizeof(double)); w->uerr = calloc(n * m, sizeof(double));
}
static void ws_free(ws_t *w) {
  free(w->d); free(w->Fx); free(w->Fu); free(w->L); free(w->Lx); free(w->Lu); free(w->Lxx); free(w->Luu);
  free(w->k); free(w->K); free(w->Vx); free(w->Vxx); free(w->xq); free(w->xxi); free(w->us);
  free(w->nq); free(w->nxi); free(w->nus); free(w->xerr); free(w->uerr);
}

/* _compute_defect for one knot (traopt_controller.py:2790-2810, :2882-2888) */
static void defect_knot(const tolg_problem *p, const dyn_cache *c, const double *q, const double *xi,
                        const double *u, const double *qnext, const double *xinext, double d[12]) {
  double fq[16], fxi[6];
  se3_t F, Xn;
  dyn_f(p, c, q, xi, u, fq, fxi);
  se3_from_matrix(fq, &F);
  se3_from_matrix(qnext, &Xn);
  se3_rminus(&F, &Xn, d);
  for (int a = 0; a < 6; a++) d[6 + a] = fxi[a] - xinext[a];
}

/* _linearization: MS (traopt_controller.py:2823-2910) / SS (:2098-2176; ms = 0 skips defects) */
static void linearize(const tolg_problem *p, const dyn_cache *c, ws_t *w, int ms) {
  int N = p->N, m = p->m;
  for (int i = 0; i < N; i++) {
    const double *q = w->xq + 16 * i, *xi = w->xxi + 6 * i, *u = w->us + m * i;
    if (ms) defect_knot(p, c, q, xi, u, w->xq + 16 * (i + 1), w->xxi + 6 * (i + 1), w->d + 12 * i);
    else memset(w->d + 12 * i, 0, 12 * sizeof(double));
    dyn_fx(p, c, q, xi, u, w->Fx + 144 * i);
    dyn_fu(p, c, q, w->Fu + 12 * m * i);
    w->L[i] = cost_l(p, c, q, xi, u, i, 0);
    cost_lx_lxx(p, c, q, xi, i, 0, w->Lx + 12 * i, w->Lxx + 144 * i);
    cost_lu_luu(p, u, i, w->Lu + m * i, w->Luu + m * m * i);
  }
  w->L[N] = cost_l(p, c, w->xq + 16 * N, w->xxi + 6 * N, NULL, N, 1);
  cost_lx_lxx(p, c, w->xq + 16 * N, w->xxi + 6 * N, N, 1, w->Lx + 12 * N, w->Lxx + 144 * N);
}

/* _backward_pass + _Q: MS (traopt_controller.py:2912-3068), SS (:2178-2321; d = 0).
 * returns 1 if "exceeded max regularization term" was hit at some knot. */
static int backward(const tolg_problem *p, ws_t *w, double max_reg) {
  int N = p->N, m = p->m, warned = 0;
  const double mu_min = 1e-6, delta0 = 2.0;
  memcpy(w->Vx + 12 * N, w->Lx + 12 * N, 12 * sizeof(double));
  memcpy(w->Vxx + 144 * N, w->Lxx + 144 * N, 144 * sizeof(double));
  for (int i = N - 1; i >= 0; i--) {
    const double *fx = w->Fx + 144 * i, *fu = w->Fu + 12 * m * i, *Vx = w->Vx + 12 * (i + 1),
                 *Vxx = w->Vxx + 144 * (i + 1), *d = w->d + 12 * i;
    double v[12], Qx[12], Qu[6], Qxx[144], Qux[72], Quu[36], T[144], Vr[144], S[36];
    mm(12, 12, 1, Vxx, d, v);
    for (int a = 0; a < 12; a++) v[a] += Vx[a];
    mtm(12, 12, 1, fx, v, Qx);
    for (int a = 0; a < 12; a++) Qx[a] += w->Lx[12 * i + a];
    mtm(12, m, 1, fu, v, Qu);
    for (int a = 0; a < m; a++) Qu[a] += w->Lu[m * i + a];
    mtm(12, 12, 12, fx, Vxx, T); /* f_x^T V_xx */
    mm(12, 12, 12, T, fx, Qxx);
    for (int a = 0; a < 144; a++) Qxx[a] += w->Lxx[144 * i + a];
    for (;;) {
      memcpy(Vr, Vxx, sizeof Vr);
      for (int a = 0; a < 12; a++) Vr[13 * a] += w->mu;
      double FT[72];
      mtm(12, m, 12, fu, Vr, FT); /* f_u^T (V_xx + mu I) : m x 12 */
      mm(m, 12, 12, FT, fx, Qux); /* l_ux = 0 */
      mm(m, 12, m, FT, fu, Quu);
      for (int a = 0; a < m * m; a++) Quu[a] += w->Luu[m * m * i + a];
      for (int a = 0; a < m; a++) for (int b = 0; b < m; b++) S[a * m + b] = Quu[a * m + b] + Quu[b * m + a];
      if (!is_pos_def(m, S)) {
        w->delta = fmax(1.0, w->delta) * delta0;
        w->mu = fmax(mu_min, w->mu * w->delta);
        if (max_reg > 0 && w->mu >= max_reg) { warned = 1; break; }
      } else {
        w->delta = fmin(1.0, w->delta) / delta0;
        w->mu *= w->delta;
        if (w->mu <= mu_min) w->mu = 0.0;
        break;
      }
    }
    double *k = w->k + m * i, *K = w->K + 12 * m * i;
    for (int a = 0; a < m; a++) k[a] = Qu[a];
    memcpy(K, Qux, sizeof(double) * (size_t)(12 * m));
    lu_solve(m, 1, Quu, k);
    lu_solve(m, 12, Quu, K);
    for (int a = 0; a < m; a++) k[a] = -k[a];
    for (int a = 0; a < 12 * m; a++) K[a] = -K[a];
    /* V_x = Q_x + K^T Q_uu k + K^T Q_u + Q_ux^T k */
    double Quuk[6], KtQ[72], *Vxo = w->Vx + 12 * i, *Vo = w->Vxx + 144 * i;
    mm(m, m, 1, Quu, k, Quuk);
    for (int a = 0; a < 12; a++) {
      double s = Qx[a];
      for (int b = 0; b < m; b++) s += K[12 * b + a] * Quuk[b];
      for (int b = 0; b < m; b++) s += K[12 * b + a] * Qu[b] + Qux[12 * b + a] * k[b];
      Vxo[a] = s;
    }
    /* V_xx = Q_xx + K^T Q_uu K + K^T Q_ux + Q_ux^T K ; symmetrise */
    mm(m, m, 12, Quu, K, KtQ); /* Q_uu K : m x 12 */
    for (int a = 0; a < 12; a++)
      for (int b = 0; b < 12; b++) {
        double s = Qxx[12 * a + b];
        for (int l = 0; l < m; l++) s += K[12 * l + a] * KtQ[12 * l + b];
        for (int l = 0; l < m; l++) s += K[12 * l + a] * Qux[12 * l + b] + Qux[12 * l + a] * K[12 * l + b];
        T[12 * a + b] = s;
      }
    for (int a = 0; a < 12; a++)
      for (int b = 0; b < 12; b++) Vo[12 * a + b] = 0.5 * (T[12 * a + b] + T[12 * b + a]);
  }
  return warned;
}

/* _gradient_wrt_control, MS (traopt_controller.py:3070-3093) */
static double grad_ms(const tolg_problem *p, const ws_t *w) {
  int N = p->N, m = p->m;
  double sum = 0;
  for (int t = N - 1; t >= 0; t--) {
    double v[12], g[6], n2 = 0;
    mtm(12, 12, 1, w->Vxx + 144 * (t + 1), w->d + 12 * t, v); /* V_xx^T d */
    for (int a = 0; a < 12; a++) v[a] += w->Vx[12 * (t + 1) + a];
    mtm(12, m, 1, w->Fu + 12 * m * t, v, g);
    for (int a = 0; a < m; a++) { g[a] += w->Lu[m * t + a]; n2 += g[a] * g[a]; }
    sum += sqrt(n2);
  }
  return sum / N;
}
/* _gradient_wrt_control, SS adjoint recursion (traopt_controller.py:2323-2349) */
static double grad_ss(const tolg_problem *p, const ws_t *w) {
  int N = p->N, m = p->m;
  double pa[12], sum = 0;
  memcpy(pa, w->Lx + 12 * N, sizeof pa);
  for (int t = N - 1; t >= 0; t--) {
    double g[6], np_[12], n2 = 0;
    mtm(12, m, 1, w->Fu + 12 * m * t, pa, g);
    for (int a = 0; a < m; a++) { g[a] += w->Lu[m * t + a]; n2 += g[a] * g[a]; }
    mtm(12, 12, 1, w->Fx + 144 * t, pa, np_);
    for (int a = 0; a < 12; a++) pa[a] = w->Lx[12 * t + a] + np_[a];
    sum += sqrt(n2);
  }
  return sum / N;
}

/* _trajectory_cost (traopt_controller.py:2742-2754, :2084-2096) */
static double traj_cost(const tolg_problem *p, const dyn_cache *c, const double *xq, const double *xxi,
                        const double *us) {
  double s = 0;
  for (int i = 0; i < p->N; i++) s += cost_l(p, c, xq + 16 * i, xxi + 6 * i, us + p->m * i, i, 0);
  return s + cost_l(p, c, xq + 16 * p->N, xxi + 6 * p->N, NULL, p->N, 1);
}
/* _compute_defect + _compute_defect_norm (traopt_controller.py:2790-2821) */
static double defect_norm_traj(const tolg_problem *p, const dyn_cache *c, const double *xq,
                               const double *xxi, const double *us) {
  double s = 0, d[12];
  for (int i = 0; i < p->N; i++) {
    defect_knot(p, c, xq + 16 * i, xxi + 6 * i, us + p->m * i, xq + 16 * (i + 1), xxi + 6 * (i + 1), d);
    for (int a = 0; a < 12; a++) s += d[a] * d[a];
  }
  return sqrt(s);
}

/* state deviation [Log(q^-1 q_new); xi_new - xi] (traopt_controller.py:2680-2687, :2056-2062) */
static void state_err(const double *q, const double *xi, const double *qn, const double *xin, double e[12]) {
  se3_t A, B;
  se3_from_matrix(qn, &A);
  se3_from_matrix(q, &B);
  se3_rminus(&A, &B, e);
  for (int a = 0; a < 6; a++) e[6 + a] = xin[a] - xi[a];
}

/* MS _rollout (traopt_controller.py:2641-2740); linear = 1 selects rollout == 'linear' */
static void rollout_ms(const tolg_problem *p, const dyn_cache *c, ws_t *w, double alpha, int linear) {
  int N = p->N, m = p->m;
  memcpy(w->nq, w->xq, 16 * sizeof(double));
  memcpy(w->nxi, w->xxi, 6 * sizeof(double));
  for (int i = 0; i < N; i++) {
    const double *q = w->xq + 16 * i, *xi = w->xxi + 6 * i, *qn = w->nq + 16 * i, *xin = w->nxi + 6 * i;
    const double *qnext = w->xq + 16 * (i + 1), *xinext = w->xxi + 6 * (i + 1);
    double *e = w->xerr + 12 * i, *du = w->uerr + m * i, *un = w->nus + m * i;
    state_err(q, xi, qn, xin, e);
    for (int a = 0; a < m; a++) {
      double s = alpha * w->k[m * i + a];
      for (int b = 0; b < 12; b++) s += w->K[12 * m * i + 12 * a + b] * e[b];
      du[a] = s;
      un[a] = w->us[m * i + a] + s;
    }
    const double *d = w->d + 12 * i;
    se3_t Xnext, D, R;
    se3_from_matrix(qnext, &Xnext);
    if (!linear) {
      double fqn[16], fxin[6], fq[16], fxi[6], ad[6];
      dyn_f(p, c, qn, xin, un, fqn, fxin);
      dyn_f(p, c, q, xi, w->us + m * i, fq, fxi);
      se3_t Fn, F, Fi, T1, T2;
      se3_from_matrix(fqn, &Fn);
      se3_from_matrix(fq, &F);
      for (int a = 0; a < 6; a++) ad[a] = alpha * d[a];
      se3_exp(ad, &D);
      se3_inverse(&F, &Fi);
      se3_compose(&Xnext, &D, &T1);
      se3_compose(&T1, &Fi, &T2);
      se3_compose(&T2, &Fn, &R);
      se3_to_matrix(&R, w->nq + 16 * (i + 1));
      for (int a = 0; a < 6; a++) w->nxi[6 * (i + 1) + a] = xinext[a] + fxin[a] - fxi[a] + alpha * d[6 + a];
    } else {
      double lin[12], tau[6];
      const double *fx = w->Fx + 144 * i, *fu = w->Fu + 12 * m * i;
      for (int a = 0; a < 12; a++) {
        double s = 0;
        for (int b = 0; b < 12; b++) s += fx[12 * a + b] * e[b];
        for (int b = 0; b < m; b++) s += fu[m * a + b] * du[b];
        lin[a] = s;
      }
      for (int a = 0; a < 6; a++) tau[a] = lin[a] + alpha * d[a];
      se3_exp(tau, &D);
      se3_compose(&Xnext, &D, &R);
      se3_to_matrix(&R, w->nq + 16 * (i + 1));
      for (int a = 0; a < 6; a++) w->nxi[6 * (i + 1) + a] = xinext[a] + lin[6 + a] + alpha * d[6 + a];
    }
  }
  state_err(w->xq + 16 * N, w->xxi + 6 * N, w->nq + 16 * N, w->nxi + 6 * N, w->xerr + 12 * N);
}

/* SS _rollout (traopt_controller.py:2030-2082) */
static void rollout_ss(const tolg_problem *p, const dyn_cache *c, ws_t *w, double alpha, int linear) {
  int N = p->N, m = p->m;
  memcpy(w->nq, w->xq, 16 * sizeof(double));
  memcpy(w->nxi, w->xxi, 6 * sizeof(double));
  for (int i = 0; i < N; i++) {
    const double *q = w->xq + 16 * i, *xi = w->xxi + 6 * i, *qn = w->nq + 16 * i, *xin = w->nxi + 6 * i;
    double e[12], du[6], *un = w->nus + m * i;
    state_err(q, xi, qn, xin, e);
    for (int a = 0; a < m; a++) {
      double s = alpha * w->k[m * i + a];
      for (int b = 0; b < 12; b++) s += w->K[12 * m * i + 12 * a + b] * e[b];
      du[a] = s;
      un[a] = w->us[m * i + a] + s;
    }
    if (!linear) {
      dyn_f(p, c, qn, xin, un, w->nq + 16 * (i + 1), w->nxi + 6 * (i + 1));
    } else {
      double lin[12];
      const double *fx = w->Fx + 144 * i, *fu = w->Fu + 12 * m * i;
      for (int a = 0; a < 12; a++) {
        double s = 0;
        for (int b = 0; b < 12; b++) s += fx[12 * a + b] * e[b];
        for (int b = 0; b < m; b++) s += fu[m * a + b] * du[b];
        lin[a] = s;
      }
      se3_t Xnext, D, R;
      se3_from_matrix(w->xq + 16 * (i + 1), &Xnext);
      se3_exp(lin, &D);
      se3_compose(&Xnext, &D, &R);
      se3_to_matrix(&R, w->nq + 16 * (i + 1));
      for (int a = 0; a < 6; a++) w->nxi[6 * (i + 1) + a] = w->xxi[6 * (i + 1) + a] + lin[6 + a];
    }
  }
}

/* _expected_cost_change (traopt_controller.py:2756-2769); l_ux = 0 */
static void expected_cost_change(const tolg_problem *p, const ws_t *w, double ecc[2]) {
  int N = p->N, m = p->m;
  double c1 = 0, c2 = 0;
  for (int i = 0; i <= N; i++) {
    const double *e = w->xerr + 12 * i;
    for (int a = 0; a < 12; a++) {
      c1 += w->Lx[12 * i + a] * e[a];
      for (int b = 0; b < 12; b++) c2 += e[a] * w->Lxx[144 * i + 12 * a + b] * e[b];
    }
    if (i < N) {
      const double *du = w->uerr + m * i;
      for (int a = 0; a < m; a++) {
        c1 += w->Lu[m * i + a] * du[a];
        for (int b = 0; b < m; b++) c2 += du[a] * w->Luu[m * m * i + a * m + b] * du[b];
      }
    }
  }
  ecc[0] = c1; ecc[1] = c2;
}

/* ------------------------------------------------------------------------------------------ */
/* public entry points                                                                          */
/* ------------------------------------------------------------------------------------------ */
typedef struct {
  int max_iter;
  double tol_grad, tol_defect; /* MS: both; SS: tol_grad only */
  int line_search;             /* MS only (traopt_controller.py:2549-2590) */
  int rollout_linear;          /* rollout == 'linear' */
  double max_reg;              /* 1e10 */
} tolg_options;

typedef struct {
  double *J_hist;      /* [max_iter]   cost after iteration k (what the callback appends) */
  double *grad_hist;   /* [max_iter+1] gradient norm evaluated in iteration k */
  double *defect_hist; /* [max_iter+1] MS: defect_hist[0] initial, [k+1] after iteration k */
  double *alpha_hist;  /* [max_iter]   last alpha tried */
  double *mu_hist;     /* [max_iter]   regularisation after the backward pass */
  double *J_lin;       /* [max_iter+1] cost at linearisation (J_opt = L.sum()) */
  double *trial_J;     /* [max_iter*20] cost of every line-search rollout */
  int *n_trials;       /* [max_iter] */
  int n_iters;         /* number of callback invocations */
  int converged;       /* 1 if the gradient test fired */
  int status;          /* 0 ok, 1 max-reg warning, 2 no descent direction, 3 non-finite */
} tolg_history;

static const int MS_ALPHAS = 20, SS_ALPHAS = 13;

/* iLQR_Tracking_SE3_MS.fit (traopt_controller.py:2443-2639) */
static int ms_fit_ws(const tolg_problem *p, const tolg_options *o, const double *x0_q, const double *x0_xi,
                     const double *us_init, double *xs_q, double *xs_xi, double *us, tolg_history *h,
                     const dyn_cache *cp, ws_t *wp);
int tolg_oracle_ms_fit(const tolg_problem *p, const tolg_options *o, const double *x0_q, const double *x0_xi,
                       const double *us_init, double *xs_q, double *xs_xi, double *us, tolg_history *h) {
  dyn_cache c;
  ws_t w;
  int rc = dyn_init(p, &c);
  if (rc) return rc;
  ws_alloc(&w, p->N, p->m);
  rc = ms_fit_ws(p, o, x0_q, x0_xi, us_init, xs_q, xs_xi, us, h, &c, &w);
  ws_free(&w);
  dyn_free(&c);
  return rc;
}
/* the same with the dynamics cache and the workspace of the caller (the batch driver keeps one per thread) */
static int ms_fit_ws(const tolg_problem *p, const tolg_options *o, const double *x0_q, const double *x0_xi,
                     const double *us_init, double *xs_q, double *xs_xi, double *us, tolg_history *h,
                     const dyn_cache *cp, ws_t *wp) {
#define c (*cp)
#define w (*wp)
  int N = p->N, m = p->m;
  w.mu = 1.0; w.delta = 2.0;
  memcpy(w.us, us_init, sizeof(double) * (size_t)(N * m));
  /* _initial_guess (:3123-3136) */
  memcpy(w.xq, x0_q, 16 * sizeof(double));
  memcpy(w.xxi, x0_xi, 6 * sizeof(double));
  memcpy(w.xq + 16, p->q_ref + 16, sizeof(double) * (size_t)(16 * N));
  memcpy(w.xxi + 6, p->xi_ref + 6, sizeof(double) * (size_t)(6 * N));
  h->n_iters = 0; h->converged = 0; h->status = 0;
  double d_weight_prev = 10.0; /* _defect_mu0 */
  for (int it = 0; it < o->max_iter; it++) {
    int accepted = 0;
    double alpha = 1.0, J_new = 0, dn_new = 0;
    linearize(p, &c, &w, 1);
    double dn = 0, J_opt = 0;
    for (int i = 0; i < 12 * N; i++) dn += w.d[i] * w.d[i];
    dn = sqrt(dn);
    if (it == 0) h->defect_hist[0] = dn;
    for (int i = 0; i <= N; i++) J_opt += w.L[i];
    h->J_lin[it] = J_opt;
    if (backward(p, &w, o->max_reg)) h->status = 1;
    double g = grad_ms(p, &w);
    h->grad_hist[it] = g;
    if (g < o->tol_grad && dn < o->tol_defect) { h->converged = 1; break; }
    int ntr = 0;
    if (o->line_search) {
      double ecc[2];
      rollout_ms(p, &c, &w, 1.0, 1); /* rollout="linear" (:2550) */
      expected_cost_change(p, &w, ecc);
      double d_weight;
      if (dn < ((IS_SO3(p->kind)) ? 1e-14 : 1e-12)) d_weight = d_weight_prev; /* _defect_kappa (:2777, SO3 :1090) */
      else d_weight = fmax(10.0, 10.0 + fabs(ecc[0] + 0.5 * ecc[1]) / ((1 - 0.5) * dn));
      d_weight_prev = d_weight;
      double merit = J_opt + d_weight * dn;
      for (int a = 0; a < ((IS_SO3(p->kind)) ? SS_ALPHAS : MS_ALPHAS); a++) { /* SO3 MS: 13 alphas (:1160) */
        alpha = pow(1.1, -(double)(a * a));
        rollout_ms(p, &c, &w, alpha, o->rollout_linear);
        J_new = traj_cost(p, &c, w.nq, w.nxi, w.nus);
        dn_new = defect_norm_traj(p, &c, w.nq, w.nxi, w.nus);
        h->trial_J[20 * it + ntr++] = J_new;
        double J_exp = alpha * ecc[0] + 0.5 * alpha * alpha * ecc[1];
        double merit_new = J_new + d_weight * dn_new;
        if (merit_new - merit < 0.05 * (J_exp - alpha * d_weight * dn)) { accepted = 1; break; }
      }
    } else {
      alpha = 1.0;
      rollout_ms(p, &c, &w, alpha, o->rollout_linear);
      J_new = traj_cost(p, &c, w.nq, w.nxi, w.nus);
      dn_new = defect_norm_traj(p, &c, w.nq, w.nxi, w.nus);
      h->trial_J[20 * it + ntr++] = J_new;
      accepted = 1;
    }
    h->n_trials[it] = ntr;
    if (accepted) {
      J_opt = J_new;
      memcpy(w.xq, w.nq, sizeof(double) * (size_t)(16 * (N + 1)));
      memcpy(w.xxi, w.nxi, sizeof(double) * (size_t)(6 * (N + 1)));
      memcpy(w.us, w.nus, sizeof(double) * (size_t)(N * m));
    }
    /* on_iteration (:2621-2626) */
    h->J_hist[it] = J_opt; h->defect_hist[it + 1] = dn_new; h->alpha_hist[it] = alpha; h->mu_hist[it] = w.mu;
    h->n_iters = it + 1;
    if (!(J_opt == J_opt) || isinf(J_opt)) { h->status = 3; break; }
    if (!accepted) { h->status = 2; break; }
  }
  memcpy(xs_q, w.xq, sizeof(double) * (size_t)(16 * (N + 1)));
  memcpy(xs_xi, w.xxi, sizeof(double) * (size_t)(6 * (N + 1)));
  memcpy(us, w.us, sizeof(double) * (size_t)(N * m));
  return 0;
#undef c
#undef w
}

/* iLQR_Tracking_SE3.fit (traopt_controller.py:1880-2013) */
static int ss_fit_ws(const tolg_problem *p, const tolg_options *o, const double *x0_q, const double *x0_xi,
                     const double *us_init, double *xs_q, double *xs_xi, double *us, tolg_history *h,
                     const dyn_cache *cp, ws_t *wp);
int tolg_oracle_ss_fit(const tolg_problem *p, const tolg_options *o, const double *x0_q, const double *x0_xi,
                       const double *us_init, double *xs_q, double *xs_xi, double *us, tolg_history *h) {
  dyn_cache c;
  ws_t w;
  int rc = dyn_init(p, &c);
  if (rc) return rc;
  ws_alloc(&w, p->N, p->m);
  rc = ss_fit_ws(p, o, x0_q, x0_xi, us_init, xs_q, xs_xi, us, h, &c, &w);
  ws_free(&w);
  dyn_free(&c);
  return rc;
}
static int ss_fit_ws(const tolg_problem *p, const tolg_options *o, const double *x0_q, const double *x0_xi,
                     const double *us_init, double *xs_q, double *xs_xi, double *us, tolg_history *h,
                     const dyn_cache *cp, ws_t *wp) {
#define c (*cp)
#define w (*wp)
  int N = p->N, m = p->m;
  w.mu = 1.0; w.delta = 2.0;
  memcpy(w.us, us_init, sizeof(double) * (size_t)(N * m));
  /* _init_rollout (:2015-2028) */
  memcpy(w.xq, x0_q, 16 * sizeof(double));
  memcpy(w.xxi, x0_xi, 6 * sizeof(double));
  for (int i = 0; i < N; i++)
    dyn_f(p, &c, w.xq + 16 * i, w.xxi + 6 * i, w.us + m * i, w.xq + 16 * (i + 1), w.xxi + 6 * (i + 1));
  h->n_iters = 0; h->converged = 0; h->status = 0;
  for (int it = 0; it < o->max_iter; it++) {
    int accepted = 0, ntr = 0;
    double alpha = 1.0, J_opt = 0;
    linearize(p, &c, &w, 0);
    for (int i = 0; i <= N; i++) J_opt += w.L[i];
    h->J_lin[it] = J_opt;
    double g = grad_ss(p, &w);
    h->grad_hist[it] = g;
    if (g < o->tol_grad) { h->converged = 1; break; }
    if (backward(p, &w, o->max_reg)) h->status = 1;
    for (int a = 0; a < SS_ALPHAS; a++) {
      alpha = pow(1.1, -(double)(a * a));
      rollout_ss(p, &c, &w, alpha, o->rollout_linear);
      double J_new = traj_cost(p, &c, w.nq, w.nxi, w.nus);
      h->trial_J[20 * it + ntr++] = J_new;
      if (J_new < J_opt) {
        J_opt = J_new;
        memcpy(w.xq, w.nq, sizeof(double) * (size_t)(16 * (N + 1)));
        memcpy(w.xxi, w.nxi, sizeof(double) * (size_t)(6 * (N + 1)));
        memcpy(w.us, w.nus, sizeof(double) * (size_t)(N * m));
        accepted = 1;
        break;
      }
    }
    h->n_trials[it] = ntr;
    h->J_hist[it] = J_opt; h->alpha_hist[it] = alpha; h->mu_hist[it] = w.mu;
    h->n_iters = it + 1;
    if (!accepted) { h->status = 2; break; }
  }
  memcpy(xs_q, w.xq, sizeof(double) * (size_t)(16 * (N + 1)));
  memcpy(xs_xi, w.xxi, sizeof(double) * (size_t)(6 * (N + 1)));
  memcpy(us, w.us, sizeof(double) * (size_t)(N * m));
  return 0;
#undef c
#undef w
}

/* Batch driver: B independent fits (the reference's joblib fan-out,
 * visualization/perturb_all_compute.py:240-250), trajectories dealt to `threads` OpenMP threads (<= 0: the
 * OpenMP default); every thread owns one dynamics cache, workspace and scratch history for all its
 * trajectories.  Histories are [B][max_iter(+1)] row-major.  Returns the number of threads the parallel
 * region actually ran with (< 0: error). */
int tolg_oracle_fit_batch(int mode_ms, const tolg_problem *p, const tolg_options *o, int B,
                          const double *x0_q, const double *x0_xi, const double *us_init,
                          double *xs_q, double *xs_xi, double *us, double *J_hist, double *grad_hist,
                          double *defect_hist, int *iters, int *status, int *converged, int threads,
                          double *mu_hist /* [B][max_iter] regularisation after each backward pass, or NULL */) {
  int N = p->N, m = p->m, K = o->max_iter, used = 1, err = 0;
  if (threads <= 0) threads = omp_get_max_threads();
#pragma omp parallel num_threads(threads)
  {
    dyn_cache c;
    ws_t w;
    tolg_history h;
    int ok = dyn_init(p, &c) == 0;
    if (ok) ws_alloc(&w, N, m);
    h.alpha_hist = malloc(sizeof(double) * (size_t)(K + 1));
    h.mu_hist = malloc(sizeof(double) * (size_t)(K + 1));
    h.J_lin = malloc(sizeof(double) * (size_t)(K + 1));
    h.trial_J = malloc(sizeof(double) * (size_t)(K + 1) * 20);
    h.n_trials = malloc(sizeof(int) * (size_t)(K + 1));
#pragma omp single
    used = omp_get_num_threads();
    if (!ok) {
#pragma omp atomic write
      err = 1;
    }
#pragma omp for schedule(dynamic, 1)
    for (int b = 0; b < B; b++) {
      if (!ok) continue;
      h.J_hist = J_hist + (size_t)b * K;
      h.grad_hist = grad_hist + (size_t)b * (K + 1);
      h.defect_hist = defect_hist + (size_t)b * (K + 1);
      if (mode_ms)
        ms_fit_ws(p, o, x0_q + 16 * (size_t)b, x0_xi + 6 * (size_t)b, us_init + (size_t)b * N * m,
                  xs_q + (size_t)b * 16 * (N + 1), xs_xi + (size_t)b * 6 * (N + 1), us + (size_t)b * N * m, &h, &c, &w);
      else
        ss_fit_ws(p, o, x0_q + 16 * (size_t)b, x0_xi + 6 * (size_t)b, us_init + (size_t)b * N * m,
                  xs_q + (size_t)b * 16 * (N + 1), xs_xi + (size_t)b * 6 * (N + 1), us + (size_t)b * N * m, &h, &c, &w);
      iters[b] = h.n_iters; status[b] = h.status; converged[b] = h.converged;
      if (mu_hist)
        for (int k = 0; k < K; k++) mu_hist[(size_t)b * K + k] = (k < h.n_iters) ? h.mu_hist[k] : NAN;
    }
    free(h.alpha_hist); free(h.mu_hist); free(h.J_lin); free(h.trial_J); free(h.n_trials);
    if (ok) { ws_free(&w); dyn_free(&c); }
  }
  return err ? -1 : used;
}

/* ---- element-level exports for unit tests -------------------------------------------------- */
void tolg_oracle_se3_exp(const double tau[6], double M[16]) { se3_t X; se3_exp(tau, &X); se3_to_matrix(&X, M); }
void tolg_oracle_se3_log(const double M[16], double tau[6]) { se3_t X; se3_from_matrix(M, &X); se3_log(&X, tau); }
void tolg_oracle_se3_ljac(const double tau[6], double J[36]) { se3_ljac(tau, J); }
void tolg_oracle_se3_rjac(const double tau[6], double J[36]) { se3_rjac(tau, J); }
void tolg_oracle_se3_rjacinv(const double tau[6], double J[36]) { se3_rjacinv(tau, J); }
void tolg_oracle_se3_adj(const double M[16], double Ad[36]) { se3_t X; se3_from_matrix(M, &X); se3_adj(&X, Ad); }
void tolg_oracle_project(const double M[16], double O[16]) { se3_t X; se3_from_matrix(M, &X); se3_to_matrix(&X, O); }
void tolg_oracle_lminus(const double A[16], const double B[16], double e[6], double J[36]) {
  se3_t X, Y; se3_from_matrix(A, &X); se3_from_matrix(B, &Y); se3_lminus(&X, &Y, e, J);
}
void tolg_oracle_rminus(const double A[16], const double B[16], double e[6]) {
  se3_t X, Y; se3_from_matrix(A, &X); se3_from_matrix(B, &Y); se3_rminus(&X, &Y, e);
}
int tolg_oracle_f(const tolg_problem *p, const double q[16], const double xi[6], const double *u,
                  double qn[16], double xin[6]) {
  dyn_cache c; int rc = dyn_init(p, &c); if (rc) return rc;
  dyn_f(p, &c, q, xi, u, qn, xin); dyn_free(&c); return 0;
}
int tolg_oracle_fx_fu(const tolg_problem *p, const double q[16], const double xi[6], const double *u, double Fx[144],
                      double *Fu) {
  dyn_cache c; int rc = dyn_init(p, &c); if (rc) return rc;
  dyn_fx(p, &c, q, xi, u, Fx); dyn_fu(p, &c, q, Fu); dyn_free(&c); return 0;
}
int tolg_oracle_cost(const tolg_problem *p, const double q[16], const double xi[6], const double *u, int i,
                     int terminal, double *l, double lx[12], double lxx[144], double *lu, double *luu) {
  dyn_cache c; int rc = dyn_init(p, &c); if (rc) return rc;
  *l = cost_l(p, &c, q, xi, u, i, terminal);
  cost_lx_lxx(p, &c, q, xi, i, terminal, lx, lxx);
  if (!terminal) cost_lu_luu(p, u, i, lu, luu);
  dyn_free(&c); return 0;
}
/* one full linearisation + backward pass on a given trajectory: the per-knot quantities the HIP
 * kernels are checked against (d, F_x, l_x, l_xx, k, K, V_x, V_xx at knot 0, gradient norm) */
int tolg_oracle_lin_backward(const tolg_problem *p, int ms, double mu_in, double delta_in, double max_reg,
                             const double *xs_q, const double *xs_xi, const double *us, double *d,
                             double *Fx, double *Lx, double *Lxx, double *k, double *K, double *Vx0,
                             double *Vxx0, double *J, double *grad, double *mu_out, double *delta_out) {
  dyn_cache c; ws_t w; int N = p->N, m = p->m, rc = dyn_init(p, &c); if (rc) return rc;
  ws_alloc(&w, N, m);
  memcpy(w.xq, xs_q, sizeof(double) * (size_t)(16 * (N + 1)));
  memcpy(w.xxi, xs_xi, sizeof(double) * (size_t)(6 * (N + 1)));
  memcpy(w.us, us, sizeof(double) * (size_t)(N * m));
  w.mu = mu_in; w.delta = delta_in;
  linearize(p, &c, &w, ms);
  backward(p, &w, max_reg);
  double s = 0; for (int i = 0; i <= N; i++) s += w.L[i];
  *J = s; *grad = ms ? grad_ms(p, &w) : grad_ss(p, &w);
  *mu_out = w.mu; *delta_out = w.delta;
  if (d) memcpy(d, w.d, sizeof(double) * (size_t)(12 * N));
  if (Fx) memcpy(Fx, w.Fx, sizeof(double) * (size_t)(144 * N));
  if (Lx) memcpy(Lx, w.Lx, sizeof(double) * (size_t)(12 * (N + 1)));
  if (Lxx) memcpy(Lxx, w.Lxx, sizeof(double) * (size_t)(144 * (N + 1)));
  if (k) memcpy(k, w.k, sizeof(double) * (size_t)(N * m));
  if (K) memcpy(K, w.K, sizeof(double) * (size_t)(N * m * 12));
  if (Vx0) memcpy(Vx0, w.Vx, 12 * sizeof(double));
  if (Vxx0) memcpy(Vxx0, w.Vxx, 144 * sizeof(double));
  ws_free(&w); dyn_free(&c); return 0;
}
