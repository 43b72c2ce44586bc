// tolg_lie.h -- SO(3)/SE(3) device math for the gfx950 kernels (fp64, registers only).
//
// Semantics follow what the reference obtains from manifpy at its call sites
// (traoptlibrary/traopt_dynamics.py:783,823; traopt_cost.py:668,778; traopt_controller.py:2683,
// 2714,2804): unit-quaternion + translation poses, Exp/Log, rplus/lminus/rminus and their
// Jacobians in the library's [omega, v] twist order (traoptlibrary/traopt_utilis.py:43-92,387-399).
// Poses never exist as 4x4 matrices on the device: a pose is (unit quaternion xyzw, translation),
// which is also what the reference's arithmetic runs on (every 4x4 it holds is re-derived from a
// unit quaternion, traopt_utilis.py:331-354), so results agree to rounding.
#pragma once
#include <hip/hip_runtime.h>

#define TOLG_DEV __device__ __forceinline__
#define TOLG_EPS 1e-10  // manif Constants<double>::eps (small-angle switch)

namespace tolg {

struct V3 { double x, y, z; };
struct Q4 { double x, y, z, w; };
struct Pose { Q4 q; V3 t; };

TOLG_DEV V3 v3(double x, double y, double z) { V3 r; r.x = x; r.y = y; r.z = z; return r; }
TOLG_DEV V3 operator+(V3 a, V3 b) { return v3(a.x + b.x, a.y + b.y, a.z + b.z); }
TOLG_DEV V3 operator-(V3 a, V3 b) { return v3(a.x - b.x, a.y - b.y, a.z - b.z); }
TOLG_DEV V3 operator*(double s, V3 a) { return v3(s * a.x, s * a.y, s * a.z); }
TOLG_DEV V3 neg(V3 a) { return v3(-a.x, -a.y, -a.z); }
TOLG_DEV double dot(V3 a, V3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
TOLG_DEV V3 cross(V3 a, V3 b) {
  return v3(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x);
}

TOLG_DEV Q4 qmul(Q4 a, Q4 b) {
  Q4 r;
  r.x = a.w * b.x + a.x * b.w + a.y * b.z - a.z * b.y;
  r.y = a.w * b.y + a.y * b.w + a.z * b.x - a.x * b.z;
  r.z = a.w * b.z + a.z * b.w + a.x * b.y - a.y * b.x;
  r.w = a.w * b.w - a.x * b.x - a.y * b.y - a.z * b.z;
  return r;
}
TOLG_DEV Q4 qconj(Q4 a) { Q4 r; r.x = -a.x; r.y = -a.y; r.z = -a.z; r.w = a.w; return r; }
TOLG_DEV Q4 qnormalize(Q4 a) {
  double s = 1.0 / sqrt(a.x * a.x + a.y * a.y + a.z * a.z + a.w * a.w);
  Q4 r; r.x = a.x * s; r.y = a.y * s; r.z = a.z * s; r.w = a.w * s; return r;
}
// Re-normalisation of a product of unit quaternions: |q|^2 = 1 + d with |d| ~ 1e-15, so
// 1/sqrt(1+d) = 1 - d/2 + O(d^2) is exact to double precision without sqrt or division.
TOLG_DEV Q4 qrenorm(Q4 a) {
  double n2 = a.x * a.x + a.y * a.y + a.z * a.z + a.w * a.w;
  if (fabs(n2 - 1.0) > 1e-7) return qnormalize(a);
  double s = 1.5 - 0.5 * n2;
  Q4 r; r.x = a.x * s; r.y = a.y * s; r.z = a.z * s; r.w = a.w * s; return r;
}
// R(q) v for a unit quaternion
TOLG_DEV V3 qrot(Q4 q, V3 v) {
  V3 u = v3(q.x, q.y, q.z);
  V3 t = 2.0 * cross(u, v);
  return v + q.w * t + cross(u, t);
}
TOLG_DEV V3 qrot_inv(Q4 q, V3 v) { return qrot(qconj(q), v); }
// row-major rotation matrix of a unit quaternion (Eigen toRotationMatrix)
TOLG_DEV void q_to_R(Q4 q, double R[9]) {
  double tx = 2 * q.x, ty = 2 * q.y, tz = 2 * q.z;
  double twx = tx * q.w, twy = ty * q.w, twz = tz * q.w;
  double txx = tx * q.x, txy = ty * q.x, txz = tz * q.x;
  double tyy = ty * q.y, tyz = tz * q.y, tzz = tz * q.z;
  R[0] = 1 - (tyy + tzz); R[1] = txy - twz;       R[2] = txz + twy;
  R[3] = txy + twz;       R[4] = 1 - (txx + tzz); R[5] = tyz - twx;
  R[6] = txz - twy;       R[7] = tyz + twx;       R[8] = 1 - (txx + tyy);
}
// scipy Rotation.from_matrix(...).as_quat() (traopt_utilis.py:167-181): the projection the
// reference applies to every 4x4 it touches.
TOLG_DEV Q4 R_to_q(const double R[9]) {
  double tr = R[0] + R[4] + R[8];
  double qx, qy, qz, qw;
  if (tr >= R[0] && tr >= R[4] && tr >= R[8]) {
    qx = R[7] - R[5]; qy = R[2] - R[6]; qz = R[3] - R[1]; qw = 1 + tr;
  } else if (R[0] >= R[4] && R[0] >= R[8]) {
    qx = 1 - tr + 2 * R[0]; qy = R[3] + R[1]; qz = R[6] + R[2]; qw = R[7] - R[5];
  } else if (R[4] >= R[8]) {
    qy = 1 - tr + 2 * R[4]; qz = R[7] + R[5]; qx = R[1] + R[3]; qw = R[2] - R[6];
  } else {
    qz = 1 - tr + 2 * R[8]; qx = R[2] + R[6]; qy = R[5] + R[7]; qw = R[3] - R[1];
  }
  Q4 q; q.x = qx; q.y = qy; q.z = qz; q.w = qw;
  return qnormalize(q);
}

// Coefficients of V(w) = Jl(w) = I + a W + b W^2 and of the SE(3) Q block (Barfoot 7.86).
struct SO3Coef { double a, b, c1, c2, c3; };
TOLG_DEV SO3Coef so3_coef(double th2, bool want_q) {
  SO3Coef k;
  if (th2 <= TOLG_EPS) {
    k.a = 0.5; k.b = 0.0;  // manif small-angle ljac: I + W/2
    k.c1 = 1.0 / 6 - th2 / 120; k.c2 = 1.0 / 24 - th2 / 720; k.c3 = 1.0 / 120 - th2 / 2520;
  } else {
    double th = sqrt(th2), s, c;
    sincos(th, &s, &c);
    double i2 = 1.0 / th2;
    k.a = (1 - c) * i2;
    k.b = (th - s) * i2 / th;
    if (want_q) {
      k.c1 = k.b;
      k.c2 = (th2 + 2 * c - 2) * (0.5 * i2 * i2);
      k.c3 = (2 * th - 3 * s + th * c) * (0.5 * i2 * i2 / th);
    } else {
      k.c1 = k.c2 = k.c3 = 0;
    }
  }
  return k;
}
// manif SO3Tangent::exp -> unit quaternion
TOLG_DEV Q4 so3_exp(V3 w) {
  double th2 = dot(w, w);
  Q4 q;
  if (th2 > TOLG_EPS) {
    double th = sqrt(th2), s, c;
    sincos(0.5 * th, &s, &c);
    s /= th;
    q.x = s * w.x; q.y = s * w.y; q.z = s * w.z; q.w = c;
  } else {
    q.x = 0.5 * w.x; q.y = 0.5 * w.y; q.z = 0.5 * w.z; q.w = 1.0;
    q = qnormalize(q);
  }
  return q;
}
// manif SO3::log (sign-fixed, quaternion based)
TOLG_DEV V3 so3_log(Q4 q) {
  double s2 = q.x * q.x + q.y * q.y + q.z * q.z, c;
  if (s2 > TOLG_EPS) {
    double s = sqrt(s2);
    double two = 2.0 * ((q.w < 0.0) ? atan2(-s, -q.w) : atan2(s, q.w));
    c = two / s;
  } else {
    // q and -q are the same rotation.  The reference only ever takes Log of quaternions that scipy
    // just derived from a matrix (w > 0 near the identity); here quaternions are composed directly,
    // so the small-angle branch has to pick the w > 0 representative itself.
    c = (q.w < 0.0) ? -2.0 : 2.0;
  }
  return v3(c * q.x, c * q.y, c * q.z);
}
// V(w) v with precomputed coefficients
TOLG_DEV V3 ljac_apply(V3 w, SO3Coef k, V3 v) {
  V3 wv = cross(w, v);
  return v + k.a * wv + k.b * cross(w, wv);
}
// coefficient of W^2 in V(w)^-1 = I - W/2 + c W^2
TOLG_DEV double ljacinv_coef(double th2) {
  if (th2 <= TOLG_EPS) return 0.0;
  double th = sqrt(th2), s, c;
  sincos(th, &s, &c);
  return 1.0 / th2 - (1 + c) / (2 * th * s);
}
TOLG_DEV V3 ljacinv_apply(V3 w, double c, V3 v) {
  V3 wv = cross(w, v);
  return v - 0.5 * wv + c * cross(w, wv);
}

// Exp on SE(3) with ONE sincos: sin t = 2 s c, 1 - cos t = 2 s^2 for (s, c) = sincos(t/2)
TOLG_DEV Pose se3_exp(V3 w, V3 v) {
  Pose X;
  double th2 = dot(w, w);
  SO3Coef k;
  if (th2 > TOLG_EPS) {
    double th = sqrt(th2), sh, ch;
    sincos(0.5 * th, &sh, &ch);
    double ith = 1.0 / th, i2 = ith * ith;
    double so = sh * ith;
    X.q.x = so * w.x; X.q.y = so * w.y; X.q.z = so * w.z; X.q.w = ch;
    k.a = 2.0 * sh * sh * i2;
    k.b = (th - 2.0 * sh * ch) * i2 * ith;
  } else {
    X.q = so3_exp(w);
    k.a = 0.5; k.b = 0.0;
  }
  k.c1 = k.c2 = k.c3 = 0;
  X.t = ljac_apply(w, k, v);
  return X;
}
// Log on SE(3); the V^-1 coefficient 1/t^2 - (1+cos t)/(2 t sin t) = 1/t^2 - cot(t/2)/(2t) comes
// straight from the quaternion (cot(t/2) = |w|/|qv|): no sincos
TOLG_DEV void se3_log(Pose X, V3& w, V3& v) {
  Q4 q = X.q;
  double s2 = q.x * q.x + q.y * q.y + q.z * q.z, c, cl;
  if (s2 > TOLG_EPS) {
    double s = sqrt(s2);
    double two = 2.0 * ((q.w < 0.0) ? atan2(-s, -q.w) : atan2(s, q.w));
    double is = 1.0 / s, itw = 1.0 / two;
    c = two * is;
    cl = itw * itw - 0.5 * fabs(q.w) * is * fabs(itw);
    if (two * two <= TOLG_EPS) cl = 0.0;  // manif's small-angle V^-1 = I - W/2
  } else {
    c = (q.w < 0.0) ? -2.0 : 2.0;
    cl = 0.0;
  }
  w = v3(c * q.x, c * q.y, c * q.z);
  v = ljacinv_apply(w, cl, X.t);
}
// ------------------------------------------------------------------------------------------------
// Series fast paths for the sequential rollout (K3) and the linearisation, where Exp and Log sit on the critical
// path of a wave that has nothing else to run.  Inside their domains they need no sqrt, no sincos / atan2 and no
// division; truncation errors are below 1e-17 relative, i.e. they agree with the closed forms above to rounding.
// Shape of every function: the series runs unconditionally as straight-line code, manif's small-angle values
// (th2 <= 1e-10) are selected in, and the lanes outside the domain (a rotation of more than 1 rad per step, a
// deviation of more than 60 degrees) are redone with the closed form behind a wave-uniform branch that a
// tracking solve never takes after its first iterations.  The series have two lengths: where the argument is
// `small` (per-step rotation below 0.2 rad, deviation below 0.06 rad -- the steady state of a tracking solve) the
// terms past the short length are below 1e-17 of the sum, and a wave whose lanes are all small skips them.
// Every VALUE is decided per lane from the lane's own argument: a trajectory's result does not depend on its
// neighbours in the wave.  The two wave-uniform facts -- does any lane need the long tier / the closed form --
// come in a SeriesGate, which a caller with several evaluations in a row computes once for all of them (a
// vector-compare -> scalar-branch round trip costs ~17 cycles on a chain that has nothing to hide it behind).
// ------------------------------------------------------------------------------------------------
TOLG_DEV bool any_lane(bool p) { return __builtin_amdgcn_ballot_w64(p) != 0ull; }
struct SeriesGate { bool any_long, any_fb; };
#ifdef TOLG_TIER_COUNT
// Diagnostic build (tools/tier_share.py, never the product): how many lanes / gates (= waves at one rollout step or linearised
// knot) fall into each tier -- [0..2] lanes short / long / closed form, [3..5] gates by the most expensive tier they execute.
__device__ unsigned long long g_tier[6];
#endif
TOLG_DEV SeriesGate series_gate(bool small, bool dom) {
  SeriesGate g;
  g.any_long = any_lane(!small);
  g.any_fb = any_lane(!dom);
#ifdef TOLG_TIER_COUNT
  {
    const unsigned long long ex = __builtin_amdgcn_ballot_w64(true), bs = __builtin_amdgcn_ballot_w64(small),
                             bd = __builtin_amdgcn_ballot_w64(dom);
    const unsigned me = __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
    if (ex != 0 && me == (unsigned)__builtin_ctzll(ex)) {  // the lowest active lane reports for the wave
      atomicAdd(&g_tier[0], (unsigned long long)__builtin_popcountll(bs & bd));
      atomicAdd(&g_tier[1], (unsigned long long)__builtin_popcountll(ex & ~bs & bd));
      atomicAdd(&g_tier[2], (unsigned long long)__builtin_popcountll(ex & ~bd));
      atomicAdd(&g_tier[g.any_fb ? 5 : g.any_long ? 4 : 3], 1ull);
    }
  }
#endif
  return g;
}
// Per-lane predicates of the four argument kinds.  A gate built from log_small / log_dom of y = |q_v|^2 also
// covers the evaluations at the resulting angle th2 = (2 asin sqrt y)^2: y < 1e-3 gives th2 < 4.002e-3 (inside
// coef_small and ljinv_small), y < 1/4 gives th2 < 1.0967 (inside coef_dom and ljinv_dom).  horner2 trusts the
// gate (any_long false = every lane small for EVERY series evaluated under it), so the thresholds are tied together
// at compile time: callers share one gate between Exp and coefficient series in both directions (roll_step: an
// exp_small gate over the (th - sin th)/th^3 series; lin_knot: a coef_small gate over so3_exp_fast), and between
// Log and the series evaluated at the Log's angle.
constexpr double kExpSmall = 0.04, kExpDomHi = 1.0;        // 5 / 6 / 6 terms reach 1e-17
constexpr double kLogSmallY = 1e-3, kLogDomY = 0.25;       // 6 / 5 terms; long tier: a deviation of up to 60 degrees
constexpr double kCoefSmall = 0.04, kCoefDom = 1.21;       // 6 terms of each (long tier: 10 / 9, below 1e-17 up to 1.21)
constexpr double kLjinvSmall = 0.01, kLjinvDom = 1.21;     // 5 terms (long tier: 12)
// upper bound of (2 asin x)^2: asin x = x + x^3/6 + 3x^5/40 + 15x^7/336 + ..., coefficients decreasing, so the
// tail after the x^7 term is below (35/1152) x^9 / (1 - x^2)
constexpr double angle2_ub(double x) {
  const double x2 = x * x;
  const double a = x * (1.0 + x2 * (1.0 / 6 + x2 * (3.0 / 40 + x2 * (15.0 / 336)))) + (35.0 / 1152) * x2 * x2 * x2 * x2 * x / (1.0 - x2);
  return 4.0 * a * a;
}
constexpr double kLogSmallX = 0.0316228, kLogDomX = 0.5;   // >= sqrt of the y thresholds
static_assert(kLogSmallX * kLogSmallX >= kLogSmallY && kLogDomX * kLogDomX >= kLogDomY, "x bounds of the Log thresholds");
static_assert(kExpSmall == kCoefSmall, "one gate serves Exp and coefficient series in both directions");
static_assert(kExpDomHi <= kCoefDom, "an exp_dom gate covers so3_coef_fast at the same angle");
static_assert(angle2_ub(kLogSmallX) <= kLjinvSmall && angle2_ub(kLogSmallX) <= kCoefSmall,
              "a log_small gate covers the series evaluated at the Log's angle");
static_assert(angle2_ub(kLogDomX) <= kLjinvDom && angle2_ub(kLogDomX) <= kCoefDom,
              "a log_dom gate covers the domains of the series evaluated at the Log's angle");
TOLG_DEV bool exp_small(double th2) { return th2 < kExpSmall; }
TOLG_DEV bool exp_dom(double th2) { return th2 > TOLG_EPS && th2 < kExpDomHi; }
TOLG_DEV bool log_small(double y) { return y < kLogSmallY; }
TOLG_DEV bool log_dom(double y) { return y < kLogDomY; }
TOLG_DEV bool coef_small(double th2) { return th2 < kCoefSmall; }
TOLG_DEV bool coef_dom(double th2) { return th2 < kCoefDom; }
TOLG_DEV bool ljinv_small(double th2) { return th2 < kLjinvSmall; }
TOLG_DEV bool ljinv_dom(double th2) { return th2 < kLjinvDom; }
// r y + c as one three-address instruction.  Left to itself the compiler turns a Horner step whose coefficient
// lives in a register across the knot loop into v_mov_b64 + v_fmac_f64 (the two-address form clobbers its addend),
// i.e. two issue slots of a wave that is issue-bound at one fp64 instruction per ~5 cycles.
TOLG_DEV double horner_step(double r, double y, double c) {
  double d;
  asm("v_fma_f64 %0, %1, %2, %3" : "=v"(d) : "v"(r), "v"(y), "v"(c));
  return d;
}
// sum_k c[k] y^k, Horner
template <int N>
TOLG_DEV double horner(const double (&c)[N], double y) {
  double r = c[N - 1];
#pragma unroll
  for (int k = N - 2; k >= 0; k--) r = horner_step(r, y, c[k]);
  return r;
}
// Two-tier Horner: lanes with `small` take the low NS coefficients only.  any_long (wave-uniform) false promises
// that every lane is small.  The empty volatile asm keeps the compiler from speculating the upper part back into
// the straight-line code.
template <int NS, int N>
TOLG_DEV double horner2(const double (&c)[N], double y, bool small, bool any_long) {
  static_assert(NS >= 1 && NS <= N, "tier");
  double r = c[NS - 1];
  if (__builtin_expect(any_long, 0)) {
    asm volatile("");
    double rl = c[N - 1];
#pragma unroll
    for (int k = N - 2; k >= NS - 1; k--) rl = horner_step(rl, y, c[k]);
    r = small ? r : rl;
  }
#pragma unroll
  for (int k = NS - 2; k >= 0; k--) r = horner_step(r, y, c[k]);
  return r;
}
// half-angle series shared by the two Exp: sin(x)/x and cos(x) in y = x^2 = th2/4
TOLG_DEV void half_angle_series(double th2, bool sm, bool any_long, double& so, double& cw) {
  const double S[9] = {1.0, -1.0 / 6, 1.0 / 120, -1.0 / 5040, 1.0 / 362880, -1.0 / 39916800, 1.0 / 6227020800.0,
                       -1.0 / 1307674368000.0, 1.0 / 355687428096000.0};
  const double Cc[9] = {1.0, -1.0 / 2, 1.0 / 24, -1.0 / 720, 1.0 / 40320, -1.0 / 3628800, 1.0 / 479001600,
                        -1.0 / 87178291200.0, 1.0 / 20922789888000.0};
  const double y = 0.25 * th2;
  so = 0.5 * horner2<5>(S, y, sm, any_long);   // sin(th/2)/th
  cw = horner2<6>(Cc, y, sm, any_long);        // cos(th/2)
}
TOLG_DEV Pose se3_exp_fast(V3 w, V3 v, SeriesGate g) {
  const double th2 = dot(w, w);
  const bool sm = exp_small(th2);
  const double Bc[10] = {1.0 / 6, -1.0 / 120, 1.0 / 5040, -1.0 / 362880, 1.0 / 39916800, -1.0 / 6227020800.0,
                         1.0 / 1307674368000.0, -1.0 / 355687428096000.0, 1.0 / 121645100408832000.0,
                         -1.0 / 51090942171709440000.0};
  double so, cw;
  half_angle_series(th2, sm, g.any_long, so, cw);
  Pose X;
  X.q.x = so * w.x; X.q.y = so * w.y; X.q.z = so * w.z; X.q.w = cw;
  SO3Coef k;
  k.a = 2.0 * so * so;                          // (1 - cos th)/th^2 = 2 sin^2(th/2)/th^2
  k.b = horner2<6>(Bc, th2, sm, g.any_long);    // (th - sin th)/th^3
  k.c1 = k.c2 = k.c3 = 0;
  X.t = ljac_apply(w, k, v);
  if (__builtin_expect(g.any_fb, 0)) {
    asm volatile("");
    if (!exp_dom(th2)) X = se3_exp(w, v);
  }
  return X;
}
TOLG_DEV Pose se3_exp_fast(V3 w, V3 v) {
  const double th2 = dot(w, w);
  return se3_exp_fast(w, v, series_gate(exp_small(th2), exp_dom(th2)));
}
TOLG_DEV double quat_vec2(Q4 q) { return q.x * q.x + q.y * q.y + q.z * q.z; }  // sin^2 of the half angle
TOLG_DEV void se3_log_fast(Pose X, V3& w, V3& v, SeriesGate g) {
  const Q4 q = X.q;
  const double y = quat_vec2(q);
  const bool tiny = !(y > TOLG_EPS), sm = log_small(y);
  // asin(sqrt y)/sqrt y: the first six Taylor coefficients (the short tier: y < 1e-3), then y^6 times the degree-10
  // interpolant of the remainder at the Chebyshev nodes of [0, 1/4] -- 5.4e-19 relative on the whole interval with 17
  // coefficients, where the 14 Taylor terms of rounds 2-3 reached 1e-17 on y < 1/16 only: the singularity at y = 1 is
  // as far away either way, but the interpolant spends its degrees of freedom on the interval, not on a disc
  // (tools/gen_log_series.py, 60-digit arithmetic).  A deviation of up to 60 degrees (29 before) stays on the series:
  // the rollouts and linearisations of a solve's first iterations, where the closed forms cost a stage rollout half
  // its time again (0.41 against 0.28 ms).
  // 1/t^2 - cot(t/2)/(2t) = sum |B_{2k+2}|/(2k+2)! t^2k: 12 terms reach 4e-19 on t^2 <= 1.21.
  const double A[17] = {1.0, 0.16666666666666667, 0.075, 0.044642857142857143, 0.030381944444444444,
                        0.022372159090909091, 0.017352764423078705, 0.013964843748277154, 0.011551801170464363,
                        0.009761592565719998, 0.0083908710662676442, 0.0073027366952445512, 0.0065578263896588765,
                        0.0049481126918453019, 0.0087132163323646033, -0.0048408125598408092, 0.01700421839934941};
  const double L[12] = {0.083333333333333333, 0.0013888888888888889, 3.3068783068783069e-5, 8.2671957671957672e-7,
                        2.0876756987868099e-8, 5.2841901386874932e-10, 1.3382536530684679e-11, 3.3896802963225829e-13,
                        8.5860620562778446e-15, 2.1748686985580619e-16, 5.5090028283602295e-18, 1.3954464685812523e-19};
  double c = 2.0 * horner2<6>(A, y, sm, g.any_long);  // angle / |q_v|
  const double t2 = c * c * y;                        // angle^2 (< 4.1e-3 when sm)
  double cl = horner2<5>(L, t2, sm, g.any_long);
  if (tiny) c = 2.0;                      // manif's small-angle Log: 2 q_v
  if (tiny || t2 <= TOLG_EPS) cl = 0.0;   // and its small-angle V^-1 = I - W/2
  if (q.w < 0.0) c = -c;                  // q and -q are the same rotation: take the w > 0 representative
  w = v3(c * q.x, c * q.y, c * q.z);
  v = ljacinv_apply(w, cl, X.t);
  if (__builtin_expect(g.any_fb, 0)) {
    asm volatile("");
    if (!log_dom(y)) se3_log(X, w, v);
  }
}
TOLG_DEV void se3_log_fast(Pose X, V3& w, V3& v) {
  const double y = quat_vec2(X.q);
  se3_log_fast(X, w, v, series_gate(log_small(y), log_dom(y)));
}
// Coefficients of V(w) and of the SE(3) Q block as series in th^2 (th^2 < 1.21: one time step of rotation, or a
// tracking error below one radian; otherwise the closed forms).  (1 - cos t)/t^2 = sum (-1)^k t^2k/(2k+2)!,
// (t - sin t)/t^3 = sum (-1)^k t^2k/(2k+3)!, (t^2 + 2 cos t - 2)/(2 t^4) = sum (-1)^k t^2k/(2k+4)!,
// (2t - 3 sin t + t cos t)/(2 t^5) = sum (-1)^k (k+1) t^2k/(2k+5)!.  No sqrt / sincos / division, and none of the
// cancellation the closed forms suffer for small angles.  For th^2 <= 1e-10 a, b, c1 take so3_coef's constants;
// its two-term c2 and c3 are what the series rounds to there.
TOLG_DEV SO3Coef so3_coef_fast(double th2, bool want_q, SeriesGate g) {
  const bool tiny = !(th2 > TOLG_EPS), sm = coef_small(th2);
  const double A[10] = {0.5, -0.041666666666666664, 0.001388888888888889, -2.48015873015873e-05, 2.755731922398589e-07,
                        -2.08767569878681e-09, 1.1470745597729725e-11, -4.779477332387385e-14, 1.5619206968586225e-16,
                        -4.110317623312165e-19};
  const double B[10] = {0.16666666666666666, -0.008333333333333333, 0.0001984126984126984, -2.7557319223985893e-06,
                        2.505210838544172e-08, -1.6059043836821613e-10, 7.647163731819816e-13, -2.8114572543455206e-15,
                        8.22063524662433e-18, -1.9572941063391263e-20};
  SO3Coef k;
  const double a = horner2<6>(A, th2, sm, g.any_long), b = horner2<6>(B, th2, sm, g.any_long);
  k.a = tiny ? 0.5 : a;
  k.b = tiny ? 0.0 : b;
  if (want_q) {
    const double C2[9] = {0.041666666666666664, -0.001388888888888889, 2.48015873015873e-05, -2.755731922398589e-07,
                          2.08767569878681e-09, -1.1470745597729725e-11, 4.779477332387385e-14, -1.5619206968586225e-16,
                          4.110317623312165e-19};
    const double C3[9] = {0.008333333333333333, -0.0003968253968253968, 8.267195767195768e-06, -1.0020843354176688e-07,
                          8.029521918410807e-10, -4.58829823909189e-12, 1.9680200780418645e-14, -6.576508197299464e-17,
                          1.7615646957052136e-19};
    k.c1 = b;
    k.c2 = horner2<6>(C2, th2, sm, g.any_long);
    k.c3 = horner2<6>(C3, th2, sm, g.any_long);
  } else {
    k.c1 = k.c2 = k.c3 = 0;
  }
  if (__builtin_expect(g.any_fb, 0)) {
    asm volatile("");
    if (!coef_dom(th2)) k = so3_coef(th2, want_q);
  }
  return k;
}
TOLG_DEV SO3Coef so3_coef_fast(double th2, bool want_q) {
  return so3_coef_fast(th2, want_q, series_gate(coef_small(th2), coef_dom(th2)));
}
// coefficient of W^2 in V(w)^-1 (ljacinv_coef) as the series of se3_log_fast
TOLG_DEV double ljacinv_coef_fast(double th2, SeriesGate g) {
  const double L[12] = {0.083333333333333333, 0.0013888888888888889, 3.3068783068783069e-5, 8.2671957671957672e-7,
                        2.0876756987868099e-8, 5.2841901386874932e-10, 1.3382536530684679e-11, 3.3896802963225829e-13,
                        8.5860620562778446e-15, 2.1748686985580619e-16, 5.5090028283602295e-18, 1.3954464685812523e-19};
  double r = horner2<5>(L, th2, ljinv_small(th2), g.any_long);
  if (!(th2 > TOLG_EPS)) r = 0.0;
  if (__builtin_expect(g.any_fb, 0)) {
    asm volatile("");
    if (!ljinv_dom(th2)) r = ljacinv_coef(th2);
  }
  return r;
}
TOLG_DEV double ljacinv_coef_fast(double th2) {
  return ljacinv_coef_fast(th2, series_gate(ljinv_small(th2), ljinv_dom(th2)));
}
// so3_exp with the half-angle series of se3_exp_fast
TOLG_DEV Q4 so3_exp_fast(V3 w, SeriesGate g) {
  const double th2 = dot(w, w);
  double so, cw;
  half_angle_series(th2, exp_small(th2), g.any_long, so, cw);
  Q4 q;
  q.x = so * w.x; q.y = so * w.y; q.z = so * w.z; q.w = cw;
  if (__builtin_expect(g.any_fb, 0)) {
    asm volatile("");
    if (!exp_dom(th2)) q = so3_exp(w);
  }
  return q;
}
TOLG_DEV Q4 so3_exp_fast(V3 w) {
  const double th2 = dot(w, w);
  return so3_exp_fast(w, series_gate(exp_small(th2), exp_dom(th2)));
}
TOLG_DEV Pose se3_compose(Pose A, Pose B) {
  Pose C;
  C.q = qmul(A.q, B.q);
  C.t = A.t + qrot(A.q, B.t);
  return C;
}
TOLG_DEV Pose se3_inverse(Pose A) {
  Pose B;
  B.q = qconj(A.q);
  B.t = neg(qrot(B.q, A.t));
  return B;
}
// the reference re-derives a unit quaternion from every matrix it receives: renormalise
TOLG_DEV Pose se3_project(Pose A) { A.q = qrenorm(A.q); return A; }

// ---- 3x3 helpers (row-major, fully unrolled so they stay in registers) ----------------------
TOLG_DEV void skew(V3 w, double S[9]) {
  S[0] = 0;    S[1] = -w.z; S[2] = w.y;
  S[3] = w.z;  S[4] = 0;    S[5] = -w.x;
  S[6] = -w.y; S[7] = w.x;  S[8] = 0;
}
TOLG_DEV void mul33(const double A[9], const double B[9], double C[9]) {
#pragma unroll
  for (int i = 0; i < 3; i++)
#pragma unroll
    for (int j = 0; j < 3; j++)
      C[3 * i + j] = A[3 * i] * B[j] + A[3 * i + 1] * B[3 + j] + A[3 * i + 2] * B[6 + j];
}
// Jl(w) as a matrix
TOLG_DEV void ljac33(V3 w, SO3Coef k, double J[9]) {
  double W[9], W2[9];
  skew(w, W);
  mul33(W, W, W2);
#pragma unroll
  for (int i = 0; i < 9; i++) J[i] = k.a * W[i] + k.b * W2[i];
  J[0] += 1; J[4] += 1; J[8] += 1;
}
TOLG_DEV void ljacinv33(V3 w, double c, double J[9]) {
  double W[9], W2[9];
  skew(w, W);
  mul33(W, W, W2);
#pragma unroll
  for (int i = 0; i < 9; i++) J[i] = -0.5 * W[i] + c * W2[i];
  J[0] += 1; J[4] += 1; J[8] += 1;
}
// Barfoot Q(rho, theta) = P/2 + c1 (WP + PW + WPW) + c2 (W^2 P + P W^2 - 3 WPW) + c3 (WPW^2 + W^2 PW), P = [rho]x,
// W = [theta]x, with coefficients of |theta|.  Products of skew matrices collapse to outer products:
// WP = rho theta^T - s I, PW = theta rho^T - s I, WPW = -s W, W^2 P = c theta^T - s W, P W^2 = -theta c^T - s W,
// WPW^2 = W^2 PW = -s W^2, W^2 = theta theta^T - |theta|^2 I, with s = theta . rho, c = theta x rho.  Hence
// Q = P/2 + c1 (rho theta^T + theta rho^T) + c2 (c theta^T - theta c^T) + (c2 - c1) s W - 2 c3 s theta theta^T
//     + 2 s (c3 |theta|^2 - c1) I                 (~60 multiply-adds instead of seven 3x3 products)
TOLG_DEV void Q33(V3 rho, V3 th, SO3Coef k, double Q[9]) {
  const double s = dot(th, rho), t2 = dot(th, th);
  const V3 c = cross(th, rho);
  const double r[3] = {rho.x, rho.y, rho.z}, t[3] = {th.x, th.y, th.z}, cc[3] = {c.x, c.y, c.z};
  const double dg = 2.0 * s * (k.c3 * t2 - k.c1), m3 = -2.0 * k.c3 * s, ws = (k.c2 - k.c1) * s;
#pragma unroll
  for (int i = 0; i < 3; i++)
#pragma unroll
    for (int j = 0; j < 3; j++)
      Q[3 * i + j] = k.c1 * (r[i] * t[j] + t[i] * r[j]) + k.c2 * (cc[i] * t[j] - t[i] * cc[j]) + m3 * (t[i] * t[j]) +
                     ((i == j) ? dg : 0.0);
  // + [rho/2 + ws theta]x
  const V3 a = v3(0.5 * rho.x + ws * th.x, 0.5 * rho.y + ws * th.y, 0.5 * rho.z + ws * th.z);
  Q[1] -= a.z; Q[2] += a.y; Q[3] += a.z; Q[5] -= a.x; Q[6] -= a.y; Q[7] += a.x;
}

}  // namespace tolg
