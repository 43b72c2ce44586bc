// tolg_kernels.hip -- gfx950 kernels + C ABI (include/tolg.h) for batched tracking-iLQR on SE(3).
//
// Hot path of chenghuailin/trajectory_optimization_matrix_lie_groups rebuilt for MI355X:
//   K1 k_linearize : one thread per (trajectory, knot): dynamics f, MS defect, F_x, cost
//                    gradient/Gauss-Newton Hessian   (traopt_controller.py:2823-2910,
//                    traopt_dynamics.py:763-850,1373-1482, traopt_cost.py:675-867)
//   K2 k_backward  : Riccati sweep, 16 lanes per trajectory (4 trajectories per wavefront), one
//                    12x12 column per lane held in VGPRs, lane-to-lane operand broadcast with
//                    DPP row_newbcast fused into v_fmac_f64 -- no LDS traffic for the matrix
//                    products (traopt_controller.py:2912-3093)
//   K3 k_rollout   : closed-loop nonlinear/linear rollout (traopt_controller.py:2641-2740,
//                    2030-2082)
// Device layout: structure-of-arrays with the batch index fastest, poses as unit quaternion +
// translation (see tolg_lie.h).  The 4x4 layout of the reference exists only at the C ABI.
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <type_traits>
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <new>
#include <vector>

#include "../../include/tolg.h"
#include "tolg_lie.h"

namespace tolg {

// ------------------------------------------------------------------------------------------------
// constants shared by the whole batch (device memory, read through the scalar cache)
// ------------------------------------------------------------------------------------------------
struct Consts {
  int kind, m, N, diagJ;  // diagJ: Ib and Jv are diagonal (every reference script): 12 constants instead of 36
  double dt, mass, grav, pend_k;  // pend_k = m l / 2 of Pendulum3dDyanmics (0 otherwise)
  double J[36], Jinv[36], Ib[9];
  // J = blkdiag(Ib, Jv) (checked in tolg_create): the 3x3 blocks and their inverses
  double Jv[9], Ibinv[9], Jvinv[9];
  double W1[36], W2[36], P1[36], P2[36];  // Q / P diagonal 6x6 blocks
  double R[36];                           // m x m
  // F_u = [0; B2] dt with B2 = J^-1 Pu = blkdiag(Bt (3x3), Bb (3 x (m-3))), already times dt:
  // Bt = Ib^-1 dt; Bb = Jv^-1 dt (m = 6) or its third column (drone: u3 is the body-z force)
  double Bt[9], Bb[9];
  // gravity block of F_x is linear in rte = R^T (0,0,-1): A21 = sum_a rte_a Llin[a] (times dt, 6x6
  // row-major each, no m*g: App. C-Q2)
  double Llin[3][36];
};

// Device-side view of the constants used by the three hot kernels: address space 4 (constant), so
// that uniform reads become scalar loads (lgkmcnt) and never queue behind in-flight record
// prefetches on vmcnt.  Cold kernels keep the generic pointer: with a few hundred invariant scalar
// loads hoisted out of their knot loops they would only spill SGPRs.
// F_u's constants (Bt / Bb: fu_entry, dynk_load) are read through the GENERIC pointer everywhere, once, ahead of
// the knot loops, and pinned in vector registers / LDS: reading them through this view inside the knot loop of
// k_rollout_eval (the round-2 line-search kernel, cost and defect on the rollout chain; retired in round 3 -- the
// stages now run k_rollout_ls / k_rollout_eval_t on the generic pointer) gave wrong costs and memory-aperture faults
// (profiles/r02_as4_cold_kernel_bisect.md; the ISA of the failing build was re-read in round 3 -- scalar tuple
// s[8:23] loaded ahead of the loop, copied to AGPRs by v_accvgpr_write inside it, nothing found that overwrites it --
// and the cause is still open), so no kernel depends on that construct.
#if defined(__HIP_DEVICE_COMPILE__)
typedef __attribute__((address_space(4))) Consts DConsts;
#else
typedef Consts DConsts;  // the host pass only parses the kernels
#endif

// Cache policy of the streams that are written once by one launch and read once by the next (VERDICT r3 item 1a;
// MI355X_MICROARCH.md "nt-weights").  Values: 0 default, 1 nt, 2 sc1, 3 sc0 sc1, 7 sc1 nt (loads also 4..6).  Measured on one box
// each, three rounds, through tools/ab_libs.sh (profiles/r04_nt_policy_ab_{1,2,3}_*.txt; fused launch / backward sweep in ms, base
// 0.272 / 0.322):
//   TOLG_NT_RING   the fused launch's input ring (gains, controls, nominal states; LDS-DMA)   nt: 0.256            -> 1
//   TOLG_NT_REC    the backward sweep's record ring (LDS-DMA)            nt: sweep 0.320 but the NEXT fused launch 0.305  -> 0
//   TOLG_NT_GKST   the backward sweep's gain stores                      nt: fused launch 0.261 (the reader gains)        -> 1
//   TOLG_NT_RECST  lin_knot's record stores     nt: 0.265 alone, sc1 0.280 alone; beside RING = GKST = 1: nt 0.251, sc1 0.241  -> 2
//   TOLG_NT_CURST  the fused launch's commit of the new trajectory       nt: 0.271 alone; with the four above 0.239       -> 1
//   TOLG_NT_EC     k_expected_change_ring's record and gain streams (merit search)                                         -> 0
// The effects do not add (sc1 records alone lose, beside nt gains they win): the table is of combinations, not of knobs.
// Together: fused launch 0.272 -> 0.239 ms, sweep 0.322 -> 0.327, 1 680 -> 1 775 batch-iterations/s.
#ifndef TOLG_NT_RING
#define TOLG_NT_RING 1
#endif
#ifndef TOLG_NT_REC
#define TOLG_NT_REC 0
#endif
#ifndef TOLG_NT_EC
#define TOLG_NT_EC 0
#endif
#ifndef TOLG_NT_RECST
#define TOLG_NT_RECST 2
#endif
#ifndef TOLG_NT_GKST
#define TOLG_NT_GKST 1
#endif
#ifndef TOLG_NT_RECST_K1   // the record stores of k_linearize (line-search modes, split schedule), tuned on their own
#define TOLG_NT_RECST_K1 0
#endif
#ifndef TOLG_NT_CURST
#define TOLG_NT_CURST 1
#endif
#define TOLG_POL_0 ""
#define TOLG_POL_1 " nt"
#define TOLG_POL_2 " sc1"
#define TOLG_POL_3 " sc0 sc1"
#define TOLG_POL_4 " sc0 sc1 nt"
#define TOLG_POL_5 " sc0"
#define TOLG_POL_6 " sc0 nt"
#define TOLG_POL_7 " sc1 nt"
// the same policies as the aux operand of the buffer builtins (gfx940 encoding: sc0 = 1, nt = 2, sc1 = 16)
#define TOLG_AUX(x) ((x) == 1 ? 2 : (x) == 2 ? 16 : (x) == 3 ? 17 : (x) == 4 ? 19 : (x) == 5 ? 1 : (x) == 6 ? 3 : (x) == 7 ? 18 : 0)
#define TOLG_POL_CAT(x) TOLG_POL_##x
#define TOLG_POL(x) TOLG_POL_CAT(x)

struct Params {
  const Consts* c;
  int B, Bp, N, m;
  double* cur;         // [13][N+1][Bp]  qx qy qz qw tx ty tz | xi(6)
  double* cur_u;       // [m][N][Bp]
  double* cand;
  double* cand_u;
  const double* ref;   // [N+1][13] reference pose (quat, pos) and twist, batch-shared
  double* REC;         // [N+1][recF][Bp] compact knot records written by K1 (fields: REC_*)
  int recF, fLUU;      // fields per record for this model / solve (rec_fields()), field index of REC_LUU (AL only)
  int fA22, pad2;      // field index of the stored I + H dt block, -1 when the backward sweep rebuilds it from REC_XI
  double* SC;          // [N+1][Bp] stage costs
  double* SD;          // [N][Bp]   squared defects
  double* GK;          // [N][m*13][Bp]     K (cols 0..11) | k (col 12), row-major per knot
  double *mu, *delta, *Jc, *dn, *grad;
  int *active, *iters, *status, *conv;
  double *J_hist, *grad_hist, *defect_hist, *alpha_hist, *mu_hist;
  int max_iter, pad;
  double tol_grad, tol_defect, max_reg;
  // line search (SS backtracking, MS merit search): speculative candidates live in NSLOT slot buffers
  double* slot_x;      // [NSLOT][N+1][13][Bp]
  double* slot_u;      // [NSLOT][N][m][Bp]
  double* Jtrial;      // [Bp][20] cost of the rollout with alpha_k
  double* dtrial;      // [Bp][20] its defect norm (MS)
  double* ecc;         // [Bp][2]  expected cost change of the linear alpha=1 rollout (MS)
  double* dweight;     // [Bp][2]  current / previous defect weight (MS)
  double* ls_alpha;    // [Bp]     last alpha tried this iteration
  int* ls_accept;      // [Bp]     accepted alpha index or -1
  int* ls_slot;        // [Bp]     slot holding the accepted candidate in this stage or -1
  int* k2_redo;        // [Bp/4]   groups of four the fast backward sweep handed to the full one (tolg_backward3.h, FAST)
  int* k2_hint;        // [Bp/4]   ... and whether a group's last sweep needed the general path (then the fast attempt is skipped)
  int* ec_redo;        // [Bp]     trajectories k_expected_change_ring hands to k_expected_change (tolg_expected_change.h)
  double* ED;          // [N+1][Bp][32] rollout = 'linear': deviation e_i (0..11) and control step du_i (16..16+m) of the alpha = 1
                       //          linear rollout, written by k_expected_change_ring<.., STORE>
  int affine, pad3;    // this solve takes its linear-rollout candidates from ED (every trajectory with ec_redo == 0)
  // line search, round 3 form: the stages roll out only (quad rollouts over a compacted list of the undecided
  // trajectories), costs and defects of the candidates are evaluated in parallel over the knots
  int* ls_list;        // [2][Bp]  undecided trajectories after a stage (two lists: a select builds the next while ...)
  int* ls_count;       // [2]      ... the stage's kernels still read the current one
  int* ls_pos;         // [2][Bp]  position of a trajectory on its list: the quad form keeps a stage's candidates densely, by position
  double* LSC;         // [NSLOT][N+1][Bp] stage costs of the candidates
  double* LSD;         // [NSLOT][N][Bp]   squared defects of the candidates (MS)
  // augmented-Lagrangian box input constraint (ALConstrainedCost + InputConstraint), caller-owned
  const double* al_lb;      // [m] or null (= AL off)
  const double* al_ub;      // [m]
  const double* al_lambda;  // [B][N][2m]
  const double* al_imu;     // [B][N][2m] diagonal of I_mu
};
enum { NSLOT = 15, NALPHA_MS = 20, NALPHA_SS = 13 };  // slots: the widest stage (the merit search's last: step sizes 5 .. 19; the first alpha goes straight to the candidate arrays)

// every array is knot-major [knot][field][Bp]: one knot of one field is a contiguous run over the batch
#define SIDX(c, i, b) ((((size_t)(i)) * 13 + (size_t)(c)) * (size_t)P.Bp + (size_t)(b))
#define UIDX(c, i, b) ((((size_t)(i)) * (size_t)P.m + (size_t)(c)) * (size_t)P.Bp + (size_t)(b))

// Raw buffer access for the sequential kernels: the descriptor (SRD) addresses one knot, the lane
// supplies a 32-bit byte offset (its trajectory), the field offset is a wave-uniform SGPR -- no
// 64-bit per-lane address arithmetic in the sweep loops, and out-of-range reads return 0.
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
TOLG_DEV __amdgpu_buffer_rsrc_t mkbuf(const double* p, unsigned bytes) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<double*>(p), 0, bytes, 0x00020000);
}
TOLG_DEV double bld(__amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff) {
  return __builtin_bit_cast(double, __builtin_amdgcn_raw_buffer_load_b64(r, voff, soff, 0));
}
TOLG_DEV void bst(__amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff, double x) {
  __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2, x), r, voff, soff, 0);
}
// two neighbouring doubles with one 16-byte access (offsets must be 16-byte aligned)
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef double f64x2 __attribute__((ext_vector_type(2)));
TOLG_DEV void bld2(__amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff, double& x0, double& x1) {
  f64x2 v = __builtin_bit_cast(f64x2, __builtin_amdgcn_raw_buffer_load_b128(r, voff, soff, 0));
  x0 = v.x; x1 = v.y;
}
TOLG_DEV void bst2(__amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff, double x0, double x1) {
  f64x2 v = {x0, x1};
  __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), r, voff, soff, 0);
}
// a plain or non-temporal global store (the A/B knobs above)
template <int POL, class T>
TOLG_DEV void gst(T* p, T v) {
  if constexpr (POL == 0) *p = v;
  else if constexpr (POL == 1) __builtin_nontemporal_store(v, p);
  else {
    // the other policies have no builtin for a global store: inline asm.  The s_nop 1: a store of more than 8 bytes reads its
    // data late, a VALU write of those registers needs two wait states behind it, and the compiler does not look in here.
    static_assert(POL == 2 || POL == 3 || POL == 7, "store policies: 0 default, 1 nt, 2 sc1, 3 sc0 sc1, 7 sc1 nt");
    if constexpr (sizeof(T) == 16) {
      if constexpr (POL == 2) asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" :: "v"(p), "v"(v) : "memory");
      else if constexpr (POL == 3) asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1\n\ts_nop 1" :: "v"(p), "v"(v) : "memory");
      else asm volatile("global_store_dwordx4 %0, %1, off sc1 nt\n\ts_nop 1" :: "v"(p), "v"(v) : "memory");
    } else {
      if constexpr (POL == 2) asm volatile("global_store_dwordx2 %0, %1, off sc1" :: "v"(p), "v"(v) : "memory");
      else if constexpr (POL == 3) asm volatile("global_store_dwordx2 %0, %1, off sc0 sc1" :: "v"(p), "v"(v) : "memory");
      else asm volatile("global_store_dwordx2 %0, %1, off sc1 nt" :: "v"(p), "v"(v) : "memory");
    }
  }
}
TOLG_DEV void bst2_gk(__amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff, double x0, double x1) {  // K2's gain stores
  f64x2 v = {x0, x1};
  __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), r, voff, soff, TOLG_AUX(TOLG_NT_GKST));
}
// Knot record: everything K2 (and the line-search rollouts) need from the linearisation.  3x3 blocks are
// stored column-major (column c at +3c).  The record holds what every model needs (REC_BASE + m fields),
// then the fields only some models / solves have: the gravity direction (models with a gravity block),
// the augmented-Lagrangian l_uu diagonal (AL solves); Params::recF is the size for the solve at hand.
// What is NOT here: the alpha = 1 rollout factors M_i = x_{i+1} Exp(d_q) f_q(x_i,u_i)^-1 and
// c_i = xi_{i+1} - f_xi(x_i,u_i) + d_xi of traopt_controller.py:2713-2716.  With d = Log(x_{i+1}^-1 f_q)
// they are the identity and zero up to rounding (Exp(Log(X)) = X), so the alpha = 1 step is
// x^_{i+1} = f(x^_i, u^_i); only the alpha < 1 line-search rollouts build the factors (from REC_D).
enum {
  REC_RI = 0,    // R(Exp(xi dt))^T                     -> F_x[0:3,0:3] = F_x[3:6,3:6]
  REC_TRI = 9,   // [t_inv]x R^T                        -> F_x[3:6,0:3]
  REC_JR = 18,   // dt Jr(w dt)                         -> F_x[0:3,6:9] = F_x[3:6,9:12]
  REC_QR = 27,   // dt Q(-v dt,-w dt)                   -> F_x[3:6,6:9]
  REC_D = 36,    // defect (12)
  REC_LXX = 48,  // l_xx pose block, symmetric packed (21; 69 is padding: pairs below)
  REC_LX = 70,   // l_x (12)
  REC_XI = 82,   // the twist of the knot (6, stored w0 v0 w1 v1 w2 v2): what F_x[6:12,6:12] = I + H dt is a function of
  REC_LU = 88,   // l_u = 2 R u (+ augmented-Lagrangian term) (m)
  REC_BASE = 88,
  // REC_LU + m: R^T (0,0,-1) (gravity direction in the body frame, 3 + 1 padding), gravity models only
  // Params::fLUU: diagonal added to l_uu = 2 R by the augmented Lagrangian (m), AL solves only
  // Params::fA22: I + H dt itself (6x6, column-major, 36 fields), only for the models whose backward sweep reads it
  // from the record (dense inertia, pendulum: k_backward).  With diagonal inertia blocks -- every reference script --
  // a column of I + H dt is two entries of the form alpha w_k + beta v_k per 3x3 block, which k_backward3 rebuilds
  // from the twist: 30 of the 112 fields a fused launch wrote per knot in round 2 (the launch is paced by its memory
  // traffic: -0.04 ms measured then), and a knot of records of four trajectories fits three 1 KB LDS-DMA
  // instructions of the backward sweep instead of four.
  REC_FMAX = REC_BASE + 6 + 4 + 6 + 36,
  // Pendulum3dDyanmics only: F_u[6:9,0:3] = J^-1 skew(m rho) R^T dt (state dependent), row-major, in the
  // slot of REC_TRI -- that block of F_x is identically zero without a translation, and an extra
  // field would leave a never-written hole in every record group of all the other models
  REC_BU = REC_TRI
};
__host__ __device__ inline int rec_rte(int m) { return REC_LU + m; }
__host__ __device__ constexpr int rec_fields(int m, bool grav, bool al, bool a22) {
  return REC_BASE + m + (grav ? 4 : 0) + (al ? m : 0) + (a22 ? 36 : 0);
}
// Records and gains are interleaved by four trajectories and by field pairs:
// REC [knot][b / 4][field / 2][b % 4][field % 2], GK [knot][b / 4][column j][row u / 2][b % 4][u % 2].
// The four trajectories of one K2 wavefront then own one contiguous run per knot (every 64-byte
// sector it touches is entirely its own; with a plain [knot][field][Bp] layout K2 fetched every
// 128-byte line four times), while a thread that owns one trajectory moves two neighbouring fields with
// one 16-byte access: K1 stores, K2's A22 / l_x / l_u loads and gain stores, K3's gain-row loads.
// Runs that are read or written as pairs start on even fields (REC_D, REC_LX, REC_A22, REC_LU); recF is even.
// Store-pattern microbenchmark: profiles/r01_store_microbench.txt.
#define RIDX(i, f, b) \
  (((size_t)(i)) * (size_t)P.recF * (size_t)P.Bp + (((size_t)(b)) >> 2) * ((size_t)P.recF * 4) + (((size_t)(f)) >> 1) * 8 + (((size_t)(b)) & 3) * 2 + (((size_t)(f)) & 1))
#define GKIDX(i, u, b, j) \
  (((size_t)(i)) * 13 * (size_t)P.m * (size_t)P.Bp + (((size_t)(b)) >> 2) * 13 * (size_t)P.m * 4 +    \
   (((size_t)(j)) * ((size_t)P.m / 2) + (((size_t)(u)) >> 1)) * 8 + (((size_t)(b)) & 3) * 2 + (((size_t)(u)) & 1))
// byte offsets for the buffer accessors: the lane's base inside one knot, then field / gain-entry offsets
#define REC_VR(b) ((unsigned)((b) >> 2) * ((unsigned)P.recF * 32u) + (unsigned)((b) & 3) * 16u)
#define GK_VG(b, M_) ((unsigned)((b) >> 2) * (13u * (M_) * 32u) + (unsigned)((b) & 3) * 16u)
#define FOFF(f) ((((unsigned)(f)) >> 1) * 64u + (((unsigned)(f)) & 1u) * 8u)
#define GOFF(u, j, M_) ((((unsigned)(j)) * ((M_) / 2u) + (((unsigned)(u)) >> 1)) * 64u + (((unsigned)(u)) & 1u) * 8u)
__host__ __device__ inline bool so3_family(int kind) { return kind == TOLG_DYN_SO3 || kind == TOLG_DYN_PENDULUM3D; }
__host__ __device__ inline int sym6(int r, int c) { return r <= c ? c * (c + 1) / 2 + r : r * (r + 1) / 2 + c; }

// ------------------------------------------------------------------------------------------------
// shared per-thread dynamics pieces
// ------------------------------------------------------------------------------------------------
struct State { Pose X; V3 w, v; };

// one knot of a state array through its descriptor: 13 fields, sB bytes apart
TOLG_DEV State load_state_b(__amdgpu_buffer_rsrc_t r, unsigned vb, unsigned sB) {
  State S;
  S.X.q.x = bld(r, vb, 0); S.X.q.y = bld(r, vb, sB); S.X.q.z = bld(r, vb, 2 * sB); S.X.q.w = bld(r, vb, 3 * sB);
  S.X.t = v3(bld(r, vb, 4 * sB), bld(r, vb, 5 * sB), bld(r, vb, 6 * sB));
  S.w = v3(bld(r, vb, 7 * sB), bld(r, vb, 8 * sB), bld(r, vb, 9 * sB));
  S.v = v3(bld(r, vb, 10 * sB), bld(r, vb, 11 * sB), bld(r, vb, 12 * sB));
  return S;
}
TOLG_DEV void store_state_b(__amdgpu_buffer_rsrc_t r, unsigned vb, unsigned sB, const State& S) {
  bst(r, vb, 0, S.X.q.x); bst(r, vb, sB, S.X.q.y); bst(r, vb, 2 * sB, S.X.q.z); bst(r, vb, 3 * sB, S.X.q.w);
  bst(r, vb, 4 * sB, S.X.t.x); bst(r, vb, 5 * sB, S.X.t.y); bst(r, vb, 6 * sB, S.X.t.z);
  bst(r, vb, 7 * sB, S.w.x); bst(r, vb, 8 * sB, S.w.y); bst(r, vb, 9 * sB, S.w.z);
  bst(r, vb, 10 * sB, S.v.x); bst(r, vb, 11 * sB, S.v.y); bst(r, vb, 12 * sB, S.v.z);
}

TOLG_DEV State load_state(const Params& P, const double* __restrict__ s, int i, int b) {
  State S;
  S.X.q.x = s[SIDX(0, i, b)]; S.X.q.y = s[SIDX(1, i, b)]; S.X.q.z = s[SIDX(2, i, b)]; S.X.q.w = s[SIDX(3, i, b)];
  S.X.t = v3(s[SIDX(4, i, b)], s[SIDX(5, i, b)], s[SIDX(6, i, b)]);
  S.w = v3(s[SIDX(7, i, b)], s[SIDX(8, i, b)], s[SIDX(9, i, b)]);
  S.v = v3(s[SIDX(10, i, b)], s[SIDX(11, i, b)], s[SIDX(12, i, b)]);
  return S;
}
template <int NT = 0>
TOLG_DEV void store_state(const Params& P, double* __restrict__ s, int i, int b, const State& S) {
  const double x[13] = {S.X.q.x, S.X.q.y, S.X.q.z, S.X.q.w, S.X.t.x, S.X.t.y, S.X.t.z, S.w.x, S.w.y, S.w.z, S.v.x, S.v.y, S.v.z};
#pragma unroll
  for (int c = 0; c < 13; c++) gst<NT>(&s[SIDX(c, i, b)], x[c]);
}

// fd_euler (traopt_dynamics.py:763-787 SE3, :1373-1401 Drone, :1049-1077 RigidBody):
// q+ = q Exp(xi dt) (re-normalised), xi+ = xi + J^-1 (ad(xi)^T J xi + g(q) + Pu u) dt
TOLG_DEV V3 mv33(const double* A, V3 x) {
  return v3(A[0] * x.x + A[1] * x.y + A[2] * x.z, A[3] * x.x + A[4] * x.y + A[5] * x.z,
            A[6] * x.x + A[7] * x.y + A[8] * x.z);
}
// entry (6 + r, u) of F_u (m columns), r = 0..5
template <int M, class CT>
TOLG_DEV double fu_entry(const CT& C, int r, int u) {
  if (r < 3) return (u < 3) ? C.Bt[3 * r + u] : 0.0;
  return (u >= 3) ? C.Bb[3 * (r - 3) + (u - 3)] : 0.0;
}
// PK: 0 the model is never the pendulum, 1 always, 2 decided at run time (kernel-uniform).  The sequential
// rollout instantiates 0 / 1 separately: even an untaken uniform branch costs its schedule 2-3 %.
template <int M, class CT, int PK = 2>
TOLG_DEV State dyn_f(const CT& C, const State& S, const double (&u)[M]) {
  State F;
  const double dt = C.dt;
  Pose E = se3_exp(dt * S.w, dt * S.v);
  F.X = se3_project(se3_compose(S.X, E));
  // J = blkdiag(Ib, Jv): J xi = [Ib w; Jv v]
  const bool dj = C.diagJ != 0;  // kernel-uniform
  V3 y1, y2;
  if (dj) {
    y1 = v3(C.Ib[0] * S.w.x, C.Ib[4] * S.w.y, C.Ib[8] * S.w.z);
    y2 = v3(C.Jv[0] * S.v.x, C.Jv[4] * S.v.y, C.Jv[8] * S.v.z);
  } else {
    y1 = mv33(C.Ib, S.w);
    y2 = mv33(C.Jv, S.v);
  }
  V3 top = cross(y1, S.w) + cross(y2, S.v);  // ad(xi)^T (J xi), upper half
  V3 bot = cross(y2, S.w);
  if (PK == 1 || (PK == 2 && C.kind == TOLG_DYN_PENDULUM3D)) {
    // Pendulum3dDyanmics.fd_euler (traopt_dynamics.py:531-552): g_term + M =
    // skew(m rho) R^T (g (0,0,-1) + u) with rho = l/2 (0,0,-1), i.e. k (w_y, -w_x, 0), k = m l / 2
    V3 wv = qrot_inv(S.X.q, v3(u[0], u[1], u[2] - C.grav));
    top = top + v3(C.pend_k * wv.y, -C.pend_k * wv.x, 0.0);
  } else {
    if (C.grav != 0.0) bot = bot + (C.mass * C.grav) * qrot_inv(S.X.q, v3(0, 0, -1.0));
    // Pu u: identity (SE3 / RigidBody) or the drone selector (u0..2 -> torque, u3 -> f_z)
    top = top + v3(u[0], u[1], u[2]);
  }
  if constexpr (M == 6) bot = bot + v3(u[3], u[4], u[5]);
  else bot = bot + v3(0, 0, u[3]);
  if (dj) {
    F.w = S.w + v3(C.Bt[0] * top.x, C.Bt[4] * top.y, C.Bt[8] * top.z);  // Bt = Ib^-1 dt (K1 / probe: no knot loop around it)
    F.v = S.v + dt * v3(C.Jvinv[0] * bot.x, C.Jvinv[4] * bot.y, C.Jvinv[8] * bot.z);
  } else {
    F.w = S.w + dt * mv33(C.Ibinv, top);
    F.v = S.v + dt * mv33(C.Jvinv, bot);
  }
  return F;
}

// The constants of the diagonal-inertia path, pinned in vector registers for the sequential rollout.
// Read on demand they are scalar loads the allocator prefers to re-issue over keeping 28 SGPRs alive:
// fourteen `s_load; s_waitcnt lgkmcnt(0)` pairs per knot, each an exposed scalar-cache round trip on a
// wave that has nothing else to run.
struct DynK { double dt, ib[3], jv[3], bt[3], jvi[3], mg; bool diag; };
TOLG_DEV double pin_v(double x) { asm volatile("" : "+v"(x)); return x; }
template <class CT>
TOLG_DEV DynK dynk_load(const CT& C) {
  DynK K;
  K.diag = C.diagJ != 0;
  K.dt = pin_v(C.dt);
#pragma unroll
  for (int a = 0; a < 3; a++) {
    K.ib[a] = pin_v(C.Ib[4 * a]); K.jv[a] = pin_v(C.Jv[4 * a]);
    K.bt[a] = pin_v(C.Bt[4 * a]); K.jvi[a] = pin_v(C.Jvinv[4 * a]);
  }
  K.mg = pin_v(C.mass * C.grav);
  return K;
}
// dyn_f with the pinned constants: the same expressions in the same order as the diagJ branch above,
// in two halves.  The pose half needs the state only, so the rollout runs it while the gains it is
// about to multiply are still in flight.
TOLG_DEV Pose dyn_pose_k(const DynK& K, const State& S, SeriesGate g) {
  Pose E = se3_exp_fast(K.dt * S.w, K.dt * S.v, g);
  return se3_project(se3_compose(S.X, E));
}
TOLG_DEV Pose dyn_pose_k(const DynK& K, const State& S) {
  const V3 wd = K.dt * S.w;
  const double th2 = dot(wd, wd);
  return dyn_pose_k(K, S, series_gate(exp_small(th2), exp_dom(th2)));
}
template <int M, class CT, int PK>
TOLG_DEV void dyn_twist_k(const DynK& K, const CT& C, const State& S, const double (&u)[M], State& F) {
  const double dt = K.dt;
  V3 y1 = v3(K.ib[0] * S.w.x, K.ib[1] * S.w.y, K.ib[2] * S.w.z);
  V3 y2 = v3(K.jv[0] * S.v.x, K.jv[1] * S.v.y, K.jv[2] * S.v.z);
  V3 top = cross(y1, S.w) + cross(y2, S.v);
  V3 bot = cross(y2, S.w);
  if (PK == 1) {
    V3 wv = qrot_inv(S.X.q, v3(u[0], u[1], u[2] - C.grav));
    top = top + v3(C.pend_k * wv.y, -C.pend_k * wv.x, 0.0);
  } else {
    if (K.mg != 0.0) bot = bot + K.mg * qrot_inv(S.X.q, v3(0, 0, -1.0));
    top = top + v3(u[0], u[1], u[2]);
  }
  if constexpr (M == 6) bot = bot + v3(u[3], u[4], u[5]);
  else bot = bot + v3(0, 0, u[3]);
  F.w = S.w + v3(K.bt[0] * top.x, K.bt[1] * top.y, K.bt[2] * top.z);
  F.v = S.v + dt * v3(K.jvi[0] * bot.x, K.jvi[1] * bot.y, K.jvi[2] * bot.z);
}
template <int M, class CT, int PK>
TOLG_DEV State dyn_f_k(const DynK& K, const CT& C, const State& S, const double (&u)[M]) {
  if (!K.diag) return dyn_f<M, CT, PK>(C, S, u);
  State F;
  F.X = dyn_pose_k(K, S);
  dyn_twist_k<M, CT, PK>(K, C, S, u, F);
  return F;
}

// dyn_f_k for kernels that serve every model: the pendulum decided at run time (kernel-uniform)
template <int M, class CT>
TOLG_DEV State dyn_f_any(const DynK& K, const CT& C, const State& S, const double (&u)[M]) {
  if (C.kind == TOLG_DYN_PENDULUM3D) return dyn_f_k<M, CT, 1>(K, C, S, u);
  return dyn_f_k<M, CT, 0>(K, C, S, u);
}

// ------------------------------------------------------------------------------------------------
// pack / unpack between the reference's 4x4 AoS layout (C ABI) and the device SoA layout
// ------------------------------------------------------------------------------------------------
TOLG_DEV Pose pose_from_m16(const double* __restrict__ Mx) {
  double R[9] = {Mx[0], Mx[1], Mx[2], Mx[4], Mx[5], Mx[6], Mx[8], Mx[9], Mx[10]};
  Pose X;
  X.q = R_to_q(R);
  X.t = v3(Mx[3], Mx[7], Mx[11]);
  return X;
}
TOLG_DEV void pose_to_m16(Pose X, double* __restrict__ Mx) {
  double R[9];
  q_to_R(X.q, R);
  Mx[0] = R[0]; Mx[1] = R[1]; Mx[2] = R[2];  Mx[3] = X.t.x;
  Mx[4] = R[3]; Mx[5] = R[4]; Mx[6] = R[5];  Mx[7] = X.t.y;
  Mx[8] = R[6]; Mx[9] = R[7]; Mx[10] = R[8]; Mx[11] = X.t.z;
  Mx[12] = 0; Mx[13] = 0; Mx[14] = 0; Mx[15] = 1;
}

__global__ void k_pack_ref(int N, const double* __restrict__ q_ref, const double* __restrict__ xi_ref,
                           double* __restrict__ ref) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i > N) return;
  Pose X = pose_from_m16(q_ref + 16 * (size_t)i);
  double* r = ref + 13 * (size_t)i;
  r[0] = X.q.x; r[1] = X.q.y; r[2] = X.q.z; r[3] = X.q.w; r[4] = X.t.x; r[5] = X.t.y; r[6] = X.t.z;
  for (int a = 0; a < 6; a++) r[7 + a] = xi_ref[6 * (size_t)i + a];
}

// MS _initial_guess (traopt_controller.py:3123-3136): knot 0 = x0, knots 1..N = reference;
// SS: only knot 0 (the rest comes from k_init_rollout).  Padded trajectories replicate b = B-1.
__global__ void k_init(Params P, const double* __restrict__ x0_q, const double* __restrict__ x0_xi,
                       const double* __restrict__ us_init, int ms) {
  size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= (size_t)(P.N + 1) * P.Bp) return;
  int b = (int)(t % P.Bp), i = (int)(t / P.Bp);
  int bs = b < P.B ? b : P.B - 1;
  State S;
  if (i == 0) {
    S.X = pose_from_m16(x0_q + 16 * (size_t)bs);
    const double* x = x0_xi + 6 * (size_t)bs;
    S.w = v3(x[0], x[1], x[2]);
    S.v = v3(x[3], x[4], x[5]);
  } else {
    const double* r = P.ref + 13 * (size_t)i;
    S.X.q.x = r[0]; S.X.q.y = r[1]; S.X.q.z = r[2]; S.X.q.w = r[3];
    S.X.t = v3(r[4], r[5], r[6]);
    S.w = v3(r[7], r[8], r[9]);
    S.v = v3(r[10], r[11], r[12]);
  }
  if (i == 0 || ms) store_state(P, P.cur, i, b, S);
  if (i < P.N)
    for (int a = 0; a < P.m; a++) P.cur_u[UIDX(a, i, b)] = us_init[((size_t)bs * P.N + i) * P.m + a];
  if (i == 0) {
    P.mu[b] = 1.0; P.delta[b] = 2.0; P.active[b] = 1; P.iters[b] = 0; P.status[b] = 0; P.conv[b] = 0;
    P.grad[b] = 0; P.Jc[b] = 0; P.dn[b] = 0; P.ls_alpha[b] = 1.0; P.ls_accept[b] = -1; P.ls_slot[b] = -1;
  }
}

// arbitrary trajectories in (unit-parity entry points)
__global__ void k_pack_traj(Params P, const double* __restrict__ xs_q, const double* __restrict__ xs_xi,
                            const double* __restrict__ us, const double* __restrict__ mu_delta) {
  size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= (size_t)(P.N + 1) * P.Bp) return;
  int b = (int)(t % P.Bp), i = (int)(t / P.Bp);
  int bs = b < P.B ? b : P.B - 1;
  State S;
  S.X = pose_from_m16(xs_q + ((size_t)bs * (P.N + 1) + i) * 16);
  const double* x = xs_xi + ((size_t)bs * (P.N + 1) + i) * 6;
  S.w = v3(x[0], x[1], x[2]);
  S.v = v3(x[3], x[4], x[5]);
  store_state(P, P.cur, i, b, S);
  if (i < P.N)
    for (int a = 0; a < P.m; a++) P.cur_u[UIDX(a, i, b)] = us[((size_t)bs * P.N + i) * P.m + a];
  if (i == 0) {
    P.mu[b] = mu_delta ? mu_delta[2 * bs] : 1.0;
    P.delta[b] = mu_delta ? mu_delta[2 * bs + 1] : 2.0;
    P.active[b] = 1; P.iters[b] = 0; P.status[b] = 0; P.conv[b] = 0; P.ls_alpha[b] = 1.0; P.ls_accept[b] = -1;
    P.ls_slot[b] = -1;
  }
}

__global__ void k_unpack_traj(Params P, const double* __restrict__ s, const double* __restrict__ su,
                              double* __restrict__ xs_q, double* __restrict__ xs_xi, double* __restrict__ us) {
  size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= (size_t)(P.N + 1) * P.Bp) return;
  int b = (int)(t % P.Bp), i = (int)(t / P.Bp);
  if (b >= P.B) return;
  State S = load_state(P, s, i, b);
  if (xs_q) pose_to_m16(S.X, xs_q + ((size_t)b * (P.N + 1) + i) * 16);
  if (xs_xi) {
    double* x = xs_xi + ((size_t)b * (P.N + 1) + i) * 6;
    x[0] = S.w.x; x[1] = S.w.y; x[2] = S.w.z; x[3] = S.v.x; x[4] = S.v.y; x[5] = S.v.z;
  }
  if (us && i < P.N)
    for (int a = 0; a < P.m; a++) us[((size_t)b * P.N + i) * P.m + a] = su[UIDX(a, i, b)];
}

// SS _init_rollout (traopt_controller.py:2015-2028): x_{i+1} = f(x_i, u_i), one thread per trajectory
template <int M>
__global__ void k_init_rollout(Params P) {
  int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= P.Bp) return;
  const Consts& C = *P.c;
  State S = load_state(P, P.cur, 0, b);
  for (int i = 0; i < P.N; i++) {
    double u[M];
#pragma unroll
    for (int a = 0; a < M; a++) u[a] = P.cur_u[UIDX(a, i, b)];
    S = dyn_f<M>(C, S, u);
    store_state(P, P.cur, i + 1, b, S);
  }
}

// A22 = I + H dt (column-major), H = J^-1 (coadjoint([v, w]) J + G)  <- literal swapped twist (App. C-Q1)
// (traopt_dynamics.py:802-837, :1416-1469).  With J = blkdiag(Ib, Jv):
// coadjoint([v,w]) J + G = [[S(Ib w) - Sv Ib, m Sv - Sw Jv],[m Sv, -Sv Jv]]
template <class CT>
TOLG_DEV void a22_build(const CT& C, V3 w, V3 v, double (&a22)[36]) {
  const double dt = C.dt;
  double Sv[9], Sw[9], SIw[9], M11[9], M12[9], M21[9], M22[9], T1[9], T2[9];
  skew(v, Sv);
  skew(w, Sw);
  skew(mv33(C.Ib, w), SIw);
  mul33(Sv, C.Ib, T1);
  mul33(Sw, C.Jv, T2);
#pragma unroll
  for (int k = 0; k < 9; k++) { M11[k] = SIw[k] - T1[k]; M12[k] = C.mass * Sv[k] - T2[k]; M21[k] = C.mass * Sv[k]; }
  mul33(Sv, C.Jv, T1);
#pragma unroll
  for (int k = 0; k < 9; k++) M22[k] = -T1[k];
  if (so3_family(C.kind)) {
    // SO3Dynamics.f_x (and Pendulum3dDyanmics.f_x :566-567) (traopt_dynamics.py:385-400): H = J^-1 (skew(w)^T J + skew(J w)) -- the SO(3)
    // model has no swapped-twist quirk; the unused linear-velocity block is the identity
    mul33(Sw, C.Ib, T1);
#pragma unroll
    for (int k = 0; k < 9; k++) { M11[k] = SIw[k] - T1[k]; M12[k] = 0; M21[k] = 0; M22[k] = 0; }
  }
  double H11[9], H12[9], H21[9], H22[9];
  mul33(C.Ibinv, M11, H11);
  mul33(C.Ibinv, M12, H12);
  mul33(C.Jvinv, M21, H21);
  mul33(C.Jvinv, M22, H22);
#pragma unroll
  for (int r = 0; r < 3; r++)
#pragma unroll
    for (int c = 0; c < 3; c++) {
      double id = (r == c) ? 1.0 : 0.0;
      a22[6 * c + r] = id + dt * H11[3 * r + c];
      a22[6 * (c + 3) + r] = dt * H12[3 * r + c];
      a22[6 * c + r + 3] = dt * H21[3 * r + c];
      a22[6 * (c + 3) + r + 3] = id + dt * H22[3 * r + c];
    }
}
// the block of knot i of trajectory b, from the record where it is stored, from the record's twist where it is not
TOLG_DEV void a22_get(const Params& P, const Consts& C, int i, int b, double (&a22)[36]) {
  if (P.fA22 >= 0) {
#pragma unroll
    for (int k = 0; k < 36; k++) a22[k] = P.REC[RIDX(i, P.fA22 + k, b)];
  } else {
    a22_build(C, v3(P.REC[RIDX(i, REC_XI, b)], P.REC[RIDX(i, REC_XI + 2, b)], P.REC[RIDX(i, REC_XI + 4, b)]),
              v3(P.REC[RIDX(i, REC_XI + 1, b)], P.REC[RIDX(i, REC_XI + 3, b)], P.REC[RIDX(i, REC_XI + 5, b)]), a22);
  }
}

// N consecutive record fields from F0, as 16-byte stores wherever a field pair is complete
template <int F0, int N, int NT = TOLG_NT_RECST>
TOLG_DEV void rec_run(const Params& P, int i, int b, const double (&v)[N]) {
  constexpr int H = F0 & 1;  // an odd first field goes alone
  if constexpr (H) gst<NT>(&P.REC[RIDX(i, F0, b)], v[0]);
#pragma unroll
  for (int k = H; k + 1 < N; k += 2) {
    f64x2 w = {v[k], v[k + 1]};
    gst<NT>(reinterpret_cast<f64x2*>(&P.REC[RIDX(i, F0 + k, b)]), w);
  }
  if constexpr (((N - H) & 1) != 0) gst<NT>(&P.REC[RIDX(i, F0 + N - 1, b)], v[N - 1]);
}

// ------------------------------------------------------------------------------------------------
// K1: linearisation of one knot (dynamics f, MS defect, F_x blocks, cost gradient / Gauss-Newton Hessian).
// lin_knot is the body; k_linearize runs it one thread per (trajectory, knot) on a stored trajectory,
// k_rollout_lin (below) runs it in helper wavefronts on the states the rollout has just produced.
// ms: 1 multiple shooting, 0 single shooting, 2 probe (tolg_eval_knot: f(x,u) goes to knot i of P.cand,
// the tracking error to REC_D, nothing is read from knot i+1).  next_state() returns the state of knot
// i + 1 (called once, late, and only for ms == 1 on a non-terminal knot).
// ------------------------------------------------------------------------------------------------
// CLOSED: the caller guarantees x_{i+1} = f(x_i, u_i) by construction (the fused rollout, whose step is exactly
// that): the defect is zero and is not recomputed (the reference obtains rounding noise of ~1e-16 per entry here).
// Exp / Log and the Jacobian coefficients use the series forms of tolg_lie.h inside their convergence domains.
// lcost: where the stage cost goes instead of P.SC (the fused kernel sums the costs of its own trajectories itself,
// from LDS; a CLOSED trajectory has no defect to sum either).
// TERM: -1 = decided per lane (i == N), 0 / 1 = the caller knows that no / every lane it calls with sits on the
// terminal knot.  With a wave-uniform answer the weight matrices below are selected once per wave and read with
// scalar loads; a per-lane select turns every one of their ~150 reads into a vector-memory load of one address.
template <int M, bool CLOSED = false, int TERM = -1, class CT, class NextFn>
TOLG_DEV void lin_knot(const Params& P, const CT& C, int i, int b, int ms, const State& S, const double (&u)[M],
                       NextFn next_state, double* lcost = nullptr) {
  const bool term = TERM < 0 ? (i == P.N) : (TERM == 1);
  // store policy of the record: the fused launch's helpers (CLOSED) and K1 on its own were tuned separately (TOLG_NT_* note)
  constexpr int RP = CLOSED ? TOLG_NT_RECST : TOLG_NT_RECST_K1;
  const double dt = C.dt;
  // One gate for all the series evaluations of the knot: the tracking error (Log, then V^-1 and Q at its angle,
  // which the Log bounds cover) and the step rotation (V, Q, Exp at dt w).
  const V3 wd = dt * S.w, vd = dt * S.v;
  const double* r = P.ref + 13 * (size_t)i;
  Pose Xr;
  Xr.q.x = r[0]; Xr.q.y = r[1]; Xr.q.z = r[2]; Xr.q.w = r[3];
  Xr.t = v3(r[4], r[5], r[6]);
  const Pose De = se3_compose(S.X, se3_inverse(Xr));
  const double ye = quat_vec2(De.q), th2d = dot(wd, wd);
  const SeriesGate sg = series_gate(log_small(ye) && coef_small(th2d), log_dom(ye) && exp_dom(th2d));
  // ---------------- cost: e = Log(X Xref^-1), J_e = Jr^-1(e) Ad(Xref)  (traopt_cost.py:659-839)
  {
    V3 ew, ev;
    se3_log_fast(De, ew, ev, sg);
    // weights: l_xx switches to P at the terminal knot; l and l_x too, except for the SO3 cost which
    // keeps Q there (traopt_cost.py:434-438, :480-483 vs :530-531)
    const bool so3 = so3_family(C.kind);
    // (pointers into C keep its address space: scalar loads when C is the constant-space view and term is uniform)
    const auto W1 = term ? &C.P1[0] : &C.W1[0];
    const auto W2 = term ? &C.P2[0] : &C.W2[0];
    const auto G1 = (term && !so3) ? &C.P1[0] : &C.W1[0];
    const auto G2 = (term && !so3) ? &C.P2[0] : &C.W2[0];
    double e[6] = {ew.x, ew.y, ew.z, ev.x, ev.y, ev.z};
    double ve[6] = {S.w.x - r[7], S.w.y - r[8], S.w.z - r[9], S.v.x - r[10], S.v.y - r[11], S.v.z - r[12]};
    double th2 = dot(ew, ew);
    double Ji[9], Qr[9], T1[9], Bm[9], Rr[9], Tr[9], Ja[9], Jb[9];
    ljacinv33(neg(ew), ljacinv_coef_fast(th2, sg), Ji);  // Jr^-1(w) = Jl^-1(-w)
    Q33(neg(ev), neg(ew), so3_coef_fast(th2, true, sg), Qr);
    mul33(Ji, Qr, T1);
    mul33(T1, Ji, Bm);  // rjacinv lower-left block = -Bm
    q_to_R(Xr.q, Rr);
    skew(Xr.t, Tr);
    mul33(Ji, Rr, Ja);
    mul33(Ji, Tr, T1);
#pragma unroll
    for (int k = 0; k < 9; k++) T1[k] -= Bm[k];
    mul33(T1, Rr, Jb);
    // J_e = [[Ja, 0],[Jb, Ja]]
    double Je[36];
#pragma unroll
    for (int a = 0; a < 3; a++)
#pragma unroll
      for (int c = 0; c < 3; c++) {
        Je[6 * a + c] = Ja[3 * a + c];
        Je[6 * a + c + 3] = 0;
        Je[6 * (a + 3) + c] = Jb[3 * a + c];
        Je[6 * (a + 3) + c + 3] = Ja[3 * a + c];
      }
    if (ms == 2) {
      double ed[12];
#pragma unroll
      for (int a = 0; a < 6; a++) { ed[a] = e[a]; ed[6 + a] = ve[a]; }
      rec_run<REC_D, 12, RP>(P, i, b, ed);
    }
    double We[6], W2v[6], l = 0;
#pragma unroll
    for (int a = 0; a < 6; a++) {
      double s1 = 0, s2 = 0;
#pragma unroll
      for (int k = 0; k < 6; k++) { s1 += G1[6 * a + k] * e[k]; s2 += G2[6 * a + k] * ve[k]; }
      We[a] = s1; W2v[a] = s2;
      l += e[a] * s1 + ve[a] * s2;
    }
    if (!term) {
#pragma unroll
      for (int a = 0; a < M; a++)
#pragma unroll
        for (int k = 0; k < M; k++) l += u[a] * C.R[a * M + k] * u[k];
    }
    if (!term) {  // l_u = 2 R u (traopt_cost.py:792-804)
      double lu[M], luu[M];
#pragma unroll
      for (int a = 0; a < M; a++) {
        double sacc = 0;
#pragma unroll
        for (int k = 0; k < M; k++) sacc += 2.0 * C.R[a * M + k] * u[k];
        lu[a] = sacc;
        luu[a] = 0.0;
      }
      if (P.al_lb) {
        // LA = l + lambda^T g + g^T I_mu g / 2, g = [lb - u; u - ub], g_u = [-I; I]
        // (traopt_cost.py:1219-1224, :1262-1266, :1302-1306; traopt_constraints.py:130-133, :167-169)
        const int bs = b < P.B ? b : P.B - 1;
        const double* lam = P.al_lambda + ((size_t)bs * P.N + i) * 2 * M;
        const double* imu = P.al_imu + ((size_t)bs * P.N + i) * 2 * M;
#pragma unroll
        for (int a = 0; a < M; a++) {
          double g1 = P.al_lb[a] - u[a], g2 = u[a] - P.al_ub[a];
          l += lam[a] * g1 + lam[M + a] * g2 + 0.5 * (g1 * imu[a] * g1 + g2 * imu[M + a] * g2);
          lu[a] += -(lam[a] + imu[a] * g1) + (lam[M + a] + imu[M + a] * g2);
          luu[a] = imu[a] + imu[M + a];
        }
      }
      rec_run<REC_LU, M, RP>(P, i, b, lu);
      if (P.al_lb) {  // the field exists in AL solves only (Params::fLUU, even)
#pragma unroll
        for (int k = 0; k < M; k += 2) {
          f64x2 w = {luu[k], luu[k + 1]};
          *reinterpret_cast<f64x2*>(&P.REC[RIDX(i, P.fLUU + k, b)]) = w;
        }
      }
    }
    if (lcost) *lcost = l;
    else P.SC[(size_t)i * P.Bp + b] = l;
    double WJ[36];
#pragma unroll
    for (int a = 0; a < 6; a++)
#pragma unroll
      for (int c = 0; c < 6; c++) {
        double s = 0;
#pragma unroll
        for (int k = 0; k < 6; k++) s += W1[6 * a + k] * Je[6 * k + c];
        WJ[6 * a + c] = s;
      }
    double lxx[21], lxv[12];
#pragma unroll
    for (int a = 0; a < 6; a++) {
#pragma unroll
      for (int c = a; c < 6; c++) {  // upper triangle; l_xx = 2 Je^T W1 Je is symmetric for symmetric W1
        double s = 0, s2 = 0;
#pragma unroll
        for (int k = 0; k < 6; k++) { s += Je[6 * k + a] * WJ[6 * k + c]; s2 += Je[6 * k + c] * WJ[6 * k + a]; }
        lxx[sym6(a, c)] = s + s2;  // == 2 * symmetric part
      }
      double s = 0;
#pragma unroll
      for (int k = 0; k < 6; k++) s += Je[6 * k + a] * We[k];
      lxv[a] = 2 * s;
      lxv[6 + a] = 2 * W2v[a];
    }
    rec_run<REC_LXX, 21, RP>(P, i, b, lxx);
    rec_run<REC_LX, 12, RP>(P, i, b, lxv);
  }
  if (term) return;
  // ---------------- dynamics Jacobian blocks (traopt_dynamics.py:802-837, :1416-1469)
  SO3Coef kc = so3_coef_fast(dot(wd, wd), true, sg);
  Pose E;
  E.q = so3_exp_fast(wd, sg);
  E.t = ljac_apply(wd, kc, vd);
  {
    // Ad(Exp(tau))^-1 = Ad(E^-1) = [[Ri,0],[[ti]x Ri, Ri]]
    Pose Ei = se3_inverse(E);
    double Ri[9], Ti[9], TR[9];
    q_to_R(Ei.q, Ri);
    skew(Ei.t, Ti);
    mul33(Ti, Ri, TR);
    // Jr(tau) dt = [[Jr3,0],[Qr,Jr3]] dt, Jr(tau) = Jl(-tau)
    double Jr3[9], Qr[9];
    ljac33(neg(wd), kc, Jr3);
    Q33(neg(vd), neg(wd), kc, Qr);
    static_assert(REC_RI == 0 && REC_TRI == 9 && REC_JR == 18 && REC_QR == 27, "the four pose blocks form one run");
    double blk[36];
#pragma unroll
    for (int a = 0; a < 3; a++)
#pragma unroll
      for (int c = 0; c < 3; c++) {
        blk[REC_RI + 3 * c + a] = Ri[3 * a + c];
        blk[REC_TRI + 3 * c + a] = TR[3 * a + c];
        blk[REC_JR + 3 * c + a] = dt * Jr3[3 * a + c];
        blk[REC_QR + 3 * c + a] = dt * Qr[3 * a + c];
      }
    if (C.kind == TOLG_DYN_PENDULUM3D) {  // the REC_TRI slot is REC_BU there (written below), the block is zero
#pragma unroll
      for (int k = 0; k < 9; k++) P.REC[RIDX(i, REC_RI + k, b)] = blk[k];
#pragma unroll
      for (int k = 18; k < 36; k++) P.REC[RIDX(i, k, b)] = blk[k];
    } else {
      rec_run<0, 36, RP>(P, i, b, blk);
    }
  }
  {
    const double xi6[6] = {S.w.x, S.v.x, S.w.y, S.v.y, S.w.z, S.v.z};  // (w_k, v_k) pairs: one 16-byte read each in k_backward3
    rec_run<REC_XI, 6, RP>(P, i, b, xi6);
    if (P.fA22 >= 0) {  // kernel-uniform: the models whose backward sweep reads the block from the record
      double a22[36];
      a22_build(C, S.w, S.v, a22);
#pragma unroll
      for (int k = 0; k < 36; k += 2) {
        f64x2 w2 = {a22[k], a22[k + 1]};
        *reinterpret_cast<f64x2*>(&P.REC[RIDX(i, P.fA22 + k, b)]) = w2;
      }
    }
  }
  if (C.grav != 0.0) {  // gravity models only (the field does not exist otherwise)
    V3 rte = qrot_inv(S.X.q, v3(0, 0, -1.0));
    if (C.kind == TOLG_DYN_PENDULUM3D) {
      // Pendulum3dDyanmics.f_x / f_u (traopt_dynamics.py:574-609).  L1 + L2 = skew(m rho) skew(w) with
      // w = R^T (g (0,0,-1) + u): the lower-left block stays linear in one body-frame vector, which takes
      // the place of R^T e3 here; F_u = J^-1 skew(m rho) R^T dt goes to REC_BU.
      rte = qrot_inv(S.X.q, v3(u[0], u[1], u[2] - C.grav));
      double Rm[9], SR[9], Bu[9];
      q_to_R(S.X.q, Rm);
#pragma unroll
      for (int c = 0; c < 3; c++) {  // skew(m rho) R^T = k [row 1 of R^T; -row 0 of R^T; 0]
        SR[c] = C.pend_k * Rm[3 * c + 1];
        SR[3 + c] = -C.pend_k * Rm[3 * c + 0];
        SR[6 + c] = 0.0;
      }
      mul33(C.Ibinv, SR, Bu);
#pragma unroll
      for (int k = 0; k < 9; k++) P.REC[RIDX(i, REC_BU + k, b)] = dt * Bu[k];
    }
    const double rt[3] = {rte.x, rte.y, rte.z};
    rec_run<REC_LU + M, 3, RP>(P, i, b, rt);
  }
  // ---------------- defect d = [Log(x_{i+1}^-1 f_q); f_xi - xi_{i+1}]  (traopt_controller.py:2882-2888)
  double d[12];
  if (ms == 2) {  // probe: export f(x, u) itself
    State F = dyn_f<M>(C, S, u);
    store_state(P, P.cand, i, b, F);
    return;
  }
  if (ms && !CLOSED) {
    State F = dyn_f<M>(C, S, u);
    State Sn = next_state();
    V3 dw, dv;
    se3_log(se3_compose(se3_inverse(Sn.X), F.X), dw, dv);
    d[0] = dw.x; d[1] = dw.y; d[2] = dw.z; d[3] = dv.x; d[4] = dv.y; d[5] = dv.z;
    V3 dxw = F.w - Sn.w, dxv = F.v - Sn.v;
    d[6] = dxw.x; d[7] = dxw.y; d[8] = dxw.z; d[9] = dxv.x; d[10] = dxv.y; d[11] = dxv.z;
  } else {
#pragma unroll
    for (int a = 0; a < 12; a++) d[a] = 0;
  }
  double d2 = 0;
#pragma unroll
  for (int a = 0; a < 12; a++) d2 += d[a] * d[a];
  if (!(CLOSED && lcost)) {  // the fused launch neither stores nor sums a defect that is zero by construction
    rec_run<REC_D, 12, RP>(P, i, b, d);
    P.SD[(size_t)i * P.Bp + b] = d2;
  }
}

TOLG_DEV bool ls_quad_form(const Params& P, int list, int nslots);  // (line-search stages, below)
// two waves per SIMD: the kernel needed 260 registers, four past the limit for a second wave -- at 256 (12 bytes of scratch for
// m = 6) 0.232 -> 0.217 ms per call at 4096 x 200; three waves (168 registers, 284 bytes of scratch): 0.31 ms
#ifndef TOLG_K1_WPE
#define TOLG_K1_WPE 2
#endif
template <int M>
__global__ __launch_bounds__(256, TOLG_K1_WPE) void k_linearize(Params P, const double* __restrict__ src,
                                                    const double* __restrict__ src_u, double* __restrict__ dst,
                                                    double* __restrict__ dst_u, int ms, int i0, int ni, int ls_list, int ls_nslots) {
  // knots [i0, i0 + ni); dst / dst_u (optional): the trajectory is copied there while it is read.
  // ls_nslots > 0 (round 4): the candidates the LAST stage of a line search accepted are read where that stage left them -- slot
  // ls_slot[b], at the trajectory's position on the stage's list or at b (k_ls_copy's rule) -- instead of behind a copy into the
  // candidate arrays; everything accepted earlier is in the candidate arrays (src) as before
  const DConsts& C = *(const DConsts*)P.c;
  size_t t = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (t >= (size_t)ni * P.Bp) return;
  const int b = (int)(t % P.Bp), i = i0 + (int)(t / P.Bp);
  if (!P.active[b]) return;
  int e = b;
  if (ls_nslots > 0) {
    const int sl = P.ls_slot[b];
    if (sl >= 0) {
      e = (ls_list >= 0 && ls_quad_form(P, ls_list, ls_nslots)) ? P.ls_pos[(size_t)ls_list * P.Bp + b] : b;
      src = P.slot_x + (size_t)sl * 13 * P.Bp * (P.N + 1);
      src_u = P.slot_u + (size_t)sl * M * P.Bp * P.N;
    }
  }
  State S = load_state(P, src, i, e);
  if (dst) store_state(P, dst, i, b, S);
  double u[M];
#pragma unroll
  for (int a = 0; a < M; a++) u[a] = 0;
  if (i < P.N) {
#pragma unroll
    for (int a = 0; a < M; a++) {
      u[a] = src_u[UIDX(a, i, e)];
      if (dst_u) dst_u[UIDX(a, i, b)] = u[a];
    }
  }
  lin_knot<M>(P, C, i, b, ms, S, u, [&]() { return load_state(P, src, i + 1, e); });
}

// per-trajectory sums of the stage costs / squared defects, fixed order (deterministic);
// also the on_iteration bookkeeping of traopt_controller.py:2621-2626
__global__ void k_reduce(Params P, int it) {
  int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= P.Bp || !P.active[b]) return;
  // fixed summation order (knot 0, 1, 2, ...); 64 waves cannot hide memory latency, so the loads go out
  // sixteen knots (32 loads) at a time
  double J = 0, d2 = 0;
  int i = 0;
  for (; i + 16 <= P.N; i += 16) {
    double c[16], d[16];
#pragma unroll
    for (int k = 0; k < 16; k++) { c[k] = P.SC[(size_t)(i + k) * P.Bp + b]; d[k] = P.SD[(size_t)(i + k) * P.Bp + b]; }
#pragma unroll
    for (int k = 0; k < 16; k++) { J += c[k]; d2 += d[k]; }
  }
  for (; i < P.N; i++) { J += P.SC[(size_t)i * P.Bp + b]; d2 += P.SD[(size_t)i * P.Bp + b]; }
  J += P.SC[(size_t)P.N * P.Bp + b];
  double dn = sqrt(d2);
  // a cost that grows tenfold in an iteration: the solve is diverging (accept-always steps only: the searches take no such step).
  // The fast backward sweep's four-knot symmetrisation period was measured on solves that converge (tolg_backward3.h); a group
  // with such a member takes the full kernel -- every knot, like the reference -- for its next sweeps.
  if (it >= 0 && b < P.B && J > 10.0 * P.Jc[b]) P.k2_hint[b >> 2] = 8;
  P.Jc[b] = J;
  P.dn[b] = dn;
  if (b >= P.B) return;
  if (it < 0) {
    if (P.defect_hist) P.defect_hist[(size_t)b * (P.max_iter + 1)] = dn;
    return;
  }
  if (P.J_hist) P.J_hist[(size_t)b * P.max_iter + it] = J;
  if (P.defect_hist) P.defect_hist[(size_t)b * (P.max_iter + 1) + it + 1] = dn;
  if (P.alpha_hist) P.alpha_hist[(size_t)b * P.max_iter + it] = P.ls_alpha[b];
  P.iters[b] = it + 1;
  if (!(J == J) || isinf(J)) { P.status[b] = TOLG_ST_NONFINITE; P.active[b] = 0; }
}

// ------------------------------------------------------------------------------------------------
// K2: backward Riccati sweep.  16 lanes per trajectory; lane j holds column j of every 12x12
// (lane 12: the vector column V_x / d / k, lane 13: SS adjoint), rows live in 12 VGPR pairs.
// C = P^T Q and C = P Q are rank-1 updates whose left factor comes from another lane of the same
// 16-lane row through DPP row_newbcast, fused into v_fmac_f64 (DP-ALU DPP on gfx90a+).
// ------------------------------------------------------------------------------------------------
#ifndef TOLG_DPP_BUILTIN
#define FM(a, p, q, L) "v_fmac_f64_dpp " a ", " p ", " q " row_newbcast:" #L " row_mask:0xf bank_mask:0xf\n\t"
// acc[i] += P[k][i] * q   with p = register holding row k of P (column per lane): P^T Q, one k
TOLG_DEV void rank1_bi(double (&acc)[12], double p, double q) {
  asm volatile("s_nop 1\n\t" FM("%0", "%12", "%13", 0) FM("%1", "%12", "%13", 1) FM("%2", "%12", "%13", 2)
                   FM("%3", "%12", "%13", 3) FM("%4", "%12", "%13", 4) FM("%5", "%12", "%13", 5)
                       FM("%6", "%12", "%13", 6) FM("%7", "%12", "%13", 7) FM("%8", "%12", "%13", 8)
                           FM("%9", "%12", "%13", 9) FM("%10", "%12", "%13", 10) FM("%11", "%12", "%13", 11)
               : "+v"(acc[0]), "+v"(acc[1]), "+v"(acc[2]), "+v"(acc[3]), "+v"(acc[4]), "+v"(acc[5]), "+v"(acc[6]),
                 "+v"(acc[7]), "+v"(acc[8]), "+v"(acc[9]), "+v"(acc[10]), "+v"(acc[11])
               : "v"(p), "v"(q));
}
// same, restricted to the 3-row blocks of P^T that are structurally non-zero (F_x block pattern)
#define FM3(a0, a1, a2, p, q, L0, L1, L2) FM(a0, p, q, L0) FM(a1, p, q, L1) FM(a2, p, q, L2)
// rows {0,1,2, 6,7,8}: used for k in the first row block of F_x = [Ri 0 Jr 0]
TOLG_DEV void rank1_bi_02(double (&acc)[12], double p, double q) {
  asm volatile("s_nop 1\n\t" FM3("%0", "%1", "%2", "%6", "%7", 0, 1, 2) FM3("%3", "%4", "%5", "%6", "%7", 6, 7, 8)
               : "+v"(acc[0]), "+v"(acc[1]), "+v"(acc[2]), "+v"(acc[6]), "+v"(acc[7]), "+v"(acc[8])
               : "v"(p), "v"(q));
}
// rows {6..11}: k in the lower row blocks [0 0 A22] (SE3Dynamics)
TOLG_DEV void rank1_bi_23(double (&acc)[12], double p, double q) {
  asm volatile("s_nop 1\n\t" FM3("%0", "%1", "%2", "%6", "%7", 6, 7, 8) FM3("%3", "%4", "%5", "%6", "%7", 9, 10, 11)
               : "+v"(acc[6]), "+v"(acc[7]), "+v"(acc[8]), "+v"(acc[9]), "+v"(acc[10]), "+v"(acc[11])
               : "v"(p), "v"(q));
}
// rows {0,1,2, 6..11}: lower row blocks with the gravity block [A21 0 A22] (Drone / RigidBody)
TOLG_DEV void rank1_bi_023(double (&acc)[12], double p, double q) {
  asm volatile("s_nop 1\n\t" FM3("%0", "%1", "%2", "%9", "%10", 0, 1, 2) FM3("%3", "%4", "%5", "%9", "%10", 6, 7, 8)
                   FM3("%6", "%7", "%8", "%9", "%10", 9, 10, 11)
               : "+v"(acc[0]), "+v"(acc[1]), "+v"(acc[2]), "+v"(acc[6]), "+v"(acc[7]), "+v"(acc[8]), "+v"(acc[9]),
                 "+v"(acc[10]), "+v"(acc[11])
               : "v"(p), "v"(q));
}
#define FMK(a, p, K) "v_fmac_f64_dpp " a ", " p ", %24 row_newbcast:" #K " row_mask:0xf bank_mask:0xf\n\t"
#define RANK1_BK(K)                                                                                               \
  asm volatile("s_nop 1\n\t" FMK("%0", "%12", K) FMK("%1", "%13", K) FMK("%2", "%14", K) FMK("%3", "%15", K)        \
                   FMK("%4", "%16", K) FMK("%5", "%17", K) FMK("%6", "%18", K) FMK("%7", "%19", K)                 \
                       FMK("%8", "%20", K) FMK("%9", "%21", K) FMK("%10", "%22", K) FMK("%11", "%23", K)           \
               : "+v"(acc[0]), "+v"(acc[1]), "+v"(acc[2]), "+v"(acc[3]), "+v"(acc[4]), "+v"(acc[5]), "+v"(acc[6]), \
                 "+v"(acc[7]), "+v"(acc[8]), "+v"(acc[9]), "+v"(acc[10]), "+v"(acc[11])                            \
               : "v"(Pm[0]), "v"(Pm[1]), "v"(Pm[2]), "v"(Pm[3]), "v"(Pm[4]), "v"(Pm[5]), "v"(Pm[6]), "v"(Pm[7]),   \
                 "v"(Pm[8]), "v"(Pm[9]), "v"(Pm[10]), "v"(Pm[11]), "v"(q))
// acc[i] += P[i][K] * q: P Q, one k (= K); Pm[i] is the register holding row i of P
template <int K>
TOLG_DEV void rank1_bk(double (&acc)[12], const double (&Pm)[12], double q) {
  if constexpr (K == 0) RANK1_BK(0);
  if constexpr (K == 1) RANK1_BK(1);
  if constexpr (K == 2) RANK1_BK(2);
  if constexpr (K == 3) RANK1_BK(3);
  if constexpr (K == 4) RANK1_BK(4);
  if constexpr (K == 5) RANK1_BK(5);
  if constexpr (K == 6) RANK1_BK(6);
  if constexpr (K == 7) RANK1_BK(7);
  if constexpr (K == 8) RANK1_BK(8);
  if constexpr (K == 9) RANK1_BK(9);
  if constexpr (K == 10) RANK1_BK(10);
  if constexpr (K == 11) RANK1_BK(11);
}
// ---- merged DPP blocks: tolg_dpp_blocks.h, with and without the leading hazard s_nop
#define TOLG_BLK(n) n
#define TOLG_BLK_NOP "s_nop 1\n\t"
#include "tolg_dpp_blocks.h"
#undef TOLG_BLK
#undef TOLG_BLK_NOP
#define TOLG_BLK(n) n##_nn
#define TOLG_BLK_NOP ""
#include "tolg_dpp_blocks.h"
#undef TOLG_BLK
#undef TOLG_BLK_NOP
#endif

// x of the lane six to the right in the same 16-lane row (0 past the row end).  64-bit DPP exists only for
// row_newbcast: the shift is done on the two halves (the type-generic builtin on a double converted the
// result numerically here).
TOLG_DEV double row_shl6(double x) {
  const unsigned long long u = __builtin_bit_cast(unsigned long long, x);
  const unsigned lo = (unsigned)__builtin_amdgcn_update_dpp(0, (int)(unsigned)u, 0x106, 0xf, 0xf, false);
  const unsigned hi = (unsigned)__builtin_amdgcn_update_dpp(0, (int)(unsigned)(u >> 32), 0x106, 0xf, 0xf, false);
  return __builtin_bit_cast(double, ((unsigned long long)hi << 32) | lo);
}
template <int L>
TOLG_DEV double bcast(double x) {  // value of x in lane L of this 16-lane row
  return __builtin_amdgcn_update_dpp(0.0, x, 0x150 + L, 0xf, 0xf, false);
}
// acc[u] += P[u]@lane K * q for u < M (one k of a P Q product with M rows)
template <int M, int K>
TOLG_DEV void quu_acc(double (&acc)[M], const double (&Pm)[M], double q) {
#ifndef TOLG_DPP_BUILTIN
  if constexpr (M == 6) {
    asm volatile("s_nop 1\n\t"
                 "v_fmac_f64_dpp %0, %6, %12 row_newbcast:%13 row_mask:0xf bank_mask:0xf\n\t"
                 "v_fmac_f64_dpp %1, %7, %12 row_newbcast:%13 row_mask:0xf bank_mask:0xf\n\t"
                 "v_fmac_f64_dpp %2, %8, %12 row_newbcast:%13 row_mask:0xf bank_mask:0xf\n\t"
                 "v_fmac_f64_dpp %3, %9, %12 row_newbcast:%13 row_mask:0xf bank_mask:0xf\n\t"
                 "v_fmac_f64_dpp %4, %10, %12 row_newbcast:%13 row_mask:0xf bank_mask:0xf\n\t"
                 "v_fmac_f64_dpp %5, %11, %12 row_newbcast:%13 row_mask:0xf bank_mask:0xf\n\t"
                 : "+v"(acc[0]), "+v"(acc[1]), "+v"(acc[2]), "+v"(acc[3]), "+v"(acc[4]), "+v"(acc[5])
                 : "v"(Pm[0]), "v"(Pm[1]), "v"(Pm[2]), "v"(Pm[3]), "v"(Pm[4]), "v"(Pm[5]), "v"(q), "n"(K));
  } else {
    asm volatile("s_nop 1\n\t"
                 "v_fmac_f64_dpp %0, %4, %8 row_newbcast:%9 row_mask:0xf bank_mask:0xf\n\t"
                 "v_fmac_f64_dpp %1, %5, %8 row_newbcast:%9 row_mask:0xf bank_mask:0xf\n\t"
                 "v_fmac_f64_dpp %2, %6, %8 row_newbcast:%9 row_mask:0xf bank_mask:0xf\n\t"
                 "v_fmac_f64_dpp %3, %7, %8 row_newbcast:%9 row_mask:0xf bank_mask:0xf\n\t"
                 : "+v"(acc[0]), "+v"(acc[1]), "+v"(acc[2]), "+v"(acc[3])
                 : "v"(Pm[0]), "v"(Pm[1]), "v"(Pm[2]), "v"(Pm[3]), "v"(q), "n"(K));
  }
#else
#pragma unroll
  for (int u = 0; u < M; u++) acc[u] += bcast<K>(Pm[u]) * q;
#endif
}
#ifdef TOLG_DPP_BUILTIN
TOLG_DEV void rank1_bi(double (&acc)[12], double p, double q) {
  acc[0] += bcast<0>(p) * q; acc[1] += bcast<1>(p) * q; acc[2] += bcast<2>(p) * q; acc[3] += bcast<3>(p) * q;
  acc[4] += bcast<4>(p) * q; acc[5] += bcast<5>(p) * q; acc[6] += bcast<6>(p) * q; acc[7] += bcast<7>(p) * q;
  acc[8] += bcast<8>(p) * q; acc[9] += bcast<9>(p) * q; acc[10] += bcast<10>(p) * q; acc[11] += bcast<11>(p) * q;
}
template <int K>
TOLG_DEV void rank1_bk(double (&acc)[12], const double (&Pm)[12], double q) {
#pragma unroll
  for (int i = 0; i < 12; i++) acc[i] += bcast<K>(Pm[i]) * q;
}
TOLG_DEV void rank1_bi_02(double (&acc)[12], double p, double q) {
  acc[0] += bcast<0>(p) * q; acc[1] += bcast<1>(p) * q; acc[2] += bcast<2>(p) * q;
  acc[6] += bcast<6>(p) * q; acc[7] += bcast<7>(p) * q; acc[8] += bcast<8>(p) * q;
}
TOLG_DEV void rank1_bi_23(double (&acc)[12], double p, double q) {
  acc[6] += bcast<6>(p) * q; acc[7] += bcast<7>(p) * q; acc[8] += bcast<8>(p) * q;
  acc[9] += bcast<9>(p) * q; acc[10] += bcast<10>(p) * q; acc[11] += bcast<11>(p) * q;
}
TOLG_DEV void rank1_bi_023(double (&acc)[12], double p, double q) {
  acc[0] += bcast<0>(p) * q; acc[1] += bcast<1>(p) * q; acc[2] += bcast<2>(p) * q;
  rank1_bi_23(acc, p, q);
}
TOLG_DEV void rank1_bk3_0(double (&a)[12], const double (&Pm)[12], double q0, double q1, double q2) { rank1_bk<0>(a, Pm, q0); rank1_bk<1>(a, Pm, q1); rank1_bk<2>(a, Pm, q2); }
TOLG_DEV void rank1_bk3_3(double (&a)[12], const double (&Pm)[12], double q0, double q1, double q2) { rank1_bk<3>(a, Pm, q0); rank1_bk<4>(a, Pm, q1); rank1_bk<5>(a, Pm, q2); }
TOLG_DEV void rank1_bk3_6(double (&a)[12], const double (&Pm)[12], double q0, double q1, double q2) { rank1_bk<6>(a, Pm, q0); rank1_bk<7>(a, Pm, q1); rank1_bk<8>(a, Pm, q2); }
TOLG_DEV void rank1_bk3_9(double (&a)[12], const double (&Pm)[12], double q0, double q1, double q2) { rank1_bk<9>(a, Pm, q0); rank1_bk<10>(a, Pm, q1); rank1_bk<11>(a, Pm, q2); }
TOLG_DEV void rank1_bi_02x3(double (&a)[12], const double (&p)[3], const double (&q)[3]) { for (int k = 0; k < 3; k++) rank1_bi_02(a, p[k], q[k]); }
TOLG_DEV void rank1_bi_x3(double (&a)[12], const double (&p)[3], const double (&q)[3]) { for (int k = 0; k < 3; k++) rank1_bi(a, p[k], q[k]); }
TOLG_DEV void rank1_bi_23x6(double (&a)[12], const double (&p)[6], const double (&q)[6]) { for (int k = 0; k < 6; k++) rank1_bi_23(a, p[k], q[k]); }
TOLG_DEV void rank1_bi_023x6(double (&a)[12], const double (&p)[6], const double (&q)[6]) { for (int k = 0; k < 6; k++) rank1_bi_023(a, p[k], q[k]); }
TOLG_DEV void rank1_bi_x6(double (&a)[12], const double (&p)[6], const double (&q)[6]) { for (int k = 0; k < 6; k++) rank1_bi(a, p[k], q[k]); }
TOLG_DEV void rank1_bi_x4(double (&a)[12], const double (&p)[4], const double (&q)[4]) { for (int k = 0; k < 4; k++) rank1_bi(a, p[k], q[k]); }
#define rank1_bk3_0_nn rank1_bk3_0
#define rank1_bk3_3_nn rank1_bk3_3
#define rank1_bk3_6_nn rank1_bk3_6
#define rank1_bk3_9_nn rank1_bk3_9
#define rank1_bi_02x3_nn rank1_bi_02x3
#define rank1_bi_x3_nn rank1_bi_x3
#define rank1_bi_23x6_nn rank1_bi_23x6
#define rank1_bi_023x6_nn rank1_bi_023x6
#define rank1_bi_x6_nn rank1_bi_x6
#define rank1_bi_x4_nn rank1_bi_x4
#endif

// Cholesky of the symmetric part of Q (in place: on return the lower triangle of Q holds L with the
// diagonal replaced by its reciprocal-free value, dinv the reciprocals).  Only the lower triangle
// and diagonal of Q are read; the strict upper triangle keeps the original entries for lu_solve.
// 1/sqrt(d) to double precision: v_rsq_f64 seed + two Goldschmidt/Newton refinements (no IEEE
// division or sqrt sequence on the Cholesky critical path)
TOLG_DEV double rsqrt_nr(double d) {
  double y = __builtin_amdgcn_rsq(d);
  double g = d * y, h = 0.5 * y;
  double r = fma(-h, g, 0.5);
  g = fma(g, r, g); h = fma(h, r, h);
  r = fma(-h, g, 0.5);
  h = fma(h, r, h);
  return 2.0 * h;
}
// Cholesky of the symmetric part of Q.  L's strict lower triangle and dinv[j] = 1/L[j][j] are
// produced (the diagonal itself is never needed by the solves).  Only Q[i][j], i >= j, is read.
template <int M>
TOLG_DEV bool chol_sym(double (&L)[M][M], const double (&Q)[M][M], double (&dinv)[M]) {
  bool ok = true;
#pragma unroll
  for (int j = 0; j < M; j++) {
    double d = Q[j][j];
#pragma unroll
    for (int k = 0; k < j; k++) d -= L[j][k] * L[j][k];
    ok = ok && (d > 0.0);
    dinv[j] = rsqrt_nr(ok ? d : 1.0);
#pragma unroll
    for (int i = j + 1; i < M; i++) {
      double s = Q[i][j];
#pragma unroll
      for (int k = 0; k < j; k++) s -= L[i][k] * L[j][k];
      L[i][j] = s * dinv[j];
    }
  }
  return ok;
}
template <int M>
TOLG_DEV void chol_solve(const double (&L)[M][M], const double (&dinv)[M], double (&x)[M]) {
#pragma unroll
  for (int i = 0; i < M; i++) {
    double s = x[i];
#pragma unroll
    for (int k = 0; k < i; k++) s -= L[i][k] * x[k];
    x[i] = s * dinv[i];
  }
#pragma unroll
  for (int i = M - 1; i >= 0; i--) {
    double s = x[i];
#pragma unroll
    for (int k = i + 1; k < M; k++) s -= L[k][i] * x[k];
    x[i] = s * dinv[i];
  }
}
// np.linalg.solve semantics (LU, partial pivoting) for the max-regularisation exit where Q_uu is
// not PD (traopt_controller.py:2983-2985, :2994-2995).  Branch-free compare-and-swap pivoting.
template <int M>
TOLG_DEV void lu_solve(double (&A)[M][M], double (&x)[M]) {
#pragma unroll
  for (int c = 0; c < M; c++) {
#pragma unroll
    for (int r = c + 1; r < M; r++) {
      bool sw = fabs(A[r][c]) > fabs(A[c][c]);
#pragma unroll
      for (int k = 0; k < M; k++) {
        double a = A[c][k], bb = A[r][k];
        A[c][k] = sw ? bb : a;
        A[r][k] = sw ? a : bb;
      }
      double a = x[c], bb = x[r];
      x[c] = sw ? bb : a;
      x[r] = sw ? a : bb;
    }
    double pinv = 1.0 / A[c][c];
#pragma unroll
    for (int r = c + 1; r < M; r++) {
      double f = A[r][c] * pinv;
#pragma unroll
      for (int k = c + 1; k < M; k++) A[r][k] -= f * A[c][k];
      x[r] -= f * x[c];
    }
  }
#pragma unroll
  for (int i = M - 1; i >= 0; i--) {
    double s = x[i];
#pragma unroll
    for (int k = i + 1; k < M; k++) s -= A[i][k] * x[k];
    x[i] = s / A[i][i];
  }
}

// ------------------------------------------------------------------------------------------------
// Q_uu (m x m, column c in lane c of the 16-lane row) is factored where it lies: right-looking L D L^T, pivot by
// pivot, the rank-1 update of the trailing columns one DPP-fused multiply-add per row (the multiplier of column c,
// Q[j][c] / d_j, is that lane's own).  Every lane then solves for its own right-hand side (its column of
// [Q_ux | Q_u]) with the factor entries broadcast from the lanes that hold them.  Compared with the former
// replicated Cholesky (21 broadcasts + a full factorisation in every lane) this is ~45 instructions and 70
// registers less per knot; the pivots are those of the Cholesky factor squared, so the positive-definiteness test
// (is_pos_def(Q_uu + Q_uu^T), traopt_utilis.py:320-329) is the same test: every pivot > 0.
// ------------------------------------------------------------------------------------------------
TOLG_DEV double rcp_nr(double d) {  // 1/d to double precision: v_rcp_f64 seed + two Newton steps
  double r = __builtin_amdgcn_rcp(d);
  double e = fma(-d, r, 1.0);
  r = fma(r, e, r);
  e = fma(-d, r, 1.0);
  return fma(r, e, r);
}
#ifndef TOLG_DPP_BUILTIN
#define DFA(acc, p, q, L) "v_fmac_f64_dpp " acc ", " p ", " q " row_newbcast:" L " row_mask:0xf bank_mask:0xf\n\t"
// a[i] += a[i]@lane J * w for the rows below pivot J
template <int M, int J>
TOLG_DEV void ldl_update(double (&a)[M], double w) {
  constexpr int R = M - 1 - J;
  if constexpr (R == 5)
    asm volatile("s_nop 1\n\t" DFA("%0", "%0", "%5", "%6") DFA("%1", "%1", "%5", "%6") DFA("%2", "%2", "%5", "%6")
                     DFA("%3", "%3", "%5", "%6") DFA("%4", "%4", "%5", "%6")
                 : "+v"(a[J + 1]), "+v"(a[J + 2]), "+v"(a[J + 3]), "+v"(a[J + 4]), "+v"(a[J + 5]) : "v"(w), "n"(J));
  if constexpr (R == 4)
    asm volatile("s_nop 1\n\t" DFA("%0", "%0", "%4", "%5") DFA("%1", "%1", "%4", "%5") DFA("%2", "%2", "%4", "%5")
                     DFA("%3", "%3", "%4", "%5")
                 : "+v"(a[J + 1]), "+v"(a[J + 2]), "+v"(a[J + 3]), "+v"(a[J + 4]) : "v"(w), "n"(J));
  if constexpr (R == 3)
    asm volatile("s_nop 1\n\t" DFA("%0", "%0", "%3", "%4") DFA("%1", "%1", "%3", "%4") DFA("%2", "%2", "%3", "%4")
                 : "+v"(a[J + 1]), "+v"(a[J + 2]), "+v"(a[J + 3]) : "v"(w), "n"(J));
  if constexpr (R == 2)
    asm volatile("s_nop 1\n\t" DFA("%0", "%0", "%2", "%3") DFA("%1", "%1", "%2", "%3")
                 : "+v"(a[J + 1]), "+v"(a[J + 2]) : "v"(w), "n"(J));
  if constexpr (R == 1)
    asm volatile("s_nop 1\n\t" DFA("%0", "%0", "%1", "%2") : "+v"(a[J + 1]) : "v"(w), "n"(J));
}
// y += sum_{k < I} p@lane k * q[k]     (row I of a forward substitution: p = this lane's copy of factor row I)
template <int M, int I>
TOLG_DEV void ldl_fwd_row(double& y, double p, const double (&q)[M]) {
  if constexpr (I == 1) asm volatile("s_nop 1\n\t" DFA("%0", "%1", "%2", "0") : "+v"(y) : "v"(p), "v"(q[0]));
  if constexpr (I == 2)
    asm volatile("s_nop 1\n\t" DFA("%0", "%1", "%2", "0") DFA("%0", "%1", "%3", "1") : "+v"(y) : "v"(p), "v"(q[0]), "v"(q[1]));
  if constexpr (I == 3)
    asm volatile("s_nop 1\n\t" DFA("%0", "%1", "%2", "0") DFA("%0", "%1", "%3", "1") DFA("%0", "%1", "%4", "2")
                 : "+v"(y) : "v"(p), "v"(q[0]), "v"(q[1]), "v"(q[2]));
  if constexpr (I == 4)
    asm volatile("s_nop 1\n\t" DFA("%0", "%1", "%2", "0") DFA("%0", "%1", "%3", "1") DFA("%0", "%1", "%4", "2")
                     DFA("%0", "%1", "%5", "3")
                 : "+v"(y) : "v"(p), "v"(q[0]), "v"(q[1]), "v"(q[2]), "v"(q[3]));
  if constexpr (I == 5)
    asm volatile("s_nop 1\n\t" DFA("%0", "%1", "%2", "0") DFA("%0", "%1", "%3", "1") DFA("%0", "%1", "%4", "2")
                     DFA("%0", "%1", "%5", "3") DFA("%0", "%1", "%6", "4")
                 : "+v"(y) : "v"(p), "v"(q[0]), "v"(q[1]), "v"(q[2]), "v"(q[3]), "v"(q[4]));
}
// t += sum_{k > I} a[k]@lane I * x[k]  (row I of the back substitution: column I of the factor lives in lane I)
template <int M, int I>
TOLG_DEV void ldl_bwd_row(double& t, const double (&a)[M], const double (&x)[M]) {
  constexpr int R = M - 1 - I;
  if constexpr (R == 1) asm volatile("s_nop 1\n\t" DFA("%0", "%1", "%2", "%3") : "+v"(t) : "v"(a[I + 1]), "v"(x[I + 1]), "n"(I));
  if constexpr (R == 2)
    asm volatile("s_nop 1\n\t" DFA("%0", "%1", "%3", "%5") DFA("%0", "%2", "%4", "%5")
                 : "+v"(t) : "v"(a[I + 1]), "v"(a[I + 2]), "v"(x[I + 1]), "v"(x[I + 2]), "n"(I));
  if constexpr (R == 3)
    asm volatile("s_nop 1\n\t" DFA("%0", "%1", "%4", "%7") DFA("%0", "%2", "%5", "%7") DFA("%0", "%3", "%6", "%7")
                 : "+v"(t) : "v"(a[I + 1]), "v"(a[I + 2]), "v"(a[I + 3]), "v"(x[I + 1]), "v"(x[I + 2]), "v"(x[I + 3]), "n"(I));
  if constexpr (R == 4)
    asm volatile("s_nop 1\n\t" DFA("%0", "%1", "%5", "%9") DFA("%0", "%2", "%6", "%9") DFA("%0", "%3", "%7", "%9")
                     DFA("%0", "%4", "%8", "%9")
                 : "+v"(t) : "v"(a[I + 1]), "v"(a[I + 2]), "v"(a[I + 3]), "v"(a[I + 4]), "v"(x[I + 1]), "v"(x[I + 2]),
                   "v"(x[I + 3]), "v"(x[I + 4]), "n"(I));
  if constexpr (R == 5)
    asm volatile("s_nop 1\n\t" DFA("%0", "%1", "%6", "%11") DFA("%0", "%2", "%7", "%11") DFA("%0", "%3", "%8", "%11")
                     DFA("%0", "%4", "%9", "%11") DFA("%0", "%5", "%10", "%11")
                 : "+v"(t) : "v"(a[I + 1]), "v"(a[I + 2]), "v"(a[I + 3]), "v"(a[I + 4]), "v"(a[I + 5]), "v"(x[I + 1]),
                   "v"(x[I + 2]), "v"(x[I + 3]), "v"(x[I + 4]), "v"(x[I + 5]), "n"(I));
}
#else
template <int M, int J>
TOLG_DEV void ldl_update(double (&a)[M], double w) {
#pragma unroll
  for (int i = J + 1; i < M; i++) a[i] += bcast<J>(a[i]) * w;
}
template <int M, int I>
TOLG_DEV void ldl_fwd_row(double& y, double p, const double (&q)[M]) {
  if constexpr (I > 0) y += bcast<0>(p) * q[0];
  if constexpr (I > 1) y += bcast<1>(p) * q[1];
  if constexpr (I > 2) y += bcast<2>(p) * q[2];
  if constexpr (I > 3) y += bcast<3>(p) * q[3];
  if constexpr (I > 4) y += bcast<4>(p) * q[4];
}
template <int M, int I>
TOLG_DEV void ldl_bwd_row(double& t, const double (&a)[M], const double (&x)[M]) {
#pragma unroll
  for (int k = I + 1; k < M; k++) t += bcast<I>(a[k]) * x[k];
}
#endif
// In place: on return a[i] (i > c) of lane c holds column c of the factor times its pivot (L[i][c] d_c), rinv[j] =
// 1 / d_j in every lane.  jl = lane index within the 16-lane row.  Returns "every pivot > 0".
template <int M, int J = 0>
TOLG_DEV bool ldl_factor(double (&a)[M], double (&rinv)[M], int jl, bool ok = true) {
  const double d = bcast<J>(a[J]);
  ok = ok && (d > 0.0);
  rinv[J] = rcp_nr(d);  // a non-positive pivot leaves garbage behind it: the caller discards the factors when !ok
  if constexpr (J + 1 < M) {
    // multiplier of this lane's column: -Q[J][c] / d_J for the columns right of the pivot, 0 for the others (their
    // rows below J are already final factor entries and must not move)
    const double w = (jl > J) ? -a[J] * rinv[J] : 0.0;
    ldl_update<M, J>(a, w);
    return ldl_factor<M, J + 1>(a, rinv, jl, ok);
  } else {
    return ok;
  }
}
// x <- (L D L^T)^-1 x for this lane's right-hand side
template <int M>
TOLG_DEV void ldl_solve(const double (&a)[M], const double (&rinv)[M], double (&x)[M]) {
  // forward, unit lower triangle: y_i = b_i - sum_{k<i} (a[i]@k / d_k) y_k; zn_k = -y_k / d_k is what the next stage needs too
  double zn[M];
  zn[0] = -x[0] * rinv[0];
  if constexpr (M > 1) { ldl_fwd_row<M, 1>(x[1], a[1], zn); zn[1] = -x[1] * rinv[1]; }
  if constexpr (M > 2) { ldl_fwd_row<M, 2>(x[2], a[2], zn); zn[2] = -x[2] * rinv[2]; }
  if constexpr (M > 3) { ldl_fwd_row<M, 3>(x[3], a[3], zn); zn[3] = -x[3] * rinv[3]; }
  if constexpr (M > 4) { ldl_fwd_row<M, 4>(x[4], a[4], zn); zn[4] = -x[4] * rinv[4]; }
  if constexpr (M > 5) { ldl_fwd_row<M, 5>(x[5], a[5], zn); zn[5] = -x[5] * rinv[5]; }
  // backward: x_i = y_i / d_i - (1 / d_i) sum_{k>i} a[k]@i x_k
  x[M - 1] = -zn[M - 1];
  if constexpr (M > 1) { double t = 0; ldl_bwd_row<M, M - 2>(t, a, x); x[M - 2] = -fma(rinv[M - 2], t, zn[M - 2]); }
  if constexpr (M > 2) { double t = 0; ldl_bwd_row<M, M - 3>(t, a, x); x[M - 3] = -fma(rinv[M - 3], t, zn[M - 3]); }
  if constexpr (M > 3) { double t = 0; ldl_bwd_row<M, M - 4>(t, a, x); x[M - 4] = -fma(rinv[M - 4], t, zn[M - 4]); }
  if constexpr (M > 4) { double t = 0; ldl_bwd_row<M, M - 5>(t, a, x); x[M - 5] = -fma(rinv[M - 5], t, zn[M - 5]); }
  if constexpr (M > 5) { double t = 0; ldl_bwd_row<M, M - 6>(t, a, x); x[M - 6] = -fma(rinv[M - 6], t, zn[M - 6]); }
}

// VARB: F_u differs from knot to knot (Pendulum3dDyanmics): its 3x3 block is read from REC_BU.
// GRAV: the model has a gravity block A21 in F_x (Drone / RigidBody / Pendulum); SE3 / SO3 instantiate
// without it and save its 18 per-lane coefficients (36 VGPRs in a kernel that already spills to AGPRs).
// DIAGJ: I_b and J_v are diagonal (every reference script): F_u's two 3x3 blocks are diagonal (the drone's
// J_v^-1 e_z column has one entry), so each input touches one row of (V + mu I) F_x instead of three.
// FAST (round 4): as for k_backward3 (tolg_backward3.h) -- only the first, unregularised attempt compiled in; a group that meets
// anything else is flagged in P.k2_redo and redone by the full kernel launched behind (flag bit 2), P.k2_hint keeps groups whose
// last sweep needed the retry loop away from the fast attempt.
template <int M, bool VARB = false, bool GRAV = true, bool DIAGJ = false, bool FAST = false>
__global__ __launch_bounds__(64) void k_backward(Params P, int it, int flags) {
  // flags: bit 0 multiple shooting; bit 1 the records come from the fused rollout, whose trajectories are closed
  // (x_{i+1} = f(x_i, u_i)): the defect field is not written there and reads as zero here; bit 2: only the groups the fast
  // kernel handed back
  if ((flags & 4) && !P.k2_redo[blockIdx.x]) return;
  const int ms = flags & 1;
  const bool closed = (flags & 2) != 0;
  const DConsts& C = *(const DConsts*)P.c;
  const int lane = threadIdx.x, g = lane >> 4, j = lane & 15;
  const int b = blockIdx.x * 4 + g;  // Bp is a multiple of 4
  const bool act = P.active[b] != 0;
  if constexpr (FAST) {
    const bool back = __any(act && P.mu[b] != 0.0) || P.k2_hint[blockIdx.x] != 0;
    if (lane == 0) P.k2_redo[blockIdx.x] = back ? 1 : 0;
    if (back) return;
  }
  bool failed = false;  // FAST: a knot the first attempt did not settle; full kernel: the retry loop ran somewhere (-> k2_hint)
  if (!__any(act)) return;
  const int N = P.N;
  // (FAST: 252 registers would let two workgroups share a SIMD, and the dispatcher then packs some SIMDs with two waves and
  // leaves others idle -- the dense-inertia SS bench ran this kernel at 0.62 ms against 0.41 with an even spread; the transpose
  // scratch is over-allocated to 35 KB so that a CU takes four workgroups, one per SIMD)
  __shared__ double TR[FAST ? 28 : 4][12 * 13];
  __shared__ double TRZ[12];  // zeros: what the vector lanes "transpose-read" (see the symmetrisation below)

  // lane-dependent constants
  const double m12 = (j < 12) ? 1.0 : 0.0;                  // matrix columns
  const double mW2 = (j >= 6 && j < 12) ? 1.0 : 0.0;        // columns of the velocity block (2 W2 lives there)
  const double mvec = (j == 12 || j == 13) ? 1.0 : 0.0;     // vector columns (V_x, SS adjoint)
  constexpr bool grav = GRAV;
  // Per-lane constant columns (2 W2, 2 R, and the two views of F_u).  They live in LDS, not in 48 VGPRs:
  // the kernel is far over the 256 architectural registers and every value parked in an AGPR costs a
  // v_accvgpr copy per use, while LDS is otherwise idle here.  Row stride 25 doubles: conflict-free.
  __shared__ double KC[16][31];
  // Lanes 0..M-1 and lanes 6..11 need different constants and never look at each other's: one slot
  // holds 2R (lanes < M) or the row view of F_u (lanes 6..11), the other the column view of F_u (lanes
  // < M) or 2 W2 (lanes 6..11).  Where a lane reads the "wrong" half the result is never used, except
  // in the l_xx update, which masks it.
  enum { KC_RB = 0, KC_BW = 6, KC_BT = 12, KC_BB = 21, KC_BD = 30 };  // KC_BT/BB: same in every row; KC_BD: B[6+j][j]
  if (g == 0) {
#pragma unroll
    for (int r = 0; r < 6; r++) {
      KC[j][KC_BW + r] = (j >= 6 && j < 12) ? 2.0 * C.W2[6 * r + (j - 6)] : (j < M) ? fu_entry<M>(*P.c, r, j) : 0.0;  // B[6+r][j]
    }
#pragma unroll
    for (int u = 0; u < 6; u++) {
      KC[j][KC_RB + u] = (u < M && j < M) ? 2.0 * C.R[(u < M ? u : 0) * M + j]
                         : (u < M && j >= 6 && j < 12) ? fu_entry<M>(*P.c, j - 6, u < M ? u : 0) : 0.0;  // B[j][u]
    }
#pragma unroll
    for (int k = 0; k < 9; k++) { KC[j][KC_BT + k] = P.c->Bt[k]; KC[j][KC_BB + k] = P.c->Bb[k]; }
    KC[j][KC_BD] = (j < M) ? fu_entry<M>(*P.c, j < 6 ? j : 0, j) : 0.0;
  }
  if (lane < 12) TRZ[lane] = 0.0;
  __builtin_amdgcn_wave_barrier();
  const double* KCj = KC[j];
  // V <- (V + V^T) / 2 for the matrix columns, V <- V for the vector columns, without a per-entry select: a vector
  // lane reads zeros for the transposed entry and scales by 1 instead of 1/2 (bitwise what the select gave)
  const double* trd = (j < 12) ? &TR[g][j * 13] : TRZ;
  const double hsym = (j < 12) ? 0.5 : 1.0;
  const double kneg = (j == 13) ? 0.0 : -1.0;  // gains are -Q_uu^-1 [Q_ux | Q_u]; the adjoint lane gets none
  // which record fields make up column j of [F_x | d] (rows 0..2, 3..5) and of [l_xx | l_x]
  int fT = REC_D, fM = REC_D + 3;
  double mT = 0.0, mM = 0.0;
  if (j < 3) { fT = REC_RI + 3 * j; fM = REC_TRI + 3 * j; mT = 1; mM = VARB ? 0 : 1; }  // VARB: the slot holds REC_BU
  else if (j < 6) { fM = REC_RI + 3 * (j - 3); mM = 1; }
  else if (j < 9) { fT = REC_JR + 3 * (j - 6); fM = REC_QR + 3 * (j - 6); mT = 1; mM = 1; }
  else if (j < 12) { fM = REC_JR + 3 * (j - 9); mM = 1; }
  else if (j == 12 && !closed) { mT = 1; mM = 1; }
  int fL[6];
#pragma unroll
  for (int r = 0; r < 6; r++) fL[r] = (j < 6) ? REC_LXX + sym6(r, j) : REC_LX + r;
  const double mLT = (j < 6 || j == 12 || j == 13) ? 1.0 : 0.0;
  // rows 6..11 of column j: lanes 6..11 read their A22 column, lane 12 reads d[6:12], lanes 0..2
  // (gravity models) build their A21 column from R^T e3 with per-lane constants
  const int fB = (j >= 6 && j < 12) ? P.fA22 + 6 * (j - 6) : REC_D + 6;  // (this sweep's models store the block)
  const unsigned sB = (unsigned)P.Bp * 8u, vb = (unsigned)b * 8u;
  const unsigned vr = REC_VR(b);
  // Lanes for which a block of the column is structurally zero load it from an out-of-range offset: raw
  // buffer loads return 0 there, which replaces thirty mask multiplications per knot (x * 1.0 was exact).
  const unsigned OOB = 0x40000000u;
  unsigned oT[3], oM[3];  // fT / fM are odd for every second lane: per-row offsets instead of base + r
#pragma unroll
  for (int r = 0; r < 3; r++) { oT[r] = (mT != 0.0) ? vr + FOFF(fT + r) : OOB; oM[r] = (mM != 0.0) ? vr + FOFF(fM + r) : OOB; }
  const bool hasB = (j >= 6 && j < 12) || (j == 12 && !closed), isVec = (j == 12 || j == 13);
  const unsigned vBt = hasB ? vr + FOFF(fB) : OOB;  // fB is even (REC_A22, REC_D even): rows r, r+1 are one 16-byte pair
  const unsigned vVec = isVec ? vr : OOB;           // fields only the vector columns read (l_x[6:12], l_u)
  const unsigned vG = GK_VG(b, M) + GOFF(0, (j < 13 ? j : 12), M);
  // the l_uu field exists in AL solves only: without AL every lane reads it out of range (= 0), branch-free
  const unsigned vUU = (P.al_lb != nullptr) ? vr + FOFF(P.fLUU + (j < M ? j : 0)) : OOB;
  unsigned vL[6];
#pragma unroll
  for (int r = 0; r < 6; r++) vL[r] = (mLT != 0.0) ? vr + FOFF(fL[r]) : OOB;
  const size_t recStride = (size_t)P.recF * P.Bp, uStride = (size_t)M * P.Bp, gStride = (size_t)13 * M * P.Bp;
  double Cg[GRAV ? 3 : 1][6];
  if constexpr (GRAV) {
#pragma unroll
    for (int a = 0; a < 3; a++)
#pragma unroll
      for (int r = 0; r < 6; r++) Cg[a][r] = (j < 3) ? C.Llin[a][6 * r + (j < 3 ? j : 0)] : 0.0;
  }
  double mu = P.mu[b], delta = P.delta[b];
  int warned = 0;

  // terminal condition: V = [l_xx(N) | l_x(N)] with P weights (traopt_controller.py:2956-2957)
  double V[12];
  {
    __amdgpu_buffer_rsrc_t rR = mkbuf(P.REC + recStride * N, (unsigned)P.recF * sB);
#pragma unroll
    for (int r = 0; r < 6; r++) {
      double t1 = bld(rR, vL[r], 0), t2 = bld(rR, vVec, FOFF(REC_LX + 6 + r));
      V[r] = t1;
      double p2 = (j >= 6 && j < 12) ? 2.0 * C.P2[6 * r + (j - 6)] : 0.0;
      V[6 + r] = t2 + p2;
    }
  }
  double gsum = 0;
#ifdef TOLG_STAMPS
  unsigned long long st_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0}, st_t = __builtin_amdgcn_s_memtime();
#define STAMP(k) { __builtin_amdgcn_sched_barrier(0); unsigned long long t_ = __builtin_amdgcn_s_memtime(); st_acc[k] += t_ - st_t; st_t = t_; __builtin_amdgcn_sched_barrier(0); }
#else
#define STAMP(k)
#endif

  // Raw loads of one knot (column j of [F_x | d], of [l_xx | l_x], the controls), issued one knot
  // ahead of their use.  Nothing here may consume a loaded value: that would put the wait for the
  // data right behind the request and undo the prefetch.
  struct BwdIn { double t[3], m[3], bt[6], g[GRAV ? 3 : 1], lt[6], lb[6], lu[M], luu, bu[VARB ? 9 : 1]; };
  auto load_knot = [&](int i, BwdIn& in) {
    __amdgpu_buffer_rsrc_t rR = mkbuf(P.REC + recStride * i, (unsigned)P.recF * sB);
#pragma unroll
    for (int r = 0; r < 3; r++) {
      in.t[r] = bld(rR, oT[r], 0);
      in.m[r] = bld(rR, oM[r], 0);
    }
#pragma unroll
    for (int r = 0; r < 6; r += 2) bld2(rR, vBt, FOFF(r), in.bt[r], in.bt[r + 1]);
    if constexpr (GRAV) {
#pragma unroll
      for (int a = 0; a < 3; a++) in.g[a] = bld(rR, REC_VR(b), FOFF(REC_LU + M + a));
    }
#pragma unroll
    for (int r = 0; r < 6; r++) in.lt[r] = bld(rR, vL[r], 0);
#pragma unroll
    for (int r = 0; r < 6; r += 2) bld2(rR, vVec, FOFF(REC_LX + 6 + r), in.lb[r], in.lb[r + 1]);
#pragma unroll
    for (int a = 0; a < M; a += 2) bld2(rR, vVec, FOFF(REC_LU + a), in.lu[a], in.lu[a + 1]);
    in.luu = bld(rR, vUU, 0);  // lane u < M: the AL addition to l_uu[u][u]
    if constexpr (VARB) {
#pragma unroll
      for (int k = 0; k < 9; k++) in.bu[k] = bld(rR, REC_VR(b), FOFF(REC_BU + k));
    }
  };

  // The gains of knot i are stored half a step late, right before the prefetch of step i-1 is issued:
  // vmcnt is in-order, and with the stores issued at the end of a step the s_waitcnt vmcnt(0) at the
  // top of the next one (prefetched fields) also waited for their write acknowledgements.
  double Kst[M];
#pragma unroll
  for (int u = 0; u < M; u++) Kst[u] = 0;
  auto store_gains = [&](int knot) {
    if (act && j < 13) {
      __amdgpu_buffer_rsrc_t rGs = mkbuf(P.GK + gStride * knot, 13 * M * sB);
#pragma unroll
      for (int u = 0; u < M; u += 2) bst2(rGs, vG, GOFF(u, 0, M), Kst[u], Kst[u + 1]);
    }
  };
  auto step = [&](int i, BwdIn& in) {
    double A[12], Lc[12], lu[M];
    const double luu_i = in.luu;
#pragma unroll
    for (int r = 0; r < 3; r++) { A[r] = in.t[r]; A[3 + r] = in.m[r]; }
#pragma unroll
    for (int r = 0; r < 6; r++) A[6 + r] = in.bt[r];
    if constexpr (GRAV) {
#pragma unroll
      for (int r = 0; r < 6; r++) A[6 + r] += in.g[0] * Cg[0][r] + in.g[1] * Cg[1][r] + in.g[2] * Cg[2][r];
    }
#pragma unroll
    for (int r = 0; r < 6; r++) { Lc[r] = in.lt[r]; Lc[6 + r] = in.lb[r]; }  // + 2 W2: added to Qh below
#pragma unroll
    for (int a = 0; a < M; a++) lu[a] = in.lu[a];  // l_u = 2 R u rides in the vector columns (zero elsewhere)
    STAMP(0)
    // ---- Z = V [F_x | d]  (+ V_x in the vector column -> w = V_x + V_xx d; adjoint passes through)
    double Z[12];
#pragma unroll
    for (int r = 0; r < 12; r++) Z[r] = mvec * V[r];
    // for the adjoint lane (13) A = 0 so Z stays p; for lane 12 A = d.  V_x must not feed the
    // broadcast side: only lanes 0..11 are ever sources of rank1_bk.
    rank1_bk3_0(Z, V, A[0], A[1], A[2]); rank1_bk3_3(Z, V, A[3], A[4], A[5]);
    rank1_bk3_6(Z, V, A[6], A[7], A[8]); rank1_bk3_9(Z, V, A[9], A[10], A[11]);
    STAMP(1)
    // ---- Qh = [l_xx | l_x] + F_x^T Z
    double Qh[12];
#pragma unroll
    for (int r = 0; r < 12; r++) Qh[r] = Lc[r];
    // F_x = [Ri 0 Jr 0; TRi Ri Qr Jr; A21 0 A22 A22]: skip the structurally zero 3-row blocks
    {
      const double a0[3] = {A[0], A[1], A[2]}, z0[3] = {Z[0], Z[1], Z[2]}, a1[3] = {A[3], A[4], A[5]}, z1[3] = {Z[3], Z[4], Z[5]};
      const double a2[6] = {A[6], A[7], A[8], A[9], A[10], A[11]}, z2[6] = {Z[6], Z[7], Z[8], Z[9], Z[10], Z[11]};
      rank1_bi_02x3(Qh, a0, z0);
      rank1_bi_x3(Qh, a1, z1);
      if constexpr (GRAV) rank1_bi_023x6(Qh, a2, z2);
      else rank1_bi_23x6(Qh, a2, z2);
    }

    // The raw fields of this knot are consumed (A, Lc, lu, luu_i live in their own registers from here
    // on): request the next knot's fields into the same buffer now, so they arrive while the
    // regularisation / Cholesky / gain / V-update half of the step runs.  One buffer instead of a
    // ping-pong pair keeps the loads in VGPRs (with two, the allocator parked them in AGPRs and had to
    // wait for them right away to copy them back).
    STAMP(2)
    // Constants of the second half, requested BEFORE the prefetch so that their LDS / scalar-cache
    // latency passes while the 31 buffer loads issue (read where they are used, each cost an exposed
    // lgkmcnt(0) wait: ~10 per knot).
    double kBW[6], kRB[M], cBt[9], cBb[9];
    const double kBd = KCj[KC_BD];
#pragma unroll
    for (int r = 0; r < 6; r++) kBW[r] = KCj[KC_BW + r];
#pragma unroll
    for (int u = 0; u < M; u++) kRB[u] = KCj[KC_RB + u];
#pragma unroll
    for (int k = 0; k < 9; k++) {
      if (DIAGJ && k != 0 && k != 4 && k != 8 && !(M == 4 && k == 6)) { cBt[k] = 0; cBb[k] = 0; continue; }
      cBt[k] = KCj[KC_BT + k]; cBb[k] = KCj[KC_BB + k];
    }
    // this knot's input matrix (VARB): the 3x3 block of REC_BU for inputs 0..2 in place of the constants
    double BtS[VARB ? 9 : 1], BlocS[VARB ? 6 : 1], BrowS[VARB ? M : 1];
    if constexpr (VARB) {
#pragma unroll
      for (int k = 0; k < 9; k++) BtS[k] = in.bu[k];
#pragma unroll
      for (int r = 0; r < 6; r++) BlocS[r] = kBW[r];
#pragma unroll
      for (int u = 0; u < M; u++) BrowS[u] = kRB[u];
#pragma unroll
      for (int r = 0; r < 3; r++) BlocS[r] = (j == 0) ? BtS[3 * r] : (j == 1) ? BtS[3 * r + 1] : (j == 2) ? BtS[3 * r + 2] : 0.0;
#pragma unroll
      for (int u = 0; u < 3; u++) BrowS[u] = (j == 6) ? BtS[u] : (j == 7) ? BtS[3 + u] : (j == 8) ? BtS[6 + u] : 0.0;
    }
    __builtin_amdgcn_sched_barrier(0);
    if (i < N - 1) store_gains(i + 1);
    if (i > 0) load_knot(i - 1, in);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int r = 0; r < 6; r++) Qh[6 + r] += mW2 * kBW[r];
    STAMP(3)
    // ---- regularised Q_ux | Q_u, Q_uu; PD test; gains   (traopt_controller.py:2964-2995, :3052-3060)
    double Quh[M], Kh[M], Uf[M], rinv[M];
    bool use_lu = false;
    double mu_used = 0.0;  // the mu of the last attempt (the max-regularisation exit solves with that attempt's matrices)
    // Q_ux | Q_u (Quh) and Q_uu (column per lane) for a given mu
    auto build_quu = [&](double mu_, double (&Quu)[M]) {
      const double muA = m12 * mu_;
      // X' = (V + mu I) F_x, rows 6..11 (B has no other non-zero rows); vector columns: w
      double Xp[6];
#pragma unroll
      for (int k = 0; k < 6; k++) Xp[k] = Z[6 + k] + muA * A[6 + k];
      // T = B^T (V + mu I), rows u, column per lane
      double T[M];
#pragma unroll
      for (int u = 0; u < M; u++) {  // B2 is block diagonal: inputs 0..2 see rows 6..8, the rest rows 9..11
        double s = lu[u], tt = muA * (VARB ? BrowS[VARB ? u : 0] : kRB[u]);
        const int k0 = (u < 3) ? 0 : 3;
#pragma unroll
        for (int k = 0; k < 3; k++) {
          // the only row a diagonal block couples to input u (the dense pendulum block keeps all three)
          if (DIAGJ && !(VARB && u < 3) && k != ((u < 3) ? u : (M == 6 ? u - 3 : 2))) continue;
          const double bku = (u < 3) ? (VARB ? BtS[VARB ? 3 * k + u : 0] : cBt[3 * k + u]) : cBb[3 * k + (u - 3)];
          s += bku * Xp[k0 + k];
          tt += bku * V[6 + k0 + k];
        }
        Quh[u] = s;
        T[u] = m12 * tt;
      }
      // Q_uu = 2R + T B: column per lane (lanes 0..M-1): Quu[u] += T[u]@lane(6+k) * B[6+k][lane]
#pragma unroll
      for (int u = 0; u < M; u++) Quu[u] = kRB[u] + ((j == u) ? luu_i : 0.0);
      if constexpr (DIAGJ && M == 6 && !VARB) {
        // diagonal F_u blocks: column j of B has the single entry B[6+j][j], so (T B)[u][j] is T[u] of lane
        // j + 6 times that entry -- one row shift (DPP row_shl:6) and one FMA per input instead of six
        // broadcast blocks
#pragma unroll
        for (int u = 0; u < M; u++) Quu[u] = fma(row_shl6(T[u]), kBd, Quu[u]);
      } else if constexpr (VARB) {
        quu_acc<M, 6>(Quu, T, BlocS[0]); quu_acc<M, 7>(Quu, T, BlocS[1]); quu_acc<M, 8>(Quu, T, BlocS[2]);
        quu_acc<M, 9>(Quu, T, BlocS[3 % (VARB ? 6 : 1)]); quu_acc<M, 10>(Quu, T, BlocS[4 % (VARB ? 6 : 1)]);
        quu_acc<M, 11>(Quu, T, BlocS[5 % (VARB ? 6 : 1)]);
      } else {
        quu_acc<M, 6>(Quu, T, kBW[0]); quu_acc<M, 7>(Quu, T, kBW[1]); quu_acc<M, 8>(Quu, T, kBW[2]);
        quu_acc<M, 9>(Quu, T, kBW[3]); quu_acc<M, 10>(Quu, T, kBW[4]); quu_acc<M, 11>(Quu, T, kBW[5]);
      }
    };
    // one regularisation attempt with the current mu: the matrices above, Q_uu factored in place (L D L^T)
    auto attempt = [&]() -> bool {
      mu_used = mu;
      build_quu(mu, Uf);
      return ldl_factor<M>(Uf, rinv, j);
    };
    // regularisation schedule (traopt_controller.py:2975-2995); returns true when this knot is settled
    auto schedule = [&](bool pd) -> bool {
      if (!pd) {
        delta = fmax(1.0, delta) * 2.0;
        mu = fmax(1e-6, mu * delta);
        if (P.max_reg > 0 && mu >= P.max_reg) { warned = 1; use_lu = true; return true; }
        return false;
      }
      delta = fmin(1.0, delta) * 0.5;
      mu *= delta;
      if (mu <= 1e-6) mu = 0.0;
      return true;
    };
    // The first attempt runs for every lane, straight-line (inactive trajectories compute on stale data
    // and ignore the outcome): as the body of a retry loop it cost ~50 register initialisations per knot
    // for the loop-carried factors.  The retry loop itself is entered only when some active trajectory
    // of this wavefront failed the PD test.
    bool done = true;
    if constexpr (FAST) {
      // mu = 0 on entry and for as long as every pivot is positive (schedule(true) keeps it there); a non-positive pivot
      // anywhere in the wave ends the sweep behind this step (no branch in mid-step) and the full kernel redoes the group
      const bool pd = attempt();
      failed = failed || __any(act && !pd);
      if (act) { delta = fmin(1.0, delta) * 0.5; mu = 0.0; }
    } else {
      {
        const bool pd = attempt();
        if (act) done = schedule(pd);
      }
      if (!__all(done)) {
        failed = true;
        for (;;) {
          if (!done) done = schedule(attempt());
          if (__all(done)) break;
        }
      }
    }
    STAMP(4)
    // gradient term: ||Q_u|| in the MS vector lane, ||l_u + F_u^T p|| in the SS adjoint lane
    {
      double s = 0;
#pragma unroll
      for (int u = 0; u < M; u++) s += Quh[u] * Quh[u];
      gsum += (s > 0.0) ? s * rsqrt_nr(s) : 0.0;  // sqrt(s) without the IEEE sqrt sequence (2 ulp)
    }
    // gains: [K | k] = -Q_uu^-1 [Q_ux | Q_u]; the adjoint lane gets none (kneg: -1, or 0 in that lane)
#pragma unroll
    for (int u = 0; u < M; u++) Kh[u] = Quh[u];
    if (!FAST && __any(use_lu)) {
      // max-regularisation exit with a non-PD Q_uu: np.linalg.solve semantics on the matrix itself (rare path:
      // rebuild it -- the factorisation ran in place -- and replicate it to every lane)
      double Kl[M], Ac[M][M], Qc[M];
      build_quu(mu_used, Qc);
#pragma unroll
      for (int u = 0; u < M; u++) {
        Kl[u] = Kh[u];
        Ac[u][0] = bcast<0>(Qc[u]); Ac[u][1] = bcast<1>(Qc[u]); Ac[u][2] = bcast<2>(Qc[u]); Ac[u][3] = bcast<3>(Qc[u]);
        if constexpr (M > 4) { Ac[u][4] = bcast<4>(Qc[u]); Ac[u][5] = bcast<5>(Qc[u]); }
      }
      lu_solve<M>(Ac, Kl);
      ldl_solve<M>(Uf, rinv, Kh);
#pragma unroll
      for (int u = 0; u < M; u++) Kh[u] = use_lu ? Kl[u] : Kh[u];
    } else {
      ldl_solve<M>(Uf, rinv, Kh);
    }
#pragma unroll
    for (int u = 0; u < M; u++) Kh[u] = kneg * Kh[u];
#pragma unroll
    for (int u = 0; u < M; u++) Kst[u] = Kh[u];
    STAMP(5)
    // ---- V <- Qh + Q_ux^T [K | k]   (== Eq. 11b/11c of traopt_controller.py:2998-3003 for the
    // exact gains), then symmetrise the matrix columns through LDS (traopt_controller.py:3004)
    double Vn[12];
#pragma unroll
    for (int r = 0; r < 12; r++) Vn[r] = Qh[r];
    if constexpr (M == 6) rank1_bi_x6(Vn, Quh, Kh);
    else rank1_bi_x4(Vn, Quh, Kh);
    // One wavefront per workgroup: LDS operations of a wave execute in order, so the transpose
    // needs no s_barrier -- and must not use __syncthreads(), whose vmcnt(0) would drain the
    // prefetched loads and the gain stores twice per knot.  wave_barrier only pins the compiler.
    __builtin_amdgcn_wave_barrier();
    if (j < 12) {
#pragma unroll
      for (int r = 0; r < 12; r++) TR[g][r * 13 + j] = Vn[r];
    }
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int r = 0; r < 12; r++) V[r] = hsym * (Vn[r] + trd[r]);
    STAMP(6)
  };

  BwdIn in;
  load_knot(N - 1, in);
  for (int i = N - 1; i >= 0 && !(FAST && failed); i--) step(i, in);
  if constexpr (FAST) {
    if (failed) {
      if (lane == 0) { P.k2_redo[blockIdx.x] = 1; P.k2_hint[blockIdx.x] = 8; }
      return;
    }
  } else {
    // (the hint counts down from eight after a sweep with a non-positive pivot: note in tolg_backward3.h)
    if (lane == 0) { const int hn = (it == 0) ? 0 : P.k2_hint[blockIdx.x]; P.k2_hint[blockIdx.x] = failed ? 8 : (hn > 0 ? hn - 1 : 0); }
  }
  store_gains(0);
#ifdef TOLG_STAMPS
  STAMP(7)
  if (blockIdx.x == 7 && lane == 0 && P.mu_hist) { for (int k = 0; k < 8; k++) P.mu_hist[(size_t)28 * P.max_iter + k] = (double)st_acc[k]; }
#endif
  // ---- epilogue: gradient norm, convergence test (traopt_controller.py:2527-2532, :1937-1942)
  double grad = (ms ? bcast<12>(gsum) : bcast<13>(gsum)) / (double)N;
  if (act && j == 0) {
    P.mu[b] = mu;
    P.delta[b] = delta;
    P.grad[b] = grad;
    if (warned) P.status[b] = TOLG_ST_MAXREG;
    if (it >= 0 && b < P.B) {
      if (P.grad_hist) P.grad_hist[(size_t)b * (P.max_iter + 1) + it] = grad;
      if (P.mu_hist && it < P.max_iter) P.mu_hist[(size_t)b * P.max_iter + it] = mu;
    }
    if (it >= 0) {
      bool conv = ms ? (grad < P.tol_grad && P.dn[b] < P.tol_defect) : (grad < P.tol_grad);
      if (conv) { P.conv[b] = 1; P.active[b] = 0; }
    }
  }
}

// ------------------------------------------------------------------------------------------------
// K3: closed-loop rollout, one thread per trajectory (traopt_controller.py:2641-2740 MS,
// :2030-2082 SS).  The reference's MS step
//     q^_{i+1} = q_{i+1} Exp(alpha d_q) f_q(x_i,u_i)^-1 f_q(x^_i,u^_i),
//     xi^_{i+1} = xi_{i+1} + f_xi(x^_i,u^_i) - f_xi(x_i,u_i) + alpha d_xi
// is M_i o f_q(x^_i,u^_i), c_i + f_xi(x^_i,u^_i) with (M_i, c_i) independent of the new trajectory.
// For alpha = 1 (and for single shooting) M_i = x_{i+1} Exp(Log(x_{i+1}^-1 f_q)) f_q^-1 is the identity
// and c_i is zero up to rounding, so the sequential chain holds one Log, one Exp and the K dx product
// per knot; the alpha < 1 line-search steps build the factors from the stored defect.
// ------------------------------------------------------------------------------------------------
template <int M, class CT>
TOLG_DEV void fx_apply(const Params& P, const CT& C, int i, int b, const double (&e)[12],
                       const double (&du)[M], double (&lin)[12]) {
  // lin = F_x e + F_u du from the compact record (rollout == 'linear')
  double Ri[9], TRi[9], Jr[9], Qr[9];
#pragma unroll
  for (int k = 0; k < 9; k++) {
    Ri[k] = P.REC[RIDX(i, REC_RI + k, b)];
    TRi[k] = (C.kind == TOLG_DYN_PENDULUM3D) ? 0.0 : P.REC[RIDX(i, REC_TRI + k, b)];  // pendulum: the slot is REC_BU
    Jr[k] = P.REC[RIDX(i, REC_JR + k, b)]; Qr[k] = P.REC[RIDX(i, REC_QR + k, b)];
  }
#pragma unroll
  for (int r = 0; r < 3; r++) {  // blocks are column-major: X[r][c] = X_[3c + r]
    double t = 0, m = 0;
#pragma unroll
    for (int c = 0; c < 3; c++) {
      t += Ri[3 * c + r] * e[c] + Jr[3 * c + r] * e[6 + c];
      m += TRi[3 * c + r] * e[c] + Ri[3 * c + r] * e[3 + c] + Qr[3 * c + r] * e[6 + c] + Jr[3 * c + r] * e[9 + c];
    }
    lin[r] = t;
    lin[3 + r] = m;
  }
  double rte[3] = {0, 0, 0};  // the field exists for gravity models only
  if (C.grav != 0.0) { rte[0] = P.REC[RIDX(i, REC_LU + M, b)]; rte[1] = P.REC[RIDX(i, REC_LU + M + 1, b)]; rte[2] = P.REC[RIDX(i, REC_LU + M + 2, b)]; }
  // velocity block times e[6:12].  Where the record holds it (dense inertia, pendulum) it is read entry by entry;
  // where it does not (diagonal inertia: every reference script) it is applied WITHOUT being formed -- with
  // J = blkdiag(Ib, Jv), (coadjoint([v, w]) J + G) [a; c] = [(Ib w) x a - v x (Ib a) + m v x c - w x (Jv c);
  // m v x a - v x (Jv c)] (a22_build), six cross products instead of a 36-entry matrix in registers: the linear
  // rollouts of the merit search (k_expected_change, the LINEAR line-search rollouts) spilled 160-470 registers to scratch with
  // the matrix form, 4.2 ms per k_expected_change call at 4096 x 200.
  double he[6] = {0, 0, 0, 0, 0, 0};
  if (P.fA22 < 0) {
    const V3 w = v3(P.REC[RIDX(i, REC_XI, b)], P.REC[RIDX(i, REC_XI + 2, b)], P.REC[RIDX(i, REC_XI + 4, b)]);
    const V3 v = v3(P.REC[RIDX(i, REC_XI + 1, b)], P.REC[RIDX(i, REC_XI + 3, b)], P.REC[RIDX(i, REC_XI + 5, b)]);
    const V3 a = v3(e[6], e[7], e[8]), c = v3(e[9], e[10], e[11]);
    V3 top, bot = v3(0, 0, 0);
    if (so3_family(C.kind)) {
      top = cross(mv33(C.Ib, w), a) - cross(w, mv33(C.Ib, a));
    } else {
      top = (cross(mv33(C.Ib, w), a) - cross(v, mv33(C.Ib, a))) + (C.mass * cross(v, c) - cross(w, mv33(C.Jv, c)));
      bot = C.mass * cross(v, a) - cross(v, mv33(C.Jv, c));
    }
    const V3 ht = mv33(C.Ibinv, top), hb = mv33(C.Jvinv, bot);
    he[0] = C.dt * ht.x; he[1] = C.dt * ht.y; he[2] = C.dt * ht.z; he[3] = C.dt * hb.x; he[4] = C.dt * hb.y; he[5] = C.dt * hb.z;
  }
#pragma unroll
  for (int r = 0; r < 6; r++) {
    double sacc = e[6 + r] + he[r];
    if (P.fA22 >= 0) {
#pragma unroll
      for (int c = 0; c < 6; c++) sacc += (P.REC[RIDX(i, P.fA22 + 6 * c + r, b)] - (r == c ? 1.0 : 0.0)) * e[6 + c];
    }
#pragma unroll
    for (int c = 0; c < 3; c++) {
      double l = rte[0] * C.Llin[0][6 * r + c] + rte[1] * C.Llin[1][6 * r + c] + rte[2] * C.Llin[2][6 * r + c];
      sacc += l * e[c];
    }
#pragma unroll
    for (int k = 0; k < M; k++) {
      double fu = fu_entry<M>(C, r, k);
      if (C.kind == TOLG_DYN_PENDULUM3D && r < 3 && k < 3) fu = P.REC[RIDX(i, REC_BU + 3 * r + k, b)];
      sacc += fu * du[k];
    }
    lin[6 + r] = sacc;
  }
}

// What one rollout step reads.  Four lanes share a trajectory: every lane runs the (scalar) Log / Exp
// chain redundantly, but the gain product K dx -- the bulk of the loads and FMAs -- is split by rows
// (lane q owns rows q and q + 4) and the resulting du is broadcast inside the quad with DPP
// quad_perm.  The nominal state (needed first, by Log) is fetched one knot ahead (ping-pong
// registers); gains, controls and the rollout factors are requested at the top of the step and arrive
// while Log runs.  A few hundred waves cannot hide memory latency any other way.
template <int M>
struct RollIn {
  double G[2][13];   // gain rows 2q and 2q + 1 of [K | k] (one 16-byte pair per column)
  double u[M];
};
template <int L>
TOLG_DEV double quad_bcast(double x) {  // value of x in lane L of this lane's quad
  // on the two 32-bit halves: only row_newbcast exists as a 64-bit DPP move, and the type-generic builtin
  // applied to a double has been seen to convert its result numerically (see row_shl6)
  constexpr int ctrl = L | (L << 2) | (L << 4) | (L << 6);
  const unsigned long long u = __builtin_bit_cast(unsigned long long, x);
  const unsigned lo = (unsigned)__builtin_amdgcn_update_dpp(0, (int)(unsigned)u, ctrl, 0xf, 0xf, false);
  const unsigned hi = (unsigned)__builtin_amdgcn_update_dpp(0, (int)(unsigned)(u >> 32), ctrl, 0xf, 0xf, false);
  return __builtin_bit_cast(double, ((unsigned long long)hi << 32) | lo);
}
TOLG_DEV State roll_load_state(const Params& P, int i, unsigned vb, unsigned sB) {
  return load_state_b(mkbuf(P.cur + (size_t)13 * P.Bp * i, 13 * sB), vb, sB);
}
template <int M, bool ALPHA1>
TOLG_DEV void roll_load(const Params& P, int i, int b, int q, unsigned vb, unsigned sB, RollIn<M>& R) {
  const size_t uStride = (size_t)M * P.Bp, gStride = (size_t)13 * M * P.Bp;
  __amdgpu_buffer_rsrc_t rU = mkbuf(P.cur_u + uStride * i, M * sB), rG = mkbuf(P.GK + gStride * i, 13 * M * sB);
  const int qp = (2 * q < M) ? q : M / 2 - 1;  // lanes past the last row pair re-read it (their product is unused)
  const unsigned vg = GK_VG(b, M) + GOFF(2 * qp, 0, M);
#pragma unroll
  for (int k = 0; k < 13; k++) bld2(rG, vg, GOFF(0, k, M), R.G[0][k], R.G[1][k]);
#pragma unroll
  for (int a = 0; a < M; a++) R.u[a] = bld(rU, vb, a * sB);
}

#ifdef TOLG_STAMPS
struct RStamps { unsigned long long acc[8], t; };
#define RSTAMP(k) { __builtin_amdgcn_sched_barrier(0); unsigned long long t_ = __builtin_amdgcn_s_memtime(); ST.acc[k] += t_ - ST.t; ST.t = t_; __builtin_amdgcn_sched_barrier(0); }
#define RST_PARAM , RStamps& ST
#define RST_ARG , ST
#else
#define RSTAMP(k)
#define RST_PARAM
#define RST_ARG
#endif
// STORE: the writer lane stores u^_i and x^_{i+1} to the candidate arrays; otherwise the caller takes them
// (un_out and the return value) -- the fused rollout hands them to its linearisation wavefronts through LDS.
// load_in(R): requests this step's gains and controls (from HBM, or from the LDS input ring of the fused kernel).
// what the line-search evaluation and the expected-cost-change kernels need from inside a step
template <int M>
struct RollProbe { double e[12], du[M]; State Fn; };
// what a caller of the merit search's step (alpha < 1) may already hold of the NOMINAL trajectory: the defect d_i (have_d) and
// x_{i+1} (have_x) -- by value, so that they stay in registers
struct RollPre { bool have_d = false, have_x = false; double d[12]; State Sx; };
template <int M, bool LINEAR, bool ALPHA1, int PK, bool STORE, class CT, class LoadFn>
TOLG_DEV State roll_step(const Params& P, const CT& C, const DynK& DK, int i, int b, int q, bool writer, unsigned vb,
                         unsigned sB, double alpha, const State& So, const State& Sn, double (&un_out)[M],
                         LoadFn load_in, RollProbe<M>* probe RST_PARAM, const RollPre& pre = RollPre()) {
  const size_t stStride = (size_t)13 * P.Bp, recStride = (size_t)P.recF * P.Bp, uStride = (size_t)M * P.Bp;
  RSTAMP(0)
  RollIn<M> R;
  load_in(R);  // in flight while Log runs
  __builtin_amdgcn_sched_barrier(0);
  RSTAMP(1)
  // state deviation [Log(q^-1 q_new); xi_new - xi]   (traopt_controller.py:2680-2687)
  // one gate for the two series evaluations of the step (Log here, Exp in the pose half of the dynamics)
  V3 ew, ev;
  const Pose Dx = se3_compose(se3_inverse(So.X), Sn.X);
  const V3 wdt = DK.dt * Sn.w;
  const double yl = quat_vec2(Dx.q), th2e = dot(wdt, wdt);
  const SeriesGate sg = series_gate(log_small(yl) && exp_small(th2e), log_dom(yl) && exp_dom(th2e));
  se3_log_fast(Dx, ew, ev, sg);
  double e[12] = {ew.x, ew.y, ew.z, ev.x, ev.y, ev.z, Sn.w.x - So.w.x, Sn.w.y - So.w.y, Sn.w.z - So.w.z,
                  Sn.v.x - So.v.x, Sn.v.y - So.v.y, Sn.v.z - So.v.z};
  // du = alpha k + K dx: this lane's two rows, then quad broadcast (identical bits in all four lanes)
  // the pose half of f(x^, u^) does not depend on u^: it runs here, ahead of the gain product, and gives
  // the gain loads issued at the top of the step another ~700 cycles to land
  State Fn;
  if (!LINEAR && DK.diag) Fn.X = dyn_pose_k(DK, Sn, sg);
  RSTAMP(2)
  double mine[2];
#pragma unroll
  for (int sidx = 0; sidx < 2; sidx++) {
    double sacc = alpha * R.G[sidx][12];
#pragma unroll
    for (int k = 0; k < 12; k++) sacc += R.G[sidx][k] * e[k];
    mine[sidx] = sacc;
  }
  double un[M], du[M];
  du[0] = quad_bcast<0>(mine[0]); du[1] = quad_bcast<0>(mine[1]);
  du[2] = quad_bcast<1>(mine[0]); du[3] = quad_bcast<1>(mine[1]);
  if constexpr (M == 6) { du[4] = quad_bcast<2>(mine[0]); du[5] = quad_bcast<2>(mine[1]); }
#pragma unroll
  for (int a = 0; a < M; a++) un[a] = R.u[a] + du[a];
  if (probe) {
#pragma unroll
    for (int a = 0; a < 12; a++) probe->e[a] = e[a];
#pragma unroll
    for (int a = 0; a < M; a++) probe->du[a] = du[a];
  }
  RSTAMP(3)
  State Nx;
  if constexpr (!LINEAR) {
    if (DK.diag) dyn_twist_k<M, CT, PK>(DK, C, Sn, un, Fn);
    else Fn = dyn_f<M, CT, PK>(C, Sn, un);
    if (probe) probe->Fn = Fn;
    if constexpr (ALPHA1) {
      // alpha = 1 (and single shooting): the factors of :2713-2716 are the identity and zero up to rounding
      // (see the note at the record layout), the step is x^_{i+1} = f(x^_i, u^_i)
      RSTAMP(4)
      Nx = Fn;
    } else {
      // line-search step (traopt_controller.py:2713-2716): x^_{i+1} = x_{i+1} Exp(alpha d_q) f_q(x_i,u_i)^-1 f_q(x^_i,u^_i),
      // xi^_{i+1} = xi_{i+1} - f_xi(x_i,u_i) + alpha d_xi + f_xi(x^_i,u^_i).  The reference evaluates the nominal dynamics again
      // for the two factors; with the stored defect d = [Log(x_{i+1}^-1 f_q); f_xi - xi_{i+1}] they are functions of the nominal
      // trajectory alone: f_q = x_{i+1} Exp(d_q), so x_{i+1} Exp(alpha d_q) f_q^-1 = x_{i+1} Exp((alpha - 1) d_q) x_{i+1}^-1
      // (exponentials of one twist commute) and xi_{i+1} - f_xi + alpha d_xi = (alpha - 1) d_xi -- equal to the reference's
      // factors up to the rounding of Exp(Log(X)) = X, which is the noise its own factors carry (note at the record layout).
      // Round 4: no second dynamics evaluation and no closed-form Exp on the chain, the inputs requested at the top of the
      // step: the 4-alpha stage of the merit search 0.78 -> see DESIGN section 5.
      // (requesting d and x_{i+1} at the top of the step, with the gains, was measured: 25 more values live across the Log,
      // 416 -> 447 registers, the stage 0.24 -> 0.31 ms; they are requested here)
      // SxPre / dPre: the caller already holds x_{i+1} (its next nominal state, fetched a knot ahead) and the defect of this
      // knot (fetched with it): nothing is requested on the chain then
      __amdgpu_buffer_rsrc_t rR = mkbuf(P.REC + recStride * i, (unsigned)P.recF * sB);
      double dms[12];
#pragma unroll
      for (int a = 0; a < 12; a++) dms[a] = pre.have_d ? pre.d[a] : bld(rR, REC_VR(b), FOFF(REC_D + a));
      const State Sx = pre.have_x ? pre.Sx : load_state_b(mkbuf(P.cur + stStride * (i + 1), 13 * sB), vb, sB);
      const double am1 = alpha - 1.0;
      const Pose Mx = se3_compose(se3_compose(Sx.X, se3_exp_fast(am1 * v3(dms[0], dms[1], dms[2]), am1 * v3(dms[3], dms[4], dms[5]))),
                                  se3_inverse(Sx.X));
      const V3 cw = am1 * v3(dms[6], dms[7], dms[8]);
      const V3 cv = am1 * v3(dms[9], dms[10], dms[11]);
      RSTAMP(4)
      Nx.X = se3_project(se3_compose(Mx, Fn.X));
      Nx.w = cw + Fn.w;
      Nx.v = cv + Fn.v;
    }
  } else {
    __amdgpu_buffer_rsrc_t rR = mkbuf(P.REC + recStride * i, (unsigned)P.recF * sB);
    double lin[12], d[12];
    fx_apply<M>(P, *P.c, i, b, e, du, lin);
#pragma unroll
    for (int a = 0; a < 12; a++) d[a] = alpha * bld(rR, REC_VR(b), FOFF(REC_D + a));
    State Sx = load_state_b(mkbuf(P.cur + stStride * (i + 1), 13 * sB), vb, sB);
    Pose D = se3_exp(v3(lin[0] + d[0], lin[1] + d[1], lin[2] + d[2]), v3(lin[3] + d[3], lin[4] + d[4], lin[5] + d[5]));
    Nx.X = se3_project(se3_compose(Sx.X, D));
    Nx.w = Sx.w + v3(lin[6] + d[6], lin[7] + d[7], lin[8] + d[8]);
    Nx.v = Sx.v + v3(lin[9] + d[9], lin[10] + d[10], lin[11] + d[11]);
  }
  RSTAMP(5)
  if constexpr (STORE) {
    if (writer) {
      __amdgpu_buffer_rsrc_t rCU = mkbuf(P.cand_u + uStride * i, M * sB);
#pragma unroll
      for (int a = 0; a < M; a++) bst(rCU, vb, a * sB, un[a]);
      store_state_b(mkbuf(P.cand + stStride * (i + 1), 13 * sB), vb, sB, Nx);
    }
  }
#pragma unroll
  for (int a = 0; a < M; a++) un_out[a] = un[a];
  RSTAMP(6)
  return Nx;
}

// The fused rollout's step (accept-always, alpha = 1): the twist half of roll_step.  The pose half of the dynamics,
// X_{i+1} = project(X_i Exp(xi_i dt)), depends on the state alone; a second wavefront runs that chain one step ahead
// (k_rollout_lin) and get_pose() hands over X_i.  Sn holds xi_i on entry (its pose is filled in here) and xi_{i+1}
// on return.  Same expressions in the same order as roll_step.
template <int M, class CT, class LoadFn, class PoseFn>
TOLG_DEV void roll_step_twist(const CT& C, const DynK& DK, const State& So, State& Sn, double (&un_out)[M],
                              LoadFn load_in, PoseFn get_pose RST_PARAM) {
  RSTAMP(0)
  RollIn<M> R;
  load_in(R);  // in flight while Log runs
  __builtin_amdgcn_sched_barrier(0);
  RSTAMP(1)
  Sn.X = get_pose();
  RSTAMP(5)
  // state deviation [Log(q^-1 q_new); xi_new - xi]   (traopt_controller.py:2680-2687)
  V3 ew, ev;
  const Pose Dx = se3_compose(se3_inverse(So.X), Sn.X);
  const double yl = quat_vec2(Dx.q);
  se3_log_fast(Dx, ew, ev, series_gate(log_small(yl), log_dom(yl)));
  const double e[12] = {ew.x, ew.y, ew.z, ev.x, ev.y, ev.z, Sn.w.x - So.w.x, Sn.w.y - So.w.y, Sn.w.z - So.w.z,
                        Sn.v.x - So.v.x, Sn.v.y - So.v.y, Sn.v.z - So.v.z};
  RSTAMP(2)
  double mine[2];
#pragma unroll
  for (int sidx = 0; sidx < 2; sidx++) {
    double sacc = R.G[sidx][12];
#pragma unroll
    for (int k = 0; k < 12; k++) sacc += R.G[sidx][k] * e[k];
    mine[sidx] = sacc;
  }
  double un[M], du[M];
  du[0] = quad_bcast<0>(mine[0]); du[1] = quad_bcast<0>(mine[1]);
  du[2] = quad_bcast<1>(mine[0]); du[3] = quad_bcast<1>(mine[1]);
  if constexpr (M == 6) { du[4] = quad_bcast<2>(mine[0]); du[5] = quad_bcast<2>(mine[1]); }
#pragma unroll
  for (int a = 0; a < M; a++) un[a] = R.u[a] + du[a];
  RSTAMP(3)
  State Fn;
  if (DK.diag) dyn_twist_k<M, CT, 0>(DK, C, Sn, un, Fn);
  else Fn = dyn_f<M, CT, 0>(C, Sn, un);
  RSTAMP(4)
  Sn.w = Fn.w;
  Sn.v = Fn.v;
#pragma unroll
  for (int a = 0; a < M; a++) un_out[a] = un[a];
}

template <int M, bool LINEAR, bool ALPHA1, int PK = 0>
__global__ __launch_bounds__(64) void k_rollout(Params P, double alpha, int i0, int i1) {
  // knots [i0, i1): a rollout can be issued in segments so that the re-linearisation of finished
  // knots (K1, on a second stream) overlaps the remaining sequential sweep
  typedef typename std::conditional<LINEAR, Consts, DConsts>::type CT;
  const CT& C = *(const CT*)P.c;
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  int b = t >> 2;
  const int q = t & 3;
  // quads past the batch replay the last trajectory (DPP needs whole quads alive) and store nothing
  const bool live = b < P.Bp;
  if (!live) b = P.Bp - 1;
  if (!P.active[b]) return;  // quad-uniform
  if (LINEAR && P.affine && !P.ec_redo[b]) return;  // (quad-uniform) its linear rollout comes from the affine recursion
  const bool writer = live && q == 0;  // one lane of a live quad stores
  const unsigned sB = (unsigned)P.Bp * 8u, vb = (unsigned)b * 8u;
  State Sn;  // new trajectory, knot i
  if (i0 == 0) {
    Sn = load_state_b(mkbuf(P.cur, 13 * sB), vb, sB);
    if (writer) store_state_b(mkbuf(P.cand, 13 * sB), vb, sB, Sn);
  } else {
    Sn = load_state_b(mkbuf(P.cand + (size_t)13 * P.Bp * i0, 13 * sB), vb, sB);
  }
  const DynK DK = dynk_load(*P.c);  // generic pointer: see the note at DConsts
#ifdef TOLG_STAMPS
  RStamps ST;
  for (int k = 0; k < 8; k++) ST.acc[k] = 0;
  ST.t = __builtin_amdgcn_s_memtime();
#endif
  State Sa = roll_load_state(P, i0, vb, sB), Sb = Sa;
  double un_[M];
  for (int i = i0; i < i1; i += 2) {
    if (i + 1 < i1) Sb = roll_load_state(P, i + 1, vb, sB);
    __builtin_amdgcn_sched_barrier(0);
    Sn = roll_step<M, LINEAR, ALPHA1, PK, true>(P, C, DK, i, b, q, writer, vb, sB, alpha, Sa, Sn, un_,
                                                [&](RollIn<M>& R) { roll_load<M, ALPHA1>(P, i, b, q, vb, sB, R); }, nullptr RST_ARG);
    if (i + 1 >= i1) break;
    if (i + 2 < i1) Sa = roll_load_state(P, i + 2, vb, sB);
    __builtin_amdgcn_sched_barrier(0);
    Sn = roll_step<M, LINEAR, ALPHA1, PK, true>(P, C, DK, i + 1, b, q, writer, vb, sB, alpha, Sb, Sn, un_,
                                                [&](RollIn<M>& R) { roll_load<M, ALPHA1>(P, i + 1, b, q, vb, sB, R); }, nullptr RST_ARG);
  }
#ifdef TOLG_STAMPS
  if (blockIdx.x == 5 && threadIdx.x == 0 && P.alpha_hist) { for (int k = 0; k < 8; k++) P.alpha_hist[(size_t)80 * P.max_iter + k] = (double)ST.acc[k]; }
#endif
}

// ------------------------------------------------------------------------------------------------
// K3 + K1 fused: the accept-always MS iteration (line_search=False, rollout='nonlinear': the setting of
// every benchmark_*.py) rolls out and re-linearises in ONE launch.  A 256-thread workgroup owns 16
// trajectories and runs one wavefront on each SIMD of its CU:
//   wave 0  the sequential rollout (roll_step above, four lanes per trajectory).  A dependent chain that
//           leaves three SIMDs of the CU idle -- and, here, issues no vector-memory instruction at all;
//   wave 1  the loader: streams the gains, controls and nominal states of the knots ahead of wave 0 from
//           HBM into an LDS input ring with LDS-DMA (global_load_lds_dwordx4: no registers, 1 KB per
//           instruction), up to RL_DEPTH knots ahead.  Wave 0 reads them with ds_read: its chain no longer sees HBM
//           latency, nor the address-path queueing behind the helpers' record stores (with wave 0 loading
//           from HBM itself the fused launch took 0.43 ms, the rollout alone 0.28: profiles/r02_*);
//   waves 2, 3  linearisation helpers: the linearisation of knot i needs only (x^_i, u^_i, x^_{i+1}), so
//           it can run one step behind the rollout.  Wave 0 publishes every new state / control into an LDS
//           state ring instead of storing it to HBM; a helper pass takes 64 lanes = 4 consecutive knots x 16
//           trajectories through lin_knot (helper h owns every second group of four knots), writes the knot
//           records and commits the new trajectory to P.cur / P.cur_u in place.
// In-place is safe: knot i of P.cur / P.cur_u is overwritten only after rollout step i has completed, and
// the loader fetched the old value before that step could start; nothing else reads it during the launch.
// No candidate trajectory is written or re-read, and the separate K1 launch leaves the critical path (one
// helper pass, ~7 us, remains as a tail).
//
// Synchronisation is LDS only (one workgroup = one CU), without s_barrier: LDS operations of a wave execute
// in order, so "data, then counter" needs no fence on the producer side; LDS-DMA data is published only
// after the loader's own counted s_waitcnt vmcnt has retired it.  sync[0] = rollout steps completed
// (states 0..sync[0], controls 0..sync[0]-1 are in the state ring); sync[1] = knots whose inputs are in the
// input ring; sync[2+h] = groups helper h has finished.  Wait-for graph: loader and helpers wait for wave 0;
// wave 0 waits for the loader (which then depends only on steps wave 0 has already completed) and for a
// helper only to reuse a state-ring slot (same argument): no cycle.  Every poll loop is bounded: a stuck
// counter ends the kernel with TOLG_ST_INTERNAL instead of hanging the GPU.
// ------------------------------------------------------------------------------------------------
enum { RL_RING = 24, RL_PAIRS = 10, RL_DEPTH = 5, RL_NH = 2, RL_POLLS = 1 << 21 };
// the counters are accessed through an LDS-address-space pointer: a volatile access through a generic pointer
// becomes a flat load / store followed by s_waitcnt vmcnt(0), which would drain the poller's own memory queue
typedef volatile __attribute__((address_space(3))) int* rl_sync_t;
TOLG_DEV bool rl_wait_ge(rl_sync_t p, int v) {
  for (int n = 0; n < RL_POLLS; n++) {
    if (*p >= v) return true;
    __builtin_amdgcn_s_sleep(4);
  }
  return false;
}
// state ring slot: [pair][trajectory][2] doubles: the pose in pairs 0..3 (quaternion, translation, padding), the
// twist in pairs 4..6, the controls in 7..9.  Pose and twist of a knot come from different wavefronts.
TOLG_DEV void rl_put_pose(double* slot, int tt, const Pose& X) {
  f64x2* p = reinterpret_cast<f64x2*>(slot) + tt;
  p[0 * 16] = f64x2{X.q.x, X.q.y}; p[1 * 16] = f64x2{X.q.z, X.q.w}; p[2 * 16] = f64x2{X.t.x, X.t.y};
  p[3 * 16] = f64x2{X.t.z, 0.0};
}
TOLG_DEV void rl_put_twist(double* slot, int tt, V3 w, V3 v) {
  f64x2* p = reinterpret_cast<f64x2*>(slot) + tt;
  p[4 * 16] = f64x2{w.x, w.y}; p[5 * 16] = f64x2{w.z, v.x}; p[6 * 16] = f64x2{v.y, v.z};
}
TOLG_DEV Pose rl_get_pose(const double* slot, int tt) {
  const f64x2* p = reinterpret_cast<const f64x2*>(slot) + tt;
  const f64x2 a = p[0 * 16], b = p[1 * 16], c = p[2 * 16], d = p[3 * 16];
  Pose X;
  X.q.x = a.x; X.q.y = a.y; X.q.z = b.x; X.q.w = b.y;
  X.t = v3(c.x, c.y, d.x);
  return X;
}
TOLG_DEV void rl_get_twist(const double* slot, int tt, V3& w, V3& v) {
  const f64x2* p = reinterpret_cast<const f64x2*>(slot) + tt;
  const f64x2 e = p[4 * 16], f = p[5 * 16], g = p[6 * 16];
  w = v3(e.x, e.y, f.x);
  v = v3(f.y, g.x, g.y);
}
TOLG_DEV void rl_put_state(double* slot, int tt, const State& S) {
  rl_put_pose(slot, tt, S.X);
  rl_put_twist(slot, tt, S.w, S.v);
}
TOLG_DEV State rl_get_state(const double* slot, int tt) {
  State S;
  S.X = rl_get_pose(slot, tt);
  rl_get_twist(slot, tt, S.w, S.v);
  return S;
}
// input ring slot (bytes): the gains of the workgroup's four 4-trajectory groups exactly as they lie in GK
// (contiguous), then 13 rows of 16 nominal-state values, then M rows of 16 controls (rows of P.cur / P.cur_u)
template <int M>
struct RlIn {
  enum { GSZ = 13 * (M / 2) * 64, GAINS = 4 * GSZ, STATE = 13 * 128, CTRL = M * 128, SLOT = GAINS + STATE + CTRL,
         NDMA = (GAINS + 1023) / 1024 + 2 + 1 };  // LDS-DMA instructions per knot
};
#define TOLG_DEF_DMA(SUF, POL)                                                                                                   \
  TOLG_DEV void rl_dma16##SUF(const void* gsrc, unsigned lds_dst) {                                                              \
    unsigned keep;                                                                                                               \
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off" POL "\n\ts_mov_b32 m0, %0"    \
                 : "=&s"(keep) : "v"(gsrc), "s"(lds_dst) : "memory");                                                            \
  }                                                                                                                              \
  TOLG_DEV void rl_dma16x4##SUF(const void* sbase, unsigned voff, unsigned lds_dst) {                                            \
    unsigned keep;                                                                                                               \
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 2\n\t"                                                          \
                 "global_load_lds_dwordx4 %1, %2" POL "\n\tglobal_load_lds_dwordx4 %1, %2 offset:1024" POL "\n\t"                 \
                 "global_load_lds_dwordx4 %1, %2 offset:2048" POL "\n\tglobal_load_lds_dwordx4 %1, %2 offset:3072" POL "\n\t"     \
                 "s_mov_b32 m0, %0"                                                                                              \
                 : "=&s"(keep) : "v"(voff), "s"(sbase), "s"(lds_dst) : "memory");                                                \
  }                                                                                                                              \
  TOLG_DEV void rl_dma16x3##SUF(const void* sbase, unsigned voff, unsigned lds_dst) {                                            \
    unsigned keep;                                                                                                               \
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 2\n\t"                                                          \
                 "global_load_lds_dwordx4 %1, %2" POL "\n\tglobal_load_lds_dwordx4 %1, %2 offset:1024" POL "\n\t"                 \
                 "global_load_lds_dwordx4 %1, %2 offset:2048" POL "\n\t"                                                         \
                 "s_mov_b32 m0, %0"                                                                                              \
                 : "=&s"(keep) : "v"(voff), "s"(sbase), "s"(lds_dst) : "memory");                                                \
  }
// rl_dma16: one 16-byte-per-lane LDS-DMA, lane l copies 16 bytes from gsrc (its own address) to LDS byte address lds_dst + 16 l.
// In asm, so that the compiler keeps no vmcnt bookkeeping for it (it would drain the DMA queue before every LDS poll);
// M0, the DMA's LDS base, is compiler-reserved: saved and restored inside the statement (cdna_hip_programming.md §5).
// rl_dma16x4 / x3: four / three consecutive KB with one M0 setting: the instruction offset moves both the global and the LDS
// address.  base is wave-uniform (SGPR pair), voff the lane's byte offset.  The s_nop 2: base may have been written by a
// v_readfirstlane directly in front of the statement (uniform_ptr), and a VALU write of an SGPR must be five wait states ahead
// of the vector-memory instruction that reads it -- the compiler pads its own code, not inline asm (tools/dpp_hazard_lint.py,
// second rule, found k_rollout_lin's bursts at 3-4).  (M0 handling is a quarter of an LDS-DMA's issue time:
// tools/lds_dma_microbench.hip)
TOLG_DEF_DMA(, TOLG_POL(TOLG_NT_RING))       // the fused launch's input ring
TOLG_DEF_DMA(_rec, TOLG_POL(TOLG_NT_REC))    // the backward sweep's record ring
TOLG_DEV const void* uniform_ptr(const void* p) {  // a wave-uniform address, in a form the "s" constraint accepts
  const unsigned long long a = reinterpret_cast<unsigned long long>(p);
  return reinterpret_cast<const void*>(((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)(a >> 32)) << 32) |
                                       (unsigned)__builtin_amdgcn_readfirstlane((int)a));
}
#include "tolg_backward3.h"

template <int M>
TOLG_DEV void rl_in_load(const char* slot, int tt, int q, RollIn<M>& R) {
  const int qp = (2 * q < M) ? q : M / 2 - 1;  // lanes past the last row pair re-read it (their product is unused)
  const char* g = slot + (tt >> 2) * RlIn<M>::GSZ + qp * 64 + (tt & 3) * 16;
#pragma unroll
  for (int k = 0; k < 13; k++) {
    const f64x2 w = *reinterpret_cast<const f64x2*>(g + k * (M / 2) * 64);
    R.G[0][k] = w.x; R.G[1][k] = w.y;
  }
  const double* u = reinterpret_cast<const double*>(slot + RlIn<M>::GAINS + RlIn<M>::STATE) + tt;
#pragma unroll
  for (int a = 0; a < M; a++) R.u[a] = u[a * 16];
}
template <int M>
TOLG_DEV State rl_in_state(const char* slot, int tt) {
  const double* x = reinterpret_cast<const double*>(slot + RlIn<M>::GAINS) + tt;
  State S;
  S.X.q.x = x[0]; S.X.q.y = x[16]; S.X.q.z = x[32]; S.X.q.w = x[48];
  S.X.t = v3(x[64], x[80], x[96]);
  S.w = v3(x[112], x[128], x[144]);
  S.v = v3(x[160], x[176], x[192]);
  return S;
}
template <int M>
constexpr size_t rl_static_lds() { return (size_t)RL_DEPTH * RlIn<M>::SLOT + (size_t)RL_RING * RL_PAIRS * 256 + 32; }
template <int M>
__global__ __launch_bounds__(256) void k_rollout_lin(Params P, int it) {
  typedef RlIn<M> IN;
  const DConsts& C = *(const DConsts*)P.c;
  // One LDS object, carved by hand: the input ring first -- the LDS-DMA base register M0 is used with its
  // classic 16-bit range, so every DMA destination stays below 64 KB -- then the state ring, then the counters.
  static_assert(RL_DEPTH * IN::SLOT <= 65536 && IN::SLOT % 16 == 0, "LDS-DMA destinations must stay below 64 KB");
  // ... and the partial cost sums [helper][knot of a pass][trajectory]: every helper lane adds up the stage costs of
  // the knots it linearises (one knot of every second pass: a fixed order), the helper that finishes last adds the
  // eight partial sums of each trajectory in a fixed order and does the per-iteration bookkeeping (k_reduce's job in
  // the split schedule).  No table over the horizon: the launch has no horizon limit (round 2 kept [N + 1][16] stage
  // costs in dynamic LDS and fell back to the split schedule beyond N = 313 for m = 6).
  __shared__ __attribute__((aligned(16))) char lds[rl_static_lds<M>()];
  __shared__ double lpart[RL_NH][4][16];
  char (*inring)[IN::SLOT] = reinterpret_cast<char (*)[IN::SLOT]>(lds);
  double (*ring)[RL_PAIRS * 32] = reinterpret_cast<double (*)[RL_PAIRS * 32]>(lds + RL_DEPTH * IN::SLOT);
  int* sync = reinterpret_cast<int*>(lds + RL_DEPTH * IN::SLOT + RL_RING * RL_PAIRS * 256);
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int b0 = blockIdx.x * 16, N = P.N;
  {
    // is any of the workgroup's trajectories still being solved?  (same answer in every wave: all leave or none)
    int bb = b0 + (lane & 15);
    if (bb >= P.Bp) bb = P.Bp - 1;
    if (!__any(P.active[bb] != 0)) return;
  }
  if (threadIdx.x < 8) sync[threadIdx.x] = (threadIdx.x == 0) ? -1 : 0;
  __syncthreads();
  const rl_sync_t vs = (rl_sync_t)sync;
  const unsigned sB = (unsigned)P.Bp * 8u;
  if (wave == 0) {
    // ---------------- the rollout (k_rollout<M, false, true, 0> with LDS in place of HBM on both sides)
    int b = b0 + (lane >> 2);
    const int q = lane & 3, tt = lane >> 2;
    if (b >= P.Bp) b = P.Bp - 1;  // quads past the batch compute on whatever the loader fetched; the helpers ignore their slots
    const bool writer = q == 0;
    const unsigned vb = (unsigned)b * 8u;
    bool ok = true;
    int loaded = 0;  // last value seen of sync[1]
    auto need_inputs = [&](int knots) -> bool {  // wait until the inputs of knots 0 .. knots-1 are in the input ring
      if (loaded >= knots) return true;
      if (!rl_wait_ge(vs + 1, knots)) return false;
      loaded = vs[1];
      asm volatile("" ::: "memory");
      return true;
    };
    if (!need_inputs(N < 2 ? N : 2)) { if (writer && b0 + tt < P.Bp) P.status[b0 + tt] = TOLG_ST_INTERNAL; return; }
    State Sn = rl_in_state<M>(inring[0], tt);  // x^_0 = x_0
    if (writer) rl_put_state(ring[0], tt, Sn);
    asm volatile("" ::: "memory");
    __builtin_amdgcn_wave_barrier();
    if (lane == 0) vs[0] = 0;
    const DynK DK = dynk_load(*P.c);  // generic pointer: see the note at DConsts
    State Sa = Sn, Sb = Sn;
    double un[M];
#ifdef TOLG_STAMPS
    RStamps ST;
    for (int k = 0; k < 8; k++) ST.acc[k] = 0;
    ST.t = __builtin_amdgcn_s_memtime();
#endif
    // The pose of x^_i comes from the pose wavefront (sync[7] = poses published).  It is read speculatively
    // right after this wave's own publish of step i - 1 -- counter first, then the data: LDS operations execute in
    // order, so a counter value >= i vouches for the data read behind it -- and only re-read after a poll if the
    // pose wave was late (it has a whole step of slack).
    int pose_seen = 0;
    Pose Xspec = Sn.X;
    auto finish_step = [&](int i) {  // publish u^_i and the twist of x^_{i+1}
      if (writer) {
        f64x2* pu = reinterpret_cast<f64x2*>(ring[i % RL_RING]) + 7 * 16 + tt;
#pragma unroll
        for (int a = 0; a < M; a += 2) pu[(a / 2) * 16] = f64x2{un[a], un[a + 1]};
        rl_put_twist(ring[(i + 1) % RL_RING], tt, Sn.w, Sn.v);
      }
      asm volatile("" ::: "memory");
      __builtin_amdgcn_wave_barrier();
      if (lane == 0) vs[0] = i + 1;
      pose_seen = vs[7];
      asm volatile("" ::: "memory");
      Xspec = rl_get_pose(ring[(i + 1) % RL_RING], tt);
    };
    // State-ring back-pressure, checked once per eight steps for the eight slots ahead (a poll is an LDS round
    // trip on the critical chain): steps i .. i+7 overwrite the slots of knots up to `old` = i + 8 - RL_RING, which
    // is safe once every group of four knots up to the one holding `old` has been linearised (a group also
    // reads the first state of the next group).  Helper h owns groups h, h + RL_NH, ...: it must have finished
    // (old/4 - h) / RL_NH + 1 of them.  (The pose wave writes slot i + 1 only after it has seen step i - 1
    // published, which this check precedes.)
    auto slot_free = [&](int i) -> bool {
      if ((i & 7) != 0 || i + 8 < RL_RING) return true;
      const int go = (i + 8 - RL_RING) / 4;
#pragma unroll
      for (int h = 0; h < RL_NH; h++)
        if (go >= h && !rl_wait_ge(vs + 2 + h, (go - h) / RL_NH + 1)) return false;
      return true;
    };
    // what step i needs before it starts: ring slots to publish into, its inputs, and the nominal state of knot
    // i + 1 (read one knot ahead, as the HBM version prefetched it)
    auto ahead = [&](int i, State& Snext) -> bool {
      if (!slot_free(i)) return false;
      if (i + 1 < N) {
        if (!need_inputs(i + 2)) return false;
        Snext = rl_in_state<M>(inring[(i + 1) % RL_DEPTH], tt);
      }
      return true;
    };
    // one step.  So holds the nominal state of knot i on entry and of knot i + 2 on return: the look-ahead for
    // step i + 1 runs BEFORE this step's publish, so that its counter reads do not wait behind the LDS writes of
    // the publish (LDS operations of a wave return in order).
    auto step = [&](int i, State& So) -> bool {
      __builtin_amdgcn_sched_barrier(0);
      const char* slot = inring[i % RL_DEPTH];
      bool got = true;
      roll_step_twist<M>(C, DK, So, Sn, un, [&](RollIn<M>& R) { rl_in_load<M>(slot, tt, q, R); },
                         [&]() -> Pose {
                           if (pose_seen < i) {
                             got = rl_wait_ge(vs + 7, i);
                             asm volatile("" ::: "memory");
                             Xspec = rl_get_pose(ring[i % RL_RING], tt);
                           }
                           return Xspec;
                         } RST_ARG);
      if (!got) return false;
      if (i + 1 < N && !ahead(i + 1, So)) return false;
      finish_step(i);
      RSTAMP(7)
      return true;
    };
    ok = ahead(0, Sb);
    for (int i = 0; ok && i < N; i += 2) {
      if (!(ok = step(i, Sa))) break;
      if (i + 1 >= N) break;
      if (!(ok = step(i + 1, Sb))) break;
    }
#ifdef TOLG_STAMPS
    if (blockIdx.x == 5 && threadIdx.x == 0 && P.alpha_hist) { for (int k = 0; k < 8; k++) P.alpha_hist[(size_t)80 * P.max_iter + k] = (double)ST.acc[k]; }
#endif
    if (!ok && writer && b0 + tt < P.Bp) P.status[b0 + tt] = TOLG_ST_INTERNAL;
    return;
  }
  if (wave == 1) {
    // ---------------- the loader: inputs of knot k go to slot k % RL_DEPTH once step k - RL_DEPTH has completed
    const size_t stStride = (size_t)13 * P.Bp, uStride = (size_t)M * P.Bp, gStride = (size_t)13 * M * P.Bp;
    const unsigned lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)lds;
    const int ntraj = (P.Bp - b0 < 16) ? P.Bp - b0 : 16;              // Bp is a multiple of 4
    const int gchunks = (ntraj / 4) * (IN::GSZ / 16);                 // 16-byte pieces of gains this workgroup owns
    auto issue = [&](int k) {
      const unsigned dst = (unsigned)__builtin_amdgcn_readfirstlane((int)(lds0 + (unsigned)(k % RL_DEPTH) * IN::SLOT));
      const char* gk = reinterpret_cast<const char*>(P.GK + gStride * k + (size_t)(b0 >> 2) * 13 * M * 4);
      // every instruction is issued whatever the workgroup's share (the counted vmcnt below relies on NDMA per knot):
      // pieces of groups past the batch re-read piece 0 into their (unused) place.  A full workgroup moves its run
      // four KB per M0 setting.
      constexpr int QUADS = (IN::GAINS / 1024) / 4;
      if (ntraj == 16) {
#pragma unroll
        for (int g4 = 0; g4 < QUADS; g4++) rl_dma16x4(uniform_ptr(gk + g4 * 4096), (unsigned)lane * 16u, dst + g4 * 4096);
      }
#pragma unroll
      for (int c = 0; c < (IN::GAINS + 1023) / 1024; c++) {
        if (c < 4 * QUADS && ntraj == 16) continue;
        const int ch = c * 64 + lane;
        if (ch < IN::GAINS / 16) rl_dma16(gk + (size_t)(ch < gchunks ? ch : 0) * 16, dst + c * 1024);
      }
      // states: 13 rows (fields) x 128 bytes; controls: M rows x 128 bytes; piece = 16 bytes, 8 per row
      const char* xs = reinterpret_cast<const char*>(P.cur + stStride * k + b0);
#pragma unroll
      for (int c = 0; c < 2; c++) {
        const int ch = c * 64 + lane;
        // (pieces of trajectories past the batch re-read piece 0 of the row: a short last workgroup must not read
        // beyond the end of the arrays)
        if (ch < 13 * 8) rl_dma16(xs + (size_t)(ch >> 3) * sB + (((ch & 7) * 2 < ntraj) ? (ch & 7) : 0) * 16, dst + IN::GAINS + c * 1024);
      }
      const char* us = reinterpret_cast<const char*>(P.cur_u + uStride * k + b0);
      if (lane < M * 8) rl_dma16(us + (size_t)(lane >> 3) * sB + (((lane & 7) * 2 < ntraj) ? (lane & 7) : 0) * 16, dst + IN::GAINS + IN::STATE);
    };
    // Inputs: knot k may be issued once step k - RL_DEPTH has completed; a knot is published once its DMAs have
    // retired -- the memory queue retires in order, so "at most n knots still outstanding" is a counted s_waitcnt
    // with an immediate.  No polling on the memory side: iteration i of the pose loop below (which starts when step
    // i - 1 has completed) issues knot i + 4 into the slot step i - 1 has finished with and publishes knot i + 2, issued
    // two iterations (~3 us) earlier; wave 0 asks for it at the end of its step i.
    static_assert(2 * IN::NDMA <= 63 && RL_DEPTH >= 5, "vmcnt is a 6-bit counter; the look-ahead below spans 5 slots");
    int issued = 0;
    auto publish_loaded = [&](int target) {  // knots < target have landed (issued - target in 0..2)
      switch (issued - target) {
        case 0: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
        case 1: asm volatile("s_waitcnt vmcnt(%0)" :: "n"(1 * IN::NDMA) : "memory"); break;
        default: asm volatile("s_waitcnt vmcnt(%0)" :: "n"(2 * IN::NDMA) : "memory"); break;
      }
      __builtin_amdgcn_wave_barrier();
      if (lane == 0) vs[1] = target;
    };
    while (issued < N && issued < RL_DEPTH) issue(issued++);
    publish_loaded(N < 3 ? N : 3);
    // ---------------- ... and the pose chain X_{i+1} = project(X_i Exp(xi_i dt)) (the pose half of f, dyn_pose_k), one
    // step ahead of wave 0, which produces xi_i at the end of its step i - 1 and needs X_{i+1} at the start of step
    // i + 1.  Four lanes per trajectory, as in wave 0 (same lane -> trajectory map, lane 0 of a quad writes).
    const int tt = lane >> 2;
    const bool writer = (lane & 3) == 0;
    const DynK DK = dynk_load(*P.c);  // generic pointer: see the note at DConsts
    State S = rl_in_state<M>(inring[0], tt);  // x^_0 = x_0
#ifdef TOLG_STAMPS
    unsigned long long pw[3] = {0, 0, 0}, pw_t = __builtin_amdgcn_s_memtime();
#define PSTAMP(k) { __builtin_amdgcn_sched_barrier(0); unsigned long long t_ = __builtin_amdgcn_s_memtime(); pw[k] += t_ - pw_t; pw_t = t_; __builtin_amdgcn_sched_barrier(0); }
#else
#define PSTAMP(k)
#endif
    for (int i = 0; i < N; i++) {
      if (i > 0) {
        bool seen = false;
        for (int n = 0; n < RL_POLLS && !(seen = __builtin_amdgcn_readfirstlane(vs[0]) >= i); n++) __builtin_amdgcn_s_sleep(1);
        if (!seen) return;  // wave 0 reports the failure (its own poll of sync[7] runs out)
        asm volatile("" ::: "memory");
        PSTAMP(0)
        rl_get_twist(ring[i % RL_RING], tt, S.w, S.v);
        if (issued < N) issue(issued++);  // knot i + 4, into the slot step i - 1 has finished with
        publish_loaded(i + 3 < N ? i + 3 : N);
        PSTAMP(1)
      }
      S.X = dyn_pose_k(DK, S);
      if (writer) rl_put_pose(ring[(i + 1) % RL_RING], tt, S.X);
      asm volatile("" ::: "memory");
      __builtin_amdgcn_wave_barrier();
      if (lane == 0) vs[7] = i + 1;
      PSTAMP(2)
    }
#ifdef TOLG_STAMPS
    if (blockIdx.x == 5 && lane == 0 && P.alpha_hist)
      for (int k = 0; k < 3; k++) P.alpha_hist[(size_t)83 * P.max_iter + k] = (double)pw[k];
#endif
    return;
  }
  // ---------------- linearisation helpers: pass g covers knots 4g .. 4g+3 (lane / 16) of the 16 trajectories (lane % 16)
  const int h = wave - 2, kk = lane >> 4, tt = lane & 15, b = b0 + tt;
  const bool mine = b < P.Bp && P.active[b < P.Bp ? b : 0] != 0;
  const int ngroups = (N + 1 + 3) / 4;
  int done = 0;
  double jpart = 0.0;  // stage costs of this lane's knots (kk, kk + 8, ... of helper 0; kk + 4, kk + 12, ... of helper 1)
#ifdef TOLG_STAMPS
  unsigned long long hs_wait = 0, hs_work = 0, hs_t = __builtin_amdgcn_s_memtime();
#endif
  for (int g = h; g < ngroups; g += RL_NH) {
    const int need = (4 * g + 4 < N) ? 4 * g + 4 : N;  // the last state this pass reads
#ifdef TOLG_STAMPS
    { unsigned long long t_ = __builtin_amdgcn_s_memtime(); hs_work += t_ - hs_t; hs_t = t_; }
#endif
    if (!rl_wait_ge(vs, need) || !rl_wait_ge(vs + 7, need)) {  // twists (wave 0) and poses (wave 1) up to `need`
      if (mine && kk == 0) P.status[b] = TOLG_ST_INTERNAL;
      return;
    }
    asm volatile("" ::: "memory");
#ifdef TOLG_STAMPS
    { unsigned long long t_ = __builtin_amdgcn_s_memtime(); hs_wait += t_ - hs_t; hs_t = t_; }
#endif
    const int i = 4 * g + kk;
    if (mine && i <= N) {
      const double* slot = ring[i % RL_RING];
      const State S = rl_get_state(slot, tt);
      double u[M];
#pragma unroll
      for (int a = 0; a < M; a++) u[a] = 0;
      if (i < N) {
        const f64x2* pu = reinterpret_cast<const f64x2*>(slot) + 7 * 16 + tt;
#pragma unroll
        for (int a = 0; a < M; a += 2) { const f64x2 w = pu[(a / 2) * 16]; u[a] = w.x; u[a + 1] = w.y; }
#pragma unroll
        for (int a = 0; a < M; a++) gst<TOLG_NT_CURST>(&P.cur_u[UIDX(a, i, b)], u[a]);
      }
      if (i > 0) store_state<TOLG_NT_CURST>(P, P.cur, i, b, S);  // the accepted candidate becomes the nominal trajectory
      // the terminal knot (in the last group only) goes separately: see lin_knot's TERM
      double lc = 0.0;
      if (i < N) lin_knot<M, true, 0>(P, C, i, b, 1, S, u, [&]() { return S; }, &lc);
      if (4 * g + 3 >= N && i == N) lin_knot<M, true, 1>(P, C, i, b, 1, S, u, [&]() { return S; }, &lc);
      jpart += lc;
    }
    done++;
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // every ring read of this pass has returned
    __builtin_amdgcn_wave_barrier();
    if (lane == 0) vs[2 + h] = done;
  }
  // ---- the helper that finishes last: trajectory costs from the eight partial sums (fixed order: deterministic; the
  // order differs from k_reduce's knot order, i.e. fused and split schedules agree to rounding, not to the bit), then
  // the on_iteration bookkeeping of traopt_controller.py:2621-2626.  LDS operations of a wave execute in order, so
  // the other helper's writes precede its arrival on the counter.
  lpart[h][kk][tt] = jpart;
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_wave_barrier();
  int last = 0;
  if (lane == 0) last = __hip_atomic_fetch_add(&sync[6], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) == RL_NH - 1;
  last = __builtin_amdgcn_readfirstlane(last);
  if (last && mine && kk == 0) {
    asm volatile("" ::: "memory");
    double J = 0;
#pragma unroll
    for (int hh = 0; hh < RL_NH; hh++)
#pragma unroll
      for (int k4 = 0; k4 < 4; k4++) J += lpart[hh][k4][tt];
    if (b < P.B && J > 10.0 * P.Jc[b]) P.k2_hint[b >> 2] = 8;  // a diverging solve: its group's next sweeps are the full kernel's (k_reduce)
    P.Jc[b] = J;
    P.dn[b] = 0.0;  // closed by construction
    if (b < P.B) {
      if (P.J_hist) P.J_hist[(size_t)b * P.max_iter + it] = J;
      if (P.defect_hist) P.defect_hist[(size_t)b * (P.max_iter + 1) + it + 1] = 0.0;
      if (P.alpha_hist) P.alpha_hist[(size_t)b * P.max_iter + it] = P.ls_alpha[b];
      P.iters[b] = it + 1;
      if (!(J == J) || isinf(J)) { P.status[b] = TOLG_ST_NONFINITE; P.active[b] = 0; }
    }
  }
#ifdef TOLG_STAMPS
  if (blockIdx.x == 5 && lane == 0 && P.alpha_hist) {
    P.alpha_hist[(size_t)(81 + h) * P.max_iter + 0] = (double)hs_wait;
    P.alpha_hist[(size_t)(81 + h) * P.max_iter + 1] = (double)hs_work;
    P.alpha_hist[(size_t)(81 + h) * P.max_iter + 2] = (double)done;
  }
#endif
}

// ------------------------------------------------------------------------------------------------
// Line search.  The reference tries alpha_k = 1.1^(-k^2) one after the other with a full rollout +
// trajectory cost each (SS: traopt_controller.py:1972-1990, accept J_new < J_opt; MS:
// :2549-2590, Armijo test on the merit J + w ||d||).  Here a stage evaluates several alphas of
// every still-undecided trajectory at once (candidates kept in slot buffers), then the first alpha
// that passes -- in the reference's order -- wins.  Stage forms: run_ls_stage.
// ------------------------------------------------------------------------------------------------
TOLG_DEV double ls_alpha_k(int k) { return pow(1.1, -(double)(k * k)); }
// A wide stage (several alphas of the trajectories still undecided) has two forms, chosen on the device from the number
// of undecided trajectories n (list `list` of the previous select): four lanes per (trajectory, alpha) over the
// compacted list + parallel evaluation (k_rollout_ls, k_ls_eval, k_ls_sum) while n * alphas <= LS_QUAD_MAX -- a few
// hundred waves, each at the speed of a lone rollout --, one thread per (trajectory, alpha) with cost and defect on the
// chain (k_rollout_eval_t) beyond: 64 waves per alpha whatever n is.  Measured at 4096 x 200, 12 alphas: quad form
// 0.45 + 0.15 ms at n ~ 600, 1.36 + 0.39 ms at n ~ 2500; thread form 1.3-1.4 ms for any n.  Every kernel of a stage is
// launched; the ones whose form is not in turn leave at once.
// (round 4, with the rollouts in two waves per sixteen quads: 20000 -> 32000 measured +1.2 % on the SS line, 64000 -- the quad
// form for every list of a 4096 batch -- the same; beyond ~36000 quads the quad form's 4.4 rounds of 0.3 ms pass the thread
// form's one chain of 1.4 ms)
enum { LS_QUAD_MAX = 32000 };
TOLG_DEV bool ls_quad_form(const Params& P, int list, int nslots) { return list < 0 || P.ls_count[list] * nslots <= LS_QUAD_MAX; }

// stage cost l(x, u, i) / terminal cost (traopt_cost.py:675-738)
// FAST: the series forms of tolg_lie.h (what the rollouts and the linearisation use) in place of the closed-form Log
template <int M, bool FAST = false>
TOLG_DEV double knot_cost(const Params& P, const Consts& C, int i, int b, const State& S, const double (&u)[M], bool term) {
  const double* r = P.ref + 13 * (size_t)i;
  Pose Xr;
  Xr.q.x = r[0]; Xr.q.y = r[1]; Xr.q.z = r[2]; Xr.q.w = r[3];
  Xr.t = v3(r[4], r[5], r[6]);
  V3 ew, ev;
  if constexpr (FAST) se3_log_fast(se3_compose(S.X, se3_inverse(Xr)), ew, ev);
  else se3_log(se3_compose(S.X, se3_inverse(Xr)), ew, ev);
  const bool so3 = so3_family(C.kind);  // the SO3 terminal cost is weighted with Q (App. C-Q3)
  const double* W1 = (term && !so3) ? C.P1 : C.W1;
  const double* W2 = (term && !so3) ? C.P2 : C.W2;
  double e[6] = {ew.x, ew.y, ew.z, ev.x, ev.y, ev.z};
  double ve[6] = {S.w.x - r[7], S.w.y - r[8], S.w.z - r[9], S.v.x - r[10], S.v.y - r[11], S.v.z - r[12]};
  double l = 0;
#pragma unroll
  for (int a = 0; a < 6; a++) {
    double s1 = 0, s2 = 0;
#pragma unroll
    for (int k = 0; k < 6; k++) { s1 += W1[6 * a + k] * e[k]; s2 += W2[6 * a + k] * ve[k]; }
    l += e[a] * s1 + ve[a] * s2;
  }
  if (!term) {
#pragma unroll
    for (int a = 0; a < M; a++)
#pragma unroll
      for (int k = 0; k < M; k++) l += u[a] * C.R[a * M + k] * u[k];
    if (P.al_lb) {
      const int bs = b < P.B ? b : P.B - 1;
      const double* lam = P.al_lambda + ((size_t)bs * P.N + i) * 2 * M;
      const double* imu = P.al_imu + ((size_t)bs * P.N + i) * 2 * M;
#pragma unroll
      for (int a = 0; a < M; a++) {
        double g1 = P.al_lb[a] - u[a], g2 = u[a] - P.al_ub[a];
        l += lam[a] * g1 + lam[M + a] * g2 + 0.5 * (g1 * imu[a] * g1 + g2 * imu[M + a] * g2);
      }
    }
  }
  return l;
}

// one thread per (trajectory, slot): rollout with alpha_{a0+slot}, its cost and (MS) defect norm on the chain -- the
// form of a wide stage with MANY undecided trajectories (64 waves per alpha keep a 12-alpha stage within one wave per
// SIMD whatever their number; ls_quad_form decides on the device, the quad form -- k_rollout_ls -- takes the short lists)
template <int M, bool MS, bool LINEAR>
__global__ __launch_bounds__(64) void k_rollout_eval_t(Params P, int a0, int nslots, int list) {
  const Consts& C = *P.c;
  // (round 4: the series forms of tolg_lie.h in this kernel and in k_ls_eval were measured -- SS 414 -> 404, merit 500 -> 482: a
  // wave of 64 unrelated (trajectory, alpha) pairs nearly always holds a lane that needs the long tier or the fallback, and then
  // runs all three; the closed forms stay)
  const int b = blockIdx.x * 64 + threadIdx.x, slot = blockIdx.y;
  if (b >= P.Bp || slot >= nslots) return;
  if (ls_quad_form(P, list, nslots)) return;  // the quad form's turn
  if (!P.active[b] || P.ls_accept[b] >= 0) return;
  if (LINEAR && P.affine && !P.ec_redo[b]) return;  // its candidates come from the affine recursion (k_ls_eval_affine)
  const int N = P.N, ai = a0 + slot;
  const double alpha = ls_alpha_k(ai);
  const unsigned sB = (unsigned)P.Bp * 8u, vb = (unsigned)b * 8u;
  const size_t stStride = (size_t)13 * P.Bp, recStride = (size_t)P.recF * P.Bp, uStride = (size_t)M * P.Bp,
               gStride = (size_t)13 * M * P.Bp;
  double* sx = P.slot_x + (size_t)slot * stStride * (N + 1);
  double* su = P.slot_u + (size_t)slot * uStride * N;
  State Sn = load_state_b(mkbuf(P.cur, 13 * sB), vb, sB);
  store_state_b(mkbuf(sx, 13 * sB), vb, sB, Sn);
  double J = 0, d2 = 0;
  for (int i = 0; i < N; i++) {
    __amdgpu_buffer_rsrc_t rR = mkbuf(P.REC + recStride * i, (unsigned)P.recF * sB), rG = mkbuf(P.GK + gStride * i, 13 * M * sB);
    __amdgpu_buffer_rsrc_t rU = mkbuf(P.cur_u + uStride * i, M * sB);
    State So = load_state_b(mkbuf(P.cur + stStride * i, 13 * sB), vb, sB);
    V3 ew, ev;
    se3_log(se3_compose(se3_inverse(So.X), Sn.X), ew, ev);
    double e[12] = {ew.x, ew.y, ew.z, ev.x, ev.y, ev.z, Sn.w.x - So.w.x, Sn.w.y - So.w.y, Sn.w.z - So.w.z,
                    Sn.v.x - So.v.x, Sn.v.y - So.v.y, Sn.v.z - So.v.z};
    double u[M], un[M], du[M];
#pragma unroll
    for (int a = 0; a < M; a++) {
      double sacc = alpha * bld(rG, GK_VG(b, M), GOFF(a, 12, M));
#pragma unroll
      for (int k = 0; k < 12; k++) sacc += bld(rG, GK_VG(b, M), GOFF(a, k, M)) * e[k];
      u[a] = bld(rU, vb, a * sB);
      du[a] = sacc;
      un[a] = u[a] + sacc;
    }
    J += knot_cost<M>(P, C, i, b, Sn, un, false);
    State Nx;
    if constexpr (!LINEAR) {
      State Fn = dyn_f<M>(C, Sn, un);
      if constexpr (MS) {
        double d[12];
#pragma unroll
        for (int a = 0; a < 12; a++) d[a] = bld(rR, REC_VR(b), FOFF(REC_D + a));
        // the factors from the stored defect alone (note in roll_step): x_{i+1} Exp((alpha - 1) d_q) x_{i+1}^-1, (alpha - 1) d_xi
        const State Sx = load_state_b(mkbuf(P.cur + stStride * (i + 1), 13 * sB), vb, sB);
        const double am1 = alpha - 1.0;
        const Pose Mx = se3_compose(se3_compose(Sx.X, se3_exp_fast(am1 * v3(d[0], d[1], d[2]), am1 * v3(d[3], d[4], d[5]))),
                                    se3_inverse(Sx.X));
        Nx.X = se3_project(se3_compose(Mx, Fn.X));
        Nx.w = am1 * v3(d[6], d[7], d[8]) + Fn.w;
        Nx.v = am1 * v3(d[9], d[10], d[11]) + Fn.v;
      } else {
        Nx = Fn;  // SS: x^_{i+1} = f(x^_i, u^_i) (traopt_controller.py:2073-2080)
      }
      if constexpr (MS) {  // new defect Log(x^_{i+1}^-1 f_q(x^_i,u^_i)), f_xi - xi^_{i+1}
        V3 dw, dv;
        se3_log(se3_compose(se3_inverse(Nx.X), Fn.X), dw, dv);
        V3 xw = Fn.w - Nx.w, xv = Fn.v - Nx.v;
        d2 += dot(dw, dw) + dot(dv, dv) + dot(xw, xw) + dot(xv, xv);
      }
    } else {
      double lin[12], d[12];
      fx_apply<M>(P, C, i, b, e, du, lin);
#pragma unroll
      for (int a = 0; a < 12; a++) d[a] = MS ? alpha * bld(rR, REC_VR(b), FOFF(REC_D + a)) : 0.0;
      State Sx = load_state_b(mkbuf(P.cur + stStride * (i + 1), 13 * sB), vb, sB);
      Pose D = se3_exp(v3(lin[0] + d[0], lin[1] + d[1], lin[2] + d[2]), v3(lin[3] + d[3], lin[4] + d[4], lin[5] + d[5]));
      Nx.X = se3_project(se3_compose(Sx.X, D));
      Nx.w = Sx.w + v3(lin[6] + d[6], lin[7] + d[7], lin[8] + d[8]);
      Nx.v = Sx.v + v3(lin[9] + d[9], lin[10] + d[10], lin[11] + d[11]);
      if constexpr (MS) {
        State Fn = dyn_f<M>(C, Sn, un);
        V3 dw, dv;
        se3_log(se3_compose(se3_inverse(Nx.X), Fn.X), dw, dv);
        V3 xw = Fn.w - Nx.w, xv = Fn.v - Nx.v;
        d2 += dot(dw, dw) + dot(dv, dv) + dot(xw, xw) + dot(xv, xv);
      }
    }
    __amdgpu_buffer_rsrc_t rSU = mkbuf(su + uStride * i, M * sB);
#pragma unroll
    for (int a = 0; a < M; a++) bst(rSU, vb, a * sB, un[a]);
    store_state_b(mkbuf(sx + stStride * (i + 1), 13 * sB), vb, sB, Nx);
    Sn = Nx;
  }
  double uz[M];
#pragma unroll
  for (int a = 0; a < M; a++) uz[a] = 0;
  J += knot_cost<M>(P, C, N, b, Sn, uz, true);
  P.Jtrial[(size_t)b * 20 + ai] = J;
  P.dtrial[(size_t)b * 20 + ai] = sqrt(d2);
}

// ---- line-search stages, round 3 form -----------------------------------------------------------------------------
// A stage is three launches: (1) k_rollout_ls -- the closed-loop rollouts alone, the quad form of K3 (roll_step with
// STORE), one quad per (undecided trajectory, alpha), the undecided trajectories taken from a compacted list so that a
// wide stage of a few stragglers is a few waves and not a sweep over the batch; (2) k_ls_eval -- stage costs and (MS)
// squared defects of the stored candidates, one thread per (trajectory, knot, alpha): none of it depends on the
// rollout chain, on which the round-2 kernel (k_rollout_eval, retired) carried it (a second Log, a cost with its own Log, per knot: stage 1 took
// 0.68 ms SS / 1.10 ms MS against 0.33 ms for the bare rollout); (3) k_ls_sum -- the sums in knot order, the order of
// _trajectory_cost / _compute_defect_norm (traopt_controller.py:2742-2754, :2790-2821).
template <int M, bool MS, bool LINEAR, int PK>
__global__ __launch_bounds__(64) void k_rollout_ls(Params P, int a0, int nslots, int direct, int list) {
  // constants as in K3 (k_rollout): the address-space-4 view for the nonlinear step, the generic pointer for the linear
  // one (fx_apply reads F_u's constants in the knot loop: note at DConsts)
  typedef typename std::conditional<LINEAR, Consts, DConsts>::type CT;
  const CT& C = *(const CT*)P.c;
  const int t = blockIdx.x * 64 + threadIdx.x, slot = blockIdx.y;
  const int quad = t >> 2, q = t & 3;
  if (slot >= nslots) return;
  int b;
  bool live;
  if (list >= 0) {
    // the undecided trajectories of the previous stage's select (k_ls_select), in no particular order; one alpha per
    // wave.  (Four trajectories x four alphas per wave, so that the four quads of a trajectory share their gain and
    // nominal-state loads, was measured: slower -- SS 343 -> 307 it/s; a wave then mixes step sizes whose rollouts take
    // different branches of the series gates.)
    const int n = P.ls_count[list];
    if (n * nslots > LS_QUAD_MAX) return;  // the thread form's turn (ls_quad_form)
    if ((quad & ~15) >= n) return;  // wave-uniform: none of this wave's 16 quads has work
    live = quad < n;
    b = P.ls_list[(size_t)list * P.Bp + (live ? quad : n - 1)];  // idle quads replay the last entry and store nothing
  } else {
    live = quad < P.Bp;
    b = live ? quad : P.Bp - 1;
    if (!P.active[b] || P.ls_accept[b] >= 0) return;  // quad-uniform
  }
  if (LINEAR && P.affine && !P.ec_redo[b]) return;  // (quad-uniform) its candidates come from the affine recursion
  const bool writer = live && q == 0;
  const int N = P.N, ai = a0 + slot;
  const double alpha = ls_alpha_k(ai);
  const unsigned sB = (unsigned)P.Bp * 8u, vb = (unsigned)b * 8u;
  // where the candidate goes: the candidate arrays at b (a one-alpha stage), or slot `slot` at the trajectory's POSITION
  // on the list -- dense, so that the stores of a wave fill whole lines and k_ls_eval reads whole lines (kept at b, a
  // few hundred undecided trajectories scattered over the batch cost eight times their bytes, here and in the evaluation)
  const unsigned vs = (list >= 0) ? (unsigned)(live ? quad : 0) * 8u : vb;
  const size_t stStride = (size_t)13 * P.Bp, uStride = (size_t)M * P.Bp;
  double* sx = direct ? P.cand : P.slot_x + (size_t)slot * stStride * (N + 1);
  double* su = direct ? P.cand_u : P.slot_u + (size_t)slot * uStride * N;
  State Sn = load_state_b(mkbuf(P.cur, 13 * sB), vb, sB);
  if (writer) store_state_b(mkbuf(sx, 13 * sB), vs, sB, Sn);
  const DynK DK = dynk_load(*P.c);  // generic pointer: see the note at DConsts
#ifdef TOLG_STAMPS
  RStamps ST;
  for (int k = 0; k < 8; k++) ST.acc[k] = 0;
  ST.t = __builtin_amdgcn_s_memtime();
#endif
  double un[M];
  auto store = [&](int i) {  // u^_i and x^_{i+1}, at the candidate's place (vs)
    if (writer) {
      __amdgpu_buffer_rsrc_t rSU = mkbuf(su + uStride * i, M * sB);
#pragma unroll
      for (int a = 0; a < M; a++) bst(rSU, vs, a * sB, un[a]);
      store_state_b(mkbuf(sx + stStride * (i + 1), 13 * sB), vs, sB, Sn);
    }
  };
  // K3's loop: two steps per trip, the nominal state fetched two knots ahead into ping-pong registers.  Single shooting
  // steps x^+ = f(x^, u^) for every alpha (:2073-2080): roll_step's ALPHA1 form
  // The merit search's step (alpha < 1: roll_step's factor form) also needs x_{i+1} and the stored defect d_i of the NOMINAL
  // trajectory: x_{i+1} is the state this loop fetches ahead anyway, d_i rides in a second ping-pong pair fetched with it
  // (round 4: requested inside the step, behind the gain product, their latency sat on the chain of every knot)
  constexpr bool FACT = MS && !LINEAR;
  const size_t recStride = (size_t)P.recF * P.Bp;
  double dA[FACT ? 12 : 1], dB[FACT ? 12 : 1];
  auto load_d = [&](int i, double (&d)[FACT ? 12 : 1]) {
    if constexpr (FACT) {
      __amdgpu_buffer_rsrc_t rR = mkbuf(P.REC + recStride * i, (unsigned)P.recF * sB);
#pragma unroll
      for (int a = 0; a < 12; a++) d[a] = bld(rR, REC_VR(b), FOFF(REC_D + a));
    }
  };
  auto mkpre = [&](const double (&d)[FACT ? 12 : 1], const State& Sx, bool have_x) {
    RollPre pr;
    if constexpr (FACT) {
      pr.have_d = true; pr.have_x = have_x; pr.Sx = Sx;
#pragma unroll
      for (int a = 0; a < 12; a++) pr.d[a] = d[a];
    }
    return pr;
  };
  State Sa = roll_load_state(P, 0, vb, sB), Sb = Sa;
  load_d(0, dA);
  for (int i = 0; i < N; i += 2) {
    if (i + 1 < N) { Sb = roll_load_state(P, i + 1, vb, sB); load_d(i + 1, dB); }
    __builtin_amdgcn_sched_barrier(0);
    Sn = roll_step<M, LINEAR, !MS, PK, false>(P, C, DK, i, b, q, writer, vb, sB, alpha, Sa, Sn, un,
                                               [&](RollIn<M>& R) { roll_load<M, !MS>(P, i, b, q, vb, sB, R); }, nullptr RST_ARG,
                                               mkpre(dA, Sb, i + 1 < N));
    store(i);
    if (i + 1 >= N) break;
    if (i + 2 < N) { Sa = roll_load_state(P, i + 2, vb, sB); load_d(i + 2, dA); }
    __builtin_amdgcn_sched_barrier(0);
    Sn = roll_step<M, LINEAR, !MS, PK, false>(P, C, DK, i + 1, b, q, writer, vb, sB, alpha, Sb, Sn, un,
                                               [&](RollIn<M>& R) { roll_load<M, !MS>(P, i + 1, b, q, vb, sB, R); }, nullptr RST_ARG,
                                               mkpre(dB, Sa, i + 2 < N));
    store(i + 1);
  }
}
// ---- ... and the same rollouts in TWO wavefronts per sixteen quads (round 4).  A stage of the merit search that rolls out four
// step sizes of the thousand trajectories the first one did not settle is 250 waves on 1024 SIMDs: its time is the length of
// ONE dependent chain, 200 steps of ~2.9 us (k_rollout_ls) -- and a step's chain runs through two groups of transcendental
// functions that do not depend on each other: the Log of the deviation from the nominal state (-> the control, -> the next
// TWIST) and the Exp of the twist (-> the next POSE).  As in the fused launch (k_rollout_lin) one wavefront carries the twist
// chain and a second one the pose chain, one step ahead, twist and pose handed over through a four-slot LDS ring; the merit
// search's factors (roll_step's note: x_{i+1} Exp((alpha - 1) d_q) x_{i+1}^-1 on the pose, (alpha - 1) d_xi on the twist) are
// functions of the nominal trajectory alone and are formed off the chain, while the wave waits for the other one.
// Inputs come from HBM as in k_rollout_ls (gains requested at the top of a step, nominal state and defect a knot ahead): the
// trajectories of a list are scattered over the batch, there is nothing for an LDS-DMA loader to stream.
// Synchronisation: sync[0] = twists published (knots 0 .. sync[0]), sync[1] = poses published; data, then counter, by LDS
// operations of one wave, which execute in order.  The twist wave's step i needs pose i (pose wave's step i - 1), the pose wave's
// step i needs twist i (twist wave's step i - 1): each is at most one step ahead of the other, so a slot written at step i
// (knot i + 1) replaces knot i - 3, which both have long read.  Every poll is bounded (TOLG_ST_INTERNAL instead of a hang).
// FACT: the merit search's step for alpha < 1; otherwise x^+ = f(x^, u^) (single shooting, and alpha = 1 of both searches).
// NT twist waves share ONE pose wave (a pose lane per quad: the pose chain has no product to split over a quad's lanes): a workgroup
// of NT = 3 twist waves + the pose wave carries 48 quads on the four SIMDs of a CU -- two waves per sixteen quads (NT = 1) put
// a pose wave that mostly waits on every second SIMD, which a stage that fills the chip several times over (single shooting's
// twelve step sizes) pays for in full.
enum { L2_RING = 4 };
// (two waves per SIMD -- __launch_bounds__(128, 2) on the NT = 1 form: 256 registers, 220-340 bytes of scratch on the chain -- was
// measured: merit 554 -> 414 it/s, SS 450 -> 308)
template <int M, bool FACT, int NT, int PK = 0>   // PK: 1 = Pendulum3dDyanmics (dyn_f)
__global__ __launch_bounds__(64 * (NT + 1)) void k_rollout_ls2(Params P, int a0, int nslots, int direct, int list) {
  constexpr int NQ = 16 * NT;  // quads per workgroup
  const DConsts& C = *(const DConsts*)P.c;
  __shared__ f64x2 ring[L2_RING][10][NQ];  // [slot][pair][quad]: pairs 0..3 the pose (quaternion, translation, padding), 4..6 the twist,
                                           // 7..9 the control that led to it (u^_{i-1} in the slot of knot i)
  __shared__ int sync[NT + 2];            // twists published by twist wave k (0 .. NT-1), poses published (NT), any live quad (NT + 1)
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const bool pose_wave = wave == NT;
  const int slot = blockIdx.y;
  // this lane's quad: a twist wave's lanes form sixteen quads of four, a pose-wave lane IS a quad (lanes past NQ idle along)
  const int tq = pose_wave ? lane : wave * 16 + (lane >> 2);
  const int tt = tq < NQ ? tq : NQ - 1;
  const int quad = blockIdx.x * NQ + tt, q = pose_wave ? 0 : (lane & 3);
  if (slot >= nslots) return;
  // (every return up to the barriers is taken by the whole workgroup)
  int b;
  bool live;
  if (list >= 0) {
    const int n = P.ls_count[list];
    if (n * nslots > LS_QUAD_MAX) return;  // the thread form's turn (ls_quad_form)
    if ((int)blockIdx.x * NQ >= n) return;
    live = quad < n && tq < NQ;
    b = P.ls_list[(size_t)list * P.Bp + (quad < n ? quad : n - 1)];  // idle quads replay the last entry and store nothing
  } else {
    if ((int)blockIdx.x * NQ >= P.Bp) return;
    b = quad < P.Bp ? quad : P.Bp - 1;
    live = quad < P.Bp && tq < NQ && P.active[b] && P.ls_accept[b] < 0;  // (decided or finished trajectories compute along and store nothing)
  }
  if (threadIdx.x < NT + 2) sync[threadIdx.x] = 0;
  __syncthreads();
  const rl_sync_t vsy = (rl_sync_t)sync;
  if (list < 0) {  // (kernel-uniform branch) a workgroup none of whose quads is undecided leaves
    if (__any(live) && lane == 0) vsy[NT + 1] = 1;
    __syncthreads();
    if (vsy[NT + 1] == 0) return;
  }
  const bool writer = live && q == 0;
  const int N = P.N, ai = a0 + slot;
  const double alpha = ls_alpha_k(ai), am1 = alpha - 1.0;
  const unsigned sB = (unsigned)P.Bp * 8u, vb = (unsigned)b * 8u;
  const unsigned vs = (list >= 0) ? (unsigned)(live ? quad : 0) * 8u : vb;  // where the candidate goes (k_rollout_ls)
  const size_t stStride = (size_t)13 * P.Bp, uStride = (size_t)M * P.Bp, recStride = (size_t)P.recF * P.Bp;
  double* sx = direct ? P.cand : P.slot_x + (size_t)slot * stStride * (N + 1);
  double* su = direct ? P.cand_u : P.slot_u + (size_t)slot * uStride * N;
  const DynK DK = dynk_load(*P.c);  // generic pointer: see the note at DConsts
  if (!pose_wave) {
    // ---------------- a twist chain: xi^_{i+1} from x^_i (pose from the pose wave), u^_i on the way
    State Sn = load_state_b(mkbuf(P.cur, 13 * sB), vb, sB);  // x^_0 = x_0
    double dA[FACT ? 6 : 1], dB[FACT ? 6 : 1];
    auto load_d = [&](int i, double (&d)[FACT ? 6 : 1]) {  // the twist half of the stored defect
      if constexpr (FACT) {
        __amdgpu_buffer_rsrc_t rR = mkbuf(P.REC + recStride * i, (unsigned)P.recF * sB);
#pragma unroll
        for (int a = 0; a < 6; a++) d[a] = bld(rR, REC_VR(b), FOFF(REC_D + 6 + a));
      }
    };
    bool ok = true;
    auto step = [&](int i, const State& So, const double (&d)[FACT ? 6 : 1]) {
      RollIn<M> R;
      roll_load<M, !FACT>(P, i, b, q, vb, sB, R);  // in flight while the pose arrives and Log runs
      __builtin_amdgcn_sched_barrier(0);
      if (i > 0) {
        if (!rl_wait_ge(vsy + NT, i)) { ok = false; return; }
        asm volatile("" ::: "memory");
        const f64x2 a = ring[i % L2_RING][0][tt], bq = ring[i % L2_RING][1][tt], c = ring[i % L2_RING][2][tt], d3 = ring[i % L2_RING][3][tt];
        Sn.X.q.x = a.x; Sn.X.q.y = a.y; Sn.X.q.z = bq.x; Sn.X.q.w = bq.y;
        Sn.X.t = v3(c.x, c.y, d3.x);
      }
      V3 ew, ev;
      const Pose Dx = se3_compose(se3_inverse(So.X), Sn.X);
      const double yl = quat_vec2(Dx.q);
      se3_log_fast(Dx, ew, ev, series_gate(log_small(yl), log_dom(yl)));
      const double e[12] = {ew.x, ew.y, ew.z, ev.x, ev.y, ev.z, Sn.w.x - So.w.x, Sn.w.y - So.w.y, Sn.w.z - So.w.z,
                            Sn.v.x - So.v.x, Sn.v.y - So.v.y, Sn.v.z - So.v.z};
      double mine[2];
#pragma unroll
      for (int sidx = 0; sidx < 2; sidx++) {
        double sacc = alpha * R.G[sidx][12];
#pragma unroll
        for (int k = 0; k < 12; k++) sacc += R.G[sidx][k] * e[k];
        mine[sidx] = sacc;
      }
      double un[M], du[M];
      du[0] = quad_bcast<0>(mine[0]); du[1] = quad_bcast<0>(mine[1]);
      du[2] = quad_bcast<1>(mine[0]); du[3] = quad_bcast<1>(mine[1]);
      if constexpr (M == 6) { du[4] = quad_bcast<2>(mine[0]); du[5] = quad_bcast<2>(mine[1]); }
#pragma unroll
      for (int a = 0; a < M; a++) un[a] = R.u[a] + du[a];
      State Fn;
      if (DK.diag) dyn_twist_k<M, DConsts, PK>(DK, C, Sn, un, Fn);
      else Fn = dyn_f<M, DConsts, PK>(C, Sn, un);
      if constexpr (FACT) {  // xi^_{i+1} = (alpha - 1) d_xi + f_xi(x^_i, u^_i)   (roll_step)
        Sn.w = am1 * v3(d[0], d[1], d[2]) + Fn.w;
        Sn.v = am1 * v3(d[3], d[4], d[5]) + Fn.v;
      } else {
        Sn.w = Fn.w;
        Sn.v = Fn.v;
      }
      // (no global store on this wave: the pose wave, which waits most of the time, stores the whole candidate -- a store in
      // the memory queue makes every counted wait for a prefetched load a wait for the store as well)
      if (q == 0) {
        const int s1 = (i + 1) % L2_RING;
        ring[s1][4][tt] = f64x2{Sn.w.x, Sn.w.y}; ring[s1][5][tt] = f64x2{Sn.w.z, Sn.v.x}; ring[s1][6][tt] = f64x2{Sn.v.y, Sn.v.z};
#pragma unroll
        for (int a = 0; a < M; a += 2) ring[s1][7 + a / 2][tt] = f64x2{un[a], un[a + 1]};
      }
      asm volatile("" ::: "memory");
      __builtin_amdgcn_wave_barrier();
      if (lane == 0) vsy[wave] = i + 1;
    };
    State Sa = roll_load_state(P, 0, vb, sB), Sb = Sa;
    load_d(0, dA);
    for (int i = 0; ok && i < N; i += 2) {
      if (i + 1 < N) { Sb = roll_load_state(P, i + 1, vb, sB); load_d(i + 1, dB); }
      __builtin_amdgcn_sched_barrier(0);
      step(i, Sa, dA);
      if (!ok || i + 1 >= N) break;
      if (i + 2 < N) { Sa = roll_load_state(P, i + 2, vb, sB); load_d(i + 2, dA); }
      __builtin_amdgcn_sched_barrier(0);
      step(i + 1, Sb, dB);
    }
    if (!ok && writer) P.status[b] = TOLG_ST_INTERNAL;
    return;
  }
  // ---------------- the pose chains: X^_{i+1} = [M_i] project(X^_i Exp(xi^_i dt)), M_i = X_{i+1} Exp((alpha - 1) d_q) X_{i+1}^-1
  const bool mine = tq < NQ;
  State S = load_state_b(mkbuf(P.cur, 13 * sB), vb, sB);
  if (writer) store_state_b(mkbuf(sx, 13 * sB), vs, sB, S);
  // twist of x^_i and the control u^_{i-1} that led to it, from slot i of the ring to the candidate's place
  auto store_twist_u = [&](int i, V3 w, V3 v) {
    if (writer) {
      __amdgpu_buffer_rsrc_t rSU = mkbuf(su + uStride * (i - 1), M * sB), rSX = mkbuf(sx + stStride * i, 13 * sB);
#pragma unroll
      for (int a = 0; a < M; a += 2) {
        const f64x2 u2 = ring[i % L2_RING][7 + a / 2][tt];
        bst(rSU, vs, a * sB, u2.x); bst(rSU, vs, (a + 1) * sB, u2.y);
      }
      bst(rSX, vs, 7 * sB, w.x); bst(rSX, vs, 8 * sB, w.y); bst(rSX, vs, 9 * sB, w.z);
      bst(rSX, vs, 10 * sB, v.x); bst(rSX, vs, 11 * sB, v.y); bst(rSX, vs, 12 * sB, v.z);
    }
  };
  struct NomQ { Pose X; double d[6]; };
  auto load_nom = [&](int i, NomQ& n) {  // pose of the nominal x_{i+1} and the pose half of the stored defect d_i
    if constexpr (FACT) {
      __amdgpu_buffer_rsrc_t rX = mkbuf(P.cur + stStride * (i + 1), 13 * sB), rR = mkbuf(P.REC + recStride * i, (unsigned)P.recF * sB);
      n.X.q.x = bld(rX, vb, 0); n.X.q.y = bld(rX, vb, sB); n.X.q.z = bld(rX, vb, 2 * sB); n.X.q.w = bld(rX, vb, 3 * sB);
      n.X.t = v3(bld(rX, vb, 4 * sB), bld(rX, vb, 5 * sB), bld(rX, vb, 6 * sB));
#pragma unroll
      for (int a = 0; a < 6; a++) n.d[a] = bld(rR, REC_VR(b), FOFF(REC_D + a));
    }
  };
  NomQ nA, nB;
  load_nom(0, nA);
  bool ok = true;
  auto pstep = [&](int i, const NomQ& n) {
    Pose Mx;
    if constexpr (FACT)
      Mx = se3_compose(se3_compose(n.X, se3_exp_fast(am1 * v3(n.d[0], n.d[1], n.d[2]), am1 * v3(n.d[3], n.d[4], n.d[5]))), se3_inverse(n.X));
    if (i > 0) {
#pragma unroll
      for (int k = 0; k < NT; k++)
        if (!rl_wait_ge(vsy + k, i)) { ok = false; return; }
      asm volatile("" ::: "memory");
      const f64x2 e = ring[i % L2_RING][4][tt], f = ring[i % L2_RING][5][tt], g = ring[i % L2_RING][6][tt];
      S.w = v3(e.x, e.y, f.x);
      S.v = v3(f.y, g.x, g.y);
    }
    Pose F = dyn_pose_k(DK, S);
    if constexpr (FACT) F = se3_project(se3_compose(Mx, F));
    S.X = F;
    if (mine) {
      const int s1 = (i + 1) % L2_RING;
      ring[s1][0][tt] = f64x2{S.X.q.x, S.X.q.y}; ring[s1][1][tt] = f64x2{S.X.q.z, S.X.q.w};
      ring[s1][2][tt] = f64x2{S.X.t.x, S.X.t.y}; ring[s1][3][tt] = f64x2{S.X.t.z, 0.0};
    }
    asm volatile("" ::: "memory");
    __builtin_amdgcn_wave_barrier();
    if (lane == 0) vsy[NT] = i + 1;
    if (writer) {
      __amdgpu_buffer_rsrc_t rSX = mkbuf(sx + stStride * (i + 1), 13 * sB);
      bst(rSX, vs, 0, S.X.q.x); bst(rSX, vs, sB, S.X.q.y); bst(rSX, vs, 2 * sB, S.X.q.z); bst(rSX, vs, 3 * sB, S.X.q.w);
      bst(rSX, vs, 4 * sB, S.X.t.x); bst(rSX, vs, 5 * sB, S.X.t.y); bst(rSX, vs, 6 * sB, S.X.t.z);
    }
    if (i > 0) store_twist_u(i, S.w, S.v);  // (behind the publish: off the chain)
  };
  for (int i = 0; ok && i < N; i += 2) {
    if (i + 1 < N) load_nom(i + 1, nB);
    __builtin_amdgcn_sched_barrier(0);
    pstep(i, nA);
    if (!ok || i + 1 >= N) break;
    if (i + 2 < N) load_nom(i + 2, nA);
    __builtin_amdgcn_sched_barrier(0);
    pstep(i + 1, nB);
  }
  if (ok) {  // the last twist and control
#pragma unroll
    for (int k = 0; k < NT; k++)
      if (!rl_wait_ge(vsy + k, N)) return;  // (the twist wave reports it)
    asm volatile("" ::: "memory");
    const f64x2 e = ring[N % L2_RING][4][tt], f = ring[N % L2_RING][5][tt], g = ring[N % L2_RING][6][tt];
    store_twist_u(N, v3(e.x, e.y, f.x), v3(f.y, g.x, g.y));
  }
}
// stage cost l(x^_i, u^_i) (traopt_cost.py:675-738) and, MS, the squared defect
// |Log(x^_{i+1}^-1 f_q(x^_i, u^_i))|^2 + |f_xi - xi^_{i+1}|^2 (:2790-2812) of every stored candidate of the stage
// (four waves per SIMD -- 127 registers and 28 bytes of scratch instead of 136 -- was measured: no change)
template <int M, bool MS>
__global__ __launch_bounds__(256) void k_ls_eval(Params P, int nslots, int direct, int list) {
  const Consts& C = *P.c;
  const int slot = blockIdx.y, N = P.N;
  // block -> (knot, 256 candidates) with the KNOT as the fast index over consecutive blocks.  On a list only the first
  // ceil(n / 256) blocks of a knot's row have work; with the row as the fast index (16 blocks per knot at Bp = 4096) those are
  // the same residues mod 8 for every knot, i.e. the same few XCDs: the 4-alpha stage of the merit search (n ~ 1000) ran its
  // 4000 x 201 evaluations on half the chip, 0.168 ms against 0.052 for the same count on the flags
  const int i = (int)(blockIdx.x % (unsigned)(N + 1));
  // e: where the candidate lies (its position on the list, or b itself for a stage that runs on the flags)
  const int e = (int)(blockIdx.x / (unsigned)(N + 1)) * 256 + (int)threadIdx.x;
  if (slot >= nslots || e >= P.Bp) return;
  int b = e;
  if (list >= 0) {
    const int n = P.ls_count[list];
    if (n * nslots > LS_QUAD_MAX || e >= n) return;
    b = P.ls_list[(size_t)list * P.Bp + e];
  } else if (!P.active[b] || P.ls_accept[b] >= 0) return;
  if (P.affine && !P.ec_redo[b]) return;  // k_ls_eval_affine's
  const size_t stStride = (size_t)13 * P.Bp, uStride = (size_t)M * P.Bp;
  const double* sx = direct ? P.cand : P.slot_x + (size_t)slot * stStride * (N + 1);
  const double* su = direct ? P.cand_u : P.slot_u + (size_t)slot * uStride * N;
  const State S = load_state(P, sx, i, e);
  double u[M];
#pragma unroll
  for (int a = 0; a < M; a++) u[a] = (i < N) ? su[UIDX(a, i, e)] : 0.0;
  P.LSC[((size_t)slot * (N + 1) + i) * P.Bp + e] = knot_cost<M>(P, C, i, b, S, u, i == N);
  if constexpr (MS) {
    if (i < N) {
      const State Nx = load_state(P, sx, i + 1, e), Fn = dyn_f<M>(C, S, u);
      V3 dw, dv;
      se3_log(se3_compose(se3_inverse(Nx.X), Fn.X), dw, dv);
      const V3 xw = Fn.w - Nx.w, xv = Fn.v - Nx.v;
      P.LSD[((size_t)slot * N + i) * P.Bp + e] = dot(dw, dw) + dot(dv, dv) + dot(xw, xw) + dot(xv, xv);
    }
  }
}
// affine_part: the sums of k_ls_eval_affine's trajectories (every list length: there is no thread form for them); otherwise
// the stored candidates' -- in a solve that runs the affine path, only the trajectories that path handed back
template <bool MS>
__global__ void k_ls_sum(Params P, int a0, int nslots, int list, int affine_part) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  const int e = t % P.Bp, slot = t / P.Bp, N = P.N;
  if (slot >= nslots) return;
  int b = e;
  if (list >= 0) {
    const int n = P.ls_count[list];
    if ((!affine_part && n * nslots > LS_QUAD_MAX) || e >= n) return;
    b = P.ls_list[(size_t)list * P.Bp + e];
  } else if (!P.active[b] || P.ls_accept[b] >= 0) return;
  if (affine_part ? (P.ec_redo[b] != 0) : (P.affine && !P.ec_redo[b])) return;
  const double* c = P.LSC + (size_t)slot * (N + 1) * P.Bp + e;
  const double* d = P.LSD + (size_t)slot * N * P.Bp + e;
  double J = 0, d2 = 0;
  int i = 0;
  for (; i + 16 <= N; i += 16) {  // fixed order; the loads go out sixteen knots at a time (k_reduce)
    double cc[16], dd[16];
#pragma unroll
    for (int k = 0; k < 16; k++) { cc[k] = c[(size_t)(i + k) * P.Bp]; dd[k] = MS ? d[(size_t)(i + k) * P.Bp] : 0.0; }
#pragma unroll
    for (int k = 0; k < 16; k++) { J += cc[k]; d2 += dd[k]; }
  }
  for (; i < N; i++) { J += c[(size_t)i * P.Bp]; if (MS) d2 += d[(size_t)i * P.Bp]; }
  J += c[(size_t)N * P.Bp];
  P.Jtrial[(size_t)b * 20 + a0 + slot] = J;
  P.dtrial[(size_t)b * 20 + a0 + slot] = sqrt(d2);
}

// ---- rollout = 'linear' without rollouts (round 4).  k_expected_change_ring<.., STORE> has left e_i and du_i of the alpha = 1
// linear rollout in P.ED; the candidate of step size alpha is x_i (+) alpha e_i, u_i + alpha du_i (note at that kernel).  One
// thread per (trajectory, knot, alpha) builds it where it is needed: k_ls_eval_affine for the stage costs and (MS) squared
// defects of a line-search stage -- k_ls_eval's outputs at k_ls_eval's places, so k_ls_sum / k_ls_select go on unchanged --,
// k_affine_commit for the accepted one (or, `all`, the alpha = 1 candidate of an accept-always iteration).  Trajectories the
// recursion handed back (ec_redo) are left to the statement-form kernels, which in such a solve touch nothing else.
template <int M>
TOLG_DEV void affine_candidate(const Params& P, int i, int b, double alpha, State& S, double (&u)[M]) {
  const State So = load_state(P, P.cur, i, b);
  const f64x2* ed = reinterpret_cast<const f64x2*>(P.ED + ((size_t)i * P.Bp + b) * 32);
  if (i == 0) {
    S = So;  // xs_new[0] = xs[0] (:2666)
  } else {
    // x_i (+) alpha e_i: the right-plus of :2730-2733 (q_next_mnf + tangent), re-normalised like every group operation here
    const f64x2 e0 = ed[0], e1 = ed[1], e2 = ed[2], e3 = ed[3], e4 = ed[4], e5 = ed[5];
    // (closed forms here and in k_ls_eval_affine: the series forms were measured slower in these one-thread-per-(trajectory,
    // knot, alpha) kernels -- linear merit 354 against 375 it/s -- for the reason noted at k_rollout_eval_t)
    const Pose D = se3_exp(alpha * v3(e0.x, e0.y, e1.x), alpha * v3(e1.y, e2.x, e2.y));
    S.X = se3_project(se3_compose(So.X, D));
    S.w = So.w + alpha * v3(e3.x, e3.y, e4.x);
    S.v = So.v + alpha * v3(e4.y, e5.x, e5.y);
  }
  if (i < P.N) {
#pragma unroll
    for (int a = 0; a < M; a += 2) {
      const f64x2 d2 = ed[8 + a / 2];
      u[a] = P.cur_u[UIDX(a, i, b)] + alpha * d2.x;
      u[a + 1] = P.cur_u[UIDX(a + 1, i, b)] + alpha * d2.y;
    }
  } else {
#pragma unroll
    for (int a = 0; a < M; a++) u[a] = 0.0;
  }
}
template <int M, bool MS>
__global__ __launch_bounds__(256) void k_ls_eval_affine(Params P, int a0, int nslots, int list) {
  const Consts& C = *P.c;
  const int slot = blockIdx.y, N = P.N;
  const int i = (int)(blockIdx.x % (unsigned)(N + 1));  // the knot as the fast index over blocks: note at k_ls_eval
  const int e = (int)(blockIdx.x / (unsigned)(N + 1)) * 256 + (int)threadIdx.x;
  if (slot >= nslots || e >= P.Bp) return;
  int b = e;
  if (list >= 0) {
    if (e >= P.ls_count[list]) return;
    b = P.ls_list[(size_t)list * P.Bp + e];
  } else if (!P.active[b] || P.ls_accept[b] >= 0) return;
  if (P.ec_redo[b]) return;
  const double alpha = ls_alpha_k(a0 + slot);
  State S;
  double u[M];
  affine_candidate<M>(P, i, b, alpha, S, u);
  P.LSC[((size_t)slot * (N + 1) + i) * P.Bp + e] = knot_cost<M>(P, C, i, b, S, u, i == N);
  if constexpr (MS) {
    if (i < N) {
      State Nx;
      double un[M];
      affine_candidate<M>(P, i + 1, b, alpha, Nx, un);
      const State Fn = dyn_f<M>(C, S, u);
      V3 dw, dv;
      se3_log(se3_compose(se3_inverse(Nx.X), Fn.X), dw, dv);
      const V3 xw = Fn.w - Nx.w, xv = Fn.v - Nx.v;
      P.LSD[((size_t)slot * N + i) * P.Bp + e] = dot(dw, dw) + dot(dv, dv) + dot(xw, xw) + dot(xv, xv);
    }
  }
}
template <int M>
__global__ __launch_bounds__(256) void k_affine_commit(Params P, int a0, int all) {
  const size_t t = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (t >= (size_t)(P.N + 1) * P.Bp) return;
  const int b = (int)(t % P.Bp), i = (int)(t / P.Bp);
  if (!P.active[b] || P.ec_redo[b]) return;
  double alpha = 1.0;
  if (!all) {
    const int s = P.ls_slot[b];
    if (s < 0) return;
    alpha = ls_alpha_k(a0 + s);
  }
  State S;
  double u[M];
  affine_candidate<M>(P, i, b, alpha, S, u);
  store_state(P, P.cand, i, b, S);
  if (i < P.N) {
#pragma unroll
    for (int a = 0; a < M; a++) P.cand_u[UIDX(a, i, b)] = u[a];
  }
}

// MS merit search preparation (traopt_controller.py:2550-2557): linear alpha = 1 rollout (not stored),
// _expected_cost_change (:2756-2769), _update_defect_weight (:2774-2788).  Four lanes per trajectory, the step is
// roll_step's linear form (round 2: one thread per trajectory, 3.9 ms per call -- half of a merit-search iteration).
// REDO: only the trajectories k_expected_change_ring (tolg_expected_change.h) handed back.
template <int M, int PK, bool REDO = false>
__global__ __launch_bounds__(64) void k_expected_change(Params P) {
  const Consts& C = *P.c;  // generic pointer (note at DConsts)
  const int t = blockIdx.x * 64 + threadIdx.x;
  int b = t >> 2;
  const int q = t & 3;
  const bool live = b < P.Bp;
  if (!live) b = P.Bp - 1;
  // the constant weight blocks of the quadratic model, in LDS: read from the constants in memory they were ~100 loads per
  // knot on a chain that is nothing but load latencies.  Filled by the first 36 lanes of the wave BEFORE any quad
  // leaves: a quad that is inactive (or, with REDO, not handed back) must still write its share of the table.
  __shared__ double sW2[36], sP2[36], sR2[36];
  if (threadIdx.x < 36) {
    sW2[threadIdx.x] = 2.0 * C.W2[threadIdx.x]; sP2[threadIdx.x] = 2.0 * C.P2[threadIdx.x];
    sR2[threadIdx.x] = (threadIdx.x < M * M) ? 2.0 * C.R[threadIdx.x] : 0.0;
  }
  __builtin_amdgcn_s_waitcnt(0xc07f);  // lgkmcnt(0): single-wave workgroup, LDS is in order
  __builtin_amdgcn_wave_barrier();
  if (!P.active[b]) return;  // quad-uniform
  if (REDO && !P.ec_redo[b]) return;
  const bool writer = live && q == 0;
  const int N = P.N;
  const unsigned sB = (unsigned)P.Bp * 8u, vb = (unsigned)b * 8u;
  const size_t recStride = (size_t)P.recF * P.Bp;
  State Sn = load_state_b(mkbuf(P.cur, 13 * sB), vb, sB);
  const DynK DK = dynk_load(C);
  double c1 = 0, c2 = 0;
  State So = roll_load_state(P, 0, vb, sB);
#ifdef TOLG_STAMPS
  RStamps ST;
  for (int k = 0; k < 8; k++) ST.acc[k] = 0;
  ST.t = __builtin_amdgcn_s_memtime();
#endif
  // l_x e and e^T l_xx e with l_xx = blkdiag(l_xx11, 2 W2)
  auto state_terms = [&](int i, const double (&e)[12]) {
    __amdgpu_buffer_rsrc_t rR = mkbuf(P.REC + recStride * i, (unsigned)P.recF * sB);
    const double* W2 = (i == N) ? sP2 : sW2;
#pragma unroll
    for (int a = 0; a < 12; a++) c1 += bld(rR, REC_VR(b), FOFF(REC_LX + a)) * e[a];
#pragma unroll
    for (int a = 0; a < 6; a++)
#pragma unroll
      for (int k = 0; k < 6; k++) {
        c2 += e[a] * bld(rR, REC_VR(b), FOFF(REC_LXX + sym6(a, k))) * e[k];
        c2 += e[6 + a] * W2[6 * a + k] * e[6 + k];
      }
  };
  for (int i = 0; i < N; i++) {
    State Sx = So;
    So = roll_load_state(P, i + 1, vb, sB);  // knot N too: the terminal deviation below
    __builtin_amdgcn_sched_barrier(0);
    double un[M];
    RollProbe<M> pr;
    const State Nx = roll_step<M, true, false, PK, false>(P, C, DK, i, b, q, writer, vb, sB, 1.0, Sx, Sn, un,
                                                          [&](RollIn<M>& R) { roll_load<M, false>(P, i, b, q, vb, sB, R); }, &pr RST_ARG);
    state_terms(i, pr.e);
    __amdgpu_buffer_rsrc_t rR = mkbuf(P.REC + recStride * i, (unsigned)P.recF * sB);
#pragma unroll
    for (int a = 0; a < M; a++) {
      c1 += bld(rR, REC_VR(b), FOFF(REC_LU + a)) * pr.du[a];
#pragma unroll
      for (int k = 0; k < M; k++) c2 += pr.du[a] * sR2[a * M + k] * pr.du[k];
      if (P.al_lb) c2 += pr.du[a] * bld(rR, REC_VR(b), FOFF(P.fLUU + a)) * pr.du[a];
    }
    Sn = Nx;
  }
  {  // terminal knot: deviation of the rolled-out state from the nominal one
    V3 ew, ev;
    se3_log(se3_compose(se3_inverse(So.X), Sn.X), ew, ev);
    const double e[12] = {ew.x, ew.y, ew.z, ev.x, ev.y, ev.z, Sn.w.x - So.w.x, Sn.w.y - So.w.y, Sn.w.z - So.w.z,
                          Sn.v.x - So.v.x, Sn.v.y - So.v.y, Sn.v.z - So.v.z};
    state_terms(N, e);
  }
  if (!writer) return;
  P.ecc[2 * b] = c1;
  P.ecc[2 * b + 1] = c2;
  double dn = P.dn[b], wprev = P.dweight[2 * b + 1], w;
  if (dn < ((so3_family(C.kind)) ? 1e-14 : 1e-12)) w = wprev;  // _defect_kappa (SE3 :2410, SO3 :1090)
  else w = fmax(10.0, 10.0 + fabs(c1 + 0.5 * c2) / ((1.0 - 0.5) * dn));
  P.dweight[2 * b] = w;
  P.dweight[2 * b + 1] = w;
}

#include "tolg_expected_change.h"

__global__ void k_ls_begin(Params P, int first_fit_iteration) {
  int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b < 2) P.ls_count[b] = 0;
  if (b >= P.Bp) return;
  P.ls_accept[b] = -1;
  P.ls_slot[b] = -1;
  P.ls_alpha[b] = 1.0;
  if (first_fit_iteration) { P.dweight[2 * b] = 10.0; P.dweight[2 * b + 1] = 10.0; }  // _defect_mu0
}

// first alpha (in the reference's order) of this stage that passes the acceptance test
// ... and the trajectories that stay undecided go on list `out` (compacted, for the next stage's rollouts)
template <bool MS>
__global__ void k_ls_select(Params P, int a0, int nslots, int out) {
  int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= P.Bp || !P.active[b] || P.ls_accept[b] >= 0) return;
  const double J0 = P.Jc[b], dn = P.dn[b];
  for (int s = 0; s < nslots; s++) {
    const int ai = a0 + s;
    const double alpha = ls_alpha_k(ai), Jn = P.Jtrial[(size_t)b * 20 + ai];
    P.ls_alpha[b] = alpha;  // "alpha" the callback sees is the last one tried (App. C-Q12)
    bool ok;
    if (MS) {
      const double w = P.dweight[2 * b], c1 = P.ecc[2 * b], c2 = P.ecc[2 * b + 1];
      const double merit = J0 + w * dn, merit_new = Jn + w * P.dtrial[(size_t)b * 20 + ai];
      const double Jexp = alpha * c1 + 0.5 * alpha * alpha * c2;
      ok = (merit_new - merit) < 0.05 * (Jexp - alpha * w * dn);
    } else {
      ok = Jn < J0;
    }
    if (ok) { P.ls_accept[b] = ai; P.ls_slot[b] = s; return; }
  }
  // (an ordered compaction -- one workgroup, a scan -- instead of the counter was measured: no gain, 405 -> 395 it/s;
  // waves take their turns at the counter nearly in order as it is)
  const int pos = atomicAdd(&P.ls_count[out], 1);
  P.ls_list[(size_t)out * P.Bp + pos] = b;
  P.ls_pos[(size_t)out * P.Bp + b] = pos;
}

// the accepted candidate of a wide stage -> the candidate arrays.  list / nslots: the stage's list and width -- the quad
// form kept its candidates by list position, the thread form by trajectory (ls_quad_form)
__global__ void k_ls_copy(Params P, int list, int nslots) {
  size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= (size_t)(P.N + 1) * P.Bp) return;
  const int b = (int)(t % P.Bp), i = (int)(t / P.Bp);
  const int s = P.ls_slot[b];
  if (s < 0 || !P.active[b]) return;
  if (P.affine && !P.ec_redo[b]) return;  // k_affine_commit's
  const int e = (list >= 0 && ls_quad_form(P, list, nslots)) ? P.ls_pos[(size_t)list * P.Bp + b] : b;
  const size_t stStride = (size_t)13 * P.Bp, uStride = (size_t)P.m * P.Bp;
  const double* sx = P.slot_x + (size_t)s * stStride * (P.N + 1);
  const double* su = P.slot_u + (size_t)s * uStride * P.N;
  for (int c = 0; c < 13; c++) P.cand[SIDX(c, i, b)] = sx[SIDX(c, i, e)];
  if (i < P.N)
    for (int c = 0; c < P.m; c++) P.cand_u[UIDX(c, i, b)] = su[UIDX(c, i, e)];
}
__global__ void k_ls_clear_slot(Params P, int reset_list) {
  int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b == 0) P.ls_count[reset_list] = 0;  // the list the NEXT stage's select fills (this stage's rollouts have read it)
  if (b < P.Bp) P.ls_slot[b] = -1;
}
// trajectories without an acceptable step: callback with the unchanged cost, then stop
// (traopt_controller.py:2621-2633, :1996-2007)
__global__ void k_ls_finish(Params P, int it) {
  int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= P.Bp || !P.active[b] || P.ls_accept[b] >= 0) return;
  P.status[b] = TOLG_ST_NODESCENT;
  P.active[b] = 0;
  P.iters[b] = it + 1;
  if (b >= P.B) return;
  if (P.J_hist) P.J_hist[(size_t)b * P.max_iter + it] = P.Jc[b];
  if (P.alpha_hist) P.alpha_hist[(size_t)b * P.max_iter + it] = P.ls_alpha[b];
  if (P.defect_hist) P.defect_hist[(size_t)b * (P.max_iter + 1) + it + 1] = P.dn[b];
}

// Augmented-Lagrangian outer update (AL_iLQR_Tracking_SE3_MS._al_update_param,
// traopt_controller.py:3270-3290) and the constraint evaluation of :3242-3250, one thread per
// trajectory: lambda <- max(0, lambda + I_mu g), mu <- min(mu_scale mu, mu_max),
// I_mu <- 0 where (g < 0 and lambda_new == 0) else mu_new; maxviol = max g over all knots.
__global__ void k_al_update(int B, int N, int m, const double* __restrict__ us, const double* __restrict__ lb,
                            const double* __restrict__ ub, double* __restrict__ lam, double* __restrict__ imu,
                            double* __restrict__ mu, double mu_scale, double mu_max, double tol_constr,
                            double* __restrict__ maxviol, int* __restrict__ al_conv) {
  int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= B || al_conv[b]) return;  // a converged problem keeps its multipliers
  double mv = 0.0;                   // the terminal knot contributes g = 0
  for (int i = 0; i < N; i++) {
    const double* u = us + ((size_t)b * N + i) * m;
    for (int k = 0; k < 2 * m; k++) mv = fmax(mv, (k < m) ? lb[k] - u[k] : u[k - m] - ub[k - m]);
  }
  maxviol[b] = mv;
  if (mv < tol_constr) { al_conv[b] = 1; return; }  // traopt_controller.py:3250, :3262-3263
  const double mu_new = fmin(mu[b] * mu_scale, mu_max);
  for (int i = 0; i < N; i++) {
    const double* u = us + ((size_t)b * N + i) * m;
    double* l = lam + ((size_t)b * N + i) * 2 * m;
    double* im = imu + ((size_t)b * N + i) * 2 * m;
    for (int k = 0; k < 2 * m; k++) {
      double g = (k < m) ? lb[k] - u[k] : u[k - m] - ub[k - m];
      double ln = fmax(0.0, l[k] + im[k] * g);
      l[k] = ln;
      im[k] = (g < 0.0 && ln == 0.0) ? 0.0 : mu_new;
    }
  }
  mu[b] = mu_new;
}

// ---- single-knot probe (tolg_eval_knot): the reference's per-knot plugin methods f, f_x, f_u, l, l_x,
// l_u, l_xx, l_uu, _err evaluated for n states at knot i
__global__ void k_probe_pack(Params P, int i, const double* __restrict__ x_q, const double* __restrict__ x_xi,
                             const double* __restrict__ u) {
  int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= P.Bp) return;
  int bs = b < P.B ? b : P.B - 1;
  State S;
  S.X = pose_from_m16(x_q + 16 * (size_t)bs);
  const double* x = x_xi + 6 * (size_t)bs;
  S.w = v3(x[0], x[1], x[2]);
  S.v = v3(x[3], x[4], x[5]);
  store_state(P, P.cur, i, b, S);
  if (i < P.N)
    for (int a = 0; a < P.m; a++) P.cur_u[UIDX(a, i, b)] = u ? u[(size_t)bs * P.m + a] : 0.0;
  P.active[b] = 1;
}
__global__ void k_probe_export(Params P, int i, double* __restrict__ f_q, double* __restrict__ f_xi,
                               double* __restrict__ Fx, double* __restrict__ Fu, double* __restrict__ l,
                               double* __restrict__ lx, double* __restrict__ lxx, double* __restrict__ lu,
                               double* __restrict__ luu, double* __restrict__ err) {
  const Consts& C = *P.c;
  int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= P.B) return;
  const bool term = (i == P.N);
  const int m = P.m;
  if (l) l[b] = P.SC[(size_t)i * P.Bp + b];
  if (lx) for (int r = 0; r < 12; r++) lx[(size_t)b * 12 + r] = P.REC[RIDX(i, REC_LX + r, b)];
  if (lxx) {
    const double* W2 = term ? C.P2 : C.W2;
    double* L = lxx + (size_t)b * 144;
    for (int k = 0; k < 144; k++) L[k] = 0;
    for (int r = 0; r < 6; r++)
      for (int c = 0; c < 6; c++) {
        L[12 * r + c] = P.REC[RIDX(i, REC_LXX + sym6(r, c), b)];
        L[12 * (r + 6) + c + 6] = 2.0 * W2[6 * r + c];
      }
  }
  if (err) for (int r = 0; r < 12; r++) err[(size_t)b * 12 + r] = P.REC[RIDX(i, REC_D + r, b)];
  if (term) return;
  if (lu) for (int a = 0; a < m; a++) lu[(size_t)b * m + a] = P.REC[RIDX(i, REC_LU + a, b)];
  if (luu)
    for (int a = 0; a < m; a++)
      for (int k = 0; k < m; k++)
        luu[((size_t)b * m + a) * m + k] = 2.0 * C.R[a * m + k] + ((a == k && P.al_lb) ? P.REC[RIDX(i, P.fLUU + a, b)] : 0.0);
  if (f_q || f_xi) {  // f(x, u) was left in knot i of the candidate array by the probe-mode linearisation
    State F = load_state(P, P.cand, i, b);
    if (f_q) pose_to_m16(F.X, f_q + 16 * (size_t)b);
    if (f_xi) {
      double* x = f_xi + (size_t)b * 6;
      x[0] = F.w.x; x[1] = F.w.y; x[2] = F.w.z; x[3] = F.v.x; x[4] = F.v.y; x[5] = F.v.z;
    }
  }
  if (Fu) {
    double* F = Fu + (size_t)b * 12 * m;
    for (int k = 0; k < 12 * m; k++) F[k] = 0;
    for (int r = 0; r < 6; r++)
      for (int k = 0; k < m; k++)
        F[(6 + r) * m + k] = (r < 3) ? (k < 3 ? (C.kind == TOLG_DYN_PENDULUM3D ? P.REC[RIDX(i, REC_BU + 3 * r + k, b)] : C.Bt[3 * r + k]) : 0.0)
                                     : (k >= 3 ? C.Bb[3 * (r - 3) + (k - 3)] : 0.0);
  }
  if (Fx) {
    double* F = Fx + (size_t)b * 144;
    for (int k = 0; k < 144; k++) F[k] = 0;
    for (int r = 0; r < 3; r++)
      for (int c = 0; c < 3; c++) {
        double ri = P.REC[RIDX(i, REC_RI + 3 * c + r, b)], jr = P.REC[RIDX(i, REC_JR + 3 * c + r, b)];
        F[12 * r + c] = ri; F[12 * (r + 3) + c + 3] = ri;
        F[12 * (r + 3) + c] = (C.kind == TOLG_DYN_PENDULUM3D) ? 0.0 : P.REC[RIDX(i, REC_TRI + 3 * c + r, b)];
        F[12 * r + c + 6] = jr; F[12 * (r + 3) + c + 9] = jr;
        F[12 * (r + 3) + c + 6] = P.REC[RIDX(i, REC_QR + 3 * c + r, b)];
      }
    double rte[3];
    for (int a = 0; a < 3; a++) rte[a] = (C.grav != 0.0) ? P.REC[RIDX(i, rec_rte(P.m) + a, b)] : 0.0;
    double a22[36];
    a22_get(P, C, i, b, a22);
    for (int r = 0; r < 6; r++)
      for (int c = 0; c < 6; c++) {
        double lsum = 0;
        for (int a = 0; a < 3; a++) lsum += rte[a] * C.Llin[a][6 * r + c];
        F[12 * (r + 6) + c + 6] = a22[6 * c + r];
        F[12 * (r + 6) + c] = lsum;
      }
  }
}

// export kernels for the unit-parity entry point
__global__ void k_export_lin(Params P, double* __restrict__ Fx, double* __restrict__ d, double* __restrict__ lx,
                             double* __restrict__ lxx11, double* __restrict__ kk, double* __restrict__ K) {
  const Consts& C = *P.c;
  size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= (size_t)(P.N + 1) * P.B) return;
  int b = (int)(t % P.B), i = (int)(t / P.B);
  size_t bi = (size_t)b * (P.N + 1) + i, bn = (size_t)b * P.N + i;
  for (int r = 0; r < 6; r++) {
    if (lxx11) for (int c = 0; c < 6; c++) lxx11[(bi * 6 + r) * 6 + c] = P.REC[RIDX(i, REC_LXX + sym6(r, c), b)];
    if (lx) { lx[bi * 12 + r] = P.REC[RIDX(i, REC_LX + r, b)]; lx[bi * 12 + 6 + r] = P.REC[RIDX(i, REC_LX + 6 + r, b)]; }
  }
  if (i == P.N) return;
  if (Fx) {  // re-assemble the dense 12x12 the reference returns from the compact record
    double* F = Fx + bn * 144;
    for (int k = 0; k < 144; k++) F[k] = 0;
    for (int r = 0; r < 3; r++)
      for (int c = 0; c < 3; c++) {
        double ri = P.REC[RIDX(i, REC_RI + 3 * c + r, b)], jr = P.REC[RIDX(i, REC_JR + 3 * c + r, b)];
        F[12 * r + c] = ri; F[12 * (r + 3) + c + 3] = ri;
        F[12 * (r + 3) + c] = (C.kind == TOLG_DYN_PENDULUM3D) ? 0.0 : P.REC[RIDX(i, REC_TRI + 3 * c + r, b)];
        F[12 * r + c + 6] = jr; F[12 * (r + 3) + c + 9] = jr;
        F[12 * (r + 3) + c + 6] = P.REC[RIDX(i, REC_QR + 3 * c + r, b)];
      }
    double rte[3];
    for (int a = 0; a < 3; a++) rte[a] = (C.grav != 0.0) ? P.REC[RIDX(i, rec_rte(P.m) + a, b)] : 0.0;
    double a22[36];
    a22_get(P, C, i, b, a22);
    for (int r = 0; r < 6; r++)
      for (int c = 0; c < 6; c++) {
        double l = 0;
        for (int a = 0; a < 3; a++) l += rte[a] * C.Llin[a][6 * r + c];
        F[12 * (r + 6) + c + 6] = a22[6 * c + r];
        F[12 * (r + 6) + c] = l;
      }
  }
  for (int r = 0; r < 12; r++)
    if (d) d[bn * 12 + r] = P.REC[RIDX(i, REC_D + r, b)];
  for (int u = 0; u < P.m; u++) {
    if (K) for (int c = 0; c < 12; c++) K[(bn * P.m + u) * 12 + c] = P.GK[GKIDX(i, u, b, c)];
    if (kk) kk[bn * P.m + u] = P.GK[GKIDX(i, u, b, 12)];
  }
}
// number of trajectories still being iterated (integer atomics: order-independent)
// diagnostic kernel of tolg_selftest_series (include/tolg.h): one lane per argument set
__global__ __launch_bounds__(64) void k_selftest_series(int n, const double* __restrict__ args, double* __restrict__ out) {
  const int t = blockIdx.x * 64 + threadIdx.x;
  if (t >= n) return;
  const double* a = args + 8 * (size_t)t;
  const V3 w = v3(a[0], a[1], a[2]), v = v3(a[3], a[4], a[5]);
  const double th2s = a[6];
  const bool shared = a[7] != 0.0;
  const double th2 = dot(w, w);
  const Pose Xc = se3_exp(w, v);  // closed form: the argument of the Log test
  const double y = quat_vec2(Xc.q);
  // the gate lin_knot builds: tracking-error Log and the series at its angle, plus the step rotation
  const SeriesGate gs = series_gate(log_small(y) && coef_small(th2s), log_dom(y) && exp_dom(th2s));
  double* o = out + 24 * (size_t)t;
  V3 lw, lv;
  if (shared) se3_log_fast(Xc, lw, lv, gs);
  else se3_log_fast(Xc, lw, lv);
  const double th2l = dot(lw, lw);  // the Log's angle: what lin_knot feeds the coefficient series with
  const SO3Coef k = shared ? so3_coef_fast(th2l, true, gs) : so3_coef_fast(th2, true);
  o[0] = k.a; o[1] = k.b; o[2] = k.c1; o[3] = k.c2; o[4] = k.c3;
  o[5] = shared ? ljacinv_coef_fast(th2l, gs) : ljacinv_coef_fast(th2);
  // Exp under its own gate (the rollout builds it from exp_small / exp_dom of the same rotation)
  const Pose E = se3_exp_fast(w, v);
  o[6] = E.q.x; o[7] = E.q.y; o[8] = E.q.z; o[9] = E.q.w; o[10] = E.t.x; o[11] = E.t.y; o[12] = E.t.z;
  o[13] = lw.x; o[14] = lw.y; o[15] = lw.z; o[16] = lv.x; o[17] = lv.y; o[18] = lv.z;
  // the step rotation about w's axis with angle^2 th2s, under the shared gate (so3_exp_fast + coefficient a)
  const double sc = (th2 > 0.0) ? sqrt(th2s / th2) : 0.0;
  const V3 ws = sc * w;
  const Q4 q = shared ? so3_exp_fast(ws, gs) : so3_exp_fast(ws);
  o[19] = q.x; o[20] = q.y; o[21] = q.z; o[22] = q.w;
  o[23] = (shared ? so3_coef_fast(dot(ws, ws), true, gs) : so3_coef_fast(dot(ws, ws), true)).a;
}

__global__ void k_active_count(Params P, int* __restrict__ out) {
  int b = blockIdx.x * blockDim.x + threadIdx.x;
  const bool a = b < P.B && P.active[b] != 0;
  const unsigned long long m = __ballot(a);
  if ((threadIdx.x & 63) == 0 && m) atomicAdd(out, __popcll(m));
}
__global__ void k_export_scalars(Params P, double* J, double* dn, double* grad, double* mu_delta, int* iters,
                                 int* status, int* conv) {
  int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= P.B) return;
  if (J) J[b] = P.Jc[b];
  if (dn) dn[b] = P.dn[b];
  if (grad) grad[b] = P.grad[b];
  if (mu_delta) { mu_delta[2 * b] = P.mu[b]; mu_delta[2 * b + 1] = P.delta[b]; }
  if (iters) iters[b] = P.iters[b];
  if (status) status[b] = P.status[b];
  if (conv) conv[b] = P.conv[b];
}

}  // namespace tolg

// ================================================================================================
// host side: handle, workspace carving, C ABI
// ================================================================================================
using namespace tolg;

struct tolg_handle_s {
  tolg_problem prob;
  Consts hc;
  int max_batch, Bp_max;
  char* ws;
  size_t ws_bytes;
  Params P;  // pointers carved for Bp_max; per-solve Bp may be smaller (arrays are re-strided)
  Params run;         // parameters of the solve in flight (tolg_solve_begin .. tolg_solve_end)
  tolg_options run_opt;
  int run_it;         // iterations issued so far
  bool running;
  int lds_per_block;  // hipDeviceAttributeMaxSharedMemoryPerBlock of the current device (160 KB on MI355X)
  int rec_closed = 0; // the knot records were last written by the fused rollout (no defect field, see k_backward)
  const double *al_lb, *al_ub, *al_lambda, *al_imu;  // augmented-Lagrangian terms (null = off)
  // early exit of a sliced solve (tolg_solve_iterate_until): two device counters, their pinned host copies, two events
  int *d_cnt = nullptr, *h_cnt = nullptr;
  hipEvent_t cnt_ev[2] = {nullptr, nullptr};
  // merit search: the linear alpha = 1 rollout (k_expected_change) runs on a side stream beside the first line-search stage
  hipStream_t side = nullptr;
  hipEvent_t side_ev[2] = {nullptr, nullptr};
  // timing
  bool timing;
  std::vector<hipEvent_t> ev;  // pairs
  std::vector<int> ev_kind;    // 0 backward, 1 rollout, 2 linearize
  size_t ev_used;
};

static size_t align_up(size_t x, size_t a) { return (x + a - 1) / a * a; }

static int host_inv6(const double A[36], double Ai[36]) {
  double Mx[6][12];
  for (int i = 0; i < 6; i++)
    for (int j = 0; j < 6; j++) { Mx[i][j] = A[6 * i + j]; Mx[i][j + 6] = (i == j); }
  for (int c = 0; c < 6; c++) {
    int p = c;
    for (int r = c + 1; r < 6; r++) if (fabs(Mx[r][c]) > fabs(Mx[p][c])) p = r;
    if (Mx[p][c] == 0) return -1;
    if (p != c) for (int j = 0; j < 12; j++) { double t = Mx[c][j]; Mx[c][j] = Mx[p][j]; Mx[p][j] = t; }
    double d = Mx[c][c];
    for (int j = 0; j < 12; j++) Mx[c][j] /= d;
    for (int r = 0; r < 6; r++) if (r != c) {
      double f = Mx[r][c];
      if (f != 0) for (int j = 0; j < 12; j++) Mx[r][j] -= f * Mx[c][j];
    }
  }
  for (int i = 0; i < 6; i++) for (int j = 0; j < 6; j++) Ai[6 * i + j] = Mx[i][j + 6];
  return 0;
}

struct Carve {
  char* base; size_t off;
  template <typename T> T* take(size_t n) {
    off = align_up(off, 256);
    T* p = base ? reinterpret_cast<T*>(base + off) : nullptr;
    off += n * sizeof(T);
    return p;
  }
};

static size_t carve_all(const tolg_problem* pr, int Bp, char* base, Params* P, Consts** dc) {
  Carve c{base, 0};
  size_t N = (size_t)pr->N, m = (size_t)pr->m, B = (size_t)Bp;
  Consts* cc = c.take<Consts>(1);
  double* ref = c.take<double>((N + 1) * 13);
  double* cur = c.take<double>(13 * (N + 1) * B);
  double* cur_u = c.take<double>(m * N * B);
  double* cand = c.take<double>(13 * (N + 1) * B);
  double* cand_u = c.take<double>(m * N * B);
  double* REC = c.take<double>((N + 1) * (size_t)REC_FMAX * B);
  double* SC = c.take<double>((N + 1) * B);
  double* SD = c.take<double>(N * B);
  double* GK = c.take<double>(N * m * B * 13);
  double* mu = c.take<double>(B);
  double* delta = c.take<double>(B);
  double* Jc = c.take<double>(B);
  double* dn = c.take<double>(B);
  double* grad = c.take<double>(B);
  int* active = c.take<int>(B);
  int* iters = c.take<int>(B);
  int* status = c.take<int>(B);
  int* conv = c.take<int>(B);
  double* slot_x = c.take<double>((size_t)NSLOT * 13 * (N + 1) * B);
  double* slot_u = c.take<double>((size_t)NSLOT * m * N * B);
  double* Jtrial = c.take<double>(20 * B);
  double* dtrial = c.take<double>(20 * B);
  double* ecc = c.take<double>(2 * B);
  double* dweight = c.take<double>(2 * B);
  double* ls_alpha = c.take<double>(B);
  int* ls_accept = c.take<int>(B);
  int* ls_slot = c.take<int>(B);
  int* k2_redo = c.take<int>(B / 4 + 1);
  int* k2_hint = c.take<int>(B / 4 + 1);
  int* ec_redo = c.take<int>(B);
  int* ls_list = c.take<int>(2 * B);
  int* ls_count = c.take<int>(64);
  int* ls_pos = c.take<int>(2 * B);
  double* LSC = c.take<double>((size_t)NSLOT * (N + 1) * B);
  double* LSD = c.take<double>((size_t)NSLOT * N * B);
  double* ED = c.take<double>((N + 1) * 32 * B);
  if (P) {
    P->ED = ED; P->affine = 0; P->pad3 = 0;
    P->k2_redo = k2_redo; P->k2_hint = k2_hint;
    P->ec_redo = ec_redo;
    P->ls_list = ls_list; P->ls_count = ls_count; P->ls_pos = ls_pos; P->LSC = LSC; P->LSD = LSD;
    P->slot_x = slot_x; P->slot_u = slot_u; P->Jtrial = Jtrial; P->dtrial = dtrial; P->ecc = ecc;
    P->dweight = dweight; P->ls_alpha = ls_alpha; P->ls_accept = ls_accept; P->ls_slot = ls_slot;
    P->c = cc; P->ref = ref; P->cur = cur; P->cur_u = cur_u; P->cand = cand; P->cand_u = cand_u;
    P->REC = REC; P->SC = SC; P->SD = SD; P->GK = GK; P->mu = mu; P->delta = delta; P->Jc = Jc;
    P->dn = dn; P->grad = grad; P->active = active; P->iters = iters; P->status = status; P->conv = conv;
  }
  if (dc) *dc = cc;
  return align_up(c.off, 256);
}

static int check_problem(const tolg_problem* p) {
  if (!p) return TOLG_E_ARG;
  if (p->N < 1 || !(p->dt > 0)) return TOLG_E_ARG;
  if (p->kind == TOLG_DYN_DRONE) { if (p->m != 4) return TOLG_E_ARG; }
  else if (p->kind == TOLG_DYN_SE3 || p->kind == TOLG_DYN_RIGIDBODY || p->kind == TOLG_DYN_SO3 || p->kind == TOLG_DYN_PENDULUM3D) { if (p->m != 6) return TOLG_E_ARG; }
  else return TOLG_E_ARG;
  return 0;
}

extern "C" size_t tolg_workspace_bytes(const tolg_problem* prob, int32_t max_batch) {
  if (check_problem(prob) || max_batch < 1) return 0;
  int Bp = (max_batch + 3) / 4 * 4;
  return carve_all(prob, Bp, nullptr, nullptr, nullptr);
}

extern "C" const char* tolg_version(void) { return "tolg-hip 0.3 (gfx950)"; }

// LDS is not cleared between launches: a kernel that reads LDS it has not written usually meets what the previous
// launch of the same kernel left at the same offsets -- the right values -- and passes its tests (k_expected_change did,
// for one commit: DESIGN.md section 4).  k_poison_lds fills the LDS of every CU with NaNs; TOLG_POISON_LDS=1 in the
// environment (read once) puts one behind EVERY launch of the library, on the null stream (a debugging mode: it
// serialises the streams), so that such a read shows anywhere in the test suite; the unit-parity entry point
// tolg_expected_change always runs one first.
__global__ __launch_bounds__(256) void k_poison_lds() {
  __shared__ double junk[8192];
  for (int k = threadIdx.x; k < 8192; k += 256) junk[k] = __builtin_nan("");
  __syncthreads();
  if (junk[(threadIdx.x * 33) & 8191] == 0.0) __builtin_trap();  // keeps the stores alive; never true
}
static bool getenv_flag(const char* name) {  // (read at every call: a handful of launches per iteration; used by A/B tests only)
  const char* e = getenv(name);
  return e && e[0] == '1';
}
static bool poison_lds_mode() {
  static const bool v = [] { const char* e = getenv("TOLG_POISON_LDS"); return e && e[0] == '1'; }();
  return v;
}
#define LAUNCH_CHECK()                                         \
  do {                                                         \
    hipError_t e_ = hipGetLastError();                         \
    if (e_ != hipSuccess) {                                    \
      fprintf(stderr, "tolg: launch failed at %s:%d: %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); \
      return TOLG_E_LAUNCH;                                    \
    }                                                          \
    if (poison_lds_mode()) hipLaunchKernelGGL(k_poison_lds, dim3(2048), dim3(256), 0, 0); \
  } while (0)

extern "C" int tolg_create(const tolg_problem* prob, const double* d_q_ref, const double* d_xi_ref,
                           int32_t max_batch, void* d_workspace, size_t workspace_bytes, void* stream,
                           tolg_handle_t* out) {
  if (check_problem(prob) || !d_q_ref || !d_xi_ref || !d_workspace || !out || max_batch < 1) return TOLG_E_ARG;
  if (workspace_bytes < tolg_workspace_bytes(prob, max_batch)) return TOLG_E_WORKSPACE;
  if ((reinterpret_cast<uintptr_t>(d_workspace) & 255) != 0) return TOLG_E_ARG;
  // one knot of records must stay below the out-of-range offset K2 uses for structurally-zero loads (1 GiB)
  if ((size_t)((max_batch + 3) / 4 * 4) * REC_FMAX * 8 >= 0x40000000ull) return TOLG_E_ARG;
  tolg_handle_s* h = new (std::nothrow) tolg_handle_s();
  if (!h) return TOLG_E_ARG;
  h->prob = *prob;
  h->max_batch = max_batch;
  h->Bp_max = (max_batch + 3) / 4 * 4;
  h->ws = static_cast<char*>(d_workspace);
  h->ws_bytes = workspace_bytes;
  h->timing = false;
  h->ev_used = 0;
  h->running = false;
  h->run_it = 0;
  h->al_lb = h->al_ub = h->al_lambda = h->al_imu = nullptr;
  {
    int dev = 0, lds = 0;
    if (hipGetDevice(&dev) != hipSuccess ||
        hipDeviceGetAttribute(&lds, hipDeviceAttributeMaxSharedMemoryPerBlock, dev) != hipSuccess) lds = 65536;
    h->lds_per_block = lds;
  }
  Consts& c = h->hc;
  memset(&c, 0, sizeof c);
  c.kind = prob->kind; c.m = prob->m; c.N = prob->N; c.diagJ = 0; c.dt = prob->dt;
  memcpy(c.J, prob->J, sizeof c.J);
  if (host_inv6(prob->J, c.Jinv)) { tolg_destroy(h); return TOLG_E_SINGULAR; }
  for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) c.Ib[3 * i + j] = prob->J[6 * i + j];
  c.mass = prob->J[6 * 4 + 4];                      // traopt_dynamics.py:663
  c.grav = (prob->kind == TOLG_DYN_SE3 || prob->kind == TOLG_DYN_SO3) ? 0.0 : 9.8;  // traopt_dynamics.py:1245, :466
  c.pend_k = (prob->kind == TOLG_DYN_PENDULUM3D) ? prob->pend_mass * prob->pend_length / 2 : 0.0;
  for (int i = 0; i < 6; i++)
    for (int j = 0; j < 6; j++) {
      c.W1[6 * i + j] = prob->Q[12 * i + j];
      c.W2[6 * i + j] = prob->Q[12 * (i + 6) + j + 6];
      c.P1[6 * i + j] = prob->P[12 * i + j];
      c.P2[6 * i + j] = prob->P[12 * (i + 6) + j + 6];
    }
  int m = prob->m;
  for (int i = 0; i < m * m; i++) c.R[i] = prob->R[i];
  // J must be blkdiag(Ib, Jv) -- the structure the reference itself assumes (traopt_dynamics.py:640-665:
  // "J: Inertia matrix, diag(I_b, m * I_3)"; G in f_x is built from Ib and m only)
  for (int i = 0; i < 3; i++)
    for (int j = 0; j < 3; j++) {
      if (prob->J[6 * i + j + 3] != 0.0 || prob->J[6 * (i + 3) + j] != 0.0) { tolg_destroy(h); return TOLG_E_ARG; }
      c.Jv[3 * i + j] = prob->J[6 * (i + 3) + j + 3];
      c.Ibinv[3 * i + j] = c.Jinv[6 * i + j];
      c.Jvinv[3 * i + j] = c.Jinv[6 * (i + 3) + j + 3];
    }
  c.diagJ = 1;
  for (int i = 0; i < 3; i++)
    for (int j = 0; j < 3; j++)
      if (i != j && (c.Ib[3 * i + j] != 0.0 || c.Jv[3 * i + j] != 0.0)) c.diagJ = 0;
  // F_u = Bt dt (traopt_dynamics.py:668-670, :850; Pu of the drone :1250-1254)
  for (int i = 0; i < 9; i++) { c.Bt[i] = c.Ibinv[i] * prob->dt; c.Bb[i] = 0; }
  for (int i = 0; i < 3; i++) {
    if (prob->kind == TOLG_DYN_DRONE) c.Bb[3 * i] = c.Jvinv[3 * i + 2] * prob->dt;
    else for (int j = 0; j < 3; j++) c.Bb[3 * i + j] = c.Jvinv[3 * i + j] * prob->dt;
  }
  // gravity Jacobian L = J^-1 [[0,0],[skew(R^T e3), 0]] dt as a linear map of R^T e3
  // (traopt_dynamics.py:1445-1458; no m*g)
  for (int a = 0; a < 3; a++) {
    double e3[3] = {0, 0, 0}, S[9];
    e3[a] = 1.0;
    S[0] = 0; S[1] = -e3[2]; S[2] = e3[1]; S[3] = e3[2]; S[4] = 0; S[5] = -e3[0]; S[6] = -e3[1]; S[7] = e3[0]; S[8] = 0;
    for (int i = 0; i < 6; i++)
      for (int j = 0; j < 6; j++) {
        double sacc = 0;
        if (j < 3 && c.grav != 0.0 && prob->kind != TOLG_DYN_PENDULUM3D)
          for (int k = 0; k < 3; k++) sacc += c.Jinv[6 * i + 3 + k] * S[3 * k + j];
        if (j < 3 && i < 3 && prob->kind == TOLG_DYN_PENDULUM3D) {
          // Pendulum3dDyanmics.f_x (traopt_dynamics.py:574-588): L = J^-1 skew(m rho) skew(w), w = R^T(g e + u),
          // skew(m rho) = k [[0,1,0],[-1,0,0],[0,0,0]]
          const double SS[9] = {c.pend_k * S[3 + 0], c.pend_k * S[3 + 1], c.pend_k * S[3 + 2],
                                -c.pend_k * S[0], -c.pend_k * S[1], -c.pend_k * S[2], 0, 0, 0};
          for (int k = 0; k < 3; k++) sacc += c.Ibinv[3 * i + k] * SS[3 * k + j];
        }
        c.Llin[a][6 * i + j] = sacc * prob->dt;
      }
  }
  hipStream_t st = static_cast<hipStream_t>(stream);
  Consts* dc = nullptr;
  memset(&h->P, 0, sizeof h->P);
  carve_all(prob, h->Bp_max, h->ws, &h->P, &dc);
  h->P.N = prob->N; h->P.m = prob->m;
  if (hipMemcpyAsync(dc, &h->hc, sizeof(Consts), hipMemcpyHostToDevice, st) != hipSuccess) { tolg_destroy(h); return TOLG_E_LAUNCH; }
  int N = prob->N;
  hipLaunchKernelGGL(k_pack_ref, dim3((N + 1 + 63) / 64), dim3(64), 0, st, N, d_q_ref, d_xi_ref,
                     const_cast<double*>(h->P.ref));
  LAUNCH_CHECK();
  *out = h;
  return 0;
}

extern "C" void tolg_destroy(tolg_handle_t h) {
  if (h) {
    if (h->d_cnt) (void)hipFree(h->d_cnt);
    if (h->h_cnt) (void)hipHostFree(h->h_cnt);
    for (int k = 0; k < 2; k++) if (h->cnt_ev[k]) (void)hipEventDestroy(h->cnt_ev[k]);
    for (int k = 0; k < 2; k++) if (h->side_ev[k]) (void)hipEventDestroy(h->side_ev[k]);
    if (h->side) (void)hipStreamDestroy(h->side);
  }
  if (!h) return;
  for (auto e : h->ev) (void)hipEventDestroy(e);
  delete h;
}

extern "C" void tolg_enable_timing(tolg_handle_t h, int32_t on) {
  if (!h) return;
  h->timing = on != 0;
  if (h->timing && h->ev.empty()) {
    h->ev.resize(2 * 1024);
    for (auto& e : h->ev) (void)hipEventCreate(&e);
    h->ev_kind.resize(1024);
  }
  h->ev_used = 0;
}

namespace {
struct Timed {
  // Two ways to time what a scope launches.  The general one records an event in front of and behind whatever the scope
  // puts on the stream -- two marker packets, which cost the stream ~7 us per pair.  The one for a scope that is ONE kernel
  // (`ext`: the backward sweep's third form, the fused launch) hands the event pair to the dispatch itself
  // (hipExtLaunchKernelGGL: start / stop timestamps of the kernel's own completion signal, no extra packet).
  tolg_handle_s* h; hipStream_t st; int kind; bool on, ext;
  bool count = true;  // false: its time adds to its kind, but it is not one more launch of the sweep (the redo behind the fast sweep)
  Timed(tolg_handle_s* h_, hipStream_t s, int k, bool ext_ = false) : h(h_), st(s), kind(k), ext(ext_) {
    on = h->timing;
    if (on && h->ev_used == h->ev_kind.size()) {  // grow the event pool: a long solve must not silently stop being timed
      const size_t n = h->ev_kind.size() + 1024;
      h->ev.resize(2 * n);
      for (size_t i = 2 * h->ev_kind.size(); i < 2 * n; i++)
        if (hipEventCreate(&h->ev[i]) != hipSuccess) { h->ev.resize(2 * h->ev_kind.size()); on = false; break; }
      if (on) h->ev_kind.resize(n);
    }
    if (on && !ext) (void)hipEventRecord(h->ev[2 * h->ev_used], st);
  }
  template <typename F, typename... Args>
  void launch(F kernel, dim3 grid, dim3 blk, Args... args) {  // (ext scopes: exactly one call)
    if (on && ext) hipExtLaunchKernelGGL(kernel, grid, blk, 0, st, h->ev[2 * h->ev_used], h->ev[2 * h->ev_used + 1], 0, args...);
    else hipLaunchKernelGGL(kernel, grid, blk, 0, st, args...);
  }
  ~Timed() {
    if (on) {
      if (!ext) (void)hipEventRecord(h->ev[2 * h->ev_used + 1], st);
      h->ev_kind[h->ev_used] = kind | (count ? 0 : 8); h->ev_used++;
    }
  }
};
}  // namespace

extern "C" int tolg_kernel_time(tolg_handle_t h, int32_t reset, double* ms_backward, double* ms_rollout,
                                double* ms_linearize, int64_t* launches) {
  if (!h) return TOLG_E_ARG;
  double acc[3] = {0, 0, 0};
  int64_t nb = 0;
  for (size_t i = 0; i < h->ev_used; i++) {
    (void)hipEventSynchronize(h->ev[2 * i + 1]);
    float ms = 0;
    if (hipEventElapsedTime(&ms, h->ev[2 * i], h->ev[2 * i + 1]) == hipSuccess) acc[h->ev_kind[i] & 7] += ms;
    if (h->ev_kind[i] == 0) nb++;
  }
  if (ms_backward) *ms_backward = acc[0];
  if (ms_rollout) *ms_rollout = acc[1];
  if (ms_linearize) *ms_linearize = acc[2];
  if (launches) *launches = nb;
  if (reset) h->ev_used = 0;
  return 0;
}

static Params params_for(tolg_handle_s* h, int B) {
  Params P = h->P;
  P.B = B;
  P.Bp = (B + 3) / 4 * 4;
  P.J_hist = P.grad_hist = P.defect_hist = P.alpha_hist = P.mu_hist = nullptr;
  P.max_iter = 0; P.tol_grad = 0; P.tol_defect = 0; P.max_reg = 1e10;
  P.al_lb = h->al_lb; P.al_ub = h->al_ub; P.al_lambda = h->al_lambda; P.al_imu = h->al_imu;
  const bool grav = h->hc.grav != 0.0;
  // the velocity block of F_x is stored only for the models the third form of the backward sweep does not cover
  const bool a22 = !(h->hc.diagJ != 0 && h->prob.kind != TOLG_DYN_PENDULUM3D);
  P.recF = rec_fields(P.m, grav, P.al_lb != nullptr, a22);
  P.fLUU = REC_LU + P.m + (grav ? 4 : 0);
  P.fA22 = a22 ? P.fLUU + (P.al_lb != nullptr ? P.m : 0) : -1;
  P.pad2 = 0;
  P.affine = 0;
  return P;
}

template <int M>
static int run_linearize(tolg_handle_s* h, const Params& P, hipStream_t st, const double* src, const double* src_u,
                         double* dst, double* dst_u, int ms, int i0 = 0, int ni = -1, int ls_list = -1, int ls_nslots = 0) {
  if (ni < 0) ni = P.N + 1;
  size_t n = (size_t)ni * P.Bp;
  h->rec_closed = 0;  // K1 writes the defect field
  Timed t(h, st, 2);
  hipLaunchKernelGGL(k_linearize<M>, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, P, src, src_u, dst, dst_u, ms,
                     i0, ni, ls_list, ls_nslots);
  LAUNCH_CHECK();
  return 0;
}
template <int M>
static int run_backward(tolg_handle_s* h, const Params& P, hipStream_t st, int it, int ms) {
  const bool dj = h->hc.diagJ != 0;
  ms = (ms ? 1 : 0) | (h->rec_closed ? 2 : 0);
  const dim3 grid(P.Bp / 4), blk(64);
  // diagonal inertia blocks and a constant input matrix (every reference script except the pendulum): the third form
  // of the sweep (tolg_backward3.h).  Dense inertia and the pendulum keep k_backward.
  if (dj && h->prob.kind != TOLG_DYN_PENDULUM3D) {
    const bool al = P.al_lb != nullptr, grav = h->hc.grav != 0.0;
    // From the second sweep of a solve on: the fast kernel (no general path compiled in: 254 registers instead of 395), then
    // the full kernel for the groups it handed back (flag bit 2) -- usually none.  The first sweep starts from mu = 1
    // (traopt_controller.py:2394): straight to the full kernel; so do the one-sweep entry points (it < 0).
    const bool fast = it > 0 && !getenv_flag("TOLG_K2_FULL_ONLY");
    for (int pass = fast ? 0 : 1; pass < 2; pass++) {
      Timed t(h, st, 0, true);  // each pass is one launch: timed through its dispatch
      t.count = (pass == 0) || !fast;
      const int fl = ms | ((pass == 1 && fast) ? 4 : 0);
      if (pass == 0) {
        if (!grav) { if (al) t.launch(k_backward3<M, false, true, true>, grid, blk, P, it, fl); else t.launch(k_backward3<M, false, false, true>, grid, blk, P, it, fl); }
        else { if (al) t.launch(k_backward3<M, true, true, true>, grid, blk, P, it, fl); else t.launch(k_backward3<M, true, false, true>, grid, blk, P, it, fl); }
      } else {
        if (!grav) { if (al) t.launch(k_backward3<M, false, true, false>, grid, blk, P, it, fl); else t.launch(k_backward3<M, false, false, false>, grid, blk, P, it, fl); }
        else { if (al) t.launch(k_backward3<M, true, true, false>, grid, blk, P, it, fl); else t.launch(k_backward3<M, true, false, false>, grid, blk, P, it, fl); }
      }
      LAUNCH_CHECK();
    }
    return 0;
  }
  // dense inertia blocks, the pendulum: the general sweep, in the same two passes
  const bool fast = it > 0 && !getenv_flag("TOLG_K2_FULL_ONLY");
  for (int pass = fast ? 0 : 1; pass < 2; pass++) {
    Timed t(h, st, 0, true);
    t.count = (pass == 0) || !fast;
    const int fl = ms | ((pass == 1 && fast) ? 4 : 0);
    if (M == 6 && h->prob.kind == TOLG_DYN_PENDULUM3D) {
      if (pass == 0) { if (dj) t.launch(k_backward<6, true, true, true, true>, grid, blk, P, it, fl); else t.launch(k_backward<6, true, true, false, true>, grid, blk, P, it, fl); }
      else { if (dj) t.launch(k_backward<6, true, true, true, false>, grid, blk, P, it, fl); else t.launch(k_backward<6, true, true, false, false>, grid, blk, P, it, fl); }
    } else if (M == 6 && h->hc.grav == 0.0) {
      if (pass == 0) t.launch(k_backward<6, false, false, false, true>, grid, blk, P, it, fl);
      else t.launch(k_backward<6, false, false, false, false>, grid, blk, P, it, fl);
    } else {
      if (pass == 0) t.launch(k_backward<M, false, true, false, true>, grid, blk, P, it, fl);
      else t.launch(k_backward<M, false, true, false, false>, grid, blk, P, it, fl);
    }
    LAUNCH_CHECK();
  }
  return 0;
}
template <int M>
static int run_rollout_ms(tolg_handle_s* h, const Params& P, hipStream_t st, double alpha, int linear, int ms = 1,
                          int i0 = 0, int i1 = -1) {
  if (i1 < 0) i1 = P.N;
  Timed t(h, st, 1);
  // four lanes per trajectory; 64-thread groups so that the 256 waves of a 4096-batch land on 256
  // different CUs: the sweep streams ~16 KB per wave-step from HBM and one CU sustains ~10 B/cycle of
  // misses (4 waves on one CU: 0.96 ms instead of 0.47, SQ_VMEM_TA_*_FIFO_FULL x7)
  dim3 grid((P.Bp * 4 + 63) / 64), blk(64);
  if (linear) hipLaunchKernelGGL((k_rollout<M, true, false>), grid, blk, 0, st, P, alpha, i0, i1);
  else if (M == 6 && h->prob.kind == TOLG_DYN_PENDULUM3D) {
    if (alpha == 1.0 || !ms) hipLaunchKernelGGL((k_rollout<6, false, true, 1>), grid, blk, 0, st, P, alpha, i0, i1);
    else hipLaunchKernelGGL((k_rollout<6, false, false, 1>), grid, blk, 0, st, P, alpha, i0, i1);
  }
  else if (alpha == 1.0 || !ms) hipLaunchKernelGGL((k_rollout<M, false, true>), grid, blk, 0, st, P, alpha, i0, i1);
  else hipLaunchKernelGGL((k_rollout<M, false, false>), grid, blk, 0, st, P, alpha, i0, i1);
  LAUNCH_CHECK();
  return 0;
}
// the merit search's preparation in its ring form (tolg_expected_change.h): gravity / dense-inertia instantiations
template <int M>
static void launch_ec_ring(const tolg_handle_s* h, const Params& P, hipStream_t st) {
  const dim3 gr(P.Bp / 4), blk(64);
  const bool grav = h->hc.grav != 0.0, dense = P.fA22 >= 0;
  if (M == 6 && h->prob.kind == TOLG_DYN_PENDULUM3D) {
    hipLaunchKernelGGL((k_expected_change_ring<6, true, false, true, true>), gr, blk, 0, st, P);
  } else if (dense) {
    if (grav) hipLaunchKernelGGL((k_expected_change_ring<M, true, false, true>), gr, blk, 0, st, P);
    else hipLaunchKernelGGL((k_expected_change_ring<M, false, false, true>), gr, blk, 0, st, P);
  } else {
    if (grav) hipLaunchKernelGGL((k_expected_change_ring<M, true>), gr, blk, 0, st, P);
    else hipLaunchKernelGGL((k_expected_change_ring<M, false>), gr, blk, 0, st, P);
  }
}
// rollout = 'linear', models of the third backward form: the alpha = 1 linear rollout as an affine recursion, e_i and du_i
// left in P.ED (tolg_expected_change.h, STORE), plus -- what the kernel is named after -- the expected cost change and the
// defect weight of the merit search.  The trajectories it hands back get their expected change from the statement form.
template <int M>
static int run_affine_dev(tolg_handle_s* h, const Params& P, hipStream_t st, bool merit) {
  Timed t(h, st, 1);
  const dim3 gr(P.Bp / 4), blk(64);
  const bool pend = M == 6 && h->prob.kind == TOLG_DYN_PENDULUM3D;
  if (pend) {  // (the knot's input-matrix block from the record run as well)
    hipLaunchKernelGGL((k_expected_change_ring<6, true, true, true, true>), gr, blk, 0, st, P);
  } else if (P.fA22 >= 0) {  // dense inertia blocks: the velocity block from the record run
    if (h->hc.grav != 0.0) hipLaunchKernelGGL((k_expected_change_ring<M, true, true, true>), gr, blk, 0, st, P);
    else hipLaunchKernelGGL((k_expected_change_ring<M, false, true, true>), gr, blk, 0, st, P);
  } else {
    if (h->hc.grav != 0.0) hipLaunchKernelGGL((k_expected_change_ring<M, true, true>), gr, blk, 0, st, P);
    else hipLaunchKernelGGL((k_expected_change_ring<M, false, true>), gr, blk, 0, st, P);
  }
  LAUNCH_CHECK();
  if (merit) {
    if (pend) hipLaunchKernelGGL((k_expected_change<6, 1, true>), dim3((P.Bp * 4 + 63) / 64), dim3(64), 0, st, P);
    else hipLaunchKernelGGL((k_expected_change<M, 0, true>), dim3((P.Bp * 4 + 63) / 64), dim3(64), 0, st, P);
    LAUNCH_CHECK();
  }
  return 0;
}
// One stage of the speculative line search: alphas a0 .. a0 + n - 1 of every still-undecided trajectory at once.
// stage = 0, 1, 2 ...: stage 0 takes the undecided trajectories from the flags (all active ones), stage s > 0 from the
// list select s - 1 compacted (lists alternate: select s fills list s & 1 while this stage's kernels read the other).
// A one-alpha stage writes its candidate in place (no slot, no copy).
template <int M, bool MS>
static int run_ls_stage(tolg_handle_s* h, const Params& P, hipStream_t st, int stage, int a0, int n, int linear,
                        hipEvent_t before_select = nullptr, bool last = false) {
  const int direct = n == 1;
  if (n > NSLOT) return TOLG_E_ARG;
  const bool pend = M == 6 && h->prob.kind == TOLG_DYN_PENDULUM3D;
  const int list_in = stage == 0 ? -1 : (stage - 1) & 1, list_out = stage & 1;
  // the first try, alpha = 1: x^+ = f(x^, u^) for single shooting (:2073-2080) and for the merit search alike (the
  // factors of :2713-2716 are the identity: note at the record layout) -- K3 itself, written straight into the candidate
  // arrays (every active trajectory is undecided at this point)
  const bool k3 = direct && a0 == 0 && !linear;
  // the rollouts of a stage in two wavefronts per sixteen quads (k_rollout_ls2): the nonlinear rollouts of every model (PK = 1: the
  // pendulum); TOLG_LS_ONEWAVE=1 keeps the one-wave forms (K3 for the first try, k_rollout_ls) for comparisons
  const bool two = !linear && !getenv_flag("TOLG_LS_ONEWAVE");
  if (k3 && !two) {
    int rc = run_rollout_ms<M>(h, P, st, 1.0, 0, MS ? 1 : 0);
    if (rc) return rc;
  }
  {
    Timed t(h, st, 1);
    if (!direct && (size_t)P.Bp * n > (size_t)LS_QUAD_MAX) {  // the thread form of a wide stage: runs when the list is long
                                                               // (ls_quad_form), leaves at once otherwise; not launched where no list of this batch can be that long
      dim3 grid((P.Bp + 63) / 64, n), blk(64);
      if (linear) hipLaunchKernelGGL((k_rollout_eval_t<M, MS, true>), grid, blk, 0, st, P, a0, n, list_in);
      else hipLaunchKernelGGL((k_rollout_eval_t<M, MS, false>), grid, blk, 0, st, P, a0, n, list_in);
      LAUNCH_CHECK();
    }
    {
      dim3 grid((P.Bp * 4 + 63) / 64, n), blk(64);  // four lanes per (trajectory, alpha)
      if (two) {
        // twist waves per workgroup (k_rollout_ls2): the merit search's stages are a few hundred quads -- one chain's length, and
        // a pose wave of its own per sixteen quads keeps that chain shortest (NT = 1; with NT = 3 merit 596 -> 526 it/s); single
        // shooting's twelve step sizes fill the chip several times over and gain from fewer waves per quad (NT = 3: SS 517 -> 542)
        if (pend) {  // (m = 6 only; one twist wave per pose wave in every stage)
          const dim3 g2((P.Bp + 15) / 16, n), b2(128);
          if (MS && !k3) hipLaunchKernelGGL((k_rollout_ls2<6, true, 1, 1>), g2, b2, 0, st, P, a0, n, direct, list_in);
          else hipLaunchKernelGGL((k_rollout_ls2<6, false, 1, 1>), g2, b2, 0, st, P, a0, n, direct, list_in);
        } else if (MS) {
          const dim3 g2((P.Bp + 15) / 16, n), b2(128);
          if (!k3) hipLaunchKernelGGL((k_rollout_ls2<M, true, 1>), g2, b2, 0, st, P, a0, n, direct, list_in);
          else hipLaunchKernelGGL((k_rollout_ls2<M, false, 1>), g2, b2, 0, st, P, a0, n, direct, list_in);
        } else if (direct) {  // (its first try is one chain again: NT = 1, 542 -> 547)
          hipLaunchKernelGGL((k_rollout_ls2<M, false, 1>), dim3((P.Bp + 15) / 16, n), dim3(128), 0, st, P, a0, n, direct, list_in);
        } else {
          hipLaunchKernelGGL((k_rollout_ls2<M, false, 3>), dim3((P.Bp + 47) / 48, n), dim3(256), 0, st, P, a0, n, direct, list_in);
        }
      } else if (k3) {
      } else if (pend) {
        if (linear) hipLaunchKernelGGL((k_rollout_ls<6, MS, true, 1>), grid, blk, 0, st, P, a0, n, direct, list_in);
        else hipLaunchKernelGGL((k_rollout_ls<6, MS, false, 1>), grid, blk, 0, st, P, a0, n, direct, list_in);
      } else {
        if (linear) hipLaunchKernelGGL((k_rollout_ls<M, MS, true, 0>), grid, blk, 0, st, P, a0, n, direct, list_in);
        else hipLaunchKernelGGL((k_rollout_ls<M, MS, false, 0>), grid, blk, 0, st, P, a0, n, direct, list_in);
      }
      LAUNCH_CHECK();
    }
    const size_t nn = (size_t)(P.N + 1) * P.Bp;
    const unsigned evb = (unsigned)(P.N + 1) * (unsigned)((P.Bp + 255) / 256);  // (knot, 256 candidates) blocks: k_ls_eval
    hipLaunchKernelGGL((k_ls_eval<M, MS>), dim3(evb, n), dim3(256), 0, st, P, n, direct, list_in);
    LAUNCH_CHECK();
    hipLaunchKernelGGL((k_ls_sum<MS>), dim3((unsigned)(((size_t)P.Bp * n + 63) / 64)), dim3(64), 0, st, P, a0, n, list_in, 0);
    LAUNCH_CHECK();
    if (P.affine) {  // the candidates that come from the affine recursion: built where they are evaluated
      hipLaunchKernelGGL((k_ls_eval_affine<M, MS>), dim3(evb, n), dim3(256), 0, st, P, a0, n, list_in);
      LAUNCH_CHECK();
      hipLaunchKernelGGL((k_ls_sum<MS>), dim3((unsigned)(((size_t)P.Bp * n + 63) / 64)), dim3(64), 0, st, P, a0, n, list_in, 1);
      LAUNCH_CHECK();
    }
  }
  if (before_select && hipStreamWaitEvent(st, before_select, 0) != hipSuccess) return TOLG_E_LAUNCH;
  hipLaunchKernelGGL((k_ls_select<MS>), dim3((P.Bp + 63) / 64), dim3(64), 0, st, P, a0, n, list_out);
  LAUNCH_CHECK();
  // the last stage of a search leaves its accepted candidates in their slots: k_linearize reads them there (no copy).  Not in a
  // solve on the affine path, whose candidates exist only where k_affine_commit writes them
  const bool keep = last && !direct && !P.affine;
  if (!direct && !keep) {
    size_t nn = (size_t)(P.N + 1) * P.Bp;
    hipLaunchKernelGGL(k_ls_copy, dim3((unsigned)((nn + 255) / 256)), dim3(256), 0, st, P, list_in, n);
    LAUNCH_CHECK();
  }
  if (P.affine) {
    size_t nn = (size_t)(P.N + 1) * P.Bp;
    hipLaunchKernelGGL(k_affine_commit<M>, dim3((unsigned)((nn + 255) / 256)), dim3(256), 0, st, P, a0, 0);
    LAUNCH_CHECK();
  }
  if (!keep) {  // (kept: k_linearize still needs ls_slot AND the length of this stage's list, which decides where a candidate lies;
                // k_ls_begin resets both at the top of the next iteration)
    hipLaunchKernelGGL(k_ls_clear_slot, dim3((P.Bp + 63) / 64), dim3(64), 0, st, P, list_out ^ 1);
    LAUNCH_CHECK();
  }
  return 0;
}

// iLQR_Tracking_SE3_MS loop body (traopt_controller.py:2522-2626)
template <int M>
static int iterate_ms(tolg_handle_s* h, const Params& P, const tolg_options* opt, hipStream_t st, int it0, int n) {
  int rc;
  for (int it = it0; it < it0 + n; it++) {
    int ls_list_last = -1, ls_n_last = 0;  // set by the merit search below: where its last stage left what it accepted
    if ((rc = run_backward<M>(h, P, st, it, 1))) return rc;
    if (!opt->line_search && !opt->rollout_linear && h->prob.kind != TOLG_DYN_PENDULUM3D &&
        opt->schedule != TOLG_SCHED_SPLIT && rl_static_lds<M>() + sizeof(double) * RL_NH * 4 * 16 <= (size_t)h->lds_per_block) {
      // accept-always nonlinear rollout and the re-linearisation of the new trajectory in one launch
      {
        Timed t(h, st, 1, true);
        t.launch(k_rollout_lin<M>, dim3((P.Bp + 15) / 16), dim3(256), P, it);
        LAUNCH_CHECK();
        h->rec_closed = 1;  // its records carry no defect field (zero by construction): K2 reads zeros instead
      }
      continue;  // the fused launch also sums the costs and does the bookkeeping of k_reduce
    } else if (!opt->line_search) {
      if (P.affine) {  // x^ = x (+) e, u^ = u + du from the affine recursion; the statement form for what it hands back
        if ((rc = run_affine_dev<M>(h, P, st, false))) return rc;
        Timed t(h, st, 1);
        const size_t nn = (size_t)(P.N + 1) * P.Bp;
        hipLaunchKernelGGL(k_affine_commit<M>, dim3((unsigned)((nn + 255) / 256)), dim3(256), 0, st, P, 0, 1);
        LAUNCH_CHECK();
      }
      if ((rc = run_rollout_ms<M>(h, P, st, 1.0, opt->rollout_linear))) return rc;
    } else if (P.affine) {
      hipLaunchKernelGGL(k_ls_begin, dim3((P.Bp + 63) / 64), dim3(64), 0, st, P, it == 0 ? 1 : 0);
      LAUNCH_CHECK();
      if ((rc = run_affine_dev<M>(h, P, st, true))) return rc;
      if ((rc = run_ls_stage<M, true>(h, P, st, 0, 0, 1, 1))) return rc;
      if ((rc = run_ls_stage<M, true>(h, P, st, 1, 1, 4, 1))) return rc;
      // (8 + 7 here: on the affine path the deep stages are not empty -- a sixth of the benchmark's trajectories search to the
      // end -- and one stage of 15 was measured slower, 531 -> 487 it/s)
      if ((rc = run_ls_stage<M, true>(h, P, st, 2, 5, 8, 1))) return rc;
      if (!so3_family(h->prob.kind))
        if ((rc = run_ls_stage<M, true>(h, P, st, 3, 13, 7, 1))) return rc;
      hipLaunchKernelGGL(k_ls_finish, dim3((P.Bp + 63) / 64), dim3(64), 0, st, P, it);
      LAUNCH_CHECK();
    } else {
      hipLaunchKernelGGL(k_ls_begin, dim3((P.Bp + 63) / 64), dim3(64), 0, st, P, it == 0 ? 1 : 0);
      LAUNCH_CHECK();
      // the defect weight's linear rollout on a side stream: the first stage's rollout does not need it, its select does
      // (both are 256-wave latency chains; side by side they take the longer one's time, not the sum)
      if (!h->side) {  // all or nothing, as for the active-count buffers of tolg_solve_iterate_until
        hipStream_t sd = nullptr;
        hipEvent_t ev[2] = {nullptr, nullptr};
        bool ok = hipStreamCreateWithFlags(&sd, hipStreamNonBlocking) == hipSuccess;
        for (int k = 0; ok && k < 2; k++) ok = hipEventCreateWithFlags(&ev[k], hipEventDisableTiming) == hipSuccess;
        if (!ok) {
          for (int k = 0; k < 2; k++) if (ev[k]) (void)hipEventDestroy(ev[k]);
          if (sd) (void)hipStreamDestroy(sd);
          return TOLG_E_LAUNCH;
        }
        h->side_ev[0] = ev[0]; h->side_ev[1] = ev[1]; h->side = sd;
      }
      if (hipEventRecord(h->side_ev[0], st) != hipSuccess || hipStreamWaitEvent(h->side, h->side_ev[0], 0) != hipSuccess) return TOLG_E_LAUNCH;
      if (M == 6 && h->prob.kind == TOLG_DYN_PENDULUM3D && opt->schedule == TOLG_SCHED_SPLIT)
        hipLaunchKernelGGL((k_expected_change<6, 1>), dim3((P.Bp * 4 + 63) / 64), dim3(64), 0, h->side, P);
      else if (opt->schedule != TOLG_SCHED_SPLIT) {
        // the affine recursion in the backward sweep's lane map, inputs through an LDS ring (tolg_expected_change.h);
        // behind it the statement-by-statement form for the trajectories it hands back (rotation deviations near pi)
        launch_ec_ring<M>(h, P, h->side);
        LAUNCH_CHECK();
        if (M == 6 && h->prob.kind == TOLG_DYN_PENDULUM3D)
          hipLaunchKernelGGL((k_expected_change<6, 1, true>), dim3((P.Bp * 4 + 63) / 64), dim3(64), 0, h->side, P);
        else
          hipLaunchKernelGGL((k_expected_change<M, 0, true>), dim3((P.Bp * 4 + 63) / 64), dim3(64), 0, h->side, P);
      } else
        hipLaunchKernelGGL((k_expected_change<M, 0>), dim3((P.Bp * 4 + 63) / 64), dim3(64), 0, h->side, P);
      LAUNCH_CHECK();
      if (hipEventRecord(h->side_ev[1], h->side) != hipSuccess) return TOLG_E_LAUNCH;
      // staged: the first try alone (one quad rollout, written in place), then 4 + 8 (+ 7) alphas of the trajectories
      // still undecided -- iLQR_Tracking_SO3_MS searches 13 alphas (:1160), the SE3 one 20 (:2472).  On the benchmark
      // workload ~75 % of the active trajectories accept the first alpha, nearly all the others the second
      // (tools/ls_alpha_histogram.py); one stage of 19 took 5.2 ms against 3 x 0.7 in round 2.
      // (the first TWO step sizes in the first stage -- most trajectories that reject the first accept the second -- was
      // measured: 432 -> 387 it/s; 512 rollout waves of the general MS step beside the expected-change kernel cost more
      // than the nearly empty second stage saves)
      if ((rc = run_ls_stage<M, true>(h, P, st, 0, 0, 1, opt->rollout_linear, h->side_ev[1]))) return rc;
      // (round 4: 1 + 12 + 7 instead of 1 + 4 + 8 + 7 -- one latency chain fewer -- measured: 455 -> 418 it/s; the twelve-wide
      // stage rolls out eight step sizes nobody needed for most of its trajectories)
      // (the last stage leaves what it accepted in its slots: the re-linearisation reads it there)
      const bool so3f = so3_family(h->prob.kind);
      if ((rc = run_ls_stage<M, true>(h, P, st, 1, 1, 4, opt->rollout_linear))) return rc;
      // (round 4, end: the step sizes 5 .. 19 in ONE last stage instead of 8 + 7 -- both are nearly empty on every workload seen,
      // and an empty stage is still seven small launches, 0.05 ms of a 1.6 ms iteration)
      if ((rc = run_ls_stage<M, true>(h, P, st, 2, 5, so3f ? 8 : 15, opt->rollout_linear, nullptr, true))) return rc;
      hipLaunchKernelGGL(k_ls_finish, dim3((P.Bp + 63) / 64), dim3(64), 0, st, P, it);
      LAUNCH_CHECK();
      ls_list_last = 1;   // the list the last stage ran on: (stage - 1) & 1
      ls_n_last = so3f ? 8 : 15;
    }
    // the accepted candidate becomes the nominal trajectory while it is re-linearised
    if ((rc = run_linearize<M>(h, P, st, P.cand, P.cand_u, P.cur, P.cur_u, 1, 0, -1, ls_list_last, ls_n_last))) return rc;
    hipLaunchKernelGGL(k_reduce, dim3((P.Bp + 63) / 64), dim3(64), 0, st, P, it);
    LAUNCH_CHECK();
  }
  return 0;
}

// iLQR_Tracking_SE3 loop body (traopt_controller.py:1926-2007): gradient test and backward pass share
// one sweep; 13-alpha backtracking in two speculative stages (the first try, then the other twelve)
template <int M>
static int iterate_ss(tolg_handle_s* h, const Params& P, const tolg_options* opt, hipStream_t st, int it0, int n) {
  int rc;
  for (int it = it0; it < it0 + n; it++) {
    if ((rc = run_backward<M>(h, P, st, it, 0))) return rc;
    hipLaunchKernelGGL(k_ls_begin, dim3((P.Bp + 63) / 64), dim3(64), 0, st, P, 0);
    LAUNCH_CHECK();
    if (P.affine && (rc = run_affine_dev<M>(h, P, st, false))) return rc;
    if ((rc = run_ls_stage<M, false>(h, P, st, 0, 0, 1, opt->rollout_linear))) return rc;
    // (1 + 4 + 8 like the merit search was measured: 405 -> 339 it/s on iterations 3..23 of the benchmark solve, whose
    // searches end at the 6th to 10th step size -- tools/ls_alpha_histogram.py; it would pay from iteration ~45 on, where
    // the median accepted step size is the second one)
    if ((rc = run_ls_stage<M, false>(h, P, st, 1, 1, NALPHA_SS - 1, opt->rollout_linear, nullptr, true))) return rc;
    hipLaunchKernelGGL(k_ls_finish, dim3((P.Bp + 63) / 64), dim3(64), 0, st, P, it);
    LAUNCH_CHECK();
    // (what the twelve-alpha stage accepted is read from its slots -- no k_ls_copy: 0.15-0.2 ms of a 2.4 ms iteration; on the
    // affine path the candidates are in the candidate arrays, where k_affine_commit wrote them)
    if ((rc = run_linearize<M>(h, P, st, P.cand, P.cand_u, P.cur, P.cur_u, 0, 0, -1, 0, P.affine ? 0 : NALPHA_SS - 1))) return rc;
    hipLaunchKernelGGL(k_reduce, dim3((P.Bp + 63) / 64), dim3(64), 0, st, P, it);
    LAUNCH_CHECK();
  }
  return 0;
}

extern "C" int tolg_solve_begin(tolg_handle_t h, const tolg_options* opt, int32_t B, const double* d_x0_q,
                                const double* d_x0_xi, const double* d_us_init, double* d_J_hist,
                                double* d_grad_hist, double* d_defect_hist, double* d_alpha_hist, double* d_mu_hist,
                                void* stream) {
  if (!h || !opt || B < 1 || B > h->max_batch || !d_x0_q || !d_x0_xi || !d_us_init) return TOLG_E_ARG;
  if (opt->max_iter < 0) return TOLG_E_ARG;
  if (opt->mode != TOLG_MODE_MS && opt->mode != TOLG_MODE_SS) return TOLG_E_ARG;
  const int ms = opt->mode == TOLG_MODE_MS;
  hipStream_t st = static_cast<hipStream_t>(stream);
  Params P = params_for(h, B);
  P.J_hist = d_J_hist; P.grad_hist = d_grad_hist; P.defect_hist = d_defect_hist; P.alpha_hist = d_alpha_hist;
  P.mu_hist = d_mu_hist; P.max_iter = opt->max_iter; P.tol_grad = opt->tol_grad; P.tol_defect = opt->tol_defect;
  P.max_reg = opt->max_reg;
  // rollout = 'linear' as an affine recursion (k_expected_change_ring<.., STORE>): every model since the end of round 4;
  // TOLG_SCHED_SPLIT keeps the statement-form rollouts for every trajectory (the A/B partner in the tests)
  P.affine = (opt->rollout_linear && opt->schedule != TOLG_SCHED_SPLIT) ? 1 : 0;
  size_t n = (size_t)(P.N + 1) * P.Bp;
  hipLaunchKernelGGL(k_init, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, P, d_x0_q, d_x0_xi, d_us_init, ms);
  LAUNCH_CHECK();
  if (!ms) {  // SS: dynamically feasible initial trajectory (_init_rollout, traopt_controller.py:2015-2028)
    if (P.m == 4) hipLaunchKernelGGL(k_init_rollout<4>, dim3((P.Bp + 63) / 64), dim3(64), 0, st, P);
    else hipLaunchKernelGGL(k_init_rollout<6>, dim3((P.Bp + 63) / 64), dim3(64), 0, st, P);
    LAUNCH_CHECK();
  }
  int rc = (P.m == 4) ? run_linearize<4>(h, P, st, P.cur, P.cur_u, nullptr, nullptr, ms)
                      : run_linearize<6>(h, P, st, P.cur, P.cur_u, nullptr, nullptr, ms);
  if (rc) return rc;
  hipLaunchKernelGGL(k_reduce, dim3((P.Bp + 63) / 64), dim3(64), 0, st, P, -1);
  LAUNCH_CHECK();
  h->run = P; h->run_opt = *opt; h->run_it = 0; h->running = true;
  return 0;
}

extern "C" int tolg_solve_iterate(tolg_handle_t h, int32_t n_iter, void* stream) {
  if (!h || !h->running || n_iter < 0) return TOLG_E_ARG;
  if (h->run_it + n_iter > h->run_opt.max_iter) return TOLG_E_ARG;
  hipStream_t st = static_cast<hipStream_t>(stream);
  int rc;
  if (h->run_opt.mode == TOLG_MODE_MS)
    rc = (h->run.m == 4) ? iterate_ms<4>(h, h->run, &h->run_opt, st, h->run_it, n_iter)
                         : iterate_ms<6>(h, h->run, &h->run_opt, st, h->run_it, n_iter);
  else
    rc = (h->run.m == 4) ? iterate_ss<4>(h, h->run, &h->run_opt, st, h->run_it, n_iter)
                         : iterate_ss<6>(h, h->run, &h->run_opt, st, h->run_it, n_iter);
  if (rc) return rc;
  h->run_it += n_iter;
  return 0;
}

static int solve_export(tolg_handle_t h, double* d_xs_q, double* d_xs_xi, double* d_us, int32_t* d_iters,
                        int32_t* d_status, int32_t* d_converged, void* stream, bool end) {
  if (!h || !h->running) return TOLG_E_ARG;
  hipStream_t st = static_cast<hipStream_t>(stream);
  const Params& P = h->run;
  size_t n = (size_t)(P.N + 1) * P.Bp;
  hipLaunchKernelGGL(k_unpack_traj, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, P, P.cur, P.cur_u, d_xs_q,
                     d_xs_xi, d_us);
  LAUNCH_CHECK();
  hipLaunchKernelGGL(k_export_scalars, dim3((P.B + 63) / 64), dim3(64), 0, st, P, nullptr, nullptr, nullptr, nullptr,
                     d_iters, d_status, d_converged);
  LAUNCH_CHECK();
  if (end) h->running = false;
  return 0;
}
extern "C" int tolg_solve_end(tolg_handle_t h, double* d_xs_q, double* d_xs_xi, double* d_us, int32_t* d_iters,
                              int32_t* d_status, int32_t* d_converged, void* stream) {
  return solve_export(h, d_xs_q, d_xs_xi, d_us, d_iters, d_status, d_converged, stream, true);
}
extern "C" int tolg_solve_peek(tolg_handle_t h, double* d_xs_q, double* d_xs_xi, double* d_us, int32_t* d_iters,
                               int32_t* d_status, int32_t* d_converged, void* stream) {
  return solve_export(h, d_xs_q, d_xs_xi, d_us, d_iters, d_status, d_converged, stream, false);
}

extern "C" int tolg_solve_active_count(tolg_handle_t h, int32_t* d_count, void* stream) {
  if (!h || !h->running || !d_count) return TOLG_E_ARG;
  hipStream_t st = static_cast<hipStream_t>(stream);
  if (hipMemsetAsync(d_count, 0, sizeof(int32_t), st) != hipSuccess) return TOLG_E_LAUNCH;
  hipLaunchKernelGGL(k_active_count, dim3((h->run.B + 255) / 256), dim3(256), 0, st, h->run, d_count);
  LAUNCH_CHECK();
  return 0;
}

extern "C" int tolg_solve_iterate_until(tolg_handle_t h, int32_t n_iter, int32_t check_every, int32_t* n_issued, void* stream) {
  if (!h || !h->running || n_iter < 0 || check_every < 0) return TOLG_E_ARG;
  if (h->run_it + n_iter > h->run_opt.max_iter) return TOLG_E_ARG;
  hipStream_t st = static_cast<hipStream_t>(stream);
  const tolg_options& o = h->run_opt;
  // nothing can end a solve whose tolerances are zero (accept-always MS): one slice, no read-back
  const bool can_stop = check_every > 0 && (o.tol_grad > 0 || o.line_search || o.mode != TOLG_MODE_MS);
  int rc, done = 0;
  if (!can_stop) {
    if ((rc = tolg_solve_iterate(h, n_iter, stream))) return rc;
    if (n_issued) *n_issued = n_iter;
    return 0;
  }
  if (!h->d_cnt) {
    // all or nothing: the handle sees the counters only when every piece exists (a partial set-up left behind by a failed
    // call would be taken for a complete one by the next call)
    int *dc = nullptr, *hc = nullptr;
    hipEvent_t ev[2] = {nullptr, nullptr};
    bool ok = hipMalloc((void**)&dc, 2 * sizeof(int)) == hipSuccess &&
              hipHostMalloc((void**)&hc, 2 * sizeof(int), hipHostMallocDefault) == hipSuccess;
    for (int k = 0; ok && k < 2; k++) ok = hipEventCreateWithFlags(&ev[k], hipEventDisableTiming) == hipSuccess;
    if (!ok) {
      if (dc) (void)hipFree(dc);
      if (hc) (void)hipHostFree(hc);
      for (int k = 0; k < 2; k++) if (ev[k]) (void)hipEventDestroy(ev[k]);
      return TOLG_E_LAUNCH;
    }
    h->h_cnt = hc; h->cnt_ev[0] = ev[0]; h->cnt_ev[1] = ev[1]; h->d_cnt = dc;
  }
  // Slices of check_every iterations; behind each, the count of trajectories still iterating goes to pinned host
  // memory.  The host looks at the count of the slice BEFORE the one it has just queued, so the device always has a
  // slice in its queue while the host waits: a converged batch costs at most one slice of launches whose workgroups
  // exit at once (traopt_controller.py:2528-2532 is the per-trajectory exit this adds up to).
  int slice = 0, pending = -1;
  while (done < n_iter) {
    const int step = (n_iter - done < check_every) ? n_iter - done : check_every;
    if ((rc = tolg_solve_iterate(h, step, stream))) return rc;
    done += step;
    const int k = slice & 1;
    if (hipMemsetAsync(h->d_cnt + k, 0, sizeof(int), st) != hipSuccess) return TOLG_E_LAUNCH;
    hipLaunchKernelGGL(k_active_count, dim3((h->run.B + 255) / 256), dim3(256), 0, st, h->run, h->d_cnt + k);
    LAUNCH_CHECK();
    if (hipMemcpyAsync(h->h_cnt + k, h->d_cnt + k, sizeof(int), hipMemcpyDeviceToHost, st) != hipSuccess) return TOLG_E_LAUNCH;
    if (hipEventRecord(h->cnt_ev[k], st) != hipSuccess) return TOLG_E_LAUNCH;
    if (pending >= 0) {
      if (hipEventSynchronize(h->cnt_ev[pending]) != hipSuccess) return TOLG_E_LAUNCH;
      if (h->h_cnt[pending] == 0) break;
    }
    pending = k;
    slice++;
  }
  if (n_issued) *n_issued = done;
  return 0;
}

extern "C" int tolg_solve_batch(tolg_handle_t h, const tolg_options* opt, int32_t B, const double* d_x0_q,
                                const double* d_x0_xi, const double* d_us_init, double* d_xs_q, double* d_xs_xi,
                                double* d_us, double* d_J_hist, double* d_grad_hist, double* d_defect_hist,
                                double* d_alpha_hist, double* d_mu_hist, int32_t* d_iters, int32_t* d_status,
                                int32_t* d_converged, void* stream) {
  int rc = tolg_solve_begin(h, opt, B, d_x0_q, d_x0_xi, d_us_init, d_J_hist, d_grad_hist, d_defect_hist, d_alpha_hist,
                            d_mu_hist, stream);
  if (rc) return rc;
  if ((rc = tolg_solve_iterate_until(h, opt->max_iter, opt->check_every, nullptr, stream))) return rc;
  return tolg_solve_end(h, d_xs_q, d_xs_xi, d_us, d_iters, d_status, d_converged, stream);
}

extern "C" int tolg_set_al(tolg_handle_t h, const double* d_lb, const double* d_ub, const double* d_lambda,
                           const double* d_imu) {
  if (!h || h->running) return TOLG_E_ARG;
  if (d_lb && (!d_ub || !d_lambda || !d_imu)) return TOLG_E_ARG;
  h->al_lb = d_lb; h->al_ub = d_lb ? d_ub : nullptr;
  h->al_lambda = d_lb ? d_lambda : nullptr; h->al_imu = d_lb ? d_imu : nullptr;
  return 0;
}

extern "C" int tolg_al_update(tolg_handle_t h, int32_t B, const double* d_us, const double* d_lb, const double* d_ub,
                              double* d_lambda, double* d_imu, double* d_mu, double mu_scale, double mu_max,
                              double tol_constr, double* d_maxviol, int32_t* d_al_converged, void* stream) {
  if (!h || B < 1 || !d_us || !d_lb || !d_ub || !d_lambda || !d_imu || !d_mu || !d_maxviol || !d_al_converged)
    return TOLG_E_ARG;
  hipStream_t st = static_cast<hipStream_t>(stream);
  hipLaunchKernelGGL(k_al_update, dim3((B + 63) / 64), dim3(64), 0, st, B, h->prob.N, h->prob.m, d_us, d_lb, d_ub,
                     d_lambda, d_imu, d_mu, mu_scale, mu_max, tol_constr, d_maxviol, d_al_converged);
  LAUNCH_CHECK();
  return 0;
}

extern "C" int tolg_linearize_backward(tolg_handle_t h, int32_t ms, double max_reg, int32_t B, const double* d_xs_q,
                                       const double* d_xs_xi, const double* d_us, double* d_mu_delta, double* d_Fx,
                                       double* d_d, double* d_lx, double* d_lxx11, double* d_k, double* d_K,
                                       double* d_J, double* d_dnorm, double* d_grad, void* stream) {
  // uses the handle's workspace (k_pack_traj resets the trajectories, mu / delta, the masks): not during a solve
  if (!h || h->running || B < 1 || B > h->max_batch || !d_xs_q || !d_xs_xi || !d_us) return TOLG_E_ARG;
  hipStream_t st = static_cast<hipStream_t>(stream);
  Params P = params_for(h, B);
  P.max_reg = max_reg;
  size_t n = (size_t)(P.N + 1) * P.Bp;
  hipLaunchKernelGGL(k_pack_traj, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, P, d_xs_q, d_xs_xi, d_us,
                     (const double*)d_mu_delta);
  LAUNCH_CHECK();
  int rc;
  if (P.m == 4) {
    if ((rc = run_linearize<4>(h, P, st, P.cur, P.cur_u, nullptr, nullptr, ms))) return rc;
  } else {
    if ((rc = run_linearize<6>(h, P, st, P.cur, P.cur_u, nullptr, nullptr, ms))) return rc;
  }
  hipLaunchKernelGGL(k_reduce, dim3((P.Bp + 63) / 64), dim3(64), 0, st, P, -1);
  LAUNCH_CHECK();
  if ((rc = (P.m == 4) ? run_backward<4>(h, P, st, -1, ms) : run_backward<6>(h, P, st, -1, ms))) return rc;
  size_t ne = (size_t)(P.N + 1) * B;
  hipLaunchKernelGGL(k_export_lin, dim3((unsigned)((ne + 127) / 128)), dim3(128), 0, st, P, d_Fx, d_d, d_lx, d_lxx11,
                     d_k, d_K);
  LAUNCH_CHECK();
  hipLaunchKernelGGL(k_export_scalars, dim3((B + 63) / 64), dim3(64), 0, st, P, d_J, d_dnorm, d_grad, d_mu_delta,
                     nullptr, nullptr, nullptr);
  LAUNCH_CHECK();
  return 0;
}

extern "C" int tolg_eval_knot(tolg_handle_t h, int32_t i, int32_t n, const double* d_x_q, const double* d_x_xi,
                              const double* d_u, double* d_f_q, double* d_f_xi, double* d_Fx, double* d_Fu,
                              double* d_l, double* d_lx, double* d_lxx, double* d_lu, double* d_luu, double* d_err,
                              void* stream) {
  if (!h || h->running || n < 1 || n > h->max_batch || i < 0 || i > h->prob.N || !d_x_q || !d_x_xi) return TOLG_E_ARG;
  if (i < h->prob.N && !d_u) return TOLG_E_ARG;
  hipStream_t st = static_cast<hipStream_t>(stream);
  Params P = params_for(h, n);
  hipLaunchKernelGGL(k_probe_pack, dim3((P.Bp + 63) / 64), dim3(64), 0, st, P, i, d_x_q, d_x_xi, d_u);
  LAUNCH_CHECK();
  int rc = (P.m == 4) ? run_linearize<4>(h, P, st, P.cur, P.cur_u, nullptr, nullptr, 2, i, 1)
                      : run_linearize<6>(h, P, st, P.cur, P.cur_u, nullptr, nullptr, 2, i, 1);
  if (rc) return rc;
  hipLaunchKernelGGL(k_probe_export, dim3((n + 63) / 64), dim3(64), 0, st, P, i, d_f_q, d_f_xi, d_Fx, d_Fu, d_l, d_lx,
                     d_lxx, d_lu, d_luu, d_err);
  LAUNCH_CHECK();
  return 0;
}

__global__ void k_export_ecc(Params P, double* __restrict__ ecc, int* __restrict__ flag) {
  int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= P.B) return;
  if (ecc) { ecc[2 * b] = P.ecc[2 * b]; ecc[2 * b + 1] = P.ecc[2 * b + 1]; }
  if (flag) flag[b] = P.ec_redo[b];
}
__global__ void k_clear_ecc(Params P) {
  int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= P.Bp) return;
  P.ecc[2 * b] = __builtin_nan(""); P.ecc[2 * b + 1] = __builtin_nan(""); P.ec_redo[b] = 0;
  P.dweight[2 * b] = 10.0; P.dweight[2 * b + 1] = 10.0;
}
extern "C" int tolg_expected_change(tolg_handle_t h, int32_t form, int32_t B, double* d_ecc, int32_t* d_flag, void* stream) {
  // works on what tolg_linearize_backward left in the workspace (trajectory, records, gains): not during a solve
  if (!h || h->running || B < 1 || B > h->max_batch || form < 0 || form > 2) return TOLG_E_ARG;
  hipStream_t st = static_cast<hipStream_t>(stream);
  Params P = params_for(h, B);
  const bool pend = h->prob.kind == TOLG_DYN_PENDULUM3D;
  hipLaunchKernelGGL(k_clear_ecc, dim3((P.Bp + 63) / 64), dim3(64), 0, st, P);
  LAUNCH_CHECK();
  hipLaunchKernelGGL(k_poison_lds, dim3(2048), dim3(256), 0, st);
  LAUNCH_CHECK();
  const dim3 gq((P.Bp * 4 + 63) / 64), gr(P.Bp / 4), blk(64);
  if (form == 0) {
    if (P.m == 4) hipLaunchKernelGGL((k_expected_change<4, 0>), gq, blk, 0, st, P);
    else if (h->prob.kind == TOLG_DYN_PENDULUM3D) hipLaunchKernelGGL((k_expected_change<6, 1>), gq, blk, 0, st, P);
    else hipLaunchKernelGGL((k_expected_change<6, 0>), gq, blk, 0, st, P);
  } else {
    if (P.m == 4) launch_ec_ring<4>(h, P, st);
    else launch_ec_ring<6>(h, P, st);
    LAUNCH_CHECK();
    if (form == 2) {
      if (P.m == 4) hipLaunchKernelGGL((k_expected_change<4, 0, true>), gq, blk, 0, st, P);
      else if (pend) hipLaunchKernelGGL((k_expected_change<6, 1, true>), gq, blk, 0, st, P);
      else hipLaunchKernelGGL((k_expected_change<6, 0, true>), gq, blk, 0, st, P);
    }
  }
  LAUNCH_CHECK();
  hipLaunchKernelGGL(k_export_ecc, dim3((B + 63) / 64), dim3(64), 0, st, P, d_ecc, d_flag);
  LAUNCH_CHECK();
  return 0;
}

extern "C" int tolg_rollout(tolg_handle_t h, int32_t ms, int32_t rollout_linear, double alpha, int32_t B,
                            double* d_xs_q_new, double* d_xs_xi_new, double* d_us_new, void* stream) {
  if (!h || h->running || B < 1 || B > h->max_batch) return TOLG_E_ARG;  // overwrites the candidate arrays
  hipStream_t st = static_cast<hipStream_t>(stream);
  Params P = params_for(h, B);
  int rc = (P.m == 4) ? run_rollout_ms<4>(h, P, st, alpha, rollout_linear, ms) : run_rollout_ms<6>(h, P, st, alpha, rollout_linear, ms);
  if (rc) return rc;
  size_t n = (size_t)(P.N + 1) * P.Bp;
  hipLaunchKernelGGL(k_unpack_traj, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, P, P.cand, P.cand_u, d_xs_q_new,
                     d_xs_xi_new, d_us_new);
  LAUNCH_CHECK();
  return 0;
}

extern "C" int tolg_selftest_series(int32_t n, const double* d_args, double* d_out, void* stream) {
  if (n < 64 || (n % 64) != 0 || !d_args || !d_out) return TOLG_E_ARG;
  hipLaunchKernelGGL(k_selftest_series, dim3(n / 64), dim3(64), 0, static_cast<hipStream_t>(stream), n, d_args, d_out);
  LAUNCH_CHECK();
  return 0;
}

#ifdef TOLG_TIER_COUNT
// diagnostic builds only (tools/tier_share.py): read / reset the series-tier counters of tolg_lie.h.  Not in include/tolg.h.
extern "C" int tolg_debug_tier_counts(unsigned long long* out6, int reset) {
  if (hipDeviceSynchronize() != hipSuccess) return TOLG_E_LAUNCH;
  if (out6 && hipMemcpyFromSymbol(out6, HIP_SYMBOL(tolg::g_tier), 6 * sizeof(unsigned long long)) != hipSuccess) return TOLG_E_LAUNCH;
  if (reset) {
    const unsigned long long z[6] = {0, 0, 0, 0, 0, 0};
    if (hipMemcpyToSymbol(HIP_SYMBOL(tolg::g_tier), z, sizeof z) != hipSuccess) return TOLG_E_LAUNCH;
  }
  return 0;
}
#endif
