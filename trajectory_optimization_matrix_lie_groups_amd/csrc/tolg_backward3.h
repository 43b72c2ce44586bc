// tolg_backward3.h -- K2, third form (round 3): the Riccati sweep of traopt_controller.py:2912-3068 for models whose
// input matrix has one entry per column (diagonal inertia blocks: every reference script), i.e. SE3 / RigidBody / SO3
// (m = 6) and Drone (m = 4).  Included by tolg_kernels.hip inside namespace tolg, after the K2 helpers.
//
// Same lane map as k_backward (16 lanes per trajectory, lane j = column j of every 12 x 12, lane 12 the vector column,
// lane 13 the single-shooting adjoint; products by DPP row_newbcast fused into v_fmac_f64).  What changed, and why
// (profiles/r02_final_sq_counters.json: 61 % VALU, 23 % parked on s_waitcnt, 13 % other instruction issue):
//
// 1. The knot records come through LDS.  k_backward fetched its fields with 22-31 gather loads per wave and knot
//    (every lane its own 8 bytes) into a register buffer that the next knot's prefetch overwrote mid-step -- 62 VGPRs,
//    ~55 copies out of them per knot, and a wait at the top of every step.  Here one LDS-DMA burst (4-5
//    global_load_lds_dwordx4, 1 KB each, fully coalesced: the four trajectories of a wave own one contiguous run per
//    knot) brings a knot into a 2-slot LDS ring a whole step ahead, and each lane picks its fields with ds_read
//    straight into the registers that use them.  Structural zeros are read from a zeroed LDS pad.  The velocity block
//    F_x[6:12,6:12] = I + H dt is not in the record at all (REC_XI, the knot's twist, is): with diagonal inertia blocks
//    a column of it is the identity column plus, per 3x3 block, two entries of the form alpha w_k + beta v_k with
//    lane constants -- 18 multiply-adds against 30 fields of record (three LDS-DMA instructions per knot, not four).
// 2. F_u = [0; B] with B = D S (D diagonal m x m, S a row selector), so Q_uu = 2R + D (V + mu I)_SS D is congruent to
//    Mt = (V + mu I)_SS + 2 D^-1 R D^-1 (+ D^-1 l_uu^AL D^-1), and Q_ux = D G with G = rows S of (V + mu I) F_x.
//    Mt's columns already sit in the lanes that hold the columns of V_SS: no T = B^T V product, no row shift; the
//    pivots of Mt are those of Q_uu divided by positive numbers, so "every pivot > 0" is still is_pos_def(Q_uu+Q_uu^T).
// 3. V' = Q_xx - G^T Mt^-1 G = Q_xx - Y^T Dl^-1 Y with Y = L^-1 G (Mt = L Dl L^T): only the FORWARD substitution is on
//    the path to the next knot; the backward substitution (gains K = -D^-1 L^-T Dl^-1 Y, needed for the output only)
//    runs beside the rank-m update.  The update term is symmetric up to the rounding of its fused multiply-adds, so
//    what has to be symmetrised (the reference's V <- (V + V^T)/2, :3004) is Q_xx = l_xx + F_x^T V F_x alone -- and
//    that is known long before the factorisation ends: its LDS transpose round trip, which sat on the critical path
//    of every knot in k_backward, passes behind the factorisation here.  It cannot be dropped altogether: the
//    antisymmetric part of V is an unstable mode of this form of the recursion (S' = F_x^T S F_x + (B K)^T S (B K): the
//    cross terms that make the closed loop contract are symmetric and never see it); left alone it grew from 1e-16 to
//    1e-8 over 130-200 knots (tests/test_gpu_fused.py, test_gpu_parity.py caught it).  It does not have to happen at
//    every knot either (the reference symmetrises every knot, :3004): since late round 3 every TOLG_K3_SYMP-th knot
//    (4; note at `sym_now` below) -- between two symmetrisations the mode grows by ~1.1 per knot on the benchmark
//    workload (4 at worst), i.e. to < 3e-14 of |V| before it is removed; tests/test_gpu_symmetrisation.py bounds the
//    difference to the every-knot oracle on long-horizon, low-R problems.  Since the end of round 4 that period is the FAST
//    kernel's alone: the full kernel symmetrises at every knot (note at `sym_now`: far into a divergence the mode outgrows
//    four knots, and a fast sweep that meets a non-positive pivot hands its group to the full kernel).
template <int M>
__host__ __device__ constexpr int urow(int u) { return (u < 3 || M == 6) ? 6 + u : 11; }  // state row driven by input u

#ifndef TOLG_DPP_BUILTIN
#define DF3(acc, p, q, L) "v_fmac_f64_dpp " acc ", " p ", " q " row_newbcast:" L " row_mask:0xf bank_mask:0xf\n\t"
// a[i] += a[i]@lane LN * w for the rows below pivot J
template <int M, int J, int LN>
TOLG_DEV void ldl3_update(double (&a)[M], double w) {
  constexpr int R = M - 1 - J;
  // Pivot 0: the columns come straight from compiler-scheduled code, which may place the instruction that writes one
  // of them directly in front of the block (a VALU write needs two wait states before a DPP read; nothing interlocks).
  // The nop takes the columns as operands so that their producers cannot sink below it.  Later pivots read what the
  // previous update block wrote, several instructions back in program order.
  if constexpr (J == 0) {
    if constexpr (M == 6)
      asm volatile("s_nop 1" : "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]));
    else
      asm volatile("s_nop 1" : "+v"(a[1]), "+v"(a[2]), "+v"(a[3]));
  }
  if constexpr (R == 5)
    asm volatile(DF3("%0", "%0", "%5", "%6") DF3("%1", "%1", "%5", "%6") DF3("%2", "%2", "%5", "%6")
                     DF3("%3", "%3", "%5", "%6") DF3("%4", "%4", "%5", "%6")
                 : "+v"(a[J + 1]), "+v"(a[J + 2]), "+v"(a[J + 3]), "+v"(a[J + 4]), "+v"(a[J + 5]) : "v"(w), "n"(LN));
  if constexpr (R == 4)
    asm volatile(DF3("%0", "%0", "%4", "%5") DF3("%1", "%1", "%4", "%5") DF3("%2", "%2", "%4", "%5")
                     DF3("%3", "%3", "%4", "%5")
                 : "+v"(a[J + 1]), "+v"(a[J + 2]), "+v"(a[J + 3]), "+v"(a[J + 4]) : "v"(w), "n"(LN));
  if constexpr (R == 3)
    asm volatile(DF3("%0", "%0", "%3", "%4") DF3("%1", "%1", "%3", "%4") DF3("%2", "%2", "%3", "%4")
                 : "+v"(a[J + 1]), "+v"(a[J + 2]), "+v"(a[J + 3]) : "v"(w), "n"(LN));
  if constexpr (R == 2)
    asm volatile(DF3("%0", "%0", "%2", "%3") DF3("%1", "%1", "%2", "%3")
                 : "+v"(a[J + 1]), "+v"(a[J + 2]) : "v"(w), "n"(LN));
  if constexpr (R == 1)
    asm volatile(DF3("%0", "%0", "%1", "%2") : "+v"(a[J + 1]) : "v"(w), "n"(LN));
}
// y += sum_{k < I} p@lane urow(k) * q[k]  (row I of the forward substitution; p = this lane's a[I])
template <int M, int I, int LO = 0>
TOLG_DEV void ldl3_fwd_row(double& y, double p, const double (&q)[M]) {
  // (the wait states between a column's return from an AGPR and its DPP read: ldl3_tie_columns, in front of row 1)
  if constexpr (I == 1)
    asm volatile(DF3("%0", "%1", "%2", "%3") : "+v"(y) : "v"(p), "v"(q[0]), "n"(urow<M>(0) + LO));
  if constexpr (I == 2)
    asm volatile(DF3("%0", "%1", "%2", "%4") DF3("%0", "%1", "%3", "%5")
                 : "+v"(y) : "v"(p), "v"(q[0]), "v"(q[1]), "n"(urow<M>(0) + LO), "n"(urow<M>(1) + LO));
  if constexpr (I == 3)
    asm volatile(DF3("%0", "%1", "%2", "%5") DF3("%0", "%1", "%3", "%6") DF3("%0", "%1", "%4", "%7")
                 : "+v"(y) : "v"(p), "v"(q[0]), "v"(q[1]), "v"(q[2]), "n"(urow<M>(0) + LO), "n"(urow<M>(1) + LO), "n"(urow<M>(2) + LO));
  if constexpr (I == 4)
    asm volatile(DF3("%0", "%1", "%2", "%6") DF3("%0", "%1", "%3", "%7") DF3("%0", "%1", "%4", "%8")
                     DF3("%0", "%1", "%5", "%9")
                 : "+v"(y) : "v"(p), "v"(q[0]), "v"(q[1]), "v"(q[2]), "v"(q[3]), "n"(urow<M>(0) + LO), "n"(urow<M>(1) + LO),
                   "n"(urow<M>(2) + LO), "n"(urow<M>(3) + LO));
  if constexpr (I == 5)
    asm volatile(DF3("%0", "%1", "%2", "%7") DF3("%0", "%1", "%3", "%8") DF3("%0", "%1", "%4", "%9")
                     DF3("%0", "%1", "%5", "%10") DF3("%0", "%1", "%6", "%11")
                 : "+v"(y) : "v"(p), "v"(q[0]), "v"(q[1]), "v"(q[2]), "v"(q[3]), "v"(q[4]), "n"(urow<M>(0) + LO),
                   "n"(urow<M>(1) + LO), "n"(urow<M>(2) + LO), "n"(urow<M>(3) + LO), "n"(urow<M>(4) + LO));
}
// t += sum_{k > I} a[k]@lane LN * x[k]  (row I of the back substitution: column I of the factor lives in lane LN = urow(I))
template <int M, int I, int LN>
TOLG_DEV void ldl3_bwd_row(double& t, const double (&a)[M], const double (&x)[M]) {
  constexpr int R = M - 1 - I;
  if constexpr (R == 1) asm volatile(DF3("%0", "%1", "%2", "%3") : "+v"(t) : "v"(a[I + 1]), "v"(x[I + 1]), "n"(LN));
  if constexpr (R == 2)
    asm volatile(DF3("%0", "%1", "%3", "%5") DF3("%0", "%2", "%4", "%5")
                 : "+v"(t) : "v"(a[I + 1]), "v"(a[I + 2]), "v"(x[I + 1]), "v"(x[I + 2]), "n"(LN));
  if constexpr (R == 3)
    asm volatile(DF3("%0", "%1", "%4", "%7") DF3("%0", "%2", "%5", "%7") DF3("%0", "%3", "%6", "%7")
                 : "+v"(t) : "v"(a[I + 1]), "v"(a[I + 2]), "v"(a[I + 3]), "v"(x[I + 1]), "v"(x[I + 2]), "v"(x[I + 3]), "n"(LN));
  if constexpr (R == 4)
    asm volatile(DF3("%0", "%1", "%5", "%9") DF3("%0", "%2", "%6", "%9") DF3("%0", "%3", "%7", "%9")
                     DF3("%0", "%4", "%8", "%9")
                 : "+v"(t) : "v"(a[I + 1]), "v"(a[I + 2]), "v"(a[I + 3]), "v"(a[I + 4]), "v"(x[I + 1]), "v"(x[I + 2]),
                   "v"(x[I + 3]), "v"(x[I + 4]), "n"(LN));
  if constexpr (R == 5)
    asm volatile(DF3("%0", "%1", "%6", "%11") DF3("%0", "%2", "%7", "%11") DF3("%0", "%3", "%8", "%11")
                     DF3("%0", "%4", "%9", "%11") DF3("%0", "%5", "%10", "%11")
                 : "+v"(t) : "v"(a[I + 1]), "v"(a[I + 2]), "v"(a[I + 3]), "v"(a[I + 4]), "v"(a[I + 5]), "v"(x[I + 1]),
                   "v"(x[I + 2]), "v"(x[I + 3]), "v"(x[I + 4]), "v"(x[I + 5]), "n"(LN));
}
#else
template <int M, int J, int LN>
TOLG_DEV void ldl3_update(double (&a)[M], double w) {
#pragma unroll
  for (int i = J + 1; i < M; i++) a[i] += bcast<LN>(a[i]) * w;
}
template <int M, int I, int LO = 0>
TOLG_DEV void ldl3_fwd_row(double& y, double p, const double (&q)[M]) {
  if constexpr (I > 0) y += bcast<urow<M>(0) + LO>(p) * q[0];
  if constexpr (I > 1) y += bcast<urow<M>(1) + LO>(p) * q[1];
  if constexpr (I > 2) y += bcast<urow<M>(2) + LO>(p) * q[2];
  if constexpr (I > 3) y += bcast<urow<M>(3) + LO>(p) * q[3];
  if constexpr (I > 4) y += bcast<urow<M>(4) + LO>(p) * q[4];
}
template <int M, int I, int LN>
TOLG_DEV void ldl3_bwd_row(double& t, const double (&a)[M], const double (&x)[M]) {
#pragma unroll
  for (int k = I + 1; k < M; k++) t += bcast<LN>(a[k]) * x[k];
}
#endif

// Issue costs that shape this code (profiles/r03_valu_issue_microbench.txt, one wave per SIMD, cycles per instruction):
// v_fma_f64 / v_fmac_f64_dpp 5.6 independent, 8.9 dependent; v_mov_b64 8.2; s_nop 1 8.9; v_rcp_f64 17.  The sweep is
// the SUM of its issue costs (interleaving independent work into the chains bought nothing), so what counts is the
// number of instructions and their kind: no hazard nops where the DPP operand is old, no register copies, no
// zero-initialised accumulators, the negated reciprocal straight from v_rcp_f64(-d).
//
// Mt (m x m, column c in lane urow(c)) factored where it lies: right-looking L Dl L^T.  wm[J] is this lane's update
// mask for pivot J: 1 for the columns right of the pivot, 0 for every other lane (their rows below J are final factor
// entries, or not part of Mt at all).  On return a[i] (i > c) of lane urow(c) holds L[i][c] Dl_c, nri[J] = -1 / Dl_J in
// every lane, d[J] the pivots ("every pivot > 0" is the positive-definiteness test).
#ifndef TOLG_DPP_BUILTIN
// pivot J broadcast to the row (+ this lane's masked multiplier numerator).  Wait states in front of the DPP read of
// a[J]: it was written by the first multiply-add of the previous pivot's update, M - 1 - J more of them follow, then
// the v_mul here -- enough except behind the last update (nothing in between) and at pivot 0 (a[0] comes straight from
// the instruction in front of the statement).
template <int M, int J, int LO = 0>
TOLG_DEV void ldl3_head(const double (&a)[M], double wmJ, double& d, double& pre) {
  if constexpr (J == M - 1) {
    asm volatile("s_nop 1\n\tv_mov_b64_dpp %0, %1 row_newbcast:%2 row_mask:0xf bank_mask:0xf" : "=&v"(d) : "v"(a[J]), "n"(urow<M>(J) + LO));
    pre = 0.0;
  } else if constexpr (J == 0) {
    asm volatile("v_mul_f64 %1, %2, %3\n\ts_nop 0\n\tv_mov_b64_dpp %0, %2 row_newbcast:%4 row_mask:0xf bank_mask:0xf"
                 : "=&v"(d), "=&v"(pre) : "v"(a[J]), "v"(wmJ), "n"(urow<M>(J) + LO));
  } else {
    asm volatile("v_mul_f64 %1, %2, %3\n\tv_mov_b64_dpp %0, %2 row_newbcast:%4 row_mask:0xf bank_mask:0xf"
                 : "=&v"(d), "=&v"(pre) : "v"(a[J]), "v"(wmJ), "n"(urow<M>(J) + LO));
  }
}
#else
template <int M, int J, int LO = 0>
TOLG_DEV void ldl3_head(const double (&a)[M], double wmJ, double& d, double& pre) {
  d = bcast<urow<M>(J) + LO>(a[J]);
  pre = a[J] * wmJ;
}
#endif
template <int M, int J = 0, int LO = 0>
TOLG_DEV void ldl3_factor(double (&a)[M], double (&nri)[M], double (&d)[M], const double (&wm)[M]) {
  double pre;
  ldl3_head<M, J, LO>(a, wm[J], d[J], pre);
  // -1 / d: v_rcp_f64 of -d, refined (below).  A non-positive pivot leaves garbage behind it:
  // the caller discards the factors then.  (One step instead of two, 2.2e-15, was measured and is no faster.)
  // (late round 3: ONE third-order step x (1 + e + e^2), e = 1 + d x, instead of two Newton steps -- three dependent
  // multiply-adds instead of four, and e^3 ~ 1e-22 is below the e^4 of the pair only on paper: both are exact to the last bit)
  double x = __builtin_amdgcn_rcp(-d[J]);
  const double e = fma(d[J], x, 1.0);
  x = fma(e, fma(x, e, x), x);
  nri[J] = x;
  if constexpr (J + 1 < M) {
    ldl3_update<M, J, urow<M>(J) + LO>(a, pre * x);  // a[i] -= a[i]@pivot lane * a[J] / d for the columns right of the pivot
    ldl3_factor<M, J + 1, LO>(a, nri, d, wm);
  }
}
template <int M>
TOLG_DEV bool ldl3_all_positive(const double (&d)[M]) {
  bool ok = d[0] > 0.0;
#pragma unroll
  for (int u = 1; u < M; u++) ok = ok && (d[u] > 0.0);
  return ok;
}
// Two wait states with the factor columns a[1..M-1] as operands: every one of them is a DPP operand of the rows that
// follow, and the allocator is free to bring any of them back from an AGPR directly in front of its row (the lint found
// exactly that in a build whose allocation differed).  Tied to one nop they are in VGPRs here, ahead of all rows.
template <int M>
TOLG_DEV void ldl3_tie_columns(const double (&a)[M]) {
#ifndef TOLG_DPP_BUILTIN
  double(&w)[M] = const_cast<double(&)[M]>(a);
  if constexpr (M == 6) asm volatile("s_nop 1" : "+v"(w[1]), "+v"(w[2]), "+v"(w[3]), "+v"(w[4]), "+v"(w[5]));
  else asm volatile("s_nop 1" : "+v"(w[1]), "+v"(w[2]), "+v"(w[3]));
#endif
}
// forward substitution in place: y <- L^-1 y, zn_k = -y_k / Dl_k (nri = -1 / Dl, zeroed in the adjoint lane)
template <int M, int LO = 0>
TOLG_DEV void ldl3_forward(const double (&a)[M], const double (&nri)[M], double (&y)[M], double (&zn)[M]) {
  ldl3_tie_columns<M>(a);
  zn[0] = y[0] * nri[0];
  if constexpr (M > 1) { ldl3_fwd_row<M, 1, LO>(y[1], a[1], zn); zn[1] = y[1] * nri[1]; }
  if constexpr (M > 2) { ldl3_fwd_row<M, 2, LO>(y[2], a[2], zn); zn[2] = y[2] * nri[2]; }
  if constexpr (M > 3) { ldl3_fwd_row<M, 3, LO>(y[3], a[3], zn); zn[3] = y[3] * nri[3]; }
  if constexpr (M > 4) { ldl3_fwd_row<M, 4, LO>(y[4], a[4], zn); zn[4] = y[4] * nri[4]; }
  if constexpr (M > 5) { ldl3_fwd_row<M, 5, LO>(y[5], a[5], zn); zn[5] = y[5] * nri[5]; }
}
// back substitution on nx = -x (x = Mt^-1 of the right-hand side), IN PLACE on the forward result y, which nobody
// needs any more once the rank-m update has read it: y_i += sum_{k>i} (L Dl)[k][i] nx_k, nx_i = -y_i / Dl_i;
// nx_{M-1} = zn_{M-1}.  No zero-initialised accumulators, no copies.
template <int M, int LO = 0>
TOLG_DEV void ldl3_backward_nx(const double (&a)[M], const double (&nri)[M], double (&y)[M], const double (&zn)[M], double (&nx)[M]) {
  nx[M - 1] = zn[M - 1];
  if constexpr (M > 1) { ldl3_bwd_row<M, M - 2, urow<M>(M - 2) + LO>(y[M - 2], a, nx); nx[M - 2] = y[M - 2] * nri[M - 2]; }
  if constexpr (M > 2) { ldl3_bwd_row<M, M - 3, urow<M>(M - 3) + LO>(y[M - 3], a, nx); nx[M - 3] = y[M - 3] * nri[M - 3]; }
  if constexpr (M > 3) { ldl3_bwd_row<M, M - 4, urow<M>(M - 4) + LO>(y[M - 4], a, nx); nx[M - 4] = y[M - 4] * nri[M - 4]; }
  if constexpr (M > 4) { ldl3_bwd_row<M, M - 5, urow<M>(M - 5) + LO>(y[M - 5], a, nx); nx[M - 5] = y[M - 5] * nri[M - 5]; }
  if constexpr (M > 5) { ldl3_bwd_row<M, M - 6, urow<M>(M - 6) + LO>(y[M - 6], a, nx); nx[M - 6] = y[M - 6] * nri[M - 6]; }
}

// LDS slot: up to 5 KB of records (REC_FMAX fields x 32 bytes = 4288), then a 256-byte pad of zeros at the same
// offset in both slots (lane offsets are slot-independent; the slot base is an instruction immediate)
enum { B3_DATA = 5120, B3_ZBYTES = 256,
       // the identity columns the lanes of the velocity block start from: six lanes x three 16-byte pairs 64 bytes apart
       // (the shape of a column read from the record), also at the same offset in both slots
       B3_ID = B3_DATA + B3_ZBYTES, B3_IDBYTES = 6 * 192, B3_SLOT = B3_ID + B3_IDBYTES,
       // transpose scratch of the symmetrisation: [trajectory][row][14] doubles (even stride: a row is six aligned
       // 16-byte pairs), then a dump for the writes of the lanes that hold no matrix column
       B3_TRS = 14, B3_TR = 2 * B3_SLOT, B3_TRBYTES = 4 * 12 * B3_TRS * 8, B3_DUMP = B3_TR + B3_TRBYTES,
       // per-lane constants of the velocity-block rebuild: [lane of the row j][16] doubles, read back every knot (in
       // registers they pushed 14 more values into AGPRs: +0.03 ms)
       B3_KC = B3_DUMP + 12 * B3_TRS * 8, B3_LDS = B3_KC + 16 * 16 * 8 };

// AL: augmented-Lagrangian solve (the records carry the l_uu diagonal; decides the record size with M and GRAV).
// FAST (round 4): the sweep WITHOUT the general path in its kernel.  The common case -- no regularisation left, every pivot
// positive -- is all the kernel can do; a wave that meets anything else (mu != 0 on entry, a non-positive pivot at some knot)
// flags its group of four trajectories in P.k2_redo and leaves, and the full kernel (FAST = false), launched behind it with
// flag bit 2, redoes exactly those groups from the terminal knot: nothing the fast wave wrote survives (gains of the knots it
// got through are overwritten, mu / delta / gradient are written in the epilogue only).  Why two kernels: with the retry loop,
// the max-regularisation exit and its pivoted LU compiled in, the kernel needs 395 unified registers (139 of them AGPRs, each
// use a v_accvgpr copy); without them 254 and no AGPR -- 0.338 -> 0.304 ms at 4096 x 200 on one box (tools/ab_libs.sh;
// profiles/r04_k2_fast_only_ab.txt), and two waves fit a SIMD at batches beyond 4096.  The first sweep of a solve (mu = 1) goes
// to the full kernel directly; a sweep with nothing to redo pays one launch of waves that read a flag and leave.
template <int M, bool GRAV, bool AL, bool FAST = false>
__global__ __launch_bounds__(64) void k_backward3(Params P, int it, int flags) {
  // flags: bit 0 multiple shooting; bit 1 the records come from the fused rollout, whose trajectories are closed
  // (x_{i+1} = f(x_i, u_i)): the defect field is not written there and reads as zero here
  // bit 2: fallback behind the FAST kernel: only the groups of four that kernel flagged
  if ((flags & 4) && !P.k2_redo[blockIdx.x]) return;
  const int ms = flags & 1;
  const bool closed = (flags & 2) != 0;
  const DConsts& C = *(const DConsts*)P.c;
  const int lane = threadIdx.x, g = lane >> 4, j = lane & 15;
  const int b = blockIdx.x * 4 + g;  // Bp is a multiple of 4
  const bool act = P.active[b] != 0;
  if constexpr (FAST) {
    // hand the group back at once if any of its trajectories still carries regularisation, or if its last sweep needed the
    // general path somewhere (P.k2_hint, kept by the full kernel: a group that keeps meeting non-positive pivots -- single
    // shooting with small input weights does, sweep after sweep -- must not pay for a fast attempt that dies half way
    // every time; the full kernel clears the hint after a sweep that never left the fast path).  Wave-uniform, and before
    // anything is requested from memory.
    const bool back = __any(act && P.mu[b] != 0.0) || P.k2_hint[blockIdx.x] != 0;
    if (lane == 0) P.k2_redo[blockIdx.x] = back ? 1 : 0;
    if (back) return;
  }
  if (!__any(act)) return;
  const int N = P.N;
  // LDS: two record slots (one knot of this wave's four trajectories each, exactly as it lies in REC), a zeroed pad
  // that stands in for structurally-zero fields
  // (FAST fits two waves on a SIMD by its registers; measured at 8192 x 200 that is SLOWER than one at a time -- 0.80 against
  // 0.66 ms, twice the waves on the same LDS-DMA and DPP paths -- so its LDS request is padded to a quarter of a CU's LDS and
  // a CU takes four workgroups, one per SIMD, as it does for the full kernel through its registers)
  __shared__ __attribute__((aligned(16))) char lds[FAST ? (B3_LDS > 34 * 1024 ? B3_LDS : 34 * 1024) : B3_LDS];
  if (lane < B3_ZBYTES / 8) {
    reinterpret_cast<double*>(lds + B3_DATA)[lane] = 0.0;
    reinterpret_cast<double*>(lds + B3_SLOT + B3_DATA)[lane] = 0.0;
  }
  for (int k = lane; k < B3_IDBYTES / 8; k += 64) {  // identity column c6 = k / 24: row r at pair r / 2 (64 bytes apart), half r % 2
    const int c6 = k / 24, w = k % 24, pr = w / 8, hf = w % 8;
    const double v = (hf < 2 && 2 * pr + hf == c6) ? 1.0 : 0.0;
    reinterpret_cast<double*>(lds + B3_ID)[k] = v;
    reinterpret_cast<double*>(lds + B3_SLOT + B3_ID)[k] = v;
  }
  __builtin_amdgcn_wave_barrier();
  const unsigned lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)lds;

  // ---- lane-dependent constants
  const double m12 = (j < 12) ? 1.0 : 0.0;  // matrix columns
  // input owned by this lane's column (the lane holds column c of Mt), -1 if none
  int mycol = -1;
#pragma unroll
  for (int u = 0; u < M; u++) if (j == urow<M>(u)) mycol = u;
  // b_u = F_u[urow(u)][u] (wave-uniform), and the per-lane constants: 2 W2 column (lanes 6..11), the column of
  // Rt = 2 D^-1 R D^-1 (lanes that hold a column of Mt), the update masks of the factorisation
  double bu[M], ibu[M];
#pragma unroll
  for (int u = 0; u < M; u++) { bu[u] = fu_entry<M>(*P.c, urow<M>(u) - 6, u); ibu[u] = 1.0 / bu[u]; }  // generic pointer: note at DConsts
  double kBW[6], Rt[M], wm[M];
#pragma unroll
  for (int r = 0; r < 6; r++) kBW[r] = (j >= 6 && j < 12) ? 2.0 * C.W2[6 * r + (j - 6)] : 0.0;
  {
    double ibc = 0.0;
#pragma unroll
    for (int u = 0; u < M; u++) if (mycol == u) ibc = ibu[u];
#pragma unroll
    for (int u = 0; u < M; u++) {
      Rt[u] = (mycol >= 0) ? 2.0 * C.R[u * M + (mycol >= 0 ? mycol : 0)] * ibu[u] * ibc : 0.0;
      wm[u] = (mycol > u) ? 1.0 : 0.0;
    }
  }
  // which record fields make up column j of [F_x | d] (rows 0..2, 3..5) and of [l_xx | l_x]
  int fT = REC_D, fM = REC_D + 3;
  bool hT = false, hM = false;
  if (j < 3) { fT = REC_RI + 3 * j; fM = REC_TRI + 3 * j; hT = true; hM = true; }
  else if (j < 6) { fM = REC_RI + 3 * (j - 3); hM = true; }
  else if (j < 9) { fT = REC_JR + 3 * (j - 6); fM = REC_QR + 3 * (j - 6); hT = true; hM = true; }
  else if (j < 12) { fM = REC_JR + 3 * (j - 9); hM = true; }
  else if (j == 12 && !closed) { hT = true; hM = true; }
  const bool vcol = j >= 6 && j < 12;  // columns of the velocity block
  const bool hasB = (j == 12 && !closed), isVec = (j == 12 || j == 13);
  const int fB = REC_D + 6;
  const bool hL = (j < 6 || isVec);
  // LDS byte offsets inside a slot (the slot base is added as an immediate); fields that are structurally zero for
  // this lane point into the zeroed pad.  Loads of pairs use bases on even fields (REC_D, REC_LX, REC_LU).
  const unsigned lg = (unsigned)g * 16u;
  const unsigned ZP = (unsigned)B3_DATA;
  unsigned oT[3], oM[3], oL[6];
#pragma unroll
  for (int r = 0; r < 3; r++) { oT[r] = hT ? lg + FOFF(fT + r) : ZP; oM[r] = hM ? lg + FOFF(fM + r) : ZP; }
#pragma unroll
  for (int r = 0; r < 6; r++) oL[r] = hL ? lg + FOFF((j < 6) ? REC_LXX + sym6(r, j) : REC_LX + r) : ZP;
  // rows 6..11 of the column, three 16-byte pairs: the defect (vector column of an open trajectory), the identity
  // column the velocity block is built on, zeros
  const unsigned oB = hasB ? lg + FOFF(fB) : vcol ? (unsigned)B3_ID + (unsigned)(j - 6) * 192u : ZP;
  // velocity block, column c6 = j - 6 = 3 Cb + cc: I + dt J^-1 (coadjoint([v, w]) J + G) (a22_build; swapped twist of
  // App. C-Q1 included).  With J = blkdiag(diag(a), diag(c)) and S(x)[r][c] = sg(r,c) x_k, k = 3 - r - c, the only
  // non-zero off-diagonal entries of the column sit in rows kA = cc + 1 and kB = cc + 2 (mod 3) of each block row:
  //   block (0,0): dt / a_r sg (a_k w_k - a_cc v_k)      block (0,1): dt / a_r sg (m v_k - c_cc w_k)
  //   block (1,0): dt / c_r sg m v_k                       block (1,1): -dt / c_r sg c_cc v_k
  // SO3 family (no swap, traopt_dynamics.py:385-400): block (0,0) dt / a_r sg (a_k - a_cc) w_k, nothing else.
  unsigned oXA = lg + FOFF(REC_XI), oXB = oXA;  // the (w_k, v_k) pairs of k = kA, kB (REC_XI is stored w0 v0 w1 v1 w2 v2)
  {
    double aA[2] = {0, 0}, bA[2] = {0, 0}, aB[2] = {0, 0}, bB[2] = {0, 0}, mA[3] = {0, 0, 0}, mB[3] = {0, 0, 0};
    if (vcol) {
      const Consts& G = *P.c;
      const int Cb = (j - 6) / 3, cc = (j - 6) % 3, kA = (cc + 1) % 3, kB = (cc + 2) % 3;
      auto sg = [](int r, int c) { return ((c - r + 3) % 3 == 1) ? -1.0 : 1.0; };
      const double dt = G.dt, ms = G.mass;
      const double iaA = G.Ibinv[4 * kA], iaB = G.Ibinv[4 * kB], icA = G.Jvinv[4 * kA], icB = G.Jvinv[4 * kB];
      const double a_kA = G.Ib[4 * kA], a_kB = G.Ib[4 * kB], a_cc = G.Ib[4 * cc], c_cc = G.Jv[4 * cc];
      const double sA = sg(kA, cc), sB_ = sg(kB, cc);
      if (so3_family(G.kind)) {
        if (Cb == 0) { aA[0] = dt * iaA * sA * (a_kB - a_cc); aB[0] = dt * iaB * sB_ * (a_kA - a_cc); }
      } else if (Cb == 0) {
        aA[0] = dt * iaA * sA * a_kB; bA[0] = -dt * iaA * sA * a_cc;
        aB[0] = dt * iaB * sB_ * a_kA; bB[0] = -dt * iaB * sB_ * a_cc;
        bA[1] = dt * icA * sA * ms; bB[1] = dt * icB * sB_ * ms;
      } else {
        aA[0] = -dt * iaA * sA * c_cc; bA[0] = dt * iaA * sA * ms;
        aB[0] = -dt * iaB * sB_ * c_cc; bB[0] = dt * iaB * sB_ * ms;
        bA[1] = -dt * icA * sA * c_cc; bB[1] = -dt * icB * sB_ * c_cc;
      }
      mA[kA] = 1.0; mB[kB] = 1.0;
      oXA = lg + FOFF(REC_XI + 2 * kA); oXB = lg + FOFF(REC_XI + 2 * kB);
    }
    if (g == 0) {  // one row of the table per lane index j: the four trajectories of the wave share it
      double* kc = reinterpret_cast<double*>(lds + B3_KC) + 16 * j;
      kc[0] = aA[0]; kc[1] = bA[0]; kc[2] = aB[0]; kc[3] = bB[0]; kc[4] = bA[1]; kc[5] = bB[1];
#pragma unroll
      for (int r = 0; r < 3; r++) { kc[6 + 2 * r] = mA[r]; kc[7 + 2 * r] = mB[r]; }
    }
  }
  const unsigned oKC = (unsigned)B3_KC + 128u * (unsigned)j;
  const unsigned oV = isVec ? lg + FOFF(REC_LX + 6) : ZP;        // l_x[6:12] (vector columns)
  const unsigned oU = isVec ? lg + FOFF(REC_LU) : ZP;            // l_u (vector columns)
  const unsigned oG = lg + FOFF(REC_LU + M);                     // gravity direction (every lane), GRAV only
  constexpr bool al = AL;
  constexpr unsigned blockBytes = (unsigned)rec_fields(M, GRAV, AL, false) * 32u;  // one knot of this wave's four trajectories
  constexpr int NKB = (int)((blockBytes + 1023u) / 1024u);                        // LDS-DMA instructions per knot (3 or 4)
  static_assert(NKB == 3 || NKB == 4, "record block of four trajectories: 3 or 4 KB");
  const unsigned oUU = (al && mycol >= 0) ? lg + FOFF(P.fLUU + (mycol >= 0 ? mycol : 0)) : ZP;  // l_uu^AL[c][c] in the lane of Mt's column c
  // symmetrisation scratch: lane j < 12 writes its column (entry r at row r) and reads row j back as six pairs;
  // the vector lanes write into the dump, read zeros and scale by 1 instead of 1/2
  const unsigned wTR = (j < 12) ? (unsigned)B3_TR + ((unsigned)g * 12u * B3_TRS + (unsigned)j) * 8u : (unsigned)B3_DUMP;
  const unsigned rTR = (j < 12) ? (unsigned)B3_TR + ((unsigned)g * 12u * B3_TRS + (unsigned)j * B3_TRS) * 8u : ZP;
  const double hsym = (j < 12) ? 0.5 : 1.0;
  const unsigned sB = (unsigned)P.Bp * 8u;
  const unsigned vr = REC_VR(b);
  const unsigned vG = GK_VG(b, M) + GOFF(0, (j < 13 ? j : 12), M);
  const size_t recStride = (size_t)P.recF * P.Bp, gStride = (size_t)13 * M * P.Bp;
  double Cg[GRAV ? 3 : 1][6];
  if constexpr (GRAV) {
#pragma unroll
    for (int a = 0; a < 3; a++)
#pragma unroll
      for (int r = 0; r < 6; r++) Cg[a][r] = (j < 3) ? C.Llin[a][6 * r + (j < 3 ? j : 0)] : 0.0;
  }
  double mu = P.mu[b], delta = P.delta[b];
  int warned = 0;

  // one knot of records into LDS slot s (wave-uniform source, 16 bytes per lane and instruction)
  auto dma_from = [&](const char* src, int s) {
    const unsigned dst = (unsigned)__builtin_amdgcn_readfirstlane((int)(lds0 + (unsigned)s * B3_SLOT));
    if constexpr (NKB == 3) rl_dma16x3_rec(uniform_ptr(src), (unsigned)lane * 16u, dst);
    else rl_dma16x4_rec(uniform_ptr(src), (unsigned)lane * 16u, dst);
  };
  auto dma_knot = [&](int i, int s) {
    dma_from(reinterpret_cast<const char*>(P.REC + recStride * i) + (size_t)blockIdx.x * blockBytes, s);
  };
  // running addresses of the knot loop (one 64-bit subtraction per knot instead of the multiplications of the indexed
  // form -- the wave issues its scalar instructions in line with the vector ones): the records of knot i - 2 and the
  // gains of knot i + 1 at step i
  const char* rec_run = reinterpret_cast<const char*>(P.REC + recStride * (size_t)(N > 2 ? N - 2 : 0)) + (size_t)blockIdx.x * blockBytes;
  const double* gk_run = P.GK + gStride * (size_t)(N + 1);
  const size_t recStrideB = recStride * 8;

  // terminal condition: V = [l_xx(N) | l_x(N)] with P weights (traopt_controller.py:2956-2957)
  double V[12];
  {
    __amdgpu_buffer_rsrc_t rR = mkbuf(P.REC + recStride * N, (unsigned)P.recF * sB);
    const unsigned OOB = 0x40000000u;
#pragma unroll
    for (int r = 0; r < 6; r++) {
      const int fl = (j < 6) ? REC_LXX + sym6(r, j) : REC_LX + r;
      double t1 = bld(rR, hL ? vr + FOFF(fl) : OOB, 0), t2 = bld(rR, isVec ? vr + FOFF(REC_LX + 6 + r) : OOB, 0);
      V[r] = t1;
      double p2 = (j >= 6 && j < 12) ? 2.0 * C.P2[6 * r + (j - 6)] : 0.0;
      V[6 + r] = t2 + p2;
    }
  }
  double gsum = 0;
  double Kst[M];
#pragma unroll
  for (int u = 0; u < M; u++) Kst[u] = 0;
  auto store_gains = [&](const double* gk) {
    if (act && j < 13) {
      __amdgpu_buffer_rsrc_t rGs = mkbuf(gk, 13 * M * sB);
#pragma unroll
      for (int u = 0; u < M; u += 2) bst2_gk(rGs, vG, GOFF(u, 0, M), Kst[u], Kst[u + 1]);
    }
  };
#ifdef TOLG_STAMPS
  unsigned long long st_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0}, st_t = __builtin_amdgcn_s_memtime();
  const unsigned long long st_rt0 = __builtin_amdgcn_s_memrealtime(), st_ct0 = st_t;
#endif

  bool failed = false;  // (FAST) a knot the fast path could not settle: wave-uniform
  // ---- one knot.  SLOT (compile time): the LDS slot that holds knot i; the loop below is unrolled by two.
  auto step = [&](int i, auto slot_tag) {
    constexpr int SLOT = decltype(slot_tag)::value;
    const char* sl = lds + SLOT * B3_SLOT;
    auto ld = [&](unsigned off) -> double { return *reinterpret_cast<const double*>(sl + off); };
    auto ld2 = [&](unsigned off, int k, double& x0, double& x1) {  // pair k of a run that starts on an even field
      const f64x2 w = *reinterpret_cast<const f64x2*>(sl + off + k * 64);
      x0 = w.x; x1 = w.y;
    };
    // The records of knot i were requested two steps ago; the memory queue retires in order, so "everything but
    // the last step's gain stores and record request" is a counted wait.  (Step 0 is preceded by a step that
    // requested nothing.)
    if (i == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(%0)" :: "n"(M / 2 + NKB) : "memory");
    // (One VALU instruction stands in front of most of these reads: the lane's address comes back from an AGPR.  With
    // absolute addresses and the slot as an instruction immediate the v_add of the slot-relative form went, the AGPR
    // read stayed: no gain, not kept.)
    // (Fencing these reads into use order -- the six that open the Z product first, [l_xx | l_x] and l_u behind its first
    // block -- was measured: +0.02 ms.  The compiler's order stays.)
    double A[12], Qh[12], lu[M], luu_i = 0.0;
    // (Reading the six fields that open the Z product at the end of the previous step -- they are in the other slot by
    // then -- was measured late in round 3: 0.336 against 0.337 ms, nothing.  Not kept.)
#pragma unroll
    for (int r = 0; r < 3; r++) { A[r] = ld(oT[r]); A[3 + r] = ld(oM[r]); }
#pragma unroll
    for (int r = 0; r < 6; r += 2) ld2(oB, r / 2, A[6 + r], A[7 + r]);
#pragma unroll
    for (int r = 0; r < 6; r++) Qh[r] = ld(oL[r]);
#pragma unroll
    for (int r = 0; r < 6; r += 2) ld2(oV, r / 2, Qh[6 + r], Qh[7 + r]);
#pragma unroll
    for (int a = 0; a < M; a += 2) ld2(oU, a / 2, lu[a], lu[a + 1]);
    if constexpr (al) luu_i = ld(oUU);
    if constexpr (GRAV) {
      double gv[3];
#pragma unroll
      for (int a = 0; a < 3; a++) gv[a] = *reinterpret_cast<const double*>(sl + oG + (FOFF(REC_LU + M + a) - FOFF(REC_LU + M)));
#pragma unroll
      for (int r = 0; r < 6; r++) A[6 + r] += gv[0] * Cg[0][r] + gv[1] * Cg[1][r] + gv[2] * Cg[2][r];
    }
    // twist pairs and lane constants of the velocity-block rebuild (used between the two halves of the Z product)
    const f64x2 xA = *reinterpret_cast<const f64x2*>(sl + oXA), xB = *reinterpret_cast<const f64x2*>(sl + oXB);  // (w, v) of kA, kB
    const f64x2* kc = reinterpret_cast<const f64x2*>(lds + oKC);
    const f64x2 c0 = kc[0], c1 = kc[1], c2 = kc[2];  // aA0 bA0 | aB0 bB0 | bA1 bB1
    const f64x2 mk0 = kc[3], mk1 = kc[4], mk2 = kc[5];  // (row r == kA, row r == kB), r = 0..2
    STAMP(0)
    // ---- Z = V [F_x | d]  (+ V_x in the vector column -> w = V_x + V_xx d; the adjoint passes through)
    double Z[12];
#pragma unroll
    for (int r = 0; r < 12; r++) Z[r] = (1.0 - m12) * V[r];
    rank1_bk3_0(Z, V, A[0], A[1], A[2]); rank1_bk3_3_nn(Z, V, A[3], A[4], A[5]);
    {  // velocity block of F_x on top of its identity column (lanes 6..11; every constant is zero elsewhere); here, behind
      // the first half of the product, its LDS reads have long returned
      const double eA0 = fma(c0.x, xB.x, c0.y * xB.y), eB0 = fma(c1.x, xA.x, c1.y * xA.y), eA1 = c2.x * xB.y, eB1 = c2.y * xA.y;
      A[6] = fma(mk0.y, eB0, fma(mk0.x, eA0, A[6])); A[9] = fma(mk0.y, eB1, fma(mk0.x, eA1, A[9]));
      A[7] = fma(mk1.y, eB0, fma(mk1.x, eA0, A[7])); A[10] = fma(mk1.y, eB1, fma(mk1.x, eA1, A[10]));
      A[8] = fma(mk2.y, eB0, fma(mk2.x, eA0, A[8])); A[11] = fma(mk2.y, eB1, fma(mk2.x, eA1, A[11]));
    }
    rank1_bk3_6_nn(Z, V, A[6], A[7], A[8]); rank1_bk3_9_nn(Z, V, A[9], A[10], A[11]);
    STAMP(1)
    // the slot is consumed (every ds_read above has returned: Z needed them): last knot's gains go out, then the
    // records of knot i - 2 come into this slot.  Stores first: the wait at the top of a step covers both, in order.
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#ifdef TOLG_STAMPS2  // finer split of this phase: (1) ends behind the wait, (2) is the gain stores alone, the DMA issue counts into (3)
    STAMP(1)
#endif
    gk_run -= gStride;
    if (i < N - 1) store_gains(gk_run);
#ifdef TOLG_STAMPS2
    STAMP(2)
#endif
    rec_run -= recStrideB;
    if (i >= 2) dma_from(rec_run, SLOT);
    __builtin_amdgcn_sched_barrier(0);
#ifndef TOLG_STAMPS2
    STAMP(2)
#endif
    // ---- regularised G = rows S of (V + mu I)[F_x | d] (+ D^-1 l_u in the vector columns), Mt; PD test
    // (traopt_controller.py:2964-2995, :3052-3060)
    bool use_lu = false;
    double mu_used = 0.0;
    auto build = [&](double mu_, double (&Gm)[M], double (&Mt)[M]) {
      const double muA = m12 * mu_;
#pragma unroll
      for (int u = 0; u < M; u++) {
        Gm[u] = fma(lu[u], ibu[u], Z[urow<M>(u)]);
        Mt[u] = V[urow<M>(u)] + Rt[u];
      }
      if (__any(mu_ != 0.0)) {
#pragma unroll
        for (int u = 0; u < M; u++) {
          Gm[u] = fma(muA, A[urow<M>(u)], Gm[u]);
          Mt[u] += (mycol == u) ? mu_ : 0.0;
        }
      }
      if constexpr (al) {
#pragma unroll
        for (int u = 0; u < M; u++) Mt[u] += (mycol == u) ? luu_i * ibu[u] * ibu[u] : 0.0;
      }
    };
    // regularisation schedule (traopt_controller.py:2975-2995); returns true when this knot is settled
    auto schedule = [&](bool pd) -> bool {
      if (!pd) {
        if constexpr (!FAST) failed = true;  // (full kernel: a non-positive pivot in this sweep -> P.k2_hint; per lane, any-reduced at the end)
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
    // ---- Qh = [l_xx | l_x] + F_x^T Z   (F_x = [Ri 0 Jr 0; TRi Ri Qr Jr; A21 0 A22 A22]: zero 3-row blocks skipped)
    // (2 W2's columns from an LDS image read in place of the zeros of the l_x[6:12] loads -- six instructions and six live
    // doubles fewer -- measured late in round 3: 0.331 against 0.327 ms, the allocation it led to was the slower one.)
#pragma unroll
    for (int r = 0; r < 6; r++) Qh[6 + r] += kBW[r];
    bool done = !act;  // (general path) inactive trajectories are settled from the start
    {
      const double a0[3] = {A[0], A[1], A[2]}, z0[3] = {Z[0], Z[1], Z[2]}, a1[3] = {A[3], A[4], A[5]}, z1[3] = {Z[3], Z[4], Z[5]};
      const double a2[6] = {A[6], A[7], A[8], A[9], A[10], A[11]}, z2[6] = {Z[6], Z[7], Z[8], Z[9], Z[10], Z[11]};
      rank1_bi_02x3_nn(Qh, a0, z0);
      rank1_bi_x3_nn(Qh, a1, z1);
      if constexpr (GRAV) rank1_bi_023x6_nn(Qh, a2, z2);
      else rank1_bi_23x6_nn(Qh, a2, z2);
    }
    // Q_xx on its way through LDS for the symmetrisation: written here, read back (transposed) a few hundred cycles
    // later -- behind the factorisation / the gradient term / the forward substitution, not in front of the next knot
    // Every TOLG_K3_SYMP-th knot only (knots i = 0 mod 4 by default: a wave-uniform branch around the transpose in the
    // steps of slot 0): between two symmetrisations the antisymmetric part grows by the factor of that many knots (~1.1 per
    // knot measured over whole sweeps, 4 at worst: 2.6e-14 of |V| before it is removed), which leaves it at rounding
    // level.  Period 2 -> 4 (late round 3): -10 instructions per knot on average, 0.335 -> 0.324 ms with the column tie
    // of ldl3_forward (same box).  Knot 0 is always one of them.
#ifndef TOLG_K3_SYMP
#define TOLG_K3_SYMP 4
#endif
    static_assert(TOLG_K3_SYMP >= 1 && (TOLG_K3_SYMP & (TOLG_K3_SYMP - 1)) == 0, "symmetrisation period: a power of two");
    // The FULL kernel symmetrises at every knot, as the reference does (round 4, end).  The growth bound above was measured on
    // solves that converge; far into a divergence -- |F_x| of 1e3 .. 7e4, |V| of 1e17, an input weight of 1e-6 -- the mode
    // outgrows four knots: V_SS came out indefinite at a knot where the oracle's was positive definite, in two of 700 random
    // problems (seeds 50312 and 50349, profiles/r04c_parity_fuzz_final_tree.txt; period 2 cured one, period 1 both).  The fast
    // kernel keeps the period: such a sweep meets a non-positive pivot, hands its group back, and the full kernel redoes it from
    // the terminal knot with the antisymmetric part removed at every knot.
    constexpr int SYMP = FAST ? TOLG_K3_SYMP : 1;
    const bool sym_now = SYMP == 1 || (SLOT == 0 && (SYMP == 2 || (i & (SYMP - 1)) == 0));
    if (sym_now) {
#pragma unroll
      for (int r = 0; r < 12; r++) *reinterpret_cast<double*>(lds + wTR + r * (B3_TRS * 8)) = Qh[r];
    }
    double T[12];
    auto read_T = [&]() {
#pragma unroll
      for (int k = 0; k < 6; k++) {
        const f64x2 t = *reinterpret_cast<const f64x2*>(lds + rTR + 16 * k);
        T[2 * k] = t.x; T[2 * k + 1] = t.y;
      }
    };
    auto symmetrise = [&]() {
      if (!sym_now) return;
      read_T();
#pragma unroll
      for (int r = 0; r < 12; r++) Qh[r] = hsym * (Qh[r] + T[r]);
    };
    // gradient term: ||Q_u|| = ||D G|| in the MS vector lane, ||l_u + F_u^T p|| in the SS adjoint lane
    auto grad_term = [&](const double (&G)[M]) {
      double s0 = 0, s1 = 0;
#pragma unroll
      for (int u = 0; u < M; u += 2) {
        const double q0 = bu[u] * G[u], q1 = bu[u + 1] * G[u + 1];
        s0 = fma(q0, q0, s0); s1 = fma(q1, q1, s1);
      }
      const double s_ = s0 + s1;
      // sqrt(s) = s rsqrt(s), v_rsq_f64 + one refinement step (4e-15 relative: the gradient norm is compared with 1e-6)
      // (late round 3: the step on the root itself, r' = r + (y / 2)(s - r^2) with r = s y -- four instructions behind the
      // v_rsq_f64 instead of six, the same quadratic convergence)
      const double y = __builtin_amdgcn_rsq(s_);
      const double r_ = s_ * y, h_ = 0.5 * y;
      const double rt = fma(h_, fma(-r_, r_, s_), r_);
      gsum += (s_ > 0.0) ? rt : 0.0;
    };
    // what follows a settled factorisation: forward substitution, V <- sym(Q_xx) - Y^T Dl^-1 Y (== Eq. 11b/11c of
    // traopt_controller.py:2998-3004 for the exact gains), back substitution in place, gains [K | k] = D^-1 nx
    auto finish = [&](double (&Y)[M], const double (&Uf)[M], double (&nri)[M]) {
      grad_term(Y);
      STAMP(4)
      if (!ms) {  // the single-shooting adjoint lane takes no gain correction
#pragma unroll
        for (int u = 0; u < M; u++) nri[u] = (j == 13) ? 0.0 : nri[u];
      }
      double zn[M], nx[M];
      ldl3_forward<M>(Uf, nri, Y, zn);
      symmetrise();
      STAMP(5)
      if constexpr (M == 6) rank1_bi_x6_nn(Qh, Y, zn);
      else rank1_bi_x4_nn(Qh, Y, zn);
      ldl3_backward_nx<M>(Uf, nri, Y, zn, nx);
#pragma unroll
      for (int u = 0; u < M; u++) Kst[u] = ibu[u] * nx[u];
    };
    // The common case -- no regularisation left (mu decays to 0 within the first six knots of a solve and stays there),
    // first attempt positive definite for every trajectory of the wave -- is straight-line code on its own variables:
    // no mu terms, no merge copies.  Everything else (mu != 0, a failed attempt, the max-regularisation exit) takes
    // the general path below, which starts the knot's attempts from scratch.
    bool settled = false;
#ifdef TOLG_K3_NOFAST
    if (false) {
#else
    if (FAST || !__any(act && mu != 0.0)) {
#endif
      double Y[M], Uf[M], nri[M], dv[M];
#pragma unroll
      for (int u = 0; u < M; u++) {
        Y[u] = fma(lu[u], ibu[u], Z[urow<M>(u)]);
        Uf[u] = V[urow<M>(u)] + Rt[u];
        if constexpr (al) Uf[u] += (mycol == u) ? luu_i * ibu[u] * ibu[u] : 0.0;
      }
      ldl3_factor<M>(Uf, nri, dv, wm);
      const bool pd = ldl3_all_positive<M>(dv);
      STAMP(3)
      if (!__any(act && !pd)) {
        if (act) { delta = fmin(1.0, delta) * 0.5; mu = 0.0; }  // schedule(true) with mu == 0 (:2986-2991)
        finish(Y, Uf, nri);
        settled = true;
      }
    }
    if constexpr (FAST) {
      if (!settled) { failed = true; return; }  // a non-positive pivot: the full kernel redoes this group (wave-uniform)
    } else if (!settled) {
      double Y2[M], U2[M], nr2[M], d2[M];
      for (;;) {
        if (!done) {
          mu_used = mu;
          build(mu, Y2, U2);
          ldl3_factor<M>(U2, nr2, d2, wm);
          done = schedule(ldl3_all_positive<M>(d2));
        }
        if (__all(done)) break;
      }
      // every lane: the factorisation of its own settled attempt (a lane that settled early sat out the later rounds)
      build(mu_used, Y2, U2);
      ldl3_factor<M>(U2, nr2, d2, wm);
      if (!__any(use_lu)) {
        finish(Y2, U2, nr2);
      } else {
        // max-regularisation exit with a non-PD Q_uu: np.linalg.solve semantics.  Mt is rebuilt -- the factorisation
        // ran in place -- replicated to every lane and solved by LU with partial pivoting; the value update takes the
        // unfactored form V' = Q_xx - G^T x for those trajectories.
        double Gk[M], Mc[M], Ac[M][M], Xl[M], zn[M], nx[M];
        build(mu_used, Gk, Mc);
        grad_term(Gk);
#pragma unroll
        for (int u = 0; u < M; u++) {
          Xl[u] = Gk[u];
#pragma unroll
          for (int c = 0; c < M; c++) {
            double v = 0;
            if (c == 0) v = bcast<urow<M>(0)>(Mc[u]);
            if (c == 1) v = bcast<urow<M>(1)>(Mc[u]);
            if (c == 2) v = bcast<urow<M>(2)>(Mc[u]);
            if (c == 3) v = bcast<urow<M>(3)>(Mc[u]);
            if constexpr (M > 4) {
              if (c == 4) v = bcast<urow<M>(4)>(Mc[u]);
              if (c == 5) v = bcast<urow<M>(5)>(Mc[u]);
            }
            Ac[u][c] = v;
          }
        }
        lu_solve<M>(Ac, Xl);
#pragma unroll
        for (int u = 0; u < M; u++) nr2[u] = (j == 13) ? 0.0 : nr2[u];
        ldl3_forward<M>(U2, nr2, Y2, zn);
        // lanes of a max-regularised trajectory: (Y, zn) <- (G, -x) so that one rank-m update serves both kinds
        double Yu[M];
#pragma unroll
        for (int u = 0; u < M; u++) {
          const double xl = (j == 13) ? 0.0 : Xl[u];
          Yu[u] = use_lu ? Gk[u] : Y2[u];
          zn[u] = use_lu ? -xl : zn[u];
        }
        symmetrise();
        if constexpr (M == 6) rank1_bi_x6(Qh, Yu, zn);
        else rank1_bi_x4(Qh, Yu, zn);
        ldl3_backward_nx<M>(U2, nr2, Y2, zn, nx);
#pragma unroll
        for (int u = 0; u < M; u++) Kst[u] = ibu[u] * (use_lu ? -((j == 13) ? 0.0 : Xl[u]) : nx[u]);
      }
    }
#pragma unroll
    for (int r = 0; r < 12; r++) V[r] = Qh[r];
    STAMP(6)
  };

  // prologue: knots N-1 and N-2 into the two slots
  dma_knot(N - 1, (N - 1) & 1);
  if (N >= 2) dma_knot(N - 2, (N - 2) & 1);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  int i = N - 1;
  if (i & 1) { step(i, std::integral_constant<int, 1>()); i--; }
  for (; i >= 1 && !(FAST && failed); i -= 2) {
    step(i, std::integral_constant<int, 0>());
    if (FAST && failed) break;
    step(i - 1, std::integral_constant<int, 1>());
  }
  if (i == 0 && !(FAST && failed)) step(0, std::integral_constant<int, 0>());
  if constexpr (FAST) {
    if (failed) {
      // every LDS-DMA request of this wave must have landed before its LDS can go to another workgroup
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      if (lane == 0) { P.k2_redo[blockIdx.x] = 1; P.k2_hint[blockIdx.x] = 8; }
      return;
    }
  } else {
    // (the hint counts down: a sweep that met a non-positive pivot keeps the group on the full kernel for the next eight sweeps,
    // each clean one takes one off -- a group that alternates between clean and regularised sweeps would otherwise pay for a
    // fast attempt that dies half way every other sweep.  Regularisation that is merely left over on entry and decays -- the
    // first sweep of every solve starts from mu = 1 -- does not count: the fast kernel checks mu itself.)
    const bool anyf = __any(failed);
    // (the first sweep of a solve starts the count: what the workspace held before is not a hint)
    if (lane == 0) { const int hn = (it == 0) ? 0 : P.k2_hint[blockIdx.x]; P.k2_hint[blockIdx.x] = anyf ? 8 : (hn > 0 ? hn - 1 : 0); }
  }
  store_gains(P.GK);
#ifdef TOLG_STAMPS
  STAMP(7)
  if (blockIdx.x == 7 && lane == 0 && P.mu_hist) {
    for (int k = 0; k < 8; k++) P.mu_hist[(size_t)28 * P.max_iter + k] = (double)st_acc[k];
    P.mu_hist[(size_t)29 * P.max_iter + 0] = (double)(__builtin_amdgcn_s_memrealtime() - st_rt0);  // 100 MHz ticks
    P.mu_hist[(size_t)29 * P.max_iter + 1] = (double)(__builtin_amdgcn_s_memtime() - st_ct0);
  }
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
