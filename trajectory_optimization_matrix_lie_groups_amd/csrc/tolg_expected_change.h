// tolg_expected_change.h -- the merit search's preparation (traopt_controller.py:2550-2557): the linear alpha = 1
// rollout (_rollout(..., rollout="linear"), :2730-2737), _expected_cost_change (:2756-2769) and _update_defect_weight
// (:2774-2788), in the lane map of the backward sweep.  Included by tolg_kernels.hip inside namespace tolg.
//
// Why a second form.  k_expected_change (tolg_kernels.hip) walks the reference's statements: per knot
// e_i = Log(x_i^-1 x^_i), du_i = k_i + K_i e_i, x^_{i+1} = x_{i+1} Exp(F_x e_i + F_u du_i + d_i) -- four lanes per
// trajectory, every input a load behind the previous knot's result: 2.46 ms per call at 4096 x 200, the longest launch
// of a 4.3 ms merit-search iteration (profiles/r03_final_kernel_stats_merit.csv).  But the group operations cancel:
// the deviation the next knot measures is Log(x_{i+1}^-1 x_{i+1} Exp(v)) = v for a rotation part below pi, so the
// rollout is the affine recursion  e_{i+1} = F_x e_i + F_u (k_i + K_i e_i) + d_i,  e_0 = 0, on 12-vectors.  That is a
// row-per-lane mat-vec chain of ~110 vector instructions per knot whose inputs do not depend on the chain at all:
//   * 16 lanes per trajectory, lane r = row r of every product (rows 0..5 pose, 6..11 twist; lanes u < m also own
//     row u of the gains), the operand vector broadcast by DPP row_newbcast; four trajectories per wave, one wave per
//     workgroup -- the shape of k_backward3, whose record / gain layouts make a knot of a wave's four trajectories one
//     contiguous run each;
//   * records and gains of knot i + 3 requested by LDS-DMA (7 instructions) while knot i is computed from a 4-slot
//     LDS ring; the wait is a counted vmcnt (the queue retires in order);
//   * the velocity block I + H dt is applied from the knot's twist (REC_XI) with lane constants, as in k_backward3.
// A trajectory whose rotation deviation reaches 3 rad at some knot (or is not finite) is flagged instead of written;
// k_expected_change, launched behind this kernel with REDO, recomputes exactly those with the reference's group
// operations.  Results agree with that kernel to rounding (Exp / Log round trips removed, sums
// taken per lane over the horizon, then across the lanes).  Models: a constant input matrix -- every reference script except the
// pendulum, which keeps k_expected_change.  DENSE (round 4): inertia blocks that are not diagonal -- the velocity block I + H dt
// is then a field of the record (Params::fA22 >= 0, 36 more doubles: the record run of four trajectories passes 4 KB and takes a
// fifth LDS-DMA chunk) and its rows are read from the ring like the pose rows instead of being rebuilt from the twist.
enum { EC_DEPTH = 4, EC_GKB = 3072 };
template <bool DENSE>
struct EcLds {  // one ring slot: the record run (4 or 5 KB), the gain run, a zero pad; behind the ring the scratch of the final sums
  enum { RECB = DENSE ? 5120 : 4096, ZP = RECB + EC_GKB, SLOT = ZP + 64, SCR = EC_DEPTH * SLOT, LDS = SCR + 4 * 16 * 2 * 8 };
};

// n KB of one contiguous block by LDS-DMA, 16 bytes per lane and instruction; v[k] is the lane's byte offset for
// chunk k, clamped by the caller so that no lane reads past the block (lanes past its end re-read its last 16 bytes).
// Same M0 / wait-state handling as rl_dma16x3.
TOLG_DEV void ec_dma2(const void* sbase, unsigned v0, unsigned v1, unsigned lds_dst) {
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %4\n\ts_nop 2\n\t"
               "global_load_lds_dwordx4 %1, %3" TOLG_POL(TOLG_NT_EC) "\n\tglobal_load_lds_dwordx4 %2, %3 offset:1024" TOLG_POL(TOLG_NT_EC) "\n\t"
               "s_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(v0), "v"(v1), "s"(sbase), "s"(lds_dst) : "memory");
}
TOLG_DEV void ec_dma3(const void* sbase, unsigned v0, unsigned v1, unsigned v2, unsigned lds_dst) {
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %5\n\ts_nop 2\n\t"
               "global_load_lds_dwordx4 %1, %4" TOLG_POL(TOLG_NT_EC) "\n\tglobal_load_lds_dwordx4 %2, %4 offset:1024" TOLG_POL(TOLG_NT_EC) "\n\t"
               "global_load_lds_dwordx4 %3, %4 offset:2048" TOLG_POL(TOLG_NT_EC) "\n\t"
               "s_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(v0), "v"(v1), "v"(v2), "s"(sbase), "s"(lds_dst) : "memory");
}
TOLG_DEV void ec_dma1(const void* sbase, unsigned v0, unsigned lds_dst) {  // (the instruction offset ends at 4095: a fifth KB is
  unsigned keep;                                                           // a request of its own, with its own LDS base)
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 2\n\t"
               "global_load_lds_dwordx4 %1, %2" TOLG_POL(TOLG_NT_EC) "\n\t"
               "s_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(v0), "s"(sbase), "s"(lds_dst) : "memory");
}
TOLG_DEV void ec_dma4(const void* sbase, unsigned v0, unsigned v1, unsigned v2, unsigned v3_, unsigned lds_dst) {
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %6\n\ts_nop 2\n\t"
               "global_load_lds_dwordx4 %1, %5" TOLG_POL(TOLG_NT_EC) "\n\tglobal_load_lds_dwordx4 %2, %5 offset:1024" TOLG_POL(TOLG_NT_EC) "\n\t"
               "global_load_lds_dwordx4 %3, %5 offset:2048" TOLG_POL(TOLG_NT_EC) "\n\tglobal_load_lds_dwordx4 %4, %5 offset:3072" TOLG_POL(TOLG_NT_EC) "\n\t"
               "s_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(v0), "v"(v1), "v"(v2), "v"(v3_), "s"(sbase), "s"(lds_dst) : "memory");
}

// STORE (round 4, rollout = 'linear'): the recursion also WRITES what it computes -- e_i and du_i = k_i + K_i e_i at alpha = 1 --
// to P.ED, [knot][trajectory][32] doubles (e in 0..11, du in 16..16+m; one 512-byte run per wave, array and knot).  The linear
// rollout of the reference (traopt_controller.py:2720-2737 MS, :2065-2071 SS) is this affine recursion with alpha k_i and
// alpha d_i in place of k_i and d_i and e_0 = 0, so e_i(alpha) = alpha e_i(1): EVERY candidate of a line search is
// x_i (+) alpha e_i, u_i + alpha du_i -- one sweep instead of one sequential rollout per step size (k_ls_eval_affine,
// k_affine_commit in tolg_kernels.hip).  Trajectories handed back (a rotation deviation of 3 rad or more somewhere: Log(Exp(v))
// is no longer v there, and the wrapped recursion is not linear in alpha) keep the statement-form rollouts.
// The stores are two global_store_dwordx2 per step, issued by every lane of every wave that runs the loop (lanes without a
// row write zeros into the padding), in asm so that they take a KNOWN place in the in-order memory queue: the counted waits
// on the DMA ring add them up.
// VARB (with DENSE and GRAV: Pendulum3dDyanmics): the input matrix differs from knot to knot -- its 3 x 3 block F_u[6:9, 0:3] is the
// record's REC_BU (which takes REC_TRI's place; the block it displaces is zero for this model) and is read from the ring.
template <int M, bool GRAV, bool STORE = false, bool DENSE = false, bool VARB = false>
__global__ __launch_bounds__(64) void k_expected_change_ring(Params P) {
  static_assert(!VARB || (DENSE && GRAV && M == 6), "VARB: the pendulum's instantiation");
  typedef EcLds<DENSE> L;
  constexpr unsigned EC_RECB = L::RECB, EC_ZP = L::ZP, EC_SLOT = L::SLOT, EC_SCR = L::SCR, EC_LDS = L::LDS;
  const int lane = threadIdx.x, g = lane >> 4, j = lane & 15;
  const int b = blockIdx.x * 4 + g;  // Bp is a multiple of 4
  const bool act = P.active[b] != 0;
  if (!__any(act)) return;
  const Consts& G = *P.c;  // generic pointer, read ahead of the knot loop only (note at DConsts)
  const int N = P.N;
  const bool al = P.al_lb != nullptr;
  __shared__ __attribute__((aligned(16))) char lds[EC_LDS];
  if (lane < 8 * EC_DEPTH) reinterpret_cast<double*>(lds + (lane >> 3) * EC_SLOT + EC_ZP)[lane & 7] = 0.0;
  __builtin_amdgcn_wave_barrier();
  const unsigned lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)lds;

  // ---- what this lane reads from a slot (byte offsets; structurally-zero entries point into the slot's zero pad)
  const unsigned lg = (unsigned)g * 16u, ZP = (unsigned)EC_ZP;
  const int r3 = j % 3;
  unsigned oF[12];  // row j of [F_x[0:6, :]]: lanes 0..2 [Ri 0 Jr 0], lanes 3..5 [TRi Ri Qr Jr] (blocks column-major)
#pragma unroll
  for (int c = 0; c < 12; c++) {
    int f = -1;
    const int cb = c / 3, cc = c % 3;
    if (j < 3) f = (cb == 0) ? REC_RI : (cb == 2) ? REC_JR : -1;
    else if (j < 6) f = (cb == 0) ? (VARB ? -1 : REC_TRI) : (cb == 1) ? REC_RI : (cb == 2) ? REC_QR : REC_JR;
    oF[c] = (f >= 0) ? lg + FOFF(f + 3 * cc + r3) : ZP;
  }
  const unsigned oD = (j < 12) ? lg + FOFF(REC_D + j) : ZP;
  const unsigned oLX = (j < 12) ? lg + FOFF(REC_LX + j) : ZP;
  const unsigned oLU = (j < M) ? lg + FOFF(REC_LU + j) : ZP;
  const unsigned oLUU = (al && j < M) ? lg + FOFF(P.fLUU + (j < M ? j : 0)) : ZP;
  unsigned oLXX[6];
#pragma unroll
  for (int c = 0; c < 6; c++) oLXX[c] = (j < 6) ? lg + FOFF(REC_LXX + sym6(j < 6 ? j : 0, c)) : ZP;
  // gains: row u = j of [K | k] (lanes past the last row read row 0; their du is masked)
  const unsigned oK = (unsigned)EC_RECB + lg + GOFF((j < M ? j : 0), 0, M);
  const double mU = (j < M) ? 1.0 : 0.0;
  // velocity rows: with p = (rr + 1) % 3, n = (rr + 2) % 3 the row's off-diagonal entries multiply a_n, a_p, c_n, c_p
  // (a = e[6:9], c = e[9:12]) with coefficients linear in (w_p, v_p) resp. (w_n, v_n) -- fx_apply's six cross products
  // (coadjoint([v, w]) J + G with the swapped twist of App. C-Q1; SO3 family: (Ib w) x a - w x (Ib a) alone), row by row
  const bool vrow = j >= 6 && j < 12;
  const int rr = vrow ? (j - 6) % 3 : 0, pI = (rr + 1) % 3, nI = (rr + 2) % 3;
  const unsigned oXp = lg + FOFF(REC_XI + 2 * pI), oXn = lg + FOFF(REC_XI + 2 * nI);  // (w_k, v_k) pairs
  // DENSE: row j - 6 of the stored I + H dt (column-major block, a22_build) instead
  unsigned oA[DENSE ? 6 : 1];
  if constexpr (DENSE) {
#pragma unroll
    for (int c = 0; c < 6; c++) oA[c] = vrow ? lg + FOFF((unsigned)P.fA22 + 6u * c + (unsigned)(j - 6)) : ZP;
  }
  double cf[8];  // (alpha, beta) of a_n, a_p, c_n, c_p
  double mN[3], mP[3];
#pragma unroll
  for (int k = 0; k < 3; k++) { mN[k] = (vrow && k == nI) ? 1.0 : 0.0; mP[k] = (vrow && k == pI) ? 1.0 : 0.0; }
  const double mV = vrow ? 1.0 : 0.0;  // identity part of the velocity block
  const int j6 = vrow ? j - 6 : 0;
  {
    // unconditional loads with clamped indices, masks afterwards: one batch of loads instead of a round trip per branch
    const double ibp = G.Ib[4 * pI], ibn = G.Ib[4 * nI], jvp = G.Jv[4 * pI], jvn = G.Jv[4 * nI], ms = G.mass;
    const bool so3 = so3_family(G.kind);
    const double ibi = G.Ibinv[4 * rr], jvi = G.Jvinv[4 * rr], dt = G.dt;
    const double sT = (vrow && j < 9) ? dt * ibi : 0.0;           // rows 6..8
    const double sB = (vrow && j >= 9 && !so3) ? dt * jvi : 0.0;  // rows 9..11
    if (so3) {
      cf[0] = sT * (ibp - ibn); cf[1] = 0.0; cf[2] = sT * (ibp - ibn); cf[3] = 0.0;
      cf[4] = 0.0; cf[5] = 0.0; cf[6] = 0.0; cf[7] = 0.0;
    } else {
      cf[0] = sT * ibp; cf[1] = sB * ms - sT * ibn; cf[2] = -sT * ibn; cf[3] = sT * ibp - sB * ms;
      cf[4] = -sT * jvn; cf[5] = sT * ms - sB * jvn; cf[6] = sT * jvp; cf[7] = sB * jvp - sT * ms;
    }
  }
  // VARB: rows 6..8 take their first three entries from the record (row-major 3 x 3 at REC_BU)
  unsigned oBU[VARB ? 3 : 1];
  if constexpr (VARB) {
#pragma unroll
    for (int k = 0; k < 3; k++) oBU[k] = (vrow && j < 9) ? lg + FOFF(REC_BU + 3 * (j - 6) + k) : ZP;
  }
  double fuc[M], R2r[M], W2r[6], P2r[6];
#pragma unroll
  for (int k = 0; k < M; k++) {
    fuc[k] = (VARB && vrow && j < 9 && k < 3) ? 0.0 : mV * fu_entry<M>(G, j6, k);  // F_u[j][k]
    R2r[k] = mU * 2.0 * G.R[(j < M ? j : 0) * M + k];  // l_uu row
  }
#pragma unroll
  for (int c = 0; c < 6; c++) {
    W2r[c] = mV * 2.0 * G.W2[6 * j6 + c];  // l_xx twist block, row j - 6 (traopt_cost.py:702)
    P2r[c] = mV * 2.0 * G.P2[6 * j6 + c];
  }
  double LL[GRAV ? 3 : 1][3];  // gravity block A21 = sum_a rte_a Llin[a]: row j - 6, columns 0..2
  if constexpr (GRAV) {
#pragma unroll
    for (int a = 0; a < 3; a++)
#pragma unroll
      for (int c = 0; c < 3; c++) LL[a][c] = mV * G.Llin[a][6 * j6 + c];
  }
  const unsigned oRTE = lg + FOFF(REC_LU + M);

  // ---- DMA sources: one knot of this wave's four trajectories is one contiguous run of records and one of gains
  const unsigned recBytes = (unsigned)P.recF * 32u;
  constexpr unsigned gkBytes = 13u * M * 32u;
  const int nkbR = (int)((recBytes + 1023u) / 1024u);
  constexpr int nkbG = (int)((gkBytes + 1023u) / 1024u);
  static_assert(nkbG == 2 || nkbG == 3, "gain block of four trajectories: 2 or 3 KB");
  unsigned vR[5], vG[3];
#pragma unroll
  for (int k = 0; k < 5; k++) {
    const unsigned want = (unsigned)lane * 16u + 1024u * k, last = recBytes - 16u;
    vR[k] = ((want < last) ? want : last) - 1024u * k;  // chunks past the block are never issued (nkbR)
  }
#pragma unroll
  for (int k = 0; k < 3; k++) {
    const unsigned want = (unsigned)lane * 16u + 1024u * k, last = gkBytes - 16u;
    vG[k] = ((want < last) ? want : last) - 1024u * k;
  }
  const size_t recStrideB = (size_t)P.recF * P.Bp * 8, gStrideB = (size_t)13 * M * P.Bp * 8;
  const char* recBase = reinterpret_cast<const char*>(P.REC) + (size_t)blockIdx.x * recBytes;
  const char* gkBase = reinterpret_cast<const char*>(P.GK) + (size_t)blockIdx.x * gkBytes;
  auto dma_knot = [&](int i, int s) {
    const unsigned dst = (unsigned)__builtin_amdgcn_readfirstlane((int)(lds0 + (unsigned)s * EC_SLOT));
    const void* rs = uniform_ptr(recBase + recStrideB * (size_t)i);
    const void* gs = uniform_ptr(gkBase + gStrideB * (size_t)i);
    if constexpr (DENSE) {  // 130..136 fields: 4160..4352 bytes
      ec_dma4(rs, vR[0], vR[1], vR[2], vR[3], dst);
      ec_dma1(rs, vR[4] + 4096u, dst + 4096u);
    }
    else if (nkbR == 4) ec_dma4(rs, vR[0], vR[1], vR[2], vR[3], dst);
    else ec_dma3(rs, vR[0], vR[1], vR[2], dst);
    if constexpr (nkbG == 3) ec_dma3(gs, vG[0], vG[1], vG[2], dst + EC_RECB);
    else ec_dma2(gs, vG[0], vG[1], dst + EC_RECB);
  };
  const int ndma = nkbR + nkbG;  // 5..8 instructions per knot

  // every constant is in its register before the sweep starts: the compiler's bookkeeping of its own loads (the
  // constants come through the generic pointer) must never meet the DMA queue inside the knot loop, where a
  // compiler-placed vmcnt would drain the requests of the knots ahead
#pragma unroll
  for (int c = 0; c < 6; c++) { W2r[c] = pin_v(W2r[c]); P2r[c] = pin_v(P2r[c]); }
#pragma unroll
  for (int k = 0; k < M; k++) { fuc[k] = pin_v(fuc[k]); R2r[k] = pin_v(R2r[k]); }
#pragma unroll
  for (int k = 0; k < 8; k++) cf[k] = pin_v(cf[k]);
  if constexpr (GRAV) {
#pragma unroll
    for (int a = 0; a < 3; a++)
#pragma unroll
      for (int c = 0; c < 3; c++) LL[a][c] = pin_v(LL[a][c]);
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  double e = 0.0, a1 = 0.0, a2 = 0.0;
  bool bad = false;
  // (STORE) this lane's slot of knot 0 in P.ED and the distance to the next knot's
  double* ed_run = STORE ? P.ED + ((size_t)b * 32 + j) : nullptr;
  const size_t edStride = (size_t)P.Bp * 32;
  // e^T l_xx e, this lane's row: pose rows against the record's block, twist rows against 2 W2 (or 2 P2)
  auto quad_state = [&](const double (&Lr)[6], const double (&Wr)[6]) {
    double t = bcast<0>(e) * Lr[0];
    t = fma(bcast<1>(e), Lr[1], t); t = fma(bcast<2>(e), Lr[2], t); t = fma(bcast<3>(e), Lr[3], t);
    t = fma(bcast<4>(e), Lr[4], t); t = fma(bcast<5>(e), Lr[5], t);
    double s = bcast<6>(e) * Wr[0];
    s = fma(bcast<7>(e), Wr[1], s); s = fma(bcast<8>(e), Wr[2], s); s = fma(bcast<9>(e), Wr[3], s);
    s = fma(bcast<10>(e), Wr[4], s); s = fma(bcast<11>(e), Wr[5], s);
    a2 = fma(e, t + s, a2);
  };

  auto step = [&](int i, auto slot_tag) {
    constexpr int SLOT = decltype(slot_tag)::value;
    const char* sl = lds + SLOT * EC_SLOT;
    auto ld = [&](unsigned off) -> double { return *reinterpret_cast<const double*>(sl + off); };
    // knot i was requested EC_DEPTH - 1 steps ago; behind it at most the requests of knots i + 1 .. i + EC_DEPTH - 2
    // (STORE: behind knot i's request also sit the two stores of each of the last EC_DEPTH - 1 steps -- once that many steps
    // have run.  The first EC_DEPTH - 1 steps have fewer stores behind the request and take the count without them, which
    // waits for a little more than it must; with the full count they did not wait for the DMA at all when it was late: a race
    // that one run in some dozens lost, found as a flaky linear-rollout case.)
    constexpr int NST = STORE ? 2 * (EC_DEPTH - 1) : 0;
    if (STORE && i < EC_DEPTH - 1) {
      if (i + EC_DEPTH - 2 <= N - 1) {
        if (ndma == 8) asm volatile("s_waitcnt vmcnt(%0)" :: "n"((EC_DEPTH - 2) * 8) : "memory");
        else if (ndma == 7) asm volatile("s_waitcnt vmcnt(%0)" :: "n"((EC_DEPTH - 2) * 7) : "memory");
        else if (ndma == 6) asm volatile("s_waitcnt vmcnt(%0)" :: "n"((EC_DEPTH - 2) * 6) : "memory");
        else asm volatile("s_waitcnt vmcnt(%0)" :: "n"((EC_DEPTH - 2) * 5) : "memory");
      } else {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      }
    } else if (i + EC_DEPTH - 2 <= N - 1) {
      if (ndma == 8) asm volatile("s_waitcnt vmcnt(%0)" :: "n"((EC_DEPTH - 2) * 8 + NST) : "memory");
      else if (ndma == 7) asm volatile("s_waitcnt vmcnt(%0)" :: "n"((EC_DEPTH - 2) * 7 + NST) : "memory");
      else if (ndma == 6) asm volatile("s_waitcnt vmcnt(%0)" :: "n"((EC_DEPTH - 2) * 6 + NST) : "memory");
      else asm volatile("s_waitcnt vmcnt(%0)" :: "n"((EC_DEPTH - 2) * 5 + NST) : "memory");
    } else {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the last steps: fewer requests behind this knot's than the count assumes
    }
    // the slot of knot i - 1 is free (every read of it has returned): knot i + EC_DEPTH - 1 goes there
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    if (i + EC_DEPTH - 1 <= N - 1) dma_knot(i + EC_DEPTH - 1, (SLOT + EC_DEPTH - 1) % EC_DEPTH);
    double Kr[13], F[12], Lr[6];
#pragma unroll
    for (int c = 0; c < 13; c++) Kr[c] = ld(oK + c * (M / 2) * 64);
#pragma unroll
    for (int c = 0; c < 12; c++) F[c] = ld(oF[c]);
#pragma unroll
    for (int c = 0; c < 6; c++) Lr[c] = ld(oLXX[c]);
    const double d = ld(oD), lx = ld(oLX), lu = ld(oLU), luu = ld(oLUU);
    const f64x2 Xp = *reinterpret_cast<const f64x2*>(sl + oXp), Xn = *reinterpret_cast<const f64x2*>(sl + oXn);
    double Ar[DENSE ? 6 : 1];
    if constexpr (DENSE) {
#pragma unroll
      for (int c = 0; c < 6; c++) Ar[c] = ld(oA[c]);
    }
    if constexpr (GRAV) {
      const double g0 = ld(oRTE), g1 = ld(oRTE + (FOFF(REC_LU + M + 1) - FOFF(REC_LU + M))),
                   g2 = ld(oRTE + (FOFF(REC_LU + M + 2) - FOFF(REC_LU + M)));
#pragma unroll
      for (int c = 0; c < 3; c++) F[c] += g0 * LL[0][c] + g1 * LL[1][c] + g2 * LL[2][c];
    }
    // du = k + K e   (traopt_controller.py:2689-2690, alpha = 1)
    double du = Kr[12];
    du = fma(bcast<0>(e), Kr[0], du); du = fma(bcast<1>(e), Kr[1], du); du = fma(bcast<2>(e), Kr[2], du);
    du = fma(bcast<3>(e), Kr[3], du); du = fma(bcast<4>(e), Kr[4], du); du = fma(bcast<5>(e), Kr[5], du);
    du = fma(bcast<6>(e), Kr[6], du); du = fma(bcast<7>(e), Kr[7], du); du = fma(bcast<8>(e), Kr[8], du);
    du = fma(bcast<9>(e), Kr[9], du); du = fma(bcast<10>(e), Kr[10], du); du = fma(bcast<11>(e), Kr[11], du);
    du *= mU;
    if constexpr (STORE) {
      asm volatile("global_store_dwordx2 %0, %1, off\n\tglobal_store_dwordx2 %0, %2, off offset:128" :: "v"(ed_run), "v"(e), "v"(du) : "memory");
      ed_run += edStride;
    }
    // _expected_cost_change, knot i (:2760-2765; l_ux = 0 for the tracking costs)
    a1 = fma(lx, e, a1);
    a1 = fma(lu, du, a1);
    quad_state(Lr, W2r);
    {
      double t = luu * du;
      t = fma(bcast<0>(du), R2r[0], t); t = fma(bcast<1>(du), R2r[1], t);
      t = fma(bcast<2>(du), R2r[2], t); t = fma(bcast<3>(du), R2r[3], t);
      if constexpr (M == 6) { t = fma(bcast<4>(du), R2r[4], t); t = fma(bcast<5>(du), R2r[5], t); }
      a2 = fma(du, t, a2);
    }
    // e' = F_x e + F_u du + d   (:2730-2737 with Log(Exp(v)) = v)
    double y = DENSE ? 0.0 : mV * e;
    y = fma(bcast<0>(e), F[0], y); y = fma(bcast<1>(e), F[1], y); y = fma(bcast<2>(e), F[2], y);
    y = fma(bcast<3>(e), F[3], y); y = fma(bcast<4>(e), F[4], y); y = fma(bcast<5>(e), F[5], y);
    y = fma(bcast<6>(e), F[6], y); y = fma(bcast<7>(e), F[7], y); y = fma(bcast<8>(e), F[8], y);
    y = fma(bcast<9>(e), F[9], y); y = fma(bcast<10>(e), F[10], y); y = fma(bcast<11>(e), F[11], y);
    if constexpr (DENSE) {
      y = fma(bcast<6>(e), Ar[0], y); y = fma(bcast<7>(e), Ar[1], y); y = fma(bcast<8>(e), Ar[2], y);
      y = fma(bcast<9>(e), Ar[3], y); y = fma(bcast<10>(e), Ar[4], y); y = fma(bcast<11>(e), Ar[5], y);
    } else {
      const double cAn = fma(cf[0], Xp.x, cf[1] * Xp.y), cAp = fma(cf[2], Xn.x, cf[3] * Xn.y);
      const double cCn = fma(cf[4], Xp.x, cf[5] * Xp.y), cCp = fma(cf[6], Xn.x, cf[7] * Xn.y);
      const double sAn = fma(bcast<8>(e), mN[2], fma(bcast<7>(e), mN[1], bcast<6>(e) * mN[0]));
      const double sAp = fma(bcast<8>(e), mP[2], fma(bcast<7>(e), mP[1], bcast<6>(e) * mP[0]));
      const double sCn = fma(bcast<11>(e), mN[2], fma(bcast<10>(e), mN[1], bcast<9>(e) * mN[0]));
      const double sCp = fma(bcast<11>(e), mP[2], fma(bcast<10>(e), mP[1], bcast<9>(e) * mP[0]));
      y = fma(cAn, sAn, y); y = fma(cAp, sAp, y); y = fma(cCn, sCn, y); y = fma(cCp, sCp, y);
    }
    if constexpr (VARB) {
      y = fma(bcast<0>(du), ld(oBU[0]) + fuc[0], y); y = fma(bcast<1>(du), ld(oBU[1]) + fuc[1], y);
      y = fma(bcast<2>(du), ld(oBU[2]) + fuc[2], y);
    } else {
      y = fma(bcast<0>(du), fuc[0], y); y = fma(bcast<1>(du), fuc[1], y);
      y = fma(bcast<2>(du), fuc[2], y);
    }
    y = fma(bcast<3>(du), fuc[3], y);
    if constexpr (M == 6) { y = fma(bcast<4>(du), fuc[4], y); y = fma(bcast<5>(du), fuc[5], y); }
    y += d;
    {  // Log(Exp(v)) = v needs a rotation part below pi: |v_rot| < 3 stays here, anything else is handed back
      const double r0 = bcast<0>(y), r1 = bcast<1>(y), r2 = bcast<2>(y);
      bad = bad || !(fma(r2, r2, fma(r1, r1, r0 * r0)) < 9.0);
    }
    e = y;
  };

  // prologue: knots 0 .. EC_DEPTH - 2
#pragma unroll
  for (int k = 0; k < EC_DEPTH - 1; k++) if (k <= N - 1) dma_knot(k, k);
  int i = 0;
  for (; i + 3 < N; i += 4) {
    step(i, std::integral_constant<int, 0>());
    step(i + 1, std::integral_constant<int, 1>());
    step(i + 2, std::integral_constant<int, 2>());
    step(i + 3, std::integral_constant<int, 3>());
  }
  if (i < N) { step(i, std::integral_constant<int, 0>()); i++; }
  if (i < N) { step(i, std::integral_constant<int, 1>()); i++; }
  if (i < N) { step(i, std::integral_constant<int, 2>()); i++; }
  if constexpr (STORE) asm volatile("global_store_dwordx2 %0, %1, off" :: "v"(ed_run), "v"(e) : "memory");  // e_N
  // terminal knot (:2766-2768): l_x(N) and the pose block of l_xx(N) by plain loads -- issued here, behind the sweep, so
  // that the compiler's own vmcnt bookkeeping never meets the DMA queue inside the knot loop
  {
    double lxN = 0.0, LrN[6];
    __amdgpu_buffer_rsrc_t rR = mkbuf(P.REC + (size_t)P.recF * P.Bp * N, (unsigned)P.recF * (unsigned)P.Bp * 8u);
    const unsigned OOB = 0x40000000u, vr = REC_VR(b);
    lxN = bld(rR, (j < 12) ? vr + FOFF(REC_LX + (j < 12 ? j : 0)) : OOB, 0);
#pragma unroll
    for (int c = 0; c < 6; c++) LrN[c] = bld(rR, (j < 6) ? vr + FOFF(REC_LXX + sym6(j < 6 ? j : 0, c)) : OOB, 0);
    a1 = fma(lxN, e, a1);
    quad_state(LrN, P2r);
  }

  // ---- sums over the lanes of a trajectory (fixed order), defect weight
  double* scr = reinterpret_cast<double*>(lds + EC_SCR) + (size_t)lane * 2;
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  scr[0] = a1; scr[1] = a2;
  __builtin_amdgcn_wave_barrier();
  const unsigned long long bm = __ballot(bad);
  const bool redo = ((bm >> (16 * g)) & 0xffffull) != 0;
  if (j != 0 || !act) return;
  if (P.ec_redo) P.ec_redo[b] = redo ? 1 : 0;
  if (redo) return;
  double c1 = 0.0, c2 = 0.0;
  const double* row = reinterpret_cast<const double*>(lds + EC_SCR) + (size_t)(lane) * 2;
#pragma unroll
  for (int k = 0; k < 12; k++) { c1 += row[2 * k]; c2 += row[2 * k + 1]; }
  P.ecc[2 * b] = c1;
  P.ecc[2 * b + 1] = c2;
  const double dn = P.dn[b], wprev = P.dweight[2 * b + 1];
  double w;
  if (dn < ((so3_family(G.kind)) ? 1e-14 : 1e-12)) w = wprev;  // _defect_kappa (SE3 :2410, SO3 :1090)
  else w = fmax(10.0, 10.0 + fabs(c1 + 0.5 * c2) / ((1.0 - 0.5) * dn));
  P.dweight[2 * b] = w;
  P.dweight[2 * b + 1] = w;
}
