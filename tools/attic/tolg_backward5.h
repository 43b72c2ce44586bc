// tolg_backward5.h -- K2, fifth form (round 3): the sweep of tolg_backward3.h carried by a PAIR of waves per group of four
// trajectories.  Included by tolg_kernels.hip inside namespace tolg, after tolg_backward3.h (LDL^T helpers, DPP macro),
// ONLY under -DTOLG_K2_V5.
//
// STATUS: the second measured negative result on "two waves per SIMD" (all 104 GPU parity tests pass with it;
// profiles/r03_k5_*).  It does what it was built for -- no duplicated instructions (560 per knot and group against
// k_backward3's 571), wave B's critical path 3 470 cycles per knot against 4 180, both measured at the same 1.97 GHz
// in one run (tools/k5_stamps.py, tools/k2_stamps.py) -- and the LAUNCH is slower: 0.455 ms against 0.343 under
// rocprofv3, 0.40-0.42 against 0.355 in the bench (two-wave workgroups, roles by wave index).  The two waves of a pair
// are unequal (A: 180 vector instructions per knot, B: 350 on the critical path), so which waves share a SIMD matters;
// the form below makes that exact -- four-wave workgroups, roles from the SIMD a wave finds itself on, one A and one B
// per SIMD (tools/hwid_probe.hip) -- and is slower still (0.437 ms): with a busy A wave on its SIMD and two groups on every
// barrier, B's path is 4 010 cycles per knot (LDS round trips at the top and the tail of a step take 43 % of it), as long
// as the single wave's whole knot.  What the second wave takes off B it gives back in LDS latency and shared issue slots,
// and a second launch redoes the handed-back groups.
//
// Why.  k_backward3 is one wave per SIMD issuing 571 vector instructions per knot at 5.8 cycles each; two resident waves
// issue at 4.4-4.5 per SIMD (profiles/r03_valu_issue_microbench.txt).  The half-column form of tolg_backward4.h got two
// waves per SIMD by giving a trajectory two DPP rows of ONE wave -- and lost, because the serial part of a knot (the
// factorisation of Mt, the two substitutions, the gradient term: one DPP row per trajectory, ~170 instructions) then ran
// once per two trajectories instead of once per four.  Here the four trajectories of a group keep one DPP row each in
// BOTH waves of a 128-thread workgroup, and the waves split the ROWS of every column-distributed 12 x 14 object:
//   wave A: rows 0..5  -- half of Z = V [F_x | d], rows 0..5 of Q_xx (which need only its own half of Z: the lower-left
//                         blocks of F_x are zero), half of the rank-m update; it also feeds the record ring (LDS-DMA) and
//                         writes the velocity-block image of the next knot;
//   wave B: rows 6..11 -- the other halves, and the whole serial chain (rows S = 6.. are the rows the input drives:
//                         Mt = V_SS + ..., G = rows S of Z live in its registers), the gains and their stores.
// No instruction is duplicated except the loads of the column of [F_x | d] (both halves of Z multiply by the whole
// column).  What crosses goes through LDS behind two workgroup barriers per knot (s_barrier of two waves; LDS
// operations waited for in front of it, nothing else: __syncthreads would drain the DMA queue):
//   barrier 1 (Z done):     A's half of Z -> B (Q_xx rows 6.. need it); the record slot is consumed -> A refills it;
//   barrier 2 (Y, zn done): B's Y = L^-1 G, zn = -Dl^-1 Y -> A (its rows of V' = Q_xx + Y^T zn); both halves of Q_xx
//                           for the symmetrisation's transpose; B's verdict on the factorisation.
// Like tolg_backward4.h this kernel is the COMMON case only (mu == 0 at the start of the sweep, every Q_uu positive
// definite): a workgroup that meets anything else stops and hands its four trajectories to k_backward3, launched behind
// it (Params::k2_redo).  m = 6 without gravity block and AL terms (SE3 / rigid body / SO3); everything else is
// k_backward3's.
//
// Record ring: four slots of 3 KB (one knot of the group = 94 fields x 32 B, three DMA bursts), requested four steps
// ahead -- a step is ~1 us here, half of k_backward3's.

// acc[i] += p[i]@lane L * q, i = 0..5
template <int L>
TOLG_DEV void w5_cols(double (&acc)[6], const double (&p)[6], double q) {
#ifndef TOLG_DPP_BUILTIN
  asm volatile(DF3("%0", "%6", "%12", "%13") DF3("%1", "%7", "%12", "%13") DF3("%2", "%8", "%12", "%13")
                   DF3("%3", "%9", "%12", "%13") DF3("%4", "%10", "%12", "%13") DF3("%5", "%11", "%12", "%13")
               : "+v"(acc[0]), "+v"(acc[1]), "+v"(acc[2]), "+v"(acc[3]), "+v"(acc[4]), "+v"(acc[5])
               : "v"(p[0]), "v"(p[1]), "v"(p[2]), "v"(p[3]), "v"(p[4]), "v"(p[5]), "v"(q), "n"(L));
#else
#pragma unroll
  for (int i = 0; i < 6; i++) acc[i] += bcast<L>(p[i]) * q;
#endif
}
// acc[R0 + i] += p@lane (LB + R0 + i) * q, i = 0..NR-1 (NR = 3 or 6; LB = 0: wave A, 6: wave B)
template <int LB, int R0, int NR>
TOLG_DEV void w5_rows(double (&acc)[6], double p, double q) {
  static_assert((NR == 3 && (R0 == 0 || R0 == 3)) || (NR == 6 && R0 == 0), "row blocks of 3 or all 6");
#ifndef TOLG_DPP_BUILTIN
  if constexpr (NR == 6)
    asm volatile(DF3("%0", "%6", "%7", "%8") DF3("%1", "%6", "%7", "%9") DF3("%2", "%6", "%7", "%10")
                     DF3("%3", "%6", "%7", "%11") DF3("%4", "%6", "%7", "%12") DF3("%5", "%6", "%7", "%13")
                 : "+v"(acc[0]), "+v"(acc[1]), "+v"(acc[2]), "+v"(acc[3]), "+v"(acc[4]), "+v"(acc[5])
                 : "v"(p), "v"(q), "n"(LB), "n"(LB + 1), "n"(LB + 2), "n"(LB + 3), "n"(LB + 4), "n"(LB + 5));
  else
    asm volatile(DF3("%0", "%3", "%4", "%5") DF3("%1", "%3", "%4", "%6") DF3("%2", "%3", "%4", "%7")
                 : "+v"(acc[R0]), "+v"(acc[R0 + 1]), "+v"(acc[R0 + 2])
                 : "v"(p), "v"(q), "n"(LB + R0), "n"(LB + R0 + 1), "n"(LB + R0 + 2));
#else
#pragma unroll
  for (int i = 0; i < NR; i++) {
    if (R0 + i == 0) acc[0] += bcast<LB + 0>(p) * q;
    if (R0 + i == 1) acc[1] += bcast<LB + 1>(p) * q;
    if (R0 + i == 2) acc[2] += bcast<LB + 2>(p) * q;
    if (R0 + i == 3) acc[3] += bcast<LB + 3>(p) * q;
    if (R0 + i == 4) acc[4] += bcast<LB + 4>(p) * q;
    if (R0 + i == 5) acc[5] += bcast<LB + 5>(p) * q;
  }
#endif
}
// ldl3_factor with the positive-definiteness test folded in (no pivot array kept)
template <int M, int J = 0>
TOLG_DEV void w5_factor_ok(double (&a)[M], double (&nri)[M], const double (&wm)[M], bool& ok) {
  double pre, d;
  ldl3_head<M, J>(a, wm[J], d, pre);
  ok = ok && (d > 0.0);
  double x = __builtin_amdgcn_rcp(-d);
  x = fma(x, fma(d, x, 1.0), x);
  x = fma(x, fma(d, x, 1.0), x);
  nri[J] = x;
  if constexpr (J + 1 < M) {
    ldl3_update<M, J, urow<M>(J)>(a, pre * x);
    w5_factor_ok<M, J + 1>(a, nri, wm, ok);
  }
}
// the two waves of the workgroup meet; their LDS operations have completed (the DMA queue is left alone)
TOLG_DEV void w5_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// LDS of a workgroup.  Slot (relative offsets, the slot base is an instruction immediate): the group's records of one
// knot exactly as they lie in REC (3 KB), a pad of zeros for structurally-zero fields, the constant 2 W2 block that opens
// Q_xx[6:12, 6:12], the image of this knot's velocity block F_x[6:12, 6:12] (48-byte columns: identity + the four
// twist-dependent entries per column that wave A writes a step ahead).
enum { S5_DATA = 3072, S5_ZP = S5_DATA, S5_ZBYTES = 256, S5_KB = S5_ZP + S5_ZBYTES, S5_KBBYTES = 6 * 48,
       S5_VB = S5_KB + S5_KBBYTES, S5_VBBYTES = 4 * 6 * 48, S5_SLOT = S5_VB + S5_VBBYTES, S5_NSLOT = 4,
       S5_XZ = S5_NSLOT * S5_SLOT, S5_XZBYTES = 64 * 48,            // A's half of Z: [lane][6]
       S5_XY = S5_XZ + S5_XZBYTES, S5_XYBYTES = 64 * 96,            // B's Y | zn: [lane][6 | 6]
       S5_TR = S5_XY + S5_XYBYTES, S5_TRBYTES = 4 * 144 * 8,        // symmetrisation: [trajectory][column][row]
       S5_TDUMP = S5_TR + S5_TRBYTES, S5_TDBYTES = 2 * 16 * 48,     // ... writes of the lanes that hold no matrix column
       S5_TZERO = S5_TDUMP + S5_TDBYTES, S5_TZBYTES = 512,          // ... what they read back (zeros)
       S5_BU = S5_TZERO + S5_TZBYTES, S5_IBU = S5_BU + 64,          // b_u, 1 / b_u
       S5_RT = S5_IBU + 64, S5_RTBYTES = 16 * 48,                   // columns of 2 D^-1 R D^-1 by lane index: [j][6]
       S5_FLAG = S5_RT + S5_RTBYTES, S5_LDS = S5_FLAG + 16 };
static_assert(S5_SLOT % 16 == 0 && S5_KB % 16 == 0 && S5_VB % 16 == 0, "16-byte aligned regions");

// Which two waves share a SIMD decides everything here (A issues ~180 vector instructions per knot, B ~350 on the critical
// path): a SIMD must get one of each.  tools/hwid_probe.hip (profiles/r03_hwid_probe.txt): the four waves of a 256-thread
// workgroup always land on the four SIMDs of one CU (cyclic order 0 -> 2 -> 1 -> 3 from a varying start), and with 70 KB
// of LDS per workgroup a CU hosts exactly workgroups b and b + 256 of a 512-workgroup grid.  So a workgroup carries TWO
// groups of four trajectories, a wave reads the SIMD it runs on from HW_ID, and the role follows from that: SIMDs 0, 1
// take the A waves and 2, 3 the B waves in workgroups with (blockIdx / 256) even, the other way round in the odd ones;
// the waves on SIMDs s and s + 2 form a pair.  The claims are checked through LDS before anything else: a workgroup
// whose waves do not form two pairs hands both groups to k_backward3.
template <int M>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 2))) void k_backward5(Params P, int it, int flags) {
  static_assert(M == 6, "SE3 / rigid body / SO3");
  const int ms = flags & 1;
  const bool closed = (flags & 2) != 0;
  const DConsts& C = *(const DConsts*)P.c;
  const int lane = threadIdx.x & 63;
  const unsigned simd = (__builtin_amdgcn_s_getreg(4 | (31 << 11)) >> 4) & 3u;  // HW_REG_HW_ID: SIMD_ID
  const int par = (int)(blockIdx.x >> 8) & 1;
  const bool isB = ((int)(simd >> 1) ^ par) != 0;
  const int gi = (int)(simd & 1u);              // which of the workgroup's two groups
  const int H = isB ? 1 : 0;
  __shared__ __attribute__((aligned(16))) char lds_all[2 * S5_LDS];
  __shared__ int claim[4];
  if (threadIdx.x < 4) claim[threadIdx.x] = 0;
  w5_barrier();
  if (lane == 0) atomicAdd(&claim[2 * gi + H], 1);
  w5_barrier();
  const bool paired = claim[0] == 1 && claim[1] == 1 && claim[2] == 1 && claim[3] == 1;
  char* lds = lds_all + gi * S5_LDS;
  const int grp = 2 * (int)blockIdx.x + gi;     // group of four trajectories (record / gain group index)
  const int ngrp = P.Bp / 4;
  if (grp >= ngrp) return;                      // (odd number of groups: the last workgroup carries one)
  const int g = lane >> 4, j = lane & 15;
  const int b = grp * 4 + g;
  const bool act = P.active[b] != 0;
  const int N = P.N;
  const unsigned lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)lds;
  double mu = P.mu[b], delta = P.delta[b];
  const int tl = lane + 64 * H;                 // thread index inside the pair
  // ---- LDS constants (both waves take a share; the first barrier below covers them)
  for (int k = tl; k < S5_NSLOT * (S5_SLOT - S5_DATA) / 8; k += 128) {
    const int s = k / ((S5_SLOT - S5_DATA) / 8), o = S5_DATA + 8 * (k % ((S5_SLOT - S5_DATA) / 8));
    double v = 0.0;
    if (o >= S5_KB && o < S5_VB) { const int e = (o - S5_KB) / 8, c6 = e / 6, r6 = e % 6; v = 2.0 * C.W2[6 * r6 + c6]; }  // [c6][r6]
    if (o >= S5_VB) { const int e = (o - S5_VB) / 8, c6 = (e / 6) % 6, r6 = e % 6; v = (c6 == r6) ? 1.0 : 0.0; }          // [g][c6][r6]
    *reinterpret_cast<double*>(lds + s * S5_SLOT + o) = v;
  }
  if (tl < 64) reinterpret_cast<double*>(lds + S5_TZERO)[tl] = 0.0;
  if (tl < 8) {
    const int u = tl < M ? tl : 0;
    const double bq = (tl < M) ? fu_entry<M>(*P.c, urow<M>(u) - 6, u) : 1.0;  // generic pointer: note at DConsts
    reinterpret_cast<double*>(lds + S5_BU)[tl] = bq;
    reinterpret_cast<double*>(lds + S5_IBU)[tl] = 1.0 / bq;
  }
  if (tl == 0) *reinterpret_cast<int*>(lds + S5_FLAG) = 0;
  if (tl < 16) {  // column of 2 D^-1 R D^-1 that the lane of column jj holds (zero where it holds none)
    const int jj = tl;
    int mc = -1;
#pragma unroll
    for (int u = 0; u < M; u++) if (jj == urow<M>(u)) mc = u;
    double ibu0[M], ibc = 0.0;
#pragma unroll
    for (int u = 0; u < M; u++) ibu0[u] = 1.0 / fu_entry<M>(*P.c, urow<M>(u) - 6, u);
#pragma unroll
    for (int u = 0; u < M; u++) if (mc == u) ibc = ibu0[u];
#pragma unroll
    for (int u = 0; u < M; u++)
      reinterpret_cast<double*>(lds + S5_RT)[jj * 6 + u] = (mc >= 0) ? 2.0 * C.R[u * M + (mc >= 0 ? mc : 0)] * ibu0[u] * ibc : 0.0;
  }

  // ---- lane-dependent constants (lane j = column j: 0..11 matrix columns, 12 the vector column, 13 the SS adjoint)
  const double m12 = (j < 12) ? 1.0 : 0.0;
  const bool isVec = (j == 12 || j == 13), hasD = (j == 12 && !closed);
  const bool vcol = j >= 6 && j < 12;
  const unsigned lg = (unsigned)g * 16u;
  const unsigned ZP = (unsigned)S5_ZP;
  auto fx_off = [&](int k) -> unsigned {  // field of [F_x | d][k][j], or where it is structurally 0 / in the image
    if (j == 12) return hasD ? lg + FOFF(REC_D + k) : ZP;
    if (j > 12) return ZP;
    const int c = j;
    if (k < 3) {
      if (c < 3) return lg + FOFF(REC_RI + 3 * c + k);
      if (c >= 6 && c < 9) return lg + FOFF(REC_JR + 3 * (c - 6) + k);
      return ZP;
    }
    if (k < 6) {
      if (c < 3) return lg + FOFF(REC_TRI + 3 * c + (k - 3));
      if (c < 6) return lg + FOFF(REC_RI + 3 * (c - 3) + (k - 3));
      if (c < 9) return lg + FOFF(REC_QR + 3 * (c - 6) + (k - 3));
      return lg + FOFF(REC_JR + 3 * (c - 9) + (k - 3));
    }
    return (c >= 6) ? (unsigned)S5_VB + (unsigned)((g * 6 + (c - 6)) * 6 + (k - 6)) * 8u : ZP;
  };
  unsigned oA[12], oL[6];
#pragma unroll
  for (int k = 0; k < 12; k++) oA[k] = fx_off(k);
#pragma unroll
  for (int r = 0; r < 6; r++) {
    const int gr = 6 * H + r;  // global row
    oL[r] = isVec ? lg + FOFF(REC_LX + gr) : (j < 6 && gr < 6) ? lg + FOFF(REC_LXX + sym6(gr, j))
            : (vcol && gr >= 6) ? (unsigned)S5_KB + (unsigned)((j - 6) * 6 + (gr - 6)) * 8u : ZP;  // 2 W2 opens rows 6.. of columns 6..
  }
  const unsigned oU = (isVec && isB) ? lg + FOFF(REC_LU) : ZP;  // l_u: vector columns (G lives in wave B)
  const unsigned xz = (unsigned)S5_XZ + (unsigned)lane * 48u, xy = (unsigned)S5_XY + (unsigned)lane * 96u;
  const unsigned wTR = (j < 12) ? (unsigned)S5_TR + ((unsigned)g * 144u + (unsigned)j * 12u + 6u * (unsigned)H) * 8u
                                : (unsigned)S5_TDUMP + (unsigned)(H * 16 + g * 4 + (j - 12)) * 48u;
  const unsigned rTR = (j < 12) ? (unsigned)S5_TR + ((unsigned)g * 144u + 6u * (unsigned)H * 12u + (unsigned)j) * 8u : (unsigned)S5_TZERO;
  const double hsym2 = (j < 12) ? 0.5 : 0.0;  // weight of (Q^T - Q) in the symmetrisation (vector lanes: none)
  const unsigned sB = (unsigned)P.Bp * 8u;
  const unsigned vr = REC_VR(b);
  const size_t recStride = (size_t)P.recF * P.Bp, gStride = (size_t)13 * M * P.Bp;
  constexpr unsigned blockBytes = (unsigned)rec_fields(M, false, false, false) * 32u;
  static_assert(blockBytes <= S5_DATA, "three DMA bursts per knot");

  // terminal condition: V = [l_xx(N) | l_x(N)] with P weights (traopt_controller.py:2956-2957), rows 6H..
  double V[6];
  {
    __amdgpu_buffer_rsrc_t rR = mkbuf(P.REC + recStride * N, (unsigned)P.recF * sB);
    const unsigned OOB = 0x40000000u;
#pragma unroll
    for (int r = 0; r < 6; r++) {
      const int gr = 6 * H + r;
      const bool has = isVec || (j < 6 && gr < 6);
      const int fl = isVec ? REC_LX + gr : REC_LXX + sym6(gr < 6 ? gr : 0, j < 6 ? j : 0);
      const double t1 = bld(rR, has ? vr + FOFF(fl) : OOB, 0);
      const double p2 = (vcol && gr >= 6) ? 2.0 * C.P2[6 * (gr - 6) + (j - 6)] : 0.0;
      V[r] = t1 + p2;
    }
  }
  auto of_flag = [&]() -> bool { return *reinterpret_cast<const int*>(lds + S5_FLAG) != 0; };  // (read behind a barrier statement, which is a compiler memory fence)
  bool stopped = __any(act && mu != 0.0) || !paired;  // regularised knots ahead (the first sweep of a solve): k_backward3
  // (the same four trajectories in both waves: the same verdict, so the barriers below stay matched)
  const bool run = __any(act) && !stopped && N >= S5_NSLOT;
  if (N < S5_NSLOT) stopped = true;
  double gsum = 0;

  if (run && !isB) {
    // =============================================== wave A: rows 0..5, record ring, velocity-block image
    // velocity block, column c6 = j - 6 = 3 Cb + cc (derivation in tolg_backward3.h): entries alpha w_k + beta v_k in rows
    // kA = cc + 1, kB = cc + 2 (mod 3) of each 3-row block
    unsigned oXA = ZP, oXB = ZP, oWA = 0, oWB = 0;
    double c_aA0 = 0, c_bA0 = 0, c_aB0 = 0, c_bB0 = 0, c_bA1 = 0, c_bB1 = 0;
    if (vcol) {
      const Consts& G = *P.c;
      const int Cb = (j - 6) / 3, cc = (j - 6) % 3, kA = (cc + 1) % 3, kB = (cc + 2) % 3;
      auto sg = [](int r, int c) { return ((c - r + 3) % 3 == 1) ? -1.0 : 1.0; };
      const double dt = G.dt, mass = G.mass;
      const double iaA = G.Ibinv[4 * kA], iaB = G.Ibinv[4 * kB], icA = G.Jvinv[4 * kA], icB = G.Jvinv[4 * kB];
      const double a_kA = G.Ib[4 * kA], a_kB = G.Ib[4 * kB], a_cc = G.Ib[4 * cc], c_cc = G.Jv[4 * cc];
      const double sA = sg(kA, cc), sB_ = sg(kB, cc);
      if (so3_family(G.kind)) {
        if (Cb == 0) { c_aA0 = dt * iaA * sA * (a_kB - a_cc); c_aB0 = dt * iaB * sB_ * (a_kA - a_cc); }
      } else if (Cb == 0) {
        c_aA0 = dt * iaA * sA * a_kB; c_bA0 = -dt * iaA * sA * a_cc;
        c_aB0 = dt * iaB * sB_ * a_kA; c_bB0 = -dt * iaB * sB_ * a_cc;
        c_bA1 = dt * icA * sA * mass; c_bB1 = dt * icB * sB_ * mass;
      } else {
        c_aA0 = -dt * iaA * sA * c_cc; c_bA0 = dt * iaA * sA * mass;
        c_aB0 = -dt * iaB * sB_ * c_cc; c_bB0 = dt * iaB * sB_ * mass;
        c_bA1 = -dt * icA * sA * c_cc; c_bB1 = -dt * icB * sB_ * c_cc;
      }
      oXA = lg + FOFF(REC_XI + 2 * kA); oXB = lg + FOFF(REC_XI + 2 * kB);
      const unsigned img = (unsigned)S5_VB + (unsigned)((g * 6 + (j - 6)) * 6) * 8u;
      oWA = img + (unsigned)kA * 8u; oWB = img + (unsigned)kB * 8u;  // rows 6 + k (and 9 + k at +24)
    }
    auto velocity_block = [&](int s) {  // the image of the knot that lies in slot s (its records have landed)
      if (!vcol) return;
      char* sn = lds + s * S5_SLOT;
      const f64x2 xA = *reinterpret_cast<const f64x2*>(sn + oXA), xB = *reinterpret_cast<const f64x2*>(sn + oXB);  // (w, v) of kA, kB
      const double eA0 = fma(c_aA0, xB.x, c_bA0 * xB.y), eB0 = fma(c_aB0, xA.x, c_bB0 * xA.y), eA1 = c_bA1 * xB.y, eB1 = c_bB1 * xA.y;
      *reinterpret_cast<double*>(sn + oWA) = eA0; *reinterpret_cast<double*>(sn + oWB) = eB0;
      *reinterpret_cast<double*>(sn + oWA + 24) = eA1; *reinterpret_cast<double*>(sn + oWB + 24) = eB1;
    };
    auto dma_knot = [&](int i) {
      const char* src = reinterpret_cast<const char*>(P.REC + recStride * i) + (size_t)grp * blockBytes;
      const unsigned dst = (unsigned)__builtin_amdgcn_readfirstlane((int)(lds0 + (unsigned)(i & 3) * S5_SLOT));
      rl_dma16x3(uniform_ptr(src), (unsigned)lane * 16u, dst);
    };
    // The LDS reads of a knot's column of [F_x | d] -- what the Z product opens with -- are issued a step ahead, behind barrier 2 of the previous
    // step (slot and image of the next knot are complete by then), into the register set the next step computes on: two
    // sets, chosen by the parity of the slot at compile time -- no copies.
    double RA[2][12];
    auto loadA = [&](auto slot_tag) {
      constexpr int S = decltype(slot_tag)::value;
      const char* sl = lds + S * S5_SLOT;
#pragma unroll
      for (int k = 0; k < 12; k++) RA[S & 1][k] = *reinterpret_cast<const double*>(sl + oA[k]);
    };
    auto stepA = [&](int i, auto slot_tag) -> bool {
      constexpr int SLOT = decltype(slot_tag)::value;
      const char* sl = lds + SLOT * S5_SLOT;
      double (&A)[12] = RA[SLOT & 1];
      double Qh[6];
#pragma unroll
      for (int r = 0; r < 6; r++) Qh[r] = *reinterpret_cast<const double*>(sl + oL[r]);  // (back before the Z product ends)
      double Z[6];
#pragma unroll
      for (int r = 0; r < 6; r++) Z[r] = (1.0 - m12) * V[r];
      w5_cols<0>(Z, V, A[0]); w5_cols<1>(Z, V, A[1]); w5_cols<2>(Z, V, A[2]); w5_cols<3>(Z, V, A[3]);
      w5_cols<4>(Z, V, A[4]); w5_cols<5>(Z, V, A[5]); w5_cols<6>(Z, V, A[6]); w5_cols<7>(Z, V, A[7]);
      w5_cols<8>(Z, V, A[8]); w5_cols<9>(Z, V, A[9]); w5_cols<10>(Z, V, A[10]); w5_cols<11>(Z, V, A[11]);
      {
        f64x2* w = reinterpret_cast<f64x2*>(lds + xz);
        w[0] = f64x2{Z[0], Z[1]}; w[1] = f64x2{Z[2], Z[3]}; w[2] = f64x2{Z[4], Z[5]};
      }
      w5_barrier();  // ---- 1: Z is complete in both waves, the slot is consumed
      // rows 0..5 of Q_xx = l_xx + F_x^T Z need rows 0..5 of Z only (F_x[6:12, 0:6] = 0): columns 0..2 of F_x against rows
      // 0..5, columns 3..5 against rows 3..5
      w5_rows<0, 0, 3>(Qh, A[0], Z[0]); w5_rows<0, 0, 3>(Qh, A[1], Z[1]); w5_rows<0, 0, 3>(Qh, A[2], Z[2]);
      w5_rows<0, 0, 6>(Qh, A[3], Z[3]); w5_rows<0, 0, 6>(Qh, A[4], Z[4]); w5_rows<0, 0, 6>(Qh, A[5], Z[5]);
      constexpr bool SYM = (SLOT & 1) == 0;
      if constexpr (SYM) {
        f64x2* w = reinterpret_cast<f64x2*>(lds + wTR);
        w[0] = f64x2{Qh[0], Qh[1]}; w[1] = f64x2{Qh[2], Qh[3]}; w[2] = f64x2{Qh[4], Qh[5]};
      }
      if (i >= S5_NSLOT) dma_knot(i - S5_NSLOT);  // refill the slot (behind the product: no branch target in front of a DPP block)
      // the records of knot i - 1 have landed (everything but the younger requests has): its velocity-block image
      if (i >= 1) {
        if (i >= 4) asm volatile("s_waitcnt vmcnt(9)" ::: "memory");
        else if (i == 3) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
        else if (i == 2) asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        velocity_block((i - 1) & 3);
      }
      w5_barrier();  // ---- 2: Y, zn, the other half of Q_xx and B's verdict are in LDS
      const int flag = *reinterpret_cast<const int*>(lds + S5_FLAG);  // (tested at the end of the step: nothing waits for it)
      double Yx[M], zx[M];
      if (i >= 1) loadA(std::integral_constant<int, (SLOT + 3) & 3>());  // knot i - 1
      {
        const f64x2* r_ = reinterpret_cast<const f64x2*>(lds + xy);
#pragma unroll
        for (int u = 0; u < M; u += 2) {
          const f64x2 a = r_[u / 2], c = r_[3 + u / 2];
          Yx[u] = a.x; Yx[u + 1] = a.y; zx[u] = c.x; zx[u + 1] = c.y;
        }
      }
      // V' = sym(Q_xx) + Y^T zn as (Q_xx + Y^T zn) + (Q_xx^T - Q_xx) / 2: the transposed entries come back from LDS while
      // the update runs
      double T[6], Q0[6];
      if constexpr (SYM) {
#pragma unroll
        for (int r = 0; r < 6; r++) { T[r] = *reinterpret_cast<const double*>(lds + rTR + 96 * r); Q0[r] = Qh[r]; }
      }
      w5_rows<0, 0, 6>(Qh, Yx[0], zx[0]); w5_rows<0, 0, 6>(Qh, Yx[1], zx[1]); w5_rows<0, 0, 6>(Qh, Yx[2], zx[2]);
      w5_rows<0, 0, 6>(Qh, Yx[3], zx[3]); w5_rows<0, 0, 6>(Qh, Yx[4], zx[4]); w5_rows<0, 0, 6>(Qh, Yx[5], zx[5]);
      if constexpr (SYM) {
#pragma unroll
        for (int r = 0; r < 6; r++) Qh[r] = fma(hsym2, T[r] - Q0[r], Qh[r]);
      }
#pragma unroll
      for (int r = 0; r < 6; r++) V[r] = Qh[r];
      return flag != 0;
    };
    // prologue: the last four knots into the ring, the image of the last one
    for (int k = 1; k <= S5_NSLOT; k++) dma_knot(N - k);
    asm volatile("s_waitcnt vmcnt(9)" ::: "memory");
    w5_barrier();  // ---- 0a: the LDS constants are written (the image below goes on top of them)
    velocity_block((N - 1) & 3);
    w5_barrier();  // ---- 0b: slot and image of knot N - 1 are ready
    switch ((N - 1) & 3) {
      case 0: loadA(std::integral_constant<int, 0>()); break;
      case 1: loadA(std::integral_constant<int, 1>()); break;
      case 2: loadA(std::integral_constant<int, 2>()); break;
      default: loadA(std::integral_constant<int, 3>()); break;
    }
    for (int i = N - 1; i >= 0 && !stopped; i--) {
      switch (i & 3) {
        case 0: stopped = stepA(i, std::integral_constant<int, 0>()); break;
        case 1: stopped = stepA(i, std::integral_constant<int, 1>()); break;
        case 2: stopped = stepA(i, std::integral_constant<int, 2>()); break;
        default: stopped = stepA(i, std::integral_constant<int, 3>()); break;
      }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // (no LDS-DMA of this wave in flight past this point)
  } else if (run) {
    // =============================================== wave B: rows 6..11, the factorisation chain, the gains
    int mycol = -1;  // input whose column of Mt sits in this lane
#pragma unroll
    for (int u = 0; u < M; u++) if (j == urow<M>(u)) mycol = u;
    double wm[M];
#pragma unroll
    for (int u = 0; u < M; u++) wm[u] = (mycol > u) ? 1.0 : 0.0;
    const unsigned oRT = (unsigned)S5_RT + (unsigned)j * 48u;
    auto ibu_load = [&](double (&ib)[M]) {
      const f64x2* p = reinterpret_cast<const f64x2*>(lds + S5_IBU);
#pragma unroll
      for (int u = 0; u < M; u += 2) { const f64x2 w = p[u / 2]; ib[u] = w.x; ib[u + 1] = w.y; }
    };
    const unsigned vG = GK_VG(b, M) + GOFF(0, (j < 13 ? j : 12), M);
    const double* gk_run = P.GK + gStride * (size_t)(N + 1);
    double Kst[M];
#pragma unroll
    for (int u = 0; u < M; u++) Kst[u] = 0;
    auto store_gains = [&](const double* gk) {
      if (act && j < 13) {
        __amdgpu_buffer_rsrc_t rGs = mkbuf(gk, 13 * M * sB);
#pragma unroll
        for (int u = 0; u < M; u += 2) bst2(rGs, vG, GOFF(u, 0, M), Kst[u], Kst[u + 1]);
      }
    };
#ifdef TOLG_STAMPS5
    unsigned long long st5[8] = {0, 0, 0, 0, 0, 0, 0, 0}, st5_t = __builtin_amdgcn_s_memtime();
    const unsigned long long rt0 = __builtin_amdgcn_s_memrealtime(), ct0 = st5_t;
#define STAMP5(k) { __builtin_amdgcn_sched_barrier(0); unsigned long long t_ = __builtin_amdgcn_s_memtime(); st5[k] += t_ - st5_t; st5_t = t_; __builtin_amdgcn_sched_barrier(0); }
#else
#define STAMP5(k)
#endif
    double RA[2][12];
    auto loadB = [&](auto slot_tag) {
      constexpr int S = decltype(slot_tag)::value;
      const char* sl = lds + S * S5_SLOT;
#pragma unroll
      for (int k = 0; k < 12; k++) RA[S & 1][k] = *reinterpret_cast<const double*>(sl + oA[k]);
    };
    auto stepB = [&](int i, auto slot_tag) -> bool {
      constexpr int SLOT = decltype(slot_tag)::value;
      const char* sl = lds + SLOT * S5_SLOT;
      double (&A)[12] = RA[SLOT & 1];
      double Qh[6], lu[M], ib[M], bu[M];
#pragma unroll
      for (int r = 0; r < 6; r++) Qh[r] = *reinterpret_cast<const double*>(sl + oL[r]);  // (these are back before the Z product ends)
#pragma unroll
      for (int a = 0; a < M; a += 2) {
        const f64x2 w = *reinterpret_cast<const f64x2*>(sl + oU + (a / 2) * 64);
        lu[a] = w.x; lu[a + 1] = w.y;
      }
      double Rt[M];
      ibu_load(ib);
      {
        const f64x2* bt = reinterpret_cast<const f64x2*>(lds + S5_BU);
        const f64x2* rt = reinterpret_cast<const f64x2*>(lds + oRT);
#pragma unroll
        for (int u = 0; u < M; u += 2) {
          const f64x2 w = bt[u / 2], r2 = rt[u / 2];
          bu[u] = w.x; bu[u + 1] = w.y; Rt[u] = r2.x; Rt[u + 1] = r2.y;
        }
      }
      STAMP5(0)
      double Z[6];
#pragma unroll
      for (int r = 0; r < 6; r++) Z[r] = (1.0 - m12) * V[r];
      w5_cols<0>(Z, V, A[0]); w5_cols<1>(Z, V, A[1]); w5_cols<2>(Z, V, A[2]); w5_cols<3>(Z, V, A[3]);
      w5_cols<4>(Z, V, A[4]); w5_cols<5>(Z, V, A[5]); w5_cols<6>(Z, V, A[6]); w5_cols<7>(Z, V, A[7]);
      w5_cols<8>(Z, V, A[8]); w5_cols<9>(Z, V, A[9]); w5_cols<10>(Z, V, A[10]); w5_cols<11>(Z, V, A[11]);
      STAMP5(1)
      w5_barrier();  // ---- 1
      STAMP5(2)
      gk_run -= gStride;
      if (i < N - 1) store_gains(gk_run);
      double X[6];  // rows 0..5 of this lane's column of Z (wave A's)
      {
        const f64x2* r_ = reinterpret_cast<const f64x2*>(lds + xz);
        const f64x2 x0 = r_[0], x1 = r_[1], x2 = r_[2];
        X[0] = x0.x; X[1] = x0.y; X[2] = x1.x; X[3] = x1.y; X[4] = x2.x; X[5] = x2.y;
      }
      // rows 6..11 of Q_xx: columns 6..11 of F_x against all of Z -- own rows first (the exchange is still on its way) --
      // except rows 0..2 of F_x against its columns 9..11 (zero block)
      w5_rows<6, 0, 6>(Qh, A[6], Z[0]); w5_rows<6, 0, 6>(Qh, A[7], Z[1]); w5_rows<6, 0, 6>(Qh, A[8], Z[2]);
      w5_rows<6, 0, 6>(Qh, A[9], Z[3]); w5_rows<6, 0, 6>(Qh, A[10], Z[4]); w5_rows<6, 0, 6>(Qh, A[11], Z[5]);
      w5_rows<6, 0, 3>(Qh, A[0], X[0]); w5_rows<6, 0, 3>(Qh, A[1], X[1]); w5_rows<6, 0, 3>(Qh, A[2], X[2]);
      w5_rows<6, 0, 6>(Qh, A[3], X[3]); w5_rows<6, 0, 6>(Qh, A[4], X[4]); w5_rows<6, 0, 6>(Qh, A[5], X[5]);
      constexpr bool SYM = (SLOT & 1) == 0;
      if constexpr (SYM) {
        f64x2* w = reinterpret_cast<f64x2*>(lds + wTR);
        w[0] = f64x2{Qh[0], Qh[1]}; w[1] = f64x2{Qh[2], Qh[3]}; w[2] = f64x2{Qh[4], Qh[5]};
      }
      STAMP5(3)
      // ---- G = rows S of Z (+ D^-1 l_u in the vector columns), Mt = V_SS + 2 D^-1 R D^-1; factorisation and PD test
      // (traopt_controller.py:2964-2995 with mu == 0, :3052-3060): rows S = 6 + u are this wave's registers u
      double Y[M], Uf[M], nri[M], zn[M];
#pragma unroll
      for (int u = 0; u < M; u++) {
        Y[u] = fma(lu[u], ib[u], Z[urow<M>(u) - 6]);
        Uf[u] = V[urow<M>(u) - 6] + Rt[u];
      }
      bool ok = true;
      w5_factor_ok<M>(Uf, nri, wm, ok);
      if (__any(act && !ok)) *reinterpret_cast<int*>(lds + S5_FLAG) = 1;  // the regularisation loop is k_backward3's
      STAMP5(4)
      {  // gradient term: ||Q_u|| = ||D G|| (vector lane, MS) / ||l_u + F_u^T p|| (adjoint lane, SS)
        double s0 = 0, s1 = 0;
#pragma unroll
        for (int u = 0; u < M; u += 2) {
          const double q0 = bu[u] * Y[u], q1 = bu[u + 1] * Y[u + 1];
          s0 = fma(q0, q0, s0); s1 = fma(q1, q1, s1);
        }
        const double s_ = s0 + s1;
        double y = __builtin_amdgcn_rsq(s_);
        { const double g_ = s_ * y, h_ = 0.5 * y; y = 2.0 * fma(h_, fma(-h_, g_, 0.5), h_); }
        gsum += (s_ > 0.0) ? s_ * y : 0.0;
      }
      if (!ms) {  // the single-shooting adjoint lane takes no gain correction
#pragma unroll
        for (int u = 0; u < M; u++) nri[u] = (j == 13) ? 0.0 : nri[u];
      }
      ldl3_forward<M>(Uf, nri, Y, zn);
      {
        f64x2* w = reinterpret_cast<f64x2*>(lds + xy);
#pragma unroll
        for (int u = 0; u < M; u += 2) { w[u / 2] = f64x2{Y[u], Y[u + 1]}; w[3 + u / 2] = f64x2{zn[u], zn[u + 1]}; }
      }
      STAMP5(5)
      w5_barrier();  // ---- 2
      STAMP5(6)
      const int flag = *reinterpret_cast<const int*>(lds + S5_FLAG);  // (tested at the end of the step)
      if (i >= 1) loadB(std::integral_constant<int, (SLOT + 3) & 3>());  // knot i - 1: its reads fly while the rest of this step runs
      if (act) delta = fmin(1.0, delta) * 0.5;  // schedule(true) with mu == 0 (:2986-2991)
      double T[6], Q0[6];
      if constexpr (SYM) {
#pragma unroll
        for (int r = 0; r < 6; r++) { T[r] = *reinterpret_cast<const double*>(lds + rTR + 96 * r); Q0[r] = Qh[r]; }
      }
      // V' rows 6.. = sym(Q_xx) + Y^T zn (== Eq. 11b/11c of traopt_controller.py:2998-3004 for the exact gains), in the
      // form (Q_xx + Y^T zn) + (Q_xx^T - Q_xx) / 2; the next knot waits for this, the back substitution (in place on Y)
      // and the gains do not
      w5_rows<6, 0, 6>(Qh, Y[0], zn[0]); w5_rows<6, 0, 6>(Qh, Y[1], zn[1]); w5_rows<6, 0, 6>(Qh, Y[2], zn[2]);
      w5_rows<6, 0, 6>(Qh, Y[3], zn[3]); w5_rows<6, 0, 6>(Qh, Y[4], zn[4]); w5_rows<6, 0, 6>(Qh, Y[5], zn[5]);
      if constexpr (SYM) {
#pragma unroll
        for (int r = 0; r < 6; r++) Qh[r] = fma(hsym2, T[r] - Q0[r], Qh[r]);
      }
#pragma unroll
      for (int r = 0; r < 6; r++) V[r] = Qh[r];
      double nx[M];
      ldl3_backward_nx<M>(Uf, nri, Y, zn, nx);
#pragma unroll
      for (int u = 0; u < M; u++) Kst[u] = ib[u] * nx[u];
      STAMP5(7)
      return flag != 0;
    };
    w5_barrier();  // ---- 0a
    w5_barrier();  // ---- 0b
    switch ((N - 1) & 3) {
      case 0: loadB(std::integral_constant<int, 0>()); break;
      case 1: loadB(std::integral_constant<int, 1>()); break;
      case 2: loadB(std::integral_constant<int, 2>()); break;
      default: loadB(std::integral_constant<int, 3>()); break;
    }
    for (int i = N - 1; i >= 0 && !stopped; i--) {
      switch (i & 3) {
        case 0: stopped = stepB(i, std::integral_constant<int, 0>()); break;
        case 1: stopped = stepB(i, std::integral_constant<int, 1>()); break;
        case 2: stopped = stepB(i, std::integral_constant<int, 2>()); break;
        default: stopped = stepB(i, std::integral_constant<int, 3>()); break;
      }
    }
    if (!stopped) store_gains(P.GK);
#ifdef TOLG_STAMPS5
    if (grp == 7 && lane == 0 && P.mu_hist) {
      for (int k = 0; k < 8; k++) P.mu_hist[(size_t)28 * P.max_iter + k] = (double)st5[k];
      P.mu_hist[(size_t)29 * P.max_iter + 0] = (double)(__builtin_amdgcn_s_memrealtime() - rt0);  // 100 MHz ticks
      P.mu_hist[(size_t)29 * P.max_iter + 1] = (double)(__builtin_amdgcn_s_memtime() - ct0);
    }
#endif
  }
  if (isB && lane == 0) P.k2_redo[grp] = stopped ? 1 : 0;
  if (stopped || !isB) return;  // a stopped group's sweep is k_backward3's: nothing of the epilogue may have happened
  // ---- epilogue: gradient norm, convergence test (traopt_controller.py:2527-2532, :1937-1942) -- wave B
  const double grad = (ms ? bcast<12>(gsum) : bcast<13>(gsum)) / (double)N;
  if (act && j == 0) {
    P.mu[b] = 0.0;
    P.delta[b] = delta;
    P.grad[b] = grad;
    if (it >= 0 && b < P.B) {
      if (P.grad_hist) P.grad_hist[(size_t)b * (P.max_iter + 1) + it] = grad;
      if (P.mu_hist && it < P.max_iter) P.mu_hist[(size_t)b * P.max_iter + it] = 0.0;
    }
    if (it >= 0) {
      bool conv = ms ? (grad < P.tol_grad && P.dn[b] < P.tol_defect) : (grad < P.tol_grad);
      if (conv) { P.conv[b] = 1; P.active[b] = 0; }
    }
  }
}
