// tolg_backward4.h -- K2, fourth form (round 3): the sweep of tolg_backward3.h with HALF-COLUMN lanes, so that two waves
// are resident per SIMD.  Included by tolg_kernels.hip inside namespace tolg, after tolg_backward3.h (whose LDL^T
// helpers and DPP macro it uses), ONLY under -DTOLG_K2_V4.
//
// STATUS: a measured negative result, kept as the record of the experiment (all 104 GPU parity tests pass with it;
// profiles/r03_k4_*).  4096 x 200 SE3 on MI355X: 0.516 ms per sweep against 0.355 ms for k_backward3.  The products
// halve per wave as planned (72 + 63 + 36 multiply-adds against 144 + 90 + 72), but the factorisation, the two
// substitutions and the gradient term are a serial chain that one DPP row carries for one trajectory: here it runs once
// per TWO trajectories instead of once per four, so a wave issues 376 vector instructions per knot, not half of 571,
// and the two waves of a SIMD together 752 -- 32 % more work than the one wave of k_backward3.  Two resident waves do
// issue denser (a SIMD is busy 68 % of the time against 61 %), not 32 % denser: each wave still parks 27 % of its
// time on s_waitcnt, now mostly LDS round trips between the two rows (57 LDS instructions per knot against 35).
// The batch is 4096: there is no third wave to bring in.  Two resident waves pay only for work that splits without
// duplication, and this sweep's critical chain does not.
//
// Why.  k_backward3 runs one wave per SIMD (4096 trajectories = 1024 waves of four) at ~330 registers, and a lone wave
// issues an independent fp64 multiply-add every 5.5 cycles and a dependent one every 8.9, where two resident waves
// get 4.4 and 4.5 per SIMD (profiles/r03_valu_issue_microbench.txt, occupancy sweep).  The sweep is nothing but such
// instructions.  Here a trajectory takes 32 lanes (two DPP rows), a wave holds two trajectories, a 128-thread
// workgroup the four trajectories of one record group: 2048 waves, <= 256 registers, two per SIMD.
//
// Lane map.  Row h (0 / 1) of a trajectory holds matrix ROWS 6h..6h+5 of every column-distributed 12 x 14 object
// (V, Z, Q_xx ...: six registers per lane instead of twelve).  Within row h, lane j < 12 holds COLUMN (j + 6h) mod 12;
// lanes 12, 13 the vector / adjoint column.  Full-length per-lane vectors (the column of [F_x | d], 12 entries) are
// stored rotated the same way: register m holds entry (m + 6h) mod 12.  With both rotations one instruction stream
// serves both halves, although a DPP row_newbcast selects the same lane n in every row:
//   Z[r'] += V[r']@lane n * A[n]        n = 0..11: lane n holds column k = (n + 6h) mod 12, register n entry k -- the same k
//   Q[r'] += A[m]@lane r' * Zf[m]       lane r' holds column 6h + r', i.e. the global row of accumulator r'
//   V'[r'] += Y[u]@lane r' * zn[u]      likewise
// and the rows S = 6.. that the input drives (F_u = [0; B]) are the registers of row 1, the columns of Mt = V_SS + ...
// its lanes 0..5: the factorisation, both substitutions and the gains live in row 1 (row 0 executes the same
// instructions on values nobody reads).
//
// What crosses between the rows goes through LDS (in-order per wave: no barrier, no flag):
//   - Z: every lane writes its six entries, reads the six of the sibling lane (same column, other row).  Row 1 needs
//     rows 0..5 of Z for Q[6..11] (F_x^T's blocks Jr, Qr); row 0 multiplies what it reads by structural zeros
//     (or by the gravity block, for models that have one);
//   - Y = L^-1 G and zn = -Dl^-1 Y, which row 1 computes: row 0 needs both for its rows of the rank-m update.  Every
//     lane reads the row-1 lane of its column (row 1 its own slot), so no lane-dependent select is needed;
//   - the transpose of the symmetrisation, as before.
// Scope.  This kernel is the COMMON case only: no regularisation left (mu == 0 for every active trajectory of the wave
// when the sweep starts -- true from the second iteration of a solve on) and every factorisation positive definite.
// A wave that meets anything else stops, and its workgroup hands its four trajectories to k_backward3, which is
// launched behind this kernel for the flagged groups (Params::k2_redo; it starts their sweep again from the terminal
// knot: the outputs of this kernel for a flagged group are either overwritten -- gains -- or were never written --
// the epilogue).  The retry loop and the max-regularisation exit raised this kernel's register demand from 188 to
// 300-360 when they were compiled in, i.e. spills into scratch inside the knot loop; where they are they cost one
// near-empty launch per sweep.
// The records are staged as in k_backward3 (2-slot ring, LDS-DMA), but a wave fetches only its two trajectories'
// halves of each 64-byte run (per-lane source addresses; the sibling wave of the workgroup takes the other halves, so
// every line is still fetched once per CU).
#ifndef TOLG_DPP_BUILTIN
// acc[i] += p[i]@lane L * q, i = 0..5
template <int L>
TOLG_DEV void dpp6_cols(double (&acc)[6], const double (&p)[6], double q) {
  asm volatile(DF3("%0", "%6", "%12", "%13") DF3("%1", "%7", "%12", "%13") DF3("%2", "%8", "%12", "%13")
                   DF3("%3", "%9", "%12", "%13") DF3("%4", "%10", "%12", "%13") DF3("%5", "%11", "%12", "%13")
               : "+v"(acc[0]), "+v"(acc[1]), "+v"(acc[2]), "+v"(acc[3]), "+v"(acc[4]), "+v"(acc[5])
               : "v"(p[0]), "v"(p[1]), "v"(p[2]), "v"(p[3]), "v"(p[4]), "v"(p[5]), "v"(q), "n"(L));
}
// acc[R0 + i] += p@lane (R0 + i) * q, i = 0..NR-1 (NR = 3 or 6)
template <int R0, int NR>
TOLG_DEV void dpp_rows(double (&acc)[6], double p, double q) {
  static_assert((NR == 3 && (R0 == 0 || R0 == 3)) || (NR == 6 && R0 == 0), "row blocks of 3 or all 6");
  if constexpr (NR == 6)
    asm volatile(DF3("%0", "%6", "%7", "0") DF3("%1", "%6", "%7", "1") DF3("%2", "%6", "%7", "2")
                     DF3("%3", "%6", "%7", "3") DF3("%4", "%6", "%7", "4") DF3("%5", "%6", "%7", "5")
                 : "+v"(acc[0]), "+v"(acc[1]), "+v"(acc[2]), "+v"(acc[3]), "+v"(acc[4]), "+v"(acc[5]) : "v"(p), "v"(q));
  else if constexpr (R0 == 0)
    asm volatile(DF3("%0", "%3", "%4", "0") DF3("%1", "%3", "%4", "1") DF3("%2", "%3", "%4", "2")
                 : "+v"(acc[0]), "+v"(acc[1]), "+v"(acc[2]) : "v"(p), "v"(q));
  else
    asm volatile(DF3("%0", "%3", "%4", "3") DF3("%1", "%3", "%4", "4") DF3("%2", "%3", "%4", "5")
                 : "+v"(acc[3]), "+v"(acc[4]), "+v"(acc[5]) : "v"(p), "v"(q));
}
#else
template <int L>
TOLG_DEV void dpp6_cols(double (&acc)[6], const double (&p)[6], double q) {
#pragma unroll
  for (int i = 0; i < 6; i++) acc[i] += bcast<L>(p[i]) * q;
}
template <int R0, int NR>
TOLG_DEV void dpp_rows(double (&acc)[6], double p, double q) {
  if constexpr (R0 == 0) { acc[0] += bcast<0>(p) * q; acc[1] += bcast<1>(p) * q; acc[2] += bcast<2>(p) * q; }
  if constexpr (R0 == 3 || NR == 6) { acc[3] += bcast<3>(p) * q; acc[4] += bcast<4>(p) * q; acc[5] += bcast<5>(p) * q; }
}
#endif

// ldl3_factor with the positive-definiteness test folded in (no pivot array kept)
template <int M, int J = 0, int LO = 0>
TOLG_DEV void ldl3_factor_ok(double (&a)[M], double (&nri)[M], const double (&wm)[M], bool& ok) {
  double pre, d;
  ldl3_head<M, J, LO>(a, wm[J], d, pre);
  ok = ok && (d > 0.0);
  double x = __builtin_amdgcn_rcp(-d);
  x = fma(x, fma(d, x, 1.0), x);
  x = fma(x, fma(d, x, 1.0), x);
  nri[J] = x;
  if constexpr (J + 1 < M) {
    ldl3_update<M, J, urow<M>(J) + LO>(a, pre * x);
    ldl3_factor_ok<M, J + 1, LO>(a, nri, wm, ok);
  }
}

// two 1-KB LDS-DMA bursts with per-lane source offsets (the second's instruction offset moves both sides by 1024)
TOLG_DEV void rl_dma16x2v(const void* sbase, unsigned voff0, unsigned voff1, unsigned lds_dst) {
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %4\n\ts_nop 2\n\t"
               "global_load_lds_dwordx4 %1, %3\n\tglobal_load_lds_dwordx4 %2, %3 offset:1024\n\t"
               "s_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(voff0), "v"(voff1), "s"(sbase), "s"(lds_dst) : "memory");
}

// LDS of one wave.  Record slot: the wave's two trajectories of one knot, field f of trajectory t at
// (f >> 1) * 32 + t * 16 + (f & 1) * 8 (2 KB: what two DMA bursts write); a pad of zeros for structurally-zero fields
// (b128 reads of three pairs 32 bytes apart); the constant 1.0; the image of the velocity block F_x[6:12, 6:12] of
// this knot, one 48-byte column per (trajectory, column) -- identity plus the four twist-dependent entries per column
// that a pre-pass writes a step ahead; the constant 2 W2 block that opens Q_xx[6:12, 6:12] (read in the place of the
// zero l_xx entries of those lanes).  Same offsets in both slots: the slot base is an instruction immediate.
enum { B4_DATA = 2048, B4_ZBYTES = 96, B4_ONE = B4_DATA + B4_ZBYTES, B4_VB = B4_ONE + 32, B4_VBBYTES = 12 * 48,
       B4_VDUMP = B4_VB + B4_VBBYTES, B4_VDBYTES = 64,             // pre-pass writes of the lanes without a velocity column
       B4_KB = B4_VDUMP + B4_VDBYTES, B4_KBBYTES = 6 * 48, B4_SLOT = B4_KB + B4_KBBYTES,
       B4_XZ = 2 * B4_SLOT, B4_XZBYTES = 64 * 48,                  // Z exchange: [lane][6]
       B4_XY = B4_XZ + B4_XZBYTES, B4_XYBYTES = 32 * 96,           // Y / zn exchange: [trajectory][row-1 lane][6 | 6]
       B4_TR = B4_XY + B4_XYBYTES, B4_TRBYTES = 2 * 144 * 8,       // symmetrisation: [trajectory][column][row]
       B4_TDUMP = B4_TR + B4_TRBYTES, B4_TDBYTES = 16 * 48,        // ... writes of the vector lanes
       B4_TZERO = B4_TDUMP + B4_TDBYTES, B4_TZBYTES = 512,         // ... what the vector lanes read back (zeros)
       B4_BU = B4_TZERO + B4_TZBYTES, B4_IBU = B4_BU + 64, B4_LDS = B4_IBU + 64 };  // b_u, 1 / b_u (wave-uniform): read where needed
static_assert(B4_SLOT % 16 == 0 && B4_VB % 16 == 0 && B4_KB % 16 == 0, "16-byte aligned regions");
#define FOFF2(f) ((((unsigned)(f)) >> 1) * 32u + (((unsigned)(f)) & 1u) * 8u)

#ifndef K4_WAVES
#define K4_WAVES __attribute__((amdgpu_waves_per_eu(2, 2)))
#endif
#ifndef K4_NUM_VGPR
#define K4_NUM_VGPR
#endif
template <int M, bool GRAV, bool AL>
__global__ __launch_bounds__(128) K4_WAVES K4_NUM_VGPR void k_backward4(Params P, int it, int flags) {
  static_assert(M == 6 && !GRAV && !AL, "first cut: SE3 / rigid body / SO3 without gravity block and AL terms");
  constexpr int LO = -6;  // lane of input u's column of Mt: urow(u) - 6 (row 1)
  const int ms = flags & 1;
  const bool closed = (flags & 2) != 0;
  const DConsts& C = *(const DConsts*)P.c;
  const int wv = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int t = lane >> 5, h = (lane >> 4) & 1, j = lane & 15;
  const int b = blockIdx.x * 4 + 2 * wv + t;  // Bp is a multiple of 4
  const bool act = P.active[b] != 0;
  const int N = P.N;
  __shared__ __attribute__((aligned(16))) char lds_all[2 * B4_LDS];
  __shared__ int wave_stopped[2];
  double mu = P.mu[b], delta = P.delta[b];
  char* lds = lds_all + wv * B4_LDS;
  for (int k = lane; k < (B4_SLOT - B4_DATA) / 8; k += 64) {
    const int o = B4_DATA + 8 * k;
    double v = 0.0;
    if (o >= B4_ONE && o < B4_VB) v = 1.0;
    if (o >= B4_VB && o < B4_VDUMP) { const int e = (o - B4_VB) / 8, c6 = (e / 6) % 6, r6 = e % 6; v = (c6 == r6) ? 1.0 : 0.0; }  // [t][c6][r6]
    if (o >= B4_KB) { const int e = (o - B4_KB) / 8, c6 = e / 6, r6 = e % 6; v = 2.0 * C.W2[6 * r6 + c6]; }                   // [c6][r6]
    *reinterpret_cast<double*>(lds + o) = v;
    *reinterpret_cast<double*>(lds + B4_SLOT + o) = v;
  }
  reinterpret_cast<double*>(lds + B4_TZERO)[lane] = 0.0;
  if (lane < 8) {
    const double bq = (lane < M) ? fu_entry<M>(*P.c, urow<M>(lane < M ? lane : 0) - 6, lane < M ? lane : 0) : 1.0;  // generic pointer: note at DConsts
    reinterpret_cast<double*>(lds + B4_BU)[lane] = bq;
    reinterpret_cast<double*>(lds + B4_IBU)[lane] = 1.0 / bq;
  }
  __builtin_amdgcn_wave_barrier();
  const unsigned lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)lds;

  // ---- lane-dependent constants
  const bool row1 = h != 0;
  const int col = (j < 12) ? (j + 6 * h) % 12 : j;  // the column this lane holds
  const double m12 = (j < 12) ? 1.0 : 0.0;
  int mycol = -1;  // input whose column of Mt sits in this lane (row 1; row 0 mimics it on junk)
#pragma unroll
  for (int u = 0; u < M; u++) if (j == urow<M>(u) + LO) mycol = u;
  double Rt[M], wm[M];
  {
    double ibu0[M], ibc = 0.0;
#pragma unroll
    for (int u = 0; u < M; u++) ibu0[u] = 1.0 / fu_entry<M>(*P.c, urow<M>(u) - 6, u);
#pragma unroll
    for (int u = 0; u < M; u++) if (mycol == u) ibc = ibu0[u];
#pragma unroll
    for (int u = 0; u < M; u++) {
      Rt[u] = (mycol >= 0) ? 2.0 * C.R[u * M + (mycol >= 0 ? mycol : 0)] * ibu0[u] * ibc : 0.0;
      wm[u] = (mycol > u) ? 1.0 : 0.0;
    }
  }
  auto ibu_load = [&](double (&ib)[M]) {
    const f64x2* p = reinterpret_cast<const f64x2*>(lds + B4_IBU);
#pragma unroll
    for (int u = 0; u < M; u += 2) { const f64x2 w = p[u / 2]; ib[u] = w.x; ib[u + 1] = w.y; }
  };
  const bool isVec = (j == 12 || j == 13), hasD = (j == 12 && !closed);
  const bool vcol = j < 12 && col >= 6;  // columns of the velocity block
  const unsigned lg = (unsigned)t * 16u;
  const unsigned ZP = (unsigned)B4_DATA, ONE = (unsigned)B4_ONE;
  // field of [F_x | d][k][col], or where it is structurally 0 / 1
  auto fx_off = [&](int k) -> unsigned {
    const int c = col;
    if (j == 12) return hasD ? lg + FOFF2(REC_D + k) : ZP;
    if (j > 12) return ZP;
    if (k < 3) {
      if (c < 3) return lg + FOFF2(REC_RI + 3 * c + k);
      if (c >= 6 && c < 9) return lg + FOFF2(REC_JR + 3 * (c - 6) + k);
      return ZP;
    }
    if (k < 6) {
      if (c < 3) return lg + FOFF2(REC_TRI + 3 * c + (k - 3));
      if (c < 6) return lg + FOFF2(REC_RI + 3 * (c - 3) + (k - 3));
      if (c < 9) return lg + FOFF2(REC_QR + 3 * (c - 6) + (k - 3));
      return lg + FOFF2(REC_JR + 3 * (c - 9) + (k - 3));
    }
    return (c >= 6) ? (unsigned)B4_VB + (unsigned)((t * 6 + (c - 6)) * 6 + (k - 6)) * 8u : ZP;  // velocity block: the image
  };
  unsigned oA[12], oL[6];
#pragma unroll
  for (int m = 0; m < 12; m++) oA[m] = fx_off((m + 6 * h) % 12);
#pragma unroll
  for (int r = 0; r < 6; r++) {
    const int gr = 6 * h + r;  // global row
    oL[r] = isVec ? lg + FOFF2(REC_LX + gr) : (j < 12 && col < 6 && gr < 6) ? lg + FOFF2(REC_LXX + sym6(gr, col))
            : (j < 12 && col >= 6 && gr >= 6) ? (unsigned)B4_KB + (unsigned)((col - 6) * 6 + (gr - 6)) * 8u : ZP;  // 2 W2 opens rows 6.. of columns 6..
  }
  const unsigned oU = (isVec && row1) ? lg + FOFF2(REC_LU) : ZP;  // l_u: vector columns, row 1 (G lives there)
  // velocity block, column c6 = col - 6 = 3 Cb + cc (tolg_backward3.h has the derivation): entries alpha w_k + beta v_k
  // in rows kA = cc + 1, kB = cc + 2 (mod 3) of each 3-row block.  The lanes of row 0 that hold such a column compute
  // the four entries a step ahead and write them into the column's image in LDS, where both rows read the column from.
  unsigned oXA = ZP, oXB = ZP, oWA = (unsigned)B4_VDUMP, oWB = oWA;
  const bool prep = vcol && !row1;
  double c_aA0 = 0, c_bA0 = 0, c_aB0 = 0, c_bB0 = 0, c_bA1 = 0, c_bB1 = 0;
  if (prep) {
    const Consts& G = *P.c;
    const int Cb = (col - 6) / 3, cc = (col - 6) % 3, kA = (cc + 1) % 3, kB = (cc + 2) % 3;
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
    oXA = lg + FOFF2(REC_XI + 2 * kA); oXB = lg + FOFF2(REC_XI + 2 * kB);
    const unsigned img = (unsigned)B4_VB + (unsigned)((t * 6 + (col - 6)) * 6) * 8u;
    oWA = img + (unsigned)kA * 8u; oWB = img + (unsigned)kB * 8u;  // rows 6 + k (and 9 + k at +24)
  }
  // the pre-pass for the knot that lies in slot s (its records have landed)
  auto velocity_block = [&](int s) {
    if (!prep) return;
    char* sn = lds + s * B4_SLOT;
    const f64x2 xA = *reinterpret_cast<const f64x2*>(sn + oXA), xB = *reinterpret_cast<const f64x2*>(sn + oXB);  // (w, v) of kA, kB
    const double eA0 = fma(c_aA0, xB.x, c_bA0 * xB.y), eB0 = fma(c_aB0, xA.x, c_bB0 * xA.y), eA1 = c_bA1 * xB.y, eB1 = c_bB1 * xA.y;
    *reinterpret_cast<double*>(sn + oWA) = eA0; *reinterpret_cast<double*>(sn + oWB) = eB0;
    *reinterpret_cast<double*>(sn + oWA + 24) = eA1; *reinterpret_cast<double*>(sn + oWB + 24) = eB1;
  };
  // exchange slots: own, the sibling lane's (same column, other row), the row-1 lane of this lane's column
  const int jsib = (j < 12) ? (j + 6) % 12 : j;
  const unsigned xzW = (unsigned)B4_XZ + (unsigned)lane * 48u;
  const unsigned xzR = (unsigned)B4_XZ + (unsigned)(32 * t + 16 * (1 - h) + jsib) * 48u;
  const unsigned xyW = (unsigned)B4_XY + (unsigned)(16 * t + j) * 96u;  // (written by row 1 only)
  const unsigned xyR = (unsigned)B4_XY + (unsigned)(16 * t + (row1 ? j : jsib)) * 96u;
  // symmetrisation scratch, column-major per trajectory: the lane writes rows 6h.. of its column (48 contiguous
  // bytes), reads row `col` of the columns 6h.. (96 bytes apart); vector lanes write a dump and read zeros
  const unsigned wTR = (j < 12) ? (unsigned)B4_TR + ((unsigned)t * 144u + (unsigned)col * 12u + 6u * (unsigned)h) * 8u
                                : (unsigned)B4_TDUMP + (unsigned)((lane >> 4) * 4 + (j - 12)) * 48u;
  const unsigned rTR = (j < 12) ? (unsigned)B4_TR + ((unsigned)t * 144u + 6u * (unsigned)h * 12u + (unsigned)col) * 8u : (unsigned)B4_TZERO;
  const double hsym = (j < 12) ? 0.5 : 1.0;
  const unsigned sB = (unsigned)P.Bp * 8u;
  const unsigned vr = REC_VR(b);
  const unsigned vG = GK_VG(b, M) + GOFF(0, (j < 12 ? col : 12), M);
  const size_t recStride = (size_t)P.recF * P.Bp, gStride = (size_t)13 * M * P.Bp;
  constexpr unsigned blockBytes4 = (unsigned)rec_fields(M, GRAV, AL, false) * 32u;  // one knot of the workgroup's four trajectories
  constexpr int NPAIR = rec_fields(M, GRAV, AL, false) / 2;
  static_assert(rec_fields(M, GRAV, AL, false) % 2 == 0 && 2 * NPAIR <= 128, "two DMA bursts per knot and wave");

  // one knot of this wave's two trajectories into LDS slot s: chunk q = 2 * pair + trajectory (16 bytes) from
  // pair * 64 + (2 wv + trajectory) * 16 of the group's run; chunks past the record re-read its last pair (into the
  // slot's unused tail)
  unsigned dv0, dv1;
  {
    const int q0 = lane, q1 = 64 + lane;
    const int p0 = q0 >> 1, p1 = (q1 >> 1) < NPAIR ? (q1 >> 1) : NPAIR - 1;
    dv0 = (unsigned)p0 * 64u + (unsigned)(2 * wv + (q0 & 1)) * 16u;
    dv1 = (unsigned)p1 * 64u + (unsigned)(2 * wv + (q1 & 1)) * 16u - 1024u;
  }
  auto dma_from = [&](const char* src, int s) {
    const unsigned dst = (unsigned)__builtin_amdgcn_readfirstlane((int)(lds0 + (unsigned)s * B4_SLOT));
    rl_dma16x2v(uniform_ptr(src), dv0, dv1, dst);
  };
  auto dma_knot = [&](int i, int s) {
    dma_from(reinterpret_cast<const char*>(P.REC + recStride * i) + (size_t)blockIdx.x * blockBytes4, s);
  };
  const char* rec_run = reinterpret_cast<const char*>(P.REC + recStride * (size_t)(N > 2 ? N - 2 : 0)) + (size_t)blockIdx.x * blockBytes4;
  const double* gk_run = P.GK + gStride * (size_t)(N + 1);
  const size_t recStrideB = recStride * 8;

  // terminal condition: V = [l_xx(N) | l_x(N)] with P weights (traopt_controller.py:2956-2957), rows 6h..
  double V[6];
  {
    __amdgpu_buffer_rsrc_t rR = mkbuf(P.REC + recStride * N, (unsigned)P.recF * sB);
    const unsigned OOB = 0x40000000u;
#pragma unroll
    for (int r = 0; r < 6; r++) {
      const int gr = 6 * h + r;
      const bool has = isVec || (j < 12 && col < 6 && gr < 6);
      const int fl = isVec ? REC_LX + gr : REC_LXX + sym6(gr < 6 ? gr : 0, col < 6 ? col : 0);
      const double t1 = bld(rR, has ? vr + FOFF(fl) : OOB, 0);
      const double p2 = (j < 12 && col >= 6 && gr >= 6) ? 2.0 * C.P2[6 * (gr - 6) + (col - 6)] : 0.0;
      V[r] = t1 + p2;
    }
  }
  double gsum = 0;
  double Kst[M];
#pragma unroll
  for (int u = 0; u < M; u++) Kst[u] = 0;
  auto store_gains = [&](const double* gk) {
    if (act && row1 && j < 13) {
      __amdgpu_buffer_rsrc_t rGs = mkbuf(gk, 13 * M * sB);
#pragma unroll
      for (int u = 0; u < M; u += 2) bst2(rGs, vG, GOFF(u, 0, M), Kst[u], Kst[u + 1]);
    }
  };
  // the trajectory's verdict on a row-1 predicate (row 0 computes on junk)
  auto of_row1 = [&](bool p) -> bool { return ((__ballot(p) >> (32 * t + 16)) & 1ull) != 0; };

  // ---- one knot.  SLOT (compile time): the LDS slot that holds knot i; the loop below is unrolled by two.
  auto step = [&](int i, auto slot_tag) {
    constexpr int SLOT = decltype(slot_tag)::value;
    const char* sl = lds + SLOT * B4_SLOT;
    auto ld = [&](unsigned off) -> double { return *reinterpret_cast<const double*>(sl + off); };
    auto ld2 = [&](unsigned off, int k, double& x0, double& x1) {
      const f64x2 w = *reinterpret_cast<const f64x2*>(sl + off + k * 32);
      x0 = w.x; x1 = w.y;
    };
    // (the records of knot i have landed and its velocity-block image is written: the tail of the previous step, or the
    // prologue, waited and ran the pre-pass)
    double A[12], Qh[6], lu[M];
#pragma unroll
    for (int m = 0; m < 12; m++) A[m] = ld(oA[m]);
    // ---- Z = V [F_x | d], rows 6h..  (+ V_x in the vector column; the adjoint passes through)
    double Z[6];
#pragma unroll
    for (int r = 0; r < 6; r++) Z[r] = (1.0 - m12) * V[r];
    dpp6_cols<0>(Z, V, A[0]); dpp6_cols<1>(Z, V, A[1]); dpp6_cols<2>(Z, V, A[2]); dpp6_cols<3>(Z, V, A[3]);
    dpp6_cols<4>(Z, V, A[4]); dpp6_cols<5>(Z, V, A[5]); dpp6_cols<6>(Z, V, A[6]); dpp6_cols<7>(Z, V, A[7]);
    dpp6_cols<8>(Z, V, A[8]); dpp6_cols<9>(Z, V, A[9]); dpp6_cols<10>(Z, V, A[10]); dpp6_cols<11>(Z, V, A[11]);
#pragma unroll
    for (int r = 0; r < 6; r++) Qh[r] = ld(oL[r]);
#pragma unroll
    for (int a = 0; a < M; a += 2) ld2(oU, a / 2, lu[a], lu[a + 1]);
    // the slot is consumed: last knot's gains go out, the records of knot i - 2 come into this slot.  Stores first: the
    // wait at the end of a step covers both, in order.
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    gk_run -= gStride;
    if (i < N - 1) store_gains(gk_run);
    rec_run -= recStrideB;
    if (i >= 2) dma_from(rec_run, SLOT);
    __builtin_amdgcn_sched_barrier(0);
    // the other six rows of this lane's column of Z: on their way while the first half of the Qh product runs
    double X[6];
    {
      f64x2* w = reinterpret_cast<f64x2*>(lds + xzW);
      w[0] = f64x2{Z[0], Z[1]}; w[1] = f64x2{Z[2], Z[3]}; w[2] = f64x2{Z[4], Z[5]};
      const f64x2* r_ = reinterpret_cast<const f64x2*>(lds + xzR);
      const f64x2 x0 = r_[0], x1 = r_[1], x2 = r_[2];
      X[0] = x0.x; X[1] = x0.y; X[2] = x1.x; X[3] = x1.y; X[4] = x2.x; X[5] = x2.y;
    }
    // ---- Qh = [l_xx | l_x] + F_x^T Z, rows 6h..: register m of the column of F_x pairs with row (m + 6h) mod 12 of Z,
    // i.e. the lane's own six rows for m < 6 and the sibling's for m >= 6.  Skipped: what is structurally zero in BOTH
    // rows (rows 3..5 of the accumulators against m = 6..8).
    dpp_rows<0, 6>(Qh, A[0], Z[0]); dpp_rows<0, 6>(Qh, A[1], Z[1]); dpp_rows<0, 6>(Qh, A[2], Z[2]);
    dpp_rows<0, 6>(Qh, A[3], Z[3]); dpp_rows<0, 6>(Qh, A[4], Z[4]); dpp_rows<0, 6>(Qh, A[5], Z[5]);
    dpp_rows<0, 3>(Qh, A[6], X[0]); dpp_rows<0, 3>(Qh, A[7], X[1]); dpp_rows<0, 3>(Qh, A[8], X[2]);
    dpp_rows<0, 6>(Qh, A[9], X[3]); dpp_rows<0, 6>(Qh, A[10], X[4]); dpp_rows<0, 6>(Qh, A[11], X[5]);
    // Q_xx on its way through LDS for the symmetrisation (every second knot, as in k_backward3)
    constexpr bool SYM = SLOT == 0;
    if constexpr (SYM) {
      f64x2* w = reinterpret_cast<f64x2*>(lds + wTR);
      w[0] = f64x2{Qh[0], Qh[1]}; w[1] = f64x2{Qh[2], Qh[3]}; w[2] = f64x2{Qh[4], Qh[5]};
    }
    double T[6];
    auto symmetrise = [&]() {
      if constexpr (!SYM) return;
#pragma unroll
      for (int r = 0; r < 6; r++) T[r] = *reinterpret_cast<const double*>(lds + rTR + 96 * r);
#pragma unroll
      for (int r = 0; r < 6; r++) Qh[r] = hsym * (Qh[r] + T[r]);
    };
    auto grad_term = [&](const double (&G)[M]) {  // ||Q_u|| = ||D G|| (vector lane, MS) / ||l_u + F_u^T p|| (adjoint lane, SS)
      double s0 = 0, s1 = 0;
      const f64x2* bt = reinterpret_cast<const f64x2*>(lds + B4_BU);
#pragma unroll
      for (int u = 0; u < M; u += 2) {
        const f64x2 bu2 = bt[u / 2];
        const double q0 = bu2.x * G[u], q1 = bu2.y * G[u + 1];
        s0 = fma(q0, q0, s0); s1 = fma(q1, q1, s1);
      }
      const double s_ = s0 + s1;
      double y = __builtin_amdgcn_rsq(s_);
      { const double g_ = s_ * y, h_ = 0.5 * y; y = 2.0 * fma(h_, fma(-h_, g_, 0.5), h_); }
      gsum += (s_ > 0.0) ? s_ * y : 0.0;
    };
    // Y and zn of this lane's column as row 1 computed them (row 1 reads its own slot back)
    auto xy_write = [&](const double (&Yw)[M], const double (&zw)[M]) {
      if (row1) {
        f64x2* w = reinterpret_cast<f64x2*>(lds + xyW);
#pragma unroll
        for (int u = 0; u < M; u += 2) { w[u / 2] = f64x2{Yw[u], Yw[u + 1]}; w[3 + u / 2] = f64x2{zw[u], zw[u + 1]}; }
      }
    };
    auto xy_read = [&](double (&Yx)[M], double (&zx)[M]) {
      const f64x2* r_ = reinterpret_cast<const f64x2*>(lds + xyR);
#pragma unroll
      for (int u = 0; u < M; u += 2) {
        const f64x2 a = r_[u / 2], c = r_[3 + u / 2];
        Yx[u] = a.x; Yx[u + 1] = a.y; zx[u] = c.x; zx[u + 1] = c.y;
      }
    };
    auto update = [&](const double (&Yx)[M], const double (&zx)[M]) {  // V' rows 6h.. = sym(Q_xx) + Y^T zn
#pragma unroll
      for (int u = 0; u < M; u++) {
        if (u == 0) dpp_rows<0, 6>(Qh, Yx[0], zx[0]);
        if (u == 1) dpp_rows<0, 6>(Qh, Yx[1], zx[1]);
        if (u == 2) dpp_rows<0, 6>(Qh, Yx[2], zx[2]);
        if (u == 3) dpp_rows<0, 6>(Qh, Yx[3], zx[3]);
        if constexpr (M > 4) {
          if (u == 4) dpp_rows<0, 6>(Qh, Yx[4], zx[4]);
          if (u == 5) dpp_rows<0, 6>(Qh, Yx[5], zx[5]);
        }
      }
    };
    // what follows a settled factorisation (== Eq. 11b/11c of traopt_controller.py:2998-3004 for the exact gains)
    auto finish = [&](double (&Y)[M], const double (&Uf)[M], double (&nri)[M]) {
      grad_term(Y);
      if (!ms) {  // the single-shooting adjoint lane takes no gain correction
#pragma unroll
        for (int u = 0; u < M; u++) nri[u] = (j == 13) ? 0.0 : nri[u];
      }
      double zn[M], nx[M], Yx[M], zx[M];
      ldl3_forward<M, LO>(Uf, nri, Y, zn);
      xy_write(Y, zn);
      xy_read(Yx, zx);
      ldl3_backward_nx<M, LO>(Uf, nri, Y, zn, nx);  // (in place on Y: the exchange has its copy)
      {
        double ib[M];
        ibu_load(ib);
#pragma unroll
        for (int u = 0; u < M; u++) Kst[u] = ib[u] * nx[u];
      }
      symmetrise();
      update(Yx, zx);
    };
    // ---- G = rows S of Z (+ D^-1 l_u in the vector columns), Mt = V_SS + 2 D^-1 R D^-1; factorisation and PD test
    // (traopt_controller.py:2964-2995 with mu == 0, :3052-3060).  Row 1: rows S are its registers urow - 6.
    double Y[M], Uf[M], nri[M];
    {
      double ib[M];
      ibu_load(ib);
#pragma unroll
      for (int u = 0; u < M; u++) {
        Y[u] = fma(lu[u], ib[u], Z[urow<M>(u) + LO]);
        Uf[u] = V[urow<M>(u) + LO] + Rt[u];
      }
    }
    bool ok = true;
    ldl3_factor_ok<M, 0, LO>(Uf, nri, wm, ok);
    if (__any(act && !of_row1(ok))) return true;  // a non-PD Q_uu: the regularisation loop is k_backward3's
    if (act) delta = fmin(1.0, delta) * 0.5;       // schedule(true) with mu == 0 (:2986-2991)
    finish(Y, Uf, nri);
#pragma unroll
    for (int r = 0; r < 6; r++) V[r] = Qh[r];
    // the next knot (other slot): requested two steps ago -- everything but this step's gain stores and request has
    // landed; then its velocity-block image
    if (i >= 1) {
      if (i == 1) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(%0)" :: "n"(M / 2 + 2) : "memory");
      velocity_block(1 - SLOT);
    }
    return false;
  };

  // ---- the sweep (per wave; the two waves of a workgroup meet only at the barrier behind it)
  bool stopped = __any(act && mu != 0.0);  // regularised knots ahead (the first sweep of a solve): k_backward3
  if (__any(act) && !stopped) {
    // prologue: knots N-1 and N-2 into the two slots
    dma_knot(N - 1, (N - 1) & 1);
    if (N >= 2) dma_knot(N - 2, (N - 2) & 1);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    velocity_block((N - 1) & 1);
    int i = N - 1;
    if (i & 1) { stopped = step(i, std::integral_constant<int, 1>()); i--; }
    for (; i >= 1 && !stopped; i -= 2) {
      stopped = step(i, std::integral_constant<int, 0>());
      if (!stopped) stopped = step(i - 1, std::integral_constant<int, 1>());
    }
    if (i == 0 && !stopped) stopped = step(0, std::integral_constant<int, 0>());
    if (!stopped) store_gains(P.GK);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // (no LDS-DMA of this wave in flight past this point)
  }
  if (lane == 0) wave_stopped[wv] = stopped ? 1 : 0;
  __syncthreads();
  const bool redo = (wave_stopped[0] | wave_stopped[1]) != 0;
  if (threadIdx.x == 0) P.k2_redo[blockIdx.x] = redo ? 1 : 0;
  if (redo) return;  // the group's sweep is k_backward3's: nothing of the epilogue may have happened
  // ---- epilogue: gradient norm, convergence test (traopt_controller.py:2527-2532, :1937-1942) -- row 1
  double grad = (ms ? bcast<12>(gsum) : bcast<13>(gsum)) / (double)N;
  if (act && row1 && j == 0) {
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
