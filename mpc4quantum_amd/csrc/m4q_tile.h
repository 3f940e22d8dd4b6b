// m4q_tile.h - the two sweeps of one QP solve on fp64 matrix-core tiles (v_mfma_f64_4x4x4_4b_f64), real arithmetic.
//
// Why.  In the DPP layout of m4q_device.h one 16-lane row owns one member and a lane owns one matrix column: at d = 3 that
// fills 9 (8 traceless) of 16 lanes, every cross-lane sum is a chain of unit-multiplier FMAs, and the fp64 pipe spends
// ~2,400 cycles per horizon index of the backward sweep on four members.  v_mfma_f64_4x4x4_4b_f64 runs FOUR independent
// 4x4x4 products per instruction at the same pipe rate (16.7 cycles, tools/ubench_mfma.hip) with every slot useful when
// the matrices are cut into 4x4 tiles - and an n = 8 matrix (the traceless coordinates of a qutrit, m4q_mpc.h) is exactly
// 2 x 2 tiles.  Reductions come for free: a product with a replicated operand IS the broadcast.
//
// Lane map of the instruction on gfx950, found by experiment (tools/mfma_layout.hip, profiles/r03_mfma_layout.txt):
//     block (= member) mb = (lane >> 2) & 3,   r = lane >> 4,   q = lane & 3          lane = 16 r + 4 mb + q
//     A operand:  A[i][k]  in lane (r = k, q = i)        B operand:  B[k][j]  in lane (r = k, q = j)
//     C / D:      D[i][j]  in lane (r = i, q = j)
// So with every 4x4 tile held "naturally" - X[r][q] in lane (r, q) of the member's 16 lanes - one instruction computes
//     mm(X, Y, C) = C + X^T Y                      (the A operand reads its tile transposed)
// and nothing ever has to be moved between lanes:
//   * P is symmetric and the recursion only needs P S = P^T S and S^T (P S): both are X^T Y forms of natural tiles;
//   * a vector is kept ROW-REPLICATED (v[4K + r] in every q of tile K): then  rowrep(M^T v) = sum_K mm(M[K][I], v[K]),
//     colrep(M^T v) = sum_K mm(v[K], M[K][J])  and a dot product  sum_K mm(x[K], y[K])  lands REPLICATED in all 16 lanes;
//   * an outer product b k^T is an elementwise product of a row-replicated b and a column-replicated k.
// The only values that cross lanes outside an MFMA are the m x m matrix G = R + B^H P B and h (m <= 3): they come out of one
// product as a natural tile and are spread to the member's lanes through 16 doubles of LDS.
//
// Scope: the clipped Riccati solve of the closed loop (riccati_backward / rollout_forward of m4q_mpc.h, same arithmetic:
// Joseph form, Delta_t = -B_t u_t, xbar_{t+1}) for real models on NS coordinates with a constant target over the window.
// Everything else (M4Q_QP_REF_LQR, the exact box QP, time-varying targets, complex models) stays on the DPP sweeps.
//
// Reference arithmetic: mpc4quantum/lqr.py:28-79 (+ Delta, optimize.py:41), mpc4quantum/linearize.py:43-70.
#pragma once
#include "m4q_mpc.h"

namespace m4q {

struct TileGeo {
  int mb, r, q;
  __device__ __forceinline__ TileGeo() {
    const int lane = threadIdx.x;
    r = lane >> 4;
    mb = (lane >> 2) & 3;
    q = lane & 3;
  }
};

// C + X^T Y on the four members' tiles at once
__device__ __forceinline__ double mm(double x, double y, double c) { return __builtin_amdgcn_mfma_f64_4x4x4f64(x, y, c, 0, 0, 0); }

// Hand-over between the closed-loop state machine (DPP rows: member = lane >> 4) and the tile sweeps (member = (lane >> 2) & 3):
// per member 32 doubles and 4 words of LDS.
constexpr int TILE_IO_DOUBLES = 32;     // [0, 16) x_cur | [16, 19) lo0 | [19, 22) hi0 | [22, 25) u_first | [25] chk | [26, 32) spare
constexpr int TILE_IO_WORDS = 4;        // [0] flags (1 running, 2 shift_out) | [1] xbm byte offset | [2] ubm byte offset | [3] spare
constexpr int TILE_GB_DOUBLES = 18;     // the G / h tile of one member: 16 doubles at a pitch of 18 - every lane of a member reads the
                                        // same entry, and at pitch 16 (128 bytes) members 0 / 2 and 1 / 3 read the same banks: 46 % of the
                                        // tile kernel's LDS-active cycles were bank conflicts (gpurun_out/r04tile counters)
constexpr int TILE_LDS_BYTES = 4 * (TILE_IO_DOUBLES * 8 + TILE_IO_WORDS * 4 + TILE_GB_DOUBLES * 8);

template <int NS, int NU, int ORDER>
struct TileSweeps {
  static constexpr int NT = (NS + 3) / 4;
  static constexpr int NP = PowTab<NU, ORDER>::NP;
#ifndef M4Q_TILE_PF
#define M4Q_TILE_PF 3
#endif
  static constexpr int PF = M4Q_TILE_PF;          // horizon indices the operand fetch runs ahead of the arithmetic
  static_assert(NU <= 3, "W = [B | c] must fit one column tile");
  TileGeo L;
  const double* mdl;     // LDS, this member: [1+NP][NS][PITCH]
  int T;
  GView Xg, Ug, gains;        // positioned on this member (lane offsets), workspace
  GView xbm, ubm;             // target window of this member
  const double* Q; const double* Qf; const double* R;   // LDS, shared: [NS][NS], [NS][NS], [NU][NU]
  volatile __attribute__((address_space(3))) double* gb;        // LDS, this member: 16 doubles

  // ---- operands ------------------------------------------------------------------------------------------------------------
  __device__ __forceinline__ double mdl_nat(int p, int I, int J) const {            // block p, element [4I + r][4J + q]
    const int i = 4 * I + L.r, k = 4 * J + L.q;
    const double v = mdl[ModelPitch<NS>::at(p, i < NS ? i : 0, k < NS ? k : 0)];
    return (i < NS && k < NS) ? v : 0.0;
  }
  __device__ __forceinline__ double mdl_tr(int p, int I, int J) const {             // block p transposed: element [4J + q][4I + r]
    const int i = 4 * J + L.q, k = 4 * I + L.r;
    const double v = mdl[ModelPitch<NS>::at(p, i < NS ? i : 0, k < NS ? k : 0)];
    return (i < NS && k < NS) ? v : 0.0;
  }
  __device__ __forceinline__ double sym_nat(const double* M, int I, int J) const {  // shared cost matrix [NS][NS]
    const int i = 4 * I + L.r, k = 4 * J + L.q;
    const double v = M[(i < NS ? i : 0) * NS + (k < NS ? k : 0)];
    return (i < NS && k < NS) ? v : 0.0;
  }
  // row-replicated vector from a global array: element 4K + r of the vector that starts at `base`
  __device__ __forceinline__ void ld_row(const GView& v, unsigned base, double (&out)[NT]) const {
#pragma unroll
    for (int K = 0; K < NT; ++K) {
      const int e = 4 * K + L.r;
      const double x = v.ld<double>(base + (e < NS ? e : 0));
      out[K] = e < NS ? x : 0.0;
    }
  }
  // rowrep(M^T v):  out[I] = c[I] + sum_K mm(M[K][I], v[K])
  __device__ __forceinline__ void matvec_t(const double (&M)[NT][NT], const double (&v)[NT], const double (&c)[NT], double (&out)[NT]) const {
#pragma unroll
    for (int I = 0; I < NT; ++I) {
      double acc = c[I];
#pragma unroll
      for (int K = 0; K < NT; ++K) acc = mm(M[K][I], v[K], acc);
      out[I] = acc;
    }
  }

  struct Ops {            // what one horizon index reads from memory: the linearisation point and the control target
    double ug[NU], ub[NU];
    double xg[NT];
  };
  __device__ __forceinline__ Ops load(int t) const {
    Ops o;
#pragma unroll
    for (int s = 0; s < NU; ++s) {
      o.ug[s] = Ug.ld<double>(t * NU + s);
      o.ub[s] = ubm.ld<double>(t * NU + s);
    }
    ld_row(Xg, (unsigned)t * NS, o.xg);
    return o;
  }
  // B_t[:, s] = sum_p (N_p x_g) d mono_p / d u_s  (linearize.py:50-59), row-replicated.  NpT: natural tiles of N_p^T.
  // (A first version formed N_p x_g for all t in a pre-pass, four indices per product, and kept them in a workspace: fewer
  //  products, but 15 KB more workspace traffic per solve, and the sweeps wait on memory - profiles/r03_tile_log.txt)
  __device__ __forceinline__ void controls(const Ops& o, const Poly<NU, ORDER>& po, const double (&NpT)[NP][NT][NT], double (&b)[NU][NT]) const {
    double nx[NP][NT], zero[NT];
#pragma unroll
    for (int K = 0; K < NT; ++K) zero[K] = 0.0;
#pragma unroll
    for (int p = 0; p < NP; ++p) matvec_t(NpT[p], o.xg, zero, nx[p]);
#pragma unroll
    for (int s = 0; s < NU; ++s)
#pragma unroll
      for (int K = 0; K < NT; ++K) {
        if constexpr (ORDER == 1) {
          b[s][K] = nx[s][K];                                                       // monomial p is u_p (order1_is_identity)
        } else {
          double acc = 0.0;
#pragma unroll
          for (int p = 0; p < NP; ++p) acc = fma(po.dpu[s][p], nx[p][K], acc);
          b[s][K] = acc;
        }
      }
  }

  // ---- backward Riccati sweep (lqr.py:28-65 with Delta and xbar_{t+1}); gains [t][col 0..NS][NU] -------------------------
  __device__ __forceinline__ void backward(bool store_ok) const {
    static_assert(ORDER != 1 || order1_is_identity<NU>(), "order-1 library must list u_1 .. u_m in order");
    if constexpr (ORDER == 1) { backward_o1(store_ok); return; }      // (m4q_tile2.h: packed accesses, operands two indices ahead)
    // model tiles, natural
    double M[1 + NP][NT][NT];
#pragma unroll
    for (int p = 0; p <= NP; ++p)
#pragma unroll
      for (int I = 0; I < NT; ++I)
#pragma unroll
        for (int J = 0; J < NT; ++J) M[p][I][J] = mdl_nat(p, I, J);
    double NpT[NP][NT][NT];
#pragma unroll
    for (int p = 0; p < NP; ++p)
#pragma unroll
      for (int K = 0; K < NT; ++K)
#pragma unroll
        for (int I = 0; I < NT; ++I) NpT[p][K][I] = mdl_tr(1 + p, K, I);
    double Qt[NT][NT], P[NT][NT], pv[NT];
#pragma unroll
    for (int I = 0; I < NT; ++I) {
      pv[I] = 0.0;
#pragma unroll
      for (int J = 0; J < NT; ++J) { Qt[I][J] = sym_nat(Q, I, J); P[I][J] = sym_nat(Qf, I, J); }
    }
    // constant target over the window: xbar, and M_p xbar once per sweep (rowrep; (M_p^T)^T xbar through the transposed tiles)
    double xb[NT], tt[1 + NP][NT], zero[NT];
    ld_row(xbm, 0, xb);
#pragma unroll
    for (int K = 0; K < NT; ++K) zero[K] = 0.0;
#pragma unroll
    for (int p = 0; p <= NP; ++p) {
      double Mt[NT][NT];
#pragma unroll
      for (int K = 0; K < NT; ++K)
#pragma unroll
        for (int I = 0; I < NT; ++I) Mt[K][I] = mdl_tr(p, K, I);
      matvec_t(Mt, xb, zero, tt[p]);
    }
    // lane masks as numbers: column selectors of W and row selectors of the gain coefficients
    double mq[NU + 1], mr[NU];
#pragma unroll
    for (int s = 0; s <= NU; ++s) mq[s] = L.q == s ? 1.0 : 0.0;
#pragma unroll
    for (int s = 0; s < NU; ++s) mr[s] = L.r == s ? 1.0 : 0.0;
    double Rm[NU][NU];
#pragma unroll
    for (int s = 0; s < NU; ++s)
#pragma unroll
      for (int l = 0; l < NU; ++l) Rm[s][l] = R[s * NU + l];

    // Operands are fetched PF horizon indices ahead: one index of this sweep lasts ~1,400 cycles per wavefront and the
    // workspace comes from beyond the L2 (the first version fetched one index ahead, as the DPP sweeps do at ~2,400 cycles per
    // index, and spent 52 % of its wavefront time at s_waitcnt: profiles/r03_pmc_tile_prefetch1.txt).  A ring of PF operand
    // sets, the loop unrolled PF times so that every set lives in fixed registers.
    auto step = [&](int t, const Ops& cur) __attribute__((always_inline)) {
      Poly<NU, ORDER> po;
      po.eval(cur.ug);
      double At[NT][NT], b[NU][NT], c[NT];
#pragma unroll
      for (int I = 0; I < NT; ++I)
#pragma unroll
        for (int J = 0; J < NT; ++J) {
          double a = M[0][I][J];
#pragma unroll
          for (int p = 0; p < NP; ++p) a = fma(po.pu[p], M[1 + p][I][J], a);       // A_t = A + sum_p polyu_p N_p (linearize.py:43-48)
          At[I][J] = a;
        }
      controls(cur, po, NpT, b);
#pragma unroll
      for (int K = 0; K < NT; ++K) {
        double a = tt[0][K] - xb[K];                                                // A_t xbar - xbar_{t+1}
#pragma unroll
        for (int p = 0; p < NP; ++p) a = fma(po.pu[p], tt[1 + p][K], a);
#pragma unroll
        for (int s = 0; s < NU; ++s) a = fma(b[s][K], cur.ub[s] - cur.ug[s], a);    // + B ubar + Delta, Delta = -B u_g
        c[K] = a;
      }
      // W = [b_0 .. b_{m-1} | c | 0] as natural column tiles; Y = P W + [0 | p | 0]
      double W[NT], Y[NT];
#pragma unroll
      for (int K = 0; K < NT; ++K) {
        double w = mq[NU] * c[K];
#pragma unroll
        for (int s = 0; s < NU; ++s) w = fma(mq[s], b[s][K], w);
        W[K] = w;
      }
#pragma unroll
      for (int I = 0; I < NT; ++I) {
        double acc = mq[NU] * pv[I];
#pragma unroll
        for (int K = 0; K < NT; ++K) acc = mm(P[K][I], W[K], acc);                   // P symmetric: P^T W = P W
        Y[I] = acc;
      }
      // H = Y^T A_t: rows 0..m-1 = B^H P A_t.   G4 = W^T Y: [s][s'] = B^H P B, [s][m] = B^H (P c + p)
      double H[NT];
#pragma unroll
      for (int J = 0; J < NT; ++J) {
        double acc = 0.0;
#pragma unroll
        for (int K = 0; K < NT; ++K) acc = mm(Y[K], At[K][J], acc);
        H[J] = acc;
      }
      double G4 = 0.0;
#pragma unroll
      for (int K = 0; K < NT; ++K) G4 = mm(W[K], Y[K], G4);
      // the m x m system, spread to the member's lanes through LDS
      gb[L.r * 4 + L.q] = G4;
      wave_sync();
      cplx g[NU][NU], ginv[NU][NU];
      double h[NU];
#pragma unroll
      for (int s = 0; s < NU; ++s) {
#pragma unroll
        for (int l = s; l < NU; ++l) g[s][l] = mk(gb[s * 4 + l] + Rm[s][l], 0.0);
        h[s] = gb[s * 4 + NU];
      }
      wave_sync();
      herm_inverse<NU>(g, ginv);
      // gains: K = -G^-1 H (column- and row-replicated), k = -G^-1 h                                          lqr.py:61-62
      double Kc[NU][NT], Kr[NU][NT], kk[NU];
#pragma unroll
      for (int s = 0; s < NU; ++s) {
        double cf = 0.0, ks = 0.0;
#pragma unroll
        for (int l = 0; l < NU; ++l) {
          cf = fma(-ginv[s][l].re, mr[l], cf);                                       // tile: -G^-1[s][r] in every q
          ks = fma(-ginv[s][l].re, h[l], ks);
        }
        kk[s] = ks;
#pragma unroll
        for (int J = 0; J < NT; ++J) {
          Kc[s][J] = mm(cf, H[J], 0.0);                                              // [i][j] = sum_k cf[k] H[k][j] = K_s[4J + j], every i
          Kr[s][J] = mm(H[J], cf, 0.0);                                              // [i][j] = sum_k H[k][i] cf[k] = K_s[4J + i], every j
        }
      }
      if (store_ok && L.r == 0) {
        const unsigned gt = (unsigned)t * (NS + 1) * NU;
#pragma unroll
        for (int s = 0; s < NU; ++s) {
#pragma unroll
          for (int J = 0; J < NT; ++J)
            if (4 * J + L.q < NS) gains.st<double>(gt + (4 * J + L.q) * NU + s, Kc[s][J]);
          if (L.q == 0) gains.st<double>(gt + NS * NU + s, kk[s]);
        }
      }
      // closed loop: S = A_t + B K, s = c + B k
      double S[NT][NT], sv[NT];
#pragma unroll
      for (int I = 0; I < NT; ++I) {
        double a = c[I];
#pragma unroll
        for (int s = 0; s < NU; ++s) a = fma(b[s][I], kk[s], a);
        sv[I] = a;
#pragma unroll
        for (int J = 0; J < NT; ++J) {
          double e = At[I][J];
#pragma unroll
          for (int s = 0; s < NU; ++s) e = fma(b[s][I], Kc[s][J], e);
          S[I][J] = e;
        }
      }
      // P S, P s + p
      double PS[NT][NT], w[NT];
#pragma unroll
      for (int I = 0; I < NT; ++I) {
#pragma unroll
        for (int J = 0; J < NT; ++J) {
          double acc = 0.0;
#pragma unroll
          for (int K = 0; K < NT; ++K) acc = mm(P[K][I], S[K][J], acc);
          PS[I][J] = acc;
        }
      }
      matvec_t(P, sv, pv, w);
      // P <- Q + S^H P S + K^H R K;  p <- S^H (P s + p) + K^H R k                                             lqr.py:64-65
      double RK[NU][NT], Rk[NU];
#pragma unroll
      for (int s = 0; s < NU; ++s) {
        double a = 0.0;
#pragma unroll
        for (int l = 0; l < NU; ++l) a = fma(Rm[s][l], kk[l], a);
        Rk[s] = a;
#pragma unroll
        for (int J = 0; J < NT; ++J) {
          double e = 0.0;
#pragma unroll
          for (int l = 0; l < NU; ++l) e = fma(Rm[s][l], Kc[l][J], e);
          RK[s][J] = e;                                                              // (R K)[s][4J + q], every r
        }
      }
      double Pn[NT][NT], pn[NT];
#pragma unroll
      for (int I = 0; I < NT; ++I) {
        double a = 0.0;
#pragma unroll
        for (int s = 0; s < NU; ++s) a = fma(Kr[s][I], Rk[s], a);
        double acc = a;
#pragma unroll
        for (int K = 0; K < NT; ++K) acc = mm(S[K][I], w[K], acc);
        pn[I] = acc;
#pragma unroll
        for (int J = 0; J < NT; ++J) {
          double e = Qt[I][J];
#pragma unroll
          for (int s = 0; s < NU; ++s) e = fma(Kr[s][I], RK[s][J], e);
          double acc2 = e;
#pragma unroll
          for (int K = 0; K < NT; ++K) acc2 = mm(S[K][I], PS[K][J], acc2);
          Pn[I][J] = acc2;
        }
      }
#pragma unroll
      for (int I = 0; I < NT; ++I) {
        pv[I] = pn[I];
#pragma unroll
        for (int J = 0; J < NT; ++J) P[I][J] = Pn[I][J];
      }
    };
    Ops ring[PF];
#pragma unroll
    for (int i = 0; i < PF; ++i) ring[i] = load(T - 1 - i > 0 ? T - 1 - i : 0);
    int t = T - 1;
    for (; t >= PF - 1; t -= PF) {
#pragma unroll
      for (int i = 0; i < PF; ++i) {
        M4Q_NO_HOIST();
        step(t - i, ring[i]);
        ring[i] = load(t - i - PF > 0 ? t - i - PF : 0);
      }
    }
#pragma unroll
    for (int i = 0; i < PF - 1; ++i)
      if (t - i >= 0) step(t - i, ring[i]);
  }

  __device__ __forceinline__ void backward_o1(bool store_ok) const;       // defined in m4q_tile2.h

  // ---- forward rollout with clipping (lqr.py:67-79; optimize.py:41); returns sum |x|^2 + sum u^2 (finite iff all are) -----
  // shift_out: the solution goes straight into the next step's (shifted) guess (mpc.py:271-272), see rollout_forward.
  __device__ __forceinline__ double forward(const double (&x0)[NT], double sat, const double (&lo0)[NU], const double (&hi0)[NU],
                                            const GView& Xd, const GView& Ud, bool shift_out, bool store_ok, double (&u_first)[NU]) const {
    double MT[1 + NP][NT][NT];
#pragma unroll
    for (int p = 0; p <= NP; ++p)
#pragma unroll
      for (int K = 0; K < NT; ++K)
#pragma unroll
        for (int I = 0; I < NT; ++I) MT[p][K][I] = mdl_tr(p, K, I);
    double NpTf[NP][NT][NT];
#pragma unroll
    for (int p = 0; p < NP; ++p)
#pragma unroll
      for (int K = 0; K < NT; ++K)
#pragma unroll
        for (int I = 0; I < NT; ++I) NpTf[p][K][I] = MT[1 + p][K][I];
    double xb[NT], x[NT], cx[NT], zero[NT];
    ld_row(xbm, 0, xb);
    const int xs_shift = shift_out ? 0 : 1, us_shift = shift_out ? -1 : 0;
    const bool wr = store_ok && L.q == 0;
#pragma unroll
    for (int K = 0; K < NT; ++K) {
      x[K] = x0[K];
      cx[K] = 0.0;
      zero[K] = 0.0;
      if (wr && !shift_out && 4 * K + L.r < NS) Xd.st<double>(4 * K + L.r, x[K]);
    }
    double cu = 0.0;
    struct FOps {
      Ops o;
      double Kr[NU][NT], kk[NU];
    };
    auto fload = [&](int t) __attribute__((always_inline)) {
      FOps f;
      f.o = load(t);
      const unsigned gt = (unsigned)t * (NS + 1) * NU;
#pragma unroll
      for (int s = 0; s < NU; ++s) {
#pragma unroll
        for (int K = 0; K < NT; ++K) {
          const int e = 4 * K + L.r;
          const double v = gains.ld<double>(gt + (e < NS ? e : 0) * NU + s);
          f.Kr[s][K] = e < NS ? v : 0.0;
        }
        f.kk[s] = gains.ld<double>(gt + NS * NU + s);
      }
      return f;
    };
    auto step = [&](int t, const FOps& cur) __attribute__((always_inline)) {
      Poly<NU, ORDER> po;
      po.eval(cur.o.ug);
      double AtT[NT][NT], b[NU][NT], ax[NT], dx[NT];
#pragma unroll
      for (int K = 0; K < NT; ++K)
#pragma unroll
        for (int I = 0; I < NT; ++I) {
          double a = MT[0][K][I];
#pragma unroll
          for (int p = 0; p < NP; ++p) a = fma(po.pu[p], MT[1 + p][K][I], a);
          AtT[K][I] = a;
        }
      controls(cur.o, po, NpTf, b);
      matvec_t(AtT, x, zero, ax);                                                   // (A_t^T)^T x = A_t x
      double u[NU];
#pragma unroll
      for (int K = 0; K < NT; ++K) dx[K] = x[K] - xb[K];
#pragma unroll
      for (int s = 0; s < NU; ++s) {
        double acc = cur.kk[s] + cur.o.ub[s];
#pragma unroll
        for (int K = 0; K < NT; ++K) acc = mm(cur.Kr[s][K], dx[K], acc);            // K_s . dx, in every lane          lqr.py:75
        double lo = -sat, hi = sat;
        if (t == 0) { lo = fmax(lo, lo0[s]); hi = fmin(hi, hi0[s]); }
        acc = fmin(fmax(acc, lo), hi);                                              // lqr.py:76
        u[s] = acc;
        if (t == 0) u_first[s] = acc;
        cu = fma(acc, acc, cu);
      }
#pragma unroll
      for (int K = 0; K < NT; ++K) {
        double xn = ax[K];
#pragma unroll
        for (int s = 0; s < NU; ++s) xn = fma(b[s][K], u[s] - cur.o.ug[s], xn);     // A_t x + B u + Delta
        x[K] = xn;
        cx[K] = fma(xn, xn, cx[K]);
        if (wr && 4 * K + L.r < NS) {
          Xd.st<double>((unsigned)(t + xs_shift) * NS + 4 * K + L.r, xn);
          if (shift_out && t == T - 1) Xd.st<double>((unsigned)T * NS + 4 * K + L.r, xn);      // repeat the last column
        }
      }
      if (wr && L.r == 0) {
#pragma unroll
        for (int s = 0; s < NU; ++s) {
          if (t + us_shift >= 0) Ud.st<double>((unsigned)(t + us_shift) * NU + s, u[s]);
          if (shift_out && t == T - 1) Ud.st<double>((unsigned)(T - 1) * NU + s, u[s]);
        }
      }
    };
    FOps ring[PF];
#pragma unroll
    for (int i = 0; i < PF; ++i) ring[i] = fload(i < T ? i : T - 1);
    int t = 0;
    for (; t + PF <= T; t += PF) {
#pragma unroll
      for (int i = 0; i < PF; ++i) {
        M4Q_NO_HOIST();
        step(t + i, ring[i]);
        ring[i] = fload(t + i + PF < T ? t + i + PF : T - 1);
      }
    }
#pragma unroll
    for (int i = 0; i < PF - 1; ++i)
      if (t + i < T) step(t + i, ring[i]);
    double tot = cu;
#pragma unroll
    for (int K = 0; K < NT; ++K) tot = mm(cx[K], 1.0, tot);                         // sum_k cx[k]: padded entries are zero
    return tot;
  }
};

}  // namespace m4q
