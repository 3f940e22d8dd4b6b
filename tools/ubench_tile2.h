// m4q_tile2.h - the backward Riccati sweep on fp64 matrix-core tiles with G independent member groups INTERLEAVED in one
// wavefront ("two, then four, members per 16-lane block-slot").
//
// m4q_tile.h runs one group of four members per wavefront: one v_mfma_f64_4x4x4_4b_f64 = four members' 4x4x4 products, and nearly
// every product of a horizon index consumes the one before (P -> Y = P W -> H, G -> gains -> S -> P S -> S^T (P S) -> P).  Here a
// wavefront carries G such groups (4 G members): every statement of the index is issued for all groups before the next statement,
// so a group's dependent instruction finds G - 1 independent ones of the same kind between itself and its operand.  Same
// arithmetic, same lane map, same operand layout as m4q_tile.h (read its header first); group g's member mb is member 4 g + mb of
// the wavefront.
//
// Measured in isolation by tools/ubench_tile_chain.hip (profiles/r04_tile_chain.txt): two groups in one wavefront are slower than one
// group in each of two wavefronts.  A measurement harness, not product code: the closed-loop kernel runs csrc/m4q_tile3.h.
#pragma once
#include "m4q_tile3.h"

namespace m4q {

template <int NS, int NU, int ORDER, int G>
struct TileBackwardG {
  static constexpr int NT = (NS + 3) / 4;
  static constexpr int NP = PowTab<NU, ORDER>::NP;
#ifndef M4Q_TILE2_PF
#define M4Q_TILE2_PF 2
#endif
  static constexpr int PF = M4Q_TILE2_PF;
  static_assert(NU <= 3, "W = [B | c] must fit one column tile");
  TileGeo L;
  int T;
  const double* Q; const double* Qf; const double* R;   // LDS, shared
  struct Member {                 // what differs between the groups: this lane's member of group g
    const double* mdl;            // LDS [1+NP][NS][PITCH]
    GView Xg, Ug, gains, xbm, ubm;
    volatile __attribute__((address_space(3))) double* gb;       // LDS, 16 doubles
  } mem[G];

  __device__ __forceinline__ double mdl_nat(const double* mdl, int p, int I, int J) const {
    const int i = 4 * I + L.r, k = 4 * J + L.q;
    const double v = mdl[ModelPitch<NS>::at(p, i < NS ? i : 0, k < NS ? k : 0)];
    return (i < NS && k < NS) ? v : 0.0;
  }
  __device__ __forceinline__ double mdl_tr(const double* mdl, int p, int I, int J) const {
    const int i = 4 * J + L.q, k = 4 * I + L.r;
    const double v = mdl[ModelPitch<NS>::at(p, i < NS ? i : 0, k < NS ? k : 0)];
    return (i < NS && k < NS) ? v : 0.0;
  }
  __device__ __forceinline__ double sym_nat(const double* M, int I, int J) const {
    const int i = 4 * I + L.r, k = 4 * J + L.q;
    const double v = M[(i < NS ? i : 0) * NS + (k < NS ? k : 0)];
    return (i < NS && k < NS) ? v : 0.0;
  }
  __device__ __forceinline__ void ld_row(const GView& v, unsigned base, double (&out)[NT]) const {
#pragma unroll
    for (int K = 0; K < NT; ++K) {
      const int e = 4 * K + L.r;
      const double x = v.ld<double>(base + (e < NS ? e : 0));
      out[K] = e < NS ? x : 0.0;
    }
  }
  struct Ops {
    double ug[NU], ub[NU];
    double xg[NT];
  };
  __device__ __forceinline__ Ops load(const Member& m, int t) const {
    Ops o;
    ldn<NU>(m.Ug, t * NU, o.ug);                    // (one 16-byte access per pair, as the DPP sweeps: m4q_device.h ldn / stn)
    ldn<NU>(m.ubm, t * NU, o.ub);
    ld_row(m.Xg, (unsigned)t * NS, o.xg);
    return o;
  }

#define M4Q_G for (int g = 0; g < G; ++g)
#ifndef M4Q_T2_EXP
#define M4Q_T2_EXP 0      // tools/ubench_tile_chain.hip only: timing-only ablations (results wrong): 1 no N_p x_g products, 2 no gain
#endif                    // stores, 4 no LDS exchange / inverse, 8 no gain broadcasts (Kc, Kr), 16 no P S / S^T P S products
  __device__ __forceinline__ void backward(bool store_ok) const {
    static_assert(ORDER == 1, "interleaved tile sweep: order 1 (what the measurement needs)");
    double M[G][1 + NP][NT][NT], NpT[G][NP][NT][NT], P[G][NT][NT], pv[G][NT], xb[G][NT], tt[G][1 + NP][NT];
    double Qt[NT][NT];
#pragma unroll
    for (int I = 0; I < NT; ++I)
#pragma unroll
      for (int J = 0; J < NT; ++J) Qt[I][J] = sym_nat(Q, I, J);
#pragma unroll
    M4Q_G {
#pragma unroll
      for (int p = 0; p <= NP; ++p)
#pragma unroll
        for (int I = 0; I < NT; ++I)
#pragma unroll
          for (int J = 0; J < NT; ++J) M[g][p][I][J] = mdl_nat(mem[g].mdl, p, I, J);
#pragma unroll
      for (int p = 0; p < NP; ++p)
#pragma unroll
        for (int K = 0; K < NT; ++K)
#pragma unroll
          for (int I = 0; I < NT; ++I) NpT[g][p][K][I] = mdl_tr(mem[g].mdl, 1 + p, K, I);
#pragma unroll
      for (int I = 0; I < NT; ++I) {
        pv[g][I] = 0.0;
#pragma unroll
        for (int J = 0; J < NT; ++J) P[g][I][J] = sym_nat(Qf, I, J);
      }
      ld_row(mem[g].xbm, 0, xb[g]);
#pragma unroll
      for (int p = 0; p <= NP; ++p) {
#pragma unroll
        for (int I = 0; I < NT; ++I) {
          double acc = 0.0;
#pragma unroll
          for (int K = 0; K < NT; ++K) acc = mm(mdl_tr(mem[g].mdl, p, K, I), xb[g][K], acc);
          tt[g][p][I] = acc;
        }
      }
    }
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

    auto step = [&](int t, const Ops (&cur)[G]) __attribute__((always_inline)) {
      double At[G][NT][NT], b[G][NU][NT], c[G][NT], W[G][NT], Y[G][NT], H[G][NT], G4[G];
      // A_t, B_t = [N_s x_g], c
#pragma unroll
      M4Q_G
#pragma unroll
        for (int I = 0; I < NT; ++I)
#pragma unroll
          for (int J = 0; J < NT; ++J) {
            double a = M[g][0][I][J];
#pragma unroll
            for (int p = 0; p < NP; ++p) a = fma(cur[g].ug[p], M[g][1 + p][I][J], a);
            At[g][I][J] = a;
          }
#pragma unroll
      for (int s = 0; s < NU; ++s)
#pragma unroll
        for (int I = 0; I < NT; ++I) {
#pragma unroll
          M4Q_G b[g][s][I] = 0.0;
#pragma unroll
          for (int K = 0; K < NT; ++K)
#pragma unroll
            M4Q_G {
              if constexpr (M4Q_T2_EXP & 1) b[g][s][I] = fma(NpT[g][s][K][I], cur[g].xg[K], b[g][s][I]);
              else b[g][s][I] = mm(NpT[g][s][K][I], cur[g].xg[K], b[g][s][I]);
            }
        }
#pragma unroll
      M4Q_G
#pragma unroll
        for (int K = 0; K < NT; ++K) {
          double a = tt[g][0][K] - xb[g][K];
#pragma unroll
          for (int p = 0; p < NP; ++p) a = fma(cur[g].ug[p], tt[g][1 + p][K], a);
#pragma unroll
          for (int s = 0; s < NU; ++s) a = fma(b[g][s][K], cur[g].ub[s] - cur[g].ug[s], a);
          c[g][K] = a;
          double w = mq[NU] * a;
#pragma unroll
          for (int s = 0; s < NU; ++s) w = fma(mq[s], b[g][s][K], w);
          W[g][K] = w;
        }
      // Y = P W + [0 | p | 0]
#pragma unroll
      for (int I = 0; I < NT; ++I) {
#pragma unroll
        M4Q_G Y[g][I] = mq[NU] * pv[g][I];
#pragma unroll
        for (int K = 0; K < NT; ++K)
#pragma unroll
          M4Q_G Y[g][I] = mm(P[g][K][I], W[g][K], Y[g][I]);
      }
      // G4 = W^T Y first (the m x m system is on the critical path), then H = Y^T A_t
#pragma unroll
      M4Q_G G4[g] = 0.0;
#pragma unroll
      for (int K = 0; K < NT; ++K)
#pragma unroll
        M4Q_G G4[g] = mm(W[g][K], Y[g][K], G4[g]);
      if constexpr (!(M4Q_T2_EXP & 4)) {
#pragma unroll
        M4Q_G mem[g].gb[L.r * 4 + L.q] = G4[g];
      }
#pragma unroll
      for (int J = 0; J < NT; ++J) {
#pragma unroll
        M4Q_G H[g][J] = 0.0;
#pragma unroll
        for (int K = 0; K < NT; ++K)
#pragma unroll
          M4Q_G H[g][J] = mm(Y[g][K], At[g][K][J], H[g][J]);
      }
      wave_sync();
      double cf[G][NU], kk[G][NU];
#pragma unroll
      M4Q_G {
        cplx gm[NU][NU], ginv[NU][NU];
        double h[NU];
        if constexpr (M4Q_T2_EXP & 4) {
#pragma unroll
          for (int s = 0; s < NU; ++s) {
#pragma unroll
            for (int l = 0; l < NU; ++l) ginv[s][l] = mk(s == l ? 0.5 + 1e-3 * G4[g] : 0.0, 0.0);
            h[s] = G4[g];
          }
        } else {
#pragma unroll
          for (int s = 0; s < NU; ++s) {
#pragma unroll
            for (int l = s; l < NU; ++l) gm[s][l] = mk(mem[g].gb[s * 4 + l] + Rm[s][l], 0.0);
            h[s] = mem[g].gb[s * 4 + NU];
          }
          herm_inverse<NU>(gm, ginv);
        }
#pragma unroll
        for (int s = 0; s < NU; ++s) {
          double cs = 0.0, ks = 0.0;
#pragma unroll
          for (int l = 0; l < NU; ++l) {
            cs = fma(-ginv[s][l].re, mr[l], cs);
            ks = fma(-ginv[s][l].re, h[l], ks);
          }
          cf[g][s] = cs;
          kk[g][s] = ks;
        }
      }
      wave_sync();
      double Kc[G][NU][NT], Kr[G][NU][NT];
#pragma unroll
      for (int s = 0; s < NU; ++s)
#pragma unroll
        for (int J = 0; J < NT; ++J)
#pragma unroll
          M4Q_G {
            if constexpr (M4Q_T2_EXP & 8) { Kc[g][s][J] = cf[g][s] * H[g][J]; Kr[g][s][J] = cf[g][s] - H[g][J]; }
            else {
              Kc[g][s][J] = mm(cf[g][s], H[g][J], 0.0);
              Kr[g][s][J] = mm(H[g][J], cf[g][s], 0.0);
            }
          }
      if (!(M4Q_T2_EXP & 2) && store_ok) {
        // gains [t][col][s]: the m entries of a column are one tuple; Kc is replicated over r, kk over the member's 16 lanes: every
        // lane stores (same bytes from the four r of a q) - no exec mask to set up
        const unsigned gt = (unsigned)t * (NS + 1) * NU;
#pragma unroll
        M4Q_G {
#pragma unroll
          for (int J = 0; J < NT; ++J) {
            double kc[NU];
#pragma unroll
            for (int s = 0; s < NU; ++s) kc[s] = Kc[g][s][J];
            if (NS % 4 == 0 || 4 * J + L.q < NS) stn<NU>(mem[g].gains, gt + (4 * J + L.q) * NU, kc);
          }
          stn<NU>(mem[g].gains, gt + NS * NU, kk[g]);
        }
      }
      double S[G][NT][NT], sv[G][NT], PS[G][NT][NT], w[G][NT];
#pragma unroll
      M4Q_G
#pragma unroll
        for (int I = 0; I < NT; ++I) {
          double a = c[g][I];
#pragma unroll
          for (int s = 0; s < NU; ++s) a = fma(b[g][s][I], kk[g][s], a);
          sv[g][I] = a;
#pragma unroll
          for (int J = 0; J < NT; ++J) {
            double e = At[g][I][J];
#pragma unroll
            for (int s = 0; s < NU; ++s) e = fma(b[g][s][I], Kc[g][s][J], e);
            S[g][I][J] = e;
          }
        }
#pragma unroll
      for (int I = 0; I < NT; ++I) {
#pragma unroll
        for (int J = 0; J < NT; ++J) {
#pragma unroll
          M4Q_G PS[g][I][J] = 0.0;
#pragma unroll
          for (int K = 0; K < NT; ++K)
#pragma unroll
            M4Q_G {
              if constexpr (M4Q_T2_EXP & 16) PS[g][I][J] = fma(P[g][K][I], S[g][K][J], PS[g][I][J]);
              else PS[g][I][J] = mm(P[g][K][I], S[g][K][J], PS[g][I][J]);
            }
        }
#pragma unroll
        M4Q_G w[g][I] = pv[g][I];
#pragma unroll
        for (int K = 0; K < NT; ++K)
#pragma unroll
          M4Q_G w[g][I] = mm(P[g][K][I], sv[g][K], w[g][I]);
      }
      double RK[G][NU][NT], Rk[G][NU];
#pragma unroll
      M4Q_G
#pragma unroll
        for (int s = 0; s < NU; ++s) {
          double a = 0.0;
#pragma unroll
          for (int l = 0; l < NU; ++l) a = fma(Rm[s][l], kk[g][l], a);
          Rk[g][s] = a;
#pragma unroll
          for (int J = 0; J < NT; ++J) {
            double e = 0.0;
#pragma unroll
            for (int l = 0; l < NU; ++l) e = fma(Rm[s][l], Kc[g][l][J], e);
            RK[g][s][J] = e;
          }
        }
#pragma unroll
      for (int I = 0; I < NT; ++I) {
        double pn[G];
#pragma unroll
        M4Q_G {
          double a = 0.0;
#pragma unroll
          for (int s = 0; s < NU; ++s) a = fma(Kr[g][s][I], Rk[g][s], a);
          pn[g] = a;
        }
#pragma unroll
        for (int K = 0; K < NT; ++K)
#pragma unroll
          M4Q_G pn[g] = mm(S[g][K][I], w[g][K], pn[g]);
#pragma unroll
        M4Q_G pv[g][I] = pn[g];
#pragma unroll
        for (int J = 0; J < NT; ++J) {
          double acc[G];
#pragma unroll
          M4Q_G {
            double e = Qt[I][J];
#pragma unroll
            for (int s = 0; s < NU; ++s) e = fma(Kr[g][s][I], RK[g][s][J], e);
            acc[g] = e;
          }
#pragma unroll
          for (int K = 0; K < NT; ++K)
#pragma unroll
            M4Q_G {
              if constexpr (M4Q_T2_EXP & 16) acc[g] = fma(S[g][K][I], PS[g][K][J], acc[g]);
              else acc[g] = mm(S[g][K][I], PS[g][K][J], acc[g]);
            }
#pragma unroll
          M4Q_G P[g][I][J] = acc[g];       // (P's old tiles are dead: every product that reads them has been issued above)
        }
      }
    };
    Ops ring[PF][G];
#pragma unroll
    for (int i = 0; i < PF; ++i)
#pragma unroll
      M4Q_G ring[i][g] = load(mem[g], T - 1 - i > 0 ? T - 1 - i : 0);
    int t = T - 1;
    for (; t >= PF - 1; t -= PF) {
#pragma unroll
      for (int i = 0; i < PF; ++i) {
        M4Q_NO_HOIST();
        step(t - i, ring[i]);
#pragma unroll
        M4Q_G ring[i][g] = load(mem[g], t - i - PF > 0 ? t - i - PF : 0);
      }
    }
#pragma unroll
    for (int i = 0; i < PF - 1; ++i)
      if (t - i >= 0) step(t - i, ring[i]);
  }
#undef M4Q_G
};

}  // namespace m4q
