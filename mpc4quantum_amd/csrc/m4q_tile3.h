// m4q_tile3.h - the backward Riccati sweep on fp64 matrix-core tiles with TIME-BATCHED operands.
//
// What round 4's measurements said about the tile sweep of round 3 (per-index operands; profiles/r04_tile_chain.txt):
//   * in isolation it is 18-21 % faster than the DPP sweep (151 against 185 SIMD-ns per member-index), two interleaved groups per
//     wavefront are NOT faster than one group in each of two wavefronts, and a tile wavefront does not suffer from a DPP wavefront
//     on its SIMD;
//   * inside the closed-loop kernel it was 8 % SLOWER than the DPP sweep: there its operands come from a workspace that other
//     phases of other wavefronts are hammering, the loads take longer than two horizon indices of tile arithmetic, and a deeper
//     ring of per-index operand sets costs more registers than the kernel has (97 spilled VGPRs at three indices ahead).
// So the operands are fetched per BLOCK of four horizon indices, as tiles whose column q holds time t0 - q:
//     XT[K]   [r][q] = x_g(t0 - q)[4K + r]          one 8-byte load per K
//     UG, UB  the controls / control targets of time t0 - q (the m-tuple, one 16-byte load each at m = 2)
// 4 loads per 4 indices instead of 16, six doubles per lane per block, fetched a whole block (4-7 indices) ahead.  The products
// N_s x_g of the four indices are ONE product per (s, I, K) - B_s[I] = sum_K mm(N_s^T[K][I], XT[K]) has column q = time t0 - q -
// instead of one per index (8 -> 2 MFMAs per index), and an index takes its own column, and its own controls, with quad-permute
// moves (v_mov_b32_dpp quad_perm:[j,j,j,j]: 32-bit VALU, not the fp64 pipe).
// Arithmetic, lane map and gains layout as m4q_tile.h (read its header first).  Order-1 libraries.
#pragma once
#include "m4q_tile.h"

namespace m4q {

// the value lane q = J of every quad holds, in all four lanes of the quad (two 32-bit DPP moves)
template <int J>
__device__ __forceinline__ double quad_bcast(double x) {
  constexpr int ctrl = J | (J << 2) | (J << 4) | (J << 6);
  const int lo = __builtin_amdgcn_mov_dpp(__double2loint(x), ctrl, 0xf, 0xf, true);
  const int hi = __builtin_amdgcn_mov_dpp(__double2hiint(x), ctrl, 0xf, 0xf, true);
  return __hiloint2double(hi, lo);
}

// a wave-uniform double into scalar registers
__device__ __forceinline__ double to_sgpr(double x) {
  return __hiloint2double(__builtin_amdgcn_readfirstlane(__double2hiint(x)), __builtin_amdgcn_readfirstlane(__double2loint(x)));
}

#ifndef M4Q_TILE_GB_BATCH
#define M4Q_TILE_GB_BATCH 1
#endif
#ifndef M4Q_TILE_CBLK
#define M4Q_TILE_CBLK 1               // the affine column c of a block's four indices formed once per block (bit-identical; config 3 29.8 -> 29.3 ms)
#endif
// PINNED: the sweep of the exact box-QP iteration (riccati_backward<PINNED> of m4q_mpc.h, same algebra): controls of the working
// set `stat` ([T][NU]: 0 free, +1 / -1 pinned at the upper / lower bound) are constants of their stage; the stored row of a pinned
// control is the affine form of its multiplier.
template <int NS, int NU, int ORDER, bool PINNED = false>
struct TileBackwardB {
  static constexpr int NT = (NS + 3) / 4;
  static constexpr int NP = PowTab<NU, ORDER>::NP;
  static_assert(ORDER == 1 && NP == NU, "time-batched tile sweep: order-1 libraries (monomial p is u_p)");
  static_assert(NU <= 3, "W = [B | c] must fit one column tile");
  TileGeo L;
  int T;
  const double* Q; const double* Qf; const double* R;   // LDS, shared
  const double* mdl;                                    // LDS [1+NP][NS][PITCH], this lane's member
  GView Xg, Ug, gains, xbm, ubm;                        // positioned on this lane's member
  volatile __attribute__((address_space(3))) double* gb;       // LDS, 16 doubles, this lane's member
  // PINNED: the working set of this lane's member, the box, and the first control's band (u_prev +- du: [0, NU) lo0, [3, 3 + NU) hi0
  // at gb + TILE_PIN_OFFSET - read at t = 0 only).
  GView stat;
  double sat = 0.0;
  // Registers.  The kernels around this sweep keep up to a hundred registers of their own across it, so what the sweep can hold in
  // scalar registers (R, the box, the lane masks of the column / row selectors) or re-read from LDS at every index (Q's tiles at
  // n = 4k, the band) it does, and the lane's addresses are made values of THIS sweep (see backward()).  With all of that in
  // vector registers the first pinned sweep had the compiler reload spilled addresses inside the horizon loop, where every
  // reload's s_waitcnt vmcnt(0) also waits for the operand loads issued a block ahead: config 3 exact 224 ms, against 208 on DPP
  // rows and 190 in this form; the clipped kernel lost its last 44 spilled registers, 32.07 -> 31.26 ms
  // (profiles/r04_ab_experiments.txt; the model's tiles from LDS as well: 32.0 and 194 ms - they stay in registers).

  __device__ __forceinline__ double mdl_nat(int p, int I, int J) const {
    const int i = 4 * I + L.r, k = 4 * J + L.q;
    const double v = mdl[ModelPitch<NS>::at(p, i < NS ? i : 0, k < NS ? k : 0)];
    return (i < NS && k < NS) ? v : 0.0;
  }
  __device__ __forceinline__ double mdl_tr(int p, int I, int J) const {
    const int i = 4 * J + L.q, k = 4 * I + L.r;
    const double v = mdl[ModelPitch<NS>::at(p, i < NS ? i : 0, k < NS ? k : 0)];
    return (i < NS && k < NS) ? v : 0.0;
  }
  __device__ __forceinline__ double sym_nat(const double* M, int I, int J) const {
    const int i = 4 * I + L.r, k = 4 * J + L.q;
    const double v = M[(i < NS ? i : 0) * NS + (k < NS ? k : 0)];
    return (i < NS && k < NS) ? v : 0.0;
  }

  struct Blk {                // operands of the four indices tb, tb - 1, tb - 2, tb - 3: this lane holds time tb - q
    double xt[NT];
    double ug[NU], ub[NU];
    double st[PINNED ? NU : 1];
  };
  __device__ __forceinline__ Blk load_blk(int tb) const {
    Blk b;
    int tq = tb - L.q;
    tq = tq < 0 ? 0 : (tq > T - 1 ? T - 1 : tq);
#pragma unroll
    for (int K = 0; K < NT; ++K) {
      const int e = 4 * K + L.r;
      const double x = Xg.ld<double>((unsigned)tq * NS + (e < NS ? e : 0));
      b.xt[K] = e < NS ? x : 0.0;
    }
    ldn<NU>(Ug, (unsigned)tq * NU, b.ug);
    ldn<NU>(ubm, (unsigned)tq * NU, b.ub);
    if constexpr (PINNED) ldn<NU>(stat, (unsigned)tq * NU, b.st);
    return b;
  }

  __device__ __forceinline__ void backward(bool store_ok) {
    {
      // the lane's addresses as values of THIS sweep: as the kernel-long values they are, the register allocator spills them and
      // reloads them at every use inside the horizon loop
      asm volatile("" : "+v"(Xg.off), "+v"(Ug.off), "+v"(gains.off), "+v"(ubm.off));
      if constexpr (PINNED) asm volatile("" : "+v"(stat.off));
      asm volatile("" : "+v"(L.q), "+v"(L.r));
      L.q &= 3; L.r &= 3;
      const __attribute__((address_space(3))) double* ml = (const __attribute__((address_space(3))) double*)mdl;
      const __attribute__((address_space(3))) double* ql = (const __attribute__((address_space(3))) double*)Q;
      asm volatile("" : "+v"(gb), "+v"(ml), "+v"(ql));
      mdl = (const double*)ml;
      Q = (const double*)ql;
    }
    constexpr bool QLDS = NS % 4 == 0;           // Q's tiles re-read from LDS at every index (no masking at n = 4k)
    double M[1 + NP][NT][NT], NpT[NP][NT][NT], P[NT][NT], pv[NT], xb[NT], tt[1 + NP][NT], Qt[QLDS ? 1 : NT][QLDS ? 1 : NT];
    const double* Qlane = Q + L.r * NS + L.q;
#pragma unroll
    for (int I = 0; I < NT; ++I)
#pragma unroll
      for (int J = 0; J < NT; ++J) {
        if constexpr (!QLDS) Qt[I][J] = sym_nat(Q, I, J);
        P[I][J] = sym_nat(Qf, I, J);
#pragma unroll
        for (int p = 0; p <= NP; ++p) M[p][I][J] = mdl_nat(p, I, J);
#pragma unroll
        for (int p = 0; p < NP; ++p) NpT[p][I][J] = mdl_tr(1 + p, I, J);
      }
#pragma unroll
    for (int K = 0; K < NT; ++K) {
      const int e = 4 * K + L.r;
      const double x = xbm.ld<double>(e < NS ? e : 0);
      xb[K] = e < NS ? x : 0.0;
      pv[K] = 0.0;
    }
#pragma unroll
    for (int p = 0; p <= NP; ++p)
#pragma unroll
      for (int I = 0; I < NT; ++I) {
        double acc = 0.0;
#pragma unroll
        for (int K = 0; K < NT; ++K) acc = mm(mdl_tr(p, K, I), xb[K], acc);
        tt[p][I] = acc;
      }
#pragma unroll
    for (int I = 0; I < NT; ++I) tt[0][I] -= xb[I];                  // A xbar - xbar_{t+1} (constant target)
    bool isq[NU + 1], isr[NU];                                        // column / row selectors: lane masks (scalar registers)
#pragma unroll
    for (int s = 0; s <= NU; ++s) isq[s] = L.q == s;
#pragma unroll
    for (int s = 0; s < NU; ++s) isr[s] = L.r == s;
    double Rm[NU][NU];
#pragma unroll
    for (int s = 0; s < NU; ++s)
#pragma unroll
      for (int l = 0; l < NU; ++l) Rm[s][l] = to_sgpr(R[s * NU + l]);
    const double satu = to_sgpr(sat);

    // one horizon index; ug / ub: its controls and control targets, b: rowrep(N_s x_g), all replicated over the member's lanes
    auto step = [&](int t, const double (&ug)[NU], const double (&ub)[NU], const double (&b)[NU][NT],
                    const double (&stv)[NU], const double (&cin)[NT]) __attribute__((always_inline)) {
      double At[NT][NT], c[NT], W[NT], Y[NT], H[NT];
#pragma unroll
      for (int I = 0; I < NT; ++I)
#pragma unroll
        for (int J = 0; J < NT; ++J) {
          double a = M[0][I][J];
#pragma unroll
          for (int p = 0; p < NP; ++p) a = fma(ug[p], M[1 + p][I][J], a);       // A_t = A + sum_p u_p N_p (linearize.py:43-48)
          At[I][J] = a;
        }
#pragma unroll
      for (int K = 0; K < NT; ++K) {
        double a;
        if constexpr (M4Q_TILE_CBLK) {
          a = cin[K];                                                           // (formed for the block's four indices at once: block())
        } else {
          a = tt[0][K];
#pragma unroll
          for (int p = 0; p < NP; ++p) a = fma(ug[p], tt[1 + p][K], a);
#pragma unroll
          for (int s = 0; s < NU; ++s) a = fma(b[s][K], ub[s] - ug[s], a);      // + B ubar + Delta, Delta = -B u_g
        }
        c[K] = a;
        double w = isq[NU] ? a : 0.0;                                           // W = [B | c | 0]
#pragma unroll
        for (int s = 0; s < NU; ++s) w = isq[s] ? b[s][K] : w;
        W[K] = w;
      }
#pragma unroll
      for (int I = 0; I < NT; ++I) {
        double acc = isq[NU] ? pv[I] : 0.0;
#pragma unroll
        for (int K = 0; K < NT; ++K) acc = mm(P[K][I], W[K], acc);              // Y = P W + [0 | p | 0]
        Y[I] = acc;
      }
      double G4 = 0.0;
#pragma unroll
      for (int K = 0; K < NT; ++K) G4 = mm(W[K], Y[K], G4);                     // W^T Y: B^H P B and B^H (P c + p)
      gb[L.r * 4 + L.q] = G4;
#pragma unroll
      for (int J = 0; J < NT; ++J) {
        double acc = 0.0;
#pragma unroll
        for (int K = 0; K < NT; ++K) acc = mm(Y[K], At[K][J], acc);            // Y^T A_t: rows 0..m-1 = B^H P A_t
        H[J] = acc;
      }
      wave_sync();
      cplx gm[NU][NU], ginv[NU][NU];
      double h[NU];
#if M4Q_TILE_GB_BATCH
      {
        // (all reads of the G / h tile issued before the first use: gb is volatile, and read-add-read-add made each of the
        //  NU (NU + 3) / 2 reads its own LDS round trip on the index's dependent chain)
        double gv[NU][NU], hv[NU];
#pragma unroll
        for (int s = 0; s < NU; ++s) {
#pragma unroll
          for (int l = s; l < NU; ++l) gv[s][l] = gb[s * 4 + l];
          hv[s] = gb[s * 4 + NU];
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int s = 0; s < NU; ++s) {
#pragma unroll
          for (int l = s; l < NU; ++l) gm[s][l] = mk(gv[s][l] + Rm[s][l], 0.0);
          h[s] = hv[s];
        }
      }
#else
#pragma unroll
      for (int s = 0; s < NU; ++s) {
#pragma unroll
        for (int l = s; l < NU; ++l) gm[s][l] = mk(gb[s * 4 + l] + Rm[s][l], 0.0);
        h[s] = gb[s * 4 + NU];
      }
#endif
      wave_sync();
      bool fix[NU];
      double dufix[NU], hraw[NU], Gf[NU][NU];
#pragma unroll
      for (int s = 0; s < NU; ++s) { fix[s] = false; dufix[s] = 0.0; hraw[s] = 0.0; }
      if constexpr (PINNED) {
        // controls pinned at a bound are constants of the stage: K row = [0 | du_fix]; the free ones respond to them:
        //   G_ff du_f = -(H_f dx + h_f + G_fp du_p)                                          (riccati_backward<PINNED>, m4q_mpc.h)
#pragma unroll
        for (int s = 0; s < NU; ++s) {
          double lo = -satu, hi = satu;
          if (t == 0) { lo = fmax(lo, gb[TILE_PIN_OFFSET + s]); hi = fmin(hi, gb[TILE_PIN_OFFSET + 3 + s]); }
          fix[s] = stv[s] != 0.0;
          dufix[s] = fix[s] ? (stv[s] > 0.0 ? hi : lo) - ub[s] : 0.0;
#pragma unroll
          for (int l = 0; l < NU; ++l) Gf[s][l] = s <= l ? gm[s][l].re : gm[l][s].re;
        }
#pragma unroll
        for (int s = 0; s < NU; ++s) {
#pragma unroll
          for (int l = 0; l < NU; ++l) h[s] = fma(Gf[s][l], dufix[l], h[s]);
          hraw[s] = h[s];
        }
#pragma unroll
        for (int s = 0; s < NU; ++s) {
#pragma unroll
          for (int l = s; l < NU; ++l)
            if (fix[s] || fix[l]) gm[s][l] = mk(s == l ? 1.0 : 0.0, 0.0);
          if (fix[s]) h[s] = 0.0;
        }
      }
      herm_inverse<NU>(gm, ginv);
      // cf[s]: the coefficients of H's rows in K_s (a tile: -G^-1[s][r] in every q; rows of pinned controls do not enter);
      // cst[s]: the same for the row that is STORED - K_s for a free control, for a pinned one the affine form of its multiplier,
      //   mu_s(dx) = [H_s + sum_{l free} G_sl K_l] dx + h_s + sum_l G_sl du_l
      double cf[NU], kk[NU], cst[NU], kst[NU];
#pragma unroll
      for (int s = 0; s < NU; ++s) {
        double cs = 0.0, ks = 0.0;
#pragma unroll
        for (int l = 0; l < NU; ++l) {
          cs = (isr[l] && !fix[l]) ? -ginv[s][l].re : cs;
          ks = fma(-ginv[s][l].re, h[l], ks);
        }
        cf[s] = cs;
        kk[s] = fix[s] ? dufix[s] : ks;
      }
#pragma unroll
      for (int s = 0; s < NU; ++s) {
        cst[s] = cf[s];
        kst[s] = kk[s];
        if constexpr (PINNED) {
          double m = 0.0, c0 = hraw[s];
#pragma unroll
          for (int l = 0; l < NU; ++l) {
            m = fma(fix[l] ? 0.0 : Gf[s][l], cf[l], m);
            c0 = fma(fix[l] ? 0.0 : Gf[s][l], kk[l], c0);           // (pinned l: G_sl du_fix_l is already inside hraw)
          }
          m = isr[s] ? m + 1.0 : m;             // (+ e_s: a 0/1 tile would be one more lane constant to keep or spill)
          cst[s] = fix[s] ? m : cf[s];
          kst[s] = fix[s] ? c0 : kk[s];
        }
      }
      double Kc[NU][NT], Kr[NU][NT], Ks[NU][NT];
#pragma unroll
      for (int s = 0; s < NU; ++s)
#pragma unroll
        for (int J = 0; J < NT; ++J) {
          Ks[s][J] = mm(cst[s], H[J], 0.0);                                     // the stored row: [4J + q] in every r
          Kc[s][J] = PINNED ? (fix[s] ? 0.0 : Ks[s][J]) : Ks[s][J];             // K_s[4J + q] in every r (a pinned control: 0)
          Kr[s][J] = mm(H[J], cf[s], 0.0);                                      // K_s[4J + r] in every q
        }
      if (store_ok) {
        // gains [t][col][s]: the m entries of a column are one tuple; Kc is replicated over r, kk over the member's 16 lanes: every
        // lane stores (the same bytes from the four r of a q) - no exec mask to set up
        const unsigned gt = (unsigned)t * (NS + 1) * NU;
#pragma unroll
        for (int J = 0; J < NT; ++J) {
          double kc[NU];
#pragma unroll
          for (int s = 0; s < NU; ++s) kc[s] = Ks[s][J];
          if (NS % 4 == 0 || 4 * J + L.q < NS) stn<NU>(gains, gt + (4 * J + L.q) * NU, kc);
        }
        stn<NU>(gains, gt + NS * NU, kst);
      }
      double S[NT][NT], sv[NT], PS[NT][NT], w[NT];
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
#pragma unroll
      for (int I = 0; I < NT; ++I) {
#pragma unroll
        for (int J = 0; J < NT; ++J) {
          double acc = 0.0;
#pragma unroll
          for (int K = 0; K < NT; ++K) acc = mm(P[K][I], S[K][J], acc);
          PS[I][J] = acc;
        }
        double acc = pv[I];
#pragma unroll
        for (int K = 0; K < NT; ++K) acc = mm(P[K][I], sv[K], acc);
        w[I] = acc;
      }
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
          RK[s][J] = e;
        }
      }
#pragma unroll
      for (int I = 0; I < NT; ++I) {
        double pn = 0.0;
#pragma unroll
        for (int s = 0; s < NU; ++s) pn = fma(Kr[s][I], Rk[s], pn);
#pragma unroll
        for (int K = 0; K < NT; ++K) pn = mm(S[K][I], w[K], pn);
        pv[I] = pn;
#pragma unroll
        for (int J = 0; J < NT; ++J) {
#if defined(M4Q_EXP) && (M4Q_EXP & 16)
          // timing-only (RESULTS WRONG): the lower off-diagonal tiles of the symmetric P are not computed - what a sweep that derived
          // them from the upper ones FOR FREE would save (the 4 x 4 transposes it would really need are not paid here)
          if (J < I) { P[I][J] = P[J][I]; continue; }
#endif
          double e;
          if constexpr (QLDS) e = Qlane[4 * I * NS + 4 * J];
          else e = Qt[I][J];
#pragma unroll
          for (int s = 0; s < NU; ++s) e = fma(Kr[s][I], RK[s][J], e);
#pragma unroll
          for (int K = 0; K < NT; ++K) e = mm(S[K][I], PS[K][J], e);
          P[I][J] = e;                    // (P's old tiles are dead: every product that reads them has been issued above)
        }
      }
    };
    // the four indices of a block: index tb - j takes column j of the block's tiles
    auto block = [&](int tb, int cnt, const Blk& cur) __attribute__((always_inline)) {
      double BT[NU][NT];
#pragma unroll
      for (int s = 0; s < NU; ++s)
#pragma unroll
        for (int I = 0; I < NT; ++I) {
          double acc = 0.0;
#pragma unroll
          for (int K = 0; K < NT; ++K) acc = mm(NpT[s][K][I], cur.xt[K], acc);   // [r][q] = (N_s x_g(tb - q))[4I + r]
          BT[s][I] = acc;
        }
      // the affine column c = A xbar - xbar+ + sum_p u_p N_p xbar + B (ubar - u_g) of the block's four indices as ONE tile per K (lane
      // q holds time tb - q, as the operands do): the same fused multiply-adds in the same order as per index - bit-identical - issued
      // once per block instead of once per index (M4Q_TILE_CBLK)
      double CT[NT];
      if constexpr (M4Q_TILE_CBLK) {
#pragma unroll
        for (int K = 0; K < NT; ++K) {
          double a = tt[0][K];
#pragma unroll
          for (int p = 0; p < NP; ++p) a = fma(cur.ug[p], tt[1 + p][K], a);
#pragma unroll
          for (int s = 0; s < NU; ++s) a = fma(BT[s][K], cur.ub[s] - cur.ug[s], a);
          CT[K] = a;
        }
      }
      static_for<0, 4>([&](auto jj) {
        constexpr int j = decltype(jj)::value;
        if (j < cnt) {
          double ug[NU], ub[NU], b[NU][NT], stv[NU], cin[NT];
#pragma unroll
          for (int s = 0; s < NU; ++s) {
            ug[s] = quad_bcast<j>(cur.ug[s]);
            ub[s] = (PINNED || !M4Q_TILE_CBLK) ? quad_bcast<j>(cur.ub[s]) : 0.0;
            stv[s] = PINNED ? quad_bcast<j>(cur.st[s]) : 0.0;
#pragma unroll
            for (int I = 0; I < NT; ++I) b[s][I] = quad_bcast<j>(BT[s][I]);
          }
#pragma unroll
          for (int K = 0; K < NT; ++K) cin[K] = M4Q_TILE_CBLK ? quad_bcast<j>(CT[K]) : 0.0;
          step(tb - j, ug, ub, b, stv, cin);
        }
      });
    };
    // the first block takes the (T - 1) % 4 + 1 top indices, so that every later block is a full one; operands are fetched one
    // block ahead
    int tb = T - 1;
    int cnt = ((T - 1) & 3) + 1;
    // (one block ahead, the next set copied into place: two sets swapping roles over two blocks per trip measured 32.35 against
    //  32.25 ms - more spills -, two blocks ahead 32.93: profiles/r04_ab_experiments.txt)
    Blk cur = load_blk(tb);
    while (tb >= 0) {
      M4Q_NO_HOIST();
      const Blk nxt = load_blk(tb - cnt);
      block(tb, cnt, cur);
      tb -= cnt;
      cnt = 4;
      cur = nxt;
    }
  }
};

}  // namespace m4q
