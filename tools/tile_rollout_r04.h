// tools/tile_rollout_r04.h - the forward rollout on fp64 matrix-core tiles with time-batched operands, as measured in round 4 and
// NOT kept (profiles/r04_ab_experiments.txt, experiment 16): config 3 31.00 -> 30.75 ms, config 5's share 102.5 -> 102.2, config 2
// 2.94 -> 3.34 ms.  99 instructions per horizon index for four members (10 MFMAs, 29 quad moves) against the DPP rollout's 127: the
// same SIMD time, and a longer dependent chain where the launch is latency bound (d = 2).  It was included from m4q_tile3.h and
// called from the TILE branch of mpc_kernel in place of rollout_forward (hand-over of x0, the first control's band, the shift flag
// through one more LDS block per member; results - first control, finiteness sum - back the same way); the GPU parity tests of the
// tile path passed with it.  Kept here as the record of that form; nothing includes it.
#pragma once
#include "../mpc4quantum_amd/csrc/m4q_tile3.h"

namespace m4q {

// ---------------------------------------------------------------------------------------------
// The forward rollout with clipping (rollout_forward<WANT_COST = false, TCF = true> of m4q_mpc.h, same arithmetic: lqr.py:67-79;
// optimize.py:41) on the same tiles, operands time-batched as in the sweep: column q of a block's tiles holds time t0 + q.
//   x, A_t x:    vectors indexed by r (x[K] = x[4K + r] in every q); A_t x = sum_K mm(A_t^T tile [K][I], x[K]) stays in that form
//   K_s . dx:    sum_K mm(dx[K], KR[s][K]) with KR[s][K] [r][q] = K_s(t0 + q)[4K + r]: column q of the result is the feedback of
//                time t0 + q in every row - index j takes column j with two quad moves; no gain tile is ever broadcast
//   B_t:         N_s x_g of the block's four indices is one product per (s, I), as in the sweep
//   stores:      x_{t+1} and u_t of the four indices are collected into tiles (column q = time) and stored once per block:
//                2 + 1 vector-memory instructions per FOUR indices where the DPP rollout issues 2 per index; 7 loads per block
// Round 3's tile rollout (per-index operands, masked per-index stores) took 2.3 times the DPP rollout's time; this form is what the
// round-4 measurements say a rollout on tiles has to look like.
// Per member, through LDS (fb, TILE_GB_DOUBLES doubles): in [0, NS) x0 | [8, 8 + NU) lo0 | [11, 11 + NU) hi0; out [14, 14 + NU) the
// first control | [17] sum |x|^2 + sum u^2 (finite iff every state and control is).
// ---------------------------------------------------------------------------------------------
constexpr int TILE_FB_LO = 8, TILE_FB_HI = 11, TILE_FB_U0 = 14, TILE_FB_CHK = 17;
static_assert(TILE_FB_CHK < TILE_GB_DOUBLES, "per-member rollout block");

template <int NS, int NU, int ORDER>
struct TileForwardB {
  static constexpr int NT = (NS + 3) / 4;
  static constexpr int NP = PowTab<NU, ORDER>::NP;
  static_assert(ORDER == 1 && NP == NU, "time-batched tile rollout: order-1 libraries");
  static_assert(NU <= 3 && NS <= 8, "per-member LDS block");
  TileGeo L;
  int T;
  const double* mdl;                                    // LDS [1+NP][NS][PITCH], this lane's member
  GView Xg, Ug, gains, xbm, ubm;                        // positioned on this lane's member
  GView Xd, Ud;                                         // destination: (X_o, U_o) or - shift - the next step's guess (X_g, U_g)
  int shift;                                            // this lane's member: 1 = the solution goes into the shifted guess
  double sat;
  volatile __attribute__((address_space(3))) double* fb;       // LDS, this lane's member

  __device__ __forceinline__ double mdl_tr(int p, int I, int J) const {
    const int i = 4 * J + L.q, k = 4 * I + L.r;
    const double v = mdl[ModelPitch<NS>::at(p, i < NS ? i : 0, k < NS ? k : 0)];
    return (i < NS && k < NS) ? v : 0.0;
  }
  struct Blk {                // operands of the four indices t0, t0 + 1, t0 + 2, t0 + 3: this lane holds time t0 + q
    double xt[NT];
    double ug[NU], ub[NU], kk[NU];
    double kr[NT][NU];
  };
  __device__ __forceinline__ Blk load_blk(int t0) const {
    Blk b;
    int tq = t0 + L.q;
    tq = tq > T - 1 ? T - 1 : tq;
    const unsigned gt = (unsigned)tq * (NS + 1) * NU;
#pragma unroll
    for (int K = 0; K < NT; ++K) {
      const int e = 4 * K + L.r;
      const double x = Xg.ld<double>((unsigned)tq * NS + (e < NS ? e : 0));
      b.xt[K] = e < NS ? x : 0.0;
      ldn<NU>(gains, gt + (e < NS ? e : 0) * NU, b.kr[K]);
      if (NS % 4 != 0) {
#pragma unroll
        for (int s = 0; s < NU; ++s) b.kr[K][s] = e < NS ? b.kr[K][s] : 0.0;
      }
    }
    ldn<NU>(Ug, (unsigned)tq * NU, b.ug);
    ldn<NU>(ubm, (unsigned)tq * NU, b.ub);
    ldn<NU>(gains, gt + NS * NU, b.kk);
    return b;
  }

  // store_ok: this lane's member is running (a member that is not computes along and stores nothing)
  __device__ __forceinline__ void forward(bool store_ok) {
    {
      // (lane addresses as values of this rollout: see TileBackwardB::backward)
      asm volatile("" : "+v"(Xg.off), "+v"(Ug.off), "+v"(gains.off), "+v"(ubm.off), "+v"(Xd.off), "+v"(Ud.off));
      asm volatile("" : "+v"(L.q), "+v"(L.r));
      L.q &= 3; L.r &= 3;
      const __attribute__((address_space(3))) double* ml = (const __attribute__((address_space(3))) double*)mdl;
      asm volatile("" : "+v"(fb), "+v"(ml));
      mdl = (const double*)ml;
    }
    double MT[1 + NP][NT][NT];                          // [p][K][I]: [r][q] = M_p[4I + q][4K + r]
#pragma unroll
    for (int p = 0; p <= NP; ++p)
#pragma unroll
      for (int K = 0; K < NT; ++K)
#pragma unroll
        for (int I = 0; I < NT; ++I) MT[p][K][I] = mdl_tr(p, K, I);
    bool isq[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) isq[j] = L.q == j;
    const double satu = to_sgpr(sat);
    const int xs_shift = 1 - shift, us_shift = -shift;
    double x[NT], xb[NT], cx[NT];
#pragma unroll
    for (int K = 0; K < NT; ++K) {
      const int e = 4 * K + L.r;
      const double v0 = fb[e < NS ? e : 0];
      const double vb = xbm.ld<double>(e < NS ? e : 0);
      x[K] = e < NS ? v0 : 0.0;
      xb[K] = e < NS ? vb : 0.0;
      cx[K] = 0.0;
      if (store_ok && shift == 0 && (NS % 4 == 0 || e < NS)) Xd.st<double>(e, x[K]);
    }
    double cu = 0.0, ulast[NU], ufirst[NU];
#pragma unroll
    for (int s = 0; s < NU; ++s) { ulast[s] = 0.0; ufirst[s] = 0.0; }

    auto block = [&](int t0, int cnt, const Blk& cur) __attribute__((always_inline)) {
      double BT[NU][NT];
#pragma unroll
      for (int s = 0; s < NU; ++s)
#pragma unroll
        for (int I = 0; I < NT; ++I) {
          double acc = 0.0;
#pragma unroll
          for (int K = 0; K < NT; ++K) acc = mm(MT[1 + s][K][I], cur.xt[K], acc);     // [r][q] = (N_s x_g(t0 + q))[4I + r]
          BT[s][I] = acc;
        }
      double xs[NT], us[NU];                                                          // column q: x_{t0+q+1}, u_{t0+q}
#pragma unroll
      for (int K = 0; K < NT; ++K) xs[K] = 0.0;
#pragma unroll
      for (int s = 0; s < NU; ++s) us[s] = 0.0;
      static_for<0, 4>([&](auto jj) {
        constexpr int j = decltype(jj)::value;
        if (j < cnt) {
          const int t = t0 + j;
          double ug[NU], ub[NU], kk[NU], b[NU][NT];
#pragma unroll
          for (int s = 0; s < NU; ++s) {
            ug[s] = quad_bcast<j>(cur.ug[s]);
            ub[s] = quad_bcast<j>(cur.ub[s]);
            kk[s] = quad_bcast<j>(cur.kk[s]);
#pragma unroll
            for (int I = 0; I < NT; ++I) b[s][I] = quad_bcast<j>(BT[s][I]);
          }
          double ax[NT], dx[NT], u[NU];
#pragma unroll
          for (int I = 0; I < NT; ++I) {
            double acc = 0.0;
#pragma unroll
            for (int K = 0; K < NT; ++K) {
              double a = MT[0][K][I];
#pragma unroll
              for (int p = 0; p < NP; ++p) a = fma(ug[p], MT[1 + p][K][I], a);         // A_t = A + sum_p u_p N_p (linearize.py:43-48)
              acc = mm(a, x[K], acc);                                                 // A_t x
            }
            ax[I] = acc;
          }
#pragma unroll
          for (int K = 0; K < NT; ++K) dx[K] = x[K] - xb[K];
#pragma unroll
          for (int s = 0; s < NU; ++s) {
            double acc = 0.0;
#pragma unroll
            for (int K = 0; K < NT; ++K) acc = mm(dx[K], cur.kr[K][s], acc);           // column q: K_s(t0 + q) . dx
            double uk = quad_bcast<j>(acc) + kk[s] + ub[s];                           // lqr.py:75
            double lo = -satu, hi = satu;
            if (t == 0) { lo = fmax(lo, fb[TILE_FB_LO + s]); hi = fmin(hi, fb[TILE_FB_HI + s]); }
            uk = fmin(fmax(uk, lo), hi);                                              // lqr.py:76
            u[s] = uk;
            cu = fma(uk, uk, cu);
            us[s] = isq[j] ? uk : us[s];
            ulast[s] = uk;
            if (t == 0) ufirst[s] = uk;
          }
#pragma unroll
          for (int K = 0; K < NT; ++K) {
            double xn = ax[K];
#pragma unroll
            for (int s = 0; s < NU; ++s) xn = fma(b[s][K], u[s] - ug[s], xn);         // A_t x + B u + Delta, Delta = -B u_g
            x[K] = xn;
            cx[K] = fma(xn, xn, cx[K]);
            xs[K] = isq[j] ? xn : xs[K];
          }
        }
      });
      // one store per tile and block: this lane's column is time t0 + q
      const int tq = t0 + L.q;
      if (store_ok && L.q < cnt) {
#pragma unroll
        for (int K = 0; K < NT; ++K)
          if (NS % 4 == 0 || 4 * K + L.r < NS) Xd.st<double>((unsigned)(tq + xs_shift) * NS + 4 * K + L.r, xs[K]);
        // (u is the same in the four r of a q: they store the same bytes.  A shifting member's u_0 has no slot)
        if (tq + us_shift >= 0) stn<NU>(Ud, (unsigned)(tq + us_shift) * NU, us);
      }
    };
    int t0 = 0;
    Blk cur = load_blk(0);
    while (t0 < T) {
      M4Q_NO_HOIST();
      const int cnt = T - t0 < 4 ? T - t0 : 4;
      const Blk nxt = load_blk(t0 + 4 < T ? t0 + 4 : T - 1);
      block(t0, cnt, cur);
      t0 += 4;
      cur = nxt;
    }
    if (store_ok && shift != 0) {                        // repeat the last column (mpc.py:271-272)
#pragma unroll
      for (int K = 0; K < NT; ++K)
        if (NS % 4 == 0 || 4 * K + L.r < NS) Xd.st<double>((unsigned)T * NS + 4 * K + L.r, x[K]);
      stn<NU>(Ud, (unsigned)(T - 1) * NU, ulast);
    }
    double tot = cu;
#pragma unroll
    for (int K = 0; K < NT; ++K) tot = mm(cx[K], 1.0, tot);                           // + sum_r cx[r]: padded entries are zero
#pragma unroll
    for (int s = 0; s < NU; ++s) fb[TILE_FB_U0 + s] = ufirst[s];
    fb[TILE_FB_CHK] = tot;
  }
};

}  // namespace m4q
