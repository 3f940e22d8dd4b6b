// m4q_mpc.h - per-instance MPC numerics on one 16-lane DPP row (see m4q_device.h for the layout).
//
// Reference arithmetic being replaced (citations into /root/reference/):
//   bilinear linearisation      mpc4quantum/linearize.py:34-70   (WrapModel.df_dx / df_du / Delta)
//   finite-horizon Riccati      mpc4quantum/lqr.py:28-79         (+ Delta and xbar_{t+1}: optimize.py:27-41,54)
//   line search                 mpc4quantum/mpc.py:101-125
//   plant step                  mpc4quantum/experiment.py:190-212 (ODE solved exactly: Pade-13 expm)
#pragma once
#include "m4q_device.h"

// Settled by A/B runs and no longer switches (each was a macro until round 4; the losing branches are gone from the source, their
// numbers are in profiles/r02_ab_experiments.txt, r03_exact_qp_log.txt, r04_ab_experiments.txt):
//   real path, n <= 9: the model's column and row forms read from LDS in one batch per horizon index (51.98 -> 50.38 ms), in the
//     rollout too (51.2 -> 50.85); with a constant target the N_p rows and column j of Q stay in registers over the sweep
//     (39.4 -> 38.25 ms) and the row form over the rollout (40.4 -> 39.4);
//   real path, n >= 15 (one wavefront per SIMD): the ROW form of the model and column j of Q in registers over a sweep
//     (92.9 -> 85.8, 72.2 -> 71.1 ms); both forms (87.5 ms, 209 spills) and a per-index batch (92.9) lost;
//   exact mode: the pinned sweep's constant-target form with the same hoisted reads (234 -> 222 ms), N_p rows in registers in the
//     adjoint pass (226 -> 221);
//   constant targets: A_t xbar from 1 + NP products formed once per sweep (recursions of dimension >= 8: 41.3 -> 40.3 ms);
//   scheduling barriers between the phases of a sweep: slower (53.3 -> 52.5 ms without); the two-index form of the COMPLEX sweep:
//     brings nothing and one experimental d = 4 build of it faulted - the complex sweep runs one index per trip.
#ifndef M4Q_MASK_IDLE
#define M4Q_MASK_IDLE 1      // lanes that own no column sit the two sweeps out (EXEC off): less power, higher clock (m4q_kernels.hip)
#endif
// Round 4: vector-memory instructions of the two sweeps (profiles/r04_ab_experiments.txt; every line an A/B on one box)
#ifndef M4Q_TC_XB_ONCE
#define M4Q_TC_XB_ONCE(n) ((n) < 15)  // constant-target backward sweep: xbar loaded once per sweep, not per index.  n = 15 (one wavefront
                                      // per SIMD): the sweep WITHOUT that load is slower, 73.0 against 71.2 ms on config 4 - kept per index there
#endif
#ifndef M4Q_TCF
#define M4Q_TCF(n) true               // constant-target instantiation of the rollout as well (xbar loaded once): config 3 35.77 -> 35.52 ms
#endif
#ifndef M4Q_SG_VFORM
#define M4Q_SG_VFORM 1              // shared-generator rollout: the step as one product of the row [A | N_1 .. N_m] (63.3 -> 62.5 ms at config 4)
#endif
#ifndef M4Q_STORE_ALL
// n >= 15: one wavefront per SIMD with 512 registers - loop-invariant operands of the exact mode's passes stay in registers
#ifndef M4Q_N15_HOIST
#define M4Q_N15_HOIST 1               // 0 (experiment builds): n >= 15 compiled for TWO wavefronts per SIMD - nothing held over a sweep
#endif
#define M4Q_XH15(n) (M4Q_N15_HOIST && (n) >= 15)
#define M4Q_STORE_ALL(n) true         // values replicated over a row (k, u) are stored by every lane of the row - no exec mask to set up -
                                      // instead of by lane 0: config 4 75.5 -> 73.0 ms, config 3 together with M4Q_TC_XB_ONCE 36.05 -> 35.6
#endif

namespace m4q {

// QP semantics flags (mirrored in include/m4q.h)
enum : int {
  QP_REF_LQR = 1,   // reproduce lqr.py as written (no Delta, xbar_{t+1}==xbar_t, cost built on xbar, absolute cost)
  QP_DU_BAND = 2,   // clip the first control to u_prev +- du as well (optimize.py:29-30)
  QP_EXACT_BOX = 4, // solve the box-constrained QP to optimality (projected Newton) instead of clipping the Riccati rollout
  QP_TARG_CONST = 256,   // internal (set by the host when every column of X_targ is the same): xbar_t does not depend on t
  QP_NO_TILE = 512,      // internal (M4Q_OPT_NO_TILE in the exact mode, whose kernel holds both forms of the pinned sweep)
};

// ---------------------------------------------------------------------------------------------
// Control-monomial table, in the order of linearize.create_power_list (linearize.py:92-116):
// the exponent of the LAST control is the outermost loop, total degree <= ORDER.
// ---------------------------------------------------------------------------------------------
constexpr int binom(int n, int k) {
  int r = 1;
  for (int i = 1; i <= k; ++i) r = r * (n - k + i) / i;
  return r;
}

template <int NU, int ORDER>
struct PowTab {
  static constexpr int COUNT = binom(ORDER + NU, NU);   // including the constant
  static constexpr int NP = COUNT - 1;
  int e[COUNT][NU > 0 ? NU : 1];
  constexpr PowTab() : e{} {
    // enumerate [0..ORDER]^NU with the first control as the fastest digit, keep total degree <= ORDER
    int total = 1;
    for (int k = 0; k < NU; ++k) total *= (ORDER + 1);
    int idx = 0;
    for (int code = 0; code < total; ++code) {
      int c = code, sum = 0;
      int cur[3] = {0, 0, 0};
      for (int k = 0; k < NU; ++k) {
        cur[k] = c % (ORDER + 1);
        c /= (ORDER + 1);
        sum += cur[k];
      }
      if (sum <= ORDER) {
        for (int k = 0; k < NU; ++k) e[idx][k] = cur[k];
        ++idx;
      }
    }
  }
};

template <int E>
__device__ __forceinline__ double ipow(double u) {
  if constexpr (E <= 0) return 1.0;
  else if constexpr (E == 1) return u;
  else return u * ipow<E - 1>(u);
}

// polyu_p(u) and d polyu_p / d u_k for the NP non-constant monomials
template <int NU, int ORDER>
struct Poly {
  static constexpr PowTab<NU, ORDER> tab{};
  static constexpr int NP = PowTab<NU, ORDER>::NP;
  double pu[NP];
  double dpu[NU][NP];
  __device__ __forceinline__ void eval(const double (&u)[NU]) {
    static_for<0, NP>([&](auto pp) {
      constexpr int p = decltype(pp)::value;
      double v = 1.0;
      static_for<0, NU>([&](auto kk) {
        constexpr int k = decltype(kk)::value;
        v *= ipow<tab.e[p + 1][k]>(u[k]);
      });
      pu[p] = v;
      static_for<0, NU>([&](auto kk) {
        constexpr int k = decltype(kk)::value;
        constexpr int ek = tab.e[p + 1][k];
        if constexpr (ek == 0) {
          dpu[k][p] = 0.0;
        } else {
          double d = (double)ek * ipow<ek - 1>(u[k]);
          static_for<0, NU>([&](auto ll) {
            constexpr int l = decltype(ll)::value;
            if constexpr (l != k) d *= ipow<tab.e[p + 1][l]>(u[l]);
          });
          dpu[k][p] = d;
        }
      });
    });
  }
};

// order 1: the library after the constant is u_1 .. u_m in this order (linearize.py:113-116)
template <int NU>
constexpr bool order1_is_identity() {
  constexpr PowTab<NU, 1> tab{};
  for (int p = 0; p < NU; ++p)
    for (int k = 0; k < NU; ++k)
      if (tab.e[p + 1][k] != (p == k ? 1 : 0)) return false;
  return true;
}

template <int NX>
struct ModelPitch {
  // Row pitch (in elements) of a model block in LDS.  Both access patterns must stay off a single bank:
  // column owners (lane j reads [r][j]) and row owners (lane r reads [r][k], lane stride = one row).
  // n = 4, 9: pitch n is conflict-free as is.  n = 16: a row is a multiple of the bank period; instead of
  // padding (17, which costs d=4 a workgroup per CU) the column is XOR-swizzled with the row.
  // n = 8 (the traceless coordinates of d = 3): a row of 8 doubles puts row owners 0 and 4 on one bank; padded to 9.
  static constexpr int value = NX == 8 ? 9 : NX;
  static __device__ __forceinline__ int at(int blk, int r, int k) {
    if constexpr (NX == 16) return (blk * NX + r) * NX + (k ^ r);
    else return (blk * NX + r) * value + k;
  }
};

// ---------------------------------------------------------------------------------------------
// Hermitian operator basis used by the real path.  Slot c = a*d + b of the n = d*d state holds
//   a == b : rho_aa            a < b : sqrt2 Re rho_ab  (symmetric part of the pair)
//   a > b  : sqrt2 Im rho_ab   (antisymmetric part of the pair (b, a))
// i.e. x = W r with x_ab = (r_ab - i r_ba)/sqrt2 (a < b), x_ab = (r_ba + i r_ab)/sqrt2 (a > b), x_aa = r_aa.
// W is unitary; a Liouvillian-generated model, Hermitian states and Hermitian costs are all real in it.
// Conversions go through a per-row LDS scratch of NX elements (one wave per workgroup: wave_sync() orders
// the exchange).
// ---------------------------------------------------------------------------------------------
template <int NX, int D>
__device__ __forceinline__ cplx basis_to_complex(double r, double* sc, int j, int jj) {
  const int a = j / D, b = j - (j / D) * D;
  if (jj < NX) sc[jj] = r;
  wave_sync();
  const double partner = sc[b * D + a];
  wave_sync();
  const double rs = 0.70710678118654752440;
  if (a == b) return mk(r, 0.0);
  return a < b ? mk(r * rs, -partner * rs) : mk(partner * rs, r * rs);
}
template <int NX, int D>
__device__ __forceinline__ double basis_to_real(cplx x, cplx* sc, int j, int jj) {
  const int a = j / D, b = j - (j / D) * D;
  if (jj < NX) sc[jj] = x;
  wave_sync();
  const cplx partner = sc[b * D + a];
  wave_sync();
  const double rs = 0.70710678118654752440;
  if (a == b) return x.re;
  return a < b ? (x.re + partner.re) * rs : (x.im - partner.im) * rs;
}
template <class S> struct BasisIO;
template <> struct BasisIO<cplx> {
  template <int NX, int D> static __device__ __forceinline__ cplx to_state(cplx x, cplx*, int, int) { return x; }
  template <int NX, int D> static __device__ __forceinline__ cplx to_complex(cplx x, cplx*, int, int) { return x; }
};
template <> struct BasisIO<double> {
  template <int NX, int D> static __device__ __forceinline__ double to_state(cplx x, cplx* sc, int j, int jj) {
    return basis_to_real<NX, D>(x, sc, j, jj);
  }
  template <int NX, int D> static __device__ __forceinline__ cplx to_complex(double r, cplx* sc, int j, int jj) {
    return basis_to_complex<NX, D>(r, reinterpret_cast<double*>(sc), j, jj);
  }
};

// ---------------------------------------------------------------------------------------------
// Traceless Hermitian basis (path 2).  A trace-preserving, unital model - every Taylor-truncated Liouvillian - leaves the
// identity component of rho invariant and decoupled from the rest, so the recursion can run on the n_s = d*d - 1 traceless
// coordinates (the cost's cross term with the constant component vanishes when state and target have equal trace; the host
// checks all of it, m4q_capi.hip).  The diagonal slots (a, a) of the Hermitian basis above are rotated by the orthogonal
//   O[a][0] = 1/sqrt(d);   O[a][l] = 1/sqrt(l(l+1)) (a < l),  -l/sqrt(l(l+1)) (a == l),  0 (a > l)       l = 1 .. d-1
// (generalised Gell-Mann diagonal matrices); slot (0, 0) becomes the trace coordinate tau = tr(rho)/sqrt(d) and is dropped:
// traceless coordinate s = slot c - 1.  Lane conventions as above: complex data one slot per lane (jj < NX), traceless
// data one coordinate per lane (jj < NX - 1).
// ---------------------------------------------------------------------------------------------
template <int D>
__device__ __forceinline__ double tl_coef(int a, int l) {      // O[a][l], l >= 1
  const double s = 1.0 / sqrt((double)(l * (l + 1)));
  return a < l ? s : (a == l ? -(double)l * s : 0.0);
}
template <int NX, int D>
__device__ __forceinline__ double tl_to_state(cplx x, cplx* sc, int jj, double& tau) {
  if (jj < NX) sc[jj] = x;
  wave_sync();
  const int s = jj < NX - 1 ? jj : NX - 2;
  const int c = s + 1, a = c / D, b = c - (c / D) * D;
  const double rs = 0.70710678118654752440;
  double t = 0.0, diag = 0.0;
#pragma unroll
  for (int k = 0; k < D; ++k) {
    const double xk = sc[k * D + k].re;
    t += xk;
    diag = fma(tl_coef<D>(k, a > 0 ? a : 1), xk, diag);        // (used when a == b: l = a >= 1)
  }
  tau = t * (1.0 / sqrt((double)D));
  const cplx own = sc[c], partner = sc[b * D + a];
  wave_sync();
  if (a == b) return diag;
  return a < b ? (own.re + partner.re) * rs : (own.im - partner.im) * rs;
}
template <int NX, int D>
__device__ __forceinline__ cplx tl_to_complex(double r, double tau, cplx* scc, int jj) {
  double* sc = reinterpret_cast<double*>(scc);
  if (jj < NX - 1) sc[jj] = r;
  wave_sync();
  const int c = jj < NX ? jj : NX - 1, a = c / D, b = c - (c / D) * D;
  const double rs = 0.70710678118654752440;
  cplx out;
  if (a == b) {
    double v = tau * (1.0 / sqrt((double)D));
#pragma unroll
    for (int l = 1; l < D; ++l) v = fma(tl_coef<D>(a, l), sc[l * D + l - 1], v);
    out = mk(v, 0.0);
  } else {
    const double own = sc[c - 1], partner = sc[b * D + a - 1];
    out = a < b ? mk(own * rs, -partner * rs) : mk(partner * rs, own * rs);
  }
  wave_sync();
  return out;
}

// tl_to_complex of one whole trajectory node by ONE lane: r[0 .. NX-2] are the node's traceless coordinates, out[c] the NX slots of
// vec(rho).  Same arithmetic per slot as tl_to_complex (same bits); every index is a compile-time constant.
template <int NX, int D>
__device__ __forceinline__ void tl_node_to_complex(const double (&r)[NX - 1], double tau, cplx (&out)[NX]) {
  const double rs = 0.70710678118654752440;
#pragma unroll
  for (int c = 0; c < NX; ++c) {
    const int a = c / D, b = c - (c / D) * D;
    if (a == b) {
      double v = tau * (1.0 / sqrt((double)D));
#pragma unroll
      for (int l = 1; l < D; ++l) v = fma(tl_coef<D>(a, l), r[l * D + l - 1], v);
      out[c] = mk(v, 0.0);
    } else {
      const double own = r[c - 1], partner = r[b * D + a - 1];
      out[c] = a < b ? mk(own * rs, -partner * rs) : mk(partner * rs, own * rs);
    }
  }
}

// What the closed-loop kernel calls: S, and whether the state lives in the traceless coordinates (TL).  `tau` is the member's
// trace coordinate: written by to_state, read by to_complex (ignored unless TL).
template <class S, bool TL> struct Basis {
  template <int NX, int D> static __device__ __forceinline__ S to_state(cplx x, cplx* sc, int j, int jj, double&) {
    return BasisIO<S>::template to_state<NX, D>(x, sc, j, jj);
  }
  template <int NX, int D> static __device__ __forceinline__ cplx to_complex(S r, double, cplx* sc, int j, int jj) {
    return BasisIO<S>::template to_complex<NX, D>(r, sc, j, jj);
  }
};
template <> struct Basis<double, true> {
  template <int NX, int D> static __device__ __forceinline__ double to_state(cplx x, cplx* sc, int, int jj, double& tau) {
    return tl_to_state<NX, D>(x, sc, jj, tau);
  }
  template <int NX, int D> static __device__ __forceinline__ cplx to_complex(double r, double tau, cplx* sc, int, int jj) {
    return tl_to_complex<NX, D>(r, tau, sc, jj);
  }
};

// ---------------------------------------------------------------------------------------------
// Linearisation providers.  fetch(t): the per-t operands that come from memory (issued one horizon
// index ahead of their use so the loads overlap the previous index's arithmetic).  col(): column j of
// A_t.  rows(): for the row this lane owns, (A_t v)_j for a distributed vector v, row j of B_t and
// Delta_t[j].  Views are positioned on this lane's instance; lane-dependent parts are 32-bit offsets.
// ---------------------------------------------------------------------------------------------
// Do the batched model reads fit the register file at two wavefronts per SIMD?  Both forms of (1 + NP) n x n blocks are 2 (1 + NP) n
// doubles per lane: 54 at (n, m, order) = (9, 2, 1); at order 2 (NP = 5) the 108 doubles spill inside the horizon loop
// (tools/hot_loops.py: 39 scratch loads per trip), so those shapes keep the interleaved reads.
template <int NX, int NU, int ORDER>
constexpr bool batch_fits() { return NX <= 9 && (1 + PowTab<NU, ORDER>::NP) * NX <= 27; }

// one model element from LDS
template <class S>
__device__ __forceinline__ S mld(const S* mdl, int idx) { return mdl[idx]; }

// Real path, backward sweep: the column form (for A_t's column) and the row form (for A_t xbar and row j of B_t) of the model
// are read from LDS in ONE batch at the top of the horizon index - at that point only P and the prefetched operands are live,
// so the 2 (1 + NP) NX doubles fit - instead of in small groups each followed by its own wait: a wavefront alone on its SIMD
// spent a quarter of every horizon index in a dozen serialised LDS read-to-use latencies (profiles/README.md).
template <class S, int NX, int NU, int ORDER>
struct ModelRegs {
  static constexpr int NP = PowTab<NU, ORDER>::NP;
  S col[1 + NP][NX];     // col[p][i] = block p, element [i][j]
  S row[1 + NP][NX];     // row[p][k] = block p, element [j][k]
  // (P: a FusedProv - its el(p, r, k) is element [r][k] of block p wherever that block lives)
  template <class P>
  __device__ __forceinline__ void load(const P& pv, int j) {
#pragma unroll
    for (int p = 0; p <= NP; ++p) {
#pragma unroll
      for (int i = 0; i < NX; ++i) col[p][i] = pv.el(p, i, j);
    }
    load_rows(pv, j);
  }
  template <class P>
  __device__ __forceinline__ void load_rows(const P& pv, int j) {
#pragma unroll
    for (int p = 0; p <= NP; ++p) {
#pragma unroll
      for (int k = 0; k < NX; ++k) row[p][k] = pv.el(p, j, k);
    }
  }
};

// SG (order-1 ensembles built from SHARED generators and per-member scales, m4q_session_build_models): member i's model is
//   [I + dt s_i0 L_0 | dt s_i1 L_1 | ... | dt s_im L_m]                       (vectorize.py:8-49 at order 1; tests/test_mpc4quantum.py:147-188)
// i.e. (1 + m) matrices common to every member and (1 + m) numbers of its own.  The workgroup then holds ONE copy of N_k = dt L_k
// (`mdn`) and per member only A_i = I + s_i0 dt L_0 (`mdl`); the scales ride on the controls: with u~_k = s_ik u_k
//   A_t = A_i + sum_k u~_k N_k,   B_t[:, k] = s_ik (N_k xg),   Delta_t = -sum_k (N_k xg) u~_k
// - 12.6 instead of 28.8 KB of LDS per workgroup at n = 15, which is what lets d = 4 run two wavefronts per SIMD.
template <class S, int NX, int NU, int ORDER, bool SG = false>
struct FusedProv {
  static constexpr int ORDER_ = ORDER;
  static constexpr bool SG_ = SG;
  static constexpr int NP = PowTab<NU, ORDER>::NP;
  static constexpr int PITCH = ModelPitch<NX>::value;
  static_assert(!SG || (ORDER == 1 && sizeof(S) == sizeof(double)), "shared generators: order-1 real paths");
  const S* mdl;       // LDS, [1+NP][NX][PITCH]: block 0 = A, block 1+p = N_p   (model.py:95-103).  SG: this member's A alone
  const S* mdn = nullptr;      // SG: LDS, [NP][NX][PITCH], the workgroup's shared N_p
  double sc[SG ? NU : 1];     // SG: this member's scales of the control operators (row-uniform)
  GView Xg;           // guess trajectory [T+1][NX], positioned at element 0 of this instance
  GView Ug;           // [T][NU]
  int j;              // lane in row, clamped to NX-1

  // element [r][k] of block p (p is a compile-time constant at every call site once the loops are unrolled)
  __device__ __forceinline__ S el(int p, int r, int k) const {
    if constexpr (SG) return p == 0 ? mld(mdl, ModelPitch<NX>::at(0, r, k)) : mld(mdn, ModelPitch<NX>::at(p - 1, r, k));
    else return mld(mdl, ModelPitch<NX>::at(p, r, k));
  }
  struct Lin {
    double u[NU];     // (SG: the scaled controls u~)
    S xg;
  };
  __device__ __forceinline__ Lin fetch(int t) const {
    Lin l;
    ldn<NU>(Ug, t * NU, l.u);
    if constexpr (SG) {
#pragma unroll
      for (int k = 0; k < NU; ++k) l.u[k] *= sc[k];
    }
    l.xg = Xg.ld<S>(t * NX + j);
    return l;
  }
  // A_t = A + sum_p polyu_p N_p   (linearize.py:43-48)
  __device__ __forceinline__ void col(const Lin& l, S (&Ac)[NX]) const {
    Poly<NU, ORDER> po;
    po.eval(l.u);
#pragma unroll
    for (int i = 0; i < NX; ++i) {
      S a = el(0, i, j);
#pragma unroll
      for (int p = 0; p < NP; ++p) cmac_r(a, el(1 + p, i, j), po.pu[p]);
      Ac[i] = a;
    }
  }
  // col() with its (1 + NP) NX LDS reads issued as one batch
  __device__ __forceinline__ void col_batch(const Lin& l, S (&Ac)[NX]) const {
    Poly<NU, ORDER> po;
    po.eval(l.u);
    S col[1 + NP][NX];
#pragma unroll
    for (int p = 0; p <= NP; ++p) {
#pragma unroll
      for (int i = 0; i < NX; ++i) col[p][i] = el(p, i, j);
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int i = 0; i < NX; ++i) {
      S a = col[0][i];
#pragma unroll
      for (int p = 0; p < NP; ++p) cmac_r(a, col[1 + p][i], po.pu[p]);
      Ac[i] = a;
    }
  }
  // col() and rows() from registers loaded in one batch (real path of the backward sweep)
  __device__ __forceinline__ void col_rows(const Lin& l, S v, S (&Ac)[NX], S& av, S (&Brow)[NU], S& dlt) const {
    ModelRegs<S, NX, NU, ORDER> r;
    r.load(*this, j);
    __builtin_amdgcn_sched_barrier(0);
    col_rows(r, l, v, Ac, av, Brow, dlt);
  }
  // ... from registers the caller holds (n = 16 at one wavefront per SIMD: the whole model stays in registers over a sweep)
  __device__ __forceinline__ void col_rows(const ModelRegs<S, NX, NU, ORDER>& r, const Lin& l, S v, S (&Ac)[NX], S& av, S (&Brow)[NU],
                                           S& dlt) const {
    Poly<NU, ORDER> po;
    po.eval(l.u);
#pragma unroll
    for (int i = 0; i < NX; ++i) {
      S a = r.col[0][i];
#pragma unroll
      for (int p = 0; p < NP; ++p) cmac_r(a, r.col[1 + p][i], po.pu[p]);
      Ac[i] = a;
    }
    rows_from(r, l, po, v, av, Brow, dlt);
  }
  __device__ __forceinline__ void rows_from(const ModelRegs<S, NX, NU, ORDER>& r, const Lin& l, const Poly<NU, ORDER>& po, S v, S& av,
                                            S (&Brow)[NU], S& dlt) const {
    S nx[NP], arow[NX];
#pragma unroll
    for (int k = 0; k < NX; ++k) arow[k] = r.row[0][k];
#pragma unroll
    for (int p = 0; p < NP; ++p) {
#pragma unroll
      for (int k = 0; k < NX; ++k) cmac_r(arow[k], r.row[1 + p][k], po.pu[p]);
      nx[p] = dot_lane_index<false, false, NX>(l.xg, r.row[1 + p]);
    }
    av = dot_lane_index<false, false, NX>(v, arow);
    finish_rows(l, po, nx, Brow, dlt);
  }
  __device__ __forceinline__ void rows(const ModelRegs<S, NX, NU, ORDER>& r, const Lin& l, S v, S& av, S (&Brow)[NU], S& dlt) const {
    Poly<NU, ORDER> po;
    po.eval(l.u);
    rows_from(r, l, po, v, av, Brow, dlt);
  }
  // Constant target (xbar_t = xbar for the whole window - every reference scenario but one): A_t xbar = A xbar + sum_p polyu_p
  // (N_p xbar), so the 1 + NP products with xbar are formed ONCE per sweep (tt) and the row form of A_t - NX loads, NP NX FMAs to
  // build it, an NX-term dot - is not needed in the sweep at all.
  __device__ __forceinline__ void target_terms(S xbar, S (&tt)[1 + NP]) const {
#pragma unroll
    for (int p = 0; p <= NP; ++p) {
      S row[NX];
#pragma unroll
      for (int k = 0; k < NX; ++k) row[k] = el(p, j, k);
      tt[p] = dot_lane_index<false, false, NX>(xbar, row);
    }
  }
  __device__ __forceinline__ S av_from_terms(const Poly<NU, ORDER>& po, const S (&tt)[1 + NP]) const {
    S av = tt[0];
#pragma unroll
    for (int p = 0; p < NP; ++p) cmac_r(av, tt[1 + p], po.pu[p]);
    return av;
  }
  __device__ __forceinline__ void col_rows_tc(const Lin& l, const S (&tt)[1 + NP], S (&Ac)[NX], S& av, S (&Brow)[NU], S& dlt) const {
    Poly<NU, ORDER> po;
    po.eval(l.u);
    S col[1 + NP][NX], row[NP][NX];
#pragma unroll
    for (int p = 0; p <= NP; ++p) {
#pragma unroll
      for (int i = 0; i < NX; ++i) col[p][i] = el(p, i, j);
    }
#pragma unroll
    for (int p = 0; p < NP; ++p) {
#pragma unroll
      for (int k = 0; k < NX; ++k) row[p][k] = el(1 + p, j, k);
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int i = 0; i < NX; ++i) {
      S a = col[0][i];
#pragma unroll
      for (int p = 0; p < NP; ++p) cmac_r(a, col[1 + p][i], po.pu[p]);
      Ac[i] = a;
    }
    S nx[NP];
#pragma unroll
    for (int p = 0; p < NP; ++p) nx[p] = dot_lane_index<false, false, NX>(l.xg, row[p]);
    av = av_from_terms(po, tt);
    finish_rows(l, po, nx, Brow, dlt);
  }
  __device__ __forceinline__ void rows_tc(const ModelRegs<S, NX, NU, ORDER>& r, const Lin& l, const S (&tt)[1 + NP], S& av, S (&Brow)[NU],
                                          S& dlt) const {
    Poly<NU, ORDER> po;
    po.eval(l.u);
    S nx[NP];
#pragma unroll
    for (int p = 0; p < NP; ++p) nx[p] = dot_lane_index<false, false, NX>(l.xg, r.row[1 + p]);
    av = av_from_terms(po, tt);
    finish_rows(l, po, nx, Brow, dlt);
  }
  // ... with the N_p rows read from LDS one at a time (n = 15 at two wavefronts per SIMD: neither a batch nor a hoist fits)
  __device__ __forceinline__ void rows_tc_lds(const Lin& l, const S (&tt)[1 + NP], S& av, S (&Brow)[NU], S& dlt) const {
    Poly<NU, ORDER> po;
    po.eval(l.u);
    S nx[NP];
#pragma unroll
    for (int p = 0; p < NP; ++p) {
      S row[NX];
#pragma unroll
      for (int k = 0; k < NX; ++k) row[k] = el(1 + p, j, k);
      nx[p] = dot_lane_index<false, false, NX>(l.xg, row);
    }
    av = av_from_terms(po, tt);
    finish_rows(l, po, nx, Brow, dlt);
  }
  __device__ __forceinline__ void finish_rows(const Lin& l, const Poly<NU, ORDER>& po, const S (&nx)[NP], S (&Brow)[NU], S& dlt) const {
    dlt = zero_of<S>();
#pragma unroll
    for (int k = 0; k < NU; ++k) {
      S b = zero_of<S>();
      if constexpr (ORDER == 1) {
        b = nx[k];
      } else {
#pragma unroll
        for (int p = 0; p < NP; ++p) cmac_r(b, nx[p], po.dpu[k][p]);
      }
      cmac_r(dlt, b, -l.u[k]);
      if constexpr (SG) b = cscale(b, sc[k]);
      Brow[k] = b;
    }
  }
  // row j of B_t alone (the adjoint pass needs neither A_t v nor Delta_t)
  __device__ __forceinline__ void brows(const Lin& l, S (&Brow)[NU]) const {
    if constexpr (ORDER == 1) {
#pragma unroll
      for (int k = 0; k < NU; ++k) {
        S row[NX];
#pragma unroll
        for (int c = 0; c < NX; ++c) row[c] = el(1 + k, j, c);
        Brow[k] = dot_lane_index<false, false, NX>(l.xg, row);          // monomial p is u_p (order1_is_identity)
        if constexpr (SG) Brow[k] = cscale(Brow[k], sc[k]);
      }
    } else {
      S av, dlt;
      rows(l, zero_of<S>(), av, Brow, dlt);
    }
  }
  // B_t[:,k] = sum_p (N_p x) c_kp dmono_kp(u)  (linearize.py:50-59);  Delta_t = f - A_t x - B_t u = -B_t u (:68-69)
  __device__ __forceinline__ void rows(const Lin& l, S v, S& av, S (&Brow)[NU], S& dlt) const {
    Poly<NU, ORDER> po;
    po.eval(l.u);
    S nx[NP];
    if constexpr (sizeof(S) == sizeof(double) && batch_fits<NX, NU, ORDER>()) {
      // as below with the (1 + NP) NX row elements read from LDS in one batch
      S row[1 + NP][NX];
#pragma unroll
      for (int p = 0; p <= NP; ++p) {
#pragma unroll
        for (int k = 0; k < NX; ++k) row[p][k] = el(p, j, k);
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int p = 0; p < NP; ++p) {
#pragma unroll
        for (int k = 0; k < NX; ++k) cmac_r(row[0][k], row[1 + p][k], po.pu[p]);
        nx[p] = dot_lane_index<false, false, NX>(l.xg, row[1 + p]);
      }
      av = dot_lane_index<false, false, NX>(v, row[0]);
    } else if constexpr (sizeof(S) == sizeof(double)) {
      // real path: whole rows in registers, one long dot per accumulator (8 FMAs per statement)
      //   av = sum_k lane_k(v) A_t[j][k],   nx[p] = sum_k lane_k(xg) N_p[j][k]
      S arow[NX];
#pragma unroll
      for (int k = 0; k < NX; ++k) arow[k] = el(0, j, k);
#pragma unroll
      for (int p = 0; p < NP; ++p) {
        S nrow[NX];
#pragma unroll
        for (int k = 0; k < NX; ++k) {
          nrow[k] = el(1 + p, j, k);
          cmac_r(arow[k], nrow[k], po.pu[p]);
        }
        nx[p] = dot_lane_index<false, false, NX>(l.xg, nrow);
      }
      av = dot_lane_index<false, false, NX>(v, arow);
    } else {
      // complex path (register bound): one statement per column k with the NP+1 accumulators
      S accs[NP + 1], srcs[NP + 1];
#pragma unroll
      for (int p = 0; p <= NP; ++p) { accs[p] = zero_of<S>(); srcs[p] = p < NP ? l.xg : v; }
      static_for<0, NX>([&](auto kk) {
        constexpr int k = decltype(kk)::value;
        S own[NP + 1];
        own[NP] = el(0, j, k);
#pragma unroll
        for (int p = 0; p < NP; ++p) {
          own[p] = el(1 + p, j, k);
          cmac_r(own[NP], own[p], po.pu[p]);
        }
        emit_terms<IdxSameLaneVec<k>, false, false, false, 0, NP + 1>(accs, srcs, own);
      });
#pragma unroll
      for (int p = 0; p < NP; ++p) nx[p] = accs[p];
      av = accs[NP];
    }
    dlt = zero_of<S>();
#pragma unroll
    for (int k = 0; k < NU; ++k) {
      S b = zero_of<S>();
      if constexpr (ORDER == 1) {
        static_assert(ORDER != 1 || order1_is_identity<NU>(), "order-1 library must list u_1 .. u_m in order");
        b = nx[k];                           // monomial p is u_p: d(mono_p)/du_k = delta_pk
      } else {
#pragma unroll
        for (int p = 0; p < NP; ++p) cmac_r(b, nx[p], po.dpu[k][p]);
      }
      cmac_r(dlt, b, -l.u[k]);
      if constexpr (SG) b = cscale(b, sc[k]);
      Brow[k] = b;
    }
  }
};
// is this provider a FusedProv, and of which kind
template <class P> struct fused_kind { static constexpr bool any = false, sg = false; };
template <class S, int NX, int NU, int ORDER, bool SG> struct fused_kind<FusedProv<S, NX, NU, ORDER, SG>> {
  static constexpr bool any = true, sg = SG;
};

template <int NX, int NU>
struct ExplicitProv {
  static constexpr int ORDER_ = 1;   // (the QP kernels do not depend on the library order)
  GView A_ls;   // [T][NX][NX]   positioned at this instance
  GView B_ls;   // [T][NX][NU]
  GView D_ls;   // [T][NX]
  bool has_delta;
  int j;
  struct Lin {
    int t;
  };
  __device__ __forceinline__ Lin fetch(int t) const { Lin l; l.t = t; return l; }
  __device__ __forceinline__ void col(const Lin& l, cplx (&Ac)[NX]) const {
#pragma unroll
    for (int i = 0; i < NX; ++i) Ac[i] = A_ls.ld<cplx>((l.t * NX + i) * NX + j);
  }
  __device__ __forceinline__ void col_rows(const Lin& l, cplx v, cplx (&Ac)[NX], cplx& av, cplx (&Brow)[NU], cplx& dlt) const {
    col(l, Ac);
    rows(l, v, av, Brow, dlt);
  }
  __device__ __forceinline__ void brows(const Lin& l, cplx (&Brow)[NU]) const {
#pragma unroll
    for (int k = 0; k < NU; ++k) Brow[k] = B_ls.ld<cplx>((l.t * NX + j) * NU + k);
  }
  __device__ __forceinline__ void rows(const Lin& l, cplx v, cplx& av, cplx (&Brow)[NU], cplx& dlt) const {
    cplx arow[NX];
#pragma unroll
    for (int k = 0; k < NX; ++k) arow[k] = A_ls.ld<cplx>((l.t * NX + j) * NX + k);
    av = dot_lane_index<false, false, NX>(v, arow);
#pragma unroll
    for (int k = 0; k < NU; ++k) Brow[k] = B_ls.ld<cplx>((l.t * NX + j) * NU + k);
    dlt = has_delta ? D_ls.ld<cplx>(l.t * NX + j) : czero();
  }
};

// Stage costs (shared by the ensemble, wave-uniform pointers; LDS in the fused kernel).
// Q(t) for t < T, Qf at t == T; R(t).
// TR: transposed copies of Q and Qf are staged as well (QT, QfT).  (Q v)_j needs ROW j of Q in lane j; at a row pitch of 64 bytes
// (n = 8 doubles, the traceless coordinates of d = 3) row owners j and j + 4 read the same LDS banks - the exact mode's rollouts
// and adjoint pass form Q e at every horizon index and spent 9.2 % of their LDS-active cycles in bank conflicts
// (profiles/r03_pmc_config3_B65536_real_exact.txt; the clipped kernels read columns only: 0.4 %).  Column j of Q^T is the same
// numbers at consecutive addresses, whatever Q is (no symmetry assumed).
template <class S, bool TR = false>
struct CostRef {
  const S* Q;
  const S* Qf;
  long q_stride;   // elements between Q(t) and Q(t+1); 0 = constant
  const S* R;
  long r_stride;
  const S* QT = nullptr;    // TR only (constant Q: q_stride == 0)
  const S* QfT = nullptr;
  __device__ __forceinline__ const S* q(int t, int T) const { return t == T ? Qf : Q + (long)t * q_stride; }
  __device__ __forceinline__ const S* qT(int t, int T) const { return t == T ? QfT : QT; }
  __device__ __forceinline__ const S* r(int t) const { return R + (long)t * r_stride; }
};

// The horizon window of one instance: targets (views at window column 0, element 0).
struct Window {
  GView xbm;   // [T+1][NX] of S
  GView ubm;   // [T][NU]
};

// (Q v)_j = sum_i Q[j][i] v_i with Q row-major in memory, v distributed
template <int NX, class S>
__device__ __forceinline__ S qrow_times(const S* Qt, S v, int j) {
  S qr[NX];
#pragma unroll
  for (int i = 0; i < NX; ++i) qr[i] = Qt[j * NX + i];
  return dot_lane_index<false, false, NX>(v, qr);
}
// the same with the stage cost of horizon index t taken from `cost` (its transposed copy when there is one)
template <int NX, class S, bool TR>
__device__ __forceinline__ S qrow_times(const CostRef<S, TR>& cost, int t, int T, S v, int j) {
  if constexpr (TR) {
    const S* Qt = cost.qT(t, T);
    S qr[NX];
#pragma unroll
    for (int i = 0; i < NX; ++i) qr[i] = Qt[i * NX + j];
    return dot_lane_index<false, false, NX>(v, qr);
  } else {
    return qrow_times<NX>(cost.q(t, T), v, j);
  }
}

// row j of the stage cost Q into registers (the closed loop's stage cost does not change along the horizon: index 0 serves t < T)
template <int NX, class S, bool TR>
__device__ __forceinline__ void load_qrow(const CostRef<S, TR>& cost, int T, int j, S (&qr)[NX]) {
  if constexpr (TR) {
    const S* Qt = cost.qT(0, T);
#pragma unroll
    for (int i = 0; i < NX; ++i) qr[i] = Qt[i * NX + j];
  } else {
    const S* Qt = cost.q(0, T);
#pragma unroll
    for (int i = 0; i < NX; ++i) qr[i] = Qt[j * NX + i];
  }
}

// ---------------------------------------------------------------------------------------------
// Backward Riccati sweep on z = [x - xbar; 1], V = [[P, p], [p^H, pi]]  (lqr.py:28-65).
// pi never enters a gain and is not carried.  Lane j owns column j of P and element j of p.
//   K_t = -(R + B^H P B)^-1 [B^H P A_t , B^H (P c_t + p)]           lqr.py:61-62
//   S   = [A_t + B Kx , c_t + B k]
//   P  <- Q + Kx^H R Kx + Sx^H P Sx ;  p <- -Q r + Kx^H R k + Sx^H (P s + p)   lqr.py:64-65
// gains layout: [t][col 0..NX][NU]  (col NX holds k); the view is positioned at the instance.
// ---------------------------------------------------------------------------------------------
// keeps loop-invariant LDS/global loads inside the horizon loops (hoisted, they cost hundreds of VGPRs)
#define M4Q_NO_HOIST() asm volatile("" ::: "memory")

struct Box {       // |u| <= sat, first control also within [lo0, hi0]
  double sat;
  template <int NU>
  __device__ __forceinline__ void at(int t, int k, const double (&lo0)[NU], const double (&hi0)[NU], double& lo, double& hi) const {
    lo = -sat;
    hi = sat;
    if (t == 0) { lo = fmax(lo, lo0[k]); hi = fmin(hi, hi0[k]); }
  }
};

// Working set of the exact box-QP solver (see box_qp_iterate): stat [T][NU] doubles, 0 free, +1 / -1 pinned at the
// upper / lower bound.
template <int NU>
struct PinCtx {
  GView stat;
  Box box;
  double lo0[NU], hi0[NU];
  // adjoint_pass (per row, `derive`): re-derives this row's working set from the gradient at the iterate (Xk, Uk) - the adjoint
  // recursion lam_t = Q e_t + A_t^H lam_{t+1}, g_t = R (u_t - ub_t) + Re B_t^H lam_{t+1}: a control is pinned iff it sits on a
  // bound with the gradient pushing outward.  nchg receives the number of entries that changed with respect to what `stat` held.
  bool tconst = false;      // xbar_t is the same for every t (QP_TARG_CONST): the rollouts and the adjoint pass load it once
  bool derive = false;
  GView Xk, Uk;
  int nchg = 0;
  // rollout_policy (per row, `pdas`): the primal-dual update of the working set, in place: a free control that the unclipped policy
  // puts beyond a bound is pinned there, a pinned one whose multiplier has the wrong sign is let go (RolloutInfo::nchg counts).
  bool pdas = false;
  // pinned value of control k at horizon index t, or free
  __device__ __forceinline__ bool pinned(int t, int k, double& value) const {
    const double st = stat.ld<double>(t * NU + k);
    double lo, hi;
    box.at<NU>(t, k, lo0, hi0, lo, hi);
    value = st > 0.0 ? hi : lo;
    return st != 0.0;
  }
};

template <class S, int NX, int NU, class Prov, bool PINNED = false, bool TC = false, bool TR = false>
__device__ __forceinline__ void riccati_backward(const Prov& prov, int T, const Window& win, const CostRef<S, TR>& cost, int flags,
                                                  const GView& gains, int j, bool store_ok, PinCtx<NU>* pin = nullptr) {
  const bool ref = (flags & QP_REF_LQR) != 0;
  S Pc[NX];
  S pv = zero_of<S>();
  S xb_next = win.xbm.ld<S>(T * NX + j);        // xbar_{t+1} of the first iteration
  {
    const S* Qt = cost.q(T, T);
#pragma unroll
    for (int i = 0; i < NX; ++i) Pc[i] = Qt[i * NX + j];
    if (ref) pv = cneg(qrow_times<NX>(Qt, xb_next, j));
  }
  // operands of horizon index t are fetched while index t+1 is being worked on.  The loop below runs two indices per
  // trip: the two operand sets and the two copies of (P, p) swap roles, so that neither is ever copied.
  struct Ops {
    typename Prov::Lin lin;
    S xb;
    double ub[NU];
    double stv[NU];           // PINNED: working set at this index
  };
  auto load = [&](int t) __attribute__((always_inline)) {
    Ops o;
    o.lin = prov.fetch(t);
    if constexpr (TC && (M4Q_TC_XB_ONCE(NX) || fused_kind<Prov>::sg)) o.xb = xb_next;   // (constant target: the column loaded above serves every index)
    else o.xb = win.xbm.ld<S>(t * NX + j);
    ldn<NU>(win.ubm, t * NU, o.ub);
    if constexpr (PINNED) ldn<NU>(pin->stat, t * NU, o.stv);
    return o;
  };
  // n = 16 real path: one wavefront per SIMD owns all 512 registers, and alone on its SIMD it cannot hide the LDS
  // read-to-use latency of the model at every horizon index: the ROW form of the model is read once per sweep and kept in
  // registers (measured A/B on config 4: 92.9 -> 85.8 ms; profiles/r02_ab_experiments.txt)
  // (a shared-generator provider, FusedProv<..., SG>, is compiled for two wavefronts per SIMD at n = 15: nothing is held over a sweep)
  constexpr bool HOIST_SMALL = TC && sizeof(S) == sizeof(double) && NX < 15 &&
                               std::is_same<Prov, FusedProv<S, NX, NU, Prov::ORDER_>>::value && batch_fits<NX, NU, Prov::ORDER_>();
  constexpr bool HOIST = (std::is_same<Prov, FusedProv<S, NX, NU, Prov::ORDER_>>::value && sizeof(S) == sizeof(double) &&
                          M4Q_XH15(NX)) || HOIST_SMALL;
  ModelRegs<S, NX, NU, Prov::ORDER_> mregs;
  if constexpr (HOIST) mregs.load_rows(prov, j);
  // constant target: real fused path with batched (n <= 9) or hoisted (n = 16, mode 2) model reads
  constexpr bool TCON = TC && sizeof(S) == sizeof(double) &&
                        ((std::is_same<Prov, FusedProv<S, NX, NU, Prov::ORDER_>>::value && ((batch_fits<NX, NU, Prov::ORDER_>()) || HOIST)) ||
                         fused_kind<Prov>::sg);
  S tterm[PowTab<NU, Prov::ORDER_>::NP + 1];
  if constexpr (TCON) prov.target_terms(xb_next, tterm);
  // the same family: column j of Q (the closed loop's stage cost does not change along the horizon) read once per sweep
  constexpr bool QHOIST = HOIST_SMALL || HOIST;          // (n = 15 pinned sweep as well: config 4 exact 1,768 -> 1,760 ms)
  S Qcol[NX];
  if constexpr (QHOIST) {
    const S* Q0 = cost.q(0, T);
#pragma unroll
    for (int i = 0; i < NX; ++i) Qcol[i] = Q0[i * NX + j];
  }
  auto step = [&](int t, const Ops& cur, Ops& nxt, const S (&Pc)[NX], const S pv, S (&Pn)[NX], S& pv_out) __attribute__((always_inline)) {
    M4Q_NO_HOIST();
    nxt = load(t > 0 ? t - 1 : 0);
    const typename Prov::Lin& lin = cur.lin;
    const S xb = cur.xb;
    const double (&ub)[NU] = cur.ub;

    S Ac[NX];
    const S xb1 = xb_next;
    S ax, Brow[NU], dlt;
    // (measured, config 3 real path, A/B on one box: 51.98 -> 50.38 ms; n = 16 would need 256 registers for the batch)
    if constexpr (TCON && HOIST_SMALL) {
      prov.col_batch(lin, Ac);
      prov.rows_tc(mregs, lin, tterm, ax, Brow, dlt);
    } else if constexpr (TCON && HOIST) {
      prov.col(lin, Ac);                       // (its 60 LDS reads as one batch: no change, 71.3 ms either way)
      prov.rows_tc(mregs, lin, tterm, ax, Brow, dlt);
    } else if constexpr (TCON && fused_kind<Prov>::sg) {
      prov.col(lin, Ac);
      prov.rows_tc_lds(lin, tterm, ax, Brow, dlt);
    } else if constexpr (TCON) {
      prov.col_rows_tc(lin, tterm, Ac, ax, Brow, dlt);
    } else if constexpr (HOIST) {
      prov.col(lin, Ac);
      prov.rows(mregs, lin, xb, ax, Brow, dlt);
    } else if constexpr (sizeof(S) == sizeof(double) && batch_fits<NX, NU, Prov::ORDER_>()) {
      prov.col_rows(lin, xb, Ac, ax, Brow, dlt);
    } else {
      prov.col(lin, Ac);
      prov.rows(lin, xb, ax, Brow, dlt);
    }

    // affine column of the augmented dynamics
    S c = ax;
#pragma unroll
    for (int k = 0; k < NU; ++k) cmac_r(c, Brow[k], ub[k]);
    c = ref ? csub(c, xb) : cadd(c, csub(dlt, xb1));          // lqr.py:45  |  optimize.py:41

    // BhP[k] = (B^H P)[k][j] = sum_i conj(B[i][k]) P[i][j]
    S BhP[NU];
#pragma unroll
    for (int k = 0; k < NU; ++k) BhP[k] = dot_lane_index<true, false, NX>(Brow[k], Pc);
    const S w = dot_lane_index<false, true, NX>(c, Pc, pv);   // (P c + p)_j

    // G = R + B^H P B (replicated), h = B^H (P c + p)
    const S* Rt = cost.r(t);
    cplx g[NU][NU], ginv[NU][NU];
    S h[NU];
#pragma unroll
    for (int k = 0; k < NU; ++k) {
#pragma unroll
      for (int l = k; l < NU; ++l) {
        const S prod = cmul(BhP[k], Brow[l]);
        const S rkl = Rt[k * NU + l];
        if (l == k) g[k][l] = mk(real_of(rkl) + rowsum<NX>(real_of(prod)), 0.0);
        else g[k][l] = as_cplx(cadd(rkl, rowsum<NX>(prod)));
      }
      h[k] = rowsum<NX>(cmul(cconj(Brow[k]), w));
    }
    // Hh[l] = (B^H P A_t)[l][j] = sum_i BhP[l][i] A_t[i][j]
    S Hh[NU];
#pragma unroll
    for (int l = 0; l < NU; ++l) Hh[l] = dot_lane_index<false, false, NX>(BhP[l], Ac);
    bool fix[NU];
    double dufix[NU];
    S Hraw[NU], hraw[NU], Graw[NU][NU];          // PINNED only
#pragma unroll
    for (int k = 0; k < NU; ++k) { fix[k] = false; dufix[k] = 0.0; }
    if constexpr (PINNED) {
      // controls pinned at a bound are constants of the stage: K row = [0 | du_fix]; the free ones respond to them:
      //   G_ff du_f = -(H_f dx + h_f + G_fp du_p)
      double stv[NU];
#pragma unroll
      for (int k = 0; k < NU; ++k) stv[k] = cur.stv[k];
#pragma unroll
      for (int k = 0; k < NU; ++k) {
        double lo, hi;
        pin->box.template at<NU>(t, k, pin->lo0, pin->hi0, lo, hi);
        fix[k] = stv[k] != 0.0;
        dufix[k] = fix[k] ? (stv[k] > 0.0 ? hi : lo) - ub[k] : 0.0;
      }
#pragma unroll
      for (int k = 0; k < NU; ++k) {
#pragma unroll
        for (int l = 0; l < NU; ++l) {
          const cplx gkl = k <= l ? g[k][l] : cconj(g[l][k]);
          cmac_r(h[k], from_cplx<S>(gkl), dufix[l]);
        }
      }
      // (what the stage's gradient is for a pinned control, kept for its multiplier row below: H_k, h_k + G_k. du_pinned, G_k.)
#pragma unroll
      for (int k = 0; k < NU; ++k) {
        Hraw[k] = Hh[k];
        hraw[k] = h[k];
#pragma unroll
        for (int l = 0; l < NU; ++l) Graw[k][l] = from_cplx<S>(k <= l ? g[k][l] : cconj(g[l][k]));
      }
#pragma unroll
      for (int k = 0; k < NU; ++k) {
#pragma unroll
        for (int l = k; l < NU; ++l)
          if (fix[k] || fix[l]) g[k][l] = mk(k == l ? 1.0 : 0.0, 0.0);
        if (fix[k]) { Hh[k] = zero_of<S>(); h[k] = zero_of<S>(); }
      }
    }
    herm_inverse<NU>(g, ginv);
    S Kx[NU], kk[NU];
#pragma unroll
    for (int k = 0; k < NU; ++k) {
      S a = zero_of<S>(), b = zero_of<S>();
#pragma unroll
      for (int l = 0; l < NU; ++l) {
        cmac(a, from_cplx<S>(ginv[k][l]), Hh[l]);
        cmac(b, from_cplx<S>(ginv[k][l]), h[l]);
      }
      Kx[k] = cneg(a);
      kk[k] = fix[k] ? from_real<S>(dufix[k]) : cneg(b);
    }
    // what is stored for the rollout: the gain row [Kx | k] of a free control; for a PINNED one (whose gain row would be the
    // trivial [0 | du_fix]) the affine form of its multiplier along the policy's own trajectory,
    //   mu_k(dx) = [H_k + sum_{l free} G_kl Kx_l] dx + h_k + sum_l G_kl du_l        (the stage's gradient with respect to u_k)
    // so that the rollout - which forms row . dx + const for every control anyway - can tell whether the face minimiser it has
    // just produced satisfies the sign conditions (KKT), without an adjoint pass.
    S Kst[NU], kst[NU];
#pragma unroll
    for (int k = 0; k < NU; ++k) { Kst[k] = Kx[k]; kst[k] = kk[k]; }
    if constexpr (PINNED) {
#pragma unroll
      for (int k = 0; k < NU; ++k) {
        S m = Hraw[k], c0 = hraw[k];
#pragma unroll
        for (int l = 0; l < NU; ++l) {
          cmac(m, Graw[k][l], fix[l] ? zero_of<S>() : Kx[l]);
          cmac(c0, Graw[k][l], fix[l] ? zero_of<S>() : kk[l]);           // (pinned l: G_kl du_fix_l is already inside hraw)
        }
        Kst[k] = fix[k] ? m : Kx[k];
        kst[k] = fix[k] ? c0 : kk[k];
      }
    }
    if (store_ok) {
      const unsigned gt = (unsigned)t * (NX + 1) * NU;
      stn<NU>(gains, gt + j * NU, Kst);
      if (M4Q_STORE_ALL(NX) || j == 0) stn<NU>(gains, gt + NX * NU, kst);   // (replicated over the row: every lane may write the same bytes)
    }

    // closed loop: Sx = A_t + B Kx (column j, in place): Ac[i] += lane_i(B[i][k]) Kx[k];  s = c + B k
#pragma unroll
    for (int k = 0; k < NU; ++k) mac_lane_index<false, false, NX>(Ac, Brow[k], Kx[k]);
    S s = c;
#pragma unroll
    for (int k = 0; k < NU; ++k) cmac(s, Brow[k], kk[k]);

    S PSc[NX];
    matmul_cols<NX>(PSc, Pc, Ac);                              // P Sx
    const S ws = dot_lane_index<false, true, NX>(s, Pc, pv);  // (P s + p)_j

    S RK[NU], Rk[NU];
#pragma unroll
    for (int k = 0; k < NU; ++k) {
      RK[k] = zero_of<S>();
      Rk[k] = zero_of<S>();
#pragma unroll
      for (int l = 0; l < NU; ++l) {
        const S rkl = Rt[k * NU + l];
        cmac(RK[k], rkl, Kx[l]);
        cmac(Rk[k], rkl, kk[l]);
      }
    }
    const S* Qt = cost.q(t, T);
#pragma unroll
    for (int i = 0; i < NX; ++i) Pn[i] = QHOIST ? Qcol[i] : Qt[i * NX + j];
    matmul_cols_hn_acc<NX>(Pn, Ac, PSc);                       // + Sx^H P Sx
#pragma unroll
    for (int k = 0; k < NU; ++k) mac_lane_index<true, false, NX>(Pn, Kx[k], RK[k]);   // + Kx^H R Kx
    S pn = matvec_h<NX>(Ac, ws);                               // Sx^H (P s + p)
#pragma unroll
    for (int k = 0; k < NU; ++k) cmac_cj(pn, Kx[k], Rk[k]);
    if (ref) pn = csub(pn, qrow_times<NX>(Qt, xb, j));        // - Q xbar_t   (lqr.py:54-58)
    pv_out = pn;
    xb_next = xb;
  };
  S Pd[NX];
  S pd = zero_of<S>();
  Ops opsA = load(T - 1), opsB;
  // The two-index form serves S = double only: on the complex path it brings nothing (bound by FMA issue, not by the copies) and
  // one experimental d = 4 build of it faulted in round 2 (closed in DESIGN.md 4.6: the faulting tree was never committed, rebuilds
  // of the neighbouring commits pass a static audit and the GPU suite; the form is gone from the source).
  if constexpr (sizeof(S) == sizeof(double)) {
    int t = T - 1;
    for (; t >= 1; t -= 2) {
      step(t, opsA, opsB, Pc, pv, Pd, pd);
      step(t - 1, opsB, opsA, Pd, pd, Pc, pv);
    }
    if (t == 0) step(0, opsA, opsB, Pc, pv, Pd, pd);
  } else {
    // complex path: one index per trip and plain copies (twice the registers per matrix: the two-index form does not pay)
    for (int t = T - 1; t >= 0; --t) {
      step(t, opsA, opsB, Pc, pv, Pd, pd);
#pragma unroll
      for (int i = 0; i < NX; ++i) Pc[i] = Pd[i];
      pv = pd;
      opsA = opsB;
    }
  }
}

// ---------------------------------------------------------------------------------------------
// Working set of the exact box-QP solver from the gradient at the iterate (Xk, Uk): the adjoint recursion alone, no Riccati
// arithmetic - ~100 vector instructions per horizon index against ~800 of the pinned sweep.  (Round 2 ran it inside the sweep:
// "no pass of its own", but then every solve ended with a full sweep whose only product was the verdict "nothing changed".)
// Operands are fetched PF indices ahead (an index is too short to hide the workspace's latency behind the previous one; four
// ahead spilled inside the loop).
// ---------------------------------------------------------------------------------------------
template <class S, int NX, int NU, class Prov, bool TR = false>
__device__ __forceinline__ void adjoint_pass(const Prov& prov, int T, const Window& win, const CostRef<S, TR>& cost, PinCtx<NU>& pin, int j,
                                             bool store_ok) {
  constexpr int PF = 2;
  struct Ops {
    typename Prov::Lin lin;
    S xb, xk;
    double ub[NU], stv[NU], uk[NU];
  };
  const S xb0 = win.xbm.ld<S>(j);
  auto load = [&](int t) __attribute__((always_inline)) {
    Ops o;
    o.lin = prov.fetch(t);
    o.xb = xb0;
    if (!pin.tconst) o.xb = win.xbm.ld<S>(t * NX + j);
    o.xk = pin.Xk.template ld<S>(t * NX + j);
    ldn<NU>(win.ubm, t * NU, o.ub);
    ldn<NU>(pin.stat, t * NU, o.stv);
    ldn<NU>(pin.Uk, t * NU, o.uk);
    return o;
  };
  pin.nchg = 0;
  S lam = qrow_times<NX>(cost, T, T, csub(pin.Xk.template ld<S>(T * NX + j), win.xbm.ld<S>(T * NX + j)), j);
  constexpr bool AHOIST = std::is_same<Prov, FusedProv<S, NX, NU, Prov::ORDER_>>::value &&
                          sizeof(S) == sizeof(double) && batch_fits<NX, NU, Prov::ORDER_>() && Prov::ORDER_ == 1;
  // n >= 15 (one wavefront per SIMD, 512 registers): both forms of the model and Q's row stay in registers over the pass -
  // at one wavefront per SIMD every LDS read-to-use latency is exposed
  constexpr bool AH15 = std::is_same<Prov, FusedProv<S, NX, NU, Prov::ORDER_>>::value && sizeof(S) == sizeof(double) && M4Q_XH15(NX) &&
                        Prov::ORDER_ == 1;
  ModelRegs<S, NX, NU, Prov::ORDER_> mregs;
  S Qrow[AH15 ? NX : 1];
  if constexpr (AH15) {
    mregs.load(prov, j);
    load_qrow<NX>(cost, T, j, Qrow);
  } else if constexpr (AHOIST) {
    mregs.load_rows(prov, j);
  }
  auto step = [&](int t, const Ops& cur) __attribute__((always_inline)) {
    S Ac[NX], Brow[NU];
    if constexpr (AH15) {
#pragma unroll
      for (int i = 0; i < NX; ++i) {
        S a = mregs.col[0][i];
#pragma unroll
        for (int p = 0; p < NU; ++p) cmac_r(a, mregs.col[1 + p][i], cur.lin.u[p]);        // (order 1: monomial p is u_p)
        Ac[i] = a;
      }
#pragma unroll
      for (int k = 0; k < NU; ++k) Brow[k] = dot_lane_index<false, false, NX>(cur.lin.xg, mregs.row[1 + k]);
    } else if constexpr (AHOIST) {
      // (the column form held in registers as well - 24 LDS reads per index less - measured at n = 8: 183-186 ms either way)
      prov.col_batch(cur.lin, Ac);
#pragma unroll
      for (int k = 0; k < NU; ++k) Brow[k] = dot_lane_index<false, false, NX>(cur.lin.xg, mregs.row[1 + k]);   // (order 1: monomial p is u_p)
    } else {
      prov.col(cur.lin, Ac);
      prov.brows(cur.lin, Brow);
    }
    const S* Rt = cost.r(t);
#pragma unroll
    for (int k = 0; k < NU; ++k) {
      double gk = rowsum<NX>(real_of(cmul(cconj(Brow[k]), lam)));
#pragma unroll
      for (int l = 0; l < NU; ++l) gk = fma(real_of(Rt[k * NU + l]), cur.uk[l] - cur.ub[l], gk);
      double lo, hi;
      pin.box.template at<NU>(t, k, pin.lo0, pin.hi0, lo, hi);
      const double eps = 1e-12 * pin.box.sat;
      double sv = 0.0;
      if (cur.uk[k] <= lo + eps && gk > 0.0) sv = -1.0;
      if (cur.uk[k] >= hi - eps && gk < 0.0) sv = 1.0;
      if (hi - lo <= 2 * eps) sv = 1.0;                // degenerate interval: nothing to optimise
      if (pin.derive) {
        pin.nchg += cur.stv[k] != sv ? 1 : 0;
        if (store_ok && j == 0) pin.stat.template st<double>(t * NU + k, sv);
      }
    }
    const S e = csub(cur.xk, cur.xb);
    S qe;
    if constexpr (AH15) qe = dot_lane_index<false, false, NX>(e, Qrow);
    else qe = qrow_times<NX>(cost, t, T, e, j);
    lam = dot_lane_index<false, true, NX>(lam, Ac, qe);   // Q_t e_t + A_t^H lam
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

// ---------------------------------------------------------------------------------------------
// Forward rollout with clipping (lqr.py:67-79; dynamics with Delta: optimize.py:41).
// x distributed (lane j holds x_t[j]).  WANT_COST: return the objective (replicated over the row);
// otherwise return sum |x|^2 + sum u^2, which is finite exactly when every state and control is -
// all the closed loop needs for its exit code 3 (mpc.py:200-203).  u_first receives the first control.
// shift_out (per row): the solution goes straight into the NEXT step's guess - Xg/Ug, already shifted as
// mpc.py:271-272 does (x_{t+1} -> Xg[t], u_t -> Ug[t-1], last column repeated) - which is what a warm
// step (alpha = 1, mpc.py:208-212) ends up with after its update and shift passes.  Safe in place: index
// t of the guess has been read (one iteration ahead) before it is overwritten.
// ---------------------------------------------------------------------------------------------
// TCF: xbar_t is the same for every t (the caller has seen QP_TARG_CONST): one load per rollout instead of one per index.
template <class S, int NX, int NU, bool WANT_COST, bool TCF = false, class Prov, bool TR = false>
__device__ __forceinline__ double rollout_forward(const Prov& prov, int T, S x0, const Window& win, const CostRef<S, TR>& cost,
                                                   int flags, const GView& gains, double sat, const double (&lo0)[NU],
                                                   const double (&hi0)[NU], const GView& Xo, const GView& Uo, int j,
                                                   bool store_ok, double (&u_first)[NU], bool shift_out = false,
                                                   const GView* Xg = nullptr, const GView* Ug = nullptr) {
  const bool ref = (flags & QP_REF_LQR) != 0;
  S xb0 = zero_of<S>();
  if constexpr (TCF) xb0 = win.xbm.ld<S>(j);
  S x = x0;
  // destination views: lane offsets select between (Xo, Uo) and the shifted guess
  GView Xd = Xo, Ud = Uo;
  int xs_shift = 1, us_shift = 0;
  if (Xg != nullptr) {
    Xd.off = shift_out ? Xg->off : Xo.off;
    Ud.off = shift_out ? Ug->off : Uo.off;
    xs_shift = shift_out ? 0 : 1;
    us_shift = shift_out ? -1 : 0;
  }
  if (store_ok && !shift_out) Xd.st<S>(j, x);
  double cx = 0.0;     // per-lane share of the state cost
  double cu = 0.0;     // control cost (replicated)
  struct Ops {
    typename Prov::Lin lin;
    S xb;
    double ub[NU];
    S Kx[NU];
    double kre[NU];
  };
  auto load = [&](int t) __attribute__((always_inline)) {
    Ops o;
    o.lin = prov.fetch(t);
    if constexpr (TCF) o.xb = xb0;
    else o.xb = win.xbm.ld<S>(t * NX + j);
    const unsigned gt = (unsigned)t * (NX + 1) * NU;
    ldn<NU>(win.ubm, t * NU, o.ub);
    ldn<NU>(gains, gt + j * NU, o.Kx);
    S kk[NU];
    ldn<NU>(gains, gt + NX * NU, kk);
#pragma unroll
    for (int k = 0; k < NU; ++k) o.kre[k] = real_of(kk[k]);
    return o;
  };
  constexpr bool HOIST = std::is_same<Prov, FusedProv<S, NX, NU, Prov::ORDER_>>::value && sizeof(S) == sizeof(double) &&
                         (M4Q_XH15(NX) || (batch_fits<NX, NU, Prov::ORDER_>()));
  ModelRegs<S, NX, NU, Prov::ORDER_> mregs;
  if constexpr (HOIST) mregs.load_rows(prov, j);
  // one horizon index: `cur` holds its operands, those of the next index are fetched into `nxt` meanwhile.  The loop
  // below runs two indices per trip with the two operand sets swapping roles, so that no set is ever copied.
  // (Fetching two indices ahead with three rotating sets was measured: +1.6 %, profiles/r02_ab_experiments.txt.)
  auto step = [&](int t, const Ops& cur, Ops& nxt) __attribute__((always_inline)) {
    M4Q_NO_HOIST();
    nxt = load(t + 1 < T ? t + 1 : t);
    // (SGV, shared generators: the step as ONE product of the row [A | N_1 .. N_m] read from LDS once,
    //  x+ = A_i x + sum_k N_k (u~g_k x + (u~_k - u~g_k) xg) with u~ = s u - no row of A_t is built: NU NX FMAs per index fewer)
    constexpr bool SGV = M4Q_SG_VFORM && fused_kind<Prov>::sg;
    S ax, Brow[NU], dlt;
    if constexpr (SGV) { (void)ax; (void)Brow; (void)dlt; }
    else if constexpr (HOIST) prov.rows(mregs, cur.lin, x, ax, Brow, dlt);
    else prov.rows(cur.lin, x, ax, Brow, dlt);
    const S dx = csub(x, cur.xb);
    double u[NU];
#pragma unroll
    for (int k = 0; k < NU; ++k) {
      const double part = real_of(cmul(cur.Kx[k], dx));
      double uk = rowsum<NX>(part) + cur.kre[k] + cur.ub[k];           // lqr.py:75
      double lo = -sat, hi = sat;
      if (t == 0) {
        lo = fmax(lo, lo0[k]);
        hi = fmin(hi, hi0[k]);
      }
      uk = fmin(fmax(uk, lo), hi);                                     // lqr.py:76
      u[k] = uk;
      if (t == 0) u_first[k] = uk;
    }
    S xn;
    if constexpr (SGV) {
      {
        S row[NX];
#pragma unroll
        for (int k = 0; k < NX; ++k) row[k] = prov.el(0, j, k);
        xn = dot_lane_index<false, false, NX>(x, row);
      }
#pragma unroll
      for (int p = 0; p < NU; ++p) {
        const double ugp = cur.lin.u[p];                       // (already scaled by the member's s_p)
        const double dup = ref ? u[p] * prov.sc[p] : fma(u[p], prov.sc[p], -ugp);
        S v = cscale(cur.lin.xg, dup);
        cmac_r(v, x, ugp);
        S row[NX];
#pragma unroll
        for (int k = 0; k < NX; ++k) row[k] = prov.el(1 + p, j, k);
        xn = dot_lane_index<false, false, NX>(v, row, xn);
      }
    } else {
      xn = ref ? ax : cadd(ax, dlt);
#pragma unroll
      for (int k = 0; k < NU; ++k) cmac_r(xn, Brow[k], u[k]);
    }
    if constexpr (WANT_COST) {
      const S* Rt = cost.r(t);
      const S e = ref ? xn : dx;
      const S qe = qrow_times<NX>(cost, ref ? t + 1 : t, T, e, j);
      cx += dot_re(e, qe);
#pragma unroll
      for (int k = 0; k < NU; ++k) {
#pragma unroll
        for (int l = 0; l < NU; ++l) {
          const double ek = ref ? u[k] : u[k] - cur.ub[k];
          const double el = ref ? u[l] : u[l] - cur.ub[l];
          cu += ek * real_of(Rt[k * NU + l]) * el;
        }
      }
    } else {
      cx += norm2(xn);
#pragma unroll
      for (int k = 0; k < NU; ++k) cu = fma(u[k], u[k], cu);
    }
    x = xn;
    if (store_ok) {
      Xd.st<S>((t + xs_shift) * NX + j, x);
      // (u is replicated over the row: every lane writes the same bytes.  A shifting row's u_0 has no slot: it goes to slot 0,
      //  which the same lane's store of u_1 overwrites - no exec mask to set up)
      if (M4Q_STORE_ALL(NX) || j == 0) stn<NU>(Ud, (unsigned)(t + us_shift > 0 ? t + us_shift : 0) * NU, u);
      if (shift_out && t == T - 1) {
        Xd.st<S>(T * NX + j, x);                       // repeat the last column
        if (M4Q_STORE_ALL(NX) || j == 0) stn<NU>(Ud, (T - 1) * NU, u);
      }
    }
  };

  Ops opsA = load(0), opsB;
  int t = 0;
  for (; t + 1 < T; t += 2) {
    step(t, opsA, opsB);
    step(t + 1, opsB, opsA);
  }
  if (t < T) step(t, opsA, opsB);
  if constexpr (WANT_COST) {
    if (!ref) {
      const S e = csub(x, win.xbm.ld<S>(T * NX + j));
      const S qe = qrow_times<NX>(cost, T, T, e, j);
      cx += dot_re(e, qe);
    }
  }
  return rowsum<NX>(cx) + cu;
}

// ---------------------------------------------------------------------------------------------
// Exact box-constrained QP (the statement of optimize.quad_program, optimize.py:27-43,54, which the live reference
// hands to OSQP): a primal active-set method on the Riccati factorisation.
//   iterate  u^k feasible, x^k its (linearised-model) trajectory, J^k its objective, W the working set (controls
//   pinned on a bound);
//   1. Riccati sweep with the controls of W held (riccati_backward<PINNED>): policy of the minimiser of J over the face;
//   2. closed-loop rollout of that policy, clipping the free controls that leave the box (the feedback re-plans the
//      later ones around each clipped value).  Nothing clipped: it is the face minimiser u_N.  Either way the point is
//      feasible and becomes the next iterate if J decreases (large changes of the active set in one step);
//   3. otherwise the classical step: unclipped rollout -> u_N, move along the segment u^k -> u_N up to the first bound
//      met (ratio test), pin the control(s) met there - all of them: with a step of length zero (free controls sitting
//      on a bound the Newton step wants to cross, common with these stiff Hessians) they would otherwise come one per
//      sweep.  J(a) = J_N + (J^k - J_N)(1 - a)^2 along the segment, so this always decreases J and needs no
//      evaluation; trajectories blend linearly;
//   4. W is re-derived from the gradient (adjoint_pass, once per call of box_qp_iterate: a control on a bound
//      with the gradient pushing outward is pinned, all others free) after 2; kept, plus the blocking controls, after 3.
// Converged when a face minimiser has multipliers of the right sign on every pinned control (KKT): read off the policy rollout
// (RolloutInfo::nbad, from the multiplier rows riccati_backward<PINNED> stores for pinned controls), or - for a multiplier the two
// computations put on different sides of zero - when the re-derived working set comes back unchanged.
// J decreases strictly from iterate to iterate, so no face is visited twice.
// ---------------------------------------------------------------------------------------------
struct QpStats {   // per row, counted by the caller
  int sweeps = 0, ratio_steps = 0;                   // pinned Riccati sweeps, ratio-test steps
  int end_kkt = 0, end_precision = 0, end_cap = 0;   // how the solve ended
};

// What a policy rollout reports besides the objective (all replicated over the row).
struct RolloutInfo {
  double dmax;       // max |u - u^k|
  bool outside;      // some free control left the box (before clipping)
  double alpha;      // ratio test: largest step along u^k -> u that stays in the box (<= 1)
  int nbad;          // pinned controls whose multiplier along this rollout has the wrong sign (meaningful when !outside: the
                     // rollout is then the minimiser over its face, and nbad == 0 is the KKT test)
  int nchg;          // PinCtx::pdas: entries of the working set this rollout changed (0: the unclipped rollout is feasible and
                     // every multiplier has the right sign - the optimum)
};

// closed-loop rollout of the policy in `gains` (pinned controls sit on their bound), with (clip) or without clipping
// of the free controls.  Writes the trial point (Xc, Uc); returns the objective.
template <class S, int NX, int NU, class Prov, bool TR = false>
__device__ __forceinline__ double rollout_policy(const Prov& prov, int T, S x0, const Window& win, const CostRef<S, TR>& cost,
                                                 const GView& gains, const PinCtx<NU>& pin, const GView& Uk, bool clip,
                                                 const GView& Xc, const GView& Uc, int j, bool store_ok, RolloutInfo& info) {
  S x = x0;
  if (store_ok) Xc.st<S>(j, x);
  info.dmax = 0.0;
  info.outside = false;
  info.alpha = 1.0;
  info.nbad = 0;
  info.nchg = 0;
  double cx = 0.0, cu = 0.0;
  // operands of horizon index t+1 are fetched while index t computes (two sets swapping roles, as in rollout_forward): the
  // step is a short dependent chain and every operand comes from the workspace, i.e. from beyond the L2
  struct Ops {
    typename Prov::Lin lin;
    S xb;
    double ub[NU], uk[NU], st[NU];
    S Kx[NU];
    double kre[NU];
  };
  const S xb0 = win.xbm.ld<S>(j);
  auto load = [&](int t) __attribute__((always_inline)) {
    Ops o;
    o.lin = prov.fetch(t);
    o.xb = xb0;
    if (!pin.tconst) o.xb = win.xbm.ld<S>(t * NX + j);
    const unsigned gt = (unsigned)t * (NX + 1) * NU;
    ldn<NU>(win.ubm, t * NU, o.ub);
    ldn<NU>(Uk, t * NU, o.uk);
    ldn<NU>(pin.stat, t * NU, o.st);
    ldn<NU>(gains, gt + j * NU, o.Kx);
    S kk[NU];
    ldn<NU>(gains, gt + NX * NU, kk);
#pragma unroll
    for (int k = 0; k < NU; ++k) o.kre[k] = real_of(kk[k]);
    return o;
  };
  // (the row form of the model in registers over the rollout, as rollout_forward: n <= 9 real paths; exact mode 190 -> 184 ms with
  //  this in the policy and open rollouts and their replicated u stores unmasked: profiles/r04_ab_experiments.txt)
  constexpr bool HOIST = std::is_same<Prov, FusedProv<S, NX, NU, Prov::ORDER_>>::value && sizeof(S) == sizeof(double) &&
                         (M4Q_XH15(NX) || batch_fits<NX, NU, Prov::ORDER_>());
  ModelRegs<S, NX, NU, Prov::ORDER_> mregs;
  if constexpr (HOIST) mregs.load_rows(prov, j);
  constexpr bool QH = HOIST && M4Q_XH15(NX);          // (n >= 15, one wavefront per SIMD: Q's row in registers as well; at n = 8,
                                                       //  two wavefronts per SIMD, the 16 registers cost more: 185 -> 190 ms)
  S Qrow[QH ? NX : 1];
  if constexpr (QH) load_qrow<NX>(cost, T, j, Qrow);
  auto step = [&](int t, const Ops& cur, Ops& nxt) __attribute__((always_inline)) {
    M4Q_NO_HOIST();
    nxt = load(t + 1 < T ? t + 1 : t);
    S ax, Brow[NU], dlt;
    if constexpr (HOIST) prov.rows(mregs, cur.lin, x, ax, Brow, dlt);
    else prov.rows(cur.lin, x, ax, Brow, dlt);
    const S dx = csub(x, cur.xb);
    if constexpr (QH) cx += dot_re(dx, dot_lane_index<false, false, NX>(dx, Qrow));
    else cx += dot_re(dx, qrow_times<NX>(cost, t, T, dx, j));
    const S* Rt = cost.r(t);
    S xn = cadd(ax, dlt);
    double un[NU], eu[NU];
#pragma unroll
    for (int k = 0; k < NU; ++k) {
      double lo, hi;
      pin.box.template at<NU>(t, k, pin.lo0, pin.hi0, lo, hi);
      const bool fixd = cur.st[k] != 0.0;
      const double pv = cur.st[k] > 0.0 ? hi : lo;                 // pinned value (PinCtx::pinned)
      const double uk = cur.uk[k];
      const double part = real_of(cmul(cur.Kx[k], dx));
      const double lin = rowsum<NX>(part) + cur.kre[k];            // free: du = Kx dx + k;  pinned: its multiplier (riccati_backward)
      const double v = lin + cur.ub[k];
      un[k] = fixd ? pv : v;
      // pinned at the upper bound wants a gradient pushing up (mu < 0), at the lower one mu > 0 - the rule of adjoint_pass;
      // a degenerate interval is pinned whatever the sign
      const bool wrong = fixd && hi - lo > 2e-12 * pin.box.sat && !(cur.st[k] > 0.0 ? lin < 0.0 : lin > 0.0);
      if (wrong) ++info.nbad;
      const bool above = un[k] > hi, below = un[k] < lo;
      if (pin.pdas) {
        const double ns = fixd ? (wrong ? 0.0 : cur.st[k]) : (above ? 1.0 : below ? -1.0 : 0.0);
        if (ns != cur.st[k]) {
          ++info.nchg;
          if (store_ok && j == 0) pin.stat.template st<double>(t * NU + k, ns);
        }
      }
      if (above || below) {
        info.outside = true;
        info.alpha = fmin(info.alpha, ((above ? hi : lo) - uk) / (un[k] - uk));        // u^k is feasible: 0 <= a < 1
      }
      if (clip) un[k] = fmin(fmax(un[k], lo), hi);
      eu[k] = un[k] - cur.ub[k];
      info.dmax = fmax(info.dmax, fabs(un[k] - uk));
      cmac_r(xn, Brow[k], un[k]);
    }
#pragma unroll
    for (int k = 0; k < NU; ++k)
#pragma unroll
      for (int l = 0; l < NU; ++l) cu = fma(eu[k] * real_of(Rt[k * NU + l]), eu[l], cu);
    x = xn;
    if (store_ok) {
      Xc.st<S>((t + 1) * NX + j, x);
      if (M4Q_STORE_ALL(NX) || j == 0) stn<NU>(Uc, t * NU, un);      // (replicated over the row: same bytes from every lane)
    }
  };
  Ops opsA = load(0), opsB;
  int t = 0;
  for (; t + 1 < T; t += 2) {
    step(t, opsA, opsB);
    step(t + 1, opsB, opsA);
  }
  if (t < T) step(t, opsA, opsB);
  const S e = csub(x, win.xbm.ld<S>(T * NX + j));
  cx += dot_re(e, qrow_times<NX>(cost, T, T, e, j));
  return rowsum<NX>(cx) + cu;
}

// open-loop rollout of u = clip(U) (the starting point of a solve) with the objective of optimize.py:33-34,54;
// writes the clipped controls and their trajectory to (Xc, Uc).
template <class S, int NX, int NU, class Prov, bool TR = false>
__device__ __forceinline__ double rollout_open(const Prov& prov, int T, S x0, const Window& win, const CostRef<S, TR>& cost,
                                               const GView& U, const Box& box, const double (&lo0)[NU], const double (&hi0)[NU],
                                               const GView& Xc, const GView& Uc, int j, bool store_ok) {
  S x = x0;
  if (store_ok) Xc.st<S>(j, x);
  double cx = 0.0, cu = 0.0;
  struct Ops {                      // operands of index t+1 fetched while index t computes (as rollout_policy)
    typename Prov::Lin lin;
    S xb;
    double ub[NU], uin[NU];
  };
  auto load = [&](int t) __attribute__((always_inline)) {
    Ops o;
    o.lin = prov.fetch(t);
    o.xb = win.xbm.ld<S>(t * NX + j);
    ldn<NU>(win.ubm, t * NU, o.ub);
    ldn<NU>(U, t * NU, o.uin);
    return o;
  };
  constexpr bool HOIST = std::is_same<Prov, FusedProv<S, NX, NU, Prov::ORDER_>>::value && sizeof(S) == sizeof(double) &&
                         (M4Q_XH15(NX) || batch_fits<NX, NU, Prov::ORDER_>());
  ModelRegs<S, NX, NU, Prov::ORDER_> mregs;
  if constexpr (HOIST) mregs.load_rows(prov, j);
  constexpr bool QH = HOIST && M4Q_XH15(NX);
  S Qrow[QH ? NX : 1];
  if constexpr (QH) load_qrow<NX>(cost, T, j, Qrow);
  auto step = [&](int t, const Ops& cur, Ops& nxt) __attribute__((always_inline)) {
    M4Q_NO_HOIST();
    nxt = load(t + 1 < T ? t + 1 : t);
    S ax, Brow[NU], dlt;
    if constexpr (HOIST) prov.rows(mregs, cur.lin, x, ax, Brow, dlt);
    else prov.rows(cur.lin, x, ax, Brow, dlt);
    const S e = csub(x, cur.xb);
    if constexpr (QH) cx += dot_re(e, dot_lane_index<false, false, NX>(e, Qrow));
    else cx += dot_re(e, qrow_times<NX>(cost, t, T, e, j));
    const S* Rt = cost.r(t);
    double u[NU], eu[NU];
    S xn = cadd(ax, dlt);
#pragma unroll
    for (int k = 0; k < NU; ++k) {
      double lo, hi;
      box.at<NU>(t, k, lo0, hi0, lo, hi);
      u[k] = fmin(fmax(cur.uin[k], lo), hi);
      eu[k] = u[k] - cur.ub[k];
      cmac_r(xn, Brow[k], u[k]);
    }
#pragma unroll
    for (int k = 0; k < NU; ++k)
#pragma unroll
      for (int l = 0; l < NU; ++l) cu = fma(eu[k] * real_of(Rt[k * NU + l]), eu[l], cu);
    x = xn;
    if (store_ok) {
      Xc.st<S>((t + 1) * NX + j, x);
      if (M4Q_STORE_ALL(NX) || j == 0) stn<NU>(Uc, t * NU, u);
    }
  };
  Ops opsA = load(0), opsB;
  int t = 0;
  for (; t + 1 < T; t += 2) {
    step(t, opsA, opsB);
    step(t + 1, opsB, opsA);
  }
  if (t < T) step(t, opsA, opsB);
  const S e = csub(x, win.xbm.ld<S>(T * NX + j));
  cx += dot_re(e, qrow_times<NX>(cost, T, T, e, j));
  return rowsum<NX>(cx) + cu;
}

// Per-row state of a box-QP solve in progress (uniform inside a row).  The buffers: two trajectory pairs
// (Xa, Ua) / (Xb, Ub) - which must share their wave-uniform bases, rows pick theirs by lane offset - ping-pong;
// `cur_is_a` tells which one holds the iterate (and, once `busy` drops, the answer).
// The row's booleans live as bits of ONE integer (a VGPR): a bool that differs between rows is a 64-bit lane mask in an SGPR pair, and
// the dozen of them that are live across the sweeps of an iteration were most of what the exact kernels spilled (60-84 SGPRs per
// instantiation into VGPR lanes).  Read where needed (v_and + v_cmp), and made opaque after every sweep (settle) so that the
// comparisons are not hoisted back across it.
template <int BIT>
struct RowBit {
  int& w;
  __device__ __forceinline__ operator bool() const { return (w & BIT) != 0; }
  __device__ __forceinline__ RowBit& operator=(bool v) { w = v ? (w | BIT) : (w & ~BIT); return *this; }
};
struct BoxQpRow {
  int w = 2 | 8;
  __device__ __forceinline__ RowBit<1> busy() { return RowBit<1>{w}; }
  __device__ __forceinline__ RowBit<2> need_adj() { return RowBit<2>{w}; }    // the working set has to be (re-)derived from the gradient at the iterate
  __device__ __forceinline__ RowBit<4> face_min() { return RowBit<4>{w}; }    // the iterate minimises J over the face of the working set in `stat`
  __device__ __forceinline__ RowBit<8> cur_is_a() { return RowBit<8>{w}; }
  // scratch bits of box_qp_iterate (meaningless between calls)
  __device__ __forceinline__ RowBit<16> was_busy() { return RowBit<16>{w}; }
  __device__ __forceinline__ RowBit<32> going() { return RowBit<32>{w}; }
  __device__ __forceinline__ RowBit<64> moved() { return RowBit<64>{w}; }
  __device__ __forceinline__ RowBit<512> was_pdas() { return RowBit<512>{w}; }
  // second phase of a solve (see box_qp_iterate): primal-dual active-set iterations on the working set alone
  __device__ __forceinline__ RowBit<128> pdas() { return RowBit<128>{w}; }
  __device__ __forceinline__ RowBit<256> pdas_done() { return RowBit<256>{w}; }   // ... was entered once in this solve
  __device__ __forceinline__ void settle() { asm volatile("" : "+v"(w)); }
  // several bits at once: one v_and + one v_cmp instead of a lane mask per bit
  __device__ __forceinline__ bool is(int mask, int pattern) const { return (w & mask) == pattern; }
  int pdas_its = 0;
  int stalls = 0, iters = 0;
  double Jk = 0.0;
  QpStats stats;
  // (Xa, Ua) hold a feasible point with its linearised trajectory, J its objective
  __device__ __forceinline__ void begin(double J) {
    w = 1 | 2 | 8;
    stalls = 0;
    iters = 0;
    Jk = J;
    stats = QpStats();
  }
};

// Iterations a solve may spend in its primal-dual phase before it falls back (CPU prototypes tests/probes/pdas_proto.py,
// pdas_switch_proto.py: config 3's hard solves - the first warm steps, whose shifted guess is poor - need up to 22 on the members
// tried; config 4's often cycle, and pay for the try).  Measured on config 3, 65,536 members: cap 16 298 ms, 24 278, 40 254, 64 263,
// 100 272 (profiles/r03_exact_qp_log.txt).
#ifndef M4Q_PDAS_CAP
#define M4Q_PDAS_CAP 40
#endif
constexpr int PDAS_CAP = M4Q_PDAS_CAP;

// The pinned sweep of box_qp_iterate on matrix-core tiles instead of DPP rows (m4q_tile3.h: TileBackwardB<..., PINNED>), where the
// kernel provides it: an object with `enabled` and sweep(T, win, pin, store_ok).  NoTileSweep: DPP rows.
struct NoTileSweep {
  static constexpr bool enabled = false;
  template <int NU>
  __device__ __forceinline__ void sweep(int, const Window&, const PinCtx<NU>&, bool) const {}
};

// One iteration of the exact solve for the rows of the wavefront that have one in progress (r.busy()).  Returns true for
// the rows whose solve ended in this call: r.Jk is then the objective of the answer, r.cur_is_a() says where it is.
template <class S, int NX, int NU, class Prov, bool TR = false, class TileSweep = NoTileSweep>
__device__ __forceinline__ bool box_qp_iterate(const Prov& prov, int T, S x0, const Window& win, const CostRef<S, TR>& cost, int flags,
                                               const GView& gains, PinCtx<NU>& pin, GView Xa, GView Ua, GView Xb, GView Ub,
                                               BoxQpRow& r, int j, int jj, bool lane_ok, PhaseClock* pc = nullptr,
                                               const TileSweep& tile = TileSweep()) {
  PhaseClock none;
  PhaseClock& clk = pc ? *pc : none;
  r.was_busy() = r.busy();
  int jl = jj;                                       // (lane_ok is recomputed from this after every sweep, for the same reason)
  auto settle = [&]() __attribute__((always_inline)) { r.settle(); asm volatile("" : "+v"(jl)); };
  auto ok = [&]() __attribute__((always_inline)) { return jl < NX; };
  (void)lane_ok;
  // (every sweep of the iteration gets the horizon through an opaque copy: loop bounds and unroll-remainder predicates derived
  //  from it then live inside that sweep instead of across the whole iteration, where they were spilled to VGPR lanes)
  auto Tf = [&]() __attribute__((always_inline)) { int v = T; asm volatile("" : "+s"(v)); return v; };
  const Box& box = pin.box;
  // Rows whose iterate has just changed (or that start a solve) re-derive their working set from the gradient at the iterate
  // (adjoint recursion only); a face minimiser whose working set comes back unchanged has multipliers of the right sign: KKT,
  // done.  ONE such pass per call, at its start, serves the rows that start a solve and the rows the previous call moved: the
  // KKT verdict of a face minimiser normally comes from the policy rollout itself (RolloutInfo::nbad), so a solve still ends
  // in the pass of its last Riccati sweep.  (Round 3 until then: a second pass at the end of the call, 27 % of the launch.)
  auto derive_working_sets = [&]() __attribute__((always_inline)) {
    if (__any(r.busy() && r.need_adj())) {
      pin.derive = r.busy() && r.need_adj();
      pin.Xk = Xa; pin.Uk = Ua;
      pin.Xk.off = r.cur_is_a() ? Xa.off : Xb.off;
      pin.Uk.off = r.cur_is_a() ? Ua.off : Ub.off;
      adjoint_pass<S, NX, NU>(prov, Tf(), win, cost, pin, j, r.busy() && r.need_adj() && ok());
      wave_sync();
      settle();
      const bool adj = r.busy() && r.need_adj();
      if (adj && r.face_min() && pin.nchg == 0) { r.busy() = false; ++r.stats.end_kkt; }
      if (adj) { r.need_adj() = false; r.face_min() = false; }
    }
  };
  clk.mark(15);
  derive_working_sets();                     // (rows that start a solve, rows that moved in the previous call)
  clk.mark(1);
  // per-row source / destination: same wave-uniform bases, lane offsets swapped
  GView Xk = Xa, Uk = Ua, Xc = Xb, Uc = Ub;
  Xk.off = r.cur_is_a() ? Xa.off : Xb.off;
  Uk.off = r.cur_is_a() ? Ua.off : Ub.off;
  Xc.off = r.cur_is_a() ? Xb.off : Xa.off;
  Uc.off = r.cur_is_a() ? Ub.off : Ua.off;
  if (__any(r.busy())) {
    r.going() = r.busy();
    r.moved() = false;
    r.was_pdas() = r.pdas();
    if (r.going()) ++r.iters;
    // (constant target: the sweep's constant-target form, as in the clipped mode - a wave-uniform choice between two instantiations)
    if constexpr (TileSweep::enabled) {
      // (constant target: the pinned sweep on matrix-core tiles, as the clipped mode's backward sweep - m4q_tile3.h)
      // (M4Q_OPT_NO_TILE: the DPP sweep for any target, which this kernel holds anyway)
      if ((flags & (QP_TARG_CONST | QP_NO_TILE)) == QP_TARG_CONST) tile.sweep(Tf(), win, pin, r.going());
      else riccati_backward<S, NX, NU, Prov, true>(prov, Tf(), win, cost, flags, gains, j, r.going() && ok(), &pin);
    } else if constexpr (sizeof(S) == sizeof(double) && NX >= 8) {
      if ((flags & QP_TARG_CONST) != 0) riccati_backward<S, NX, NU, Prov, true, true>(prov, Tf(), win, cost, flags, gains, j, r.going() && ok(), &pin);
      else riccati_backward<S, NX, NU, Prov, true>(prov, Tf(), win, cost, flags, gains, j, r.going() && ok(), &pin);
    } else {
      riccati_backward<S, NX, NU, Prov, true>(prov, Tf(), win, cost, flags, gains, j, r.going() && ok(), &pin);
    }
    wave_sync();
    settle();
    clk.mark(10);
    RolloutInfo ri;
    // (rows in their primal-dual phase: the UNCLIPPED rollout of the face minimiser, which updates the working set as it goes)
    pin.pdas = r.is(32 | 128, 32 | 128);
    const double Jc = rollout_policy<S, NX, NU>(prov, Tf(), x0, win, cost, gains, pin, Uk, !r.is(32 | 128, 32 | 128), Xc, Uc, j,
                                                r.going() && ok(), ri);
    pin.pdas = false;
    wave_sync();
    settle();
    clk.mark(11);
    const bool going = r.going();
    if (going) ++r.stats.sweeps;
    bool moved = false;
    if (going && r.pdas()) {
      // primal-dual phase: nothing changed = the unclipped face minimiser is feasible with multipliers of the right sign: the optimum
      if (ri.nchg == 0) {
        moved = true;
        r.Jk = Jc;
        r.busy() = false;
        ++r.stats.end_kkt;
      } else if (++r.pdas_its >= PDAS_CAP) {
        // (a cycle or a slow crawl: back to the iteration that cannot cycle, from the feasible iterate, which has not moved.  Detecting
        //  cycles of period <= 4 by a signature of the set bought nothing: 258 against 254 ms - the solves that give up crawl)
        r.pdas() = false;
        r.need_adj() = true;
        r.face_min() = false;
      }
    } else if (going && !(ri.dmax > 1e-13 * box.sat)) {
      // the policy reproduces the iterate: it is the minimiser of its face (or NaN)
      if (!(ri.dmax == ri.dmax) || ++r.stalls > 1) { r.busy() = false; ++r.stats.end_precision; }
      else if (!ri.outside && ri.nbad == 0) { r.busy() = false; ++r.stats.end_kkt; }      // face minimiser, multipliers of the right sign
      r.face_min() = true;
      r.need_adj() = true;
    } else if (going && (Jc < r.Jk || (!ri.outside && Jc <= r.Jk + 1e-12 * fabs(r.Jk) && r.stalls < 2))) {
      // (a face minimiser is taken even without a visible decrease: near the optimum J is flat to working precision
      //  long before the controls are, and the Newton point is the more accurate of the two)
      r.stalls = Jc < r.Jk ? 0 : r.stalls + 1;
      moved = true;
      r.Jk = Jc;
      r.face_min() = !ri.outside;
      r.need_adj() = true;
      // nothing clipped: the new iterate minimises J over the face; if every pinned control's multiplier has the right sign it
      // is the optimum (KKT) - known from the rollout itself.  Otherwise the working set is re-derived from the gradient.
      if (!ri.outside && ri.nbad == 0) { r.busy() = false; ++r.stats.end_kkt; }
    } else if (going && !ri.outside) {
      r.busy() = false;                                          // face minimiser without decrease: working precision
      ++r.stats.end_precision;
    }
    // classical step for the rows whose clipped rollout did not decrease J
    r.moved() = moved;
    // rows whose clipped trial did not lower J.  The first time in a solve: the solve enters its primal-dual phase - the unclipped
    // rollout of the same policy makes the first update of the working set; the iterate stays.  After that phase (it gave up):
    // the classical ratio step.
    constexpr int SECOND = 1 | 32 | 64 | 4 | 512;                // busy, going, !moved, !face_min, !was_pdas
    if (__any(r.is(SECOND, 1 | 32))) {
      RolloutInfo rn;
      pin.pdas = r.is(SECOND | 256, 1 | 32);                     // ... and the primal-dual phase not entered yet
      const double Jn = rollout_policy<S, NX, NU>(prov, Tf(), x0, win, cost, gains, pin, Uk, false, Xc, Uc, j, r.is(SECOND, 1 | 32) && ok(), rn);
      pin.pdas = false;
      wave_sync();
      settle();
      clk.mark(12);
      const bool second = r.is(SECOND, 1 | 32);
      const bool enter = second && !r.pdas_done();
      if (enter) {
        r.pdas_done() = true;
        if (rn.nchg == 0) {                                      // (cannot happen after a clipped trial; if it does, this is the optimum)
          r.moved() = true;
          r.Jk = Jn;
          r.busy() = false;
          ++r.stats.end_kkt;
        } else {
          r.pdas() = true;
          r.pdas_its = 1;
        }
      }
      const bool ratio = second && !enter;
      if (ratio) {
        ++r.stats.ratio_steps;
        const double al = rn.alpha;
        // blend in place: trial = iterate + al (Newton - iterate); blocking controls land exactly on their bound
        // (the row's 16 lanes share the contiguous elements; a batch issues all its loads before the first store)
        {
          constexpr int U = 12;
          const int count = (T + 1) * NX;
          for (int e0 = jj; e0 < count; e0 += 16 * U) {
            S xk[U], xn[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
              const int e = e0 + 16 * u < count ? e0 + 16 * u : count - 1;
              xk[u] = Xk.ld<S>(e);
              xn[u] = Xc.ld<S>(e);
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int u = 0; u < U; ++u)
              if (e0 + 16 * u < count) Xc.st<S>(e0 + 16 * u, cadd(xk[u], cscale(csub(xn[u], xk[u]), al)));
          }
        }
        for (int e = jj; e < T * NU; e += 16) {
          const double uk = Uk.ld<double>(e), un = Uc.ld<double>(e);
          double v = fma(al, un - uk, uk);
          double lo, hi;
          box.at<NU>(e / NU, e % NU, pin.lo0, pin.hi0, lo, hi);
          const bool above = un > hi, below = un < lo;
          if ((above || below) && ((above ? hi : lo) - uk) / (un - uk) <= al + 1e-14) {
            v = above ? hi : lo;                               // blocking: lands exactly on its bound and is pinned
            pin.stat.template st<double>(e, above ? 1.0 : -1.0);
          }
          Uc.st<double>(e, v);
        }
        const double q = 1.0 - al;
        r.Jk = fma(r.Jk - Jn, q * q, Jn);                      // J along the segment to the face minimiser
        r.moved() = true;
        // al == 0 (degenerate): free controls sit on the bound the step wants to cross; they are pinned now, nothing moved
        r.stalls = al > 0.0 ? 0 : r.stalls + 1;
        if (r.stalls > 2 * NU * T) { r.busy() = false; ++r.stats.end_cap; }
      }
    }
    clk.mark(13);
    if (r.going() && r.moved()) r.cur_is_a() = !r.cur_is_a();
    if (r.busy() && r.iters >= 100 + 2 * NU * T) { r.busy() = false; ++r.stats.end_cap; }   // (the path adds at least one control per step)
    wave_sync();
  }
  return r.was_busy() && !r.busy();
}

// ---------------------------------------------------------------------------------------------
// Line search of mpc.iqp_line_search (mpc.py:101-125), as written: Z stacks
// [Re X.flatten(), Im X.flatten(), Re U.flatten(), Im U.flatten()] with X flattened row-major
// over (n, T+1), while the cost is block-diagonal with one 2n x 2n block per horizon node and one
// 2m x 2m block per control.  Cq/Cqf/Cr are the symmetrised real blocks (host-built).
// jj is the UNCLAMPED lane in the row: all 16 lanes take part.  Views are positioned at element 0
// of the instance (no lane term).
// ---------------------------------------------------------------------------------------------
template <int NX, int NU>
struct ZView {
  int T;
  GView Xg, Xo, Xt;   // [T+1][NX]
  GView Ug, Uo, Ut;   // [T][NU]
  // returns (guess - target, opt - guess) at flat index q of Z
  __device__ __forceinline__ void at(int q, double& e, double& d) const {
    const int nxt = NX * (T + 1);
    double g = 0.0, o = 0.0, tg = 0.0;
    if (q < 2 * nxt) {
      const bool im = q >= nxt;
      const int f = im ? q - nxt : q;
      const int i = f / (T + 1);
      const int t = f - i * (T + 1);
      const unsigned idx = 2 * (t * NX + i) + (im ? 1 : 0);
      g = Xg.ld<double>(idx);
      o = Xo.ld<double>(idx);
      tg = Xt.ld<double>(idx);
    } else {
      const int f = q - 2 * nxt;
      if (f < NU * T) {
        const int k = f / T;
        const int t = f - k * T;
        g = Ug.ld<double>(t * NU + k);
        o = Uo.ld<double>(t * NU + k);
        tg = Ut.ld<double>(t * NU + k);
      }
    }
    e = g - tg;
    d = o - g;
  }
};

template <int S, class ZV>
__device__ __forceinline__ void ls_block(const ZV& z, int base, const double* C, int jj, double& num, double& den,
                                         double& nrm) {
  // rows jj and jj+16 of an S x S block (S <= 32)
  constexpr int NR = (S + 15) / 16;
  double e[NR], d[NR];
#pragma unroll
  for (int r = 0; r < NR; ++r) {
    const int row = jj + 16 * r;
    e[r] = 0.0;
    d[r] = 0.0;
    if (row < S) z.at(base + row, e[r], d[r]);
  }
  double ye[NR], yd[NR];
#pragma unroll
  for (int r = 0; r < NR; ++r) { ye[r] = 0.0; yd[r] = 0.0; }
  static_for<0, S>([&](auto cc) {
    constexpr int c = decltype(cc)::value;
#pragma unroll
    for (int r = 0; r < NR; ++r) {
      const int row = jj + 16 * r;
      const double w = (row < S) ? C[row * S + c] : 0.0;
      fmac_bc<c % 16>(ye[r], e[c / 16], w);
      fmac_bc<c % 16>(yd[r], d[c / 16], w);
    }
  });
#pragma unroll
  for (int r = 0; r < NR; ++r) {
    num = fma(ye[r], d[r], num);
    den = fma(yd[r], d[r], den);
    nrm = fma(d[r], d[r], nrm);
  }
}

// Fast path when every cost block is diagonal (real diagonal Q, Qf, R - the case of all reference
// scenarios): the block products collapse to one weight per Z entry.  Each lane walks ITS state element
// over the horizon in natural (coalesced) order and looks the weight of the Z slot(s) that element lands
// in up in a 2n-entry table, so the reference's layout quirk costs a few integer divisions per element.
// wq/wqf: 2*NX weights, wr: 2*NU weights (LDS).  S = double: the lane's element is a Hermitian-basis
// coordinate that feeds two Z slots (Re or Im of rho_ab and rho_ba) with coefficient 1/sqrt2 each.
template <class S, int NX, int NU, int D>
__device__ __forceinline__ void line_search_diag(const ZView<NX, NU>& z, const double* wq, const double* wqf,
                                                 const double* wr, int jj, double& alpha, double& step_norm) {
  const int T = z.T;
  const int nxt = NX * (T + 1);
  double num = 0.0, den = 0.0, nrm = 0.0;
  auto weight = [&](int q) {
    const int blk = q / (2 * NX);
    const int r = q - blk * (2 * NX);
    return (blk == T ? wqf : wq)[r];
  };
  if (jj < NX) {
    const int a = jj / D, b = jj - (jj / D) * D;
    const int partner = b * D + a;
#pragma unroll 4
    for (int t = 0; t <= T; ++t) {
      const S g = z.Xg.template ld<S>(t * NX + jj), o = z.Xo.template ld<S>(t * NX + jj), tg = z.Xt.template ld<S>(t * NX + jj);
      const int q0 = jj * (T + 1) + t;          // Z slot of Re x_jj'; Im sits nxt further
      if constexpr (sizeof(S) == sizeof(cplx)) {
        const cplx gc = as_cplx(g), oc = as_cplx(o), tc = as_cplx(tg);
        const double w0 = weight(q0), w1 = weight(q0 + nxt);
        const double e0 = gc.re - tc.re, d0 = oc.re - gc.re, e1 = gc.im - tc.im, d1 = oc.im - gc.im;
        num = fma(w0 * e0, d0, fma(w1 * e1, d1, num));
        den = fma(w0 * d0, d0, fma(w1 * d1, d1, den));
        nrm = fma(d0, d0, fma(d1, d1, nrm));
      } else {
        const double e = real_of(g) - real_of(tg), d = real_of(o) - real_of(g);
        const int off = a > b ? nxt : 0;        // antisymmetric coordinates live in the imaginary halves
        const double w = a == b ? weight(q0) : 0.5 * (weight(q0 + off) + weight(partner * (T + 1) + t + off));
        num = fma(w * e, d, num);
        den = fma(w * d, d, den);
        nrm = fma(d, d, nrm);
      }
    }
  }
  // controls: Z = [U.flatten() (k-major over (m, T)), zeros]; block b covers 2m consecutive slots
  for (int f = jj; f < NU * T; f += 16) {
    const int k = f / T;
    const int t = f - k * T;
    const double g = z.Ug.template ld<double>(t * NU + k), o = z.Uo.template ld<double>(t * NU + k),
                 tg = z.Ut.template ld<double>(t * NU + k);
    const double w = wr[f % (2 * NU)];
    const double e = g - tg, d = o - g;
    num = fma(w * e, d, num);
    den = fma(w * d, d, den);
    nrm = fma(d, d, nrm);
  }
  num = rowsum<16>(num);
  den = rowsum<16>(den);
  nrm = rowsum<16>(nrm);
  alpha = -num / den;
  step_norm = fabs(alpha) * sqrt(nrm);
}

// The same diagonal-cost line search when the trajectories are held in the traceless coordinates (n_s = NX - 1 per node).
// The reference's sums run over the slots of vec(rho): lane c < NX takes slot c.  An off-diagonal slot pair is one traceless
// coordinate each (as above); a diagonal slot (a, a) is the combination sum_l O[a][l] r_l of the d - 1 traceless diagonal
// coordinates (the trace coordinate of guess, solution and target is the same number: it drops out of e and d).
template <int NX, int NU, int D>
__device__ __forceinline__ void line_search_tl(const ZView<NX - 1, NU>& z, const double* wq, const double* wqf, const double* wr,
                                               int jj, double& alpha, double& step_norm) {
  constexpr int NS = NX - 1, NO = NX - D;        // traceless coordinates per node; off-diagonal slots per node
  const int T = z.T;
  const int nxt = NX * (T + 1);
  double num = 0.0, den = 0.0, nrm = 0.0;
  auto weight = [&](int q) {
    const int blk = q / (2 * NX);
    const int r = q - blk * (2 * NX);
    return (blk == T ? wqf : wq)[r];
  };
  // All 16 lanes of the row share the (node, slot) pairs, and every batch of U pairs per lane issues ALL its loads before the
  // first use: the pass is bound by memory latency (the trajectories come from beyond the L2), and left to the compiler an
  // unrolled loop waits for each pair's loads in turn (measured: 60,000 cycles per line search, a third of a QP solve;
  // profiles/r03_phase_clock.txt).
  // (a) off-diagonal slots: one traceless coordinate each.  i-th off-diagonal slot of a node: c = i + 1 + i / D.
  {
    constexpr int U = 8;
    const int count = NO * (T + 1);
    for (int f0 = jj; f0 < count; f0 += 16 * U) {
      double g[U], o[U], tg[U];
      int cc[U], tt[U];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int f = f0 + 16 * u < count ? f0 + 16 * u : count - 1;
        const int t = f / NO, i = f - t * NO;
        const int c = i + 1 + i / D;
        cc[u] = c; tt[u] = t;
        g[u] = z.Xg.template ld<double>(t * NS + c - 1);
        o[u] = z.Xo.template ld<double>(t * NS + c - 1);
        tg[u] = z.Xt.template ld<double>(t * NS + c - 1);
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int c = cc[u], t = tt[u], a = c / D, b = c - a * D;
        const double e = g[u] - tg[u], d = o[u] - g[u];
        const int q0 = c * (T + 1) + t;           // Z slot of Re x_c; Im sits nxt further
        const int off = a > b ? nxt : 0;          // antisymmetric coordinates live in the imaginary halves
        double w = 0.5 * (weight(q0 + off) + weight((b * D + a) * (T + 1) + t + off));
        w = f0 + 16 * u < count ? w : 0.0;
        num = fma(w * e, d, num);
        den = fma(w * d, d, den);
        nrm = fma(f0 + 16 * u < count ? d : 0.0, d, nrm);
      }
    }
  }
  // (b) diagonal slots (a, a): the combination sum_l O[a][l] r_l of the d - 1 traceless diagonal coordinates
  {
    constexpr int U = 4;
    const int count = D * (T + 1);
    for (int f0 = jj; f0 < count; f0 += 16 * U) {
      double g[U][D - 1], o[U][D - 1], tg[U][D - 1];
      int aa[U], tt[U];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int f = f0 + 16 * u < count ? f0 + 16 * u : count - 1;
        const int t = f / D, a = f - t * D;
        aa[u] = a; tt[u] = t;
#pragma unroll
        for (int l = 1; l < D; ++l) {
          g[u][l - 1] = z.Xg.template ld<double>(t * NS + l * D + l - 1);
          o[u][l - 1] = z.Xo.template ld<double>(t * NS + l * D + l - 1);
          tg[u][l - 1] = z.Xt.template ld<double>(t * NS + l * D + l - 1);
        }
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int a = aa[u], t = tt[u];
        double gs = 0.0, os = 0.0, ts = 0.0;
#pragma unroll
        for (int l = 1; l < D; ++l) {
          const double cf = tl_coef<D>(a, l);
          gs = fma(cf, g[u][l - 1], gs);
          os = fma(cf, o[u][l - 1], os);
          ts = fma(cf, tg[u][l - 1], ts);
        }
        const double e = gs - ts, d = os - gs;
        const double w = f0 + 16 * u < count ? weight((a * D + a) * (T + 1) + t) : 0.0;
        num = fma(w * e, d, num);
        den = fma(w * d, d, den);
        nrm = fma(f0 + 16 * u < count ? d : 0.0, d, nrm);
      }
    }
  }
  // (c) controls: Z = [U.flatten() (k-major over (m, T)), zeros]; block b covers 2m consecutive slots
  {
    constexpr int U = 8;
    const int count = NU * T;
    for (int f0 = jj; f0 < count; f0 += 16 * U) {
      double g[U], o[U], tg[U];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int f = f0 + 16 * u < count ? f0 + 16 * u : count - 1;
        const int k = f / T, t = f - k * T;
        g[u] = z.Ug.template ld<double>(t * NU + k);
        o[u] = z.Uo.template ld<double>(t * NU + k);
        tg[u] = z.Ut.template ld<double>(t * NU + k);
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int f = f0 + 16 * u;
        const double w = f < count ? wr[f % (2 * NU)] : 0.0;
        const double e = g[u] - tg[u], d = o[u] - g[u];
        num = fma(w * e, d, num);
        den = fma(w * d, d, den);
        nrm = fma(f < count ? d : 0.0, d, nrm);
      }
    }
  }
  num = rowsum<16>(num);
  den = rowsum<16>(den);
  nrm = rowsum<16>(nrm);
  alpha = -num / den;
  step_norm = fabs(alpha) * sqrt(nrm);
}

// line_search_tl with one trajectory NODE per lane: a lane loads the n_s coordinates of guess, solution and target of its node as
// 16-byte pairs (n_s even) and forms every slot of that node - a third of the vector-memory instructions of the slot-per-lane form
// above, whose loads are single coordinates picked out of the nodes (the off-diagonal coordinates of a node are not contiguous).
// Same terms, same weights; the sums run in a different order.  One node per lane and trip: two or three in flight at once cost
// the kernel 78 / 219 spilled registers (32.2 / 37.4 ms against 29.95; profiles/r04_ab_experiments.txt).
// TCONST: the target is the same at every node (the caller has seen QP_TARG_CONST): loaded once.
template <int NX, int NU, int D, bool TCONST = false>
__device__ __forceinline__ void line_search_tl_nodes(const ZView<NX - 1, NU>& z, const double* wq, const double* wqf, const double* wr,
                                                     int jj, double& alpha, double& step_norm) {
  constexpr int NS = NX - 1;
  static_assert(NS % 2 == 0, "node-per-lane line search: 16-byte pairs");
  const int T = z.T;
  const int nxt = NX * (T + 1);
  double num = 0.0, den = 0.0, nrm = 0.0;
  auto weight = [&](int q) {
    const int blk = q / (2 * NX);
    const int r = q - blk * (2 * NX);
    return (blk == T ? wqf : wq)[r];
  };
  constexpr int ROUNDS = 1;
  double tgc[TCONST ? NS : 1];
  if constexpr (TCONST) {
#pragma unroll
    for (int h = 0; h < NS / 2; ++h) {
      double c2[2];
      ldn<2>(z.Xt, (unsigned)(2 * h), c2);
      tgc[2 * h] = c2[0]; tgc[2 * h + 1] = c2[1];
    }
  }
  for (int t0 = jj; t0 <= T; t0 += 16 * ROUNDS) {
    double g[ROUNDS][NS], o[ROUNDS][NS], tg[ROUNDS][NS];
#pragma unroll
    for (int u = 0; u < ROUNDS; ++u) {
      const int t = t0 + 16 * u <= T ? t0 + 16 * u : T;
#pragma unroll
      for (int h = 0; h < NS / 2; ++h) {
        double a2[2], b2[2], c2[2];
        ldn<2>(z.Xg, (unsigned)(t * NS + 2 * h), a2);
        ldn<2>(z.Xo, (unsigned)(t * NS + 2 * h), b2);
        if constexpr (TCONST) { c2[0] = tgc[2 * h]; c2[1] = tgc[2 * h + 1]; }
        else ldn<2>(z.Xt, (unsigned)(t * NS + 2 * h), c2);
        g[u][2 * h] = a2[0]; g[u][2 * h + 1] = a2[1];
        o[u][2 * h] = b2[0]; o[u][2 * h + 1] = b2[1];
        tg[u][2 * h] = c2[0]; tg[u][2 * h + 1] = c2[1];
      }
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int u = 0; u < ROUNDS; ++u) {
      const int t = t0 + 16 * u;
      const bool live = t <= T;
      // off-diagonal slots c = a D + b, a != b: traceless coordinate c - 1
#pragma unroll
      for (int c = 1; c < NX; ++c) {
        const int a = c / D, b = c - a * D;
        if (a == b) continue;
        const double e = g[u][c - 1] - tg[u][c - 1], d = o[u][c - 1] - g[u][c - 1];
        const int off = a > b ? nxt : 0;            // antisymmetric coordinates live in the imaginary halves
        double w = 0.5 * (weight(c * (T + 1) + (live ? t : T) + off) + weight((b * D + a) * (T + 1) + (live ? t : T) + off));
        w = live ? w : 0.0;
        num = fma(w * e, d, num);
        den = fma(w * d, d, den);
        nrm = fma(live ? d : 0.0, d, nrm);
      }
      // diagonal slots (a, a): sum_l O[a][l] r_l over the d - 1 traceless diagonal coordinates
#pragma unroll
      for (int a = 0; a < D; ++a) {
        double gs = 0.0, os = 0.0, ts = 0.0;
#pragma unroll
        for (int l = 1; l < D; ++l) {
          const double cf = tl_coef<D>(a, l);
          gs = fma(cf, g[u][l * D + l - 1], gs);
          os = fma(cf, o[u][l * D + l - 1], os);
          ts = fma(cf, tg[u][l * D + l - 1], ts);
        }
        const double e = gs - ts, d = os - gs;
        const double w = live ? weight((a * D + a) * (T + 1) + t) : 0.0;
        num = fma(w * e, d, num);
        den = fma(w * d, d, den);
        nrm = fma(live ? d : 0.0, d, nrm);
      }
    }
  }
  // controls: Z = [U.flatten() (k-major over (m, T)), zeros]; block b covers 2m consecutive slots
  {
    constexpr int U = 8;
    const int count = NU * T;
    for (int f0 = jj; f0 < count; f0 += 16 * U) {
      double g[U], o[U], tg[U];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int f = f0 + 16 * u < count ? f0 + 16 * u : count - 1;
        const int k = f / T, t = f - k * T;
        g[u] = z.Ug.template ld<double>(t * NU + k);
        o[u] = z.Uo.template ld<double>(t * NU + k);
        tg[u] = z.Ut.template ld<double>(t * NU + k);
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int f = f0 + 16 * u;
        const double w = f < count ? wr[f % (2 * NU)] : 0.0;
        const double e = g[u] - tg[u], d = o[u] - g[u];
        num = fma(w * e, d, num);
        den = fma(w * d, d, den);
        nrm = fma(f < count ? d : 0.0, d, nrm);
      }
    }
  }
  num = rowsum<16>(num);
  den = rowsum<16>(den);
  nrm = rowsum<16>(nrm);
  alpha = -num / den;
  step_norm = fabs(alpha) * sqrt(nrm);
}

template <int NX, int NU>
__device__ __forceinline__ void line_search(const ZView<NX, NU>& z, const double* Cq, const double* Cqf,
                                            const double* Cr, int jj, double& alpha, double& step_norm) {
  const int T = z.T;
  double num = 0.0, den = 0.0, nrm = 0.0;
  for (int b = 0; b <= T; ++b) ls_block<2 * NX>(z, 2 * NX * b, b == T ? Cqf : Cq, jj, num, den, nrm);
  const int ubase = 2 * NX * (T + 1);
  for (int b = 0; b < T; ++b) ls_block<2 * NU>(z, ubase + 2 * NU * b, Cr, jj, num, den, nrm);
  num = rowsum<16>(num);
  den = rowsum<16>(den);
  nrm = rowsum<16>(nrm);
  alpha = -num / den;                                  // mpc.py:121
  step_norm = fabs(alpha) * sqrt(nrm);                 // mpc.py:122
}

// ---------------------------------------------------------------------------------------------
// expm of a column-owned N x N complex matrix: scaling and squaring with the [13/13] Pade
// approximant (Higham 2005, Alg. 2.3, always degree 13), linear solve by Gauss-Jordan with
// partial pivoting done with selects (register files cannot be indexed by a run-time row).
// ---------------------------------------------------------------------------------------------
template <int N>
__device__ __forceinline__ void expm_cols(cplx (&A)[N], int j) {
  const double theta13 = 5.371920351148152;
  double cs = 0.0;
#pragma unroll
  for (int i = 0; i < N; ++i) cs += sqrt(A[i].re * A[i].re + A[i].im * A[i].im);
  const double nrm = rowmax<N>(cs);
  int s = 0;
  if (finite_d(nrm) && nrm > theta13) s = ilogb(nrm / theta13) + 1;   // NaN/inf: the result is NaN, caught upstream
  if (s > 60) s = 60;
  const double sc = ldexp(1.0, -s);
#pragma unroll
  for (int i = 0; i < N; ++i) A[i] = cscale(A[i], sc);

  const double b[14] = {64764752532480000.0, 32382376266240000.0, 7771770303897600.0, 1187353796428800.0,
                        129060195264000.0,   10559470521600.0,    670442572800.0,     33522128640.0,
                        1323241920.0,        40840800.0,          960960.0,           16380.0,
                        182.0,               1.0};
  cplx A2[N], A4[N], A6[N], W[N], Um[N], Vm[N];
  matmul_cols<N>(A2, A, A);
  matmul_cols<N>(A4, A2, A2);
  matmul_cols<N>(A6, A4, A2);
#pragma unroll
  for (int i = 0; i < N; ++i)
    W[i] = mk(b[13] * A6[i].re + b[11] * A4[i].re + b[9] * A2[i].re, b[13] * A6[i].im + b[11] * A4[i].im + b[9] * A2[i].im);
  matmul_cols<N>(Vm, A6, W);
#pragma unroll
  for (int i = 0; i < N; ++i) {
    const double id = (i == j) ? b[1] : 0.0;
    W[i] = mk(Vm[i].re + b[7] * A6[i].re + b[5] * A4[i].re + b[3] * A2[i].re + id,
              Vm[i].im + b[7] * A6[i].im + b[5] * A4[i].im + b[3] * A2[i].im);
  }
  matmul_cols<N>(Um, A, W);                                   // U = A (A6 (b13 A6 + b11 A4 + b9 A2) + b7 A6 + ... + b1 I)
#pragma unroll
  for (int i = 0; i < N; ++i)
    W[i] = mk(b[12] * A6[i].re + b[10] * A4[i].re + b[8] * A2[i].re, b[12] * A6[i].im + b[10] * A4[i].im + b[8] * A2[i].im);
  matmul_cols<N>(Vm, A6, W);
#pragma unroll
  for (int i = 0; i < N; ++i) {
    const double id = (i == j) ? b[0] : 0.0;
    Vm[i] = mk(Vm[i].re + b[6] * A6[i].re + b[4] * A4[i].re + b[2] * A2[i].re + id,
               Vm[i].im + b[6] * A6[i].im + b[4] * A4[i].im + b[2] * A2[i].im);
  }
  // solve (V - U) X = (V + U)
  cplx Qm[N], Pm[N];
#pragma unroll
  for (int i = 0; i < N; ++i) {
    Qm[i] = csub(Vm[i], Um[i]);
    Pm[i] = cadd(Vm[i], Um[i]);
  }
  static_for<0, N>([&](auto kk) {
    constexpr int k = decltype(kk)::value;
    cplx ck[N];
#pragma unroll
    for (int i = 0; i < N; ++i) ck[i] = bcast<k>(Qm[i]);      // pivot column, replicated
    int piv = k;
    double best = ck[k].re * ck[k].re + ck[k].im * ck[k].im;
#pragma unroll
    for (int i = k + 1; i < N; ++i) {
      const double mag = ck[i].re * ck[i].re + ck[i].im * ck[i].im;
      if (mag > best) { best = mag; piv = i; }
    }
#pragma unroll
    for (int i = k + 1; i < N; ++i) {
      const bool sw = (piv == i);
      const cplx q_i = Qm[i], p_i = Pm[i], c_i = ck[i];
      Qm[i] = csel(sw, Qm[k], q_i);
      Pm[i] = csel(sw, Pm[k], p_i);
      ck[i] = csel(sw, ck[k], c_i);
      Qm[k] = csel(sw, q_i, Qm[k]);
      Pm[k] = csel(sw, p_i, Pm[k]);
      ck[k] = csel(sw, c_i, ck[k]);
    }
    const double den = 1.0 / (ck[k].re * ck[k].re + ck[k].im * ck[k].im);
    const cplx pinv = mk(ck[k].re * den, -ck[k].im * den);
    Qm[k] = cmul(Qm[k], pinv);
    Pm[k] = cmul(Pm[k], pinv);
#pragma unroll
    for (int i = 0; i < N; ++i) {
      if (i != k) {
        cmac(Qm[i], cneg(ck[i]), Qm[k]);
        cmac(Pm[i], cneg(ck[i]), Pm[k]);
      }
    }
  });
  // undo the scaling: square s times (s differs between the four rows of the wave)
  for (int it = 0; __any(it < s); ++it) {
    cplx X2[N];
    matmul_cols<N>(X2, Pm, Pm);
    const bool on = it < s;
#pragma unroll
    for (int i = 0; i < N; ++i) Pm[i] = csel(on, X2[i], Pm[i]);
  }
#pragma unroll
  for (int i = 0; i < N; ++i) A[i] = Pm[i];
}

// Plant kinds (mirrored in include/m4q.h)
enum : int { PLANT_NONE = 0, PLANT_HAMILTONIAN = 1, PLANT_GENERATOR = 2 };

// rho+ = U rho U^H with U = expm(-i dt (H0 + sum_k u_k H_k)); x = vec_r(rho)  (experiment.py:190-212).
// sc: per-instance LDS scratch of at least D*D + 2*NX complex.  One wave per block: __syncthreads
// is the wave's own LDS fence.
template <int NX, int NU, int D>
__device__ __forceinline__ cplx plant_hamiltonian(cplx x, const double (&u)[NU], const GView& H0, const GView& Hk, double dt,
                                                  cplx* sc, int j, int jj) {
  static_assert(D * D == NX, "state is a vectorised d x d density matrix");
  const int jc = j < D ? j : D - 1;
  cplx G[D];
#pragma unroll
  for (int i = 0; i < D; ++i) {
    cplx hsum = H0.ld<cplx>(i * D + jc);
#pragma unroll
    for (int k = 0; k < NU; ++k) cmac_r(hsum, Hk.ld<cplx>((k * D + i) * D + jc), u[k]);
    G[i] = mk(hsum.im * dt, -hsum.re * dt);              // -i dt H
  }
  expm_cols<D>(G, jc);
  cplx* Us = sc;             // [D][D]
  cplx* xs = sc + D * D;     // [NX]
  cplx* Ms = xs + NX;        // [NX]
  if (jj < D) {
#pragma unroll
    for (int a = 0; a < D; ++a) Us[a * D + jj] = G[a];
  }
  if (jj < NX) xs[jj] = x;
  wave_sync();
  const int a = j / D, e = j - (j / D) * D;
  cplx m = czero();
#pragma unroll
  for (int c = 0; c < D; ++c) cmac(m, Us[a * D + c], xs[c * D + e]);      // (U rho)[a][e]
  if (jj < NX) Ms[jj] = m;
  wave_sync();
  cplx out = czero();
#pragma unroll
  for (int c = 0; c < D; ++c) cmac_cj(out, Us[e * D + c], Ms[a * D + c]); // sum_c (U rho)[a][c] conj(U[e][c])
  wave_sync();
  return out;
}

// x+ = expm(dt (L0 + sum_k u_k L_k)) x for a general generator on vec_r(rho).  The lane owns
// ROW j of L, i.e. column j of L^T; expm(L^T) = expm(L)^T, so it ends up with row j of the propagator.
template <int NX, int NU>
__device__ __forceinline__ cplx plant_generator(cplx x, const double (&u)[NU], const GView& L0, const GView& Lk, double dt,
                                                int j) {
  cplx G[NX];
#pragma unroll
  for (int i = 0; i < NX; ++i) {
    cplx l = L0.ld<cplx>(j * NX + i);
#pragma unroll
    for (int k = 0; k < NU; ++k) cmac_r(l, Lk.ld<cplx>((k * NX + j) * NX + i), u[k]);
    G[i] = cscale(l, dt);
  }
  expm_cols<NX>(G, j);
  return matvec_t<NX>(G, x);
}

}  // namespace m4q
