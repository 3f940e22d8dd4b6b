// m4q_device.h - device primitives for the batched MPC engine (gfx950 / CDNA4 only).
//
// Execution model: ONE DPP ROW (16 lanes) OWNS ONE MPC INSTANCE, four instances per wavefront.
// An n x n complex matrix (n = d*d <= 16) is distributed by COLUMN: lane j of the row keeps
// column j in VGPRs.  Vectors are distributed one element per lane.  The only cross-lane traffic
// the dense algebra needs is "every lane of the row reads lane k's register", which gfx950 does
// in the register file with the DP-ALU DPP control row_newbcast:k - no LDS round trip.
//
//   C = M * B    (B, C column-owned)   C[i][j] += bcast_k(M[i]) * B[k]      (M column-owned)
//   C = M^H * B                        C[i][j] += conj(bcast_i(M[k])) * B[k]
//   y = M^H v    (v distributed)       y_j     += conj(M[i]) * bcast_i(v)
//
// All 64 lanes stay active for the whole kernel (DPP reads need live source lanes): control flow
// is wave-uniform and per-instance conditions are applied with selects.
#pragma once
#include <hip/hip_runtime.h>
#include <utility>

namespace m4q {

struct cplx {
  double re, im;
};

__device__ __forceinline__ cplx mk(double re, double im) { cplx c; c.re = re; c.im = im; return c; }
__device__ __forceinline__ cplx czero() { return mk(0.0, 0.0); }
__device__ __forceinline__ cplx cadd(cplx a, cplx b) { return mk(a.re + b.re, a.im + b.im); }
__device__ __forceinline__ cplx csub(cplx a, cplx b) { return mk(a.re - b.re, a.im - b.im); }
__device__ __forceinline__ cplx cneg(cplx a) { return mk(-a.re, -a.im); }
__device__ __forceinline__ cplx cconj(cplx a) { return mk(a.re, -a.im); }
__device__ __forceinline__ cplx cscale(cplx a, double s) { return mk(a.re * s, a.im * s); }
__device__ __forceinline__ cplx cmul(cplx a, cplx b) {
  return mk(fma(a.re, b.re, -(a.im * b.im)), fma(a.re, b.im, a.im * b.re));
}
// acc += a * b
__device__ __forceinline__ void cmac(cplx& acc, cplx a, cplx b) {
  acc.re = fma(a.re, b.re, acc.re);
  acc.re = fma(-a.im, b.im, acc.re);
  acc.im = fma(a.re, b.im, acc.im);
  acc.im = fma(a.im, b.re, acc.im);
}
// acc += conj(a) * b
__device__ __forceinline__ void cmac_cj(cplx& acc, cplx a, cplx b) {
  acc.re = fma(a.re, b.re, acc.re);
  acc.re = fma(a.im, b.im, acc.re);
  acc.im = fma(a.re, b.im, acc.im);
  acc.im = fma(-a.im, b.re, acc.im);
}
// acc += a * s  (s real)
__device__ __forceinline__ void cmac_r(cplx& acc, cplx a, double s) {
  acc.re = fma(a.re, s, acc.re);
  acc.im = fma(a.im, s, acc.im);
}
__device__ __forceinline__ cplx csel(bool c, cplx a, cplx b) { return mk(c ? a.re : b.re, c ? a.im : b.im); }

// ---- addressing: wave-uniform base (SGPR pair) + per-lane 32-bit BYTE offset -------------------
// Every global array is reached as  base + zext(off)  so loads/stores use the saddr form and a
// pointer costs one VGPR, not a 64-bit pair per array kept live across the horizon loops.
// Per-lane offsets only ever span the four instances of one quad, far below 4 GiB.
struct GView {
  char* base;      // must be wave-uniform
  unsigned off;    // per-lane, bytes
  template <class T>
  __device__ __forceinline__ T ld(unsigned idx) const {
    return *reinterpret_cast<const T*>(base + (size_t)(off + idx * (unsigned)sizeof(T)));
  }
  template <class T>
  __device__ __forceinline__ void st(unsigned idx, T v) const {
    *reinterpret_cast<T*>(base + (size_t)(off + idx * (unsigned)sizeof(T))) = v;
  }
  // same array, base advanced by a uniform number of elements
  template <class T>
  __device__ __forceinline__ GView shifted(long elems) const {
    GView v; v.base = base + elems * (long)sizeof(T); v.off = off; return v;
  }
  // same array, lane offset advanced
  template <class T>
  __device__ __forceinline__ GView lane(unsigned elems) const {
    GView v; v.base = base; v.off = off + elems * (unsigned)sizeof(T); return v;
  }
};
template <class T>
__device__ __forceinline__ GView gview(const T* p, long uniform_elems, unsigned lane_elems) {
  GView v;
  v.base = reinterpret_cast<char*>(const_cast<T*>(p)) + uniform_elems * (long)sizeof(T);
  v.off = lane_elems * (unsigned)sizeof(T);
  return v;
}

// ---- compile-time loop with an integral_constant index (DPP controls are immediates) ----
template <int I>
using ic = std::integral_constant<int, I>;

template <int B, int E, class F>
__device__ __forceinline__ void static_for(F&& f) {
  if constexpr (B < E) {
    f(ic<B>{});
    static_for<B + 1, E>(f);
  }
}

// ---- row broadcast: every lane of a 16-lane DPP row reads lane K of its own row ----
// DP-ALU DPP (gfx90a+) supports exactly one control for 64-bit operands: row_newbcast:K.  It is
// available on v_fmac_f64 (VOP2) and v_mov_b64 (VOP1), so the broadcast
// is folded INTO the FMA: acc += lane_K(a) * b is one instruction and no temporary exists.
// A VALU write of a VGPR needs 2 wait states before a DPP read of it and hipcc pads nothing inside
// an asm string (cdna_hip_programming.md 5.7 item 2): every block opens with s_nop 1; inside a
// block no instruction DPP-reads a register written by the block.
#define M4Q_DPP " row_mask:0xf bank_mask:0xf\n\t"

#ifndef M4Q_BCAST_SHFL
template <int K>
__device__ __forceinline__ double bcast(double x) {
  double r;
  asm("s_nop 1\n\tv_mov_b64_dpp %0, %1 row_newbcast:%2" M4Q_DPP : "=v"(r) : "v"(x), "n"(K));
  return r;
}
// acc += lane_K(a) * b
template <int K>
__device__ __forceinline__ void fmac_bc(double& acc, double a, double b) {
  asm("s_nop 1\n\tv_fmac_f64_dpp %0, %1, %2 row_newbcast:%3" M4Q_DPP : "+v"(acc) : "v"(a), "v"(b), "n"(K));
}
// acc += lane_K(a).  v_add_f64 / v_max_f64 have no VOP2 encoding on gfx9, hence no DPP form: the add is an
// FMA with a unit multiplier, the max goes through v_mov_b64_dpp.
template <int K>
__device__ __forceinline__ void add_bc(double& acc, double a) {
  fmac_bc<K>(acc, a, 1.0);
}
template <int K>
__device__ __forceinline__ void max_bc(double& acc, double a) {
  acc = fmax(acc, bcast<K>(a));
}
// acc += lane_K(a) * b            (complex; SA/SB/SC/SD are the signs of the four products)
//   re += a.re b.re ; im += a.re b.im ; re += SC a.im b.im ; im += SD a.im b.re
#define M4Q_CMAC_ASM(SC, SD)                                                                             \
  asm("s_nop 1\n\t"                                                                                      \
      "v_fmac_f64_dpp %0, %2, %4 row_newbcast:%6" M4Q_DPP "v_fmac_f64_dpp %1, %2, %5 row_newbcast:%6" M4Q_DPP \
      "v_fmac_f64_dpp %0, " SC "%3, %5 row_newbcast:%6" M4Q_DPP "v_fmac_f64_dpp %1, " SD "%3, %4 row_newbcast:%6" M4Q_DPP \
      : "+v"(acc.re), "+v"(acc.im)                                                                       \
      : "v"(a.re), "v"(a.im), "v"(b.re), "v"(b.im), "n"(K))
template <int K>
__device__ __forceinline__ void cmac_bc(cplx& acc, cplx a, cplx b) {          // acc += lane_K(a) * b
  M4Q_CMAC_ASM("-", "");
}
template <int K>
__device__ __forceinline__ void cmac_cjbc(cplx& acc, cplx a, cplx b) {        // acc += conj(lane_K(a)) * b
  M4Q_CMAC_ASM("", "-");
}
// acc += lane_K(a) * conj(b):  re += a.re b.re + a.im b.im ; im += -a.re b.im + a.im b.re
template <int K>
__device__ __forceinline__ void cmac_bc_cjown(cplx& acc, cplx a, cplx b) {
  asm("s_nop 1\n\t"
      "v_fmac_f64_dpp %0, %2, %4 row_newbcast:%6" M4Q_DPP "v_fmac_f64_dpp %1, -%2, %5 row_newbcast:%6" M4Q_DPP
      "v_fmac_f64_dpp %0, %3, %5 row_newbcast:%6" M4Q_DPP "v_fmac_f64_dpp %1, %3, %4 row_newbcast:%6" M4Q_DPP
      : "+v"(acc.re), "+v"(acc.im)
      : "v"(a.re), "v"(a.im), "v"(b.re), "v"(b.im), "n"(K));
}
#else
// reference implementation of the same primitives through ds_bpermute (debug / A-B builds)
template <int K>
__device__ __forceinline__ double bcast(double x) { return __shfl(x, K, 16); }
template <int K>
__device__ __forceinline__ void fmac_bc(double& acc, double a, double b) { acc = fma(bcast<K>(a), b, acc); }
template <int K>
__device__ __forceinline__ void add_bc(double& acc, double a) { acc += bcast<K>(a); }
template <int K>
__device__ __forceinline__ void max_bc(double& acc, double a) { acc = fmax(acc, bcast<K>(a)); }
template <int K>
__device__ __forceinline__ void cmac_bc(cplx& acc, cplx a, cplx b) { cmac(acc, mk(bcast<K>(a.re), bcast<K>(a.im)), b); }
template <int K>
__device__ __forceinline__ void cmac_cjbc(cplx& acc, cplx a, cplx b) { cmac_cj(acc, mk(bcast<K>(a.re), bcast<K>(a.im)), b); }
template <int K>
__device__ __forceinline__ void cmac_bc_cjown(cplx& acc, cplx a, cplx b) {
  cmac(acc, mk(bcast<K>(a.re), bcast<K>(a.im)), cconj(b));
}
#endif
template <int K>
__device__ __forceinline__ cplx bcast(cplx x) {
  return mk(bcast<K>(x.re), bcast<K>(x.im));
}

// Sum / max over lanes 0..N-1 of the row; result replicated in every lane of the row.
template <int N>
__device__ __forceinline__ double rowsum(double v) {
  double s = 0.0;
  static_for<0, N>([&](auto k) { add_bc<decltype(k)::value>(s, v); });
  return s;
}
template <int N>
__device__ __forceinline__ cplx rowsum(cplx v) {
  return mk(rowsum<N>(v.re), rowsum<N>(v.im));
}
template <int N>
__device__ __forceinline__ double rowmax(double v) {
  double s = bcast<0>(v);
  static_for<1, N>([&](auto k) { max_bc<decltype(k)::value>(s, v); });
  return s;
}

// C[:, j] = M * B[:, j]   with M, B column-owned (N x N):  C[i] += lane_k(M[i]) * B[k]
template <int N>
__device__ __forceinline__ void matmul_cols(cplx (&C)[N], const cplx (&M)[N], const cplx (&Bc)[N]) {
#pragma unroll
  for (int i = 0; i < N; ++i) C[i] = czero();
  static_for<0, N>([&](auto kk) {
    constexpr int k = decltype(kk)::value;
#pragma unroll
    for (int i = 0; i < N; ++i) cmac_bc<k>(C[i], M[i], Bc[k]);
  });
}

// C[:, j] += M^H * B[:, j]   with M, B column-owned:  C[i] += conj(lane_i(M[k])) * B[k]
template <int N>
__device__ __forceinline__ void matmul_cols_hn_acc(cplx (&C)[N], const cplx (&M)[N], const cplx (&Bc)[N]) {
#pragma unroll
  for (int k = 0; k < N; ++k) {
    static_for<0, N>([&](auto ii) {
      constexpr int i = decltype(ii)::value;
      cmac_cjbc<i>(C[i], M[k], Bc[k]);
    });
  }
}

// y_j = sum_i conj(M[i][j]) v_i  with M column-owned, v distributed  (= (M^H v)_j; = (M v)_j if M Hermitian)
template <int N>
__device__ __forceinline__ cplx matvec_h(const cplx (&M)[N], cplx v) {
  cplx y = czero();
  static_for<0, N>([&](auto ii) {
    constexpr int i = decltype(ii)::value;
    cmac_bc_cjown<i>(y, v, M[i]);
  });
  return y;
}

__device__ __forceinline__ bool finite_d(double x) { return __builtin_isfinite(x); }

// Inverse of an M x M Hermitian positive-definite matrix held replicated in every lane
// (g[k][l], k <= l meaningful; diagonal real).  Closed forms for M <= 3.
template <int M>
__device__ __forceinline__ void herm_inverse(const cplx (&g)[M][M], cplx (&inv)[M][M]) {
  if constexpr (M == 1) {
    inv[0][0] = mk(1.0 / g[0][0].re, 0.0);
  } else if constexpr (M == 2) {
    const double a = g[0][0].re, d = g[1][1].re;
    const cplx b = g[0][1];
    const double det = a * d - (b.re * b.re + b.im * b.im);
    const double r = 1.0 / det;
    inv[0][0] = mk(d * r, 0.0);
    inv[1][1] = mk(a * r, 0.0);
    inv[0][1] = mk(-b.re * r, -b.im * r);
    inv[1][0] = cconj(inv[0][1]);
  } else {
    static_assert(M == 3, "dim_u <= 3");
    const double a = g[0][0].re, d = g[1][1].re, f = g[2][2].re;
    const cplx b = g[0][1], c = g[0][2], e = g[1][2];
    const double bb = b.re * b.re + b.im * b.im, cc = c.re * c.re + c.im * c.im, ee = e.re * e.re + e.im * e.im;
    // cofactors of [[a, b, c], [b*, d, e], [c*, e*, f]]
    const double c00 = d * f - ee, c11 = a * f - cc, c22 = a * d - bb;
    const cplx be = cmul(b, e);                  // b e
    const cplx c01 = csub(cmul(c, cconj(e)), cscale(b, f));        // -(b f - c e*)
    const cplx c02 = csub(be, cscale(c, d));                       // b e - c d
    const cplx c12 = csub(cmul(cconj(b), c), cscale(e, a));        // -(a e - b* c)
    const double det = a * c00 + (b.re * c01.re + b.im * c01.im) + (c.re * c02.re + c.im * c02.im);
    // row-0 expansion: det = a C00 + Re(conj(b) c01) + Re(conj(c) c02)
    const double r = 1.0 / det;
    inv[0][0] = mk(c00 * r, 0.0);
    inv[1][1] = mk(c11 * r, 0.0);
    inv[2][2] = mk(c22 * r, 0.0);
    inv[0][1] = cscale(c01, r);
    inv[0][2] = cscale(c02, r);
    inv[1][2] = cscale(c12, r);
    inv[1][0] = cconj(inv[0][1]);
    inv[2][0] = cconj(inv[0][2]);
    inv[2][1] = cconj(inv[1][2]);
  }
}

}  // namespace m4q
