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

// ---- ordering inside one wavefront -------------------------------------------------------------
// A workgroup is ONE wavefront.  Lanes exchange data through global memory (workspace) and LDS; the
// hardware performs a wave's memory instructions in issue order, so all that is needed between a
// store by one lane and a load by another is that the COMPILER keeps their order: a wavefront-scope
// release/acquire fence emits no instruction (no vmcnt(0) drain, no s_barrier), unlike __syncthreads().
__device__ __forceinline__ void wave_sync() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
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
// available on v_fmac_f64 (VOP2) and v_mov_b64 (VOP1), so the broadcast is folded INTO the FMA:
// acc += lane_K(a) * b is one instruction and no temporary exists.  Measured on MI355X
// (tools/ubench_dpp.hip): v_fmac_f64_dpp issues at the plain v_fma_f64 rate (4.1-4.3 cycles per
// wave-instruction per SIMD, 78 TFLOP/s chip-wide); an unfused v_mov_b64_dpp + FMA pair runs at 0.66x.
// v_add/max/mul_f64 have no VOP2 encoding on gfx9, hence no DPP form.
// A VALU write of a VGPR needs 2 wait states before a DPP read of it and hipcc pads nothing inside an
// asm string (cdna_hip_programming.md 5.7 item 2): every asm statement opens with `s_nop 1`; the
// multi-term statements of m4q_dpp_gen.h pay it once per 8-16 FMAs.
#define M4Q_DPP " row_mask:0xf bank_mask:0xf\n\t"

#ifndef M4Q_BCAST_SHFL
template <int K>
__device__ __forceinline__ double bcast(double x) {
  double r;
  asm("s_nop 1\n\tv_mov_b64_dpp %0, %1 row_newbcast:%2" M4Q_DPP : "=v"(r) : "v"(x), "n"(K));
  return r;
}
#else
template <int K>
__device__ __forceinline__ double bcast(double x) { return __shfl(x, K, 16); }
#endif
template <int K>
__device__ __forceinline__ cplx bcast(cplx x) {
  return mk(bcast<K>(x.re), bcast<K>(x.im));
}

}  // namespace m4q
#include "m4q_dpp_gen.h"
namespace m4q {

// ---- scalar-generic layer: the same algorithms run on S = cplx (any model) and S = double (models that
// preserve Hermiticity, expressed in a Hermitian operator basis where everything is real) ----
__device__ __forceinline__ double cadd(double a, double b) { return a + b; }
__device__ __forceinline__ double csub(double a, double b) { return a - b; }
__device__ __forceinline__ double cneg(double a) { return -a; }
__device__ __forceinline__ double cconj(double a) { return a; }
__device__ __forceinline__ double cscale(double a, double s) { return a * s; }
__device__ __forceinline__ double cmul(double a, double b) { return a * b; }
__device__ __forceinline__ void cmac(double& acc, double a, double b) { acc = fma(a, b, acc); }
__device__ __forceinline__ void cmac_cj(double& acc, double a, double b) { acc = fma(a, b, acc); }
__device__ __forceinline__ void cmac_r(double& acc, double a, double s) { acc = fma(a, s, acc); }
__device__ __forceinline__ double csel(bool c, double a, double b) { return c ? a : b; }
__device__ __forceinline__ double real_of(double a) { return a; }
__device__ __forceinline__ double real_of(cplx a) { return a.re; }
__device__ __forceinline__ double norm2(double a) { return a * a; }
__device__ __forceinline__ double norm2(cplx a) { return fma(a.re, a.re, a.im * a.im); }
// Re(conj(a) b)
__device__ __forceinline__ double dot_re(double a, double b) { return a * b; }
__device__ __forceinline__ double dot_re(cplx a, cplx b) { return fma(a.re, b.re, a.im * b.im); }
template <class S> __device__ __forceinline__ S zero_of();
template <> __device__ __forceinline__ double zero_of<double>() { return 0.0; }
template <> __device__ __forceinline__ cplx zero_of<cplx>() { return czero(); }
__device__ __forceinline__ cplx as_cplx(cplx a) { return a; }
__device__ __forceinline__ cplx as_cplx(double a) { return mk(a, 0.0); }
template <class S> __device__ __forceinline__ S from_cplx(cplx a);
template <> __device__ __forceinline__ cplx from_cplx<cplx>(cplx a) { return a; }
template <> __device__ __forceinline__ double from_cplx<double>(cplx a) { return a.re; }
template <class S> __device__ __forceinline__ S from_real(double r);
template <> __device__ __forceinline__ double from_real<double>(double r) { return r; }
template <> __device__ __forceinline__ cplx from_real<cplx>(double r) { return mk(r, 0.0); }

// N-term statements, uniform over S (the conjugation flags are meaningless for double)
template <bool CA, bool CB, int K0>
__device__ __forceinline__ void macN(double& c0, double a0, double b0) { fmacN<K0>(c0, a0, b0); }
template <bool CA, bool CB, int K0, int K1>
__device__ __forceinline__ void macN(double& c0, double a0, double b0, double& c1, double a1, double b1) {
  fmacN<K0, K1>(c0, a0, b0, c1, a1, b1);
}
template <bool CA, bool CB, int K0, int K1, int K2>
__device__ __forceinline__ void macN(double& c0, double a0, double b0, double& c1, double a1, double b1, double& c2, double a2,
                                     double b2) {
  fmacN<K0, K1, K2>(c0, a0, b0, c1, a1, b1, c2, a2, b2);
}
template <bool CA, bool CB, int K0, int K1, int K2, int K3>
__device__ __forceinline__ void macN(double& c0, double a0, double b0, double& c1, double a1, double b1, double& c2, double a2,
                                     double b2, double& c3, double a3, double b3) {
  fmacN<K0, K1, K2, K3>(c0, a0, b0, c1, a1, b1, c2, a2, b2, c3, a3, b3);
}
template <bool CA, bool CB, int K0>
__device__ __forceinline__ void macN(cplx& c0, cplx a0, cplx b0) { cmacN<CA, CB, K0>(c0, a0, b0); }
template <bool CA, bool CB, int K0, int K1>
__device__ __forceinline__ void macN(cplx& c0, cplx a0, cplx b0, cplx& c1, cplx a1, cplx b1) {
  cmacN<CA, CB, K0, K1>(c0, a0, b0, c1, a1, b1);
}
template <bool CA, bool CB, int K0, int K1, int K2>
__device__ __forceinline__ void macN(cplx& c0, cplx a0, cplx b0, cplx& c1, cplx a1, cplx b1, cplx& c2, cplx a2, cplx b2) {
  cmacN<CA, CB, K0, K1, K2>(c0, a0, b0, c1, a1, b1, c2, a2, b2);
}
template <bool CA, bool CB, int K0, int K1, int K2, int K3>
__device__ __forceinline__ void macN(cplx& c0, cplx a0, cplx b0, cplx& c1, cplx a1, cplx b1, cplx& c2, cplx a2, cplx b2,
                                     cplx& c3, cplx a3, cplx b3) {
  cmacN<CA, CB, K0, K1, K2, K3>(c0, a0, b0, c1, a1, b1, c2, a2, b2, c3, a3, b3);
}

// single-term conveniences
template <int K, class S>
__device__ __forceinline__ void cmac_bc(S& acc, S a, S b) { macN<false, false, K>(acc, a, b); }       // acc += lane_K(a) * b
template <int K, class S>
__device__ __forceinline__ void cmac_cjbc(S& acc, S a, S b) { macN<true, false, K>(acc, a, b); }      // acc += conj(lane_K(a)) * b
template <int K>
__device__ __forceinline__ void fmac_bc(double& acc, double a, double b) { fmacN<K>(acc, a, b); }

// acc[i] += opA(lane_K(a[i])) * opB(b)   for i in [I0, N): same lane, a chunk of up to four per statement
template <bool CA, bool CB, int K, int I0, int N, class S>
__device__ __forceinline__ void mac_same_lane(S (&acc)[N], const S (&a)[N], S b) {
  if constexpr (N - I0 >= 4) {
    macN<CA, CB, K, K, K, K>(acc[I0], a[I0], b, acc[I0 + 1], a[I0 + 1], b, acc[I0 + 2], a[I0 + 2], b, acc[I0 + 3], a[I0 + 3], b);
    mac_same_lane<CA, CB, K, I0 + 4, N>(acc, a, b);
  } else if constexpr (N - I0 == 3) {
    macN<CA, CB, K, K, K>(acc[I0], a[I0], b, acc[I0 + 1], a[I0 + 1], b, acc[I0 + 2], a[I0 + 2], b);
  } else if constexpr (N - I0 == 2) {
    macN<CA, CB, K, K>(acc[I0], a[I0], b, acc[I0 + 1], a[I0 + 1], b);
  } else if constexpr (N - I0 == 1) {
    macN<CA, CB, K>(acc[I0], a[I0], b);
  }
}
// acc[i] += opA(lane_K(a)) * opB(b[i])   same lane, same broadcast source, per-term own operand
template <bool CA, bool CB, int K, int I0, int N, class S>
__device__ __forceinline__ void mac_same_src(S (&acc)[N], S a, const S (&b)[N]) {
  if constexpr (N - I0 >= 4) {
    macN<CA, CB, K, K, K, K>(acc[I0], a, b[I0], acc[I0 + 1], a, b[I0 + 1], acc[I0 + 2], a, b[I0 + 2], acc[I0 + 3], a, b[I0 + 3]);
    mac_same_src<CA, CB, K, I0 + 4, N>(acc, a, b);
  } else if constexpr (N - I0 == 3) {
    macN<CA, CB, K, K, K>(acc[I0], a, b[I0], acc[I0 + 1], a, b[I0 + 1], acc[I0 + 2], a, b[I0 + 2]);
  } else if constexpr (N - I0 == 2) {
    macN<CA, CB, K, K>(acc[I0], a, b[I0], acc[I0 + 1], a, b[I0 + 1]);
  } else if constexpr (N - I0 == 1) {
    macN<CA, CB, K>(acc[I0], a, b[I0]);
  }
}
// acc[i] += opA(lane_i(a)) * opB(b)   for i in [I0, N): the broadcast lane is the row index
template <bool CA, bool CB, int I0, int N, class S>
__device__ __forceinline__ void mac_lane_index(S (&acc)[N], S a, S b) {
  if constexpr (N - I0 >= 4) {
    macN<CA, CB, I0, I0 + 1, I0 + 2, I0 + 3>(acc[I0], a, b, acc[I0 + 1], a, b, acc[I0 + 2], a, b, acc[I0 + 3], a, b);
    mac_lane_index<CA, CB, I0 + 4, N>(acc, a, b);
  } else if constexpr (N - I0 == 3) {
    macN<CA, CB, I0, I0 + 1, I0 + 2>(acc[I0], a, b, acc[I0 + 1], a, b, acc[I0 + 2], a, b);
  } else if constexpr (N - I0 == 2) {
    macN<CA, CB, I0, I0 + 1>(acc[I0], a, b, acc[I0 + 1], a, b);
  } else if constexpr (N - I0 == 1) {
    macN<CA, CB, I0>(acc[I0], a, b);
  }
}
// part[c] += opA(lane_i(v)) * opB(m[i]) over i in [I0, N), four running partial sums
template <bool CA, bool CB, int I0, int N, class S>
__device__ __forceinline__ void dot_lane_index_acc(S (&part)[4], S v, const S (&m)[N]) {
  if constexpr (N - I0 >= 4) {
    macN<CA, CB, I0, I0 + 1, I0 + 2, I0 + 3>(part[0], v, m[I0], part[1], v, m[I0 + 1], part[2], v, m[I0 + 2], part[3], v, m[I0 + 3]);
    dot_lane_index_acc<CA, CB, I0 + 4, N>(part, v, m);
  } else if constexpr (N - I0 == 3) {
    macN<CA, CB, I0, I0 + 1, I0 + 2>(part[0], v, m[I0], part[1], v, m[I0 + 1], part[2], v, m[I0 + 2]);
  } else if constexpr (N - I0 == 2) {
    macN<CA, CB, I0, I0 + 1>(part[0], v, m[I0], part[1], v, m[I0 + 1]);
  } else if constexpr (N - I0 == 1) {
    macN<CA, CB, I0>(part[0], v, m[I0]);
  }
}
// sum_i opA(lane_i(v)) * opB(m[i])
template <bool CA, bool CB, int N, class S>
__device__ __forceinline__ S dot_lane_index(S v, const S (&m)[N]) {
  S part[4] = {zero_of<S>(), zero_of<S>(), zero_of<S>(), zero_of<S>()};
  dot_lane_index_acc<CA, CB, 0, N>(part, v, m);
  return cadd(cadd(part[0], part[1]), cadd(part[2], part[3]));
}

// Sum over lanes 0..N-1 of the row; result replicated in every lane of the row.
template <int I0, int N>
__device__ __forceinline__ void rowsum_acc(double (&p)[8], double v, double one) {
  if constexpr (N - I0 >= 8) {
    fmacN<I0, I0 + 1, I0 + 2, I0 + 3, I0 + 4, I0 + 5, I0 + 6, I0 + 7>(p[0], v, one, p[1], v, one, p[2], v, one, p[3], v, one, p[4],
                                                                      v, one, p[5], v, one, p[6], v, one, p[7], v, one);
    rowsum_acc<I0 + 8, N>(p, v, one);
  } else if constexpr (N - I0 >= 4) {
    fmacN<I0, I0 + 1, I0 + 2, I0 + 3>(p[0], v, one, p[1], v, one, p[2], v, one, p[3], v, one);
    rowsum_acc<I0 + 4, N>(p, v, one);
  } else if constexpr (N - I0 == 3) {
    fmacN<I0, I0 + 1, I0 + 2>(p[4], v, one, p[5], v, one, p[6], v, one);
  } else if constexpr (N - I0 == 2) {
    fmacN<I0, I0 + 1>(p[4], v, one, p[5], v, one);
  } else if constexpr (N - I0 == 1) {
    fmacN<I0>(p[4], v, one);
  }
}
template <int N>
__device__ __forceinline__ double rowsum(double v) {
  double p[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  rowsum_acc<0, N>(p, v, 1.0);
  return ((p[0] + p[1]) + (p[2] + p[3])) + ((p[4] + p[5]) + (p[6] + p[7]));
}
template <int N>
__device__ __forceinline__ cplx rowsum(cplx v) {
  return mk(rowsum<N>(v.re), rowsum<N>(v.im));
}
template <int N>
__device__ __forceinline__ double rowmax(double v) {
  double s = bcast<0>(v);
  static_for<1, N>([&](auto k) { s = fmax(s, bcast<decltype(k)::value>(v)); });
  return s;
}

// C[:, j] = M * B[:, j]   with M, B column-owned (N x N):  C[i] += lane_k(M[i]) * B[k]
template <int N, class S>
__device__ __forceinline__ void matmul_cols(S (&C)[N], const S (&M)[N], const S (&Bc)[N]) {
#pragma unroll
  for (int i = 0; i < N; ++i) C[i] = zero_of<S>();
  static_for<0, N>([&](auto kk) {
    constexpr int k = decltype(kk)::value;
    mac_same_lane<false, false, k, 0, N>(C, M, Bc[k]);
  });
}

// C[:, j] += M^H * B[:, j]   with M, B column-owned:  C[i] += conj(lane_i(M[k])) * B[k]
template <int N, class S>
__device__ __forceinline__ void matmul_cols_hn_acc(S (&C)[N], const S (&M)[N], const S (&Bc)[N]) {
#pragma unroll
  for (int k = 0; k < N; ++k) mac_lane_index<true, false, 0, N>(C, M[k], Bc[k]);
}

// y_j = sum_i conj(M[i][j]) v_i  with M column-owned, v distributed  (= (M^H v)_j; = (M v)_j if M Hermitian)
template <int N, class S>
__device__ __forceinline__ S matvec_h(const S (&M)[N], S v) {
  return dot_lane_index<false, true, N>(v, M);
}
// y_j = sum_i M[i][j] v_i  = (M^T v)_j
template <int N, class S>
__device__ __forceinline__ S matvec_t(const S (&M)[N], S v) {
  return dot_lane_index<false, false, N>(v, M);
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
