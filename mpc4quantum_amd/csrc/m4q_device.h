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
#include <type_traits>
#include <utility>

// Development-only switches - a debug form of the row broadcast (ds_bpermute instead of DPP), an override of the DPP wait-state
// pad, the per-phase clock, the two-index complex sweep - exist only in builds that say so: build.py passes -DM4Q_DEV for
// `--dev` alone and tools/build_variant.sh (whose libraries never ship) always.  The timing-only ablation paths of round 2
// ("results wrong": M4Q_EXP) are gone from the sources; what they measured is in profiles/r02_ab_experiments.txt.
#if !defined(M4Q_DEV) && (defined(M4Q_BCAST_SHFL) || defined(M4Q_NOP) || defined(M4Q_DEV_PHASE_CLOCK) || defined(M4Q_TWO_INDEX_COMPLEX) || \
                          defined(M4Q_EXP))
#error "development switch without -DM4Q_DEV: use build.py --dev or tools/build_variant.sh"
#endif

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
// The base is typed as a GLOBAL-address-space pointer (m4q_args.h): kernel arguments are read out of the kernarg
// segment where they are used, and a pointer that comes out of memory would otherwise be a flat one.
#if !defined(M4Q_GLOBAL) || !defined(M4Q_KERNEL_TU)
#undef M4Q_GLOBAL
#define M4Q_GLOBAL __attribute__((address_space(1)))
#endif
typedef double d2_t __attribute__((ext_vector_type(2)));

// loads / stores through global pointers (cplx travels as a 16-byte vector: one dwordx4 access)
template <class T> struct GMem;
template <> struct GMem<double> {
  static __device__ __forceinline__ double ld(const M4Q_GLOBAL char* p) { return *reinterpret_cast<const M4Q_GLOBAL double*>(p); }
  static __device__ __forceinline__ void st(M4Q_GLOBAL char* p, double v) { *reinterpret_cast<M4Q_GLOBAL double*>(p) = v; }
};
template <> struct GMem<int> {
  static __device__ __forceinline__ int ld(const M4Q_GLOBAL char* p) { return *reinterpret_cast<const M4Q_GLOBAL int*>(p); }
  static __device__ __forceinline__ void st(M4Q_GLOBAL char* p, int v) { *reinterpret_cast<M4Q_GLOBAL int*>(p) = v; }
};
template <> struct GMem<cplx> {
  static __device__ __forceinline__ cplx ld(const M4Q_GLOBAL char* p) {
    const d2_t v = *reinterpret_cast<const M4Q_GLOBAL d2_t*>(p);
    return mk(v.x, v.y);
  }
  static __device__ __forceinline__ void st(M4Q_GLOBAL char* p, cplx c) {
    d2_t v; v.x = c.re; v.y = c.im;
    *reinterpret_cast<M4Q_GLOBAL d2_t*>(p) = v;
  }
};
// element idx of a global array (wave-uniform or per-lane pointer alike)
template <class T>
__device__ __forceinline__ T gld(const M4Q_GLOBAL T* p, long idx) { return GMem<T>::ld(reinterpret_cast<const M4Q_GLOBAL char*>(p + idx)); }
template <class T>
__device__ __forceinline__ void gst(M4Q_GLOBAL T* p, long idx, T v) { GMem<T>::st(reinterpret_cast<M4Q_GLOBAL char*>(p + idx), v); }

struct GView {
  M4Q_GLOBAL char* base;   // must be wave-uniform
  unsigned off;            // per-lane, bytes
  template <class T>
  __device__ __forceinline__ T ld(unsigned idx) const {
    return GMem<T>::ld(base + (size_t)(off + idx * (unsigned)sizeof(T)));
  }
  template <class T>
  __device__ __forceinline__ void st(unsigned idx, T v) const {
    GMem<T>::st(base + (size_t)(off + idx * (unsigned)sizeof(T)), v);
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
// N doubles at consecutive addresses (the m controls of one horizon index, a lane's m gain entries): ONE 16-byte access when
// N == 2.  Callers guarantee the alignment: every array the kernels address this way starts on a 256-byte boundary, rows are
// whole numbers of N-tuples and idx is a multiple of N.  (Left to the compiler these were two dwordx2 accesses with an address
// add each: it cannot see the alignment.  The backward sweep issues 4 instead of 8 vector-memory instructions per horizon index
// at (n, m) = (8, 2), the rollout 7 instead of 13.)
template <int N>
__device__ __forceinline__ void ldn(const GView& v, unsigned idx, double (&out)[N]) {
  if constexpr (N == 2) {
    const d2_t w = *reinterpret_cast<const M4Q_GLOBAL d2_t*>(v.base + (size_t)(v.off + idx * 8u));
    out[0] = w.x; out[1] = w.y;
  } else {
#pragma unroll
    for (int k = 0; k < N; ++k) out[k] = v.ld<double>(idx + k);
  }
}
template <int N>
__device__ __forceinline__ void stn(const GView& v, unsigned idx, const double (&in)[N]) {
  if constexpr (N == 2) {
    d2_t w; w.x = in[0]; w.y = in[1];
    *reinterpret_cast<M4Q_GLOBAL d2_t*>(v.base + (size_t)(v.off + idx * 8u)) = w;
  } else {
#pragma unroll
    for (int k = 0; k < N; ++k) v.st<double>(idx + k, in[k]);
  }
}
// the same for an array of S = double or cplx elements (cplx: one 16-byte access per element, as before)
template <int N>
__device__ __forceinline__ void ldn(const GView& v, unsigned idx, cplx (&out)[N]) {
#pragma unroll
  for (int k = 0; k < N; ++k) out[k] = v.ld<cplx>(idx + k);
}
template <int N>
__device__ __forceinline__ void stn(const GView& v, unsigned idx, const cplx (&in)[N]) {
#pragma unroll
  for (int k = 0; k < N; ++k) v.st<cplx>(idx + k, in[k]);
}

template <class T>
__device__ __forceinline__ GView gview(const M4Q_GLOBAL T* p, long uniform_elems, unsigned lane_elems) {
  GView v;
  v.base = (M4Q_GLOBAL char*)p + uniform_elems * (long)sizeof(T);
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

// ---- development builds (-DM4Q_DEV_PHASE_CLOCK): wavefront time per phase of the persistent loop, summed on the constant
// 100 MHz clock; an empty object otherwise.  Slots: see m4q_session_qp_stats (M4Q_PHASE_TRACE=1 prints them).
struct PhaseClock {
#if defined(M4Q_DEV_PHASE_CLOCK)
  unsigned long long acc[16] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
  unsigned long long t = __builtin_amdgcn_s_memrealtime();
  __device__ __forceinline__ void mark(int i) {
    const unsigned long long n = __builtin_amdgcn_s_memrealtime();
    acc[i] += n - t;
    t = n;
  }
  __device__ __forceinline__ void count(int i) { ++acc[i]; }
#else
  __device__ __forceinline__ void mark(int) {}
  __device__ __forceinline__ void count(int) {}
#endif
};

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
#ifndef M4Q_NOP
#define M4Q_NOP "s_nop 1\n\t"
#endif

#ifndef M4Q_BCAST_SHFL
template <int K>
__device__ __forceinline__ double bcast(double x) {
  double r;
  asm(M4Q_NOP "v_mov_b64_dpp %0, %1 row_newbcast:%2" M4Q_DPP : "=v"(r) : "v"(x), "n"(K));
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

// ---- term emitter: a compile-time list of multiply-accumulate terms
//        acc[IDX::acc(q)] += opA(lane_{IDX::lane(q)}(a[IDX::a(q)])) * opB(b[IDX::b(q)]),   q in [Q0, QN)
// is cut into asm statements of up to 16 (double) / 4 (cplx) terms.  DOT: every term of the list adds into
// acc[IDX::acc(Q0)] (dependent FMA chains are free on gfx950, so no partial sums); otherwise a statement
// never holds two terms with the same accumulator (IDX::distinct = guaranteed run of distinct accumulators).
template <class S> struct ChunkMax { static constexpr int value = 4; };
template <> struct ChunkMax<double> { static constexpr int value = 16; };

template <class IDX, bool CA, bool CB, bool DOT, int Q0, int NA, int NB, int NC, size_t... I>
__device__ __forceinline__ void emit_block(double (&acc)[NA], const double (&a)[NB], const double (&b)[NC], std::index_sequence<I...>) {
  if constexpr (DOT) fdotN<IDX::lane(Q0 + (int)I)...>(acc[IDX::acc(Q0)], a[IDX::a(Q0 + (int)I)]..., b[IDX::b(Q0 + (int)I)]...);
  else fmacN<IDX::lane(Q0 + (int)I)...>(acc[IDX::acc(Q0 + (int)I)]..., a[IDX::a(Q0 + (int)I)]..., b[IDX::b(Q0 + (int)I)]...);
}
template <class IDX, bool CA, bool CB, bool DOT, int Q0, int NA, int NB, int NC, size_t... I>
__device__ __forceinline__ void emit_block(cplx (&acc)[NA], const cplx (&a)[NB], const cplx (&b)[NC], std::index_sequence<I...>) {
  if constexpr (DOT) cdotN<CA, CB, IDX::lane(Q0 + (int)I)...>(acc[IDX::acc(Q0)], a[IDX::a(Q0 + (int)I)]..., b[IDX::b(Q0 + (int)I)]...);
  else cmacN<CA, CB, IDX::lane(Q0 + (int)I)...>(acc[IDX::acc(Q0 + (int)I)]..., a[IDX::a(Q0 + (int)I)]..., b[IDX::b(Q0 + (int)I)]...);
}
// two runs of ND distinct accumulators in one statement (real only): term i adds into the accumulator of term i % ND
template <class IDX, int Q0, int ND, int NA, int NB, int NC, size_t... I, size_t... J>
__device__ __forceinline__ void emit_block_wrap(double (&acc)[NA], const double (&a)[NB], const double (&b)[NC], std::index_sequence<I...>,
                                                std::index_sequence<J...>) {
  fmac2xN<IDX::lane(Q0 + (int)J)...>(acc[IDX::acc(Q0 + (int)I)]..., a[IDX::a(Q0 + (int)J)]..., b[IDX::b(Q0 + (int)J)]...);
}
template <class IDX, int Q0, int ND>
constexpr bool wrap_ok() {       // the second run must hit the same accumulators in the same order
  for (int i = 0; i < ND; ++i)
    if (IDX::acc(Q0 + i) != IDX::acc(Q0 + ND + i)) return false;
  return true;
}

template <class IDX, bool CA, bool CB, bool DOT, int Q0, int QN, class S, int NA, int NB, int NC>
__device__ __forceinline__ void emit_terms(S (&acc)[NA], const S (&a)[NB], const S (&b)[NC]) {
  constexpr int left = QN - Q0;
  if constexpr (!DOT && sizeof(S) == sizeof(double) && IDX::distinct >= 2 && IDX::distinct <= 16 && left >= 2 * IDX::distinct) {
    if constexpr (wrap_ok<IDX, Q0, IDX::distinct>()) {
      constexpr int ND = IDX::distinct;
      emit_block_wrap<IDX, Q0, ND>(acc, a, b, std::make_index_sequence<ND>{}, std::make_index_sequence<2 * ND>{});
      emit_terms<IDX, CA, CB, DOT, Q0 + 2 * ND, QN>(acc, a, b);
      return;
    }
  }
  if constexpr (left > 0) {
    constexpr int cap0 = ChunkMax<S>::value;
    constexpr int cap = DOT ? cap0 : (cap0 < IDX::distinct ? cap0 : IDX::distinct);
    constexpr int CH = left < cap ? left : cap;
    emit_block<IDX, CA, CB, DOT, Q0>(acc, a, b, std::make_index_sequence<CH>{});
    emit_terms<IDX, CA, CB, DOT, Q0 + CH, QN>(acc, a, b);
  }
}

struct IdxLaneIndexScalar {   // acc[q] += lane_q(a[0]) * b[0]
  static constexpr int distinct = 1 << 20;
  static constexpr int acc(int q) { return q; }
  static constexpr int a(int) { return 0; }
  static constexpr int b(int) { return 0; }
  static constexpr int lane(int q) { return q; }
};
struct IdxDotLane {           // acc[0] += lane_q(a[0]) * b[q]
  static constexpr int distinct = 1;
  static constexpr int acc(int) { return 0; }
  static constexpr int a(int) { return 0; }
  static constexpr int b(int q) { return q; }
  static constexpr int lane(int q) { return q; }
};
struct IdxRowsum {            // acc[0] += lane_q(a[0]) * b[0]
  static constexpr int distinct = 1;
  static constexpr int acc(int) { return 0; }
  static constexpr int a(int) { return 0; }
  static constexpr int b(int) { return 0; }
  static constexpr int lane(int q) { return q; }
};
template <int K>
struct IdxSameLane {          // acc[q] += lane_K(a[q]) * b[0]
  static constexpr int distinct = 1 << 20;
  static constexpr int acc(int q) { return q; }
  static constexpr int a(int q) { return q; }
  static constexpr int b(int) { return 0; }
  static constexpr int lane(int) { return K; }
};
template <int K>
struct IdxSameLaneVec {       // acc[q] += lane_K(a[q]) * b[q]
  static constexpr int distinct = 1 << 20;
  static constexpr int acc(int q) { return q; }
  static constexpr int a(int q) { return q; }
  static constexpr int b(int q) { return q; }
  static constexpr int lane(int) { return K; }
};
template <int N>
struct IdxMatmul {            // C[i] += lane_k(M[i]) * B[k],   q = k*N + i
  static constexpr int distinct = N;
  static constexpr int acc(int q) { return q % N; }
  static constexpr int a(int q) { return q % N; }
  static constexpr int b(int q) { return q / N; }
  static constexpr int lane(int q) { return q / N; }
};
template <int N>
struct IdxMatmulHN {          // C[i] += conj(lane_i(M[k])) * B[k],   q = k*N + i
  static constexpr int distinct = N;
  static constexpr int acc(int q) { return q % N; }
  static constexpr int a(int q) { return q / N; }
  static constexpr int b(int q) { return q / N; }
  static constexpr int lane(int q) { return q % N; }
};

// single-term conveniences
template <int K, class S>
__device__ __forceinline__ void cmac_bc(S& acc, S a, S b) {       // acc += lane_K(a) * b
  S c1[1] = {acc};
  const S a1[1] = {a}, b1[1] = {b};
  emit_terms<IdxSameLane<K>, false, false, false, 0, 1>(c1, a1, b1);
  acc = c1[0];
}
template <int K>
__device__ __forceinline__ void fmac_bc(double& acc, double a, double b) { fmacN<K>(acc, a, b); }

// acc[i] += opA(lane_K(a[i])) * opB(b)   for i in [0, N)
template <bool CA, bool CB, int K, int N, class S>
__device__ __forceinline__ void mac_same_lane(S (&acc)[N], const S (&a)[N], S b) {
  const S b1[1] = {b};
  emit_terms<IdxSameLane<K>, CA, CB, false, 0, N>(acc, a, b1);
}
// acc[i] += opA(lane_i(a)) * opB(b)   for i in [0, N): the broadcast lane is the row index
template <bool CA, bool CB, int N, class S>
__device__ __forceinline__ void mac_lane_index(S (&acc)[N], S a, S b) {
  const S a1[1] = {a}, b1[1] = {b};
  emit_terms<IdxLaneIndexScalar, CA, CB, false, 0, N>(acc, a1, b1);
}
// init + sum_i opA(lane_i(v)) * opB(m[i])
template <bool CA, bool CB, int N, class S>
__device__ __forceinline__ S dot_lane_index(S v, const S (&m)[N], S init) {
  S c1[1] = {init};
  const S a1[1] = {v};
  emit_terms<IdxDotLane, CA, CB, true, 0, N>(c1, a1, m);
  return c1[0];
}
template <bool CA, bool CB, int N, class S>
__device__ __forceinline__ S dot_lane_index(S v, const S (&m)[N]) {
  return dot_lane_index<CA, CB, N>(v, m, zero_of<S>());
}

// Sum over lanes 0..N-1 of the row; result replicated in every lane of the row.
template <int N>
__device__ __forceinline__ double rowsum(double v) {
  double c1[1] = {0.0};
  const double a1[1] = {v}, b1[1] = {1.0};
  emit_terms<IdxRowsum, false, false, true, 0, N>(c1, a1, b1);
  return c1[0];
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
  emit_terms<IdxMatmul<N>, false, false, false, 0, N * N>(C, M, Bc);
}

// C[:, j] += M^H * B[:, j]   with M, B column-owned:  C[i] += conj(lane_i(M[k])) * B[k]
template <int N, class S>
__device__ __forceinline__ void matmul_cols_hn_acc(S (&C)[N], const S (&M)[N], const S (&Bc)[N]) {
  emit_terms<IdxMatmulHN<N>, true, false, false, 0, N * N>(C, M, Bc);
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
