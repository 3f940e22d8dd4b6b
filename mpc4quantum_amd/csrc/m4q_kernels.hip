// m4q_kernels.hip - kernels for ONE problem shape, selected at compile time:
//     hipcc --offload-arch=gfx950 -DM4Q_NX=9 -DM4Q_NU=2 -DM4Q_ORDER=1 -c m4q_kernels.hip
// One wavefront per workgroup; each of its four DPP rows runs one MPC instance (m4q_device.h).
// The persistent kernel strides over instance quads, so a fixed pool of per-row workspace stays
// cache resident whatever the ensemble size.
#define M4Q_KERNEL_TU 1
#include "m4q_args.h"
#include "m4q_mpc.h"
#include "m4q_tile3.h"

#ifndef M4Q_NX
#error "compile with -DM4Q_NX -DM4Q_NU -DM4Q_ORDER"
#endif
// Two objects per shape (build.py): the core object (libm4q_hip.so) holds every kernel except the closed-loop kernels with the
// GENERATOR plant (x+ = expm(dt (L0 + sum u_k L_k)) x on n x n operators: QExperiment with collapse operators, LExperiment - outside
// SURVEY section 8, and the largest objects of the library: up to 1 KB of scratch per lane); those are compiled with
// -DM4Q_VARIANT_GEN into libm4q_hip_gen.so, which m4q_capi.hip loads on the first session that asks for that plant.
#if defined(M4Q_PLANT_ONLY) || defined(M4Q_VARIANT_GEN)
#define M4Q_NO_AUX 1
#endif

#define M4Q_CAT_(a, b, c, d) a##b##_##c##_##d
#define M4Q_CAT(a, b, c, d) M4Q_CAT_(a, b, c, d)

namespace m4q {
// every shape lives in its own namespace: the shapes are linked into one library
namespace M4Q_CAT(shape_, M4Q_NX, M4Q_NU, M4Q_ORDER) {

constexpr int NX = M4Q_NX;
constexpr int NU = M4Q_NU;
constexpr int ORDER = M4Q_ORDER;
// n = d*d for vectorised density matrices.  A model space that is not a square (n = 8: two reduced qubit states,
// experiment.py:238-306) has no device plant and no Hermitian-basis path: only the complex QP machinery is built.
constexpr int DD = (NX == 4) ? 2 : (NX == 9) ? 3 : (NX == 16) ? 4 : 1;
constexpr bool SQUARE = DD * DD == NX;
constexpr int NP = PowTab<NU, ORDER>::NP;
constexpr int PITCH = ModelPitch<NX>::value;
constexpr int MODEL_ELEMS = (1 + NP) * NX * PITCH;        // per instance, elements (of S) in LDS
// the traceless path (m4q_mpc.h) runs the recursion on NX - 1 coordinates; everything it stages is no larger than the above
constexpr int SCRATCH_ELEMS = (SQUARE ? DD * DD : 0) + 2 * NX;   // plant / basis-change scratch per instance (complex)
constexpr int ROWS = 4;                                    // instances per wavefront
// Timing-only ablations (development builds, tools/build_variant.sh ... -DM4Q_EXP=<bits>; RESULTS WRONG, never shipped - m4q_device.h
// refuses the macro without -DM4Q_DEV).  What they bound is in profiles/r05_ab_experiments.txt.
//   1  every line-search step stops after exactly 3 SQP iterations and no member ever fails: the work of a launch no longer depends
//      on the numbers, so the variants below can be compared with a baseline built with bit 1 alone
//   2  all workgroups of an XCD share ONE per-row workspace (guesses, solution, gains): every workspace access hits the L2 -
//      the launch time that is left is what the workspace's L2 misses cost
//   4  the four members of a workgroup share ONE model slot in LDS: the LDS footprint of a kernel whose members' models are
//      (1 + m) shared generators - what d = 4 would need to run two wavefronts per SIMD (with -DM4Q_WAVES_REAL=2 -DM4Q_N15_HOIST=0)
//  16  (m4q_tile3.h) the lower off-diagonal tiles of the symmetric P are copied, not computed: bound of a symmetric-P sweep
#ifndef M4Q_EXP
#define M4Q_EXP 0
#endif
constexpr int MODEL_ROWS = (M4Q_EXP & 4) ? 1 : ROWS;       // model slots per workgroup in LDS
// the backward sweep on matrix-core tiles (m4q_tile3.h) is built where it is the faster form: d = 2, 3 (3 and 8 traceless coordinates)
// with an order-1 library.  At d = 4 (15 coordinates = 4 x 4 tiles) it does not fit the register file (543 spilled VGPRs, 505 against
// 71 ms on config 4) and full DPP rows leave nothing to gain; the host asks m4q_shape_*()->has_tile.
constexpr bool HAS_TILE = SQUARE && ORDER == 1 && NX - 1 <= 8;
// the shared-generator form of the clipped traceless kernel (FusedProv<..., SG>, m4q_mpc.h: one set of N_k = dt L_k per workgroup, per
// member only A_i = I + s_i0 dt L_0 and its scales) is built where the per-member models are what keeps the kernel at one wavefront
// per SIMD: d = 4 (28.8 KB of models per workgroup against 12.6).  Sessions whose models came from m4q_session_build_models with
// shared generators run it (path 4); the host asks m4q_shape_*()->has_sg.
constexpr bool HAS_SG = SQUARE && ORDER == 1 && NX == 16;

// register budget: waves per SIMD the kernels are compiled for (512 / budget VGPRs per lane).  d = 2 was compiled for four until the
// end of round 3: at 128 registers every d = 2 closed-loop kernel spilled (59-372 VGPRs), and its launches are chains of
// dependent passes that a third and fourth wavefront do not shorten: config 2 5.3 -> 4.2 ms at two (131,072 members: 23.6 -> 19.5 ms).
#ifndef M4Q_WAVES
#define M4Q_WAVES ((M4Q_NX <= 9) ? 2 : 1)
#endif
#ifndef M4Q_WAVES_REAL
#define M4Q_WAVES_REAL ((M4Q_NX <= 9) ? 2 : 1)
#endif
#define M4Q_OCC __attribute__((amdgpu_waves_per_eu(M4Q_WAVES, 8)))
template <class S> struct WavesFor { static constexpr int value = M4Q_WAVES; };
template <> struct WavesFor<double> { static constexpr int value = M4Q_WAVES_REAL; };

extern __shared__ __align__(16) unsigned char m4q_lds_raw[];

// copy one instance's model (DMDc.A layout, n x n(1+P) row-major) into its LDS block [1+P][n][PITCH]
template <int N, class S>
__device__ __forceinline__ void stage_model(S* dst, const M4Q_GLOBAL S* src, int jj) {
  constexpr int W = N * (1 + NP);
  // (not unrolled: unrolling this loop costs registers and time, profiles/r02_ab_experiments.txt)
#pragma unroll 1
  for (int e = jj; e < N * W; e += 16) {
    const int i = e / W;
    const int pk = e - i * W;
    const int p = pk / N;
    const int k = pk - p * N;
    dst[ModelPitch<N>::at(p, i, k)] = gld(src, e);
  }
}

// geometry of one lane inside its quad
struct LaneGeo {
  int g, jj, j;
  bool lane_ok;
  __device__ __forceinline__ LaneGeo() {
    const int lane = threadIdx.x;
    g = lane >> 4;
    jj = lane & 15;
    j = jj < NX ? jj : NX - 1;
    lane_ok = jj < NX;
  }
};

// ---------------------------------------------------------------------------------------------
// The fused closed loop (replaces mpc.py:161-292).
//
// Work items: an instance's run is cut at MPC step 2.  Steps 0-1 iterate the SQP to convergence (14..100
// solves, data dependent); steps >= 2 are one solve each (mpc.py:208-212).  The queue hands out all the
// "head" items (steps [step_begin, 2)) first, then the uniform "tail" items (steps [2, step_end)), so the
// long, uneven pieces are packed first and the launch drains on short uniform ones.  A tail item may be
// drawn by another workgroup - possibly on another XCD - than the one that ran its head: the head
// publishes the instance state with an agent-scope release and a flag, the tail polls the flag (without
// blocking its wavefront) and acquires.
//
// Scheduling: the wavefront is a four-slot machine.  Every DPP row pulls its OWN next instance from a
// device-wide atomic queue and carries its own (instance, MPC step, SQP iteration); one pass of the main
// loop performs one QP solve for each row at whatever point of its run that row has reached.  SQP
// iteration counts at steps 0-1 range from 14 to 100 (config 3): with rows in lockstep a quad runs at
// the pace of its slowest member and a static split leaves the slowest wave 1.6x the mean.
//
// S = cplx: any model.  S = double: Hermiticity-preserving models in the Hermitian operator basis
// (a quarter of the FMAs, half the LDS and workspace traffic); the host decides (m4q_capi.hip).
// ---------------------------------------------------------------------------------------------
constexpr int COST_ELEMS = 2 * NX * NX + NU * NU;         // Q, Qf, R staged in LDS once per workgroup
constexpr int WLS_DOUBLES = 4 * NX + 2 * NU;               // diagonal line-search weights, staged after them
constexpr int STASH_INTS = 12;                             // per-row words of RowStash
// The complex path parks x_meas in the plant / basis-change scratch (idle during a solve: 108 complex slots at d = 3, 64 needed)
// instead of a block of its own: with one the kernel needed 21,480 B of LDS, a seventh of a CU's 160 KB was 1,000 B too small for an
// eighth workgroup, and the launch ran on 1,792 wavefronts instead of 2,048 (round 3's +3.5 % on that path until this was seen in
// SQ_WAVE_CYCLES / GRBM_GUI_ACTIVE: 54 against 63).
template <class S, bool SG = false> constexpr bool stash_x_in_scratch() { return (SG || sizeof(S) == sizeof(cplx)) && ROWS * SCRATCH_ELEMS >= 64; }
template <class S, bool SG = false> constexpr int stash_bytes() {
  return 8 /* watchdog deadline */ + ROWS * STASH_INTS * 4 + (stash_x_in_scratch<S, SG>() ? 0 : 64 * 16) /* x_meas, one S per lane */;
}

// sizes of the staged model and costs for a recursion on N coordinates
template <int N> constexpr int model_elems() { return (1 + NP) * N * ModelPitch<N>::value; }
template <int N> constexpr int cost_elems() { return 2 * N * N + NU * NU; }

// Transposed copies of Q and Qf behind the costs (CostRef<S, TR>, m4q_mpc.h): the exact mode on a recursion whose rows are 64 bytes
// long (n = 8 doubles), where the row reads of (Q e)_j conflict
template <class S, int N, bool EXACT> constexpr bool cost_transposed() { return EXACT && sizeof(S) == sizeof(double) && N == 8; }
// the exact mode's pinned sweep runs on matrix-core tiles where the clipped mode's backward sweep does (traceless real path of a
// shape with HAS_TILE): config 3 exact 208 -> 190 ms (profiles/r04_ab_experiments.txt)
template <class S, bool TL, bool EXACT> constexpr bool exact_tile() { return EXACT && TL && HAS_TILE && sizeof(S) == sizeof(double); }
// SG: ROWS blocks A_i and ONE set of NP blocks N_k instead of ROWS x (1 + NP) blocks
template <int N, bool SG> constexpr int model_lds_elems() {
  return SG ? (ROWS + NP) * N * ModelPitch<N>::value : MODEL_ROWS * model_elems<N>();
}
template <class S, bool TL = false, bool TILE = false, bool EXACT = false, bool SG = false>
constexpr size_t mpc_lds_layout_bytes() {
  constexpr int N = TL ? NX - 1 : NX;
  return sizeof(S) * (size_t)(model_lds_elems<N, SG>() + cost_elems<N>() + (cost_transposed<S, N, EXACT>() ? 2 * N * N : 0)) +
         sizeof(cplx) * (size_t)(ROWS * SCRATCH_ELEMS) + sizeof(double) * (size_t)WLS_DOUBLES + (size_t)stash_bytes<S, SG>() +
         (TILE ? (size_t)TILE_LDS_BYTES : 0) + (exact_tile<S, TL, EXACT>() ? (size_t)(TILE_LDS_BYTES + TILE_PIN_LDS_BYTES) : 0);
}

__device__ __forceinline__ int row_bcast_int(int v) { return __shfl(v, 0, 16); }
// the same wave-uniform number, opaque to the compiler: what is derived from it is computed (and dies) where it is used
__device__ __forceinline__ int fresh(int v) { asm volatile("" : "+s"(v)); return v; }

// Kernel arguments are read out of the kernarg segment WHERE THEY ARE USED.  Taken as a by-value parameter the 46 fields of
// MpcArgs are loaded at kernel entry, stay live for the whole persistent loop and - the kernel has 106 scalar registers -
// end up in VGPR lanes (v_writelane / v_readlane: 161 "spilled SGPRs" in the round-2 build of the headline kernel, 445
// v_readlane per pass of the main loop).  kargs() returns the segment pointer through an opaque asm, once per phase of the
// loop: the compiler can neither hoist the loads of one phase to the kernel entry nor keep their results alive into the next.
typedef const __attribute__((address_space(4))) MpcArgs KArgs;
__device__ __forceinline__ KArgs* kargs() {
  KArgs* p = (KArgs*)__builtin_amdgcn_kernarg_segment_ptr();      // explicit arguments start at offset 0
  asm volatile("" : "+s"(p));
  return p;
}

// Per-row state that only the bookkeeping phases of the loop need, parked in LDS while the two sweeps run (they take the
// whole register file; left to the compiler these values went to scratch: 178 spilled VGPRs in the round-2 headline kernel).
// Row-uniform words are written by the row's lane 0 and read back as an LDS broadcast; volatile, so that the values are
// really re-read after the sweeps instead of being carried in registers across them.
#define M4Q_LDS __attribute__((address_space(3)))
template <class S>
struct RowStash {
  volatile M4Q_LDS int* w;          // [ROWS][STASH_INTS]
  volatile M4Q_LDS double* xm;      // [64][2]
  __device__ __forceinline__ void put_int(int g, int jj, int slot, int v) const { if (jj == 0) w[g * STASH_INTS + slot] = v; }
  __device__ __forceinline__ int get_int(int g, int slot) const { return w[g * STASH_INTS + slot]; }
  __device__ __forceinline__ void put_x(double v) const { xm[2 * threadIdx.x] = v; }
  __device__ __forceinline__ void put_x(cplx v) const { xm[2 * threadIdx.x] = v.re; xm[2 * threadIdx.x + 1] = v.im; }
  __device__ __forceinline__ void get_x(double& v) const { v = xm[2 * threadIdx.x]; }
  __device__ __forceinline__ void get_x(cplx& v) const { v.re = xm[2 * threadIdx.x]; v.im = xm[2 * threadIdx.x + 1]; }
};

// the exact mode's pinned sweep on tiles: what a row hands to the tile layout (member = (lane >> 2) & 3 there, lane >> 4 in rows)
template <int NS>
struct ExactTile {
  static constexpr bool enabled = true;
  const double* lds_models;
  volatile M4Q_LDS double* tgb; volatile M4Q_LDS int* tiw; volatile M4Q_LDS double* tpin;
  GView Xg, Ug, gains;
  const double* Q; const double* Qf; const double* R;
  unsigned sX, sU, sG;
  int g, jj;
  __device__ __forceinline__ void sweep(int T, const Window& win, const PinCtx<NU>& pin, bool going) const {
    if (jj == 0) {
      tiw[g * TILE_IO_WORDS + 0] = going ? 1 : 0;
      tiw[g * TILE_IO_WORDS + 1] = (int)win.xbm.off;
      tiw[g * TILE_IO_WORDS + 2] = (int)win.ubm.off;
      tiw[g * TILE_IO_WORDS + 3] = (int)pin.stat.off;
#pragma unroll
      for (int k = 0; k < NU; ++k) { tpin[g * TILE_PIN_DOUBLES + k] = pin.lo0[k]; tpin[g * TILE_PIN_DOUBLES + 3 + k] = pin.hi0[k]; }
    }
    wave_sync();
    TileBackwardB<NS, NU, ORDER, true> ts;
    const int mb = ts.L.mb;
    const int dm = mb - g;
    ts.mdl = lds_models + (MODEL_ROWS == ROWS ? mb : 0) * model_elems<NS>();
    ts.T = T;
    ts.Xg = Xg; ts.Xg.off = Xg.off + (unsigned)(dm * (int)(sX * sizeof(double)));
    ts.Ug = Ug; ts.Ug.off = Ug.off + (unsigned)(dm * (int)(sU * sizeof(double)));
    ts.gains = gains; ts.gains.off = gains.off + (unsigned)(dm * (int)(sG * sizeof(double)));
    ts.xbm = win.xbm; ts.xbm.off = (unsigned)tiw[mb * TILE_IO_WORDS + 1];
    ts.ubm = win.ubm; ts.ubm.off = (unsigned)tiw[mb * TILE_IO_WORDS + 2];
    ts.stat = pin.stat; ts.stat.off = (unsigned)tiw[mb * TILE_IO_WORDS + 3];
    ts.sat = pin.box.sat;
    ts.Q = Q; ts.Qf = Qf; ts.R = R;
    ts.gb = tgb + mb * TILE_GB_DOUBLES;
    const bool run_t = tiw[mb * TILE_IO_WORDS + 0] != 0;
    ts.backward(run_t);
    wave_sync();
  }
};

// TL: the state lives in the NX - 1 traceless coordinates (S = double only; m4q_mpc.h).  NS = dimension of the recursion;
// the I/O side (xs, the SQP-guess checkpoint, the plant) stays NX complex numbers per node.
// TILE: the two sweeps of the clipped solve run on fp64 matrix-core tiles (m4q_tile.h) instead of DPP rows.
// smallest recursion dimension that gets the constant-target instantiation of the sweeps (round 2: 15 - at d = 3 the second
// instantiation cost more in register allocation than it saved; with round 3's lower pressure it pays: config 3 41.3 -> 40.3 ms,
// config 5's share 138.1 -> 134.7; d = 2 indifferent)
constexpr int TC_MIN_N = 8;
// EXACT: further cuts of an instance's run after step 2 (strictly increasing, > 2; see the kernel)
// (round 3, when every cut cost two passes of 41 basis changes: one cut at 5.  Round 4, with the guess handed over as it is:
//  {4, 7, 12} - config 3 185-187 ms either way, config 4 1,747 -> 1,657, config 5's share 4,315 -> 4,155; {5, 10}: 1,704 at
//  config 4; a cut at every step 3..9: 189 ms at config 3; profiles/r04_ab_experiments.txt)
#ifndef M4Q_EXACT_CUTS
#define M4Q_EXACT_CUTS 4, 7, 12
#endif
constexpr int XCUTS[] = {M4Q_EXACT_CUTS};
constexpr int NXC = (int)(sizeof(XCUTS) / sizeof(int));
#ifndef M4Q_WAVES_EXACT
#define M4Q_WAVES_EXACT(S) WavesFor<S>::value
#endif
#ifndef M4Q_PUBLISH_NODES
#define M4Q_PUBLISH_NODES 1
#endif
#ifndef M4Q_LS_NODES
#define M4Q_LS_NODES 1
#endif
#ifndef M4Q_UPD16
#define M4Q_UPD16 1
#endif
#ifndef M4Q_PIECE_RAW
#define M4Q_PIECE_RAW 1
#endif
#ifndef M4Q_WAVES_SG
#define M4Q_WAVES_SG 2
#endif
#ifndef M4Q_WAVES_TILE
#define M4Q_WAVES_TILE 2
#endif
// Development builds (-DM4Q_DEV_PHASE_CLOCK): PhaseClock (m4q_device.h) sums the 100 MHz clock over the phases of the main loop; every
// wavefront adds its sums to queue[8..23] (u64) on exit; M4Q_PHASE_TRACE=1 makes m4q_session_qp_stats print them.
#if defined(M4Q_DEV_PHASE_CLOCK)
#define M4Q_PHASE_FLUSH() if (threadIdx.x == 0) { for (int i = 0; i < 16; ++i) __hip_atomic_fetch_add((M4Q_GLOBAL unsigned long long*)kargs()->queue + 8 + i, pc.acc[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
#else
#define M4Q_PHASE_FLUSH()
#endif
#define M4Q_PHASE_DECL PhaseClock pc;
#define M4Q_PHASE_MARK(i) pc.mark(i);
template <class S, int PLANT, bool EXACT, bool TL = false, bool TILE = false, bool SG = false>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(SG ? M4Q_WAVES_SG : TILE ? M4Q_WAVES_TILE : EXACT ? M4Q_WAVES_EXACT(S) : WavesFor<S>::value, 8))) void mpc_kernel(MpcArgs) {
  static_assert(!TL || (sizeof(S) == sizeof(double) && SQUARE), "the traceless path is a real path of a d x d density matrix");
  static_assert(!TILE || (TL && !EXACT && ORDER == 1), "tile sweep: clipped solve on the traceless real coordinates, order-1 libraries");
  static_assert(!SG || (TL && !TILE && !EXACT && ORDER == 1), "shared generators: the clipped traceless kernel of an order-1 library");
  constexpr int NS = TL ? NX - 1 : NX;
  using Prov = FusedProv<S, NS, NU, ORDER, SG>;
  // LDS: [4 x scratch (complex)] [4 x model (S)] [Q Qf R (S)] [line-search weights] [watchdog deadline, row stash]
  cplx* scratch = reinterpret_cast<cplx*>(m4q_lds_raw);
  S* lds = reinterpret_cast<S*>(scratch + ROWS * SCRATCH_ELEMS);
  const LaneGeo L;
  const int g = L.g, jj = L.jj;
  const int j = jj < NS ? jj : NS - 1;           // this lane's state coordinate (lanes that own none shadow the last)
  const int jio = L.j;                           // ... and its slot of vec(rho) on the I/O side
  const bool lane_ok = jj < NS, lane_io = L.lane_ok;
  double tau = 0.0;                              // TL: the member's trace coordinate tr(rho)/sqrt(d) (row-uniform)
  constexpr bool QTR = cost_transposed<S, NS, EXACT>();
  // (SG: MODEL_K is one block - the member's A_i -, the NP shared blocks N_k follow the ROWS of them)
  constexpr int MODEL_K = SG ? NS * ModelPitch<NS>::value : model_elems<NS>(), COST_K = cost_elems<NS>() + (QTR ? 2 * NS * NS : 0);
  S* mdl = lds + (MODEL_ROWS == ROWS || SG ? g : 0) * MODEL_K;
  S* mdn = lds + ROWS * MODEL_K;                   // (SG only)
  scratch += g * SCRATCH_ELEMS;
  S* ldsQ = lds + model_lds_elems<NS, SG>();
  double* ldsW = reinterpret_cast<double*>(ldsQ + COST_K);
  volatile M4Q_LDS unsigned long long* wd_slot = (volatile M4Q_LDS unsigned long long*)(ldsW + WLS_DOUBLES);
  RowStash<S> stash;
  stash.w = (volatile M4Q_LDS int*)(ldsW + WLS_DOUBLES + 1);
  stash.xm = stash_x_in_scratch<S, SG>() ? (volatile M4Q_LDS double*)m4q_lds_raw : (volatile M4Q_LDS double*)(stash.w + ROWS * STASH_INTS);
  // TILE: hand-over block between the DPP-row state machine and the tile sweeps, and the G / h broadcast tiles
  volatile M4Q_LDS double* tgb = stash.xm + 64 * 2;
  volatile M4Q_LDS int* tiw = (volatile M4Q_LDS int*)(tgb + ROWS * TILE_GB_DOUBLES);
  volatile M4Q_LDS double* tpin = (volatile M4Q_LDS double*)(tiw + ROWS * TILE_IO_WORDS);      // (exact_tile kernels only)
  int T0, flags;
  bool ls_diag, two_phase;
  int n_pieces;                // work items per instance: head [step_begin, 2), then [2, XCUTS[0]), ... , [.., step_end)
  auto piece_cut = [](int i) __attribute__((always_inline)) {       // where piece i (>= 1) begins
    int v = 2;
#pragma unroll
    for (int k = 0; k < NXC; ++k)
      if (i == k + 2) v = XCUTS[k];
    return v;
  };
  auto piece_of = [&](int begin, int step_begin) __attribute__((always_inline)) {   // which piece begins there
    int ph = begin == step_begin ? 0 : 1;
#pragma unroll
    for (int k = 0; k < NXC; ++k)
      if (begin == XCUTS[k] && begin != step_begin) ph = k + 2;
    return ph;
  };
  {
    KArgs* a = kargs();
    T0 = a->T;
    flags = a->flags;
    const M4Q_GLOBAL S* gQ = (const M4Q_GLOBAL S*)a->Q;
    const M4Q_GLOBAL S* gQf = (const M4Q_GLOBAL S*)a->Qf;
    const M4Q_GLOBAL S* gR = (const M4Q_GLOBAL S*)a->R;
    for (int e = threadIdx.x; e < 2 * NS * NS + NU * NU; e += 64)
      ldsQ[e] = e < NS * NS ? gld(gQ, e) : (e < 2 * NS * NS ? gld(gQf, e - NS * NS) : gld(gR, e - 2 * NS * NS));
    if constexpr (QTR) {
      for (int e = threadIdx.x; e < 2 * NS * NS; e += 64) {
        const int blk = e / (NS * NS), r = (e / NS) % NS, c = e % NS;
        ldsQ[cost_elems<NS>() + e] = gld(blk ? gQf : gQ, c * NS + r);
      }
    }
    if constexpr (SG) {
      // the workgroup's copy of N_k = dt L_k (a->gens: [1 + NP][NS][NS] doubles, already scaled by dt; block 0 = dt L_0 is read
      // per member when its A_i is formed)
      for (int e = threadIdx.x; e < NP * NS * NS; e += 64) {
        const int p = e / (NS * NS), r = (e / NS) % NS, c = e % NS;
        mdn[ModelPitch<NS>::at(p, r, c)] = gld((const M4Q_GLOBAL S*)a->gens, NS * NS + e);
      }
    }
    ls_diag = a->Wls != nullptr;
    if (ls_diag) {
      for (int e = threadIdx.x; e < WLS_DOUBLES; e += 64) ldsW[e] = gld(a->Wls, e);
    }
    // cut the run in two work items per instance when the launch covers both regimes
    two_phase = a->step_begin < 2 && a->step_end > 2;
    // EXACT: a third piece.  The hard solves of the exact mode are the first warm steps (their shifted guess is poor: 30-130
    // active-set iterations against 1-2), so a tail [2, step_end) is as uneven as a head; cut again (XCUTS) the launch drains on the
    // late steps' uniform one-sweep solves instead (round 3, config 3: no cut 255 ms, at 8 245-247, at 6 236-239, at 5 233, at 4 244).
    n_pieces = two_phase ? 2 : 1;
    if constexpr (EXACT) {
#pragma unroll
      for (int i = 0; i < NXC; ++i)
        if (two_phase && a->step_end > XCUTS[i]) ++n_pieces;
    }
    // Watchdog.  The loop below ends when the queue is empty and every row has finished, and a tail item waits for a flag another
    // workgroup sets: exits that depend on data.  A persistent kernel whose wavefronts never finish takes the GPU (and on this pool
    // the host's other GPUs) down with it, so every wavefront also leaves once the constant 100 MHz clock has advanced
    // deadline_ticks since it started; the host finds queue[1] set and fails the launch loudly (M4Q_E_TIMEOUT).  Every loop of
    // the kernel without a data-independent trip bound passes through this check: the main loop below (one pass = at most one
    // QP solve or one active-set iteration per row; the head-flag poll is a pass of it that only sleeps), nothing else - the
    // horizon loops run T trips, the plant's squaring loop at most 60, the staging loops over fixed sizes.
    if (threadIdx.x == 0) *wd_slot = __builtin_amdgcn_s_memrealtime() + a->deadline_ticks;
  }
  CostRef<S, QTR> cost;
  cost.Q = ldsQ; cost.Qf = ldsQ + NS * NS; cost.q_stride = 0; cost.R = ldsQ + 2 * NS * NS; cost.r_stride = 0;
  if constexpr (QTR) { cost.QT = ldsQ + cost_elems<NS>(); cost.QfT = cost.QT + NS * NS; }
  // workspace of this resident row: wave-uniform base per workgroup, lane part = row within the wave
  const unsigned sX = (unsigned)(T0 + 1) * NS, sU = (unsigned)T0 * NU, sG = (unsigned)T0 * (NS + 1) * NU;
  const unsigned sXc = (unsigned)(T0 + 1) * NX;            // the SQP-guess checkpoint field: complex, NX per node
  constexpr bool PIECE_RAW = M4Q_PIECE_RAW && sizeof(S) == sizeof(double);     // (between items of one launch: see the publish step)
  // Lanes NX..15 of a row own no column (7 of 16 at d = 3, 12 of 16 at d = 2).  Left enabled they run the sweeps on a copy of
  // column NX-1's data: harmless for the results, but the fp64 pipe spends power on them, and the clock this chip holds under an
  // fp64-dense load follows the power.  EXEC is therefore off for them during the two sweeps (every DPP source is a lane < NX;
  // results bit-identical): the complex path runs at 2.30 GHz instead of 2.04 (133.3 -> 117.6 ms, config 3), the real path at 2.29
  // instead of 2.18 (51.2 -> 50.4 ms; config 5's share 171.2 -> 166.3 ms).  profiles/r02_ab_experiments.txt, r02_clock_ramp.txt.
  constexpr bool MASK_IDLE = M4Q_MASK_IDLE && NS < 16 && !EXACT && !TILE;
  GView Xg, Ug, Xo, Uo, gains, Xalt, Ualt, pin_stat;
  {
    KArgs* a = kargs();
    const long wsb = (M4Q_EXP & 2) ? (long)(blockIdx.x & 7) : (long)blockIdx.x;
    M4Q_GLOBAL S* wX = (M4Q_GLOBAL S*)a->ws_Xg;
    Xg = gview(wX, wsb * ROWS * sX, g * sX);
    Ug = gview(a->ws_Ug, wsb * ROWS * sU, g * sU);
    // (Xo, Uo) live in the same allocations right behind all the (Xg, Ug): same wave-uniform base, so a row can
    // direct its rollout output to either by its lane offset alone
    Xo = gview(wX, wsb * ROWS * sX, g * sX + gridDim.x * ROWS * sX);
    Uo = gview(a->ws_Ug, wsb * ROWS * sU, g * sU + gridDim.x * ROWS * sU);
    gains = gview((M4Q_GLOBAL S*)a->ws_gains, wsb * ROWS * sG, g * sG);
    // EXACT (M4Q_QP_EXACT_BOX): third trajectory pair, working set and Newton point of the projected-Newton solver,
    // again behind the others in the same allocations
    Xalt = gview(wX, wsb * ROWS * sX, g * sX + 2 * gridDim.x * ROWS * sX);
    Ualt = gview(a->ws_Ug, wsb * ROWS * sU, g * sU + 2 * gridDim.x * ROWS * sU);
    pin_stat = gview(a->ws_Ug, wsb * ROWS * sU, g * sU + 3 * gridDim.x * ROWS * sU);
  }
  const bool band = (flags & QP_DU_BAND) != 0;

  // per-row state (uniform inside a row)
  long b = 0;
  bool active = false, need_new = true, pending = false;
  int step = 0, iter = 0, code = 0, done_steps = 0;
  int row_begin = 0, row_end = 0;
  S x_cur = zero_of<S>();
  S x_meas = zero_of<S>();     // last MEASURED state (xs[k * measure_freq]); equals x_cur when measure_freq == 1
  double uprev[NU];
#pragma unroll
  for (int k = 0; k < NU; ++k) uprev[k] = 0.0;
  double gsc[SG ? NU : 1];     // SG: this row's member's scales of the control operators
#pragma unroll
  for (int k = 0; k < (SG ? NU : 1); ++k) gsc[k] = 1.0;
  BoxQpRow qp;                 // EXACT: the box-QP solve this row has in progress (spans iterations of the loop below)
  int qp_passes = 0;           // passes of this wavefront through the solver iteration (statistics)
  unsigned xt_off = 0, ut_off = 0, op0_off = 0, ops_off = 0;      // this row's member inside the per-member arrays (bytes)
  wave_sync();
  M4Q_PHASE_DECL

  while (true) {
    // The horizon and the row's workspace offsets go through an opaque asm once per pass: everything derived from them
    // (unroll-remainder predicates of the horizon loops, 64-bit element addresses) is then computed in the phase that uses it.
    // Hoisted to the kernel entry - they are invariants of this loop - those values lived through every phase and were what
    // the allocator spilled (predicate pairs into VGPR lanes, addresses into scratch).
    int T = T0;
    asm volatile("" : "+s"(T));
    asm volatile("" : "+v"(Xg.off), "+v"(Ug.off), "+v"(Xo.off), "+v"(Uo.off), "+v"(gains.off));
    if constexpr (EXACT) asm volatile("" : "+v"(Xalt.off), "+v"(Ualt.off), "+v"(pin_stat.off));
    Prov prov;
    prov.mdl = mdl; prov.Xg = Xg; prov.Ug = Ug; prov.j = j;
    if constexpr (SG) prov.mdn = mdn;        // (prov.sc: set after the draw below - a row may take a new member in this very pass)

    if (__builtin_amdgcn_s_memrealtime() > *wd_slot) {
      if (threadIdx.x == 0) __hip_atomic_store(kargs()->queue + 1, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      break;
    }
    M4Q_PHASE_MARK(3)
    // ---- rows without work draw the next item; tail items wait (without blocking) for their head ----
    if (__any(need_new || pending)) {
      KArgs* a = kargs();
      const int B = a->B, n_items = n_pieces * B;
      const int step_begin = a->step_begin, step_end = a->step_end, mf = a->measure_freq;
      const long sXs = (long)(a->n_steps + 1) * NX, sUs = (long)a->n_steps * NU;
      int nb = 0;
      if (need_new && jj == 0) nb = __hip_atomic_fetch_add(a->queue, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      nb = row_bcast_int(nb);
      if (need_new) {
        need_new = false;
        if (nb < n_items) {
          const int ph = nb / B;                               // 0 head, then the later pieces in order
          b = nb - ph * B;
          row_begin = ph == 0 ? step_begin : piece_cut(ph);
          row_end = ph == n_pieces - 1 ? step_end : piece_cut(ph + 1);
          pending = true;
        }
      }
      // a tail item starts once the head of its instance has been published
      // (head_done[b] counts the pieces of the instance that have been published)
      int ready = 1;
      const bool later = two_phase && row_begin != step_begin;
      if (pending && later && jj == 0)
        ready = __hip_atomic_load(a->head_done + b, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= piece_of(row_begin, step_begin) ? 1 : 0;
      ready = row_bcast_int(ready);
      const bool fresh = pending && ready != 0;
      if (__any(fresh && later)) {
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      }
      if (fresh) { pending = false; active = true; }
      wave_sync();
      if (fresh) {
        if constexpr (SG) {
          // A_i = I + s_i0 (dt L_0) from the shared generator; the member's other scales ride on the controls (FusedProv<..., SG>)
          const M4Q_GLOBAL double* sc = a->scales + b * (1 + NU);
          const double s0 = gld(sc, 0);
#pragma unroll
          for (int k = 0; k < NU; ++k) gsc[k] = gld(sc, 1 + k);

#pragma unroll 1
          for (int e = jj; e < NS * NS; e += 16) {
            const int r = e / NS, c = e - r * NS;
            mdl[ModelPitch<NS>::at(0, r, c)] = fma(s0, gld((const M4Q_GLOBAL S*)a->gens, e), r == c ? 1.0 : 0.0);
          }
        } else {
          stage_model<NS>(mdl, (const M4Q_GLOBAL S*)a->models + b * a->model_stride, jj);
        }
        xt_off = (unsigned)(b * a->xt_stride) * (unsigned)sizeof(S);
        ut_off = (unsigned)(b * a->ut_stride) * (unsigned)sizeof(double);
        op0_off = (unsigned)(b * a->op0_stride) * (unsigned)sizeof(cplx);
        ops_off = (unsigned)(b * a->ops_stride) * (unsigned)sizeof(cplx);
        step = row_begin;
        iter = 0;
        code = 0;
        done_steps = 0;
        if (row_begin <= 1) {                                    // u_prev of steps 0 and 1: U_ref[:, 0] (mpc.py:185)
          GView u0 = gview(a->u_targ, 0, 0);
          u0.off = ut_off;
#pragma unroll
          for (int k = 0; k < NU; ++k) uprev[k] = u0.ld<double>(k);
        }
      }
      if (__any(fresh && row_begin == 0)) {
        if (fresh && row_begin == 0) {
          // X_guess = tile(x0), U_guess = 0 (mpc.py:141-142); xs[0] = x0 (:160)
          const S x0 = gld((const M4Q_GLOBAL S*)a->x0s, b * NS + j);
          x_cur = x0;
          x_meas = x0;
          if (lane_ok) {
#pragma unroll 8
            for (int t = 0; t <= T; ++t) Xg.st<S>(t * NS + j, x0);
          }
          if (lane_io) gst(a->xs, b * sXs + jio, gld(a->x0c, b * NX + jio));
          for (int e = jj; e < T * NU; e += 16) Ug.st<double>(e, 0.0);
        }
        if constexpr (TL) {
          // the member's trace coordinate, from the complex x0 (every lane of the row ends up with it)
          double t0 = 0.0;
          const cplx xc = (fresh && row_begin == 0) ? gld(a->x0c, b * NX + jio) : czero();
          (void)Basis<S, TL>::template to_state<NX, DD>(xc, scratch, j, jj, t0);
          if (fresh && row_begin == 0) tau = t0;
        }
      }
      if (__any(fresh && row_begin != 0)) {
        // resume: the SQP guess, state and exit code of an earlier item or launch (fields X_GUESS/U_GUESS/XS/US/CODES).
        // From an earlier LAUNCH (or the host) the stored guess is complex in the original basis; the basis change needs every lane
        // (LDS exchange).  From an earlier item of THIS launch (real paths) it is the working guess itself, as the item left it in
        // the same field (PIECE_RAW: see the publish step) - a flat copy, no basis change and no rounding.
        const bool rs = fresh && row_begin != 0;
        const bool rs_raw = PIECE_RAW && rs && later;
        double tdum = 0.0, tnew = 0.0;
        if (__any(rs && !rs_raw)) {
          for (int t0 = 0; t0 <= T; t0 += 8) {
            cplx v[8];
#pragma unroll
            for (int q = 0; q < 8; ++q) v[q] = (rs && !rs_raw) ? gld(a->Xg, b * sXc + (t0 + q <= T ? t0 + q : T) * NX + jio) : czero();
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int q = 0; q < 8; ++q) {
              if (t0 + q <= T) {
                const S r = Basis<S, TL>::template to_state<NX, DD>(v[q], scratch, j, jj, tdum);
                if (rs && !rs_raw && lane_ok) Xg.st<S>((t0 + q) * NS + j, r);
              }
            }
          }
        }
        if constexpr (PIECE_RAW) {
          if (rs_raw) {
            const M4Q_GLOBAL double* raw = (const M4Q_GLOBAL double*)(a->Xg + b * sXc);
            const int count = (T + 1) * NS;
            if constexpr (M4Q_UPD16 && NS % 2 == 0) {
              constexpr int U = 6;
              for (int e0 = 2 * jj; e0 < count; e0 += 32 * U) {
                d2_t v[U];
#pragma unroll
                for (int u = 0; u < U; ++u) v[u] = *reinterpret_cast<const M4Q_GLOBAL d2_t*>(raw + (e0 + 32 * u < count ? e0 + 32 * u : count - 2));
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int u = 0; u < U; ++u) {
                  if (e0 + 32 * u < count) {
                    const double r[2] = {v[u].x, v[u].y};
                    stn<2>(Xg, e0 + 32 * u, r);
                  }
                }
              }
            } else {
              constexpr int U = 12;
              for (int e0 = jj; e0 < count; e0 += 16 * U) {
                double v[U];
#pragma unroll
                for (int u = 0; u < U; ++u) v[u] = gld(raw, e0 + 16 * u < count ? e0 + 16 * u : count - 1);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int u = 0; u < U; ++u)
                  if (e0 + 16 * u < count) Xg.st<double>(e0 + 16 * u, v[u]);
              }
            }
          }
        }
        const cplx xc = rs ? gld(a->xs, b * sXs + (long)row_begin * NX + jio) : czero();
        const S r = Basis<S, TL>::template to_state<NX, DD>(xc, scratch, j, jj, tdum);
        const cplx xm = rs ? gld(a->xs, b * sXs + (long)(row_begin / mf) * mf * NX + jio) : czero();
        const S rm = Basis<S, TL>::template to_state<NX, DD>(xm, scratch, j, jj, tnew);
        if (rs) {
          x_cur = r;
          x_meas = rm;
          tau = tnew;
          for (int e = jj; e < T * NU; e += 16) Ug.st<double>(e, gld(a->Ug, b * sU + e));
          code = gld(a->codes, b);
          done_steps = gld(a->steps_done, b);
#pragma unroll
          for (int k = 0; k < NU; ++k) uprev[k] = row_begin > 1 ? gld(a->us, b * sUs + (long)(row_begin - 1) * NU + k) : uprev[k];
          if (code != 0) step = row_end;          // finished earlier (exit code set by the host or the device)
        }
      }
      wave_sync();
    }
    M4Q_PHASE_MARK(0)
    if (!__any(active || pending)) {
      M4Q_PHASE_FLUSH()
      if (EXACT && threadIdx.x == 0)
        __hip_atomic_fetch_add((M4Q_GLOBAL unsigned long long*)kargs()->queue + 7, (unsigned long long)qp_passes, __ATOMIC_RELAXED,
                               __HIP_MEMORY_SCOPE_AGENT);
      break;
    }
    if (!__any(active)) {
      __builtin_amdgcn_s_sleep(8);                 // only tail items whose head is still running: poll again
      continue;
    }
    const bool running = active && step < row_end;
    if constexpr (SG) {
#pragma unroll
      for (int k = 0; k < NU; ++k) prov.sc[k] = gsc[k];
    }

    // ---- one QP solve per row ----
    // park what the sweeps do not need (the compiler would keep it in scratch across them)
    stash.put_int(g, jj, 0, (int)(b & 0xffffffff));
    stash.put_int(g, jj, 1, (int)(b >> 32));
    stash.put_int(g, jj, 2, code);
    stash.put_int(g, jj, 3, done_steps);
    stash.put_int(g, jj, 4, row_begin);
    stash.put_int(g, jj, 5, (active ? 1 : 0) | (need_new ? 2 : 0) | (pending ? 4 : 0));
    stash.put_int(g, jj, 6, (int)op0_off);
    stash.put_int(g, jj, 7, (int)ops_off);
    stash.put_x(x_meas);
    double uapp[NU];
#pragma unroll
    for (int k = 0; k < NU; ++k) uapp[k] = 0.0;
    double chk = 0.0;
    bool solved = running;       // the rows whose QP is solved in this pass and that go on to the line search / update / plant below.
                                 // Clipped mode: every running row, each pass.  EXACT: a row's solve spans several passes (one
                                 // active-set iteration per pass), so that no row waits for the slowest solve of its wavefront.
    bool capped = false;         // EXACT: the solve stopped at its iteration cap (not converged): exit code 2
    bool use_ls;
    Window win;
    {
      KArgs* a = kargs();
      // target window: X_ref = X_targ[:, :T+1] for steps 0 and 1, then X_targ[:, step-1:...] (mpc.py:145,276)
      const int w = step <= 1 ? 0 : step - 1;
      win.xbm = gview((const M4Q_GLOBAL S*)a->x_targ, 0, 0);
      win.xbm.off = xt_off + (unsigned)w * NS * (unsigned)sizeof(S);
      win.ubm = gview(a->u_targ, 0, 0);
      win.ubm.off = ut_off + (unsigned)w * NU * (unsigned)sizeof(double);
      const double sat = a->sat, du = a->du;
      double lo0[NU], hi0[NU];
#pragma unroll
      for (int k = 0; k < NU; ++k) {
        // u_prev = us[step-1] if step > 1 else U_ref[:, 0] (mpc.py:185): `uprev` holds whichever applies (set when the row took its
        // item and when a step ends) - read from the target array here it was a dependent global load in every pass of steps 0 and 1
        const double up = uprev[k];
        lo0[k] = band ? up - du : -sat;
        hi0[k] = band ? up + du : sat;
      }
      use_ls = !(a->warm_start && step > 1);        // mpc.py:208-213
      const bool st = running && lane_ok;
      if constexpr (TILE) {
        // ---- hand the row's solve to the tile layout: member = (lane >> 2) & 3 there, lane >> 4 here ----
        if (jj == 0) {
          tiw[g * TILE_IO_WORDS + 0] = running ? 1 : 0;
          tiw[g * TILE_IO_WORDS + 1] = (int)win.xbm.off;
          tiw[g * TILE_IO_WORDS + 2] = (int)win.ubm.off;
        }
        wave_sync();
        {
          TileBackwardB<NS, NU, ORDER> ts;
          const int mb = ts.L.mb;
          const int dm = mb - g;                                 // this lane's member there minus its member here
          ts.mdl = reinterpret_cast<const double*>(lds) + (MODEL_ROWS == ROWS ? mb : 0) * MODEL_K;
          ts.T = T;
          ts.Xg = Xg; ts.Xg.off = Xg.off + (unsigned)(dm * (int)(sX * sizeof(double)));
          ts.Ug = Ug; ts.Ug.off = Ug.off + (unsigned)(dm * (int)(sU * sizeof(double)));
          ts.gains = gains; ts.gains.off = gains.off + (unsigned)(dm * (int)(sG * sizeof(double)));
          const int tfl = tiw[mb * TILE_IO_WORDS + 0];
          ts.xbm = win.xbm; ts.xbm.off = (unsigned)tiw[mb * TILE_IO_WORDS + 1];
          ts.ubm = win.ubm; ts.ubm.off = (unsigned)tiw[mb * TILE_IO_WORDS + 2];
          ts.Q = reinterpret_cast<const double*>(cost.Q); ts.Qf = reinterpret_cast<const double*>(cost.Qf);
          ts.R = reinterpret_cast<const double*>(cost.R);
          ts.gb = tgb + mb * TILE_GB_DOUBLES;
          M4Q_PHASE_MARK(3)
          ts.backward((tfl & 1) != 0);
          wave_sync();
          M4Q_PHASE_MARK(1)
        }
        wave_sync();
        // the rollout on DPP rows (on tiles it was built twice: round 3's per-index form took 2.3 times as long, round 4's time-batched
        // form - tools/tile_rollout_r04.h - the same SIMD time at d = 3 and 14 % more of the launch at d = 2)
        // (idle lanes sit it out as in the DPP kernels: MASK_IDLE itself is off for TILE because the tile sweep needs all 64 lanes)
        constexpr bool MASK_FWD = M4Q_MASK_IDLE && NS < 16;
        if (!MASK_FWD || lane_ok)         // (the tile path runs with a constant target only)
          chk = rollout_forward<S, NS, NU, false, true>(prov, T, x_cur, win, cost, flags, gains, sat, lo0, hi0, Xo, Uo, j, st, uapp, !use_ls, &Xg, &Ug);
        if constexpr (MASK_FWD) {
          chk = bcast<0>(chk);
#pragma unroll
          for (int k = 0; k < NU; ++k) uapp[k] = bcast<0>(uapp[k]);
        }
      } else if constexpr (!EXACT) {
        // (xbar_t the same for every t: the sweep needs no row form of A_t - wave-uniform choice between two instantiations.
        //  Round 2: n = 16 only - config 4 85.5 -> 84.2 ms, while at n = 9 the kernel with both instantiations was SLOWER, 50.65 -> 51.7 ms,
        //  although it executes 27 vector instructions fewer per horizon index (profiles/r02_ab_experiments.txt).  Round 3, with the
        //  kernel off the register ceiling: n >= 8 - M4Q_TC_MIN_N above.)
        constexpr bool HAS_TC = NS >= TC_MIN_N && sizeof(S) == sizeof(double);
        bool tc = false;
        if constexpr (HAS_TC) tc = (flags & QP_TARG_CONST) != 0;
        if constexpr (HAS_TC) {
          if (tc && (!MASK_IDLE || lane_ok))
            riccati_backward<S, NS, NU, Prov, false, true>(prov, T, win, cost, flags, gains, j, st);
        }
        M4Q_PHASE_MARK(3)
        if (!tc && (!MASK_IDLE || lane_ok)) riccati_backward<S, NS, NU>(prov, T, win, cost, flags, gains, j, st);
        wave_sync();
        M4Q_PHASE_MARK(1)
        constexpr bool HAS_TCF = HAS_TC && M4Q_TCF(NS);
        bool tcf = false;
        if constexpr (HAS_TCF) tcf = tc;
        if constexpr (HAS_TCF) {
          if (tc && (!MASK_IDLE || lane_ok))
            chk = rollout_forward<S, NS, NU, false, true>(prov, T, x_cur, win, cost, flags, gains, sat, lo0, hi0, Xo, Uo, j, st, uapp,
                                                          !use_ls, &Xg, &Ug);
        }
        if (!tcf && (!MASK_IDLE || lane_ok))
          chk = rollout_forward<S, NS, NU, false>(prov, T, x_cur, win, cost, flags, gains, sat, lo0, hi0, Xo, Uo, j, st, uapp,
                                                  !use_ls, &Xg, &Ug);
        if constexpr (MASK_IDLE) {
          // the row's scalars back into the lanes that sat the sweep out
          chk = bcast<0>(chk);
#pragma unroll
          for (int k = 0; k < NU; ++k) uapp[k] = bcast<0>(uapp[k]);
        }
      } else {
        PinCtx<NU> pin;
        pin.stat = pin_stat;
        pin.box.sat = sat;
        pin.tconst = (flags & QP_TARG_CONST) != 0;
#pragma unroll
        for (int k = 0; k < NU; ++k) { pin.lo0[k] = lo0[k]; pin.hi0[k] = hi0[k]; }
        // rows starting a solve: the current SQP guess (the shifted previous solution on warm steps: nearly the right
        // working set), clipped into the box and rolled out through the linearised model.  The linearisation point
        // (Xg, Ug) stays untouched until the solve is over.
        const bool start = running && !qp.busy();
        double J0 = 0.0;
        M4Q_PHASE_MARK(3)
        if (__any(start)) {
          J0 = rollout_open<S, NS, NU>(prov, fresh(T), x_cur, win, cost, Ug, pin.box, lo0, hi0, Xo, Uo, j, start && lane_ok);
          wave_sync();
        }
        M4Q_PHASE_MARK(14)
        bool bad_start = false;
        if (start) {
          qp.begin(J0);
          if (!finite_d(J0)) { qp.busy() = false; bad_start = true; }
        }
        if (__any(qp.busy())) ++qp_passes;
        bool ended;
        if constexpr (exact_tile<S, TL, EXACT>()) {
          // the pinned sweep on tiles: what a row hands to the tile layout (member = (lane >> 2) & 3 there, lane >> 4 here)
          ExactTile<NS> xt;
          xt.lds_models = reinterpret_cast<const double*>(lds);
          xt.tgb = tgb; xt.tiw = tiw; xt.tpin = tpin;
          xt.Xg = Xg; xt.Ug = Ug; xt.gains = gains;
          xt.Q = reinterpret_cast<const double*>(cost.Q); xt.Qf = reinterpret_cast<const double*>(cost.Qf); xt.R = reinterpret_cast<const double*>(cost.R);
          xt.sX = sX; xt.sU = sU; xt.sG = sG;
          xt.g = g; xt.jj = jj;
          ended = box_qp_iterate<S, NS, NU>(prov, fresh(T), x_cur, win, cost, flags, gains, pin, Xo, Uo, Xalt, Ualt, qp, j, jj, lane_ok, &pc, xt);
        } else {
          ended = box_qp_iterate<S, NS, NU>(prov, fresh(T), x_cur, win, cost, flags, gains, pin, Xo, Uo, Xalt, Ualt, qp, j, jj, lane_ok, &pc);
        }
        solved = running && (ended || bad_start);
        capped = solved && !bad_start && qp.stats.end_cap > 0;
        chk = qp.Jk;
        GView Xs = Xo, Us = Uo;
        Xs.off = qp.cur_is_a() ? Xo.off : Xalt.off;
        Us.off = qp.cur_is_a() ? Uo.off : Ualt.off;
        if (solved && jj == 0) {
          M4Q_GLOBAL unsigned long long* cnt = (M4Q_GLOBAL unsigned long long*)kargs()->queue;
          __hip_atomic_fetch_add(cnt + 1, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          __hip_atomic_fetch_add(cnt + 2, (unsigned long long)qp.stats.sweeps, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          __hip_atomic_fetch_add(cnt + 3, (unsigned long long)qp.stats.ratio_steps, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          __hip_atomic_fetch_add(cnt + 4, (unsigned long long)qp.stats.end_kkt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          __hip_atomic_fetch_add(cnt + 5, (unsigned long long)qp.stats.end_precision, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          __hip_atomic_fetch_add(cnt + 6, (unsigned long long)qp.stats.end_cap, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        if (solved && !bad_start) {
#pragma unroll
          for (int k = 0; k < NU; ++k) uapp[k] = Us.ld<double>(k);
          // (the row's 16 lanes share the contiguous elements; a batch issues all its loads before the first store)
          auto copy_x = [&](const GView& dst, const GView& src, int shift) __attribute__((always_inline)) {
            constexpr int U = 12;
            const int count = (T + 1) * NS, last = T * NS;
            for (int e0 = jj; e0 < count; e0 += 16 * U) {
              S v[U];
#pragma unroll
              for (int u = 0; u < U; ++u) {
                const int e = e0 + 16 * u < count ? e0 + 16 * u : count - 1;
                v[u] = src.ld<S>(e < last ? e + shift : e);
              }
              __builtin_amdgcn_sched_barrier(0);
#pragma unroll
              for (int u = 0; u < U; ++u)
                if (e0 + 16 * u < count) dst.st<S>(e0 + 16 * u, v[u]);
            }
          };
          auto copy_u = [&](const GView& dst, const GView& src, int shift) __attribute__((always_inline)) {
            constexpr int U = 8;
            const int count = T * NU, last = (T - 1) * NU;
            for (int e0 = jj; e0 < count; e0 += 16 * U) {
              double v[U];
#pragma unroll
              for (int u = 0; u < U; ++u) {
                const int e = e0 + 16 * u < count ? e0 + 16 * u : count - 1;
                v[u] = src.ld<double>(e < last ? e + shift : e);
              }
              __builtin_amdgcn_sched_barrier(0);
#pragma unroll
              for (int u = 0; u < U; ++u)
                if (e0 + 16 * u < count) dst.st<double>(e0 + 16 * u, v[u]);
            }
          };
          if (use_ls) {
            if (!qp.cur_is_a()) {
              copy_x(Xo, Xs, 0);
              copy_u(Uo, Us, 0);
            }
          } else {
            // warm step (alpha = 1, mpc.py:208-212): the solution becomes the next guess, shifted (mpc.py:271-272)
            copy_x(Xg, Xs, NS);
            copy_u(Ug, Us, NU);
          }
        }
      }
    }
    wave_sync();
    M4Q_PHASE_MARK(EXACT ? 15 : 2)
#if defined(M4Q_DEV_PHASE_CLOCK)
    if (__any(running && !(kargs()->warm_start && step > 1))) pc.count(9);      // passes with a line search
#endif
    // back from the stash
    b = ((long)stash.get_int(g, 1) << 32) | (long)(unsigned)stash.get_int(g, 0);
    code = stash.get_int(g, 2);
    done_steps = stash.get_int(g, 3);
    row_begin = stash.get_int(g, 4);
    {
      const int fl = stash.get_int(g, 5);
      active = (fl & 1) != 0; need_new = (fl & 2) != 0; pending = (fl & 4) != 0;
    }
    op0_off = (unsigned)stash.get_int(g, 6);
    ops_off = (unsigned)stash.get_int(g, 7);
    stash.get_x(x_meas);
    // exit code 3: non-finite objective (mpc.py:200-203).  exit code 2 (EXACT only): the solver gave up - the analogue of
    // the solver warning mpc.py:183-197 turns into code 2; either way the member's run ends here (mpc.py:196,203,231).
    const bool fail = (M4Q_EXP & 1) ? false : (!finite_d(chk) || capped);
    if (solved) ++iter;
    double alpha = 1.0;
    bool fin = true;
    if (__any(solved && use_ls)) {
      KArgs* a = kargs();
      ZView<NS, NU> z;
      z.T = T; z.Xg = Xg; z.Xo = Xo; z.Xt = win.xbm; z.Ug = Ug; z.Uo = Uo; z.Ut = win.ubm;
      double al = 1.0, stepn = 0.0;
      if constexpr (TL) {
        // (the host selects TL only with diagonal costs)
        if constexpr (M4Q_LS_NODES && NS % 2 == 0) line_search_tl_nodes<NX, NU, DD, TILE>(z, ldsW, ldsW + 2 * NX, ldsW + 4 * NX, jj, al, stepn);
        else line_search_tl<NX, NU, DD>(z, ldsW, ldsW + 2 * NX, ldsW + 4 * NX, jj, al, stepn);
      } else if (ls_diag) {
        line_search_diag<S, NX, NU, DD>(z, ldsW, ldsW + 2 * NX, ldsW + 4 * NX, jj, al, stepn);
      } else {
        if constexpr (sizeof(S) == sizeof(cplx)) line_search<NX, NU>(z, a->Cq, a->Cqf, a->Cr, jj, al, stepn);
      }
      if (use_ls) { alpha = al; fin = (M4Q_EXP & 1) ? iter >= 3 : stepn < a->ls_tol; }   // mpc.py:224
    }
    wave_sync();
    M4Q_PHASE_MARK(4)
    const bool upd = solved && !fail && use_ls;       // warm steps wrote the shifted guess in the rollout
    // X_guess += alpha (X_opt - X_guess) (mpc.py:228-229)
    // (these small per-element passes are latency bound: unrolled so that several loads are in flight)
    // (the row's 16 lanes share the (T + 1) NS contiguous elements, and a batch of U elements per lane issues all its loads
    //  before the first store: left to the compiler the unrolled loop waited for every pair in turn - 22,000 cycles per update)
    if constexpr (M4Q_UPD16 && NS % 2 == 0 && sizeof(S) == sizeof(double)) {
      // (n even: rows of the workspace start on 16-byte boundaries and hold an even number of doubles - two elements per access)
      if (upd) {
        constexpr int U = 6;
        const int count = (T + 1) * NS;
        for (int e0 = 2 * jj; e0 < count; e0 += 32 * U) {
          double xg[U][2], xo[U][2];
#pragma unroll
          for (int u = 0; u < U; ++u) {
            const int e = e0 + 32 * u < count ? e0 + 32 * u : count - 2;
            ldn<2>(Xg, e, xg[u]);
            ldn<2>(Xo, e, xo[u]);
          }
          __builtin_amdgcn_sched_barrier(0);
#pragma unroll
          for (int u = 0; u < U; ++u) {
            if (e0 + 32 * u < count) {
              double r[2];
#pragma unroll
              for (int h = 0; h < 2; ++h) r[h] = cadd(xg[u][h], cscale(csub(xo[u][h], xg[u][h]), alpha));
              stn<2>(Xg, e0 + 32 * u, r);
            }
          }
        }
      }
    } else if (upd) {
      constexpr int U = 12;
      const int count = (T + 1) * NS;
      for (int e0 = jj; e0 < count; e0 += 16 * U) {
        S xg[U], xo[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
          const int e = e0 + 16 * u < count ? e0 + 16 * u : count - 1;
          xg[u] = Xg.ld<S>(e);
          xo[u] = Xo.ld<S>(e);
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int u = 0; u < U; ++u)
          if (e0 + 16 * u < count) Xg.st<S>(e0 + 16 * u, cadd(xg[u], cscale(csub(xo[u], xg[u]), alpha)));
      }
    }
    if constexpr (M4Q_UPD16 && NU % 2 == 0) {
      if (upd) {
        constexpr int V = 4;
        const int cu = T * NU;
        for (int e0 = 2 * jj; e0 < cu; e0 += 32 * V) {
          double ug[V][2], uo[V][2];
#pragma unroll
          for (int u = 0; u < V; ++u) {
            const int e = e0 + 32 * u < cu ? e0 + 32 * u : cu - 2;
            ldn<2>(Ug, e, ug[u]);
            ldn<2>(Uo, e, uo[u]);
          }
          __builtin_amdgcn_sched_barrier(0);
#pragma unroll
          for (int u = 0; u < V; ++u) {
            if (e0 + 32 * u < cu) {
              double r[2];
#pragma unroll
              for (int h = 0; h < 2; ++h) r[h] = ug[u][h] + alpha * (uo[u][h] - ug[u][h]);
              stn<2>(Ug, e0 + 32 * u, r);
            }
          }
        }
      }
    } else if (upd) {
      {
        constexpr int V = 8;
        const int cu = T * NU;
        for (int e0 = jj; e0 < cu; e0 += 16 * V) {
          double ug[V], uo[V];
#pragma unroll
          for (int u = 0; u < V; ++u) {
            const int e = e0 + 16 * u < cu ? e0 + 16 * u : cu - 1;
            ug[u] = Ug.ld<double>(e);
            uo[u] = Uo.ld<double>(e);
          }
          __builtin_amdgcn_sched_barrier(0);
#pragma unroll
          for (int u = 0; u < V; ++u)
            if (e0 + 16 * u < cu) Ug.st<double>(e0 + 16 * u, ug[u] + alpha * (uo[u] - ug[u]));
        }
      }
    }
    wave_sync();

    M4Q_PHASE_MARK(8)
    // ---- rows that finished their MPC step: apply, propagate, shift ----
    bool step_done;
    {
      KArgs* a = kargs();
      step_done = solved && (fail || fin || iter >= a->max_iter);
    }
    if (__any(step_done)) {
      KArgs* a = kargs();
      const int mf = a->measure_freq, n_steps = a->n_steps;
      const long sXs = (long)(n_steps + 1) * NX, sUs = (long)n_steps * NU;
      if (step_done && fail) code = finite_d(chk) ? 2 : 3;
      if (step_done && jj == 0) gst(a->qp_solves, b * n_steps + step, iter);
      const bool ok = step_done && !fail;
      // apply U_opt[:, 0] (mpc.py:250), propagate the plant (mpc.py:256-260)
      if (ok) {
#pragma unroll
        for (int k = 0; k < NU; ++k) uprev[k] = step >= 1 ? uapp[k] : uprev[k];     // (step 1 still takes U_ref[:, 0]: mpc.py:185)
        if (jj == 0) {
#pragma unroll
          for (int k = 0; k < NU; ++k) gst(a->us, b * sUs + (long)step * NU + k, uapp[k]);
        }
      }
      if constexpr (PLANT != PLANT_NONE) {
        // (step+1) % measure_freq == 0: propagate the plant from the last measured state over the last measure_freq
        // intervals (mpc.py:252-260); the held controls are stacked newest first against an increasing time grid
        // (:257), i.e. replayed in reversed order.  Otherwise the model closes the loop (mpc.py:261-267).
        const bool measure = (step + 1) % mf == 0;
        cplx xn = czero();
        if (__any(ok && measure)) {
          GView op0 = gview(a->op0, 0, 0), ops = gview(a->ops, 0, 0);
          op0.off = op0_off;
          ops.off = ops_off;
          const double dt = a->dt;
          cplx xc = Basis<S, TL>::template to_complex<NX, DD>(x_meas, tau, scratch, j, jj);
          wave_sync();
          for (int i = 0; i < mf; ++i) {
            double ui[NU];
#pragma unroll
            for (int k = 0; k < NU; ++k)
              ui[k] = (i == 0 || !(ok && measure)) ? uapp[k] : gld(a->us, b * sUs + (long)(step - i) * NU + k);
            if constexpr (PLANT == PLANT_HAMILTONIAN) xc = plant_hamiltonian<NX, NU, DD>(xc, ui, op0, ops, dt, scratch, jio, jj);
            else xc = plant_generator<NX, NU>(xc, ui, op0, ops, dt, jio);
          }
          xn = xc;
        }
        double tnew = 0.0;
        S rn = Basis<S, TL>::template to_state<NX, DD>(xn, scratch, j, jj, tnew);
        if (mf > 1 && __any(ok && !measure)) {
          typename Prov::Lin lin;
#pragma unroll
          for (int k = 0; k < NU; ++k) lin.u[k] = SG ? uapp[k] * gsc[k] : uapp[k];        // (SG: the provider works on the scaled controls)
          lin.xg = x_cur;
          S pred, Bdummy[NU], ddummy;
          prov.rows(lin, x_cur, pred, Bdummy, ddummy);                 // A x + N (polyu (x) x) = A_t(u) x  (model.py:81-93)
          const cplx pc = Basis<S, TL>::template to_complex<NX, DD>(pred, tau, scratch, j, jj);
          if (!measure) { rn = pred; xn = pc; }
        }
        if (ok) x_cur = rn;
        if (ok && measure) { x_meas = rn; if constexpr (TL) tau = tnew; }
        if (ok && lane_io) gst(a->xs, b * sXs + (long)(step + 1) * NX + jio, xn);
      }
      // shift_guess (mpc.py:71-73,271-272): drop column 0, repeat the last
      const bool shift_now = ok && use_ls;          // (a warm step's rollout has already shifted)
      if (shift_now && lane_ok) {
        for (int t0 = 0; t0 < T; t0 += 8) {
          S buf[8];
#pragma unroll
          for (int q = 0; q < 8; ++q) buf[q] = Xg.ld<S>((t0 + q + 1 <= T ? t0 + q + 1 : T) * NS + j);
#pragma unroll
          for (int q = 0; q < 8; ++q)
            if (t0 + q < T) Xg.st<S>((t0 + q) * NS + j, buf[q]);
        }
      }
      if (shift_now && jj < NU) {
        for (int t0 = 0; t0 + 1 < T; t0 += 8) {
          double buf[8];
#pragma unroll
          for (int q = 0; q < 8; ++q) buf[q] = Ug.ld<double>((t0 + q + 1 < T ? t0 + q + 1 : T - 1) * NU + jj);
#pragma unroll
          for (int q = 0; q < 8; ++q)
            if (t0 + q + 1 < T) Ug.st<double>((t0 + q) * NU + jj, buf[q]);
        }
      }
      if (ok) done_steps = step + 1;
      if (step_done) {
        iter = 0;
        step = fail ? row_end : step + 1;
      }
      wave_sync();
      if constexpr (PLANT == PLANT_NONE) {
        // the caller writes xs[step+1] before the next launch; inside one launch carry what is there
        const bool carry = ok && step < row_end;
        const cplx xc = carry ? gld(a->xs, b * sXs + (long)step * NX + jio) : czero();
        double tnew = 0.0;
        const S rn = Basis<S, TL>::template to_state<NX, DD>(xc, scratch, j, jj, tnew);
        if (carry) { x_cur = rn; if constexpr (TL) tau = tnew; }
      }
    }

    M4Q_PHASE_MARK(5)
    // ---- rows that finished their item: publish the resumable state, free the slot ----
    const bool finished = active && step >= row_end;
    if (__any(finished)) {
      KArgs* a = kargs();
      // The last item of an instance publishes the guess as the host sees it: complex, original basis (eight nodes' loads in flight
      // at once; the basis change of each goes through LDS).  An earlier item's guess is only ever read by the next item of the same
      // launch: on the real paths it goes into the same field as it is (PIECE_RAW: (T + 1) NS doubles, flat) - 41 basis changes less
      // on either side of every cut, and no rounding at the cuts.
      const bool last_piece = row_end == a->step_end;
      if constexpr (TL && M4Q_PUBLISH_NODES && NS % 2 == 0) {
        // (traceless path, n_s even: one trajectory node per lane - its coordinates as 16-byte pairs, its NX slots formed and stored
        //  by that lane (tl_node_to_complex) - instead of a slot per lane with an exchange through LDS per node)
        if (finished && last_piece) {
          for (int t = jj; t <= T; t += 16) {
            double r[NS];
#pragma unroll
            for (int h = 0; h < NS / 2; ++h) {
              double p2[2];
              ldn<2>(Xg, (unsigned)(t * NS + 2 * h), p2);
              r[2 * h] = p2[0]; r[2 * h + 1] = p2[1];
            }
            cplx out[NX];
            tl_node_to_complex<NX, DD>(r, tau, out);
#pragma unroll
            for (int c = 0; c < NX; ++c) gst(a->Xg, b * sXc + t * NX + c, out[c]);
          }
        }
      } else if (__any(finished && (!PIECE_RAW || last_piece))) {
        for (int t0 = 0; t0 <= T; t0 += 8) {
          S v[8];
#pragma unroll
          for (int q = 0; q < 8; ++q) v[q] = Xg.ld<S>((t0 + q <= T ? t0 + q : T) * NS + j);
          __builtin_amdgcn_sched_barrier(0);
#pragma unroll
          for (int q = 0; q < 8; ++q) {
            if (t0 + q <= T) {
              const cplx xc = Basis<S, TL>::template to_complex<NX, DD>(v[q], tau, scratch, j, jj);
              if (finished && (!PIECE_RAW || last_piece) && lane_io) gst(a->Xg, b * sXc + (t0 + q) * NX + jio, xc);
            }
          }
        }
      }
      if constexpr (PIECE_RAW) {
        if (finished && !last_piece) {
          M4Q_GLOBAL double* raw = (M4Q_GLOBAL double*)(a->Xg + b * sXc);
          const int count = (T + 1) * NS;
          if constexpr (M4Q_UPD16 && NS % 2 == 0) {
            constexpr int U = 6;
            for (int e0 = 2 * jj; e0 < count; e0 += 32 * U) {
              double v[U][2];
#pragma unroll
              for (int u = 0; u < U; ++u) ldn<2>(Xg, e0 + 32 * u < count ? e0 + 32 * u : count - 2, v[u]);
              __builtin_amdgcn_sched_barrier(0);
#pragma unroll
              for (int u = 0; u < U; ++u) {
                if (e0 + 32 * u < count) {
                  d2_t w; w.x = v[u][0]; w.y = v[u][1];
                  *reinterpret_cast<M4Q_GLOBAL d2_t*>(raw + e0 + 32 * u) = w;
                }
              }
            }
          } else {
            constexpr int U = 12;
            for (int e0 = jj; e0 < count; e0 += 16 * U) {
              double v[U];
#pragma unroll
              for (int u = 0; u < U; ++u) v[u] = Xg.ld<double>(e0 + 16 * u < count ? e0 + 16 * u : count - 1);
              __builtin_amdgcn_sched_barrier(0);
#pragma unroll
              for (int u = 0; u < U; ++u)
                if (e0 + 16 * u < count) gst(raw, e0 + 16 * u, v[u]);
            }
          }
        }
      }
      if (finished) {
        for (int e = jj; e < T * NU; e += 16) gst(a->Ug, b * sU + e, Ug.ld<double>(e));
        if (jj == 0) {
          gst(a->codes, b, code);
          gst(a->steps_done, b, done_steps);
        }
      }
      if (two_phase && __any(finished && row_end != kargs()->step_end)) {
        // head item: make the state visible to whichever workgroup draws the tail (G16: stores drained,
        // agent-scope release, drained again, then the flag)
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        wave_sync();
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (finished && row_end != a->step_end && jj == 0)
          __hip_atomic_store(a->head_done + b, piece_of(row_begin, a->step_begin) + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
      if (finished) {
        active = false;
        need_new = true;
      }
      wave_sync();
    }
    M4Q_PHASE_MARK(6)
    pc.count(7);
  }
}

#ifndef M4Q_NO_AUX          // (a plant-only shape - m4q_shapes.inc - builds plant_kernel alone; the generator-plant object none of these)
// ---------------------------------------------------------------------------------------------
// WrapModel.get_model_along_traj for B trajectories (linearize.py:61-70)
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(64) M4Q_OCC void linearize_kernel(LinArgs a) {
  cplx* lds = reinterpret_cast<cplx*>(m4q_lds_raw);
  const LaneGeo L;
  const int g = L.g, jj = L.jj, j = L.j;
  cplx* mdl = lds + g * MODEL_ELEMS;
  const int nquads = (a.B + ROWS - 1) / ROWS;
  const unsigned sX = (unsigned)a.T * NX, sU = (unsigned)a.T * NU;
  for (int quad = blockIdx.x; quad < nquads; quad += gridDim.x) {
    const long q0 = (long)quad * ROWS;
    const bool valid = q0 + g < a.B;
    const unsigned gl = valid ? g : (unsigned)(a.B - 1 - q0);
    wave_sync();
    stage_model<NX>(mdl, a.models + (q0 + gl) * a.model_stride, jj);
    wave_sync();
    FusedProv<cplx, NX, NU, ORDER> prov;
    prov.mdl = mdl;
    prov.Xg = gview(a.X, q0 * sX, gl * sX);
    prov.Ug = gview(a.U, q0 * sU, gl * sU);
    prov.j = j;
    const GView Ao = gview(a.A_ls, q0 * sX * NX, gl * sX * NX);
    const GView Bo = gview(a.B_ls, q0 * sX * NU, gl * sX * NU);
    const GView Do = gview(a.D_ls, q0 * sX, gl * sX);
    for (int t = 0; t < a.T; ++t) {
      const auto lin = prov.fetch(t);
      cplx Ac[NX];
      prov.col(lin, Ac);
      cplx av, Brow[NU], dlt;
      prov.rows(lin, czero(), av, Brow, dlt);
      if (valid && L.lane_ok) {
#pragma unroll
        for (int i = 0; i < NX; ++i) Ao.st<cplx>((t * NX + i) * NX + j, Ac[i]);
#pragma unroll
        for (int k = 0; k < NU; ++k) Bo.st<cplx>((t * NX + j) * NU + k, Brow[k]);
        Do.st<cplx>(t * NX + j, dlt);
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------
// quad_program for B explicit linear time-varying problems (optimize.py:12-60 / lqr.py:14-79)
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(64) M4Q_OCC void qp_kernel(QpArgs a) {
  const LaneGeo L;
  const int g = L.g, jj = L.jj, j = L.j;
  const int T = a.T;
  const int nquads = (a.B + ROWS - 1) / ROWS;
  CostRef<cplx> cost;
  // (stage costs of the explicit QP stay in global memory: generic pointers the compiler traces back to the kernel argument)
  cost.Q = (const cplx*)a.Q_ls; cost.Qf = (const cplx*)a.Q_ls + (long)T * NX * NX; cost.q_stride = (long)NX * NX;
  cost.R = (const cplx*)a.R_ls; cost.r_stride = (long)NU * NU;
  const unsigned sX = (unsigned)(T + 1) * NX, sU = (unsigned)T * NU, sG = (unsigned)T * (NX + 1) * NU;
  const unsigned sA = (unsigned)T * NX * NX, sB = (unsigned)T * NX * NU, sD = (unsigned)T * NX;
  for (int quad = blockIdx.x; quad < nquads; quad += gridDim.x) {
    const long q0 = (long)quad * ROWS;
    const bool valid = q0 + g < a.B;
    const unsigned gl = valid ? g : (unsigned)(a.B - 1 - q0);
    const long b = q0 + gl;
    ExplicitProv<NX, NU> prov;
    prov.A_ls = gview(a.A_ls, q0 * sA, gl * sA);
    prov.B_ls = gview(a.B_ls, q0 * sB, gl * sB);
    prov.has_delta = a.D_ls != nullptr;
    prov.D_ls = gview(a.D_ls ? a.D_ls : a.A_ls, q0 * sD, gl * sD);
    prov.j = j;
    Window win;
    win.xbm = gview(a.X_bm, q0 * a.xbm_stride, gl * (unsigned)a.xbm_stride);
    win.ubm = gview(a.U_bm, q0 * a.ubm_stride, gl * (unsigned)a.ubm_stride);
    const GView gains = gview(a.gains, q0 * sG, gl * sG);
    const GView Xo = gview(a.X_opt, q0 * sX, gl * sX);
    const GView Uo = gview(a.U_opt, q0 * sU, gl * sU);
    const bool st = valid && L.lane_ok;
    riccati_backward<cplx, NX, NU>(prov, T, win, cost, a.flags, gains, j, st);
    wave_sync();
    double lo0[NU], hi0[NU], u_first[NU];
#pragma unroll
    for (int k = 0; k < NU; ++k) {
      const bool band = (a.flags & QP_DU_BAND) != 0 && a.u_prev != nullptr;
      const double up = band ? gld(a.u_prev, b * NU + k) : 0.0;
      lo0[k] = band ? up - a.du : -a.sat;
      hi0[k] = band ? up + a.du : a.sat;
    }
    const cplx x0 = gld(a.x_init, b * NX + j);
    const bool exact = (a.flags & QP_EXACT_BOX) != 0;
    // the exact solver ping-pongs between two trajectory pairs of ONE allocation (rows pick theirs by lane offset):
    // X_alt = [B][T+1][n] twice, U_alt likewise; the clipped rollout, its starting point, goes straight into the first
    const GView Xa = gview(exact ? a.X_alt : a.X_opt, q0 * sX, gl * sX);
    const GView Ua = gview(exact ? a.U_alt : a.U_opt, q0 * sU, gl * sU);
    const double obj = rollout_forward<cplx, NX, NU, true>(prov, T, x0, win, cost, a.flags, gains, a.sat, lo0, hi0, Xa, Ua, j, st,
                                                              u_first);
    wave_sync();
    double obj_out = obj;
    if (exact) {
      const GView Xb = gview(a.X_alt, q0 * sX, gl * sX + (unsigned)a.B * sX);
      const GView Ub = gview(a.U_alt, q0 * sU, gl * sU + (unsigned)a.B * sU);
      const GView stat = gview(a.pin_stat, q0 * sU, gl * sU);
      PinCtx<NU> pin;
      pin.stat = stat;
      pin.box.sat = a.sat;
#pragma unroll
      for (int k = 0; k < NU; ++k) { pin.lo0[k] = lo0[k]; pin.hi0[k] = hi0[k]; }
      BoxQpRow row;
      if (valid) row.begin(obj);
      while (__any(row.busy())) box_qp_iterate<cplx, NX, NU>(prov, T, x0, win, cost, a.flags, gains, pin, Xa, Ua, Xb, Ub, row, j, jj, L.lane_ok);
      obj_out = row.Jk;
      const QpStats& stats = row.stats;
      GView Xs = Xa, Us = Ua;
      Xs.off = row.cur_is_a() ? Xa.off : Xb.off;
      Us.off = row.cur_is_a() ? Ua.off : Ub.off;
      if (st) {
        for (int t = 0; t <= T; ++t) Xo.st<cplx>(t * NX + j, Xs.ld<cplx>(t * NX + j));
        if (j == 0)
          for (int i = 0; i < T * NU; ++i) Uo.st<double>(i, Us.ld<double>(i));
      }
      if (valid && jj == 0 && a.sweep_counts) gst(a.sweep_counts, b, stats.sweeps);
      wave_sync();
    }
    if (valid && jj == 0) gst(a.cost, b, obj_out);
  }
}

// ---------------------------------------------------------------------------------------------
// One held-control plant step for B states (experiment.py:202-212)
#endif  // M4Q_NO_AUX

#ifndef M4Q_VARIANT_GEN
// ---------------------------------------------------------------------------------------------
template <int PLANT>
__global__ __launch_bounds__(64) M4Q_OCC void plant_kernel(PlantArgs a) {
  cplx* lds = reinterpret_cast<cplx*>(m4q_lds_raw);
  const LaneGeo L;
  const int g = L.g, jj = L.jj, j = L.j;
  cplx* scratch = lds + g * SCRATCH_ELEMS;
  const int nquads = (a.B + ROWS - 1) / ROWS;
  for (int quad = blockIdx.x; quad < nquads; quad += gridDim.x) {
    const long q0 = (long)quad * ROWS;
    const bool valid = q0 + g < a.B;
    const unsigned gl = valid ? g : (unsigned)(a.B - 1 - q0);
    const long b = q0 + gl;
    const cplx x = gld(a.x, b * NX + j);
    double u[NU];
#pragma unroll
    for (int k = 0; k < NU; ++k) u[k] = gld(a.u, b * NU + k);
    const GView op0 = gview(a.op0, q0 * a.op0_stride, gl * (unsigned)a.op0_stride);
    const GView ops = gview(a.ops, q0 * a.ops_stride, gl * (unsigned)a.ops_stride);
    cplx xn;
    if constexpr (PLANT == PLANT_HAMILTONIAN)
      xn = plant_hamiltonian<NX, NU, DD>(x, u, op0, ops, a.dt, scratch, j, jj);
    else
      xn = plant_generator<NX, NU>(x, u, op0, ops, a.dt, j);
    if (valid && jj < NX) gst(a.x_next, b * NX + j, xn);
  }
}

// ---------------------------------------------------------------------------------------------
// discretize_homogeneous for B generator sets (vectorize.py:8-49): Taylor/Dyson expansion of
// exp(dt (G_0 + sum_k u_k G_k)) to ORDER, every word of operators multiplied out and binned by its control
// monomial.  One row per instance; generators staged in LDS, products column-owned in registers.
#endif  // M4Q_VARIANT_GEN

// ---------------------------------------------------------------------------------------------
constexpr int find_monomial(int c0, int c1, int c2) {
  constexpr PowTab<NU, ORDER> tab{};
  const int want[3] = {c0, c1, c2};
  for (int p = 0; p < PowTab<NU, ORDER>::COUNT; ++p) {
    bool same = true;
    for (int k = 0; k < NU; ++k) same = same && tab.e[p][k] == want[k];
    if (same) return p;
  }
  return -1;
}
constexpr int word_block(int a, int b) {      // monomial of a word of one (b < 0) or two letters; letter 0 = drift
  int c[3] = {0, 0, 0};
  if (a > 0) ++c[a - 1];
  if (b > 0) ++c[b - 1];
  return find_monomial(c[0], c[1], c[2]);
}

// N: matrix dimension (NX, or NX - 1 for the traceless blocks of the lifted generators)
template <class S, int N>
__global__ __launch_bounds__(64) void discretize_kernel(DiscArgs a) {
  static_assert(ORDER == 1 || ORDER == 2, "orders 1 and 2 are compiled");
  S* lds = reinterpret_cast<S*>(m4q_lds_raw);
  const LaneGeo L;
  const int g = L.g, jj = L.jj;
  const int j = jj < N ? jj : N - 1;
  constexpr int NX = N;                           // (shadows the shape's NX inside this kernel)
  constexpr int GEN_ELEMS = (1 + NU) * NX * NX;
  constexpr int W = NX * (1 + NP);
  S* G = lds + g * GEN_ELEMS;                      // [1+m][n][n] row-major, scaled
  const int nquads = (a.B + ROWS - 1) / ROWS;
  const M4Q_GLOBAL S* gens = (const M4Q_GLOBAL S*)a.gens;
  M4Q_GLOBAL S* models = (M4Q_GLOBAL S*)a.models;
  for (int quad = blockIdx.x; quad < nquads; quad += gridDim.x) {
    const long q0 = (long)quad * ROWS;
    const bool valid = q0 + g < a.B;
    const long b = valid ? q0 + g : a.B - 1;
    wave_sync();
    for (int e = jj; e < GEN_ELEMS; e += 16) {
      const double sc = a.scales ? gld(a.scales, b * (1 + NU) + e / (NX * NX)) : 1.0;
      G[e] = cscale(gld(gens, b * a.gen_stride + e), sc);
    }
    wave_sync();
    M4Q_GLOBAL S* out = models + b * (long)NX * W;
    // accumulate every block's column j in registers: blk[p][i]
    S blk[1 + NP][NX];
#pragma unroll
    for (int p = 0; p <= NP; ++p)
#pragma unroll
      for (int i = 0; i < NX; ++i) blk[p][i] = zero_of<S>();
#pragma unroll
    for (int i = 0; i < NX; ++i) blk[0][i] = from_real<S>(i == j ? 1.0 : 0.0);       // k = 0: identity
    static_for<0, 1 + NU>([&](auto aa) {
      constexpr int la = decltype(aa)::value;
      S Ga[NX];                                       // column j of G_a
#pragma unroll
      for (int i = 0; i < NX; ++i) Ga[i] = G[(la * NX + i) * NX + j];
      constexpr int p1 = word_block(la, -1);
#pragma unroll
      for (int i = 0; i < NX; ++i) cmac_r(blk[p1][i], Ga[i], a.dt);                    // k = 1: dt G_a
      if constexpr (ORDER >= 2) {
        static_for<0, 1 + NU>([&](auto bb) {
          constexpr int lb = decltype(bb)::value;
          S Gb[NX], prod[NX];
#pragma unroll
          for (int i = 0; i < NX; ++i) Gb[i] = G[(lb * NX + i) * NX + j];
          matmul_cols<NX>(prod, Ga, Gb);             // column j of G_a G_b (word "a then b": entry @ A_a @ A_b)
          constexpr int p2 = word_block(la, lb);
#pragma unroll
          for (int i = 0; i < NX; ++i) cmac_r(blk[p2][i], prod[i], 0.5 * a.dt * a.dt);
        });
      }
    });
    if (valid && jj < N) {
#pragma unroll
      for (int p = 0; p <= NP; ++p)
#pragma unroll
        for (int i = 0; i < NX; ++i) gst(out, i * W + p * NX + j, blk[p][i]);
    }
  }
}

// ---------------------------------------------------------------------------------------------
// host-side launchers for this shape
// ---------------------------------------------------------------------------------------------
static size_t mpc_lds_bytes(int path, int exact) {
  if constexpr (HAS_SG) { if (path == 4 && !exact) return mpc_lds_layout_bytes<double, true, false, false, true>(); }
  if constexpr (SQUARE) {
    if constexpr (HAS_TILE) { if (path == 3 && !exact) return mpc_lds_layout_bytes<double, true, true>(); }
    if (path >= 2) return exact ? mpc_lds_layout_bytes<double, true, false, true>() : mpc_lds_layout_bytes<double, true, false>();
  }
  if (path) return exact ? mpc_lds_layout_bytes<double, false, false, true>() : mpc_lds_layout_bytes<double>();
  return mpc_lds_layout_bytes<cplx>();
}

template <class K>
static int prep_lds(K kern, size_t bytes) {
  if (bytes > 64 * 1024) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
    if (e != hipSuccess) return -(int)e;
  }
  return 0;
}

// what to do with the kernel instance picked by (arithmetic path, plant kind, QP mode)
struct LaunchOp {
  const MpcArgs& a;
  int grid;
  hipStream_t s;
  int unsupported() const { return -(int)hipErrorInvalidValue; }
  template <class S, int PLANT, bool EXACT, bool TL, bool TILE, bool SG = false>
  int run() const {
    const size_t lds = mpc_lds_layout_bytes<S, TL, TILE, EXACT, SG>();
    int rc = prep_lds(mpc_kernel<S, PLANT, EXACT, TL, TILE, SG>, lds);
    if (rc) return rc;
    hipLaunchKernelGGL((mpc_kernel<S, PLANT, EXACT, TL, TILE, SG>), dim3(grid), dim3(64), lds, s, a);
    return -(int)hipGetLastError();
  }
};
struct OccupancyOp {
  int unsupported() const { return 0; }
  template <class S, int PLANT, bool EXACT, bool TL, bool TILE, bool SG = false>
  int run() const {
    int nb = 0;
    hipError_t e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, mpc_kernel<S, PLANT, EXACT, TL, TILE, SG>, 64,
                                                                 mpc_lds_layout_bytes<S, TL, TILE, EXACT, SG>());
    return e != hipSuccess ? -(int)e : nb;
  }
};

template <class S, bool EXACT, bool TL, bool TILE, class Op, bool SG = false>
static int pick_plant(const Op& op, int plant_kind) {
  if constexpr (!SQUARE) {
#ifdef M4Q_VARIANT_GEN
    return op.unsupported();
#else
    return op.template run<S, PLANT_NONE, EXACT, false, false>();
#endif
  } else {
#ifdef M4Q_VARIANT_GEN
    if (plant_kind == PLANT_GENERATOR) return op.template run<S, PLANT_GENERATOR, EXACT, TL, TILE, SG>();
    return op.unsupported();
#else
    if (plant_kind == PLANT_HAMILTONIAN) return op.template run<S, PLANT_HAMILTONIAN, EXACT, TL, TILE, SG>();
    if (plant_kind == PLANT_GENERATOR) return op.unsupported();          // (libm4q_hip_gen.so: the host routes such sessions there)
    return op.template run<S, PLANT_NONE, EXACT, TL, TILE, SG>();
#endif
  }
}

// path: 0 complex, 1 real (Hermitian basis, NX coordinates), 2 real traceless (NX - 1 coordinates), 3 traceless with the backward
// sweep on matrix-core tiles (clipped solve: its own kernel; exact solve: the pinned sweep inside the one exact kernel, chosen by
// QP_TARG_CONST and QP_NO_TILE in the flags)
template <class Op>
static int pick_kernel(const Op& op, int plant_kind, int path, int exact, int unsupported) {
  if constexpr (!SQUARE) {
    if (path || plant_kind != PLANT_NONE) return unsupported;
  }
  if constexpr (HAS_SG) {
    // path 4: the clipped traceless kernel on shared generators (sessions built by m4q_session_build_models; the host falls back to
    // path 2 - per-member models - for the exact mode)
    if (path == 4 && !exact) return pick_plant<double, false, true, false, Op, true>(op, plant_kind);
  }
  if (path == 4) path = 2;
  if constexpr (SQUARE) {
    if constexpr (HAS_TILE) { if (path == 3 && !exact) return pick_plant<double, false, true, true>(op, plant_kind); }
    if (path >= 2) return exact ? pick_plant<double, true, true, false>(op, plant_kind) : pick_plant<double, false, true, false>(op, plant_kind);
    if (path == 1) return exact ? pick_plant<double, true, false, false>(op, plant_kind) : pick_plant<double, false, false, false>(op, plant_kind);
  }
  return exact ? pick_plant<cplx, true, false, false>(op, plant_kind) : pick_plant<cplx, false, false, false>(op, plant_kind);
}

#ifndef M4Q_PLANT_ONLY
static int launch_mpc(const MpcArgs& a, int plant_kind, int path, int grid, hipStream_t s) {
  return pick_kernel(LaunchOp{a, grid, s}, plant_kind, path, (a.flags & QP_EXACT_BOX) != 0, -(int)hipErrorInvalidValue);
}

static int occupancy(int plant_kind, int path, int exact) {
  return pick_kernel(OccupancyOp{}, plant_kind, path, exact, 0);
}

#else
static int launch_mpc(const MpcArgs&, int, int, int, hipStream_t) { return -(int)hipErrorInvalidValue; }
static int occupancy(int, int, int) { return 0; }
#endif

static int grid_for(int B) {
  const int nquads = (B + ROWS - 1) / ROWS;
  return nquads < 4096 ? (nquads > 0 ? nquads : 1) : 4096;
}

#ifndef M4Q_NO_AUX
static int launch_linearize(const LinArgs& a, hipStream_t s) {
  const size_t lds = sizeof(cplx) * (size_t)(ROWS * MODEL_ELEMS);
  int rc = prep_lds(linearize_kernel, lds);
  if (rc) return rc;
  hipLaunchKernelGGL(linearize_kernel, dim3(grid_for(a.B)), dim3(64), lds, s, a);
  return -(int)hipGetLastError();
}

static int launch_qp(const QpArgs& a, hipStream_t s) {
  hipLaunchKernelGGL(qp_kernel, dim3(grid_for(a.B)), dim3(64), 0, s, a);
  return -(int)hipGetLastError();
}

#else
static int launch_linearize(const LinArgs&, hipStream_t) { return -(int)hipErrorInvalidValue; }
static int launch_qp(const QpArgs&, hipStream_t) { return -(int)hipErrorInvalidValue; }
#endif

#ifndef M4Q_VARIANT_GEN
static int launch_plant(const PlantArgs& a, hipStream_t s) {
  if constexpr (!SQUARE) {
    return -(int)hipErrorInvalidValue;
  } else {
  const size_t lds = sizeof(cplx) * (size_t)(ROWS * SCRATCH_ELEMS);
  if (a.kind == PLANT_HAMILTONIAN)
    hipLaunchKernelGGL(plant_kernel<PLANT_HAMILTONIAN>, dim3(grid_for(a.B)), dim3(64), lds, s, a);
  else
    hipLaunchKernelGGL(plant_kernel<PLANT_GENERATOR>, dim3(grid_for(a.B)), dim3(64), lds, s, a);
  return -(int)hipGetLastError();
  }
}

#else
static int launch_plant(const PlantArgs&, hipStream_t) { return -(int)hipErrorInvalidValue; }
#endif

#ifndef M4Q_NO_AUX
// path: 0 complex generators, 1 real n x n (lifted to the Hermitian basis), 2 real (n-1) x (n-1) (their traceless blocks)
static int launch_discretize(const DiscArgs& a, int path, hipStream_t s) {
  if (!SQUARE && path) return -(int)hipErrorInvalidValue;
  const size_t elems = (size_t)ROWS * (1 + NU) * NX * NX;
  if constexpr (SQUARE) {
    if (path == 2) {
      hipLaunchKernelGGL((discretize_kernel<double, NX - 1>), dim3(grid_for(a.B)), dim3(64), elems * sizeof(double), s, a);
      return -(int)hipGetLastError();
    }
  }
  if (path == 1) hipLaunchKernelGGL((discretize_kernel<double, NX>), dim3(grid_for(a.B)), dim3(64), elems * sizeof(double), s, a);
  else hipLaunchKernelGGL((discretize_kernel<cplx, NX>), dim3(grid_for(a.B)), dim3(64), elems * sizeof(cplx), s, a);
  return -(int)hipGetLastError();
}

#else
static int launch_discretize(const DiscArgs&, int, hipStream_t) { return -(int)hipErrorInvalidValue; }
#endif

static int power_list(int32_t* out) {
  constexpr PowTab<NU, ORDER> tab{};
  for (int p = 0; p < PowTab<NU, ORDER>::COUNT; ++p)
    for (int k = 0; k < NU; ++k) out[p * NU + k] = tab.e[p][k];
  return PowTab<NU, ORDER>::COUNT;
}

static const ShapeOps* shape_ops() {
#ifdef M4Q_PLANT_ONLY
  constexpr int plant_only = 1;
#else
  constexpr int plant_only = 0;
#endif
  static const ShapeOps ops = {NX, NU, ORDER, NP, DD, HAS_TILE ? 1 : 0, HAS_SG ? 1 : 0, plant_only, mpc_lds_bytes, launch_mpc, launch_linearize, launch_qp, launch_plant,
                               launch_discretize, power_list, occupancy};
  return &ops;
}

}  // namespace shape_<nx>_<nu>_<order>
}  // namespace m4q

// registration symbol of this shape: m4q_shape_<nx>_<nu>_<order> (linked into libm4q_hip.so), or - the generator-plant object -
// m4q_shapeg_<nx>_<nu>_<order>, which libm4q_hip.so looks up in libm4q_hip_gen.so with dlsym
#ifdef M4Q_VARIANT_GEN
extern "C" __attribute__((visibility("default"))) const m4q::ShapeOps* M4Q_CAT(m4q_shapeg_, M4Q_NX, M4Q_NU, M4Q_ORDER)() {
  return m4q::M4Q_CAT(shape_, M4Q_NX, M4Q_NU, M4Q_ORDER)::shape_ops();
}
#else
extern "C" __attribute__((visibility("hidden"))) const m4q::ShapeOps* M4Q_CAT(m4q_shape_, M4Q_NX, M4Q_NU, M4Q_ORDER)() {
  return m4q::M4Q_CAT(shape_, M4Q_NX, M4Q_NU, M4Q_ORDER)::shape_ops();
}
#endif
