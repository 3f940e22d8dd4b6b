// m4q_args.h - plain argument blocks passed by value to the kernels, shared by the per-shape
// kernel translation units (m4q_kernels.hip) and the host side of the C ABI (m4q_capi.hip).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

// Pointer fields of the argument blocks: plain pointers for the host pass; for the device pass the SAME 8 bytes typed as
// global-address-space pointers, so that a pointer read out of the kernarg segment at its point of use (kargs() in
// m4q_kernels.hip) still yields global_load / global_store instructions, not flat ones.
// (Only in the kernel translation units, which define M4Q_KERNEL_TU: the host code of m4q_capi.hip, which fills these fields, is
// also parsed by its device pass.)
#if defined(__HIP_DEVICE_COMPILE__) && defined(M4Q_KERNEL_TU)
#define M4Q_GLOBAL __attribute__((address_space(1)))
#else
#define M4Q_GLOBAL
#endif
#define M4Q_P(T) T M4Q_GLOBAL*

namespace m4q {

struct cplx;

// Arrays marked S are complex (cplx) on the general path and double on the real path (models that preserve
// Hermiticity, expressed in the Hermitian operator basis of m4q_mpc.h); the host picks the kernel.
struct MpcArgs {
  int B, T, n_steps, max_iter, warm_start, flags, step_begin, step_end, measure_freq;
  double dt, sat, du, ls_tol;
  M4Q_P(const void) models;  long model_stride;   // S [B|1][n][n(1+P)]
  M4Q_P(const cplx) x0c;                          // [B][n] complex, as given (becomes xs[:, 0])
  M4Q_P(const void) x0s;                          // S [B][n]
  M4Q_P(const void) x_targ;  long xt_stride;      // S [B|1][cols][n]
  M4Q_P(const double) u_targ; long ut_stride;     // [B|1][cols][m]
  M4Q_P(const void) Q; M4Q_P(const void) Qf; M4Q_P(const void) R;            // S
  M4Q_P(const double) Cq; M4Q_P(const double) Cqf; M4Q_P(const double) Cr;   // line-search blocks (mpc.py:103-116)
  M4Q_P(const double) Wls;                        // [2n + 2n + 2m] diagonals of those blocks, or nullptr if one is not diagonal
  M4Q_P(const cplx) op0; long op0_stride;         // plant operators
  M4Q_P(const cplx) ops; long ops_stride;
  M4Q_P(cplx) xs; M4Q_P(double) us; M4Q_P(int) codes; M4Q_P(int) steps_done; M4Q_P(int) qp_solves;
  M4Q_P(cplx) Xg; M4Q_P(double) Ug;                     // per-instance SQP guess  [B][T+1][n] complex, [B][T][m] (resumable state)
  // per resident row (grid*4 of them): working guess followed by the QP solution (S [2][rows][T+1][n], [2][rows][T][m]), gains (S)
  M4Q_P(void) ws_Xg; M4Q_P(double) ws_Ug; M4Q_P(void) ws_gains;
  M4Q_P(int) queue;                               // 64 B zeroed before every launch: [0] next work item to hand out; [1] set by the watchdog;
                                            // as u64 [1..3]: exact-QP counters (solves, Newton iterations, arc trials)
  M4Q_P(int) head_done;                           // [B] set when an instance's head item (steps < 2) has been published; zeroed likewise
  unsigned long long deadline_ticks;        // watchdog: the launch abandons itself (queue[1] = 1) once s_memrealtime (100 MHz) has
                                            // advanced this far since the wavefront started; every wavefront reaches this exit
  // shared-generator sessions (path 4): dt L_k on the recursion's coordinates, [1 + m][ns][ns] doubles, and the members' scales
  // [B][1 + m]; the kernel forms A_i = I + s_i0 dt L_0 itself and never reads `models`
  M4Q_P(const double) gens; M4Q_P(const double) scales;
};

struct LinArgs {
  int B, T;
  M4Q_P(const cplx) models; long model_stride;
  M4Q_P(const cplx) X; M4Q_P(const double) U;           // [B][T][n], [B][T][m]
  M4Q_P(cplx) A_ls; M4Q_P(cplx) B_ls; M4Q_P(cplx) D_ls;
};

struct QpArgs {
  int B, T, flags;
  double sat, du;
  M4Q_P(const cplx) x_init;
  M4Q_P(const cplx) X_bm; long xbm_stride;
  M4Q_P(const double) U_bm; long ubm_stride;
  M4Q_P(const cplx) Q_ls; M4Q_P(const cplx) R_ls;       // [T+1][n][n], [T][m][m]
  M4Q_P(const cplx) A_ls; M4Q_P(const cplx) B_ls; M4Q_P(const cplx) D_ls;
  M4Q_P(const double) u_prev;
  M4Q_P(cplx) X_opt; M4Q_P(double) U_opt; M4Q_P(double) cost; M4Q_P(cplx) gains;   // gains: caller buffer or workspace [B][T][n+1][m]
  // QP_EXACT_BOX workspace: two trajectory pairs in one allocation each (X_alt [2][B][T+1][n], U_alt [2][B][T][m]),
  // working set [B][T][m], sweeps per instance [B] (diagnostic, may be null)
  M4Q_P(cplx) X_alt; M4Q_P(double) U_alt; M4Q_P(double) pin_stat; M4Q_P(int) sweep_counts;
};

// discretize_homogeneous for B generator sets (vectorize.py:8-49).  gens: S [B|1][1+m][n][n] (row-major), scaled per
// instance by scales [B][1+m] when given; models: S [B][n][n(1+P)].
struct DiscArgs {
  int B;
  double dt;
  M4Q_P(const void) gens; long gen_stride;
  M4Q_P(const double) scales;
  M4Q_P(void) models;
};

struct PlantArgs {
  int B, kind;
  double dt;
  M4Q_P(const cplx) x; M4Q_P(const double) u;
  M4Q_P(const cplx) op0; long op0_stride;
  M4Q_P(const cplx) ops; long ops_stride;
  M4Q_P(cplx) x_next;
};

// one entry per compiled (dim_x, dim_u, order)
struct ShapeOps {
  int nx, nu, order, np, d;
  int has_tile;                                                     // the tile form of the backward sweep is built for this shape (path 3)
  int has_sg;                                                       // the shared-generator form of the clipped traceless kernel (path 4)
  int plant_only;                                                   // only plant_kernel is built (m4q_shapes.inc): serves m4q_plant_step_batch
  size_t (*mpc_lds_bytes)(int real_path, int exact_qp);
  int (*launch_mpc)(const MpcArgs&, int plant_kind, int real_path, int grid, hipStream_t);
  int (*launch_linearize)(const LinArgs&, hipStream_t);
  int (*launch_qp)(const QpArgs&, hipStream_t);
  int (*launch_plant)(const PlantArgs&, hipStream_t);
  int (*launch_discretize)(const DiscArgs&, int real_path, hipStream_t);
  int (*power_list)(int32_t* out);
  int (*occupancy)(int plant_kind, int real_path, int exact_qp);   // resident workgroups per CU of the fused kernel
};

}  // namespace m4q
