// m4q_args.h - plain argument blocks passed by value to the kernels, shared by the per-shape
// kernel translation units (m4q_kernels.hip) and the host side of the C ABI (m4q_capi.hip).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace m4q {

struct cplx;

// Arrays marked S are complex (cplx) on the general path and double on the real path (models that preserve
// Hermiticity, expressed in the Hermitian operator basis of m4q_mpc.h); the host picks the kernel.
struct MpcArgs {
  int B, T, n_steps, max_iter, warm_start, flags, step_begin, step_end, measure_freq;
  double dt, sat, du, ls_tol;
  const void* models;  long model_stride;   // S [B|1][n][n(1+P)]
  const cplx* x0c;                          // [B][n] complex, as given (becomes xs[:, 0])
  const void* x0s;                          // S [B][n]
  const void* x_targ;  long xt_stride;      // S [B|1][cols][n]
  const double* u_targ; long ut_stride;     // [B|1][cols][m]
  const void* Q; const void* Qf; const void* R;            // S
  const double* Cq; const double* Cqf; const double* Cr;   // line-search blocks (mpc.py:103-116)
  const double* Wls;                        // [2n + 2n + 2m] diagonals of those blocks, or nullptr if one is not diagonal
  const cplx* op0; long op0_stride;         // plant operators
  const cplx* ops; long ops_stride;
  cplx* xs; double* us; int* codes; int* steps_done; int* qp_solves;
  cplx* Xg; double* Ug;                     // per-instance SQP guess  [B][T+1][n] complex, [B][T][m] (resumable state)
  // per resident row (grid*4 of them): working guess followed by the QP solution (S [2][rows][T+1][n], [2][rows][T][m]), gains (S)
  void* ws_Xg; double* ws_Ug; void* ws_gains;
  int* queue;                               // 64 B zeroed before every launch: [0] next work item to hand out; [1] set by the watchdog;
                                            // as u64 [1..3]: exact-QP counters (solves, Newton iterations, arc trials)
  int* head_done;                           // [B] set when an instance's head item (steps < 2) has been published; zeroed likewise
  unsigned long long deadline_ticks;        // watchdog: the launch abandons itself (queue[1] = 1) once s_memrealtime (100 MHz) has
                                            // advanced this far since the wavefront started; every wavefront reaches this exit
};

struct LinArgs {
  int B, T;
  const cplx* models; long model_stride;
  const cplx* X; const double* U;           // [B][T][n], [B][T][m]
  cplx* A_ls; cplx* B_ls; cplx* D_ls;
};

struct QpArgs {
  int B, T, flags;
  double sat, du;
  const cplx* x_init;
  const cplx* X_bm; long xbm_stride;
  const double* U_bm; long ubm_stride;
  const cplx* Q_ls; const cplx* R_ls;       // [T+1][n][n], [T][m][m]
  const cplx* A_ls; const cplx* B_ls; const cplx* D_ls;
  const double* u_prev;
  cplx* X_opt; double* U_opt; double* cost; cplx* gains;   // gains: caller buffer or workspace [B][T][n+1][m]
  // QP_EXACT_BOX workspace: two trajectory pairs in one allocation each (X_alt [2][B][T+1][n], U_alt [2][B][T][m]),
  // working set [B][T][m], sweeps per instance [B] (diagnostic, may be null)
  cplx* X_alt; double* U_alt; double* pin_stat; int* sweep_counts;
};

// discretize_homogeneous for B generator sets (vectorize.py:8-49).  gens: S [B|1][1+m][n][n] (row-major), scaled per
// instance by scales [B][1+m] when given; models: S [B][n][n(1+P)].
struct DiscArgs {
  int B;
  double dt;
  const void* gens; long gen_stride;
  const double* scales;
  void* models;
};

struct PlantArgs {
  int B, kind;
  double dt;
  const cplx* x; const double* u;
  const cplx* op0; long op0_stride;
  const cplx* ops; long ops_stride;
  cplx* x_next;
};

// one entry per compiled (dim_x, dim_u, order)
struct ShapeOps {
  int nx, nu, order, np, d;
  size_t (*mpc_lds_bytes)(int real_path);
  int (*launch_mpc)(const MpcArgs&, int plant_kind, int real_path, int grid, hipStream_t);
  int (*launch_linearize)(const LinArgs&, hipStream_t);
  int (*launch_qp)(const QpArgs&, hipStream_t);
  int (*launch_plant)(const PlantArgs&, hipStream_t);
  int (*launch_discretize)(const DiscArgs&, int real_path, hipStream_t);
  int (*power_list)(int32_t* out);
  int (*occupancy)(int plant_kind, int real_path, int exact_qp);   // resident workgroups per CU of the fused kernel
};

}  // namespace m4q
