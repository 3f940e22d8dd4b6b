/* m4q.h - C ABI of libm4q_hip.so: batched receding-horizon MPC for quantum state preparation on
 * AMD MI355X (gfx950).  This is the drop-in boundary for the hot path of andgoldschmidt/MPC4quantum.
 *
 * The reference has no FFI: its boundary for this path is three Python call signatures
 * (citations into the reference tree):
 *     mpc4quantum/mpc.py:128-129       mpc(x0, dim_u, order, X_targ, U_targ, clock, experiment, model, Q, R, Qf, ...)
 *     mpc4quantum/optimize.py:12       quad_program(x_init, X_bm, U_bm, Q_ls, R_ls, A_ls, B_ls, Delta_ls, u_prev, sat, du)
 *     mpc4quantum/linearize.py:61-70   WrapModel.get_model_along_traj(xs, us, ts)
 * and the plant call  mpc4quantum/experiment.py:202-212  QExperiment.simulate(x0, ts, us).
 * Each entry point below names the reference interface it replaces.  mpc4quantum_amd/ binds them
 * with ctypes; INTEGRATION.md shows the stub a maintainer of the reference would add.
 *
 * Conventions
 *   - complex numbers are interleaved (re, im) doubles; all arithmetic is fp64;
 *   - every array is C-contiguous with the ENSEMBLE AXIS OUTERMOST and TIME before state:
 *       trajectories  X[b][t][i]   (the reference holds one instance as X[i][t]);
 *   - a model is the reference's DMDc.A block matrix, A[b][i][p*n + k], n x n(1+P) (model.py:95-103,
 *     column layout of linearize.krtimes, linearize.py:80-89); P = number of non-constant control
 *     monomials of the library of the given order (linearize.py:113-120);
 *   - "*_per_instance" = 0 means one array shared by the whole ensemble (no leading b axis);
 *   - host entry points (m4q_*_batch) take caller-owned HOST buffers, copy, launch, copy back;
 *     the session API keeps everything resident in HBM;
 *   - return value: 0 on success, a negative number on failure (-hipError_t for runtime errors,
 *     M4Q_E_* otherwise); m4q_last_error() gives the message.  One host thread per session.
 */
#ifndef M4Q_H
#define M4Q_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#if defined(__GNUC__)
#define M4Q_API __attribute__((visibility("default")))
#else
#define M4Q_API
#endif

#define M4Q_E_UNSUPPORTED (-1001) /* (dim_x, dim_u, order) has no compiled kernel */
#define M4Q_E_BADARG (-1002)
#define M4Q_E_NODEVICE (-1003)
#define M4Q_E_TIMEOUT (-1004) /* the closed-loop launch abandoned itself: its watchdog (M4Q_KERNEL_TIMEOUT_S, default 300 s of device
                                 time) expired before every member had finished; results of that launch are not valid */

/* qp_flags */
#define M4Q_QP_REF_LQR 1 /* reproduce mpc4quantum/lqr.py:14-79 as written (no Delta, no du band) */
#define M4Q_QP_DU_BAND 2 /* also clip the first control to u_prev +- du (optimize.py:29-30) */
#define M4Q_QP_EXACT_BOX 4 /* solve the box-constrained QP of optimize.py:27-54 to optimality (projected Newton on the
                              Riccati factorisation) instead of clipping the unconstrained rollout; not with
                              M4Q_QP_REF_LQR */

/* plant_kind */
#define M4Q_PLANT_NONE 0        /* caller supplies xs[step+1] between m4q_session_run calls */
#define M4Q_PLANT_HAMILTONIAN 1 /* rho+ = U rho U^H, U = expm(-i dt (H0 + sum_k u_k H_k)), d x d operators */
#define M4Q_PLANT_GENERATOR 2   /* x+ = expm(dt (L0 + sum_k u_k L_k)) x, n x n operators */

/* options (m4q_problem.reserved).  By default a session whose model, states, targets and costs are real in a
 * Hermitian operator basis (every vectorised-Liouvillian model and Hermitian state is) runs the closed loop in
 * that basis with real arithmetic - a quarter of the flops of the complex recursion of lqr.py, same results to
 * rounding; anything else runs the general complex path.  This bit forces the complex path. */
#define M4Q_OPT_FORCE_COMPLEX 1
/* A real-path session whose model also leaves the identity component of rho alone (trace-preserving and unital: every
 * vectorised Liouvillian -i[H, .] and its Taylor truncation) and whose initial states and targets share one trace runs the
 * recursion on the d*d - 1 traceless Hermitian coordinates - one dimension fewer, same results to rounding.  This bit keeps
 * such a session on the d*d-coordinate real path. */
#define M4Q_OPT_NO_TRACELESS 2
/* A traceless session with a constant target over the horizon window runs the BACKWARD sweep of the clipped solve on fp64
 * matrix-core tiles (v_mfma_f64_4x4x4_4b_f64: one member per 16-lane block, its operands fetched four horizon indices at a time,
 * csrc/m4q_tile3.h) and the rollout on DPP rows, wherever that form is built: d = 2 and d = 3 (3 and 8 traceless coordinates: at
 * d = 3 the DPP layout leaves half of every row idle) with an order-1 model.  Same results to rounding; config 3 35.5 -> 29.9 ms,
 * config 2 4.05 -> 2.9 ms, config 5's share 117.8 -> 104.4 ms (profiles/r04_ab_experiments.txt).  At d = 4 the DPP rows are full
 * and the tile form does not fit the register file (505 against 71 ms): not built.
 * The same holds for the pinned sweep of an M4Q_QP_EXACT_BOX solve (its time-batched tile form: config 3 exact 208 -> 190 ms).
 * M4Q_OPT_NO_TILE (or M4Q_NO_TILE=1 in the environment) keeps a session on the DPP sweeps (exact mode: the DPP pinned sweep for
 * general targets, which that kernel holds anyway).  M4Q_OPT_TILE is accepted and ignored (round 3's opt-in bit: the tile sweep is
 * no longer an option to ask for). */
#define M4Q_OPT_TILE 4
#define M4Q_OPT_NO_TILE 8
/* A traceless session whose per-member models were built by m4q_session_build_models from ONE set of generators and per-member
 * scales, with an order-1 library - member i's model is [I + dt s_i0 L_0 | dt s_i1 L_1 | ...] (vectorize.py:8-49 at order 1) -
 * runs the clipped solve on the shared generators wherever that form is built (d = 4): the workgroup holds one copy of dt L_k and per
 * member only I + dt s_i0 L_0, the other scales ride on the controls; 12.6 instead of 28.8 KB of LDS per workgroup, which lets d = 4
 * run two wavefronts per SIMD.  Same results to rounding (m4q_session_path: 4).  This bit (or M4Q_NO_SG=1 in the environment) keeps
 * such a session on its per-member models.  Uploaded models (m4q_session_upload) always do. */
#define M4Q_OPT_NO_SG 16

/* exit codes per instance (mpc.py:130,195,202,291): 0 normal, 1 exit_condition (host side),
 * 2 solver gave up (mpc.py:183-197 turns a cvxpy/OSQP warning into this; here: an M4Q_QP_EXACT_BOX solve that stopped at
 *   its iteration cap - the clipped Riccati solve cannot produce it), 3 non-finite objective (mpc.py:200-203; also where
 *   the reference would raise on NaN data: a batched engine cannot raise for one member).
 * A non-zero code ends that member's run at the step where it occurred: steps_done says how many steps are valid. */

typedef struct m4q_problem {
  int32_t dim_x;   /* n = d*d: 4, 9 or 16; also 8 (two reduced qubit states, experiment.py:238-306) with M4Q_PLANT_NONE */
  int32_t dim_u;   /* m */
  int32_t order;   /* control-library order (1 or 2) */
  int32_t horizon; /* T (StepClock.horizon, mpc.py:17) */
  int32_t n_steps; /* StepClock.n_steps (mpc.py:18) */
  int32_t max_iter;   /* SQP iteration cap per MPC step (mpc.py:128, default 100) */
  int32_t warm_start; /* mpc.py:208 */
  int32_t qp_flags;
  int32_t plant_kind;
  int32_t model_per_instance;
  int32_t plant_per_instance;
  int32_t target_per_instance;
  int32_t target_cols; /* columns of X_targ; U_targ has the same count (extra ones unused) */
  int32_t reserved;    /* options: M4Q_OPT_FORCE_COMPLEX | M4Q_OPT_NO_TRACELESS | M4Q_OPT_TILE | M4Q_OPT_NO_TILE | M4Q_OPT_NO_SG */
  int32_t measure_freq; /* StepClock.measure_freq (mpc.py:19,252-267): the plant is measured every measure_freq-th step, the
                           model closes the loop in between; 0 or 1 = every step */
  int32_t reserved2;
  double dt;     /* StepClock.dt */
  double sat;    /* |u| <= sat (optimize.py:43, lqr.py:76) */
  double du;     /* first-control band (optimize.py:29-30); ignored without M4Q_QP_DU_BAND */
  double ls_tol; /* SQP stop: ||alpha dZ|| < ls_tol (mpc.py:224, 1e-4) */
} m4q_problem;

M4Q_API const char* m4q_last_error(void);
M4Q_API const char* m4q_version(void);
/* number of HIP devices visible; < 0 on error */
M4Q_API int m4q_device_count(void);
/* 1 if a kernel exists for this shape */
M4Q_API int m4q_supported(int32_t dim_x, int32_t dim_u, int32_t order);
/* number of non-constant monomials P for (order, dim_u) (linearize.size_of_library - 1) */
M4Q_API int m4q_library_size(int32_t order, int32_t dim_u);
/* exponent table in the reference's order, out[(P+1)*dim_u] (linearize.create_power_list) */
M4Q_API int m4q_power_list(int32_t order, int32_t dim_u, int32_t* out);

/* ---- fine-grained host entry points --------------------------------------------------------- */

/* replaces WrapModel.get_model_along_traj (linearize.py:61-70) for B trajectories.
 * models [B|1][n][n(1+P)] c, X [B][T][n] c (the T linearisation points), U [B][T][m] r
 * -> A_ls [B][T][n][n] c, B_ls [B][T][n][m] c, Delta_ls [B][T][n] c */
M4Q_API int m4q_linearize_batch(int32_t B, int32_t dim_x, int32_t dim_u, int32_t order, int32_t T, const double* models,
                        int32_t model_per_instance, const double* X, const double* U, double* A_ls, double* B_ls,
                        double* Delta_ls);

/* replaces quad_program (optimize.py:12-60 statement; lqr.py:14-79 arithmetic) for B problems.
 * x_init [B][n] c, X_bm [B|1][T+1][n] c, U_bm [B|1][T][m] r, Q_ls [T+1][n][n] c, R_ls [T][m][m] c,
 * A_ls [B][T][n][n] c, B_ls [B][T][n][m] c, Delta_ls [B][T][n] c (NULL = zero),
 * u_prev [B][m] r (NULL = no band) -> X_opt [B][T+1][n] c, U_opt [B][T][m] r, cost [B] r,
 * gains [B][T][n+1][m] c (NULL to skip; gains[b][t][col][k] = Gains[t][k][col] of lqr.py:61).
 * With M4Q_QP_EXACT_BOX the gains are those of the final active-set iteration: a control that is free at the optimum has its
 * feedback row; a control PINNED on a bound at (t, k) has, in its slot, the affine form of its multiplier along the optimal
 * trajectory (row . (x_t - xbar_t) + const = dJ/du_tk, sign opposite to the side it is pinned on) - not the [0 | u - ubar] row
 * a constant control would have. */
M4Q_API int m4q_quad_program_batch(int32_t B, int32_t dim_x, int32_t dim_u, int32_t T, int32_t qp_flags, double sat, double du,
                           const double* x_init, const double* X_bm, const double* U_bm, int32_t bm_per_instance,
                           const double* Q_ls, const double* R_ls, const double* A_ls, const double* B_ls,
                           const double* Delta_ls, const double* u_prev, double* X_opt, double* U_opt, double* cost,
                           double* gains);

/* replaces vectorize.discretize_homogeneous (vectorize.py:8-49) for B operator sets: the Taylor/Dyson expansion of
 * exp(dt (G_0 + sum_k u_k G_k)) to `order`, binned by control monomial.
 * generators [B|1][1+m][n][n] c, scaled per member by scales [B][1+m] r when given (NULL = 1) -> models [B][n][n(1+P)] c */
M4Q_API int m4q_discretize_batch(int32_t B, int32_t dim_x, int32_t dim_u, int32_t order, double dt, const double* generators,
                         int32_t gen_per_instance, const double* scales, double* models);

/* replaces QExperiment.simulate over one held-control step (experiment.py:202-212, mpc.py:256-260).
 * x [B][n] c, u [B][m] r, op0 [B|1][k][k] c, ops [B|1][m][k][k] c with k = d (HAMILTONIAN) or n (GENERATOR)
 * -> x_next [B][n] c */
M4Q_API int m4q_plant_step_batch(int32_t B, int32_t dim_x, int32_t dim_u, int32_t plant_kind, double dt, const double* x,
                         const double* u, const double* op0, const double* ops, int32_t plant_per_instance,
                         double* x_next);

/* replaces the whole mpc() loop body (mpc.py:161-292) for B closed loops, all n_steps in one launch.
 * models [B|1][n][n(1+P)] c, x0 [B][n] c, X_targ [B|1][cols][n] c, U_targ [B|1][cols][m] r,
 * Q, Qf [n][n] c, R [m][m] c, op0/ops as in m4q_plant_step_batch
 * -> xs [B][n_steps+1][n] c, us [B][n_steps][m] r, exit_codes [B], steps_done [B], qp_solves [B][n_steps] */
M4Q_API int m4q_mpc_batch(const m4q_problem* p, int32_t B, const double* models, const double* x0, const double* X_targ,
                  const double* U_targ, const double* Q, const double* R, const double* Qf, const double* op0,
                  const double* ops, double* xs, double* us, int32_t* exit_codes, int32_t* steps_done,
                  int32_t* qp_solves);

/* ---- resident session (inputs stay in HBM; used by bench.py and by step-wise host plants) ---- */
typedef struct m4q_session m4q_session;

enum m4q_field {
  M4Q_F_MODELS = 0,
  M4Q_F_X0 = 1,
  M4Q_F_X_TARG = 2,
  M4Q_F_U_TARG = 3,
  M4Q_F_Q = 4,
  M4Q_F_R = 5,
  M4Q_F_QF = 6,
  M4Q_F_OP0 = 7,
  M4Q_F_OPS = 8,
  M4Q_F_XS = 9,         /* [B][n_steps+1][n] c */
  M4Q_F_US = 10,        /* [B][n_steps][m] r */
  M4Q_F_CODES = 11,     /* [B] i32 */
  M4Q_F_STEPS_DONE = 12,/* [B] i32 */
  M4Q_F_QP_SOLVES = 13, /* [B][n_steps] i32 */
  M4Q_F_X_GUESS = 14,   /* [B][T+1][n] c   SQP guess carried between MPC steps (mpc.py:141,228,271) */
  M4Q_F_U_GUESS = 15,   /* [B][T][m] r     together with XS/US/CODES this is the whole resumable state */
  M4Q_F_COUNT = 16
};

/* device < 0: keep the current device */
M4Q_API int m4q_session_create(const m4q_problem* p, int32_t B, int32_t device, m4q_session** out);
M4Q_API void m4q_session_destroy(m4q_session* s);
/* size in bytes the session expects for a field */
M4Q_API size_t m4q_session_field_bytes(const m4q_session* s, int32_t field);
/* host -> device / device -> host copy of a whole field (synchronous w.r.t. the session stream) */
M4Q_API int m4q_session_upload(m4q_session* s, int32_t field, const void* host, size_t bytes);
M4Q_API int m4q_session_download(m4q_session* s, int32_t field, void* host, size_t bytes);
/* upload/download one MPC step's column of XS for every instance: host [B][n] c (host plants) */
M4Q_API int m4q_session_put_state(m4q_session* s, int32_t step, const void* host);
M4Q_API int m4q_session_get_state(m4q_session* s, int32_t step, void* host);
/* fill M4Q_F_MODELS on the device from continuous-time generators (as m4q_discretize_batch; dt, generators
 * [B|1][1+m][n][n] c and scales [B][1+m] r (or NULL) are host buffers): no model ever crosses PCIe */
M4Q_API int m4q_session_build_models(m4q_session* s, double dt, const double* generators, int32_t gen_per_instance,
                             const double* scales);
/* raw device pointer of a field (for collectives on the results); NULL if unknown */
M4Q_API void* m4q_session_device_ptr(m4q_session* s, int32_t field);
/* use caller-owned DEVICE memory for an output field (XS, US, CODES, STEPS_DONE, QP_SOLVES) */
M4Q_API int m4q_session_bind_output(m4q_session* s, int32_t field, void* device_ptr, size_t bytes);
/* enqueue MPC steps [step_begin, step_end) on the session stream; step_begin == 0 re-initialises the guesses */
M4Q_API int m4q_session_run(m4q_session* s, int32_t step_begin, int32_t step_end);
M4Q_API int m4q_session_sync(m4q_session* s);
/* mark instances finished from the host (exit_condition, mpc.py:289-292): codes [B] i32, nonzero = stop */
M4Q_API int m4q_session_set_codes(m4q_session* s, const int32_t* codes);
/* kernel time of the launches since the last call, from HIP events on the session stream */
M4Q_API int m4q_session_kernel_ms(m4q_session* s, double* total_ms, int32_t* launches);
/* arithmetic path the uploaded problem will run on: 0 complex, 1 real (Hermitian operator basis, d*d coordinates),
 * 2 real on the d*d - 1 traceless coordinates, 3 the same with the sweeps on matrix-core tiles, 4 the traceless clipped solve on
 * shared generators (M4Q_OPT_NO_SG) */
M4Q_API int m4q_session_path(const m4q_session* s);
/* M4Q_QP_EXACT_BOX sessions: counters of the last launch - out[0] QP solves, out[1] Riccati sweeps with pinned
 * controls, out[2] ratio-test steps, out[3..5] solves ended by the KKT test / at working precision / by the iteration
 * cap (the last two leave a feasible, possibly sub-optimal point).  All zero for a clipped-Riccati session. */
M4Q_API int m4q_session_qp_stats(m4q_session* s, int64_t* out6);
/* resident bytes and launch geometry, for reports */
M4Q_API int m4q_session_info(const m4q_session* s, int64_t* hbm_bytes, int32_t* grid, int32_t* lds_bytes);
/* final-state-only results: copy xs[:, n_steps, :] of the session's state history into dst_dev [B][n] c (device memory),
 * enqueued on the session stream behind the launches queued so far */
M4Q_API int m4q_session_copy_final_state(m4q_session* s, void* dst_dev);
/* the watchdog flag (0 / 1, int32) of the launches queued so far into dst_dev (device memory), on the session stream: the
 * status word of a gather buffer, so that the gathering rank sees M4Q_E_TIMEOUT of any rank in the bytes it receives */
M4Q_API int m4q_session_copy_status(m4q_session* s, void* dst_dev);

/* ---- ensemble sharding over the GPUs of one node: one process per GPU, ONE gather of results at the end ----------
 * The reference has no counterpart: mpc4quantum/mpc.py:128-304 runs one closed loop and has no cross-instance data
 * flow, which is exactly why an ensemble shards with no collective on the data path.  The communicator is RCCL
 * (librccl.so, loaded on first use) over xGMI; the unique id travels between the processes by whatever the launcher
 * offers (mpc4quantum_amd/distributed.py: a file keyed by the launcher's MASTER_PORT). */
typedef struct m4q_comm m4q_comm;
#define M4Q_UNIQUE_ID_BYTES 128
#define M4Q_E_COMM (-1005) /* librccl.so missing, or an RCCL call failed (m4q_last_error() has ncclGetErrorString) */

/* rank 0: a fresh RCCL unique id (ncclGetUniqueId), id128 = M4Q_UNIQUE_ID_BYTES caller-owned bytes */
M4Q_API int m4q_comm_unique_id(void* id128);
/* every rank, same id: ncclCommInitRank on `device` (< 0: the current one); collective - returns when all ranks joined */
M4Q_API int m4q_comm_create(int32_t rank, int32_t world, const void* id128, int32_t device, m4q_comm** out);
M4Q_API void m4q_comm_destroy(m4q_comm* c);
/* the one collective of a job: ncclGather of `bytes` bytes from every rank's send_dev into recv_dev on rank dst
 * (world * bytes there; ignored elsewhere).  Enqueued on the communicator's own stream BEHIND everything queued so far
 * on `after`'s stream (after may be NULL), so the kernel that fills send_dev needs no host synchronisation; `slot`
 * (0..7) names the completion event m4q_comm_wait blocks on - two buffers can alternate so that the gather of run k
 * travels under the kernel of run k+1. */
M4Q_API int m4q_comm_gather(m4q_comm* c, m4q_session* after, const void* send_dev, void* recv_dev, size_t bytes, int32_t dst,
                            int32_t slot);
/* host blocks until the gather last enqueued under `slot` has completed (slot < 0: every collective enqueued so far) */
M4Q_API int m4q_comm_wait(m4q_comm* c, int32_t slot);
/* small host-side reductions for reports and fences (op 0 = sum, 1 = max) over n <= 64 doubles, in place;
 * n = 0 is a barrier.  Blocks until done. */
M4Q_API int m4q_comm_allreduce_f64(m4q_comm* c, double* inout_host, int32_t n, int32_t op);

/* device memory owned by the library (gather buffers): zero-filled; device < 0 = the current one */
M4Q_API int m4q_device_alloc(size_t bytes, int32_t device, void** out);
M4Q_API int m4q_device_free(void* dev);
M4Q_API int m4q_device_read(void* host, const void* dev, size_t bytes);  /* blocking device -> host copy */
M4Q_API int m4q_device_write(void* dev, const void* host, size_t bytes); /* blocking host -> device copy */

#ifdef __cplusplus
}
#endif
#endif /* M4Q_H */
