// m4q_capi.hip - host side of the C ABI declared in include/m4q.h.
// Owns device memory, streams and events; dispatches to the per-shape kernel objects.
#include <hip/hip_runtime.h>
#include <dlfcn.h>
#include <rccl/rccl.h>      // types and prototypes only: librccl.so is loaded with dlopen on first use (a CPU-only import never needs it)

#include <algorithm>
#include <cmath>
#include <complex>
#include <cstdlib>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <mutex>
#include <string>
#include <vector>

#include "../../include/m4q.h"
#include "m4q_args.h"

namespace m4q {
struct cplx { double re, im; };
}
using m4q::cplx;

// per-shape registration functions (m4q_kernels.hip compiled once per shape)
#define M4Q_SHAPE(nx, nu, ord) extern "C" const m4q::ShapeOps* m4q_shape_##nx##_##nu##_##ord();
#include "m4q_shapes.inc"
#undef M4Q_SHAPE

extern "C" __attribute__((visibility("hidden"))) void dim_d_anchor() {}      // an address inside this library, for dladdr

namespace {

thread_local std::string g_err;

int fail(int code, const char* fmt, ...) {
  char buf[512];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof(buf), fmt, ap);
  va_end(ap);
  g_err = buf;
  return code;
}

#define HIP_TRY(expr)                                                                          \
  do {                                                                                         \
    hipError_t e_ = (expr);                                                                    \
    if (e_ != hipSuccess) return fail(-(int)e_, "%s: %s", #expr, hipGetErrorString(e_));       \
  } while (0)

// inside m4q_session_create, once the session object exists: a failing HIP call must not leak it
#define HIP_TRY_OWNED(sess, expr)                                                              \
  do {                                                                                         \
    hipError_t e_ = (expr);                                                                    \
    if (e_ != hipSuccess) {                                                                    \
      m4q_session_destroy(sess);                                                               \
      return fail(-(int)e_, "%s: %s", #expr, hipGetErrorString(e_));                           \
    }                                                                                          \
  } while (0)

// plant_ok: plant-only shapes (m4q_shapes.inc) count too - for m4q_plant_step_batch; every other entry point needs the full set
const m4q::ShapeOps* find_shape(int nx, int nu, int order, bool plant_ok = false) {
  static const m4q::ShapeOps* table[] = {
#define M4Q_SHAPE(nx, nu, ord) m4q_shape_##nx##_##nu##_##ord(),
#include "m4q_shapes.inc"
#undef M4Q_SHAPE
  };
  for (const m4q::ShapeOps* s : table)
    if (s->nx == nx && s->nu == nu && s->order == order && (plant_ok || !s->plant_only)) return s;
  return nullptr;
}

// any compiled order for (nx, nu): the QP and plant kernels do not depend on the library order
const m4q::ShapeOps* find_shape_any_order(int nx, int nu, bool plant_ok = false) {
  for (int ord = 1; ord <= 3; ++ord)
    if (const m4q::ShapeOps* s = find_shape(nx, nu, ord, plant_ok)) return s;
  return nullptr;
}

// The closed-loop kernels with the GENERATOR plant live in a second library, libm4q_hip_gen.so, next to this one (M4Q_GEN_LIB
// overrides the path): loaded the first time a session asks for that plant, never otherwise.  Returns the ops whose launch_mpc /
// occupancy / mpc_lds_bytes serve such a session, or nullptr with the reason in `why`.
const m4q::ShapeOps* gen_shape(int nx, int nu, int order, std::string& why) {
  static std::mutex mu;
  static void* handle = nullptr;
  static std::string load_error;
  std::lock_guard<std::mutex> lock(mu);
  if (!handle && load_error.empty()) {
    std::string path;
    if (const char* env = std::getenv("M4Q_GEN_LIB")) {
      path = env;
    } else {
      Dl_info info{};
      if (dladdr(reinterpret_cast<const void*>(&dim_d_anchor), &info) && info.dli_fname) {
        path = info.dli_fname;
        const size_t slash = path.find_last_of('/');
        path = (slash == std::string::npos ? std::string() : path.substr(0, slash + 1)) + "libm4q_hip_gen.so";
      }
    }
    handle = path.empty() ? nullptr : dlopen(path.c_str(), RTLD_NOW | RTLD_LOCAL);
    if (!handle) {
      const char* err = path.empty() ? nullptr : dlerror();          // (dlerror() clears itself: ask once)
      load_error = "the generator-plant kernels are in libm4q_hip_gen.so, which did not load (" + path + "): " + (err ? err : "?");
    }
  }
  if (!handle) { why = load_error; return nullptr; }
  char name[64];
  snprintf(name, sizeof(name), "m4q_shapeg_%d_%d_%d", nx, nu, order);
  typedef const m4q::ShapeOps* (*fn_t)();
  fn_t fn = reinterpret_cast<fn_t>(dlsym(handle, name));
  if (!fn) { why = std::string("libm4q_hip_gen.so has no ") + name; return nullptr; }
  return fn();
}

int dim_d(int nx) { return nx == 4 ? 2 : nx == 9 ? 3 : nx == 16 ? 4 : 0; }

struct DevBuf {
  void* p = nullptr;
  size_t bytes = 0;
  bool owned = true;
  int alloc(size_t n) {
    release();
    bytes = n;
    owned = true;
    if (n == 0) return 0;
    hipError_t e = hipMalloc(&p, n);
    if (e != hipSuccess) { p = nullptr; return fail(-(int)e, "hipMalloc(%zu): %s", n, hipGetErrorString(e)); }
    return 0;
  }
  void release() {
    if (p && owned) (void)hipFree(p);
    p = nullptr;
    bytes = 0;
  }
  ~DevBuf() { release(); }
};

// symmetrised real-ified cost block of iqp_line_search (mpc.py:92-93,103-104,112-116)
void ls_block(const double* M, int k, std::vector<double>& out) {
  const int s = 2 * k;
  std::vector<double> c((size_t)s * s);
  for (int i = 0; i < k; ++i)
    for (int j = 0; j < k; ++j) {
      const double re = M[2 * (i * k + j)], im = M[2 * (i * k + j) + 1];
      c[(size_t)i * s + j] = re;
      c[(size_t)i * s + (j + k)] = -im;
      c[(size_t)(i + k) * s + j] = im;
      c[(size_t)(i + k) * s + (j + k)] = re;
    }
  out.resize((size_t)s * s);
  for (int i = 0; i < s; ++i)
    for (int j = 0; j < s; ++j) out[(size_t)i * s + j] = 0.5 * (c[(size_t)i * s + j] + c[(size_t)j * s + i]);
}


// ---- Hermitian operator basis of the real path (same convention as csrc/m4q_mpc.h) ----------------
// slot c = a*d + b:  a == b: rho_aa;  a < b: sqrt2 Re rho_ab;  a > b: sqrt2 Im rho_ab.
// Column c of the unitary W (x = W r) has at most two entries; (W^H v)_c and (M W)_{.,c} cost O(1).
struct HermBasis {
  int d, n;
  explicit HermBasis(int d_) : d(d_), n(d_ * d_) {}
  // out = W^H v  (complex n-vector, stride 1)
  void lift_vec(const std::complex<double>* v, std::complex<double>* out) const {
    const double rs = 0.70710678118654752440;
    const std::complex<double> I(0, 1);
    for (int a = 0; a < d; ++a)
      for (int b = 0; b < d; ++b) {
        const int c = a * d + b, ct = b * d + a;
        if (a == b) out[c] = v[c];
        else if (a < b) out[c] = (v[c] + v[ct]) * rs;            // conj(1/sqrt2) (x_ab + x_ba)
        else out[c] = (v[c] - v[ct]) * (-I * rs);                // conj(+i/sqrt2) x_ab + conj(-i/sqrt2) x_ba
      }
  }
  // M (n x n, row-major, leading dimension ld) -> W^H M W, written to out (n x n, leading dimension ldo)
  void lift_mat(const std::complex<double>* M, long ld, std::complex<double>* out, long ldo) const {
    const double rs = 0.70710678118654752440;
    const std::complex<double> I(0, 1);
    std::vector<std::complex<double>> Y((size_t)n * n);
    for (int i = 0; i < n; ++i)
      for (int a = 0; a < d; ++a)
        for (int b = 0; b < d; ++b) {
          const int c = a * d + b, ct = b * d + a;
          const std::complex<double> m1 = M[i * ld + c], m2 = M[i * ld + ct];
          if (a == b) Y[(size_t)i * n + c] = m1;
          else if (a < b) Y[(size_t)i * n + c] = (m1 + m2) * rs;
          else Y[(size_t)i * n + c] = (m1 - m2) * (I * rs);      // W[(a,b),c] = +i/sqrt2, W[(b,a),c] = -i/sqrt2
        }
    std::vector<std::complex<double>> col(n), lifted(n);
    for (int j = 0; j < n; ++j) {
      for (int i = 0; i < n; ++i) col[i] = Y[(size_t)i * n + j];
      lift_vec(col.data(), lifted.data());
      for (int i = 0; i < n; ++i) out[i * ldo + j] = lifted[i];
    }
  }
};

// real part of a lifted array + the size of what was dropped, relative to the array's scale
struct LiftStat {
  double max_im = 0.0, max_abs = 0.0;
  void see(std::complex<double> v) {
    max_im = std::max(max_im, std::fabs(v.imag()));
    max_abs = std::max(max_abs, std::abs(v));
  }
  bool real_enough() const { return max_im <= 1e-13 * std::max(1.0, max_abs); }
};

// ---- traceless coordinates (path 2; csrc/m4q_mpc.h): the diagonal slots (a, a) of the Hermitian basis rotated by the orthogonal
// O[a][0] = 1/sqrt(d), O[a][l] = 1/sqrt(l(l+1)) (a < l), -l/sqrt(l(l+1)) (a == l), 0 (a > l); slot (0, 0) becomes the trace
// coordinate and is dropped when the model leaves it alone.  Works on the REAL arrays the Hermitian lift produced.
struct Traceless {
  int d, n;
  std::vector<double> O;                 // n x n: identity off the diagonal slots
  explicit Traceless(int d_) : d(d_), n(d_ * d_), O((size_t)d_ * d_ * d_ * d_, 0.0) {
    for (int c = 0; c < n; ++c) O[(size_t)c * n + c] = 1.0;
    for (int a = 0; a < d; ++a)
      for (int l = 0; l < d; ++l) {
        double v;
        if (l == 0) v = 1.0 / std::sqrt((double)d);
        else v = a < l ? 1.0 / std::sqrt((double)l * (l + 1)) : (a == l ? -(double)l / std::sqrt((double)l * (l + 1)) : 0.0);
        O[(size_t)(a * d + a) * n + (l * d + l)] = v;
      }
  }
  // r (n) -> O^T r: out[0] = trace coordinate, out[1..n) = traceless coordinates
  void vec(const double* r, double* out) const {
    for (int c = 0; c < n; ++c) {
      double acc = 0.0;
      for (int k = 0; k < n; ++k) acc += O[(size_t)k * n + c] * r[k];
      out[c] = acc;
    }
  }
  // M (n x n, leading dimension ld) -> O^T M O (n x n, dense, into out)
  void mat(const double* M, long ld, double* out) const {
    std::vector<double> Y((size_t)n * n, 0.0);
    for (int i = 0; i < n; ++i)
      for (int k = 0; k < n; ++k) {
        const double m = M[i * ld + k];
        if (m != 0.0)
          for (int c = 0; c < n; ++c) Y[(size_t)i * n + c] += m * O[(size_t)k * n + c];
      }
    for (int r = 0; r < n; ++r)
      for (int c = 0; c < n; ++c) {
        double acc = 0.0;
        for (int i = 0; i < n; ++i) acc += O[(size_t)i * n + r] * Y[(size_t)i * n + c];
        out[(size_t)r * n + c] = acc;
      }
  }
};

// how well the trace coordinate decouples: largest entry of row 0 / column 0 off what a decoupled block must hold
struct DecoupleStat {
  double worst = 0.0, scale = 0.0;
  void see_block(const double* M, int n, bool identity_block) {     // M = O^T block O
    for (int k = 0; k < n; ++k) {
      const double want = (k == 0 && identity_block) ? 1.0 : 0.0;
      worst = std::max(worst, std::fabs(M[k] - want));                       // row 0
      worst = std::max(worst, std::fabs(M[(size_t)k * n] - want));           // column 0
    }
    for (int e = 0; e < n * n; ++e) scale = std::max(scale, std::fabs(M[e]));
  }
  bool ok() const { return worst <= 1e-12 * std::max(1.0, scale); }
};

}  // namespace

struct m4q_session {
  m4q_problem prob{};
  int B = 0;
  int device = 0;
  const m4q::ShapeOps* shape = nullptr;
  const m4q::ShapeOps* mpc_ops = nullptr;      // launch_mpc / occupancy / mpc_lds_bytes: `shape`, or libm4q_hip_gen.so's for the generator plant
  hipStream_t stream = nullptr;
  hipEvent_t ev0 = nullptr, ev1 = nullptr;
  int grid = 1;                 // resident workgroups of the launch (paths 0-3); the per-row workspace is sized for max(grid, grid_sg)
  int grid_sg = 0;              // ... of the shared-generator kernel (path 4: two wavefronts per SIMD at d = 4), 0 when not available
  // shared-generator form (m4q_session_build_models with ONE generator set, order 1, traceless blocks): dt L_k on the traceless
  // coordinates [1 + m][n-1][n-1] and the members' scales [B][1 + m]; the kernel of path 4 reads these instead of the models
  DevBuf sg_gens, sg_scales;
  bool sg_ok = false, no_sg = false;
  DevBuf f[M4Q_F_COUNT];
  DevBuf Cq, Cqf, Cr, Wls, wsXg, wsUg, wsG, queue, head_done;
  bool no_tile = false;
  size_t fbytes[M4Q_F_COUNT]{};
  std::vector<std::pair<hipEvent_t, hipEvent_t>> pending;   // (start, stop) of launches not yet read by kernel_ms
  double folded_ms = 0.0;                                    // launches already completed and folded out of `pending`
  int folded_n = 0;
  double ms_total = 0.0;
  int launches = 0;
  bool costs_dirty = true;
  bool ls_diag = false;
  // real path: inputs lifted to the Hermitian operator basis at upload time (doubles), and whether each was real there
  DevBuf r_models, r_x0, r_xtarg, r_Q, r_Qf, r_R;
  bool herm_ok[M4Q_F_COUNT] = {};
  // traceless path: the same inputs on the n - 1 traceless coordinates, whether the trace coordinate decouples from each, and
  // the range of trace coordinates seen in x0 / X_targ (state and target must share ONE trace for the cost to restrict)
  DevBuf t_models, t_x0, t_xtarg, t_Q, t_Qf;
  bool tl_ok[M4Q_F_COUNT] = {};
  double tau_x0[2] = {0, 0}, tau_targ[2] = {0, 0};
  bool no_traceless = false;
  bool force_complex = false;
  bool launched = false;        // a closed-loop launch has been enqueued since the watchdog flag was last read
  bool targ_const = false;      // every column of X_targ equals the first (per member, if per-member): xbar does not depend on t
  bool use_real(bool diag) const {
    return !force_complex && diag && herm_ok[M4Q_F_MODELS] && herm_ok[M4Q_F_X0] && herm_ok[M4Q_F_X_TARG] &&
           herm_ok[M4Q_F_Q] && herm_ok[M4Q_F_QF] && herm_ok[M4Q_F_R];
  }
  bool use_real() const { return use_real(ls_diag); }
  bool use_traceless(bool diag) const {
    if (!use_real(diag) || no_traceless || !tl_ok[M4Q_F_MODELS] || !tl_ok[M4Q_F_X0] || !tl_ok[M4Q_F_X_TARG] || !tl_ok[M4Q_F_Q] ||
        !tl_ok[M4Q_F_QF])
      return false;
    const double lo = std::min(tau_x0[0], tau_targ[0]), hi = std::max(tau_x0[1], tau_targ[1]);
    return hi - lo <= 1e-12 * std::max(1.0, std::fabs(hi));
  }
  // 0 complex, 1 real (Hermitian basis), 2 real traceless.  diag: the line-search blocks of the costs are diagonal (known after
  // the first run; m4q_session_path answers as if they were before that)
  //                         3 traceless with the backward sweep of the clipped solve / the pinned sweep of the exact solve on
  //                           matrix-core tiles (constant targets only)
  //                         4 traceless clipped solve on shared generators (models built by m4q_session_build_models from one
  //                           generator set; where that kernel is built: d = 4)
  int path(bool diag) const {
    if (!use_traceless(diag)) return use_real(diag) ? 1 : 0;
    if (sg_ok && !no_sg && grid_sg > 0 && !(prob.qp_flags & M4Q_QP_EXACT_BOX) && !(prob.qp_flags & M4Q_QP_REF_LQR)) return 4;
    return (!no_tile && targ_const) ? 3 : 2;
  }
  int path() const { return path(ls_diag); }
  std::vector<double> hQ, hQf, hR;
};

extern "C" {

const char* m4q_last_error(void) { return g_err.c_str(); }
const char* m4q_version(void) { return "m4q-hip 0.1 (gfx950)"; }

int m4q_device_count(void) {
  int n = 0;
  hipError_t e = hipGetDeviceCount(&n);
  if (e != hipSuccess) return fail(-(int)e, "hipGetDeviceCount: %s", hipGetErrorString(e));
  return n;
}

int m4q_supported(int32_t dim_x, int32_t dim_u, int32_t order) { return find_shape(dim_x, dim_u, order) ? 1 : 0; }

int m4q_library_size(int32_t order, int32_t dim_u) {
  if (order < 0 || dim_u < 1) return fail(M4Q_E_BADARG, "bad (order, dim_u)");
  // C(order + m, m) - 1
  long r = 1;
  for (int i = 1; i <= dim_u; ++i) r = r * (order + i) / i;
  return (int)r - 1;
}

int m4q_power_list(int32_t order, int32_t dim_u, int32_t* out) {
  for (int nx : {4, 9, 16})
    if (const m4q::ShapeOps* s = find_shape(nx, dim_u, order)) return s->power_list(out);
  return fail(M4Q_E_UNSUPPORTED, "no compiled kernel with (order=%d, dim_u=%d)", order, dim_u);
}

// ------------------------------------------------------------------------------------------
// session
// ------------------------------------------------------------------------------------------
int m4q_session_create(const m4q_problem* p, int32_t B, int32_t device, m4q_session** out) {
  if (!p || !out || B <= 0) return fail(M4Q_E_BADARG, "m4q_session_create: bad argument");
  const m4q::ShapeOps* sh = find_shape(p->dim_x, p->dim_u, p->order);
  if (!sh) return fail(M4Q_E_UNSUPPORTED, "no kernel for dim_x=%d dim_u=%d order=%d", p->dim_x, p->dim_u, p->order);
  if (p->horizon < 1 || p->n_steps < 1 || p->target_cols < p->horizon + 1 + (p->n_steps > 1 ? p->n_steps - 2 : 0))
    return fail(M4Q_E_BADARG, "horizon/n_steps/target_cols inconsistent (need target_cols >= n_steps + horizon - 1)");
  if (!(p->sat > 0)) return fail(M4Q_E_BADARG, "sat must be positive (the reference crashes on sat=None, mpc.py Q5)");
  if (p->qp_flags & ~(M4Q_QP_REF_LQR | M4Q_QP_DU_BAND | M4Q_QP_EXACT_BOX))
    return fail(M4Q_E_BADARG, "qp_flags has bits outside M4Q_QP_REF_LQR | M4Q_QP_DU_BAND | M4Q_QP_EXACT_BOX (0x%x)", p->qp_flags);
  if ((p->qp_flags & M4Q_QP_EXACT_BOX) && (p->qp_flags & M4Q_QP_REF_LQR))
    return fail(M4Q_E_BADARG, "M4Q_QP_EXACT_BOX cannot be combined with M4Q_QP_REF_LQR");
  {
    // per-instance targets / plant operators are reached through 32-bit byte offsets from one base
    const double lim = 4294967296.0;
    const double kk = p->plant_kind == M4Q_PLANT_GENERATOR ? p->dim_x : dim_d(p->dim_x);
    if (p->target_per_instance && (double)B * p->target_cols * p->dim_x * 16.0 >= lim)
      return fail(M4Q_E_BADARG, "per-instance targets must stay below 4 GiB in total");
    if (p->plant_per_instance && (double)B * p->dim_u * kk * kk * 16.0 >= lim)
      return fail(M4Q_E_BADARG, "per-instance plant operators must stay below 4 GiB in total");
  }
  int ndev = 0;
  hipError_t e = hipGetDeviceCount(&ndev);
  if (e != hipSuccess || ndev == 0) return fail(M4Q_E_NODEVICE, "no HIP device: %s", hipGetErrorString(e));
  if (device >= 0) HIP_TRY(hipSetDevice(device));
  m4q_session* s = new m4q_session();
  s->prob = *p;
  s->B = B;
  s->shape = sh;
  s->mpc_ops = sh;
  if (p->plant_kind == M4Q_PLANT_GENERATOR) {
    std::string why;
    s->mpc_ops = gen_shape(p->dim_x, p->dim_u, p->order, why);
    if (!s->mpc_ops) {
      delete s;
      return fail(M4Q_E_UNSUPPORTED, "%s", why.c_str());
    }
  }
  s->force_complex = (p->reserved & 1) != 0 || std::getenv("M4Q_FORCE_COMPLEX") != nullptr || sh->d * sh->d != p->dim_x;
  // (M4Q_QP_REF_LQR builds its cost terms on xbar itself, lqr.py:54-58: the trace coordinate of the target does not drop out)
  s->no_traceless = (p->reserved & M4Q_OPT_NO_TRACELESS) != 0 || std::getenv("M4Q_NO_TRACELESS") != nullptr ||
                    (p->qp_flags & M4Q_QP_REF_LQR) != 0;
  if (sh->d * sh->d != p->dim_x && p->plant_kind != M4Q_PLANT_NONE) {
    delete s;
    return fail(M4Q_E_UNSUPPORTED, "dim_x=%d is not a vectorised density matrix: no device plant (use M4Q_PLANT_NONE)", p->dim_x);
  }
  HIP_TRY_OWNED(s, hipGetDevice(&s->device));
  HIP_TRY_OWNED(s, hipStreamCreateWithFlags(&s->stream, hipStreamNonBlocking));
  HIP_TRY_OWNED(s, hipEventCreate(&s->ev0));
  HIP_TRY_OWNED(s, hipEventCreate(&s->ev1));
  const size_t n = p->dim_x, m = p->dim_u, P = sh->np, T = p->horizon, ns = p->n_steps, cols = p->target_cols;
  const size_t k = p->plant_kind == M4Q_PLANT_GENERATOR ? n : (size_t)sh->d;
  const size_t C = 16;
  size_t* fb = s->fbytes;
  fb[M4Q_F_MODELS] = (p->model_per_instance ? B : 1) * n * n * (1 + P) * C;
  fb[M4Q_F_X0] = (size_t)B * n * C;
  fb[M4Q_F_X_TARG] = (p->target_per_instance ? B : 1) * cols * n * C;
  fb[M4Q_F_U_TARG] = (p->target_per_instance ? B : 1) * cols * m * 8;
  fb[M4Q_F_Q] = n * n * C;
  fb[M4Q_F_QF] = n * n * C;
  fb[M4Q_F_R] = m * m * C;
  fb[M4Q_F_OP0] = p->plant_kind == M4Q_PLANT_NONE ? 0 : (p->plant_per_instance ? B : 1) * k * k * C;
  fb[M4Q_F_OPS] = p->plant_kind == M4Q_PLANT_NONE ? 0 : (p->plant_per_instance ? B : 1) * m * k * k * C;
  fb[M4Q_F_XS] = (size_t)B * (ns + 1) * n * C;
  fb[M4Q_F_US] = (size_t)B * ns * m * 8;
  fb[M4Q_F_CODES] = (size_t)B * 4;
  fb[M4Q_F_STEPS_DONE] = (size_t)B * 4;
  fb[M4Q_F_QP_SOLVES] = (size_t)B * ns * 4;
  fb[M4Q_F_X_GUESS] = (size_t)B * (T + 1) * n * C;
  fb[M4Q_F_U_GUESS] = (size_t)B * T * m * 8;
  int rc = 0;
  for (int i = 0; i < M4Q_F_COUNT && !rc; ++i) rc = s->f[i].alloc(fb[i]);
  // resident grid: as many workgroups as the device holds at once (persistent, quad-strided)
  hipDeviceProp_t prop;
  HIP_TRY_OWNED(s, hipGetDeviceProperties(&prop, s->device));
  // the grid (and the per-row workspace) is sized for whichever path keeps more workgroups resident
  const int exact = (p->qp_flags & M4Q_QP_EXACT_BOX) ? 1 : 0;
  const m4q::ShapeOps* mo = s->mpc_ops;
  int per_cu = std::max(mo->occupancy(p->plant_kind, 0, exact), s->force_complex ? 0 : mo->occupancy(p->plant_kind, 1, exact));
  // the tile form of the backward sweep is what a traceless session with a constant target runs wherever it is built (d = 2, 3 with an
  // order-1 library: include/m4q.h, M4Q_OPT_NO_TILE)
  s->no_tile = !sh->has_tile || (p->reserved & M4Q_OPT_NO_TILE) != 0 || std::getenv("M4Q_NO_TILE") != nullptr;
  if (!s->force_complex && !s->no_traceless) per_cu = std::max(per_cu, mo->occupancy(p->plant_kind, 2, exact));
  if (!s->force_complex && !s->no_traceless && !s->no_tile && !exact) per_cu = std::max(per_cu, mo->occupancy(p->plant_kind, 3, 0));
  if (per_cu < 1) per_cu = 1;
  if (const char* cap = std::getenv("M4Q_WGS_PER_CU")) {       // tuning experiments: fewer resident workgroups per CU
    const int v = std::atoi(cap);
    if (v >= 1 && v < per_cu) per_cu = v;
  }
  const int nquads = (B + 3) / 4;
  long resident = (long)per_cu * prop.multiProcessorCount;
  s->grid = (int)(nquads < resident ? nquads : resident);
  // the shared-generator kernel (path 4) keeps more workgroups resident than the per-member-model kernels of its shape: its own grid
  s->no_sg = !sh->has_sg || (p->reserved & M4Q_OPT_NO_SG) != 0 || std::getenv("M4Q_NO_SG") != nullptr || exact ||
             s->force_complex || s->no_traceless || p->order != 1;
  if (!s->no_sg) {
    int pc = mo->occupancy(p->plant_kind, 4, 0);
    if (const char* cap = std::getenv("M4Q_WGS_PER_CU")) { const int v = std::atoi(cap); if (v >= 1 && v < pc) pc = v; }
    if (pc >= 1) {
      const long res_sg = (long)pc * prop.multiProcessorCount;
      s->grid_sg = (int)(nquads < res_sg ? nquads : res_sg);
    }
  }
  const size_t rows = (size_t)std::max(s->grid, s->grid_sg) * 4;
  // [Xg rows][Xo rows] and [Ug rows][Uo rows]; the exact QP adds [Xalt rows] and [Ualt rows][working-set rows]
  if (!rc) rc = s->wsXg.alloc((exact ? 3 : 2) * rows * (T + 1) * n * C);
  if (!rc) rc = s->wsUg.alloc((exact ? 4 : 2) * rows * T * m * 8);
  if (!rc) rc = s->queue.alloc(192);       // 64 B of queue / watchdog / solver counters + 16 u64 phase clocks (dev builds)
  if (!rc) rc = s->head_done.alloc((size_t)B * 4);
  if (!rc) rc = s->wsG.alloc(rows * T * (n + 1) * m * C);
  if (!rc) rc = s->Cq.alloc(4 * n * n * 8);
  if (!rc) rc = s->Cqf.alloc(4 * n * n * 8);
  if (!rc) rc = s->Cr.alloc(4 * m * m * 8);
  if (!rc) rc = s->Wls.alloc((4 * n + 2 * m) * 8);
  if (rc) { m4q_session_destroy(s); return rc; }
  for (int i : {M4Q_F_XS, M4Q_F_US, M4Q_F_CODES, M4Q_F_STEPS_DONE, M4Q_F_QP_SOLVES})
    HIP_TRY_OWNED(s, hipMemsetAsync(s->f[i].p, 0, fb[i], s->stream));
  HIP_TRY_OWNED(s, hipStreamSynchronize(s->stream));
  *out = s;
  return 0;
}

void m4q_session_destroy(m4q_session* s) {
  if (!s) return;
  for (auto& pr : s->pending) { (void)hipEventDestroy(pr.first); (void)hipEventDestroy(pr.second); }
  if (s->ev0) (void)hipEventDestroy(s->ev0);
  if (s->ev1) (void)hipEventDestroy(s->ev1);
  if (s->stream) (void)hipStreamDestroy(s->stream);
  delete s;
}

size_t m4q_session_field_bytes(const m4q_session* s, int32_t field) {
  if (!s || field < 0 || field >= M4Q_F_COUNT) return 0;
  return s->fbytes[field];
}

// After the stream has drained: did the last closed-loop launch leave through its watchdog?
static int check_watchdog(m4q_session* s) {
  if (!s->launched) return 0;
  s->launched = false;
  int flag = 0;
  HIP_TRY(hipMemcpy(&flag, (const char*)s->queue.p + 4, 4, hipMemcpyDeviceToHost));
  if (flag) HIP_TRY(hipMemset((char*)s->queue.p + 4, 0, 4));
  if (flag)
    return fail(M4Q_E_TIMEOUT, "the closed-loop kernel abandoned the launch: its watchdog expired (M4Q_KERNEL_TIMEOUT_S, default 300 s "
                               "of device time) before every ensemble member had finished; this launch's results are not valid");
  return 0;
}

// lift one uploaded input to the Hermitian basis: keeps the real part on the device (doubles), remembers whether the
// imaginary part was negligible.  vec_len = n for vectors; matrices are n x n blocks laid side by side (nblk per row set).
static int lift_upload(m4q_session* s, int32_t field, const std::complex<double>* src, size_t count_items, bool matrix, int nblk,
                       DevBuf& dst) {
  const HermBasis hb(s->shape->d);
  const int n = hb.n;
  LiftStat st;
  std::vector<double> out;
  if (!matrix) {
    out.resize(count_items * n);
    std::vector<std::complex<double>> tmp(n);
    for (size_t it = 0; it < count_items; ++it) {
      hb.lift_vec(src + it * n, tmp.data());
      for (int i = 0; i < n; ++i) { st.see(tmp[i]); out[it * n + i] = tmp[i].real(); }
    }
  } else {
    // count_items matrices of shape n x (n * nblk), row-major: block p occupies columns [p*n, (p+1)*n)
    const long ld = (long)n * nblk;
    out.resize(count_items * n * ld);
    std::vector<std::complex<double>> tmp((size_t)n * n);
    for (size_t it = 0; it < count_items; ++it)
      for (int p = 0; p < nblk; ++p) {
        hb.lift_mat(src + it * n * ld + (long)p * n, ld, tmp.data(), n);
        for (int i = 0; i < n; ++i)
          for (int j = 0; j < n; ++j) {
            st.see(tmp[(size_t)i * n + j]);
            out[it * n * ld + i * ld + (long)p * n + j] = tmp[(size_t)i * n + j].real();
          }
      }
  }
  s->herm_ok[field] = st.real_enough();
  int rc = dst.alloc(out.size() * sizeof(double));
  if (rc) return rc;
  HIP_TRY(hipMemcpy(dst.p, out.data(), out.size() * sizeof(double), hipMemcpyHostToDevice));
  // ... and on the traceless coordinates
  s->tl_ok[field] = false;
  if (!s->herm_ok[field] || s->no_traceless) return 0;
  const Traceless tl(s->shape->d);
  const int ns = n - 1;
  std::vector<double> tout, tmp((size_t)n * n);
  DevBuf* tdst = field == M4Q_F_MODELS ? &s->t_models : field == M4Q_F_X0 ? &s->t_x0 : field == M4Q_F_X_TARG ? &s->t_xtarg :
                 field == M4Q_F_Q ? &s->t_Q : &s->t_Qf;
  if (!matrix) {
    tout.resize(count_items * ns);
    double lo = 0.0, hi = 0.0;
    for (size_t it = 0; it < count_items; ++it) {
      tl.vec(out.data() + it * n, tmp.data());
      for (int i = 0; i < ns; ++i) tout[it * ns + i] = tmp[1 + i];
      lo = it ? std::min(lo, tmp[0]) : tmp[0];
      hi = it ? std::max(hi, tmp[0]) : tmp[0];
    }
    double* range = field == M4Q_F_X0 ? s->tau_x0 : s->tau_targ;
    range[0] = lo; range[1] = hi;
    s->tl_ok[field] = true;
  } else {
    const long ld = (long)n * nblk, lds = (long)ns * nblk;
    tout.resize(count_items * ns * lds);
    DecoupleStat dc;
    for (size_t it = 0; it < count_items; ++it)
      for (int p = 0; p < nblk; ++p) {
        tl.mat(out.data() + it * n * ld + (long)p * n, ld, tmp.data());
        // models: block 0 must carry the trace coordinate through unchanged, the N_p blocks must not touch it; costs: the cross
        // terms with the (constant, shared) trace coordinate drop out of the objective's minimiser, nothing to check
        if (field == M4Q_F_MODELS) dc.see_block(tmp.data(), n, p == 0);
        for (int i = 0; i < ns; ++i)
          for (int j = 0; j < ns; ++j) tout[it * ns * lds + i * lds + (long)p * ns + j] = tmp[(size_t)(1 + i) * n + 1 + j];
      }
    s->tl_ok[field] = field == M4Q_F_MODELS ? dc.ok() : true;
  }
  rc = tdst->alloc(tout.size() * sizeof(double));
  if (rc) return rc;
  HIP_TRY(hipMemcpy(tdst->p, tout.data(), tout.size() * sizeof(double), hipMemcpyHostToDevice));
  return 0;
}

int m4q_session_upload(m4q_session* s, int32_t field, const void* host, size_t bytes) {
  if (!s || field < 0 || field >= M4Q_F_COUNT || !host) return fail(M4Q_E_BADARG, "m4q_session_upload: bad argument");
  if (bytes != s->fbytes[field]) return fail(M4Q_E_BADARG, "field %d expects %zu bytes, got %zu", field, s->fbytes[field], bytes);
  if (bytes == 0) return 0;
  HIP_TRY(hipMemcpyAsync(s->f[field].p, host, bytes, hipMemcpyHostToDevice, s->stream));
  HIP_TRY(hipStreamSynchronize(s->stream));
  const size_t n = s->prob.dim_x, m = s->prob.dim_u, P = s->shape->np;
  const auto* ch = static_cast<const std::complex<double>*>(host);
  int rc = 0;
  if (field == M4Q_F_Q) { s->hQ.assign((const double*)host, (const double*)host + 2 * n * n); s->costs_dirty = true; }
  if (field == M4Q_F_QF) { s->hQf.assign((const double*)host, (const double*)host + 2 * n * n); s->costs_dirty = true; }
  if (field == M4Q_F_R) { s->hR.assign((const double*)host, (const double*)host + 2 * m * m); s->costs_dirty = true; }
  if (field == M4Q_F_X_TARG) {
    const size_t cols = s->prob.target_cols, items = bytes / (16 * n * cols);
    bool same = true;
    for (size_t it = 0; it < items && same; ++it)
      for (size_t c = 1; c < cols && same; ++c)
        same = std::memcmp(ch + (it * cols + c) * n, ch + it * cols * n, 16 * n) == 0;
    s->targ_const = same;
  }
  if (!s->force_complex) {
    if (field == M4Q_F_MODELS) s->sg_ok = false;            // uploaded models: not (known to be) an ensemble of scaled shared generators
    if (field == M4Q_F_MODELS) rc = lift_upload(s, field, ch, bytes / (16 * n * n * (1 + P)), true, (int)(1 + P), s->r_models);
    if (field == M4Q_F_X0) rc = lift_upload(s, field, ch, bytes / (16 * n), false, 1, s->r_x0);
    if (field == M4Q_F_X_TARG) rc = lift_upload(s, field, ch, bytes / (16 * n), false, 1, s->r_xtarg);
    if (field == M4Q_F_Q) rc = lift_upload(s, field, ch, 1, true, 1, s->r_Q);
    if (field == M4Q_F_QF) rc = lift_upload(s, field, ch, 1, true, 1, s->r_Qf);
    if (field == M4Q_F_R) {
      // R acts on the (real) controls: the real path needs Im R = 0
      LiftStat st;
      std::vector<double> rr(m * m);
      for (size_t i = 0; i < m * m; ++i) { st.see(ch[i]); rr[i] = ch[i].real(); }
      s->herm_ok[field] = st.real_enough();
      rc = s->r_R.alloc(rr.size() * 8);
      if (!rc) HIP_TRY(hipMemcpy(s->r_R.p, rr.data(), rr.size() * 8, hipMemcpyHostToDevice));
    }
  }
  return rc;
}

int m4q_session_path(const m4q_session* s) {
  if (!s) return fail(M4Q_E_BADARG, "m4q_session_path: null session");
  // the line-search weights are derived from Q, Qf, R at the first run; before that, answer from the uploads alone
  const bool diag_known = !s->costs_dirty;
  return s->path(diag_known ? s->ls_diag : true);
}

int m4q_session_download(m4q_session* s, int32_t field, void* host, size_t bytes) {
  if (!s || field < 0 || field >= M4Q_F_COUNT || !host) return fail(M4Q_E_BADARG, "m4q_session_download: bad argument");
  if (bytes != s->fbytes[field]) return fail(M4Q_E_BADARG, "field %d holds %zu bytes, asked %zu", field, s->fbytes[field], bytes);
  if (bytes == 0) return 0;
  HIP_TRY(hipMemcpyAsync(host, s->f[field].p, bytes, hipMemcpyDeviceToHost, s->stream));
  HIP_TRY(hipStreamSynchronize(s->stream));
  return check_watchdog(s);
}

int m4q_session_put_state(m4q_session* s, int32_t step, const void* host) {
  if (!s || !host || step < 0 || step > s->prob.n_steps) return fail(M4Q_E_BADARG, "m4q_session_put_state: bad argument");
  const size_t row = (size_t)s->prob.dim_x * 16;
  HIP_TRY(hipMemcpy2DAsync((char*)s->f[M4Q_F_XS].p + (size_t)step * row, (size_t)(s->prob.n_steps + 1) * row, host, row, row,
                           s->B, hipMemcpyHostToDevice, s->stream));
  HIP_TRY(hipStreamSynchronize(s->stream));
  return check_watchdog(s);
}

int m4q_session_get_state(m4q_session* s, int32_t step, void* host) {
  if (!s || !host || step < 0 || step > s->prob.n_steps) return fail(M4Q_E_BADARG, "m4q_session_get_state: bad argument");
  const size_t row = (size_t)s->prob.dim_x * 16;
  HIP_TRY(hipMemcpy2DAsync(host, row, (const char*)s->f[M4Q_F_XS].p + (size_t)step * row, (size_t)(s->prob.n_steps + 1) * row,
                           row, s->B, hipMemcpyDeviceToHost, s->stream));
  HIP_TRY(hipStreamSynchronize(s->stream));
  return check_watchdog(s);
}

void* m4q_session_device_ptr(m4q_session* s, int32_t field) {
  if (!s || field < 0 || field >= M4Q_F_COUNT) return nullptr;
  return s->f[field].p;
}

int m4q_session_bind_output(m4q_session* s, int32_t field, void* device_ptr, size_t bytes) {
  if (!s || !device_ptr || field < M4Q_F_XS || field >= M4Q_F_COUNT) return fail(M4Q_E_BADARG, "m4q_session_bind_output: bad field");
  if (bytes != s->fbytes[field]) return fail(M4Q_E_BADARG, "field %d expects %zu bytes, got %zu", field, s->fbytes[field], bytes);
  s->f[field].release();
  s->f[field].p = device_ptr;
  s->f[field].bytes = bytes;
  s->f[field].owned = false;
  return 0;
}

static int refresh_costs(m4q_session* s) {
  if (!s->costs_dirty) return 0;
  const int n = s->prob.dim_x, m = s->prob.dim_u;
  if (s->hQ.empty() || s->hQf.empty() || s->hR.empty()) return fail(M4Q_E_BADARG, "Q, Qf and R must be uploaded before running");
  std::vector<double> c, w;
  bool diag = true;
  auto take_diag = [&](int k) {
    const int sz = 2 * k;
    for (int i = 0; i < sz; ++i)
      for (int j2 = 0; j2 < sz; ++j2)
        if (i != j2 && c[(size_t)i * sz + j2] != 0.0) diag = false;
    for (int i = 0; i < sz; ++i) w.push_back(c[(size_t)i * sz + i]);
  };
  ls_block(s->hQ.data(), n, c);
  HIP_TRY(hipMemcpy(s->Cq.p, c.data(), c.size() * 8, hipMemcpyHostToDevice));
  take_diag(n);
  ls_block(s->hQf.data(), n, c);
  HIP_TRY(hipMemcpy(s->Cqf.p, c.data(), c.size() * 8, hipMemcpyHostToDevice));
  take_diag(n);
  ls_block(s->hR.data(), m, c);
  HIP_TRY(hipMemcpy(s->Cr.p, c.data(), c.size() * 8, hipMemcpyHostToDevice));
  take_diag(m);
  HIP_TRY(hipMemcpy(s->Wls.p, w.data(), w.size() * 8, hipMemcpyHostToDevice));
  s->ls_diag = diag;
  s->costs_dirty = false;
  return 0;
}

int m4q_session_run(m4q_session* s, int32_t step_begin, int32_t step_end) {
  if (!s || step_begin < 0 || step_end > s->prob.n_steps || step_begin >= step_end)
    return fail(M4Q_E_BADARG, "m4q_session_run: bad step range [%d, %d)", step_begin, step_end);
  int rc = refresh_costs(s);
  if (rc) return rc;
  const m4q_problem& p = s->prob;
  const size_t n = p.dim_x, m = p.dim_u, P = s->shape->np;
  const size_t k = p.plant_kind == M4Q_PLANT_GENERATOR ? n : (size_t)s->shape->d;
  const int path = s->path();
  const bool real_path = path != 0;
  const size_t ns = path >= 2 ? n - 1 : n;         // dimension of the recursion
  m4q::MpcArgs a{};
  a.B = s->B; a.T = p.horizon; a.n_steps = p.n_steps; a.max_iter = p.max_iter; a.warm_start = p.warm_start;
  a.flags = p.qp_flags | (s->targ_const ? 256 : 0) | (s->no_tile ? 512 : 0);      // QP_TARG_CONST, QP_NO_TILE (csrc/m4q_mpc.h), internal
  a.step_begin = step_begin; a.step_end = step_end;
  a.measure_freq = p.measure_freq > 1 ? p.measure_freq : 1;
  a.dt = p.dt; a.sat = p.sat; a.du = p.du; a.ls_tol = p.ls_tol;
  a.models = path >= 2 ? s->t_models.p : real_path ? s->r_models.p : s->f[M4Q_F_MODELS].p;
  a.gens = (const double*)s->sg_gens.p; a.scales = (const double*)s->sg_scales.p;
  a.model_stride = p.model_per_instance ? (long)(ns * ns * (1 + P)) : 0;
  a.x0c = (const cplx*)s->f[M4Q_F_X0].p;
  a.x0s = path >= 2 ? s->t_x0.p : real_path ? s->r_x0.p : s->f[M4Q_F_X0].p;
  a.x_targ = path >= 2 ? s->t_xtarg.p : real_path ? s->r_xtarg.p : s->f[M4Q_F_X_TARG].p;
  a.xt_stride = p.target_per_instance ? (long)(p.target_cols * ns) : 0;
  a.u_targ = (const double*)s->f[M4Q_F_U_TARG].p; a.ut_stride = p.target_per_instance ? (long)(p.target_cols * m) : 0;
  a.Q = path >= 2 ? s->t_Q.p : real_path ? s->r_Q.p : s->f[M4Q_F_Q].p;
  a.Qf = path >= 2 ? s->t_Qf.p : real_path ? s->r_Qf.p : s->f[M4Q_F_QF].p;
  a.R = real_path ? s->r_R.p : s->f[M4Q_F_R].p;
  a.Cq = (const double*)s->Cq.p; a.Cqf = (const double*)s->Cqf.p; a.Cr = (const double*)s->Cr.p;
  a.Wls = s->ls_diag ? (const double*)s->Wls.p : nullptr;
  a.op0 = (const cplx*)s->f[M4Q_F_OP0].p; a.op0_stride = p.plant_per_instance ? (long)(k * k) : 0;
  a.ops = (const cplx*)s->f[M4Q_F_OPS].p; a.ops_stride = p.plant_per_instance ? (long)(m * k * k) : 0;
  if (p.plant_kind == M4Q_PLANT_NONE) { a.op0 = (const cplx*)s->f[M4Q_F_Q].p; a.ops = a.op0; a.op0_stride = a.ops_stride = 0; }
  a.xs = (cplx*)s->f[M4Q_F_XS].p; a.us = (double*)s->f[M4Q_F_US].p;
  a.codes = (int*)s->f[M4Q_F_CODES].p; a.steps_done = (int*)s->f[M4Q_F_STEPS_DONE].p; a.qp_solves = (int*)s->f[M4Q_F_QP_SOLVES].p;
  a.Xg = (cplx*)s->f[M4Q_F_X_GUESS].p; a.Ug = (double*)s->f[M4Q_F_U_GUESS].p;
  a.ws_Xg = s->wsXg.p; a.ws_Ug = (double*)s->wsUg.p;
  a.ws_gains = s->wsG.p;
  a.queue = (int*)s->queue.p;
  a.head_done = (int*)s->head_done.p;
  {
    double seconds = 300.0;                                  // below the 7 minutes of silence after which a GPU box kills a job
    if (const char* e = std::getenv("M4Q_KERNEL_TIMEOUT_S")) {
      const double v = std::atof(e);
      if (v > 0) seconds = v;
    }
    a.deadline_ticks = (unsigned long long)(seconds * 1e8);  // s_memrealtime counts at 100 MHz
  }
  if (s->launched) {
    // an earlier launch has not been synchronised yet: its watchdog flag (bytes 4..8) must survive until check_watchdog reads it
    HIP_TRY(hipMemsetAsync(s->queue.p, 0, 4, s->stream));
    HIP_TRY(hipMemsetAsync((char*)s->queue.p + 8, 0, 184, s->stream));
  } else {
    HIP_TRY(hipMemsetAsync(s->queue.p, 0, 192, s->stream));
  }
  HIP_TRY(hipMemsetAsync(s->head_done.p, 0, (size_t)s->B * 4, s->stream));
  if (step_begin == 0) {
    HIP_TRY(hipMemsetAsync(s->f[M4Q_F_QP_SOLVES].p, 0, s->fbytes[M4Q_F_QP_SOLVES], s->stream));
    HIP_TRY(hipMemsetAsync(s->f[M4Q_F_CODES].p, 0, s->fbytes[M4Q_F_CODES], s->stream));
    HIP_TRY(hipMemsetAsync(s->f[M4Q_F_STEPS_DONE].p, 0, s->fbytes[M4Q_F_STEPS_DONE], s->stream));
  }
  // a long step-by-step run never reads its timings: fold finished launches so the event list stays short
  while (s->pending.size() > 64 && hipEventQuery(s->pending.front().second) == hipSuccess) {
    float ms = 0.f;
    if (hipEventElapsedTime(&ms, s->pending.front().first, s->pending.front().second) == hipSuccess) {
      s->folded_ms += ms;
      ++s->folded_n;
    }
    (void)hipEventDestroy(s->pending.front().first);
    (void)hipEventDestroy(s->pending.front().second);
    s->pending.erase(s->pending.begin());
  }
  hipEvent_t e0, e1;
  HIP_TRY(hipEventCreate(&e0));
  HIP_TRY(hipEventCreate(&e1));
  HIP_TRY(hipEventRecord(e0, s->stream));
  rc = s->mpc_ops->launch_mpc(a, p.plant_kind, path, path == 4 ? s->grid_sg : s->grid, s->stream);
  if (rc) return fail(rc, "mpc kernel launch failed: %s", hipGetErrorString((hipError_t)(-rc)));
  HIP_TRY(hipEventRecord(e1, s->stream));
  s->pending.emplace_back(e0, e1);
  s->launched = true;
  return 0;
}

int m4q_session_sync(m4q_session* s) {
  if (!s) return fail(M4Q_E_BADARG, "m4q_session_sync: null session");
  HIP_TRY(hipStreamSynchronize(s->stream));
  return check_watchdog(s);
}

int m4q_session_set_codes(m4q_session* s, const int32_t* codes) {
  if (!s || !codes) return fail(M4Q_E_BADARG, "m4q_session_set_codes: bad argument");
  HIP_TRY(hipMemcpyAsync(s->f[M4Q_F_CODES].p, codes, (size_t)s->B * 4, hipMemcpyHostToDevice, s->stream));
  HIP_TRY(hipStreamSynchronize(s->stream));
  return check_watchdog(s);
}

int m4q_session_kernel_ms(m4q_session* s, double* total_ms, int32_t* launches) {
  if (!s) return fail(M4Q_E_BADARG, "m4q_session_kernel_ms: null session");
  HIP_TRY(hipStreamSynchronize(s->stream));
  double tot = s->folded_ms;
  int n = s->folded_n;
  s->folded_ms = 0.0;
  s->folded_n = 0;
  for (auto& pr : s->pending) {
    float ms = 0.f;
    HIP_TRY(hipEventElapsedTime(&ms, pr.first, pr.second));
    tot += ms;
    ++n;
    (void)hipEventDestroy(pr.first);
    (void)hipEventDestroy(pr.second);
  }
  s->pending.clear();
  if (total_ms) *total_ms = tot;
  if (launches) *launches = n;
  return check_watchdog(s);     // (timings of a launch that left through its watchdog are not handed out as if it had finished)
}

int m4q_session_qp_stats(m4q_session* s, int64_t* out6) {
  if (!s || !out6) return fail(M4Q_E_BADARG, "m4q_session_qp_stats: bad argument");
  HIP_TRY(hipStreamSynchronize(s->stream));
  unsigned long long q[24];
  HIP_TRY(hipMemcpy(q, s->queue.p, sizeof(q), hipMemcpyDeviceToHost));
  for (int i = 0; i < 6; ++i) out6[i] = (int64_t)q[1 + i];
  if (std::getenv("M4Q_QP_TRACE"))
    fprintf(stderr, "m4q: exact QP: %llu row sweeps in %llu wavefront passes (x4 rows = %llu): lane efficiency %.2f\n", q[2], q[7],
            4 * q[7], q[7] ? (double)q[2] / (4.0 * (double)q[7]) : 0.0);
  if (std::getenv("M4Q_PHASE_TRACE")) {          // -DM4Q_DEV_PHASE_CLOCK builds: wavefront time per phase of the main loop, 100 MHz ticks
    static const char* names[16] = {"draw/resume", "backward | exact: adjoint pass", "forward", "handover", "line search", "step done (plant, shift)", "publish",
                                    "passes", "guess update", "passes with a line search", "exact: pinned sweep", "exact: policy rollout",
                                    "exact: ratio rollout", "exact: blend", "exact: open rollout", "exact: bookkeeping + copies"};
    unsigned long long tot = 0;
    for (int i = 0; i < 16; ++i) tot += (i == 7 || i == 9) ? 0 : q[8 + i];
    for (int i = 0; i < 16; ++i)
      fprintf(stderr, "m4q phase %-30s %14llu %s  %5.1f %%\n", names[i], q[8 + i], (i == 7 || i == 9) ? "     " : "ticks",
              (i != 7 && i != 9 && tot) ? 100.0 * (double)q[8 + i] / (double)tot : 0.0);
  }
  return check_watchdog(s);
}

int m4q_session_info(const m4q_session* s, int64_t* hbm_bytes, int32_t* grid, int32_t* lds_bytes) {
  if (!s) return fail(M4Q_E_BADARG, "m4q_session_info: null session");
  int64_t tot = 0;
  for (int i = 0; i < M4Q_F_COUNT; ++i) tot += (int64_t)s->f[i].bytes;
  tot += (int64_t)(s->wsXg.bytes + s->wsUg.bytes + s->wsG.bytes);
  if (hbm_bytes) *hbm_bytes = tot;
  if (grid) *grid = s->path() == 4 ? s->grid_sg : s->grid;
  if (lds_bytes) *lds_bytes = (int32_t)s->mpc_ops->mpc_lds_bytes(s->path(), (s->prob.qp_flags & M4Q_QP_EXACT_BOX) != 0);
  return 0;
}

// ------------------------------------------------------------------------------------------
// one-shot host entry points
// ------------------------------------------------------------------------------------------
namespace {
struct Tmp {
  std::vector<DevBuf*> bufs;
  ~Tmp() { for (DevBuf* b : bufs) delete b; }
  int up(const void* host, size_t bytes, void** out) {
    DevBuf* b = new DevBuf();
    bufs.push_back(b);
    int rc = b->alloc(bytes);
    if (rc) return rc;
    if (host && bytes) {
      hipError_t e = hipMemcpy(b->p, host, bytes, hipMemcpyHostToDevice);
      if (e != hipSuccess) return fail(-(int)e, "hipMemcpy H2D: %s", hipGetErrorString(e));
    }
    *out = b->p;
    return 0;
  }
};
int down(void* host, const void* dev, size_t bytes) {
  if (!host || !bytes) return 0;
  hipError_t e = hipMemcpy(host, dev, bytes, hipMemcpyDeviceToHost);
  if (e != hipSuccess) return fail(-(int)e, "hipMemcpy D2H: %s", hipGetErrorString(e));
  return 0;
}
int need_device() {
  int n = 0;
  hipError_t e = hipGetDeviceCount(&n);
  if (e != hipSuccess || n == 0) return fail(M4Q_E_NODEVICE, "no HIP device: %s", hipGetErrorString(e));
  return 0;
}
}  // namespace

int m4q_linearize_batch(int32_t B, int32_t dim_x, int32_t dim_u, int32_t order, int32_t T, const double* models,
                        int32_t model_per_instance, const double* X, const double* U, double* A_ls, double* B_ls,
                        double* Delta_ls) {
  const m4q::ShapeOps* sh = find_shape(dim_x, dim_u, order);
  if (!sh) return fail(M4Q_E_UNSUPPORTED, "no kernel for dim_x=%d dim_u=%d order=%d", dim_x, dim_u, order);
  if (B <= 0 || T <= 0 || !models || !X || !U || !A_ls || !B_ls || !Delta_ls) return fail(M4Q_E_BADARG, "m4q_linearize_batch: bad argument");
  int rc = need_device();
  if (rc) return rc;
  const size_t n = dim_x, m = dim_u, P = sh->np, C = 16;
  Tmp t;
  m4q::LinArgs a{};
  a.B = B; a.T = T;
  a.model_stride = model_per_instance ? (long)(n * n * (1 + P)) : 0;
  void *d_models, *d_X, *d_U, *d_A, *d_B, *d_D;
  if ((rc = t.up(models, (model_per_instance ? B : 1) * n * n * (1 + P) * C, &d_models))) return rc;
  if ((rc = t.up(X, (size_t)B * T * n * C, &d_X))) return rc;
  if ((rc = t.up(U, (size_t)B * T * m * 8, &d_U))) return rc;
  if ((rc = t.up(nullptr, (size_t)B * T * n * n * C, &d_A))) return rc;
  if ((rc = t.up(nullptr, (size_t)B * T * n * m * C, &d_B))) return rc;
  if ((rc = t.up(nullptr, (size_t)B * T * n * C, &d_D))) return rc;
  a.models = (const cplx*)d_models; a.X = (const cplx*)d_X; a.U = (const double*)d_U;
  a.A_ls = (cplx*)d_A; a.B_ls = (cplx*)d_B; a.D_ls = (cplx*)d_D;
  rc = sh->launch_linearize(a, nullptr);
  if (rc) return fail(rc, "linearize launch failed");
  HIP_TRY(hipDeviceSynchronize());
  if ((rc = down(A_ls, d_A, (size_t)B * T * n * n * C))) return rc;
  if ((rc = down(B_ls, d_B, (size_t)B * T * n * m * C))) return rc;
  return down(Delta_ls, d_D, (size_t)B * T * n * C);
}

int m4q_quad_program_batch(int32_t B, int32_t dim_x, int32_t dim_u, int32_t T, int32_t qp_flags, double sat, double du,
                           const double* x_init, const double* X_bm, const double* U_bm, int32_t bm_per_instance,
                           const double* Q_ls, const double* R_ls, const double* A_ls, const double* B_ls,
                           const double* Delta_ls, const double* u_prev, double* X_opt, double* U_opt, double* cost,
                           double* gains) {
  const m4q::ShapeOps* sh = find_shape_any_order(dim_x, dim_u);
  if (!sh) return fail(M4Q_E_UNSUPPORTED, "no kernel for dim_x=%d dim_u=%d", dim_x, dim_u);
  if (B <= 0 || T <= 0 || !x_init || !X_bm || !U_bm || !Q_ls || !R_ls || !A_ls || !B_ls || !X_opt || !U_opt || !cost)
    return fail(M4Q_E_BADARG, "m4q_quad_program_batch: bad argument");
  if (!(sat > 0)) return fail(M4Q_E_BADARG, "sat must be positive");
  if (qp_flags & ~(M4Q_QP_REF_LQR | M4Q_QP_DU_BAND | M4Q_QP_EXACT_BOX))
    return fail(M4Q_E_BADARG, "qp_flags has bits outside M4Q_QP_REF_LQR | M4Q_QP_DU_BAND | M4Q_QP_EXACT_BOX (0x%x)", qp_flags);
  if ((qp_flags & M4Q_QP_EXACT_BOX) && (qp_flags & M4Q_QP_REF_LQR))
    return fail(M4Q_E_BADARG, "M4Q_QP_EXACT_BOX cannot be combined with M4Q_QP_REF_LQR");
  int rc = need_device();
  if (rc) return rc;
  const size_t n = dim_x, m = dim_u, C = 16;
  Tmp t;
  m4q::QpArgs a{};
  a.B = B; a.T = T; a.flags = qp_flags; a.sat = sat; a.du = du;
  void *d_x, *d_xb, *d_ub, *d_q, *d_r, *d_a, *d_b, *d_d = nullptr, *d_up = nullptr, *d_xo, *d_uo, *d_c, *d_g, *d_it = nullptr;
  if ((rc = t.up(x_init, (size_t)B * n * C, &d_x))) return rc;
  if ((rc = t.up(X_bm, (bm_per_instance ? B : 1) * (size_t)(T + 1) * n * C, &d_xb))) return rc;
  if ((rc = t.up(U_bm, (bm_per_instance ? B : 1) * (size_t)T * m * 8, &d_ub))) return rc;
  if ((rc = t.up(Q_ls, (size_t)(T + 1) * n * n * C, &d_q))) return rc;
  if ((rc = t.up(R_ls, (size_t)T * m * m * C, &d_r))) return rc;
  if ((rc = t.up(A_ls, (size_t)B * T * n * n * C, &d_a))) return rc;
  if ((rc = t.up(B_ls, (size_t)B * T * n * m * C, &d_b))) return rc;
  if (Delta_ls && (rc = t.up(Delta_ls, (size_t)B * T * n * C, &d_d))) return rc;
  if (u_prev && (rc = t.up(u_prev, (size_t)B * m * 8, &d_up))) return rc;
  if ((rc = t.up(nullptr, (size_t)B * (T + 1) * n * C, &d_xo))) return rc;
  if ((rc = t.up(nullptr, (size_t)B * T * m * 8, &d_uo))) return rc;
  if ((rc = t.up(nullptr, (size_t)B * 8, &d_c))) return rc;
  if ((rc = t.up(nullptr, (size_t)B * T * (n + 1) * m * C, &d_g))) return rc;
  a.x_init = (const cplx*)d_x;
  a.X_bm = (const cplx*)d_xb; a.xbm_stride = bm_per_instance ? (long)((T + 1) * n) : 0;
  a.U_bm = (const double*)d_ub; a.ubm_stride = bm_per_instance ? (long)(T * m) : 0;
  a.Q_ls = (const cplx*)d_q; a.R_ls = (const cplx*)d_r;
  a.A_ls = (const cplx*)d_a; a.B_ls = (const cplx*)d_b; a.D_ls = (const cplx*)d_d; a.u_prev = (const double*)d_up;
  a.X_opt = (cplx*)d_xo; a.U_opt = (double*)d_uo; a.cost = (double*)d_c; a.gains = (cplx*)d_g;
  if (qp_flags & M4Q_QP_EXACT_BOX) {
    void *d_xa, *d_ua, *d_st;
    if ((double)B * (T + 1) * n * C * 2 >= 4294967296.0)
      return fail(M4Q_E_BADARG, "M4Q_QP_EXACT_BOX: batch too large for one call (trajectory workspace must stay below 4 GiB)");
    if ((rc = t.up(nullptr, 2 * (size_t)B * (T + 1) * n * C, &d_xa))) return rc;
    if ((rc = t.up(nullptr, 2 * (size_t)B * T * m * 8, &d_ua))) return rc;
    if ((rc = t.up(nullptr, (size_t)B * T * m * 8, &d_st))) return rc;
    a.X_alt = (cplx*)d_xa; a.U_alt = (double*)d_ua; a.pin_stat = (double*)d_st;
    if (getenv("M4Q_QP_TRACE")) {
      if ((rc = t.up(nullptr, (size_t)B * 4, &d_it))) return rc;
      a.sweep_counts = (int*)d_it;
    }
  }
  rc = sh->launch_qp(a, nullptr);
  if (rc) return fail(rc, "qp launch failed");
  HIP_TRY(hipDeviceSynchronize());
  if (d_it) {                                             // diagnostic: Newton iterations per instance
    std::vector<int> it(B);
    if ((rc = down(it.data(), d_it, (size_t)B * 4))) return rc;
    long sum = 0;
    int mx = 0;
    for (int v : it) { sum += v; mx = v > mx ? v : mx; }
    fprintf(stderr, "m4q: exact box QP: %d instances, pinned sweeps mean %.2f max %d\n", B, (double)sum / B, mx);
  }
  if ((rc = down(X_opt, d_xo, (size_t)B * (T + 1) * n * C))) return rc;
  if ((rc = down(U_opt, d_uo, (size_t)B * T * m * 8))) return rc;
  if ((rc = down(cost, d_c, (size_t)B * 8))) return rc;
  return down(gains, d_g, (size_t)B * T * (n + 1) * m * C);
}

int m4q_discretize_batch(int32_t B, int32_t dim_x, int32_t dim_u, int32_t order, double dt, const double* generators,
                         int32_t gen_per_instance, const double* scales, double* models) {
  const m4q::ShapeOps* sh = find_shape(dim_x, dim_u, order);
  if (!sh) return fail(M4Q_E_UNSUPPORTED, "no kernel for dim_x=%d dim_u=%d order=%d", dim_x, dim_u, order);
  if (B <= 0 || !generators || !models) return fail(M4Q_E_BADARG, "m4q_discretize_batch: bad argument");
  int rc = need_device();
  if (rc) return rc;
  const size_t n = dim_x, m = dim_u, P = sh->np, C = 16;
  Tmp t;
  m4q::DiscArgs a{};
  a.B = B; a.dt = dt;
  void *d_g, *d_s = nullptr, *d_m;
  if ((rc = t.up(generators, (gen_per_instance ? B : 1) * (1 + m) * n * n * C, &d_g))) return rc;
  if (scales && (rc = t.up(scales, (size_t)B * (1 + m) * 8, &d_s))) return rc;
  if ((rc = t.up(nullptr, (size_t)B * n * n * (1 + P) * C, &d_m))) return rc;
  a.gens = d_g; a.gen_stride = gen_per_instance ? (long)((1 + m) * n * n) : 0;
  a.scales = (const double*)d_s; a.models = d_m;
  rc = sh->launch_discretize(a, 0, nullptr);
  if (rc) return fail(rc, "discretize launch failed");
  HIP_TRY(hipDeviceSynchronize());
  return down(models, d_m, (size_t)B * n * n * (1 + P) * C);
}

int m4q_session_build_models(m4q_session* s, double dt, const double* generators, int32_t gen_per_instance,
                             const double* scales) {
  if (!s || !generators) return fail(M4Q_E_BADARG, "m4q_session_build_models: bad argument");
  const m4q_problem& p = s->prob;
  const size_t n = p.dim_x, m = p.dim_u, P = s->shape->np, C = 16;
  const size_t nset = gen_per_instance ? (size_t)s->B : 1;
  const size_t nmodels = p.model_per_instance ? (size_t)s->B : 1;
  if ((gen_per_instance || scales) && !p.model_per_instance)
    return fail(M4Q_E_BADARG, "per-instance generators or scales need model_per_instance = 1");
  Tmp t;
  m4q::DiscArgs a{};
  a.B = (int)nmodels; a.dt = dt;
  void *d_g, *d_s = nullptr;
  int rc;
  if ((rc = t.up(generators, nset * (1 + m) * n * n * C, &d_g))) return rc;
  if (scales && (rc = t.up(scales, (size_t)s->B * (1 + m) * 8, &d_s))) return rc;
  a.gens = d_g; a.gen_stride = gen_per_instance ? (long)((1 + m) * n * n) : 0;
  a.scales = (const double*)d_s; a.models = s->f[M4Q_F_MODELS].p;
  rc = s->shape->launch_discretize(a, 0, s->stream);
  if (rc) return fail(rc, "discretize launch failed");
  s->herm_ok[M4Q_F_MODELS] = false;
  s->sg_ok = false;
  if (!s->force_complex) {
    // the same expansion in the Hermitian operator basis: lift the generators (few, or one set per member)
    const HermBasis hb(s->shape->d);
    const auto* g = reinterpret_cast<const std::complex<double>*>(generators);
    std::vector<std::complex<double>> tmp(n * n);
    std::vector<double> lifted(nset * (1 + m) * n * n);
    LiftStat st;
    for (size_t q = 0; q < nset * (1 + m); ++q) {
      hb.lift_mat(g + q * n * n, (long)n, tmp.data(), (long)n);
      for (size_t e = 0; e < n * n; ++e) { st.see(tmp[e]); lifted[q * n * n + e] = tmp[e].real(); }
    }
    s->tl_ok[M4Q_F_MODELS] = false;
    if (st.real_enough()) {
      void* d_gr;
      if ((rc = t.up(lifted.data(), lifted.size() * 8, &d_gr))) return rc;
      if ((rc = s->r_models.alloc(nmodels * n * n * (1 + P) * 8))) return rc;
      m4q::DiscArgs r = a;
      r.gens = d_gr; r.models = s->r_models.p;
      rc = s->shape->launch_discretize(r, 1, s->stream);
      if (rc) return fail(rc, "discretize launch failed");
      s->herm_ok[M4Q_F_MODELS] = true;
      if (!s->no_traceless) {
        // ... and on the traceless coordinates: generators that leave the trace coordinate alone (row 0 and column 0 of O^T G O
        // zero: trace-preserving and unital, every -i[H, .] is) have block-diagonal products, so the expansion of their
        // (n-1) x (n-1) blocks IS the traceless block of the model
        const Traceless tl(s->shape->d);
        const size_t m1 = n - 1;
        std::vector<double> blocks(nset * (1 + m) * m1 * m1), rot(n * n);
        DecoupleStat dc;
        for (size_t q = 0; q < nset * (1 + m); ++q) {
          tl.mat(lifted.data() + q * n * n, (long)n, rot.data());
          dc.see_block(rot.data(), (int)n, false);
          for (size_t i = 0; i < m1; ++i)
            for (size_t j2 = 0; j2 < m1; ++j2) blocks[q * m1 * m1 + i * m1 + j2] = rot[(1 + i) * n + 1 + j2];
        }
        if (dc.ok()) {
          void* d_gt;
          if ((rc = t.up(blocks.data(), blocks.size() * 8, &d_gt))) return rc;
          if ((rc = s->t_models.alloc(nmodels * m1 * m1 * (1 + P) * 8))) return rc;
          m4q::DiscArgs q2 = a;
          q2.gens = d_gt; q2.gen_stride = gen_per_instance ? (long)((1 + m) * m1 * m1) : 0; q2.models = s->t_models.p;
          rc = s->shape->launch_discretize(q2, 2, s->stream);
          if (rc) return fail(rc, "discretize launch failed");
          s->tl_ok[M4Q_F_MODELS] = true;
          // shared-generator form (path 4): ONE generator set, order 1 - member i's model is [I + dt s_i0 L_0 | dt s_ik L_k]; keep
          // dt L_k on the traceless coordinates and the scales (ones when none were given)
          if (!gen_per_instance && p.order == 1 && p.model_per_instance && !s->no_sg) {
            std::vector<double> g(blocks.size());
            for (size_t e = 0; e < blocks.size(); ++e) g[e] = dt * blocks[e];
            std::vector<double> sc((size_t)s->B * (1 + m), 1.0);
            if (scales) std::copy(scales, scales + sc.size(), sc.begin());
            if ((rc = s->sg_gens.alloc(g.size() * 8)) || (rc = s->sg_scales.alloc(sc.size() * 8))) return rc;
            HIP_TRY(hipMemcpy(s->sg_gens.p, g.data(), g.size() * 8, hipMemcpyHostToDevice));
            HIP_TRY(hipMemcpy(s->sg_scales.p, sc.data(), sc.size() * 8, hipMemcpyHostToDevice));
            s->sg_ok = true;
          }
        }
      }
    }
  }
  HIP_TRY(hipStreamSynchronize(s->stream));
  return 0;
}

int m4q_plant_step_batch(int32_t B, int32_t dim_x, int32_t dim_u, int32_t plant_kind, double dt, const double* x,
                         const double* u, const double* op0, const double* ops, int32_t plant_per_instance,
                         double* x_next) {
  const m4q::ShapeOps* sh = find_shape_any_order(dim_x, dim_u, /*plant_ok=*/true);
  if (!sh) return fail(M4Q_E_UNSUPPORTED, "no kernel for dim_x=%d dim_u=%d", dim_x, dim_u);
  if (B <= 0 || !x || !u || !op0 || !ops || !x_next || (plant_kind != M4Q_PLANT_HAMILTONIAN && plant_kind != M4Q_PLANT_GENERATOR))
    return fail(M4Q_E_BADARG, "m4q_plant_step_batch: bad argument");
  int rc = need_device();
  if (rc) return rc;
  const size_t n = dim_x, m = dim_u, C = 16;
  const size_t k = plant_kind == M4Q_PLANT_GENERATOR ? n : (size_t)sh->d;
  Tmp t;
  m4q::PlantArgs a{};
  a.B = B; a.kind = plant_kind; a.dt = dt;
  void *d_x, *d_u, *d_0, *d_k, *d_o;
  if ((rc = t.up(x, (size_t)B * n * C, &d_x))) return rc;
  if ((rc = t.up(u, (size_t)B * m * 8, &d_u))) return rc;
  if ((rc = t.up(op0, (plant_per_instance ? B : 1) * k * k * C, &d_0))) return rc;
  if ((rc = t.up(ops, (plant_per_instance ? B : 1) * m * k * k * C, &d_k))) return rc;
  if ((rc = t.up(nullptr, (size_t)B * n * C, &d_o))) return rc;
  a.x = (const cplx*)d_x; a.u = (const double*)d_u;
  a.op0 = (const cplx*)d_0; a.op0_stride = plant_per_instance ? (long)(k * k) : 0;
  a.ops = (const cplx*)d_k; a.ops_stride = plant_per_instance ? (long)(m * k * k) : 0;
  a.x_next = (cplx*)d_o;
  rc = sh->launch_plant(a, nullptr);
  if (rc) return fail(rc, "plant launch failed");
  HIP_TRY(hipDeviceSynchronize());
  return down(x_next, d_o, (size_t)B * n * C);
}

int m4q_mpc_batch(const m4q_problem* p, int32_t B, const double* models, const double* x0, const double* X_targ,
                  const double* U_targ, const double* Q, const double* R, const double* Qf, const double* op0,
                  const double* ops, double* xs, double* us, int32_t* exit_codes, int32_t* steps_done,
                  int32_t* qp_solves) {
  if (!p || !models || !x0 || !X_targ || !U_targ || !Q || !R || !Qf || !xs || !us)
    return fail(M4Q_E_BADARG, "m4q_mpc_batch: bad argument");
  if (p->plant_kind == M4Q_PLANT_NONE) return fail(M4Q_E_BADARG, "m4q_mpc_batch needs a device plant; use the session API for host plants");
  if (!op0 || !ops) return fail(M4Q_E_BADARG, "m4q_mpc_batch: plant operators missing");
  m4q_session* s = nullptr;
  int rc = m4q_session_create(p, B, -1, &s);
  if (rc) return rc;
  struct Guard { m4q_session* s; ~Guard() { m4q_session_destroy(s); } } guard{s};
  const void* in[9] = {models, x0, X_targ, U_targ, Q, R, Qf, op0, ops};
  for (int f = M4Q_F_MODELS; f <= M4Q_F_OPS; ++f)
    if ((rc = m4q_session_upload(s, f, in[f], s->fbytes[f]))) return rc;
  if ((rc = m4q_session_run(s, 0, p->n_steps))) return rc;
  if ((rc = m4q_session_sync(s))) return rc;
  if ((rc = m4q_session_download(s, M4Q_F_XS, xs, s->fbytes[M4Q_F_XS]))) return rc;
  if ((rc = m4q_session_download(s, M4Q_F_US, us, s->fbytes[M4Q_F_US]))) return rc;
  if (exit_codes && (rc = m4q_session_download(s, M4Q_F_CODES, exit_codes, s->fbytes[M4Q_F_CODES]))) return rc;
  if (steps_done && (rc = m4q_session_download(s, M4Q_F_STEPS_DONE, steps_done, s->fbytes[M4Q_F_STEPS_DONE]))) return rc;
  if (qp_solves && (rc = m4q_session_download(s, M4Q_F_QP_SOLVES, qp_solves, s->fbytes[M4Q_F_QP_SOLVES]))) return rc;
  return 0;
}

int m4q_session_copy_final_state(m4q_session* s, void* dst_dev) {
  if (!s || !dst_dev) return fail(M4Q_E_BADARG, "m4q_session_copy_final_state: bad argument");
  const size_t row = (size_t)s->prob.dim_x * 16;
  HIP_TRY(hipMemcpy2DAsync(dst_dev, row, (const char*)s->f[M4Q_F_XS].p + (size_t)s->prob.n_steps * row,
                           (size_t)(s->prob.n_steps + 1) * row, row, s->B, hipMemcpyDeviceToDevice, s->stream));
  return 0;
}

// the watchdog flag of the launches queued so far, as one int32 in caller-owned device memory (a gather buffer's status word:
// rank dst learns from the gathered bytes that some rank's launch abandoned itself); enqueued on the session stream
int m4q_session_copy_status(m4q_session* s, void* dst_dev) {
  if (!s || !dst_dev) return fail(M4Q_E_BADARG, "m4q_session_copy_status: bad argument");
  HIP_TRY(hipMemcpyAsync(dst_dev, (const char*)s->queue.p + 4, 4, hipMemcpyDeviceToDevice, s->stream));
  return 0;
}

// ------------------------------------------------------------------------------------------
// communicator: RCCL through dlopen (no link-time dependency), its own stream, one event per gather slot
// ------------------------------------------------------------------------------------------
namespace {
struct Rccl {
  void* h = nullptr;
  decltype(&ncclGetUniqueId) GetUniqueId = nullptr;
  decltype(&ncclCommInitRank) CommInitRank = nullptr;
  decltype(&ncclCommDestroy) CommDestroy = nullptr;
  decltype(&ncclGather) Gather = nullptr;
  decltype(&ncclAllReduce) AllReduce = nullptr;
  decltype(&ncclGetErrorString) GetErrorString = nullptr;
  std::string why;
};
Rccl* rccl() {
  static Rccl r;
  static bool tried = false;
  if (tried) return &r;
  tried = true;
  const char* env = std::getenv("M4Q_RCCL_LIB");
  const char* names[] = {env, "librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1", "/opt/rocm/lib/librccl.so"};
  for (const char* n : names) {
    if (!n || !*n) continue;
    r.h = dlopen(n, RTLD_NOW | RTLD_LOCAL);
    if (r.h) break;
    const char* e = dlerror();          // NULL when no error is pending: never into a std::string as it is
    r.why = e ? e : "dlopen failed (no dlerror text)";
  }
  if (!r.h) return &r;
  bool ok = true;
  auto sym = [&](const char* n) { void* p = dlsym(r.h, n); if (!p) { ok = false; r.why = std::string("missing symbol ") + n; } return p; };
  r.GetUniqueId = (decltype(r.GetUniqueId))sym("ncclGetUniqueId");
  r.CommInitRank = (decltype(r.CommInitRank))sym("ncclCommInitRank");
  r.CommDestroy = (decltype(r.CommDestroy))sym("ncclCommDestroy");
  r.Gather = (decltype(r.Gather))sym("ncclGather");
  r.AllReduce = (decltype(r.AllReduce))sym("ncclAllReduce");
  r.GetErrorString = (decltype(r.GetErrorString))sym("ncclGetErrorString");
  if (!ok) { dlclose(r.h); r.h = nullptr; }
  return &r;
}
int need_rccl(Rccl** out) {
  Rccl* r = rccl();
  if (!r->h) return fail(M4Q_E_COMM, "librccl.so could not be loaded (%s); set M4Q_RCCL_LIB", r->why.c_str());
  *out = r;
  return 0;
}
#define NCCL_TRY(r, expr)                                                                     \
  do {                                                                                         \
    ncclResult_t e_ = (expr);                                                                  \
    if (e_ != ncclSuccess) return fail(M4Q_E_COMM, "%s: %s", #expr, (r)->GetErrorString(e_)); \
  } while (0)
}  // namespace

struct m4q_comm {
  ncclComm_t comm = nullptr;
  int rank = 0, world = 1, device = 0;
  hipStream_t stream = nullptr;
  hipEvent_t slot[8] = {};
  hipEvent_t dep = nullptr;
  double* scratch = nullptr;        // 64 doubles in, 64 out (m4q_comm_allreduce_f64)
};

int m4q_comm_unique_id(void* id128) {
  static_assert(sizeof(ncclUniqueId) == M4Q_UNIQUE_ID_BYTES, "unique id size");
  if (!id128) return fail(M4Q_E_BADARG, "m4q_comm_unique_id: null buffer");
  Rccl* r;
  int rc = need_rccl(&r);
  if (rc) return rc;
  ncclUniqueId id;
  NCCL_TRY(r, r->GetUniqueId(&id));
  std::memcpy(id128, &id, sizeof(id));
  return 0;
}

int m4q_comm_create(int32_t rank, int32_t world, const void* id128, int32_t device, m4q_comm** out) {
  if (!out || !id128 || world < 1 || rank < 0 || rank >= world) return fail(M4Q_E_BADARG, "m4q_comm_create: bad argument");
  int rc = need_device();
  if (rc) return rc;
  Rccl* r;
  if ((rc = need_rccl(&r))) return rc;
  if (device >= 0) HIP_TRY(hipSetDevice(device));
  m4q_comm* c = new m4q_comm();
  c->rank = rank;
  c->world = world;
  struct Guard { m4q_comm* c; ~Guard() { if (c) m4q_comm_destroy(c); } } guard{c};
  HIP_TRY(hipGetDevice(&c->device));
  HIP_TRY(hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking));
  for (auto& e : c->slot) HIP_TRY(hipEventCreateWithFlags(&e, hipEventDisableTiming));
  HIP_TRY(hipEventCreateWithFlags(&c->dep, hipEventDisableTiming));
  HIP_TRY(hipMalloc((void**)&c->scratch, 128 * sizeof(double)));
  ncclUniqueId id;
  std::memcpy(&id, id128, sizeof(id));
  NCCL_TRY(r, r->CommInitRank(&c->comm, world, id, rank));
  guard.c = nullptr;
  *out = c;
  return 0;
}

void m4q_comm_destroy(m4q_comm* c) {
  if (!c) return;
  if (c->stream) (void)hipStreamSynchronize(c->stream);
  if (c->comm) (void)rccl()->CommDestroy(c->comm);
  for (auto& e : c->slot) if (e) (void)hipEventDestroy(e);
  if (c->dep) (void)hipEventDestroy(c->dep);
  if (c->scratch) (void)hipFree(c->scratch);
  if (c->stream) (void)hipStreamDestroy(c->stream);
  delete c;
}

int m4q_comm_gather(m4q_comm* c, m4q_session* after, const void* send_dev, void* recv_dev, size_t bytes, int32_t dst, int32_t slot) {
  if (!c || !send_dev || bytes == 0 || dst < 0 || dst >= c->world || slot < 0 || slot >= 8 || (c->rank == dst && !recv_dev))
    return fail(M4Q_E_BADARG, "m4q_comm_gather: bad argument");
  Rccl* r = rccl();
  if (after) {
    HIP_TRY(hipEventRecord(c->dep, after->stream));
    HIP_TRY(hipStreamWaitEvent(c->stream, c->dep, 0));
  }
  NCCL_TRY(r, r->Gather(send_dev, recv_dev, bytes, ncclUint8, dst, c->comm, c->stream));
  HIP_TRY(hipEventRecord(c->slot[slot], c->stream));
  return 0;
}

int m4q_comm_wait(m4q_comm* c, int32_t slot) {
  if (!c || slot >= 8) return fail(M4Q_E_BADARG, "m4q_comm_wait: bad argument");
  if (slot < 0) HIP_TRY(hipStreamSynchronize(c->stream));
  else HIP_TRY(hipEventSynchronize(c->slot[slot]));
  return 0;
}

int m4q_comm_allreduce_f64(m4q_comm* c, double* inout_host, int32_t n, int32_t op) {
  if (!c || n < 0 || n > 64 || (n > 0 && !inout_host) || (op != 0 && op != 1)) return fail(M4Q_E_BADARG, "m4q_comm_allreduce_f64: bad argument");
  Rccl* r = rccl();
  double one = 0.0;
  const int cnt = n > 0 ? n : 1;                      // n = 0: a barrier (one dummy element)
  HIP_TRY(hipMemcpyAsync(c->scratch, n > 0 ? inout_host : &one, cnt * sizeof(double), hipMemcpyHostToDevice, c->stream));
  NCCL_TRY(r, r->AllReduce(c->scratch, c->scratch + 64, cnt, ncclDouble, op == 0 ? ncclSum : ncclMax, c->comm, c->stream));
  HIP_TRY(hipMemcpyAsync(n > 0 ? inout_host : &one, c->scratch + 64, cnt * sizeof(double), hipMemcpyDeviceToHost, c->stream));
  HIP_TRY(hipStreamSynchronize(c->stream));
  return 0;
}

int m4q_device_alloc(size_t bytes, int32_t device, void** out) {
  if (!out || bytes == 0) return fail(M4Q_E_BADARG, "m4q_device_alloc: bad argument");
  int rc = need_device();
  if (rc) return rc;
  if (device >= 0) HIP_TRY(hipSetDevice(device));
  void* p = nullptr;
  HIP_TRY(hipMalloc(&p, bytes));
  hipError_t e = hipMemset(p, 0, bytes);
  if (e != hipSuccess) { (void)hipFree(p); return fail(-(int)e, "hipMemset: %s", hipGetErrorString(e)); }
  *out = p;
  return 0;
}

int m4q_device_free(void* dev) {
  if (dev) HIP_TRY(hipFree(dev));
  return 0;
}

int m4q_device_read(void* host, const void* dev, size_t bytes) {
  if (!host || !dev) return fail(M4Q_E_BADARG, "m4q_device_read: bad argument");
  HIP_TRY(hipMemcpy(host, dev, bytes, hipMemcpyDeviceToHost));
  return 0;
}

int m4q_device_write(void* dev, const void* host, size_t bytes) {
  if (!host || !dev) return fail(M4Q_E_BADARG, "m4q_device_write: bad argument");
  HIP_TRY(hipMemcpy(dev, host, bytes, hipMemcpyHostToDevice));
  return 0;
}

}  // extern "C"
