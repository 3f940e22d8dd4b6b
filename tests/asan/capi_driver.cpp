// Address-sanitised drive of the host side of the C ABI (include/m4q.h) - SURVEY.md section 5 "Race detection / sanitizers".
// Built by tests/test_asan_host.py:  m4q_capi.hip's host pass with -fsanitize=address, linked with the product's kernel
// objects and this driver.  Every entry point is called on its argument-validation paths and - this box has no GPU - on its
// no-device path; with a GPU present (M4Q_ASAN_DEVICE=1) the session field tables, bind_output, put_state / get_state and
// create / destroy cycles run for real.  Any heap overflow, use after free or double free in the pointer / size handling of
// m4q_capi.hip aborts the process with an ASan report; the driver itself checks every return code.
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "../../include/m4q.h"

static int failures = 0;
#define EXPECT(cond)                                                          \
  do {                                                                        \
    if (!(cond)) { ++failures; std::printf("FAIL %s:%d  %s   [%s]\n", __FILE__, __LINE__, #cond, m4q_last_error()); } \
  } while (0)

static m4q_problem problem(int n, int m, int order, int T, int ns) {
  m4q_problem p;
  std::memset(&p, 0, sizeof(p));
  p.dim_x = n; p.dim_u = m; p.order = order; p.horizon = T; p.n_steps = ns; p.max_iter = 100; p.warm_start = 1;
  p.qp_flags = M4Q_QP_DU_BAND; p.plant_kind = M4Q_PLANT_HAMILTONIAN; p.target_cols = ns + T + 1;
  p.dt = 0.25; p.sat = 1.0; p.du = 0.5; p.ls_tol = 1e-4;
  return p;
}

int main() {
  const bool have_dev = m4q_device_count() > 0;
  std::printf("devices: %d\n", m4q_device_count());
  // ---- pure host entry points
  EXPECT(std::strlen(m4q_version()) > 0);
  EXPECT(m4q_supported(9, 2, 1) == 1 && m4q_supported(9, 7, 1) == 0);
  EXPECT(m4q_library_size(1, 2) == 2 && m4q_library_size(2, 2) == 5 && m4q_library_size(2, 3) == 9);
  EXPECT(m4q_library_size(-1, 2) == M4Q_E_BADARG && m4q_library_size(1, 0) == M4Q_E_BADARG);
  {
    std::vector<int32_t> tab(6 * 2, -1);
    EXPECT(m4q_power_list(2, 2, tab.data()) == 6);                 // (P + 1) x dim_u entries, none past the end (ASan)
    EXPECT(tab[0] == 0 && tab[1] == 0 && tab[2] == 1 && tab[3] == 0);
    EXPECT(m4q_power_list(5, 2, tab.data()) == M4Q_E_UNSUPPORTED);
  }
  // ---- session creation: every validation branch, then the device / no-device branch
  m4q_session* s = nullptr;
  m4q_problem p = problem(9, 2, 1, 8, 4);
  EXPECT(m4q_session_create(nullptr, 4, -1, &s) == M4Q_E_BADARG);
  EXPECT(m4q_session_create(&p, 0, -1, &s) == M4Q_E_BADARG);
  EXPECT(m4q_session_create(&p, 4, -1, nullptr) == M4Q_E_BADARG);
  { m4q_problem q = p; q.dim_u = 7; EXPECT(m4q_session_create(&q, 4, -1, &s) == M4Q_E_UNSUPPORTED); }
  { m4q_problem q = p; q.horizon = 0; EXPECT(m4q_session_create(&q, 4, -1, &s) == M4Q_E_BADARG); }
  { m4q_problem q = p; q.target_cols = 3; EXPECT(m4q_session_create(&q, 4, -1, &s) == M4Q_E_BADARG); }
  { m4q_problem q = p; q.sat = 0.0; EXPECT(m4q_session_create(&q, 4, -1, &s) == M4Q_E_BADARG); }
  { m4q_problem q = p; q.qp_flags = 256; EXPECT(m4q_session_create(&q, 4, -1, &s) == M4Q_E_BADARG); }        // internal bit refused
  { m4q_problem q = p; q.qp_flags = 64; EXPECT(m4q_session_create(&q, 4, -1, &s) == M4Q_E_BADARG); }
  { m4q_problem q = p; q.qp_flags = M4Q_QP_EXACT_BOX | M4Q_QP_REF_LQR; EXPECT(m4q_session_create(&q, 4, -1, &s) == M4Q_E_BADARG); }
  { m4q_problem q = p; q.target_per_instance = 1; q.target_cols = 100000; q.n_steps = 4;
    EXPECT(m4q_session_create(&q, 1 << 20, -1, &s) == M4Q_E_BADARG); }                                       // 4 GiB offset limit
  // ---- null-session paths of every session entry point
  double ms = 0; int32_t nl = 0; int64_t st6[6]; int64_t hb = 0; int32_t gr = 0, ld = 0; char byte[16] = {0};
  EXPECT(m4q_session_field_bytes(nullptr, 0) == 0);
  EXPECT(m4q_session_upload(nullptr, 0, byte, 16) == M4Q_E_BADARG);
  EXPECT(m4q_session_download(nullptr, 0, byte, 16) == M4Q_E_BADARG);
  EXPECT(m4q_session_put_state(nullptr, 0, byte) == M4Q_E_BADARG);
  EXPECT(m4q_session_get_state(nullptr, 0, byte) == M4Q_E_BADARG);
  EXPECT(m4q_session_device_ptr(nullptr, 0) == nullptr);
  EXPECT(m4q_session_bind_output(nullptr, M4Q_F_XS, byte, 16) == M4Q_E_BADARG);
  EXPECT(m4q_session_run(nullptr, 0, 1) == M4Q_E_BADARG);
  EXPECT(m4q_session_sync(nullptr) == M4Q_E_BADARG);
  EXPECT(m4q_session_set_codes(nullptr, nullptr) == M4Q_E_BADARG);
  EXPECT(m4q_session_kernel_ms(nullptr, &ms, &nl) == M4Q_E_BADARG);
  EXPECT(m4q_session_qp_stats(nullptr, st6) == M4Q_E_BADARG);
  EXPECT(m4q_session_info(nullptr, &hb, &gr, &ld) == M4Q_E_BADARG);
  EXPECT(m4q_session_path(nullptr) == M4Q_E_BADARG);
  EXPECT(m4q_session_build_models(nullptr, 0.25, nullptr, 0, nullptr) == M4Q_E_BADARG);
  EXPECT(m4q_session_copy_final_state(nullptr, byte) == M4Q_E_BADARG);
  EXPECT(m4q_session_copy_status(nullptr, byte) == M4Q_E_BADARG);
  m4q_session_destroy(nullptr);
  // ---- one-shot entry points: argument validation first, then the device check
  std::vector<double> buf(1 << 16, 0.0);
  double* b = buf.data();
  EXPECT(m4q_linearize_batch(1, 9, 7, 1, 4, b, 0, b, b, b, b, b) == M4Q_E_UNSUPPORTED);
  EXPECT(m4q_linearize_batch(0, 9, 2, 1, 4, b, 0, b, b, b, b, b) == M4Q_E_BADARG);
  EXPECT(m4q_linearize_batch(1, 9, 2, 1, 4, nullptr, 0, b, b, b, b, b) == M4Q_E_BADARG);
  EXPECT(m4q_quad_program_batch(1, 9, 2, 4, 0, 0.0, 0.1, b, b, b, 0, b, b, b, b, b, b, b, b, b, b) == M4Q_E_BADARG);      // sat
  EXPECT(m4q_quad_program_batch(1, 9, 2, 4, 512, 1.0, 0.1, b, b, b, 0, b, b, b, b, b, b, b, b, b, b) == M4Q_E_BADARG);    // flags
  EXPECT(m4q_quad_program_batch(1, 9, 2, 4, 5, 1.0, 0.1, b, b, b, 0, b, b, b, b, b, b, b, b, b, b) == M4Q_E_BADARG);      // exact + ref
  EXPECT(m4q_quad_program_batch(1, 9, 2, 0, 0, 1.0, 0.1, b, b, b, 0, b, b, b, b, b, b, b, b, b, b) == M4Q_E_BADARG);      // T
  EXPECT(m4q_discretize_batch(0, 9, 2, 1, 0.25, b, 0, nullptr, b) == M4Q_E_BADARG);
  EXPECT(m4q_discretize_batch(1, 9, 2, 3, 0.25, b, 0, nullptr, b) == M4Q_E_UNSUPPORTED);
  EXPECT(m4q_plant_step_batch(1, 9, 2, 7, 0.25, b, b, b, b, 0, b) == M4Q_E_BADARG);
  EXPECT(m4q_plant_step_batch(1, 9, 2, M4Q_PLANT_HAMILTONIAN, 0.25, nullptr, b, b, b, 0, b) == M4Q_E_BADARG);
  EXPECT(m4q_mpc_batch(nullptr, 1, b, b, b, b, b, b, b, b, b, b, b, nullptr, nullptr, nullptr) == M4Q_E_BADARG);
  { m4q_problem q = p; q.plant_kind = M4Q_PLANT_NONE;
    EXPECT(m4q_mpc_batch(&q, 1, b, b, b, b, b, b, b, b, b, b, b, nullptr, nullptr, nullptr) == M4Q_E_BADARG); }
  EXPECT(m4q_mpc_batch(&p, 1, b, b, b, b, b, b, b, nullptr, b, b, b, nullptr, nullptr, nullptr) == M4Q_E_BADARG);
  // ---- communicator and device-memory entry points
  void* dp = nullptr; m4q_comm* cm = nullptr; char id[M4Q_UNIQUE_ID_BYTES] = {0};
  EXPECT(m4q_comm_unique_id(nullptr) == M4Q_E_BADARG);
  EXPECT(m4q_comm_create(0, 0, id, -1, &cm) == M4Q_E_BADARG);
  EXPECT(m4q_comm_create(2, 2, id, -1, &cm) == M4Q_E_BADARG);
  EXPECT(m4q_comm_create(0, 1, nullptr, -1, &cm) == M4Q_E_BADARG);
  EXPECT(m4q_comm_gather(nullptr, nullptr, byte, byte, 16, 0, 0) == M4Q_E_BADARG);
  EXPECT(m4q_comm_wait(nullptr, 0) == M4Q_E_BADARG);
  EXPECT(m4q_comm_allreduce_f64(nullptr, b, 1, 0) == M4Q_E_BADARG);
  m4q_comm_destroy(nullptr);
  EXPECT(m4q_device_alloc(0, -1, &dp) == M4Q_E_BADARG);
  EXPECT(m4q_device_alloc(16, -1, nullptr) == M4Q_E_BADARG);
  EXPECT(m4q_device_read(nullptr, byte, 1) == M4Q_E_BADARG && m4q_device_write(byte, nullptr, 1) == M4Q_E_BADARG);
  EXPECT(m4q_device_free(nullptr) == 0);

  if (!have_dev) {
    // no-device path of everything that needs one: a clean M4Q_E_NODEVICE (or the runtime's own error), nothing allocated or leaked
    EXPECT(m4q_session_create(&p, 4, -1, &s) == M4Q_E_NODEVICE && s == nullptr);
    EXPECT(m4q_linearize_batch(1, 9, 2, 1, 4, b, 0, b, b, b, b, b) == M4Q_E_NODEVICE);
    EXPECT(m4q_quad_program_batch(1, 9, 2, 4, 0, 1.0, 0.1, b, b, b, 0, b, b, b, b, b, b, b, b, b, b) == M4Q_E_NODEVICE);
    EXPECT(m4q_discretize_batch(1, 9, 2, 1, 0.25, b, 0, nullptr, b) == M4Q_E_NODEVICE);
    EXPECT(m4q_plant_step_batch(1, 9, 2, M4Q_PLANT_HAMILTONIAN, 0.25, b, b, b, b, 0, b) == M4Q_E_NODEVICE);
    EXPECT(m4q_mpc_batch(&p, 1, b, b, b, b, b, b, b, b, b, b, b, nullptr, nullptr, nullptr) == M4Q_E_NODEVICE);
    EXPECT(m4q_device_alloc(16, -1, &dp) == M4Q_E_NODEVICE);
    EXPECT(m4q_comm_create(0, 1, id, -1, &cm) == M4Q_E_NODEVICE);
  } else {
    // with a device: field tables, size checks, bind_output, put/get_state ranges, create/destroy cycles
    for (int cyc = 0; cyc < 3; ++cyc) {
      EXPECT(m4q_session_create(&p, 5, -1, &s) == 0 && s != nullptr);
      for (int f = -1; f <= M4Q_F_COUNT; ++f) {
        const size_t nb = m4q_session_field_bytes(s, f);
        EXPECT((f < 0 || f >= M4Q_F_COUNT) ? nb == 0 : nb > 0);
        if (f >= 0 && f < M4Q_F_COUNT) {
          std::vector<char> h(nb + 8);
          EXPECT(m4q_session_upload(s, f, h.data(), nb + 8) == M4Q_E_BADARG);
          EXPECT(m4q_session_download(s, f, h.data(), nb - 1) == M4Q_E_BADARG);
          EXPECT(m4q_session_download(s, f, h.data(), nb) == 0);
        }
      }
      std::vector<char> col(5 * 9 * 16);
      EXPECT(m4q_session_put_state(s, -1, col.data()) == M4Q_E_BADARG && m4q_session_put_state(s, 5, col.data()) == M4Q_E_BADARG);
      EXPECT(m4q_session_get_state(s, 5, col.data()) == M4Q_E_BADARG);
      EXPECT(m4q_session_put_state(s, 4, col.data()) == 0 && m4q_session_get_state(s, 4, col.data()) == 0);
      EXPECT(m4q_session_bind_output(s, M4Q_F_MODELS, col.data(), 16) == M4Q_E_BADARG);      // inputs cannot be bound
      EXPECT(m4q_device_alloc(m4q_session_field_bytes(s, M4Q_F_US), -1, &dp) == 0);
      EXPECT(m4q_session_bind_output(s, M4Q_F_US, dp, 8) == M4Q_E_BADARG);
      EXPECT(m4q_session_bind_output(s, M4Q_F_US, dp, m4q_session_field_bytes(s, M4Q_F_US)) == 0);
      EXPECT(m4q_session_run(s, 2, 2) == M4Q_E_BADARG && m4q_session_run(s, 0, 5) == M4Q_E_BADARG);
      EXPECT(m4q_session_run(s, 0, 4) == M4Q_E_BADARG);                                        // Q, Qf, R not uploaded yet
      m4q_session_destroy(s);
      EXPECT(m4q_device_free(dp) == 0);
      s = nullptr;
    }
  }
  std::printf(failures ? "asan driver: %d FAILURES\n" : "asan driver: all checks passed\n", failures);
  return failures ? 1 : 0;
}
