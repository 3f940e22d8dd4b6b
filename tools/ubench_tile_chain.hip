// Microbenchmark for VERDICT r3 item 3: the backward Riccati sweep of BASELINE config 3's shape (8 traceless coordinates, 2
// controls, order 1, T = 40, constant target) in isolation - no state machine, no rollout - in three forms:
//   dpp     riccati_backward<double, 8, 2, ..., TC> on DPP rows, 4 members per wavefront (what the product's headline kernel runs)
//   tile G  tools/ubench_tile2.h: the same sweep on v_mfma_f64_4x4x4_4b_f64 tiles with G groups of 4 members interleaved per wavefront
// at W wavefronts per SIMD (residency forced with an LDS pad).  Prints SIMD-nanoseconds per member-index (lower = better; the
// whole chip: 1024 SIMDs) and checks the tile forms' gains against the DPP form's.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I mpc4quantum_amd/csrc -I tools tools/ubench_tile_chain.hip -o tools/bin/ubench_tile_chain
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "m4q_tile3.h"
#include "ubench_tile2.h"

using namespace m4q;
constexpr int NS = 8, NU = 2, ORDER = 1, NP = 2, PITCH = ModelPitch<NS>::value;
constexpr int MODEL_K = (1 + NP) * NS * PITCH;

struct Args {
  const double* models;   // [M][1+NP][NS][NS] row-major
  const double* costs;    // Q [64] Qf [64] R [4]
  const double* Xg;       // [M][T+1][NS]
  const double* Ug;       // [M][T][NU]
  const double* xbm;      // [T+1][NS] (constant target), shared
  const double* ubm;      // [T][NU], shared
  double* gains;          // [M][T][NS+1][NU]
  unsigned long long* role_ticks;   // [2][2]: per role (0 tile / 1 dpp): sum of the wavefronts' 100 MHz ticks inside their sweeps, wavefront count
  int T, reps, members;   // members: total, a multiple of the members per wavefront
};

extern __shared__ __align__(16) unsigned char lds_raw[];

// MIX: wavefronts with an odd blockIdx run the DPP sweep, the even ones the tile sweep (G = 1): do the two forms disturb each other
// when they share a SIMD, as a tile sweep and the DPP phases of the other wavefront do in the product's hybrid kernel?
template <int G, bool TILE, int WAVES, bool STORE = true, bool MIX = false, bool BATCH = false>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(WAVES, WAVES))) void k(Args a) {
  constexpr int MW = TILE ? 4 * G : 4;                       // members per wavefront
  double* lds = reinterpret_cast<double*>(lds_raw);
  double* ldsQ = lds + MW * MODEL_K;
  volatile __attribute__((address_space(3))) double* gb = (volatile __attribute__((address_space(3))) double*)(ldsQ + 2 * NS * NS + NU * NU);
  const int lane = threadIdx.x;
  const long m0 = ((long)blockIdx.x * MW) % a.members;
  for (int e = lane; e < MW * (1 + NP) * NS * NS; e += 64) {
    const int mi = e / ((1 + NP) * NS * NS), r = e % ((1 + NP) * NS * NS);
    const int p = r / (NS * NS), i = (r / NS) % NS, kx = r % NS;
    lds[mi * MODEL_K + ModelPitch<NS>::at(p, i, kx)] = a.models[(m0 + mi) * (1 + NP) * NS * NS + r];
  }
  for (int e = lane; e < 2 * NS * NS + NU * NU; e += 64) ldsQ[e] = a.costs[e];
  __syncthreads();
  const int T = a.T;
  const unsigned long long tick0 = __builtin_amdgcn_s_memrealtime();
  const int role = (TILE && !(MIX && (blockIdx.x & 1))) ? 0 : 1;
  const unsigned sX = (unsigned)(T + 1) * NS, sU = (unsigned)T * NU, sG = (unsigned)T * (NS + 1) * NU;
  if constexpr (BATCH) {                                     // m4q_tile3.h: time-batched operands, one group
    TileBackwardB<NS, NU, ORDER> ts;
    ts.T = T;
    ts.Q = ldsQ; ts.Qf = ldsQ + NS * NS; ts.R = ldsQ + 2 * NS * NS;
    const int mi = ts.L.mb;
    ts.mdl = lds + mi * MODEL_K;
    ts.Xg = gview((const M4Q_GLOBAL double*)a.Xg, m0 * sX, mi * sX);
    ts.Ug = gview((const M4Q_GLOBAL double*)a.Ug, m0 * sU, mi * sU);
    ts.gains = gview((const M4Q_GLOBAL double*)a.gains, m0 * sG, mi * sG);
    ts.xbm = gview((const M4Q_GLOBAL double*)a.xbm, 0, 0);
    ts.ubm = gview((const M4Q_GLOBAL double*)a.ubm, 0, 0);
    ts.gb = gb + mi * 16;
    for (int r = 0; r < a.reps; ++r) {
      asm volatile("" ::: "memory");
      ts.backward(STORE);
    }
  } else if (TILE && !(MIX && (blockIdx.x & 1))) {
    TileBackwardG<NS, NU, ORDER, G> ts;
    ts.T = T;
    ts.Q = ldsQ; ts.Qf = ldsQ + NS * NS; ts.R = ldsQ + 2 * NS * NS;
#pragma unroll
    for (int g = 0; g < G; ++g) {
      const int mi = 4 * g + ts.L.mb;
      ts.mem[g].mdl = lds + mi * MODEL_K;
      ts.mem[g].Xg = gview((const M4Q_GLOBAL double*)a.Xg, m0 * sX, mi * sX);
      ts.mem[g].Ug = gview((const M4Q_GLOBAL double*)a.Ug, m0 * sU, mi * sU);
      ts.mem[g].gains = gview((const M4Q_GLOBAL double*)a.gains, m0 * sG, mi * sG);
      ts.mem[g].xbm = gview((const M4Q_GLOBAL double*)a.xbm, 0, 0);
      ts.mem[g].ubm = gview((const M4Q_GLOBAL double*)a.ubm, 0, 0);
      ts.mem[g].gb = gb + mi * 16;
    }
    for (int r = 0; r < a.reps; ++r) {
      asm volatile("" ::: "memory");
      ts.backward(STORE);
    }
  } else {
    const int g = lane >> 4, jj = lane & 15;
    const int j = jj < NS ? jj : NS - 1;
    FusedProv<double, NS, NU, ORDER> prov;
    prov.mdl = lds + g * MODEL_K;
    prov.Xg = gview((const M4Q_GLOBAL double*)a.Xg, m0 * sX, g * sX);
    prov.Ug = gview((const M4Q_GLOBAL double*)a.Ug, m0 * sU, g * sU);
    prov.j = j;
    Window win;
    win.xbm = gview((const M4Q_GLOBAL double*)a.xbm, 0, 0);
    win.ubm = gview((const M4Q_GLOBAL double*)a.ubm, 0, 0);
    CostRef<double> cost;
    cost.Q = ldsQ; cost.Qf = ldsQ + NS * NS; cost.q_stride = 0; cost.R = ldsQ + 2 * NS * NS; cost.r_stride = 0;
    const GView gains = gview((const M4Q_GLOBAL double*)a.gains, m0 * sG, g * sG);
    for (int r = 0; r < a.reps; ++r) {
      asm volatile("" ::: "memory");
      if (jj < NS) riccati_backward<double, NS, NU, FusedProv<double, NS, NU, ORDER>, false, true>(prov, T, win, cost, QP_TARG_CONST, gains, j, STORE);
    }
  }
  if (threadIdx.x == 0) {
    atomicAdd(a.role_ticks + 2 * role, __builtin_amdgcn_s_memrealtime() - tick0);
    atomicAdd(a.role_ticks + 2 * role + 1, 1ull);
  }
}

static double rnd(unsigned long long& h) {
  h ^= h >> 29; h *= 0xBF58476D1CE4E5B9ull; h ^= h >> 32; h += 0x9E3779B97F4A7C15ull;
  return (double)(h >> 11) * (1.0 / 9007199254740992.0) - 0.5;
}

template <int G, bool TILE, int WAVES, bool STORE = true, bool MIX = false, bool BATCH = false>
static double run(const char* name, Args a, std::vector<double>* out_gains, const std::vector<double>* ref) {
  constexpr int MW = TILE ? 4 * G : 4;
  const size_t lds_need = sizeof(double) * (size_t)(MW * MODEL_K + 2 * NS * NS + NU * NU + MW * 16);
  // residency: exactly 4 * WAVES workgroups per CU (160 KB of LDS per CU)
  size_t lds = 160 * 1024 / (4 * WAVES) - 512;
  if (lds < lds_need) { printf("%-28s needs %zu B of LDS, %zu allowed at %d waves/SIMD: skipped\n", name, lds_need, lds, WAVES); return 0; }
  hipFuncSetAttribute(reinterpret_cast<const void*>(k<G, TILE, WAVES, STORE, MIX, BATCH>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  int nb = 0;
  hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, k<G, TILE, WAVES, STORE, MIX, BATCH>, 64, lds);
  const int grid = 256 * 4 * WAVES * 2;                      // two rounds of the resident set
  hipMemset(a.gains, 0, sizeof(double) * (size_t)a.members * a.T * (NS + 1) * NU);
  hipLaunchKernelGGL((k<G, TILE, WAVES, STORE, MIX, BATCH>), dim3(grid), dim3(64), lds, 0, a);
  hipError_t e = hipDeviceSynchronize();
  hipMemset(a.role_ticks, 0, 32);
  if (e != hipSuccess) { printf("%s: %s\n", name, hipGetErrorString(e)); exit(1); }
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  float best = 1e30f;
  for (int it = 0; it < 3; ++it) {
    hipEventRecord(e0);
    hipLaunchKernelGGL((k<G, TILE, WAVES, STORE, MIX, BATCH>), dim3(grid), dim3(64), lds, 0, a);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    best = ms < best ? ms : best;
  }
  const double member_idx = (double)grid * MW * a.T * a.reps;
  const double simd_ns = best * 1e6 * 1024.0 / member_idx;
  std::vector<double> h((size_t)a.members * a.T * (NS + 1) * NU);
  hipMemcpy(h.data(), a.gains, h.size() * sizeof(double), hipMemcpyDeviceToHost);
  double err = 0, mag = 0;
  if (ref)
    for (size_t i = 0; i < h.size(); ++i) { err = fmax(err, fabs(h[i] - (*ref)[i])); mag = fmax(mag, fabs((*ref)[i])); }
  if (out_gains) *out_gains = h;
  printf("%-28s waves/SIMD %d (occupancy query: %d WG/CU)  %8.3f ms  %7.2f SIMD-ns per member-index", name, WAVES, nb, best, simd_ns);
  if (ref) printf("   max |gain - dpp gain| %.2e (of %.2e)", err, mag);
  unsigned long long rt[4];
  hipMemcpy(rt, a.role_ticks, 32, hipMemcpyDeviceToHost);
  // mean wavefront time inside its sweeps, per role, as ns per wavefront-index (3 timed launches summed)
  for (int r = 0; r < 2; ++r)
    if (rt[2 * r + 1]) printf("   [%s wavefronts: %.0f ns per wavefront-index]", r ? "dpp" : "tile", 10.0 * rt[2 * r] / rt[2 * r + 1] / (a.T * a.reps));
  printf("\n");
  return simd_ns;
}

int main(int argc, char** argv) {
  const int T = 40;
  const int reps = (argc > 2) ? atoi(argv[2]) : 8;          // sweeps per wavefront: 8 = 1 ms launches; 400 = 50 ms (the clock settles under load)
  // distinct members; wavefronts wrap around them.  Default 1,024: the operands (13.6 KB per member) stay in the L2.  `ubench_tile_chain big`:
  // 32,768 members = 445 MB of trajectories and gains - every wavefront of the largest launch has its own, as in the product, where
  // the per-row workspace comes from beyond the L2
  const int members = (argc > 1) ? 32768 : 16 * 64;
  unsigned long long h = 12345;
  std::vector<double> models((size_t)members * 3 * 64), costs(132, 0.0), Xg((size_t)members * (T + 1) * NS), Ug((size_t)members * T * NU),
      xbm((T + 1) * NS), ubm(T * NU, 0.0);
  for (int m = 0; m < members; ++m) {
    for (int p = 0; p < 3; ++p) {
      double A[8][8];
      for (int i = 0; i < 8; ++i) for (int kx = 0; kx < 8; ++kx) A[i][kx] = rnd(h);
      for (int i = 0; i < 8; ++i)
        for (int kx = 0; kx < 8; ++kx)
          models[((size_t)m * 3 + p) * 64 + i * 8 + kx] = (p == 0 ? (i == kx ? 1.0 : 0.0) : 0.0) + 0.3 * (A[i][kx] - A[kx][i]);
    }
    for (int e = 0; e < (T + 1) * NS; ++e) Xg[(size_t)m * (T + 1) * NS + e] = 0.6 * rnd(h);
    for (int e = 0; e < T * NU; ++e) Ug[(size_t)m * T * NU + e] = 0.4 * rnd(h);
  }
  for (int i = 0; i < 8; ++i) { costs[i * 8 + i] = (i == 0 || i == 3 || i == 7) ? 1.0 : 0.0; costs[64 + i * 8 + i] = costs[i * 8 + i]; }
  costs[128] = 0.05; costs[131] = 0.05;
  for (int t = 0; t <= T; ++t) for (int i = 0; i < NS; ++i) xbm[t * NS + i] = i == 3 ? 0.7 : (i == 7 ? -0.4 : 0.0);
  Args a;
  double *dm, *dc, *dx, *du, *dxb, *dub, *dg;
  hipMalloc(&dm, models.size() * 8); hipMalloc(&dc, costs.size() * 8); hipMalloc(&dx, Xg.size() * 8); hipMalloc(&du, Ug.size() * 8);
  hipMalloc(&dxb, xbm.size() * 8); hipMalloc(&dub, ubm.size() * 8); hipMalloc(&dg, sizeof(double) * (size_t)members * T * (NS + 1) * NU);
  hipMemcpy(dm, models.data(), models.size() * 8, hipMemcpyHostToDevice); hipMemcpy(dc, costs.data(), costs.size() * 8, hipMemcpyHostToDevice);
  hipMemcpy(dx, Xg.data(), Xg.size() * 8, hipMemcpyHostToDevice); hipMemcpy(du, Ug.data(), Ug.size() * 8, hipMemcpyHostToDevice);
  hipMemcpy(dxb, xbm.data(), xbm.size() * 8, hipMemcpyHostToDevice); hipMemcpy(dub, ubm.data(), ubm.size() * 8, hipMemcpyHostToDevice);
  unsigned long long* drt;
  hipMalloc(&drt, 32);
  a.role_ticks = drt;
  a.models = dm; a.costs = dc; a.Xg = dx; a.Ug = du; a.xbm = dxb; a.ubm = dub; a.gains = dg; a.T = T; a.reps = reps; a.members = members;
  std::vector<double> ref;
#if M4Q_T2_EXP
  printf("ablation build M4Q_T2_EXP = %d (timing only, gains wrong)\n", M4Q_T2_EXP);
  run<1, true, 2>("tile, G = 1, ablated", a, nullptr, nullptr);
#else
  run<1, false, 2>("dpp rows (product)", a, &ref, nullptr);
  run<1, false, 1>("dpp rows", a, nullptr, &ref);
  run<1, false, 2, false>("dpp rows, gains not stored", a, nullptr, nullptr);
  run<1, true, 2, false>("tile, G = 1, gains not stored", a, nullptr, nullptr);
  run<1, true, 2>("tile, G = 1 (m4q_tile.h form)", a, nullptr, &ref);
  run<1, true, 2, true, false, true>("tile, time-batched operands", a, nullptr, &ref);
  run<1, true, 1, true, false, true>("tile, time-batched operands", a, nullptr, &ref);
  run<1, true, 2, true, true>("half tile, half dpp wavefronts", a, nullptr, &ref);
  run<1, true, 1>("tile, G = 1", a, nullptr, &ref);
  run<2, true, 1>("tile, G = 2", a, nullptr, &ref);
  run<2, true, 2>("tile, G = 2", a, nullptr, &ref);
  run<3, true, 1>("tile, G = 3", a, nullptr, &ref);
  run<4, true, 1>("tile, G = 4", a, nullptr, &ref);
#endif
  return 0;
}
