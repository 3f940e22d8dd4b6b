// Lane maps of v_mfma_f64_4x4x4_4b_f64 on gfx950, found by experiment (the guides give the 16x16x4 map only).
//   hipcc --offload-arch=gfx950 -O2 tools/mfma_layout.hip -o tools/bin/mfma_layout && ./tools/bin/mfma_layout
// One wavefront.  Probe 1: A one-hot in lane la, B = 1 + lane  ->  D[lane] = the B element that met A's one.
//                 Probe 2: B one-hot in lane lb, A = 1 + lane.
// From the two tables: which (block, i, k) an A lane holds, which (block, k, j) a B lane holds, which (block, i, j) a D lane holds.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

__global__ __launch_bounds__(64) void probe(double* out) {
  const int lane = threadIdx.x;
  for (int hot = 0; hot < 64; ++hot) {
    const double a1 = lane == hot ? 1.0 : 0.0, b1 = 1.0 + lane;
    out[hot * 64 + lane] = __builtin_amdgcn_mfma_f64_4x4x4f64(a1, b1, 0.0, 0, 0, 0);
    const double a2 = 1.0 + lane, b2 = lane == hot ? 1.0 : 0.0;
    out[4096 + hot * 64 + lane] = __builtin_amdgcn_mfma_f64_4x4x4f64(a2, b2, 0.0, 0, 0, 0);
  }
}

int main() {
  double* d;
  hipMalloc(&d, 8192 * sizeof(double));
  hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, d);
  std::vector<double> h(8192);
  hipMemcpy(h.data(), d, 8192 * sizeof(double), hipMemcpyDeviceToHost);
  // A one-hot at la: D lanes that are non-zero = outputs (i fixed by la, every j); value - 1 = the B lane that supplied B[k][j]
  printf("# probe 1: A one-hot lane -> list of (D lane : B lane)\n");
  for (int la = 0; la < 64; ++la) {
    printf("A%02d:", la);
    for (int l = 0; l < 64; ++l)
      if (h[la * 64 + l] != 0.0) printf(" D%02d<-B%02d", l, (int)h[la * 64 + l] - 1);
    printf("\n");
  }
  printf("# probe 2: B one-hot lane -> list of (D lane : A lane)\n");
  for (int lb = 0; lb < 64; ++lb) {
    printf("B%02d:", lb);
    for (int l = 0; l < 64; ++l)
      if (h[4096 + lb * 64 + l] != 0.0) printf(" D%02d<-A%02d", l, (int)h[4096 + lb * 64 + l] - 1);
    printf("\n");
  }
  return 0;
}
