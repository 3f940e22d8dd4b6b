// Microbenchmark: fp64 MFMA on gfx950 against the fp64 VALU FMA the kernels use, and whether the two overlap.
//   hipcc --offload-arch=gfx950 -O3 tools/ubench_mfma.hip -o tools/bin/ubench_mfma && ./tools/bin/ubench_mfma
// One workgroup = one wave; the grid puts `w` waves on every SIMD of the chip.  Questions:
//   1. cycles per v_mfma_f64_16x16x4_f64 / v_mfma_f64_4x4x4_4b_f64 on one SIMD, independent and dependent accumulators;
//   2. does fp64 VALU work issued between MFMAs of the SAME wave hide under them (separate pipes) or add;
//   3. do an MFMA-only wave and a VALU-only wave on the same SIMD run concurrently (sum of rates) or share one DP pipe.
// FLOP: 16x16x4 = 2048 per instruction, 4x4x4_4b = 512, a wave-wide v_fma_f64 = 128.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

typedef double v4d __attribute__((ext_vector_type(4)));

#define R8(x) x x x x x x x x

// MODE 0: 16x16x4, 4 independent accumulators      MODE 1: 16x16x4, one accumulator (dependent chain)
// MODE 2: 4x4x4_4b, 8 independent accumulators     MODE 3: 4x4x4_4b, one accumulator (dependent chain)
// MODE 4: v_fma_f64 only (8 independent)
// MODE 5: per 16x16x4 MFMA, NV v_fma_f64 in the same wave (NV = template arg)
// MODE 6: per 4x4x4_4b MFMA, NV v_fma_f64 in the same wave
// MODE 7: waves with even blockIdx run MODE 0's loop, odd ones MODE 4's (pairs share a SIMD at >= 2 waves/SIMD)
template <int MODE, int NV>
__global__ __launch_bounds__(64) void k(double* out, int iters, int* role_count) {
  const double a = 1.0 + threadIdx.x * 1e-9, b = 1.0 - threadIdx.x * 1e-9;
  v4d c0 = {0, 0, 0, 0}, c1 = c0, c2 = c0, c3 = c0;
  double s0 = 0, s1 = 1, s2 = 2, s3 = 3, s4 = 4, s5 = 5, s6 = 6, s7 = 7;
  double x = 1.0000001, y = 0.9999999;
  int mode = MODE;
  if (MODE == 7) mode = (blockIdx.x & 1) ? 4 : 0;
  int reps = 1;
  if (MODE == 8) {
    // role by the wave's slot on its SIMD (HW_REG_HW_ID bits 3:0 = WAVE_ID): even slots MFMA, odd slots VALU with 16x the
    // instructions, so that both roles need about the same number of DP-pipe cycles per iteration
    const unsigned hw = __builtin_amdgcn_s_getreg((5 << 11) | (0 << 6) | 4);
    mode = (hw & 1) ? 4 : 0;
    reps = (hw & 1) ? 16 : 1;
  }
  for (int i = 0; i < iters; ++i) {
    if (mode == 0) {
      R8(c0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c0, 0, 0, 0); c1 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c1, 0, 0, 0);
         c2 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c2, 0, 0, 0); c3 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c3, 0, 0, 0);)
    } else if (mode == 1) {
      R8(c0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c0, 0, 0, 0); c0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c0, 0, 0, 0);
         c0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c0, 0, 0, 0); c0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c0, 0, 0, 0);)
    } else if (mode == 2) {
      R8(s0 = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, s0, 0, 0, 0); s1 = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, s1, 0, 0, 0);
         s2 = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, s2, 0, 0, 0); s3 = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, s3, 0, 0, 0);)
    } else if (mode == 3) {
      R8(s0 = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, s0, 0, 0, 0); s0 = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, s0, 0, 0, 0);
         s0 = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, s0, 0, 0, 0); s0 = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, s0, 0, 0, 0);)
    } else if (mode == 4) {
      for (int r = 0; r < reps; ++r)
        asm volatile(R8("v_fma_f64 %0, %8, %9, %0\n\tv_fma_f64 %1, %8, %9, %1\n\tv_fma_f64 %2, %8, %9, %2\n\tv_fma_f64 %3, %8, %9, %3\n\t")
                     : "+v"(s0), "+v"(s1), "+v"(s2), "+v"(s3), "+v"(s4), "+v"(s5), "+v"(s6), "+v"(s7) : "v"(x), "v"(y));
    } else if (mode == 5 || mode == 6) {
#pragma unroll
      for (int r = 0; r < 8; ++r) {
        if (mode == 5) {
          if (r & 1) c1 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c1, 0, 0, 0);
          else c0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c0, 0, 0, 0);
        } else {
          if (r & 1) s6 = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, s6, 0, 0, 0);
          else s7 = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, s7, 0, 0, 0);
        }
#pragma unroll
        for (int v = 0; v < NV; ++v) {
          if ((v & 3) == 0) asm volatile("v_fma_f64 %0, %1, %2, %0" : "+v"(s0) : "v"(x), "v"(y));
          if ((v & 3) == 1) asm volatile("v_fma_f64 %0, %1, %2, %0" : "+v"(s1) : "v"(x), "v"(y));
          if ((v & 3) == 2) asm volatile("v_fma_f64 %0, %1, %2, %0" : "+v"(s2) : "v"(x), "v"(y));
          if ((v & 3) == 3) asm volatile("v_fma_f64 %0, %1, %2, %0" : "+v"(s3) : "v"(x), "v"(y));
        }
      }
    }
  }
  if (MODE == 7 && threadIdx.x == 0) atomicAdd(role_count + (blockIdx.x & 1), 1);
  if (MODE == 8 && threadIdx.x == 0) atomicAdd(role_count + (mode == 4 ? 1 : 0), 1);
  out[blockIdx.x * 64 + threadIdx.x] = c0[0] + c1[1] + c2[2] + c3[3] + s0 + s1 + s2 + s3 + s4 + s5 + s6 + s7;
}

template <int MODE, int NV>
void run(const char* name, int waves_per_simd, double mfma_per_iter, double mfma_flop, double valu_per_iter) {
  const int grid = 256 * 4 * waves_per_simd, iters = 1000;
  double* d;
  int* rc;
  hipMalloc(&d, (size_t)grid * 64 * 8);
  hipMalloc(&rc, 8);
  hipMemset(rc, 0, 8);
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL((k<MODE, NV>), dim3(grid), dim3(64), 0, 0, d, iters, rc);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  hipLaunchKernelGGL((k<MODE, NV>), dim3(grid), dim3(64), 0, 0, d, iters, rc);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  double nm = mfma_per_iter * iters * grid, nv = valu_per_iter * iters * grid;
  if (MODE == 7) { nm *= 0.5; nv *= 0.5; }
  if (MODE == 8) {
    int h[2];
    hipMemcpy(h, rc, 8, hipMemcpyDeviceToHost);     // both launches counted: halve
    nm = mfma_per_iter * iters * (h[0] / 2.0);
    nv = valu_per_iter * 16 * iters * (h[1] / 2.0);
    printf("   (waves by role, per launch: MFMA %d, VALU %d)\n", h[0] / 2, h[1] / 2);
  }
  const double simd_cycles = 1024.0 * ms * 1e-3 * 2.4e9;
  printf("%-52s w/SIMD %d : %8.3f ms  MFMA %6.2f TF  VALU %6.2f TF  cyc/MFMA/SIMD %6.1f  cyc/iter/wave %7.1f\n", name, waves_per_simd, ms,
         nm * mfma_flop / ms / 1e9, nv * 128 / ms / 1e9, nm > 0 ? simd_cycles / nm : 0.0,
         ms * 1e-3 * 2.4e9 / iters);
  hipFree(d);
  hipFree(rc);
}

int main() {
  for (int w : {1, 2, 4}) {
    run<0, 0>("mfma_f64_16x16x4 x4 independent", w, 32, 2048, 0);
    run<1, 0>("mfma_f64_16x16x4 dependent chain", w, 32, 2048, 0);
    run<2, 0>("mfma_f64_4x4x4_4b x4 independent", w, 32, 512, 0);
    run<3, 0>("mfma_f64_4x4x4_4b dependent chain", w, 32, 512, 0);
    run<4, 0>("v_fma_f64 x4 independent", w, 0, 0, 32);
    run<5, 0>("16x16x4 + 0 v_fma_f64 per MFMA (same wave)", w, 8, 2048, 0);
    run<5, 4>("16x16x4 + 4 v_fma_f64 per MFMA (same wave)", w, 8, 2048, 32);
    run<5, 8>("16x16x4 + 8 v_fma_f64 per MFMA (same wave)", w, 8, 2048, 64);
    run<5, 12>("16x16x4 + 12 v_fma_f64 per MFMA (same wave)", w, 8, 2048, 96);
    run<5, 16>("16x16x4 + 16 v_fma_f64 per MFMA (same wave)", w, 8, 2048, 128);
    run<6, 0>("4x4x4_4b + 0 v_fma_f64 per MFMA (same wave)", w, 8, 512, 0);
    run<6, 2>("4x4x4_4b + 2 v_fma_f64 per MFMA (same wave)", w, 8, 512, 16);
    run<6, 4>("4x4x4_4b + 4 v_fma_f64 per MFMA (same wave)", w, 8, 512, 32);
    run<6, 8>("4x4x4_4b + 8 v_fma_f64 per MFMA (same wave)", w, 8, 512, 64);
    if (w >= 2) run<7, 0>("MFMA-only waves beside VALU-only waves (by blockIdx)", w, 32, 2048, 32);
    if (w >= 2) run<8, 0>("MFMA waves beside 16x VALU waves (by SIMD slot)", w, 32, 2048, 32);
  }
  return 0;
}
