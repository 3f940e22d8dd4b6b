// Microbenchmark: issue rate of fp64 FMA forms on gfx950 (one workgroup = one wave; grid fills the chip).
//   hipcc --offload-arch=gfx950 -O3 tools/ubench_dpp.hip -o gpurun_out/ubench_dpp && ./gpurun_out/ubench_dpp
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define DPP " row_newbcast:3 row_mask:0xf bank_mask:0xf\n\t"
#define R16(x) x x x x x x x x x x x x x x x x

// MODE 0: plain v_fma_f64, 8 independent accumulators
// MODE 1: v_fmac_f64_dpp, 8 independent accumulators
// MODE 2: v_fmac_f64_dpp, 2 accumulators alternating (dependent at distance 2)
// MODE 3: v_fmac_f64_dpp, 1 accumulator (dependent chain)
// MODE 4: as MODE 1 with an s_nop 1 in front of every 4
// MODE 5: as MODE 2 with an s_nop 1 in front of every 4 (the cmac block of the kernel: re, im, re, im)
// MODE 6: v_mov_b64_dpp + plain fma (unfused broadcast)
// lanes: only lanes (threadIdx.x & 15) < lanes of every 16-lane row take part (EXEC off for the others) - the `ubench_dpp lanes` mode
template <int MODE>
__global__ __launch_bounds__(64) void k(double* out, int iters, int lanes) {
  double a0 = threadIdx.x * 1e-9, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
  double x = 1.0000001, y = 0.9999999;
  if (lanes < 0) {
    // `lanes` mode with data that toggles: every lane its own full-mantissa operands (a hash of the lane and workgroup), so that the
    // multiplier arrays see different bits in neighbouring lanes and from one accumulator to the next
    lanes = -lanes;
    unsigned long long h = (threadIdx.x + 64ull * blockIdx.x + 1) * 0x9E3779B97F4A7C15ull;
    auto next = [&]() { h ^= h >> 29; h *= 0xBF58476D1CE4E5B9ull; h ^= h >> 32; return 0.5 + (double)(h >> 11) * (1.0 / 9007199254740992.0); };
    x = next(); y = -next() * 1e-9;
    a0 = next(); a1 = next(); a2 = next(); a3 = next(); a4 = next(); a5 = next(); a6 = next(); a7 = next();
  }
  if ((int)(threadIdx.x & 15) < lanes)
  for (int i = 0; i < iters; ++i) {
    if constexpr (MODE == 0) {
      asm volatile(R16("v_fma_f64 %0, %8, %9, %0\n\tv_fma_f64 %1, %8, %9, %1\n\tv_fma_f64 %2, %8, %9, %2\n\tv_fma_f64 %3, %8, %9, %3\n\t"
                       "v_fma_f64 %4, %8, %9, %4\n\tv_fma_f64 %5, %8, %9, %5\n\tv_fma_f64 %6, %8, %9, %6\n\tv_fma_f64 %7, %8, %9, %7\n\t")
                   : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(x), "v"(y));
    } else if constexpr (MODE == 1) {
      asm volatile(R16("v_fmac_f64_dpp %0, %8, %9" DPP "v_fmac_f64_dpp %1, %8, %9" DPP "v_fmac_f64_dpp %2, %8, %9" DPP "v_fmac_f64_dpp %3, %8, %9" DPP
                       "v_fmac_f64_dpp %4, %8, %9" DPP "v_fmac_f64_dpp %5, %8, %9" DPP "v_fmac_f64_dpp %6, %8, %9" DPP "v_fmac_f64_dpp %7, %8, %9" DPP)
                   : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(x), "v"(y));
    } else if constexpr (MODE == 2) {
      asm volatile(R16("v_fmac_f64_dpp %0, %8, %9" DPP "v_fmac_f64_dpp %1, %8, %9" DPP "v_fmac_f64_dpp %0, %8, %9" DPP "v_fmac_f64_dpp %1, %8, %9" DPP
                       "v_fmac_f64_dpp %0, %8, %9" DPP "v_fmac_f64_dpp %1, %8, %9" DPP "v_fmac_f64_dpp %0, %8, %9" DPP "v_fmac_f64_dpp %1, %8, %9" DPP)
                   : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(x), "v"(y));
    } else if constexpr (MODE == 3) {
      asm volatile(R16("v_fmac_f64_dpp %0, %8, %9" DPP "v_fmac_f64_dpp %0, %8, %9" DPP "v_fmac_f64_dpp %0, %8, %9" DPP "v_fmac_f64_dpp %0, %8, %9" DPP
                       "v_fmac_f64_dpp %0, %8, %9" DPP "v_fmac_f64_dpp %0, %8, %9" DPP "v_fmac_f64_dpp %0, %8, %9" DPP "v_fmac_f64_dpp %0, %8, %9" DPP)
                   : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(x), "v"(y));
    } else if constexpr (MODE == 4) {
      asm volatile(R16("s_nop 1\n\tv_fmac_f64_dpp %0, %8, %9" DPP "v_fmac_f64_dpp %1, %8, %9" DPP "v_fmac_f64_dpp %2, %8, %9" DPP "v_fmac_f64_dpp %3, %8, %9" DPP
                       "s_nop 1\n\tv_fmac_f64_dpp %4, %8, %9" DPP "v_fmac_f64_dpp %5, %8, %9" DPP "v_fmac_f64_dpp %6, %8, %9" DPP "v_fmac_f64_dpp %7, %8, %9" DPP)
                   : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(x), "v"(y));
    } else if constexpr (MODE == 5) {
      asm volatile(R16("s_nop 1\n\tv_fmac_f64_dpp %0, %8, %9" DPP "v_fmac_f64_dpp %1, %8, %9" DPP "v_fmac_f64_dpp %0, %8, %9" DPP "v_fmac_f64_dpp %1, %8, %9" DPP
                       "s_nop 1\n\tv_fmac_f64_dpp %2, %8, %9" DPP "v_fmac_f64_dpp %3, %8, %9" DPP "v_fmac_f64_dpp %2, %8, %9" DPP "v_fmac_f64_dpp %3, %8, %9" DPP)
                   : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(x), "v"(y));
    } else {
      double t0, t1, t2, t3;
      asm volatile(R16("v_mov_b64_dpp %8, %12" DPP "v_fma_f64 %0, %8, %13, %0\n\tv_fma_f64 %1, %8, %13, %1\n\t"
                       "v_mov_b64_dpp %9, %12" DPP "v_fma_f64 %2, %9, %13, %2\n\tv_fma_f64 %3, %9, %13, %3\n\t"
                       "v_mov_b64_dpp %10, %12" DPP "v_fma_f64 %4, %10, %13, %4\n\tv_fma_f64 %5, %10, %13, %5\n\t"
                       "v_mov_b64_dpp %11, %12" DPP "v_fma_f64 %6, %11, %13, %6\n\tv_fma_f64 %7, %11, %13, %7\n\t")
                   : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7), "=&v"(t0), "=&v"(t1), "=&v"(t2), "=&v"(t3)
                   : "v"(x), "v"(y));
    }
  }
  out[blockIdx.x * 64 + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
}

template <int MODE>
void run(const char* name, int waves_per_simd, int iters = 2000, int lanes = 16) {
  const int grid = 256 * 4 * waves_per_simd;
  double* d;
  hipMalloc(&d, (size_t)grid * 64 * 8);
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL(k<MODE>, dim3(grid), dim3(64), 0, 0, d, iters, lanes);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  hipLaunchKernelGGL(k<MODE>, dim3(grid), dim3(64), 0, 0, d, iters, lanes);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  const double fmas = (double)grid * iters * 16 * 8;                 // wave-level FMA instructions
  const double per_simd_cycle = fmas / (1024.0 * ms * 1e-3 * 2.4e9);  // FMA wave-instr per SIMD per cycle @2.4 GHz
  printf("%-44s waves/SIMD %d : %8.3f ms  %6.2f TFLOP/s  cycles/FMA/SIMD %.2f\n", name, waves_per_simd, ms, fmas * 128 / ms / 1e9,
         1.0 / per_simd_cycle);
  hipFree(d);
}

int main(int argc, char** argv) {
  if (argc > 1 && argv[1][0] == 'l' && argv[1][1] == 'a') {
    // `ubench_dpp lanes`: the sustained stream with 16, 9 and 4 of every 16 lanes enabled.  The instruction stream is the same; if
    // the launch gets shorter with fewer live lanes, the clock the chip holds under fp64 load follows the power of the lanes at work
    for (int rep = 0; rep < 2; ++rep)
      for (int lanes : {16, 9, 4}) {
        char name[96];
        snprintf(name, sizeof name, "SUSTAINED v_fmac_f64_dpp x8, %2d of 16 lanes live", lanes);
        run<1>(name, 2, 300000, lanes);
        snprintf(name, sizeof name, "SUSTAINED same, per-lane random operands, %2d of 16", lanes);
        run<1>(name, 2, 300000, -lanes);
      }
    return 0;
  }
  if (argc > 1) {
    // `ubench_dpp long`: launches of ~0.1-0.2 s - the rate the chip SUSTAINS once its clock has settled under an fp64-dense load
    // (the 1 ms launches below finish before it does)
    for (int rep = 0; rep < 3; ++rep) {
      run<0>("SUSTAINED v_fma_f64 x8 independent", 2, 300000);
      run<1>("SUSTAINED v_fmac_f64_dpp x8 independent", 2, 300000);
      run<5>("SUSTAINED v_fmac_f64_dpp dist 2 + s_nop 1 per 4", 2, 300000);
    }
    return 0;
  }
  for (int w : {1, 2, 4}) {
    run<0>("v_fma_f64 x8 independent", w);
    run<1>("v_fmac_f64_dpp x8 independent", w);
    run<2>("v_fmac_f64_dpp 2 accumulators (dist 2)", w);
    run<3>("v_fmac_f64_dpp 1 accumulator (chain)", w);
    run<4>("v_fmac_f64_dpp x8 indep + s_nop 1 per 4", w);
    run<5>("v_fmac_f64_dpp dist 2 + s_nop 1 per 4 (cmac)", w);
    run<6>("v_mov_b64_dpp + 2 v_fma_f64", w);
  }
  return 0;
}
