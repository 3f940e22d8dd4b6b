// Microbenchmark: can the LDS crossbar feed broadcast operands to fp64 FMAs at the FMA issue rate?
// Pattern under test (a candidate for two 8-lane instances per DPP row): every FMA takes one operand that is
// "lane k of my group of 8" - two ds_swizzle_b32 in bit-mask mode (and_mask 0x18, or_mask k) for the two halves of a
// double, software-pipelined one set of eight operands ahead - against v_fmac_f64_dpp row_newbcast (one instruction).
//   hipcc --offload-arch=gfx950 -O3 tools/ubench_swizzle.hip -o gpurun_out/ubench_swizzle && ./gpurun_out/ubench_swizzle
#include <hip/hip_runtime.h>
#include <cstdio>

union D2 { double d; struct { int lo, hi; } w; };

// t[I] = the value lane I of my group of 8 holds (both halves), for I = 0..7
template <int I>
__device__ __forceinline__ void bcast8_one(D2 (&t)[8], const D2& src) {
  t[I].w.lo = __builtin_amdgcn_ds_swizzle(src.w.lo, 0x0018 | (I << 5));
  t[I].w.hi = __builtin_amdgcn_ds_swizzle(src.w.hi, 0x0018 | (I << 5));
}
__device__ __forceinline__ void bcast8(D2 (&t)[8], const D2& src) {
  bcast8_one<0>(t, src); bcast8_one<1>(t, src); bcast8_one<2>(t, src); bcast8_one<3>(t, src);
  bcast8_one<4>(t, src); bcast8_one<5>(t, src); bcast8_one<6>(t, src); bcast8_one<7>(t, src);
}

// MODE 0: swizzle-fed FMAs, pipelined one set ahead.  MODE 1: v_fmac_f64_dpp row_newbcast (reference).
template <int MODE>
__global__ __launch_bounds__(64) void k(double* out, int iters) {
  double acc[8];
  for (int i = 0; i < 8; ++i) acc[i] = threadIdx.x * 1e-9 + i;
  double y = 0.9999999;
  D2 src, src2;
  src.d = 1.0000001 + threadIdx.x * 1e-12;
  src2.d = 0.9999993 + threadIdx.x * 1e-12;
  if constexpr (MODE == 0) {
    D2 a[8], b[8];
    bcast8(a, src);                                    // prologue: set a in flight
    for (int it = 0; it < iters; ++it) {
      bcast8(b, src2);
#pragma unroll
      for (int i = 0; i < 8; ++i) acc[i] = __builtin_fma(a[i].d, y, acc[i]);
      src.d += 1e-13;                                  // (new operands every time: nothing to hoist or merge)
      bcast8(a, src);
#pragma unroll
      for (int i = 0; i < 8; ++i) acc[i] = __builtin_fma(b[i].d, y, acc[i]);
      src2.d -= 1e-13;
    }
    for (int i = 0; i < 8; ++i) acc[0] += a[i].d;
  } else {
    for (int it = 0; it < iters; ++it) {
      asm volatile(
          "v_fmac_f64_dpp %0, %8, %9 row_newbcast:0 row_mask:0xf bank_mask:0xf\n\tv_fmac_f64_dpp %1, %8, %9 row_newbcast:1 row_mask:0xf bank_mask:0xf\n\t"
          "v_fmac_f64_dpp %2, %8, %9 row_newbcast:2 row_mask:0xf bank_mask:0xf\n\tv_fmac_f64_dpp %3, %8, %9 row_newbcast:3 row_mask:0xf bank_mask:0xf\n\t"
          "v_fmac_f64_dpp %4, %8, %9 row_newbcast:4 row_mask:0xf bank_mask:0xf\n\tv_fmac_f64_dpp %5, %8, %9 row_newbcast:5 row_mask:0xf bank_mask:0xf\n\t"
          "v_fmac_f64_dpp %6, %8, %9 row_newbcast:6 row_mask:0xf bank_mask:0xf\n\tv_fmac_f64_dpp %7, %8, %9 row_newbcast:7 row_mask:0xf bank_mask:0xf\n\t"
          "v_fmac_f64_dpp %0, %8, %9 row_newbcast:0 row_mask:0xf bank_mask:0xf\n\tv_fmac_f64_dpp %1, %8, %9 row_newbcast:1 row_mask:0xf bank_mask:0xf\n\t"
          "v_fmac_f64_dpp %2, %8, %9 row_newbcast:2 row_mask:0xf bank_mask:0xf\n\tv_fmac_f64_dpp %3, %8, %9 row_newbcast:3 row_mask:0xf bank_mask:0xf\n\t"
          "v_fmac_f64_dpp %4, %8, %9 row_newbcast:4 row_mask:0xf bank_mask:0xf\n\tv_fmac_f64_dpp %5, %8, %9 row_newbcast:5 row_mask:0xf bank_mask:0xf\n\t"
          "v_fmac_f64_dpp %6, %8, %9 row_newbcast:6 row_mask:0xf bank_mask:0xf\n\tv_fmac_f64_dpp %7, %8, %9 row_newbcast:7 row_mask:0xf bank_mask:0xf\n\t"
          : "+v"(acc[0]), "+v"(acc[1]), "+v"(acc[2]), "+v"(acc[3]), "+v"(acc[4]), "+v"(acc[5]), "+v"(acc[6]), "+v"(acc[7])
          : "v"(src.d), "v"(y));
    }
  }
  double s = 0;
  for (int i = 0; i < 8; ++i) s += acc[i];
  out[blockIdx.x * 64 + threadIdx.x] = s;
}

template <int MODE>
void run(const char* name, int waves_per_simd) {
  const int grid = 256 * 4 * waves_per_simd, iters = 4000;
  double* d;
  hipMalloc(&d, (size_t)grid * 64 * 8);
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL(k<MODE>, dim3(grid), dim3(64), 0, 0, d, iters);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  hipLaunchKernelGGL(k<MODE>, dim3(grid), dim3(64), 0, 0, d, iters);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  const double fmas = (double)grid * iters * 16;                      // wave-level FMA instructions
  printf("%-40s waves/SIMD %d : %8.3f ms  %6.2f TFLOP/s  cycles/FMA/SIMD %.2f\n", name, waves_per_simd, ms, fmas * 128 / ms / 1e9,
         1024.0 * ms * 1e-3 * 2.4e9 / fmas);
  hipFree(d);
}

int main() {
  for (int w : {1, 2, 4}) {
    run<0>("fma fed by 2 x ds_swizzle (groups of 8)", w);
    run<1>("v_fmac_f64_dpp row_newbcast", w);
  }
  return 0;
}
