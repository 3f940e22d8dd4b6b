// Calibration of rocprofv3's FETCH_SIZE / WRITE_SIZE on gfx950 for the access widths the MPC kernel uses (8 and 16 bytes per
// lane), on known byte counts:  rocprofv3 --pmc FETCH_SIZE --output-format csv -d <dir> -- ./tools/bin/ubench_fetch
// (and again with --pmc WRITE_SIZE).  Each kernel streams BYTES once (far larger than the 256 MiB Infinity Cache), so
// counter / BYTES is the factor to apply (MI355X_MICROARCH.md: 0.5 for 16 B/lane reads; others "uncalibrated").
#include <hip/hip_runtime.h>
#include <cstdio>

template <class T>
__global__ __launch_bounds__(256) void rd(const T* __restrict__ src, double* out, size_t n) {
  double acc = 0;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    const T v = src[i];
    acc += reinterpret_cast<const double*>(&v)[0];
  }
  if (acc == 1.2345e-300) out[0] = acc;
}
template <class T>
__global__ __launch_bounds__(256) void wr(T* __restrict__ dst, size_t n) {
  T v;
  for (unsigned q = 0; q < sizeof(T) / 8; ++q) reinterpret_cast<double*>(&v)[q] = (double)threadIdx.x;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) dst[i] = v;
}
// the kernel's pattern: 9 of every 16 lanes active, 8 B per lane, rows of 9 doubles back to back
__global__ __launch_bounds__(64) void rd_rows9(const double* __restrict__ src, double* out, size_t nrows) {
  const int g = threadIdx.x >> 4, jj = threadIdx.x & 15;
  double acc = 0;
  for (size_t r = (size_t)blockIdx.x * 4 + g; r < nrows; r += (size_t)gridDim.x * 4)
    if (jj < 9) acc += src[r * 9 + jj];
  if (acc == 1.2345e-300) out[0] = acc;
}
__global__ __launch_bounds__(64) void wr_rows9(double* __restrict__ dst, size_t nrows) {
  const int g = threadIdx.x >> 4, jj = threadIdx.x & 15;
  for (size_t r = (size_t)blockIdx.x * 4 + g; r < nrows; r += (size_t)gridDim.x * 4)
    if (jj < 9) dst[r * 9 + jj] = (double)jj;
}

int main() {
  const size_t BYTES = (size_t)2 << 30;     // 2 GiB
  void* buf;
  double* out;
  hipMalloc(&buf, BYTES);
  hipMalloc(&out, 64);
  hipMemset(buf, 0, BYTES);
  const int grid = 256 * 16;
  hipLaunchKernelGGL((rd<double>), dim3(grid), dim3(256), 0, 0, (const double*)buf, out, BYTES / 8);
  hipLaunchKernelGGL((rd<double2>), dim3(grid), dim3(256), 0, 0, (const double2*)buf, out, BYTES / 16);
  hipLaunchKernelGGL((wr<double>), dim3(grid), dim3(256), 0, 0, (double*)buf, BYTES / 8);
  hipLaunchKernelGGL((wr<double2>), dim3(grid), dim3(256), 0, 0, (double2*)buf, BYTES / 16);
  const size_t nrows = BYTES / 72;
  hipLaunchKernelGGL(rd_rows9, dim3(grid), dim3(64), 0, 0, (const double*)buf, out, nrows);
  hipLaunchKernelGGL(wr_rows9, dim3(grid), dim3(64), 0, 0, (double*)buf, nrows);
  hipDeviceSynchronize();
  printf("bytes per kernel: rd/wr double, double2: %zu   rows9: %zu\n", BYTES, nrows * 72);
  return 0;
}
