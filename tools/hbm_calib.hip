// dev tool: calibrates rocprofv3's FETCH_SIZE / WRITE_SIZE on THIS path's access pattern -- one 8-byte element per lane,
// lanes of a wavefront on consecutive addresses ([field][env] state arrays) -- against a known byte count, as
// MI355X_MICROARCH.md (HBM section) asks before an absolute figure is trusted.  Run each kernel under
//   rocprofv3 --kernel-trace --pmc FETCH_SIZE -- ./hbm_calib      and      ... --pmc WRITE_SIZE -- ./hbm_calib
// (tools/run_hbm_calib.sh) and divide: bytes moved / (counter x 1024).
#include <hip/hip_runtime.h>

#include <cstdio>

__global__ void k_calib_read8(const double* src, double* sink, size_t n) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  double acc = 0;
  for (; i < n; i += stride) acc += src[i];
  if (acc == 1.2345e300) sink[0] = acc;  // (never true: keeps the loads)
}
__global__ void k_calib_write8(double* dst, size_t n) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  for (; i < n; i += stride) dst[i] = (double)i;
}

int main() {
  const size_t n = (size_t)64 << 20;  // 64 Mi doubles = 512 MiB: past the 256 MiB Infinity Cache
  double *a = nullptr, *sink = nullptr;
  if (hipMalloc(&a, n * sizeof(double)) != hipSuccess || hipMalloc(&sink, 64) != hipSuccess) return 1;
  (void)hipMemset(a, 0, n * sizeof(double));
  (void)hipDeviceSynchronize();
  for (int rep = 0; rep < 3; ++rep) {
    hipLaunchKernelGGL(k_calib_read8, dim3(4096), dim3(256), 0, 0, a, sink, n);
    hipLaunchKernelGGL(k_calib_write8, dim3(4096), dim3(256), 0, 0, a, n);
  }
  (void)hipDeviceSynchronize();
  std::printf("bytes per launch: %zu\n", n * sizeof(double));
  return 0;
}
