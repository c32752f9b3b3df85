// dev probe: do two kernels launched back to back on ONE stream with hipExtAnyOrderLaunch overlap on gfx950?
// build: hipcc --offload-arch=gfx950 -O2 tools/anyorder_probe.hip -o tools/anyorder_probe.bin
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <chrono>
#include <cstdio>
__global__ void spin(long long cycles, int* out) {
  const long long t0 = __builtin_readcyclecounter();
  while (__builtin_readcyclecounter() - t0 < cycles) {}
  if (threadIdx.x == 0) out[blockIdx.x] = 1;
}
int main() {
  int* d; hipMalloc(&d, 4096 * sizeof(int));
  hipStream_t s; hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
  const long long cyc = 210000;  // ~100 us
  for (int mode = 0; mode < 2; ++mode)
    for (int grid : {1, 1024}) {
      for (int rep = 0; rep < 3; ++rep) {
        hipStreamSynchronize(s);
        auto t0 = std::chrono::steady_clock::now();
        for (int k = 0; k < 10; ++k) {
          if (mode == 0) hipLaunchKernelGGL(spin, dim3(grid), dim3(64), 0, s, cyc, d);
          else hipExtLaunchKernelGGL(spin, dim3(grid), dim3(64), 0, s, nullptr, nullptr, hipExtAnyOrderLaunch, cyc, d);
        }
        hipStreamSynchronize(s);
        const double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
        if (rep == 2) printf("mode %s grid %d: 10 launches of ~100 us each took %.1f us (%s)\n", mode ? "anyorder" : "ordered", grid, us, hipGetErrorString(hipGetLastError()));
      }
    }
  return 0;
}
