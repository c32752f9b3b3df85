// kbench.hip -- development micro-benchmark for the substep kernel (not part of the product).
// Replays the real FR3 DevModel (tools/dump_model.py) through variants of the substep loop and
// prints microseconds per 17-substep launch over 4096 environments.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/kbench.hip -o gpurun_out/kbench && gpurun_out/kbench model.bin
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "../robot-control-stack_amd/csrc/dyn.h"

using namespace rcsh;
using T = Topo<7, true>;

#define CK(x)                                                                       \
  do {                                                                              \
    hipError_t e = (x);                                                             \
    if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } \
  } while (0)

__constant__ DevModel c_model;

enum Src { kConst = 0, kLds = 1, kGlobal = 2 };

template <int BLOCK, int SRC>
__global__ void __launch_bounds__(BLOCK) k_steps(const DevModel* gm, double* S, int n, int nsteps) {
  __shared__ double lds[Stage<T, BLOCK>::COUNT * BLOCK];
  __shared__ DevModel lm;
  const int e = blockIdx.x * BLOCK + threadIdx.x;
  if (SRC == kLds) {
    const int words = sizeof(DevModel) / 8;
#pragma unroll
    for (int it = 0; it < (words + BLOCK - 1) / BLOCK; ++it) { const int k = it * BLOCK + threadIdx.x; if (k < words) ((double*)&lm)[k] = ((const double*)gm)[k]; }
    __syncthreads();
  }
  if (e >= n) return;
  const DevModel& m = SRC == kConst ? c_model : (SRC == kLds ? lm : *gm);
  Stage<T, BLOCK> st{lds + threadIdx.x};
  double q[T::NL], qd[T::NL], c[T::NU];
  for (int i = 0; i < T::NL; ++i) { q[i] = S[i * n + e]; qd[i] = S[(16 + i) * n + e]; }
  for (int i = 0; i < T::NU; ++i) c[i] = S[(32 + i) * n + e];
  Smooth<T> sm;
  for (int i = 0; i < T::NL; ++i) { st.q(i) = q[i]; st.v(i) = qd[i]; }
  for (int i = 0; i < T::NU; ++i) st.c(i) = c[i];
  for (int s = 0; s < nsteps; ++s) substep<T, BLOCK>(m, st);
  for (int i = 0; i < T::NL; ++i) { q[i] = st.q(i); qd[i] = st.v(i); }
  sm.linkP[0] = st.link(9);
  for (int i = 0; i < T::NL; ++i) { S[i * n + e] = q[i]; S[(16 + i) * n + e] = qd[i]; }
  S[48 * n + e] = sm.linkP[0];
}

template <int BLOCK, int SRC>
double run(const char* name, const DevModel* dm, double* dS, const std::vector<double>& init, int n, int nsteps, int iters) {
  CK(hipMemcpy(dS, init.data(), init.size() * 8, hipMemcpyHostToDevice));
  hipEvent_t a, b;
  CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  const int grid = (n + BLOCK - 1) / BLOCK;
  for (int i = 0; i < 3; ++i) hipLaunchKernelGGL((k_steps<BLOCK, SRC>), dim3(grid), dim3(BLOCK), 0, 0, dm, dS, n, nsteps);
  CK(hipDeviceSynchronize());
  CK(hipEventRecord(a));
  for (int i = 0; i < iters; ++i) hipLaunchKernelGGL((k_steps<BLOCK, SRC>), dim3(grid), dim3(BLOCK), 0, 0, dm, dS, n, nsteps);
  CK(hipEventRecord(b));
  CK(hipEventSynchronize(b));
  float ms = 0;
  CK(hipEventElapsedTime(&ms, a, b));
  std::vector<double> out(init.size());
  CK(hipMemcpy(out.data(), dS, out.size() * 8, hipMemcpyDeviceToHost));
  double chk = 0;
  for (int i = 0; i < 9; ++i) chk += out[i * n + 5];
  const double us = ms * 1e3 / iters;
  printf("%-28s block %3d grid %4d : %9.1f us / launch  (%.2f M env-steps/s)  chk %.12f\n", name, BLOCK, grid, us, n / us, chk);
  return us;
}

int main(int argc, char** argv) {
  if (argc < 2) { printf("usage: kbench model.bin [n] [nsteps]\n"); return 1; }
  const int n = argc > 2 ? atoi(argv[2]) : 4096;
  const int nsteps = argc > 3 ? atoi(argv[3]) : 17;
  DevModel hm;
  FILE* f = fopen(argv[1], "rb");
  if (!f || fread(&hm, 1, sizeof(hm), f) != sizeof(hm)) { printf("cannot read %s (%zu bytes expected)\n", argv[1], sizeof(hm)); return 1; }
  fclose(f);
  CK(hipMemcpyToSymbol(HIP_SYMBOL(c_model), &hm, sizeof(hm)));
  DevModel* dm;
  CK(hipMalloc(&dm, sizeof(hm)));
  CK(hipMemcpy(dm, &hm, sizeof(hm), hipMemcpyHostToDevice));
  std::vector<double> init((size_t)64 * n, 0.0);
  const double qh[9] = {0, -0.785398163, 0, -2.35619449, 0, 1.570796327, 0.785398163, 0.02, 0.02};
  srand(1);
  for (int e = 0; e < n; ++e) {
    for (int i = 0; i < 9; ++i) init[i * n + e] = qh[i] + (i < 7 ? 0.05 * (rand() / (double)RAND_MAX - 0.5) : 0);
    for (int i = 0; i < 7; ++i) init[(32 + i) * n + e] = qh[i] + 0.08 * (rand() / (double)RAND_MAX - 0.5);
    init[(32 + 7) * n + e] = (e & 1) ? 255.0 : 0.0;
  }
  double* dS;
  CK(hipMalloc(&dS, init.size() * 8));
  const int iters = 30;
  run<32, kConst>("const-mem model", dm, dS, init, n, nsteps, iters);
  run<32, kConst>("const-mem model", dm, dS, init, n, nsteps, iters);
  run<16, kConst>("const-mem model", dm, dS, init, n, nsteps, iters);
  run<8, kLds>("LDS model", dm, dS, init, n, nsteps, iters);
  run<32, kLds>("LDS model", dm, dS, init, n, nsteps, iters);
  run<16, kLds>("LDS model", dm, dS, init, n, nsteps, iters);
  run<32, kGlobal>("global-pointer model", dm, dS, init, n, nsteps, iters);
  return 0;
}
