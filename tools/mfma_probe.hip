// mfma_probe.hip -- development probe: operand layout and dependent-issue latency of v_mfma_f64_4x4x4_4b_f64
// (one 4x4x4 product per 16-lane block = per team).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
__global__ void k_layout(double* out) {
  const int l = threadIdx.x;
  for (int p = 0; p < 64; ++p) {
    const double a = l + 1;                    // lane l holds l+1 in A
    const double b = l == p ? 1.0 : 0.0;       // B is one-hot at lane p (of the whole wave)
    const double d = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, 0.0, 0, 0, 0);
    out[p * 64 + threadIdx.x] = d;
  }
}
__global__ void k_chain(double* out, long long* cyc, int n) {
  double x = 1.0 + threadIdx.x * 1e-3, acc = (threadIdx.x & 15) % 5 == 0 ? 1.0 : 0.0;
  const double b = 0.5 + 1e-3 * threadIdx.x;
  long long t0 = __builtin_readcyclecounter();
  for (int i = 0; i < n; ++i) {
#pragma unroll
    for (int u = 0; u < 8; ++u) acc = __builtin_amdgcn_mfma_f64_4x4x4f64(acc, b, 0.0, 0, 0, 0);  // D feeds A: the chain
  }
  long long t1 = __builtin_readcyclecounter();
  out[threadIdx.x] = acc + x;
  if (threadIdx.x == 0) cyc[0] = t1 - t0;
}
int main() {
  double* out; long long* cyc;
  hipMalloc(&out, 64 * 64 * 8); hipMalloc(&cyc, 8);
  k_layout<<<1, 64>>>(out);
  static double h[64 * 64];
  hipMemcpy(h, out, sizeof(h), hipMemcpyDeviceToHost);
  for (int p = 0; p < 64; ++p) {
    printf("B lane %2d:", p);
    for (int l = 0; l < 64; ++l) if (h[p * 64 + l] != 0) printf("  D%-2d<-A%-2d", l, (int)h[p * 64 + l] - 1);
    printf("\n");
  }
  k_chain<<<1, 64>>>(out, cyc, 1000);
  long long c; hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
  printf("dependent chain (D -> A): %.1f cycles per v_mfma_f64_4x4x4\n", c / 8000.0);
  return 0;
}
