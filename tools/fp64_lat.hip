// Micro-benchmark: cycles per v_fma_f64 / v_mul_f64 / v_add_f64 for dependent and independent chains, with 16 or 64 active
// lanes, and the issue->use latency of ds_read_b64 in a dependent chain.  One wave on one SIMD (the regime of the
// simulation kernel at 4096 environments).
#include <hip/hip_runtime.h>
#include <cstdio>
template <int CHAINS>
__global__ void k_fma(double* out, long long* cyc, int iters, int active) {
  if ((int)threadIdx.x >= active) return;
  double a[CHAINS];
  for (int c = 0; c < CHAINS; ++c) a[c] = out[c] + threadIdx.x;
  const double b = out[8], d = out[9];
  long long t0 = __builtin_readcyclecounter();
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int u = 0; u < 16; ++u)
#pragma unroll
      for (int c = 0; c < CHAINS; ++c) a[c] = __builtin_fma(a[c], b, d);
  }
  long long t1 = __builtin_readcyclecounter();
  double s = 0;
  for (int c = 0; c < CHAINS; ++c) s += a[c];
  out[16 + threadIdx.x] = s;
  if (threadIdx.x == 0) cyc[0] = t1 - t0;
}
__global__ void k_lds(double* out, long long* cyc, int iters, int active) {
  __shared__ double buf[1024];
  for (int i = threadIdx.x; i < 1024; i += blockDim.x) buf[i] = (double)((i * 7 + 64) & 1023);
  __syncthreads();
  if ((int)threadIdx.x >= active) return;
  int idx = threadIdx.x;
  long long t0 = __builtin_readcyclecounter();
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int u = 0; u < 16; ++u) idx = (int)buf[idx];
  }
  long long t1 = __builtin_readcyclecounter();
  out[16 + threadIdx.x] = idx;
  if (threadIdx.x == 0) cyc[0] = t1 - t0;
}
int main() {
  double* out; long long* cyc;
  hipMalloc(&out, 4096); hipMalloc(&cyc, 64);
  hipMemset(out, 0, 4096);
  const int iters = 1000;
  long long h;
  for (int active : {16, 64}) {
#define RUN(C) k_fma<C><<<1, 64>>>(out, cyc, iters, active); hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost); \
    printf("fma_f64 chains=%d active=%d: %.2f cycles/instr\n", C, active, (double)h / (iters * 16.0 * C));
    RUN(1) RUN(2) RUN(4) RUN(8)
    k_lds<<<1, 64>>>(out, cyc, iters, active); hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost);
    printf("ds_read_b64 dependent (+cvt) active=%d: %.2f cycles/read\n", active, (double)h / (iters * 16.0));
  }
  return 0;
}
