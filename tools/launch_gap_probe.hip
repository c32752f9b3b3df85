// dev tool: what a dependent, (nearly) empty kernel launch costs on this device, by resource footprint.
//   hipcc --offload-arch=gfx950 -O2 tools/launch_gap_probe.hip -o /tmp/launch_gap_probe && /tmp/launch_gap_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <chrono>
__global__ void k_plain(const int* flag, int* out) { if (*flag) out[threadIdx.x] = 1; }
template <int BYTES>
__global__ void k_scratch(const int* flag, int* out) {
  if (!*flag) return;
  volatile int priv[BYTES / 4];
  for (int i = 0; i < BYTES / 4; ++i) priv[i] = i * out[i & 63];
  int s = 0;
  for (int i = 0; i < BYTES / 4; ++i) s += priv[(i * 7 + out[0]) % (BYTES / 4)];
  out[threadIdx.x] = s;
}
template <int BYTES>
__global__ void k_lds(const int* flag, int* out) {
  __shared__ int sh[BYTES / 4];
  if (!*flag) return;
  for (int i = threadIdx.x; i < BYTES / 4; i += 64) sh[i] = i;
  __syncthreads();
  out[threadIdx.x] = sh[(threadIdx.x * 97) % (BYTES / 4)];
}
struct Big { double x[400]; };
__global__ void k_bigarg(Big b, const int* flag, int* out) { if (*flag) out[threadIdx.x] = (int)b.x[threadIdx.x]; }
__global__ void __launch_bounds__(64) k_work(int* out, int iters) {  // ~100 us of dependent work per wave
  double a = out[0];
  for (int i = 0; i < iters; ++i) a = a * 1.0000001 + 1e-9;
  if (a == 12345.678) out[1] = 1;
}
template <class F>
double time_pairs(F&& second, int* out, int reps, int iters, hipStream_t st) {
  for (int i = 0; i < 20; ++i) { hipLaunchKernelGGL(k_work, dim3(1024), dim3(64), 0, st, out, iters); second(); }
  hipStreamSynchronize(st);
  auto t0 = std::chrono::steady_clock::now();
  for (int i = 0; i < reps; ++i) { hipLaunchKernelGGL(k_work, dim3(1024), dim3(64), 0, st, out, iters); second(); }
  hipStreamSynchronize(st);
  return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / reps;
}
int main() {
  int *flag, *out;
  hipMalloc(&flag, 4); hipMalloc(&out, 4096 * 4);
  hipMemset(flag, 0, 4); hipMemset(out, 0, 4096 * 4);
  hipStream_t st; hipStreamCreate(&st);
  const int reps = 400, iters = 12000;
  Big big{};
  double base = time_pairs([&] {}, out, reps, iters, st);
  printf("work kernel alone: %.2f us per launch\n", base);
  for (int g : {8, 1024}) {
    printf("grid %d: + plain %.2f", g, time_pairs([&] { hipLaunchKernelGGL(k_plain, dim3(g), dim3(64), 0, st, flag, out); }, out, reps, iters, st) - base);
    printf(", + scratch 256 B %.2f", time_pairs([&] { hipLaunchKernelGGL(k_scratch<256>, dim3(g), dim3(64), 0, st, flag, out); }, out, reps, iters, st) - base);
    printf(", + scratch 2304 B %.2f", time_pairs([&] { hipLaunchKernelGGL(k_scratch<2304>, dim3(g), dim3(64), 0, st, flag, out); }, out, reps, iters, st) - base);
    printf(", + LDS 40 KB %.2f", time_pairs([&] { hipLaunchKernelGGL(k_lds<40960>, dim3(g), dim3(64), 0, st, flag, out); }, out, reps, iters, st) - base);
    printf(", + 3.2 KB of arguments %.2f us\n", time_pairs([&] { hipLaunchKernelGGL(k_bigarg, dim3(g), dim3(64), 0, st, big, flag, out); }, out, reps, iters, st) - base);
  }
  return 0;
}
