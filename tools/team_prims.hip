// team_prims.hip -- development check of the DPP / bpermute semantics team.h relies on, plus their cost.
#include <hip/hip_runtime.h>
#include <cstdio>
#include "../robot-control-stack_amd/csrc/team.h"
using namespace rcsh;
__global__ void k(double* out, long long* cyc) {
  const int l = threadIdx.x;
  const double x = 100.0 + l;
  out[0 * 64 + l] = row_up<1>(x);
  out[1 * 64 + l] = row_up<2>(x);
  out[2 * 64 + l] = row_down<1>(x);
  out[3 * 64 + l] = row_up_or<4>(-1.0, x);
  out[4 * 64 + l] = lane_get(x, (l & 48) + 6);
  out[5 * 64 + l] = (double)team_ballot((l & 15) < 3 || l == 63);
  // cost of a 12-double shift + add round
  double v[12];
  for (int k2 = 0; k2 < 12; ++k2) v[k2] = x * (k2 + 1);
  long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < 100; ++it) {
#pragma unroll
    for (int k2 = 0; k2 < 12; ++k2) v[k2] += row_up<1>(v[k2]);
#pragma unroll
    for (int k2 = 0; k2 < 12; ++k2) v[k2] += row_up<2>(v[k2]);
  }
  long long t1 = __builtin_readcyclecounter();
  double s = 0;
  for (int k2 = 0; k2 < 12; ++k2) s += v[k2];
  out[6 * 64 + l] = s;
  __shared__ double buf[64 * 12];
  long long t2 = __builtin_readcyclecounter();
  for (int it = 0; it < 100; ++it) {
#pragma unroll
    for (int k2 = 0; k2 < 12; ++k2) buf[l * 12 + k2] = v[k2];
    __syncthreads();
#pragma unroll
    for (int k2 = 0; k2 < 12; ++k2) v[k2] += buf[((l + 63) & 63) * 12 + k2];
    __syncthreads();
  }
  long long t3 = __builtin_readcyclecounter();
  for (int k2 = 0; k2 < 12; ++k2) s += v[k2];
  out[7 * 64 + l] = s;
  if (l == 0) { cyc[0] = t1 - t0; cyc[1] = t3 - t2; }
}
int main() {
  double* out; long long* cyc;
  hipMalloc(&out, 8 * 64 * 8); hipMalloc(&cyc, 64);
  k<<<1, 64>>>(out, cyc);
  double h[8 * 64]; long long hc[2];
  hipMemcpy(h, out, sizeof(h), hipMemcpyDeviceToHost); hipMemcpy(hc, cyc, 16, hipMemcpyDeviceToHost);
  const char* names[6] = {"row_up<1>", "row_up<2>", "row_down<1>", "row_up_or<4>(-1)", "lane_get(team lane 6)", "team_ballot"};
  for (int r = 0; r < 6; ++r) { printf("%-22s:", names[r]); for (int l = 0; l < 20; ++l) printf(" %g", h[r * 64 + l]); printf(" ... l63 %g\n", h[r * 64 + 63]); }
  printf("24 x (2 dpp mov + add) : %.1f cycles per double-shift-add\n", hc[0] / 2400.0);
  printf("LDS exchange of 12 doubles (write, sync, read+add, sync): %.1f cycles per round\n", hc[1] / 100.0);
  return 0;
}
