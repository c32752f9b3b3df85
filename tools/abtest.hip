// dev tool: compare smooth_dynamics (device) between two model blobs on the same states
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "../robot-control-stack_amd/csrc/dyn.h"
using namespace rcsh;
using T = Topo<7, true>;
__global__ void __launch_bounds__(32) k(const DevModel* a, const DevModel* b, const double* qin, double* out) {
  __shared__ double lds[Stage<T, 32>::COUNT * 32];
  __shared__ DevModel lm[2];
  const int words = sizeof(DevModel) / 8;
  for (int k = threadIdx.x; k < words; k += 32) { ((double*)&lm[0])[k] = ((const double*)a)[k]; ((double*)&lm[1])[k] = ((const double*)b)[k]; }
  __syncthreads();
  Stage<T, 32> st{lds + threadIdx.x};
  double q[9], qd[9];
  for (int i = 0; i < 9; ++i) { q[i] = qin[i * 32 + threadIdx.x]; qd[i] = qin[(9 + i) * 32 + threadIdx.x]; }
  Smooth<T> s0, s1;
  double M0[45];
  smooth_dynamics<T, 32>(lm[0], q, qd, st, s0);
  stage_fence();
  for (int k = 0; k < 45; ++k) M0[k] = st.M(k);
  stage_fence();
  smooth_dynamics<T, 32>(lm[1], q, qd, st, s1);
  stage_fence();
  double dm = 0, db = 0, dg = 0;
  for (int k = 0; k < 45; ++k) dm = fmax(dm, fabs(M0[k] - st.M(k)));
  for (int i = 0; i < 9; ++i) { db = fmax(db, fabs(s0.bias[i] - s1.bias[i])); dg = fmax(dg, fabs(s0.gc[i] - s1.gc[i])); }
  out[threadIdx.x] = dm; out[32 + threadIdx.x] = db; out[64 + threadIdx.x] = dg;
  out[96 + threadIdx.x] = s0.bias[7]; out[128 + threadIdx.x] = s1.bias[7];
}
int main(int argc, char** argv) {
  DevModel ha, hb; FILE* f = fopen(argv[1], "rb"); fread(&ha, 1, sizeof(ha), f); fclose(f); f = fopen(argv[2], "rb"); fread(&hb, 1, sizeof(hb), f); fclose(f);
  DevModel *da, *db_; hipMalloc(&da, sizeof(ha)); hipMalloc(&db_, sizeof(hb)); hipMemcpy(da, &ha, sizeof(ha), hipMemcpyHostToDevice); hipMemcpy(db_, &hb, sizeof(hb), hipMemcpyHostToDevice);
  std::vector<double> q(18 * 32); srand(3);
  const double qh[9] = {0, -0.785398163, 0, -2.35619449, 0, 1.570796327, 0.785398163, 1e-6, -2e-6};
  for (int e = 0; e < 32; ++e) for (int i = 0; i < 9; ++i) { q[i * 32 + e] = qh[i] + (i < 7 ? 0.3 * (rand() / (double)RAND_MAX - 0.5) : 0); q[(9 + i) * 32 + e] = 0.5 * (rand() / (double)RAND_MAX - 0.5); }
  double *dq, *dout; hipMalloc(&dq, q.size() * 8); hipMalloc(&dout, 160 * 8); hipMemcpy(dq, q.data(), q.size() * 8, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k, dim3(1), dim3(32), 0, 0, da, db_, dq, dout); hipDeviceSynchronize();
  std::vector<double> o(160); hipMemcpy(o.data(), dout, 160 * 8, hipMemcpyDeviceToHost);
  double dm = 0, db = 0, dg = 0; for (int e = 0; e < 32; ++e) { dm = fmax(dm, o[e]); db = fmax(db, o[32 + e]); dg = fmax(dg, o[64 + e]); }
  printf("device A/B: dM %g dbias %g dgc %g   bias7 %.15g %.15g\n", dm, db, dg, o[96], o[128]);
}
