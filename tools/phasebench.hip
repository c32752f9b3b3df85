// phasebench.hip -- development micro-benchmark (not part of the product): cycles per call of each role function of
// the 4-wave substep (dyn4.h), run alone by ONE wave with 16 active lanes and the model in LDS.  Gives the
// instruction-level floor of each phase without barriers or neighbours.
//   tools/phasebench.bin model.bin
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

#include "../robot-control-stack_amd/csrc/dyn.h"
#include "../robot-control-stack_amd/csrc/dyn4.h"

using namespace rcsh;
using T = Topo<7, true>;
constexpr int kLanes = 16;
using ST = Stage4<T, kLanes>;

#define CK(x)                                                                       \
  do {                                                                              \
    hipError_t e = (x);                                                             \
    if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } \
  } while (0)

template <int WHICH>
__global__ void __launch_bounds__(64) k_phase(const DevModel* gm, const double* init, long long* cyc, double* sink, int reps) {
  __shared__ DevModel lm;
  __shared__ double lds[ST::COUNT * kLanes];
  {
    constexpr int kWords = sizeof(DevModel) / 8;
    for (int k = threadIdx.x; k < kWords; k += 64) ((double*)&lm)[k] = ((const double*)gm)[k];
    for (int k = threadIdx.x; k < ST::COUNT * kLanes; k += 64) lds[k] = 0.0;
    __syncthreads();
  }
  if (threadIdx.x >= kLanes) return;
  const DevModel& m = lm;
  const ST st{lds + threadIdx.x};
  for (int i = 0; i < T::NL; ++i) { st.q(i) = init[i] + 1e-3 * threadIdx.x; st.v(i) = 0.01 * (i + 1); }
  for (int i = 0; i < T::NU; ++i) st.c(i) = init[16 + i];
  st.active() = 1.0;
  // a consistent state for the later phases
  phaseA_inertia<T, kLanes, 0>(m, st, true);
  phaseA_inertia<T, kLanes, 1>(m, st, true);
  phaseA_inertia<T, kLanes, 2>(m, st, true);
  phaseA_motion<T, kLanes>(m, st, true);
  phaseB_wrench<T, kLanes, 0>(st); phaseB_wrench<T, kLanes, 1>(st); phaseB_wrench<T, kLanes, 2>(st); phaseB_wrench<T, kLanes, 3>(st);
  phaseC_rows<T, kLanes, 0>(m, st); phaseC_rows<T, kLanes, 1>(m, st); phaseC_rows<T, kLanes, 2>(m, st); phaseC_rows<T, kLanes, 3>(m, st);
  phaseD_actuation<T, kLanes>(m, st);
  double A[T::NTRI];
  Rows<T, kLanes> rows;
  phaseD_implicit_factor<T, kLanes>(m, st, A);
  phaseD_rows<T, kLanes>(m, st, rows);
  stage_fence();
  const long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < reps; ++it) {
    if (WHICH == 0) phaseA_motion<T, kLanes>(m, st, true);
    if (WHICH == 1) phaseA_inertia<T, kLanes, 0>(m, st, true);
    if (WHICH == 2) phaseA_inertia<T, kLanes, 1>(m, st, true);
    if (WHICH == 3) phaseA_inertia<T, kLanes, 2>(m, st, true);
    if (WHICH == 4) phaseB_wrench<T, kLanes, 1>(st);
    if (WHICH == 5) phaseC_rows<T, kLanes, 0>(m, st);
    if (WHICH == 6) phaseD_actuation<T, kLanes>(m, st);
    if (WHICH == 7) phaseD_implicit_factor<T, kLanes>(m, st, A);
    if (WHICH == 8) { phaseD_rows<T, kLanes>(m, st, rows); if (rows.has_eq || rows.limrows) build_factor_H<T, kLanes>(st, rows, rows.limrows, A); }
    if (WHICH == 9) phaseD_constraint_solve<T, kLanes>(st, rows, A);
    if (WHICH == 10) phaseD_integrate<T, kLanes>(m, st, A);
    stage_fence();
  }
  const long long t1 = __builtin_readcyclecounter();
  double s = 0;
  for (int k = 0; k < ST::COUNT; ++k) s += st.at(k);
  for (int k = 0; k < T::NTRI; ++k) s += A[k];
  sink[threadIdx.x] = s;
  if (threadIdx.x == 0) cyc[0] = t1 - t0;
}

int main(int argc, char** argv) {
  if (argc < 2) { printf("usage: phasebench model.bin\n"); return 1; }
  DevModel hm;
  FILE* f = fopen(argv[1], "rb");
  if (!f || fread(&hm, 1, sizeof(hm), f) != sizeof(hm)) { printf("cannot read %s\n", argv[1]); return 1; }
  fclose(f);
  DevModel* dm; double* dinit; long long* dc; double* sink;
  CK(hipMalloc(&dm, sizeof(hm))); CK(hipMemcpy(dm, &hm, sizeof(hm), hipMemcpyHostToDevice));
  double init[32] = {0, -0.785398163, 0, -2.35619449, 0, 1.570796327, 0.785398163, 0.02, 0.02};
  const double tgt[8] = {0.02, -0.75, 0.03, -2.3, 0.01, 1.5, 0.8, 255.0};
  for (int i = 0; i < 8; ++i) init[16 + i] = tgt[i];
  CK(hipMalloc(&dinit, sizeof(init))); CK(hipMemcpy(dinit, init, sizeof(init), hipMemcpyHostToDevice));
  CK(hipMalloc(&dc, 64)); CK(hipMalloc(&sink, 4096));
  const int reps = 200;
  const char* names[11] = {"A motion (W3)", "A inertia W0", "A inertia W1", "A inertia W2", "B wrench W1", "C rows W0",
                           "D actuation", "D implicit factor", "D rows+H factor", "D constraint solve", "D integrate"};
#define RUN(W) { hipLaunchKernelGGL((k_phase<W>), dim3(1), dim3(64), 0, 0, dm, dinit, dc, sink, reps); long long h; \
    CK(hipMemcpy(&h, dc, 8, hipMemcpyDeviceToHost)); printf("%-22s %8.0f cycles/call\n", names[W], (double)h / reps); }
  RUN(0) RUN(1) RUN(2) RUN(3) RUN(4) RUN(5) RUN(6) RUN(7) RUN(8) RUN(9) RUN(10)
  return 0;
}
