// team.h -- cross-lane primitives of the "team" kernels: 16 consecutive lanes (one DPP row) work on ONE
// environment, lane t owning link / joint t.  A wavefront carries four teams.
//
// Why a team: at the batch sizes the reference's users run (<= a few thousand environments) there are fewer
// environments than SIMD lanes on the chip, and a CDNA4 SIMD issues one FP64 instruction per ~5 cycles whether 1
// or 64 lanes are active.  Spreading one environment's links over otherwise idle lanes shortens the dependent
// instruction chain of a substep instead of leaving 3/4 of the machine dark.
//
// Exchange inside a team uses DPP row shifts (register to register, no LDS round trip); `row_shr:n` delivers to
// lane t the value of lane t-n of the same row, lanes with t < n receive zero (bound_ctrl) -- exactly the
// ancestor at distance n along a serial chain.
#pragma once
#include <cstdint>

#include "dyn.h"

namespace rcsh {

#if defined(__HIP__)
#define RCSH_D __device__ inline __attribute__((always_inline))

constexpr int kTeamLanes = 16;

RCSH_D int lo32(double x) { return __double2loint(x); }
RCSH_D int hi32(double x) { return __double2hiint(x); }
RCSH_D double mk64(int hi, int lo) { return __hiloint2double(hi, lo); }

// value of lane t-N (same row); zero where there is no such lane
template <int N>
RCSH_D double row_up(double x) {
  const int lo = __builtin_amdgcn_update_dpp(0, lo32(x), 0x110 + N, 0xf, 0xf, true);
  const int hi = __builtin_amdgcn_update_dpp(0, hi32(x), 0x110 + N, 0xf, 0xf, true);
  return mk64(hi, lo);
}
// value of lane t+N (same row); zero where there is no such lane
template <int N>
RCSH_D double row_down(double x) {
  const int lo = __builtin_amdgcn_update_dpp(0, lo32(x), 0x100 + N, 0xf, 0xf, true);
  const int hi = __builtin_amdgcn_update_dpp(0, hi32(x), 0x100 + N, 0xf, 0xf, true);
  return mk64(hi, lo);
}
// value of lane t-N where that lane exists, `old` elsewhere
template <int N>
RCSH_D double row_up_or(double old, double x) {
  const int lo = __builtin_amdgcn_update_dpp(lo32(old), lo32(x), 0x110 + N, 0xf, 0xf, false);
  const int hi = __builtin_amdgcn_update_dpp(hi32(old), hi32(x), 0x110 + N, 0xf, 0xf, false);
  return mk64(hi, lo);
}
// the same, delivered only to the lanes of the 4-lane banks selected by BANKS (bit b: lanes 4b..4b+3 of every
// row); all other lanes receive zero / `old`
template <int N, int BANKS>
RCSH_D double row_up_banks(double x) {
  const int lo = __builtin_amdgcn_update_dpp(0, lo32(x), 0x110 + N, 0xf, BANKS, false);
  const int hi = __builtin_amdgcn_update_dpp(0, hi32(x), 0x110 + N, 0xf, BANKS, false);
  return mk64(hi, lo);
}
template <int N, int BANKS>
RCSH_D double row_up_or_banks(double old, double x) {
  const int lo = __builtin_amdgcn_update_dpp(lo32(old), lo32(x), 0x110 + N, 0xf, BANKS, false);
  const int hi = __builtin_amdgcn_update_dpp(hi32(old), hi32(x), 0x110 + N, 0xf, BANKS, false);
  return mk64(hi, lo);
}
// value of lane (t - N) mod 16 of the same row
template <int N>
RCSH_D double row_rotate(double x) {
  const int lo = __builtin_amdgcn_update_dpp(0, lo32(x), 0x120 + N, 0xf, 0xf, true);
  const int hi = __builtin_amdgcn_update_dpp(0, hi32(x), 0x120 + N, 0xf, 0xf, true);
  return mk64(hi, lo);
}
// arbitrary lane of the wave (LDS crossbar, no memory): src is an absolute lane index
RCSH_D double lane_get(double x, int src) {
  const int lo = __builtin_amdgcn_ds_bpermute(src << 2, lo32(x));
  const int hi = __builtin_amdgcn_ds_bpermute(src << 2, hi32(x));
  return mk64(hi, lo);
}
// the same for a source lane that is the same in every lane of the wave: into scalar registers (v_readlane), no LDS crossbar
RCSH_D int wave_read(int x, int src) { return __builtin_amdgcn_readlane(x, __builtin_amdgcn_readfirstlane(src)); }
RCSH_D double wave_read(double x, int src) {
  const int s = __builtin_amdgcn_readfirstlane(src);
  return mk64(__builtin_amdgcn_readlane(hi32(x), s), __builtin_amdgcn_readlane(lo32(x), s));
}
// A pointer handed to a non-inlined function has lost its address space: the compiler would reach LDS through flat
// instructions.  The round trip through an LDS-qualified pointer tells it (InferAddressSpaces) where the memory is.
template <class P>
RCSH_D P* in_lds(P* p) {
  return (P*)(__attribute__((address_space(3))) P*)p;
}
// minimum over the team's 16 lanes, in every lane (cyclic rotations within the DPP row)
RCSH_D double team_min(double x) {
  x = fmin(x, row_rotate<1>(x));
  x = fmin(x, row_rotate<2>(x));
  x = fmin(x, row_rotate<4>(x));
  x = fmin(x, row_rotate<8>(x));
  return x;
}
// 16-bit mask of the team's lanes for which `pred` holds
RCSH_D uint32_t team_ballot(bool pred) {
  const uint64_t b = __ballot(pred);
  return (uint32_t)(b >> (threadIdx.x & 48)) & 0xffffu;
}
#endif

}  // namespace rcsh
