// sa_group_impl.h -- the launch switch over the fill_group instances of ONE kind; included by sa_group_{nw,sw}.hip
// (translation units of their own: the instances take minutes to compile and build side by side with the others).
#pragma once
#include "sa_group.h"
#include "sa_launch.h"

namespace sa {

// fill_group<KIND, K, flags> (sa_group.h): NW / SW, K in {1,2,4,8}; notrace / frac as for launch_persistent
template <int KIND>
static void launch_group(int K, dim3 grid, size_t lds, hipStream_t st, const FillArgs& a, bool notrace, bool frac) {
  const dim3 blk(64 * kWavesPerBlock);
#define SA_GROUP_CASES(F)                                                                        \
  switch (K) {                                                                                   \
    case 1: hipLaunchKernelGGL((fill_group<KIND, 1, F>), grid, blk, lds, st, a); break;          \
    case 2: hipLaunchKernelGGL((fill_group<KIND, 2, F>), grid, blk, lds, st, a); break;          \
    case 4: hipLaunchKernelGGL((fill_group<KIND, 4, F>), grid, blk, lds, st, a); break;          \
    default: hipLaunchKernelGGL((fill_group<KIND, 8, F>), grid, blk, lds, st, a); break;         \
  }
  if constexpr (KIND == KIND_NW) {
    if (frac) {
      if (notrace) {
        SA_GROUP_CASES(FLAG_FRAC | FLAG_NOTRACE)
      } else {
        SA_GROUP_CASES(FLAG_FRAC)
      }
      return;
    }
  }
  if (notrace) {
    SA_GROUP_CASES(FLAG_NOTRACE)
  } else {
    SA_GROUP_CASES(0)
  }
#undef SA_GROUP_CASES
}

}  // namespace sa
