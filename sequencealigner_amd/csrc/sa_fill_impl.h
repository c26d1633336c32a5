// sa_fill_impl.h -- the launch switches over the fill-kernel instances of ONE kind; included by sa_fill_{nw,sw,lcs}.hip,
// each of which instantiates its kind only.
#pragma once
#include "sa_launch.h"

namespace sa {

template <int KIND>
static void launch_fill(int K, int fl, dim3 grid, size_t lds, hipStream_t st, const FillArgs& a) {
  // instantiated combinations: plain K in {1,2,4,8}; FRAC only with K=2; DUMP (+FRAC) only with K=1
  if (fl == 0) {
    switch (K) {
      case 1: hipLaunchKernelGGL((fill_tile<KIND, 1, 0>), grid, dim3(64 * kWavesPerBlock), lds, st, a); break;
      case 2: hipLaunchKernelGGL((fill_tile<KIND, 2, 0>), grid, dim3(64 * kWavesPerBlock), lds, st, a); break;
      case 4: hipLaunchKernelGGL((fill_tile<KIND, 4, 0>), grid, dim3(64 * kWavesPerBlock), lds, st, a); break;
      default: hipLaunchKernelGGL((fill_tile<KIND, 8, 0>), grid, dim3(64 * kWavesPerBlock), lds, st, a); break;
    }
  } else if (fl == FLAG_FRAC) {
    hipLaunchKernelGGL((fill_tile<KIND, 2, FLAG_FRAC>), grid, dim3(64 * kWavesPerBlock), lds, st, a);
  } else if (fl == FLAG_DUMP) {
    hipLaunchKernelGGL((fill_tile<KIND, 1, FLAG_DUMP>), grid, dim3(64 * kWavesPerBlock), lds, st, a);
  } else {
    hipLaunchKernelGGL((fill_tile<KIND, 1, FLAG_DUMP | FLAG_FRAC>), grid, dim3(64 * kWavesPerBlock), lds, st, a);
  }
}

// notrace: the FLAG_NOTRACE instances (score-only calls, the checkpoint pass): no codes are packed or stored
// frac: NW with a fractional gap penalty (the FLAG_FRAC instances: one more VALU op per gap add; SW and LCS never need them)
template <int KIND>
static void launch_persistent(int K, dim3 grid, size_t lds, hipStream_t st, const FillArgs& a, bool notrace = false, bool lprof = false,
                       bool frac = false) {
  const dim3 blk(64 * kWavesPerBlock);
  if constexpr (KIND == KIND_NW) {
    if (frac) {
      constexpr int F = FLAG_FRAC, FN = FLAG_FRAC | FLAG_NOTRACE;
      switch (notrace ? -K : K) {
        case 1: hipLaunchKernelGGL((fill_persistent<KIND, 1, F>), grid, blk, lds, st, a); break;
        case 2: hipLaunchKernelGGL((fill_persistent<KIND, 2, F>), grid, blk, lds, st, a); break;
        case 4: hipLaunchKernelGGL((fill_persistent<KIND, 4, F>), grid, blk, lds, st, a); break;
        case 8: hipLaunchKernelGGL((fill_persistent<KIND, 8, F>), grid, blk, lds, st, a); break;
        case -1: hipLaunchKernelGGL((fill_persistent<KIND, 1, FN>), grid, blk, lds, st, a); break;
        case -2: hipLaunchKernelGGL((fill_persistent<KIND, 2, FN>), grid, blk, lds, st, a); break;
        case -4: hipLaunchKernelGGL((fill_persistent<KIND, 4, FN>), grid, blk, lds, st, a); break;
        default: hipLaunchKernelGGL((fill_persistent<KIND, 8, FN>), grid, blk, lds, st, a); break;
      }
      return;
    }
  }
  if constexpr (KIND == KIND_LCS) {
    if (lprof) {  // profile-driven LCS (small alphabets)
      constexpr int F = FLAG_LPROF, FN = FLAG_LPROF | FLAG_NOTRACE;
      switch (notrace ? -K : K) {
        case 1: hipLaunchKernelGGL((fill_persistent<KIND, 1, F>), grid, blk, lds, st, a); break;
        case 2: hipLaunchKernelGGL((fill_persistent<KIND, 2, F>), grid, blk, lds, st, a); break;
        case 4: hipLaunchKernelGGL((fill_persistent<KIND, 4, F>), grid, blk, lds, st, a); break;
        case 8: hipLaunchKernelGGL((fill_persistent<KIND, 8, F>), grid, blk, lds, st, a); break;
        case 16: hipLaunchKernelGGL((fill_persistent<KIND, 16, F>), grid, blk, lds, st, a); break;
        case -1: hipLaunchKernelGGL((fill_persistent<KIND, 1, FN>), grid, blk, lds, st, a); break;
        case -2: hipLaunchKernelGGL((fill_persistent<KIND, 2, FN>), grid, blk, lds, st, a); break;
        case -4: hipLaunchKernelGGL((fill_persistent<KIND, 4, FN>), grid, blk, lds, st, a); break;
        case -16: hipLaunchKernelGGL((fill_persistent<KIND, 16, FN>), grid, blk, lds, st, a); break;
        default: hipLaunchKernelGGL((fill_persistent<KIND, 8, FN>), grid, blk, lds, st, a); break;
      }
      return;
    }
  }
  if (notrace) {
    switch (K) {
      case 1: hipLaunchKernelGGL((fill_persistent<KIND, 1, FLAG_NOTRACE>), grid, blk, lds, st, a); break;
      case 2: hipLaunchKernelGGL((fill_persistent<KIND, 2, FLAG_NOTRACE>), grid, blk, lds, st, a); break;
      case 4: hipLaunchKernelGGL((fill_persistent<KIND, 4, FLAG_NOTRACE>), grid, blk, lds, st, a); break;
      default: hipLaunchKernelGGL((fill_persistent<KIND, 8, FLAG_NOTRACE>), grid, blk, lds, st, a); break;
    }
    return;
  }
  switch (K) {
    case 1: hipLaunchKernelGGL((fill_persistent<KIND, 1, 0>), grid, blk, lds, st, a); break;
    case 2: hipLaunchKernelGGL((fill_persistent<KIND, 2, 0>), grid, blk, lds, st, a); break;
    case 4: hipLaunchKernelGGL((fill_persistent<KIND, 4, 0>), grid, blk, lds, st, a); break;
    default: hipLaunchKernelGGL((fill_persistent<KIND, 8, 0>), grid, blk, lds, st, a); break;
  }
}

}  // namespace sa
