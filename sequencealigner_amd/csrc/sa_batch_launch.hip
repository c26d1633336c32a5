// sa_batch_launch.hip -- the instances of the many-pair kernels (sa_batch.h) and the kind dispatch of the fill launches
#include "sa_launch.h"

namespace sa {

template <int K>
static void launch_batch(hipStream_t st, int count, int alpha, const BatchArgs& a, bool trace) {
  if (trace)
    hipLaunchKernelGGL((sw_batch_kernel<K, true>), dim3(count), dim3(64), batch_lds_bytes<K>(alpha, a.ycap), st, a);
  else
    hipLaunchKernelGGL((sw_batch_kernel<K, false>), dim3(count), dim3(64), batch_lds_bytes<K>(alpha, a.ycap), st, a);
}

template <int K>
static void launch_nw_batch(hipStream_t st, int count, int alpha, const BatchArgs& a) {
  if (a.fo || a.fe)
    hipLaunchKernelGGL((nw_batch_kernel<K, true>), dim3(count), dim3(64), batch_lds_bytes<K>(alpha, a.ycap), st, a);
  else
    hipLaunchKernelGGL((nw_batch_kernel<K, false>), dim3(count), dim3(64), batch_lds_bytes<K>(alpha, a.ycap), st, a);
}

void launch_nw_batch_k(int K, hipStream_t st, int count, int alpha, const BatchArgs& a) {
  switch (K) {
    case 1: launch_nw_batch<1>(st, count, alpha, a); break;
    case 2: launch_nw_batch<2>(st, count, alpha, a); break;
    case 3: launch_nw_batch<3>(st, count, alpha, a); break;
    case 4: launch_nw_batch<4>(st, count, alpha, a); break;
    case 5: launch_nw_batch<5>(st, count, alpha, a); break;
    case 6: launch_nw_batch<6>(st, count, alpha, a); break;
    case 7: launch_nw_batch<7>(st, count, alpha, a); break;
    case 8: launch_nw_batch<8>(st, count, alpha, a); break;
    case 9: launch_nw_batch<9>(st, count, alpha, a); break;
    case 10: launch_nw_batch<10>(st, count, alpha, a); break;
    case 11: launch_nw_batch<11>(st, count, alpha, a); break;
    case 12: launch_nw_batch<12>(st, count, alpha, a); break;
    case 13: launch_nw_batch<13>(st, count, alpha, a); break;
    case 14: launch_nw_batch<14>(st, count, alpha, a); break;
    case 15: launch_nw_batch<15>(st, count, alpha, a); break;
    default: launch_nw_batch<16>(st, count, alpha, a); break;
  }
}

template <int K>
static void launch_lcs_batch(hipStream_t st, int count, const BatchArgs& a) {
  hipLaunchKernelGGL((lcs_batch_kernel<K>), dim3(count), dim3(64), (size_t)a.ycap + 64, st, a);
}

void launch_lcs_batch_k(int K, hipStream_t st, int count, const BatchArgs& a) {
  switch (K) {
    case 1: launch_lcs_batch<1>(st, count, a); break;
    case 2: launch_lcs_batch<2>(st, count, a); break;
    case 3: launch_lcs_batch<3>(st, count, a); break;
    case 4: launch_lcs_batch<4>(st, count, a); break;
    case 5: launch_lcs_batch<5>(st, count, a); break;
    case 6: launch_lcs_batch<6>(st, count, a); break;
    case 7: launch_lcs_batch<7>(st, count, a); break;
    case 8: launch_lcs_batch<8>(st, count, a); break;
    case 9: launch_lcs_batch<9>(st, count, a); break;
    case 10: launch_lcs_batch<10>(st, count, a); break;
    case 11: launch_lcs_batch<11>(st, count, a); break;
    case 12: launch_lcs_batch<12>(st, count, a); break;
    case 13: launch_lcs_batch<13>(st, count, a); break;
    case 14: launch_lcs_batch<14>(st, count, a); break;
    case 15: launch_lcs_batch<15>(st, count, a); break;
    default: launch_lcs_batch<16>(st, count, a); break;
  }
}

void launch_batch_k(int K, hipStream_t st, int count, int alpha, const BatchArgs& a, bool trace) {
  switch (K) {
    case 1: launch_batch<1>(st, count, alpha, a, trace); break;
    case 2: launch_batch<2>(st, count, alpha, a, trace); break;
    case 3: launch_batch<3>(st, count, alpha, a, trace); break;
    case 4: launch_batch<4>(st, count, alpha, a, trace); break;
    case 5: launch_batch<5>(st, count, alpha, a, trace); break;
    case 6: launch_batch<6>(st, count, alpha, a, trace); break;
    case 7: launch_batch<7>(st, count, alpha, a, trace); break;
    case 8: launch_batch<8>(st, count, alpha, a, trace); break;
    case 9: launch_batch<9>(st, count, alpha, a, trace); break;
    case 10: launch_batch<10>(st, count, alpha, a, trace); break;
    case 11: launch_batch<11>(st, count, alpha, a, trace); break;
    case 12: launch_batch<12>(st, count, alpha, a, trace); break;
    case 13: launch_batch<13>(st, count, alpha, a, trace); break;
    case 14: launch_batch<14>(st, count, alpha, a, trace); break;
    case 15: launch_batch<15>(st, count, alpha, a, trace); break;
    default: launch_batch<16>(st, count, alpha, a, trace); break;
  }
}


void launch_batch_tracebacks(int which, hipStream_t st, const BatchArgs& a) {
  const dim3 grid((a.n_desc + 63) / 64), blk(64);
  switch (which) {
    case 0: hipLaunchKernelGGL(sw_batch_traceback, grid, blk, 0, st, a); break;
    case 1: hipLaunchKernelGGL(nw_batch_traceback, grid, blk, 0, st, a); break;
    case 2: hipLaunchKernelGGL(lcs_batch_traceback, grid, blk, 0, st, a); break;
    default: hipLaunchKernelGGL(sw_batch_diag_walk, grid, blk, 0, st, a); break;
  }
}

void launch_fill_kind(int kind, int K, int fl, dim3 grid, size_t lds, hipStream_t st, const FillArgs& a) {
  if (kind == KIND_NW)
    launch_fill_nw(K, fl, grid, lds, st, a);
  else if (kind == KIND_SW)
    launch_fill_sw(K, fl, grid, lds, st, a);
  else
    launch_fill_lcs(K, fl, grid, lds, st, a);
}

void launch_persistent_kind(int kind, int K, dim3 grid, size_t lds, hipStream_t st, const FillArgs& a, bool notrace, bool lprof, bool frac) {
  if (kind == KIND_NW)
    launch_persistent_nw(K, grid, lds, st, a, notrace, lprof, frac);
  else if (kind == KIND_SW)
    launch_persistent_sw(K, grid, lds, st, a, notrace, lprof, frac);
  else
    launch_persistent_lcs(K, grid, lds, st, a, notrace, lprof, frac);
}

void launch_group_kind(int kind, int K, dim3 grid, size_t lds, hipStream_t st, const FillArgs& a, bool notrace, bool frac) {
  if (kind == KIND_NW)
    launch_group_nw(K, grid, lds, st, a, notrace, frac);
  else
    launch_group_sw(K, grid, lds, st, a, notrace, frac);
}

}  // namespace sa
