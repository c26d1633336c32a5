// sa_launch.h -- launch interface of the fill-kernel and batch-kernel instances, which are compiled in translation
// units of their own (sa_fill_nw.hip, sa_fill_sw.hip, sa_fill_lcs.hip, sa_batch_launch.hip) so that the library
// builds in parallel; used by seqalign_hip.hip.
#pragma once
#include <hip/hip_runtime.h>

#include "sa_batch.h"
#include "sa_kernels.h"

namespace sa {

// fill_tile<kind, K, fl>: the launch-per-diagonal schedule.  Instantiated: plain K in {1,2,4,8}; FRAC only with K=2;
// DUMP (+FRAC) only with K=1
void launch_fill_kind(int kind, int K, int fl, dim3 grid, size_t lds, hipStream_t st, const FillArgs& a);
// fill_persistent<kind, K, flags>.  notrace: the FLAG_NOTRACE instances (score-only calls, the checkpoint pass);
// lprof: profile-driven LCS (small alphabets; also K = 16); frac: NW with a fractional gap penalty
void launch_persistent_kind(int kind, int K, dim3 grid, size_t lds, hipStream_t st, const FillArgs& a, bool notrace, bool lprof,
                            bool frac);
// fill_group<kind, K, flags> (sa_group.h; NW and SW): four consecutive strips per workgroup, hand-offs through LDS
void launch_group_kind(int kind, int K, dim3 grid, size_t lds, hipStream_t st, const FillArgs& a, bool notrace, bool frac);
void launch_group_nw(int K, dim3 grid, size_t lds, hipStream_t st, const FillArgs& a, bool notrace, bool frac);
void launch_group_sw(int K, dim3 grid, size_t lds, hipStream_t st, const FillArgs& a, bool notrace, bool frac);
// per-kind pieces of the functions above (one translation unit each)
void launch_fill_nw(int K, int fl, dim3 grid, size_t lds, hipStream_t st, const FillArgs& a);
void launch_fill_sw(int K, int fl, dim3 grid, size_t lds, hipStream_t st, const FillArgs& a);
void launch_fill_lcs(int K, int fl, dim3 grid, size_t lds, hipStream_t st, const FillArgs& a);
void launch_persistent_nw(int K, dim3 grid, size_t lds, hipStream_t st, const FillArgs& a, bool notrace, bool lprof, bool frac);
void launch_persistent_sw(int K, dim3 grid, size_t lds, hipStream_t st, const FillArgs& a, bool notrace, bool lprof, bool frac);
void launch_persistent_lcs(int K, dim3 grid, size_t lds, hipStream_t st, const FillArgs& a, bool notrace, bool lprof, bool frac);

// the many-pair kernels (sa_batch.h), K = rows per lane 1..16
void launch_batch_k(int K, hipStream_t st, int count, int alpha, const BatchArgs& a, bool trace = false);
void launch_nw_batch_k(int K, hipStream_t st, int count, int alpha, const BatchArgs& a);
void launch_lcs_batch_k(int K, hipStream_t st, int count, const BatchArgs& a);
void launch_batch_tracebacks(int which, hipStream_t st, const BatchArgs& a);  // 0 SW walk, 1 NW, 2 LCS, 3 SW ungapped diagonal

}  // namespace sa
