// sa_fill_nw.hip -- the NW instances of fill_tile / fill_persistent (sa_kernels.h), a translation unit of their own
#include "sa_fill_impl.h"

namespace sa {
void launch_fill_nw(int K, int fl, dim3 grid, size_t lds, hipStream_t st, const FillArgs& a) { launch_fill<KIND_NW>(K, fl, grid, lds, st, a); }
void launch_persistent_nw(int K, dim3 grid, size_t lds, hipStream_t st, const FillArgs& a, bool notrace, bool lprof, bool frac) {
  launch_persistent<KIND_NW>(K, grid, lds, st, a, notrace, lprof, frac);
}
}  // namespace sa
