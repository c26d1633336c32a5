// sa_fill_sw.hip -- the SW instances of fill_tile / fill_persistent (sa_kernels.h), a translation unit of their own
#include "sa_fill_impl.h"

namespace sa {
void launch_fill_sw(int K, int fl, dim3 grid, size_t lds, hipStream_t st, const FillArgs& a) { launch_fill<KIND_SW>(K, fl, grid, lds, st, a); }
void launch_persistent_sw(int K, dim3 grid, size_t lds, hipStream_t st, const FillArgs& a, bool notrace, bool lprof, bool frac) {
  launch_persistent<KIND_SW>(K, grid, lds, st, a, notrace, lprof, frac);
}
}  // namespace sa
