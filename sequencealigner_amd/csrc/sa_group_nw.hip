// sa_group_nw.hip -- the NW instances of fill_group (sa_group.h), a translation unit of their own
#include "sa_group_impl.h"

namespace sa {
void launch_group_nw(int K, dim3 grid, size_t lds, hipStream_t st, const FillArgs& a, bool notrace, bool frac) {
  launch_group<KIND_NW>(K, grid, lds, st, a, notrace, frac);
}
}  // namespace sa
