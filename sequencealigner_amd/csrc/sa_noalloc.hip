// sa_noalloc.hip -- device scratch of sa_sw_batch_noalloc, the one entry point that allocates nothing.
// Its device buffers (sequence staging, pair descriptors, per-pair results, error flag, scoring tables) are this
// file-scope array instead of the ctx's hipMalloc'ed workspaces.  Every GPU the library runs this entry point on holds
// its own copy at its own address until the process ends.  In host code the NAME of the array is only a handle: the
// device address comes from hipGetSymbolAddress, once per device (na_scratch_for_device below, called with that
// device current on the entry point's first call there), and only that pointer is ever handed to kernels, memsets
// and copies.  A translation unit (= code object) of its own: a process that never calls the entry point never
// loads the 33 MiB.
#include "sa_noalloc.h"

#include <atomic>
#include <mutex>

static __device__ __align__(256) unsigned char scratch[sa::kNaBytes];

namespace sa {

namespace {
std::mutex g_na_mu;
unsigned char* g_na_ptr[64] = {nullptr};
std::atomic<int> g_na_busy[64];
}  // namespace

hipError_t na_scratch_for_device(int dev, unsigned char** out) {
  *out = nullptr;
  if (dev < 0 || dev >= 64) return hipErrorInvalidDevice;
  std::lock_guard<std::mutex> lk(g_na_mu);
  if (!g_na_ptr[dev]) {
    int cur = -1;
    if (hipGetDevice(&cur) != hipSuccess || cur != dev) return hipErrorInvalidDevice;  // each GPU has its own copy at its own address
    void* p = nullptr;
    const hipError_t e = hipGetSymbolAddress(&p, HIP_SYMBOL(scratch));
    if (e != hipSuccess) {
      (void)hipGetLastError();
      return e;
    }
    g_na_ptr[dev] = (unsigned char*)p;
  }
  *out = g_na_ptr[dev];
  return hipSuccess;
}

NaBusy::NaBusy(int d) : dev(d), held(d >= 0 && d < 64 && g_na_busy[d].exchange(1, std::memory_order_acquire) == 0) {}
NaBusy::~NaBusy() {
  if (held) g_na_busy[dev].store(0, std::memory_order_release);
}

}  // namespace sa
