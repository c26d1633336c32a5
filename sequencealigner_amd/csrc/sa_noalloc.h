// sa_noalloc.h -- sa_sw_batch_noalloc, the one entry point that allocates nothing: layout of its device scratch (a
// file-scope static __device__ array in sa_noalloc.hip) and the host-side handles on it; used by seqalign_hip.hip.
#pragma once
#include <hip/hip_runtime.h>

#include <cstddef>

#include "../../include/seqalign_hip.h"
#include "sa_batch.h"

namespace sa {

constexpr size_t kNaMaxPairs = SA_NOALLOC_MAX_PAIRS;
constexpr size_t kNaMaxSeq = SA_NOALLOC_MAX_SEQ_BYTES;
constexpr size_t kNaDescOff = kNaMaxSeq + 256;                               // sequences first (+ slack for the kernels' reads)
constexpr size_t kNaOutOff = kNaDescOff + kNaMaxPairs * sizeof(PairDesc);  // score | end_i | end_j, kNaMaxPairs ints each
constexpr size_t kNaErrOff = kNaOutOff + 3 * kNaMaxPairs * 4;                // 64 bytes
constexpr size_t kNaSubOff = kNaErrOff + 64;                                 // 1024 bytes
constexpr size_t kNaMapOff = kNaSubOff + 1024;                               // 256 bytes
constexpr size_t kNaBytes = (kNaMapOff + 256 + 255) & ~size_t(255);
constexpr size_t kNaHostBytes = kNaMaxPairs * sizeof(PairDesc) + 3 * kNaMaxPairs * 4 + 64;  // pinned staging per ctx

// Device address of the scratch array on device `dev`, which must be the CURRENT device: hipGetSymbolAddress once per
// device (return code checked), the pointer kept per device.  Only this pointer is ever handed to kernels, memsets and copies.
hipError_t na_scratch_for_device(int dev, unsigned char** out);

// Every ctx and thread of the process shares a device's one array: a call holds this flag from its first write to the
// array until its results are on the host, and a second call arriving meanwhile is refused, not interleaved.
struct NaBusy {
  int dev;
  bool held;
  explicit NaBusy(int d);
  ~NaBusy();
  NaBusy(const NaBusy&) = delete;
  NaBusy& operator=(const NaBusy&) = delete;
};

}  // namespace sa
