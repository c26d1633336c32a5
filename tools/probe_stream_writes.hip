// How fast can N waves write when each streams its own contiguous region with 1 KB non-temporal stores (the record of the
// bit-vector LCS: 489 strips, 256 MB each)?  hipcc --offload-arch=gfx950 -O3 -o tools/probe_stream_writes tools/probe_stream_writes.hip
// Prints TB/s for a number of waves, waves per workgroup, a per-wave start skew, and the store count between two waits.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

__global__ void stream(uint32_t* base, size_t dwords_per_wave, size_t groups, int skew_groups) {
  const int lane = threadIdx.x & 63;
  const size_t wave = (size_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  uint32_t* p = base + wave * dwords_per_wave + lane * 4;
  const u32x4 v = {1u, 2u, 3u, (uint32_t)wave};
  // start `skew_groups * wave` groups into the region (modulo its size), as strips that run a fixed lag apart do
  size_t g = ((size_t)skew_groups * wave) % groups;
  for (size_t q = 0; q < groups; ++q) {
    __builtin_nontemporal_store(v, reinterpret_cast<u32x4*>(p + g * 256));
    g = g + 1 == groups ? 0 : g + 1;
  }
}

int main() {
  const size_t total = (size_t)120 << 30;  // bytes
  uint32_t* d;
  CK(hipMalloc(&d, total));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  const int waves_list[] = {489, 978, 1956, 3912, 7824};
  for (int wpb : {2, 4})
    for (int skew : {0, 50})
      for (int waves : waves_list) {
        const int blocks = waves / wpb;
        const size_t per_wave = (total / ((size_t)blocks * wpb)) & ~(size_t)1023;  // bytes, whole 1 KB groups
        const size_t groups = per_wave / 1024;
        for (int rep = 0; rep < 2; ++rep) {
          CK(hipEventRecord(e0));
          hipLaunchKernelGGL(stream, dim3(blocks), dim3(64 * wpb), 0, 0, d, per_wave / 4, groups, skew);
          CK(hipEventRecord(e1));
          CK(hipEventSynchronize(e1));
          float ms;
          CK(hipEventElapsedTime(&ms, e0, e1));
          if (rep == 1)
            printf("waves %5d (x%d per workgroup) skew %3d groups: %7.2f ms  %5.2f TB/s\n", blocks * wpb, wpb, skew, ms,
                   (double)groups * 1024 * blocks * wpb / ms / 1e9);
        }
      }
  return 0;
}
