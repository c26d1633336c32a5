// Does a wave that writes 1 KB every ~110 ns (the bit-vector LCS: one record store per four steps) get stalled by its stores?
// N waves, each streaming its own 256 MB region, ~50 dependent VALU instructions between two stores; optionally a coherent
// load every 8 stores that is waited for 32 stores later (the hand-off records).  Prints the time against the store-free loop.
// hipcc --offload-arch=gfx950 -O3 -o tools/probe_paced_writes tools/probe_paced_writes.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

template <int MODE>  // 0: no stores, 1: stores, 2: stores over the same 8 KB, 3: stores + coherent loads waited for by count
__global__ void paced(uint32_t* base, size_t dwords_per_wave, int groups, const int* recs, int* out) {
  const int lane = threadIdx.x & 63;
  const size_t wave = (size_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  uint32_t* p = base + wave * dwords_per_wave + lane * 4;
  const int* r = recs + wave * 4096 + lane;
  uint32_t x = lane, acc = 0;
  int pend[4] = {0, 0, 0, 0};
  for (int g = 0; g < groups; g += 32) {
#pragma unroll
    for (int q = 0; q < 32; ++q) {
      asm volatile(
          "v_add_u32 %0, %0, %0\n\tv_add_u32 %0, 1, %0\n\tv_add_u32 %0, %0, %0\n\tv_add_u32 %0, 1, %0\n\tv_add_u32 %0, %0, %0\n\t"
          "v_add_u32 %0, 1, %0\n\tv_add_u32 %0, %0, %0\n\tv_add_u32 %0, 1, %0\n\tv_add_u32 %0, %0, %0\n\tv_add_u32 %0, 1, %0\n\t"
          "v_add_u32 %0, %0, %0\n\tv_add_u32 %0, 1, %0\n\tv_add_u32 %0, %0, %0\n\tv_add_u32 %0, 1, %0\n\tv_add_u32 %0, %0, %0\n\t"
          "v_add_u32 %0, 1, %0\n\tv_add_u32 %0, %0, %0\n\tv_add_u32 %0, 1, %0\n\tv_add_u32 %0, %0, %0\n\tv_add_u32 %0, 1, %0\n\t"
          "v_add_u32 %0, %0, %0\n\tv_add_u32 %0, 1, %0\n\tv_add_u32 %0, %0, %0\n\tv_add_u32 %0, 1, %0\n\tv_add_u32 %0, %0, %0\n\t"
          "v_add_u32 %0, 1, %0\n\tv_add_u32 %0, %0, %0\n\tv_add_u32 %0, 1, %0\n\tv_add_u32 %0, %0, %0\n\tv_add_u32 %0, 1, %0\n\t"
          "v_add_u32 %0, %0, %0\n\tv_add_u32 %0, 1, %0\n\tv_add_u32 %0, %0, %0\n\tv_add_u32 %0, 1, %0\n\tv_add_u32 %0, %0, %0\n\t"
          "v_add_u32 %0, 1, %0\n\tv_add_u32 %0, %0, %0\n\tv_add_u32 %0, 1, %0\n\tv_add_u32 %0, %0, %0\n\tv_add_u32 %0, 1, %0\n\t"
          "v_add_u32 %0, %0, %0\n\tv_add_u32 %0, 1, %0\n\tv_add_u32 %0, %0, %0\n\tv_add_u32 %0, 1, %0\n\tv_add_u32 %0, %0, %0\n\t"
          "v_add_u32 %0, 1, %0\n\tv_add_u32 %0, %0, %0\n\tv_add_u32 %0, 1, %0\n\tv_add_u32 %0, %0, %0\n\tv_add_u32 %0, 1, %0"
          : "+v"(x));
      if (MODE != 0) {
        const u32x4 v = {x, 2u, 3u, 4u};
        __builtin_nontemporal_store(v, reinterpret_cast<u32x4*>(p + (MODE == 2 ? (size_t)(q & 7) : (size_t)(g + q)) * 256));
      }
      if (MODE == 3 && (q & 7) == 7) {  // a coherent load per 8 stores, used 32 stores later: 3 loads + 32 stores since
        const int k = q >> 3;
        asm volatile("s_waitcnt vmcnt(35)" : "+v"(pend[k])::"memory");
        acc += pend[k];
        asm volatile("global_load_dword %0, %1, off sc1" : "=v"(pend[k]) : "v"(r + ((g + q) & 4032)) : "memory");
      }
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" : "+v"(pend[0]), "+v"(pend[1]), "+v"(pend[2]), "+v"(pend[3])::"memory");
  if (x + acc == 0x12345u) out[0] = 1;
}

int main() {
  const int waves = 488, wpb = 2;
  const int groups = 249984;  // 1 M steps / 4
  const size_t per_wave = (size_t)groups * 1024;
  uint32_t* d;
  int *recs, *out;
  CK(hipMalloc(&d, per_wave * waves));
  CK(hipMalloc(&recs, (size_t)waves * 4096 * 4 + 4096));
  CK(hipMalloc(&out, 64));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  for (int mode = 0; mode < 4; ++mode)
    for (int rep = 0; rep < 2; ++rep) {
      CK(hipEventRecord(e0));
      const dim3 grid(waves / wpb), block(64 * wpb);
      if (mode == 0) hipLaunchKernelGGL(paced<0>, grid, block, 0, 0, d, per_wave / 4, groups, recs, out);
      if (mode == 1) hipLaunchKernelGGL(paced<1>, grid, block, 0, 0, d, per_wave / 4, groups, recs, out);
      if (mode == 2) hipLaunchKernelGGL(paced<2>, grid, block, 0, 0, d, per_wave / 4, groups, recs, out);
      if (mode == 3) hipLaunchKernelGGL(paced<3>, grid, block, 0, 0, d, per_wave / 4, groups, recs, out);
      CK(hipEventRecord(e1));
      CK(hipEventSynchronize(e1));
      float ms;
      CK(hipEventElapsedTime(&ms, e0, e1));
      if (rep == 1) {
        const char* what[] = {"no stores", "stores, 1 KB per group, own 256 MB region", "stores over the same 8 KB", "stores + a coherent load per 8, waited for by count"};
        printf("%-55s %7.2f ms  (%.1f ns per group, %.2f TB/s)\n", what[mode], ms, ms * 1e6 / groups, mode ? (double)per_wave * waves / ms / 1e9 : 0.0);
      }
    }
  return 0;
}
