// Probe: what a workgroup barrier per N VALU instructions costs four waves (one per SIMD) that run the same code in
// lock step, and what an LDS hand-off (ds_write_b64 before, ds_read_b64 + use after the barrier) adds.
// Build: hipcc --offload-arch=gfx950 -O3 tools/probe_barrier.hip -o tools/probe_barrier
#include <hip/hip_runtime.h>
#include <cstdio>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

#define V8 asm volatile("v_max_i32 %0, %0, %4\n v_max_i32 %1, %1, %5\n v_max_i32 %2, %2, %6\n v_max_i32 %3, %3, %7\n" \
                        "v_max_i32 %4, %4, %0\n v_max_i32 %5, %5, %1\n v_max_i32 %6, %6, %2\n v_max_i32 %7, %7, %3" \
                        : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));
typedef int i2 __attribute__((ext_vector_type(2)));

// MODE 0: no barrier; 1: s_barrier; 2: s_waitcnt lgkmcnt(0) + s_barrier with an LDS write before and a dependent LDS read after;
// 3: like 2 without the barrier (LDS round trip only); 4: barrier, only waves 0 of the workgroup do the VALU work;
// 5: barrier with skewed arrivals; 6: the same work without the barrier
template <int MODE, int GROUPS>
__global__ void __launch_bounds__(256) ticks(int* out, int iters, int seed) {
  extern __shared__ int lds[];
  int a0 = threadIdx.x + seed, a1 = a0 * 3 + 1, a2 = a0 ^ 0x55, a3 = a0 + 7, a4 = a1 + 9, a5 = a2 + 11, a6 = a3 + 13, a7 = a4 + 17;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int waddr = ((wave + 1) & 3) * 512 + lane * 8, raddr = wave * 512 + lane * 8;
  i2 r = {0, 0};
#pragma unroll 1
  for (int i = 0; i < iters; ++i) {
    if (MODE != 4 || wave == 0) {
#pragma unroll
      for (int j = 0; j < GROUPS; ++j) { V8 }
    }
    if (MODE == 5 || MODE == 6) {  // skewed arrivals: 0..3 x 8 more VALU instructions, a different wave the slowest in every tick
      const int extra = (i + wave) & 3;
      for (int j = 0; j < extra; ++j) { V8 }
    }
    if (MODE == 2 || MODE == 3) {
      i2 w = {a0, a1};
      asm volatile("ds_write_b64 %0, %1" ::"v"(waddr), "v"(w));
    }
    if (MODE == 1 || MODE == 4 || MODE == 5) asm volatile("s_barrier" ::: "memory");
    if (MODE == 2) asm volatile("s_waitcnt lgkmcnt(0)\n s_barrier" ::: "memory");
    if (MODE == 2 || MODE == 3) {
      asm volatile("ds_read_b64 %0, %1\n s_waitcnt lgkmcnt(0)" : "=v"(r) : "v"(raddr));
      a0 ^= r.x & 1;
    }
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + r.y;
}

template <int MODE, int GROUPS>
int run(int blocks, double ghz, int* dout, double* cyc) {
  const int iters = 20000;
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  ticks<MODE, GROUPS><<<blocks, 256, 4096>>>(dout, 8, 1);
  CK(hipDeviceSynchronize());
  float best = 1e30f;
  for (int rep = 0; rep < 3; ++rep) {
    CK(hipEventRecord(e0));
    ticks<MODE, GROUPS><<<blocks, 256, 4096>>>(dout, iters, 1);
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    best = ms < best ? ms : best;
  }
  *cyc = best * 1e-3 * ghz * 1e9 / iters;
  return 0;
}

template <int GROUPS>
int sweep(int n_cu, double ghz, int* dout) {
  for (int wps : {1, 2}) {
    double c[7];
    if (run<0, GROUPS>(n_cu * wps, ghz, dout, &c[0]) || run<1, GROUPS>(n_cu * wps, ghz, dout, &c[1]) || run<2, GROUPS>(n_cu * wps, ghz, dout, &c[2]) ||
        run<3, GROUPS>(n_cu * wps, ghz, dout, &c[3]) || run<4, GROUPS>(n_cu * wps, ghz, dout, &c[4]) || run<5, GROUPS>(n_cu * wps, ghz, dout, &c[5]) || run<6, GROUPS>(n_cu * wps, ghz, dout, &c[6]))
      return 1;
    printf("     skewed arrivals (each wave 0..3 x 8 more VALU, rotating): no barrier %7.1f cycles per tick (mean work), with s_barrier %7.1f (every tick then takes the slowest wave: + ~60 expected)\n", c[6], c[5]);
    printf("%4d VALU per tick, %d workgroup(s) per CU: tick %7.1f cycles; + s_barrier %6.1f; + LDS write, waitcnt, s_barrier, LDS read %6.1f; LDS write + read, no barrier %6.1f; "
           "barrier with one working wave %6.1f\n", 8 * GROUPS, wps, c[0], c[1] - c[0], c[2] - c[0], c[3] - c[0], c[4] - c[0]);
  }
  return 0;
}

int main() {
  int* dout;
  CK(hipMalloc(&dout, 64 << 20));
  hipDeviceProp_t p;
  CK(hipGetDeviceProperties(&p, 0));
  const double ghz = p.clockRate / 1e6;
  printf("device %s CUs=%d clock=%.2f GHz\n", p.gcnArchName, p.multiProcessorCount, ghz);
  if (sweep<16>(p.multiProcessorCount, ghz, dout) || sweep<64>(p.multiProcessorCount, ghz, dout) || sweep<192>(p.multiProcessorCount, ghz, dout)) return 1;
  return 0;
}
