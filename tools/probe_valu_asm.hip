// Probe: issue rate of the VALU instructions the DP kernels are made of, as inline-asm streams the compiler
// cannot fold, hoist or fuse: cycles per wave64 instruction per SIMD at 1 / 2 / 4 waves
// per SIMD.  Eight independent accumulators, round robin, so that neither dependency latency nor the two wait
// states between a VALU write and a DPP read of the same register are on the path.
// Build: hipcc --offload-arch=gfx950 -O3 tools/probe_valu_asm.hip -o tools/probe_valu_asm
#include <hip/hip_runtime.h>
#include <cstdio>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

enum { ADD, MAX, MAX3, SDWA, DPP, ALIGNBIT, ANDB, LSHLADD, FMA, PKADD16, PKMAX16, N_KINDS };
static const char* kNames[N_KINDS] = {"v_add_u32", "v_max_i32", "v_max3_i32", "v_add_u32_sdwa", "v_mov_b32_dpp wave_shr:1", "v_alignbit_b32",
                                      "v_and_b32", "v_lshl_add_u32", "v_fma_f32", "v_pk_add_u16", "v_pk_max_i16"};

#define ONE(KIND, D, S)                                                                                                       \
  if (KIND == ADD) asm volatile("v_add_u32 %0, %0, %1" : "+v"(D) : "v"(S));                                                   \
  else if (KIND == MAX) asm volatile("v_max_i32 %0, %0, %1" : "+v"(D) : "v"(S));                                              \
  else if (KIND == MAX3) asm volatile("v_max3_i32 %0, %0, %1, %2" : "+v"(D) : "v"(S), "v"(k));                                \
  else if (KIND == SDWA) asm volatile("v_add_u32_sdwa %0, %0, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_1" : "+v"(D) : "v"(S)); \
  else if (KIND == DPP) asm volatile("v_mov_b32_dpp %0, %1 wave_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(D) : "v"(S));       \
  else if (KIND == ALIGNBIT) asm volatile("v_alignbit_b32 %0, %1, %0, 2" : "+v"(D) : "v"(S));                                 \
  else if (KIND == ANDB) asm volatile("v_and_b32 %0, -4, %0" : "+v"(D));                                                     \
  else if (KIND == LSHLADD) asm volatile("v_lshl_add_u32 %0, %0, 1, %1" : "+v"(D) : "v"(S));                                  \
  else if (KIND == FMA) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(D) : "v"(S));                                          \
  else if (KIND == PKADD16) asm volatile("v_pk_add_u16 %0, %0, %1" : "+v"(D) : "v"(S));                                       \
  else asm volatile("v_pk_max_i16 %0, %0, %1" : "+v"(D) : "v"(S));

// 8 instructions: accumulator i reads accumulator i+4 (written four instructions earlier at the least)
#define EIGHT(KIND) ONE(KIND, a0, a4) ONE(KIND, a1, a5) ONE(KIND, a2, a6) ONE(KIND, a3, a7) ONE(KIND, a4, a0) ONE(KIND, a5, a1) ONE(KIND, a6, a2) ONE(KIND, a7, a3)
#define SIXTYFOUR(KIND) EIGHT(KIND) EIGHT(KIND) EIGHT(KIND) EIGHT(KIND) EIGHT(KIND) EIGHT(KIND) EIGHT(KIND) EIGHT(KIND)

template <int KIND>
__global__ void __launch_bounds__(256) stream(int* out, int iters, int seed) {
  int a0 = threadIdx.x + seed, a1 = a0 * 3 + 1, a2 = a0 ^ 0x55, a3 = a0 + 7, a4 = a1 + 9, a5 = a2 + 11, a6 = a3 + 13, a7 = a4 + 17;
  const int k = seed * 5 + 3;
  (void)k;
#pragma unroll 1
  for (int i = 0; i < iters; ++i) {
    SIXTYFOUR(KIND) SIXTYFOUR(KIND) SIXTYFOUR(KIND) SIXTYFOUR(KIND)  // 256 VALU instructions, 3 scalar ones for the loop
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
}

template <int KIND>
int run(int waves_per_simd, int n_cu, double ghz, int* dout) {
  const int iters = 2048, blocks = n_cu * waves_per_simd;  // 256 threads = 4 waves per block, one per SIMD
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  stream<KIND><<<blocks, 256>>>(dout, 8, 1);
  CK(hipDeviceSynchronize());
  float best = 1e30f;
  for (int rep = 0; rep < 3; ++rep) {
    CK(hipEventRecord(e0));
    stream<KIND><<<blocks, 256>>>(dout, iters, 1);
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    best = ms < best ? ms : best;
  }
  const double insts = (double)iters * 256;  // per wave
  const double cyc = best * 1e-3 * ghz * 1e9 / (insts * waves_per_simd);
  printf("%-26s %d wave(s)/SIMD  %8.3f ms  %5.2f cycles per wave64 instruction per SIMD  -> %5.1f T lane-ops/s on %d CUs\n", kNames[KIND],
         waves_per_simd, best, cyc, insts * 64.0 * 4 * n_cu * waves_per_simd / (best * 1e-3) / 1e12, n_cu);
  return 0;
}

template <int KIND>
int sweep(int n_cu, double ghz, int* dout) {
  for (int w : {1, 2, 4})
    if (run<KIND>(w, n_cu, ghz, dout)) return 1;
  return 0;
}

int main() {
  int* dout;
  CK(hipMalloc(&dout, 64 << 20));
  hipDeviceProp_t p;
  CK(hipGetDeviceProperties(&p, 0));
  const double ghz = p.clockRate / 1e6;
  printf("device %s CUs=%d clock=%.2f GHz (cycles are counted at this clock)\n", p.gcnArchName, p.multiProcessorCount, ghz);
  const int n = p.multiProcessorCount;
  if (sweep<ADD>(n, ghz, dout) || sweep<MAX>(n, ghz, dout) || sweep<MAX3>(n, ghz, dout) || sweep<SDWA>(n, ghz, dout) || sweep<DPP>(n, ghz, dout) ||
      sweep<ALIGNBIT>(n, ghz, dout) || sweep<ANDB>(n, ghz, dout) || sweep<LSHLADD>(n, ghz, dout) || sweep<FMA>(n, ghz, dout) ||
      sweep<PKADD16>(n, ghz, dout) || sweep<PKMAX16>(n, ghz, dout))
    return 1;
  return 0;
}
