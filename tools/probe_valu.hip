// Probe: DPP wave_shr/wave_shl semantics and int32 VALU issue rate on gfx950.
// Build: hipcc --offload-arch=gfx950 -O3 probe_valu.hip -o probe_valu
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

__global__ void dpp_sem(int* out, const int* in) {
  int v = in[threadIdx.x];
  int old = -1000 - (int)threadIdx.x;
  out[threadIdx.x] = __builtin_amdgcn_update_dpp(old, v, 0x138, 0xf, 0xf, false);        // wave_shr:1
  out[64 + threadIdx.x] = __builtin_amdgcn_update_dpp(old, v, 0x130, 0xf, 0xf, false);   // wave_shl:1
  out[128 + threadIdx.x] = __builtin_amdgcn_update_dpp(old, v, 0x13C, 0xf, 0xf, false);  // wave_ror:1
  out[192 + threadIdx.x] = __builtin_amdgcn_update_dpp(old, v, 0x134, 0xf, 0xf, false);  // wave_rol:1
}

template <int MODE>
__global__ void __launch_bounds__(256) rate(int* out, int iters, int seed) {
  int a = threadIdx.x + seed, b = a * 3 + 1, c = a ^ 0x55, d = a + 7, e = b + 9, f = c + 11, g = d + 13, h = e + 17;
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int u = 0; u < 16; ++u) {
      if (MODE == 0) {  // independent add/max chains
        a = max(a + 3, b); b = max(b + 5, c); c = max(c + 7, d); d = max(d + 9, e);
        e = max(e + 1, f); f = max(f + 2, g); g = max(g + 4, h); h = max(h + 6, a);
      } else if (MODE == 1) {  // DPP wave_shr movs interleaved
        a = __builtin_amdgcn_update_dpp(a, b, 0x138, 0xf, 0xf, false) + 1;
        b = __builtin_amdgcn_update_dpp(b, c, 0x138, 0xf, 0xf, false) + 1;
        c = __builtin_amdgcn_update_dpp(c, d, 0x138, 0xf, 0xf, false) + 1;
        d = __builtin_amdgcn_update_dpp(d, e, 0x138, 0xf, 0xf, false) + 1;
        e = __builtin_amdgcn_update_dpp(e, f, 0x138, 0xf, 0xf, false) + 1;
        f = __builtin_amdgcn_update_dpp(f, g, 0x138, 0xf, 0xf, false) + 1;
        g = __builtin_amdgcn_update_dpp(g, h, 0x138, 0xf, 0xf, false) + 1;
        h = __builtin_amdgcn_update_dpp(h, a, 0x138, 0xf, 0xf, false) + 1;
      } else {  // single dependent chain
        a = max(a + 3, b); a = max(a + 5, c); a = max(a + 7, d); a = max(a + 9, e);
        a = max(a + 1, f); a = max(a + 2, g); a = max(a + 4, h); a = max(a + 6, b);
      }
    }
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = a + b + c + d + e + f + g + h;
}

template <int MODE>
int run(const char* name, int blocks, int threads, int* dout) {
  int iters = 4096;
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  rate<MODE><<<blocks, threads>>>(dout, 16, 1);
  CK(hipDeviceSynchronize());
  CK(hipEventRecord(e0));
  rate<MODE><<<blocks, threads>>>(dout, iters, 1);
  CK(hipEventRecord(e1));
  CK(hipEventSynchronize(e1));
  float ms; CK(hipEventElapsedTime(&ms, e0, e1));
  double insts = (double)iters * 16 * 16;  // VALU instrs per wave (add+max per line = 2 x 8)
  double waves = (double)blocks * threads / 64;
  double lane_ops = insts * waves * 64;
  printf("%-10s blocks=%5d threads=%4d  %.3f ms  %.2f T lane-ops/s  cycles/instr/wave@2.4GHz=%.2f\n", name, blocks,
         threads, ms, lane_ops / ms / 1e9, ms * 1e-3 * 2.4e9 / insts);
  return 0;
}

int main() {
  int *din, *dout;
  CK(hipMalloc(&din, 4096)); CK(hipMalloc(&dout, 64 << 20));
  std::vector<int> h(64); for (int i = 0; i < 64; ++i) h[i] = i;
  CK(hipMemcpy(din, h.data(), 256, hipMemcpyHostToDevice));
  dpp_sem<<<1, 64>>>(dout, din);
  std::vector<int> o(256);
  CK(hipMemcpy(o.data(), dout, 1024, hipMemcpyDeviceToHost));
  const char* nm[4] = {"wave_shr", "wave_shl", "wave_ror", "wave_rol"};
  for (int k = 0; k < 4; ++k) {
    printf("%s: lane0=%d lane1=%d lane15=%d lane16=%d lane31=%d lane32=%d lane62=%d lane63=%d\n", nm[k], o[k * 64], o[k * 64 + 1],
           o[k * 64 + 15], o[k * 64 + 16], o[k * 64 + 31], o[k * 64 + 32], o[k * 64 + 62], o[k * 64 + 63]);
  }
  hipDeviceProp_t p; CK(hipGetDeviceProperties(&p, 0));
  printf("device %s CUs=%d clock=%d kHz\n", p.name, p.multiProcessorCount, p.clockRate);
  // 1 wave per SIMD: 256 CUs * 4 waves ; 2 / 4 / 8 waves per SIMD
  run<0>("indep", 256, 256, dout);
  run<0>("indep", 512, 256, dout);
  run<0>("indep", 1024, 256, dout);
  run<0>("indep", 2048, 256, dout);
  run<0>("indep-w64", 1024, 64, dout);
  run<0>("indep-w64", 2048, 64, dout);
  run<0>("indep-w64", 4096, 64, dout);
  run<1>("dpp", 256, 256, dout);
  run<1>("dpp", 512, 256, dout);
  run<1>("dpp", 1024, 256, dout);
  run<2>("chain", 256, 256, dout);
  run<2>("chain", 512, 256, dout);
  run<2>("chain", 1024, 256, dout);
  return 0;
}
