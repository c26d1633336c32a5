// Probe: what one LDS instruction costs a LONE wave per SIMD (and 2 per SIMD) when it sits between VALU instructions:
// streams of {8 x v_max_i32, one LDS op}; the difference to the stream without the LDS op is the op's issue cost.
// Build: hipcc --offload-arch=gfx950 -O3 tools/probe_lds_issue.hip -o tools/probe_lds_issue
#include <hip/hip_runtime.h>
#include <cstdio>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

enum { NONE, R32, R64, R96B, R96, W64, W64SAME, W32, R32W64R96, N_KINDS };
static const char* kNames[N_KINDS] = {"none", "ds_read_b32 (per-lane addr)", "ds_read_b64 (per-lane addr)", "ds_read_b96 (broadcast addr)",
                                      "ds_read_b96 (per-lane addr)", "ds_write_b64 (per-lane addr)", "ds_write_b64 (63 lanes one addr)",
                                      "ds_write_b32 (per-lane addr)", "read_b32 + write_b64 + read_b96(bcast)"};

#define V8 asm volatile("v_max_i32 %0, %0, %4\n v_max_i32 %1, %1, %5\n v_max_i32 %2, %2, %6\n v_max_i32 %3, %3, %7\n" \
                        "v_max_i32 %4, %4, %0\n v_max_i32 %5, %5, %1\n v_max_i32 %6, %6, %2\n v_max_i32 %7, %7, %3" \
                        : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));

typedef int i3 __attribute__((ext_vector_type(3)));
typedef int i2 __attribute__((ext_vector_type(2)));

template <int KIND>
__global__ void __launch_bounds__(256) stream(int* out, int iters, int seed) {
  extern __shared__ int lds[];
  int a0 = threadIdx.x + seed, a1 = a0 * 3 + 1, a2 = a0 ^ 0x55, a3 = a0 + 7, a4 = a1 + 9, a5 = a2 + 11, a6 = a3 + 13, a7 = a4 + 17;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int base = wave * 4096;
  const int addr4 = base + lane * 4, addr8 = base + lane * 8, addr16 = base + lane * 16, addrb = base;
  const int addrsame = base + (lane == 63 ? 2048 : 1024);
  int r0 = 0;
  i2 r2 = {0, 0};
  i3 r3 = {0, 0, 0};
  i2 wv = {a0, a1};
  for (int i = threadIdx.x; i < 4096; i += 256) lds[i] = i;
  __syncthreads();
#pragma unroll 1
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int j = 0; j < 32; ++j) {
      V8
      if (KIND == R32) asm volatile("ds_read_b32 %0, %1\n s_waitcnt lgkmcnt(6)" : "=v"(r0) : "v"(addr4));
      if (KIND == R64) asm volatile("ds_read_b64 %0, %1\n s_waitcnt lgkmcnt(6)" : "=v"(r2) : "v"(addr8));
      if (KIND == R96B) asm volatile("ds_read_b96 %0, %1\n s_waitcnt lgkmcnt(6)" : "=v"(r3) : "v"(addrb));
      if (KIND == R96) asm volatile("ds_read_b96 %0, %1\n s_waitcnt lgkmcnt(6)" : "=v"(r3) : "v"(addr16));
      if (KIND == W64) asm volatile("ds_write_b64 %0, %1\n s_waitcnt lgkmcnt(6)" ::"v"(addr8), "v"(wv));
      if (KIND == W64SAME) asm volatile("ds_write_b64 %0, %1\n s_waitcnt lgkmcnt(6)" ::"v"(addrsame), "v"(wv));
      if (KIND == W32) asm volatile("ds_write_b32 %0, %1\n s_waitcnt lgkmcnt(6)" ::"v"(addr4), "v"(a0));
      if (KIND == R32W64R96)
        asm volatile("ds_read_b96 %0, %3\n ds_read_b32 %1, %4\n ds_write_b64 %5, %2\n s_waitcnt lgkmcnt(6)" : "=v"(r3), "=v"(r0) : "v"(wv), "v"(addrb), "v"(addr4), "v"(addr8 + 2048));
    }
  }
  asm volatile("s_waitcnt lgkmcnt(0)");
  out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + r0 + r2.x + r3.x;
}

template <int KIND>
int run(int wps, int n_cu, double ghz, int* dout, double* cyc_out) {
  const int iters = 1024, blocks = n_cu * wps;
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  stream<KIND><<<blocks, 256, 16384>>>(dout, 8, 1);
  CK(hipDeviceSynchronize());
  float best = 1e30f;
  for (int rep = 0; rep < 3; ++rep) {
    CK(hipEventRecord(e0));
    stream<KIND><<<blocks, 256, 16384>>>(dout, iters, 1);
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    best = ms < best ? ms : best;
  }
  const double groups = (double)iters * 32;  // {8 VALU + op} groups per wave
  *cyc_out = best * 1e-3 * ghz * 1e9 / groups;
  return 0;
}

template <int KIND>
int report(int n_cu, double ghz, int* dout, const double* base) {
  for (int i = 0; i < 2; ++i) {
    double c;
    if (run<KIND>(i + 1, n_cu, ghz, dout, &c)) return 1;
    printf("%-42s %d wave(s)/SIMD: %6.1f cycles per {8 VALU + op} per wave, op costs %5.1f cycles of wave time\n", kNames[KIND], i + 1, c, c - base[i]);
  }
  return 0;
}

int main() {
  int* dout;
  CK(hipMalloc(&dout, 64 << 20));
  hipDeviceProp_t p;
  CK(hipGetDeviceProperties(&p, 0));
  const double ghz = p.clockRate / 1e6;
  const int n = p.multiProcessorCount;
  double base[2];
  for (int i = 0; i < 2; ++i)
    if (run<NONE>(i + 1, n, ghz, dout, &base[i])) return 1;
  printf("device %s CUs=%d clock=%.2f GHz; 8 VALU alone: %.1f cycles (1 wave/SIMD), %.1f (2 waves/SIMD, per wave)\n", p.gcnArchName, n, ghz, base[0], base[1]);
  if (report<R32>(n, ghz, dout, base) || report<R64>(n, ghz, dout, base) || report<R96B>(n, ghz, dout, base) || report<R96>(n, ghz, dout, base) ||
      report<W64>(n, ghz, dout, base) || report<W64SAME>(n, ghz, dout, base) || report<W32>(n, ghz, dout, base) || report<R32W64R96>(n, ghz, dout, base))
    return 1;
  return 0;
}
