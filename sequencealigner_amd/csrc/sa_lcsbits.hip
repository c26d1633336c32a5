// sa_lcsbits.hip -- LCS fill as bit-vector arithmetic (gfx950).
//
// The reference's LCS (main.cpp:1577-1587) is L[i][j] = L[i-1][j-1] + 1 on a match, else max(L[i-1][j], L[i][j-1]),
// with the pointer D on a match, else U if L[i-1][j] >= L[i][j-1], else L.  Adjacent cells of L differ by 0 or 1, and
// one column of vertical differences is a bit vector that follows from the column to its left by ONE addition
// (Crochemore et al. 2001 / Hyyro 2004): with V = the column's bits (bit r SET iff L[r+1][j] = L[r][j]) and
// M = the rows whose symbol equals y[j],
//      U = V & M;   V' = (V + U) | (V & ~M)
// -- 3 integer operations for 32 cells instead of 5 per cell.  The pointers follow from V' alone (sa_kernels.h,
// bits_code), so V' IS the traceback record: 1 bit per cell instead of 2, written once, streaming.
//
// Mapping: the same anti-diagonal wavefront as the other kernels.  A wave owns a strip of 2048 rows, lane l the 32 rows
// of its dword, and sweeps the columns with a skew of kBitsSkew = 2 columns per lane (lane l is at column t - 2 l at
// step t).  The addition's carry out of lane l at column c is the carry into lane l+1 at the same column, i.e. two
// steps later: the 64-bit carry-out mask of v_addc_co_u32, shifted left by one, IS the carry-in mask of the step after
// next -- one scalar shift, nothing crosses lanes through the VALU.  (With a skew of one column the mask would be
// needed by the very next instruction group, right behind the trip vector ALU -> scalar register -> scalar ALU ->
// vector ALU.  A skew of 4 measured the same as 2: with 2 the trip is off the critical path.)
// Bit 0 of that mask comes from the strip above (hand-off records as in fill_persistent: here one carry bit per
// column), bit 63 goes to the strip below.  Rows beyond m and columns outside 0..n-1 need no predication: the padding
// symbol has empty masks, so U = 0, nothing carries and V' = V.
//
// The kernel is bound by its stores: one dword per lane per step, m n / 8 bytes in all (125 GB at 1 Mb x 1 Mb).
#include "sa_lcsbits.h"

namespace sa {

template <bool TRACE>
__global__ void __launch_bounds__(64 * kWavesPerBlock) lcs_bits_persistent(const FillArgs a) {
  extern __shared__ uint32_t lds_all[];
  const int lane = threadIdx.x & 63;
  uint32_t* lds = lds_all + (threadIdx.x >> 6) * (a.alpha * 64);
  const int NB = a.W / 64;  // steps per strip: n + 2 * 63, rounded up to whole blocks
  const int ypad = (a.alpha - 1) * 256;  // LDS byte offset of the padding symbol's (empty) masks
  constexpr int SUBN = kBitsSub;         // steps per sub-block (record hand-off granularity)
  constexpr int NSUB = 64 / SUBN;        // sub-blocks per block of 64 steps
  constexpr int D = NSUB;                // records are requested 64 steps ahead (D must divide NSUB: static register choice)                  

  while (true) {
    int s = 0;
    if (lane == 0) s = atomicAdd(a.ticket, 1);
    s = a.s_first + __builtin_amdgcn_readfirstlane(s);
    if (s > a.s_last) break;
    wave_lds_fence();  // the previous strip's LDS reads are done
    const uint32_t* gp = a.prof + (size_t)s * a.alpha * 64;
    for (int i = lane; i < a.alpha * 64; i += 64) lds[i] = gp[i];
    wave_lds_fence();

    // slots and generations as in fill_persistent (ho_store): matrix row 0 is "strip -1", slot 0, generation 0
    const int gen_r = ho_gen((s / a.ring) % 3), gen_w = ho_gen(((s + 1) / a.ring) % 3);
    // record of column c at index c + kBitsRecord0 of a slot; lane 63 meets column t - 63 * kBitsSkew at step t
    const int2* rd = a.bot2 + (size_t)(s % a.ring) * a.NP + kBitsRecord0 + min(lane, SUBN - 1);
    int2* const sink = a.bot2 + (size_t)((s + 1) % a.ring) * a.NP;  // entry 0 of the slot: nobody reads it
    int2* const wr = sink + (kBitsRecord0 - 63 * kBitsSkew) + (lane - (64 - SUBN));
    // Only the first dword of a record is used here (carry bit | generation) and only that is loaded: with the 8-byte
    // load the second register is dead, gets reused at once, and the write-after-write wait for the load in flight
    // (s_waitcnt vmcnt(0), 0.85 us) lands at the top of every sub-block -- 53 of the 54 ns per step, measured.
    auto rec_load = [](const int2* p) { return ld_coherent(reinterpret_cast<const int*>(p)); };
    int pend[D];
#pragma unroll
    for (int j = 0; j < D; ++j) pend[j] = rec_load(rd + SUBN * j);

    // carry bits of the columns of a sub-block, from the records in `cur` (lanes 0..SUBN-1); requests the records
    // D sub-blocks further on into the same registers.  A bounded wait that expires raises `dead` (looked at once per
    // block: the sub-block boundary is straight-line code but for this one branch, which is not taken).
    bool dead = false;
    auto acquire = [&](int& cur, const int Ts, uint32_t& inb) {
      const int need = (lane < SUBN) & (Ts + lane < a.n);
      if (__builtin_expect(__ballot(need & ho_stale(cur, gen_r)) != 0ull, 0)) {
        int spins = 0;
        while (true) {
          __builtin_amdgcn_s_sleep(1);
          cur = rec_load(rd + Ts);
          if (__ballot(need & ho_stale(cur, gen_r)) == 0ull) break;
          // (wave-uniform on purpose: everything derived from the exit stays in scalar registers)
          if (++spins > a.spin_limit || ((spins & 255) == 0 && __builtin_amdgcn_readfirstlane(ld_coherent(a.err)) != 0)) {
            dead = true;
            break;
          }
        }
      }
      inb = (uint32_t)__ballot(need & (cur & 1));
      cur = rec_load(rd + Ts + SUBN * D);
    };

    uint32_t V = 0xffffffffu;  // column 0 of L is all zeros: no vertical step anywhere
    if (a.vleft) {  // column stripe: the vertical differences of the column left of this stripe
      if (a.rows_in) {  // they cross from the neighbour GPU while this kernel runs: the strip's rows must have arrived
        const int need = min(a.m, (s + 1) * kBitsStripRows);
        if (ld_coherent(a.rows_in) < need) {
          const long long t0 = wall_clock64();
          int spins = 0;
          while (ld_coherent(a.rows_in) < need) {
            __builtin_amdgcn_s_sleep(16);
            if (++spins > a.spin_limit || ((spins & 255) == 0 && ld_coherent(a.err) != 0)) {
              dead = true;
              break;
            }
          }
          if (lane == 0 && a.wait_ticks) atomicAdd(a.wait_ticks, (unsigned long long)(wall_clock64() - t0));
        }
        asm volatile("" ::: "memory");  // the border load stays behind the poll
      }
      V = (uint32_t)ld_coherent(reinterpret_cast<const int*>(a.vleft) + s * 64 + lane);
    }
    // What travels down the lanes is not the symbol but the LDS address of its mask for the receiving lane: lane l
    // takes the address lane l-1 used kBitsSkew steps ago, plus 4 (prof[sym][lane] is a dword per lane) -- one
    // v_add_u32_dpp; lane 0 takes the next entry of the in-buffer yb (the block's 64 symbols, rotated once per step).
    // This stream of masks runs ONE SUB-BLOCK AHEAD of the arithmetic and its instructions sit between those of the
    // carry chain (and -> add with carry -> bitop is a chain of dependent instructions; back to back each waits for
    // the one before: 39 ns per step measured against 25 with independent work between them).
    // (as absolute LDS addresses, so that the read is ds_read_b32 on the travelling value itself)
    typedef __attribute__((address_space(3))) const uint32_t lds_u32;
    const int lds0 = (int)(uintptr_t)(__attribute__((address_space(3))) uint32_t*)lds;  // this wave's slice of the dynamic LDS
    int adp[kBitsSkew];  // the addresses this lane used 1 .. kBitsSkew steps ago; before column 0: the padding symbol
#pragma unroll
    for (int q = 0; q < kBitsSkew; ++q) adp[q] = lds0 + ypad + lane * 4;
    const int four = 4;
    int yb = a.yenc[lane] * 256 + lds0;  // block 0
    int ny = a.yenc[64 + lane];          // block 1
    int nyn = 0;
    uint32_t Mc[SUBN], Mn[SUBN];             // masks of the current / the next sub-block
    auto lds_mask = [](int ad) { return *(lds_u32*)(uintptr_t)(uint32_t)ad; };
#pragma unroll
    for (int u = 0; u < SUBN; ++u) {  // sub-block 0
      const int nyb = dpp_shl1_any(yb);
      int ad = yb;  // lane 0 keeps this (its source lane does not exist)
      asm("v_add_u32_dpp %0, %1, %2 wave_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(ad) : "v"(adp[kBitsSkew - 1]), "v"(four));
      yb = nyb;
#pragma unroll
      for (int q = kBitsSkew - 1; q > 0; --q) adp[q] = adp[q - 1];
      adp[0] = ad;
      Mc[u] = lds_mask(ad);
    }
    uint32_t inb = 0;
    acquire(pend[0], 0, inb);
    // carry-out masks of the last kBitsSkew steps (bit l = carry out of lane l): the oldest, shifted left by one, is due now
    unsigned long long cop[kBitsSkew];
#pragma unroll
    for (int q = 0; q < kBitsSkew; ++q) cop[q] = 0;
    uint32_t* tdst = TRACE ? a.trace + ((size_t)(s - a.trace_s0) * ((size_t)a.W * 64 + kBitsStripPad)) + lane * 4 : nullptr;
    uint32_t* tdst2 = TRACE ? tdst + 16 * 64 : nullptr;

#pragma unroll 1
    for (int b = 0; b < NB && !dead; ++b) {
      const int Tb = 64 * b;
      auto sub_block = [&](auto subc) {
        constexpr int sub = decltype(subc)::value;
        const int Ts = Tb + SUBN * sub;
        if (sub == NSUB - 1) {  // the mask stream enters the next block
          yb = ny * 256 + lds0;
          ny = nyn;
        }
        // wave-uniform by construction; said explicitly, because a value that crosses the block loop is otherwise
        // kept in vector registers and the carry operand of v_addc_co_u32 must be a scalar pair
#pragma unroll
        for (int q = 0; q < kBitsSkew; ++q)
          cop[q] = ((unsigned long long)(uint32_t)__builtin_amdgcn_readfirstlane((int)(cop[q] >> 32)) << 32) |
                   (uint32_t)__builtin_amdgcn_readfirstlane((int)cop[q]);
        inb = (uint32_t)__builtin_amdgcn_readfirstlane((int)inb);
        unsigned long long cos[16];
        uint32_t vrec = 0;
        uint32_t Vh[kBitsStoreRegs];
#pragma unroll
        for (int u = 0; u < SUBN; ++u) {
          // One step: U = V & M; V' = (V + U + carry) | (V & ~M), the carry-in mask being the carry-out mask of
          // kBitsSkew steps ago shifted up one lane, with the strip above's bit for this column in lane 0 -- and, in
          // between, the address of the mask this lane needs 16 steps from now.  The wave issues in order: a scalar
          // instruction that reads a mask a vector instruction has just written stalls everything behind it, so the
          // shift sits here, where the mask is used (the compiler would hoist it up to the addition that made the
          // mask), and the bits for the strip below are collected after the 16 steps.
          const uint32_t bit = (inb >> u) & 1u;
          uint32_t Vn, U, sum;
          unsigned long long co;
          int nyb, ad = yb;  // lane 0 keeps yb's entry (its source lane does not exist)
          asm volatile(
              "s_lshl_b64 vcc, %[c2], 1\n\t"
              "v_and_b32 %[u], %[v], %[m]\n\t"
              "s_or_b32 vcc_lo, vcc_lo, %[bit]\n\t"
              "v_mov_b32_dpp %[nyb], %[yb] wave_shl:1 row_mask:0xf bank_mask:0xf\n\t"
              "v_addc_co_u32 %[sum], %[co], %[v], %[u], vcc\n\t"
              "v_add_u32_dpp %[ad], %[adk], %[four] wave_shr:1 row_mask:0xf bank_mask:0xf\n\t"
              "v_bitop3_b32 %[vn], %[sum], %[v], %[m] bitop3:0xf4"  // sum | (v & ~m)
              : [vn] "=v"(Vn), [u] "=&v"(U), [sum] "=&v"(sum), [co] "=&s"(co), [nyb] "=&v"(nyb), [ad] "+v"(ad)
              : [c2] "s"(cop[kBitsSkew - 1]), [bit] "s"(bit), [v] "v"(V), [m] "v"(Mc[u]), [yb] "v"(yb), [adk] "v"(adp[kBitsSkew - 1]),
                [four] "v"(four)
              : "vcc");
          V = Vn;
          Mn[u] = lds_mask(ad);
          // the record leaves four steps at a time, [strip][step / 4][lane][step % 4]: 1 KB in one piece per store
          // instruction (a quarter of the memory operations of a dword per step; 16 steps at a time as 4 x 16 bytes at a
          // 64-byte lane stride measured 70 ms against 47: every instruction then touches 64 lines in part)
          Vh[u % kBitsStoreRegs] = V;
          if (TRACE && (u & 3) == 3) {
            const int q0 = (u % kBitsStoreRegs) & ~3;  // the quad that just filled up (u is a constant once unrolled)
            const u32x4 v4 = {Vh[q0], Vh[q0 + 1], Vh[q0 + 2], Vh[q0 + 3]};
            __builtin_nontemporal_store(v4, reinterpret_cast<u32x4*>((u < 16 ? tdst : tdst2) + ((u & 15) >> 2) * 256));  // offsets reach 4 KB
          }
          yb = nyb;
#pragma unroll
          for (int q = kBitsSkew - 1; q > 0; --q) adp[q] = adp[q - 1], cop[q] = cop[q - 1];
          adp[0] = ad;
          cop[0] = co;
          cos[u & 15] = co;
          // lane 64 - SUBN + u will write the record of step Ts + u (lane 63's carry out of column Ts + u - 63 * kBitsSkew):
          // bit 63 of the step's carry-out mask = bit 31 of its upper half, dropped into that lane -- 16 steps at a time,
          // when the oldest of the 16 masks is long since written
          if ((u & 15) == 15) {
#pragma unroll
            for (int q = 0; q < 16; ++q)
              asm volatile("v_writelane_b32 %0, %1, %2" : "+v"(vrec) : "s"((uint32_t)(cos[q] >> 32)), "n"(64 - SUBN + u - 15 + q));
          }
        }
#pragma unroll
        for (int u = 0; u < SUBN; ++u) Mc[u] = Mn[u];
        acquire(pend[(sub + 1) % D], Ts + SUBN, inb);
        if (sub == 0) nyn = a.yenc[Tb + 128 + lane];  // the symbols of the block after next: wanted three sub-blocks from now
        if (TRACE) tdst += SUBN * 64, tdst2 += SUBN * 64;
        // the other lanes write the slot's entry 0, so that the store is one unpredicated instruction
        ho_store(lane >= 64 - SUBN ? wr + Ts : sink, (int)(vrec >> 31), 0, gen_w);
      };
      sub_block(std::integral_constant<int, 0>());
      sub_block(std::integral_constant<int, 1>());
      if constexpr (NSUB == 4) {
        sub_block(std::integral_constant<int, 2>());
        sub_block(std::integral_constant<int, 3>());
      }
    }
    if (dead) {
      if (lane == 0) st_coherent(a.err, 1);
      break;
    }
    if (a.vright) {  // the stripe's last column for the right neighbour (bits_to_border turns it into border rows)
      if (a.chunk_done) {
        st_coherent(reinterpret_cast<int*>(a.vright) + s * 64 + lane, (int)V);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (lane == 0) __hip_atomic_fetch_add(a.chunk_done + s / a.strips_per_chunk, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      } else {
        a.vright[s * 64 + lane] = V;
      }
    }
    // L[m][n] = the vertical steps of the last column = the clear bits among the rows < m
    const int r0 = (s * 64 + lane) * kBitsRowsPerLane;
    const int nvalid = min(max(a.m - r0, 0), 32);
    const uint32_t vm = nvalid == 32 ? 0xffffffffu : ((1u << nvalid) - 1u);
    int z = __builtin_popcount(~V & vm);
    for (int o = 32; o > 0; o >>= 1) z += __shfl_xor(z, o);
    if (lane == 0 && z) atomicAdd(a.score_out, z);
  }
}

__global__ void build_bits_profile(const uint8_t* __restrict__ xs, int m, int alpha, int S, uint32_t* __restrict__ prof) {
  const size_t total = (size_t)S * alpha * 64;
  for (size_t g = (size_t)blockIdx.x * blockDim.x + threadIdx.x; g < total; g += (size_t)gridDim.x * blockDim.x) {
    const int lane = (int)(g & 63);
    const int sym = (int)((g >> 6) % alpha);
    const int s = (int)((g >> 6) / alpha);
    const int r0 = (s * 64 + lane) * kBitsRowsPerLane;
    uint32_t w = 0;
    if (sym < alpha - 1) {  // the padding symbol matches nothing
#pragma unroll 8
      for (int b = 0; b < 32; ++b)
        if (r0 + b < m && xs[r0 + b] == sym) w |= 1u << b;
    }
    prof[g] = w;
  }
}

// Column stripes travel as (H, E) border rows (include/seqalign_hip.h: 8 bytes per matrix row, H = L[i][c]); the kernel
// above wants vertical-difference bits.  Rows r_lo + 1 .. r_hi of a border -> bits r_lo .. r_hi - 1 (bit r set iff
// L[r + 1] = L[r]; rows beyond m: set).  r_lo is a multiple of 32.
__global__ void border_to_bits(const int2* __restrict__ lb, int m, int r_lo, int r_hi, uint32_t* __restrict__ vbits) {
  const int nd = (r_hi - r_lo + 31) / 32;
  for (int t = blockIdx.x * blockDim.x + threadIdx.x; t < nd; t += gridDim.x * blockDim.x) {
    const int r0 = r_lo + 32 * t;
    uint32_t w = 0;
    int prev = r0 < m ? lb[r0].x : 0;
#pragma unroll 8
    for (int b = 0; b < 32; ++b) {
      const int r = r0 + b;
      bool same = true;
      if (r < m) {
        const int cur = lb[r + 1].x;
        same = cur == prev;
        prev = cur;
      }
      if (same) w |= 1u << b;
    }
    vbits[r0 / 32] = w;
  }
}

// ... and back: bits r_lo .. r_hi - 1 of the stripe's last column -> border rows r_lo + 1 .. r_hi,
// L[i] = L[r_lo] + the clear bits among the rows r_lo .. i - 1 (one workgroup; the fill kernel may still be running: sc1 loads)
__global__ void __launch_bounds__(256) bits_to_border(const uint32_t* __restrict__ vbits, int m, int r_lo, int r_hi, int2* __restrict__ rb) {
  __shared__ int part[256];
  __shared__ int carry;
  if (threadIdx.x == 0) carry = r_lo == 0 ? 0 : rb[r_lo].x;
  __syncthreads();
  const int nd = (r_hi - r_lo + 31) / 32;
  for (int t0 = 0; t0 < nd; t0 += 256) {
    const int t = t0 + threadIdx.x;
    const int r0 = r_lo + 32 * t;
    uint32_t w = 0xffffffffu;
    if (t < nd) w = (uint32_t)ld_coherent(reinterpret_cast<const int*>(vbits) + r0 / 32);
    const int nvalid = min(max(m - r0, 0), 32);
    const uint32_t vm = nvalid == 32 ? 0xffffffffu : ((1u << nvalid) - 1u);
    const int cnt = __builtin_popcount(~w & vm);
    part[threadIdx.x] = cnt;
    __syncthreads();
    for (int o = 1; o < 256; o <<= 1) {  // inclusive scan
      const int v = threadIdx.x >= o ? part[threadIdx.x - o] : 0;
      __syncthreads();
      part[threadIdx.x] += v;
      __syncthreads();
    }
    int L = carry + part[threadIdx.x] - cnt;  // L[r0]
    if (t < nd) {
      for (int b = 0; b < 32; ++b) {
        const int r = r0 + b;
        if (r >= m || r >= r_hi) break;
        L += ((w >> b) & 1u) ? 0 : 1;
        rb[r + 1] = make_int2(L, 0);
      }
    }
    __syncthreads();
    if (threadIdx.x == 255) carry += part[255];
    __syncthreads();
  }
}

void bits_launch_border_to_bits(hipStream_t st, const int2* lb, int m, int r_lo, int r_hi, uint32_t* vbits) {
  const int nd = (r_hi - r_lo + 31) / 32;
  hipLaunchKernelGGL(border_to_bits, dim3(std::max(1, std::min((nd + 255) / 256, 256))), dim3(256), 0, st, lb, m, r_lo, r_hi, vbits);
}
void bits_launch_bits_to_border(hipStream_t st, const uint32_t* vbits, int m, int r_lo, int r_hi, int2* rb) {
  hipLaunchKernelGGL(bits_to_border, dim3(1), dim3(256), 0, st, vbits, m, r_lo, r_hi, rb);
}

int bits_steps(int n) { return ((n + 63 * kBitsSkew + 63) / 64) * 64; }
int bits_slot_len(int n) { return bits_steps(n) + kBitsRecord0 + 192; }
static_assert(kBitsRecord0 >= 63 * kBitsSkew + 2, "records of the columns before column 0 need room in front");

void bits_launch_profile(hipStream_t st, const uint8_t* xs, int m, int alpha, int S, uint32_t* prof) {
  const size_t total = (size_t)S * alpha * 64;
  const int blocks = (int)std::min<size_t>((total + 255) / 256, 65535);
  hipLaunchKernelGGL(build_bits_profile, dim3(blocks), dim3(256), 0, st, xs, m, alpha, S, prof);
}

void bits_launch_fill(hipStream_t st, int grid_blocks, int waves_per_block, const FillArgs& a) {
  const size_t lds = (size_t)a.alpha * 64 * 4 * waves_per_block;
  if (lds > 64 * 1024) {  // beyond the default limit of dynamic LDS (alphabets of more than 127 / 63 symbols at two / four waves)
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&lcs_bits_persistent<true>), hipFuncAttributeMaxDynamicSharedMemorySize, kBitsLdsBytes);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&lcs_bits_persistent<false>), hipFuncAttributeMaxDynamicSharedMemorySize, kBitsLdsBytes);
  }
  if (a.trace)
    hipLaunchKernelGGL(lcs_bits_persistent<true>, dim3(grid_blocks), dim3(64 * waves_per_block), lds, st, a);
  else
    hipLaunchKernelGGL(lcs_bits_persistent<false>, dim3(grid_blocks), dim3(64 * waves_per_block), lds, st, a);
}

void bits_launch_traceback(hipStream_t st, const TraceArgs& t) {
  hipLaunchKernelGGL((traceback_walk<true, 0>), dim3(1), dim3(64), 0, st, t);
}

}  // namespace sa
