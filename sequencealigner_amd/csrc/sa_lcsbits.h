// sa_lcsbits.h -- launch interface of the bit-vector LCS kernels (sa_lcsbits.hip), used by seqalign_hip.hip.
#pragma once
#include <hip/hip_runtime.h>

#include "sa_kernels.h"

namespace sa {

constexpr int kBitsRowsPerLane = 32;                      // one dword of vertical differences per lane
constexpr int kBitsStripRows = 64 * kBitsRowsPerLane;     // 2048 rows per strip (wavefront)
constexpr int kBitsSub = 32;                              // steps between record hand-offs (16 or 32)
#ifndef SA_BITS_STORE_REGS
#define SA_BITS_STORE_REGS 16
#endif
// Registers the record dwords of consecutive steps rotate through.  A store's four are not written again before
// kBitsStoreRegs - 3 steps have passed: a VALU write to a register that a store in front of it has not read yet stalls the
// wave until the memory pipeline has taken the data, and with four registers (rewritten two steps after the store) that
// stall was 4 ns of a lone strip's 31 ns per step (one strip of 1 M columns: 31.5 -> 27.5 ms with its record, 25.4 without).
constexpr int kBitsStoreRegs = SA_BITS_STORE_REGS;
constexpr int kBitsRecord0 = 256;                         // index of column 0 in a slot of hand-off records
constexpr int kBitsMaxAlpha = 256;                        // profile symbols incl. the padding symbol (codes are bytes): 256 B of LDS each per wave
constexpr int kBitsLdsBytes = 160 * 1024;                 // LDS of a CU (gfx950); a workgroup's waves share it

// prof[s][sym][lane] = the rows (bit b = row (s * 64 + lane) * 32 + b) whose symbol is `sym`
void bits_launch_profile(hipStream_t st, const uint8_t* xs, int m, int alpha, int S, uint32_t* prof);
// steps per strip (a multiple of 64) and record-slot length for n columns
int bits_steps(int n);
int bits_slot_len(int n);
// the fill: FillArgs as for fill_persistent (W = steps per strip, trace = [S][W][64] dwords or null: score only)
// waves_per_block: 1, 2 or 4 waves per workgroup (each wave works on its own; fewer per workgroup spreads them over more CUs)
void bits_launch_fill(hipStream_t st, int grid_blocks, int waves_per_block, const FillArgs& a);
// column stripes: (H, E) border rows r_lo + 1 .. r_hi <-> vertical-difference bits r_lo .. r_hi - 1 (r_lo a multiple of 32)
void bits_launch_border_to_bits(hipStream_t st, const int2* lb, int m, int r_lo, int r_hi, uint32_t* vbits);
void bits_launch_bits_to_border(hipStream_t st, const uint32_t* vbits, int m, int r_lo, int r_hi, int2* rb);
// the walker over that format (TraceArgs::xs / ys / W set)
void bits_launch_traceback(hipStream_t st, const TraceArgs& t);

}  // namespace sa
