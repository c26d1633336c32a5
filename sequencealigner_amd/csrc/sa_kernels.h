// sa_kernels.h -- CDNA4 (gfx950) device code of the DP alignment engine.
//
// One wavefront (64 lanes) owns a STRIP of 64*K consecutive rows; lane l owns the K rows
// [l*K, l*K+K) of the strip and sweeps the columns with a one-column skew per lane
// (lane l is at column c = t - l at strip-local step t), i.e. the wave walks an
// anti-diagonal.  The cell dependencies that cross lanes (H and F of the row above, the
// target symbol) move one lane down per step with DPP wave_shr:1 -- they never touch
// memory.  The strip's bottom row (H,F per column) is collected in a register that is
// rotated with DPP wave_shl:1 and flushed coalesced every 64 steps (fill_tile) or every 16 steps
// (fill_persistent, as self-validating records); the strip below reads it the same way.  The query profile of the strip is staged in LDS as
// prof[symbol][lane] (one dword, or two for K=8, holding the lane's K substitution
// scores), so a lane's read hits bank == lane whatever symbol it looks up: conflict free.
// Direction codes are packed 2 bits/cell, 16 steps per dword, and written as one
// 16*K-byte contiguous piece per lane (1 KiB..2 KiB contiguous per wave instruction).
//
// The matrix is cut into TILES = (strip s) x (C consecutive global steps); tile (s,q)
// covers global steps [q*C, q*C+C) where global step = strip-local step + 64*s.  Tile
// (s,q) depends only on tiles (s-1,q), (s-1,q-1) and (s,q-1), so launch d runs every
// tile with s+q == d and stream order is the only synchronisation.
//
// NW / SW arithmetic is done on TAGGED values: every score v is held as 4*v and the three
// candidates of a cell carry their identity in the two low bits (M: +2, E: +1, F: +0), so one
// v_max3 yields both the value and the reference's tie-break (M over E over F, main.cpp:649-671:
// 4M+2 beats 4E+1 iff M >= E, 4E+1 beats 4F iff E >= F) and the direction code is just the two
// low bits of the result -- no compares, no selects; v_alignbit shifts it into the trace word.
// SW adds max(.,3): a zero cell gets tag 3 = the reference's 'X' pointer (main.cpp:1010).
//
// Recurrences restate the reference exactly (see DESIGN.md for the proof obligations):
//   NW  main.cpp:622-672   SW  main.cpp:939-1013   LCS  main.cpp:1577-1587
#pragma once
#include <hip/hip_runtime.h>
#include <type_traits>
#include <utility>
#include <climits>
#include <cstdint>

namespace sa {

constexpr int kNeg = INT_MIN / 2;  // main.cpp:703-704
enum { KIND_NW = 0, KIND_SW = 1, KIND_LCS = 2 };

struct FillArgs {
  int m, n;   // rows, columns
  int S, C;   // strips, steps per tile (multiple of 64)
  int W;      // trace dwords per (lane,k) per strip
  int go, ge; // gap penalties in register scale (NW/SW: 4 x floor(penalty); LCS unused)
  int fo, fe; // FRAC: register-scale unit (4) if the open / extend penalty has a fractional part, else 0
  int alpha;  // profile symbols
  int NP;     // length of one bottom-row ring buffer
  const uint8_t* yenc;   // NW/SW: table index per column; LCS: raw byte.  Padded with zeros.
  const uint32_t* prof;  // NW/SW: [S][alpha][64][PW] dwords
  const uint8_t* xpad;   // LCS: raw query bytes padded to S*64*K
  const int2* lb;        // left border per matrix row i = 0..m: (H[i][0], E[i][0]) of this stripe
  int2* rb;              // optional: right border out, (H[i][n], E[i][n]) for i = 1..m (column stripes)
  int* dp;               // optional (DUMP): (m+1)x(n+1) row-major S / L matrix, interior cells
  int* botH;             // fill_tile: [ring][NP]  bottom row H of strip s lives in slot wr_slot(s), index p = c+64
  int* botF;             // fill_tile: [ring][NP]
  int2* bot2;            // fill_persistent: [ring][NP] hand-off records {H | generation bit, F}, same slots and index
  int ring;              // fill_tile: 2, strips alternate between two slots.  fill_persistent: slots of hand-off records, S+1
                         // (every strip keeps its own: checkpoints, small inputs) or the launch's waves + 2 (see ho_store)
  int* state;            // [S][NF][64] lane state carried from tile to tile
  uint32_t* trace;       // [S][W][64][K]; unused by FLAG_NOTRACE kernels (score-only / checkpoint pass)
  int trace_s0;          // strip whose codes sit at the start of `trace` (banded recomputation)
  int* rowbest;          // SW: best score per row
  int* rowbestc;         // SW: first column (0-based) attaining it
  int* score_out;        // NW/LCS: H[m][n]
  int d, s_lo, s_hi;     // launch index, first / last strip of this launch
  // persistent strip pipeline (fill_persistent): one launch, hand-off every 16 steps
  int* ticket;           // next strip to hand out, relative to s_first (zeroed)
  int s_first, s_last;   // strips this persistent launch sweeps (the whole matrix: 0 .. S-1)
  int* err;              // set to 1 when a bounded spin ran out (zeroed)
  int spin_limit;        // polls before a waiting wave gives up
  // persistent pipeline over column stripes: borders cross GPUs while the kernel runs
#ifdef SA_DEBUG_HANDOFF
  long long* dbg;        // [S][4]: start clock, end clock, re-read episodes, re-reads
  long long* dbgblk;     // [8][dbgblk_stride]: clock at the start of every 64-step block of eight chosen strips (tools/analyze_blocks.py)
  int dbgblk_stride;
#endif
  const int* rows_in;    // rows of `lb` that have arrived (raised by stripe_signal on the receive stream); null: all there
  int* chunk_done;       // [chunks] strips of border chunk k whose `rb` rows are written (stripe_wait polls it); null: no right neighbour
  int strips_per_chunk;
  unsigned long long* wait_ticks;  // column stripes: 100 MHz ticks the strips of this launch spent waiting for rows of `lb` (summed over strips)
  // bit-vector LCS over column stripes (sa_lcsbits.hip): the stripe's left / right border column as vertical-difference
  // bits, one dword per (strip, lane) = 32 rows; null: the matrix' own column 0 (all set) / no right neighbour
  const uint32_t* vleft;
  uint32_t* vright;
};

// Agent-scope relaxed atomic accesses (global_load / global_store .. sc1: the store writes through to L2, the
// load is not served from this CU's L1): flags and borders that other waves or other streams touch while a
// kernel runs (MI355X_MICROARCH.md, "Hand-offs measured with sc1 loads").
__device__ __forceinline__ int ld_coherent(const int* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void st_coherent(int* p, int v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

// Inter-wave hand-off of the strip bottom rows inside ONE launch (persistent kernel only): one 8-byte RECORD
// per column, {H with a 2-bit generation in bits 31:30, F}, written by ONE 8-byte agent-scope atomic store and read
// by ONE 8-byte atomic load, so a record is seen whole or not at all and carries its own validity: there is no
// progress flag, no store-queue drain before a publish and no second round trip for the data behind a poll.
// Strip s writes slot (s + 1) mod ring with generation ((s + 1) / ring) mod 3 (matrix row 0 counts as strip -1: slot 0,
// generation 0, written by the host; every other record starts as generation 3 = "not yet written"), so what a slot
// held before -- the records of strip s - ring -- carries the generation before it.  A reader may start polling
// arbitrarily early, so that alone would not do (the records of strip s - 3 ring look like those of strip s); the ring
// is therefore at least two slots longer than the launch has waves (or one slot per strip when that is fewer).
// Strips are drawn in order and strip j cannot finish before strip j-1, so the unfinished strips are a contiguous
// run of at most `waves` strips: while strip s is running, strip s - ring has finished -- every record of the
// slot is that strip's or newer, and nobody still reads what strip s overwrites.
// H is a 4x-scaled score + tag of a real cell, |H| < 2^29 (check_scoring: 4 x 1.2e8), so it survives the round trip
// through 30 bits; F travels as it is.
__device__ __forceinline__ void ho_store(int2* p, int h, int f, int gen) {
  const int hp = (h & 0x3fffffff) | gen;
  const unsigned long long v = ((unsigned long long)(uint32_t)f << 32) | (uint32_t)hp;
  __hip_atomic_store(reinterpret_cast<unsigned long long*>(p), v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ int2 ho_load(const int2* p) {
  const unsigned long long v = __hip_atomic_load(reinterpret_cast<const unsigned long long*>(p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  return make_int2((int)(uint32_t)v, (int)(v >> 32));
}
__device__ __forceinline__ int ho_value(int hp) { return (int)((uint32_t)hp << 2) >> 2; }  // sign-extend bit 29
__device__ __forceinline__ bool ho_stale(int hp, int gen) { return (uint32_t)(hp ^ gen) >= 0x40000000u; }
__device__ __forceinline__ int ho_gen(int g) { return (int)((uint32_t)g << 30); }
constexpr int kGenInit = (int)0xC0000000u;  // generation 3

// slot of the bottom-row buffers strip s reads (the row above it; row 0 of the matrix sits in the
// last slot) and writes
__device__ __forceinline__ int rd_slot(const FillArgs& a, int s) { return a.ring == 2 ? ((s & 1) ^ 1) : (s == 0 ? a.ring - 1 : s - 1); }
__device__ __forceinline__ int wr_slot(const FillArgs& a, int s) { return a.ring == 2 ? (s & 1) : s; }

__device__ __forceinline__ int dpp_shr1(int old, int src) {  // lane l <- src of lane l-1, lane 0 <- old
  return __builtin_amdgcn_update_dpp(old, src, 0x138, 0xf, 0xf, false);
}
__device__ __forceinline__ int dpp_shl1(int old, int src) {  // lane l <- src of lane l+1, lane 63 <- old
  return __builtin_amdgcn_update_dpp(old, src, 0x130, 0xf, 0xf, false);
}
// lane l <- src of lane l-1, lane 0 <- 0 through DPP's bound_ctrl: no register has to hold the zero
__device__ __forceinline__ int dpp_shr1_zero(int src) { return __builtin_amdgcn_update_dpp(0, src, 0x138, 0xf, 0xf, true); }
// rotate-left whose lane 63 is a don't-care: no tie to an old value, so no register copy
__device__ __forceinline__ int dpp_shl1_any(int src) { return __builtin_amdgcn_mov_dpp(src, 0x130, 0xf, 0xf, false); }

// FLAG_LPROF (persistent LCS, at most 31 distinct bytes in the pair): the match term comes from an LDS profile
// like the NW one (byte +6 on a match, -62 otherwise: one SDWA add) instead of a compare and a select per cell.
// FLAG_START (with BND): the block is a strip's first one and the matrix has more than 63 columns, so the only lanes to
// freeze are those that have not reached column 0 yet (no test against n).  The first block is on the critical path of
// every chain of strips: the strip below starts once this strip's lane 63 has left it.
// FLAG_TAIL (with BND): a block behind the first one, so the only lanes to freeze are those that have passed column n - 1.
// The last blocks are on the critical path too: a strip can finish only 80 steps after the strip above it has.
// NW / SW with one of the two: the cells of a frozen lane are not computed and thrown away by selects, they are skipped
// (one divergent `if` around the K cells, i.e. an exec mask): kMaskedEdge.
enum { FLAG_FRAC = 1, FLAG_DUMP = 2, FLAG_PERSIST = 4, FLAG_NOTRACE = 8, FLAG_LPROF = 16, FLAG_START = 32, FLAG_TAIL = 64 };
template <int KIND, int FLAGS>
struct EdgeMode {
  // Measured (100 kb x 100 kb, K = 2): masked 14.2 ms against 13.0 with selects -- the branch around the cells cuts the
  // compiler's profile prefetch at every step -- so it stays off; the code is kept for the record.
  static constexpr bool kMasked = false && KIND != KIND_LCS && (FLAGS & (FLAG_START | FLAG_TAIL)) != 0 && !(FLAGS & FLAG_DUMP);
};

// Workgroups are 4 independent waves (256 threads): the hardware places the 4 waves of one
// workgroup on the 4 SIMDs of a CU, which single-wave workgroups do not guarantee -- and a wave
// that shares its SIMD runs at half speed and, in a chained pipeline, sets everybody's pace.
// Each wave owns a private slice of the dynamic LDS; there is no workgroup barrier anywhere.
constexpr int kWavesPerBlock = 4;

// LDS written by some lanes of a wave and read by others of the SAME wave: the LDS executes one
// wave's instructions in order, so only the compiler has to be kept from reordering.
__device__ __forceinline__ void wave_lds_fence() {
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
  __builtin_amdgcn_wave_barrier();
}

// score + gap.  The reference adds a double penalty to an int and truncates the sum toward
// zero (main.cpp:603-604, 625-634, 958-964).  With g = G + f, G = floor(g), 0 < f < 1:
// trunc(v + g) = w for w >= 0 and w + 1 for w < 0, where w = v + G.
// `unit` is the value of 1 in the caller's scale (1, or 4 for tagged values; a tag < 4 in the low
// bits does not change the sign test).
template <bool FRAC>
__device__ __forceinline__ int add_gap(int v, int G, int frac_unit) {
  const int w = v + G;
  if (!FRAC) return w;
  return w + ((w >> 31) & frac_unit);
}

template <int KIND, int K>
struct Geo {
  static constexpr int PW = (K <= 4) ? 1 : K / 4;  // profile dwords per (symbol, lane)
  static constexpr int ROWB = 64 * PW * 4;         // LDS bytes per symbol
  // lane-state fields carried between tiles
  static constexpr int NF = (KIND == KIND_LCS) ? K + 3 : (KIND == KIND_SW ? 4 * K + 4 : 2 * K + 4);
};

template <int KIND, int K>
struct Lane {
  int H[K];      // H of the cell to the left, per owned row
  int Hg[K];     // H + gap_open of the cell to the left (feeds E of this column and F of the row below)
  int E[K];      // E of the cell to the left (unused for LCS)
  int passH;     // H of the lane's bottom row, handed to the lane below at the next step
  int passF;
  int ypass;     // target symbol handed down
  int dprev;     // H received from above at the previous step = diagonal of row 0
  int best[K];   // SW running best per row
  int bestc[K];  // SW first column attaining it
};

// One skewed step of one lane: K cells of column c.
template <int KIND, int K, bool BND, int FLAGS>
__device__ __forceinline__ void lane_step(Lane<KIND, K>& st, int& ibH, int& ibF, int& yb, int& obH, int& obF,
                                          uint32_t (&tacc)[K], const int u, const char* ldsLane,
                                          const uint32_t (&xr)[K], const FillArgs& a, const int c, int* dprow,
                                          uint32_t (&sbk)[K], uint32_t (&skp)[K]) {
  constexpr bool FRAC = (FLAGS & FLAG_FRAC) != 0;
  constexpr bool DUMP = (FLAGS & FLAG_DUMP) != 0;
  const int go = a.go, ge = a.ge, n = a.n;
  using G = Geo<KIND, K>;
  // rotate the in-buffers first (lane 63 is a don't-care, so no copy), then let the shifted-in
  // values overwrite the old in-buffer registers in place
  const int nibH = dpp_shl1_any(ibH);
  const int nyb = dpp_shl1_any(yb);
  int up = dpp_shr1(ibH, st.passH);
  int uf = 0;
  if (KIND != KIND_LCS) {
    const int nibF = dpp_shl1_any(ibF);
    uf = dpp_shr1(ibF, st.passF);
    ibF = nibF;
  }
  const int yv = dpp_shr1(yb, st.ypass);
  ibH = nibH;
  yb = nyb;

  constexpr bool LPROF = (FLAGS & FLAG_LPROF) != 0;
  uint32_t pw[G::PW];
  if (KIND != KIND_LCS || LPROF) {
    if (G::PW == 1) {
      pw[0] = *reinterpret_cast<const uint32_t*>(ldsLane + yv);
    } else if (G::PW == 2) {
      const uint2 t = *reinterpret_cast<const uint2*>(ldsLane + yv);
      pw[0] = t.x;
      pw[G::PW - 1] = t.y;
    } else {  // 16 rows per lane (LCS over a small alphabet): one ds_read_b128
      const uint4 t = *reinterpret_cast<const uint4*>(ldsLane + yv);
      pw[0] = t.x;
      pw[1 % G::PW] = t.y;
      pw[2 % G::PW] = t.z;
      pw[3 % G::PW] = t.w;
    }
  }
  int diag = st.dprev;
  st.dprev = up;
  st.ypass = yv;
  const bool act = BND ? ((FLAGS & FLAG_START) ? c >= 0 : ((FLAGS & FLAG_TAIL) ? c < n : (c >= 0 && c < n))) : true;
  constexpr bool MASKED = BND && EdgeMode<KIND, FLAGS>::kMasked;  // frozen lanes skip the cells
  constexpr bool SEL = BND && !MASKED;                            // frozen lanes compute and keep their old values by selects
  int f_last = 0;
  int upg = (KIND != KIND_LCS) ? add_gap<FRAC>(up, go, a.fo) : 0;  // (H + gap_open) of the cell above, tag 0
  auto cells = [&]() {
  #pragma unroll
    for (int k = 0; k < K; ++k) {
      const int left = st.H[k];
      int h;
      if (KIND == KIND_LCS) {
        // Tagged like NW: registers hold 4 * L + 1 (the value as a LEFT candidate, tag 1); the cell above
        // becomes the UP candidate with + 1 (tag 2) and the diagonal one, on a match, 4 * (L_diag + 1) + 3.
        // A match always wins (L_diag + 1 >= L_up, L_left), a tie between up and left goes to up
        // (main.cpp:1581), so one max3 yields value and direction: 7 ops per cell, no compare masks.
        // a mismatch (diag - 62) can never reach `left` (L[i][j-1] >= L[i-1][j-1])
        const int dc = LPROF ? diag + (int)(int8_t)(pw[k / 4] >> (8 * (k & 3))) : ((xr[k] == (uint32_t)yv) ? diag + 6 : 0);
        const int ht = max(max(dc, up + 1), left);
        tacc[k] = __builtin_amdgcn_alignbit((uint32_t)ht, tacc[k], 2);
        if (K >= 16 && !(FLAGS & FLAG_NOTRACE)) asm volatile("" : "+v"(tacc[k]));  // pack now (as for SW below): 16 x 16 live maxima do not fit
        // one v_and_or_b32; written as C the compiler splits it into an AND and two ORs per cell
        asm("v_and_or_b32 %0, %1, -4, 1" : "=v"(h) : "v"(ht));
        if (SEL) h = act ? h : left;
      } else {
        // tagged candidates: see the header of this file
        const int sg = (int)(int8_t)(pw[k / 4] >> (8 * (k & 3)));  // 4*sigma + 2
        int f = max(upg, add_gap<FRAC>(uf, ge, a.fe));               // tag 0
        int e = max(st.Hg[k], add_gap<FRAC>(st.E[k], ge, a.fe));     // tag 1
        int mv = diag + sg;                                          // tag 2
        if (KIND == KIND_SW) {
          // E and F are floored individually because the floored values are carried forward
          // (main.cpp:968-973); the floor of M only ever feeds the cell maximum, where the 'X' floor
          // below (tag 3 beats every tagged value <= 2) subsumes it.
          f = max(f, 0);
          e = max(e, 1);
        }
        int ht = max(max(mv, e), f);
        if (KIND == KIND_SW) ht = max(ht, 3);
        tacc[k] = __builtin_amdgcn_alignbit((uint32_t)ht, tacc[k], 2);
        // SW: pack the code here and now.  Left alone the compiler sinks all 16 x K packing ops to the
        // end of the sub-block and keeps every cell's maximum alive until then, which for SW (row
        // bests on top) no longer fits 256 VGPRs and turns into AGPR copies.
        if (KIND == KIND_SW) asm volatile("" : "+v"(tacc[k]));
        h = ht & ~3;
        if (SEL) {  // a frozen lane keeps H and E of its rows (and passF, below); what it computes for F goes nowhere
          e = act ? e : st.E[k];
          h = act ? h : left;
        }
        st.E[k] = e;
        uf = f;
        f_last = f;
      }
      if (KIND == KIND_SW) {
        // first row-major maximum (main.cpp:1124-1127), two ops per cell: inside a 16-step sub-block a
        // row keeps max over its cells of (h << 2) + (15 - u) -- larger score first, then the earlier
        // column; run_block folds it into (best, bestc) with a strict compare once per sub-block.
        // h = 4 * score < 2^29 (check_scoring), so the key stays below 2^31.
        uint32_t key = ((uint32_t)h << 2) + (uint32_t)(15 - u);
        if (SEL) key = act ? key : 0u;
        // Two steps share one v_max3_u32: the even step parks its key, the odd one folds both.  Here and now
        // (the empty asm): left alone the compiler sinks all 16 x K key computations to the end of the
        // sub-block and keeps every cell's value alive until then (spills once the four sub-blocks are in line).
        if ((u & 1) == 0) {
          skp[k] = key;
          asm volatile("" : "+v"(skp[k]));
        } else {
          sbk[k] = max(max(sbk[k], skp[k]), key);
          asm volatile("" : "+v"(sbk[k]));
        }
      }
      if (DUMP) {
        if (act && dprow[k] >= 0) a.dp[(size_t)dprow[k] + c + 1] = h >> 2;
      }
      diag = left;
      st.H[k] = h;
      up = h;
      if (KIND != KIND_LCS) {
        upg = add_gap<FRAC>(h, go, a.fo);
        st.Hg[k] = upg + 1;  // tag 1: this value opens E of the next column
      }
    }
    st.passH = st.H[K - 1];
    if (KIND != KIND_LCS) st.passF = SEL ? (act ? f_last : st.passF) : f_last;
  };
  if (MASKED) {
    if (act) cells();
  } else {
    cells();
  }
  obH = dpp_shl1(st.passH, obH);
  if (KIND != KIND_LCS) obF = dpp_shl1(st.passF, obF);
}

// The codes are written once and read back only by the traceback: streaming (non-temporal) stores.
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
template <int K>
__device__ __forceinline__ void store_trace(uint32_t* dst, const uint32_t (&t)[K]) {
  if (K == 1) {
    __builtin_nontemporal_store(t[0], dst);
  } else if (K == 2) {
    u32x2 v = {t[0], t[1]};
    __builtin_nontemporal_store(v, reinterpret_cast<u32x2*>(dst));
  } else {
#pragma unroll
    for (int k = 0; k < K; k += 4) {
      u32x4 v = {t[k], t[k + 1], t[k + 2], t[k + 3]};
      __builtin_nontemporal_store(v, reinterpret_cast<u32x4*>(dst + k));
    }
  }
}

// loads of one block's inputs (fill_tile): the 64 bottom-row entries of the strip above (in-buffers) and the
// 64 target symbols lane 0 will consume
template <int KIND, int K, int FLAGS>
__device__ __forceinline__ void load_block_inputs(const FillArgs& a, const int s, const int Tb, const int lane, int& ibH,
                                                  int& ibF, int& yb) {
  using G = Geo<KIND, K>;
  const int* rdH = a.botH + (size_t)rd_slot(a, s) * a.NP;
  const int* rdF = a.botF + (size_t)rd_slot(a, s) * a.NP;
  ibF = 0;
  ibH = rdH[Tb + 64 + lane];
  if (KIND != KIND_LCS) ibF = rdF[Tb + 64 + lane];
  yb = a.yenc[Tb + lane];
  if (KIND != KIND_LCS || (FLAGS & FLAG_LPROF)) yb *= G::ROWB;
}

// 64 steps of one strip (fill_tile: one bottom-row hand-off in, one out)
template <int KIND, int K, bool BND, int FLAGS>
__device__ __forceinline__ void run_block(Lane<KIND, K>& st, const FillArgs& a, const int s, const int Tb,
                                          const int lane, const char* ldsLane, const uint32_t (&xr)[K], int* dprow,
                                          int ibH, int ibF, int yb) {
  int* wrH = a.botH + (size_t)wr_slot(a, s) * a.NP;
  int* wrF = a.botF + (size_t)wr_slot(a, s) * a.NP;
  int obH = 0, obF = 0;
  uint32_t* tdst = a.trace + (((size_t)(s - a.trace_s0) * a.W + (Tb >> 4)) * 64 + lane) * K;
#pragma unroll 1
  for (int sub = 0; sub < 4; ++sub) {
    uint32_t tacc[K], sbk[K], skp[K];
#pragma unroll
    for (int k = 0; k < K; ++k) tacc[k] = 0, sbk[k] = 0, skp[k] = 0;
    const int cbase = Tb + 16 * sub - lane;
#pragma unroll
    for (int u = 0; u < 16; ++u)
      lane_step<KIND, K, BND, FLAGS>(st, ibH, ibF, yb, obH, obF, tacc, u, ldsLane, xr, a, cbase + u, dprow, sbk, skp);
    if (KIND == KIND_SW) {
#pragma unroll
      for (int k = 0; k < K; ++k) {
        const int v = (int)(sbk[k] >> 4) << 2;  // the 4 * score of the sub-block's best cell in this row
        const bool better = v > st.best[k];     // strict: earlier sub-blocks hold the smaller columns
        st.best[k] = better ? v : st.best[k];
        st.bestc[k] = better ? cbase + 15 - (int)(sbk[k] & 15u) : st.bestc[k];
      }
    }
    // A compile-time switch: score-only / checkpoint passes (FLAG_NOTRACE) drop the stores and with them
    // the 16 x K packing ops.  As a run-time test on a.trace it cost the normal path 2.5 %: the compiler
    // sank all packing ops of the sub-block into the branch instead of spreading them over the steps.
    if (!(FLAGS & FLAG_NOTRACE)) store_trace<K>(tdst, tacc);
    tdst += 64 * K;
  }
  wrH[Tb + 1 + lane] = obH;
  if (KIND != KIND_LCS) wrF[Tb + 1 + lane] = obF;
}

// lane state at the first column of a strip, from the left border (values of matrix column 0 of
// this stripe); NW / SW registers hold 4*score (+ tag)
// coh: the border arrived from another GPU while this kernel was running (sc1 loads)
template <int KIND, int K, int FLAGS>
__device__ __forceinline__ void strip_begin(Lane<KIND, K>& st, const FillArgs& a, const int r0, const bool coh = false) {
  constexpr int SH = 2;
  constexpr int TAG = (KIND == KIND_LCS) ? 1 : 0;  // LCS registers carry the left-candidate tag
  const int* lbi = reinterpret_cast<const int*>(a.lb);
#pragma unroll
  for (int k = 0; k < K; ++k) {
    const int row = min(r0 + k + 1, a.m);
    const int2 b = coh ? make_int2(ld_coherent(lbi + 2 * row), ld_coherent(lbi + 2 * row + 1)) : a.lb[row];
    st.H[k] = b.x * (1 << SH) + TAG;
    st.E[k] = (KIND == KIND_LCS) ? 0 : b.y * (1 << SH) + 1;
    st.best[k] = 0;
    st.bestc[k] = -1;
  }
  st.passH = st.H[K - 1];
  st.passF = (KIND == KIND_NW) ? kNeg : 0;
  st.ypass = 0;
  st.dprev = (coh ? ld_coherent(lbi + 2 * min(r0, a.m)) : a.lb[min(r0, a.m)].x) * (1 << SH) + TAG;
}

// outputs of a finished strip: right border (column stripes), SW row bests, the final score
template <int KIND, int K>
__device__ __forceinline__ void strip_end(const Lane<KIND, K>& st, const FillArgs& a, const int s, const int rowg, const int r0) {
  constexpr int SH = 2;
  if (a.rb) {  // the lane's rows froze at the stripe's last column
#pragma unroll
    for (int k = 0; k < K; ++k)
      if (r0 + k < a.m) {
        if (a.chunk_done) {  // read by a send that starts while this kernel is still running
          int* o = reinterpret_cast<int*>(a.rb + r0 + k + 1);
          st_coherent(o, st.H[k] >> SH);
          st_coherent(o + 1, st.E[k] >> SH);
        } else {
          a.rb[r0 + k + 1] = make_int2(st.H[k] >> SH, st.E[k] >> SH);
        }
      }
  }
  if (KIND == KIND_SW) {
#pragma unroll
    for (int k = 0; k < K; ++k)
      if (r0 + k < a.m) {
        a.rowbest[r0 + k] = st.best[k] >> SH;
        a.rowbestc[r0 + k] = st.bestc[k];
      }
  } else if (s == a.S - 1) {
    const int rl = a.m - 1;
    if (rl / K == rowg) {
#pragma unroll
      for (int k = 0; k < K; ++k)
        if (k == rl % K) *a.score_out = st.H[k] >> SH;
    }
  }
}

template <int KIND, int K, int FLAGS>
__global__ void __launch_bounds__(64 * kWavesPerBlock) fill_tile(const FillArgs a) {
  constexpr bool DUMP = (FLAGS & FLAG_DUMP) != 0;
  using G = Geo<KIND, K>;
  extern __shared__ uint32_t lds_all[];
  const int lane = threadIdx.x & 63;
  const int wave = threadIdx.x >> 6;
  uint32_t* lds = lds_all + wave * (a.alpha * 64 * G::PW);
  const int s = a.s_lo + blockIdx.x * kWavesPerBlock + wave;
  if (s > a.s_hi) return;
  const int q = a.d - s;
  const int qmin = (64 * s) / a.C;
  const int qmax = (64 * s + a.n + 62) / a.C;
  if (s >= a.S || q < qmin || q > qmax) return;
  const int T0 = q * a.C - 64 * s;  // strip-local step at the tile start (<0: padding before the strip starts)
  const int rowg = s * 64 + lane;   // global row-group index
  const int r0 = rowg * K;          // first owned row (0-based)

  uint32_t xr[K];
  if (KIND == KIND_LCS) {
#pragma unroll
    for (int k = 0; k < K; ++k) xr[k] = a.xpad[(size_t)r0 + k];
  } else {
#pragma unroll
    for (int k = 0; k < K; ++k) xr[k] = 0;
    const uint32_t* gp = a.prof + (size_t)s * a.alpha * 64 * G::PW;
    for (int i = lane; i < a.alpha * 64 * G::PW; i += 64) lds[i] = gp[i];
    wave_lds_fence();
  }
  const char* ldsLane = reinterpret_cast<const char*>(lds) + lane * 4 * G::PW;

  int dprow[K];  // DUMP: offset of the lane's rows in the dp matrix, -1 for padding rows
#pragma unroll
  for (int k = 0; k < K; ++k) dprow[k] = (DUMP && r0 + k < a.m) ? (r0 + k + 1) * (a.n + 1) : -1;

  Lane<KIND, K> st;
  int* sp = a.state + ((size_t)s * G::NF) * 64 + lane;
  if (q == qmin) {
    strip_begin<KIND, K, FLAGS>(st, a, r0);
  } else {
    int f = 0;
#pragma unroll
    for (int k = 0; k < K; ++k) st.H[k] = sp[64 * f++];
    if (KIND != KIND_LCS) {
#pragma unroll
      for (int k = 0; k < K; ++k) st.E[k] = sp[64 * f++];
    }
    st.passH = sp[64 * f++];
    if (KIND != KIND_LCS) st.passF = sp[64 * f++];
    st.ypass = sp[64 * f++];
    st.dprev = sp[64 * f++];
    if (KIND == KIND_SW) {
#pragma unroll
      for (int k = 0; k < K; ++k) st.best[k] = sp[64 * f++];
#pragma unroll
      for (int k = 0; k < K; ++k) st.bestc[k] = sp[64 * f++];
    }
  }

  if (KIND != KIND_LCS) {
#pragma unroll
    for (int k = 0; k < K; ++k) st.Hg[k] = add_gap<(FLAGS & FLAG_FRAC) != 0>(st.H[k], a.go, a.fo) + 1;
  }

#pragma unroll 1
  for (int b = 0; b < a.C / 64; ++b) {
    const int Tb = T0 + 64 * b;
    if (Tb < 0) continue;
    if (Tb >= a.n + 63) break;
    const bool interior = !DUMP && (Tb >= 64) && (Tb + 63 < a.n);
    int ibH, ibF, yb;
    load_block_inputs<KIND, K, FLAGS>(a, s, Tb, lane, ibH, ibF, yb);
    if (interior)
      run_block<KIND, K, false, FLAGS>(st, a, s, Tb, lane, ldsLane, xr, dprow, ibH, ibF, yb);
    else
      run_block<KIND, K, true, FLAGS>(st, a, s, Tb, lane, ldsLane, xr, dprow, ibH, ibF, yb);
  }

  if (q == qmax) {
    strip_end<KIND, K>(st, a, s, rowg, r0);
  } else {
    int f = 0;
#pragma unroll
    for (int k = 0; k < K; ++k) sp[64 * f++] = st.H[k];
    if (KIND != KIND_LCS) {
#pragma unroll
      for (int k = 0; k < K; ++k) sp[64 * f++] = st.E[k];
    }
    sp[64 * f++] = st.passH;
    if (KIND != KIND_LCS) sp[64 * f++] = st.passF;
    sp[64 * f++] = st.ypass;
    sp[64 * f++] = st.dprev;
    if (KIND == KIND_SW) {
#pragma unroll
      for (int k = 0; k < K; ++k) sp[64 * f++] = st.best[k];
#pragma unroll
      for (int k = 0; k < K; ++k) sp[64 * f++] = st.bestc[k];
    }
  }
}

// (The kernels that are not templates have internal linkage: this header is included by seqalign_hip.hip and by
// sa_lcsbits.hip.)
// ---- persistent strip pipeline ---------------------------------------------------------------
// One launch for the whole fill.  A wave draws strips from a ticket counter in ascending order and
// sweeps each one in sub-blocks of 16 steps.  Lane 0 of strip s consumes, at its step t, the bottom-row
// record of column t of strip s-1, which lane 63 of that strip produces at ITS step t + 63 and hands
// down (ho_store, lanes 48..63 of the rotating out-buffers) at the end of the 16-step sub-block that
// contains that step: consecutive strips run 80 steps + one store-to-load latency apart (they ran two
// 64-step blocks + a flag round trip apart when the hand-off was per block behind a progress word), and
// nothing is saved or restored between tiles.  The records of the NEXT sub-block are requested
// right after those of the current one have been installed, so the load has 16 steps to land; if the
// producer has not written them yet (their generation bit says so), the wave re-reads them (bounded).
// Deadlock-free for any grid size: tickets are only handed to running waves, in order, so the lowest
// unfinished strip is always held by a resident wave whose predecessor is complete.  Every spin is
// bounded (spin_limit); a wave that gives up raises *err and every waiting wave leaves when it sees it.

// How many sub-blocks ahead the records are requested.  An agent-scope load takes about 0.85 us on an idle MI355X
// (measured: one strip alone, tools/perf_step.py); a 16-step sub-block of NW / SW lasts 0.73 us (K = 1) .. 2.6 us
// (K = 8) and one sub-block ahead is enough (two cost K = 8 its registers: 164 -> 179 ns per step); the LCS cell is
// cheaper (0.56 us per sub-block at K = 1) and asks two sub-blocks ahead.  Must divide 4 (sub-blocks per block).
constexpr int kSubBlocksPerBlock = 4;  // a 64-step block is four 16-step sub-blocks (run_block_p expands them in line)
template <int KIND, int K>
struct HandoffDepth {
  static constexpr int D = KIND == KIND_LCS ? 2 : 1;
  // pend[(sub + 1) % D] is the request issued D sub-blocks earlier only if the block's sub-block count is a multiple of D
  static_assert(kSubBlocksPerBlock % D == 0, "the request distance must divide the sub-blocks of a block");
};
// The generation scheme of the hand-off records (ho_store) holds while the ring has at least two slots more than the
// launch has waves, or one slot per strip (+ the slot of matrix row 0): checked on the host before every launch.
inline bool handoff_ring_ok(int ring, int strips, int grid_waves) { return ring >= strips + 1 || ring >= grid_waves + 2; }

// Take the hand-off records of the sub-block that starts at step Ts (lanes 0..15: columns Ts .. Ts+15), requested
// D sub-blocks earlier into `cur`, install them in the in-buffers and request those D sub-blocks further on into
// the same registers.  A record is valid when it carries the generation of the strip above; columns >= n are never
// produced (and never used: their cells are inactive).  Returns false when the bounded wait expired.
#ifdef SA_DEBUG_HANDOFF
__device__ int g_dbg_episodes, g_dbg_rereads;  // per wave would be better; one strip per wave at a time is close enough
#endif
template <int KIND, int K>
__device__ __forceinline__ bool handoff_acquire(const FillArgs& a, int2& cur, const int2* rd, const int Ts, const int lane,
                                                const int gen_r, int& ibH, int& ibF, int* dbgc = nullptr) {
  // no short-circuit: one straight-line test on all lanes, so that the wait for the load sits on every path
  const int need = (lane < 16) & (Ts + lane < a.n);
  if (__builtin_expect(__ballot(need & ho_stale(cur.x, gen_r)) != 0ull, 0)) {
    // every way out of this loop has waited for its load (the test comes first), so that where the paths join the
    // compiler knows that nothing loaded is pending and leaves the fast path's counted waits alone
    int spins = 0;
    while (true) {
      __builtin_amdgcn_s_sleep(1);
      cur = ho_load(rd + Ts);
      if (__ballot(need & ho_stale(cur.x, gen_r)) == 0ull) break;
      if (++spins > a.spin_limit || ((spins & 255) == 0 && ld_coherent(a.err) != 0)) return false;
    }
#ifdef SA_DEBUG_HANDOFF
    if (dbgc) dbgc[0] += 1, dbgc[1] += spins;
#endif
  }
  ibH = ho_value(cur.x);
  ibF = cur.y;
  cur = ho_load(rd + Ts + 16 * HandoffDepth<KIND, K>::D);  // in flight during the next D sub-blocks
  return true;
}

// 64 steps (four sub-blocks expanded in line) of one strip.  On entry ibH / ibF hold the records of the block's
// first sub-block (handoff_acquire), on return those of the next block's first.  The records of the following
// sub-block are taken right after a sub-block's 16 steps and BEFORE its own stores are issued: whatever the wait
// for the request has to wait for (vmcnt counts loads and stores together, in order, and the compiler waits for
// everything where the re-read path joins) was issued at least 16 steps earlier.
// ny: the next block's 64 target symbols, requested during sub-block 2.  Returns false when a bounded wait expired.
template <int KIND, int K, bool BND, int FLAGS>
__device__ __forceinline__ bool run_block_p(Lane<KIND, K>& st, const FillArgs& a, const int s, const int Tb, const int lane,
                                            const char* ldsLane, const uint32_t (&xr)[K], int* dprow, int yb, int& ny,
                                            int2 (&pend)[HandoffDepth<KIND, K>::D], int& ibH, int& ibF, int& obH, int& obF,
                                            const int2* rd, int2* wr, int2* sink, const int gen_r, const int gen_w, int* dbgc = nullptr) {
  constexpr int D = HandoffDepth<KIND, K>::D;
  uint32_t* tdst = a.trace + (((size_t)(s - a.trace_s0) * a.W + (Tb >> 4)) * 64 + lane) * K;
  // the four sub-blocks, expanded by hand: a loop this size (16 rows per lane) is not always unrolled in full, and
  // pend[] then turns into a scratch array
  auto sub_block = [&](auto subc) -> bool {
    constexpr int sub = decltype(subc)::value;
    const int Ts = Tb + 16 * sub;
    uint32_t tacc[K], sbk[K], skp[K];
#pragma unroll
    for (int k = 0; k < K; ++k) tacc[k] = 0, sbk[k] = 0, skp[k] = 0;
    const int cbase = Ts - lane;
#pragma unroll
    for (int u = 0; u < 16; ++u)
      lane_step<KIND, K, BND, FLAGS>(st, ibH, ibF, yb, obH, obF, tacc, u, ldsLane, xr, a, cbase + u, dprow, sbk, skp);
    if (BND && EdgeMode<KIND, FLAGS>::kMasked) {
      // frozen lanes skipped their cells (lane_step): a row's parked key may still wait to be folded, and the codes of a
      // lane that froze after `act` steps of this sub-block sit in the top 2 * act bits of its words instead of the bottom
      if (KIND == KIND_SW) {
#pragma unroll
        for (int k = 0; k < K; ++k) sbk[k] = max(sbk[k], skp[k]);
      }
      if (FLAGS & FLAG_TAIL) {
        const int sh = 2 * (16 - min(max(a.n - cbase, 0), 16));  // 32 (no active step: the words are still 0) shifts by 0
#pragma unroll
        for (int k = 0; k < K; ++k) tacc[k] >>= (sh & 31);
      }
    }
    if (KIND == KIND_SW) {
#pragma unroll
      for (int k = 0; k < K; ++k) {
        const int v = (int)(sbk[k] >> 4) << 2;  // the 4 * score of the sub-block's best cell in this row
        const bool better = v > st.best[k];     // strict: earlier sub-blocks hold the smaller columns
        st.best[k] = better ? v : st.best[k];
        st.bestc[k] = better ? cbase + 15 - (int)(sbk[k] & 15u) : st.bestc[k];
      }
    }
    if (!handoff_acquire<KIND, K>(a, pend[(sub + 1) % D], rd, Ts + 16, lane, gen_r, ibH, ibF, dbgc)) return false;
    if (sub == 2) ny = a.yenc[Tb + 64 + lane];
    if (!(FLAGS & FLAG_NOTRACE)) store_trace<K>(tdst, tacc);
    tdst += 64 * K;
    // The out-buffers are rotated once per step: lane 48 + u holds the record of step Ts + u, i.e. of the
    // bottom-row column Ts + u - 63 (index p = Ts + u + 1).  Interior blocks let every lane store -- the lower 48 into
    // the slot's unused entry 0 (`sink`) -- so that the store sits on every path and the compiler can count it in
    // its waits; boundary blocks predicate it.
    if (!BND)
      ho_store(lane >= 48 ? wr + Ts : sink, obH, obF, gen_w);
    else if (lane >= 48)
      ho_store(wr + Ts, obH, obF, gen_w);
    return true;
  };
  return sub_block(std::integral_constant<int, 0>()) && sub_block(std::integral_constant<int, 1>()) &&
         sub_block(std::integral_constant<int, 2>()) && sub_block(std::integral_constant<int, 3>());
}

template <int KIND, int K, int FLAGS>
__global__ void __launch_bounds__(64 * kWavesPerBlock) fill_persistent(const FillArgs a) {
  using G = Geo<KIND, K>;
  constexpr int PF = FLAGS | FLAG_PERSIST;
  constexpr bool YSCALE = KIND != KIND_LCS || (FLAGS & FLAG_LPROF);
  extern __shared__ uint32_t lds_all[];
  const int lane = threadIdx.x & 63;
  uint32_t* lds = lds_all + (threadIdx.x >> 6) * (a.alpha * 64 * G::PW);
  const int NB = (a.n + 63 + 63) / 64;  // blocks with at least one active lane
  const char* ldsLane = reinterpret_cast<const char*>(lds) + lane * 4 * G::PW;
  int dprow[K];
#pragma unroll
  for (int k = 0; k < K; ++k) dprow[k] = -1;

  while (true) {
    int s = 0;
    if (lane == 0) s = atomicAdd(a.ticket, 1);
    s = a.s_first + __builtin_amdgcn_readfirstlane(s);
    if (s > a.s_last) break;
    const int rowg = s * 64 + lane;
    const int r0 = rowg * K;
    uint32_t xr[K];
    if (KIND == KIND_LCS && !(FLAGS & FLAG_LPROF)) {
#pragma unroll
      for (int k = 0; k < K; ++k) xr[k] = a.xpad[(size_t)r0 + k];
    } else {
#pragma unroll
      for (int k = 0; k < K; ++k) xr[k] = 0;
      wave_lds_fence();  // the previous strip's LDS reads are done
      const uint32_t* gp = a.prof + (size_t)s * a.alpha * 64 * G::PW;
      for (int i = lane; i < a.alpha * 64 * G::PW; i += 64) lds[i] = gp[i];
      wave_lds_fence();
    }
    bool dead = false;
    if (a.rows_in) {  // column stripes: the strip's rows of the left border must have arrived
      const int need = min(a.m, (s + 1) * 64 * K);
      int spins = 0;
      if (ld_coherent(a.rows_in) < need) {
        const long long t0 = wall_clock64();
        while (ld_coherent(a.rows_in) < need) {
          __builtin_amdgcn_s_sleep(16);
          if (++spins > a.spin_limit || ((spins & 255) == 0 && ld_coherent(a.err) != 0)) {
            dead = true;
            break;
          }
        }
        if (lane == 0 && a.wait_ticks) atomicAdd(a.wait_ticks, (unsigned long long)(wall_clock64() - t0));
      }
      if (dead) {
        if (lane == 0) st_coherent(a.err, 1);
        break;
      }
      asm volatile("" ::: "memory");  // border loads stay behind the poll
    }
    Lane<KIND, K> st;
    strip_begin<KIND, K, FLAGS>(st, a, r0, a.rows_in != nullptr);
    if (KIND != KIND_LCS) {
#pragma unroll
      for (int k = 0; k < K; ++k) st.Hg[k] = add_gap<(FLAGS & FLAG_FRAC) != 0>(st.H[k], a.go, a.fo) + 1;
    }
    // slot and generation of the records this strip reads / writes (see ho_store); matrix row 0 is "strip -1"
    const int gen_r = ho_gen((s / a.ring) % 3), gen_w = ho_gen(((s + 1) / a.ring) % 3);
    // lanes 0..15 of sub-block Ts read the record of column Ts + lane; the other lanes read lane 15's again (same cache
    // line: no traffic of their own, and the load stays one unpredicated instruction)
    const int2* rd = a.bot2 + (size_t)(s % a.ring) * a.NP + 64 + min(lane, 15);
    // lanes 48..63 of sub-block Ts write the record of step Ts + lane - 48; in interior blocks, where the store is
    // unpredicated, the other lanes write into the slot's entry 0, which nobody reads
    int2* const wr_slot_base = a.bot2 + (size_t)((s + 1) % a.ring) * a.NP;
    int2* wr = wr_slot_base + 1 + (lane - 48);
    int2 pend[HandoffDepth<KIND, K>::D];
#pragma unroll
    for (int j = 0; j < HandoffDepth<KIND, K>::D; ++j) pend[j] = ho_load(rd + 16 * j);
    int ny = a.yenc[lane];
    int ibH = 0, ibF = 0, obH = 0, obF = 0;
    int* dbgc = nullptr;
#ifdef SA_DEBUG_HANDOFF
    int dbg_counts[2] = {0, 0};
    dbgc = dbg_counts;
#endif
    dead = !handoff_acquire<KIND, K>(a, pend[0], rd, 0, lane, gen_r, ibH, ibF);
#ifdef SA_DEBUG_HANDOFF
    const long long dbg_t0 = wall_clock64();
#endif
#ifdef SA_DEBUG_HANDOFF
    int dbg_slot = -1;
    {
      const int picks[8] = {0, 1, 2, 3, a.S / 2, a.S / 2 + 1, a.S - 2, a.S - 1};
      for (int q = 0; q < 8; ++q)
        if (picks[q] == s) dbg_slot = q;
    }
#endif
#pragma unroll 1
    for (int b = 0; b < NB && !dead; ++b) {
#ifdef SA_DEBUG_HANDOFF
      if (a.dbgblk && dbg_slot >= 0 && lane == 0 && b < a.dbgblk_stride) a.dbgblk[(size_t)dbg_slot * a.dbgblk_stride + b] = wall_clock64();
#endif
      const int Tb = 64 * b;
      const int yb = YSCALE ? ny * G::ROWB : ny;
      const bool interior = (Tb >= 64) && (Tb + 63 < a.n);
      bool ok;
      if (interior)
        ok = run_block_p<KIND, K, false, PF>(st, a, s, Tb, lane, ldsLane, xr, dprow, yb, ny, pend, ibH, ibF, obH, obF, rd, wr, wr_slot_base, gen_r, gen_w, dbgc);
      else if (b == 0 && a.n > 63)  // the strip's first block of a matrix that is wider than it (Tb = 0 as a literal)
        ok = run_block_p<KIND, K, true, PF | FLAG_START>(st, a, s, 0, lane, ldsLane, xr, dprow, yb, ny, pend, ibH, ibF, obH, obF, rd, wr, wr_slot_base, gen_r, gen_w, dbgc);
      else if (b > 0)               // the last blocks: every lane is past column 0
        ok = run_block_p<KIND, K, true, PF | FLAG_TAIL>(st, a, s, Tb, lane, ldsLane, xr, dprow, yb, ny, pend, ibH, ibF, obH, obF, rd, wr, wr_slot_base, gen_r, gen_w, dbgc);
      else
        ok = run_block_p<KIND, K, true, PF>(st, a, s, Tb, lane, ldsLane, xr, dprow, yb, ny, pend, ibH, ibF, obH, obF, rd, wr, wr_slot_base, gen_r, gen_w, dbgc);
      dead = !ok;
    }
#ifdef SA_DEBUG_HANDOFF
    if (a.dbg && lane == 0) {
      a.dbg[4 * s + 0] = dbg_t0;
      a.dbg[4 * s + 1] = wall_clock64();
      a.dbg[4 * s + 2] = dbg_counts[0];
      a.dbg[4 * s + 3] = dbg_counts[1];
    }
#endif
    if (dead) {
      if (lane == 0) st_coherent(a.err, 1);
      break;
    }
    strip_end<KIND, K>(st, a, s, rowg, r0);
    if (a.chunk_done) {  // the strip's right-border rows are out: count it into its chunk
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      if (lane == 0) __hip_atomic_fetch_add(a.chunk_done + s / a.strips_per_chunk, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
}

// host-initialised records of matrix row 0 (slot 0, generation 0): H[0][c+1] at index c + 64
// h4 given: the host's chain; else H[0][c+1] = hconst + c * hstep in register scale (NW with integral penalties:
// 4 (go + c ge), main.cpp:719-725; SW: 0; LCS: the tag)
static __global__ void init_row0_records(int2* __restrict__ dst, const int* __restrict__ h4, int n, int hconst, int f, int hstep = 0) {
  for (int c = blockIdx.x * blockDim.x + threadIdx.x; c < n; c += gridDim.x * blockDim.x) {
    const int h = h4 ? h4[c] : hconst + c * hstep;
    dst[64 + c] = make_int2(h & 0x3fffffff, f);
  }
}

// left border of the whole matrix for integral penalties, written on the device: row i holds (H[i][0], E[i][0]) =
// (h1 + (i - 1) hstep, e) for i >= 1 and (0, e) for row 0 -- NW: go + (i - 1) max(go, ge), the closed form of the
// chain main.cpp:601-606 builds when nothing is truncated; SW / LCS: zeros
static __global__ void init_left_border(int2* __restrict__ lb, int m, int h1, int hstep, int e) {
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i <= m; i += gridDim.x * blockDim.x)
    lb[i] = make_int2(i == 0 ? 0 : h1 + (i - 1) * hstep, e);
}

// Column stripes with the persistent kernel: the two ends of a border hand-off, each a one-wave
// kernel on the caller's receive / send stream, so that RCCL transfers overlap the running fill.
// stripe_signal follows the receive of chunk k in stream order and raises rows_in;
// stripe_wait precedes the send of chunk k and returns once all strips of the chunk have counted
// themselves into chunk_done[k] (bounded; raises *err when it gives up).
static __global__ void stripe_signal(int* rows_in, int rows) {
  if (threadIdx.x == 0) st_coherent(rows_in, rows);
}
static __global__ void stripe_wait(const int* cnt, int target, int* err, int spin_limit) {
  if (threadIdx.x != 0) return;
  int spins = 0;
  while (ld_coherent(cnt) < target) {
    __builtin_amdgcn_s_sleep(32);
    if (++spins > spin_limit) {
      st_coherent(err, 1);
      break;
    }
    if ((spins & 63) == 0 && ld_coherent(err) != 0) break;
  }
}

// ---- query profile: prof[s][sym][lane] = the lane's K substitution scores for `sym` ----
// xidx is the table index per query residue (padded rows get score 0).
template <int K>
__global__ void build_profile(const uint8_t* __restrict__ xidx, int m, const int8_t* __restrict__ sub, int alpha,
                              int S, uint32_t* __restrict__ prof) {
  constexpr int PW = (K <= 4) ? 1 : K / 4;
  const size_t total = (size_t)S * alpha * 64;
  for (size_t g = (size_t)blockIdx.x * blockDim.x + threadIdx.x; g < total; g += (size_t)gridDim.x * blockDim.x) {
    const int lane = (int)(g & 63);
    const int sym = (int)((g >> 6) % alpha);
    const int s = (int)((g >> 6) / alpha);
    const int r0 = (s * 64 + lane) * K;
    uint32_t w[PW];
#pragma unroll
    for (int i = 0; i < PW; ++i) w[i] = 0;
#pragma unroll
    for (int k = 0; k < K; ++k) {
      int v = 0;
      if (r0 + k < m) v = sub[(int)xidx[r0 + k] * alpha + sym];
      v = 4 * v + 2;  // tagged M candidate (host guarantees |sigma| <= 31)
      w[k / 4] |= (uint32_t)(uint8_t)(int8_t)v << (8 * (k & 3));
    }
#pragma unroll
    for (int i = 0; i < PW; ++i) prof[g * PW + i] = w[i];
  }
}

// ---- SW: reduce per-row bests to the first row-major maximum (main.cpp:1124-1127) ----
static __global__ void sw_reduce_max(const int* __restrict__ rowbest, int m, int* __restrict__ best) {
  int v = 0;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < m; i += gridDim.x * blockDim.x) v = max(v, rowbest[i]);
  for (int o = 32; o > 0; o >>= 1) v = max(v, __shfl_xor(v, o));
  if ((threadIdx.x & 63) == 0 && v > 0) atomicMax(best, v);
}
static __global__ void sw_reduce_pos(const int* __restrict__ rowbest, const int* __restrict__ rowbestc, int m,
                              const int* __restrict__ best, unsigned long long* __restrict__ pos) {
  const int b = *best;
  if (b <= 0) return;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < m; i += gridDim.x * blockDim.x)
    if (rowbest[i] == b) atomicMin(pos, ((unsigned long long)(uint32_t)(i + 1) << 32) | (uint32_t)(rowbestc[i] + 1));
}

// ---- traceback ----
struct TraceArgs {
  int kind, m, n, K, W;
  int start_i, start_j;   // < 0: default start (NW/LCS: (m,n); SW: the reduced best cell)
  int stop_at_col0;       // column stripes (not the first): stop when j reaches 0 and hand i over
  int i_min;              // banded recomputation: `trace` holds the codes of matrix rows i_min+1 .. only; the
                          // walk stops when it reaches row i_min (with j > 0) and hands (i, j) over
  long long nops0;        // moves already written to `ops` by the walks over the bands below
  const uint32_t* trace;
  // bit-vector LCS (sa_lcsbits.hip): `trace` holds one dword per (strip, step, lane) = 32 rows of one column of vertical
  // differences, W = steps per strip; xs / ys are the symbols per row / column (a diagonal move is xs == ys)
  const uint8_t* xs;
  const uint8_t* ys;
  const int* sw_best;                  // SW: best score (device)
  const unsigned long long* sw_pos;    // SW: (i<<32)|j of the end cell (device)
  unsigned long long* ops2;            // out: the moves, two bits each, 32 to a word (expand_ops makes bytes of them)
  long long* hdr;                      // out: n_ops, end_i, end_j, start_i, start_j
};

// Walker: one wavefront follows the pointers RUN by RUN.  At cell (i,j) lane d looks at the
// three cells d steps away along the diagonal, the row and the column; the direction stored at
// (i,j) selects one of the three ballots and count-trailing-ones gives the length of the run of
// identical moves, which the lanes then emit as bytes in parallel.  One memory round trip per
// run instead of one per move (the three gathers are in flight together).
__device__ __forceinline__ uint32_t trace_code(const uint32_t* __restrict__ trace, int i, int j, int lgK, int W, bool tagged) {
  const int r = i - 1, c = j - 1;
  const int g = r >> lgK, k = r & ((1 << lgK) - 1);
  const int s = g >> 6, l = g & 63;
  const int t = c + l;
  const size_t idx = ((((size_t)s * W + (t >> 4)) * 64 + l) << lgK) + k;
  const uint32_t raw = (trace[idx] >> (2 * (t & 15))) & 3u;
  // NW / SW cells store the winner's tag (M 2, E 1, F 0, SW zero cell 3); LCS: diagonal 3, up 2, left 1
  return tagged ? (raw == 3u ? 3u : 2u - raw) : (raw == 3u ? 0u : raw);
}

// A 64-row window of codes held in registers: lane d keeps three consecutive trace dwords (48
// skewed steps) of matrix row i-d, centred on the cell the pure diagonal through the entry cell
// would visit.  The path is then followed through all 64 rows without touching memory; an offset
// `o` (left moves shift it by -1, up moves by +1) says how far the path has drifted from that
// diagonal, and 48 positions tolerate |o| <= 16.
struct TraceWindow {
  uint32_t w0, w1, w2;
  int pb;  // position of the diagonal cell inside the 48-position window (16..31)
};

__device__ __forceinline__ TraceWindow load_trace_window(const uint32_t* __restrict__ trace, int i, int j0, int d, int lgK, int W) {
  const int r = i - d - 1, c = j0 - d - 1;
  const int g = r >> lgK, k = r & ((1 << lgK) - 1);
  const int s = g >> 6, l = g & 63;
  const int t = c + l;
  const int wi = t >> 4;
  // one 64-bit index (32 x 32 -> 64 multiply: s and W are non-negative), its two neighbours a word stride away: the three
  // gathers of a window cost ~15 instructions of address arithmetic instead of ~45 (the walker is bound by instruction issue)
  const size_t idx = ((((size_t)(uint32_t)s * (uint32_t)W + (uint32_t)wi) * 64 + (uint32_t)l) << lgK) + (uint32_t)k;
  const ptrdiff_t stride = (ptrdiff_t)64 << lgK;
  const uint32_t* p = trace + idx;
  TraceWindow w;
  w.w0 = p[-stride];  // wi >= 1: the window's columns start at 17 or later (window_ok)
  w.w1 = p[0];
  w.w2 = p[wi + 1 <= W - 1 ? stride : 0];
  w.pb = 16 + (t & 15);
  return w;
}

// op code (0 diag, 1 left, 2 up, 3 stop) at window position q, 4 if q is outside the window
__device__ __forceinline__ uint32_t window_code(const TraceWindow& w, int q, bool tagged) {
  const uint32_t dw = q < 16 ? w.w0 : (q < 32 ? w.w1 : w.w2);
  const uint32_t raw = (dw >> (2 * (q & 15))) & 3u;
  const uint32_t op = tagged ? (raw == 3u ? 3u : 2u - raw) : raw;
  return (q >= 0 && q < 48) ? op : 4u;
}

// the window path needs 64 rows above and room for 16 columns of drift on both sides
__device__ __forceinline__ bool window_ok(int i, int j) { return i >= 64 && j >= 81; }

constexpr int kTraceStageWords = 4096;  // 64-bit words of packed moves (32 each) staged in LDS between flushes

// The moves leave the walker packed two bits each (0 diagonal, 1 left, 2 up), 32 to a 64-bit word, in traceback
// order; expand_ops turns them into the one-byte-per-move list of the C-ABI.  Everything here is wave-uniform and
// lives in scalar registers: a move costs a shift and an OR, a diagonal run of any length one addition.
struct MoveEmitter {
  unsigned long long acc;   // the word being filled
  int k;                    // moves in acc, 0..31
  int staged;               // complete words waiting in LDS
  long long words_out;      // complete words already in global memory
  unsigned long long* stage;
  unsigned long long* out;
  int lane;

  __device__ __forceinline__ void push_word() {
    stage[staged] = acc;  // every lane stores the same word to the same place: no divergent branch
    ++staged;
    acc = 0;
    k = 0;
    if (staged == kTraceStageWords) flush();
  }
  __device__ __forceinline__ void flush() {
    wave_lds_fence();  // the staged words were written by lane 0
    for (int q = lane; q < staged; q += 64) out[words_out + q] = stage[q];
    words_out += staged;
    staged = 0;
    wave_lds_fence();  // later staging writes stay behind these reads
  }
  __device__ __forceinline__ void emit(uint32_t op) {
    acc |= (unsigned long long)op << (2 * k);
    if (++k == 32) push_word();
  }
  __device__ __forceinline__ void emit_run(uint32_t op, long long L) {
    const unsigned long long pat = op == 0 ? 0ull : (op == 1 ? 0x5555555555555555ull : 0xAAAAAAAAAAAAAAAAull);
    while (L > 0) {
      const int take = (int)min((long long)(32 - k), L);
      const unsigned long long bits = take == 32 ? pat : (pat & ((1ull << (2 * take)) - 1ull));
      acc |= bits << (2 * k);
      k += take;
      L -= take;
      if (k == 32) push_word();
    }
  }
  __device__ __forceinline__ long long total() const { return (words_out + staged) * 32 + k; }
  __device__ __forceinline__ void finish() {  // the partial word goes out too (the next band continues it)
    flush();
    if (lane == 0) out[words_out] = acc;
  }
};

// ---- bit-vector LCS (sa_lcsbits.hip): the same walker over one bit per cell ----
// The fill stores, per column, the vertical differences of the L matrix as bit vectors: bit (r & 31) of dword
// [strip][t / 4][lane][t % 4], t = step c + kBitsSkew * lane (row r 0-based in word g = r >> 5 = strip * 64 + lane, column c 0-based) is CLEAR iff
// L[r+1][c+1] = L[r][c+1] + 1.  The reference's pointer at a cell (main.cpp:1577-1587) follows from it: diagonal iff
// the symbols match, else up iff L[i-1][j] >= L[i][j-1], which in a mismatch cell (L[i][j] = max of the two) is
// "the vertical difference is 0", i.e. the bit is SET; else left.
constexpr int kBitsStripPad = 0;  // dwords between the records of consecutive strips (17 x 256 B, against channel hot spots: no change)
constexpr int kBitsSkew = 2;  // the fill's skew (sa_lcsbits.hip): lane l meets column c at step c + kBitsSkew * l
__device__ __forceinline__ uint32_t bits_word(const TraceArgs& a, int g, int c) {
  const int sidx = g >> 6, l = g & 63;
  // the strip's base is scalar in the window loads (g is wave-uniform there); inside a strip 32-bit dword indices do
  // (W * 64 < 2^32: the host refuses longer targets), so the loads are scalar base + 32-bit lane offset
  // (banded recomputation: `trace` holds the strips from row i_min on, 2048 rows each)
  const uint32_t* sb = a.trace + (size_t)(sidx - (a.i_min >> 11)) * ((size_t)a.W * 64 + kBitsStripPad);
  const uint32_t t = (uint32_t)(c + kBitsSkew * l);
  return sb[((t & ~3u) << 6) + (uint32_t)(l * 4) + (t & 3u)];
}
// op code (0 diagonal, 1 left, 2 up) of cell (i, j), 1-based
__device__ __forceinline__ uint32_t bits_code(const TraceArgs& a, int i, int j) {
  const int r = i - 1, c = j - 1;
  if (a.xs[r] == a.ys[c]) return 0u;
  return ((bits_word(a, r >> 5, c) >> (r & 31)) & 1u) ? 2u : 1u;
}
// The 64-row window of the walker, by COLUMNS: lane q keeps rows i-63 .. i of column j0 - 79 + q (a0..a2, three dwords
// because the rows need not start on a word) and of column j0 - 15 + q (b0..b2; 31 of them are used), the symbol of
// both columns, and the symbol of row i - q.  96 columns = the 64 the diagonal through the entry cell crosses + 16 of
// drift on both sides, ~200 sectors per window (the 2-bit codes of the other kernels take about as many).
struct BitsWindow {
  uint32_t a0, a1, a2, b0, b1, b2;
  uint32_t ya, yb, xv;
};
__device__ __forceinline__ BitsWindow load_bits_window(const TraceArgs& a, int i, int j0, int q) {
  const int g0 = (i - 64) >> 5;
  // the last word the record holds: the walk's entry row bounds it (a band's record ends with the entry row's strip)
  const int glast = (((a.start_i >= 0 ? a.start_i : a.m) + 2047) / 2048) * 64 - 1;
  const int g1 = g0 + 1, g2 = min(g0 + 2, glast);
  const int cA = j0 - 80 + q, cB = min(j0 - 16 + q, a.n - 1);
  // Addresses as in bits_word, with what the six of a window share computed once (the walker is bound by instruction
  // issue: ~105 instructions of address arithmetic per window before, ~55 now): the three words are in the strip of the
  // first or in the one after it, and everything is unsigned 32-bit but for one 32 x 32 -> 64 multiply.
  const uint32_t rec = (uint32_t)a.W * 64u + (uint32_t)kBitsStripPad;  // dwords of a strip's record (< 2^32: host)
  const uint32_t* base0 = a.trace + (size_t)(uint32_t)((g0 >> 6) - (a.i_min >> 11)) * rec;
  const uint32_t* base1 = base0 + ((g1 >> 6) != (g0 >> 6) ? rec : 0u);
  const uint32_t* base2 = base0 + ((g2 >> 6) != (g0 >> 6) ? rec : 0u);
  const uint32_t l0 = (uint32_t)g0 & 63u, l1 = (uint32_t)g1 & 63u, l2 = (uint32_t)g2 & 63u;
  auto off = [](uint32_t c, uint32_t l) {
    const uint32_t t = c + (uint32_t)kBitsSkew * l;
    return ((t & ~3u) << 6) + (t & 3u) + l * 4u;
  };
  const uint32_t oa0 = off((uint32_t)cA, l0), oa1 = off((uint32_t)cA, l1), oa2 = off((uint32_t)cA, l2);
  uint32_t ob0, ob1, ob2;
  if (j0 + 47 <= a.n - 1) {  // (wave-uniform) no lane's second column is clamped: it is 64 columns = 16 groups of four steps further on
    ob0 = oa0 + 4096u, ob1 = oa1 + 4096u, ob2 = oa2 + 4096u;
  } else {
    ob0 = off((uint32_t)cB, l0), ob1 = off((uint32_t)cB, l1), ob2 = off((uint32_t)cB, l2);
  }
  BitsWindow w;
  w.a0 = base0[oa0];
  w.a1 = base1[oa1];
  w.a2 = base2[oa2];
  w.b0 = base0[ob0];
  w.b1 = base1[ob1];
  w.b2 = base2[ob2];
  w.ya = a.ys[cA];
  w.yb = a.ys[cB];
  w.xv = a.xs[i - 1 - q];
  return w;
}
// ---- the scalar walk through one 64-row window (traceback_walk) ----
// State, all wave-uniform: dcur = rows of the window already left behind, o = drift from the window's diagonal
// (-16..15), k / acc = the emitter's word being filled.  diagmask[q], q = o + 16: the 64-row mask "the cell at drift
// q - 16 is a diagonal move"; Flo / Fhi: lane r holds row r's 32 two-bit codes (drift -16 first).  Per round: the
// diagonal run from (dcur, o) is the run of ones in mask[o] from bit dcur (s_ff1 of its complement) and costs one
// addition to k (zero codes); then ONE gap move: its code comes out of the row's scalar copy, goes into acc, and moves
// dcur / o.  Leaves -- with everything consistent, the caller looks at k, dcur, o and the returned code -- when the
// window is done (dcur >= 64), an SW stop code (3) turns up, or the path drifted out of the window.  A full word of 32
// moves is staged right here (out of line: v_mov_b64 + ds_write_b64 at the LDS byte address `saddr`, which advances by
// 8): leaving the loop for it cost two exits per window, each ~350 cycles of branches and compiled bookkeeping for a
// lone wave.  vcc is the 64-bit scratch pair (inline asm cannot name the halves of a 64-bit operand).
// Round 3: the masks are built by 32 compares that write their 64-bit results straight into s[36:99]
// (v_cmp_eq_u32_e64 sdst / the SDWA form with a byte select for the bit-vector record) and are fetched in the round loop
// with s_movrels_b64 (M0 = 2 (o + 16), which is also the shift that brings the row's code at this drift down).  Rounds 1-2
// parked them in lanes 0..31 of two VGPRs: a compare, two wait states (gfx950 wants them between a VALU write of a
// scalar register and a v_writelane that reads it; the compiler does not look inside inline assembly) and two
// v_writelane per drift, two v_readlane per round.  Set-up and loop are ONE asm statement: the masks live in registers
// the compiler only knows as clobbered.  M0 is not used by anything else in this kernel (gfx9 DS instructions do not
// read it).  The bit-vector window: ca / cb, lane q holds the 64 row bits (bit d = row i - d) of column j0 - 79 + q /
// j0 - 15 + q; the gap move at (d, o) sits in column j0 - d + o = lane 79 - d + o of ca, or of cb when that is 64 or
// more (v_readlane takes the lane modulo 64).
#define SA_WALK_MASK_CLOBBERS "s36", "s37", "s38", "s39", "s40", "s41", "s42", "s43", "s44", "s45", "s46", "s47", "s48", "s49", "s50", "s51", "s52", "s53", "s54", "s55", "s56", "s57", "s58", "s59", "s60", "s61", "s62", "s63", "s64", "s65", "s66", "s67", "s68", "s69", "s70", "s71", "s72", "s73", "s74", "s75", "s76", "s77", "s78", "s79", "s80", "s81", "s82", "s83", "s84", "s85", "s86", "s87", "s88", "s89", "s90", "s91", "s92", "s93", "s94", "s95", "s96", "s97", "s98", "s99"
// CODES: 0 the LCS cell kernels' pointers (3 diagonal, 2 up, 1 left), 1 NW tags (M 2, E 1, F 0), 2 SW tags (+ 3: stop cell)
template <int CODES>
__device__ __forceinline__ uint32_t walk_window(const uint32_t Flo, const uint32_t Fhi, int& dcur, int& o, int& k, unsigned long long& acc,
                                                uint32_t& saddr) {
  uint32_t c = 0, t0, t1, L, vt;
  unsigned long long wtmp;
  if constexpr (CODES == 1) {
    asm volatile(
        "v_bfe_u32 %[vt], %[flo], 0, 2\n\t"
        "v_cmp_eq_u32_e64 s[36:37], %[vt], 2\n\t"
        "v_bfe_u32 %[vt], %[flo], 2, 2\n\t"
        "v_cmp_eq_u32_e64 s[38:39], %[vt], 2\n\t"
        "v_bfe_u32 %[vt], %[flo], 4, 2\n\t"
        "v_cmp_eq_u32_e64 s[40:41], %[vt], 2\n\t"
        "v_bfe_u32 %[vt], %[flo], 6, 2\n\t"
        "v_cmp_eq_u32_e64 s[42:43], %[vt], 2\n\t"
        "v_bfe_u32 %[vt], %[flo], 8, 2\n\t"
        "v_cmp_eq_u32_e64 s[44:45], %[vt], 2\n\t"
        "v_bfe_u32 %[vt], %[flo], 10, 2\n\t"
        "v_cmp_eq_u32_e64 s[46:47], %[vt], 2\n\t"
        "v_bfe_u32 %[vt], %[flo], 12, 2\n\t"
        "v_cmp_eq_u32_e64 s[48:49], %[vt], 2\n\t"
        "v_bfe_u32 %[vt], %[flo], 14, 2\n\t"
        "v_cmp_eq_u32_e64 s[50:51], %[vt], 2\n\t"
        "v_bfe_u32 %[vt], %[flo], 16, 2\n\t"
        "v_cmp_eq_u32_e64 s[52:53], %[vt], 2\n\t"
        "v_bfe_u32 %[vt], %[flo], 18, 2\n\t"
        "v_cmp_eq_u32_e64 s[54:55], %[vt], 2\n\t"
        "v_bfe_u32 %[vt], %[flo], 20, 2\n\t"
        "v_cmp_eq_u32_e64 s[56:57], %[vt], 2\n\t"
        "v_bfe_u32 %[vt], %[flo], 22, 2\n\t"
        "v_cmp_eq_u32_e64 s[58:59], %[vt], 2\n\t"
        "v_bfe_u32 %[vt], %[flo], 24, 2\n\t"
        "v_cmp_eq_u32_e64 s[60:61], %[vt], 2\n\t"
        "v_bfe_u32 %[vt], %[flo], 26, 2\n\t"
        "v_cmp_eq_u32_e64 s[62:63], %[vt], 2\n\t"
        "v_bfe_u32 %[vt], %[flo], 28, 2\n\t"
        "v_cmp_eq_u32_e64 s[64:65], %[vt], 2\n\t"
        "v_bfe_u32 %[vt], %[flo], 30, 2\n\t"
        "v_cmp_eq_u32_e64 s[66:67], %[vt], 2\n\t"
        "v_bfe_u32 %[vt], %[fhi], 0, 2\n\t"
        "v_cmp_eq_u32_e64 s[68:69], %[vt], 2\n\t"
        "v_bfe_u32 %[vt], %[fhi], 2, 2\n\t"
        "v_cmp_eq_u32_e64 s[70:71], %[vt], 2\n\t"
        "v_bfe_u32 %[vt], %[fhi], 4, 2\n\t"
        "v_cmp_eq_u32_e64 s[72:73], %[vt], 2\n\t"
        "v_bfe_u32 %[vt], %[fhi], 6, 2\n\t"
        "v_cmp_eq_u32_e64 s[74:75], %[vt], 2\n\t"
        "v_bfe_u32 %[vt], %[fhi], 8, 2\n\t"
        "v_cmp_eq_u32_e64 s[76:77], %[vt], 2\n\t"
        "v_bfe_u32 %[vt], %[fhi], 10, 2\n\t"
        "v_cmp_eq_u32_e64 s[78:79], %[vt], 2\n\t"
        "v_bfe_u32 %[vt], %[fhi], 12, 2\n\t"
        "v_cmp_eq_u32_e64 s[80:81], %[vt], 2\n\t"
        "v_bfe_u32 %[vt], %[fhi], 14, 2\n\t"
        "v_cmp_eq_u32_e64 s[82:83], %[vt], 2\n\t"
        "v_bfe_u32 %[vt], %[fhi], 16, 2\n\t"
        "v_cmp_eq_u32_e64 s[84:85], %[vt], 2\n\t"
        "v_bfe_u32 %[vt], %[fhi], 18, 2\n\t"
        "v_cmp_eq_u32_e64 s[86:87], %[vt], 2\n\t"
        "v_bfe_u32 %[vt], %[fhi], 20, 2\n\t"
        "v_cmp_eq_u32_e64 s[88:89], %[vt], 2\n\t"
        "v_bfe_u32 %[vt], %[fhi], 22, 2\n\t"
        "v_cmp_eq_u32_e64 s[90:91], %[vt], 2\n\t"
        "v_bfe_u32 %[vt], %[fhi], 24, 2\n\t"
        "v_cmp_eq_u32_e64 s[92:93], %[vt], 2\n\t"
        "v_bfe_u32 %[vt], %[fhi], 26, 2\n\t"
        "v_cmp_eq_u32_e64 s[94:95], %[vt], 2\n\t"
        "v_bfe_u32 %[vt], %[fhi], 28, 2\n\t"
        "v_cmp_eq_u32_e64 s[96:97], %[vt], 2\n\t"
        "v_bfe_u32 %[vt], %[fhi], 30, 2\n\t"
        "v_cmp_eq_u32_e64 s[98:99], %[vt], 2\n\t"
        "s_nop 3\n\t"
        "s_add_i32 %[o], %[o], 16\n\t"      // the loop keeps o + 16 (the mask's index)
        "Ltop_%=:\n\t"
        "s_lshl_b32 m0, %[o], 1\n\t"
        "s_sub_i32 %[t1], 64, %[d]\n\t"     // (also the wait state between the write of M0 and s_movrels)
        "s_movrels_b64 vcc, s[36:37]\n\t"
        "s_lshr_b64 vcc, vcc, %[d]\n\t"
        "s_ff0_i32_b64 %[L], vcc\n\t"       // -1 (no zero: d = 0 and 64 diagonal moves) is the largest unsigned value
        "s_min_u32 %[L], %[L], %[t1]\n\t"   // d = 64 can arrive here (an up move out of the last row): the shift was by 0 then
        "s_add_i32 %[k], %[k], %[L]\n\t"
        "s_add_i32 %[d], %[d], %[L]\n\t"
        "s_cmp_ge_i32 %[k], 32\n\t"
        "s_cbranch_scc1 Lword_%=\n\t"
        "Lback_%=:\n\t"
        "s_cmp_ge_i32 %[d], 64\n\t"
        "s_cbranch_scc1 Lout_%=\n\t"
        "v_readlane_b32 vcc_hi, %[fhi], %[d]\n\t"
        "v_readlane_b32 vcc_lo, %[flo], %[d]\n\t"
        "s_lshr_b64 vcc, vcc, m0\n\t"         // M0 is still 2 (o + 16): the row's code at this drift
        "s_andn2_b32 %[t1], 1, vcc_lo\n\t"     // tag F = 0: up (1), tag E = 1: left (0); NW has no other code off the diagonal
        "s_lshl1_add_u32 %[t0], %[k], %[t1]\n\t"
        "s_bitset1_b64 %[acc], %[t0]\n\t"       // the op is one set bit of the move's two: 2k (left) or 2k + 1 (up)
        "s_add_i32 %[k], %[k], 1\n\t"
        "s_add_i32 %[d], %[d], %[t1]\n\t"
        "s_lshl1_add_u32 %[o], %[t1], %[o]\n\t"
        "s_add_i32 %[o], %[o], -1\n\t"
        "s_cmp_lt_u32 %[o], 32\n\t"
        "s_cbranch_scc1 Ltop_%=\n\t"
        "s_branch Lout_%=\n\t"
        "Lword_%=:\n\t"                       // a word of 32 moves is full: stage it, start the next
        "v_mov_b64 %[wtmp], %[acc]\n\t"
        "ds_write_b64 %[saddr], %[wtmp]\n\t"
        "v_add_u32 %[saddr], 8, %[saddr]\n\t"
        "s_mov_b64 %[acc], 0\n\t"
        "s_sub_i32 %[k], %[k], 32\n\t"
        "s_cmp_ge_i32 %[k], 32\n\t"
        "s_cbranch_scc1 Lword_%=\n\t"
        "s_branch Lback_%=\n\t"
        "Lout_%=:\n\t"
        "s_add_i32 %[o], %[o], -16\n\t"
        : [d] "+s"(dcur), [o] "+s"(o), [k] "+s"(k), [acc] "+s"(acc), [c] "+s"(c), [t0] "=&s"(t0), [t1] "=&s"(t1), [L] "=&s"(L),
          [saddr] "+v"(saddr), [wtmp] "=&v"(wtmp), [vt] "=&v"(vt)
        : [flo] "v"(Flo), [fhi] "v"(Fhi)
        : "scc", "vcc", "memory", SA_WALK_MASK_CLOBBERS);
  } else if constexpr (CODES == 2) {
    asm volatile(
        "v_bfe_u32 %[vt], %[flo], 0, 2\n\t"
        "v_cmp_eq_u32_e64 s[36:37], %[vt], 2\n\t"
        "v_bfe_u32 %[vt], %[flo], 2, 2\n\t"
        "v_cmp_eq_u32_e64 s[38:39], %[vt], 2\n\t"
        "v_bfe_u32 %[vt], %[flo], 4, 2\n\t"
        "v_cmp_eq_u32_e64 s[40:41], %[vt], 2\n\t"
        "v_bfe_u32 %[vt], %[flo], 6, 2\n\t"
        "v_cmp_eq_u32_e64 s[42:43], %[vt], 2\n\t"
        "v_bfe_u32 %[vt], %[flo], 8, 2\n\t"
        "v_cmp_eq_u32_e64 s[44:45], %[vt], 2\n\t"
        "v_bfe_u32 %[vt], %[flo], 10, 2\n\t"
        "v_cmp_eq_u32_e64 s[46:47], %[vt], 2\n\t"
        "v_bfe_u32 %[vt], %[flo], 12, 2\n\t"
        "v_cmp_eq_u32_e64 s[48:49], %[vt], 2\n\t"
        "v_bfe_u32 %[vt], %[flo], 14, 2\n\t"
        "v_cmp_eq_u32_e64 s[50:51], %[vt], 2\n\t"
        "v_bfe_u32 %[vt], %[flo], 16, 2\n\t"
        "v_cmp_eq_u32_e64 s[52:53], %[vt], 2\n\t"
        "v_bfe_u32 %[vt], %[flo], 18, 2\n\t"
        "v_cmp_eq_u32_e64 s[54:55], %[vt], 2\n\t"
        "v_bfe_u32 %[vt], %[flo], 20, 2\n\t"
        "v_cmp_eq_u32_e64 s[56:57], %[vt], 2\n\t"
        "v_bfe_u32 %[vt], %[flo], 22, 2\n\t"
        "v_cmp_eq_u32_e64 s[58:59], %[vt], 2\n\t"
        "v_bfe_u32 %[vt], %[flo], 24, 2\n\t"
        "v_cmp_eq_u32_e64 s[60:61], %[vt], 2\n\t"
        "v_bfe_u32 %[vt], %[flo], 26, 2\n\t"
        "v_cmp_eq_u32_e64 s[62:63], %[vt], 2\n\t"
        "v_bfe_u32 %[vt], %[flo], 28, 2\n\t"
        "v_cmp_eq_u32_e64 s[64:65], %[vt], 2\n\t"
        "v_bfe_u32 %[vt], %[flo], 30, 2\n\t"
        "v_cmp_eq_u32_e64 s[66:67], %[vt], 2\n\t"
        "v_bfe_u32 %[vt], %[fhi], 0, 2\n\t"
        "v_cmp_eq_u32_e64 s[68:69], %[vt], 2\n\t"
        "v_bfe_u32 %[vt], %[fhi], 2, 2\n\t"
        "v_cmp_eq_u32_e64 s[70:71], %[vt], 2\n\t"
        "v_bfe_u32 %[vt], %[fhi], 4, 2\n\t"
        "v_cmp_eq_u32_e64 s[72:73], %[vt], 2\n\t"
        "v_bfe_u32 %[vt], %[fhi], 6, 2\n\t"
        "v_cmp_eq_u32_e64 s[74:75], %[vt], 2\n\t"
        "v_bfe_u32 %[vt], %[fhi], 8, 2\n\t"
        "v_cmp_eq_u32_e64 s[76:77], %[vt], 2\n\t"
        "v_bfe_u32 %[vt], %[fhi], 10, 2\n\t"
        "v_cmp_eq_u32_e64 s[78:79], %[vt], 2\n\t"
        "v_bfe_u32 %[vt], %[fhi], 12, 2\n\t"
        "v_cmp_eq_u32_e64 s[80:81], %[vt], 2\n\t"
        "v_bfe_u32 %[vt], %[fhi], 14, 2\n\t"
        "v_cmp_eq_u32_e64 s[82:83], %[vt], 2\n\t"
        "v_bfe_u32 %[vt], %[fhi], 16, 2\n\t"
        "v_cmp_eq_u32_e64 s[84:85], %[vt], 2\n\t"
        "v_bfe_u32 %[vt], %[fhi], 18, 2\n\t"
        "v_cmp_eq_u32_e64 s[86:87], %[vt], 2\n\t"
        "v_bfe_u32 %[vt], %[fhi], 20, 2\n\t"
        "v_cmp_eq_u32_e64 s[88:89], %[vt], 2\n\t"
        "v_bfe_u32 %[vt], %[fhi], 22, 2\n\t"
        "v_cmp_eq_u32_e64 s[90:91], %[vt], 2\n\t"
        "v_bfe_u32 %[vt], %[fhi], 24, 2\n\t"
        "v_cmp_eq_u32_e64 s[92:93], %[vt], 2\n\t"
        "v_bfe_u32 %[vt], %[fhi], 26, 2\n\t"
        "v_cmp_eq_u32_e64 s[94:95], %[vt], 2\n\t"
        "v_bfe_u32 %[vt], %[fhi], 28, 2\n\t"
        "v_cmp_eq_u32_e64 s[96:97], %[vt], 2\n\t"
        "v_bfe_u32 %[vt], %[fhi], 30, 2\n\t"
        "v_cmp_eq_u32_e64 s[98:99], %[vt], 2\n\t"
        "s_nop 3\n\t"
        "s_add_i32 %[o], %[o], 16\n\t"      // the loop keeps o + 16 (the mask's index)
        "Ltop_%=:\n\t"
        "s_lshl_b32 m0, %[o], 1\n\t"
        "s_sub_i32 %[t1], 64, %[d]\n\t"     // (also the wait state between the write of M0 and s_movrels)
        "s_movrels_b64 vcc, s[36:37]\n\t"
        "s_lshr_b64 vcc, vcc, %[d]\n\t"
        "s_ff0_i32_b64 %[L], vcc\n\t"       // -1 (no zero: d = 0 and 64 diagonal moves) is the largest unsigned value
        "s_min_u32 %[L], %[L], %[t1]\n\t"   // d = 64 can arrive here (an up move out of the last row): the shift was by 0 then
        "s_add_i32 %[k], %[k], %[L]\n\t"
        "s_add_i32 %[d], %[d], %[L]\n\t"
        "s_cmp_ge_i32 %[k], 32\n\t"
        "s_cbranch_scc1 Lword_%=\n\t"
        "Lback_%=:\n\t"
        "s_cmp_ge_i32 %[d], 64\n\t"
        "s_cbranch_scc1 Lout_%=\n\t"
        "v_readlane_b32 vcc_hi, %[fhi], %[d]\n\t"
        "v_readlane_b32 vcc_lo, %[flo], %[d]\n\t"
        "s_lshr_b64 vcc, vcc, m0\n\t"         // M0 is still 2 (o + 16): the row's code at this drift
        "s_and_b32 %[c], vcc_lo, 3\n\t"
        "s_cmp_eq_u32 %[c], 3\n\t"             // SW's stop cell
        "s_cbranch_scc1 Lout_%=\n\t"
        "s_sub_i32 %[t1], 1, %[c]\n\t"          // 1 for an up move (tag F = 0), 0 for a left move (tag E = 1)
        "s_lshl1_add_u32 %[t0], %[k], %[t1]\n\t"
        "s_bitset1_b64 %[acc], %[t0]\n\t"       // the op is one set bit of the move's two: 2k (left) or 2k + 1 (up)
        "s_add_i32 %[k], %[k], 1\n\t"
        "s_add_i32 %[d], %[d], %[t1]\n\t"
        "s_lshl1_add_u32 %[o], %[t1], %[o]\n\t"
        "s_add_i32 %[o], %[o], -1\n\t"
        "s_cmp_lt_u32 %[o], 32\n\t"
        "s_cbranch_scc1 Ltop_%=\n\t"
        "s_branch Lout_%=\n\t"
        "Lword_%=:\n\t"                       // a word of 32 moves is full: stage it, start the next
        "v_mov_b64 %[wtmp], %[acc]\n\t"
        "ds_write_b64 %[saddr], %[wtmp]\n\t"
        "v_add_u32 %[saddr], 8, %[saddr]\n\t"
        "s_mov_b64 %[acc], 0\n\t"
        "s_sub_i32 %[k], %[k], 32\n\t"
        "s_cmp_ge_i32 %[k], 32\n\t"
        "s_cbranch_scc1 Lword_%=\n\t"
        "s_branch Lback_%=\n\t"
        "Lout_%=:\n\t"
        "s_add_i32 %[o], %[o], -16\n\t"
        : [d] "+s"(dcur), [o] "+s"(o), [k] "+s"(k), [acc] "+s"(acc), [c] "+s"(c), [t0] "=&s"(t0), [t1] "=&s"(t1), [L] "=&s"(L),
          [saddr] "+v"(saddr), [wtmp] "=&v"(wtmp), [vt] "=&v"(vt)
        : [flo] "v"(Flo), [fhi] "v"(Fhi)
        : "scc", "vcc", "memory", SA_WALK_MASK_CLOBBERS);
  } else {
    asm volatile(
        "v_bfe_u32 %[vt], %[flo], 0, 2\n\t"
        "v_cmp_eq_u32_e64 s[36:37], %[vt], 3\n\t"
        "v_bfe_u32 %[vt], %[flo], 2, 2\n\t"
        "v_cmp_eq_u32_e64 s[38:39], %[vt], 3\n\t"
        "v_bfe_u32 %[vt], %[flo], 4, 2\n\t"
        "v_cmp_eq_u32_e64 s[40:41], %[vt], 3\n\t"
        "v_bfe_u32 %[vt], %[flo], 6, 2\n\t"
        "v_cmp_eq_u32_e64 s[42:43], %[vt], 3\n\t"
        "v_bfe_u32 %[vt], %[flo], 8, 2\n\t"
        "v_cmp_eq_u32_e64 s[44:45], %[vt], 3\n\t"
        "v_bfe_u32 %[vt], %[flo], 10, 2\n\t"
        "v_cmp_eq_u32_e64 s[46:47], %[vt], 3\n\t"
        "v_bfe_u32 %[vt], %[flo], 12, 2\n\t"
        "v_cmp_eq_u32_e64 s[48:49], %[vt], 3\n\t"
        "v_bfe_u32 %[vt], %[flo], 14, 2\n\t"
        "v_cmp_eq_u32_e64 s[50:51], %[vt], 3\n\t"
        "v_bfe_u32 %[vt], %[flo], 16, 2\n\t"
        "v_cmp_eq_u32_e64 s[52:53], %[vt], 3\n\t"
        "v_bfe_u32 %[vt], %[flo], 18, 2\n\t"
        "v_cmp_eq_u32_e64 s[54:55], %[vt], 3\n\t"
        "v_bfe_u32 %[vt], %[flo], 20, 2\n\t"
        "v_cmp_eq_u32_e64 s[56:57], %[vt], 3\n\t"
        "v_bfe_u32 %[vt], %[flo], 22, 2\n\t"
        "v_cmp_eq_u32_e64 s[58:59], %[vt], 3\n\t"
        "v_bfe_u32 %[vt], %[flo], 24, 2\n\t"
        "v_cmp_eq_u32_e64 s[60:61], %[vt], 3\n\t"
        "v_bfe_u32 %[vt], %[flo], 26, 2\n\t"
        "v_cmp_eq_u32_e64 s[62:63], %[vt], 3\n\t"
        "v_bfe_u32 %[vt], %[flo], 28, 2\n\t"
        "v_cmp_eq_u32_e64 s[64:65], %[vt], 3\n\t"
        "v_bfe_u32 %[vt], %[flo], 30, 2\n\t"
        "v_cmp_eq_u32_e64 s[66:67], %[vt], 3\n\t"
        "v_bfe_u32 %[vt], %[fhi], 0, 2\n\t"
        "v_cmp_eq_u32_e64 s[68:69], %[vt], 3\n\t"
        "v_bfe_u32 %[vt], %[fhi], 2, 2\n\t"
        "v_cmp_eq_u32_e64 s[70:71], %[vt], 3\n\t"
        "v_bfe_u32 %[vt], %[fhi], 4, 2\n\t"
        "v_cmp_eq_u32_e64 s[72:73], %[vt], 3\n\t"
        "v_bfe_u32 %[vt], %[fhi], 6, 2\n\t"
        "v_cmp_eq_u32_e64 s[74:75], %[vt], 3\n\t"
        "v_bfe_u32 %[vt], %[fhi], 8, 2\n\t"
        "v_cmp_eq_u32_e64 s[76:77], %[vt], 3\n\t"
        "v_bfe_u32 %[vt], %[fhi], 10, 2\n\t"
        "v_cmp_eq_u32_e64 s[78:79], %[vt], 3\n\t"
        "v_bfe_u32 %[vt], %[fhi], 12, 2\n\t"
        "v_cmp_eq_u32_e64 s[80:81], %[vt], 3\n\t"
        "v_bfe_u32 %[vt], %[fhi], 14, 2\n\t"
        "v_cmp_eq_u32_e64 s[82:83], %[vt], 3\n\t"
        "v_bfe_u32 %[vt], %[fhi], 16, 2\n\t"
        "v_cmp_eq_u32_e64 s[84:85], %[vt], 3\n\t"
        "v_bfe_u32 %[vt], %[fhi], 18, 2\n\t"
        "v_cmp_eq_u32_e64 s[86:87], %[vt], 3\n\t"
        "v_bfe_u32 %[vt], %[fhi], 20, 2\n\t"
        "v_cmp_eq_u32_e64 s[88:89], %[vt], 3\n\t"
        "v_bfe_u32 %[vt], %[fhi], 22, 2\n\t"
        "v_cmp_eq_u32_e64 s[90:91], %[vt], 3\n\t"
        "v_bfe_u32 %[vt], %[fhi], 24, 2\n\t"
        "v_cmp_eq_u32_e64 s[92:93], %[vt], 3\n\t"
        "v_bfe_u32 %[vt], %[fhi], 26, 2\n\t"
        "v_cmp_eq_u32_e64 s[94:95], %[vt], 3\n\t"
        "v_bfe_u32 %[vt], %[fhi], 28, 2\n\t"
        "v_cmp_eq_u32_e64 s[96:97], %[vt], 3\n\t"
        "v_bfe_u32 %[vt], %[fhi], 30, 2\n\t"
        "v_cmp_eq_u32_e64 s[98:99], %[vt], 3\n\t"
        "s_nop 3\n\t"
        "s_add_i32 %[o], %[o], 16\n\t"      // the loop keeps o + 16 (the mask's index)
        "Ltop_%=:\n\t"
        "s_lshl_b32 m0, %[o], 1\n\t"
        "s_sub_i32 %[t1], 64, %[d]\n\t"     // (also the wait state between the write of M0 and s_movrels)
        "s_movrels_b64 vcc, s[36:37]\n\t"
        "s_lshr_b64 vcc, vcc, %[d]\n\t"
        "s_ff0_i32_b64 %[L], vcc\n\t"       // -1 (no zero: d = 0 and 64 diagonal moves) is the largest unsigned value
        "s_min_u32 %[L], %[L], %[t1]\n\t"   // d = 64 can arrive here (an up move out of the last row): the shift was by 0 then
        "s_add_i32 %[k], %[k], %[L]\n\t"
        "s_add_i32 %[d], %[d], %[L]\n\t"
        "s_cmp_ge_i32 %[k], 32\n\t"
        "s_cbranch_scc1 Lword_%=\n\t"
        "Lback_%=:\n\t"
        "s_cmp_ge_i32 %[d], 64\n\t"
        "s_cbranch_scc1 Lout_%=\n\t"
        "v_readlane_b32 vcc_hi, %[fhi], %[d]\n\t"
        "v_readlane_b32 vcc_lo, %[flo], %[d]\n\t"
        "s_lshr_b64 vcc, vcc, m0\n\t"         // M0 is still 2 (o + 16): the row's code at this drift
        "s_bfe_u32 %[t1], vcc_lo, 0x10001\n\t"   // bit 1 of the code (1 left, 2 up): 1 for an up move
        "s_lshl1_add_u32 %[t0], %[k], %[t1]\n\t"
        "s_bitset1_b64 %[acc], %[t0]\n\t"       // the op is one set bit of the move's two: 2k (left) or 2k + 1 (up)
        "s_add_i32 %[k], %[k], 1\n\t"
        "s_add_i32 %[d], %[d], %[t1]\n\t"
        "s_lshl1_add_u32 %[o], %[t1], %[o]\n\t"
        "s_add_i32 %[o], %[o], -1\n\t"
        "s_cmp_lt_u32 %[o], 32\n\t"
        "s_cbranch_scc1 Ltop_%=\n\t"
        "s_branch Lout_%=\n\t"
        "Lword_%=:\n\t"                       // a word of 32 moves is full: stage it, start the next
        "v_mov_b64 %[wtmp], %[acc]\n\t"
        "ds_write_b64 %[saddr], %[wtmp]\n\t"
        "v_add_u32 %[saddr], 8, %[saddr]\n\t"
        "s_mov_b64 %[acc], 0\n\t"
        "s_sub_i32 %[k], %[k], 32\n\t"
        "s_cmp_ge_i32 %[k], 32\n\t"
        "s_cbranch_scc1 Lword_%=\n\t"
        "s_branch Lback_%=\n\t"
        "Lout_%=:\n\t"
        "s_add_i32 %[o], %[o], -16\n\t"
        : [d] "+s"(dcur), [o] "+s"(o), [k] "+s"(k), [acc] "+s"(acc), [c] "+s"(c), [t0] "=&s"(t0), [t1] "=&s"(t1), [L] "=&s"(L),
          [saddr] "+v"(saddr), [wtmp] "=&v"(wtmp), [vt] "=&v"(vt)
        : [flo] "v"(Flo), [fhi] "v"(Fhi)
        : "scc", "vcc", "memory", SA_WALK_MASK_CLOBBERS);
  }
  return c;
}

// the same for the bit-vector record: masks from the symbols (yv: the lane's 32 target symbols as bytes, xv: the symbol of its row)
__device__ __forceinline__ void walk_window_bits(const uint32_t (&yv)[8], const uint32_t xv, const uint32_t ca_lo, const uint32_t ca_hi,
                                                 const uint32_t cb_lo, const uint32_t cb_hi, int& dcur, int& o, int& k,
                                                 unsigned long long& acc, uint32_t& saddr) {
  uint32_t t0, t1, L;
  unsigned long long wtmp;
  asm volatile(
      "v_cmp_eq_u32_sdwa s[36:37], %[y0], %[xv] src0_sel:BYTE_0 src1_sel:DWORD\n\t"
      "v_cmp_eq_u32_sdwa s[38:39], %[y0], %[xv] src0_sel:BYTE_1 src1_sel:DWORD\n\t"
      "v_cmp_eq_u32_sdwa s[40:41], %[y0], %[xv] src0_sel:BYTE_2 src1_sel:DWORD\n\t"
      "v_cmp_eq_u32_sdwa s[42:43], %[y0], %[xv] src0_sel:BYTE_3 src1_sel:DWORD\n\t"
      "v_cmp_eq_u32_sdwa s[44:45], %[y1], %[xv] src0_sel:BYTE_0 src1_sel:DWORD\n\t"
      "v_cmp_eq_u32_sdwa s[46:47], %[y1], %[xv] src0_sel:BYTE_1 src1_sel:DWORD\n\t"
      "v_cmp_eq_u32_sdwa s[48:49], %[y1], %[xv] src0_sel:BYTE_2 src1_sel:DWORD\n\t"
      "v_cmp_eq_u32_sdwa s[50:51], %[y1], %[xv] src0_sel:BYTE_3 src1_sel:DWORD\n\t"
      "v_cmp_eq_u32_sdwa s[52:53], %[y2], %[xv] src0_sel:BYTE_0 src1_sel:DWORD\n\t"
      "v_cmp_eq_u32_sdwa s[54:55], %[y2], %[xv] src0_sel:BYTE_1 src1_sel:DWORD\n\t"
      "v_cmp_eq_u32_sdwa s[56:57], %[y2], %[xv] src0_sel:BYTE_2 src1_sel:DWORD\n\t"
      "v_cmp_eq_u32_sdwa s[58:59], %[y2], %[xv] src0_sel:BYTE_3 src1_sel:DWORD\n\t"
      "v_cmp_eq_u32_sdwa s[60:61], %[y3], %[xv] src0_sel:BYTE_0 src1_sel:DWORD\n\t"
      "v_cmp_eq_u32_sdwa s[62:63], %[y3], %[xv] src0_sel:BYTE_1 src1_sel:DWORD\n\t"
      "v_cmp_eq_u32_sdwa s[64:65], %[y3], %[xv] src0_sel:BYTE_2 src1_sel:DWORD\n\t"
      "v_cmp_eq_u32_sdwa s[66:67], %[y3], %[xv] src0_sel:BYTE_3 src1_sel:DWORD\n\t"
      "v_cmp_eq_u32_sdwa s[68:69], %[y4], %[xv] src0_sel:BYTE_0 src1_sel:DWORD\n\t"
      "v_cmp_eq_u32_sdwa s[70:71], %[y4], %[xv] src0_sel:BYTE_1 src1_sel:DWORD\n\t"
      "v_cmp_eq_u32_sdwa s[72:73], %[y4], %[xv] src0_sel:BYTE_2 src1_sel:DWORD\n\t"
      "v_cmp_eq_u32_sdwa s[74:75], %[y4], %[xv] src0_sel:BYTE_3 src1_sel:DWORD\n\t"
      "v_cmp_eq_u32_sdwa s[76:77], %[y5], %[xv] src0_sel:BYTE_0 src1_sel:DWORD\n\t"
      "v_cmp_eq_u32_sdwa s[78:79], %[y5], %[xv] src0_sel:BYTE_1 src1_sel:DWORD\n\t"
      "v_cmp_eq_u32_sdwa s[80:81], %[y5], %[xv] src0_sel:BYTE_2 src1_sel:DWORD\n\t"
      "v_cmp_eq_u32_sdwa s[82:83], %[y5], %[xv] src0_sel:BYTE_3 src1_sel:DWORD\n\t"
      "v_cmp_eq_u32_sdwa s[84:85], %[y6], %[xv] src0_sel:BYTE_0 src1_sel:DWORD\n\t"
      "v_cmp_eq_u32_sdwa s[86:87], %[y6], %[xv] src0_sel:BYTE_1 src1_sel:DWORD\n\t"
      "v_cmp_eq_u32_sdwa s[88:89], %[y6], %[xv] src0_sel:BYTE_2 src1_sel:DWORD\n\t"
      "v_cmp_eq_u32_sdwa s[90:91], %[y6], %[xv] src0_sel:BYTE_3 src1_sel:DWORD\n\t"
      "v_cmp_eq_u32_sdwa s[92:93], %[y7], %[xv] src0_sel:BYTE_0 src1_sel:DWORD\n\t"
      "v_cmp_eq_u32_sdwa s[94:95], %[y7], %[xv] src0_sel:BYTE_1 src1_sel:DWORD\n\t"
      "v_cmp_eq_u32_sdwa s[96:97], %[y7], %[xv] src0_sel:BYTE_2 src1_sel:DWORD\n\t"
      "v_cmp_eq_u32_sdwa s[98:99], %[y7], %[xv] src0_sel:BYTE_3 src1_sel:DWORD\n\t"
      "s_nop 3\n\t"
      "s_add_i32 %[o], %[o], 16\n\t"      // the loop keeps o + 16 (the mask's index)
      "Ltop_%=:\n\t"
      "s_lshl_b32 m0, %[o], 1\n\t"
      "s_sub_i32 %[t1], 64, %[d]\n\t"     // (also the wait state between the write of M0 and s_movrels)
      "s_movrels_b64 vcc, s[36:37]\n\t"
      "s_lshr_b64 vcc, vcc, %[d]\n\t"
      "s_ff0_i32_b64 %[L], vcc\n\t"       // -1 (no zero: d = 0 and 64 diagonal moves) is the largest unsigned value
      "s_min_u32 %[L], %[L], %[t1]\n\t"   // d = 64 can arrive here (an up move out of the last row): the shift was by 0 then
      "s_add_i32 %[k], %[k], %[L]\n\t"
      "s_add_i32 %[d], %[d], %[L]\n\t"
      "s_cmp_ge_i32 %[k], 32\n\t"
      "s_cbranch_scc1 Lword_%=\n\t"
      "Lback_%=:\n\t"
      "s_cmp_ge_i32 %[d], 64\n\t"
      "s_cbranch_scc1 Lout_%=\n\t"
      "s_sub_i32 %[t0], 63, %[d]\n\t"       // column j0 - d + o: lane 79 - d + o of ca, or of cb when that is 64 or more
      "s_add_i32 %[t0], %[t0], %[o]\n\t"
      "v_readlane_b32 vcc_hi, %[cahi], %[t0]\n\t"
      "v_readlane_b32 vcc_lo, %[calo], %[t0]\n\t"
      "v_readlane_b32 %[t1], %[cblo], %[t0]\n\t"
      "v_readlane_b32 %[L], %[cbhi], %[t0]\n\t"
      "s_cmp_lt_u32 %[t0], 64\n\t"
      "s_cselect_b32 vcc_lo, vcc_lo, %[t1]\n\t"
      "s_cselect_b32 vcc_hi, vcc_hi, %[L]\n\t"
      "s_lshr_b64 vcc, vcc, %[d]\n\t"
      "s_and_b32 %[t1], vcc_lo, 1\n\t"          // set: up (op 2), clear: left (op 1)
      "s_lshl1_add_u32 %[t0], %[k], %[t1]\n\t"
      "s_bitset1_b64 %[acc], %[t0]\n\t"       // the op is one set bit of the move's two: 2k (left) or 2k + 1 (up)
      "s_add_i32 %[k], %[k], 1\n\t"
      "s_add_i32 %[d], %[d], %[t1]\n\t"
      "s_lshl1_add_u32 %[o], %[t1], %[o]\n\t"
      "s_add_i32 %[o], %[o], -1\n\t"
      "s_cmp_lt_u32 %[o], 32\n\t"
      "s_cbranch_scc1 Ltop_%=\n\t"
      "s_branch Lout_%=\n\t"
      "Lword_%=:\n\t"                       // a word of 32 moves is full: stage it, start the next
      "v_mov_b64 %[wtmp], %[acc]\n\t"
      "ds_write_b64 %[saddr], %[wtmp]\n\t"
      "v_add_u32 %[saddr], 8, %[saddr]\n\t"
      "s_mov_b64 %[acc], 0\n\t"
      "s_sub_i32 %[k], %[k], 32\n\t"
      "s_cmp_ge_i32 %[k], 32\n\t"
      "s_cbranch_scc1 Lword_%=\n\t"
      "s_branch Lback_%=\n\t"
      "Lout_%=:\n\t"
      "s_add_i32 %[o], %[o], -16\n\t"
      : [d] "+s"(dcur), [o] "+s"(o), [k] "+s"(k), [acc] "+s"(acc), [t0] "=&s"(t0), [t1] "=&s"(t1), [L] "=&s"(L), [saddr] "+v"(saddr),
        [wtmp] "=&v"(wtmp)
      : [y0] "v"(yv[0]), [y1] "v"(yv[1]), [y2] "v"(yv[2]), [y3] "v"(yv[3]), [y4] "v"(yv[4]), [y5] "v"(yv[5]), [y6] "v"(yv[6]), [y7] "v"(yv[7]),
        [xv] "v"(xv), [calo] "v"(ca_lo), [cahi] "v"(ca_hi), [cblo] "v"(cb_lo), [cbhi] "v"(cb_hi)
      : "scc", "vcc", "memory", SA_WALK_MASK_CLOBBERS);
}

// BITS: the bit-vector LCS format above instead of 2-bit codes (bands: the record starts with row i_min's strip).
// CODES: what the 2-bit codes are (walk_window): the LCS cell kernels' pointers, NW tags, SW tags; a template parameter,
// because the window loops as branches of one function leave the register allocator with SGPR values it cannot place.
template <bool BITS, int CODES = 0>
__global__ void __launch_bounds__(64) traceback_walk(const TraceArgs a) {
  static_assert(!(BITS && CODES != 0), "the bit-vector record is LCS only");
  // The packed moves are staged in LDS and written out in bulk: with no global stores inside the walk, the
  // only vector-memory operations in flight are the loads of the NEXT window, so waiting for them
  // at the top of the next round (vmcnt counts loads and stores together, in order) waits for
  // nothing else -- the prefetch really overlaps the walk through the current window.
  __shared__ unsigned long long stage[kTraceStageWords];
  __shared__ __attribute__((aligned(16))) uint8_t ywin[BITS ? 128 : 16];  // BITS: the target symbols of the window's 96 columns
  const int d = threadIdx.x;
  constexpr bool tagged = CODES != 0;
  int lgK = 0;
  while ((1 << lgK) < a.K) ++lgK;
  int i = a.m, j = a.n;
  if (a.start_i >= 0) {
    i = a.start_i;
    j = a.start_j;
  } else if (a.kind == KIND_SW) {
    if (*a.sw_best <= 0) {
      i = j = 0;
    } else {
      const unsigned long long p = *a.sw_pos;
      i = (int)(p >> 32);
      j = (int)(p & 0xffffffffu);
    }
  }
  i = __builtin_amdgcn_readfirstlane(i);
  j = __builtin_amdgcn_readfirstlane(j);
  if (d == 0) {
    a.hdr[1] = i;
    a.hdr[2] = j;
  }
  MoveEmitter em;
  em.stage = stage;
  em.out = a.ops2;
  em.lane = d;
  em.staged = 0;
  em.words_out = a.nops0 >> 5;  // the walks over the bands below left nops0 moves, the last word partial
  em.k = (int)(a.nops0 & 31);
  em.acc = 0;
  if (em.k) {
    const unsigned long long w = a.ops2[em.words_out];
    em.acc = (((unsigned long long)(uint32_t)__builtin_amdgcn_readfirstlane((int)(w >> 32)) << 32) |
              (unsigned long long)(uint32_t)__builtin_amdgcn_readfirstlane((int)w)) & ((1ull << (2 * em.k)) - 1ull);
  }
  const int i_min = a.i_min;
  // raw codes: NW / SW store the winner's tag (M 2, E 1, F 0, SW zero cell 3), LCS diagonal 3, up 2, left 1
  bool have = false;  // nw holds the window of entry cell (pi, pj)
  int pi = 0, pj = 0;
  using Window = std::conditional_t<BITS, BitsWindow, TraceWindow>;
  Window nw{};
  auto load_window = [&](int wi, int wj) {
    if constexpr (BITS)
      return load_bits_window(a, wi, wj, d);
    else
      return load_trace_window(a.trace, wi - i_min, wj, d, lgK, a.W);
  };
  bool stopped = false;
  while (i > i_min && j > 0 && !stopped) {
    if (em.staged > kTraceStageWords - 16) em.flush();  // a window adds at most 4 words
    if (window_ok(i - i_min, j)) {
      // ---- fast path: 64 rows per memory round trip, the next window already in flight ----
      if (!(have && pi == i && abs(j - pj) <= 14)) {
        // no usable prediction: fetch the window of this very cell and come round again
        pi = i;
        pj = j;
        have = true;
        nw = load_window(pi, pj);
        continue;
      }
      const Window w = nw;
      const int j0 = pj;
      int o = j - pj;
      // prefetch along the diagonal; a path that drifts more than 14 columns in 64 rows misses it
      have = window_ok(i - 64 - i_min, j - 64);
      if (have) {
        pi = i - 64;
        pj = j - 64;
        nw = load_window(pi, pj);
      }
      const int o_in = o;
      uint32_t Flo = 0, Fhi = 0, ca_lo = 0, ca_hi = 0, cb_lo = 0, cb_hi = 0;
      uint32_t yvs[8] = {0, 0, 0, 0, 0, 0, 0, 0};  // BITS: the lane's 32 target symbols
      if constexpr (BITS) {
        // rows i-64 .. i-1 (0-based) start at bit sh of the three dwords; reversed, so that bit d is row i - d
        const int sh = (i - 64) & 31;
        ca_lo = __builtin_bitreverse32(__builtin_amdgcn_alignbit(w.a2, w.a1, sh));
        ca_hi = __builtin_bitreverse32(__builtin_amdgcn_alignbit(w.a1, w.a0, sh));
        cb_lo = __builtin_bitreverse32(__builtin_amdgcn_alignbit(w.b2, w.b1, sh));
        cb_hi = __builtin_bitreverse32(__builtin_amdgcn_alignbit(w.b1, w.b0, sh));
        // diagmask from the symbols: row i - d meets, at drift o, column j0 - d + o = entry 79 - d + o of the window
        wave_lds_fence();  // the previous window's reads are done
        ywin[d] = (uint8_t)w.ya;
        ywin[64 + d] = (uint8_t)w.yb;
        wave_lds_fence();
        // the lane's 32 symbols, entries 63 - d .. 94 - d: nine aligned dwords, realigned by bytes; the compares then
        // pick their byte themselves (one LDS round trip per window instead of 32 byte reads with a wait each)
        const uint32_t* yw32 = reinterpret_cast<const uint32_t*>(ywin);
        const int yw0 = (63 - d) >> 2;
        const uint32_t ysh = (uint32_t)(63 - d) & 3u;
        uint32_t yr[9], yv[8];
#pragma unroll
        for (int k = 0; k < 9; ++k) yr[k] = yw32[yw0 + k];
#pragma unroll
        for (int k = 0; k < 8; ++k) yv[k] = __builtin_amdgcn_alignbyte(yr[k + 1], yr[k], ysh);
        for (int k = 0; k < 8; ++k) yvs[k] = yv[k];
      } else {
      // The lane's 32 codes around its own cell of the entry diagonal, as one 64-bit value: the code
      // at drift o (-16..15) is bits [2(o+16), 2(o+16)+1] (two funnel shifts per window).
      const int sh0 = 2 * (w.pb - 16);
      Flo = __builtin_amdgcn_alignbit(w.w1, w.w0, sh0), Fhi = __builtin_amdgcn_alignbit(w.w2, w.w1, sh0);
      // Walk the window on the scalar unit.  diagmask[q] (an SGPR pair, q = drift + 16) says for which of the 64 rows the
      // cell at that drift is a diagonal move: one compare per drift, 32 per window, all lanes at once (walk_window).
      // Then every round of the loop is "the diagonal run from here (a shift and a find-first-zero on that mask,
      // possibly empty), then ONE gap move (the row's codes sit in two scalar registers: a shift, an OR into the
      // emitter)" -- no vector work and no memory access per move, one round per gap move.
      }
#ifdef SA_DEBUG_BITS
      if constexpr (BITS) {
        printf("W i=%d j0=%d o=%d lane=%d xv=%u ya=%u yb=%u a=%08x %08x %08x b=%08x %08x %08x ca=%08x%08x cb=%08x%08x\n", i, j0, o, d, w.xv, w.ya,
               w.yb, w.a0, w.a1, w.a2, w.b0, w.b1, w.b2, ca_hi, ca_lo, cb_hi, cb_lo);
      }
#endif
      int dcur = 0;
      unsigned long long acc = em.acc;  // the emitter's state in scalars of our own for the duration of the window
      int k = em.k, staged = em.staged;
      // LDS byte address of the next staged word (all lanes hold the same value and store the same word)
      const uint32_t stage0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) unsigned long long*)stage;
      uint32_t saddr = stage0 + 8u * (uint32_t)staged;
      while (true) {
        // One round = the diagonal run from (dcur, o), then one gap move; rounds follow each other inside the
        // assembly until something rare happens (see walk_rounds).  Written by hand because the compiled loop
        // took four taken branches per round, and a lone wave pays ~50 cycles of instruction fetch for each.
        uint32_t c = 0;
        if constexpr (BITS)
          walk_window_bits(yvs, w.xv, ca_lo, ca_hi, cb_lo, cb_hi, dcur, o, k, acc, saddr);
        else
          c = walk_window<CODES>(Flo, Fhi, dcur, o, k, acc, saddr);
        while (k >= 32) {  // the loop was left right after the move that filled a word
          stage[((uint32_t)__builtin_amdgcn_readfirstlane((int)saddr) - stage0) >> 3] = acc;  // all lanes, same word, same place
          saddr += 8;
          acc = 0;
          k -= 32;
        }
        if (dcur >= 64) break;
        if (c == 3u) {  // SW stop cell
          stopped = true;
          break;
        }
        if (o < -16 || o > 15) break;  // drifted out of the window: re-centre
      }
      staged = (int)((uint32_t)__builtin_amdgcn_readfirstlane((int)saddr) - stage0) >> 3;
      em.acc = acc;
      em.k = k;
      em.staged = staged;
      if (dcur == 0 && o == o_in && !stopped) have = false;  // out of the window at once: take the generic path below
      i -= dcur;
      j = j0 - dcur + o;
      if (dcur != 0 || o != o_in || stopped) continue;
    }
    // ---- generic path (borders, SW tails, tiny matrices): one run per memory round trip ----
    have = false;
    const bool vM = (i - d > i_min) && (j - d >= 1);
    const bool vE = (j - d >= 1);
    const bool vF = (i - d > i_min);
    uint32_t cM = 3u, cE = 3u, cF = 3u;
    if constexpr (BITS) {
      if (vM) cM = bits_code(a, i - d, j - d);
      if (vE) cE = bits_code(a, i, j - d);
      if (vF) cF = bits_code(a, i - d, j);
    } else {
      if (vM) cM = trace_code(a.trace, i - d - i_min, j - d, lgK, a.W, tagged);
      if (vE) cE = trace_code(a.trace, i - i_min, j - d, lgK, a.W, tagged);
      if (vF) cF = trace_code(a.trace, i - d - i_min, j, lgK, a.W, tagged);
    }
    const uint32_t c0 = (uint32_t)__builtin_amdgcn_readfirstlane((int)cM);
    if (c0 == 3u) break;  // SW stop cell ('X', main.cpp:1233)
    unsigned long long run;
    if (c0 == 0u)
      run = __ballot(cM == 0u);
    else if (c0 == 1u)
      run = __ballot(cE == 1u);
    else
      run = __ballot(cF == 2u);
    const int L = (~run == 0ull) ? 64 : __builtin_ctzll(~run);
    em.emit_run(c0, L);
    if (c0 != 1u) i -= L;
    if (c0 != 2u) j -= L;
    // a diagonal run that ends inside the window also tells us the next move (the code of the
    // cell that ended it); take that one move now instead of paying a round trip for it
    if (c0 == 0u && L < 64) {
      const uint32_t cn = (uint32_t)__builtin_amdgcn_readlane((int)cM, L);
      if (cn == 1u || cn == 2u) {
        em.emit(cn);
        if (cn == 1u) --j; else --i;
      }
    }
  }
  if (i > 0 && j > 0) {
    // NW / LCS: the walk reached the top row of a recomputed band; the host continues with the band
    // above from (i, j).  SW: it stopped at its 'X' cell.
  } else if (a.kind != KIND_SW && a.stop_at_col0) {
    if (i == 0) {  // row 0 of the matrix: left moves to the stripe's edge
      em.emit_run(1u, j);
      j = 0;
    }
  } else if (a.kind != KIND_SW) {
    // border walk: row 0 is all 'E'/'e' (main.cpp:727-728), column 0 all 'F'/'f' (main.cpp:609-616);
    // LCS drains the same way (main.cpp:1623-1634)
    em.emit_run((i == 0) ? 1u : 2u, (i == 0) ? j : i);
    i = j = 0;
  }
  const long long nops = em.total();
  em.finish();
  if (d == 0) {
    a.hdr[0] = nops;
    a.hdr[3] = i;
    a.hdr[4] = j;
  }
}

// packed moves -> one byte per move (the C-ABI's op list); hdr[0] = number of moves
static __global__ void expand_ops(const unsigned long long* __restrict__ ops2, uint8_t* __restrict__ ops, const long long* __restrict__ hdr) {
  const long long n = hdr[0];
  for (long long q = (long long)blockIdx.x * blockDim.x + threadIdx.x; q < n; q += (long long)gridDim.x * blockDim.x)
    ops[q] = (uint8_t)((ops2[q >> 5] >> (2 * (q & 31))) & 3ull);
}

}  // namespace sa
