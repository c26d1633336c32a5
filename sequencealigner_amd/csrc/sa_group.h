// sa_group.h -- fill_group<KIND, K, FLAGS>: the persistent strip pipeline with the four waves of a workgroup on four
// CONSECUTIVE strips, in lock step, handing the strip bottom rows down through LDS (NW / SW; round 3).
//
// fill_persistent (sa_kernels.h) hands every bottom row over as agent-scope records: 80 steps of lag + about 4 us of
// store -> poll latency per strip, which is what a 100 kb x 100 kb fill (782 strips of 128 rows) mostly consists of.
// Here a workgroup draws a GROUP ticket = four consecutive strips, wave w takes strip 4 g + w, and
//   * the step is fill_persistent's (lane_step: in- and out-buffers in registers, rotated by DPP).  A first version
//     moved the buffers into LDS (lane 63 wrote its cell at every step, lane 0 read the cell above at every step:
//     7 instructions of per-step overhead instead of 11) and was SLOWER: an LDS instruction costs a lone wave 13 - 21
//     cycles of issue time against 5 for a VALU instruction (tools/probe_lds_issue.hip, profiles/r03_probe_lds_issue.txt),
//     so three of them per step cost more than the seven DPP moves they replaced.
//   * shortly before the end of a 16-step sub-block lanes 48..63 write 16 records {H, F} into an LDS ring (one
//     ds_write_b64 per 16 steps) and at its end the wave publishes a progress word (sub-blocks done); the wave below
//     reads that word and the 16 records of its next sub-block (two ds_reads, issued a step early so that their latency
//     passes behind the arithmetic) and goes on if the word says they are there, else re-reads (bounded).  Column c of a
//     strip's bottom row is produced at step c + 63 of that strip, so a sub-block of the strip below (columns 16 i ..
//     16 i + 15) needs the strip above to have finished its sub-block i + 4: consecutive strips of a group run 80 steps
//     + one LDS round trip apart, with no store -> poll latency on top.  A ring is 64 records (one 64-step block), record
//     p = column + 64 in entry p mod 64, so all offsets are immediates; a wave also reads the progress word of the wave
//     BELOW and does not overwrite records that wave has not taken yet (it never waits there unless the wave below stalls).
//     A second version synchronised the four waves with one s_barrier per sub-block instead: the barrier costs 100 cycles,
//     ties every wave to the slowest of the four in every tick and, as an asm with a memory clobber, cut the compiler's
//     profile prefetch at every sub-block: + 10 ns per step at any K (profiles/r03_probe_barrier.txt,
//     profiles/r03_group_v2_barrier_step_ns.txt).
//   * only the group's top wave polls hand-off records (those of the group above, as in fill_persistent); every wave
//     still writes its strip's records (the group below, checkpointed runs and banded recomputation read them), so the
//     ring-of-records invariants of sa_kernels.h hold unchanged: groups are drawn in order, a workgroup holds four
//     consecutive strips, unfinished strips span at most `waves` strips.
// Barriers are used between groups only (three per group).  A wave whose bounded wait expires raises *err, stops
// waiting for good and goes on computing garbage until its strip ends; the host reports SA_ERR_HIP.
#pragma once
#include "sa_kernels.h"

namespace sa {

constexpr int kGrpLag = 5;                 // sub-blocks by which a wave trails the wave above
constexpr int kGrpRingBytes = 64 * 8;      // one ring: 64 records {H, F}
constexpr int kGrpDummyOff = 5 * kGrpRingBytes;
constexpr int kGrpDummyBytes = 64 * 8 + 512;  // where the lanes that have nothing to say write (+ the largest immediate offset, 3 x 128)
constexpr int kGrpProgOff = kGrpDummyOff + kGrpDummyBytes;  // progress words: [0] "the wave above the top wave", [w + 1] wave w, [5] "below the bottom wave"
constexpr int kGrpWordOff = kGrpProgOff + 32;               // the group ticket
constexpr int kGrpProfOff = (kGrpWordOff + 32 + 255) & ~255;  // the four waves' query profiles follow
constexpr int kGrpInf = 0x3fffffff;

inline size_t group_lds_bytes(int alpha, int PW) { return (size_t)kGrpProfOff + (size_t)kWavesPerBlock * alpha * 64 * PW * 4; }

typedef int i32x2 __attribute__((ext_vector_type(2)));

// LDS accesses by absolute 32-bit LDS address (no base register, no add in front of the instruction)
template <class T>
__device__ __forceinline__ T lds_ld(int addr) {
  typedef __attribute__((address_space(3))) T LT;
  return *(const LT*)(unsigned)addr;
}
template <class T>
__device__ __forceinline__ T lds_ld_volatile(int addr) {  // issued where it stands, in order with the other volatile accesses and asm statements
  typedef __attribute__((address_space(3))) T LT;
  return *(const volatile LT*)(unsigned)addr;
}
template <class T>
__device__ __forceinline__ void lds_st(int addr, T v) {
  typedef __attribute__((address_space(3))) T LT;
  *(LT*)(unsigned)addr = v;
}

// all LDS accesses of this wave done, then the workgroup barrier; no wait for global loads / stores (a __syncthreads()
// would drain the trace stores and the record requests in flight)
__device__ __forceinline__ void group_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// handoff_acquire (sa_kernels.h) with an enable: `live` = 1 for a group's top wave until a bounded wait of its expires,
// else 0: nothing is waited for then, the request is still issued (a valid address, nobody looks at the result).
// Straight-line for every wave: a branch around it makes the pending request a phi, and the register copy behind the
// phi waits for the request just issued (vmcnt(0): 1 us per tick).
template <int KIND, int K>
__device__ __forceinline__ void handoff_acquire_g(const FillArgs& a, int2& cur, const int2* rd, const int Ts, const int lane, const int gen_r,
                                                  int& ibH, int& ibF, int& live) {
  const int need = (lane < 16 * live) & (Ts + lane < a.n);
  if (__builtin_expect(__ballot(need & ho_stale(cur.x, gen_r)) != 0ull, 0)) {
    int spins = 0;
    while (true) {
      __builtin_amdgcn_s_sleep(1);
      cur = ho_load(rd + Ts);
      if (__ballot(need & ho_stale(cur.x, gen_r)) == 0ull) break;
      if (++spins > a.spin_limit || ((spins & 255) == 0 && ld_coherent(a.err) != 0)) {  // give up: out of the way of the fast path
        live = 0;
        if (lane == 0) st_coherent(a.err, 1);
        break;
      }
    }
  }
  ibH = ho_value(cur.x);
  ibF = cur.y;
  cur = ho_load(rd + Ts + 16);  // in flight during the next sub-block
}

struct GroupWave {
  int rin;     // per lane: LDS address of record (lane & 15) of sub-block 0 in the ring this wave reads
  int wrec;    // per lane: lanes 48..63: entry (lane - 48) of the ring this wave writes; the other lanes: a place to write nothing
  int p_up, p_own, p_dn;  // LDS addresses of the progress words of the wave above, of this wave and of the wave below
};

// The hand-off writes go out as asm statements WITHOUT a memory clobber: as far as the compiler knows they touch no
// memory, so its profile reads (which they cannot alias) keep moving across them; nobody in this wave reads them back.
template <int OFF>
__device__ __forceinline__ void lds_write_records(int addr, int h, int f) {
  i32x2 o = {h, f};
  asm volatile("ds_write_b64 %0, %1 offset:%2" ::"v"(addr), "v"(o), "n"(OFF));
}
__device__ __forceinline__ void lds_write_word(int addr, int v) { asm volatile("ds_write_b32 %0, %1" ::"v"(addr), "v"(v)); }

// 64 steps (four sub-blocks, expanded in line) of one strip.  On entry ibH / ibF hold the records of the block's first
// sub-block, on return those of the next block's first.  j0: index of the block's first sub-block within the strip.
template <int KIND, int K, bool BND, int FLAGS>
__device__ __forceinline__ void run_block_g(Lane<KIND, K>& st, const FillArgs& a, const int s, const int Tb, const int lane, int& live,
                                            int& nowait, const char* ldsLane, const uint32_t (&xr)[K], int* dprow, const GroupWave& gw, int yb,
                                            int& ny, int2& pend, int& ibH, int& ibF, int& obH, int& obF, const int2* rd, int2* wr, int2* sink,
                                            const int gen_r, const int gen_w) {
  uint32_t* tdst = a.trace + (((size_t)(s - a.trace_s0) * a.W + (Tb >> 4)) * 64 + lane) * K;
  const int j0 = Tb >> 4;
  auto sub_block = [&](auto subc) {
    constexpr int sub = decltype(subc)::value;
    const int Ts = Tb + 16 * sub;
    const int jn = j0 + sub + 1;  // the sub-block after this one
    uint32_t tacc[K], sbk[K], skp[K];
#pragma unroll
    for (int k = 0; k < K; ++k) tacc[k] = 0, sbk[k] = 0, skp[k] = 0;
    const int cbase = Ts - lane;
    int p_up = 0, p_dn = 0;
    i32x2 lrec = {0, 0};
#pragma unroll
    for (int u = 0; u < 16; ++u) {
      if (u == 15) {
        // After 15 steps lane 48 holds the record of the LAST step of the sub-block before and lanes 49..63 those of
        // steps Ts .. Ts + 14: the records with p = Ts .. Ts + 15 (p = step + 1 = column + 64), exactly the 16 columns
        // one sub-block of the wave below takes in.  Into ring entries p mod 64, one step early: the write has landed
        // when the progress word goes out behind it.  And the look at the neighbours, one step early too: the progress
        // word of the wave above FIRST, then the records it vouches for.
        lds_write_records<128 * sub>(gw.wrec, obH, obF);
        p_up = lds_ld_volatile<int>(gw.p_up);
        p_dn = lds_ld_volatile<int>(gw.p_dn);
        lrec = lds_ld_volatile<i32x2>(gw.rin + ((sub + 1) & 3) * 128);
      }
      lane_step<KIND, K, BND, FLAGS>(st, ibH, ibF, yb, obH, obF, tacc, u, ldsLane, xr, a, cbase + u, dprow, sbk, skp);
    }
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
    lds_write_word(gw.p_own, jn);  // sub-blocks 0 .. jn - 1 of this strip are in the ring
    if (KIND == KIND_SW) {
#pragma unroll
      for (int k = 0; k < K; ++k) {
        const int v = (int)(sbk[k] >> 4) << 2;
        const bool better = v > st.best[k];
        st.best[k] = better ? v : st.best[k];
        st.bestc[k] = better ? cbase + 15 - (int)(sbk[k] & 15u) : st.bestc[k];
      }
    }
    if (!(FLAGS & FLAG_NOTRACE)) store_trace<K>(tdst, tacc);
    tdst += 64 * K;
    if (!BND)
      ho_store(lane >= 48 ? wr + Ts : sink, obH, obF, gen_w);
    else if (lane >= 48)
      ho_store(wr + Ts, obH, obF, gen_w);
    // the group's top wave: the records of the next sub-block from the group above (requested one sub-block ago)
    int hH = 0, hF = 0;
    const int was_live = live;  // a wave that gives up here still takes what it got from the records, not from the LDS ring
    handoff_acquire_g<KIND, K>(a, pend, rd, Ts + 16, lane, gen_r, hH, hF, live);
    if (sub == 2) ny = a.yenc[Tb + 64 + lane];
    // Sub-block jn takes the records of sub-block jn + 4 of the wave above (jn + 5 published), and its own records will
    // overwrite those of this wave's sub-block jn - 4, which sub-block jn - 8 of the wave below took in: that one must
    // be finished (jn - 7 published by the wave below).
    if (__builtin_expect(((__builtin_amdgcn_readfirstlane(p_up) < jn + kGrpLag) | (__builtin_amdgcn_readfirstlane(p_dn) < jn - 7)) & (nowait == 0), 0)) {
      int spins = 0;
      while (true) {
        __builtin_amdgcn_s_sleep(1);
        p_up = lds_ld_volatile<int>(gw.p_up);
        p_dn = lds_ld_volatile<int>(gw.p_dn);
        lrec = lds_ld_volatile<i32x2>(gw.rin + ((sub + 1) & 3) * 128);
        if (__builtin_amdgcn_readfirstlane(p_up) >= jn + kGrpLag && __builtin_amdgcn_readfirstlane(p_dn) >= jn - 7) break;
        if (++spins > a.spin_limit || ((spins & 255) == 0 && ld_coherent(a.err) != 0)) {
          nowait = 1;
          if (lane == 0) st_coherent(a.err, 1);
          break;
        }
      }
    }
    ibH = was_live ? hH : lrec.x;
    ibF = was_live ? hF : lrec.y;
  };
  sub_block(std::integral_constant<int, 0>());
  sub_block(std::integral_constant<int, 1>());
  sub_block(std::integral_constant<int, 2>());
  sub_block(std::integral_constant<int, 3>());
}

template <int KIND, int K, int FLAGS>
__global__ void __launch_bounds__(64 * kWavesPerBlock, 2) fill_group(const FillArgs a) {
  using G = Geo<KIND, K>;
  constexpr int PF = FLAGS | FLAG_PERSIST;
  extern __shared__ __align__(16) char glds[];
  const int lbase = (int)(unsigned)(size_t)glds;  // the low half of a generic LDS pointer is the LDS address
  const int lane = threadIdx.x & 63;
  const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int NB = (a.n + 63 + 63) / 64;
  const int prof_bytes = a.alpha * 64 * G::PW * 4;
  GroupWave gw;
  gw.rin = lbase + w * kGrpRingBytes + (lane & 15) * 8;
  gw.wrec = lane >= 48 ? lbase + (w + 1) * kGrpRingBytes + (lane - 48) * 8 : lbase + kGrpDummyOff + lane * 8;
  gw.p_up = lbase + kGrpProgOff + 4 * w;
  gw.p_own = gw.p_up + 4;
  gw.p_dn = gw.p_up + 8;
  uint32_t* lds = reinterpret_cast<uint32_t*>(glds + kGrpProfOff + w * prof_bytes);
  const char* ldsLane = reinterpret_cast<const char*>(lds) + lane * 4 * G::PW;
  const int gword = lbase + kGrpWordOff;
  int dprow[K];
  uint32_t xr[K];
#pragma unroll
  for (int k = 0; k < K; ++k) dprow[k] = -1, xr[k] = 0;

  while (true) {
    // group ticket: strips s_first + 4 g .. + 3; nobody starts a group once an error is up
    group_barrier();  // every wave is done with the group before: rings and progress words are free
    if (threadIdx.x == 0) {
      int g = atomicAdd(a.ticket, 1);
      if (ld_coherent(a.err) != 0) g = kGrpInf;
      lds_st<int>(gword, g);
      lds_st<int>(lbase + kGrpProgOff, kGrpInf);
      lds_st<int>(lbase + kGrpProgOff + 4 * (kWavesPerBlock + 1), kGrpInf);
    }
    group_barrier();
    const int g = __builtin_amdgcn_readfirstlane(lds_ld<int>(gword));
    if (g > (a.s_last - a.s_first) / kWavesPerBlock) break;
    const int s = a.s_first + kWavesPerBlock * g + w;
    lds_st<int>(gw.p_own, s > a.s_last ? kGrpInf : 0);  // a strip that does not exist never holds the wave above back
    group_barrier();
    if (s > a.s_last) continue;
    int live = w == 0 ? 1 : 0;  // 1: this wave takes hand-off records from the group above
    const int rowg = s * 64 + lane;
    const int r0 = rowg * K;
    {
      const uint32_t* gp = a.prof + (size_t)s * a.alpha * 64 * G::PW;
      for (int i = lane; i < a.alpha * 64 * G::PW; i += 64) lds[i] = gp[i];
      wave_lds_fence();
    }
    Lane<KIND, K> st;
    strip_begin<KIND, K, FLAGS>(st, a, r0, false);
#pragma unroll
    for (int k = 0; k < K; ++k) st.Hg[k] = add_gap<(FLAGS & FLAG_FRAC) != 0>(st.H[k], a.go, a.fo) + 1;
    const int gen_r = ho_gen((s / a.ring) % 3), gen_w = ho_gen(((s + 1) / a.ring) % 3);
    // Waves that are not the group's top issue the same requests (the code is straight-line) but look at nothing: they
    // read slot `ring`, which nobody writes.  (Reading the wave above's slot, they waited every tick for a line that
    // wave was writing at that moment: + 190 ns per tick.)
    const int2* rd = a.bot2 + (size_t)(live ? s % a.ring : a.ring) * a.NP + 64 + min(lane, 15);
    int2* const wr_slot_base = a.bot2 + (size_t)((s + 1) % a.ring) * a.NP;
    int2* wr = wr_slot_base + 1 + (lane - 48);
    int2 pend = ho_load(rd);
    int ny = a.yenc[lane];
    int ibH = 0, ibF = 0, obH = 0, obF = 0;
    const int top = live;
    handoff_acquire_g<KIND, K>(a, pend, rd, 0, lane, gen_r, ibH, ibF, live);
#ifdef SA_DEBUG_HANDOFF
    const long long dbg_t0 = wall_clock64();
#endif
    int nowait = 0;
    if (!top) {  // the first sub-block's records from the wave above (its sub-block 4): columns 0..15 = entries 0..15
      int spins = 0;
      i32x2 lrec;
      while (true) {
        const int p_up = lds_ld_volatile<int>(gw.p_up);
        lrec = lds_ld_volatile<i32x2>(gw.rin);
        if (__builtin_amdgcn_readfirstlane(p_up) >= kGrpLag) break;
        __builtin_amdgcn_s_sleep(2);
        if (++spins > a.spin_limit || ((spins & 255) == 0 && ld_coherent(a.err) != 0)) {
          nowait = 1;
          if (lane == 0) st_coherent(a.err, 1);
          break;
        }
      }
      ibH = lrec.x;
      ibF = lrec.y;
    }
#pragma unroll 1
    for (int b = 0; b < NB; ++b) {
      const int Tb = 64 * b;
      const int yb = ny * G::ROWB;
      const bool interior = (Tb >= 64) && (Tb + 63 < a.n);
      if (interior)
        run_block_g<KIND, K, false, PF>(st, a, s, Tb, lane, live, nowait, ldsLane, xr, dprow, gw, yb, ny, pend, ibH, ibF, obH, obF, rd, wr, wr_slot_base, gen_r, gen_w);
      else if (b == 0 && a.n > 63)
        run_block_g<KIND, K, true, PF | FLAG_START>(st, a, s, 0, lane, live, nowait, ldsLane, xr, dprow, gw, yb, ny, pend, ibH, ibF, obH, obF, rd, wr, wr_slot_base, gen_r, gen_w);
      else if (b > 0)
        run_block_g<KIND, K, true, PF | FLAG_TAIL>(st, a, s, Tb, lane, live, nowait, ldsLane, xr, dprow, gw, yb, ny, pend, ibH, ibF, obH, obF, rd, wr, wr_slot_base, gen_r, gen_w);
      else
        run_block_g<KIND, K, true, PF>(st, a, s, Tb, lane, live, nowait, ldsLane, xr, dprow, gw, yb, ny, pend, ibH, ibF, obH, obF, rd, wr, wr_slot_base, gen_r, gen_w);
    }
#ifdef SA_DEBUG_HANDOFF
    if (a.dbg && lane == 0) {
      a.dbg[4 * s + 0] = dbg_t0;
      a.dbg[4 * s + 1] = wall_clock64();
    }
#endif
    // The record of the strip's LAST step (lane 63 now; p = 64 NB, entry 0) would have gone out with the next sub-block's:
    // its column, 64 (NB - 1), can be a real one.  Then "done": the last sub-blocks of the wave below ask for sub-blocks
    // that do not exist.
    lds_write_records<0>(lane == 63 ? lbase + (w + 1) * kGrpRingBytes : lbase + kGrpDummyOff + lane * 8, obH, obF);
    lds_write_word(gw.p_own, kGrpInf);
    strip_end<KIND, K>(st, a, s, rowg, r0);  // garbage if a wait expired above: *err is up then and the host reports it
  }
}

}  // namespace sa
