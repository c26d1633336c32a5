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
//   * at the end of a 16-step sub-block lanes 48..63 write the sub-block's 16 records {H, F} into an LDS ring (one
//     ds_write_b64 per 16 steps), the waves meet in ONE s_barrier (a tick), and lanes 0..15 of the wave below read the 16
//     records of their next sub-block (one ds_read_b64).  Wave w runs its sub-block i in tick i + 5 w: column c of a
//     strip's bottom row is produced at step c + 63 of that strip, so a sub-block of the strip below (columns 16 i ..
//     16 i + 15) needs the strip above to have finished its sub-block i + 4 -- a lag of five ticks = 80 steps, with no
//     store -> poll latency on top.  A ring is 64 records (one 64-step block): record of column c sits in entry
//     (c + 64) mod 64, so the offsets are immediates.
//   * only the group's top wave polls hand-off records (those of the group above, as in fill_persistent); every wave
//     still writes its strip's records (the group below, checkpointed runs and banded recomputation read them), so the
//     ring-of-records invariants of sa_kernels.h hold unchanged: groups are drawn in order, a workgroup holds four
//     consecutive strips, unfinished strips span at most `waves` strips.
//   * a wave that waits sits in s_barrier and takes no issue slots, so two workgroups per CU (two waves per SIMD) work
//     here, where fill_persistent's polling waves starved the working ones.
// Every wave executes the same number of barriers per group (4 NB + 15) whatever happens: a top wave whose bounded wait
// expires raises *err, stops waiting and goes on computing garbage until the group ends; the host reports SA_ERR_HIP.
#pragma once
#include "sa_kernels.h"

namespace sa {

constexpr int kGrpLag = 5;                 // ticks by which a wave trails the wave above
constexpr int kGrpRingBytes = 64 * 16;     // one ring: 64 entries {H, F, profile-row address, -}
constexpr int kGrpDummyOff = 5 * kGrpRingBytes;
constexpr int kGrpDummyBytes = 64 * 8 + 64 * 16;  // where the lanes that have nothing to say write (largest immediate: 63 x 16 + 8)
constexpr int kGrpWordOff = kGrpDummyOff + kGrpDummyBytes;
constexpr int kGrpProfOff = (kGrpWordOff + 64 + 255) & ~255;  // the four waves' query profiles follow

inline size_t group_lds_bytes(int alpha, int PW) { return (size_t)kGrpProfOff + (size_t)kWavesPerBlock * alpha * 64 * PW * 4; }

typedef int i32x2 __attribute__((ext_vector_type(2)));
typedef int i32x3 __attribute__((ext_vector_type(3)));

// LDS accesses by absolute 32-bit LDS address (no base register, no add in front of the instruction)
template <class T>
__device__ __forceinline__ T lds_ld(int addr) {
  typedef __attribute__((address_space(3))) T LT;
  return *(const LT*)(unsigned)addr;
}
template <class T>
__device__ __forceinline__ void lds_st(int addr, T v) {
  typedef __attribute__((address_space(3))) T LT;
  *(LT*)(unsigned)addr = v;
}

// all LDS accesses of this wave done, then the workgroup barrier; no wait for global loads / stores (a __syncthreads()
// would drain the trace stores and the record requests in flight)
__device__ __forceinline__ void group_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

template <int K>
struct GrpAhead {
  static constexpr int RA = K >= 4 ? 2 : 3;  // steps ahead at which ring entries are requested
  static constexpr int YA = K >= 4 ? 1 : 2;  // steps ahead at which the profile words are requested
};

// handoff_acquire (sa_kernels.h) with an enable: `live` = 1 for a group's top wave until a bounded wait of its expires,
// else 0: nothing is waited for then, the request is still issued (a valid address, nobody looks at the result)
template <int KIND, int K>
__device__ __forceinline__ bool handoff_acquire_g(const FillArgs& a, int2& cur, const int2* rd, const int Ts, const int lane, const int gen_r,
                                                  int& ibH, int& ibF, const int live) {
  const int need = (lane < 16 * live) & (Ts + lane < a.n);
  if (__builtin_expect(__ballot(need & ho_stale(cur.x, gen_r)) != 0ull, 0)) {
    int spins = 0;
    while (true) {
      __builtin_amdgcn_s_sleep(1);
      cur = ho_load(rd + Ts);
      if (__ballot(need & ho_stale(cur.x, gen_r)) == 0ull) break;
      if (++spins > a.spin_limit || ((spins & 255) == 0 && ld_coherent(a.err) != 0)) return false;
    }
  }
  ibH = ho_value(cur.x);
  ibF = cur.y;
  cur = ho_load(rd + Ts + 16);  // in flight during the next sub-block
  return true;
}

struct GroupWave {
  int ring_in;   // LDS address of the ring lane 0 reads
  int wout;      // per lane: address of the ring lane 63 writes / of this lane's dummy slot
  int outrd;     // per lane: ring_out + (lane & 15) * 16, the record this lane stores to global memory
  int ystage;    // per lane: address of the profile-row field of entry (lane - 1) mod 64 of the in-ring
  int dummy;     // per lane: a place to write nothing
  int yrow0;     // LDS address of this wave's profile (the address lane 0 uses for symbol 0)
  int lstride;   // profile bytes per lane, in a VGPR (operand of v_add_u32_dpp)
};

// the profile-row address of one step: lane 0 keeps the ring's, lane l takes lane l-1's of the step before + the lane
// stride; then the profile words behind it
template <int PW>
__device__ __forceinline__ void y_stage(int& ypass, const int ringz, const int lstride, uint32_t (&pw)[PW]) {
  int yv = ringz;
  asm("v_add_u32_dpp %0, %1, %2 wave_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(yv) : "v"(ypass), "v"(lstride));
  ypass = yv;
  if (PW == 1) {
    pw[0] = (uint32_t)lds_ld<int>(yv);
  } else {
    const i32x2 t = lds_ld<i32x2>(yv);
    pw[0] = (uint32_t)t.x;
    pw[PW - 1] = (uint32_t)t.y;
  }
}

// K cells of column c of one lane (NW / SW); rec = {H, F} of the cell above lane 0 (lanes > 0 take lane l-1's bottom
// cell of the step before); lane 63's bottom cell goes to the out-ring.
template <int KIND, int K, bool BND, int FLAGS>
__device__ __forceinline__ void lane_cells_g(Lane<KIND, K>& st, uint32_t (&tacc)[K], const int u, const i32x3 rec,
                                             const uint32_t (&pw)[Geo<KIND, K>::PW], const int waddr, const FillArgs& a, const int c,
                                             uint32_t (&sbk)[K], uint32_t (&skp)[K]) {
  static_assert(KIND != KIND_LCS, "fill_group: NW and SW");
  constexpr bool FRAC = (FLAGS & FLAG_FRAC) != 0;
  const int go = a.go, ge = a.ge, n = a.n;
  int up = dpp_shr1(rec.x, st.passH);
  int uf = dpp_shr1(rec.y, st.passF);
  int diag = st.dprev;
  st.dprev = up;
  const bool act = BND ? (c >= 0 && c < n) : true;
  int f_last = 0;
  int upg = add_gap<FRAC>(up, go, a.fo);
#pragma unroll
  for (int k = 0; k < K; ++k) {
    const int left = st.H[k];
    const int sg = (int)(int8_t)(pw[k / 4] >> (8 * (k & 3)));
    int f = max(upg, add_gap<FRAC>(uf, ge, a.fe));
    int e = max(st.Hg[k], add_gap<FRAC>(st.E[k], ge, a.fe));
    int mv = diag + sg;
    if (KIND == KIND_SW) {
      f = max(f, 0);
      e = max(e, 1);
    }
    int ht = max(max(mv, e), f);
    if (KIND == KIND_SW) ht = max(ht, 3);
    tacc[k] = __builtin_amdgcn_alignbit((uint32_t)ht, tacc[k], 2);
    if (KIND == KIND_SW) asm volatile("" : "+v"(tacc[k]));
    int h = ht & ~3;
    if (BND) {
      e = act ? e : st.E[k];
      f = act ? f : st.passF;
      h = act ? h : left;
    }
    st.E[k] = e;
    uf = f;
    f_last = f;
    if (KIND == KIND_SW) {
      uint32_t key = ((uint32_t)h << 2) + (uint32_t)(15 - u);
      if (BND) key = act ? key : 0u;
      if ((u & 1) == 0) {
        skp[k] = key;
        asm volatile("" : "+v"(skp[k]));
      } else {
        sbk[k] = max(max(sbk[k], skp[k]), key);
        asm volatile("" : "+v"(sbk[k]));
      }
    }
    diag = left;
    st.H[k] = h;
    up = h;
    upg = add_gap<FRAC>(h, go, a.fo);
    st.Hg[k] = upg + 1;
  }
  st.passH = st.H[K - 1];
  st.passF = BND ? (act ? f_last : st.passF) : f_last;
  // lane 63's bottom-row cell is the record of this step; every other lane writes into the dummy area
  i32x2 o = {st.passH, st.passF};
  lds_st<i32x2>(waddr, o);
}

// 64 steps (four sub-blocks = four ticks, expanded in line) of one strip.  pwc: the profile words of the first YA
// steps, requested before the barrier that precedes the block (and on return those of the next block).
template <int KIND, int K, bool BND, int FLAGS>
__device__ __forceinline__ void run_block_g(Lane<KIND, K>& st, const FillArgs& a, const int s, const int Tb, const int lane, int& live,
                                            const GroupWave& gw, uint32_t (&pwc)[GrpAhead<K>::YA][Geo<KIND, K>::PW], int& ycur, int& ynext,
                                            int2& pend, const int2* rd, int2* wr, int2* sink, const int gen_r,
                                            const int gen_w) {
  using G = Geo<KIND, K>;
  constexpr int RA = GrpAhead<K>::RA, YA = GrpAhead<K>::YA, PW = G::PW;
  uint32_t* tdst = a.trace + (((size_t)(s - a.trace_s0) * a.W + (Tb >> 4)) * 64 + lane) * K;
  uint32_t tprev[K];  // codes and records of the sub-block before, stored behind the barrier
  i32x2 oprev = {0, 0};
#pragma unroll
  for (int k = 0; k < K; ++k) tprev[k] = 0;
  auto stores = [&](const int Ts, const uint32_t (&t)[K], const i32x2 orec) {
    if (!(FLAGS & FLAG_NOTRACE)) store_trace<K>(tdst, t);
    tdst += 64 * K;
    // lane 48 + u stores the record of step Ts + u (bottom-row column Ts + u - 63, index p = Ts + u + 1)
    if (!BND)
      ho_store(lane >= 48 ? wr + Ts : sink, orec.x, orec.y, gen_w);
    else if (lane >= 48)
      ho_store(wr + Ts, orec.x, orec.y, gen_w);
  };
  auto sub_block = [&](auto subc) {
    constexpr int sub = decltype(subc)::value;
    const int Ts = Tb + 16 * sub;
    i32x3 rq[16];
    uint32_t pwq[16 + YA][PW];
#pragma unroll
    for (int u = 0; u < RA; ++u) {  // the address fields of the first YA entries were used before the barrier
      if (u < YA) {
        const i32x2 t = lds_ld<i32x2>(gw.ring_in + ((16 * sub + u + 63) & 63) * 16);
        rq[u] = i32x3{t.x, t.y, 0};
      } else {
        rq[u] = lds_ld<i32x3>(gw.ring_in + ((16 * sub + u + 63) & 63) * 16);
      }
    }
#pragma unroll
    for (int u = 0; u < YA; ++u)
#pragma unroll
      for (int q = 0; q < PW; ++q) pwq[u][q] = pwc[u][q];
    if (sub > 0) stores(Ts - 16, tprev, oprev);
    uint32_t tacc[K], sbk[K], skp[K];
#pragma unroll
    for (int k = 0; k < K; ++k) tacc[k] = 0, sbk[k] = 0, skp[k] = 0;
    const int cbase = Ts - lane;
    int zn[YA];  // address fields of the next sub-block's first entries
#pragma unroll
    for (int u = 0; u < 16; ++u) {
      if (u + RA < 16) rq[u + RA] = lds_ld<i32x3>(gw.ring_in + ((16 * sub + u + RA + 63) & 63) * 16);
      if (u == 8) {
#pragma unroll
        for (int q = 0; q < YA; ++q) zn[q] = lds_ld<int>(gw.ring_in + ((16 * sub + 16 + q + 63) & 63) * 16 + 8);
      }
      y_stage<PW>(st.ypass, u + YA < 16 ? rq[(u + YA) & 15].z : zn[(u + YA) & 15 & (YA - 1)], gw.lstride, pwq[u + YA]);
      lane_cells_g<KIND, K, BND, FLAGS>(st, tacc, u, rq[u], pwq[u], gw.wout + (16 * sub + u) * 16, a, cbase + u, sbk, skp);
    }
#pragma unroll
    for (int u = 0; u < YA; ++u)
#pragma unroll
      for (int q = 0; q < PW; ++q) pwc[u][q] = pwq[16 + u][q];
    // the sub-block's 16 records, back from the out-ring (the LDS works through one wave's accesses in order)
    const i32x2 orec = lds_ld<i32x2>(gw.outrd + 256 * sub);
    if (KIND == KIND_SW) {
#pragma unroll
      for (int k = 0; k < K; ++k) {
        const int v = (int)(sbk[k] >> 4) << 2;
        const bool better = v > st.best[k];
        st.best[k] = better ? v : st.best[k];
        st.bestc[k] = better ? cbase + 15 - (int)(sbk[k] & 15u) : st.bestc[k];
      }
    }
    {
      // the records of the next sub-block (columns Ts + 16 + lane, lanes 0..15), requested one sub-block ago: into
      // entries (column - 1) mod 64 of this wave's own in-ring.  Straight-line for every wave (a wave that is not the
      // group's top, or has given up, waits for nothing and stages into the dummy area): a branch around it makes
      // `pend` a phi, and the register copy behind the phi waits for the request just issued (vmcnt(0), 1 us per tick).
      int hH = 0, hF = 0;
      if (!handoff_acquire_g<KIND, K>(a, pend, rd, Ts + 16, lane, gen_r, hH, hF, live)) {
        live = 0;
        if (lane == 0) st_coherent(a.err, 1);
      }
      const int ent = (lane + 16 * (sub + 1) + 63) & 63;
      i32x2 hv = {hH, hF};
      lds_st<i32x2>(lane < 16 * live ? gw.ring_in + ent * 16 : gw.dummy, hv);
    }
    // the entries this sub-block consumed take the profile-row addresses of the same columns of the next block
    lds_st<int>((lane >> 4) == sub ? gw.ystage : gw.dummy, ycur);
    if (sub == 2) ynext = a.yenc[Tb + 128 + lane];
    if (sub == 3) {
      stores(Ts, tacc, orec);
    } else {
#pragma unroll
      for (int k = 0; k < K; ++k) tprev[k] = tacc[k];
      oprev = orec;
    }
    group_barrier();
  };
  sub_block(std::integral_constant<int, 0>());
  sub_block(std::integral_constant<int, 1>());
  sub_block(std::integral_constant<int, 2>());
  sub_block(std::integral_constant<int, 3>());
  ycur = gw.yrow0 + ynext * G::ROWB;
}

template <int KIND, int K, int FLAGS>
__global__ void __launch_bounds__(64 * kWavesPerBlock, 2) fill_group(const FillArgs a) {
  using G = Geo<KIND, K>;
  constexpr int PF = FLAGS | FLAG_PERSIST;
  constexpr int YA = GrpAhead<K>::YA;
  extern __shared__ __align__(16) char glds[];
  const int lbase = (int)(unsigned)(size_t)glds;  // the low half of a generic LDS pointer is the LDS address
  const int lane = threadIdx.x & 63;
  const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int NB = (a.n + 63 + 63) / 64;
  const int prof_bytes = a.alpha * 64 * G::PW * 4;
  GroupWave gw;
  gw.ring_in = lbase + w * kGrpRingBytes;
  gw.dummy = lbase + kGrpDummyOff + lane * 8;
  gw.wout = lane == 63 ? lbase + (w + 1) * kGrpRingBytes : gw.dummy;
  gw.outrd = lbase + (w + 1) * kGrpRingBytes + (lane & 15) * 16;
  gw.ystage = gw.ring_in + ((lane + 63) & 63) * 16 + 8;
  gw.yrow0 = lbase + kGrpProfOff + w * prof_bytes;
  gw.lstride = 4 * G::PW;
  asm volatile("" : "+v"(gw.lstride));  // keep it in a VGPR
  const int gword = lbase + kGrpWordOff;

  while (true) {
    // group ticket: strips s_first + 4 g .. + 3; nobody starts a group once an error is up
    if (threadIdx.x == 0) {
      int g = atomicAdd(a.ticket, 1);
      if (ld_coherent(a.err) != 0) g = 0x3fffffff;
      lds_st<int>(gword, g);
    }
    group_barrier();
    const int g = __builtin_amdgcn_readfirstlane(lds_ld<int>(gword));
    group_barrier();  // everybody has read the word before the next group's is written
    if (g > (a.s_last - a.s_first) / kWavesPerBlock) break;
    const int s = a.s_first + kWavesPerBlock * g + w;
    if (s > a.s_last) {  // the last group may be short: keep the barrier count
      for (int i = 0; i < 4 * NB + kGrpLag * (kWavesPerBlock - 1); ++i) group_barrier();
      continue;
    }
    int live = w == 0 ? 1 : 0;  // 1: this wave takes hand-off records from the group above
    const int rowg = s * 64 + lane;
    const int r0 = rowg * K;
    {
      const uint32_t* gp = a.prof + (size_t)s * a.alpha * 64 * G::PW;
      for (int i = lane; i < a.alpha * 64 * G::PW; i += 64) lds_st<int>(gw.yrow0 + 4 * i, (int)gp[i]);
    }
    Lane<KIND, K> st;
    strip_begin<KIND, K, FLAGS>(st, a, r0, false);
#pragma unroll
    for (int k = 0; k < K; ++k) st.Hg[k] = add_gap<(FLAGS & FLAG_FRAC) != 0>(st.H[k], a.go, a.fo) + 1;
    st.ypass = gw.yrow0;  // any valid address: the lanes' first lookups (columns < 0) only have to stay inside the profile
    const int gen_r = ho_gen((s / a.ring) % 3), gen_w = ho_gen(((s + 1) / a.ring) % 3);
    const int2* rd = a.bot2 + (size_t)(s % a.ring) * a.NP + 64 + min(lane, 15);
    int2* const wr_slot_base = a.bot2 + (size_t)((s + 1) % a.ring) * a.NP;
    int2* wr = wr_slot_base + 1 + (lane - 48);
    // block 0's profile-row addresses go into the ring now, block 1's wait in ycur
    lds_st<int>(gw.ystage, gw.yrow0 + (int)a.yenc[lane] * G::ROWB);
    int ycur = gw.yrow0 + (int)a.yenc[64 + lane] * G::ROWB, ynext = 0;
    int2 pend = make_int2(0, 0);
    {
      pend = ho_load(rd);
      int hH = 0, hF = 0;
      if (!handoff_acquire_g<KIND, K>(a, pend, rd, 0, lane, gen_r, hH, hF, live)) {
        live = 0;
        if (lane == 0) st_coherent(a.err, 1);
      }
      i32x2 hv = {hH, hF};
      lds_st<i32x2>(lane < 16 * live ? gw.ring_in + ((lane + 63) & 63) * 16 : gw.dummy, hv);
    }
    // the profile words of the first YA steps
    uint32_t pwc[YA][G::PW];
#pragma unroll
    for (int q = 0; q < YA; ++q) y_stage<G::PW>(st.ypass, lds_ld<int>(gw.ring_in + ((q + 63) & 63) * 16 + 8), gw.lstride, pwc[q]);
#ifdef SA_DEBUG_HANDOFF
    const long long dbg_t0 = wall_clock64();
#endif
    for (int i = 0; i < kGrpLag * w; ++i) group_barrier();
#pragma unroll 1
    for (int b = 0; b < NB; ++b) {
      const int Tb = 64 * b;
      const bool interior = (Tb >= 64) && (Tb + 63 < a.n);
      if (interior)
        run_block_g<KIND, K, false, PF>(st, a, s, Tb, lane, live, gw, pwc, ycur, ynext, pend, rd, wr, wr_slot_base, gen_r, gen_w);
      else
        run_block_g<KIND, K, true, PF>(st, a, s, Tb, lane, live, gw, pwc, ycur, ynext, pend, rd, wr, wr_slot_base, gen_r, gen_w);
    }
    for (int i = 0; i < kGrpLag * (kWavesPerBlock - 1 - w); ++i) group_barrier();
#ifdef SA_DEBUG_HANDOFF
    if (a.dbg && lane == 0) {
      a.dbg[4 * s + 0] = dbg_t0;
      a.dbg[4 * s + 1] = wall_clock64();
    }
#endif
    strip_end<KIND, K>(st, a, s, rowg, r0);  // garbage if a wait expired above: *err is up then and the host reports it
  }
}

}  // namespace sa
