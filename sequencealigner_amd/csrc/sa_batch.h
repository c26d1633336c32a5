// sa_batch.h -- inter-task Smith-Waterman for many independent pairs: the first fill pass of the
// reference's localalign() (main.cpp:1099-1128) per pair (score + end cell), and -- TRACE
// instances -- the same sweep with 2-bit direction codes stored per cell so that a second kernel
// can walk every pair's local alignment back to its 'X' cell (main.cpp:1182-1265 per pair).
//
// One wavefront per pair.  The pair's query (rows) is spread over the 64 lanes, K = ceil(m/64)
// rows per lane (K is a template parameter, 1..16, pairs are grouped by K on the host), and the
// wave sweeps the target with the same one-column-per-lane skew and DPP hand-off as the
// single-pair kernel -- but nothing leaves the registers: no traceback codes, no strip hand-off.
// Out-of-range columns need no predication: target positions outside [0,n) carry a PAD symbol
// whose substitution score is -128 for every row, so with non-positive gap penalties cells left
// of column 0 evaluate to exactly 0 (the reference's border) and cells right of column n-1 can
// never beat, or tie ahead of, a real cell in the row-major order the reference uses.
//
// First-row-major-maximum tracking (main.cpp:1124-1127) costs two VALU ops per cell: every row
// keeps max over columns of key = (H << CB) + (2^CB-1 - column); the larger key is the larger
// score and, among equal scores, the smaller column.
#pragma once
#include <hip/hip_runtime.h>
#include <climits>
#include <cstdint>

#include "sa_kernels.h"

namespace sa {

struct PairDesc {
  long long off_x, off_y;
  int m, n;
  int idx;   // position in the caller's arrays
  int pad;
  long long off_trace;  // TRACE: first dword of the pair's codes in BatchArgs::trace
  long long off_ops;    // TRACE: first byte of the pair's op slot (capacity m + n, 4-byte aligned)
};

struct BatchArgs {
  const uint8_t* seqs;
  const PairDesc* desc;   // the pairs of this launch
  int n_desc;
  const int8_t* sub;      // alpha x alpha
  const uint8_t* idx256;
  int alpha;
  int go, ge;             // floor of the penalties (SW: a fractional penalty acts exactly like its floor, see fill_impl)
  int fo, fe;             // NW: 4 (the register-scale unit) if the open / extend penalty has a fractional part, else 0
  int cgo, cge;           // NW: ceil of the penalties: the border chains stay <= 0, where trunc(v + g) = v + ceil(g)
  int CB;                 // column bits of the tracking key
  int ycap;               // LDS bytes reserved for the encoded target (sized for the longest target of the launch)
  int* score;
  int* end_i;
  int* end_j;
  int* err;               // set to 1 if any pair had an invalid residue
  // alignments (TRACE kernels + sw_batch_traceback)
  uint32_t* trace;        // per pair [ceil((n+63)/16)][64 lanes][K] dwords, 16 skewed steps of 2-bit codes each
  uint8_t* ops;           // per pair a slot of m + n bytes
  int* n_ops;
  int* start_i;
  int* start_j;
};

constexpr int kBatchMaxN = 4095;  // LDS room for the encoded target; the host admits n <= kBatchMaxN - 200
constexpr int kBatchMaxM = 1024;  // 64 lanes x 16 rows
constexpr int kBatchCB = 13;      // column bits of the first-maximum key (2^13 > kBatchMaxN + 128)

// (v << SH) + c as ONE instruction.  Written in C the compiler rebuilds c (a per-step counter) from its loop
// invariants and emits a shift plus a three-operand add per cell.
template <int SH>
__device__ __forceinline__ uint32_t shl_add(uint32_t v, uint32_t c) {
  uint32_t r;
  asm("v_lshl_add_u32 %0, %1, %2, %3" : "=v"(r) : "v"(v), "n"(SH), "v"(c));
  return r;
}

// max of three as ONE v_max3_i32.  e and f are known to be non-negative, so the compiler turns the outer signed
// max into an unsigned one and then cannot fuse the two into a max3.
__device__ __forceinline__ int max3_i32(int x, int y, int z) {
  int r;
  asm("v_max3_i32 %0, %1, %2, %3" : "=v"(r) : "v"(x), "v"(y), "v"(z));
  return r;
}

template <int K>
struct BatchGeo {
  static constexpr int PW = (K + 3) / 4;  // profile dwords per (symbol, lane): 1..4 (lane stride 12 B is conflict-free too)
  static constexpr int ROWB = 64 * PW * 4;
};

// TRACE: values are held as 4*score with the candidate's identity in the two low bits (M 2, E 1,
// F 0, zero cell 3), exactly as in the single-pair kernels (sa_kernels.h), so the direction code
// is the low two bits of the cell's maximum; the PAD score becomes -32 (4*(-32)+2 fits the int8
// profile; any negative PAD keeps the two properties above).
template <int K, bool TRACE>
__global__ void __launch_bounds__(64) sw_batch_kernel(const BatchArgs a) {
  using G = BatchGeo<K>;
  extern __shared__ uint32_t lds[];
  // LDS layout: prof[(alpha+1)][64][PW] dwords | yenc[a.ycap] bytes | map[256] | sub[alpha*alpha]
  const int lane = threadIdx.x;
  const PairDesc pd = a.desc[blockIdx.x];
  const int m = pd.m, n = pd.n, alpha = a.alpha;
  uint32_t* prof = lds;
  uint8_t* yenc = reinterpret_cast<uint8_t*>(lds + (alpha + 1) * 64 * G::PW);
  uint8_t* map = yenc + a.ycap;
  int8_t* subt = reinterpret_cast<int8_t*>(map + 256);

  for (int i = lane; i < 256; i += 64) map[i] = a.idx256[i];
  for (int i = lane; i < alpha * alpha; i += 64) subt[i] = a.sub[i];
  __syncthreads();

  // encode the target; positions outside [0,n) get the PAD symbol (= alpha)
  bool bad = false;
  const uint8_t* y = a.seqs + pd.off_y;
  const int ylen = ((n + 63 + 63) / 64) * 64 + 128;
  for (int i = lane; i < ylen; i += 64) {
    uint8_t v = (uint8_t)alpha;
    if (i < n) {
      v = map[y[i]];
      bad |= (v == 255);
      if (v == 255) v = 0;
    }
    yenc[i] = v;
  }
  // query profile of this lane's K rows
  const uint8_t* x = a.seqs + pd.off_x;
  int xi[K];
#pragma unroll
  for (int k = 0; k < K; ++k) {
    const int r = lane * K + k;
    int v = -1;
    if (r < m) {
      v = map[x[r]];
      bad |= (v == 255);
      if (v == 255) v = 0;
    }
    xi[k] = v;
  }
  for (int sym = 0; sym <= alpha; ++sym) {
    uint32_t w[G::PW];
#pragma unroll
    for (int i = 0; i < G::PW; ++i) w[i] = 0;
#pragma unroll
    for (int k = 0; k < K; ++k) {
      int v = (sym == alpha) ? (TRACE ? -32 : -128) : (xi[k] < 0 ? 0 : (int)subt[xi[k] * alpha + sym]);
      if (TRACE) v = 4 * v + 2;  // tagged M candidate; the host guarantees |sigma| <= 31
      w[k / 4] |= (uint32_t)(uint8_t)(int8_t)v << (8 * (k & 3));
    }
#pragma unroll
    for (int i = 0; i < G::PW; ++i) prof[(sym * 64 + lane) * G::PW + i] = w[i];
  }
  __syncthreads();

  const char* ldsLane = reinterpret_cast<const char*>(prof) + lane * 4 * G::PW;
  const int go = TRACE ? 4 * a.go : a.go, ge = TRACE ? 4 * a.ge : a.ge;
  // Column bits of the tracking key: a compile-time constant, so that the key is ONE v_lshl_add_u32 per cell
  // (as a launch argument it cost a shift and a three-operand add).  n + 63 + 64 < 2^13 (kBatchMaxN), and
  // 19 bits of score hold 127 x 1024.
  constexpr int CB = kBatchCB;
  int H[K], E[K];
  uint32_t bk[K];
#pragma unroll
  for (int k = 0; k < K; ++k) {
    H[k] = 0;
    E[k] = TRACE ? 1 : 0;
    bk[k] = 0;
  }
  uint32_t* tdst = TRACE ? a.trace + pd.off_trace + (size_t)lane * K : nullptr;
  int passH = 0, passF = 0, dprev = 0;
  int ypass = alpha * G::ROWB;
  const int cbase = (1 << CB) - 64;  // key low bits = cbase - column >= 0 for every column the sweep touches
  int cinv = cbase + lane;           // column = t - lane
  const int nsteps = ((n + 63 + 15) / 16) * 16;
  int yb = 0;

  auto load_prof = [&](int yv, uint32_t (&pw)[G::PW]) {
    if (G::PW == 1) {
      pw[0] = *reinterpret_cast<const uint32_t*>(ldsLane + yv);
    } else if (G::PW == 2) {
      const uint2 t = *reinterpret_cast<const uint2*>(ldsLane + yv);
      pw[0] = t.x;
      pw[G::PW - 1] = t.y;
    } else if (G::PW == 3) {
      const uint32_t* t = reinterpret_cast<const uint32_t*>(ldsLane + yv);
      pw[0] = t[0];
      pw[1 % G::PW] = t[1];
      pw[2 % G::PW] = t[2];
    } else {
      const uint4 t = *reinterpret_cast<const uint4*>(ldsLane + yv);
      pw[0] = t.x;
      pw[1 % G::PW] = t.y;
      pw[2 % G::PW] = t.z;
      pw[3 % G::PW] = t.w;
    }
  };

  // software pipeline: the profile word of step t+1 is fetched from LDS while step t computes
  yb = (int)yenc[lane] * G::ROWB;
  int yv = dpp_shr1(yb, ypass);
  yb = dpp_shl1(yb, yb);
  // two profile buffers used alternately (16 steps: back at buffer 0 when the loop comes round): a single
  // buffer refilled through a copy costs PW moves per step
  uint32_t pwb[2][G::PW];
  load_prof(yv, pwb[0]);

#pragma unroll 1
  for (int t0 = 0; t0 < nsteps; t0 += 16) {
    uint32_t tacc[TRACE ? K : 1];
    if (TRACE) {
#pragma unroll
      for (int k = 0; k < K; ++k) tacc[k] = 0;
    }
#pragma unroll
    for (int u = 0; u < 16; ++u) {
      const int t = t0 + u;
      // hand-off from the lane above (values of the previous step); lane 0 sees the zero border
      int up = dpp_shr1_zero(passH);
      int uf = dpp_shr1_zero(passF);
      // next step's symbol and profile word
      if (u == 15 && ((t0 + 16) & 63) == 0) yb = (int)yenc[t + 1 + lane] * G::ROWB;
      ypass = yv;
      const int yv_next = dpp_shr1(yb, ypass);
      yb = dpp_shl1_any(yb);  // lane 63 is reloaded before it matters
      const uint32_t (&pw)[G::PW] = pwb[u & 1];
      load_prof(yv_next, pwb[(u & 1) ^ 1]);

      int diag = dprev;
      dprev = up;
#pragma unroll
      for (int k = 0; k < K; ++k) {
        const int sg = (int)(int8_t)(pw[k / 4] >> (8 * (k & 3)));
        const int left = H[k];
        int f, e, h;
        if (TRACE) {
          f = max(max(up + go, uf + ge), 0);            // tag 0
          e = max(max(left + go + 1, E[k] + ge), 1);    // tag 1
          // M needs no floor of its own: it only feeds the cell maximum, where the 'X' floor (tag 3 beats
          // every tagged value <= 2) subsumes it
          const int ht = max(max3_i32(diag + sg, e, f), 3);  // a zero cell wins with tag 3 = 'X'
          tacc[k] = __builtin_amdgcn_alignbit((uint32_t)ht, tacc[k], 2);
          // pack here and now: left alone the compiler sinks the packing to the end of the sub-block and parks
          // every cell's maximum in an AGPR until then
          asm volatile("" : "+v"(tacc[k]));
          h = ht & ~3;
          bk[k] = max(bk[k], shl_add<CB - 2>((uint32_t)h, (uint32_t)cinv));
        } else {
          f = max(max(up + go, uf + ge), 0);
          e = max(max(left + go, E[k] + ge), 0);
          h = max3_i32(diag + sg, e, f);
          bk[k] = max(bk[k], shl_add<CB>((uint32_t)h, (uint32_t)cinv));
        }
        E[k] = e;
        diag = left;
        H[k] = h;
        up = h;
        uf = f;
      }
      passH = up;
      passF = uf;
      --cinv;
      yv = yv_next;
    }
    if (TRACE) {
#pragma unroll
      for (int k = 0; k < K; ++k) tdst[k] = tacc[k];
      tdst += 64 * K;
    }
  }

  // per lane: first row (ascending) with the best score; keys of rows >= m are ignored
  const uint32_t cmask = (1u << CB) - 1u;
  int bs = 0, brow = 0, bcol = 0;
#pragma unroll
  for (int k = 0; k < K; ++k) {
    const int r = lane * K + k;
    const int s = (int)(bk[k] >> CB);
    if (r < m && s > bs) {
      bs = s;
      brow = r;
      bcol = cbase - (int)(bk[k] & cmask);
    }
  }
  unsigned long long key = bs > 0 ? (((unsigned long long)(uint32_t)bs << 32) | ((unsigned long long)(0xffffu - brow) << 16) |
                                     (unsigned long long)(0xffffu - bcol))
                                  : 0ull;
  for (int o = 32; o > 0; o >>= 1) {
    const unsigned long long other = __shfl_xor(key, o);
    key = other > key ? other : key;
  }
  const unsigned long long anybad = __ballot(bad);
  if (lane == 0) {
    if (anybad) {
      a.score[pd.idx] = INT_MIN;
      a.end_i[pd.idx] = 0;
      a.end_j[pd.idx] = 0;
      *a.err = 1;
    } else if (key == 0ull) {
      a.score[pd.idx] = 0;
      a.end_i[pd.idx] = 0;
      a.end_j[pd.idx] = 0;
    } else {
      a.score[pd.idx] = (int)(key >> 32);
      a.end_i[pd.idx] = (int)(0xffffu - ((key >> 16) & 0xffffu)) + 1;
      a.end_j[pd.idx] = (int)(0xffffu - (key & 0xffffu)) + 1;
    }
  }
}

// ---- global alignment (Needleman-Wunsch) of many pairs --------------------------------------
// Same sweep as the TRACE instances above with the reference's NW recurrence (main.cpp:622-672:
// no floors, tie-break M over E over F via the tags) and its borders (main.cpp:601-620, 719-729):
// H[i][0] = go + (i-1) max(go, ge), E[i][0] = -inf, H[0][j] = go + (j-1) ge, F[0][j] = -inf,
// H[0][0] = 0 -- closed forms of the reference's chains for the integral penalties this path
// admits.  Borders need the cells outside [0, n) to keep their hands off the state, so sub-blocks
// that touch the first 64 or the last 64+ steps run a masked instance (MASK) that freezes a
// lane's registers while its column is outside the matrix; everything in between is unmasked.
// After the sweep every lane holds H[i][n] of its rows, the score is H[m][n].
// FRAC: fractional penalties, int + double truncated toward zero per operation (main.cpp:603-604, 625-634): inside
// the matrix add_gap<true> (floor of the penalty + 1 for a negative sum, sa_kernels.h); the borders are sums of
// non-positive terms, where the truncation is "+ ceil(penalty)", so their closed forms keep their shape with
// cgo / cge: H[i][0] = cgo + (i-1) max(cgo, cge), H[0][j] = cgo + (j-1) cge.
template <int K, bool MASK, bool FRAC>
__device__ __forceinline__ void nw_batch_steps(int t0, int lane, int n, int go, int ge, int fo, int fe, int cgo, int cge,
                                               const char* ldsLane, const uint8_t* yenc,
                                               int (&H)[K], int (&E)[K], int& passH, int& passF, int& dprev, int& ypass, int& yb,
                                               int& yv, uint32_t (&pw_io)[BatchGeo<K>::PW], uint32_t* tdst) {
  using G = BatchGeo<K>;
  uint32_t tacc[K];
#pragma unroll
  for (int k = 0; k < K; ++k) tacc[k] = 0;
  uint32_t pwb[2][G::PW];  // two profile buffers used alternately: no copy per step
#pragma unroll
  for (int i = 0; i < G::PW; ++i) pwb[0][i] = pw_io[i];
#pragma unroll
  for (int u = 0; u < 16; ++u) {
    const uint32_t (&pw)[G::PW] = pwb[u & 1];
    uint32_t (&pw_next)[G::PW] = pwb[(u & 1) ^ 1];
    const int t = t0 + u;
    // lane 0 sits under row 0 of the matrix: H[0][t+1] = cgo + t cge (tag-free), F[0][.] = -inf
    int up = dpp_shr1(cgo + t * cge, passH);
    int uf = dpp_shr1(kNeg, passF);
    if (u == 15 && ((t0 + 16) & 63) == 0) yb = (int)yenc[t + 1 + lane] * G::ROWB;
    ypass = yv;
    const int yv_next = dpp_shr1(yb, ypass);
    yb = dpp_shl1_any(yb);  // lane 63 is reloaded before it matters
    if (G::PW == 1) {
      pw_next[0] = *reinterpret_cast<const uint32_t*>(ldsLane + yv_next);
    } else if (G::PW == 2) {
      const uint2 q = *reinterpret_cast<const uint2*>(ldsLane + yv_next);
      pw_next[0] = q.x;
      pw_next[G::PW - 1] = q.y;
    } else if (G::PW == 3) {
      const uint32_t* q = reinterpret_cast<const uint32_t*>(ldsLane + yv_next);
      pw_next[0] = q[0];
      pw_next[1 % G::PW] = q[1];
      pw_next[2 % G::PW] = q[2];
    } else {
      const uint4 q = *reinterpret_cast<const uint4*>(ldsLane + yv_next);
      pw_next[0] = q.x;
      pw_next[1 % G::PW] = q.y;
      pw_next[2 % G::PW] = q.z;
      pw_next[3 % G::PW] = q.w;
    }
    const int c = t - lane;
    const bool act = !MASK || (c >= 0 && c < n);
    int diag = dprev;
    if (MASK)
      dprev = act ? up : dprev;
    else
      dprev = up;
    int f_last = passF;
#pragma unroll
    for (int k = 0; k < K; ++k) {
      const int sg = (int)(int8_t)(pw[k / 4] >> (8 * (k & 3)));  // 4 sigma + 2
      const int left = H[k];
      int f = max(add_gap<FRAC>(up, go, fo), add_gap<FRAC>(uf, ge, fe));            // tag 0
      int e = max(add_gap<FRAC>(left, go, fo) + 1, add_gap<FRAC>(E[k], ge, fe));    // tag 1
      const int mv = diag + sg;                 // tag 2
      const int ht = max(max(mv, e), f);
      tacc[k] = __builtin_amdgcn_alignbit((uint32_t)ht, tacc[k], 2);
      int h = ht & ~3;
      if (MASK) {
        h = act ? h : left;
        e = act ? e : E[k];
        f = act ? f : f_last;
      }
      E[k] = e;
      diag = left;
      H[k] = h;
      up = h;
      uf = f;
      f_last = f;
    }
    passH = up;
    passF = uf;
    yv = yv_next;
  }
#pragma unroll
  for (int i = 0; i < G::PW; ++i) pw_io[i] = pwb[0][i];
#pragma unroll
  for (int k = 0; k < K; ++k) tdst[k] = tacc[k];
}

template <int K, bool FRAC>
__global__ void __launch_bounds__(64) nw_batch_kernel(const BatchArgs a) {
  using G = BatchGeo<K>;
  extern __shared__ uint32_t lds[];
  const int lane = threadIdx.x;
  const PairDesc pd = a.desc[blockIdx.x];
  const int m = pd.m, n = pd.n, alpha = a.alpha;
  uint32_t* prof = lds;
  uint8_t* yenc = reinterpret_cast<uint8_t*>(lds + (alpha + 1) * 64 * G::PW);
  uint8_t* map = yenc + a.ycap;
  int8_t* subt = reinterpret_cast<int8_t*>(map + 256);
  for (int i = lane; i < 256; i += 64) map[i] = a.idx256[i];
  for (int i = lane; i < alpha * alpha; i += 64) subt[i] = a.sub[i];
  __syncthreads();
  bool bad = false;
  const uint8_t* y = a.seqs + pd.off_y;
  const int ylen = ((n + 63 + 63) / 64) * 64 + 128;
  for (int i = lane; i < ylen; i += 64) {
    uint8_t v = (uint8_t)alpha;
    if (i < n) {
      v = map[y[i]];
      bad |= (v == 255);
      if (v == 255) v = 0;
    }
    yenc[i] = v;
  }
  const uint8_t* x = a.seqs + pd.off_x;
  int xi[K];
#pragma unroll
  for (int k = 0; k < K; ++k) {
    const int r = lane * K + k;
    int v = -1;
    if (r < m) {
      v = map[x[r]];
      bad |= (v == 255);
      if (v == 255) v = 0;
    }
    xi[k] = v;
  }
  for (int sym = 0; sym <= alpha; ++sym) {
    uint32_t w[G::PW];
#pragma unroll
    for (int i = 0; i < G::PW; ++i) w[i] = 0;
#pragma unroll
    for (int k = 0; k < K; ++k) {
      int v = (sym == alpha || xi[k] < 0) ? 0 : (int)subt[xi[k] * alpha + sym];
      v = 4 * v + 2;
      w[k / 4] |= (uint32_t)(uint8_t)(int8_t)v << (8 * (k & 3));
    }
#pragma unroll
    for (int i = 0; i < G::PW; ++i) prof[(sym * 64 + lane) * G::PW + i] = w[i];
  }
  __syncthreads();

  const char* ldsLane = reinterpret_cast<const char*>(prof) + lane * 4 * G::PW;
  const int go = 4 * a.go, ge = 4 * a.ge, fo = a.fo, fe = a.fe;
  const int cgo = 4 * a.cgo, cge = 4 * a.cge, gmax = max(cgo, cge);
  int H[K], E[K];
#pragma unroll
  for (int k = 0; k < K; ++k) {
    const int i = lane * K + k + 1;          // matrix row of this register
    H[k] = cgo + (i - 1) * gmax;             // H[i][0]
    E[k] = kNeg | 1;                         // E[i][0] = -inf, tag 1
  }
  const int r0 = lane * K;                   // H[r0][0]: the diagonal of the lane's first cell
  int dprev = r0 == 0 ? 0 : cgo + (r0 - 1) * gmax;
  int passH = H[K - 1], passF = kNeg;
  int ypass = alpha * G::ROWB;
  const int nsteps = ((n + 63 + 15) / 16) * 16;
  int yb = (int)yenc[lane] * G::ROWB;
  int yv = dpp_shr1(yb, ypass);
  yb = dpp_shl1(yb, yb);
  uint32_t pw[G::PW];
  if (G::PW == 1) {
    pw[0] = *reinterpret_cast<const uint32_t*>(ldsLane + yv);
  } else if (G::PW == 2) {
    const uint2 q = *reinterpret_cast<const uint2*>(ldsLane + yv);
    pw[0] = q.x;
    pw[G::PW - 1] = q.y;
  } else if (G::PW == 3) {
    const uint32_t* q = reinterpret_cast<const uint32_t*>(ldsLane + yv);
    pw[0] = q[0];
    pw[1 % G::PW] = q[1];
    pw[2 % G::PW] = q[2];
  } else {
    const uint4 q = *reinterpret_cast<const uint4*>(ldsLane + yv);
    pw[0] = q.x;
    pw[1 % G::PW] = q.y;
    pw[2 % G::PW] = q.z;
    pw[3 % G::PW] = q.w;
  }
  uint32_t* tdst = a.trace + pd.off_trace + (size_t)lane * K;
#pragma unroll 1
  for (int t0 = 0; t0 < nsteps; t0 += 16) {
    // unmasked only if every lane's column is inside [0, n) for all 16 steps
    if (t0 >= 64 && t0 + 15 < n)
      nw_batch_steps<K, false, FRAC>(t0, lane, n, go, ge, fo, fe, cgo, cge, ldsLane, yenc, H, E, passH, passF, dprev, ypass, yb, yv, pw, tdst);
    else
      nw_batch_steps<K, true, FRAC>(t0, lane, n, go, ge, fo, fe, cgo, cge, ldsLane, yenc, H, E, passH, passF, dprev, ypass, yb, yv, pw, tdst);
    tdst += 64 * K;
  }
  // H[m][n]: row m-1 lives in lane (m-1)/K, register (m-1)%K
  const int lm = (m - 1) / K, km = (m - 1) - lm * K;
  int hm = 0;
#pragma unroll
  for (int k = 0; k < K; ++k)
    if (k == km) hm = H[k];
  hm = __shfl(hm, lm);
  const unsigned long long anybad = __ballot(bad);
  if (lane == 0) {
    a.score[pd.idx] = anybad ? INT_MIN : (hm >> 2);
    a.end_i[pd.idx] = anybad ? 0 : m;
    a.end_j[pd.idx] = anybad ? 0 : n;
    if (anybad) *a.err = 1;
  }
}

// One thread per pair: from (m, n) along the stored winners (M 2, E 1, F 0) to a border, then the
// border walk of the reference (row 0: left moves, column 0: up moves; main.cpp:781-816).
static __global__ void nw_batch_traceback(const BatchArgs a) {
  const int q = blockIdx.x * blockDim.x + threadIdx.x;
  if (q >= a.n_desc) return;
  const PairDesc pd = a.desc[q];
  const int K = (pd.m + 63) / 64;
  int i = pd.m, j = pd.n;
  const uint32_t* tr = a.trace + pd.off_trace;
  uint32_t* out = reinterpret_cast<uint32_t*>(a.ops + pd.off_ops);
  int nops = 0;
  uint32_t acc = 0;
  if (a.score[pd.idx] != INT_MIN) {
    while (i > 0 || j > 0) {
      uint32_t op;
      if (i == 0) {
        op = 1u;
      } else if (j == 0) {
        op = 2u;
      } else {
        const int r = i - 1, c = j - 1;
        const int l = r / K, k = r - l * K;
        const int t = c + l;
        op = 2u - ((tr[((size_t)(t >> 4) * 64 + l) * K + k] >> (2 * (t & 15))) & 3u);
      }
      acc |= op << (8 * (nops & 3));
      if ((nops & 3) == 3) {
        out[nops >> 2] = acc;
        acc = 0;
      }
      ++nops;
      if (op != 1u) --i;
      if (op != 2u) --j;
    }
    if (nops & 3) out[nops >> 2] = acc;
  }
  a.n_ops[pd.idx] = nops;
  a.start_i[pd.idx] = 0;
  a.start_j[pd.idx] = 0;
}

// ---- LCS of many pairs ------------------------------------------------------------------------
// lcs() (main.cpp:1575-1640) per pair: raw byte equality, tie between up and left goes to up.  Tagged
// like the single-pair LCS kernel (registers hold 4 L + 1; diagonal 3, up 2, left 1).  No masking is
// needed anywhere: columns outside [0, n) carry a symbol that equals no byte, so cells left of
// column 0 stay at the zero border and cells right of column n-1 keep L[i][n] (L is monotone), and
// rows beyond m-1 only ever feed rows below them.
template <int K>
__global__ void __launch_bounds__(64) lcs_batch_kernel(const BatchArgs a) {
  extern __shared__ uint32_t lds[];
  uint8_t* ybytes = reinterpret_cast<uint8_t*>(lds);
  const int lane = threadIdx.x;
  const PairDesc pd = a.desc[blockIdx.x];
  const int m = pd.m, n = pd.n;
  const uint8_t* y = a.seqs + pd.off_y;
  const int ylen = ((n + 63 + 63) / 64) * 64 + 128;
  for (int i = lane; i < ylen; i += 64) ybytes[i] = i < n ? y[i] : 0;
  const uint8_t* x = a.seqs + pd.off_x;
  uint32_t xr[K];
#pragma unroll
  for (int k = 0; k < K; ++k) {
    const int r = lane * K + k;
    xr[k] = r < m ? (uint32_t)x[r] : 0x200u;  // a padding row matches nothing
  }
  __syncthreads();
  auto ysym = [&](int pos) { return pos < n ? (int)ybytes[pos] : 0x100; };  // a padding column matches nothing
  int H[K];
#pragma unroll
  for (int k = 0; k < K; ++k) H[k] = 1;
  int passH = 1, dprev = 1;
  int ypass = 0x100;
  const int nsteps = ((n + 63 + 15) / 16) * 16;
  int yb = ysym(lane);
  uint32_t* tdst = a.trace + pd.off_trace + (size_t)lane * K;
#pragma unroll 1
  for (int t0 = 0; t0 < nsteps; t0 += 16) {
    uint32_t tacc[K];
#pragma unroll
    for (int k = 0; k < K; ++k) tacc[k] = 0;
#pragma unroll
    for (int u = 0; u < 16; ++u) {
      const int t = t0 + u;
      int up = dpp_shr1(1, passH);  // lane 0 sits under row 0: L = 0
      const int yv = dpp_shr1(yb, ypass);
      yb = dpp_shl1_any(yb);  // lane 63 is reloaded before it matters
      if (u == 15 && ((t0 + 16) & 63) == 0) yb = ysym(t + 1 + lane);
      ypass = yv;
      int diag = dprev;
      dprev = up;
#pragma unroll
      for (int k = 0; k < K; ++k) {
        const int left = H[k];
        const int dc = (xr[k] == (uint32_t)yv) ? diag + 6 : 0;
        const int ht = max(max(dc, up + 1), left);
        tacc[k] = __builtin_amdgcn_alignbit((uint32_t)ht, tacc[k], 2);
        int h;
        asm("v_and_or_b32 %0, %1, -4, 1" : "=v"(h) : "v"(ht));  // (ht & ~3) | 1 in one op, as in sa_kernels.h
        diag = left;
        H[k] = h;
        up = h;
      }
      passH = up;
    }
#pragma unroll
    for (int k = 0; k < K; ++k) tdst[k] = tacc[k];
    tdst += 64 * K;
  }
  const int lm = (m - 1) / K, km = (m - 1) - lm * K;
  int hm = 0;
#pragma unroll
  for (int k = 0; k < K; ++k)
    if (k == km) hm = H[k];
  hm = __shfl(hm, lm);
  if (lane == 0) {
    a.score[pd.idx] = hm >> 2;
    a.end_i[pd.idx] = m;
    a.end_j[pd.idx] = n;
  }
}

// One thread per pair: lcs() traceback (main.cpp:1602-1635) over the stored codes (3 diagonal, 2 up,
// 1 left), then the drains along the borders.
static __global__ void lcs_batch_traceback(const BatchArgs a) {
  const int q = blockIdx.x * blockDim.x + threadIdx.x;
  if (q >= a.n_desc) return;
  const PairDesc pd = a.desc[q];
  const int K = (pd.m + 63) / 64;
  int i = pd.m, j = pd.n;
  const uint32_t* tr = a.trace + pd.off_trace;
  uint32_t* out = reinterpret_cast<uint32_t*>(a.ops + pd.off_ops);
  int nops = 0;
  uint32_t acc = 0;
  while (i > 0 || j > 0) {
    uint32_t op;
    if (i == 0) {
      op = 1u;
    } else if (j == 0) {
      op = 2u;
    } else {
      const int r = i - 1, c = j - 1;
      const int l = r / K, k = r - l * K;
      const int t = c + l;
      const uint32_t raw = (tr[((size_t)(t >> 4) * 64 + l) * K + k] >> (2 * (t & 15))) & 3u;
      op = raw == 3u ? 0u : raw;
    }
    acc |= op << (8 * (nops & 3));
    if ((nops & 3) == 3) {
      out[nops >> 2] = acc;
      acc = 0;
    }
    ++nops;
    if (op != 1u) --i;
    if (op != 2u) --j;
  }
  if (nops & 3) out[nops >> 2] = acc;
  a.n_ops[pd.idx] = nops;
  a.start_i[pd.idx] = 0;
  a.start_j[pd.idx] = 0;
}

// dwords of codes one pair occupies
inline long long batch_trace_dwords(int m, int n) {
  const int K = (m + 63) / 64;
  const long long nsteps = ((n + 63 + 15) / 16) * 16;
  return nsteps / 16 * 64 * K;
}

// One thread per pair follows the pointers from the pair's end cell to its 'X' cell
// (main.cpp:1226-1262) and writes the moves, end -> start, four to a dword.
static __global__ void sw_batch_traceback(const BatchArgs a) {
  const int q = blockIdx.x * blockDim.x + threadIdx.x;
  if (q >= a.n_desc) return;
  const PairDesc pd = a.desc[q];
  const int K = (pd.m + 63) / 64;
  int i = a.end_i[pd.idx], j = a.end_j[pd.idx];
  const int sc = a.score[pd.idx];
  const uint32_t* tr = a.trace + pd.off_trace;
  uint32_t* out = reinterpret_cast<uint32_t*>(a.ops + pd.off_ops);
  int nops = 0;
  uint32_t acc = 0;
  if (sc > 0) {
    while (i > 0 && j > 0) {
      const int r = i - 1, c = j - 1;
      const int l = r / K, k = r - l * K;
      const int t = c + l;
      const uint32_t raw = (tr[((size_t)(t >> 4) * 64 + l) * K + k] >> (2 * (t & 15))) & 3u;
      if (raw == 3u) break;  // 'X'
      const uint32_t op = 2u - raw;
      acc |= op << (8 * (nops & 3));
      if ((nops & 3) == 3) {
        out[nops >> 2] = acc;
        acc = 0;
      }
      ++nops;
      if (op != 1u) --i;
      if (op != 2u) --j;
    }
    if (nops & 3) out[nops >> 2] = acc;
  } else {
    i = j = 0;
  }
  a.n_ops[pd.idx] = nops;
  a.start_i[pd.idx] = sc > 0 ? i : 0;
  a.start_j[pd.idx] = sc > 0 ? j : 0;
}

// The seed_aligner's perform_sw_in_window traceback (main_fmindex.cpp:989-1030) for negative gap-open
// penalties: its state machine starts in the match state at the best cell (a cell that ends a gap cannot
// be the maximum) and, as written, re-enters the match state after every diagonal move, so the
// "alignment" is the ungapped diagonal through the best cell for as long as S > 0 -- and S > 0 is
// exactly "the stored code is not the zero-cell tag 3".  One thread per window counts those steps.
static __global__ void sw_batch_diag_walk(const BatchArgs a) {
  const int q = blockIdx.x * blockDim.x + threadIdx.x;
  if (q >= a.n_desc) return;
  const PairDesc pd = a.desc[q];
  const int K = (pd.m + 63) / 64;
  int i = a.end_i[pd.idx], j = a.end_j[pd.idx];
  const int sc = a.score[pd.idx];
  const uint32_t* tr = a.trace + pd.off_trace;
  int len = 0;
  if (sc > 0) {
    while (i > 0 && j > 0) {
      const int r = i - 1, c = j - 1;
      const int l = r / K, k = r - l * K;
      const int t = c + l;
      if (((tr[((size_t)(t >> 4) * 64 + l) * K + k] >> (2 * (t & 15))) & 3u) == 3u) break;
      ++len;
      --i;
      --j;
    }
  } else {
    i = j = 0;
  }
  a.n_ops[pd.idx] = len;
  a.start_i[pd.idx] = sc > 0 ? i : 0;
  a.start_j[pd.idx] = sc > 0 ? j : 0;
}

// room for a target of n residues: the kernels touch ((n + 126) / 64) * 64 + 128 encoded bytes
inline int batch_ycap(int nmax) { return ((nmax + 126) / 64) * 64 + 128 + 64; }

template <int K>
inline size_t batch_lds_bytes(int alpha, int ycap) {
  return (size_t)(alpha + 1) * 64 * BatchGeo<K>::PW * 4 + (size_t)ycap + 256 + (size_t)alpha * alpha + 16;
}

}  // namespace sa
