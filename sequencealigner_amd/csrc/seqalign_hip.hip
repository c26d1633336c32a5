// seqalign_hip.hip -- host side of libseqalign_hip.so: the C-ABI declared in
// include/seqalign_hip.h on top of the gfx950 kernels in sa_kernels.h.
//
// What runs where: residue validation / encoding, the two O(m+n) border chains and the
// launch schedule are host code; the O(m*n) fill, the SW arg-max and the traceback walk
// are device code.  There is no CPU fallback for any of the device parts.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <mutex>
#include <vector>

#include "../../include/seqalign_hip.h"
#include "sa_batch.h"
#include "sa_kernels.h"
#include "sa_group.h"
#include "sa_launch.h"
#include "sa_noalloc.h"
#include "sa_lcsbits.h"
#include "sa_tables.h"

using namespace sa;

namespace {

struct Buf {
  void* p = nullptr;
  size_t cap = 0;
};
struct HostBuf {  // page-locked host staging, grow-only
  void* p = nullptr;
  size_t cap = 0;
};

}  // namespace

struct sa_ctx {
  int device = 0;
  hipStream_t own_stream = nullptr;
  hipStream_t copy_stream = nullptr;  // batch alignments: a chunk's moves leave the device while the next chunk computes
  hipEvent_t ev_chunk = nullptr;
  hipStream_t stream = nullptr;
  hipEvent_t ev[7] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};  // [4]: stripe flags reset; [5], [6]: around a recomputed band
  std::string err;
  int force_K = 0, force_C = 0;
  size_t batch_trace_budget = (size_t)32 << 30;  // sa_sw_batch_align: device bytes of direction codes per chunk
  int persist = 1;  // 1: persistent strip pipeline (one launch) where it applies
  int bits_wpb = 0; // bit-vector LCS: waves per workgroup, 0 = automatic (option "lcs_bits_waves_per_block", for measurements)
  int lcs_bits = 1; // 1: LCS of a whole pair over at most 255 distinct bytes runs as bit-vector arithmetic (sa_lcsbits.hip)
  int wps = 1;      // persistent kernel: waves per SIMD to launch
  int group = 0;    // 1: NW / SW fills of a whole matrix run fill_group (sa_group.h: four strips per workgroup, LDS hand-offs)
  int max_wg = 0;   // persistent kernels: at most this many workgroups per launch (0: as many as the GPU holds); tests use it to
                    // make small inputs take several rounds of tickets (ring-slot reuse, generations)
  int n_cu = 256;
  // cached device workspaces
  Buf xenc, yenc, prof, sub, colb, bot, state, trace, rowbest, small, ops;
  // batch workspaces
  Buf b_seq, b_desc, b_out;
  HostBuf h_desc, h_out;  // pair descriptors / per-pair results: copied at PCIe speed, never zero-filled
  const uint8_t* dev_pool = nullptr;  // sa_batch_set_device_pool: the batch entry points read the sequences here
  int64_t dev_pool_bytes = 0;
  Buf lbuf, dpbuf, pbuf;
  Buf vlb, vrb;  // bit-vector LCS over column stripes: the stripe's left / right border column as bits
  // geometry of the last fill, for the traceback that follows it
  struct {
    int valid = 0, kind = 0, m = 0, n = 0, K = 0, W = 0, first_stripe = 1;
    // checkpointed fill: the codes did not fit, only every strip's bottom row was kept; the
    // traceback recomputes `band` strips at a time from these arguments
    int ckpt = 0, band = 0, S = 0, lprof = 0;
    int bits = 0;  // the codes are the bit vectors of sa_lcsbits.hip (W = steps per strip)
    sa::FillArgs args;
    size_t lds = 0;
  } last;
  size_t trace_budget = 0;  // bytes of codes one call may hold; 0 = 88 % of the device memory
  int force_band = 0;       // > 0: always checkpoint and recompute this many strips at a time
  // sa_sw_batch_noalloc: this device's address of `scratch` (sa_noalloc.hip; looked up on the first call) and a page-locked
  // staging block set up by sa_create
  unsigned char* na_scratch = nullptr;
  unsigned char* na_host = nullptr;
};

namespace {

int fail(sa_ctx* ctx, int code, const char* fmt, ...) {
  char buf[512];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof buf, fmt, ap);
  va_end(ap);
  if (ctx) ctx->err = buf;
  return code;
}

#define HIPCK(ctx, call)                                                                            \
  do {                                                                                              \
    hipError_t e_ = (call);                                                                         \
    if (e_ != hipSuccess)                                                                           \
      return fail(ctx, e_ == hipErrorOutOfMemory ? SA_ERR_NOMEM : SA_ERR_HIP, "%s failed: %s (%s:%d)", #call, \
                  hipGetErrorString(e_), __FILE__, __LINE__);                                       \
  } while (0)

int ensure_host(sa_ctx* ctx, HostBuf& b, size_t bytes) {
  if (bytes <= b.cap) return SA_OK;
  if (b.p) (void)hipHostFree(b.p);
  b.p = nullptr;
  b.cap = 0;
  const size_t want = (bytes + (bytes >> 2) + 4095) & ~size_t(4095);
  hipError_t e = hipHostMalloc(&b.p, want, hipHostMallocDefault);
  if (e != hipSuccess) {
    (void)hipGetLastError();
    b.p = nullptr;
    return fail(ctx, SA_ERR_NOMEM, "hipHostMalloc(%zu bytes) failed: %s", want, hipGetErrorString(e));
  }
  b.cap = want;
  return SA_OK;
}

int ensure(sa_ctx* ctx, Buf& b, size_t bytes) {
  if (bytes <= b.cap) return SA_OK;
  if (b.p) HIPCK(ctx, hipFree(b.p));
  b.p = nullptr;
  b.cap = 0;
  size_t want = (bytes + 255) & ~size_t(255);
  hipError_t e = hipMalloc(&b.p, want);
  if (e != hipSuccess) {
    (void)hipGetLastError();
    b.p = nullptr;
    return fail(ctx, SA_ERR_NOMEM, "hipMalloc(%zu bytes) failed: %s", want, hipGetErrorString(e));
  }
  b.cap = want;
  return SA_OK;
}

void release(Buf& b) {
  if (b.p) (void)hipFree(b.p);
  b.p = nullptr;
  b.cap = 0;
}

// ctx->ops: [0, m + n + 64) one byte per move (what goes back to the host), then the walker's packed moves
size_t ops_bytes_offset(int m, int n) { return ((size_t)m + n + 64 + 255) & ~size_t(255); }

void launch_expand_ops(sa_ctx* ctx, int m, int n, long long n_max, const long long* d_hdr) {
  const int blocks = (int)std::max<long long>(1, std::min<long long>((n_max + 255) / 256, 4096));
  hipLaunchKernelGGL(expand_ops, dim3(blocks), dim3(256), 0, ctx->stream, (const unsigned long long*)((char*)ctx->ops.p + ops_bytes_offset(m, n)),
                     (uint8_t*)ctx->ops.p, d_hdr);
}

// int + double truncated back to int, as every "score + GAP" of the reference (main.cpp:603).
inline int add_gap(int v, double g) { return (int)((double)v + g); }

struct Tuning {
  int K, C;
};

constexpr double kHandoffLatencyUs = 2.9;  // record store (sc1) -> visible to the other CU -> its poll returns, MI355X

// Pick rows-per-lane K and tile length C by a small cost model: a wave issues one VALU
// instruction per 4 cycles and one wave saturates its SIMD (measured, profiles/), so the time
// is max(critical path over launches, total work / 1024 SIMDs).
// n_upstream: columns owned by the stripes to the left (multi-GPU): this stripe cannot start
// before its first strip has crossed them, at one column per step.
// workgroups one persistent launch may use: one per CU and wave-per-SIMD, fewer when the option says so
int persist_wg_cap(const sa_ctx* ctx) {
  const int cap = ctx->n_cu * std::max(1, ctx->wps);
  return ctx->max_wg > 0 ? std::min(cap, ctx->max_wg) : cap;
}

// rows per lane forced by sa_set_tuning; 16 exists only for the profile-driven persistent LCS, everything else takes 8 then
int forced_K(const sa_ctx* ctx, bool has16 = false) { return (ctx->force_K == 16 && !has16) ? 8 : ctx->force_K; }

Tuning choose_tuning(const sa_ctx* ctx, int kind, int64_t m, int64_t n, int64_t n_upstream = 0) {
  const double cell = kind == KIND_LCS ? 8.0 : (kind == KIND_SW ? 19.0 : 15.0);
  const double step0 = kind == KIND_LCS ? 9.0 : 13.0;
  const double launch_cycles = 2.5e-6 * 2.4e9;
  double best = 1e300;
  Tuning t{4, 256};
  for (int K : {1, 2, 4, 8}) {
    for (int C : {64, 128, 256, 512}) {
      if (forced_K(ctx) && K != forced_K(ctx)) continue;
      if (ctx->force_C && C != ctx->force_C) continue;
      const double S = std::ceil((double)m / (64.0 * K));
      const double L = std::ceil((64.0 * S + n + 63) / C) + S;
      const double tile = C * (cell * K + step0) * 4.0 + 3000.0;  // + state/profile load
      const double crit = L * (tile + launch_cycles) + (double)n_upstream * ((cell * K + step0) * 4.0 + launch_cycles / C);
      const double thr = S * std::ceil((n + 63.0) / C) * tile / 1024.0;
      const double est = std::max(crit, thr) + 0.05 * std::min(crit, thr);
      if (est < best) {
        best = est;
        t = {K, C};
      }
    }
  }
  if (forced_K(ctx)) t.K = forced_K(ctx);
  if (ctx->force_C) t.C = ctx->force_C;
  return t;
}

// Steps by which a strip trails the strip above it in the persistent pipeline: the lane skew of a strip (64), the
// hand-off granularity (16), the distance at which the records are requested (16) and the store-to-load latency of an
// agent-scope hand-off (measured start-to-start lags, tools/perf_handoff.py: 9.1 us at 62 ns per step, 19.3 us at 164).
double strip_lag(double step_cycles) { return 96.0 + kHandoffLatencyUs * 1e-6 * 2.4e9 / step_cycles; }

// Steps until the last of S strips of n columns is done when `waves` resident waves draw strips in
// order: strip s starts when the strip above is `lag` steps ahead AND a wave is free, i.e.
// start(s) = max(lag s, start(s - waves) + n).
double pipeline_steps(double S, double waves, double n, double lag) {
  double start = 0;
  for (double j = 0; j * waves <= S - 1; j += 1) start = std::max(start, lag * (S - 1 - j * waves) + j * n);
  return start + n;
}

// Rows per lane for the persistent pipeline on one GPU: the schedule above priced with the measured
// step cost (choose_stripe_K below).  E.g. 786 432 rows: K=8 gives 1536 strips = 2 rounds with half
// the waves idle in the second, K=4 gives 3 full rounds of cheaper steps and wins by 10 %.
// The profile-driven LCS also has a 16-rows-per-lane instance: its cell is 5 instructions, so the unrolled block still
// fits the instruction cache (the NW / SW one does not), and 1 Mb of rows becomes 977 strips = ONE round.
int choose_persist_K(const sa_ctx* ctx, int kind, int64_t m, int64_t n, bool lprof = false, bool has16 = false) {
  if (ctx->force_K) return forced_K(ctx, has16);
  const double cell = kind == KIND_LCS ? (lprof ? 5.0 : 7.0) : (kind == KIND_SW ? 12.0 : 10.0);
  const double waves = 4.0 * persist_wg_cap(ctx);
  double best = 1e300;
  int bestK = 1;
  for (int K : {1, 2, 4, 8, 16}) {
    if (K == 16 && !has16) continue;
    const double step = (cell * K + 18.0) * 4.0;
    const double S = std::ceil((double)m / (64.0 * K));
    const double est = pipeline_steps(S, waves, (double)n + 63.0, strip_lag(step)) * step;
    if (est < best) {
      best = est;
      bestK = K;
    }
  }
  return bestK;
}

// Rows per lane for the persistent pipeline over column stripes.  Every rank evaluates the same
// model of the LAST stripe's finishing time and so picks the same K: its first strip can start
// once each stripe to its left has emitted its first border chunk (T_first each), then it works
// through its own strips, ceil(S / resident waves) rounds of one sweep each, plus the ramp of the
// strip chain (consecutive strips run strip_lag() steps apart).  A large K is the cheapest per cell but the
// slowest per step, so the more GPUs share the columns, the smaller K gets.  Step costs measured
// on MI355X (tools/perf_stripe_model.py): (10 K + 18) issue slots of 4 cycles for NW.
// R > 0: K must divide the border chunk into whole strips.
int choose_stripe_K(const sa_ctx* ctx, int kind, int64_t m, int64_t n, int64_t n_total, int64_t R, bool lprof = false) {
  const double cell = kind == KIND_LCS ? (lprof ? 5.0 : 7.0) : (kind == KIND_SW ? 12.0 : 10.0);
  const double step0 = 18.0;
  const double waves = 4.0 * persist_wg_cap(ctx);
  double best = 1e300;
  int bestK = 0;
  for (int K : {1, 2, 4, 8}) {
    if (R > 0 && R % (64 * K) != 0) continue;
    if (forced_K(ctx) && K != forced_K(ctx)) continue;
    const double step = (cell * K + step0) * 4.0;
    const double S = std::ceil((double)m / (64.0 * K));
    const double hops = n > 0 ? std::max(0.0, std::round((double)n_total / (double)n) - 1.0) : 0.0;
    const double lag = strip_lag(step);
    const double first = ((double)n + (R > 0 ? (double)R / (64.0 * K) : 1.0) * lag) * step;
    const double est = hops * first + pipeline_steps(S, waves, (double)n + 63.0, lag) * step;
    if (est < best) {
      best = est;
      bestK = K;
    }
  }
  return bestK ? bestK : 1;
}

void launch_profile(int K, hipStream_t st, const uint8_t* xidx, int m, const int8_t* sub, int alpha, int S,
                    uint32_t* prof) {
  const size_t total = (size_t)S * alpha * 64;
  const int blocks = (int)std::min<size_t>((total + 255) / 256, 65535);
  switch (K) {
    case 1: hipLaunchKernelGGL(build_profile<1>, dim3(blocks), dim3(256), 0, st, xidx, m, sub, alpha, S, prof); break;
    case 2: hipLaunchKernelGGL(build_profile<2>, dim3(blocks), dim3(256), 0, st, xidx, m, sub, alpha, S, prof); break;
    case 4: hipLaunchKernelGGL(build_profile<4>, dim3(blocks), dim3(256), 0, st, xidx, m, sub, alpha, S, prof); break;
    case 16: hipLaunchKernelGGL(build_profile<16>, dim3(blocks), dim3(256), 0, st, xidx, m, sub, alpha, S, prof); break;
    default: hipLaunchKernelGGL(build_profile<8>, dim3(blocks), dim3(256), 0, st, xidx, m, sub, alpha, S, prof); break;
  }
}

// The pair the reference's score function would throw on first, scanning cells in its
// row-major fill order (main.cpp:640 inside :622, :738): returns false if every cell is valid.
bool first_invalid_pair(const uint8_t* x, int64_t m, const uint8_t* y, int64_t n, const uint8_t* map,
                        unsigned char* bx, unsigned char* by) {
  if (m == 0 || n == 0) return false;
  int64_t jbad = -1;
  for (int64_t j = 0; j < n; ++j)
    if (map[y[j]] == 255) {
      jbad = j;
      break;
    }
  for (int64_t i = 0; i < m; ++i) {
    if (map[x[i]] == 255) {
      *bx = x[i];
      *by = y[0];
      return true;
    }
    if (jbad >= 0) {
      *bx = x[i];
      *by = y[jbad];
      return true;
    }
  }
  return false;
}

// Degenerate shapes never reach the GPU: with m==0 or n==0 there is no cell to fill and the
// reference's traceback only walks a border (main.cpp:796-806, 1623-1634).
void border_only_result(int kind, int64_t m, int64_t n, const std::vector<int>& rowb, const std::vector<int>& colb,
                        sa_result* out) {
  out->end_i = (kind == KIND_SW) ? 0 : m;
  out->end_j = (kind == KIND_SW) ? 0 : n;
  out->start_i = out->start_j = 0;
  out->score = 0;
  if (kind == KIND_SW) return;
  if (kind == KIND_NW) out->score = (m == 0) ? rowb[n] : colb[m];
  const int64_t len = m + n;
  if (len > 0) {
    out->ops = (uint8_t*)malloc((size_t)len);
    memset(out->ops, m == 0 ? 1 : 2, (size_t)len);
    out->n_ops = len;
  }
}

struct StripeDesc {  // internal mirror of sa_stripe_desc; all-null = the whole matrix on this GPU
  int64_t n_total = 0, col_begin = 0;
  int is_first = 1, is_last = 1;
  const int32_t* left = nullptr;  // device, (m+1) x {H,E}
  int32_t* right = nullptr;       // device, (m+1) x {H,E}
  int64_t chunk_rows = 0;
  sa_stripe_event_fn on_event = nullptr;
  void* user = nullptr;
  hipStream_t stream_in = nullptr, stream_out = nullptr;
};

int check_scoring(sa_ctx* ctx, int kind, const sa_scoring* sc, int64_t m, int64_t n_total) {
  if (kind == KIND_LCS) {  // registers hold 4 L + 1, hand-off records 30 bits of it
    if (std::min(m, n_total) >= (1 << 27)) return fail(ctx, SA_ERR_ARG, "LCS of sequences beyond 2^27 residues would overflow the 4x-scaled arithmetic");
    return SA_OK;
  }
  if (!sc || !sc->sub || !sc->idx256 || sc->alpha < 1 || sc->alpha > 32) return fail(ctx, SA_ERR_ARG, "bad scoring");
  if (!(std::fabs(sc->gap_open) <= 1e6) || !(std::fabs(sc->gap_extend) <= 1e6)) return fail(ctx, SA_ERR_ARG, "gap penalty out of range");
  int smax = 0;
  for (int i = 0; i < sc->alpha * sc->alpha; ++i) smax = std::max(smax, std::abs((int)sc->sub[i]));
  if (smax > 31) return fail(ctx, SA_ERR_UNSUPPORTED, "substitution scores beyond +-31 do not fit the int8 tagged profile");
  const double worst = ((double)m + n_total + 2) * std::max({std::fabs(sc->gap_open) + 1, std::fabs(sc->gap_extend) + 1, (double)smax});
  if (worst > 1.2e8) return fail(ctx, SA_ERR_ARG, "score range would overflow the 4x-scaled int32 arithmetic");
  return SA_OK;
}

// The batch kernels hold 4x-scaled int32 values for pairs of at most kBatchMaxM x kBatchMaxN cells: the same worst-case
// bound as check_scoring, on those limits (e.g. H[i][0] = 4 go + (i-1) 4 max(go, ge) in nw_batch_kernel).
int check_batch_scoring(sa_ctx* ctx, const sa_scoring* sc) {
  int smax = 0;
  for (int i = 0; i < sc->alpha * sc->alpha; ++i) smax = std::max(smax, std::abs((int)sc->sub[i]));
  const double worst = ((double)kBatchMaxM + kBatchMaxN + 2) * std::max({std::fabs(sc->gap_open) + 1, std::fabs(sc->gap_extend) + 1, (double)smax});
  if (worst > 1.2e8) return fail(ctx, SA_ERR_ARG, "gap penalties of this size would overflow the batch kernels' 4x-scaled int32 arithmetic");
  return SA_OK;
}

// Fill of one stripe of columns (the whole matrix when sd == nullptr): uploads, profile, the
// per-diagonal launch schedule, SW arg-max.  Everything is enqueued on ctx->stream; events
// ev[0] (start), ev[1] (first fill launch), ev[2] (fill done) are recorded.
int fill_impl(sa_ctx* ctx, int kind, const uint8_t* x, int m, const uint8_t* y, int n, const sa_scoring* sc,
              const StripeDesc* sd, uint32_t flags, sa_result* out, const std::vector<int>& rowb, const std::vector<int>& colb) {
  HIPCK(ctx, hipSetDevice(ctx->device));
  hipStream_t st = ctx->stream;
  const bool dump = (flags & SA_FLAG_KEEP_DP) != 0;
  int go = 0, ge = 0, fo = 0, fe = 0, alpha = 1;
  if (kind != KIND_LCS) {
    // register scale of the NW / SW kernels: 4 x score (+ a 2-bit candidate tag, sa_kernels.h)
    go = 4 * (int)std::floor(sc->gap_open);
    ge = 4 * (int)std::floor(sc->gap_extend);
    fo = sc->gap_open != std::floor(sc->gap_open) ? 4 : 0;
    fe = sc->gap_extend != std::floor(sc->gap_extend) ? 4 : 0;
    // SW needs no sign fix: trunc(v + g) = v + floor(g) whenever that is >= 0, and <= 0 otherwise -- and the reference
    // floors E and F at 0 individually (main.cpp:968-970), so a fractional penalty acts exactly like its floor.
    if (kind == KIND_SW) fo = fe = 0;
    alpha = sc->alpha;
  }
  const int fl = ((fo || fe) ? FLAG_FRAC : 0) | (dump ? FLAG_DUMP : 0);
  Tuning tun = choose_tuning(ctx, kind, m, n, sd ? sd->col_begin : 0);
  // column stripes: border chunk k = matrix rows k*R+1 .. min(m,(k+1)*R)
  const bool ev_in = sd && !sd->is_first && sd->on_event;
  const bool ev_out = sd && !sd->is_last && sd->on_event;
  const int64_t R = sd ? std::max<int64_t>(sd->chunk_rows, 1) : 0;
  const int64_t n_chunks = sd ? (m + R - 1) / R : 0;
  // persistent strip pipeline: plain integral configuration; over column stripes the borders
  // must be able to cross while the kernel runs (the caller's two transfer streams)
  const bool striped = sd && !(sd->is_first && sd->is_last);
  const bool persist = ctx->persist && !(fl & FLAG_DUMP) &&
                       (!striped || ((!ev_in || sd->stream_in) && (!ev_out || (sd->stream_out && R % 64 == 0))));
  // LCS over a small alphabet (DNA, proteins: at most 31 distinct bytes in the pair) takes its match term from an
  // LDS profile (FLAG_LPROF); any other input keeps the byte compare
  uint8_t lmap[256];
  bool lprof = false;
  if (kind == KIND_LCS && persist) {
    bool seen[256] = {false};
    for (int i = 0; i < m; ++i) seen[x[i]] = true;
    for (int j = 0; j < n; ++j) seen[y[j]] = true;
    int A = 0;
    for (int b = 0; b < 256; ++b)
      if (seen[b]) lmap[b] = (uint8_t)std::min(A++, 255);
    if (A <= 31) {
      lprof = true;
      alpha = A + 1;  // + the padding symbol
    }
  }
  if (fl & FLAG_DUMP) {
    tun.K = 1, tun.C = 64;
  } else if ((fl & FLAG_FRAC) && !persist) {
    tun.K = 2;
  } else if (persist && striped) {
    tun.K = choose_stripe_K(ctx, kind, m, n, sd->n_total, ev_out ? R : 0, lprof);
    tun.C = 64;
  } else if (persist) {
    tun.K = choose_persist_K(ctx, kind, m, n, lprof, /*has16=*/lprof && alpha <= 16);  // 16 rows per lane: profile within 64 KB of LDS
    tun.C = 64;
  }
  const int K = tun.K, C = tun.C;
  const int PW = (K <= 4) ? 1 : K / 4;
  const int S = (m + 64 * K - 1) / (64 * K);
  const int Tend = ((n + 63 + 63) / 64) * 64;
  const int W = Tend / 16;
  const int NP = Tend + 256;
  const int NF = (kind == KIND_LCS) ? K + 3 : (kind == KIND_SW ? 4 * K + 4 : 2 * K + 4);
  const size_t rows_pad = (size_t)S * 64 * K;
  out->rows_per_lane = K;
  out->tile_steps = C;
  const bool first = !sd || sd->is_first;

  int rc;
  if ((rc = ensure(ctx, ctx->xenc, rows_pad + 64))) return rc;
  if ((rc = ensure(ctx, ctx->yenc, (size_t)Tend + 256))) return rc;
  if (first && (rc = ensure(ctx, ctx->lbuf, ((size_t)m + 1) * 8))) return rc;
  // Codes: 2 bits per cell.  When they do not fit the budget (2 Mb x 2 Mb is 1 TB) the fill keeps the
  // bottom row of EVERY strip instead (checkpoints, 8 B per column per strip) and the traceback
  // recomputes the strips it walks through, a band at a time.  Score-only calls
  // store no codes at all.
  const size_t strip_trace_bytes = (size_t)W * 64 * K * 4;
  const bool no_tb = (flags & SA_FLAG_NO_TRACEBACK) != 0 && !sd && persist;
  size_t budget = ctx->trace_budget;
  if (!budget) {
    size_t free_b = 0, total_b = 0;
    HIPCK(ctx, hipMemGetInfo(&free_b, &total_b));
    budget = (size_t)((double)total_b * 0.88);
  }
  // what the call holds besides the codes: the hand-off ring of a one-slab run (8 GB at 1 Mb x 1 Mb), the query profile
  // and the O(m + n) buffers.  They come out of the same memory, so they count against the same budget.
  const int grid_blocks = std::min((S + kWavesPerBlock - 1) / kWavesPerBlock, persist_wg_cap(ctx));
  // fill_group (sa_group.h): NW / SW over a whole matrix (no border traffic while it runs), profile + rings within 64 KB of LDS
  const bool group = persist && ctx->group && kind != KIND_LCS && !striped && group_lds_bytes(alpha, PW) <= ((size_t)64 << 10);
  const size_t ring_slab = persist ? (size_t)std::min(S + 1, grid_blocks * kWavesPerBlock + 2) + (group ? 1 : 0) : 2;
  const size_t fixed_bytes = 2 * ring_slab * NP * 4 + (size_t)S * alpha * 64 * PW * 4 + (size_t)S * NF * 64 * 4 + ((size_t)m + 1) * 8 +
                             (kind == KIND_SW ? (size_t)2 * m * 4 : 0) + 2 * ((size_t)m + n) + rows_pad + (size_t)Tend + ((size_t)4 << 20);
  const bool ckpt = !no_tb && !sd && ((size_t)S * strip_trace_bytes + fixed_bytes > budget || (ctx->force_band > 0 && persist));
  if (ckpt && !persist)
    return fail(ctx, SA_ERR_NOMEM, "%.1f GB of traceback codes exceed the budget and checkpointed recomputation needs the persistent "
                "schedule (integral gaps, no matrix dump)", (double)S * strip_trace_bytes / 1e9);
  // slots of bottom rows: two for the launch-per-diagonal schedule; the persistent pipeline takes one per strip when
  // it keeps checkpoints or has few strips, else two more than the launch has waves (sa_kernels.h, ho_store)
  const int ring = !persist ? 2 : (ckpt ? S + 1 : std::min(S + 1, grid_blocks * kWavesPerBlock + 2));
  int band = S;
  if (ckpt) {
    const size_t ck_bytes = (size_t)2 * ring * NP * 4 + (fixed_bytes - 2 * ring_slab * NP * 4);  // checkpoint rows + the rest
    if (ck_bytes + strip_trace_bytes > budget) return fail(ctx, SA_ERR_NOMEM, "checkpoint rows (%.1f GB) do not fit the budget", ck_bytes / 1e9);
    band = (int)std::max<size_t>(1, std::min<size_t>((size_t)S, (budget - ck_bytes) / strip_trace_bytes));
    if (ctx->force_band > 0) band = std::min(band, ctx->force_band);
  }
  // The big cached workspaces of an earlier, differently shaped call may be in the way (the code slab
  // of a 1 Mb x 1 Mb pair is 250 GB, the checkpoint rows of a 2 Mb x 2 Mb pair 62 GB): if any of them
  // has to grow, let all of them go first so that the new sizes are what has to fit.
  {
    const size_t need_bot = (size_t)2 * (ring + (group ? 1 : 0)) * NP * 4, need_trace = no_tb ? 0 : (size_t)band * strip_trace_bytes;
    const size_t need_dp = dump ? ((size_t)m + 1) * ((size_t)n + 1) * 4 : 0;
    if (need_bot > ctx->bot.cap || need_trace > ctx->trace.cap || need_dp > ctx->dpbuf.cap) {
      release(ctx->bot);
      release(ctx->trace);
      release(ctx->dpbuf);
    }
  }
  if ((rc = ensure(ctx, ctx->bot, (size_t)2 * (ring + (group ? 1 : 0)) * NP * 4))) return rc;  // fill_group: + one slot nobody writes
  if ((rc = ensure(ctx, ctx->state, (size_t)S * NF * 64 * 4))) return rc;
  if ((rc = ensure(ctx, ctx->small, 256))) return rc;
  const size_t pbuf_ints = (size_t)S + 64 + (size_t)n_chunks;  // [0] ticket, [32] rows_in, [34..35] wait ticks, [64 + S ..] chunk_done
  if (persist && (rc = ensure(ctx, ctx->pbuf, pbuf_ints * 4))) return rc;
  if ((rc = ensure(ctx, ctx->ops, ops_bytes_offset(m, n) + ((size_t)m + n) / 4 + 1024))) return rc;  // one byte per move | packed moves
  if (kind != KIND_LCS || lprof) {
    if ((rc = ensure(ctx, ctx->prof, (size_t)S * alpha * 64 * PW * 4))) return rc;
    if ((rc = ensure(ctx, ctx->sub, 32 * 32))) return rc;
  }
  if (kind == KIND_SW && (rc = ensure(ctx, ctx->rowbest, (size_t)2 * m * 4))) return rc;
  if (dump && (rc = ensure(ctx, ctx->dpbuf, ((size_t)m + 1) * ((size_t)n + 1) * 4))) return rc;
  const size_t trace_bytes = no_tb ? 0 : (size_t)band * strip_trace_bytes;
  if (trace_bytes && (rc = ensure(ctx, ctx->trace, trace_bytes))) return rc;

  // host-side encoding (O(m+n))
  std::vector<uint8_t> xe(rows_pad + 64, 0), ye((size_t)Tend + 256, 0);
  if (lprof) {
    for (int i = 0; i < m; ++i) xe[i] = lmap[x[i]];
    std::fill(ye.begin(), ye.end(), (uint8_t)(alpha - 1));  // padding symbol: matches nothing
    for (int j = 0; j < n; ++j) ye[j] = lmap[y[j]];
  } else if (kind == KIND_LCS) {
    memcpy(xe.data(), x, (size_t)m);
    memcpy(ye.data(), y, (size_t)n);
  } else {
    for (int i = 0; i < m; ++i) xe[i] = sc->idx256[x[i]];
    for (int j = 0; j < n; ++j) ye[j] = sc->idx256[y[j]];
  }
  // E[i][0]: main.cpp:619 (-inf) / :1105 (0).  The kernel multiplies border values by 4, so -inf
  // is handed over as INT_MIN/8.
  const int e_border = (kind == KIND_NW) ? kNeg / 4 : 0;
  // Whole matrix, integral penalties, persistent schedule: both borders are closed forms and are written on the device
  // (rowb / colb are empty then); otherwise the host's chains are uploaded.
  const bool dev_borders = rowb.empty();
  std::vector<int> lbh;
  if (first && !dev_borders) {
    lbh.resize(((size_t)m + 1) * 2);
    for (int i = 0; i <= m; ++i) {
      lbh[2 * (size_t)i] = colb[i];
      lbh[2 * (size_t)i + 1] = e_border;
    }
  }

  HIPCK(ctx, hipEventRecord(ctx->ev[0], st));
  HIPCK(ctx, hipMemcpyAsync(ctx->xenc.p, xe.data(), xe.size(), hipMemcpyHostToDevice, st));
  HIPCK(ctx, hipMemcpyAsync(ctx->yenc.p, ye.data(), ye.size(), hipMemcpyHostToDevice, st));
  const int2* lb;
  const int gi_open = sc ? (int)sc->gap_open : 0, gi_ext = sc ? (int)sc->gap_extend : 0;
  if (first && dev_borders) {
    hipLaunchKernelGGL(init_left_border, dim3(std::min((m + 256) / 256, 2048)), dim3(256), 0, st, (int2*)ctx->lbuf.p, m,
                       kind == KIND_NW ? gi_open : 0, kind == KIND_NW ? std::max(gi_open, gi_ext) : 0, e_border);
    lb = (const int2*)ctx->lbuf.p;
  } else if (first) {
    HIPCK(ctx, hipMemcpyAsync(ctx->lbuf.p, lbh.data(), lbh.size() * 4, hipMemcpyHostToDevice, st));
    lb = (const int2*)ctx->lbuf.p;
  } else {
    // entry 0 of the caller's left border is the matrix' row 0 at this stripe's left edge
    const int corner[2] = {rowb[0], e_border};
    HIPCK(ctx, hipMemcpyAsync((void*)sd->left, corner, 8, hipMemcpyHostToDevice, st));
    lb = (const int2*)sd->left;
  }
  int* botH = (int*)ctx->bot.p;
  int* botF = botH + (size_t)ring * NP;
  int2* bot2 = (int2*)ctx->bot.p;  // the persistent pipeline's view of the same memory: [ring][NP] records {H | generation, F}
  // strip 0 reads the last slot: row 0 of the matrix, H[0][j] at p = j + 63.  Every other slot is
  // written by its strip before the strip below reads it.
  int* row0H = botH + (size_t)(ring - 1) * NP;
  int* row0F = botF + (size_t)(ring - 1) * NP;
  std::vector<int> rowb4;
  if (kind == KIND_NW && !dev_borders) {
    rowb4.resize((size_t)n);
    for (int j = 0; j < n; ++j) rowb4[j] = 4 * rowb[j + 1];
  }
  if (persist) {
    // Every record starts with generation 3 = "not yet written" (strips write generations 0..2); matrix row 0 goes
    // into slot 0 with generation 0, as "strip -1".
    HIPCK(ctx, hipMemsetD32Async((hipDeviceptr_t)bot2, (int)0xC0000000u, (size_t)2 * ring * NP, st));
    const int* d_h4 = nullptr;
    if (kind == KIND_NW && !dev_borders) {
      if ((rc = ensure(ctx, ctx->colb, (size_t)n * 4))) return rc;
      HIPCK(ctx, hipMemcpyAsync(ctx->colb.p, rowb4.data(), (size_t)n * 4, hipMemcpyHostToDevice, st));
      d_h4 = (const int*)ctx->colb.p;
    }
    // LCS registers hold 4 * L + 1 (sa_kernels.h): row 0 is all ones; SW: zeros; F[0][j] = -inf (NW) / 0
    hipLaunchKernelGGL(init_row0_records, dim3(std::min((n + 255) / 256, 2048)), dim3(256), 0, st, bot2, d_h4, n,
                       kind == KIND_LCS ? 1 : (kind == KIND_NW && dev_borders ? 4 * gi_open : 0), kind == KIND_NW ? kNeg : 0,
                       kind == KIND_NW && dev_borders ? 4 * gi_ext : 0);
  } else {
    if (kind == KIND_LCS) {
      if (ring == 2)
        HIPCK(ctx, hipMemsetD32Async((hipDeviceptr_t)botH, 1, (size_t)2 * NP, st));
      else
        HIPCK(ctx, hipMemsetD32Async((hipDeviceptr_t)row0H, 1, (size_t)NP, st));
    } else if (ring == 2) {
      HIPCK(ctx, hipMemsetAsync(botH, 0, (size_t)2 * NP * 4, st));
      HIPCK(ctx, hipMemsetD32Async((hipDeviceptr_t)botF, kind == KIND_NW ? kNeg : 0, (size_t)2 * NP, st));
    } else {
      HIPCK(ctx, hipMemsetAsync(row0H, 0, (size_t)NP * 4, st));
      HIPCK(ctx, hipMemsetD32Async((hipDeviceptr_t)row0F, kind == KIND_NW ? kNeg : 0, (size_t)NP, st));
    }
    if (kind == KIND_NW) HIPCK(ctx, hipMemcpyAsync(row0H + 64, rowb4.data(), (size_t)n * 4, hipMemcpyHostToDevice, st));
  }
  HIPCK(ctx, hipMemsetAsync(ctx->small.p, 0, 256, st));
  int* d_score = (int*)ctx->small.p;                                           // [0]   NW/LCS score / SW best
  unsigned long long* d_pos = (unsigned long long*)((char*)ctx->small.p + 8);  // [8]   SW end cell
  if (kind == KIND_SW) HIPCK(ctx, hipMemsetAsync(d_pos, 0xff, 8, st));
  int8_t lsub[32 * 32];
  if (lprof) {
    // build_profile stores 4 v + 2: v = 1 gives the match increment 6 (4 (L + 1) + 3 from 4 L + 1), v = -16 gives -62
    for (int i = 0; i < alpha; ++i)
      for (int j = 0; j < alpha; ++j) lsub[i * alpha + j] = (i == j && i < alpha - 1) ? 1 : -16;
    HIPCK(ctx, hipMemcpyAsync(ctx->sub.p, lsub, (size_t)alpha * alpha, hipMemcpyHostToDevice, st));
    HIPCK(ctx, hipStreamSynchronize(st));  // lsub is on the stack
    launch_profile(K, st, (const uint8_t*)ctx->xenc.p, m, (const int8_t*)ctx->sub.p, alpha, S, (uint32_t*)ctx->prof.p);
  } else if (kind != KIND_LCS) {
    HIPCK(ctx, hipMemcpyAsync(ctx->sub.p, sc->sub, (size_t)alpha * alpha, hipMemcpyHostToDevice, st));
    launch_profile(K, st, (const uint8_t*)ctx->xenc.p, m, (const int8_t*)ctx->sub.p, alpha, S, (uint32_t*)ctx->prof.p);
  }

  FillArgs a{};
  a.m = m;
  a.n = n;
  a.S = S;
  a.C = C;
  a.W = W;
  a.go = go;
  a.ge = ge;
  a.fo = fo;
  a.fe = fe;
  a.alpha = alpha;
  a.NP = NP;
  a.yenc = (const uint8_t*)ctx->yenc.p;
  a.prof = (const uint32_t*)ctx->prof.p;
  a.xpad = (const uint8_t*)ctx->xenc.p;
  a.lb = lb;
  a.rb = (sd && !sd->is_last) ? (int2*)sd->right : nullptr;
  a.dp = dump ? (int*)ctx->dpbuf.p : nullptr;
  a.botH = botH;
  a.botF = botF;
  a.bot2 = bot2;
  a.state = (int*)ctx->state.p;
  a.trace = (ckpt || no_tb) ? nullptr : (uint32_t*)ctx->trace.p;  // checkpoint pass / score only: no codes
  a.trace_s0 = 0;
  a.ring = ring;
  a.s_first = 0;
  a.s_last = S - 1;
  a.rowbest = (int*)ctx->rowbest.p;
  a.rowbestc = a.rowbest ? a.rowbest + m : nullptr;
  a.score_out = d_score;
  const size_t lds = (kind == KIND_LCS && !lprof) ? 0 : (size_t)alpha * 64 * PW * 4 * kWavesPerBlock;
  auto f_lo = [&](int s) { return (long long)s + (64LL * s) / C; };
  auto f_hi = [&](int s) { return (long long)s + (64LL * s + n + 62) / C; };
  const long long dmax = f_hi(S - 1);
  int64_t chunks_in = 0, chunks_out = 0;
  int slo = 0, shi = -1, launches = 0;
  if (persist) {
    HIPCK(ctx, hipMemsetAsync(ctx->pbuf.p, 0, pbuf_ints * 4, st));
    a.ticket = (int*)ctx->pbuf.p;
#ifdef SA_DEBUG_HANDOFF
    {
      static void* dbgbuf = nullptr;
      static size_t dbgcap = 0;
      if (dbgcap < (size_t)S * 32) {
        if (dbgbuf) (void)hipFree(dbgbuf);
        HIPCK(ctx, hipMalloc(&dbgbuf, (size_t)S * 32));
        dbgcap = (size_t)S * 32;
      }
      HIPCK(ctx, hipMemsetAsync(dbgbuf, 0, (size_t)S * 32, st));
      a.dbg = (long long*)dbgbuf;
      static void* blkbuf = nullptr;
      static size_t blkcap = 0;
      const int NBd = (n + 126) / 64;
      if (getenv("SA_DEBUG_BLOCKS_FILE")) {
        if (blkcap < (size_t)NBd * 64) {
          if (blkbuf) (void)hipFree(blkbuf);
          HIPCK(ctx, hipMalloc(&blkbuf, (size_t)NBd * 64));
          blkcap = (size_t)NBd * 64;
        }
        HIPCK(ctx, hipMemsetAsync(blkbuf, 0, (size_t)NBd * 64, st));
        a.dbgblk = (long long*)blkbuf;
        a.dbgblk_stride = NBd;
      }
    }
#endif
    a.err = (int*)ctx->small.p + 40;  // byte 160 of the small block, read back with the score
    a.spin_limit = 1 << 22;
    if (striped && ev_in) {
      a.rows_in = (int*)ctx->pbuf.p + 32;
      a.wait_ticks = (unsigned long long*)((int*)ctx->pbuf.p + 34);
    }
    if (striped && ev_out) {
      a.chunk_done = (int*)ctx->pbuf.p + 64 + S;
      a.strips_per_chunk = (int)(R / (64 * K));
    }
    if (a.rows_in || a.chunk_done) {
      // the transfer streams may not touch the flags before they are reset
      HIPCK(ctx, hipEventRecord(ctx->ev[4], st));
      if (a.rows_in) HIPCK(ctx, hipStreamWaitEvent(sd->stream_in, ctx->ev[4], 0));
      if (a.chunk_done) HIPCK(ctx, hipStreamWaitEvent(sd->stream_out, ctx->ev[4], 0));
    }
  }
  HIPCK(ctx, hipEventRecord(ctx->ev[1], st));
  if (persist) {
    const dim3 grid(grid_blocks);
    if (!handoff_ring_ok(a.ring, S, grid_blocks * kWavesPerBlock))
      return fail(ctx, SA_ERR_HIP, "internal: %d record slots for %d strips on %d waves", a.ring, S, grid_blocks * kWavesPerBlock);
    if (group)
      launch_group_kind(kind, K, grid, group_lds_bytes(alpha, PW), st, a, ckpt || no_tb, kind == KIND_NW && (fl & FLAG_FRAC) != 0);
    else
      launch_persistent_kind(kind, K, grid, lds, st, a, ckpt || no_tb, lprof, kind == KIND_NW && (fl & FLAG_FRAC) != 0);
    launches = 1;
    // border hand-offs, all enqueued now and resolved on the device while the fill runs
    for (int64_t k = 0; k < n_chunks && (a.rows_in || a.chunk_done); ++k) {
      if (a.rows_in) {
        if (sd->on_event(sd->user, SA_EV_NEED_LEFT, k)) return fail(ctx, SA_ERR_ARG, "stripe callback failed (need chunk %lld)", (long long)k);
        hipLaunchKernelGGL(stripe_signal, dim3(1), dim3(64), 0, sd->stream_in, (int*)a.rows_in, (int)std::min<int64_t>(m, (k + 1) * R));
      }
      if (a.chunk_done) {
        const int target = (int)(std::min<int64_t>(S, (k + 1) * a.strips_per_chunk) - k * a.strips_per_chunk);
        hipLaunchKernelGGL(stripe_wait, dim3(1), dim3(64), 0, sd->stream_out, a.chunk_done + k, target, a.err, 1 << 22);
        if (sd->on_event(sd->user, SA_EV_RIGHT_READY, k)) return fail(ctx, SA_ERR_ARG, "stripe callback failed (ready chunk %lld)", (long long)k);
      }
    }
  }
  for (long long d = 0; !persist && d <= dmax; ++d) {
    while (shi + 1 < S && f_lo(shi + 1) <= d) ++shi;
    while (slo < S && f_hi(slo) < d) ++slo;
    if (slo > shi) continue;
    if (ev_in) {  // rows 1 .. (shi+1)*64K must have arrived before this launch
      const int64_t need_rows = std::min<int64_t>(m, (int64_t)(shi + 1) * 64 * K);
      while (chunks_in < n_chunks && chunks_in * R < need_rows) {
        if (sd->on_event(sd->user, SA_EV_NEED_LEFT, chunks_in)) return fail(ctx, SA_ERR_ARG, "stripe callback failed (need chunk %lld)", (long long)chunks_in);
        ++chunks_in;
      }
    }
    a.d = (int)d;
    a.s_lo = slo;
    a.s_hi = shi;
    const dim3 grid((shi - slo + kWavesPerBlock) / kWavesPerBlock);
    launch_fill_kind(kind, K, kind == KIND_LCS ? (fl & FLAG_DUMP) : fl, grid, lds, st, a);
    ++launches;
    if (ev_out) {  // strips with f_hi(s) <= d are complete after this launch
      int done = slo;
      while (done < S && f_hi(done) <= d) ++done;
      const int64_t rows_done = std::min<int64_t>(m, (int64_t)done * 64 * K);
      while (chunks_out < n_chunks && std::min<int64_t>(m, (chunks_out + 1) * R) <= rows_done) {
        if (sd->on_event(sd->user, SA_EV_RIGHT_READY, chunks_out)) return fail(ctx, SA_ERR_ARG, "stripe callback failed (ready chunk %lld)", (long long)chunks_out);
        ++chunks_out;
      }
    }
  }
  HIPCK(ctx, hipGetLastError());
#ifdef SA_DEBUG_HANDOFF
  if (persist && a.dbg) {
    std::vector<long long> h((size_t)4 * S);
    HIPCK(ctx, hipStreamSynchronize(st));
    if (a.dbgblk) {  // per-block clocks of eight strips: raw int64 [8][stride], little endian, behind a 2-int header
      std::vector<long long> hb((size_t)8 * a.dbgblk_stride);
      HIPCK(ctx, hipMemcpy(hb.data(), a.dbgblk, hb.size() * 8, hipMemcpyDeviceToHost));
      if (FILE* f = fopen(getenv("SA_DEBUG_BLOCKS_FILE"), "wb")) {
        const int hdr[2] = {a.dbgblk_stride, S};
        fwrite(hdr, 4, 2, f);
        fwrite(hb.data(), 8, hb.size(), f);
        fclose(f);
      }
    }
    HIPCK(ctx, hipMemcpy(h.data(), a.dbg, h.size() * 8, hipMemcpyDeviceToHost));
    double lag = 0, dur = 0, ep = 0, rr = 0;
    const int lim = std::min(S, grid_blocks * kWavesPerBlock);
    for (int q = 1; q < lim; ++q) lag += (double)(h[4 * q] - h[4 * (q - 1)]);
    {  // fill_group: hops inside a workgroup and hops between workgroups apart
      double in = 0, out = 0;
      int nin = 0, nout = 0;
      for (int q = 1; q < lim; ++q) (q % kWavesPerBlock ? (++nin, in) : (++nout, out)) += (double)(h[4 * q + 1] - h[4 * (q - 1) + 1]);
      fprintf(stderr, "[handoff] group=%d: end-to-end lag of strips 4g+1..3 %.2f us, of strips 4g %.2f us\n", (int)group, in / std::max(1, nin) / 100.0, out / std::max(1, nout) / 100.0);
    }
    for (int q = 0; q < S; ++q) dur += (double)(h[4 * q + 1] - h[4 * q]), ep += (double)h[4 * q + 2], rr += (double)h[4 * q + 3];
    fprintf(stderr, "[handoff] K=%d S=%d n=%d: mean start-to-start lag of the first %d strips %.2f us, mean strip duration %.1f us (%.1f ns/step), "
            "re-read episodes per strip %.1f, re-reads per strip %.1f; span first start -> last end %.1f us; strip 0 alone: %.1f ns/step\n", K, S, n, lim, lag / std::max(1, lim - 1) / 100.0,
            dur / S / 100.0, dur / S * 10.0 / (n + 63), ep / S, rr / S, (double)(h[4 * (S - 1) + 1] - h[0]) / 100.0, (double)(h[1] - h[0]) * 10.0 / (n + 63));
  }
#endif
  out->n_launches = launches;
  if (kind == KIND_SW) {
    const int rb = std::min((m + 255) / 256, 1024);
    hipLaunchKernelGGL(sw_reduce_max, dim3(rb), dim3(256), 0, st, a.rowbest, m, d_score);
    hipLaunchKernelGGL(sw_reduce_pos, dim3(rb), dim3(256), 0, st, a.rowbest, a.rowbestc, m, d_score, d_pos);
  }
  HIPCK(ctx, hipEventRecord(ctx->ev[2], st));
  ctx->last.valid = 1;
  ctx->last.kind = kind;
  ctx->last.m = m;
  ctx->last.n = n;
  ctx->last.K = K;
  ctx->last.W = W;
  ctx->last.first_stripe = first ? 1 : 0;
  ctx->last.ckpt = ckpt ? 1 : 0;
  ctx->last.band = band;
  ctx->last.S = S;
  ctx->last.args = a;
  ctx->last.lds = lds;
  ctx->last.lprof = lprof;
  ctx->last.bits = 0;
  // cells this fill computed, by the work done per cell (with / without direction codes)
  (a.trace ? out->cells_trace : out->cells_notrace) = (int64_t)m * n;
  return SA_OK;
}

// LCS of one whole pair as bit-vector arithmetic (sa_lcsbits.hip): 32 rows per lane and one bit of traceback record
// per cell.  Applies when the pair holds at most 255 distinct bytes (DNA, proteins, text), the record (m n / 8 bytes)
// fits the budget or can be recomputed in bands from kept carry records, and nothing asks for the cell-by-cell kernels
// (matrix dump, forced rows per lane).  *used = false: the caller takes fill_impl instead.  Events as in fill_impl.
int lcs_bits_fill(sa_ctx* ctx, const uint8_t* x, int m, const uint8_t* y, int n, uint32_t flags, sa_result* out, bool* used,
                  const StripeDesc* sd = nullptr) {
  *used = false;
  if (!ctx->lcs_bits || !ctx->persist || ctx->force_K || (flags & SA_FLAG_KEEP_DP)) return SA_OK;
  if (ctx->force_band > 0 && sd) return SA_OK;
  // Column stripes (round 3): the same kernel, the stripe's left border column as its first V and its last V as the right
  // border; the borders travel as (H, E) rows as before and are converted on the transfer streams.  Needs the overlapped
  // schedule (the kernel is persistent) and border chunks of whole strips.
  const bool striped = sd && !(sd->is_first && sd->is_last);
  const bool ev_in = sd && !sd->is_first, ev_out = sd && !sd->is_last;
  const int64_t R = sd ? std::max<int64_t>(sd->chunk_rows, 1) : 0;
  if (striped && ((ev_in && !(sd->on_event && sd->stream_in)) || (ev_out && !(sd->on_event && sd->stream_out)) || R % kBitsStripRows != 0)) return SA_OK;
  const int64_t n_chunks = striped ? (m + R - 1) / R : 0;
  uint8_t lmap[256];
  bool seen[256] = {false};
  for (int i = 0; i < m; ++i) seen[x[i]] = true;
  for (int j = 0; j < n; ++j) seen[y[j]] = true;
  int A = 0;
  for (int b = 0; b < 256; ++b)
    if (seen[b]) lmap[b] = (uint8_t)std::min(A++, 255);
  if (A + 1 > kBitsMaxAlpha) return SA_OK;
  const int alpha = A + 1;  // + the padding symbol (empty masks)
  HIPCK(ctx, hipSetDevice(ctx->device));
  hipStream_t st = ctx->stream;
  const int S = (m + kBitsStripRows - 1) / kBitsStripRows;
  const int Tend = bits_steps(n);
  const int NP = bits_slot_len(n);
  const bool no_tb = (flags & SA_FLAG_NO_TRACEBACK) != 0;
  const size_t trace_bytes = no_tb ? 0 : (size_t)S * ((size_t)Tend * 64 + kBitsStripPad) * 4;
  size_t budget = ctx->trace_budget;
  if (!budget) {
    size_t free_b = 0, total_b = 0;
    HIPCK(ctx, hipMemGetInfo(&free_b, &total_b));
    budget = (size_t)((double)total_b * 0.88);
  }
  if ((size_t)Tend * 64 + kBitsStripPad >= ((size_t)1 << 32)) return SA_OK;  // the walker indexes a strip's record with 32 bits
  // Waves per workgroup: as few as still give every strip a wave in one round.  A CU sustains ~25 GB/s of record writes
  // (so does a plain fill: 6.9 TB/s over 256 CUs), and 489 strips as 123 workgroups of four waves use 123 CUs: 2.7 TB/s,
  // 47 ms at 1 Mb x 1 Mb; as 245 workgroups of two waves they use 245.
  const int cap = persist_wg_cap(ctx);
  // A wave's profile is 256 B of LDS per symbol (DNA: 1.3 KB; every byte value but one: 64 KB), a workgroup's waves share the CU's 160 KB.
  const int wpb_fit = std::max(1, std::min(kWavesPerBlock, (int)(kBitsLdsBytes / ((size_t)alpha * 256))));
  const int wpb = std::min(wpb_fit == 3 ? 2 : wpb_fit, ctx->bits_wpb ? ctx->bits_wpb : (S <= 2 * cap ? 2 : kWavesPerBlock));  // (single-wave workgroups are not guaranteed a SIMD of their own)
  const int grid_blocks = std::min((S + wpb - 1) / wpb, cap);
  const size_t rows_pad = (size_t)S * kBitsStripRows;
  const size_t strip_rec_bytes = ((size_t)Tend * 64 + kBitsStripPad) * 4;
  const size_t other_bytes = (size_t)S * alpha * 64 * 4 + 2 * ((size_t)m + n) + rows_pad + (size_t)Tend + ((size_t)4 << 20);
  int ring = std::min(S + 1, grid_blocks * wpb + 2);
  // Round 3: a record beyond the budget (2 Mb x 2 Mb: 500 GB) no longer sends the pair to the cell-by-cell kernels.  As
  // there (fill_impl), the fill keeps the carry records of EVERY strip instead (ring = S + 1 slots, 8 B per column per
  // strip: 16 GB at 2 Mb x 2 Mb) and writes no record; the traceback recomputes `band` strips at a time WITH their record
  // (every strip starts from the all-ones column 0, so the kept carries are all a band needs) and walks them.
  const bool ckpt = !no_tb && !sd && (trace_bytes + (size_t)2 * ring * NP * 4 + other_bytes > budget || ctx->force_band > 0);
  int band = S;
  size_t trace_alloc = trace_bytes;
  if (ckpt) {
    ring = S + 1;
    const size_t ck_bytes = (size_t)2 * ring * NP * 4 + other_bytes;
    if (ck_bytes + strip_rec_bytes > budget) return SA_OK;  // not even one strip's record beside the carries: the cell kernels' problem
    band = (int)std::max<size_t>(1, std::min<size_t>((size_t)S, (budget - ck_bytes) / strip_rec_bytes));
    if (ctx->force_band > 0) band = std::min(band, ctx->force_band);
    trace_alloc = (size_t)band * strip_rec_bytes;
  } else if (trace_bytes + (size_t)2 * ring * NP * 4 + other_bytes > budget) {
    return SA_OK;  // (score-only calls and stripes never get here: no record / a stripe's record is its own problem)
  }
  int rc;
  {
    const size_t need_bot = (size_t)2 * ring * NP * 4;
    if (need_bot > ctx->bot.cap || trace_alloc > ctx->trace.cap) {
      release(ctx->bot);
      release(ctx->trace);
      release(ctx->dpbuf);
    }
  }
  if ((rc = ensure(ctx, ctx->xenc, rows_pad + 64))) return rc;
  if ((rc = ensure(ctx, ctx->yenc, (size_t)Tend + 256))) return rc;
  if ((rc = ensure(ctx, ctx->bot, (size_t)2 * ring * NP * 4))) return rc;
  if ((rc = ensure(ctx, ctx->small, 256))) return rc;
  const size_t pbuf_ints = (size_t)S + 64 + (size_t)n_chunks;  // [0] ticket, [32] rows_in, [34..35] wait ticks, [64 + S ..] chunk_done (as fill_impl)
  if ((rc = ensure(ctx, ctx->pbuf, pbuf_ints * 4))) return rc;
  if (ev_in && (rc = ensure(ctx, ctx->vlb, (size_t)S * 64 * 4))) return rc;
  if (ev_out && (rc = ensure(ctx, ctx->vrb, (size_t)S * 64 * 4))) return rc;
  if ((rc = ensure(ctx, ctx->ops, ops_bytes_offset(m, n) + ((size_t)m + n) / 4 + 1024))) return rc;
  if ((rc = ensure(ctx, ctx->prof, (size_t)S * alpha * 64 * 4))) return rc;
  if (trace_alloc && (rc = ensure(ctx, ctx->trace, trace_alloc))) return rc;
  std::vector<uint8_t> xe(rows_pad + 64, 0), ye((size_t)Tend + 256, (uint8_t)(alpha - 1));
  for (int i = 0; i < m; ++i) xe[i] = lmap[x[i]];
  for (int j = 0; j < n; ++j) ye[j] = lmap[y[j]];

  HIPCK(ctx, hipEventRecord(ctx->ev[0], st));
  HIPCK(ctx, hipMemcpyAsync(ctx->xenc.p, xe.data(), xe.size(), hipMemcpyHostToDevice, st));
  HIPCK(ctx, hipMemcpyAsync(ctx->yenc.p, ye.data(), ye.size(), hipMemcpyHostToDevice, st));
  int2* bot2 = (int2*)ctx->bot.p;
  // records: generation 3 = "not yet written"; the carries into matrix row 1 are zero ("strip -1", slot 0, generation 0)
  HIPCK(ctx, hipMemsetD32Async((hipDeviceptr_t)bot2, (int)0xC0000000u, (size_t)2 * ring * NP, st));
  // (init_row0_records writes column c at index c + 64)
  hipLaunchKernelGGL(init_row0_records, dim3(std::min((n + 255) / 256, 2048)), dim3(256), 0, st, bot2 + (kBitsRecord0 - 64), (const int*)nullptr, n, 0, 0);
  HIPCK(ctx, hipMemsetAsync(ctx->small.p, 0, 256, st));
  HIPCK(ctx, hipMemsetAsync(ctx->pbuf.p, 0, pbuf_ints * 4, st));
  if (ev_in) {
    HIPCK(ctx, hipMemsetAsync(ctx->vlb.p, 0xff, (size_t)S * 64 * 4, st));  // rows beyond m: no vertical step
    const int corner[2] = {0, 0};  // entry 0 of the caller's left border is the matrix' row 0 at this stripe's left edge
    HIPCK(ctx, hipMemcpyAsync((void*)sd->left, corner, 8, hipMemcpyHostToDevice, st));
  }
  bits_launch_profile(st, (const uint8_t*)ctx->xenc.p, m, alpha, S, (uint32_t*)ctx->prof.p);
  FillArgs a{};
  a.m = m;
  a.n = n;
  a.S = S;
  a.C = 64;
  a.W = Tend;
  a.alpha = alpha;
  a.NP = NP;
  a.yenc = (const uint8_t*)ctx->yenc.p;
  a.prof = (const uint32_t*)ctx->prof.p;
  a.xpad = (const uint8_t*)ctx->xenc.p;
  a.bot2 = bot2;
  a.ring = ring;
  a.trace = (no_tb || ckpt) ? nullptr : (uint32_t*)ctx->trace.p;  // checkpoint pass: no record
  a.trace_s0 = 0;
  a.score_out = (int*)ctx->small.p;
  a.ticket = (int*)ctx->pbuf.p;
  a.s_first = 0;
  a.s_last = S - 1;
  a.err = (int*)ctx->small.p + 40;
  a.spin_limit = 1 << 22;
  if (ev_in) {
    a.vleft = (const uint32_t*)ctx->vlb.p;
    a.rows_in = (int*)ctx->pbuf.p + 32;
    a.wait_ticks = (unsigned long long*)((int*)ctx->pbuf.p + 34);
  }
  if (ev_out) {
    a.vright = (uint32_t*)ctx->vrb.p;
    a.chunk_done = (int*)ctx->pbuf.p + 64 + S;
    a.strips_per_chunk = (int)(R / kBitsStripRows);
  }
  if (a.rows_in || a.chunk_done) {  // the transfer streams may not touch the flags before they are reset
    HIPCK(ctx, hipEventRecord(ctx->ev[4], st));
    if (a.rows_in) HIPCK(ctx, hipStreamWaitEvent(sd->stream_in, ctx->ev[4], 0));
    if (a.chunk_done) HIPCK(ctx, hipStreamWaitEvent(sd->stream_out, ctx->ev[4], 0));
  }
  HIPCK(ctx, hipEventRecord(ctx->ev[1], st));
  if (!handoff_ring_ok(a.ring, S, grid_blocks * wpb))
    return fail(ctx, SA_ERR_HIP, "internal: %d record slots for %d strips on %d waves", a.ring, S, grid_blocks * wpb);
  bits_launch_fill(st, grid_blocks, wpb, a);
  // border hand-offs, all enqueued now and resolved on the device while the fill runs (as in fill_impl), with the
  // conversion between border rows and bits on the transfer streams
  for (int64_t k = 0; k < n_chunks; ++k) {
    const int r_lo = (int)(k * R), r_hi = (int)std::min<int64_t>(m, (k + 1) * R);
    if (a.rows_in) {
      if (sd->on_event(sd->user, SA_EV_NEED_LEFT, k)) return fail(ctx, SA_ERR_ARG, "stripe callback failed (need chunk %lld)", (long long)k);
      bits_launch_border_to_bits(sd->stream_in, (const int2*)sd->left, m, r_lo, r_hi, (uint32_t*)ctx->vlb.p);
      hipLaunchKernelGGL(stripe_signal, dim3(1), dim3(64), 0, sd->stream_in, (int*)a.rows_in, r_hi);
    }
    if (a.chunk_done) {
      const int target = (int)(std::min<int64_t>(S, (k + 1) * a.strips_per_chunk) - k * a.strips_per_chunk);
      hipLaunchKernelGGL(stripe_wait, dim3(1), dim3(64), 0, sd->stream_out, a.chunk_done + k, target, a.err, 1 << 22);
      bits_launch_bits_to_border(sd->stream_out, (const uint32_t*)ctx->vrb.p, m, r_lo, r_hi, (int2*)sd->right);
      if (sd->on_event(sd->user, SA_EV_RIGHT_READY, k)) return fail(ctx, SA_ERR_ARG, "stripe callback failed (ready chunk %lld)", (long long)k);
    }
  }
  HIPCK(ctx, hipGetLastError());
  HIPCK(ctx, hipEventRecord(ctx->ev[2], st));
  out->n_launches = 1;
  out->rows_per_lane = kBitsRowsPerLane;
  out->tile_steps = 64;
  ctx->last.valid = 1;
  ctx->last.kind = KIND_LCS;
  ctx->last.m = m;
  ctx->last.n = n;
  ctx->last.K = kBitsRowsPerLane;
  ctx->last.W = Tend;
  ctx->last.first_stripe = (!sd || sd->is_first) ? 1 : 0;
  ctx->last.ckpt = ckpt ? 1 : 0;
  ctx->last.band = band;
  ctx->last.S = S;
  ctx->last.args = a;
  ctx->last.lds = (size_t)wpb;  // bit-vector LCS: the waves per workgroup of the launch (the band launches of a checkpointed run repeat it)
  ctx->last.lprof = 0;
  ctx->last.bits = 1;
  (a.trace ? out->cells_trace : out->cells_notrace) = (int64_t)m * n;
  *used = true;
  return SA_OK;
}

// the walker over 2-bit codes: NW / SW tags or the LCS cell kernels' pointers
static void launch_traceback_walk(hipStream_t st, const TraceArgs& t) {
  if (t.kind == KIND_NW)
    hipLaunchKernelGGL((traceback_walk<false, 1>), dim3(1), dim3(64), 0, st, t);
  else if (t.kind == KIND_SW)
    hipLaunchKernelGGL((traceback_walk<false, 2>), dim3(1), dim3(64), 0, st, t);
  else
    hipLaunchKernelGGL((traceback_walk<false, 0>), dim3(1), dim3(64), 0, st, t);
}

// Traceback after a checkpointed fill: from the start cell upwards, recompute the codes of `band`
// strips at a time (the persistent kernel restricted to those strips, reading the kept bottom row
// of the strip above the band) and let the walker cross the band; it hands (i, j) over at the
// band's top row.  Same op list as the one-slab traceback; costs a second fill of the rows walked.
int traceback_ckpt(sa_ctx* ctx, int start_i, int start_j, sa_result* out) {
  hipStream_t st = ctx->stream;
  const int kind = ctx->last.kind, m = ctx->last.m, n = ctx->last.n, K = ctx->last.K, S = ctx->last.S, B = ctx->last.band;
  struct {
    int score;
    int pad;
    unsigned long long pos;
    int rest[40];
  } h_sc{};
  HIPCK(ctx, hipMemcpyAsync(&h_sc, ctx->small.p, sizeof h_sc, hipMemcpyDeviceToHost, st));
  HIPCK(ctx, hipStreamSynchronize(st));
  if (h_sc.rest[36] != 0) return fail(ctx, SA_ERR_HIP, "persistent fill: a bounded wait on the previous strip expired (results discarded)");
  int i = m, j = n;
  if (start_i >= 0) {
    i = start_i, j = start_j;
  } else if (kind == KIND_SW) {
    i = h_sc.score > 0 ? (int)(h_sc.pos >> 32) : 0;
    j = h_sc.score > 0 ? (int)(h_sc.pos & 0xffffffffu) : 0;
  }
  out->score = h_sc.score;
  out->end_i = i;
  out->end_j = j;
  long long nops = 0;
  FillArgs a = ctx->last.args;
  a.trace = (uint32_t*)ctx->trace.p;
  long long* d_hdr = (long long*)((char*)ctx->small.p + 64);
  int* d_err = (int*)ctx->small.p + 40;
  while (i > 0 && j > 0) {
    const int s_hi = (i - 1) / (64 * K);
    const int s_lo = std::max(0, s_hi - B + 1);
    HIPCK(ctx, hipMemsetAsync(ctx->pbuf.p, 0, ((size_t)S + 64) * 4, st));
    a.s_first = s_lo;
    a.s_last = s_hi;
    a.trace_s0 = s_lo;
    const dim3 grid(std::min((s_hi - s_lo + 1 + kWavesPerBlock - 1) / kWavesPerBlock, persist_wg_cap(ctx)));
    HIPCK(ctx, hipEventRecord(ctx->ev[5], st));
    if (ctx->last.bits) {  // the bit-vector LCS: the band's strips again, this time with their record; the score stays the first pass's
      const int wpb = (int)ctx->last.lds;
      a.score_out = (int*)ctx->small.p + 48;
      bits_launch_fill(st, std::min((s_hi - s_lo + 1 + wpb - 1) / wpb, persist_wg_cap(ctx)), wpb, a);
    } else {
      launch_persistent_kind(kind, K, grid, ctx->last.lds, st, a, false, kind == KIND_LCS && ctx->last.lprof != 0, kind == KIND_NW && (a.fo != 0 || a.fe != 0));
    }
    HIPCK(ctx, hipEventRecord(ctx->ev[6], st));
    ++out->n_launches;
    out->cells_trace += (int64_t)(std::min<int64_t>(m, (int64_t)(s_hi + 1) * 64 * K) - (int64_t)s_lo * 64 * K) * n;
    TraceArgs t{};
    t.kind = kind;
    t.m = m;
    t.n = n;
    t.K = K;
    t.W = ctx->last.W;
    t.start_i = i;
    t.start_j = j;
    t.stop_at_col0 = 0;
    t.i_min = s_lo * 64 * K;
    t.nops0 = nops;
    t.trace = (const uint32_t*)ctx->trace.p;
    t.sw_best = (const int*)ctx->small.p;
    t.sw_pos = (const unsigned long long*)((char*)ctx->small.p + 8);
    t.ops2 = (unsigned long long*)((char*)ctx->ops.p + ops_bytes_offset(m, n));
    t.hdr = d_hdr;
    if (ctx->last.bits) {
      t.xs = (const uint8_t*)ctx->xenc.p;
      t.ys = (const uint8_t*)ctx->yenc.p;
      bits_launch_traceback(st, t);
    } else {
      launch_traceback_walk(st, t);
    }
    HIPCK(ctx, hipGetLastError());
    long long h_hdr[5] = {0, 0, 0, 0, 0};
    int h_err = 0;
    HIPCK(ctx, hipMemcpyAsync(h_hdr, d_hdr, sizeof h_hdr, hipMemcpyDeviceToHost, st));
    HIPCK(ctx, hipMemcpyAsync(&h_err, d_err, 4, hipMemcpyDeviceToHost, st));
    HIPCK(ctx, hipStreamSynchronize(st));
    if (h_err) return fail(ctx, SA_ERR_HIP, "persistent fill (band %d..%d): a bounded wait expired (results discarded)", s_lo, s_hi);
    float t_band = 0;
    HIPCK(ctx, hipEventElapsedTime(&t_band, ctx->ev[5], ctx->ev[6]));
    out->kernel_fill_ms += t_band;  // run_pair adds the checkpoint pass and takes the bands out of traceback_ms
    nops = h_hdr[0];
    const int ni = (int)h_hdr[3], nj = (int)h_hdr[4];
    const bool handover = ni == t.i_min && ni > 0 && nj > 0 && ni < i;
    i = ni;
    j = nj;
    if (!handover) break;
  }
  if (nops > 0) {
    out->ops = (uint8_t*)malloc((size_t)nops);
    if (!out->ops) return fail(ctx, SA_ERR_NOMEM, "host malloc of %lld ops failed", nops);
    launch_expand_ops(ctx, m, n, nops, d_hdr);
    HIPCK(ctx, hipMemcpyAsync(out->ops, ctx->ops.p, (size_t)nops, hipMemcpyDeviceToHost, st));
  }
  HIPCK(ctx, hipEventRecord(ctx->ev[3], st));
  HIPCK(ctx, hipEventSynchronize(ctx->ev[3]));
  out->n_ops = nops;
  out->start_i = i;
  out->start_j = j;
  return SA_OK;
}

// Traceback over the codes of the last fill + D2H of score / end cell / op list.  Synchronises.
int traceback_impl(sa_ctx* ctx, int start_i, int start_j, bool want_tb, sa_result* out) {
  if (ctx->last.ckpt && want_tb) return traceback_ckpt(ctx, start_i, start_j, out);
  hipStream_t st = ctx->stream;
  const int kind = ctx->last.kind, m = ctx->last.m, n = ctx->last.n;
  int* d_score = (int*)ctx->small.p;
  unsigned long long* d_pos = (unsigned long long*)((char*)ctx->small.p + 8);
  long long* d_hdr = (long long*)((char*)ctx->small.p + 64);
  struct {
    int score;
    int pad;
    unsigned long long pos;
    int rest[40];
  } h_sc{};
  long long h_hdr[5] = {0, 0, 0, 0, 0};
  if (want_tb) {
    TraceArgs t{};
    t.kind = kind;
    t.m = m;
    t.n = n;
    t.K = ctx->last.K;
    t.W = ctx->last.W;
    t.start_i = start_i;
    t.start_j = start_j;
    t.stop_at_col0 = ctx->last.first_stripe ? 0 : 1;
    t.trace = (const uint32_t*)ctx->trace.p;
    t.sw_best = d_score;
    t.sw_pos = d_pos;
    t.ops2 = (unsigned long long*)((char*)ctx->ops.p + ops_bytes_offset(m, n));
    t.hdr = d_hdr;
    if (ctx->last.bits) {
      t.xs = (const uint8_t*)ctx->xenc.p;
      t.ys = (const uint8_t*)ctx->yenc.p;
      bits_launch_traceback(st, t);
    } else {
      launch_traceback_walk(st, t);
    }
    launch_expand_ops(ctx, m, n, (long long)m + n, d_hdr);
    HIPCK(ctx, hipGetLastError());
    HIPCK(ctx, hipMemcpyAsync(h_hdr, d_hdr, sizeof h_hdr, hipMemcpyDeviceToHost, st));
  }
  HIPCK(ctx, hipMemcpyAsync(&h_sc, ctx->small.p, sizeof h_sc, hipMemcpyDeviceToHost, st));
  HIPCK(ctx, hipStreamSynchronize(st));
  if (h_sc.rest[36] != 0)  // int index 40 of the small block
    return fail(ctx, SA_ERR_HIP, "persistent fill: a bounded wait on the previous strip expired (results discarded)");
  if (want_tb && h_hdr[0] > 0) {
    out->ops = (uint8_t*)malloc((size_t)h_hdr[0]);
    if (!out->ops) return fail(ctx, SA_ERR_NOMEM, "host malloc of %lld ops failed", h_hdr[0]);
    HIPCK(ctx, hipMemcpyAsync(out->ops, ctx->ops.p, (size_t)h_hdr[0], hipMemcpyDeviceToHost, st));
  }
  HIPCK(ctx, hipEventRecord(ctx->ev[3], st));
  HIPCK(ctx, hipEventSynchronize(ctx->ev[3]));
  out->score = h_sc.score;
  if (kind == KIND_SW) {
    if (h_sc.score > 0) {
      out->end_i = (int64_t)(h_sc.pos >> 32);
      out->end_j = (int64_t)(h_sc.pos & 0xffffffffu);
    }
    out->start_i = out->end_i;
    out->start_j = out->end_j;
  } else {
    out->end_i = start_i >= 0 ? start_i : m;
    out->end_j = start_i >= 0 ? start_j : n;
  }
  if (want_tb) {
    out->n_ops = h_hdr[0];
    out->start_i = h_hdr[3];
    out->start_j = h_hdr[4];
  }
  return SA_OK;
}

// The two O(m+n) border chains, literally as the reference builds them (host):
// row 0 main.cpp:719-725, column 0 main.cpp:601-606.  SW / LCS borders are zero.
void border_chains(int kind, int64_t m, int64_t n, const sa_scoring* sc, std::vector<int>& rowb, std::vector<int>& colb) {
  rowb.assign((size_t)n + 1, 0);
  colb.assign((size_t)m + 1, 0);
  if (kind != KIND_NW) return;
  if (sc->gap_open == std::floor(sc->gap_open) && sc->gap_extend == std::floor(sc->gap_extend) && std::fabs(sc->gap_open) < 1e6 &&
      std::fabs(sc->gap_extend) < 1e6) {
    // integral penalties: the same two chains in int arithmetic (exact: no truncation happens)
    const int go = (int)sc->gap_open, ge = (int)sc->gap_extend;
    for (int64_t j = 1; j <= n; ++j) rowb[j] = (j == 1) ? go : rowb[j - 1] + ge;
    int fprev = kNeg;
    for (int64_t i = 1; i <= m; ++i) {
      fprev = std::max(colb[i - 1] + go, fprev + ge);
      colb[i] = fprev;
    }
    return;
  }
  for (int64_t j = 1; j <= n; ++j) rowb[j] = (j == 1) ? add_gap(rowb[0], sc->gap_open) : add_gap(rowb[j - 1], sc->gap_extend);
  int fprev = kNeg;
  for (int64_t i = 1; i <= m; ++i) {
    const int fo = add_gap(colb[i - 1], sc->gap_open), fe = add_gap(fprev, sc->gap_extend);
    fprev = std::max(fo, fe);
    colb[i] = fprev;
  }
}

int run_pair(sa_ctx* ctx, int kind, const uint8_t* x, int64_t m64, const uint8_t* y, int64_t n64,
             const sa_scoring* sc, uint32_t flags, sa_result* out) {
  if (!ctx) return SA_ERR_ARG;
  if (!out) return fail(ctx, SA_ERR_ARG, "out is NULL");
  memset(out, 0, sizeof *out);
  if (m64 < 0 || n64 < 0 || (m64 > 0 && !x) || (n64 > 0 && !y)) return fail(ctx, SA_ERR_ARG, "bad sequence arguments");
  if (m64 > 0x7ff00000LL || n64 > 0x7ff00000LL) return fail(ctx, SA_ERR_ARG, "sequence longer than int32 indexing (reference Q7)");
  const int m = (int)m64, n = (int)n64;
  // SA_TIMING=1: host-side stages of the call on stderr (where the time outside the kernels goes)
  static const bool timing = getenv("SA_TIMING") != nullptr;
  const auto tp0 = std::chrono::steady_clock::now();
  auto lap = [&](const char* what) {
    if (timing) fprintf(stderr, "[sa_timing] %-28s %8.3f ms since call\n", what, std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - tp0).count());
  };
  int rc = check_scoring(ctx, kind, sc, m, n);
  if (rc) return rc;
  if (kind != KIND_LCS && first_invalid_pair(x, m, y, n, sc->idx256, &out->bad_x, &out->bad_y))
    return fail(ctx, SA_ERR_INVALID_CODE, "invalid residue pair %c,%c", out->bad_x, out->bad_y);
  const bool dump = (flags & SA_FLAG_KEEP_DP) != 0;
  if (dump && ((double)m + 1) * ((double)n + 1) > 1.5e9)
    return fail(ctx, SA_ERR_NOMEM, "SA_FLAG_KEEP_DP: a %d x %d int32 matrix is beyond the 6 GB dump limit", m + 1, n + 1);

  lap("checks, residue scan");
  // borders: closed forms written on the device when nothing else needs the host's chains (fill_impl)
  const bool integral = kind != KIND_NW || (sc->gap_open == std::floor(sc->gap_open) && sc->gap_extend == std::floor(sc->gap_extend));
  const bool dev_borders = ctx->persist && !dump && m > 0 && n > 0 && integral;
  std::vector<int> rowb, colb;
  if (!dev_borders) border_chains(kind, m, n, sc, rowb, colb);
  lap("border chains");
  if (dump) {
    out->dp = (int32_t*)calloc(((size_t)m + 1) * ((size_t)n + 1), 4);
    if (!out->dp) return fail(ctx, SA_ERR_NOMEM, "host malloc of the dp matrix failed");
  }
  if (m == 0 || n == 0) {
    border_only_result(kind, m, n, rowb, colb, out);
  } else {
    bool bits = false;
    if (kind == KIND_LCS && (rc = lcs_bits_fill(ctx, x, m, y, n, flags, out, &bits))) return rc;
    if (!bits && (rc = fill_impl(ctx, kind, x, m, y, n, sc, nullptr, flags, out, rowb, colb))) return rc;
    lap("fill enqueued");
    if (dump) HIPCK(ctx, hipMemcpyAsync(out->dp, ctx->dpbuf.p, ((size_t)m + 1) * ((size_t)n + 1) * 4, hipMemcpyDeviceToHost, ctx->stream));
    if ((rc = traceback_impl(ctx, -1, -1, !(flags & SA_FLAG_NO_TRACEBACK), out))) return rc;
    lap("traceback done (sync)");
    float t01 = 0, t12 = 0, t23 = 0;
    HIPCK(ctx, hipEventElapsedTime(&t01, ctx->ev[0], ctx->ev[1]));
    HIPCK(ctx, hipEventElapsedTime(&t12, ctx->ev[1], ctx->ev[2]));
    HIPCK(ctx, hipEventElapsedTime(&t23, ctx->ev[2], ctx->ev[3]));
    // a checkpointed traceback recomputed bands of codes: their launches are fill work (traceback_ckpt summed them
    // into kernel_fill_ms), so that kernel_fill_ms covers EVERY fill launch of the call
    const float t_bands = out->kernel_fill_ms;
    out->fill_ms = t01 + t12 + t_bands;
    out->kernel_fill_ms = t12 + t_bands;
    out->traceback_ms = t23 - t_bands;
  }
  if (dump) {  // borders of the dumped matrix (the kernel writes interior cells only)
    const size_t w = (size_t)n + 1;
    for (int j = 0; j <= n; ++j) out->dp[j] = rowb[j];
    for (int i = 0; i <= m; ++i) out->dp[(size_t)i * w] = colb[i];
  }
  return SA_OK;
}

// ---- column stripes (multi-GPU, one process per GPU; the caller moves the borders) ----
int run_stripe_fill(sa_ctx* ctx, int kind, const uint8_t* x, int64_t m64, const uint8_t* y, int64_t n64, const sa_scoring* sc,
                    const sa_stripe_desc* d, sa_result* out) {
  if (!ctx) return SA_ERR_ARG;
  if (!out || !d) return fail(ctx, SA_ERR_ARG, "NULL argument");
  memset(out, 0, sizeof *out);
  if (m64 <= 0 || n64 <= 0 || !x || !y) return fail(ctx, SA_ERR_ARG, "a stripe needs m > 0 and n > 0");
  if (m64 > 0x7ff00000LL || d->n_total > 0x7ff00000LL || d->col_begin < 0 || d->col_begin + n64 > d->n_total)
    return fail(ctx, SA_ERR_ARG, "bad stripe geometry");
  if ((!d->is_first && !d->left) || (!d->is_last && !d->right) || d->chunk_rows <= 0) return fail(ctx, SA_ERR_ARG, "bad stripe borders");
  const int m = (int)m64, n = (int)n64;
  int rc = check_scoring(ctx, kind, sc, m, d->n_total);
  if (rc) return rc;
  if (kind != KIND_LCS) {
    if (first_invalid_pair(x, m, y, n, sc->idx256, &out->bad_x, &out->bad_y))
      return fail(ctx, SA_ERR_INVALID_CODE, "invalid residue pair %c,%c", out->bad_x, out->bad_y);
  }
  // row 0 of this stripe = the global row-0 chain from column col_begin on
  std::vector<int> rowb_all, colb;
  border_chains(kind, d->is_first ? m : 0, d->col_begin + n, sc, rowb_all, colb);
  std::vector<int> rowb(rowb_all.begin() + d->col_begin, rowb_all.end());
  StripeDesc sd;
  sd.n_total = d->n_total;
  sd.col_begin = d->col_begin;
  sd.is_first = d->is_first;
  sd.is_last = d->is_last;
  sd.left = d->left;
  sd.right = d->right;
  sd.chunk_rows = d->chunk_rows;
  sd.on_event = d->on_event;
  sd.user = d->user;
  sd.stream_in = (hipStream_t)d->stream_in;
  sd.stream_out = (hipStream_t)d->stream_out;
  if ((sd.stream_in && sd.stream_in == ctx->stream) || (sd.stream_out && sd.stream_out == ctx->stream))
    return fail(ctx, SA_ERR_ARG, "stream_in / stream_out must differ from the ctx stream");
  bool bits = false;
  if (kind == KIND_LCS && (rc = lcs_bits_fill(ctx, x, m, y, n, 0, out, &bits, &sd))) return rc;
  if (!bits && (rc = fill_impl(ctx, kind, x, m, y, n, sc, &sd, 0, out, rowb, colb))) return rc;
  int h_small[41] = {0};  // [0] score / SW stripe best, [2..3] SW end cell, [40] persistent-kernel error flag
  HIPCK(ctx, hipMemcpyAsync(h_small, ctx->small.p, sizeof h_small, hipMemcpyDeviceToHost, ctx->stream));
  HIPCK(ctx, hipStreamSynchronize(ctx->stream));
  if (h_small[40] != 0) {
    ctx->last.valid = 0;
    return fail(ctx, SA_ERR_HIP, "persistent stripe fill: a bounded wait (previous strip / left border / border send) expired");
  }
  const int h_score = h_small[0];
  if (ctx->last.args.wait_ticks) {  // 100 MHz ticks summed over the strips -> ms per wave of the launch
    unsigned long long ticks = 0;
    HIPCK(ctx, hipMemcpy(&ticks, ctx->last.args.wait_ticks, 8, hipMemcpyDeviceToHost));
    const int S = ctx->last.S;
    const int waves = std::min(S, std::min((S + kWavesPerBlock - 1) / kWavesPerBlock, persist_wg_cap(ctx)) * kWavesPerBlock);
    out->border_wait_ms = (float)((double)ticks / 1e5 / std::max(1, waves));
  }
  float t01 = 0, t12 = 0;
  HIPCK(ctx, hipEventElapsedTime(&t01, ctx->ev[0], ctx->ev[1]));
  HIPCK(ctx, hipEventElapsedTime(&t12, ctx->ev[1], ctx->ev[2]));
  out->fill_ms = t01 + t12;
  out->kernel_fill_ms = t12;
  out->score = h_score;  // NW / LCS: meaningful on the last stripe only; SW: this stripe's best
  out->end_i = m;
  out->end_j = n;
  if (kind == KIND_SW) {  // first strict row-major maximum inside the stripe, j local (0,0 if the best is 0)
    unsigned long long pos;
    memcpy(&pos, &h_small[2], 8);
    out->end_i = h_score > 0 ? (int64_t)(pos >> 32) : 0;
    out->end_j = h_score > 0 ? (int64_t)(pos & 0xffffffffu) : 0;
  }
  return SA_OK;
}

int run_stripe_traceback(sa_ctx* ctx, int64_t start_i, int64_t start_j, sa_result* out) {
  if (!ctx) return SA_ERR_ARG;
  if (!out) return fail(ctx, SA_ERR_ARG, "out is NULL");
  memset(out, 0, sizeof *out);
  if (!ctx->last.valid) return fail(ctx, SA_ERR_ARG, "no fill to trace back");
  if (start_i < 0 || start_i > ctx->last.m || start_j < 0 || start_j > ctx->last.n) return fail(ctx, SA_ERR_ARG, "start cell outside the stripe");
  HIPCK(ctx, hipSetDevice(ctx->device));
  HIPCK(ctx, hipEventRecord(ctx->ev[2], ctx->stream));
  int rc = traceback_impl(ctx, (int)start_i, (int)start_j, true, out);
  if (rc) return rc;
  float t23 = 0;
  HIPCK(ctx, hipEventElapsedTime(&t23, ctx->ev[2], ctx->ev[3]));
  out->traceback_ms = t23;
  return SA_OK;
}

// Host op pools of sa_sw_batch_align: page-locked (the 1.3 GB of moves of a million pairs come back at
// link speed instead of through the pageable-copy staging) and recycled -- sa_free_pool parks one
// buffer here and the next call takes it back if it is large enough, so a serving loop pins once.
struct PoolCache {
  std::mutex mu;
  void* p = nullptr;
  size_t cap = 0;
} g_pool_cache;

uint8_t* pool_acquire(size_t bytes) {
  {
    std::lock_guard<std::mutex> lk(g_pool_cache.mu);
    if (g_pool_cache.p && g_pool_cache.cap >= bytes) {
      void* p = g_pool_cache.p;
      g_pool_cache.p = nullptr;
      return (uint8_t*)p;
    }
  }
  void* p = nullptr;
  // 16 bytes in front remember the capacity for pool_release
  if (hipHostMalloc(&p, bytes + 16, hipHostMallocDefault) != hipSuccess) {
    (void)hipGetLastError();
    return nullptr;
  }
  *(size_t*)p = bytes;
  return (uint8_t*)p + 16;
}

void pool_release(uint8_t* pool) {
  if (!pool) return;
  void* base = pool - 16;
  const size_t cap = *(size_t*)base;
  void* drop = base;
  {
    std::lock_guard<std::mutex> lk(g_pool_cache.mu);
    if (!g_pool_cache.p || g_pool_cache.cap < cap) {
      drop = g_pool_cache.p ? (char*)g_pool_cache.p - 16 : nullptr;
      g_pool_cache.p = pool;
      g_pool_cache.cap = cap;
    }
  }
  if (drop) (void)hipHostFree(drop);
}

// Gap penalties of the many-pair kernels.  The reference adds a double penalty to an int and truncates toward zero
// per operation (main.cpp:603-604, 958-964).  SW floors E and F at 0 individually, so a fractional penalty acts
// exactly like its floor (see fill_impl); NW takes the floor plus the sign fix of add_gap<true> inside the matrix and
// the ceiling along its non-positive borders.
void set_batch_gaps(BatchArgs& a, const sa_scoring* sc, bool nw) {
  a.go = (int)std::floor(sc->gap_open);
  a.ge = (int)std::floor(sc->gap_extend);
  a.cgo = (int)std::ceil(sc->gap_open);
  a.cge = (int)std::ceil(sc->gap_extend);
  a.fo = (nw && sc->gap_open != std::floor(sc->gap_open)) ? 4 : 0;
  a.fe = (nw && sc->gap_extend != std::floor(sc->gap_extend)) ? 4 : 0;
}

// Outputs of the aligning variant (sa_sw_batch_align): per pair the 'X' cell and the op list.
struct BatchAlign {
  int kind;           // KIND_SW: local alignments; KIND_NW: global alignments (end = (m, n), start = (0, 0))
  int32_t* start_i;
  int32_t* start_j;
  int64_t* ops_off;   // byte offset of the pair's ops in the pool: sum over earlier pairs of (m + n) rounded up to 4
  int32_t* n_ops;
  uint8_t** pool;
  int64_t* pool_bytes;
  int diag_only = 0;  // KIND_SW: the seed_aligner's window traceback (ungapped diagonal while S > 0): n_ops = its length, no op pool
};

// Many independent pairs: one wavefront per pair (sa_batch.h).  Pairs the fast kernel does not
// admit (query > 1024 rows, very long target) go through the single-pair SW kernels.
// al == nullptr: score + end cell only.  Otherwise the fill also stores direction codes (in chunks
// that bound the device memory taken by them) and a second kernel walks every pair back.
int run_batch(sa_ctx* ctx, const uint8_t* seqs, const int64_t* off_x, const int32_t* len_x, const int64_t* off_y,
              const int32_t* len_y, int64_t n_pairs, const sa_scoring* sc, int32_t* score, int32_t* end_i, int32_t* end_j,
              float* kernel_ms, const BatchAlign* al = nullptr) {
  if (!ctx) return SA_ERR_ARG;
  if (kernel_ms) *kernel_ms = 0.f;
  if (al) {
    if ((al->kind == KIND_SW && (!al->start_i || !al->start_j)) || !al->n_ops || (!al->diag_only && (!al->ops_off || !al->pool)))
      return fail(ctx, SA_ERR_ARG, "bad batch arguments");
    if (al->diag_only && sc && !(sc->gap_open < 0))
      return fail(ctx, SA_ERR_UNSUPPORTED, "the window traceback is the ungapped diagonal only for a negative gap-open penalty");
    if (al->pool) *al->pool = nullptr;
    if (al->pool_bytes) *al->pool_bytes = 0;
  }
  const bool resident = ctx->dev_pool != nullptr;
  if (n_pairs < 0 || (n_pairs > 0 && ((!seqs && !resident) || !off_x || !len_x || !off_y || !len_y || !score || !end_i || !end_j)))
    return fail(ctx, SA_ERR_ARG, "bad batch arguments");
  const bool lcs = al && al->kind == KIND_LCS;  // raw byte equality: no scoring scheme
  static const int8_t kNoSub[1] = {0};
  static const uint8_t kNoMap[256] = {0};
  static const sa_scoring kNoScoring{0.0, 0.0, kNoSub, 1, kNoMap};
  if (lcs) sc = &kNoScoring;
  if (!sc || !sc->sub || !sc->idx256 || sc->alpha < 1 || sc->alpha > 31) return fail(ctx, SA_ERR_ARG, "bad scoring");
  if (sc->gap_open > 0 || sc->gap_extend > 0 || sc->gap_open < -1e6 || sc->gap_extend < -1e6)
    return fail(ctx, SA_ERR_ARG, "batch path needs non-positive gap penalties");
  if (int rcs = check_batch_scoring(ctx, sc)) return rcs;
  if (n_pairs == 0) return SA_OK;
  if (n_pairs > 0x7fffff00LL) return fail(ctx, SA_ERR_ARG, "too many pairs");
  HIPCK(ctx, hipSetDevice(ctx->device));
  hipStream_t st = ctx->stream;
  const int alpha = sc->alpha;
  int64_t pool_total = 0;
  if (al) {
    int smax = 0;
    for (int i = 0; i < alpha * alpha; ++i) smax = std::max(smax, std::abs((int)sc->sub[i]));
    if (smax > 31) return fail(ctx, SA_ERR_UNSUPPORTED, "substitution scores beyond +-31 do not fit the int8 tagged profile");
    for (int64_t k = 0; k < n_pairs; ++k) {
      if (len_x[k] < 0 || len_y[k] < 0) return fail(ctx, SA_ERR_ARG, "negative length in pair %lld", (long long)k);
      al->n_ops[k] = 0;
      if (al->start_i) al->start_i[k] = al->start_j[k] = 0;
      if (al->diag_only) continue;
      pool_total += ((int64_t)len_x[k] + len_y[k] + 3) & ~3LL;  // the slots are laid out below, in processing order
    }
    if (!al->diag_only) {
      *al->pool = pool_acquire((size_t)std::max<int64_t>(pool_total, 4));
      if (!*al->pool) return fail(ctx, SA_ERR_NOMEM, "host malloc of the %lld-byte op pool failed", (long long)pool_total);
      if (al->pool_bytes) *al->pool_bytes = pool_total;
    }
  }
  struct PoolGuard {  // the pool goes back to the caller only with SA_OK / SA_ERR_INVALID_CODE
    const BatchAlign* al;
    sa_ctx* ctx;
    bool keep = false;
    ~PoolGuard() {
      if (ctx->copy_stream) (void)hipStreamSynchronize(ctx->copy_stream);  // no copy may outlive the pool
      if (al && al->pool && !keep && *al->pool) {
        pool_release(*al->pool);
        *al->pool = nullptr;
      }
    }
  } guard{al, ctx};

  // group by rows-per-lane K = ceil(m/64); counting sort keeps the caller's order inside a group
  std::vector<int64_t> slow, empty_side, zeros;
  std::vector<int> group_count(17, 0), group_nmax(17, 0);
  int64_t pool_end = 0;
  for (int64_t k = 0; k < n_pairs; ++k) {
    if (len_x[k] < 0 || len_y[k] < 0 || off_x[k] < 0 || off_y[k] < 0) return fail(ctx, SA_ERR_ARG, "negative offset/length in pair %lld", (long long)k);
    pool_end = std::max(pool_end, std::max(off_x[k] + len_x[k], off_y[k] + len_y[k]));
    if (len_x[k] == 0 || len_y[k] == 0) {
      score[k] = 0;
      end_i[k] = end_j[k] = 0;
      zeros.push_back(k);
      if (al && al->kind != KIND_SW) empty_side.push_back(k);  // a border walk (main.cpp: row 0 / column 0)
    } else if (len_x[k] > kBatchMaxM || len_y[k] > kBatchMaxN - 200) {
      if (al && al->diag_only)
        return fail(ctx, SA_ERR_UNSUPPORTED, "window %lld (%d x %d) is beyond the batch kernels' %d x %d", (long long)k, len_x[k],
                    len_y[k], kBatchMaxM, kBatchMaxN - 200);
      slow.push_back(k);
    } else {
      const int K = (len_x[k] + 63) / 64;
      ++group_count[K];
      group_nmax[K] = std::max(group_nmax[K], len_y[k]);
    }
  }
  PairDesc* desc = nullptr;  // page-locked, sorted by K (big pairs first)
  size_t n_desc = 0;
  std::vector<int> group_start(18, 0);
  for (int K = 16; K >= 1; --K) group_start[K - 1 >= 0 ? K - 1 : 0] = 0;  // (filled below)
  {
    int acc = 0;
    for (int K = 16; K >= 1; --K) {  // big pairs first
      group_start[K] = acc;
      acc += group_count[K];
    }
    n_desc = (size_t)acc;
    if (int rcd = ensure_host(ctx, ctx->h_desc, std::max<size_t>(n_desc, 1) * sizeof(PairDesc))) return rcd;
    desc = (PairDesc*)ctx->h_desc.p;
    std::vector<int> cur(group_start.begin(), group_start.end());
    for (int64_t k = 0; k < n_pairs; ++k) {
      if (len_x[k] == 0 || len_y[k] == 0 || len_x[k] > kBatchMaxM || len_y[k] > kBatchMaxN - 200) continue;
      const int K = (len_x[k] + 63) / 64;
      desc[(size_t)cur[K]++] = PairDesc{off_x[k], off_y[k], len_x[k], len_y[k], (int)k, 0, 0, 0};
    }
  }
  if (al && !al->diag_only) {
    // op slots in PROCESSING order (the K-sorted list, then the pairs handled on the host): the moves of a chunk are
    // one contiguous range of the pool and can be copied out while the next chunk computes
    int64_t off = 0;
    for (size_t q = 0; q < n_desc; ++q) {
      al->ops_off[desc[q].idx] = off;
      off += ((int64_t)desc[q].m + desc[q].n + 3) & ~3LL;
    }
    for (int64_t k = 0; k < n_pairs; ++k) {
      if (!(len_x[k] == 0 || len_y[k] == 0 || len_x[k] > kBatchMaxM || len_y[k] > kBatchMaxN - 200)) continue;
      al->ops_off[k] = off;
      off += ((int64_t)len_x[k] + len_y[k] + 3) & ~3LL;
    }
  }

  int rc = SA_OK;
  int h_err = 0;
  if (resident && pool_end > ctx->dev_pool_bytes)
    return fail(ctx, SA_ERR_ARG, "a pair ends at byte %lld of a %lld-byte device pool", (long long)pool_end, (long long)ctx->dev_pool_bytes);
  if (n_desc) {
    if (!resident && (rc = ensure(ctx, ctx->b_seq, (size_t)pool_end + 64))) return rc;
    if ((rc = ensure(ctx, ctx->b_desc, n_desc * sizeof(PairDesc)))) return rc;
    if ((rc = ensure(ctx, ctx->b_out, (size_t)n_pairs * 24 + 1024 + 2048))) return rc;
    // aligning variant: cut the (K-sorted) pair list into chunks whose codes fit the trace budget
    // (code offsets are relative to the chunk's buffer); the op slots live in a device image of the
    // pool, and each chunk's range goes back on the copy stream as soon as its walks are done
    struct Chunk {
      size_t lo, hi;
      long long trace_dwords, ops_bytes;
    };
    std::vector<Chunk> chunks;
    if (al) {
      const long long budget = (long long)ctx->batch_trace_budget / 4;
      Chunk c{0, 0, 0, 0};
      for (size_t q = 0; q < n_desc; ++q) {
        const long long td = batch_trace_dwords(desc[q].m, desc[q].n);
        if (c.hi > c.lo && c.trace_dwords + td > budget) {
          chunks.push_back(c);
          c = Chunk{q, q, 0, 0};
        }
        desc[q].off_trace = c.trace_dwords;
        desc[q].off_ops = al->diag_only ? 0 : al->ops_off[desc[q].idx];
        c.trace_dwords += td;
        c.ops_bytes += ((long long)desc[q].m + desc[q].n + 3) & ~3LL;
        c.hi = q + 1;
      }
      chunks.push_back(c);
      long long tmax = 0;
      for (const Chunk& ch : chunks) tmax = std::max(tmax, ch.trace_dwords);
      if ((rc = ensure(ctx, ctx->trace, (size_t)tmax * 4))) return rc;
      if ((rc = ensure(ctx, ctx->ops, (size_t)pool_total + 64))) return rc;
    }
    if (!resident) HIPCK(ctx, hipMemcpyAsync(ctx->b_seq.p, seqs, (size_t)pool_end, hipMemcpyHostToDevice, st));
    HIPCK(ctx, hipMemcpyAsync(ctx->b_desc.p, desc, n_desc * sizeof(PairDesc), hipMemcpyHostToDevice, st));
    int* d_score = (int*)ctx->b_out.p;
    int* d_ei = d_score + n_pairs;
    int* d_ej = d_ei + n_pairs;
    int* d_nops = d_ej + n_pairs;
    int* d_si = d_nops + n_pairs;
    int* d_sj = d_si + n_pairs;
    int* d_err = d_sj + n_pairs;
    int8_t* d_sub = (int8_t*)(d_err + 16);
    uint8_t* d_map = (uint8_t*)(d_sub + 1024);
    HIPCK(ctx, hipMemsetAsync(d_err, 0, 64, st));
    HIPCK(ctx, hipMemcpyAsync(d_sub, sc->sub, (size_t)alpha * alpha, hipMemcpyHostToDevice, st));
    HIPCK(ctx, hipMemcpyAsync(d_map, sc->idx256, 256, hipMemcpyHostToDevice, st));
    BatchArgs a{};
    a.seqs = resident ? ctx->dev_pool : (const uint8_t*)ctx->b_seq.p;
    a.sub = d_sub;
    a.idx256 = d_map;
    a.alpha = alpha;
    set_batch_gaps(a, sc, al && al->kind == KIND_NW);
    a.score = d_score;
    a.end_i = d_ei;
    a.end_j = d_ej;
    a.err = d_err;
    a.trace = (uint32_t*)ctx->trace.p;
    a.ops = (uint8_t*)ctx->ops.p;
    a.n_ops = d_nops;
    a.start_i = d_si;
    a.start_j = d_sj;
    if (!al) chunks.push_back({0, n_desc, 0, 0});
    const bool pool_out = al && !al->diag_only && pool_total > 0;
    if (pool_out && !ctx->copy_stream) {
      HIPCK(ctx, hipStreamCreateWithFlags(&ctx->copy_stream, hipStreamNonBlocking));
      HIPCK(ctx, hipEventCreateWithFlags(&ctx->ev_chunk, hipEventDisableTiming));
    }
    float ms_total = 0;
    for (const auto& ch : chunks) {
      HIPCK(ctx, hipEventRecord(ctx->ev[0], st));
      for (int K = 16; K >= 1; --K) {  // the part of group K inside this chunk
        const size_t glo = std::max<size_t>(ch.lo, (size_t)group_start[K]);
        const size_t ghi = std::min<size_t>(ch.hi, (size_t)group_start[K] + group_count[K]);
        if (glo >= ghi) continue;
        a.desc = (const PairDesc*)ctx->b_desc.p + glo;
        a.n_desc = (int)(ghi - glo);
        a.CB = kBatchCB;
        a.ycap = batch_ycap(group_nmax[K]);
        if (lcs)
          launch_lcs_batch_k(K, st, a.n_desc, a);
        else if (al && al->kind == KIND_NW)
          launch_nw_batch_k(K, st, a.n_desc, alpha, a);
        else
          launch_batch_k(K, st, a.n_desc, alpha, a, al != nullptr);
      }
      if (al) {
        a.desc = (const PairDesc*)ctx->b_desc.p + ch.lo;
        a.n_desc = (int)(ch.hi - ch.lo);
        if (lcs)
          launch_batch_tracebacks(2, st, a);
        else if (al->kind == KIND_NW)
          launch_batch_tracebacks(1, st, a);
        else if (al->diag_only)
          launch_batch_tracebacks(3, st, a);
        else
          launch_batch_tracebacks(0, st, a);
      }
      HIPCK(ctx, hipGetLastError());
      HIPCK(ctx, hipEventRecord(ctx->ev[1], st));
      if (pool_out && ch.ops_bytes > 0) {
        const int64_t lo = al->ops_off[desc[ch.lo].idx];
        HIPCK(ctx, hipEventRecord(ctx->ev_chunk, st));
        HIPCK(ctx, hipStreamWaitEvent(ctx->copy_stream, ctx->ev_chunk, 0));
        HIPCK(ctx, hipMemcpyAsync(*al->pool + lo, (const uint8_t*)ctx->ops.p + lo, (size_t)ch.ops_bytes, hipMemcpyDeviceToHost,
                                  ctx->copy_stream));
      }
      HIPCK(ctx, hipStreamSynchronize(st));
      float ms = 0;
      HIPCK(ctx, hipEventElapsedTime(&ms, ctx->ev[0], ctx->ev[1]));
      ms_total += ms;
    }
    // results come back for every index; slots of pairs handled elsewhere are overwritten below
    if (kernel_ms) *kernel_ms = ms_total;
    if (!al) {  // the kernels wrote in the caller's order: straight into the caller's arrays
      HIPCK(ctx, hipMemcpyAsync(score, d_score, (size_t)n_pairs * 4, hipMemcpyDeviceToHost, st));
      HIPCK(ctx, hipMemcpyAsync(end_i, d_ei, (size_t)n_pairs * 4, hipMemcpyDeviceToHost, st));
      HIPCK(ctx, hipMemcpyAsync(end_j, d_ej, (size_t)n_pairs * 4, hipMemcpyDeviceToHost, st));
      HIPCK(ctx, hipMemcpyAsync(&h_err, d_err, 4, hipMemcpyDeviceToHost, st));
      HIPCK(ctx, hipStreamSynchronize(st));
      for (int64_t k : zeros) score[k] = end_i[k] = end_j[k] = 0;  // their device slots were never written
      n_desc = 0;
    } else {  // the same for the aligning variants: six arrays in the caller's order
      HIPCK(ctx, hipMemcpyAsync(score, d_score, (size_t)n_pairs * 4, hipMemcpyDeviceToHost, st));
      HIPCK(ctx, hipMemcpyAsync(end_i, d_ei, (size_t)n_pairs * 4, hipMemcpyDeviceToHost, st));
      HIPCK(ctx, hipMemcpyAsync(end_j, d_ej, (size_t)n_pairs * 4, hipMemcpyDeviceToHost, st));
      HIPCK(ctx, hipMemcpyAsync(al->n_ops, d_nops, (size_t)n_pairs * 4, hipMemcpyDeviceToHost, st));
      if (al->start_i) {
        HIPCK(ctx, hipMemcpyAsync(al->start_i, d_si, (size_t)n_pairs * 4, hipMemcpyDeviceToHost, st));
        HIPCK(ctx, hipMemcpyAsync(al->start_j, d_sj, (size_t)n_pairs * 4, hipMemcpyDeviceToHost, st));
      }
      HIPCK(ctx, hipMemcpyAsync(&h_err, d_err, 4, hipMemcpyDeviceToHost, st));
      HIPCK(ctx, hipStreamSynchronize(st));
      if (pool_out) HIPCK(ctx, hipStreamSynchronize(ctx->copy_stream));
      auto clear = [&](int64_t k) {
        al->n_ops[k] = 0;
        if (al->start_i) al->start_i[k] = al->start_j[k] = 0;
      };
      for (int64_t k : zeros) {  // device slots that were never written
        score[k] = end_i[k] = end_j[k] = 0;
        clear(k);
      }
      for (int64_t k : slow) clear(k);
      if (h_err)  // a pair with an invalid residue: score INT32_MIN, nothing else
        for (size_t q = 0; q < n_desc; ++q)
          if (score[desc[q].idx] == INT32_MIN) clear(desc[q].idx);
    }
  }
  std::vector<uint8_t> hx, hy;  // a resident pool: the single-pair path takes host buffers
  for (int64_t k : slow) {
    sa_result r;
    const uint8_t *px = seqs + off_x[k], *py = seqs + off_y[k];
    if (resident) {
      hx.resize((size_t)len_x[k]);
      hy.resize((size_t)len_y[k]);
      HIPCK(ctx, hipMemcpy(hx.data(), ctx->dev_pool + off_x[k], hx.size(), hipMemcpyDeviceToHost));
      HIPCK(ctx, hipMemcpy(hy.data(), ctx->dev_pool + off_y[k], hy.size(), hipMemcpyDeviceToHost));
      px = hx.data();
      py = hy.data();
    }
    const int rc1 = run_pair(ctx, al ? al->kind : KIND_SW, px, len_x[k], py, len_y[k], sc,
                             al ? 0 : SA_FLAG_NO_TRACEBACK, &r);
    if (rc1 == SA_OK) {
      score[k] = (int32_t)r.score;
      end_i[k] = (int32_t)r.end_i;
      end_j[k] = (int32_t)r.end_j;
      if (al) {
        al->n_ops[k] = (int32_t)r.n_ops;
        if (al->start_i) {
          al->start_i[k] = (int32_t)r.start_i;
          al->start_j[k] = (int32_t)r.start_j;
        }
        if (r.n_ops) memcpy(*al->pool + al->ops_off[k], r.ops, (size_t)r.n_ops);
      }
      if (kernel_ms) *kernel_ms += r.kernel_fill_ms;
    } else if (rc1 == SA_ERR_INVALID_CODE) {
      score[k] = INT32_MIN;
      end_i[k] = end_j[k] = 0;
      h_err = 1;
    } else {
      return rc1;
    }
    sa_free_result(&r);
  }
  for (int64_t k : empty_side) {  // global alignment against nothing: the reference's border chain and walk
    std::vector<int> rowb, colb;
    border_chains(al->kind, len_x[k], len_y[k], sc, rowb, colb);
    sa_result r{};
    border_only_result(al->kind, len_x[k], len_y[k], rowb, colb, &r);
    score[k] = (int32_t)r.score;
    end_i[k] = len_x[k];
    end_j[k] = len_y[k];
    al->n_ops[k] = (int32_t)r.n_ops;
    if (r.n_ops) memcpy(*al->pool + al->ops_off[k], r.ops, (size_t)r.n_ops);
    sa_free_result(&r);
  }
  guard.keep = true;
  if (h_err) return fail(ctx, SA_ERR_INVALID_CODE, "at least one pair holds a residue outside the scoring alphabet (its score is INT32_MIN)");
  return SA_OK;
}

// sa_sw_batch with every device buffer inside `scratch` and every host buffer inside the ctx's pinned staging
// block: no hipMalloc / hipFree / hipHostMalloc, no std::vector.  Same kernels, same results.
int run_batch_noalloc(sa_ctx* ctx, const uint8_t* seqs, const int64_t* off_x, const int32_t* len_x, const int64_t* off_y,
                      const int32_t* len_y, int64_t n_pairs, const sa_scoring* sc, int32_t* score, int32_t* end_i, int32_t* end_j,
                      float* kernel_ms) {
  if (!ctx) return SA_ERR_ARG;
  if (kernel_ms) *kernel_ms = 0.f;
  if (!ctx->na_host) return fail(ctx, SA_ERR_HIP, "sa_sw_batch_noalloc: the pinned staging block was not set up by sa_create");
  const bool resident = ctx->dev_pool != nullptr;
  if (n_pairs < 0 || (n_pairs > 0 && ((!seqs && !resident) || !off_x || !len_x || !off_y || !len_y || !score || !end_i || !end_j)))
    return fail(ctx, SA_ERR_ARG, "bad batch arguments");
  if (n_pairs > (int64_t)kNaMaxPairs)
    return fail(ctx, SA_ERR_ARG, "sa_sw_batch_noalloc: %lld pairs exceed SA_NOALLOC_MAX_PAIRS = %zu (split the batch or use sa_sw_batch)",
                (long long)n_pairs, kNaMaxPairs);
  if (!sc || !sc->sub || !sc->idx256 || sc->alpha < 1 || sc->alpha > 31) return fail(ctx, SA_ERR_ARG, "bad scoring");
  if (sc->gap_open > 0 || sc->gap_extend > 0 || sc->gap_open < -1e6 || sc->gap_extend < -1e6)
    return fail(ctx, SA_ERR_ARG, "batch path needs non-positive gap penalties");
  if (int rcs = check_batch_scoring(ctx, sc)) return rcs;
  if (n_pairs == 0) return SA_OK;
  HIPCK(ctx, hipSetDevice(ctx->device));
  hipStream_t st = ctx->stream;
  const int alpha = sc->alpha;
  // this device's address of the scratch array: looked up on the entry point's first call on the device (with the device
  // current), kept per device in sa_noalloc.hip and per ctx here
  if (!ctx->na_scratch) {
    const hipError_t e = na_scratch_for_device(ctx->device, &ctx->na_scratch);
    if (e != hipSuccess || !ctx->na_scratch) return fail(ctx, SA_ERR_HIP, "sa_sw_batch_noalloc: hipGetSymbolAddress(scratch) failed: %s", hipGetErrorString(e));
  }
  const NaBusy busy(ctx->device);  // released on every return path below
  if (!busy.held) return fail(ctx, SA_ERR_ARG, "sa_sw_batch_noalloc: scratch in use (another call on GPU %d has not returned yet)", ctx->device);

  // counting sort by rows per lane K = ceil(m / 64), big pairs first, straight into the pinned staging block
  int group_count[18] = {0}, group_nmax[18] = {0}, group_start[18] = {0}, cur[18];
  int64_t pool_end = 0;
  for (int64_t k = 0; k < n_pairs; ++k) {
    if (len_x[k] < 0 || len_y[k] < 0 || off_x[k] < 0 || off_y[k] < 0) return fail(ctx, SA_ERR_ARG, "negative offset/length in pair %lld", (long long)k);
    pool_end = std::max(pool_end, std::max(off_x[k] + len_x[k], off_y[k] + len_y[k]));
    if (len_x[k] == 0 || len_y[k] == 0) continue;
    if (len_x[k] > kBatchMaxM || len_y[k] > kBatchMaxN - 200)
      return fail(ctx, SA_ERR_UNSUPPORTED, "sa_sw_batch_noalloc: pair %lld (%d x %d) is beyond the batch kernels' %d x %d (sa_sw_batch takes it)",
                  (long long)k, len_x[k], len_y[k], kBatchMaxM, kBatchMaxN - 200);
    const int K = (len_x[k] + 63) / 64;
    ++group_count[K];
    group_nmax[K] = std::max(group_nmax[K], len_y[k]);
  }
  if (resident) {
    if (pool_end > ctx->dev_pool_bytes)
      return fail(ctx, SA_ERR_ARG, "a pair ends at byte %lld of a %lld-byte device pool", (long long)pool_end, (long long)ctx->dev_pool_bytes);
  } else if (pool_end > (int64_t)kNaMaxSeq) {
    return fail(ctx, SA_ERR_ARG, "sa_sw_batch_noalloc: the pairs span %lld sequence bytes, beyond SA_NOALLOC_MAX_SEQ_BYTES = %zu "
                "(split the batch, set a device pool, or use sa_sw_batch)", (long long)pool_end, kNaMaxSeq);
  }
  int n_desc = 0;
  for (int K = 16; K >= 1; --K) {
    group_start[K] = n_desc;
    n_desc += group_count[K];
  }
  for (int K = 0; K < 18; ++K) cur[K] = group_start[K];
  PairDesc* h_desc = (PairDesc*)ctx->na_host;
  int32_t* h_out = (int32_t*)(ctx->na_host + kNaMaxPairs * sizeof(PairDesc));
  for (int64_t k = 0; k < n_pairs; ++k) {
    if (len_x[k] == 0 || len_y[k] == 0) continue;
    const int K = (len_x[k] + 63) / 64;
    h_desc[cur[K]++] = PairDesc{off_x[k], off_y[k], len_x[k], len_y[k], (int)k, 0, 0, 0};
  }

  unsigned char* d = ctx->na_scratch;  // this device's address of `scratch`
  int* d_score = (int*)(d + kNaOutOff);
  int* d_ei = d_score + n_pairs;
  int* d_ej = d_ei + n_pairs;
  int* d_err = (int*)(d + kNaErrOff);
  int h_err = 0;
  if (n_desc) {
    // the array keeps the last call's contents: the error flag must start at zero; every other word that is read
    // back (score / end cell of the n_desc pairs) is written by this call's kernels, the rest is overwritten below
    HIPCK(ctx, hipMemsetAsync(d_err, 0, 64, st));
    if (!resident) HIPCK(ctx, hipMemcpyAsync(d, seqs, (size_t)pool_end, hipMemcpyHostToDevice, st));
    HIPCK(ctx, hipMemcpyAsync(d + kNaDescOff, h_desc, (size_t)n_desc * sizeof(PairDesc), hipMemcpyHostToDevice, st));
    HIPCK(ctx, hipMemcpyAsync(d + kNaSubOff, sc->sub, (size_t)alpha * alpha, hipMemcpyHostToDevice, st));
    HIPCK(ctx, hipMemcpyAsync(d + kNaMapOff, sc->idx256, 256, hipMemcpyHostToDevice, st));
    BatchArgs a{};
    a.seqs = resident ? ctx->dev_pool : (const uint8_t*)d;
    a.sub = (const int8_t*)(d + kNaSubOff);
    a.idx256 = (const uint8_t*)(d + kNaMapOff);
    a.alpha = alpha;
    set_batch_gaps(a, sc, false);
    a.score = d_score;
    a.end_i = d_ei;
    a.end_j = d_ej;
    a.err = d_err;
    a.CB = kBatchCB;
    HIPCK(ctx, hipEventRecord(ctx->ev[0], st));
    for (int K = 16; K >= 1; --K) {
      if (!group_count[K]) continue;
      a.desc = (const PairDesc*)(d + kNaDescOff) + group_start[K];
      a.n_desc = group_count[K];
      a.ycap = batch_ycap(group_nmax[K]);
      launch_batch_k(K, st, a.n_desc, alpha, a, false);
    }
    HIPCK(ctx, hipGetLastError());
    HIPCK(ctx, hipEventRecord(ctx->ev[1], st));
    HIPCK(ctx, hipMemcpyAsync(h_out, d_score, (size_t)n_pairs * 12, hipMemcpyDeviceToHost, st));
    HIPCK(ctx, hipMemcpyAsync(h_out + 3 * n_pairs, d_err, 4, hipMemcpyDeviceToHost, st));
    HIPCK(ctx, hipStreamSynchronize(st));
    if (kernel_ms) HIPCK(ctx, hipEventElapsedTime(kernel_ms, ctx->ev[0], ctx->ev[1]));
    h_err = h_out[3 * n_pairs];
  }
  for (int64_t k = 0; k < n_pairs; ++k) {
    const bool none = len_x[k] == 0 || len_y[k] == 0 || !n_desc;  // no cell: the reference's zero borders, score 0
    score[k] = none ? 0 : h_out[k];
    end_i[k] = none ? 0 : h_out[n_pairs + k];
    end_j[k] = none ? 0 : h_out[2 * n_pairs + k];
  }
  if (h_err) return fail(ctx, SA_ERR_INVALID_CODE, "at least one pair holds a residue outside the scoring alphabet (its score is INT32_MIN)");
  return SA_OK;
}

struct Builtin {
  int8_t sub[2][32 * 32];
  uint8_t map[2][256];
  sa_scoring sc[2];
  Builtin() {
    sa_tables::parse_table(sa_tables::kNuc44Text, 15, sub[0]);
    sa_tables::parse_table(sa_tables::kBlosum62Text, 24, sub[1]);
    sa_tables::dna_idx256(map[0]);
    sa_tables::prot_idx256(map[1]);
    sc[0] = sa_scoring{-5.0, -1.0, sub[0], 15, map[0]};
    sc[1] = sa_scoring{-5.0, -1.0, sub[1], 24, map[1]};
  }
};

}  // namespace

extern "C" {

sa_ctx* sa_create(int device_id) {
  static const bool timing = getenv("SA_TIMING") != nullptr;
  const auto tp0 = std::chrono::steady_clock::now();
  auto lap = [&](const char* what) {
    if (timing) fprintf(stderr, "[sa_create] %-28s %8.3f ms since call\n", what, std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - tp0).count());
  };
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || device_id < 0 || device_id >= ndev) return nullptr;
  lap("hipGetDeviceCount (HIP init)");
  if (hipSetDevice(device_id) != hipSuccess) return nullptr;
  hipDeviceProp_t prop;
  if (hipGetDeviceProperties(&prop, device_id) != hipSuccess) return nullptr;
  lap("set device, properties");
  if (strncmp(prop.gcnArchName, "gfx950", 6) != 0) {
    fprintf(stderr, "seqalign_hip: device %d is %s; this library carries gfx950 code only\n", device_id, prop.gcnArchName);
    return nullptr;
  }
  sa_ctx* c = new sa_ctx;
  c->device = device_id;
  c->n_cu = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
  if (const char* e = getenv("SA_PERSIST")) c->persist = atoi(e);
  if (const char* e = getenv("SA_GROUP")) c->group = atoi(e);
  if (const char* e = getenv("SA_WPS")) c->wps = std::max(1, std::min(4, atoi(e)));
  if (hipStreamCreateWithFlags(&c->own_stream, hipStreamNonBlocking) != hipSuccess) {
    delete c;
    return nullptr;
  }
  c->stream = c->own_stream;
  for (auto& e : c->ev)
    if (hipEventCreate(&e) != hipSuccess) {
      delete c;
      return nullptr;
    }
  c->err.reserve(512);
  lap("stream, events");
  // sa_sw_batch_noalloc: the pinned staging block, so that the entry point itself never allocates (its device scratch is
  // a static array in a code object of its own, sa_noalloc.hip, whose address the entry point looks up on its first call
  // on this device: a process that never calls it never loads those 33 MiB).  A failure here only disables that entry point.
  void* hp = nullptr;
  if (hipHostMalloc(&hp, kNaHostBytes, hipHostMallocDefault) == hipSuccess)
    c->na_host = (unsigned char*)hp;
  else
    (void)hipGetLastError();
  lap("pinned staging block");
  return c;
}

void sa_destroy(sa_ctx* c) {
  if (!c) return;
  (void)hipSetDevice(c->device);
  sa_release_workspace(c);
  for (auto& e : c->ev)
    if (e) (void)hipEventDestroy(e);
  if (c->own_stream) (void)hipStreamDestroy(c->own_stream);
  if (c->copy_stream) (void)hipStreamDestroy(c->copy_stream);
  if (c->ev_chunk) (void)hipEventDestroy(c->ev_chunk);
  if (c->na_host) (void)hipHostFree(c->na_host);
  delete c;
}

const char* sa_last_error(const sa_ctx* c) { return c ? c->err.c_str() : "null context"; }

int sa_set_stream(sa_ctx* c, void* s) {
  if (!c) return SA_ERR_ARG;
  c->stream = s ? (hipStream_t)s : c->own_stream;
  return SA_OK;
}

int sa_set_tuning(sa_ctx* c, int K, int C) {
  if (!c) return SA_ERR_ARG;
  if (!(K == 0 || K == 1 || K == 2 || K == 4 || K == 8 || K == 16)) return fail(c, SA_ERR_ARG, "rows_per_lane must be 0,1,2,4,8 (or 16: LCS only)");
  if (C < 0 || C % 64 != 0 || C > 4096) return fail(c, SA_ERR_ARG, "tile_steps must be a multiple of 64 (<=4096)");
  c->force_K = K;
  c->force_C = C;
  return SA_OK;
}

int sa_set_option(sa_ctx* c, const char* name, int64_t value) {
  if (!c || !name) return SA_ERR_ARG;
  if (!strcmp(name, "persistent")) {
    c->persist = value != 0;
  } else if (!strcmp(name, "lcs_bits_waves_per_block")) {
    if (!(value == 0 || value == 1 || value == 2 || value == 4)) return fail(c, SA_ERR_ARG, "lcs_bits_waves_per_block must be 0, 1, 2 or 4");
    c->bits_wpb = (int)value;
  } else if (!strcmp(name, "lcs_bits")) {  // 0: LCS always through the cell-by-cell kernels
    c->lcs_bits = value != 0;
  } else if (!strcmp(name, "checkpoint_band")) {  // 0: checkpoint only when the codes do not fit
    if (value < 0) return fail(c, SA_ERR_ARG, "checkpoint_band must be >= 0");
    c->force_band = (int)value;
  } else if (!strcmp(name, "trace_budget_mb")) {  // 0: automatic
    if (value < 0) return fail(c, SA_ERR_ARG, "trace_budget_mb must be >= 0");
    c->trace_budget = (size_t)value << 20;
  } else if (!strcmp(name, "batch_trace_budget_mb")) {
    if (value < 1) return fail(c, SA_ERR_ARG, "batch_trace_budget_mb must be positive");
    c->batch_trace_budget = (size_t)value << 20;
  } else if (!strcmp(name, "max_workgroups")) {  // 0: as many as the GPU holds
    if (value < 0) return fail(c, SA_ERR_ARG, "max_workgroups must be >= 0");
    c->max_wg = (int)std::min<int64_t>(value, 1 << 20);
  } else if (!strcmp(name, "group")) {  // 1: fill_group where it applies (NW / SW, whole matrix); 0: fill_persistent
    c->group = value != 0;
  } else if (!strcmp(name, "waves_per_simd")) {
    if (value < 1 || value > 4) return fail(c, SA_ERR_ARG, "waves_per_simd must be 1..4");
    c->wps = (int)value;
  } else {
    return fail(c, SA_ERR_ARG, "unknown option %s", name);
  }
  return SA_OK;
}

int sa_release_workspace(sa_ctx* c) {
  if (!c) return SA_ERR_ARG;
  (void)hipSetDevice(c->device);
  for (Buf* b : {&c->xenc, &c->yenc, &c->prof, &c->sub, &c->colb, &c->bot, &c->state, &c->trace, &c->rowbest, &c->small,
                 &c->ops, &c->b_seq, &c->b_desc, &c->b_out, &c->lbuf, &c->dpbuf, &c->pbuf, &c->vlb, &c->vrb})
    release(*b);
  for (HostBuf* h : {&c->h_desc, &c->h_out}) {
    if (h->p) (void)hipHostFree(h->p);
    h->p = nullptr;
    h->cap = 0;
  }
  return SA_OK;
}

void sa_free_result(sa_result* r) {
  if (!r) return;
  free(r->ops);
  free(r->dp);
  r->ops = nullptr;
  r->dp = nullptr;
  r->n_ops = 0;
}

const sa_scoring* sa_builtin_scoring(int mode) {
  static const Builtin b;
  if (mode == SA_MODE_DNA) return &b.sc[0];
  if (mode == SA_MODE_PROTEIN) return &b.sc[1];
  return nullptr;
}

int sa_nw(sa_ctx* ctx, const uint8_t* x, int64_t m, const uint8_t* y, int64_t n, const sa_scoring* sc, uint32_t flags,
          sa_result* out) {
  return run_pair(ctx, KIND_NW, x, m, y, n, sc, flags, out);
}

int sa_sw(sa_ctx* ctx, const uint8_t* x, int64_t m, const uint8_t* y, int64_t n, const sa_scoring* sc, uint32_t flags,
          sa_result* out) {
  return run_pair(ctx, KIND_SW, x, m, y, n, sc, flags, out);
}

int sa_lcs(sa_ctx* ctx, const uint8_t* x, int64_t m, const uint8_t* y, int64_t n, uint32_t flags, sa_result* out) {
  return run_pair(ctx, KIND_LCS, x, m, y, n, nullptr, flags, out);
}

int sa_stripe_fill(sa_ctx* ctx, int kind, const uint8_t* x, int64_t m, const uint8_t* y_stripe, int64_t n_stripe,
                   const sa_scoring* sc, const sa_stripe_desc* desc, sa_result* out) {
  if (kind != SA_KIND_NW && kind != SA_KIND_LCS && kind != SA_KIND_SW) return fail(ctx, SA_ERR_ARG, "bad kind");
  return run_stripe_fill(ctx, kind, x, m, y_stripe, n_stripe, sc, desc, out);
}

int sa_stripe_traceback(sa_ctx* ctx, int64_t start_i, int64_t start_j, sa_result* out) {
  return run_stripe_traceback(ctx, start_i, start_j, out);
}

int sa_batch_set_device_pool(sa_ctx* ctx, const uint8_t* d_seqs, int64_t bytes) {
  if (!ctx) return SA_ERR_ARG;
  if (d_seqs && bytes <= 0) return fail(ctx, SA_ERR_ARG, "a device pool needs a positive size");
  // no hipPointerGetAttributes check: the pool may have been allocated through another HIP runtime instance in
  // this process (PyTorch ships its own), which this one cannot look up although the GPU address space is shared
  ctx->dev_pool = d_seqs;
  ctx->dev_pool_bytes = d_seqs ? bytes : 0;
  return SA_OK;
}

int sa_sw_batch(sa_ctx* ctx, const uint8_t* seqs, const int64_t* off_x, const int32_t* len_x, const int64_t* off_y,
                const int32_t* len_y, int64_t n_pairs, const sa_scoring* sc, int32_t* score, int32_t* end_i,
                int32_t* end_j, float* kernel_ms) {
  return run_batch(ctx, seqs, off_x, len_x, off_y, len_y, n_pairs, sc, score, end_i, end_j, kernel_ms);
}

int sa_sw_batch_align(sa_ctx* ctx, const uint8_t* seqs, const int64_t* off_x, const int32_t* len_x, const int64_t* off_y,
                      const int32_t* len_y, int64_t n_pairs, const sa_scoring* sc, int32_t* score, int32_t* end_i,
                      int32_t* end_j, int32_t* start_i, int32_t* start_j, int64_t* ops_off, int32_t* n_ops,
                      uint8_t** ops_pool, int64_t* ops_pool_bytes, float* kernel_ms) {
  const BatchAlign al{KIND_SW, start_i, start_j, ops_off, n_ops, ops_pool, ops_pool_bytes};
  return run_batch(ctx, seqs, off_x, len_x, off_y, len_y, n_pairs, sc, score, end_i, end_j, kernel_ms, &al);
}

int sa_sw_window_batch(sa_ctx* ctx, const uint8_t* seqs, const int64_t* off_x, const int32_t* len_x, const int64_t* off_y,
                       const int32_t* len_y, int64_t n_pairs, const sa_scoring* sc, int32_t* score, int32_t* end_i,
                       int32_t* end_j, int32_t* start_i, int32_t* start_j, int32_t* aligned_len, float* kernel_ms) {
  BatchAlign al{KIND_SW, start_i, start_j, nullptr, aligned_len, nullptr, nullptr};
  al.diag_only = 1;
  return run_batch(ctx, seqs, off_x, len_x, off_y, len_y, n_pairs, sc, score, end_i, end_j, kernel_ms, &al);
}

int sa_lcs_batch_align(sa_ctx* ctx, const uint8_t* seqs, const int64_t* off_x, const int32_t* len_x, const int64_t* off_y,
                       const int32_t* len_y, int64_t n_pairs, int32_t* score, int64_t* ops_off, int32_t* n_ops,
                       uint8_t** ops_pool, int64_t* ops_pool_bytes, float* kernel_ms) {
  if (!ctx) return SA_ERR_ARG;
  if (n_pairs < 0 || (n_pairs > 0 && !score)) return fail(ctx, SA_ERR_ARG, "bad batch arguments");
  std::vector<int32_t> ei((size_t)std::max<int64_t>(n_pairs, 1)), ej((size_t)std::max<int64_t>(n_pairs, 1));
  const BatchAlign al{KIND_LCS, nullptr, nullptr, ops_off, n_ops, ops_pool, ops_pool_bytes};
  return run_batch(ctx, seqs, off_x, len_x, off_y, len_y, n_pairs, nullptr, score, ei.data(), ej.data(), kernel_ms, &al);
}

int sa_nw_batch_align(sa_ctx* ctx, const uint8_t* seqs, const int64_t* off_x, const int32_t* len_x, const int64_t* off_y,
                      const int32_t* len_y, int64_t n_pairs, const sa_scoring* sc, int32_t* score, int64_t* ops_off,
                      int32_t* n_ops, uint8_t** ops_pool, int64_t* ops_pool_bytes, float* kernel_ms) {
  if (!ctx) return SA_ERR_ARG;
  if (n_pairs < 0 || (n_pairs > 0 && !score)) return fail(ctx, SA_ERR_ARG, "bad batch arguments");
  std::vector<int32_t> ei((size_t)std::max<int64_t>(n_pairs, 1)), ej((size_t)std::max<int64_t>(n_pairs, 1));
  const BatchAlign al{KIND_NW, nullptr, nullptr, ops_off, n_ops, ops_pool, ops_pool_bytes};
  return run_batch(ctx, seqs, off_x, len_x, off_y, len_y, n_pairs, sc, score, ei.data(), ej.data(), kernel_ms, &al);
}

void sa_free_pool(uint8_t* pool) { pool_release(pool); }

int sa_sw_batch_noalloc(sa_ctx* ctx, const uint8_t* seqs, const int64_t* off_x, const int32_t* len_x, const int64_t* off_y,
                        const int32_t* len_y, int64_t n_pairs, const sa_scoring* sc, int32_t* score, int32_t* end_i,
                        int32_t* end_j, float* kernel_ms) {
  return run_batch_noalloc(ctx, seqs, off_x, len_x, off_y, len_y, n_pairs, sc, score, end_i, end_j, kernel_ms);
}

}  // extern "C"
