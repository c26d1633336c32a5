/*
 * seqalign_hip.h -- C-ABI of libseqalign_hip.so, the MI355X (gfx950) DP alignment engine.
 *
 * Drop-in boundary for the reference's three hot-path entry points.  The reference
 * (bibymaths/SequenceAligner) has no FFI; its seam is three free functions inside
 * src/main.cpp that take two residue strings plus the globals GAP_OPEN/GAP_EXTEND and
 * write result files.  Each sa_* entry point below replaces the fill + traceback part
 * of one of them and returns what the reference's writers need (score, end cell and
 * the traceback op list); see INTEGRATION.md for the binding a maintainer would add.
 *
 * Conventions
 *   - plain C, no exceptions across the boundary; every call returns SA_OK (0) or a
 *     negative SA_ERR_* code, sa_last_error(ctx) gives the message
 *   - inputs are caller-owned host buffers of raw residue bytes (exactly what the
 *     reference's processFasta produces, main.cpp:290-308); they may be freed as soon
 *     as the call returns
 *   - outputs inside sa_result are library-owned host buffers, released by
 *     sa_free_result()
 *   - one sa_ctx per GPU; a ctx is not thread-safe; device workspaces are cached in
 *     the ctx between calls (the first call at a given size allocates)
 *   - x = query = rows (length m), y = target = columns (length n); matrix cells are
 *     1-based (i, j) exactly as in the reference
 *   - ops[]: one byte per traceback move in traceback order (end cell -> start cell):
 *       0 = diagonal  x[i-1] over y[j-1]      (reference pointer 'M' / LCS 'D')
 *       1 = left      '-'    over y[j-1]      (reference pointer 'E','e' / LCS 'L')
 *       2 = up        x[i-1] over '-'         (reference pointer 'F','f' / LCS 'U')
 */
#ifndef SEQALIGN_HIP_H
#define SEQALIGN_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SA_OK 0
#define SA_ERR_INVALID_CODE (-2) /* residue outside the scoring alphabet: the reference throws
                                    "Invalid DNA code: x,y" / "Invalid protein code: x,y"
                                    (main.cpp:120-121, 134-136); bad_x/bad_y carry x,y */
#define SA_ERR_NOMEM (-3)
#define SA_ERR_HIP (-4)
#define SA_ERR_ARG (-5)
#define SA_ERR_UNSUPPORTED (-6)

#define SA_MODE_DNA 0     /* EDNAFULL via char2idx, main.cpp:57-78, 117-123 */
#define SA_MODE_PROTEIN 1 /* EBLOSUM62 via prot_idx, main.cpp:81-109, 131-138 */

/* flags */
#define SA_FLAG_KEEP_DP 1u  /* also return the full (m+1)x(n+1) int32 S (or LCS L) matrix:
                               what --binary/--txt dump (main.cpp:825-831, 1364-1368, 1686-1691);
                               limited to 1.5e9 cells */
#define SA_FLAG_NO_TRACEBACK 2u /* score (and SW end cell) only */

#define SA_KIND_NW 0
#define SA_KIND_SW 1
#define SA_KIND_LCS 2

typedef struct sa_ctx sa_ctx;

/* Scoring = the reference's ScoreFn (main.cpp:38) + GAP_OPEN/GAP_EXTEND (main.cpp:41-43). */
typedef struct {
  double gap_open;       /* negative; added as int+double and truncated per op (main.cpp:603) */
  double gap_extend;
  const int8_t* sub;     /* alpha x alpha substitution scores, row-major, sub[ix*alpha+iy] */
  int32_t alpha;         /* <= 32 (the many-pair entry points: <= 31, one symbol is their padding) */
  const uint8_t* idx256; /* residue byte -> table index, 255 = invalid (the reference throws) */
} sa_scoring;

typedef struct {
  int64_t score;            /* NW: S[m][n] (main.cpp:766); SW: best S (main.cpp:1124); LCS: L[m][n] */
  int64_t end_i, end_j;     /* cell where the traceback starts: NW/LCS (m,n); SW first row-major max */
  int64_t start_i, start_j; /* cell where it stopped: NW/LCS (0,0); SW the 'X' cell (main.cpp:1233) */
  uint8_t* ops;             /* n_ops traceback moves, end -> start */
  int64_t n_ops;
  int32_t* dp;              /* SA_FLAG_KEEP_DP: (m+1)*(n+1) row-major, else NULL */
  unsigned char bad_x, bad_y; /* SA_ERR_INVALID_CODE: the pair the reference would report */
  /* device-side timing of the last call, milliseconds (HIP events on the ctx stream) */
  float fill_ms;            /* H2D of sequences + profile build + fill kernels */
  float traceback_ms;       /* traceback kernel + D2H of the op list */
  float kernel_fill_ms;     /* fill kernels only */
  int32_t n_launches;       /* fill kernel launches */
  int32_t rows_per_lane;    /* K chosen for this call */
  int32_t tile_steps;       /* C chosen for this call */
  /* cells computed by the fill launches counted in kernel_fill_ms, by the work done per cell: with direction
   * codes (the normal fill; the bands a checkpointed traceback recomputes) / without (score-only calls, the
   * checkpoint pass).  A checkpointed call has cells_notrace = m*n and cells_trace = the recomputed cells;
   * its band launches are counted in fill_ms / kernel_fill_ms, not in traceback_ms. */
  int64_t cells_trace, cells_notrace;
  /* sa_stripe_fill with stream_in: milliseconds an average wave of the persistent launch spent waiting for rows of
   * the left border to arrive (summed over its strips; 0 on the first stripe and on the in-line schedule) */
  float border_wait_ms;
} sa_result;

/* ---- lifetime ---- */
sa_ctx* sa_create(int device_id);           /* NULL on failure (no GPU / no gfx950 code object) */
void sa_destroy(sa_ctx* ctx);
const char* sa_last_error(const sa_ctx* ctx);
int sa_set_stream(sa_ctx* ctx, void* hip_stream); /* hipStream_t; NULL = the ctx's own stream */
void sa_free_result(sa_result* r);
/* frees the ctx's cached device workspaces (they are re-allocated on demand) */
int sa_release_workspace(sa_ctx* ctx);
/* tuning overrides, 0 = automatic: rows per lane K in {1,2,4,8} (16: the LCS cell kernel over small alphabets only,
 * 8 elsewhere), tile steps C multiple of 64.  A forced K also keeps sa_lcs on the cell-by-cell kernels. */
int sa_set_tuning(sa_ctx* ctx, int rows_per_lane, int tile_steps);
/* engine options: "persistent" 0/1 (fill as one persistent strip-pipelined launch instead of one
 * launch per tile diagonal; same results), "waves_per_simd" 1..4 (grid of the persistent launch),
 * "batch_trace_budget_mb" (sa_sw_batch_align: device bytes of direction codes per chunk),
 * "trace_budget_mb" (single pair: bytes of 2-bit codes a call may hold, 0 = 88 % of the device; a pair
 * whose codes exceed it is filled with checkpoints -- the bottom row of every strip -- and the
 * traceback recomputes the strips it crosses, a band at a time: same results, about twice the fill
 * work, and e.g. 2 Mb x 2 Mb on one GPU), "checkpoint_band" N > 0 (always take that path, N strips per band),
 * "lcs_bits" 0/1 (default 1: sa_lcs on a whole pair over at most 255 distinct bytes runs as bit-vector
 * arithmetic -- 32 rows per lane, sa_result.rows_per_lane = 32, one bit of traceback record per cell; same
 * results as the cell-by-cell kernels, which still serve dumps, pairs that hold all 256 byte values and stripes whose
 * border chunks are not whole 2048-row strips or that run the in-line schedule),
 * "group" 0/1 (default 0: NW / SW fills of a whole matrix run fill_group -- four consecutive strips per workgroup, their
 * bottom rows handed down through LDS -- instead of fill_persistent; same results, measured no faster: DESIGN.md 3.1c),
 * "lcs_bits_waves_per_block" 0 (automatic) / 1 / 2 / 4 (measurements: how many strips share a workgroup there),
 * "max_workgroups" N (0 = as many as the GPU holds: a persistent launch uses at most N workgroups of four waves;
 * tests use it to make small inputs take several rounds of strips per wave; same results) */
int sa_set_option(sa_ctx* ctx, const char* name, int64_t value);

/* The reference's two built-in scoring schemes with default penalties -5/-1
 * (main.cpp:41-43): EDNAFULL indexed by char2idx (including its index-order quirk) and
 * EBLOSUM62 indexed by prot_idx.  The returned struct is static; copy it to change gaps. */
const sa_scoring* sa_builtin_scoring(int mode);

/* ---- hot path ---- */

/* replaces the fill + traceback of globalalign(), main.cpp:691-818 */
int sa_nw(sa_ctx* ctx, const uint8_t* x, int64_t m, const uint8_t* y, int64_t n,
          const sa_scoring* sc, uint32_t flags, sa_result* out);

/* replaces both fill passes + traceback of localalign(), main.cpp:1099-1128, 1182-1265 */
int sa_sw(sa_ctx* ctx, const uint8_t* x, int64_t m, const uint8_t* y, int64_t n,
          const sa_scoring* sc, uint32_t flags, sa_result* out);

/* replaces the fill + traceback of lcs(), main.cpp:1575-1640 (raw byte equality) */
int sa_lcs(sa_ctx* ctx, const uint8_t* x, int64_t m, const uint8_t* y, int64_t n,
           uint32_t flags, sa_result* out);

/* Many independent pairs, score + end cell only: the first fill pass of localalign()
 * (main.cpp:1099-1128) per pair.  Pair k is x = seqs[off_x[k] .. +len_x[k]),
 * y = seqs[off_y[k] .. +len_y[k]).  Outputs are caller-owned arrays of n_pairs.
 * A pair with an invalid residue gets score = INT32_MIN and the call returns
 * SA_ERR_INVALID_CODE after finishing the other pairs.
 * Penalties: non-positive, and -- for this and the other many-pair entry points below -- small enough for the batch
 * kernels' 4x-scaled int32 arithmetic over their 1024 x 4095 limits: 5121 x max(|gap_open| + 1, |gap_extend| + 1,
 * max |sub|) <= 1.2e8, i.e. penalties down to -23 431; beyond that the call returns SA_ERR_ARG (sa_sw / sa_nw
 * take such a pair, one at a time). */
int sa_sw_batch(sa_ctx* ctx, const uint8_t* seqs, const int64_t* off_x, const int32_t* len_x,
                const int64_t* off_y, const int32_t* len_y, int64_t n_pairs,
                const sa_scoring* sc, int32_t* score, int32_t* end_i, int32_t* end_j,
                float* kernel_ms);

/* sa_sw_batch for callers that cannot allocate (tight serving loops, harnesses that forbid hipMalloc, code that runs
 * while another thread captures a HIP graph).  NO COUNTERPART IN THE REFERENCE.  Same inputs, outputs, kernels and
 * results as sa_sw_batch, but the call allocates nothing: no hipMalloc / hipFree / hipHostMalloc and no host
 * containers.  It is NOT itself capturable: it copies the caller's pageable buffers in, synchronises the ctx stream
 * and reads the results back before it returns.  Its device
 * scratch (sequence staging, pair descriptors, per-pair results, error flag, scoring tables) is one file-scope
 * `static __device__` array of SA_NOALLOC_MAX_SEQ_BYTES + 48 + 12 bytes per pair of SA_NOALLOC_MAX_PAIRS + 2 KiB
 * (33 MiB) that every GPU this entry point has run on holds until the process ends (the array is a code object of its
 * own, loaded by the entry point's first call on a device, which also looks its address up, once per device);
 * sa_create() pins the 1 MB host staging block the call uses.  Limits, rejected with SA_ERR_ARG / SA_ERR_UNSUPPORTED and a message:
 * n_pairs <= SA_NOALLOC_MAX_PAIRS; unless a device pool is set (sa_batch_set_device_pool: the sequences are then read
 * in place), every pair must end within the first SA_NOALLOC_MAX_SEQ_BYTES of `seqs`; every pair must fit the batch
 * kernels (len_x <= 1024, len_y <= 3895: sa_sw_batch's single-pair fallback allocates and is not taken).
 * All calls on one GPU (from any ctx) share that GPU's array, which keeps the last call's contents (the call clears
 * what must start at zero).  Calls of one process that overlap in time on one GPU do not race: the array has a busy
 * flag per device and the second call returns SA_ERR_ARG ("scratch in use") at once; calls from different PROCESSES
 * each have their own copy of the array. */
#define SA_NOALLOC_MAX_PAIRS 16384
#define SA_NOALLOC_MAX_SEQ_BYTES (32 << 20)
int sa_sw_batch_noalloc(sa_ctx* ctx, const uint8_t* seqs, const int64_t* off_x, const int32_t* len_x,
                        const int64_t* off_y, const int32_t* len_y, int64_t n_pairs,
                        const sa_scoring* sc, int32_t* score, int32_t* end_i, int32_t* end_j,
                        float* kernel_ms);

/* Sequence pool already in device memory (a caller that produced or staged the sequences on the GPU, e.g. a
 * torch tensor): while a pool is set, every batch entry point below reads pair k at d_seqs + off_x[k] /
 * off_y[k], ignores its `seqs` argument (NULL is fine) and copies nothing host -> device but the pair
 * descriptors.  The memory stays the caller's and must stay valid and unchanged during those calls; work queued
 * on other streams that writes it must have completed.  d_seqs = NULL returns to host pools.  No counterpart in
 * the reference (extension of the config-5 workload). */
int sa_batch_set_device_pool(sa_ctx* ctx, const uint8_t* d_seqs, int64_t bytes);

/* Many independent pairs with their full local alignments: per pair what sa_sw returns (score, end
 * cell, the 'X' cell where the traceback stopped, the op list end -> start).  The reference has no
 * batch mode -- it aligns one pair per process (processFasta reads one record, main.cpp:290-308) --
 * so this is localalign()'s fill + traceback (main.cpp:1099-1128, 1182-1265) applied to every pair
 * (SURVEY.md 8f-1).  The ops of pair k are ops_pool[ops_off[k] .. ops_off[k] + n_ops[k]); the library
 * assigns ops_off (slots of len_x + len_y rounded up to 4, in its processing order, not the caller's).  *ops_pool is library-owned,
 * page-locked host memory, released with sa_free_pool() (which keeps one buffer for the next call,
 * so a loop of calls pins memory once).  Device memory for the direction codes is bounded by
 * processing the pairs in chunks (sa_set_option "batch_trace_budget_mb", default 32768).
 * Invalid residues: as sa_sw_batch (score INT32_MIN, no ops, SA_ERR_INVALID_CODE after the rest). */
int sa_sw_batch_align(sa_ctx* ctx, const uint8_t* seqs, const int64_t* off_x, const int32_t* len_x,
                      const int64_t* off_y, const int32_t* len_y, int64_t n_pairs, const sa_scoring* sc,
                      int32_t* score, int32_t* end_i, int32_t* end_j, int32_t* start_i, int32_t* start_j,
                      int64_t* ops_off, int32_t* n_ops, uint8_t** ops_pool, int64_t* ops_pool_bytes,
                      float* kernel_ms);
void sa_free_pool(uint8_t* pool);

/* Many independent pairs, global alignment: globalalign()'s fill + traceback (main.cpp:691-818) applied
 * to every pair -- the shape of the seed_aligner's per-segment align_segment_globally calls
 * (main_fmindex.cpp:1049-1143, one per anchored segment).  score[k] = S[m][n]; the op list of pair k
 * (end (m, n) -> start (0, 0), border walks included) sits in the pool as for sa_sw_batch_align.
 * A pair with an empty side is the reference's border walk.  Non-positive penalties (fractional ones included:
 * truncated toward zero per operation like the reference's int + double sums). */
int sa_nw_batch_align(sa_ctx* ctx, const uint8_t* seqs, const int64_t* off_x, const int32_t* len_x,
                      const int64_t* off_y, const int32_t* len_y, int64_t n_pairs, const sa_scoring* sc,
                      int32_t* score, int64_t* ops_off, int32_t* n_ops, uint8_t** ops_pool,
                      int64_t* ops_pool_bytes, float* kernel_ms);

/* Many independent pairs, LCS: lcs()'s fill + traceback (main.cpp:1575-1640) per pair (the shape of the
 * seed_aligner's compute_lcs_for_segment calls, main_fmindex.cpp:1145-1207).  score[k] = L[m][n], op
 * list as above (diagonal moves are the LCS characters). */
int sa_lcs_batch_align(sa_ctx* ctx, const uint8_t* seqs, const int64_t* off_x, const int32_t* len_x,
                       const int64_t* off_y, const int32_t* len_y, int64_t n_pairs, int32_t* score,
                       int64_t* ops_off, int32_t* n_ops, uint8_t** ops_pool, int64_t* ops_pool_bytes,
                       float* kernel_ms);

/* Many windows, the seed_aligner's windowed local alignment: perform_sw_in_window (main_fmindex.cpp:938-1047)
 * per pair, x = the query, y = the reference window.  Its fill is localalign()'s (same floors, same first
 * strict row-major maximum); its traceback is a three-state machine that, as written, starts in the match
 * state at the best cell and re-enters the match state after each diagonal step, so with a negative
 * gap-open penalty (required here; SA_ERR_UNSUPPORTED otherwise) the alignment it returns is the
 * UNGAPPED diagonal through the best cell for as long as S > 0: x[start_i .. end_i) against
 * y[start_j .. end_j), aligned_len = end_i - start_i = end_j - start_j (0 and all-zero cells when the
 * score is 0).  The reference's query_start / target_start are q_offset + start_i / t_offset + start_j
 * and its *_end are q_offset + end_i - 1 / t_offset + end_j - 1.  Windows are limited to the batch
 * kernels' 1024 x 3895 (SA_ERR_UNSUPPORTED beyond).  Invalid residues as sa_sw_batch. */
int sa_sw_window_batch(sa_ctx* ctx, const uint8_t* seqs, const int64_t* off_x, const int32_t* len_x,
                       const int64_t* off_y, const int32_t* len_y, int64_t n_pairs, const sa_scoring* sc,
                       int32_t* score, int32_t* end_i, int32_t* end_j, int32_t* start_i, int32_t* start_j,
                       int32_t* aligned_len, float* kernel_ms);

/* ---- one long pair over several GPUs: column stripes ----
 *
 * Replaces the reference's MPI row-block chain (main.cpp:1050-1147, 1151-1177), which is serial;
 * here GPU g owns the columns [col_begin, col_begin+n_stripe) of the same matrix and every
 * process (one per GPU) calls sa_stripe_fill on its own stripe.  The only data that crosses
 * GPUs is the stripe's border column: for every matrix row i = 1..m the pair (H[i][c], E[i][c])
 * at the stripe's last column, 8 bytes per row, laid out as int32[m+1][2] (entry 0 is written by
 * the library).  The library does not move it: while it enqueues the fill it calls on_event
 *   SA_EV_NEED_LEFT(k)   before the first launch that reads rows of chunk k of `left`
 *   SA_EV_RIGHT_READY(k) after the launch that completes rows of chunk k of `right`
 * (chunk k = matrix rows k*chunk_rows+1 .. min(m,(k+1)*chunk_rows), both in ascending k) and the
 * caller enqueues the recv / send of that chunk on the ctx's stream (RCCL send/recv, a peer
 * copy, ...).  NW and LCS: the score is valid on the last stripe.  SW (reference analogue: the
 * cross-rank arg-max of main.cpp:1151-1177): out->score / end_i / end_j are the stripe's own first
 * strict row-major maximum (end_j local to the stripe); the caller picks the winner over all stripes
 * by (score descending, end_i ascending, global column ascending) and starts the traceback there;
 * the walk ends at the 'X' cell (out->start_j >= 1) or reaches the stripe's left edge
 * (out->start_j == 0 with start_i >= 1 on a stripe that is not the first: continue on the left
 * neighbour at its last column, same row).
 * sa_stripe_traceback walks this stripe's codes from (start_i, start_j) (j local to the stripe,
 * 1-based) and stops at the stripe's left edge unless it is the first stripe; out->start_i is
 * the row to continue from on the stripe to the left (at its last column).  The op lists of the
 * stripes, concatenated right to left, are the op list sa_nw / sa_lcs return for the whole pair.
 */
#define SA_EV_NEED_LEFT 0
#define SA_EV_RIGHT_READY 1
typedef int (*sa_stripe_event_fn)(void* user, int event, int64_t chunk); /* return 0 to continue */

typedef struct {
  int64_t n_total;       /* columns of the whole matrix */
  int64_t col_begin;     /* first column of this stripe, 0-based */
  int32_t is_first, is_last;
  const int32_t* left;   /* device int32[m+1][2], NULL iff is_first */
  int32_t* right;        /* device int32[m+1][2], NULL iff is_last */
  int64_t chunk_rows;
  sa_stripe_event_fn on_event;
  void* user;
  /* Optional (both NULL = the launch-per-diagonal schedule above, transfers on the ctx stream):
   * hipStream_t of the caller's receive and send streams.  When given (and the configuration
   * allows the persistent pipeline: no matrix dump, chunk_rows a multiple of 512) the stripe is ONE
   * persistent launch on the ctx stream and the border crosses while it runs: the library calls
   * SA_EV_NEED_LEFT(k) for every chunk right away -- the caller enqueues the receive of chunk k on
   * stream_in, the library follows it with a one-wave kernel that tells the running fill the rows
   * are there -- and SA_EV_RIGHT_READY(k) after enqueueing, on stream_out, a one-wave kernel that
   * returns when the fill has written chunk k of `right`; the caller enqueues the send of chunk k
   * on stream_out.  Neither stream may be the ctx stream. */
  void* stream_in;
  void* stream_out;
} sa_stripe_desc;

int sa_stripe_fill(sa_ctx* ctx, int kind, const uint8_t* x, int64_t m, const uint8_t* y_stripe, int64_t n_stripe,
                   const sa_scoring* sc, const sa_stripe_desc* desc, sa_result* out);
int sa_stripe_traceback(sa_ctx* ctx, int64_t start_i, int64_t start_j, sa_result* out);

#ifdef __cplusplus
}
#endif
#endif /* SEQALIGN_HIP_H */
