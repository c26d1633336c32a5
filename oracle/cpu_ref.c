/*
 * oracle/cpu_ref.c -- scalar CPU restatement of the reference DP hot path.
 *
 * TEST INFRASTRUCTURE ONLY (see cpu_ref.h).  Written from the behaviour of
 * /root/reference/src/main.cpp; every routine cites the lines it follows.
 * Parity status: PINNED against the compiled reference through tests/golden/.
 *
 * Arithmetic note (reference quirk Q3): GAP_OPEN / GAP_EXTEND are `double`
 * (main.cpp:41-43) and every "score + gap" is evaluated as int + double and
 * assigned back to an int, i.e. truncated toward zero per operation
 * (main.cpp:603-604, 625-634, 958-964).  add_gap() reproduces exactly that.
 */
#include "cpu_ref.h"

#include <limits.h>
#include <pthread.h>
#include <stdlib.h>
#include <string.h>

#define NEG_INF (INT_MIN / 2) /* main.cpp:703-704, 619 */

/* ---- substitution tables ------------------------------------------------
 * NUC.4.4 in the row/column order the reference header stores it
 * (EDNAFULL.h:11-26: A T G C S W R Y K M B V H D N) ... */
static const signed char k_nuc44[15][15] = {
    /*A*/ {5, -4, -4, -4, -4, 1, 1, -4, -4, 1, -4, -1, -1, -1, -2},
    /*T*/ {-4, 5, -4, -4, -4, 1, -4, 1, 1, -4, -1, -4, -1, -1, -2},
    /*G*/ {-4, -4, 5, -4, 1, -4, 1, -4, 1, -4, -1, -1, -4, -1, -2},
    /*C*/ {-4, -4, -4, 5, 1, -4, -4, 1, -4, 1, -1, -1, -1, -4, -2},
    /*S*/ {-4, -4, 1, 1, -1, -4, -2, -2, -2, -2, -1, -1, -3, -3, -1},
    /*W*/ {1, 1, -4, -4, -4, -1, -2, -2, -2, -2, -3, -3, -1, -1, -1},
    /*R*/ {1, -4, 1, -4, -2, -2, -1, -4, -2, -2, -3, -1, -3, -1, -1},
    /*Y*/ {-4, 1, -4, 1, -2, -2, -4, -1, -2, -2, -1, -3, -1, -3, -1},
    /*K*/ {-4, 1, 1, -4, -2, -2, -2, -2, -1, -4, -1, -3, -3, -1, -1},
    /*M*/ {1, -4, -4, 1, -2, -2, -2, -2, -4, -1, -3, -1, -1, -3, -1},
    /*B*/ {-4, -1, -1, -1, -1, -3, -3, -1, -1, -3, -1, -2, -2, -2, -1},
    /*V*/ {-1, -4, -1, -1, -1, -3, -1, -3, -3, -1, -2, -1, -2, -2, -1},
    /*H*/ {-1, -1, -4, -1, -3, -1, -3, -1, -3, -1, -2, -2, -1, -2, -1},
    /*D*/ {-1, -1, -1, -4, -3, -1, -1, -3, -1, -3, -2, -2, -2, -1, -1},
    /*N*/ {-2, -2, -2, -2, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1}};

/* BLOSUM62, order A R N D C Q E G H I L K M F P S T W Y V B Z X * (EBLOSUM62.h:10-34) */
static const signed char k_blosum62[24][24] = {
    {4, -1, -2, -2, 0, -1, -1, 0, -2, -1, -1, -1, -1, -2, -1, 1, 0, -3, -2, 0, -2, -1, 0, -4},
    {-1, 5, 0, -2, -3, 1, 0, -2, 0, -3, -2, 2, -1, -3, -2, -1, -1, -3, -2, -3, -1, 0, -1, -4},
    {-2, 0, 6, 1, -3, 0, 0, 0, 1, -3, -3, 0, -2, -3, -2, 1, 0, -4, -2, -3, 3, 0, -1, -4},
    {-2, -2, 1, 6, -3, 0, 2, -1, -1, -3, -4, -1, -3, -3, -1, 0, -1, -4, -3, -3, 4, 1, -1, -4},
    {0, -3, -3, -3, 9, -3, -4, -3, -3, -1, -1, -3, -1, -2, -3, -1, -1, -2, -2, -1, -3, -3, -2, -4},
    {-1, 1, 0, 0, -3, 5, 2, -2, 0, -3, -2, 1, 0, -3, -1, 0, -1, -2, -1, -2, 0, 3, -1, -4},
    {-1, 0, 0, 2, -4, 2, 5, -2, 0, -3, -3, 1, -2, -3, -1, 0, -1, -3, -2, -2, 1, 4, -1, -4},
    {0, -2, 0, -1, -3, -2, -2, 6, -2, -4, -4, -2, -3, -3, -2, 0, -2, -2, -3, -3, -1, -2, -1, -4},
    {-2, 0, 1, -1, -3, 0, 0, -2, 8, -3, -3, -1, -2, -1, -2, -1, -2, -2, 2, -3, 0, 0, -1, -4},
    {-1, -3, -3, -3, -1, -3, -3, -4, -3, 4, 2, -3, 1, 0, -3, -2, -1, -3, -1, 3, -3, -3, -1, -4},
    {-1, -2, -3, -4, -1, -2, -3, -4, -3, 2, 4, -2, 2, 0, -3, -2, -1, -2, -1, 1, -4, -3, -1, -4},
    {-1, 2, 0, -1, -3, 1, 1, -2, -1, -3, -2, 5, -1, -3, -1, 0, -1, -3, -2, -2, 0, 1, -1, -4},
    {-1, -1, -2, -3, -1, 0, -2, -3, -2, 1, 2, -1, 5, 0, -2, -1, -1, -1, -1, 1, -3, -1, -1, -4},
    {-2, -3, -3, -3, -2, -3, -3, -3, -1, 0, 0, -3, 0, 6, -4, -2, -2, 1, 3, -1, -3, -3, -1, -4},
    {-1, -2, -2, -1, -3, -1, -1, -2, -2, -3, -3, -1, -2, -4, 7, -1, -1, -4, -3, -2, -2, -1, -2, -4},
    {1, -1, 1, 0, -1, 0, 0, 0, -1, -2, -2, 0, -1, -2, -1, 4, 1, -3, -2, -2, 0, 0, 0, -4},
    {0, -1, 0, -1, -1, -1, -1, -2, -2, -1, -1, -1, -1, -2, -1, 1, 5, -2, -2, 0, -1, -1, 0, -4},
    {-3, -3, -4, -4, -2, -2, -3, -2, -2, -3, -2, -3, -1, 1, -4, -3, -2, 11, 2, -3, -4, -3, -2, -4},
    {-2, -2, -2, -3, -2, -1, -2, -3, 2, -1, -1, -2, -1, 3, -3, -2, -2, 2, 7, -1, -3, -2, -1, -4},
    {0, -3, -3, -3, -1, -2, -2, -3, -3, 3, 1, -2, 1, -1, -2, -2, 0, -3, -1, 4, -3, -2, -1, -4},
    {-2, -1, 3, 4, -3, 0, 1, -1, 0, -3, -4, 0, -3, -3, -2, 0, -1, -4, -3, -3, 4, 1, -1, -4},
    {-1, 0, 0, 1, -3, 3, 4, -2, 0, -3, -3, 1, -1, -3, -1, 0, -1, -3, -2, -2, 1, 4, -1, -4},
    {0, -1, -1, -1, -2, -1, -1, -1, -1, -1, -1, -1, -1, -1, -2, 0, 0, -2, -1, -1, -1, -1, -1, -4},
    {-4, -4, -4, -4, -4, -4, -4, -4, -4, -4, -4, -4, -4, -4, -4, -4, -4, -4, -4, -4, -4, -4, -4, 1}};

/* residue -> table index; 255 = the reference throws (main.cpp:57-109).
 * NOTE (quirk Q1): the DNA map's order A C G T R Y S W K M B D H V N is NOT
 * the order of the NUC.4.4 rows above; the reference indexes the table with
 * it anyway (main.cpp:122) and so does this oracle. */
static unsigned char g_dna_idx[256], g_prot_idx[256];
static pthread_once_t g_once = PTHREAD_ONCE_INIT;
static void init_maps(void) {
  memset(g_dna_idx, 255, 256);
  memset(g_prot_idx, 255, 256);
  const char* d = "ACGTRYSWKMBDHVN";
  for (int i = 0; d[i]; ++i) g_dna_idx[(unsigned char)d[i]] = (unsigned char)i;
  g_dna_idx['U'] = 3;  /* main.cpp:64 */
  g_dna_idx['X'] = 14; /* main.cpp:76 */
  const char* p = "ARNDCQEGHILKMFPSTWYVBZX*";
  for (int i = 0; p[i]; ++i) g_prot_idx[(unsigned char)p[i]] = (unsigned char)i;
}

/* mode ORC_MODE_CUSTOM: a caller-supplied 256 x 256 table (INT16_MIN = invalid pair), used by the
 * tests of other scoring schemes (the seed_aligner's tables, main_fmindex.cpp:25-100) */
static int16_t g_custom[256 * 256];
void orc_set_custom_table(const int16_t* t) { memcpy(g_custom, t, sizeof g_custom); }

int orc_sigma(int mode, unsigned char x, unsigned char y, int* ok) {
  if (mode == ORC_MODE_CUSTOM) {
    const int16_t v = g_custom[(unsigned)x * 256u + y];
    *ok = v != INT16_MIN;
    return *ok ? v : 0;
  }
  pthread_once(&g_once, init_maps);
  const unsigned char* map = mode == ORC_MODE_DNA ? g_dna_idx : g_prot_idx;
  unsigned ix = map[x], iy = map[y];
  if (ix == 255 || iy == 255) {
    *ok = 0;
    return 0;
  }
  *ok = 1;
  return mode == ORC_MODE_DNA ? k_nuc44[ix][iy] : k_blosum62[ix][iy];
}

typedef struct {
  int mode;
  double go, ge;
  int bad;
  unsigned char bad_x, bad_y;
} scoring;

static inline int sigma(scoring* sc, unsigned char x, unsigned char y) {
  int ok;
  int v = orc_sigma(sc->mode, x, y, &ok);
  if (!ok && !sc->bad) {
    sc->bad = 1;
    sc->bad_x = x;
    sc->bad_y = y;
  }
  return v;
}

/* int + double, truncated back to int: main.cpp:603-604 and friends */
static inline int add_gap(int v, double g) { return (int)((double)v + g); }
static inline int imax(int a, int b) { return a > b ? a : b; }

static void ops_push(uint8_t** ops, int64_t* n, int64_t* cap, uint8_t op) {
  if (*n == *cap) {
    *cap = *cap ? *cap * 2 : 1024;
    *ops = (uint8_t*)realloc(*ops, (size_t)*cap);
  }
  (*ops)[(*n)++] = op;
}

void orc_free(orc_result* r) {
  if (!r) return;
  free(r->ops);
  free(r->dp);
  free(r->ptr);
  r->ops = NULL;
  r->dp = NULL;
  r->ptr = NULL;
}

/* ------------------------------------------------------------------ NW --- */

/* One affine row, main.cpp:574-673.  tr receives the reference's pointer
 * letters ('M','E','e','F','f'). */
static void nw_row(scoring* sc, int64_t i, const uint8_t* x, const uint8_t* y, int64_t n,
                   const int* pS, const int* pF, const int* pE, int* cS, int* cF, int* cE,
                   char* tr) {
  /* column 0: main.cpp:601-620 */
  int fo = add_gap(pS[0], sc->go), fe = add_gap(pF[0], sc->ge);
  cF[0] = imax(fo, fe);
  cS[0] = cF[0];
  tr[0] = (cF[0] == fo && cF[0] >= fe) ? 'F' : 'f';
  cE[0] = NEG_INF;
  for (int64_t j = 1; j <= n; ++j) { /* main.cpp:622-672 */
    int f_open = add_gap(pS[j], sc->go), f_ext = add_gap(pF[j], sc->ge);
    cF[j] = imax(f_open, f_ext);
    int e_open = add_gap(cS[j - 1], sc->go), e_ext = add_gap(cE[j - 1], sc->ge);
    cE[j] = imax(e_open, e_ext);
    int s = sigma(sc, x[i - 1], y[j - 1]);
    if (sc->bad) return; /* the reference throws here (main.cpp:121) */
    int dg = imax(pS[j - 1], imax(pE[j - 1], pF[j - 1])); /* main.cpp:641-643 */
    int mv = dg + s;
    if (mv >= cE[j] && mv >= cF[j]) {
      cS[j] = mv;
      tr[j] = 'M';
    } else if (cE[j] >= cF[j]) {
      cS[j] = cE[j];
      tr[j] = (cE[j] == e_open && cE[j] >= e_ext) ? 'E' : 'e';
    } else {
      cS[j] = cF[j];
      tr[j] = (cF[j] == f_open && cF[j] >= f_ext) ? 'F' : 'f';
    }
  }
}

int orc_nw(const uint8_t* x, int64_t m, const uint8_t* y, int64_t n, int mode, double go,
           double ge, int keep_dp, orc_result* out) {
  memset(out, 0, sizeof *out);
  scoring sc = {mode, go, ge, 0, 0, 0};
  size_t w = (size_t)n + 1;
  char* T = (char*)malloc((size_t)(m + 1) * w); /* fullTraceMatrix, main.cpp:706 */
  int* rows = (int*)malloc(6 * w * sizeof(int));
  int* dp = keep_dp ? (int*)malloc((size_t)(m + 1) * w * sizeof(int)) : NULL; /* :732 */
  if (!T || !rows || (keep_dp && !dp)) {
    free(T);
    free(rows);
    free(dp);
    return ORC_ERR_NOMEM;
  }
  int *pS = rows, *pF = rows + w, *pE = rows + 2 * w, *cS = rows + 3 * w, *cF = rows + 4 * w,
      *cE = rows + 5 * w;
  /* row 0: main.cpp:702-729 */
  pS[0] = 0;
  pF[0] = NEG_INF;
  pE[0] = NEG_INF;
  T[0] = 'S';
  for (int64_t j = 1; j <= n; ++j) {
    pE[j] = (j == 1) ? add_gap(pS[j - 1], go) : add_gap(pE[j - 1], ge);
    pS[j] = pE[j];
    pF[j] = NEG_INF;
    T[j] = (j == 1 && pE[j] == add_gap(pS[j - 1], go)) ? 'E' : 'e';
  }
  if (dp) memcpy(dp, pS, w * sizeof(int));
  for (int64_t i = 1; i <= m; ++i) { /* main.cpp:738-762 */
    nw_row(&sc, i, x, y, n, pS, pF, pE, cS, cF, cE, T + (size_t)i * w);
    if (sc.bad) break;
    if (dp) memcpy(dp + (size_t)i * w, cS, w * sizeof(int));
    int* t;
    t = pS, pS = cS, cS = t;
    t = pF, pF = cF, cF = t;
    t = pE, pE = cE, cE = t;
  }
  if (sc.bad) {
    out->bad_x = sc.bad_x;
    out->bad_y = sc.bad_y;
    free(T);
    free(rows);
    free(dp);
    return ORC_ERR_INVALID_CODE;
  }
  out->score = pS[n]; /* main.cpp:766 */
  /* traceback: main.cpp:771-818 */
  int64_t ci = m, cj = n, cap = 0;
  out->end_i = m;
  out->end_j = n;
  while (ci > 0 || cj > 0) {
    char t = T[(size_t)ci * w + cj];
    if (t == 'M') {
      ops_push(&out->ops, &out->n_ops, &cap, 0);
      --ci, --cj;
    } else if (t == 'F' || t == 'f') {
      ops_push(&out->ops, &out->n_ops, &cap, 2);
      --ci;
    } else if (t == 'E' || t == 'e') {
      ops_push(&out->ops, &out->n_ops, &cap, 1);
      --cj;
    } else {
      break;
    }
  }
  out->start_i = ci;
  out->start_j = cj;
  out->dp = dp;
  free(T);
  free(rows);
  return ORC_OK;
}

/* ------------------------------------------------------------------ SW --- */

typedef struct {
  int s, e, f;
  char ptr;
} sw_cell;

/* main.cpp:939-1013 */
static inline sw_cell sw_cell_eval(scoring* sc, int s_d, int e_d, int f_d, int s_l, int e_l,
                                   int s_u, int f_u, unsigned char cx, unsigned char cy) {
  sw_cell c;
  int sub = sigma(sc, cx, cy);
  int mv = imax(s_d, imax(e_d, f_d)) + sub;
  int e_open = add_gap(s_l, sc->go), e_ext = add_gap(e_l, sc->ge);
  int ev = imax(e_open, e_ext);
  int f_open = add_gap(s_u, sc->go), f_ext = add_gap(f_u, sc->ge);
  int fv = imax(f_open, f_ext);
  mv = imax(0, mv); /* main.cpp:968-970 */
  ev = imax(0, ev);
  fv = imax(0, fv);
  c.e = ev;
  c.f = fv;
  if (mv >= ev && mv >= fv) {
    c.s = mv;
    c.ptr = mv > 0 ? 'M' : 'X';
  } else if (ev >= fv) {
    c.s = ev;
    c.ptr = ev > 0 ? ((ev == e_open && ev >= e_ext) ? 'E' : 'e') : 'X';
  } else {
    c.s = fv;
    c.ptr = fv > 0 ? ((fv == f_open && fv >= f_ext) ? 'F' : 'f') : 'X';
  }
  if (c.s == 0) c.ptr = 'X'; /* main.cpp:1010 */
  return c;
}

/* first fill pass, main.cpp:1099-1128 (single rank): best = first strict max
 * in i-major, j-minor order. dp (optional) receives the S matrix. */
static int sw_fill(scoring* sc, const uint8_t* x, int64_t m, const uint8_t* y, int64_t n,
                   int* dp, int* best, int64_t* bi, int64_t* bj) {
  size_t w = (size_t)n + 1;
  int* rows = (int*)calloc(6 * w, sizeof(int));
  if (!rows) return ORC_ERR_NOMEM;
  int *pS = rows, *pE = rows + w, *pF = rows + 2 * w, *cS = rows + 3 * w, *cE = rows + 4 * w,
      *cF = rows + 5 * w;
  *best = 0;
  *bi = 0;
  *bj = 0;
  if (dp) memset(dp, 0, w * sizeof(int));
  for (int64_t i = 1; i <= m && !sc->bad; ++i) {
    cS[0] = cE[0] = cF[0] = 0; /* main.cpp:1104-1106 */
    for (int64_t j = 1; j <= n; ++j) {
      sw_cell c = sw_cell_eval(sc, pS[j - 1], pE[j - 1], pF[j - 1], cS[j - 1], cE[j - 1],
                               pS[j], pF[j], x[i - 1], y[j - 1]);
      if (sc->bad) break;
      cS[j] = c.s;
      cE[j] = c.e;
      cF[j] = c.f;
      if (c.s > *best) { /* main.cpp:1124 */
        *best = c.s;
        *bi = i;
        *bj = j;
      }
    }
    if (dp) memcpy(dp + (size_t)i * w, cS, w * sizeof(int));
    int* t;
    t = pS, pS = cS, cS = t;
    t = pE, pE = cE, cE = t;
    t = pF, pF = cF, cF = t;
  }
  free(rows);
  return sc->bad ? ORC_ERR_INVALID_CODE : ORC_OK;
}

int orc_sw_score(const uint8_t* x, int64_t m, const uint8_t* y, int64_t n, int mode, double go,
                 double ge, int32_t* score, int32_t* end_i, int32_t* end_j) {
  scoring sc = {mode, go, ge, 0, 0, 0};
  int best;
  int64_t bi, bj;
  int rc = sw_fill(&sc, x, m, y, n, NULL, &best, &bi, &bj);
  if (rc != ORC_OK) return rc;
  *score = best;
  *end_i = (int32_t)bi;
  *end_j = (int32_t)bj;
  return ORC_OK;
}

int orc_sw(const uint8_t* x, int64_t m, const uint8_t* y, int64_t n, int mode, double go,
           double ge, int keep_dp, orc_result* out) {
  memset(out, 0, sizeof *out);
  scoring sc = {mode, go, ge, 0, 0, 0};
  size_t w = (size_t)n + 1;
  int* dp = keep_dp ? (int*)malloc((size_t)(m + 1) * w * sizeof(int)) : NULL;
  int best;
  int64_t bi, bj;
  int rc = sw_fill(&sc, x, m, y, n, dp, &best, &bi, &bj);
  if (rc != ORC_OK) {
    out->bad_x = sc.bad_x;
    out->bad_y = sc.bad_y;
    free(dp);
    return rc;
  }
  out->score = best;
  out->end_i = bi;
  out->end_j = bj;
  out->start_i = bi;
  out->start_j = bj;
  out->dp = dp;
  if (bi > 0 && bj > 0) { /* main.cpp:1195-1264: recompute the prefix with pointers */
    size_t tw = (size_t)bj + 1;
    char* P = (char*)malloc((size_t)(bi + 1) * tw);
    int* rows = (int*)calloc(6 * tw, sizeof(int));
    if (!P || !rows) {
      free(P);
      free(rows);
      return ORC_ERR_NOMEM;
    }
    memset(P, 'X', (size_t)(bi + 1) * tw); /* main.cpp:1203-1204 */
    int *pS = rows, *pE = rows + tw, *pF = rows + 2 * tw, *cS = rows + 3 * tw,
        *cE = rows + 4 * tw, *cF = rows + 5 * tw;
    for (int64_t i = 1; i <= bi; ++i) {
      cS[0] = cE[0] = cF[0] = 0;
      for (int64_t j = 1; j <= bj; ++j) {
        sw_cell c = sw_cell_eval(&sc, pS[j - 1], pE[j - 1], pF[j - 1], cS[j - 1], cE[j - 1],
                                 pS[j], pF[j], x[i - 1], y[j - 1]);
        cS[j] = c.s;
        cE[j] = c.e;
        cF[j] = c.f;
        P[(size_t)i * tw + j] = c.ptr;
      }
      int* t;
      t = pS, pS = cS, cS = t;
      t = pE, pE = cE, cE = t;
      t = pF, pF = cF, cF = t;
    }
    int64_t ci = bi, cj = bj, cap = 0;
    while (ci > 0 || cj > 0) { /* main.cpp:1232-1259; 'X' <=> S==0 */
      char p = P[(size_t)ci * tw + cj];
      if (p == 'M') {
        ops_push(&out->ops, &out->n_ops, &cap, 0);
        --ci, --cj;
      } else if (p == 'F' || p == 'f') {
        ops_push(&out->ops, &out->n_ops, &cap, 2);
        --ci;
      } else if (p == 'E' || p == 'e') {
        ops_push(&out->ops, &out->n_ops, &cap, 1);
        --cj;
      } else {
        break;
      }
    }
    out->start_i = ci;
    out->start_j = cj;
    free(P);
    free(rows);
  }
  return ORC_OK;
}

/* ----------------------------------------------------------------- LCS --- */

int orc_lcs(const uint8_t* x, int64_t m, const uint8_t* y, int64_t n, int keep_dp,
            orc_result* out) {
  memset(out, 0, sizeof *out);
  size_t w = (size_t)n + 1;
  char* B = (char*)malloc((size_t)(m + 1) * w); /* main.cpp:1568 */
  int* L = (int*)calloc(keep_dp ? (size_t)(m + 1) * w : 2 * w, sizeof(int));
  if (!B || !L) {
    free(B);
    free(L);
    return ORC_ERR_NOMEM;
  }
  memset(B, ' ', (size_t)(m + 1) * w);
  for (int64_t i = 1; i <= m; ++i) { /* main.cpp:1575-1588 */
    int* up = keep_dp ? L + (size_t)(i - 1) * w : L + (size_t)((i - 1) & 1) * w;
    int* cur = keep_dp ? L + (size_t)i * w : L + (size_t)(i & 1) * w;
    char* b = B + (size_t)i * w;
    cur[0] = 0;
    for (int64_t j = 1; j <= n; ++j) {
      if (x[i - 1] == y[j - 1]) {
        cur[j] = up[j - 1] + 1;
        b[j] = 'D';
      } else if (up[j] >= cur[j - 1]) {
        cur[j] = up[j];
        b[j] = 'U';
      } else {
        cur[j] = cur[j - 1];
        b[j] = 'L';
      }
    }
  }
  out->score = keep_dp ? L[(size_t)m * w + n] : L[(size_t)(m & 1) * w + n];
  if (m == 0) out->score = 0;
  int64_t ci = m, cj = n, cap = 0;
  out->end_i = m;
  out->end_j = n;
  while (ci > 0 && cj > 0) { /* main.cpp:1602-1620 */
    char b = B[(size_t)ci * w + cj];
    if (b == 'D') {
      ops_push(&out->ops, &out->n_ops, &cap, 0);
      --ci, --cj;
    } else if (b == 'U') {
      ops_push(&out->ops, &out->n_ops, &cap, 2);
      --ci;
    } else {
      ops_push(&out->ops, &out->n_ops, &cap, 1);
      --cj;
    }
  }
  while (ci > 0) { /* main.cpp:1623-1628 */
    ops_push(&out->ops, &out->n_ops, &cap, 2);
    --ci;
  }
  while (cj > 0) { /* main.cpp:1629-1634 */
    ops_push(&out->ops, &out->n_ops, &cap, 1);
    --cj;
  }
  out->start_i = 0;
  out->start_j = 0;
  if (keep_dp) {
    out->dp = L;
    out->ptr = B;
  } else {
    free(L);
    free(B);
  }
  return ORC_OK;
}

/* ------------------------------------------------ packed (2-bit) variants --- */

static inline void put2(uint64_t* T, size_t idx, unsigned v) {
  T[idx >> 5] |= (uint64_t)v << ((idx & 31) * 2);
}
static inline unsigned get2(const uint64_t* T, size_t idx) {
  return (unsigned)(T[idx >> 5] >> ((idx & 31) * 2)) & 3u;
}

int orc_nw_packed(const uint8_t* x, int64_t m, const uint8_t* y, int64_t n, int mode, double go,
                  double ge, orc_result* out) {
  memset(out, 0, sizeof *out);
  scoring sc = {mode, go, ge, 0, 0, 0};
  size_t w = (size_t)n + 1;
  size_t cells = (size_t)(m + 1) * w;
  uint64_t* T = (uint64_t*)calloc((cells + 31) / 32, sizeof(uint64_t));
  int* rows = (int*)malloc(6 * w * sizeof(int));
  char* tr = (char*)malloc(w);
  if (!T || !rows || !tr) {
    free(T);
    free(rows);
    free(tr);
    return ORC_ERR_NOMEM;
  }
  int *pS = rows, *pF = rows + w, *pE = rows + 2 * w, *cS = rows + 3 * w, *cF = rows + 4 * w,
      *cE = rows + 5 * w;
  pS[0] = 0;
  pF[0] = pE[0] = NEG_INF;
  for (int64_t j = 1; j <= n; ++j) {
    pE[j] = (j == 1) ? add_gap(pS[j - 1], go) : add_gap(pE[j - 1], ge);
    pS[j] = pE[j];
    pF[j] = NEG_INF;
    put2(T, (size_t)j, 1);
  }
  for (int64_t i = 1; i <= m; ++i) {
    nw_row(&sc, i, x, y, n, pS, pF, pE, cS, cF, cE, tr);
    if (sc.bad) break;
    size_t base = (size_t)i * w;
    for (int64_t j = 0; j <= n; ++j)
      put2(T, base + j, tr[j] == 'M' ? 0u : ((tr[j] == 'E' || tr[j] == 'e') ? 1u : 2u));
    int* t;
    t = pS, pS = cS, cS = t;
    t = pF, pF = cF, cF = t;
    t = pE, pE = cE, cE = t;
  }
  int rc = ORC_OK;
  if (sc.bad) {
    out->bad_x = sc.bad_x;
    out->bad_y = sc.bad_y;
    rc = ORC_ERR_INVALID_CODE;
  } else {
    out->score = pS[n];
    int64_t ci = m, cj = n, cap = 0;
    out->end_i = m;
    out->end_j = n;
    while (ci > 0 || cj > 0) {
      unsigned c = get2(T, (size_t)ci * w + cj);
      ops_push(&out->ops, &out->n_ops, &cap, (uint8_t)c);
      if (c == 0)
        --ci, --cj;
      else if (c == 1)
        --cj;
      else
        --ci;
    }
  }
  free(T);
  free(rows);
  free(tr);
  return rc;
}

int orc_lcs_packed(const uint8_t* x, int64_t m, const uint8_t* y, int64_t n, orc_result* out) {
  memset(out, 0, sizeof *out);
  size_t w = (size_t)n + 1;
  size_t cells = (size_t)(m + 1) * w;
  uint64_t* T = (uint64_t*)calloc((cells + 31) / 32, sizeof(uint64_t));
  int* L = (int*)calloc(2 * w, sizeof(int));
  if (!T || !L) {
    free(T);
    free(L);
    return ORC_ERR_NOMEM;
  }
  for (int64_t i = 1; i <= m; ++i) {
    int* up = L + (size_t)((i - 1) & 1) * w;
    int* cur = L + (size_t)(i & 1) * w;
    size_t base = (size_t)i * w;
    cur[0] = 0;
    for (int64_t j = 1; j <= n; ++j) {
      if (x[i - 1] == y[j - 1]) {
        cur[j] = up[j - 1] + 1; /* 'D' = 0 */
      } else if (up[j] >= cur[j - 1]) {
        cur[j] = up[j];
        put2(T, base + j, 2); /* 'U' */
      } else {
        cur[j] = cur[j - 1];
        put2(T, base + j, 1); /* 'L' */
      }
    }
  }
  out->score = m ? L[(size_t)(m & 1) * w + n] : 0;
  int64_t ci = m, cj = n, cap = 0;
  out->end_i = m;
  out->end_j = n;
  while (ci > 0 && cj > 0) {
    unsigned c = get2(T, (size_t)ci * w + cj);
    ops_push(&out->ops, &out->n_ops, &cap, (uint8_t)c);
    if (c == 0)
      --ci, --cj;
    else if (c == 1)
      --cj;
    else
      --ci;
  }
  while (ci > 0) ops_push(&out->ops, &out->n_ops, &cap, 2), --ci;
  while (cj > 0) ops_push(&out->ops, &out->n_ops, &cap, 1), --cj;
  free(T);
  free(L);
  return ORC_OK;
}

/* --------------------------------------------------------------- batch --- */

typedef struct {
  const uint8_t* seqs;
  const int64_t *off_x, *off_y;
  const int32_t *len_x, *len_y;
  int64_t lo, hi;
  int mode;
  double go, ge;
  int32_t *score, *end_i, *end_j;
  int rc;
} batch_job;

static void* batch_worker(void* p) {
  batch_job* b = (batch_job*)p;
  for (int64_t k = b->lo; k < b->hi; ++k) {
    int rc = orc_sw_score(b->seqs + b->off_x[k], b->len_x[k], b->seqs + b->off_y[k], b->len_y[k],
                          b->mode, b->go, b->ge, &b->score[k], &b->end_i[k], &b->end_j[k]);
    if (rc != ORC_OK) b->rc = rc;
  }
  return NULL;
}

int orc_sw_score_batch(const uint8_t* seqs, const int64_t* off_x, const int32_t* len_x,
                       const int64_t* off_y, const int32_t* len_y, int64_t n_pairs, int mode,
                       double go, double ge, int n_threads, int32_t* score, int32_t* end_i,
                       int32_t* end_j) {
  if (n_threads < 1) n_threads = 1;
  if (n_threads > 256) n_threads = 256;
  pthread_t th[256];
  batch_job jobs[256];
  for (int t = 0; t < n_threads; ++t) {
    batch_job j = {seqs,  off_x, off_y, len_x, len_y, n_pairs * t / n_threads,
                   n_pairs * (t + 1) / n_threads, mode, go, ge, score, end_i, end_j, ORC_OK};
    jobs[t] = j;
    if (n_threads == 1)
      batch_worker(&jobs[t]);
    else
      pthread_create(&th[t], NULL, batch_worker, &jobs[t]);
  }
  int rc = ORC_OK;
  for (int t = 0; t < n_threads; ++t) {
    if (n_threads > 1) pthread_join(th[t], NULL);
    if (jobs[t].rc != ORC_OK) rc = jobs[t].rc;
  }
  return rc;
}

/* -------------------------------------------------------------- banded --- */

/* Banded NW for property checks at sizes the full oracle cannot reach.
 * Row i covers columns [lo_i, hi_i], lo_i = max(0, c_i - band),
 * hi_i = min(n, c_i + band), c_i = i*n/m.  Cells outside the band read as
 * -inf.  Same recurrence / tie-breaks as nw_row. */
int orc_nw_banded(const uint8_t* x, int64_t m, const uint8_t* y, int64_t n, int mode, double go,
                  double ge, int64_t band, orc_result* out, int* touched_edge) {
  memset(out, 0, sizeof *out);
  *touched_edge = 0;
  scoring sc = {mode, go, ge, 0, 0, 0};
  size_t bw = (size_t)(2 * band + 1);
  size_t w = (size_t)n + 2;
  uint64_t* T = (uint64_t*)calloc(((size_t)(m + 1) * bw + 31) / 32, sizeof(uint64_t));
  int* rows = (int*)malloc(6 * w * sizeof(int));
  if (!T || !rows) {
    free(T);
    free(rows);
    return ORC_ERR_NOMEM;
  }
  int *pS = rows, *pF = rows + w, *pE = rows + 2 * w, *cS = rows + 3 * w, *cF = rows + 4 * w,
      *cE = rows + 5 * w;
  for (size_t k = 0; k < 6 * w; ++k) rows[k] = NEG_INF;
#define CTR(i) ((m) ? (int64_t)((__int128)(i) * n / m) : 0)
#define LO(i) (CTR(i) - band < 0 ? 0 : CTR(i) - band)
#define HI(i) (CTR(i) + band > n ? n : CTR(i) + band)
  pS[0] = 0;
  for (int64_t j = 1; j <= HI(0); ++j) {
    pE[j] = (j == 1) ? add_gap(pS[j - 1], go) : add_gap(pE[j - 1], ge);
    pS[j] = pE[j];
    pF[j] = NEG_INF;
  }
  for (int64_t i = 1; i <= m && !sc.bad; ++i) {
    int64_t lo = LO(i), hi = HI(i), plo = LO(i - 1), phi = HI(i - 1);
    size_t base = (size_t)i * bw;
    /* cells of the previous row outside its band are -inf */
    for (int64_t j = phi + 1; j <= hi; ++j) pS[j] = pF[j] = pE[j] = NEG_INF;
    if (lo > 0) {
      cS[lo - 1] = cE[lo - 1] = cF[lo - 1] = NEG_INF;
      if (lo - 1 < plo) pS[lo - 1] = pE[lo - 1] = pF[lo - 1] = NEG_INF;
    }
    for (int64_t j = lo; j <= hi; ++j) {
      unsigned code;
      if (j == 0) {
        int fo = add_gap(pS[0], go), fe = add_gap(pF[0], ge);
        cF[0] = imax(fo, fe);
        cS[0] = cF[0];
        cE[0] = NEG_INF;
        code = 2;
      } else {
        int f_open = pS[j] <= NEG_INF ? NEG_INF : add_gap(pS[j], go);
        int f_ext = pF[j] <= NEG_INF ? NEG_INF : add_gap(pF[j], ge);
        if (pF[j] == NEG_INF && pS[j] != NEG_INF) f_ext = add_gap(NEG_INF, ge);
        cF[j] = imax(f_open, f_ext);
        int e_open = cS[j - 1] <= NEG_INF ? NEG_INF : add_gap(cS[j - 1], go);
        int e_ext = cE[j - 1] <= NEG_INF ? NEG_INF : add_gap(cE[j - 1], ge);
        if (cE[j - 1] == NEG_INF && cS[j - 1] != NEG_INF) e_ext = add_gap(NEG_INF, ge);
        cE[j] = imax(e_open, e_ext);
        int s = sigma(&sc, x[i - 1], y[j - 1]);
        if (sc.bad) break;
        int dg = imax(pS[j - 1], imax(pE[j - 1], pF[j - 1]));
        int mv = dg <= NEG_INF ? NEG_INF : dg + s;
        if (mv >= cE[j] && mv >= cF[j]) {
          cS[j] = mv;
          code = 0;
        } else if (cE[j] >= cF[j]) {
          cS[j] = cE[j];
          code = 1;
        } else {
          cS[j] = cF[j];
          code = 2;
        }
      }
      put2(T, base + (size_t)(j - lo), code);
    }
    int* t;
    t = pS, pS = cS, cS = t;
    t = pF, pF = cF, cF = t;
    t = pE, pE = cE, cE = t;
  }
  int rc = ORC_OK;
  if (sc.bad) {
    out->bad_x = sc.bad_x;
    out->bad_y = sc.bad_y;
    rc = ORC_ERR_INVALID_CODE;
  } else {
    out->score = pS[n];
    int64_t ci = m, cj = n, cap = 0;
    out->end_i = m;
    out->end_j = n;
    while (ci > 0 || cj > 0) {
      unsigned c;
      if (ci == 0)
        c = 1;
      else {
        int64_t lo = LO(ci), hi = HI(ci);
        if ((lo > 0 && cj <= lo + 1) || (hi < n && cj >= hi - 1)) *touched_edge = 1;
        if (cj < lo || cj > hi) {
          *touched_edge = 2;
          break;
        }
        c = get2(T, (size_t)ci * bw + (size_t)(cj - lo));
      }
      ops_push(&out->ops, &out->n_ops, &cap, (uint8_t)c);
      if (c == 0)
        --ci, --cj;
      else if (c == 1)
        --cj;
      else
        --ci;
    }
  }
#undef CTR
#undef LO
#undef HI
  free(T);
  free(rows);
  return rc;
}

/* Banded SW (main.cpp:1099-1128 fill + arg-max, :1182-1265 traceback) for sizes the full oracle cannot reach: row i
 * covers columns [lo_i, hi_i] around c_i = i*n/m as in orc_nw_banded.  Cells outside the band read as S = E = F = 0,
 * the floor every SW cell has anyway (main.cpp:968-970), i.e. a LOWER bound of their true values; in-band values are
 * therefore lower bounds of the true ones and equal them wherever the best alignment ending in the cell stays inside
 * the band.  The result (score, first row-major maximum, op list) is exact when the best local alignment and the
 * cells that could tie with it lie inside the band -- true for the ~90 %-identity pairs it is used on; the walk reports
 * through *touched_edge when it comes within a column of the band's edge.  One pointer per in-band cell, 2 bits
 * (0 M, 1 E, 2 F, 3 X).  Validated against orc_sw in tests/test_oracle_golden.py at sizes both can run. */
int orc_sw_banded(const uint8_t* x, int64_t m, const uint8_t* y, int64_t n, int mode, double go, double ge,
                  int64_t band, orc_result* out, int* touched_edge) {
  memset(out, 0, sizeof *out);
  *touched_edge = 0;
  scoring sc = {mode, go, ge, 0, 0, 0};
  size_t bw = (size_t)(2 * band + 1);
  size_t w = (size_t)n + 2;
  uint64_t* T = (uint64_t*)calloc(((size_t)(m + 1) * bw + 31) / 32, sizeof(uint64_t));
  int* rows = (int*)calloc(6 * w, sizeof(int));
  if (!T || !rows) {
    free(T);
    free(rows);
    return ORC_ERR_NOMEM;
  }
  int *pS = rows, *pE = rows + w, *pF = rows + 2 * w, *cS = rows + 3 * w, *cE = rows + 4 * w, *cF = rows + 5 * w;
#define CTR(i) ((m) ? (int64_t)((__int128)(i) * n / m) : 0)
#define LO(i) (CTR(i) - band < 0 ? 0 : CTR(i) - band)
#define HI(i) (CTR(i) + band > n ? n : CTR(i) + band)
  int best = 0;
  int64_t bi = 0, bj = 0;
  for (int64_t i = 1; i <= m && !sc.bad; ++i) {
    int64_t lo = LO(i), hi = HI(i), plo = LO(i - 1), phi = HI(i - 1);
    size_t base = (size_t)i * bw;
    /* cells of the previous row outside ITS band, and the cell left of this row's band, read as zero */
    for (int64_t j = phi + 1; j <= hi; ++j) pS[j] = pE[j] = pF[j] = 0;
    for (int64_t j = (lo > 0 ? lo - 1 : 0); j < plo; ++j) pS[j] = pE[j] = pF[j] = 0;
    if (lo > 0) cS[lo - 1] = cE[lo - 1] = cF[lo - 1] = 0;
    for (int64_t j = lo; j <= hi; ++j) {
      if (j == 0) { /* main.cpp:1104-1106 */
        cS[0] = cE[0] = cF[0] = 0;
        put2(T, base, 3);
        continue;
      }
      sw_cell c = sw_cell_eval(&sc, pS[j - 1], pE[j - 1], pF[j - 1], cS[j - 1], cE[j - 1], pS[j], pF[j], x[i - 1], y[j - 1]);
      if (sc.bad) break;
      cS[j] = c.s;
      cE[j] = c.e;
      cF[j] = c.f;
      put2(T, base + (size_t)(j - lo), c.ptr == 'M' ? 0u : ((c.ptr == 'E' || c.ptr == 'e') ? 1u : ((c.ptr == 'F' || c.ptr == 'f') ? 2u : 3u)));
      if (c.s > best) { /* main.cpp:1124 */
        best = c.s;
        bi = i;
        bj = j;
      }
    }
    int* t;
    t = pS, pS = cS, cS = t;
    t = pE, pE = cE, cE = t;
    t = pF, pF = cF, cF = t;
  }
  int rc = ORC_OK;
  if (sc.bad) {
    out->bad_x = sc.bad_x;
    out->bad_y = sc.bad_y;
    rc = ORC_ERR_INVALID_CODE;
  } else {
    out->score = best;
    out->end_i = bi;
    out->end_j = bj;
    int64_t ci = bi, cj = bj, cap = 0;
    while (ci > 0 && cj > 0) { /* main.cpp:1232-1259 */
      int64_t lo = LO(ci), hi = HI(ci);
      if ((lo > 0 && cj <= lo + 1) || (hi < n && cj >= hi - 1)) *touched_edge = 1;
      if (cj < lo || cj > hi) {
        *touched_edge = 2;
        break;
      }
      unsigned c = get2(T, (size_t)ci * bw + (size_t)(cj - lo));
      if (c == 3) break;
      ops_push(&out->ops, &out->n_ops, &cap, (uint8_t)c);
      if (c == 0)
        --ci, --cj;
      else if (c == 1)
        --cj;
      else
        --ci;
    }
    out->start_i = ci;
    out->start_j = cj;
  }
#undef CTR
#undef LO
#undef HI
  free(T);
  free(rows);
  return rc;
}

/* Banded LCS (main.cpp:1575-1592 fill, :1602-1635 traceback) for sizes the full oracle cannot reach: row i
 * covers columns [lo_i, hi_i] around c_i = i*n/m as in orc_nw_banded.  L is monotone in i and in j, so a cell
 * outside the band is replaced by a LOWER bound of its true value: to the right of the previous row's band the
 * last value inside it, to the left of the current row's band the (older) value the row buffer still holds.
 * In-band values are therefore lower bounds of the true ones, and the result is exact whenever the
 * reference's traceback path stays inside the band (reported through *touched_edge as in orc_nw_banded;
 * tests/test_oracle_golden.py checks it against orc_lcs at sizes both can run). */
int orc_lcs_banded(const uint8_t* x, int64_t m, const uint8_t* y, int64_t n, int64_t band, orc_result* out,
                   int* touched_edge) {
  memset(out, 0, sizeof *out);
  *touched_edge = 0;
  size_t bw = (size_t)(2 * band + 1);
  size_t w = (size_t)n + 2;
  uint64_t* T = (uint64_t*)calloc(((size_t)(m + 1) * bw + 31) / 32, sizeof(uint64_t));
  int* L = (int*)calloc(2 * w, sizeof(int));
  if (!T || !L) {
    free(T);
    free(L);
    return ORC_ERR_NOMEM;
  }
#define CTR(i) ((m) ? (int64_t)((__int128)(i) * n / m) : 0)
#define LO(i) (CTR(i) - band < 0 ? 0 : CTR(i) - band)
#define HI(i) (CTR(i) + band > n ? n : CTR(i) + band)
  for (int64_t i = 1; i <= m; ++i) {
    int* up = L + (size_t)((i - 1) & 1) * w;
    int* cur = L + (size_t)(i & 1) * w;
    int64_t lo = LO(i), hi = HI(i), phi = HI(i - 1);
    size_t base = (size_t)i * bw;
    for (int64_t j = phi + 1; j <= hi; ++j) up[j] = up[phi]; /* right of the previous row's band */
    if (lo == 0)
      cur[0] = 0;
    else if (cur[lo - 1] < up[lo - 1])
      cur[lo - 1] = up[lo - 1]; /* L[i][lo-1] >= L[i-1][lo-1] */
    for (int64_t j = (lo == 0 ? 1 : lo); j <= hi; ++j) {
      unsigned code;
      if (x[i - 1] == y[j - 1]) {
        cur[j] = up[j - 1] + 1;
        code = 0; /* 'D' */
      } else if (up[j] >= cur[j - 1]) {
        cur[j] = up[j];
        code = 2; /* 'U' */
      } else {
        cur[j] = cur[j - 1];
        code = 1; /* 'L' */
      }
      put2(T, base + (size_t)(j - lo), code);
    }
  }
  out->score = m ? L[(size_t)(m & 1) * w + n] : 0;
  int64_t ci = m, cj = n, cap = 0;
  out->end_i = m;
  out->end_j = n;
  while (ci > 0 && cj > 0) {
    int64_t lo = LO(ci), hi = HI(ci);
    if ((lo > 0 && cj <= lo + 1) || (hi < n && cj >= hi - 1)) *touched_edge = 1;
    if (cj < lo || cj > hi) {
      *touched_edge = 2;
      break;
    }
    unsigned c = get2(T, (size_t)ci * bw + (size_t)(cj - lo));
    ops_push(&out->ops, &out->n_ops, &cap, (uint8_t)c);
    if (c == 0)
      --ci, --cj;
    else if (c == 1)
      --cj;
    else
      --ci;
  }
  if (*touched_edge != 2) {
    while (ci > 0) ops_push(&out->ops, &out->n_ops, &cap, 2), --ci;
    while (cj > 0) ops_push(&out->ops, &out->n_ops, &cap, 1), --cj;
  }
#undef CTR
#undef LO
#undef HI
  free(T);
  free(L);
  return ORC_OK;
}

/* -------------------------------------------------------------- stripes --- */

struct orc_stripe {
  int kind, is_first;
  int64_t m, n;
  uint8_t* T; /* (m+1) x (n+1) direction codes 0/1/2, SW stop cell 3 */
  int best;   /* SW: first strict row-major maximum inside this stripe (main.cpp:1124-1127) */
  int64_t best_i, best_j; /* 1-based, j local to the stripe */
};

/* One column stripe of the NW / SW / LCS matrix (kind 0 / 1 / 2), row by row, with the left border taken from `left`
 * (values of the column just left of the stripe) and the stripe's last column written to
 * `right`.  Same recurrences as nw_row / orc_lcs; only the source of column 0 differs. */
int orc_stripe_fill(int kind, const uint8_t* x, int64_t m, const uint8_t* y, int64_t n, int mode, double go, double ge,
                    int64_t n_total, int64_t col_begin, int is_first, int is_last, int32_t* left, int32_t* right,
                    int64_t chunk_rows, orc_stripe_event_fn on_event, void* user, orc_stripe** handle, int64_t* score) {
  (void)n_total;
  scoring sc = {mode, go, ge, 0, 0, 0};
  size_t w = (size_t)n + 1;
  orc_stripe* h = (orc_stripe*)calloc(1, sizeof *h);
  h->kind = kind;
  h->is_first = is_first;
  h->m = m;
  h->n = n;
  h->T = (uint8_t*)calloc((size_t)(m + 1) * w, 1);
  int* rows = (int*)malloc(6 * w * sizeof(int));
  int *pS = rows, *pF = rows + w, *pE = rows + 2 * w, *cS = rows + 3 * w, *cF = rows + 4 * w, *cE = rows + 5 * w;
  /* row 0: the global chain evaluated up to this stripe */
  int s0 = 0, e0 = NEG_INF;
  for (int64_t j = 1; j <= col_begin + n; ++j) {
    if (kind == 0) {
      e0 = (j == 1) ? add_gap(s0, go) : add_gap(e0, ge);
      s0 = e0;
    }
    if (j >= col_begin) {
      pS[j - col_begin] = s0;
      pE[j - col_begin] = e0;
      pF[j - col_begin] = NEG_INF;
      if (j > col_begin) h->T[j - col_begin] = 1;
    }
  }
  if (col_begin == 0) {
    pS[0] = 0;
    pE[0] = pF[0] = NEG_INF;
  }
  if (kind != 0)
    for (size_t j = 0; j < w; ++j) pS[j] = pE[j] = pF[j] = 0;
  int64_t chunks_in = 0, chunks_out = 0, n_chunks = (m + chunk_rows - 1) / chunk_rows;
  for (int64_t i = 1; i <= m; ++i) {
    if (!is_first && on_event && chunks_in < n_chunks && (i - 1) / chunk_rows >= chunks_in) {
      if (on_event(user, 0, chunks_in)) return -9;
      ++chunks_in;
    }
    uint8_t* t = h->T + (size_t)i * w;
    if (is_first) {
      if (kind == 0) {
        int fo = add_gap(pS[0], go), fe = add_gap(pF[0], ge);
        cF[0] = imax(fo, fe);
        cS[0] = cF[0];
        cE[0] = NEG_INF;
      } else {
        cS[0] = cE[0] = cF[0] = 0;
      }
      t[0] = 2;
    } else {
      cS[0] = left[2 * i];
      cE[0] = left[2 * i + 1];
      cF[0] = (kind == 0) ? NEG_INF : 0; /* never read: F of column 0 feeds nothing to the right */
    }
    for (int64_t j = 1; j <= n; ++j) {
      if (kind == 0) {
        int f_open = add_gap(pS[j], go), f_ext = add_gap(pF[j], ge);
        cF[j] = imax(f_open, f_ext);
        int e_open = add_gap(cS[j - 1], go), e_ext = add_gap(cE[j - 1], ge);
        cE[j] = imax(e_open, e_ext);
        int sg = sigma(&sc, x[i - 1], y[j - 1]);
        int mv = pS[j - 1] + sg;
        if (mv >= cE[j] && mv >= cF[j]) {
          cS[j] = mv;
          t[j] = 0;
        } else if (cE[j] >= cF[j]) {
          cS[j] = cE[j];
          t[j] = 1;
        } else {
          cS[j] = cF[j];
          t[j] = 2;
        }
      } else if (kind == 1) {
        /* SW cell, main.cpp:939-1013; diagonal max(S,E,F) == S (Q8), also on the stripe border */
        sw_cell c = sw_cell_eval(&sc, pS[j - 1], pS[j - 1], pS[j - 1], cS[j - 1], cE[j - 1], pS[j], pF[j], x[i - 1], y[j - 1]);
        cS[j] = c.s;
        cE[j] = c.e;
        cF[j] = c.f;
        t[j] = c.ptr == 'X' ? 3 : (c.ptr == 'M' ? 0 : ((c.ptr == 'E' || c.ptr == 'e') ? 1 : 2));
        if (c.s > h->best) {
          h->best = c.s;
          h->best_i = i;
          h->best_j = j;
        }
      } else {
        if (x[i - 1] == y[j - 1]) {
          cS[j] = pS[j - 1] + 1;
          t[j] = 0;
        } else if (pS[j] >= cS[j - 1]) {
          cS[j] = pS[j];
          t[j] = 2;
        } else {
          cS[j] = cS[j - 1];
          t[j] = 1;
        }
        cE[j] = cF[j] = 0;
      }
    }
    if (!is_last) {
      right[2 * i] = cS[n];
      right[2 * i + 1] = cE[n];
      if (on_event && (i % chunk_rows == 0 || i == m)) {
        if (on_event(user, 1, chunks_out)) return -9;
        ++chunks_out;
      }
    }
    int* tt;
    tt = pS, pS = cS, cS = tt;
    tt = pF, pF = cF, cF = tt;
    tt = pE, pE = cE, cE = tt;
  }
  *score = (kind == 1) ? h->best : pS[n];
  free(rows);
  *handle = h;
  return sc.bad ? ORC_ERR_INVALID_CODE : ORC_OK;
}

int orc_stripe_traceback(orc_stripe* h, int64_t ci, int64_t cj, orc_result* out) {
  memset(out, 0, sizeof *out);
  size_t w = (size_t)h->n + 1;
  int64_t cap = 0;
  out->end_i = ci;
  out->end_j = cj;
  while (h->kind == 1 ? (ci > 0 && cj > 0) : (h->is_first ? (ci > 0 || cj > 0) : (cj > 0))) {
    unsigned c = (ci == 0) ? 1u : (cj == 0 ? 2u : h->T[(size_t)ci * w + cj]);
    if (c == 3u) break; /* SW stop cell, main.cpp:1233 */
    ops_push(&out->ops, &out->n_ops, &cap, (uint8_t)c);
    if (c == 0)
      --ci, --cj;
    else if (c == 1)
      --cj;
    else
      --ci;
  }
  out->start_i = ci;
  out->start_j = cj;
  return ORC_OK;
}

void orc_stripe_best(const orc_stripe* h, int64_t* best, int64_t* bi, int64_t* bj) {
  *best = h->best;
  *bi = h->best_i;
  *bj = h->best_j;
}

void orc_stripe_free(orc_stripe* h) {
  if (!h) return;
  free(h->T);
  free(h);
}
