/*
 * seqalign_host.h -- host-side (no GPU) half of the drop-in: the reference's input parsing and
 * byte-exact result writers, fed by the op list that libseqalign_hip.so returns.
 *
 * These replace the non-DP parts of the reference's three entry points:
 *   sah_read_fasta        processFasta()                      main.cpp:290-308
 *   sah_accession         getAccession()                      main.cpp:212-230
 *   sah_gene_symbol       getGeneSymbol()                     main.cpp:243-277
 *   sah_write_global      result section of globalalign()     main.cpp:825-907
 *   sah_write_local       result section of localalign()      main.cpp:1364-1368, 1402-1413, 1471-1543
 *   sah_write_lcs         result section of lcs()             main.cpp:1637-1707
 * Plain C ABI so that tests (ctypes) and other hosts can use the writers without the CLI.
 */
#ifndef SEQALIGN_HOST_H
#define SEQALIGN_HOST_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ops[] convention: see seqalign_hip.h (0 diagonal, 1 left, 2 up; end -> start). */

typedef struct {
  const char* outdir;
  const char* header1;   /* FASTA header of the query, without '>' */
  const char* header2;
  int mode;              /* SA_MODE_DNA / SA_MODE_PROTEIN: only selects the ID parsing */
  double gap_open, gap_extend;
  const uint8_t* x;
  int64_t m;
  const uint8_t* y;
  int64_t n;
  int64_t score;
  int64_t end_i, end_j;  /* traceback start cell */
  const uint8_t* ops;
  int64_t n_ops;
  int64_t time_ms;
  int verbose;           /* --verbose: print the reference's summary + coloured alignment to stdout */
  int binary, txt;       /* --binary / --txt: dump dp (and lcs pointers) when dp != NULL */
  const int32_t* dp;     /* optional (m+1)x(n+1) S / L matrix */
} sah_job;

/* malloc'd outputs, free with sah_free.  Returns 0, or -1 if the file cannot be opened
 * (the reference throws "Error: Unable to open <file>"). */
int sah_read_fasta(const char* path, char** header, char** seq, int64_t* seq_len);
void sah_free(void* p);

/* Every record of a multi-FASTA file (the --batch extension of the CLI; the reference's reader
 * above keeps the first header and concatenates all records, Q6).  Same line rules as
 * sah_read_fasta inside a record: blank lines skipped, no '\r' stripping.  *headers and *seqs are
 * malloc'd arrays of *n_records malloc'd strings, *lens a malloc'd array; release everything with
 * sah_free_records. */
int sah_read_fasta_records(const char* path, int64_t* n_records, char*** headers, char*** seqs, int64_t** lens);
void sah_free_records(int64_t n_records, char** headers, char** seqs, int64_t* lens);

/* copy the ID into buf (NUL-terminated, truncated to cap) */
void sah_accession(const char* header, int mode, char* buf, int64_t cap);
void sah_gene_symbol(const char* header, int mode, char* buf, int64_t cap);

int sah_write_global(const sah_job* job);
int sah_write_local(const sah_job* job);
int sah_write_lcs(const sah_job* job);

#ifdef __cplusplus
}
#endif
#endif
