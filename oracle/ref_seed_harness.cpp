// ref_seed_harness.cpp -- TEST INFRASTRUCTURE, build container only.
//
// Makes the three DP helpers of the reference's seed_aligner callable: the unmodified
// /root/reference/src/main_fmindex.cpp is compiled *where it lies* (included below, its main()
// renamed) and thin extern "C" wrappers expose
//   align_segment_globally   main_fmindex.cpp:1049-1143
//   compute_lcs_for_segment  main_fmindex.cpp:1145-1207
//   perform_sw_in_window     main_fmindex.cpp:938-1047
// plus its two score functions (main_fmindex.cpp:374-390).  Output: oracle/_ref/libref_seed.so
// (git-ignored).  tests/golden/make_golden_seed.py drives it to produce committed fixtures; nothing
// of the reference's text is copied.
#define main seed_aligner_reference_main
#include "/root/reference/src/main_fmindex.cpp"
#undef main

#include <cstring>

namespace {
void put(const std::string& s, char* dst, long cap) {
  const long n = std::min<long>((long)s.size(), cap - 1);
  memcpy(dst, s.data(), (size_t)n);
  dst[n] = 0;
}
ScoreFn fn_of(int mode) { return mode == 0 ? &edna_score : &blosum62_score; }
}  // namespace

extern "C" {

// returns 0, or 1 if the reference threw (invalid residue)
int ref_seed_score(int mode, int x, int y, int* out) {
  try {
    *out = fn_of(mode)((char)x, (char)y);
    return 0;
  } catch (const std::exception&) {
    return 1;
  }
}

int ref_seg_nw(const char* s1, long m, const char* s2, long n, int mode, double go, double ge, int* score, char* a1, char* a2,
               long cap) {
  try {
    const AlignmentResult r = align_segment_globally(std::string(s1, (size_t)m), std::string(s2, (size_t)n), fn_of(mode), go, ge);
    *score = r.score;
    put(r.aligned_seq1, a1, cap);
    put(r.aligned_seq2, a2, cap);
    return 0;
  } catch (const std::exception&) {
    return 1;
  }
}

int ref_seg_lcs(const char* s1, long m, const char* s2, long n, int* len, char* lcs, char* g1, char* g2, long cap) {
  const LcsSegmentResult r = compute_lcs_for_segment(std::string(s1, (size_t)m), std::string(s2, (size_t)n));
  *len = r.lcs_length;
  put(r.lcs_string, lcs, cap);
  put(r.gapped_seq1, g1, cap);
  put(r.gapped_seq2, g2, cap);
  return 0;
}

// out[0..4] = score, query_end_orig, target_end_orig, query_start_orig, target_start_orig
int ref_win_sw(const char* s1, long m, const char* s2, long n, int mode, double go, double ge, int q_off, int t_off, int* out,
               char* a1, char* a2, long cap) {
  try {
    const AlignmentResult r = perform_sw_in_window(std::string(s1, (size_t)m), std::string(s2, (size_t)n), fn_of(mode), go, ge, q_off, t_off);
    out[0] = r.score;
    out[1] = r.query_end_orig;
    out[2] = r.target_end_orig;
    out[3] = r.query_start_orig;
    out[4] = r.target_start_orig;
    put(r.aligned_seq1, a1, cap);
    put(r.aligned_seq2, a2, cap);
    return 0;
  } catch (const std::exception&) {
    return 1;
  }
}

}  // extern "C"
