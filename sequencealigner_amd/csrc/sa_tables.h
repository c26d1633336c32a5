// sa_tables.h -- the reference's two built-in substitution schemes, restated as int8
// tables (the reference stores them as double, src/EDNAFULL.h:33, src/EBLOSUM62.h:41;
// every entry is integral so int8 is exact).
//
// Quirk Q1 (kept on purpose): the reference indexes the NUC.4.4 table,
// whose rows are in the order A T G C S W R Y K M B V H D N (EDNAFULL.h:11-26), with a
// residue map in the order A C G T R Y S W K M B D H V N (main.cpp:60-76).  The table
// below is the NUC.4.4 matrix in ITS file order and dna_idx256() is the reference's map,
// so sigma(x,y) = table[map[x]][map[y]] reproduces the reference literally.
#pragma once
#include <cstdint>
#include <cstdlib>
#include <cstring>

namespace sa_tables {

// NCBI NUC.4.4, file order ATGCSWRYKMBVHDN
static const char* const kNuc44Text =
    " 5 -4 -4 -4 -4  1  1 -4 -4  1 -4 -1 -1 -1 -2 "
    "-4  5 -4 -4 -4  1 -4  1  1 -4 -1 -4 -1 -1 -2 "
    "-4 -4  5 -4  1 -4  1 -4  1 -4 -1 -1 -4 -1 -2 "
    "-4 -4 -4  5  1 -4 -4  1 -4  1 -1 -1 -1 -4 -2 "
    "-4 -4  1  1 -1 -4 -2 -2 -2 -2 -1 -1 -3 -3 -1 "
    " 1  1 -4 -4 -4 -1 -2 -2 -2 -2 -3 -3 -1 -1 -1 "
    " 1 -4  1 -4 -2 -2 -1 -4 -2 -2 -3 -1 -3 -1 -1 "
    "-4  1 -4  1 -2 -2 -4 -1 -2 -2 -1 -3 -1 -3 -1 "
    "-4  1  1 -4 -2 -2 -2 -2 -1 -4 -1 -3 -3 -1 -1 "
    " 1 -4 -4  1 -2 -2 -2 -2 -4 -1 -3 -1 -1 -3 -1 "
    "-4 -1 -1 -1 -1 -3 -3 -1 -1 -3 -1 -2 -2 -2 -1 "
    "-1 -4 -1 -1 -1 -3 -1 -3 -3 -1 -2 -1 -2 -2 -1 "
    "-1 -1 -4 -1 -3 -1 -3 -1 -3 -1 -2 -2 -1 -2 -1 "
    "-1 -1 -1 -4 -3 -1 -1 -3 -1 -3 -2 -2 -2 -1 -1 "
    "-2 -2 -2 -2 -1 -1 -1 -1 -1 -1 -1 -1 -1 -1 -1 ";

// NCBI BLOSUM62, file order ARNDCQEGHILKMFPSTWYVBZX*
static const char* const kBlosum62Text =
    " 4 -1 -2 -2  0 -1 -1  0 -2 -1 -1 -1 -1 -2 -1  1  0 -3 -2  0 -2 -1  0 -4 "
    "-1  5  0 -2 -3  1  0 -2  0 -3 -2  2 -1 -3 -2 -1 -1 -3 -2 -3 -1  0 -1 -4 "
    "-2  0  6  1 -3  0  0  0  1 -3 -3  0 -2 -3 -2  1  0 -4 -2 -3  3  0 -1 -4 "
    "-2 -2  1  6 -3  0  2 -1 -1 -3 -4 -1 -3 -3 -1  0 -1 -4 -3 -3  4  1 -1 -4 "
    " 0 -3 -3 -3  9 -3 -4 -3 -3 -1 -1 -3 -1 -2 -3 -1 -1 -2 -2 -1 -3 -3 -2 -4 "
    "-1  1  0  0 -3  5  2 -2  0 -3 -2  1  0 -3 -1  0 -1 -2 -1 -2  0  3 -1 -4 "
    "-1  0  0  2 -4  2  5 -2  0 -3 -3  1 -2 -3 -1  0 -1 -3 -2 -2  1  4 -1 -4 "
    " 0 -2  0 -1 -3 -2 -2  6 -2 -4 -4 -2 -3 -3 -2  0 -2 -2 -3 -3 -1 -2 -1 -4 "
    "-2  0  1 -1 -3  0  0 -2  8 -3 -3 -1 -2 -1 -2 -1 -2 -2  2 -3  0  0 -1 -4 "
    "-1 -3 -3 -3 -1 -3 -3 -4 -3  4  2 -3  1  0 -3 -2 -1 -3 -1  3 -3 -3 -1 -4 "
    "-1 -2 -3 -4 -1 -2 -3 -4 -3  2  4 -2  2  0 -3 -2 -1 -2 -1  1 -4 -3 -1 -4 "
    "-1  2  0 -1 -3  1  1 -2 -1 -3 -2  5 -1 -3 -1  0 -1 -3 -2 -2  0  1 -1 -4 "
    "-1 -1 -2 -3 -1  0 -2 -3 -2  1  2 -1  5  0 -2 -1 -1 -1 -1  1 -3 -1 -1 -4 "
    "-2 -3 -3 -3 -2 -3 -3 -3 -1  0  0 -3  0  6 -4 -2 -2  1  3 -1 -3 -3 -1 -4 "
    "-1 -2 -2 -1 -3 -1 -1 -2 -2 -3 -3 -1 -2 -4  7 -1 -1 -4 -3 -2 -2 -1 -2 -4 "
    " 1 -1  1  0 -1  0  0  0 -1 -2 -2  0 -1 -2 -1  4  1 -3 -2 -2  0  0  0 -4 "
    " 0 -1  0 -1 -1 -1 -1 -2 -2 -1 -1 -1 -1 -2 -1  1  5 -2 -2  0 -1 -1  0 -4 "
    "-3 -3 -4 -4 -2 -2 -3 -2 -2 -3 -2 -3 -1  1 -4 -3 -2 11  2 -3 -4 -3 -2 -4 "
    "-2 -2 -2 -3 -2 -1 -2 -3  2 -1 -1 -2 -1  3 -3 -2 -2  2  7 -1 -3 -2 -1 -4 "
    " 0 -3 -3 -3 -1 -2 -2 -3 -3  3  1 -2  1 -1 -2 -2  0 -3 -1  4 -3 -2 -1 -4 "
    "-2 -1  3  4 -3  0  1 -1  0 -3 -4  0 -3 -3 -2  0 -1 -4 -3 -3  4  1 -1 -4 "
    "-1  0  0  1 -3  3  4 -2  0 -3 -3  1 -1 -3 -1  0 -1 -3 -2 -2  1  4 -1 -4 "
    " 0 -1 -1 -1 -2 -1 -1 -1 -1 -1 -1 -1 -1 -1 -2  0  0 -2 -1 -1 -1 -1 -1 -4 "
    "-4 -4 -4 -4 -4 -4 -4 -4 -4 -4 -4 -4 -4 -4 -4 -4 -4 -4 -4 -4 -4 -4 -4  1 ";

// whitespace-separated integers, alpha*alpha of them
inline void parse_table(const char* text, int alpha, int8_t* out) {
  const char* p = text;
  for (int k = 0; k < alpha * alpha; ++k) {
    char* end = nullptr;
    out[k] = (int8_t)strtol(p, &end, 10);
    p = end;
  }
}

inline void fill_map(uint8_t* map, const char* order) {
  memset(map, 255, 256);
  for (int i = 0; order[i]; ++i) map[(unsigned char)order[i]] = (uint8_t)i;
}

inline void dna_idx256(uint8_t* map) {
  fill_map(map, "ACGTRYSWKMBDHVN");  // main.cpp:60-75
  map['U'] = 3;                      // main.cpp:64
  map['X'] = 14;                     // main.cpp:76
}
inline void prot_idx256(uint8_t* map) { fill_map(map, "ARNDCQEGHILKMFPSTWYVBZX*"); }  // main.cpp:84-107

}  // namespace sa_tables
