// sa_host.cpp -- host half of the drop-in (no HIP): FASTA input, sequence IDs and the
// reference's result files, rebuilt byte for byte from the op list the GPU engine returns.
// Interface and the reference lines each function stands in for: include/seqalign_host.h.
#include "../../include/seqalign_host.h"

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <chrono>
#include <cstring>
#include <fstream>
#include <iomanip>
#include <iostream>
#include <sstream>
#include <string>
#include <vector>

namespace {

const int kLineWidth = 80;  // main.cpp:45

std::string accession(const std::string& h, int mode) {
  if (mode == 0) {  // DNA: first whitespace-delimited token
    std::istringstream is(h);
    std::string tok;
    is >> tok;
    return tok;
  }
  const size_t a = h.find('|');
  const size_t b = h.find('|', a + 1);  // npos + 1 wraps to 0, exactly as in the reference
  if (a != std::string::npos && b != std::string::npos) return h.substr(a + 1, b - a - 1);
  return h;
}

std::string gene_symbol(const std::string& h, int mode) {
  if (mode == 0) {
    const size_t o = h.find('(');
    const size_t c = o == std::string::npos ? std::string::npos : h.find(')', o + 1);
    if (o != std::string::npos && c != std::string::npos && c > o + 1) return h.substr(o + 1, c - o - 1);
    return "";
  }
  const size_t a = h.find('|');
  const size_t b = a == std::string::npos ? std::string::npos : h.find('|', a + 1);
  if (b != std::string::npos) {
    const size_t u = h.find('_', b + 1);
    if (u != std::string::npos && u > b + 1) return h.substr(b + 1, u - b - 1);
  }
  const size_t g = h.find("GN=");
  if (g != std::string::npos) {
    const size_t s = g + 3;
    size_t e = h.find_first_of(" ;", s);
    if (e == std::string::npos) e = h.size();
    if (e > s) return h.substr(s, e - s);
  }
  return "";
}

struct Aligned {
  std::string ax, ay;  // start -> end
  std::string lcs;     // residues on diagonal moves, start -> end
  size_t gaps = 0, matches = 0;
};

// Replays the op list from the end cell: the strings the reference builds while tracing back
// (main.cpp:781-795) and then reverses (main.cpp:855-856).
Aligned replay(const sah_job& j) {
  Aligned r;
  const int64_t L = j.n_ops;
  r.ax.resize((size_t)L);
  r.ay.resize((size_t)L);
  int64_t ci = j.end_i, cj = j.end_j;
  for (int64_t k = 0; k < L; ++k) {
    const size_t pos = (size_t)(L - 1 - k);
    const uint8_t op = j.ops[k];
    if (op == 0) {
      r.ax[pos] = (char)j.x[ci - 1];
      r.ay[pos] = (char)j.y[cj - 1];
      --ci, --cj;
    } else if (op == 1) {
      r.ax[pos] = '-';
      r.ay[pos] = (char)j.y[cj - 1];
      --cj;
    } else {
      r.ax[pos] = (char)j.x[ci - 1];
      r.ay[pos] = '-';
      --ci;
    }
  }
  for (size_t k = 0; k < (size_t)L; ++k) {
    if (r.ax[k] == '-' || r.ay[k] == '-')
      ++r.gaps;
    else if (r.ax[k] == r.ay[k])
      ++r.matches;
  }
  return r;
}

// "<j> <i>\n" per visited cell (column first, main.cpp:835).  forward=false: end -> start as the
// global writer does; forward=true: start -> end (local main.cpp:1262, lcs main.cpp:1640).
bool write_path(const std::string& file, const sah_job& j, bool forward) {
  FILE* f = fopen(file.c_str(), "w");
  if (!f) return false;
  std::vector<char> buf(1 << 20);
  setvbuf(f, buf.data(), _IOFBF, buf.size());
  const int64_t L = j.n_ops;
  if (!forward) {
    int64_t ci = j.end_i, cj = j.end_j;
    fprintf(f, "%lld %lld\n", (long long)cj, (long long)ci);
    for (int64_t k = 0; k < L; ++k) {
      const uint8_t op = j.ops[k];
      if (op != 1) --ci;
      if (op != 2) --cj;
      fprintf(f, "%lld %lld\n", (long long)cj, (long long)ci);
    }
  } else {
    int64_t ci = j.end_i, cj = j.end_j;
    for (int64_t k = 0; k < L; ++k) {
      if (j.ops[k] != 1) --ci;
      if (j.ops[k] != 2) --cj;
    }
    fprintf(f, "%lld %lld\n", (long long)cj, (long long)ci);
    for (int64_t k = L - 1; k >= 0; --k) {
      if (j.ops[k] != 1) ++ci;
      if (j.ops[k] != 2) ++cj;
      fprintf(f, "%lld %lld\n", (long long)cj, (long long)ci);
    }
  }
  fclose(f);
  return true;
}

void write_plain_alignment(std::ostream& os, const std::string& id1, const std::string& id2, const std::string& a1,
                           const std::string& a2) {  // main.cpp:319-329: unwrapped lines
  os << '>' << id1 << '\n' << a1 << '\n' << '>' << id2 << '\n' << a2 << '\n';
}

// The reference redraws a progress bar every 100 rows of its fill loop under --verbose (showProgressBar,
// main.cpp:159-201; call sites :757-761, :1130-1136, :1589-1591), each redraw starting with '\r'.  Here the fill is
// one GPU launch, so only the last state is drawn -- what a terminal shows once the reference's loop is through:
// the full bar, 100 %, m/m, the elapsed time since the first bar of the process and an ETA of 00:00.
void final_progress_bar(int64_t m) {
  if (m <= 0) return;  // the reference's loop body never runs
  static const auto start_time = std::chrono::steady_clock::now();
  const long elapsed = (long)std::chrono::duration_cast<std::chrono::seconds>(std::chrono::steady_clock::now() - start_time).count();
  auto hms = [](long secs) {
    std::ostringstream os;
    if (secs / 3600) os << secs / 3600 << ":";
    os << std::setw(2) << std::setfill('0') << (secs % 3600) / 60 << ":" << std::setw(2) << std::setfill('0') << secs % 60;
    return os.str();
  };
  std::cout << "\r[" << std::string(100, '=') << "] " << std::setw(3) << 100 << "% " << m << "/" << m << " Elapsed: " << hms(elapsed)
            << " ETA: " << hms(0) << std::flush;
}

void print_colored(const std::string& a, const std::string& b, std::ostream& os) {  // main.cpp:356-414
  const char *reset = "\033[0m", *green = "\033[32m", *red = "\033[31m", *cyan = "\033[36m";
  const size_t len = a.size();
  if (len == 0) {
    os << "No alignment to print.\n";
    return;
  }
  size_t p1 = 0, p2 = 0;
  for (size_t i = 0; i < len; i += kLineWidth) {
    const size_t e = std::min(i + (size_t)kLineWidth, len);
    size_t e1 = p1, e2 = p2;
    auto colour = [&](size_t k) { return a[k] == b[k] ? green : ((a[k] == '-' || b[k] == '-') ? red : cyan); };
    os << std::setw(6) << p1 + 1 << " ";
    for (size_t k = i; k < e; ++k) {
      os << colour(k) << a[k] << reset;
      if (a[k] != '-') ++e1;
    }
    os << " " << e1 << "\n       ";
    for (size_t k = i; k < e; ++k) os << (a[k] == b[k] ? "|" : ((a[k] == '-' || b[k] == '-') ? " " : "."));
    os << "\n" << std::setw(6) << p2 + 1 << " ";
    for (size_t k = i; k < e; ++k) {
      os << colour(k) << b[k] << reset;
      if (b[k] != '-') ++e2;
    }
    os << " " << e2 << "\n\n";
    p1 = e1;
    p2 = e2;
  }
}

// DP dumps: main.cpp:422-468 (int matrices), :476-528 (LCS pointer characters)
void dump_int_matrix(const std::string& base, const int32_t* dp, int64_t rows, int64_t cols, bool binary) {
  if (binary) {
    std::ofstream out(base + ".bin", std::ios::binary);
    if (!out) {
      std::cerr << "Error: Cannot write binary DP matrix to " << base + ".bin" << "\n";
      return;
    }
    const int32_t r = (int32_t)rows, c = (int32_t)cols;
    out.write((const char*)&r, 4);
    out.write((const char*)&c, 4);
    out.write((const char*)dp, (std::streamsize)(rows * cols * 4));
  } else {
    FILE* f = fopen((base + ".txt").c_str(), "w");
    if (!f) {
      std::cerr << "Error: Cannot write DP matrix to " << base + ".txt" << "\n";
      return;
    }
    std::vector<char> buf(1 << 20);
    setvbuf(f, buf.data(), _IOFBF, buf.size());
    for (int64_t i = 0; i < rows; ++i) {
      for (int64_t k = 0; k < cols; ++k) {
        fprintf(f, "%5d", dp[i * cols + k]);
        if (k != cols - 1) fputc(' ', f);
      }
      fputc('\n', f);
    }
    fclose(f);
  }
}

void dump_lcs_pointers(const std::string& base, const sah_job& j, bool binary) {
  const int64_t rows = j.m + 1, cols = j.n + 1;
  std::vector<char> row((size_t)cols);
  auto fill_row = [&](int64_t i) {  // main.cpp:1577-1587, rebuilt from L
    row[0] = ' ';
    for (int64_t c = 1; c < cols; ++c) {
      if (i == 0)
        row[(size_t)c] = ' ';
      else if (j.x[i - 1] == j.y[c - 1])
        row[(size_t)c] = 'D';
      else
        row[(size_t)c] = j.dp[(i - 1) * cols + c] >= j.dp[i * cols + c - 1] ? 'U' : 'L';
    }
  };
  if (binary) {
    std::ofstream out(base + ".bin", std::ios::binary);
    if (!out) {
      std::cerr << "Error: Cannot open " << base + ".bin" << " for writing.\n";
      return;
    }
    const int32_t r = (int32_t)rows, c = (int32_t)cols;
    out.write((const char*)&r, 4);
    out.write((const char*)&c, 4);
    for (int64_t i = 0; i < rows; ++i) {
      fill_row(i);
      out.write(row.data(), cols);
    }
  } else {
    std::ofstream out(base + ".txt");
    if (!out) {
      std::cerr << "Error: Cannot open " << base + ".txt" << "\n";
      return;
    }
    std::string line;
    for (int64_t i = 0; i < rows; ++i) {
      fill_row(i);
      line.clear();
      for (int64_t c = 0; c < cols; ++c) {
        line.push_back(row[(size_t)c]);
        if (c + 1 < cols) line.push_back(' ');
      }
      line.push_back('\n');
      out << line;
    }
  }
}

}  // namespace

extern "C" {

void sah_free(void* p) { free(p); }

int sah_read_fasta(const char* path, char** header, char** seq, int64_t* seq_len) {
  std::ifstream f(path);
  if (!f) return -1;
  std::string line, h, s;
  bool have = false;
  while (std::getline(f, line)) {  // main.cpp:297-307: first header only, every other line appended
    if (line.empty()) continue;
    if (line[0] == '>') {
      if (!have) {
        h = line.substr(1);
        have = true;
      }
      continue;
    }
    s += line;
  }
  *header = (char*)malloc(h.size() + 1);
  memcpy(*header, h.c_str(), h.size() + 1);
  *seq = (char*)malloc(s.size() + 1);
  memcpy(*seq, s.data(), s.size());
  (*seq)[s.size()] = 0;
  *seq_len = (int64_t)s.size();
  return 0;
}

int sah_read_fasta_records(const char* path, int64_t* n_records, char*** headers, char*** seqs, int64_t** lens) {
  std::ifstream f(path);
  if (!f) return -1;
  std::vector<std::string> hs, ss;
  std::string line;
  while (std::getline(f, line)) {
    if (line.empty()) continue;
    if (line[0] == '>') {
      hs.push_back(line.substr(1));
      ss.emplace_back();
      continue;
    }
    if (ss.empty()) {  // sequence lines before any header: one unnamed record
      hs.emplace_back();
      ss.emplace_back();
    }
    ss.back() += line;
  }
  const size_t n = hs.size();
  *n_records = (int64_t)n;
  *headers = (char**)malloc((n + 1) * sizeof(char*));
  *seqs = (char**)malloc((n + 1) * sizeof(char*));
  *lens = (int64_t*)malloc((n + 1) * sizeof(int64_t));
  for (size_t k = 0; k < n; ++k) {
    (*headers)[k] = (char*)malloc(hs[k].size() + 1);
    memcpy((*headers)[k], hs[k].c_str(), hs[k].size() + 1);
    (*seqs)[k] = (char*)malloc(ss[k].size() + 1);
    memcpy((*seqs)[k], ss[k].data(), ss[k].size());
    (*seqs)[k][ss[k].size()] = 0;
    (*lens)[k] = (int64_t)ss[k].size();
  }
  return 0;
}

void sah_free_records(int64_t n_records, char** headers, char** seqs, int64_t* lens) {
  for (int64_t k = 0; k < n_records; ++k) {
    free(headers[k]);
    free(seqs[k]);
  }
  free(headers);
  free(seqs);
  free(lens);
}

void sah_accession(const char* header, int mode, char* buf, int64_t cap) {
  const std::string a = accession(header, mode);
  snprintf(buf, (size_t)cap, "%s", a.c_str());
}

void sah_gene_symbol(const char* header, int mode, char* buf, int64_t cap) {
  const std::string g = gene_symbol(header, mode);
  snprintf(buf, (size_t)cap, "%s", g.c_str());
}

int sah_write_global(const sah_job* jp) {
  const sah_job& j = *jp;
  const std::string out = j.outdir;
  const Aligned al = replay(j);
  if (j.dp) {  // main.cpp:825-831: --binary wins over --txt
    if (j.binary)
      dump_int_matrix(out + "/global_dp_matrix", j.dp, j.m + 1, j.n + 1, true);
    else if (j.txt)
      dump_int_matrix(out + "/global_dp_matrix", j.dp, j.m + 1, j.n + 1, false);
  }
  write_path(out + "/global_path.txt", j, false);
  const size_t total = al.ax.size();
  const double identity = double(al.matches) / total;  // 0/0 -> nan exactly like main.cpp:846
  const double coverage = double(total - al.gaps) / total;
  const std::string acc1 = accession(j.header1, j.mode), acc2 = accession(j.header2, j.mode);
  const std::string g1 = gene_symbol(j.header1, j.mode), g2 = gene_symbol(j.header2, j.mode);
  if (j.verbose) {  // main.cpp:757-761 (progress), :858-873
    final_progress_bar(j.m);
    std::cout << "\n\nGlobal Alignment Score: " << j.score << "\n"
              << "Gap Open: " << j.gap_open << "\nGap Extend: " << j.gap_extend << "\n"
              << "Matches: " << al.matches << "\nGaps:    " << al.gaps << "\nTotal:   " << total << "\n"
              << "Identity: " << identity * 100.0f << "%\nCoverage: " << coverage * 100.0f << "%\n"
              << "Time:    " << j.time_ms << " ms\nQuery:   " << acc1 << "\nTarget:  " << acc2 << "\n"
              << "QueryID:  " << g1 << "\nTargetID:  " << g2 << "\n\n\n";
    print_colored(al.ax, al.ay, std::cout);
  }
  {
    std::ofstream f(out + "/global_alignment.fasta");
    if (f)
      write_plain_alignment(f, acc1, acc2, al.ax, al.ay);
    else
      std::cerr << "Error: Unable to open output file global_alignment.fasta\n";
  }
  std::ofstream js(out + "/global_stats.json");
  if (!js) {
    std::cerr << "Error: cannot open global_stats.json\n";
    return -1;
  }
  // field spacing is part of the format (main.cpp:886-903; note the single leading space of gap_open)
  js << std::fixed << std::setprecision(6) << "{\n"
     << "  \"method\":      \"global\",\n"
     << " \"gap_open\":   " << j.gap_open << ",\n"
     << "  \"gap_extend\": " << j.gap_extend << ",\n"
     << "  \"score\":       " << j.score << ",\n"
     << "  \"matches\":     " << al.matches << ",\n"
     << "  \"gaps\":        " << al.gaps << ",\n"
     << "  \"total\":       " << total << ",\n"
     << "  \"identity\":    " << identity << ",\n"
     << "  \"coverage\":    " << coverage << ",\n"
     << "  \"time_ms\":     " << j.time_ms << ",\n"
     << "  \"query\":       \"" << acc1 << "\",\n"
     << "  \"target\":      \"" << acc2 << "\",\n"
     << "  \"queryid\":       \"" << g1 << "\",\n"
     << "  \"targetid\":       \"" << g2 << "\",\n"
     << "  \"query_length\": " << j.m << ",\n"
     << "  \"target_length\": " << j.n << "\n"
     << "}\n";
  return 0;
}

int sah_write_local(const sah_job* jp) {
  const sah_job& j = *jp;
  const std::string out = j.outdir;
  const Aligned al = replay(j);
  if (j.dp) {  // main.cpp:1364-1368
    if (j.binary)
      dump_int_matrix(out + "/local_dp_matrix", j.dp, j.m + 1, j.n + 1, true);
    else if (j.txt)
      dump_int_matrix(out + "/local_dp_matrix", j.dp, j.m + 1, j.n + 1, false);
  }
  // the path exists only when a traceback ran (best cell inside the matrix, main.cpp:1195, 1402)
  if (j.score > 0 && j.end_i > 0 && j.end_j > 0) {
    if (!write_path(out + "/local_path.txt", j, true))
      std::cerr << "Error: Cannot open " << out + "/local_path.txt" << " for writing.\n";
  }
  const size_t total = al.ax.size();
  const double identity = total > 0 ? double(al.matches) / total : 0.0;
  const double coverage = total > 0 ? double(total - al.gaps) / total : 0.0;
  const std::string acc1 = accession(j.header1, j.mode), acc2 = accession(j.header2, j.mode);
  const std::string g1 = gene_symbol(j.header1, j.mode), g2 = gene_symbol(j.header2, j.mode);
  if (j.verbose) {  // main.cpp:1130-1136 (progress), :1490-1511
    final_progress_bar(j.m);
    std::cout << "\n\nLocal Alignment Score: " << j.score << "\nGap Open: " << j.gap_open << "\nGap Extend: " << j.gap_extend
              << "\n";
    if (total > 0)
      std::cout << "Matches: " << al.matches << "\nGaps:    " << al.gaps << "\nTotal Aligned Length: " << total << "\n"
                << "Identity (of aligned region): " << identity * 100.0 << "%\n"
                << "Coverage (of aligned region): " << coverage * 100.0 << "%\n";
    std::cout << "Time:    " << j.time_ms << " ms\nQuery:   " << acc1 << "\nTarget:  " << acc2 << "\nQueryID: " << g1
              << "\nTargetID: " << g2 << "\n";
    print_colored(al.ax, al.ay, std::cout);
  }
  {
    std::ofstream f(out + "/local_alignment.fasta");
    if (f) write_plain_alignment(f, acc1, acc2, al.ax, al.ay);
  }
  std::ofstream js(out + "/local_stats.json");
  if (!js) return -1;
  js << std::fixed << std::setprecision(6) << "{\n"  // main.cpp:1521-1541
     << "  \"method\":   \"local\",\n"
     << "  \"gap_open\": " << j.gap_open << ",\n"
     << "  \"gap_extend\": " << j.gap_extend << ",\n"
     << "  \"score\":    " << j.score << ",\n";
  if (total > 0)
    js << "  \"matches\":  " << al.matches << ",\n"
       << "  \"gaps\":     " << al.gaps << ",\n"
       << "  \"aligned_length\":    " << total << ",\n"
       << "  \"identity\": " << identity << ",\n"
       << "  \"coverage_aligned\": " << coverage << ",\n";
  js << "  \"time_ms\":  " << j.time_ms << ",\n"
     << "  \"query\":    \"" << acc1 << "\",\n"
     << "  \"target\":   \"" << acc2 << "\",\n"
     << "  \"queryid\":  \"" << g1 << "\",\n"
     << "  \"targetid\": \"" << g2 << "\",\n"
     << "  \"query_length_original\": " << j.m << ",\n"
     << "  \"target_length_original\": " << j.n << "\n"
     << "}\n";
  return 0;
}

int sah_write_lcs(const sah_job* jp) {
  const sah_job& j = *jp;
  const std::string out = j.outdir;
  const Aligned al = replay(j);
  std::string lcs;
  {
    int64_t ci = j.end_i;
    std::string rev;
    for (int64_t k = 0; k < j.n_ops; ++k) {
      if (j.ops[k] == 0) rev.push_back((char)j.x[ci - 1]);
      if (j.ops[k] != 1) --ci;
    }
    lcs.assign(rev.rbegin(), rev.rend());
  }
  const std::string acc1 = accession(j.header1, j.mode), acc2 = accession(j.header2, j.mode);
  {
    std::ofstream f(out + "/lcs.fasta");
    if (!f) {
      std::cerr << "Error: Cannot open " << out + "/lcs.fasta for writing\n";
    } else {  // main.cpp:338-344, 1648
      f << '>' << acc1 + "_" + acc2 << "_LCS_len=" << lcs.size() << "\n";
      for (size_t i = 0; i < lcs.size(); i += 80) f << lcs.substr(i, 80) << "\n";
    }
  }
  {
    std::ofstream f(out + "/lcs_alignment.fasta");
    if (!f)
      std::cerr << "Error: Cannot open " << out + "/lcs_alignment.fasta for writing\n";
    else
      write_plain_alignment(f, acc1 + "_LCS_aligned", acc2 + "_LCS_aligned", al.ax, al.ay);
  }
  if (!write_path(out + "/lcs_path.txt", j, true))
    std::cerr << "Error: Cannot open " << out + "/lcs_path.txt" << " for writing.\n";
  if (j.dp) {  // main.cpp:1680-1691
    if (j.binary) {
      dump_lcs_pointers(out + "/lcs_traceback_pointers", j, true);
      dump_int_matrix(out + "/lcs_dp_lengths", j.dp, j.m + 1, j.n + 1, true);
    } else if (j.txt) {
      dump_lcs_pointers(out + "/lcs_traceback_pointers", j, false);
      dump_int_matrix(out + "/lcs_dp_lengths", j.dp, j.m + 1, j.n + 1, false);
    }
  }
  if (j.verbose) {  // main.cpp:1589-1593 (progress + newline), :1694-1707
    final_progress_bar(j.m);
    if (j.m > 0) std::cout << '\n';
    std::cout << "LCS Length: " << lcs.size() << "\n\n\nLCS Based Alignment:\n";
    print_colored(al.ax, al.ay, std::cout);
    std::cout << "\nLongest Common Subsequence String:\n";
    for (size_t k = 0; k < lcs.size(); k += kLineWidth) std::cout << lcs.substr(k, kLineWidth) << "\n";
    std::cout << "\n";
  }
  return 0;
}

}  // extern "C"
