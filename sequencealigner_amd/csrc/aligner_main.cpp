// aligner_main.cpp -- the `aligner` CLI of the reference (main.cpp:1720-1839), same flags, same
// exit codes, same result files; the DP runs on one MI355X through libseqalign_hip.so.
// No MPI: one process drives the GPU.
//
// Extension: --batch treats --query / --target as multi-FASTA files with the same
// number of records; record k of one is aligned with record k of the other and the usual result
// files go to <outdir>/pair_<k>/ (k from 1), plus <outdir>/batch_summary.tsv.  --choice 2 runs all
// pairs through one sa_sw_batch_align call; the other choices loop over the pairs.
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <climits>
#include <cmath>
#include <cstring>
#include <filesystem>
#include <iostream>
#include <stdexcept>
#include <string>
#include <vector>

#include "../../include/seqalign_hip.h"
#include "../../include/seqalign_host.h"

namespace {

struct Options {
  std::string query, target, outdir = ".";
  int choice = -1;
  int mode = SA_MODE_DNA;
  bool verbose = false, binary = false, txt = false;
  double gap_open = -5.0, gap_extend = -1.0;  // main.cpp:41-43
  int device = 0;
  bool batch = false;
};

struct Seq {
  std::string header, residues;
};

Seq read_fasta(const std::string& path) {
  char *h = nullptr, *s = nullptr;
  int64_t len = 0;
  if (sah_read_fasta(path.c_str(), &h, &s, &len) != 0) throw std::runtime_error("Error: Unable to open " + path);
  Seq r{h, std::string(s, (size_t)len)};
  sah_free(h);
  sah_free(s);
  return r;
}

// The reference reports bad residues by throwing from its score function (main.cpp:120-121);
// main() catches, prints "Error: <what>" and still exits 0 (main.cpp:1832-1838).
[[noreturn]] void throw_engine_error(sa_ctx* ctx, int rc, const sa_result& r, int mode) {
  if (rc == SA_ERR_INVALID_CODE) {
    std::string msg = mode == SA_MODE_DNA ? "Invalid DNA code: " : "Invalid protein code: ";
    msg += (char)r.bad_x;
    msg += ",";
    msg += (char)r.bad_y;
    throw std::runtime_error(msg);
  }
  throw std::runtime_error(std::string("GPU engine: ") + sa_last_error(ctx));
}

void run_one(sa_ctx* ctx, int which, const Options& o, const Seq& q, const Seq& t) {
  sa_scoring sc = *sa_builtin_scoring(o.mode);
  sc.gap_open = o.gap_open;
  sc.gap_extend = o.gap_extend;
  const uint8_t* x = (const uint8_t*)q.residues.data();
  const uint8_t* y = (const uint8_t*)t.residues.data();
  const int64_t m = (int64_t)q.residues.size(), n = (int64_t)t.residues.size();
  const uint32_t flags = (o.binary || o.txt) ? SA_FLAG_KEEP_DP : 0u;
  sa_result r;
  const auto t0 = std::chrono::high_resolution_clock::now();
  int rc;
  if (which == 1)
    rc = sa_nw(ctx, x, m, y, n, &sc, flags, &r);
  else if (which == 2)
    rc = sa_sw(ctx, x, m, y, n, &sc, flags, &r);
  else
    rc = sa_lcs(ctx, x, m, y, n, flags, &r);
  const auto t1 = std::chrono::high_resolution_clock::now();
  if (rc != SA_OK) throw_engine_error(ctx, rc, r, o.mode);
  sah_job job{};
  job.outdir = o.outdir.c_str();
  job.header1 = q.header.c_str();
  job.header2 = t.header.c_str();
  job.mode = o.mode;
  job.gap_open = o.gap_open;
  job.gap_extend = o.gap_extend;
  job.x = x;
  job.m = m;
  job.y = y;
  job.n = n;
  job.score = r.score;
  job.end_i = r.end_i;
  job.end_j = r.end_j;
  job.ops = r.ops;
  job.n_ops = r.n_ops;
  job.time_ms = std::chrono::duration_cast<std::chrono::milliseconds>(t1 - t0).count();
  job.verbose = o.verbose;
  job.binary = o.binary;
  job.txt = o.txt;
  job.dp = r.dp;
  if (which == 1)
    sah_write_global(&job);
  else if (which == 2)
    sah_write_local(&job);
  else
    sah_write_lcs(&job);
  sa_free_result(&r);
}

std::vector<Seq> read_records(const std::string& path) {
  int64_t n = 0;
  char **h = nullptr, **s = nullptr;
  int64_t* len = nullptr;
  if (sah_read_fasta_records(path.c_str(), &n, &h, &s, &len) != 0) throw std::runtime_error("Error: Unable to open " + path);
  std::vector<Seq> out;
  for (int64_t k = 0; k < n; ++k) out.push_back(Seq{h[k], std::string(s[k], (size_t)len[k])});
  sah_free_records(n, h, s, len);
  return out;
}

void run_batch_mode(sa_ctx* ctx, const Options& o) {
  const std::vector<Seq> qs = read_records(o.query), ts = read_records(o.target);
  if (qs.size() != ts.size())
    throw std::runtime_error("--batch needs the same number of records in --query (" + std::to_string(qs.size()) + ") and --target (" +
                             std::to_string(ts.size()) + ")");
  const int64_t np = (int64_t)qs.size();
  auto pair_dir = [&](int64_t k) { return o.outdir + "/pair_" + std::to_string(k + 1); };
  for (int64_t k = 0; k < np; ++k) std::filesystem::create_directories(pair_dir(k));
  FILE* sum = fopen((o.outdir + "/batch_summary.tsv").c_str(), "w");
  if (!sum) throw std::runtime_error("Error: Unable to open " + o.outdir + "/batch_summary.tsv");
  fprintf(sum, "pair\tqueryid\ttargetid\tquery_length\ttarget_length\tmethod\tscore\tend_i\tend_j\tstart_i\tstart_j\taligned_length\n");
  auto ids = [&](int64_t k, char (&a)[256], char (&b)[256]) {
    sah_accession(qs[k].header.c_str(), o.mode, a, sizeof a);
    sah_accession(ts[k].header.c_str(), o.mode, b, sizeof b);
  };
  // The many-pair kernels take non-positive penalties (fractional ones included); anything else (the reference accepts
  // any double, main.cpp:41-43) goes pair by pair through the single-pair entry points below, for every method alike.
  const bool batch_ok = o.gap_open <= 0 && o.gap_extend <= 0 && std::fabs(o.gap_open) <= 1e4 && std::fabs(o.gap_extend) <= 1e4;
  if ((o.choice == 2 || o.choice == 4) && batch_ok) {
    // one engine call for every pair
    std::string pool;
    std::vector<int64_t> ox(np), oy(np), ops_off(np);
    std::vector<int32_t> lx(np), ly(np), score(np), ei(np), ej(np), si(np), sj(np), nops(np);
    for (int64_t k = 0; k < np; ++k) {
      ox[k] = (int64_t)pool.size();
      lx[k] = (int32_t)qs[k].residues.size();
      pool += qs[k].residues;
      oy[k] = (int64_t)pool.size();
      ly[k] = (int32_t)ts[k].residues.size();
      pool += ts[k].residues;
    }
    sa_scoring sc = *sa_builtin_scoring(o.mode);
    sc.gap_open = o.gap_open;
    sc.gap_extend = o.gap_extend;
    uint8_t* ops_pool = nullptr;
    int64_t pool_bytes = 0;
    float kms = 0;
    const auto t0 = std::chrono::high_resolution_clock::now();
    const int rc = sa_sw_batch_align(ctx, (const uint8_t*)pool.data(), ox.data(), lx.data(), oy.data(), ly.data(), np, &sc, score.data(),
                                     ei.data(), ej.data(), si.data(), sj.data(), ops_off.data(), nops.data(), &ops_pool, &pool_bytes, &kms);
    const auto t1 = std::chrono::high_resolution_clock::now();
    if (rc != SA_OK && rc != SA_ERR_INVALID_CODE) throw std::runtime_error(std::string("GPU engine: ") + sa_last_error(ctx));
    const int64_t ms_each = np ? std::chrono::duration_cast<std::chrono::milliseconds>(t1 - t0).count() / np : 0;
    for (int64_t k = 0; k < np; ++k) {
      char a[256], b[256];
      ids(k, a, b);
      if (score[k] == INT32_MIN) {  // what the reference prints for this pair (main.cpp:120-121, 1832-1838); no files
        std::cerr << "Error: pair " << (k + 1) << ": invalid residue for --mode " << (o.mode == SA_MODE_DNA ? "dna" : "protein") << "\n";
        fprintf(sum, "%lld\t%s\t%s\t%d\t%d\tlocal\tERROR\t\t\t\t\t\n", (long long)k + 1, a, b, lx[k], ly[k]);
        continue;
      }
      const std::string dir = pair_dir(k);
      sah_job job{};
      job.outdir = dir.c_str();
      job.header1 = qs[k].header.c_str();
      job.header2 = ts[k].header.c_str();
      job.mode = o.mode;
      job.gap_open = o.gap_open;
      job.gap_extend = o.gap_extend;
      job.x = (const uint8_t*)pool.data() + ox[k];
      job.m = lx[k];
      job.y = (const uint8_t*)pool.data() + oy[k];
      job.n = ly[k];
      job.score = score[k];
      job.end_i = ei[k];
      job.end_j = ej[k];
      job.ops = ops_pool + ops_off[k];
      job.n_ops = nops[k];
      job.time_ms = ms_each;
      job.verbose = o.verbose;
      sah_write_local(&job);
      fprintf(sum, "%lld\t%s\t%s\t%d\t%d\tlocal\t%d\t%d\t%d\t%d\t%d\t%d\n", (long long)k + 1, a, b, lx[k], ly[k], score[k], ei[k], ej[k],
              si[k], sj[k], nops[k]);
    }
    sa_free_pool(ops_pool);
  }
  if ((o.choice == 1 || o.choice == 4) && batch_ok) {
    // global alignments: one engine call for every pair (integral penalties; otherwise the loop below)
    std::string pool;
    std::vector<int64_t> ox(np), oy(np), ops_off(np);
    std::vector<int32_t> lx(np), ly(np), score(np), nops(np);
    for (int64_t k = 0; k < np; ++k) {
      ox[k] = (int64_t)pool.size();
      lx[k] = (int32_t)qs[k].residues.size();
      pool += qs[k].residues;
      oy[k] = (int64_t)pool.size();
      ly[k] = (int32_t)ts[k].residues.size();
      pool += ts[k].residues;
    }
    sa_scoring sc = *sa_builtin_scoring(o.mode);
    sc.gap_open = o.gap_open;
    sc.gap_extend = o.gap_extend;
    uint8_t* ops_pool = nullptr;
    int64_t pool_bytes = 0;
    float kms = 0;
    const auto t0 = std::chrono::high_resolution_clock::now();
    const int rc = sa_nw_batch_align(ctx, (const uint8_t*)pool.data(), ox.data(), lx.data(), oy.data(), ly.data(), np, &sc, score.data(),
                                     ops_off.data(), nops.data(), &ops_pool, &pool_bytes, &kms);
    const auto t1 = std::chrono::high_resolution_clock::now();
    if (rc != SA_OK && rc != SA_ERR_INVALID_CODE) throw std::runtime_error(std::string("GPU engine: ") + sa_last_error(ctx));
    const int64_t ms_each = np ? std::chrono::duration_cast<std::chrono::milliseconds>(t1 - t0).count() / np : 0;
    for (int64_t k = 0; k < np; ++k) {
      char a[256], b[256];
      ids(k, a, b);
      if (score[k] == INT32_MIN) {
        std::cerr << "Error: pair " << (k + 1) << ": invalid residue for --mode " << (o.mode == SA_MODE_DNA ? "dna" : "protein") << "\n";
        fprintf(sum, "%lld\t%s\t%s\t%d\t%d\tglobal\tERROR\t\t\t\t\t\n", (long long)k + 1, a, b, lx[k], ly[k]);
        continue;
      }
      const std::string dir = pair_dir(k);
      sah_job job{};
      job.outdir = dir.c_str();
      job.header1 = qs[k].header.c_str();
      job.header2 = ts[k].header.c_str();
      job.mode = o.mode;
      job.gap_open = o.gap_open;
      job.gap_extend = o.gap_extend;
      job.x = (const uint8_t*)pool.data() + ox[k];
      job.m = lx[k];
      job.y = (const uint8_t*)pool.data() + oy[k];
      job.n = ly[k];
      job.score = score[k];
      job.end_i = lx[k];
      job.end_j = ly[k];
      job.ops = ops_pool + ops_off[k];
      job.n_ops = nops[k];
      job.time_ms = ms_each;
      job.verbose = o.verbose;
      sah_write_global(&job);
      fprintf(sum, "%lld\t%s\t%s\t%d\t%d\tglobal\t%d\t%d\t%d\t0\t0\t%d\n", (long long)k + 1, a, b, lx[k], ly[k], score[k], lx[k], ly[k], nops[k]);
    }
    sa_free_pool(ops_pool);
  }
  for (int which : {2, 1, 3}) {
    if (which != 3 && batch_ok) continue;  // done above
    if (o.choice != which && o.choice != 4) continue;
    for (int64_t k = 0; k < np; ++k) {
      Options ok = o;
      ok.outdir = pair_dir(k);
      ok.binary = ok.txt = false;
      try {
        run_one(ctx, which, ok, qs[k], ts[k]);
      } catch (const std::exception& e) {
        std::cerr << "Error: pair " << (k + 1) << ": " << e.what() << "\n";
      }
    }
  }
  fclose(sum);
}

// `mpirun -np P ./aligner ...` keeps working: the reference's ranks > 0 only help rank 0 with
// localalign's rows and otherwise repeat rank 0's work and rewrite its files (main.cpp:1050-1147).
// Here one process drives the GPU, so a rank > 0 (as the common launchers export it) has nothing to do.
int launcher_rank() {
  for (const char* name : {"OMPI_COMM_WORLD_RANK", "PMI_RANK", "PMIX_RANK", "MV2_COMM_WORLD_RANK"}) {
    const char* v = std::getenv(name);
    if (v && *v) return std::atoi(v);
  }
  return 0;
}

}  // namespace

int main(int argc, char** argv) {
  sa_ctx* ctx = nullptr;
  // SA_TIMING=1: where a cold process spends its time (stderr), cf. bench.py small_pairs
  const bool timing = std::getenv("SA_TIMING") != nullptr;
  const auto tp0 = std::chrono::steady_clock::now();
  auto lap = [&](const char* what) {
    if (timing) std::fprintf(stderr, "[cli_timing] %-24s %8.3f ms since main\n", what, std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - tp0).count());
  };
  try {
    Options o;
    for (int i = 1; i < argc; ++i) {  // main.cpp:1731-1776
      const std::string arg = argv[i];
      if (arg == "--query" && i + 1 < argc) {
        o.query = argv[++i];
      } else if (arg == "--target" && i + 1 < argc) {
        o.target = argv[++i];
      } else if (arg == "--choice" && i + 1 < argc) {
        o.choice = std::stoi(argv[++i]);
      } else if (arg == "--mode" && i + 1 < argc) {
        const std::string v = argv[++i];
        if (v == "dna") {
          o.mode = SA_MODE_DNA;
        } else if (v == "protein") {
          o.mode = SA_MODE_PROTEIN;
        } else {
          std::cerr << "Unknown mode: " << v << "\n";
          return 1;
        }
      } else if (arg == "--outdir" && i + 1 < argc) {
        o.outdir = argv[++i];
      } else if (arg == "--verbose") {
        o.verbose = true;
      } else if (arg == "--binary") {
        o.binary = true;
      } else if (arg == "--txt") {
        o.txt = true;
      } else if (arg == "--help") {
        std::cout << "Usage: ./aligner --query <file1> --target <file2> --choice "
                     "<1|2|3|4> [--mode dna|protein] [--outdir DIR] [--verbose]\n"
                     "  --choice: 1=global, 2=local, 3=LCS, 4=all\n";
        return 0;
      } else if (arg == "--gap_open" && i + 1 < argc) {
        o.gap_open = std::stod(argv[++i]);
      } else if (arg == "--gap_extend" && i + 1 < argc) {
        o.gap_extend = std::stod(argv[++i]);
      } else if (arg == "--device" && i + 1 < argc) {  // extension: which GPU
        o.device = std::stoi(argv[++i]);
      } else if (arg == "--batch") {  // extension: multi-FASTA, record k vs record k
        o.batch = true;
      } else {
        std::cerr << "Unknown option: " << arg << "\n";
        return 1;
      }
    }
    if (o.query.empty() || o.target.empty() || o.choice == -1) {
      std::cerr << "Missing required arguments: --query, --target, --choice\n";
      return 1;
    }
    if (launcher_rank() > 0) return 0;
    std::filesystem::create_directories(o.outdir);
    if (o.batch && o.choice >= 1 && o.choice <= 4) {
      ctx = sa_create(o.device);
      if (!ctx) throw std::runtime_error("no gfx950 GPU available (this build has no CPU path)");
      run_batch_mode(ctx, o);
      sa_destroy(ctx);
      return 0;
    }
    const Seq q = read_fasta(o.query), t = read_fasta(o.target);
    lap("args + FASTA read");
    if (o.choice >= 1 && o.choice <= 4) {
      ctx = sa_create(o.device);
      if (!ctx) throw std::runtime_error("no gfx950 GPU available (this build has no CPU path)");
      lap("sa_create");
      if (o.choice == 4) {
        run_one(ctx, 1, o, q, t);
        run_one(ctx, 2, o, q, t);
        run_one(ctx, 3, o, q, t);
      } else {
        run_one(ctx, o.choice, o, q, t);
      }
      lap("align + result files");
    } else {
      std::cerr << "Invalid method. Use --choice 1/2/3/4.\n";
    }
  } catch (const std::exception& e) {
    std::cerr << "Error: " << e.what() << "\n";  // and exit 0, like the reference
  }
  if (ctx) sa_destroy(ctx);
  lap("sa_destroy");
  return 0;
}
