// main.cpp — `swarm` command-line front end over the C ABI (swa_cli_main; the executable is host/launcher.cpp) (drop-in for the reference CLI:
// same options, same FASTA input, same output files, same log lines; src/swarm.cc:96-124,
// 269-463, 486-630).  Everything compute-heavy goes through libswarm_amd.so to the GPU; the
// greedy clustering and the writers are the host code in cluster_d1.cpp / cluster_dn.cpp.
//
//   d = 1      swa_d1_index_build + swa_d1_network -> swa_d1_cluster [-> swa_d1_fastidious -> graft]
//   d >= 2     swa_dn_cluster (the whole graph of pairs within d at once, or one swa_scan_step per swarm generation)
//   d = 0      swa_derep -> swa_d0_cluster
//
// Environment (no new flags, to stay drop-in): SWARM_AMD_DEVICE = HIP device ordinal (default 0);
// SWARM_AMD_DEVICES = 0,1,2,... : d >= 1 on several GPUs (swa_multi_*: one rank per entry, RCCL exchange).
#include <chrono>
#include <future>
#include <thread>
#include "../../../include/swarm_amd.h"
#include "../../../include/swarm_amd_host.h"
#include "pool.h"
#include "hostdb.h"

#include <getopt.h>
#include <sys/resource.h>
#include <omp.h>
#include <unistd.h>

#include <algorithm>
#include <cinttypes>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <numeric>
#include <string>
#include <vector>

namespace {

struct Options {
  int64_t threads = 1, bloom_bits = 16, differences = 1, mismatch_penalty = 4, match_reward = 5;
  int64_t gap_open = 12, gap_extend = 4, ceiling = 0, append_abundance = 0, boundary = 3;
  bool help = false, version = false, fastidious = false, usearch = false, mothur = false, no_break = false,
       disable_sse3 = false;
  std::string input = "-", network, structure, seeds, stats, uclust, output = "-", log;
  int64_t pen_mismatch = 18, pen_gapopen = 24, pen_gapextend = 13;
  bool used[26] = {};
};

FILE * g_log = stderr;

[[noreturn]] void die(const std::string & msg) {
  std::fprintf(stderr, "\nError: %s\n", msg.c_str());
  std::exit(EXIT_FAILURE);
}

[[noreturn]] void die_raw(const std::string & text) {       // text already carries "\nError: ..."
  std::fputs(text.c_str(), stderr);
  if (text.empty() || text.back() != '\n') { std::fputc('\n', stderr); }
  std::exit(EXIT_FAILURE);
}

const char * kVersionText =
    "Swarm 3.1.6 interface — swarm-amd (MI355X / gfx950 neighbour-finding back end)\n"
    "Drop-in for https://github.com/torognes/swarm: same options, input and output formats.\n\n";

const char * kUsageText =
    "Usage: swarm [OPTIONS] [FASTAFILE]\n\n"
    "General options:\n"
    " -h, --help                          display this help and exit\n"
    " -t, --threads INTEGER               accepted for compatibility (the GPU does the work)\n"
    " -v, --version                       display version information and exit\n\n"
    "Clustering options:\n"
    " -d, --differences INTEGER           resolution (1)\n"
    " -n, --no-otu-breaking               never break clusters (not recommended!)\n\n"
    "Fastidious options (only when d = 1):\n"
    " -b, --boundary INTEGER              min mass of large clusters (3)\n"
    " -c, --ceiling INTEGER               max memory in MB for Bloom filter (unlim.)\n"
    " -f, --fastidious                    link nearby low-abundance swarms\n"
    " -y, --bloom-bits INTEGER            bits used per Bloom filter entry (16)\n\n"
    "Input/output options:\n"
    " -a, --append-abundance INTEGER      value to use when abundance is missing\n"
    " -i, --internal-structure FILENAME   write internal cluster structure to file\n"
    " -j, --network-file FILENAME         dump sequence network to file\n"
    " -l, --log FILENAME                  log to file, not to stderr\n"
    " -o, --output-file FILENAME          output result to file (stdout)\n"
    " -r, --mothur                        output using mothur-like format\n"
    " -s, --statistics-file FILENAME      dump cluster statistics to file\n"
    " -u, --uclust-file FILENAME          output using UCLUST-like format to file\n"
    " -w, --seeds FILENAME                write cluster representatives to FASTA file\n"
    " -z, --usearch-abundance             abundance annotation in usearch style\n\n"
    "Pairwise alignment advanced options (only when d > 1):\n"
    " -m, --match-reward INTEGER          reward for nucleotide match (5)\n"
    " -p, --mismatch-penalty INTEGER      penalty for nucleotide mismatch (4)\n"
    " -g, --gap-opening-penalty INTEGER   gap open penalty (12)\n"
    " -e, --gap-extension-penalty INTEGER gap extension penalty (4)\n"
    " -x, --disable-sse3                  accepted for compatibility (no effect on the GPU)\n\n";

int64_t number(const char * text, const char * option) {
  char * end = nullptr;
  const int64_t v = std::strtol(text, &end, 10);
  if (*end != '\0') {
    die(std::string("Invalid numeric argument for option ") + option + ".\n\n"
        "Frequent causes are:\n"
        " - a missing space between an argument and the next option,\n"
        " - a long option name not starting with a double dash\n"
        "   (swarm accepts '--help' or '-h', but not '-help')\n\n"
        "Please see 'swarm --help' for more details.");
  }
  return v;
}

Options parse(int argc, char ** argv) {
  static const struct option longopts[] = {
      {"append-abundance", required_argument, nullptr, 'a'}, {"boundary", required_argument, nullptr, 'b'},
      {"ceiling", required_argument, nullptr, 'c'}, {"differences", required_argument, nullptr, 'd'},
      {"gap-extension-penalty", required_argument, nullptr, 'e'}, {"fastidious", no_argument, nullptr, 'f'},
      {"gap-opening-penalty", required_argument, nullptr, 'g'}, {"help", no_argument, nullptr, 'h'},
      {"internal-structure", required_argument, nullptr, 'i'}, {"log", required_argument, nullptr, 'l'},
      {"network-file", required_argument, nullptr, 'j'}, {"match-reward", required_argument, nullptr, 'm'},
      {"no-otu-breaking", no_argument, nullptr, 'n'}, {"output-file", required_argument, nullptr, 'o'},
      {"mismatch-penalty", required_argument, nullptr, 'p'}, {"mothur", no_argument, nullptr, 'r'},
      {"statistics-file", required_argument, nullptr, 's'}, {"threads", required_argument, nullptr, 't'},
      {"uclust-file", required_argument, nullptr, 'u'}, {"version", no_argument, nullptr, 'v'},
      {"seeds", required_argument, nullptr, 'w'}, {"disable-sse3", no_argument, nullptr, 'x'},
      {"bloom-bits", required_argument, nullptr, 'y'}, {"usearch-abundance", no_argument, nullptr, 'z'},
      {nullptr, 0, nullptr, 0}};
  Options o;
  int ch = 0;
  while ((ch = getopt_long(argc, argv, "a:b:c:d:e:fg:hi:j:l:m:no:p:rs:t:u:vw:xy:z", longopts, nullptr)) != -1) {
    if (ch >= 'a' && ch <= 'z') {
      if (o.used[ch - 'a']) {
        const char * name = "?";
        for (const auto & lo : longopts) { if (lo.name != nullptr && lo.val == ch) { name = lo.name; break; } }
        die(std::string("Option -") + (char)ch + " or --" + name + " specified more than once.");
      }
      o.used[ch - 'a'] = true;
    }
    switch (ch) {
      case 'a': o.append_abundance = number(optarg, "-a or --append-abundance"); break;
      case 'b': o.boundary = number(optarg, "-b or --boundary"); break;
      case 'c': o.ceiling = number(optarg, "-c or --ceiling"); break;
      case 'd': o.differences = number(optarg, "-d or --differences"); break;
      case 'e': o.gap_extend = number(optarg, "-e or --gap-extension-penalty"); break;
      case 'f': o.fastidious = true; break;
      case 'g': o.gap_open = number(optarg, "-g or --gap-opening-penalty"); break;
      case 'h': o.help = true; break;
      case 'i': o.structure = optarg; break;
      case 'j': o.network = optarg; break;
      case 'l': o.log = optarg; break;
      case 'm': o.match_reward = number(optarg, "-m or --match-reward"); break;
      case 'n': o.no_break = true; break;
      case 'o': o.output = optarg; break;
      case 'p': o.mismatch_penalty = number(optarg, "-p or --mismatch-penalty"); break;
      case 'r': o.mothur = true; break;
      case 's': o.stats = optarg; break;
      case 't': o.threads = number(optarg, "-t or --threads"); break;
      case 'u': o.uclust = optarg; break;
      case 'v': o.version = true; break;
      case 'w': o.seeds = optarg; break;
      case 'x': o.disable_sse3 = true; break;
      case 'y': o.bloom_bits = number(optarg, "-y or --bloom-bits"); break;
      case 'z': o.usearch = true; break;
      default:
        std::fputs(kVersionText, stderr);
        std::fputs(kUsageText, stderr);
        std::fputc('\n', stderr);
        std::exit(EXIT_FAILURE);
    }
  }
  if (optind < argc) { o.input = argv[optind]; }
  // scoring: (2m + 2p, 2g, m + 2e) / gcd   (src/swarm.cc:466-483)
  o.pen_mismatch = 2 * o.match_reward + 2 * o.mismatch_penalty;
  o.pen_gapopen = 2 * o.gap_open;
  o.pen_gapextend = o.match_reward + 2 * o.gap_extend;
  const int64_t f = std::gcd(std::gcd(o.pen_mismatch, o.pen_gapopen), o.pen_gapextend);
  if (f != 0) { o.pen_mismatch /= f; o.pen_gapopen /= f; o.pen_gapextend /= f; }
  return o;
}

void validate(const Options & o) {            // src/swarm.cc:486-630, same order, same texts
  if (o.threads < 1 || o.threads > 512) { die("Illegal number of threads specified with -t or --threads, must be in the range 1 to 512."); }
  if (o.differences < 0 || o.differences > 255) { die("Illegal number of differences specified with -d or --differences, must be in the range 0 to 255."); }
  if (o.fastidious && o.differences != 1) { die("Fastidious mode (specified with -f or --fastidious) only works when the resolution (specified with -d or --differences) is 1."); }
  if (o.disable_sse3 && o.differences < 2) { die("Option --disable-sse3 or -x has no effect when d < 2 (SSE3 instructions are only used when d > 1)."); }
  if (!o.fastidious) {
    if (o.used['b' - 'a']) { die("Option -b or --boundary specified without -f or --fastidious."); }
    if (o.used['c' - 'a']) { die("Option -c or --ceiling specified without -f or --fastidious."); }
    if (o.used['y' - 'a']) { die("Option -y or --bloom-bits specified without -f or --fastidious."); }
  }
  if (o.differences < 2) {
    if (o.used['m' - 'a']) { die("Option -m or --match-reward specified when d < 2."); }
    if (o.used['p' - 'a']) { die("Option -p or --mismatch-penalty specified when d < 2."); }
    if (o.used['g' - 'a']) { die("Option -g or --gap-opening-penalty specified when d < 2."); }
    if (o.used['e' - 'a']) { die("Option -e or --gap-extension-penalty specified when d < 2."); }
  }
  if (o.gap_open < 0) { die("Illegal gap opening penalty specified with -g or --gap-opening-penalty, must not be negative."); }
  if (o.gap_extend < 0) { die("Illegal gap extension penalty specified with -e or --gap-extension-penalty, must not be negative."); }
  if (o.gap_open + o.gap_extend < 1) { die("Illegal gap penalties specified, the sum of the gap open and the gap extension penalty must be at least 1."); }
  if (o.match_reward < 1) { die("Illegal match reward specified with -m or --match-reward, must be at least 1."); }
  if (o.mismatch_penalty < 1) { die("Illegal mismatch penalty specified with -p or --mismatch-penalty, must be at least 1."); }
  if (o.boundary < 2) { die("Illegal boundary specified with -b or --boundary, must be at least 2."); }
  if (o.used['c' - 'a'] && (o.ceiling < 40 || o.ceiling > (1 << 30))) { die("Illegal memory ceiling specified with -c or --ceiling, must be in the range 8 to 1,073,741,824 MB."); }
  if (o.bloom_bits < 2 || o.bloom_bits > 64) { die("Illegal number of Bloom filter bits specified with -y or --bloom-bits, must be in the range 2 to 64."); }
  if (o.used[0] && o.append_abundance < 1) { die("Illegal abundance value specified with -a or --append-abundance, must be at least 1."); }
  if (!o.network.empty() && o.differences != 1) { die("A network file can only written when d = 1."); }
  if (o.version) { std::fputs(kVersionText, stderr); std::exit(EXIT_SUCCESS); }
  if (o.help) { std::fputs(kVersionText, stderr); std::fputs(kUsageText, stderr); std::exit(EXIT_SUCCESS); }
  const int64_t sat16 = std::min<int64_t>(65535 / o.pen_mismatch, (65535 - o.pen_gapopen) / o.pen_gapextend);
  if (o.differences > sat16) { die("Resolution (d) too high for the given scoring system."); }
  if (o.pen_mismatch > 255) { die("Alignment scoring system yielded a mismatch penalty greater than 255, please use different parameter values."); }
}

// a phase line of the log: "<prompt> 100%" (the reference animates a percentage on a tty)
// SWARM_AMD_TIMING=1: seconds since process start at every milestone, on stderr
void stamp(const char * what) {
  static const bool on = std::getenv("SWARM_AMD_TIMING") != nullptr;
  static const auto t0 = std::chrono::steady_clock::now();
  if (on) { std::fprintf(stderr, "[t %8.3f] %s\n", std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count(), what); }
}

// SWARM_AMD_TIMING: what the process has cost so far (CPU seconds, page faults, context switches)
void stamp_usage(const char * what) {
  if (std::getenv("SWARM_AMD_TIMING") == nullptr) { return; }
  struct rusage u{};
  getrusage(RUSAGE_SELF, &u);
  std::fprintf(stderr, "[usage] %-28s user %.3f s  sys %.3f s  minor faults %ld  major %ld  switches %ld voluntary, %ld involuntary\n", what,
               (double)u.ru_utime.tv_sec + 1e-6 * (double)u.ru_utime.tv_usec, (double)u.ru_stime.tv_sec + 1e-6 * (double)u.ru_stime.tv_usec,
               u.ru_minflt, u.ru_majflt, u.ru_nvcsw, u.ru_nivcsw);
}

void phase(const Options &, const char * prompt) {
  std::fprintf(g_log, "%s 100%%\n", prompt);
  std::fflush(g_log);
  stamp(prompt);
}

void check_writer(int rc, const char * what, const swa_d1_result * res = nullptr) {
  // (a writer that could not fetch what it prints from the device says so: not the reference's "Unable to open")
  if (rc == SWA_E_DEVICE && res != nullptr && swa_d1_result_error(res)[0] != '\0') { die(swa_d1_result_error(res)); }
  if (rc != SWA_OK) { die(std::string("Unable to open ") + what + " file for writing."); }
}

}  // namespace

// The command line proper.  The executable (host/launcher.cpp) is a few lines that set the OpenMP wait policy and load this
// library: libgomp reads OMP_WAIT_POLICY when it is loaded, and workers that SPIN between parallel regions (its default)
// fight the HIP runtime's helper threads for the cores — measured at 10 M amplicons on a 256-thread host, the same run
// takes 1.9 s with the default and 0.5 s with `passive`.
extern "C" int swa_cli_main(int argc, char ** argv) {
  stamp("start");
  // (OpenMP teams of the host phases: the CPUs this process may really use — affinity, cgroup quota: pool.h — and 32 at most)
  if (std::getenv("OMP_NUM_THREADS") == nullptr) { omp_set_num_threads(std::max(1, std::min<int>(omp_get_max_threads(), (int)std::min(swa_host_cpus(), 32u)))); }
  Options o = parse(argc, argv);
  validate(o);
  if (!o.log.empty()) {
    g_log = std::strcmp(o.log.c_str(), "-") == 0 ? stdout : std::fopen(o.log.c_str(), "w");
    if (g_log == nullptr) { die("Unable to open log file for writing."); }
  }
  std::fputs(kVersionText, g_log);
  std::fprintf(g_log, "Database file:     %s\n", o.input.c_str());
  std::fprintf(g_log, "Output file:       %s\n", o.output.c_str());
  if (!o.stats.empty()) { std::fprintf(g_log, "Statistics file:   %s\n", o.stats.c_str()); }
  if (!o.uclust.empty()) { std::fprintf(g_log, "Uclust file:       %s\n", o.uclust.c_str()); }
  if (!o.structure.empty()) { std::fprintf(g_log, "Int. struct. file  %s\n", o.structure.c_str()); }
  if (!o.network.empty()) { std::fprintf(g_log, "Network file       %s\n", o.network.c_str()); }
  std::fprintf(g_log, "Resolution (d):    %" PRId64 "\n", o.differences);
  std::fprintf(g_log, "Threads:           %" PRId64 "\n", o.threads);
  if (o.differences > 1) {
    std::fprintf(g_log, "Scores:            match: %" PRId64 ", mismatch: %" PRId64 "\n", o.match_reward, o.mismatch_penalty);
    std::fprintf(g_log, "Gap penalties:     opening: %" PRId64 ", extension: %" PRId64 "\n", o.gap_open, o.gap_extend);
    std::fprintf(g_log, "Converted costs:   mismatch: %" PRId64 ", gap opening: %" PRId64 ", gap extension: %" PRId64 "\n",
                 o.pen_mismatch, o.pen_gapopen, o.pen_gapextend);
  }
  std::fprintf(g_log, "Break clusters:    %s\n", o.no_break ? "No" : "Yes");
  if (o.fastidious) { std::fprintf(g_log, "Fastidious:        Yes, with boundary %" PRId64 "\n", o.boundary); }
  else { std::fprintf(g_log, "Fastidious:        No\n"); }
  std::fprintf(g_log, "\n");

  // The GPU context (HIP runtime start-up, ~0.25 s) is created by a helper thread while this one reads the FASTA file.
  std::vector<int> devices;
  if (const char * list = std::getenv("SWARM_AMD_DEVICES")) {
    for (const char * p = list; *p != '\0';) {
      char * end = nullptr;
      const long v = std::strtol(p, &end, 10);
      if (end == p) { break; }
      devices.push_back((int)v);
      p = (*end == ',') ? end + 1 : end;
    }
  }
  const bool use_multi = devices.size() > 1 && o.differences >= 1;
  swa_ctx * early_ctx = nullptr;
  int early_rc = SWA_OK;
  std::thread early;
  // what the reader says as soon as the packed words are final (before it sorts): the helper thread copies them to the GPU
  // while this thread sorts (swa_db_stage_words; the GPU puts them in db order later, swa_db_upload_unordered)
  struct Words { std::vector<const uint64_t *> pools; std::vector<uint64_t> counts; };
  std::promise<Words> words_promise;
  std::future<Words> words_ready = words_promise.get_future();
  bool words_told = false;
  const char * dev_env = std::getenv("SWARM_AMD_DEVICE");
  const int device = !devices.empty() ? devices[0] : (dev_env != nullptr ? std::atoi(dev_env) : 0);
  auto start_helper = [&]() {
    const int warm_for = (int)o.differences;
    early = std::thread([&early_ctx, &early_rc, &words_ready, device, warm_for]() {
      early_rc = swa_ctx_create(device, nullptr, &early_ctx);
      stamp("(helper thread) context created");
      // (the first download's set-up beside the code objects and the staging copy: SWARM_AMD_WARM_DOWNLOADS=0 leaves it to the
      // first result's download)
      std::thread downloads;
      {
        const char * wd = std::getenv("SWARM_AMD_WARM_DOWNLOADS");
        if (early_rc == SWA_OK && (wd == nullptr || wd[0] != '0')) {
          swa_ctx * c = early_ctx;
          downloads = std::thread([c]() { (void)swa_ctx_warmup_downloads(c); stamp("(second helper thread) download path warm"); });
        }
      }
      struct Join { std::thread & t; ~Join() { if (t.joinable()) { t.join(); } } } join_downloads{downloads};
      // (the anomaly hunt, tools/stress/cold_runs.sh: no warm-up = every code object loaded by the step's own first launch;
      // loading them on a thread of their own beside the copy below was tried: no gain — lease r5w)
      if (early_rc == SWA_OK && std::getenv("SWARM_AMD_NO_WARMUP") == nullptr) { (void)swa_ctx_warmup_for(early_ctx, warm_for); }
      stamp("(helper thread) code objects loaded, first copies done");
      const Words w = words_ready.get();
      if (early_rc == SWA_OK && !w.pools.empty()) {
        (void)swa_db_stage_words(early_ctx, w.pools.data(), w.counts.data(), (uint32_t)w.pools.size());   // (a failure shows at the upload)
        stamp("(helper thread) packed words on their way to the GPU");
      }
    });
  };
  if (!use_multi) { start_helper(); }

  // ---- read the database (seam L2, host side)
  swa_hostdb * db = nullptr;
  struct Tell { std::promise<Words> * promise; bool * told; } tell{&words_promise, &words_told};
  const swa_words_ready_fn on_words = [](void * user, const uint64_t * const * pools, const uint64_t * counts, uint32_t pieces) {
    auto * t = static_cast<Tell *>(user);
    Words w;
    w.pools.assign(pools, pools + pieces);
    w.counts.assign(counts, counts + pieces);
    *t->told = true;
    t->promise->set_value(std::move(w));
  };
  int rc = swa_hostdb_read_fasta_staged(o.input.c_str(), o.usearch ? 1 : 0, o.append_abundance, o.differences > 1 ? 1 : 0,
                                        use_multi ? nullptr : on_words, &tell, &db);
  if (!words_told) { words_promise.set_value(Words{}); }
  stamp("database read and ordered");
  stamp_usage("after the read");
  if (early.joinable()) { early.join(); }
  if (rc != SWA_OK) { die_raw(db != nullptr ? swa_hostdb_error(db) : "\nError: out of memory"); }
  phase(o, "Reading sequences:");
  phase(o, "Indexing database:");
  phase(o, "Abundance sorting:");
  swa_db_unordered_view uview{};
  swa_hostdb_unordered_view(db, &uview);
  std::fprintf(g_log, "Database info:     %" PRIu64 " nt in %u sequences, longest %u nt\n", swa_hostdb_nucleotides(db),
               uview.n, uview.longest);

  const uint32_t n = uview.n;
  swa_ctx * ctx = nullptr;
  swa_multi * multi = nullptr;           // d = 1 on several GPUs: SWARM_AMD_DEVICES=0,1,2,... (one rank per entry)
  if (n > 0) {
    if (use_multi) {
      if (swa_multi_create(devices.data(), (int)devices.size(), &multi) != SWA_OK) {
        die(multi != nullptr ? swa_multi_last_error(multi) : "no usable gfx950 GPU (this build has no CPU fallback).");
      }
      if (std::getenv("SWARM_AMD_MULTI_REPORT") != nullptr) {      // (tools/scale_check.sh asserts on this line)
        std::fprintf(stderr, "multi: %d ranks, exchange = %s\n", swa_multi_size(multi), swa_multi_uses_rccl(multi) ? "rccl" : "device-to-device copies");
      }
      swa_db_view view{};                                    // (several GPUs: the db-order host arrays, gathered on the host)
      swa_hostdb_view(db, &view);
      if (swa_multi_db_upload(multi, &view) != SWA_OK) { die(swa_multi_last_error(multi)); }
      stamp("database uploaded (all ranks)");
      ctx = swa_multi_ctx(multi, 0);
    } else {
      if (early_rc != SWA_OK || early_ctx == nullptr) { die("no usable gfx950 GPU (this build has no CPU fallback)."); }
      ctx = early_ctx;
      stamp("context created");
      if (swa_db_upload_unordered(ctx, &uview) != SWA_OK) { die(swa_last_error(ctx)); }
      stamp("database uploaded");
    }
  }

  if (o.differences == 0) {
    // ---- d = 0: identical-sequence search on the GPU, cluster bookkeeping on the host
    std::vector<uint32_t> first_identical(n);
    if (n > 0 && swa_derep(ctx, first_identical.data()) != SWA_OK) { die(swa_last_error(ctx)); }
    swa_d0_result * res = nullptr;
    if (swa_d0_cluster(db, first_identical.data(), &res) != SWA_OK) { die("dereplication failed"); }
    phase(o, "Dereplicating:    ");
    phase(o, "Sorting:          ");
    check_writer(swa_d0_write_swarms(res, db, o.output.c_str(), o.mothur, o.usearch, o.append_abundance, o.differences), "output");
    phase(o, "Writing swarms:   ");
    if (!o.seeds.empty()) { check_writer(swa_d0_write_seeds(res, db, o.seeds.c_str(), o.usearch), "seeds"); phase(o, "Writing seeds:    "); }
    if (!o.uclust.empty()) { check_writer(swa_d0_write_uclust(res, db, o.uclust.c_str(), o.usearch, o.append_abundance), "uclust"); phase(o, "Writing UCLUST:   "); }
    if (!o.structure.empty()) { check_writer(swa_d0_write_structure(res, db, o.structure.c_str(), o.usearch), "internal structure"); phase(o, "Writing structure:"); }
    if (!o.stats.empty()) { check_writer(swa_d0_write_stats(res, db, o.stats.c_str(), o.usearch), "statistics"); phase(o, "Writing stats:    "); }
    uint64_t sum[3];
    swa_d0_result_summary(res, sum);
    std::fprintf(g_log, "\nNumber of swarms:  %" PRIu64 "\nLargest swarm:     %" PRIu64 "\nHeaviest swarm:    %" PRIu64 "\n", sum[0],
                 sum[1], sum[2]);
    swa_d0_result_free(res);
  } else if (o.differences == 1) {
    // ---- seam B1: the network on the GPU
    swa_d1_result * prepared = nullptr;
    int prepare_rc = SWA_OK;
    std::thread preparing;
    std::vector<uint64_t> offsets;              // (the CSR only comes to the host for -j and from several GPUs)
    std::vector<uint32_t> neighbours;
    bool resident = false;
    if (multi != nullptr || !o.network.empty()) { offsets.assign((size_t)n + 1, 0); }
    const char * dup_text = "some fasta entries have identical sequences.\n"
                            "Swarm expects dereplicated fasta files.\n"
                            "Such files can be produced with swarm or vsearch:\n"
                            " swarm -d 0 -w derep.fasta -o /dev/null input.fasta\n"
                            "or\n"
                            " vsearch --derep_fulllength input.fasta --sizein --sizeout --output derep.fasta\n";
    if (n > 0 && multi != nullptr) {
      uint64_t total = 0;
      int dup = 0;
      neighbours.resize(std::max<size_t>(4 * (size_t)n, 1024));
      for (;;) {
        rc = swa_multi_d1_network(multi, o.no_break ? 1 : 0, offsets.data(), neighbours.data(), neighbours.size(), &total, &dup);
        if (rc == SWA_E_DUPLICATES) { die(dup_text); }
        if (rc == SWA_E_CAPACITY) { neighbours.resize(total); continue; }
        if (rc != SWA_OK) { die(swa_multi_last_error(multi)); }
        break;
      }
      neighbours.resize(total);
      phase(o, "Hashing sequences:");
      phase(o, "Building network: ");
    } else if (n > 0) {
      // beside the index and the network (12 ms at 10 M): the result's arrays sized and pinned, so that the member order comes
      // home as one DMA (SWARM_AMD_PIN_RESULTS=0: the staged copy of round 5)
      {
        const char * pin = std::getenv("SWARM_AMD_PIN_RESULTS");
        if (pin == nullptr || pin[0] != '0') {
          preparing = std::thread([&prepared, &prepare_rc, ctx, db]() {
            prepare_rc = swa_d1_result_prepare(ctx, db, &prepared);
            stamp("(helper thread) result arrays pinned");
          });
        }
      }
      int dup = 0;
      rc = swa_d1_index_build(ctx, &dup);
      if (rc != SWA_OK && preparing.joinable()) { preparing.join(); }     // (no exit under a thread that is talking to the runtime)
      if (rc == SWA_E_DUPLICATES) {
        die("some fasta entries have identical sequences.\n"
            "Swarm expects dereplicated fasta files.\n"
            "Such files can be produced with swarm or vsearch:\n"
            " swarm -d 0 -w derep.fasta -o /dev/null input.fasta\n"
            "or\n"
            " vsearch --derep_fulllength input.fasta --sizein --sizeout --output derep.fasta\n");
      }
      if (rc != SWA_OK) { die(swa_last_error(ctx)); }
      phase(o, "Hashing sequences:");
      // the network stays in HBM: the agglomeration runs on it there; it only comes to the host for -j
      uint64_t total = 0;
      rc = swa_d1_network_resident(ctx, o.no_break ? 1 : 0, &total);
      if (rc != SWA_OK && preparing.joinable()) { preparing.join(); }
      if (rc == SWA_E_DUPLICATES) { die(dup_text); }        // (identical sequences met by the prefix pass of the pair kernels)
      if (rc != SWA_OK) { die(swa_last_error(ctx)); }
      resident = true;
      if (!o.network.empty()) {
        neighbours.resize(total);
        if (swa_d1_network_fetch(ctx, offsets.data(), neighbours.data(), neighbours.size()) != SWA_OK) { die(swa_last_error(ctx)); }
      }
      phase(o, "Building network: ");
    }
    if (!o.network.empty()) {
      check_writer(swa_d1_write_network(db, offsets.data(), neighbours.data(), o.network.c_str(), o.usearch, o.append_abundance), "network");
      phase(o, "Dumping network:  ");
    }
    // ---- host: greedy clustering over the neighbour lists
    swa_d1_result * res = nullptr;
    if (preparing.joinable()) { preparing.join(); }
    if (resident && prepared != nullptr && prepare_rc == SWA_OK) {
      res = prepared;
      if (swa_d1_cluster_resident_prepared(ctx, db, res) != SWA_OK) { die(swa_last_error(ctx)); }
    } else if (resident) { if (swa_d1_cluster_resident_lazy(ctx, db, &res) != SWA_OK) { die(swa_last_error(ctx)); } }
    else if (swa_d1_cluster(db, offsets.data(), neighbours.data(), &res) != SWA_OK) { die("clustering failed"); }
    phase(o, "Clustering:       ");
    uint64_t sum[4];
    swa_d1_result_summary(res, sum);

    if (o.fastidious) {
      std::fprintf(g_log, "\nResults before fastidious processing:\n");
      std::fprintf(g_log, "Number of swarms:  %" PRIu64 "\n", sum[0]);
      std::fprintf(g_log, "Largest swarm:     %" PRIu64 "\n\n", sum[1]);
      swa_vec<uint8_t> is_light(n);                         // (every entry is written by swa_d1_light_flags: each amplicon is in one swarm)
      uint64_t st[5];
      if (swa_d1_light_flags(res, o.boundary, is_light.data(), st) != SWA_OK) { die(swa_d1_result_error(res)); }
      phase(o, "Counting amplicons in heavy and light swarms");
      std::fprintf(g_log, "Heavy swarms: %" PRIu64 ", with %" PRIu64 " amplicons\n", st[3], st[4]);
      std::fprintf(g_log, "Light swarms: %" PRIu64 ", with %" PRIu64 " amplicons\n", st[0], st[1]);
      std::fprintf(g_log, "Total length of amplicons in light swarms: %" PRIu64 "\n", st[2]);
      if (st[0] == 0 || st[3] == 0) {
        std::fprintf(g_log, "Only light or heavy swarms found - no need for further analysis.\n");
      } else {
        // sizing as src/algod1.cc:1337-1396 (the --ceiling shrink works on the GPU's memory budget:
        // the filter lives in HBM, so the host RSS terms of the reference do not apply)
        uint64_t bits = (uint64_t)o.bloom_bits;
        if (o.ceiling != 0) {
          const uint64_t budget = (uint64_t)o.ceiling * 1024ull * 1024ull;
          const uint64_t new_bits = 8ull * budget / (7ull * st[2]);
          if (new_bits < bits) {
            if (new_bits < 2) { die("Insufficient memory remaining for Bloom filter."); }
            std::fprintf(g_log, "Reducing memory used for Bloom filter due to --ceiling option.\n");
            bits = new_bits;
          }
        }
        swa_vec<uint32_t> graft(n);                         // (the download writes every entry; its pages are faulted in by all threads first:
        {                                                   //  one thread's fill of 40 MB was 10 ms of this phase)
          char * gp = reinterpret_cast<char *>(graft.data());
          const int64_t pages = (int64_t)(((size_t)n * sizeof(uint32_t) + 4095) / 4096);
#pragma omp parallel for schedule(static)
          for (int64_t k = 0; k < pages; ++k) { gp[k * 4096] = 0; }
        }
        uint64_t counters[8] = {};
        // the log line needs m and k before the passes run: same arithmetic as the library
        unsigned k = (unsigned)(0.4 * (double)bits);
        if (k < 1) { k = 1; }
        uint64_t m = st[2] * 7ull * bits;
        if (m < 64) { m = 64; }
        std::fprintf(g_log, "Bloom filter: bits=%" PRIu64 ", m=%" PRIu64 ", k=%u, size=%.1fMB\n", bits, m, k,
                     (double)m / (8.0 * 1024.0 * 1024.0));
        if (multi != nullptr) {
          rc = swa_multi_d1_fastidious(multi, is_light.data(), st[2], (uint32_t)bits, graft.data(), counters);
          if (rc != SWA_OK) { die(swa_multi_last_error(multi)); }
        } else {
          rc = swa_d1_fastidious(ctx, is_light.data(), st[2], (uint32_t)bits, graft.data(), counters);
          if (rc != SWA_OK) { die(swa_last_error(ctx)); }
        }
        phase(o, "Adding light swarm amplicons to Bloom filter");
        std::fprintf(g_log, "Generated %" PRIu64 " variants from light swarms\n", counters[0]);
        phase(o, "Checking heavy swarm amplicons against Bloom filter");
        std::fprintf(g_log, "Heavy variants: %" PRIu64 "\n", counters[1]);
        std::fprintf(g_log, "Got %" PRIu64 " graft candidates\n", counters[2]);
        const uint32_t grafts = swa_d1_graft(res, graft.data());
        phase(o, "Grafting light swarms on heavy swarms");
        std::fprintf(g_log, "Made %u grafts\n\n", grafts);
      }
    }

    check_writer(swa_d1_write_swarms(res, db, o.output.c_str(), o.mothur, o.usearch, o.append_abundance, o.differences), "output", res);
    phase(o, "Writing swarms:   ");
    if (!o.seeds.empty()) { check_writer(swa_d1_write_seeds(res, db, o.seeds.c_str(), o.usearch), "seeds", res); phase(o, "Writing seeds:    "); }
    if (!o.structure.empty()) { check_writer(swa_d1_write_structure(res, db, o.structure.c_str(), o.usearch), "internal structure", res); phase(o, "Writing structure:"); }
    if (!o.uclust.empty()) {
      check_writer(swa_d1_write_uclust(res, db, o.uclust.c_str(), o.usearch, o.append_abundance, (uint64_t)o.pen_mismatch,
                                       (uint64_t)o.pen_gapopen, (uint64_t)o.pen_gapextend), "uclust", res);
      phase(o, "Writing UCLUST:   ");
    }
    if (!o.stats.empty()) { check_writer(swa_d1_write_stats(res, db, o.stats.c_str(), o.usearch), "statistics", res); phase(o, "Writing stats:    "); }
    swa_d1_result_summary(res, sum);
    std::fprintf(g_log, "\nNumber of swarms:  %" PRIu64 "\nLargest swarm:     %" PRIu64 "\nMax generations:   %" PRIu64 "\n", sum[0],
                 sum[1], sum[2]);
    // (the result's arrays are left to the kernel with the rest, unless an orderly teardown was asked for: unmapping them
    // here is time the caller waits, lease r5j)
    if (std::getenv("SWARM_AMD_FULL_TEARDOWN") != nullptr) { swa_d1_result_free(res); }
  } else {
    // ---- d >= 2: host greedy loop, every q-gram / alignment step on the GPU
    swa_dn_result * res = nullptr;
    if (n > 0) {
      rc = multi != nullptr ? swa_dn_cluster_multi(multi, db, o.differences, o.no_break ? 1 : 0, (uint64_t)o.pen_mismatch,
                                                   (uint64_t)o.pen_gapopen, (uint64_t)o.pen_gapextend, &res)
                            : swa_dn_cluster(ctx, db, o.differences, o.no_break ? 1 : 0, (uint64_t)o.pen_mismatch, (uint64_t)o.pen_gapopen,
                                             (uint64_t)o.pen_gapextend, &res);
      if (rc != SWA_OK) { die(res != nullptr ? swa_dn_result_error(res) : "clustering failed"); }
    } else {
      rc = SWA_OK;
    }
    phase(o, "Find qgram vects: ");
    phase(o, "Clustering:       ");
    uint64_t sum[3] = {0, 0, 0};
    if (res != nullptr) {
      check_writer(swa_dn_write_swarms(res, db, o.output.c_str(), o.mothur, o.usearch, o.append_abundance), "output");
      if (!o.structure.empty()) { check_writer(swa_dn_write_structure(res, db, o.structure.c_str(), o.usearch), "internal structure"); }
      if (!o.uclust.empty()) { check_writer(swa_dn_write_uclust(res, db, o.uclust.c_str(), o.usearch, o.append_abundance), "uclust"); }
      if (!o.stats.empty()) { check_writer(swa_dn_write_stats(res, db, o.stats.c_str(), o.usearch), "statistics"); }
      if (!o.seeds.empty()) { check_writer(swa_dn_write_seeds(res, db, o.seeds.c_str(), o.usearch), "seeds"); phase(o, "Writing seeds:    "); }
      swa_dn_result_summary(res, sum);
      swa_dn_result_free(res);
    } else {
      // empty input: the reference creates empty output files
      for (const std::string * p : {&o.output, &o.structure, &o.uclust, &o.stats, &o.seeds}) {
        if (!p->empty() && *p != "-") { FILE * fp = std::fopen(p->c_str(), "w"); if (fp != nullptr) { std::fclose(fp); } }
      }
    }
    std::fprintf(g_log, "\nNumber of swarms:  %" PRIu64 "\nLargest swarm:     %" PRIu64 "\nMax generations:   %" PRIu64 "\n", sum[0],
                 sum[1], sum[2]);
  }
  stamp("results written");
  stamp_usage("results written");
  // Every output file is closed at this point.  What the process still holds costs at exit by what the kernel has to take
  // apart on ONE thread — ~75 ms per GB of host pages on the bench host, nothing measurable for device memory — or next
  // to nothing when a kernel worker that still holds the address space does it on its own time.  Freeing
  // anything here first only adds to the caller's wait (measured: in parallel, in slices, on helper threads — leases
  // c, i), so everything is left to the kernel (SWARM_AMD_FULL_TEARDOWN=1 keeps the orderly path, e.g. under a leak checker).
  if (std::getenv("SWARM_AMD_FULL_TEARDOWN") == nullptr) {
    if (g_log != stderr && g_log != stdout) { std::fclose(g_log); }
    std::fflush(nullptr);
    std::_Exit(EXIT_SUCCESS);
  }
  if (multi != nullptr) { swa_multi_destroy(multi); }
  else if (ctx != nullptr) { swa_ctx_destroy(ctx); }
  stamp("device released");
  swa_hostdb_free(db);
  stamp("host database released");
  if (g_log != stderr && g_log != stdout) { std::fclose(g_log); }
  return EXIT_SUCCESS;
}
