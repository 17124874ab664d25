// pool.h — the host side's worker threads, started once.
// The reader (fasta_db.cpp) runs a dozen short parallel phases; a std::thread per phase and worker costs a stack mapping
// each (the process's memory-map lock, which the HIP runtime starting up beside the reader wants all the time) and showed
// as stalls of 50-80 ms in single phases at 10 M amplicons (profiles/r05/NOTES.md).  Workers here are created on first
// use, wait on a condition variable between phases and claim task numbers from a shared counter, so a phase may have
// more tasks than there are workers.  One phase at a time (a second caller waits); the calling thread works too.
#pragma once

#include <sched.h>

#include <atomic>
#include <condition_variable>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

// CPUs this process may actually use: the hardware's, cut down to the scheduler affinity and to the cgroup's CPU quota
// (cpu.max "quota period": quota / period CPUs' worth of time per period).  The quota matters more than it looks: the
// bench host shows 256 CPUs to a container that may use 16 — 64 busy threads spend a period's quota in a quarter of the
// period and are then ALL stopped for the rest of it (the GPU helper thread and the main thread included): phases of the
// reader took 19 or 96 ms by where in the period they began, and HIP's start-up 0.12 or 0.3 s (lease r5e).
inline unsigned swa_host_cpus() {
  static const unsigned cpus = [] {
    unsigned n = std::thread::hardware_concurrency();
    if (n < 1) { n = 1; }
    cpu_set_t set;
    if (sched_getaffinity(0, sizeof(set), &set) == 0) { const int c = CPU_COUNT(&set); if (c > 0 && (unsigned)c < n) { n = (unsigned)c; } }
    auto quota_of = [](const char * path, const char * period_path) -> double {
      double quota = -1.0, period = 100000.0;
      if (FILE * f = std::fopen(path, "r")) {
        char first[64] = {0};
        if (period_path == nullptr) {                        // cgroup v2: "max 100000" or "1600000 100000"
          if (std::fscanf(f, "%63s %lf", first, &period) >= 1 && std::strcmp(first, "max") != 0) { quota = std::atof(first); }
        } else if (std::fscanf(f, "%lf", &quota) != 1) { quota = -1.0; }
        std::fclose(f);
      }
      if (period_path != nullptr) { if (FILE * f = std::fopen(period_path, "r")) { if (std::fscanf(f, "%lf", &period) != 1) { period = 100000.0; } std::fclose(f); } }
      return quota > 0.0 && period > 0.0 ? quota / period : -1.0;
    };
    double q = quota_of("/sys/fs/cgroup/cpu.max", nullptr);
    // a quota set on a nested, non-namespaced cgroup: the process's own path ("0::/a/b" in /proc/self/cgroup) and its parents
    if (FILE * f = std::fopen("/proc/self/cgroup", "r")) {
      char line[512];
      while (std::fgets(line, sizeof(line), f) != nullptr) {
        if (std::strncmp(line, "0::", 3) != 0) { continue; }
        std::string path(line + 3);
        while (!path.empty() && (path.back() == '\n' || path.back() == '/')) { path.pop_back(); }
        while (!path.empty()) {
          const double nested = quota_of(("/sys/fs/cgroup" + path + "/cpu.max").c_str(), nullptr);
          if (nested > 0.0 && (q <= 0.0 || nested < q)) { q = nested; }
          const size_t cut = path.rfind('/');
          if (cut == std::string::npos) { break; }
          path.resize(cut);
        }
      }
      std::fclose(f);
    }
    if (q <= 0.0) { q = quota_of("/sys/fs/cgroup/cpu/cpu.cfs_quota_us", "/sys/fs/cgroup/cpu/cpu.cfs_period_us"); }
    if (q > 0.0) { const unsigned c = (unsigned)(q + 0.5); if (c >= 1 && c < n) { n = c; } }
    return n;
  }();
  return cpus;
}

class swa_pool {
 public:
  static swa_pool & get() {
    static swa_pool pool;
    return pool;
  }
  unsigned size() const { return (unsigned)workers_.size() + 1u; }     // workers + the caller

  // fn(t) for every t in [0, tasks), on up to size() threads; returns when all are done
  template <class F>
  void run(unsigned tasks, F && fn) {
    if (tasks == 0) { return; }
    // a task of a running phase that starts a phase of its own (a resize inside a pooled loop, a sort reached from one):
    // the phase mutex is not recursive and the workers are busy with the outer phase — run it here
    if (tasks == 1 || workers_.empty() || in_phase()) { for (unsigned t = 0; t < tasks; ++t) { fn(t); } return; }
    std::lock_guard<std::mutex> one_phase(phase_);
    struct Mark { Mark() { in_phase() = true; } ~Mark() { in_phase() = false; } } mark;
    std::function<void(unsigned)> job = std::ref(fn);
    {
      std::lock_guard<std::mutex> lk(m_);
      job_ = &job;
      tasks_ = tasks;
      next_.store(0, std::memory_order_relaxed);
      pending_ = tasks;
      ++generation_;
    }
    wake_.notify_all();
    work(job, tasks);
    std::unique_lock<std::mutex> lk(m_);
    done_.wait(lk, [&] { return pending_ == 0 && busy_ == 0; });
    job_ = nullptr;
  }

 private:
  swa_pool() {
    const char * env = std::getenv("SWARM_AMD_HOST_THREADS");
    unsigned n = env != nullptr ? (unsigned)std::atoi(env) : swa_host_cpus();
    if (n < 1) { n = 1; }
    if (n > 64) { n = 64; }
    for (unsigned i = 1; i < n; ++i) { workers_.emplace_back([this] { loop(); }); }
  }
  ~swa_pool() {
    { std::lock_guard<std::mutex> lk(m_); stop_ = true; ++generation_; }
    wake_.notify_all();
    for (auto & w : workers_) { w.join(); }
  }
  static bool & in_phase() { static thread_local bool inside = false; return inside; }
  void work(const std::function<void(unsigned)> & job, unsigned tasks) {
    unsigned finished = 0;
    for (;;) {
      const unsigned t = next_.fetch_add(1, std::memory_order_relaxed);
      if (t >= tasks) { break; }
      job(t);
      ++finished;
    }
    if (finished != 0) {
      std::lock_guard<std::mutex> lk(m_);
      pending_ -= finished;
      if (pending_ == 0) { done_.notify_all(); }
    }
  }
  void loop() {
    uint64_t seen = 0;
    for (;;) {
      const std::function<void(unsigned)> * job = nullptr;
      unsigned tasks = 0;
      {
        std::unique_lock<std::mutex> lk(m_);
        wake_.wait(lk, [&] { return generation_ != seen; });
        seen = generation_;
        if (stop_) { return; }
        job = job_;
        tasks = tasks_;
        if (job == nullptr) { continue; }
        ++busy_;                                // (the phase's owner waits for every worker that picked the job up)
      }
      in_phase() = true;
      work(*job, tasks);
      in_phase() = false;
      {
        std::lock_guard<std::mutex> lk(m_);
        --busy_;
        if (pending_ == 0 && busy_ == 0) { done_.notify_all(); }
      }
    }
  }

  std::vector<std::thread> workers_;
  std::mutex phase_, m_;
  std::condition_variable wake_, done_;
  const std::function<void(unsigned)> * job_ = nullptr;
  unsigned tasks_ = 0, pending_ = 0, busy_ = 0;
  std::atomic<unsigned> next_{0};
  uint64_t generation_ = 0;
  bool stop_ = false;
};
