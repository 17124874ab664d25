// sort_bench.cpp — the reader's sample sort (fasta_db.cpp: parallel_sample_sort, restated) on 10 M synthetic 16-byte records against
// libstdc++'s parallel multiway merge sort on 32-byte ones, by thread count: usage sort_bench THREADS
// parallel sample sort (header under test)

#include <algorithm>
#include <atomic>
#include <cstdint>
#include <thread>
#include <vector>
template <typename F>
static void ps_run(unsigned threads, F && fn) {
  if (threads <= 1) { fn(0u); return; }
  std::vector<std::thread> pool; pool.reserve(threads);
  for (unsigned t = 0; t < threads; ++t) { pool.emplace_back([&fn, t] { fn(t); }); }
  for (auto & th : pool) { th.join(); }
}
// sorts a[0, n) with `less` (a strict weak order); tmp[0, n) is scratch; the result is in a
template <class Rec, class Less>
void parallel_sample_sort(Rec * a, Rec * tmp, uint64_t n, unsigned threads, Less less) {
  if (threads <= 1 || n < 100000) { std::sort(a, a + n, less); return; }
  const unsigned buckets = std::min<unsigned>(threads * 8u, 1024u);
  const uint64_t nsample = (uint64_t)buckets * 64u;
  std::vector<Rec> sample(nsample);
  for (uint64_t i = 0; i < nsample; ++i) { sample[i] = a[(n - 1) * i / (nsample - 1)]; }
  std::sort(sample.begin(), sample.end(), less);
  std::vector<Rec> split(buckets - 1);
  for (unsigned b = 1; b < buckets; ++b) { split[b - 1] = sample[(uint64_t)b * 64u]; }
  std::vector<uint64_t> count((size_t)threads * buckets, 0);
  std::vector<uint16_t> where(n);
  ps_run(threads, [&](unsigned t) {
    uint64_t * c = &count[(size_t)t * buckets];
    for (uint64_t i = n * t / threads; i < n * (t + 1) / threads; ++i) {
      // bucket = number of splitters not greater than the record
      unsigned lo = 0, hi = buckets - 1;
      while (lo < hi) { const unsigned mid = (lo + hi) / 2; if (less(a[i], split[mid])) { hi = mid; } else { lo = mid + 1; } }
      where[i] = (uint16_t)lo; ++c[lo];
    }
  });
  std::vector<uint64_t> start(buckets + 1, 0);
  { uint64_t at = 0;
    for (unsigned b = 0; b < buckets; ++b) { start[b] = at; for (unsigned t = 0; t < threads; ++t) { const uint64_t c = count[(size_t)t * buckets + b]; count[(size_t)t * buckets + b] = at; at += c; } }
    start[buckets] = at; }
  ps_run(threads, [&](unsigned t) {
    uint64_t * c = &count[(size_t)t * buckets];
    for (uint64_t i = n * t / threads; i < n * (t + 1) / threads; ++i) { tmp[c[where[i]]++] = a[i]; }
  });
  std::atomic<unsigned> next{0};
  ps_run(threads, [&](unsigned) {
    for (;;) { const unsigned b = next.fetch_add(1); if (b >= buckets) { break; }
      std::sort(tmp + start[b], tmp + start[b + 1], less);
      std::copy(tmp + start[b], tmp + start[b + 1], a + start[b]); }
  });
}
#include <parallel/algorithm>
#include <omp.h>
#include <chrono>
#include <cstdio>
#include <random>
struct R16 { uint64_t key8; uint32_t ab, entry; };
struct R32 { uint64_t abundance, key8; uint32_t entry, words, hdr; };
int main(int argc, char**argv){
  uint64_t n = 10000000; unsigned T = argc>1?atoi(argv[1]):8;
  std::mt19937_64 g(1);
  std::vector<R16> a(n), tmp(n); std::vector<R32> b(n);
  for (uint64_t i=0;i<n;i++){ double u=(g()%1000000+1)/1e6; uint32_t ab = (uint32_t)(1.0/ (u*u)) ; if (ab<1) ab=1; char h[16]; snprintf(h,16,"s%llu_%u",(unsigned long long)i,ab); uint64_t k=0; for(int j=0;j<8;j++) k=(k<<8)|(unsigned char)(h[j]); a[i]={k,ab,(uint32_t)i}; b[i]={ab,k,(uint32_t)i,5,12}; }
  auto l16=[](const R16&x,const R16&y){ if(x.ab!=y.ab) return x.ab>y.ab; if (x.key8!=y.key8) return x.key8<y.key8; return x.entry<y.entry; };
  auto l32=[](const R32&x,const R32&y){ if(x.abundance!=y.abundance) return x.abundance>y.abundance; if (x.key8!=y.key8) return x.key8<y.key8; return x.entry<y.entry; };
  auto t0=std::chrono::steady_clock::now();
  parallel_sample_sort(a.data(), tmp.data(), n, T, l16);
  auto t1=std::chrono::steady_clock::now();
  omp_set_num_threads(T);
  __gnu_parallel::sort(b.begin(), b.end(), l32);
  auto t2=std::chrono::steady_clock::now();
  bool ok=true; for(uint64_t i=0;i<n;i++) if(a[i].entry!=b[i].entry){ok=false;break;}
  printf("sample16 %.1f ms   gnu32 %.1f ms  same=%d\n", std::chrono::duration<double,std::milli>(t1-t0).count(), std::chrono::duration<double,std::milli>(t2-t1).count(), ok);
}
