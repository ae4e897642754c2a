// host vertex sort emulation (ExactDelaunay::sort_ties) on lists like the benchmark's: 7.3 k points, 65 duplicate pairs
//   g++ -O3 -std=c++17 -I opencl-structure-from-motion_amd/csrc -I include tools/micro/sort_bench.cpp opencl-structure-from-motion_amd/csrc/build/vsm_host.o -lpthread -o /tmp/sort_bench
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <random>
#include <vector>
#include "vsm_host.h"
int main(int argc, char **argv) {
  const int n = argc > 1 ? atoi(argv[1]) : 7350, dup = argc > 2 ? atoi(argv[2]) : 65, lists = 64;
  std::mt19937 rng(7);
  std::vector<std::vector<uint64_t>> keys(lists);
  for (auto &k : keys) {
    for (int i = 0; i < n - dup; i++) {
      const uint64_t x = rng() % 1242, y = rng() % 375;
      k.push_back((x << 34) | (y << 20) | (uint64_t)i);
    }
    for (int i = 0; i < dup; i++) {  // a second match at the pixel of an earlier one
      const uint64_t src = k[rng() % (n - dup)];
      k.push_back((src & ~0xfffffull) | (uint64_t)(n - dup + i));
    }
    for (int i = n - 1; i > 0; i--) {  // list order is not pixel order
      const int j = rng() % (i + 1);
      std::swap(k[i], k[j]);
    }
    for (int i = 0; i < n; i++) k[i] = (k[i] & ~0xfffffull) | (uint64_t)i;
  }
  ExactDelaunay d;
  std::vector<int32_t> out(2 * n + 2);
  long sum = 0;
  for (int rep = 0; rep < 3; rep++) {
    const auto t0 = std::chrono::steady_clock::now();
    for (int r = 0; r < 20; r++)
      for (auto &k : keys) sum += d.sort_ties(k.data(), n, out.data(), n);
    const double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
    printf("%.1f us per list of %d (%ld)\n", us / (20 * lists), n, sum);
  }
  return 0;
}
