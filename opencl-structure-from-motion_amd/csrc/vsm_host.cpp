// Host-side stages of the matcher path (see vsm_host.h).  Plain C++17, no HIP.
#include "vsm_host.h"

#include <ctype.h>
#include <pthread.h>
#include <sched.h>
#include <stdio.h>

#include <math.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <map>
#include <string>
#include <chrono>
#if defined(__x86_64__)
#include <immintrin.h>
#endif

// VSM_DEBUG_TIMING: phase times on stderr (read once)
static bool vsm_host_debug_timing() {
  static const bool on = getenv("VSM_DEBUG_TIMING") != nullptr;
  return on;
}

// =======================================================================================
// VsmPool
// =======================================================================================
static inline void cpu_relax() {
#if defined(__x86_64__) || defined(__i386__)
  __builtin_ia32_pause();
#endif
}

// ---- host threads near the GPU (vsm_host.h) ----
// One record per device (PCI bus id): a second handle on another GPU gets its own node.
// ... and spread over the socket's L3 domains (VSM_HOST_AFFINITY=1: the node only, no spreading).  What the pool does with the
// lists the GPU's DMA wrote - 25 ms of vertex sorts over keys it has to fetch, 64 MB of gap closing - is bound by memory, and a
// core complex has its own link to memory: fourteen threads on two of the socket's eight complexes run the call in 4.5 ms, on
// one 4.9-5.3, spread over all eight 4.15 (tools/step_spread.sh, the rank confined to 16 / 8 cores / the whole socket) - and
// left to the scheduler a process now and then stays packed (the "slower process", DESIGN_HISTORY.md 6c).  The library's threads are
// dealt over the complexes IN PROPORTION TO THE CPUS each complex has inside the allowed set (a cpuset that cuts a complex
// leaves it its share, not half the pool); a thread may use all CPUs of its complex: the scheduler still chooses the core, and
// ranks that share a socket share it evenly.
struct VsmAffinity {
  cpu_set_t near;                 // the device's node within what the process may use
  bool narrow = false;            // ... and that is fewer CPUs than the process may use
  std::vector<cpu_set_t> order;   // pin sequence: thread i of this device -> order[i % size]
  cpu_set_t used;                 // union of what the threads are confined to
  bool any = false;
  std::atomic<unsigned> next{0};
};
static std::mutex g_near_mu;
static std::map<std::string, VsmAffinity *> g_aff;
static std::atomic<VsmAffinity *> g_aff_last{nullptr};
static thread_local VsmAffinity *t_aff = nullptr;

static void parse_cpulist(char *line, cpu_set_t *out, const cpu_set_t *within) {
  for (char *tok = strtok(line, ",\n"); tok; tok = strtok(nullptr, ",\n")) {
    int a = -1, b = -1;
    if (sscanf(tok, "%d-%d", &a, &b) != 2 && sscanf(tok, "%d", &a) == 1) b = a;
    for (int c = a; a >= 0 && c <= b && c < CPU_SETSIZE; c++)
      if (!within || CPU_ISSET(c, within)) CPU_SET(c, out);
  }
}

void vsm_affinity_from_device(const char *pci_bus_id) {
  std::lock_guard<std::mutex> lk(g_near_mu);
  t_aff = nullptr;
  const char *e = getenv("VSM_HOST_AFFINITY");
  char bdf[64] = {0};
  if ((e && atoi(e) == 0) || !pci_bus_id || strlen(pci_bus_id) >= sizeof(bdf)) return;
  strcpy(bdf, pci_bus_id);
  for (char *c = bdf; *c; c++) *c = (char)tolower(*c);
  auto it = g_aff.find(bdf);
  if (it != g_aff.end()) {
    t_aff = it->second;
    g_aff_last.store(t_aff);
    return;
  }
  VsmAffinity *A = new VsmAffinity();
  CPU_ZERO(&A->near);
  CPU_ZERO(&A->used);
  g_aff[bdf] = A;  // (looked up once per device, whatever comes of it)
  char path[160];
  snprintf(path, sizeof(path), "/sys/bus/pci/devices/%s/local_cpulist", bdf);
  FILE *f = fopen(path, "r");
  if (!f) return;
  char line[1024] = {0};
  cpu_set_t near, allowed, both;
  CPU_ZERO(&near);
  if (fgets(line, sizeof(line), f)) parse_cpulist(line, &near, nullptr);
  fclose(f);
  if (sched_getaffinity(0, sizeof(allowed), &allowed) != 0) return;
  CPU_AND(&both, &near, &allowed);
  if (CPU_COUNT(&both) < 2) return;  // (a process whose allowed CPUs miss the node stays as it is)
  A->near = both;
  A->narrow = CPU_COUNT(&both) < CPU_COUNT(&allowed);
  std::vector<cpu_set_t> doms;
  if (!(e && atoi(e) == 1)) {  // the L3 domains of the CPUs the threads may use near the GPU
    cpu_set_t left = both;
    for (int c = 0; c < CPU_SETSIZE; c++) {
      if (!CPU_ISSET(c, &left)) continue;
      cpu_set_t dom;
      CPU_ZERO(&dom);
      char p3[160], l3[1024] = {0};
      snprintf(p3, sizeof(p3), "/sys/devices/system/cpu/cpu%d/cache/index3/shared_cpu_list", c);
      FILE *f3 = fopen(p3, "r");
      if (f3 && fgets(l3, sizeof(l3), f3)) parse_cpulist(l3, &dom, &both);
      if (f3) fclose(f3);
      if (CPU_COUNT(&dom) == 0) CPU_SET(c, &dom);
      for (int k = 0; k < CPU_SETSIZE; k++)
        if (CPU_ISSET(k, &dom)) CPU_CLR(k, &left);
      doms.push_back(dom);
    }
  }
  if (doms.size() >= 2) {
    // round r takes every domain that has more than r CPUs: interleaved, and in proportion to the domains' sizes
    int most = 0;
    for (const cpu_set_t &d : doms) most = std::max(most, CPU_COUNT(&d));
    for (int r = 0; r < most; r++)
      for (const cpu_set_t &d : doms)
        if (CPU_COUNT(&d) > r) A->order.push_back(d);
    A->used = both;
    A->any = true;
  } else if (A->narrow) {
    A->order.push_back(both);
    A->used = both;
    A->any = true;
  }
  if (A->any) {
    t_aff = A;
    g_aff_last.store(A);
  }
}
// One triangulation is split over these threads and its upper merges read what the others have just built: in one L3
// domain that is a shared-cache hit, across domains a transfer between core complexes.  VSM_FJ_DOMAIN: -1 deals the workers
// over the domains like the pool's threads (the form up to round 5), k >= 0 (default 0) puts all of them into domain k.
// Default: the device's ordinal (vsm_forkjoin_domain_hint, from vsm_create) - the ranks of a node's four GPUs per socket, a
// process each, then take a domain each instead of all sitting on the socket's first.
static std::atomic<int> g_fj_hint{0};
void vsm_forkjoin_domain_hint(int device) { g_fj_hint.store(device < 0 ? 0 : device, std::memory_order_relaxed); }
static int fj_domain() {
  static const int env = [] {
    const char *e = getenv("VSM_FJ_DOMAIN");
    return e ? atoi(e) : INT32_MIN;
  }();
  return env != INT32_MIN ? env : g_fj_hint.load(std::memory_order_relaxed);
}

void *vsm_affinity_current() { return t_aff; }
// the CPUs of L3 domain `domain` of the record (where vsm_pin_this_thread_together puts its callers)
int vsm_affinity_domain_cpus(void *aff, int domain, int *out, int cap) {
  VsmAffinity *A = aff ? (VsmAffinity *)aff : g_aff_last.load();
  if (!A || !A->any || A->order.empty()) return 0;
  const cpu_set_t &dom = A->order[(size_t)domain % A->order.size()];
  int n = 0;
  for (int c = 0; c < CPU_SETSIZE; c++)
    if (CPU_ISSET(c, &dom)) {
      if (n < cap) out[n] = c;
      n++;
    }
  return n;
}
int vsm_forkjoin_domain() { return fj_domain(); }
static thread_local int t_domain = -1;
void vsm_pin_this_thread(void *aff) {
  VsmAffinity *A = aff ? (VsmAffinity *)aff : t_aff;
  if (!A || !A->any || A->order.empty()) return;
  const unsigned at = A->next.fetch_add(1, std::memory_order_relaxed) % A->order.size();
  const cpu_set_t &dom = A->order[at];
  (void)pthread_setaffinity_np(pthread_self(), sizeof(dom), &dom);
  t_domain = (int)at;
}
int vsm_thread_domain() { return t_domain; }
// every thread that asks lands in ONE L3 domain of the device's node (the fork-join workers: they take turns on one mesh)
void vsm_pin_this_thread_together(void *aff, int domain) {
  VsmAffinity *A = aff ? (VsmAffinity *)aff : t_aff;
  if (!A || !A->any || A->order.empty()) return;
  const cpu_set_t &dom = A->order[(size_t)domain % A->order.size()];
  (void)pthread_setaffinity_np(pthread_self(), sizeof(dom), &dom);
  t_domain = (int)((size_t)domain % A->order.size());
}
// The physical cores of an L3 domain (hardware threads that share a core, within the domain's allowed CPUs).  The fork-join
// threads of one triangulation spin between its phases: two of them on the two hardware threads of ONE core halve each
// other, and the caller's serial stretches (flows, bucketing, egomotion) beside a spinning sibling run 10-15 % slower - left
// to the scheduler a process came up in either mode (live VO 0.62 or 0.70 ms per frame, tools/vo_timing.py).  A core each:
// worker i on core i of the domain, core 0 for the caller (vsm_forkjoin_cpus names it).
static std::vector<cpu_set_t> domain_cores(const cpu_set_t &dom) {
  std::vector<cpu_set_t> cores;
  cpu_set_t left = dom;
  for (int c = 0; c < CPU_SETSIZE; c++) {
    if (!CPU_ISSET(c, &left)) continue;
    cpu_set_t core;
    CPU_ZERO(&core);
    char p[160], l[512] = {0};
    snprintf(p, sizeof(p), "/sys/devices/system/cpu/cpu%d/topology/thread_siblings_list", c);
    FILE *f = fopen(p, "r");
    if (f && fgets(l, sizeof(l), f)) parse_cpulist(l, &core, &dom);
    if (f) fclose(f);
    if (!CPU_ISSET(c, &core)) CPU_SET(c, &core);
    for (int k = 0; k < CPU_SETSIZE; k++)
      if (CPU_ISSET(k, &core)) CPU_CLR(k, &left);
    cores.push_back(core);
  }
  return cores;
}
bool vsm_forkjoin_per_core() {
  static const bool on = [] {
    const char *e = getenv("VSM_FJ_CORES");
    return !(e && atoi(e) == 0);
  }();
  return on && fj_domain() >= 0;
}
int vsm_affinity_core_cpus(void *aff, int domain, int core, int *out, int cap) {
  VsmAffinity *A = aff ? (VsmAffinity *)aff : g_aff_last.load();
  if (!A || !A->any || A->order.empty()) return 0;
  const std::vector<cpu_set_t> cores = domain_cores(A->order[(size_t)domain % A->order.size()]);
  if (cores.empty()) return 0;
  const cpu_set_t &cs = cores[(size_t)core % cores.size()];
  int n = 0;
  for (int c = 0; c < CPU_SETSIZE; c++)
    if (CPU_ISSET(c, &cs)) {
      if (n < cap) out[n] = c;
      n++;
    }
  return n;
}
void vsm_pin_this_thread_core(void *aff, int domain, int core) {
  VsmAffinity *A = aff ? (VsmAffinity *)aff : t_aff;
  if (!A || !A->any || A->order.empty()) return;
  const std::vector<cpu_set_t> cores = domain_cores(A->order[(size_t)domain % A->order.size()]);
  if (cores.empty()) return;
  const cpu_set_t &cs = cores[(size_t)core % cores.size()];
  (void)pthread_setaffinity_np(pthread_self(), sizeof(cs), &cs);
  t_domain = (int)((size_t)domain % A->order.size());
}
int vsm_affinity_cpus(int *out, int cap) {  // the CPUs the threads of the device looked up last are confined to (their union)
  VsmAffinity *A = g_aff_last.load();
  if (!A || !A->any) return 0;
  int n = 0;
  for (int c = 0; c < CPU_SETSIZE; c++)
    if (CPU_ISSET(c, &A->used)) {
      if (n < cap) out[n] = c;
      n++;
    }
  return n;
}

VsmPool::VsmPool(int threads) : nthreads_(threads < 1 ? 1 : threads), aff_(vsm_affinity_current()) {
  spin_us_ = 100;  // millisecond-sized tasks: a wake-up is cheap next to them, spinning burns quota
  if (const char *e = getenv("VSM_POOL_SPIN_US")) spin_us_ = std::max(0, atoi(e));  // (measurements)
  for (int i = 1; i < nthreads_; i++) threads_.emplace_back([this] { worker(); });
}

VsmPool::~VsmPool() {
  {
    std::lock_guard<std::mutex> lk(mu_);
    stop_ = true;
    posted_.fetch_add(1);
  }
  cv_.notify_all();
  for (auto &t : threads_) t.join();
}

// claims and runs one task of the oldest batch that still has unclaimed tasks
bool VsmPool::work_one() {
  Ticket b;
  int idx = -1;
  {
    std::lock_guard<std::mutex> lk(mu_);
    while (!queue_.empty() && queue_.front()->next.load(std::memory_order_relaxed) >= queue_.front()->n) queue_.pop_front();
    if (queue_.empty()) return false;
    b = queue_.front();
    idx = b->next.fetch_add(1, std::memory_order_relaxed);
  }
  b->fn(idx);
  b->done.fetch_add(1, std::memory_order_release);
  return true;
}

void VsmPool::worker() {
  vsm_pin_this_thread(aff_);
  for (;;) {
    if (stop_) return;
    // (the post counter is read BEFORE the queue is looked at: a batch posted in between changes it, so the wait below
    // returns at once - read afterwards, that batch would be slept through until the next post, and the look-ahead path's
    // fire-and-forget batches have no waiting caller to run them instead)
    const uint64_t seen = posted_.load(std::memory_order_acquire);
    if (work_one()) continue;
    // idle: spin on the post counter, then block
    auto t0 = std::chrono::steady_clock::now();
    while (posted_.load(std::memory_order_acquire) == seen) {
      cpu_relax();
      if (std::chrono::duration_cast<std::chrono::microseconds>(std::chrono::steady_clock::now() - t0).count() > spin_us_) {
        std::unique_lock<std::mutex> lk(mu_);
        sleepers_.fetch_add(1);
        cv_.wait(lk, [&] { return posted_.load(std::memory_order_acquire) != seen || stop_; });
        sleepers_.fetch_sub(1);
        break;
      }
    }
  }
}

VsmPool::Ticket VsmPool::submit(int ntasks, std::function<void(int)> fn, bool urgent) {
  Ticket b = std::make_shared<Batch>();
  b->fn = std::move(fn);
  b->n = ntasks < 0 ? 0 : ntasks;
  if (b->n == 0) return b;
  if (nthreads_ == 1) {  // no workers: run inline
    for (int i = 0; i < b->n; i++) b->fn(i);
    b->next = b->n;
    b->done = b->n;
    return b;
  }
  {
    std::lock_guard<std::mutex> lk(mu_);
    if (urgent) {
      // ahead of the background batches, but behind the urgent ones that are already waiting: first come, first served among
      // them (the look-ahead path's vertex sorts of chunk k must not be overtaken by those of chunk k + 1)
      auto it = queue_.begin();
      while (it != queue_.end() && (*it)->urgent) ++it;
      b->urgent = true;
      queue_.insert(it, b);
    } else {
      queue_.push_back(b);
    }
    posted_.fetch_add(1, std::memory_order_release);
  }
  // (as many wake-ups as there are tasks: every woken worker that finds the queue empty spins for spin_us_ before it sleeps
  // again, and sixteen of them spinning after every small batch is what pushes a rank over its CPU quota)
  const int ns = sleepers_.load();
  if (ns > 0) {
    if (b->n >= ns)
      cv_.notify_all();
    else
      for (int i = 0; i < b->n; i++) cv_.notify_one();
  }
  return b;
}

void VsmPool::wait(const Ticket &t) {
  while (t->done.load(std::memory_order_acquire) < t->n) {
    if (!work_one()) cpu_relax();
  }
}

void VsmPool::run(int ntasks, const std::function<void(int)> &fn) {
  if (ntasks <= 0) return;
  if (nthreads_ == 1 || ntasks == 1) {
    for (int i = 0; i < ntasks; i++) fn(i);
    return;
  }
  Ticket t = submit(ntasks, fn, true);
  wait(t);
}

// ---------------------------------------------------------------------------------------
// VsmForkJoin: lock-free fork-join for the fine-grained phases inside one Delaunay
// ---------------------------------------------------------------------------------------
VsmForkJoin::VsmForkJoin(int threads) : nthreads_(threads < 1 ? 1 : threads), aff_(vsm_affinity_current()) {
  // (Measured, per-frame matchFeatures(2) at 1242 x 375, tools/frame_timing.py: 578 us with the workers dealt over the
  // domains, 548 together, 500-520 with the caller's thread inside the domain as well - vsm_forkjoin_cpus() says where that
  // is.  A caller that only posts and waits while one more worker takes its share: 539 against 539, not kept.)
  for (int i = 1; i < nthreads_; i++) threads_.emplace_back([this, i] { worker(i); });
}

VsmForkJoin::~VsmForkJoin() {
  {
    std::lock_guard<std::mutex> lk(mu_);
    stop_ = true;
    gen_.store(0xffffffffffull, std::memory_order_release);
  }
  cv_.notify_all();
  for (auto &t : threads_) t.join();
}

// Task claiming is a CAS on (generation << 32 | next index): a worker that is late for
// generation g can never claim an index of generation g+1 with g's (possibly destroyed) closure.
bool VsmForkJoin::claim(uint64_t g, int n, int &idx) {
  uint64_t v = next_.load(std::memory_order_acquire);
  for (;;) {
    if ((v >> 32) != g || (int)(v & 0xffffffffu) >= n) return false;
    if (next_.compare_exchange_weak(v, v + 1, std::memory_order_acq_rel)) {
      idx = (int)(v & 0xffffffffu);
      return true;
    }
  }
}

void VsmForkJoin::worker(int index) {
  if (fj_domain() < 0)
    vsm_pin_this_thread(aff_);
  else if (vsm_forkjoin_per_core())
    vsm_pin_this_thread_core(aff_, fj_domain(), index);  // (core 0 is the caller's: vsm_forkjoin_cpus)
  else
    vsm_pin_this_thread_together(aff_, fj_domain());
  uint64_t seen = 0;
  for (;;) {
    // wait for a new generation: spin first, then block
    auto t0 = std::chrono::steady_clock::now();
    while (gen_.load(std::memory_order_acquire) == seen) {
      cpu_relax();
      if (std::chrono::duration_cast<std::chrono::microseconds>(std::chrono::steady_clock::now() - t0).count() > spin_us_) {
        std::unique_lock<std::mutex> lk(mu_);
        sleepers_.fetch_add(1);
        cv_.wait(lk, [&] { return gen_.load(std::memory_order_acquire) != seen; });
        sleepers_.fetch_sub(1);
        break;
      }
    }
    seen = gen_.load(std::memory_order_acquire);
    if (stop_) return;
    const std::function<void(int)> *fn = fn_;
    const int n = ntasks_;
    int i;
    while (claim(seen, n, i)) {
      (*fn)(i);
      done_.fetch_add(1, std::memory_order_acq_rel);
    }
  }
}

void VsmForkJoin::run(int ntasks, const std::function<void(int)> &fn) {
  if (ntasks <= 0) return;
  if (nthreads_ == 1 || ntasks == 1) {
    for (int i = 0; i < ntasks; i++) fn(i);
    return;
  }
  uint64_t g;
  {
    std::lock_guard<std::mutex> lk(mu_);
    g = (gen_.load(std::memory_order_relaxed) + 1) & 0xffffffffu;
    if (g == 0) g = 1;
    fn_ = &fn;
    ntasks_ = ntasks;
    done_.store(0, std::memory_order_relaxed);
    next_.store(g << 32, std::memory_order_release);
    gen_.store(g, std::memory_order_release);
  }
  if (sleepers_.load() > 0) cv_.notify_all();
  int i;
  while (claim(g, ntasks, i)) {
    fn(i);
    done_.fetch_add(1, std::memory_order_acq_rel);
  }
  while (done_.load(std::memory_order_acquire) < ntasks) cpu_relax();
}


// =======================================================================================
// ExactDelaunay: Triangle 1.6 divide-and-conquer with alternating cuts, decision for decision
// (viso/triangle.cpp: vertexsort :5447, vertexmedian :5513, alternateaxes :5583, mergehulls
// :5639, divconqrecurse :5963, divconqdelaunay :6161).  Triangle's robust float predicates
// (:2707, :3335) return the exact sign; on integer coordinates that is the integer sign used here.
//
// What is kept bit-for-bit: the randomised quicksort (it decides which of two duplicate points
// stays in the mesh, :6183-6197), the recursion shape (n>>1 splits, alternating axes, leaves of
// 2/3 vertices sorted by x) and every decision of the merge.  What is free to differ, because the
// result does not depend on it: vertexmedian's pivots (any exact selection yields the same
// partition of distinct points -> std::nth_element), triangle numbering (slot = 2*position, so
// sub-problems never share an allocator and can run on different threads), memory layout.
// Vertices are renumbered by their final sorted position so every sub-problem touches one
// contiguous slice of pt_.  tri_[t*8+o] = neighbour handle t2*4+o2, tri_[t*8+4+o] = vertex
// position or -1 for the ghost ("NULL") corner of a bounding triangle.
// =======================================================================================
#define KXY(k) VSM_KXY(k)
static inline uint64_t key_yx(uint64_t k) { return (((k >> 20) & 0x3fffu) << 14) | (k >> 34); }

// Triangle's generator (randomnation, :4046) is a linear congruence that starts from 1 for every triangulation
// (triangleinit, :4031): the SEEDS are one fixed sequence, whatever is sorted - only what a seed is divided by depends on the
// data.  The emulated sort is one long chain of dependent steps (a partition's pivot comes out of the array the partition
// before it left behind), and the multiply + modulo of the congruence is a tenth of every link: the first 128 k seeds are
// tabulated once per process, a draw is a load.
namespace {
constexpr uint32_t kSeedTable = 1u << 17;
const uint32_t *seed_table() {
  static const std::vector<uint32_t> t = [] {
    std::vector<uint32_t> v(kSeedTable);
    uint64_t sd = 1;
    for (uint32_t i = 0; i < kSeedTable; i++) {
      sd = (sd * 1366u + 150889u) % 714025u;
      v[i] = (uint32_t)sd;
    }
    return v;
  }();
  return t.data();
}
}  // namespace

uint32_t ExactDelaunay::rnd(uint32_t choices) {  // randomnation, :4046
  seed_ = (seed_ * 1366u + 150889u) % 714025u;
  // seed / (714025 / choices + 1): most calls come from small sub-arrays, whose divisor and its
  // 64-bit reciprocal are tabulated ((seed * magic) >> 64 is the exact quotient for 32-bit operands)
  struct Recip {
    uint64_t magic[256];
    Recip() {
      magic[0] = 0;
      for (uint32_t c = 1; c < 256; c++) magic[c] = UINT64_MAX / (714025u / c + 1) + 1;
    }
  };
  static const Recip table;
  if (choices < 256) return (uint32_t)(((__uint128_t)seed_ * table.magic[choices]) >> 64);
  return (uint32_t)(seed_ / (714025u / choices + 1));
}

// vertexsort (:5447) on packed (x,y) keys.  The Hoare partition is reproduced exactly, but without
// its data-dependent scan loops (a coin-flip branch per element).  Where the scans can stop is known
// up front: the left scan at keys >= pivot, the right scan at keys <= pivot, and the keys strictly
// between the two scan positions are always the original ones (swaps happen at the scan positions
// only).  So one branch-free pass writes two bit masks (GE, LE; eight keys per instruction where the
// CPU has AVX-512) and the partition becomes a loop over set bits:
//   left scan from `left`:   first GE bit in (left, right), else it stops at `right` itself (the key
//                            swapped in there is >= pivot; before the first swap the pivot is inside)
//   right scan from `right`: last LE bit in (left', right), else it stops at left' if that key equals
//                            the pivot, else at left' - 1 (the `left <= right` test of the reference)
//   swap while left' < right'.
// `left`, `right` and the element order after every step are those of the reference.
namespace {
inline void stop_masks_plain(const uint64_t *a, int32_t n, uint64_t ge_key, uint64_t lt_key, uint64_t *GE, uint64_t *LE) {
  for (int32_t w = 0; w * 64 < n; w++) {
    const int32_t cnt = std::min(64, n - w * 64);
    uint64_t g = 0, l = 0;
    for (int32_t i = 0; i < cnt; i++) {
      const uint64_t k = a[w * 64 + i];
      g |= (uint64_t)(k >= ge_key) << i;
      l |= (uint64_t)(k < lt_key) << i;
    }
    GE[w] = g;
    LE[w] = l;
  }
}
#if defined(__x86_64__)
__attribute__((target("avx512f"))) void stop_masks_avx512(const uint64_t *a, int32_t n, uint64_t ge_key, uint64_t lt_key, uint64_t *GE,
                                                          uint64_t *LE) {
  const __m512i vg = _mm512_set1_epi64((long long)ge_key), vl = _mm512_set1_epi64((long long)lt_key);
  for (int32_t w = 0; w * 64 < n; w++) {
    const int32_t cnt = std::min(64, n - w * 64);
    uint64_t g = 0, l = 0;
    for (int32_t i = 0; i < cnt; i += 8) {
      const __mmask8 valid = cnt - i >= 8 ? (__mmask8)0xff : (__mmask8)((1u << (cnt - i)) - 1);
      const __m512i v = _mm512_maskz_loadu_epi64(valid, (const void *)(a + w * 64 + i));
      g |= (uint64_t)_mm512_mask_cmpge_epu64_mask(valid, v, vg) << i;
      l |= (uint64_t)_mm512_mask_cmplt_epu64_mask(valid, v, vl) << i;
    }
    GE[w] = g;
    LE[w] = l;
  }
}
#ifndef VSM_SORT_WORD
#define VSM_SORT_WORD 1
#endif
__attribute__((target("avx512f"))) inline void stop_masks_word_avx512(const uint64_t *a, int32_t n, uint64_t ge_key, uint64_t lt_key, uint64_t &GE,
                                                                    uint64_t &LE) {  // n <= 64
  const __m512i vg = _mm512_set1_epi64((long long)ge_key), vl = _mm512_set1_epi64((long long)lt_key);
  uint64_t g = 0, l = 0;
  for (int32_t i = 0; i < n; i += 8) {
    const __mmask8 valid = n - i >= 8 ? (__mmask8)0xff : (__mmask8)((1u << (n - i)) - 1);
    const __m512i v = _mm512_maskz_loadu_epi64(valid, (const void *)(a + i));
    g |= (uint64_t)_mm512_mask_cmpge_epu64_mask(valid, v, vg) << i;
    l |= (uint64_t)_mm512_mask_cmplt_epu64_mask(valid, v, vl) << i;
  }
  GE = g;
  LE = l;
}
const bool kHaveAvx512 = __builtin_cpu_supports("avx512f") && __builtin_cpu_supports("avx512bw") && __builtin_cpu_supports("avx512vl") &&
                         __builtin_cpu_supports("bmi2") && !(getenv("VSM_NO_AVX512") && atoi(getenv("VSM_NO_AVX512")) != 0);
#else
void stop_masks_avx512(const uint64_t *, int32_t, uint64_t, uint64_t, uint64_t *, uint64_t *) {}
inline void stop_masks_word_avx512(const uint64_t *, int32_t, uint64_t, uint64_t, uint64_t &, uint64_t &) {}
const bool kHaveAvx512 = false;
#endif

// Parts of 3 to kTinyMax keys - most partitions of a sort - straight from a table: what the Hoare loop does depends
// only on which keys are below, equal to or above the pivot, i.e. on the two masks; the table (filled once by running
// the reference loop on every combination) holds the resulting order and the final (left, right) in four bytes: eight
// 3-bit source positions, left, right + 1.  With AVX-512 the keys are one register: masked load, two compares, a lookup,
// one permute, masked store.
struct TinyTable {
  static constexpr int kMax = 7;
  std::vector<uint32_t> e[kMax + 1];  // [n][GE | LE << n]
  TinyTable() {
    for (int n = 3; n <= kMax; n++) {
      e[n].resize((size_t)1 << (2 * n));
      for (int ge = 0; ge < (1 << n); ge++)
        for (int le = 0; le < (1 << n); le++) {
          int key[8], src[8];
          for (int i = 0; i < 8; i++) {
            src[i] = i;
            const bool g = (ge >> i) & 1, l = (le >> i) & 1;
            key[i] = g && l ? 1 : (g ? 2 : 0);  // pivot value 1 (a key neither >= nor <= does not occur)
          }
          int left = -1, right = n;  // vertexsort's loop (:5467-5487), bounds made safe for combinations that cannot occur
          while (left < right) {
            do {
              left++;
            } while (left <= right && left < n && key[left] < 1);
            do {
              right--;
            } while (left <= right && right >= 0 && key[right] > 1);
            if (left < right) {
              std::swap(key[left], key[right]);
              std::swap(src[left], src[right]);
            }
          }
          uint32_t w = 0;
          for (int i = 0; i < 8; i++) w |= (uint32_t)src[i] << (3 * i);
          w |= (uint32_t)(left & 15) << 24;
          w |= (uint32_t)((right + 1) & 15) << 28;
          e[n][ge | (le << n)] = w;
        }
    }
  }
};
constexpr int kTinyMax = TinyTable::kMax;

#if defined(__x86_64__)
__attribute__((target("avx512f,avx512bw,avx512vl,bmi2"))) inline void tiny_partition_avx512(uint64_t *a, int32_t n, uint64_t ge_key, uint64_t lt_key,
                                                                                             const TinyTable &tab, int32_t &left, int32_t &right) {
  const __mmask8 valid = (__mmask8)((1u << n) - 1);
  const __m512i v = _mm512_maskz_loadu_epi64(valid, (const void *)a);
  const unsigned ge = _mm512_mask_cmpge_epu64_mask(valid, v, _mm512_set1_epi64((long long)ge_key));
  const unsigned le = _mm512_mask_cmplt_epu64_mask(valid, v, _mm512_set1_epi64((long long)lt_key));
  const uint32_t w = tab.e[n][ge | (le << n)];
  const __m512i idx = _mm512_cvtepu8_epi64(_mm_cvtsi64_si128((long long)_pdep_u64(w & 0xffffffu, 0x0707070707070707ull)));
  _mm512_mask_storeu_epi64((void *)a, valid, _mm512_permutexvar_epi64(idx, v));
  left = (int32_t)((w >> 24) & 15);
  right = (int32_t)(w >> 28) - 1;
}
// A part of 3 to kTinyMax keys with EVERYTHING below it in the recursion, in one register: the keys are loaded once, every
// partition of the sub-tree is a table lookup and a permute of the register's lanes (the table's source positions, moved to
// the part's offset inside the register), the pending parts wait on a stack of eight bytes, two-key parts are settled
// after the one store at the end (they draw no random number, so when does not matter).  The depth-first order - left
// part first - and with it the order of the random draws is the reference's.  What it saves is memory: the plain form
// stores a part and loads its halves again, a masked 64-byte store feeding a load at another offset each time.
#ifndef VSM_SORT_SUBTREE
#define VSM_SORT_SUBTREE 1
#endif
__attribute__((target("avx512f,avx512bw,avx512vl,bmi2"))) void tiny_subtree_avx512(uint64_t *a, int32_t n, const TinyTable &tab, uint64_t &seed, const uint32_t *seeds, uint32_t &draw) {
  static const struct Magic {
    uint64_t m[8];
    Magic() {
      m[0] = 0;
      for (uint32_t c = 1; c < 8; c++) m[c] = UINT64_MAX / (714025u / c + 1) + 1;  // (ExactDelaunay::rnd)
    }
  } magic;
  const __mmask8 valid = (__mmask8)((1u << n) - 1);
  __m512i v = _mm512_maskz_loadu_epi64(valid, (const void *)a);
  uint32_t stk[8];
  int sp = 0, np = 0;
  uint32_t pair_at = 0;
  stk[sp++] = (uint32_t)n << 8;
  while (sp > 0) {
    const uint32_t t = stk[--sp];
    const int o = (int)(t & 0xffu), m = (int)(t >> 8);  // m >= 3
    seed = draw < kSeedTable ? seeds[draw] : (seed * 1366u + 150889u) % 714025u;
    draw++;
    const uint32_t r = (uint32_t)(((__uint128_t)seed * magic.m[m]) >> 64);
    // the pivot's key in every lane, its index bits cleared: keys >= that are the GE stops, keys below that + 2^20 the LE stops
    const __m512i pk = _mm512_permutexvar_epi64(_mm512_set1_epi64((long long)(o + (int)r)), v);
    const __m512i vg = _mm512_slli_epi64(_mm512_srli_epi64(pk, 20), 20), vl = _mm512_add_epi64(vg, _mm512_set1_epi64(1ll << 20));
    const unsigned in = (1u << m) - 1;
    const unsigned ge = ((unsigned)_mm512_cmpge_epu64_mask(v, vg) >> o) & in;
    const unsigned le = ((unsigned)_mm512_cmplt_epu64_mask(v, vl) >> o) & in;
    const uint32_t w = tab.e[m][ge | (le << m)];
    // byte i of `rel`: the source lane (inside the part) of the part's lane i, identity from m on; moved up by o lanes, the
    // o lanes below keep themselves
    const uint64_t rel = _pdep_u64(w & 0xffffffu, 0x0707070707070707ull);
    const uint64_t full = o ? (((rel + 0x0101010101010101ull * (uint64_t)o) << (8 * o)) | (0x0706050403020100ull & ((1ull << (8 * o)) - 1))) : rel;
    v = _mm512_permutexvar_epi64(_mm512_cvtepu8_epi64(_mm_cvtsi64_si128((long long)full)), v);
    const int left = (int)((w >> 24) & 15), right = (int)(w >> 28) - 1, rn = m - right - 1;
    pair_at = (pair_at & ((1u << (3 * np)) - 1)) | ((uint32_t)o << (3 * np));  // (written either way, kept if it is a pair)
    np += left == 2;
    pair_at = (pair_at & ((1u << (3 * np)) - 1)) | ((uint32_t)(o + right + 1) << (3 * np));
    np += rn == 2;
    stk[sp] = (uint32_t)(o + right + 1) | ((uint32_t)rn << 8);
    sp += rn > 2;
    stk[sp] = (uint32_t)o | ((uint32_t)left << 8);
    sp += left > 2;
  }
  _mm512_mask_storeu_epi64((void *)a, valid, v);
  for (int k = 0; k < np; k++) {
    uint64_t *p = a + ((pair_at >> (3 * k)) & 7u);
    const uint64_t x = p[0], y = p[1];
    const bool sw = KXY(x) > KXY(y);
    p[0] = sw ? y : x;
    p[1] = sw ? x : y;
  }
}
#else
inline void tiny_partition_avx512(uint64_t *, int32_t, uint64_t, uint64_t, const TinyTable &, int32_t &, int32_t &) {}
#define VSM_SORT_SUBTREE 0
inline void tiny_subtree_avx512(uint64_t *, int32_t, const TinyTable &, uint64_t &, const uint32_t *, uint32_t &) {}
#endif

// first set bit at a position >= from, or n
inline int32_t next_bit(const uint64_t *m, int32_t from, int32_t n) {
  if (from >= n) return n;
  int32_t w = from >> 6;
  uint64_t x = m[w] & (~0ull << (from & 63));
  const int32_t nw = (n + 63) >> 6;
  while (!x) {
    if (++w >= nw) return n;
    x = m[w];
  }
  return w * 64 + __builtin_ctzll(x);
}
// last set bit at a position <= from, or -1
inline int32_t prev_bit(const uint64_t *m, int32_t from) {
  if (from < 0) return -1;
  int32_t w = from >> 6;
  uint64_t x = m[w] & (~0ull >> (63 - (from & 63)));
  while (!x) {
    if (--w < 0) return -1;
    x = m[w];
  }
  return w * 64 + 63 - __builtin_clzll(x);
}
}  // namespace

// the same table for the device's emulation (vsm_dc.hip: tie_sort): sizes 3..7 one after the other, [GE | LE << n] inside a size
int vsm_host_tiny_table(uint32_t *out, int cap) {
  static const TinyTable tab;
  int at = 0;
  for (int n = 3; n <= TinyTable::kMax; n++) {
    if (at + (int)tab.e[n].size() > cap) return -1;
    memcpy(out + at, tab.e[n].data(), tab.e[n].size() * sizeof(uint32_t));
    at += (int)tab.e[n].size();
  }
  return at;
}

// The recursion itself is an explicit stack in the reference's depth-first order (left part first: the
// random numbers are drawn in that order).  Two-element parts draw no number, so they are settled on the
// spot, without a branch; parts of three or more are pushed, again without a branch (the slot is written
// either way, the stack pointer moves by the condition): what is left to mispredict per partition is the
// exit of the bit loop.
void ExactDelaunay::vertex_sort(uint64_t *a0, int32_t n0) {
  if (n0 < 2) return;
  if (n0 == 2) {
    if (KXY(a0[0]) > KXY(a0[1])) std::swap(a0[0], a0[1]);
    return;
  }
  sort_stack_.resize((size_t)n0 + 2);  // a pending right part per level of the recursion at most
  uint64_t *stack = sort_stack_.data();
  int32_t sp = 0;
  stack[sp++] = (uint64_t)(uint32_t)n0 << 32;  // (offset, length)
  // (draws from the tabulated seeds while the generator stands where triangleinit left it - always, as the code is used)
  const uint32_t *seeds = seed_table();
  uint32_t draw = seed_ == 1 ? 0u : kSeedTable;
  static const struct Recip {
    uint64_t magic[256];
    Recip() {
      magic[0] = 0;
      for (uint32_t c = 1; c < 256; c++) magic[c] = UINT64_MAX / (714025u / c + 1) + 1;
    }
  } recip;
  auto draw_rnd = [&](uint32_t choices) -> uint32_t {  // ExactDelaunay::rnd with the seed from the table
    seed_ = draw < kSeedTable ? seeds[draw] : (seed_ * 1366u + 150889u) % 714025u;
    draw++;
    if (choices < 256) return (uint32_t)(((__uint128_t)seed_ * recip.magic[choices]) >> 64);
    return (uint32_t)(seed_ / (714025u / choices + 1));
  };
  auto settle_pair = [](uint64_t *p, bool is_pair) {  // vertexsort on two elements if is_pair, nothing otherwise
    const uint64_t x = p[0], y = p[1];
    const bool sw = is_pair & (KXY(x) > KXY(y));
    p[0] = sw ? y : x;
    p[1] = sw ? x : y;
  };
  while (sp > 0) {
    const uint64_t top = stack[--sp];
    const int32_t off = (int32_t)(uint32_t)top, n = (int32_t)(top >> 32);  // n >= 3
    uint64_t *a = a0 + off;
    if (VSM_SORT_SUBTREE && n <= kTinyMax && kHaveAvx512) {  // the part and everything below it at once
      static const TinyTable *tiny = new TinyTable();
      tiny_subtree_avx512(a, n, *tiny, seed_, seeds, draw);
      continue;
    }
    const uint64_t pv = KXY(a[draw_rnd((uint32_t)n)]);
    // KXY(k) >= pv  <=>  k >= pv << 20;  KXY(k) <= pv  <=>  k < (pv + 1) << 20
    int32_t left = -1, right = n;
    if (n <= kTinyMax && kHaveAvx512) {
      static const TinyTable *tiny = new TinyTable();
      tiny_partition_avx512(a, n, pv << 20, (pv + 1) << 20, *tiny, left, right);
    } else if (VSM_SORT_WORD && n <= 64 && kHaveAvx512) {
      // one mask word per side: the scans are a count of trailing / leading zeros each, the masks never leave their registers
      uint64_t ge, le;
      stop_masks_word_avx512(a, n, pv << 20, (pv + 1) << 20, ge, le);
      for (;;) {
        const uint64_t g = left + 1 < 64 ? ge & (~0ull << (left + 1)) : 0ull;
        const int32_t l = std::min(g ? (int32_t)__builtin_ctzll(g) : n, right);
        if (l == right) {
          left = l;
          right = l - 1;
          break;
        }
        const uint64_t x = right - 1 >= 0 ? le & (~0ull >> (64 - right)) : 0ull;  // positions <= right - 1 (right >= 1 here: l < right)
        int32_t r = x ? 63 - (int32_t)__builtin_clzll(x) : -1;
        if (r <= l) r = ((le >> l) & 1) ? l : l - 1;
        left = l;
        right = r;
        if (l >= r) break;
        std::swap(a[l], a[r]);
      }
    } else {
    const int32_t nw = (n + 63) >> 6;
    uint64_t *GE = stop_.data(), *LE = stop_.data() + nw;
    if (kHaveAvx512)
      stop_masks_avx512(a, n, pv << 20, (pv + 1) << 20, GE, LE);
    else
      stop_masks_plain(a, n, pv << 20, (pv + 1) << 20, GE, LE);
    for (;;) {
      const int32_t l = std::min(next_bit(GE, left + 1, n), right);
      if (l == right) {  // ran into the key it swapped there itself: the right scan gives up at once
        left = l;
        right = l - 1;
        break;
      }
      int32_t r = prev_bit(LE, right - 1);
      if (r <= l) r = ((LE[l >> 6] >> (l & 63)) & 1) ? l : l - 1;
      left = l;
      right = r;
      if (l >= r) break;
      std::swap(a[l], a[r]);
    }
    }
    // the parts [0, left) and (right, n): `if (left > 1) vertexsort(...)`, `if (right < n - 2) vertexsort(...)`
    const int32_t rn = n - right - 1;
    settle_pair(a, left == 2);
    settle_pair(rn == 2 ? a + right + 1 : a, rn == 2);  // (a[0], a[1] exist and are rewritten unchanged otherwise)
    stack[sp] = (uint64_t)(uint32_t)(off + right + 1) | ((uint64_t)(uint32_t)rn << 32);
    sp += rn > 2;
    stack[sp] = (uint64_t)(uint32_t)off | ((uint64_t)(uint32_t)left << 32);
    sp += left > 2;
  }
}

// alternateaxes (:5583) for the whole tree at once.  The reference re-partitions every node with a
// randomised quick-select; for distinct points the outcome is just "the n>>1 smallest keys along
// the node's axis go left, leaves (2-3 points) are ordered by x", so it is computed here the way
// kd-trees are built from presorted lists: every node carries its points once in x order and once
// in y order; a cut along one axis halves that list in place and stable-partitions the other
// (one branch-free pass with an L1-resident rank lookup).  O(n) sequential work per level
// instead of an nth_element per node.  Points are named by their rank in the (x,y)-sorted array.
void ExactDelaunay::kd_order(int32_t m, VsmForkJoin *pool) {
  xl_.resize(m);
  yl_.resize(m);
  yr_.resize(m);
  tmp_.resize(m);
  ord_.resize(m);
  // y order: LSD radix sort of (y,x) (28 bits) carrying the x-rank
  {
    std::vector<uint64_t> &k0 = ybuf0_, &k1 = ybuf1_;
    k0.resize(m);
    k1.resize(m);
    for (int32_t i = 0; i < m; i++) k0[i] = (key_yx(key_[i]) << 32) | (uint32_t)i;
    uint64_t *src = k0.data(), *dst = k1.data();
    for (int pass = 0; pass < 3; pass++) {
      const int sh = 32 + pass * 10;
      uint32_t cnt[1025] = {0};
      for (int32_t i = 0; i < m; i++) cnt[((src[i] >> sh) & 1023) + 1]++;
      for (int b = 0; b < 1024; b++) cnt[b + 1] += cnt[b];
      for (int32_t i = 0; i < m; i++) dst[cnt[(src[i] >> sh) & 1023]++] = src[i];
      std::swap(src, dst);
    }
    for (int32_t i = 0; i < m; i++) {
      const uint32_t e = (uint32_t)(src[i] & 0xffffffffu);
      yl_[i] = e;
      yr_[e] = (uint32_t)i;
    }
  }
  for (int32_t i = 0; i < m; i++) xl_[i] = (uint32_t)i;
  // explicit stack of (offset, n, axis); lists of a node occupy [off, off+n) of xl_/yl_.  Sub-trees
  // touch disjoint slices, so once the top levels have produced enough of them they are dealt out
  // to the pool (each task with its own partition scratch, a slice of tmp_).
  struct Nd {
    int32_t off, n, axis;
  };
  const uint32_t *yr = yr_.data();
  auto split = [&](const Nd &nd, uint32_t *tmp) -> int32_t {  // partitions the node's lists, returns the cut
    uint32_t *xl = xl_.data() + nd.off, *yl = yl_.data() + nd.off;
    const int32_t div = nd.n >> 1;
    int32_t l = 0, r = 0;
    if (nd.axis == 0) {  // cut in x: xl splits in place, yl is partitioned by x-rank
      const uint32_t pivot = xl[div];
      for (int32_t i = 0; i < nd.n; i++) {
        const uint32_t e = yl[i];
        const int left = e < pivot;
        yl[l] = e;
        tmp[r] = e;
        l += left;
        r += 1 - left;
      }
      memcpy(yl + l, tmp, (size_t)r * sizeof(uint32_t));
    } else {  // cut in y
      const uint32_t pivot = yr[yl[div]];
      for (int32_t i = 0; i < nd.n; i++) {
        const uint32_t e = xl[i];
        const int left = yr[e] < pivot;
        xl[l] = e;
        tmp[r] = e;
        l += left;
        r += 1 - left;
      }
      memcpy(xl + l, tmp, (size_t)r * sizeof(uint32_t));
    }
    return div;
  };
  auto subtree = [&](Nd root) {  // depth-first over one sub-tree; scratch = tmp_ over the node's own slice
    Nd st[64];
    int sp = 0;
    st[sp++] = root;
    uint32_t *tmp = tmp_.data() + root.off;
    while (sp > 0) {
      const Nd nd = st[--sp];
      if (nd.n <= 3) {
        const uint32_t *xl = xl_.data() + nd.off;
        for (int32_t i = 0; i < nd.n; i++) ord_[nd.off + i] = xl[i];
        continue;
      }
      const int32_t div = split(nd, tmp);
      st[sp++] = Nd{nd.off + div, nd.n - div, 1 - nd.axis};
      st[sp++] = Nd{nd.off, div, 1 - nd.axis};
    }
  };
  const int nthreads = pool ? pool->size() : 1;
  if (nthreads <= 1 || m < 2048) {
    subtree(Nd{0, m, 0});
  } else {
    std::vector<Nd> level{Nd{0, m, 0}};
    while ((int)level.size() < nthreads) {  // top levels: the root alone, then a thread per node (their slices are disjoint)
      std::vector<Nd> next(level.size() * 2);
      auto one = [&](int t) {
        const Nd nd = level[t];
        const int32_t div = split(nd, tmp_.data() + nd.off);
        next[2 * t] = Nd{nd.off, div, 1 - nd.axis};
        next[2 * t + 1] = Nd{nd.off + div, nd.n - div, 1 - nd.axis};
      };
      if (level.size() > 1)
        pool->run((int)level.size(), one);
      else
        one(0);
      level.swap(next);
    }
    pool->run((int)level.size(), [&](int t) { subtree(level[t]); });
  }
  // bring the keys into their final order
  k2_.resize(m);
  for (int32_t i = 0; i < m; i++) k2_[i] = key_[ord_[i]];
  key_.swap(k2_);
}

// tree layout: the x-sorted array is cut in the middle, the halves alternate their cut axis.  Nodes small
// enough become tasks; the internal nodes are listed by depth (left to right), those of at most
// device_top_points points apart
int32_t ExactDelaunay::build_tree(int32_t off, int32_t n, int axis, int32_t depth) {
  const int32_t me = (int32_t)nodes_.size();
  nodes_.push_back(Node{off, n, axis, 0, -1, -1, {0, 0}, {0, 0}});
  if (n > max_task_ && n > 3) {
    std::vector<std::vector<int32_t>> &lists = n <= device_top_ ? dev_levels_ : levels_;
    if ((int32_t)lists.size() <= depth) lists.resize(depth + 1);
    lists[depth].push_back(me);
    const int32_t divider = n >> 1;
    const int32_t l = build_tree(off, divider, 1 - axis, depth + 1);
    const int32_t r = build_tree(off + divider, n - divider, 1 - axis, depth + 1);
    nodes_[me].left = l;
    nodes_[me].right = r;
  } else {
    tasks_.push_back(Task{off, n, axis, me});
  }
  return me;
}

void ExactDelaunay::list_triangles() {
  if (listed_) return;
  listed_ = true;
  ntri_out_ = 0;
  tri_out_.resize((size_t)m_ * 6);
  for (int32_t t = 0; t < 2 * m_; t++)
    if (slot_vertices(t, &tri_out_[(size_t)ntri_out_ * 3])) ntri_out_++;
}

bool ExactDelaunay::prepare(const int32_t *x, const int32_t *y, int32_t n, int32_t max_task_points, VsmForkJoin *pool,
                            int32_t device_top_points, bool defer_order, bool defer_ties) {
  ntri_out_ = 0;
  m_ = 0;
  listed_ = true;
  tasks_.clear();
  dmerges_.clear();
  dlevels_.clear();
  for (auto &lv : levels_) lv.clear();  // (the lists keep their memory)
  for (auto &lv : dev_levels_) lv.clear();
  nodes_.clear();
  seed_ = 1;  // triangleinit(), :4031
  has_ties_ = false;
  ties_resolved_ = true;
  patches_.clear();
  n_in_ = n;
  if (n < 2) return false;
  key_.resize(n);
  stop_.resize(2 * ((size_t)n / 64 + 2));
  for (int32_t i = 0; i < n; i++) key_[i] = ((uint64_t)(uint32_t)x[i] << 34) | ((uint64_t)(uint32_t)y[i] << 20) | (uint32_t)i;
  uint64_t *a = key_.data();
  auto clk = [] { return std::chrono::steady_clock::now(); };
  auto ns = [](std::chrono::steady_clock::time_point u, std::chrono::steady_clock::time_point v) {
    return (long)std::chrono::duration_cast<std::chrono::nanoseconds>(v - u).count();
  };
  const auto p0 = clk();
  if (defer_ties) {
    emu_.assign(key_.begin(), key_.end());
    // stable LSD radix sort by (x, y) (28 bits): equal points stay in input order
    rs_.resize(n);
    uint64_t *src = a, *dst = rs_.data();
    for (int pass = 0; pass < 3; pass++) {
      const int sh = 20 + pass * 10;
      uint32_t cnt[1025] = {0};
      for (int32_t i = 0; i < n; i++) cnt[((src[i] >> sh) & 1023) + 1]++;
      for (int b = 0; b < 1024; b++) cnt[b + 1] += cnt[b];
      for (int32_t i = 0; i < n; i++) dst[cnt[(src[i] >> sh) & 1023]++] = src[i];
      std::swap(src, dst);
    }
    if (src != a) memcpy(a, src, (size_t)n * sizeof(uint64_t));
  } else {
    vertex_sort(a, n);
  }
  const auto p1 = clk();
  int32_t m = 0;  // duplicates: the first one in sorted order survives (:6183)
  for (int32_t j = 1; j < n; j++)
    if (KXY(a[m]) != KXY(a[j])) a[++m] = a[j];
  m++;
  if (defer_ties && m < n) {
    has_ties_ = true;
    ties_resolved_ = false;
  }
  if (m < 2) return false;
  if ((size_t)m * 16 > tri_.size()) tri_.resize((size_t)m * 16);
  pt_.resize(m);
  id_.resize(m);
  for (int t = 2 * m - 2; t < 2 * m; t++)  // the two unused slots
    for (int k = 0; k < 3; k++) tri_[(size_t)t * 8 + 4 + k] = -1;
  key_.resize(m);
  ordered_ = !defer_order;
  if (!defer_order) kd_order(m, pool);
  t_sort_ = ns(p0, p1);
  t_kd_ = ns(p1, clk());
  m_ = m;
  listed_ = false;
  max_task_ = max_task_points < 2 ? 2 : max_task_points;
  device_top_ = device_top_points;
  build_tree(0, m, 0, 0);
  for (int d = (int)dev_levels_.size() - 1; d >= 0; d--) {
    if (dev_levels_[d].empty()) continue;
    dlevels_.push_back((int32_t)dev_levels_[d].size());
    for (int32_t q : dev_levels_[d]) dmerges_.push_back(Merge{nodes_[q].off, nodes_[q].n, nodes_[q].axis, q, nodes_[q].left, nodes_[q].right});
  }
  return true;
}

// one internal node: what divconqrecurse does after its two recursive calls (viso/triangle.cpp, see DcMesh::recurse)
void ExactDelaunay::merge_node(int32_t q) {
  const DcMesh mesh = this->mesh();
  Node &nd = nodes_[q];
  Node &l = nodes_[nd.left], &r = nodes_[nd.right];
  nd.fl = l.fl;
  nd.fr = r.fr;
  OTri il = l.fr, ir = r.fl;
  int32_t tcur = 2 * (nd.off + (nd.n >> 1)) - 2;
  mesh.merge_hulls(nd.fl, il, ir, nd.fr, nd.axis, tcur);
}

void ExactDelaunay::solve_merges() {
  for (const Merge &mg : dmerges_) merge_node(mg.node);  // deepest level first
}

void ExactDelaunay::solve_tasks(VsmForkJoin *pool) {
  const DcMesh mesh = this->mesh();
  // (-DVSM_DC_ITER=1: the explicit-stack form the GPU lanes use, dc_build_iter - same result)
#ifndef VSM_DC_ITER
#define VSM_DC_ITER 0
#endif
  constexpr bool iter = VSM_DC_ITER != 0;
  auto one = [&](int t) {
    Node &nd = nodes_[tasks_[t].node];
    if (iter)
      dc_build_iter<40>(mesh, nd.off, nd.n, nd.axis, nd.fl, nd.fr);
    else
      mesh.recurse(nd.off, nd.n, nd.axis, nd.fl, nd.fr);
  };
  if (pool && pool->size() > 1 && tasks_.size() > 1) {
    pool->run((int)tasks_.size(), one);
  } else {
    for (int t = 0; t < (int)tasks_.size(); t++) one(t);
  }
}

void ExactDelaunay::finish(VsmForkJoin *pool) {
  for (int li = (int)levels_.size() - 1; li >= 0; li--) {  // bottom-up
    const auto &lv = levels_[li];
    if (pool && pool->size() > 1 && lv.size() > 1) {
      pool->run((int)lv.size(), [&](int t) { merge_node(lv[t]); });
    } else {
      for (int32_t q : lv) merge_node(q);
    }
  }
}

void ExactDelaunay::resolve_ties() {
  if (ties_resolved_) return;
  seed_ = 1;  // triangleinit(), :4031
  uint64_t *e = emu_.data();
  const int32_t n = n_in_;
  vertex_sort(e, n);
  for (int32_t i = 0; i < n;) {  // of equal points the first one in this order is the vertex (:6183)
    int32_t j = i + 1;
    while (j < n && KXY(e[j]) == KXY(e[i])) j++;
    if (j - i > 1) {
      int32_t rep = (int32_t)(e[i] & 0xfffffu);
      for (int32_t k = i + 1; k < j; k++) rep = std::min(rep, (int32_t)(e[k] & 0xfffffu));
      const int32_t first = (int32_t)(e[i] & 0xfffffu);
      if (rep != first) patches_.push_back(std::make_pair(rep, first));
    }
    i = j;
  }
  ties_resolved_ = true;
}

int32_t ExactDelaunay::sort_ties(const uint64_t *keys, int32_t n, int32_t *pairs, int32_t cap) {
  if (n < 2) return 0;
  emu_.assign(keys, keys + n);
  stop_.resize(2 * ((size_t)n / 64 + 2));
  seed_ = 1;  // triangleinit(), :4031
  uint64_t *e = emu_.data();
  vertex_sort(e, n);
  int32_t np = 0;
  for (int32_t i = 0; i < n;) {  // of equal points the first one in this order is the vertex (:6183)
    int32_t j = i + 1;
    while (j < n && KXY(e[j]) == KXY(e[i])) j++;
    if (j - i > 1) {
      int32_t rep = (int32_t)(e[i] & 0xfffffu);
      for (int32_t k = i + 1; k < j; k++) rep = std::min(rep, (int32_t)(e[k] & 0xfffffu));
      const int32_t first = (int32_t)(e[i] & 0xfffffu);
      if (rep != first) {
        if (np >= cap) return -1;
        pairs[2 * np] = rep;
        pairs[2 * np + 1] = first;
        np++;
      }
    }
    i = j;
  }
  return np;
}

void ExactDelaunay::apply_ties() {
  if (!ties_resolved_) resolve_ties();
  if (patches_.empty()) return;
  tie_bits_.assign(((size_t)n_in_ + 63) / 64, 0);
  for (const auto &pt : patches_) tie_bits_[pt.first >> 6] |= 1ull << (pt.first & 63);
  for (int32_t pos = 0; pos < m_; pos++) {
    const int32_t q = id_[pos];
    if (!((tie_bits_[q >> 6] >> (q & 63)) & 1)) continue;
    for (const auto &pt : patches_)
      if (pt.first == q) {
        id_[pos] = pt.second;
        break;
      }
  }
  patches_.clear();
}

void ExactDelaunay::run(const int32_t *x, const int32_t *y, int32_t n, VsmForkJoin *pool, VsmPool *async) {
  static const bool dbg = vsm_host_debug_timing();
  const int nthreads = pool ? pool->size() : 1;
  // one task (the whole array) when single-threaded, about one sub-tree per thread otherwise
  int32_t max_task = n;
  if (nthreads > 1 && n >= 256) {
    int depth = 0;
    while ((1 << depth) < nthreads) depth++;
    max_task = std::max(63, (n >> depth) + 1);
  }
  // with somebody to take it, the emulated vertex sort runs next to the triangulation instead of in front of it
#ifndef VSM_TIES_APART_MIN
#define VSM_TIES_APART_MIN 1024
#endif
  const bool apart = async != nullptr && async->size() > 1 && n >= VSM_TIES_APART_MIN;
  if (!prepare(x, y, n, max_task, pool, 0, false, apart)) return;
  VsmPool::Ticket ties;
  if (has_ties_) ties = async->submit(1, [this](int) { resolve_ties(); }, true);
  const auto p2 = std::chrono::steady_clock::now();
  solve_tasks(pool);
  finish(pool);
  if (ties) {
    async->wait(ties);
    apply_ties();
  }
  if (dbg && n > 3000) {
    static std::atomic<long> calls{0}, t_sort{0}, t_kd{0}, t_dc{0};
    t_sort += t_sort_;
    t_kd += t_kd_;
    t_dc += (long)std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now() - p2).count();
    if (++calls % 100 == 0)
      fprintf(stderr, "  ExactDelaunay (n > 3000, %d threads), mean us: sort %.0f, kd order %.0f, divide&conquer %.0f\n",
              nthreads, t_sort / 1e3 / calls, t_kd / 1e3 / calls, t_dc / 1e3 / calls);
  }
}

// =======================================================================================
// O1 removeOutliers, viso/matcher.cpp:1207-1377
// =======================================================================================
// `in` may be the list the GPU exported (host-mapped memory): it is read twice, sequentially, and
// only the survivors are copied.  The per-match quantities the support test compares (flow and
// disparity, :1290-1340) are gathered into compact arrays first, so the triangle loop touches
// 12 bytes per vertex instead of a 48-byte record.
void vsm_host_outliers_begin(VsmHostWork &w, const vsm_p_match *in, int32_t n, int method) {
  w.x.resize(n);
  w.y.resize(n);
  w.fu.resize(n);
  w.fv.resize(n);
  w.dp.resize(n);
  for (int32_t i = 0; i < n; i++) {
    const vsm_p_match &a = in[i];
    w.x[i] = (int32_t)a.u1c;
    w.y[i] = (int32_t)a.v1c;
    w.fu[i] = a.u1c - a.u1p;
    w.fv[i] = a.v1c - a.v1p;
    w.dp[i] = method == 1 ? a.u1c - a.u2c : a.u1p - a.u2p;
  }
}

void vsm_host_outliers_begin_xy(VsmHostWork &w, const uint32_t *xy, int32_t n) {
  w.x.resize(n);
  w.y.resize(n);
  for (int32_t i = 0; i < n; i++) {
    w.x[i] = (int32_t)(xy[i] & 0xffffu);
    w.y[i] = (int32_t)(xy[i] >> 16);
  }
}

void vsm_host_outliers_begin_flows(VsmHostWork &w, const vsm_p_match *in, int32_t n, int method) {
  w.fu.resize(n);
  w.fv.resize(n);
  w.dp.resize(n);
  float *fu = w.fu.data(), *fv = w.fv.data(), *dp = w.dp.data();
  auto part = [&](int32_t i0, int32_t i1) {
    for (int32_t i = i0; i < i1; i++) {
      const vsm_p_match &a = in[i];
      fu[i] = a.u1c - a.u1p;
      fv[i] = a.v1c - a.v1p;
      dp[i] = method == 1 ? a.u1c - a.u2c : a.u1p - a.u2p;
    }
  };
  const int T = w.pool ? w.pool->size() : 1;
  if (T > 1 && n >= 2048)  // (the per-frame path's final list, 350 KB where the device exported it: 13 -> 4 us)
    w.pool->run(T, [&](int k) { part((int32_t)((int64_t)n * k / T), (int32_t)((int64_t)n * (k + 1) / T)); });
  else
    part(0, n);
}

// support of every match from the triangulation in w.del, survivors (support >= 4) to `out`
void vsm_host_outliers_end(VsmHostWork &w, const vsm_params &p, const vsm_p_match *in, int32_t n, int method,
                           std::vector<vsm_p_match> &out) {
  vsm_host_count_support(w, p, n, method);
  const int T = w.pool ? w.pool->size() : 1;
  if (T > 1 && n >= 2048) {
    // the per-frame path's final list (350 KB where the device exported it): every fork-join thread counts the survivors of
    // its range, then copies them to their place behind those of the ranges in front
    const int32_t *support = w.support.data();
    int32_t cnt[64];
    const int Tn = std::min(T, 64);
    w.pool->run(Tn, [&](int k) {
      const int32_t i0 = (int32_t)((int64_t)n * k / Tn), i1 = (int32_t)((int64_t)n * (k + 1) / Tn);
      int32_t c = 0;
      for (int32_t i = i0; i < i1; i++) c += support[i] >= 4;
      cnt[k] = c;
    });
    int32_t total = 0;
    for (int k = 0; k < Tn; k++) {
      const int32_t c = cnt[k];
      cnt[k] = total;
      total += c;
    }
    out.resize((size_t)total);  // (about the previous frame's size: little to initialise)
    vsm_p_match *dst = out.data();
    w.pool->run(Tn, [&](int k) {
      const int32_t i0 = (int32_t)((int64_t)n * k / Tn), i1 = (int32_t)((int64_t)n * (k + 1) / Tn);
      vsm_p_match *d = dst + cnt[k];
      int32_t i = i0;
      while (i < i1) {
        while (i < i1 && support[i] < 4) i++;
        int32_t j = i;
        while (j < i1 && support[j] >= 4) j++;
        if (j > i) {
          memcpy((void *)d, (const void *)(in + i), (size_t)(j - i) * sizeof(vsm_p_match));
          d += j - i;
        }
        i = j;
      }
    });
    return;
  }
  vsm_host_keep_supported(in, w.support.data(), n, out);
}

void vsm_host_count_support(VsmHostWork &w, const vsm_params &p, int32_t n, int method) {
  const float ftol = (float)p.outlier_flow_tolerance, dtol = (float)p.outlier_disp_tolerance;
  const float *fu = w.fu.data(), *fv = w.fv.data(), *dp = w.dp.data();
  const ExactDelaunay &del = w.del;
  // (support is a plain sum over triangle edges: the slot order is as good as Triangle's output order,
  // and so is any split of the slots over threads - integer sums)
  auto votes_of = [&](int32_t t0, int32_t t1, int32_t *support) {
    for (int32_t t = t0; t < t1; t++) {
      int32_t q[3];
      if (!del.slot_vertices(t, q)) continue;
      static const int E[3][2] = {{0, 1}, {1, 2}, {0, 2}};
      for (int e = 0; e < 3; e++) {
        const int32_t a = q[E[e][0]], b = q[E[e][1]];
        const bool flow_ok = fabsf(fu[a] - fu[b]) + fabsf(fv[a] - fv[b]) < ftol;
        const bool disp_ok = fabsf(dp[a] - dp[b]) < dtol;
        const bool ok = method == 0 ? flow_ok : (method == 1 ? disp_ok : (disp_ok && flow_ok));
        if (ok) {
          support[a]++;
          support[b]++;
        }
      }
    }
  };
  const int32_t slots = del.num_slots();
  const int T = w.pool ? w.pool->size() : 1;
  if (T > 1 && n >= 2048) {  // the per-frame path: every thread counts a range of slots into its own array
    w.support.resize(n);
    w.support_parts.resize((size_t)T * n);
    w.pool->run(T, [&](int k) {
      int32_t *mine = w.support_parts.data() + (size_t)k * n;
      memset(mine, 0, (size_t)n * sizeof(int32_t));
      votes_of((int32_t)((int64_t)slots * k / T), (int32_t)((int64_t)slots * (k + 1) / T), mine);
    });
    w.pool->run(T, [&](int k) {
      const int32_t i0 = (int32_t)((int64_t)n * k / T), i1 = (int32_t)((int64_t)n * (k + 1) / T);
      int32_t *sum = w.support.data();
      for (int32_t i = i0; i < i1; i++) sum[i] = w.support_parts[i];
      for (int q = 1; q < T; q++) {
        const int32_t *part = w.support_parts.data() + (size_t)q * n;
        for (int32_t i = i0; i < i1; i++) sum[i] += part[i];
      }
    });
    return;
  }
  // One thread: the triangle slots follow the kd order of the points, so per-POSITION copies of what the test compares
  // (16 bytes per point, one cache line for a triangle's corners more often than not) make the walk local; the counts
  // go back to the matches through the ids at the end.
  const int32_t m = del.points();
  const int32_t *id = del.ids();
  if (m >= 1024) {
    struct Q {
      float fu, fv, dp;
      int32_t votes;
    };
    w.support_pos.resize((size_t)m * 4);
    Q *qp = reinterpret_cast<Q *>(w.support_pos.data());
    for (int32_t pos = 0; pos < m; pos++) {
      const int32_t i = id[pos];
      qp[pos] = Q{fu[i], fv[i], dp[i], 0};
    }
    for (int32_t t = 0; t < slots; t++) {
      const int32_t *v = del.slot_record(t);
      if ((v[0] | v[1] | v[2]) < 0) continue;
      Q &a = qp[v[1]], &b = qp[v[2]], &c3 = qp[v[0]];
      Q *const tri3[3] = {&a, &b, &c3};
      static const int E[3][2] = {{0, 1}, {1, 2}, {0, 2}};
      for (int e = 0; e < 3; e++) {
        Q &x = *tri3[E[e][0]], &y = *tri3[E[e][1]];
        const bool flow_ok = fabsf(x.fu - y.fu) + fabsf(x.fv - y.fv) < ftol;
        const bool disp_ok = fabsf(x.dp - y.dp) < dtol;
        const bool ok = method == 0 ? flow_ok : (method == 1 ? disp_ok : (disp_ok && flow_ok));
        x.votes += ok;
        y.votes += ok;
      }
    }
    w.support.assign(n, 0);
    for (int32_t pos = 0; pos < m; pos++) w.support[id[pos]] = qp[pos].votes;
    return;
  }
  w.support.assign(n, 0);
  votes_of(0, slots, w.support.data());
}

// survivors of the support test (support >= 4, viso/matcher.cpp:1369-1371), in list order; almost all matches
// survive, so they are copied run by run
void vsm_host_keep_supported(const vsm_p_match *in, const int32_t *support, int32_t n, std::vector<vsm_p_match> &out) {
  out.clear();
  out.reserve((size_t)std::max(n, 0));  // (no resize: that would write the whole list once for nothing)
  int32_t i = 0;
  while (i < n) {
    while (i < n && support[i] < 4) i++;
    int32_t j = i;
    while (j < n && support[j] >= 4) j++;
    if (j > i) out.insert(out.end(), in + i, in + j);
    i = j;
  }
}

// the same inside the list itself (the runs move down over the gaps)
void vsm_host_keep_supported(std::vector<vsm_p_match> &list, const int32_t *support) {
  const int32_t n = (int32_t)list.size();
  vsm_p_match *base = list.data(), *dst = base;
  int32_t i = 0;
  while (i < n) {
    while (i < n && support[i] < 4) i++;
    int32_t j = i;
    while (j < n && support[j] >= 4) j++;
    if (j > i) {
      if (dst != base + i) memmove((void *)dst, (const void *)(base + i), (size_t)(j - i) * sizeof(vsm_p_match));
      dst += j - i;
    }
    i = j;
  }
  list.resize((size_t)(dst - base));
}

void vsm_host_remove_outliers_from(VsmHostWork &w, const vsm_params &p, const vsm_p_match *in, int32_t n, int method,
                                   std::vector<vsm_p_match> &out) {
  if (n <= 3) {  // the reference leaves short lists alone (:1210)
    out.assign(in, in + std::max(n, 0));
    return;
  }
  vsm_host_outliers_begin(w, in, n, method);
  static const bool dbg = vsm_host_debug_timing();
  const auto c0 = std::chrono::steady_clock::now();
  w.del.run(w.x.data(), w.y.data(), n, w.pool, w.async);
  const auto c1 = std::chrono::steady_clock::now();
  vsm_host_outliers_end(w, p, in, n, method, out);
  if (dbg && n > 3000) {
    static std::atomic<long> calls{0}, us_del{0}, us_rest{0};
    const auto c2 = std::chrono::steady_clock::now();
    us_del += std::chrono::duration_cast<std::chrono::microseconds>(c1 - c0).count();
    us_rest += std::chrono::duration_cast<std::chrono::microseconds>(c2 - c1).count();
    if (++calls % 199 == 0)
      fprintf(stderr, "  removeOutliers (n > 3000), mean: Delaunay %.1f us, support + compaction %.1f us\n",
              (double)us_del / calls, (double)us_rest / calls);
  }
}

void vsm_host_remove_outliers(VsmHostWork &w, const vsm_params &p, std::vector<vsm_p_match> &m, int method) {
  if ((int32_t)m.size() <= 3) return;
  w.tmp_list.swap(m);
  vsm_host_remove_outliers_from(w, p, w.tmp_list.data(), (int32_t)w.tmp_list.size(), method, m);
}

// =======================================================================================
// M4 computePriorStatistics, viso/matcher.cpp:734-868
// =======================================================================================
void vsm_host_prior_statistics(const vsm_params &p, const int32_t *dims_c, const std::vector<vsm_p_match> &m,
                               int method, std::vector<float> &ranges) {
  const float bs = (float)p.match_binsize;
  const int ub = (int)ceilf((float)dims_c[0] / bs), vb = (int)ceilf((float)dims_c[1] / bs);
  const int nb = ub * vb, ns = method == 2 ? 4 : 2;
  // The reference adds every match to the nine bins around its own (:789-812).  Minimum, maximum and count do not care about
  // the order: here a match goes into its OWN bin only and the bins are then widened over their 3 x 3 neighbourhoods - the
  // same values with a ninth of the updates (18 -> 5 us for a first-pass list of 870 matches).
  static thread_local std::vector<float> lo0, hi0, lo, hi;
  static thread_local std::vector<int32_t> cnt0, cnt;
  lo0.assign((size_t)nb * 8, +1000000.f);
  hi0.assign((size_t)nb * 8, -1000000.f);
  cnt0.assign(nb, 0);
  auto flows_of = [&](const vsm_p_match &it, float *d, float &ur, float &vr) {
    for (int i = 0; i < 8; i++) d[i] = 0;
    ur = it.u1c;
    vr = it.v1c;
    if (method == 0) {
      d[0] = it.u1p - it.u1c;
      d[1] = it.v1p - it.v1c;
      d[2] = it.u1c - it.u1p;
      d[3] = it.v1c - it.v1p;
    } else if (method == 1) {
      d[0] = it.u2c - it.u1c;
      d[2] = it.u1c - it.u2c;
    } else {
      d[0] = it.u2p - it.u1p;
      d[2] = it.u2c - it.u2p;
      d[3] = it.v2c - it.v2p;
      d[4] = it.u1c - it.u2c;
      d[6] = it.u1p - it.u1c;
      d[7] = it.v1p - it.v1c;
      ur = it.u1p;
      vr = it.v1p;
    }
  };
  bool strays = false;  // matches whose own bin lies outside the grid (the reference clamps the neighbourhood's ends, not its middle)
  for (const vsm_p_match &it : m) {
    float d[8], ur, vr;
    flows_of(it, d, ur, vr);
    const int ubin = (int)floorf(ur / bs), vbin = (int)floorf(vr / bs);
    if (ubin < 0 || ubin >= ub || vbin < 0 || vbin >= vb) {
      strays = true;
      continue;
    }
    const int b = vbin * ub + ubin;
    cnt0[b]++;
    for (int i = 0; i < ns * 2; i++) {
      lo0[b * 8 + i] = std::min(lo0[b * 8 + i], d[i]);
      hi0[b * 8 + i] = std::max(hi0[b * 8 + i], d[i]);
    }
  }
  lo.assign((size_t)nb * 8, +1000000.f);
  hi.assign((size_t)nb * 8, -1000000.f);
  cnt.assign(nb, 0);
  for (int v = 0; v < vb; v++)
    for (int u = 0; u < ub; u++) {
      const int b = v * ub + u;
      for (int v2 = std::max(v - 1, 0); v2 <= std::min(v + 1, vb - 1); v2++)
        for (int u2 = std::max(u - 1, 0); u2 <= std::min(u + 1, ub - 1); u2++) {
          const int s2 = v2 * ub + u2;
          if (!cnt0[s2]) continue;
          cnt[b] += cnt0[s2];
          for (int i = 0; i < ns * 2; i++) {
            lo[b * 8 + i] = std::min(lo[b * 8 + i], lo0[s2 * 8 + i]);
            hi[b * 8 + i] = std::max(hi[b * 8 + i], hi0[s2 * 8 + i]);
          }
        }
    }
  if (strays)
    for (const vsm_p_match &it : m) {
      float d[8], ur, vr;
      flows_of(it, d, ur, vr);
      const int ubin = (int)floorf(ur / bs), vbin = (int)floorf(vr / bs);
      if (!(ubin < 0 || ubin >= ub || vbin < 0 || vbin >= vb)) continue;
      const int u0 = std::min(std::max(ubin - 1, 0), ub - 1), u1 = std::min(std::max(ubin + 1, 0), ub - 1);
      const int v0 = std::min(std::max(vbin - 1, 0), vb - 1), v1 = std::min(std::max(vbin + 1, 0), vb - 1);
      for (int v = v0; v <= v1; v++)
        for (int u = u0; u <= u1; u++) {
          const int b = v * ub + u;
          cnt[b]++;
          for (int i = 0; i < ns * 2; i++) {
            lo[b * 8 + i] = std::min(lo[b * 8 + i], d[i]);
            hi[b * 8 + i] = std::max(hi[b * 8 + i], d[i]);
          }
        }
    }
  ranges.assign((size_t)nb * 16, 0.f);
  for (int b = 0; b < nb; b++) {
    float *r = &ranges[(size_t)b * 16];
    for (int i = 0; i < ns; i++) {
      float l[2], h[2];
      for (int k = 0; k < 2; k++) {
        l[k] = cnt[b] ? lo[b * 8 + 2 * i + k] : (float)(-p.match_radius);
        h[k] = cnt[b] ? hi[b * 8 + 2 * i + k] : (float)(+p.match_radius);
        const float delta = h[k] - l[k];
        if (delta < 20) {  // widen to at least 20 px (:845-854)
          const float g = ceilf((20 - delta) / 2);
          l[k] -= g;
          h[k] += g;
        }
      }
      r[i] = l[0];
      r[4 + i] = h[0];
      r[8 + i] = l[1];
      r[12 + i] = h[1];
    }
  }
}

// =======================================================================================
// parabolicFitting tail, viso/matcher.cpp:1425-1453; Matrix::operator* and Matrix::solve
// (viso/matrix.cpp) evaluated in the same operation order, in double
// =======================================================================================
static const double kA[9][6] = {{1, 1, 1, -1, -1, 1}, {0, 1, 0, 0, -1, 1}, {1, 1, -1, 1, -1, 1},
                                {1, 0, 0, -1, 0, 1},  {0, 0, 0, 0, 0, 1},  {1, 0, 0, 1, 0, 1},
                                {1, 1, -1, -1, 1, 1}, {0, 1, 0, 0, 1, 1},  {1, 1, 1, 1, 1, 1}};

static bool gauss_jordan6(double A[6][6], double B[6]) {
  int ipiv[6] = {0, 0, 0, 0, 0, 0};
  int icol = 0, irow = 0;
  for (int i = 0; i < 6; i++) {
    double big = 0.0;
    for (int j = 0; j < 6; j++)
      if (ipiv[j] != 1)
        for (int k = 0; k < 6; k++)
          if (ipiv[k] == 0 && fabs(A[j][k]) >= big) {
            big = fabs(A[j][k]);
            irow = j;
            icol = k;
          }
    ++ipiv[icol];
    if (irow != icol) {
      for (int l = 0; l < 6; l++) std::swap(A[irow][l], A[icol][l]);
      std::swap(B[irow], B[icol]);
    }
    if (fabs(A[icol][icol]) < 1e-20) return false;
    const double pivinv = 1.0 / A[icol][icol];
    A[icol][icol] = 1.0;
    for (int l = 0; l < 6; l++) A[icol][l] *= pivinv;
    B[icol] *= pivinv;
    for (int ll = 0; ll < 6; ll++)
      if (ll != icol) {
        const double dum = A[ll][icol];
        A[ll][icol] = 0.0;
        for (int l = 0; l < 6; l++) A[ll][l] -= A[icol][l] * dum;
        B[ll] -= B[icol] * dum;
      }
  }
  return true;
}

bool vsm_host_parabolic_update(const int32_t *c9, int du, int dv, float &u2, float &v2) {
  double b[6], AtA[6][6];
  for (int i = 0; i < 6; i++) {
    double s = 0;
    for (int k = 0; k < 9; k++) s += kA[k][i] * (double)c9[k];
    b[i] = s;
    for (int j = 0; j < 6; j++) {
      double t = 0;
      for (int k = 0; k < 9; k++) t += kA[k][i] * kA[k][j];
      AtA[i][j] = t;
    }
  }
  if (!gauss_jordan6(AtA, b)) return false;
  const float divisor = (float)(b[2] * b[2] - 4.0 * b[0] * b[1]);
  if (fabsf(divisor) < 1e-8 || fabs(b[2]) < 1e-8) return false;
  const float ddv = (float)((2.0 * b[0] * b[4] - b[2] * b[3]) / divisor);
  const float ddu = (float)(-(b[4] + 2.0 * b[1] * ddv) / b[2]);
  if (fabsf(ddu) >= 1.0 || fabsf(ddv) >= 1.0) return false;
  u2 = (float)(u2 + ((float)du - 3.0 + ddu));
  v2 = (float)(v2 + ((float)dv - 3.0 + ddv));
  return true;
}

// =======================================================================================
// B1 bucketFeatures, viso/matcher.cpp:243-284.  std::random_shuffle (libstdc++ stl_algo.h:
// for i in 1..n-1: swap(a[i], a[rand() % (i+1)])) driven by the C library rand(), whose state
// the reference's callers seed with srand(0) (viso/viso.cpp:35).
// =======================================================================================
//
// The shuffles need one rand() per match (about 10k per frame pair) and the C library takes a lock
// for each; LibcRandStream reads the same stream in bulk instead.  glibc's rand() is random()'s
// additive-feedback generator r[i] = r[i-31] + r[i-3] (TYPE_3) over a state array that
// setstate()/initstate() hand out: park the library on a scratch state, step the caller's array
// directly, then hand it back with the advanced read position encoded the way setstate() decodes
// it.  A self-test against the library decides once whether that layout holds; if not, every
// number comes from rand() itself.  Either way the caller's srand()/rand() stream stays coherent.
namespace {
class LibcRandStream {
 public:
  LibcRandStream() {
    static const bool usable = self_test();
    if (!usable) return;
    old_ = setstate(park_);  // park_ holds a valid state since the self-test
    int32_t *arr = (int32_t *)old_;
    if (arr && arr[0] % 5 == 3) {
      tbl_ = arr + 1;
      rear_ = arr[0] / 5;
    }
  }
  ~LibcRandStream() {
    if (!old_) return;
    if (tbl_) ((int32_t *)old_)[0] = 5 * rear_ + 3;
    setstate(old_);
  }
  inline uint32_t next() {
    if (!tbl_) return (uint32_t)rand();
    int front = rear_ + 3;
    if (front >= 31) front -= 31;
    const uint32_t v = (uint32_t)tbl_[front] + (uint32_t)tbl_[rear_];
    tbl_[front] = (int32_t)v;
    if (++rear_ == 31) rear_ = 0;
    return v >> 1;
  }

 private:
  static bool self_test() {
    char *probe = park_;
    char *old = initstate(20240229u, probe, sizeof(park_));
    if (!old) return false;
    int32_t copy[32];
    memcpy(copy, probe, sizeof(copy));
    bool same = copy[0] % 5 == 3 || copy[0] == 3;
    int rear = copy[0] / 5;
    for (int i = 0; i < 200 && same; i++) {
      int front = rear + 3;
      if (front >= 31) front -= 31;
      const uint32_t v = (uint32_t)copy[1 + front] + (uint32_t)copy[1 + rear];
      copy[1 + front] = (int32_t)v;
      if (++rear == 31) rear = 0;
      same = (long)(v >> 1) == random();
    }
    // and the encoded position must round-trip through setstate()
    if (same) {
      char *mine = setstate(old);
      same = mine == probe && ((int32_t *)probe)[0] == 5 * rear + 3;
    } else {
      setstate(old);
    }
    return same;
  }
  alignas(8) static char park_[128];  // where the library's generator waits meanwhile
  char *old_ = nullptr;
  int32_t *tbl_ = nullptr;
  int rear_ = 0;
};
alignas(8) char LibcRandStream::park_[128];
}  // namespace

template <class Rnd>
static void bucket_impl(std::vector<vsm_p_match> &m, int max_features, float bw, float bh, Rnd &rnd) {
  float u_max = 0, v_max = 0;
  for (const vsm_p_match &it : m) {
    if (it.u1c > u_max) u_max = it.u1c;
    if (it.v1c > v_max) v_max = it.v1c;
  }
  const int cols = (int)floorf(u_max / bw) + 1, rows = (int)floorf(v_max / bh) + 1;
  const size_t nb = (size_t)cols * rows, n = m.size();
  // counting sort of match indices by bucket (stable = the reference's push_back order)
  static thread_local std::vector<uint32_t> cell, start, order;
  static thread_local std::vector<vsm_p_match> kept;
  cell.resize(n);
  start.assign(nb + 1, 0);
  order.resize(n);
  for (size_t i = 0; i < n; i++) {
    const float qu = m[i].u1c / bw, qv = m[i].v1c / bh;  // floor == truncation for the usual non-negative quotients
    cell[i] = (uint32_t)((qv >= 0 ? (int)qv : (int)floorf(qv)) * cols + (qu >= 0 ? (int)qu : (int)floorf(qu)));
    start[cell[i] + 1]++;
  }
  for (size_t b = 0; b < nb; b++) start[b + 1] += start[b];
  for (size_t i = 0; i < n; i++) order[start[cell[i]]++] = (uint32_t)i;  // start[b] is now the END of bucket b
  kept.clear();
  constexpr uint32_t kModTable = 512;
  static const uint64_t *mod_magic = [] {
    static uint64_t t[kModTable + 1];
    for (uint32_t d = 1; d <= kModTable; d++) t[d] = UINT64_MAX / d + 1;
    return t;
  }();
  {
    size_t lo = 0;
    for (size_t b = 0; b < nb; b++) {
      const size_t hi = start[b];
      uint32_t *bk = order.data() + lo;
      const size_t sz = hi - lo;
      for (size_t i = 1; i < sz; i++) {
        const uint32_t v = rnd.next(), d = (uint32_t)(i + 1);
        // v % d without a divide for the usual small buckets (Lemire's fastmod, exact for 32-bit operands)
        const size_t j = d <= kModTable ? (size_t)(((__uint128_t)(mod_magic[d] * v) * d) >> 64) : (size_t)(v % d);
        if (i != j) std::swap(bk[i], bk[j]);
      }
      for (size_t k = 0; k < sz && (int)k < max_features; k++) kept.push_back(m[bk[k]]);
      lo = hi;
    }
  }
  m.assign(kept.begin(), kept.end());
}

void vsm_host_bucket(std::vector<vsm_p_match> &m, int max_features, float bw, float bh) {
  LibcRandStream rnd;  // the process's own rand() stream, read in bulk
  bucket_impl(m, max_features, bw, bh, rnd);
}
void vsm_host_bucket_with(std::vector<vsm_p_match> &m, int max_features, float bw, float bh, VsmRandStream &rnd) {
  bucket_impl(m, max_features, bw, bh, rnd);
}

// =======================================================================================
// getGain, viso/matcher.cpp:286-324 (keeps the reference's dims_p clamp for both windows)
// =======================================================================================
static float window_mean(const uint8_t *I, int bpl, int u0, int u1, int v0, int v1) {
  float mean = 0;
  for (int v = v0; v <= v1; v++)
    for (int u = u0; u <= u1; u++) mean += (float)I[v * bpl + u];
  return mean / (float)((u1 - u0 + 1) * (v1 - v0 + 1));
}

float vsm_host_gain(const uint8_t *I1p, const uint8_t *I1c, const int32_t *dims_p, const int32_t *dims_c,
                    const std::vector<vsm_p_match> &m, const int32_t *inliers, int32_t n) {
  if (!I1p || !I1c || m.empty() || n == 0) return 1;
  const int ws = 3;
  float gain = 0;
  int num = 0;
  auto clampi = [](int v, int hi) { return std::min(std::max(v, 0), hi); };
  for (int32_t k = 0; k < n; k++) {
    const int32_t i = inliers[k];
    if (i < (int32_t)m.size()) {
      const vsm_p_match &q = m[i];
      const float mp = window_mean(I1p, dims_p[2], clampi((int)q.u1p - ws, dims_p[0]), clampi((int)q.u1p + ws, dims_p[0]),
                                   clampi((int)q.v1p - ws, dims_p[1]), clampi((int)q.v1p + ws, dims_p[1]));
      const float mc = window_mean(I1c, dims_c[2], clampi((int)q.u1c - ws, dims_p[0]), clampi((int)q.u1c + ws, dims_p[0]),
                                   clampi((int)q.v1c - ws, dims_p[1]), clampi((int)q.v1c + ws, dims_p[1]));
      if (mp > 10) {
        gain += mc / mp;
        num++;
      }
    }
  }
  return num > 0 ? gain / (float)num : 1;
}
