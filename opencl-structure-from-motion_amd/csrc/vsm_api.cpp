// C-ABI of libvisomatch.so (include/visomatch.h): host orchestration of the matcher path.
//
//   vsm_push_back : H2D (or D2D) ingest -> [halve] -> Sobel full / Sobel+blob+corner -> NMS ->
//                   ordered feature emission + descriptors -> bin sort            (all on the GPU)
//   vsm_match     : pass-1 match chain -> export -> host Delaunay support + prior boxes -> H2D ->
//                   pass-2 match chain -> refinement -> export -> host Delaunay support
//   vsm_sequence_run : the same work for a whole sequence with look-ahead: the frames of a chunk
//                   go through every kernel in ONE launch each (grid z/y = image / frame pair) and
//                   the host stages of the chunk's pairs run in parallel on the pool.
//
// A VsmCtx is the device-resident working set for F frame slots and P frame pairs; the streaming
// ring buffer is the (F=2, P=1) instance, sequences use three banks of frame slots and two of pairs (F=6C images, P=2C).
// There is no CPU fallback: without a usable HIP device vsm_create() returns NULL.
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <chrono>
#include <memory>
#include <thread>
#include <unordered_map>
#include <vector>

#include "vsm_host.h"
#include "vsm_dc_gpu.h"
#include "vsm_internal.h"

#define HIPCHK(expr)                                                                              \
  do {                                                                                            \
    hipError_t e_ = (expr);                                                                       \
    if (e_ != hipSuccess) {                                                                       \
      fprintf(stderr, "visomatch: HIP error %s at %s:%d (%s)\n", hipGetErrorString(e_), __FILE__, \
              __LINE__, #expr);                                                                   \
      return VSM_EHIP;                                                                            \
    }                                                                                             \
  } while (0)

static inline int32_t bpl16(int32_t w) { return w + 15 - (w - 1) % 16; }  // viso/matcher.cpp:160
static inline size_t al256(size_t x) { return (x + 255) & ~(size_t)255; }
// CPUs this process may actually use: the cgroup CPU quota (containers), else the hardware
// thread count.  Spinning more threads than the quota only gets the whole process throttled.
static int cpu_budget() {
  unsigned hc = std::thread::hardware_concurrency();
  int budget = hc ? (int)hc : 8;
  if (FILE *f = fopen("/sys/fs/cgroup/cpu.max", "r")) {
    char q[64];
    long period = 0;
    if (fscanf(f, "%63s %ld", q, &period) == 2 && strcmp(q, "max") != 0 && period > 0) {
      long quota = atol(q);
      if (quota > 0) budget = std::min(budget, (int)std::max(1L, quota / period));
    }
    fclose(f);
  }
  return budget;
}

// VSM_DEBUG_TIMING=1: per-chunk phase times of vsm_sequence_run on stderr (read once)
static bool vsm_debug_timing() {
  static const bool on = getenv("VSM_DEBUG_TIMING") != nullptr;
  return on;
}

static inline double now_us() {
  return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

// ---------------------------------------------------------------------------------------
// device working set
// ---------------------------------------------------------------------------------------
struct VsmCtx {
  bool ready = false;
  bool has_heads = false;  // the dense sets carry head records (option match_heads at the time the context was made)
  VsmDims dims{};
  int nframes = 0, npairs = 0;
  uint8_t *arena = nullptr;
  size_t arena_bytes = 0;
  std::vector<VsmImage> h_imgs;  // image id = frame_slot*2 + side
  VsmImage *d_imgs = nullptr;
  std::vector<VsmPair> h_pairs;
  VsmPair *d_pairs = nullptr;
  VsmJob *d_jobs = nullptr;
  int16_t *f1 = nullptr, *f2 = nullptr;  // transient filter responses of one feature launch
  size_t f_stride = 0;
  int32_t cap_set[2] = {0, 0};
  size_t ranges_stride = 0;  // floats per pair
  float *d_ranges = nullptr;
  // pinned host memory.  hm_* is host-mapped: kernels write feature counts and exported match
  // lists straight into it, so results need a stream sync but no D2H copy.
  uint8_t *hm_block = nullptr;
  int32_t *hm_counts = nullptr;  // [nframes*2 images][2 sets]
  int32_t *hm_lcount = nullptr;  // [npairs][2]
  std::vector<vsm_p_match *> hm_list1, hm_list2;  // host views, per pair
  float *h_ranges = nullptr;     // pinned [npairs][ranges_stride]
  VsmJob *h_jobs = nullptr;      // pinned [npairs]
};

// ---------------------------------------------------------------------------------------
// Large device blocks are kept, not handed back.  The driver wipes released VRAM in the background - 23 GB/s on this box,
// 44 ms per GB - on a DMA engine the look-ahead path's own copies (keys, early export, host-fed frames) then queue behind:
// for that long every call of the process takes 6.8-7.0 ms instead of 4.0 (tools/hipfree_probe.py; a context re-created for
// another chunk size gave 3 GB back, a closed handle 4 GB - the "slower mode of whole processes" of rounds 3-4, and the
// reason bench.py's later legs ran 10-30 % below the same legs in a process of their own).  So: a released block of
// 8 MB or more waits in a process-wide cache and the next request of its size class (up to 1.5 x) takes it; beyond
// VSM_DEVICE_POOL_MB (default 24 GB, 0 = hand everything back at once) the oldest blocks do go back to the driver.
// ---------------------------------------------------------------------------------------
namespace {
struct DevPool {
  struct Block {
    void *p;
    size_t bytes;
    int device;
  };
  std::mutex mu;
  std::unordered_map<void *, Block> live;  // blocks handed out (8 MB and more only)
  std::vector<Block> cached;               // oldest first
  size_t cached_bytes = 0;
  static constexpr size_t kMin = (size_t)8 << 20;
  static size_t cap() {
    static const size_t v = (size_t)(getenv("VSM_DEVICE_POOL_MB") ? atoll(getenv("VSM_DEVICE_POOL_MB")) : 24 * 1024) << 20;
    return v;
  }
};
DevPool &dev_pool() {
  static DevPool *p = new DevPool();  // (never destroyed: handles may be closed from static destructors)
  return *p;
}
}  // namespace

hipError_t vsm_dev_alloc(void **out, size_t bytes) {
  *out = nullptr;
  if (bytes < DevPool::kMin || DevPool::cap() == 0) return hipMalloc(out, bytes);
  DevPool &P = dev_pool();
  int dev = 0;
  (void)hipGetDevice(&dev);
  {
    std::lock_guard<std::mutex> lk(P.mu);
    size_t best = P.cached.size();
    for (size_t i = 0; i < P.cached.size(); i++) {
      const DevPool::Block &b = P.cached[i];
      if (b.device == dev && b.bytes >= bytes && b.bytes <= bytes + bytes / 2 && (best == P.cached.size() || b.bytes < P.cached[best].bytes)) best = i;
    }
    if (best < P.cached.size()) {
      const DevPool::Block b = P.cached[best];
      P.cached.erase(P.cached.begin() + (long)best);
      P.cached_bytes -= b.bytes;
      P.live[b.p] = b;
      *out = b.p;
      return hipSuccess;
    }
  }
  hipError_t e = hipMalloc(out, bytes);
  if (e != hipSuccess) {  // (the cache may hold what is missing)
    (void)hipGetLastError();
    std::vector<DevPool::Block> drop;
    {
      std::lock_guard<std::mutex> lk(P.mu);
      drop.swap(P.cached);
      P.cached_bytes = 0;
    }
    for (const DevPool::Block &b : drop) (void)hipFree(b.p);
    e = hipMalloc(out, bytes);
    if (e != hipSuccess) return e;
  }
  std::lock_guard<std::mutex> lk(P.mu);
  P.live[*out] = DevPool::Block{*out, bytes, dev};
  return hipSuccess;
}

void vsm_dev_free(void *p) {
  if (!p) return;
  DevPool &P = dev_pool();
  std::vector<DevPool::Block> drop;
  {
    std::lock_guard<std::mutex> lk(P.mu);
    auto it = P.live.find(p);
    if (it == P.live.end()) {
      drop.push_back(DevPool::Block{p, 0, 0});  // (a small block: straight back)
    } else {
      P.cached.push_back(it->second);
      P.cached_bytes += it->second.bytes;
      P.live.erase(it);
      while (P.cached_bytes > DevPool::cap() && !P.cached.empty()) {
        drop.push_back(P.cached.front());
        P.cached_bytes -= P.cached.front().bytes;
        P.cached.erase(P.cached.begin());
      }
    }
  }
  for (const DevPool::Block &b : drop) (void)hipFree(b.p);
}

void vsm_device_pool_stats(int64_t out[3]) {
  DevPool &P = dev_pool();
  std::lock_guard<std::mutex> lk(P.mu);
  out[0] = (int64_t)P.cached.size();
  out[1] = (int64_t)P.cached_bytes;
  out[2] = (int64_t)P.live.size();
}

void vsm_device_pool_trim(void) {
  DevPool &P = dev_pool();
  std::vector<DevPool::Block> drop;
  {
    std::lock_guard<std::mutex> lk(P.mu);
    drop.swap(P.cached);
    P.cached_bytes = 0;
  }
  for (const DevPool::Block &b : drop) (void)hipFree(b.p);
}

static void ctx_destroy(VsmCtx &c) {
  if (c.arena) vsm_dev_free(c.arena);
  if (c.hm_block) (void)hipHostFree(c.hm_block);
  if (c.h_ranges) (void)hipHostFree(c.h_ranges);
  if (c.h_jobs) (void)hipHostFree(c.h_jobs);
  c = VsmCtx();
}

static int nms_cells(int32_t len, int32_t n) {  // loop count of "for (i=n+margin; i<len-n-margin; i+=n+1)"
  int32_t span = len - 2 * n - 2 * VSM_MARGIN;
  return span > 0 ? (span + n) / (n + 1) : 0;
}

// One arena per context, zero-filled once, so row padding stays 0.
static int ctx_create(VsmCtx &c, const vsm_params &p, int32_t w, int32_t hh, int nframes, int npairs, hipStream_t stream, bool heads) {
  ctx_destroy(c);
  c.has_heads = heads;
  VsmDims &d = c.dims;
  d.w = w;
  d.h = hh;
  d.bpl = bpl16(w);
  d.scale = p.half_resolution ? 2 : 1;
  d.mw = p.half_resolution ? w / 2 : w;
  d.mh = p.half_resolution ? hh / 2 : hh;
  d.mbpl = p.half_resolution ? bpl16(d.mw) : d.bpl;
  if (d.mw <= 0 || d.mh <= 0 || d.w >= 16384 || d.h >= 16384) return VSM_EDIMS;  // 14-bit coordinates
  d.ub = (int32_t)ceilf((float)w / (float)p.match_binsize);
  d.vb = (int32_t)ceilf((float)hh / (float)p.match_binsize);
  const int nb = 4 * d.ub * d.vb;
  if (p.match_binsize < 1 || p.match_binsize > 32768 || nb > (1 << 22)) {
    fprintf(stderr, "visomatch: match_binsize %d is not usable (%d bins)\n", p.match_binsize, nb);
    return VSM_EARG;
  }
  if (p.nms_n < 1 || p.nms_n > 31) {
    fprintf(stderr, "visomatch: nms_n %d outside the supported range 1..31\n", p.nms_n);
    return VSM_EARG;
  }
  int32_t ns = p.nms_n * 3;  // viso/matcher.cpp:685-687
  if (ns > 10) ns = p.nms_n > 10 ? p.nms_n : 10;
  const int32_t nn[2] = {ns, p.nms_n};
  int32_t ncu[2], ncv[2];
  for (int k = 0; k < 2; k++) {
    ncu[k] = nms_cells(d.mw, nn[k]);
    ncv[k] = nms_cells(d.mh, nn[k]);
    c.cap_set[k] = 4 * ncu[k] * ncv[k];
    if (c.cap_set[k] < 4) c.cap_set[k] = 4;
  }
  c.nframes = nframes;
  c.npairs = npairs;
  const int nimg = nframes * 2;
  const size_t full = al256((size_t)d.bpl * d.h + 64), mres = al256((size_t)d.mbpl * d.mh + 64);
  size_t off = 0;
  auto take = [&](size_t bytes) {
    size_t o = off;
    off += al256(bytes);
    return o;
  };
  struct SetOff {
    size_t feat, count, cand, cell_off, bin_start, bin_cnt, binid, s_idx, s_uv, s_desc, s_rank, tmp, heads;
  };
  struct ImgOff {
    size_t img, imgm, du, dv, duvt;
    SetOff set[2];
  };
  std::vector<ImgOff> io(nimg);
  for (int i = 0; i < nimg; i++) {
    io[i].img = take(full);
    io[i].imgm = p.half_resolution ? take(mres) : io[i].img;
    io[i].du = take(mres);
    io[i].dv = take(mres);
    io[i].duvt = p.half_resolution ? take((size_t)d.bpl * ((d.h + 7) & ~7) * 2 + 256) : 0;  // (+ one tile row of slack: windows load the tile to their right)
    for (int k = 0; k < 2; k++) {
      const size_t cap = c.cap_set[k];
      io[i].set[k].feat = take(cap * 48);
      io[i].set[k].count = take(4);
      io[i].set[k].cand = take((size_t)(ncu[k] * ncv[k] + 1) * 16);
      io[i].set[k].cell_off = take((size_t)(ncu[k] * ncv[k] + 2) * 4);
      io[i].set[k].bin_start = take((size_t)(nb * VSM_VSUB + 1) * 4);
      io[i].set[k].bin_cnt = take((size_t)(nb * VSM_VSUB + 1) * 4);
      io[i].set[k].s_rank = take(cap * 4);
      io[i].set[k].binid = take(cap * 4);
      io[i].set[k].s_idx = take(cap * 4);
      io[i].set[k].s_uv = take(cap * 8);
      io[i].set[k].s_desc = take(cap * 32);
      io[i].set[k].tmp = take(cap * 4);
      io[i].set[k].heads = k == 1 && heads ? take((size_t)nb * VSM_VSUB * 64) : 0;
    }
  }
  c.f_stride = mres;  // int16 elements per image plane
  const size_t o_f1 = take(c.f_stride * 2 * nimg), o_f2 = take(c.f_stride * 2 * nimg);
  const size_t qcap = (size_t)(c.cap_set[0] > c.cap_set[1] ? c.cap_set[0] : c.cap_set[1]);
  c.ranges_stride = (size_t)d.ub * d.vb * 16;
  struct PairOff {
    size_t raw, flag, bc, l1, l2, cnt, pf;
  };
  std::vector<PairOff> po(npairs);
  for (int j = 0; j < npairs; j++) {
    po[j].raw = take(qcap * 48);
    po[j].flag = take(qcap * 4);
    po[j].bc = take((qcap / 256 + 2) * 4);
    po[j].l1 = take((size_t)c.cap_set[0] * 48);
    po[j].l2 = take(qcap * 48);
    po[j].cnt = take(16);
    po[j].pf = p.refinement == 2 ? take(qcap * 3 * 12 * 4) : 0;  // (per-frame ring: one pair; look-ahead banks: every pair)
  }
  const size_t o_rng = take(c.ranges_stride * 4 * npairs);
  const size_t o_imgs = take((size_t)nimg * sizeof(VsmImage));
  const size_t o_pairs = take((size_t)npairs * sizeof(VsmPair));
  const size_t o_jobs = take((size_t)npairs * sizeof(VsmJob));
  c.arena_bytes = off;
  HIPCHK(vsm_dev_alloc((void **)&c.arena, c.arena_bytes));
  HIPCHK(hipMemsetAsync(c.arena, 0, c.arena_bytes, stream));
  // host-mapped result block: [image counts][list counts][list1, list2 per pair]
  const size_t l1 = al256((size_t)c.cap_set[0] * 48), l2 = al256(qcap * 48);
  const size_t hc = al256((size_t)nimg * 2 * 4), hl = al256((size_t)npairs * 2 * 4);
  const size_t hm_bytes = hc + hl + (l1 + l2) * npairs;
  HIPCHK(hipHostMalloc((void **)&c.hm_block, hm_bytes, hipHostMallocMapped));
  memset(c.hm_block, 0, hc + hl);
  uint8_t *dblock = nullptr;
  HIPCHK(hipHostGetDevicePointer((void **)&dblock, c.hm_block, 0));
  c.hm_counts = (int32_t *)c.hm_block;
  c.hm_lcount = (int32_t *)(c.hm_block + hc);
  HIPCHK(hipHostMalloc((void **)&c.h_ranges, c.ranges_stride * 4 * npairs, hipHostMallocDefault));
  HIPCHK(hipHostMalloc((void **)&c.h_jobs, sizeof(VsmJob) * npairs, hipHostMallocDefault));
  uint8_t *b = c.arena;
  c.h_imgs.resize(nimg);
  for (int i = 0; i < nimg; i++) {
    VsmImage &im = c.h_imgs[i];
    im.img = b + io[i].img;
    im.imgm = b + io[i].imgm;
    im.du = b + io[i].du;
    im.dv = b + io[i].dv;
    im.du_full = p.half_resolution ? nullptr : im.du;
    im.dv_full = p.half_resolution ? nullptr : im.dv;
    im.duv_tiled = p.half_resolution ? b + io[i].duvt : nullptr;
    for (int k = 0; k < 2; k++) {
      VsmSet &s = im.set[k];
      s.feat = (int32_t *)(b + io[i].set[k].feat);
      s.count = (int32_t *)(b + io[i].set[k].count);
      s.count_host = (int32_t *)dblock + (i * 2 + k);
      s.cand = (int32_t *)(b + io[i].set[k].cand);
      s.cell_off = (int32_t *)(b + io[i].set[k].cell_off);
      s.bin_start = (int32_t *)(b + io[i].set[k].bin_start);
      s.bin_cnt = (int32_t *)(b + io[i].set[k].bin_cnt);
      s.binid = (int32_t *)(b + io[i].set[k].binid);
      s.s_rank = (int32_t *)(b + io[i].set[k].s_rank);
      s.s_idx = (int32_t *)(b + io[i].set[k].s_idx);
      s.s_uv = (uint32_t *)(b + io[i].set[k].s_uv);
      s.s_desc = (uint4 *)(b + io[i].set[k].s_desc);
      s.tmp = (int32_t *)(b + io[i].set[k].tmp);
      s.heads = k == 1 && heads ? (uint4 *)(b + io[i].set[k].heads) : nullptr;
      s.cap = c.cap_set[k];
      s.nms_n = nn[k];
      s.ncu = ncu[k];
      s.ncv = ncv[k];
    }
  }
  c.f1 = (int16_t *)(b + o_f1);
  c.f2 = (int16_t *)(b + o_f2);
  c.d_ranges = (float *)(b + o_rng);
  c.h_pairs.resize(npairs);
  c.hm_list1.resize(npairs);
  c.hm_list2.resize(npairs);
  for (int j = 0; j < npairs; j++) {
    VsmPair &pr = c.h_pairs[j];
    pr.raw = (vsm_p_match *)(b + po[j].raw);
    pr.flag = (int32_t *)(b + po[j].flag);
    pr.blockcnt = (int32_t *)(b + po[j].bc);
    pr.list1 = (vsm_p_match *)(b + po[j].l1);
    pr.list2 = (vsm_p_match *)(b + po[j].l2);
    pr.count = (int32_t *)(b + po[j].cnt);
    pr.ranges = c.d_ranges + (size_t)j * c.ranges_stride;
    pr.pf = po[j].pf ? (int32_t *)(b + po[j].pf) : nullptr;
    const size_t lo = hc + hl + (l1 + l2) * j;
    c.hm_list1[j] = (vsm_p_match *)(c.hm_block + lo);
    c.hm_list2[j] = (vsm_p_match *)(c.hm_block + lo + l1);
    pr.hlist1 = (vsm_p_match *)(dblock + lo);
    pr.hlist2 = (vsm_p_match *)(dblock + lo + l1);
    pr.hcount = (int32_t *)(dblock + hc) + 2 * j;
  }
  c.d_imgs = (VsmImage *)(b + o_imgs);
  c.d_pairs = (VsmPair *)(b + o_pairs);
  c.d_jobs = (VsmJob *)(b + o_jobs);
  {
    // (the tables cross by the copy kernel out of a page-locked twin, like every other host-to-device transfer of the path: a
    // runtime copy of more than 16 KB out of pageable memory that finds the DMA engine busy goes through a shader copy on a
    // hardware queue the runtime creates for it and keeps - DESIGN.md section 6)
    const size_t bi = (size_t)nimg * sizeof(VsmImage), bp = (size_t)npairs * sizeof(VsmPair);
    uint8_t *twin = nullptr;
    HIPCHK(hipHostMalloc((void **)&twin, al256(bi) + al256(bp), hipHostMallocDefault));
    memcpy(twin, c.h_imgs.data(), bi);
    memcpy(twin + al256(bi), c.h_pairs.data(), bp);
    hipError_t e = vsm_upload(stream, c.d_imgs, twin, bi);
    if (e == hipSuccess) e = vsm_upload(stream, c.d_pairs, twin + al256(bi), bp);
    if (e == hipSuccess) e = hipStreamSynchronize(stream);
    (void)hipHostFree(twin);
    HIPCHK(e);
  }
  c.ready = true;
  return VSM_OK;
}

// Look-ahead final stage, shared between host and GPU (vsm_dc.hip): per pair bank the pinned / device
// slabs that carry the prepared keys and task lists to the GPU and the triangle records back.
struct DcBank {
  int npairs = 0, stride_pts = 0, stride_tasks = 0;
  uint64_t *d_key = nullptr, *h_key = nullptr, *d_key_sorted = nullptr;  // h_key: (x,y) order if the GPU orders them, else kd order
  uint32_t *d_kd = nullptr;  // scratch of k_dc_kd_order
  uint64_t *d_tie_keys = nullptr;  // per pair [stride_pts]: keys of the pass-2 list in list order (k_dc_tie_keys)
  int32_t *d_tie_n = nullptr;      // per pair: list length
  float *d_flow = nullptr, *h_flow = nullptr;          // per pair [3][stride_pts]: flow u, flow v, disparity of every match
  int32_t *d_support = nullptr, *h_support = nullptr;  // per pair [stride_pts]: support count of every match
  uint32_t *d_pt = nullptr, *h_pt = nullptr;
  int32_t *d_id = nullptr, *h_id = nullptr, *d_tri = nullptr, *h_tri = nullptr;
  uint32_t *d_trip = nullptr, *h_trip = nullptr;  // packed triangle records, [2 * stride_pts][3] per pair (VsmDcJob::tri_packed)
  VsmDcTask *d_tasks = nullptr, *h_tasks = nullptr;
  VsmDcMerge *d_merges = nullptr, *h_merges = nullptr;  // stride_tasks per pair (a binary tree has fewer internal nodes than leaves)
  VsmDcHull *d_hulls = nullptr, *h_hulls = nullptr;    // by node number: 2 * stride_tasks per pair
  VsmDcJob *d_jobs = nullptr, *h_jobs = nullptr;
  std::vector<int32_t> m, nt, nn;  // per pair: distinct points, tasks (nt < 0: the host solves the sub-trees), tree nodes
  void release() {
    vsm_dev_free(d_key);
    vsm_dev_free(d_key_sorted);
    vsm_dev_free(d_kd);
    vsm_dev_free(d_tie_keys);
    vsm_dev_free(d_tie_n);
    vsm_dev_free(d_flow);
    vsm_dev_free(d_support);
    (void)hipHostFree(h_flow);
    (void)hipHostFree(h_support);
    vsm_dev_free(d_pt);
    vsm_dev_free(d_id);
    vsm_dev_free(d_tri);
    vsm_dev_free(d_trip);
    (void)hipHostFree(h_trip);
    vsm_dev_free(d_tasks);
    vsm_dev_free(d_merges);
    vsm_dev_free(d_hulls);
    vsm_dev_free(d_jobs);
    (void)hipHostFree(h_key);
    (void)hipHostFree(h_pt);
    (void)hipHostFree(h_id);
    (void)hipHostFree(h_tri);
    (void)hipHostFree(h_tasks);
    (void)hipHostFree(h_merges);
    (void)hipHostFree(h_hulls);
    (void)hipHostFree(h_jobs);
    *this = DcBank();
  }
  bool reserve(int pairs, int pts, int tasks) {
    pts = (pts + 1) & ~1;  // (the long-list y order views two adjacent scratch arrays as 64-bit items)
    if (pairs <= npairs && pts <= stride_pts && tasks <= stride_tasks) return true;
    release();
    npairs = pairs;
    stride_pts = pts;
    stride_tasks = tasks;
    const size_t P = (size_t)pairs * pts, T = (size_t)pairs * tasks;
    bool ok = vsm_dev_alloc((void **)&d_key, P * 8) == hipSuccess && vsm_dev_alloc((void **)&d_key_sorted, P * 8) == hipSuccess &&
              vsm_dev_alloc((void **)&d_kd, P * 4 * VSM_DC_KD_SCRATCH) == hipSuccess && vsm_dev_alloc((void **)&d_pt, P * 4) == hipSuccess &&
              vsm_dev_alloc((void **)&d_flow, P * 12) == hipSuccess && vsm_dev_alloc((void **)&d_support, P * 4) == hipSuccess &&
              vsm_dev_alloc((void **)&d_tie_keys, P * 8) == hipSuccess && vsm_dev_alloc((void **)&d_tie_n, (size_t)pairs * 4) == hipSuccess &&
              hipHostMalloc((void **)&h_flow, P * 12, hipHostMallocDefault) == hipSuccess &&
              hipHostMalloc((void **)&h_support, P * 4, hipHostMallocDefault) == hipSuccess &&
              vsm_dev_alloc((void **)&d_id, P * 4) == hipSuccess && vsm_dev_alloc((void **)&d_tri, P * 64) == hipSuccess &&
              vsm_dev_alloc((void **)&d_trip, P * 24) == hipSuccess && hipHostMalloc((void **)&h_trip, P * 24, hipHostMallocDefault) == hipSuccess &&
              vsm_dev_alloc((void **)&d_tasks, T * sizeof(VsmDcTask)) == hipSuccess &&
              vsm_dev_alloc((void **)&d_merges, T * sizeof(VsmDcMerge)) == hipSuccess &&
              vsm_dev_alloc((void **)&d_hulls, 2 * T * sizeof(VsmDcHull)) == hipSuccess &&
              vsm_dev_alloc((void **)&d_jobs, pairs * sizeof(VsmDcJob)) == hipSuccess &&
              hipHostMalloc((void **)&h_key, P * 8, hipHostMallocDefault) == hipSuccess &&
              hipHostMalloc((void **)&h_pt, P * 4, hipHostMallocDefault) == hipSuccess &&
              hipHostMalloc((void **)&h_id, P * 4, hipHostMallocDefault) == hipSuccess &&
              hipHostMalloc((void **)&h_tri, P * 64, hipHostMallocDefault) == hipSuccess &&
              hipHostMalloc((void **)&h_tasks, T * sizeof(VsmDcTask), hipHostMallocDefault) == hipSuccess &&
              hipHostMalloc((void **)&h_merges, T * sizeof(VsmDcMerge), hipHostMallocDefault) == hipSuccess &&
              hipHostMalloc((void **)&h_hulls, 2 * T * sizeof(VsmDcHull), hipHostMallocDefault) == hipSuccess &&
              hipHostMalloc((void **)&h_jobs, pairs * sizeof(VsmDcJob), hipHostMallocDefault) == hipSuccess;
    m.assign(pairs, 0);
    nt.assign(pairs, 0);
    nn.assign(pairs, 0);
    if (!ok) release();
    return ok;
  }
};

// ---------------------------------------------------------------------------------------
// Measurement / test switches of a handle: read from the environment ONCE, at vsm_create (the calls themselves never
// look at the environment), and settable afterwards through vsm_set_option (bench.py's "alone" pass, tools/).
struct VsmSwitches {
  int seq_v2 = 1;          // VSM_SEQ_V2: 0 = the host-shared look-ahead form
  int seq_chunk = 0;       // VSM_SEQ_CHUNK: frames per look-ahead chunk (0 = by host threads)
  int seq_dc_streams = 3;  // VSM_SEQ_DC_STREAMS: side streams of the GPU-resident form (1..3; with the main and the null stream: five hardware queues)
  int seq_serial = 0;      // VSM_SEQ_SERIAL: nothing overlaps (every kernel's time alone)
  int seq_gpu_sorts = -1;  // VSM_SEQ_GPU_SORTS: percent of a chunk's vertex sorts done on the device (-1 = by host threads)
  int front = 1;           // option "front" (not read from the environment): the fused front end
  int seq_early_export = -1;  // VSM_SEQ_EARLY_EXPORT: the refined lists cross PCIe beside the triangulation, survivor bits follow, the pool
                              // closes the gaps (1); survivors compacted on the device, one DMA copy at the chain's end (0); -1: by pool size
  // settable through vsm_set_option only (tests, tools); none of them is read from the environment:
  int dc_gpu = -1;           // host-shared look-ahead form: the GPU's share of the final stage: -1 = chunks with at least as many pairs as host threads, 0 = never, 1 = always
  int dc_full = -1;          // ... everything after the vertex sort on the GPU: -1 = with six host threads or fewer, 0 / 1
  int dc_fault_inject = 0;   // ... tests: the completion callback of the GPU share is "lost" (dc_wait()'s watchdog has to notice)
  int dc_watchdog_ms = 20000;  // ... how long dc_wait() listens for that callback before it asks the stream itself
  int seq_keys_dma = -1;     // GPU-resident form: the keys reach the host's vertex sort by a DMA copy - 2: on the fifth stream behind an event, so that the chunk's mesh does not stand behind 6-8 MB crossing PCIe (ranks with ten pool threads and more, where the last mesh bounds the call: median step 3.90 -> 3.83 ms); 1: on the chain's own stream (fewer threads: the pool bounds the call and wants its keys first - 4.88 against 5.01 ms with eight threads, 6.71 against 7.14 with four) - or by the key kernel's own stores into host-mapped memory (0); -1: by pool size
  int seq_ties1_null = 1;    // ... the pass-1 chain's vertex sort (one wave per list) on the null stream (1) or on side stream cs[k + 2] (0)
  int seq_keys_pieces = 0;   // ... in this many pieces, an event behind each (the pool starts on the first lists while the others cross); 0: four with ten pool threads and more, else one
  int seq_warm_gaps = 1;     // ... the pool reads the last chunk's exported lists into its L3 caches while it waits for their survivor bits, and a list's gaps are closed by a thread of the domain that read it
  int seq_ties1_host = -1;   // ... the pass-1 lists' vertex sorts on the pool while the device triangulates (1) or on the device, one wave per list (0); -1: by pool size
  int seq_block_after_p2 = 0;  // ... a chunk's block kernel waits for the next chunk's second matching pass
  int seq_last_first = 1;    // ... a chain's sort + kd order kernel goes in with its head, and the block kernel of the last chunk but one waits for the last chunk's
  int seq_first_chunk = 0;   // ... frames of the call's first chunk (0: like the others)
  int seq_p2_first = -1;     // ... a chunk's second pass in front of the features of chunk k + 2 (the order host-resident inputs get): -1 = by pool size
  int seq_export_budget = 2; // ... pieces of the early export submitted behind a chunk's keys where the next chunk's keys follow at once (sequence_run_v2: export_some)
  int seq_null_stream = 1;   // ... its fifth stream (early exports, the device's vertex sorts) is the process's null stream (1) or a non-blocking stream of
                             // the library's own (0: for applications that keep work of their own on the null stream - INTEGRATION.md)
  int seq_host_inorder = 1;  // ... host-resident inputs: chunk by chunk in the order of arrival - the caller's thread waits for a chunk's feature counts only when everything of the chunk in front is enqueued (0: the run-ahead order of HBM-resident inputs)
  int seq_defer_refine = 0;  // ... a chunk's refinement behind the NEXT chunk's second-pass matching where that follows at once (the chain's keys do not need it)
  int seq_host_pinned = 0;   // ... host-resident input images are in page-locked memory (the caller's promise): DMA straight out of them, no gather pass
  int match_heads = 0;       // the second matching pass takes a bin's start and its first 15 candidates' coordinates from one 64-byte head record (k_feat_heads) instead of bin starts + coordinate runs: measured slower, off; read when a context is made
  int fused_features = 1;    // filters + suppression of the matching resolution out of one LDS tile (k_feat_dense / k_feat_sparse; default radii) or the separate kernels (0)
  int feat_order = 1;        // feature records + bin-sorted copy by k_feat_scan / k_feat_order (tiles of whole search bins) or by k_scan_cells / k_emit / k_bin_* (0)
  int multi_host_pass1 = -1; // vsm_multi_process: the first-pass lists' removeOutliers + prior boxes on the host pool (1) or by the device chain (0); -1: host for K <= pool threads
  int multi_shared_ego = 0;  // vsm_multi_process: idle pool threads take RANSAC hypotheses of the sequences' egomotion (1: measured, no gain), every sequence on one thread (0)
  int frame_early_xy = 1;    // per-frame path: the pass-2 list's pixels cross in front of the list, the host triangulates while the refinement and the export run (vsm_match)
  int filter_planes = 0;     // vsm_push_back keeps f1 / f2 in HBM for vsm_get_filter_responses (the fused tiles write them on the side)
  static int env_int(const char *name, int dflt) {
    const char *e = getenv(name);
    return e ? atoi(e) : dflt;
  }
  void from_environment() {
    seq_v2 = env_int("VSM_SEQ_V2", 1) != 0;
    seq_chunk = std::max(0, env_int("VSM_SEQ_CHUNK", 0));
    seq_dc_streams = std::min(3, std::max(1, env_int("VSM_SEQ_DC_STREAMS", 3)));
    seq_serial = env_int("VSM_SEQ_SERIAL", 0) != 0;
    seq_gpu_sorts = env_int("VSM_SEQ_GPU_SORTS", -1);
    seq_early_export = env_int("VSM_SEQ_EARLY_EXPORT", -1);
    multi_host_pass1 = env_int("VSM_MULTI_HOST_PASS1", -1);    // (measurements: vsm_multi_create takes no options)
    multi_shared_ego = env_int("VSM_MULTI_SHARED_EGO", 0);
  }
  bool set(const char *name, int v) {
    if (!strcmp(name, "seq_v2")) seq_v2 = v != 0;
    else if (!strcmp(name, "seq_chunk")) seq_chunk = std::max(0, v);
    else if (!strcmp(name, "seq_dc_streams")) seq_dc_streams = std::min(3, std::max(1, v));
    else if (!strcmp(name, "seq_serial")) seq_serial = v != 0;
    else if (!strcmp(name, "seq_gpu_sorts")) seq_gpu_sorts = v;
    else if (!strcmp(name, "front")) front = v != 0;
    else if (!strcmp(name, "seq_early_export")) seq_early_export = v;
    else if (!strcmp(name, "dc_gpu")) dc_gpu = v;
    else if (!strcmp(name, "dc_full")) dc_full = v;
    else if (!strcmp(name, "dc_fault_inject")) dc_fault_inject = v;
    else if (!strcmp(name, "dc_watchdog_ms")) dc_watchdog_ms = std::max(1, v);
    else if (!strcmp(name, "seq_keys_dma")) seq_keys_dma = v;
    else if (!strcmp(name, "seq_export_budget")) seq_export_budget = v;
    else if (!strcmp(name, "seq_first_chunk")) seq_first_chunk = std::max(0, v);
    else if (!strcmp(name, "seq_p2_first")) seq_p2_first = v;
    else if (!strcmp(name, "seq_last_first")) seq_last_first = v != 0;
    else if (!strcmp(name, "seq_keys_pieces")) seq_keys_pieces = v;
    else if (!strcmp(name, "seq_block_after_p2")) seq_block_after_p2 = v != 0;
    else if (!strcmp(name, "seq_ties1_host")) seq_ties1_host = v;
    else if (!strcmp(name, "seq_warm_gaps")) seq_warm_gaps = v != 0;
    else if (!strcmp(name, "seq_ties1_null")) seq_ties1_null = v != 0;
    else if (!strcmp(name, "seq_null_stream")) seq_null_stream = v != 0;
    else if (!strcmp(name, "seq_host_pinned")) seq_host_pinned = v != 0;
    else if (!strcmp(name, "seq_defer_refine")) seq_defer_refine = v != 0;
    else if (!strcmp(name, "seq_host_inorder")) seq_host_inorder = v != 0;
    else if (!strcmp(name, "match_heads")) match_heads = v != 0;
    else if (!strcmp(name, "fused_features")) fused_features = v != 0;
    else if (!strcmp(name, "filter_planes")) filter_planes = v != 0;
    else if (!strcmp(name, "feat_order")) feat_order = v != 0;
    else if (!strcmp(name, "frame_early_xy")) frame_early_xy = v != 0;
    else if (!strcmp(name, "multi_host_pass1")) multi_host_pass1 = v;
    else if (!strcmp(name, "multi_shared_ego")) multi_shared_ego = v;
    else return false;
    return true;
  }
};

struct vsm_handle {
  vsm_params param;  // match_radius already halved for half_resolution (viso/matcher.cpp:59-60)
  void *aff = nullptr;  // CPU record of the handle's device (vsm_host.h): where the threads created for this handle confine themselves
  VsmSwitches sw;
  int device = 0;
  hipStream_t stream = nullptr;
  hipEvent_t input_ev = nullptr;   // vsm_wait_for_stream(): the producer stream's marker
  hipEvent_t idle_wait = nullptr;  // blocking-sync event (pass 2 of a chunk done): the look-ahead caller sleeps on it, its CPU goes to the host pool
  hipEvent_t seq_ev[2] = {nullptr, nullptr};  // look-ahead markers: features done / pass 1 done (blocking sync too)
  static constexpr int kDcBanks = 4;  // chunks whose final stage may be in flight at once
  struct DcBank *dc_bank[kDcBanks] = {nullptr, nullptr, nullptr, nullptr};  // look-ahead: GPU share of the exact Delaunay
  hipStream_t tie_stream[2] = {nullptr, nullptr};  // the emulated vertex sorts of a chunk's pairs (k_dc_ties_of_lists), alternating
  hipEvent_t tie_ev = nullptr;                     // pass-2 lists compacted
  hipEvent_t tie_copied[2] = {nullptr, nullptr};   // per pair bank: its lists' keys have been copied out
  bool tie_copied_set[2] = {false, false};
  int32_t *hm_ties = nullptr, *d_ties = nullptr;   // host-mapped verdicts [kDcBanks][chunk][VSM_DC_TIE_OUT_INTS]
  int ties_chunk = 0;
  hipStream_t dc_stream[2] = {nullptr, nullptr};  // alternate per chunk: one chunk's records travel while the next one's kernels run
  std::vector<VsmHostWork> seq_work;               // per pair of every Delaunay bank: state between the two host halves
  VsmCtx ring;  // streaming ring buffer: 2 frame slots, 1 pair
  VsmCtx seq;   // look-ahead sequences: 3 banks of C frame slots, 2 banks of C pairs
  int seq_chunk = 0;

  // ring buffer state (Matcher's prev/curr pointers, viso/matcher.cpp:108-155)
  int cur = 0;
  bool have[2] = {false, false};   // frame slot holds a left image
  bool right[2] = {false, false};  // ... and a right image
  int32_t n_feat[2][2][2] = {};    // [slot][side][set]
  int32_t dims_p[3] = {0, 0, 0}, dims_c[3] = {0, 0, 0};
  bool f_valid = false;            // f1/f2 hold the responses of the current left image
  bool counts_pending = false;     // an asynchronous push is in flight
  int pending_slot = 0, pending_imgs = 0;

  // results of the streaming API
  std::vector<vsm_p_match> stage[5];
  std::vector<vsm_p_match> matched;
  std::vector<float> ranges;
  std::vector<int32_t> pf;
  int capture_stage2 = 0;
  bool stage3_in_hm = false;     // stage 3 is the ring's host-mapped pass-2 list as exported (vsm_stage_get)
  uint32_t *xy_host = nullptr, *xy_dev = nullptr;  // the pass-2 list's pixels, x | y << 16, host-mapped (vsm_match: early_xy)
  size_t xy_cap = 0;
  hipEvent_t xy_ev = nullptr, done_ev = nullptr, push_ev = nullptr;
  bool push_ev_set = false;      // push_ev is recorded behind the pending push (settle)
  VsmHostWork work;
  int64_t counters[5] = {0, 0, 0, 0, 0};
  double timings[5] = {0, 0, 0, 0, 0};
  std::vector<uint8_t> gainI[2];
  // results of the sequence API
  std::vector<std::vector<vsm_p_match>> seq_matches;
  // ... or, per frame, a view of 48-byte records that the device wrote straight into the host-mapped result arena (the
  // GPU-resident look-ahead form on hosts with few threads: no host copy inside the call at all)
  struct SeqView {
    const vsm_p_match *p = nullptr;
    int32_t n = 0;
  };
  std::vector<SeqView> seq_view;
  double seq_timings[4] = {0, 0, 0, 0};
  bool dc_gpu_broken = false;    // a Delaunay stream reported a HIP error once: the host-shared form keeps off the GPU share from then on
  std::atomic<int> seq_hip_error{0};  // set by a chunk whose GPU share failed during the current vsm_sequence_run
  struct Seq2 *seq2 = nullptr;   // GPU-resident look-ahead path (vsm_seq2.inc): streams, slabs, result arena
  int32_t seq_v2_frames = 0;     // > 0: the last sequence's results are in seq2's arena, not in seq_matches

  uint8_t *stage_host = nullptr;  // pinned staging for host images
  size_t stage_bytes = 0;
  // look-ahead calls fed from host memory: two pinned slots a chunk's images are gathered into (by the pool, in parallel)
  // and their device twins; an event per slot says when its upload has been consumed
  uint8_t *seq_stage_h[2] = {nullptr, nullptr}, *seq_stage_d[2] = {nullptr, nullptr};
  size_t seq_stage_bytes = 0;
  hipEvent_t seq_stage_ev[2] = {nullptr, nullptr};
  int seq_stage_next = 0;
  VsmProf prof;
  VsmPool *pool = nullptr;
  VsmForkJoin *fj = nullptr;
};

// Matcher::range records (u_min[4], u_max[4], v_min[4], v_max[4], viso/matcher.h:152-157) go to the
// device stage-major -- {u_min, u_max, v_min, v_max} of stage 0, then stage 1, ... -- so that a chain
// fetches the box of one stage with a single 16-byte load
static void ranges_to_device_layout(float *dst, const float *src, size_t n_floats) {
  for (size_t b = 0; b + 16 <= n_floats; b += 16)
    for (int stage = 0; stage < 4; stage++)
      for (int k = 0; k < 4; k++) dst[b + stage * 4 + k] = src[b + k * 4 + stage];
}

VsmPool *vsm_pool_of(vsm_handle *h) { return h->pool; }
VsmForkJoin *vsm_forkjoin_of(vsm_handle *h) { return h->fj; }
double vsm_now_us() { return now_us(); }
static void seq2_destroy(vsm_handle *h);  // vsm_seq2.inc

extern "C" {

const char *vsm_version(void) { return "visomatch 0.3 (gfx950)"; }

void vsm_default_params(vsm_params *p) {
  memset(p, 0, sizeof(*p));
  p->nms_n = 3;
  p->nms_tau = 50;
  p->match_binsize = 50;
  p->match_radius = 200;
  p->match_disp_tolerance = 2;
  p->outlier_disp_tolerance = 5;
  p->outlier_flow_tolerance = 5;
  p->multi_stage = 1;
  p->half_resolution = 1;
  p->refinement = 1;
}

vsm_handle *vsm_create(const vsm_params *p) {
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) {
    fprintf(stderr, "visomatch: no HIP device available (this library has no CPU path)\n");
    return nullptr;
  }
  vsm_handle *h = new vsm_handle();
  h->param = *p;
  if (p->half_resolution) h->param.match_radius /= 2;
  h->sw.from_environment();
  {
    int dev = 0;
    char bdf[64] = {0};
    if (hipGetDevice(&dev) == hipSuccess && hipDeviceGetPCIBusId(bdf, (int)sizeof(bdf), dev) == hipSuccess)
      vsm_affinity_from_device(bdf);  // (before the pools start their threads)
    h->aff = vsm_affinity_current();
    vsm_forkjoin_domain_hint(dev);  // (the fork-join workers' L3 domain: a domain per device of the socket)
  }
  {
    // host threads (the caller's thread included): VSM_HOST_THREADS frame-parallel workers for
    // the look-ahead API, at most 8 of them for the sub-problems of one triangulation (streaming)
    int nt = 16;
    if (const char *e = getenv("VSM_HOST_THREADS")) nt = atoi(e);
    nt = std::max(1, std::min(nt, cpu_budget()));
    // (the look-ahead caller sleeps in a blocking event wait while the GPU works, so all nt budgeted
    // CPUs go to pool workers: nt workers + the caller's thread)
    h->pool = new VsmPool(nt + 1);
    int fjt = nt < 8 ? nt : 8;
    if (const char *e = getenv("VSM_FJ_THREADS")) fjt = std::max(1, std::min(atoi(e), nt));  // (measurements: the fork-join pool of the per-frame path's Delaunay)
    h->fj = new VsmForkJoin(fjt);
    h->work.pool = h->fj;
    h->work.async = h->pool;
  }
  // k_dc_subtrees (vsm_dc.hip) recurses a few levels deep: make sure every thread has the stack for it
  {
    size_t cur = 0;
    if (hipDeviceGetLimit(&cur, hipLimitStackSize) == hipSuccess && cur < 4096) (void)hipDeviceSetLimit(hipLimitStackSize, 4096);
  }
  if (hipGetDevice(&h->device) != hipSuccess || hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking) != hipSuccess ||
      hipEventCreateWithFlags(&h->idle_wait, hipEventBlockingSync | hipEventDisableTiming) != hipSuccess ||
      hipEventCreateWithFlags(&h->seq_ev[0], hipEventBlockingSync | hipEventDisableTiming) != hipSuccess ||
      hipEventCreateWithFlags(&h->seq_ev[1], hipEventBlockingSync | hipEventDisableTiming) != hipSuccess) {
    fprintf(stderr, "visomatch: HIP initialisation failed: %s\n", hipGetErrorString(hipGetLastError()));
    delete h->pool;
    delete h->fj;
    delete h;
    return nullptr;
  }
  return h;
}

static void reset_ring_state(vsm_handle *h) {
  if (h->stage3_in_hm && h->ring.hm_lcount && !h->ring.hm_list2.empty()) {  // the ring's host-mapped block is about to go: stage 3 moves out of it
    const int32_t n = std::max(h->ring.hm_lcount[1], 0);
    h->stage[3].assign(h->ring.hm_list2[0], h->ring.hm_list2[0] + n);
  }
  h->stage3_in_hm = false;
  h->have[0] = h->have[1] = h->right[0] = h->right[1] = false;
  h->f_valid = false;
  h->counts_pending = false;
  h->push_ev_set = false;
  memset(h->n_feat, 0, sizeof(h->n_feat));
}

void vsm_destroy(vsm_handle *h) {
  if (!h) return;
  (void)hipStreamSynchronize(h->stream);
  for (hipStream_t st : h->dc_stream)
    if (st) (void)hipStreamSynchronize(st);
  for (hipStream_t st : h->tie_stream)
    if (st) (void)hipStreamSynchronize(st);
  seq2_destroy(h);
  ctx_destroy(h->ring);
  ctx_destroy(h->seq);
  if (h->stage_host) (void)hipHostFree(h->stage_host);
  if (h->xy_host) (void)hipHostFree(h->xy_host);
  if (h->xy_ev) (void)hipEventDestroy(h->xy_ev);
  if (h->done_ev) (void)hipEventDestroy(h->done_ev);
  if (h->push_ev) (void)hipEventDestroy(h->push_ev);
  for (int k = 0; k < 2; k++) {
    if (h->seq_stage_h[k]) (void)hipHostFree(h->seq_stage_h[k]);
    if (h->seq_stage_d[k]) vsm_dev_free(h->seq_stage_d[k]);
    if (h->seq_stage_ev[k]) (void)hipEventDestroy(h->seq_stage_ev[k]);
  }
  for (int b = 0; b < vsm_handle::kDcBanks; b++)
    if (h->dc_bank[b]) {
      h->dc_bank[b]->release();
      delete h->dc_bank[b];
    }
  for (hipStream_t st : h->dc_stream)
    if (st) (void)hipStreamDestroy(st);
  for (hipStream_t st : h->tie_stream)
    if (st) (void)hipStreamDestroy(st);
  if (h->tie_ev) (void)hipEventDestroy(h->tie_ev);
  for (hipEvent_t e : h->tie_copied)
    if (e) (void)hipEventDestroy(e);
  if (h->hm_ties) (void)hipHostFree(h->hm_ties);
  if (h->idle_wait) (void)hipEventDestroy(h->idle_wait);
  if (h->input_ev) (void)hipEventDestroy(h->input_ev);
  for (hipEvent_t e : h->seq_ev)
    if (e) (void)hipEventDestroy(e);
  if (h->stream) (void)hipStreamDestroy(h->stream);
  delete h->pool;
  delete h->fj;
  delete h;
}

void vsm_set_intrinsics(vsm_handle *h, double f, double cu, double cv, double base) {
  h->param.f = f;
  h->param.cu = cu;
  h->param.cv = cv;
  h->param.base = base;
}

// completes an asynchronous push: waits for the stream and takes the feature counts the
// kernels wrote into host-mapped memory
static int settle(vsm_handle *h) {
  if (!h->counts_pending) return VSM_OK;
  if (h->push_ev_set)
    HIPCHK(hipEventSynchronize(h->push_ev));  // (recorded behind the push's last kernel: cheaper to wait for than the stream)
  else
    HIPCHK(hipStreamSynchronize(h->stream));
  h->push_ev_set = false;
  HIPCHK(hipGetLastError());
  h->prof.resolve();
  const int slot = h->pending_slot;
  memset(h->n_feat[slot], 0, sizeof(h->n_feat[slot]));
  for (int k = 0; k < h->pending_imgs; k++)
    for (int s2 = 0; s2 < 2; s2++) h->n_feat[slot][k][s2] = h->ring.hm_counts[(slot * 2 + k) * 2 + s2];
  h->counts_pending = false;
  return VSM_OK;
}

static int push_common(vsm_handle *h, const uint8_t *I1, const uint8_t *I2, int32_t w, int32_t hh, int32_t bpl,
                       int replace, bool on_device) {
  if (w <= 0 || hh <= 0 || bpl < w || I1 == nullptr) {
    fprintf(stderr, "ERROR: Image dimension mismatch!\n");  // viso/matcher.cpp:103-106
    return VSM_EDIMS;
  }
  HIPCHK(hipSetDevice(h->device));
  VsmCtx &c = h->ring;
  if (!c.ready || w != c.dims.w || hh != c.dims.h) {
    // a change of image size restarts the ring buffer (the reference would match across sizes)
    (void)hipStreamSynchronize(h->stream);
    reset_ring_state(h);
    int rc = ctx_create(c, h->param, w, hh, 2, 1, h->stream, h->sw.match_heads != 0);
    if (rc != VSM_OK) return rc;
  }
  if (h->counts_pending) {  // a previous asynchronous push has not been settled yet
    int rc = settle(h);
    if (rc != VSM_OK) return rc;
  }
  if (!replace) {  // viso/matcher.cpp:123-155: curr becomes prev
    h->cur ^= 1;
    memcpy(h->dims_p, h->dims_c, sizeof(h->dims_p));
  }
  const int slot = h->cur;
  h->dims_c[0] = w;
  h->dims_c[1] = hh;
  h->dims_c[2] = c.dims.bpl;
  const int n_img = I2 ? 2 : 1;
  // (VSM_FRONT=0: the three separate passes - ingest, halving, full-resolution Sobel - instead of the fused front end)
  const bool fused_front = h->param.half_resolution && h->sw.front;
  if (on_device) {
    if (fused_front)
      vsm_launch_front(h->stream, h->prof, c.d_imgs, slot * 2, I1, I2, 0, bpl, 1, c.dims, 1);
    else
      vsm_launch_ingest(h->stream, h->prof, c.d_imgs, slot * 2, I1, I2, 0, bpl, 1, c.dims);
  } else {
    // Host images: rows go through our own pinned, pre-padded staging buffer (a pageable 2-D copy
    // is staged row by row by the runtime and costs milliseconds).  The caller's buffer is free as
    // soon as the memcpy below returns; the staging buffer is reused only after settle().
    const size_t plane = (size_t)c.dims.bpl * hh;
    if (h->stage_bytes < 2 * plane) {
      if (h->stage_host) (void)hipHostFree(h->stage_host);
      HIPCHK(hipHostMalloc((void **)&h->stage_host, 2 * plane, hipHostMallocDefault));
      memset(h->stage_host, 0, 2 * plane);  // pad columns stay 0
      h->stage_bytes = 2 * plane;
    }
    const uint8_t *srcs[2] = {I1, I2};
    for (int k = 0; k < n_img; k++) {
      uint8_t *st = h->stage_host + k * plane;
      // only the w image bytes of a row are taken; pad bytes are 0 by definition
      for (int32_t v = 0; v < hh; v++) memcpy(st + (size_t)v * c.dims.bpl, srcs[k] + (size_t)v * bpl, w);
      HIPCHK(hipMemcpyAsync(c.h_imgs[slot * 2 + k].img, st, plane, hipMemcpyHostToDevice, h->stream));
    }
    // (the padded copies are in place: they are the fused front end's source)
    if (fused_front)
      vsm_launch_front(h->stream, h->prof, c.d_imgs, slot * 2, c.h_imgs[slot * 2].img, I2 ? c.h_imgs[slot * 2 + 1].img : nullptr, 0,
                       c.dims.bpl, 1, c.dims, 0);
  }
  const int f_planes = vsm_launch_features(h->stream, h->prof, c.d_imgs, slot * 2, n_img, c.dims, c.f1, c.f2, c.f_stride, h->param.nms_tau,
                                           h->param.multi_stage, h->param.half_resolution, h->param.match_binsize, c.h_imgs.data(),
                                           fused_front ? 1 : 0, (h->sw.fused_features ? 1 : 0) | (h->sw.filter_planes ? 2 : 0) | (h->sw.feat_order ? 0 : 4));
  HIPCHK(hipGetLastError());
  if (!h->push_ev) HIPCHK(hipEventCreateWithFlags(&h->push_ev, hipEventDisableTiming));
  HIPCHK(hipEventRecord(h->push_ev, h->stream));
  h->push_ev_set = true;
  h->have[slot] = true;
  h->right[slot] = (I2 != nullptr);
  h->pending_slot = slot;
  h->pending_imgs = n_img;
  h->counts_pending = true;
  h->f_valid = f_planes != 0;
  h->gainI[0].clear();
  h->gainI[1].clear();
  // Host images were copied into the staging buffer, so the caller may free or overwrite them as
  // soon as we return (matcherMex does).  Device-resident images: the source must stay valid until
  // the next call that needs the push's results settles it.  Either way the push is asynchronous.
  return VSM_OK;
}

int vsm_wait_for_stream(vsm_handle *h, void *hip_stream) {
  HIPCHK(hipSetDevice(h->device));
  if (!h->input_ev) HIPCHK(hipEventCreateWithFlags(&h->input_ev, hipEventDisableTiming));
  HIPCHK(hipEventRecord(h->input_ev, (hipStream_t)hip_stream));
  HIPCHK(hipStreamWaitEvent(h->stream, h->input_ev, 0));
  return VSM_OK;
}

int vsm_push_back(vsm_handle *h, const uint8_t *I1, const uint8_t *I2, int32_t w, int32_t hh, int32_t bpl, int replace) {
  return push_common(h, I1, I2, w, hh, bpl, replace, false);
}

int vsm_push_back_device(vsm_handle *h, const uint8_t *dI1, const uint8_t *dI2, int32_t w, int32_t hh, int32_t bpl,
                         int replace) {
  return push_common(h, dI1, dI2, w, hh, bpl, replace, true);
}

// sanity checks of viso/matcher.cpp:190-212 on feature counts n[image 1p,2p,1c,2c][set]
// (a NULL set and an empty set are both "count 0" here)
static bool match_ready(const vsm_params &p, int method, const int32_t n[4][2]) {
  if (method == 0) {
    if (n[0][1] == 0 || n[2][1] == 0) return false;
    if (p.multi_stage && (n[0][0] == 0 || n[2][0] == 0)) return false;
  } else if (method == 1) {
    if (n[2][1] == 0 || n[3][1] == 0) return false;
    if (p.multi_stage && (n[2][0] == 0 || n[3][0] == 0)) return false;
  } else {
    if (n[0][1] == 0 || n[1][1] == 0 || n[2][1] == 0 || n[3][1] == 0) return false;
    if (p.multi_stage && (n[0][0] == 0 || n[1][0] == 0 || n[2][0] == 0 || n[3][0] == 0)) return false;
  }
  return true;
}

static VsmMatchCfg make_cfg(const vsm_params &p, int method, int heads) {
  VsmMatchCfg cfg;
  memset(&cfg, 0, sizeof(cfg));
  cfg.method = method;
  cfg.heads = heads;
  cfg.binsize = p.match_binsize;
  cfg.bin_magic = p.match_binsize >= 2 ? (uint32_t)(((1ull << 32) + (uint64_t)p.match_binsize - 1) / (uint64_t)p.match_binsize) : 0u;
  cfg.radius = p.match_radius;
  cfg.disp_tol = p.match_disp_tolerance;
  cfg.f = p.f;
  cfg.cu = p.cu;
  cfg.cv = p.cv;
  cfg.base = p.base;
  return cfg;
}

int vsm_match(vsm_handle *h, int32_t method, const double *Tr) {
  VsmCtx &c = h->ring;
  if (!c.ready) return VSM_ENOTREADY;
  HIPCHK(hipSetDevice(h->device));
  {
    int rc = settle(h);
    if (rc != VSM_OK) return rc;
  }
  const vsm_params &p = h->param;
  const int sc = h->cur, sp = h->cur ^ 1;
  int32_t n[4][2];
  for (int s = 0; s < 2; s++) {
    n[0][s] = h->have[sp] ? h->n_feat[sp][0][s] : 0;
    n[1][s] = h->have[sp] ? h->n_feat[sp][1][s] : 0;
    n[2][s] = h->have[sc] ? h->n_feat[sc][0][s] : 0;
    n[3][s] = h->have[sc] ? h->n_feat[sc][1][s] : 0;
  }
  if (!match_ready(p, method, n)) return VSM_ENOTREADY;
  const double t0 = now_us();
  for (int s = 0; s < 5; s++) h->stage[s].clear();
  h->matched.clear();
  memset(h->counters, 0, sizeof(h->counters));

  VsmMatchCfg cfg = make_cfg(p, method, h->sw.match_heads && c.has_heads);
  VsmJob job;
  memset(&job, 0, sizeof(job));
  // stereo matching only needs the current pair: "prev" then aliases the current slot
  job.img_prev = (method == 1 ? sc : sp) * 2;
  job.img_curr = sc * 2;
  const int qimg = method == 2 ? 0 : 2;  // quad iterates the previous left features, flow/stereo the current
  job.nq[0] = p.multi_stage ? n[qimg][0] : 0;
  job.nq[1] = n[qimg][1];
  job.use_tr = Tr ? 1 : 0;
  if (Tr) memcpy(job.t, Tr, 12 * sizeof(double));
  const int stages = method == 2 ? 4 : 2;
  const VsmDims dp = c.dims, dc = c.dims;  // one size per handle

  double t1 = t0, t2 = t0;
  if (p.multi_stage) {
    cfg.sparse = 1;
    cfg.use_prior = 0;
    if (!vsm_launch_match(h->stream, h->prof, c.d_imgs, c.d_pairs, nullptr, job, 1, c.dims, cfg, job.nq[0], 1))
      vsm_launch_export(h->stream, h->prof, c.d_pairs, 1, 0, job.nq[0]);
    if (!h->done_ev) HIPCHK(hipEventCreateWithFlags(&h->done_ev, hipEventDisableTiming));
    HIPCHK(hipEventRecord(h->done_ev, h->stream));
    HIPCHK(hipEventSynchronize(h->done_ev));  // the list was written into host-mapped memory
    h->stage[0].assign(c.hm_list1[0], c.hm_list1[0] + c.hm_lcount[0]);
    h->counters[0] += (int64_t)job.nq[0] * stages;
    t1 = now_us();
    h->stage[1] = h->stage[0];
    vsm_host_remove_outliers(h->work, p, h->stage[1], method);
    const double ta = now_us();
    vsm_host_prior_statistics(p, h->dims_c, h->stage[1], method, h->ranges);
    const double tb = now_us();
    ranges_to_device_layout(c.h_ranges, h->ranges.data(), h->ranges.size());
    HIPCHK(vsm_upload(h->stream, c.d_ranges, c.h_ranges, h->ranges.size() * sizeof(float)));
    t2 = now_us();
    if (vsm_debug_timing()) {
      static double acc[3] = {0, 0, 0};
      static long calls = 0;
      acc[0] += ta - t1;
      acc[1] += tb - ta;
      acc[2] += t2 - tb;
      if (++calls % 100 == 0)
        fprintf(stderr, "  vsm_match pass-1 host stage, mean us: removeOutliers %.1f, prior statistics %.1f, boxes' upload %.1f (lists of %zu)\n",
                acc[0] / calls, acc[1] / calls, acc[2] / calls, h->stage[0].size());
    }
  }
  cfg.sparse = 0;
  cfg.use_prior = p.multi_stage ? 1 : 0;
  const int nq2 = job.nq[1];
  h->counters[0] += (int64_t)nq2 * stages;
  // The final removeOutliers' triangulation only needs the list's pixels (u1c, v1c), and those are final once the list is
  // compacted (the refinement moves the other three points, viso/matcher.cpp:1544-1577): they cross first, 4 bytes per
  // match, and the host triangulates while the refinement, the list's export and its 350 KB over PCIe are under way; the
  // support test then reads flow and disparity from the refined list.  (Not with refinement = 2, whose fits drop matches,
  // nor when the unrefined list is wanted as a stage view.)
  const bool early_xy = h->sw.frame_early_xy && p.refinement != 2 && !h->capture_stage2 && nq2 > 0;
  h->stage3_in_hm = false;
  if (early_xy) {
    if (h->xy_cap < (size_t)nq2) {
      if (h->xy_host) (void)hipHostFree(h->xy_host);
      h->xy_host = nullptr;
      h->xy_cap = 0;
      const size_t cap = ((size_t)nq2 + 4095) / 4096 * 4096;
      HIPCHK(hipHostMalloc((void **)&h->xy_host, cap * sizeof(uint32_t), hipHostMallocMapped));
      HIPCHK(hipHostGetDevicePointer((void **)&h->xy_dev, h->xy_host, 0));
      h->xy_cap = cap;
    }
    if (!h->xy_ev) HIPCHK(hipEventCreateWithFlags(&h->xy_ev, hipEventDisableTiming));
    if (!vsm_launch_match(h->stream, h->prof, c.d_imgs, c.d_pairs, nullptr, job, 1, c.dims, cfg, nq2, 2, h->xy_dev))
      vsm_launch_export_xy(h->stream, c.d_pairs, h->xy_dev, nq2);
    HIPCHK(hipEventRecord(h->xy_ev, h->stream));
  } else {
    vsm_launch_match(h->stream, h->prof, c.d_imgs, c.d_pairs, nullptr, job, 1, c.dims, cfg, nq2);
  }
  // The list size is still on the device: the refinement / export grids are sized for the worst
  // case (every query matched) and surplus threads exit at once.
  if (h->capture_stage2 && p.refinement == 1)  // debug view: keep the unrefined list (raw is free again)
    HIPCHK(hipMemcpyAsync(c.h_pairs[0].raw, c.h_pairs[0].list2, (size_t)nq2 * sizeof(vsm_p_match), hipMemcpyDeviceToDevice,
                          h->stream));
  if (p.refinement > 0)
    vsm_launch_refine(h->stream, h->prof, c.d_imgs, c.d_pairs, nullptr, job, 1, dp, dc, method, p.refinement, nq2);
  // sub-pixel refinement: the fits' least-squares tail and the removal of failed matches on the device too (as in the look-ahead
  // path; 7 k matches x 3 fits were 3 ms of one host thread) - unless the stage views are wanted, which show the list in between
  const bool fits_on_device = p.refinement == 2 && !h->capture_stage2 && nq2 <= VSM_PARA_MAX_LIST;
  if (fits_on_device) vsm_launch_parabolic_apply(h->stream, c.d_pairs, 1);
  vsm_launch_export(h->stream, h->prof, c.d_pairs, 1, 1, nq2);
  if (early_xy) {
    // (an event behind the export: waiting for it when it has long fired costs 2 us, hipStreamSynchronize on the idle stream 18)
    if (!h->done_ev) HIPCHK(hipEventCreateWithFlags(&h->done_ev, hipEventDisableTiming));
    HIPCHK(hipEventRecord(h->done_ev, h->stream));
    HIPCHK(hipEventSynchronize(h->xy_ev));
    const int32_t n2 = c.hm_lcount[1];  // (written by the compaction, in front of the pixels)
    const double tk = now_us();
    if (n2 > 3) {
      vsm_host_outliers_begin_xy(h->work, h->xy_host, n2);
      h->work.del.run(h->work.x.data(), h->work.y.data(), n2, h->work.pool, h->work.async);
    }
    const double td = now_us();
    HIPCHK(hipEventSynchronize(h->done_ev));
    HIPCHK(hipGetLastError());
    h->prof.resolve();
    const double ts = now_us();
    const vsm_p_match *in = c.hm_list2[0];  // stage 3 is read where the device left it (vsm_stage_get)
    h->stage3_in_hm = true;
    h->counters[3] = n2;
    double tf = ts;
    if (n2 > 3) {
      vsm_host_outliers_begin_flows(h->work, in, n2, method);
      tf = now_us();
      vsm_host_outliers_end(h->work, p, in, n2, method, h->stage[4]);
    } else {
      h->stage[4].assign(in, in + std::max(n2, 0));  // the reference leaves short lists alone (:1210)
    }
    const double te = now_us();
    h->matched = h->stage[4];
    h->counters[4] = (int64_t)h->matched.size();
    const double t4 = now_us();
    if (vsm_debug_timing()) {
      static double acc[5] = {0, 0, 0, 0, 0};
      static long calls = 0;
      acc[0] += td - tk;
      acc[1] += ts - td;
      acc[2] += tf - ts;
      acc[3] += te - tf;
      acc[4] += t4 - te;
      if (++calls % 100 == 0)
        fprintf(stderr, "  vsm_match final host stage, mean us: triangulation %.1f, wait for the list %.1f, flows %.1f, support + survivors %.1f, copy %.1f\n",
                acc[0] / calls, acc[1] / calls, acc[2] / calls, acc[3] / calls, acc[4] / calls);
    }
    h->timings[0] = t1 - t0;
    h->timings[1] = t2 - t1;
    h->timings[2] = tk - t2;  // until the pixels are on the host
    h->timings[3] = t4 - tk;
    h->timings[4] = t4 - t0;
    return VSM_OK;
  }
  HIPCHK(hipStreamSynchronize(h->stream));
  HIPCHK(hipGetLastError());
  const int32_t n2 = c.hm_lcount[1];
  if (fits_on_device) {
    h->stage[3].assign(c.hm_list2[0], c.hm_list2[0] + n2);
  } else if (p.refinement == 1) {
    h->stage[3].assign(c.hm_list2[0], c.hm_list2[0] + n2);
    if (h->capture_stage2) {
      h->stage[2].resize(n2);
      if (n2) HIPCHK(hipMemcpy(h->stage[2].data(), c.h_pairs[0].raw, (size_t)n2 * sizeof(vsm_p_match), hipMemcpyDeviceToHost));
    }
  } else {
    h->stage[2].assign(c.hm_list2[0], c.hm_list2[0] + n2);
    if (p.refinement == 2) {
      const size_t nn2 = h->stage[2].size();
      h->pf.resize(nn2 * 36);
      if (nn2) HIPCHK(hipMemcpy(h->pf.data(), c.h_pairs[0].pf, nn2 * 36 * sizeof(int32_t), hipMemcpyDeviceToHost));
      h->stage[3].clear();
      for (size_t i = 0; i < nn2; i++) {  // viso/matcher.cpp:1541-1581: a failed fit drops the match
        vsm_p_match m = h->stage[2][i];
        bool ok = true;
        float *tu[3] = {&m.u1p, &m.u2c, &m.u2p}, *tv[3] = {&m.v1p, &m.v2c, &m.v2p};
        for (int st = 0; st < 3 && ok; st++) {
          const int32_t *r = &h->pf[(i * 3 + st) * 12];
          if (r[0] == 2) continue;
          ok = r[0] == 1 && vsm_host_parabolic_update(r + 3, r[1], r[2], *tu[st], *tv[st]);
        }
        if (ok) h->stage[3].push_back(m);
      }
    } else {
      h->stage[3] = h->stage[2];
    }
  }
  h->prof.resolve();
  const double t3 = now_us();
  h->counters[3] = (int64_t)h->stage[3].size();
  h->stage[4] = h->stage[3];
  vsm_host_remove_outliers(h->work, p, h->stage[4], method);
  h->matched = h->stage[4];
  h->counters[4] = (int64_t)h->matched.size();
  const double t4 = now_us();
  h->timings[0] = t1 - t0;
  h->timings[1] = t2 - t1;
  h->timings[2] = t3 - t2;
  h->timings[3] = t4 - t3;
  h->timings[4] = t4 - t0;
  return VSM_OK;
}

int32_t vsm_num_matches(vsm_handle *h) { return (int32_t)h->matched.size(); }

int32_t vsm_get_matches(vsm_handle *h, vsm_p_match *out, int32_t cap) {
  int32_t n = (int32_t)h->matched.size();
  if (n > cap) n = cap;
  if (n > 0) memcpy(out, h->matched.data(), (size_t)n * sizeof(vsm_p_match));
  return n;
}

int vsm_bucket(vsm_handle *h, int32_t max_features, float bw, float bh) {
  vsm_host_bucket(h->matched, max_features, bw, bh);
  return VSM_OK;
}

float vsm_gain(vsm_handle *h, const int32_t *inliers, int32_t n) {
  const int sc = h->cur, sp = h->cur ^ 1;
  VsmCtx &c = h->ring;
  if (!c.ready || !h->have[sp] || !h->have[sc] || h->matched.empty() || n == 0 || settle(h) != VSM_OK) return 1;
  const size_t bytes = (size_t)c.dims.bpl * c.dims.h;
  for (int k = 0; k < 2; k++) {  // left images come back from HBM on first use
    if (h->gainI[k].size() != bytes) {
      h->gainI[k].resize(bytes);
      if (hipMemcpy(h->gainI[k].data(), c.h_imgs[(k == 0 ? sp : sc) * 2].img, bytes, hipMemcpyDeviceToHost) != hipSuccess)
        return 1;
    }
  }
  return vsm_host_gain(h->gainI[0].data(), h->gainI[1].data(), h->dims_p, h->dims_c, h->matched, inliers, n);
}

// ---------------------------------------------------------------------------------------
// Final stage of a look-ahead chunk (exact Delaunay support test), shared between host and GPU:
//   A  host pool, per pair: copy the exported list, per-match arrays, ExactDelaunay::prepare (emulated
//      sort, kd order, tree); keys and sub-tree tasks go to the bank's pinned slab
//   G  GPU, second stream: all sub-trees of all pairs in one launch (vsm_dc.hip), records back
//   B  host pool, per pair: adopt the records, the merges above the sub-trees, support, survivors
// The stages hand over to each other without the caller's thread: the last A task to finish enqueues
// G, a host function at the end of G submits B.  VSM_DC_GPU=0 keeps everything on the host.
// ---------------------------------------------------------------------------------------
struct DcChunk {
  vsm_handle *h = nullptr;
  VsmCtx *ctx = nullptr;
  vsm_params p;
  int method = 0, leaf = 16, top = 240;
  bool device_kd = true;  // the GPU orders the keys (k_dc_kd_order); else ExactDelaunay::prepare does
  bool block = true;      // sub-trees of <= VSM_DC_BLOCK_POINTS points, one wave each inside LDS (k_dc_block); else leaf / top
  bool full = true;       // (with block) all merge levels and the support test on the GPU too: only the counts come back
  bool packed = false;    // (with block, not full) the triangle records come back as 12-byte packed words (set when G is enqueued)
  bool ties_gpu = false;  // Triangle's randomised vertex sort runs on the GPU (k_dc_ties_of_lists), beside everything else
  const int32_t *ties = nullptr;   // its verdicts, [n][VSM_DC_TIE_OUT_INTS] (host-mapped)
  std::atomic<int> ties_done{1};
  int bank = 0, n = 0, f0 = 0, first_pair = 0, work0 = 0;
  std::shared_ptr<std::vector<char>> valid;
  std::atomic<int> a_left{0};
  DcBank *B = nullptr;         // its slabs
  VsmHostWork *work = nullptr; // its per-pair host state [n]
  int pass = 1;            // 1: the final stage (pass-2 lists -> seq_matches); 0: pass-1 lists, survivors stay in work[i].tmp_list
  hipStream_t stream = nullptr;
  int chunk = 0;           // the look-ahead chunk it belongs to
  bool submitted = false;  // dc_submit_a() has run (caller's thread only)
  bool a_waited = false;   // (caller's thread only)
  std::atomic<int> stage{0};  // 0: A running, 1: G enqueued, 2: B submitted
  std::atomic<bool> b_once{false};  // B is submitted by whoever comes first: the end of G, or dc_wait() giving up on it
  VsmPool::Ticket a, b;
  // VSM_DEBUG_TIMING: when the stages changed hands, and the task time summed over the pool
  double t_a0 = 0, t_g0 = 0, t_g1 = 0, t_b0 = 0, t_b1 = 0;
  std::atomic<long long> a_ns{0}, b_ns{0}, part_ns[8] = {};  // A: copy, arrays, prepare, slab; B: records, merges, support
  std::atomic<int> b_left{0};
};

// B for pair i of the chunk: the GPU's records adopted, the merges above them, support test, survivors
static void dc_task_b(DcChunk *ch, int i) {
  {
    const double t0 = vsm_now_us();
    if (!(*ch->valid)[i]) return;
    vsm_handle *h = ch->h;
    VsmHostWork &wk = ch->work[i];
    std::vector<vsm_p_match> dummy_out;
    std::vector<vsm_p_match> &out = ch->pass == 1 ? h->seq_matches[ch->f0 + i] : dummy_out;
    const int32_t nl = (int32_t)wk.tmp_list.size();
    if (nl <= 3) {  // the reference leaves short lists alone (:1210)
      if (ch->pass == 1) out.assign(wk.tmp_list.begin(), wk.tmp_list.end());
      return;
    }
    const DcBank &B = *ch->B;
    const int32_t m = B.m[i], nt = B.nt[i];
    if (ch->block && ch->full && nt > 0) {  // the GPU went all the way: keep the matches with support >= 4 (:1369)
      const double t1 = vsm_now_us();
      const int32_t *support = B.h_support + (size_t)i * B.stride_pts;
      vsm_host_keep_supported(wk.tmp_list, support);  // in place, then the buffers change hands
      if (ch->pass == 1) out.swap(wk.tmp_list);
      ch->part_ns[6].fetch_add((long long)((vsm_now_us() - t1) * 1e3), std::memory_order_relaxed);
      return;
    }
    if (m >= 2) {
      if (nt > 0) {  // adopt what the GPU built
        // (copied, not used in place: the merges and the support test chase pointers through these arrays, and
        // on the pinned slab - small pages, no prefetch-friendly order - that cost 25 % of the whole run)
        const DcMesh mesh = wk.del.mesh();
        if (ch->packed) {
          const uint32_t *src = B.h_trip + (size_t)i * B.stride_pts * 6;
          int32_t *dst = mesh.tri;
          for (int32_t t = 0; t < 2 * m; t++, src += 3, dst += 8) {
            for (int o = 0; o < 3; o++) {
              const uint32_t wv = src[o], nb = wv & 0x1ffffu, vx = wv >> 17;
              dst[o] = nb == 0x1ffffu ? -1 : (int32_t)nb;
              dst[4 + o] = vx == 0x7fffu ? -1 : (int32_t)vx;
            }
          }
        } else {
          memcpy(mesh.tri, B.h_tri + (size_t)i * B.stride_pts * 16, (size_t)m * 16 * sizeof(int32_t));
        }
        memcpy(mesh.pt, B.h_pt + (size_t)i * B.stride_pts, (size_t)m * 4);
        memcpy(mesh.id, B.h_id + (size_t)i * B.stride_pts, (size_t)m * 4);
        const VsmDcHull *hu = B.h_hulls + (size_t)i * 2 * B.stride_tasks;
        auto take = [&](int32_t q) { wk.del.set_node_hull(q, ExactDelaunay::OTri{hu[q].fl_t, hu[q].fl_o}, ExactDelaunay::OTri{hu[q].fr_t, hu[q].fr_o}); };
        for (const ExactDelaunay::Task &tk : wk.del.tasks()) take(tk.node);
        for (const ExactDelaunay::Merge &mg : wk.del.device_merges()) take(mg.node);
      } else {
        wk.del.order_keys();
        wk.del.solve_tasks();
        wk.del.solve_merges();
      }
      if (ch->ties_gpu) {  // which match stands for a shared pixel: the GPU's verdict (or, where it declined, the host's)
        while (!ch->ties_done.load(std::memory_order_acquire)) std::this_thread::yield();
        const int32_t *row = ch->ties + (size_t)i * VSM_DC_TIE_OUT_INTS;
        if (row[0] >= 0 && row[0] <= VSM_DC_TIE_PATCHES) wk.del.set_ties(row + 1, row[0]);
        wk.del.apply_ties();
      }
      const double t1 = vsm_now_us();
      ch->part_ns[4].fetch_add((long long)((t1 - t0) * 1e3), std::memory_order_relaxed);
      wk.del.finish();
      ch->part_ns[5].fetch_add((long long)((vsm_now_us() - t1) * 1e3), std::memory_order_relaxed);
    }
    const double t2 = vsm_now_us();
    vsm_host_count_support(wk, ch->p, nl, ch->method);
    vsm_host_keep_supported(wk.tmp_list, wk.support.data());  // in place, then the buffers change hands
    if (ch->pass == 1) out.swap(wk.tmp_list);
    ch->part_ns[6].fetch_add((long long)((vsm_now_us() - t2) * 1e3), std::memory_order_relaxed);
  }
}

static void dc_submit_b(DcChunk *ch) {
  vsm_handle *h = ch->h;
  ch->t_b0 = vsm_now_us();
  ch->b_left.store(ch->n, std::memory_order_relaxed);
  ch->b = h->pool->submit(ch->n, [ch](int i) {
    const double t0 = vsm_now_us();
    dc_task_b(ch, i);
    ch->b_ns.fetch_add((long long)((vsm_now_us() - t0) * 1e3), std::memory_order_relaxed);
    if (ch->b_left.fetch_sub(1, std::memory_order_acq_rel) == 1) ch->t_b1 = vsm_now_us();
  });
  ch->stage.store(2, std::memory_order_release);
}

static void dc_after_gpu(void *arg) {  // runs on a HIP runtime thread: no HIP calls
  DcChunk *ch = (DcChunk *)arg;
  ch->t_g1 = vsm_now_us();
  if (!ch->b_once.exchange(true)) dc_submit_b(ch);
}

// from the pool thread that finished the chunk's last A task; wait_here: from the caller's thread, which waits for the
// GPU's part itself and submits nothing (pass 0)
static void dc_enqueue_gpu(DcChunk *ch, bool wait_here = false) {
  vsm_handle *h = ch->h;
  (void)hipSetDevice(h->device);
  ch->t_g0 = vsm_now_us();
  DcBank &B = *ch->B;
  int maxt = 0, maxm = 0, maxin = 0, maxn = 0, maxlev = 0, maxg = 0, lev_nodes[VSM_DC_MAX_LEVELS] = {0};
  for (int i = 0; i < ch->n; i++) {
    VsmDcJob &jb = B.h_jobs[i];  // (the A task left the level table in it)
    jb.key = B.d_key + (size_t)i * B.stride_pts;
    jb.key_sorted = ch->device_kd && B.nt[i] > 0 ? B.d_key_sorted + (size_t)i * B.stride_pts : nullptr;
    jb.kd_scratch = B.d_kd + (size_t)i * B.stride_pts * VSM_DC_KD_SCRATCH;
    jb.kd_stride = B.stride_pts;
    jb.pt = B.d_pt + (size_t)i * B.stride_pts;
    jb.id = B.d_id + (size_t)i * B.stride_pts;
    jb.tri = B.d_tri + (size_t)i * B.stride_pts * 16;
    jb.tasks = B.d_tasks + (size_t)i * B.stride_tasks;
    jb.merges = B.d_merges + (size_t)i * B.stride_tasks;
    jb.hulls = B.d_hulls + (size_t)i * 2 * B.stride_tasks;
    jb.flow_u = B.d_flow + (size_t)i * 3 * B.stride_pts;
    jb.flow_v = jb.flow_u + B.stride_pts;
    jb.disp = jb.flow_v + B.stride_pts;
    jb.support = ch->full ? B.d_support + (size_t)i * B.stride_pts : nullptr;
    jb.ntasks = std::max(B.nt[i], 0);
    jb.m = B.m[i];
    maxt = std::max(maxt, jb.ntasks);
    if (B.nt[i] > 0) {
      maxm = std::max(maxm, B.m[i]);
      maxin = std::max(maxin, jb.n_in);
      maxn = std::max(maxn, B.nn[i]);
      maxlev = std::max(maxlev, jb.nlevels);
      maxg = std::max(maxg, jb.level_off[jb.nlevels]);
      for (int l = 0; l < jb.nlevels; l++) lev_nodes[l] = std::max(lev_nodes[l], jb.level_off[l + 1] - jb.level_off[l]);
    } else {
      jb.nlevels = 0;
    }
  }
  ch->packed = ch->block && !ch->full && maxm > 0 && maxm <= VSM_DC_PACKED_MAX_POINTS;
  for (int i = 0; i < ch->n; i++) B.h_jobs[i].tri_packed = ch->packed ? B.d_trip + (size_t)i * B.stride_pts * 6 : nullptr;
  // only the used part of every pair's slab row travels: rows of maxm points / maxt tasks
  const size_t sp = (size_t)B.stride_pts, st = (size_t)B.stride_tasks, rows = (size_t)ch->n;
  hipStream_t s2 = ch->stream;
  bool ok = true;
  if (maxt > 0) {
    ok = hipMemcpy2DAsync(ch->device_kd ? B.d_key_sorted : B.d_key, sp * 8, B.h_key, sp * 8, (size_t)maxm * 8, rows, hipMemcpyHostToDevice,
                          s2) == hipSuccess &&
         hipMemcpy2DAsync(B.d_tasks, st * sizeof(VsmDcTask), B.h_tasks, st * sizeof(VsmDcTask), (size_t)maxt * sizeof(VsmDcTask), rows,
                          hipMemcpyHostToDevice, s2) == hipSuccess &&
         (maxg == 0 || hipMemcpy2DAsync(B.d_merges, st * sizeof(VsmDcMerge), B.h_merges, st * sizeof(VsmDcMerge),
                                        (size_t)maxg * sizeof(VsmDcMerge), rows, hipMemcpyHostToDevice, s2) == hipSuccess) &&
         hipMemcpyAsync(B.d_jobs, B.h_jobs, ch->n * sizeof(VsmDcJob), hipMemcpyHostToDevice, s2) == hipSuccess &&
         // (k_dc_block writes every slot of every sub-tree; the per-lane kernels rely on empty slots reading -1)
         (ch->block || hipMemset2DAsync(B.d_tri, sp * 64, 0xff, (size_t)maxm * 64, rows, s2) == hipSuccess);
    if (ok) {
      VsmProf &pf = h->prof;
      if (ch->device_kd) {
        pf.begin(VSM_K_DC_KD, s2);
        vsm_dc_launch_kd_order(s2, B.d_jobs, ch->n);
        pf.end(s2);
      }
      pf.begin(VSM_K_DC_BLOCK, s2);
      if (ch->block) {
        vsm_dc_launch_blocks(s2, B.d_jobs, ch->n, maxt);
      } else {
        vsm_dc_launch_subtrees(s2, B.d_jobs, ch->n, maxt);
      }
      pf.end(s2);
      if (maxlev > 0) {
        pf.begin(VSM_K_DC_MERGE, s2);
        for (int l = 0; l < maxlev; l++) vsm_dc_launch_merge_level(s2, B.d_jobs, ch->n, l, lev_nodes[l]);
        pf.end(s2);
      }
      if (ch->block && ch->full) {
        // the triangulations are complete on the device: count the support there, only the counts travel
        ok = hipMemcpy2DAsync(B.d_flow, sp * 12, B.h_flow, sp * 12, sp * 12, rows, hipMemcpyHostToDevice, s2) == hipSuccess &&
             hipMemset2DAsync(B.d_support, sp * 4, 0, (size_t)maxin * 4, rows, s2) == hipSuccess;
        if (ok) {
          pf.begin(VSM_K_DC_SUPPORT, s2);
          vsm_dc_launch_support(s2, B.d_jobs, ch->n, maxm, ch->method, (float)ch->p.outlier_flow_tolerance, (float)ch->p.outlier_disp_tolerance);
          pf.end(s2);
          ok = hipMemcpy2DAsync(B.h_support, sp * 4, B.d_support, sp * 4, (size_t)maxin * 4, rows, hipMemcpyDeviceToHost, s2) == hipSuccess;
        }
      } else {
        ok = (ch->packed ? hipMemcpy2DAsync(B.h_trip, sp * 24, B.d_trip, sp * 24, (size_t)maxm * 24, rows, hipMemcpyDeviceToHost, s2)
                         : hipMemcpy2DAsync(B.h_tri, sp * 64, B.d_tri, sp * 64, (size_t)maxm * 64, rows, hipMemcpyDeviceToHost, s2)) == hipSuccess &&
             hipMemcpy2DAsync(B.h_pt, sp * 4, B.d_pt, sp * 4, (size_t)maxm * 4, rows, hipMemcpyDeviceToHost, s2) == hipSuccess &&
             hipMemcpy2DAsync(B.h_id, sp * 4, B.d_id, sp * 4, (size_t)maxm * 4, rows, hipMemcpyDeviceToHost, s2) == hipSuccess &&
             hipMemcpy2DAsync(B.h_hulls, 2 * st * sizeof(VsmDcHull), B.d_hulls, 2 * st * sizeof(VsmDcHull), (size_t)maxn * sizeof(VsmDcHull),
                              rows, hipMemcpyDeviceToHost, s2) == hipSuccess;
      }
    }
  }
  ch->stage.store(1, std::memory_order_release);
  if (wait_here) {
    if (maxt > 0 && !(ok && hipStreamSynchronize(s2) == hipSuccess)) {
      (void)hipStreamSynchronize(s2);
      for (int i = 0; i < ch->n; i++)
        if (B.nt[i] > 0) B.nt[i] = -1;
    }
    return;
  }
  // (option dc_fault_inject = 1, tests only: the completion callback is "lost" - dc_wait()'s watchdog has to notice)
  const bool lose_callback = ch->h->sw.dc_fault_inject == 1;
  if (ok && maxt > 0 && lose_callback) return;
  if (ok && maxt > 0 && hipLaunchHostFunc(s2, dc_after_gpu, ch) == hipSuccess) return;
  // nothing for the GPU, or it could not be used: the host solves the sub-trees too
  if (maxt > 0) {
    (void)hipStreamSynchronize(s2);
    for (int i = 0; i < ch->n; i++)
      if (B.nt[i] > 0) B.nt[i] = -1;
  }
  if (!ch->b_once.exchange(true)) dc_submit_b(ch);
}

// A for pair i of the chunk: the list out of host-mapped memory, the per-match arrays, the host's part of the triangulation
static void dc_task_a(DcChunk *ch, int i) {
  {
    const double t0 = vsm_now_us();
    VsmHostWork &wk = ch->work[i];
    DcBank &B = *ch->B;
    B.m[i] = 0;
    B.nt[i] = 0;
    wk.tmp_list.clear();
    if ((*ch->valid)[i]) {
      const int pj = ch->first_pair + i;
      // one wide copy out of the host-mapped export, then cache-resident work
      if (ch->pass == 1)
        wk.tmp_list.assign(ch->ctx->hm_list2[pj], ch->ctx->hm_list2[pj] + ch->ctx->hm_lcount[2 * pj + 1]);
      else
        wk.tmp_list.assign(ch->ctx->hm_list1[pj], ch->ctx->hm_list1[pj] + ch->ctx->hm_lcount[2 * pj]);
      const int32_t nl = (int32_t)wk.tmp_list.size();
      const double t1 = vsm_now_us();
      ch->part_ns[0].fetch_add((long long)((t1 - t0) * 1e3), std::memory_order_relaxed);
      if (nl > 3) {
        vsm_host_outliers_begin(wk, wk.tmp_list.data(), nl, ch->method);
        const double t2 = vsm_now_us();
        ch->part_ns[1].fetch_add((long long)((t2 - t1) * 1e3), std::memory_order_relaxed);
        // (ties_gpu: the emulated vertex sort is the GPU's, ExactDelaunay::prepare's defer_ties; on the host it would only
        // move 150 us per pair from in front of the GPU's part to beside it, and cost a radix sort on top)
        const bool prepared = ch->block ? wk.del.prepare(wk.x.data(), wk.y.data(), nl, VSM_DC_BLOCK_POINTS, nullptr, ch->full ? INT32_MAX : 0,
                                                         ch->device_kd, ch->ties_gpu)
                                        : wk.del.prepare(wk.x.data(), wk.y.data(), nl, ch->leaf, nullptr, ch->top, ch->device_kd, ch->ties_gpu);
        ch->part_ns[2].fetch_add((long long)((vsm_now_us() - t2) * 1e3), std::memory_order_relaxed);
        if (prepared) {
          const int32_t m = wk.del.points(), nt = (int32_t)wk.del.tasks().size(), ng = (int32_t)wk.del.device_merges().size();
          const std::vector<int32_t> &lv = wk.del.device_levels();
          B.m[i] = m;
          B.nn[i] = wk.del.num_nodes();
          if (m < nl) ch->part_ns[7].fetch_add(1, std::memory_order_relaxed);  // pairs with duplicate points
          if (m > B.stride_pts || nl > B.stride_pts || nt > B.stride_tasks || ng > B.stride_tasks || B.nn[i] > 2 * B.stride_tasks ||
              (int)lv.size() > VSM_DC_MAX_LEVELS || (ch->device_kd && m > VSM_DC_KD_MAX_POINTS)) {
            B.nt[i] = -1;  // does not fit the slab: this pair stays on the host
          } else {
            memcpy(B.h_key + (size_t)i * B.stride_pts, wk.del.mesh().key, (size_t)m * 8);
            memcpy(B.h_tasks + (size_t)i * B.stride_tasks, wk.del.tasks().data(), (size_t)nt * sizeof(VsmDcTask));
            memcpy(B.h_merges + (size_t)i * B.stride_tasks, wk.del.device_merges().data(), (size_t)ng * sizeof(VsmDcMerge));
            if (ch->full) {
              float *fl = B.h_flow + (size_t)i * 3 * B.stride_pts;
              memcpy(fl, wk.fu.data(), (size_t)nl * 4);
              memcpy(fl + B.stride_pts, wk.fv.data(), (size_t)nl * 4);
              memcpy(fl + 2 * (size_t)B.stride_pts, wk.dp.data(), (size_t)nl * 4);
            }
            VsmDcJob &jb = B.h_jobs[i];
            jb.n_in = nl;
            jb.nlevels = (int32_t)lv.size();
            jb.level_off[0] = 0;
            for (int l = 0; l < jb.nlevels; l++) jb.level_off[l + 1] = jb.level_off[l] + lv[l];
            B.nt[i] = nt;
          }
        }
      }
    }
    ch->a_ns.fetch_add((long long)((vsm_now_us() - t0) * 1e3), std::memory_order_relaxed);
  }
}

static void dc_submit_a(DcChunk *ch) {
  ch->submitted = true;
  ch->a_left.store(ch->n, std::memory_order_relaxed);
  ch->t_a0 = vsm_now_us();
  ch->a = ch->h->pool->submit(ch->n, [ch](int i) {
    dc_task_a(ch, i);
    if (ch->a_left.fetch_sub(1, std::memory_order_acq_rel) == 1) dc_enqueue_gpu(ch);  // the last one hands over
  });
}

// Until the chunk's final lists are in seq_matches.  The GPU's part normally takes a millisecond or two and reports back
// through a host function on its stream.  If nothing has been heard after the watchdog time (option dc_watchdog_ms, default
// 20 s) the stream itself is asked: hipStreamSynchronize() either returns an error - the device faulted; that is logged
// with HIP's own message, remembered in the handle (no GPU share from then on) and reported by vsm_sequence_run as
// VSM_EHIP - or it returns success, in which case the device's results are complete and only the callback went missing.
// Either way nothing on the device can touch the chunk's slabs any more when the host takes over, and the chunk and
// its bank stay alive until then (they are owned by vsm_sequence_run, which calls this for every chunk before it returns).
static void dc_wait(DcChunk *ch) {
  vsm_handle *h = ch->h;
  const double watchdog_us = (double)h->sw.dc_watchdog_ms * 1e3;
  if (!ch->submitted) {  // (left early, between setting it up and submitting it: only its vertex sorts may be in flight)
    const double t0 = vsm_now_us();
    while (!ch->ties_done.load(std::memory_order_acquire)) {
      std::this_thread::sleep_for(std::chrono::microseconds(50));
      if (vsm_now_us() - t0 > watchdog_us) {
        const hipError_t e = hipStreamSynchronize(h->tie_stream[ch->bank & 1]);
        if (e != hipSuccess) {
          fprintf(stderr, "visomatch: the vertex-sort stream of the Delaunay stage failed: %s\n", hipGetErrorString(e));
          h->dc_gpu_broken = true;
          h->seq_hip_error.store(1);
        }
        ch->ties_done.store(1, std::memory_order_release);
      }
    }
    return;
  }
  const double t0 = vsm_now_us();
  while (ch->stage.load(std::memory_order_acquire) < 2) {
    std::this_thread::sleep_for(std::chrono::microseconds(50));
    if (ch->stage.load(std::memory_order_acquire) == 1 && vsm_now_us() - t0 > watchdog_us && !ch->b_once.exchange(true)) {
      const hipError_t e = hipStreamSynchronize(ch->stream);  // (blocks until the stream has drained or failed)
      hipError_t et = hipSuccess;
      if (ch->ties_gpu) et = hipStreamSynchronize(h->tie_stream[ch->bank & 1]);
      if (e != hipSuccess || et != hipSuccess) {
        fprintf(stderr, "visomatch: the GPU share of the Delaunay stage failed (%s); finishing the chunk on the host, no GPU share from now on\n",
                hipGetErrorString(e != hipSuccess ? e : et));
        h->dc_gpu_broken = true;
        h->seq_hip_error.store(1);
        DcBank &B = *ch->B;
        for (int i = 0; i < ch->n; i++)
          if (B.nt[i] > 0) B.nt[i] = -1;
        ch->full = false;
        if (ch->ties_gpu) {  // (its vertex sorts will not report either: the host's verdicts)
          for (int i = 0; i < ch->n; i++) const_cast<int32_t *>(ch->ties)[(size_t)i * VSM_DC_TIE_OUT_INTS] = -1;
          ch->ties_done.store(1, std::memory_order_release);
        }
      } else {
        fprintf(stderr, "visomatch: the GPU share of the Delaunay stage finished without reporting back; continuing with its results\n");
        if (ch->ties_gpu) ch->ties_done.store(1, std::memory_order_release);
      }
      dc_submit_b(ch);
    }
  }
  h->pool->wait(ch->b);
}

// Host images of a look-ahead chunk: a pageable 2-D copy is staged row by row by the runtime (milliseconds per image), so
// the pool gathers the chunk's 2 n images into a pinned slot (w bytes per row), one upload follows, and k_ingest reads the
// device twin like any device-resident input.  Two slots alternate; a slot is reused once its ingest has run.
static int seq_ingest_host_frames(vsm_handle *h, VsmCtx &c, int first_img, const uint8_t *left, const uint8_t *right, int64_t frame_stride,
                                  int32_t bpl, int32_t w, int32_t hh, int32_t f0, int n) {
  const size_t img = (size_t)w * hh, need = 2 * img * (size_t)h->seq_chunk;
  if (h->seq_stage_bytes < need) {
    (void)hipStreamSynchronize(h->stream);
    for (int k = 0; k < 2; k++) {
      if (h->seq_stage_h[k]) (void)hipHostFree(h->seq_stage_h[k]);
      if (h->seq_stage_d[k]) vsm_dev_free(h->seq_stage_d[k]);
      h->seq_stage_h[k] = h->seq_stage_d[k] = nullptr;
      HIPCHK(hipHostMalloc((void **)&h->seq_stage_h[k], need, hipHostMallocDefault));
      HIPCHK(vsm_dev_alloc((void **)&h->seq_stage_d[k], need));
      if (!h->seq_stage_ev[k]) HIPCHK(hipEventCreateWithFlags(&h->seq_stage_ev[k], hipEventDisableTiming));
      HIPCHK(hipEventRecord(h->seq_stage_ev[k], h->stream));
    }
    h->seq_stage_bytes = need;
  }
  const int slot = h->seq_stage_next;
  h->seq_stage_next ^= 1;
  HIPCHK(hipEventSynchronize(h->seq_stage_ev[slot]));  // its previous content has been ingested
  uint8_t *dst = h->seq_stage_h[slot];
  const int sides = right ? 2 : 1;
  h->pool->run(sides * n, [&](int t) {
    const int i = right ? t >> 1 : t, side = right ? (t & 1) : 0;
    const uint8_t *src = (side ? right : left) + (size_t)(f0 + i) * frame_stride;
    uint8_t *d = dst + ((size_t)side * n + i) * img;
    if (bpl == w) {
      memcpy(d, src, img);
    } else {
      for (int32_t v = 0; v < hh; v++) memcpy(d + (size_t)v * w, src + (size_t)v * bpl, w);
    }
  });
  HIPCHK(hipMemcpyAsync(h->seq_stage_d[slot], dst, (size_t)sides * img * n, hipMemcpyHostToDevice, h->stream));
  const uint8_t *d0 = h->seq_stage_d[slot];
  const bool fused = h->param.half_resolution && h->sw.front;
  // (the front kernels number their images first + 2 * frame + side: consecutive mono frames go through as (even, odd) pairs)
  auto front = [&](int first, const uint8_t *s0, const uint8_t *s1, size_t stride, int frames) {
    if (frames <= 0) return;
    if (fused)
      vsm_launch_front(h->stream, h->prof, c.d_imgs, first, s0, s1, stride, w, frames, c.dims, 0);
    else
      vsm_launch_ingest(h->stream, h->prof, c.d_imgs, first, s0, s1, stride, w, frames, c.dims);
  };
  if (right) {
    front(first_img, d0, d0 + img * n, img, n);
  } else {
    front(first_img, d0, d0 + img, 2 * img, n / 2);
    if (n & 1) front(first_img + n - 1, d0 + img * (size_t)(n - 1), nullptr, img, 1);
  }
  HIPCHK(hipEventRecord(h->seq_stage_ev[slot], h->stream));
  return VSM_OK;
}

}  // extern "C"
#include "vsm_seq2.inc"
#include "vsm_multi.inc"
extern "C" {

// ---------------------------------------------------------------------------------------
// Look-ahead sequence API.  Semantics: exactly pushBack(frame f) + matchFeatures(method, Tr[f])
// for f = 0..n-1 on a fresh matcher.  Frames are processed in chunks of C: every kernel runs once
// per chunk over all its images / pairs, and the host stages (exact Delaunay support, prior
// statistics) of the chunk's pairs run concurrently on the pool, one pair per task.
// ---------------------------------------------------------------------------------------
static int sequence_fallback(vsm_handle *h, const uint8_t *left, const uint8_t *right, int64_t frame_stride, int on_device,
                             int32_t n_frames, int32_t w, int32_t hh, int32_t bpl, int32_t method, const double *Tr,
                             const uint8_t *Tr_valid) {
  // rarely used configurations (mono input, refinement==2) go frame by frame on a fresh ring
  (void)hipStreamSynchronize(h->stream);
  for (hipStream_t st : h->dc_stream)
    if (st) (void)hipStreamSynchronize(st);
  for (hipStream_t st : h->tie_stream)
    if (st) (void)hipStreamSynchronize(st);
  seq2_destroy(h);
  ctx_destroy(h->ring);
  reset_ring_state(h);
  h->matched.clear();
  for (int32_t f = 0; f < n_frames; f++) {
    const uint8_t *l = left + (size_t)f * frame_stride, *r = right ? right + (size_t)f * frame_stride : nullptr;
    int rc = push_common(h, l, r, w, hh, bpl, 0, on_device != 0);
    if (rc != VSM_OK) return rc;
    const double *t = (Tr && (!Tr_valid || Tr_valid[f])) ? Tr + (size_t)f * 12 : nullptr;
    rc = vsm_match(h, method, t);
    if (rc != VSM_OK && rc != VSM_ENOTREADY) return rc;
    h->seq_matches[f] = h->matched;
  }
  return VSM_OK;
}

int vsm_sequence_run(vsm_handle *h, const uint8_t *left, const uint8_t *right, int64_t frame_stride, int on_device,
                     int32_t n_frames, int32_t w, int32_t hh, int32_t bpl, int32_t method, const double *Tr,
                     const uint8_t *Tr_valid) {
  if (w <= 0 || hh <= 0 || bpl < w || left == nullptr || n_frames <= 0) {
    fprintf(stderr, "ERROR: Image dimension mismatch!\n");
    return VSM_EDIMS;
  }
  const double t_entry = now_us();
  HIPCHK(hipSetDevice(h->device));
  const vsm_params &p = h->param;
  h->seq_v2_frames = 0;
  h->seq_matches.resize(n_frames);  // keeps the capacity of earlier runs: no page-fault storm
  for (auto &v : h->seq_matches) v.clear();
  h->seq_view.assign(n_frames, vsm_handle::SeqView());
  // (mono input can only be flow-matched; the host-shared form below has no sub-pixel refinement - its fits and dropped matches
  // are the GPU-resident form's, or the per-frame code's)
  if ((!right && (method != 0 || !h->sw.seq_v2)) || (p.refinement == 2 && !h->sw.seq_v2))
    return sequence_fallback(h, left, right, frame_stride, on_device, n_frames, w, hh, bpl, method, Tr, Tr_valid);

  // The GPU-resident form (vsm_seq2.inc) takes the run unless VSM_SEQ_V2=0 asks for the host-shared form below, or
  // it declines (lists beyond what its device-side vertex sort / kd order were written for).
  // The host-shared form lives off the rank's host threads (200 frames 1242x375, round 3: 7.6 ms with 14 pool threads, 15 with 8,
  // 24 with 2); the GPU-resident one keeps only Triangle's vertex sort and the closing of the result lists on the host (5.0-5.2 ms
  // with 14 threads, 6.3 with 8, 8.0 with 4, 8.9 with 2, 9.1 with 1 - results in host memory included), so it is the default
  // whatever the thread count.
  if (h->sw.seq_v2) {
    const int rc = sequence_run_v2(h, left, right, frame_stride, on_device, n_frames, w, hh, bpl, method, Tr, Tr_valid);
    if (rc != VSM_SEQ2_DECLINED) return rc;
    h->seq_v2_frames = 0;
    if (!right || p.refinement == 2) return sequence_fallback(h, left, right, frame_stride, on_device, n_frames, w, hh, bpl, method, Tr, Tr_valid);
  }
  int C = h->sw.seq_chunk > 0 ? h->sw.seq_chunk : 50;
  if (C > n_frames) C = n_frames;
  VsmCtx &c = h->seq;
  if (!c.ready || c.dims.w != w || c.dims.h != hh || h->seq_chunk != C || c.npairs != 2 * C || c.nframes != 3 * C) {
    (void)hipStreamSynchronize(h->stream);
    int rc = ctx_create(c, p, w, hh, 3 * C, 2 * C, h->stream, h->sw.match_heads != 0);  // three banks of frames, two of pairs
    if (rc != VSM_OK) return rc;
    h->seq_chunk = C;
  }
  // Software pipeline over chunks.  GPU order: pass 1 of chunk k, features of chunk k+1, pass 2 of
  // chunk k - so the GPU has the next chunk's features to compute while the pool does chunk k's
  // prior statistics, and the caller's thread never waits for features.  Frames live in three banks
  // (chunk k+1's features must not overwrite the last frame of chunk k-1, which chunk k's first pair
  // reads); pairs in two (the final host stage of chunk k reads pair bank k&1 in host-mapped memory
  // while the GPU runs chunk k+1 on the other).
  std::vector<VsmPool::Ticket> tickets;  // final stages that stay on the host ...
  std::vector<int> ticket_chunk;         // ... and the chunk each belongs to
  // final stage: see DcChunk above
  // (options dc_gpu / dc_full, vsm_set_option.  Round 1's sub-variants of this form - per-lane sub-trees of VSM_DC_LEAF points
  // with merge levels up to VSM_DC_TOP points in global memory, the kd order on the host, the vertex sort on one wave of
  // the device - are compile-time choices now: tools/build_variant.sh NAME "-DVSM_DC_BLOCK_FORM=0 -DVSM_DC_LEAF=64 ...")
#ifndef VSM_DC_LEAF
#define VSM_DC_LEAF 16
#endif
#ifndef VSM_DC_KD_ON_GPU
#define VSM_DC_KD_ON_GPU 1
#endif
#ifndef VSM_DC_BLOCK_FORM
#define VSM_DC_BLOCK_FORM 1
#endif
#ifndef VSM_DC_TOP
#define VSM_DC_TOP 240
#endif
#ifndef VSM_DC_TIES_ON_GPU
#define VSM_DC_TIES_ON_GPU 0
#endif
  const VsmSwitches &sw = h->sw;
  const bool dc_env = sw.dc_gpu != 0;
  const bool dc_forced = sw.dc_gpu > 0;
  const int dc_leaf = std::max(3, VSM_DC_LEAF);
  const bool dc_kd = VSM_DC_KD_ON_GPU != 0;      // kd order of the keys on the GPU too
  const bool dc_block = VSM_DC_BLOCK_FORM != 0;  // k_dc_block instead of leaf / top
  // (with block) the merges above the sub-trees and the support test on the GPU too: a third less host work per
  // pair, but the large merges are slow there (a dependent L2 round trip per step), so it pays when the host has
  // few cores for this rank (200 frames 1242x375, ms: 16 threads 9.1 shared / 14.2 full, 8: 12.9 / 14.9, 4: 21.5 / 18.6,
  // 2: 33.1 / 25.8; host only: 14.0, 25.3, 43.4, 70.8); option dc_full = 0 / 1 decides otherwise
  const bool dc_full = sw.dc_full >= 0 ? sw.dc_full != 0 : h->pool->size() <= 6;
  const int dc_top = VSM_DC_TOP;  // merge nodes up to this size follow on the GPU
  bool dc_gpu = dc_env && !h->dc_gpu_broken;
  h->seq_hip_error.store(0);
  for (hipStream_t &st : h->dc_stream)
    if (dc_gpu && !st) dc_gpu = hipStreamCreateWithFlags(&st, hipStreamNonBlocking) == hipSuccess;  // (stream priorities make no measurable difference)
  // -DVSM_DC_TIES_ON_GPU=1: Triangle's randomised vertex sort on the GPU too (k_dc_ties_of_keys, one wave per pair, started right
  // behind the pass-2 compaction).  Exact, and it takes 150 us per pair off the host, but one wave needs 3.2 ms for a
  // 7.4 k list (0.65 us per partition, all dependent scalar work) - longer than everything else of a chunk together, so
  // the B stage ends up waiting for it: 13.4 ms per 200 frames against 9.0.  Off unless asked for.
  const bool dc_ties = dc_gpu && VSM_DC_TIES_ON_GPU != 0;
  if (dc_ties) {
    bool ok = true;
    for (hipStream_t &st : h->tie_stream)
      if (ok && !st) ok = hipStreamCreateWithFlags(&st, hipStreamNonBlocking) == hipSuccess;
    if (ok && !h->tie_ev) ok = hipEventCreateWithFlags(&h->tie_ev, hipEventDisableTiming) == hipSuccess;
    for (hipEvent_t &e : h->tie_copied)
      if (ok && !e) ok = hipEventCreateWithFlags(&e, hipEventDisableTiming) == hipSuccess;
    h->tie_copied_set[0] = h->tie_copied_set[1] = false;
    if (ok && (!h->hm_ties || h->ties_chunk < C)) {
      if (h->hm_ties) (void)hipHostFree(h->hm_ties);
      h->hm_ties = nullptr;
      ok = hipHostMalloc((void **)&h->hm_ties, (size_t)vsm_handle::kDcBanks * C * VSM_DC_TIE_OUT_INTS * sizeof(int32_t), hipHostMallocMapped) == hipSuccess &&
           hipHostGetDevicePointer((void **)&h->d_ties, h->hm_ties, 0) == hipSuccess;
      h->ties_chunk = C;
    }
    if (!ok && h->hm_ties) {
      (void)hipHostFree(h->hm_ties);
      h->hm_ties = nullptr;
    }
  }
  if (dc_gpu) {
    for (int b = 0; b < vsm_handle::kDcBanks; b++)
      if (!h->dc_bank[b]) h->dc_bank[b] = new DcBank();
    if ((int)h->seq_work.size() < vsm_handle::kDcBanks * C) h->seq_work.resize((size_t)vsm_handle::kDcBanks * C);
  }
  std::vector<std::unique_ptr<DcChunk>> chunks;
  struct Drain {  // whatever way this function is left, nothing of it may still be running
    std::vector<std::unique_ptr<DcChunk>> &c;
    ~Drain() {
      for (auto &ch : c) dc_wait(ch.get());
    }
  } drain{chunks};
  const int32_t dims_c[3] = {w, hh, c.dims.bpl};
  int32_t nprev[2][2] = {{0, 0}, {0, 0}};  // feature counts [side][set] of the previous chunk's last frame
  double tg = 0, thost = 0;
  std::atomic<long long> mid_ns[2] = {};  // VSM_DEBUG_TIMING: pass-1 outlier removal, prior statistics (task time)
  const double tstart = now_us();
  // Chunk boundaries: chunks of C frames; a sequence of at least three chunks starts (and ends) with a half chunk - the
  // host pool has nothing to do until the first chunk's lists exist, and nothing overlaps the last chunk's final stage
  std::vector<int32_t> chunk_start;
  {
    const bool taper = true;
    const int32_t half = C / 2;
    int32_t f = 0;
    if (taper && half >= 8 && n_frames >= 3 * C) {
      chunk_start.push_back(0);
      f = half;
      while (n_frames - f > C + half) {
        chunk_start.push_back(f);
        f += C;
      }
      if (n_frames - f > C) {  // between C and 3C/2 frames left: a full chunk and a short one
        chunk_start.push_back(f);
        f = n_frames - std::min<int32_t>(half, n_frames - f - 1);
      }
      chunk_start.push_back(f);
    } else {
      for (; f < n_frames; f += C) chunk_start.push_back(f);
    }
    chunk_start.push_back(n_frames);
  }
  const int nchunks = (int)chunk_start.size() - 1;
  auto launch_features_of = [&](int k) -> hipError_t {  // ingest + all feature kernels of chunk k, then the marker
    const int32_t f0 = chunk_start[k];
    const int n = chunk_start[k + 1] - f0;
    const int first_img = 2 * (k % 3) * C;
    const bool fused_front = p.half_resolution && h->sw.front;
    if (on_device) {
      if (fused_front)
        vsm_launch_front(h->stream, h->prof, c.d_imgs, first_img, left + (size_t)f0 * frame_stride, right + (size_t)f0 * frame_stride,
                         (size_t)frame_stride, bpl, n, c.dims, 0);
      else
        vsm_launch_ingest(h->stream, h->prof, c.d_imgs, first_img, left + (size_t)f0 * frame_stride,
                          right + (size_t)f0 * frame_stride, (size_t)frame_stride, bpl, n, c.dims);
    } else {
      if (seq_ingest_host_frames(h, c, first_img, left, right, frame_stride, bpl, w, hh, f0, n) != VSM_OK) return hipErrorUnknown;
    }
    vsm_launch_features(h->stream, h->prof, c.d_imgs, first_img, 2 * n, c.dims, c.f1, c.f2, c.f_stride, p.nms_tau,
                        p.multi_stage, p.half_resolution, p.match_binsize, c.h_imgs.data(), fused_front ? 1 : 0, (h->sw.fused_features ? 1 : 0) | (h->sw.feat_order ? 0 : 4));
    return hipEventRecord(h->seq_ev[0], h->stream);
  };
  // what a chunk needs from one step to the next
  struct SeqChunk {
    int32_t f0 = 0;
    int n = 0, bank = 0, first_img = 0, first_pair = 0;
    int max_nq[2] = {0, 0};
    std::shared_ptr<std::vector<char>> validp;
    DcChunk *dc = nullptr;  // its final stage, if that is shared with the GPU
    double t_pass2 = 0;
  };
  std::vector<SeqChunk> sc(nchunks);
  // First step of chunk k: wait for its features, one job per frame, pass 1 (if there is one) and its export.
  // Order on the stream: ... pass 2 of k-1, features of k+1, pass 1 of k+1, pass 2 of k, features of k+2 ...: while the
  // pool computes chunk k's prior statistics the GPU has pass 2 of chunk k-1 and the features of chunk k+1 to do, and
  // pass 1 of chunk k+1 is over before its prior statistics are wanted.
  auto start_chunk = [&](int k, bool then_features) -> int {
    SeqChunk &q = sc[k];
    q.f0 = chunk_start[k];
    q.n = chunk_start[k + 1] - q.f0;
    q.bank = k & 1;
    q.first_img = 2 * (k % 3) * C;
    q.first_pair = q.bank * C;
    const int32_t f0 = q.f0;
    const int n = q.n, first_img = q.first_img, first_pair = q.first_pair;
    const VsmPair *d_pairs = c.d_pairs + first_pair;
    VsmJob *h_jobs = c.h_jobs + first_pair, *d_jobs = c.d_jobs + first_pair;
    int *max_nq = q.max_nq;
    const double tl0 = now_us();
    HIPCHK(hipEventSynchronize(h->seq_ev[0]));  // the chunk's feature counts are in host-mapped memory
    HIPCHK(hipGetLastError());
    if (vsm_debug_timing()) fprintf(stderr, "  chunk %d: feature wait %.0f us\n", k, now_us() - tl0);
    // ---- one job per frame of the chunk ----
    max_nq[0] = max_nq[1] = 0;
    q.validp = std::make_shared<std::vector<char>>(n, 0);
    std::vector<char> &valid = *q.validp;
    for (int i = 0; i < n; i++) {
      const int32_t f = f0 + i;
      VsmJob &jb = h_jobs[i];
      memset(&jb, 0, sizeof(jb));
      const int img_c = first_img + 2 * i;
      int img_p;
      int32_t cnt[4][2];
      for (int s = 0; s < 2; s++) {
        cnt[2][s] = c.hm_counts[img_c * 2 + s];
        cnt[3][s] = c.hm_counts[(img_c + 1) * 2 + s];
      }
      if (method == 1) {
        img_p = img_c;
        for (int s = 0; s < 2; s++) cnt[0][s] = cnt[1][s] = 0;
      } else if (i > 0) {
        img_p = img_c - 2;
        for (int s = 0; s < 2; s++) {
          cnt[0][s] = c.hm_counts[img_p * 2 + s];
          cnt[1][s] = c.hm_counts[(img_p + 1) * 2 + s];
        }
      } else {  // the previous frame is the last one of the previous chunk's bank
        img_p = 2 * ((k + 2) % 3) * C + 2 * ((k > 0 ? chunk_start[k] - chunk_start[k - 1] : 1) - 1);
        for (int s = 0; s < 2; s++) {
          cnt[0][s] = f > 0 ? nprev[0][s] : 0;
          cnt[1][s] = f > 0 ? nprev[1][s] : 0;
        }
      }
      jb.img_prev = img_p;
      jb.img_curr = img_c;
      if (match_ready(p, method, cnt)) {
        valid[i] = 1;
        const int qimg = method == 2 ? 0 : 2;
        jb.nq[0] = p.multi_stage ? cnt[qimg][0] : 0;
        jb.nq[1] = cnt[qimg][1];
        if (Tr && (!Tr_valid || Tr_valid[f])) {
          jb.use_tr = 1;
          memcpy(jb.t, Tr + (size_t)f * 12, 12 * sizeof(double));
        }
      }
      max_nq[0] = std::max(max_nq[0], jb.nq[0]);
      max_nq[1] = std::max(max_nq[1], jb.nq[1]);
    }
    for (int s = 0; s < 2; s++) {  // remember the last frame's counts for the next chunk
      nprev[0][s] = c.hm_counts[(first_img + 2 * (n - 1)) * 2 + s];
      nprev[1][s] = c.hm_counts[(first_img + 2 * (n - 1) + 1) * 2 + s];
    }
    HIPCHK(vsm_upload(h->stream, d_jobs, h_jobs, sizeof(VsmJob) * n));
    if (p.multi_stage) {
      VsmMatchCfg cfg = make_cfg(p, method, h->sw.match_heads && c.has_heads);
      VsmJob dummy;
      memset(&dummy, 0, sizeof(dummy));
      cfg.sparse = 1;
      cfg.use_prior = 0;
      vsm_launch_match(h->stream, h->prof, c.d_imgs, d_pairs, d_jobs, dummy, n, c.dims, cfg, max_nq[0]);
      vsm_launch_export(h->stream, h->prof, d_pairs, n, 0, max_nq[0]);
      HIPCHK(hipEventRecord(h->seq_ev[1], h->stream));
    }
    if (then_features && k + 1 < nchunks) HIPCHK(launch_features_of(k + 1));
    return VSM_OK;
  };
  // Last step of chunk j: its pass 2 is over, the final stage goes to the pool (and from there to the GPU and back)
  VsmCtx *cp = &c;
  auto finalize = [&](int j) -> int {
    const SeqChunk &q = sc[j];
    const double t0 = now_us();
    HIPCHK(hipEventSynchronize(h->idle_wait));
    HIPCHK(hipGetLastError());
    // (kernel timing: the spans are read once, after the last chunk - the event pool grows over the sequence instead of
    // the pipeline being drained per chunk, so the profiled pass overlaps its kernels like any other)
    // (kernel timing is resolved at the end of the run, when the Delaunay streams have drained too)
    if (vsm_debug_timing()) fprintf(stderr, "  chunk %d: pass2 launched %.0f us ago, waited %.0f us for it\n", j, t0 - q.t_pass2, now_us() - t0);
    tg += now_us() - t0;
    if (q.dc) {
      dc_submit_a(q.dc);
    } else {
      const vsm_params pcopy = p;
      const std::shared_ptr<std::vector<char>> validp = q.validp;
      const int32_t f0 = q.f0;
      const int first_pair = q.first_pair;
      ticket_chunk.push_back(j);
      tickets.push_back(h->pool->submit(q.n, [h, cp, pcopy, validp, f0, first_pair, method](int i) {
        if (!(*validp)[i]) return;
        static thread_local VsmHostWork tw;
        const int pj = first_pair + i;
        std::vector<vsm_p_match> &out = h->seq_matches[f0 + i];
        // one wide copy out of the host-mapped export, then cache-resident work
        tw.tmp_list.assign(cp->hm_list2[pj], cp->hm_list2[pj] + cp->hm_lcount[2 * pj + 1]);
        vsm_host_remove_outliers_from(tw, pcopy, tw.tmp_list.data(), (int32_t)tw.tmp_list.size(), method, out);
      }));
    }
    return VSM_OK;
  };
  HIPCHK(launch_features_of(0));
  {
    const int rc = start_chunk(0, true);
    if (rc != VSM_OK) return rc;
  }
  for (int32_t k = 0; k < nchunks; k++) {
    const SeqChunk &q = sc[k];
    const int32_t f0 = q.f0;
    const int n = q.n, bank = q.bank, first_pair = q.first_pair;
    const VsmPair *d_pairs = c.d_pairs + first_pair;
    const VsmJob *d_jobs = c.d_jobs + first_pair;
    const int *max_nq = q.max_nq;
    const std::shared_ptr<std::vector<char>> validp = q.validp;
    const std::vector<char> &valid = *validp;
    VsmMatchCfg cfg = make_cfg(p, method, h->sw.match_heads && c.has_heads);
    VsmJob dummy;
    memset(&dummy, 0, sizeof(dummy));
    double ta = now_us();
    if (p.multi_stage) {
      const double tl1 = now_us();
      HIPCHK(hipEventSynchronize(h->seq_ev[1]));
      double tb = now_us();
      if (vsm_debug_timing()) fprintf(stderr, "  chunk %d: pass1 sync %.0f us\n", k, tb - tl1);
      tg += tb - ta;
      h->pool->run(n, [&](int i) {  // queued behind the previous chunk's final stage (FIFO)
        static thread_local VsmHostWork tw;
        static thread_local std::vector<float> rg;
        static thread_local std::vector<vsm_p_match> m1;
        const int pj = first_pair + i;
        const double t0 = now_us();
        m1.clear();
        if (valid[i]) m1.assign(c.hm_list1[pj], c.hm_list1[pj] + c.hm_lcount[2 * pj]);
        vsm_host_remove_outliers(tw, p, m1, method);
        const double t1 = now_us();
        vsm_host_prior_statistics(p, dims_c, m1, method, rg);
        ranges_to_device_layout(c.h_ranges + (size_t)pj * c.ranges_stride, rg.data(), rg.size());
        mid_ns[0].fetch_add((long long)((t1 - t0) * 1e3), std::memory_order_relaxed);
        mid_ns[1].fetch_add((long long)((now_us() - t1) * 1e3), std::memory_order_relaxed);
      });
      ta = now_us();
      thost += ta - tb;
      HIPCHK(vsm_upload(h->stream, c.d_ranges + (size_t)first_pair * c.ranges_stride, c.h_ranges + (size_t)first_pair * c.ranges_stride,
                        c.ranges_stride * 4 * n));
    }
    if (k > 0) {  // pass 2 of the previous chunk ran meanwhile
      const int rc = finalize(k - 1);
      if (rc != VSM_OK) return rc;
    }
    if (k + 1 < nchunks) {  // pass 1 of the next chunk goes in front of this chunk's pass 2
      const int rc = start_chunk(k + 1, false);
      if (rc != VSM_OK) return rc;
    }
    ta = now_us();
    // the export below overwrites this pair bank's host lists: chunk k-2 must be done with them
    const double tw0 = now_us();
    for (auto &ch : chunks)  // (they copied the lists out first thing)
      if (ch->chunk <= k - 2 && !ch->a_waited) {
        h->pool->wait(ch->a);
        ch->a_waited = true;
      }
    for (size_t q = 0; q < tickets.size(); q++)
      if (ticket_chunk[q] <= k - 2 && tickets[q]) {
        h->pool->wait(tickets[q]);
        tickets[q].reset();
      }
    if (vsm_debug_timing() && now_us() - tw0 > 2000) fprintf(stderr, "  chunk %d: waited %.0f us for chunk %d's final stage\n", k, now_us() - tw0, k - 2);
    // the chunk's final stage is set up here already: the GPU's emulated vertex sorts start right behind the compaction
    // The GPU share pays when the pool has other pairs to work on while the GPU has this chunk's (its part is
    // latency-bound): a chunk with fewer pairs than pool threads stays on the host, unless VSM_DC_GPU=1 insists
    bool use_dc = dc_gpu && (dc_forced || n >= h->pool->size());
    const int dc_q = (int)chunks.size(), dc_b = dc_q % vsm_handle::kDcBanks;
    if (use_dc) {
      if (dc_q >= vsm_handle::kDcBanks) dc_wait(chunks[dc_q - vsm_handle::kDcBanks].get());  // its slabs are reused now
      // slab sizes from this chunk's longest possible list (every pair's list is at most max_nq[1] long)
      const int pts = ((max_nq[1] + 63) / 64) * 64 + 64, tsk = 2 * pts / std::max(dc_leaf, 2) + 16;
      if (!h->dc_bank[dc_b]->reserve(C, pts, tsk)) {
        fprintf(stderr, "visomatch: no memory for the GPU share of the Delaunay stage, staying on the host\n");
        dc_gpu = use_dc = false;
      }
    }
    if (use_dc) {
      chunks.emplace_back(new DcChunk());
      DcChunk *ch = chunks.back().get();
      ch->h = h;
      ch->ctx = cp;
      ch->p = p;
      ch->method = method;
      ch->leaf = dc_leaf;
      ch->top = dc_top;
      ch->device_kd = dc_kd;
      ch->block = dc_block;
      ch->full = dc_full;
      ch->chunk = k;
      ch->bank = dc_b;
      ch->n = n;
      ch->f0 = f0;
      ch->first_pair = first_pair;
      ch->work0 = ch->bank * C;
      ch->B = h->dc_bank[dc_b];
      ch->work = h->seq_work.data() + ch->work0;
      ch->stream = h->dc_stream[dc_b & 1];
      ch->valid = validp;
      ch->ties_gpu = dc_ties && ch->block && !ch->full && h->hm_ties != nullptr;
      ch->ties = h->hm_ties + (size_t)dc_b * h->ties_chunk * VSM_DC_TIE_OUT_INTS;
      sc[k].dc = ch;
    }
    // (this pair bank's pass-2 lists are about to be rewritten: the copy of chunk k-2's keys out of them comes first)
    if (h->tie_copied_set[bank]) HIPCHK(hipStreamWaitEvent(h->stream, h->tie_copied[bank], 0));
    cfg.sparse = 0;
    cfg.use_prior = p.multi_stage ? 1 : 0;
    vsm_launch_match(h->stream, h->prof, c.d_imgs, d_pairs, d_jobs, dummy, n, c.dims, cfg, max_nq[1]);
    if (use_dc && chunks.back()->ties_gpu) {
      DcChunk *ch = chunks.back().get();
      hipStream_t ts = h->tie_stream[dc_b & 1];
      ch->ties_done.store(0, std::memory_order_relaxed);
      DcBank &B = *h->dc_bank[dc_b];
      bool ok = hipEventRecord(h->tie_ev, h->stream) == hipSuccess && hipStreamWaitEvent(ts, h->tie_ev, 0) == hipSuccess;
      if (ok) {
        vsm_dc_launch_tie_keys(ts, d_pairs, n, max_nq[1], B.d_tie_keys, B.stride_pts, B.d_tie_n);
        ok = hipEventRecord(h->tie_copied[bank], ts) == hipSuccess;
        h->tie_copied_set[bank] = ok;
      }
      if (ok) {
        vsm_dc_launch_ties_of_keys(ts, n, B.d_tie_keys, B.stride_pts, B.d_tie_n, h->d_ties + (size_t)dc_b * h->ties_chunk * VSM_DC_TIE_OUT_INTS,
                                   VSM_DC_TIE_OUT_INTS);
        ok = hipLaunchHostFunc(ts, [](void *arg) { ((DcChunk *)arg)->ties_done.store(1, std::memory_order_release); }, ch) == hipSuccess;
      }
      if (!ok) {  // the host does it then (in A, as without this)
        (void)hipStreamSynchronize(ts);
        ch->ties_gpu = false;
        ch->ties_done.store(1, std::memory_order_relaxed);
      }
    }
    if (p.refinement > 0)
      vsm_launch_refine(h->stream, h->prof, c.d_imgs, d_pairs, d_jobs, dummy, n, c.dims, c.dims, method, p.refinement,
                        max_nq[1]);
    vsm_launch_export(h->stream, h->prof, d_pairs, n, 1, max_nq[1]);
    sc[k].t_pass2 = now_us();
    HIPCHK(hipEventRecord(h->idle_wait, h->stream));
    if (k + 2 < nchunks) HIPCHK(launch_features_of(k + 2));
    tg += now_us() - ta;
  }
  if (nchunks > 0) {
    const int rc = finalize(nchunks - 1);
    if (rc != VSM_OK) return rc;
  }
  {
    const double tb = now_us();
    for (auto &ch : chunks) dc_wait(ch.get());
    if (vsm_debug_timing())
      for (auto &ch : chunks)
        fprintf(stderr, "  final stage of %d pairs: A %.0f..%.0f us (tasks %.0f us), G ..%.0f, B %.0f..%.0f (tasks %.0f us)\n", ch->n,
                ch->t_a0 - tstart, ch->t_g0 - tstart, ch->a_ns.load() * 1e-3, ch->t_g1 - tstart, ch->t_b0 - tstart, ch->t_b1 - tstart,
                ch->b_ns.load() * 1e-3);
    if (vsm_debug_timing() && !chunks.empty()) {
      double part[8] = {0};
      for (auto &ch : chunks)
        for (int q = 0; q < 8; q++) part[q] += ch->part_ns[q].load() * 1e-3 / n_frames;
      fprintf(stderr, "  per pair, us: pass-1 outliers %.0f prior statistics %.0f | A copy %.0f arrays %.0f prepare %.0f | B records %.0f merges %.0f support+survivors %.0f; pairs with duplicate points: %.0f\n",
              mid_ns[0].load() * 1e-3 / n_frames, mid_ns[1].load() * 1e-3 / n_frames, part[0], part[1], part[2], part[4], part[5], part[6], part[7] * n_frames * 1e3);
    }
    for (auto &t : tickets)
      if (t) h->pool->wait(t);
    thost += now_us() - tb;
    if (h->prof.on) {
      HIPCHK(hipStreamSynchronize(h->stream));
      for (hipStream_t st : h->dc_stream)
        if (st) HIPCHK(hipStreamSynchronize(st));
      h->prof.resolve();
    }
  }
  h->seq_timings[0] = tg;
  h->seq_timings[1] = thost;
  h->seq_timings[2] = now_us() - tstart;
  h->seq_timings[3] = (double)C;
  if (vsm_debug_timing())
    fprintf(stderr, "seq: entry->start %.0f us, gpu %.0f, host %.0f, total %.0f\n", tstart - t_entry, tg, thost,
            h->seq_timings[2]);
  // (a Delaunay stream that failed: the lists are complete - the host finished those chunks - but the caller must know)
  return h->seq_hip_error.load() ? VSM_EHIP : VSM_OK;
}

int32_t vsm_sequence_num_matches(vsm_handle *h, int32_t frame) {
  if (frame < 0 || frame >= (int32_t)h->seq_matches.size()) return 0;
  if (frame < (int32_t)h->seq_view.size() && h->seq_view[frame].p) return h->seq_view[frame].n;
  return (int32_t)h->seq_matches[frame].size();
}

// (both look-ahead forms end with every frame's list in the reference's 48-byte p_match form in host memory)
int32_t vsm_sequence_get_matches(vsm_handle *h, int32_t frame, vsm_p_match *out, int32_t cap) {
  if (frame < 0 || frame >= (int32_t)h->seq_matches.size()) return 0;
  const bool view = frame < (int32_t)h->seq_view.size() && h->seq_view[frame].p;
  int32_t n = view ? h->seq_view[frame].n : (int32_t)h->seq_matches[frame].size();
  if (n > cap) n = cap;
  if (n > 0) memcpy(out, view ? h->seq_view[frame].p : h->seq_matches[frame].data(), (size_t)n * sizeof(vsm_p_match));
  return n;
}

void vsm_sequence_get_timings(vsm_handle *h, double *out4) { memcpy(out4, h->seq_timings, sizeof(h->seq_timings)); }
int32_t vsm_sequence_path(vsm_handle *h) { return h->seq_v2_frames > 0 ? 2 : 1; }
int32_t vsm_local_cpus(int32_t *out, int32_t cap) { return vsm_affinity_cpus(out, out ? cap : 0); }
int32_t vsm_forkjoin_cpus(int32_t *out, int32_t cap) {
  if (vsm_forkjoin_domain() < 0) return 0;
  if (vsm_forkjoin_per_core()) {  // the workers have a core each: the caller's is core 0 of their domain
    const int32_t n = vsm_affinity_core_cpus(nullptr, vsm_forkjoin_domain(), 0, out, out ? cap : 0);
    if (n > 0) return n;
  }
  return vsm_affinity_domain_cpus(nullptr, vsm_forkjoin_domain(), out, out ? cap : 0);
}

int vsm_host_register(const void *p, uint64_t bytes) {
  if (!p || bytes == 0) return VSM_EARG;
  const hipError_t e = hipHostRegister(const_cast<void *>(p), (size_t)bytes, hipHostRegisterDefault);
  if (e == hipSuccess) return VSM_OK;
  (void)hipGetLastError();
  return e == hipErrorHostMemoryAlreadyRegistered ? VSM_OK : VSM_EHIP;
}
int vsm_host_unregister(const void *p) {
  if (!p) return VSM_EARG;
  const hipError_t e = hipHostUnregister(const_cast<void *>(p));
  if (e != hipSuccess) (void)hipGetLastError();
  return e == hipSuccess ? VSM_OK : VSM_EHIP;
}
int vsm_set_option(vsm_handle *h, const char *name, int32_t value) { return (h && name && h->sw.set(name, value)) ? VSM_OK : VSM_EARG; }

// ---- stage-level views ----
static bool which_set(vsm_handle *h, int32_t which, int &img, int &set, int32_t &n) {
  if (which < 0 || which > 7 || !h->ring.ready) return false;
  if (settle(h) != VSM_OK) return false;
  const int prev = (which & 3) < 2, side = which & 1;
  set = which >> 2;
  const int slot = prev ? (h->cur ^ 1) : h->cur;
  img = slot * 2 + side;
  n = h->have[slot] ? h->n_feat[slot][side][set] : 0;
  return true;
}

int32_t vsm_num_features(vsm_handle *h, int32_t which) {
  int img, set;
  int32_t n;
  return which_set(h, which, img, set, n) ? n : 0;
}

int32_t vsm_get_features(vsm_handle *h, int32_t which, int32_t *out, int32_t cap) {
  int img, set;
  int32_t n;
  if (!which_set(h, which, img, set, n)) return 0;
  if (n > cap) n = cap;
  if (n > 0 && hipMemcpy(out, h->ring.h_imgs[img].set[set].feat, (size_t)n * 48, hipMemcpyDeviceToHost) != hipSuccess) return 0;
  return n;
}

void vsm_set_stage_capture(vsm_handle *h, int on) { h->capture_stage2 = on ? 1 : 0; }

// (stage 3 of the per-frame path - the refined list in front of the final removeOutliers - stays where the device exported
// it, in the ring's host-mapped block, until the next vsm_match: no copy of it inside the call)
int32_t vsm_stage_size(vsm_handle *h, int32_t s) {
  if (s == 3 && h->stage3_in_hm) return std::max(h->ring.hm_lcount[1], 0);
  return (s >= 0 && s < 5) ? (int32_t)h->stage[s].size() : 0;
}

int32_t vsm_stage_get(vsm_handle *h, int32_t s, vsm_p_match *out, int32_t cap) {
  if (s < 0 || s >= 5) return 0;
  const bool hm = s == 3 && h->stage3_in_hm;
  int32_t n = hm ? std::max(h->ring.hm_lcount[1], 0) : (int32_t)h->stage[s].size();
  if (n > cap) n = cap;
  if (n > 0) memcpy(out, hm ? h->ring.hm_list2[0] : h->stage[s].data(), (size_t)n * sizeof(vsm_p_match));
  return n;
}

int32_t vsm_num_ranges(vsm_handle *h) { return (int32_t)(h->ranges.size() / 16); }

int32_t vsm_get_ranges(vsm_handle *h, float *out, int32_t cap_bins) {
  int32_t n = (int32_t)(h->ranges.size() / 16);
  if (n > cap_bins) n = cap_bins;
  if (n > 0) memcpy(out, h->ranges.data(), (size_t)n * 64);
  return n;
}

int32_t vsm_get_gradients(vsm_handle *h, int32_t which, int32_t full, uint8_t *du, uint8_t *dv) {
  VsmCtx &c = h->ring;
  if (!c.ready || which < 0 || which > 3 || settle(h) != VSM_OK) return 0;
  const int slot = which < 2 ? (h->cur ^ 1) : h->cur, side = which & 1;
  if (!h->have[slot] || (side && !h->right[slot])) return 0;
  if (full && !h->param.half_resolution) return 0;
  const VsmImage &im = c.h_imgs[slot * 2 + side];
  const int32_t bytes = full ? c.dims.bpl * c.dims.h : c.dims.mbpl * c.dims.mh;
  if (full) {  // the tiled plane (8 x 8 tiles; a tile row = du 0-3, dv 0-3, du 4-7, dv 4-7: vsm_tiled_at in vsm_kernels.hip), un-tiled here
    const int bpl = c.dims.bpl, hh = c.dims.h;
    std::vector<uint8_t> t((size_t)bpl * ((hh + 7) & ~7) * 2);
    if (hipMemcpy(t.data(), im.duv_tiled, t.size(), hipMemcpyDeviceToHost) != hipSuccess) return 0;
    for (int y = 0; y < hh; y++)
      for (int x = 0; x < bpl; x++) {
        const size_t a = ((size_t)(y >> 3) * (size_t)(bpl >> 3) + (size_t)(x >> 3)) * 128 + (size_t)((y & 7) * 16 + ((x & 4) << 1) + (x & 3));
        if (du) du[(size_t)y * bpl + x] = t[a];
        if (dv) dv[(size_t)y * bpl + x] = t[a + 4];
      }
    return bytes;
  }
  if (du && hipMemcpy(du, im.du, bytes, hipMemcpyDeviceToHost) != hipSuccess) return 0;
  if (dv && hipMemcpy(dv, im.dv, bytes, hipMemcpyDeviceToHost) != hipSuccess) return 0;
  return bytes;
}

int32_t vsm_get_filter_responses(vsm_handle *h, int16_t *f1, int16_t *f2) {
  VsmCtx &c = h->ring;
  if (!c.ready || !h->f_valid || settle(h) != VSM_OK) return 0;
  const int32_t n = c.dims.mbpl * c.dims.mh;
  if (f1 && hipMemcpy(f1, c.f1, (size_t)n * 2, hipMemcpyDeviceToHost) != hipSuccess) return 0;
  if (f2 && hipMemcpy(f2, c.f2, (size_t)n * 2, hipMemcpyDeviceToHost) != hipSuccess) return 0;
  return n;
}

static const char *kKernelNames[VSM_K_COUNT] = {
    "k_ingest", "k_halve", "k_filters<true>", "k_filters<false>", "k_nms:dense", "k_nms:sparse", "k_scan_cells", "k_emit", "k_bin_scan",
    "k_bin_scatter", "k_bin_rank", "k_match<16>:pass1", "k_compact_matches:pass1", "k_match<16>:pass2",
    "k_compact_matches:pass2", "k_refine", "k_export_list", "k_front",
    "k_dc_keys", "k_dc_vertex_sort", "k_dc_prepare_kd_order", "k_dc_block", "k_dc_merge", "k_dc_support", "k_dc_compact", "k_dc_prior",
    "k_feat_dense", "k_feat_sparse", "k_feat_scan", "k_feat_order"};

void vsm_set_profiling(vsm_handle *h, int on) {
  h->prof.on = on != 0;
  h->prof.print_spans = on >= 1100;
  h->prof.only = on >= 1100 ? on - 1100 : (on >= 100 ? on - 100 : -1);
  if (on) {
    memset(h->prof.total_ms, 0, sizeof(h->prof.total_ms));
    memset(h->prof.launches, 0, sizeof(h->prof.launches));
  }
}

int32_t vsm_num_kernels(void) { return VSM_K_COUNT; }

const char *vsm_kernel_name(int32_t id) { return (id >= 0 && id < VSM_K_COUNT) ? kKernelNames[id] : ""; }

void vsm_get_kernel_stats(vsm_handle *h, double *total_ms, int64_t *launches) {
  memcpy(total_ms, h->prof.total_ms, sizeof(h->prof.total_ms));
  memcpy(launches, h->prof.launches, sizeof(h->prof.launches));
}

int32_t vsm_host_delaunay(const int32_t *x, const int32_t *y, int32_t n, int32_t *tris, int32_t cap, int32_t threads) {
  ExactDelaunay d;
  VsmForkJoin pool(threads);
  VsmPool side(2);  // with more than one thread the emulated vertex sort runs next to the triangulation (as in a handle)
  d.run(x, y, n, threads > 1 ? &pool : nullptr, threads > 1 ? &side : nullptr);
  const int32_t nt = d.num_triangles();
  for (int32_t i = 0; i < nt && i < cap; i++)
    for (int k = 0; k < 3; k++) tris[i * 3 + k] = d.triangles()[i * 3 + k];
  return nt;
}

int32_t vsm_host_delaunay_split(const int32_t *x, const int32_t *y, int32_t n, int32_t *tris, int32_t cap,
                                int32_t max_task_points, int32_t device_top_points) {
  ExactDelaunay d;
  // (device_top_points < 0: additionally with the emulated vertex sort set apart, see ExactDelaunay::prepare)
  if (d.prepare(x, y, n, max_task_points, nullptr, std::max(device_top_points, 0), false, device_top_points < 0)) {
    d.solve_tasks();
    d.solve_merges();
    d.finish();
    d.resolve_ties();
    d.apply_ties();
  }
  const int32_t nt = d.num_triangles();
  for (int32_t i = 0; i < nt && i < cap; i++)
    for (int k = 0; k < 3; k++) tris[i * 3 + k] = d.triangles()[i * 3 + k];
  return nt;
}

// One prepared triangulation on the device, `copies` times (test hook and micro-benchmark below)
namespace {
struct DcDeviceCopy {
  std::vector<VsmDcJob> jobs;
  VsmDcTask *d_tasks = nullptr;
  VsmDcMerge *d_merges = nullptr;
  VsmDcJob *d_jobs = nullptr;
  int nt = 0, nn = 0, m = 0, nlev = 0, lev_nodes[VSM_DC_MAX_LEVELS] = {0};
  size_t tri_bytes = 0;
  bool device_kd = false;  // the keys arrive in (x,y) order and k_dc_kd_order runs first
  bool block = false;      // k_dc_block instead of k_dc_subtrees + k_dc_merge_level
  bool create(ExactDelaunay &d, int copies) {
    static_assert(sizeof(VsmDcTask) == sizeof(ExactDelaunay::Task), "task layout");
    static_assert(sizeof(VsmDcMerge) == sizeof(ExactDelaunay::Merge), "merge layout");
    m = d.points();
    nt = (int)d.tasks().size();
    nn = d.num_nodes();
    const int ng = (int)d.device_merges().size();
    nlev = (int)d.device_levels().size();
    if (nlev > VSM_DC_MAX_LEVELS) return false;
    tri_bytes = (size_t)m * 2 * 8 * sizeof(int32_t);
    jobs.assign(copies, VsmDcJob());
    if (hipMalloc((void **)&d_tasks, sizeof(VsmDcTask) * nt) != hipSuccess ||
        hipMalloc((void **)&d_merges, sizeof(VsmDcMerge) * std::max(ng, 1)) != hipSuccess ||
        hipMalloc((void **)&d_jobs, sizeof(VsmDcJob) * copies) != hipSuccess)
      return false;
    (void)hipMemcpy(d_tasks, d.tasks().data(), sizeof(VsmDcTask) * nt, hipMemcpyHostToDevice);
    (void)hipMemcpy(d_merges, d.device_merges().data(), sizeof(VsmDcMerge) * ng, hipMemcpyHostToDevice);
    for (VsmDcJob &job : jobs) {
      if (hipMalloc((void **)&job.key, (size_t)m * 8) != hipSuccess || hipMalloc((void **)&job.pt, (size_t)m * 4) != hipSuccess ||
          hipMalloc((void **)&job.id, (size_t)m * 4) != hipSuccess || hipMalloc((void **)&job.tri, tri_bytes) != hipSuccess ||
          hipMalloc((void **)&job.hulls, sizeof(VsmDcHull) * nn) != hipSuccess)
        return false;
      if (device_kd && (hipMalloc((void **)&job.key_sorted, (size_t)m * 8) != hipSuccess ||
                        hipMalloc((void **)&job.kd_scratch, (size_t)m * 4 * VSM_DC_KD_SCRATCH) != hipSuccess))
        return false;
      job.kd_stride = m;
      job.tasks = d_tasks;
      job.merges = d_merges;
      job.ntasks = nt;
      job.m = m;
      job.nlevels = nlev;
      job.level_off[0] = 0;
      for (int l = 0; l < nlev; l++) {
        job.level_off[l + 1] = job.level_off[l] + d.device_levels()[l];
        lev_nodes[l] = d.device_levels()[l];
      }
    }
    return hipMemcpy(d_jobs, jobs.data(), sizeof(VsmDcJob) * copies, hipMemcpyHostToDevice) == hipSuccess;
  }
  void load(const DcMesh &mesh, hipStream_t s) {
    for (VsmDcJob &job : jobs) {
      (void)hipMemcpyAsync(device_kd ? (void *)job.key_sorted : (void *)job.key, mesh.key, (size_t)m * 8, hipMemcpyHostToDevice, s);
      (void)hipMemsetAsync(job.tri, 0xff, tri_bytes, s);
    }
  }
  void launch(hipStream_t s) {
    if (device_kd) vsm_dc_launch_kd_order(s, d_jobs, (int)jobs.size());
    if (block) {
      vsm_dc_launch_blocks(s, d_jobs, (int)jobs.size(), nt);
      return;
    }
    vsm_dc_launch_subtrees(s, d_jobs, (int)jobs.size(), nt);
    for (int l = 0; l < nlev; l++) vsm_dc_launch_merge_level(s, d_jobs, (int)jobs.size(), l, lev_nodes[l]);
  }
  ~DcDeviceCopy() {
    for (VsmDcJob &job : jobs) {
      (void)hipFree(job.key);
      (void)hipFree(job.pt);
      (void)hipFree(job.id);
      (void)hipFree(job.tri);
      (void)hipFree(job.hulls);
      (void)hipFree((void *)job.key_sorted);
      (void)hipFree(job.kd_scratch);
    }
    (void)hipFree(d_tasks);
    (void)hipFree(d_merges);
    (void)hipFree(d_jobs);
  }
};
}  // namespace

// test hook: prepare on the host; sub-trees and the merge nodes of at most device_top_points points on
// the GPU (vsm_dc.hip); the merges above them on the host
int32_t vsm_debug_delaunay_gpu(const int32_t *x, const int32_t *y, int32_t n, int32_t *tris, int32_t cap,
                               int32_t max_task_points, int32_t device_top_points, int32_t device_kd) {
  ExactDelaunay d;
  if (d.prepare(x, y, n, max_task_points, nullptr, std::max(device_top_points, 0), device_kd != 0)) {
    if (d.points() > VSM_DC_KD_MAX_POINTS) return -1;
    const DcMesh mesh = d.mesh();
    DcDeviceCopy dev;
    dev.device_kd = device_kd != 0;
    dev.block = device_top_points < 0;
    if (!dev.create(d, 1)) return -1;
    dev.load(mesh, nullptr);
    dev.launch(nullptr);
    if (hipDeviceSynchronize() != hipSuccess) return -1;
    std::vector<VsmDcHull> hulls(dev.nn);
    (void)hipMemcpy(mesh.tri, dev.jobs[0].tri, dev.tri_bytes, hipMemcpyDeviceToHost);
    (void)hipMemcpy(mesh.pt, dev.jobs[0].pt, (size_t)dev.m * 4, hipMemcpyDeviceToHost);
    (void)hipMemcpy(mesh.id, dev.jobs[0].id, (size_t)dev.m * 4, hipMemcpyDeviceToHost);
    (void)hipMemcpy(hulls.data(), dev.jobs[0].hulls, sizeof(VsmDcHull) * dev.nn, hipMemcpyDeviceToHost);
    auto take = [&](int32_t q) {
      d.set_node_hull(q, ExactDelaunay::OTri{hulls[q].fl_t, hulls[q].fl_o}, ExactDelaunay::OTri{hulls[q].fr_t, hulls[q].fr_o});
    };
    for (const ExactDelaunay::Task &tk : d.tasks()) take(tk.node);
    for (const ExactDelaunay::Merge &mg : d.device_merges()) take(mg.node);
    d.finish();
  }
  const int32_t ntri = d.num_triangles();
  for (int32_t i = 0; i < ntri && i < cap; i++)
    for (int k = 0; k < 3; k++) tris[i * 3 + k] = d.triangles()[i * 3 + k];
  return ntri;
}

// micro-benchmark of the GPU side of the Delaunay stage: `njobs` copies of one prepared triangulation per
// launch sequence (sub-trees, then the merge levels up to device_top_points); returns microseconds per
// sequence (kernels only, HIP events), -1 on error
double vsm_debug_dc_bench(const int32_t *x, const int32_t *y, int32_t n, int32_t max_task_points, int32_t device_top_points,
                          int32_t device_kd, int32_t njobs, int32_t reps) {
  ExactDelaunay d;
  if (!d.prepare(x, y, n, max_task_points, nullptr, std::max(device_top_points, 0), device_kd != 0)) return -1;
  const DcMesh mesh = d.mesh();
  DcDeviceCopy dev;
  dev.device_kd = device_kd != 0;
  dev.block = device_top_points < 0;
  if (!dev.create(d, njobs)) return -1;
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0);
  (void)hipEventCreate(&e1);
  double total = 0;
  for (int r = 0; r < reps + 1; r++) {
    dev.load(mesh, nullptr);
    (void)hipEventRecord(e0, nullptr);
    dev.launch(nullptr);
    (void)hipEventRecord(e1, nullptr);
    if (hipDeviceSynchronize() != hipSuccess) return -1;
    float ms = 0;
    (void)hipEventElapsedTime(&ms, e0, e1);
    if (r > 0) total += ms * 1e3;
  }
  (void)hipEventDestroy(e0);
  (void)hipEventDestroy(e1);
  return total / reps;
}

// test hooks: which matches at shared pixels stand for their points, (index the radix-sorted keys carry, index
// Triangle's randomised vertex sort puts first) pairs - from the host emulation and from k_dc_ties; -1: not done
int32_t vsm_host_ties(const int32_t *x, const int32_t *y, int32_t n, int32_t *pairs, int32_t cap) {
  ExactDelaunay d;
  d.prepare(x, y, n, n, nullptr, 0, true, true);
  d.resolve_ties();
  const auto &t = d.ties();
  for (size_t k = 0; k < t.size() && (int32_t)k < cap; k++) {
    pairs[2 * k] = t[k].first;
    pairs[2 * k + 1] = t[k].second;
  }
  return (int32_t)t.size();
}

int32_t vsm_debug_ties_gpu(const int32_t *x, const int32_t *y, int32_t n, int32_t *pairs, int32_t cap, double *kernel_us) {
  ExactDelaunay d;
  d.prepare(x, y, n, n, nullptr, 0, true, true);
  VsmDcJob job;
  memset(&job, 0, sizeof(job));
  VsmDcJob *d_job = nullptr;
  uint64_t *d_keys = nullptr;
  int32_t *d_out = nullptr;
  std::vector<int32_t> out(1 + 2 * VSM_DC_TIE_PATCHES, 0);
  if (hipMalloc((void **)&d_keys, (size_t)std::max(n, 1) * 8) != hipSuccess || hipMalloc((void **)&d_out, out.size() * 4) != hipSuccess ||
      hipMalloc((void **)&d_job, sizeof(job)) != hipSuccess)
    return -2;
  (void)hipMemcpy(d_keys, d.tie_keys(), (size_t)n * 8, hipMemcpyHostToDevice);
  (void)hipMemset(d_out, 0, out.size() * 4);
  job.tie_keys = d_keys;
  job.tie_out = d_out;
  job.n_in = n;
  (void)hipMemcpy(d_job, &job, sizeof(job), hipMemcpyHostToDevice);
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0);
  (void)hipEventCreate(&e1);
  (void)hipEventRecord(e0, nullptr);
  vsm_dc_launch_ties(nullptr, d_job, 1);
  (void)hipEventRecord(e1, nullptr);
  const bool ok = hipDeviceSynchronize() == hipSuccess;
  float ms = 0;
  (void)hipEventElapsedTime(&ms, e0, e1);
  if (kernel_us) *kernel_us = ms * 1e3;
  (void)hipMemcpy(out.data(), d_out, out.size() * 4, hipMemcpyDeviceToHost);
  (void)hipEventDestroy(e0);
  (void)hipEventDestroy(e1);
  (void)hipFree(d_keys);
  (void)hipFree(d_out);
  (void)hipFree(d_job);
  if (!ok) return -2;
  for (int32_t k = 0; k < out[0] && k < cap; k++) {
    pairs[2 * k] = out[1 + 2 * k];
    pairs[2 * k + 1] = out[2 + 2 * k];
  }
  return out[0];
}

int32_t vsm_host_outliers_and_prior(const vsm_params *p, const vsm_p_match *list, int32_t n, int32_t method, vsm_p_match *out,
                                    int32_t cap, float *ranges, int32_t w, int32_t hh) {
  VsmHostWork wk;
  std::vector<vsm_p_match> m(list, list + std::max(n, 0));
  vsm_host_remove_outliers(wk, *p, m, method);
  if (ranges) {
    const int32_t dims[3] = {w, hh, w};
    std::vector<float> rg;
    vsm_host_prior_statistics(*p, dims, m, method, rg);
    ranges_to_device_layout(ranges, rg.data(), rg.size());
  }
  for (size_t i = 0; i < m.size() && (int32_t)i < cap; i++) out[i] = m[i];
  return (int32_t)m.size();
}

int32_t vsm_host_outliers_and_prior_threads(const vsm_params *p, const vsm_p_match *list, int32_t n, int32_t method, vsm_p_match *out,
                                            int32_t cap, float *ranges, int32_t w, int32_t hh, int32_t threads) {
  if (threads <= 1) return vsm_host_outliers_and_prior(p, list, n, method, out, cap, ranges, w, hh);
  VsmForkJoin fj(std::min(threads, 64));
  VsmPool apart(2);  // (as in a handle: Triangle's emulated vertex sort runs beside the triangulation of lists from 1024 matches)
  VsmHostWork wk;
  wk.pool = &fj;
  wk.async = &apart;
  std::vector<vsm_p_match> m;
  if (n > 3) {  // vsm_match's final stage (early_xy): pixels first, then flows, votes, survivors
    std::vector<uint32_t> xy((size_t)n);
    for (int32_t i = 0; i < n; i++) xy[i] = (uint32_t)(int32_t)list[i].u1c | ((uint32_t)(int32_t)list[i].v1c << 16);
    vsm_host_outliers_begin_xy(wk, xy.data(), n);
    wk.del.run(wk.x.data(), wk.y.data(), n, wk.pool, wk.async);
    vsm_host_outliers_begin_flows(wk, list, n, method);
    vsm_host_outliers_end(wk, *p, list, n, method, m);
  } else {
    m.assign(list, list + std::max(n, 0));  // the reference leaves short lists alone (viso/matcher.cpp:1210)
  }
  if (ranges) {
    const int32_t dims[3] = {w, hh, w};
    std::vector<float> rg;
    vsm_host_prior_statistics(*p, dims, m, method, rg);
    ranges_to_device_layout(ranges, rg.data(), rg.size());
  }
  for (size_t i = 0; i < m.size() && (int32_t)i < cap; i++) out[i] = m[i];
  return (int32_t)m.size();
}

int32_t vsm_debug_dc2(const vsm_params *p, const vsm_p_match *list, int32_t n, int32_t method, int32_t gpu_ties, int32_t copies,
                      vsm_p_match *out, int32_t cap, float *ranges, int32_t w, int32_t hh, double *kernel_us) {
  if (copies < 1) copies = 1;
  const int ub = (int)ceilf((float)w / (float)p->match_binsize), vb = (int)ceilf((float)hh / (float)p->match_binsize);
  if (ranges && ub * vb > 1024) return -2;
  Dc2Bank B;
  if (!B.reserve(copies, std::max(n, 64), true)) return -1;
  vsm_p_match *d_list = nullptr, *d_out = nullptr;
  int32_t *d_cnt = nullptr;
  float *d_ranges = nullptr;
  const size_t lb = (size_t)std::max(n, 1) * sizeof(vsm_p_match), rb = (size_t)ub * vb * 16 * 4;
  bool ok = hipMalloc((void **)&d_list, lb) == hipSuccess && hipMalloc((void **)&d_out, lb * copies) == hipSuccess &&
            hipMalloc((void **)&d_cnt, 4) == hipSuccess && hipMalloc((void **)&d_ranges, rb * copies) == hipSuccess;
  int32_t result = -1;
  if (ok) {
    (void)hipMemcpy(d_list, list, (size_t)n * sizeof(vsm_p_match), hipMemcpyHostToDevice);
    (void)hipMemcpy(d_cnt, &n, 4, hipMemcpyHostToDevice);
    for (int i = 0; i < copies; i++)
      B.fill_job(i, d_list, d_cnt, gpu_ties != 0, d_out + (size_t)i * std::max(n, 1), nullptr, d_ranges + (size_t)i * ub * vb * 16);
    (void)hipMemcpy(B.d_jobs, B.h_jobs, sizeof(VsmDc2Job) * copies, hipMemcpyHostToDevice);
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0);
    (void)hipEventCreate(&e1);
    (void)hipEventRecord(e0, nullptr);
    dc2_enqueue_triangulation(nullptr, B, copies, n, gpu_ties != 0);
    if (!gpu_ties) {  // the host's verdicts (the look-ahead path computes them while the device triangulates)
      ok = hipDeviceSynchronize() == hipSuccess;
      ExactDelaunay sorter;
      for (int i = 0; i < copies && ok; i++) {
        int32_t *o = B.host_ties_of(i);
        const int32_t nn = B.h_n[i];
        o[0] = nn > 3 ? sorter.sort_ties(B.host_keys(i), nn, o + 1, (B.ties_stride - 1) / 2) : 0;
      }
    }
    // (VSM_DC2_EXPECT_LONG=0, tests: a list of 8193..16384 matches through the narrow form inside the LDS kernel, as in a handle
    // that has not met a long list yet)
    const char *xl = getenv("VSM_DC2_EXPECT_LONG");
    dc2_enqueue_mesh(nullptr, B, copies, n, g_no_prof, false, nullptr, xl ? atoi(xl) != 0 : true);
    dc2_enqueue_votes(nullptr, B, copies, n, method, (float)p->outlier_flow_tolerance, (float)p->outlier_disp_tolerance);
    if (ranges) vsm_dc2_launch_prior(nullptr, B.d_jobs, copies, method, p->match_binsize, p->match_radius, w, hh, ub, vb);
    (void)hipEventRecord(e1, nullptr);
    ok = ok && hipDeviceSynchronize() == hipSuccess && hipGetLastError() == hipSuccess;
    float ms = 0;
    (void)hipEventElapsedTime(&ms, e0, e1);
    if (kernel_us) *kernel_us = ms * 1e3;
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
#ifdef VSM_DC2_DUMP_FILE  // debugging aid (tools/dc2_dump.py on a -DVSM_DC2_DUMP_FILE=\"path\" build): the first job's triangulation as the device left it
    {
      const char *dump = VSM_DC2_DUMP_FILE;  // debugging aid: the first job's triangulation as the device left it
      int32_t mn[2] = {0, 0};
      (void)hipMemcpy(mn, B.pp[0].mn, 8, hipMemcpyDeviceToHost);
      const int32_t mm = std::max(mn[0], 0);
      std::vector<int32_t> tri((size_t)mm * 16), idv(mm);
      std::vector<uint32_t> ptv(mm);
      (void)hipMemcpy(tri.data(), B.pp[0].tri, tri.size() * 4, hipMemcpyDeviceToHost);
      (void)hipMemcpy(ptv.data(), B.pp[0].pt, ptv.size() * 4, hipMemcpyDeviceToHost);
      (void)hipMemcpy(idv.data(), B.pp[0].id, idv.size() * 4, hipMemcpyDeviceToHost);
      if (FILE *f = fopen(dump, "wb")) {
        fwrite(mn, 4, 2, f);
        fwrite(tri.data(), 4, tri.size(), f);
        fwrite(ptv.data(), 4, ptv.size(), f);
        fwrite(idv.data(), 4, idv.size(), f);
        fclose(f);
      }
    }
#endif
    if (ok && *B.h_error) {
      result = -2;
    } else if (ok) {
      // every copy must have produced the same survivors
      std::vector<int32_t> cnt(copies);
      std::vector<vsm_p_match> first, other;
      bool same = true;
      for (int i = 0; i < copies && same; i++) {
        (void)hipMemcpy(&cnt[i], B.pp[i].out_count, 4, hipMemcpyDeviceToHost);
        same = cnt[i] == cnt[0] && cnt[i] >= 0 && cnt[i] <= n;
        if (!same) break;
        std::vector<vsm_p_match> &dst = i == 0 ? first : other;
        dst.resize(cnt[i]);
        if (cnt[i]) (void)hipMemcpy(dst.data(), d_out + (size_t)i * std::max(n, 1), (size_t)cnt[i] * sizeof(vsm_p_match), hipMemcpyDeviceToHost);
        if (i > 0) same = memcmp(first.data(), other.data(), (size_t)cnt[0] * sizeof(vsm_p_match)) == 0;
      }
      if (same) {
        result = cnt[0];
        for (int32_t i = 0; i < result && i < cap; i++) out[i] = first[i];
        if (ranges) (void)hipMemcpy(ranges, d_ranges + (size_t)(copies - 1) * ub * vb * 16, rb, hipMemcpyDeviceToHost);
      } else {
        result = -3;
      }
    }
  }
  (void)hipFree(d_list);
  (void)hipFree(d_out);
  (void)hipFree(d_cnt);
  (void)hipFree(d_ranges);
  B.release();
  return result;
}

void vsm_get_counters(vsm_handle *h, int64_t *out5) { memcpy(out5, h->counters, sizeof(h->counters)); }
void vsm_get_timings(vsm_handle *h, double *out5) { memcpy(out5, h->timings, sizeof(h->timings)); }

}  // extern "C"
