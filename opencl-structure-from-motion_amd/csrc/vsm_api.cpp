// C-ABI of libvisomatch.so (include/visomatch.h): host orchestration of the matcher path.
//
//   vsm_push_back : H2D (or D2D) ingest -> [halve] -> Sobel full / Sobel+blob+corner -> NMS ->
//                   ordered feature emission + descriptors -> bin sort            (all on the GPU)
//   vsm_match     : pass-1 match chain -> D2H -> host Delaunay support + prior boxes -> H2D ->
//                   pass-2 match chain -> refinement -> D2H -> host Delaunay support
//
// There is no CPU fallback: without a usable HIP device vsm_create() returns NULL.
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <chrono>
#include <vector>

#include "vsm_host.h"
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
static inline double now_us() {
  return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

struct vsm_handle {
  vsm_params param;  // match_radius already halved for half_resolution (viso/matcher.cpp:59-60)
  int device = 0;
  hipStream_t stream = nullptr;

  // device state
  bool allocated = false;
  VsmDims dims{};
  uint8_t *arena = nullptr;
  size_t arena_bytes = 0;
  VsmImage h_imgs[4];  // image id = frame_slot*2 + side
  VsmImage *d_imgs = nullptr;
  int16_t *f1 = nullptr, *f2 = nullptr;
  size_t f_stride = 0;
  VsmPair pair{};
  int32_t cap_set[2] = {0, 0};

  // pinned host memory.  hm_* is host-mapped: kernels write the feature counts and the
  // compacted match lists straight into it, so results need a stream sync but no D2H copy.
  int32_t *h_counts = nullptr;  // scratch [16]
  float *h_ranges = nullptr;
  uint8_t *hm_block = nullptr;
  int32_t *hm_counts = nullptr;  // [8] feature counts (image*2+set) + [2] list sizes
  vsm_p_match *hm_list1 = nullptr, *hm_list2 = nullptr;
  bool counts_pending = false;   // a push is in flight: sync before reading hm_counts
  int pending_slot = 0, pending_imgs = 0;

  // ring buffer state (Matcher's prev/curr pointers, viso/matcher.cpp:108-155)
  int cur = 0;
  bool have[2] = {false, false};   // frame slot holds a left image
  bool right[2] = {false, false};  // ... and a right image
  int32_t n_feat[2][2][2] = {};    // [slot][side][set]
  int32_t dims_p[3] = {0, 0, 0}, dims_c[3] = {0, 0, 0};
  bool f_valid = false;  // f1/f2 hold the responses of the current left image

  // results
  std::vector<vsm_p_match> stage[5];
  std::vector<vsm_p_match> matched;
  std::vector<float> ranges;
  std::vector<int32_t> pf;
  int capture_stage2 = 0;
  VsmHostWork work;
  int64_t counters[5] = {0, 0, 0, 0, 0};
  double timings[5] = {0, 0, 0, 0, 0};
  std::vector<uint8_t> gainI[2];
  VsmProf prof;
  VsmPool *pool = nullptr;
};

extern "C" {

const char *vsm_version(void) { return "visomatch 0.1 (gfx950)"; }

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
  {
    int nt = 8;  // host threads for the Delaunay sub-problems (the caller's thread included)
    if (const char *e = getenv("VSM_HOST_THREADS")) nt = atoi(e);
    unsigned hc = std::thread::hardware_concurrency();
    if (hc && (unsigned)nt > hc) nt = (int)hc;
    h->pool = new VsmPool(nt);
    h->work.pool = h->pool;
  }
  if (hipGetDevice(&h->device) != hipSuccess || hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking) != hipSuccess ||
      hipHostMalloc((void **)&h->h_counts, 16 * sizeof(int32_t), hipHostMallocDefault) != hipSuccess) {
    fprintf(stderr, "visomatch: HIP initialisation failed: %s\n", hipGetErrorString(hipGetLastError()));
    delete h;
    return nullptr;
  }
  return h;
}

static void release_device(vsm_handle *h) {
  if (h->arena) (void)hipFree(h->arena);
  if (h->h_ranges) (void)hipHostFree(h->h_ranges);
  if (h->hm_block) (void)hipHostFree(h->hm_block);
  h->arena = nullptr;
  h->h_ranges = nullptr;
  h->hm_block = nullptr;
  h->counts_pending = false;
  h->allocated = false;
  h->have[0] = h->have[1] = h->right[0] = h->right[1] = false;
  h->f_valid = false;
  memset(h->n_feat, 0, sizeof(h->n_feat));
}

void vsm_destroy(vsm_handle *h) {
  if (!h) return;
  (void)hipStreamSynchronize(h->stream);
  release_device(h);
  if (h->h_counts) (void)hipHostFree(h->h_counts);
  if (h->stream) (void)hipStreamDestroy(h->stream);
  delete h->pool;
  delete h;
}

void vsm_set_intrinsics(vsm_handle *h, double f, double cu, double cv, double base) {
  h->param.f = f;
  h->param.cu = cu;
  h->param.cv = cv;
  h->param.base = base;
}

static int nms_cells(int32_t len, int32_t n) {  // loop count of "for (i=n+margin; i<len-n-margin; i+=n+1)"
  int32_t span = len - 2 * n - 2 * VSM_MARGIN;
  return span > 0 ? (span + n) / (n + 1) : 0;
}

// One arena per handle: 4 image slots (prev/curr x left/right), the transient filter responses
// of one push and the buffers of one frame pair.  Zero-filled once, so row padding stays 0.
static int allocate(vsm_handle *h, int32_t w, int32_t hh) {
  if (h->allocated) {
    (void)hipStreamSynchronize(h->stream);
    release_device(h);
  }
  const vsm_params &p = h->param;
  VsmDims &d = h->dims;
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
  if (p.match_binsize < 1 || nb > (1 << 22)) {
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
    h->cap_set[k] = 4 * ncu[k] * ncv[k];
    if (h->cap_set[k] < 4) h->cap_set[k] = 4;
  }
  const size_t full = al256((size_t)d.bpl * d.h + 64), mres = al256((size_t)d.mbpl * d.mh + 64);
  // pass 1: size
  size_t off = 0;
  auto take = [&](size_t bytes) {
    size_t o = off;
    off += al256(bytes);
    return o;
  };
  struct SetOff {
    size_t feat, count, cand, cell_off, bin_start, bin_cnt, binid, s_idx, s_uv, s_desc, tmp;
  };
  struct ImgOff {
    size_t img, imgm, du, dv, duf, dvf;
    SetOff set[2];
  } io[4];
  for (int i = 0; i < 4; i++) {
    io[i].img = take(full);
    io[i].imgm = p.half_resolution ? take(mres) : io[i].img;
    io[i].du = take(mres);
    io[i].dv = take(mres);
    io[i].duf = p.half_resolution ? take(full) : io[i].du;
    io[i].dvf = p.half_resolution ? take(full) : io[i].dv;
    for (int k = 0; k < 2; k++) {
      const size_t cap = h->cap_set[k];
      io[i].set[k].feat = take(cap * 48);
      io[i].set[k].count = take(4);
      io[i].set[k].cand = take((size_t)(ncu[k] * ncv[k] + 1) * 16);
      io[i].set[k].cell_off = take((size_t)(ncu[k] * ncv[k] + 2) * 4);
      io[i].set[k].bin_start = take((size_t)(nb + 1) * 4);
      io[i].set[k].bin_cnt = take((size_t)(nb + 1) * 4);
      io[i].set[k].binid = take(cap * 4);
      io[i].set[k].s_idx = take(cap * 4);
      io[i].set[k].s_uv = take(cap * 8);
      io[i].set[k].s_desc = take(cap * 32);
      io[i].set[k].tmp = take(cap * 4);
    }
  }
  h->f_stride = mres / 2 * 2;  // elements per image plane (int16), keeps 256-byte alignment
  const size_t o_f1 = take(h->f_stride * 2 * 2), o_f2 = take(h->f_stride * 2 * 2);
  const size_t qcap = (size_t)(h->cap_set[0] > h->cap_set[1] ? h->cap_set[0] : h->cap_set[1]);
  const size_t o_raw = take(qcap * 48), o_flag = take(qcap * 4), o_l1 = take(qcap * 48), o_l2 = take(qcap * 48);
  const size_t o_bc = take((qcap / 256 + 2) * 4);
  const size_t o_cnt = take(16), o_rng = take((size_t)d.ub * d.vb * 64);
  const size_t o_pf = p.refinement == 2 ? take(qcap * 3 * 12 * 4) : 0;
  const size_t o_imgs = take(4 * sizeof(VsmImage));
  h->arena_bytes = off;
  HIPCHK(hipMalloc((void **)&h->arena, h->arena_bytes));
  HIPCHK(hipMemsetAsync(h->arena, 0, h->arena_bytes, h->stream));
  uint8_t *b = h->arena;
  for (int i = 0; i < 4; i++) {
    VsmImage &im = h->h_imgs[i];
    im.img = b + io[i].img;
    im.imgm = b + io[i].imgm;
    im.du = b + io[i].du;
    im.dv = b + io[i].dv;
    im.du_full = b + io[i].duf;
    im.dv_full = b + io[i].dvf;
    for (int k = 0; k < 2; k++) {
      VsmSet &s = im.set[k];
      s.feat = (int32_t *)(b + io[i].set[k].feat);
      s.count = (int32_t *)(b + io[i].set[k].count);
      s.cand = (int32_t *)(b + io[i].set[k].cand);
      s.cell_off = (int32_t *)(b + io[i].set[k].cell_off);
      s.bin_start = (int32_t *)(b + io[i].set[k].bin_start);
      s.bin_cnt = (int32_t *)(b + io[i].set[k].bin_cnt);
      s.binid = (int32_t *)(b + io[i].set[k].binid);
      s.s_idx = (int32_t *)(b + io[i].set[k].s_idx);
      s.s_uv = (int2 *)(b + io[i].set[k].s_uv);
      s.s_desc = (uint4 *)(b + io[i].set[k].s_desc);
      s.tmp = (int32_t *)(b + io[i].set[k].tmp);
      s.cap = h->cap_set[k];
      s.nms_n = nn[k];
      s.ncu = ncu[k];
      s.ncv = ncv[k];
    }
  }
  h->f1 = (int16_t *)(b + o_f1);
  h->f2 = (int16_t *)(b + o_f2);
  h->pair.raw = (vsm_p_match *)(b + o_raw);
  h->pair.flag = (int32_t *)(b + o_flag);
  h->pair.blockcnt = (int32_t *)(b + o_bc);
  h->pair.list1 = (vsm_p_match *)(b + o_l1);
  h->pair.list2 = (vsm_p_match *)(b + o_l2);
  h->pair.count = (int32_t *)(b + o_cnt);
  h->pair.ranges = (float *)(b + o_rng);
  h->pair.pf = p.refinement == 2 ? (int32_t *)(b + o_pf) : nullptr;
  h->d_imgs = (VsmImage *)(b + o_imgs);
  {  // host-mapped result block: [counts 256 B][list1][list2]
    const size_t l1 = al256((size_t)h->cap_set[0] * 48), l2 = al256(qcap * 48);
    HIPCHK(hipHostMalloc((void **)&h->hm_block, 256 + l1 + l2, hipHostMallocMapped));
    memset(h->hm_block, 0, 256 + l1 + l2);
    uint8_t *dblock = nullptr;
    HIPCHK(hipHostGetDevicePointer((void **)&dblock, h->hm_block, 0));
    h->hm_counts = (int32_t *)h->hm_block;
    h->hm_list1 = (vsm_p_match *)(h->hm_block + 256);
    h->hm_list2 = (vsm_p_match *)(h->hm_block + 256 + l1);
    for (int i = 0; i < 4; i++)
      for (int k = 0; k < 2; k++) h->h_imgs[i].set[k].count_host = (int32_t *)dblock + (i * 2 + k);
    h->pair.hcount = (int32_t *)dblock + 8;
    h->pair.hlist1 = (vsm_p_match *)(dblock + 256);
    h->pair.hlist2 = (vsm_p_match *)(dblock + 256 + l1);
  }
  HIPCHK(hipMemcpyAsync(h->d_imgs, h->h_imgs, 4 * sizeof(VsmImage), hipMemcpyHostToDevice, h->stream));
  HIPCHK(hipHostMalloc((void **)&h->h_ranges, (size_t)d.ub * d.vb * 64, hipHostMallocDefault));
  HIPCHK(hipStreamSynchronize(h->stream));
  h->allocated = true;
  return VSM_OK;
}

// completes an asynchronous push: waits for the stream and takes the feature counts the
// kernels wrote into host-mapped memory
static int settle(vsm_handle *h) {
  if (!h->counts_pending) return VSM_OK;
  HIPCHK(hipStreamSynchronize(h->stream));
  HIPCHK(hipGetLastError());
  h->prof.resolve();
  const int slot = h->pending_slot;
  memset(h->n_feat[slot], 0, sizeof(h->n_feat[slot]));
  for (int k = 0; k < h->pending_imgs; k++)
    for (int s2 = 0; s2 < 2; s2++) h->n_feat[slot][k][s2] = h->hm_counts[(slot * 2 + k) * 2 + s2];
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
  if (!h->allocated || w != h->dims.w || hh != h->dims.h) {
    // a change of image size restarts the ring buffer (the reference would match across sizes)
    int rc = allocate(h, w, hh);
    if (rc != VSM_OK) return rc;
  }
  if (!replace) {  // viso/matcher.cpp:123-155: curr becomes prev
    h->cur ^= 1;
    memcpy(h->dims_p, h->dims_c, sizeof(h->dims_p));
  }
  const int slot = h->cur;
  h->dims_c[0] = w;
  h->dims_c[1] = hh;
  h->dims_c[2] = h->dims.bpl;
  const int n_img = I2 ? 2 : 1;
  if (h->counts_pending) {  // a previous asynchronous push has not been settled yet
    int rc = settle(h);
    if (rc != VSM_OK) return rc;
  }
  uint8_t *dst0 = h->h_imgs[slot * 2].img, *dst1 = h->h_imgs[slot * 2 + 1].img;
  if (on_device) {
    vsm_launch_ingest(h->stream, h->prof, I1, I2, bpl, dst0, dst1, h->dims);
  } else {  // pageable source: HIP stages the rows
    HIPCHK(hipMemcpy2DAsync(dst0, h->dims.bpl, I1, bpl, w, hh, hipMemcpyHostToDevice, h->stream));
    if (I2) HIPCHK(hipMemcpy2DAsync(dst1, h->dims.bpl, I2, bpl, w, hh, hipMemcpyHostToDevice, h->stream));
  }
  vsm_launch_features(h->stream, h->prof, h->d_imgs, slot * 2, n_img, h->dims, h->f1, h->f2, h->f_stride, h->param.nms_tau,
                      h->param.multi_stage, h->param.half_resolution, h->param.match_binsize, h->h_imgs);
  HIPCHK(hipGetLastError());
  h->have[slot] = true;
  h->right[slot] = (I2 != nullptr);
  h->pending_slot = slot;
  h->pending_imgs = n_img;
  h->counts_pending = true;
  h->f_valid = true;
  h->gainI[0].clear();
  h->gainI[1].clear();
  // Host images: the caller may free or overwrite them as soon as we return (matcherMex does),
  // so the transfer must have completed.  Device-resident images: the push stays asynchronous and
  // is settled by the next call that needs its results (the source must stay valid until then).
  return on_device ? VSM_OK : settle(h);
}

int vsm_push_back(vsm_handle *h, const uint8_t *I1, const uint8_t *I2, int32_t w, int32_t hh, int32_t bpl, int replace) {
  return push_common(h, I1, I2, w, hh, bpl, replace, false);
}

int vsm_push_back_device(vsm_handle *h, const uint8_t *dI1, const uint8_t *dI2, int32_t w, int32_t hh, int32_t bpl,
                         int replace) {
  return push_common(h, dI1, dI2, w, hh, bpl, replace, true);
}

int vsm_match(vsm_handle *h, int32_t method, const double *Tr) {
  if (!h->allocated) return VSM_ENOTREADY;
  HIPCHK(hipSetDevice(h->device));
  {
    int rc = settle(h);
    if (rc != VSM_OK) return rc;
  }
  const vsm_params &p = h->param;
  const int sc = h->cur, sp = h->cur ^ 1;
  auto N = [&](int slot, int side, int set) { return h->have[slot] ? h->n_feat[slot][side][set] : 0; };
  // sanity checks of viso/matcher.cpp:190-212 (a NULL set and an empty set are both "count 0" here)
  if (method == 0) {
    if (N(sp, 0, 1) == 0 || N(sc, 0, 1) == 0) return VSM_ENOTREADY;
    if (p.multi_stage && (N(sp, 0, 0) == 0 || N(sc, 0, 0) == 0)) return VSM_ENOTREADY;
  } else if (method == 1) {
    if (N(sc, 0, 1) == 0 || N(sc, 1, 1) == 0) return VSM_ENOTREADY;
    if (p.multi_stage && (N(sc, 0, 0) == 0 || N(sc, 1, 0) == 0)) return VSM_ENOTREADY;
  } else {
    if (N(sp, 0, 1) == 0 || N(sp, 1, 1) == 0 || N(sc, 0, 1) == 0 || N(sc, 1, 1) == 0) return VSM_ENOTREADY;
    if (p.multi_stage && (N(sp, 0, 0) == 0 || N(sp, 1, 0) == 0 || N(sc, 0, 0) == 0 || N(sc, 1, 0) == 0))
      return VSM_ENOTREADY;
  }
  const double t0 = now_us();
  for (int s = 0; s < 5; s++) h->stage[s].clear();
  h->matched.clear();
  memset(h->counters, 0, sizeof(h->counters));

  VsmMatchCfg cfg;
  memset(&cfg, 0, sizeof(cfg));
  cfg.method = method;
  cfg.binsize = p.match_binsize;
  cfg.radius = p.match_radius;
  cfg.disp_tol = p.match_disp_tolerance;
  cfg.f = p.f;
  cfg.cu = p.cu;
  cfg.cv = p.cv;
  cfg.base = p.base;
  cfg.use_tr = Tr ? 1 : 0;
  if (Tr) memcpy(cfg.t, Tr, 12 * sizeof(double));
  // stereo matching only needs the current pair: point "prev" at the current slot so that no
  // pointer is dangling; it is never dereferenced for method 1
  const int img_prev = (method == 1 ? sc : sp) * 2, img_curr = sc * 2;
  const int stages = method == 2 ? 4 : 2;
  const int qslot = method == 2 ? sp : sc;
  VsmDims dp = h->dims, dc = h->dims;  // one size per handle

  double t1 = t0, t2 = t0;
  if (p.multi_stage) {
    cfg.sparse = 1;
    cfg.use_prior = 0;
    const int nq = N(qslot, 0, 0);
    vsm_launch_match(h->stream, h->prof, h->d_imgs, img_prev, img_curr, h->pair, h->dims, cfg, nq, 0);
    vsm_launch_export(h->stream, h->prof, h->pair, 0, nq);
    HIPCHK(hipStreamSynchronize(h->stream));  // the list was written into host-mapped memory
    h->stage[0].assign(h->hm_list1, h->hm_list1 + h->hm_counts[8]);
    h->counters[0] += (int64_t)nq * stages;
    t1 = now_us();
    h->stage[1] = h->stage[0];
    vsm_host_remove_outliers(h->work, p, h->stage[1], method);
    vsm_host_prior_statistics(p, h->dims_c, h->stage[1], method, h->ranges);
    memcpy(h->h_ranges, h->ranges.data(), h->ranges.size() * sizeof(float));
    HIPCHK(hipMemcpyAsync(h->pair.ranges, h->h_ranges, h->ranges.size() * sizeof(float), hipMemcpyHostToDevice, h->stream));
    t2 = now_us();
  }
  cfg.sparse = 0;
  cfg.use_prior = p.multi_stage ? 1 : 0;
  const int nq2 = N(qslot, 0, 1);
  vsm_launch_match(h->stream, h->prof, h->d_imgs, img_prev, img_curr, h->pair, h->dims, cfg, nq2, 1);
  h->counters[0] += (int64_t)nq2 * stages;
  // The list size is still on the device: the refinement / export grids are sized for the worst
  // case (every query matched) and surplus threads exit at once.
  if (h->capture_stage2 && p.refinement == 1)  // debug view: keep the unrefined list (raw is free again)
    HIPCHK(hipMemcpyAsync(h->pair.raw, h->pair.list2, (size_t)nq2 * sizeof(vsm_p_match), hipMemcpyDeviceToDevice, h->stream));
  if (p.refinement > 0)
    vsm_launch_refine(h->stream, h->prof, h->d_imgs, img_prev, img_curr, h->pair, dp, dc, method, p.refinement, nq2,
                      h->pair.count + 1);
  vsm_launch_export(h->stream, h->prof, h->pair, 1, nq2);
  HIPCHK(hipStreamSynchronize(h->stream));
  HIPCHK(hipGetLastError());
  const int32_t n2 = h->hm_counts[9];
  if (p.refinement == 1) {
    h->stage[3].assign(h->hm_list2, h->hm_list2 + n2);
    if (h->capture_stage2) {
      h->stage[2].resize(n2);
      if (n2) HIPCHK(hipMemcpy(h->stage[2].data(), h->pair.raw, (size_t)n2 * sizeof(vsm_p_match), hipMemcpyDeviceToHost));
    }
  } else {
    h->stage[2].assign(h->hm_list2, h->hm_list2 + n2);
    if (p.refinement == 2) {
      const size_t n = h->stage[2].size();
      h->pf.resize(n * 36);
      if (n) HIPCHK(hipMemcpy(h->pf.data(), h->pair.pf, n * 36 * sizeof(int32_t), hipMemcpyDeviceToHost));
      h->stage[3].clear();
      for (size_t i = 0; i < n; i++) {  // viso/matcher.cpp:1541-1581: a failed fit drops the match
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
  if (!h->allocated || !h->have[sp] || !h->have[sc] || h->matched.empty() || n == 0 || settle(h) != VSM_OK) return 1;
  const size_t bytes = (size_t)h->dims.bpl * h->dims.h;
  for (int k = 0; k < 2; k++) {  // left images come back from HBM on first use
    if (h->gainI[k].size() != bytes) {
      h->gainI[k].resize(bytes);
      if (hipMemcpy(h->gainI[k].data(), h->h_imgs[(k == 0 ? sp : sc) * 2].img, bytes, hipMemcpyDeviceToHost) != hipSuccess)
        return 1;
    }
  }
  return vsm_host_gain(h->gainI[0].data(), h->gainI[1].data(), h->dims_p, h->dims_c, h->matched, inliers, n);
}

// ---- stage-level views ----
static bool which_set(vsm_handle *h, int32_t which, int &img, int &set, int32_t &n) {
  if (which < 0 || which > 7 || !h->allocated) return false;
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
  if (n > 0 && hipMemcpy(out, h->h_imgs[img].set[set].feat, (size_t)n * 48, hipMemcpyDeviceToHost) != hipSuccess) return 0;
  return n;
}

int32_t vsm_stage_size(vsm_handle *h, int32_t s) { return (s >= 0 && s < 5) ? (int32_t)h->stage[s].size() : 0; }

int32_t vsm_stage_get(vsm_handle *h, int32_t s, vsm_p_match *out, int32_t cap) {
  if (s < 0 || s >= 5) return 0;
  int32_t n = (int32_t)h->stage[s].size();
  if (n > cap) n = cap;
  if (n > 0) memcpy(out, h->stage[s].data(), (size_t)n * sizeof(vsm_p_match));
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
  if (!h->allocated || which < 0 || which > 3 || settle(h) != VSM_OK) return 0;
  const int slot = which < 2 ? (h->cur ^ 1) : h->cur, side = which & 1;
  if (!h->have[slot] || (side && !h->right[slot])) return 0;
  if (full && !h->param.half_resolution) return 0;
  const VsmImage &im = h->h_imgs[slot * 2 + side];
  const int32_t bytes = full ? h->dims.bpl * h->dims.h : h->dims.mbpl * h->dims.mh;
  if (du && hipMemcpy(du, full ? im.du_full : im.du, bytes, hipMemcpyDeviceToHost) != hipSuccess) return 0;
  if (dv && hipMemcpy(dv, full ? im.dv_full : im.dv, bytes, hipMemcpyDeviceToHost) != hipSuccess) return 0;
  return bytes;
}

int32_t vsm_get_filter_responses(vsm_handle *h, int16_t *f1, int16_t *f2) {
  if (!h->allocated || !h->f_valid || settle(h) != VSM_OK) return 0;
  const int32_t n = h->dims.mbpl * h->dims.mh;
  if (f1 && hipMemcpy(f1, h->f1, (size_t)n * 2, hipMemcpyDeviceToHost) != hipSuccess) return 0;
  if (f2 && hipMemcpy(f2, h->f2, (size_t)n * 2, hipMemcpyDeviceToHost) != hipSuccess) return 0;
  return n;
}

void vsm_set_stage_capture(vsm_handle *h, int on) { h->capture_stage2 = on ? 1 : 0; }

static const char *kKernelNames[VSM_K_COUNT] = {"k_ingest", "k_halve", "k_filters<true>", "k_filters<false>", "k_nms", "k_scan_cells", "k_emit",
                                                "k_bin_scan", "k_bin_scatter", "k_bin_rank", "k_match<16>:pass1", "k_compact_matches:pass1", "k_match<16>:pass2",
                                                "k_compact_matches:pass2", "k_refine", "k_export_list"};

void vsm_set_profiling(vsm_handle *h, int on) {
  h->prof.on = on != 0;
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
  VsmPool pool(threads);
  d.run(x, y, n, threads > 1 ? &pool : nullptr);
  const int32_t nt = d.num_triangles();
  for (int32_t i = 0; i < nt && i < cap; i++)
    for (int k = 0; k < 3; k++) tris[i * 3 + k] = d.triangles()[i * 3 + k];
  return nt;
}

void vsm_get_counters(vsm_handle *h, int64_t *out5) { memcpy(out5, h->counters, sizeof(h->counters)); }
void vsm_get_timings(vsm_handle *h, double *out5) { memcpy(out5, h->timings, sizeof(h->timings)); }

}  // extern "C"
