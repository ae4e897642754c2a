// Internal structures shared by the host orchestration (vsm_api.cpp) and the gfx950 kernels
// (vsm_kernels.hip).  Nothing here is part of the C-ABI (include/visomatch.h).
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include <mutex>
#include <vector>

#include "visomatch.h"

#define VSM_MARGIN 6  // Matcher::margin = 5+1, viso/matcher.cpp:56

// One feature set (sparse = pass 1, dense = pass 2) of one image, resident in HBM.
//   feat      T1 records int32[12] in NMS emission order (what m1c1.. hold in the reference)
//   s_*       the same features regrouped by (class, u_bin, v_bin) bin for the radius search:
//             coordinates and 32-byte descriptors are contiguous per bin so a lane group reads
//             a bin as one coalesced run; inside a bin entries ascend by original index, so
//             "position in the sorted arrays" is exactly findMatch's traversal order
//             (viso/matcher.cpp:937-940) for a fixed class.
struct VsmSet {
  int32_t *feat;       // [cap][12]
  int32_t *count;      // device scalar
  int32_t *count_host; // the same number in host-mapped pinned memory (read after a stream sync)
  int32_t *cand;       // [ncu*ncv*4] packed NMS survivors: u | v<<14 | valid<<31
  int32_t *cell_off;   // [ncu*ncv+1] exclusive prefix of survivors per cell (emission order)
  int32_t *bin_start;  // [4*ub*vb*VSM_VSUB + 1] start of every fine bin (class, u-bin, v sub-row) in the sorted arrays
  int32_t *bin_cnt;    // [4*ub*vb*VSM_VSUB] histogram, then scatter cursor
  int32_t *binid;      // [cap] fine bin of every feature (emission order)
  int32_t *s_rank;     // [cap] by sorted position: place in the reference's (u_bin, v_bin, index) visiting order
  int32_t *s_idx;      // [cap] sorted position -> original index
  uint32_t *s_uv;      // [cap] u | v << 16 (coordinates < 16384), bin-sorted; 16-byte loads may over-read <= 3 entries
  uint4 *s_desc;       // [cap][2]
  uint4 *heads;        // [4*ub*vb*VSM_VSUB][4] head record of every fine bin: its start, then the coordinates of the 15 candidates from there on (dense set; else null)
  int32_t *tmp;        // [cap] scratch
  int32_t cap, nms_n, ncu, ncv;
};

struct VsmImage {
  uint8_t *img;      // [h][bpl]  padded copy of the input (pad = 0)
  uint8_t *imgm;     // matching-resolution image: half image, or == img
  uint8_t *du, *dv;  // matching-resolution Sobel responses
  uint8_t *du_full, *dv_full;  // full-resolution Sobel responses when the matching resolution IS the full one (== du, dv); else null
  uint8_t *duv_tiled;          // half_resolution: both full-resolution responses in 8 x 8 tiles (vsm_tiled_at in vsm_kernels.hip); else null
  VsmSet set[2];     // 0 sparse, 1 dense
};

// The match bins of the reference (match_binsize, "speed only") fix the order in which findMatch
// visits candidates, i.e. who wins a cost tie.  The sorted arrays here use a finer key -- every
// v-bin is cut into VSM_VSUB sub-rows -- so a search window touches far fewer candidates, and each
// candidate carries its place in the reference's order (s_rank) to settle ties identically.
#define VSM_VSUB 5

struct VsmDims {
  int32_t w, h, bpl;     // full resolution
  int32_t mw, mh, mbpl;  // matching resolution
  int32_t scale;         // 2 if half_resolution else 1
  int32_t ub, vb;        // match bins over the full-resolution image
};

struct VsmMatchCfg {  // common to all pairs of a launch
  int32_t method, use_prior, sparse;
  int32_t binsize, radius, disp_tol;
  int32_t heads, pad_;  // the second pass reads the per-bin head records (k_feat_heads) instead of bin starts + coordinate runs
  uint32_t bin_magic;  // ceil(2^32 / binsize): x / binsize == mulhi(x, bin_magic) for 0 <= x < 2^32 / binsize (binsize >= 2)
  double f, cu, cv, base;
};

// one frame pair of a (possibly batched) launch
struct VsmJob {
  int32_t img_prev, img_curr;  // left image ids; the right images are +1
  int32_t nq[2];               // queries of the sparse / dense pass
  int32_t use_tr, pad;
  double t[12];                // rows 0..2 of Tr_delta (viso/matcher.cpp:989-1002)
};

// per frame-pair buffers
struct VsmPair {
  vsm_p_match *raw;    // [cap_query] per-query result of the match chain
  int32_t *flag;       // [cap_query]
  int32_t *blockcnt;   // [cap_query/256 + 1] survivors per 256-query block
  vsm_p_match *list1;  // compacted pass-1 list
  vsm_p_match *list2;  // compacted pass-2 list (unrefined: refinement writes to hlist2)
  int32_t *count;      // [2] list sizes
  vsm_p_match *hlist1, *hlist2;  // host-mapped pinned copies the kernels write directly (no D2H copy)
  int32_t *hcount;               // [2] host-mapped list sizes
  float *ranges;       // [ub*vb][16]
  int32_t *pf;         // refinement==2 scratch: [cap][3][12] {status,du,dv,c0..c8}
};

// Optional per-kernel timing with HIP events recorded on the handle's own stream (bench.py's
// roofline leg).  Off by default: events cost a few microseconds per launch.
enum VsmKernelId {
  VSM_K_INGEST = 0, VSM_K_HALVE, VSM_K_SOBEL_FULL, VSM_K_FILTERS, VSM_K_NMS, VSM_K_NMS_SPARSE, VSM_K_SCAN, VSM_K_EMIT, VSM_K_BINSCAN, VSM_K_BINSCATTER, VSM_K_BINRANK,
  VSM_K_MATCH1, VSM_K_COMPACT1, VSM_K_MATCH2, VSM_K_COMPACT2, VSM_K_REFINE, VSM_K_EXPORT,
  VSM_K_FRONT,  // fused ingest + half-resolution image + full-resolution Sobel (half_resolution = 1)
  // the exact Delaunay stage of the look-ahead forms (their own streams; launched from the caller's and from pool threads)
  VSM_K_DC_KEYS, VSM_K_DC_TIES, VSM_K_DC_KD, VSM_K_DC_BLOCK, VSM_K_DC_MERGE, VSM_K_DC_SUPPORT, VSM_K_DC_COMPACT, VSM_K_DC_PRIOR,
  // the fused matching-resolution image side (filters + suppression out of one LDS tile; vsm_feat.h)
  VSM_K_FEAT_DENSE, VSM_K_FEAT_SPARSE,
  // feature records + bin-sorted copy in two kernels (k_feat_scan, k_feat_order)
  VSM_K_FEAT_SCAN, VSM_K_FEAT_ORDER,
  VSM_K_COUNT
};
struct VsmProf {
  bool on = false;
  std::mutex mu;  // begin() .. end() of one launch hold it: spans come from several threads
  std::vector<hipEvent_t> pool;
  size_t used = 0;
  struct Span { int id; hipEvent_t a, b; };
  std::vector<Span> open;
  double total_ms[VSM_K_COUNT] = {};
  int64_t launches[VSM_K_COUNT] = {};
  hipEvent_t get() {
    if (used == pool.size()) {
      hipEvent_t e;
      (void)hipEventCreate(&e);
      pool.push_back(e);
    }
    return pool[used++];
  }
  bool print_spans = false;  // (tools/span_probe.py: every launch's span on stderr; vsm_set_profiling(h, 1100 + id))
  int only = -1;  // >= 0: spans of this kernel id only (two event records per launch of ONE kernel instead of two per launch of
                  // every kernel on every stream: event records are packets too, and a queue that is never empty slows
                  // whatever runs beside it - tools/l2_invalidate_probe.py)
  static bool &skipping() {
    static thread_local bool v = false;
    return v;
  }
  void begin(int id, hipStream_t s) {
    if (!on) return;
    skipping() = only >= 0 && id != only;
    if (skipping()) return;
    mu.lock();
    Span sp{id, get(), get()};
    (void)hipEventRecord(sp.a, s);
    open.push_back(sp);
  }
  void end(hipStream_t s) {
    if (!on || skipping()) return;
    (void)hipEventRecord(open.back().b, s);
    mu.unlock();
  }
  void resolve() {  // call after every stream that carries spans has been synchronised
    std::lock_guard<std::mutex> lk(mu);
    for (const Span &sp : open) {
      float ms = 0;
      if (hipEventElapsedTime(&ms, sp.a, sp.b) == hipSuccess) {
        if (only >= 0 && print_spans) fprintf(stderr, "  span kernel %d: %.1f us\n", sp.id, ms * 1e3);
        total_ms[sp.id] += ms;
        launches[sp.id]++;
      }
    }
    open.clear();
    used = 0;
  }
};

// ---- launchers (vsm_kernels.hip) ----
// small table in pinned (hipHostMalloc) memory -> HBM by a kernel on stream s (see k_upload)
hipError_t vsm_upload(hipStream_t s, void *dst_device, const void *src_pinned, size_t bytes);
// Device memory of the library's large blocks (vsm_api.cpp): a released block of 8 MB or more goes to a process-wide cache
// instead of back to the driver, and the next request of about its size takes it from there.
hipError_t vsm_dev_alloc(void **p, size_t bytes);
void vsm_dev_free(void *p);
void vsm_launch_ingest(hipStream_t s, VsmProf &pf, const VsmImage *d_imgs, int first, const uint8_t *src0,
                       const uint8_t *src1, size_t frame_stride, int32_t src_bpl, int n_frames, const VsmDims &d);
// fused: bit 0 = the fused filter + suppression tiles where the radii allow, bit 1 = they also write f1 / f2 (debug getter),
// bit 2 = NOT the two-kernel records + bin order (k_feat_scan / k_feat_order) but k_scan_cells / k_emit / k_bin_*
// half_resolution = 1 only: caller image(s) -> [padded copy if write_img], half-resolution image, full-resolution Sobel planes
// in one pass; vsm_launch_features(front_done = 1) then skips its own halving and full-resolution Sobel
void vsm_launch_front(hipStream_t s, VsmProf &pf, const VsmImage *d_imgs, int first, const uint8_t *src0, const uint8_t *src1,
                      size_t frame_stride, int32_t src_bpl, int n_frames, const VsmDims &d, int write_img);
// returns 1 if f1 / f2 hold the launch's filter responses afterwards (unfused kernels, or fused bit 1)
int vsm_launch_features(hipStream_t s, VsmProf &pf, const VsmImage *d_imgs, int first, int n_img, const VsmDims &d,
                        int16_t *f1, int16_t *f2, size_t f_stride, int nms_tau, int multi_stage, int half_res,
                        int binsize, const VsmImage *h_imgs, int front_done = 0, int fused = 1);
// fuse_export (quad matching only; the return value says whether it happened): 1 - the compacted list also goes to its host-mapped
// copy (what vsm_launch_export does), 2 - pair 0's pixels go to xy_dst (what vsm_launch_export_xy does)
bool vsm_launch_match(hipStream_t s, VsmProf &pf, const VsmImage *d_imgs, const VsmPair *d_pairs, const VsmJob *d_jobs,
                      const VsmJob &job0, int npairs, const VsmDims &d, const VsmMatchCfg &cfg, int max_nq, int fuse_export = 0, uint32_t *xy_dst = nullptr);
void vsm_launch_export(hipStream_t s, VsmProf &pf, const VsmPair *d_pairs, int npairs, int pass, int n_upper);
void vsm_launch_export_xy(hipStream_t s, const VsmPair *d_pairs, uint32_t *dst_host_mapped, int n_upper);  // pair 0's pass-2 pixels, x | y << 16
#define VSM_PARA_MAX_LIST 16384  // matches per pair the batched tail of refinement==2 takes
void vsm_launch_parabolic_apply(hipStream_t s, const VsmPair *d_pairs, int npairs);
void vsm_launch_refine(hipStream_t s, VsmProf &pf, const VsmImage *d_imgs, const VsmPair *d_pairs, const VsmJob *d_jobs,
                       const VsmJob &job0, int npairs, const VsmDims &dp, const VsmDims &dc, int method, int refinement,
                       int n_upper);
