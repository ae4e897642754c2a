// TEST INFRASTRUCTURE ONLY -- never linked into the product library.
//
// C-ABI harness around the *real* reference matcher (libviso2 as vendored in
// dphoyes/OpenCL-Structure-from-Motion).  The reference sources are compiled
// where they lie under /root/reference/viso by oracle/Makefile (target
// _ref/libvisoref.so); nothing from the reference is copied into this repo.
//
// The harness reaches private stages of `class Matcher` (viso/matcher.h:138-246)
// through the usual access-specifier macro trick so that stage-level goldens
// (features, raw matching output, ranges, refinement, outlier removal) can be
// captured without editing the reference.
//
// Row padding: the reference never initialises the pad columns of its
// 16-byte-aligned image copies (viso/matcher.cpp:163-175, :639-645), and sparse
// descriptors read them.  oracle/Makefile links this
// file with -Wl,--wrap=posix_memalign; the wrapper below zero-fills every
// allocation, which pins pad = 0 (the definition the HIP path and the C
// restatement use as well).

#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <math.h>
#include <vector>
#include <array>
#include <iostream>
#include <algorithm>
#include <iterator>
#include <random>

#define private public
#define protected public
#include "matcher.h"
#include "viso_stereo.h"
#include "viso_mono.h"
#undef private
#undef protected
#include "filter.h"
#include "triangle.h"

extern "C" int __real_posix_memalign(void **memptr, size_t alignment, size_t size);
extern "C" int __wrap_posix_memalign(void **memptr, size_t alignment, size_t size) {
  int rc = __real_posix_memalign(memptr, alignment, size);
  if (rc == 0 && *memptr) memset(*memptr, 0, size);
  return rc;
}

namespace {

struct RefMatcher {
  Matcher *m;
  // stage captures of the last staged match (viso/matcher.cpp:183-241 split open)
  std::vector<Matcher::p_match> stage[6];
  explicit RefMatcher(const Matcher::parameters &p) : m(new Matcher(p)) {}
  ~RefMatcher() { delete m; }
};

Matcher::parameters make_params(const int32_t *ip, const double *dp) {
  Matcher::parameters p;
  p.nms_n = ip[0];
  p.nms_tau = ip[1];
  p.match_binsize = ip[2];
  p.match_radius = ip[3];
  p.match_disp_tolerance = ip[4];
  p.outlier_disp_tolerance = ip[5];
  p.outlier_flow_tolerance = ip[6];
  p.multi_stage = ip[7];
  p.half_resolution = ip[8];
  p.refinement = ip[9];
  p.f = dp[0];
  p.cu = dp[1];
  p.cv = dp[2];
  p.base = dp[3];
  return p;
}

void feature_set(Matcher *m, int which, int32_t *&ptr, int32_t &n) {
  switch (which) {
    case 0: ptr = m->m1p1; n = m->n1p1; break;
    case 1: ptr = m->m2p1; n = m->n2p1; break;
    case 2: ptr = m->m1c1; n = m->n1c1; break;
    case 3: ptr = m->m2c1; n = m->n2c1; break;
    case 4: ptr = m->m1p2; n = m->n1p2; break;
    case 5: ptr = m->m2p2; n = m->n2p2; break;
    case 6: ptr = m->m1c2; n = m->n1c2; break;
    default: ptr = m->m2c2; n = m->n2c2; break;
  }
  if (!ptr) n = 0;
}

}  // namespace

extern "C" {

void *ref_matcher_create(const int32_t *ip, const double *dp) {
  return new RefMatcher(make_params(ip, dp));
}

void ref_matcher_destroy(void *h) { delete (RefMatcher *)h; }

void ref_matcher_set_intrinsics(void *h, double f, double cu, double cv, double base) {
  ((RefMatcher *)h)->m->setIntrinsics(f, cu, cv, base);
}

void ref_matcher_push(void *h, uint8_t *I1, uint8_t *I2, int32_t w, int32_t hh, int32_t bpl, int32_t replace) {
  uint32_t dims[3] = {(uint32_t)w, (uint32_t)hh, (uint32_t)bpl};
  ((RefMatcher *)h)->m->pushBack(I1, I2, dims, replace != 0);
}

// plain Matcher::matchFeatures (viso/matcher.cpp:183)
void ref_matcher_match(void *h, int32_t method, const double *Tr) {
  Matcher *m = ((RefMatcher *)h)->m;
  if (Tr) {
    Matrix T(4, 4);
    T.eye();
    for (int i = 0; i < 3; i++)
      for (int j = 0; j < 4; j++) T.val[i][j] = Tr[i * 4 + j];
    m->matchFeatures(method, &T);
  } else {
    m->matchFeatures(method);
  }
}

// Same control flow as Matcher::matchFeatures (viso/matcher.cpp:183-241) but
// keeps a copy of the match list after every private stage:
//   stage 0: pass-1 matching()          stage 1: pass-1 removeOutliers()
//   stage 2: pass-2 matching()          stage 3: refinement()
//   stage 4: final removeOutliers()  (== getMatches())
// returns 0 if the sanity checks made matchFeatures return early.
int32_t ref_matcher_match_staged(void *h, int32_t method, const double *Tr) {
  RefMatcher *r = (RefMatcher *)h;
  Matcher *m = r->m;
  for (int s = 0; s < 6; s++) r->stage[s].clear();
  if (method == 0) {
    if (m->m1p2 == 0 || m->n1p2 == 0 || m->m1c2 == 0 || m->n1c2 == 0) return 0;
    if (m->param.multi_stage)
      if (m->m1p1 == 0 || m->n1p1 == 0 || m->m1c1 == 0 || m->n1c1 == 0) return 0;
  } else if (method == 1) {
    if (m->m1c2 == 0 || m->n1c2 == 0 || m->m2c2 == 0 || m->n2c2 == 0) return 0;
    if (m->param.multi_stage)
      if (m->m1c1 == 0 || m->n1c1 == 0 || m->m2c1 == 0 || m->n2c1 == 0) return 0;
  } else {
    if (m->m1p2 == 0 || m->n1p2 == 0 || m->m2p2 == 0 || m->n2p2 == 0 || m->m1c2 == 0 || m->n1c2 == 0 ||
        m->m2c2 == 0 || m->n2c2 == 0)
      return 0;
    if (m->param.multi_stage)
      if (m->m1p1 == 0 || m->n1p1 == 0 || m->m2p1 == 0 || m->n2p1 == 0 || m->m1c1 == 0 || m->n1c1 == 0 ||
          m->m2c1 == 0 || m->n2c1 == 0)
        return 0;
  }
  Matrix T(4, 4);
  T.eye();
  Matrix *Tp = 0;
  if (Tr) {
    for (int i = 0; i < 3; i++)
      for (int j = 0; j < 4; j++) T.val[i][j] = Tr[i * 4 + j];
    Tp = &T;
  }
  m->p_matched_1.clear();
  m->p_matched_2.clear();
  if (m->param.multi_stage) {
    m->matching(m->m1p1, m->m2p1, m->m1c1, m->m2c1, m->n1p1, m->n2p1, m->n1c1, m->n2c1, m->p_matched_1, method,
                false, Tp);
    r->stage[0] = m->p_matched_1;
    m->removeOutliers(m->p_matched_1, method);
    r->stage[1] = m->p_matched_1;
    m->computePriorStatistics(m->p_matched_1, method);
    m->matching(m->m1p2, m->m2p2, m->m1c2, m->m2c2, m->n1p2, m->n2p2, m->n1c2, m->n2c2, m->p_matched_2, method,
                true, Tp);
    r->stage[2] = m->p_matched_2;
    if (m->param.refinement > 0) m->refinement(m->p_matched_2, method);
    r->stage[3] = m->p_matched_2;
    m->removeOutliers(m->p_matched_2, method);
    r->stage[4] = m->p_matched_2;
  } else {
    m->matching(m->m1p2, m->m2p2, m->m1c2, m->m2c2, m->n1p2, m->n2p2, m->n1c2, m->n2c2, m->p_matched_2, method,
                false, Tp);
    r->stage[2] = m->p_matched_2;
    if (m->param.refinement > 0) m->refinement(m->p_matched_2, method);
    r->stage[3] = m->p_matched_2;
    m->removeOutliers(m->p_matched_2, method);
    r->stage[4] = m->p_matched_2;
  }
  return 1;
}

int32_t ref_matcher_stage_size(void *h, int32_t s) { return (int32_t)((RefMatcher *)h)->stage[s].size(); }

void ref_matcher_stage_get(void *h, int32_t s, void *out) {
  RefMatcher *r = (RefMatcher *)h;
  if (!r->stage[s].empty()) memcpy(out, r->stage[s].data(), r->stage[s].size() * sizeof(Matcher::p_match));
}

int32_t ref_matcher_num_ranges(void *h) { return (int32_t)((RefMatcher *)h)->m->ranges.size(); }

void ref_matcher_get_ranges(void *h, float *out) {
  Matcher *m = ((RefMatcher *)h)->m;
  if (!m->ranges.empty()) memcpy(out, m->ranges.data(), m->ranges.size() * sizeof(Matcher::range));
}

int32_t ref_matcher_num_matches(void *h) { return (int32_t)((RefMatcher *)h)->m->p_matched_2.size(); }

void ref_matcher_get_matches(void *h, void *out) {
  std::vector<Matcher::p_match> v = ((RefMatcher *)h)->m->getMatches();
  if (!v.empty()) memcpy(out, v.data(), v.size() * sizeof(Matcher::p_match));
}

void ref_matcher_bucket(void *h, int32_t max_features, float bw, float bh) {
  ((RefMatcher *)h)->m->bucketFeatures(max_features, bw, bh);
}

float ref_matcher_gain(void *h, const int32_t *inl, int32_t n) {
  std::vector<int32_t> v(inl, inl + n);
  return ((RefMatcher *)h)->m->getGain(v);
}

int32_t ref_matcher_num_features(void *h, int32_t which) {
  int32_t *p, n;
  feature_set(((RefMatcher *)h)->m, which, p, n);
  return n;
}

void ref_matcher_get_features(void *h, int32_t which, int32_t *out) {
  int32_t *p, n;
  feature_set(((RefMatcher *)h)->m, which, p, n);
  if (n) memcpy(out, p, (size_t)n * 12 * sizeof(int32_t));
}

// gradient planes of the current/previous left/right image:
// which: 0=1p 1=2p 2=1c 3=2c ; full: 0 = matching resolution, 1 = full resolution
// returns plane size in bytes (bpl*h) or 0 when absent; copies du then dv.
int32_t ref_matcher_get_gradients(void *h, int32_t which, int32_t full, uint8_t *du, uint8_t *dv) {
  Matcher *m = ((RefMatcher *)h)->m;
  uint8_t *pu = 0, *pv = 0;
  const int32_t *dims = (which < 2) ? m->dims_p : m->dims_c;
  if (!full) {
    uint8_t *us[4] = {m->I1p_du, m->I2p_du, m->I1c_du, m->I2c_du};
    uint8_t *vs[4] = {m->I1p_dv, m->I2p_dv, m->I1c_dv, m->I2c_dv};
    pu = us[which];
    pv = vs[which];
  } else {
    uint8_t *us[4] = {m->I1p_du_full, m->I2p_du_full, m->I1c_du_full, m->I2c_du_full};
    uint8_t *vs[4] = {m->I1p_dv_full, m->I2p_dv_full, m->I1c_dv_full, m->I2c_dv_full};
    pu = us[which];
    pv = vs[which];
  }
  if (!pu || !pv) return 0;
  int32_t d[3] = {dims[0], dims[1], dims[2]};
  if (!full && m->param.half_resolution) m->getHalfResolutionDimensions(dims, d);
  int32_t sz = d[2] * d[1];
  memcpy(du, pu, sz);
  memcpy(dv, pv, sz);
  return sz;
}

// ---- free-standing stages -------------------------------------------------

static void *zalloc(size_t n) {
  void *p = 0;
  if (__real_posix_memalign(&p, 16, n + 256)) return 0;
  memset(p, 0, n + 256);
  return p;
}

// filter::sobel5x5 as called from computeFeatures (viso/matcher.cpp:675):
// (I, I_du, I_dv, bpl, h).  Buffers are w*h bytes; w must be a multiple of 16.
void ref_sobel5x5(const uint8_t *in, uint8_t *du, uint8_t *dv, int32_t w, int32_t h) {
  size_t n = (size_t)w * h;
  uint8_t *i = (uint8_t *)zalloc(n), *a = (uint8_t *)zalloc(n), *b = (uint8_t *)zalloc(n);
  memcpy(i, in, n);
  filter::sobel5x5(i, a, b, w, h);
  memcpy(du, a, n);
  memcpy(dv, b, n);
  free(i);
  free(a);
  free(b);
}

void ref_blob5x5(const uint8_t *in, int16_t *out, int32_t w, int32_t h) {
  size_t n = (size_t)w * h;
  uint8_t *i = (uint8_t *)zalloc(n);
  int16_t *o = (int16_t *)zalloc(n * 2);
  memcpy(i, in, n);
  filter::blob5x5(i, o, w, h);
  memcpy(out, o, n * 2);
  free(i);
  free(o);
}

void ref_checkerboard5x5(const uint8_t *in, int16_t *out, int32_t w, int32_t h) {
  size_t n = (size_t)w * h;
  uint8_t *i = (uint8_t *)zalloc(n);
  int16_t *o = (int16_t *)zalloc(n * 2);
  memcpy(i, in, n);
  filter::checkerboard5x5(i, o, w, h);
  memcpy(out, o, n * 2);
  free(i);
  free(o);
}

// Matcher::createHalfResolutionImage (viso/matcher.cpp:636); out is bpl_half*h_half
void ref_half_image(const uint8_t *in, int32_t w, int32_t h, int32_t bpl, uint8_t *out) {
  Matcher::parameters p;
  Matcher m(p);
  int32_t dims[3] = {w, h, bpl}, dh[3];
  m.getHalfResolutionDimensions(dims, dh);
  uint8_t *r = m.createHalfResolutionImage((uint8_t *)in, dims);
  memcpy(out, r, (size_t)dh[2] * dh[1]);
  free(r);
}

// Matcher::nonMaximumSuppression (viso/matcher.cpp:330); out gets {u,v,val,c} per maximum
int32_t ref_nms(const int16_t *f1, const int16_t *f2, int32_t w, int32_t h, int32_t bpl, int32_t n, int32_t tau,
                int32_t *out, int32_t cap) {
  Matcher::parameters p;
  p.nms_tau = tau;
  Matcher m(p);
  int32_t dims[3] = {w, h, bpl};
  std::vector<Matcher::maximum> mx;
  m.nonMaximumSuppression((int16_t *)f1, (int16_t *)f2, dims, mx, n);
  int32_t k = 0;
  for (size_t i = 0; i < mx.size() && k < cap; i++, k++) {
    out[k * 4 + 0] = mx[i].u;
    out[k * 4 + 1] = mx[i].v;
    out[k * 4 + 2] = mx[i].val;
    out[k * 4 + 3] = mx[i].c;
  }
  return (int32_t)mx.size();
}

// triangulate("zQB") exactly as Matcher::removeOutliers calls it
// (viso/matcher.cpp:1214-1256).  returns number of triangles.
int32_t ref_triangulate(const float *pts, int32_t n, int32_t *tris, int32_t cap) {
  struct triangulateio in, out;
  in.numberofpoints = n;
  in.pointlist = (float *)malloc((size_t)n * 2 * sizeof(float));
  memcpy(in.pointlist, pts, (size_t)n * 2 * sizeof(float));
  in.numberofpointattributes = 0;
  in.pointattributelist = NULL;
  in.pointmarkerlist = NULL;
  in.numberofsegments = 0;
  in.numberofholes = 0;
  in.numberofregions = 0;
  in.regionlist = NULL;
  out.pointlist = NULL;
  out.pointattributelist = NULL;
  out.pointmarkerlist = NULL;
  out.trianglelist = NULL;
  out.triangleattributelist = NULL;
  out.neighborlist = NULL;
  out.segmentlist = NULL;
  out.segmentmarkerlist = NULL;
  out.edgelist = NULL;
  out.edgemarkerlist = NULL;
  char sw[] = "zQB";
  triangulate(sw, &in, &out, NULL);
  int32_t nt = out.numberoftriangles;
  for (int32_t i = 0; i < nt && i < cap; i++) {
    tris[i * 3 + 0] = out.trianglelist[i * 3 + 0];
    tris[i * 3 + 1] = out.trianglelist[i * 3 + 1];
    tris[i * 3 + 2] = out.trianglelist[i * 3 + 2];
  }
  free(in.pointlist);
  free(out.pointlist);
  free(out.trianglelist);
  return nt;
}

// Matcher::removeOutliers on an arbitrary match list (viso/matcher.cpp:1207)
int32_t ref_remove_outliers(const int32_t *ip, const double *dp, void *matches, int32_t n, int32_t method) {
  Matcher m(make_params(ip, dp));
  std::vector<Matcher::p_match> v((Matcher::p_match *)matches, (Matcher::p_match *)matches + n);
  m.removeOutliers(v, method);
  if (!v.empty()) memcpy(matches, v.data(), v.size() * sizeof(Matcher::p_match));
  return (int32_t)v.size();
}

// ---- VisualOdometryStereo (viso/viso_stereo.cpp:33) for Tr_delta fixtures ---

void *ref_vo_stereo_create(const int32_t *ip, double f, double cu, double cv, double base, int32_t bucket_max,
                           double bucket_w, double bucket_h) {
  VisualOdometryStereo::parameters p;
  double dp[4] = {f, cu, cv, base};
  Matcher::parameters mp = make_params(ip, dp);
  p.match = mp;
  p.calib.f = f;
  p.calib.cu = cu;
  p.calib.cv = cv;
  p.base = base;
  p.bucket.max_features = bucket_max;
  p.bucket.bucket_width = bucket_w;
  p.bucket.bucket_height = bucket_h;
  return new VisualOdometryStereo(p);
}

void ref_vo_stereo_destroy(void *h) { delete (VisualOdometryStereo *)h; }

// returns: bit0 = process() result, bit1 = Tr_valid *before* the call (i.e.
// whether Tr_delta was handed to matchFeatures).  tr_in gets the Tr_delta that
// was used by matchFeatures, tr_out the one after updateMotion.
int32_t ref_vo_stereo_process(void *h, uint8_t *I1, uint8_t *I2, int32_t w, int32_t hh, int32_t bpl, int32_t replace,
                              double *tr_in, double *tr_out) {
  VisualOdometryStereo *vo = (VisualOdometryStereo *)h;
  uint32_t dims[3] = {(uint32_t)w, (uint32_t)hh, (uint32_t)bpl};
  int32_t valid_before = vo->Tr_valid ? 2 : 0;
  for (int i = 0; i < 4; i++)
    for (int j = 0; j < 4; j++) tr_in[i * 4 + j] = vo->Tr_delta.val[i][j];
  bool ok = vo->process(I1, I2, dims, replace != 0);
  for (int i = 0; i < 4; i++)
    for (int j = 0; j < 4; j++) tr_out[i * 4 + j] = vo->Tr_delta.val[i][j];
  return (ok ? 1 : 0) | valid_before;
}

int32_t ref_vo_num_matches(void *h) { return (int32_t)((VisualOdometryStereo *)h)->matcher->p_matched_2.size(); }

void ref_vo_get_matches(void *h, void *out) {
  std::vector<Matcher::p_match> &v = ((VisualOdometryStereo *)h)->matcher->p_matched_2;
  if (!v.empty()) memcpy(out, v.data(), v.size() * sizeof(Matcher::p_match));
}

// stereo-specific knobs (viso/viso_stereo.h:33-44); call right after create
void ref_vo_stereo_set_ransac(void *h, int32_t iters, double inlier_threshold, int32_t reweighting) {
  VisualOdometryStereo *vo = (VisualOdometryStereo *)h;
  vo->param.ransac_iters = iters;
  vo->param.inlier_threshold = inlier_threshold;
  vo->param.reweighting = reweighting != 0;
}

// VisualOdometry::process(std::vector<p_match>) (viso/viso.h:74-77): egomotion on a given match list
int32_t ref_vo_process_matches(void *h, const void *m, int32_t n, double *tr_out) {
  VisualOdometryStereo *vo = (VisualOdometryStereo *)h;
  std::vector<Matcher::p_match> v((size_t)n);
  if (n) memcpy(v.data(), m, (size_t)n * sizeof(Matcher::p_match));
  bool ok = vo->process(v);
  for (int i = 0; i < 4; i++)
    for (int j = 0; j < 4; j++) tr_out[i * 4 + j] = vo->Tr_delta.val[i][j];
  return ok ? 1 : 0;
}

// the bucketed list estimateMotion saw, and its inliers
int32_t ref_vo_num_bucketed(void *h) { return (int32_t)((VisualOdometryStereo *)h)->p_matched.size(); }
void ref_vo_get_bucketed(void *h, void *out) {
  std::vector<Matcher::p_match> &v = ((VisualOdometryStereo *)h)->p_matched;
  if (!v.empty()) memcpy(out, v.data(), v.size() * sizeof(Matcher::p_match));
}
int32_t ref_vo_num_inliers(void *h) { return (int32_t)((VisualOdometryStereo *)h)->inliers.size(); }
void ref_vo_get_inliers(void *h, int32_t *out) {
  std::vector<int32_t> &v = ((VisualOdometryStereo *)h)->inliers;
  if (!v.empty()) memcpy(out, v.data(), v.size() * sizeof(int32_t));
}

// ---- VisualOdometryMono (viso/viso_mono.cpp) and the Matrix routines it leans on ----------------
// estimateMotion prints timer lines to std::cout (viso/timer.hh); they are swallowed here.
namespace {
struct QuietCout {
  std::streambuf *old;
  QuietCout() : old(std::cout.rdbuf(nullptr)) {}
  ~QuietCout() { std::cout.rdbuf(old); }
};
void copy_T(VisualOdometry *vo, double *out) {
  for (int i = 0; i < 4; i++)
    for (int j = 0; j < 4; j++) out[i * 4 + j] = vo->Tr_delta.val[i][j];
}
}  // namespace

void *ref_vo_mono_create(const int32_t *ip, double f, double cu, double cv, double height, double pitch,
                         int32_t ransac_iters, double inlier_threshold, double motion_threshold, int32_t bucket_max,
                         double bucket_w, double bucket_h) {
  VisualOdometryMono::parameters p;
  double dp[4] = {1, 0, 0, 1};
  p.match = make_params(ip, dp);
  p.calib.f = f;
  p.calib.cu = cu;
  p.calib.cv = cv;
  p.height = height;
  p.pitch = pitch;
  p.ransac_iters = ransac_iters;
  p.inlier_threshold = inlier_threshold;
  p.motion_threshold = motion_threshold;
  p.bucket.max_features = bucket_max;
  p.bucket.bucket_width = bucket_w;
  p.bucket.bucket_height = bucket_h;
  return new VisualOdometryMono(p);
}
void ref_vo_mono_destroy(void *h) { delete (VisualOdometryMono *)h; }

int32_t ref_vo_mono_process(void *h, uint8_t *I, int32_t w, int32_t hh, int32_t bpl, int32_t replace, double *tr_out) {
  VisualOdometryMono *vo = (VisualOdometryMono *)h;
  uint32_t dims[3] = {(uint32_t)w, (uint32_t)hh, (uint32_t)bpl};
  QuietCout q;
  bool ok = vo->process(I, dims, replace != 0);
  copy_T(vo, tr_out);
  return ok ? 1 : 0;
}

int32_t ref_vo_mono_process_matches(void *h, const void *m, int32_t n, double *tr_out) {
  VisualOdometryMono *vo = (VisualOdometryMono *)h;
  std::vector<Matcher::p_match> v((size_t)n);
  if (n) memcpy(v.data(), m, (size_t)n * sizeof(Matcher::p_match));
  QuietCout q;
  bool ok = ((VisualOdometry *)vo)->process(v);
  copy_T(vo, tr_out);
  return ok ? 1 : 0;
}
int32_t ref_vo_mono_num_bucketed(void *h) { return (int32_t)((VisualOdometryMono *)h)->p_matched.size(); }
void ref_vo_mono_get_bucketed(void *h, void *out) {
  std::vector<Matcher::p_match> &v = ((VisualOdometryMono *)h)->p_matched;
  if (!v.empty()) memcpy(out, v.data(), v.size() * sizeof(Matcher::p_match));
}
int32_t ref_vo_mono_num_inliers(void *h) { return (int32_t)((VisualOdometryMono *)h)->inliers.size(); }
void ref_vo_mono_get_inliers(void *h, int32_t *out) {
  std::vector<int32_t> &v = ((VisualOdometryMono *)h)->inliers;
  if (!v.empty()) memcpy(out, v.data(), v.size() * sizeof(int32_t));
}

// Matrix::svd (viso/matrix.cpp:586-850) on a row-major m x n matrix: U m x m, W min(m,n), V n x n
void ref_matrix_svd(const double *A, int32_t m, int32_t n, double *U, double *W, double *V) {
  Matrix a(m, n, A), u, w, v;
  a.svd(u, w, v);
  for (int i = 0; i < m; i++)
    for (int j = 0; j < m; j++) U[i * m + j] = u.val[i][j];
  for (int i = 0; i < std::min(m, n); i++) W[i] = w.val[i][0];
  for (int i = 0; i < n; i++)
    for (int j = 0; j < n; j++) V[i * n + j] = v.val[i][j];
}
// Matrix::det (viso/matrix.cpp:407-422) of a row-major n x n matrix
double ref_matrix_det(const double *A, int32_t n) {
  Matrix a(n, n, A);
  return a.det();
}
// VisualOdometryMono::fundamentalMatrix (viso/viso_mono.cpp:254-284) on (already normalised) matches
void ref_mono_fundamental(void *h, const void *m, int32_t n, const int32_t *active, int32_t na, double *F9) {
  VisualOdometryMono *vo = (VisualOdometryMono *)h;
  std::vector<Matcher::p_match> v((size_t)n);
  if (n) memcpy(v.data(), m, (size_t)n * sizeof(Matcher::p_match));
  std::vector<int32_t> act(active, active + na);
  Matrix F;
  vo->fundamentalMatrix(v, act, F);
  for (int i = 0; i < 3; i++)
    for (int j = 0; j < 3; j++) F9[i * 3 + j] = F.val[i][j];
}

}  // extern "C"
