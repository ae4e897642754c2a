// Host-side stages of matchFeatures that stay on the CPU (SURVEY.md section 8a rows O1, M4, B1 and
// the least-squares tail of refinement==2): exact Delaunay support test, prior statistics,
// bucketing, gain.  Product code -- independent of oracle/.
#pragma once

#include <stdint.h>

#include <vector>

#include "visomatch.h"

// Exact replica of Triangle 1.6's divide-and-conquer Delaunay ("zQB", as called from
// Matcher::removeOutliers, viso/matcher.cpp:1255-1256) for integer-valued points, with a
// reusable workspace (no allocation in steady state) and exact int64 predicates.
class ExactDelaunay {
 public:
  // points are (x[i], y[i]); after run(), triangles() lists vertex triples by input index
  void run(const int32_t *x, const int32_t *y, int32_t n);
  int32_t num_triangles() const { return ntri_out_; }
  const int32_t *triangles() const { return tri_out_.data(); }

 private:
  struct OTri {
    int32_t t, o;
  };
  const int32_t *x_ = nullptr, *y_ = nullptr;
  std::vector<int32_t> nb_, vx_, order_, tri_out_;
  int32_t ntri_ = 0, ntri_out_ = 0;
  uint64_t seed_ = 1;

  uint32_t rnd(uint32_t choices);
  void partition(int32_t *a, int32_t n, int axis, int32_t &left, int32_t &right);
  void sort2(int32_t *a, int axis);
  void vertex_sort(int32_t *a, int32_t n);
  void vertex_median(int32_t *a, int32_t n, int32_t median, int axis);
  void alternate_axes(int32_t *a, int32_t n, int axis);
  void recurse(int32_t *a, int32_t n, int axis, OTri &farleft, OTri &farright);
  void merge_hulls(OTri &farleft, OTri &innerleft, OTri &innerright, OTri &farright, int axis);

  OTri make();
  inline OTri sym(OTri a) const {
    int32_t e = nb_[a.t * 3 + a.o];
    return OTri{e >> 2, e & 3};
  }
  static inline OTri lnext(OTri a) { return OTri{a.t, a.o == 2 ? 0 : a.o + 1}; }
  static inline OTri lprev(OTri a) { return OTri{a.t, a.o == 0 ? 2 : a.o - 1}; }
  inline int32_t org(OTri a) const { return vx_[a.t * 3 + (a.o == 2 ? 0 : a.o + 1)]; }
  inline int32_t dest(OTri a) const { return vx_[a.t * 3 + (a.o == 0 ? 2 : a.o - 1)]; }
  inline int32_t apex(OTri a) const { return vx_[a.t * 3 + a.o]; }
  inline void set_org(OTri a, int32_t v) { vx_[a.t * 3 + (a.o == 2 ? 0 : a.o + 1)] = v; }
  inline void set_dest(OTri a, int32_t v) { vx_[a.t * 3 + (a.o == 0 ? 2 : a.o - 1)] = v; }
  inline void set_apex(OTri a, int32_t v) { vx_[a.t * 3 + a.o] = v; }
  inline void bond(OTri a, OTri b) {
    nb_[a.t * 3 + a.o] = b.t * 4 + b.o;
    nb_[b.t * 3 + b.o] = a.t * 4 + a.o;
  }
  inline int64_t ccw(int32_t a, int32_t b, int32_t c) const {
    return (int64_t)(x_[a] - x_[c]) * (y_[b] - y_[c]) - (int64_t)(y_[a] - y_[c]) * (x_[b] - x_[c]);
  }
  inline int64_t incircle(int32_t a, int32_t b, int32_t c, int32_t d) const {
    int64_t adx = x_[a] - x_[d], ady = y_[a] - y_[d], bdx = x_[b] - x_[d], bdy = y_[b] - y_[d];
    int64_t cdx = x_[c] - x_[d], cdy = y_[c] - y_[d];
    return (adx * adx + ady * ady) * (bdx * cdy - cdx * bdy) + (bdx * bdx + bdy * bdy) * (cdx * ady - adx * cdy) +
           (cdx * cdx + cdy * cdy) * (adx * bdy - bdx * ady);
  }
};

struct VsmHostWork {
  ExactDelaunay del;
  std::vector<int32_t> x, y, support;
};

// Matcher::removeOutliers, viso/matcher.cpp:1207-1377 (in place; order preserved)
void vsm_host_remove_outliers(VsmHostWork &w, const vsm_params &p, std::vector<vsm_p_match> &m, int method);

// Matcher::computePriorStatistics, viso/matcher.cpp:734-868 -> ranges[bin][16]
void vsm_host_prior_statistics(const vsm_params &p, const int32_t *dims_c, const std::vector<vsm_p_match> &m,
                               int method, std::vector<float> &ranges);

// least-squares tail of Matcher::parabolicFitting, viso/matcher.cpp:1425-1453.
// c9 = 3x3 costs around the 7x7 minimum at (du,dv); returns false if the match must be dropped.
bool vsm_host_parabolic_update(const int32_t *c9, int du, int dv, float &u2, float &v2);

// Matcher::bucketFeatures, viso/matcher.cpp:243-284
void vsm_host_bucket(std::vector<vsm_p_match> &m, int max_features, float bucket_width, float bucket_height);

// Matcher::getGain, viso/matcher.cpp:286-324
float vsm_host_gain(const uint8_t *I1p, const uint8_t *I1c, const int32_t *dims_p, const int32_t *dims_c,
                    const std::vector<vsm_p_match> &m, const int32_t *inliers, int32_t n);
