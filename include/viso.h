/* viso.h -- drop-in replacement for the reference's viso/viso.h (class VisualOdometry,
 * viso/viso.h:28-131): the public types (calibration, bucketing, parameters), the accessors and
 * process(matches) are kept; the work is done by libvisomatch.so through its C-ABI
 * (include/visomatch.h, vsm_vo_stereo_* / vsm_vo_mono_*), so viso/viso.cpp, viso/viso_stereo.cpp,
 * viso/viso_mono.cpp, viso/matcher.cpp, viso/filter.cpp and viso/triangle.cpp drop out of the build
 * (INTEGRATION.md).  Like the reference header it relies on the project's own "matrix.h" for `Matrix`.
 *
 * The reference's protected estimateMotion / updateMotion plumbing lives inside the library; what
 * a derived class provides here is the handful of forwarding hooks below.
 */
#ifndef VISO_H
#define VISO_H

#include "matrix.h"
#include "matcher.h"

class VisualOdometry {

public:

  // camera calibration (viso/viso.h:33-42): focal length and principal point in pixels
  struct calibration {
    double f, cu, cv;
    calibration () : f(1), cu(0), cv(0) {}
  };

  // bucketing (viso/viso.h:45-54): at most max_features matches per bucket_width x bucket_height cell
  struct bucketing {
    int32_t max_features;
    double  bucket_width, bucket_height;
    bucketing () : max_features(2), bucket_width(50), bucket_height(50) {}
  };

  // viso/viso.h:57-61
  struct parameters {
    Matcher::parameters         match;
    VisualOdometry::bucketing   bucket;
    VisualOdometry::calibration calib;
  };

  virtual ~VisualOdometry () {}

  // egomotion from matches computed elsewhere (viso/viso.h:74-77)
  bool process (std::vector<Matcher::p_match> p_matched_) {
    return hookProcessMatches(p_matched_.empty() ? 0 : reinterpret_cast<vsm_p_match*>(&p_matched_[0]),
                              (int32_t)p_matched_.size()) != 0;
  }

  // Tr_delta: previous -> current camera coordinates, kept from the last success (viso/viso.h:79-87)
  Matrix getMotion () {
    double t[16];
    hookMotion(t);
    return Matrix(4,4,t);
  }

  // matches of the internal matcher after bucketing (viso/viso.h:89)
  std::vector<Matcher::p_match> getMatches () {
    std::vector<Matcher::p_match> out((size_t)hookMatches(0,0));
    if (!out.empty())
      hookMatches(reinterpret_cast<vsm_p_match*>(&out[0]),(int32_t)out.size());
    return out;
  }

  int32_t getNumberOfMatches () { return hookMatches(0,0); }   // viso/viso.h:92
  int32_t getNumberOfInliers () { return hookInliers(0,0); }   // viso/viso.h:95

  std::vector<int32_t> getInlierIndices () {                   // viso/viso.h:98
    std::vector<int32_t> out((size_t)hookInliers(0,0));
    if (!out.empty())
      hookInliers(&out[0],(int32_t)out.size());
    return out;
  }

  float getGain (std::vector<int32_t> inliers_) {              // viso/viso.h:103
    return hookGain(inliers_.empty() ? 0 : &inliers_[0],(int32_t)inliers_.size());
  }

  // "f00 f01 ... f23" like the reference's stream operator (viso/viso.h:106-113)
  friend std::ostream& operator<< (std::ostream &os,VisualOdometry &viso) {
    Matrix p = viso.getMotion();
    for (int32_t i=0; i<3; i++)
      for (int32_t j=0; j<4; j++)
        os << p.val[i][j] << ((i==2 && j==3) ? "" : " ");
    return os;
  }

protected:

  VisualOdometry () {}

  // forwarding hooks (out == 0: return the count only)
  virtual int     hookProcessMatches (const vsm_p_match *m,int32_t n) = 0;
  virtual void    hookMotion (double *t16) = 0;
  virtual int32_t hookMatches (vsm_p_match *out,int32_t cap) = 0;
  virtual int32_t hookInliers (int32_t *out,int32_t cap) = 0;
  virtual float   hookGain (const int32_t *inliers,int32_t n) = 0;

  static void copyMatchParameters (const Matcher::parameters &in,vsm_params &p) {
    p.nms_n = in.nms_n;
    p.nms_tau = in.nms_tau;
    p.match_binsize = in.match_binsize;
    p.match_radius = in.match_radius;
    p.match_disp_tolerance = in.match_disp_tolerance;
    p.outlier_disp_tolerance = in.outlier_disp_tolerance;
    p.outlier_flow_tolerance = in.outlier_flow_tolerance;
    p.multi_stage = in.multi_stage;
    p.half_resolution = in.half_resolution;
    p.refinement = in.refinement;
    p.f = in.f; p.cu = in.cu; p.cv = in.cv; p.base = in.base;
  }

  static void noDevice () {
    std::cerr << "ERROR: visomatch: no usable HIP device (this library has no CPU path)" << std::endl;
    abort();
  }

private:

  VisualOdometry (const VisualOdometry&);
  VisualOdometry& operator= (const VisualOdometry&);
};

#endif
