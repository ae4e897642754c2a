/* viso.h -- drop-in replacement for the reference's viso/viso.h (class VisualOdometry,
 * viso/viso.h:28-131) for callers that only use the stereo pipeline: the public types
 * (calibration, bucketing, parameters), the accessors and process(matches) are kept; the work is
 * done by libvisomatch.so through its C-ABI (include/visomatch.h, vsm_vo_stereo_*), so
 * viso/viso.cpp, viso/viso_stereo.cpp, viso/matcher.cpp, viso/filter.cpp and viso/triangle.cpp drop
 * out of the build.  Like the reference header it relies on the project's own
 * "matrix.h" for `Matrix`.
 *
 * The base class here is concrete and owns the stereo handle: the reference's pure-virtual
 * estimateMotion / updateMotion plumbing lives inside the library.
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

  ~VisualOdometry () { if (vo) vsm_vo_stereo_destroy(vo); }

  // egomotion from matches computed elsewhere (viso/viso.h:74-77)
  bool process (std::vector<Matcher::p_match> p_matched_) {
    return vsm_vo_stereo_process_matches(vo, p_matched_.empty() ? 0 : reinterpret_cast<vsm_p_match*>(&p_matched_[0]),
                                         (int32_t)p_matched_.size()) != 0;
  }

  // Tr_delta: previous -> current camera coordinates, kept from the last success (viso/viso.h:79-87)
  Matrix getMotion () {
    double t[16];
    vsm_vo_stereo_get_motion(vo,t);
    return Matrix(4,4,t);
  }

  // matches of the internal matcher after bucketing (viso/viso.h:89)
  std::vector<Matcher::p_match> getMatches () {
    std::vector<Matcher::p_match> out((size_t)vsm_vo_stereo_num_matches(vo));
    if (!out.empty())
      vsm_vo_stereo_get_matches(vo,reinterpret_cast<vsm_p_match*>(&out[0]),(int32_t)out.size());
    return out;
  }

  int32_t getNumberOfMatches () { return vsm_vo_stereo_num_matches(vo); }   // viso/viso.h:92
  int32_t getNumberOfInliers () { return vsm_vo_stereo_num_inliers(vo); }   // viso/viso.h:95

  std::vector<int32_t> getInlierIndices () {                                // viso/viso.h:98
    std::vector<int32_t> out((size_t)vsm_vo_stereo_num_inliers(vo));
    if (!out.empty())
      vsm_vo_stereo_get_inliers(vo,&out[0],(int32_t)out.size());
    return out;
  }

  float getGain (std::vector<int32_t> inliers_) {                           // viso/viso.h:103
    return vsm_vo_stereo_gain(vo,inliers_.empty() ? 0 : &inliers_[0],(int32_t)inliers_.size());
  }

  // "f00 f01 ... f23" like the reference's stream operator (viso/viso.h:106-113)
  friend std::ostream& operator<< (std::ostream &os,VisualOdometry &viso) {
    Matrix p = viso.getMotion();
    for (int32_t i=0; i<3; i++)
      for (int32_t j=0; j<4; j++)
        os << p.val[i][j] << ((i==2 && j==3) ? "" : " ");
    return os;
  }

  vsm_vo_stereo* native () { return vo; }

protected:

  VisualOdometry () : vo(0) {}
  vsm_vo_stereo *vo;

private:

  VisualOdometry (const VisualOdometry&);
  VisualOdometry& operator= (const VisualOdometry&);
};

#endif
