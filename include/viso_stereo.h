/* viso_stereo.h -- drop-in replacement for the reference's viso/viso_stereo.h (class
 * VisualOdometryStereo, viso/viso_stereo.h:28-88): same parameters, same process() contract, same
 * results bit for bit (tests/test_gpu_parity.py), executed by libvisomatch.so: feature matching on the
 * MI355X, bucketing and the RANSAC / Gauss-Newton egomotion on the host pool.
 */
#ifndef VISO_STEREO_H
#define VISO_STEREO_H

#include "viso.h"

class VisualOdometryStereo : public VisualOdometry {

public:

  // viso/viso_stereo.h:33-44: baseline in metres, RANSAC iterations, inlier threshold in pixels,
  // down-weighting of matches far from the principal point
  struct parameters : public VisualOdometry::parameters {
    double  base;
    int32_t ransac_iters;
    double  inlier_threshold;
    bool    reweighting;
    parameters () : base(1.0), ransac_iters(200), inlier_threshold(2.0), reweighting(true) {}
  };

  // viso/viso_stereo.cpp:27-29 (+ the base constructor's srand(0), viso/viso.cpp:35)
  VisualOdometryStereo (parameters param) : vo(0) {
    vsm_vo_stereo_params p;
    vsm_vo_stereo_default_params(&p);
    copyMatchParameters(param.match,p.match);
    p.bucket_max_features = param.bucket.max_features;
    p.bucket_width = param.bucket.bucket_width;
    p.bucket_height = param.bucket.bucket_height;
    p.f = param.calib.f; p.cu = param.calib.cu; p.cv = param.calib.cv;
    p.base = param.base;
    p.ransac_iters = param.ransac_iters;
    p.inlier_threshold = param.inlier_threshold;
    p.reweighting = param.reweighting ? 1 : 0;
    vo = vsm_vo_stereo_create(&p);
    if (!vo) noDevice();
  }

  ~VisualOdometryStereo () { vsm_vo_stereo_destroy(vo); }

  // viso/viso_stereo.cpp:33-40: dims = {width, height, bytes per line}; false on failure
  bool process (uint8_t *I1,uint8_t *I2,uint32_t* dims,bool replace=false) {
    return vsm_vo_stereo_process(vo,I1,I2,(int32_t)dims[0],(int32_t)dims[1],(int32_t)dims[2],replace?1:0) != 0;
  }

  using VisualOdometry::process;

  vsm_vo_stereo* native () { return vo; }

protected:

  int     hookProcessMatches (const vsm_p_match *m,int32_t n) { return vsm_vo_stereo_process_matches(vo,m,n); }
  void    hookMotion (double *t16) { vsm_vo_stereo_get_motion(vo,t16); }
  int32_t hookMatches (vsm_p_match *out,int32_t cap) { return out ? vsm_vo_stereo_get_matches(vo,out,cap) : vsm_vo_stereo_num_matches(vo); }
  int32_t hookInliers (int32_t *out,int32_t cap) { return out ? vsm_vo_stereo_get_inliers(vo,out,cap) : vsm_vo_stereo_num_inliers(vo); }
  float   hookGain (const int32_t *inliers,int32_t n) { return vsm_vo_stereo_gain(vo,inliers,n); }

private:

  vsm_vo_stereo *vo;
};

#endif
