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
  VisualOdometryStereo (parameters param) {
    vsm_vo_stereo_params p;
    vsm_vo_stereo_default_params(&p);
    p.match.nms_n = param.match.nms_n;
    p.match.nms_tau = param.match.nms_tau;
    p.match.match_binsize = param.match.match_binsize;
    p.match.match_radius = param.match.match_radius;
    p.match.match_disp_tolerance = param.match.match_disp_tolerance;
    p.match.outlier_disp_tolerance = param.match.outlier_disp_tolerance;
    p.match.outlier_flow_tolerance = param.match.outlier_flow_tolerance;
    p.match.multi_stage = param.match.multi_stage;
    p.match.half_resolution = param.match.half_resolution;
    p.match.refinement = param.match.refinement;
    p.match.f = param.match.f; p.match.cu = param.match.cu; p.match.cv = param.match.cv; p.match.base = param.match.base;
    p.bucket_max_features = param.bucket.max_features;
    p.bucket_width = param.bucket.bucket_width;
    p.bucket_height = param.bucket.bucket_height;
    p.f = param.calib.f; p.cu = param.calib.cu; p.cv = param.calib.cv;
    p.base = param.base;
    p.ransac_iters = param.ransac_iters;
    p.inlier_threshold = param.inlier_threshold;
    p.reweighting = param.reweighting ? 1 : 0;
    vo = vsm_vo_stereo_create(&p);
    if (!vo) {
      std::cerr << "ERROR: visomatch: no usable HIP device (this library has no CPU path)" << std::endl;
      abort();
    }
  }

  ~VisualOdometryStereo () {}

  // viso/viso_stereo.cpp:33-40: dims = {width, height, bytes per line}; false on failure
  bool process (uint8_t *I1,uint8_t *I2,uint32_t* dims,bool replace=false) {
    return vsm_vo_stereo_process(vo,I1,I2,(int32_t)dims[0],(int32_t)dims[1],(int32_t)dims[2],replace?1:0) != 0;
  }

  using VisualOdometry::process;
};

#endif
