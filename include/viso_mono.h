/* viso_mono.h -- drop-in replacement for the reference's viso/viso_mono.h (class VisualOdometryMono,
 * viso/viso_mono.h:28-90) AND for its OpenCL subclass VisualOdometryMono_CL (viso/viso_mono_cl.h):
 * same parameters, same process() contract, results equal to the reference's CPU class bit for bit.
 * libvisomatch.so runs the matcher and the inner loops the reference hands to OpenCL -- 8-point fits,
 * Sampson inlier counting, triangulation, ground-plane vote -- as HIP kernels on the MI355X.
 */
#ifndef VISO_MONO_H
#define VISO_MONO_H

#include "viso.h"

class VisualOdometryMono : public VisualOdometry {

public:

  // viso/viso_mono.h:33-46: camera height above ground (m) and pitch (rad, negative = pointing down)
  // fix the scale; RANSAC iterations, Sampson-distance threshold, small-motion rejection
  struct parameters : public VisualOdometry::parameters {
    double  height, pitch;
    int32_t ransac_iters;
    double  inlier_threshold, motion_threshold;
    parameters () : height(1.0), pitch(0.0), ransac_iters(2000), inlier_threshold(0.00001), motion_threshold(100.0) {}
  };

  VisualOdometryMono (parameters param) : vo(0) {          // viso/viso_mono.cpp:27-28
    vsm_vo_mono_params p;
    vsm_vo_mono_default_params(&p);
    copyMatchParameters(param.match,p.match);
    p.bucket_max_features = param.bucket.max_features;
    p.bucket_width = param.bucket.bucket_width;
    p.bucket_height = param.bucket.bucket_height;
    p.f = param.calib.f; p.cu = param.calib.cu; p.cv = param.calib.cv;
    p.height = param.height;
    p.pitch = param.pitch;
    p.ransac_iters = param.ransac_iters;
    p.inlier_threshold = param.inlier_threshold;
    p.motion_threshold = param.motion_threshold;
    vo = vsm_vo_mono_create(&p);
    if (!vo) noDevice();
  }

  ~VisualOdometryMono () { vsm_vo_mono_destroy(vo); }

  // viso/viso_mono.cpp:33-39: dims = {width, height, bytes per line}; false on small motion or failure
  bool process (uint8_t *I,uint32_t* dims,bool replace=false) {
    return vsm_vo_mono_process(vo,I,(int32_t)dims[0],(int32_t)dims[1],(int32_t)dims[2],replace?1:0) != 0;
  }

  using VisualOdometry::process;

  vsm_vo_mono* native () { return vo; }

protected:

  int     hookProcessMatches (const vsm_p_match *m,int32_t n) { return vsm_vo_mono_process_matches(vo,m,n); }
  void    hookMotion (double *t16) { vsm_vo_mono_get_motion(vo,t16); }
  int32_t hookMatches (vsm_p_match *out,int32_t cap) { return out ? vsm_vo_mono_get_matches(vo,out,cap) : vsm_vo_mono_num_matches(vo); }
  int32_t hookInliers (int32_t *out,int32_t cap) { return out ? vsm_vo_mono_get_inliers(vo,out,cap) : vsm_vo_mono_num_inliers(vo); }
  float   hookGain (const int32_t *inliers,int32_t n) { return vsm_vo_mono_gain(vo,inliers,n); }

private:

  vsm_vo_mono *vo;
};

#endif
