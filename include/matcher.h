/* matcher.h -- drop-in replacement for the reference's viso/matcher.h.
 *
 * Same class, same nested types, same public member functions and the same observable
 * behaviour as `class Matcher` of libviso2 as vendored in dphoyes/OpenCL-Structure-from-Motion
 * (viso/matcher.h:37-136, viso/matcher.cpp), but every call forwards to the MI355X-native
 * library libvisomatch.so through its C-ABI (include/visomatch.h).  Callers --
 * VisualOdometry{,Stereo,Mono} (viso/viso.cpp:32,39, viso/viso_stereo.cpp:27,34-38,
 * viso/viso_mono.cpp:34-37) and the MATLAB wrappers (matlab/matcherMex.cpp) -- compile against
 * this header unchanged; drop viso/matcher.cpp, viso/filter.cpp and viso/triangle.cpp from the
 * build and link -lvisomatch instead.
 *
 * Like the reference header this one includes the project's own "matrix.h" (callers expect
 * `Matrix` to be visible and pass `Matrix *Tr_delta`).
 */
#ifndef __MATCHER_H__
#define __MATCHER_H__

#include <stdint.h>
#include <stdlib.h>
#include <iostream>
#include <vector>
// headers the reference's matcher.h pulled in and its callers rely on transitively
#include <stdio.h>
#include <string.h>
#include <math.h>
#include <algorithm>

#include "matrix.h"
#include "visomatch.h"

class Matcher {

public:

  // Tunables; names, order and defaults are the public contract of viso/matcher.h:42-69.
  //   nms_n, nms_tau ........ NMS cell radius and response threshold
  //   match_binsize ......... side of the square search bins (speed only)
  //   match_radius .......... +-search window in pixels; match_disp_tolerance: +-rows for stereo
  //   outlier_*_tolerance ... Delaunay-support test thresholds
  //   multi_stage ........... sparse pass first, its statistics narrow the dense pass
  //   half_resolution ....... detect/match at half size, refine at full size
  //   refinement ............ 0 off, 1 integer relocation, 2 parabolic sub-pixel fit
  //   f,cu,cv,base .......... calibration, only used to predict matches from Tr_delta
  struct parameters {
    int32_t nms_n, nms_tau;
    int32_t match_binsize, match_radius, match_disp_tolerance;
    int32_t outlier_disp_tolerance, outlier_flow_tolerance;
    int32_t multi_stage, half_resolution, refinement;
    double  f,cu,cv,base;
    parameters ()
      : nms_n(3), nms_tau(50), match_binsize(50), match_radius(200), match_disp_tolerance(2),
        outlier_disp_tolerance(5), outlier_flow_tolerance(5), multi_stage(1), half_resolution(1),
        refinement(1), f(1), cu(0), cv(0), base(1) {}
  };

  // constructor (viso/matcher.cpp:33-61).  Binds the current HIP device.
  Matcher(parameters param) : handle(0) {
    vsm_params p;
    p.nms_n = param.nms_n;
    p.nms_tau = param.nms_tau;
    p.match_binsize = param.match_binsize;
    p.match_radius = param.match_radius;
    p.match_disp_tolerance = param.match_disp_tolerance;
    p.outlier_disp_tolerance = param.outlier_disp_tolerance;
    p.outlier_flow_tolerance = param.outlier_flow_tolerance;
    p.multi_stage = param.multi_stage;
    p.half_resolution = param.half_resolution;
    p.refinement = param.refinement;
    p.f = param.f; p.cu = param.cu; p.cv = param.cv; p.base = param.base;
    handle = vsm_create(&p);
    if (!handle) {
      std::cerr << "ERROR: visomatch: no usable HIP device (this matcher has no CPU path)" << std::endl;
      abort();
    }
  }

  // deconstructor (viso/matcher.cpp:64-93)
  ~Matcher() { vsm_destroy(handle); }

  // intrinsics (viso/matcher.h:78-83)
  void setIntrinsics(double f,double cu,double cv,double base) { vsm_set_intrinsics(handle,f,cu,cv,base); }

  // One match (viso/matcher.h:86-100): pixel position and feature index in the previous /
  // current x left(1) / right(2) image; slots a method does not fill hold -1.  Same 48-byte
  // layout as vsm_p_match, so lists cross the C boundary without conversion.
  struct p_match {
    float u1p,v1p; int32_t i1p;
    float u2p,v2p; int32_t i2p;
    float u1c,v1c; int32_t i1c;
    float u2c,v2c; int32_t i2c;
    p_match(){}
    p_match(float u1p_,float v1p_,int32_t i1p_,float u2p_,float v2p_,int32_t i2p_,
            float u1c_,float v1c_,int32_t i1c_,float u2c_,float v2c_,int32_t i2c_)
      : u1p(u1p_),v1p(v1p_),i1p(i1p_),u2p(u2p_),v2p(v2p_),i2p(i2p_),
        u1c(u1c_),v1c(v1c_),i1c(i1c_),u2c(u2c_),v2c(v2c_),i2c(i2c_) {}
  };

  // pushBack (viso/matcher.cpp:95-181): dims = {width, height, bytes per line}.
  // Bad dims print "ERROR: Image dimension mismatch!" and leave the state untouched.
  void pushBack (uint8_t *I1,uint8_t* I2,uint32_t* dims,const bool replace) {
    vsm_push_back(handle,I1,I2,(int32_t)dims[0],(int32_t)dims[1],(int32_t)dims[2],replace?1:0);
  }
  void pushBack (uint8_t *I1,uint32_t* dims,const bool replace) { pushBack(I1,0,dims,replace); }

  // the MATLAB wrappers of the reference still pass int32_t dims (matlab/matcherMex.cpp:105-111)
  void pushBack (uint8_t *I1,uint8_t* I2,int32_t* dims,const bool replace) {
    vsm_push_back(handle,I1,I2,dims[0],dims[1],dims[2],replace?1:0);
  }
  void pushBack (uint8_t *I1,int32_t* dims,const bool replace) { pushBack(I1,(uint8_t*)0,dims,replace); }

  // matchFeatures (viso/matcher.cpp:183-241): method 0 = flow, 1 = stereo, 2 = quad matching.
  // With missing ring-buffer entries it returns silently and keeps the previous matches.
  void matchFeatures(int32_t method, Matrix *Tr_delta = 0) {
    if (Tr_delta) {
      double t[12];
      for (int32_t i=0; i<3; i++)
        for (int32_t j=0; j<4; j++)
          t[i*4+j] = Tr_delta->val[i][j];   // what viso/matcher.cpp:989-1002 reads
      vsm_match(handle,method,t);
    } else {
      vsm_match(handle,method,0);
    }
  }

  // bucketFeatures (viso/matcher.cpp:243-284)
  void bucketFeatures(int32_t max_features,float bucket_width,float bucket_height) {
    vsm_bucket(handle,max_features,bucket_width,bucket_height);
  }

  // getMatches (viso/matcher.h:131)
  std::vector<Matcher::p_match> getMatches() {
    static_assert(sizeof(p_match)==sizeof(vsm_p_match),"p_match layout");
    std::vector<Matcher::p_match> out((size_t)vsm_num_matches(handle));
    if (!out.empty())
      vsm_get_matches(handle,reinterpret_cast<vsm_p_match*>(&out[0]),(int32_t)out.size());
    return out;
  }

  // getGain (viso/matcher.cpp:286-324)
  float getGain (std::vector<int32_t> inliers) {
    return vsm_gain(handle,inliers.empty() ? 0 : &inliers[0],(int32_t)inliers.size());
  }

  // access to the C handle (stage-level views, profiling)
  vsm_handle* native() { return handle; }

private:

  Matcher(const Matcher&);             // the reference's copy would double-free; forbid it
  Matcher& operator=(const Matcher&);

  vsm_handle *handle;
};

#endif
