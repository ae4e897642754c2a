/* TEST INFRASTRUCTURE ONLY -- CPU restatement ("oracle") of libviso2's matcher
 * hot path as vendored in dphoyes/OpenCL-Structure-from-Motion (viso/matcher.cpp,
 * viso/filter.cpp, the Delaunay part of viso/triangle.cpp).
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may use
 * this library; the product (libvisomatch.so) never links or calls it.
 *
 * Parity status: PINNED -- every stage is checked against the real reference
 * compiled in place (oracle/_ref, see oracle/Makefile) by tests/test_oracle_vs_ref.py
 * and against the committed golden vectors in tests/golden/ generated from it.
 */
#ifndef VISO_ORACLE_H
#define VISO_ORACLE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* Matcher::parameters, viso/matcher.h:42-69 */
typedef struct {
  int32_t nms_n, nms_tau, match_binsize, match_radius, match_disp_tolerance;
  int32_t outlier_disp_tolerance, outlier_flow_tolerance, multi_stage, half_resolution, refinement;
  double f, cu, cv, base;
} vo_params;

/* Matcher::p_match, viso/matcher.h:86-100 (48 bytes) */
typedef struct {
  float u1p, v1p;
  int32_t i1p;
  float u2p, v2p;
  int32_t i2p;
  float u1c, v1c;
  int32_t i1c;
  float u2c, v2c;
  int32_t i2c;
} vo_match;

/* Matcher::range, viso/matcher.h:152-157 (64 bytes) */
typedef struct {
  float u_min[4], u_max[4], v_min[4], v_max[4];
} vo_range;

typedef struct vo_matcher vo_matcher;

void vo_default_params(vo_params *p);

/* ---- free-standing stages (all images are bpl x h, bpl % 16 == 0) ---- */
int32_t vo_bpl16(int32_t w); /* w + 15 - (w-1)%16, viso/matcher.cpp:160 */
void vo_half_image(const uint8_t *in, int32_t w, int32_t h, int32_t bpl, uint8_t *out);
void vo_sobel5x5(const uint8_t *in, uint8_t *du, uint8_t *dv, int32_t bpl, int32_t h);
void vo_blob5x5(const uint8_t *in, int16_t *out, int32_t bpl, int32_t h);
void vo_checkerboard5x5(const uint8_t *in, int16_t *out, int32_t bpl, int32_t h);
int32_t vo_nms(const int16_t *f1, const int16_t *f2, int32_t w, int32_t h, int32_t bpl, int32_t n, int32_t tau,
               int32_t *out4, int32_t cap);
void vo_descriptor(const uint8_t *du, const uint8_t *dv, int32_t bpl, int32_t u, int32_t v, uint8_t *desc32);
/* Delaunay triangulation identical to Triangle 1.6 "zQB" on integer-valued points */
int32_t vo_delaunay(const float *pts_xy, int32_t n, int32_t *tris, int32_t cap);
int32_t vo_remove_outliers(const vo_params *p, vo_match *m, int32_t n, int32_t method);

/* ---- the matcher object (mirrors class Matcher, viso/matcher.h:37-136) ---- */
vo_matcher *vo_create(const vo_params *p);
void vo_destroy(vo_matcher *m);
void vo_set_intrinsics(vo_matcher *m, double f, double cu, double cv, double base);
/* returns 0 on success, -1 for the reference's "Image dimension mismatch" */
int32_t vo_push_back(vo_matcher *m, const uint8_t *I1, const uint8_t *I2, int32_t w, int32_t h, int32_t bpl,
                     int32_t replace);
/* returns 1 if matching ran, 0 if the reference's sanity checks return early, -2 unsupported */
int32_t vo_match_features(vo_matcher *m, int32_t method, const double *Tr_delta_rowmajor_3x4_or_4x4);
int32_t vo_num_matches(const vo_matcher *m);
void vo_get_matches(const vo_matcher *m, vo_match *out);
void vo_bucket_features(vo_matcher *m, int32_t max_features, float bucket_width, float bucket_height);
float vo_get_gain(const vo_matcher *m, const int32_t *inliers, int32_t n);

/* stage captures of the last vo_match_features():
 * 0 pass-1 matching, 1 pass-1 outlier removal, 2 pass-2 matching, 3 refinement, 4 final */
int32_t vo_stage_size(const vo_matcher *m, int32_t stage);
void vo_stage_get(const vo_matcher *m, int32_t stage, vo_match *out);
int32_t vo_num_ranges(const vo_matcher *m);
void vo_get_ranges(const vo_matcher *m, vo_range *out);
/* which: 0=1p1 1=2p1 2=1c1 3=2c1 4=1p2 5=2p2 6=1c2 7=2c2 ; records are int32[12] */
int32_t vo_num_features(const vo_matcher *m, int32_t which);
void vo_get_features(const vo_matcher *m, int32_t which, int32_t *out);
/* which: 0=1p 1=2p 2=1c 3=2c; returns plane bytes or 0 */
int32_t vo_get_gradients(const vo_matcher *m, int32_t which, int32_t full, uint8_t *du, uint8_t *dv);

/* work counters of the last vo_match_features() (for the bench's algorithmic-bytes model,
 * SURVEY.md section 8d): {findMatch calls Q, candidates visited C, SADs S, matches refined M, matches out,
 * and the pass-1 share Q1, C1, S1} */
void vo_get_counters(const vo_matcher *m, int64_t *out8);

/* ---- stereo egomotion, the caller downstream of the matcher (viso_ego_oracle.c) ---- */
/* VisualOdometry::calibration + VisualOdometryStereo::parameters, viso/viso.h:33-42, viso/viso_stereo.h:33-44 */
typedef struct {
  double f, cu, cv, base;
  int32_t ransac_iters;
  double inlier_threshold;
  int32_t reweighting;
} vo_ego_params;
void vo_ego_default_params(vo_ego_params *e);
/* the process-wide sampler of VisualOdometry::getRandomSample (viso/viso.cpp:93), seed 71 */
void vo_ego_sampler_seed(uint32_t s);
uint32_t vo_ego_sampler_state(void);
/* VisualOdometryStereo::estimateMotion; 1 ok (tr6 filled), 0 failed, -1 fewer than 6 matches */
int32_t vo_estimate_motion_stereo(const vo_match *m, int32_t n, const vo_ego_params *ep, double *tr6, int32_t *inliers,
                                  int32_t *n_inliers);
void vo_tr_vector_to_matrix(const double *tr6, double *T16);

typedef struct vo_stereo vo_stereo; /* VisualOdometryStereo, viso/viso_stereo.h:28-88 */
vo_stereo *vo_stereo_create(const vo_params *mp, int32_t bucket_max, double bucket_w, double bucket_h,
                            const vo_ego_params *ep);
void vo_stereo_destroy(vo_stereo *v);
int32_t vo_stereo_process(vo_stereo *v, const uint8_t *I1, const uint8_t *I2, int32_t w, int32_t h, int32_t bpl,
                          int32_t replace);
int32_t vo_stereo_process_matches(vo_stereo *v, const vo_match *m, int32_t n);
void vo_stereo_get_motion(const vo_stereo *v, double *T16);
int32_t vo_stereo_tr_valid(const vo_stereo *v);
int32_t vo_stereo_num_matches(const vo_stereo *v);
void vo_stereo_get_matches(const vo_stereo *v, vo_match *out);
int32_t vo_stereo_num_inliers(const vo_stereo *v);
void vo_stereo_get_inliers(const vo_stereo *v, int32_t *out);
vo_matcher *vo_stereo_matcher(vo_stereo *v);

/* ---- monocular egomotion (viso_mono_oracle.c), the path the reference offloads to OpenCL ---- */
/* VisualOdometryMono::parameters + calibration, viso/viso_mono.h:33-46, viso/viso.h:33-42 */
typedef struct {
  double f, cu, cv;
  double height, pitch;
  int32_t ransac_iters;
  double inlier_threshold, motion_threshold;
} vo_mono_params;
void vo_mono_default_params(vo_mono_params *p);
/* Matrix::svd / Matrix::det of the reference (viso/matrix.cpp:586-850, :407-422), row-major */
void vo_matrix_svd(const double *A, int32_t m, int32_t n, double *U, double *W, double *V);
double vo_matrix_det(const double *A, int32_t n);
/* VisualOdometryMono::fundamentalMatrix on (normalised) matches, viso/viso_mono.cpp:264-294 */
void vo_mono_fundamental(const vo_match *m, const int32_t *active, int32_t na, double *F9);
/* VisualOdometryMono::estimateMotion; 1 ok (tr6 filled) / 0; *n_inliers = -1 when the reference's
 * inlier list would be left untouched (early exits before the RANSAC) */
int32_t vo_estimate_motion_mono(const vo_match *m, int32_t n, const vo_mono_params *p, double *tr6, int32_t *inliers,
                                int32_t *n_inliers);
typedef struct vo_mono vo_mono; /* VisualOdometryMono, viso/viso_mono.h:28-90 */
vo_mono *vo_mono_create(const vo_params *mp, int32_t bucket_max, double bucket_w, double bucket_h,
                        const vo_mono_params *ep);
void vo_mono_destroy(vo_mono *v);
int32_t vo_mono_process(vo_mono *v, const uint8_t *I, int32_t w, int32_t h, int32_t bpl, int32_t replace);
int32_t vo_mono_process_matches(vo_mono *v, const vo_match *m, int32_t n);
void vo_mono_get_motion(const vo_mono *v, double *T16);
int32_t vo_mono_num_matches(const vo_mono *v);
void vo_mono_get_matches(const vo_mono *v, vo_match *out);
int32_t vo_mono_num_inliers(const vo_mono *v);
void vo_mono_get_inliers(const vo_mono *v, int32_t *out);

#ifdef __cplusplus
}
#endif
#endif
