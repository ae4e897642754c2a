/* TEST INFRASTRUCTURE ONLY -- CPU restatement ("oracle") of libviso2's stereo egomotion step, the
 * caller on the far side of the matcher hot path:
 *
 *   VisualOdometryStereo::process        viso/viso_stereo.cpp:33-40
 *   VisualOdometry::updateMotion         viso/viso.cpp:42-58
 *   VisualOdometryStereo::estimateMotion viso/viso_stereo.cpp:42-146   (RANSAC + Gauss-Newton)
 *   ... getInlier / updateParameters / computeObservations / computeResidualsAndJacobian
 *                                        viso/viso_stereo.cpp:148-315
 *   VisualOdometry::getRandomSample      viso/viso.cpp:91-108  (std::default_random_engine(71) +
 *                                        std::uniform_int_distribution of libstdc++ 11)
 *   Matrix::solve                        viso/matrix.cpp:424-513 (Gauss-Jordan, full pivoting)
 *
 * Everything is double arithmetic in the reference's evaluation order (compile with
 * -ffp-contract=off); sin/cos come from the host libm exactly like the reference build.
 *
 * Parity status: PINNED by tests/test_oracle_vs_ref.py (against oracle/_ref in a fresh process,
 * because the reference's sampler is a function-local static) and by the Tr_delta trail of
 * tests/golden/cfg2_seq200_tr.npz.
 */
#include "viso_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

/* ------------------------------------------------------------------------------------------ */
/* the sampler: std::minstd_rand0 (a=16807, m=2^31-1) seeded with 71, process-wide like the    */
/* function-local static in viso/viso.cpp:93                                                   */
/* ------------------------------------------------------------------------------------------ */
static uint32_t g_lcg = 71;

void vo_ego_sampler_seed(uint32_t s) {
  uint32_t r = s % 2147483647u; /* linear_congruential_engine::seed with c == 0 */
  g_lcg = r ? r : 1u;
}
uint32_t vo_ego_sampler_state(void) { return g_lcg; }

static uint64_t lcg_step(void) {
  g_lcg = (uint32_t)(((uint64_t)g_lcg * 16807u) % 2147483647u);
  return g_lcg;
}

/* uniform_int_distribution<unsigned>(lo,hi)(engine), libstdc++ bits/uniform_int_dist.h,
 * the "downscaling, two divisions" branch (the engine's range 2^31-3 is not 2^32-1/2^64-1) */
static uint32_t draw_between(uint32_t lo, uint32_t hi) {
  const uint64_t eng_range = 2147483646ull - 1ull;
  const uint64_t span = (uint64_t)hi - (uint64_t)lo;
  uint64_t r;
  if (eng_range > span) {
    const uint64_t buckets = span + 1;
    const uint64_t scale = eng_range / buckets;
    const uint64_t limit = buckets * scale;
    do r = lcg_step() - 1ull;
    while (r >= limit);
    r /= scale;
  } else {
    r = lcg_step() - 1ull; /* span == eng_range; larger spans cannot occur for match counts */
  }
  return (uint32_t)(r + lo);
}

uint32_t vo_ego_draw_between(uint32_t lo, uint32_t hi) { return draw_between(lo, hi); } /* for viso_mono_oracle.c */

/* partial Fisher-Yates over 0..n-1, first `k` entries (viso/viso.cpp:96-105) */
static void sample_k(int32_t *scratch, int32_t n, int32_t k, int32_t *out) {
  for (int32_t i = 0; i < n; i++) scratch[i] = i;
  for (int32_t i = 0; i < k; i++) {
    uint32_t j = draw_between((uint32_t)i, (uint32_t)(n - 1));
    int32_t t = scratch[i];
    scratch[i] = scratch[j];
    scratch[j] = t;
  }
  for (int32_t i = 0; i < k; i++) out[i] = scratch[i];
}

/* ------------------------------------------------------------------------------------------ */
/* Matrix::solve for a 6x6 system with one right-hand side (viso/matrix.cpp:424-513, eps 1e-20) */
/* ------------------------------------------------------------------------------------------ */
static int gj6(double a[6][6], double b[6]) {
  int32_t used[6] = {0, 0, 0, 0, 0, 0};
  for (int32_t step = 0; step < 6; step++) {
    double best = 0.0;
    int32_t pr = 0, pc = 0;
    for (int32_t r = 0; r < 6; r++) {
      if (used[r] == 1) continue;
      for (int32_t c = 0; c < 6; c++)
        if (used[c] == 0 && fabs(a[r][c]) >= best) {
          best = fabs(a[r][c]);
          pr = r;
          pc = c;
        }
    }
    used[pc]++;
    if (pr != pc) {
      for (int32_t c = 0; c < 6; c++) {
        double t = a[pr][c];
        a[pr][c] = a[pc][c];
        a[pc][c] = t;
      }
      double t = b[pr];
      b[pr] = b[pc];
      b[pc] = t;
    }
    if (fabs(a[pc][pc]) < 1e-20) return 0;
    double inv = 1.0 / a[pc][pc];
    a[pc][pc] = 1.0;
    for (int32_t c = 0; c < 6; c++) a[pc][c] *= inv;
    b[pc] *= inv;
    for (int32_t r = 0; r < 6; r++) {
      if (r == pc) continue;
      double f = a[r][pc];
      a[r][pc] = 0.0;
      for (int32_t c = 0; c < 6; c++) a[r][c] -= a[pc][c] * f;
      b[r] -= b[pc] * f;
    }
  }
  return 1; /* the column unscrambling of the reference only touches the inverse, not b */
}

/* ------------------------------------------------------------------------------------------ */
/* projection model                                                                           */
/* ------------------------------------------------------------------------------------------ */
typedef struct {
  double R[3][3];     /* rotation                       viso_stereo.cpp:230-232 */
  double dR[3][3][3]; /* d R / d rx, ry, rz             viso_stereo.cpp:233-241 */
  double t[3];
} pose6;

static void pose_from_vector(const double tr[6], pose6 *p) {
  double sx = sin(tr[0]), cx = cos(tr[0]);
  double sy = sin(tr[1]), cy = cos(tr[1]);
  double sz = sin(tr[2]), cz = cos(tr[2]);
  p->R[0][0] = +cy * cz;                p->R[0][1] = -cy * sz;                p->R[0][2] = +sy;
  p->R[1][0] = +sx * sy * cz + cx * sz; p->R[1][1] = -sx * sy * sz + cx * cz; p->R[1][2] = -sx * cy;
  p->R[2][0] = -cx * sy * cz + sx * sz; p->R[2][1] = +cx * sy * sz + sx * cz; p->R[2][2] = +cx * cy;
  /* d/drx (first row is identically zero) */
  p->dR[0][0][0] = 0;                       p->dR[0][0][1] = 0;                       p->dR[0][0][2] = 0;
  p->dR[0][1][0] = +cx * sy * cz - sx * sz; p->dR[0][1][1] = -cx * sy * sz - sx * cz; p->dR[0][1][2] = -cx * cy;
  p->dR[0][2][0] = +sx * sy * cz + cx * sz; p->dR[0][2][1] = -sx * sy * sz + cx * cz; p->dR[0][2][2] = -sx * cy;
  /* d/dry */
  p->dR[1][0][0] = -sy * cz;      p->dR[1][0][1] = +sy * sz;      p->dR[1][0][2] = +cy;
  p->dR[1][1][0] = +sx * cy * cz; p->dR[1][1][1] = -sx * cy * sz; p->dR[1][1][2] = +sx * sy;
  p->dR[1][2][0] = -cx * cy * cz; p->dR[1][2][1] = +cx * cy * sz; p->dR[1][2][2] = -cx * sy;
  /* d/drz (third column is identically zero) */
  p->dR[2][0][0] = -cy * sz;                p->dR[2][0][1] = -cy * cz;                p->dR[2][0][2] = 0;
  p->dR[2][1][0] = -sx * sy * sz + cx * cz; p->dR[2][1][1] = -sx * sy * cz - cx * sz; p->dR[2][1][2] = 0;
  p->dR[2][2][0] = +cx * sy * sz + sx * cz; p->dR[2][2][1] = +cx * sy * cz - sx * sz; p->dR[2][2][2] = 0;
  p->t[0] = tr[3];
  p->t[1] = tr[4];
  p->t[2] = tr[5];
}

typedef struct {
  const vo_match *m;
  int32_t n;
  const vo_ego_params *ep;
  double *X, *Y, *Z; /* back-projected previous-frame points, viso_stereo.cpp:70-75 */
} ego_ctx;

/* camera-frame point, viso_stereo.cpp:257-259 */
static void transform_point(const pose6 *p, const ego_ctx *c, int32_t idx, double out[3]) {
  double X = c->X[idx], Y = c->Y[idx], Z = c->Z[idx];
  out[0] = p->R[0][0] * X + p->R[0][1] * Y + p->R[0][2] * Z + p->t[0];
  out[1] = p->R[1][0] * X + p->R[1][1] * Y + p->R[1][2] * Z + p->t[1];
  out[2] = p->R[2][0] * X + p->R[2][1] * Y + p->R[2][2] * Z + p->t[2];
}

/* predictions in the order (u left, v left, u right, v right), viso_stereo.cpp:300-303 */
static void predict4(const ego_ctx *c, const double P[3], double out[4]) {
  const vo_ego_params *e = c->ep;
  double Xr = P[0] - e->base;
  out[0] = e->f * P[0] / P[2] + e->cu;
  out[1] = e->f * P[1] / P[2] + e->cv;
  out[2] = e->f * Xr / P[2] + e->cu;
  out[3] = e->f * P[1] / P[2] + e->cv;
}

static void observe4(const vo_match *m, double out[4]) { /* viso_stereo.cpp:211-214 */
  out[0] = m->u1c;
  out[1] = m->v1c;
  out[2] = m->u2c;
  out[3] = m->v2c;
}

/* getInlier, viso_stereo.cpp:148-166 (the Jacobian it also fills is never read) */
static int32_t collect_inliers(const ego_ctx *c, const double tr[6], int32_t *out) {
  pose6 p;
  pose_from_vector(tr, &p);
  double thr2 = c->ep->inlier_threshold * c->ep->inlier_threshold;
  int32_t k = 0;
  for (int32_t i = 0; i < c->n; i++) {
    double P[3], pr[4], ob[4];
    transform_point(&p, c, i, P);
    predict4(c, P, pr);
    observe4(&c->m[i], ob);
    double d0 = ob[0] - pr[0], d1 = ob[1] - pr[1], d2 = ob[2] - pr[2], d3 = ob[3] - pr[3];
    if (d0 * d0 + d1 * d1 + d2 * d2 + d3 * d3 < thr2) out[k++] = i;
  }
  return k;
}

enum { EGO_UPDATED = 0, EGO_FAILED = 1, EGO_CONVERGED = 2 };

/* one Gauss-Newton step over the `na` active matches, viso_stereo.cpp:168-206 + 217-315.
 * Jw holds the 4*na x 6 Jacobian, rw the 4*na residuals (both caller-provided scratch). */
static int gn_step(const ego_ctx *c, const int32_t *active, int32_t na, double tr[6], double eps, double *Jw,
                   double *rw) {
  if (na < 3) return EGO_FAILED;
  const vo_ego_params *e = c->ep;
  pose6 p;
  pose_from_vector(tr, &p);
  for (int32_t i = 0; i < na; i++) {
    int32_t idx = active[i];
    double P[3], pr[4], ob[4];
    transform_point(&p, c, idx, P);
    observe4(&c->m[idx], ob);
    double w = 1.0;
    if (e->reweighting) w = 1.0 / (fabs(ob[0] - e->cu) / fabs(e->cu) + 0.05);
    double Xr = P[0] - e->base;
    double X = c->X[idx], Y = c->Y[idx], Z = c->Z[idx];
    for (int32_t j = 0; j < 6; j++) {
      double dX, dY, dZ;
      if (j == 0) {
        dX = 0;
        dY = p.dR[0][1][0] * X + p.dR[0][1][1] * Y + p.dR[0][1][2] * Z;
        dZ = p.dR[0][2][0] * X + p.dR[0][2][1] * Y + p.dR[0][2][2] * Z;
      } else if (j == 1) {
        dX = p.dR[1][0][0] * X + p.dR[1][0][1] * Y + p.dR[1][0][2] * Z;
        dY = p.dR[1][1][0] * X + p.dR[1][1][1] * Y + p.dR[1][1][2] * Z;
        dZ = p.dR[1][2][0] * X + p.dR[1][2][1] * Y + p.dR[1][2][2] * Z;
      } else if (j == 2) {
        dX = p.dR[2][0][0] * X + p.dR[2][0][1] * Y;
        dY = p.dR[2][1][0] * X + p.dR[2][1][1] * Y;
        dZ = p.dR[2][2][0] * X + p.dR[2][2][1] * Y;
      } else {
        dX = j == 3 ? 1 : 0;
        dY = j == 4 ? 1 : 0;
        dZ = j == 5 ? 1 : 0;
      }
      double zz = P[2] * P[2];
      Jw[(4 * i + 0) * 6 + j] = w * e->f * (dX * P[2] - P[0] * dZ) / zz;
      Jw[(4 * i + 1) * 6 + j] = w * e->f * (dY * P[2] - P[1] * dZ) / zz;
      Jw[(4 * i + 2) * 6 + j] = w * e->f * (dX * P[2] - Xr * dZ) / zz;
      Jw[(4 * i + 3) * 6 + j] = w * e->f * (dY * P[2] - P[1] * dZ) / zz;
    }
    predict4(c, P, pr);
    for (int32_t q = 0; q < 4; q++) rw[4 * i + q] = w * (ob[q] - pr[q]);
  }
  /* normal equations, viso_stereo.cpp:183-196 */
  double A[6][6], B[6];
  int32_t rows = 4 * na;
  for (int32_t m = 0; m < 6; m++) {
    for (int32_t n = 0; n < 6; n++) {
      double s = 0;
      for (int32_t i = 0; i < rows; i++) s += Jw[i * 6 + m] * Jw[i * 6 + n];
      A[m][n] = s;
    }
    double s = 0;
    for (int32_t i = 0; i < rows; i++) s += Jw[i * 6 + m] * rw[i];
    B[m] = s;
  }
  if (!gj6(A, B)) return EGO_FAILED;
  int converged = 1;
  for (int32_t m = 0; m < 6; m++) {
    tr[m] += 1.0 * B[m];
    if (fabs(B[m]) > eps) converged = 0;
  }
  return converged ? EGO_CONVERGED : EGO_UPDATED;
}

static int gn_run(const ego_ctx *c, const int32_t *active, int32_t na, double tr[6], double eps, int32_t cap,
                  double *Jw, double *rw) {
  /* while (result==UPDATED) { step; if (iter++ > cap || CONVERGED) break; }   viso_stereo.cpp:100-104,121-125 */
  int result = EGO_UPDATED;
  int32_t iter = 0;
  while (result == EGO_UPDATED) {
    result = gn_step(c, active, na, tr, eps, Jw, rw);
    if (iter++ > cap || result == EGO_CONVERGED) break;
  }
  return result;
}

void vo_ego_default_params(vo_ego_params *e) { /* viso.h:33-42, viso_stereo.h:38-43 */
  e->f = 1;
  e->cu = 0;
  e->cv = 0;
  e->base = 1.0;
  e->ransac_iters = 200;
  e->inlier_threshold = 2.0;
  e->reweighting = 1;
}

/* estimateMotion, viso_stereo.cpp:42-146.
 * returns 1 and fills tr6 on success; 0 on failure; -1 when n < 6 (the reference then returns
 * before it clears its inlier list, so *n_inliers is left untouched in that case). */
int32_t vo_estimate_motion_stereo(const vo_match *m, int32_t n, const vo_ego_params *ep, double *tr6, int32_t *inliers,
                                  int32_t *n_inliers) {
  if (n < 6) return -1;
  ego_ctx c;
  c.m = m;
  c.n = n;
  c.ep = ep;
  c.X = (double *)malloc(sizeof(double) * 3 * (size_t)n);
  c.Y = c.X + n;
  c.Z = c.Y + n;
  double *Jw = (double *)malloc(sizeof(double) * 24 * (size_t)n);
  double *rw = (double *)malloc(sizeof(double) * 4 * (size_t)n);
  int32_t *scratch = (int32_t *)malloc(sizeof(int32_t) * 2 * (size_t)n);
  int32_t *cur = scratch + n;
  for (int32_t i = 0; i < n; i++) {
    float df = m[i].u1p - m[i].u2p; /* float max against 0.0001f, then widened */
    if (!(df > 0.0001f)) df = 0.0001f;
    double d = df;
    c.X[i] = (m[i].u1p - ep->cu) * ep->base / d;
    c.Y[i] = (m[i].v1p - ep->cv) * ep->base / d;
    c.Z[i] = ep->f * ep->base / d;
  }
  int32_t best_n = 0;
  double best_tr[6] = {0, 0, 0, 0, 0, 0};
  int have_best = 0;
  for (int32_t k = 0; k < ep->ransac_iters; k++) {
    int32_t act[3];
    sample_k(scratch, n, 3, act);
    double tr[6] = {0, 0, 0, 0, 0, 0};
    int res = gn_run(&c, act, 3, tr, 1e-6, 20, Jw, rw);
    if (res != EGO_FAILED) {
      int32_t cn = collect_inliers(&c, tr, cur);
      if (cn > best_n) {
        best_n = cn;
        memcpy(inliers, cur, sizeof(int32_t) * (size_t)cn);
        memcpy(best_tr, tr, sizeof(best_tr));
        have_best = 1;
      }
    }
  }
  *n_inliers = best_n;
  int ok = 0;
  if (best_n >= 6 && have_best) {
    int res = gn_run(&c, inliers, best_n, best_tr, 1e-8, 100, Jw, rw);
    ok = res == EGO_CONVERGED;
  }
  if (ok) memcpy(tr6, best_tr, sizeof(best_tr));
  free(c.X);
  free(Jw);
  free(rw);
  free(scratch);
  return ok;
}

/* transformationVectorToMatrix, viso/viso.cpp:60-89 (row-major 4x4) */
void vo_tr_vector_to_matrix(const double *tr6, double *T16) {
  pose6 p;
  pose_from_vector(tr6, &p);
  for (int32_t r = 0; r < 3; r++) {
    for (int32_t q = 0; q < 3; q++) T16[r * 4 + q] = p.R[r][q];
    T16[r * 4 + 3] = p.t[r];
  }
  T16[12] = 0;
  T16[13] = 0;
  T16[14] = 0;
  T16[15] = 1;
}

/* ------------------------------------------------------------------------------------------ */
/* VisualOdometryStereo as an object (viso/viso.cpp:28-58, viso/viso_stereo.cpp:27-40)        */
/* ------------------------------------------------------------------------------------------ */
struct vo_stereo {
  vo_matcher *matcher;
  vo_ego_params ep;
  int32_t bucket_max;
  double bucket_w, bucket_h;
  double T[16];
  int32_t tr_valid;
  vo_match *matched;
  int32_t n_matched, cap_matched;
  int32_t *inliers;
  int32_t n_inliers, cap_inliers;
};

vo_stereo *vo_stereo_create(const vo_params *mp, int32_t bucket_max, double bucket_w, double bucket_h,
                            const vo_ego_params *ep) {
  vo_stereo *v = (vo_stereo *)calloc(1, sizeof(vo_stereo));
  v->matcher = vo_create(mp);
  v->ep = *ep;
  v->bucket_max = bucket_max;
  v->bucket_w = bucket_w;
  v->bucket_h = bucket_h;
  for (int32_t i = 0; i < 16; i++) v->T[i] = (i % 5 == 0) ? 1.0 : 0.0;
  srand(0); /* viso/viso.cpp:35 */
  vo_set_intrinsics(v->matcher, ep->f, ep->cu, ep->cv, ep->base);
  return v;
}

void vo_stereo_destroy(vo_stereo *v) {
  if (!v) return;
  vo_destroy(v->matcher);
  free(v->matched);
  free(v->inliers);
  free(v);
}

/* updateMotion on the current v->matched list */
static int32_t stereo_update_motion(vo_stereo *v) {
  if (v->n_matched > v->cap_inliers) {
    v->cap_inliers = v->n_matched;
    v->inliers = (int32_t *)realloc(v->inliers, sizeof(int32_t) * (size_t)v->cap_inliers);
  }
  double tr[6];
  int32_t rc = vo_estimate_motion_stereo(v->matched, v->n_matched, &v->ep, tr, v->inliers, &v->n_inliers);
  if (rc != 1) return 0;
  vo_tr_vector_to_matrix(tr, v->T);
  v->tr_valid = 1;
  return 1;
}

static void stereo_set_matched(vo_stereo *v, int32_t n) {
  if (n > v->cap_matched) {
    v->cap_matched = n;
    v->matched = (vo_match *)realloc(v->matched, sizeof(vo_match) * (size_t)n);
  }
  v->n_matched = n;
}

int32_t vo_stereo_process(vo_stereo *v, const uint8_t *I1, const uint8_t *I2, int32_t w, int32_t h, int32_t bpl,
                          int32_t replace) {
  vo_push_back(v->matcher, I1, I2, w, h, bpl, replace);
  vo_match_features(v->matcher, 2, v->tr_valid ? v->T : 0);
  vo_bucket_features(v->matcher, v->bucket_max, (float)v->bucket_w, (float)v->bucket_h);
  stereo_set_matched(v, vo_num_matches(v->matcher));
  if (v->n_matched) vo_get_matches(v->matcher, v->matched);
  return stereo_update_motion(v);
}

/* VisualOdometry::process(std::vector<p_match>), viso/viso.h:74-77 */
int32_t vo_stereo_process_matches(vo_stereo *v, const vo_match *m, int32_t n) {
  stereo_set_matched(v, n);
  if (n) memcpy(v->matched, m, sizeof(vo_match) * (size_t)n);
  return stereo_update_motion(v);
}

void vo_stereo_get_motion(const vo_stereo *v, double *T16) { memcpy(T16, v->T, sizeof(v->T)); }
int32_t vo_stereo_tr_valid(const vo_stereo *v) { return v->tr_valid; }
int32_t vo_stereo_num_matches(const vo_stereo *v) { return v->n_matched; }
void vo_stereo_get_matches(const vo_stereo *v, vo_match *out) {
  if (v->n_matched) memcpy(out, v->matched, sizeof(vo_match) * (size_t)v->n_matched);
}
int32_t vo_stereo_num_inliers(const vo_stereo *v) { return v->n_inliers; }
void vo_stereo_get_inliers(const vo_stereo *v, int32_t *out) {
  if (v->n_inliers) memcpy(out, v->inliers, sizeof(int32_t) * (size_t)v->n_inliers);
}
vo_matcher *vo_stereo_matcher(vo_stereo *v) { return v->matcher; }
