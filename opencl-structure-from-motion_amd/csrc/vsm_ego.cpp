// Stereo egomotion on top of the matcher: VisualOdometryStereo::process and what it calls
// (viso/viso_stereo.cpp:33-315, viso/viso.cpp:28-108), SURVEY.md section 8 row f-2.
//
// Why this runs on the host cores and not on the GPU: the result has to equal the reference's
// bit for bit because Tr_delta feeds the next frame's match prediction, and every Gauss-Newton
// step takes sin/cos of the running estimate -- the device math library rounds those differently
// from the host libm the reference is built on.  The work is small (200 three-point fits + 200
// inlier counts over a few hundred bucketed matches) and embarrassingly parallel over the RANSAC
// hypotheses, so it is spread over the matcher's host pool instead:
//   * the 200 samples are drawn first, sequentially, from the process-wide sampler;
//   * hypotheses are fitted and scored independently (structure-of-arrays scene, one pass per
//     hypothesis, no Jacobian for the scoring);
//   * the winner is the first hypothesis with strictly more inliers than all before it, which is
//     what the reference's sequential loop keeps; only its inlier list is materialised.
// Sums keep the reference's order (rows ascending), products are never fused (-ffp-contract=off).
#include <math.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <atomic>
#include <mutex>
#include <vector>

#include "vsm_host.h"

namespace {

// ---- VisualOdometry::getRandomSample's engine (viso/viso.cpp:93): std::default_random_engine
// of libstdc++ is minstd_rand0; one instance per process, seeded 71, shared by all VO objects ----
struct Sampler {
  std::mutex mu;
  uint32_t state = 71;
  uint64_t next() {
    state = (uint32_t)(((uint64_t)state * 16807ull) % 2147483647ull);
    return state;
  }
  // std::uniform_int_distribution<unsigned>(lo, hi) over an engine whose range is [1, 2^31-2]
  // (libstdc++ bits/uniform_int_dist.h, the generic down-scaling branch with rejection)
  uint32_t between(uint32_t lo, uint32_t hi) {
    const uint64_t engine_span = 2147483645ull;
    const uint64_t want = (uint64_t)hi - lo;
    uint64_t v;
    if (engine_span > want) {
      const uint64_t cells = want + 1, per_cell = engine_span / cells, reject_from = cells * per_cell;
      do v = next() - 1;
      while (v >= reject_from);
      v /= per_cell;
    } else {
      v = next() - 1;
    }
    return (uint32_t)(v + lo);
  }
  // the same draw with the two divisions of a (lo, hi) pair hoisted out: RANSAC draws from the same
  // few ranges thousands of times.  v / per_cell = (v * magic) >> 64 exactly for v < 2^32.
  uint32_t between(const VsmDrawPlan &p) {
    uint64_t v;
    do v = next() - 1;
    while (v >= p.reject_from);
    if (p.per_cell) v = (uint64_t)(((__uint128_t)v * p.magic) >> 64);
    return (uint32_t)(v + p.lo);
  }
};
Sampler g_sampler;

struct Pose {
  double r[9];      // rotation, row-major
  double dx[6];     // rows 1,2 of dR/drx (row 0 is zero)
  double dy[9];     // dR/dry
  double dz[6];     // columns 0,1 of dR/drz, rows 0..2 (column 2 is zero)
  double t[3];
  explicit Pose(const double *tr) {
    const double sx = sin(tr[0]), cx = cos(tr[0]), sy = sin(tr[1]), cy = cos(tr[1]), sz = sin(tr[2]), cz = cos(tr[2]);
    r[0] = +cy * cz;                r[1] = -cy * sz;                r[2] = +sy;
    r[3] = +sx * sy * cz + cx * sz; r[4] = -sx * sy * sz + cx * cz; r[5] = -sx * cy;
    r[6] = -cx * sy * cz + sx * sz; r[7] = +cx * sy * sz + sx * cz; r[8] = +cx * cy;
    dx[0] = +cx * sy * cz - sx * sz; dx[1] = -cx * sy * sz - sx * cz; dx[2] = -cx * cy;
    dx[3] = +sx * sy * cz + cx * sz; dx[4] = -sx * sy * sz + cx * cz; dx[5] = -sx * cy;
    dy[0] = -sy * cz;      dy[1] = +sy * sz;      dy[2] = +cy;
    dy[3] = +sx * cy * cz; dy[4] = -sx * cy * sz; dy[5] = +sx * sy;
    dy[6] = -cx * cy * cz; dy[7] = +cx * cy * sz; dy[8] = -cx * sy;
    dz[0] = -cy * sz;                dz[1] = -cy * cz;
    dz[2] = -sx * sy * sz + cx * cz; dz[3] = -sx * sy * cz - cx * sz;
    dz[4] = +cx * sy * sz + sx * cz; dz[5] = +cx * sy * cz - sx * sz;
    t[0] = tr[3];
    t[1] = tr[4];
    t[2] = tr[5];
  }
};

// Matrix::solve (viso/matrix.cpp:424-513) for 6 unknowns and one right-hand side
bool solve6(double (&a)[6][6], double (&b)[6]) {
  int taken[6] = {0, 0, 0, 0, 0, 0};
  for (int round = 0; round < 6; round++) {
    double big = 0.0;
    int row = 0, col = 0;
    for (int j = 0; j < 6; j++)
      if (taken[j] != 1)
        for (int k = 0; k < 6; k++)
          if (taken[k] == 0 && fabs(a[j][k]) >= big) {
            big = fabs(a[j][k]);
            row = j;
            col = k;
          }
    ++taken[col];
    if (row != col) {
      for (int l = 0; l < 6; l++) std::swap(a[row][l], a[col][l]);
      std::swap(b[row], b[col]);
    }
    if (fabs(a[col][col]) < 1e-20) return false;
    const double scale = 1.0 / a[col][col];
    a[col][col] = 1.0;
    for (int l = 0; l < 6; l++) a[col][l] *= scale;
    b[col] *= scale;
    for (int other = 0; other < 6; other++) {
      if (other == col) continue;
      const double m = a[other][col];
      a[other][col] = 0.0;
      for (int l = 0; l < 6; l++) a[other][l] -= a[col][l] * m;
      b[other] -= b[col] * m;
    }
  }
  return true;
}

enum Step { UPDATED, FAILED, CONVERGED };

struct Camera {
  double f, cu, cv, base;
};

// number of matches in [lo,hi) whose four reprojections lie within sqrt(limit) pixels of the
// observations (getInlier, viso_stereo.cpp:148-166).  Plain IEEE double arithmetic lane by lane, so
// the wider clones count exactly what the scalar one counts.
__attribute__((target_clones("avx512f", "avx2", "default")))
int count_within(const double *__restrict X, const double *__restrict Y, const double *__restrict Z,
                 const double *__restrict ou1, const double *__restrict ov1, const double *__restrict ou2,
                 const double *__restrict ov2, int lo, int hi, const double *__restrict r, const double *__restrict t,
                 Camera c, double limit) {
  int count = 0;
  for (int i = lo; i < hi; i++) {
    const double xc = r[0] * X[i] + r[1] * Y[i] + r[2] * Z[i] + t[0];
    const double yc = r[3] * X[i] + r[4] * Y[i] + r[5] * Z[i] + t[1];
    const double zc = r[6] * X[i] + r[7] * Y[i] + r[8] * Z[i] + t[2];
    const double xr = xc - c.base;
    const double e0 = ou1[i] - (c.f * xc / zc + c.cu);
    const double e1 = ov1[i] - (c.f * yc / zc + c.cv);
    const double e2 = ou2[i] - (c.f * xr / zc + c.cu);
    const double e3 = ov2[i] - (c.f * yc / zc + c.cv);
    count += (e0 * e0 + e1 * e1 + e2 * e2 + e3 * e3 < limit) ? 1 : 0;
  }
  return count;
}

class EgoStereo {
 public:
  vsm_vo_stereo_params par;

  // scene of one estimate call, structure of arrays
  int n = 0;
  std::vector<double> X, Y, Z, ou1, ov1, ou2, ov2, wgt;

  void load(const vsm_p_match *m, int count) {
    n = count;
    for (auto *v : {&X, &Y, &Z, &ou1, &ov1, &ou2, &ov2, &wgt}) v->resize((size_t)n);
    for (int i = 0; i < n; i++) {
      const float disp = std::max(m[i].u1p - m[i].u2p, 0.0001f);  // viso_stereo.cpp:71
      const double d = disp;
      X[i] = (m[i].u1p - par.cu) * par.base / d;
      Y[i] = (m[i].v1p - par.cv) * par.base / d;
      Z[i] = par.f * par.base / d;
      ou1[i] = m[i].u1c;
      ov1[i] = m[i].v1c;
      ou2[i] = m[i].u2c;
      ov2[i] = m[i].v2c;
      wgt[i] = par.reweighting ? 1.0 / (fabs(ou1[i] - par.cu) / fabs(par.cu) + 0.05) : 1.0;  // :262-264
    }
  }

  // inlier count of a hypothesis.  `floor` is a count some other hypothesis has already reached:
  // once this one cannot reach it any more it cannot be the (first) maximum, and -2 is returned.
  int score(const double *tr, int floor) const {
    const Pose p(tr);
    const Camera c = {par.f, par.cu, par.cv, par.base};
    const double limit = par.inlier_threshold * par.inlier_threshold;
    int count = 0;
    for (int lo = 0; lo < n; lo += 128) {
      const int hi = std::min(n, lo + 128);
      count += count_within(X.data(), Y.data(), Z.data(), ou1.data(), ov1.data(), ou2.data(), ov2.data(), lo, hi, p.r, p.t,
                            c, limit);
      if (count + (n - hi) < floor) return -2;
    }
    return count;
  }

  // the inlier indices themselves (only needed for the winning hypothesis)
  int collect(const double *tr, int32_t *list) const {
    const Pose p(tr);
    const Camera c = {par.f, par.cu, par.cv, par.base};
    const double limit = par.inlier_threshold * par.inlier_threshold;
    int count = 0;
    for (int i = 0; i < n; i++)
      if (count_within(X.data(), Y.data(), Z.data(), ou1.data(), ov1.data(), ou2.data(), ov2.data(), i, i + 1, p.r, p.t, c,
                       limit))
        list[count++] = i;
    return count;
  }

  // one Gauss-Newton update over the matches act[0..na) (updateParameters, :168-206).  The 6x6
  // normal matrix is symmetric and products commute, so 21 + 6 running sums filled in one pass
  // over the rows give the same bits as the reference's 36 + 6 separate passes.
  Step update(const int32_t *act, int na, double *tr, double eps) const {
    if (na < 3) return FAILED;
    const Pose p(tr);
    const double f = par.f, cu = par.cu, cv = par.cv, base = par.base;
    double upper[21], rhs[6];
    for (double &v : upper) v = 0;
    for (double &v : rhs) v = 0;
    for (int a = 0; a < na; a++) {
      const int i = act[a];
      const double x = X[i], y = Y[i], z = Z[i];
      const double xc = p.r[0] * x + p.r[1] * y + p.r[2] * z + p.t[0];
      const double yc = p.r[3] * x + p.r[4] * y + p.r[5] * z + p.t[1];
      const double zc = p.r[6] * x + p.r[7] * y + p.r[8] * z + p.t[2];
      const double xr = xc - base, w = wgt[i], zz = zc * zc;
      // derivative of the camera-frame point with respect to each parameter (:270-289)
      double gx[6], gy[6], gz[6];
      gx[0] = 0;
      gy[0] = p.dx[0] * x + p.dx[1] * y + p.dx[2] * z;
      gz[0] = p.dx[3] * x + p.dx[4] * y + p.dx[5] * z;
      gx[1] = p.dy[0] * x + p.dy[1] * y + p.dy[2] * z;
      gy[1] = p.dy[3] * x + p.dy[4] * y + p.dy[5] * z;
      gz[1] = p.dy[6] * x + p.dy[7] * y + p.dy[8] * z;
      gx[2] = p.dz[0] * x + p.dz[1] * y;
      gy[2] = p.dz[2] * x + p.dz[3] * y;
      gz[2] = p.dz[4] * x + p.dz[5] * y;
      gx[3] = 1; gy[3] = 0; gz[3] = 0;
      gx[4] = 0; gy[4] = 1; gz[4] = 0;
      gx[5] = 0; gy[5] = 0; gz[5] = 1;
      double jac[4][6], res[4];
      for (int j = 0; j < 6; j++) {
        jac[0][j] = w * f * (gx[j] * zc - xc * gz[j]) / zz;
        jac[1][j] = w * f * (gy[j] * zc - yc * gz[j]) / zz;
        jac[2][j] = w * f * (gx[j] * zc - xr * gz[j]) / zz;
        jac[3][j] = w * f * (gy[j] * zc - yc * gz[j]) / zz;
      }
      res[0] = w * (ou1[i] - (f * xc / zc + cu));
      res[1] = w * (ov1[i] - (f * yc / zc + cv));
      res[2] = w * (ou2[i] - (f * xr / zc + cu));
      res[3] = w * (ov2[i] - (f * yc / zc + cv));
      for (int q = 0; q < 4; q++) {
        int k = 0;
        for (int m = 0; m < 6; m++) {
          for (int c = m; c < 6; c++) upper[k++] += jac[q][m] * jac[q][c];
          rhs[m] += jac[q][m] * res[q];
        }
      }
    }
    double A[6][6], B[6];
    int k = 0;
    for (int m = 0; m < 6; m++)
      for (int c = m; c < 6; c++) A[m][c] = A[c][m] = upper[k++];
    for (int m = 0; m < 6; m++) B[m] = rhs[m];
    if (!solve6(A, B)) return FAILED;
    bool settled = true;
    for (int m = 0; m < 6; m++) {
      tr[m] += B[m];
      if (fabs(B[m]) > eps) settled = false;
    }
    return settled ? CONVERGED : UPDATED;
  }

  // the reference's "while (UPDATED) { step; if (iter++ > cap || CONVERGED) break; }" (:100-104, :121-125)
  Step iterate(const int32_t *act, int na, double *tr, double eps, int cap) const {
    Step s = UPDATED;
    for (int iter = 0; s == UPDATED; iter++) {
      s = update(act, na, tr, eps);
      if (iter > cap || s == CONVERGED) break;
    }
    return s;
  }

  struct Hypothesis {
    int32_t pick[3];
    double tr[6];
    int score;  // -1: the fit failed
  };
  std::vector<Hypothesis> hyp;
  std::vector<int32_t> deck;

  // estimateMotion (:42-146).  1 = success (tr6 set), 0 = failure, -1 = fewer than 6 matches
  // (the reference returns before clearing its inlier list in that case, so `inliers` is kept).
  Sampler *sampler = &g_sampler;  // (the process's sampler; the lock-step multi-sequence API gives every sequence its own)
  template <class Runner>
  int estimate(const vsm_p_match *m, int count, Runner *pool, double *tr6, std::vector<int32_t> &inliers) {
    if (count < 6) return -1;
    load(m, count);
    const int iters = std::max(par.ransac_iters, 0);
    hyp.resize((size_t)iters);
    deck.resize((size_t)n);
    {
      std::lock_guard<std::mutex> lock(sampler->mu);
      const VsmDrawPlan plan[3] = {vsm_sampler_plan(0, (uint32_t)(n - 1)), vsm_sampler_plan(1, (uint32_t)(n - 1)),
                                   vsm_sampler_plan(2, (uint32_t)(n - 1))};
      for (int i = 0; i < n; i++) deck[i] = i;
      for (int k = 0; k < iters; k++) {  // partial shuffle of 0..n-1, first three (viso.cpp:96-105), undone afterwards
        int swapped[3];
        for (int i = 0; i < 3; i++) {
          swapped[i] = (int)sampler->between(plan[i]);
          std::swap(deck[i], deck[swapped[i]]);
        }
        for (int i = 0; i < 3; i++) hyp[k].pick[i] = deck[i];
        for (int i = 2; i >= 0; i--) std::swap(deck[i], deck[swapped[i]]);
      }
    }
    std::atomic<int> reached{0};  // best count seen so far by any hypothesis (only ever a lower bound)
    auto fit = [&](int k) {
      Hypothesis &h = hyp[k];
      for (double &v : h.tr) v = 0;
      if (iterate(h.pick, 3, h.tr, 1e-6, 20) == FAILED) {
        h.score = -1;
        return;
      }
      h.score = score(h.tr, reached.load(std::memory_order_relaxed));
      int seen = reached.load(std::memory_order_relaxed);
      while (h.score > seen && !reached.compare_exchange_weak(seen, h.score, std::memory_order_relaxed)) {
      }
    };
    const int lanes = pool ? std::min(pool->size(), 16) : 1;
    if (lanes > 1 && iters >= 2 * lanes) {
      const int per = (iters + lanes * 2 - 1) / (lanes * 2);
      const int tasks = (iters + per - 1) / per;
      pool->run(tasks, [&](int t) {
        const int hi = std::min(iters, (t + 1) * per);
        for (int k = t * per; k < hi; k++) fit(k);
      });
    } else {
      for (int k = 0; k < iters; k++) fit(k);
    }
    int best = -1, best_score = 0;
    for (int k = 0; k < iters; k++)
      if (hyp[k].score > best_score) {
        best_score = hyp[k].score;
        best = k;
      }
    inliers.clear();
    if (best < 0) return 0;
    inliers.resize((size_t)n);
    inliers.resize((size_t)collect(hyp[best].tr, inliers.data()));
    if (inliers.size() < 6) return 0;
    double tr[6];
    memcpy(tr, hyp[best].tr, sizeof(tr));
    if (iterate(inliers.data(), (int)inliers.size(), tr, 1e-8, 100) != CONVERGED) return 0;
    memcpy(tr6, tr, sizeof(tr));
    return 1;
  }
};

void pose_matrix(const double *tr, double *T) {  // transformationVectorToMatrix, viso/viso.cpp:60-89
  const Pose p(tr);
  for (int r = 0; r < 3; r++) {
    for (int c = 0; c < 3; c++) T[r * 4 + c] = p.r[r * 3 + c];
    T[r * 4 + 3] = p.t[r];
  }
  T[12] = T[13] = T[14] = 0;
  T[15] = 1;
}

}  // namespace

void vsm_sampler_lock() { g_sampler.mu.lock(); }
void vsm_sampler_unlock() { g_sampler.mu.unlock(); }
uint32_t vsm_sampler_between(uint32_t lo, uint32_t hi) { return g_sampler.between(lo, hi); }
VsmDrawPlan vsm_sampler_plan(uint32_t lo, uint32_t hi) {
  VsmDrawPlan p;
  const uint64_t engine_span = 2147483645ull, want = (uint64_t)hi - lo;
  p.lo = lo;
  if (engine_span > want) {
    const uint64_t cells = want + 1;
    p.per_cell = engine_span / cells;
    p.reject_from = cells * p.per_cell;
    p.magic = p.per_cell > 1 ? UINT64_MAX / p.per_cell + 1 : 0;
    if (p.per_cell == 1) p.per_cell = 0;  // division by one: nothing to do
  } else {
    p.per_cell = 0;
    p.reject_from = UINT64_MAX;
    p.magic = 0;
  }
  return p;
}
uint32_t vsm_sampler_draw(const VsmDrawPlan &p) { return g_sampler.between(p); }
void vsm_pose_matrix(const double *tr6, double *T16) { pose_matrix(tr6, T16); }

struct vsm_vo_stereo {
  vsm_handle *matcher = nullptr;
  EgoStereo ego;
  double T[16];
  bool valid = false;
  std::vector<vsm_p_match> matched;  // VisualOdometry::p_matched (bucketed)
  std::vector<int32_t> inliers;
  double timings[4] = {0, 0, 0, 0};
};

static int update_motion(vsm_vo_stereo *v) {  // VisualOdometry::updateMotion, viso/viso.cpp:42-58
  double tr[6];
  const int rc = v->ego.estimate(v->matched.data(), (int)v->matched.size(), vsm_forkjoin_of(v->matcher), tr, v->inliers);
  if (rc != 1) return 0;
  pose_matrix(tr, v->T);
  v->valid = true;
  return 1;
}

static int after_push(vsm_vo_stereo *v) {  // viso/viso_stereo.cpp:35-39
  const double t0 = vsm_now_us();
  vsm_match(v->matcher, 2, v->valid ? v->T : nullptr);
  const double t1 = vsm_now_us();
  vsm_bucket(v->matcher, v->ego.par.bucket_max_features, (float)v->ego.par.bucket_width, (float)v->ego.par.bucket_height);
  v->matched.resize((size_t)vsm_num_matches(v->matcher));
  if (!v->matched.empty()) vsm_get_matches(v->matcher, v->matched.data(), (int32_t)v->matched.size());
  const double t2 = vsm_now_us();
  const int ok = update_motion(v);
  const double t3 = vsm_now_us();
  v->timings[0] = t1 - t0;
  v->timings[1] = t2 - t1;
  v->timings[2] = t3 - t2;
  v->timings[3] = t3 - t0;
  return ok;
}

// ---- one sequence of the lock-step multi-sequence API ----
struct VsmEgoSeq {
  EgoStereo ego;
  Sampler sampler;    // seeded 71 like a fresh process of the reference
  VsmRandStream rnd;  // srand(0), viso/viso.cpp:35
};
VsmEgoSeq *vsm_ego_seq_create(const vsm_vo_stereo_params *p) {
  VsmEgoSeq *e = new VsmEgoSeq();
  e->ego.par = *p;
  e->ego.sampler = &e->sampler;
  e->rnd.seed(0);
  return e;
}
void vsm_ego_seq_destroy(VsmEgoSeq *e) { delete e; }
int vsm_ego_seq_step(VsmEgoSeq *e, std::vector<vsm_p_match> &matches, double *T16, bool *valid, std::vector<int32_t> &inliers, VsmPool *shared) {
  vsm_host_bucket_with(matches, e->ego.par.bucket_max_features, (float)e->ego.par.bucket_width, (float)e->ego.par.bucket_height, e->rnd);
  double tr[6];
  const int rc = shared ? e->ego.estimate(matches.data(), (int)matches.size(), shared, tr, inliers)
                        : e->ego.estimate(matches.data(), (int)matches.size(), (VsmForkJoin *)nullptr, tr, inliers);
  if (rc != 1) return 0;
  pose_matrix(tr, T16);
  *valid = true;
  return 1;
}

extern "C" {

void vsm_vo_stereo_default_params(vsm_vo_stereo_params *p) {
  memset(p, 0, sizeof(*p));
  vsm_default_params(&p->match);
  p->match.f = 1;
  p->match.base = 1;
  p->bucket_max_features = 2;
  p->bucket_width = 50;
  p->bucket_height = 50;
  p->f = 1;
  p->base = 1.0;
  p->ransac_iters = 200;
  p->inlier_threshold = 2.0;
  p->reweighting = 1;
}

vsm_vo_stereo *vsm_vo_stereo_create(const vsm_vo_stereo_params *p) {
  vsm_handle *m = vsm_create(&p->match);
  if (!m) return nullptr;
  vsm_vo_stereo *v = new vsm_vo_stereo();
  v->matcher = m;
  v->ego.par = *p;
  for (int i = 0; i < 16; i++) v->T[i] = (i % 5 == 0) ? 1.0 : 0.0;
  srand(0);  // viso/viso.cpp:35 (bucketing shuffles with rand())
  vsm_set_intrinsics(m, p->f, p->cu, p->cv, p->base);  // viso/viso_stereo.cpp:28
  return v;
}

void vsm_vo_stereo_destroy(vsm_vo_stereo *v) {
  if (!v) return;
  vsm_destroy(v->matcher);
  delete v;
}

int vsm_vo_stereo_process(vsm_vo_stereo *v, const uint8_t *I1, const uint8_t *I2, int32_t w, int32_t h, int32_t bpl,
                          int replace) {
  vsm_push_back(v->matcher, I1, I2, w, h, bpl, replace);  // a dims error is reported there; the reference carries on
  return after_push(v);
}

int vsm_vo_stereo_process_device(vsm_vo_stereo *v, const uint8_t *dI1, const uint8_t *dI2, int32_t w, int32_t h,
                                 int32_t bpl, int replace) {
  vsm_push_back_device(v->matcher, dI1, dI2, w, h, bpl, replace);
  return after_push(v);
}

int vsm_vo_stereo_process_matches(vsm_vo_stereo *v, const vsm_p_match *m, int32_t n) {
  v->matched.assign(m, m + (n > 0 ? n : 0));
  return update_motion(v);
}

void vsm_vo_stereo_get_motion(vsm_vo_stereo *v, double *T16) { memcpy(T16, v->T, sizeof(v->T)); }
int vsm_vo_stereo_motion_valid(vsm_vo_stereo *v) { return v->valid ? 1 : 0; }

int32_t vsm_vo_stereo_num_matches(vsm_vo_stereo *v) { return (int32_t)v->matched.size(); }
int32_t vsm_vo_stereo_get_matches(vsm_vo_stereo *v, vsm_p_match *out, int32_t cap) {
  const int32_t n = std::min((int32_t)v->matched.size(), cap);
  if (n > 0) memcpy(out, v->matched.data(), (size_t)n * sizeof(vsm_p_match));
  return n;
}
int32_t vsm_vo_stereo_num_inliers(vsm_vo_stereo *v) { return (int32_t)v->inliers.size(); }
int32_t vsm_vo_stereo_get_inliers(vsm_vo_stereo *v, int32_t *out, int32_t cap) {
  const int32_t n = std::min((int32_t)v->inliers.size(), cap);
  if (n > 0) memcpy(out, v->inliers.data(), (size_t)n * sizeof(int32_t));
  return n;
}
float vsm_vo_stereo_gain(vsm_vo_stereo *v, const int32_t *inliers, int32_t n) { return vsm_gain(v->matcher, inliers, n); }
vsm_handle *vsm_vo_stereo_matcher(vsm_vo_stereo *v) { return v->matcher; }
void vsm_vo_stereo_get_timings(vsm_vo_stereo *v, double *out4) { memcpy(out4, v->timings, sizeof(v->timings)); }

int32_t vsm_host_estimate_motion_stereo(const vsm_vo_stereo_params *p, const vsm_p_match *m, int32_t n, int32_t threads,
                                        double *tr6, double *T16, int32_t *inliers, int32_t *n_inliers) {
  EgoStereo ego;
  ego.par = *p;
  std::vector<int32_t> keep;
  int rc;
  if (threads > 1) {
    VsmPool pool(threads);
    rc = ego.estimate(m, n, &pool, tr6, keep);
  } else {
    rc = ego.estimate(m, n, (VsmPool *)nullptr, tr6, keep);
  }
  if (rc == 1 && T16) pose_matrix(tr6, T16);
  if (rc >= 0) {
    *n_inliers = (int32_t)keep.size();
    if (!keep.empty()) memcpy(inliers, keep.data(), keep.size() * sizeof(int32_t));
  }
  return rc;
}

void vsm_vo_sampler_seed(uint32_t s) {
  std::lock_guard<std::mutex> lock(g_sampler.mu);
  const uint32_t r = s % 2147483647u;
  g_sampler.state = r ? r : 1u;
}

}  // extern "C"
