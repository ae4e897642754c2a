/* TEST INFRASTRUCTURE ONLY -- CPU restatement ("oracle") of libviso2's monocular egomotion, the
 * part of dphoyes/OpenCL-Structure-from-Motion that the reference itself offloads to OpenCL
 * (SURVEY.md section 8, row f-4).  The definition followed here is the reference's CPU class (double
 * arithmetic); its OpenCL variant (viso/viso_mono_cl.cpp, viso/kernels/plane_and_inliers.cl)
 * computes the same two inner loops in float on whatever device runs it.
 *
 *   VisualOdometryMono::process / estimateMotion   viso/viso_mono.cpp:33-187
 *   ... ransacEstimateF :41-73, findBestPlane :75-101, smallerThanMedian :189-213,
 *       normalizeFeaturePoints :215-262, fundamentalMatrix :264-294, getInlier :296-344,
 *       EtoRt :346-392, triangulateChieral :394-431
 *   Matrix::svd (Numerical Recipes svdcmp + ordering + sign convention)  viso/matrix.cpp:586-850
 *   Matrix::lu / det :407-422, :521-580; operator* :270-284; operator/ :294-327
 *   VisualOdometry::getRandomSample  viso/viso.cpp:91-108 (shared sampler, see viso_ego_oracle.c)
 *
 * Parity status: PINNED by tests/test_mono_oracle.py (SVD / determinant / fundamental matrices /
 * whole estimates against oracle/_ref in a fresh process, and tests/golden/mono_cases.npz).
 */
#include "viso_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

uint32_t vo_ego_draw_between(uint32_t lo, uint32_t hi); /* viso_ego_oracle.c */

/* ------------------------------------------------------------------------------------------ */
/* small dense matrices, row-major                                                             */
/* ------------------------------------------------------------------------------------------ */
typedef struct {
  int m, n;
  double *a;
} dm;
#define AT(M, i, j) ((M).a[(size_t)(i) * (M).n + (j)])

static dm dm_new(int m, int n) {
  dm r;
  r.m = m;
  r.n = n;
  r.a = (double *)calloc((size_t)(m > 0 ? m : 1) * (size_t)(n > 0 ? n : 1), sizeof(double));
  return r;
}
static void dm_free(dm *x) {
  free(x->a);
  x->a = 0;
}
static dm dm_copy(const dm *x) {
  dm r = dm_new(x->m, x->n);
  memcpy(r.a, x->a, sizeof(double) * (size_t)x->m * (size_t)x->n);
  return r;
}
/* operator*: every entry starts at 0 and adds its products in k order (viso/matrix.cpp:279-282) */
static dm dm_mul(const dm *A, const dm *B) {
  dm C = dm_new(A->m, B->n);
  for (int i = 0; i < A->m; i++)
    for (int j = 0; j < B->n; j++) {
      double s = 0;
      for (int k = 0; k < A->n; k++) s += AT(*A, i, k) * AT(*B, k, j);
      AT(C, i, j) = s;
    }
  return C;
}
static dm dm_T(const dm *A) {
  dm C = dm_new(A->n, A->m);
  for (int i = 0; i < A->m; i++)
    for (int j = 0; j < A->n; j++) AT(C, j, i) = AT(*A, i, j);
  return C;
}
static dm dm_from(int m, int n, const double *v) {
  dm r = dm_new(m, n);
  memcpy(r.a, v, sizeof(double) * (size_t)m * (size_t)n);
  return r;
}
static dm dm_diag(const double *w, int n) {
  dm r = dm_new(n, n);
  for (int i = 0; i < n; i++) AT(r, i, i) = w[i];
  return r;
}

static double nr_pythag(double a, double b) { /* viso/matrix.cpp:852-860 */
  double absa = fabs(a), absb = fabs(b);
  if (absa > absb) {
    double q = absb / absa;
    return absa * sqrt(1.0 + (q == 0.0 ? 0.0 : q * q));
  }
  if (absb == 0.0) return 0.0;
  double q = absa / absb;
  return absb * sqrt(1.0 + (q == 0.0 ? 0.0 : q * q));
}
static double nr_sign(double a, double b) { return b >= 0.0 ? fabs(a) : -fabs(a); }

/* Matrix::svd.  U2: m x m, W: min(m,n) values, V: n x n.  Householder bidiagonalisation, implicit
 * shifted QR sweeps (at most 30 per singular value), then shell sort by decreasing singular value
 * and the sign convention "more positive than negative entries per (u,v) column pair". */
static void dm_svd(const dm *A, dm *U2, double *W, dm *V) {
  const int m = A->m, n = A->n;
  dm U = dm_copy(A);
  *U2 = dm_new(m, m);
  *V = dm_new(n, n);
  double *w = (double *)calloc((size_t)n, sizeof(double));
  double *rv1 = (double *)calloc((size_t)n, sizeof(double));
  int flag, i, its, j, jj, k, l = 0, nm = 0;
  double anorm = 0.0, c, f, g = 0.0, h, s, scale = 0.0, x, y, z;
  for (i = 0; i < n; i++) { /* bidiagonal form */
    l = i + 1;
    rv1[i] = scale * g;
    g = s = scale = 0.0;
    if (i < m) {
      for (k = i; k < m; k++) scale += fabs(AT(U, k, i));
      if (scale) {
        for (k = i; k < m; k++) {
          AT(U, k, i) /= scale;
          s += AT(U, k, i) * AT(U, k, i);
        }
        f = AT(U, i, i);
        g = -nr_sign(sqrt(s), f);
        h = f * g - s;
        AT(U, i, i) = f - g;
        for (j = l; j < n; j++) {
          for (s = 0.0, k = i; k < m; k++) s += AT(U, k, i) * AT(U, k, j);
          f = s / h;
          for (k = i; k < m; k++) AT(U, k, j) += f * AT(U, k, i);
        }
        for (k = i; k < m; k++) AT(U, k, i) *= scale;
      }
    }
    w[i] = scale * g;
    g = s = scale = 0.0;
    if (i < m && i != n - 1) {
      for (k = l; k < n; k++) scale += fabs(AT(U, i, k));
      if (scale) {
        for (k = l; k < n; k++) {
          AT(U, i, k) /= scale;
          s += AT(U, i, k) * AT(U, i, k);
        }
        f = AT(U, i, l);
        g = -nr_sign(sqrt(s), f);
        h = f * g - s;
        AT(U, i, l) = f - g;
        for (k = l; k < n; k++) rv1[k] = AT(U, i, k) / h;
        for (j = l; j < m; j++) {
          for (s = 0.0, k = l; k < n; k++) s += AT(U, j, k) * AT(U, i, k);
          for (k = l; k < n; k++) AT(U, j, k) += s * rv1[k];
        }
        for (k = l; k < n; k++) AT(U, i, k) *= scale;
      }
    }
    {
      double cand = fabs(w[i]) + fabs(rv1[i]);
      anorm = anorm > cand ? anorm : cand;
    }
  }
  for (i = n - 1; i >= 0; i--) { /* right-hand transformations */
    if (i < n - 1) {
      if (g) {
        for (j = l; j < n; j++) AT(*V, j, i) = (AT(U, i, j) / AT(U, i, l)) / g;
        for (j = l; j < n; j++) {
          for (s = 0.0, k = l; k < n; k++) s += AT(U, i, k) * AT(*V, k, j);
          for (k = l; k < n; k++) AT(*V, k, j) += s * AT(*V, k, i);
        }
      }
      for (j = l; j < n; j++) AT(*V, i, j) = AT(*V, j, i) = 0.0;
    }
    AT(*V, i, i) = 1.0;
    g = rv1[i];
    l = i;
  }
  for (i = (m < n ? m : n) - 1; i >= 0; i--) { /* left-hand transformations */
    l = i + 1;
    g = w[i];
    for (j = l; j < n; j++) AT(U, i, j) = 0.0;
    if (g) {
      g = 1.0 / g;
      for (j = l; j < n; j++) {
        for (s = 0.0, k = l; k < m; k++) s += AT(U, k, i) * AT(U, k, j);
        f = (s / AT(U, i, i)) * g;
        for (k = i; k < m; k++) AT(U, k, j) += f * AT(U, k, i);
      }
      for (j = i; j < m; j++) AT(U, j, i) *= g;
    } else {
      for (j = i; j < m; j++) AT(U, j, i) = 0.0;
    }
    AT(U, i, i) += 1.0;
  }
  for (k = n - 1; k >= 0; k--) { /* diagonalisation */
    for (its = 0; its < 30; its++) {
      flag = 1;
      for (l = k; l >= 0; l--) {
        nm = l - 1;
        if ((double)(fabs(rv1[l]) + anorm) == anorm) {
          flag = 0;
          break;
        }
        if ((double)(fabs(w[nm]) + anorm) == anorm) break;
      }
      if (flag) {
        c = 0.0;
        s = 1.0;
        for (i = l; i <= k; i++) {
          f = s * rv1[i];
          rv1[i] = c * rv1[i];
          if ((double)(fabs(f) + anorm) == anorm) break;
          g = w[i];
          h = nr_pythag(f, g);
          w[i] = h;
          h = 1.0 / h;
          c = g * h;
          s = -f * h;
          for (j = 0; j < m; j++) {
            y = AT(U, j, nm);
            z = AT(U, j, i);
            AT(U, j, nm) = y * c + z * s;
            AT(U, j, i) = z * c - y * s;
          }
        }
      }
      z = w[k];
      if (l == k) {
        if (z < 0.0) {
          w[k] = -z;
          for (j = 0; j < n; j++) AT(*V, j, k) = -AT(*V, j, k);
        }
        break;
      }
      x = w[l];
      nm = k - 1;
      y = w[nm];
      g = rv1[nm];
      h = rv1[k];
      f = ((y - z) * (y + z) + (g - h) * (g + h)) / (2.0 * h * y);
      g = nr_pythag(f, 1.0);
      f = ((x - z) * (x + z) + h * ((y / (f + nr_sign(g, f))) - h)) / x;
      c = s = 1.0;
      for (j = l; j <= nm; j++) {
        i = j + 1;
        g = rv1[i];
        y = w[i];
        h = s * g;
        g = c * g;
        z = nr_pythag(f, h);
        rv1[j] = z;
        c = f / z;
        s = h / z;
        f = x * c + g * s;
        g = g * c - x * s;
        h = y * s;
        y *= c;
        for (jj = 0; jj < n; jj++) {
          x = AT(*V, jj, j);
          z = AT(*V, jj, i);
          AT(*V, jj, j) = x * c + z * s;
          AT(*V, jj, i) = z * c - x * s;
        }
        z = nr_pythag(f, h);
        w[j] = z;
        if (z) {
          z = 1.0 / z;
          c = f * z;
          s = h * z;
        }
        f = c * g + s * y;
        x = c * y - s * g;
        for (jj = 0; jj < m; jj++) {
          y = AT(U, jj, j);
          z = AT(U, jj, i);
          AT(U, jj, j) = y * c + z * s;
          AT(U, jj, i) = z * c - y * s;
        }
      }
      rv1[l] = 0.0;
      rv1[k] = f;
      w[k] = x;
    }
  }
  { /* order by decreasing singular value (shell sort, increments 13, 4, 1, ...) */
    int inc = 1;
    double sw;
    double *su = (double *)malloc(sizeof(double) * (size_t)m), *sv = (double *)malloc(sizeof(double) * (size_t)n);
    do {
      inc *= 3;
      inc++;
    } while (inc <= n);
    do {
      inc /= 3;
      for (i = inc; i < n; i++) {
        sw = w[i];
        for (k = 0; k < m; k++) su[k] = AT(U, k, i);
        for (k = 0; k < n; k++) sv[k] = AT(*V, k, i);
        j = i;
        while (w[j - inc] < sw) {
          w[j] = w[j - inc];
          for (k = 0; k < m; k++) AT(U, k, j) = AT(U, k, j - inc);
          for (k = 0; k < n; k++) AT(*V, k, j) = AT(*V, k, j - inc);
          j -= inc;
          if (j < inc) break;
        }
        w[j] = sw;
        for (k = 0; k < m; k++) AT(U, k, j) = su[k];
        for (k = 0; k < n; k++) AT(*V, k, j) = sv[k];
      }
    } while (inc > 1);
    free(su);
    free(sv);
  }
  for (k = 0; k < n; k++) { /* sign convention */
    int neg = 0;
    for (i = 0; i < m; i++) neg += AT(U, i, k) < 0.0;
    for (j = 0; j < n; j++) neg += AT(*V, j, k) < 0.0;
    if (neg > (m + n) / 2) {
      for (i = 0; i < m; i++) AT(U, i, k) = -AT(U, i, k);
      for (j = 0; j < n; j++) AT(*V, j, k) = -AT(*V, j, k);
    }
  }
  {
    const int r = m < n ? m : n;
    for (i = 0; i < r; i++) W[i] = w[i];
    for (i = 0; i < m; i++)
      for (j = 0; j < r; j++) AT(*U2, i, j) = AT(U, i, j);
  }
  free(w);
  free(rv1);
  dm_free(&U);
}

/* Matrix::det via Crout LU with implicit pivoting (viso/matrix.cpp:407-422, :521-580) */
static double dm_det(const dm *Ain) {
  const int n = Ain->n;
  dm A = dm_copy(Ain);
  double *vv = (double *)malloc(sizeof(double) * (size_t)n);
  double d = 1.0;
  int ok = 1, imax = 0;
  for (int i = 0; i < n && ok; i++) {
    double big = 0.0;
    for (int j = 0; j < n; j++) {
      double t = fabs(AT(A, i, j));
      if (t > big) big = t;
    }
    if (big == 0.0) ok = 0;
    else vv[i] = 1.0 / big;
  }
  for (int j = 0; j < n && ok; j++) {
    for (int i = 0; i < j; i++) {
      double sum = AT(A, i, j);
      for (int k = 0; k < i; k++) sum -= AT(A, i, k) * AT(A, k, j);
      AT(A, i, j) = sum;
    }
    double big = 0.0;
    for (int i = j; i < n; i++) {
      double sum = AT(A, i, j);
      for (int k = 0; k < j; k++) sum -= AT(A, i, k) * AT(A, k, j);
      AT(A, i, j) = sum;
      double dum = vv[i] * fabs(sum);
      if (dum >= big) {
        big = dum;
        imax = i;
      }
    }
    if (j != imax) {
      for (int k = 0; k < n; k++) {
        double t = AT(A, imax, k);
        AT(A, imax, k) = AT(A, j, k);
        AT(A, j, k) = t;
      }
      d = -d;
      vv[imax] = vv[j];
    }
    if (j != n - 1) {
      double dum = 1.0 / AT(A, j, j);
      for (int i = j + 1; i < n; i++) AT(A, i, j) *= dum;
    }
  }
  /* Matrix::det ignores lu()'s failure flag and multiplies the diagonal it finds */
  for (int i = 0; i < n; i++) d *= AT(A, i, i);
  free(vv);
  dm_free(&A);
  return d;
}

void vo_matrix_svd(const double *A, int32_t m, int32_t n, double *U, double *W, double *V) {
  dm a = dm_from(m, n, A), u, v;
  dm_svd(&a, &u, W, &v);
  memcpy(U, u.a, sizeof(double) * (size_t)m * (size_t)m);
  memcpy(V, v.a, sizeof(double) * (size_t)n * (size_t)n);
  dm_free(&a);
  dm_free(&u);
  dm_free(&v);
}
double vo_matrix_det(const double *A, int32_t n) {
  dm a = dm_from(n, n, A);
  double d = dm_det(&a);
  dm_free(&a);
  return d;
}

/* ------------------------------------------------------------------------------------------ */
/* the estimate                                                                                */
/* ------------------------------------------------------------------------------------------ */
void vo_mono_default_params(vo_mono_params *p) { /* viso/viso_mono.h:33-46, viso/viso.h:33-42 */
  p->f = 1;
  p->cu = 0;
  p->cv = 0;
  p->height = 1.0;
  p->pitch = 0.0;
  p->ransac_iters = 2000;
  p->inlier_threshold = 0.00001;
  p->motion_threshold = 100.0;
}

/* enforce rank 2: F = U diag(w0,w1,0) V^T (viso/viso_mono.cpp:289-293, :125-129) */
static dm rank2(const dm *F) {
  dm U, V;
  double W[3];
  dm_svd(F, &U, W, &V);
  W[2] = 0;
  dm D = dm_diag(W, 3), UD = dm_mul(&U, &D), Vt = dm_T(&V), R = dm_mul(&UD, &Vt);
  dm_free(&U);
  dm_free(&V);
  dm_free(&D);
  dm_free(&UD);
  dm_free(&Vt);
  return R;
}

/* fundamentalMatrix, viso/viso_mono.cpp:264-294: products of the match coordinates are FLOAT products */
void vo_mono_fundamental(const vo_match *m, const int32_t *active, int32_t na, double *F9) {
  dm A = dm_new(na, 9);
  for (int i = 0; i < na; i++) {
    const vo_match *q = &m[active[i]];
    AT(A, i, 0) = q->u1c * q->u1p;
    AT(A, i, 1) = q->u1c * q->v1p;
    AT(A, i, 2) = q->u1c;
    AT(A, i, 3) = q->v1c * q->u1p;
    AT(A, i, 4) = q->v1c * q->v1p;
    AT(A, i, 5) = q->v1c;
    AT(A, i, 6) = q->u1p;
    AT(A, i, 7) = q->v1p;
    AT(A, i, 8) = 1;
  }
  dm U, V;
  double *W = (double *)malloc(sizeof(double) * 9);
  dm_svd(&A, &U, W, &V);
  dm F = dm_new(3, 3);
  for (int k = 0; k < 9; k++) F.a[k] = AT(V, k, 8); /* column of the smallest singular value */
  dm R = rank2(&F);
  memcpy(F9, R.a, sizeof(double) * 9);
  free(W);
  dm_free(&A);
  dm_free(&U);
  dm_free(&V);
  dm_free(&F);
  dm_free(&R);
}

/* getInlier (Sampson distance), viso/viso_mono.cpp:296-344 */
static int32_t mono_inliers(const vo_match *m, int32_t n, const double *F, double thr, int32_t *out) {
  int32_t k = 0;
  for (int32_t i = 0; i < n; i++) {
    double u1 = m[i].u1p, v1 = m[i].v1p, u2 = m[i].u1c, v2 = m[i].v1c;
    double Fx1u = F[0] * u1 + F[1] * v1 + F[2];
    double Fx1v = F[3] * u1 + F[4] * v1 + F[5];
    double Fx1w = F[6] * u1 + F[7] * v1 + F[8];
    double Ftx2u = F[0] * u2 + F[3] * v2 + F[6];
    double Ftx2v = F[1] * u2 + F[4] * v2 + F[7];
    double x2tFx1 = u2 * Fx1u + v2 * Fx1v + Fx1w;
    double d = x2tFx1 * x2tFx1 / (Fx1u * Fx1u + Fx1v * Fx1v + Ftx2u * Ftx2u + Ftx2v * Ftx2v);
    if (fabs(d) < thr) out[k++] = i;
  }
  return k;
}

/* normalizeFeaturePoints, viso/viso_mono.cpp:215-262: the match fields are floats, so the centring
 * and scaling round to float, and the radius sqrt(u*u+v*v) is float arithmetic throughout */
static int mono_normalize(vo_match *m, int32_t n, double *Tp9, double *Tc9) {
  double cpu = 0, cpv = 0, ccu = 0, ccv = 0;
  for (int32_t i = 0; i < n; i++) {
    cpu += m[i].u1p;
    cpv += m[i].v1p;
    ccu += m[i].u1c;
    ccv += m[i].v1c;
  }
  cpu /= (double)n;
  cpv /= (double)n;
  ccu /= (double)n;
  ccv /= (double)n;
  for (int32_t i = 0; i < n; i++) {
    m[i].u1p = (float)(m[i].u1p - cpu);
    m[i].v1p = (float)(m[i].v1p - cpv);
    m[i].u1c = (float)(m[i].u1c - ccu);
    m[i].v1c = (float)(m[i].v1c - ccv);
  }
  double sp = 0, sc = 0;
  for (int32_t i = 0; i < n; i++) {
    sp += sqrtf(m[i].u1p * m[i].u1p + m[i].v1p * m[i].v1p);
    sc += sqrtf(m[i].u1c * m[i].u1c + m[i].v1c * m[i].v1c);
  }
  if (fabs(sp) < 1e-10 || fabs(sc) < 1e-10) return 0;
  sp = sqrt(2.0) * (double)n / sp;
  sc = sqrt(2.0) * (double)n / sc;
  for (int32_t i = 0; i < n; i++) {
    m[i].u1p = (float)(m[i].u1p * sp);
    m[i].v1p = (float)(m[i].v1p * sp);
    m[i].u1c = (float)(m[i].u1c * sc);
    m[i].v1c = (float)(m[i].v1c * sc);
  }
  const double tp[9] = {sp, 0, -sp * cpu, 0, sp, -sp * cpv, 0, 0, 1};
  const double tc[9] = {sc, 0, -sc * ccu, 0, sc, -sc * ccv, 0, 0, 1};
  memcpy(Tp9, tp, sizeof(tp));
  memcpy(Tc9, tc, sizeof(tc));
  return 1;
}

/* triangulateChieral, viso/viso_mono.cpp:394-431: X is 4 x n; returns the chirality count */
static int32_t mono_triangulate(const vo_match *m, int32_t n, const dm *K, const dm *R, const dm *t, dm *X) {
  *X = dm_new(4, n);
  dm P1 = dm_new(3, 4), Rt = dm_new(3, 4);
  for (int i = 0; i < 3; i++)
    for (int j = 0; j < 3; j++) {
      AT(P1, i, j) = AT(*K, i, j);
      AT(Rt, i, j) = AT(*R, i, j);
    }
  for (int i = 0; i < 3; i++) AT(Rt, i, 3) = AT(*t, i, 0);
  dm P2 = dm_mul(K, &Rt);
  dm J = dm_new(4, 4);
  for (int32_t i = 0; i < n; i++) {
    for (int j = 0; j < 4; j++) {
      AT(J, 0, j) = AT(P1, 2, j) * m[i].u1p - AT(P1, 0, j);
      AT(J, 1, j) = AT(P1, 2, j) * m[i].v1p - AT(P1, 1, j);
      AT(J, 2, j) = AT(P2, 2, j) * m[i].u1c - AT(P2, 0, j);
      AT(J, 3, j) = AT(P2, 2, j) * m[i].v1c - AT(P2, 1, j);
    }
    dm U, V;
    double W[4];
    dm_svd(&J, &U, W, &V);
    for (int r = 0; r < 4; r++) AT(*X, r, i) = AT(V, r, 3);
    dm_free(&U);
    dm_free(&V);
  }
  dm AX1 = dm_mul(&P1, X), BX1 = dm_mul(&P2, X);
  int32_t num = 0;
  for (int32_t i = 0; i < n; i++)
    if (AT(AX1, 2, i) * AT(*X, 3, i) > 0 && AT(BX1, 2, i) * AT(*X, 3, i) > 0) num++;
  dm_free(&P1);
  dm_free(&Rt);
  dm_free(&P2);
  dm_free(&J);
  dm_free(&AX1);
  dm_free(&BX1);
  return num;
}

static int cmp_double(const void *a, const void *b) {
  double x = *(const double *)a, y = *(const double *)b;
  return (x > y) - (x < y);
}

/* estimateMotion, viso/viso_mono.cpp:103-187.
 * returns 1 (tr6 filled) / 0; *n_inliers < 0 on return means "inlier list untouched" (the early
 * exits before ransacEstimateF clears it). */
int32_t vo_estimate_motion_mono(const vo_match *matched, int32_t n, const vo_mono_params *p, double *tr6,
                                int32_t *inliers, int32_t *n_inliers) {
  *n_inliers = -1;
  if (n < 10) return 0;
  const double Kd[9] = {p->f, 0, p->cu, 0, p->f, p->cv, 0, 0, 1};
  dm K = dm_from(3, 3, Kd);
  vo_match *nm = (vo_match *)malloc(sizeof(vo_match) * (size_t)n);
  memcpy(nm, matched, sizeof(vo_match) * (size_t)n);
  double Tp9[9], Tc9[9];
  if (!mono_normalize(nm, n, Tp9, Tc9)) {
    free(nm);
    dm_free(&K);
    return 0;
  }
  /* ransacEstimateF, :41-73 */
  int32_t best = 0;
  int32_t *cur = (int32_t *)malloc(sizeof(int32_t) * (size_t)n), *deck = (int32_t *)malloc(sizeof(int32_t) * (size_t)n);
  for (int32_t k = 0; k < p->ransac_iters; k++) {
    int32_t act[8];
    for (int32_t i = 0; i < n; i++) deck[i] = i;
    for (int32_t i = 0; i < 8; i++) {
      uint32_t j = vo_ego_draw_between((uint32_t)i, (uint32_t)(n - 1));
      int32_t t = deck[i];
      deck[i] = deck[j];
      deck[j] = t;
    }
    memcpy(act, deck, sizeof(act));
    double F9[9];
    vo_mono_fundamental(nm, act, 8, F9);
    int32_t c = mono_inliers(nm, n, F9, p->inlier_threshold, cur);
    if (c > best) {
      best = c;
      memcpy(inliers, cur, sizeof(int32_t) * (size_t)c);
    }
  }
  *n_inliers = best;
  free(cur);
  free(deck);
  int ok = 0;
  dm X = {0, 0, 0}, R = {0, 0, 0}, t = {0, 0, 0};
  if (best >= 10) {
    double F9[9];
    vo_mono_fundamental(nm, inliers, best, F9);
    /* F = Tc^T F Tp, E = K^T F K, rank 2 again (:121-129) */
    dm F = dm_from(3, 3, F9), Tp = dm_from(3, 3, Tp9), Tc = dm_from(3, 3, Tc9);
    dm TcT = dm_T(&Tc), a = dm_mul(&TcT, &F), Fd = dm_mul(&a, &Tp);
    dm KT = dm_T(&K), b = dm_mul(&KT, &Fd), E0 = dm_mul(&b, &K);
    dm E = rank2(&E0);
    /* EtoRt, :346-392 */
    const double Wd[9] = {0, -1, 0, +1, 0, 0, 0, 0, 1}, Zd[9] = {0, +1, 0, -1, 0, 0, 0, 0, 0};
    dm Wm = dm_from(3, 3, Wd), Zm = dm_from(3, 3, Zd);
    dm U, V;
    double S[3];
    dm_svd(&E, &U, S, &V);
    dm UT = dm_T(&U), VT = dm_T(&V), WT = dm_T(&Wm);
    dm UZ = dm_mul(&U, &Zm), T = dm_mul(&UZ, &UT);
    dm UW = dm_mul(&U, &Wm), Ra = dm_mul(&UW, &VT);
    dm UWt = dm_mul(&U, &WT), Rb = dm_mul(&UWt, &VT);
    dm tt = dm_new(3, 1);
    AT(tt, 0, 0) = AT(T, 2, 1);
    AT(tt, 1, 0) = AT(T, 0, 2);
    AT(tt, 2, 0) = AT(T, 1, 0);
    if (dm_det(&Ra) < 0)
      for (int i = 0; i < 9; i++) Ra.a[i] = -Ra.a[i];
    if (dm_det(&Rb) < 0)
      for (int i = 0; i < 9; i++) Rb.a[i] = -Rb.a[i];
    dm tneg = dm_new(3, 1);
    for (int i = 0; i < 3; i++) tneg.a[i] = -tt.a[i];
    const dm *Rs[4] = {&Ra, &Ra, &Rb, &Rb}, *ts[4] = {&tt, &tneg, &tt, &tneg};
    int32_t max_in = 0;
    for (int c = 0; c < 4; c++) {
      dm Xc;
      int32_t num = mono_triangulate(matched, n, &K, Rs[c], ts[c], &Xc);
      if (num > max_in) {
        max_in = num;
        dm_free(&X);
        dm_free(&R);
        dm_free(&t);
        X = Xc;
        R = dm_copy(Rs[c]);
        t = dm_copy(ts[c]);
      } else {
        dm_free(&Xc);
      }
    }
    if (max_in > 0) {
      /* X / X(3,:), points in front, median of the L1 norms (:137-161) */
      int32_t np = 0;
      double *y = (double *)malloc(sizeof(double) * (size_t)n), *zc = (double *)malloc(sizeof(double) * (size_t)n);
      double *dist = (double *)malloc(sizeof(double) * (size_t)n);
      for (int32_t i = 0; i < n; i++) {
        double w4 = AT(X, 3, i);
        double x0 = w4 != 0 ? AT(X, 0, i) / w4 : 0, x1 = w4 != 0 ? AT(X, 1, i) / w4 : 0;
        double x2 = w4 != 0 ? AT(X, 2, i) / w4 : 0;
        if (x2 > 0) {
          y[np] = x1;
          zc[np] = x2;
          dist[np] = fabs(x0) + fabs(x1) + fabs(x2);
          np++;
        }
      }
      if (np >= 10) {
        double *sorted = (double *)malloc(sizeof(double) * (size_t)np);
        memcpy(sorted, dist, sizeof(double) * (size_t)np);
        qsort(sorted, (size_t)np, sizeof(double), cmp_double);
        double median = sorted[np / 2];
        free(sorted);
        if (!(median > p->motion_threshold)) {
          double sigma = median / 50.0;
          double weight = 1.0 / (2.0 * sigma * sigma);
          double threshold = median / p->motion_threshold;
          /* findBestPlane, :75-101 */
          double n0 = cos(-p->pitch), n1 = sin(-p->pitch);
          double *d = dist; /* reuse */
          for (int32_t i = 0; i < np; i++) {
            double s = 0;
            s += n0 * y[i];
            s += n1 * zc[i];
            d[i] = s;
          }
          double best_sum = 0;
          int32_t best_idx = 0;
          for (int32_t i = 0; i < np; i++) {
            if (d[i] > threshold) {
              double sum = 0;
              for (int32_t j = 0; j < np; j++) {
                double dd = d[j] - d[i];
                sum += exp(-dd * dd * weight);
              }
              if (sum > best_sum) {
                best_sum = sum;
                best_idx = i;
              }
            }
          }
          double best_d = d[best_idx];
          double ry = asin(AT(R, 0, 2));
          double rx = asin(-AT(R, 1, 2) / cos(ry));
          double rz = asin(-AT(R, 0, 1) / cos(ry));
          tr6[0] = rx;
          tr6[1] = ry;
          tr6[2] = rz;
          for (int i = 0; i < 3; i++) tr6[3 + i] = AT(t, i, 0) * p->height / best_d;
          ok = 1;
        }
      }
      free(y);
      free(zc);
      free(dist);
    }
    dm *all[] = {&F, &Tp, &Tc, &TcT, &a, &Fd, &KT, &b, &E0, &E, &Wm, &Zm, &U, &V, &UT, &VT, &WT, &UZ, &T, &UW, &Ra, &UWt,
                 &Rb, &tt, &tneg};
    for (size_t i = 0; i < sizeof(all) / sizeof(all[0]); i++) dm_free(all[i]);
  }
  dm_free(&X);
  dm_free(&R);
  dm_free(&t);
  dm_free(&K);
  free(nm);
  return ok;
}

/* ------------------------------------------------------------------------------------------ */
/* VisualOdometryMono as an object (viso/viso_mono.cpp:27-39)                                  */
/* ------------------------------------------------------------------------------------------ */
struct vo_mono {
  vo_matcher *matcher;
  vo_mono_params mp;
  int32_t bucket_max;
  double bucket_w, bucket_h;
  double T[16];
  int32_t tr_valid;
  vo_match *matched;
  int32_t n_matched, cap_matched;
  int32_t *inliers;
  int32_t n_inliers, cap_inliers;
};

vo_mono *vo_mono_create(const vo_params *mp, int32_t bucket_max, double bucket_w, double bucket_h,
                        const vo_mono_params *ep) {
  vo_mono *v = (vo_mono *)calloc(1, sizeof(vo_mono));
  v->matcher = vo_create(mp);
  v->mp = *ep;
  v->bucket_max = bucket_max;
  v->bucket_w = bucket_w;
  v->bucket_h = bucket_h;
  for (int32_t i = 0; i < 16; i++) v->T[i] = (i % 5 == 0) ? 1.0 : 0.0;
  srand(0); /* viso/viso.cpp:35 */
  return v;
}
void vo_mono_destroy(vo_mono *v) {
  if (!v) return;
  vo_destroy(v->matcher);
  free(v->matched);
  free(v->inliers);
  free(v);
}
static void mono_set_matched(vo_mono *v, int32_t n) {
  if (n > v->cap_matched) {
    v->cap_matched = n;
    v->matched = (vo_match *)realloc(v->matched, sizeof(vo_match) * (size_t)n);
  }
  v->n_matched = n;
}
static int32_t mono_update_motion(vo_mono *v) {
  if (v->n_matched > v->cap_inliers) {
    v->cap_inliers = v->n_matched;
    v->inliers = (int32_t *)realloc(v->inliers, sizeof(int32_t) * (size_t)v->cap_inliers);
  }
  double tr[6];
  int32_t ni = -1;
  int32_t rc = vo_estimate_motion_mono(v->matched, v->n_matched, &v->mp, tr, v->inliers, &ni);
  if (ni >= 0) v->n_inliers = ni;
  if (rc != 1) return 0;
  vo_tr_vector_to_matrix(tr, v->T);
  v->tr_valid = 1;
  return 1;
}
int32_t vo_mono_process(vo_mono *v, const uint8_t *I, int32_t w, int32_t h, int32_t bpl, int32_t replace) {
  vo_push_back(v->matcher, I, 0, w, h, bpl, replace);
  vo_match_features(v->matcher, 0, 0);
  vo_bucket_features(v->matcher, v->bucket_max, (float)v->bucket_w, (float)v->bucket_h);
  mono_set_matched(v, vo_num_matches(v->matcher));
  if (v->n_matched) vo_get_matches(v->matcher, v->matched);
  return mono_update_motion(v);
}
int32_t vo_mono_process_matches(vo_mono *v, const vo_match *m, int32_t n) {
  mono_set_matched(v, n);
  if (n) memcpy(v->matched, m, sizeof(vo_match) * (size_t)n);
  return mono_update_motion(v);
}
void vo_mono_get_motion(const vo_mono *v, double *T16) { memcpy(T16, v->T, sizeof(v->T)); }
int32_t vo_mono_num_matches(const vo_mono *v) { return v->n_matched; }
void vo_mono_get_matches(const vo_mono *v, vo_match *out) {
  if (v->n_matched) memcpy(out, v->matched, sizeof(vo_match) * (size_t)v->n_matched);
}
int32_t vo_mono_num_inliers(const vo_mono *v) { return v->n_inliers; }
void vo_mono_get_inliers(const vo_mono *v, int32_t *out) {
  if (v->n_inliers) memcpy(out, v->inliers, sizeof(int32_t) * (size_t)v->n_inliers);
}
