// Small dense linear algebra with the exact arithmetic of the reference's `Matrix` class
// (viso/matrix.cpp): singular value decomposition (:586-850, Numerical Recipes' svdcmp followed by the
// reference's ordering and sign convention), determinant by Crout LU with implicit pivoting
// (:407-422, :521-580), matrix product with the reference's summation order (:270-284).
// The mono egomotion (viso/viso_mono.cpp) is a chain of these; its results feed the caller's pose,
// so every operation here keeps the reference's order of floating-point operations -- build with
// -ffp-contract=off.  Header-only, no allocation, usable from host and device code alike.
//
// Provenance: `svd` below follows viso/matrix.cpp:586-850 step for step, which itself is the `svdcmp` routine of
// Numerical Recipes in C (Householder bidiagonalisation + implicit-shift QR, same variable roles: anorm, rv1, flag,
// nm, its) plus the reference's descending sort and sign convention.  The closeness is deliberate and unavoidable:
// the results must be bit-identical in double precision, which pins the order of every operation of a published
// algorithm; only the loop interchange for n <= 16 and the device form (vsm_svd_coop.h) are this repository's own.
#pragma once

#include <math.h>
#include <stdint.h>

#if defined(__HIPCC__)
#define VSM_HD __host__ __device__
#else
#define VSM_HD
#endif

namespace vsm_la {

VSM_HD inline double hypot_nr(double a, double b) {  // Matrix::pythag, viso/matrix.cpp:852-860
  const double absa = fabs(a), absb = fabs(b);
  if (absa > absb) {
    const double q = absb / absa;
    return absa * sqrt(1.0 + (q == 0.0 ? 0.0 : q * q));
  }
  if (absb == 0.0) return 0.0;
  const double q = absa / absb;
  return absb * sqrt(1.0 + (q == 0.0 ? 0.0 : q * q));
}

VSM_HD inline double with_sign(double a, double b) { return b >= 0.0 ? fabs(a) : -fabs(a); }

// A = U diag(w) V^T for a row-major m x n matrix held in `u` (row stride ldu), overwritten by the
// m x n factor U.  w[n], v[n*n] (row-major), rv1[n] scratch, col[m > n ? m : n] scratch.
// On return the singular values are in decreasing order and every (u column, v column) pair has
// at most half of its entries negative, exactly as Matrix::svd leaves them.
// ES = distance between consecutive elements of every array: 1 for ordinary arrays; a kernel that
// keeps one matrix per lane in LDS uses its block size, so that lanes never share a bank.
template <int ES = 1>
VSM_HD inline void svd_nr(double *u, int m, int n, int ldu, double *w_, double *v, double *rv1_, double *col_) {
#define U_(i, j) u[((i) * ldu + (j)) * ES]
#define V_(i, j) v[((i) * n + (j)) * ES]
#define w(i) w_[(i) * ES]
#define rv1(i) rv1_[(i) * ES]
#define col(i) col_[(i) * ES]
  int flag, i, its, j, jj, k, l = 0, nm = 0;
  double anorm = 0.0, c, f, g = 0.0, h, s, scale = 0.0, x, y, z;
  for (i = 0; i < n * n; i++) v[i * ES] = 0.0;
  // Householder reduction to bidiagonal form
  for (i = 0; i < n; i++) {
    l = i + 1;
    rv1(i) = scale * g;
    g = s = scale = 0.0;
    if (i < m) {
      for (k = i; k < m; k++) scale += fabs(U_(k, i));
      if (scale) {
        for (k = i; k < m; k++) {
          U_(k, i) /= scale;
          s += U_(k, i) * U_(k, i);
        }
        f = U_(i, i);
        g = -with_sign(sqrt(s), f);
        h = f * g - s;
        U_(i, i) = f - g;
        if (n <= 16) {
          // the columns j are independent: walk the rows once for all of them (rows are contiguous),
          // every column's sum still adds its terms in row order
          double sj[16];
          for (j = l; j < n; j++) sj[j] = 0.0;
          for (k = i; k < m; k++) {
            const double ui = U_(k, i);
            for (j = l; j < n; j++) sj[j] += ui * U_(k, j);
          }
          for (j = l; j < n; j++) sj[j] = sj[j] / h;
          for (k = i; k < m; k++) {
            const double ui = U_(k, i);
            for (j = l; j < n; j++) U_(k, j) += sj[j] * ui;
          }
        } else {
          for (j = l; j < n; j++) {
            for (s = 0.0, k = i; k < m; k++) s += U_(k, i) * U_(k, j);
            f = s / h;
            for (k = i; k < m; k++) U_(k, j) += f * U_(k, i);
          }
        }
        for (k = i; k < m; k++) U_(k, i) *= scale;
      }
    }
    w(i) = scale * g;
    g = s = scale = 0.0;
    if (i < m && i != n - 1) {
      for (k = l; k < n; k++) scale += fabs(U_(i, k));
      if (scale) {
        for (k = l; k < n; k++) {
          U_(i, k) /= scale;
          s += U_(i, k) * U_(i, k);
        }
        f = U_(i, l);
        g = -with_sign(sqrt(s), f);
        h = f * g - s;
        U_(i, l) = f - g;
        for (k = l; k < n; k++) rv1(k) = U_(i, k) / h;
        for (j = l; j < m; j++) {
          for (s = 0.0, k = l; k < n; k++) s += U_(j, k) * U_(i, k);
          for (k = l; k < n; k++) U_(j, k) += s * rv1(k);
        }
        for (k = l; k < n; k++) U_(i, k) *= scale;
      }
    }
    const double t = fabs(w(i)) + fabs(rv1(i));
    anorm = anorm > t ? anorm : t;
  }
  // accumulate the right-hand transformations
  for (i = n - 1; i >= 0; i--) {
    if (i < n - 1) {
      if (g) {
        for (j = l; j < n; j++) V_(j, i) = (U_(i, j) / U_(i, l)) / g;
        for (j = l; j < n; j++) {
          for (s = 0.0, k = l; k < n; k++) s += U_(i, k) * V_(k, j);
          for (k = l; k < n; k++) V_(k, j) += s * V_(k, i);
        }
      }
      for (j = l; j < n; j++) V_(i, j) = V_(j, i) = 0.0;
    }
    V_(i, i) = 1.0;
    g = rv1(i);
    l = i;
  }
  // accumulate the left-hand transformations
  for (i = (m < n ? m : n) - 1; i >= 0; i--) {
    l = i + 1;
    g = w(i);
    for (j = l; j < n; j++) U_(i, j) = 0.0;
    if (g) {
      g = 1.0 / g;
      if (n <= 16) {
        double sj[16];
        for (j = l; j < n; j++) sj[j] = 0.0;
        for (k = l; k < m; k++) {
          const double ui = U_(k, i);
          for (j = l; j < n; j++) sj[j] += ui * U_(k, j);
        }
        const double uii = U_(i, i);
        for (j = l; j < n; j++) sj[j] = (sj[j] / uii) * g;
        for (k = i; k < m; k++) {
          const double ui = U_(k, i);
          for (j = l; j < n; j++) U_(k, j) += sj[j] * ui;
        }
      } else {
        for (j = l; j < n; j++) {
          for (s = 0.0, k = l; k < m; k++) s += U_(k, i) * U_(k, j);
          f = (s / U_(i, i)) * g;
          for (k = i; k < m; k++) U_(k, j) += f * U_(k, i);
        }
      }
      for (j = i; j < m; j++) U_(j, i) *= g;
    } else {
      for (j = i; j < m; j++) U_(j, i) = 0.0;
    }
    U_(i, i) += 1.0;
  }
  // diagonalise the bidiagonal form: implicit shifted QR, at most 30 sweeps per singular value
  for (k = n - 1; k >= 0; k--) {
    for (its = 0; its < 30; its++) {
      flag = 1;
      for (l = k; l >= 0; l--) {
        nm = l - 1;
        if ((double)(fabs(rv1(l)) + anorm) == anorm) {
          flag = 0;
          break;
        }
        if ((double)(fabs(w(nm)) + anorm) == anorm) break;
      }
      if (flag) {
        c = 0.0;
        s = 1.0;
        for (i = l; i <= k; i++) {
          f = s * rv1(i);
          rv1(i) = c * rv1(i);
          if ((double)(fabs(f) + anorm) == anorm) break;
          g = w(i);
          h = hypot_nr(f, g);
          w(i) = h;
          h = 1.0 / h;
          c = g * h;
          s = -f * h;
          for (j = 0; j < m; j++) {
            y = U_(j, nm);
            z = U_(j, i);
            U_(j, nm) = y * c + z * s;
            U_(j, i) = z * c - y * s;
          }
        }
      }
      z = w(k);
      if (l == k) {  // converged; make the singular value non-negative
        if (z < 0.0) {
          w(k) = -z;
          for (j = 0; j < n; j++) V_(j, k) = -V_(j, k);
        }
        break;
      }
      x = w(l);
      nm = k - 1;
      y = w(nm);
      g = rv1(nm);
      h = rv1(k);
      f = ((y - z) * (y + z) + (g - h) * (g + h)) / (2.0 * h * y);
      g = hypot_nr(f, 1.0);
      f = ((x - z) * (x + z) + h * ((y / (f + with_sign(g, f))) - h)) / x;
      c = s = 1.0;
      for (j = l; j <= nm; j++) {
        i = j + 1;
        g = rv1(i);
        y = w(i);
        h = s * g;
        g = c * g;
        z = hypot_nr(f, h);
        rv1(j) = z;
        c = f / z;
        s = h / z;
        f = x * c + g * s;
        g = g * c - x * s;
        h = y * s;
        y *= c;
        for (jj = 0; jj < n; jj++) {
          x = V_(jj, j);
          z = V_(jj, i);
          V_(jj, j) = x * c + z * s;
          V_(jj, i) = z * c - x * s;
        }
        z = hypot_nr(f, h);
        w(j) = z;
        if (z) {
          z = 1.0 / z;
          c = f * z;
          s = h * z;
        }
        f = c * g + s * y;
        x = c * y - s * g;
        for (jj = 0; jj < m; jj++) {
          y = U_(jj, j);
          z = U_(jj, i);
          U_(jj, j) = y * c + z * s;
          U_(jj, i) = z * c - y * s;
        }
      }
      rv1(l) = 0.0;
      rv1(k) = f;
      w(k) = x;
    }
  }
  // decreasing order (shell sort with the 1, 4, 13, ... increments), columns of u and v follow
  int inc = 1;
  do {
    inc = inc * 3 + 1;
  } while (inc <= n);
  do {
    inc /= 3;
    for (i = inc; i < n; i++) {
      const double sw = w(i);
      for (k = 0; k < m; k++) col(k) = U_(k, i);
      j = i;
      // (the v column is moved through rv1, which is free now)
      for (k = 0; k < n; k++) rv1(k) = V_(k, i);
      while (w(j - inc) < sw) {
        w(j) = w(j - inc);
        for (k = 0; k < m; k++) U_(k, j) = U_(k, j - inc);
        for (k = 0; k < n; k++) V_(k, j) = V_(k, j - inc);
        j -= inc;
        if (j < inc) break;
      }
      w(j) = sw;
      for (k = 0; k < m; k++) U_(k, j) = col(k);
      for (k = 0; k < n; k++) V_(k, j) = rv1(k);
    }
  } while (inc > 1);
  // sign convention: flip a (u, v) column pair when more than half of its entries are negative
  for (k = 0; k < n; k++) {
    int neg = 0;
    for (i = 0; i < m; i++) neg += U_(i, k) < 0.0 ? 1 : 0;
    for (j = 0; j < n; j++) neg += V_(j, k) < 0.0 ? 1 : 0;
    if (neg > (m + n) / 2) {
      for (i = 0; i < m; i++) U_(i, k) = -U_(i, k);
      for (j = 0; j < n; j++) V_(j, k) = -V_(j, k);
    }
  }
#undef U_
#undef V_
#undef w
#undef rv1
#undef col
}

// C (m x p) = A (m x n) * B (n x p), row-major; every entry is 0 + a_i0*b_0j + a_i1*b_1j + ...
VSM_HD inline void mul(const double *A, const double *B, double *C, int m, int n, int p) {
  for (int i = 0; i < m; i++)
    for (int j = 0; j < p; j++) {
      double s = 0;
      for (int k = 0; k < n; k++) s += A[i * n + k] * B[k * p + j];
      C[i * p + j] = s;
    }
}
VSM_HD inline void transpose(const double *A, double *T, int m, int n) {
  for (int i = 0; i < m; i++)
    for (int j = 0; j < n; j++) T[j * m + i] = A[i * n + j];
}

// determinant of a 3x3 (Matrix::det: LU with implicit pivoting; a zero row leaves the matrix
// undecomposed and the product of its diagonal is returned, like the reference does)
VSM_HD inline double det3(const double *Ain) {
  double a[9], vv[3];
  for (int i = 0; i < 9; i++) a[i] = Ain[i];
  double d = 1.0;
  bool ok = true;
  for (int i = 0; i < 3 && ok; i++) {
    double big = 0.0;
    for (int j = 0; j < 3; j++) {
      const double t = fabs(a[i * 3 + j]);
      if (t > big) big = t;
    }
    if (big == 0.0) ok = false;
    else vv[i] = 1.0 / big;
  }
  int imax = 0;
  for (int j = 0; j < 3 && ok; j++) {
    for (int i = 0; i < j; i++) {
      double sum = a[i * 3 + j];
      for (int k = 0; k < i; k++) sum -= a[i * 3 + k] * a[k * 3 + j];
      a[i * 3 + j] = sum;
    }
    double big = 0.0;
    for (int i = j; i < 3; i++) {
      double sum = a[i * 3 + j];
      for (int k = 0; k < j; k++) sum -= a[i * 3 + k] * a[k * 3 + j];
      a[i * 3 + j] = sum;
      const double dum = vv[i] * fabs(sum);
      if (dum >= big) {
        big = dum;
        imax = i;
      }
    }
    if (j != imax) {
      for (int k = 0; k < 3; k++) {
        const double t = a[imax * 3 + k];
        a[imax * 3 + k] = a[j * 3 + k];
        a[j * 3 + k] = t;
      }
      d = -d;
      vv[imax] = vv[j];
    }
    if (j != 2) {
      const double dum = 1.0 / a[j * 3 + j];
      for (int i = j + 1; i < 3; i++) a[i * 3 + j] *= dum;
    }
  }
  for (int i = 0; i < 3; i++) d *= a[i * 3 + i];
  return d;
}

// rank-2 projection of a 3x3: U diag(w0, w1, 0) V^T (viso/viso_mono.cpp:289-293, :125-129)
VSM_HD inline void rank2_3x3(const double *F, double *out) {
  double u[9], w[3], v[9], rv1[3], col[3];
  for (int i = 0; i < 9; i++) u[i] = F[i];
  svd_nr(u, 3, 3, 3, w, v, rv1, col);
  w[2] = 0;
  double D[9] = {w[0], 0, 0, 0, w[1], 0, 0, 0, w[2]}, UD[9], Vt[9];
  mul(u, D, UD, 3, 3, 3);
  transpose(v, Vt, 3, 3);
  mul(UD, Vt, out, 3, 3, 3);
}

}  // namespace vsm_la
