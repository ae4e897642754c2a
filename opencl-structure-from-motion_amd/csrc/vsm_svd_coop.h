// Matrix::svd (viso/matrix.cpp:586-850) for ONE small M x N matrix worked on by a group of 16 lanes
// of a wavefront: the scalar recurrences are evaluated redundantly by every lane of the group (same
// operands, same order, hence the same bits), the loops over independent columns / rows are dealt
// one index per lane.  Every individual sum still adds its terms in the reference's order, so the
// result equals vsm_la::svd_nr bit for bit; what changes is the length of the dependent instruction
// chain per lane (about N times shorter).  U (M x N, row-major), V (N x N), W[N], RV[N] live in LDS.
// Device only.  `ln` = lane index inside the group (0..15); lanes >= max(M, N) only follow along.
#pragma once

#include "vsm_linalg.h"

namespace vsm_la {

// the lanes of a group exchange data through LDS between these points; a wavefront executes its
// LDS instructions in order, the barrier only keeps the compiler from moving accesses across it
#define VSM_GROUP_SYNC() __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); __builtin_amdgcn_wave_barrier()

template <int M, int N>
__device__ inline void svd_group(volatile double *U, volatile double *V, volatile double *W, volatile double *RV, int ln) {
#define U_(i, j) U[(i) * N + (j)]
#define V_(i, j) V[(i) * N + (j)]
  int flag, i, its, j, k, l = 0, nm = 0;
  double anorm = 0.0, c, f, g = 0.0, h, s, scale = 0.0, x, y, z;
  for (i = ln; i < N * N; i += 16) V[i] = 0.0;
  VSM_GROUP_SYNC();
  // ---- Householder reduction to bidiagonal form ----
  for (i = 0; i < N; i++) {
    l = i + 1;
    if (ln == 0) RV[i] = scale * g;
    const double rvi = scale * g;
    g = s = scale = 0.0;
    if (i < M) {
      for (k = i; k < M; k++) scale += fabs(U_(k, i));
      if (scale) {
        f = 0;
        for (k = i; k < M; k++) {
          const double t = U_(k, i) / scale;
          if (k == i) f = t;
          s += t * t;
        }
        VSM_GROUP_SYNC();
        if (ln >= i && ln < M) U_(ln, i) = U_(ln, i) / scale;  // lane k owns row k of column i
        g = -with_sign(sqrt(s), f);
        h = f * g - s;
        VSM_GROUP_SYNC();
        if (ln == 0) U_(i, i) = f - g;
        VSM_GROUP_SYNC();
        if (ln >= l && ln < N) {  // lane j owns column j
          double sj = 0.0;
          for (k = i; k < M; k++) sj += U_(k, i) * U_(k, ln);
          const double fj = sj / h;
          for (k = i; k < M; k++) U_(k, ln) += fj * U_(k, i);
        }
        VSM_GROUP_SYNC();
        if (ln >= i && ln < M) U_(ln, i) *= scale;
        VSM_GROUP_SYNC();
      }
    }
    const double wi = scale * g;
    if (ln == 0) W[i] = wi;
    g = s = scale = 0.0;
    if (i < M && i != N - 1) {
      for (k = l; k < N; k++) scale += fabs(U_(i, k));
      if (scale) {
        f = 0;
        for (k = l; k < N; k++) {
          const double t = U_(i, k) / scale;
          if (k == l) f = t;
          s += t * t;
        }
        VSM_GROUP_SYNC();
        if (ln >= l && ln < N) U_(i, ln) = U_(i, ln) / scale;
        g = -with_sign(sqrt(s), f);
        h = f * g - s;
        VSM_GROUP_SYNC();
        if (ln == 0) U_(i, l) = f - g;
        VSM_GROUP_SYNC();
        if (ln >= l && ln < N) RV[ln] = U_(i, ln) / h;
        VSM_GROUP_SYNC();
        if (ln >= l && ln < M) {  // lane j owns row j
          double sj = 0.0;
          for (k = l; k < N; k++) sj += U_(ln, k) * U_(i, k);
          for (k = l; k < N; k++) U_(ln, k) += sj * RV[k];
        }
        VSM_GROUP_SYNC();
        if (ln >= l && ln < N) U_(i, ln) *= scale;
        VSM_GROUP_SYNC();
      }
    }
    const double t = fabs(wi) + fabs(rvi);
    anorm = anorm > t ? anorm : t;
    VSM_GROUP_SYNC();
  }
  // ---- accumulate the right-hand transformations ----
  for (i = N - 1; i >= 0; i--) {
    if (i < N - 1) {
      if (g) {
        if (ln >= l && ln < N) V_(ln, i) = (U_(i, ln) / U_(i, l)) / g;
        VSM_GROUP_SYNC();
        if (ln >= l && ln < N) {  // lane j owns column j of V
          double sj = 0.0;
          for (k = l; k < N; k++) sj += U_(i, k) * V_(k, ln);
          for (k = l; k < N; k++) V_(k, ln) += sj * V_(k, i);
        }
        VSM_GROUP_SYNC();
      }
      if (ln >= l && ln < N) V_(i, ln) = V_(ln, i) = 0.0;
    }
    if (ln == 0) V_(i, i) = 1.0;
    VSM_GROUP_SYNC();
    g = RV[i];
    l = i;
  }
  // ---- accumulate the left-hand transformations ----
  for (i = (M < N ? M : N) - 1; i >= 0; i--) {
    l = i + 1;
    g = W[i];
    if (ln >= l && ln < N) U_(i, ln) = 0.0;
    VSM_GROUP_SYNC();
    if (g) {
      g = 1.0 / g;
      if (ln >= l && ln < N) {  // lane j owns column j
        double sj = 0.0;
        for (k = l; k < M; k++) sj += U_(k, i) * U_(k, ln);
        const double fj = (sj / U_(i, i)) * g;
        for (k = i; k < M; k++) U_(k, ln) += fj * U_(k, i);
      }
      VSM_GROUP_SYNC();
      if (ln >= i && ln < M) U_(ln, i) *= g;
    } else {
      if (ln >= i && ln < M) U_(ln, i) = 0.0;
    }
    VSM_GROUP_SYNC();
    if (ln == 0) U_(i, i) = U_(i, i) + 1.0;
    VSM_GROUP_SYNC();
  }
  // ---- diagonalisation of the bidiagonal form: the rotations are dealt one row per lane ----
  for (k = N - 1; k >= 0; k--) {
    for (its = 0; its < 30; its++) {
      flag = 1;
      for (l = k; l >= 0; l--) {
        nm = l - 1;
        if ((double)(fabs(RV[l]) + anorm) == anorm) {
          flag = 0;
          break;
        }
        if ((double)(fabs(W[nm]) + anorm) == anorm) break;
      }
      if (flag) {
        c = 0.0;
        s = 1.0;
        for (i = l; i <= k; i++) {
          f = s * RV[i];
          const double rvn = c * RV[i];
          VSM_GROUP_SYNC();
          if (ln == 0) RV[i] = rvn;
          if ((double)(fabs(f) + anorm) == anorm) {
            VSM_GROUP_SYNC();
            break;
          }
          g = W[i];
          h = hypot_nr(f, g);
          VSM_GROUP_SYNC();
          if (ln == 0) W[i] = h;
          h = 1.0 / h;
          c = g * h;
          s = -f * h;
          if (ln < M) {
            y = U_(ln, nm);
            z = U_(ln, i);
            U_(ln, nm) = y * c + z * s;
            U_(ln, i) = z * c - y * s;
          }
          VSM_GROUP_SYNC();
        }
      }
      z = W[k];
      if (l == k) {
        if (z < 0.0) {
          VSM_GROUP_SYNC();
          if (ln == 0) W[k] = -z;
          if (ln < N) V_(ln, k) = -V_(ln, k);
          VSM_GROUP_SYNC();
        }
        break;
      }
      x = W[l];
      nm = k - 1;
      y = W[nm];
      g = RV[nm];
      h = RV[k];
      f = ((y - z) * (y + z) + (g - h) * (g + h)) / (2.0 * h * y);
      g = hypot_nr(f, 1.0);
      f = ((x - z) * (x + z) + h * ((y / (f + with_sign(g, f))) - h)) / x;
      c = s = 1.0;
      for (j = l; j <= nm; j++) {
        i = j + 1;
        g = RV[i];
        y = W[i];
        h = s * g;
        g = c * g;
        z = hypot_nr(f, h);
        VSM_GROUP_SYNC();
        if (ln == 0) RV[j] = z;
        c = f / z;
        s = h / z;
        f = x * c + g * s;
        g = g * c - x * s;
        h = y * s;
        y *= c;
        if (ln < N) {
          const double vx = V_(ln, j), vz = V_(ln, i);
          V_(ln, j) = vx * c + vz * s;
          V_(ln, i) = vz * c - vx * s;
        }
        z = hypot_nr(f, h);
        if (ln == 0) W[j] = z;
        if (z) {
          z = 1.0 / z;
          c = f * z;
          s = h * z;
        }
        f = c * g + s * y;
        x = c * y - s * g;
        if (ln < M) {
          const double uy = U_(ln, j), uz = U_(ln, i);
          U_(ln, j) = uy * c + uz * s;
          U_(ln, i) = uz * c - uy * s;
        }
        VSM_GROUP_SYNC();
      }
      VSM_GROUP_SYNC();
      if (ln == 0) {
        RV[l] = 0.0;
        RV[k] = f;
        W[k] = x;
      }
      VSM_GROUP_SYNC();
    }
  }
  // ---- decreasing order (shell sort, increments ... 13, 4, 1): lane r moves row r of U and V ----
  int inc = 1;
  do {
    inc = inc * 3 + 1;
  } while (inc <= N);
  do {
    inc /= 3;
    for (i = inc; i < N; i++) {
      const double sw = W[i];
      const double su = ln < M ? U_(ln, i) : 0.0, sv = ln < N ? V_(ln, i) : 0.0;
      j = i;
      while (W[j - inc] < sw) {
        const double wprev = W[j - inc];
        VSM_GROUP_SYNC();
        if (ln == 0) W[j] = wprev;
        if (ln < M) U_(ln, j) = U_(ln, j - inc);
        if (ln < N) V_(ln, j) = V_(ln, j - inc);
        VSM_GROUP_SYNC();
        j -= inc;
        if (j < inc) break;
      }
      VSM_GROUP_SYNC();
      if (ln == 0) W[j] = sw;
      if (ln < M) U_(ln, j) = su;
      if (ln < N) V_(ln, j) = sv;
      VSM_GROUP_SYNC();
    }
  } while (inc > 1);
  // ---- sign convention: every lane counts redundantly, rows are flipped one per lane ----
  for (k = 0; k < N; k++) {
    int neg = 0;
    for (i = 0; i < M; i++) neg += U_(i, k) < 0.0 ? 1 : 0;
    for (j = 0; j < N; j++) neg += V_(j, k) < 0.0 ? 1 : 0;
    VSM_GROUP_SYNC();
    if (neg > (M + N) / 2) {
      if (ln < M) U_(ln, k) = -U_(ln, k);
      if (ln < N) V_(ln, k) = -V_(ln, k);
    }
    VSM_GROUP_SYNC();
  }
#undef U_
#undef V_
}

}  // namespace vsm_la
