/* TEST INFRASTRUCTURE ONLY -- see viso_oracle.h.
 *
 * Scalar C99 restatement of the reference's per-frame matcher path.  Every
 * function cites the reference file:line it follows.  Filters are written from
 * their closed forms, not from the SSE code.
 *
 * Conventions pinned here (and used identically by the HIP path):
 *   - image rows are padded to bpl = w+15-(w-1)%16 and the pad bytes are 0
 *     (the reference leaves them uninitialised; oracle/_ref pins them to 0 too);
 *   - filters see the image as a 1-D stream of bpl*h bytes (reads before the
 *     first / after the last byte give 0), exactly like the reference's
 *     row-wrapping SSE loops;
 *   - all double arithmetic is contraction-free (-ffp-contract=off).
 */
#include "viso_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

#define VO_MARGIN 6 /* Matcher::margin = 5+1, viso/matcher.cpp:56 */

static int32_t imin(int32_t a, int32_t b) { return a < b ? a : b; }
static int32_t imax(int32_t a, int32_t b) { return a > b ? a : b; }

void vo_default_params(vo_params *p) { /* viso/matcher.h:57-68 */
  memset(p, 0, sizeof(*p));
  p->nms_n = 3;
  p->nms_tau = 50;
  p->match_binsize = 50;
  p->match_radius = 200;
  p->match_disp_tolerance = 2;
  p->outlier_disp_tolerance = 5;
  p->outlier_flow_tolerance = 5;
  p->multi_stage = 1;
  p->half_resolution = 1;
  p->refinement = 1;
}

int32_t vo_bpl16(int32_t w) { return w + 15 - (w - 1) % 16; } /* viso/matcher.cpp:160,633 */

/* ------------------------------------------------------------------------- */
/* F0  createHalfResolutionImage, viso/matcher.cpp:636-647                    */
/* ------------------------------------------------------------------------- */
void vo_half_image(const uint8_t *in, int32_t w, int32_t h, int32_t bpl, uint8_t *out) {
  int32_t wh = w / 2, hh = h / 2, bh = vo_bpl16(wh);
  memset(out, 0, (size_t)bh * hh);
  for (int32_t v = 0; v < hh; v++)
    for (int32_t u = 0; u < wh; u++) {
      int32_t s = (int32_t)in[(v * 2 + 0) * bpl + u * 2 + 0] + (int32_t)in[(v * 2 + 0) * bpl + u * 2 + 1] +
                  (int32_t)in[(v * 2 + 1) * bpl + u * 2 + 0] + (int32_t)in[(v * 2 + 1) * bpl + u * 2 + 1];
      out[v * bh + u] = (uint8_t)(s / 4);
    }
}

/* ------------------------------------------------------------------------- */
/* F1  filter::sobel5x5, viso/filter.cpp:316-324 (column pass :183-233, row   */
/*     passes :105-127 and :71-100).  Called as sobel5x5(I, I_du, I_dv, bpl, h)*/
/*     (viso/matcher.cpp:675-676): out_v -> du, out_h -> dv.                  */
/* ------------------------------------------------------------------------- */
static inline int32_t px(const uint8_t *in, int64_t f, int64_t n) { return (f >= 0 && f < n) ? in[f] : 0; }

/* column pass at stream position f: smooth = (1,4,6,4,1), deriv = (1,2,0,-2,-1)
 * down the column; only defined for f in [2w, (h-2)w) (memset elsewhere, :185-186) */
static inline void sobel_cols(const uint8_t *in, int64_t f, int32_t w, int32_t h, int32_t *smooth, int32_t *deriv) {
  int64_t n = (int64_t)w * h;
  if (f < 2 * (int64_t)w || f >= (int64_t)(h - 2) * w) {
    *smooth = 0;
    *deriv = 0;
    return;
  }
  int32_t a = px(in, f - 2 * w, n), b = px(in, f - w, n), c = px(in, f, n), d = px(in, f + w, n),
          e = px(in, f + 2 * w, n);
  *smooth = a + 4 * b + 6 * c + 4 * d + e;
  *deriv = a + 2 * b - 2 * d - e;
}

static inline uint8_t pack_u8(int32_t x) { return (uint8_t)(x < 0 ? 0 : (x > 255 ? 255 : x)); } /* _mm_packus_epi16 */

void vo_sobel5x5(const uint8_t *in, uint8_t *du, uint8_t *dv, int32_t w, int32_t h) {
  int64_t n = (int64_t)w * h;
  for (int64_t f = 0; f < n; f++) {
    int32_t s[5], d[5];
    for (int k = 0; k < 5; k++) sobel_cols(in, f + k - 2, w, h, &s[k], &d[k]);
    /* du: (1,2,0,-2,-1) along the row of the column-smoothed image, >>7 arithmetic on int16, +128 */
    int32_t hu = s[0] + 2 * s[1] - 2 * s[3] - s[4];
    int32_t hv = d[0] + 4 * d[1] + 6 * d[2] + 4 * d[3] + d[4];
    du[f] = pack_u8(((int32_t)(int16_t)hu >> 7) + 128);
    dv[f] = pack_u8(((int32_t)(int16_t)hv >> 7) + 128);
  }
}

/* ------------------------------------------------------------------------- */
/* F2  filter::blob5x5, viso/filter.cpp:343-365: -box5 + 2*box3 + 7*centre     */
/*     defined for x in [3,w-3], y in [3,h-3]; 0 elsewhere (never read).       */
/* ------------------------------------------------------------------------- */
void vo_blob5x5(const uint8_t *in, int16_t *out, int32_t w, int32_t h) {
  memset(out, 0, (size_t)w * h * sizeof(int16_t));
  for (int32_t y = 3; y <= h - 3; y++)
    for (int32_t x = 3; x <= w - 3; x++) {
      int32_t b5 = 0, b3 = 0;
      for (int32_t j = -2; j <= 2; j++)
        for (int32_t i = -2; i <= 2; i++) {
          int32_t p = in[(y + j) * w + x + i];
          b5 += p;
          if (j >= -1 && j <= 1 && i >= -1 && i <= 1) b3 += p;
        }
      out[y * w + x] = (int16_t)(-b5 + 2 * b3 + 7 * (int32_t)in[y * w + x]);
    }
}

/* ------------------------------------------------------------------------- */
/* F3  filter::checkerboard5x5, viso/filter.cpp:331-336 (:235-274):            */
/*     c (x) c with c = (1,1,0,-1,-1); top-left quadrant positive.            */
/* ------------------------------------------------------------------------- */
void vo_checkerboard5x5(const uint8_t *in, int16_t *out, int32_t w, int32_t h) {
  static const int32_t c[5] = {1, 1, 0, -1, -1};
  memset(out, 0, (size_t)w * h * sizeof(int16_t));
  for (int32_t y = 2; y <= h - 3; y++)
    for (int32_t x = 2; x <= w - 3; x++) {
      int32_t s = 0;
      for (int32_t j = 0; j < 5; j++)
        for (int32_t i = 0; i < 5; i++) s += c[j] * c[i] * (int32_t)in[(y + j - 2) * w + x + i - 2];
      out[y * w + x] = (int16_t)s;
    }
}

/* ------------------------------------------------------------------------- */
/* N1  nonMaximumSuppression, viso/matcher.cpp:330-431                        */
/* ------------------------------------------------------------------------- */
typedef struct {
  int32_t u, v, val, c;
} vo_max;

typedef struct {
  vo_max *a;
  int32_t n, cap;
} vo_maxvec;

static void maxvec_push(vo_maxvec *mv, int32_t u, int32_t v, int32_t val, int32_t c) {
  if (mv->n == mv->cap) {
    mv->cap = mv->cap ? mv->cap * 2 : 1024;
    mv->a = (vo_max *)realloc(mv->a, (size_t)mv->cap * sizeof(vo_max));
  }
  vo_max m = {u, v, val, c};
  mv->a[mv->n++] = m;
}

/* does any pixel of the (2n+1)^2 window around (ci,cj), clipped at the right/bottom
 * margin and lying outside the cell [i,i+n]x[j,j+n], beat `val`?  sign=-1: smaller, +1: larger */
static int nms_suppressed(const int16_t *f, int32_t bpl, int32_t w, int32_t h, int32_t n, int32_t i, int32_t j,
                          int32_t ci, int32_t cj, int32_t val, int sign) {
  int32_t i_hi = imin(ci + n, w - 1 - VO_MARGIN), j_hi = imin(cj + n, h - 1 - VO_MARGIN);
  for (int32_t i2 = ci - n; i2 <= i_hi; i2++)
    for (int32_t j2 = cj - n; j2 <= j_hi; j2++) {
      int32_t cur = f[j2 * bpl + i2];
      int better = sign < 0 ? (cur < val) : (cur > val);
      if (better && (i2 < i || i2 > i + n || j2 < j || j2 > j + n)) return 1;
    }
  return 0;
}

static void nms_run(const int16_t *f1, const int16_t *f2, int32_t w, int32_t h, int32_t bpl, int32_t n, int32_t tau,
                    vo_maxvec *out) {
  for (int32_t i = n + VO_MARGIN; i < w - n - VO_MARGIN; i += n + 1)
    for (int32_t j = n + VO_MARGIN; j < h - n - VO_MARGIN; j += n + 1) {
      int32_t mni[2], mnj[2], mxi[2], mxj[2], mnv[2], mxv[2];
      const int16_t *fs[2] = {f1, f2};
      for (int k = 0; k < 2; k++) {
        mni[k] = mxi[k] = i;
        mnj[k] = mxj[k] = j;
        mnv[k] = mxv[k] = fs[k][j * bpl + i];
      }
      /* u outer, v inner, strict compares: first extreme in that order wins (:356-380) */
      for (int32_t i2 = i; i2 <= i + n; i2++)
        for (int32_t j2 = j; j2 <= j + n; j2++)
          for (int k = 0; k < 2; k++) {
            int32_t cur = fs[k][j2 * bpl + i2];
            if (cur < mnv[k]) {
              mni[k] = i2;
              mnj[k] = j2;
              mnv[k] = cur;
            } else if (cur > mxv[k]) {
              mxi[k] = i2;
              mxj[k] = j2;
              mxv[k] = cur;
            }
          }
      for (int k = 0; k < 2; k++) { /* class order f1min, f1max, f2min, f2max (:382-428) */
        if (!nms_suppressed(fs[k], bpl, w, h, n, i, j, mni[k], mnj[k], mnv[k], -1) && mnv[k] <= -tau)
          maxvec_push(out, mni[k], mnj[k], mnv[k], 2 * k + 0);
        if (!nms_suppressed(fs[k], bpl, w, h, n, i, j, mxi[k], mxj[k], mxv[k], +1) && mxv[k] >= tau)
          maxvec_push(out, mxi[k], mxj[k], mxv[k], 2 * k + 1);
      }
    }
}

int32_t vo_nms(const int16_t *f1, const int16_t *f2, int32_t w, int32_t h, int32_t bpl, int32_t n, int32_t tau,
               int32_t *out4, int32_t cap) {
  vo_maxvec mv = {0, 0, 0};
  nms_run(f1, f2, w, h, bpl, n, tau, &mv);
  for (int32_t k = 0; k < mv.n && k < cap; k++) {
    out4[k * 4 + 0] = mv.a[k].u;
    out4[k * 4 + 1] = mv.a[k].v;
    out4[k * 4 + 2] = mv.a[k].val;
    out4[k * 4 + 3] = mv.a[k].c;
  }
  int32_t n_out = mv.n;
  free(mv.a);
  return n_out;
}

/* ------------------------------------------------------------------------- */
/* D1  computeDescriptor, viso/matcher.cpp:433-477: 16 taps x (du,dv)          */
/* ------------------------------------------------------------------------- */
static const int8_t DESC_DV[16] = {-1, +1, -1, +1, -1, +1, -1, +1, -5, +5, -5, +5, -3, +3, -3, +3};
static const int8_t DESC_DU[16] = {-3, -3, -1, -1, +3, +3, +1, +1, -1, -1, +1, +1, -5, -5, +5, +5};

void vo_descriptor(const uint8_t *du, const uint8_t *dv, int32_t bpl, int32_t u, int32_t v, uint8_t *d) {
  for (int k = 0; k < 16; k++) {
    int32_t a = (v + DESC_DV[k]) * bpl + u + DESC_DU[k];
    d[2 * k + 0] = du[a];
    d[2 * k + 1] = dv[a];
  }
}

/* computeSmallDescriptor, viso/matcher.cpp:479-506: 12 du taps + 4 dv taps */
static void small_descriptor(const uint8_t *du, const uint8_t *dv, int32_t bpl, int32_t u, int32_t v, uint8_t *d) {
  int32_t a2 = v * bpl + u, a1 = a2 - bpl, a0 = a1 - bpl, a3 = a2 + bpl, a4 = a3 + bpl;
  d[0] = du[a0];
  d[1] = du[a1 - 2];
  d[2] = du[a1];
  d[3] = du[a1 + 2];
  d[4] = du[a2 - 1];
  d[5] = du[a2];
  d[6] = du[a2];
  d[7] = du[a2 + 1];
  d[8] = du[a3 - 2];
  d[9] = du[a3];
  d[10] = du[a3 + 2];
  d[11] = du[a4];
  d[12] = dv[a1];
  d[13] = dv[a2 - 1];
  d[14] = dv[a2 + 1];
  d[15] = dv[a3];
}

static int32_t sad_bytes(const uint8_t *a, const uint8_t *b, int32_t n) { /* viso/simd.hh:384-445 */
  int32_t s = 0;
  for (int32_t i = 0; i < n; i++) s += a[i] > b[i] ? a[i] - b[i] : b[i] - a[i];
  return s;
}

/* ------------------------------------------------------------------------- */
/* per-image feature state (what computeFeatures leaves behind, :649-732)      */
/* ------------------------------------------------------------------------- */
typedef struct {
  int32_t present;
  uint8_t *I;              /* bpl x h copy of the input (pad = 0) */
  uint8_t *du, *dv;        /* matching-resolution gradients */
  uint8_t *du_full, *dv_full; /* full-resolution gradients (half_resolution only) */
  int32_t *m1, *m2;        /* sparse / dense feature records int32[12] */
  int32_t n1, n2;
} vo_image;

static void image_free(vo_image *im) {
  free(im->I);
  free(im->du);
  free(im->dv);
  free(im->du_full);
  free(im->dv_full);
  free(im->m1);
  free(im->m2);
  memset(im, 0, sizeof(*im));
}

struct vo_matcher {
  vo_params param; /* match_radius already halved when half_resolution (:59-60) */
  vo_image img[4]; /* 0=1p 1=2p 2=1c 3=2c */
  int32_t dims_p[3], dims_c[3];
  vo_match *stage[5];
  int32_t stage_n[5];
  vo_match *matched; /* p_matched_2 */
  int32_t n_matched;
  vo_range *ranges;
  int32_t n_ranges;
  int64_t counters[8]; /* Q,C,S,M,M_out + pass-1 share Q1,C1,S1 */
};

static int32_t *pack_features(const vo_maxvec *mv, const uint8_t *du, const uint8_t *dv, int32_t bpl, int32_t s) {
  if (mv->n == 0) return 0;
  int32_t *out = (int32_t *)malloc((size_t)mv->n * 12 * sizeof(int32_t));
  for (int32_t k = 0; k < mv->n; k++) { /* :715-718 */
    int32_t *r = out + 12 * k;
    r[0] = mv->a[k].u * s;
    r[1] = mv->a[k].v * s;
    r[2] = 0;
    r[3] = mv->a[k].c;
    vo_descriptor(du, dv, bpl, mv->a[k].u, mv->a[k].v, (uint8_t *)(r + 4));
  }
  return out;
}

/* C1  computeFeatures, viso/matcher.cpp:649-732 */
static void compute_features(const vo_params *p, vo_image *im, const int32_t *dims) {
  int32_t w = dims[0], h = dims[1], bpl = dims[2];
  int32_t mw = w, mh = h, mbpl = bpl;
  const uint8_t *Im = im->I;
  uint8_t *half = 0;
  if (p->half_resolution) {
    mw = w / 2;
    mh = h / 2;
    mbpl = vo_bpl16(mw);
    half = (uint8_t *)malloc((size_t)mbpl * mh + 16);
    vo_half_image(im->I, w, h, bpl, half);
    Im = half;
    im->du_full = (uint8_t *)malloc((size_t)bpl * h);
    im->dv_full = (uint8_t *)malloc((size_t)bpl * h);
    vo_sobel5x5(im->I, im->du_full, im->dv_full, bpl, h);
  }
  im->du = (uint8_t *)malloc((size_t)mbpl * mh);
  im->dv = (uint8_t *)malloc((size_t)mbpl * mh);
  int16_t *f1 = (int16_t *)malloc((size_t)mbpl * mh * 2), *f2 = (int16_t *)malloc((size_t)mbpl * mh * 2);
  vo_sobel5x5(Im, im->du, im->dv, mbpl, mh);
  vo_blob5x5(Im, f1, mbpl, mh);
  vo_checkerboard5x5(Im, f2, mbpl, mh);
  int32_t s = p->half_resolution ? 2 : 1;
  vo_maxvec mv1 = {0, 0, 0}, mv2 = {0, 0, 0};
  if (p->multi_stage) { /* :684-690 */
    int32_t ns = p->nms_n * 3;
    if (ns > 10) ns = imax(p->nms_n, 10);
    nms_run(f1, f2, mw, mh, mbpl, ns, p->nms_tau, &mv1);
  }
  nms_run(f1, f2, mw, mh, mbpl, p->nms_n, p->nms_tau, &mv2);
  im->n1 = mv1.n;
  im->n2 = mv2.n;
  im->m1 = pack_features(&mv1, im->du, im->dv, mbpl, s);
  im->m2 = pack_features(&mv2, im->du, im->dv, mbpl, s);
  free(mv1.a);
  free(mv2.a);
  free(f1);
  free(f2);
  free(half);
}

vo_matcher *vo_create(const vo_params *p) { /* Matcher::Matcher, :33-61 */
  vo_matcher *m = (vo_matcher *)calloc(1, sizeof(vo_matcher));
  m->param = *p;
  if (p->half_resolution) m->param.match_radius /= 2;
  return m;
}

static void clear_stages(vo_matcher *m) {
  for (int s = 0; s < 5; s++) {
    free(m->stage[s]);
    m->stage[s] = 0;
    m->stage_n[s] = 0;
  }
}

void vo_destroy(vo_matcher *m) {
  if (!m) return;
  for (int k = 0; k < 4; k++) image_free(&m->img[k]);
  clear_stages(m);
  free(m->matched);
  free(m->ranges);
  free(m);
}

void vo_set_intrinsics(vo_matcher *m, double f, double cu, double cv, double base) { /* matcher.h:78-83 */
  m->param.f = f;
  m->param.cu = cu;
  m->param.cv = cv;
  m->param.base = base;
}

/* P0  pushBack, viso/matcher.cpp:95-181 */
int32_t vo_push_back(vo_matcher *m, const uint8_t *I1, const uint8_t *I2, int32_t w, int32_t h, int32_t bpl,
                     int32_t replace) {
  if (w <= 0 || h <= 0 || bpl < w || I1 == 0) return -1;
  if (replace) {
    image_free(&m->img[2]);
    image_free(&m->img[3]);
  } else {
    image_free(&m->img[0]);
    image_free(&m->img[1]);
    m->img[0] = m->img[2];
    m->img[1] = m->img[3];
    memset(&m->img[2], 0, sizeof(vo_image));
    memset(&m->img[3], 0, sizeof(vo_image));
    memcpy(m->dims_p, m->dims_c, sizeof(m->dims_p));
  }
  m->dims_c[0] = w;
  m->dims_c[1] = h;
  m->dims_c[2] = vo_bpl16(w);
  const uint8_t *src[2] = {I1, I2};
  for (int k = 0; k < 2; k++) {
    vo_image *im = &m->img[2 + k];
    /* the reference allocates I2c even for a mono push (:164) but never fills it */
    im->I = (uint8_t *)calloc((size_t)m->dims_c[2] * h + 16, 1);
    if (!src[k]) continue;
    for (int32_t v = 0; v < h; v++) memcpy(im->I + (size_t)v * m->dims_c[2], src[k] + (size_t)v * bpl, (size_t)w);
    im->present = 1;
    compute_features(&m->param, im, m->dims_c);
  }
  return 0;
}

/* ------------------------------------------------------------------------- */
/* M1  createIndexVector, viso/matcher.cpp:870-890 -- CSR form                 */
/* ------------------------------------------------------------------------- */
typedef struct {
  int32_t *start; /* bins+1 */
  int32_t *idx;   /* n, ascending feature index inside each bin */
} vo_bins;

static void bins_build(const vo_params *p, const int32_t *m, int32_t n, int32_t ub, int32_t vb, vo_bins *b) {
  int32_t nb = 4 * ub * vb;
  b->start = (int32_t *)calloc((size_t)nb + 1, sizeof(int32_t));
  b->idx = (int32_t *)malloc((size_t)(n > 0 ? n : 1) * sizeof(int32_t));
  int32_t *bin = (int32_t *)malloc((size_t)(n > 0 ? n : 1) * sizeof(int32_t));
  for (int32_t i = 0; i < n; i++) {
    int32_t u = m[12 * i + 0], v = m[12 * i + 1], c = m[12 * i + 3];
    int32_t u_bin = imin((int32_t)floorf((float)u / (float)p->match_binsize), ub - 1);
    int32_t v_bin = imin((int32_t)floorf((float)v / (float)p->match_binsize), vb - 1);
    bin[i] = (c * vb + v_bin) * ub + u_bin;
    b->start[bin[i] + 1]++;
  }
  for (int32_t k = 0; k < nb; k++) b->start[k + 1] += b->start[k];
  int32_t *cur = (int32_t *)malloc((size_t)nb * sizeof(int32_t));
  memcpy(cur, b->start, (size_t)nb * sizeof(int32_t));
  for (int32_t i = 0; i < n; i++) b->idx[cur[bin[i]]++] = i;
  free(cur);
  free(bin);
}

static void bins_free(vo_bins *b) {
  free(b->start);
  free(b->idx);
}

/* M2  findMatch, viso/matcher.cpp:892-963 */
static int32_t find_match(vo_matcher *M, const int32_t *m1, int32_t i1, const int32_t *m2, const vo_bins *k2,
                          int32_t ub, int32_t vb, int32_t stat_bin, int32_t stage, int flow, int use_prior, double u_,
                          double v_) {
  const vo_params *p = &M->param;
  int32_t min_ind = 0;
  double min_cost = 10000000;
  int32_t u1 = m1[12 * i1 + 0], v1 = m1[12 * i1 + 1], c = m1[12 * i1 + 3];
  const uint8_t *d1 = (const uint8_t *)(m1 + 12 * i1 + 4);
  float u_min, u_max, v_min, v_max;
  if (use_prior) {
    u_min = u1 + M->ranges[stat_bin].u_min[stage];
    u_max = u1 + M->ranges[stat_bin].u_max[stage];
    v_min = v1 + M->ranges[stat_bin].v_min[stage];
    v_max = v1 + M->ranges[stat_bin].v_max[stage];
  } else {
    u_min = (float)(u1 - p->match_radius);
    u_max = (float)(u1 + p->match_radius);
    v_min = (float)(v1 - p->match_radius);
    v_max = (float)(v1 + p->match_radius);
  }
  if (!flow) {
    v_min = (float)(v1 - p->match_disp_tolerance);
    v_max = (float)(v1 + p->match_disp_tolerance);
  }
  float bs = (float)p->match_binsize;
  int32_t u_bin_min = imin(imax((int32_t)floorf(u_min / bs), 0), ub - 1);
  int32_t u_bin_max = imin(imax((int32_t)floorf(u_max / bs), 0), ub - 1);
  int32_t v_bin_min = imin(imax((int32_t)floorf(v_min / bs), 0), vb - 1);
  int32_t v_bin_max = imin(imax((int32_t)floorf(v_max / bs), 0), vb - 1);
  M->counters[0]++;
  for (int32_t u_bin = u_bin_min; u_bin <= u_bin_max; u_bin++)
    for (int32_t v_bin = v_bin_min; v_bin <= v_bin_max; v_bin++) {
      int32_t k = (c * vb + v_bin) * ub + u_bin;
      for (int32_t q = k2->start[k]; q < k2->start[k + 1]; q++) {
        int32_t i2 = k2->idx[q];
        int32_t u2 = m2[12 * i2 + 0], v2 = m2[12 * i2 + 1];
        M->counters[1]++;
        if ((float)u2 >= u_min && (float)u2 <= u_max && (float)v2 >= v_min && (float)v2 <= v_max) {
          M->counters[2]++;
          double cost = (double)sad_bytes(d1, (const uint8_t *)(m2 + 12 * i2 + 4), 32);
          if (u_ >= 0 && v_ >= 0) {
            double du = (double)u2 - u_;
            double dv = (double)v2 - v_;
            double dist = sqrt(du * du + dv * dv);
            cost += 4 * dist;
          }
          if (cost < min_cost) {
            min_ind = i2;
            min_cost = cost;
          }
        }
      }
    }
  return min_ind;
}

typedef struct {
  vo_match *a;
  int32_t n, cap;
} vo_matchvec;

static void mv_push(vo_matchvec *v, float u1p, float v1p, int32_t i1p, float u2p, float v2p, int32_t i2p, float u1c,
                    float v1c, int32_t i1c, float u2c, float v2c, int32_t i2c) {
  if (v->n == v->cap) {
    v->cap = v->cap ? v->cap * 2 : 1024;
    v->a = (vo_match *)realloc(v->a, (size_t)v->cap * sizeof(vo_match));
  }
  vo_match m = {u1p, v1p, i1p, u2p, v2p, i2p, u1c, v1c, i1c, u2c, v2c, i2c};
  v->a[v->n++] = m;
}

/* M3  matching, viso/matcher.cpp:965-1205 */
static void matching(vo_matcher *M, int sparse, vo_matchvec *out, int32_t method, int use_prior, const double *Tr) {
  const vo_params *p = &M->param;
  const int32_t *m1p = sparse ? M->img[0].m1 : M->img[0].m2, *m2p = sparse ? M->img[1].m1 : M->img[1].m2;
  const int32_t *m1c = sparse ? M->img[2].m1 : M->img[2].m2, *m2c = sparse ? M->img[3].m1 : M->img[3].m2;
  int32_t n1p = sparse ? M->img[0].n1 : M->img[0].n2, n2p = sparse ? M->img[1].n1 : M->img[1].n2;
  int32_t n1c = sparse ? M->img[2].n1 : M->img[2].n2, n2c = sparse ? M->img[3].n1 : M->img[3].n2;
  int32_t ub = (int32_t)ceilf((float)M->dims_c[0] / (float)p->match_binsize);
  int32_t vb = (int32_t)ceilf((float)M->dims_c[1] / (float)p->match_binsize);
  float bs = (float)p->match_binsize;
  uint8_t *Mpix = (uint8_t *)calloc((size_t)M->dims_c[0] * M->dims_c[1], 1);
  double t00 = 0, t01 = 0, t02 = 0, t03 = 0, t10 = 0, t11 = 0, t12 = 0, t13 = 0, t20 = 0, t21 = 0, t22 = 0, t23 = 0;
  if (Tr) {
    t00 = Tr[0]; t01 = Tr[1]; t02 = Tr[2]; t03 = Tr[3];
    t10 = Tr[4]; t11 = Tr[5]; t12 = Tr[6]; t13 = Tr[7];
    t20 = Tr[8]; t21 = Tr[9]; t22 = Tr[10]; t23 = Tr[11];
  }
  if (method == 0) { /* flow, :1006-1041 */
    vo_bins k1p, k1c;
    bins_build(p, m1p, n1p, ub, vb, &k1p);
    bins_build(p, m1c, n1c, ub, vb, &k1c);
    for (int32_t i1c = 0; i1c < n1c; i1c++) {
      int32_t u1c = m1c[12 * i1c], v1c = m1c[12 * i1c + 1];
      int32_t u_bin = imin((int32_t)floorf((float)u1c / bs), ub - 1);
      int32_t v_bin = imin((int32_t)floorf((float)v1c / bs), vb - 1);
      int32_t stat_bin = v_bin * ub + u_bin;
      int32_t i1p = find_match(M, m1c, i1c, m1p, &k1p, ub, vb, stat_bin, 0, 1, use_prior, -1, -1);
      int32_t i1c2 = find_match(M, m1p, i1p, m1c, &k1c, ub, vb, stat_bin, 1, 1, use_prior, -1, -1);
      if (i1c2 == i1c) {
        int32_t u1p = m1p[12 * i1p], v1p = m1p[12 * i1p + 1];
        if (Mpix[v1c * M->dims_c[0] + u1c] == 0) {
          mv_push(out, (float)u1p, (float)v1p, i1p, -1, -1, -1, (float)u1c, (float)v1c, i1c, -1, -1, -1);
          Mpix[v1c * M->dims_c[0] + u1c] = 1;
        }
      }
    }
    bins_free(&k1p);
    bins_free(&k1c);
  } else if (method == 1) { /* stereo, :1045-1084 */
    vo_bins k1c, k2c;
    bins_build(p, m1c, n1c, ub, vb, &k1c);
    bins_build(p, m2c, n2c, ub, vb, &k2c);
    for (int32_t i1c = 0; i1c < n1c; i1c++) {
      int32_t u1c = m1c[12 * i1c], v1c = m1c[12 * i1c + 1];
      int32_t u_bin = imin((int32_t)floorf((float)u1c / bs), ub - 1);
      int32_t v_bin = imin((int32_t)floorf((float)v1c / bs), vb - 1);
      int32_t stat_bin = v_bin * ub + u_bin;
      int32_t i2c = find_match(M, m1c, i1c, m2c, &k2c, ub, vb, stat_bin, 0, 0, use_prior, -1, -1);
      int32_t i1c2 = find_match(M, m2c, i2c, m1c, &k1c, ub, vb, stat_bin, 1, 0, use_prior, -1, -1);
      if (i1c2 == i1c) {
        int32_t u2c = m2c[12 * i2c], v2c = m2c[12 * i2c + 1];
        if (u1c >= u2c && Mpix[v1c * M->dims_c[0] + u1c] == 0) {
          mv_push(out, -1, -1, -1, -1, -1, -1, (float)u1c, (float)v1c, i1c, (float)u2c, (float)v2c, i2c);
          Mpix[v1c * M->dims_c[0] + u1c] = 1;
        }
      }
    }
    bins_free(&k1c);
    bins_free(&k2c);
  } else { /* quad, :1088-1153 */
    vo_bins k1p, k2p, k1c, k2c;
    bins_build(p, m1p, n1p, ub, vb, &k1p);
    bins_build(p, m2p, n2p, ub, vb, &k2p);
    bins_build(p, m1c, n1c, ub, vb, &k1c);
    bins_build(p, m2c, n2c, ub, vb, &k2c);
    for (int32_t i1p = 0; i1p < n1p; i1p++) {
      int32_t u1p = m1p[12 * i1p], v1p = m1p[12 * i1p + 1];
      int32_t u_bin = imin((int32_t)floorf((float)u1p / bs), ub - 1);
      int32_t v_bin = imin((int32_t)floorf((float)v1p / bs), vb - 1);
      int32_t stat_bin = v_bin * ub + u_bin;
      int32_t i2p = find_match(M, m1p, i1p, m2p, &k2p, ub, vb, stat_bin, 0, 0, use_prior, -1, -1);
      int32_t u2p = m2p[12 * i2p], v2p = m2p[12 * i2p + 1];
      int32_t i2c, i1c, i1p2;
      if (Tr) { /* :1114-1128 */
        double d = (double)u1p - (double)u2p;
        if (!(d > 1.0)) d = 1.0; /* max(d,1.0) */
        double x1p = ((double)u1p - p->cu) * p->base / d;
        double y1p = ((double)v1p - p->cv) * p->base / d;
        double z1p = p->f * p->base / d;
        double x2c = t00 * x1p + t01 * y1p + t02 * z1p + t03 - p->base;
        double y2c = t10 * x1p + t11 * y1p + t12 * z1p + t13;
        double z2c = t20 * x1p + t21 * y1p + t22 * z1p + t23;
        double u2c_ = p->f * x2c / z2c + p->cu;
        double v2c_ = p->f * y2c / z2c + p->cv;
        i2c = find_match(M, m2p, i2p, m2c, &k2c, ub, vb, stat_bin, 1, 1, use_prior, u2c_, v2c_);
      } else {
        i2c = find_match(M, m2p, i2p, m2c, &k2c, ub, vb, stat_bin, 1, 1, use_prior, -1, -1);
      }
      i1c = find_match(M, m2c, i2c, m1c, &k1c, ub, vb, stat_bin, 2, 0, use_prior, -1, -1);
      if (Tr)
        i1p2 = find_match(M, m1c, i1c, m1p, &k1p, ub, vb, stat_bin, 3, 1, use_prior, (double)u1p, (double)v1p);
      else
        i1p2 = find_match(M, m1c, i1c, m1p, &k1p, ub, vb, stat_bin, 3, 1, use_prior, -1, -1);
      if (i1p2 == i1p) {
        int32_t u2c = m2c[12 * i2c], v2c = m2c[12 * i2c + 1];
        int32_t u1c = m1c[12 * i1c], v1c = m1c[12 * i1c + 1];
        if (u1p >= u2p && u1c >= u2c)
          mv_push(out, (float)u1p, (float)v1p, i1p, (float)u2p, (float)v2p, i2p, (float)u1c, (float)v1c, i1c,
                  (float)u2c, (float)v2c, i2c);
      }
    }
    bins_free(&k1p);
    bins_free(&k2p);
    bins_free(&k1c);
    bins_free(&k2c);
  }
  free(Mpix);
}

/* ------------------------------------------------------------------------- */
/* M4  computePriorStatistics, viso/matcher.cpp:734-868                        */
/* ------------------------------------------------------------------------- */
static void prior_statistics(vo_matcher *M, const vo_match *pm, int32_t n, int32_t method) {
  const vo_params *p = &M->param;
  int32_t ub = (int32_t)ceilf((float)M->dims_c[0] / (float)p->match_binsize);
  int32_t vb = (int32_t)ceilf((float)M->dims_c[1] / (float)p->match_binsize);
  int32_t nb = ub * vb, num_stages = method == 2 ? 4 : 2;
  float bs = (float)p->match_binsize;
  float *dmin = (float *)malloc((size_t)nb * 8 * sizeof(float)), *dmax = (float *)malloc((size_t)nb * 8 * sizeof(float));
  int32_t *cnt = (int32_t *)calloc((size_t)nb, sizeof(int32_t));
  for (int32_t k = 0; k < nb * 8; k++) {
    dmin[k] = +1000000.f;
    dmax[k] = -1000000.f;
  }
  for (int32_t q = 0; q < n; q++) {
    const vo_match *it = &pm[q];
    float d[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    float ur, vr;
    if (method == 0) {
      d[0] = it->u1p - it->u1c;
      d[1] = it->v1p - it->v1c;
      d[2] = it->u1c - it->u1p;
      d[3] = it->v1c - it->v1p;
    } else if (method == 1) {
      d[0] = it->u2c - it->u1c;
      d[2] = it->u1c - it->u2c;
    } else {
      d[0] = it->u2p - it->u1p;
      d[2] = it->u2c - it->u2p;
      d[3] = it->v2c - it->v2p;
      d[4] = it->u1c - it->u2c;
      d[6] = it->u1p - it->u1c;
      d[7] = it->v1p - it->v1c;
    }
    if (method < 2) {
      ur = it->u1c;
      vr = it->v1c;
    } else {
      ur = it->u1p;
      vr = it->v1p;
    }
    int32_t ubin = (int32_t)floorf(ur / bs), vbin = (int32_t)floorf(vr / bs);
    int32_t u0 = imin(imax(ubin - 1, 0), ub - 1), u1 = imin(imax(ubin + 1, 0), ub - 1);
    int32_t v0 = imin(imax(vbin - 1, 0), vb - 1), v1 = imin(imax(vbin + 1, 0), vb - 1);
    for (int32_t v = v0; v <= v1; v++)
      for (int32_t u = u0; u <= u1; u++) {
        int32_t b = v * ub + u;
        cnt[b]++;
        for (int32_t i = 0; i < num_stages * 2; i++) {
          if (d[i] < dmin[b * 8 + i]) dmin[b * 8 + i] = d[i];
          if (d[i] > dmax[b * 8 + i]) dmax[b * 8 + i] = d[i];
        }
      }
  }
  free(M->ranges);
  M->ranges = (vo_range *)calloc((size_t)nb, sizeof(vo_range));
  M->n_ranges = nb;
  for (int32_t b = 0; b < nb; b++) {
    float lo[8], hi[8];
    for (int i = 0; i < 8; i++) {
      if (cnt[b] > 0) {
        lo[i] = dmin[b * 8 + i];
        hi[i] = dmax[b * 8 + i];
      } else {
        lo[i] = (float)(-p->match_radius);
        hi[i] = (float)(+p->match_radius);
      }
    }
    if (cnt[b] > 0) /* stages beyond num_stages keep the +-1e6 init of the reference (:827-828) */
      for (int i = num_stages * 2; i < 8; i++) {
        lo[i] = +1000000.f;
        hi[i] = -1000000.f;
      }
    vo_range *r = &M->ranges[b];
    for (int32_t i = 0; i < num_stages; i++) { /* :842-861 */
      float delta_u = hi[i * 2 + 0] - lo[i * 2 + 0];
      if (delta_u < 20) {
        lo[i * 2 + 0] -= ceilf((20 - delta_u) / 2);
        hi[i * 2 + 0] += ceilf((20 - delta_u) / 2);
      }
      float delta_v = hi[i * 2 + 1] - lo[i * 2 + 1];
      if (delta_v < 20) {
        lo[i * 2 + 1] -= ceilf((20 - delta_v) / 2);
        hi[i * 2 + 1] += ceilf((20 - delta_v) / 2);
      }
      r->u_min[i] = lo[i * 2 + 0];
      r->u_max[i] = hi[i * 2 + 0];
      r->v_min[i] = lo[i * 2 + 1];
      r->v_max[i] = hi[i * 2 + 1];
    }
  }
  free(dmin);
  free(dmax);
  free(cnt);
}

/* ------------------------------------------------------------------------- */
/* Delaunay triangulation: restates Triangle 1.6's divide-and-conquer          */
/* (viso/triangle.cpp: vertexsort :5447, vertexmedian :5513, alternateaxes     */
/* :5583, mergehulls :5639, divconqrecurse :5963, divconqdelaunay :6161) on an */
/* index-based triangle store with exact int64 predicates.  Inputs are         */
/* integer-valued (u1c,v1c are never refined), so the sign of Triangle's       */
/* adaptive float predicates (:2707, :3335) equals the integer sign.           */
/* ------------------------------------------------------------------------- */
typedef struct {
  int32_t t, o;
} otri; /* oriented triangle handle */

typedef struct {
  const int32_t *x, *y;
  int32_t *nb;  /* [t*3+o] encoded neighbour t2*4+o2, or -1 */
  int32_t *vx;  /* [t*3+o] vertex id or -1 (the "NULL" ghost apex) */
  int32_t ntri, cap;
  uint64_t seed; /* randomseed, :550/:4031 */
} dmesh;

static const int P1[3] = {1, 2, 0}, M1[3] = {2, 0, 1};

static otri o_sym(const dmesh *m, otri a) {
  int32_t e = m->nb[a.t * 3 + a.o];
  otri r = {e >> 2, e & 3};
  return r;
}
static otri o_lnext(otri a) { otri r = {a.t, P1[a.o]}; return r; }
static otri o_lprev(otri a) { otri r = {a.t, M1[a.o]}; return r; }
static int32_t o_org(const dmesh *m, otri a) { return m->vx[a.t * 3 + P1[a.o]]; }
static int32_t o_dest(const dmesh *m, otri a) { return m->vx[a.t * 3 + M1[a.o]]; }
static int32_t o_apex(const dmesh *m, otri a) { return m->vx[a.t * 3 + a.o]; }
static void o_setorg(dmesh *m, otri a, int32_t v) { m->vx[a.t * 3 + P1[a.o]] = v; }
static void o_setdest(dmesh *m, otri a, int32_t v) { m->vx[a.t * 3 + M1[a.o]] = v; }
static void o_setapex(dmesh *m, otri a, int32_t v) { m->vx[a.t * 3 + a.o] = v; }
static void o_bond(dmesh *m, otri a, otri b) {
  m->nb[a.t * 3 + a.o] = b.t * 4 + b.o;
  m->nb[b.t * 3 + b.o] = a.t * 4 + a.o;
}
static otri o_make(dmesh *m) {
  int32_t t = m->ntri++;
  for (int k = 0; k < 3; k++) {
    m->nb[t * 3 + k] = -1;
    m->vx[t * 3 + k] = -1;
  }
  otri r = {t, 0};
  return r;
}

static int64_t d_ccw(const dmesh *m, int32_t a, int32_t b, int32_t c) {
  return (int64_t)(m->x[a] - m->x[c]) * (m->y[b] - m->y[c]) - (int64_t)(m->y[a] - m->y[c]) * (m->x[b] - m->x[c]);
}

static int64_t d_incircle(const dmesh *m, int32_t a, int32_t b, int32_t c, int32_t d) {
  int64_t adx = m->x[a] - m->x[d], ady = m->y[a] - m->y[d];
  int64_t bdx = m->x[b] - m->x[d], bdy = m->y[b] - m->y[d];
  int64_t cdx = m->x[c] - m->x[d], cdy = m->y[c] - m->y[d];
  int64_t al = adx * adx + ady * ady, bl = bdx * bdx + bdy * bdy, cl = cdx * cdx + cdy * cdy;
  return al * (bdx * cdy - cdx * bdy) + bl * (cdx * ady - adx * cdy) + cl * (adx * bdy - bdx * ady);
}

static uint32_t d_random(dmesh *m, uint32_t choices) { /* randomnation, :4046-4050 */
  m->seed = (m->seed * 1366u + 150889u) % 714025u;
  return (uint32_t)(m->seed / (714025u / choices + 1));
}

/* key order: primary coordinate `axis`, secondary the other one */
static int d_less(const dmesh *m, int32_t a, int32_t pk1, int32_t pk2, int axis) {
  int32_t k1 = axis ? m->y[a] : m->x[a], k2 = axis ? m->x[a] : m->y[a];
  return k1 < pk1 || (k1 == pk1 && k2 < pk2);
}
static int d_greater(const dmesh *m, int32_t a, int32_t pk1, int32_t pk2, int axis) {
  int32_t k1 = axis ? m->y[a] : m->x[a], k2 = axis ? m->x[a] : m->y[a];
  return k1 > pk1 || (k1 == pk1 && k2 > pk2);
}

/* Hoare partition shared by vertexsort (:5447) and vertexmedian (:5513) */
static void d_partition(dmesh *m, int32_t *arr, int32_t n, int axis, int32_t *left_out, int32_t *right_out) {
  int32_t pivot = (int32_t)d_random(m, (uint32_t)n);
  int32_t pk1 = axis ? m->y[arr[pivot]] : m->x[arr[pivot]];
  int32_t pk2 = axis ? m->x[arr[pivot]] : m->y[arr[pivot]];
  int32_t left = -1, right = n;
  while (left < right) {
    do {
      left++;
    } while (left <= right && d_less(m, arr[left], pk1, pk2, axis));
    do {
      right--;
    } while (left <= right && d_greater(m, arr[right], pk1, pk2, axis));
    if (left < right) {
      int32_t t = arr[left];
      arr[left] = arr[right];
      arr[right] = t;
    }
  }
  *left_out = left;
  *right_out = right;
}

static void d_sort2(const dmesh *m, int32_t *arr, int axis) {
  int32_t pk1 = axis ? m->y[arr[1]] : m->x[arr[1]], pk2 = axis ? m->x[arr[1]] : m->y[arr[1]];
  if (d_greater(m, arr[0], pk1, pk2, axis)) {
    int32_t t = arr[1];
    arr[1] = arr[0];
    arr[0] = t;
  }
}

static void d_vertexsort(dmesh *m, int32_t *arr, int32_t n) {
  if (n == 2) {
    d_sort2(m, arr, 0);
    return;
  }
  int32_t left, right;
  d_partition(m, arr, n, 0, &left, &right);
  if (left > 1) d_vertexsort(m, arr, left);
  if (right < n - 2) d_vertexsort(m, arr + right + 1, n - right - 1);
}

static void d_vertexmedian(dmesh *m, int32_t *arr, int32_t n, int32_t median, int axis) {
  if (n == 2) {
    d_sort2(m, arr, axis);
    return;
  }
  int32_t left, right;
  d_partition(m, arr, n, axis, &left, &right);
  if (left > median) d_vertexmedian(m, arr, left, median, axis);
  if (right < median - 1) d_vertexmedian(m, arr + right + 1, n - right - 1, median - right - 1, axis);
}

static void d_alternateaxes(dmesh *m, int32_t *arr, int32_t n, int axis) {
  int32_t divider = n >> 1;
  if (n <= 3) axis = 0;
  d_vertexmedian(m, arr, n, divider, axis);
  if (n - divider >= 2) {
    if (divider >= 2) d_alternateaxes(m, arr, divider, 1 - axis);
    d_alternateaxes(m, arr + divider, n - divider, 1 - axis);
  }
}

/* mergehulls, viso/triangle.cpp:5639-5940 */
static void d_mergehulls(dmesh *m, otri *farleft, otri *innerleft, otri *innerright, otri *farright, int axis) {
  otri leftcand, rightcand, baseedge, nextedge, sidecasing, topcasing, outercasing, checkedge;
  int32_t innerleftdest = o_dest(m, *innerleft), innerleftapex = o_apex(m, *innerleft);
  int32_t innerrightorg = o_org(m, *innerright), innerrightapex = o_apex(m, *innerright);
  int32_t farleftpt, farrightpt, farleftapex, farrightapex, checkvertex;
  if (axis == 1) { /* horizontal cut: move the handles to bottom/top-most vertices (:5666-5702) */
    farleftpt = o_org(m, *farleft);
    farleftapex = o_apex(m, *farleft);
    farrightpt = o_dest(m, *farright);
    farrightapex = o_apex(m, *farright);
    while (m->y[farleftapex] < m->y[farleftpt]) {
      *farleft = o_sym(m, o_lnext(*farleft));
      farleftpt = farleftapex;
      farleftapex = o_apex(m, *farleft);
    }
    checkedge = o_sym(m, *innerleft);
    checkvertex = o_apex(m, checkedge);
    while (m->y[checkvertex] > m->y[innerleftdest]) {
      *innerleft = o_lnext(checkedge);
      innerleftapex = innerleftdest;
      innerleftdest = checkvertex;
      checkedge = o_sym(m, *innerleft);
      checkvertex = o_apex(m, checkedge);
    }
    while (m->y[innerrightapex] < m->y[innerrightorg]) {
      *innerright = o_sym(m, o_lnext(*innerright));
      innerrightorg = innerrightapex;
      innerrightapex = o_apex(m, *innerright);
    }
    checkedge = o_sym(m, *farright);
    checkvertex = o_apex(m, checkedge);
    while (m->y[checkvertex] > m->y[farrightpt]) {
      *farright = o_lnext(checkedge);
      farrightapex = farrightpt;
      farrightpt = checkvertex;
      checkedge = o_sym(m, *farright);
      checkvertex = o_apex(m, checkedge);
    }
  }
  /* lower common tangent (:5704-5725) */
  int changemade;
  do {
    changemade = 0;
    if (d_ccw(m, innerleftdest, innerleftapex, innerrightorg) > 0) {
      *innerleft = o_sym(m, o_lprev(*innerleft));
      innerleftdest = innerleftapex;
      innerleftapex = o_apex(m, *innerleft);
      changemade = 1;
    }
    if (d_ccw(m, innerrightapex, innerrightorg, innerleftdest) > 0) {
      *innerright = o_sym(m, o_lnext(*innerright));
      innerrightorg = innerrightapex;
      innerrightapex = o_apex(m, *innerright);
      changemade = 1;
    }
  } while (changemade);
  leftcand = o_sym(m, *innerleft);
  rightcand = o_sym(m, *innerright);
  baseedge = o_make(m); /* bottom bounding triangle (:5729-5742) */
  o_bond(m, baseedge, *innerleft);
  baseedge = o_lnext(baseedge);
  o_bond(m, baseedge, *innerright);
  baseedge = o_lnext(baseedge);
  o_setorg(m, baseedge, innerrightorg);
  o_setdest(m, baseedge, innerleftdest);
  farleftpt = o_org(m, *farleft);
  if (innerleftdest == farleftpt) *farleft = o_lnext(baseedge);
  farrightpt = o_dest(m, *farright);
  if (innerrightorg == farrightpt) *farright = o_lprev(baseedge);
  int32_t lowerleft = innerleftdest, lowerright = innerrightorg;
  int32_t upperleft = o_apex(m, leftcand), upperright = o_apex(m, rightcand);
  for (;;) {
    int leftfinished = d_ccw(m, upperleft, lowerleft, lowerright) <= 0;
    int rightfinished = d_ccw(m, upperright, lowerleft, lowerright) <= 0;
    if (leftfinished && rightfinished) { /* top bounding triangle (:5771-5812) */
      nextedge = o_make(m);
      o_setorg(m, nextedge, lowerleft);
      o_setdest(m, nextedge, lowerright);
      o_bond(m, nextedge, baseedge);
      nextedge = o_lnext(nextedge);
      o_bond(m, nextedge, rightcand);
      nextedge = o_lnext(nextedge);
      o_bond(m, nextedge, leftcand);
      if (axis == 1) { /* restore left/right-most handles */
        farleftpt = o_org(m, *farleft);
        farleftapex = o_apex(m, *farleft);
        farrightpt = o_dest(m, *farright);
        farrightapex = o_apex(m, *farright);
        checkedge = o_sym(m, *farleft);
        checkvertex = o_apex(m, checkedge);
        while (m->x[checkvertex] < m->x[farleftpt]) {
          *farleft = o_lprev(checkedge);
          farleftapex = farleftpt;
          farleftpt = checkvertex;
          checkedge = o_sym(m, *farleft);
          checkvertex = o_apex(m, checkedge);
        }
        while (m->x[farrightapex] > m->x[farrightpt]) {
          *farright = o_sym(m, o_lprev(*farright));
          farrightpt = farrightapex;
          farrightapex = o_apex(m, *farright);
        }
      }
      return;
    }
    if (!leftfinished) { /* eat non-Delaunay edges of the left hull (:5814-5860) */
      nextedge = o_sym(m, o_lprev(leftcand));
      int32_t nextapex = o_apex(m, nextedge);
      if (nextapex >= 0) {
        int badedge = d_incircle(m, lowerleft, lowerright, upperleft, nextapex) > 0;
        while (badedge) {
          nextedge = o_lnext(nextedge);
          topcasing = o_sym(m, nextedge);
          nextedge = o_lnext(nextedge);
          sidecasing = o_sym(m, nextedge);
          o_bond(m, nextedge, topcasing);
          o_bond(m, leftcand, sidecasing);
          leftcand = o_lnext(leftcand);
          outercasing = o_sym(m, leftcand);
          nextedge = o_lprev(nextedge);
          o_bond(m, nextedge, outercasing);
          o_setorg(m, leftcand, lowerleft);
          o_setdest(m, leftcand, -1);
          o_setapex(m, leftcand, nextapex);
          o_setorg(m, nextedge, -1);
          o_setdest(m, nextedge, upperleft);
          o_setapex(m, nextedge, nextapex);
          upperleft = nextapex;
          nextedge = sidecasing;
          nextapex = o_apex(m, nextedge);
          badedge = nextapex >= 0 ? d_incircle(m, lowerleft, lowerright, upperleft, nextapex) > 0 : 0;
        }
      }
    }
    if (!rightfinished) { /* same for the right hull (:5862-5908) */
      nextedge = o_sym(m, o_lnext(rightcand));
      int32_t nextapex = o_apex(m, nextedge);
      if (nextapex >= 0) {
        int badedge = d_incircle(m, lowerleft, lowerright, upperright, nextapex) > 0;
        while (badedge) {
          nextedge = o_lprev(nextedge);
          topcasing = o_sym(m, nextedge);
          nextedge = o_lprev(nextedge);
          sidecasing = o_sym(m, nextedge);
          o_bond(m, nextedge, topcasing);
          o_bond(m, rightcand, sidecasing);
          rightcand = o_lprev(rightcand);
          outercasing = o_sym(m, rightcand);
          nextedge = o_lnext(nextedge);
          o_bond(m, nextedge, outercasing);
          o_setorg(m, rightcand, -1);
          o_setdest(m, rightcand, lowerright);
          o_setapex(m, rightcand, nextapex);
          o_setorg(m, nextedge, upperright);
          o_setdest(m, nextedge, -1);
          o_setapex(m, nextedge, nextapex);
          upperright = nextapex;
          nextedge = sidecasing;
          nextapex = o_apex(m, nextedge);
          badedge = nextapex >= 0 ? d_incircle(m, lowerleft, lowerright, upperright, nextapex) > 0 : 0;
        }
      }
    }
    if (leftfinished || (!rightfinished && d_incircle(m, upperleft, lowerleft, lowerright, upperright) > 0)) {
      /* knit lowerleft -- upperright (:5911-5919) */
      o_bond(m, baseedge, rightcand);
      baseedge = o_lprev(rightcand);
      o_setdest(m, baseedge, lowerleft);
      lowerright = upperright;
      rightcand = o_sym(m, baseedge);
      upperright = o_apex(m, rightcand);
    } else { /* knit upperleft -- lowerright (:5920-5929) */
      o_bond(m, baseedge, leftcand);
      baseedge = o_lnext(leftcand);
      o_setorg(m, baseedge, lowerright);
      lowerleft = upperleft;
      leftcand = o_sym(m, baseedge);
      upperleft = o_apex(m, leftcand);
    }
  }
}

/* divconqrecurse, viso/triangle.cpp:5963-6107 */
static void d_recurse(dmesh *m, int32_t *arr, int32_t n, int axis, otri *farleft, otri *farright) {
  if (n == 2) {
    *farleft = o_make(m);
    o_setorg(m, *farleft, arr[0]);
    o_setdest(m, *farleft, arr[1]);
    *farright = o_make(m);
    o_setorg(m, *farright, arr[1]);
    o_setdest(m, *farright, arr[0]);
    o_bond(m, *farleft, *farright);
    *farleft = o_lprev(*farleft);
    *farright = o_lnext(*farright);
    o_bond(m, *farleft, *farright);
    *farleft = o_lprev(*farleft);
    *farright = o_lnext(*farright);
    o_bond(m, *farleft, *farright);
    *farleft = o_lprev(*farright);
    return;
  }
  if (n == 3) {
    otri midtri = o_make(m), tri1 = o_make(m), tri2 = o_make(m), tri3 = o_make(m);
    int64_t area = d_ccw(m, arr[0], arr[1], arr[2]);
    if (area == 0) { /* collinear: two edges */
      o_setorg(m, midtri, arr[0]);
      o_setdest(m, midtri, arr[1]);
      o_setorg(m, tri1, arr[1]);
      o_setdest(m, tri1, arr[0]);
      o_setorg(m, tri2, arr[2]);
      o_setdest(m, tri2, arr[1]);
      o_setorg(m, tri3, arr[1]);
      o_setdest(m, tri3, arr[2]);
      o_bond(m, midtri, tri1);
      o_bond(m, tri2, tri3);
      midtri = o_lnext(midtri);
      tri1 = o_lprev(tri1);
      tri2 = o_lnext(tri2);
      tri3 = o_lprev(tri3);
      o_bond(m, midtri, tri3);
      o_bond(m, tri1, tri2);
      midtri = o_lnext(midtri);
      tri1 = o_lprev(tri1);
      tri2 = o_lnext(tri2);
      tri3 = o_lprev(tri3);
      o_bond(m, midtri, tri1);
      o_bond(m, tri2, tri3);
      *farleft = tri1;
      *farright = tri2;
    } else {
      o_setorg(m, midtri, arr[0]);
      o_setdest(m, tri1, arr[0]);
      o_setorg(m, tri3, arr[0]);
      if (area > 0) {
        o_setdest(m, midtri, arr[1]);
        o_setorg(m, tri1, arr[1]);
        o_setdest(m, tri2, arr[1]);
        o_setapex(m, midtri, arr[2]);
        o_setorg(m, tri2, arr[2]);
        o_setdest(m, tri3, arr[2]);
      } else {
        o_setdest(m, midtri, arr[2]);
        o_setorg(m, tri1, arr[2]);
        o_setdest(m, tri2, arr[2]);
        o_setapex(m, midtri, arr[1]);
        o_setorg(m, tri2, arr[1]);
        o_setdest(m, tri3, arr[1]);
      }
      o_bond(m, midtri, tri1);
      midtri = o_lnext(midtri);
      o_bond(m, midtri, tri2);
      midtri = o_lnext(midtri);
      o_bond(m, midtri, tri3);
      tri1 = o_lprev(tri1);
      tri2 = o_lnext(tri2);
      o_bond(m, tri1, tri2);
      tri1 = o_lprev(tri1);
      tri3 = o_lprev(tri3);
      o_bond(m, tri1, tri3);
      tri2 = o_lnext(tri2);
      tri3 = o_lprev(tri3);
      o_bond(m, tri2, tri3);
      *farleft = tri1;
      if (area > 0)
        *farright = tri2;
      else
        *farright = o_lnext(*farleft);
    }
    return;
  }
  int32_t divider = n >> 1;
  otri innerleft, innerright;
  d_recurse(m, arr, divider, 1 - axis, farleft, &innerleft);
  d_recurse(m, arr + divider, n - divider, 1 - axis, &innerright, farright);
  d_mergehulls(m, farleft, &innerleft, &innerright, farright, axis);
}

/* returns the number of triangles; tris gets vertex ids (input order numbering, "z") */
static int32_t delaunay_int(const int32_t *x, const int32_t *y, int32_t n, int32_t *tris, int32_t cap) {
  if (n < 2) return 0;
  dmesh m;
  m.x = x;
  m.y = y;
  m.cap = 4 * n + 16;
  m.nb = (int32_t *)malloc((size_t)m.cap * 3 * sizeof(int32_t));
  m.vx = (int32_t *)malloc((size_t)m.cap * 3 * sizeof(int32_t));
  m.ntri = 0;
  m.seed = 1;
  int32_t *arr = (int32_t *)malloc((size_t)n * sizeof(int32_t));
  for (int32_t i = 0; i < n; i++) arr[i] = i;
  d_vertexsort(&m, arr, n);
  int32_t i = 0; /* drop duplicates, keep the first in sorted order (:6183-6197) */
  for (int32_t j = 1; j < n; j++)
    if (!(x[arr[i]] == x[arr[j]] && y[arr[i]] == y[arr[j]])) arr[++i] = arr[j];
  i++;
  int32_t nt = 0;
  if (i >= 2) {
    int32_t divider = i >> 1;
    if (i - divider >= 2) {
      if (divider >= 2) d_alternateaxes(&m, arr, divider, 1);
      d_alternateaxes(&m, arr + divider, i - divider, 1);
    }
    otri hl, hr;
    d_recurse(&m, arr, i, 0, &hl, &hr);
    for (int32_t t = 0; t < m.ntri; t++) {
      const int32_t *v = &m.vx[t * 3];
      if (v[0] >= 0 && v[1] >= 0 && v[2] >= 0) {
        if (nt < cap) { /* corner order of writeelements: org, dest, apex of orient 0 */
          tris[nt * 3 + 0] = v[1];
          tris[nt * 3 + 1] = v[2];
          tris[nt * 3 + 2] = v[0];
        }
        nt++;
      }
    }
  }
  free(arr);
  free(m.nb);
  free(m.vx);
  return nt;
}

int32_t vo_delaunay(const float *pts, int32_t n, int32_t *tris, int32_t cap) {
  int32_t *x = (int32_t *)malloc((size_t)(n > 0 ? n : 1) * sizeof(int32_t));
  int32_t *y = (int32_t *)malloc((size_t)(n > 0 ? n : 1) * sizeof(int32_t));
  for (int32_t i = 0; i < n; i++) {
    x[i] = (int32_t)pts[2 * i];
    y[i] = (int32_t)pts[2 * i + 1];
  }
  int32_t nt = delaunay_int(x, y, n, tris, cap);
  free(x);
  free(y);
  return nt;
}

/* ------------------------------------------------------------------------- */
/* O1  removeOutliers, viso/matcher.cpp:1207-1377                              */
/* ------------------------------------------------------------------------- */
static int32_t remove_outliers(const vo_params *p, vo_match *pm, int32_t n, int32_t method) {
  if (n <= 3) return n;
  float *pts = (float *)malloc((size_t)n * 2 * sizeof(float));
  for (int32_t i = 0; i < n; i++) {
    pts[2 * i] = pm[i].u1c;
    pts[2 * i + 1] = pm[i].v1c;
  }
  int32_t cap = 2 * n + 16;
  int32_t *tris = (int32_t *)malloc((size_t)cap * 3 * sizeof(int32_t));
  int32_t nt = vo_delaunay(pts, n, tris, cap);
  int32_t *support = (int32_t *)calloc((size_t)n, sizeof(int32_t));
  float ftol = (float)p->outlier_flow_tolerance, dtol = (float)p->outlier_disp_tolerance;
  for (int32_t t = 0; t < nt; t++) {
    int32_t q[3] = {tris[3 * t], tris[3 * t + 1], tris[3 * t + 2]};
    float fu[3], fv[3], dp[3];
    for (int k = 0; k < 3; k++) {
      const vo_match *a = &pm[q[k]];
      if (method == 0) {
        fu[k] = a->u1c - a->u1p;
        fv[k] = a->v1c - a->v1p;
        dp[k] = 0;
      } else if (method == 1) {
        fu[k] = fv[k] = 0;
        dp[k] = a->u1c - a->u2c;
      } else {
        fu[k] = a->u1c - a->u1p;
        fv[k] = a->v1c - a->v1p;
        dp[k] = a->u1p - a->u2p;
      }
    }
    static const int E[3][2] = {{0, 1}, {1, 2}, {0, 2}};
    for (int e = 0; e < 3; e++) {
      int a = E[e][0], b = E[e][1], ok;
      if (method == 0)
        ok = fabsf(fu[a] - fu[b]) + fabsf(fv[a] - fv[b]) < ftol;
      else if (method == 1)
        ok = fabsf(dp[a] - dp[b]) < dtol;
      else
        ok = fabsf(dp[a] - dp[b]) < dtol && fabsf(fu[a] - fu[b]) + fabsf(fv[a] - fv[b]) < ftol;
      if (ok) {
        support[q[a]]++;
        support[q[b]]++;
      }
    }
  }
  int32_t k = 0;
  for (int32_t i = 0; i < n; i++)
    if (support[i] >= 4) pm[k++] = pm[i];
  free(pts);
  free(tris);
  free(support);
  return k;
}

int32_t vo_remove_outliers(const vo_params *p, vo_match *m, int32_t n, int32_t method) {
  return remove_outliers(p, m, n, method);
}

/* ------------------------------------------------------------------------- */
/* R1  refinement / relocateMinimum / parabolicFitting, :1379-1585             */
/* ------------------------------------------------------------------------- */
static void relocate_minimum(const uint8_t *du1, const uint8_t *dv1, const int32_t *dims1, const uint8_t *du2,
                             const uint8_t *dv2, const int32_t *dims2, float u1, float v1, float *u2, float *v2) {
  if (*u2 - 2 < VO_MARGIN || *u2 + 2 > dims2[0] - 1 - VO_MARGIN || *v2 - 2 < VO_MARGIN ||
      *v2 + 2 > dims2[1] - 1 - VO_MARGIN)
    return;
  uint8_t ref[16], cur[16];
  small_descriptor(du1, dv1, dims1[2], (int32_t)u1, (int32_t)v1, ref);
  int32_t min_ind = 0, min_cost = 0;
  for (int32_t dv = 0; dv < 5; dv++)
    for (int32_t du = 0; du < 5; du++) {
      small_descriptor(du2, dv2, dims2[2], (int32_t)*u2 + du - 2, (int32_t)*v2 + dv - 2, cur);
      int32_t c = sad_bytes(ref, cur, 16);
      if ((dv == 0 && du == 0) || c < min_cost) {
        min_ind = dv * 5 + du;
        min_cost = c;
      }
    }
  *u2 += (float)(min_ind % 5) - 2.0;
  *v2 += (float)(min_ind / 5) - 2.0;
}

/* Matrix::solve (Gauss-Jordan, viso/matrix.cpp:424-519) specialised to 6x6, one rhs */
static int gaussj6(double A[6][6], double B[6]) {
  int indxc[6], indxr[6], ipiv[6] = {0, 0, 0, 0, 0, 0};
  int icol = 0, irow = 0;
  for (int i = 0; i < 6; i++) {
    double big = 0.0;
    for (int j = 0; j < 6; j++)
      if (ipiv[j] != 1)
        for (int k = 0; k < 6; k++)
          if (ipiv[k] == 0)
            if (fabs(A[j][k]) >= big) {
              big = fabs(A[j][k]);
              irow = j;
              icol = k;
            }
    ++ipiv[icol];
    if (irow != icol) {
      for (int l = 0; l < 6; l++) {
        double t = A[irow][l];
        A[irow][l] = A[icol][l];
        A[icol][l] = t;
      }
      double t = B[irow];
      B[irow] = B[icol];
      B[icol] = t;
    }
    indxr[i] = irow;
    indxc[i] = icol;
    if (fabs(A[icol][icol]) < 1e-20) return 0;
    double pivinv = 1.0 / A[icol][icol];
    A[icol][icol] = 1.0;
    for (int l = 0; l < 6; l++) A[icol][l] *= pivinv;
    B[icol] *= pivinv;
    for (int ll = 0; ll < 6; ll++)
      if (ll != icol) {
        double dum = A[ll][icol];
        A[ll][icol] = 0.0;
        for (int l = 0; l < 6; l++) A[ll][l] -= A[icol][l] * dum;
        B[ll] -= B[icol] * dum;
      }
  }
  (void)indxr;
  (void)indxc; /* the column unscramble only touches A, which is discarded */
  return 1;
}

static const double PF_A[9][6] = {{1, 1, 1, -1, -1, 1}, {0, 1, 0, 0, -1, 1}, {1, 1, -1, 1, -1, 1},
                                  {1, 0, 0, -1, 0, 1},  {0, 0, 0, 0, 0, 1},  {1, 0, 0, 1, 0, 1},
                                  {1, 1, -1, -1, 1, 1}, {0, 1, 0, 0, 1, 1},  {1, 1, 1, 1, 1, 1}}; /* :1508-1516 */

static int parabolic_fitting(const uint8_t *du1, const uint8_t *dv1, const int32_t *dims1, const uint8_t *du2,
                             const uint8_t *dv2, const int32_t *dims2, float u1, float v1, float *u2, float *v2) {
  if (*u2 - 3 < VO_MARGIN || *u2 + 3 > dims2[0] - 1 - VO_MARGIN || *v2 - 3 < VO_MARGIN ||
      *v2 + 3 > dims2[1] - 1 - VO_MARGIN)
    return 0;
  uint8_t ref[16], cur[16];
  small_descriptor(du1, dv1, dims1[2], (int32_t)u1, (int32_t)v1, ref);
  int32_t cost[49];
  for (int32_t dv = 0; dv < 7; dv++)
    for (int32_t du = 0; du < 7; du++) {
      small_descriptor(du2, dv2, dims2[2], (int32_t)*u2 + du - 3, (int32_t)*v2 + dv - 3, cur);
      cost[dv * 7 + du] = sad_bytes(ref, cur, 16);
    }
  int32_t min_ind = 0, min_cost = cost[0];
  for (int32_t i = 1; i < 49; i++)
    if (cost[i] < min_cost) {
      min_ind = i;
      min_cost = cost[i];
    }
  int32_t du = min_ind % 7, dv = min_ind / 7;
  if (du == 0 || du == 6 || dv == 0 || dv == 6) return 0;
  double c[9];
  for (int32_t i = -1; i <= 1; i++)
    for (int32_t j = -1; j <= 1; j++) c[(i + 1) * 3 + (j + 1)] = cost[(dv + i) * 7 + (du + j)];
  /* b = At*c ; AtA = At*A   (Matrix::operator*, viso/matrix.cpp, k-innermost accumulation from 0) */
  double b[6], AtA[6][6];
  for (int i = 0; i < 6; i++) {
    double s = 0;
    for (int k = 0; k < 9; k++) s += PF_A[k][i] * c[k];
    b[i] = s;
    for (int j = 0; j < 6; j++) {
      double t = 0;
      for (int k = 0; k < 9; k++) t += PF_A[k][i] * PF_A[k][j];
      AtA[i][j] = t;
    }
  }
  if (!gaussj6(AtA, b)) return 0;
  float divisor = (float)(b[2] * b[2] - 4.0 * b[0] * b[1]);
  if (fabsf(divisor) < 1e-8 || fabs(b[2]) < 1e-8) return 0;
  float ddv = (float)((2.0 * b[0] * b[4] - b[2] * b[3]) / divisor);
  float ddu = (float)(-(b[4] + 2.0 * b[1] * ddv) / b[2]);
  if (fabsf(ddu) >= 1.0 || fabsf(ddv) >= 1.0) return 0;
  *u2 = (float)(*u2 + ((float)du - 3.0 + ddu));
  *v2 = (float)(*v2 + ((float)dv - 3.0 + ddv));
  return 1;
}

static int32_t refinement(vo_matcher *M, vo_match *pm, int32_t n, int32_t method) { /* :1498-1585 */
  const vo_params *p = &M->param;
  const uint8_t *du[4], *dv[4];
  for (int k = 0; k < 4; k++) {
    du[k] = p->half_resolution ? M->img[k].du_full : M->img[k].du;
    dv[k] = p->half_resolution ? M->img[k].dv_full : M->img[k].dv;
  }
  int32_t out = 0;
  for (int32_t q = 0; q < n; q++) {
    vo_match it = pm[q];
    int ok = 1;
    /* pairs: (target image, target dims, &u, &v) in the reference's order 1p, 2c, 2p */
    for (int step = 0; step < 3 && ok; step++) {
      int tgt;
      float *tu, *tv;
      const int32_t *tdims;
      if (step == 0) {
        if (!(method == 0 || method == 2)) continue;
        tgt = 0; tu = &it.u1p; tv = &it.v1p; tdims = M->dims_p;
      } else if (step == 1) {
        if (!(method == 1 || method == 2)) continue;
        tgt = 3; tu = &it.u2c; tv = &it.v2c; tdims = M->dims_c;
      } else {
        if (method != 2) continue;
        tgt = 1; tu = &it.u2p; tv = &it.v2p; tdims = M->dims_p;
      }
      if (p->refinement == 2)
        ok = parabolic_fitting(du[2], dv[2], M->dims_c, du[tgt], dv[tgt], tdims, it.u1c, it.v1c, tu, tv);
      else
        relocate_minimum(du[2], dv[2], M->dims_c, du[tgt], dv[tgt], tdims, it.u1c, it.v1c, tu, tv);
    }
    if (ok) pm[out++] = it;
  }
  return out;
}

/* ------------------------------------------------------------------------- */
/* M0  matchFeatures, viso/matcher.cpp:183-241                                 */
/* ------------------------------------------------------------------------- */
static void stage_store(vo_matcher *M, int s, const vo_match *a, int32_t n) {
  free(M->stage[s]);
  M->stage[s] = (vo_match *)malloc((size_t)(n > 0 ? n : 1) * sizeof(vo_match));
  if (n > 0) memcpy(M->stage[s], a, (size_t)n * sizeof(vo_match));
  M->stage_n[s] = n;
}

int32_t vo_match_features(vo_matcher *M, int32_t method, const double *Tr) {
  const vo_params *p = &M->param;
  const vo_image *i1p = &M->img[0], *i2p = &M->img[1], *i1c = &M->img[2], *i2c = &M->img[3];
  if (method == 0) {
    if (i1p->n2 == 0 || i1c->n2 == 0) return 0;
    if (p->multi_stage && (i1p->n1 == 0 || i1c->n1 == 0)) return 0;
  } else if (method == 1) {
    if (i1c->n2 == 0 || i2c->n2 == 0) return 0;
    if (p->multi_stage && (i1c->n1 == 0 || i2c->n1 == 0)) return 0;
  } else {
    if (i1p->n2 == 0 || i2p->n2 == 0 || i1c->n2 == 0 || i2c->n2 == 0) return 0;
    if (p->multi_stage && (i1p->n1 == 0 || i2p->n1 == 0 || i1c->n1 == 0 || i2c->n1 == 0)) return 0;
  }
  clear_stages(M);
  memset(M->counters, 0, sizeof(M->counters));
  vo_matchvec pm1 = {0, 0, 0}, pm2 = {0, 0, 0};
  if (p->multi_stage) {
    matching(M, 1, &pm1, method, 0, Tr);
    stage_store(M, 0, pm1.a, pm1.n);
    M->counters[5] = M->counters[0];
    M->counters[6] = M->counters[1];
    M->counters[7] = M->counters[2];
    pm1.n = remove_outliers(p, pm1.a, pm1.n, method);
    stage_store(M, 1, pm1.a, pm1.n);
    prior_statistics(M, pm1.a, pm1.n, method);
    matching(M, 0, &pm2, method, 1, Tr);
  } else {
    matching(M, 0, &pm2, method, 0, Tr);
  }
  stage_store(M, 2, pm2.a, pm2.n);
  M->counters[3] = pm2.n;
  if (p->refinement > 0) pm2.n = refinement(M, pm2.a, pm2.n, method);
  stage_store(M, 3, pm2.a, pm2.n);
  pm2.n = remove_outliers(p, pm2.a, pm2.n, method);
  stage_store(M, 4, pm2.a, pm2.n);
  M->counters[4] = pm2.n;
  free(M->matched);
  M->matched = pm2.a;
  M->n_matched = pm2.n;
  free(pm1.a);
  return 1;
}

int32_t vo_num_matches(const vo_matcher *m) { return m->n_matched; }
void vo_get_matches(const vo_matcher *m, vo_match *out) {
  if (m->n_matched) memcpy(out, m->matched, (size_t)m->n_matched * sizeof(vo_match));
}
int32_t vo_stage_size(const vo_matcher *m, int32_t s) { return m->stage_n[s]; }
void vo_stage_get(const vo_matcher *m, int32_t s, vo_match *out) {
  if (m->stage_n[s]) memcpy(out, m->stage[s], (size_t)m->stage_n[s] * sizeof(vo_match));
}
int32_t vo_num_ranges(const vo_matcher *m) { return m->n_ranges; }
void vo_get_ranges(const vo_matcher *m, vo_range *out) {
  if (m->n_ranges) memcpy(out, m->ranges, (size_t)m->n_ranges * sizeof(vo_range));
}
void vo_get_counters(const vo_matcher *m, int64_t *out8) { memcpy(out8, m->counters, sizeof(m->counters)); }

int32_t vo_num_features(const vo_matcher *m, int32_t which) {
  static const int IMG[8] = {0, 1, 2, 3, 0, 1, 2, 3};
  const vo_image *im = &m->img[IMG[which]];
  return which < 4 ? im->n1 : im->n2;
}
void vo_get_features(const vo_matcher *m, int32_t which, int32_t *out) {
  static const int IMG[8] = {0, 1, 2, 3, 0, 1, 2, 3};
  const vo_image *im = &m->img[IMG[which]];
  int32_t n = which < 4 ? im->n1 : im->n2;
  const int32_t *src = which < 4 ? im->m1 : im->m2;
  if (n) memcpy(out, src, (size_t)n * 12 * sizeof(int32_t));
}

int32_t vo_get_gradients(const vo_matcher *m, int32_t which, int32_t full, uint8_t *du, uint8_t *dv) {
  const vo_image *im = &m->img[which];
  const int32_t *dims = which < 2 ? m->dims_p : m->dims_c;
  const uint8_t *pu = full ? im->du_full : im->du, *pv = full ? im->dv_full : im->dv;
  if (!pu || !pv) return 0;
  int32_t bpl = dims[2], h = dims[1];
  if (!full && m->param.half_resolution) {
    bpl = vo_bpl16(dims[0] / 2);
    h = dims[1] / 2;
  }
  memcpy(du, pu, (size_t)bpl * h);
  memcpy(dv, pv, (size_t)bpl * h);
  return bpl * h;
}

/* ------------------------------------------------------------------------- */
/* B1  bucketFeatures, viso/matcher.cpp:243-284.  std::random_shuffle of        */
/*     libstdc++ (bits/stl_algo.h): for i in 1..n-1 swap(a[i], a[rand()%(i+1)]) */
/*     driven by the C library rand() exactly like the reference.              */
/* ------------------------------------------------------------------------- */
void vo_bucket_features(vo_matcher *M, int32_t max_features, float bucket_width, float bucket_height) {
  int32_t n = M->n_matched;
  float u_max = 0, v_max = 0;
  for (int32_t i = 0; i < n; i++) {
    if (M->matched[i].u1c > u_max) u_max = M->matched[i].u1c;
    if (M->matched[i].v1c > v_max) v_max = M->matched[i].v1c;
  }
  int32_t cols = (int32_t)floorf(u_max / bucket_width) + 1, rows = (int32_t)floorf(v_max / bucket_height) + 1;
  int32_t nb = cols * rows;
  int32_t *start = (int32_t *)calloc((size_t)nb + 1, sizeof(int32_t));
  int32_t *bin = (int32_t *)malloc((size_t)(n > 0 ? n : 1) * sizeof(int32_t));
  for (int32_t i = 0; i < n; i++) {
    int32_t u = (int32_t)floorf(M->matched[i].u1c / bucket_width), v = (int32_t)floorf(M->matched[i].v1c / bucket_height);
    bin[i] = v * cols + u;
    start[bin[i] + 1]++;
  }
  for (int32_t k = 0; k < nb; k++) start[k + 1] += start[k];
  vo_match *sorted = (vo_match *)malloc((size_t)(n > 0 ? n : 1) * sizeof(vo_match));
  int32_t *cur = (int32_t *)malloc((size_t)(nb > 0 ? nb : 1) * sizeof(int32_t));
  memcpy(cur, start, (size_t)nb * sizeof(int32_t));
  for (int32_t i = 0; i < n; i++) sorted[cur[bin[i]]++] = M->matched[i];
  int32_t out = 0;
  for (int32_t b = 0; b < nb; b++) {
    vo_match *a = sorted + start[b];
    int32_t cnt = start[b + 1] - start[b];
    for (int32_t i = 1; i < cnt; i++) {
      int32_t j = rand() % (i + 1);
      if (i != j) {
        vo_match t = a[i];
        a[i] = a[j];
        a[j] = t;
      }
    }
    for (int32_t k = 0; k < cnt && k < max_features; k++) M->matched[out++] = a[k];
  }
  M->n_matched = out;
  free(start);
  free(bin);
  free(sorted);
  free(cur);
}

/* getGain, viso/matcher.cpp:286-324 (keeps the reference's use of dims_p for both windows) */
static float window_mean(const uint8_t *I, int32_t bpl, int32_t u0, int32_t u1, int32_t v0, int32_t v1) {
  float mean = 0;
  for (int32_t v = v0; v <= v1; v++)
    for (int32_t u = u0; u <= u1; u++) mean += (float)I[v * bpl + u];
  return mean / (float)((u1 - u0 + 1) * (v1 - v0 + 1));
}

float vo_get_gain(const vo_matcher *M, const int32_t *inliers, int32_t n) {
  if (!M->img[0].I || !M->img[2].I || M->n_matched == 0 || n == 0) return 1;
  const int32_t ws = 3;
  float gain = 0;
  int32_t num = 0;
  for (int32_t k = 0; k < n; k++) {
    int32_t i = inliers[k];
    if (i < M->n_matched) {
      const vo_match *q = &M->matched[i];
      int32_t u0 = imin(imax((int32_t)q->u1p - ws, 0), M->dims_p[0]), u1 = imin(imax((int32_t)q->u1p + ws, 0), M->dims_p[0]);
      int32_t v0 = imin(imax((int32_t)q->v1p - ws, 0), M->dims_p[1]), v1 = imin(imax((int32_t)q->v1p + ws, 0), M->dims_p[1]);
      float mp = window_mean(M->img[0].I, M->dims_p[2], u0, u1, v0, v1);
      u0 = imin(imax((int32_t)q->u1c - ws, 0), M->dims_p[0]);
      u1 = imin(imax((int32_t)q->u1c + ws, 0), M->dims_p[0]);
      v0 = imin(imax((int32_t)q->v1c - ws, 0), M->dims_p[1]);
      v1 = imin(imax((int32_t)q->v1c + ws, 0), M->dims_p[1]);
      float mc = window_mean(M->img[2].I, M->dims_c[2], u0, u1, v0, v1);
      if (mp > 10) {
        gain += mc / mp;
        num++;
      }
    }
  }
  return num > 0 ? gain / (float)num : 1;
}
