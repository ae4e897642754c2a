// CPU emulation of the fused matching-resolution kernels k_feat_dense / k_feat_sparse (vsm_kernels.hip): the very
// per-thread functions of csrc/vsm_feat.h, walked tile by tile and thread by thread with the kernels' phase structure
// (fill | patches | suppression; the sparse tile one plane at a time).  Test infrastructure only: `tests/test_feat_emu.py`
// compares the planes and survivors with the oracle's filters and nonMaximumSuppression.
#include <stdint.h>
#include <string.h>

#include <vector>

#include "vsm_feat.h"

static int cdiv(int a, int b) { return (a + b - 1) / b; }

extern "C" {

// geometry the launcher uses (kept in one place for kernels, launcher and emulation: vsm_feat_tiles_* below mirror it)
void emu_feat_tiles(int mbpl, int mh, int ncu_d, int ncv_d, int ncu_s, int ncv_s, int32_t *out) {
  out[0] = cdiv(mbpl + 8, 128) > cdiv(ncu_d + 3, 32) ? cdiv(mbpl + 8, 128) : cdiv(ncu_d + 3, 32);
  out[1] = cdiv(mh + 4, 48) > cdiv(ncv_d + 2, 12) ? cdiv(mh + 4, 48) : cdiv(ncv_d + 2, 12);
  out[2] = cdiv(ncu_s, 16);
  out[3] = cdiv(ncv_s, 4);
}

void emu_feat_dense(const uint8_t *img, int mw, int mh, int mbpl, int tau, int ncu, int ncv, uint8_t *du, uint8_t *dv,
                    int16_t *f1, int16_t *f2, int32_t *cand) {
  typedef VfDense G;
  int32_t tl[4];
  emu_feat_tiles(mbpl, mh, ncu, ncv, 0, 0, tl);
  alignas(16) static uint32_t s_img[G::IH * G::IWD];   // (16-byte stores, like the kernels' LDS arrays)
  alignas(16) static int16_t s_f[2 * G::FH * G::FS];
  for (int ty = 0; ty < tl[1]; ty++)
    for (int tx = 0; tx < tl[0]; tx++) {
      // poison: nothing may depend on what a previous tile left behind
      memset(s_img, 0xa5, sizeof(s_img));
      memset(s_f, 0x5a, sizeof(s_f));
      for (int t = 0; t < 256; t++) vf_fill<G, 256>(s_img, img, mbpl * mh, mbpl, tx, ty, t);
      for (int t = 0; t < G::PC * G::PR; t++) vf_dense_patch(s_img, s_f, t, tx, ty, mbpl, mh, du, dv, f1, f2);
      for (int it = 0; it < G::CU * G::CV * 2; it++) vf_dense_nms(s_f, it, tx, ty, mw, mh, 6, tau, ncu, ncv, cand);
    }
}

void emu_feat_sparse(const uint8_t *img, int mw, int mh, int mbpl, int tau, int ncu, int ncv, int32_t *cand) {
  typedef VfSparse G;
  int32_t tl[4];
  emu_feat_tiles(mbpl, mh, 0, 0, ncu, ncv, tl);
  alignas(16) static uint32_t s_img[G::IH * G::IWD];
  alignas(16) static int16_t s_f[G::FH * G::FS];
  std::vector<VfSparseKeep> keep(G::PC * G::PR);
  for (int ty = 0; ty < tl[3]; ty++)
    for (int tx = 0; tx < tl[2]; tx++) {
      memset(s_img, 0xa5, sizeof(s_img));
      memset(s_f, 0x5a, sizeof(s_f));
      for (int t = 0; t < 256; t++) vf_fill<G, 256>(s_img, img, mbpl * mh, mbpl, tx, ty, t);
      for (int t = 0; t < G::PC * G::PR; t++) vf_sparse_patch(s_img, s_f, t, keep[t]);
      for (int it = 0; it < G::CU * G::CV; it++) vf_sparse_nms<1>(s_f, it, 0, 0, tx, ty, mw, mh, 6, tau, ncu, ncv, cand);
      for (int t = 0; t < G::PC * G::PR; t++) vf_sparse_store_f2(s_f, t, keep[t]);
      for (int it = 0; it < G::CU * G::CV; it++) vf_sparse_nms<1>(s_f, it, 0, 1, tx, ty, mw, mh, 6, tau, ncu, ncv, cand);
    }
}
}
