// Fused matching-resolution image side (filters + non-maximum suppression out of one LDS tile), shared by the gfx950
// kernels (vsm_kernels.hip: k_feat_dense, k_feat_sparse) and by the CPU emulation the "not gpu" tests run
// (tests/emu/feat_emu.cpp walks the same per-thread functions tile by tile, thread by thread).
//
// What is fused (file:line = the reference):
//   F1 filter::sobel5x5          viso/filter.cpp:316-324   du, dv of the matching-resolution image -> HBM (the descriptors read them)
//   F2 filter::blob5x5           viso/filter.cpp:343-365   f1 -> LDS only
//   F3 filter::checkerboard5x5   viso/filter.cpp:331-336   f2 -> LDS only
//   N1 nonMaximumSuppression     viso/matcher.cpp:330-431  both scales, from the LDS planes
// so that f1 / f2 (8 of the 19 bytes per matching-resolution pixel of the unfused kernels) never reach HBM.
//
// The filters run over the image as ONE byte stream (row stride bpl, pad bytes 0, positions outside the stream 0), exactly
// like the reference's SSE loops: an LDS tile is filled by stream position, so a tap that leaves a row at its end reads
// the neighbouring row's bytes as the reference does, whatever the tile.
//
// Arithmetic: 16-bit lanes, two pixels per instruction (v_pk_*).  Ranges (8-bit input): column sums of the binomial
// <= 4080, of the derivative +-765, row passes +-12240; box sums <= 6375; f1 in [-4080, 4080], f2 in [-2040, 2040]:
// everything fits int16, and (x >> 7) + 128 of a Sobel response lies in [32, 223] - the reference's unsigned saturation
// never acts on it.
#pragma once

#include <stdint.h>

#if defined(__HIPCC__)
#define VF_HD __host__ __device__ __forceinline__
#define VF_HDM __host__ __device__ __forceinline__
#else
#define VF_HD static inline __attribute__((always_inline))
#define VF_HDM inline __attribute__((always_inline))
#endif

typedef short vf_s2 __attribute__((ext_vector_type(2)));

VF_HD uint32_t vf_perm(uint32_t hi, uint32_t lo, uint32_t sel) {  // v_perm_b32: selector bytes 0-3 = lo, 4-7 = hi, 0x0c = 0
#if defined(__HIP_DEVICE_COMPILE__)
  return __builtin_amdgcn_perm(hi, lo, sel);
#else
  const uint64_t src = ((uint64_t)hi << 32) | lo;
  uint32_t r = 0;
  for (int k = 0; k < 4; k++) {
    const uint32_t s = (sel >> (8 * k)) & 0xffu;
    const uint32_t b = s <= 7u ? (uint32_t)((src >> (8 * s)) & 0xffu) : 0u;
    r |= b << (8 * k);
  }
  return r;
#endif
}
VF_HD uint32_t vf_alignbyte(uint32_t hi, uint32_t lo, uint32_t sh) {
#if defined(__HIP_DEVICE_COMPILE__)
  return __builtin_amdgcn_alignbyte(hi, lo, sh);
#else
  return (uint32_t)((((uint64_t)hi << 32) | lo) >> (8 * sh));
#endif
}
VF_HD uint32_t vf_bits(vf_s2 v) { return __builtin_bit_cast(uint32_t, v); }
VF_HD vf_s2 vf_pair(uint32_t v) { return __builtin_bit_cast(vf_s2, v); }
// (a.y, b.x): the pair one column further to the right of a, b = the next aligned pair
VF_HD vf_s2 vf_shift(vf_s2 a, vf_s2 b) { return vf_pair(vf_alignbyte(vf_bits(b), vf_bits(a), 2u)); }

// 16 bytes of the image stream at 16-byte aligned position p (n = bpl * h, a multiple of 16): zeros outside the stream
struct vf_u4 {
  uint32_t x, y, z, w;
};
VF_HD vf_u4 vf_stream16(const uint8_t *in, int p, int n) {
  vf_u4 r = {0u, 0u, 0u, 0u};
  if (p >= 0 && p < n) {
#if defined(__HIP_DEVICE_COMPILE__)
    typedef uint32_t u4v __attribute__((ext_vector_type(4)));
    const u4v v = *(const __attribute__((address_space(1))) u4v *)(in + p);
    r.x = v.x, r.y = v.y, r.z = v.z, r.w = v.w;
#else
    const uint32_t *q = (const uint32_t *)(in + p);
    r.x = q[0], r.y = q[1], r.z = q[2], r.w = q[3];
#endif
  }
  return r;
}

// 16 bytes to a 16-byte aligned address (ds_write_b128 on LDS)
VF_HD void vf_store16(void *p, uint32_t a, uint32_t b, uint32_t c, uint32_t d) {
  typedef uint32_t u4v __attribute__((ext_vector_type(4)));
  u4v v = {a, b, c, d};
  *(u4v *)p = v;
}

// ---------------------------------------------------------------------------------------
// One thread's patch: 8 columns x ROWS rows of the four filters from a window of ROWS + 4 rows x 16 stream bytes
// (W[r][q], dwords; pixel i of the patch is byte 4 + i of a window row, the patch's first row is window row 2).
// Byte pairs (2+2j, 3+2j), j = 0..5, become 16-bit lane pairs P[r][j]; the patch's four output pairs are j = 1..4.
// ---------------------------------------------------------------------------------------
template <int ROWS>
struct VfWindow {
  uint32_t w[ROWS + 4][4];
  VF_HDM vf_s2 pair(int r, int j) const {  // compile-time r, j after unrolling
    const uint32_t v = w[r][(j + 1) >> 1];
    return vf_pair(vf_perm(0u, v, (j & 1) ? 0x0c010c00u : 0x0c030c02u));
  }
  VF_HDM int byte(int r, int i) const { return (int)((w[r][i >> 2] >> (8 * (i & 3))) & 0xffu); }
};

// Sobel row pass of one output row from the binomial (S) and derivative (D) column sums of the six pairs:
// du = S (x) [1 2 0 -2 -1], dv = D (x) [1 4 6 4 1], then (x >> 7) + 128 as bytes: du[0] = pixels 0-3, du[1] = 4-7
VF_HD void vf_sobel_row(const vf_s2 (&S)[6], const vf_s2 (&D)[6], uint32_t (&du)[2], uint32_t (&dv)[2]) {
  vf_s2 qs[5], qd[5];
#pragma unroll
  for (int i = 0; i < 5; i++) {
    qs[i] = vf_shift(S[i], S[i + 1]);
    qd[i] = vf_shift(D[i], D[i + 1]);
  }
  const vf_s2 c128 = {128, 128};
  vf_s2 u[4], v[4];
#pragma unroll
  for (int m = 1; m <= 4; m++) {
    const vf_s2 hu = (S[m - 1] - S[m + 1]) + (qs[m - 1] - qs[m]) * (short)2;
    const vf_s2 hv = (D[m - 1] + D[m + 1]) + (qd[m - 1] + qd[m]) * (short)4 + D[m] * (short)6;
    u[m - 1] = (hu >> 7) + c128;
    v[m - 1] = (hv >> 7) + c128;
  }
  du[0] = vf_perm(vf_bits(u[1]), vf_bits(u[0]), 0x06040200u);
  du[1] = vf_perm(vf_bits(u[3]), vf_bits(u[2]), 0x06040200u);
  dv[0] = vf_perm(vf_bits(v[1]), vf_bits(v[0]), 0x06040200u);
  dv[1] = vf_perm(vf_bits(v[3]), vf_bits(v[2]), 0x06040200u);
}

// blob + corner row pass of one output row from the column sums of the six pairs: C3 = rows 1..3, C5 = rows 0..4,
// Cc = rows 0,1 minus rows 3,4; ctr[m-1] = the centre row's pixels of output pair m.
//   f1 = -box5 + 2 box3 + 7 centre,  f2 = Cc (x) [1 1 0 -1 -1]
VF_HD void vf_blob_row(const vf_s2 (&C3)[6], const vf_s2 (&C5)[6], const vf_s2 (&Cc)[6], const vf_s2 (&ctr)[4], vf_s2 (&f1)[4],
                       vf_s2 (&f2)[4]) {
  vf_s2 q3[5], q5[5], qc[5];
#pragma unroll
  for (int i = 0; i < 5; i++) {
    q3[i] = vf_shift(C3[i], C3[i + 1]);
    q5[i] = vf_shift(C5[i], C5[i + 1]);
    qc[i] = vf_shift(Cc[i], Cc[i + 1]);
  }
#pragma unroll
  for (int m = 1; m <= 4; m++) {
    const vf_s2 b3 = q3[m - 1] + C3[m] + q3[m];
    const vf_s2 b5 = (C5[m - 1] + C5[m + 1]) + (q5[m - 1] + q5[m]) + C5[m];
    f1[m - 1] = b3 * (short)2 + ctr[m - 1] * (short)7 - b5;
    f2[m - 1] = (Cc[m - 1] - Cc[m + 1]) + (qc[m - 1] - qc[m]);
  }
}

// Rows at the top / bottom of the image: the reference's column pass is zero at stream positions outside rows [2, h-3]
// ([lo, hi) = those rows' positions; both even, like the position g of a pair's first pixel, so a pair is in or out as
// a whole).  f0 = stream position of the patch row's first pixel; pair j starts at f0 - 2 + 2j.
VF_HD void vf_sobel_zero_outside(vf_s2 (&S)[6], vf_s2 (&D)[6], int f0, int lo, int hi) {
#pragma unroll
  for (int j = 0; j < 6; j++) {
    const int g = f0 - 2 + 2 * j;
    const uint32_t keep = (uint32_t)0 - (uint32_t)(g >= lo && g < hi);
    S[j] = vf_pair(vf_bits(S[j]) & keep);
    D[j] = vf_pair(vf_bits(D[j]) & keep);
  }
}

// column sums of one window column pair j for output row rr (window rows rr .. rr+4)
template <int ROWS>
VF_HD void vf_columns_sobel(const VfWindow<ROWS> &W, int rr, int j, vf_s2 &S, vf_s2 &D) {
  const vf_s2 a = W.pair(rr, j), b = W.pair(rr + 1, j), c = W.pair(rr + 2, j), dd = W.pair(rr + 3, j), e = W.pair(rr + 4, j);
  S = (a + e) + (b + dd) * (short)4 + c * (short)6;
  D = (a - e) + (b - dd) * (short)2;
}
// The same sums for ALL rows of a patch at once: [1 4 6 4 1] = four cascaded pair adds down the window's rows, [1 2 0 -2 -1]
// = the second stage ([1 2 1]) two rows up minus two rows down - the stages are shared between the patch's rows: 4 ROWS + 10
// packed operations per column pair where the row-by-row form takes 10 per row (same 16-bit lanes, no value beyond 16 * 255)
template <int ROWS>
VF_HD void vf_columns_sobel_all(const VfWindow<ROWS> &W, int j, vf_s2 (&S)[ROWS], vf_s2 (&D)[ROWS]) {
  vf_s2 r[ROWS + 4], p1[ROWS + 3], p2[ROWS + 2], p3[ROWS + 1];
#pragma unroll
  for (int i = 0; i < ROWS + 4; i++) r[i] = W.pair(i, j);
#pragma unroll
  for (int i = 0; i < ROWS + 3; i++) p1[i] = r[i] + r[i + 1];
#pragma unroll
  for (int i = 0; i < ROWS + 2; i++) p2[i] = p1[i] + p1[i + 1];
#pragma unroll
  for (int i = 0; i < ROWS + 1; i++) p3[i] = p2[i] + p2[i + 1];
#pragma unroll
  for (int rr = 0; rr < ROWS; rr++) {
    S[rr] = p3[rr] + p3[rr + 1];
    D[rr] = p2[rr] - p2[rr + 2];
  }
}
template <int ROWS>
VF_HD void vf_columns_blob(const VfWindow<ROWS> &W, int rr, int j, vf_s2 &C3, vf_s2 &C5, vf_s2 &Cc) {
  const vf_s2 a = W.pair(rr, j), b = W.pair(rr + 1, j), c = W.pair(rr + 2, j), dd = W.pair(rr + 3, j), e = W.pair(rr + 4, j);
  C3 = (b + dd) + c;
  C5 = C3 + (a + e);
  Cc = (a + b) - (dd + e);
}

// the masks of the reference's loops for a dumped response (blob: columns / rows 3 .. ; corner: 2 ..; both to bpl-3, h-3)
VF_HD int vf_f1_mask(int x, int y, int bpl, int h) { return x >= 3 && x <= bpl - 3 && y >= 3 && y <= h - 3; }
VF_HD int vf_f2_mask(int x, int y, int bpl, int h) { return x >= 2 && x <= bpl - 3 && y >= 2 && y <= h - 3; }

// ---------------------------------------------------------------------------------------
// Tile geometries.  A tile's LDS image covers stream columns [X0 + IX, X0 + IX + 4 * IWD) and rows [Y0 + IY, .. + IH) with
// (X0, Y0) = (tx * TW, ty * TH); its patch grid (= the f planes in LDS) starts at (X0 + FX, Y0 + FY), PC x PR patches of
// 8 x ROWS pixels.  Cells of the suppression grid: origin = N + margin + cell * (N + 1) in the image.
//   dense (N = 3):  a tile owns du / dv of the first 16 x 12 patches and the 32 x 12 cells whose origin has f-plane
//                   coordinates (5 + 4 c, 5 + 4 r): image cell (32 tx - 3 + c, 12 ty - 2 + r)
//   sparse (N = 9): a tile owns the 16 x 4 cells with f-plane origin (11 + 10 c, 9 + 10 r): image cell (16 tx + c, 4 ty + r)
// ---------------------------------------------------------------------------------------
struct VfDense {
  static constexpr int N = 3, ROWS = 4;
  static constexpr int TW = 128, TH = 48;
  static constexpr int PC = 17, PR = 14, PC_OWN = 16, PR_OWN = 12;
  static constexpr int IX = -16, IY = -6, IWD = 40, IH = 60;  // LDS image: 160 bytes x 60 rows
  static constexpr int FX = -8, FY = -4;
  static constexpr int FW = PC * 8, FH = PR * ROWS, FS = 136;  // f planes: 136 x 56 values; rows of 272 bytes: a patch row is one 16-byte store
  static constexpr int CU = 32, CV = 12, CELL_X0 = 5, CELL_Y0 = 5, CELL_U0 = -3, CELL_V0 = -2;
  static constexpr int WIN_DW = 1;  // first window dword of patch column 0 in an LDS image row
};
struct VfSparse {
  static constexpr int N = 9, ROWS = 6;
  static constexpr int TW = 160, TH = 40;
  static constexpr int PC = 23, PR = 10;
  static constexpr int IX = 0, IY = 4, IWD = 48, IH = 64;  // 192 bytes x 64 rows
  static constexpr int FX = 4, FY = 6;
  static constexpr int FW = PC * 8, FH = PR * ROWS, FS = 184;  // 184 x 60 values; rows of 368 bytes
  static constexpr int CU = 16, CV = 4, CELL_X0 = 11, CELL_Y0 = 9, CELL_U0 = 0, CELL_V0 = 0;
  static constexpr int WIN_DW = 0;
};

// LDS image fill: 16-byte pieces of the stream, thread t of NT; all of a thread's loads are requested before the first
// LDS store waits for one
template <class G, int NT>
VF_HD void vf_fill(uint32_t *s_img, const uint8_t *in, int n, int bpl, int tx, int ty, int t) {
  constexpr int per_row = G::IWD / 4, total = G::IH * per_row, iters = (total + NT - 1) / NT;
  vf_u4 v[iters];
#pragma unroll
  for (int i = 0; i < iters; i++) {
    const int e = t + i * NT, r = e / per_row, g = e - r * per_row;
    v[i] = vf_stream16(in, e < total ? (ty * G::TH + G::IY + r) * bpl + tx * G::TW + G::IX + 16 * g : -16, n);
  }
#pragma unroll
  for (int i = 0; i < iters; i++) {
    const int e = t + i * NT, r = e / per_row, g = e - r * per_row;
    if (e < total) vf_store16(s_img + r * G::IWD + 4 * g, v[i].x, v[i].y, v[i].z, v[i].w);
  }
}

template <class G>
VF_HD void vf_load_window(const uint32_t *s_img, int pc, int pr, VfWindow<G::ROWS> &W) {
#pragma unroll
  for (int r = 0; r < G::ROWS + 4; r++) {
    const uint32_t *row = s_img + (G::ROWS * pr + r) * G::IWD + 2 * pc + G::WIN_DW;
#pragma unroll
    for (int q = 0; q < 4; q++) W.w[r][q] = row[q];
  }
}

// ---------------------------------------------------------------------------------------
// N1: one (cell, filter) of the suppression grid from an LDS plane f (STR int16 per row).  The first-wins extrema of
// the cell in the reference's scan order (u outer, v inner, strict compare, viso/matcher.cpp:356-380) are the minima of the
// key (value, scan position); a candidate is the extremum of its own cell, so "no strictly better value in its
// (2N+1)^2 window outside the cell" (:383-428) is "the window extremum equals the candidate's value".  LANES lanes share
// an item (lane l8 owns cell columns / window rows l8, l8 + LANES, ...) and combine by width-LANES shuffles; fast = no
// window of this tile is cut by the clip limits (lim_i, lim_j) = (w-1-margin, h-1-margin) in plane coordinates.
// Returns the packed survivors (plane coordinates + (u0, v0); 0 = none).
// ---------------------------------------------------------------------------------------
#if defined(__HIP_DEVICE_COMPILE__)
#define VF_SHFL_XOR(v, o, w) __shfl_xor((v), (o), (w))
#else
#define VF_SHFL_XOR(v, o, w) (v)  // LANES = 1 on the host: never reached
#endif

template <int N, int LANES, int STR, bool fast>
VF_HD void vf_nms_item(const int16_t *f, int li, int lj, int l8, int lim_i, int lim_j, int tau, int u0, int v0, int32_t &cmin,
                       int32_t &cmax) {
  constexpr int N1 = N + 1, W = 2 * N + 1;
  static_assert((STR & 1) == 0, "plane rows must stay dword aligned");
  uint32_t kmin = 0xffffffffu, kmax = 0xffffffffu;
  {
    const int16_t *c0 = f + lj * STR + li;
#pragma unroll
    for (int t = 0; t < (N1 + LANES - 1) / LANES; t++) {
      const int di = l8 + t * LANES;  // this lane's column(s) of the cell
      if (di < N1) {
#pragma unroll
        for (int dj = 0; dj < N1; dj++) {
          const int val = c0[dj * STR + di];
          const uint32_t o = (uint32_t)(di * N1 + dj);
          const uint32_t a = ((uint32_t)(val + 32768) << 10) | o, b = ((uint32_t)(32767 - val) << 10) | o;
          kmin = a < kmin ? a : kmin;
          kmax = b < kmax ? b : kmax;
        }
      }
    }
  }
#pragma unroll
  for (int o = LANES / 2; o >= 1; o >>= 1) {
    const uint32_t a = (uint32_t)VF_SHFL_XOR((int)kmin, o, LANES), b = (uint32_t)VF_SHFL_XOR((int)kmax, o, LANES);
    kmin = a < kmin ? a : kmin;
    kmax = b < kmax ? b : kmax;
  }
  const int mnv = (int)(kmin >> 10) - 32768, mno = (int)(kmin & 1023u);
  const int mxv = 32767 - (int)(kmax >> 10), mxo = (int)(kmax & 1023u);
  const int mni = li + mno / N1, mnj = lj + mno % N1, mxi = li + mxo / N1, mxj = lj + mxo % N1;
  int wmn, wmx;  // extrema over this lane's share of the two windows
  {
    // A window row is W = 2N+1 values from column mi-N on: (W+1)/2 aligned dwords cover it whatever the parity of that
    // column, with one value too many - the last one (even start) or the first (odd start) - which is replaced by the
    // neutral element; then two values per v_pk_min_i16 / v_pk_max_i16.  In tiles the clip limits cut (fast = false),
    // columns beyond lim_i become neutral through one v_bfi per dword (the masks do not depend on the row) and rows
    // beyond lim_j are left out.
    constexpr int ND = (W + 1) / 2;
    const int cn = mni - N, cx = mxi - N;
    const bool pn1 = (cn & 1) != 0, px1 = (cx & 1) != 0;
    const uint32_t *pn = (const uint32_t *)(f + (mnj - N) * STR + (cn & ~1)), *px = (const uint32_t *)(f + (mxj - N) * STR + (cx & ~1));
    const int nvn = lim_i - (cn & ~1) + 1, nvx = lim_i - (cx & ~1) + 1;  // columns within the limit, counted from the first dword
    vf_s2 amn = {32767, 32767}, amx = {-32768, -32768};
#pragma unroll
    for (int t = 0; t < (W + LANES - 1) / LANES; t++) {
      const int r = l8 + t * LANES;
      if (r < W && (fast || (mnj - N + r <= lim_j))) {
        const uint32_t *rn = pn + r * (STR / 2);
#pragma unroll
        for (int c = 0; c < ND; c++) {
          uint32_t vn = rn[c];
          if (c == 0) vn = pn1 ? ((vn & 0xffff0000u) | 0x00007fffu) : vn;
          if (c == ND - 1) vn = pn1 ? vn : ((vn & 0x0000ffffu) | 0x7fff0000u);
          if (!fast) {
            const uint32_t keep = nvn >= 2 * c + 2 ? 0xffffffffu : (nvn == 2 * c + 1 ? 0x0000ffffu : 0u);
            vn = (vn & keep) | (0x7fff7fffu & ~keep);
          }
          amn = __builtin_elementwise_min(amn, vf_pair(vn));
        }
      }
      if (r < W && (fast || (mxj - N + r <= lim_j))) {
        const uint32_t *rx = px + r * (STR / 2);
#pragma unroll
        for (int c = 0; c < ND; c++) {
          uint32_t vx = rx[c];
          if (c == 0) vx = px1 ? ((vx & 0xffff0000u) | 0x00008000u) : vx;
          if (c == ND - 1) vx = px1 ? vx : ((vx & 0x0000ffffu) | 0x80000000u);
          if (!fast) {
            const uint32_t keep = nvx >= 2 * c + 2 ? 0xffffffffu : (nvx == 2 * c + 1 ? 0x0000ffffu : 0u);
            vx = (vx & keep) | (0x80008000u & ~keep);
          }
          amx = __builtin_elementwise_max(amx, vf_pair(vx));
        }
      }
    }
    wmn = (int)amn.x < (int)amn.y ? (int)amn.x : (int)amn.y;
    wmx = (int)amx.x > (int)amx.y ? (int)amx.x : (int)amx.y;
  }
#pragma unroll
  for (int o = LANES / 2; o >= 1; o >>= 1) {
    const int a = VF_SHFL_XOR(wmn, o, LANES), b = VF_SHFL_XOR(wmx, o, LANES);
    wmn = a < wmn ? a : wmn;
    wmx = b > wmx ? b : wmx;
  }
  const bool vmin = (mnv <= -tau) && wmn >= mnv, vmax = (mxv >= tau) && wmx <= mxv;
  cmin = vmin ? (int32_t)(0x80000000u | (uint32_t)(mni + u0) | ((uint32_t)(mnj + v0) << 14)) : 0;
  cmax = vmax ? (int32_t)(0x80000000u | (uint32_t)(mxi + u0) | ((uint32_t)(mxj + v0) << 14)) : 0;
}

// ---------------------------------------------------------------------------------------
// N1 for the dense scale (N = 3), one lane per (cell, filter), WITHOUT data-dependent LDS addresses: the lane reads the
// 10 x 10 values around its cell - rows lj-3 .. lj+6, columns li-3 .. li+6, five aligned dwords per row (li is odd) - which
// hold both candidates' windows wherever the candidates lie in the cell.  Neighbouring lanes read neighbouring 8-byte
// pieces, so the reads are wide and free of bank conflicts (windows addressed by the candidate's position cost three to
// four LDS cycles per read: the suppression was bound by them).  The cell scan works on those registers; a window is
// selected from them: region rows [dy, dy+6] = rows 3..6 always, a suffix of rows 0..2 and a prefix of rows 7..9 chosen by
// dy, columns [dx, dx+6] by four dword masks.  The clip limits only shorten the prefixes to the right / below:
// dxr = min(dx, lim_i - (li+3)), dyb = min(dy, lim_j - (lj+3)) (a cell itself never crosses the limits).
// ---------------------------------------------------------------------------------------
template <bool MAX>
VF_HD vf_s2 vf_mm(vf_s2 a, vf_s2 b) { return MAX ? __builtin_elementwise_max(a, b) : __builtin_elementwise_min(a, b); }
VF_HD uint32_t vf_bfi(uint32_t mask, uint32_t a, uint32_t b) { return (a & mask) | (b & ~mask); }  // v_bfi_b32

VF_HD uint32_t vf_all(bool c) { return (uint32_t)0 - (uint32_t)c; }  // all ones if c (selects are written as v_bfi on such masks:
                                                                       // left as ?: the compiler turns them into divergent branches)
template <bool MAX>
VF_HD int vf_dense_window(const uint32_t (&R)[10][5], int dx, int dy, int dxr, int dyb) {
  const uint32_t neutral = MAX ? 0x80008000u : 0x7fff7fffu;
  const uint32_t t0 = vf_all(dy == 0), t1 = vf_all(dy == 1), t2 = vf_all(dy == 2);
  const uint32_t b3 = vf_all(dyb >= 3), b2 = vf_all(dyb == 2), b1 = vf_all(dyb == 1);
  uint32_t col[5];
#pragma unroll
  for (int c = 0; c < 5; c++) {
    const vf_s2 s2 = vf_pair(R[2][c]), s1 = vf_mm<MAX>(vf_pair(R[1][c]), s2), s0 = vf_mm<MAX>(vf_pair(R[0][c]), s1);
    const vf_s2 p7 = vf_pair(R[7][c]), p8 = vf_mm<MAX>(p7, vf_pair(R[8][c])), p9 = vf_mm<MAX>(p8, vf_pair(R[9][c]));
    const vf_s2 mid = vf_mm<MAX>(vf_mm<MAX>(vf_pair(R[3][c]), vf_pair(R[4][c])), vf_mm<MAX>(vf_pair(R[5][c]), vf_pair(R[6][c])));
    const uint32_t top = vf_bfi(t0, vf_bits(s0), vf_bfi(t1, vf_bits(s1), vf_bfi(t2, vf_bits(s2), neutral)));
    const uint32_t bot = vf_bfi(b3, vf_bits(p9), vf_bfi(b2, vf_bits(p8), vf_bfi(b1, vf_bits(p7), neutral)));
    col[c] = vf_bits(vf_mm<MAX>(vf_mm<MAX>(vf_pair(top), vf_pair(bot)), mid));
  }
  // columns [dx, dxr + 6] of the region's ten: column 2c is the low half of dword c
  const uint32_t m0 = (vf_all(dx == 0) & 0x0000ffffu) | (vf_all(dx <= 1) & 0xffff0000u), m1 = vf_all(dx <= 2) | 0xffff0000u;
  const uint32_t m3 = vf_all(dxr >= 1) | 0x0000ffffu, m4 = (vf_all(dxr >= 2) & 0x0000ffffu) | (vf_all(dxr >= 3) & 0xffff0000u);
  vf_s2 w = vf_pair(col[2]);
  w = vf_mm<MAX>(w, vf_pair(vf_bfi(m0, col[0], neutral)));
  w = vf_mm<MAX>(w, vf_pair(vf_bfi(m1, col[1], neutral)));
  w = vf_mm<MAX>(w, vf_pair(vf_bfi(m3, col[3], neutral)));
  w = vf_mm<MAX>(w, vf_pair(vf_bfi(m4, col[4], neutral)));
  return MAX ? ((int)w.x > (int)w.y ? (int)w.x : (int)w.y) : ((int)w.x < (int)w.y ? (int)w.x : (int)w.y);
}

template <int STR>
VF_HD void vf_nms_dense_region(const int16_t *f, int li, int lj, int lim_i, int lim_j, int tau, int u0, int v0, int32_t &cmin,
                               int32_t &cmax) {
  static_assert((STR & 1) == 0, "plane rows must stay dword aligned");
  uint32_t R[10][5];
  {
    const uint32_t *base = (const uint32_t *)(f + (lj - 3) * STR + (li - 3));  // li - 3 is even
#pragma unroll
    for (int r = 0; r < 10; r++) {
#pragma unroll
      for (int c = 0; c < 5; c++) R[r][c] = base[r * (STR / 2) + c];
    }
  }
  // first-wins extrema of the cell (region rows / columns 3..6) in the reference's scan order (u outer, v inner, strict
  // compare, viso/matcher.cpp:356-380): minimum of the key (value, scan position)
  uint32_t kmin = 0xffffffffu, kmax = 0xffffffffu;
#pragma unroll
  for (int di = 0; di < 4; di++) {
#pragma unroll
    for (int dj = 0; dj < 4; dj++) {
      const int q = 3 + di;
      const uint32_t dw = R[3 + dj][q >> 1];
      const int val = (q & 1) ? ((int)dw >> 16) : (int)(short)(dw & 0xffffu);
      const uint32_t o = (uint32_t)(di * 4 + dj);
      const uint32_t a = ((uint32_t)(val + 32768) << 10) | o, b = ((uint32_t)(32767 - val) << 10) | o;
      kmin = a < kmin ? a : kmin;
      kmax = b < kmax ? b : kmax;
    }
  }
  const int mnv = (int)(kmin >> 10) - 32768, mno = (int)(kmin & 1023u);
  const int mxv = 32767 - (int)(kmax >> 10), mxo = (int)(kmax & 1023u);
  const int dxn = mno >> 2, dyn = mno & 3, dxx = mxo >> 2, dyx = mxo & 3;
  const int cx = lim_i - (li + 3), cy = lim_j - (lj + 3);  // columns / rows within the limits beyond the cell (>= 3: no cut)
  const int wmn = vf_dense_window<false>(R, dxn, dyn, dxn < cx ? dxn : cx, dyn < cy ? dyn : cy);
  const int wmx = vf_dense_window<true>(R, dxx, dyx, dxx < cx ? dxx : cx, dyx < cy ? dyx : cy);
  const bool vmin = (mnv <= -tau) && wmn >= mnv, vmax = (mxv >= tau) && wmx <= mxv;
  cmin = vmin ? (int32_t)(0x80000000u | (uint32_t)(li + dxn + u0) | ((uint32_t)(lj + dyn + v0) << 14)) : 0;
  cmax = vmax ? (int32_t)(0x80000000u | (uint32_t)(li + dxx + u0) | ((uint32_t)(lj + dyx + v0) << 14)) : 0;
}

// ---------------------------------------------------------------------------------------
// dense tile, thread t < PC * PR: its patch -> du / dv (HBM, owned patches), f1 / f2 (LDS planes; HBM too when
// dump_f1 != null: the debug getter of the filter responses)
// ---------------------------------------------------------------------------------------
VF_HD void vf_dense_patch(const uint32_t *s_img, int16_t *s_f, int t, int tx, int ty, int bpl, int h, uint8_t *du_plane,
                          uint8_t *dv_plane, int16_t *dump_f1, int16_t *dump_f2) {
  typedef VfDense G;
  const int pr = t / G::PC, pc = t - pr * G::PC;
  VfWindow<G::ROWS> W;
  vf_load_window<G>(s_img, pc, pr, W);
  const int x0 = tx * G::TW + G::FX + 8 * pc, y0 = ty * G::TH + G::FY + G::ROWS * pr;
  const bool own = pc < G::PC_OWN && pr < G::PR_OWN && x0 >= 0 && x0 < bpl;
  int16_t *f1 = s_f + (G::ROWS * pr) * G::FS + 8 * pc, *f2 = f1 + G::FH * G::FS;
#pragma unroll
  for (int rr = 0; rr < G::ROWS; rr++) {
    const int y = y0 + rr;
    vf_s2 C3[6], C5[6], Cc[6], ctr[4], o1[4], o2[4];
#pragma unroll
    for (int j = 0; j < 6; j++) vf_columns_blob<G::ROWS>(W, rr, j, C3[j], C5[j], Cc[j]);
#pragma unroll
    for (int m = 0; m < 4; m++) ctr[m] = W.pair(rr + 2, m + 1);
    vf_blob_row(C3, C5, Cc, ctr, o1, o2);
    vf_store16(f1 + rr * G::FS, vf_bits(o1[0]), vf_bits(o1[1]), vf_bits(o1[2]), vf_bits(o1[3]));
    vf_store16(f2 + rr * G::FS, vf_bits(o2[0]), vf_bits(o2[1]), vf_bits(o2[2]), vf_bits(o2[3]));
    if (own && y >= 0 && y < h) {
      uint32_t du[2], dv[2];
      vf_s2 S[6], D[6];
#pragma unroll
      for (int j = 0; j < 6; j++) vf_columns_sobel<G::ROWS>(W, rr, j, S[j], D[j]);
      if (y < 3 || y > h - 4) vf_sobel_zero_outside(S, D, y * bpl + x0, 2 * bpl, (h - 2) * bpl);
      vf_sobel_row(S, D, du, dv);
      uint32_t *pu = (uint32_t *)(du_plane + (size_t)y * bpl + x0), *pv = (uint32_t *)(dv_plane + (size_t)y * bpl + x0);
      pu[0] = du[0], pu[1] = du[1];
      pv[0] = dv[0], pv[1] = dv[1];
      if (dump_f1) {
#pragma unroll
        for (int m = 0; m < 4; m++) {
          const int x = x0 + 2 * m;
          dump_f1[(size_t)y * bpl + x] = vf_f1_mask(x, y, bpl, h) ? o1[m].x : (short)0;
          dump_f1[(size_t)y * bpl + x + 1] = vf_f1_mask(x + 1, y, bpl, h) ? o1[m].y : (short)0;
          dump_f2[(size_t)y * bpl + x] = vf_f2_mask(x, y, bpl, h) ? o2[m].x : (short)0;
          dump_f2[(size_t)y * bpl + x + 1] = vf_f2_mask(x + 1, y, bpl, h) ? o2[m].y : (short)0;
        }
      }
    }
  }
}

// dense tile, suppression item it < CU * CV * 2: lanes 0-31 of a wave = the 32 cells of a cell row in f1, lanes 32-63 the
// same in f2 (a group of 32 lanes reads 256 contiguous bytes per LDS instruction); survivors of image cell (ci, cj) -> cand
VF_HD void vf_dense_nms(const int16_t *s_f, int it, int tx, int ty, int mw, int mh, int margin, int tau, int ncu, int ncv,
                        int32_t *cand) {
  typedef VfDense G;
  static_assert(G::CU == 32, "a cell row per half wave");
  const int lcu = it & 31, k = (it >> 5) & 1, lcv = it >> 6;
  const int ci = G::CU * tx + G::CELL_U0 + lcu, cj = G::CV * ty + G::CELL_V0 + lcv;
  const int u0 = tx * G::TW + G::FX, v0 = ty * G::TH + G::FY;
  const int lim_i = mw - 1 - margin - u0, lim_j = mh - 1 - margin - v0;
  int32_t cmin, cmax;
  vf_nms_dense_region<G::FS>(s_f + k * G::FH * G::FS, G::CELL_X0 + (G::N + 1) * lcu, G::CELL_Y0 + (G::N + 1) * lcv, lim_i, lim_j, tau, u0,
                             v0, cmin, cmax);
  if (ci >= 0 && ci < ncu && cj >= 0 && cj < ncv) {
    int32_t *c = cand + (size_t)(ci * ncv + cj) * 4 + 2 * k;
    c[0] = cmin;
    c[1] = cmax;
  }
}

// ---------------------------------------------------------------------------------------
// sparse tile: one f plane in LDS at a time.  A thread's patch gives f1 (to LDS at once) and f2 (kept in registers
// until the first plane's suppression is through).
// ---------------------------------------------------------------------------------------
struct VfSparseKeep {
  uint32_t f2[VfSparse::ROWS][4];
};
VF_HD void vf_sparse_patch(const uint32_t *s_img, int16_t *s_f, int t, VfSparseKeep &keep) {
  typedef VfSparse G;
  const int pr = t / G::PC, pc = t - pr * G::PC;
  VfWindow<G::ROWS> W;
  vf_load_window<G>(s_img, pc, pr, W);
  int16_t *f1 = s_f + (G::ROWS * pr) * G::FS + 8 * pc;
#pragma unroll
  for (int rr = 0; rr < G::ROWS; rr++) {
    vf_s2 C3[6], C5[6], Cc[6], ctr[4], o1[4], o2[4];
#pragma unroll
    for (int j = 0; j < 6; j++) vf_columns_blob<G::ROWS>(W, rr, j, C3[j], C5[j], Cc[j]);
#pragma unroll
    for (int m = 0; m < 4; m++) ctr[m] = W.pair(rr + 2, m + 1);
    vf_blob_row(C3, C5, Cc, ctr, o1, o2);
    vf_store16(f1 + rr * G::FS, vf_bits(o1[0]), vf_bits(o1[1]), vf_bits(o1[2]), vf_bits(o1[3]));
#pragma unroll
    for (int m = 0; m < 4; m++) keep.f2[rr][m] = vf_bits(o2[m]);
  }
}
VF_HD void vf_sparse_store_f2(int16_t *s_f, int t, const VfSparseKeep &keep) {
  typedef VfSparse G;
  const int pr = t / G::PC, pc = t - pr * G::PC;
  int16_t *f = s_f + (G::ROWS * pr) * G::FS + 8 * pc;
#pragma unroll
  for (int rr = 0; rr < G::ROWS; rr++) {
    vf_store16(f + rr * G::FS, keep.f2[rr][0], keep.f2[rr][1], keep.f2[rr][2], keep.f2[rr][3]);
  }
}
// item it < CU * CV (x LANES lanes: l8), filter k: survivors of image cell (ci, cj)
template <int LANES>
VF_HD void vf_sparse_nms(const int16_t *s_f, int it, int l8, int k, int tx, int ty, int mw, int mh, int margin, int tau, int ncu,
                         int ncv, int32_t *cand) {
  typedef VfSparse G;
  const int lcv = it / G::CU, lcu = it - lcv * G::CU;
  const int ci = G::CU * tx + G::CELL_U0 + lcu, cj = G::CV * ty + G::CELL_V0 + lcv;
  const int u0 = tx * G::TW + G::FX, v0 = ty * G::TH + G::FY;
  const int lim_i = mw - 1 - margin - u0, lim_j = mh - 1 - margin - v0;
  const bool fast = lim_i >= G::FW - 1 && lim_j >= G::FH - 1;
  int32_t cmin, cmax;
  const int li = G::CELL_X0 + (G::N + 1) * lcu, lj = G::CELL_Y0 + (G::N + 1) * lcv;
  if (fast)  // (the same for all items of a tile)
    vf_nms_item<G::N, LANES, G::FS, true>(s_f, li, lj, l8, lim_i, lim_j, tau, u0, v0, cmin, cmax);
  else
    vf_nms_item<G::N, LANES, G::FS, false>(s_f, li, lj, l8, lim_i, lim_j, tau, u0, v0, cmin, cmax);
  if (l8 == 0 && ci < ncu && cj < ncv) {
    int32_t *c = cand + (size_t)(ci * ncv + cj) * 4 + 2 * k;
    c[0] = cmin;
    c[1] = cmax;
  }
}
