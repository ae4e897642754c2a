// gfx950 (MI355X, CDNA4) kernels of the matcher hot path.  Wave = 64 lanes.
//
// All kernels are integer/byte work bounded by HBM (or, for one 1242x375 pair, by launch and
// dependent-load latency): no MFMA anywhere.  Every kernel takes a z (or y) grid dimension
// over images / frame pairs so the same code serves the per-frame API (2 images, 1 pair) and
// batched sequences (hundreds of images per launch), which is what fills 256 CUs.
//
// Semantics are those of the reference (pad bytes pinned to 0, filters on the 1-D byte stream);
// file:line citations below refer to the reference repository.

#include <stdlib.h>

#include <algorithm>

#include "vsm_internal.h"
#include "vsm_feat.h"

#define WAVE 64

// XCD-aware block remap (MI355X: 8 XCDs, each with a private 4 MiB L2; hardware deals blocks
// round-robin over the XCDs).  Batched launches are flattened to 1-D and logical block
// L = (b % 8) * ceil(n/8) + b / 8, so every XCD walks one contiguous eighth of the (pair-major)
// work and the eight L2s stop fetching the same image lines.  Speed only, never correctness.
// Pointers that come out of the VsmImage / VsmSet tables are "generic" to the compiler, which then
// emits flat_load (address-space check, and every wait on LDS traffic also waits for them).  They
// all point into HBM: these helpers load through an explicit global-address-space pointer.
#define VSM_AS1 __attribute__((address_space(1)))
typedef uint32_t vsm_u4 __attribute__((ext_vector_type(4)));
typedef int32_t vsm_i4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ uint4 ldg_u4(const void *p) {
  const vsm_u4 v = *(const VSM_AS1 vsm_u4 *)p;
  return make_uint4(v.x, v.y, v.z, v.w);
}
__device__ __forceinline__ int4 ldg_i4(const void *p) {
  const vsm_i4 v = *(const VSM_AS1 vsm_i4 *)p;
  return make_int4(v.x, v.y, v.z, v.w);
}
// base + 32-bit byte offset: with a wave-uniform base the backend keeps the base in scalar registers and the offset in one
// vector register (global_load ... v_off, s[base]) instead of building a 64-bit address per lane
__device__ __forceinline__ uint4 ldg_u4_at(const void *base, uint32_t byte_off) {
  const vsm_u4 v = *(const VSM_AS1 vsm_u4 *)((const VSM_AS1 char *)base + byte_off);
  return make_uint4(v.x, v.y, v.z, v.w);
}
// 16 bytes at a dword-aligned (not 16-byte-aligned) offset: one global_load_dwordx4 all the same
typedef uint32_t vsm_u4_a4 __attribute__((ext_vector_type(4), aligned(4)));
__device__ __forceinline__ uint4 ldg_u4_at_dw(const void *base, uint32_t byte_off) {
  const vsm_u4_a4 v = *(const VSM_AS1 vsm_u4_a4 *)((const VSM_AS1 char *)base + byte_off);
  return make_uint4(v.x, v.y, v.z, v.w);
}
__device__ __forceinline__ uint32_t ldg_u32_at(const void *base, uint32_t byte_off) {
  return *(const VSM_AS1 uint32_t *)((const VSM_AS1 char *)base + byte_off);
}
__device__ __forceinline__ int32_t ldg_i32(const void *p) { return *(const VSM_AS1 int32_t *)p; }
__device__ __forceinline__ uint32_t ldg_u32(const void *p) { return *(const VSM_AS1 uint32_t *)p; }

__device__ __forceinline__ int xcd_remap(int b, int nblocks) {
  const int per = (nblocks + 7) >> 3;
  return (b & 7) * per + (b >> 3);
}

#ifdef VSM_FEAT_TIMING  // experiments (tools/build_variant.sh NAME -DVSM_FEAT_TIMING, tools/feat_timing.py): cycles per phase of every wave
__device__ unsigned int vsm_ft_rec[5][1 << 16][10];  // [kernel][wave] start (low bits), phase lengths ...
extern "C" int vsm_debug_feat_rec(unsigned int *out, int reset) {
  if (hipMemcpyFromSymbol(out, HIP_SYMBOL(vsm_ft_rec), sizeof(vsm_ft_rec)) != hipSuccess) return -1;
  (void)reset;
  return 0;
}
#define FT_DECL unsigned long long ft_t[10]; int ft_n = 0
#define FT_STAMP ft_t[ft_n++] = __builtin_amdgcn_s_memtime()
#define FT_FLUSH(kern)                                                              \
  do {                                                                              \
    const unsigned wv = ((blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x) * (blockDim.x >> 6) + (threadIdx.x >> 6); \
    if ((threadIdx.x & 63) == 0 && wv < (1u << 16)) {                               \
      vsm_ft_rec[kern][wv][0] = (unsigned)ft_t[0] | 1u;                             \
      for (int q = 1; q < ft_n; q++) vsm_ft_rec[kern][wv][q] = (unsigned)(ft_t[q] - ft_t[q - 1]); \
    }                                                                               \
  } while (0)
#else
#define FT_DECL
#define FT_STAMP
#define FT_FLUSH(base)
#endif

// ---------------------------------------------------------------------------------------
// ingest: caller image (row stride src_bpl) -> padded HBM copy [h][bpl], pad bytes = 0
// (Matcher::pushBack row copy, viso/matcher.cpp:163-175, with the pad pinned to 0)
// ---------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) k_ingest(const VsmImage *__restrict__ imgs, int first, const uint8_t *__restrict__ src0,
                                                const uint8_t *__restrict__ src1, size_t frame_stride, int src_bpl,
                                                int sides, int w, int h, int bpl) {
  // blockIdx.z = frame*sides + side; image id = first + 2*frame + side.  Threads are laid over the
  // padded image as one stream of dwords.  Caller rows need not be dword aligned (a 1242-pixel row
  // stride is not): every output dword comes from the two aligned source dwords around it and a byte
  // funnel shift; nothing past the last pixel of a row is touched.
  const int fr = blockIdx.z / sides, side = blockIdx.z - fr * sides;
  const uint8_t *__restrict__ src = (side ? src1 : src0) + (size_t)fr * frame_stride;
  uint8_t *__restrict__ dst = imgs[first + 2 * fr + side].img;
  const int per_row = bpl >> 2;
  const int gidx = blockIdx.x * blockDim.x + threadIdx.x;
  const int y = gidx / per_row, x4 = (gidx - y * per_row) * 4;
  if (y >= h) return;
  uint32_t v = 0;
  if (x4 < w) {
    const uintptr_t a = (uintptr_t)(src + (size_t)y * src_bpl + x4);
    const uint32_t sh = (uint32_t)(a & 3);
    const int nvalid = min(4, w - x4);  // bytes of this dword that belong to the row
    const uint32_t lo = ldg_u32((const void *)(a - sh));
    const uint32_t hi = ((int)(4 - sh) < nvalid) ? ldg_u32((const void *)(a - sh + 4)) : 0u;
    v = __builtin_amdgcn_alignbyte(hi, lo, sh);
    if (nvalid < 4) v &= (1u << (8 * nvalid)) - 1u;
  }
  *(VSM_AS1 uint32_t *)(dst + (size_t)y * bpl + x4) = v;
}

// ---------------------------------------------------------------------------------------
// F0 createHalfResolutionImage, viso/matcher.cpp:636-647: (a+b+c+d)/4 over 2x2, pad = 0
// ---------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) k_halve(const VsmImage *__restrict__ imgs, int first, VsmDims d) {
  const VsmImage &im = imgs[first + blockIdx.z];
  int x4 = (blockIdx.x * blockDim.x + threadIdx.x) * 4;
  int y = blockIdx.y;
  if (x4 >= d.mbpl) return;
  const uint8_t *r0 = im.img + (size_t)(2 * y) * d.bpl + 2 * x4;
  const uint8_t *r1 = r0 + d.bpl;
  uint32_t a0 = 0, a1 = 0, b0 = 0, b1 = 0;
  if (2 * x4 < d.bpl) {
    a0 = *(const uint32_t *)r0;
    b0 = *(const uint32_t *)r1;
  }
  if (2 * x4 + 4 < d.bpl) {
    a1 = *(const uint32_t *)(r0 + 4);
    b1 = *(const uint32_t *)(r1 + 4);
  }
  uint32_t out = 0;
#pragma unroll
  for (int k = 0; k < 4; k++) {
    uint32_t ta = k < 2 ? a0 : a1, tb = k < 2 ? b0 : b1;
    int sh = (k & 1) * 16;
    uint32_t s = ((ta >> sh) & 0xff) + ((ta >> (sh + 8)) & 0xff) + ((tb >> sh) & 0xff) + ((tb >> (sh + 8)) & 0xff);
    if (x4 + k < d.mw) out |= (s >> 2) << (8 * k);
  }
  *(uint32_t *)(im.imgm + (size_t)y * d.mbpl + x4) = out;
}

// ---------------------------------------------------------------------------------------
// F1-F3 image filters on the byte stream of one image (row wrap = the reference's SSE loops):
//   du,dv : filter::sobel5x5, viso/filter.cpp:316-324 (+128, >>7, unsigned saturate)
//   f1    : filter::blob5x5,  viso/filter.cpp:343-365  (-box5 + 2*box3 + 7*centre)
//   f2    : filter::checkerboard5x5, viso/filter.cpp:331-336 (c (x) c, c = 1,1,0,-1,-1)
// One thread produces a 4 x 4 pixel patch from an 8-row x 12-byte window held packed in 24
// registers (24 dword loads for 16 pixels; horizontal neighbours share lines in L1); per patch
// row it stores one dword of du, one of dv and, for the matching-resolution image, 8 bytes each
// of f1 and f2.  FULL = true: full-resolution image -> du_full,dv_full only.
// ---------------------------------------------------------------------------------------
// ---------------------------------------------------------------------------------------
// Full-resolution Sobel planes of half_resolution = 1 (read by the refinement only, as scattered 9 x 9 neighbourhoods):
// ONE plane of 8 x 8-pixel tiles, 128 bytes each = one cache line; a tile row is 16 bytes: du of pixels 0-3, dv of pixels
// 0-3, du of 4-7, dv of 4-7 (what a filter thread produces for its 4-pixel patch row is one 8-byte store).  A refinement
// window (9 rows x 9 columns of both responses) lies in exactly 4 lines instead of 16, its row in two 16-byte loads.
// du of pixel (x, y) at vsm_tiled_at(bpl, x, y), dv VSM_TILED_DV bytes further.
// ---------------------------------------------------------------------------------------
#define VSM_TILED_DV 4
__host__ __device__ __forceinline__ size_t vsm_tiled_at(int bpl, int x, int y) {
  return ((size_t)(y >> 3) * (size_t)(bpl >> 3) + (size_t)(x >> 3)) * 128 + (size_t)((y & 7) * 16 + ((x & 4) << 1) + (x & 3));
}

#define FPB(r, i) ((int)((Wn[(r)][(i) >> 2] >> (8 * ((i)&3))) & 0xffu))
template <bool FULL>
__global__ void __launch_bounds__(256)
    k_filters(const VsmImage *__restrict__ imgs, int first, int bpl, int h, int16_t *__restrict__ f1base,
              int16_t *__restrict__ f2base, size_t f_stride) {
  const VsmImage &im = imgs[first + blockIdx.z];
  const uint8_t *__restrict__ in = FULL ? im.img : im.imgm;
  const int n = bpl * h;
  const int x4 = (blockIdx.x * 64 + (threadIdx.x & 63)) * 4;  // 64 x 4 threads per block: 256 px x 16 rows
  const int y0 = (blockIdx.y * 4 + (threadIdx.x >> 6)) * 4;
  if (x4 >= bpl || y0 >= h) return;
  // window rows y0-2 .. y0+5, stream bytes x4-4 .. x4+7 of each
  uint32_t Wn[8][3];
#pragma unroll
  for (int r = 0; r < 8; r++) {
    const int base = (y0 + r - 2) * bpl + x4;
#pragma unroll
    for (int q = 0; q < 3; q++) {
      const int a = base + (q - 1) * 4;
      Wn[r][q] = (a >= 0 && a < n) ? *(const uint32_t *)(in + a) : 0u;
    }
  }
  const int lo = 2 * bpl, hi = (h - 2) * bpl;
  uint8_t *__restrict__ odu = FULL ? im.duv_tiled : im.du;
  uint8_t *__restrict__ odv = FULL ? im.duv_tiled + VSM_TILED_DV : im.dv;
  int16_t *f1 = FULL ? nullptr : f1base + (size_t)blockIdx.z * f_stride;
  int16_t *f2 = FULL ? nullptr : f2base + (size_t)blockIdx.z * f_stride;
#pragma unroll
  for (int rr = 0; rr < 4; rr++) {
    const int y = y0 + rr;
    if (y >= h) break;
    const int f0 = y * bpl + x4;
    // column pass at stream positions f0-2 .. f0+5 (window index 2..9); zero outside rows [2,h-3]
    int S[8], D[8];
#pragma unroll
    for (int i = 0; i < 8; i++) {
      const int g = f0 + i - 2;
      const bool ok = g >= lo && g < hi;
      const int a = FPB(rr, i + 2), b = FPB(rr + 1, i + 2), c = FPB(rr + 2, i + 2), dd = FPB(rr + 3, i + 2),
                e = FPB(rr + 4, i + 2);
      S[i] = ok ? a + 4 * b + 6 * c + 4 * dd + e : 0;
      D[i] = ok ? a + 2 * b - 2 * dd - e : 0;
    }
    uint32_t du = 0, dv = 0;
#pragma unroll
    for (int k = 0; k < 4; k++) {
      const int hu = S[k] + 2 * S[k + 1] - 2 * S[k + 3] - S[k + 4];
      const int hv = D[k] + 4 * D[k + 1] + 6 * D[k + 2] + 4 * D[k + 3] + D[k + 4];
      du |= (uint32_t)min(max((hu >> 7) + 128, 0), 255) << (8 * k);
      dv |= (uint32_t)min(max((hv >> 7) + 128, 0), 255) << (8 * k);
    }
    const size_t o0 = FULL ? vsm_tiled_at(bpl, x4, y) : (size_t)f0;
    *(uint32_t *)(odu + o0) = du;
    *(uint32_t *)(odv + o0) = dv;
    if (!FULL) {
      int16_t o1[4], o2[4];
#pragma unroll
      for (int k = 0; k < 4; k++) {
        const int i = 4 + k, x = x4 + k;
        int b5 = 0, b3 = 0, ck = 0;
#pragma unroll
        for (int r = 0; r < 5; r++) {
          const int r3 = FPB(rr + r, i - 1) + FPB(rr + r, i) + FPB(rr + r, i + 1);
          const int r5 = r3 + FPB(rr + r, i - 2) + FPB(rr + r, i + 2);
          const int rd = FPB(rr + r, i - 2) + FPB(rr + r, i - 1) - FPB(rr + r, i + 1) - FPB(rr + r, i + 2);
          b5 += r5;
          if (r >= 1 && r <= 3) b3 += r3;
          ck += (r < 2) ? rd : (r > 2 ? -rd : 0);
        }
        const bool in1 = x >= 3 && x <= bpl - 3 && y >= 3 && y <= h - 3;
        const bool in2 = x >= 2 && x <= bpl - 3 && y >= 2 && y <= h - 3;
        o1[k] = in1 ? (int16_t)(-b5 + 2 * b3 + 7 * FPB(rr + 2, i)) : (int16_t)0;
        o2[k] = in2 ? (int16_t)ck : (int16_t)0;
      }
      *(uint2 *)(f1 + f0) = make_uint2((uint16_t)o1[0] | ((uint32_t)(uint16_t)o1[1] << 16),
                                       (uint16_t)o1[2] | ((uint32_t)(uint16_t)o1[3] << 16));
      *(uint2 *)(f2 + f0) = make_uint2((uint16_t)o2[0] | ((uint32_t)(uint16_t)o2[1] << 16),
                                       (uint16_t)o2[2] | ((uint32_t)(uint16_t)o2[3] << 16));
    }
  }
}
#undef FPB

// ---------------------------------------------------------------------------------------
// Fused front end for half_resolution = 1: ONE pass over the caller's image produces what k_ingest + k_halve +
// k_filters<true> produced in three (each streaming the full image): the padded copy (only where something reads it
// later - the per-frame path's getGain), the half-resolution image (F0, viso/matcher.cpp:636-647) and the full-resolution
// Sobel planes (F1, viso/filter.cpp:316-324), which only the refinement reads.
// A block owns a tile of 128 x 64 pixels.  Its 68 x 136 bytes of input (2 rows / 4 bytes of halo) go to LDS as coalesced
// loads in the layout of the PADDED byte stream (row stride bpl, pad bytes 0, positions outside the image 0 - the
// reference's filters run over that stream and wrap at row ends): caller rows need no alignment (aligned loads and a
// byte funnel shift per LDS dword, as in k_ingest).  Then every thread takes an 8 x 4 patch from an 8-row x 16-byte LDS
// window (two pixels per instruction, vsm_feat.h) and 8 half-resolution pixels.
// ---------------------------------------------------------------------------------------
#define FRONT_TW 128
#define FRONT_TH 64
#ifndef FRONT_LPAD
#define FRONT_LPAD 16  // (8 is all the patches read; with 16 a row is nine WHOLE 16-byte items and an interior tile never takes the dword path)
#endif
#define FRONT_LW (FRONT_TW + FRONT_LPAD)  // bytes per LDS row
#define FRONT_LH (FRONT_TH + 4)
__global__ void __launch_bounds__(256)
    k_front(const VsmImage *__restrict__ imgs, int first, const uint8_t *__restrict__ src0, const uint8_t *__restrict__ src1,
            size_t frame_stride, int src_bpl, int sides, VsmDims d, int write_img, int nbx, int nby, int n_img) {
  __shared__ __attribute__((aligned(16))) uint32_t s_in[FRONT_LH][FRONT_LW / 4];
  // (XCD-aware placement: neighbouring tiles share the 128-byte lines their unaligned rows straddle and two halo rows; dealt
  // round-robin over the eight L2s every such line was fetched from HBM twice: 1.41 x the kernel's bytes)
  const int lb = xcd_remap(blockIdx.x, gridDim.x);
  const int zi = lb / (nbx * nby), rem = lb - zi * (nbx * nby);
  if (zi >= n_img) return;
  const int by = rem / nbx, bx = rem - by * nbx;
  const int fr = zi / sides, side = zi - fr * sides;
  const uint8_t *__restrict__ src = (side ? src1 : src0) + (size_t)fr * frame_stride;
  const VsmImage &im = imgs[first + 2 * fr + side];
  const int bpl = d.bpl, h = d.h, w = d.w;
  const int x0 = bx * FRONT_TW, y0 = by * FRONT_TH;
  const int t = threadIdx.x;
  // ---- tile of the padded stream into LDS ----
  // Items of four LDS dwords (16 stream bytes; FRONT_LPAD = 8: the last item of a row has two).  An item that lies inside one image row
  // takes ONE aligned 16-byte load plus one dword and four byte-aligns; the others (row ends, tile edges, rows outside the
  // image) go dword by dword: two aligned dwords + an align each, bytes beyond the row's w and positions outside the
  // image 0.  (Every 4-byte piece used to cost two loads: the kernel was bound by the texture addresser's lane rate.)
  auto load_dword = [&](int y, int x) -> uint32_t {  // stream position y * bpl + x; x may run into the neighbouring stream rows
    if (x < 0) {
      x += bpl;
      y -= 1;
    } else if (x >= bpl) {
      x -= bpl;
      y += 1;
    }
    uint32_t v = 0;
    if (y >= 0 && y < h && x < w) {
      const uintptr_t a = (uintptr_t)(src + (size_t)y * src_bpl + x);
      const uint32_t sh = (uint32_t)(a & 3);
      const int nvalid = min(4, w - x);  // bytes of this dword that belong to the row
      const uint32_t lo = ldg_u32((const void *)(a - sh));
      const uint32_t hi = ((int)(4 - sh) < nvalid) ? ldg_u32((const void *)(a - sh + 4)) : 0u;
      v = __builtin_amdgcn_alignbyte(hi, lo, sh);
      if (nvalid < 4) v &= (1u << (8 * nvalid)) - 1u;
    }
    return v;
  };
  FT_DECL;
  FT_STAMP;
  constexpr int kItems = (FRONT_LW / 4 + 3) / 4;  // per row
  constexpr int kIters = (FRONT_LH * kItems + 255) / 256;
  // (all of a thread's wide loads are requested before the first LDS store waits for one: three round trips become one)
  uint4 q[kIters];
  uint32_t q4[kIters];
#pragma unroll
  for (int i = 0; i < kIters; i++) {
    const int e = t + 256 * i, r = e / kItems, g = e - r * kItems;
    const int y = y0 - 2 + r, x = x0 - 4 + 16 * g;
    const int nd = min(4, FRONT_LW / 4 - 4 * g);
    q[i] = make_uint4(0u, 0u, 0u, 0u);
    q4[i] = 0u;
    if (e < FRONT_LH * kItems && nd == 4 && y >= 0 && y < h && x >= 0 && x + 20 <= w) {
      const uintptr_t a = (uintptr_t)(src + (size_t)y * src_bpl + x);
      const uint32_t sh = (uint32_t)(a & 3);
      q[i] = ldg_u4((const void *)(a - sh));
      q4[i] = ldg_u32((const void *)(a - sh + 16));
    }
  }
#pragma unroll
  for (int i = 0; i < kIters; i++) {
    const int e = t + 256 * i, r = e / kItems, g = e - r * kItems;
    if (e >= FRONT_LH * kItems) continue;
    const int y = y0 - 2 + r, x = x0 - 4 + 16 * g;
    const int nd = min(4, FRONT_LW / 4 - 4 * g);
    if (nd == 4 && y >= 0 && y < h && x >= 0 && x + 20 <= w) {
      const uint32_t sh = (uint32_t)((uintptr_t)(src + (size_t)y * src_bpl + x) & 3);
      uint4 v;
      v.x = __builtin_amdgcn_alignbyte(q[i].y, q[i].x, sh);
      v.y = __builtin_amdgcn_alignbyte(q[i].z, q[i].y, sh);
      v.z = __builtin_amdgcn_alignbyte(q[i].w, q[i].z, sh);
      v.w = __builtin_amdgcn_alignbyte(q4[i], q[i].w, sh);
      if ((FRONT_LW & 15) == 0) {
        *(uint4 *)&s_in[r][4 * g] = v;  // (rows of whole items are 16-byte aligned)
      } else {
        s_in[r][4 * g] = v.x;
        s_in[r][4 * g + 1] = v.y;
        s_in[r][4 * g + 2] = v.z;
        s_in[r][4 * g + 3] = v.w;
      }
    } else if ((y < 0 || y >= h) && x >= 0 && x + 16 <= bpl) {  // a row outside the image whose item does not wrap into one inside
      for (int k = 0; k < nd; k++) s_in[r][4 * g + k] = 0u;
    } else {
      for (int k = 0; k < nd; k++) s_in[r][4 * g + k] = load_dword(y, x + 4 * k);
    }
  }
  FT_STAMP;
  __syncthreads();
  FT_STAMP;
  // ---- padded copy (pad bytes 0), where asked for ----
  if (write_img) {
    for (int e = t; e < FRONT_TH * (FRONT_TW / 4); e += 256) {
      const int r = e / (FRONT_TW / 4), c = e - r * (FRONT_TW / 4);
      const int y = y0 + r, x4 = x0 + 4 * c;
      if (y < h && x4 < bpl) *(VSM_AS1 uint32_t *)(im.img + (size_t)y * bpl + x4) = s_in[r + 2][c + 1];
    }
  }
  // ---- half-resolution image: 64 x 32 pixels of this tile, 2 x 4 per thread ----
#pragma unroll
  for (int k2 = 0; k2 < FRONT_TH / 32; k2++) {
    const int idx = t + 256 * k2, hy = idx >> 4, hx4 = (idx & 15) * 4;
    const int my = (y0 >> 1) + hy, mx4 = (x0 >> 1) + hx4;
    if (my < d.mh && mx4 < d.mbpl) {
      const uint32_t a0 = s_in[2 + 2 * hy][1 + (hx4 >> 1)], a1 = s_in[2 + 2 * hy][2 + (hx4 >> 1)];
      const uint32_t b0 = s_in[3 + 2 * hy][1 + (hx4 >> 1)], b1 = s_in[3 + 2 * hy][2 + (hx4 >> 1)];
      uint32_t out = 0;
#pragma unroll
      for (int k = 0; k < 4; k++) {
        const uint32_t ta = k < 2 ? a0 : a1, tb = k < 2 ? b0 : b1;
        const int sh = (k & 1) * 16;
        const uint32_t sum = ((ta >> sh) & 0xff) + ((ta >> (sh + 8)) & 0xff) + ((tb >> sh) & 0xff) + ((tb >> (sh + 8)) & 0xff);
        if (mx4 + k < d.mw) out |= (sum >> 2) << (8 * k);
      }
      *(VSM_AS1 uint32_t *)(im.imgm + (size_t)my * d.mbpl + mx4) = out;
    }
  }
  FT_STAMP;
  // ---- full-resolution Sobel: an 8 x 4 patch per thread from a window of 8 rows x 16 bytes, two pixels per instruction in
  // 16-bit lanes (vsm_feat.h: the same column / row passes as the matching-resolution tiles); a patch row = du 0-3, dv 0-3,
  // du 4-7, dv 4-7 = one 16-byte row of the tiled plane ----
  static_assert(FRONT_TW == 128 && FRONT_TH == 64, "16 x 16 patches of 8 x 4 pixels");
  const int tx = t & 15, ty = t >> 4;
  const int x8 = x0 + 8 * tx, yb = y0 + 4 * ty;
#ifndef FRONT_LINE_STORES
#define FRONT_LINE_STORES 0
#endif
#if FRONT_LINE_STORES
  // (Measured, off: 106.3 against 107.1 us per 200 images in the pipeline - the kernel is bound by its vector instructions,
  // not by the shape of its stores - for 9 registers and 16 KB of LDS more.)
  // The patch rows leave through LDS so that every store instruction writes WHOLE 128-byte lines of the tiled plane (a
  // line = 8 rows of one 8-pixel column = the patches of two threads): a wave's 32 lines are staged in 4 KB of its own
  // (pieces XOR-swizzled by the line so that the 16 lanes of a patch column do not share banks), then lane L of store k
  // takes piece L & 7 of line 8 k + (L >> 3) - 1 KB of contiguous bytes per instruction instead of 16-byte pieces of 32 lines.
  __shared__ __attribute__((aligned(16))) vsm_u4 s_out[4][256];
  vsm_u4 o[4];
#pragma unroll
  for (int rr = 0; rr < 4; rr++) o[rr].x = o[rr].y = o[rr].z = o[rr].w = 0u;
  if (x8 < bpl && yb < h) {
    VfWindow<4> W;
#pragma unroll
    for (int r = 0; r < 8; r++) {
#pragma unroll
      for (int q = 0; q < 4; q++) W.w[r][q] = s_in[4 * ty + r][2 * tx + q];
    }
#pragma unroll
    for (int rr = 0; rr < 4; rr++) {
      const int y = yb + rr;
      if (y >= h) break;
      vf_s2 S[6], D[6];
#pragma unroll
      for (int j = 0; j < 6; j++) vf_columns_sobel<4>(W, rr, j, S[j], D[j]);
      if (y < 3 || y > h - 4) vf_sobel_zero_outside(S, D, y * bpl + x8, 2 * bpl, (h - 2) * bpl);
      uint32_t du[2], dv[2];
      vf_sobel_row(S, D, du, dv);
      o[rr].x = du[0];
      o[rr].y = dv[0];
      o[rr].z = du[1];
      o[rr].w = dv[1];
    }
  }
  {
    const int wv = t >> 6, lane = t & 63, tyl = ty & 3;
    const int lw = (tyl >> 1) * 16 + tx;  // the thread's line among its wave's 32
#pragma unroll
    for (int rr = 0; rr < 4; rr++) s_out[wv][lw * 8 + (((tyl & 1) * 4 + rr) ^ (lw & 7))] = o[rr];
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
    for (int k = 0; k < 4; k++) {
      const int l = k * 8 + (lane >> 3), pc = lane & 7;
      const vsm_u4 v = s_out[wv][l * 8 + (pc ^ (l & 7))];
      const int xx = x0 + 8 * (l & 15), yy = y0 + 16 * wv + 8 * (l >> 4) + pc;
      if (xx < bpl && yy < h) *(VSM_AS1 vsm_u4 *)(im.duv_tiled + vsm_tiled_at(bpl, xx, yy)) = v;
    }
  }
#else
  if (x8 >= bpl || yb >= h) return;
  VfWindow<4> W;
#pragma unroll
  for (int r = 0; r < 8; r++) {
#pragma unroll
    for (int q = 0; q < 4; q++) W.w[r][q] = s_in[4 * ty + r][2 * tx + q];
  }
#ifndef FRONT_CASCADE
#define FRONT_CASCADE 1  // the column sums of the patch's four rows share their stages (vf_columns_sobel_all)
#endif
#if FRONT_CASCADE
  vf_s2 Sa[4][6], Da[4][6];
#pragma unroll
  for (int j = 0; j < 6; j++) {
    vf_s2 sj[4], dj[4];
    vf_columns_sobel_all<4>(W, j, sj, dj);
#pragma unroll
    for (int rr = 0; rr < 4; rr++) Sa[rr][j] = sj[rr], Da[rr][j] = dj[rr];
  }
#endif
#pragma unroll
  for (int rr = 0; rr < 4; rr++) {
    const int y = yb + rr;
    if (y >= h) break;
#if FRONT_CASCADE
    vf_s2 (&S)[6] = Sa[rr], (&D)[6] = Da[rr];
#else
    vf_s2 S[6], D[6];
#pragma unroll
    for (int j = 0; j < 6; j++) vf_columns_sobel<4>(W, rr, j, S[j], D[j]);
#endif
    if (y < 3 || y > h - 4) vf_sobel_zero_outside(S, D, y * bpl + x8, 2 * bpl, (h - 2) * bpl);
    uint32_t du[2], dv[2];
    vf_sobel_row(S, D, du, dv);
    vsm_u4 o;
    o.x = du[0];
    o.y = dv[0];
    o.z = du[1];
    o.w = dv[1];
    *(VSM_AS1 vsm_u4 *)(im.duv_tiled + vsm_tiled_at(bpl, x8, y)) = o;
  }
#endif
  FT_STAMP;
  FT_FLUSH(4);
}

// ---------------------------------------------------------------------------------------
// N1 nonMaximumSuppression, viso/matcher.cpp:330-431 (Neubeck & Van Gool alg. 4).
// One wavefront per (cell, filter).  Lanes load the (n+1)^2 cell pixels (u fastest: contiguous
// int16 rows) and a 64-lane min-reduction over the key (value, scan position) yields the
// reference's first-wins extremum ("u outer, v inner, strict compare", :356-380): position
// o = di*(n+1)+dj.  The (2n+1)^2 suppression windows (:383-428) are then tested lane-parallel and
// folded with a ballot.  blockIdx.y: 0 = f1, 1 = f2; blockIdx.z = image*2 + set.
// Survivors go to cand[cell*4 + class] as u | v<<14 | 1<<31.
// ---------------------------------------------------------------------------------------
__device__ __forceinline__ uint32_t wave_min_u32(uint32_t v) {
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) v = min(v, (uint32_t)__shfl_xor((int)v, o, 64));
  return v;
}

__device__ __forceinline__ bool nms_suppressed_wave(const int16_t *__restrict__ f, int bpl, int w, int h, int n, int i,
                                                    int j, int ci, int cj, int val, bool want_min, int lane) {
  const int W = 2 * n + 1;
  const int i_hi = min(ci + n, w - 1 - VSM_MARGIN), j_hi = min(cj + n, h - 1 - VSM_MARGIN);
  bool hit = false;
  for (int e = lane; e < W * W; e += 64) {
    const int dj = e / W, di = e - dj * W;
    const int i2 = ci - n + di, j2 = cj - n + dj;
    if (i2 <= i_hi && j2 <= j_hi) {
      const int cur = f[j2 * bpl + i2];
      const bool better = want_min ? (cur < val) : (cur > val);
      hit |= better && (i2 < i || i2 > i + n || j2 < j || j2 > j + n);
    }
  }
  return __ballot(hit) != 0ull;
}

__global__ void __launch_bounds__(256) k_nms(const VsmImage *__restrict__ imgs, int first, VsmDims d,
                                             const int16_t *__restrict__ f1base, const int16_t *__restrict__ f2base,
                                             size_t f_stride, int tau, int set_lo, int set_hi) {
  const int zi = blockIdx.z >> 1, si = (blockIdx.z & 1);
  if (si < set_lo || si > set_hi) return;
  const VsmSet &st = imgs[first + zi].set[si];
  const int ncells = st.ncu * st.ncv;
  const int lane = threadIdx.x & 63;
  const int t = blockIdx.x * 4 + (threadIdx.x >> 6);  // wave-uniform
  if (t >= ncells) return;
  const int k = blockIdx.y;
  const int16_t *__restrict__ f = (k ? f2base : f1base) + (size_t)zi * f_stride;
  const int n = st.nms_n, n1 = n + 1;
  const int cj = t / st.ncu, ci = t - cj * st.ncu;
  const int i = n + VSM_MARGIN + ci * n1, j = n + VSM_MARGIN + cj * n1;
  uint32_t kmin = 0xffffffffu, kmax = 0xffffffffu;
  for (int e = lane; e < n1 * n1; e += 64) {
    const int dj = e / n1, di = e - dj * n1;
    const int val = f[(j + dj) * d.mbpl + i + di];
    const uint32_t o = (uint32_t)(di * n1 + dj);
    kmin = min(kmin, ((uint32_t)(val + 32768) << 10) | o);
    kmax = min(kmax, ((uint32_t)(32767 - val) << 10) | o);
  }
  kmin = wave_min_u32(kmin);
  kmax = wave_min_u32(kmax);
  const int mnv = (int)(kmin >> 10) - 32768, mno = kmin & 1023;
  const int mxv = 32767 - (int)(kmax >> 10), mxo = kmax & 1023;
  const int mni = i + mno / n1, mnj = j + mno % n1, mxi = i + mxo / n1, mxj = j + mxo % n1;
  bool vmin = mnv <= -tau, vmax = mxv >= tau;  // wave-uniform
  if (vmin) vmin = !nms_suppressed_wave(f, d.mbpl, d.mw, d.mh, n, i, j, mni, mnj, mnv, true, lane);
  if (vmax) vmax = !nms_suppressed_wave(f, d.mbpl, d.mw, d.mh, n, i, j, mxi, mxj, mxv, false, lane);
  if (lane == 0) {
    int32_t *c = st.cand + (size_t)(ci * st.ncv + cj) * 4 + 2 * k;
    c[0] = vmin ? (int32_t)(0x80000000u | (uint32_t)mni | ((uint32_t)mnj << 14)) : 0;
    c[1] = vmax ? (int32_t)(0x80000000u | (uint32_t)mxi | ((uint32_t)mxj << 14)) : 0;
  }
}

// Tile variant for small n (the dense set, n = nms_n <= 4): a 256-thread block stages the f1/f2
// responses of 16 x 8 cells plus the n-pixel halo in LDS (coalesced row reads, 11 KB at n = 3) and
// every thread then resolves one (cell, filter) entirely from LDS -- (n+1)^2 + 2 (2n+1)^2 reads --
// instead of spending a whole wavefront per cell.  Same scan order, same strict compares.
#define NMS_TCU 16
#define NMS_TCV 8
#define NMS_TILE_MAXN 4
__global__ void __launch_bounds__(256)
    k_nms_tile(const VsmImage *__restrict__ imgs, int first, VsmDims d, const int16_t *__restrict__ f1base,
               const int16_t *__restrict__ f2base, size_t f_stride, int tau, int si, int nbx, int n_img) {
  constexpr int MAXW = NMS_TCU * (NMS_TILE_MAXN + 1) + 2 * NMS_TILE_MAXN;  // 88
  constexpr int MAXH = NMS_TCV * (NMS_TILE_MAXN + 1) + 2 * NMS_TILE_MAXN;  // 48
  __shared__ int16_t s_f[2][MAXH][MAXW + 2];
  const int lb = xcd_remap(blockIdx.x, gridDim.x);
  const int zi = lb / nbx, bx = lb - zi * nbx;  // (image, tile)
  if (zi >= n_img) return;
  const VsmSet &st = imgs[first + zi].set[si];
  const int n = st.nms_n, n1 = n + 1;
  const int tiles_u = (st.ncu + NMS_TCU - 1) / NMS_TCU;
  const int tu = bx % tiles_u, tv = bx / tiles_u;
  const int cu0 = tu * NMS_TCU, cv0 = tv * NMS_TCV;
  const int u0 = VSM_MARGIN + cu0 * n1, v0 = VSM_MARGIN + cv0 * n1;  // = first cell origin - n
  const int tw = NMS_TCU * n1 + 2 * n, th = NMS_TCV * n1 + 2 * n;
  const int16_t *__restrict__ f1 = f1base + (size_t)zi * f_stride;
  const int16_t *__restrict__ f2 = f2base + (size_t)zi * f_stride;
  for (int e = threadIdx.x; e < tw * th; e += 256) {
    const int y = e / tw, x = e - y * tw;
    const int u = u0 + x, v = v0 + y;
    const bool in = u < d.mbpl && v < d.mh;
    s_f[0][y][x] = in ? f1[v * d.mbpl + u] : (int16_t)0;
    s_f[1][y][x] = in ? f2[v * d.mbpl + u] : (int16_t)0;
  }
  __syncthreads();
  const int k = threadIdx.x & 1, cl = threadIdx.x >> 1;
  const int lcu = cl % NMS_TCU, lcv = cl / NMS_TCU;
  const int ci = cu0 + lcu, cj = cv0 + lcv;
  if (ci >= st.ncu || cj >= st.ncv) return;
  const int16_t(*f)[MAXW + 2] = s_f[k];
  // tile-local coordinates of the cell origin; image coordinates = local + (u0, v0)
  const int li = n + lcu * n1, lj = n + lcv * n1;
  int mni = li, mnj = lj, mxi = li, mxj = lj;
  int mnv = f[lj][li], mxv = mnv;
  for (int i2 = li; i2 <= li + n; i2++)
    for (int j2 = lj; j2 <= lj + n; j2++) {
      const int cur = f[j2][i2];
      if (cur < mnv) {
        mni = i2;
        mnj = j2;
        mnv = cur;
      } else if (cur > mxv) {
        mxi = i2;
        mxj = j2;
        mxv = cur;
      }
    }
  // clip limits of the suppression windows (w-1-margin, h-1-margin) in tile-local coordinates
  const int lim_i = d.mw - 1 - VSM_MARGIN - u0, lim_j = d.mh - 1 - VSM_MARGIN - v0;
  auto suppressed = [&](int ci2, int cj2, int val, bool want_min) -> bool {
    const int i_hi = min(ci2 + n, lim_i), j_hi = min(cj2 + n, lim_j);
    for (int i2 = ci2 - n; i2 <= i_hi; i2++)
      for (int j2 = cj2 - n; j2 <= j_hi; j2++) {
        const int cur = f[j2][i2];
        const bool better = want_min ? (cur < val) : (cur > val);
        if (better && (i2 < li || i2 > li + n || j2 < lj || j2 > lj + n)) return true;
      }
    return false;
  };
  const bool vmin = (mnv <= -tau) && !suppressed(mni, mnj, mnv, true);
  const bool vmax = (mxv >= tau) && !suppressed(mxi, mxj, mxv, false);
  int32_t *c = st.cand + (size_t)(ci * st.ncv + cj) * 4 + 2 * k;
  c[0] = vmin ? (int32_t)(0x80000000u | (uint32_t)(mni + u0) | ((uint32_t)(mnj + v0) << 14)) : 0;
  c[1] = vmax ? (int32_t)(0x80000000u | (uint32_t)(mxi + u0) | ((uint32_t)(mxj + v0) << 14)) : 0;
}

// Tile variant for mid-size n (the sparse set, 5 <= n <= 10): 4 x 4 cells + halo in LDS, 8 lanes
// per (cell, filter): the lanes stride over the cell / window pixels and combine with width-8
// shuffles (min of (value, scan position) keys, OR of the suppression hits).
#define NMS8_TC 4
#define NMS8_MAXN 10
__global__ void __launch_bounds__(256)
    k_nms_tile8(const VsmImage *__restrict__ imgs, int first, VsmDims d, const int16_t *__restrict__ f1base,
                const int16_t *__restrict__ f2base, size_t f_stride, int tau, int si, int nbx, int n_img) {
  constexpr int MAXD = NMS8_TC * (NMS8_MAXN + 1) + 2 * NMS8_MAXN;  // 64
  __shared__ int16_t s_f[2][MAXD][MAXD + 2];
  const int lb = xcd_remap(blockIdx.x, gridDim.x);
  const int zi = lb / nbx, bx = lb - zi * nbx;  // (image, tile)
  if (zi >= n_img) return;
  const VsmSet &st = imgs[first + zi].set[si];
  const int n = st.nms_n, n1 = n + 1;
  const int tiles_u = (st.ncu + NMS8_TC - 1) / NMS8_TC;
  const int tu = bx % tiles_u, tv = bx / tiles_u;
  const int cu0 = tu * NMS8_TC, cv0 = tv * NMS8_TC;
  const int u0 = VSM_MARGIN + cu0 * n1, v0 = VSM_MARGIN + cv0 * n1;
  const int tw = NMS8_TC * n1 + 2 * n;
  const int16_t *__restrict__ f1 = f1base + (size_t)zi * f_stride;
  const int16_t *__restrict__ f2 = f2base + (size_t)zi * f_stride;
  for (int e = threadIdx.x; e < tw * tw; e += 256) {
    const int y = e / tw, x = e - y * tw;
    const int u = u0 + x, v = v0 + y;
    const bool in = u < d.mbpl && v < d.mh;
    s_f[0][y][x] = in ? f1[v * d.mbpl + u] : (int16_t)0;
    s_f[1][y][x] = in ? f2[v * d.mbpl + u] : (int16_t)0;
  }
  __syncthreads();
  const int l8 = threadIdx.x & 7, item = threadIdx.x >> 3;  // 32 items: 16 cells x 2 filters
  const int k = item & 1, cl = item >> 1;
  const int lcu = cl % NMS8_TC, lcv = cl / NMS8_TC;
  const int ci = cu0 + lcu, cj = cv0 + lcv;
  const bool live = ci < st.ncu && cj < st.ncv;  // dead items still take part in the shuffles
  const int16_t(*f)[MAXD + 2] = s_f[k];
  const int li = n + lcu * n1, lj = n + lcv * n1;
  uint32_t kmin = 0xffffffffu, kmax = 0xffffffffu;
  for (int e = l8; e < n1 * n1; e += 8) {
    const int dj = e / n1, di = e - dj * n1;
    const int val = f[lj + dj][li + di];
    const uint32_t o = (uint32_t)(di * n1 + dj);
    kmin = min(kmin, ((uint32_t)(val + 32768) << 10) | o);
    kmax = min(kmax, ((uint32_t)(32767 - val) << 10) | o);
  }
#pragma unroll
  for (int o = 4; o >= 1; o >>= 1) {
    kmin = min(kmin, (uint32_t)__shfl_xor((int)kmin, o, 8));
    kmax = min(kmax, (uint32_t)__shfl_xor((int)kmax, o, 8));
  }
  const int mnv = (int)(kmin >> 10) - 32768, mno = kmin & 1023;
  const int mxv = 32767 - (int)(kmax >> 10), mxo = kmax & 1023;
  const int mni = li + mno / n1, mnj = lj + mno % n1, mxi = li + mxo / n1, mxj = lj + mxo % n1;
  const int lim_i = d.mw - 1 - VSM_MARGIN - u0, lim_j = d.mh - 1 - VSM_MARGIN - v0;
  const int W = 2 * n + 1;
  auto suppressed = [&](int ci2, int cj2, int val, bool want_min) -> bool {
    const int i_hi = min(ci2 + n, lim_i), j_hi = min(cj2 + n, lim_j);
    int hit = 0;
    for (int e = l8; e < W * W; e += 8) {
      const int dj = e / W, di = e - dj * W;
      const int i2 = ci2 - n + di, j2 = cj2 - n + dj;
      if (i2 <= i_hi && j2 <= j_hi) {
        const int cur = f[j2][i2];
        const bool better = want_min ? (cur < val) : (cur > val);
        hit |= (better && (i2 < li || i2 > li + n || j2 < lj || j2 > lj + n)) ? 1 : 0;
      }
    }
#pragma unroll
    for (int o = 4; o >= 1; o >>= 1) hit |= __shfl_xor(hit, o, 8);
    return hit != 0;
  };
  // both tests run unconditionally so that all 8 lanes of every item reach the shuffles together
  const bool smin = suppressed(mni, mnj, mnv, true);
  const bool smax = suppressed(mxi, mxj, mxv, false);
  if (live && l8 == 0) {
    const bool vmin = (mnv <= -tau) && !smin, vmax = (mxv >= tau) && !smax;
    int32_t *c = st.cand + (size_t)(ci * st.ncv + cj) * 4 + 2 * k;
    c[0] = vmin ? (int32_t)(0x80000000u | (uint32_t)(mni + u0) | ((uint32_t)(mnj + v0) << 14)) : 0;
    c[1] = vmax ? (int32_t)(0x80000000u | (uint32_t)(mxi + u0) | ((uint32_t)(mxj + v0) << 14)) : 0;
  }
}

// Fixed-radius variants for the two radii the default parameters produce (dense n = 3, sparse n = 9;
// viso/matcher.cpp:685-687): same LDS tiling as above, but with n a compile-time constant every
// LDS read of the cell scan and of the (2n+1)^2 suppression window has an immediate offset from one
// per-thread base address, and the window test is branch-free: a candidate is the extremum of its
// own cell, so "no strictly better value in the window outside the cell" (:383-428) is simply
// "the window extremum equals the candidate's value".  Tiles that touch the right/bottom clip limit
// (w-1-margin, h-1-margin) take a bounded loop instead.  N = 3: one thread per (cell, filter);
// N = 9: 8 lanes per (cell, filter), each owning window rows l8, l8+8, l8+16.
template <int N, int TCU, int TCV, int LANES>
__global__ void __launch_bounds__(256)
    k_nms_fixed(const VsmImage *__restrict__ imgs, int first, VsmDims d, const int16_t *__restrict__ f1base,
                const int16_t *__restrict__ f2base, size_t f_stride, int tau, int si, int nbx, int n_img) {
  constexpr int N1 = N + 1, W = 2 * N + 1;
  constexpr int TW = TCU * N1 + 2 * N, TH = TCV * N1 + 2 * N;
  constexpr int STR = (((TW + 1) / 2) | 1) * 2;  // int16 per LDS row: an odd number of dwords
  static_assert(TCU * TCV * 2 * LANES == 256, "one item per LANES threads");
  static_assert((N1 & 1) == 0, "tile origins must stay dword aligned");
  __shared__ int16_t s_f[2][TH][STR];
  const int lb = xcd_remap(blockIdx.x, gridDim.x);
  const int zi = lb / nbx, bx = lb - zi * nbx;  // (image, tile)
  if (zi >= n_img) return;
  const VsmSet &st = imgs[first + zi].set[si];
  const int tiles_u = (st.ncu + TCU - 1) / TCU;
  const int tu = bx % tiles_u, tv = bx / tiles_u;
  const int cu0 = tu * TCU, cv0 = tv * TCV;
  const int u0 = VSM_MARGIN + cu0 * N1, v0 = VSM_MARGIN + cv0 * N1;  // = first cell origin - N (even)
  const int16_t *__restrict__ f1 = f1base + (size_t)zi * f_stride;
  const int16_t *__restrict__ f2 = f2base + (size_t)zi * f_stride;
  for (int e = threadIdx.x; e < (TW / 2) * TH; e += 256) {  // two pixels per load
    const int y = e / (TW / 2), x = 2 * (e - y * (TW / 2));
    const int u = u0 + x, v = v0 + y;
    const bool in = u + 1 < d.mbpl && v < d.mh;
    const uint32_t a = in ? *(const uint32_t *)(f1 + v * d.mbpl + u) : 0u;
    const uint32_t b = in ? *(const uint32_t *)(f2 + v * d.mbpl + u) : 0u;
    *(uint32_t *)&s_f[0][y][x] = a;
    *(uint32_t *)&s_f[1][y][x] = b;
  }
  __syncthreads();
  const int l8 = threadIdx.x % LANES, item = threadIdx.x / LANES;
  const int k = item & 1, cl = item >> 1;
  const int lcu = cl % TCU, lcv = cl / TCU;
  const int ci = cu0 + lcu, cj = cv0 + lcv;
  const bool live = ci < st.ncu && cj < st.ncv;  // dead items still take part in the shuffles
  const int16_t(*f)[STR] = s_f[k];
  const int li = N + lcu * N1, lj = N + lcv * N1;  // tile-local cell origin
  // first-wins extrema of the cell in the reference's scan order (u outer, v inner, strict compare,
  // :356-380): minimum of the key (value, scan position)
  uint32_t kmin = 0xffffffffu, kmax = 0xffffffffu;
  {
    const int16_t *c0 = &f[lj][li];
#pragma unroll
    for (int t = 0; t < (N1 + LANES - 1) / LANES; t++) {
      const int di = l8 + t * LANES;  // this lane's column(s) of the cell
      if (di < N1) {
#pragma unroll
        for (int dj = 0; dj < N1; dj++) {
          const int val = c0[dj * STR + di];
          const uint32_t o = (uint32_t)(di * N1 + dj);
          kmin = min(kmin, ((uint32_t)(val + 32768) << 10) | o);
          kmax = min(kmax, ((uint32_t)(32767 - val) << 10) | o);
        }
      }
    }
  }
#pragma unroll
  for (int o = LANES / 2; o >= 1; o >>= 1) {
    kmin = min(kmin, (uint32_t)__shfl_xor((int)kmin, o, LANES));
    kmax = min(kmax, (uint32_t)__shfl_xor((int)kmax, o, LANES));
  }
  const int mnv = (int)(kmin >> 10) - 32768, mno = kmin & 1023;
  const int mxv = 32767 - (int)(kmax >> 10), mxo = kmax & 1023;
  const int mni = li + mno / N1, mnj = lj + mno % N1, mxi = li + mxo / N1, mxj = lj + mxo % N1;
  const int lim_i = d.mw - 1 - VSM_MARGIN - u0, lim_j = d.mh - 1 - VSM_MARGIN - v0;
  int wmn = 32767, wmx = -32768;  // extrema over this lane's share of the two windows
  if (lim_i >= TW - 1 && lim_j >= TH - 1) {  // block-uniform: no window of this tile is clipped
    // A window row is W = 2N+1 values from column mi-N on: (W+1)/2 aligned dwords cover it whatever the parity of that
    // column, with one value too many - the last one (even start) or the first (odd start) - which is replaced by the
    // neutral element; then two values per v_pk_min_i16 / v_pk_max_i16.  (Half the LDS reads of the value-by-value walk:
    // the kernels spend most of their time here.)
    typedef short vsm_s2 __attribute__((ext_vector_type(2)));
    constexpr int ND = (W + 1) / 2;
    const int cn = mni - N, cx = mxi - N;
    const bool pn1 = (cn & 1) != 0, px1 = (cx & 1) != 0;
    const uint32_t *pn = (const uint32_t *)&f[mnj - N][cn & ~1], *px = (const uint32_t *)&f[mxj - N][cx & ~1];
    vsm_s2 amn = {32767, 32767}, amx = {-32768, -32768};
#pragma unroll
    for (int t = 0; t < (W + LANES - 1) / LANES; t++) {
      const int r = l8 + t * LANES;
      if (r < W) {
        const uint32_t *rn = pn + r * (STR / 2), *rx = px + r * (STR / 2);
#pragma unroll
        for (int c = 0; c < ND; c++) {
          uint32_t vn = rn[c], vx = rx[c];
          if (c == 0) {
            vn = pn1 ? ((vn & 0xffff0000u) | 0x00007fffu) : vn;
            vx = px1 ? ((vx & 0xffff0000u) | 0x00008000u) : vx;
          }
          if (c == ND - 1) {
            vn = pn1 ? vn : ((vn & 0x0000ffffu) | 0x7fff0000u);
            vx = px1 ? vx : ((vx & 0x0000ffffu) | 0x80000000u);
          }
          amn = __builtin_elementwise_min(amn, __builtin_bit_cast(vsm_s2, vn));
          amx = __builtin_elementwise_max(amx, __builtin_bit_cast(vsm_s2, vx));
        }
      }
    }
    wmn = min((int)amn.x, (int)amn.y);
    wmx = max((int)amx.x, (int)amx.y);
  } else {
    for (int r = l8; r < W; r += LANES) {
      if (mnj - N + r <= lim_j)
        for (int c = 0; c < W && mni - N + c <= lim_i; c++) wmn = min(wmn, (int)f[mnj - N + r][mni - N + c]);
      if (mxj - N + r <= lim_j)
        for (int c = 0; c < W && mxi - N + c <= lim_i; c++) wmx = max(wmx, (int)f[mxj - N + r][mxi - N + c]);
    }
  }
#pragma unroll
  for (int o = LANES / 2; o >= 1; o >>= 1) {
    wmn = min(wmn, __shfl_xor(wmn, o, LANES));
    wmx = max(wmx, __shfl_xor(wmx, o, LANES));
  }
  if (live && l8 == 0) {
    const bool vmin = (mnv <= -tau) && wmn >= mnv, vmax = (mxv >= tau) && wmx <= mxv;
    int32_t *c = st.cand + (size_t)(ci * st.ncv + cj) * 4 + 2 * k;
    c[0] = vmin ? (int32_t)(0x80000000u | (uint32_t)(mni + u0) | ((uint32_t)(mnj + v0) << 14)) : 0;
    c[1] = vmax ? (int32_t)(0x80000000u | (uint32_t)(mxi + u0) | ((uint32_t)(mxj + v0) << 14)) : 0;
  }
}

// ---------------------------------------------------------------------------------------
// The fused matching-resolution image side for the default suppression radii (dense n = 3, sparse n = 9): F1-F3 + N1 out of
// one LDS tile per workgroup, f1 / f2 never in HBM.  All per-thread work is in vsm_feat.h (shared with the CPU emulation
// of the "not gpu" tests); here: the phases.
//   k_feat_dense   tile = 128 x 48 owned pixels: image tile (+ 8 / 6 pixels of halo) -> LDS, 17 x 14 patches of 8 x 4 pixels,
//                  one per thread: du / dv (HBM), f1 / f2 (LDS), then 32 x 12 cells x 2 filters, a thread each
//   k_feat_sparse  tile = 16 x 4 cells of 10 x 10 pixels: the 19 x 19 windows want 11 pixels of halo, so this scale gets its
//                  own coarser tile and recomputes f1 / f2 (12 of 26 operations per pixel) rather than doubling the dense
//                  tile's filter work: 23 x 10 patches of 8 x 6 pixels; one response plane in LDS at a time (34 KB: four
//                  workgroups per compute unit), f2 waits in registers; 8 lanes per (cell, filter)
// XCD-aware placement: neighbouring tiles (which share halo lines) follow each other on one XCD's L2.
// ---------------------------------------------------------------------------------------


template <bool DUMP>
__global__ void __launch_bounds__(256)
    k_feat_dense(const VsmImage *__restrict__ imgs, int first, VsmDims d, int tau, int tiles_x, int nbx, int n_img,
                 int16_t *__restrict__ f1base, int16_t *__restrict__ f2base, size_t f_stride) {
  typedef VfDense G;
  __shared__ __attribute__((aligned(16))) uint32_t s_img[G::IH * G::IWD];
  __shared__ __attribute__((aligned(16))) int16_t s_f[2 * G::FH * G::FS];
  const int lb = xcd_remap(blockIdx.x, gridDim.x);
  const int zi = lb / nbx, bx = lb - zi * nbx;
  if (zi >= n_img) return;
  const VsmImage &im = imgs[first + zi];
  const VsmSet &st = im.set[1];
  const int ty = bx / tiles_x, tx = bx - ty * tiles_x;
  const int t = threadIdx.x;
  FT_DECL;
  FT_STAMP;
  vf_fill<G, 256>(s_img, im.imgm, d.mbpl * d.mh, d.mbpl, tx, ty, t);
  FT_STAMP;
  __syncthreads();
  FT_STAMP;
  if (t < G::PC * G::PR)
    vf_dense_patch(s_img, s_f, t, tx, ty, d.mbpl, d.mh, im.du, im.dv, DUMP ? f1base + (size_t)zi * f_stride : nullptr,
                   DUMP ? f2base + (size_t)zi * f_stride : nullptr);
  FT_STAMP;
  __syncthreads();
  FT_STAMP;
  for (int it = t; it < G::CU * G::CV * 2; it += 256) vf_dense_nms(s_f, it, tx, ty, d.mw, d.mh, VSM_MARGIN, tau, st.ncu, st.ncv, st.cand);
  FT_STAMP;
  FT_FLUSH(0);
}

__global__ void __launch_bounds__(256)
    k_feat_sparse(const VsmImage *__restrict__ imgs, int first, VsmDims d, int tau, int tiles_x, int nbx, int n_img) {
  typedef VfSparse G;
  __shared__ __attribute__((aligned(16))) uint32_t s_img[G::IH * G::IWD];
  __shared__ __attribute__((aligned(16))) int16_t s_f[G::FH * G::FS];
  const int lb = xcd_remap(blockIdx.x, gridDim.x);
  const int zi = lb / nbx, bx = lb - zi * nbx;
  if (zi >= n_img) return;
  const VsmImage &im = imgs[first + zi];
  const VsmSet &st = im.set[0];
  const int ty = bx / tiles_x, tx = bx - ty * tiles_x;
  const int t = threadIdx.x;
  FT_DECL;
  FT_STAMP;
  vf_fill<G, 256>(s_img, im.imgm, d.mbpl * d.mh, d.mbpl, tx, ty, t);
  FT_STAMP;
  __syncthreads();
  FT_STAMP;
  VfSparseKeep keep;
  if (t < G::PC * G::PR) vf_sparse_patch(s_img, s_f, t, keep);
  FT_STAMP;
  __syncthreads();
  FT_STAMP;
  static_assert((G::CU * G::CV) % 32 == 0, "whole rounds of 32 items x 8 lanes");
  for (int it = t >> 3; it < G::CU * G::CV; it += 32) vf_sparse_nms<8>(s_f, it, t & 7, 0, tx, ty, d.mw, d.mh, VSM_MARGIN, tau, st.ncu, st.ncv, st.cand);
  FT_STAMP;
  __syncthreads();
  if (t < G::PC * G::PR) vf_sparse_store_f2(s_f, t, keep);
  __syncthreads();
  FT_STAMP;
  for (int it = t >> 3; it < G::CU * G::CV; it += 32) vf_sparse_nms<8>(s_f, it, t & 7, 1, tx, ty, d.mw, d.mh, VSM_MARGIN, tau, st.ncu, st.ncv, st.cand);
  FT_STAMP;
  FT_FLUSH(1);
}

// tiles per image of the two kernels (the CPU emulation mirrors this: tests/emu/feat_emu.cpp)
static void vsm_feat_tiles(const VsmDims &d, const VsmSet &dense, int &dx, int &dy) {
  dx = std::max((d.mbpl + 8 + 127) / 128, (dense.ncu + 3 + 31) / 32);
  dy = std::max((d.mh + 4 + 47) / 48, (dense.ncv + 2 + 11) / 12);
}

// block-wide exclusive scan of one int per thread (blockDim.x == 1024); returns the exclusive
// prefix and the block total.  Wave shuffles + one LDS hop.
__device__ __forceinline__ int block_excl_scan_1024(int v, int &total, int *s_w /*[17]*/) {
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  int x = v;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    int y = __shfl_up(x, o, 64);
    if (lane >= o) x += y;
  }
  __syncthreads();  // protects s_w reuse across calls
  if (lane == 63) s_w[wv] = x;
  __syncthreads();
  if (wv == 0) {
    int w = lane < 16 ? s_w[lane] : 0;
#pragma unroll
    for (int o = 1; o < 16; o <<= 1) {
      int y = __shfl_up(w, o, 64);
      if (lane >= o) w += y;
    }
    if (lane < 16) s_w[lane] = w;  // inclusive wave totals
  }
  __syncthreads();
  total = s_w[15];
  int wbase = wv ? s_w[wv - 1] : 0;
  return wbase + x - v;
}

// ---------------------------------------------------------------------------------------
// Feature index = rank in the reference's emission order (cells u-major / v-minor, classes
// f1min,f1max,f2min,f2max inside a cell, viso/matcher.cpp:344-430): exclusive prefix sum of the
// per-cell survivor counts.  One 1024-thread block per (image, set); every thread owns a run of
// consecutive cells.
// ---------------------------------------------------------------------------------------
__global__ void __launch_bounds__(1024) k_scan_cells(const VsmImage *__restrict__ imgs, int first, int set_lo, int nb) {
  __shared__ int s_w[17];
  const VsmSet &st = imgs[first + blockIdx.z].set[blockIdx.y];
  if ((int)blockIdx.y < set_lo) {
    if (threadIdx.x == 0) {
      *st.count = 0;
      *st.count_host = 0;
    }
    return;
  }
  for (int b = threadIdx.x; b < nb; b += 1024) st.bin_cnt[b] = 0;  // histogram filled by k_emit
  const int ncells = st.ncu * st.ncv;
  const int chunk = (ncells + 1023) / 1024;
  const int c0 = min((int)threadIdx.x * chunk, ncells), c1 = min(c0 + chunk, ncells);
  int sum = 0, total, run;
  if (chunk <= 8) {  // (the usual case: all of a thread's cells requested at once, their counts kept for the second sweep)
    int cnt[8];
#pragma unroll
    for (int k = 0; k < 8; k++) {
      cnt[k] = 0;
      if (c0 + k < c1) {
        const int4 v = ldg_i4(st.cand + (size_t)(c0 + k) * 4);
        cnt[k] = (v.x < 0) + (v.y < 0) + (v.z < 0) + (v.w < 0);
      }
    }
#pragma unroll
    for (int k = 0; k < 8; k++) sum += cnt[k];
    run = block_excl_scan_1024(sum, total, s_w);
#pragma unroll
    for (int k = 0; k < 8; k++)
      if (c0 + k < c1) {
        st.cell_off[c0 + k] = run;
        run += cnt[k];
      }
  } else {
    for (int c = c0; c < c1; c++) {
      const int4 v = *(const int4 *)(st.cand + (size_t)c * 4);
      sum += (v.x < 0) + (v.y < 0) + (v.z < 0) + (v.w < 0);
    }
    run = block_excl_scan_1024(sum, total, s_w);
    for (int c = c0; c < c1; c++) {
      const int4 v = *(const int4 *)(st.cand + (size_t)c * 4);
      st.cell_off[c] = run;
      run += (v.x < 0) + (v.y < 0) + (v.z < 0) + (v.w < 0);
    }
  }
  if (threadIdx.x == 0) {
    st.cell_off[ncells] = total;
    *st.count = total;
    *st.count_host = total;
  }
}

// ---------------------------------------------------------------------------------------
// T1 feature records (viso/matcher.cpp:707-731) + D1 descriptor gather (computeDescriptor,
// viso/matcher.cpp:433-477).  16 lanes per cell: lane j < 12 produces dword j of the 48-byte
// record {u*s, v*s, 0, class, d1..d8} of each survivor of the cell -- a descriptor dword is two
// taps x (du,dv) = 4 byte gathers -- so every record leaves as one coalesced 48-byte run.
// ---------------------------------------------------------------------------------------
__constant__ int8_t c_desc_dv[16] = {-1, +1, -1, +1, -1, +1, -1, +1, -5, +5, -5, +5, -3, +3, -3, +3};
__constant__ int8_t c_desc_du[16] = {-3, -3, -1, -1, +3, +3, +1, +1, -1, -1, +1, +1, -5, -5, +5, +5};

// fine v row of a (non-negative) coordinate: v-bin * VSM_VSUB + sub-row inside the bin; monotonic in v
__device__ __forceinline__ int vfine_of(int v, int binsize, int vb) {
  const int vbin = min(v / binsize, vb - 1);
  return vbin * VSM_VSUB + min(((v - vbin * binsize) * VSM_VSUB) / binsize, VSM_VSUB - 1);
}

// fine bin id; id / VSM_VSUB is the reference's bin (class * ub + u_bin) * vb + v_bin (viso/matcher.cpp:881-888)
__device__ __forceinline__ int bin_of(int u, int v, int c, int binsize, int ub, int vb) {
  const int ubin = min(u / binsize, ub - 1);
  return (c * ub + ubin) * (vb * VSM_VSUB) + vfine_of(v, binsize, vb);
}

// the same with the division by the bin size as a multiply-high (cfg.bin_magic): k_match runs it several times per stage,
// and an integer division by a run-time value costs ~20 instructions.  Exact for 0 <= x < 2^32 / binsize; the arguments
// here are below 2^17 and the host refuses bin sizes above 32768.
__device__ __forceinline__ int div_bin(int x, const VsmMatchCfg &cfg) {
  return cfg.binsize == 1 ? x : (int)__umulhi((uint32_t)x, cfg.bin_magic);
}
__device__ __forceinline__ int vfine_fast(int v, const VsmMatchCfg &cfg, int vb) {
  const int vbin = min(div_bin(v, cfg), vb - 1);
  return vbin * VSM_VSUB + min(div_bin((v - vbin * cfg.binsize) * VSM_VSUB, cfg), VSM_VSUB - 1);
}

// Tile kernel: a block owns a rectangle of NMS cells (up to 128 x 32 matching-resolution pixels).
// It first stages the Sobel responses of that rectangle plus the 5-pixel descriptor halo in LDS,
// du and dv interleaved per pixel (coalesced dword row reads of both planes), then one thread per
// cell writes the records of the cell's survivors: the 16 taps of computeDescriptor are 16-bit LDS
// reads at compile-time offsets, already in the byte order of the record (du,dv of tap 0, du,dv of
// tap 1, ...; viso/matcher.cpp:445-476), and the 48 bytes leave as three 16-byte stores.
#define EMIT_TW 128
#define EMIT_TH 32
#define EMIT_HALO 5
#define EMIT_LW (EMIT_TW + 2 * EMIT_HALO + 4)  // + alignment slack of the row reads: 142 pixels
#define EMIT_LH (EMIT_TH + 2 * EMIT_HALO)
__global__ void __launch_bounds__(256) k_emit(const VsmImage *__restrict__ imgs, int first, VsmDims d, int set_lo,
                                              int binsize, int nbx, int n_img) {
  __shared__ uint16_t s_g[EMIT_LH][EMIT_LW + 2];  // du | dv << 8
  const int lb = xcd_remap(blockIdx.x, gridDim.x);
  const int zi = lb / nbx, bx = lb - zi * nbx;
  if (zi >= n_img) return;
  const VsmImage &im = imgs[first + zi];
  const int si = blockIdx.y;
  if (si < set_lo) return;
  const VsmSet &st = im.set[si];
  const int n1 = st.nms_n + 1;
  const int tcu = max(EMIT_TW / n1, 1), tcv = max(EMIT_TH / n1, 1);  // cells per tile
  const int tiles_u = (st.ncu + tcu - 1) / tcu, tiles_v = (st.ncv + tcv - 1) / tcv;
  if (bx >= tiles_u * tiles_v) return;
  const int tu = bx % tiles_u, tv = bx / tiles_u;
  const int cu0 = tu * tcu, cv0 = tv * tcv;
  // pixel rectangle covered by the descriptors of this tile's cells; cells wider than the tile
  // (n > 31) cannot be staged and take the direct path below
  const int ut = st.nms_n + VSM_MARGIN + cu0 * n1 - EMIT_HALO, vt = st.nms_n + VSM_MARGIN + cv0 * n1 - EMIT_HALO;
  const int ua = ut & ~3;
  const bool staged = n1 <= EMIT_TH;
  if (staged) {
    for (int e = threadIdx.x; e < (EMIT_LW / 4 + 1) * EMIT_LH; e += 256) {
      const int y = e / (EMIT_LW / 4 + 1), x = 4 * (e - y * (EMIT_LW / 4 + 1));
      const int u = ua + x, v = vt + y;
      if (x + 3 < EMIT_LW + 2) {
        uint32_t a = 0, b = 0;
        if (u >= 0 && u + 3 < d.mbpl && v >= 0 && v < d.mh) {
          a = ldg_u32(im.du + (size_t)v * d.mbpl + u);
          b = ldg_u32(im.dv + (size_t)v * d.mbpl + u);
        }
        // interleave: pixel p -> du_p | dv_p << 8
        const uint32_t lo = (a & 0xffu) | ((b & 0xffu) << 8) | ((a & 0xff00u) << 8) | ((b & 0xff00u) << 16);
        const uint32_t hi = ((a >> 16) & 0xffu) | (((b >> 16) & 0xffu) << 8) | ((a >> 24) << 16) | ((b >> 24) << 24);
        *(uint2 *)&s_g[y][x] = make_uint2(lo, hi);
      }
    }
    __syncthreads();
  }
  for (int t = threadIdx.x; t < tcu * tcv; t += 256) {
    const int lcv = t % tcv, lcu = t / tcv;  // v fastest: neighbouring threads own neighbouring cand[] entries
    const int ci = cu0 + lcu, cj = cv0 + lcv;
    if (ci >= st.ncu || cj >= st.ncv) continue;
    const int cell = ci * st.ncv + cj;
    const int4 c4 = ldg_i4(st.cand + (size_t)cell * 4);
    if ((c4.x | c4.y | c4.z | c4.w) >= 0) continue;
    const int cc4[4] = {c4.x, c4.y, c4.z, c4.w};
    int pos = ldg_i32(st.cell_off + cell);
#pragma unroll
    for (int g = 0; g < 4; g++) {
      const int cc = cc4[g];
      if (cc >= 0) continue;
      const int u = cc & 0x3fff, v = (cc >> 14) & 0x3fff;
      uint32_t t16[16];
      if (staged) {
        const uint16_t *c0 = &s_g[v - vt][u - ua];
#pragma unroll
        for (int m = 0; m < 16; m++) {
          constexpr int8_t kdv[16] = {-1, +1, -1, +1, -1, +1, -1, +1, -5, +5, -5, +5, -3, +3, -3, +3};
          constexpr int8_t kdu[16] = {-3, -3, -1, -1, +3, +3, +1, +1, -1, -1, +1, +1, -5, -5, +5, +5};
          t16[m] = c0[kdv[m] * (EMIT_LW + 2) + kdu[m]];
        }
      } else {
#pragma unroll
        for (int m = 0; m < 16; m++) {
          const int a = (v + c_desc_dv[m]) * d.mbpl + u + c_desc_du[m];
          t16[m] = (uint32_t)im.du[a] | ((uint32_t)im.dv[a] << 8);
        }
      }
      vsm_u4 r0, r1, r2;
      r0.x = (uint32_t)(u * d.scale);
      r0.y = (uint32_t)(v * d.scale);
      r0.z = 0u;
      r0.w = (uint32_t)g;
      r1.x = t16[0] | (t16[1] << 16);
      r1.y = t16[2] | (t16[3] << 16);
      r1.z = t16[4] | (t16[5] << 16);
      r1.w = t16[6] | (t16[7] << 16);
      r2.x = t16[8] | (t16[9] << 16);
      r2.y = t16[10] | (t16[11] << 16);
      r2.z = t16[12] | (t16[13] << 16);
      r2.w = t16[14] | (t16[15] << 16);
      VSM_AS1 vsm_u4 *rec = (VSM_AS1 vsm_u4 *)(st.feat + (size_t)pos * 12);
      rec[0] = r0;
      rec[1] = r1;
      rec[2] = r2;
      // M1 createIndexVector (viso/matcher.cpp:870-890): histogram of the search bins
      const int b = bin_of(u * d.scale, v * d.scale, g, binsize, d.ub, d.vb);
      st.binid[pos] = b;
      atomicAdd(&st.bin_cnt[b], 1);
      pos++;
    }
  }
}

// ---------------------------------------------------------------------------------------
// T1 + D1 + M1 in two kernels (the default path; k_scan_cells / k_emit / k_bin_* above and below remain for geometries it
// declines):
//   k_feat_scan   one workgroup per (image, set): ONE pass over the survivors gives the exclusive prefix of the cells'
//                 survivor counts (feature index = rank in the reference's emission order, viso/matcher.cpp:344-430) and,
//                 through a histogram in LDS, the start of every fine search bin (createIndexVector, :870-890)
//   k_feat_order  a workgroup owns a TILE OF WHOLE SEARCH BINS (bu coarse u-bins x one v-bin) of one image, both sets: it
//                 stages the Sobel responses of the tile's rectangle + the 5-pixel descriptor halo in LDS once, collects the
//                 survivors that fall into its bins from the cells overlapping the rectangle, and - because every feature
//                 of a bin is then in the workgroup - ranks them inside their fine bin and inside the reference's bin by
//                 counting smaller feature indices in a short LDS list per (bin, class).  Each survivor's 48-byte record
//                 goes to feat[index] and, in the same breath, its coordinates / descriptor / index / reference rank to
//                 their place in the bin-sorted arrays: no bin ids, scatter cursors or unordered slots in HBM, no second
//                 read of the records (k_bin_rank gathered them back: 2.5 x its algorithmic bytes).
// ---------------------------------------------------------------------------------------
// division by the search bin size as a multiply-high (exact for 0 <= x < 2^32 / binsize; the arguments are image coordinates)
struct VsmBinDiv {
  int32_t binsize;
  uint32_t magic;  // ceil(2^32 / binsize), 0 for binsize 1
  __device__ __forceinline__ int div(int x) const { return binsize == 1 ? x : (int)__umulhi((uint32_t)x, magic); }
  __device__ __forceinline__ int vsub(int v, int vbin) const { return min(div((v - vbin * binsize) * VSM_VSUB), VSM_VSUB - 1); }  // sub-row inside the bin
};
struct VsmOrderPlan {
  int32_t bu;                  // coarse u-bins per tile
  int32_t tiles_u, tiles_v;    // tiles per image (tiles_v = v-bins)
  int32_t stage_w, stage_h;    // staged Sobel responses: uint16 (du | dv << 8) per pixel
  int32_t cells_cap[2];        // cells of a set that can overlap a tile's rectangle
  int32_t cells_max;           // the larger of the two
  int32_t lcap[2];             // entries a (bin, class) list can take
  int32_t o_cand, o_bs, o_cnt, o_list, o_sv, lds_bytes;  // LDS layout (bytes)
};

__device__ __forceinline__ int ceil_div_pos(int a, int b) { return (a + b - 1) / b; }

__global__ void __launch_bounds__(1024) k_feat_scan(const VsmImage *__restrict__ imgs, int first, int set_lo, VsmDims d, VsmBinDiv bd, int nb) {
  extern __shared__ int s_dyn[];
  __shared__ int s_w[17];
  int *s_hist = s_dyn;
  const VsmSet &st = imgs[first + blockIdx.z].set[blockIdx.y];
  if ((int)blockIdx.y < set_lo) {
    if (threadIdx.x == 0) {
      *st.count = 0;
      *st.count_host = 0;
    }
    return;
  }
  FT_DECL;
  FT_STAMP;
  for (int b = threadIdx.x; b < nb; b += 1024) s_hist[b] = 0;
  __syncthreads();
  FT_STAMP;
  const int ncells = st.ncu * st.ncv;
  const int chunk = (ncells + 1023) / 1024;
  const int c0 = min((int)threadIdx.x * chunk, ncells), c1 = min(c0 + chunk, ncells);
  int sum = 0, total, run;
  auto tally = [&](const int4 &v) -> int {  // survivors of a cell into the histogram; returns their number
    const int cc[4] = {v.x, v.y, v.z, v.w};
    int n = 0;
#pragma unroll
    for (int g = 0; g < 4; g++)
      if (cc[g] < 0) {
        n++;
        const int u = (cc[g] & 0x3fff) * d.scale, v = ((cc[g] >> 14) & 0x3fff) * d.scale;
        const int ubin = min(bd.div(u), d.ub - 1), vbin = min(bd.div(v), d.vb - 1);
        atomicAdd(&s_hist[(g * d.ub + ubin) * (d.vb * VSM_VSUB) + vbin * VSM_VSUB + bd.vsub(v, vbin)], 1);  // = bin_of()
      }
    return n;
  };
  if (chunk <= 8) {  // (the usual case: all of a thread's cells requested at once, their counts kept for the second sweep)
    int4 v[8];
#pragma unroll
    for (int k = 0; k < 8; k++) v[k] = (c0 + k < c1) ? ldg_i4(st.cand + (size_t)(c0 + k) * 4) : make_int4(0, 0, 0, 0);
    int cnt[8];
#pragma unroll
    for (int k = 0; k < 8; k++) {
      cnt[k] = tally(v[k]);
      sum += cnt[k];
    }
    FT_STAMP;
    run = block_excl_scan_1024(sum, total, s_w);
    FT_STAMP;
#pragma unroll
    for (int k = 0; k < 8; k++)
      if (c0 + k < c1) {
        st.cell_off[c0 + k] = run;
        run += cnt[k];
      }
  } else {
    for (int c = c0; c < c1; c++) sum += tally(ldg_i4(st.cand + (size_t)c * 4));
    run = block_excl_scan_1024(sum, total, s_w);
    for (int c = c0; c < c1; c++) {
      const int4 v = ldg_i4(st.cand + (size_t)c * 4);  // (L2)
      st.cell_off[c] = run;
      run += (v.x < 0) + (v.y < 0) + (v.z < 0) + (v.w < 0);
    }
  }
  if (threadIdx.x == 0) {
    st.cell_off[ncells] = total;
    *st.count = total;
    *st.count_host = total;
  }
  // (block_excl_scan_1024's barriers also order the histogram's atomics before the reads below)
  const int bchunk = (nb + 1023) / 1024;
  const int b0 = min((int)threadIdx.x * bchunk, nb), b1 = min(b0 + bchunk, nb);
  int bsum = 0;
  for (int b = b0; b < b1; b++) bsum += s_hist[b];
  int btotal;
  FT_STAMP;
  int brun = block_excl_scan_1024(bsum, btotal, s_w);
  FT_STAMP;
  for (int b = b0; b < b1; b++) {
    st.bin_start[b] = brun;
    brun += s_hist[b];
  }
  if (threadIdx.x == 0) st.bin_start[nb] = btotal;
  FT_STAMP;
  FT_FLUSH(3);
}

__global__ void __launch_bounds__(256) k_feat_order(const VsmImage *__restrict__ imgs, int first, VsmDims d, int set_lo, VsmBinDiv bd,
                                                    VsmOrderPlan pl, int nbx, int n_img) {
  const int binsize = bd.binsize;
  extern __shared__ __attribute__((aligned(16))) unsigned char s_raw[];
  uint16_t *s_g = (uint16_t *)s_raw;                      // [stage_h][stage_w]
  int4 *s_cand = (int4 *)(s_raw + pl.o_cand);            // [cells_max] the cells overlapping the rectangle (one set at a time)
  int *s_bs = (int *)(s_raw + pl.o_bs);                  // [2][bu * 4][VSM_VSUB + 1]: bin_start of the tile's fine bins, per set
  int *s_cnt = (int *)(s_raw + pl.o_cnt);                // [bu * 4] list lengths, [bu * 4] = number of survivors
  uint32_t *s_list = (uint32_t *)(s_raw + pl.o_list);    // [bu * 4][lcap]: index * 8 + v sub-row
  uint32_t *s_sv = (uint32_t *)(s_raw + pl.o_sv);        // survivors of the tile: feature index << 11 | cell slot * 4 + class
  const int lb = xcd_remap(blockIdx.x, gridDim.x);
  const int zi = lb / nbx, bx = lb - zi * nbx;
  if (zi >= n_img) return;
  const VsmImage &im = imgs[first + zi];
  const int tv = bx / pl.tiles_u, tu = bx - tv * pl.tiles_u;
  const int t = threadIdx.x;
  // the tile's bins and their pixel rectangle at the matching resolution (bin = min(coordinate * scale / binsize, bins - 1))
  const int ub0 = tu * pl.bu, ub1 = min(ub0 + pl.bu, d.ub);
  const int x_lo = ub0 == 0 ? 0 : min(ceil_div_pos(ub0 * binsize, d.scale), d.mw);
  const int x_hi = ub1 == d.ub ? d.mw : min(ceil_div_pos(ub1 * binsize, d.scale), d.mw);
  const int y_lo = tv == 0 ? 0 : min(ceil_div_pos(tv * binsize, d.scale), d.mh);
  const int y_hi = tv + 1 == d.vb ? d.mh : min(ceil_div_pos((tv + 1) * binsize, d.scale), d.mh);
  const int nl = pl.bu * 4;
  FT_DECL;
  FT_STAMP;
  // ---- the starts of the tile's fine bins (both sets) and the Sobel responses of the rectangle + 5 pixels of halo, du | dv << 8
  // per pixel.  (Requesting EVERY load of the workgroup up front, the responses in registers too, was tried: what the extra
  // registers and LDS cost in resident workgroups made the kernel slower, 108 -> 114-150 us per 220 images.) ----
  for (int si = set_lo; si < 2; si++)
    for (int e = t; e < nl * (VSM_VSUB + 1); e += 256) {
      const int l = e / (VSM_VSUB + 1), k = e - l * (VSM_VSUB + 1);
      const int ubin = min(ub0 + (l >> 2), d.ub - 1);
      s_bs[si * nl * (VSM_VSUB + 1) + e] = ldg_i32(im.set[si].bin_start + ((l & 3) * d.ub + ubin) * (d.vb * VSM_VSUB) + tv * VSM_VSUB + k);
    }
  // the cells of both sets that can hold a pixel of the rectangle (<= 2 per thread and set: cells_cap <= 512), requested now,
  // kept in registers until their set's turn: their wait is over by the time the responses below have arrived
  int cell_h[2], ncell[2];
  int4 pc[2][2];
  int po[2][2];
#pragma unroll
  for (int si = 0; si < 2; si++) {
    const VsmSet &st = im.set[si];
    const int n = st.nms_n, n1 = n + 1, o0 = n + VSM_MARGIN;  // cell c covers pixels o0 + c * n1 .. + n
    const int cu_lo = max(0, ceil_div_pos(max(x_lo - n - o0, 0), n1)), cu_hi = min(st.ncu - 1, x_hi - 1 >= o0 ? (x_hi - 1 - o0) / n1 : -1);
    const int cv_lo = max(0, ceil_div_pos(max(y_lo - n - o0, 0), n1)), cv_hi = min(st.ncv - 1, y_hi - 1 >= o0 ? (y_hi - 1 - o0) / n1 : -1);
    const int cw = cu_hi - cu_lo + 1, ch = cv_hi - cv_lo + 1;
    cell_h[si] = ch;
    ncell[si] = (si >= set_lo && cw > 0 && ch > 0) ? cw * ch : 0;  // <= cells_cap[si] by the plan
#pragma unroll
    for (int i = 0; i < 2; i++) {
      const int e = t + 256 * i;
      pc[si][i] = make_int4(0, 0, 0, 0);
      po[si][i] = 0;
      if (e < ncell[si]) {
        const int lcu = e / ch, lcv = e - lcu * ch;  // v fastest: neighbouring threads read neighbouring cand[] entries
        const int cell = (cu_lo + lcu) * st.ncv + cv_lo + lcv;
        pc[si][i] = ldg_i4(st.cand + (size_t)cell * 4);
        po[si][i] = ldg_i32(st.cell_off + cell);
      }
    }
  }
  const int xa = (x_lo - EMIT_HALO) & ~3, ya = y_lo - EMIT_HALO;  // (may be negative: nothing there is ever read)
  const int sw4 = pl.stage_w >> 2;
  for (int e = t; e < sw4 * pl.stage_h; e += 256) {
    const int y = e / sw4, x = 4 * (e - y * sw4);
    const int u = xa + x, v = ya + y;
    uint32_t a = 0, b = 0;
    if (u >= 0 && u + 3 < d.mbpl && v >= 0 && v < d.mh && u < x_hi + EMIT_HALO && v < y_hi + EMIT_HALO) {
      a = ldg_u32(im.du + (size_t)v * d.mbpl + u);
      b = ldg_u32(im.dv + (size_t)v * d.mbpl + u);
    }
    const uint32_t lo = (a & 0xffu) | ((b & 0xffu) << 8) | ((a & 0xff00u) << 8) | ((b & 0xff00u) << 16);
    const uint32_t hi = ((a >> 16) & 0xffu) | (((b >> 16) & 0xffu) << 8) | ((a >> 24) << 16) | ((b >> 24) << 24);
    *(uint2 *)&s_g[y * pl.stage_w + x] = make_uint2(lo, hi);
  }
#pragma unroll
  for (int si = 0; si < 2; si++) {
    if (si < set_lo) continue;
    const VsmSet &st = im.set[si];
    const int lcap = pl.lcap[si];
    const int4 *cand = s_cand;
    FT_STAMP;
    __syncthreads();  // (stage complete; the previous set's lists and cells are no longer read)
    if (t <= nl) s_cnt[t] = 0;
    __syncthreads();
    FT_STAMP;
#pragma unroll
    for (int i = 0; i < 2; i++) {
      const int e = t + 256 * i;
      if (e >= ncell[si]) continue;
      const int4 c4 = pc[si][i];
      s_cand[e] = c4;
      if ((c4.x | c4.y | c4.z | c4.w) >= 0) continue;
      const int cc[4] = {c4.x, c4.y, c4.z, c4.w};
      int idx = po[si][i];
#pragma unroll
      for (int g = 0; g < 4; g++) {
        if (cc[g] >= 0) continue;
        const int u = (cc[g] & 0x3fff) * d.scale, v = ((cc[g] >> 14) & 0x3fff) * d.scale;
        const int ubin = min(bd.div(u), d.ub - 1), vbin = min(bd.div(v), d.vb - 1);
        if (ubin >= ub0 && ubin < ub1 && vbin == tv) {
          const int vsub = bd.vsub(v, vbin);
          const int l = (ubin - ub0) * 4 + g;
          const int p = atomicAdd(&s_cnt[l], 1);
          if (p < lcap) s_list[l * lcap + p] = (uint32_t)idx * 8u + (uint32_t)vsub;
          const int q = atomicAdd(&s_cnt[nl], 1);
          s_sv[q] = ((uint32_t)idx << 11) | (uint32_t)(e * 4 + g);
        }
        idx++;
      }
    }
    FT_STAMP;
    __syncthreads();
    const int nsv = s_cnt[nl];
    for (int q = t; q < nsv; q += 256) {
      const uint32_t sv = s_sv[q];
      const int e = (int)((sv >> 2) & 511u), g = (int)(sv & 3u);
      const uint32_t idx = sv >> 11;
      const int4 c4 = cand[e];
      const int cc = g == 0 ? c4.x : (g == 1 ? c4.y : (g == 2 ? c4.z : c4.w));
      const int u = cc & 0x3fff, v = (cc >> 14) & 0x3fff;
      const int us = u * d.scale, vs = v * d.scale;
      const int ubin = min(bd.div(us), d.ub - 1), vbin = min(bd.div(vs), d.vb - 1);
      const int vsub = bd.vsub(vs, vbin);
      const int l = (ubin - ub0) * 4 + g;
      // ranks: smaller indices in the reference's bin (all of the list) and in the fine bin (same sub-row)
      const int cnt = min(s_cnt[l], lcap);
      const uint32_t *L = s_list + l * lcap;
      int rank = 0, crank = 0;
      for (int k = 0; k < cnt; k += 4) {  // (lists are 16-byte aligned and a multiple of four long; entries beyond cnt do not count)
        const uint4 en4 = *(const uint4 *)(L + k);
        const uint32_t en[4] = {en4.x, en4.y, en4.z, en4.w};
#pragma unroll
        for (int j = 0; j < 4; j++) {
          const int before = (k + j < cnt) && (en[j] >> 3) < idx;
          crank += before;
          rank += (before && (int)(en[j] & 7u) == vsub) ? 1 : 0;
        }
      }
      const int *bs = s_bs + (si * nl + l) * (VSM_VSUB + 1);  // bin_start of the fine bins (g, ubin, vbin, 0..)
      const int dst = bs[vsub] + rank, clo = bs[0];
      // descriptor (computeDescriptor, viso/matcher.cpp:433-477): 16 taps of (du, dv) from the staged rectangle
      const uint16_t *c0 = s_g + (v - ya) * pl.stage_w + (u - xa);
      uint32_t t16[16];
#pragma unroll
      for (int m = 0; m < 16; m++) {
        constexpr int8_t kdv[16] = {-1, +1, -1, +1, -1, +1, -1, +1, -5, +5, -5, +5, -3, +3, -3, +3};
        constexpr int8_t kdu[16] = {-3, -3, -1, -1, +3, +3, +1, +1, -1, -1, +1, +1, -5, -5, +5, +5};
        t16[m] = c0[kdv[m] * pl.stage_w + kdu[m]];
      }
      vsm_u4 r0, r1, r2;
      r0.x = (uint32_t)us;
      r0.y = (uint32_t)vs;
      r0.z = 0u;
      r0.w = (uint32_t)g;
      r1.x = t16[0] | (t16[1] << 16);
      r1.y = t16[2] | (t16[3] << 16);
      r1.z = t16[4] | (t16[5] << 16);
      r1.w = t16[6] | (t16[7] << 16);
      r2.x = t16[8] | (t16[9] << 16);
      r2.y = t16[10] | (t16[11] << 16);
      r2.z = t16[12] | (t16[13] << 16);
      r2.w = t16[14] | (t16[15] << 16);
      VSM_AS1 vsm_u4 *rec = (VSM_AS1 vsm_u4 *)(st.feat + (size_t)idx * 12);
      rec[0] = r0;
      rec[1] = r1;
      rec[2] = r2;
      VSM_AS1 vsm_u4 *sd = (VSM_AS1 vsm_u4 *)(st.s_desc + 2 * (size_t)dst);
      sd[0] = r1;
      sd[1] = r2;
      *(VSM_AS1 uint32_t *)(st.s_uv + dst) = (uint32_t)us | ((uint32_t)vs << 16);  // coordinates are < 16384
      *(VSM_AS1 int32_t *)(st.s_idx + dst) = (int32_t)idx;
      *(VSM_AS1 int32_t *)(st.s_rank + dst) = clo + crank;
    }
  }
  FT_STAMP;
  FT_FLUSH(2);
}

// ---------------------------------------------------------------------------------------
// M1 createIndexVector, viso/matcher.cpp:870-890, as a stable counting sort into the
// bin-contiguous SoA arrays (see VsmSet).  fine bin = (class*ub + u_bin)*(vb*VSM_VSUB) + v sub-row,
// so that the rows a query visits for one u_bin are one contiguous run.  Histogram: k_emit.  Then
//   k_bin_scan    exclusive scan of the histogram (one block per image/set)
//   k_bin_scatter every feature takes a slot of its bin (unordered, atomic cursor)
//   k_bin_rank    every slot finds its stable rank = number of smaller feature indices in its
//                 fine bin, and its place in the reference's coarser (u_bin, v_bin, index) order
//                 (a bin holds at most one feature per NMS cell and class: a few dozen), and
//                 writes the sorted coordinate / descriptor / index / reference-rank arrays
// ---------------------------------------------------------------------------------------
__global__ void __launch_bounds__(1024) k_bin_scan(const VsmImage *__restrict__ imgs, int first, int set_lo, int nb) {
  __shared__ int s_w[17];
  if ((int)blockIdx.y < set_lo) return;
  const VsmSet &st = imgs[first + blockIdx.z].set[blockIdx.y];
  const int chunk = (nb + 1023) / 1024;
  const int b0 = min((int)threadIdx.x * chunk, nb), b1 = min(b0 + chunk, nb);
  int sum = 0;
  for (int b = b0; b < b1; b++) sum += st.bin_cnt[b];
  int total;
  int run = block_excl_scan_1024(sum, total, s_w);
  for (int b = b0; b < b1; b++) {
    const int c = st.bin_cnt[b];
    st.bin_start[b] = run;
    st.bin_cnt[b] = run;  // becomes the scatter cursor
    run += c;
  }
  if (threadIdx.x == 0) st.bin_start[nb] = total;
}

__global__ void __launch_bounds__(256) k_bin_scatter(const VsmImage *__restrict__ imgs, int first, int set_lo) {
  if ((int)blockIdx.y < set_lo) return;
  const VsmSet &st = imgs[first + blockIdx.z].set[blockIdx.y];
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= *st.count) return;
  st.tmp[atomicAdd(&st.bin_cnt[st.binid[i]], 1)] = i;
}

__global__ void __launch_bounds__(256) k_bin_rank(const VsmImage *__restrict__ imgs, int first, int set_lo) {
  if ((int)blockIdx.y < set_lo) return;
  const VsmSet &st = imgs[first + blockIdx.z].set[blockIdx.y];
  const int p = blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= *st.count) return;
  const int idx = st.tmp[p];
  const int b = st.binid[idx];
  const int lo = st.bin_start[b], hi = st.bin_start[b + 1];
  // the reference's bin = the VSM_VSUB fine bins b - b % VSM_VSUB ..., contiguous in the sorted order
  const int cb = (b / VSM_VSUB) * VSM_VSUB;
  const int clo = st.bin_start[cb], chi = st.bin_start[cb + VSM_VSUB];
  int rank = 0, crank = 0;
  for (int q = clo; q < chi; q++) {
    const int before = st.tmp[q] < idx;
    crank += before;
    rank += (q >= lo && q < hi) ? before : 0;
  }
  const int dst = lo + rank;
  st.s_rank[dst] = clo + crank;
  const int32_t *rec = st.feat + (size_t)idx * 12;
  const int4 hd = *(const int4 *)rec;
  st.s_idx[dst] = idx;
  st.s_uv[dst] = (uint32_t)hd.x | ((uint32_t)hd.y << 16);  // coordinates are < 16384
  st.s_desc[2 * dst] = *(const uint4 *)(rec + 4);
  st.s_desc[2 * dst + 1] = *(const uint4 *)(rec + 8);
}

// ---------------------------------------------------------------------------------------
// M2/M3 findMatch + matching, viso/matcher.cpp:892-963 and :965-1153.
// A group of G lanes owns one query and walks the whole dependent chain (2 stages for flow /
// stereo, 4 for quad).  In each stage the lanes stride over the candidates of the fine bins the
// window touches (packed 4-byte coordinates, 16-byte loads; 32-byte descriptor reads only for
// in-window candidates), cost = v_sad_u8 x 8 (+ 4*sqrt(du^2+dv^2) in double when a prediction is
// active), and the winner is the lexicographic minimum of (cost, place in the reference's visiting
// order) over the group -- exactly the reference's "first minimum in (u_bin, v_bin, index) order"
// (:937-958).  Measured on MI355X the kernel is bound by instruction issue and dependent L2 round
// trips, not by bytes (SQ counters in profiles/): fewer visited candidates, a 3-instruction window
// test and judging candidates per lane rather than per visited slot are what made it faster.
// ---------------------------------------------------------------------------------------
#define VSM_NONE 0xffffffffu

__device__ __forceinline__ uint32_t sad32(const uint4 &a0, const uint4 &a1, const uint4 &b0, const uint4 &b1) {
  uint32_t s = __builtin_amdgcn_sad_u8(a0.x, b0.x, 0u);
  s = __builtin_amdgcn_sad_u8(a0.y, b0.y, s);
  s = __builtin_amdgcn_sad_u8(a0.z, b0.z, s);
  s = __builtin_amdgcn_sad_u8(a0.w, b0.w, s);
  s = __builtin_amdgcn_sad_u8(a1.x, b1.x, s);
  s = __builtin_amdgcn_sad_u8(a1.y, b1.y, s);
  s = __builtin_amdgcn_sad_u8(a1.z, b1.z, s);
  s = __builtin_amdgcn_sad_u8(a1.w, b1.w, s);
  return s;
}

// the feature a chain stage starts from: position, class and 32-byte descriptor, in registers
struct VsmQuery {
  uint32_t uv;  // u | v << 16
  int c;
  uint4 da, db;
  __device__ __forceinline__ int u() const { return (int)(uv & 0xffffu); }
  __device__ __forceinline__ int v() const { return (int)(uv >> 16); }
};

__device__ __forceinline__ VsmQuery load_query(const VsmSet &A, int i) {
  const int32_t *rec = A.feat + (size_t)i * 12;
  const int4 hd = ldg_i4(rec);
  VsmQuery q;
  q.uv = (uint32_t)hd.x | ((uint32_t)hd.y << 16);
  q.c = hd.w;
  q.da = ldg_u4(rec + 4);
  q.db = ldg_u4(rec + 8);
  return q;
}

// One findMatch (viso/matcher.cpp:892-963) for the query held in `q` against feature set B.
// Returns the winner's position in B's bin-sorted arrays (VSM_NONE if the window is empty) and
// REPLACES q by the winner (the lane that found it broadcasts coordinates + descriptor with
// width-G shuffles), so the next stage of the chain starts without going back to memory; the
// winner's feature index is only looked up once, at the end of the chain.  An empty window
// yields feature 0 of B like the reference (min_ind = 0, :898), class included.
#ifndef VSM_UVL
#define VSM_UVL 1  // 16-byte coordinate loads in flight per lane (two cost the registers of the fifth wave per SIMD: 71 -> 56 us for the first pass, 180 -> 175 for the second)
#endif
#ifndef VSM_MATCH_BLOCK
#define VSM_MATCH_BLOCK 256  // threads per block of k_match
#endif
#ifndef VSM_STEREO_BY_BIN
#define VSM_STEREO_BY_BIN 1  // the stereo-type stages (window = a few rows x the disparity range: 2-3 bins, a few candidates each) scan by bin also under prior boxes: -3.5 %
#endif
#ifndef VSM_MATCH_BALANCE
#define VSM_MATCH_BALANCE 0  // passes in which the lanes of a group even out their parked candidates before judging.  MEASURED with 2: judge rounds per wave 20.1 -> 16.6, time unchanged (alone 224-228 us either way): a wave's row of round trips is not what bounds the kernel - off
#endif
#ifndef VSM_SCAN_UNALIGNED
#define VSM_SCAN_UNALIGNED 1  // coordinate loads start at the run's first candidate (dword-aligned 16-byte loads) instead of at the 16-byte line below it
#endif
#ifdef VSM_MATCH_TIMING
extern __device__ unsigned long long vsm_mt_acc[16];
#endif
#if defined(VSM_MATCH_TIMING) && VSM_MATCH_TIMING == 2
#define VSM_MT_TRIP(k)                                                                       \
  do {                                                                                       \
    if ((int)(threadIdx.x & 63) == __ffsll((long long)__ballot(1)) - 1) atomicAdd(&vsm_mt_acc[k], 1ull); \
  } while (0)
#else
#define VSM_MT_TRIP(k)
#endif
typedef unsigned short vsm_us2 __attribute__((ext_vector_type(2)));

template <int G, bool RELOAD = true, bool MAYPRED = true, bool BYBIN = false, bool HEADS = false>
__device__ __forceinline__ uint32_t find_match(VsmQuery &q, const VsmSet &B, const VsmDims &d, const VsmMatchCfg &cfg,
                                               bool prior, float r_umin, float r_umax, float r_vmin, float r_vmax,
                                               bool flow, double u_, double v_, int lane, long long *ph = nullptr) {
  float u_min, u_max, v_min, v_max;
  const int qu = q.u(), qv = q.v();
  if (prior) {
    u_min = (float)qu + r_umin;
    u_max = (float)qu + r_umax;
    v_min = (float)qv + r_vmin;
    v_max = (float)qv + r_vmax;
  } else {
    u_min = (float)(qu - cfg.radius);
    u_max = (float)(qu + cfg.radius);
    v_min = (float)(qv - cfg.radius);
    v_max = (float)(qv + cfg.radius);
  }
  if (!flow) {
    v_min = (float)(qv - cfg.disp_tol);
    v_max = (float)(qv + cfg.disp_tol);
  }
  // The reference tests (float)u2 >= u_min && (float)u2 <= u_max (viso/matcher.cpp:943) on integer
  // coordinates: the same as lo <= u2 <= hi with lo = ceil(u_min), hi = floor(u_max).  Coordinates
  // are < 16384, so with both axes packed as 16-bit halves the whole window test is one wrapping
  // packed subtract, one packed min and one compare per candidate.
  const int lo_u = max((int)ceilf(u_min), 0), hi_u = min((int)floorf(u_max), 65535);
  const int lo_v = max((int)ceilf(v_min), 0), hi_v = min((int)floorf(v_max), 65535);
  const bool empty = hi_u < lo_u || hi_v < lo_v;
  // u-bins that can hold an in-window candidate: those of lo_u .. hi_u (a feature's bin is u / binsize, k_emit) - inside
  // the reference's floor(u_min / binsize) .. floor(u_max / binsize) (:929-932), and every candidate takes the exact window
  // test anyway; who wins a tie is settled by the candidates' ranks, not by the order of the visit
  const int ubmin = min(div_bin(min(lo_u, 65535), cfg), d.ub - 1);
  const int ubmax = min(div_bin(max(hi_u, 0), cfg), d.ub - 1);
  // fine rows that can hold an in-window candidate (a subset of the reference's v-bins vbmin..vbmax,
  // :933-934; every candidate still takes the exact window test below)
  const int vrows = d.vb * VSM_VSUB;
  const int vfmin = vfine_fast(min(lo_v, d.vb * cfg.binsize - 1), cfg, d.vb);
  const int vfmax = vfine_fast(min(max(hi_v, 0), d.vb * cfg.binsize - 1), cfg, d.vb);
  const uint32_t lo_pk = (uint32_t)lo_u | ((uint32_t)lo_v << 16);
  const uint32_t rng_pk = (uint32_t)(hi_u - lo_u) | ((uint32_t)(hi_v - lo_v) << 16);
  const bool pred = MAYPRED && (u_ >= 0 && v_ >= 0);
  // Two phases per stage.  (1) Walk the candidates: coordinates are packed (u | v << 16) and sorted
  // by fine bin, so one aligned 16-byte load brings 4 consecutive candidates of this lane
  // (VSM_UVL such loads in flight); the positions of the few that fall inside the window are parked
  // in a 4-deep per-lane register queue.  (2) Judge the parked candidates: descriptor + reference
  // rank fetch, SAD, and the double-precision distance term of a predicted match (:948-953) only
  // when the integer SAD alone does not already exceed the best cost (cost >= SAD).  A wavefront
  // runs phase 2 as many times as its busiest lane has candidates, not once per visited slot.
  // The reference keeps the FIRST minimum in its (u_bin, v_bin, index) visiting order (:937-958):
  // that is the minimum of (cost, rank), whatever order the candidates are judged in.
  // A stage that cannot have a prediction (MAYPRED = false) compares one integer key, SAD << 32 | rank; the others keep
  // the cost in double as the reference does.  The updates are selects, not branches.
  // (Round 4 tried the judging spread over the wave instead - the lanes' parked candidates compacted onto one list per wave
  // in LDS by ballots, 64 entries judged per round whoever found them, the owner's descriptor by cross-lane reads, the
  // minimum of (cost, rank) per query by ds_min_u64: results identical, but a wave-stage has 56 candidates on average
  // (1.74 per query), so four rounds become two, and the list's bookkeeping costs more than that: 345-370 us against 322.)
  double best = 10000000.0;
  uint64_t bkey = ~0ull;
  uint32_t bestq = VSM_NONE, brank = VSM_NONE;
  int nq = 0, q0p = 0, q1p = 0, q2p = 0, q3p = 0;
  auto judge = [&](int p) {
    VSM_MT_TRIP(2);
    const uint4 a = ldg_u4_at(B.s_desc, (uint32_t)p * 32u), b = ldg_u4_at(B.s_desc, (uint32_t)p * 32u + 16u);
    const uint32_t rk = ldg_u32_at(B.s_rank, (uint32_t)p * 4u);
    const uint32_t sad = sad32(q.da, q.db, a, b);
    if (!MAYPRED) {
      const uint64_t key = ((uint64_t)sad << 32) | rk;
      const bool better = key < bkey;
      bkey = better ? key : bkey;
      bestq = better ? (uint32_t)p : bestq;
    } else {
      double cost = (double)sad;
      if (cost <= best) {
        if (pred) {
          const uint32_t w = ldg_u32_at(B.s_uv, (uint32_t)p * 4u);
          double du = (double)(int)(w & 0xffffu) - u_;
          double dv = (double)(int)(w >> 16) - v_;
          double dist = sqrt(du * du + dv * dv);
          cost += 4 * dist;
        }
        const bool better = cost < best || (cost == best && rk < brank);
        best = better ? cost : best;
        brank = better ? rk : brank;
        bestq = better ? (uint32_t)p : bestq;
      }
    }
  };
  auto pop_and_judge = [&]() {  // lanes with a parked candidate take their newest one
    if (nq > 0) {
      const int p = q0p;
      q0p = q1p;
      q1p = q2p;
      q2p = q3p;
      nq--;
      judge(p);
    }
  };
  VSM_MT_TRIP(3);
#if defined(VSM_MATCH_TIMING) && VSM_MATCH_TIMING == 1
  const long long ph0 = clock64();
#endif
#if defined(VSM_MATCH_TIMING) && VSM_MATCH_TIMING == 2
  int mt_maxrun = 0;  // the longest run of candidates this lane's scans went through (the bound of a per-bin head record)
#define VSM_MT_RUN(len) mt_maxrun = max(mt_maxrun, (int)(len))
#else
#define VSM_MT_RUN(len)
#endif
  if (HEADS) {
  // The window's u-bins one after the other; of a bin's head record (k_feat_heads: 64 bytes = the run's start + the first 15
  // candidates' coordinates) every lane of the group loads its 16 / G dwords, next to the run's end: one round trip for
  // what the forms below take two or more for (bin starts, then coordinate loads that need them).
  constexpr int NDW = 16 / G;  // record dwords per lane
  auto park = [&](uint32_t w, int p, int q1) {
    const vsm_us2 off = __builtin_bit_cast(vsm_us2, w) - __builtin_bit_cast(vsm_us2, lo_pk);
    const vsm_us2 cl = __builtin_elementwise_min(off, __builtin_bit_cast(vsm_us2, rng_pk));
    if (__builtin_bit_cast(uint32_t, cl) == __builtin_bit_cast(uint32_t, off) && p < q1) {
      if (nq == 4) {  // queue full (rare): make room first
        const int pf = q3p;
        nq = 3;
        judge(pf);
      }
      q3p = q2p;
      q2p = q1p;
      q1p = q0p;
      q0p = p;
      nq++;
    }
  };
  for (int ubin = ubmin; ubin <= ubmax && !empty; ubin++) {
    VSM_MT_TRIP(0);
    const int b0 = (q.c * d.ub + ubin) * vrows;
    const uint32_t hb = (uint32_t)(b0 + vfmin) * 64u + (uint32_t)lane * (uint32_t)(4 * NDW);
    uint32_t r[NDW];
    if (NDW >= 4) {
#pragma unroll
      for (int j = 0; j < NDW / 4; j++) {
        const uint4 v = ldg_u4_at(B.heads, hb + 16u * j);
        r[4 * j + 0] = v.x;
        r[4 * j + 1] = v.y;
        r[4 * j + 2] = v.z;
        r[4 * j + 3] = v.w;
      }
    } else {
#pragma unroll
      for (int j = 0; j < NDW; j++) r[j] = ldg_u32_at(B.heads, hb + 4u * j);
    }
    const int q1 = (int)ldg_u32_at(B.bin_start, (uint32_t)(b0 + vfmax + 1) * 4u);
    const int q0 = __shfl((int)r[0], 0, G);  // (the group's lane 0 holds the record's first dword: the start)
    VSM_MT_RUN(q1 - q0);
#pragma unroll
    for (int i = 0; i < NDW; i++) {
      const int k = lane * NDW + i - 1;  // candidate number of this dword (-1: the start itself)
      if (i > 0 || lane > 0) park(r[i], q0 + k, q1);
    }
    for (int p0 = q0 + 15 + 4 * lane; p0 < q1; p0 += 4 * G) {  // a run of more than 15 candidates: the rest in 16-byte loads
      VSM_MT_TRIP(1);
      const uint4 wk = ldg_u4_at_dw(B.s_uv, (uint32_t)p0 * 4u);
      park(wk.x, p0, q1);
      park(wk.y, p0 + 1, q1);
      park(wk.z, p0 + 2, q1);
      park(wk.w, p0 + 3, q1);
    }
  }
  } else if (BYBIN) {
  // The lanes of a group take the window's u-bins in turn, each scanning its bin's run alone: a stereo stage's disparity
  // range spans 2-3 bins and an unconstrained first-pass window nine, with a handful of candidates in each - the wave goes
  // round ceil(bins / G) times instead of once per bin with most of a 16- or 32-slot sweep empty.  (BYBIN: the launches without prior
  // boxes - 88 -> 70 us per 67 pairs; with them the windows are narrow and sharing a bin's run is 2 % quicker.)
  for (int ubin = ubmin + lane; ubin <= ubmax && !empty; ubin += G) {
    VSM_MT_TRIP(0);
    const int b0 = (q.c * d.ub + ubin) * vrows;
    const int q0 = (int)ldg_u32_at(B.bin_start, (uint32_t)(b0 + vfmin) * 4u), q1 = (int)ldg_u32_at(B.bin_start, (uint32_t)(b0 + vfmax + 1) * 4u);
    VSM_MT_RUN(q1 - q0);
    for (int p0 = q0; p0 < q1; p0 += 4 * VSM_UVL) {
      VSM_MT_TRIP(1);
      uint4 wk[VSM_UVL];
#pragma unroll
      for (int j = 0; j < VSM_UVL; j++) {
        const int pj = p0 + j * 4;
        wk[j] = pj < q1 ? ldg_u4_at_dw(B.s_uv, (uint32_t)pj * 4u) : make_uint4(0, 0, 0, 0);
      }
#pragma unroll
      for (int j = 0; j < VSM_UVL; j++) {
        const uint32_t w4[4] = {wk[j].x, wk[j].y, wk[j].z, wk[j].w};
#pragma unroll
        for (int k = 0; k < 4; k++) {
          const int p = p0 + j * 4 + k;
          const vsm_us2 off = __builtin_bit_cast(vsm_us2, w4[k]) - __builtin_bit_cast(vsm_us2, lo_pk);
          const vsm_us2 cl = __builtin_elementwise_min(off, __builtin_bit_cast(vsm_us2, rng_pk));
          if (__builtin_bit_cast(uint32_t, cl) == __builtin_bit_cast(uint32_t, off) && p < q1) {
            if (nq == 4) {  // queue full (rare): make room first
              const int pf = q3p;
              nq = 3;
              judge(pf);
            }
            q3p = q2p;
            q2p = q1p;
            q1p = q0p;
            q0p = p;
            nq++;
          }
        }
      }
    }
  }
  } else {
  for (int ubin = ubmin; ubin <= ubmax && !empty; ubin++) {
    VSM_MT_TRIP(0);
    const int b0 = (q.c * d.ub + ubin) * vrows;
    const int q0 = (int)ldg_u32_at(B.bin_start, (uint32_t)(b0 + vfmin) * 4u), q1 = (int)ldg_u32_at(B.bin_start, (uint32_t)(b0 + vfmax + 1) * 4u);
    VSM_MT_RUN(q1 - q0);
#if VSM_SCAN_UNALIGNED
    for (int p0 = q0 + 4 * lane; p0 < q1; p0 += 4 * G * VSM_UVL) {  // (the run's first candidate first: only the tail needs a bound)
#else
    const uint32_t qn = (uint32_t)(q1 - q0);
    for (int p0 = (q0 & ~3) + 4 * lane; p0 < q1; p0 += 4 * G * VSM_UVL) {
#endif
      VSM_MT_TRIP(1);
      uint4 wk[VSM_UVL];
#pragma unroll
      for (int j = 0; j < VSM_UVL; j++) {
        const int pj = p0 + j * 4 * G;
#if VSM_SCAN_UNALIGNED
        wk[j] = pj < q1 ? ldg_u4_at_dw(B.s_uv, (uint32_t)pj * 4u) : make_uint4(0, 0, 0, 0);
#else
        wk[j] = pj < q1 ? ldg_u4_at(B.s_uv, (uint32_t)pj * 4u) : make_uint4(0, 0, 0, 0);
#endif
      }
#pragma unroll
      for (int j = 0; j < VSM_UVL; j++) {
        const uint32_t w4[4] = {wk[j].x, wk[j].y, wk[j].z, wk[j].w};
#pragma unroll
        for (int k = 0; k < 4; k++) {
          const int p = p0 + j * 4 * G + k;
          const vsm_us2 off = __builtin_bit_cast(vsm_us2, w4[k]) - __builtin_bit_cast(vsm_us2, lo_pk);
          const vsm_us2 cl = __builtin_elementwise_min(off, __builtin_bit_cast(vsm_us2, rng_pk));
#if VSM_SCAN_UNALIGNED
          if (__builtin_bit_cast(uint32_t, cl) == __builtin_bit_cast(uint32_t, off) && p < q1) {
#else
          if (__builtin_bit_cast(uint32_t, cl) == __builtin_bit_cast(uint32_t, off) && (uint32_t)(p - q0) < qn) {
#endif
            if (nq == 4) {  // queue full (rare): make room first
              const int pf = q3p;
              nq = 3;
              judge(pf);
            }
            q3p = q2p;
            q2p = q1p;
            q1p = q0p;
            q0p = p;
            nq++;
          }
        }
      }
    }
  }
  }
#if defined(VSM_MATCH_TIMING) && VSM_MATCH_TIMING == 1
  const long long ph1 = clock64();
#endif
#if defined(VSM_MATCH_TIMING) && VSM_MATCH_TIMING == 2
  if (!cfg.sparse) {  // dense pass: in how many wave-stages would a head record of 3 / 7 / 15 inline candidates have spared EVERY lane its coordinate loads?
    if (BYBIN) {  // (a lane scans its own bins: the record is one lane's)
      VSM_MT_TRIP(11);
      if (!__any(mt_maxrun > 3)) VSM_MT_TRIP(12);
      if (!__any(mt_maxrun > 7)) VSM_MT_TRIP(13);
      if (!__any(mt_maxrun > 15)) VSM_MT_TRIP(14);
    } else {      // (a group shares a bin's run)
      VSM_MT_TRIP(8);
      if (!__any(mt_maxrun > 7)) VSM_MT_TRIP(9);
      if (!__any(mt_maxrun > 15)) VSM_MT_TRIP(10);
    }
  }
#endif
#if VSM_MATCH_BALANCE
  // The wave judges as many rounds as its busiest lane has parked candidates (a descriptor fetch each: a round trip), and
  // who judges a candidate does not matter - the group's minimum of (cost, rank) is taken below.  So the lanes of a group
  // even their queues out first: a lane with two candidates more than its partner hands its newest one over.
  if (G >= 2) {
#pragma unroll
    for (int rep = 0; rep < VSM_MATCH_BALANCE; rep++) {
#pragma unroll
      for (int m = 1; m < G; m <<= 1) {
        const int onq = __shfl_xor(nq, m, G);
        const int sent = __shfl_xor(q0p, m, G);
        const bool give = nq > onq + 1, take = onq > nq + 1;
        if (give) {
          q0p = q1p;
          q1p = q2p;
          q2p = q3p;
          nq--;
        }
        if (take) {
          q3p = q2p;
          q2p = q1p;
          q1p = q0p;
          q0p = sent;
          nq++;
        }
      }
    }
  }
#endif
  while (__any(nq > 0)) pop_and_judge();
#if defined(VSM_MATCH_TIMING) && VSM_MATCH_TIMING == 1
  const long long ph2 = clock64();
  if (ph) {
    ph[0] += ph1 - ph0;
    ph[1] += ph2 - ph1;
  }
#endif
#pragma unroll
  for (int m = G / 2; m >= 1; m >>= 1) {
    const uint32_t oq = (uint32_t)__shfl_xor((int)bestq, m, G);
    if (!MAYPRED) {
      const uint32_t olo = (uint32_t)__shfl_xor((int)(uint32_t)bkey, m, G), ohi = (uint32_t)__shfl_xor((int)(uint32_t)(bkey >> 32), m, G);
      const uint64_t ok = ((uint64_t)ohi << 32) | olo;
      const bool better = ok < bkey;
      bkey = better ? ok : bkey;
      bestq = better ? oq : bestq;
    } else {
      const double oc = __shfl_xor(best, m, G);
      const uint32_t ork = (uint32_t)__shfl_xor((int)brank, m, G);
      const bool better = oc < best || (oc == best && ork < brank);
      best = better ? oc : best;
      bestq = better ? oq : bestq;
      brank = better ? ork : brank;
    }
  }
  if (!RELOAD) return bestq;
  if (bestq == VSM_NONE) {  // group-uniform
    q = load_query(B, 0);
    return VSM_NONE;
  }
  // every lane of the group fetches the winner's record (just touched, so it is in cache; handing it over from the lane
  // that judged it costs 20 registers and was measured 3 % quicker on pass 2, 15 % slower on pass 1; fetched by ONE lane and
  // passed on in nine shuffles - round 5 - 251 against 226 us alone: the shuffles and 32 bytes of spills cost more than the
  // lane accesses they save)
  q.uv = ldg_u32_at(B.s_uv, bestq * 4u);
  q.da = ldg_u4_at(B.s_desc, bestq * 32u);
  q.db = ldg_u4_at(B.s_desc, bestq * 32u + 16u);
  return bestq;
}

// (floor((float)u / (float)binsize) of the reference, :1020-1022, is u / binsize for these integers: u < 2^14)
__device__ __forceinline__ int stat_bin_of(int u, int v, const VsmMatchCfg &cfg, int ub, int vb) {
  return min(div_bin(v, cfg), vb - 1) * ub + min(div_bin(u, cfg), ub - 1);
}

__device__ __forceinline__ int index_of(const VsmSet &B, uint32_t pos) { return pos == VSM_NONE ? 0 : ldg_i32(B.s_idx + pos); }

#ifndef VSM_MATCH_WAVES
#define VSM_MATCH_WAVES 5  // waves per SIMD the register allocator must leave room for (96 registers, no scratch; six would spill)
#endif
#ifdef VSM_MATCH_TIMING  // experiments (tools/build_variant.sh NAME -DVSM_MATCH_TIMING): life of every wave of the dense pass
__device__ unsigned long long vsm_mt_acc[16];  // wave-level trip counts: [0] ubin iterations, [1] scan iterations, [2] judge rounds, [3] findMatch calls, [4..7] cycles of stage 1..4
extern "C" int vsm_debug_match_acc(unsigned long long *out, int reset) {
  if (hipMemcpyFromSymbol(out, HIP_SYMBOL(vsm_mt_acc), sizeof(vsm_mt_acc)) != hipSuccess) return -1;
  if (reset) {
    static unsigned long long z[16];
    if (hipMemcpyToSymbol(HIP_SYMBOL(vsm_mt_acc), z, sizeof(z)) != hipSuccess) return -1;
  }
  return 0;
}
__device__ unsigned int vsm_mt_n;
__device__ unsigned int vsm_mt[1 << 18][8];  // life, stage 1..4, bins + scan, judging, start (low bits)
extern "C" int vsm_debug_match_timing(unsigned int *out, unsigned int cap, int reset) {
  unsigned int n = 0;
  if (hipMemcpyFromSymbol(&n, HIP_SYMBOL(vsm_mt_n), 4) != hipSuccess) return -1;
  if (n > (1u << 18)) n = 1u << 18;
  if (n > cap) n = cap;
  if (n && hipMemcpyFromSymbol(out, HIP_SYMBOL(vsm_mt), (size_t)n * 32) != hipSuccess) return -1;
  if (reset) {
    const unsigned int z = 0;
    if (hipMemcpyToSymbol(HIP_SYMBOL(vsm_mt_n), &z, 4) != hipSuccess) return -1;
  }
  return (int)n;
}
#endif
template <int G, bool BYBIN = false, bool HEADS = false>
__global__ void __launch_bounds__(VSM_MATCH_BLOCK, VSM_MATCH_WAVES)
    k_match(const VsmImage *__restrict__ imgs, const VsmPair *__restrict__ pairs, const VsmJob *__restrict__ jobs,
            VsmJob job0, VsmDims d, VsmMatchCfg cfg, int nbx, int npairs) {
  // flattened grid: logical block -> (frame pair, block within pair), XCD-contiguous
  // (jobs == nullptr: the single pair `job0`)
  const int lb = xcd_remap(blockIdx.x, gridDim.x);
  const int pj = lb / nbx, bx = lb - pj * nbx;
  if (pj >= npairs) return;
  const VsmJob &jb = jobs ? jobs[pj] : job0;
  const VsmPair &pair = pairs[pj];
  const int lane = threadIdx.x & (G - 1);
  const int qi = (bx * blockDim.x + threadIdx.x) / G;
  const int si = cfg.sparse ? 0 : 1;
  if (qi >= jb.nq[si]) return;
#ifdef VSM_MATCH_TIMING
  const long long mt0 = clock64();
#if VSM_MATCH_TIMING == 3
  const long long mtw0 = wall_clock64();
#endif
  long long mtph_[2] = {0, 0}, *mtph = mtph_;
  unsigned int mtst[4] = {0, 0, 0, 0};
#else
  long long *mtph = nullptr;
#endif
  const int img_prev = jb.img_prev, img_curr = jb.img_curr;
  const VsmSet &s1p = imgs[img_prev].set[si], &s2p = imgs[img_prev + 1].set[si];
  const VsmSet &s1c = imgs[img_curr].set[si], &s2c = imgs[img_curr + 1].set[si];
  const bool prior = cfg.use_prior != 0;
  vsm_p_match m;
  bool ok = false;
  // the statistics bin of a chain is that of its start feature (:1020-1022, :1104-1106); its four
  // per-stage boxes are fetched once
  VsmQuery q = load_query(cfg.method == 2 ? s1p : s1c, qi);
  // (stage-major on the device: one 16-byte load per stage, issued one stage ahead of its use)
  const float *rg = pair.ranges + 16 * stat_bin_of(q.u(), q.v(), cfg, d.ub, d.vb);
  auto box = [&](int stage) {  // {u_min, u_max, v_min, v_max} offsets of a stage
    if (!prior) return make_float4(0, 0, 0, 0);
    const uint4 r = ldg_u4(rg + 4 * stage);
    return make_float4(__uint_as_float(r.x), __uint_as_float(r.y), __uint_as_float(r.z), __uint_as_float(r.w));
  };
  const uint32_t w0 = q.uv;
  const int u0 = q.u(), v0 = q.v();
  if (cfg.method == 0) {  // flow, :1006-1041
    const float4 r0 = box(0), r1 = box(1);
    const uint32_t p1 = find_match<G, true, false, BYBIN, HEADS>(q, s1p, d, cfg, prior, r0.x, r0.y, r0.z, r0.w, true, -1, -1, lane);
    const int u1p = q.u(), v1p = q.v();
    const uint32_t p2 = find_match<G, true, false, BYBIN, HEADS>(q, s1c, d, cfg, prior, r1.x, r1.y, r1.z, r1.w, true, -1, -1, lane);
    const int i1p = index_of(s1p, p1), i1c2 = index_of(s1c, p2);
    ok = (i1c2 == qi);
    m = {(float)u1p, (float)v1p, i1p, -1.f, -1.f, -1, (float)u0, (float)v0, qi, -1.f, -1.f, -1};
  } else if (cfg.method == 1) {  // stereo, :1045-1084
    const float4 r0 = box(0), r1 = box(1);
    const uint32_t p1 = find_match<G, true, false, BYBIN || VSM_STEREO_BY_BIN, HEADS>(q, s2c, d, cfg, prior, r0.x, r0.y, r0.z, r0.w, false, -1, -1, lane);
    const int u2c = q.u(), v2c = q.v();
    const uint32_t p2 = find_match<G, true, false, BYBIN || VSM_STEREO_BY_BIN, HEADS>(q, s1c, d, cfg, prior, r1.x, r1.y, r1.z, r1.w, false, -1, -1, lane);
    const int i2c = index_of(s2c, p1), i1c2 = index_of(s1c, p2);
    ok = (i1c2 == qi) && (u0 >= u2c);
    m = {-1.f, -1.f, -1, -1.f, -1.f, -1, (float)u0, (float)v0, qi, (float)u2c, (float)v2c, i2c};
  } else {  // quad, :1088-1153
    // (stage results stay packed u | v << 16 until the record is written: registers decide how many
    // chains a SIMD keeps in flight)
    const float4 r0 = box(0), r1 = box(1);
    const uint32_t p1 = find_match<G, true, false, BYBIN || VSM_STEREO_BY_BIN, HEADS>(q, s2p, d, cfg, prior, r0.x, r0.y, r0.z, r0.w, false, -1, -1, lane, mtph);
    const uint32_t w2p = q.uv;
#ifdef VSM_MATCH_TIMING
    const long long ms1 = clock64();
#endif
    double u2c_ = -1, v2c_ = -1;
    if (jb.use_tr) {  // :1114-1126, contraction-free double arithmetic
      double dd = (double)u0 - (double)q.u();
      if (!(dd > 1.0)) dd = 1.0;
      double x1p = ((double)u0 - cfg.cu) * cfg.base / dd;
      double y1p = ((double)v0 - cfg.cv) * cfg.base / dd;
      double z1p = cfg.f * cfg.base / dd;
      double x2c = jb.t[0] * x1p + jb.t[1] * y1p + jb.t[2] * z1p + jb.t[3] - cfg.base;
      double y2c = jb.t[4] * x1p + jb.t[5] * y1p + jb.t[6] * z1p + jb.t[7];
      double z2c = jb.t[8] * x1p + jb.t[9] * y1p + jb.t[10] * z1p + jb.t[11];
      u2c_ = cfg.f * x2c / z2c + cfg.cu;
      v2c_ = cfg.f * y2c / z2c + cfg.cv;
    }
    const float4 r2 = box(2);
    const uint32_t p2 = find_match<G, true, true, BYBIN, HEADS>(q, s2c, d, cfg, prior, r1.x, r1.y, r1.z, r1.w, true, u2c_, v2c_, lane, mtph);
    const uint32_t w2c = q.uv;
#ifdef VSM_MATCH_TIMING
    const long long ms2 = clock64();
#endif
    const float4 r3 = box(3);
    const uint32_t p3 = find_match<G, true, false, BYBIN || VSM_STEREO_BY_BIN, HEADS>(q, s1c, d, cfg, prior, r2.x, r2.y, r2.z, r2.w, false, -1, -1, lane, mtph);
    const uint32_t w1c = q.uv;
#ifdef VSM_MATCH_TIMING
    const long long ms3 = clock64();
#endif
    // stage 4 predicts the chain's own start (:1134)
    const uint32_t p4 = find_match<G, true, true, BYBIN, HEADS>(q, s1p, d, cfg, prior, r3.x, r3.y, r3.z, r3.w, true,
                                      jb.use_tr ? (double)(int)(w0 & 0xffffu) : -1.0,
                                      jb.use_tr ? (double)(int)(w0 >> 16) : -1.0, lane, mtph);
    const int i1p2 = index_of(s1p, p4);
#ifdef VSM_MATCH_TIMING
    if (!cfg.sparse && (threadIdx.x & 63) == 0) {
      const long long ms4 = clock64();
      mtst[0] = (unsigned int)(ms1 - mt0);
      mtst[1] = (unsigned int)(ms2 - ms1);
      mtst[2] = (unsigned int)(ms3 - ms2);
      mtst[3] = (unsigned int)(ms4 - ms3);
    }
#endif
    const int u2p = (int)(w2p & 0xffffu), u2c = (int)(w2c & 0xffffu), u1c = (int)(w1c & 0xffffu);
    ok = (i1p2 == qi) && (u0 >= u2p) && (u1c >= u2c);
    if (ok)
      m = {(float)u0, (float)v0, qi, (float)u2p, (float)(int)(w2p >> 16), index_of(s2p, p1), (float)u1c,
           (float)(int)(w1c >> 16), index_of(s1c, p3), (float)u2c, (float)(int)(w2c >> 16), index_of(s2c, p2)};
  }
  if (lane == 0) {
    pair.flag[qi] = ok ? 1 : 0;
    if (ok) pair.raw[qi] = m;
  }
#ifdef VSM_MATCH_TIMING
  if (!cfg.sparse && (threadIdx.x & 63) == 0) {
    const long long mt1 = clock64();
    const unsigned int k = atomicAdd(&vsm_mt_n, 1u);
    if (k < (1u << 18)) {
      vsm_mt[k][0] = (unsigned int)(mt1 - mt0);
      vsm_mt[k][1] = mtst[0];
      vsm_mt[k][2] = mtst[1];
      vsm_mt[k][3] = mtst[2];
      vsm_mt[k][4] = mtst[3];
      vsm_mt[k][5] = (unsigned int)mtph_[0];
#if VSM_MATCH_TIMING == 3  // wall clock (100 MHz, one counter for the whole device: s_memtime runs per XCD): life and start
      vsm_mt[k][6] = (unsigned int)(wall_clock64() - mtw0);
      vsm_mt[k][7] = (unsigned int)mtw0;
#else
      vsm_mt[k][6] = (unsigned int)mtph_[1];
      vsm_mt[k][7] = (unsigned int)mt0;
#endif
    }
  }
#endif
}

// ordered compaction of the accepted queries (push_back order = ascending query index) with the
// first-come pixel de-dup of flow / stereo (M[] in viso/matcher.cpp:1036-1039, :1078-1081):
// features sharing a pixel come from one NMS cell, hence are at most 3 indices apart.
// Two small kernels, 256 queries per block: k_compact_count leaves one survivor count per
// block, k_compact_write sums the counts of the blocks before it (a few hundred at most) and
// writes its survivors; nothing is serialised through one block.
__device__ __forceinline__ bool match_kept(const VsmPair &pair, int method, int i) {
  if (!pair.flag[i]) return false;
  if (method < 2) {
    const float u = pair.raw[i].u1c, v = pair.raw[i].v1c;
    for (int j = max(i - 3, 0); j < i; j++)
      if (pair.flag[j] && pair.raw[j].u1c == u && pair.raw[j].v1c == v) return false;
  }
  return true;
}

__global__ void __launch_bounds__(256)
    k_compact_count(const VsmPair *__restrict__ pairs, const VsmJob *__restrict__ jobs, VsmJob job0, int method, int pass) {
  __shared__ int s_cnt[4];
  const VsmPair &pair = pairs[blockIdx.y];
  const int n_query = (jobs ? jobs[blockIdx.y] : job0).nq[pass];
  if ((int)blockIdx.x * 256 >= n_query && blockIdx.x > 0) return;
  const int i = blockIdx.x * 256 + threadIdx.x;
  const bool keep = i < n_query && match_kept(pair, method, i);
  const unsigned long long b = __ballot(keep);
  if ((threadIdx.x & 63) == 0) s_cnt[threadIdx.x >> 6] = __popcll(b);
  __syncthreads();
  if (threadIdx.x == 0) pair.blockcnt[blockIdx.x] = s_cnt[0] + s_cnt[1] + s_cnt[2] + s_cnt[3];
}

__global__ void __launch_bounds__(256)
    k_compact_write(const VsmPair *__restrict__ pairs, const VsmJob *__restrict__ jobs, VsmJob job0, int method, int pass) {
  __shared__ int s_red[4];
  __shared__ int s_cnt[4];
  const VsmPair &pair = pairs[blockIdx.y];
  const int n_query = (jobs ? jobs[blockIdx.y] : job0).nq[pass];
  const int nblk = max((n_query + 255) / 256, 1);  // blocks that hold queries of this pair
  if ((int)blockIdx.x >= nblk) return;
  vsm_p_match *__restrict__ list = pass ? pair.list2 : pair.list1;
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  // base = survivors of all earlier blocks
  int part = 0;
  for (int b = threadIdx.x; b < (int)blockIdx.x; b += 256) part += pair.blockcnt[b];
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) part += __shfl_xor(part, o, 64);
  const int i = blockIdx.x * 256 + threadIdx.x;
  const bool keep = i < n_query && match_kept(pair, method, i);
  const unsigned long long bal = __ballot(keep);
  if (lane == 0) {
    s_red[wv] = part;
    s_cnt[wv] = __popcll(bal);
  }
  __syncthreads();
  int pos = s_red[0] + s_red[1] + s_red[2] + s_red[3];
  for (int w = 0; w < wv; w++) pos += s_cnt[w];
  pos += __popcll(bal & ((1ull << lane) - 1ull));
  if (keep) list[pos] = pair.raw[i];
  if ((int)blockIdx.x == nblk - 1 && threadIdx.x == 255) {
    const int total = pos + (keep ? 1 : 0);
    pair.count[pass] = total;
    pair.hcount[pass] = total;
  }
}

// Quad matching keeps every accepted query (no pixel de-dup, viso/matcher.cpp:1139-1151): ordered compaction of raw[] into
// the list in ONE launch (the lists behind it, and the Delaunay chain behind those, wait for it).  A workgroup of 256
// threads takes QUAD_SPAN consecutive queries of a pair: it counts the acceptance flags in front of its span itself (every
// workgroup reads the pair's flags up to its own - a few KB out of L2 - so no workgroup waits for another), scans its own and
// moves the records as 16-byte pieces, consecutive lanes consecutive pieces of raw[].  Round 3's form - four workgroups of
// 1024 threads and 16 KB of LDS per pair - took 35 us per 67 pairs with the GPU to itself and 100-170 us in the pipeline,
// whatever was in it: a 16-wave workgroup needs four free wave slots on every SIMD of one compute unit plus its LDS at the
// same moment, and beside the Delaunay chains it waits for that.  Four waves and 4 KB find a place at once.
// EXPORT (the per-frame path, where a launch of its own for the copy is 8 us of a 0.5 ms frame): 1 - every piece goes to the
// list's host-mapped copy as well (k_export_list's work), 2 - the pixel (u1c, v1c) of every match as x | y << 16 to xy_dst
// (k_export_xy's).
#define QUAD_SPAN 1024
template <int EXPORT>
__global__ void __launch_bounds__(256)
    k_compact_quad(const VsmPair *__restrict__ pairs, const VsmJob *__restrict__ jobs, VsmJob job0, int pass, uint32_t *__restrict__ xy_dst) {
  __shared__ int s_w[5];
  __shared__ int s_base[4];
  __shared__ int s_dst[QUAD_SPAN];  // place of every query of the span in the list, -1 = not accepted
  const VsmPair &pair = pairs[blockIdx.y];
  const int n_query = (jobs ? jobs[blockIdx.y] : job0).nq[pass];
  const int q0 = (int)blockIdx.x * QUAD_SPAN, q1 = min(n_query, q0 + QUAD_SPAN);
  if (q0 >= n_query && !(blockIdx.x == 0 && n_query == 0)) return;
  vsm_p_match *__restrict__ list = pass ? pair.list2 : pair.list1;
  const int t = threadIdx.x, lane = t & 63, wv = t >> 6;
  // accepted queries in front of this span: four flags per thread and load
  int before = 0;
  {
    const int4 *f4 = (const int4 *)pair.flag;  // (flag[] is 16-byte aligned, q0 a multiple of 4)
    for (int i0 = t; i0 < q0 / 4; i0 += 4 * 256) {
      int4 f[4];
#pragma unroll
      for (int k = 0; k < 4; k++) {
        const int i = i0 + 256 * k;
        f[k] = i < q0 / 4 ? f4[i] : make_int4(0, 0, 0, 0);
      }
#pragma unroll
      for (int k = 0; k < 4; k++) before += (f[k].x ? 1 : 0) + (f[k].y ? 1 : 0) + (f[k].z ? 1 : 0) + (f[k].w ? 1 : 0);
    }
  }
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) before += __shfl_xor(before, o, 64);
  if (lane == 0) s_base[wv] = before;
  // own flags: thread t owns queries q0 + 4 t .. q0 + 4 t + 3
  constexpr int RUN = QUAD_SPAN / 256;
  int keep[RUN], cnt = 0;
#pragma unroll
  for (int k = 0; k < RUN; k++) {
    const int i = q0 + t * RUN + k;
    keep[k] = i < q1 ? (pair.flag[i] ? 1 : 0) : 0;
    cnt += keep[k];
  }
  // exclusive scan over the 256 threads
  int incl = cnt;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    const int y = __shfl_up(incl, o, 64);
    if (lane >= o) incl += y;
  }
  if (lane == 63) s_w[wv] = incl;
  __syncthreads();
  int pos = incl - cnt, total = 0, base = 0;
#pragma unroll
  for (int w = 0; w < 4; w++) {
    pos += w < wv ? s_w[w] : 0;
    total += s_w[w];
    base += s_base[w];
  }
  pos += base;
#pragma unroll
  for (int k = 0; k < RUN; k++) s_dst[t * RUN + k] = keep[k] ? pos++ : -1;
  __syncthreads();
  {
    const uint4 *src = (const uint4 *)(pair.raw + q0);
    uint4 *dst = (uint4 *)list;
    uint4 *hdst = EXPORT == 1 ? (uint4 *)(pass ? pair.hlist2 : pair.hlist1) : nullptr;
    const int pieces = 3 * (q1 - q0);
    for (int p0 = t; p0 < pieces; p0 += 4 * 256) {
      uint4 v[4];
      int d[4], part[4];
#pragma unroll
      for (int k = 0; k < 4; k++) {
        const int p = p0 + 256 * k;
        const int e = p / 3;
        part[k] = p - 3 * e;
        d[k] = p < pieces ? s_dst[e] : -1;
        if (d[k] >= 0) v[k] = src[p];
      }
#pragma unroll
      for (int k = 0; k < 4; k++)
        if (d[k] >= 0) {
          dst[3 * (size_t)d[k] + part[k]] = v[k];
          if (EXPORT == 1) hdst[3 * (size_t)d[k] + part[k]] = v[k];
          if (EXPORT == 2 && part[k] == 1)  // (piece 1 of a record: v2p, i2p, u1c, v1c)
            xy_dst[d[k]] = (uint32_t)(int32_t)__uint_as_float(v[k].z) | ((uint32_t)(int32_t)__uint_as_float(v[k].w) << 16);
        }
    }
  }
  if (q1 == n_query && t == 255) {  // the span that holds the last query
    pair.count[pass] = base + total;
    pair.hcount[pass] = base + total;
  }
}

// wide copy of a finished list into host-mapped pinned memory (16 bytes per lane over PCIe):
// the host reads it after the stream sync, no D2H copy call and no second round trip
__global__ void __launch_bounds__(256)
    k_export_list(const VsmPair *__restrict__ pairs, int pass) {
  const VsmPair &pair = pairs[blockIdx.y];
  const int n16 = pair.count[pass] * 3;
  const uint4 *src = (const uint4 *)(pass ? pair.list2 : pair.list1);
  uint4 *dst = (uint4 *)(pass ? pair.hlist2 : pair.hlist1);
  for (int i = blockIdx.x * 256 + threadIdx.x; i < n16; i += gridDim.x * 256) dst[i] = src[i];
}

// the pixel of every match of the compacted pass-2 list as x | y << 16, into host-mapped memory: all the final
// removeOutliers' triangulation needs of the list ((u1c, v1c), which the refinement leaves alone, viso/matcher.cpp:1544-1577) -
// the per-frame path's host starts on it while the refinement and the list's export still run
__global__ void __launch_bounds__(256) k_export_xy(const VsmPair *__restrict__ pairs, uint32_t *__restrict__ dst) {
  const VsmPair &pair = pairs[0];
  const int n = pair.count[1];
  const vsm_p_match *__restrict__ src = pair.list2;
  for (int i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256)
    dst[i] = (uint32_t)(int32_t)src[i].u1c | ((uint32_t)(int32_t)src[i].v1c << 16);
}

// ---------------------------------------------------------------------------------------
// R1 refinement, viso/matcher.cpp:1498-1585.  One thread per (match, relocation step) evaluates
// the 25 candidate positions with the 16-byte ELAS descriptor (computeSmallDescriptor, :479-506)
// from the full-resolution Sobel planes; first-wins argmin in (dv, du) order.  Steps: 0 -> (u1p,v1p) [flow, quad], 1 -> (u2c,v2c) [stereo,
// quad], 2 -> (u2p,v2p) [quad]; each uses the unrefined (u1c,v1c) as its reference (:1544-1577).
// refinement==2 (parabolicFitting, :1379-1454): 49 lanes of a wave evaluate the 7x7 costs, the
// 3x3 neighbourhood around the minimum goes to the host, which solves the 9x6 least squares in
// double exactly as Matrix::solve does.
// ---------------------------------------------------------------------------------------
__device__ __forceinline__ uint4 small_desc(const uint8_t *__restrict__ du, const uint8_t *__restrict__ dv, int bpl,
                                            int u, int v) {
  const int a2 = v * bpl + u, a1 = a2 - bpl, a0 = a1 - bpl, a3 = a2 + bpl, a4 = a3 + bpl;
  uint4 r;
  r.x = du[a0] | (du[a1 - 2] << 8) | (du[a1] << 16) | ((uint32_t)du[a1 + 2] << 24);
  r.y = du[a2 - 1] | (du[a2] << 8) | (du[a2] << 16) | ((uint32_t)du[a2 + 1] << 24);
  r.z = du[a3 - 2] | (du[a3] << 8) | (du[a3 + 2] << 16) | ((uint32_t)du[a4] << 24);
  r.w = dv[a1] | (dv[a2 - 1] << 8) | (dv[a2 + 1] << 16) | ((uint32_t)dv[a3] << 24);
  return r;
}

// the same descriptor from the tiled plane
__device__ __forceinline__ uint4 small_desc_tiled(const uint8_t *__restrict__ t, int bpl, int u, int v) {
#define TDU(x, y) ((uint32_t)t[vsm_tiled_at(bpl, (x), (y))])
#define TDV(x, y) ((uint32_t)t[vsm_tiled_at(bpl, (x), (y)) + VSM_TILED_DV])
  uint4 r;
  r.x = TDU(u, v - 2) | (TDU(u - 2, v - 1) << 8) | (TDU(u, v - 1) << 16) | (TDU(u + 2, v - 1) << 24);
  r.y = TDU(u - 1, v) | (TDU(u, v) << 8) | (TDU(u, v) << 16) | (TDU(u + 1, v) << 24);
  r.z = TDU(u - 2, v + 1) | (TDU(u, v + 1) << 8) | (TDU(u + 2, v + 1) << 16) | (TDU(u, v + 2) << 24);
  r.w = TDV(u, v - 1) | (TDV(u - 1, v) << 8) | (TDV(u + 1, v) << 16) | (TDV(u, v + 1) << 24);
#undef TDU
#undef TDV
  return r;
}

__device__ __forceinline__ uint32_t sad16(const uint4 &a, const uint4 &b) {
  uint32_t s = __builtin_amdgcn_sad_u8(a.x, b.x, 0u);
  s = __builtin_amdgcn_sad_u8(a.y, b.y, s);
  s = __builtin_amdgcn_sad_u8(a.z, b.z, s);
  return __builtin_amdgcn_sad_u8(a.w, b.w, s);
}

// reference descriptor at (u1c, v1c) of the current left image (computeSmallDescriptor, :479-506):
// 5 du rows + 3 dv rows, columns u-2..u+2, each as two aligned dwords re-based with a funnel shift
// (8 wide loads instead of 16 byte gathers)
template <bool TILED>
__device__ __forceinline__ uint4 refine_ref_desc(const VsmImage &ref, const VsmDims &dc, int ru, int rv) {
  uint4 rd;
  const int b0 = (ru - 2) & ~3, rsh = 8 * ((ru - 2) - b0);
  uint64_t wu[5], wv[3];
  if (TILED) {
    // the two 4-pixel blocks holding columns ru-2 .. ru+2 (vsm_tiled_at of their first pixels)
    const int j0 = b0 >> 2;
#pragma unroll
    for (int r = 0; r < 5; r++) {
      const uint8_t *row = ref.duv_tiled;
      const uint32_t lo = ldg_u32(row + vsm_tiled_at(dc.bpl, 4 * j0, rv - 2 + r)), hi = ldg_u32(row + vsm_tiled_at(dc.bpl, 4 * j0 + 4, rv - 2 + r));
      wu[r] = ((((uint64_t)hi) << 32) | lo) >> rsh;
      if (r >= 1 && r <= 3) {
        const uint32_t lv = ldg_u32(row + vsm_tiled_at(dc.bpl, 4 * j0, rv - 2 + r) + VSM_TILED_DV), hv = ldg_u32(row + vsm_tiled_at(dc.bpl, 4 * j0 + 4, rv - 2 + r) + VSM_TILED_DV);
        wv[r - 1] = ((((uint64_t)hv) << 32) | lv) >> rsh;
      }
    }
  } else {
#pragma unroll
    for (int r = 0; r < 5; r++) {
      const uint32_t *pr = (const uint32_t *)(ref.du_full + (size_t)(rv - 2 + r) * dc.bpl + b0);
      wu[r] = ((((uint64_t)pr[1]) << 32) | pr[0]) >> rsh;
    }
#pragma unroll
    for (int r = 0; r < 3; r++) {
      const uint32_t *pr = (const uint32_t *)(ref.dv_full + (size_t)(rv - 1 + r) * dc.bpl + b0);
      wv[r] = ((((uint64_t)pr[1]) << 32) | pr[0]) >> rsh;
    }
  }
#define WB(w, c) ((uint32_t)((w) >> (8 * (c))) & 0xffu)
  rd.x = WB(wu[0], 2) | (WB(wu[1], 0) << 8) | (WB(wu[1], 2) << 16) | (WB(wu[1], 4) << 24);
  rd.y = WB(wu[2], 1) | (WB(wu[2], 2) << 8) | (WB(wu[2], 2) << 16) | (WB(wu[2], 3) << 24);
  rd.z = WB(wu[3], 0) | (WB(wu[3], 2) << 8) | (WB(wu[3], 4) << 16) | (WB(wu[4], 2) << 24);
  rd.w = WB(wv[0], 2) | (WB(wv[1], 1) << 8) | (WB(wv[1], 3) << 16) | (WB(wv[2], 2) << 24);
#undef WB
  return rd;
}

template <bool TILED>
__global__ void __launch_bounds__(256)
    k_refine(const VsmImage *__restrict__ imgs, const VsmPair *__restrict__ pairs, const VsmJob *__restrict__ jobs,
             VsmJob job0, VsmDims dp, VsmDims dc, int method, int nbx, int npairs) {
  // One thread per (match, relocation step).  The 9 x 9 du / 7 x 9 dv neighbourhood of the target
  // is pulled into registers with 48 independent dword loads (rows are 16-byte aligned, each row is
  // re-based to column u2-4 with a funnel shift), then the 25 candidate descriptors are pure
  // register byte-picks + v_sad_u8: no dependent gathers, no cross-lane traffic.
  // flattened grid: logical block -> (pair, block within pair), XCD-contiguous
  const int lb = xcd_remap(blockIdx.x, gridDim.x);
  const int pj = lb / nbx, bx = lb - pj * nbx;
  if (pj >= npairs) return;
  const VsmJob &jb = jobs ? jobs[pj] : job0;
  const VsmPair &pair = pairs[pj];
  const int g = bx * blockDim.x + threadIdx.x;
  const int mi = g / 3, step = g - mi * 3;
  const VsmImage &ref = imgs[jb.img_curr];
  // The reference descriptor of a match is the same for its three steps: lanes 0..21 of a wave compute those of the
  // wave's (at most 22) matches, one each - a third of the lanes in contiguous quads, which is what the texture
  // addresser's time goes by - and every lane picks its match's up with a cross-lane read.
  uint4 rd;
  {
    const int lane = threadIdx.x & 63;
    const int first_mi = (g - lane) / 3, rmi = first_mi + lane;
    uint4 mine = make_uint4(0, 0, 0, 0);
    if (lane < 22 && rmi < pair.count[1]) {
      const vsm_p_match *rm = pair.list2 + rmi;
      mine = refine_ref_desc<TILED>(ref, dc, (int)rm->u1c, (int)rm->v1c);
    }
    const int src = mi - first_mi;
    rd.x = (uint32_t)__shfl((int)mine.x, src, 64);
    rd.y = (uint32_t)__shfl((int)mine.y, src, 64);
    rd.z = (uint32_t)__shfl((int)mine.z, src, 64);
    rd.w = (uint32_t)__shfl((int)mine.w, src, 64);
  }
  if (mi >= pair.count[1]) return;
  if (step == 0 && !(method == 0 || method == 2)) return;
  if (step == 1 && !(method == 1 || method == 2)) return;
  if (step == 2 && method != 2) return;
  vsm_p_match *m = pair.list2 + mi;  // refined in place (each step owns its two fields)
  const VsmImage &tgt = step == 0 ? imgs[jb.img_prev] : (step == 1 ? imgs[jb.img_curr + 1] : imgs[jb.img_prev + 1]);
  const VsmDims &dt = step == 1 ? dc : dp;
  float *pu = step == 0 ? &m->u1p : (step == 1 ? &m->u2c : &m->u2p);
  float *pv = pu + 1;
  const float u2 = *pu, v2 = *pv;
  if (u2 - 2 < VSM_MARGIN || u2 + 2 > dt.w - 1 - VSM_MARGIN || v2 - 2 < VSM_MARGIN || v2 + 2 > dt.h - 1 - VSM_MARGIN)
    return;
  const int iu = (int)u2, iv = (int)v2;
  uint32_t U[9][3], V[9][3];
  if (TILED) {
    // tiled plane: the 9 columns iu-4 .. iu+4 start at pixel o = (iu-4) & 7 of a tile row and end in the next tile; one
    // 16-byte load per tile row brings du 0-3, dv 0-3, du 4-7, dv 4-7
    const int o = (iu - 4) & 7, b = o >> 2;
    const uint32_t sb = (uint32_t)(o & 3);
#pragma unroll
    for (int r = 0; r < 9; r++) {
      const uint8_t *pr = tgt.duv_tiled + vsm_tiled_at(dt.bpl, (iu - 4) & ~7, iv - 4 + r);
      const uint4 t0 = ldg_u4(pr), t1 = ldg_u4(pr + 128);
      const uint32_t d0 = b ? t0.z : t0.x, d1 = b ? t1.x : t0.z, d2 = b ? t1.z : t1.x;
      U[r][0] = __builtin_amdgcn_alignbyte(d1, d0, sb);
      U[r][1] = __builtin_amdgcn_alignbyte(d2, d1, sb);
      U[r][2] = d2 >> (8 * sb);
      const uint32_t e0 = b ? t0.w : t0.y, e1 = b ? t1.y : t0.w, e2 = b ? t1.w : t1.y;
      V[r][0] = __builtin_amdgcn_alignbyte(e1, e0, sb);
      V[r][1] = __builtin_amdgcn_alignbyte(e2, e1, sb);
      V[r][2] = e2 >> (8 * sb);
    }
  } else {
    const int a0 = (iu - 4) & ~3, sh = 8 * ((iu - 4) - a0);
#pragma unroll
    for (int r = 0; r < 9; r++) {
      const uint32_t *pr = (const uint32_t *)(tgt.du_full + (size_t)(iv - 4 + r) * dt.bpl + a0);
      const uint32_t d0 = pr[0], d1 = pr[1], d2 = pr[2];
      U[r][0] = (uint32_t)((((uint64_t)d1 << 32) | d0) >> sh);
      U[r][1] = (uint32_t)((((uint64_t)d2 << 32) | d1) >> sh);
      U[r][2] = d2 >> sh;
      if (r >= 1 && r <= 7) {
        const uint32_t *qr = (const uint32_t *)(tgt.dv_full + (size_t)(iv - 4 + r) * dt.bpl + a0);
        const uint32_t e0 = qr[0], e1 = qr[1], e2 = qr[2];
        V[r][0] = (uint32_t)((((uint64_t)e1 << 32) | e0) >> sh);
        V[r][1] = (uint32_t)((((uint64_t)e2 << 32) | e1) >> sh);
        V[r][2] = e2 >> sh;
      } else {
        V[r][0] = V[r][1] = V[r][2] = 0;
      }
    }
  }
#define UB(r, c) ((U[(r)][(c) >> 2] >> (8 * ((c)&3))) & 0xffu)
#define VB(r, c) ((V[(r)][(c) >> 2] >> (8 * ((c)&3))) & 0xffu)
  uint32_t best = 0xffffffffu;
  int ind = 0;
#pragma unroll
  for (int ddv = 0; ddv < 5; ddv++) {
#pragma unroll
    for (int ddu = 0; ddu < 5; ddu++) {
      const int r = ddv + 2, c = ddu + 2;  // candidate centre in window coordinates
      uint4 t;
      t.x = UB(r - 2, c) | (UB(r - 1, c - 2) << 8) | (UB(r - 1, c) << 16) | (UB(r - 1, c + 2) << 24);
      t.y = UB(r, c - 1) | (UB(r, c) << 8) | (UB(r, c) << 16) | (UB(r, c + 1) << 24);
      t.z = UB(r + 1, c - 2) | (UB(r + 1, c) << 8) | (UB(r + 1, c + 2) << 16) | (UB(r + 2, c) << 24);
      t.w = VB(r - 1, c) | (VB(r, c - 1) << 8) | (VB(r, c + 1) << 16) | (VB(r + 1, c) << 24);
      const uint32_t cost = sad16(rd, t);
      if (cost < best) {  // first minimum in (dv, du) order, viso/matcher.cpp:1484-1491
        best = cost;
        ind = ddv * 5 + ddu;
      }
    }
  }
#undef UB
#undef VB
  *pu = (float)((double)u2 + ((double)(float)(ind % 5) - 2.0));
  *pv = (float)((double)v2 + ((double)(float)(ind / 5) - 2.0));
}

__global__ void __launch_bounds__(256)
    k_parabolic_costs(const VsmImage *__restrict__ imgs, const VsmPair *__restrict__ pairs,
                      const VsmJob *__restrict__ jobs, VsmJob job0, VsmDims dp, VsmDims dc, int method) {
  const VsmJob &jb = jobs ? jobs[blockIdx.y] : job0;
  const VsmPair &pair = pairs[blockIdx.y];
  const int img_prev = jb.img_prev, img_curr = jb.img_curr;
  const int lane = threadIdx.x & 63;
  const int g = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;  // one wave per (match, step)
  const int mi = g / 3, step = g - mi * 3;
  if (mi >= pair.count[1]) return;
  int32_t *out = pair.pf + ((size_t)mi * 3 + step) * 12;
  bool active = !((step == 0 && !(method == 0 || method == 2)) || (step == 1 && !(method == 1 || method == 2)) ||
                  (step == 2 && method != 2));
  if (!active) {
    if (lane == 0) out[0] = 2;  // step not applicable
    return;
  }
  const vsm_p_match *m = pair.list2 + mi;
  const VsmImage &ref = imgs[img_curr];
  const VsmImage &tgt = step == 0 ? imgs[img_prev] : (step == 1 ? imgs[img_curr + 1] : imgs[img_prev + 1]);
  const VsmDims &dt = step == 1 ? dc : dp;
  const float u2 = step == 0 ? m->u1p : (step == 1 ? m->u2c : m->u2p);
  const float v2 = step == 0 ? m->v1p : (step == 1 ? m->v2c : m->v2p);
  if (u2 - 3 < VSM_MARGIN || u2 + 3 > dt.w - 1 - VSM_MARGIN || v2 - 3 < VSM_MARGIN || v2 + 3 > dt.h - 1 - VSM_MARGIN) {
    if (lane == 0) out[0] = 0;  // infeasible: match dropped (wave-uniform branch)
    return;
  }
  const bool tiled = ref.duv_tiled != nullptr;
  const uint4 r = tiled ? small_desc_tiled(ref.duv_tiled, dc.bpl, (int)m->u1c, (int)m->v1c)
                        : small_desc(ref.du_full, ref.dv_full, dc.bpl, (int)m->u1c, (int)m->v1c);
  uint32_t key = 0xffffffffu;
  int cost = 0;
  if (lane < 49) {
    const int ddv = lane / 7, ddu = lane - ddv * 7;
    const uint4 t = tiled ? small_desc_tiled(tgt.duv_tiled, dt.bpl, (int)u2 + ddu - 3, (int)v2 + ddv - 3)
                          : small_desc(tgt.du_full, tgt.dv_full, dt.bpl, (int)u2 + ddu - 3, (int)v2 + ddv - 3);
    cost = (int)sad16(r, t);
    key = ((uint32_t)cost << 6) | (uint32_t)lane;
  }
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) key = min(key, (uint32_t)__shfl_xor((int)key, o, 64));
  // key is now wave-uniform: first minimum in (dv, du) order
  const int ind = key & 63, du = ind % 7, dv = ind / 7;
  const bool border = (du == 0 || du == 6 || dv == 0 || dv == 6);
  int c9[9];
#pragma unroll
  for (int k = 0; k < 9; k++) {
    int src = border ? 0 : (dv + k / 3 - 1) * 7 + (du + k % 3 - 1);
    c9[k] = __shfl(cost, src, 64);
  }
  if (lane == 0) {
    if (border) {
      out[0] = 0;
    } else {
      out[0] = 1;
      out[1] = du;
      out[2] = dv;
#pragma unroll
      for (int k = 0; k < 9; k++) out[3 + k] = c9[k];
    }
  }
}

// ---------------------------------------------------------------------------------------
// refinement==2 in the batched (look-ahead) path: the least-squares tail of parabolicFitting (viso/matcher.cpp:1425-1453;
// host form: vsm_host_parabolic_update, vsm_host.cpp) and the removal of the matches whose fit fails (:1541-1581), on the
// device.  Matrix::operator* and Matrix::solve (Gauss-Jordan with full pivoting, viso/matrix.cpp) are + - * / and
// comparisons in double, evaluated here in the reference's order: IEEE arithmetic on either side, contraction off, so the
// same bits.  One workgroup per pair: every thread fits its matches and parks the updated records in raw[], a scan over
// the verdicts gives the survivors their places, the records move back into the list as 16-byte pieces.
// ---------------------------------------------------------------------------------------
// The system matrix At*A is the same for every fit, so the elimination's pivots, row swaps and multipliers are too: the
// host runs Gauss-Jordan on it ONCE (same IEEE double arithmetic, contraction off) and records, per step, what the
// reference does to the right-hand side - swap B[irow], B[icol]; B[icol] *= pivinv; B[ll] -= B[icol] * dum[ll] - and the
// device replays exactly those operations on every fit's b: the same bits as solving the whole system each time, without
// a 6 x 6 matrix per thread.
struct VsmParaPlan {
  int32_t ok, irow[6], icol[6];
  double pivinv[6], dum[6][6];
};
static const double kParaA[9][6] = {{1, 1, 1, -1, -1, 1}, {0, 1, 0, 0, -1, 1}, {1, 1, -1, 1, -1, 1}, {1, 0, 0, -1, 0, 1}, {0, 0, 0, 0, 0, 1},
                                    {1, 0, 0, 1, 0, 1},   {1, 1, -1, -1, 1, 1}, {0, 1, 0, 0, 1, 1},  {1, 1, 1, 1, 1, 1}};
static VsmParaPlan make_para_plan() {  // Matrix::solve (viso/matrix.cpp) on At*A, the right-hand side's share recorded
  VsmParaPlan pl = VsmParaPlan();
  double A[6][6];
  for (int i = 0; i < 6; i++)
    for (int j = 0; j < 6; j++) {
      double t = 0;
      for (int k = 0; k < 9; k++) t += kParaA[k][i] * kParaA[k][j];
      A[i][j] = t;
    }
  int ipiv[6] = {0, 0, 0, 0, 0, 0};
  int icol = 0, irow = 0;
  pl.ok = 1;
  for (int i = 0; i < 6; i++) {
    double big = 0.0;
    for (int j = 0; j < 6; j++)
      if (ipiv[j] != 1)
        for (int k = 0; k < 6; k++)
          if (ipiv[k] == 0 && fabs(A[j][k]) >= big) {
            big = fabs(A[j][k]);
            irow = j;
            icol = k;
          }
    ++ipiv[icol];
    pl.irow[i] = irow;
    pl.icol[i] = icol;
    if (irow != icol)
      for (int l = 0; l < 6; l++) std::swap(A[irow][l], A[icol][l]);
    if (fabs(A[icol][icol]) < 1e-20) {
      pl.ok = 0;
      return pl;
    }
    const double pivinv = 1.0 / A[icol][icol];
    pl.pivinv[i] = pivinv;
    A[icol][icol] = 1.0;
    for (int l = 0; l < 6; l++) A[icol][l] *= pivinv;
    for (int ll = 0; ll < 6; ll++)
      if (ll != icol) {
        const double dum = A[ll][icol];
        pl.dum[i][ll] = dum;
        A[ll][icol] = 0.0;
        for (int l = 0; l < 6; l++) A[ll][l] -= A[icol][l] * dum;
      }
  }
  return pl;
}
__device__ inline double para_get(const double b[6], int i) {
  return i == 0 ? b[0] : (i == 1 ? b[1] : (i == 2 ? b[2] : (i == 3 ? b[3] : (i == 4 ? b[4] : b[5]))));
}
__device__ inline void para_set(double b[6], int i, double v) {
#pragma unroll
  for (int k = 0; k < 6; k++) b[k] = k == i ? v : b[k];
}
__device__ inline bool dev_parabolic_update(const VsmParaPlan &pl, const int32_t *c9, int du, int dv, float &u2, float &v2) {
  constexpr double kA[9][6] = {{1, 1, 1, -1, -1, 1}, {0, 1, 0, 0, -1, 1}, {1, 1, -1, 1, -1, 1}, {1, 0, 0, -1, 0, 1}, {0, 0, 0, 0, 0, 1},
                               {1, 0, 0, 1, 0, 1},   {1, 1, -1, -1, 1, 1}, {0, 1, 0, 0, 1, 1},  {1, 1, 1, 1, 1, 1}};
  double b[6];
#pragma unroll
  for (int i = 0; i < 6; i++) {  // b = At * c (Matrix::operator*: the sum over k in order, zero terms included)
    double s = 0;
#pragma unroll
    for (int k = 0; k < 9; k++) s += kA[k][i] * (double)c9[k];
    b[i] = s;
  }
  if (!pl.ok) return false;
#pragma unroll
  for (int i = 0; i < 6; i++) {
    const int irow = pl.irow[i], icol = pl.icol[i];
    if (irow != icol) {
      const double x = para_get(b, irow), y = para_get(b, icol);
      para_set(b, irow, y);
      para_set(b, icol, x);
    }
    const double bc = para_get(b, icol) * pl.pivinv[i];
    para_set(b, icol, bc);
#pragma unroll
    for (int ll = 0; ll < 6; ll++)
      if (ll != icol) b[ll] -= bc * pl.dum[i][ll];
  }
  const float divisor = (float)(b[2] * b[2] - 4.0 * b[0] * b[1]);
  if ((double)fabsf(divisor) < 1e-8 || fabs(b[2]) < 1e-8) return false;
  const float ddv = (float)((2.0 * b[0] * b[4] - b[2] * b[3]) / (double)divisor);
  const float ddu = (float)(-(b[4] + 2.0 * b[1] * (double)ddv) / b[2]);
  if ((double)fabsf(ddu) >= 1.0 || (double)fabsf(ddv) >= 1.0) return false;
  u2 = (float)((double)u2 + ((double)(float)du - 3.0 + (double)ddu));
  v2 = (float)((double)v2 + ((double)(float)dv - 3.0 + (double)ddv));
  return true;
}
#define PARA_MAX_LIST 16384  // matches per pair this kernel takes (16-bit places in LDS)
__global__ void __launch_bounds__(1024) k_parabolic_apply(const VsmPair *__restrict__ pairs, VsmParaPlan pl) {
  __shared__ uint16_t s_dst[PARA_MAX_LIST];  // the match's place among the survivors, 0xffff = dropped
  __shared__ int s_w[17];
  const VsmPair &pair = pairs[blockIdx.x];
  const int n = min(pair.count[1], PARA_MAX_LIST);
  const int t = threadIdx.x;
  // thread t owns the run of matches [t * run, t * run + run)
  const int run = (n + 1023) / 1024;
  int cnt = 0;
  for (int k = 0; k < run; k++) {
    const int i = t * run + k;
    if (i >= n) break;
    vsm_p_match m = pair.list2[i];
    bool ok = true;
    float *tu[3] = {&m.u1p, &m.u2c, &m.u2p}, *tv[3] = {&m.v1p, &m.v2c, &m.v2p};
    for (int st = 0; st < 3 && ok; st++) {
      const int32_t *r = pair.pf + ((size_t)i * 3 + st) * 12;
      if (r[0] == 2) continue;  // step not applicable to the matching method
      ok = r[0] == 1 && dev_parabolic_update(pl, r + 3, r[1], r[2], *tu[st], *tv[st]);
    }
    pair.raw[i] = m;
    s_dst[i] = ok ? 1 : 0;
    cnt += ok ? 1 : 0;
  }
  int total;
  int pos = block_excl_scan_1024(cnt, total, s_w);
  for (int k = 0; k < run; k++) {
    const int i = t * run + k;
    if (i >= n) break;
    s_dst[i] = s_dst[i] ? (uint16_t)pos++ : (uint16_t)0xffffu;
  }
  __syncthreads();  // (raw[] written above is read below by other threads of this workgroup, and only by them)
  const uint4 *src = (const uint4 *)pair.raw;
  uint4 *dst = (uint4 *)pair.list2;
  for (int p = t; p < 3 * n; p += 1024) {
    const int e = p / 3;
    const uint32_t d = s_dst[e];
    if (d != 0xffffu) dst[3 * (size_t)d + (p - 3 * e)] = src[p];
  }
  if (t == 0) {
    pair.count[1] = total;
    pair.hcount[1] = total;
  }
}

// ---------------------------------------------------------------------------------------
// Small tables (job descriptions of a chunk: 10-20 KB) from pinned host memory to HBM by a kernel on the stream that needs
// them, not by hipMemcpyAsync.  The runtime takes a copy of more than 16 KB to a DMA engine, and when that engine is busy
// with another upload - two chunks' tables now and then - it falls back to a shader copy on a hardware queue of its own,
// which it creates then and there: 180 MB of context-save area mapped and touched, 6-7 ms during which every launch of
// the process waits (the "once per process" stall of round 4; tools/stall_probe.py, tools/shim/mmap_trace.c).
// ---------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) k_upload(uint32_t *__restrict__ dst, const uint32_t *__restrict__ src, int n_words) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i < n_words) dst[i] = src[i];
}
// Head records of the dense set's fine bins (VsmSet::heads): dword 0 = the bin's start in the sorted arrays, dwords 1..15 = the
// packed coordinates of the 15 candidates from there on, whatever bins they lie in (a search window's run of candidates is
// consecutive in that order: fine rows vfmin..vfmax of one (class, u-bin) column).  k_match's second pass reads a window's bin
// start AND its candidates' coordinates in one round trip: measured on the benchmark sequence (tools/match_timing.py,
// -DVSM_MATCH_TIMING=2), 15 inline candidates cover every lane's longest run in 99.9-100 % of a wave's stages, 7 in 1.7 %
// (groups sharing a run) / 41.7 % (a lane per bin), 3 in 0.4 % - a wave saves the trip only if all of its lanes do.
// One thread per (bin, quarter of the record).
__global__ void __launch_bounds__(256) k_feat_heads(const VsmImage *__restrict__ imgs, int first, int nb) {
  const VsmSet &st = imgs[first + blockIdx.y].set[1];
  const int tid = blockIdx.x * 256 + threadIdx.x, b = tid >> 2, j = tid & 3;
  if (b >= nb || st.heads == nullptr) return;
  const int start = ldg_i32(st.bin_start + b);
  uint32_t w[4];
#pragma unroll
  for (int i = 0; i < 4; i++) {
    const int at = min(start + 4 * j + i - 1, st.cap - 1);  // (beyond the set's count: whatever is there - a position >= the run's end is never taken)
    w[i] = (j == 0 && i == 0) ? (uint32_t)start : ldg_u32_at(st.s_uv, (uint32_t)max(at, 0) * 4u);
  }
  st.heads[b * 4 + j] = make_uint4(w[0], w[1], w[2], w[3]);
}

hipError_t vsm_upload(hipStream_t s, void *dst_device, const void *src_pinned, size_t bytes) {
  if (bytes == 0) return hipSuccess;
  if ((bytes & 3) || ((uintptr_t)dst_device & 3) || ((uintptr_t)src_pinned & 3)) return hipMemcpyAsync(dst_device, src_pinned, bytes, hipMemcpyHostToDevice, s);
  const int n = (int)(bytes >> 2);
  hipLaunchKernelGGL(k_upload, dim3((n + 255) / 256), dim3(256), 0, s, (uint32_t *)dst_device, (const uint32_t *)src_pinned, n);
  return hipGetLastError();
}

// =======================================================================================
// launchers
// =======================================================================================
static inline int cdiv(int a, int b) { return (a + b - 1) / b; }

// plan of k_feat_order for a geometry; false = it does not fit (huge search bins or suppression cells: the separate kernels)
static bool vsm_order_plan(const VsmDims &d, const VsmImage &im, int set_lo, int binsize, int nb, VsmOrderPlan &pl) {
  static bool attr_set = false;  // (the scan kernel's histogram may want more than the default 64 KB of dynamic LDS)
  if (!attr_set) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void *>(k_feat_scan), hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024);
    attr_set = true;
  }
  if ((size_t)nb * 4 > 150 * 1024) return false;
  const int bw = (binsize + d.scale - 1) / d.scale + 1;  // pixels of one search bin at the matching resolution (at most)
  if (bw > 72) return false;
  pl.bu = std::max(1, std::min(8, 128 / bw));
  pl.tiles_u = cdiv(d.ub, pl.bu);
  pl.tiles_v = d.vb;
  pl.stage_w = (pl.bu * bw + 13 + 3) & ~3;
  pl.stage_h = bw + 10;
  int maxcells = 0, maxl = 0;
  for (int k = 0; k < 2; k++) {
    const VsmSet &st = im.set[k];
    pl.cells_cap[k] = pl.lcap[k] = 0;
    if (k < set_lo || st.ncu * st.ncv <= 0) continue;
    const int n1 = st.nms_n + 1;
    const int cwt = std::min(st.ncu, (pl.bu * bw - 1) / n1 + 2), cht = std::min(st.ncv, (bw - 1) / n1 + 2);  // cells meeting a tile / ...
    const int cwb = std::min(st.ncu, (bw - 1) / n1 + 2);                                                  // ... one bin
    pl.cells_cap[k] = cwt * cht;
    pl.lcap[k] = (cwb * cht + 3) & ~3;
    maxcells = std::max(maxcells, pl.cells_cap[k]);
    maxl = std::max(maxl, pl.lcap[k]);
    if (st.cap >= (1 << 21)) return false;  // (survivor words: 21 bits of feature index)
  }
  if (maxcells <= 0 || maxcells > 512) return false;
  int o = pl.stage_w * pl.stage_h * 2;
  auto take = [&](int bytes) {
    o = (o + 15) & ~15;
    const int at = o;
    o += bytes;
    return at;
  };
  pl.cells_max = maxcells;
  pl.o_cand = take(maxcells * 16);
  pl.o_bs = take(2 * pl.bu * 4 * (VSM_VSUB + 1) * 4);
  pl.o_cnt = take((pl.bu * 4 + 1) * 4);
  pl.o_list = take(pl.bu * 4 * maxl * 4);
  pl.o_sv = take(maxcells * 4 * 4);
  pl.lds_bytes = (o + 15) & ~15;
  return pl.lds_bytes <= 64 * 1024;
}

void vsm_launch_ingest(hipStream_t s, VsmProf &pf, const VsmImage *d_imgs, int first, const uint8_t *src0,
                       const uint8_t *src1, size_t frame_stride, int32_t src_bpl, int n_frames, const VsmDims &d) {
  const int sides = src1 ? 2 : 1;
  dim3 grid(cdiv((d.bpl / 4) * d.h, 256), 1, n_frames * sides);
  pf.begin(VSM_K_INGEST, s);
  hipLaunchKernelGGL(k_ingest, grid, dim3(256), 0, s, d_imgs, first, src0, src1, frame_stride, src_bpl, sides, d.w, d.h,
                     d.bpl);
  pf.end(s);
}

void vsm_launch_front(hipStream_t s, VsmProf &pf, const VsmImage *d_imgs, int first, const uint8_t *src0, const uint8_t *src1,
                      size_t frame_stride, int32_t src_bpl, int n_frames, const VsmDims &d, int write_img) {
  const int sides = src1 ? 2 : 1;
  const int nbx = cdiv(d.bpl, FRONT_TW), nby = cdiv(d.h, FRONT_TH), n_img = n_frames * sides;
  dim3 grid(((nbx * nby * n_img + 7) / 8) * 8);
  pf.begin(VSM_K_FRONT, s);
  hipLaunchKernelGGL(k_front, grid, dim3(256), 0, s, d_imgs, first, src0, src1, frame_stride, src_bpl, sides, d, write_img, nbx, nby, n_img);
  pf.end(s);
}

int vsm_launch_features(hipStream_t s, VsmProf &pf, const VsmImage *d_imgs, int first, int n_img, const VsmDims &d,
                        int16_t *f1, int16_t *f2, size_t f_stride, int nms_tau, int multi_stage, int half_res,
                        int binsize, const VsmImage *h_imgs, int front_done, int fused) {
  if (half_res && !front_done) {
    pf.begin(VSM_K_HALVE, s);
    hipLaunchKernelGGL(k_halve, dim3(cdiv(d.mbpl / 4, 256), d.mh, n_img), dim3(256), 0, s, d_imgs, first, d);
    pf.end(s);
    pf.begin(VSM_K_SOBEL_FULL, s);
    hipLaunchKernelGGL(k_filters<true>, dim3(cdiv(d.bpl / 4, 64), cdiv(d.h, 16), n_img), dim3(256), 0, s, d_imgs, first,
                       d.bpl, d.h, (int16_t *)nullptr, (int16_t *)nullptr, (size_t)0);
    pf.end(s);
  }
  const int set_lo = multi_stage ? 0 : 1;
  const int nb = 4 * d.ub * d.vb * VSM_VSUB;  // fine bins
  int max_cells = 0, max_cap = 0;
  for (int k = 0; k < 2; k++) {
    max_cells = max(max_cells, h_imgs[first].set[k].ncu * h_imgs[first].set[k].ncv);
    max_cap = max(max_cap, h_imgs[first].set[k].cap);
  }
  // the fused tiles serve the default radii (dense 3, sparse 9: viso/matcher.cpp:685-687); other radii take the separate
  // filter + suppression kernels with f1 / f2 in HBM
  const VsmSet &sd = h_imgs[first].set[1], &ss = h_imgs[first].set[0];
  const bool fuse = (fused & 1) && sd.nms_n == 3 && sd.ncu * sd.ncv > 0 && (!multi_stage || (ss.nms_n == 9 && ss.ncu * ss.ncv > 0));
  if (fuse) {
    int dx, dy;
    vsm_feat_tiles(d, sd, dx, dy);
    const int nbx = dx * dy;
    pf.begin(VSM_K_FEAT_DENSE, s);
    if (fused & 2)
      hipLaunchKernelGGL(k_feat_dense<true>, dim3(((nbx * n_img + 7) / 8) * 8), dim3(256), 0, s, d_imgs, first, d, nms_tau, dx, nbx,
                         n_img, f1, f2, f_stride);
    else
      hipLaunchKernelGGL(k_feat_dense<false>, dim3(((nbx * n_img + 7) / 8) * 8), dim3(256), 0, s, d_imgs, first, d, nms_tau, dx, nbx,
                         n_img, f1, f2, f_stride);
    pf.end(s);
    if (multi_stage) {
      const int sx = cdiv(ss.ncu, VfSparse::CU), sy = cdiv(ss.ncv, VfSparse::CV);
      pf.begin(VSM_K_FEAT_SPARSE, s);
      hipLaunchKernelGGL(k_feat_sparse, dim3(((sx * sy * n_img + 7) / 8) * 8), dim3(256), 0, s, d_imgs, first, d, nms_tau, sx, sx * sy,
                         n_img);
      pf.end(s);
    }
  } else {
    pf.begin(VSM_K_FILTERS, s);
    hipLaunchKernelGGL(k_filters<false>, dim3(cdiv(d.mbpl / 4, 64), cdiv(d.mh, 16), n_img), dim3(256), 0, s, d_imgs, first,
                       d.mbpl, d.mh, f1, f2, f_stride);
    pf.end(s);
  }
  if (max_cells > 0 && !fuse) {
    // per set: small n -> LDS tile kernel, mid n -> 8-lane LDS tile kernel, else wave per cell
    for (int k = set_lo; k < 2; k++) {
      const VsmSet &st = h_imgs[first].set[k];
      if (st.ncu * st.ncv <= 0) continue;
      pf.begin(k == 1 ? VSM_K_NMS : VSM_K_NMS_SPARSE, s);
      if (st.nms_n == 3) {
        const int nbx = cdiv(st.ncu, 16) * cdiv(st.ncv, 8);
        hipLaunchKernelGGL((k_nms_fixed<3, 16, 8, 1>), dim3(((nbx * n_img + 7) / 8) * 8), dim3(256), 0, s, d_imgs, first, d,
                           f1, f2, f_stride, nms_tau, k, nbx, n_img);
      } else if (st.nms_n == 9) {
        const int nbx = cdiv(st.ncu, 4) * cdiv(st.ncv, 4);
        hipLaunchKernelGGL((k_nms_fixed<9, 4, 4, 8>), dim3(((nbx * n_img + 7) / 8) * 8), dim3(256), 0, s, d_imgs, first, d,
                           f1, f2, f_stride, nms_tau, k, nbx, n_img);
      } else if (st.nms_n <= NMS_TILE_MAXN) {
        const int nbx = cdiv(st.ncu, NMS_TCU) * cdiv(st.ncv, NMS_TCV);
        hipLaunchKernelGGL(k_nms_tile, dim3(((nbx * n_img + 7) / 8) * 8), dim3(256), 0, s, d_imgs, first, d, f1, f2, f_stride,
                           nms_tau, k, nbx, n_img);
      } else if (st.nms_n <= NMS8_MAXN) {
        const int nbx = cdiv(st.ncu, NMS8_TC) * cdiv(st.ncv, NMS8_TC);
        hipLaunchKernelGGL(k_nms_tile8, dim3(((nbx * n_img + 7) / 8) * 8), dim3(256), 0, s, d_imgs, first, d, f1, f2, f_stride,
                           nms_tau, k, nbx, n_img);
      }
      else
        hipLaunchKernelGGL(k_nms, dim3(cdiv(st.ncu * st.ncv, 4), 2, n_img * 2), dim3(256), 0, s, d_imgs, first, d, f1, f2,
                           f_stride, nms_tau, k, k);
      pf.end(s);
    }
  }
  // records + bin-sorted copy: the two-kernel form where a tile of whole search bins fits a workgroup's LDS
  VsmOrderPlan pl;
  if ((fused & 4) == 0 && max_cells > 0 && vsm_order_plan(d, h_imgs[first], set_lo, binsize, nb, pl)) {
    pf.begin(VSM_K_FEAT_SCAN, s);
    VsmBinDiv bd;
    bd.binsize = binsize;
    bd.magic = binsize >= 2 ? (uint32_t)(((1ull << 32) + (uint64_t)binsize - 1) / (uint64_t)binsize) : 0u;
    hipLaunchKernelGGL(k_feat_scan, dim3(1, 2, n_img), dim3(1024), (size_t)nb * 4, s, d_imgs, first, set_lo, d, bd, nb);
    pf.end(s);
    pf.begin(VSM_K_FEAT_ORDER, s);
    const int nbx = pl.tiles_u * pl.tiles_v;
    hipLaunchKernelGGL(k_feat_order, dim3(((nbx * n_img + 7) / 8) * 8), dim3(256), (size_t)pl.lds_bytes, s, d_imgs, first, d, set_lo, bd, pl,
                       nbx, n_img);
    pf.end(s);
    if (h_imgs[first].set[1].heads) hipLaunchKernelGGL(k_feat_heads, dim3(cdiv(nb * 4, 256), n_img), dim3(256), 0, s, d_imgs, first, nb);
    return (!fuse || (fused & 2)) ? 1 : 0;
  }
  pf.begin(VSM_K_SCAN, s);
  hipLaunchKernelGGL(k_scan_cells, dim3(1, 2, n_img), dim3(1024), 0, s, d_imgs, first, set_lo, nb);
  pf.end(s);
  if (max_cells > 0) {
    pf.begin(VSM_K_EMIT, s);
    int nbx = 1;  // tiles per image: the larger of the two sets decides the grid
    for (int k = set_lo; k < 2; k++) {
      const VsmSet &st = h_imgs[first].set[k];
      const int n1 = st.nms_n + 1;
      nbx = max(nbx, cdiv(st.ncu, max(EMIT_TW / n1, 1)) * cdiv(st.ncv, max(EMIT_TH / n1, 1)));
    }
    hipLaunchKernelGGL(k_emit, dim3(((nbx * n_img + 7) / 8) * 8, 2), dim3(256), 0, s, d_imgs, first, d, set_lo, binsize, nbx,
                       n_img);
    pf.end(s);
  }
  pf.begin(VSM_K_BINSCAN, s);
  hipLaunchKernelGGL(k_bin_scan, dim3(1, 2, n_img), dim3(1024), 0, s, d_imgs, first, set_lo, nb);
  pf.end(s);
  pf.begin(VSM_K_BINSCATTER, s);
  hipLaunchKernelGGL(k_bin_scatter, dim3(cdiv(max_cap, 256), 2, n_img), dim3(256), 0, s, d_imgs, first, set_lo);
  pf.end(s);
  pf.begin(VSM_K_BINRANK, s);
  hipLaunchKernelGGL(k_bin_rank, dim3(cdiv(max_cap, 256), 2, n_img), dim3(256), 0, s, d_imgs, first, set_lo);
  pf.end(s);
  if (h_imgs[first].set[1].heads) hipLaunchKernelGGL(k_feat_heads, dim3(cdiv(nb * 4, 256), n_img), dim3(256), 0, s, d_imgs, first, nb);
  return (!fuse || (fused & 2)) ? 1 : 0;
}

// One launch serves `npairs` frame pairs (blockIdx.y); jobs == nullptr: the single pair job0.
// pass: 0 = sparse lists (list1/hlist1/count[0]), 1 = dense lists.  max_nq bounds nq[pass].
bool vsm_launch_match(hipStream_t s, VsmProf &pf, const VsmImage *d_imgs, const VsmPair *d_pairs, const VsmJob *d_jobs,
                      const VsmJob &job0, int npairs, const VsmDims &d, const VsmMatchCfg &cfg, int max_nq, int fuse_export, uint32_t *xy_dst) {
  // lanes per query: the chain is latency-bound per wavefront, so big batches want many
  // queries per wave (G = 2..4) and a lone frame pair wants more lanes per query (G = 8).
  const long total_q = (long)npairs * max_nq;
#ifndef VSM_MATCH_GBIG
#define VSM_MATCH_GBIG 2
#endif
  const int G = total_q >= 200000 ? VSM_MATCH_GBIG : (total_q >= 30000 ? 4 : 8);
  const int pass = cfg.sparse ? 0 : 1;
  if (max_nq > 0) {
    pf.begin(cfg.sparse ? VSM_K_MATCH1 : VSM_K_MATCH2, s);
    const int nbx = cdiv(max_nq * G, VSM_MATCH_BLOCK);
    const dim3 grid(((nbx * npairs + 7) / 8) * 8);
    // without prior boxes (first pass, single-pass matching) the lanes of a group take whole u-bins
#define VSM_MATCH_LAUNCH(GG)                                                                                                       \
  do {                                                                                                                             \
    if (cfg.use_prior && cfg.heads && !cfg.sparse)                                                                                 \
      hipLaunchKernelGGL((k_match<GG, false, true>), grid, dim3(VSM_MATCH_BLOCK), 0, s, d_imgs, d_pairs, d_jobs, job0, d, cfg, nbx, npairs); \
    else if (cfg.use_prior)                                                                                                        \
      hipLaunchKernelGGL((k_match<GG, false>), grid, dim3(VSM_MATCH_BLOCK), 0, s, d_imgs, d_pairs, d_jobs, job0, d, cfg, nbx, npairs); \
    else                                                                                                                           \
      hipLaunchKernelGGL((k_match<GG, true>), grid, dim3(VSM_MATCH_BLOCK), 0, s, d_imgs, d_pairs, d_jobs, job0, d, cfg, nbx, npairs);  \
  } while (0)
    if (G == 1)
      VSM_MATCH_LAUNCH(1);
    else if (G == 2)
      VSM_MATCH_LAUNCH(2);
    else if (G == 4)
      VSM_MATCH_LAUNCH(4);
    else if (G == 16)
      VSM_MATCH_LAUNCH(16);
    else
      VSM_MATCH_LAUNCH(8);
    pf.end(s);
  }
  const int nblk = max(cdiv(max_nq, 256), 1);
  pf.begin(cfg.sparse ? VSM_K_COMPACT1 : VSM_K_COMPACT2, s);
  bool fused = false;
  if (cfg.method == 2) {
    const dim3 grid(std::max(1, cdiv(max_nq, QUAD_SPAN)), npairs);
    if (fuse_export == 1)
      hipLaunchKernelGGL(k_compact_quad<1>, grid, dim3(256), 0, s, d_pairs, d_jobs, job0, pass, nullptr);
    else if (fuse_export == 2 && xy_dst && npairs == 1)
      hipLaunchKernelGGL(k_compact_quad<2>, grid, dim3(256), 0, s, d_pairs, d_jobs, job0, pass, xy_dst);
    else
      hipLaunchKernelGGL(k_compact_quad<0>, grid, dim3(256), 0, s, d_pairs, d_jobs, job0, pass, nullptr);
    fused = fuse_export == 1 || (fuse_export == 2 && xy_dst && npairs == 1);
  } else {
    hipLaunchKernelGGL(k_compact_count, dim3(nblk, npairs), dim3(256), 0, s, d_pairs, d_jobs, job0, cfg.method, pass);
    hipLaunchKernelGGL(k_compact_write, dim3(nblk, npairs), dim3(256), 0, s, d_pairs, d_jobs, job0, cfg.method, pass);
  }
  pf.end(s);
  return fused;  // the export asked for went along with the compaction (quad matching): no launch of its own
}

void vsm_launch_export(hipStream_t s, VsmProf &pf, const VsmPair *d_pairs, int npairs, int pass, int n_upper) {
  pf.begin(VSM_K_EXPORT, s);
  hipLaunchKernelGGL(k_export_list, dim3(max(min(cdiv(n_upper * 3, 256), 256), 1), npairs), dim3(256), 0, s, d_pairs,
                     pass);
  pf.end(s);
}

void vsm_launch_export_xy(hipStream_t s, const VsmPair *d_pairs, uint32_t *dst_host_mapped, int n_upper) {
  hipLaunchKernelGGL(k_export_xy, dim3(max(min(cdiv(n_upper, 256), 64), 1)), dim3(256), 0, s, d_pairs, dst_host_mapped);
}

// the batched tail of refinement==2 (behind vsm_launch_refine): fits, dropped matches, the lists closed up again
void vsm_launch_parabolic_apply(hipStream_t s, const VsmPair *d_pairs, int npairs) {
  if (npairs <= 0) return;
  static const VsmParaPlan plan = make_para_plan();
  hipLaunchKernelGGL(k_parabolic_apply, dim3(npairs), dim3(1024), 0, s, d_pairs, plan);
}

void vsm_launch_refine(hipStream_t s, VsmProf &pf, const VsmImage *d_imgs, const VsmPair *d_pairs, const VsmJob *d_jobs,
                       const VsmJob &job0, int npairs, const VsmDims &dp, const VsmDims &dc, int method, int refinement,
                       int n_upper) {
  // n_upper bounds the list sizes (they are still device-only); surplus groups exit at once
  if (n_upper <= 0) return;
  pf.begin(VSM_K_REFINE, s);
  if (refinement == 2)
    hipLaunchKernelGGL(k_parabolic_costs, dim3(cdiv(n_upper * 3 * 64, 256), npairs), dim3(256), 0, s, d_imgs, d_pairs,
                       d_jobs, job0, dp, dc, method);
  else
  {
    const int nbx = cdiv(n_upper * 3, 256), tot = ((nbx * npairs + 7) / 8) * 8;
    if (dc.scale == 2)
      hipLaunchKernelGGL(k_refine<true>, dim3(tot), dim3(256), 0, s, d_imgs, d_pairs, d_jobs, job0, dp, dc, method, nbx, npairs);
    else
      hipLaunchKernelGGL(k_refine<false>, dim3(tot), dim3(256), 0, s, d_imgs, d_pairs, d_jobs, job0, dp, dc, method, nbx, npairs);
  }
  pf.end(s);
}
