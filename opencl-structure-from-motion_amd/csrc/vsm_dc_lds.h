// The 16-bit local mesh of the GPU-resident Delaunay kernels (vsm_dc.hip), in a header of its own so that a host build can
// run the very same accessors (tools/dc_lds_emulate.cpp walks the block kernel's lanes one after the other).
#pragma once

#include <stdint.h>

#include "vsm_dc_mesh.h"

#if defined(__HIPCC__) && defined(__HIP_DEVICE_COMPILE__)
#define DC2_AS3 __attribute__((address_space(3)))
#else
#define DC2_AS3
#endif
#if defined(__HIPCC__)
#define DC2_DEV __device__
#else
#define DC2_DEV
#endif
// 16-byte moves of records that are otherwise read and written word by word: may_alias, or type-based alias analysis may
// move a word access across the vector access to the same record
typedef int32_t dc2_v4i __attribute__((ext_vector_type(4), may_alias));
typedef uint32_t dc2_v4u __attribute__((ext_vector_type(4), may_alias));

// A sub-triangulation held entirely in LDS under LOCAL numbering (positions and slots count from the node's first
// position), 16 bytes per triangle: eight 16-bit words, 0-2 neighbour handles (slot * 4 + edge), 4-6 vertices, 0xffff = none.
// Half the bytes of the 32-bit records (more nodes resident per CU), ds_ instructions with small immediate offsets instead
// of flat accesses through rebased 64-bit pointers, and nothing to translate while the seam is walked: global numbering
// comes back when the records are written out (neighbour + 8 * off, vertex + off).  MAPPED: the top merge of a long list
// does not fit; there the records the seam can touch are cached under a 16-bit slot -> line map and stores also go
// through to global memory (see k_dc2_merge).
#ifdef DC2_COUNT_NULL_ACCESS
static long dc2_null_reads = 0, dc2_null_writes = 0, dc2_null_pts = 0;
#endif
#ifndef DC2_WORD
#define DC2_WORD uint16_t
#endif
typedef DC2_WORD dc2_word;
#define DC2_NONE ((uint32_t)(dc2_word)~(dc2_word)0)
template <bool MAPPED>
struct DcLdsMesh {
  typedef DcOTri OTri;
  DC2_AS3 dc2_word *rec;
  DC2_AS3 const uint32_t *pt;
  // MAPPED only
  DC2_AS3 uint16_t *map;
  DC2_AS3 int32_t *nrec;
  int32_t *gtri;  // the node's records in global memory (slot 0 = the node's first slot)
  int32_t rec_cap, tbase4, pbase;

  DC2_DEV static inline int32_t to_local(int32_t g, int w, int32_t tbase4, int32_t pbase) { return g < 0 ? -1 : g - (w < 4 ? tbase4 : pbase); }
  DC2_DEV inline int line_of(int32_t t) const {  // MAPPED: cache line of slot t, fetched on first use while there is room
    int i = map[t];
    if (i == 0xffff) {
      const int k = *nrec;
      if (k >= rec_cap) return -1;
      *nrec = k + 1;
      const dc2_v4i *g = (const dc2_v4i *)(gtri + (size_t)t * 8);
      const dc2_v4i a = g[0], b = g[1];
      dc2_v4u o;
      o.x = (uint32_t)(to_local(a.x, 0, tbase4, pbase) & 0xffff) | ((uint32_t)to_local(a.y, 1, tbase4, pbase) << 16);
      o.y = (uint32_t)(to_local(a.z, 2, tbase4, pbase) & 0xffff) | 0xffff0000u;
      o.z = (uint32_t)(to_local(b.x, 4, tbase4, pbase) & 0xffff) | ((uint32_t)to_local(b.y, 5, tbase4, pbase) << 16);
      o.w = (uint32_t)(to_local(b.z, 6, tbase4, pbase) & 0xffff) | 0xffff0000u;
      *(DC2_AS3 dc2_v4u *)(rec + k * 8) = o;
      map[t] = (uint16_t)k;
      i = k;
    }
    return i;
  }
  DC2_DEV inline int32_t ld(int32_t t, int w) const {
#ifdef DC2_COUNT_NULL_ACCESS
    if (t < 0) dc2_null_reads++;
#endif
    uint32_t v;
    if (MAPPED) {
      const int i = line_of(t);
      if (i < 0) return to_local(gtri[(size_t)t * 8 + w], w, tbase4, pbase);
      v = rec[i * 8 + w];
    } else {
      v = rec[t * 8 + w];
    }
    return v == DC2_NONE ? -1 : (int32_t)v;
  }
  DC2_DEV inline void st(int32_t t, int w, int32_t v) const {
#ifdef DC2_COUNT_NULL_ACCESS
    if (t < 0) dc2_null_writes++;
#endif

    if (MAPPED) {
      const int i = map[t];
      if (i != 0xffff) rec[i * 8 + w] = (dc2_word)v;
      gtri[(size_t)t * 8 + w] = v < 0 ? -1 : v + (w < 4 ? tbase4 : pbase);
    } else {
      rec[t * 8 + w] = (dc2_word)v;
    }
  }
  DC2_DEV inline OTri make(int32_t &tcur) const {
    const int32_t t = tcur++;
    if (MAPPED) {
      st(t, 0, -1);
      st(t, 1, -1);
      st(t, 2, -1);
      st(t, 4, -1);
      st(t, 5, -1);
      st(t, 6, -1);
    } else {
      dc2_v4u ones;
      ones.x = ones.y = ones.z = ones.w = 0xffffffffu;
      *(DC2_AS3 dc2_v4u *)(rec + t * 8) = ones;
      if (sizeof(dc2_word) == 4) *((DC2_AS3 dc2_v4u *)(rec + t * 8) + 1) = ones;
    }
    return OTri{t, 0};
  }
  DC2_DEV inline OTri sym(OTri a) const {
    // (a neighbour word that is followed is never the "none" word - the walks only cross edges that have a triangle
    // on the other side, tools/emulate counts the exceptions: zero - so the whole-node form reads it as it is)
    const int32_t e = (MAPPED || sizeof(dc2_word) != 2) ? ld(a.t, a.o) : (int32_t)rec[a.t * 8 + a.o];
    return OTri{e >> 2, e & 3};
  }
  DC2_DEV static inline OTri lnext(OTri a) { return OTri{a.t, a.o == 2 ? 0 : a.o + 1}; }
  DC2_DEV static inline OTri lprev(OTri a) { return OTri{a.t, a.o == 0 ? 2 : a.o - 1}; }
  // vertex words: point numbers stay below 2^15 (at most DC2_CACHE_PTS local points), so the sign-extending 16-bit read
  // turns the all-ones "no vertex" into -1 by itself - no compare and select behind every read of the seam walk
  DC2_DEV inline int32_t ldv(int32_t t, int w) const {
    if (MAPPED || sizeof(dc2_word) != 2) return ld(t, w);
    return (int32_t)(int16_t)rec[t * 8 + w];
  }
  DC2_DEV inline int32_t org(OTri a) const { return ldv(a.t, 4 + (a.o == 2 ? 0 : a.o + 1)); }
  DC2_DEV inline int32_t dest(OTri a) const { return ldv(a.t, 4 + (a.o == 0 ? 2 : a.o - 1)); }
  DC2_DEV inline int32_t apex(OTri a) const { return ldv(a.t, 4 + a.o); }
  DC2_DEV inline void set_org(OTri a, int32_t v) const { st(a.t, 4 + (a.o == 2 ? 0 : a.o + 1), v); }
  DC2_DEV inline void set_dest(OTri a, int32_t v) const { st(a.t, 4 + (a.o == 0 ? 2 : a.o - 1), v); }
  DC2_DEV inline void set_apex(OTri a, int32_t v) const { st(a.t, 4 + a.o, v); }
  DC2_DEV inline void bond(OTri a, OTri b) const {
    st(a.t, a.o, b.t * 4 + b.o);
    st(b.t, b.o, a.t * 4 + a.o);
  }
#ifdef DC2_COUNT_NULL_ACCESS
#define DC2_NULL_PT(x) if ((x) < 0) dc2_null_pts++
#else
#define DC2_NULL_PT(x)
#endif
  DC2_DEV inline int32_t px(int32_t p) const { DC2_NULL_PT(p); return (int32_t)(pt[p] & 0xffffu); }
  DC2_DEV inline int32_t py(int32_t p) const { DC2_NULL_PT(p); return (int32_t)(pt[p] >> 16); }
  // the seam loop's form: coordinates in registers; differences < 2^15, so the 32-bit products are 24-bit multiplies
  // (full rate; a 32-bit integer multiply is quarter rate)
  DC2_DEV inline uint32_t P(int32_t p) const { DC2_NULL_PT(p); return pt[p]; }
  DC2_DEV static inline int32_t mul24(int32_t a, int32_t b) {
#ifdef __HIP_DEVICE_COMPILE__
    return __mul24(a, b);
#else
    return a * b;
#endif
  }
  DC2_DEV static inline int32_t ccw_p(uint32_t pa, uint32_t pb, uint32_t pc) {
    const int32_t cx = (int32_t)(pc & 0xffffu), cy = (int32_t)(pc >> 16);
    return mul24((int32_t)(pa & 0xffffu) - cx, (int32_t)(pb >> 16) - cy) - mul24((int32_t)(pa >> 16) - cy, (int32_t)(pb & 0xffffu) - cx);
  }
  DC2_DEV static inline int64_t incircle_p(uint32_t pa, uint32_t pb, uint32_t pc, uint32_t pd) {
    const int32_t dx = (int32_t)(pd & 0xffffu), dy = (int32_t)(pd >> 16);
    const int32_t adx = (int32_t)(pa & 0xffffu) - dx, ady = (int32_t)(pa >> 16) - dy;
    const int32_t bdx = (int32_t)(pb & 0xffffu) - dx, bdy = (int32_t)(pb >> 16) - dy;
    const int32_t cdx = (int32_t)(pc & 0xffffu) - dx, cdy = (int32_t)(pc >> 16) - dy;
    return (int64_t)(mul24(adx, adx) + mul24(ady, ady)) * (mul24(bdx, cdy) - mul24(cdx, bdy)) +
           (int64_t)(mul24(bdx, bdx) + mul24(bdy, bdy)) * (mul24(cdx, ady) - mul24(adx, cdy)) +
           (int64_t)(mul24(cdx, cdx) + mul24(cdy, cdy)) * (mul24(adx, bdy) - mul24(bdx, ady));
  }
  DC2_DEV inline int32_t ccw(int32_t a, int32_t b, int32_t c) const {
    DC2_NULL_PT(a); DC2_NULL_PT(b); DC2_NULL_PT(c);
    const uint32_t pa = pt[a], pb = pt[b], pc = pt[c];
    const int32_t cx = (int32_t)(pc & 0xffffu), cy = (int32_t)(pc >> 16);
    return ((int32_t)(pa & 0xffffu) - cx) * ((int32_t)(pb >> 16) - cy) - ((int32_t)(pa >> 16) - cy) * ((int32_t)(pb & 0xffffu) - cx);
  }
  DC2_DEV inline int64_t incircle(int32_t a, int32_t b, int32_t c, int32_t d) const {
    DC2_NULL_PT(a); DC2_NULL_PT(b); DC2_NULL_PT(c); DC2_NULL_PT(d);
    const uint32_t pa = pt[a], pb = pt[b], pc = pt[c], pd = pt[d];
    const int32_t dx = (int32_t)(pd & 0xffffu), dy = (int32_t)(pd >> 16);
    const int32_t adx = (int32_t)(pa & 0xffffu) - dx, ady = (int32_t)(pa >> 16) - dy;
    const int32_t bdx = (int32_t)(pb & 0xffffu) - dx, bdy = (int32_t)(pb >> 16) - dy;
    const int32_t cdx = (int32_t)(pc & 0xffffu) - dx, cdy = (int32_t)(pc >> 16) - dy;
    return (int64_t)(adx * adx + ady * ady) * (bdx * cdy - cdx * bdy) + (int64_t)(bdx * bdx + bdy * bdy) * (cdx * ady - adx * cdy) +
           (int64_t)(cdx * cdx + cdy * cdy) * (adx * bdy - bdx * ady);
  }
};

// the block form adds what the leaves need: their keys (LDS copy), the point array writable, ids straight to global memory
struct DcBlockMesh : DcLdsMesh<false> {
  DC2_AS3 uint64_t *key;
  DC2_AS3 uint32_t *ptw;
  int32_t *gid;  // global id array at the block's first position
  DC2_DEV inline DC2_AS3 uint64_t &key_at(int32_t i) const { return key[i]; }
  DC2_DEV inline void put_point(int32_t i, uint32_t p, int32_t idv) const {
    ptw[i] = p;
    gid[i] = idv;
  }
};


// ---- lane roles inside a block sub-tree (k_dc2_block): <= 64 leaves of <= DC2_BLOCK_LEAF points, one per lane, then the
// merge levels 5 .. 0 with 32, 16, ... 1 lanes ----
#define DC2_BLOCK_LEAF 14
#define DC2_BLOCK_DEPTH 6  // ceil(480 / 2^6) <= 14
struct Dc2Hull16 {
  int16_t fl_t, fl_o, fr_t, fr_o;
};
// the leaf of `lane`: lane bits choose the path from the block's root, most significant first; a leaf reached early is
// taken by the lane whose remaining bits are zero.  idx = heap index inside the block (root 1).
DC2_DEV inline bool dc2_block_leaf_of(int lane, int32_t bn, int baxis, int32_t &off, int32_t &n, int &axis, int &idx) {
  off = 0;
  n = bn;
  axis = baxis;
  idx = 1;
  for (int b = DC2_BLOCK_DEPTH - 1; b >= 0; b--) {
    if (n <= DC2_BLOCK_LEAF) return (lane & ((2 << b) - 1)) == 0;
    const int32_t div = n >> 1;
    if ((lane >> b) & 1) {
      off += div;
      n -= div;
      idx = 2 * idx + 1;
    } else {
      n = div;
      idx = 2 * idx;
    }
    axis = 1 - axis;
  }
  return true;
}
// the merge node of `lane` (< 2^L) on level L, if that node exists (it does not below a leaf reached early)
DC2_DEV inline bool dc2_block_merge_of(int lane, int L, int32_t bn, int baxis, int32_t &off, int32_t &n, int &axis, int &idx) {
  off = 0;
  n = bn;
  axis = baxis;
  idx = 1;
  for (int b = L - 1; b >= 0; b--) {
    if (n <= DC2_BLOCK_LEAF) return false;
    const int32_t div = n >> 1;
    if ((lane >> b) & 1) {
      off += div;
      n -= div;
      idx = 2 * idx + 1;
    } else {
      n = div;
      idx = 2 * idx;
    }
    axis = 1 - axis;
  }
  return n > DC2_BLOCK_LEAF;
}
template <class M>
DC2_DEV inline void dc2_block_leaf_run(const M &mesh, int lane, int32_t bn, int baxis, DC2_AS3 Dc2Hull16 *hull) {
  int32_t off, n;
  int axis, idx;
  if (!dc2_block_leaf_of(lane, bn, baxis, off, n, axis, idx)) return;
  DcOTri fl, fr;
  dc_build_small(mesh, off, n, axis, fl, fr);  // (<= DC2_BLOCK_LEAF points: at most three cut levels)
  hull[idx] = Dc2Hull16{(int16_t)fl.t, (int16_t)fl.o, (int16_t)fr.t, (int16_t)fr.o};
}
template <class M>
DC2_DEV inline void dc2_block_merge_run(const M &mesh, int lane, int L, int32_t bn, int baxis, DC2_AS3 Dc2Hull16 *hull) {
  int32_t off, n;
  int axis, idx;
  if (lane >= (1 << L) || !dc2_block_merge_of(lane, L, bn, baxis, off, n, axis, idx)) return;
  const Dc2Hull16 l = hull[2 * idx], r = hull[2 * idx + 1];
  DcOTri fl{l.fl_t, l.fl_o}, il{l.fr_t, l.fr_o}, ir{r.fl_t, r.fl_o}, fr{r.fr_t, r.fr_o};
  int32_t tcur = 2 * (off + (n >> 1)) - 2;
  dc_merge_hulls(mesh, fl, il, ir, fr, axis, tcur);
  hull[idx] = Dc2Hull16{(int16_t)fl.t, (int16_t)fl.o, (int16_t)fr.t, (int16_t)fr.o};
}
