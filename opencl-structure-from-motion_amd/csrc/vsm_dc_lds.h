// The LDS mesh of the GPU-resident Delaunay kernels (vsm_dc.hip), in a header of its own so that a host build can run
// the very same code (tools/emulate/dc_lds_emulate.cpp walks the block kernel's lanes one after the other).
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
typedef uint16_t dc2_u16a __attribute__((may_alias));
typedef uint32_t dc2_u32a __attribute__((may_alias));

// A sub-triangulation in LDS under LOCAL numbering (positions and slots count from the node's - or the band's - first
// one), 16 bytes per triangle, as EDGE WORDS: dword o (0..2) of triangle t describes the oriented triangle (t, o) -
//     bits  0..15  the neighbour across edge o as a handle t' * 4 + o'   (0xffff: none; band form: the trap handle `rim` = outside the band)
//     bits 16..31  the apex of (t, o), i.e. vertex o of the triangle       (0xffff: the ghost corner)
// dword 3 is unused.  A handle h = t * 4 + o IS the word's index, so sym() and apex() of a handle are one 32-bit read at
// h * 4 with nothing to compute, a rotation is an add on the handle, a bond is a 16-bit store.  Half the bytes of the
// 32-bit records of the host / global-memory form (DcMesh), ds_ instructions with the handle as address, nothing to
// translate while a seam is walked: global numbering comes back when the records are written out.
#define DC2_NONE16 0xffffu
// Band form (GUARDED): a neighbour that is not in the band - or does not exist - is the handle `rim` of a reserved TRAP
// record behind the band's last line, whose own words point to itself.  Outside the seam loop following it trips the guard
// at once (nb_of); inside the loop nothing is tested per access: a check that would need the triangle behind the rim
// poisons itself (na = -2, looked at where the check is decided), and every other way onto the trap ends in a store
// into it (bonds always write both sides), which the caller finds afterwards (trap_touched) - so the walk either stayed
// inside the band or is redone on the records in global memory; a step budget bounds a walk that went astray.

template <bool GUARDED>
struct DcEdgeMesh {
  typedef DcOTri OTri;
  static constexpr bool kGuarded = GUARDED;
  static constexpr bool kFastZip = true;
  DC2_AS3 dc2_u32a *w;         // edge words, 4 per triangle
  DC2_AS3 const uint32_t *pt;  // x | y << 16 by (local) point number
  mutable int32_t tripped = 0;
  int32_t rim = 0xffff;        // GUARDED: handle of the trap record (4 * its line)
  int32_t budget = 1 << 30;    // GUARDED: seam steps + dissolved edges a walk may take
  DC2_DEV inline bool ok() const { return !GUARDED || tripped == 0; }
  DC2_DEV inline bool is_rim(uint32_t e) const { return GUARDED && (int32_t)e == rim; }
  DC2_DEV inline void make_trap() const {  // (GUARDED) the trap record: three self-pointing edges, ghost corners
    for (int o = 0; o < 4; o++) w[rim + o] = (uint32_t)rim | 0xffff0000u;
  }
  DC2_DEV inline bool trap_touched() const {
    bool t = false;
    for (int o = 0; o < 4; o++) t |= w[rim + o] != ((uint32_t)rim | 0xffff0000u);
    return t;
  }

  // ---- on handles (the seam walk's form) ----
  DC2_DEV inline uint32_t ldw(int32_t h) const { return w[h]; }
  DC2_DEV inline void stw(int32_t h, uint32_t v) const { w[h] = v; }
  DC2_DEV inline void stn(int32_t h, int32_t nb) const { ((DC2_AS3 dc2_u16a *)(w + h))[0] = (uint16_t)nb; }
  DC2_DEV inline void stv(int32_t h, int32_t v) const { ((DC2_AS3 dc2_u16a *)(w + h))[1] = (uint16_t)v; }
  DC2_DEV static inline int32_t hnext(int32_t h) { return (h & 3) == 2 ? h - 2 : h + 1; }
  DC2_DEV static inline int32_t hprev(int32_t h) { return (h & 3) == 0 ? h + 2 : h - 1; }
  // neighbour part of a word that is going to be followed; a guarded mesh refuses the marker of the band's rim
  DC2_DEV inline int32_t nb_of(uint32_t word) const {
    const uint32_t e = word & 0xffffu;
    if (GUARDED && (int32_t)e == rim) {
      tripped = 1;
      return 0;
    }
    return (int32_t)e;
  }
  DC2_DEV static inline int32_t nb_raw(uint32_t word) { return (int32_t)(word & 0xffffu); }  // (seam loop: see above)
  DC2_DEV static inline int32_t vx_of(uint32_t word) { return (int32_t)word >> 16; }  // (point numbers stay below 2^15: the ghost corner becomes -1 by itself)
  DC2_DEV inline uint32_t P(int32_t p) const { return pt[p]; }

  // ---- the generic interface of vsm_dc_mesh.h (leaves, and dc_merge_hulls as the reference form) ----
  DC2_DEV inline OTri make(int32_t &tcur) const {
    const int32_t t = tcur++;
    dc2_v4u ones;
    ones.x = ones.y = ones.z = ones.w = 0xffffffffu;
    *(DC2_AS3 dc2_v4u *)(w + 4 * t) = ones;
    return OTri{t, 0};
  }
  DC2_DEV inline OTri sym(OTri a) const {
    const int32_t e = nb_of(w[a.t * 4 + a.o]);
    return OTri{e >> 2, e & 3};
  }
  DC2_DEV static inline OTri lnext(OTri a) { return OTri{a.t, a.o == 2 ? 0 : a.o + 1}; }
  DC2_DEV static inline OTri lprev(OTri a) { return OTri{a.t, a.o == 0 ? 2 : a.o - 1}; }
  DC2_DEV inline int32_t org(OTri a) const { return vx_of(w[a.t * 4 + (a.o == 2 ? 0 : a.o + 1)]); }
  DC2_DEV inline int32_t dest(OTri a) const { return vx_of(w[a.t * 4 + (a.o == 0 ? 2 : a.o - 1)]); }
  DC2_DEV inline int32_t apex(OTri a) const { return vx_of(w[a.t * 4 + a.o]); }
  DC2_DEV inline void set_org(OTri a, int32_t v) const { stv(a.t * 4 + (a.o == 2 ? 0 : a.o + 1), v); }
  DC2_DEV inline void set_dest(OTri a, int32_t v) const { stv(a.t * 4 + (a.o == 0 ? 2 : a.o - 1), v); }
  DC2_DEV inline void set_apex(OTri a, int32_t v) const { stv(a.t * 4 + a.o, v); }
  DC2_DEV inline void bond(OTri a, OTri b) const {
    stn(a.t * 4 + a.o, b.t * 4 + b.o);
    stn(b.t * 4 + b.o, a.t * 4 + a.o);
  }
  DC2_DEV inline int32_t px(int32_t p) const { return (int32_t)(pt[p] & 0xffffu); }
  DC2_DEV inline int32_t py(int32_t p) const { return (int32_t)(pt[p] >> 16); }
  // coordinates in registers; differences < 2^15, so the 32-bit products are 24-bit multiplies (full rate)
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
  // sign of the in-circle determinant.  Every lifted length and every cross product is an integer below 2^31 and their
  // products sum to less than 2^53 in magnitude (coordinates < 2^14): exact in double, three fused multiply-adds
  // instead of three 64-bit integer multiply-adds of five instructions each
  DC2_DEV static inline int32_t incircle_s(uint32_t pa, uint32_t pb, uint32_t pc, uint32_t pd) {
    const int32_t dx = (int32_t)(pd & 0xffffu), dy = (int32_t)(pd >> 16);
    const int32_t adx = (int32_t)(pa & 0xffffu) - dx, ady = (int32_t)(pa >> 16) - dy;
    const int32_t bdx = (int32_t)(pb & 0xffffu) - dx, bdy = (int32_t)(pb >> 16) - dy;
    const int32_t cdx = (int32_t)(pc & 0xffffu) - dx, cdy = (int32_t)(pc >> 16) - dy;
#ifdef __HIP_DEVICE_COMPILE__
    const double l1 = (double)(mul24(adx, adx) + mul24(ady, ady)), l2 = (double)(mul24(bdx, bdx) + mul24(bdy, bdy)), l3 = (double)(mul24(cdx, cdx) + mul24(cdy, cdy));
    const double c1 = (double)(mul24(bdx, cdy) - mul24(cdx, bdy)), c2 = (double)(mul24(cdx, ady) - mul24(adx, cdy)), c3 = (double)(mul24(adx, bdy) - mul24(bdx, ady));
    const double det = __builtin_fma(l1, c1, __builtin_fma(l2, c2, l3 * c3));
    return det > 0 ? 1 : (det < 0 ? -1 : 0);
#else
    const int64_t det = (int64_t)(adx * adx + ady * ady) * (bdx * cdy - cdx * bdy) + (int64_t)(bdx * bdx + bdy * bdy) * (cdx * ady - adx * cdy) +
                        (int64_t)(cdx * cdx + cdy * cdy) * (adx * bdy - bdx * ady);
    return det > 0 ? 1 : (det < 0 ? -1 : 0);
#endif
  }
  DC2_DEV static inline int64_t incircle_p(uint32_t pa, uint32_t pb, uint32_t pc, uint32_t pd) { return incircle_s(pa, pb, pc, pd); }
  // ... and only "is d strictly inside the circle through a, b, c" (what the seam walk asks): the three products as
  // v_mad_i64_i32, one instruction each (the compiler's own 64-bit multiply of a known-non-negative by a signed 32-bit value
  // comes out as two v_mad_u64_u32 plus moves; DC2_INCIRCLE_F64 selects the same sum in double, exact as well and 7 % slower).
  DC2_DEV static inline bool incircle_in(uint32_t pa, uint32_t pb, uint32_t pc, uint32_t pd) {
    const int32_t dx = (int32_t)(pd & 0xffffu), dy = (int32_t)(pd >> 16);
    const int32_t adx = (int32_t)(pa & 0xffffu) - dx, ady = (int32_t)(pa >> 16) - dy;
    const int32_t bdx = (int32_t)(pb & 0xffffu) - dx, bdy = (int32_t)(pb >> 16) - dy;
    const int32_t cdx = (int32_t)(pc & 0xffffu) - dx, cdy = (int32_t)(pc >> 16) - dy;
#if defined(__HIP_DEVICE_COMPILE__) && !defined(DC2_INCIRCLE_F64)
    const int32_t l1 = mul24(adx, adx) + mul24(ady, ady), l2 = mul24(bdx, bdx) + mul24(bdy, bdy), l3 = mul24(cdx, cdx) + mul24(cdy, cdy);
    const int32_t c1 = mul24(bdx, cdy) - mul24(cdx, bdy), c2 = mul24(cdx, ady) - mul24(adx, cdy), c3 = mul24(adx, bdy) - mul24(bdx, ady);
    int64_t det;
    uint64_t carry;
    asm("v_mad_i64_i32 %0, %1, %2, %3, 0" : "=v"(det), "=s"(carry) : "v"(l1), "v"(c1));
    asm("v_mad_i64_i32 %0, %1, %2, %3, %4" : "=v"(det), "=s"(carry) : "v"(l2), "v"(c2), "v"(det));
    asm("v_mad_i64_i32 %0, %1, %2, %3, %4" : "=v"(det), "=s"(carry) : "v"(l3), "v"(c3), "v"(det));
    return det > 0;
#elif defined(__HIP_DEVICE_COMPILE__)
    const double l1 = (double)(mul24(adx, adx) + mul24(ady, ady)), l2 = (double)(mul24(bdx, bdx) + mul24(bdy, bdy)), l3 = (double)(mul24(cdx, cdx) + mul24(cdy, cdy));
    const double c1 = (double)(mul24(bdx, cdy) - mul24(cdx, bdy)), c2 = (double)(mul24(cdx, ady) - mul24(adx, cdy)), c3 = (double)(mul24(adx, bdy) - mul24(bdx, ady));
    return __builtin_fma(l1, c1, __builtin_fma(l2, c2, l3 * c3)) > 0;
#else
    return (int64_t)(adx * adx + ady * ady) * (bdx * cdy - cdx * bdy) + (int64_t)(bdx * bdx + bdy * bdy) * (cdx * ady - adx * cdy) +
               (int64_t)(cdx * cdx + cdy * cdy) * (adx * bdy - bdx * ady) >
           0;
#endif
  }
  // rotation of a handle inside its triangle: dir = true -> lnext, false -> lprev (one code path for both)
  DC2_DEV static inline int32_t hrot(int32_t h, bool next) {
    const uint32_t lut = next ? 0x09u : 0x12u;  // o -> o + 1 mod 3 / o - 1 mod 3, two bits per o
    return (h & ~3) | (int32_t)((lut >> ((h & 3) << 1)) & 3u);
  }
  DC2_DEV inline int32_t ccw(int32_t a, int32_t b, int32_t c) const { return ccw_p(pt[a], pt[b], pt[c]); }
  DC2_DEV inline int64_t incircle(int32_t a, int32_t b, int32_t c, int32_t d) const { return incircle_s(pt[a], pt[b], pt[c], pt[d]); }
};

// the band form of the merge levels (k_dc2_merge): a guarded mesh
typedef DcEdgeMesh<true> DcBandMesh;

// the block form adds what the leaves need: their keys (read where they lie, in global memory), the point array
// writable, ids straight to global memory
struct DcBlockMesh : DcEdgeMesh<false> {
  uint64_t *key;  // the block's keys in kd order (global memory; a leaf reads its two or three and puts them back in x order)
  DC2_AS3 uint32_t *ptw;
  int32_t *gid;  // global id array at the block's first position
  DC2_DEV inline uint64_t &key_at(int32_t i) const { return key[i]; }
  DC2_DEV inline void put_point(int32_t i, uint32_t p, int32_t idv) const {
    ptw[i] = p;
    gid[i] = idv;
  }
};

// ---------------------------------------------------------------------------------------
// mergehulls (viso/triangle.cpp:5639-5960) once more, decision for decision and access for access like dc_merge_hulls
// (vsm_dc_mesh.h), written for the edge words: handles are plain integers, the packed coordinates of every point in play
// travel in registers, a record that changes is written with as few stores as its words allow (a dissolved edge: nine
// stores instead of eighteen), and what a side's candidate check reads (the triangle behind the candidate edge, its apex,
// the apex's coordinates) is kept across seam steps that do not touch that side - there only the in-circle test depends
// on the seam's other end.  Reads that the reference form makes after a store stay after it (a dissolved edge reads the
// outer neighbour behind the first two bonds).  tools/emulate checks the result slot for slot against ExactDelaunay on
// the host, the GPU tests against the reference's lists.
// ---------------------------------------------------------------------------------------
template <class M>
VSM_HD inline void dc2_zip(const M &m, DcOTri &farleft_, DcOTri &innerleft_, DcOTri &innerright_, DcOTri &farright_, int axis, int32_t &tcur) {
  int32_t farleft = farleft_.t * 4 + farleft_.o, innerleft = innerleft_.t * 4 + innerleft_.o;
  int32_t innerright = innerright_.t * 4 + innerright_.o, farright = farright_.t * 4 + farright_.o;
#define DC2_RET()                                          \
  do {                                                     \
    farleft_ = DcOTri{farleft >> 2, farleft & 3};          \
    farright_ = DcOTri{farright >> 2, farright & 3};       \
    innerleft_ = DcOTri{innerleft >> 2, innerleft & 3};    \
    innerright_ = DcOTri{innerright >> 2, innerright & 3}; \
    return;                                                \
  } while (0)
  // corner points of the two inner handles
  int32_t ildest = M::vx_of(m.ldw(M::hprev(innerleft))), ilapex = M::vx_of(m.ldw(innerleft));
  int32_t irorg = M::vx_of(m.ldw(M::hnext(innerright))), irapex = M::vx_of(m.ldw(innerright));
  if (axis == 1) {  // horizontal cut: handles move to the bottom-/top-most hull vertices (:5666)
    int32_t flpt = M::vx_of(m.ldw(M::hnext(farleft))), flapex = M::vx_of(m.ldw(farleft));
    int32_t frpt = M::vx_of(m.ldw(M::hprev(farright)));
    uint32_t pf = m.P(flpt), pa = m.P(flapex);
    while ((pa >> 16) < (pf >> 16)) {
      if (M::kGuarded && !m.ok()) DC2_RET();
      farleft = m.nb_of(m.ldw(M::hnext(farleft)));
      flpt = flapex;
      pf = pa;
      flapex = M::vx_of(m.ldw(farleft));
      pa = m.P(flapex);
    }
    int32_t chk = m.nb_of(m.ldw(innerleft));
    int32_t cv = M::vx_of(m.ldw(chk));
    uint32_t pil = m.P(ildest), pcv = m.P(cv);
    while ((pcv >> 16) > (pil >> 16)) {
      if (M::kGuarded && !m.ok()) DC2_RET();
      innerleft = M::hnext(chk);
      ilapex = ildest;
      ildest = cv;
      pil = pcv;
      chk = m.nb_of(m.ldw(innerleft));
      cv = M::vx_of(m.ldw(chk));
      pcv = m.P(cv);
    }
    uint32_t pio = m.P(irorg), pia = m.P(irapex);
    while ((pia >> 16) < (pio >> 16)) {
      if (M::kGuarded && !m.ok()) DC2_RET();
      innerright = m.nb_of(m.ldw(M::hnext(innerright)));
      irorg = irapex;
      pio = pia;
      irapex = M::vx_of(m.ldw(innerright));
      pia = m.P(irapex);
    }
    chk = m.nb_of(m.ldw(farright));
    cv = M::vx_of(m.ldw(chk));
    uint32_t pfr = m.P(frpt);
    pcv = m.P(cv);
    while ((pcv >> 16) > (pfr >> 16)) {
      if (M::kGuarded && !m.ok()) DC2_RET();
      farright = M::hnext(chk);
      frpt = cv;
      pfr = pcv;
      chk = m.nb_of(m.ldw(farright));
      cv = M::vx_of(m.ldw(chk));
      pcv = m.P(cv);
    }
  }
  {  // lower common tangent (:5704)
    uint32_t pild = m.P(ildest), pila = m.P(ilapex), piro = m.P(irorg), pira = m.P(irapex);
    bool changed;
    do {
      if (M::kGuarded && !m.ok()) DC2_RET();
      changed = false;
      if (M::ccw_p(pild, pila, piro) > 0) {
        innerleft = m.nb_of(m.ldw(M::hprev(innerleft)));
        ildest = ilapex;
        pild = pila;
        ilapex = M::vx_of(m.ldw(innerleft));
        pila = m.P(ilapex);
        changed = true;
      }
      if (M::ccw_p(pira, piro, pild) > 0) {
        innerright = m.nb_of(m.ldw(M::hnext(innerright)));
        irorg = irapex;
        piro = pira;
        irapex = M::vx_of(m.ldw(innerright));
        pira = m.P(irapex);
        changed = true;
      }
    } while (changed);
  }
  int32_t leftcand = m.nb_of(m.ldw(innerleft)), rightcand = m.nb_of(m.ldw(innerright));
  // the bottom bounding triangle in slot tcur: edge 0 to innerleft, edge 1 to innerright; org of edge 2 = irorg is vertex 0,
  // its dest = ildest vertex 1, its apex the ghost
  int32_t base = 4 * tcur;
  tcur++;
  m.stw(base, (uint32_t)innerleft | ((uint32_t)(uint16_t)irorg << 16));
  m.stw(base + 1, (uint32_t)innerright | ((uint32_t)(uint16_t)ildest << 16));
  m.stw(base + 2, 0xffffffffu);
  m.stw(base + 3, 0xffffffffu);
  m.stn(innerleft, base);
  m.stn(innerright, base + 1);
  base += 2;
  if (ildest == M::vx_of(m.ldw(M::hnext(farleft)))) farleft = M::hnext(base);
  if (irorg == M::vx_of(m.ldw(M::hprev(farright)))) farright = M::hprev(base);
  int32_t ll = ildest, lr = irorg;
  int32_t ul = M::vx_of(m.ldw(leftcand)), ur = M::vx_of(m.ldw(rightcand));
  uint32_t pll = m.P(ll), plr = m.P(lr), pul = m.P(ul), pur = m.P(ur);
  // What a side's candidate check reads - the triangle behind the candidate edge, its apex, the apex's coordinates - is
  // fetched as soon as the candidate is known (with the loads that make it known) and kept until that side moves; the
  // words involved are only ever written by that side's own steps.  na == -2: the triangle behind the edge is outside the
  // band (only a check that really needs it trips the guard).  The loop body is straight-line code up to the (rare)
  // dissolving of edges: three in-circle signs and two orientations from registers, one select-driven advance, one chain
  // of four dependent LDS reads - and it is the same code whichever side moves, so lanes walking different seams stay
  // together.
  int32_t lne, lna, rne, rna, budget = m.budget;
  uint32_t lpna, rpna;
  {
    const uint32_t wl = m.ldw(M::hprev(leftcand)), wr = m.ldw(M::hnext(rightcand));
    const bool xl = m.is_rim(wl & 0xffffu), xr = m.is_rim(wr & 0xffffu);
    lne = xl ? 0 : (int32_t)(wl & 0xffffu);
    rne = xr ? 0 : (int32_t)(wr & 0xffffu);
    lna = xl ? -2 : M::vx_of(m.ldw(lne));
    rna = xr ? -2 : M::vx_of(m.ldw(rne));
    lpna = m.P(lna < 0 ? 0 : lna);
    rpna = m.P(rna < 0 ? 0 : rna);
  }
  for (;;) {
    // (bitwise logic on purpose: every test is evaluated, nothing here branches but the loop's exit and the dissolving)
    const bool lfin = M::ccw_p(pul, pll, plr) <= 0, rfin = M::ccw_p(pur, pll, plr) <= 0;
    const bool lin = M::incircle_in(pll, plr, pul, lpna), rin = M::incircle_in(pll, plr, pur, rpna);
    bool cin = M::incircle_in(pul, pll, plr, pur);
    if (M::kGuarded) m.tripped |= (int32_t)((!lfin & (lna == -2)) | (!rfin & (rna == -2)) | (--budget < 0));
    if ((lfin & rfin) | (M::kGuarded & !m.ok())) break;
    const bool lbad = !lfin & (lna >= 0) & lin, rbad = !rfin & (rna >= 0) & rin;
    if (lbad | rbad) {
      if (lbad) {  // dissolve non-Delaunay edges on the left (:5814)
        do {
          if (M::kGuarded && --budget < 0) break;
          const int32_t ne1 = M::hnext(lne), ne2 = M::hprev(lne);
          const int32_t topc = M::nb_raw(m.ldw(ne1)), sidec = M::nb_raw(m.ldw(ne2));
          const int32_t lc1 = M::hnext(leftcand), lc2 = M::hprev(leftcand);
          // bond(ne2, topc), bond(leftcand, sidec) - then the outer neighbour is read, as the reference form does
          m.stw(ne2, (uint32_t)topc | 0xffff0000u);  // (its vertex becomes the ghost: set_org(ne1, -1))
          m.stn(topc, ne2);
          m.stw(leftcand, (uint32_t)sidec | 0xffff0000u);  // (set_dest(lc1, -1))
          m.stn(sidec, leftcand);
          const int32_t outerc = M::nb_raw(m.ldw(lc1));
          m.stw(ne1, (uint32_t)outerc | ((uint32_t)(uint16_t)lna << 16));  // bond(ne1, outerc), set_apex(ne1, na)
          m.stn(outerc, ne1);
          m.stv(lc2, ll);   // set_org(lc1, ll)
          m.stv(lc1, lna);  // set_apex(lc1, na)
          m.stv(lne, ul);   // set_dest(ne1, ul)
          leftcand = lc1;
          ul = lna;
          pul = lpna;
          lne = sidec;
          lna = M::vx_of(m.ldw(lne));
          lpna = m.P(lna < 0 ? 0 : lna);
        } while ((lna >= 0) & M::incircle_in(pll, plr, pul, lpna));
      }
      if (rbad) {  // ... and on the right (:5862)
        do {
          if (M::kGuarded && --budget < 0) break;
          const int32_t ne1 = M::hprev(rne), ne2 = M::hnext(rne);
          const int32_t topc = M::nb_raw(m.ldw(ne1)), sidec = M::nb_raw(m.ldw(ne2));
          const int32_t rc1 = M::hprev(rightcand), rc2 = M::hnext(rightcand);
          m.stw(ne2, (uint32_t)topc | 0xffff0000u);  // bond(ne2, topc); set_dest(ne1, -1)
          m.stn(topc, ne2);
          m.stw(rightcand, (uint32_t)sidec | 0xffff0000u);  // bond(rightcand, sidec); set_org(rc1, -1)
          m.stn(sidec, rightcand);
          const int32_t outerc = M::nb_raw(m.ldw(rc1));
          m.stw(ne1, (uint32_t)outerc | ((uint32_t)(uint16_t)rna << 16));  // bond(ne1, outerc), set_apex(ne1, na)
          m.stn(outerc, ne1);
          m.stv(rc2, lr);   // set_dest(rc1, lr)
          m.stv(rc1, rna);  // set_apex(rc1, na)
          m.stv(rne, ur);   // set_org(ne1, ur)
          rightcand = rc1;
          ur = rna;
          pur = rpna;
          rne = sidec;
          rna = M::vx_of(m.ldw(rne));
          rpna = m.P(rna < 0 ? 0 : rna);
        } while ((rna >= 0) & M::incircle_in(pll, plr, pur, rpna));
      }
      cin = M::incircle_in(pul, pll, plr, pur);  // (the corners have moved)
    }
    // the new cross edge: ll--ur (:5911) or ul--lr (:5920); (lfin, rfin are the flags from before the dissolving, the
    // corners are the ones behind it - as in the reference form)
    const bool take_r = lfin | (!rfin & cin);
    const int32_t cand = take_r ? rightcand : leftcand;
    m.stn(base, cand);
    m.stn(cand, base);
    base = M::hrot(cand, !take_r);                        // lprev(rightcand) / lnext(leftcand)
    m.stv(M::hrot(base, !take_r), take_r ? ll : lr);      // set_dest(base, ll) / set_org(base, lr)
    const int32_t ncand = M::nb_raw(m.ldw(base));
    const uint32_t wc = m.ldw(ncand), wn = m.ldw(M::hrot(ncand, take_r));  // the check reads lnext(rightcand) / lprev(leftcand)
    const int32_t nup = M::vx_of(wc);
    const bool xn = m.is_rim(wn & 0xffffu);
    const int32_t nne = xn ? 0 : (int32_t)(wn & 0xffffu);
    const uint32_t pnup = m.P(nup);
    const int32_t nna = xn ? -2 : M::vx_of(m.ldw(nne));
    const uint32_t pnna = m.P(nna < 0 ? 0 : nna);
    lr = take_r ? ur : lr;
    plr = take_r ? pur : plr;
    ll = take_r ? ll : ul;
    pll = take_r ? pll : pul;
    rightcand = take_r ? ncand : rightcand;
    leftcand = take_r ? leftcand : ncand;
    ur = take_r ? nup : ur;
    pur = take_r ? pnup : pur;
    ul = take_r ? ul : nup;
    pul = take_r ? pul : pnup;
    rne = take_r ? nne : rne;
    rna = take_r ? nna : rna;
    rpna = take_r ? pnna : rpna;
    lne = take_r ? lne : nne;
    lna = take_r ? lna : nna;
    lpna = take_r ? lpna : pnna;
  }
  if (M::kGuarded && !m.ok()) DC2_RET();
  {  // close the seam with the top bounding triangle (:5771): org = ll is vertex 1, dest = lr vertex 2; edge 0 to base, 1 to rightcand, 2 to leftcand
    const int32_t top = 4 * tcur;
    tcur++;
    m.stw(top, (uint32_t)base | 0xffff0000u);
    m.stw(top + 1, (uint32_t)rightcand | ((uint32_t)(uint16_t)ll << 16));
    m.stw(top + 2, (uint32_t)leftcand | ((uint32_t)(uint16_t)lr << 16));
    m.stw(top + 3, 0xffffffffu);
    m.stn(base, top);
    m.stn(rightcand, top + 1);
    m.stn(leftcand, top + 2);
  }
  if (axis == 1) {  // handles back to the left-/right-most vertices
    int32_t flpt = M::vx_of(m.ldw(M::hnext(farleft))), frpt = M::vx_of(m.ldw(M::hprev(farright))), frapex = M::vx_of(m.ldw(farright));
    int32_t chk = m.nb_of(m.ldw(farleft));
    int32_t cv = M::vx_of(m.ldw(chk));
    uint32_t pfl = m.P(flpt), pcv = m.P(cv);
    while ((pcv & 0xffffu) < (pfl & 0xffffu)) {
      if (M::kGuarded && !m.ok()) DC2_RET();
      farleft = M::hprev(chk);
      flpt = cv;
      pfl = pcv;
      chk = m.nb_of(m.ldw(farleft));
      cv = M::vx_of(m.ldw(chk));
      pcv = m.P(cv);
    }
    uint32_t pfr = m.P(frpt), pfa = m.P(frapex);
    while ((pfa & 0xffffu) > (pfr & 0xffffu)) {
      if (M::kGuarded && !m.ok()) DC2_RET();
      farright = m.nb_of(m.ldw(M::hprev(farright)));
      frpt = frapex;
      pfr = pfa;
      frapex = M::vx_of(m.ldw(farright));
      pfa = m.P(frapex);
    }
  }
  DC2_RET();
#undef DC2_RET
}

// ---- lane roles inside a block sub-tree (k_dc2_block): <= 2^DC2_BLOCK_DEPTH leaves of <= DC2_BLOCK_LEAF points, one per
// lane, then the merge levels DC2_BLOCK_DEPTH-1 .. 0 with 128, 64, ... 1 lanes.  Leaves are Triangle's own leaves (two or
// three points): every level of the tree above them is a row of seam walks that the lanes of a wave start together -
// sub-trees built lane by lane in post-order (14 points per lane, the earlier form) kept a quarter of the lanes busy with
// walks of different levels at the same time ----
#ifndef DC2_BLOCK_LEAF
#define DC2_BLOCK_LEAF 3
#define DC2_BLOCK_DEPTH 8  // ceil(480 / 2^8) <= 3
#endif
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
  dc_merge(mesh, fl, il, ir, fr, axis, tcur);
  hull[idx] = Dc2Hull16{(int16_t)fl.t, (int16_t)fl.o, (int16_t)fr.t, (int16_t)fr.o};
}
