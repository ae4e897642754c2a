// The divide-and-conquer core of ExactDelaunay (vsm_host.h) as a plain view over caller-owned
// arrays, usable from host and device code: triangle records, oriented-triangle handles, exact
// integer predicates, the leaf cases and mergehulls of Triangle 1.6 (viso/triangle.cpp:5639-6160),
// decision for decision.  Sub-problems own disjoint triangle slots (leaves of the slice starting at
// position `off` use slots 2*off.., the merge at boundary b uses 2b-2 and 2b-1), so any set of
// disjoint slices can be triangulated concurrently -- by host threads or by GPU threads.
#pragma once

#include <stdint.h>

#if defined(__HIPCC__)
#define VSM_HD __host__ __device__
#else
#define VSM_HD
#endif

#define VSM_KXY(k) ((k) >> 20)  // (x, y) part of a packed key (x << 34) | (y << 20) | input index

struct DcMesh {
  struct OTri {
    int32_t t, o;
  };
  // one 32-byte record per triangle: tri[t*8 + o] = neighbour handle across edge o,
  // tri[t*8 + 4 + o] = vertex o (-1 = ghost corner); pt[p] = x | y << 16 by sorted position
  int32_t *tri;
  uint32_t *pt;
  int32_t *id;    // by sorted position: input index
  uint64_t *key;  // by sorted position (kd order): packed keys; leaves put their 2-3 keys in x order

  VSM_HD static inline void swap_keys(uint64_t &a, uint64_t &b) {
    const uint64_t t = a;
    a = b;
    b = t;
  }
  VSM_HD inline OTri make(int32_t &tcur) const {
    const int32_t t = tcur++;
    int32_t *r = &tri[(size_t)t * 8];
    r[0] = r[1] = r[2] = -1;
    r[4] = r[5] = r[6] = -1;
    return OTri{t, 0};
  }
  VSM_HD inline OTri sym(OTri a) const {
    const int32_t e = tri[(size_t)a.t * 8 + a.o];
    return OTri{e >> 2, e & 3};
  }
  VSM_HD static inline OTri lnext(OTri a) { return OTri{a.t, a.o == 2 ? 0 : a.o + 1}; }
  VSM_HD static inline OTri lprev(OTri a) { return OTri{a.t, a.o == 0 ? 2 : a.o - 1}; }
  VSM_HD inline int32_t org(OTri a) const { return tri[(size_t)a.t * 8 + 4 + (a.o == 2 ? 0 : a.o + 1)]; }
  VSM_HD inline int32_t dest(OTri a) const { return tri[(size_t)a.t * 8 + 4 + (a.o == 0 ? 2 : a.o - 1)]; }
  VSM_HD inline int32_t apex(OTri a) const { return tri[(size_t)a.t * 8 + 4 + a.o]; }
  VSM_HD inline void set_org(OTri a, int32_t v) const { tri[(size_t)a.t * 8 + 4 + (a.o == 2 ? 0 : a.o + 1)] = v; }
  VSM_HD inline void set_dest(OTri a, int32_t v) const { tri[(size_t)a.t * 8 + 4 + (a.o == 0 ? 2 : a.o - 1)] = v; }
  VSM_HD inline void set_apex(OTri a, int32_t v) const { tri[(size_t)a.t * 8 + 4 + a.o] = v; }
  VSM_HD inline void bond(OTri a, OTri b) const {
    tri[(size_t)a.t * 8 + a.o] = b.t * 4 + b.o;
    tri[(size_t)b.t * 8 + b.o] = a.t * 4 + a.o;
  }
  VSM_HD inline int32_t px(int32_t p) const { return (int32_t)(pt[p] & 0xffffu); }
  VSM_HD inline int32_t py(int32_t p) const { return (int32_t)(pt[p] >> 16); }
  // coordinates < 2^14: the orientation determinant fits int32, the in-circle one int64
  VSM_HD inline int32_t ccw(int32_t a, int32_t b, int32_t c) const {
    const uint32_t pa = pt[a], pb = pt[b], pc = pt[c];
    const int32_t cx = (int32_t)(pc & 0xffffu), cy = (int32_t)(pc >> 16);
    return ((int32_t)(pa & 0xffffu) - cx) * ((int32_t)(pb >> 16) - cy) -
           ((int32_t)(pa >> 16) - cy) * ((int32_t)(pb & 0xffffu) - cx);
  }
  VSM_HD inline int64_t incircle(int32_t a, int32_t b, int32_t c, int32_t d) const {
    const uint32_t pa = pt[a], pb = pt[b], pc = pt[c], pd = pt[d];
    const int32_t dx = (int32_t)(pd & 0xffffu), dy = (int32_t)(pd >> 16);
    const int32_t adx = (int32_t)(pa & 0xffffu) - dx, ady = (int32_t)(pa >> 16) - dy;
    const int32_t bdx = (int32_t)(pb & 0xffffu) - dx, bdy = (int32_t)(pb >> 16) - dy;
    const int32_t cdx = (int32_t)(pc & 0xffffu) - dx, cdy = (int32_t)(pc >> 16) - dy;
    return (int64_t)(adx * adx + ady * ady) * (bdx * cdy - cdx * bdy) +
           (int64_t)(bdx * bdx + bdy * bdy) * (cdx * ady - adx * cdy) +
           (int64_t)(cdx * cdx + cdy * cdy) * (adx * bdy - bdx * ady);
  }

  // merge of two triangulated halves (mergehulls, :5639); tcur = first free slot of the pair 2b-2, 2b-1
  VSM_HD inline void merge_hulls(OTri &farleft, OTri &innerleft, OTri &innerright, OTri &farright, int axis,
                                 int32_t &tcur) const;
  // one sub-problem: positions [off, off+n), already in kd order; leaves order their keys by x and
  // fill pt / id; returns the hull handles
  VSM_HD inline void recurse(int32_t off, int32_t n, int axis, OTri &farleft, OTri &farright) const;
};

VSM_HD inline void DcMesh::merge_hulls(OTri &farleft, OTri &innerleft, OTri &innerright, OTri &farright, int axis,
                                       int32_t &tcur) const {
  int32_t ildest = dest(innerleft), ilapex = apex(innerleft);
  int32_t irorg = org(innerright), irapex = apex(innerright);
  if (axis == 1) {  // horizontal cut: handles move to the bottom-/top-most hull vertices (:5666)
    int32_t flpt = org(farleft), flapex = apex(farleft);
    int32_t frpt = dest(farright);
    while (py(flapex) < py(flpt)) {
      farleft = sym(lnext(farleft));
      flpt = flapex;
      flapex = apex(farleft);
    }
    OTri chk = sym(innerleft);
    int32_t cv = apex(chk);
    while (py(cv) > py(ildest)) {
      innerleft = lnext(chk);
      ilapex = ildest;
      ildest = cv;
      chk = sym(innerleft);
      cv = apex(chk);
    }
    while (py(irapex) < py(irorg)) {
      innerright = sym(lnext(innerright));
      irorg = irapex;
      irapex = apex(innerright);
    }
    chk = sym(farright);
    cv = apex(chk);
    while (py(cv) > py(frpt)) {
      farright = lnext(chk);
      frpt = cv;
      chk = sym(farright);
      cv = apex(chk);
    }
  }
  bool changed;
  do {  // lower common tangent (:5704)
    changed = false;
    if (ccw(ildest, ilapex, irorg) > 0) {
      innerleft = sym(lprev(innerleft));
      ildest = ilapex;
      ilapex = apex(innerleft);
      changed = true;
    }
    if (ccw(irapex, irorg, ildest) > 0) {
      innerright = sym(lnext(innerright));
      irorg = irapex;
      irapex = apex(innerright);
      changed = true;
    }
  } while (changed);
  OTri leftcand = sym(innerleft), rightcand = sym(innerright);
  OTri base = make(tcur);
  bond(base, innerleft);
  base = lnext(base);
  bond(base, innerright);
  base = lnext(base);
  set_org(base, irorg);
  set_dest(base, ildest);
  if (ildest == org(farleft)) farleft = lnext(base);
  if (irorg == dest(farright)) farright = lprev(base);
  int32_t ll = ildest, lr = irorg;
  int32_t ul = apex(leftcand), ur = apex(rightcand);
  for (;;) {
    const bool lfin = ccw(ul, ll, lr) <= 0, rfin = ccw(ur, ll, lr) <= 0;
    if (lfin && rfin) {  // close the seam with the top bounding triangle (:5771)
      OTri top = make(tcur);
      set_org(top, ll);
      set_dest(top, lr);
      bond(top, base);
      top = lnext(top);
      bond(top, rightcand);
      top = lnext(top);
      bond(top, leftcand);
      if (axis == 1) {  // handles back to the left-/right-most vertices
        int32_t flpt = org(farleft), frpt = dest(farright), frapex = apex(farright);
        OTri chk = sym(farleft);
        int32_t cv = apex(chk);
        while (px(cv) < px(flpt)) {
          farleft = lprev(chk);
          flpt = cv;
          chk = sym(farleft);
          cv = apex(chk);
        }
        while (px(frapex) > px(frpt)) {
          farright = sym(lprev(farright));
          frpt = frapex;
          frapex = apex(farright);
        }
      }
      return;
    }
    if (!lfin) {  // dissolve non-Delaunay edges on the left (:5814)
      OTri ne = sym(lprev(leftcand));
      int32_t na = apex(ne);
      if (na >= 0) {
        bool bad = incircle(ll, lr, ul, na) > 0;
        while (bad) {
          ne = lnext(ne);
          OTri topc = sym(ne);
          ne = lnext(ne);
          OTri sidec = sym(ne);
          bond(ne, topc);
          bond(leftcand, sidec);
          leftcand = lnext(leftcand);
          OTri outerc = sym(leftcand);
          ne = lprev(ne);
          bond(ne, outerc);
          set_org(leftcand, ll);
          set_dest(leftcand, -1);
          set_apex(leftcand, na);
          set_org(ne, -1);
          set_dest(ne, ul);
          set_apex(ne, na);
          ul = na;
          ne = sidec;
          na = apex(ne);
          bad = na >= 0 && incircle(ll, lr, ul, na) > 0;
        }
      }
    }
    if (!rfin) {  // ... and on the right (:5862)
      OTri ne = sym(lnext(rightcand));
      int32_t na = apex(ne);
      if (na >= 0) {
        bool bad = incircle(ll, lr, ur, na) > 0;
        while (bad) {
          ne = lprev(ne);
          OTri topc = sym(ne);
          ne = lprev(ne);
          OTri sidec = sym(ne);
          bond(ne, topc);
          bond(rightcand, sidec);
          rightcand = lprev(rightcand);
          OTri outerc = sym(rightcand);
          ne = lnext(ne);
          bond(ne, outerc);
          set_org(rightcand, -1);
          set_dest(rightcand, lr);
          set_apex(rightcand, na);
          set_org(ne, ur);
          set_dest(ne, -1);
          set_apex(ne, na);
          ur = na;
          ne = sidec;
          na = apex(ne);
          bad = na >= 0 && incircle(ll, lr, ur, na) > 0;
        }
      }
    }
    if (lfin || (!rfin && incircle(ul, ll, lr, ur) > 0)) {  // new cross edge ll--ur (:5911)
      bond(base, rightcand);
      base = lprev(rightcand);
      set_dest(base, ll);
      lr = ur;
      rightcand = sym(base);
      ur = apex(rightcand);
    } else {  // new cross edge ul--lr (:5920)
      bond(base, leftcand);
      base = lnext(leftcand);
      set_org(base, lr);
      ll = ul;
      leftcand = sym(base);
      ul = apex(leftcand);
    }
  }
}


// one sub-problem: positions [off, off+n).  Selection (alternateaxes) and triangulation
// (divconqrecurse) share one recursion; the caller has already brought the right keys into
// this slice.  Triangle slots: leaves use 2*off.., the merge at boundary b uses 2b-2, 2b-1.
VSM_HD inline void DcMesh::recurse(int32_t off, int32_t n, int axis, OTri &farleft, OTri &farright) const {
  uint64_t *a = key + off;
  if (n <= 3) {  // leaf: always ordered by x (then y), :5596-5600
    if (VSM_KXY(a[0]) > VSM_KXY(a[1])) swap_keys(a[0], a[1]);
    if (n == 3) {
      if (VSM_KXY(a[1]) > VSM_KXY(a[2])) swap_keys(a[1], a[2]);
      if (VSM_KXY(a[0]) > VSM_KXY(a[1])) swap_keys(a[0], a[1]);
    }
    for (int32_t i = 0; i < n; i++) {
      pt[off + i] = (uint32_t)(a[i] >> 34) | ((uint32_t)((a[i] >> 20) & 0x3fff) << 16);
      id[off + i] = (int32_t)(a[i] & 0xfffff);
    }
    int32_t tcur = 2 * off;
    const int32_t p0 = off, p1 = off + 1, p2 = off + 2;
    if (n == 2) {  // one edge = two ghost triangles (:5978)
      farleft = make(tcur);
      set_org(farleft, p0);
      set_dest(farleft, p1);
      farright = make(tcur);
      set_org(farright, p1);
      set_dest(farright, p0);
      bond(farleft, farright);
      farleft = lprev(farleft);
      farright = lnext(farright);
      bond(farleft, farright);
      farleft = lprev(farleft);
      farright = lnext(farright);
      bond(farleft, farright);
      farleft = lprev(farright);
      return;
    }
    OTri mid = make(tcur), t1 = make(tcur), t2 = make(tcur), t3 = make(tcur);  // (:6006)
    const int32_t area = ccw(p0, p1, p2);
    if (area == 0) {
      set_org(mid, p0);
      set_dest(mid, p1);
      set_org(t1, p1);
      set_dest(t1, p0);
      set_org(t2, p2);
      set_dest(t2, p1);
      set_org(t3, p1);
      set_dest(t3, p2);
      bond(mid, t1);
      bond(t2, t3);
      mid = lnext(mid);
      t1 = lprev(t1);
      t2 = lnext(t2);
      t3 = lprev(t3);
      bond(mid, t3);
      bond(t1, t2);
      mid = lnext(mid);
      t1 = lprev(t1);
      t2 = lnext(t2);
      t3 = lprev(t3);
      bond(mid, t1);
      bond(t2, t3);
      farleft = t1;
      farright = t2;
    } else {
      const int32_t b = area > 0 ? p1 : p2, c = area > 0 ? p2 : p1;
      set_org(mid, p0);
      set_dest(t1, p0);
      set_org(t3, p0);
      set_dest(mid, b);
      set_org(t1, b);
      set_dest(t2, b);
      set_apex(mid, c);
      set_org(t2, c);
      set_dest(t3, c);
      bond(mid, t1);
      mid = lnext(mid);
      bond(mid, t2);
      mid = lnext(mid);
      bond(mid, t3);
      t1 = lprev(t1);
      t2 = lnext(t2);
      bond(t1, t2);
      t1 = lprev(t1);
      t3 = lprev(t3);
      bond(t1, t3);
      t2 = lnext(t2);
      t3 = lprev(t3);
      bond(t2, t3);
      farleft = t1;
      farright = area > 0 ? t2 : lnext(farleft);
    }
    return;
  }
  const int32_t divider = n >> 1;  // kd_order() has already arranged both halves
  OTri innerleft, innerright;
  recurse(off, divider, 1 - axis, farleft, innerleft);
  recurse(off + divider, n - divider, 1 - axis, innerright, farright);
  int32_t tcur = 2 * (off + divider) - 2;
  merge_hulls(farleft, innerleft, innerright, farright, axis, tcur);
}

