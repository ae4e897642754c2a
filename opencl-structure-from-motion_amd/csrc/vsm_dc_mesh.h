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

struct DcOTri {  // oriented triangle handle: slot + edge
  int32_t t, o;
};

// mergehulls (viso/triangle.cpp:5639-5960) over any mesh view M that offers DcMesh's accessors (sym, org, dest,
// apex, set_*, bond, make, ccw, incircle, px, py): DcMesh itself (plain arrays, host and device) and the
// LDS-cached view of the GPU's upper merge levels (DcCachedMesh, vsm_dc.hip) share this one body.
template <class M>
VSM_HD inline void dc_merge_hulls(const M &m, DcOTri &farleft, DcOTri &innerleft, DcOTri &innerright, DcOTri &farright,
                                  int axis, int32_t &tcur);

// The seam walk a mesh view wants: views over the GPU's edge words (kFastZip, vsm_dc_lds.h) bring dc2_zip, a form of the
// same walk written for them; everything else takes dc_merge_hulls.  Same decisions, same slots, same records either way.
template <class M>
VSM_HD inline void dc2_zip(const M &m, DcOTri &farleft, DcOTri &innerleft, DcOTri &innerright, DcOTri &farright, int axis, int32_t &tcur);
template <class M>
VSM_HD inline void dc_merge(const M &m, DcOTri &farleft, DcOTri &innerleft, DcOTri &innerright, DcOTri &farright, int axis, int32_t &tcur) {
  if constexpr (M::kFastZip)
    dc2_zip(m, farleft, innerleft, innerright, farright, axis, tcur);
  else
    dc_merge_hulls(m, farleft, innerleft, innerright, farright, axis, tcur);
}

// divconqrecurse + alternateaxes' leaf rule (viso/triangle.cpp:5963, :5596) over a mesh view M that additionally offers the
// leaves' access to keys and points: key_at(i) (reference to the packed key at position i), put_point(i, x | y << 16, id)
template <class M>
VSM_HD inline void dc_recurse(const M &m, int32_t off, int32_t n, int axis, DcOTri &farleft, DcOTri &farright);

struct DcMesh {
  typedef DcOTri OTri;
  // a guarded view (the band cache of the GPU's merge levels, vsm_dc.hip) can refuse an access: dc_merge_hulls then
  // leaves at the next loop head and the caller redoes the node another way; plain views never refuse
  static constexpr bool kGuarded = false;
  static constexpr bool kFastZip = false;
  VSM_HD inline bool ok() const { return true; }
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
  VSM_HD inline uint64_t &key_at(int32_t i) const { return key[i]; }
  VSM_HD inline void put_point(int32_t i, uint32_t p, int32_t idv) const {
    pt[i] = p;
    id[i] = idv;
  }
  VSM_HD inline int32_t px(int32_t p) const { return (int32_t)(pt[p] & 0xffffu); }
  VSM_HD inline int32_t py(int32_t p) const { return (int32_t)(pt[p] >> 16); }
  // the seam loop of merge_hulls keeps the packed coordinates of its four corner points beside their indices
  VSM_HD inline uint32_t P(int32_t p) const { return pt[p]; }
  VSM_HD static inline int32_t ccw_p(uint32_t pa, uint32_t pb, uint32_t pc) {
    const int32_t cx = (int32_t)(pc & 0xffffu), cy = (int32_t)(pc >> 16);
    return ((int32_t)(pa & 0xffffu) - cx) * ((int32_t)(pb >> 16) - cy) -
           ((int32_t)(pa >> 16) - cy) * ((int32_t)(pb & 0xffffu) - cx);
  }
  VSM_HD static inline int64_t incircle_p(uint32_t pa, uint32_t pb, uint32_t pc, uint32_t pd) {
    const int32_t dx = (int32_t)(pd & 0xffffu), dy = (int32_t)(pd >> 16);
    const int32_t adx = (int32_t)(pa & 0xffffu) - dx, ady = (int32_t)(pa >> 16) - dy;
    const int32_t bdx = (int32_t)(pb & 0xffffu) - dx, bdy = (int32_t)(pb >> 16) - dy;
    const int32_t cdx = (int32_t)(pc & 0xffffu) - dx, cdy = (int32_t)(pc >> 16) - dy;
    return (int64_t)(adx * adx + ady * ady) * (bdx * cdy - cdx * bdy) +
           (int64_t)(bdx * bdx + bdy * bdy) * (cdx * ady - adx * cdy) +
           (int64_t)(cdx * cdx + cdy * cdy) * (adx * bdy - bdx * ady);
  }
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

template <class M>
VSM_HD inline void dc_merge_hulls(const M &m, DcOTri &farleft, DcOTri &innerleft, DcOTri &innerright, DcOTri &farright,
                                  int axis, int32_t &tcur) {
  typedef DcOTri OTri;
  int32_t ildest = m.dest(innerleft), ilapex = m.apex(innerleft);
  int32_t irorg = m.org(innerright), irapex = m.apex(innerright);
  if (axis == 1) {  // horizontal cut: handles move to the bottom-/top-most hull vertices (:5666)
    int32_t flpt = m.org(farleft), flapex = m.apex(farleft);
    int32_t frpt = m.dest(farright);
    while (m.py(flapex) < m.py(flpt)) {
      if (M::kGuarded && !m.ok()) return;
      farleft = m.sym(M::lnext(farleft));
      flpt = flapex;
      flapex = m.apex(farleft);
    }
    OTri chk = m.sym(innerleft);
    int32_t cv = m.apex(chk);
    while (m.py(cv) > m.py(ildest)) {
      if (M::kGuarded && !m.ok()) return;
      innerleft = M::lnext(chk);
      ilapex = ildest;
      ildest = cv;
      chk = m.sym(innerleft);
      cv = m.apex(chk);
    }
    while (m.py(irapex) < m.py(irorg)) {
      if (M::kGuarded && !m.ok()) return;
      innerright = m.sym(M::lnext(innerright));
      irorg = irapex;
      irapex = m.apex(innerright);
    }
    chk = m.sym(farright);
    cv = m.apex(chk);
    while (m.py(cv) > m.py(frpt)) {
      if (M::kGuarded && !m.ok()) return;
      farright = M::lnext(chk);
      frpt = cv;
      chk = m.sym(farright);
      cv = m.apex(chk);
    }
  }
  bool changed;
  do {  // lower common tangent (:5704)
    if (M::kGuarded && !m.ok()) return;
    changed = false;
    if (m.ccw(ildest, ilapex, irorg) > 0) {
      innerleft = m.sym(M::lprev(innerleft));
      ildest = ilapex;
      ilapex = m.apex(innerleft);
      changed = true;
    }
    if (m.ccw(irapex, irorg, ildest) > 0) {
      innerright = m.sym(M::lnext(innerright));
      irorg = irapex;
      irapex = m.apex(innerright);
      changed = true;
    }
  } while (changed);
  OTri leftcand = m.sym(innerleft), rightcand = m.sym(innerright);
  OTri base = m.make(tcur);
  m.bond(base, innerleft);
  base = M::lnext(base);
  m.bond(base, innerright);
  base = M::lnext(base);
  m.set_org(base, irorg);
  m.set_dest(base, ildest);
  if (ildest == m.org(farleft)) farleft = M::lnext(base);
  if (irorg == m.dest(farright)) farright = M::lprev(base);
  int32_t ll = ildest, lr = irorg;
  int32_t ul = m.apex(leftcand), ur = m.apex(rightcand);
  // (packed coordinates of the four corners travel with their indices: every point is read once, when it becomes a corner)
  uint32_t pll = m.P(ll), plr = m.P(lr), pul = m.P(ul), pur = m.P(ur);
  for (;;) {
    if (M::kGuarded && !m.ok()) return;
    const bool lfin = M::ccw_p(pul, pll, plr) <= 0, rfin = M::ccw_p(pur, pll, plr) <= 0;

    if (lfin && rfin) {  // close the seam with the top bounding triangle (:5771)
      OTri top = m.make(tcur);
      m.set_org(top, ll);
      m.set_dest(top, lr);
      m.bond(top, base);
      top = M::lnext(top);
      m.bond(top, rightcand);
      top = M::lnext(top);
      m.bond(top, leftcand);
      if (axis == 1) {  // handles back to the left-/right-most vertices
        int32_t flpt = m.org(farleft), frpt = m.dest(farright), frapex = m.apex(farright);
        OTri chk = m.sym(farleft);
        int32_t cv = m.apex(chk);
        while (m.px(cv) < m.px(flpt)) {
          if (M::kGuarded && !m.ok()) return;
          farleft = M::lprev(chk);
          flpt = cv;
          chk = m.sym(farleft);
          cv = m.apex(chk);
        }
        while (m.px(frapex) > m.px(frpt)) {
          if (M::kGuarded && !m.ok()) return;
          farright = m.sym(M::lprev(farright));
          frpt = frapex;
          frapex = m.apex(farright);
        }
      }
      return;
    }
    if (!lfin) {  // dissolve non-Delaunay edges on the left (:5814)
      OTri ne = m.sym(M::lprev(leftcand));
      int32_t na = m.apex(ne);
      if (na >= 0) {
        uint32_t pna = m.P(na);
        bool bad = M::incircle_p(pll, plr, pul, pna) > 0;
        while (bad) {
          if (M::kGuarded && !m.ok()) return;
          ne = M::lnext(ne);
          OTri topc = m.sym(ne);
          ne = M::lnext(ne);
          OTri sidec = m.sym(ne);
          m.bond(ne, topc);
          m.bond(leftcand, sidec);
          leftcand = M::lnext(leftcand);
          OTri outerc = m.sym(leftcand);
          ne = M::lprev(ne);
          m.bond(ne, outerc);
          m.set_org(leftcand, ll);
          m.set_dest(leftcand, -1);
          m.set_apex(leftcand, na);
          m.set_org(ne, -1);
          m.set_dest(ne, ul);
          m.set_apex(ne, na);
          ul = na;
          pul = pna;
          ne = sidec;
          na = m.apex(ne);
          bad = false;
          if (na >= 0) {
            pna = m.P(na);
            bad = M::incircle_p(pll, plr, pul, pna) > 0;
          }
        }
      }
    }
    if (!rfin) {  // ... and on the right (:5862)
      OTri ne = m.sym(M::lnext(rightcand));
      int32_t na = m.apex(ne);
      if (na >= 0) {
        uint32_t pna = m.P(na);
        bool bad = M::incircle_p(pll, plr, pur, pna) > 0;
        while (bad) {
          if (M::kGuarded && !m.ok()) return;
          ne = M::lprev(ne);
          OTri topc = m.sym(ne);
          ne = M::lprev(ne);
          OTri sidec = m.sym(ne);
          m.bond(ne, topc);
          m.bond(rightcand, sidec);
          rightcand = M::lprev(rightcand);
          OTri outerc = m.sym(rightcand);
          ne = M::lnext(ne);
          m.bond(ne, outerc);
          m.set_org(rightcand, -1);
          m.set_dest(rightcand, lr);
          m.set_apex(rightcand, na);
          m.set_org(ne, ur);
          m.set_dest(ne, -1);
          m.set_apex(ne, na);
          ur = na;
          pur = pna;
          ne = sidec;
          na = m.apex(ne);
          bad = false;
          if (na >= 0) {
            pna = m.P(na);
            bad = M::incircle_p(pll, plr, pur, pna) > 0;
          }
        }
      }
    }
    if (lfin || (!rfin && M::incircle_p(pul, pll, plr, pur) > 0)) {  // new cross edge ll--ur (:5911)
      m.bond(base, rightcand);
      base = M::lprev(rightcand);
      m.set_dest(base, ll);
      lr = ur;
      plr = pur;
      rightcand = m.sym(base);
      ur = m.apex(rightcand);
      pur = m.P(ur);
    } else {  // new cross edge ul--lr (:5920)
      m.bond(base, leftcand);
      base = M::lnext(leftcand);
      m.set_org(base, lr);
      ll = ul;
      pll = pul;
      leftcand = m.sym(base);
      ul = m.apex(leftcand);
      pul = m.P(ul);
    }
  }
}


VSM_HD inline void DcMesh::merge_hulls(OTri &farleft, OTri &innerleft, OTri &innerright, OTri &farright, int axis,
                                       int32_t &tcur) const {
  dc_merge_hulls(*this, farleft, innerleft, innerright, farright, axis, tcur);
}

// one sub-problem: positions [off, off+n).  Selection (alternateaxes) and triangulation
// (divconqrecurse) share one recursion; the caller has already brought the right keys into
// this slice.  Triangle slots: leaves use 2*off.., the merge at boundary b uses 2b-2, 2b-1.
// a leaf of two or three points: ordered by x (then y), :5596-5600, then the edge / triangle cases of divconqrecurse
template <class M>
VSM_HD inline void dc_leaf(const M &m, int32_t off, int32_t n, DcOTri &farleft, DcOTri &farright) {
  typedef DcOTri OTri;
  {
    uint64_t k0 = m.key_at(off), k1 = m.key_at(off + 1), k2 = n == 3 ? m.key_at(off + 2) : 0;
    if (VSM_KXY(k0) > VSM_KXY(k1)) DcMesh::swap_keys(k0, k1);
    if (n == 3) {
      if (VSM_KXY(k1) > VSM_KXY(k2)) DcMesh::swap_keys(k1, k2);
      if (VSM_KXY(k0) > VSM_KXY(k1)) DcMesh::swap_keys(k0, k1);
    }
    m.key_at(off) = k0;
    m.key_at(off + 1) = k1;
    if (n == 3) m.key_at(off + 2) = k2;
    const uint64_t a[3] = {k0, k1, k2};
    for (int32_t i = 0; i < n; i++)
      m.put_point(off + i, (uint32_t)(a[i] >> 34) | ((uint32_t)((a[i] >> 20) & 0x3fff) << 16), (int32_t)(a[i] & 0xfffff));
    int32_t tcur = 2 * off;
    const int32_t p0 = off, p1 = off + 1, p2 = off + 2;
    if (n == 2) {  // one edge = two ghost triangles (:5978)
      farleft = m.make(tcur);
      m.set_org(farleft, p0);
      m.set_dest(farleft, p1);
      farright = m.make(tcur);
      m.set_org(farright, p1);
      m.set_dest(farright, p0);
      m.bond(farleft, farright);
      farleft = M::lprev(farleft);
      farright = M::lnext(farright);
      m.bond(farleft, farright);
      farleft = M::lprev(farleft);
      farright = M::lnext(farright);
      m.bond(farleft, farright);
      farleft = M::lprev(farright);
      return;
    }
    OTri mid = m.make(tcur), t1 = m.make(tcur), t2 = m.make(tcur), t3 = m.make(tcur);  // (:6006)
    const int32_t area = m.ccw(p0, p1, p2);
    if (area == 0) {
      m.set_org(mid, p0);
      m.set_dest(mid, p1);
      m.set_org(t1, p1);
      m.set_dest(t1, p0);
      m.set_org(t2, p2);
      m.set_dest(t2, p1);
      m.set_org(t3, p1);
      m.set_dest(t3, p2);
      m.bond(mid, t1);
      m.bond(t2, t3);
      mid = M::lnext(mid);
      t1 = M::lprev(t1);
      t2 = M::lnext(t2);
      t3 = M::lprev(t3);
      m.bond(mid, t3);
      m.bond(t1, t2);
      mid = M::lnext(mid);
      t1 = M::lprev(t1);
      t2 = M::lnext(t2);
      t3 = M::lprev(t3);
      m.bond(mid, t1);
      m.bond(t2, t3);
      farleft = t1;
      farright = t2;
    } else {
      const int32_t b = area > 0 ? p1 : p2, c = area > 0 ? p2 : p1;
      m.set_org(mid, p0);
      m.set_dest(t1, p0);
      m.set_org(t3, p0);
      m.set_dest(mid, b);
      m.set_org(t1, b);
      m.set_dest(t2, b);
      m.set_apex(mid, c);
      m.set_org(t2, c);
      m.set_dest(t3, c);
      m.bond(mid, t1);
      mid = M::lnext(mid);
      m.bond(mid, t2);
      mid = M::lnext(mid);
      m.bond(mid, t3);
      t1 = M::lprev(t1);
      t2 = M::lnext(t2);
      m.bond(t1, t2);
      t1 = M::lprev(t1);
      t3 = M::lprev(t3);
      m.bond(t1, t3);
      t2 = M::lnext(t2);
      t3 = M::lprev(t3);
      m.bond(t2, t3);
      farleft = t1;
      farright = area > 0 ? t2 : M::lnext(farleft);
    }
    return;
  }
}

template <class M>
VSM_HD inline void dc_recurse(const M &m, int32_t off, int32_t n, int axis, DcOTri &farleft, DcOTri &farright) {
  typedef DcOTri OTri;
  if (n <= 3) {
    dc_leaf(m, off, n, farleft, farright);
    return;
  }
  const int32_t divider = n >> 1;  // kd_order() has already arranged both halves
  OTri innerleft, innerright;
  dc_recurse(m, off, divider, 1 - axis, farleft, innerleft);
  dc_recurse(m, off + divider, n - divider, 1 - axis, innerright, farright);
  int32_t tcur = 2 * (off + divider) - 2;
  dc_merge(m, farleft, innerleft, innerright, farright, axis, tcur);
}

VSM_HD inline void DcMesh::recurse(int32_t off, int32_t n, int axis, OTri &farleft, OTri &farright) const {
  dc_recurse(*this, off, n, axis, farleft, farright);
}

// The same without recursion, for GPU lanes (a recursive device function needs a dynamic stack per lane): post-order walk
// of the halving tree with an explicit stack of at most MAXDEPTH pending nodes (n <= 3 * 2^(MAXDEPTH-1) points).
template <int MAXDEPTH, class M>
VSM_HD inline void dc_build_iter(const M &m, int32_t off0, int32_t n0, int axis0, DcOTri &farleft, DcOTri &farright) {
  typedef DcOTri OTri;
  // pending nodes: position, size, axis | phase << 8, and (once the left child is done) its two hull handles
  int32_t s_off[MAXDEPTH], s_n[MAXDEPTH], s_ap[MAXDEPTH], s_flt[MAXDEPTH], s_flo[MAXDEPTH], s_ilt[MAXDEPTH], s_ilo[MAXDEPTH];
  int sp = 0;
  s_off[0] = off0;
  s_n[0] = n0;
  s_ap[0] = axis0;
  OTri rl{0, 0}, rr{0, 0};  // hull handles of the node finished last
  for (;;) {
    const int32_t off = s_off[sp], n = s_n[sp], axis = s_ap[sp] & 0xff, phase = s_ap[sp] >> 8;
    if (n <= 3) {
      dc_leaf(m, off, n, rl, rr);
    } else if (phase == 0) {
      s_ap[sp] = axis | (1 << 8);
      s_off[sp + 1] = off;
      s_n[sp + 1] = n >> 1;
      s_ap[sp + 1] = 1 - axis;
      sp++;
      continue;
    } else if (phase == 1) {
      s_flt[sp] = rl.t;
      s_flo[sp] = rl.o;
      s_ilt[sp] = rr.t;
      s_ilo[sp] = rr.o;
      s_ap[sp] = axis | (2 << 8);
      const int32_t div = n >> 1;
      s_off[sp + 1] = off + div;
      s_n[sp + 1] = n - div;
      s_ap[sp + 1] = 1 - axis;
      sp++;
      continue;
    } else {
      OTri fl{s_flt[sp], s_flo[sp]}, il{s_ilt[sp], s_ilo[sp]}, ir = rl, fr = rr;
      int32_t tcur = 2 * (off + (n >> 1)) - 2;
      dc_merge(m, fl, il, ir, fr, axis, tcur);
      rl = fl;
      rr = fr;
    }
    if (sp == 0) break;
    sp--;
  }
  farleft = rl;
  farright = rr;
}

// And without any indexed local storage (on the GPU a lane's indexed private array means scratch memory or indirect
// register addressing): for sub-trees of at most 24 points, i.e. at most three cut levels above the leaves.  A pending
// node is named by its depth and path from the root - position, size and axis are recomputed from those -, its phase
// sits in two bits per level, and the left child's hull handles wait in one packed word per level (s0..s3).
template <class M>
VSM_HD inline void dc_build_small(const M &m, int32_t off0, int32_t n0, int axis0, DcOTri &farleft, DcOTri &farright) {
  typedef DcOTri OTri;
  int d = 0;
  uint32_t path = 0, phase = 0;
  uint64_t s0 = 0, s1 = 0, s2 = 0, s3 = 0;
  OTri rl{0, 0}, rr{0, 0};  // hull handles of the node finished last
  for (;;) {
    int32_t off = off0, n = n0, axis = axis0;
    for (int b = d - 1; b >= 0; b--) {
      const int32_t div = n >> 1;
      if ((path >> b) & 1) {
        off += div;
        n -= div;
      } else {
        n = div;
      }
      axis = 1 - axis;
    }
    const uint32_t ph = (phase >> (2 * d)) & 3u;
    if (n <= 3) {
      dc_leaf(m, off, n, rl, rr);
    } else if (ph == 0) {  // left child first
      phase = (phase & ~(3u << (2 * d))) | (1u << (2 * d));
      d++;
      path <<= 1;
      phase &= ~(3u << (2 * d));
      continue;
    } else if (ph == 1) {  // the left child is done: keep its handles, then the right child
      const uint64_t pk = ((uint64_t)(uint32_t)rl.t << 34) | ((uint64_t)(uint32_t)rl.o << 32) | ((uint64_t)(uint32_t)rr.t << 2) | (uint32_t)rr.o;
      if (d == 0)
        s0 = pk;
      else if (d == 1)
        s1 = pk;
      else if (d == 2)
        s2 = pk;
      else
        s3 = pk;
      phase = (phase & ~(3u << (2 * d))) | (2u << (2 * d));
      d++;
      path = (path << 1) | 1u;
      phase &= ~(3u << (2 * d));
      continue;
    } else {
      const uint64_t pk = d == 0 ? s0 : (d == 1 ? s1 : (d == 2 ? s2 : s3));
      OTri fl{(int32_t)(pk >> 34), (int32_t)((pk >> 32) & 3u)}, il{(int32_t)((pk >> 2) & 0x3fffffffu), (int32_t)(pk & 3u)}, ir = rl, fr = rr;
      int32_t tcur = 2 * (off + (n >> 1)) - 2;
      dc_merge(m, fl, il, ir, fr, axis, tcur);
      rl = fl;
      rr = fr;
    }
    if (d == 0) break;
    d--;
    path >>= 1;
  }
  farleft = rl;
  farright = rr;
}
