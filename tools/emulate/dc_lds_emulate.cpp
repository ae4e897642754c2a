// Host emulation of k_dc2_block (vsm_dc.hip): the 64 lanes of a wave one after the other, phase by phase (the lanes of a
// phase own disjoint sub-trees, so their order does not matter), on the very accessors the kernel uses (vsm_dc_lds.h) -
// against ExactDelaunay's triangulation of the same points.  Build: see tools/emulate/Makefile.
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <vector>

#include "vsm_dc_lds.h"
#include "vsm_host.h"

static uint32_t rng = 12345;
static uint32_t rnd() {
  rng ^= rng << 13;
  rng ^= rng >> 17;
  rng ^= rng << 5;
  return rng;
}

int main() {
  int bad = 0;
  for (int n : {2, 3, 4, 5, 7, 14, 15, 16, 29, 100, 333, 470, 480}) {
    for (int rep = 0; rep < 60; rep++) {
      std::vector<int32_t> x(n), y(n);
      for (int i = 0; i < n; i++) {
        x[i] = (int)(rnd() % (rep % 3 == 0 ? 20 : 600)) * 2;
        y[i] = (int)(rnd() % (rep % 3 == 0 ? 20 : 180)) * 2;
      }
      ExactDelaunay ref;
      if (!ref.prepare(x.data(), y.data(), n, 1 << 30, nullptr, 0, false, true)) continue;  // radix-sorted keys, kd order, one task
      const int32_t m = ref.points();
      std::vector<uint64_t> keys(ref.mesh().key, ref.mesh().key + m);
      ref.solve_tasks();
      ref.finish();
      // ---- the block kernel's work on a copy of the kd-ordered keys ----
      std::vector<uint32_t> words((size_t)2 * m * 4, 0xffffffffu);
      std::vector<uint64_t> key = keys;
      std::vector<uint32_t> pt(m);
      std::vector<int32_t> id(m);
      Dc2Hull16 hull[2 << DC2_BLOCK_DEPTH];
      DcBlockMesh mesh;
      mesh.w = words.data();
      mesh.pt = pt.data();
      mesh.key = key.data();
      mesh.ptw = pt.data();
      mesh.gid = id.data();
      for (int lane = 0; lane < (1 << DC2_BLOCK_DEPTH); lane++) dc2_block_leaf_run(mesh, lane, m, 0, hull);
      for (int L = DC2_BLOCK_DEPTH - 1; L >= 0; L--)
        for (int lane = 0; lane < (1 << DC2_BLOCK_DEPTH); lane++) dc2_block_merge_run(mesh, lane, L, m, 0, hull);
      // compare records (local numbering with off = 0 is the global numbering)
      const DcMesh rm = ref.mesh();
      int diffs = 0;
      for (int32_t t = 0; t < 2 * m; t++) {
        const int32_t *rv = rm.tri + (size_t)t * 8 + 4;
        const bool unused = (rv[0] & rv[1] & rv[2]) < 0;  // (the two spare slots of a triangulation: only their vertices are defined)
        for (int o = 0; o < 3; o++) {
          const uint32_t word = words[(size_t)t * 4 + o];
          const int32_t nb = (word & 0xffffu) == 0xffffu ? -1 : (int32_t)(word & 0xffffu), vx = (word >> 16) == 0xffffu ? -1 : (int32_t)(word >> 16);
          if (!unused && nb != rm.tri[(size_t)t * 8 + o]) diffs++;
          if (vx != rm.tri[(size_t)t * 8 + 4 + o]) diffs++;
        }
      }
      for (int32_t i = 0; i < m; i++) diffs += pt[i] != rm.pt[i] || id[i] != rm.id[i];
      if (diffs) {
        bad++;
        printf("n %d rep %d (m %d): %d differences\n", n, rep, m, diffs);
      }
    }
  }
  printf(bad ? "FAILED: %d cases differ\n" : "dc_lds_emulate: all cases equal\n", bad);
  return bad != 0;
}
