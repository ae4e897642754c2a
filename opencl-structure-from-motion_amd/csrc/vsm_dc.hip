// The lower levels of the exact divide-and-conquer Delaunay on the GPU.
// The host prepares every triangulation (emulated vertex sort -- it decides which duplicate point
// survives --, kd order, tree layout, ExactDelaunay::prepare) and keeps the few large merges at the
// top of the tree; the many small independent sub-trees below -- where most of mergehulls' work is,
// the seams of level k add up to ~sqrt(n 2^k) -- are triangulated here, one thread per sub-tree, by
// the very same code (DcMesh::recurse, vsm_dc_mesh.h): integer predicates, identical decisions,
// triangle slots fixed by position, so the host can continue on the arrays as if it had done the
// work itself.  The merge levels directly above the sub-trees follow level by level (k_dc_merge_level,
// one thread per merge node): their seams are still short, and there are still thousands of them per
// chunk of frame pairs.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#include "vsm_dc_gpu.h"
#include "vsm_dc_mesh.h"

__global__ void __launch_bounds__(64) k_dc_subtrees(const VsmDcJob *__restrict__ jobs, int njobs) {
  const int j = blockIdx.y;
  if (j >= njobs) return;
  const VsmDcJob jb = jobs[j];
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= jb.ntasks) return;
  const DcMesh mesh{jb.tri, jb.pt, jb.id, jb.key};
  const VsmDcTask tk = jb.tasks[t];
  DcMesh::OTri fl, fr;
  mesh.recurse(tk.off, tk.n, tk.axis, fl, fr);
  jb.hulls[tk.node] = VsmDcHull{fl.t, fl.o, fr.t, fr.o};
}

__global__ void __launch_bounds__(64) k_dc_merge_level(const VsmDcJob *__restrict__ jobs, int njobs, int level) {
  const int j = blockIdx.y;
  if (j >= njobs) return;
  const VsmDcJob jb = jobs[j];
  if (level >= jb.nlevels) return;
  const int t = jb.level_off[level] + blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= jb.level_off[level + 1]) return;
  const DcMesh mesh{jb.tri, jb.pt, jb.id, jb.key};
  const VsmDcMerge mg = jb.merges[t];
  const VsmDcHull l = jb.hulls[mg.left], r = jb.hulls[mg.right];
  DcMesh::OTri fl{l.fl_t, l.fl_o}, il{l.fr_t, l.fr_o}, ir{r.fl_t, r.fl_o}, fr{r.fr_t, r.fr_o};
  int32_t tcur = 2 * (mg.off + (mg.n >> 1)) - 2;
  mesh.merge_hulls(fl, il, ir, fr, mg.axis, tcur);
  jb.hulls[mg.node] = VsmDcHull{fl.t, fl.o, fr.t, fr.o};
}

void vsm_dc_launch_subtrees(hipStream_t s, const VsmDcJob *d_jobs, int njobs, int max_tasks) {
  if (njobs <= 0 || max_tasks <= 0) return;
  hipLaunchKernelGGL(k_dc_subtrees, dim3((max_tasks + 63) / 64, njobs), dim3(64), 0, s, d_jobs, njobs);
}

void vsm_dc_launch_merge_level(hipStream_t s, const VsmDcJob *d_jobs, int njobs, int level, int max_nodes) {
  if (njobs <= 0 || max_nodes <= 0) return;
  hipLaunchKernelGGL(k_dc_merge_level, dim3((max_nodes + 63) / 64, njobs), dim3(64), 0, s, d_jobs, njobs, level);
}
