// The lower levels of the exact divide-and-conquer Delaunay on the GPU.
// The host prepares every triangulation (emulated vertex sort -- it decides which duplicate point
// survives --, kd order, tree layout, ExactDelaunay::prepare) and keeps the few large merges at the
// top of the tree; the many small independent sub-trees below -- where most of mergehulls' work is,
// the seams of level k add up to ~sqrt(n 2^k) -- are triangulated here, one thread per sub-tree, by
// the very same code (DcMesh::recurse, vsm_dc_mesh.h): integer predicates, identical decisions,
// triangle slots fixed by position, so the host can continue on the arrays as if it had done the
// work itself.
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
  jb.hulls[t] = VsmDcHull{fl.t, fl.o, fr.t, fr.o};
}

void vsm_dc_launch_subtrees(hipStream_t s, const VsmDcJob *d_jobs, int njobs, int max_tasks) {
  if (njobs <= 0 || max_tasks <= 0) return;
  hipLaunchKernelGGL(k_dc_subtrees, dim3((max_tasks + 63) / 64, njobs), dim3(64), 0, s, d_jobs, njobs);
}
