// Interface between the host side of the exact Delaunay (ExactDelaunay, vsm_host.h) and the GPU
// solver of its sub-trees (vsm_dc.hip).
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

struct VsmDcTask {  // == ExactDelaunay::Task
  int32_t off, n, axis, node;
};
struct VsmDcHull {  // the two hull handles a sub-tree hands to the merge above it
  int32_t fl_t, fl_o, fr_t, fr_o;
};
struct VsmDcJob {  // one triangulation; all pointers are device pointers
  uint64_t *key;   // [m]  packed keys in kd order (leaves reorder their 2-3 keys by x)
  uint32_t *pt;    // [m]  out: x | y << 16 by sorted position
  int32_t *id;     // [m]  out: input index by sorted position
  int32_t *tri;    // [2m][8] out: triangle records
  const VsmDcTask *tasks;
  VsmDcHull *hulls;
  int32_t ntasks, m;
};

// one thread per sub-tree, blockIdx.y = job; max_tasks >= every job's ntasks
void vsm_dc_launch_subtrees(hipStream_t s, const VsmDcJob *d_jobs, int njobs, int max_tasks);
