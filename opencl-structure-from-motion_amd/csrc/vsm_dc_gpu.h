// Interface between the host side of the exact Delaunay (ExactDelaunay, vsm_host.h) and the GPU
// solver of its sub-trees (vsm_dc.hip).
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

struct VsmDcTask {  // == ExactDelaunay::Task
  int32_t off, n, axis, node;
};
struct VsmDcMerge {  // == ExactDelaunay::Merge: an internal node, children by node number
  int32_t off, n, axis, node, left, right;
};
struct VsmDcHull {  // the two hull handles a node hands to the merge above it
  int32_t fl_t, fl_o, fr_t, fr_o;
};
#define VSM_DC_MAX_LEVELS 6
struct VsmDcJob {  // one triangulation; all pointers are device pointers
  uint64_t *key;   // [m]  packed keys in kd order (leaves reorder their 2-3 keys by x)
  uint32_t *pt;    // [m]  out: x | y << 16 by sorted position
  int32_t *id;     // [m]  out: input index by sorted position
  int32_t *tri;    // [2m][8] out: triangle records
  const VsmDcTask *tasks;
  const VsmDcMerge *merges;  // merge nodes of the levels above the tasks, deepest level first
  VsmDcHull *hulls;          // by node number
  int32_t ntasks, m;
  int32_t nlevels, level_off[VSM_DC_MAX_LEVELS + 1];  // merges[level_off[l] .. level_off[l+1]) is level l
};

// one thread per sub-tree, blockIdx.y = job; max_tasks >= every job's ntasks
void vsm_dc_launch_subtrees(hipStream_t s, const VsmDcJob *d_jobs, int njobs, int max_tasks);
// one thread per merge node of level `level`; max_nodes >= every job's node count on that level
void vsm_dc_launch_merge_level(hipStream_t s, const VsmDcJob *d_jobs, int njobs, int level, int max_nodes);
