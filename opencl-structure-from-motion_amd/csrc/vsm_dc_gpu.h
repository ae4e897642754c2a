// Interface between the host side of the exact Delaunay (ExactDelaunay, vsm_host.h) and the GPU
// solver of its sub-trees (vsm_dc.hip).
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "visomatch.h"

struct VsmDcTask {  // == ExactDelaunay::Task
  int32_t off, n, axis, node;
};
struct VsmDcMerge {  // == ExactDelaunay::Merge: an internal node, children by node number
  int32_t off, n, axis, node, left, right;
};
struct VsmDcHull {  // the two hull handles a node hands to the merge above it
  int32_t fl_t, fl_o, fr_t, fr_o;
};
#define VSM_DC_MAX_LEVELS 8
#define VSM_DC_KD_MAX_POINTS 65536  // sizes k_dc_kd_order was written for (32-bit counters in 128 KB of LDS)
#define VSM_DC_KD_SCRATCH 7        // uint32 arrays of m entries it needs per job
struct VsmDcJob {  // one triangulation; all pointers are device pointers
  const uint64_t *key_sorted;  // [m] distinct packed keys in (x,y) order, or null: `key` arrives in kd order
  uint32_t *kd_scratch;        // [VSM_DC_KD_SCRATCH][kd_stride]
  int32_t kd_stride, pad_;
  uint64_t *key;   // [m]  packed keys in kd order (leaves reorder their 2-3 keys by x)
  uint32_t *pt;    // [m]  out: x | y << 16 by sorted position
  int32_t *id;     // [m]  out: input index by sorted position
  int32_t *tri;    // [2m][8] out: triangle records
  uint32_t *tri_packed;  // or (k_dc_block): [2m][3] words, neighbour handle (17 bits, all ones = none) | vertex << 17 (15 bits,
                         // all ones = none) per corner - what travels to the host instead of the 32-byte records
#define VSM_DC_PACKED_MAX_POINTS 16000
  const VsmDcTask *tasks;
  const VsmDcMerge *merges;  // merge nodes of the levels above the tasks, deepest level first
  VsmDcHull *hulls;          // by node number
  int32_t ntasks, m;
  int32_t nlevels, level_off[VSM_DC_MAX_LEVELS + 1];  // merges[level_off[l] .. level_off[l+1]) is level l
  // support test of removeOutliers on the finished triangulation (k_dc_support): per input match
  const float *flow_u, *flow_v, *disp;  // [n_in] u1c-u1p, v1c-v1p, disparity (vsm_host_outliers_begin)
  int32_t *support;                     // [n_in] out
  int32_t n_in, pad2_;
  // Triangle's randomised vertex sort, emulated on the device (k_dc_ties): which match stands for a pixel that
  // several matches share
  const uint64_t *tie_keys;  // [n_in] packed keys in input order, or null
  int32_t *tie_out;          // [1 + 2 * VSM_DC_TIE_PATCHES]: count (-1: not done here), then (index carried, index it should be)
};
#define VSM_DC_TIE_POINTS 10240  // lists up to this length are sorted inside LDS (12 bytes per point + masks + stack: 141 KB)
#define VSM_DC_TIE_PATCHES 255

// kd order of the jobs that bring key_sorted: one workgroup per job (ExactDelaunay::kd_order on the device)
void vsm_dc_launch_kd_order(hipStream_t s, const VsmDcJob *d_jobs, int njobs);
// one thread per sub-tree, blockIdx.y = job; max_tasks >= every job's ntasks
void vsm_dc_launch_subtrees(hipStream_t s, const VsmDcJob *d_jobs, int njobs, int max_tasks);
// one wave per sub-tree of at most VSM_DC_BLOCK_POINTS points, triangulated inside LDS (leaves of <= 14 points
// one per lane, then the merge levels); blockIdx.y = job; max_tasks >= every job's ntasks
#define VSM_DC_BLOCK_POINTS 480
void vsm_dc_launch_blocks(hipStream_t s, const VsmDcJob *d_jobs, int njobs, int max_tasks);
// removeOutliers' support count (viso/matcher.cpp:1266-1364) over the triangle slots of finished triangulations:
// thread per slot, blockIdx.y = job; max_points >= every job's m.  method: 0 flow, 1 stereo, 2 quad
void vsm_dc_launch_support(hipStream_t s, const VsmDcJob *d_jobs, int njobs, int max_points, int method, float flow_tol,
                           float disp_tol);
// one wave per job: the emulated vertex sort of the jobs that bring tie_keys
void vsm_dc_launch_ties(hipStream_t s, const VsmDcJob *d_jobs, int njobs);
// ... for the pairs of a chunk: the keys of their compacted pass-2 lists into keys[pair * stride ..] (lengths into
// counts), then one wave per pair on that copy; tie_out (device-visible) + pair * out_stride gets the verdict
struct VsmPair;
void vsm_dc_launch_tie_keys(hipStream_t s, const VsmPair *d_pairs, int npairs, int max_list, uint64_t *keys, int stride, int32_t *counts);
void vsm_dc_launch_ties_of_keys(hipStream_t s, int npairs, const uint64_t *keys, int stride, const int32_t *counts, int32_t *tie_out,
                                int out_stride);
#define VSM_DC_TIE_OUT_INTS 512
// one thread per merge node of level `level`; max_nodes >= every job's node count on that level
void vsm_dc_launch_merge_level(hipStream_t s, const VsmDcJob *d_jobs, int njobs, int level, int max_nodes);

// ---------------------------------------------------------------------------------------
// GPU-resident form of removeOutliers (viso/matcher.cpp:1207-1377) for the look-ahead path: the match list never
// leaves the device between the matching pass and the survivors.  One job per frame pair and pass; every kernel
// derives what it needs (list length, distinct points, divide-and-conquer tree) from device memory, so a whole
// chunk's chain is enqueued without a host round trip.  What may still run on the host is Triangle's randomised
// vertex sort (which match stands for a shared pixel): it is strictly serial, the host gets the keys through
// host-mapped memory and answers with (carried index, right index) patches.
// ---------------------------------------------------------------------------------------
#define VSM_DC2_MAX_DEPTH 9  // block sub-trees of <= VSM_DC_BLOCK_POINTS points below at most this many cut levels
struct VsmDc2Job {
  const vsm_p_match *list;  // the compacted match list of the pass (device) ...
  const int32_t *count;     // ... and its length (device)
  int32_t cap, kd_stride;   // list lengths beyond `cap` are not handled here (error flag); stride of the scratch arrays
  uint64_t *keys_in;        // [cap] packed keys (x << 34 | y << 20 | index) in list order (k_dc2_keys)
  uint64_t *h_keys;         // the same in host-mapped memory for the host's vertex sort, or null
  int32_t *h_n;             // host-mapped: list length (with h_keys)
  uint64_t *key_sorted;     // [cap] distinct keys in (x,y) order
  uint64_t *key;            // [cap] ... in kd order
  uint32_t *kd_scratch;     // [VSM_DC_KD_SCRATCH][kd_stride]; later: flow u, flow v, disparity per match (float)
  uint32_t *pt;             // [cap] x | y << 16 by position
  int32_t *id;              // [cap] by position: smallest input index at that pixel
  int32_t *tri;             // [2 cap][8] triangle records
  VsmDcHull *hulls;         // [2 << VSM_DC2_MAX_DEPTH] by heap index of the tree node (root = 1)
  int32_t *mn;              // [2] distinct points m, list length n (device; written by k_dc2_prepare)
  int32_t *support;         // [cap] support count per match
  int32_t *remap;           // [cap] match index a position's id stands for (identity + tie patches); later: output position
  const int32_t *tie_out;   // verdict of the vertex sort: count (-1: failed), then (carried index, right index) pairs
  vsm_p_match *out;         // survivors (device or host-mapped memory), or null (pass 1: only the statistics follow)
  int32_t *out_count;       // device
  int32_t *h_out_count;     // host-mapped, or null
  float *ranges;            // pass 1: the pair's prior boxes [ub*vb][16], device layout (k_dc2_prior)
  int32_t *error;           // host-mapped flag word: set when a pair cannot be handled on the device
  int32_t *long_seen;       // host-mapped, or null: set when a list beyond the LDS forms went through the narrow form inside k_dc2_prepare_lds
  uint32_t *h_keep;         // host-mapped, or null: [ceil(cap / 32)] survivor bits - the list itself is on the host already (DMA copy behind the
                            // refinement, while the triangulation runs), k_dc2_compact only says which of its matches stay
};
void vsm_dc2_launch_keys(hipStream_t s, const VsmDc2Job *d_jobs, int njobs, int max_list);
void vsm_dc2_launch_prepare(hipStream_t s, const VsmDc2Job *d_jobs, int njobs, int max_list = 0, bool expect_long = true);  // max_list: upper bound of the list lengths (0 = unknown); expect_long: lists beyond the LDS forms' 8192 points are likely (their own kernel comes along)
void vsm_dc2_launch_blocks(hipStream_t s, const VsmDc2Job *d_jobs, int njobs, int depth);
void vsm_dc2_launch_merges(hipStream_t s, const VsmDc2Job *d_jobs, int njobs, int depth, int max_list);  // levels depth-1 .. 0
void vsm_dc2_launch_ties(hipStream_t s, const VsmDc2Job *d_jobs, int njobs, int32_t *tie_out, int out_stride, int max_list = 0);  // max_list: upper bound of the list lengths (0 = unknown)
void vsm_dc2_launch_flows(hipStream_t s, const VsmDc2Job *d_jobs, int njobs, int max_list, int method);  // per match: flow, disparity (needs the refined list)
void vsm_dc2_launch_votes(hipStream_t s, const VsmDc2Job *d_jobs, int njobs, int max_list, int method, float flow_tol, float disp_tol);  // tie patches, votes, survivors
void vsm_dc2_launch_prior(hipStream_t s, const VsmDc2Job *d_jobs, int njobs, int method, int binsize, int radius, int w, int h,
                          int ub, int vb);
// smallest depth at which every sub-tree of a list of at most max_points points fits a block
static inline int vsm_dc2_depth(int max_points) {
  int d = 0;
  while (((long)VSM_DC_BLOCK_POINTS << d) < max_points) d++;
  return d;
}
