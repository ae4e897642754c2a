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

// ---------------------------------------------------------------------------------------
// kd order (ExactDelaunay::kd_order, vsm_host.cpp; Triangle's alternateaxes, viso/triangle.cpp:5583):
// the points, named by their rank in (x,y) order, are kept once in x order (X) and once in y order (Y);
// a node [off, off+n) of depth d is cut along axis d & 1 in the middle of that axis' list, the other
// list is stable-partitioned; leaves (<= 3 points) stay in x order.  The host walks the tree node by
// node; here one workgroup takes a whole triangulation level by level: all nodes of a depth share the
// axis, so a stable partition of every node at once is ONE exclusive scan of the "goes left" flags over
// all positions (new position = node start [+ half] + flags before it inside the node).
// ---------------------------------------------------------------------------------------
#define KD_THREADS 1024
#define KD_CHUNKS 256
#define KD_DIGITS 128

// exclusive prefix over the block of one value per thread; tot[17] is LDS scratch
__device__ inline uint32_t kd_block_scan(uint32_t v, uint32_t *tot) {
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  uint32_t x = v;
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    const uint32_t y = __shfl_up(x, d, 64);
    if (lane >= d) x += y;
  }
  if (lane == 63) tot[wv] = x;
  __syncthreads();
  if (wv == 0) {
    const uint32_t t = lane < KD_THREADS / 64 ? tot[lane] : 0;
    uint32_t s = t;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
      const uint32_t y = __shfl_up(s, d, 64);
      if (lane >= d) s += y;
    }
    if (lane < KD_THREADS / 64) tot[lane] = s - t;
  }
  __syncthreads();
  const uint32_t r = x - v + tot[wv];
  __syncthreads();
  return r;
}

__device__ inline void kd_node_at(int32_t q, int depth, int32_t m, int32_t &off, int32_t &n) {
  off = 0;
  n = m;
  for (int i = 0; i < depth; i++) {
    const int32_t div = n >> 1;
    if (q < off + div) {
      n = div;
    } else {
      off += div;
      n -= div;
    }
  }
}

__global__ void __launch_bounds__(KD_THREADS) k_dc_kd_order(const VsmDcJob *__restrict__ jobs, int njobs) {
  __shared__ uint32_t hist[KD_DIGITS * KD_CHUNKS];
  __shared__ uint32_t tot[KD_THREADS / 64 + 1];
  const VsmDcJob jb = jobs[blockIdx.x];
  const int32_t m = jb.m;
  if (!jb.key_sorted || m < 2 || m > VSM_DC_KD_MAX_POINTS) return;  // (uniform for the block)
  const uint64_t *ks = jb.key_sorted;
  uint32_t *X0 = jb.kd_scratch, *X1 = X0 + jb.kd_stride, *Y0 = X1 + jb.kd_stride, *Y1 = Y0 + jb.kd_stride;
  uint32_t *PX = Y1 + jb.kd_stride, *PY = PX + jb.kd_stride, *P = PY + jb.kd_stride;
  const int t = threadIdx.x;

  // ---- y order: stable LSD radix sort of the ranks by y (7 + 7 bits); every thread of the first
  // KD_CHUNKS owns a contiguous chunk and its own column of the histogram, so no atomics and the order
  // inside a digit is the order of the source ----
  const int32_t chunk = (m + KD_CHUNKS - 1) / KD_CHUNKS;
  for (int pass = 0; pass < 2; pass++) {
    const uint32_t *src = X1;            // pass 1 reads what pass 0 wrote (X1 is free until the levels start)
    uint32_t *dst = pass == 0 ? X1 : Y0;
    const int sh = 20 + 7 * pass;
    for (int i = t; i < KD_DIGITS * KD_CHUNKS; i += KD_THREADS) hist[i] = 0;
    __syncthreads();
    if (t < KD_CHUNKS) {
      const int32_t i0 = t * chunk, i1 = min(m, i0 + chunk);
      for (int32_t i = i0; i < i1; i++) {
        const uint32_t e = pass == 0 ? (uint32_t)i : src[i];
        hist[((ks[e] >> sh) & (KD_DIGITS - 1)) * KD_CHUNKS + t]++;
      }
    }
    __syncthreads();
    {  // exclusive scan over the histogram in (digit, chunk) order
      const int per = KD_DIGITS * KD_CHUNKS / KD_THREADS;
      uint32_t sum = 0;
      for (int i = 0; i < per; i++) sum += hist[t * per + i];
      uint32_t run = kd_block_scan(sum, tot);
      for (int i = 0; i < per; i++) {
        const uint32_t c = hist[t * per + i];
        hist[t * per + i] = run;
        run += c;
      }
    }
    __syncthreads();
    if (t < KD_CHUNKS) {
      const int32_t i0 = t * chunk, i1 = min(m, i0 + chunk);
      for (int32_t i = i0; i < i1; i++) {
        const uint32_t e = pass == 0 ? (uint32_t)i : src[i];
        const uint32_t d = (ks[e] >> sh) & (KD_DIGITS - 1);
        dst[hist[d * KD_CHUNKS + t]++] = e;
      }
    }
    __syncthreads();
  }
  for (int32_t q = t; q < m; q += KD_THREADS) {
    X0[q] = (uint32_t)q;
    PX[q] = (uint32_t)q;
    PY[Y0[q]] = (uint32_t)q;
  }
  __syncthreads();

  // ---- the levels ----
  uint32_t *X = X0, *Xn = X1, *Y = Y0, *Yn = Y1;
  const int32_t per = (m + KD_THREADS - 1) / KD_THREADS;
  for (int depth = 0; ((m + (1 << depth) - 1) >> depth) > 3; depth++) {
    const bool cut_x = (depth & 1) == 0;
    const uint32_t *S = cut_x ? Y : X;     // the list to partition
    uint32_t *D = cut_x ? Yn : Xn;
    const uint32_t *PO = cut_x ? PX : PY;  // position in the list that is cut in place
    uint32_t *PS = cut_x ? PY : PX;
    // flags + exclusive scan (thread = contiguous run of positions)
    {
      const int32_t q0 = t * per, q1 = min(m, q0 + per);
      uint32_t sum = 0;
      for (int32_t q = q0; q < q1; q++) {
        int32_t off, n;
        kd_node_at(q, depth, m, off, n);
        const uint32_t left = n <= 3 || (int32_t)PO[S[q]] < off + (n >> 1);
        P[q] = left;
        sum += left;
      }
      uint32_t run = kd_block_scan(sum, tot);
      for (int32_t q = q0; q < q1; q++) {
        const uint32_t f = P[q];
        P[q] = (run << 1) | f;  // flags before q, and q's own
        run += f;
      }
    }
    __syncthreads();
    for (int32_t q = t; q < m; q += KD_THREADS) {
      int32_t off, n;
      kd_node_at(q, depth, m, off, n);
      const uint32_t pq = P[q], before = (pq >> 1) - (P[off] >> 1), e = S[q];
      const uint32_t np = (pq & 1) ? off + before : off + (n >> 1) + ((q - off) - before);
      D[np] = e;
      PS[e] = np;
    }
    __syncthreads();
    if (cut_x) {
      uint32_t *w = Y;
      Y = Yn;
      Yn = w;
    } else {
      uint32_t *w = X;
      X = Xn;
      Xn = w;
    }
  }
  for (int32_t q = t; q < m; q += KD_THREADS) jb.key[q] = ks[X[q]];
}

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

// ---------------------------------------------------------------------------------------
// One wave per sub-tree, inside LDS.  k_dc_subtrees / k_dc_merge_level chase pointers through global
// memory, one dependent L2 round trip per step; a sub-tree of <= 480 points is 30 KB of triangle
// records, so here it is built where a step costs an LDS access: the wave cuts its slice further down
// (same halving rule as the host's tree) into <= 64 leaves of <= 14 points, one per lane, then merges
// level by level (32, 16, ... 1 lanes), and writes records, points and ids back in one coalesced sweep.
// DcMesh addresses triangles and points by their global slot / position: the LDS arrays are handed to
// it rebased by the slice offset.
// ---------------------------------------------------------------------------------------
#define DCB_LEAF 14
#define DCB_DEPTH 6  // ceil(480 / 2^6) <= 14

__global__ void __launch_bounds__(64) k_dc_block(const VsmDcJob *__restrict__ jobs, int njobs) {
  __shared__ int32_t s_tri[2 * VSM_DC_BLOCK_POINTS * 8];
  __shared__ uint64_t s_key[VSM_DC_BLOCK_POINTS];
  __shared__ uint32_t s_pt[VSM_DC_BLOCK_POINTS];
  __shared__ int32_t s_id[VSM_DC_BLOCK_POINTS];
  __shared__ VsmDcHull s_hull[2 << DCB_DEPTH];
  const int j = blockIdx.y;
  if (j >= njobs) return;
  const VsmDcJob jb = jobs[j];
  if ((int)blockIdx.x >= jb.ntasks) return;
  const VsmDcTask tk = jb.tasks[blockIdx.x];
  const int lane = threadIdx.x;
  if (tk.n > VSM_DC_BLOCK_POINTS) {  // not expected (the host cuts tasks to fit): plain recursion in global memory
    int32_t *gt = jb.tri + (size_t)2 * tk.off * 8;
    for (int i = lane; i < 2 * tk.n * 8; i += 64) gt[i] = -1;
    __syncthreads();
    if (lane == 0) {
      const DcMesh mesh{jb.tri, jb.pt, jb.id, jb.key};
      DcMesh::OTri fl, fr;
      mesh.recurse(tk.off, tk.n, tk.axis, fl, fr);
      jb.hulls[tk.node] = VsmDcHull{fl.t, fl.o, fr.t, fr.o};
    }
    return;
  }
  for (int i = lane; i < tk.n; i += 64) s_key[i] = jb.key[tk.off + i];
  for (int i = lane; i < 2 * tk.n * 8; i += 64) s_tri[i] = -1;
  __syncthreads();
  // (rebased through integers: the arithmetic must happen on the 64-bit flat address, an LDS pointer moved
  // below its window in 32 bits and converted afterwards would leave the aperture when indexed)
  const uintptr_t ot = (uintptr_t)tk.off;
  const DcMesh mesh{(int32_t *)((uintptr_t)(int32_t *)s_tri - ot * 64), (uint32_t *)((uintptr_t)(uint32_t *)s_pt - ot * 4),
                    (int32_t *)((uintptr_t)(int32_t *)s_id - ot * 4), (uint64_t *)((uintptr_t)(uint64_t *)s_key - ot * 8)};
  // leaves: lane bits choose the path from the task's root, most significant first; a leaf reached early is
  // taken by the lane whose remaining bits are zero
  {
    int32_t off = tk.off, n = tk.n, axis = tk.axis, idx = 1;
    bool mine = true;
    for (int b = DCB_DEPTH - 1; b >= 0; b--) {
      if (n <= DCB_LEAF) {
        mine = (lane & ((2 << b) - 1)) == 0;
        break;
      }
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
    if (mine) {
      DcMesh::OTri fl, fr;
      mesh.recurse(off, n, axis, fl, fr);
      s_hull[idx] = VsmDcHull{fl.t, fl.o, fr.t, fr.o};
    }
  }
  __syncthreads();
  for (int L = DCB_DEPTH - 1; L >= 0; L--) {
    if (lane < (1 << L)) {
      int32_t off = tk.off, n = tk.n, axis = tk.axis, idx = 1;
      bool exists = true;
      for (int b = L - 1; b >= 0; b--) {
        if (n <= DCB_LEAF) {
          exists = false;
          break;
        }
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
      if (exists && n > DCB_LEAF) {
        const VsmDcHull l = s_hull[2 * idx], r = s_hull[2 * idx + 1];
        DcMesh::OTri fl{l.fl_t, l.fl_o}, il{l.fr_t, l.fr_o}, ir{r.fl_t, r.fl_o}, fr{r.fr_t, r.fr_o};
        int32_t tcur = 2 * (off + (n >> 1)) - 2;
        mesh.merge_hulls(fl, il, ir, fr, axis, tcur);
        s_hull[idx] = VsmDcHull{fl.t, fl.o, fr.t, fr.o};
      }
    }
    __syncthreads();
  }
  int32_t *gt = jb.tri + (size_t)2 * tk.off * 8;
  for (int i = lane; i < 2 * tk.n * 8; i += 64) gt[i] = s_tri[i];
  for (int i = lane; i < tk.n; i += 64) {
    jb.pt[tk.off + i] = s_pt[i];
    jb.id[tk.off + i] = s_id[i];
  }
  if (lane == 0) jb.hulls[tk.node] = s_hull[1];
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

void vsm_dc_launch_kd_order(hipStream_t s, const VsmDcJob *d_jobs, int njobs) {
  if (njobs <= 0) return;
  hipLaunchKernelGGL(k_dc_kd_order, dim3(njobs), dim3(KD_THREADS), 0, s, d_jobs, njobs);
}

void vsm_dc_launch_blocks(hipStream_t s, const VsmDcJob *d_jobs, int njobs, int max_tasks) {
  if (njobs <= 0 || max_tasks <= 0) return;
  hipLaunchKernelGGL(k_dc_block, dim3(max_tasks, njobs), dim3(64), 0, s, d_jobs, njobs);
}

// The support test of removeOutliers (viso/matcher.cpp:1266-1364; vsm_host_outliers_end is the host form):
// every triangle gives each of its three edges a vote for both end points if the two matches agree in flow
// and / or disparity.  Differences, absolute values, one sum and a compare in float: the same values on
// either side; the counts are integer sums, so their order does not matter.
__global__ void __launch_bounds__(256) k_dc_support(const VsmDcJob *__restrict__ jobs, int njobs, int method, float ftol, float dtol) {
  const int j = blockIdx.y;
  if (j >= njobs) return;
  const VsmDcJob jb = jobs[j];
  if (!jb.support || jb.ntasks <= 0) return;
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= 2 * jb.m) return;
  const int32_t *v = jb.tri + (size_t)t * 8 + 4;
  const int32_t v0 = v[0], v1 = v[1], v2 = v[2];
  if ((v0 | v1 | v2) < 0) return;
  const int32_t q[3] = {jb.id[v1], jb.id[v2], jb.id[v0]};
  float fu[3], fv[3], dp[3];
#pragma unroll
  for (int k = 0; k < 3; k++) {
    fu[k] = jb.flow_u[q[k]];
    fv[k] = jb.flow_v[q[k]];
    dp[k] = jb.disp[q[k]];
  }
#pragma unroll
  for (int e = 0; e < 3; e++) {
    const int a = e == 2 ? 0 : e, b = e == 0 ? 1 : 2;  // (0,1) (1,2) (0,2)
    const bool flow_ok = fabsf(fu[a] - fu[b]) + fabsf(fv[a] - fv[b]) < ftol;
    const bool disp_ok = fabsf(dp[a] - dp[b]) < dtol;
    const bool ok = method == 0 ? flow_ok : (method == 1 ? disp_ok : (disp_ok && flow_ok));
    if (ok) {
      atomicAdd(&jb.support[q[a]], 1);
      atomicAdd(&jb.support[q[b]], 1);
    }
  }
}

void vsm_dc_launch_support(hipStream_t s, const VsmDcJob *d_jobs, int njobs, int max_points, int method, float flow_tol,
                           float disp_tol) {
  if (njobs <= 0 || max_points <= 0) return;
  hipLaunchKernelGGL(k_dc_support, dim3((2 * max_points + 255) / 256, njobs), dim3(256), 0, s, d_jobs, njobs, method, flow_tol, disp_tol);
}
