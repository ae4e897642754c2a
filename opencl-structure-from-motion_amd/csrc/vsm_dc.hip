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
#include <atomic>
#include <stdint.h>
#include <stdio.h>

#include <algorithm>
#include <mutex>
#include <vector>
#include <cmath>

#include "vsm_dc_gpu.h"
#include "vsm_dc_mesh.h"
#include "vsm_dc_lds.h"
#include "vsm_internal.h"

static const uint32_t *dc2_tiny_table();  // the tiny-partition table of the device's vertex sort on the current device (below)

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

// the levels of the kd order on the lists X (x order), Y (y order) and their inverses PX, PY (element -> position);
// T = uint16_t for lists that live in LDS (m <= KD_LDS_POINTS), uint32_t for the global scratch
template <typename T>
__device__ inline void kd_levels(T *X0, T *X1, T *Y0, T *Y1, T *PX, T *PY, uint32_t *P, const int32_t m, uint32_t *tot,
                                 const uint64_t *__restrict__ ks, uint64_t *__restrict__ key_out) {
  const int t = threadIdx.x;
  T *X = X0, *Xn = X1, *Y = Y0, *Yn = Y1;
  const int32_t per = (m + KD_THREADS - 1) / KD_THREADS;
  for (int depth = 0; ((m + (1 << depth) - 1) >> depth) > 3; depth++) {
    const bool cut_x = (depth & 1) == 0;
    const T *S = cut_x ? Y : X;     // the list to partition
    T *D = cut_x ? Yn : Xn;
    const T *PO = cut_x ? PX : PY;  // position in the list that is cut in place
    T *PS = cut_x ? PY : PX;
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
      D[np] = (T)e;
      PS[e] = (T)np;
    }
    __syncthreads();
    if (cut_x) {
      T *w = Y;
      Y = Yn;
      Yn = w;
    } else {
      T *w = X;
      X = Xn;
      Xn = w;
    }
  }
  for (int32_t q = t; q < m; q += KD_THREADS) key_out[q] = ks[X[q]];
}

// ---------------------------------------------------------------------------------------
// The same levels for lists beyond the LDS forms (config 5: 20-40 k distinct points), lists in global memory.  kd_levels above
// gives a thread a contiguous run of positions: its loads are a cache line per lane, every position's node is recomputed from
// the root at every level, the flags go through a global array twice - 93 us per level, 1.4 ms of the 2 ms a 34 k list took.
// (What bounds it then, 31 us per level of a 34 k list: one compute unit's rate of uncoalesced lane accesses - a gather and two
// scatters per point and level, ~1.5 per clock; having eight positions' loads in flight instead of one changes nothing.)
// Here thread t owns positions t, t + 1024, ...: coalesced reads of the list, the node (offset, size) of each of its positions
// lives in a register and is halved once per level, the "goes left" flags of a level are wave ballots kept in LDS (one
// 64-bit word per wave and batch) with an exclusive prefix over the words, so the number of flags before ANY position - the
// node's start included - is one LDS word, a popcount and an add.  Per level: one gather (the cut list's position of the
// element), two scatters.
// ---------------------------------------------------------------------------------------
__device__ inline void dc2_wave_sync_early() {
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
  __builtin_amdgcn_wave_barrier();
}
#define KD_WIDE_E 64  // batches of KD_THREADS positions: m <= 65535
__device__ inline void kd_levels_wide(uint32_t *X0, uint32_t *X1, uint32_t *Y0, uint32_t *Y1, uint32_t *PX, uint32_t *PY, uint32_t *NODE,
                                      const int32_t m, uint32_t *lds /* 3 * KD_WIDE_E * 16 words */, uint32_t *tot,
                                      const uint64_t *__restrict__ ks, uint64_t *__restrict__ key_out) {
  const int t = threadIdx.x, lane = t & 63, wv = t >> 6;
  constexpr int NW = KD_THREADS / 64;
  uint64_t *s_ball = (uint64_t *)lds;           // [batch][wave]: word index = position >> 6
  uint32_t *s_base = lds + 2 * KD_WIDE_E * NW;  // flags before the word
  const int E = (m + KD_THREADS - 1) / KD_THREADS;
  // NODE[q] = offset | size << 16 of the node position q belongs to at the current depth (coalesced, a register's worth per
  // position and level; 64 of them in registers spilled)
  for (int32_t q = t; q < m; q += KD_THREADS) NODE[q] = (uint32_t)m << 16;
  uint32_t *X = X0, *Xn = X1, *Y = Y0, *Yn = Y1;
  const uint64_t lt = lane == 0 ? 0ull : (~0ull >> (64 - lane));
  auto flags_before = [&](int32_t q) -> uint32_t {  // "goes left" flags of this level at positions < q
    const int w = q >> 6, l = q & 63;
    return s_base[w] + (uint32_t)__popcll(s_ball[w] & (l == 0 ? 0ull : (~0ull >> (64 - l))));
  };
  __syncthreads();
  for (int depth = 0; ((m + (1 << depth) - 1) >> depth) > 3; depth++) {
    const bool cut_x = (depth & 1) == 0;
    const uint32_t *S = cut_x ? Y : X;     // the list to partition
    uint32_t *D = cut_x ? Yn : Xn;
    const uint32_t *PO = cut_x ? PX : PY;  // position in the list that is cut in place
    uint32_t *PS = cut_x ? PY : PX;
    uint64_t fl = 0;
    for (int i = 0; i < E; i++) {
      const int32_t q = i * KD_THREADS + t;
      bool left = false;
      if (q < m) {
        const uint32_t nd = NODE[q];
        const int32_t off = (int32_t)(nd & 0xffffu), n = (int32_t)(nd >> 16);
        left = n <= 3 || (int32_t)PO[S[q]] < off + (n >> 1);
      }
      const uint64_t b = __ballot(left);
      if (lane == 0) s_ball[i * NW + wv] = b;
      fl |= (uint64_t)(left ? 1u : 0u) << i;
    }
    __syncthreads();
    {  // exclusive prefix over the E * 16 <= 1024 words
      const uint32_t c = t < E * NW ? (uint32_t)__popcll(s_ball[t]) : 0u;
      const uint32_t run = kd_block_scan(c, tot);
      if (t < E * NW) s_base[t] = run;
    }
    __syncthreads();
#pragma unroll 4
    for (int i = 0; i < E; i++) {
      const int32_t q = i * KD_THREADS + t;
      if (q < m) {
        const uint32_t nd = NODE[q];
        const int32_t off = (int32_t)(nd & 0xffffu), n = (int32_t)(nd >> 16);
        const uint32_t before = s_base[i * NW + wv] + (uint32_t)__popcll(s_ball[i * NW + wv] & lt) - flags_before(off);
        const uint32_t e = S[q];
        const uint32_t np = ((fl >> i) & 1ull) ? (uint32_t)off + before : (uint32_t)off + (uint32_t)(n >> 1) + ((uint32_t)(q - off) - before);
        D[np] = e;
        PS[e] = np;
        // the position's node one level down (kd_node_at's step)
        const int32_t div = n >> 1;
        NODE[q] = q < off + div ? ((uint32_t)off | ((uint32_t)div << 16)) : ((uint32_t)(off + div) | ((uint32_t)(n - div) << 16));
      }
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
  for (int32_t q = t; q < m; q += KD_THREADS) key_out[q] = ks[X[q]];
}

// One stable LSD pass (7 bits) over n items in global memory with every thread at work: wave w takes the contiguous
// range [w * per_wave, ...), 64 consecutive items per step (coalesced) and four steps' items requested at once, ranks a
// step's items with seven ballots, counts per (digit, wave) in LDS; one exclusive scan in (digit, wave) order gives every
// wave its start per digit.  load(i) -> the item, digit_of(item) -> 0..127, store(dst, item).  hist: 128 * 16 words.
template <typename T, typename Load, typename Digit, typename Store>
__device__ inline void kd_radix_pass_wide(const int32_t n, uint32_t *hist, uint32_t *tot, Load load, Digit digit_of, Store store) {
  const int t = threadIdx.x, lane = t & 63, wv = t >> 6;
  constexpr int NW = KD_THREADS / 64, B = 4;
  const int32_t per_wave = ((n + NW * 64 - 1) / (NW * 64)) * 64;
  const int32_t w0 = min(n, wv * per_wave), w1 = min(n, w0 + per_wave);
  const uint64_t lt = lane == 0 ? 0ull : (~0ull >> (64 - lane));
  hist[2 * t] = 0;
  hist[2 * t + 1] = 0;
  __syncthreads();
  auto same_digit = [&](uint32_t d, bool valid) -> uint64_t {
    uint64_t same = __ballot(valid);
#pragma unroll
    for (int bit = 0; bit < 7; bit++) {
      const bool on = valid && ((d >> bit) & 1u);
      const uint64_t bal = __ballot(on);
      same &= ((d >> bit) & 1u) ? bal : ~bal;
    }
    return same;
  };
  for (int32_t base = w0; base < w1; base += 64 * B) {
    T v[B];
#pragma unroll
    for (int k = 0; k < B; k++) v[k] = base + 64 * k + lane < w1 ? load(base + 64 * k + lane) : T(0);
#pragma unroll
    for (int k = 0; k < B; k++) {
      const bool valid = base + 64 * k + lane < w1;
      const uint32_t d = valid ? digit_of(v[k]) : 0u;
      const uint64_t same = same_digit(d, valid);
      if (valid && (same & lt) == 0) hist[d * NW + wv] += (uint32_t)__popcll(same);  // (the digit's first lane; a wave's LDS operations keep their order)
      dc2_wave_sync_early();
    }
  }
  __syncthreads();
  {
    const uint32_t v0 = hist[2 * t], v1 = hist[2 * t + 1];
    const uint32_t run = kd_block_scan(v0 + v1, tot);
    hist[2 * t] = run;
    hist[2 * t + 1] = run + v0;
  }
  __syncthreads();
  for (int32_t base = w0; base < w1; base += 64 * B) {
    T v[B];
#pragma unroll
    for (int k = 0; k < B; k++) v[k] = base + 64 * k + lane < w1 ? load(base + 64 * k + lane) : T(0);
#pragma unroll
    for (int k = 0; k < B; k++) {
      const bool valid = base + 64 * k + lane < w1;
      const uint32_t d = valid ? digit_of(v[k]) : 0u;
      const uint64_t same = same_digit(d, valid);
      const uint32_t before = (uint32_t)__popcll(same & lt);
      const uint32_t prior = valid ? hist[d * NW + wv] : 0u;
      dc2_wave_sync_early();
      if (valid && before == 0) hist[d * NW + wv] = prior + (uint32_t)__popcll(same);
      dc2_wave_sync_early();
      if (valid) store((int32_t)(prior + before), v[k]);
    }
  }
  __syncthreads();
}

#define KD_LDS_POINTS 8192  // 6 uint16 lists + the uint32 scan fit where the histogram was

// the whole kd order of one triangulation by one workgroup of KD_THREADS threads: ks = distinct keys in (x,y) order,
// scratch = VSM_DC_KD_SCRATCH arrays of `stride` uint32, hist / tot = the workgroup's LDS (KD_DIGITS * KD_CHUNKS and
// KD_THREADS / 64 + 1 words); key_out gets the keys in kd order
// WIDE 1: lists beyond KD_LDS_POINTS with every thread at work (kd_radix_pass_wide, kd_levels_wide); 2: every list that way,
// whatever its length - 12 KB of LDS instead of the 128 KB the other forms keep their histogram columns and 16-bit lists in
// (k_dc2_prepare_long); 0: the narrow forms only, which cost the kernel they are inlined in no registers
template <int WIDE>
__device__ inline void kd_order_body(const uint64_t *__restrict__ ks, const int32_t m, uint32_t *scratch, const int32_t stride,
                                     uint64_t *__restrict__ key_out, uint32_t *hist, uint32_t *tot, long long *stamp = nullptr) {
  uint32_t *X0 = scratch, *X1 = X0 + stride, *Y0 = X1 + stride, *Y1 = Y0 + stride;
  uint32_t *PX = Y1 + stride, *PY = PX + stride, *P = PY + stride;
  const int t = threadIdx.x;

  // ---- y order: stable LSD radix sort of the ranks by y (7 + 7 bits); every thread of the first
  // KD_CHUNKS owns a contiguous chunk and its own column of the histogram, so no atomics and the order
  // inside a digit is the order of the source ----
  const int32_t chunk = (m + KD_CHUNKS - 1) / KD_CHUNKS;
  if (WIDE == 2 || (WIDE && m > KD_LDS_POINTS)) {  // long lists: every thread at work, coalesced reads (kd_radix_pass_wide)
    // (an item = y << 32 | rank, so the rank's key is looked up once; the items of the first pass lie over PX + PY,
    // adjacent and free until the levels start)
    uint64_t *Bf = (uint64_t *)PX;
    kd_radix_pass_wide<uint64_t>(m, hist, tot, [&](int32_t i) { return ((ks[i] >> 20) & 0x3fffull) << 32 | (uint64_t)(uint32_t)i; },
                                 [&](uint64_t v) { return (uint32_t)(v >> 32) & (KD_DIGITS - 1); }, [&](int32_t dst, uint64_t v) { Bf[dst] = v; });
    kd_radix_pass_wide<uint64_t>(m, hist, tot, [&](int32_t i) { return Bf[i]; }, [&](uint64_t v) { return (uint32_t)(v >> 39) & (KD_DIGITS - 1); },
                                 [&](int32_t dst, uint64_t v) { Y0[dst] = (uint32_t)v; });
  } else
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
  if (stamp) *stamp = clock64();
  if (WIDE != 2 && m <= KD_LDS_POINTS) {
    // the lists move into LDS, over the histogram (no longer needed): a dependent access there costs an LDS round
    // trip instead of one through L2
    uint16_t *l16 = (uint16_t *)hist;
    uint16_t *lX0 = l16, *lX1 = lX0 + KD_LDS_POINTS, *lY0 = lX1 + KD_LDS_POINTS, *lY1 = lY0 + KD_LDS_POINTS;
    uint16_t *lPX = lY1 + KD_LDS_POINTS, *lPY = lPX + KD_LDS_POINTS;
    uint32_t *lP = hist + 3 * KD_LDS_POINTS;  // behind the six 16-bit lists
    for (int32_t q = t; q < m; q += KD_THREADS) {
      const uint32_t e = Y0[q];
      lX0[q] = (uint16_t)q;
      lPX[q] = (uint16_t)q;
      lY0[q] = (uint16_t)e;
      lPY[e] = (uint16_t)q;
    }
    __syncthreads();
    kd_levels<uint16_t>(lX0, lX1, lY0, lY1, lPX, lPY, lP, m, tot, ks, key_out);
    return;
  }
  for (int32_t q = t; q < m; q += KD_THREADS) {
    X0[q] = (uint32_t)q;
    PX[q] = (uint32_t)q;
    PY[Y0[q]] = (uint32_t)q;
  }
  __syncthreads();
  if (WIDE && m <= 65535)
    kd_levels_wide(X0, X1, Y0, Y1, PX, PY, P, m, hist, tot, ks, key_out);
  else
    kd_levels<uint32_t>(X0, X1, Y0, Y1, PX, PY, P, m, tot, ks, key_out);
}

__global__ void __launch_bounds__(KD_THREADS) k_dc_kd_order(const VsmDcJob *__restrict__ jobs, int njobs) {
  __shared__ uint32_t hist[KD_DIGITS * KD_CHUNKS];
  __shared__ uint32_t tot[KD_THREADS / 64 + 1];
  const VsmDcJob jb = jobs[blockIdx.x];
  const int32_t m = jb.m;
  if (!jb.key_sorted || m < 2 || m > VSM_DC_KD_MAX_POINTS) return;  // (uniform for the block)
  kd_order_body<0>(jb.key_sorted, m, jb.kd_scratch, jb.kd_stride, jb.key, hist, tot);
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

// one sub-tree `tk` by the calling wave (64 threads = the whole workgroup); `jb` brings key, tri / tri_packed, pt, id;
// the sub-tree's two hull handles go to *hull_out
__device__ inline void dc_block_body(const VsmDcJob &jb, const VsmDcTask tk, VsmDcHull *hull_out) {
  __shared__ int32_t s_tri[2 * VSM_DC_BLOCK_POINTS * 8];
  __shared__ uint64_t s_key[VSM_DC_BLOCK_POINTS];
  __shared__ uint32_t s_pt[VSM_DC_BLOCK_POINTS];
  __shared__ int32_t s_id[VSM_DC_BLOCK_POINTS];
  __shared__ VsmDcHull s_hull[2 << DCB_DEPTH];
  const int lane = threadIdx.x;
  if (tk.n > VSM_DC_BLOCK_POINTS) {  // not expected (the host cuts tasks to fit): plain recursion in global memory
    int32_t *gt = jb.tri + (size_t)2 * tk.off * 8;
    for (int i = lane; i < 2 * tk.n * 8; i += 64) gt[i] = -1;
    __syncthreads();
    if (lane == 0) {
      const DcMesh mesh{jb.tri, jb.pt, jb.id, jb.key};
      DcMesh::OTri fl, fr;
      mesh.recurse(tk.off, tk.n, tk.axis, fl, fr);
      *hull_out = VsmDcHull{fl.t, fl.o, fr.t, fr.o};
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
  if (jb.tri_packed) {  // 12 bytes per triangle for the trip to the host
    uint32_t *gp = jb.tri_packed + (size_t)2 * tk.off * 3;
    for (int i = lane; i < 2 * tk.n * 3; i += 64) {
      const int t = i / 3, o = i - 3 * t;
      gp[i] = ((uint32_t)s_tri[t * 8 + o] & 0x1ffffu) | (((uint32_t)s_tri[t * 8 + 4 + o] & 0x7fffu) << 17);
    }
  } else {
    int32_t *gt = jb.tri + (size_t)2 * tk.off * 8;
    for (int i = lane; i < 2 * tk.n * 8; i += 64) gt[i] = s_tri[i];
  }
  for (int i = lane; i < tk.n; i += 64) {
    jb.pt[tk.off + i] = s_pt[i];
    jb.id[tk.off + i] = s_id[i];
  }
  if (lane == 0) *hull_out = s_hull[1];
}

__global__ void __launch_bounds__(64) k_dc_block(const VsmDcJob *__restrict__ jobs, int njobs) {
  const int j = blockIdx.y;
  if (j >= njobs) return;
  const VsmDcJob jb = jobs[j];
  if ((int)blockIdx.x >= jb.ntasks) return;
  const VsmDcTask tk = jb.tasks[blockIdx.x];
  dc_block_body(jb, tk, jb.hulls + tk.node);
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

// ---------------------------------------------------------------------------------------
// Triangle's vertexsort (viso/triangle.cpp:5447; host form: ExactDelaunay::vertex_sort) on one wave.
// The triangulation does not depend on it (ExactDelaunay::prepare, defer_ties) - only which of several
// matches at one pixel stands for the point does - but it is the largest piece of host time per frame pair
// and strictly serial: ~4900 partitions, each needing the random number after the previous one's.  A wave
// does it in LDS with the same decisions: the Hoare partition from two ballot masks (left scan stops at
// keys >= pivot, right scan at keys <= pivot), sub-arrays of <= 64 elements entirely in registers (one
// element per lane, pivot by readlane, swaps by writelane) together with everything below them in the
// recursion, larger ones through stopper lists in LDS whose swaps are independent and run one per lane.
// ---------------------------------------------------------------------------------------
#define TIE_STACK 2048

__device__ inline uint32_t tie_rnd(uint32_t &seed, uint32_t choices) {  // randomnation, :4046
  seed = (seed * 1366u + 150889u) % 714025u;
  const uint32_t d = 714025u / choices + 1;
  uint32_t q = (uint32_t)((float)seed * (1.0f / (float)d));  // seed < 2^20: off by one at most
  if (q * d > seed) q--;
  if ((q + 1) * d <= seed) q++;
  return q;
}
// the same with the divisor and its reciprocal looked up (lane c of the two table registers holds them for c choices)
__device__ inline uint32_t tie_rnd_small(uint32_t &seed, int choices, uint32_t dtab, float rtab) {
  seed = (seed * 1366u + 150889u) % 714025u;
  const uint32_t d = (uint32_t)__builtin_amdgcn_readlane((int)dtab, choices);
  const float r = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(rtab), choices));
  uint32_t q = (uint32_t)((float)seed * r);
  if (q * d > seed) q--;
  if ((q + 1) * d <= seed) q++;
  return q;
}

__device__ inline uint32_t tie_readlane(uint32_t v, int lane) { return (uint32_t)__builtin_amdgcn_readlane((int)v, lane); }
__device__ inline uint32_t tie_writelane(uint32_t val, int lane, uint32_t old) {  // (no writelane builtin in this compiler)
  return (int)threadIdx.x == lane ? val : old;
}

// Hoare loop of the reference on two masks over positions 0..n-1 of one 64-bit word (n <= 64): calls swap(l, r)
// for every exchanged pair, returns the final (left, right)
template <typename Swap>
__device__ inline void tie_hoare64(uint64_t GE, uint64_t LE, int n, int &left, int &right, Swap swap) {
  left = -1;
  right = n;
  for (;;) {
    const uint64_t above = left + 1 >= 64 ? 0ull : (~0ull << (left + 1));
    const uint64_t g = GE & above;
    int l = g ? __builtin_ctzll(g) : 64;
    if (l > right) l = right;
    if (l == right) {
      left = l;
      right = l - 1;
      return;
    }
    const uint64_t below = right >= 64 ? ~0ull : ((1ull << right) - 1);
    const uint64_t e = LE & below;
    int r = e ? 63 - __builtin_clzll(e) : -1;
    if (r <= l) r = ((LE >> l) & 1) ? l : l - 1;
    left = l;
    right = r;
    if (l >= r) return;
    swap(l, r);
  }
}

// the keys of a match list as ExactDelaunay packs them (x, y = u1c, v1c truncated, vsm_host_outliers_begin)
struct TieFromKeys {
  const uint64_t *keys;
  __device__ inline uint32_t kxy(int i) const { return (uint32_t)(keys[i] >> 20); }
  __device__ inline uint32_t idx(int i) const { return (uint32_t)(keys[i] & 0xfffffu); }
};

// Most partitions of a sort have 3..7 keys (3500 of the 4900 of a 7.4 k list), and what the Hoare loop does to such a part
// depends only on the two masks: the host looks the outcome up (TinyTable, vsm_host.cpp) - so does the wave now.  The table
// (vsm_host_tiny_table: sizes 3..7, 21824 words) lies in LDS for the lists of the first pass (CAP 2048: 25 + 87 KB, one list
// per compute unit) and, sizes 3..5 only (5.4 KB), beside the 139 KB of a second-pass list; one LDS read (uniform address)
// and two ds_bpermute replace the rounds of mask arithmetic and lane-by-lane swaps.  MEASURED: 3.39 -> 3.21 ms for a 7.4 k
// list (5 %), 422 -> 421 us for the first-pass lists: the Hoare emulation is not what a partition costs - its ~1400 cycles
// are the scalar bookkeeping around it (the generator's modulo and division, readlanes with their wait states, the pending
// parts' stack, a handful of uniform branches), all of it one dependent chain.  Kept (it is exact and tested); what would
// help is the host's other trick, a whole sub-tree of <= 7 keys in one tight loop.
#define TIE_TINY_WORDS 21824
__host__ __device__ constexpr int tie_tiny_off(int n) { return ((1 << (2 * n)) - 64) / 3; }
template <int CAP, typename Src>
__device__ inline void tie_sort(const Src src, const int n0, int32_t *__restrict__ tie_out, const uint32_t *__restrict__ g_tiny = nullptr) {
  constexpr int TMAX = CAP <= 2048 ? 6 : 5;  // part sizes the LDS copy of the table covers (size 7 alone would be another 64 KB)
  constexpr int TWORDS = tie_tiny_off(TMAX + 1);
  __shared__ uint32_t s_k[CAP], s_i[CAP];
  __shared__ uint16_t s_L[CAP], s_R[CAP];
  __shared__ uint64_t s_ge[CAP / 64], s_le[CAP / 64];
  __shared__ uint32_t s_stack[TIE_STACK * 2];
  __shared__ __attribute__((aligned(16))) uint32_t s_tiny[TWORDS];
  const int lane = threadIdx.x;
  if (n0 > CAP) {
    if (lane == 0) tie_out[0] = -1;
    return;
  }
  const bool tiny = g_tiny != nullptr;
  if (tiny) {
    static_assert(TWORDS % 4 == 0, "16-byte pieces");
    for (int i = lane * 4; i < TWORDS; i += 256) *(uint4 *)&s_tiny[i] = *(const uint4 *)&g_tiny[i];
  }
  for (int i = lane; i < n0; i += 64) {
    s_k[i] = src.kxy(i);
    s_i[i] = src.idx(i);
  }
  __syncthreads();
  uint32_t seed = 1;  // triangleinit(), :4031
  const uint32_t dtab = lane > 0 ? 714025u / (uint32_t)lane + 1 : 1u;  // randomnation's divisor for `lane` choices
  const float rtab = 1.0f / (float)dtab;
  int sp = 0;
  bool overflow = false;
  auto push = [&](int off, int n) {
    if (sp >= TIE_STACK) {
      overflow = true;
      return;
    }
    if (lane == 0) {
      s_stack[2 * sp] = (uint32_t)off;
      s_stack[2 * sp + 1] = (uint32_t)n;
    }
    sp++;
  };
  // two elements: vertexsort's n == 2 case, no random number
  auto settle_pair_lds = [&](int at) {
    if (lane == 0) {
      const uint32_t a = s_k[at], b = s_k[at + 1];
      if (a > b) {
        s_k[at] = b;
        s_k[at + 1] = a;
        const uint32_t t = s_i[at];
        s_i[at] = s_i[at + 1];
        s_i[at + 1] = t;
      }
    }
  };
  if (n0 == 2) settle_pair_lds(0);
  if (n0 > 2) push(0, n0);
  __syncthreads();
  while (sp > 0 && !overflow) {
    sp--;
    const int off = (int)__builtin_amdgcn_readfirstlane((int)s_stack[2 * sp]);
    const int n = (int)__builtin_amdgcn_readfirstlane((int)s_stack[2 * sp + 1]);
    if (n <= 64) {
      // ---- this part and all parts below it: in registers, lane = position - off; the pending parts too
      // (lane j of `stk` holds entry j: lo | m << 8) ----
      uint32_t k = lane < n ? s_k[off + lane] : 0, ix = lane < n ? s_i[off + lane] : 0, stk = 0;
      int lsp = 0;
      int lo = 0, m = n;  // current part: lanes [lo, lo + m)
      for (;;) {
        const int pl = lo + (int)(m < 64 ? tie_rnd_small(seed, m, dtab, rtab) : tie_rnd(seed, 64u));
        const uint32_t pv = tie_readlane(k, pl);
        const bool in = lane >= lo && lane < lo + m;
        const uint64_t GE = __ballot(in && k >= pv) >> lo, LE = __ballot(in && k <= pv) >> lo;
        int left, right;
        if (tiny && m <= TMAX) {
          // the table's verdict: where every key of the part comes from, and the loop's final (left, right)
          const uint32_t w = (uint32_t)__builtin_amdgcn_readfirstlane((int)s_tiny[tie_tiny_off(m) + (int)((uint32_t)GE | ((uint32_t)LE << m))]);
          const int from = in ? lo + (int)((w >> (3 * (lane - lo))) & 7u) : lane;
          k = (uint32_t)__builtin_amdgcn_ds_bpermute(from << 2, (int)k);
          ix = (uint32_t)__builtin_amdgcn_ds_bpermute(from << 2, (int)ix);
          left = (int)((w >> 24) & 15u);
          right = (int)(w >> 28) - 1;
        } else
        tie_hoare64(GE, LE, m, left, right, [&](int l, int r) {
          const uint32_t ka = tie_readlane(k, lo + l), kb = tie_readlane(k, lo + r);
          const uint32_t ia = tie_readlane(ix, lo + l), ib = tie_readlane(ix, lo + r);
          k = tie_writelane(kb, lo + l, k);
          k = tie_writelane(ka, lo + r, k);
          ix = tie_writelane(ib, lo + l, ix);
          ix = tie_writelane(ia, lo + r, ix);
        });
        const int rn = m - right - 1, ro = lo + right + 1;
        auto settle_pair = [&](int at) {
          const uint32_t ka = tie_readlane(k, at), kb = tie_readlane(k, at + 1);
          if (ka > kb) {
            const uint32_t ia = tie_readlane(ix, at), ib = tie_readlane(ix, at + 1);
            k = tie_writelane(kb, at, k);
            k = tie_writelane(ka, at + 1, k);
            ix = tie_writelane(ib, at, ix);
            ix = tie_writelane(ia, at + 1, ix);
          }
        };
        if (left == 2) settle_pair(lo);
        if (rn == 2) settle_pair(ro);
        // depth first, left part first: the right one waits (at most one pending part per level, < 64 levels)
        if (rn > 2) {
          stk = tie_writelane((uint32_t)ro | ((uint32_t)rn << 8), lsp, stk);
          lsp++;
        }
        if (left > 2) {
          m = left;  // lo stays
          continue;
        }
        if (lsp > 0) {
          lsp--;
          const uint32_t e = tie_readlane(stk, lsp);
          lo = (int)(e & 0xffu);
          m = (int)(e >> 8);
          continue;
        }
        break;
      }
      if (lane < n) {
        s_k[off + lane] = k;
        s_i[off + lane] = ix;
      }
      continue;
    }
    // ---- a part of more than 64 elements: masks and stopper lists in LDS ----
    const uint32_t pv = (uint32_t)__builtin_amdgcn_readfirstlane((int)s_k[off + (int)tie_rnd(seed, (uint32_t)n)]);
    const int nw = (n + 63) >> 6;
    int nl = 0;
    for (int w = 0; w < nw; w++) {  // left stoppers, ascending
      const int p = w * 64 + lane;
      const bool ge = p < n && s_k[off + p] >= pv, le = p < n && s_k[off + p] <= pv;
      const uint64_t bg = __ballot(ge), bl = __ballot(le);
      if (ge) s_L[nl + __builtin_popcountll(bg & ((1ull << lane) - 1))] = (uint16_t)p;
      if (lane == 0) {
        s_ge[w] = bg;
        s_le[w] = bl;
      }
      nl += __builtin_popcountll(bg);
    }
    int nr = 0;
    for (int w = nw - 1; w >= 0; w--) {  // right stoppers, descending
      const int p = w * 64 + lane;
      const bool le = p < n && s_k[off + p] <= pv;
      const uint64_t bl = __ballot(le);
      if (le) s_R[nr + __builtin_popcountll(bl & ~((2ull << lane) - 1))] = (uint16_t)p;
      nr += __builtin_popcountll(bl);
    }
    __syncthreads();
    // the reference swaps (L[t], R[t]) while L[t] < R[t]; L ascends and R descends, so that is a prefix
    const int lim = nl < nr ? nl : nr;
    int K = 0;
    for (int t0 = 0; t0 < lim; t0 += 64) {
      const int t = t0 + lane;
      const bool go = t < lim && s_L[t] < s_R[t];
      const uint64_t b = __ballot(go);
      if (go) {
        const int a = off + s_L[t], c = off + s_R[t];
        const uint32_t ka = s_k[a], ia = s_i[a];
        s_k[a] = s_k[c];
        s_i[a] = s_i[c];
        s_k[c] = ka;
        s_i[c] = ia;
      }
      K += __builtin_popcountll(b);
      if (b != ~0ull) break;
    }
    __syncthreads();
    // the scans after the last swap (no further swap can follow): on the masks, which still describe everything
    // strictly between the two positions
    int left = K > 0 ? (int)__builtin_amdgcn_readfirstlane((int)s_L[K - 1]) : -1;
    int right = K > 0 ? (int)__builtin_amdgcn_readfirstlane((int)s_R[K - 1]) : n;
    {
      int l = n;  // first GE bit at a position > left
      for (int p = left + 1; p < n;) {
        const uint64_t word = s_ge[p >> 6] & (~0ull << (p & 63));
        if (word) {
          l = (p & ~63) + __builtin_ctzll(word);
          break;
        }
        p = (p & ~63) + 64;
      }
      if (l > right) l = right;
      if (l == right) {
        left = l;
        right = l - 1;
      } else {
        int r = -1;  // last LE bit at a position < right
        for (int p = right - 1; p >= 0;) {
          const uint64_t word = s_le[p >> 6] & (~0ull >> (63 - (p & 63)));
          if (word) {
            r = (p & ~63) + 63 - __builtin_clzll(word);
            break;
          }
          p = (p & ~63) - 1;
        }
        if (r <= l) r = ((s_le[l >> 6] >> (l & 63)) & 1) ? l : l - 1;
        left = l;
        right = r;
        // (l < r here would be one more swap: cannot happen, (L[K], R[K]) was the next candidate pair and failed)
      }
    }
    const int rn = n - right - 1;
    if (left == 2) settle_pair_lds(off);
    if (rn == 2) settle_pair_lds(off + right + 1);
    __syncthreads();
    if (rn > 2) push(off + right + 1, rn);
    if (left > 2) push(off, left);
    __syncthreads();
  }
  __syncthreads();
  if (overflow) {
    if (lane == 0) tie_out[0] = -1;
    return;
  }
  // of equal points the first one in this order is the vertex (:6183); the triangulation carries the smallest index
  int np = 0;
  for (int i0 = 0; i0 < n0 && np <= VSM_DC_TIE_PATCHES; i0 += 64) {
    const int i = i0 + lane;
    const bool start = i < n0 && (i == 0 || s_k[i - 1] != s_k[i]) && i + 1 < n0 && s_k[i + 1] == s_k[i];
    uint64_t b = __ballot(start);
    while (b && np <= VSM_DC_TIE_PATCHES) {
      const int j = i0 + __builtin_ctzll(b);
      b &= b - 1;
      if (lane == 0) {
        const uint32_t key = s_k[j];
        uint32_t rep = s_i[j];
        for (int q = j + 1; q < n0 && s_k[q] == key; q++) rep = min(rep, s_i[q]);
        if (rep != s_i[j] && np < VSM_DC_TIE_PATCHES) {
          tie_out[1 + 2 * np] = (int32_t)rep;
          tie_out[2 + 2 * np] = (int32_t)s_i[j];
        }
        s_stack[0] = rep != s_i[j];
      }
      __syncthreads();
      np += (int)__builtin_amdgcn_readfirstlane((int)s_stack[0]);
      __syncthreads();
    }
  }
  if (lane == 0) tie_out[0] = np > VSM_DC_TIE_PATCHES ? -1 : np;
}

__global__ void __launch_bounds__(64) k_dc_ties(const VsmDcJob *__restrict__ jobs, int njobs, const uint32_t *__restrict__ tiny) {
  const VsmDcJob jb = jobs[blockIdx.x];
  if (!jb.tie_keys || !jb.tie_out) return;
  tie_sort<VSM_DC_TIE_POINTS>(TieFromKeys{jb.tie_keys}, jb.n_in, jb.tie_out, tiny);
}

// The same for the pairs of a look-ahead chunk, as soon as their compacted pass-2 lists exist (refinement does
// not move u1c, v1c).  Two kernels: the keys are copied out of the pair buffers first (those are overwritten two
// chunks later, the caller orders that behind this copy), then one wave per pair sorts its private copy.
__global__ void __launch_bounds__(256) k_dc_tie_keys(const VsmPair *__restrict__ pairs, int npairs, uint64_t *__restrict__ keys, int stride,
                                                     int32_t *__restrict__ counts) {
  const VsmPair pr = pairs[blockIdx.y];
  const int n = pr.count[1];
  if (blockIdx.x == 0 && threadIdx.x == 0) counts[blockIdx.y] = n;
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n && i < stride) {
    const vsm_p_match a = pr.list2[i];
    keys[(size_t)blockIdx.y * stride + i] = ((uint64_t)(uint32_t)(int32_t)a.u1c << 34) | ((uint64_t)(uint32_t)(int32_t)a.v1c << 20) | (uint32_t)i;
  }
}

__global__ void __launch_bounds__(64) k_dc_ties_of_keys(const uint64_t *__restrict__ keys, int stride, const int32_t *__restrict__ counts,
                                                        int32_t *__restrict__ tie_out, int out_stride, const uint32_t *__restrict__ tiny) {
  int32_t *out = tie_out + (size_t)blockIdx.x * out_stride;
  const int n = counts[blockIdx.x];
  if (n < 2 || n > stride) {
    if (threadIdx.x == 0) out[0] = n < 2 ? 0 : -1;
    return;
  }
  tie_sort<VSM_DC_TIE_POINTS>(TieFromKeys{keys + (size_t)blockIdx.x * stride}, n, out, tiny);
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

void vsm_dc_launch_ties(hipStream_t s, const VsmDcJob *d_jobs, int njobs) {
  if (njobs <= 0) return;
  hipLaunchKernelGGL(k_dc_ties, dim3(njobs), dim3(64), 0, s, d_jobs, njobs, dc2_tiny_table());
}

void vsm_dc_launch_tie_keys(hipStream_t s, const VsmPair *d_pairs, int npairs, int max_list, uint64_t *keys, int stride, int32_t *counts) {
  if (npairs <= 0) return;
  hipLaunchKernelGGL(k_dc_tie_keys, dim3((std::max(max_list, 1) + 255) / 256, npairs), dim3(256), 0, s, d_pairs, npairs, keys, stride, counts);
}
void vsm_dc_launch_ties_of_keys(hipStream_t s, int npairs, const uint64_t *keys, int stride, const int32_t *counts, int32_t *tie_out,
                                int out_stride) {
  if (npairs <= 0) return;
  hipLaunchKernelGGL(k_dc_ties_of_keys, dim3(npairs), dim3(64), 0, s, keys, stride, counts, tie_out, out_stride, dc2_tiny_table());
}

// =======================================================================================
// GPU-resident removeOutliers (VsmDc2Job, vsm_dc_gpu.h): keys -> (x,y) sort + duplicate removal + kd order ->
// block sub-trees -> merge levels through an LDS record cache -> tie patches, support votes -> survivors
// (-> prior statistics for pass 1).  Every kernel finds its work from the list length / point count in device memory.
// =======================================================================================

__global__ void __launch_bounds__(256) k_dc2_keys(const VsmDc2Job *__restrict__ jobs) {
  const VsmDc2Job jb = jobs[blockIdx.y];
  const int n = min(*jb.count, jb.cap);
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i == 0 && jb.h_n) *jb.h_n = n;
  if (i >= n) return;
  const vsm_p_match a = jb.list[i];
  const uint64_t k = ((uint64_t)(uint32_t)(int32_t)a.u1c << 34) | ((uint64_t)(uint32_t)(int32_t)a.v1c << 20) | (uint32_t)i;
  jb.keys_in[i] = k;
  if (jb.h_keys) jb.h_keys[i] = k;
  jb.remap[i] = i;
  jb.support[i] = 0;
}

// one stable LSD pass (7 bits at `sh`) over n 64-bit keys; thread t < KD_CHUNKS owns a contiguous chunk and its own
// histogram column, so the order inside a digit is the order of the source
__device__ inline void dc2_radix_pass(const uint64_t *__restrict__ src, uint64_t *__restrict__ dst, const int32_t n, const int sh,
                                      uint32_t *hist, uint32_t *tot) {
  const int t = threadIdx.x;
  const int32_t chunk = (n + KD_CHUNKS - 1) / KD_CHUNKS;
  for (int i = t; i < KD_DIGITS * KD_CHUNKS; i += KD_THREADS) hist[i] = 0;
  __syncthreads();
  if (t < KD_CHUNKS) {
    const int32_t i0 = t * chunk, i1 = min(n, i0 + chunk);
    for (int32_t i = i0; i < i1; i++) hist[((src[i] >> sh) & (KD_DIGITS - 1)) * KD_CHUNKS + t]++;
  }
  __syncthreads();
  {
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
    const int32_t i0 = t * chunk, i1 = min(n, i0 + chunk);
    for (int32_t i = i0; i < i1; i++) {
      const uint64_t k = src[i];
      dst[hist[((k >> sh) & (KD_DIGITS - 1)) * KD_CHUNKS + t]++] = k;
    }
  }
  __syncthreads();
}

#ifdef DC2_PHASE_TIMING  // experiments (tools/build_variant_dc.sh): cycles per phase, summed over blocks; [row][0] counts the blocks
__device__ unsigned long long dc2_dbg[16][16];
#define DC2_T(var) const long long var = clock64()
#define DC2_ACC(row, col, a, b) atomicAdd(&dc2_dbg[row][col], (unsigned long long)((b) - (a)))
__device__ void dc2_prepare_stat(long long a, long long b, long long c, long long d) {
  atomicAdd(&dc2_dbg[12][0], 1ull);
  atomicAdd(&dc2_dbg[12][1], (unsigned long long)a);
  atomicAdd(&dc2_dbg[12][2], (unsigned long long)b);
  atomicAdd(&dc2_dbg[12][3], (unsigned long long)c);
  atomicAdd(&dc2_dbg[12][4], (unsigned long long)d);
}
extern "C" int vsm_debug_dc2_phases(unsigned long long *out, int reset) {
  if (hipMemcpyFromSymbol(out, HIP_SYMBOL(dc2_dbg), sizeof(dc2_dbg)) != hipSuccess) return -1;
  if (reset) {
    static unsigned long long z[16][16];
    if (hipMemcpyToSymbol(HIP_SYMBOL(dc2_dbg), z, sizeof(z)) != hipSuccess) return -1;
  }
  return 0;
}
#else
#define DC2_T(var)
#define DC2_ACC(row, col, a, b)
#endif
// ExactDelaunay::prepare(defer_ties) on the device: stable sort of the keys by (x, y), the first key of every pixel
// stays (it carries the smallest input index; the vertex sort's verdict is patched in later), kd order.
template <int WIDE>
__device__ inline void dc2_prepare_global(const VsmDc2Job &jb, uint32_t *hist, uint32_t *tot) {  // KD_THREADS threads, KD_DIGITS * KD_CHUNKS words of hist
  __shared__ int32_t s_m;
  const int t = threadIdx.x;
  const int32_t nl = *jb.count;
  int32_t n = min(nl, jb.cap), m = 0;
  if (nl > jb.cap || nl > VSM_DC_KD_MAX_POINTS) {  // (not expected: the host sizes the slabs from the query counts)
    if (t == 0) *jb.error = 1;
    n = 0;
  }
#ifdef DC2_PHASE_TIMING
  const long long p0 = clock64();
  long long p1 = p0, p2 = p0, p3 = p0;
#endif
  if (n > 3) {  // the reference leaves lists of up to three matches alone (viso/matcher.cpp:1210)
    uint64_t *T0 = (uint64_t *)jb.kd_scratch, *T1 = T0 + jb.kd_stride;
    if (WIDE == 2 || (WIDE && n > KD_LDS_POINTS)) {
      const uint64_t *src = jb.keys_in;
      uint64_t *dst = T0;
      for (int sh = 20; sh <= 41; sh += 7) {
        kd_radix_pass_wide<uint64_t>(n, hist, tot, [&](int32_t i) { return src[i]; }, [&](uint64_t v) { return (uint32_t)(v >> sh) & (KD_DIGITS - 1); },
                                     [&](int32_t at, uint64_t v) { dst[at] = v; });
        src = dst;
        dst = dst == T0 ? T1 : T0;
      }  // (four passes: the result is in T1)
    } else {
      dc2_radix_pass(jb.keys_in, T0, n, 20, hist, tot);
      dc2_radix_pass(T0, T1, n, 27, hist, tot);
      dc2_radix_pass(T1, T0, n, 34, hist, tot);
      dc2_radix_pass(T0, T1, n, 41, hist, tot);
    }
#ifdef DC2_PHASE_TIMING
    p1 = clock64();
#endif
    const int32_t per = (n + KD_THREADS - 1) / KD_THREADS;
    const int32_t i0 = min(n, t * per), i1 = min(n, i0 + per);
    uint32_t cnt = 0;
    for (int32_t i = i0; i < i1; i++) cnt += (i == 0 || VSM_KXY(T1[i]) != VSM_KXY(T1[i - 1])) ? 1u : 0u;
    uint32_t pos = kd_block_scan(cnt, tot);
    for (int32_t i = i0; i < i1; i++)
      if (i == 0 || VSM_KXY(T1[i]) != VSM_KXY(T1[i - 1])) jb.key_sorted[pos++] = T1[i];
    if (t == KD_THREADS - 1) s_m = (int32_t)pos;
    __syncthreads();
    m = s_m;
  }
  if (t == 0) {
    jb.mn[0] = m;
    jb.mn[1] = n;
  }
#ifdef DC2_PHASE_TIMING
  p2 = p3 = clock64();
  if (m >= 2) kd_order_body<WIDE>(jb.key_sorted, m, jb.kd_scratch, jb.kd_stride, jb.key, hist, tot, &p3);
  if (t == 0) {
    const long long p4 = clock64();
    dc2_prepare_stat(p1 - p0, p2 - p1, p3 - p2, p4 - p3);
  }
#else
  if (m >= 2) kd_order_body<WIDE>(jb.key_sorted, m, jb.kd_scratch, jb.kd_stride, jb.key, hist, tot);
#endif
}

__global__ void __launch_bounds__(KD_THREADS) k_dc2_prepare(const VsmDc2Job *__restrict__ jobs) {
  __shared__ uint32_t hist[KD_DIGITS * KD_CHUNKS];
  __shared__ uint32_t tot[KD_THREADS / 64 + 1];
  dc2_prepare_global<1>(jobs[blockIdx.x], hist, tot);
}
// ... the lists with more matches than `longer_than` only (the others belong to k_dc2_prepare_lds of the same launch pair):
// 12 KB of LDS, so a workgroup that finds nothing to do comes and goes beside whatever else is resident (with the 128 KB of
// the kernel above the launch waited 70-80 us for compute units to itself in the pipeline, long list or not)
__global__ void __launch_bounds__(KD_THREADS) k_dc2_prepare_long(const VsmDc2Job *__restrict__ jobs, int longer_than) {
  __shared__ uint32_t lds[3 * KD_WIDE_E * (KD_THREADS / 64)];
  __shared__ uint32_t tot[KD_THREADS / 64 + 1];
  static_assert(3 * KD_WIDE_E * (KD_THREADS / 64) >= KD_DIGITS * (KD_THREADS / 64), "the wide radix pass counts in the same words");
  if (*jobs[blockIdx.x].count <= longer_than) return;
  dc2_prepare_global<2>(jobs[blockIdx.x], lds, tot);
}

// ---------------------------------------------------------------------------------------
// The same preparation for lists of up to DC2_PREP_LDS matches, entirely inside LDS and sized by the launch's longest
// list (cap2 = a power of two >= it; 16 bytes of LDS per entry: 16 KB for a first-pass list of 700 matches, 128 KB for
// 8192).  The histogram form above runs 256 of its 1024 threads through global scratch six times (four 7-bit passes by
// (x, y), two by y) and its kd levels recompute every position's node from the root at every level; here
//   * the (x, y, input index) order is ONE bitonic sort of the 64-bit keys (the index in the low bits makes it the
//     stable order), stages whose partner lies inside a thread's own E entries run in registers,
//   * the y order of the distinct points is a second one over 32-bit words y << 13 | rank,
//   * a thread keeps the kd node (offset, size) of its E positions in registers and halves it once per level.
// Same output as k_dc2_prepare (mn, key_sorted, key): the block and merge kernels do not care which one ran.
// ---------------------------------------------------------------------------------------
#define DC2_PREP_LDS 8192
extern __shared__ __attribute__((aligned(16))) uint8_t dc2_prep_lds[];

__device__ inline uint32_t dc2_scan(uint32_t v, uint32_t *tot, int nwaves) {  // exclusive prefix over the workgroup
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
    const uint32_t t = lane < nwaves ? tot[lane] : 0;
    uint32_t s = t;
#pragma unroll
    for (int d = 1; d < 16; d <<= 1) {
      const uint32_t y = __shfl_up(s, d, 64);
      if (lane >= d) s += y;
    }
    if (lane < nwaves) tot[lane] = s - t;
  }
  __syncthreads();
  const uint32_t r = x - v + tot[wv];
  __syncthreads();
  return r;
}

// the same with ONE barrier: every wave adds up the other waves' totals itself; tot2 = two rows of 16 words used in turn
// (`par` alternates from call to call, and the caller has a barrier of its own between two calls)
__device__ inline uint32_t dc2_scan1(uint32_t v, uint32_t *tot2, int par, int nwaves) {
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  uint32_t x = v;
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    const uint32_t y = __shfl_up(x, d, 64);
    if (lane >= d) x += y;
  }
  uint32_t *tot = tot2 + 16 * par;
  if (lane == 63) tot[wv] = x;
  __syncthreads();
  const uint32_t w = lane < nwaves ? tot[lane] : 0;
  uint32_t s = w;
#pragma unroll
  for (int d = 1; d < 16; d <<= 1) {
    const uint32_t y = __shfl_up(s, d, 64);
    if (lane >= d) s += y;
  }
  return x - v + __shfl(s - w, wv, 64);
}
__device__ inline void dc2_wave_sync() {
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
  __builtin_amdgcn_wave_barrier();
}
// One stable LSD pass (7 bits at `sh`) over the cap = E * blockDim.x entries of a[] in LDS, in place: every thread takes E
// entries into registers - wave w the entries [64 E w, 64 E (w + 1)), 64 consecutive ones per batch -, ranks them inside the
// wave (the lanes that share a digit find each other with seven ballots; the wave's count per digit lives in
// hist[digit][wave]), one exclusive scan over hist in (digit, wave) order gives every wave its start per digit, and the
// entries go back to their new places.  No atomics, no second buffer; the order inside a digit is the order of the source.
template <typename K, int E>
__device__ inline void dc2_radix_lds(K *a, const int sh, uint32_t *hist, uint32_t *tot) {
  const int t = threadIdx.x, lane = t & 63, wv = t >> 6, nw = (int)blockDim.x >> 6;
  hist[2 * t] = 0;
  hist[2 * t + 1] = 0;  // 128 digits x nw waves = 2 x blockDim.x counters
  __syncthreads();
  K key[E];
  uint32_t place[E];  // digit * nw + wave, and the entry's number among the wave's entries of that digit so far
  uint32_t rank[E];
  const uint64_t lt = lane == 0 ? 0ull : (~0ull >> (64 - lane));
#pragma unroll
  for (int r = 0; r < E; r++) {
    key[r] = a[wv * 64 * E + 64 * r + lane];
    const uint32_t d = (uint32_t)(key[r] >> sh) & 127u;
    uint64_t same = ~0ull;
#pragma unroll
    for (int bit = 0; bit < 7; bit++) {
      const bool on = (d >> bit) & 1u;
      const uint64_t bal = __builtin_amdgcn_ballot_w64(on);
      same &= on ? bal : ~bal;
    }
    place[r] = d * (uint32_t)nw + (uint32_t)wv;
    const uint32_t before = (uint32_t)__popcll(same & lt);
    const uint32_t prior = hist[place[r]];
    dc2_wave_sync();
    if (before == 0) hist[place[r]] = prior + (uint32_t)__popcll(same);
    dc2_wave_sync();
    rank[r] = prior + before;
  }
  __syncthreads();
  {
    const uint32_t v0 = hist[2 * t], v1 = hist[2 * t + 1];
    const uint32_t run = dc2_scan(v0 + v1, tot, nw);
    hist[2 * t] = run;
    hist[2 * t + 1] = run + v0;
  }
  __syncthreads();
#pragma unroll
  for (int r = 0; r < E; r++) a[hist[place[r]] + rank[r]] = key[r];
  __syncthreads();
}

template <int E>
__global__ void __launch_bounds__(1024) k_dc2_prepare_lds(const VsmDc2Job *__restrict__ jobs, int cap2, int big_too) {
  __shared__ uint32_t tot[33];
  __shared__ int32_t s_m;
  const VsmDc2Job jb = jobs[blockIdx.x];
  const int t = threadIdx.x, nwaves = (int)blockDim.x >> 6;
  const int32_t nl = *jb.count;
  // A list of a launch that may hold longer ones (the host only knows the query counts).  big_too 2: left to
  // k_dc2_prepare_long, launched right behind this kernel for exactly those lists with every thread at work (inlined here
  // those forms cost the short lists their registers) - where long lists are expected (vsm_dc2_launch_prepare); big_too 1:
  // the odd long list of a launch of short ones takes the narrow global-memory form here, on this kernel's LDS (the launch is
  // KD_THREADS wide and 128 KB deep then), and tells the host (the launches after it bring the other kernel along).
  if (big_too == 2 && nl > cap2) return;
  if (big_too && nl > cap2) {
    if (threadIdx.x == 0 && jb.long_seen) *jb.long_seen = 1;
    dc2_prepare_global<0>(jb, (uint32_t *)dc2_prep_lds, tot);
    return;
  }
  int32_t n = min(nl, jb.cap), m = 0;
  if (nl > jb.cap || nl > cap2) {  // (not expected: the host sizes slabs and launches from the query counts)
    if (t == 0) *jb.error = 1;
    n = 0;
  }
  uint64_t *K = (uint64_t *)dc2_prep_lds;                        // [cap2] keys
  uint32_t *YK = (uint32_t *)(dc2_prep_lds + 8 * (size_t)cap2);  // [cap2] y << 13 | rank
  uint32_t *hist = (uint32_t *)(dc2_prep_lds + 14 * (size_t)cap2);  // [2 blockDim.x] the radix passes' counters
  DC2_T(p0);
  if (n > 3) {  // the reference leaves lists of up to three matches alone (viso/matcher.cpp:1210)
#pragma unroll
    for (int e = 0; e < E; e++) {
      const int i = E * t + e;
      K[i] = i < n ? jb.keys_in[i] : ~0ull;
    }
    __syncthreads();
    for (int sh = 20; sh < 48; sh += 7) dc2_radix_lds<uint64_t, E>(K, sh, hist, tot);
    DC2_T(p1);
    if (t == 0) DC2_ACC(12, 1, p0, p1);
    // the first key of every pixel stays
    uint64_t r[E];
    uint32_t first = 0, cnt = 0;
#pragma unroll
    for (int e = 0; e < E; e++) {
      const int i = E * t + e;
      r[e] = K[i];
      const uint64_t prev = e > 0 ? r[e - 1] : (i > 0 ? K[i - 1] : 0);
      if (i < n && (i == 0 || VSM_KXY(r[e]) != VSM_KXY(prev))) {
        first |= 1u << e;
        cnt++;
      }
    }
    uint32_t pos = dc2_scan(cnt, tot, nwaves);
#pragma unroll
    for (int e = 0; e < E; e++) {
      if (first & (1u << e)) {
        jb.key_sorted[pos] = r[e];
        YK[pos] = ((uint32_t)(r[e] >> 20) & 0x3fffu) << 13 | pos;
        pos++;
      }
    }
    if (t == (int)blockDim.x - 1) s_m = (int32_t)pos;
    __syncthreads();
    m = s_m;
  }
  if (t == 0) {
    jb.mn[0] = m;
    jb.mn[1] = n;
  }
  if (m < 2) return;
  DC2_T(p2);
  // ---- y order ----
  for (int i = m + t; i < cap2; i += (int)blockDim.x) YK[i] = 0xffffffffu;
  __syncthreads();
  dc2_radix_lds<uint32_t, E>(YK, 13, hist, tot);
  dc2_radix_lds<uint32_t, E>(YK, 20, hist, tot);
  DC2_T(p3);
  // ---- kd lists: X (x order), Y (y order), their inverses, double buffers, the scan words ----
  uint32_t yk[E];
#pragma unroll
  for (int e = 0; e < E; e++) yk[e] = E * t + e < m ? (YK[E * t + e] & 0x1fffu) : 0;
  __syncthreads();
  uint16_t *l16 = (uint16_t *)dc2_prep_lds;
  uint16_t *X0 = l16, *X1 = X0 + cap2, *Y0 = X1 + cap2, *Y1 = Y0 + cap2, *PX = Y1 + cap2, *PY = PX + cap2, *P = PY + cap2;
#pragma unroll
  for (int e = 0; e < E; e++) {
    const int q = E * t + e;
    if (q < m) {
      X0[q] = (uint16_t)q;
      PX[q] = (uint16_t)q;
      Y0[q] = (uint16_t)yk[e];
      PY[yk[e]] = (uint16_t)q;
    }
  }
  __syncthreads();
  // ---- kd levels (kd_levels above), the node of each of the thread's positions carried along ----
  int32_t off[E], nn[E];
#pragma unroll
  for (int e = 0; e < E; e++) {
    off[e] = 0;
    nn[e] = m;
  }
  uint16_t *X = X0, *Xn = X1, *Y = Y0, *Yn = Y1;
  for (int depth = 0; ((m + (1 << depth) - 1) >> depth) > 3; depth++) {
    const bool cut_x = (depth & 1) == 0;
    const uint16_t *S = cut_x ? Y : X;  // the list to partition
    uint16_t *D = cut_x ? Yn : Xn;
    const uint16_t *PO = cut_x ? PX : PY;  // position in the list that is cut in place
    uint16_t *PS = cut_x ? PY : PX;
    typedef uint16_t DcRow __attribute__((ext_vector_type(E), may_alias));
    uint32_t left = 0, sum = 0;
    uint32_t el[E];
    {
      const DcRow row = *(const DcRow *)(S + E * t);  // the thread's E entries in one read
#pragma unroll
      for (int e = 0; e < E; e++) el[e] = E * t + e < m ? row[e] : 0;
    }
#pragma unroll
    for (int e = 0; e < E; e++) {
      const int q = E * t + e;
      if (q < m && (nn[e] <= 3 || (int32_t)PO[el[e]] < off[e] + (nn[e] >> 1))) {
        left |= 1u << e;
        sum++;
      }
    }
    uint32_t run = dc2_scan1(sum, tot, depth & 1, nwaves);
    uint32_t mine[E];  // flags before each of the thread's positions
    {
      DcRow row;
#pragma unroll
      for (int e = 0; e < E; e++) {
        mine[e] = run;
        row[e] = (uint16_t)run;
        run += (left >> e) & 1u;
      }
      *(DcRow *)(P + E * t) = row;
    }
    __syncthreads();
    uint32_t at_off = 0;  // flags before the node's first position (consecutive positions mostly share their node)
#pragma unroll
    for (int e = 0; e < E; e++) {
      const int q = E * t + e;
      if (e == 0 || off[e] != off[e - 1]) at_off = P[off[e]];
      if (q < m) {
        const uint32_t before = mine[e] - at_off;
        const uint32_t np = ((left >> e) & 1u) ? off[e] + before : off[e] + (nn[e] >> 1) + ((q - off[e]) - before);
        D[np] = (uint16_t)el[e];
        PS[el[e]] = (uint16_t)np;
      }
      // the node of position q one level down (kd_node_at)
      const int32_t div = nn[e] >> 1;
      if (q < off[e] + div) {
        nn[e] = div;
      } else {
        off[e] += div;
        nn[e] -= div;
      }
    }
    __syncthreads();
    if (cut_x) {
      uint16_t *w = Y;
      Y = Yn;
      Yn = w;
    } else {
      uint16_t *w = X;
      X = Xn;
      Xn = w;
    }
  }
  DC2_T(p4);
#pragma unroll
  for (int e = 0; e < E; e++) {
    const int q = E * t + e;
    if (q < m) jb.key[q] = jb.key_sorted[X[q]];
  }
#ifdef DC2_PHASE_TIMING
  if (t == 0) {
    const long long p5 = clock64();
    atomicAdd(&dc2_dbg[12][0], 1ull);
    DC2_ACC(12, 3, p2, p3);
    DC2_ACC(12, 4, p3, p4);
    DC2_ACC(12, 2, p4, p5);
  }
#endif
}

// The divide-and-conquer tree is ExactDelaunay::build_tree's: [off, off+n) splits at n >> 1 while n > block points.
// A node is named by its depth and path from the root (bit per level, most significant first) and has heap index
// (1 << depth) | path.  Returns false if the path runs past a sub-tree that became a block earlier.
__device__ inline bool dc2_walk(int32_t m, int depth, uint32_t path, int32_t &off, int32_t &n, int &axis) {
  off = 0;
  n = m;
  axis = 0;
  for (int b = depth - 1; b >= 0; b--) {
    if (n <= VSM_DC_BLOCK_POINTS) return false;
    const int32_t div = n >> 1;
    if ((path >> b) & 1) {
      off += div;
      n -= div;
    } else {
      n = div;
    }
    axis = 1 - axis;
  }
  return true;
}

// REACH WORDS.  A merge level brings into LDS the triangles whose circumcircle reaches across its cut; whoever writes a
// triangle record to global memory therefore leaves the circle's extent with it, in the two words of the 32-byte record
// that are otherwise unused: word 7 = x extent, word 3 = y extent, each lo | hi << 16 as signed 16-bit pixels, rounded
// outward with 1.5 pixels of margin.  Vertices below 2^14 make every product of the circumcentre exact in double, so the
// extent is right to a billionth of a pixel before the margin: the band's closure rests on it being conservative.  A hull
// triangle (ghost corner) reaches everywhere, an unused slot nowhere; a level then decides a slot from ONE word.
#define DC2_REACH_ALL 0x7fff8000u   // lo = -32768, hi = 32767
#define DC2_REACH_NONE 0x80007fffu  // lo = 32767, hi = -32768
__device__ inline uint32_t dc2_reach_word(double c, double rad) {
  const double lo = floor(c - rad), hi = ceil(c + rad);
  if (!(lo == lo) || !(hi == hi)) return DC2_REACH_ALL;
  const int32_t l = lo < -32768.0 ? -32768 : (lo > 32767.0 ? 32767 : (int32_t)lo);
  const int32_t h = hi > 32767.0 ? 32767 : (hi < -32768.0 ? -32768 : (int32_t)hi);
  return ((uint32_t)l & 0xffffu) | ((uint32_t)h << 16);
}
// pa, pb, pc = x | y << 16
__device__ inline void dc2_reach(uint32_t pa, uint32_t pb, uint32_t pc, uint32_t &wx, uint32_t &wy) {
  const double ax = (double)(pa & 0xffffu), ay = (double)(pa >> 16);
  const double bx = (double)(pb & 0xffffu) - ax, by = (double)(pb >> 16) - ay;
  const double cx = (double)(pc & 0xffffu) - ax, cy = (double)(pc >> 16) - ay;
  const double d = 2.0 * (bx * cy - by * cx);
  const double b2 = bx * bx + by * by, c2 = cx * cx + cy * cy;
  const double ux = (cy * b2 - by * c2) / d, uy = (bx * c2 - cx * b2) / d;  // circumcentre relative to a
  const double rad = sqrt(ux * ux + uy * uy) + 1.5;
  if (!(d == d) || d == 0.0 || !(rad == rad)) {
    wx = wy = DC2_REACH_ALL;
    return;
  }
  wx = dc2_reach_word(ax + ux, rad);
  wy = dc2_reach_word(ay + uy, rad);
}
__device__ inline bool dc2_reaches(uint32_t w, bool left_half, int32_t cl, int32_t cr) {
  const int32_t lo = (int32_t)(int16_t)(w & 0xffffu), hi = (int32_t)(int16_t)(w >> 16);
  return left_half ? hi >= cr : lo <= cl;
}

// One workgroup per block sub-tree (<= VSM_DC_BLOCK_POINTS points) on the edge-word LDS mesh (vsm_dc_lds.h): 15 KB of words,
// 2 KB of points, 4 KB of hull handles.  The block is cut down to Triangle's own leaves of two or three points
// (DC2_BLOCK_DEPTH = 8 halvings, the same rule as the host's tree): 256 lanes build a leaf each - a handful of stores, no
// walk - and the eight levels above are merged level by level, a node per lane: a quarter of the block per wave (the four
// waves sit on the CU's four SIMDs), the two top levels on waves 0, 1 / wave 0.  The seam loop (dc2_zip) is straight-line
// code under selects, so the lanes of a wave that walk different seams mostly execute the same instructions; what is left
// of the divergence is the trip count.  Measured alone, 67 lists of 7.4 k: ~340 us (round 2's form - 64 leaves of <= 14
// points by a per-lane recursion, six levels - took 466).
#ifndef DC2_BLOCK_WAVES
#define DC2_BLOCK_WAVES 4
#endif
#define DC2_BLOCK_THREADS (64 * DC2_BLOCK_WAVES)
#ifndef DC2_BLOCK_PACKED
#define DC2_BLOCK_PACKED 16  // levels with at least this many nodes run lane by lane through the workgroup
#endif
#ifndef DC2_BLOCK_OCC
#define DC2_BLOCK_OCC 5
#endif
#ifndef DC2_BLOCK_TAIL
// waves that carry the levels below DC2_BLOCK_PACKED nodes (and the write-back); the others END early (s_endpgm) and the
// workgroup's later __syncthreads() count the surviving waves only.  That is how gfx9 hardware works (s_barrier waits for the
// waves of the workgroup that have not terminated), not something HIP's programming model promises: the early exits are
// compiled for gfx9 targets only - elsewhere every wave stays to the end (DC2_BLOCK_TAIL = DC2_BLOCK_WAVES); this library is
// built for gfx950 and nothing else, and test_gpu_resident_remove_outliers_chain / test_delaunay_subtrees_on_gpu pin the
// behaviour (a wave waiting for a terminated one would hang them).
#if defined(__gfx900__) || defined(__gfx906__) || defined(__gfx908__) || defined(__gfx90a__) || defined(__gfx940__) || defined(__gfx941__) || \
    defined(__gfx942__) || defined(__gfx950__) || !defined(__HIP_DEVICE_COMPILE__)
#define DC2_BLOCK_TAIL 1
#else
#define DC2_BLOCK_TAIL DC2_BLOCK_WAVES
#endif
#endif
__global__ void __launch_bounds__(DC2_BLOCK_THREADS, DC2_BLOCK_OCC) k_dc2_block(const VsmDc2Job *__restrict__ jobs, int depth) {
  __shared__ __attribute__((aligned(16))) uint32_t s_w[2 * VSM_DC_BLOCK_POINTS * 4];  // edge words (vsm_dc_lds.h)
  __shared__ uint32_t s_pt[VSM_DC_BLOCK_POINTS];
  __shared__ Dc2Hull16 s_hull[2 << DC2_BLOCK_DEPTH];
  const VsmDc2Job j2 = jobs[blockIdx.y];
  const int32_t m = j2.mn[0];
  if (m < 2) return;
  // the block at the end of path p (depth bits); a sub-tree that fits a block at a smaller depth d is taken by the
  // path whose remaining bits are zero
  int32_t boff = 0, bn = m;
  int baxis = 0, d = 0;
  uint32_t bidx = 1;
  const uint32_t path = blockIdx.x;
  for (int b = depth - 1; b >= 0 && bn > VSM_DC_BLOCK_POINTS; b--, d++) {
    const int32_t div = bn >> 1;
    if ((path >> b) & 1) {
      boff += div;
      bn -= div;
      bidx = 2 * bidx + 1;
    } else {
      bn = div;
      bidx = 2 * bidx;
    }
    baxis = 1 - baxis;
  }
  if (bn > VSM_DC_BLOCK_POINTS) {  // (the host chose too small a depth)
    if (threadIdx.x == 0) *j2.error = 2;
    return;
  }
  if (d < depth && (path & ((1u << (depth - d)) - 1u)) != 0) return;
  const int lane = threadIdx.x;
  const int wv = lane >> 6, wl = lane & 63;  // wave, lane in the wave
  DC2_T(c0);
  {
    dc2_v4u ones;
    ones.x = ones.y = ones.z = ones.w = 0xffffffffu;
    for (int i = lane; i < 2 * bn; i += DC2_BLOCK_THREADS) ((dc2_v4u *)s_w)[i] = ones;
  }
  __syncthreads();
  DcBlockMesh mesh;
  mesh.w = (DC2_AS3 dc2_u32a *)s_w;
  mesh.pt = (DC2_AS3 const uint32_t *)s_pt;
  mesh.key = j2.key + boff;  // (a leaf reads its two or three keys where they lie)
  mesh.ptw = (DC2_AS3 uint32_t *)s_pt;
  mesh.gid = j2.id + boff;
  DC2_T(c1);
  // leaf of virtual lane v: the two top bits of a path - the quarter of the block - are the wave
  {
    constexpr int kPerWave = (1 << DC2_BLOCK_DEPTH) / DC2_BLOCK_WAVES;
    for (int v = wl; v < kPerWave; v += 64) dc2_block_leaf_run(mesh, wv * kPerWave + v, bn, baxis, (DC2_AS3 Dc2Hull16 *)s_hull);
  }
  __syncthreads();
  DC2_T(c2);
#ifdef DC2_PHASE_TIMING
  long long cl = c2;
#endif
  for (int L = DC2_BLOCK_DEPTH - 1; L >= 0; L--) {
    {
      // node j of level L (2^L nodes).  The short seams of the lower levels side by side in as few waves as possible (the lanes
      // of a wave mostly execute the same instructions, and a wave costs a SIMD the same issue slots whether one lane walks
      // or sixty-four: 16 nodes on one wave take 0.7 of the time they take as 4 x 4, and leave the other SIMDs to whatever
      // else runs on the compute unit); from 8 nodes up the seams are long, differ in length, and go one or two to a wave.
      const int nodes = 1 << L;
      int node = -1;  // (one call site: the seam walk is a lot of code, and the instruction cache is shared)
      if (nodes >= DC2_BLOCK_PACKED) {
        // a wave that has no node at this level has none at any level above it: it ENDS here.  Its registers and its wave
        // slot go back to the compute unit at once (the workgroup's LDS stays until the last wave is through) and
        // s_barrier counts the surviving waves only.  What this is for is not this kernel: five resident workgroups of
        // four waves hold 5 x 96 of a SIMD's 512 vector registers, and a k_match wave (104) then has nowhere to go on that
        // compute unit for as long as the block's single-lane upper levels take (DESIGN_HISTORY.md section 6c).
        if (DC2_BLOCK_TAIL < DC2_BLOCK_WAVES && wv >= DC2_BLOCK_TAIL && wv * 64 >= nodes) return;
        node = lane < nodes ? lane : -1;
      } else if (nodes >= DC2_BLOCK_TAIL) {
        const int per_wave = nodes / DC2_BLOCK_TAIL;
        if (wv >= DC2_BLOCK_TAIL) return;
        node = wl < per_wave ? wv * per_wave + wl : -1;
      } else {
        if (wv >= nodes && DC2_BLOCK_TAIL < DC2_BLOCK_WAVES) return;
        node = wl == 0 && wv < nodes ? wv : -1;
      }
      if (node >= 0) dc2_block_merge_run(mesh, node, L, bn, baxis, (DC2_AS3 Dc2Hull16 *)s_hull);
    }
    __syncthreads();
#ifdef DC2_PHASE_TIMING
    {
      const long long cn = clock64();
      if (lane == 0) DC2_ACC(0, 3 + L, cl, cn);
      cl = cn;
    }
#endif
  }
  DC2_T(c3);
  // records out under global numbering: neighbour handles + 8 boff, vertices + boff (a record is two 16-byte stores)
  dc2_v4i *gt = (dc2_v4i *)(j2.tri + (size_t)2 * boff * 8);
  constexpr int kLive = DC2_BLOCK_TAIL < DC2_BLOCK_WAVES ? 64 : DC2_BLOCK_THREADS;  // (with early exits only wave 0 gets here)
  for (int t = lane; t < 2 * bn; t += kLive) {
    const dc2_v4u o = ((const dc2_v4u *)s_w)[t];
    auto nb = [&](uint32_t v) -> int32_t { return (v & 0xffffu) == 0xffffu ? -1 : (int32_t)(v & 0xffffu) + 8 * boff; };
    auto vx = [&](uint32_t v) -> int32_t { return (v >> 16) == 0xffffu ? -1 : (int32_t)(v >> 16) + boff; };
    dc2_v4i a, b;
    a.x = nb(o.x);
    a.y = nb(o.y);
    a.z = nb(o.z);
    b.x = vx(o.x);
    b.y = vx(o.y);
    b.z = vx(o.z);
    uint32_t wx, wy;  // reach words (above)
    if ((b.x & b.y & b.z) < 0)
      wx = wy = DC2_REACH_NONE;
    else if ((b.x | b.y | b.z) < 0)
      wx = wy = DC2_REACH_ALL;
    else
      dc2_reach(s_pt[o.x >> 16], s_pt[o.y >> 16], s_pt[o.z >> 16], wx, wy);
    a.w = (int32_t)wy;
    b.w = (int32_t)wx;
    gt[2 * t] = a;
    gt[2 * t + 1] = b;
  }
  for (int i = lane; i < bn; i += kLive) j2.pt[boff + i] = s_pt[i];
  if (lane == 0) {
    const Dc2Hull16 hl = s_hull[1];
    j2.hulls[bidx] = VsmDcHull{hl.fl_t + 2 * boff, hl.fl_o, hl.fr_t + 2 * boff, hl.fr_o};
  }
#ifdef DC2_PHASE_TIMING
  if (lane == 0) {
    const long long c4 = clock64();
    atomicAdd(&dc2_dbg[0][0], 1ull);
    DC2_ACC(0, 1, c0, c1);
    DC2_ACC(0, 2, c1, c2);
    DC2_ACC(0, 13, c3, c4);
    DC2_ACC(0, 14, c0, c4);
  }
#endif
}

// ---------------------------------------------------------------------------------------
// Merge levels above the blocks.  mergehulls walks a seam: every step reads a few triangle records and points that
// depend on the previous step, and in global memory each of those is an L2 round trip.  A workgroup takes one merge
// node and brings into LDS ONLY what the seam can touch - the band:
//   * the hull (ghost) triangles of both halves (the handles travel along them),
//   * the triangles whose circumcircle reaches across the cut (only those can be dissolved: an edge goes when a point of
//     the other half lies inside the circle through its triangle),
//   * the neighbours of all of those (read for the apex across an edge, written when a bond changes),
//   * the node's two new slots,
// a few hundred records out of the node's 2 n (535 of 14 800 for the top merge of a 7.4 k list).  The band gets a
// compact numbering of its own - line = number of band slots before it (rank in a bit mask), the points its records use
// likewise - so the walk runs on the edge-word mesh of vsm_dc_lds.h (DcBandMesh): 8-50 KB of LDS per node instead of 36
// bytes per point of the node (the top merges used to hold a compute unit's LDS each while one lane walked), nothing to
// translate during the walk, and the node itself may be of any size.  One lane then zips the seam (dc2_zip); afterwards
// all lanes write the band's records back under global numbering.  A neighbour that is not in the band is the handle of
// the band's TRAP record (vsm_dc_lds.h): a walk that would need what lies behind it - it does not happen on the test
// sets: the band is closed under the walk's accesses, DESIGN_HISTORY.md section 6b - poisons its own check or leaves a store in
// the trap, and the node is then redone by one lane on the records in global memory, which are untouched until the
// write-back.  The same happens to a node whose band does not fit the LDS the level was launched with.
// ---------------------------------------------------------------------------------------
#ifndef DC2_MERGE_THREADS
#define DC2_MERGE_THREADS 256
#endif
#ifndef DC2_BAND_BATCH
#define DC2_BAND_BATCH 8  // independent loads per thread in the band's two streaming passes (extent, reach words)
#endif
#ifdef DC2_PHASE_TIMING
#define DC2_BAND_STAT(level, col, v) atomicAdd(&dc2_dbg[1 + (level)][col], (unsigned long long)(v))
#else
#define DC2_BAND_STAT(level, col, v)
#endif

// exclusive prefix of the population counts of `words` mask words (256 threads; pre[w] = set bits before word w); returns the total
__device__ inline uint32_t dc2_mask_prefix(const uint32_t *mask, uint16_t *pre, int words, uint32_t *tot) {
  const int t = threadIdx.x, lane = t & 63, wv = t >> 6;
  const int per = (words + DC2_MERGE_THREADS - 1) / DC2_MERGE_THREADS;
  const int w0 = min(words, t * per), w1 = min(words, w0 + per);
  uint32_t sum = 0;
  for (int w = w0; w < w1; w++) sum += __popc(mask[w]);
  uint32_t x = sum;
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    const uint32_t y = __shfl_up(x, d, 64);
    if (lane >= d) x += y;
  }
  if (lane == 63) tot[wv] = x;
  __syncthreads();
  uint32_t base = 0, total = 0;
  for (int k = 0; k < DC2_MERGE_THREADS / 64; k++) {
    base += k < wv ? tot[k] : 0u;
    total += tot[k];
  }
  uint32_t run = base + x - sum;
  for (int w = w0; w < w1; w++) {
    pre[w] = (uint16_t)run;
    run += __popc(mask[w]);
  }
  __syncthreads();
  return total;
}
__device__ inline bool dc2_bit(const uint32_t *mask, int i) { return (mask[i >> 5] >> (i & 31)) & 1u; }
__device__ inline int dc2_rank(const uint32_t *mask, const uint16_t *pre, int i) {
  return (int)pre[i >> 5] + __popc(mask[i >> 5] & ((1u << (i & 31)) - 1u));
}

// dynamic LDS: [lines_cap][8] 16-bit records, [pts_cap] points, [lines_cap] line -> slot, [pts_cap] compact point -> point,
// [words_cap] band mask, [words_cap] ring mask, [words_cap / 2] point mask, their prefixes (16-bit)
__global__ void __launch_bounds__(DC2_MERGE_THREADS) k_dc2_merge(const VsmDc2Job *__restrict__ jobs, int level, int lines_cap, int pts_cap, int words_cap) {
  extern __shared__ __attribute__((aligned(16))) uint8_t dc2_lds[];
  uint16_t *s_rec = (uint16_t *)dc2_lds;
  uint32_t *s_pt = (uint32_t *)(s_rec + (size_t)lines_cap * 8);
  uint32_t *s_l2s = s_pt + pts_cap;
  uint32_t *s_c2p = s_l2s + lines_cap;
  uint32_t *s_band = s_c2p + pts_cap;
  uint32_t *s_ring = s_band + words_cap;
  uint32_t *s_pmask = s_ring + words_cap;
  uint16_t *s_bpre = (uint16_t *)(s_pmask + (words_cap + 1) / 2);
  uint16_t *s_ppre = s_bpre + words_cap;
  __shared__ int32_t s_cl, s_cr, s_fail;
  __shared__ uint32_t s_tot[DC2_MERGE_THREADS / 64];
  const VsmDc2Job jb = jobs[blockIdx.y];
  const int32_t m = jb.mn[0];
  if (m < 2) return;
  int32_t off, n;
  int axis;
  if (!dc2_walk(m, level, blockIdx.x, off, n, axis) || n <= VSM_DC_BLOCK_POINTS) return;  // no such node / it is a block
  const uint32_t idx = (1u << level) | blockIdx.x;
  const int t = threadIdx.x;
  const int32_t div = n >> 1;
  const VsmDcHull l = jb.hulls[2 * idx], r = jb.hulls[2 * idx + 1];
  const int32_t tbase = 2 * off, nslots = 2 * n, tbase4 = 4 * tbase;
  const int bwords = (nslots + 31) >> 5, pwords = (n + 31) >> 5;
  int32_t *gnode = jb.tri + (size_t)tbase * 8;
  const dc2_v4i *grec = (const dc2_v4i *)gnode;
  const uint32_t *gpt = jb.pt + off;
  DC2_T(m0);
  bool band_ok = bwords <= words_cap;
  if (t == 0) {
    s_cl = -1;
    s_cr = 1 << 30;
    s_fail = 0;
  }
  if (band_ok) {
    for (int i = t; i < bwords; i += DC2_MERGE_THREADS) {
      s_band[i] = 0;
      s_ring[i] = 0;
    }
    for (int i = t; i < pwords; i += DC2_MERGE_THREADS) s_pmask[i] = 0;
    __syncthreads();
    {  // extent of the two halves along the cut axis: largest coordinate on the left, smallest on the right
      // (DC2_BAND_BATCH independent loads in flight per thread: a thread's loads used to wait for each other, 29 round
      // trips for the 7.4 k points of a top node)
      int32_t cl = -1, cr = 1 << 30;
      for (int i0 = t; i0 < n; i0 += DC2_BAND_BATCH * DC2_MERGE_THREADS) {
        uint32_t p[DC2_BAND_BATCH];
#pragma unroll
        for (int k = 0; k < DC2_BAND_BATCH; k++) {
          const int i = i0 + k * DC2_MERGE_THREADS;
          p[k] = i < n ? gpt[i] : 0u;
        }
#pragma unroll
        for (int k = 0; k < DC2_BAND_BATCH; k++) {
          const int i = i0 + k * DC2_MERGE_THREADS;
          const int32_t c = axis == 0 ? (int32_t)(p[k] & 0xffffu) : (int32_t)(p[k] >> 16);
          if (i < div)
            cl = max(cl, c);
          else if (i < n)
            cr = min(cr, c);
        }
      }
#pragma unroll
      for (int o = 32; o >= 1; o >>= 1) {
        cl = max(cl, __shfl_xor(cl, o, 64));
        cr = min(cr, __shfl_xor(cr, o, 64));
      }
      if ((t & 63) == 0) {
        atomicMax(&s_cl, cl);
        atomicMin(&s_cr, cr);
      }
    }
    __syncthreads();
    const int32_t cl = s_cl, cr = s_cr;
    // the band's core: own slots, hull triangles, triangles whose circumcircle crosses the cut - one reach word per slot
    // (above; whoever wrote the record left it there), DC2_BAND_BATCH independent loads in flight per thread
    {
      const int32_t *gw = gnode + (axis == 0 ? 7 : 3);
      for (int s0 = t; s0 < nslots; s0 += DC2_BAND_BATCH * DC2_MERGE_THREADS) {
        uint32_t w[DC2_BAND_BATCH];
#pragma unroll
        for (int k = 0; k < DC2_BAND_BATCH; k++) {
          const int s = s0 + k * DC2_MERGE_THREADS;
          w[k] = s < nslots ? (uint32_t)gw[(size_t)s * 8] : DC2_REACH_NONE;
        }
#pragma unroll
        for (int k = 0; k < DC2_BAND_BATCH; k++) {
          const int s = s0 + k * DC2_MERGE_THREADS;
          const bool hot = s < nslots && (s == 2 * div - 2 || s == 2 * div - 1 || dc2_reaches(w[k], s < 2 * div, cl, cr));
          if (hot) atomicOr(&s_band[s >> 5], 1u << (s & 31));
        }
      }
    }
    __syncthreads();
    // ... and the neighbours of the core.  (The loops below used to test one slot per thread and round and load its record
    // where the bit was set: some lane of a wave always had one, so every one of the 58 rounds of a top node waited for a
    // round trip of its own.  Now: DC2_BAND_BATCH slots per round with their loads requested together - a cold lane reads
    // the node's own first new slot, one line for all of them -, and once the band is numbered the passes run over its
    // lines, two or three rounds.)
    const int s_own = 2 * div - 2;
    for (int s0 = t; s0 < nslots; s0 += DC2_BAND_BATCH * DC2_MERGE_THREADS) {
      dc2_v4i e[DC2_BAND_BATCH];
      bool hot[DC2_BAND_BATCH];
#pragma unroll
      for (int k = 0; k < DC2_BAND_BATCH; k++) {
        const int s = s0 + k * DC2_MERGE_THREADS;
        hot[k] = s < nslots && dc2_bit(s_band, s);
        e[k] = grec[2 * (hot[k] ? s : s_own)];
      }
#pragma unroll
      for (int k = 0; k < DC2_BAND_BATCH; k++) {
        if (!hot[k]) continue;
        const int32_t nb[3] = {e[k].x, e[k].y, e[k].z};
#pragma unroll
        for (int o = 0; o < 3; o++) {
          if (nb[o] < 0) continue;
          const int s2 = (nb[o] - tbase4) >> 2;
          if (s2 >= 0 && s2 < nslots) atomicOr(&s_ring[s2 >> 5], 1u << (s2 & 31));
        }
      }
    }
    __syncthreads();
    for (int i = t; i < bwords; i += DC2_MERGE_THREADS) s_band[i] |= s_ring[i];
    __syncthreads();
    const uint32_t nlines = dc2_mask_prefix(s_band, s_bpre, bwords, s_tot);
    uint32_t npts = 0;
    band_ok = nlines + 1 <= (uint32_t)lines_cap;  // (+ the trap record)
    if (band_ok) {
      // line -> slot (LDS only)
      for (int s = t; s < nslots; s += DC2_MERGE_THREADS)
        if (dc2_bit(s_band, s)) s_l2s[dc2_rank(s_band, s_bpre, s)] = (uint32_t)s;
      __syncthreads();
      // the points the band's records use
      for (uint32_t k0 = t; k0 < nlines; k0 += 3 * DC2_MERGE_THREADS) {
        dc2_v4i v[3];
#pragma unroll
        for (int k = 0; k < 3; k++) {
          const uint32_t ln = k0 + k * DC2_MERGE_THREADS;
          v[k] = grec[2 * (ln < nlines ? (int)s_l2s[ln] : s_own) + 1];
        }
#pragma unroll
        for (int k = 0; k < 3; k++) {
          if (k0 + k * DC2_MERGE_THREADS >= nlines) continue;
          const int32_t vv[3] = {v[k].x, v[k].y, v[k].z};
#pragma unroll
          for (int o = 0; o < 3; o++)
            if (vv[o] >= 0) atomicOr(&s_pmask[(vv[o] - off) >> 5], 1u << ((vv[o] - off) & 31));
        }
      }
      __syncthreads();
      npts = dc2_mask_prefix(s_pmask, s_ppre, pwords, s_tot);
      band_ok = npts <= (uint32_t)pts_cap;
      if (band_ok) {
        for (int i0 = t; i0 < n; i0 += DC2_BAND_BATCH * DC2_MERGE_THREADS) {
          uint32_t p[DC2_BAND_BATCH];
#pragma unroll
          for (int k = 0; k < DC2_BAND_BATCH; k++) {
            const int i = i0 + k * DC2_MERGE_THREADS;
            p[k] = gpt[i < n ? i : 0];
          }
#pragma unroll
          for (int k = 0; k < DC2_BAND_BATCH; k++) {
            const int i = i0 + k * DC2_MERGE_THREADS;
            if (i < n && dc2_bit(s_pmask, i)) {
              const int c = dc2_rank(s_pmask, s_ppre, i);
              s_pt[c] = p[k];
              s_c2p[c] = (uint32_t)i;
            }
          }
        }
        for (uint32_t k0 = t; k0 < nlines; k0 += 3 * DC2_MERGE_THREADS) {
          dc2_v4i e3[3], v3[3];
#pragma unroll
          for (int k = 0; k < 3; k++) {
            const uint32_t ln = k0 + k * DC2_MERGE_THREADS;
            const int s = ln < nlines ? (int)s_l2s[ln] : s_own;
            e3[k] = grec[2 * s];
            v3[k] = grec[2 * s + 1];
          }
#pragma unroll
          for (int k = 0; k < 3; k++) {
            const uint32_t line = k0 + k * DC2_MERGE_THREADS;
            if (line >= nlines) continue;
            const dc2_v4i e = e3[k], v = v3[k];
            auto nbw = [&](int32_t g) -> uint32_t {  // neighbour handle under the band's numbering; none / not in the band: the trap
              if (g < 0) return 4 * nlines;
              const int s2 = (g - tbase4) >> 2;
              if (s2 < 0 || s2 >= nslots || !dc2_bit(s_band, s2)) return 4 * nlines;
              return (uint32_t)(dc2_rank(s_band, s_bpre, s2) * 4 + (g & 3));
            };
            auto vxw = [&](int32_t g) -> uint32_t { return g < 0 ? 0xffffu : (uint32_t)dc2_rank(s_pmask, s_ppre, g - off); };
            dc2_v4u o;  // edge words: neighbour | apex << 16 per edge
            o.x = nbw(e.x) | (vxw(v.x) << 16);
            o.y = nbw(e.y) | (vxw(v.y) << 16);
            o.z = nbw(e.z) | (vxw(v.z) << 16);
            o.w = 0xffffffffu;
            ((dc2_v4u *)s_rec)[line] = o;
          }
        }
      }
    }
    __syncthreads();
    DC2_T(m1);
    if (band_ok) {
      if (t == 0) {
        DcBandMesh mesh;
        mesh.w = (DC2_AS3 dc2_u32a *)s_rec;
        mesh.pt = (DC2_AS3 const uint32_t *)s_pt;
        mesh.rim = 4 * (int32_t)nlines;
        mesh.budget = 8 * (int32_t)nlines + 64;
        mesh.make_trap();
        auto ln = [&](int32_t gt) -> int32_t {
          if (!dc2_bit(s_band, gt - tbase)) mesh.tripped = 1;  // (a hull handle is a hull triangle: in the band)
          return dc2_rank(s_band, s_bpre, gt - tbase);
        };
        DcOTri fl{ln(l.fl_t), l.fl_o}, il{ln(l.fr_t), l.fr_o}, ir{ln(r.fl_t), r.fl_o}, fr{ln(r.fr_t), r.fr_o};
        int32_t tcur = dc2_rank(s_band, s_bpre, 2 * div - 2);
        if (mesh.ok()) dc_merge(mesh, fl, il, ir, fr, axis, tcur);
        if (mesh.ok() && !mesh.trap_touched())
          jb.hulls[idx] = VsmDcHull{(int32_t)s_l2s[fl.t] + tbase, fl.o, (int32_t)s_l2s[fr.t] + tbase, fr.o};
        else
          s_fail = 1;
      }
      __syncthreads();
      band_ok = s_fail == 0;
#ifdef DC2_PHASE_TIMING
      if (t == 0) {
        const long long m2 = clock64();
        atomicAdd(&dc2_dbg[1 + level][0], 1ull);
        DC2_ACC(1 + level, 1, m0, m1);
        DC2_ACC(1 + level, 2, m1, m2);
        DC2_BAND_STAT(level, 3, nlines);
        DC2_BAND_STAT(level, 4, band_ok ? 0 : 1);
        atomicMax(&dc2_dbg[1 + level][6], (unsigned long long)nlines);
        atomicMax(&dc2_dbg[1 + level][7], (unsigned long long)npts);
      }
#endif
      if (band_ok) {  // the band's records back under global numbering (a record is two 16-byte stores)
        for (uint32_t k = t; k < nlines; k += DC2_MERGE_THREADS) {
          const dc2_v4u o = ((const dc2_v4u *)s_rec)[k];
          auto nb = [&](uint32_t v) -> int32_t { return v == 0xffffu ? -1 : (int32_t)(s_l2s[v >> 2] * 4 + (v & 3)) + tbase4; };
          auto vx = [&](uint32_t v) -> int32_t { return v == 0xffffu ? -1 : (int32_t)s_c2p[v] + off; };
          // (a word that still points to the trap was never touched: the record in global memory has the neighbour, or none)
          const uint32_t rimh = 4 * nlines;
          const int32_t s = (int32_t)s_l2s[k];
          const dc2_v4i old = grec[2 * s];
          dc2_v4i a, b;
          a.x = (o.x & 0xffffu) == rimh ? old.x : nb(o.x & 0xffffu);
          a.y = (o.y & 0xffffu) == rimh ? old.y : nb(o.y & 0xffffu);
          a.z = (o.z & 0xffffu) == rimh ? old.z : nb(o.z & 0xffffu);
          b.x = vx(o.x >> 16);
          b.y = vx(o.y >> 16);
          b.z = vx(o.z >> 16);
          uint32_t wx, wy;  // reach words of the record as it is now
          if ((b.x & b.y & b.z) < 0)
            wx = wy = DC2_REACH_NONE;
          else if ((b.x | b.y | b.z) < 0)
            wx = wy = DC2_REACH_ALL;
          else
            dc2_reach(s_pt[o.x >> 16], s_pt[o.y >> 16], s_pt[o.z >> 16], wx, wy);
          a.w = (int32_t)wy;
          b.w = (int32_t)wx;
          ((dc2_v4i *)gnode)[2 * s] = a;
          ((dc2_v4i *)gnode)[2 * s + 1] = b;
        }
        return;
      }
    }
  }
  // the band did not fit, or the walk left it: one lane on the plain mesh in global memory
  if (t == 0) {
    DcOTri fl{l.fl_t, l.fl_o}, il{l.fr_t, l.fr_o}, ir{r.fl_t, r.fl_o}, fr{r.fr_t, r.fr_o};
    int32_t tg = 2 * (off + div) - 2;
    const DcMesh mesh{jb.tri, jb.pt, jb.id, jb.key};
    mesh.merge_hulls(fl, il, ir, fr, axis, tg);
    jb.hulls[idx] = VsmDcHull{fl.t, fl.o, fr.t, fr.o};
#ifdef DC2_PHASE_TIMING
    atomicAdd(&dc2_dbg[1 + level][5], 1ull);
#endif
  }
  __threadfence();
  __syncthreads();
  // the plain mesh knows nothing of reach words: every record of the node gets them afresh (rare path)
  for (int sl = t; sl < nslots; sl += DC2_MERGE_THREADS) {
    // (lane 0 has just rewritten these records: loads that go to L2, not to a line this compute unit may still hold)
    const int32_t vx0 = __hip_atomic_load(gnode + (size_t)sl * 8 + 4, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const int32_t vx1 = __hip_atomic_load(gnode + (size_t)sl * 8 + 5, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const int32_t vx2 = __hip_atomic_load(gnode + (size_t)sl * 8 + 6, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    uint32_t wx, wy;
    if ((vx0 & vx1 & vx2) < 0)
      wx = wy = DC2_REACH_NONE;
    else if ((vx0 | vx1 | vx2) < 0)
      wx = wy = DC2_REACH_ALL;
    else
      dc2_reach(jb.pt[vx0], jb.pt[vx1], jb.pt[vx2], wx, wy);
    gnode[(size_t)sl * 8 + 3] = (int32_t)wy;
    gnode[(size_t)sl * 8 + 7] = (int32_t)wx;
  }
}

// Triangle's vertex sort for the jobs of a chunk on the device (one wave each): the verdicts go to
// tie_out + job * out_stride; lists the wave cannot take get -1 there
template <int CAP>
__global__ void __launch_bounds__(64) k_dc2_ties(const VsmDc2Job *__restrict__ jobs, int32_t *__restrict__ tie_out, int out_stride, const uint32_t *__restrict__ tiny) {
  const VsmDc2Job jb = jobs[blockIdx.x];
  // (tie_out == nullptr: a job table gathered from several chunks' slabs - every job says itself where its verdict goes)
  int32_t *out = tie_out ? tie_out + (size_t)blockIdx.x * out_stride : const_cast<int32_t *>(jb.tie_out);
  const int n = *jb.count;
  if (n <= 3 || n > jb.cap) {
    if (threadIdx.x == 0) out[0] = n <= 3 ? 0 : -1;
    return;
  }
  tie_sort<CAP>(TieFromKeys{jb.keys_in}, n, out, tiny);
}

// which match stands for a pixel that several share: remap[index carried] = index Triangle's sort puts first
__global__ void __launch_bounds__(64) k_dc2_apply_ties(const VsmDc2Job *__restrict__ jobs) {
  const VsmDc2Job jb = jobs[blockIdx.x];
  const int32_t c = jb.tie_out[0];
  if (c < 0) {
    if (threadIdx.x == 0 && jb.mn[0] >= 2) *jb.error = 3;
    return;
  }
  for (int k = threadIdx.x; k < c; k += 64) {
    const int32_t rep = jb.tie_out[1 + 2 * k], first = jb.tie_out[2 + 2 * k];
    if (rep >= 0 && rep < jb.cap) jb.remap[rep] = first;
  }
}

// per match what the support test compares (vsm_host_outliers_begin): flow u, flow v, disparity
__global__ void __launch_bounds__(256) k_dc2_flows(const VsmDc2Job *__restrict__ jobs, int method) {
  const VsmDc2Job jb = jobs[blockIdx.y];
  const int n = jb.mn[1];
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n || jb.mn[0] < 2) return;
  const vsm_p_match a = jb.list[i];
  float *fu = (float *)jb.kd_scratch, *fv = fu + jb.kd_stride, *dp = fv + jb.kd_stride;
  fu[i] = a.u1c - a.u1p;
  fv[i] = a.v1c - a.v1p;
  dp[i] = method == 1 ? a.u1c - a.u2c : a.u1p - a.u2p;
}

__global__ void __launch_bounds__(256) k_dc2_support(const VsmDc2Job *__restrict__ jobs, int method, float ftol, float dtol) {
  const VsmDc2Job jb = jobs[blockIdx.y];
  const int32_t m = jb.mn[0];
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (m < 2 || t >= 2 * m) return;
  const int4 v = *(const int4 *)(jb.tri + (size_t)t * 8 + 4);
  if ((v.x | v.y | v.z) < 0) return;
  const int32_t q[3] = {jb.remap[jb.id[v.y]], jb.remap[jb.id[v.z]], jb.remap[jb.id[v.x]]};
  const float *fua = (const float *)jb.kd_scratch, *fva = fua + jb.kd_stride, *dpa = fva + jb.kd_stride;
  float fu[3], fv[3], dp[3];
#pragma unroll
  for (int k = 0; k < 3; k++) {
    fu[k] = fua[q[k]];
    fv[k] = fva[q[k]];
    dp[k] = dpa[q[k]];
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

// survivors (support >= 4, viso/matcher.cpp:1369-1371; lists of up to three matches unchanged, :1210) in list order
__device__ inline bool dc2_keep(const VsmDc2Job &jb, int32_t n, int32_t i) { return n <= 3 || jb.support[i] >= 4; }

__global__ void __launch_bounds__(KD_THREADS) k_dc2_compact(const VsmDc2Job *__restrict__ jobs) {
  __shared__ uint32_t tot[KD_THREADS / 64 + 1];
  __shared__ int32_t s_total;
  const VsmDc2Job jb = jobs[blockIdx.x];
  const int t = threadIdx.x;
  const int32_t n = jb.mn[1];
  const int32_t per = (n + KD_THREADS - 1) / KD_THREADS;
  const int32_t i0 = min(n, t * per), i1 = min(n, i0 + per);
  uint32_t cnt = 0;
  for (int32_t i = i0; i < i1; i++) cnt += dc2_keep(jb, n, i) ? 1u : 0u;
  uint32_t pos = kd_block_scan(cnt, tot);
  for (int32_t i = i0; i < i1; i++) jb.remap[i] = dc2_keep(jb, n, i) ? (int32_t)pos++ : -1;  // (the tie patches have been used)
  if (t == KD_THREADS - 1) s_total = (int32_t)pos;
  __syncthreads();
  if (t == 0) {
    *jb.out_count = s_total;
    if (jb.h_out_count) *jb.h_out_count = s_total;
  }
  if (!jb.out) return;
  if (jb.h_keep) {  // the records are on the host already: one bit per match says which of them stay
    for (int32_t w = t; w < (n + 31) / 32; w += KD_THREADS) {
      uint32_t bits = 0;
      for (int b = 0; b < 32 && 32 * w + b < n; b++) bits |= jb.remap[32 * w + b] >= 0 ? (1u << b) : 0u;
      jb.h_keep[w] = bits;
    }
    return;
  }
  const uint4 *src = (const uint4 *)jb.list;
  uint4 *dst = (uint4 *)jb.out;
  for (int32_t p = t; p < 3 * n; p += KD_THREADS) {
    const int32_t e = p / 3, d = jb.remap[e];
    if (d >= 0) dst[3 * d + (p - 3 * e)] = src[p];
  }
}

// M4 computePriorStatistics (viso/matcher.cpp:734-868; host form: vsm_host_prior_statistics) over the survivors of
// the pass-1 list: per 3x3-dilated bin the min / max of the per-stage deltas, widened to at least 20 pixels.  All
// operands are integer-valued floats (pass-1 matches are unrefined), so the minima and maxima are kept as integers
// (LDS atomics) and the few float operations give the host's values exactly.  Output in the device layout of the
// match kernels: per bin {u_min, u_max, v_min, v_max} of stage 0, then stage 1, ...
#define DC2_PRIOR_MAX_BINS 1024
__global__ void __launch_bounds__(256) k_dc2_prior(const VsmDc2Job *__restrict__ jobs, int method, int binsize, int radius, int w, int h,
                                                   int ub, int vb) {
  __shared__ int32_t s_lo[DC2_PRIOR_MAX_BINS * 8], s_hi[DC2_PRIOR_MAX_BINS * 8];
  const VsmDc2Job jb = jobs[blockIdx.x];
  const int nb = ub * vb, ns = method == 2 ? 4 : 2;
  const int t = threadIdx.x;
  const int32_t n = jb.mn[1];
  for (int i = t; i < nb * 8; i += 256) {
    s_lo[i] = 1000000;
    s_hi[i] = -1000000;
  }
  __syncthreads();
  const float bs = (float)binsize;
  for (int32_t i = t; i < n; i += 256) {
    if (!dc2_keep(jb, n, i)) continue;
    const vsm_p_match it = jb.list[i];
    int32_t d[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    float ur = it.u1c, vr = it.v1c;
    if (method == 0) {
      d[0] = (int32_t)(it.u1p - it.u1c);
      d[1] = (int32_t)(it.v1p - it.v1c);
      d[2] = (int32_t)(it.u1c - it.u1p);
      d[3] = (int32_t)(it.v1c - it.v1p);
    } else if (method == 1) {
      d[0] = (int32_t)(it.u2c - it.u1c);
      d[2] = (int32_t)(it.u1c - it.u2c);
    } else {
      d[0] = (int32_t)(it.u2p - it.u1p);
      d[2] = (int32_t)(it.u2c - it.u2p);
      d[3] = (int32_t)(it.v2c - it.v2p);
      d[4] = (int32_t)(it.u1c - it.u2c);
      d[6] = (int32_t)(it.u1p - it.u1c);
      d[7] = (int32_t)(it.v1p - it.v1c);
      ur = it.u1p;
      vr = it.v1p;
    }
    const int ubin = (int)floorf(ur / bs), vbin = (int)floorf(vr / bs);
    const int u0 = min(max(ubin - 1, 0), ub - 1), u1 = min(max(ubin + 1, 0), ub - 1);
    const int v0 = min(max(vbin - 1, 0), vb - 1), v1 = min(max(vbin + 1, 0), vb - 1);
    for (int v = v0; v <= v1; v++)
      for (int u = u0; u <= u1; u++) {
        const int b = v * ub + u;
        for (int k = 0; k < 2 * ns; k++) {
          atomicMin(&s_lo[b * 8 + k], d[k]);
          atomicMax(&s_hi[b * 8 + k], d[k]);
        }
      }
  }
  __syncthreads();
  for (int b = t; b < nb; b += 256) {
    const bool any = s_lo[b * 8] != 1000000;
    float *r = jb.ranges + (size_t)b * 16;
    for (int i = 0; i < 4; i++) {
      float l[2] = {0.f, 0.f}, hh[2] = {0.f, 0.f};
      if (i < ns)
        for (int k = 0; k < 2; k++) {
          l[k] = any ? (float)s_lo[b * 8 + 2 * i + k] : (float)(-radius);
          hh[k] = any ? (float)s_hi[b * 8 + 2 * i + k] : (float)(+radius);
          const float delta = hh[k] - l[k];
          if (delta < 20) {  // widen to at least 20 px (:845-854)
            const float g = ceilf((20 - delta) / 2);
            l[k] -= g;
            hh[k] += g;
          }
        }
      *(float4 *)(r + 4 * i) = make_float4(l[0], hh[0], l[1], hh[1]);
    }
  }
}

void vsm_dc2_launch_keys(hipStream_t s, const VsmDc2Job *d_jobs, int njobs, int max_list) {
  if (njobs <= 0) return;
  hipLaunchKernelGGL(k_dc2_keys, dim3((std::max(max_list, 1) + 255) / 256, njobs), dim3(256), 0, s, d_jobs);
}
void vsm_dc2_launch_prepare(hipStream_t s, const VsmDc2Job *d_jobs, int njobs, int max_list, bool expect_long) {
  if (njobs <= 0) return;
#ifndef DC2_PREP_OLD
  if (max_list > 0) {  // (0: the caller does not know how long the lists are)
    static const bool big8 = hipFuncSetAttribute((const void *)k_dc2_prepare_lds<8>, hipFuncAttributeMaxDynamicSharedMemorySize, 16 * DC2_PREP_LDS) == hipSuccess;
    static const bool big4 = hipFuncSetAttribute((const void *)k_dc2_prepare_lds<4>, hipFuncAttributeMaxDynamicSharedMemorySize, 16 * DC2_PREP_LDS / 2) == hipSuccess;
    int cap2 = 256;
    while (cap2 < max_list && cap2 < DC2_PREP_LDS) cap2 <<= 1;
    static_assert(DC2_PREP_LDS / 8 == KD_THREADS && 16 * DC2_PREP_LDS >= 4 * KD_DIGITS * KD_CHUNKS, "the long lists of a launch run k_dc2_prepare's body on this kernel's threads and LDS");
    if (cap2 == DC2_PREP_LDS && big8) {  // (max_list counts queries: most lists of such a launch are shorter than 8192 matches, whoever is not takes the first form)
      // (the second launch costs the chain 26 us in the pipeline - 110 workgroups of sixteen waves that come and go - whether a
      // list is long or not: only where the counts say that long lists are the rule, or one has been seen)
      const bool with_long = max_list > cap2 && (expect_long || max_list > 2 * cap2);
      hipLaunchKernelGGL(k_dc2_prepare_lds<8>, dim3(njobs), dim3(cap2 / 8), (size_t)16 * cap2, s, d_jobs, cap2, with_long ? 2 : (max_list > cap2 ? 1 : 0));
      if (with_long) hipLaunchKernelGGL(k_dc2_prepare_long, dim3(njobs), dim3(KD_THREADS), 0, s, d_jobs, cap2);
      return;
    }
    if (cap2 < DC2_PREP_LDS && (big4 || 16 * cap2 <= 64 * 1024)) {
      hipLaunchKernelGGL(k_dc2_prepare_lds<4>, dim3(njobs), dim3(cap2 / 4), (size_t)14 * cap2 + 8 * (cap2 / 4), s, d_jobs, cap2, 0);
      return;
    }
  }
#endif
  hipLaunchKernelGGL(k_dc2_prepare, dim3(njobs), dim3(KD_THREADS), 0, s, d_jobs);
}
void vsm_dc2_launch_blocks(hipStream_t s, const VsmDc2Job *d_jobs, int njobs, int depth) {
  if (njobs <= 0) return;
  hipLaunchKernelGGL(k_dc2_block, dim3(1 << depth, njobs), dim3(DC2_BLOCK_THREADS), 0, s, d_jobs, depth);
}
#ifndef DC2_BAND_F
#define DC2_BAND_F 12  // LDS lines of a merge node's band: 256 + DC2_BAND_F * sqrt(points of the node); measured need: 256 + 3.2 sqrt (535 lines at the top of a 7.4 k list); 24 cost the top level 66 KB of LDS per workgroup, which waited for room beside the block kernel's
#endif
static std::atomic<int> g_dc2_band_factor{DC2_BAND_F};
// test hook: bands of 256 + f * sqrt(points) LDS lines (f < 0: the default) - with a small f the large merge nodes overflow
// their band and are redone by one lane in global memory, the path that no list of the benchmark takes
extern "C" void vsm_debug_dc2_band_factor(int32_t f) { g_dc2_band_factor.store(f < 0 ? DC2_BAND_F : f); }
void vsm_dc2_launch_merges(hipStream_t s, const VsmDc2Job *d_jobs, int njobs, int depth, int max_list) {
  if (njobs <= 0) return;
  const double band_f = (double)g_dc2_band_factor.load(std::memory_order_relaxed);
  static const bool big_lds = hipFuncSetAttribute((const void *)k_dc2_merge, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 256) == hipSuccess;
  for (int level = depth - 1; level >= 0; level--) {
    // the largest node of the level: ceil(max_list / 2^level) points.  A band is a few records per pixel of cut plus the
    // two hulls, i.e. it grows like the square root of the node: 256 + 24 sqrt(n) lines cover it several times over
    // (measured: 535 lines for the top merge of 7.4 k points), the points its records use are fewer than the lines.
    const int nmax = ((max_list + (1 << level) - 1) >> level) + 1;
    int lines_cap = (256 + (int)(band_f * std::sqrt((double)nmax)) + 63) & ~63;
    lines_cap = std::min(lines_cap, 16384);  // (16-bit line handles: line * 4 + edge, the trap record included, below 0xffff)
    int pts_cap = lines_cap;
    int words_cap = ((2 * nmax + 31) >> 5) + 1;
    auto bytes_of = [&]() {
      return (size_t)lines_cap * 16 + (size_t)pts_cap * 4 + (size_t)lines_cap * 4 + (size_t)pts_cap * 4 + (size_t)words_cap * 4 * 2 +
             (size_t)((words_cap + 1) / 2) * 4 + (size_t)words_cap * 2 * 2 + 64;
    };
    const size_t limit = big_lds ? 160 * 1024 - 1024 : 64 * 1024 - 1024;
    while (bytes_of() > limit && lines_cap > 256) {  // (lists of several ten thousand points: what does not fit goes the slow way)
      lines_cap -= 256;
      pts_cap = lines_cap;
    }
    if (bytes_of() > limit) words_cap = 0;
    hipLaunchKernelGGL(k_dc2_merge, dim3(1 << level, njobs), dim3(DC2_MERGE_THREADS), bytes_of(), s, d_jobs, level, lines_cap, pts_cap, words_cap);
  }
}
int vsm_host_tiny_table(uint32_t *out, int cap);  // (vsm_host.cpp)
// the tiny-partition table on the current device (uploaded once per device and process; nullptr = without it)
static const uint32_t *dc2_tiny_table() {
  static std::mutex mu;
  static const uint32_t *tab[64] = {};
  static bool tried[64] = {};
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return nullptr;
  std::lock_guard<std::mutex> lk(mu);
  if (!tried[dev]) {
    tried[dev] = true;
    std::vector<uint32_t> h(TIE_TINY_WORDS);
    uint32_t *d = nullptr;
    if (vsm_host_tiny_table(h.data(), TIE_TINY_WORDS) == TIE_TINY_WORDS && hipMalloc((void **)&d, TIE_TINY_WORDS * 4) == hipSuccess) {
      if (hipMemcpy(d, h.data(), TIE_TINY_WORDS * 4, hipMemcpyHostToDevice) == hipSuccess) {
        tab[dev] = d;
      } else {
        (void)hipFree(d);
      }
    }
    (void)hipGetLastError();
  }
  return tab[dev];
}
void vsm_dc2_launch_ties(hipStream_t s, const VsmDc2Job *d_jobs, int njobs, int32_t *tie_out, int out_stride, int max_list) {
  if (njobs <= 0) return;
  const uint32_t *tiny = dc2_tiny_table();
  // (the first-pass lists are a tenth of the second-pass ones: 25 KB of LDS per list instead of 141 - a list that turns out
  // longer than the bound it was launched with reports -1 like one beyond VSM_DC_TIE_POINTS)
  if (max_list > 0 && max_list <= 2048)
    hipLaunchKernelGGL(k_dc2_ties<2048>, dim3(njobs), dim3(64), 0, s, d_jobs, tie_out, out_stride, tiny);
  else
    hipLaunchKernelGGL(k_dc2_ties<VSM_DC_TIE_POINTS>, dim3(njobs), dim3(64), 0, s, d_jobs, tie_out, out_stride, tiny);
}
#define DC2_SUPPORT_LDS 12288
// The end of a chain in ONE kernel for lists of up to DC2_SUPPORT_LDS matches (k_dc2_apply_ties + k_dc2_support_lds +
// k_dc2_compact): the tie patches go into a 16-bit remap table in LDS, the votes count in LDS, the survivors' places come
// from a scan over the counts where they lie - three launches and two trips through global memory less at the tail of
// every chain, and the votes take four slots per thread and round (each is a chain of dependent loads: record -> input
// index -> flow).  One workgroup per list.
__global__ void __launch_bounds__(1024) k_dc2_finish(const VsmDc2Job *__restrict__ jobs, int method, float ftol, float dtol) {
  __shared__ int32_t s_sup[DC2_SUPPORT_LDS];
  __shared__ uint16_t s_map[DC2_SUPPORT_LDS];  // remap (tie patches), later: the survivor's place, 0xffff = dropped
  __shared__ uint32_t tot[33];
  __shared__ int32_t s_total;
  const VsmDc2Job jb = jobs[blockIdx.x];
  const int32_t m = jb.mn[0], n = jb.mn[1];
  const int t = threadIdx.x;
  for (int i = t; i < n; i += 1024) {
    s_sup[i] = 0;
    s_map[i] = (uint16_t)i;
  }
  __syncthreads();
  if (m >= 2) {
    const int32_t c = jb.tie_out[0];
    if (c < 0) {  // (the vertex sort could not take the list)
      if (t == 0) *jb.error = 3;
    } else {
      for (int k = t; k < c; k += 1024) {
        const int32_t rep = jb.tie_out[1 + 2 * k], first = jb.tie_out[2 + 2 * k];
        if (rep >= 0 && rep < n && first >= 0 && first < n) s_map[rep] = (uint16_t)first;
      }
    }
    __syncthreads();
    const float *fua = (const float *)jb.kd_scratch, *fva = fua + jb.kd_stride, *dpa = fva + jb.kd_stride;
    for (int s0 = t; s0 < 2 * m; s0 += 4 * 1024) {
      int4 v[4];
      int32_t id[4][3];
      bool live[4];
#pragma unroll
      for (int k = 0; k < 4; k++) {
        const int s = s0 + k * 1024;
        v[k] = s < 2 * m ? *(const int4 *)(jb.tri + (size_t)s * 8 + 4) : int4{-1, -1, -1, -1};
        live[k] = (v[k].x | v[k].y | v[k].z) >= 0;
      }
#pragma unroll
      for (int k = 0; k < 4; k++) {
        id[k][0] = live[k] ? jb.id[v[k].y] : 0;
        id[k][1] = live[k] ? jb.id[v[k].z] : 0;
        id[k][2] = live[k] ? jb.id[v[k].x] : 0;
      }
      int32_t q[4][3];
      float fu[4][3], fv[4][3], dp[4][3];
#pragma unroll
      for (int k = 0; k < 4; k++)
#pragma unroll
        for (int j = 0; j < 3; j++) {
          q[k][j] = live[k] ? (int32_t)s_map[min(max(id[k][j], 0), n - 1)] : 0;
          fu[k][j] = fua[q[k][j]];
          fv[k][j] = fva[q[k][j]];
          dp[k][j] = dpa[q[k][j]];
        }
#pragma unroll
      for (int k = 0; k < 4; k++) {
        if (!live[k]) continue;
#pragma unroll
        for (int e = 0; e < 3; e++) {
          const int a = e == 2 ? 0 : e, b = e == 0 ? 1 : 2;  // (0,1) (1,2) (0,2)
          const bool flow_ok = fabsf(fu[k][a] - fu[k][b]) + fabsf(fv[k][a] - fv[k][b]) < ftol;
          const bool disp_ok = fabsf(dp[k][a] - dp[k][b]) < dtol;
          const bool ok = method == 0 ? flow_ok : (method == 1 ? disp_ok : (disp_ok && flow_ok));
          if (ok) {
            atomicAdd(&s_sup[q[k][a]], 1);
            atomicAdd(&s_sup[q[k][b]], 1);
          }
        }
      }
    }
    __syncthreads();
  }
  // survivors (support >= 4, viso/matcher.cpp:1369-1371; lists of up to three matches unchanged, :1210) in list order
  const int32_t per = (n + 1023) / 1024;
  const int32_t i0 = min(n, t * per), i1 = min(n, i0 + per);
  uint32_t cnt = 0;
  for (int32_t i = i0; i < i1; i++) {
    const int32_t sup = s_sup[i];
    jb.support[i] = sup;  // (k_dc2_prior looks at it)
    cnt += (n <= 3 || sup >= 4) ? 1u : 0u;
  }
  uint32_t pos = dc2_scan(cnt, tot, 16);
  for (int32_t i = i0; i < i1; i++) s_map[i] = (n <= 3 || s_sup[i] >= 4) ? (uint16_t)pos++ : (uint16_t)0xffffu;
  if (t == 1023) s_total = (int32_t)pos;
  __syncthreads();
  if (t == 0) {
    *jb.out_count = s_total;
    if (jb.h_out_count) *jb.h_out_count = s_total;
  }
  if (!jb.out) return;
  if (jb.h_keep) {  // the records are on the host already: one bit per match says which of them stay
    for (int32_t w = t; w < (n + 31) / 32; w += 1024) {
      uint32_t bits = 0;
      for (int b = 0; b < 32 && 32 * w + b < n; b++) bits |= s_map[32 * w + b] != 0xffffu ? (1u << b) : 0u;
      jb.h_keep[w] = bits;
    }
    return;
  }
  const uint4 *src = (const uint4 *)jb.list;
  uint4 *dst = (uint4 *)jb.out;
  for (int32_t p = t; p < 3 * n; p += 1024) {
    const int32_t e = p / 3;
    const uint32_t d = s_map[e];
    if (d != 0xffffu) dst[3 * d + (p - 3 * e)] = src[p];
  }
}

void vsm_dc2_launch_flows(hipStream_t s, const VsmDc2Job *d_jobs, int njobs, int max_list, int method) {
  if (njobs <= 0) return;
  hipLaunchKernelGGL(k_dc2_flows, dim3((std::max(max_list, 1) + 255) / 256, njobs), dim3(256), 0, s, d_jobs, method);
}
// tie patches, votes, survivors (k_dc2_flows has run)
void vsm_dc2_launch_votes(hipStream_t s, const VsmDc2Job *d_jobs, int njobs, int max_list, int method, float flow_tol, float disp_tol) {
  if (njobs <= 0) return;
  if (max_list <= DC2_SUPPORT_LDS && max_list < 0xffff) {
    hipLaunchKernelGGL(k_dc2_finish, dim3(njobs), dim3(1024), 0, s, d_jobs, method, flow_tol, disp_tol);
    return;
  }
  hipLaunchKernelGGL(k_dc2_apply_ties, dim3(njobs), dim3(64), 0, s, d_jobs);
  hipLaunchKernelGGL(k_dc2_support, dim3((2 * std::max(max_list, 1) + 255) / 256, njobs), dim3(256), 0, s, d_jobs, method, flow_tol, disp_tol);
  hipLaunchKernelGGL(k_dc2_compact, dim3(njobs), dim3(KD_THREADS), 0, s, d_jobs);
}
void vsm_dc2_launch_prior(hipStream_t s, const VsmDc2Job *d_jobs, int njobs, int method, int binsize, int radius, int w, int h, int ub,
                          int vb) {
  if (njobs <= 0) return;
  hipLaunchKernelGGL(k_dc2_prior, dim3(njobs), dim3(256), 0, s, d_jobs, method, binsize, radius, w, h, ub, vb);
}
