// Monocular egomotion on top of the matcher: VisualOdometryMono::process and what it calls
// (viso/viso_mono.cpp:33-431), SURVEY.md section 8 row f-4 -- the part of the reference that its
// OpenCL kernels accelerate (viso/viso_mono_cl.cpp, viso/kernels/plane_and_inliers.cl: Sampson
// inlier masks / counts for batches of 16 fundamental matrices, and the Gaussian ground-plane vote).
//
// The definition followed here is the reference's CPU class (double arithmetic): results equal it
// bit for bit, which the OpenCL variant (float arithmetic on an arbitrary device) never promised.
// Division of labour on MI355X + host:
//   * 8-point fundamental matrices for all RANSAC hypotheses (an 8x9 and a 3x3 SVD each):
//     k_mono_fit, one thread per hypothesis running the same vsm_linalg.h code as the host (f64 +, *,
//     /, sqrt are correctly rounded on gfx950, contraction is off; a self-test at context creation
//     compares 64 device fits with host fits bit for bit and falls back to the host pool otherwise);
//   * inlier counting, hypotheses x matches Sampson distances in double: k_mono_inlier_count, one
//     launch for ALL hypotheses straight from the device-resident F array (the reference needs
//     ransac_iters/16 launches with a host round trip each).  Only +, *, / and a compare: IEEE-exact
//     on the GPU, so the counts ARE the reference's;
//   * R|t disambiguation: 4 x matches 4x4 SVDs + chirality counts: k_mono_triangulate (same
//     condition as the fits), only the chosen candidate's points come back;
//   * the sequential pieces stay on the host: sampling, the winner's inlier list, F from all inliers
//     (a tall m x 9 SVD whose dot products must keep their order), E -> R|t, median;
//   * ground-plane vote, O(n^2) exponentials: k_mono_plane_vote computes every candidate's sum with
//     the device exp(), which may differ from the host libm in the last bit, so the GPU only
//     PROPOSES: candidates within 1e-9 (relative) of the best sum are re-evaluated on the host with
//     libm's exp() in the reference's summation order, and the first maximum among them wins.
// Without a GPU context (vsm_host_estimate_motion_mono) both inner loops run on the host threads.
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <atomic>
#include <mutex>
#include <vector>

#include "vsm_host.h"
#include "vsm_linalg.h"
#include "vsm_svd_coop.h"

// ---------------------------------------------------------------------------------------
// kernels
// ---------------------------------------------------------------------------------------
struct MonoPt {
  float u1p, v1p, u1c, v1c;
};

// getInlier (viso/viso_mono.cpp:296-344) for every (hypothesis, match): blockIdx.y = hypothesis,
// one match per lane; the nine doubles of F are wave-uniform.  counts[k] += inliers.
__global__ void __launch_bounds__(256)
    k_mono_inlier_count(const MonoPt *__restrict__ pts, int n, const double *__restrict__ Fs, double thr,
                        int32_t *__restrict__ counts) {
  const double *F = Fs + 9 * (size_t)blockIdx.y;
  const double f00 = F[0], f01 = F[1], f02 = F[2], f10 = F[3], f11 = F[4], f12 = F[5], f20 = F[6], f21 = F[7], f22 = F[8];
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  bool in = false;
  if (i < n) {
    const MonoPt p = pts[i];
    const double u1 = p.u1p, v1 = p.v1p, u2 = p.u1c, v2 = p.v1c;
    const double Fx1u = f00 * u1 + f01 * v1 + f02;
    const double Fx1v = f10 * u1 + f11 * v1 + f12;
    const double Fx1w = f20 * u1 + f21 * v1 + f22;
    const double Ftx2u = f00 * u2 + f10 * v2 + f20;
    const double Ftx2v = f01 * u2 + f11 * v2 + f21;
    const double x2tFx1 = u2 * Fx1u + v2 * Fx1v + Fx1w;
    const double d = x2tFx1 * x2tFx1 / (Fx1u * Fx1u + Fx1v * Fx1v + Ftx2u * Ftx2u + Ftx2v * Ftx2v);
    in = fabs(d) < thr;
  }
  const unsigned long long b = __ballot(in);
  if ((threadIdx.x & 63) == 0 && b) atomicAdd(&counts[blockIdx.y], (int32_t)__popcll(b));
}

// findBestPlane (viso/viso_mono.cpp:75-101): sum_j exp(-(d_j - d_i)^2 * weight) for the candidates
// d_i > threshold.  blockIdx.x = tile of 256 candidates, blockIdx.y = one of VOTE_SPLIT slices of the
// j range (a few thousand points would otherwise occupy a few dozen CUs only); every slice writes
// its partial sums to part[slice][i] and the host adds them up.  The sums are proposals only (see
// the header comment), so their summation order is free.
#define VOTE_SPLIT 16
__global__ void __launch_bounds__(256)
    k_mono_plane_vote(const double *__restrict__ d, int n, double threshold, double weight, double *__restrict__ part) {
  __shared__ double s_d[256];
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  const double di = i < n ? d[i] : 0.0;
  const bool active = i < n && di > threshold;
  const int per = (n + VOTE_SPLIT - 1) / VOTE_SPLIT;
  const int jlo = blockIdx.y * per, jhi = min(n, jlo + per);
  double sum = 0;
  for (int j0 = jlo; j0 < jhi; j0 += 256) {
    __syncthreads();
    s_d[threadIdx.x] = j0 + (int)threadIdx.x < jhi ? d[j0 + threadIdx.x] : 0.0;
    __syncthreads();
    const int lim = min(256, jhi - j0);
    if (active)
      for (int j = 0; j < lim; j++) {
        const double dist = s_d[j] - di;
        sum += exp(-dist * dist * weight);
      }
  }
  if (i < n) part[(size_t)blockIdx.y * n + i] = active ? sum : 0.0;
}

// fundamentalMatrix for 8 sampled matches (viso/viso_mono.cpp:264-294): one thread per hypothesis,
// the 8x9 constraint matrix, its SVD and the rank-2 projection in private memory.  vsm_linalg.h is the
// same code the host runs; f64 +, *, /, sqrt are correctly rounded on gfx950 and contraction is off,
// so the nine doubles are the host's (checked by a self-test when the context is created).
// 16 lanes per hypothesis (vsm_svd_coop.h): the matrices of a group live in LDS, the lanes share the
// independent column / row loops of the SVD, 16 hypotheses per 256-thread block.
#define FIT_GROUP_DOUBLES 192  // U 72 + V 81 + W 9 + RV 9, padded
__global__ void __launch_bounds__(256)
    k_mono_fit(const MonoPt *__restrict__ pts, const int32_t *__restrict__ picks, int K, double *__restrict__ Fs) {
  __shared__ double s_m[16 * FIT_GROUP_DOUBLES];
  const int grp = threadIdx.x >> 4, ln = threadIdx.x & 15;
  const int k = blockIdx.x * 16 + grp;
  if (k >= K) return;  // whole groups leave together
  volatile double *A = s_m + grp * FIT_GROUP_DOUBLES, *v = A + 72, *w = v + 81, *rv1 = w + 9;
  if (ln < 8) {  // lane i fills row i of the constraint matrix
    const MonoPt q = pts[picks[k * 8 + ln]];
    volatile double *r = A + ln * 9;
    r[0] = q.u1c * q.u1p;
    r[1] = q.u1c * q.v1p;
    r[2] = q.u1c;
    r[3] = q.v1c * q.u1p;
    r[4] = q.v1c * q.v1p;
    r[5] = q.v1c;
    r[6] = q.u1p;
    r[7] = q.v1p;
    r[8] = 1;
  }
  VSM_GROUP_SYNC();
  vsm_la::svd_group<8, 9>(A, v, w, rv1, ln);
  // rank 2: the 3x3 built from the last column of V goes through the same group SVD
  double f0 = 0;
  if (ln < 9) f0 = v[ln * 9 + 8];
  VSM_GROUP_SYNC();
  volatile double *U3 = A, *V3 = A + 16, *W3 = A + 32, *R3 = A + 40;
  if (ln < 9) U3[ln] = f0;
  VSM_GROUP_SYNC();
  vsm_la::svd_group<3, 3>(U3, V3, W3, R3, ln);
  if (ln == 0) {
    double u[9], vv[9], D[9] = {W3[0], 0, 0, 0, W3[1], 0, 0, 0, 0}, UD[9], Vt[9], F[9];
    for (int i = 0; i < 9; i++) {
      u[i] = U3[i];
      vv[i] = V3[i];
    }
    vsm_la::mul(u, D, UD, 3, 3, 3);
    vsm_la::transpose(vv, Vt, 3, 3);
    vsm_la::mul(UD, Vt, F, 3, 3, 3);
    for (int i = 0; i < 9; i++) Fs[(size_t)k * 9 + i] = F[i];
  }
}

// triangulateChieral (viso/viso_mono.cpp:394-431): thread per (candidate blockIdx.y, match).
// P[c] = {P1 (3x4), P2_c (3x4)} row-major.  X[c][row][match]; chir[c] += both depths positive.
struct MonoCams {
  double P1[12], P2[4][12];
};
__global__ void __launch_bounds__(64)
    k_mono_triangulate(const MonoPt *__restrict__ raw, int n, MonoCams cams, double *__restrict__ X, int32_t *__restrict__ chir) {
  __shared__ double s_m[(16 + 16 + 4 + 4 + 4) * 64];
  const int i = blockIdx.x * blockDim.x + threadIdx.x, c = blockIdx.y;
  bool front = false;
  if (i < n) {
    const MonoPt q = raw[i];
    const double *P1 = cams.P1, *P2 = cams.P2[c];
    double *J = s_m + threadIdx.x, *v4 = J + 16 * 64, *w4 = v4 + 16 * 64, *r4 = w4 + 4 * 64, *c4 = r4 + 4 * 64;
    for (int j = 0; j < 4; j++) {
      J[(0 * 4 + j) * 64] = P1[2 * 4 + j] * q.u1p - P1[0 * 4 + j];
      J[(1 * 4 + j) * 64] = P1[2 * 4 + j] * q.v1p - P1[1 * 4 + j];
      J[(2 * 4 + j) * 64] = P2[2 * 4 + j] * q.u1c - P2[0 * 4 + j];
      J[(3 * 4 + j) * 64] = P2[2 * 4 + j] * q.v1c - P2[1 * 4 + j];
    }
    vsm_la::svd_nr<64>(J, 4, 4, 4, w4, v4, r4, c4);
    double x[4];
    for (int r = 0; r < 4; r++) x[r] = X[((size_t)c * 4 + r) * n + i] = v4[(r * 4 + 3) * 64];
    double ax = 0, bx = 0;
    for (int k = 0; k < 4; k++) ax += P1[2 * 4 + k] * x[k];
    for (int k = 0; k < 4; k++) bx += P2[2 * 4 + k] * x[k];
    front = ax * x[3] > 0 && bx * x[3] > 0;
  }
  const unsigned long long b = __ballot(front);
  if ((threadIdx.x & 63) == 0 && b) atomicAdd(&chir[c], (int32_t)__popcll(b));
}

namespace {

struct MonoGpu {  // device side of one estimator
  hipStream_t stream = nullptr;
  MonoPt *d_pts = nullptr, *d_raw = nullptr;
  double *d_F = nullptr, *d_d = nullptr, *d_sums = nullptr, *d_X = nullptr;
  int32_t *d_counts = nullptr, *d_picks = nullptr, *d_chir = nullptr;
  int32_t *h_counts = nullptr;  // pinned
  double *h_sums = nullptr;     // pinned
  double *h_X = nullptr;        // pinned, one candidate's 4 x n
  MonoPt *h_pts = nullptr;      // pinned staging of the normalised points / raw points
  int32_t *h_picks = nullptr;   // pinned staging of the samples
  double *h_F = nullptr;        // pinned, the fitted matrices coming back
  int cap_n = 0, cap_k = 0;
  bool ok = false;
  bool svd_on_device = false;   // the device reproduces the host's SVD bit for bit (self-test)
  bool init() {
    ok = hipStreamCreateWithFlags(&stream, hipStreamNonBlocking) == hipSuccess &&
         hipMalloc((void **)&d_chir, sizeof(int32_t) * 4) == hipSuccess;
    if (ok) svd_on_device = self_test();
    return ok;
  }
  // 64 random hypotheses through k_mono_fit against the host's vsm_linalg.h
  bool self_test() {
    const int n = 64, K = 64;
    if (!reserve(n, K)) return false;
    std::vector<MonoPt> pts(n);
    std::vector<int32_t> picks(K * 8);
    uint32_t s = 12345u;
    auto rnd = [&]() {
      s = s * 1664525u + 1013904223u;
      return (float)((s >> 8) & 0xffff) / 32768.0f - 1.0f;
    };
    for (auto &p : pts) p = {rnd(), rnd(), rnd(), rnd()};
    for (int k = 0; k < K; k++)
      for (int i = 0; i < 8; i++) picks[k * 8 + i] = (k * 7 + i * 5) % n;
    std::vector<double> dev(K * 9);
    if (hipMemcpyAsync(d_pts, pts.data(), sizeof(MonoPt) * n, hipMemcpyHostToDevice, stream) != hipSuccess) return false;
    if (hipMemcpyAsync(d_picks, picks.data(), sizeof(int32_t) * K * 8, hipMemcpyHostToDevice, stream) != hipSuccess) return false;
    hipLaunchKernelGGL(k_mono_fit, dim3((K + 15) / 16), dim3(256), 0, stream, d_pts, d_picks, K, d_F);
    if (hipMemcpyAsync(dev.data(), d_F, sizeof(double) * K * 9, hipMemcpyDeviceToHost, stream) != hipSuccess) return false;
    if (hipStreamSynchronize(stream) != hipSuccess) return false;
    for (int k = 0; k < K; k++) {
      double A[72], w[9], v[81], rv1[9], col[9], F0[9], F[9];
      for (int i = 0; i < 8; i++) {
        const MonoPt &q = pts[picks[k * 8 + i]];
        double *r = A + i * 9;
        r[0] = q.u1c * q.u1p; r[1] = q.u1c * q.v1p; r[2] = q.u1c; r[3] = q.v1c * q.u1p; r[4] = q.v1c * q.v1p;
        r[5] = q.v1c; r[6] = q.u1p; r[7] = q.v1p; r[8] = 1;
      }
      vsm_la::svd_nr(A, 8, 9, 9, w, v, rv1, col);
      for (int i = 0; i < 9; i++) F0[i] = v[i * 9 + 8];
      vsm_la::rank2_3x3(F0, F);
      if (memcmp(F, &dev[k * 9], sizeof(F)) != 0) return false;
    }
    return true;
  }
  bool reserve(int n, int k) {
    if (n > cap_n) {
      (void)hipFree(d_pts);
      (void)hipFree(d_raw);
      (void)hipFree(d_d);
      (void)hipFree(d_sums);
      (void)hipFree(d_X);
      (void)hipHostFree(h_sums);
      (void)hipHostFree(h_X);
      (void)hipHostFree(h_pts);
      cap_n = n + n / 2 + 256;
      if (hipHostMalloc((void **)&h_pts, sizeof(MonoPt) * cap_n, hipHostMallocDefault) != hipSuccess) return false;
      if (hipMalloc((void **)&d_pts, sizeof(MonoPt) * cap_n) != hipSuccess) return false;
      if (hipMalloc((void **)&d_raw, sizeof(MonoPt) * cap_n) != hipSuccess) return false;
      if (hipMalloc((void **)&d_X, sizeof(double) * 16 * cap_n) != hipSuccess) return false;
      if (hipHostMalloc((void **)&h_X, sizeof(double) * 4 * cap_n, hipHostMallocDefault) != hipSuccess) return false;
      if (hipMalloc((void **)&d_d, sizeof(double) * cap_n) != hipSuccess) return false;
      if (hipMalloc((void **)&d_sums, sizeof(double) * VOTE_SPLIT * cap_n) != hipSuccess) return false;
      if (hipHostMalloc((void **)&h_sums, sizeof(double) * VOTE_SPLIT * cap_n, hipHostMallocDefault) != hipSuccess) return false;
    }
    if (k > cap_k) {
      (void)hipFree(d_F);
      (void)hipFree(d_counts);
      (void)hipFree(d_picks);
      (void)hipHostFree(h_counts);
      (void)hipHostFree(h_picks);
      (void)hipHostFree(h_F);
      cap_k = k + 64;
      if (hipMalloc((void **)&d_picks, sizeof(int32_t) * 8 * cap_k) != hipSuccess) return false;
      if (hipHostMalloc((void **)&h_picks, sizeof(int32_t) * 8 * cap_k, hipHostMallocDefault) != hipSuccess) return false;
      if (hipHostMalloc((void **)&h_F, sizeof(double) * 9 * cap_k, hipHostMallocDefault) != hipSuccess) return false;
      if (hipMalloc((void **)&d_F, sizeof(double) * 9 * cap_k) != hipSuccess) return false;
      if (hipMalloc((void **)&d_counts, sizeof(int32_t) * cap_k) != hipSuccess) return false;
      if (hipHostMalloc((void **)&h_counts, sizeof(int32_t) * cap_k, hipHostMallocDefault) != hipSuccess) return false;
    }
    return true;
  }
  ~MonoGpu() {
    if (!ok) return;
    (void)hipFree(d_pts);
    (void)hipFree(d_raw);
    (void)hipFree(d_d);
    (void)hipFree(d_sums);
    (void)hipFree(d_X);
    (void)hipFree(d_F);
    (void)hipFree(d_counts);
    (void)hipFree(d_picks);
    (void)hipFree(d_chir);
    (void)hipHostFree(h_counts);
    (void)hipHostFree(h_sums);
    (void)hipHostFree(h_X);
    (void)hipHostFree(h_pts);
    (void)hipHostFree(h_picks);
    (void)hipHostFree(h_F);
    (void)hipStreamDestroy(stream);
  }
};

inline bool sampson_in(const MonoPt &p, const double *F, double thr) {
  const double u1 = p.u1p, v1 = p.v1p, u2 = p.u1c, v2 = p.v1c;
  const double Fx1u = F[0] * u1 + F[1] * v1 + F[2];
  const double Fx1v = F[3] * u1 + F[4] * v1 + F[5];
  const double Fx1w = F[6] * u1 + F[7] * v1 + F[8];
  const double Ftx2u = F[0] * u2 + F[3] * v2 + F[6];
  const double Ftx2v = F[1] * u2 + F[4] * v2 + F[7];
  const double x2tFx1 = u2 * Fx1u + v2 * Fx1v + Fx1w;
  const double d = x2tFx1 * x2tFx1 / (Fx1u * Fx1u + Fx1v * Fx1v + Ftx2u * Ftx2u + Ftx2v * Ftx2v);
  return fabs(d) < thr;
}

// fundamentalMatrix (viso/viso_mono.cpp:264-294) on normalised points; `rows` x 9 scratch in A
void fundamental(const MonoPt *pts, const int32_t *active, int na, double *A, double *col, double *F) {
  for (int i = 0; i < na; i++) {
    const MonoPt &q = pts[active[i]];
    double *r = A + (size_t)i * 9;
    r[0] = q.u1c * q.u1p;  // float products, like the reference's float fields
    r[1] = q.u1c * q.v1p;
    r[2] = q.u1c;
    r[3] = q.v1c * q.u1p;
    r[4] = q.v1c * q.v1p;
    r[5] = q.v1c;
    r[6] = q.u1p;
    r[7] = q.v1p;
    r[8] = 1;
  }
  double w[9], v[81], rv1[9];
  vsm_la::svd_nr(A, na, 9, 9, w, v, rv1, col);
  double F0[9];
  for (int k = 0; k < 9; k++) F0[k] = v[k * 9 + 8];  // singular vector of the smallest singular value
  vsm_la::rank2_3x3(F0, F);
}

template <class Runner>
void parallel_for(Runner *pool, int n, int min_chunk, const std::function<void(int, int)> &body) {
  const int lanes = pool ? std::min(pool->size(), 16) : 1;
  if (lanes <= 1 || n < 2 * min_chunk) {
    body(0, n);
    return;
  }
  const int per = std::max(min_chunk, (n + 2 * lanes - 1) / (2 * lanes));
  const int tasks = (n + per - 1) / per;
  pool->run(tasks, [&](int t) { body(t * per, std::min(n, (t + 1) * per)); });
}

class MonoEgo {
 public:
  vsm_vo_mono_params par;
  std::vector<MonoPt> pts;
  std::vector<double> Fs, X4, dvals, sums, scratch;
  std::vector<int32_t> counts, picks, deck;
  double timings[6] = {0, 0, 0, 0, 0, 0};

  // normalizeFeaturePoints (viso/viso_mono.cpp:215-262): the fields are floats, every update rounds
  bool normalise(const vsm_p_match *m, int n, double *Tp, double *Tc) {
    pts.resize((size_t)n);
    double cpu = 0, cpv = 0, ccu = 0, ccv = 0;
    for (int i = 0; i < n; i++) {
      cpu += m[i].u1p;
      cpv += m[i].v1p;
      ccu += m[i].u1c;
      ccv += m[i].v1c;
    }
    cpu /= (double)n;
    cpv /= (double)n;
    ccu /= (double)n;
    ccv /= (double)n;
    double sp = 0, sc = 0;
    for (int i = 0; i < n; i++) {
      MonoPt &p = pts[i];
      p.u1p = (float)(m[i].u1p - cpu);
      p.v1p = (float)(m[i].v1p - cpv);
      p.u1c = (float)(m[i].u1c - ccu);
      p.v1c = (float)(m[i].v1c - ccv);
    }
    for (int i = 0; i < n; i++) {
      const MonoPt &p = pts[i];
      sp += sqrtf(p.u1p * p.u1p + p.v1p * p.v1p);
      sc += sqrtf(p.u1c * p.u1c + p.v1c * p.v1c);
    }
    if (fabs(sp) < 1e-10 || fabs(sc) < 1e-10) return false;
    sp = sqrt(2.0) * (double)n / sp;
    sc = sqrt(2.0) * (double)n / sc;
    for (int i = 0; i < n; i++) {
      MonoPt &p = pts[i];
      p.u1p = (float)(p.u1p * sp);
      p.v1p = (float)(p.v1p * sp);
      p.u1c = (float)(p.u1c * sc);
      p.v1c = (float)(p.v1c * sc);
    }
    const double tp[9] = {sp, 0, -sp * cpu, 0, sp, -sp * cpv, 0, 0, 1};
    const double tc[9] = {sc, 0, -sc * ccu, 0, sc, -sc * ccv, 0, 0, 1};
    memcpy(Tp, tp, sizeof(tp));
    memcpy(Tc, tc, sizeof(tc));
    return true;
  }

  // estimateMotion (viso/viso_mono.cpp:103-187).  1 = success, 0 = failure with the inlier list
  // replaced, -1 = failure before the RANSAC started (inlier list untouched, like the reference).
  template <class Runner>
  int estimate(const vsm_p_match *m, int n, Runner *pool, MonoGpu *gpu, double *tr6, std::vector<int32_t> &inliers) {
    if (n < 10) return -1;
    const double t0 = vsm_now_us();
    double Tp[9], Tc[9];
    if (!normalise(m, n, Tp, Tc)) return -1;
    const int K = std::max(par.ransac_iters, 0);
    // --- samples: partial Fisher-Yates on a persistent identity deck, undone after each draw
    picks.resize((size_t)K * 8);
    deck.resize((size_t)n);
    for (int i = 0; i < n; i++) deck[i] = i;
    VsmDrawPlan plan[8];
    for (int i = 0; i < 8; i++) plan[i] = vsm_sampler_plan((uint32_t)i, (uint32_t)(n - 1));
    vsm_sampler_lock();
    for (int k = 0; k < K; k++) {
      int swapped[8];
      for (int i = 0; i < 8; i++) {
        swapped[i] = (int)vsm_sampler_draw(plan[i]);
        std::swap(deck[i], deck[swapped[i]]);
      }
      for (int i = 0; i < 8; i++) picks[(size_t)k * 8 + i] = deck[i];
      for (int i = 7; i >= 0; i--) std::swap(deck[i], deck[swapped[i]]);
    }
    vsm_sampler_unlock();
    // --- fundamental matrices of all hypotheses and their inlier counts
    Fs.resize((size_t)K * 9);
    counts.assign((size_t)K, 0);
    bool on_gpu = gpu && gpu->ok && K > 0 && gpu->reserve(n, K);
    bool fitted = false;
    double t1 = t0;
    if (on_gpu) {  // everything is staged through pinned memory, so nothing below blocks before the one sync
      memcpy(gpu->h_pts, pts.data(), sizeof(MonoPt) * n);
      on_gpu = hipMemcpyAsync(gpu->d_pts, gpu->h_pts, sizeof(MonoPt) * n, hipMemcpyHostToDevice, gpu->stream) == hipSuccess;
      if (on_gpu && gpu->svd_on_device) {  // hypotheses fitted on the device, F is counted where it was produced
        memcpy(gpu->h_picks, picks.data(), sizeof(int32_t) * 8 * K);
        on_gpu = hipMemcpyAsync(gpu->d_picks, gpu->h_picks, sizeof(int32_t) * 8 * K, hipMemcpyHostToDevice, gpu->stream) ==
                 hipSuccess;
        if (on_gpu) {
          hipLaunchKernelGGL(k_mono_fit, dim3((K + 15) / 16), dim3(256), 0, gpu->stream, gpu->d_pts, gpu->d_picks, K, gpu->d_F);
          fitted = true;
        }
      }
    }
    if (!fitted) {
      parallel_for(pool, K, 8, [&](int lo, int hi) {
        double A[72], col[9];
        for (int k = lo; k < hi; k++) fundamental(pts.data(), &picks[(size_t)k * 8], 8, A, col, &Fs[(size_t)k * 9]);
      });
      if (on_gpu) {
        memcpy(gpu->h_F, Fs.data(), sizeof(double) * 9 * K);
        on_gpu = hipMemcpyAsync(gpu->d_F, gpu->h_F, sizeof(double) * 9 * K, hipMemcpyHostToDevice, gpu->stream) == hipSuccess;
      }
    }
    t1 = vsm_now_us();
    if (on_gpu) {
      on_gpu = hipMemsetAsync(gpu->d_counts, 0, sizeof(int32_t) * K, gpu->stream) == hipSuccess;
      if (on_gpu) {
        hipLaunchKernelGGL(k_mono_inlier_count, dim3((n + 255) / 256, K), dim3(256), 0, gpu->stream, gpu->d_pts, n, gpu->d_F,
                           par.inlier_threshold, gpu->d_counts);
        on_gpu = hipMemcpyAsync(gpu->h_counts, gpu->d_counts, sizeof(int32_t) * K, hipMemcpyDeviceToHost, gpu->stream) ==
                 hipSuccess;
        if (on_gpu && fitted)
          on_gpu = hipMemcpyAsync(gpu->h_F, gpu->d_F, sizeof(double) * 9 * K, hipMemcpyDeviceToHost, gpu->stream) == hipSuccess;
        on_gpu = on_gpu && hipStreamSynchronize(gpu->stream) == hipSuccess;
        if (on_gpu) {
          memcpy(counts.data(), gpu->h_counts, sizeof(int32_t) * K);
          if (fitted) memcpy(Fs.data(), gpu->h_F, sizeof(double) * 9 * K);
        }
      }
    }
    if (!on_gpu) {
      if (fitted) {  // (a failed launch after a device fit: redo the fits on the host)
        parallel_for(pool, K, 8, [&](int lo, int hi) {
          double A[72], col[9];
          for (int k = lo; k < hi; k++) fundamental(pts.data(), &picks[(size_t)k * 8], 8, A, col, &Fs[(size_t)k * 9]);
        });
      }
      parallel_for(pool, K, 4, [&](int lo, int hi) {
        for (int k = lo; k < hi; k++) {
          int c = 0;
          for (int i = 0; i < n; i++) c += sampson_in(pts[i], &Fs[(size_t)k * 9], par.inlier_threshold) ? 1 : 0;
          counts[k] = c;
        }
      });
    }
    int best = -1, best_count = 0;
    for (int k = 0; k < K; k++)
      if (counts[k] > best_count) {
        best_count = counts[k];
        best = k;
      }
    inliers.clear();
    if (best >= 0)
      for (int i = 0; i < n; i++)
        if (sampson_in(pts[i], &Fs[(size_t)best * 9], par.inlier_threshold)) inliers.push_back(i);
    const double t2 = vsm_now_us();
    timings[0] = t1 - t0;  // (with device fits the kernel is only enqueued here; it shows up in [1])
    timings[1] = t2 - t1;
    if ((int)inliers.size() < 10) return 0;
    // --- F from all inliers, denormalise, essential matrix (:69-72, :121-129)
    double F[9];
    {
      scratch.resize((size_t)inliers.size() * 10);
      fundamental(pts.data(), inliers.data(), (int)inliers.size(), scratch.data(), scratch.data() + inliers.size() * 9, F);
    }
    double TcT[9], a[9], Fd[9], Kt[9], b[9], E0[9], E[9];
    const double Kd[9] = {par.f, 0, par.cu, 0, par.f, par.cv, 0, 0, 1};
    vsm_la::transpose(Tc, TcT, 3, 3);
    vsm_la::mul(TcT, F, a, 3, 3, 3);
    vsm_la::mul(a, Tp, Fd, 3, 3, 3);
    vsm_la::transpose(Kd, Kt, 3, 3);
    vsm_la::mul(Kt, Fd, b, 3, 3, 3);
    vsm_la::mul(b, Kd, E0, 3, 3, 3);
    vsm_la::rank2_3x3(E0, E);
    // --- EtoRt (:346-392)
    double U[9], S[3], V[9], rv1[3], col3[3];
    memcpy(U, E, sizeof(U));
    vsm_la::svd_nr(U, 3, 3, 3, S, V, rv1, col3);
    const double Wd[9] = {0, -1, 0, +1, 0, 0, 0, 0, 1}, Zd[9] = {0, +1, 0, -1, 0, 0, 0, 0, 0};
    double Ut[9], Vt[9], Wt[9], UZ[9], T[9], UW[9], Ra[9], UWt[9], Rb[9];
    vsm_la::transpose(U, Ut, 3, 3);
    vsm_la::transpose(V, Vt, 3, 3);
    vsm_la::transpose(Wd, Wt, 3, 3);
    vsm_la::mul(U, Zd, UZ, 3, 3, 3);
    vsm_la::mul(UZ, Ut, T, 3, 3, 3);
    vsm_la::mul(U, Wd, UW, 3, 3, 3);
    vsm_la::mul(UW, Vt, Ra, 3, 3, 3);
    vsm_la::mul(U, Wt, UWt, 3, 3, 3);
    vsm_la::mul(UWt, Vt, Rb, 3, 3, 3);
    double tt[3] = {T[2 * 3 + 1], T[0 * 3 + 2], T[1 * 3 + 0]}, tneg[3] = {-tt[0], -tt[1], -tt[2]};
    if (vsm_la::det3(Ra) < 0)
      for (double &x : Ra) x = -x;
    if (vsm_la::det3(Rb) < 0)
      for (double &x : Rb) x = -x;
    const double *Rs[4] = {Ra, Ra, Rb, Rb}, *ts[4] = {tt, tneg, tt, tneg};
    // --- triangulateChieral for the four candidates (:394-431)
    int chir[4] = {0, 0, 0, 0};
    MonoCams cams;
    {
      const double P1[12] = {Kd[0], Kd[1], Kd[2], 0, Kd[3], Kd[4], Kd[5], 0, Kd[6], Kd[7], Kd[8], 0};
      memcpy(cams.P1, P1, sizeof(P1));
      for (int c = 0; c < 4; c++) {
        double Rt[12];
        for (int i = 0; i < 3; i++) {
          for (int j = 0; j < 3; j++) Rt[i * 4 + j] = Rs[c][i * 3 + j];
          Rt[i * 4 + 3] = ts[c][i];
        }
        vsm_la::mul(Kd, Rt, cams.P2[c], 3, 3, 4);
      }
    }
    bool tri_gpu = gpu && gpu->ok && gpu->svd_on_device && gpu->reserve(n, 1);
    if (tri_gpu) {
      for (int i = 0; i < n; i++) gpu->h_pts[i] = {m[i].u1p, m[i].v1p, m[i].u1c, m[i].v1c};
      tri_gpu = hipMemcpyAsync(gpu->d_raw, gpu->h_pts, sizeof(MonoPt) * n, hipMemcpyHostToDevice, gpu->stream) == hipSuccess &&
                hipMemsetAsync(gpu->d_chir, 0, sizeof(int32_t) * 4, gpu->stream) == hipSuccess;
      if (tri_gpu) {
        hipLaunchKernelGGL(k_mono_triangulate, dim3((n + 63) / 64, 4), dim3(64), 0, gpu->stream, gpu->d_raw, n, cams, gpu->d_X,
                           gpu->d_chir);
        tri_gpu = hipMemcpyAsync(gpu->h_counts, gpu->d_chir, sizeof(int32_t) * 4, hipMemcpyDeviceToHost, gpu->stream) ==
                      hipSuccess &&
                  hipStreamSynchronize(gpu->stream) == hipSuccess;
        if (tri_gpu) memcpy(chir, gpu->h_counts, sizeof(chir));
      }
    }
    if (!tri_gpu) {
      X4.resize((size_t)4 * 4 * n);  // [candidate][row][match]
      for (int c = 0; c < 4; c++) {
        const double *P1 = cams.P1, *P2 = cams.P2[c];
        double *Xc = &X4[(size_t)c * 4 * n];
        std::atomic<int> num{0};
        parallel_for(pool, n, 64, [&](int lo, int hi) {
          int local = 0;
          for (int i = lo; i < hi; i++) {
            double J[16], w4[4], v4[16], r4[4], c4[4];
            for (int j = 0; j < 4; j++) {
              J[0 * 4 + j] = P1[2 * 4 + j] * m[i].u1p - P1[0 * 4 + j];
              J[1 * 4 + j] = P1[2 * 4 + j] * m[i].v1p - P1[1 * 4 + j];
              J[2 * 4 + j] = P2[2 * 4 + j] * m[i].u1c - P2[0 * 4 + j];
              J[3 * 4 + j] = P2[2 * 4 + j] * m[i].v1c - P2[1 * 4 + j];
            }
            vsm_la::svd_nr(J, 4, 4, 4, w4, v4, r4, c4);
            double x[4];
            for (int r = 0; r < 4; r++) x[r] = Xc[(size_t)r * n + i] = v4[r * 4 + 3];
            double ax = 0, bx = 0;  // third rows of P1*X and P2*X
            for (int k = 0; k < 4; k++) ax += P1[2 * 4 + k] * x[k];
            for (int k = 0; k < 4; k++) bx += P2[2 * 4 + k] * x[k];
            if (ax * x[3] > 0 && bx * x[3] > 0) local++;
          }
          num.fetch_add(local, std::memory_order_relaxed);
        });
        chir[c] = num.load();
      }
    }
    int pick = -1, max_in = 0;
    for (int c = 0; c < 4; c++)
      if (chir[c] > max_in) {
        max_in = chir[c];
        pick = c;
      }
    const double t3 = vsm_now_us();
    timings[2] = t3 - t2;
    if (pick < 0) return 0;  // (the reference would go on with an empty rotation matrix here)
    const double *Xc = nullptr, *R = Rs[pick], *t = ts[pick];
    if (tri_gpu) {  // only the chosen candidate's points come back
      if (hipMemcpyAsync(gpu->h_X, gpu->d_X + (size_t)pick * 4 * n, sizeof(double) * 4 * n, hipMemcpyDeviceToHost,
                         gpu->stream) != hipSuccess ||
          hipStreamSynchronize(gpu->stream) != hipSuccess)
        return 0;
      Xc = gpu->h_X;
    } else {
      Xc = &X4[(size_t)pick * 4 * n];
    }
    // --- points in front of the camera, median of their L1 norms (:137-161, :189-213)
    dvals.clear();
    std::vector<double> &l1 = sums;
    l1.clear();
    std::vector<double> yz;
    yz.reserve((size_t)2 * n);
    for (int i = 0; i < n; i++) {
      const double w4 = Xc[(size_t)3 * n + i];
      const double x0 = w4 != 0 ? Xc[i] / w4 : 0, x1 = w4 != 0 ? Xc[(size_t)n + i] / w4 : 0,
                   x2 = w4 != 0 ? Xc[(size_t)2 * n + i] / w4 : 0;
      if (x2 > 0) {
        yz.push_back(x1);
        yz.push_back(x2);
        l1.push_back(fabs(x0) + fabs(x1) + fabs(x2));
      }
    }
    const int np = (int)l1.size();
    if (np < 10) return 0;
    std::nth_element(l1.begin(), l1.begin() + np / 2, l1.end());
    const double median = l1[np / 2];
    if (median > par.motion_threshold) return 0;
    const double sigma = median / 50.0, weight = 1.0 / (2.0 * sigma * sigma), threshold = median / par.motion_threshold;
    // --- findBestPlane (:75-101)
    const double n0 = cos(-par.pitch), n1 = sin(-par.pitch);
    dvals.resize((size_t)np);
    for (int i = 0; i < np; i++) {
      double s = 0;
      s += n0 * yz[2 * i];
      s += n1 * yz[2 * i + 1];
      dvals[i] = s;
    }
    const int best_idx = best_plane(pool, gpu, np, threshold, weight);
    const double best_d = dvals[best_idx];
    const double t4 = vsm_now_us();
    timings[3] = t4 - t3;
    const double ry = asin(R[0 * 3 + 2]);
    const double rx = asin(-R[1 * 3 + 2] / cos(ry));
    const double rz = asin(-R[0 * 3 + 1] / cos(ry));
    tr6[0] = rx;
    tr6[1] = ry;
    tr6[2] = rz;
    for (int i = 0; i < 3; i++) tr6[3 + i] = t[i] * par.height / best_d;
    return 1;
  }

  // exact vote sum of candidate i: libm exp(), the reference's summation order
  double exact_sum(int i, int np, double weight) const {
    double sum = 0;
    for (int j = 0; j < np; j++) {
      const double dist = dvals[j] - dvals[i];
      sum += exp(-dist * dist * weight);
    }
    return sum;
  }

  template <class Runner>
  int best_plane(Runner *pool, MonoGpu *gpu, int np, double threshold, double weight) {
    bool on_gpu = gpu && gpu->ok && np >= 512 && gpu->reserve(np, 1);
    if (on_gpu) {
      on_gpu = hipMemcpyAsync(gpu->d_d, dvals.data(), sizeof(double) * np, hipMemcpyHostToDevice, gpu->stream) == hipSuccess;
      if (on_gpu) {
        hipLaunchKernelGGL(k_mono_plane_vote, dim3((np + 255) / 256, VOTE_SPLIT), dim3(256), 0, gpu->stream, gpu->d_d, np,
                           threshold, weight, gpu->d_sums);
        on_gpu = hipMemcpyAsync(gpu->h_sums, gpu->d_sums, sizeof(double) * VOTE_SPLIT * np, hipMemcpyDeviceToHost,
                                gpu->stream) ==
                     hipSuccess &&
                 hipStreamSynchronize(gpu->stream) == hipSuccess;
      }
    }
    if (on_gpu) {
      // the device exp() is within a few ulp of libm's: only candidates this close to the proposed
      // maximum can be the true first maximum; they are judged exactly, in index order
      double top = 0;
      for (int i = 0; i < np; i++) {  // add the slices up (into slice 0)
        double t = 0;
        for (int c = 0; c < VOTE_SPLIT; c++) t += gpu->h_sums[(size_t)c * np + i];
        gpu->h_sums[i] = t;
        top = std::max(top, t);
      }
      double best_sum = 0;
      int best_idx = 0;
      for (int i = 0; i < np; i++)
        if (dvals[i] > threshold && gpu->h_sums[i] >= top * (1.0 - 1e-9)) {
          const double s = exact_sum(i, np, weight);
          if (s > best_sum) {
            best_sum = s;
            best_idx = i;
          }
        }
      return best_idx;
    }
    sums.assign((size_t)np, 0.0);
    parallel_for(pool, np, 16, [&](int lo, int hi) {
      for (int i = lo; i < hi; i++)
        if (dvals[i] > threshold) sums[i] = exact_sum(i, np, weight);
    });
    double best_sum = 0;
    int best_idx = 0;
    for (int i = 0; i < np; i++)
      if (dvals[i] > threshold && sums[i] > best_sum) {
        best_sum = sums[i];
        best_idx = i;
      }
    return best_idx;
  }
};

}  // namespace

struct vsm_vo_mono {
  vsm_handle *matcher = nullptr;
  MonoEgo ego;
  MonoGpu gpu;
  double T[16];
  bool valid = false;
  std::vector<vsm_p_match> matched;
  std::vector<int32_t> inliers;
  double timings[4] = {0, 0, 0, 0};
};

static int mono_update_motion(vsm_vo_mono *v) {  // VisualOdometry::updateMotion, viso/viso.cpp:42-58
  double tr[6];
  std::vector<int32_t> fresh;
  const int rc = v->ego.estimate(v->matched.data(), (int)v->matched.size(), vsm_forkjoin_of(v->matcher), &v->gpu, tr, fresh);
  if (rc >= 0) v->inliers.swap(fresh);
  if (rc != 1) return 0;
  vsm_pose_matrix(tr, v->T);
  v->valid = true;
  return 1;
}

static int mono_after_push(vsm_vo_mono *v) {  // viso/viso_mono.cpp:34-38
  const double t0 = vsm_now_us();
  vsm_match(v->matcher, 0, nullptr);
  const double t1 = vsm_now_us();
  vsm_bucket(v->matcher, v->ego.par.bucket_max_features, (float)v->ego.par.bucket_width, (float)v->ego.par.bucket_height);
  v->matched.resize((size_t)vsm_num_matches(v->matcher));
  if (!v->matched.empty()) vsm_get_matches(v->matcher, v->matched.data(), (int32_t)v->matched.size());
  const double t2 = vsm_now_us();
  const int ok = mono_update_motion(v);
  const double t3 = vsm_now_us();
  v->timings[0] = t1 - t0;
  v->timings[1] = t2 - t1;
  v->timings[2] = t3 - t2;
  v->timings[3] = t3 - t0;
  return ok;
}

extern "C" {

void vsm_vo_mono_default_params(vsm_vo_mono_params *p) {
  memset(p, 0, sizeof(*p));
  vsm_default_params(&p->match);
  p->match.f = 1;
  p->match.base = 1;
  p->bucket_max_features = 2;
  p->bucket_width = 50;
  p->bucket_height = 50;
  p->f = 1;
  p->height = 1.0;
  p->pitch = 0.0;
  p->ransac_iters = 2000;
  p->inlier_threshold = 0.00001;
  p->motion_threshold = 100.0;
}

vsm_vo_mono *vsm_vo_mono_create(const vsm_vo_mono_params *p) {
  vsm_handle *m = vsm_create(&p->match);
  if (!m) return nullptr;
  vsm_vo_mono *v = new vsm_vo_mono();
  v->matcher = m;
  v->ego.par = *p;
  for (int i = 0; i < 16; i++) v->T[i] = (i % 5 == 0) ? 1.0 : 0.0;
  srand(0);  // viso/viso.cpp:35
  if (!v->gpu.init()) {
    fprintf(stderr, "visomatch: could not create the egomotion stream\n");
    vsm_destroy(m);
    delete v;
    return nullptr;
  }
  return v;
}

void vsm_vo_mono_destroy(vsm_vo_mono *v) {
  if (!v) return;
  vsm_destroy(v->matcher);
  delete v;
}

int vsm_vo_mono_process(vsm_vo_mono *v, const uint8_t *I, int32_t w, int32_t h, int32_t bpl, int replace) {
  vsm_push_back(v->matcher, I, nullptr, w, h, bpl, replace);
  return mono_after_push(v);
}
int vsm_vo_mono_process_device(vsm_vo_mono *v, const uint8_t *dI, int32_t w, int32_t h, int32_t bpl, int replace) {
  vsm_push_back_device(v->matcher, dI, nullptr, w, h, bpl, replace);
  return mono_after_push(v);
}
int vsm_vo_mono_process_matches(vsm_vo_mono *v, const vsm_p_match *m, int32_t n) {
  v->matched.assign(m, m + (n > 0 ? n : 0));
  return mono_update_motion(v);
}
void vsm_vo_mono_get_motion(vsm_vo_mono *v, double *T16) { memcpy(T16, v->T, sizeof(v->T)); }
int vsm_vo_mono_motion_valid(vsm_vo_mono *v) { return v->valid ? 1 : 0; }
int32_t vsm_vo_mono_num_matches(vsm_vo_mono *v) { return (int32_t)v->matched.size(); }
int32_t vsm_vo_mono_get_matches(vsm_vo_mono *v, vsm_p_match *out, int32_t cap) {
  const int32_t n = std::min((int32_t)v->matched.size(), cap);
  if (n > 0) memcpy(out, v->matched.data(), (size_t)n * sizeof(vsm_p_match));
  return n;
}
int32_t vsm_vo_mono_num_inliers(vsm_vo_mono *v) { return (int32_t)v->inliers.size(); }
int32_t vsm_vo_mono_get_inliers(vsm_vo_mono *v, int32_t *out, int32_t cap) {
  const int32_t n = std::min((int32_t)v->inliers.size(), cap);
  if (n > 0) memcpy(out, v->inliers.data(), (size_t)n * sizeof(int32_t));
  return n;
}
float vsm_vo_mono_gain(vsm_vo_mono *v, const int32_t *inliers, int32_t n) { return vsm_gain(v->matcher, inliers, n); }
vsm_handle *vsm_vo_mono_matcher(vsm_vo_mono *v) { return v->matcher; }
int vsm_vo_mono_device_svd(vsm_vo_mono *v) { return v->gpu.svd_on_device ? 1 : 0; }
void vsm_vo_mono_get_timings(vsm_vo_mono *v, double *out10) {
  memcpy(out10, v->timings, sizeof(v->timings));
  memcpy(out10 + 4, v->ego.timings, sizeof(v->ego.timings));
}

int32_t vsm_host_estimate_motion_mono(const vsm_vo_mono_params *p, const vsm_p_match *m, int32_t n, int32_t threads,
                                      double *tr6, double *T16, int32_t *inliers, int32_t *n_inliers) {
  MonoEgo ego;
  ego.par = *p;
  std::vector<int32_t> keep;
  int rc;
  if (threads > 1) {
    VsmPool pool(threads);
    rc = ego.estimate(m, n, &pool, (MonoGpu *)nullptr, tr6, keep);
  } else {
    rc = ego.estimate(m, n, (VsmPool *)nullptr, (MonoGpu *)nullptr, tr6, keep);
  }
  if (rc == 1 && T16) vsm_pose_matrix(tr6, T16);
  if (rc >= 0) {
    *n_inliers = (int32_t)keep.size();
    if (!keep.empty()) memcpy(inliers, keep.data(), keep.size() * sizeof(int32_t));
  }
  return rc;
}

}  // extern "C"
