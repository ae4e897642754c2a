/* visomatch.h -- C-ABI of libvisomatch.so, the MI355X-native (gfx950, HIP) replacement for
 * libviso2's per-frame matcher hot path as vendored in dphoyes/OpenCL-Structure-from-Motion.
 *
 * Every entry point names the reference interface it replaces (paths relative to the
 * reference repository).  Plain pointers and sizes only; no C++/torch types cross this line.
 * The header-only C++ class in include/matcher.h forwards the reference's own
 * `class Matcher` surface (viso/matcher.h:37-136) to these functions.
 *
 * Error convention: functions returning int give 0 (VSM_OK) on success or a negative
 * VSM_E* code.  The reference's C++ API has no error returns (bad dims print
 * "ERROR: Image dimension mismatch!" to stderr and leave the state untouched,
 * viso/matcher.cpp:103-106; matchFeatures with missing buffers returns silently and keeps
 * the previous matches, :190-216); the C++ wrapper swallows the codes to keep that behaviour.
 *
 * Threading: one handle = one HIP device + one stream; a handle is not re-entrant.
 * Distinct handles are independent (unlike the reference, whose Delaunay code keeps
 * file-scope state, viso/triangle.cpp:541-550).
 */
#ifndef VISOMATCH_H
#define VISOMATCH_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define VSM_OK 0
#define VSM_EDIMS (-1)     /* the reference's "Image dimension mismatch" */
#define VSM_ENOTREADY (-2) /* matchFeatures' silent early return: ring buffer not filled */
#define VSM_EHIP (-3)      /* HIP runtime error (message on stderr) */
#define VSM_EARG (-4)

typedef struct vsm_handle vsm_handle;

/* Matcher::parameters, viso/matcher.h:42-69 (same field order and defaults) */
typedef struct vsm_params {
  int32_t nms_n;
  int32_t nms_tau;
  int32_t match_binsize;
  int32_t match_radius;
  int32_t match_disp_tolerance;
  int32_t outlier_disp_tolerance;
  int32_t outlier_flow_tolerance;
  int32_t multi_stage;
  int32_t half_resolution;
  int32_t refinement;
  double f, cu, cv, base;
} vsm_params;

/* Matcher::p_match, viso/matcher.h:86-100 -- identical 48-byte layout */
typedef struct vsm_p_match {
  float u1p, v1p;
  int32_t i1p;
  float u2p, v2p;
  int32_t i2p;
  float u1c, v1c;
  int32_t i1c;
  float u2c, v2c;
  int32_t i2c;
} vsm_p_match;

/* Matcher::parameters::parameters(), viso/matcher.h:57-68 */
void vsm_default_params(vsm_params *p);

/* Matcher::Matcher(parameters), viso/matcher.cpp:33-61.  Binds the calling thread's current
 * HIP device.  Returns NULL if no HIP device is usable (never falls back to the CPU). */
vsm_handle *vsm_create(const vsm_params *p);

/* Matcher::~Matcher(), viso/matcher.cpp:64-93 */
void vsm_destroy(vsm_handle *h);

/* Matcher::setIntrinsics, viso/matcher.h:78-83 */
void vsm_set_intrinsics(vsm_handle *h, double f, double cu, double cv, double base);

/* Matcher::pushBack(I1,I2,dims,replace), viso/matcher.cpp:95-181 (I2 == NULL: the mono
 * overload viso/matcher.h:118).  Host images, row stride bpl.  The input may be reused as
 * soon as the call returns. */
int vsm_push_back(vsm_handle *h, const uint8_t *I1, const uint8_t *I2, int32_t width, int32_t height, int32_t bpl,
                  int replace);

/* Same, for images that already live in this device's HBM (the bench's resident-input path).
 * Ordering and lifetime contract of every entry point that takes device pointers (this one, vsm_sequence_run with
 * on_device = 1, vsm_vo_stereo_process_device, vsm_vo_mono_process_device):
 *  - the images are READ ASYNCHRONOUSLY on the handle's own (non-blocking) stream.  Work that produces them on another
 *    stream must have completed, or be ordered in front with vsm_wait_for_stream() right before the call;
 *  - they must stay valid and unchanged until the handle has consumed them: until the next synchronising call on the
 *    handle for a push (vsm_match, any getter), until the call returns for vsm_sequence_run and the VO entry points. */
int vsm_push_back_device(vsm_handle *h, const uint8_t *dI1, const uint8_t *dI2, int32_t width, int32_t height,
                         int32_t bpl, int replace);
/* orders everything the handle enqueues from now on behind the work `hip_stream` (a hipStream_t; NULL = the null stream)
 * holds at this moment: an event recorded there, a wait on the handle's stream - no host synchronisation */
int vsm_wait_for_stream(vsm_handle *h, void *hip_stream);

/* Matcher::matchFeatures(method, Tr_delta), viso/matcher.cpp:183-241.  method 0 flow, 1 stereo,
 * 2 quad.  Tr_delta: NULL or 12 doubles = rows 0..2 of the 4x4 matrix, row-major
 * (what viso/matcher.cpp:989-1002 reads). */
int vsm_match(vsm_handle *h, int32_t method, const double *Tr_delta);

/* Matcher::getMatches(), viso/matcher.h:131 */
int32_t vsm_num_matches(vsm_handle *h);
int32_t vsm_get_matches(vsm_handle *h, vsm_p_match *out, int32_t cap);

/* Matcher::bucketFeatures, viso/matcher.cpp:243-284 (uses the C library rand() like the reference) */
int vsm_bucket(vsm_handle *h, int32_t max_features, float bucket_width, float bucket_height);

/* Matcher::getGain, viso/matcher.cpp:286-324 */
float vsm_gain(vsm_handle *h, const int32_t *inliers, int32_t n);

/* ---- look-ahead API: a whole sequence at once ----
 * Semantically identical to
 *     for f in 0..n_frames-1:  pushBack(left[f], right[f], dims, false);  matchFeatures(method, Tr[f])
 * on a fresh Matcher (viso/matcher.cpp:95, :183), but the frames of a chunk (VSM_SEQ_CHUNK; default 110
 * with ten or more host-pool threads and device-resident frames, 80 with six to nine or host-resident frames, else 50, and 50 in
 * the host-shared form) go through every kernel in one launch and the host stages of the chunk's frame pairs run in
 * parallel.  left/right: n_frames images frame_stride bytes apart (host or, with on_device != 0,
 * HBM); right == NULL (mono) with stereo / quad matching goes frame by frame (the reference's matchFeatures returns early).
 * Tr_delta: NULL or n_frames x 12 doubles, Tr_valid: NULL (all valid) or n_frames flags.
 * The streaming ring buffer (vsm_push_back / vsm_match) is not touched except by the fallback.
 * STREAMS: besides the handle's own non-blocking streams the GPU-resident form uses the process's NULL stream (result
 * export copies, device vertex sorts) and drains it before it returns: an application that keeps work of its own on the
 * null stream - PyTorch's default stream is it, blocking streams synchronise with it - sets option "seq_null_stream" = 0
 * (vsm_set_option) and the library uses a non-blocking stream of its own instead. */
int vsm_sequence_run(vsm_handle *h, const uint8_t *left, const uint8_t *right, int64_t frame_stride, int on_device,
                     int32_t n_frames, int32_t width, int32_t height, int32_t bpl, int32_t method,
                     const double *Tr_delta, const uint8_t *Tr_valid);
/* getMatches() as it would read after frame `frame` */
int32_t vsm_sequence_num_matches(vsm_handle *h, int32_t frame);
int32_t vsm_sequence_get_matches(vsm_handle *h, int32_t frame, vsm_p_match *out, int32_t cap);
/* wall-clock split of the last vsm_sequence_run() on the caller's thread, microseconds: {launching and waiting
 * for the GPU, host stages it takes part in (prior statistics, final drain), total, chunk size} - the stages
 * overlap, so the first two are not what the GPU / the host pool were busy for */
void vsm_sequence_get_timings(vsm_handle *h, double *out4);
/* which form of the look-ahead path the last vsm_sequence_run took: 2 = GPU-resident (lists stay in HBM from the first
 * matching pass to the survivors; the host only runs Triangle's vertex sort), 1 = host-shared (VSM_SEQ_V2=0, or a list
 * the device chain declines) */
int32_t vsm_sequence_path(vsm_handle *h);
/* Measurement / test switches of a handle.  They are read from the environment once, by vsm_create (VSM_SEQ_V2,
 * VSM_SEQ_CHUNK, VSM_SEQ_DC_STREAMS, VSM_SEQ_SERIAL, VSM_SEQ_GPU_SORTS, VSM_SEQ_EARLY_EXPORT); this call changes one
 * afterwards: name = the variable's name without the VSM_ prefix, in lower case ("seq_serial", "seq_chunk", ...).
 * Option-only names (never read from the environment): "front" (0: separate ingest / halving / Sobel passes instead of the
 * fused front end), "dc_gpu", "dc_full", "dc_watchdog_ms", "dc_fault_inject" (the GPU's share of the final stage in the
 * host-shared form), and the scheduling experiments of the GPU-resident form recorded in DESIGN_HISTORY.md 6c:
 * "seq_keys_dma", "seq_keys_pieces", "seq_ties1_null", "seq_ties1_host", "seq_last_first", "seq_export_budget", "seq_first_chunk",
 * "seq_p2_first", "seq_block_after_p2", "seq_warm_gaps" (DESIGN.md 5, 9);
 * "seq_null_stream" (above); "seq_host_pinned" (below, vsm_host_register); "seq_host_inorder" (0: host-resident frames in the
 * run-ahead order of resident input instead of chunk by chunk as they arrive); "match_heads" (1, before the first image: the
 * second matching pass on 64-byte per-bin head records - measured slower); "fused_features" / "feat_order" (0: the separate filter, suppression, record and bin kernels
 * instead of k_feat_dense / k_feat_sparse / k_feat_scan / k_feat_order) and "filter_planes" (1: vsm_push_back keeps the
 * blob / corner responses in HBM for vsm_get_filter_responses; the fused kernels leave them in LDS otherwise).
 * None of them changes a result.  Returns VSM_OK, or VSM_EARG for an unknown name.  (No counterpart in the reference.) */
int vsm_set_option(vsm_handle *h, const char *name, int32_t value);
/* Host threads near the GPU: the library confines the threads IT creates (host pool, look-ahead poller) to the CPUs of the
 * device's NUMA node (/sys/bus/pci/devices/<bus id>/local_cpulist, within what the process may use; looked up once per
 * process by the first vsm_create; spread over that node's L3 domains, thread i on domain i mod n; VSM_HOST_AFFINITY=1: the
 * node only, =0: off) - on a two-socket MI355X node a rank whose host
 * threads run on the other socket loses 6 % of the look-ahead rate.  The caller's threads are left alone; this returns the
 * CPUs chosen (up to cap of them in out; the return value is how many there are, 0 = none) so that the caller can put the
 * thread that calls vsm_sequence_run there too, as bench.py does. */
int32_t vsm_local_cpus(int32_t *out, int32_t cap);
/* The per-frame calls (vsm_match, vsm_vo_stereo_process) split ONE triangulation - the final Matcher::removeOutliers,
 * viso/matcher.cpp:1207-1377 - over up to eight fork-join threads that all sit in one L3 domain of that node (they take turns
 * on one mesh; domain = the device ordinal mod the domains, so the ranks of one socket take one each; VSM_FJ_DOMAIN=k: domain k,
 * -1: dealt over the domains), each worker on a physical core of its own (cores
 * 1, 2, ... of the domain; VSM_FJ_CORES=0: anywhere in the domain - then two of them may share a core's hardware threads and
 * halve each other between the phases they spin through).  The caller's thread takes part in that work: this returns the
 * CPUs of the domain's core 0, which is left to it (the whole domain with VSM_FJ_CORES=0), and a caller that confines the
 * thread calling vsm_match / vsm_vo_stereo_process to them (sched_setaffinity) saves the transfers between core complexes
 * and never shares a core with a worker - 0.55 -> 0.47-0.50 ms per 1242 x 375 stereo pair, live VO 0.70 -> 0.62 ms per
 * frame; bench.py does for its per-frame legs.
 * Same conventions as vsm_local_cpus; 0 = no such domain (no node found, affinity off, VSM_FJ_DOMAIN=-1). */
int32_t vsm_forkjoin_cpus(int32_t *out, int32_t cap);
/* Host-resident input at the link's rate.  Matcher::pushBack takes pageable host pointers (viso/matcher.cpp:95-181) and so do
 * vsm_push_back / vsm_sequence_run(on_device = 0): pageable memory is gathered into a pinned buffer by the host pool before it
 * can cross PCIe by DMA.  A caller whose images live in a buffer it reuses can page-lock that buffer ONCE
 * (vsm_host_register = hipHostRegister; or allocate it with hipHostMalloc) and promise so with
 * vsm_set_option(handle, "seq_host_pinned", 1): vsm_sequence_run then copies straight out of the caller's memory.  The promise is
 * the caller's: with the option set and pageable images the copies fall back to the runtime's staged path (slow, still correct).
 * Unregister before the buffer is freed.  (No counterpart in the reference.) */
int vsm_host_register(const void *p, uint64_t bytes);
int vsm_host_unregister(const void *p);
/* Device memory the process keeps: large blocks of closed handles and re-created contexts wait in a cache
 * for the next one instead of going back to the driver, whose background clear of released VRAM slows every call for
 * 44 ms per GB.  out[0] = blocks in the cache, out[1] = their bytes, out[2] = large blocks in use.  vsm_device_pool_trim()
 * hands the cached ones back now (an application about to need the memory for itself).  (No counterpart in the reference.) */
void vsm_device_pool_stats(int64_t out[3]);
void vsm_device_pool_trim(void);

/* ---- stage-level views for parity tests (the reference's private members) ---- */

/* m1p1.. / n1p1.. (viso/matcher.h:232-235).  which: 0=1p1 1=2p1 2=1c1 3=2c1 4=1p2 5=2p2 6=1c2 7=2c2;
 * records are int32[12] = {u,v,0,class,d1..d8} (viso/matcher.cpp:716-718). */
int32_t vsm_num_features(vsm_handle *h, int32_t which);
int32_t vsm_get_features(vsm_handle *h, int32_t which, int32_t *out, int32_t cap_records);

/* match list after each private stage of the last vsm_match():
 * 0 pass-1 matching(), 1 pass-1 removeOutliers(), 2 pass-2 matching(), 3 refinement(), 4 final */
void vsm_set_stage_capture(vsm_handle *h, int on); /* stage 2 costs one extra D2H: off by default */
int32_t vsm_stage_size(vsm_handle *h, int32_t stage);
int32_t vsm_stage_get(vsm_handle *h, int32_t stage, vsm_p_match *out, int32_t cap);

/* Matcher::ranges (viso/matcher.h:152-157,245): 16 floats per statistics bin */
int32_t vsm_num_ranges(vsm_handle *h);
int32_t vsm_get_ranges(vsm_handle *h, float *out, int32_t cap_bins);

/* I?{p,c}_du/_dv[_full] (viso/matcher.h:237-240).  which: 0=1p 1=2p 2=1c 3=2c.  Returns the
 * plane size in bytes (bpl*h) or 0 when absent; du/dv may be NULL to query the size. */
int32_t vsm_get_gradients(vsm_handle *h, int32_t which, int32_t full, uint8_t *du, uint8_t *dv);

/* blob / corner filter responses of the current left image (f1,f2 of viso/matcher.cpp:651-678;
 * transient in the reference - and here: with the default suppression radii they never leave LDS, so set option
 * "filter_planes" = 1 before the push whose responses are wanted).  Returns elements per plane or 0 (none kept). */
int32_t vsm_get_filter_responses(vsm_handle *h, int16_t *f1, int16_t *f2);

/* work counters of the last vsm_match(): {findMatch calls, 0, 0, matches refined, matches out}
 * (slots 1,2 are only counted by the CPU oracle) */
void vsm_get_counters(vsm_handle *h, int64_t *out5);

/* wall-clock split of the last vsm_match() in microseconds:
 * {pass-1 GPU+sync, pass-1 host (Delaunay+prior), pass-2 GPU+sync, final host Delaunay, total} */
void vsm_get_timings(vsm_handle *h, double *out5);

/* per-kernel device time measured with HIP events on the handle's own stream (bench.py's
 * roofline leg).  vsm_set_profiling(h,1) zeroes the accumulators and starts recording; vsm_set_profiling(h, 100 + id)
 * records kernel `id` only (vsm_kernel_name): every span costs two event records on the kernel's stream, and those of
 * all kernels on all streams together disturb the pipeline they measure (1100 + id: also prints every span on stderr). */
void vsm_set_profiling(vsm_handle *h, int on);
int32_t vsm_num_kernels(void);
const char *vsm_kernel_name(int32_t id);
void vsm_get_kernel_stats(vsm_handle *h, double *total_ms, int64_t *launches);

/* host-only view of the exact Delaunay used by removeOutliers (triangulate("zQB") of
 * viso/triangle.cpp:8500 on integer points in [0,16384)^2); needs no GPU.  Returns the number
 * of triangles; tris gets vertex triples by input index. */
int32_t vsm_host_delaunay(const int32_t *x, const int32_t *y, int32_t n, int32_t *tris, int32_t cap, int32_t threads);

/* the same triangulation computed in three steps -- prepare (sort, kd order, tree), independent
 * sub-trees of at most max_task_points points, merges above them -- the form in which the look-ahead
 * path shares the work between host and GPU; must equal vsm_host_delaunay() for every split */
int32_t vsm_host_delaunay_split(const int32_t *x, const int32_t *y, int32_t n, int32_t *tris, int32_t cap,
                                int32_t max_task_points, int32_t device_top_points);

/* test hook for the shared form: (device_kd != 0: the kd order of the sorted keys,) sub-trees (at most
 * max_task_points points) on the GPU, one thread each, then the merge nodes of at most device_top_points points
 * level by level (device_top_points < 0: one wave per sub-tree inside LDS instead, any max_task_points), the rest
 * on the host; -1 on a HIP error */
int32_t vsm_debug_delaunay_gpu(const int32_t *x, const int32_t *y, int32_t n, int32_t *tris, int32_t cap,
                               int32_t max_task_points, int32_t device_top_points, int32_t device_kd);

double vsm_debug_dc_bench(const int32_t *x, const int32_t *y, int32_t n, int32_t max_task_points, int32_t device_top_points,
                          int32_t device_kd, int32_t njobs, int32_t reps);   /* kernel microseconds for njobs triangulations at once */

/* test hooks: which of several matches at one pixel stands for the point in the exact Delaunay (Triangle's
 * randomised vertex sort decides, viso/triangle.cpp:5447 + :6183): (index of the smallest-index match, index of the
 * one the sort puts first) for every pixel where they differ - from the host emulation and from the GPU's
 * (k_dc_ties, one wave; kernel_us may be null); return the count, -1 where the GPU declines (list too long) */
int32_t vsm_host_ties(const int32_t *x, const int32_t *y, int32_t n, int32_t *pairs, int32_t cap);
int32_t vsm_debug_ties_gpu(const int32_t *x, const int32_t *y, int32_t n, int32_t *pairs, int32_t cap, double *kernel_us);

/* Matcher::removeOutliers (viso/matcher.cpp:1207-1377) and Matcher::computePriorStatistics (:734-868) on a match
 * list of the caller: the host code of the per-frame path, and the GPU-resident chain of the look-ahead path
 * (keys, vertex sort - on the device if gpu_ties, else on the host -, kd order, block sub-trees, cached merge levels,
 * support votes, survivors, prior statistics) run on `copies` identical jobs at once.  Test hooks: the two must
 * agree byte for byte.  out gets the survivors, ranges (may be null) the prior boxes in the device layout of the
 * match kernels ([bin][stage][u_min, u_max, v_min, v_max], p->match_radius as given); return the number of
 * survivors, -1 on a HIP error, -2 if the device chain declines the list; kernel_us (may be null): microseconds of
 * the device chain for all copies. */
int32_t vsm_host_outliers_and_prior(const vsm_params *p, const vsm_p_match *list, int32_t n, int32_t method, vsm_p_match *out,
                                    int32_t cap, float *ranges, int32_t w, int32_t h);
/* The same host code the way vsm_match runs it on a frame's final list: `threads` fork-join threads (the caller's among
 * them), the triangulation started from the packed pixels (x | y << 16) alone, flows, votes and the survivors' copy split
 * over the threads.  threads <= 1 is vsm_host_outliers_and_prior.  No GPU needed: the CPU suite compares it with the oracle. */
int32_t vsm_host_outliers_and_prior_threads(const vsm_params *p, const vsm_p_match *list, int32_t n, int32_t method, vsm_p_match *out,
                                            int32_t cap, float *ranges, int32_t w, int32_t h, int32_t threads);
int32_t vsm_debug_dc2(const vsm_params *p, const vsm_p_match *list, int32_t n, int32_t method, int32_t gpu_ties, int32_t copies,
                      vsm_p_match *out, int32_t cap, float *ranges, int32_t w, int32_t h, double *kernel_us);
/* Test hook of the device chain's merge levels: a node's band (the records near its cut) is cached in 256 + f * sqrt(points)
 * LDS lines; a node that needs more is redone by one lane in global memory.  f < 0 restores the default (12); a small f
 * forces that second path, which no list of the benchmark takes.  Process-wide. */
void vsm_debug_dc2_band_factor(int32_t f);

/* ---- stereo visual odometry on top of the matcher ----
 * class VisualOdometryStereo, viso/viso_stereo.h:28-88 + viso/viso.h:28-131: process() =
 * pushBack + matchFeatures(2, Tr_delta if valid) + bucketFeatures + getMatches + updateMotion
 * (viso/viso_stereo.cpp:33-40), with estimateMotion's RANSAC / Gauss-Newton
 * (viso/viso_stereo.cpp:42-315) spread over the matcher's host pool. */
typedef struct {
  vsm_params match;              /* VisualOdometry::parameters::match */
  int32_t bucket_max_features;   /* VisualOdometry::bucketing, viso/viso.h:45-54 */
  double bucket_width, bucket_height;
  double f, cu, cv;              /* VisualOdometry::calibration, viso/viso.h:33-42 */
  double base;                   /* VisualOdometryStereo::parameters, viso/viso_stereo.h:33-44 */
  int32_t ransac_iters;
  double inlier_threshold;
  int32_t reweighting;
} vsm_vo_stereo_params;
typedef struct vsm_vo_stereo vsm_vo_stereo;

void vsm_vo_stereo_default_params(vsm_vo_stereo_params *p);
/* VisualOdometryStereo::VisualOdometryStereo, viso/viso_stereo.cpp:27-29 (+ srand(0), viso/viso.cpp:35) */
vsm_vo_stereo *vsm_vo_stereo_create(const vsm_vo_stereo_params *p);
void vsm_vo_stereo_destroy(vsm_vo_stereo *v);
/* VisualOdometryStereo::process, viso/viso_stereo.cpp:33-40; returns 1 (true) / 0 (false) */
int vsm_vo_stereo_process(vsm_vo_stereo *v, const uint8_t *I1, const uint8_t *I2, int32_t width, int32_t height,
                          int32_t bpl, int replace);
int vsm_vo_stereo_process_device(vsm_vo_stereo *v, const uint8_t *dI1, const uint8_t *dI2, int32_t width,
                                 int32_t height, int32_t bpl, int replace);
/* VisualOdometry::process(std::vector<p_match>), viso/viso.h:74-77 */
int vsm_vo_stereo_process_matches(vsm_vo_stereo *v, const vsm_p_match *m, int32_t n);
/* getMotion(): row-major 4x4 Tr_delta, kept from the last success (viso/viso.h:79-87) */
void vsm_vo_stereo_get_motion(vsm_vo_stereo *v, double *T16);
int vsm_vo_stereo_motion_valid(vsm_vo_stereo *v);
/* getNumberOfMatches()/the bucketed p_matched, getNumberOfInliers()/getInlierIndices(), getGain() */
int32_t vsm_vo_stereo_num_matches(vsm_vo_stereo *v);
int32_t vsm_vo_stereo_get_matches(vsm_vo_stereo *v, vsm_p_match *out, int32_t cap);
int32_t vsm_vo_stereo_num_inliers(vsm_vo_stereo *v);
int32_t vsm_vo_stereo_get_inliers(vsm_vo_stereo *v, int32_t *out, int32_t cap);
float vsm_vo_stereo_gain(vsm_vo_stereo *v, const int32_t *inliers, int32_t n);
/* the Matcher inside (VisualOdometry::matcher) */
vsm_handle *vsm_vo_stereo_matcher(vsm_vo_stereo *v);
/* wall-clock split of the last process() in microseconds: {matchFeatures, bucketing + copy,
 * egomotion, total after the push} */
void vsm_vo_stereo_get_timings(vsm_vo_stereo *v, double *out4);
/* VisualOdometry::getRandomSample draws from ONE engine per process, seeded 71
 * (viso/viso.cpp:93); so does this library.  This re-seeds it (parity tests replay fixtures that
 * were recorded from a fresh process). */
void vsm_vo_sampler_seed(uint32_t seed);

/* ---- lock-step multi-sequence stereo visual odometry (multi-sequence per GPU) ----
 * K independent sequences advance together: vsm_multi_process is VisualOdometryStereo::process (viso/viso_stereo.cpp:33-40)
 * for the next stereo pair of EVERY sequence - pushBack, matchFeatures(2, live Tr_delta of that sequence), bucketFeatures,
 * updateMotion - with one launch per kernel over all K pairs and the K egomotion estimates side by side on the host pool.
 * Each sequence has the rand() stream (bucketing, srand(0) at construction, viso/viso.cpp:35) and the RANSAC sampler
 * (viso/viso.cpp:93) that a process of its own would have, so sequence k's matches, inliers and Tr_delta equal what the
 * reference gives for that sequence alone. */
typedef struct vsm_multi vsm_multi;
vsm_multi *vsm_multi_create(const vsm_vo_stereo_params *p, int32_t n_sequences);
void vsm_multi_destroy(vsm_multi *m);
/* left / right: n_sequences images seq_stride bytes apart (sequence k's pair at k * seq_stride), host memory or, with
 * on_device != 0, HBM; ok_out: NULL or n_sequences flags = process()'s return value per sequence */
int vsm_multi_process(vsm_multi *m, const uint8_t *left, const uint8_t *right, int64_t seq_stride, int on_device, int32_t width,
                      int32_t height, int32_t bpl, int32_t *ok_out);
int32_t vsm_multi_num_sequences(vsm_multi *m);
void vsm_multi_get_motion(vsm_multi *m, int32_t seq, double *T16);   /* VisualOdometry::getMotion */
int vsm_multi_motion_valid(vsm_multi *m, int32_t seq);
/* bucketed = 0: Matcher::getMatches() as matchFeatures left it; 1: VisualOdometry::getMatches() (after bucketFeatures) */
int32_t vsm_multi_num_matches(vsm_multi *m, int32_t seq, int bucketed);
int32_t vsm_multi_get_matches(vsm_multi *m, int32_t seq, int bucketed, vsm_p_match *out, int32_t cap);
int32_t vsm_multi_num_inliers(vsm_multi *m, int32_t seq);            /* VisualOdometry::getInlierIndices */
int32_t vsm_multi_get_inliers(vsm_multi *m, int32_t seq, int32_t *out, int32_t cap);
/* wall clock of the last step, microseconds: features, first pass + its chain, second pass + final chain, bucketing + egomotion */
void vsm_multi_get_timings(vsm_multi *m, double *out4);

/* ---- monocular visual odometry ----
 * class VisualOdometryMono, viso/viso_mono.h:28-90: process() = pushBack + matchFeatures(0) +
 * bucketFeatures + getMatches + updateMotion (viso/viso_mono.cpp:33-39); estimateMotion
 * (viso/viso_mono.cpp:103-187) with the two inner loops the reference offloads to OpenCL
 * (viso/viso_mono_cl.cpp, viso/kernels/plane_and_inliers.cl) as HIP kernels.  Results equal the
 * reference's CPU class (double arithmetic) bit for bit. */
typedef struct {
  vsm_params match;              /* VisualOdometry::parameters::match */
  int32_t bucket_max_features;   /* VisualOdometry::bucketing, viso/viso.h:45-54 */
  double bucket_width, bucket_height;
  double f, cu, cv;              /* VisualOdometry::calibration, viso/viso.h:33-42 */
  double height, pitch;          /* VisualOdometryMono::parameters, viso/viso_mono.h:33-46 */
  int32_t ransac_iters;
  double inlier_threshold, motion_threshold;
} vsm_vo_mono_params;
typedef struct vsm_vo_mono vsm_vo_mono;

void vsm_vo_mono_default_params(vsm_vo_mono_params *p);
vsm_vo_mono *vsm_vo_mono_create(const vsm_vo_mono_params *p);      /* viso/viso_mono.cpp:27-28 */
void vsm_vo_mono_destroy(vsm_vo_mono *v);
/* VisualOdometryMono::process, viso/viso_mono.cpp:33-39; returns 1 (true) / 0 (false) */
int vsm_vo_mono_process(vsm_vo_mono *v, const uint8_t *I, int32_t width, int32_t height, int32_t bpl, int replace);
int vsm_vo_mono_process_device(vsm_vo_mono *v, const uint8_t *dI, int32_t width, int32_t height, int32_t bpl, int replace);
int vsm_vo_mono_process_matches(vsm_vo_mono *v, const vsm_p_match *m, int32_t n);   /* viso/viso.h:74-77 */
void vsm_vo_mono_get_motion(vsm_vo_mono *v, double *T16);
int vsm_vo_mono_motion_valid(vsm_vo_mono *v);
int32_t vsm_vo_mono_num_matches(vsm_vo_mono *v);
int32_t vsm_vo_mono_get_matches(vsm_vo_mono *v, vsm_p_match *out, int32_t cap);
int32_t vsm_vo_mono_num_inliers(vsm_vo_mono *v);
int32_t vsm_vo_mono_get_inliers(vsm_vo_mono *v, int32_t *out, int32_t cap);
float vsm_vo_mono_gain(vsm_vo_mono *v, const int32_t *inliers, int32_t n);
vsm_handle *vsm_vo_mono_matcher(vsm_vo_mono *v);
/* 1 if the hypothesis fits and triangulations run on the GPU (the device reproduced the host's SVD
 * bit for bit in the creation-time self-test), 0 if they run on the host pool */
int vsm_vo_mono_device_svd(vsm_vo_mono *v);
/* microseconds of the last process(): {matchFeatures, bucketing + copy, egomotion, total after the
 * push, then inside the egomotion: fundamental matrices, inlier counting, R|t + triangulation,
 * plane vote, 0, 0} */
void vsm_vo_mono_get_timings(vsm_vo_mono *v, double *out10);
/* host-only view of VisualOdometryMono::estimateMotion (no GPU: the two inner loops run on the host
 * threads).  1 = success, 0 = failure, -1 = failure before the RANSAC (inliers untouched). */
int32_t vsm_host_estimate_motion_mono(const vsm_vo_mono_params *p, const vsm_p_match *m, int32_t n, int32_t threads,
                                      double *tr6, double *T16, int32_t *inliers, int32_t *n_inliers);

/* host-only view of the egomotion solver (VisualOdometryStereo::estimateMotion,
 * viso/viso_stereo.cpp:42-146); needs no GPU.  Returns 1 = success (tr6 = rx,ry,rz,tx,ty,tz),
 * 0 = failure, -1 = fewer than 6 matches (inliers/n_inliers untouched, like the reference's early
 * return).  On success T16 (may be NULL) gets transformationVectorToMatrix(tr6), viso/viso.cpp:60-89.
 * inliers must hold n entries. */
int32_t vsm_host_estimate_motion_stereo(const vsm_vo_stereo_params *p, const vsm_p_match *m, int32_t n, int32_t threads,
                                        double *tr6, double *T16, int32_t *inliers, int32_t *n_inliers);

const char *vsm_version(void);

#ifdef __cplusplus
}
#endif
#endif
