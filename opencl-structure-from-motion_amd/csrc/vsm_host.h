// Host-side stages of matchFeatures that stay on the CPU (SURVEY.md section 8a rows O1, M4, B1 and
// the least-squares tail of refinement==2): exact Delaunay support test, prior statistics,
// bucketing, gain.  Product code -- independent of oracle/.
#pragma once

#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include <atomic>
#include <condition_variable>
#include <deque>
#include <memory>
#include <functional>
#include <mutex>
#include <thread>
#include <vector>

#include "visomatch.h"
#include "vsm_dc_mesh.h"

// Host threads near the GPU.  An MI355X node has two CPU sockets with four GPUs each; a rank's host threads that run on the
// other socket read the pinned buffers the GPU's DMA wrote (keys, refined lists) across the socket link, and the look-ahead
// call takes 4.45-4.57 ms instead of 4.17-4.23 (tools/numa_probe.sh; left to the scheduler: anything between, process by
// process).  vsm_affinity_from_device() reads the CPUs of the device's NUMA node (/sys/bus/pci/devices/<bdf>/local_cpulist)
// once per DEVICE; every thread the library creates for a handle on that device (pool workers, fork-join workers, the
// look-ahead poller) then confines itself to those CPUs, within what the process is allowed.  The caller's own threads are
// left alone.  VSM_HOST_AFFINITY=0 switches it off.
// Hoare-partition outcomes of parts of 3..7 keys (ExactDelaunay's TinyTable) as one array: size n starts at (4^n - 64) / 3,
// entry [GE | LE << n] = eight 3-bit source positions, left << 24, (right + 1) << 28; returns the number of entries (21824)
int vsm_host_tiny_table(uint32_t *out, int cap);
void vsm_affinity_from_device(const char *pci_bus_id);  // "0000:0d:00.0" (hipDeviceGetPCIBusId); makes that device's record the calling thread's current one
int vsm_affinity_domain_cpus(void *affinity, int domain, int *out, int cap);  // CPUs of one L3 domain of the record (nullptr: the device looked up last); returns how many
void vsm_forkjoin_domain_hint(int device);               // (vsm_create) without VSM_FJ_DOMAIN the workers' domain is the device's ordinal mod the domains
int vsm_forkjoin_domain();                               // the L3 domain the fork-join workers share (VSM_FJ_DOMAIN; -1: they are dealt over all)
void *vsm_affinity_current();                           // the calling thread's current record (what pools created now will pin their workers by)
void vsm_pin_this_thread(void *affinity = nullptr);     // nullptr: the calling thread's current record
void vsm_pin_this_thread_together(void *affinity, int domain);  // ... into L3 domain `domain` of the record, whoever asks
// ... onto physical core `core` (mod the domain's cores) of that domain: both hardware threads of the core, nobody else's
void vsm_pin_this_thread_core(void *affinity, int domain, int core);
int vsm_affinity_core_cpus(void *affinity, int domain, int core, int *out, int cap);  // that core's CPUs; returns how many (0: unknown)
bool vsm_forkjoin_per_core();  // the fork-join workers sit on a core each (VSM_FJ_CORES=0: anywhere in their domain)
int vsm_thread_domain();                                // the L3 domain (index in the pin sequence) this thread was pinned to, or -1
int vsm_affinity_cpus(int *out, int cap);  // the CPUs the threads of the device looked up last may use (for the caller who wants its own threads there too); returns how many

// Small task pool for the host stages (Delaunay sub-problems / frame pairs are independent).
// Work arrives as batches of n index tasks; batches are served FIFO.  run() is fork-join (the
// caller takes part); submit()/wait() let the caller keep the GPU busy while a batch is being
// worked on.  Workers spin for a while when idle before they block, so back-to-back frames do not
// pay a wake-up.
class VsmPool {
 public:
  struct Batch {
    std::function<void(int)> fn;
    int n = 0;
    bool urgent = false;
    std::atomic<int> next{0}, done{0};
  };
  typedef std::shared_ptr<Batch> Ticket;

  explicit VsmPool(int threads);
  ~VsmPool();
  int size() const { return nthreads_; }
  void run(int ntasks, const std::function<void(int)> &fn);  // blocking
  Ticket submit(int ntasks, std::function<void(int)> fn, bool urgent = false);  // asynchronous
  void wait(const Ticket &t);                                // helps until the batch is done
  bool help() { return work_one(); }                         // a thread with nothing else to do runs one queued task; false if there is none

 private:
  void worker();
  bool work_one();  // runs one task of the oldest unfinished batch; false if none
  int nthreads_;
  void *aff_ = nullptr;  // where the workers confine themselves (vsm_affinity_current() of the creating thread)
  std::vector<std::thread> threads_;
  std::mutex mu_;
  std::condition_variable cv_;
  std::deque<Ticket> queue_;
  std::atomic<uint64_t> posted_{0};
  std::atomic<int> sleepers_{0};
  std::atomic<bool> stop_{false};
  int spin_us_ = 2000;
};

// shared with vsm_ego.cpp (defined in vsm_api.cpp)
class VsmForkJoin;
VsmPool *vsm_pool_of(vsm_handle *h);
VsmForkJoin *vsm_forkjoin_of(vsm_handle *h);  // the spinning pool the per-frame host stages use
double vsm_now_us();
// the process-wide RANSAC sampler of VisualOdometry::getRandomSample (vsm_ego.cpp) and the pose
// matrix of a (rx,ry,rz,tx,ty,tz) vector (viso/viso.cpp:60-89), shared by the stereo and mono egomotion
void vsm_sampler_lock();
void vsm_sampler_unlock();
uint32_t vsm_sampler_between(uint32_t lo, uint32_t hi);  // call with the lock held
struct VsmDrawPlan {  // a (lo, hi) range with the divisions of the distribution done once
  uint64_t per_cell, reject_from, magic;
  uint32_t lo;
};
VsmDrawPlan vsm_sampler_plan(uint32_t lo, uint32_t hi);
uint32_t vsm_sampler_draw(const VsmDrawPlan &p);          // call with the lock held
void vsm_pose_matrix(const double *tr6, double *T16);

// Lock-free fork-join pool for the fine-grained phases inside ONE Delaunay (a dozen tasks of
// 10-100 us each): task claiming is a CAS on (generation << 32 | next index), so a worker that
// is late for generation g can never run g+1's task with g's closure.
class VsmForkJoin {
 public:
  explicit VsmForkJoin(int threads);
  ~VsmForkJoin();
  int size() const { return nthreads_; }
  void run(int ntasks, const std::function<void(int)> &fn);

 private:
  void worker(int index);
  bool claim(uint64_t g, int n, int &idx);
  int nthreads_;
  void *aff_ = nullptr;
  std::vector<std::thread> threads_;
  std::mutex mu_;
  std::condition_variable cv_;
  std::atomic<uint64_t> gen_{0}, next_{0};
  std::atomic<int> done_{0}, sleepers_{0};
  int ntasks_ = 0;
  const std::function<void(int)> *fn_ = nullptr;
  std::atomic<bool> stop_{false};
  int spin_us_ = 2000;
};

// Exact replica of Triangle 1.6's divide-and-conquer Delaunay ("zQB", as called from
// Matcher::removeOutliers, viso/matcher.cpp:1255-1256) for integer-valued points in [0,16384)^2:
// same vertex order (incl. the randomised quicksort that decides which of two duplicate points
// survives), same alternating cuts, same merge decisions, exact integer predicates.  Unlike
// Triangle it is re-entrant, allocation-free in steady state, works on cache-friendly packed keys
// and solves the independent sub-problems of the top recursion levels on several host threads.
class ExactDelaunay {
 public:
  typedef DcMesh::OTri OTri;
  // points are (x[i], y[i]); after run(), triangles() lists vertex triples by input index
  // (async: a pool that can take the emulated vertex sort while this thread and `pool` triangulate)
  void run(const int32_t *x, const int32_t *y, int32_t n, VsmForkJoin *pool = nullptr, VsmPool *async = nullptr);

  // The same in steps, so that the lower part of the tree can be triangulated elsewhere (the
  // look-ahead path hands it to the GPU, csrc/vsm_dc.hip):
  //   prepare()  emulated vertex sort, duplicate removal, kd order, tree layout; sub-trees of at
  //              most max_task_points points become tasks, the merge nodes above them with at most
  //              device_top_points points are listed apart (deepest level first).
  //              false: fewer than 2 distinct points.
  //   tasks()    slices [off, off+n) with their cut axis; a solver runs DcMesh::recurse on each and
  //              reports the two hull handles with set_node_hull() (solve_tasks() does it here)
  //   device_merges()  {slice, axis, node, children}: DcMesh::merge_hulls on the children's handles,
  //              level by level (device_levels(): nodes per level), again reported with
  //              set_node_hull() (solve_merges() does it here)
  //   finish()   the remaining merges, bottom-up.
  struct Task {
    int32_t off, n, axis, node;
  };
  struct Merge {
    int32_t off, n, axis, node, left, right;
  };
  //   (defer_order: prepare() stops before the kd order - mesh().key stays in (x,y) order - for a
  //   solver that orders the keys itself; order_keys() does it here later if it turns out to be needed)
  //   (defer_ties: the triangulation itself does not depend on Triangle's randomised vertex sort - the sorted
  //   distinct points and their kd order are what they are; the sort only decides which of several matches at
  //   the SAME pixel stands for the point, i.e. which input index a position carries.  With defer_ties the keys
  //   are ordered by a plain radix sort, every point carries the smallest of its input indices for the time
  //   being, and the emulated sort runs apart - resolve_ties(), safe next to kd order / sub-trees / merges, and
  //   nothing at all if no two matches share a pixel; apply_ties() then puts the right indices in place, after
  //   the sub-trees have been solved and before anybody reads ids)
  bool prepare(const int32_t *x, const int32_t *y, int32_t n, int32_t max_task_points, VsmForkJoin *pool = nullptr,
               int32_t device_top_points = 0, bool defer_order = false, bool defer_ties = false);
  bool has_ties() const { return has_ties_; }
  void resolve_ties();
  void apply_ties();
  // for a solver that runs the emulated sort elsewhere (k_dc_ties): the keys in input order, and its verdict
  const uint64_t *tie_keys() const { return emu_.data(); }
  int32_t tie_count() const { return n_in_; }
  void set_ties(const int32_t *pairs, int32_t count) {  // (index a point carries, index it should carry) x count
    patches_.clear();
    for (int32_t k = 0; k < count; k++) patches_.push_back(std::make_pair(pairs[2 * k], pairs[2 * k + 1]));
    ties_resolved_ = true;
  }
  const std::vector<std::pair<int32_t, int32_t>> &ties() const { return patches_; }
  // The vertex sort alone, for a solver that keeps the triangulation elsewhere (the GPU-resident look-ahead path):
  // keys = (x << 34) | (y << 20) | index in list order; writes (smallest index at a shared pixel, index the sort
  // puts first) for every shared pixel where the two differ; returns their number, or -1 if there are more than cap.
  int32_t sort_ties(const uint64_t *keys, int32_t n, int32_t *pairs, int32_t cap);
  void order_keys(VsmForkJoin *pool = nullptr) {
    if (!ordered_) kd_order(m_, pool);
    ordered_ = true;
  }
  const std::vector<Task> &tasks() const { return tasks_; }
  const std::vector<Merge> &device_merges() const { return dmerges_; }
  const std::vector<int32_t> &device_levels() const { return dlevels_; }
  int32_t num_nodes() const { return (int32_t)nodes_.size(); }
  void solve_tasks(VsmForkJoin *pool = nullptr);
  void solve_merges();
  void set_node_hull(int32_t node, OTri farleft, OTri farright) {
    nodes_[node].fl = farleft;
    nodes_[node].fr = farright;
  }
  void set_task_hull(int32_t task, OTri farleft, OTri farright) { set_node_hull(tasks_[task].node, farleft, farright); }
  void finish(VsmForkJoin *pool = nullptr);
  int32_t points() const { return m_; }  // distinct points = sorted positions
  DcMesh mesh() { return DcMesh{tri_.data(), pt_.data(), id_.data(), key_.data()}; }

  // the triangle list as vertex triples (built on first use; removeOutliers walks the slots instead)
  int32_t num_triangles() {
    list_triangles();
    return ntri_out_;
  }
  const int32_t *triangles() {
    list_triangles();
    return tri_out_.data();
  }
  // slot view: slots 0 .. num_slots()-1; a slot holds a triangle if slot_vertices() returns true
  // (vertex triple by input index, same orientation as triangles())
  int32_t num_slots() const { return 2 * m_; }
  const int32_t *slot_record(int32_t t) const { return &tri_[(size_t)t * 8 + 4]; }  // its three vertices by position (-1: none)
  const int32_t *ids() const { return id_.data(); }                                  // position -> input index
  inline bool slot_vertices(int32_t t, int32_t *q) const {
    const int32_t *v = &tri_[(size_t)t * 8 + 4];
    if ((v[0] | v[1] | v[2]) < 0) return false;
    q[0] = id_[v[1]];
    q[1] = id_[v[2]];
    q[2] = id_[v[0]];
    return true;
  }

 private:
  struct Node {  // one sub-problem of the divide-and-conquer tree
    int32_t off, n, axis, tbase, left, right;
    OTri fl, fr;
  };
  std::vector<uint64_t> key_;         // (x << 34) | (y << 20) | input index
  std::vector<uint64_t> stop_;        // scratch of the branch-free partition: two bit masks
  std::vector<uint64_t> sort_stack_;  // pending parts of the emulated quicksort
  std::vector<uint64_t> emu_, rs_;    // defer_ties: the keys in input order for the emulated sort; radix scratch
  std::vector<uint64_t> tie_bits_;
  std::vector<std::pair<int32_t, int32_t>> patches_;  // (index a point carries, index it should carry)
  int32_t n_in_ = 0;
  bool has_ties_ = false, ties_resolved_ = true;
  std::vector<uint32_t> xl_, yl_, yr_, tmp_, ord_;  // kd_order(): presorted lists, y-ranks
  std::vector<uint64_t> ybuf0_, ybuf1_, k2_;
  std::vector<uint32_t> pt_;          // by sorted position: x | y << 16
  std::vector<int32_t> id_;           // by sorted position: input index
  std::vector<int32_t> tri_, tri_out_;
  std::vector<Node> nodes_;
  std::vector<Task> tasks_;
  std::vector<Merge> dmerges_;    // merge nodes left to the sub-tree solver's side, deepest level first
  std::vector<int32_t> dlevels_;  // their number per level
  std::vector<std::vector<int32_t>> levels_, dev_levels_;  // internal nodes by depth: the host's, the sub-tree solver's
  int32_t max_task_ = 2, device_top_ = 0;
  void merge_node(int32_t q);
  int32_t ntri_out_ = 0, m_ = 0;
  bool listed_ = true, ordered_ = true;
  void list_triangles();
  uint64_t seed_ = 1;
  long t_sort_ = 0, t_kd_ = 0;  // ns, VSM_DEBUG_TIMING

  uint32_t rnd(uint32_t choices);
  void vertex_sort(uint64_t *a, int32_t n);
  void kd_order(int32_t m, VsmForkJoin *pool);
  int32_t build_tree(int32_t off, int32_t n, int axis, int32_t depth);
};

struct VsmHostWork {
  VsmPool *async = nullptr;  // may take the emulated vertex sort of a triangulation (ExactDelaunay::run)
  ExactDelaunay del;
  std::vector<int32_t> x, y, support, support_parts;
  std::vector<float> support_pos;  // per sorted position: flow u, flow v, disparity, votes (vsm_host_count_support)
  std::vector<float> fu, fv, dp;        // per match: flow and disparity, what the support test compares
  std::vector<vsm_p_match> tmp_list;
  VsmForkJoin *pool = nullptr;  // optional: threads for the sub-problems of one triangulation
};

// Matcher::removeOutliers, viso/matcher.cpp:1207-1377 (in place; order preserved)
void vsm_host_remove_outliers(VsmHostWork &w, const vsm_params &p, std::vector<vsm_p_match> &m, int method);
// the two ends of it, for callers that run the triangulation (w.del) themselves: the per-match
// arrays (coordinates, flow, disparity), then support counting + survivors
void vsm_host_outliers_begin(VsmHostWork &w, const vsm_p_match *in, int32_t n, int method);
// ... in two steps, for a caller that has the list's pixels (x | y << 16 per match) before the list itself: the triangulation
// (w.del.run on w.x / w.y) can start from the first, the support test waits for the second
void vsm_host_outliers_begin_xy(VsmHostWork &w, const uint32_t *xy, int32_t n);
void vsm_host_outliers_begin_flows(VsmHostWork &w, const vsm_p_match *in, int32_t n, int method);
void vsm_host_count_support(VsmHostWork &w, const vsm_params &p, int32_t n, int method);  // -> w.support, from w.del
void vsm_host_keep_supported(const vsm_p_match *in, const int32_t *support, int32_t n, std::vector<vsm_p_match> &out);
void vsm_host_keep_supported(std::vector<vsm_p_match> &list, const int32_t *support);  // in place
void vsm_host_outliers_end(VsmHostWork &w, const vsm_params &p, const vsm_p_match *in, int32_t n, int method,
                           std::vector<vsm_p_match> &out);
// the same from a read-only list (e.g. the host-mapped export of the GPU) into `out`
void vsm_host_remove_outliers_from(VsmHostWork &w, const vsm_params &p, const vsm_p_match *in, int32_t n, int method,
                                   std::vector<vsm_p_match> &out);

// Matcher::computePriorStatistics, viso/matcher.cpp:734-868 -> ranges[bin][16]
void vsm_host_prior_statistics(const vsm_params &p, const int32_t *dims_c, const std::vector<vsm_p_match> &m,
                               int method, std::vector<float> &ranges);

// least-squares tail of Matcher::parabolicFitting, viso/matcher.cpp:1425-1453.
// c9 = 3x3 costs around the 7x7 minimum at (du,dv); returns false if the match must be dropped.
bool vsm_host_parabolic_update(const int32_t *c9, int du, int dv, float &u2, float &v2);

// Matcher::bucketFeatures, viso/matcher.cpp:243-284
void vsm_host_bucket(std::vector<vsm_p_match> &m, int max_features, float bucket_width, float bucket_height);
// A rand() stream of one's own: the C library's generator (glibc: random()'s TYPE_3 additive feedback) behind the
// reentrant interface, seeded like srand().  The lock-step multi-sequence API gives every sequence one, so that K
// sequences in one process bucket exactly like K processes of the reference, each after its own srand(0) (viso/viso.cpp:35).
struct VsmRandStream {
  struct random_data rd;
  char state[128];
  VsmRandStream() { seed(0); }
  void seed(unsigned s) {
    memset(&rd, 0, sizeof(rd));
    memset(state, 0, sizeof(state));
    initstate_r(s, state, sizeof(state), &rd);
  }
  uint32_t next() {
    int32_t v = 0;
    random_r(&rd, &v);
    return (uint32_t)v;
  }
};
void vsm_host_bucket_with(std::vector<vsm_p_match> &m, int max_features, float bucket_width, float bucket_height, VsmRandStream &rnd);
// One sequence's stereo egomotion state for the lock-step multi-sequence API (vsm_ego.cpp): estimateMotion's working
// set, a RANSAC sampler of its own (the reference's is one std::default_random_engine per process, viso/viso.cpp:93) and
// the rand() stream its bucketing draws from.
struct VsmEgoSeq;
VsmEgoSeq *vsm_ego_seq_create(const vsm_vo_stereo_params *p);
void vsm_ego_seq_destroy(VsmEgoSeq *e);
// VisualOdometryStereo::process behind the matcher (viso/viso_stereo.cpp:37-39, viso/viso.cpp:42-58): bucketFeatures on
// `matches` (in: getMatches() after matchFeatures, out: the bucketed list), estimateMotion on one thread; on success T16
// gets the new Tr_delta and *valid becomes true.  Returns 1 / 0 like process().
// (shared: the host pool whose idle threads may take a share of this sequence's RANSAC hypotheses - the call itself runs on one
// of its threads -, or nullptr)
int vsm_ego_seq_step(VsmEgoSeq *e, std::vector<vsm_p_match> &matches, double *T16, bool *valid, std::vector<int32_t> &inliers, VsmPool *shared = nullptr);

// Matcher::getGain, viso/matcher.cpp:286-324
float vsm_host_gain(const uint8_t *I1p, const uint8_t *I1c, const int32_t *dims_p, const int32_t *dims_c,
                    const std::vector<vsm_p_match> &m, const int32_t *inliers, int32_t n);
