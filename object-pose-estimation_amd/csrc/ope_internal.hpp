// ope_internal.hpp — shared host/device definitions of libope_hip.so (gfx950 only).
#pragma once

#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdint>
#include <memory>
#include <mutex>
#include <string>
#include <utility>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "../../include/ope.h"

namespace ope {

// ---------------------------------------------------------------------------
// Device-side view of the target search index: an implicit, perfectly balanced
// binary tree of ORIENTED bounding boxes.  Node ids are heap indices (root 1,
// children 2i / 2i+1); nodes at depth D (ids 2^D … 2^(D+1)-1) are leaf buckets.
// Leaf j owns the contiguous point range [j*n >> D, (j+1)*n >> D) of `pts`.
// A node is 3 float4 (48 B):  {c.xyz, h0} {a0.xyz, h1} {a1.xyz, h2}
//   c = box centre, a0/a1 = two unit axes (the third is a0 x a1), h = half extents
//   (already inflated by the build-time rounding margin).
// The two children of node i are 96 contiguous bytes at nodes + 6*i float4.
// ---------------------------------------------------------------------------
constexpr int kNodeFloats = 12;
constexpr int kMaxDepth = 20;  // one pending-bound LDS slot per level and lane (16 M points at 16 per leaf)
constexpr size_t kPtsPad = 8;    // float4 entries allocated (zeroed) past an index's points: leaf scans fetch whole batches

struct BvhView {
  const float4 *nodes;  // (2^(D+1)) * 3 float4
  const float4 *pts;    // n points, w = ORIGINAL index (int bits)
  const float4 *nrm;    // optional normals in the same order (xyz, w = curvature)
  uint32_t n;
  int depth;            // D <= kMaxDepth
  const float4 *axis2;  // optional (device-built indexes): third box axis per node, for the packet walk (scalar fetches
                        // are cheap, the per-lane cross product is not); null -> recomputed
};

// The uniform grid over the target (grid_build.hip): cell id = (iz * dim[1] + iy) * dim[0] + ix with
// i = floor((p - lo) * inv), clamped to the table; gpts = the target points sorted by cell id.
struct GridView {
  const float4 *gpts;              // n (+ kPtsPad) points, w = ORIGINAL index
  const float4 *gnrm;              // optional normals, same order
  const uint32_t *cell_start;      // n_cells + 1
  const uint32_t *gpos_of_bvhpos;  // n: sorted position of the point at BVH position p
  float lo[3];
  float inv;                       // 1 / cell size
  float eps;                       // added to a query radius: covers the rounding of the cell expression (4e-7 x the largest coordinate)
  int dim[3];
  uint32_t n_cells, occupied;
};

// Device-side view of a (Morton-sorted) query cloud.
struct CloudView {
  const float4 *xyzw;   // n points, sorted; w = ORIGINAL index (int bits); non-finite points last
  const float4 *nrm;    // optional, same order
  uint32_t n;           // all points
  uint32_t n_valid;     // finite points (a prefix of the sorted order)
};

// Per-run ICP state living in device memory (one per context).
struct IcpState {
  double F[16];        // final_transformation_, column-major, fp64
  double Tk[16];       // last incremental transformation_
  float Ff[12];        // F rounded to fp32: rows of the 3x4 [R|t] (r00 r01 r02 tx, …)
  float Finv[12];      // inverse of F, same layout (reciprocal correspondences query the SOURCE index with it)
  double S[44];        // reduced sums of the current iteration (17, or 44 with point-to-plane)
  double Vwarm[9];     // right singular vectors of the previous iteration's covariance (warm start of the Jacobi sweeps)
  int have_Vwarm, pad_;
  double pivot[3];
  double prev_mse, cur_mse;
  double rotation_threshold, translation_threshold;
  double mse_threshold_relative, mse_threshold_absolute;
  double max_d2;       // max_corr_dist^2 (fp64, compared against fp32 d2 widened)
  double max_corr_dist;
  double surface_normal_thr, self_occluded_thr;
  long long n_corr;
  int max_iterations, failure_after_max_iter, min_correspondences;
  int iterations, converged, state, done;
  int corr_mode, k_normal_shooting, use_surface_normal_rej, use_self_occluded_rej, use_reciprocal, estimator;
  int comm_error;      // set by the peer-to-peer exchange when a peer's sums did not arrive in time (run ends)
  int chain_error;     // set by an overlapped update launch whose accumulate launch did not report within the bounded wait (run ends)
  // k-NN runs (normal shooting): the transform the previous accumulate launch searched with, so that a query's k-th
  // neighbour distance of that launch plus its own displacement since bounds this launch's search (icp_accumulate_kernel)
  float Fprev[12];
  int have_prev;       // Fprev and the stored k-th distances belong together (set by the update that follows an accumulate launch)
  int knn_acc_flag;    // written by a k-NN accumulate launch: "the stored k-th distances are of the current transform"
  // Skip certificates of the plain 1-NN search (icp_kernels.hip, "skip certificates"; round 4).
  int cert_mode;       // accumulate launches keep and use per-query certificates (sticky: set by the update step once an iteration moves the scene by less than cert_thr)
  float cert_thr;      // metres; < 0: never, +inf: from the first launch on
  float cert_cap;      // metres: the slack a certificate is credited with at most (~ the target's point spacing)
  float cert_k;        // square metres: slack of a certificate ~ cert_k / (distance of the query from the surface)
  float src_c[3];      // centre of the source cloud's bounding box (its own frame)
  float src_r;         // half its diagonal
  float last_move;     // largest displacement of a source point by the last update (estimate; what cert_thr is compared with and a certificate's worth is weighed against)
  uint32_t *host_cert; // host-visible word (pinned) the update step sets when it sets cert_mode: from then on the host launches the certifying instantiation
};
static_assert(sizeof(IcpState) % 8 == 0, "IcpState holds doubles");

// Peer-to-peer exchange of the sums (comm.cpp, icp_p2p_update_kernel): every rank owns one slot per parity in every
// rank's buffer; a slot is 2 * kP2pMaxSums 8-byte words {low: 32 data bits, high: sequence number of the exchange}.
constexpr int kP2pMaxRanks = 8;
constexpr int kP2pMaxSums = 96;   // 17 / 44 sums of the SVD / LLS estimators, 91 of the LM estimator
constexpr int kP2pSlotWords = 2 * kP2pMaxSums;
constexpr size_t kP2pBufferBytes = sizeof(unsigned long long) * 2 * kP2pMaxRanks * kP2pSlotWords;
struct P2pView {
  unsigned long long *buf[kP2pMaxRanks];   // buf[r]: rank r's buffer as mapped here (buf[rank] is this rank's own)
  int nranks, rank;
};

constexpr float kOctSlotShare = 0.42f;   // one of a group-walked chunk's eight slots lasts about this share of the chunk's per-lane duration (tools/chain_probe.py: 0.39-0.53)
constexpr float kCertWorth = 24.0f;     // a query builds a certificate when the slack it can expect is worth this many launches of the scene's current displacement
constexpr int kCertCand = 5;            // skip certificates: candidates kept per query (built by a (kCertCand + 1)-nearest walk)
constexpr int kNumSums = 17;
constexpr int kNumSumsMax = 44;  // + 21 (upper triangle of AᵀA) + 6 (Aᵀb) for the point-to-plane estimator
constexpr int kAccBlock = 512;       // threads per block of the accumulate kernel
// Launch bound of the plain 1-NN accumulate kernel, in waves per SIMD.  At 8 (64 VGPRs) the kernel spilled ~20
// VGPRs to scratch (~25 MB of scratch writes per C3 launch); 6 (80 VGPRs) has no spills and is faster although
// only 768 of the 1024 blocks are resident at a time (C3: 230 -> 197 us, C2: 78 -> 68 us).
constexpr int kAccWavesPerSimd = 6;
constexpr int kAccMaxBlocks = 1024;  // partials rows; the update kernel reduces them with 1024 threads

}  // namespace ope

// std::vector without the zero fill on resize (the host copies of multi-million-point clouds are written in full
// right after they are sized)
template <class T>
struct default_init_allocator : std::allocator<T> {
  template <class U> struct rebind { using other = default_init_allocator<U>; };
  using std::allocator<T>::allocator;
  template <class U, class... Args> void construct(U *p, Args &&...args) {
    if constexpr (sizeof...(Args) == 0) ::new (static_cast<void *>(p)) U;
    else ::new (static_cast<void *>(p)) U(std::forward<Args>(args)...);
  }
};

struct ope_ctx {
  int device = -1;
  int n_cu = 256;   // compute units of the device (MI355X: 256)
  int n_xcd = 8;    // XCDs of the device (hipDeviceAttributeNumberOfXccs; MI355X: 8)
  uint32_t wait_ticks = 200000000u;   // bound of every device-side wait of the overlapped update launches, 100 MHz ticks (ope_ctx_set_wait_limit; 2 s)
  hipStream_t own_stream = nullptr;
  hipStream_t stream = nullptr;
  std::string err;

  // ICP run state
  ope::IcpState *d_state = nullptr;
  double *d_partials = nullptr;   // [kNumSumsMax][kAccMaxBlocks]
  double *d_lm_stats = nullptr;   // 96 doubles: the 91 sums of the LM estimator (lm.hip)
  uint32_t *d_work_counter = nullptr;  // small device scratch block: word 8 = number of chunks walked by 8-lane groups
  int32_t *d_corr_match = nullptr;  // per sorted query: ORIGINAL target index or -1
  float *d_corr_d2 = nullptr;
  // cost-aware chunk schedule of the accumulate kernel
  uint32_t *d_chunk_cost = nullptr, *d_chunk_cost_sorted = nullptr, *d_chunk_ids = nullptr, *d_chunk_order = nullptr;
  uint32_t *d_slot_list = nullptr;   // the launch's slots in descending order of expected duration (plan_slots_kernel)
  bool slot_list_valid = false;
  void *d_plan_tmp = nullptr;
  size_t plan_tmp_bytes = 0, chunk_cap = 0;
  bool plan_valid = false;
  // The tree kernel's plan is made on a side stream while the next launch runs with the plan before it (api.hip,
  // enqueue_accumulate): two sets of plan outputs, `plan_cur` the one launches read.
  hipStream_t plan_stream = nullptr;
  hipEvent_t ev_acc_done = nullptr, ev_plan_done = nullptr;
  uint32_t *d_cost_snap = nullptr;                          // the costs a plan is made from (copied when the plan starts)
  uint32_t *d_plan_sorted[2] = {nullptr, nullptr}, *d_plan_order[2] = {nullptr, nullptr}, *d_plan_slots[2] = {nullptr, nullptr};
  uint32_t *d_plan_out = nullptr;                           // 2 x 8 words
  int plan_cur = 0;
  bool plan_pending = false, plan_pending_slots = false, plan_cur_slots = false;
  int acc_launches = 0;
  // Update launches overlapped with the accumulate launches (api.hip: ope_icp_iterate; icp_kernels.hip: icp_update_chained_kernel):
  // the update of iteration j is launched on its own stream next to accumulate launch j and waits, on the device, for that
  // launch's blocks; accumulate launch j + 1 follows launch j on the launch stream and its blocks wait for update j's word.
  hipStream_t upd_stream = nullptr;
  hipEvent_t ev_chain_s = nullptr, ev_chain_u = nullptr;
  bool chained = false;          // this run may overlap (decided by ope_icp_begin)
  bool chain_on = false;         // the batch being enqueued does overlap
  bool chain_open = false;       // the update stream holds launches the launch stream has not waited for yet
  bool chain_u_synced = false;   // the update stream has waited for the launch stream since the state was last written there
  int chain_fallbacks = 0;       // runs of this context that resumed in line after a bounded wait ran out (ope_icp_update_fallbacks)
  bool chain_broken = false;     // a bounded wait ran out in an earlier run (or a counter-collecting profiler is attached): runs launch their updates in line
  bool chain_recovering = false; // ope_icp_poll is re-enqueueing, in line, the iterations an overlapped run lost to a bounded wait
  uint32_t chain_seq = 0;        // overlapped accumulate launches of this run so far (= updates published once they are done)
  int64_t kernel_launches[OPE_KERNEL_KINDS] = {0, 0, 0, 0};   // per search kernel, this run (ope_icp_kernel_launches)
  uint32_t *d_hint = nullptr;       // per sorted query: leaf (heap id) of the previous iteration's match, 0 = none
  bool cert_run = false;            // the run in progress may keep skip certificates (ope_icp_begin)
  bool cert_seen = false;           // ... and the device has asked for them (or they were forced): accumulate launches take the certifying instantiation
  // host-visible words (pinned): [0] the number of the accumulate launch that last started (written by the launches), [1] "the update
  // step has set cert_mode".  The host keeps at most kPaceLead launches between what it has enqueued and [0] (pace_wait): a bounded queue,
  // and [1] is noticed a few launches late at most.
  uint32_t *h_pace = nullptr, *d_pace = nullptr;
  uint32_t launch_no = 0;           // accumulate launches enqueued in this run
  bool pace_off = false;            // a wait ran into its limit (a stream held up from outside): no pacing for the rest of the run
  // skip certificates, per sorted query: where the query was when it built its certificate and the lower bound L (metres) that walk
  // proved on its distance to every target point but the candidates {x, y, z, L}; the candidates
  float *d_cert_l = nullptr;        // L: the bound on every non-candidate (cert_q.w holds L2, the bound on everything but the nearest candidate)
  float4 *d_cert_q = nullptr;
  uint32_t *d_cert_pos = nullptr;   // [kCertCand][n]: 1 + position in the index's point order, nearest first; 0: none
  float *d_knn_rk = nullptr;        // k-NN runs: per sorted query the squared distance of the last list entry of the previous launch (+inf: none)
  size_t knn_rk_cap = 0;
  // grid path of the 1-NN search (icp_accumulate_grid_kernel)
  uint32_t *d_ghint = nullptr;      // per sorted query: 1 + position of the previous match in the grid-sorted points, 0 = none
  uint32_t *d_qorder = nullptr;     // query order of the launch: [grid-class queries | tree-class queries]
  unsigned char *d_qclass = nullptr;  // per sorted query: 1 = answered by the grid in the last launch
  void *d_part_tmp = nullptr;
  size_t part_tmp_bytes = 0;
  bool use_grid = false;
  size_t grid_cap = 0;
  uint32_t *d_chunk_keys = nullptr;      // sort keys of the tree chunks (plan step)
  hipEvent_t grid_probe_event = nullptr;  // asynchronous read-back of the grid-class query count
  uint32_t *h_grid_probe = nullptr;       // pinned
  bool grid_probe_pending = false;
  bool grid_auto = false;                 // the run may move between the grid and the tree kernel (ope_index_params.grid == 1)
  bool measuring_flag = false;   // plan_info[4] as last written (enqueue_accumulate)
  int force_plan_at = -1;                 // launch at which the tree kernel re-plans after taking over from the grid kernel
  size_t corr_cap = 0;
  ope::IcpState *h_state = nullptr;  // pinned
  const ope_cloud *run_src = nullptr;   // cleared by ope_cloud_free / ope_index_free of the handle they point at
  const ope_index *run_tgt = nullptr;
  size_t corr_run_n = 0;                 // source size of the run the correspondence buffers belong to
  ope_index *run_src_index = nullptr;  // index over the source (reciprocal correspondences only)
  ope_icp_params run_params{};
  bool run_active = false;
  int acc_blocks = 0;
  int acc_blocks_cert = 0;       // grid of the certifying tree instantiation (fewer blocks per CU)
  int launch_blocks = 0;         // blocks of the accumulate launch enqueued last
  uint32_t chain_tickets = 0;    // tickets the overlapped launches of this run will have taken once they are done
  int64_t n_src_total = 0, n_tgt_total = 0;
  int iters_enqueued = 0;
  double *d_sums_ext = nullptr;   // caller-owned 17-double buffer (e.g. a torch tensor) or null
  // setFixedCorrespondences (ope_icp_set_fixed_correspondences): four float4 per pair {source point, source normal, target point, target normal}
  float4 *d_fixed = nullptr;
  size_t n_fixed = 0, n_fixed_run = 0;   // as set / as taken by the run in progress
  const ope_cloud *fixed_src = nullptr;  // the source cloud they index (checked by ope_icp_begin)
  size_t fixed_tgt_n = 0;                // size of the target cloud they index
  bool fixed_has_nrm = false, fixed_has_src_nrm = false;

  // optional per-launch timing of the accumulate kernel (HIP events on the launch stream)
  bool prof_enabled = false;
  std::vector<hipEvent_t> prof_events;
  size_t prof_used = 0;

  // RCCL (dlopen'ed lazily)
  void *nccl_comm = nullptr;
  int comm_nranks = 1, comm_rank = 0;
  // peer-to-peer slots (comm.cpp): own buffer, the peers' buffers as opened here, whether every rank passed the self-test
  unsigned long long *p2p_mine = nullptr;
  void *p2p_peer[ope::kP2pMaxRanks] = {};
  bool p2p_ok = false;
  bool p2p_broken = false;       // an exchange timed out: the ranks' sequence numbers no longer agree; the communicator must be re-created
  int comm_transport = 0;        // OPE_COMM_AUTO / RCCL / P2P as requested
  uint32_t p2p_seq = 0;          // sequence number of the last exchange (never 0 in a slot)
  double *p2p_scratch = nullptr; // kP2pMaxSums doubles for the self-test

  // per-kernel timing of the coarse-stage / filter kernels (ope_profile_kernels): HIP events on the launch stream
  bool ktime_on = false;
  struct KernelStamp { const char *name; hipEvent_t e0, e1; double bytes; };
  std::vector<KernelStamp> kstamps;
  std::vector<hipEvent_t> kevent_pool;

  double last_fpfh_mean_neighbours = 0.0;

  bool tracing = false;   // roctx ranges around the host side of the path (ope_ctx_set_tracing)

  // scratch
  void *d_scratch = nullptr;
  size_t scratch_bytes = 0;
};

struct ope_cloud {
  ope_ctx *ctx = nullptr;
  size_t n = 0, n_valid = 0;
  float4 *d_xyzw = nullptr;
  float4 *d_nrm = nullptr;
  // Host mirrors (original order xyz; sorted position -> original index).  Clouds made on the device (ope_cloud_concat)
  // materialise them on first use: ensure_host().
  mutable std::vector<float, default_init_allocator<float>> h_xyz;   // original order, n*3
  mutable std::vector<int32_t, default_init_allocator<int32_t>> perm;   // sorted position -> original index
  mutable bool host_valid = true;
  int ensure_host() const;   // api.hip; OPE_OK or an error code (message in ctx)
  float bb_lo[3] = {0, 0, 0}, bb_hi[3] = {0, 0, 0};
  ope::CloudView view() const {
    return ope::CloudView{d_xyzw, d_nrm, (uint32_t)n, (uint32_t)n_valid};
  }
};

struct ope_index {
  ope_ctx *ctx = nullptr;
  size_t n = 0;        // indexed (finite) points
  size_t n_total = 0;  // size of the target cloud it was built from
  int depth = 0;
  float4 *d_nodes = nullptr;
  float4 *d_pts = nullptr;
  float4 *d_nrm = nullptr;
  double pivot[3] = {0, 0, 0};
  float bb_lo[3] = {0, 0, 0}, bb_hi[3] = {0, 0, 0};
  float4 *d_axis2 = nullptr;
  bool tmp_alloc = false;             // the four buffers above come from the stream's cache of temporaries (index_build_tmp)
  hipStream_t alloc_stream = nullptr;
  // uniform grid over the same points (device-built indexes)
  float4 *d_gpts = nullptr, *d_gnrm = nullptr;
  uint32_t *d_cell_start = nullptr, *d_gpos = nullptr;
  ope::GridView grid{};
  bool has_grid = false, want_grid = false;
  std::mutex grid_mutex;   // the lazy grid build (ensure_grid) may be reached from several contexts sharing the index
  int grid_mode = 1;   // ope_index_params.grid: 0 off, 1 automatic (falls back to the tree kernel on clutter-heavy sources), 2 always
  float grid_fill = 0.f;
  int grid_max_cells = 0;
  ope::BvhView view() const { return ope::BvhView{d_nodes, d_pts, d_nrm, (uint32_t)n, depth, d_axis2}; }
};

namespace ope {

// Temporaries of one entry point: blocks from a cache over hipMalloc, one free list per stream (a hipMalloc / hipFree pair of a
// few MB costs ~0.5 ms and the free synchronises the device: the front-end stages were made of them).  A block goes back to
// the list at tmp_free and is handed out again to later work ON THE SAME STREAM, which runs behind the work that used it
// before; ope_ctx_destroy returns its stream's blocks to the device.  Buffers that outlive the call (clouds, indexes) stay
// plain hipMalloc.
// (Round 3 first took these from the device's memory pool, hipMallocAsync / hipFreeAsync.  In a fresh process the first
// use of a newly grown pool block came back ZERO, or partly written, to the kernels behind a completed, synchronised copy into
// it: 4-15 of 16 runs of the C++ facade uploaded a 749-point cloud as zeros, tools/flake_hash.sh.  Memory from hipMalloc
// never did.)
struct TmpCache {
  struct Block { void *p; size_t cap; hipStream_t stream; int device; };
  std::mutex mu;
  std::vector<Block> idle, live;
  size_t idle_bytes = 0;
};
inline TmpCache &tmp_cache() { static TmpCache *c = new TmpCache(); return *c; }   // never destroyed: outlives the HIP runtime's teardown
constexpr size_t kTmpCacheMaxIdleBytes = (size_t)4 << 30;

inline void tmp_trim_locked(TmpCache &c, hipStream_t only_stream, bool all) {
  for (size_t k = 0; k < c.idle.size();) {
    if (all || c.idle[k].stream == only_stream) {
      (void)hipFree(c.idle[k].p);   // (synchronises the device: nothing still reads the block)
      c.idle_bytes -= c.idle[k].cap;
      c.idle[k] = c.idle.back();
      c.idle.pop_back();
    } else ++k;
  }
}

inline hipError_t tmp_malloc(hipStream_t s, void **p, size_t bytes) {
  *p = nullptr;
  bytes = std::max<size_t>(bytes, 256);
  size_t gran = 256;
  while (gran * 16 < bytes) gran <<= 1;                 // sizes in steps of 1/16 .. 1/8 of the request
  const size_t cap = (bytes + gran - 1) / gran * gran;
  int dev = 0;
  (void)hipGetDevice(&dev);
  TmpCache &c = tmp_cache();
  std::lock_guard<std::mutex> lock(c.mu);
  size_t best = c.idle.size();
  for (size_t k = 0; k < c.idle.size(); ++k)
    if (c.idle[k].stream == s && c.idle[k].device == dev && c.idle[k].cap >= cap && c.idle[k].cap <= 2 * cap &&
        (best == c.idle.size() || c.idle[k].cap < c.idle[best].cap))
      best = k;
  TmpCache::Block b{nullptr, cap, s, dev};
  if (best != c.idle.size()) {
    b = c.idle[best];
    c.idle_bytes -= b.cap;
    c.idle[best] = c.idle.back();
    c.idle.pop_back();
  } else {
    hipError_t e = hipMalloc(&b.p, cap);
    if (e != hipSuccess) {   // out of memory: give the idle blocks back and try once more
      (void)hipGetLastError();
      tmp_trim_locked(c, nullptr, true);
      e = hipMalloc(&b.p, cap);
      if (e != hipSuccess) return e;
    }
  }
  c.live.push_back(b);
  *p = b.p;
#ifdef OPE_POISON_TMP   // (A/B build: every temporary starts out as 0xA5 bytes — a read of memory nobody wrote changes results)
  return hipMemsetAsync(b.p, 0xA5, bytes, s);
#else
  return hipSuccess;
#endif
}

inline void tmp_free(hipStream_t, void *p) {
  if (!p) return;
  TmpCache &c = tmp_cache();
  std::lock_guard<std::mutex> lock(c.mu);
  for (size_t k = 0; k < c.live.size(); ++k)
    if (c.live[k].p == p) {
      c.idle.push_back(c.live[k]);
      c.idle_bytes += c.live[k].cap;
      c.live[k] = c.live.back();
      c.live.pop_back();
      if (c.idle_bytes > kTmpCacheMaxIdleBytes) tmp_trim_locked(c, nullptr, true);
      return;
    }
  (void)hipFree(p);   // not one of ours
}

// a stream is going away: its idle blocks go back to the device
inline void tmp_release_stream(hipStream_t s) {
  TmpCache &c = tmp_cache();
  std::lock_guard<std::mutex> lock(c.mu);
  tmp_trim_locked(c, s, false);
}

int set_err(ope_ctx *ctx, int code, const std::string &msg);

// Host -> device copy from ANY host memory (the caller's arrays, vectors, stack variables); the SOURCE may be reused when it returns
// (it has been copied into pinned memory), the device sees the data in stream order: through a pinned staging block and a
// stream-ordered DMA, so that no entry point depends on what hipMemcpyAsync does with pageable
// memory (staged at call time, pinned in place, or written through the BAR, by size and release).  Introduced while hunting the
// facade's run-to-run differences (round 3, see tmp_malloc for what they were); every upload synchronises soon after anyway.
// Staging is per host thread (contexts driven from different threads do not wait for each other; one thread's calls are serial
// anyway).  Copies of at most kSmall bytes — transform rows, counters, seeds — go through a ring of pinned slots and do NOT
// synchronise: a slot is reused only after the ring has come round, and the copy that used it is waited for then (an event).
constexpr size_t kStageChunk = (size_t)32 << 20, kStageSmall = 4096;
constexpr int kStageRing = 16;
struct UploadStage {
  unsigned char *big = nullptr; size_t cap = 0;
  unsigned char *ring = nullptr; hipEvent_t ev[kStageRing] = {}; bool used[kStageRing] = {}; int next = 0;
};
inline UploadStage &upload_stage() { static thread_local UploadStage st; return st; }
// the calling thread's pinned block, at least min(bytes, kStageChunk) large (for callers that fill it themselves: ope_cloud_upload
// gathers the caller's structs straight into it); *cap_out = its size
inline hipError_t stage_block(size_t bytes, unsigned char **block, size_t *cap_out) {
  UploadStage &st = upload_stage();
  const size_t want = std::min(std::max(bytes, (size_t)65536), kStageChunk);
  if (st.cap < want) {
    if (st.big) (void)hipHostFree(st.big);
    st.big = nullptr; st.cap = 0;
    const hipError_t e = hipHostMalloc((void **)&st.big, want, hipHostMallocPortable);
    if (e != hipSuccess) return e;
    st.cap = want;
  }
  *block = st.big;
  *cap_out = st.cap;
  return hipSuccess;
}
inline hipError_t h2d_copy(hipStream_t stream, void *dst, const void *src, size_t bytes) {
  constexpr size_t kSmall = kStageSmall;
  constexpr int kRing = kStageRing;
  UploadStage &st = upload_stage();
  if (bytes == 0) return hipSuccess;
  if (bytes <= kSmall) {
    if (st.ring == nullptr) {
      hipError_t e = hipHostMalloc((void **)&st.ring, kSmall * kRing, hipHostMallocPortable);
      if (e != hipSuccess) { st.ring = nullptr; return e; }
      for (int k = 0; k < kRing; ++k)
        if ((e = hipEventCreateWithFlags(&st.ev[k], hipEventDisableTiming)) != hipSuccess) return e;
    }
    const int k = st.next;
    st.next = (k + 1) % kRing;
    if (st.used[k]) { const hipError_t e = hipEventSynchronize(st.ev[k]); if (e != hipSuccess) return e; }
    std::memcpy(st.ring + (size_t)k * kSmall, src, bytes);
    hipError_t e = hipMemcpyAsync(dst, st.ring + (size_t)k * kSmall, bytes, hipMemcpyHostToDevice, stream);
    if (e == hipSuccess) e = hipEventRecord(st.ev[k], stream);
    st.used[k] = e == hipSuccess;
    return e;
  }
  {
    unsigned char *blk; size_t cap;
    const hipError_t e = stage_block(bytes, &blk, &cap);
    if (e != hipSuccess) return e;
  }
  for (size_t off = 0; off < bytes; off += st.cap) {
    const size_t c = std::min(st.cap, bytes - off);
    std::memcpy(st.big, static_cast<const unsigned char *>(src) + off, c);
    hipError_t e = hipMemcpyAsync(static_cast<unsigned char *>(dst) + off, st.big, c, hipMemcpyHostToDevice, stream);
    if (e == hipSuccess) e = hipStreamSynchronize(stream);
    if (e != hipSuccess) return e;
  }
  return hipSuccess;
}

#ifdef OPE_DEVELOPER
// developer probe (OPE_DUMP_HASH=1): FNV-1a checksum of a host or device buffer to stderr
inline void dev_dump_hash(const char *what, const void *p, size_t bytes, bool device) {
  static const bool on = std::getenv("OPE_DUMP_HASH") != nullptr;
  if (!on) return;
  std::vector<unsigned char> h(bytes);
  if (bytes && p) { if (device) (void)hipMemcpy(h.data(), p, bytes, hipMemcpyDeviceToHost); else std::memcpy(h.data(), p, bytes); }
  unsigned long long x = 1469598103934665603ull;
  if (p) for (unsigned char c : h) { x ^= c; x *= 1099511628211ull; }
  std::fprintf(stderr, "[hash] %s bytes %zu %016llx\n", what, bytes, x);
  if (const char *dir = std::getenv("OPE_DUMP_DIR")) {   // and the buffer itself, numbered in call order
    static int seq = 0;
    char path[512];
    std::snprintf(path, sizeof path, "%s/%03d_%zu.bin", dir, seq++, bytes);
    if (bytes <= 65536) if (FILE *f = std::fopen(path, "wb")) { std::fwrite(h.data(), 1, bytes, f); std::fclose(f); }
  }
}
#define OPE_DUMP_HASH(WHAT, P, BYTES, DEVICE) ope_dump_hash_fn(WHAT, P, BYTES, DEVICE)
inline void ope_dump_hash_fn(const char *w, const void *p, size_t b, bool d) { dev_dump_hash(w, p, b, d); }
#else
#define OPE_DUMP_HASH(WHAT, P, BYTES, DEVICE) ((void)0)
#endif

// RAII HIP-event bracket around ONE kernel launch on the context stream, recorded under `name` together with the
// launch's algorithmic bytes (SURVEY.md §8d formulas) when ope_profile_kernels(ctx, 1) is on; otherwise a no-op.
class KernelTimer {
 public:
  KernelTimer(ope_ctx *ctx, const char *name, double algorithmic_bytes, bool start_now = true);
  ~KernelTimer() { stop(); }
  KernelTimer(const KernelTimer &) = delete;
  KernelTimer &operator=(const KernelTimer &) = delete;
  void start();
  void stop();
  void set_bytes(double b);   // for byte counts only known after the launch (FPFH: the measured neighbour count)
 private:
  ope_ctx *ctx_;
  const char *name_;
  double bytes_;
  int slot_;
  bool open_;
};

// RAII roctx range (trace.cpp); a no-op unless ope_ctx_set_tracing(ctx, 1) was called and a roctx library loads.
class TraceRange {
 public:
  TraceRange(const ope_ctx *ctx, const char *name);
  ~TraceRange();
  TraceRange(const TraceRange &) = delete;
  TraceRange &operator=(const TraceRange &) = delete;
 private:
  bool on_;
};

#define OPE_HIP(ctx, call)                                                                       \
  do {                                                                                           \
    hipError_t e__ = (call);                                                                     \
    if (e__ != hipSuccess)                                                                       \
      return ::ope::set_err((ctx), OPE_EHIP, std::string(#call) + ": " + hipGetErrorString(e__)); \
  } while (0)

// Host-side index build (bvh_build.cpp).  pts: n*3 floats (finite only), ids: their ORIGINAL
// indices, nrm optional n*3.  Outputs host arrays ready for upload.
struct HostBvh {
  int depth = 0;
  std::vector<float> nodes;   // (2^(D+1))*kNodeFloats
  std::vector<float> pts4;    // n*4 (x,y,z, original index bits)
  std::vector<float> nrm4;    // n*4 or empty
};
// sampling.hip: a new cloud from n_sel ORIGINAL indices (device array) of a device-resident cloud
int select_cloud_device(ope_ctx *ctx, const ope_cloud *cloud, const int32_t *d_idx, size_t n_sel, ope_cloud **out);
// the same for order-preserving filters: keep = one byte per ORIGINAL index (device); no re-sort (sampling.hip)
int compact_cloud_device(ope_ctx *ctx, const ope_cloud *cloud, const unsigned char *d_keep, ope_cloud **out, int32_t *d_idx_out, size_t *n_out);

void build_bvh_host(const float *xyz, const int32_t *ids, const float *nrm, size_t n, int leaf_size, HostBvh &out);
// api.hip: ope_index_build's body; temporary = the index lives inside one entry point and its buffers come from (and go back to) the
// stream's cache of temporaries instead of hipMalloc / hipFree
int index_build_impl(ope_ctx *ctx, const ope_cloud *target, const ope_index_params *params, bool temporary, ope_index **out);
inline int index_build_tmp(ope_ctx *ctx, const ope_cloud *target, const ope_index_params *params, ope_index **out) {
  return index_build_impl(ctx, target, params, true, out);
}

}  // namespace ope
