// api.hip — the C ABI of libope_hip.so (see include/ope.h for what each entry point replaces).
#include <algorithm>
#include <cfloat>
#include <cmath>
#include <cstring>
#include <limits>
#include <numeric>

#include <chrono>
#include <thread>

#include "ope_internal.hpp"

namespace ope {

// launchers defined in icp_kernels.hip
void launch_icp_accumulate(hipStream_t, int, int, bool, bool, const CloudView &, const BvhView &, const BvhView &,
                           const IcpState *, double *, int32_t *, float *, uint32_t *, uint32_t *, const uint32_t *,
                           uint32_t *, const uint32_t *, bool, int, double *, const uint32_t *, float *, const uint32_t *, hipEvent_t, hipEvent_t, bool,
                           uint32_t *, uint32_t, float4 *, uint32_t *, uint32_t *, uint32_t, uint32_t, float *);
void plan_heavy(hipStream_t, const uint32_t *, uint32_t, float, float, uint32_t, uint32_t *);
void plan_slots(hipStream_t, const uint32_t *, uint32_t, const uint32_t *, uint32_t *);
int icp_accumulate_blocks_per_cu(bool, bool, bool);
int icp_accumulate_cert_blocks_per_cu(bool, bool, bool);
extern bool g_plan_no_alone;
int chunk_plan(hipStream_t, const uint32_t *, uint32_t *, const uint32_t *, uint32_t *, uint32_t, void *, size_t &);
void fill_iota(hipStream_t, uint32_t *, uint32_t);
hipError_t morton_order_device(hipStream_t, const float *, size_t, const float[3], const float[3], float4 *, int32_t *);
hipError_t concat_device(hipStream_t, const CloudView &, const float *, const CloudView &, float *, float[3], float[3]);
hipError_t build_bvh_device(hipStream_t, const float4 *, const float4 *, size_t, int, const float[3], const float[3], int *, float4 **,
                            float4 **, float4 **, float4 **, bool);
void launch_icp_accumulate_grid(hipStream_t, int, bool, const CloudView &, const BvhView &, const GridView &, const IcpState *, double *, int32_t *,
                                float *, uint32_t *, uint32_t *, const uint32_t *, unsigned char *, const uint32_t *, uint32_t *, const uint32_t *,
                                double *, hipEvent_t, hipEvent_t, bool, uint32_t *, uint32_t, uint32_t *, uint32_t, uint32_t, float4 *, uint32_t *, uint32_t *, float *);
int grid_plan(hipStream_t, bool, const unsigned char *, uint32_t, uint32_t *, uint32_t *, const uint32_t *, uint32_t *, uint32_t *, const uint32_t *,
              uint32_t *, uint32_t, uint32_t, float, float, void *, size_t);
size_t grid_plan_tmp_bytes(uint32_t, uint32_t);
void grid_count_far(hipStream_t, const float *, uint32_t, float, uint32_t *);
void grid_count_outside(hipStream_t, const CloudView &, const float[12], const float[3], const float[3], float, uint32_t *);
void grid_hints_to_leaves(hipStream_t, const uint32_t *, const uint32_t *, uint32_t, uint32_t, int, uint32_t *, uint32_t *);
hipError_t build_grid_device(hipStream_t, const float4 *, const float4 *, size_t, const float[3], const float[3], double, uint32_t, GridView *,
                             float4 **, float4 **, uint32_t **, uint32_t **);
void launch_icp_reduce_update(hipStream_t, IcpState *, const double *, double *, int, bool, uint32_t *);
void launch_icp_update(hipStream_t, IcpState *, double *, int, const float *);
void launch_icp_update_chained(hipStream_t, IcpState *, int, uint32_t *, uint32_t, uint32_t, uint32_t);
void launch_gather_fixed_pairs(hipStream_t, const CloudView &, const CloudView &, const uint32_t *, uint32_t, float4 *);
void launch_icp_fixed_pairs(hipStream_t, const IcpState *, const float4 *, uint32_t, double *, float2 *);
void launch_lm_stats(hipStream_t, int, const CloudView &, const BvhView &, const IcpState *, const int32_t *, double *);
void launch_icp_lm_update(hipStream_t, IcpState *, double *, double *);
void launch_lm_pos_to_orig(hipStream_t, const BvhView &, int32_t *, uint32_t);
void launch_nn_search(hipStream_t, const CloudView &, const BvhView &, const float *, int32_t *, float *);
void launch_knn_search(hipStream_t, const CloudView &, const BvhView &, const float *, int, int32_t *, float *);
void launch_fitness(hipStream_t, int, const CloudView &, const BvhView &, const float *, double, double *, const uint32_t *);
void launch_seed_hints(hipStream_t, const CloudView &, const BvhView &, const IcpState *, uint32_t *);
void launch_pairs_svd(hipStream_t, const float *, const float *, uint32_t, double *, int, float *);
// comm.cpp
int comm_allreduce_sums(ope_ctx *ctx, double *d_sums, int count);
bool comm_uses_p2p(const ope_ctx *ctx);
int comm_p2p_exchange_update(ope_ctx *ctx, IcpState *d_state, double *d_sums, int nsums);
int comm_p2p_exchange(ope_ctx *ctx, double *d_sums, int nsums);

static thread_local std::string g_global_err;

constexpr double kGridMaxTreeShare = 0.03, kGridMinTreeShare = 0.015;
constexpr size_t kGridMinQueries = 0;
constexpr float kHeavyLoadFactor = 1.5f;   // launches that fill the GPU: group walks for chunks beyond this multiple of a wave's fair share (C3, slots listed by duration: 1.0 183 us, 1.2 166, 1.45 153, 1.7 157; without the list 1.45 was 174)
constexpr float kHeavyMaxChunksPerWave = 1.8f;   // beyond this the launch is throughput-bound: no 8-lane group walks

// Developer A/B switches and sweeps (tools/*.py) read the environment only in builds made with -DOPE_DEVELOPER
// (`make DEVELOPER=1`); the product library's launch path depends on one environment variable only (ROCPROF_COUNTER_COLLECTION,
// ope_ctx_create: a counter-collecting profiler serialises dispatches, so updates are launched in line).
#ifdef OPE_DEVELOPER
static const char *dev_env(const char *name) { return getenv(name); }
#else
static const char *dev_env(const char *) { return nullptr; }
#endif

int set_err(ope_ctx *ctx, int code, const std::string &msg) {
  if (ctx) ctx->err = msg;
  else g_global_err = msg;
  return code;
}

static inline bool finite3(const float *p) { return std::isfinite(p[0]) && std::isfinite(p[1]) && std::isfinite(p[2]); }

// column-major 4x4 -> 12 floats, rows of [R|t]
static void colmajor_to_rows(const float *T, float rows[12]) {
  for (int r = 0; r < 3; ++r)
    for (int c = 0; c < 4; ++c) rows[4 * r + c] = T[4 * c + r];
}

static int ensure_scratch(ope_ctx *ctx, size_t bytes) {
  if (ctx->scratch_bytes >= bytes) return OPE_OK;
  if (ctx->d_scratch) OPE_HIP(ctx, hipFree(ctx->d_scratch));
  ctx->d_scratch = nullptr;
  ctx->scratch_bytes = 0;
  OPE_HIP(ctx, hipMalloc(&ctx->d_scratch, bytes));
  ctx->scratch_bytes = bytes;
  return OPE_OK;
}

// number of fp64 sums of the run in progress: 17, or 44 with the point-to-plane estimator; nothing past them is touched
static int run_nsums(const ope_ctx *ctx) { return ctx->run_params.estimator == OPE_EST_POINT_TO_PLANE_LLS ? kNumSumsMax : kNumSums; }

// A run that cannot continue: the next begin starts from a clean slate (align strength sizes included).
static void abort_run(ope_ctx *ctx) {
  ctx->run_active = false;
  ctx->run_src = nullptr;
  ctx->run_tgt = nullptr;
  ctx->n_src_total = ctx->n_tgt_total = 0;
}

static double *sums_ptr(ope_ctx *ctx) {
  if (ctx->d_sums_ext) return ctx->d_sums_ext;
  return reinterpret_cast<double *>(reinterpret_cast<unsigned char *>(ctx->d_state) + offsetof(IcpState, S));
}

// the three words the overlapped update launches and the accumulate launches meet at (icp_kernels.hip: acc_launch_begin);
// cleared with the rest of the block by ope_icp_begin
static uint32_t *chain_ptr(ope_ctx *ctx) { return ctx->d_work_counter + 32; }

// The launch stream waits for the overlapped updates enqueued so far (no host synchronisation).
static int chain_join(ope_ctx *ctx) {
  if (!ctx->chain_open) return OPE_OK;
  ctx->chain_open = false;
  OPE_HIP(ctx, hipEventRecord(ctx->ev_chain_u, ctx->upd_stream));
  OPE_HIP(ctx, hipStreamWaitEvent(ctx->stream, ctx->ev_chain_u, 0));
  return OPE_OK;
}

// One accumulate launch, optionally bracketed by HIP events on the launch stream (bench.py's roofline leg).
// Sums of an accumulate launch: by default every block adds its 17 (44) partial sums straight into the run's sums
// (fp64 atomics; the update kernel leaves them at zero again), which saves the reduction kernel and one kernel boundary
// per iteration (C3: 181 -> 174 us per step) — the addition order, and with it the last bit of the sums, varies from
// run to run.  OPE_DETERMINISTIC_SUMS=1 keeps one row per block and reduces the rows in a fixed tree instead.
static bool atomic_sums(const ope_ctx *ctx) { return ctx->run_params.deterministic_sums == 0; }

// ---- which 1-NN kernel a run uses (ope_index_params.grid == 1: automatic)
// The grid kernel is the better one while (nearly) every query has a model point within one grid cell; queries beyond that
// (clutter, or a source that is still far from aligned) are long tree walks that the tree kernel schedules better.  The
// share of such queries is counted on the device from the correspondences' d2 at the plan steps, read back asynchronously
// (no host synchronisation) and acted on with hysteresis: to the tree kernel above kGridMaxTreeShare, back to the grid
// kernel below kGridMinTreeShare.  Either kernel is exact: the choice only moves time.
static int grid_probe_issue(ope_ctx *ctx, hipStream_t stream = nullptr) {
  if (ctx->grid_probe_pending || !ctx->grid_probe_event || !ctx->grid_auto) return OPE_OK;
  if (stream == nullptr) stream = ctx->stream;
  // "far": further than eight cells from the target — clutter, whose ball no 27-cell scan will ever cover; a source that is
  // merely a few cells off at the start of a run is not counted (round 2 counted beyond ONE cell and therefore had to wait
  // for the loop to settle before it could trust the count)
  const float cell = 8.0f / ctx->run_tgt->grid.inv;
  grid_count_far(stream, ctx->d_corr_d2, (uint32_t)ctx->run_src->n_valid, cell * cell, ctx->d_work_counter + 8);
  OPE_HIP(ctx, hipMemcpyAsync(ctx->h_grid_probe, ctx->d_work_counter + 10, 4, hipMemcpyDeviceToHost, stream));
  OPE_HIP(ctx, hipEventRecord(ctx->grid_probe_event, stream));
  ctx->grid_probe_pending = true;
  return OPE_OK;
}
static float heavy_load_factor() {
  static const bool once = [] { g_plan_no_alone = dev_env("OPE_NO_ALONE") != nullptr; return true; }();  // developer A/B switch
  (void)once;
  static const float f = [] { const char *e = dev_env("OPE_HEAVY_LOAD"); return e ? (float)atof(e) : kHeavyLoadFactor; }();  // developer sweep
  return f;
}
// returns +1: switch to the grid kernel, -1: switch to the tree kernel, 0: stay
static int grid_probe_poll(ope_ctx *ctx, int it_done) {
  if (!ctx->grid_probe_pending) return 0;
  // The first count of a run (taken after launch 0) is WAITED for before launch 2 is enqueued: one short wait per run, and
  // which kernel serves which launch no longer depends on how far the host runs ahead of the GPU.  Later counts (every
  // plan step from launch 8 on, both directions) are polled without waiting.
  if (it_done == 2) { if (hipEventSynchronize(ctx->grid_probe_event) != hipSuccess) return 0; }
  else if (hipEventQuery(ctx->grid_probe_event) != hipSuccess) return 0;
  ctx->grid_probe_pending = false;
  const uint32_t n_far = *ctx->h_grid_probe;
  const double share = (double)n_far / (double)std::max<size_t>(ctx->run_src->n_valid, 1);
  if (dev_env("OPE_TRACE_GRID"))
    fprintf(stderr, "[ope] launch %d (%s kernel): %u of %zu queries beyond one cell of the target (share %.4f)\n", it_done,
            ctx->use_grid ? "grid" : "tree", n_far, ctx->run_src->n_valid, share);
  if (ctx->use_grid && share > kGridMaxTreeShare) return -1;
  if (!ctx->use_grid && share < kGridMinTreeShare) return +1;
  return 0;
}
static int switch_kernel(ope_ctx *ctx, bool to_grid, int it_done, uint32_t nch) {
  ctx->use_grid = to_grid;
  ctx->plan_valid = false;   // chunk ids mean something else to the other kernel

  if (ctx->plan_pending) { OPE_HIP(ctx, hipStreamWaitEvent(ctx->stream, ctx->ev_plan_done, 0)); ctx->plan_pending = false; }
  OPE_HIP(ctx, hipMemsetAsync(ctx->d_chunk_cost, 0, 4 * 2 * ctx->chunk_cap, ctx->stream));
  OPE_HIP(ctx, hipMemsetAsync(ctx->d_work_counter + 8, 0, 8, ctx->stream));
  if (to_grid) {
    // query order and classes start over (the hints into the grid may be stale: a stale hint is still a model point, so
    // it only costs the first launch a tree walk)
    fill_iota(ctx->stream, ctx->d_qorder, (uint32_t)std::max<size_t>(ctx->run_src->n, 1));
    OPE_HIP(ctx, hipMemsetAsync(ctx->d_qclass, 0, std::max<size_t>(ctx->run_src->n, 1), ctx->stream));
  }
  if (!to_grid && ctx->run_tgt->has_grid && ctx->d_ghint) {
    // the previous matches the grid kernel kept become the tree kernel's start leaves (grid_build.hip)
    const int rcs = ensure_scratch(ctx, std::max<size_t>(4 * ctx->run_tgt->n + 4096, (size_t)1 << 16));
    if (rcs != OPE_OK) return rcs;
    grid_hints_to_leaves(ctx->stream, ctx->d_ghint, ctx->run_tgt->d_gpos, (uint32_t)ctx->run_src->n_valid, (uint32_t)ctx->run_tgt->n, ctx->run_tgt->depth,
                         ctx->d_hint, reinterpret_cast<uint32_t *>(static_cast<unsigned char *>(ctx->d_scratch) + 4096));
  }
  // the kernel taken over plans as soon as its first launch has measured its chunks (an unplanned launch measures them all)
  ctx->force_plan_at = it_done + 1;
  return OPE_OK;
}

// The host stays at most kPaceLead accumulate launches ahead of the GPU: it reads, without a HIP call, the word every launch
// stores its number into when it starts (pinned memory).  What that buys: a bounded queue, and a device-side decision — the
// update step asking for certifying launches — reaches the launches that are enqueued no more than that many launches late
// (bench.py and ope_icp_run with check_every = 0 enqueue a whole run in one go: a hundred launches in the time the GPU
// takes for ten).  The GPU never waits for the host: kPaceLead launches are a millisecond and more of work.  A wait is bounded
// (a caller's stream may be held up by something the caller does after this call returns): two seconds, then no pacing.
constexpr uint32_t kPaceLead = 16;
static void pace_wait(ope_ctx *ctx) {
  if (ctx->h_pace == nullptr || ctx->pace_off || ctx->launch_no < kPaceLead) return;
  volatile uint32_t *seen = ctx->h_pace;
  if ((int32_t)(ctx->launch_no - *seen) < (int32_t)kPaceLead) return;
  const auto t0 = std::chrono::steady_clock::now();
  while ((int32_t)(ctx->launch_no - *seen) >= (int32_t)kPaceLead) {
    std::this_thread::yield();
    if (std::chrono::steady_clock::now() - t0 > std::chrono::seconds(2)) { ctx->pace_off = true; return; }
  }
}

static int enqueue_accumulate(ope_ctx *ctx, bool atomic_sums = false) {
  pace_wait(ctx);
  if (ctx->cert_run && !ctx->cert_seen && ctx->h_pace != nullptr && *(volatile uint32_t *)(ctx->h_pace + 1) != 0u) ctx->cert_seen = true;
  // re-sort the chunks by the cost they measured: after launches 1, 2, 4, ..., 32 and then every 32
  static const bool no_plan_env = dev_env("OPE_NO_PLAN") != nullptr;  // developer A/B switch
  const int it_done = ctx->acc_launches++;
  const uint32_t nch = (uint32_t)((ctx->run_src->n_valid + 63) / 64);
  // (a k-NN launch with dozens of chunks per resident wave is bound by throughput, its tail is one chunk of forty: the cost-sorted
  // order buys nothing there and its sorts queue up behind the launch — BuildModel's loop, 20 views of 500 k points: 1.50 s with plans,
  // 1.46-1.47 s without)
  const bool no_plan = no_plan_env || ctx->run_params.deterministic_sums != 0 ||
                       (ctx->run_params.corr_mode == OPE_CORR_NORMAL_SHOOTING && nch > 8u * (uint32_t)ctx->n_cu * 4u * 4u && !dev_env("OPE_NS_PLAN_ALWAYS"));
  // Chunk costs are double-buffered by launch parity: launch L writes half L & 1, and a plan made beside launch L (on the side
  // stream) reads the half launch L - 1 wrote — the measuring launch in front of every plan, in which every chunk reports.
  // Launch L + 1, which writes that half again, waits for the plan first (ev_plan_done).  (Round 3 copied the one buffer on the
  // side stream while launch L was rewriting it: the snapshot mixed two launches' costs.)
  uint32_t *const cost_w = ctx->d_chunk_cost + (size_t)(it_done & 1) * ctx->chunk_cap;
  const uint32_t *const cost_r = ctx->d_chunk_cost + (size_t)((it_done + 1) & 1) * ctx->chunk_cap;
  static const int plan_every = [] { const char *e = dev_env("OPE_PLAN_EVERY"); return e ? std::max(1, atoi(e)) : 32; }();  // developer sweep
  {
    // the launch before a plan step measures: no 8-lane group walks (their chunks would keep the cost of the last
    // per-lane walk they had, however old), flagged to the kernels through plan_info[4]
    const int nx = it_done + 1;
    // (round 3: letting a group-walked chunk report an estimate of its per-lane cost instead — the duration of one of its
    // slots over kOctSlotShare — did away with these launches and with a good schedule: steady state 172-186 us against
    // 152-157; a slot lasts 51-60 us whatever its chunk costs per lane.  Measured, not kept.)
    // (round 3, later: plans every 4 or 2 launches between 8 and 32 from the costs at hand, with and without the measuring launch
    // in front of plan 16 — the driver's window stayed at 176-180 us per step in every combination: measured, not kept)
    // (round 4, with the costs double-buffered: a plan after EVERY launch of 5..16, 8..16 or 8..32 — the launches that creep from
    // 150 to 175 us behind a plan then stay at 160-168: kernel 163-167 us against 161-165, value 5 250-5 590 against 5 700-5 810.  The
    // plan is at its best right behind a measuring launch, whose costs are the only complete ones; plan period 6/8/12/16/24: 32 stays best)
    const bool measuring = !no_plan && nch > 1 && ((((nx & (nx - 1)) == 0 && nx <= plan_every) || nx % plan_every == 0 || nx == ctx->force_plan_at));
    ctx->measuring_flag = measuring;   // handed to the launch as a kernel argument (round 2 kept it in device memory: two memset dispatches per plan step on the critical path)
  }
  if (ctx->use_grid) {
    // GRID instantiation (1-NN, no reciprocal check, device-built index).  Plan steps at the same launches as below;
    // the query order is re-partitioned by class at launches 1, 2, 4 and then with every plan step.
    const ope_icp_params &p = ctx->run_params;
    const bool nrm = p.use_surface_normal_rej || p.use_self_occluded_rej || p.estimator == OPE_EST_POINT_TO_PLANE_LLS;
    const bool plan_step = !no_plan && nch > 1 && it_done >= 1 &&
                           (((it_done & (it_done - 1)) == 0 && it_done <= plan_every) || it_done % plan_every == 0 || it_done == ctx->force_plan_at);
    if (grid_probe_poll(ctx, it_done) < 0) {
      const int rcs = switch_kernel(ctx, false, it_done, nch);
      if (rcs != OPE_OK) return rcs;
      --ctx->acc_launches;
      return enqueue_accumulate(ctx, atomic_sums);
    }
    if (plan_step) {
      static const float heavy_env = [] { const char *e = dev_env("OPE_HEAVY_FACTOR"); return e ? (float)atof(e) : -1.0f; }();
      const bool repart = it_done <= 4 || it_done % plan_every == 0 || it_done == ctx->force_plan_at;
      if (grid_plan(ctx->stream, repart, ctx->d_qclass, (uint32_t)ctx->run_src->n_valid, ctx->d_qorder, ctx->d_work_counter + 8, cost_r,
                    ctx->d_chunk_keys, ctx->d_chunk_cost_sorted, ctx->d_chunk_ids, ctx->d_chunk_order, nch,
                    (uint32_t)ctx->acc_blocks * (kAccBlock / 64), heavy_env, heavy_load_factor(), ctx->d_part_tmp, ctx->part_tmp_bytes) != 0)
        return set_err(ctx, OPE_EHIP, "grid plan step failed");
      ctx->plan_valid = true;
      if (it_done == 1 || it_done >= 8) {   // after launch 0 (decided before launch 2, see grid_probe_poll), then with the later plans
        const int rcp = grid_probe_issue(ctx);
        if (rcp != OPE_OK) return rcp;
      }
    }
    const bool timed = ctx->prof_enabled && ctx->prof_used < ctx->prof_events.size() / 2;
    const bool certify = ctx->cert_run && ctx->cert_seen;
    ctx->launch_blocks = certify ? std::min(ctx->acc_blocks, ctx->acc_blocks_cert) : ctx->acc_blocks;   // (the certifying instantiation holds fewer blocks per CU)
    ++ctx->kernel_launches[OPE_KERNEL_GRID];
    launch_icp_accumulate_grid(ctx->stream, ctx->launch_blocks, nrm, ctx->run_src->view(), ctx->run_tgt->view(), ctx->run_tgt->grid, ctx->d_state,
                               ctx->d_partials, ctx->d_corr_match, ctx->d_corr_d2, ctx->d_hint, ctx->d_ghint, ctx->d_qorder, ctx->d_qclass,
                               ctx->plan_valid ? ctx->d_chunk_order : nullptr, cost_w, ctx->d_work_counter + 8,
                               atomic_sums ? sums_ptr(ctx) : nullptr, timed ? ctx->prof_events[2 * ctx->prof_used] : nullptr,
                               timed ? ctx->prof_events[2 * ctx->prof_used + 1] : nullptr, ctx->measuring_flag,
                               ctx->chain_on ? chain_ptr(ctx) : nullptr, ctx->chain_seq, ctx->d_pace, ++ctx->launch_no, ctx->wait_ticks,
                               certify ? ctx->d_cert_q : nullptr, ctx->d_cert_pos, ctx->d_work_counter + 40, ctx->d_cert_l);
    if (timed) ++ctx->prof_used;
    return OPE_OK;
  }
  if (ctx->grid_auto && grid_probe_poll(ctx, it_done) > 0) {
    const int rcs = switch_kernel(ctx, true, it_done, nch);
    if (rcs != OPE_OK) return rcs;
    --ctx->acc_launches;
    return enqueue_accumulate(ctx, atomic_sums);
  }
  // ---- the tree kernel's plan.  It is made on a side stream from a copy of the costs the launches before this one measured,
  // while THIS launch still runs with the plan before it; the next launch waits for it (an event, no host synchronisation)
  // and switches over.  Round 2 sorted between two launches: ~40 us on the critical path per plan step, twice inside the
  // driver's twenty-step window.
  if (ctx->plan_pending) {
    OPE_HIP(ctx, hipStreamWaitEvent(ctx->stream, ctx->ev_plan_done, 0));
    ctx->plan_cur ^= 1;
    ctx->plan_cur_slots = ctx->plan_pending_slots;
    ctx->plan_valid = true;
    ctx->plan_pending = false;
  }
  static const bool plan_sync_env = dev_env("OPE_PLAN_SYNC") != nullptr;     // developer A/B: the plan between two launches, as in round 2
  // plan steps: after launches 1, 2, 4, ..., 32 and then every 32 (each follows a measuring launch, which runs without group
  // walks and is ~40 us slower: more frequent plans cost more than they gain)
  const bool plan_step = !no_plan && nch > 1 && it_done >= 1 &&
                         (((it_done & (it_done - 1)) == 0 && it_done <= plan_every) || it_done % plan_every == 0 || it_done == ctx->force_plan_at);
  if (plan_step) {
    // Chunks costlier than `factor` x the median chunk are walked by 8-lane groups: a third of the dependent trips for
    // ~2.7x the lane-cycles.  That trade pays while the launch is bound by its slowest wave, i.e. while there are few
    // chunks per resident wave; once the waves are busy for several rounds the extra lane-cycles only lengthen the launch.
    // Measured (tools/heavy_sweep4.py, model 100 k, steady state, kernel us at factor none / 2 / 3 / 5):
    //   0.25 chunks per wave (C2)  98 / 62 / 62 / 80      0.3 (a 1/8 shard) 130 / 70 / 69 / 100      0.6  154 / 79 / 99 / 139
    //   1.3 (500 k queries)       168 / 138 / 132 / 156   2.5 (C3)          177 / 261 / 232 / 254  <- none is best
    static const float heavy_env = [] { const char *e = dev_env("OPE_HEAVY_FACTOR"); return e ? (float)atof(e) : -1.0f; }();
    const float chunks_per_wave = (float)nch / (float)(ctx->n_cu * 4 * kAccWavesPerSimd);
    const float heavy_factor = heavy_env >= 0.f ? heavy_env
                               : (chunks_per_wave > kHeavyMaxChunksPerWave ? 0.0f : std::min(7.0f, std::max(2.0f, 1.2f + 1.5f * chunks_per_wave)));
    const float load_factor = (heavy_env < 0.f && chunks_per_wave > kHeavyMaxChunksPerWave) ? heavy_load_factor() : 0.0f;
    // (a one-launch plan — costs bucketed at 16 per octave, counting sort, same rules on the buckets — took 2.5 us per
    // iteration off the driver's twenty-step window and put 5-9 us on its search kernel: the coarser order is the worse
    // schedule while the costs still move; a one-block rocprim::block_radix_sort of the 15 625 keys took 63 us on its single
    // CU: both measured in round 2, not kept)
    const bool async = !plan_sync_env;
    hipStream_t ps = async ? ctx->plan_stream : ctx->stream;
    const int nxt = ctx->plan_cur ^ 1;
    const uint32_t *costs = cost_r;
    if (async) {
      OPE_HIP(ctx, hipEventRecord(ctx->ev_acc_done, ctx->stream));       // every launch before this one has finished ...
      OPE_HIP(ctx, hipStreamWaitEvent(ps, ctx->ev_acc_done, 0));         // ... before their costs are read (this launch writes the other half)
    }
    // the count of far queries that may move the run back to the grid kernel rides on the same side stream (three dispatches
    // off the launch stream; it reads the distances while this launch rewrites them: a count of two consecutive launches'
    // values, for a decision that is polled without waiting anyway)
    if (ctx->grid_auto && it_done >= 8) {
      const int rcp = grid_probe_issue(ctx, ps);
      if (rcp != OPE_OK) return rcp;
    }
    size_t tb = ctx->plan_tmp_bytes;
    if (chunk_plan(ps, costs, ctx->d_plan_sorted[nxt], ctx->d_chunk_ids, ctx->d_plan_order[nxt], nch, ctx->d_plan_tmp, tb) != 0)
      return set_err(ctx, OPE_EHIP, "chunk plan sort failed");
    uint32_t *out = ctx->d_plan_out + 8 * nxt;
    plan_heavy(ps, ctx->d_plan_sorted[nxt], nch, heavy_factor, load_factor, (uint32_t)ctx->acc_blocks * (kAccBlock / 64), out);
    static const bool no_slot_list = dev_env("OPE_NO_SLOT_LIST") != nullptr;  // developer A/B switch
    const bool slots = !no_slot_list && ctx->run_params.corr_mode == OPE_CORR_NEAREST && !ctx->run_params.use_reciprocal;
    if (slots) plan_slots(ps, ctx->d_plan_sorted[nxt], nch, out, ctx->d_plan_slots[nxt]);
    if (async) {
      OPE_HIP(ctx, hipEventRecord(ctx->ev_plan_done, ps));
      ctx->plan_pending = true;
      ctx->plan_pending_slots = slots;
    } else {
      ctx->plan_cur = nxt;
      ctx->plan_cur_slots = slots;
      ctx->plan_valid = true;
    }
  }
  const ope_icp_params &p = ctx->run_params;
  const bool nrm = p.corr_mode == OPE_CORR_NORMAL_SHOOTING || p.use_surface_normal_rej || p.use_self_occluded_rej ||
                   p.estimator == OPE_EST_POINT_TO_PLANE_LLS;   // (LM reads the target normals in its own kernel)
  const bool timed = ctx->prof_enabled && ctx->prof_used < ctx->prof_events.size() / 2;
  const bool recip = p.use_reciprocal != 0;
  // packet walks for coherent chunks pay off when the launch fills the GPU (C3: 195 -> 184 us); on an underfilled
  // one (a 1/8 shard, C2) the longer dependent chain of a packet costs more than its gathers save (77 -> 88 us)
  static const bool no_packet = dev_env("OPE_NO_PACKET") != nullptr;  // developer A/B switch
  static const int packet_min_env = [] { const char *e = dev_env("OPE_PACKET_MIN_CHUNKS"); return e ? atoi(e) : -1; }();  // developer sweep
  const uint32_t packet_min = packet_min_env >= 0 ? (uint32_t)packet_min_env : (uint32_t)ctx->n_cu * 4u * (uint32_t)kAccWavesPerSimd;
  const bool packet = p.tree_walk == OPE_WALK_PACKET ? (ctx->run_tgt->d_axis2 != nullptr && p.corr_mode == OPE_CORR_NEAREST && !recip)
                      : p.tree_walk == OPE_WALK_LANE ? false : (!no_packet && nch > packet_min);
  ++ctx->kernel_launches[p.corr_mode != OPE_CORR_NEAREST ? OPE_KERNEL_KNN : (packet && !recip) ? OPE_KERNEL_TREE_PACKET : OPE_KERNEL_TREE_LANE];
  const bool certify = ctx->cert_run && ctx->cert_seen && p.corr_mode == OPE_CORR_NEAREST && !recip;
  const int blocks = certify ? std::min(ctx->acc_blocks, ctx->acc_blocks_cert) : ctx->acc_blocks;   // (the certifying instantiation holds fewer blocks per CU)
  ctx->launch_blocks = blocks;
  launch_icp_accumulate(ctx->stream, blocks, p.corr_mode, nrm, recip, ctx->run_src->view(), ctx->run_tgt->view(),
                        recip ? ctx->run_src_index->view() : ctx->run_tgt->view(), ctx->d_state, ctx->d_partials, ctx->d_corr_match, ctx->d_corr_d2, ctx->d_work_counter, ctx->d_hint,
                        ctx->plan_valid ? ctx->d_plan_order[ctx->plan_cur] : nullptr, cost_w, ctx->d_work_counter + 8, packet, p.k_normal_shooting, atomic_sums ? sums_ptr(ctx) : nullptr,
                        (ctx->plan_valid && ctx->plan_cur_slots) ? ctx->d_plan_slots[ctx->plan_cur] : nullptr,
                        (p.corr_mode == OPE_CORR_NORMAL_SHOOTING && !dev_env("OPE_NO_KNN_BOUND")) ? ctx->d_knn_rk : nullptr, ctx->d_plan_out + 8 * ctx->plan_cur,
                        timed ? ctx->prof_events[2 * ctx->prof_used] : nullptr, timed ? ctx->prof_events[2 * ctx->prof_used + 1] : nullptr,
                        ctx->measuring_flag, ctx->chain_on ? chain_ptr(ctx) : nullptr, ctx->chain_seq, certify ? ctx->d_cert_q : nullptr, ctx->d_cert_pos, ctx->d_pace, ++ctx->launch_no, ctx->wait_ticks, ctx->d_cert_l);
  if (timed) ++ctx->prof_used;
  return OPE_OK;
}

// First dispatch of a stream at context creation (ope_ctx_create).  It asks for more private memory per lane than any kernel that
// runs under a bounded device-side wait (icp_update_chained_kernel 48 bytes, the accumulate instantiations 16-44): the runtime
// gives a hardware queue its scratch memory when a dispatch first needs it, with the queue stalled until its helper thread has
// allocated it — milliseconds, or seconds when the driver is busy taking back what earlier processes held.
__global__ __launch_bounds__(256) void prime_stream_kernel(uint32_t *word, int n) {
  volatile uint32_t buf[64];
  for (int i = 0; i < 64; ++i) buf[i] = (uint32_t)(i * n) + threadIdx.x;
  uint32_t acc = 0u;
  for (int i = 0; i < 64; ++i) acc += buf[(i * 7 + n) & 63];   // (run-time indices: the array stays in private memory)
  if (blockIdx.x == 0 && threadIdx.x == 0) *word = acc | 1u;
}

}  // namespace ope

using namespace ope;

extern "C" {

int ope_abi_version(void) { return OPE_ABI_VERSION; }

int ope_device_count(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) return 0;
  return n;
}

int ope_ctx_create(ope_ctx **out, int device_ordinal) {
  if (!out) return OPE_EINVAL;
  *out = nullptr;
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess || n <= 0)
    return set_err(nullptr, OPE_ENODEV, "no HIP device: libope_hip.so is GPU-only and has no CPU fallback");
  if (device_ordinal < 0 || device_ordinal >= n) return set_err(nullptr, OPE_ENODEV, "bad device ordinal");
  ope_ctx *ctx = new ope_ctx();
  ctx->device = device_ordinal;
  {
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device_ordinal) == hipSuccess && prop.multiProcessorCount > 0) ctx->n_cu = prop.multiProcessorCount;
    int nx = 0;   // XCDs: blocks are dealt to them round-robin (8 on an MI355X in SPX mode; a partitioned device reports its own)
    if (hipDeviceGetAttribute(&nx, hipDeviceAttributeNumberOfXccs, device_ordinal) == hipSuccess && nx > 0 && nx <= 64) ctx->n_xcd = nx;
  }
  if (hipSetDevice(device_ordinal) != hipSuccess || hipStreamCreate(&ctx->own_stream) != hipSuccess) {
    delete ctx;
    return set_err(nullptr, OPE_EHIP, "hipSetDevice/hipStreamCreate failed");
  }
  ctx->stream = ctx->own_stream;
  // A profiler that collects hardware counters serialises dispatches (rocprofv3 --pmc sets ROCPROF_COUNTER_COLLECTION), and a
  // serialised update launch would wait its 2 s for an accumulate launch that cannot start beside it: such processes launch their
  // updates in line from the start.  The one environment variable the product library looks at.
  if (const char *e = getenv("ROCPROF_COUNTER_COLLECTION"); e != nullptr && *e != 0 && *e != '0' && !dev_env("OPE_CHAIN_UNDER_COUNTERS")) ctx->chain_broken = true;
  if (hipMalloc(&ctx->d_state, sizeof(IcpState)) != hipSuccess ||
      hipMalloc(&ctx->d_partials, sizeof(double) * kNumSumsMax * kAccMaxBlocks) != hipSuccess ||
      hipMalloc((void **)&ctx->d_work_counter, 256) != hipSuccess ||
      hipMalloc((void **)&ctx->d_lm_stats, sizeof(double) * 96) != hipSuccess ||
      hipMalloc((void **)&ctx->d_plan_out, 64) != hipSuccess || hipMemset(ctx->d_plan_out, 0, 64) != hipSuccess ||
      hipStreamCreateWithFlags(&ctx->plan_stream, hipStreamNonBlocking) != hipSuccess ||
      hipStreamCreateWithFlags(&ctx->upd_stream, hipStreamNonBlocking) != hipSuccess ||
      hipEventCreateWithFlags(&ctx->ev_chain_s, hipEventDisableTiming) != hipSuccess ||
      hipEventCreateWithFlags(&ctx->ev_chain_u, hipEventDisableTiming) != hipSuccess ||
      hipEventCreateWithFlags(&ctx->ev_acc_done, hipEventDisableTiming) != hipSuccess ||
      hipEventCreateWithFlags(&ctx->ev_plan_done, hipEventDisableTiming) != hipSuccess ||
      hipHostMalloc((void **)&ctx->h_pace, 64, hipHostMallocMapped) != hipSuccess ||
      hipHostGetDevicePointer((void **)&ctx->d_pace, ctx->h_pace, 0) != hipSuccess ||
      hipHostMalloc((void **)&ctx->h_state, sizeof(IcpState)) != hipSuccess) {
    ope_ctx_destroy(ctx);
    return set_err(nullptr, OPE_ENOMEM, "context allocation failed");
  }
  // The runtime creates a stream's hardware queue at the stream's FIRST dispatch and gives the queue its scratch memory when a
  // dispatch first needs some.  For the update stream both used to happen at update 0 of the context's first overlapped run — with
  // the blocks of accumulate launch 1 already waiting on the device for that update's word, under a bound (seen twice in some
  // forty fresh processes, both right after other processes had left the GPU: the run fell back to in-line launches).  The
  // streams get their queues, with scratch, here instead (prime_stream_kernel).
  {
    int n_cu = 0;
    (void)hipDeviceGetAttribute(&n_cu, hipDeviceAttributeMultiprocessorCount, device_ordinal);
    const dim3 grid((unsigned)std::max(n_cu, 1) * 8u);   // (a launch that could fill the device: the scratch is sized for that)
    hipLaunchKernelGGL(prime_stream_kernel, grid, dim3(256), 0, ctx->plan_stream, ctx->d_work_counter + 60, 3);
    hipLaunchKernelGGL(prime_stream_kernel, grid, dim3(256), 0, ctx->upd_stream, ctx->d_work_counter + 61, 5);
    hipLaunchKernelGGL(prime_stream_kernel, grid, dim3(256), 0, ctx->own_stream, ctx->d_work_counter + 62, 7);
  }
  if (hipStreamSynchronize(ctx->plan_stream) != hipSuccess || hipStreamSynchronize(ctx->upd_stream) != hipSuccess ||
      hipStreamSynchronize(ctx->own_stream) != hipSuccess) {
    ope_ctx_destroy(ctx);
    return set_err(nullptr, OPE_EHIP, "context set-up: the side streams' first dispatch failed");
  }
  *out = ctx;
  return OPE_OK;
}

void ope_ctx_destroy(ope_ctx *ctx) {
  if (!ctx) return;
  (void)hipSetDevice(ctx->device);
  ope_comm_destroy(ctx);
  if (ctx->run_src_index) ope_index_free(ctx->run_src_index);
  for (hipEvent_t e : ctx->prof_events) (void)hipEventDestroy(e);
  for (auto &k : ctx->kstamps) { (void)hipEventDestroy(k.e0); (void)hipEventDestroy(k.e1); }
  for (hipEvent_t e : ctx->kevent_pool) (void)hipEventDestroy(e);
  if (ctx->d_state) (void)hipFree(ctx->d_state);
  if (ctx->d_partials) (void)hipFree(ctx->d_partials);
  if (ctx->d_work_counter) (void)hipFree(ctx->d_work_counter);
  if (ctx->d_lm_stats) (void)hipFree(ctx->d_lm_stats);
  if (ctx->d_fixed) (void)hipFree(ctx->d_fixed);
  tmp_release_stream(ctx->stream);   // the cached temporaries of this context's stream
  if (ctx->plan_stream) { (void)hipStreamSynchronize(ctx->plan_stream); (void)hipStreamDestroy(ctx->plan_stream); }
  if (ctx->upd_stream) { (void)hipStreamSynchronize(ctx->upd_stream); (void)hipStreamDestroy(ctx->upd_stream); }
  if (ctx->ev_chain_s) (void)hipEventDestroy(ctx->ev_chain_s);
  if (ctx->ev_chain_u) (void)hipEventDestroy(ctx->ev_chain_u);
  if (ctx->ev_acc_done) (void)hipEventDestroy(ctx->ev_acc_done);
  if (ctx->ev_plan_done) (void)hipEventDestroy(ctx->ev_plan_done);
  for (void *q : {(void *)ctx->d_plan_out, (void *)ctx->d_cost_snap, (void *)ctx->d_plan_sorted[0], (void *)ctx->d_plan_sorted[1], (void *)ctx->d_plan_order[0],
                  (void *)ctx->d_plan_order[1], (void *)ctx->d_plan_slots[0], (void *)ctx->d_plan_slots[1]})
    if (q) (void)hipFree(q);
  if (ctx->d_corr_match) (void)hipFree(ctx->d_corr_match);
  if (ctx->d_corr_d2) (void)hipFree(ctx->d_corr_d2);
  if (ctx->d_hint) (void)hipFree(ctx->d_hint);
  if (ctx->d_cert_q) (void)hipFree(ctx->d_cert_q);
  if (ctx->d_cert_pos) (void)hipFree(ctx->d_cert_pos);
  if (ctx->d_cert_l) (void)hipFree(ctx->d_cert_l);
  if (ctx->d_knn_rk) (void)hipFree(ctx->d_knn_rk);
  for (void *p : {(void *)ctx->d_ghint, (void *)ctx->d_qorder, (void *)ctx->d_qclass, ctx->d_part_tmp, (void *)ctx->d_chunk_keys})
    if (p) (void)hipFree(p);
  if (ctx->grid_probe_event) (void)hipEventDestroy(ctx->grid_probe_event);
  if (ctx->h_grid_probe) (void)hipHostFree(ctx->h_grid_probe);
  for (void *p : {(void *)ctx->d_chunk_cost, (void *)ctx->d_chunk_cost_sorted, (void *)ctx->d_chunk_ids,
                  (void *)ctx->d_chunk_order, (void *)ctx->d_slot_list, ctx->d_plan_tmp})
    if (p) (void)hipFree(p);
  if (ctx->d_scratch) (void)hipFree(ctx->d_scratch);
  if (ctx->h_state) (void)hipHostFree(ctx->h_state);
  if (ctx->h_pace) (void)hipHostFree(ctx->h_pace);
  if (ctx->own_stream) (void)hipStreamDestroy(ctx->own_stream);
  delete ctx;
}

int ope_ctx_set_stream(ope_ctx *ctx, void *hip_stream) {
  if (!ctx) return OPE_EINVAL;
  hipStream_t next = hip_stream ? (hipStream_t)hip_stream : ctx->own_stream;
  if (next != ctx->stream) {
    // temporaries cached for the stream that is left go back to the device (they are keyed by stream: nothing would reuse or
    // release them once the context has moved on); frees synchronise, so nothing in flight still uses them
    (void)hipSetDevice(ctx->device);
    tmp_release_stream(ctx->stream);
    if (next != ctx->own_stream) {
      // a caller's stream gets what ope_ctx_create gives the context's own: its queue and the queue's scratch memory, before any
      // launch on it runs beside an update that waits for it under a bound (prime_stream_kernel)
      hipLaunchKernelGGL(prime_stream_kernel, dim3((unsigned)std::max(ctx->n_cu, 1) * 8u), dim3(256), 0, next, ctx->d_work_counter + 62, 9);
      OPE_HIP(ctx, hipStreamSynchronize(next));
    }
  }
  ctx->stream = next;
  return OPE_OK;
}

int ope_ctx_set_wait_limit(ope_ctx *ctx, double seconds) {
  if (!ctx || !(seconds >= 0.0) || seconds > 40.0) return set_err(ctx, OPE_EINVAL, "ope_ctx_set_wait_limit: 0 <= seconds <= 40");
  ctx->wait_ticks = (uint32_t)std::min(4.0e9, seconds * 1.0e8);   // wall_clock64: 100 MHz
  return OPE_OK;
}

int ope_ctx_sync(ope_ctx *ctx) {
  if (!ctx) return OPE_EINVAL;
  OPE_HIP(ctx, hipStreamSynchronize(ctx->stream));
  if (ctx->chain_open) OPE_HIP(ctx, hipStreamSynchronize(ctx->upd_stream));   // the last overlapped update ends a few microseconds after the last accumulate launch
  return OPE_OK;
}

const char *ope_last_error(const ope_ctx *ctx) { return ctx ? ctx->err.c_str() : g_global_err.c_str(); }

// ------------------------------------------------------------------------------------------ clouds
int ope_cloud_upload(ope_ctx *ctx, const void *base, size_t n, size_t stride_bytes, size_t xyz_off,
                     ptrdiff_t normal_off, ope_cloud **out) {
  if (!ctx || !out || (n && !base) || stride_bytes < 12) return set_err(ctx, OPE_EINVAL, "ope_cloud_upload: bad argument");
  if (n > (size_t)0x7fffffff) return set_err(ctx, OPE_EINVAL, "ope_cloud_upload: more than 2^31-1 points");
  *out = nullptr;
  OPE_HIP(ctx, hipSetDevice(ctx->device));
  ope_cloud *c = new ope_cloud();
  c->ctx = ctx;
  c->n = n;
  // The host mirrors (original-order xyz, permutation) are materialised on first use, as for clouds made on the device
  // (ensure_host): most clouds are uploaded, searched and dropped without anyone asking for them.
  c->host_valid = false;
  const unsigned char *b = static_cast<const unsigned char *>(base);
  float *d_raw = nullptr;
  int32_t *d_perm = nullptr;
  hipError_t e = hipMalloc((void **)&c->d_xyzw, sizeof(float4) * std::max<size_t>(n, 1));
  if (e == hipSuccess && n) e = tmp_malloc(ctx->stream, (void **)&d_raw, 12 * n);
  if (e == hipSuccess && n) e = tmp_malloc(ctx->stream, (void **)&d_perm, 4 * n);
  // One pass over the caller's structs: xyz gathered STRAIGHT INTO the pinned staging block (a few host threads, each over a
  // contiguous range; min / max / count merge exactly), one DMA per block of 32 MB.  (Round 3 gathered into a host vector,
  // copied that into the staging block and read the permutation back: 3 ms per million points, now ~1.)
  float lo[3] = {FLT_MAX, FLT_MAX, FLT_MAX}, hi[3] = {-FLT_MAX, -FLT_MAX, -FLT_MAX};
  size_t nv = 0;
  {
    struct Part { float lo[3], hi[3]; size_t nv; };
    const unsigned hw = std::max(1u, std::thread::hardware_concurrency());
    unsigned char *blk = nullptr;
    size_t cap = 0;
    if (e == hipSuccess && n) e = stage_block(12 * n, &blk, &cap);
    const size_t per_block = cap / 12;
    for (size_t i_base = 0; e == hipSuccess && i_base < n; i_base += per_block) {
      const size_t cnt = std::min(per_block, n - i_base);
      const unsigned nt = cnt >= ((size_t)1 << 18) ? std::min(8u, hw) : 1u;
      std::vector<Part> parts(nt);
      float *dst = reinterpret_cast<float *>(blk);
      auto work = [&](unsigned t) {
        Part pt{{FLT_MAX, FLT_MAX, FLT_MAX}, {-FLT_MAX, -FLT_MAX, -FLT_MAX}, 0};
        const size_t i0 = cnt * t / nt, i1 = cnt * (t + 1) / nt;
        for (size_t i = i0; i < i1; ++i) {
          float *p = dst + 3 * i;
          std::memcpy(p, b + (i_base + i) * stride_bytes + xyz_off, 12);
          if (!finite3(p)) continue;
          ++pt.nv;
          for (int d = 0; d < 3; ++d) { pt.lo[d] = std::min(pt.lo[d], p[d]); pt.hi[d] = std::max(pt.hi[d], p[d]); }
        }
        parts[t] = pt;
      };
      std::vector<std::thread> pool;
      for (unsigned t = 1; t < nt; ++t) pool.emplace_back(work, t);
      work(0);
      for (auto &th : pool) th.join();
      for (const Part &pt : parts) {
        nv += pt.nv;
        for (int d = 0; d < 3; ++d) { lo[d] = std::min(lo[d], pt.lo[d]); hi[d] = std::max(hi[d], pt.hi[d]); }
      }
      e = hipMemcpyAsync(reinterpret_cast<unsigned char *>(d_raw) + 12 * i_base, blk, 12 * cnt, hipMemcpyHostToDevice, ctx->stream);
      if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);   // (the block is refilled, or handed back, next)
    }
  }
  c->n_valid = nv;
  if (nv == 0) { for (int d = 0; d < 3; ++d) lo[d] = hi[d] = 0.f; }
  std::memcpy(c->bb_lo, lo, sizeof lo);
  std::memcpy(c->bb_hi, hi, sizeof hi);
  // Morton order (10 bits per axis over the cloud's own bbox); non-finite points go last.  Keys, sort and the
  // gather into float4 {x, y, z, input index} run on the device (a host std::sort of 1 M keys took ~50 ms).
  float inv[3];
  for (int d = 0; d < 3; ++d) inv[d] = (hi[d] > lo[d]) ? 1023.999f / (hi[d] - lo[d]) : 0.f;
  if (e == hipSuccess && n) e = morton_order_device(ctx->stream, d_raw, n, lo, inv, c->d_xyzw, d_perm);
  tmp_free(ctx->stream, d_raw);
  tmp_free(ctx->stream, d_perm);
  if (e != hipSuccess) {
    ope_cloud_free(c);
    return set_err(ctx, OPE_EHIP, std::string("ope_cloud_upload: ") + hipGetErrorString(e));
  }
  *out = c;
  OPE_DUMP_HASH("cloud_upload xyzw", c->d_xyzw, 16 * n, true);
  if (normal_off >= 0) {
    std::vector<float> nrm(n * 3);
    for (size_t i = 0; i < n; ++i) std::memcpy(&nrm[3 * i], b + i * stride_bytes + (size_t)normal_off, 12);
    int rc = ope_cloud_set_normals(ctx, c, nrm.data());
    if (rc != OPE_OK) { ope_cloud_free(c); *out = nullptr; return rc; }
  }
  return OPE_OK;
}

int ope_cloud_set_normals(ope_ctx *ctx, ope_cloud *cloud, const float *normals_xyz) {
  if (!ctx || !cloud || !normals_xyz) return set_err(ctx, OPE_EINVAL, "ope_cloud_set_normals: bad argument");
  { const int rch = cloud->ensure_host(); if (rch != OPE_OK) return rch; }
  OPE_HIP(ctx, hipSetDevice(ctx->device));
  const size_t n = cloud->n;
  std::vector<float> packed(n * 4);
  for (size_t i = 0; i < n; ++i) {
    const size_t o = (size_t)cloud->perm[i];
    packed[4 * i + 0] = normals_xyz[3 * o + 0];
    packed[4 * i + 1] = normals_xyz[3 * o + 1];
    packed[4 * i + 2] = normals_xyz[3 * o + 2];
    packed[4 * i + 3] = 0.f;
  }
  if (!cloud->d_nrm) OPE_HIP(ctx, hipMalloc((void **)&cloud->d_nrm, sizeof(float4) * std::max<size_t>(n, 1)));
  if (n) OPE_HIP(ctx, h2d_copy(ctx->stream, cloud->d_nrm, packed.data(), sizeof(float4) * n));
  return OPE_OK;
}

size_t ope_cloud_size(const ope_cloud *cloud) { return cloud ? cloud->n : 0; }

void ope_cloud_free(ope_cloud *cloud) {
  if (!cloud) return;
  if (cloud->ctx) (void)hipSetDevice(cloud->ctx->device);
  // the last run's source: its correspondences can no longer be mapped back (ope_icp_correspondences -> OPE_ESTATE)
  if (cloud->ctx && cloud->ctx->run_src == cloud) { cloud->ctx->run_src = nullptr; cloud->ctx->run_active = false; }
  if (cloud->ctx && cloud->ctx->fixed_src == cloud) { cloud->ctx->fixed_src = nullptr; cloud->ctx->n_fixed = 0; }   // fixed correspondences index a cloud that is gone
  if (cloud->d_xyzw) (void)hipFree(cloud->d_xyzw);
  if (cloud->d_nrm) (void)hipFree(cloud->d_nrm);
  delete cloud;
}

int ope_cloud::ensure_host() const {
  if (host_valid) return OPE_OK;
  // sorted float4 {x, y, z, original index} -> original-order xyz + permutation
  std::vector<float> sorted(4 * n);
  if (n) {
    if (hipSetDevice(ctx->device) != hipSuccess || hipMemcpy(sorted.data(), d_xyzw, sizeof(float4) * n, hipMemcpyDeviceToHost) != hipSuccess)
      return ope::set_err(ctx, OPE_EHIP, "ope_cloud: could not fetch a device-made cloud");
  }
  h_xyz.resize(3 * n);
  perm.resize(n);
  for (size_t i = 0; i < n; ++i) {
    int32_t o;
    std::memcpy(&o, &sorted[4 * i + 3], 4);
    perm[i] = o;
    std::memcpy(&h_xyz[3 * (size_t)o], &sorted[4 * i], 12);
  }
  host_valid = true;
  return OPE_OK;
}

int ope_cloud_concat(ope_ctx *ctx, const ope_cloud *a, const float T_a[16], const ope_cloud *b, ope_cloud **out) {
  if (!ctx || !a || !b || !out) return set_err(ctx, OPE_EINVAL, "ope_cloud_concat: bad argument");
  *out = nullptr;
  const size_t n = a->n + b->n;
  if (n > (size_t)0x7fffffff) return set_err(ctx, OPE_EINVAL, "ope_cloud_concat: more than 2^31-1 points");
  OPE_HIP(ctx, hipSetDevice(ctx->device));
  ope_cloud *c = new ope_cloud();
  c->ctx = ctx;
  c->n = n;
  c->n_valid = a->n_valid + b->n_valid;
  c->host_valid = false;
  float rows[12];
  float *d_rows = nullptr, *d_raw = nullptr;
  int32_t *d_perm = nullptr;
  hipError_t e = hipMalloc((void **)&c->d_xyzw, sizeof(float4) * std::max<size_t>(n, 1));
  if (e == hipSuccess && n) e = hipMalloc((void **)&d_raw, 12 * n);
  if (e == hipSuccess && n) e = hipMalloc((void **)&d_perm, 4 * n);
  if (e == hipSuccess && T_a) {
    colmajor_to_rows(T_a, rows);
    e = hipMalloc((void **)&d_rows, sizeof rows);
    if (e == hipSuccess) e = h2d_copy(ctx->stream, d_rows, rows, sizeof rows);
  }
  float lo[3] = {0, 0, 0}, hi[3] = {0, 0, 0};
  if (e == hipSuccess && n) e = concat_device(ctx->stream, a->view(), d_rows, b->view(), d_raw, lo, hi);
  if (e == hipSuccess && n) {
    if (c->n_valid == 0) { for (int d = 0; d < 3; ++d) lo[d] = hi[d] = 0.f; }
    std::memcpy(c->bb_lo, lo, sizeof lo);
    std::memcpy(c->bb_hi, hi, sizeof hi);
    float inv[3];
    for (int d = 0; d < 3; ++d) inv[d] = (hi[d] > lo[d]) ? 1023.999f / (hi[d] - lo[d]) : 0.f;
    e = morton_order_device(ctx->stream, d_raw, n, lo, inv, c->d_xyzw, d_perm);
  }
  for (void *p : {(void *)d_rows, (void *)d_raw, (void *)d_perm})
    if (p) (void)hipFree(p);
  if (e != hipSuccess) {
    ope_cloud_free(c);
    return set_err(ctx, OPE_EHIP, std::string("ope_cloud_concat: ") + hipGetErrorString(e));
  }
  *out = c;
  return OPE_OK;
}

int ope_cloud_download(ope_ctx *ctx, const ope_cloud *cloud, float *out_xyz) {
  if (!ctx || !cloud || (cloud->n && !out_xyz)) return set_err(ctx, OPE_EINVAL, "ope_cloud_download: bad argument");
  const int rc = cloud->ensure_host();
  if (rc != OPE_OK) return rc;
  if (cloud->n) std::memcpy(out_xyz, cloud->h_xyz.data(), 12 * cloud->n);
  return OPE_OK;
}

// ------------------------------------------------------------------------------------------ index
void ope_index_default_params(ope_index_params *p) {
  if (p) { p->leaf_size = 16; p->grid = 1; p->grid_fill = 0.f; p->grid_max_cells = 0; }
}

int ope_index_build(ope_ctx *ctx, const ope_cloud *target, const ope_index_params *params, ope_index **out) {
  return ope::index_build_impl(ctx, target, params, false, out);
}

}  // extern "C"

// temporary: an index that lives inside one entry point (index_build_tmp, ope_internal.hpp): buffers from the stream's cache
int ope::index_build_impl(ope_ctx *ctx, const ope_cloud *target, const ope_index_params *params, bool temporary, ope_index **out) {
  if (!ctx || !target || !out) return set_err(ctx, OPE_EINVAL, "ope_index_build: bad argument");
  *out = nullptr;
  // Registration::setInputTarget: "Invalid or empty point cloud dataset given!" (registration_mod.hpp:60-64)
  if (target->n_valid == 0) return set_err(ctx, OPE_EEMPTY, "ope_index_build: invalid or empty target cloud");
  OPE_HIP(ctx, hipSetDevice(ctx->device));
  TraceRange r_build(ctx, "index_build");
  ope_index_params dp;
  ope_index_default_params(&dp);
  if (params) dp = *params;
  const size_t n = target->n_valid;
  static const bool host_build = dev_env("OPE_HOST_BUILD") != nullptr;  // developer A/B switch: the host reference builder
  if (host_build) { const int rch = target->ensure_host(); if (rch != OPE_OK) return rch; }
  if (!host_build) {
    // the finite points are the first n_valid records of the Morton-sorted device copy (w = original index)
    ope_index *ix = new ope_index();
    ix->ctx = ctx;
    ix->n = n;
    ix->n_total = target->n;
    std::memcpy(ix->bb_lo, target->bb_lo, sizeof ix->bb_lo);
    std::memcpy(ix->bb_hi, target->bb_hi, sizeof ix->bb_hi);
    for (int d = 0; d < 3; ++d) ix->pivot[d] = 0.5 * ((double)target->bb_lo[d] + (double)target->bb_hi[d]);
    if (dev_env("OPE_NO_TMP_INDEX")) temporary = false;   // developer A/B switch
    ix->tmp_alloc = temporary;
    ix->alloc_stream = ctx->stream;
    const hipError_t e = build_bvh_device(ctx->stream, target->d_xyzw, target->d_nrm, n, dp.leaf_size, target->bb_lo, target->bb_hi,
                                          &ix->depth, &ix->d_nodes, &ix->d_pts, &ix->d_nrm, &ix->d_axis2, temporary);
    if (e != hipSuccess) {
      ope_index_free(ix);
      return set_err(ctx, OPE_EHIP, std::string("ope_index_build: ") + hipGetErrorString(e));
    }
    // the bucketed side of the index (grid_build.hip) is built by the first 1-NN ICP run that uses this index: the
    // indexes behind normals, FPFH and the filters never need it
    ix->want_grid = dp.grid != 0;
    ix->grid_mode = dp.grid;
    ix->grid_fill = dp.grid_fill;
    ix->grid_max_cells = dp.grid_max_cells;
    OPE_DUMP_HASH("index_build nodes", ix->d_nodes, 48 * ((size_t)2 << ix->depth), true);
    OPE_DUMP_HASH("index_build pts", ix->d_pts, 16 * ix->n, true);
    *out = ix;
    return OPE_OK;
  }
  std::vector<float> xyz(n * 3), nrm;
  std::vector<int32_t> ids(n);
  size_t m = 0;
  for (size_t i = 0; i < target->n; ++i) {
    const float *p = &target->h_xyz[3 * i];
    if (!finite3(p)) continue;
    xyz[3 * m] = p[0]; xyz[3 * m + 1] = p[1]; xyz[3 * m + 2] = p[2];
    ids[m++] = (int32_t)i;
  }
  if (target->d_nrm) {
    // fetch normals back in original order
    std::vector<float> packed(target->n * 4);
    OPE_HIP(ctx, hipMemcpy(packed.data(), target->d_nrm, sizeof(float4) * target->n, hipMemcpyDeviceToHost));
    std::vector<float> orig(target->n * 3);
    for (size_t i = 0; i < target->n; ++i)
      for (int d = 0; d < 3; ++d) orig[3 * (size_t)target->perm[i] + d] = packed[4 * i + d];
    nrm.resize(n * 3);
    for (size_t k = 0; k < n; ++k)
      for (int d = 0; d < 3; ++d) nrm[3 * k + d] = orig[3 * (size_t)ids[k] + d];
  }
  HostBvh hb;
  build_bvh_host(xyz.data(), ids.data(), nrm.empty() ? nullptr : nrm.data(), n, dp.leaf_size, hb);
  ope_index *ix = new ope_index();
  ix->ctx = ctx;
  ix->n = n;
  ix->n_total = target->n;
  ix->depth = hb.depth;
  std::memcpy(ix->bb_lo, target->bb_lo, sizeof ix->bb_lo);
  std::memcpy(ix->bb_hi, target->bb_hi, sizeof ix->bb_hi);
  for (int d = 0; d < 3; ++d) ix->pivot[d] = 0.5 * ((double)target->bb_lo[d] + (double)target->bb_hi[d]);
  hipError_t e = hipMalloc((void **)&ix->d_nodes, sizeof(float) * hb.nodes.size());
  if (e == hipSuccess) e = h2d_copy(ctx->stream, ix->d_nodes, hb.nodes.data(), sizeof(float) * hb.nodes.size());
  if (e == hipSuccess) e = hipMalloc((void **)&ix->d_pts, sizeof(float4) * (n + kPtsPad));
  if (e == hipSuccess) e = hipMemset(ix->d_pts + n, 0, sizeof(float4) * kPtsPad);
  if (e == hipSuccess) e = h2d_copy(ctx->stream, ix->d_pts, hb.pts4.data(), sizeof(float4) * n);
  if (e == hipSuccess && !hb.nrm4.empty()) {
    e = hipMalloc((void **)&ix->d_nrm, sizeof(float4) * n);
    if (e == hipSuccess) e = h2d_copy(ctx->stream, ix->d_nrm, hb.nrm4.data(), sizeof(float4) * n);
  }
  if (e != hipSuccess) {
    ope_index_free(ix);
    return set_err(ctx, OPE_EHIP, std::string("ope_index_build: ") + hipGetErrorString(e));
  }
  *out = ix;
  return OPE_OK;
}

extern "C" {

void ope_index_free(ope_index *index) {
  if (!index) return;
  if (index->ctx) (void)hipSetDevice(index->ctx->device);
  if (index->ctx && index->ctx->run_tgt == index) { index->ctx->run_tgt = nullptr; index->ctx->run_active = false; }
  if (index->tmp_alloc) {   // (back to the cache of the stream the build and every use were enqueued on: reused behind them)
    for (void *p : {(void *)index->d_nodes, (void *)index->d_pts, (void *)index->d_nrm, (void *)index->d_axis2}) tmp_free(index->alloc_stream, p);
  } else {
    if (index->d_nodes) (void)hipFree(index->d_nodes);
    if (index->d_pts) (void)hipFree(index->d_pts);
    if (index->d_nrm) (void)hipFree(index->d_nrm);
    if (index->d_axis2) (void)hipFree(index->d_axis2);
  }
  for (void *p : {(void *)index->d_gpts, (void *)index->d_gnrm, (void *)index->d_cell_start, (void *)index->d_gpos})
    if (p) (void)hipFree(p);
  delete index;
}

// ------------------------------------------------------------------------------------------ searches
static int upload_T(ope_ctx *ctx, const float *T, float **d_T) {
  *d_T = nullptr;
  if (!T) return OPE_OK;
  float rows[12];
  colmajor_to_rows(T, rows);
  int rc = ensure_scratch(ctx, 1 << 16);
  if (rc != OPE_OK) return rc;
  OPE_HIP(ctx, h2d_copy(ctx->stream, ctx->d_scratch, rows, sizeof rows));
  *d_T = static_cast<float *>(ctx->d_scratch);
  return OPE_OK;
}

int ope_nn_search(ope_ctx *ctx, const ope_cloud *queries, const ope_index *index, const float *T, int32_t *out_idx,
                  float *out_d2) {
  if (!ctx || !queries || !index || !out_idx || !out_d2) return set_err(ctx, OPE_EINVAL, "ope_nn_search: bad argument");
  { const int rch = queries->ensure_host(); if (rch != OPE_OK) return rch; }
  OPE_HIP(ctx, hipSetDevice(ctx->device));
  const size_t n = queries->n;
  if (n == 0) return OPE_OK;
  float *d_T;
  int rc = upload_T(ctx, T, &d_T);
  if (rc != OPE_OK) return rc;
  int32_t *d_idx = nullptr;
  float *d_d2 = nullptr;
  hipError_t e = hipMalloc((void **)&d_idx, sizeof(int32_t) * n);
  if (e == hipSuccess) e = hipMalloc((void **)&d_d2, sizeof(float) * n);
  std::vector<int32_t> hi(n);
  std::vector<float> hd(n);
  if (e == hipSuccess) {
    launch_nn_search(ctx->stream, queries->view(), index->view(), d_T, d_idx, d_d2);
    e = hipMemcpyAsync(hi.data(), d_idx, sizeof(int32_t) * n, hipMemcpyDeviceToHost, ctx->stream);
  }
  if (e == hipSuccess) e = hipMemcpyAsync(hd.data(), d_d2, sizeof(float) * n, hipMemcpyDeviceToHost, ctx->stream);
  if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
  if (d_idx) (void)hipFree(d_idx);
  if (d_d2) (void)hipFree(d_d2);
  if (e != hipSuccess) return set_err(ctx, OPE_EHIP, std::string("ope_nn_search: ") + hipGetErrorString(e));
  for (size_t i = 0; i < n; ++i) {
    out_idx[queries->perm[i]] = hi[i];
    out_d2[queries->perm[i]] = hd[i];
  }
  return OPE_OK;
}

int ope_knn_search(ope_ctx *ctx, const ope_cloud *queries, const ope_index *index, const float *T, int k,
                   int32_t *out_idx, float *out_d2) {
  if (!ctx || !queries || !index || !out_idx || !out_d2 || k < 1 || k > 32)
    return set_err(ctx, OPE_EINVAL, "ope_knn_search: bad argument (1 <= k <= 32)");
  { const int rch = queries->ensure_host(); if (rch != OPE_OK) return rch; }
  OPE_HIP(ctx, hipSetDevice(ctx->device));
  const size_t n = queries->n;
  if (n == 0) return OPE_OK;
  float *d_T;
  int rc = upload_T(ctx, T, &d_T);
  if (rc != OPE_OK) return rc;
  int32_t *d_idx = nullptr;
  float *d_d2 = nullptr;
  hipError_t e = hipMalloc((void **)&d_idx, sizeof(int32_t) * n * k);
  if (e == hipSuccess) e = hipMalloc((void **)&d_d2, sizeof(float) * n * k);
  std::vector<int32_t> hi(n * k);
  std::vector<float> hd(n * k);
  if (e == hipSuccess) {
    launch_knn_search(ctx->stream, queries->view(), index->view(), d_T, k, d_idx, d_d2);
    e = hipMemcpyAsync(hi.data(), d_idx, sizeof(int32_t) * n * k, hipMemcpyDeviceToHost, ctx->stream);
  }
  if (e == hipSuccess) e = hipMemcpyAsync(hd.data(), d_d2, sizeof(float) * n * k, hipMemcpyDeviceToHost, ctx->stream);
  if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
  if (d_idx) (void)hipFree(d_idx);
  if (d_d2) (void)hipFree(d_d2);
  if (e != hipSuccess) return set_err(ctx, OPE_EHIP, std::string("ope_knn_search: ") + hipGetErrorString(e));
  for (size_t i = 0; i < n; ++i) {
    std::memcpy(out_idx + (size_t)queries->perm[i] * k, hi.data() + i * k, sizeof(int32_t) * k);
    std::memcpy(out_d2 + (size_t)queries->perm[i] * k, hd.data() + i * k, sizeof(float) * k);
  }
  return OPE_OK;
}

// ------------------------------------------------------------------------------------------ ICP
void ope_icp_default_params(ope_icp_params *p) {
  if (!p) return;
  std::memset(p, 0, sizeof *p);
  p->max_iterations = 10;
  p->transformation_epsilon = 0.0;
  p->euclidean_fitness_epsilon = -std::numeric_limits<double>::max();
  p->max_corr_dist = std::sqrt(std::numeric_limits<double>::max());
  p->min_correspondences = 3;
  p->use_reciprocal = 0;
  p->corr_mode = OPE_CORR_NEAREST;
  p->k_normal_shooting = 20;
  p->use_surface_normal_rej = 0;
  p->surface_normal_thr = 0.7;
  p->use_self_occluded_rej = 0;
  p->self_occluded_thr = 0.6;
  p->mse_threshold_absolute = 1e-12;
  p->failure_after_max_iter = 0;
  p->check_every = 10;
  p->estimator = OPE_EST_SVD;
  p->deterministic_sums = 0;
  p->tree_walk = OPE_WALK_AUTO;
  p->update_launch = OPE_UPDATE_OVERLAPPED;
  p->skip_certificates = OPE_CERT_AUTO;
}

int64_t ope_icp_overlapped_updates(const ope_ctx *ctx) { return ctx ? (int64_t)ctx->chain_seq : 0; }
int ope_icp_update_fallbacks(const ope_ctx *ctx) { return ctx ? ctx->chain_fallbacks : 0; }

int ope_icp_certificate_stats(ope_ctx *ctx, int64_t out[4]) {
  if (!ctx || !out) return OPE_EINVAL;
  out[0] = out[1] = out[2] = out[3] = 0;
  if (!ctx->d_work_counter || !ctx->d_state) return OPE_OK;
  OPE_HIP(ctx, hipSetDevice(ctx->device));
  const int rcj = chain_join(ctx);
  if (rcj != OPE_OK) return rcj;
  uint32_t w[4] = {0, 0, 0, 0};
  int mode = 0;
  float move = 0.f;
  OPE_HIP(ctx, hipMemcpyAsync(w, ctx->d_work_counter + 40, sizeof w, hipMemcpyDeviceToHost, ctx->stream));
  OPE_HIP(ctx, hipMemcpyAsync(&mode, reinterpret_cast<unsigned char *>(ctx->d_state) + offsetof(IcpState, cert_mode), sizeof mode, hipMemcpyDeviceToHost, ctx->stream));
  OPE_HIP(ctx, hipMemcpyAsync(&move, reinterpret_cast<unsigned char *>(ctx->d_state) + offsetof(IcpState, last_move), sizeof move, hipMemcpyDeviceToHost, ctx->stream));
  OPE_HIP(ctx, hipStreamSynchronize(ctx->stream));
  out[0] = (int64_t)(((uint64_t)w[1] << 32) | w[0]);
  out[1] = (int64_t)w[2];
  out[2] = mode;
  out[3] = (int64_t)((double)move * 1e9);
  return OPE_OK;
}

int ope_icp_kernel_launches(const ope_ctx *ctx, int64_t counts[OPE_KERNEL_KINDS]) {
  if (!ctx || !counts) return OPE_EINVAL;
  for (int k = 0; k < OPE_KERNEL_KINDS; ++k) counts[k] = ctx->kernel_launches[k];
  return OPE_OK;
}

int ope_icp_set_global_sizes(ope_ctx *ctx, int64_t n_src_total, int64_t n_tgt_total) {
  if (!ctx) return OPE_EINVAL;
  ctx->n_src_total = n_src_total;
  ctx->n_tgt_total = n_tgt_total;
  return OPE_OK;
}

static int icp_begin_impl(ope_ctx *ctx, const ope_cloud *src, const ope_index *tgt, const float *guess, const ope_icp_params *params);

// The uniform grid over an index's points, built on first use (the handle is logically const for its callers: the grid
// is a cache over the same points).
static int ensure_grid(ope_ctx *ctx, const ope_index *cix) {
  ope_index *ix = const_cast<ope_index *>(cix);
  std::lock_guard<std::mutex> lock(ix->grid_mutex);   // contexts or threads that share the index: one of them builds, the others wait
  if (ix->has_grid || !ix->want_grid || !ix->d_pts) return OPE_OK;
  TraceRange r(ctx, "grid_build");
  const hipError_t eg = build_grid_device(ctx->stream, ix->d_pts, ix->d_nrm, ix->n, ix->bb_lo, ix->bb_hi, ix->grid_fill, (uint32_t)std::max(ix->grid_max_cells, 0), &ix->grid, &ix->d_gpts, &ix->d_gnrm,
                                          &ix->d_cell_start, &ix->d_gpos);
  if (eg != hipSuccess) return set_err(ctx, OPE_EHIP, std::string("ope_icp_begin(grid build): ") + hipGetErrorString(eg));
  ix->grid.gpts = ix->d_gpts; ix->grid.gnrm = ix->d_gnrm; ix->grid.cell_start = ix->d_cell_start; ix->grid.gpos_of_bvhpos = ix->d_gpos;
  float amax = 0.f;
  for (int d = 0; d < 3; ++d) amax = std::max({amax, std::fabs(ix->bb_lo[d]), std::fabs(ix->bb_hi[d])});
  ix->grid.eps = 4e-7f * amax + 1e-30f;
  // the build ran on this context's stream: finished before another context's stream may read it
  OPE_HIP(ctx, hipStreamSynchronize(ctx->stream));
  ix->has_grid = true;
  return OPE_OK;
}

int ope_icp_begin(ope_ctx *ctx, const ope_cloud *src, const ope_index *tgt, const float *guess,
                  const ope_icp_params *params) {
  const int rc = icp_begin_impl(ctx, src, tgt, guess, params);
  // sizes set with ope_icp_set_global_sizes for a run that never started must not leak into the next one
  if (rc != OPE_OK && ctx) ctx->n_src_total = ctx->n_tgt_total = 0;
  return rc;
}

static int icp_begin_impl(ope_ctx *ctx, const ope_cloud *src, const ope_index *tgt, const float *guess, const ope_icp_params *params) {
  if (!ctx || !src) return set_err(ctx, OPE_EINVAL, "ope_icp_begin: bad argument");
  // Registration::initCompute: "No input target dataset was given!" (registration_mod.hpp:73-77)
  if (!tgt) return set_err(ctx, OPE_EEMPTY, "ope_icp_begin: no input target dataset was given");
  if (ctx->p2p_broken)
    return set_err(ctx, OPE_ECOMM, "ope_icp_begin: the peer-to-peer communicator timed out in an earlier run and must be re-created (ope_comm_destroy, then ope_comm_init_rank or ope_comm_p2p_open/connect)");
  ope_icp_params p;
  ope_icp_default_params(&p);
  if (params) p = *params;
  if (p.use_reciprocal && p.corr_mode != OPE_CORR_NEAREST)
    return set_err(ctx, OPE_EINVAL, "ope_icp_begin: reciprocal correspondences are defined for 1-NN estimation only");
  const bool need_src_nrm = p.corr_mode == OPE_CORR_NORMAL_SHOOTING || p.use_surface_normal_rej || p.use_self_occluded_rej;
  if (need_src_nrm && !src->d_nrm) return set_err(ctx, OPE_EINVAL, "ope_icp_begin: source normals required but absent");
  if (p.estimator != OPE_EST_SVD && p.estimator != OPE_EST_POINT_TO_PLANE_LLS && p.estimator != OPE_EST_POINT_TO_PLANE_LM)
    return set_err(ctx, OPE_EINVAL, "ope_icp_begin: unknown estimator");
  if (p.tree_walk < OPE_WALK_AUTO || p.tree_walk > OPE_WALK_PACKET) return set_err(ctx, OPE_EINVAL, "ope_icp_begin: unknown tree_walk");
  if (p.update_launch != OPE_UPDATE_OVERLAPPED && p.update_launch != OPE_UPDATE_IN_LINE) return set_err(ctx, OPE_EINVAL, "ope_icp_begin: unknown update_launch");
  if (p.skip_certificates < OPE_CERT_AUTO || p.skip_certificates > OPE_CERT_ALWAYS) return set_err(ctx, OPE_EINVAL, "ope_icp_begin: unknown skip_certificates");
  if (ctx->n_fixed > 0) {
    if (ctx->fixed_src != src || ctx->fixed_tgt_n != tgt->n_total)
      return set_err(ctx, OPE_EINVAL, "ope_icp_begin: the fixed correspondences were set for another pair of clouds (ope_icp_set_fixed_correspondences with n = 0 clears them)");
    if (p.estimator != OPE_EST_SVD) return set_err(ctx, OPE_EINVAL, "ope_icp_begin: fixed correspondences are supported with the SVD estimator only");
    if (p.use_reciprocal) return set_err(ctx, OPE_EINVAL, "ope_icp_begin: fixed correspondences and reciprocal correspondences do not combine (the reference's reciprocal estimation ignores them)");
    if ((p.use_surface_normal_rej && !ctx->fixed_has_nrm) || ((p.use_self_occluded_rej || p.corr_mode == OPE_CORR_NORMAL_SHOOTING) && !ctx->fixed_has_src_nrm))
      return set_err(ctx, OPE_EINVAL, "ope_icp_begin: the clouds the fixed correspondences were gathered from had no normals");
  }
  if ((p.estimator == OPE_EST_POINT_TO_PLANE_LLS || p.estimator == OPE_EST_POINT_TO_PLANE_LM) && !tgt->d_nrm)
    return set_err(ctx, OPE_EINVAL, "ope_icp_begin: the point-to-plane estimator needs target normals (build the index from a cloud with normals)");
  if (p.use_surface_normal_rej && !tgt->d_nrm)
    return set_err(ctx, OPE_EINVAL, "ope_icp_begin: target normals required (build the index from a cloud with normals)");
  if (p.corr_mode == OPE_CORR_NORMAL_SHOOTING && (p.k_normal_shooting < 1 || p.k_normal_shooting > 32))
    return set_err(ctx, OPE_EINVAL, "ope_icp_begin: 1 <= k_normal_shooting <= 32");
  OPE_HIP(ctx, hipSetDevice(ctx->device));
  if (ctx->plan_pending) OPE_HIP(ctx, hipStreamSynchronize(ctx->plan_stream));   // a plan of the previous run still in the making reads the buffers below
  {
    const int rcj = chain_join(ctx);   // overlapped updates of a run that was never polled or ended
    if (rcj != OPE_OK) return rcj;
    ctx->chain_u_synced = false;
  }
  // From here on the context's run state is being rebuilt: any failure leaves NO run behind (not the previous one
  // with new buffers, and not stale align-strength sizes).
  const int64_t keep_ns = ctx->n_src_total, keep_nt = ctx->n_tgt_total;
  abort_run(ctx);
  struct BeginGuard {
    ope_ctx *c; bool ok = false;
    ~BeginGuard() { if (!ok) abort_run(c); }
  } guard{ctx};
  ctx->n_src_total = keep_ns; ctx->n_tgt_total = keep_nt;
  if (ctx->corr_cap < std::max<size_t>(src->n, 1)) {   // an empty source still gets one (unused) slot
    if (ctx->d_corr_match) (void)hipFree(ctx->d_corr_match);
    if (ctx->d_corr_d2) (void)hipFree(ctx->d_corr_d2);
    if (ctx->d_hint) (void)hipFree(ctx->d_hint);
    if (ctx->d_cert_q) (void)hipFree(ctx->d_cert_q);
    if (ctx->d_cert_pos) (void)hipFree(ctx->d_cert_pos);
    if (ctx->d_cert_l) (void)hipFree(ctx->d_cert_l);
    ctx->d_cert_l = nullptr;
    ctx->d_corr_match = nullptr; ctx->d_corr_d2 = nullptr; ctx->d_hint = nullptr; ctx->d_cert_q = nullptr; ctx->d_cert_pos = nullptr; ctx->corr_cap = 0;
    OPE_HIP(ctx, hipMalloc((void **)&ctx->d_corr_match, sizeof(int32_t) * std::max<size_t>(src->n, 1)));
    OPE_HIP(ctx, hipMalloc((void **)&ctx->d_corr_d2, sizeof(float) * std::max<size_t>(src->n, 1)));
    OPE_HIP(ctx, hipMalloc((void **)&ctx->d_hint, sizeof(uint32_t) * std::max<size_t>(src->n, 1)));
    OPE_HIP(ctx, hipMalloc((void **)&ctx->d_cert_q, sizeof(float4) * std::max<size_t>(src->n, 1)));
    OPE_HIP(ctx, hipMalloc((void **)&ctx->d_cert_pos, sizeof(uint32_t) * kCertCand * std::max<size_t>(src->n, 1)));
    OPE_HIP(ctx, hipMalloc((void **)&ctx->d_cert_l, sizeof(float) * std::max<size_t>(src->n, 1)));
    ctx->corr_cap = std::max<size_t>(src->n, 1);
  }
  ctx->use_grid = false;
  ctx->force_plan_at = -1;
  // (deterministic_sums: the tree kernel in the chunks' natural order only — which kernel runs when, and which wave takes
  // which chunk, follow from measured times otherwise, and with them the grouping of the fp64 additions)
  if (p.corr_mode == OPE_CORR_NEAREST && !p.use_reciprocal && tgt->want_grid && src->n_valid > 0 && p.estimator != OPE_EST_POINT_TO_PLANE_LM &&
      !p.deterministic_sums && (tgt->grid_mode == 2 || src->n_valid >= kGridMinQueries) &&
      (tgt->has_grid || tgt->grid_mode == 2 || p.max_iterations > 1)) {   // a one-pass run (the facade's stand-alone determineCorrespondences) does not pay for a grid build
    const int rcg = ensure_grid(ctx, tgt);
    if (rcg != OPE_OK) return rcg;
    ctx->use_grid = tgt->has_grid;
  }
  ctx->grid_auto = ctx->use_grid && tgt->grid_mode == 1;
  if (ctx->grid_auto) {
    // a lower bound on the clutter share before anything is searched: source points, under the guess, more than eight cells
    // outside the target's bounding box (grid_build.hip); above the threshold the run starts on the tree kernel
    static const float I4g[16] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1};
    float rows[12];
    colmajor_to_rows(guess ? guess : I4g, rows);
    uint32_t n_out = 0;
    grid_count_outside(ctx->stream, src->view(), rows, tgt->bb_lo, tgt->bb_hi, 8.0f / tgt->grid.inv, ctx->d_work_counter + 20);
    OPE_HIP(ctx, hipMemcpyAsync(&n_out, ctx->d_work_counter + 20, 4, hipMemcpyDeviceToHost, ctx->stream));
    OPE_HIP(ctx, hipStreamSynchronize(ctx->stream));
    if ((double)n_out > kGridMaxTreeShare * (double)src->n_valid) ctx->use_grid = false;
  }
  if (ctx->use_grid || ctx->grid_auto) {   // (a run that starts on the tree kernel may still move to the grid kernel later)
    const size_t cap = std::max<size_t>(src->n, 1);
    if (ctx->grid_cap < cap) {
      for (void *q : {(void *)ctx->d_ghint, (void *)ctx->d_qorder, (void *)ctx->d_qclass, ctx->d_part_tmp})
        if (q) (void)hipFree(q);
      ctx->d_ghint = ctx->d_qorder = nullptr; ctx->d_qclass = nullptr; ctx->d_part_tmp = nullptr; ctx->grid_cap = 0;
      OPE_HIP(ctx, hipMalloc((void **)&ctx->d_ghint, 4 * cap));
      OPE_HIP(ctx, hipMalloc((void **)&ctx->d_qorder, 4 * cap));
      OPE_HIP(ctx, hipMalloc((void **)&ctx->d_qclass, cap));
      ctx->part_tmp_bytes = grid_plan_tmp_bytes((uint32_t)cap, (uint32_t)(cap / 64 + 2));
      OPE_HIP(ctx, hipMalloc(&ctx->d_part_tmp, ctx->part_tmp_bytes));
      ctx->grid_cap = cap;
    }
    if (!ctx->grid_probe_event) {
      OPE_HIP(ctx, hipEventCreateWithFlags(&ctx->grid_probe_event, hipEventDisableTiming));
      OPE_HIP(ctx, hipHostMalloc((void **)&ctx->h_grid_probe, 64));
    }
    ctx->grid_probe_pending = false;
    OPE_HIP(ctx, hipMemsetAsync(ctx->d_ghint, 0, 4 * cap, ctx->stream));
    OPE_HIP(ctx, hipMemsetAsync(ctx->d_qclass, 0, cap, ctx->stream));
    fill_iota(ctx->stream, ctx->d_qorder, (uint32_t)cap);
  }
  // every slot starts as "no correspondence" (non-finite points never get written)
  OPE_HIP(ctx, hipMemsetAsync(ctx->d_corr_match, 0xff, sizeof(int32_t) * std::max<size_t>(src->n, 1), ctx->stream));
  {
    const size_t nch = (src->n_valid + 63) / 64 + 1;
    if (ctx->chunk_cap < nch) {
      for (void *p : {(void *)ctx->d_chunk_cost, (void *)ctx->d_chunk_cost_sorted, (void *)ctx->d_chunk_ids,
                      (void *)ctx->d_chunk_order, ctx->d_plan_tmp})
        if (p) (void)hipFree(p);
      ctx->d_chunk_cost = ctx->d_chunk_cost_sorted = ctx->d_chunk_ids = ctx->d_chunk_order = nullptr;
      ctx->d_plan_tmp = nullptr;
      ctx->chunk_cap = 0;
      OPE_HIP(ctx, hipMalloc((void **)&ctx->d_chunk_cost, 4 * 2 * nch));   // two halves, by launch parity (enqueue_accumulate)
      OPE_HIP(ctx, hipMalloc((void **)&ctx->d_chunk_cost_sorted, 4 * nch));
      if (ctx->d_chunk_keys) (void)hipFree(ctx->d_chunk_keys);
      ctx->d_chunk_keys = nullptr;
      OPE_HIP(ctx, hipMalloc((void **)&ctx->d_chunk_keys, 4 * nch));
      OPE_HIP(ctx, hipMalloc((void **)&ctx->d_chunk_ids, 4 * nch));
      OPE_HIP(ctx, hipMalloc((void **)&ctx->d_chunk_order, 4 * nch));
      if (ctx->d_slot_list) (void)hipFree(ctx->d_slot_list);
      ctx->d_slot_list = nullptr;
      OPE_HIP(ctx, hipMalloc((void **)&ctx->d_slot_list, 4 * (nch + 7 * (nch / 4 + 1))));   // n chunks + 7 per group-walked chunk (at most n / 4)
      for (void *q : {(void *)ctx->d_cost_snap, (void *)ctx->d_plan_sorted[0], (void *)ctx->d_plan_sorted[1], (void *)ctx->d_plan_order[0],
                      (void *)ctx->d_plan_order[1], (void *)ctx->d_plan_slots[0], (void *)ctx->d_plan_slots[1]})
        if (q) (void)hipFree(q);
      ctx->d_cost_snap = nullptr;
      for (int k = 0; k < 2; ++k) ctx->d_plan_sorted[k] = ctx->d_plan_order[k] = ctx->d_plan_slots[k] = nullptr;
      OPE_HIP(ctx, hipMalloc((void **)&ctx->d_cost_snap, 4 * nch));
      for (int k = 0; k < 2; ++k) {
        OPE_HIP(ctx, hipMalloc((void **)&ctx->d_plan_sorted[k], 4 * nch));
        OPE_HIP(ctx, hipMalloc((void **)&ctx->d_plan_order[k], 4 * nch));
        OPE_HIP(ctx, hipMalloc((void **)&ctx->d_plan_slots[k], 4 * (nch + 7 * (nch / 4 + 1))));
      }
      size_t tb = 0;
      if (chunk_plan(ctx->stream, ctx->d_chunk_cost, ctx->d_chunk_cost_sorted, ctx->d_chunk_ids, ctx->d_chunk_order,
                     (uint32_t)nch, nullptr, tb) != 0)
        return set_err(ctx, OPE_EHIP, "ope_icp_begin: rocprim temp-storage query failed");
      OPE_HIP(ctx, hipMalloc(&ctx->d_plan_tmp, std::max<size_t>(tb, 16)));
      ctx->plan_tmp_bytes = tb;
      ctx->chunk_cap = nch;
    }
    fill_iota(ctx->stream, ctx->d_chunk_ids, (uint32_t)nch);
    OPE_HIP(ctx, hipMemsetAsync(ctx->d_chunk_cost, 0, 4 * 2 * ctx->chunk_cap, ctx->stream));
    ctx->plan_pending = false;
    ctx->plan_cur = 0;
    ctx->plan_cur_slots = false;
    ctx->plan_valid = false;
    ctx->slot_list_valid = false;
    ctx->acc_launches = 0;
    for (int64_t &k : ctx->kernel_launches) k = 0;
  }
  if (p.corr_mode == OPE_CORR_NORMAL_SHOOTING && ctx->knn_rk_cap < std::max<size_t>(src->n, 1)) {
    if (ctx->d_knn_rk) (void)hipFree(ctx->d_knn_rk);
    ctx->d_knn_rk = nullptr; ctx->knn_rk_cap = 0;
    OPE_HIP(ctx, hipMalloc((void **)&ctx->d_knn_rk, sizeof(float) * std::max<size_t>(src->n, 1)));
    ctx->knn_rk_cap = std::max<size_t>(src->n, 1);
  }
  // no start hints yet: the first iteration walks top-down (hints belong to one (src, tgt) pairing)
  OPE_HIP(ctx, hipMemsetAsync(ctx->d_hint, 0, sizeof(uint32_t) * std::max<size_t>(src->n, 1), ctx->stream));

  IcpState *h = ctx->h_state;
  std::memset(h, 0, sizeof *h);
  static const float I4[16] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1};
  const float *g = guess ? guess : I4;
  for (int i = 0; i < 16; ++i) { h->F[i] = g[i]; h->Tk[i] = I4[i]; }
  colmajor_to_rows(g, h->Ff);
  for (int d = 0; d < 3; ++d) h->pivot[d] = tgt->pivot[d];
  h->prev_mse = h->cur_mse = std::numeric_limits<double>::max();
  // thresholds wired as at icp_mod.hpp:164-168 (quirk Q1: rotation threshold = 1 - transformation_epsilon)
  h->rotation_threshold = 1.0 - p.transformation_epsilon;
  h->translation_threshold = p.transformation_epsilon;
  h->mse_threshold_relative = p.euclidean_fitness_epsilon;
  h->mse_threshold_absolute = p.mse_threshold_absolute;
  h->max_corr_dist = p.max_corr_dist;
  h->max_d2 = p.max_corr_dist * p.max_corr_dist;
  h->surface_normal_thr = p.surface_normal_thr;
  h->self_occluded_thr = p.self_occluded_thr;
  h->max_iterations = p.max_iterations;
  h->failure_after_max_iter = p.failure_after_max_iter;
  h->min_correspondences = p.min_correspondences;
  h->corr_mode = p.corr_mode;
  h->k_normal_shooting = p.k_normal_shooting;
  h->use_surface_normal_rej = p.use_surface_normal_rej;
  h->use_self_occluded_rej = p.use_self_occluded_rej;
  h->use_reciprocal = p.use_reciprocal;
  h->estimator = p.estimator;
  {
    // Skip certificates (ope.h: skip_certificates): plain 1-NN runs.  Automatic: kept once an update moves no scene point by
    // more than a quarter of the target's point spacing, estimated from the surface of its bounding box over its size (a
    // closed surface sampled by n points inside that box: spacing ~ sqrt(area / n); the factor matters little — a run that starts keeping them
    // early pays a few percent per launch for walks that report their bounds, one that starts late walks a little longer).
    ctx->cert_run = p.corr_mode == OPE_CORR_NEAREST && !p.use_reciprocal && p.deterministic_sums == 0 && p.skip_certificates != OPE_CERT_OFF &&
                    !dev_env("OPE_NO_CERT");
    const double ex = (double)tgt->bb_hi[0] - tgt->bb_lo[0], ey = (double)tgt->bb_hi[1] - tgt->bb_lo[1], ez = (double)tgt->bb_hi[2] - tgt->bb_lo[2];
    const double spacing = std::sqrt(2.0 * (ex * ey + ey * ez + ez * ex) / (double)std::max<size_t>(tgt->n, 1));
    // When launches start keeping certificates (automatic mode).  A run on a clean source — every query near the surface: the
    // grid kernel's case, or an index without a grid — profits as soon as surface points can hold one: the update that moves the
    // scene by less than spacing / 24.  A run that starts on the tree kernel BECAUSE the device counted more than 3 % of its
    // queries far outside the target (clutter) does not: its launches last as long as the walks of the far queries, whose
    // neighbours lie within microns of each other in distance, whatever the surface points save (C3's first
    // hundred iterations got 6 % slower with the early threshold) — it waits until the scene moves by less than spacing / 512,
    // shortly before those queries can hold certificates too.
    const bool cluttered = ctx->grid_auto && !ctx->use_grid;
    h->cert_thr = !ctx->cert_run ? -1.0f : p.skip_certificates == OPE_CERT_ALWAYS ? std::numeric_limits<float>::infinity()
                                 : (float)(spacing / (cluttered ? 512.0 : (double)kCertWorth));
    if (const char *e = dev_env("OPE_CERT_THR")) h->cert_thr = (float)atof(e);   // developer sweep (metres)
    h->cert_mode = (ctx->cert_run && p.skip_certificates == OPE_CERT_ALWAYS) ? 1 : 0;
    ctx->cert_seen = h->cert_mode != 0;
    h->host_cert = (ctx->cert_run && !ctx->cert_seen) ? ctx->d_pace + 1 : nullptr;
    // what a certificate is worth (icp_accumulate_kernel): the (kCertCand + 1)-th neighbour of a query D from a surface sampled at
    // `spacing` lies ~ kCertCand spacing^2 / (2 pi D) further out than the nearest one, never more than about the spacing itself
    h->cert_cap = (float)spacing;
    h->cert_k = (float)((double)kCertCand * spacing * spacing / (2.0 * 3.14159265358979323846));
    if (const char *e = dev_env("OPE_CERT_CAP")) h->cert_cap = (float)atof(e);   // developer sweep (metres)
    double r2 = 0;
    for (int d = 0; d < 3; ++d) {
      h->src_c[d] = 0.5f * (src->bb_lo[d] + src->bb_hi[d]);
      const double hd = 0.5 * ((double)src->bb_hi[d] - (double)src->bb_lo[d]);
      r2 += hd * hd;
    }
    h->src_r = (float)std::sqrt(r2);
    // positions index THIS target's point order: certificates never outlive the pairing ("no candidate 0" = no certificate);
    // the per-query worth of a certificate is read off the previous launch's distance: none yet
    if (ctx->cert_run) {
      OPE_HIP(ctx, hipMemsetAsync(ctx->d_cert_pos, 0, sizeof(uint32_t) * std::max<size_t>(src->n, 1), ctx->stream));
      OPE_HIP(ctx, hipMemsetAsync(ctx->d_corr_d2, 0x7f, sizeof(float) * std::max<size_t>(src->n, 1), ctx->stream));   // 0x7f7f7f7f = 3.4e38
    }
  }
  {
    // inverse of the guess (adjugate), rows layout
    const double a = g[0], b = g[4], c = g[8], d = g[1], e = g[5], f = g[9], gg = g[2], hh = g[6], ii = g[10];
    const double det = a * (e * ii - f * hh) - b * (d * ii - f * gg) + c * (d * hh - e * gg);
    const double id = det != 0 ? 1.0 / det : 0.0;
    const double M[9] = {(e * ii - f * hh) * id, (c * hh - b * ii) * id, (b * f - c * e) * id,
                         (f * gg - d * ii) * id, (a * ii - c * gg) * id, (c * d - a * f) * id,
                         (d * hh - e * gg) * id, (b * gg - a * hh) * id, (a * e - b * d) * id};
    for (int r = 0; r < 3; ++r) {
      h->Finv[4 * r + 0] = (float)M[3 * r]; h->Finv[4 * r + 1] = (float)M[3 * r + 1]; h->Finv[4 * r + 2] = (float)M[3 * r + 2];
      h->Finv[4 * r + 3] = (float)(-(M[3 * r] * g[12] + M[3 * r + 1] * g[13] + M[3 * r + 2] * g[14]));
    }
  }
  if (ctx->run_src_index) { ope_index_free(ctx->run_src_index); ctx->run_src_index = nullptr; }
  if (p.use_reciprocal) {
    int rc = ope_index_build(ctx, src, nullptr, &ctx->run_src_index);
    if (rc != OPE_OK) return rc;
  }
  OPE_HIP(ctx, hipMemcpyAsync(ctx->d_state, h, sizeof *h, hipMemcpyHostToDevice, ctx->stream));
  // k-NN runs of more than one pass: start leaves for the first launch (icp_kernels.hip: seed_hints_kernel)
  if (p.corr_mode == OPE_CORR_NORMAL_SHOOTING && p.max_iterations > 1 && src->n_valid > 0 && !dev_env("OPE_NO_SEED"))
    launch_seed_hints(ctx->stream, src->view(), tgt->view(), ctx->d_state, ctx->d_hint);
  if (ctx->d_sums_ext)
    OPE_HIP(ctx, hipMemsetAsync(ctx->d_sums_ext, 0, sizeof(double) * (p.estimator == OPE_EST_POINT_TO_PLANE_LLS ? kNumSumsMax : kNumSums), ctx->stream));
  OPE_HIP(ctx, hipMemsetAsync(ctx->d_work_counter, 0, 256, ctx->stream));
  OPE_HIP(ctx, hipMemsetAsync(ctx->d_lm_stats, 0, sizeof(double) * 96, ctx->stream));
  ctx->measuring_flag = false;
  // partial-sum rows of blocks that do not exist in this run must read as zero
  OPE_HIP(ctx, hipMemsetAsync(ctx->d_partials, 0, sizeof(double) * kNumSumsMax * kAccMaxBlocks, ctx->stream));
  // the pinned block is reused for read-back: make sure the upload is finished with it first
  OPE_HIP(ctx, hipStreamSynchronize(ctx->stream));
  // (every launch of the previous run is over: nothing writes the pace words any more)
  ctx->h_pace[0] = 0u; ctx->h_pace[1] = 0u;
  ctx->launch_no = 0;
  ctx->pace_off = dev_env("OPE_NO_PACE") != nullptr;

#ifdef OPE_DEVELOPER
  if (dev_env("OPE_DUMP_HASH")) {   // developer probe: checksums of everything the run reads, to stderr
    auto fnv = [&](const void *d, size_t bytes) -> unsigned long long {
      std::vector<unsigned char> h(bytes);
      if (bytes && d) (void)hipMemcpy(h.data(), d, bytes, hipMemcpyDeviceToHost);
      unsigned long long x = 1469598103934665603ull;
      if (d) for (unsigned char c : h) { x ^= c; x *= 1099511628211ull; }
      return x;
    };
    std::fprintf(stderr, "[hash] begin src n %zu valid %zu xyzw %016llx nrm %016llx | tgt n %zu depth %d nodes %016llx pts %016llx nrm %016llx\n", src->n,
                 src->n_valid, fnv(src->d_xyzw, 16 * src->n), fnv(src->d_nrm, 16 * src->n), tgt->n, tgt->depth,
                 fnv(tgt->d_nodes, 48 * ((size_t)2 << tgt->depth)), fnv(tgt->d_pts, 16 * tgt->n), fnv(tgt->d_nrm, 16 * tgt->n));
  }
#endif
  ctx->run_src = src;
  ctx->run_tgt = tgt;
  ctx->corr_run_n = src->n;
  ctx->run_params = p;
  ctx->run_active = true;
  ctx->iters_enqueued = 0;
  const int block = (p.corr_mode == OPE_CORR_NEAREST && !p.use_reciprocal) ? kAccBlock : ((p.corr_mode == OPE_CORR_NEAREST) ? kAccBlock : 256);
  // twice the waves the chunks alone would need: the chunks handed to 8-lane groups take eight slots each, and on a
  // launch that does not fill the GPU every slot should find a wave of its own (125 k queries: 84 -> 74 us)
  ctx->acc_blocks = (int)std::min<size_t>(std::max<size_t>(2 * ((src->n_valid + block - 1) / block), 1), kAccMaxBlocks);
  if (p.corr_mode == OPE_CORR_NEAREST && !p.use_reciprocal) {
    // ... but never more blocks than the GPU holds at once: the surplus would start when the first blocks end
    const bool nrm = p.use_surface_normal_rej || p.use_self_occluded_rej || p.estimator == OPE_EST_POINT_TO_PLANE_LLS;
    const uint32_t nch0 = (uint32_t)((src->n_valid + 63) / 64);
    const bool packet = p.tree_walk == OPE_WALK_PACKET ? true : p.tree_walk == OPE_WALK_LANE ? false : nch0 > (uint32_t)ctx->n_cu * 4u * (uint32_t)kAccWavesPerSimd;
    const int per_cu = icp_accumulate_blocks_per_cu(nrm, packet, tgt->has_grid && tgt->grid_mode == 2);
    if (per_cu > 0) ctx->acc_blocks = std::min(ctx->acc_blocks, per_cu * ctx->n_cu);
  }
  // Overlapped update launches (icp_kernels.hip, acc_launch_begin): plain 1-NN runs of one rank with atomic sums and an
  // estimator whose update is icp_update_kernel.  The accumulate launch then stays short of what the GPU holds, so that the
  // update's wave finds room beside it whenever it arrives.
  ctx->chained = p.update_launch == OPE_UPDATE_OVERLAPPED && !ctx->chain_broken && p.corr_mode == OPE_CORR_NEAREST && !p.use_reciprocal &&
                 p.deterministic_sums == 0 && p.estimator != OPE_EST_POINT_TO_PLANE_LM && ctx->nccl_comm == nullptr && !ctx->p2p_ok &&
                 ctx->n_fixed == 0 && !dev_env("OPE_NO_CHAIN");
  ctx->n_fixed_run = ctx->n_fixed;
  ctx->chain_on = false;
  ctx->chain_seq = 0;
  ctx->chain_tickets = 0;
  ctx->acc_blocks_cert = ctx->acc_blocks;
  if (ctx->cert_run) {
    const bool nrm = p.use_surface_normal_rej || p.use_self_occluded_rej || p.estimator == OPE_EST_POINT_TO_PLANE_LLS;
    int pc = std::min(icp_accumulate_cert_blocks_per_cu(nrm, false, false), icp_accumulate_cert_blocks_per_cu(nrm, true, false));
    if (tgt->has_grid) pc = std::min(pc, icp_accumulate_cert_blocks_per_cu(nrm, false, true));
    if (pc > 0) ctx->acc_blocks_cert = std::max(1, std::min(ctx->acc_blocks, pc * ctx->n_cu - ctx->n_xcd));
  }
  if (ctx->chained) {
    const bool nrm = p.use_surface_normal_rej || p.use_self_occluded_rej || p.estimator == OPE_EST_POINT_TO_PLANE_LLS;
    int per_cu = icp_accumulate_blocks_per_cu(nrm, false, false);
    per_cu = std::min(per_cu, icp_accumulate_blocks_per_cu(nrm, true, false));
    if (tgt->has_grid) per_cu = std::min(per_cu, icp_accumulate_blocks_per_cu(nrm, false, true));
    // one block slot left free on EVERY XCD: blocks are dealt to the eight XCDs round-robin and which XCD the update's one
    // workgroup is dealt to is not fixed; with the slot it always finds room, also when it arrives together with a launch
    // whose blocks are all about to wait for it (a host that enqueues no faster than the GPU works)
    if (per_cu > 0) ctx->acc_blocks = std::max(1, std::min(ctx->acc_blocks, per_cu * ctx->n_cu - ctx->n_xcd));
    else ctx->chained = false;
  }
  if (const char *e = dev_env("OPE_ACC_BLOCKS")) ctx->acc_blocks = std::max(1, std::min(atoi(e), (int)kAccMaxBlocks));
  if (ctx->n_src_total <= 0) ctx->n_src_total = (int64_t)src->n;
  if (ctx->n_tgt_total <= 0) ctx->n_tgt_total = (int64_t)tgt->n_total;
  guard.ok = true;
  return OPE_OK;
}

// setFixedCorrespondences (vPCL icp_mod.h:268): pairs by ORIGINAL indices into the two clouds; their points (and normals) are
// gathered once into four float4 per pair, so that the per-iteration launch needs neither cloud's order.
int ope_icp_set_fixed_correspondences(ope_ctx *ctx, const ope_cloud *src, const ope_cloud *tgt_cloud, const int32_t *index_query,
                                      const int32_t *index_match, size_t n) {
  if (!ctx) return OPE_EINVAL;
  if (ctx->run_active && ctx->n_fixed_run > 0)
    return set_err(ctx, OPE_ESTATE, "ope_icp_set_fixed_correspondences: the run in progress uses the pairs set before it (end it first)");
  OPE_HIP(ctx, hipSetDevice(ctx->device));
  if (ctx->d_fixed) { OPE_HIP(ctx, hipStreamSynchronize(ctx->stream)); OPE_HIP(ctx, hipFree(ctx->d_fixed)); ctx->d_fixed = nullptr; }
  ctx->n_fixed = 0;
  ctx->fixed_src = nullptr;
  ctx->fixed_tgt_n = 0;
  if (n == 0) return OPE_OK;   // clearCorrespondences (icp_mod.h:281)
  if (!src || !tgt_cloud || !index_query || !index_match) return set_err(ctx, OPE_EINVAL, "ope_icp_set_fixed_correspondences: bad argument");
  if (n > ((size_t)1 << 24)) return set_err(ctx, OPE_EINVAL, "ope_icp_set_fixed_correspondences: more than 2^24 pairs");
  int rc = src->ensure_host();
  if (rc == OPE_OK) rc = tgt_cloud->ensure_host();
  if (rc != OPE_OK) return rc;
  // original index -> position in the Morton order, for the indices that occur
  auto positions = [&](const ope_cloud *c, const int32_t *idx, std::vector<uint32_t> &pos, int slot) -> bool {
    std::vector<std::pair<int32_t, uint32_t>> want(n);
    for (size_t f = 0; f < n; ++f) {
      if (idx[f] < 0 || (size_t)idx[f] >= c->n) return false;
      want[f] = {idx[f], (uint32_t)f};
    }
    std::sort(want.begin(), want.end());
    for (size_t p = 0; p < c->perm.size(); ++p) {
      auto it = std::lower_bound(want.begin(), want.end(), std::make_pair(c->perm[p], (uint32_t)0));
      for (; it != want.end() && it->first == c->perm[p]; ++it) pos[2 * it->second + slot] = (uint32_t)p;
    }
    return true;
  };
  std::vector<uint32_t> pos(2 * n, 0u);
  if (!positions(src, index_query, pos, 0) || !positions(tgt_cloud, index_match, pos, 1))
    return set_err(ctx, OPE_EINVAL, "ope_icp_set_fixed_correspondences: index out of range");
  uint32_t *d_pos = nullptr;
  OPE_HIP(ctx, hipMalloc((void **)&d_pos, sizeof(uint32_t) * 2 * n));
  // (+ one float2 per pair behind the gathered points: what the last launch saw of it, ope_icp_fixed_correspondences)
  if (hipMalloc((void **)&ctx->d_fixed, sizeof(float4) * 4 * n + sizeof(float2) * n) != hipSuccess) { (void)hipFree(d_pos); ctx->d_fixed = nullptr; return set_err(ctx, OPE_ENOMEM, "ope_icp_set_fixed_correspondences: out of device memory"); }
  hipError_t e = hipMemcpyAsync(d_pos, pos.data(), sizeof(uint32_t) * 2 * n, hipMemcpyHostToDevice, ctx->stream);
  if (e == hipSuccess) {
    launch_gather_fixed_pairs(ctx->stream, src->view(), tgt_cloud->view(), d_pos, (uint32_t)n, ctx->d_fixed);
    e = hipMemsetAsync(ctx->d_fixed + 4 * n, 0, sizeof(float2) * n, ctx->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
  }
  (void)hipFree(d_pos);
  if (e != hipSuccess) { (void)hipFree(ctx->d_fixed); ctx->d_fixed = nullptr; return set_err(ctx, OPE_EHIP, "ope_icp_set_fixed_correspondences: gather failed"); }
  ctx->n_fixed = n;
  ctx->fixed_src = src;
  ctx->fixed_tgt_n = tgt_cloud->n;
  ctx->fixed_has_nrm = src->d_nrm != nullptr && tgt_cloud->d_nrm != nullptr;
  ctx->fixed_has_src_nrm = src->d_nrm != nullptr;
  return OPE_OK;
}

// the given pairs' share of the sums, after an accumulate launch (sharded runs: the caller sets them on exactly one rank, with
// indices into that rank's shard, so that they are added once)
static void enqueue_fixed_pairs(ope_ctx *ctx) {
  if (ctx->n_fixed_run == 0) return;
  launch_icp_fixed_pairs(ctx->stream, ctx->d_state, ctx->d_fixed, (uint32_t)ctx->n_fixed_run, sums_ptr(ctx),
                         (float2 *)(ctx->d_fixed + 4 * ctx->n_fixed_run));
}

// The given pairs as the last iteration of the last run saw them: the reference's correspondence estimation writes each
// pair's `distance` back through the caller's pointer every iteration (correspondence_estimation_mod.hpp:150-161), lists the
// pair in front of the searched ones, and ICP appends the survivors of the first rejector behind them (icp_mod.hpp:210-224).
int ope_icp_fixed_correspondences(ope_ctx *ctx, float *distance, int32_t *listed, int32_t *appended, size_t cap, size_t *n_out) {
  if (!ctx || !n_out) return set_err(ctx, OPE_EINVAL, "ope_icp_fixed_correspondences: bad argument");
  const size_t n = ctx->n_fixed;
  *n_out = n;
  if (n == 0 || cap == 0) return OPE_OK;
  if (ctx->n_fixed_run != n) return set_err(ctx, OPE_ESTATE, "ope_icp_fixed_correspondences: no run has used the pairs that are set");
  OPE_HIP(ctx, hipSetDevice(ctx->device));
  std::vector<float2> seen(n);
  OPE_HIP(ctx, hipMemcpyAsync(seen.data(), ctx->d_fixed + 4 * n, sizeof(float2) * n, hipMemcpyDeviceToHost, ctx->stream));
  OPE_HIP(ctx, hipStreamSynchronize(ctx->stream));
  for (size_t f = 0; f < n && f < cap; ++f) {
    int bits;
    std::memcpy(&bits, &seen[f].y, sizeof bits);
    if (distance) distance[f] = seen[f].x;
    if (listed) listed[f] = bits & 1;
    if (appended) appended[f] = (bits >> 1) & 1;
  }
  return OPE_OK;
}

int ope_icp_accumulate(ope_ctx *ctx) {
  if (!ctx || !ctx->run_active) return set_err(ctx, OPE_ESTATE, "ope_icp_accumulate: no run in progress");
  if (ctx->run_params.estimator == OPE_EST_POINT_TO_PLANE_LM)
    return set_err(ctx, OPE_EINVAL, "ope_icp_accumulate: the LM estimator iterates inside ope_icp_iterate / ope_icp_run (it reduces over the correspondences several times per iteration)");
  const bool atomic = atomic_sums(ctx);
  int rc = chain_join(ctx);
  if (rc != OPE_OK) return rc;
  ctx->chain_on = false;
  rc = enqueue_accumulate(ctx, atomic);
  if (rc != OPE_OK) return rc;
  if (!atomic)
    launch_icp_reduce_update(ctx->stream, ctx->d_state, ctx->d_partials, sums_ptr(ctx), ctx->acc_blocks, /*do_update=*/false, ctx->d_work_counter);
  enqueue_fixed_pairs(ctx);
  OPE_HIP(ctx, hipGetLastError());
  return OPE_OK;
}

void *ope_icp_sums_device(ope_ctx *ctx) {
  if (!ctx || !ctx->d_state) return nullptr;
  return sums_ptr(ctx);
}

int ope_icp_set_sums_buffer(ope_ctx *ctx, void *device_ptr) {
  if (!ctx) return OPE_EINVAL;
  {
    const int rcj = chain_join(ctx);
    if (rcj != OPE_OK) return rcj;
  }
  ctx->d_sums_ext = static_cast<double *>(device_ptr);
  // accumulate launches add into the sums: a buffer handed over mid-run starts from zero like the built-in one
  // (before a run it is cleared by ope_icp_begin, which knows how many sums the estimator uses)
  if (ctx->d_sums_ext && ctx->run_active) OPE_HIP(ctx, hipMemsetAsync(ctx->d_sums_ext, 0, sizeof(double) * run_nsums(ctx), ctx->stream));
  return OPE_OK;
}

int ope_icp_iterate(ope_ctx *ctx, int n_iterations) {
  if (!ctx || !ctx->run_active) return set_err(ctx, OPE_ESTATE, "ope_icp_iterate: no run in progress");
  // a context with a communicator takes the accumulate -> all-reduce -> update sequence, also with one rank
  // (that is how a one-GPU box exercises the path the multi-GPU runs take)
  const bool sharded = ctx->nccl_comm != nullptr || ctx->p2p_ok;
  static const bool split_update = dev_env("OPE_SPLIT_UPDATE") != nullptr;  // developer A/B switch
  const bool atomic = atomic_sums(ctx);
  // Overlapped update launches (icp_kernels.hip, acc_launch_begin): the updates go to their own stream, which waits for the
  // launch stream once per run; the launch stream waits for it only when something is about to read or rewrite the state
  // (chain_join: poll, end, the step-wise entry points, the next begin).  In between nothing on the host or in the streams
  // orders an update against the accumulate launches — the kernels do.
  const bool chained = ctx->chained && !sharded && atomic && ctx->d_sums_ext == nullptr;
  if (chained) {
    if (!ctx->chain_u_synced) {   // once per run (and after in-line updates): the update stream sees what the launch stream did to the state
      OPE_HIP(ctx, hipEventRecord(ctx->ev_chain_s, ctx->stream));
      OPE_HIP(ctx, hipStreamWaitEvent(ctx->upd_stream, ctx->ev_chain_s, 0));
      ctx->chain_u_synced = true;
    }
    ctx->chain_open = true;
  } else {
    const int rcj = chain_join(ctx);
    if (rcj != OPE_OK) return rcj;
    ctx->chain_u_synced = false;
  }
  struct ChainFlag {   // enqueue_accumulate hands the chain words to the launches of THIS batch only
    ope_ctx *c;
    ~ChainFlag() { c->chain_on = false; }
  } chain_flag{ctx};
  ctx->chain_on = chained;
  for (int b = 0; b < n_iterations; ++b) {
    TraceRange r_iter(ctx, "icp_iter");
    int rc;
    {
      TraceRange r_nn(ctx, "nn");
      rc = enqueue_accumulate(ctx, atomic);
    }
    if (rc != OPE_OK) return rc;
    TraceRange r_red(ctx, "reduce");
    if (chained) {
      // (the ticket word counts up through the run: one ticket per block of every overlapped launch so far, this one included)
      ctx->chain_tickets += (uint32_t)ctx->launch_blocks;
      launch_icp_update_chained(ctx->upd_stream, ctx->d_state, run_nsums(ctx), chain_ptr(ctx), ctx->chain_seq, ctx->chain_tickets, ctx->wait_ticks);
      ++ctx->chain_seq;
      continue;
    }
    if (ctx->run_params.estimator == OPE_EST_POINT_TO_PLANE_LM) {
      // correspondences are in place (corr_match = index positions); their 17 sums give n and the MSE, one more pass gives
      // the 91 sums the minimiser works on (lm.hip); in sharded runs both sets are summed over the ranks; the minimisation
      // and the update step are one launch (icp_lm_update_kernel).  Nothing synchronises the host.
      if (!atomic)
        launch_icp_reduce_update(ctx->stream, ctx->d_state, ctx->d_partials, sums_ptr(ctx), ctx->acc_blocks, false, ctx->d_work_counter);
      launch_lm_stats(ctx->stream, ctx->n_cu, ctx->run_src->view(), ctx->run_tgt->view(), ctx->d_state, ctx->d_corr_match, ctx->d_lm_stats);
      if (sharded) {
        if (comm_uses_p2p(ctx)) {
          rc = comm_p2p_exchange(ctx, sums_ptr(ctx), kNumSums);
          if (rc == OPE_OK) rc = comm_p2p_exchange(ctx, ctx->d_lm_stats, 91);
        } else {
          rc = comm_allreduce_sums(ctx, sums_ptr(ctx), kNumSums);
          if (rc == OPE_OK) rc = comm_allreduce_sums(ctx, ctx->d_lm_stats, 91);
        }
        if (rc != OPE_OK) return rc;
      }
      launch_icp_lm_update(ctx->stream, ctx->d_state, sums_ptr(ctx), ctx->d_lm_stats);
      continue;
    }
    if (sharded) {
      if (!atomic)
        launch_icp_reduce_update(ctx->stream, ctx->d_state, ctx->d_partials, sums_ptr(ctx), ctx->acc_blocks, false, ctx->d_work_counter);
      enqueue_fixed_pairs(ctx);
      if (comm_uses_p2p(ctx)) {
        // exchange through the peers' slots and update in one launch (no collective, no separate update kernel)
        rc = comm_p2p_exchange_update(ctx, ctx->d_state, sums_ptr(ctx), run_nsums(ctx));
        if (rc != OPE_OK) return rc;
      } else {
        rc = comm_allreduce_sums(ctx, sums_ptr(ctx), ctx->run_params.estimator == OPE_EST_POINT_TO_PLANE_LLS ? kNumSumsMax : kNumSums);
        if (rc != OPE_OK) return rc;
        launch_icp_update(ctx->stream, ctx->d_state, sums_ptr(ctx), run_nsums(ctx), nullptr);
      }
    } else if (atomic) {
      enqueue_fixed_pairs(ctx);
      launch_icp_update(ctx->stream, ctx->d_state, sums_ptr(ctx), run_nsums(ctx), nullptr);
    } else if (split_update || ctx->n_fixed_run > 0) {
      launch_icp_reduce_update(ctx->stream, ctx->d_state, ctx->d_partials, sums_ptr(ctx), ctx->acc_blocks, false, ctx->d_work_counter);
      enqueue_fixed_pairs(ctx);
      launch_icp_update(ctx->stream, ctx->d_state, sums_ptr(ctx), run_nsums(ctx), nullptr);
    } else {
      launch_icp_reduce_update(ctx->stream, ctx->d_state, ctx->d_partials, sums_ptr(ctx), ctx->acc_blocks, true, ctx->d_work_counter);
    }
  }
  ctx->iters_enqueued += n_iterations;
  OPE_HIP(ctx, hipGetLastError());
  return OPE_OK;
}

int ope_icp_profile(ope_ctx *ctx, int max_launches) {
  if (!ctx) return OPE_EINVAL;
  OPE_HIP(ctx, hipSetDevice(ctx->device));
  ctx->prof_enabled = max_launches > 0;
  ctx->prof_used = 0;
  while (ctx->prof_events.size() < 2 * (size_t)std::max(max_launches, 0)) {
    hipEvent_t e;
    OPE_HIP(ctx, hipEventCreate(&e));
    ctx->prof_events.push_back(e);
  }
  return OPE_OK;
}

int ope_icp_profile_read(ope_ctx *ctx, double *total_ms, int *n_launches) {
  if (!ctx) return OPE_EINVAL;
  OPE_HIP(ctx, hipStreamSynchronize(ctx->stream));
  double tot = 0;
  for (size_t i = 0; i < ctx->prof_used; ++i) {
    float ms = 0.f;
    OPE_HIP(ctx, hipEventElapsedTime(&ms, ctx->prof_events[2 * i], ctx->prof_events[2 * i + 1]));
    tot += ms;
  }
  if (total_ms) *total_ms = tot;
  if (n_launches) *n_launches = (int)ctx->prof_used;
  return OPE_OK;
}

int ope_icp_profile_launches(ope_ctx *ctx, float *ms, size_t cap, size_t *n_out) {
  if (!ctx || !n_out || (cap && !ms)) return OPE_EINVAL;
  OPE_HIP(ctx, hipStreamSynchronize(ctx->stream));
  const size_t n = std::min(cap, ctx->prof_used);
  for (size_t i = 0; i < n; ++i) OPE_HIP(ctx, hipEventElapsedTime(ms + i, ctx->prof_events[2 * i], ctx->prof_events[2 * i + 1]));
  *n_out = n;
  return OPE_OK;
}

#ifdef OPE_DEVELOPER
// tools/cost_probe.py: the per-chunk costs the last launches measured (s_memtime ticks >> 4), the plan's order and plan_info
int ope_debug_chunk_costs(ope_ctx *ctx, uint32_t *cost, uint32_t *order, uint32_t *plan_info4, int n) {
  if (!ctx || !ctx->run_active) return OPE_ESTATE;
  OPE_HIP(ctx, hipStreamSynchronize(ctx->stream));
  OPE_HIP(ctx, hipMemcpy(cost, ctx->d_chunk_cost + (size_t)((ctx->acc_launches + 1) & 1) * ctx->chunk_cap, 4 * (size_t)n, hipMemcpyDeviceToHost));   // the half the last launch wrote
  OPE_HIP(ctx, hipMemcpy(order, ctx->use_grid ? ctx->d_chunk_order : ctx->d_plan_order[ctx->plan_cur], 4 * (size_t)n, hipMemcpyDeviceToHost));
  OPE_HIP(ctx, hipMemcpy(plan_info4, ctx->use_grid ? ctx->d_work_counter + 8 : ctx->d_plan_out + 8 * ctx->plan_cur, 16, hipMemcpyDeviceToHost));
  return OPE_OK;
}
#endif

int ope_icp_update(ope_ctx *ctx) {
  if (!ctx || !ctx->run_active) return set_err(ctx, OPE_ESTATE, "ope_icp_update: no run in progress");
  {
    const int rcj = chain_join(ctx);
    if (rcj != OPE_OK) return rcj;
    ctx->chain_u_synced = false;
  }
  launch_icp_update(ctx->stream, ctx->d_state, sums_ptr(ctx), run_nsums(ctx), nullptr);
  OPE_HIP(ctx, hipGetLastError());
  ++ctx->iters_enqueued;
  return OPE_OK;
}

static void fill_result(const ope_ctx *ctx, ope_icp_result *r) {
  const IcpState *h = ctx->h_state;
  r->iterations = h->iterations;
  r->converged = h->converged;
  r->state = h->state;
  r->last_mse = h->cur_mse;
  r->n_corr = h->n_corr;
  const double denom = (double)(ctx->n_src_total + ctx->n_tgt_total);
  r->align_strength = denom > 0 ? (double)h->n_corr / denom : 0.0;
}

int ope_icp_poll(ope_ctx *ctx, ope_icp_result *result) {
  if (!ctx || !ctx->run_active) return set_err(ctx, OPE_ESTATE, "ope_icp_poll: no run in progress");
  {
    const int rcj = chain_join(ctx);
    if (rcj != OPE_OK) return rcj;
  }
  OPE_HIP(ctx, hipMemcpyAsync(ctx->h_state, ctx->d_state, sizeof(IcpState), hipMemcpyDeviceToHost, ctx->stream));
  OPE_HIP(ctx, hipStreamSynchronize(ctx->stream));
  if (result) fill_result(ctx, result);
  if (ctx->h_state->chain_error) {
    // An overlapped update launch waited its 2 s for an accumulate launch that could not run beside it (the GPU's block slots
    // held by other work, or a tool that serialises dispatches).  Nothing is lost but time: the update that gave up changed
    // nothing but the two flags, every launch behind it found "done" and left the sums alone, and the state still holds the
    // last completed iteration.  The iterations that did not happen are enqueued again, in line, and so are all later runs of
    // this context.
    ctx->chain_broken = true;
    ctx->chained = false;
    ++ctx->chain_fallbacks;
    const int applied = ctx->h_state->iterations, lost = ctx->iters_enqueued - applied;
    if (lost <= 0 || lost > ctx->iters_enqueued || ctx->chain_recovering)
      return set_err(ctx, OPE_EHIP, "an overlapped update launch waited 2 s for its accumulate launch and the run could not be resumed in line (ope_icp_params.update_launch = OPE_UPDATE_IN_LINE avoids the overlapped launches from the start)");
    ctx->chain_recovering = true;
    unsigned char *st8 = reinterpret_cast<unsigned char *>(ctx->d_state);
    hipError_t e = hipMemsetAsync(st8 + offsetof(IcpState, S), 0, sizeof(double) * kNumSumsMax, ctx->stream);
    if (e == hipSuccess) e = hipMemsetAsync(st8 + offsetof(IcpState, done), 0, sizeof(int), ctx->stream);
    if (e == hipSuccess) e = hipMemsetAsync(st8 + offsetof(IcpState, chain_error), 0, sizeof(int), ctx->stream);
    int rc = e == hipSuccess ? OPE_OK : set_err(ctx, OPE_EHIP, std::string("ope_icp_poll: ") + hipGetErrorString(e));
    ctx->iters_enqueued = applied;
    if (rc == OPE_OK) rc = ope_icp_iterate(ctx, lost);
    if (rc == OPE_OK) rc = ope_icp_poll(ctx, result);
    ctx->chain_recovering = false;
    return rc;
  }
  if (ctx->h_state->comm_error) {
    // the ranks' sequence numbers no longer agree and the slots hold the words of the aborted exchange: a later run could
    // accept them as fresh.  The communicator is unusable from here on (ope_icp_begin refuses) until it is re-created.
    ctx->p2p_ok = false;
    ctx->p2p_broken = true;
    return set_err(ctx, OPE_ECOMM, "a peer's sums did not arrive within 5 s (peer-to-peer exchange): the run was ended; re-create the communicator (ope_comm_destroy, then ope_comm_init_rank or ope_comm_p2p_open/connect) before the next sharded run");
  }
  return OPE_OK;
}

int ope_icp_current_transform(ope_ctx *ctx, float out_T[16]) {
  if (!ctx || !ctx->run_active || !out_T) return set_err(ctx, OPE_ESTATE, "ope_icp_current_transform: no run in progress");
  int rc = ope_icp_poll(ctx, nullptr);
  if (rc != OPE_OK) return rc;
  for (int i = 0; i < 16; ++i) out_T[i] = (float)ctx->h_state->F[i];
  return OPE_OK;
}

int ope_icp_end(ope_ctx *ctx, float out_T[16], ope_icp_result *result) {
  int rc = ope_icp_poll(ctx, result);
  if (rc != OPE_OK) {
    if (ctx && ctx->run_active) abort_run(ctx);   // e.g. a peer that stopped sending: the run is over, the context stays usable
    return rc;
  }
  if (out_T)
    for (int i = 0; i < 16; ++i) out_T[i] = (float)ctx->h_state->F[i];
  ctx->run_active = false;
  ctx->n_src_total = ctx->n_tgt_total = 0;
  return OPE_OK;
}

int ope_icp_last_incremental(ope_ctx *ctx, float out_T[16]) {
  if (!ctx || !out_T || !ctx->h_state) return set_err(ctx, OPE_EINVAL, "ope_icp_last_incremental: bad argument");
  for (int i = 0; i < 16; ++i) out_T[i] = (float)ctx->h_state->Tk[i];
  return OPE_OK;
}

int ope_icp_run(ope_ctx *ctx, const ope_cloud *src, const ope_index *tgt, const float *guess,
                const ope_icp_params *params, float out_T[16], ope_icp_result *result) {
  static const float I4[16] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1};
  if (out_T) std::memcpy(out_T, I4, sizeof I4);
  if (result) std::memset(result, 0, sizeof *result);
  int rc = ope_icp_begin(ctx, src, tgt, guess, params);
  if (rc != OPE_OK) return rc;
  const ope_icp_params &p = ctx->run_params;
  const int max_it = std::max(p.max_iterations, 1);
  int it = 0;
  while (it < max_it) {
    const int batch = p.check_every > 0 ? std::min(p.check_every, max_it - it) : (max_it - it);
    rc = ope_icp_iterate(ctx, batch);
    if (rc != OPE_OK) { abort_run(ctx); return rc; }
    it += batch;
    if (it < max_it) {
      rc = ope_icp_poll(ctx, nullptr);
      if (rc != OPE_OK) { abort_run(ctx); return rc; }
      if (ctx->h_state->done) break;
    }
  }
  return ope_icp_end(ctx, out_T, result);
}

int ope_icp_correspondences(ope_ctx *ctx, int32_t *index_query, int32_t *index_match, float *distance, size_t cap,
                            size_t *n_out) {
  if (!ctx || !n_out) return set_err(ctx, OPE_EINVAL, "ope_icp_correspondences: bad argument");
  if (!ctx->run_src || ctx->corr_run_n != ctx->run_src->n)
    return set_err(ctx, OPE_ESTATE, "ope_icp_correspondences: no finished run whose source cloud is still alive");
  const ope_cloud *src = ctx->run_src;
  const size_t n = src->n;
  { const int rch = src->ensure_host(); if (rch != OPE_OK) return rch; }
  std::vector<int32_t> hm(n);
  std::vector<float> hd(n);
  OPE_HIP(ctx, hipSetDevice(ctx->device));
  if (n) {
    int32_t *d_match = ctx->d_corr_match;
    int32_t *d_tmp_match = nullptr;
    if (ctx->run_params.estimator == OPE_EST_POINT_TO_PLANE_LM && ctx->run_tgt) {
      // LM runs keep index positions in corr_match: translate a copy to original target indices
      OPE_HIP(ctx, hipMalloc((void **)&d_tmp_match, sizeof(int32_t) * n));
      OPE_HIP(ctx, hipMemcpyAsync(d_tmp_match, ctx->d_corr_match, sizeof(int32_t) * n, hipMemcpyDeviceToDevice, ctx->stream));
      launch_lm_pos_to_orig(ctx->stream, ctx->run_tgt->view(), d_tmp_match, (uint32_t)src->n_valid);
      d_match = d_tmp_match;
    }
    struct FreeTmp { int32_t *p; ~FreeTmp() { if (p) (void)hipFree(p); } } free_tmp{d_tmp_match};
    OPE_HIP(ctx, hipMemcpyAsync(hm.data(), d_match, sizeof(int32_t) * n, hipMemcpyDeviceToHost, ctx->stream));
    OPE_HIP(ctx, hipMemcpyAsync(hd.data(), ctx->d_corr_d2, sizeof(float) * n, hipMemcpyDeviceToHost, ctx->stream));
    OPE_HIP(ctx, hipStreamSynchronize(ctx->stream));
  }
  std::vector<int32_t> by_orig(n, -1);
  std::vector<float> d_by_orig(n, 0.f);
  for (size_t i = 0; i < src->n_valid; ++i) {
    by_orig[src->perm[i]] = hm[i];
    d_by_orig[src->perm[i]] = hd[i];
  }
  size_t m = 0;
  for (size_t o = 0; o < n; ++o) {
    if (by_orig[o] < 0) continue;
    if (m < cap) {
      if (index_query) index_query[m] = (int32_t)o;
      if (index_match) index_match[m] = by_orig[o];
      if (distance) distance[m] = d_by_orig[o];
    }
    ++m;
  }
  *n_out = m;
  return OPE_OK;
}

int ope_fitness(ope_ctx *ctx, const ope_cloud *src, const ope_index *tgt, const float T[16], double max_range,
                double *score, double *sum_out, int64_t *n_out) {
  if (!ctx || !src || !tgt || !T) return set_err(ctx, OPE_EINVAL, "ope_fitness: bad argument");
  OPE_HIP(ctx, hipSetDevice(ctx->device));
  const int nblocks = (int)std::min<size_t>(std::max<size_t>((src->n_valid + 255) / 256, 1), 2048);
  int rc = ensure_scratch(ctx, 1 << 16);
  if (rc != OPE_OK) return rc;
  float rows[12];
  colmajor_to_rows(T, rows);
  float *d_T = static_cast<float *>(ctx->d_scratch);
  double *d_part = reinterpret_cast<double *>(static_cast<unsigned char *>(ctx->d_scratch) + 256);
  OPE_HIP(ctx, h2d_copy(ctx->stream, d_T, rows, sizeof rows));
  // getFitnessScore right after align (poseestimator.cpp:354-356): the run's start leaves are still there for this very pair
  const uint32_t *hint = (ctx->run_src == src && ctx->run_tgt == tgt && ctx->d_hint && ctx->corr_cap >= src->n && !ctx->run_active) ? ctx->d_hint : nullptr;
  launch_fitness(ctx->stream, nblocks, src->view(), tgt->view(), d_T, max_range, d_part, hint);
  std::vector<double> hp(2 * (size_t)nblocks);
  OPE_HIP(ctx, hipMemcpyAsync(hp.data(), d_part, sizeof(double) * hp.size(), hipMemcpyDeviceToHost, ctx->stream));
  OPE_HIP(ctx, hipStreamSynchronize(ctx->stream));
  double s = 0, c = 0;
  for (int b = 0; b < nblocks; ++b) { s += hp[2 * b]; c += hp[2 * b + 1]; }
  if (sum_out) *sum_out = s;
  if (n_out) *n_out = (int64_t)c;
  if (score) *score = c > 0 ? s / c : std::numeric_limits<double>::max();
  return OPE_OK;
}

int ope_rigid_transform_svd(ope_ctx *ctx, const float *src_xyz, const float *tgt_xyz, size_t n, float out_T[16]) {
  if (!ctx || !src_xyz || !tgt_xyz || !out_T || n < 1 || n > (size_t)0x7fffffff)
    return set_err(ctx, OPE_EINVAL, "ope_rigid_transform_svd: bad argument");
  OPE_HIP(ctx, hipSetDevice(ctx->device));
  const int nblocks = (int)std::min<size_t>((n + 255) / 256, 512);
  float *d_src = nullptr, *d_tgt = nullptr, *d_T = nullptr;
  double *d_part = nullptr;
  hipError_t e = hipMalloc((void **)&d_src, 12 * n);
  if (e == hipSuccess) e = hipMalloc((void **)&d_tgt, 12 * n);
  if (e == hipSuccess) e = hipMalloc((void **)&d_T, 64);
  if (e == hipSuccess) e = hipMalloc((void **)&d_part, sizeof(double) * kNumSums * nblocks);
  if (e == hipSuccess) e = h2d_copy(ctx->stream, d_src, src_xyz, 12 * n);
  if (e == hipSuccess) e = h2d_copy(ctx->stream, d_tgt, tgt_xyz, 12 * n);
  if (e == hipSuccess) {
    launch_pairs_svd(ctx->stream, d_src, d_tgt, (uint32_t)n, d_part, nblocks, d_T);
    e = hipMemcpyAsync(out_T, d_T, 64, hipMemcpyDeviceToHost, ctx->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
  }
  if (d_src) (void)hipFree(d_src);
  if (d_tgt) (void)hipFree(d_tgt);
  if (d_T) (void)hipFree(d_T);
  if (d_part) (void)hipFree(d_part);
  if (e != hipSuccess) return set_err(ctx, OPE_EHIP, std::string("ope_rigid_transform_svd: ") + hipGetErrorString(e));
  return OPE_OK;
}

int ope_transform_cloud(ope_ctx *ctx, const ope_cloud *cloud, const float T[16], float *out_xyz) {
  if (!ctx || !cloud || !T || !out_xyz) return set_err(ctx, OPE_EINVAL, "ope_transform_cloud: bad argument");
  { const int rch = cloud->ensure_host(); if (rch != OPE_OK) return rch; }
  for (size_t i = 0; i < cloud->n; ++i) {
    const float *p = &cloud->h_xyz[3 * i];
    float *o = out_xyz + 3 * i;
    if (!finite3(p)) { o[0] = p[0]; o[1] = p[1]; o[2] = p[2]; continue; }
    o[0] = T[0] * p[0] + T[4] * p[1] + T[8] * p[2] + T[12];
    o[1] = T[1] * p[0] + T[5] * p[1] + T[9] * p[2] + T[13];
    o[2] = T[2] * p[0] + T[6] * p[1] + T[10] * p[2] + T[14];
  }
  return OPE_OK;
}

}  // extern "C"
