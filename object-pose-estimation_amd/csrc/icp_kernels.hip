// icp_kernels.hip — the per-iteration ICP kernels (gfx950, wave64).
//
// One ICP iteration of IterativeClosestPoint::computeTransformation (vPCL impl/icp_mod.hpp:171-259)
// is two launches on one GPU:
//   icp_accumulate_kernel : steps 2-3-5a fused.  Each lane takes one source point, applies the
//       current final transform (never materialising the transformed cloud, icp_mod.hpp:246),
//       finds its exact nearest target point by BVH traversal (correspondence_estimation_mod.hpp:170),
//       applies the distance threshold (:171) and the optional normal-based rejectors
//       (correspondence_rejection_mod.h:368-391), stores the correspondence and accumulates the
//       17 sums TransformationEstimationSVD/umeyama needs.  Block sums are added into the run's sums
//       (fp64 atomics), or written to one row per block for the fixed-tree reduction (OPE_DETERMINISTIC_SUMS).
//   icp_reduce_update_kernel : fixed-order reduction of the block partials, then (one lane)
//       mean/covariance -> 3x3 Jacobi SVD in fp64 -> incremental T, final_T = T * final_T
//       (icp_mod.hpp:243-251) and DefaultConvergenceCriteria::hasConverged (:257).  The "done" flag
//       stays on the device; later launches of a batch early-out on it, so the host enqueues
//       iterations back-to-back and polls only every `check_every` iterations.
// Sharded (multi-GPU) runs split the second launch: reduce -> [all-reduce of S] -> update.
#include <cstdlib>
#include <hip/hip_ext.h>

#include "bvh_traverse.hpp"
#include "lm_solve.hpp"

namespace ope {

#ifdef OPE_DEVELOPER
// tools/chain_probe.py: per chunk {path: 0 per-lane / 1 packet / 2 groups, packet steps, packet leaf scans, packet back-ups}
__device__ uint32_t *g_chunk_stats = nullptr;
#endif

typedef const __attribute__((address_space(3))) float *lds_cfloat_ptr;   // a pointer that stays an LDS pointer
static_assert(kCertCand >= 1 && kCertCand <= 8, "skip certificates: candidates per query");

// ---- update launches overlapped with the accumulate launches (round 3; host side: api.hip, ope_icp_iterate) -------------
// In line, an iteration is accumulate -> update -> accumulate on one stream, and the two kernel boundaries around the
// 64-thread update launch cost the iteration 13-17 us (tools/gap_probe.sh).  Overlapped, update j is launched on a stream of
// its own next to accumulate launch j, is resident long before that launch ends and waits ON THE DEVICE for its blocks
// (chain[0], one ticket per block, taken after the block's sums are in); accumulate launch j + 1 follows launch j on the
// launch stream — the one boundary left runs while the update lane computes — and its blocks wait for update j's word
// (chain[1] = number of updates published) before they read the transform.  Every wait is bounded (wait_ticks of the
// 100 MHz clock, ope_ctx_set_wait_limit): a launch whose partner never shows up sets chain[2] / IcpState::chain_error and ends, so every wave of
// every launch reaches its exit.
// The two kernels run at the same time on different XCDs, each with an L2 of its own, so the handful of words they share —
// the three chain words, the sums, the transform rows, "done" — are only ever touched with agent-scope atomics (sc1: served
// at the point all XCDs agree on), ordered by s_waitcnt alone.  NO agent-scope fences: on this part a release fence writes the
// XCD's whole L2 back and an acquire fence invalidates it, and one such fence per block (the first build) evicted the
// L2-resident index under the blocks still walking it: 145 -> 260 us per launch.
// (the bound is a launch argument: ope_ctx_set_wait_limit, 2 s by default)
__device__ __forceinline__ uint32_t chain_load(const uint32_t *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void chain_store(uint32_t *p, uint32_t v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void chain_wait_own_memory_ops() { __builtin_amdgcn_s_waitcnt(0); }   // vmcnt(0) expcnt(0) lgkmcnt(0): this wave's stores and atomics have been acknowledged
// Head of an accumulate launch: in-line runs (chain == nullptr) early-out on the "done" flag; overlapped runs first wait for
// update chain_seq - 1 and then fetch the transform rows (s_const[0..11]) and "done" together, one round trip behind the word.
// Returns false (for the whole block) if the launch has nothing to do.
__device__ __forceinline__ bool acc_launch_begin(const IcpState *st, uint32_t *chain, uint32_t chain_seq, float *s_const, uint32_t wait_ticks) {
  if (chain == nullptr) return st->done == 0;
  __shared__ uint32_t s_go[2];
  if (threadIdx.x == 0) {
    uint32_t go = 1u;
    const unsigned long long t0 = wall_clock64();
    while ((int32_t)(chain_load(chain + 1) - chain_seq) < 0) {
      if (wall_clock64() - t0 > (unsigned long long)wait_ticks) { go = 0u; atomicOr(chain + 2, 1u); break; }
      __builtin_amdgcn_s_sleep(2);
    }
    s_go[0] = go;
  }
  __syncthreads();   // (the loads below are issued after the word has arrived)
  if (threadIdx.x < 12) s_const[threadIdx.x] = __hip_atomic_load(&st->Ff[threadIdx.x], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  else if (threadIdx.x == 12) s_go[1] = (uint32_t)__hip_atomic_load(&st->done, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  else if (threadIdx.x == 17) s_const[17] = __hip_atomic_load(&st->last_move, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  __syncthreads();
  return s_go[0] != 0u && s_go[1] == 0u;
}
// Tail of an accumulate launch: the block's sums are in (fp64 atomics) -> one ticket.
__device__ __forceinline__ void acc_launch_end(uint32_t *chain) {
  if (chain == nullptr) return;
  chain_wait_own_memory_ops();
  __syncthreads();
  if (threadIdx.x == 0) __hip_atomic_fetch_add(chain, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// One query's contribution to the wave's running sums {n, Σs, Σt, Σ t sᵀ, Σd²} (+ 27 normal-equation sums with the
// point-to-plane estimator): 16-lane row sums by DPP, then one ds_add_f64 per row and component into `acc` (LDS).
// t: the matched target point, tn: its normal (point-to-plane only); lanes with ok == false contribute zeros.
template <bool NRM>
__device__ __forceinline__ void add_query_sums(double *acc, lds_cfloat_ptr cs2, uint32_t lane_id, bool ok, bool p2p, float x, float y, float z,
                                               const float4 t, const float4 tn, float d2) {
    {
      // fp64 terms: differences and products of fp32 values are exact in fp64, so the 17 sums do not
      // depend (beyond 1e-16) on how queries are grouped into lanes, chunks, waves or ranks
      asm volatile("" : "+v"(cs2));
      const float psx = cs2[12], psy = cs2[13], psz = cs2[14];
      const double sx = (double)x - (double)psx, sy = (double)y - (double)psy, sz = (double)z - (double)psz;
      const double tx = (double)t.x - (double)psx, ty = (double)t.y - (double)psy, tz = (double)t.z - (double)psz;
      // 16-lane row sums by DPP (pure VALU), then one ds_add_f64 per row and component into the wave's
      // 17 LDS slots (a full 64-lane fp64 butterfly cost ~200 dependent ds_bpermutes per chunk)
      const double w = ok ? 1.0 : 0.0;
      double term[kNumSums];
      term[0] = w;
      term[1] = w * sx; term[2] = w * sy; term[3] = w * sz;
      term[4] = w * tx; term[5] = w * ty; term[6] = w * tz;
      term[7] = w * (tx * sx); term[8] = w * (tx * sy); term[9] = w * (tx * sz);
      term[10] = w * (ty * sx); term[11] = w * (ty * sy); term[12] = w * (ty * sz);
      term[13] = w * (tz * sx); term[14] = w * (tz * sy); term[15] = w * (tz * sz);
      term[16] = ok ? (double)d2 : 0.0;
#pragma unroll
      for (int k = 0; k < kNumSums; ++k) {
        const double r = row16_sum(term[k]);
        if ((lane_id & 15u) == 0u) unsafeAtomicAdd(acc + k, r);
      }
      if (NRM && p2p) {
        // TransformationEstimationPointToPlaneLLS: row v = (s x n, n), right-hand side d = n . (t - s);
        // the reference forms these products in float (operands are const float&), sums in double
        const float v0 = __fsub_rn(__fmul_rn(tn.z, y), __fmul_rn(tn.y, z));
        const float v1 = __fsub_rn(__fmul_rn(tn.x, z), __fmul_rn(tn.z, x));
        const float v2 = __fsub_rn(__fmul_rn(tn.y, x), __fmul_rn(tn.x, y));
        const float dd = __fsub_rn(__fsub_rn(__fsub_rn(__fadd_rn(__fadd_rn(__fmul_rn(tn.x, t.x), __fmul_rn(tn.y, t.y)), __fmul_rn(tn.z, t.z)),
                                                         __fmul_rn(tn.x, x)), __fmul_rn(tn.y, y)), __fmul_rn(tn.z, z));
        const double v[6] = {w * (double)v0, w * (double)v1, w * (double)v2, w * (double)tn.x, w * (double)tn.y, w * (double)tn.z};
        int slot = kNumSums;
#pragma unroll
        for (int r = 0; r < 6; ++r)
#pragma unroll
          for (int c = r; c < 6; ++c) {
            const double sum = row16_sum(v[r] * v[c]);
            if ((lane_id & 15u) == 0u) unsafeAtomicAdd(acc + slot, sum);
            ++slot;
          }
#pragma unroll
        for (int r = 0; r < 6; ++r) {
          const double sum = row16_sum(v[r] * (double)dd);
          if ((lane_id & 15u) == 0u) unsafeAtomicAdd(acc + slot, sum);
          ++slot;
        }
      }
    }
}

// ---- skip certificates: the two pieces the tree kernel and the grid kernel share (see icp_accumulate_kernel, "certified search")
// The check.  Returns true if the query at (x, y, z) is answered by its certificate: best / pos (position in the index's point
// order) then hold what a search would return.  stuck: the certificate failed where it was built (a new one would be no better).
__device__ __forceinline__ bool cert_check(const BvhView &tgt, const float4 *__restrict__ cert_q, const float *__restrict__ cert_l,
                                           const uint32_t *__restrict__ cert_pos, size_t cstride,
                                           uint32_t i, float x, float y, float z, float best_init, float last_move, float &best, uint32_t &pos,
                                           bool &stuck, bool &has_cert) {
  stuck = false;
  const uint32_t p0 = cert_pos[i];
  has_cert = p0 != 0u;
  if (!has_cert) return false;
  const v4f c = ld16(cert_q + i);   // {q_ref, L2}
  const v4f t1 = ld16(tgt.pts + (p0 - 1u));
  const float d1 = sq_dist3(__fsub_rn(x, t1.x), __fsub_rn(y, t1.y), __fsub_rn(z, t1.z));
  const float dl = __builtin_amdgcn_sqrtf(sq_dist3(__fsub_rn(x, c.x), __fsub_rn(y, c.y), __fsub_rn(z, c.z)));
  const float dl_up = __fmaf_rn(dl, 1.000001f, 1e-30f);
  // ---- first tier: c_1 alone.  From q_ref every OTHER target point — the other candidates included — was at least L2 (the
  // runner-up's distance) away; L2 - dl rounded DOWN (one float below the rounded difference; a query that sits where it walked
  // keeps its bound).  Most queries of a settled scene end here: 20 bytes and one gather.
  {
    const float A = dl == 0.f ? c.w : __uint_as_float(__float_as_uint(__fsub_rn(c.w, dl_up)) - 1u);
    const float TA = __fmul_rn(__fmul_rn(A, A), 0.999999f);
    if (A > 0.f && d1 < TA && d1 < best_init) { best = d1; pos = p0 - 1u; return true; }
  }
  // ---- second tier: the nearest of all candidates against L, the bound on every non-candidate
  uint32_t pj[kCertCand];
#pragma unroll
  for (int j = 1; j < kCertCand; ++j) pj[j] = cert_pos[(size_t)j * cstride + i];
  float b1 = d1;   // the nearest candidate, from where the query is now
  uint32_t bp = p0 - 1u;
#pragma unroll
  for (int j = 1; j < kCertCand; ++j) {
    const v4f tp = ld16(tgt.pts + ((pj[j] != 0u ? pj[j] : p0) - 1u));
    const float dj = pj[j] != 0u ? sq_dist3(__fsub_rn(x, tp.x), __fsub_rn(y, tp.y), __fsub_rn(z, tp.z)) : INFINITY;
    const bool lt = dj < b1;
    bp = lt ? pj[j] - 1u : bp;
    b1 = lt ? dj : b1;
  }
  const float Lq = cert_l[i];
  const float L1 = dl == 0.f ? Lq : __uint_as_float(__float_as_uint(__fsub_rn(Lq, dl_up)) - 1u);
  const float T = __fmul_rn(__fmul_rn(L1, L1), 0.999999f);
  // (Two candidates at the very same computed d2: which of them a search returns is a matter of its visiting order — the
  // reference's kd-tree, the walks here and the oracle each have their own — and the distance is the same bit for bit: the
  // certificate takes the first in its list, as the tests allow every search on exact ties.)
  if (L1 > 0.f && b1 < T && b1 < best_init) { best = b1; pos = bp; return true; }
  stuck = !(dl > 4.0f * last_move);
  return false;
}
// The walk that builds one: the kCertCand + 1 nearest from (x, y, z), start leaf h.  best / pos / leaf: the nearest (what the 1-NN
// search returns); `write`: this lane records the certificate.
__device__ __forceinline__ void cert_build(const BvhView &tgt, float4 *__restrict__ cert_q, float *__restrict__ cert_l, uint32_t *__restrict__ cert_pos, size_t cstride, uint32_t i,
                                           float x, float y, float z, float best_init, uint32_t h, float *stk, int stk_stride, bool build, bool write,
                                           float &best, uint32_t &pos, uint32_t &leaf) {
  constexpr int K = kCertCand + 1;
  KnnRegVisitor<K> kv;
  kv.init(build, best_init);
  if (build) bvh_traverse(tgt, x, y, z, kv, stk, stk_stride, h);
  if (build) {
    best = kv.count > 0 ? kv.d[0] : best_init;
    pos = kv.count > 0 ? kv.p[0] : kNoPos;
    leaf = kv.count > 0 ? kv.leaf : h;
    if (write) {
      // every point outside the list has a computed d2 of at least d[K-1] (a list that is not full still holds the walk's
      // starting bound there) -> a true distance of at least its square root less 2.5u; the square root is good to 1 ulp
      const float Lw = __fmul_rn(__builtin_amdgcn_sqrtf(kv.d[K - 1]), 0.9999995f);
      const float L2w = __fmul_rn(__builtin_amdgcn_sqrtf(kv.d[1]), 0.9999995f);   // the same for everything but the nearest (first tier)
#pragma unroll
      for (int j = 0; j < kCertCand; ++j) cert_pos[(size_t)j * cstride + i] = j < kv.count ? kv.p[j] + 1u : 0u;
      cert_q[i] = make_float4(x, y, z, L2w);
      cert_l[i] = Lw;
    }
  }
}
// Whether a query that has to walk builds a certificate: what one would be worth — its slack, distance of the (kCertCand + 1)-th
// neighbour less the nearest's, ~ cert_k / D for a query D from the surface, at most cert_cap — against kCertWorth launches of the
// scene's displacement (the displacements of a converging run sum to about a dozen times the current one: a certificate built then
// usually holds to the end).  Surface points build soon after the launches start keeping certificates; a point 10 cm out, whose
// neighbours all lie within microns of each other in distance, once the scene moves by less than a micron per launch.
__device__ __forceinline__ bool cert_worth_building(float d2_prev, float cert_k, float cert_cap, float last_move) {
  const float dprev = __builtin_amdgcn_sqrtf(d2_prev);           // (+inf before the first match: no slack, no certificate yet)
  return fminf(cert_cap, cert_k * __builtin_amdgcn_rcpf(dprev)) >= kCertWorth * last_move;
}

// MODE 0: 1-NN correspondences.  MODE 2: normal shooting over the k nearest, list in KREG >= k registers (every k <= 32:
// KREG = k rounded up to a multiple of four, plus 10, the class default of vPCL
// correspondence_estimation_normal_shooting_weighted.h:117; the reference uses 20, poseestimator.cpp:246,
// regmeshpcd.cpp:144), walk started at last iteration's leaf.  (MODE 1, the LDS list of round 1, is no longer dispatched.)
// NRM: source/target normals present (rejectors and/or normal shooting).
// RECIP: reciprocal correspondences (vPCL impl/correspondence_estimation_mod.hpp:216-303): keep (i, j) only
// if the nearest SOURCE point of target point j is i again.  The reference searches a kd-tree rebuilt
// over the transformed source every iteration; here the source index is built once in the source's own
// frame and queried with F^-1 * t_j (a rigid map preserves the ranking up to fp32 rounding).
#ifdef OPE_KNN_STATS
__device__ unsigned long long g_knn_stats[8];
#endif
template <int MODE, bool NRM, bool RECIP = false, bool PACKET = false, int KREG = 20, bool CERT = false>
#ifndef OPE_KNN_WAVES
#define OPE_KNN_WAVES 4   // waves per SIMD of the k-NN instantiations (developer A/B: make VARIANT=... EXTRA=-DOPE_KNN_WAVES=3).
// k = 20 on C3, tools/ns_bench.py: 4 waves (128 VGPRs, 9 spilled) 0.545-0.554 ms per iteration; 3 waves (139 VGPRs, none spilled) 0.68;
// 5 waves (96 VGPRs, 53 spilled) 0.90: the walks need the fourth wave more than their nine registers
#endif
__global__ __launch_bounds__(MODE == 0 ? kAccBlock : kKnnBlock, (MODE == 0 && !RECIP && !CERT) ? kAccWavesPerSimd : (MODE == 2 ? OPE_KNN_WAVES : 4)) void icp_accumulate_kernel(
    CloudView src, BvhView tgt, BvhView srcix, const IcpState *__restrict__ st, double *__restrict__ partials,
    int32_t *__restrict__ corr_match, float *__restrict__ corr_d2, uint32_t *__restrict__ work_counter,
    uint32_t *__restrict__ hint, const uint32_t *__restrict__ chunk_order, uint32_t *__restrict__ chunk_cost,
    const uint32_t *__restrict__ plan_info, double *__restrict__ S_atomic, const uint32_t *__restrict__ slot_list,
    float *__restrict__ knn_rk, const uint32_t *__restrict__ plan_out, uint32_t measuring_launch, uint32_t *chain_arg, uint32_t chain_seq,
    float4 *__restrict__ cert_q, uint32_t *__restrict__ cert_pos, uint32_t *pace, uint32_t launch_no, uint32_t wait_ticks, float *__restrict__ cert_l) {
  // pace (host-visible): "launch launch_no has started", i.e. every launch before it is over — the host keeps a bounded lead
  // over the GPU by it (api.hip: pace_wait), which is what lets it notice, a few launches late at most, that the update step
  // has asked for certifying launches (IcpState::cert_mode -> host_cert)
  if (pace != nullptr && blockIdx.x == 0 && threadIdx.x == 0) __hip_atomic_store(pace, launch_no, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  uint32_t *const chain = (MODE == 0 && !RECIP) ? chain_arg : nullptr;   // overlapped update launches exist for the plain 1-NN run only (api.hip)
  static_assert(!CERT || (MODE == 0 && !RECIP), "skip certificates: plain 1-NN only");
  // Per-run constants live in LDS and are re-read where they are used (through a pointer the optimiser cannot see
  // through): held in registers across the walk they were spilled to scratch, 13 dwords per lane and launch.
  __shared__ __attribute__((aligned(16))) float s_const[20];   // F rows [0..11], pivot [12..14], best0 [15], last_move [17], cert_k [18], cert_cap [19] (CERT)
  __shared__ uint32_t s_ncert;
  if (!acc_launch_begin(st, chain, chain_seq, s_const, wait_ticks)) return;
  constexpr int BLOCK = (MODE == 0) ? kAccBlock : kKnnBlock;
  constexpr bool OCT_OK = (MODE == 0) && !RECIP;  // the group traversal exists for plain 1-NN only
  __shared__ double s_red[BLOCK / 64][kNumSumsMax];
  __shared__ float s_stk[kMaxDepth + 2][BLOCK];  // pending-sibling bounds of the traversal (rows 1 .. D) and, above them, the deferred walk's leaf queue
  float *stk = &s_stk[0][threadIdx.x];
  extern __shared__ unsigned char s_dyn[];  // MODE 1: k-NN lists

  const double max_d2 = st->max_d2;
  // With a finite setMaxCorrespondenceDistance the 1-NN search only has to see points that can survive the
  // threshold test (correspondence_estimation_mod.hpp:171 rejects d2 > max_d2 afterwards anyway): start from the
  // smallest float strictly above every admissible d2 instead of +inf, so far-off queries end at the first boxes.
  float best0 = INFINITY;
  if (max_d2 < 3.0e38) {
    float f = (float)max_d2;
    if ((double)f < max_d2) f = nextafterf(f, INFINITY);
    best0 = nextafterf(f, INFINITY);
  }
  if (threadIdx.x < 12) { if (chain == nullptr) s_const[threadIdx.x] = st->Ff[threadIdx.x]; }   // (overlapped: fetched by acc_launch_begin)
  else if (threadIdx.x < 15) s_const[threadIdx.x] = (float)st->pivot[threadIdx.x - 12];
  else if (threadIdx.x == 15) s_const[15] = best0;
  else if (threadIdx.x == 17) { if (chain == nullptr) s_const[17] = st->last_move; }
  else if (threadIdx.x == 30) s_ncert = 0u;
  __syncthreads();
  if (MODE == 2 && knn_rk != nullptr && blockIdx.x == 0 && threadIdx.x == 0) const_cast<IcpState *>(st)->knn_acc_flag = 1;
  // ---- skip certificates (plain 1-NN; ope.h: ope_icp_params.skip_certificates).  A certificate is a statement about the
  // target alone: "from q_ref, the kCertCand points {c_j} are the nearest and every other target point is at least L away" — the
  // outcome of ONE (kCertCand + 1)-nearest walk from q_ref.  It says nothing about transforms or launches, so whatever happened
  // in between (updates, step-wise calls, a change of kernel and back) it proves what it proves about where the query is NOW.
  // The CERT instantiation is launched once the update step has asked for it (IcpState::cert_mode, seen by the host through
  // host_cert): the instantiation without carries none of this.
  constexpr bool cert_on = CERT;
  if (cert_on && threadIdx.x == 0) {
    s_const[18] = st->cert_k;
    s_const[19] = st->cert_cap;
    if (blockIdx.x == 0) atomicAdd(work_counter + 42, 1u);
  }
  if (cert_on) __syncthreads();
  const bool rej_sn = NRM && st->use_surface_normal_rej;
  const bool rej_so = NRM && st->use_self_occluded_rej;
  // the LM estimator (lm.hip) re-reads the matched target points: corr_match then holds their POSITION in the index
  const bool store_pos = st->estimator == OPE_EST_POINT_TO_PLANE_LM;
  const double thr_sn = st->surface_normal_thr, thr_so = st->self_occluded_thr;
  const double max_dist_unsq = st->max_corr_dist;
  const int kk = st->k_normal_shooting;

  // Running sums {n, Σs, Σt, Σ t sᵀ, Σd²}: one set of 17 fp64 LDS slots per wave (s_red), so no
  // accumulator register stays live across the traversal.
  if ((threadIdx.x & 63u) < (uint32_t)kNumSumsMax) s_red[threadIdx.x >> 6][threadIdx.x & 63u] = 0.0;
  const bool p2p = NRM && st->estimator == OPE_EST_POINT_TO_PLANE_LLS;

  const uint32_t lane_id = threadIdx.x & 63u;
  // Static, cost-aware work distribution.  Query cost is very uneven (a clutter point far from the model
  // walks 10-40x more nodes than a surface point) and the kernel ends with its slowest wave, so the
  // 64-query chunks of the Morton order are dealt to the waves in "snake" order over a list sorted by
  // the cost each chunk measured in an earlier iteration (heaviest chunks first, each wave's later
  // chunks progressively lighter): longest-processing-time-first scheduling with no atomics.  (A
  // device-wide ticket counter was tried first: ~24 k returning atomics on one word cost 130 us.)
  // The n_heavy costliest chunks (clutter: long private walks) are split into eight slots each and walked
  // by 8-lane groups (bvh_traverse_oct); all other chunks take one slot and one lane per query.
  const uint32_t n_waves = gridDim.x * (BLOCK / 64);
  const uint32_t wave_id = blockIdx.x * (BLOCK / 64) + (threadIdx.x >> 6);
  const uint32_t n_chunks = (src.n_valid + 63u) / 64u;
  // measuring_launch (a kernel argument: the host knows it when it launches): the launch before a plan step — every chunk takes the per-lane walk, so that the
  // costs the plan sorts are all of one kind and none is older than one launch
  const bool measuring = chunk_order && measuring_launch != 0u;
  // plan_out: what the plan step computed for this chunk order ([0] chunks walked by groups, [5] / [7] chunks with a wave to
  // themselves); plan_info: the launch's flags
  const uint32_t n_heavy = (OCT_OK && chunk_order && !measuring) ? min(plan_out[0], n_chunks) : 0u;
  // The plan's costliest per-lane chunks each outlast the share of work a wave has in a balanced launch (plan_info[5] of
  // them, ranks n_heavy .. n_heavy + n_alone - 1): each gets a wave to itself (waves 0 .. n_alone - 1) and that wave
  // takes nothing else; all other slots are dealt to the remaining waves in snake order.
  // With group walks in the plan the slots come as a list in descending order of expected duration (plan_slots_kernel:
  // the eight slots of a group-walked chunk merged in among the per-lane chunks they are as long as), its first
  // plan_info[7] entries being the ones that get a wave to themselves.
  const bool listed = OCT_OK && slot_list != nullptr && chunk_order != nullptr && !measuring && n_heavy > 0u;
  const uint32_t n_alone = !chunk_order ? 0u
                           : listed   ? min(plan_out[7], n_waves / 2u)
                                      : min(min(plan_out[5] + (measuring ? min(plan_out[0], n_chunks) : 0u), n_chunks - n_heavy), n_waves / 2u);
  const uint32_t n_snake = listed ? (n_chunks + 7u * n_heavy - n_alone) : (8u * n_heavy + (n_chunks - n_heavy - n_alone));
  const uint32_t snake_waves = n_waves - n_alone;
  for (uint32_t round = 0;; ++round) {
    bool oct = false;
    uint32_t ord, sub = 0;
    uint32_t slot;
    if (wave_id < n_alone) {
      if (round > 0) break;
      slot = wave_id;
      ord = n_heavy + wave_id;
    } else {
      if (round * snake_waves >= n_snake) break;
      const uint32_t w = wave_id - n_alone;
      slot = round * snake_waves + ((round & 1u) ? (snake_waves - 1u - w) : w);
      if (slot >= n_snake) continue;
      oct = slot < 8u * n_heavy;
      ord = oct ? (slot >> 3) : (n_heavy + n_alone + (slot - 8u * n_heavy));
      sub = slot & 7u;
      slot += n_alone;
    }
    if (listed) {
      const uint32_t e = slot_list[slot];
      oct = (e >> 31) != 0u;
      sub = (e >> 28) & 7u;
      ord = e & 0x0fffffffu;
    }
    const uint32_t chunk = chunk_order ? chunk_order[ord] : ord;
    const uint32_t base = chunk * 64u;
    const unsigned long long t_begin = __builtin_amdgcn_s_memtime();
    const uint32_t i = oct ? (base + sub * 8u + (lane_id >> 3)) : (base + lane_id);
    const bool active = i < src.n_valid;
    const bool owner = active && (!oct || (lane_id & 7u) == 0u);  // the one lane that reports a query
    // (the pointer keeps its LDS address space: through a generic pointer these became flat_loads, which take the
    // vector-memory path and wait on both counters)
    lds_cfloat_ptr cst = (lds_cfloat_ptr)s_const;
    asm volatile("" : "+v"(cst));   // keep the constants' loads here, inside the chunk loop
    float F[12];
#pragma unroll
    for (int k = 0; k < 12; ++k) F[k] = cst[k];
    const float4 s = src.xyzw[active ? i : base];
    const float x = xform_row(F + 0, s.x, s.y, s.z);
    const float y = xform_row(F + 4, s.x, s.y, s.z);
    const float z = xform_row(F + 8, s.x, s.y, s.z);
    float nx = 0.f, ny = 0.f, nz = 0.f;
    if (NRM && src.nrm != nullptr) {
      const float4 n4 = src.nrm[active ? i : base];
      nx = rot_row(F + 0, n4.x, n4.y, n4.z);
      ny = rot_row(F + 4, n4.x, n4.y, n4.z);
      nz = rot_row(F + 8, n4.x, n4.y, n4.z);
    }
    bool ok;
    float d2;
    uint32_t pos = 0;
    int match = -1;
    if constexpr (CERT) {
      // ---- certified search.  {q_ref, L} and the candidates c_j: when this query last built a certificate, from q_ref, every
      // target point but the candidates was at least L away.  The query is dl from q_ref now, so every non-candidate is at least
      // L - dl away; if the nearest candidate, re-measured with the search's own arithmetic, is strictly nearer than that — past
      // the rounding of every quantity involved (u = 2^-24: a computed d2 is within 5u of the true one, v_sqrt_f32 within 1 ulp;
      // the factors below leave 16u) — and strictly nearer than every other candidate, it is the unique nearest neighbour: what
      // a walk would return, bit for bit.  Nothing is written but the correspondence.
      const float best_init = active ? cst[15] : -INFINITY;
      const size_t cstride = src.n;
      bool need = active, stuck = false, has_cert = false;
      float c_best = INFINITY;
      uint32_t c_pos = 0u;
      if (active) need = !cert_check(tgt, cert_q, cert_l, cert_pos, cstride, i, x, y, z, best_init, cst[17], c_best, c_pos, stuck, has_cert);
#ifdef OPE_DEVELOPER   // tools/cert_probe.py: why certificates fail {expired, stuck where it was built, none yet}
      if (need && owner) atomicAdd(work_counter + (!has_cert ? 46 : stuck ? 45 : 44), 1u);
#endif
      // (in a slot walked by 8-lane groups the eight lanes of a group carry the same query: one of them builds)
      const bool build = need && !stuck && owner && cert_worth_building(corr_d2[i], cst[18], cst[19], cst[17]);
      const uint32_t h = need ? hint[i] : 0u;   // start leaf of the walks (queries answered from their certificate need none)
#ifdef OPE_DEVELOPER
      if (owner && build) atomicAdd(work_counter + 47, 1u);   // walks that build a certificate
#endif
      const bool fast = need && !build && !(oct && __shfl((int)build, (int)(lane_id & ~7u), 64) != 0);   // (a group whose query builds does not walk as well)
      NearestVisitor v{fast ? best_init : -INFINITY, kNoPos, 0};
      const unsigned long long fmask = __ballot(fast);
      if (fmask != 0ull) {
        if (oct) {
          if (fast) bvh_traverse_oct(tgt, x, y, z, v, &s_stk[0][threadIdx.x & ~7u], BLOCK, h);
        } else if (__popcll(fmask) <= 8) {
          // a handful of walkers in a chunk that is otherwise answered from certificates: eight lanes walk for each of them
          // (bvh_traverse_oct: a third of the dependent trips of a private walk — a launch in which next to nobody walks would
          // otherwise last as long as its one longest private walk).  Group r serves the r-th walker.
          const uint32_t r = lane_id >> 3;
          const bool serve = r < (uint32_t)__popcll(fmask);
          unsigned long long mm = fmask;
          for (uint32_t k = 0; k < r && mm != 0ull; ++k) mm &= mm - 1ull;
          const int from = mm != 0ull ? (int)__builtin_ctzll(mm) : 0;
          const float gx = __shfl(x, from, 64), gy = __shfl(y, from, 64), gz = __shfl(z, from, 64);
          const uint32_t gh = (uint32_t)__shfl((int)h, from, 64);
          NearestVisitor gv{serve ? cst[15] : -INFINITY, kNoPos, 0};
          if (serve) bvh_traverse_oct(tgt, gx, gy, gz, gv, &s_stk[0][threadIdx.x & ~7u], BLOCK, gh);
          // the r-th walker takes its result from lane 8 r
          const int back = 8 * (int)__popcll(fmask & ((1ull << lane_id) - 1ull));
          const float rb = __shfl(gv.best, back, 64);
          const uint32_t rp = (uint32_t)__shfl((int)gv.pos, back, 64), rl = (uint32_t)__shfl((int)gv.leaf, back, 64);
          if (fast) { v.best = rb; v.pos = rp; v.leaf = rl; }
        } else {
          const bool done = PACKET && bvh_traverse_packet(tgt, x, y, z, fast, v, h, stk, BLOCK);
          if (!done && fast) bvh_traverse_deferred(tgt, x, y, z, v, stk, BLOCK, h, min(8, kMaxDepth + 1 - tgt.depth));
        }
      }
      float r_best = fast ? v.best : c_best;
      uint32_t r_pos = fast ? v.pos : c_pos, r_leaf = fast ? v.leaf : h;
      if (__ballot(build) != 0ull) cert_build(tgt, cert_q, cert_l, cert_pos, cstride, i, x, y, z, best_init, h, stk, BLOCK, build, build, r_best, r_pos, r_leaf);
      {
        const uint32_t nc = (uint32_t)__popcll(__ballot(owner && !need));
        if (lane_id == 0 && nc != 0u) atomicAdd(&s_ncert, nc);
      }
      if (owner && r_leaf != h) hint[i] = r_leaf;
      const bool found = active && r_pos != kNoPos;
      pos = found ? r_pos : 0;
      ok = found && !((double)r_best > max_d2);
      d2 = found ? r_best : INFINITY;
      match = found ? (store_pos ? (int)pos : __float_as_int(tgt.pts[pos].w)) : -1;
    } else if constexpr (MODE == 0) {
      NearestVisitor v{active ? cst[15] : -INFINITY, kNoPos, 0};
      // start at the leaf that held this query's nearest neighbour one iteration ago (0 = none yet)
      const uint32_t h = active ? hint[i] : 0u;
      if (OCT_OK && oct) {
        if (active) bvh_traverse_oct(tgt, x, y, z, v, &s_stk[0][threadIdx.x & ~7u], BLOCK, h);
#ifdef OPE_DEVELOPER
        if (g_chunk_stats && lane_id == 0) {   // path, and the longest of the eight slots' walks (same units as chunk_cost)
          g_chunk_stats[4 * (size_t)chunk] = 2u;
          atomicMax(g_chunk_stats + 4 * (size_t)chunk + 1, (uint32_t)((__builtin_amdgcn_s_memtime() - t_begin) >> 4));
        }
#endif
      } else {
        // coherent chunks (a handful of start leaves for 64 queries) take one packet walk through the scalar cache
        // (PACKET instantiation: launches that fill the GPU); everything else the per-lane walk from its own leaf
#ifdef OPE_DEVELOPER
        PacketStats pst{0, 0, 0};
        const bool done = PACKET && bvh_traverse_packet(tgt, x, y, z, active, v, h, stk, BLOCK, &pst);
        if (g_chunk_stats && lane_id == 0) {
          uint32_t *o = g_chunk_stats + 4 * (size_t)chunk;
          o[0] = done ? 1u : 0u; o[1] = pst.steps; o[2] = pst.leaves; o[3] = pst.backups;
        }
#else
        const bool done = PACKET && bvh_traverse_packet(tgt, x, y, z, active, v, h, stk, BLOCK);
#endif
        // (the queue of deferred leaves lives in the rows of the parked-bound column above the tree's depth: 8 entries up to
        // 2^13 leaves, at least one always)
        if (!done && active) bvh_traverse_deferred(tgt, x, y, z, v, stk, BLOCK, h, min(8, kMaxDepth + 1 - tgt.depth));
      }
      if (owner && v.leaf != h) hint[i] = v.leaf;   // most start leaves survive an iteration: 4 MB of writes saved on C3
      const bool found = active && v.pos != kNoPos;
      ok = found && !((double)v.best > max_d2);
      d2 = found ? v.best : INFINITY;
      pos = found ? v.pos : 0;
      match = found ? (store_pos ? (int)pos : __float_as_int(tgt.pts[pos].w)) : -1;
      if (RECIP) {
        float G[12];
#pragma unroll
        for (int k = 0; k < 12; ++k) G[k] = st->Finv[k];
        const float4 t = tgt.pts[pos];
        const float bx = xform_row(G + 0, t.x, t.y, t.z), by = xform_row(G + 4, t.x, t.y, t.z), bz = xform_row(G + 8, t.x, t.y, t.z);
        NearestVisitor r{ok ? cst[15] : -INFINITY, kNoPos, 0};
        if (ok) bvh_traverse(srcix, bx, by, bz, r, stk, BLOCK);
        ok = ok && r.pos != kNoPos && !((double)r.best > max_d2) &&
             __float_as_int(srcix.pts[r.pos].w) == __float_as_int(s.w);
      }
    } else if constexpr (MODE == 2) {
      // register list of KREG >= k entries: the k nearest are the first k of the KREG nearest
      constexpr int K = KREG;
      KnnRegVisitor<K> v;
      // Search bound from the previous launch: the K list entries of that launch lay within sqrt(rk) of the query as it
      // was then, so they lie within sqrt(rk) + |displacement| of the query now, and so does this launch's K-th
      // neighbour: only points inside that ball can enter the list.  Exact (the bound is inflated past every fp32
      // rounding of the distances it is compared with); should a list still come back short, that lane searches again
      // without a bound.  Insertions and box tests for everything beyond the ball fall away: normal shooting k = 20 on C3
      // spends most of its time on list insertions of points that do not stay in the list.
      float bound0 = INFINITY;
      if (knn_rk != nullptr && active && st->have_prev) {
        const float rk = knn_rk[i];
        if (rk < 3.0e38f) {
          const float ox = xform_row(st->Fprev + 0, s.x, s.y, s.z), oy = xform_row(st->Fprev + 4, s.x, s.y, s.z), oz = xform_row(st->Fprev + 8, s.x, s.y, s.z);
          const float delta = sqrtf(sq_dist3(x - ox, y - oy, z - oz));
          const float b = sqrtf(rk) * 1.00001f + delta * 1.00001f + 1e-6f * (fabsf(x) + fabsf(y) + fabsf(z)) + 1e-30f;
          bound0 = b * b * 1.00001f;
        }
      }
      const uint32_t h0 = active ? hint[i] : 0u;
      bool todo = active;
      for (;;) {
        if (todo) {
          v.init(true, bound0);
          bvh_traverse(tgt, x, y, z, v, stk, BLOCK, h0);
        }
        const bool retry = todo && bound0 < 3.0e38f && v.count < K;   // (cannot happen while the bound holds; kept for exactness)
        if (__ballot(retry) == 0ull) break;
        todo = retry;
        bound0 = INFINITY;
      }
#ifdef OPE_KNN_STATS
      {
        unsigned long long t[5];
        for (int j = 0; j < 5; ++j) { t[j] = v.stat[j]; for (int off = 32; off >= 1; off >>= 1) t[j] += __shfl_xor(t[j], off, 64); }
        if (lane_id == 0) { for (int j = 0; j < 5; ++j) atomicAdd(&g_knn_stats[j], t[j]); atomicAdd(&g_knn_stats[5], 1ull); }
      }
#endif
      if (!active) v.init(false);
      if (active) {
        hint[i] = v.leaf;
        if (knn_rk != nullptr) knn_rk[i] = v.count >= K ? v.d[K - 1] : INFINITY;
      }
      // among the k nearest, the one with the smallest squared distance to the line (s, n)
      // (…normal_shooting_weighted.hpp:115-135; cross product in double)
      double min_dist = 1.79769313486231570815e308;
      d2 = INFINITY;
      pos = 0;
#pragma unroll
      for (int j = 0; j < K; ++j) {
        if (j < v.count && j < kk) {
          const float4 p = tgt.pts[v.p[j]];
          const double vx = (double)__fsub_rn(p.x, x), vy = (double)__fsub_rn(p.y, y), vz = (double)__fsub_rn(p.z, z);
          const double cx = (double)ny * vz - (double)nz * vy;
          const double cy = (double)nz * vx - (double)nx * vz;
          const double cz = (double)nx * vy - (double)ny * vx;
          const double dist = cx * cx + cy * cy + cz * cz;
          if (dist < min_dist) { min_dist = dist; d2 = v.d[j]; pos = v.p[j]; }
        }
      }
      // quirk Q2: squared line distance against the UNSQUARED max distance (:136)
      ok = active && v.count > 0 && !(min_dist > max_dist_unsq);
      match = v.count > 0 ? (store_pos ? (int)pos : __float_as_int(tgt.pts[pos].w)) : -1;
    } else {
      float *ld = reinterpret_cast<float *>(s_dyn) + threadIdx.x;
      uint32_t *lp = reinterpret_cast<uint32_t *>(s_dyn + sizeof(float) * BLOCK * kKnnMaxK) + threadIdx.x;
      KnnVisitor v{ld, lp, BLOCK, kk, 0, active ? INFINITY : -INFINITY};
      bvh_traverse(tgt, x, y, z, v, stk, BLOCK);
      // among the k nearest, the one with the smallest squared distance to the line (s, n)
      // (…normal_shooting_weighted.hpp:115-135; cross product in double)
      double min_dist = 1.79769313486231570815e308;
      int min_j = 0;
      for (int j = 0; j < v.count; ++j) {
        const float4 p = tgt.pts[lp[j * BLOCK]];
        const double vx = (double)__fsub_rn(p.x, x), vy = (double)__fsub_rn(p.y, y), vz = (double)__fsub_rn(p.z, z);
        const double cx = (double)ny * vz - (double)nz * vy;
        const double cy = (double)nz * vx - (double)nx * vz;
        const double cz = (double)nx * vy - (double)ny * vx;
        const double dist = cx * cx + cy * cy + cz * cz;
        if (dist < min_dist) { min_dist = dist; min_j = j; }
      }
      // quirk Q2: squared line distance against the UNSQUARED max distance (:136)
      ok = active && v.count > 0 && !(min_dist > max_dist_unsq);
      d2 = v.count > 0 ? ld[min_j * BLOCK] : INFINITY;
      pos = v.count > 0 ? lp[min_j * BLOCK] : 0;
      match = v.count > 0 ? (store_pos ? (int)pos : __float_as_int(tgt.pts[pos].w)) : -1;
    }
    if (NRM && ok && rej_sn) {
      const float4 tn = tgt.nrm[pos];
      const float score = __fadd_rn(__fadd_rn(__fmul_rn(nx, tn.x), __fmul_rn(ny, tn.y)), __fmul_rn(nz, tn.z));
      ok = (double)score > thr_sn;
    }
    if (NRM && ok && rej_so) {
      const double sl = sqrt((double)__fadd_rn(__fadd_rn(__fmul_rn(x, x), __fmul_rn(y, y)), __fmul_rn(z, z)));
      const double score = (double)nx * (-(double)x / sl) + (double)ny * (-(double)y / sl) + (double)nz * (-(double)z / sl);
      ok = score > thr_so;
    }
    ok = ok && owner;
    if (owner) {
      __builtin_nontemporal_store(ok ? match : -1, corr_match + i);   // streamed out: nothing re-reads them in this launch
      __builtin_nontemporal_store(d2, corr_d2 + i);
    }
    add_query_sums<NRM>(s_red[threadIdx.x >> 6], (lds_cfloat_ptr)s_const, lane_id, ok, p2p, x, y, z, tgt.pts[ok ? pos : 0],
                        (NRM && p2p) ? tgt.nrm[ok ? pos : 0] : make_float4(0.f, 0.f, 0.f, 0.f), d2);
    if (lane_id == 0 && !oct) chunk_cost[chunk] = (uint32_t)((__builtin_amdgcn_s_memtime() - t_begin) >> 4);
  }

  // wave slots -> block partial, fixed order
  __syncthreads();
  if (threadIdx.x < (p2p ? kNumSumsMax : kNumSums)) {
    double v = 0.0;
#pragma unroll
    for (int w = 0; w < BLOCK / 64; ++w) v += s_red[w][threadIdx.x];
    // One-GPU runs keep one row per block and reduce the rows in a fixed tree (bit-reproducible for a launch
    // geometry).  Sharded runs add straight into the 17 (44) sums that the all-reduce takes next: one launch and one
    // kernel boundary less per iteration, at the price of an addition order that varies from run to run (1e-16).
    if (S_atomic != nullptr) unsafeAtomicAdd(S_atomic + threadIdx.x, v);
    else partials[threadIdx.x * kAccMaxBlocks + blockIdx.x] = v;
  }
  if (CERT && threadIdx.x == 64 && s_ncert != 0u)
    atomicAdd(reinterpret_cast<unsigned long long *>(work_counter + 40), (unsigned long long)s_ncert);   // ope_icp_certificate_stats
  acc_launch_end(chain);
}

// ------------------------------------------------------------------------------------------
// GRID instantiation of the 1-NN accumulate step (north_star's "radix-bucketed nearest neighbour").
//
// The typical ICP query is a scene point whose previous match t_prev is still (nearly) its nearest neighbour.  Then
// every model point that can beat it lies in the ball |x - q| <= |q - t_prev|, and the ball overlaps only a handful of
// cells of the uniform grid over the model (grid_build.hip): at most 9 x-runs of at most kGridMaxSpan cells,
// each ONE contiguous run of the cell-sorted points.  Exact by construction: the cell range of the ball is computed with
// the same monotone fp32 expression that assigned the points to cells, from a radius inflated past every rounding, and
// every point of those runs is tested with the oracle's unfused distance.  Two dependent fetches (cell bounds, points)
// instead of the ~15 of a tree walk, ~20-60 distance evaluations, no LDS.
// Queries the grid cannot answer (no previous match yet, or a ball that spans more cells: clutter far from the surface)
// take the OBB-tree walk of bvh_traverse.hpp, seeded with the previous match's distance.  So that those few do not hold
// up whole waves of grid queries, the launch works through a query ORDER (qorder) that lists the grid-class queries
// first and the tree-class queries after them, re-partitioned from the per-query class flags at the plan steps; a query
// whose class has changed since is simply served by the other path, in place.
constexpr int kGridMaxSpan = 3;    // cells per axis a query's ball may span (27 cells, 9 rows of up to 3)
constexpr int kGridRowBatch = 4;   // rows whose bounds are fetched together

__device__ __forceinline__ void grid_axis_range(float c, float r, float lo, float inv, int dim, int &a, int &b) {
  // same expression as grid_cell_of (grid_build.hip): floor((x - lo) * inv), monotone in x
  // Clamped at BOTH ends like grid_cell_of: `dim` comes from double arithmetic on the host, the cells from this fp32
  // expression, so a point on the box's upper face may compute cell `dim` and is stored in cell dim - 1; a query beyond
  // that face must still reach that cell.  (A ball that lies wholly outside the box then scans the boundary cells: exact,
  // its points are simply tested.)
  a = min(max((int)floorf(__fmul_rn(__fsub_rn(__fsub_rn(c, r), lo), inv)), 0), dim - 1);
  b = min(max((int)floorf(__fmul_rn(__fsub_rn(__fadd_rn(c, r), lo), inv)), 0), dim - 1);
}

// best / gpos come in holding the previous match (distance, sorted position) and go out holding the nearest neighbour.
// Returns false, having changed nothing, if the ball spans more cells than the scan is built for.
__device__ __forceinline__ bool grid_scan(const GridView &g, float qx, float qy, float qz, float &best, uint32_t &gpos) {
  const float r = __fadd_rn(__fmul_rn(__fsqrt_rn(best), 1.0005f), g.eps);
  int ax, bx, ay, by, az, bz;
  grid_axis_range(qx, r, g.lo[0], g.inv, g.dim[0], ax, bx);
  grid_axis_range(qy, r, g.lo[1], g.inv, g.dim[1], ay, by);
  grid_axis_range(qz, r, g.lo[2], g.inv, g.dim[2], az, bz);
  const int ny = by - ay + 1, nz = bz - az + 1, nrow = ny * nz;
  if (ny > kGridMaxSpan || nz > kGridMaxSpan || bx - ax + 1 > kGridMaxSpan) return false;
  // rows in (y fastest, z) order, kGridRowBatch at a time: the bounds of a batch first (independent fetches), then its runs.
  // Most queries need one batch (<= 2 x 2 rows); a query with a larger ball (the tail of the sensor noise) takes up to three
  // instead of a whole tree walk.
  int iy = ay, iz = az;
  for (int done = 0; done < nrow; done += kGridRowBatch) {
    uint32_t rs[kGridRowBatch], re[kGridRowBatch];
#pragma unroll
    for (int k = 0; k < kGridRowBatch; ++k) {
      const uint32_t row = ((uint32_t)iz * (uint32_t)g.dim[1] + (uint32_t)iy) * (uint32_t)g.dim[0];
      rs[k] = g.cell_start[row + (uint32_t)ax];
      re[k] = g.cell_start[row + (uint32_t)bx + 1u];
      if (done + k + 1 < nrow) {   // rows past the last one repeat it (their runs are not scanned)
        const bool wrap = iy == by;
        iy = wrap ? ay : iy + 1;
        iz += wrap ? 1 : 0;
      }
    }
#pragma unroll
    for (int k = 0; k < kGridRowBatch; ++k) {
      if (done + k < nrow) {
        for (uint32_t p = rs[k]; p < re[k]; p += 4) {
          // four 16-byte loads from one base; a batch may run past the run (guarded) and, at the very end of the array,
          // into the kPtsPad zeroed entries
          const float4 *pb = g.gpts + p;
          const v4f p0 = ld16(pb), p1 = ld16(pb + 1), p2 = ld16(pb + 2), p3 = ld16(pb + 3);
          const float d0 = sq_dist3(__fsub_rn(qx, p0.x), __fsub_rn(qy, p0.y), __fsub_rn(qz, p0.z));
          const float d1 = sq_dist3(__fsub_rn(qx, p1.x), __fsub_rn(qy, p1.y), __fsub_rn(qz, p1.z));
          const float d2 = sq_dist3(__fsub_rn(qx, p2.x), __fsub_rn(qy, p2.y), __fsub_rn(qz, p2.z));
          const float d3 = sq_dist3(__fsub_rn(qx, p3.x), __fsub_rn(qy, p3.y), __fsub_rn(qz, p3.z));
          const uint32_t e = re[k];
          if (d0 < best) { best = d0; gpos = p; }
          if (p + 1 < e && d1 < best) { best = d1; gpos = p + 1; }
          if (p + 2 < e && d2 < best) { best = d2; gpos = p + 2; }
          if (p + 3 < e && d3 < best) { best = d3; gpos = p + 3; }
        }
      }
    }
  }
  return true;
}

template <bool NRM, bool CERT = false>
__global__ __launch_bounds__(kAccBlock, CERT ? 4 : kAccWavesPerSimd) void icp_accumulate_grid_kernel(
    CloudView src, BvhView tgt, GridView grid, const IcpState *__restrict__ st, double *__restrict__ partials,
    int32_t *__restrict__ corr_match, float *__restrict__ corr_d2, uint32_t *__restrict__ hint, uint32_t *__restrict__ ghint,
    const uint32_t *__restrict__ qorder, unsigned char *__restrict__ qclass, const uint32_t *__restrict__ chunk_order,
    uint32_t *__restrict__ chunk_cost, const uint32_t *__restrict__ plan_info, double *__restrict__ S_atomic, uint32_t measuring_launch,
    uint32_t *chain, uint32_t chain_seq, uint32_t *pace, uint32_t launch_no, uint32_t wait_ticks, float4 *__restrict__ cert_q,
    uint32_t *__restrict__ cert_pos, uint32_t *__restrict__ cert_stats, float *__restrict__ cert_l) {
  if (pace != nullptr && blockIdx.x == 0 && threadIdx.x == 0) __hip_atomic_store(pace, launch_no, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);   // see icp_accumulate_kernel
  __shared__ uint32_t s_ncert;
  __shared__ __attribute__((aligned(16))) float s_const[20];   // F rows [0..11], pivot [12..14], best0 [15]; [16..19]: see icp_accumulate_kernel
  if (!acc_launch_begin(st, chain, chain_seq, s_const, wait_ticks)) return;
  constexpr int BLOCK = kAccBlock;
  __shared__ double s_red[BLOCK / 64][kNumSumsMax];
  __shared__ float s_stk[kMaxDepth + 1][BLOCK];
  float *stk = &s_stk[0][threadIdx.x];
  const double max_d2 = st->max_d2;
  float best0 = INFINITY;
  if (max_d2 < 3.0e38) {
    float f = (float)max_d2;
    if ((double)f < max_d2) f = nextafterf(f, INFINITY);
    best0 = nextafterf(f, INFINITY);
  }
  if (threadIdx.x < 12) { if (chain == nullptr) s_const[threadIdx.x] = st->Ff[threadIdx.x]; }   // (overlapped: fetched by acc_launch_begin)
  else if (threadIdx.x < 15) s_const[threadIdx.x] = (float)st->pivot[threadIdx.x - 12];
  else if (threadIdx.x == 15) s_const[15] = best0;
  else if (threadIdx.x == 17) { if (chain == nullptr) s_const[17] = st->last_move; }
  else if (threadIdx.x == 18) s_const[18] = st->cert_k;
  else if (threadIdx.x == 19) s_const[19] = st->cert_cap;
  else if (threadIdx.x == 30) s_ncert = 0u;
  __syncthreads();
  if (CERT && blockIdx.x == 0 && threadIdx.x == 0) atomicAdd(cert_stats + 2, 1u);   // launches that kept certificates
  const bool rej_sn = NRM && st->use_surface_normal_rej;
  const bool rej_so = NRM && st->use_self_occluded_rej;
  const double thr_sn = st->surface_normal_thr, thr_so = st->self_occluded_thr;
  if ((threadIdx.x & 63u) < (uint32_t)kNumSumsMax) s_red[threadIdx.x >> 6][threadIdx.x & 63u] = 0.0;
  const bool p2p = NRM && st->estimator == OPE_EST_POINT_TO_PLANE_LLS;

  const uint32_t lane_id = threadIdx.x & 63u;
  const uint32_t n_waves = gridDim.x * (BLOCK / 64);
  const uint32_t wave_id = blockIdx.x * (BLOCK / 64) + (threadIdx.x >> 6);
  // the query order: n_grid_q grid-class queries, then the tree-class ones; before the first plan: all "tree-class",
  // identity order (every query still tries the grid first if it has a previous match)
  const uint32_t n_grid_q = min(plan_info[1], src.n_valid);
  const uint32_t n_tree_q = src.n_valid - n_grid_q;
  const uint32_t n_gc = (n_grid_q + 63u) / 64u, n_tc = (n_tree_q + 63u) / 64u;
  const bool measuring = chunk_order && measuring_launch != 0u;   // see icp_accumulate_kernel
  const uint32_t n_heavy = (chunk_order && !measuring) ? min(plan_info[0], n_tc) : 0u;
  // waves 0 .. n_alone-1: one of the costliest per-lane tree chunks each and nothing else; all other waves, in snake
  // order: 8 slots per heavy tree chunk (8-lane group walks), the other tree chunks, the grid chunks (see icp_accumulate_kernel)
  const uint32_t n_alone = chunk_order ? min(min(plan_info[5] + (measuring ? min(plan_info[0], n_tc) : 0u), n_tc - n_heavy), n_waves / 2u) : 0u;
  const uint32_t n_tree_slots = 8u * n_heavy + (n_tc - n_heavy - n_alone);
  const uint32_t n_snake = n_tree_slots + n_gc;
  const uint32_t snake_waves = n_waves - n_alone;
  for (uint32_t round = 0;; ++round) {
    bool oct = false, tree_part = true;
    uint32_t ord = 0, sub = 0, gchunk = 0;
    if (wave_id < n_alone) {
      if (round > 0) break;
      ord = n_heavy + wave_id;
    } else {
      if (round * snake_waves >= n_snake) break;
      const uint32_t w = wave_id - n_alone;
      const uint32_t slot = round * snake_waves + ((round & 1u) ? (snake_waves - 1u - w) : w);
      if (slot >= n_snake) continue;
      oct = slot < 8u * n_heavy;
      tree_part = slot < n_tree_slots;
      ord = oct ? (slot >> 3) : (n_heavy + n_alone + (slot - 8u * n_heavy));
      sub = slot & 7u;
      gchunk = slot - n_tree_slots;
    }
    uint32_t chunk = 0, pos0, pos_end;
    if (tree_part) {
      chunk = chunk_order ? min(chunk_order[ord], n_tc - 1u) : ord;
      pos0 = n_grid_q + chunk * 64u;
      pos_end = src.n_valid;
    } else {
      pos0 = gchunk * 64u;
      pos_end = n_grid_q;
    }
    const unsigned long long t_begin = __builtin_amdgcn_s_memtime();
    const uint32_t qpos = oct ? (pos0 + sub * 8u + (lane_id >> 3)) : (pos0 + lane_id);
    const bool active = qpos < pos_end;
    const uint32_t i = active ? qorder[qpos] : qorder[pos0];
    const bool owner = active && (!oct || (lane_id & 7u) == 0u);
    lds_cfloat_ptr cst = (lds_cfloat_ptr)s_const;
    asm volatile("" : "+v"(cst));
    float F[12];
#pragma unroll
    for (int k = 0; k < 12; ++k) F[k] = cst[k];
    const float4 s = src.xyzw[i];
    const float x = xform_row(F + 0, s.x, s.y, s.z);
    const float y = xform_row(F + 4, s.x, s.y, s.z);
    const float z = xform_row(F + 8, s.x, s.y, s.z);
    float nx = 0.f, ny = 0.f, nz = 0.f;
    if (NRM && src.nrm != nullptr) {
      const float4 n4 = src.nrm[i];
      nx = rot_row(F + 0, n4.x, n4.y, n4.z);
      ny = rot_row(F + 4, n4.x, n4.y, n4.z);
      nz = rot_row(F + 8, n4.x, n4.y, n4.z);
    }
    // ---- skip certificates (CERT instantiation; see icp_accumulate_kernel): a query answered by its certificate neither scans
    // nor walks; one that has to search and whose certificate would be worth it builds one (a 6-nearest TREE walk: candidates are
    // positions in the tree's point order, which this kernel reads its matches from as well in that case); everybody else — `live`
    // — takes the grid scan / tree walk below as in the plain instantiation.
    bool live = active, certified = false, build = false;
    float c_best = INFINITY;
    uint32_t c_pos = kNoPos, c_leaf = 0u;
    if constexpr (CERT) {
      const float best_init = active ? cst[15] : -INFINITY;
      bool stuck = false, has_cert = false;
      if (active) certified = cert_check(tgt, cert_q, cert_l, cert_pos, (size_t)src.n, i, x, y, z, best_init, cst[17], c_best, c_pos, stuck, has_cert);
      build = active && !certified && !stuck && owner && cert_worth_building(corr_d2[i], cst[18], cst[19], cst[17]);
      const bool group_builds = oct && __shfl((int)build, (int)(lane_id & ~7u), 64) != 0;   // (the eight lanes of a group carry one query)
      live = active && !certified && !build && !group_builds;
      if (__ballot(build) != 0ull) {
        const uint32_t hb = build ? hint[i] : 0u;
        cert_build(tgt, cert_q, cert_l, cert_pos, (size_t)src.n, i, x, y, z, best_init, hb, stk, BLOCK, build, build, c_best, c_pos, c_leaf);
        if (build && c_leaf != 0u && c_leaf != hb) hint[i] = c_leaf;
      }
      const uint32_t nc = (uint32_t)__popcll(__ballot(owner && certified));
      if (lane_id == 0 && nc != 0u) atomicAdd(&s_ncert, nc);
    }
    // ---- the previous match, if any: it bounds the search
    const uint32_t gh = live ? ghint[i] : 0u;
    float best = live ? cst[15] : -INFINITY;   // +inf, or just above the largest admissible d2
    uint32_t gpos = kNoPos;
    bool by_grid = false;
    if (gh != 0u) {
      const v4f t0 = ld16(grid.gpts + (gh - 1u));
      const float d0 = sq_dist3(__fsub_rn(x, t0.x), __fsub_rn(y, t0.y), __fsub_rn(z, t0.z));
      if (d0 < best) { best = d0; gpos = gh - 1u; }
      if (!oct && gpos != kNoPos) by_grid = grid_scan(grid, x, y, z, best, gpos);
    }
    // ---- everything else: the tree, seeded with what is known
    const bool need_tree = live && !by_grid;
    if (oct) {
      if (live) {
        NearestVisitor v{gpos != kNoPos ? nextafterf(best, INFINITY) : best, kNoPos, 0};
        bvh_traverse_oct(tgt, x, y, z, v, &s_stk[0][threadIdx.x & ~7u], BLOCK, hint[i]);
        if (v.pos != kNoPos) { best = v.best; gpos = grid.gpos_of_bvhpos[v.pos]; }
        if (owner && v.leaf != 0u) hint[i] = v.leaf;
      }
    } else if (__ballot(need_tree) != 0ull) {
      // lanes the grid has answered sit the walk out (best = -inf: nothing can improve them); a chunk of the tree part
      // whose lanes start from a handful of leaves (clutter far from the model does) takes ONE packet walk through the
      // scalar cache, everything else the per-lane walk (icp_accumulate_kernel)
      const uint32_t h = need_tree ? hint[i] : 0u;
      NearestVisitor v{need_tree ? (gpos != kNoPos ? nextafterf(best, INFINITY) : best) : -INFINITY, kNoPos, 0};
      const bool done = tree_part && bvh_traverse_packet(tgt, x, y, z, need_tree, v, h, stk, BLOCK);
      if (!done && need_tree) bvh_traverse(tgt, x, y, z, v, stk, BLOCK, h);
      if (need_tree) {
        if (v.pos != kNoPos) { best = v.best; gpos = grid.gpos_of_bvhpos[v.pos]; }
        if (v.leaf != 0u && v.leaf != h) hint[i] = v.leaf;
      }
    }
    // a query served by its certificate (or by the walk that built one) has its match as a position in the TREE's point order
    const bool by_cert = CERT && (certified || build);
    if (by_cert) best = c_best;
    const bool found = active && (by_cert ? c_pos != kNoPos : gpos != kNoPos);
    bool ok = found && !((double)best > max_d2);
    const float d2 = found ? best : INFINITY;
    const v4f tm = by_cert ? ld16(tgt.pts + (found ? c_pos : 0u)) : ld16(grid.gpts + (found ? gpos : 0u));
    const int match = found ? __float_as_int(tm.w) : -1;
    if (owner && !by_cert) {
      if (found && gpos + 1u != gh) ghint[i] = gpos + 1u;
      const unsigned char cls = by_grid ? 1 : 0;
      if (cls != (tree_part ? 0 : 1) || !chunk_order) qclass[i] = cls;   // the partition's expectation is written at the plan step
    }
    if (CERT && owner && build && found) ghint[i] = grid.gpos_of_bvhpos[c_pos] + 1u;   // (the scans of later launches start from it)
    if (NRM && ok && rej_sn) {
      const float4 tn = by_cert ? tgt.nrm[c_pos] : grid.gnrm[gpos];
      const float score = __fadd_rn(__fadd_rn(__fmul_rn(nx, tn.x), __fmul_rn(ny, tn.y)), __fmul_rn(nz, tn.z));
      ok = (double)score > thr_sn;
    }
    if (NRM && ok && rej_so) {
      const double sl = sqrt((double)__fadd_rn(__fadd_rn(__fmul_rn(x, x), __fmul_rn(y, y)), __fmul_rn(z, z)));
      const double score = (double)nx * (-(double)x / sl) + (double)ny * (-(double)y / sl) + (double)nz * (-(double)z / sl);
      ok = score > thr_so;
    }
    ok = ok && owner;
    if (owner) {
      __builtin_nontemporal_store(ok ? match : -1, corr_match + i);
      __builtin_nontemporal_store(d2, corr_d2 + i);
    }
    add_query_sums<NRM>(s_red[threadIdx.x >> 6], (lds_cfloat_ptr)s_const, lane_id, ok, p2p, x, y, z, make_float4(tm.x, tm.y, tm.z, tm.w),
                        (NRM && p2p) ? (by_cert ? tgt.nrm[ok ? c_pos : 0u] : grid.gnrm[ok ? gpos : 0u]) : make_float4(0.f, 0.f, 0.f, 0.f), d2);
    // tree chunks record their cost under their id, grid chunks behind them (the plan only needs their sum)
    if (lane_id == 0 && !oct) chunk_cost[tree_part ? chunk : n_tc + gchunk] = (uint32_t)((__builtin_amdgcn_s_memtime() - t_begin) >> 4);
  }
  __syncthreads();
  if (threadIdx.x < (p2p ? kNumSumsMax : kNumSums)) {
    double v = 0.0;
#pragma unroll
    for (int w = 0; w < BLOCK / 64; ++w) v += s_red[w][threadIdx.x];
    if (S_atomic != nullptr) unsafeAtomicAdd(S_atomic + threadIdx.x, v);
    else partials[threadIdx.x * kAccMaxBlocks + blockIdx.x] = v;
  }
  if (CERT && threadIdx.x == 64 && s_ncert != 0u) atomicAdd(reinterpret_cast<unsigned long long *>(cert_stats), (unsigned long long)s_ncert);
  acc_launch_end(chain);
}

// ------------------------------------------------------------------------------------------
// fp64 3x3 helpers for the update step (one lane)
// The update is a serial tail of every iteration, so its fp64 divisions / square roots use the
// hardware seed (v_rcp_f64 / v_rsq_f64) plus Newton steps instead of the ~40-instruction IEEE
// sequences: ~1e-16 relative error, 22 us -> a few us per iteration.
__device__ __forceinline__ double fast_rcp(double x) {
  double y = __builtin_amdgcn_rcp(x);
  y = y * (2.0 - x * y);
  y = y * (2.0 - x * y);
  return y;
}
__device__ __forceinline__ double fast_rsqrt(double x) {
  double y = __builtin_amdgcn_rsq(x);
  y = y * (1.5 - 0.5 * x * y * y);
  y = y * (1.5 - 0.5 * x * y * y);
  return y;
}
__device__ __forceinline__ double fast_sqrt(double x) { return x > 0.0 ? x * fast_rsqrt(x) : 0.0; }
// All indices below are compile-time constants after unrolling, so the 3x3 work stays in registers
// (a first version with run-time indices put 368 bytes per lane in scratch and took ~18 us).
template <int P, int Q>
__device__ __forceinline__ void jacobi_rotate(double (&S)[9], double (&V)[9]) {
  const double apq = S[3 * P + Q];
  if (apq == 0.0) return;
  const double theta = (S[3 * Q + Q] - S[3 * P + P]) * fast_rcp(2.0 * apq);
  const double t = (theta >= 0 ? 1.0 : -1.0) * fast_rcp(fabs(theta) + fast_sqrt(theta * theta + 1.0));
  const double c = fast_rsqrt(t * t + 1.0), s = t * c;
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    const double a = S[3 * k + P], b = S[3 * k + Q];
    S[3 * k + P] = c * a - s * b;
    S[3 * k + Q] = s * a + c * b;
  }
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    const double a = S[3 * P + k], b = S[3 * Q + k];
    S[3 * P + k] = c * a - s * b;
    S[3 * Q + k] = s * a + c * b;
  }
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    const double a = V[3 * k + P], b = V[3 * k + Q];
    V[3 * k + P] = c * a - s * b;
    V[3 * k + Q] = s * a + c * b;
  }
}

// `warm`: V holds an orthonormal basis that nearly diagonalises S already (the eigenvectors of the previous ICP
// iteration's matrix): S is moved into that basis first and one or two sweeps finish the job instead of five or six
// (the update lane is a serial tail of every iteration; its fp64 instructions issue one per 8 cycles).
__device__ __forceinline__ void jacobi_eig3(double (&S)[9], double (&V)[9], bool warm = false) {
  if (warm) {
    double SV[9], W[9];
#pragma unroll
    for (int r = 0; r < 3; ++r)
#pragma unroll
      for (int c = 0; c < 3; ++c) SV[3 * r + c] = S[3 * r] * V[c] + S[3 * r + 1] * V[3 + c] + S[3 * r + 2] * V[6 + c];
#pragma unroll
    for (int r = 0; r < 3; ++r)
#pragma unroll
      for (int c = r; c < 3; ++c) W[3 * r + c] = V[r] * SV[c] + V[3 + r] * SV[3 + c] + V[6 + r] * SV[6 + c];
    W[3] = W[1]; W[6] = W[2]; W[7] = W[5];
#pragma unroll
    for (int i = 0; i < 9; ++i) S[i] = W[i];
  } else {
#pragma unroll
    for (int i = 0; i < 9; ++i) V[i] = (i % 4 == 0) ? 1.0 : 0.0;
  }
  for (int sweep = 0; sweep < 30; ++sweep) {
    const double off = fabs(S[1]) + fabs(S[2]) + fabs(S[5]);
    const double diag = fabs(S[0]) + fabs(S[4]) + fabs(S[8]);
    // fp64 rounding keeps `off` near 1e-17*diag forever: stop at 1e-15 (rotation error ~1e-15)
    if (off <= 1e-300 || off <= 1e-15 * diag) break;
    jacobi_rotate<0, 1>(S, V);
    jacobi_rotate<0, 2>(S, V);
    jacobi_rotate<1, 2>(S, V);
  }
}

__device__ __forceinline__ double det3(const double (&M)[9]) {
  return M[0] * (M[4] * M[8] - M[5] * M[7]) - M[1] * (M[3] * M[8] - M[5] * M[6]) + M[2] * (M[3] * M[7] - M[4] * M[6]);
}

template <int A, int B>
__device__ __forceinline__ void sort_cols_desc(double (&ev)[3], double (&V)[9]) {
  if (ev[B] > ev[A]) {
    const double t = ev[A]; ev[A] = ev[B]; ev[B] = t;
#pragma unroll
    for (int r = 0; r < 3; ++r) { const double u = V[3 * r + A]; V[3 * r + A] = V[3 * r + B]; V[3 * r + B] = u; }
  }
}

// A = U diag(s) V^T, s descending (row-major 3x3)
// warm: V comes in holding the previous call's V (see jacobi_eig3)
__device__ __forceinline__ void svd3(const double (&A)[9], double (&U)[9], double (&s)[3], double (&V)[9], bool warm = false) {
  double AtA[9];
#pragma unroll
  for (int i = 0; i < 3; ++i)
#pragma unroll
    for (int j = 0; j < 3; ++j) AtA[3 * i + j] = A[i] * A[j] + A[3 + i] * A[3 + j] + A[6 + i] * A[6 + j];
  jacobi_eig3(AtA, V, warm);
  double ev[3] = {AtA[0], AtA[4], AtA[8]};
  sort_cols_desc<0, 1>(ev, V);
  sort_cols_desc<0, 2>(ev, V);
  sort_cols_desc<1, 2>(ev, V);
  // U columns = A v_c, then modified Gram-Schmidt with completion for (near-)null directions
  double u0[3], u1[3], u2[3];
#pragma unroll
  for (int r = 0; r < 3; ++r) {
    u0[r] = A[3 * r] * V[0] + A[3 * r + 1] * V[3] + A[3 * r + 2] * V[6];
    u1[r] = A[3 * r] * V[1] + A[3 * r + 1] * V[4] + A[3 * r + 2] * V[7];
    u2[r] = A[3 * r] * V[2] + A[3 * r + 1] * V[5] + A[3 * r + 2] * V[8];
  }
  s[0] = fast_sqrt(u0[0] * u0[0] + u0[1] * u0[1] + u0[2] * u0[2]);
  s[1] = fast_sqrt(u1[0] * u1[0] + u1[1] * u1[1] + u1[2] * u1[2]);
  s[2] = fast_sqrt(u2[0] * u2[0] + u2[1] * u2[1] + u2[2] * u2[2]);
  const double tiny = 1e-14 * (s[0] > 0 ? s[0] : 1.0);
  // column 0
  if (s[0] <= tiny) { u0[0] = 1; u0[1] = 0; u0[2] = 0; }
  else { const double i0 = fast_rcp(s[0]); u0[0] *= i0; u0[1] *= i0; u0[2] *= i0; }
  // column 1
  {
    const double d = u1[0] * u0[0] + u1[1] * u0[1] + u1[2] * u0[2];
    u1[0] -= d * u0[0]; u1[1] -= d * u0[1]; u1[2] -= d * u0[2];
    double n1 = fast_sqrt(u1[0] * u1[0] + u1[1] * u1[1] + u1[2] * u1[2]);
    if (s[1] <= tiny || n1 <= 1e-8 * s[1] + 1e-300) {
      // unit vector along the axis u0 is least aligned with, made orthogonal to u0
      const double ax = fabs(u0[0]), ay = fabs(u0[1]), az = fabs(u0[2]);
      const bool mx = ax < ay ? (ax < az) : false;
      const bool my = !mx && (ax < ay ? false : (ay < az));
      const double ex = mx ? 1.0 : 0.0, ey = my ? 1.0 : 0.0, ez = (!mx && !my) ? 1.0 : 0.0;
      const double dd = ex * u0[0] + ey * u0[1] + ez * u0[2];
      u1[0] = ex - dd * u0[0]; u1[1] = ey - dd * u0[1]; u1[2] = ez - dd * u0[2];
      n1 = fast_sqrt(u1[0] * u1[0] + u1[1] * u1[1] + u1[2] * u1[2]);
    }
    const double i1 = fast_rcp(n1);
    u1[0] *= i1; u1[1] *= i1; u1[2] *= i1;
  }
  // column 2
  {
    const double d0 = u2[0] * u0[0] + u2[1] * u0[1] + u2[2] * u0[2];
    u2[0] -= d0 * u0[0]; u2[1] -= d0 * u0[1]; u2[2] -= d0 * u0[2];
    const double d1 = u2[0] * u1[0] + u2[1] * u1[1] + u2[2] * u1[2];
    u2[0] -= d1 * u1[0]; u2[1] -= d1 * u1[1]; u2[2] -= d1 * u1[2];
    double n2 = fast_sqrt(u2[0] * u2[0] + u2[1] * u2[1] + u2[2] * u2[2]);
    if (s[2] <= tiny || n2 <= 1e-8 * s[2] + 1e-300) {
      u2[0] = u0[1] * u1[2] - u0[2] * u1[1];
      u2[1] = u0[2] * u1[0] - u0[0] * u1[2];
      u2[2] = u0[0] * u1[1] - u0[1] * u1[0];
      n2 = fast_sqrt(u2[0] * u2[0] + u2[1] * u2[1] + u2[2] * u2[2]);
    }
    const double i2 = fast_rcp(n2);
    u2[0] *= i2; u2[1] *= i2; u2[2] *= i2;
  }
#pragma unroll
  for (int r = 0; r < 3; ++r) { U[3 * r] = u0[r]; U[3 * r + 1] = u1[r]; U[3 * r + 2] = u2[r]; }
}

// Eigen::umeyama(src, dst, false) from the 17 sums (taken about `pivot`), column-major fp64 out.
// Vwarm (optional): 9 doubles + a validity flag carried from one ICP iteration to the next
__device__ __forceinline__ void umeyama_from_sums(const double *S, const double *pivot, double (&T)[16], double *Vwarm = nullptr,
                                                  int *have_warm = nullptr) {
  const double n = S[0];
  double sm[3], dm[3], sigma[9];
  const double inv_n = 1.0 / n;
#pragma unroll
  for (int d = 0; d < 3; ++d) { sm[d] = S[1 + d] * inv_n; dm[d] = S[4 + d] * inv_n; }
#pragma unroll
  for (int r = 0; r < 3; ++r)
#pragma unroll
    for (int c = 0; c < 3; ++c) sigma[3 * r + c] = S[7 + 3 * r + c] * inv_n - dm[r] * sm[c];
#pragma unroll
  for (int d = 0; d < 3; ++d) { sm[d] += pivot[d]; dm[d] += pivot[d]; }
  double U[9], sv[3], V[9];
  const bool warm = Vwarm != nullptr && *have_warm != 0;
  if (warm) {
#pragma unroll
    for (int i = 0; i < 9; ++i) V[i] = Vwarm[i];
  }
  svd3(sigma, U, sv, V, warm);
  if (Vwarm != nullptr) {
#pragma unroll
    for (int i = 0; i < 9; ++i) Vwarm[i] = V[i];
    *have_warm = 1;
  }
  double Sg[3] = {1, 1, 1};
  if (det3(sigma) < 0) Sg[2] = -1;
  int rank = 0;
#pragma unroll
  for (int i = 0; i < 3; ++i)
    if (!(fabs(sv[i]) <= fabs(sv[0]) * 1e-5)) ++rank;
  if (rank == 2) {
    Sg[0] = Sg[1] = 1;
    Sg[2] = (det3(U) * det3(V) > 0) ? 1 : -1;
  }
  double R[9];
#pragma unroll
  for (int i = 0; i < 3; ++i)
#pragma unroll
    for (int j = 0; j < 3; ++j) {
      double a = 0;
#pragma unroll
      for (int k = 0; k < 3; ++k) a += U[3 * i + k] * Sg[k] * V[3 * j + k];
      R[3 * i + j] = a;
    }
#pragma unroll
  for (int c = 0; c < 3; ++c)
#pragma unroll
    for (int r = 0; r < 3; ++r) T[4 * c + r] = R[3 * r + c];
  T[3] = T[7] = T[11] = 0.0;
#pragma unroll
  for (int i = 0; i < 3; ++i) T[12 + i] = dm[i] - (R[3 * i] * sm[0] + R[3 * i + 1] * sm[1] + R[3 * i + 2] * sm[2]);
  T[15] = 1.0;
}

// x = (AᵀA)^-1 Aᵀb by Cholesky (static indices: registers only), then PCL's constructTransformationMatrix.
// N holds the upper triangle of AᵀA row by row (21 values) followed by Aᵀb (6).
__device__ __forceinline__ bool point_to_plane_from_sums(const double *N, double (&T)[16]) {
  double A[6][6], b[6];
  {
    int k = 0;
#pragma unroll
    for (int r = 0; r < 6; ++r)
#pragma unroll
      for (int c = r; c < 6; ++c) { A[r][c] = N[k]; A[c][r] = N[k]; ++k; }
#pragma unroll
    for (int r = 0; r < 6; ++r) b[r] = N[21 + r];
  }
  double L[6][6];
  bool ok = true;
#pragma unroll
  for (int j = 0; j < 6; ++j) {
    double d = A[j][j];
#pragma unroll
    for (int k = 0; k < j; ++k) d -= L[j][k] * L[j][k];
    ok = ok && (d > 0.0);
    const double inv = ok ? fast_rsqrt(d) : 0.0;
    L[j][j] = d * inv;
#pragma unroll
    for (int i = j + 1; i < 6; ++i) {
      double v = A[i][j];
#pragma unroll
      for (int k = 0; k < j; ++k) v -= L[i][k] * L[j][k];
      L[i][j] = v * inv;
    }
  }
  if (!ok) return false;
  double yv[6], x[6];
#pragma unroll
  for (int i = 0; i < 6; ++i) {
    double v = b[i];
#pragma unroll
    for (int k = 0; k < i; ++k) v -= L[i][k] * yv[k];
    yv[i] = v * fast_rcp(L[i][i]);
  }
#pragma unroll
  for (int i = 5; i >= 0; --i) {
    double v = yv[i];
#pragma unroll
    for (int k = i + 1; k < 6; ++k) v -= L[k][i] * x[k];
    x[i] = v * fast_rcp(L[i][i]);
  }
  const double sa = sin(x[0]), ca = cos(x[0]), sb = sin(x[1]), cb = cos(x[1]), sg = sin(x[2]), cg = cos(x[2]);
  // column-major
  T[0] = cg * cb;                 T[4] = -sg * ca + cg * sb * sa;  T[8] = sg * sa + cg * sb * ca;   T[12] = x[3];
  T[1] = sg * cb;                 T[5] = cg * ca + sg * sb * sa;   T[9] = -cg * sa + sg * sb * ca;  T[13] = x[4];
  T[2] = -sb;                     T[6] = cb * sa;                  T[10] = cb * ca;                 T[14] = x[5];
  T[3] = 0.0; T[7] = 0.0; T[11] = 0.0; T[15] = 1.0;
  return true;
}

// Tk_ext: the incremental transform of an estimator that runs outside this kernel (LM, lm.hip), column-major float, or null
__device__ __forceinline__ void icp_update_lane(IcpState *st, const double *S, const float *Tk_ext = nullptr) {
  const double n = S[0];
  st->n_corr = (long long)n;
  // icp_mod.hpp:232-240
  if ((long long)n < (long long)st->min_correspondences) {
    st->state = OPE_CONV_NO_CORRESPONDENCES;
    st->converged = 0;
    st->done = 1;
    return;
  }
  double Tk[16];
  if (Tk_ext != nullptr) {
#pragma unroll
    for (int i = 0; i < 16; ++i) Tk[i] = (double)Tk_ext[i];
  } else if (st->estimator == OPE_EST_POINT_TO_PLANE_LLS) {
    if (!point_to_plane_from_sums(S + kNumSums, Tk)) {
      // singular normal equations: no usable step (PCL would propagate NaNs); stop with what we have
      st->state = OPE_CONV_NO_CORRESPONDENCES;
      st->converged = 0;
      st->done = 1;
      return;
    }
  } else {
    umeyama_from_sums(S, st->pivot, Tk, st->Vwarm, &st->have_Vwarm);
  }
  // transformation_ is a Matrix4f in the reference
  float Tf[16];
#pragma unroll
  for (int i = 0; i < 16; ++i) { Tf[i] = (float)Tk[i]; st->Tk[i] = (double)Tf[i]; }
  // k-NN runs: the transform the accumulate launch behind these sums searched with (see icp_accumulate_kernel, MODE 2)
#pragma unroll
  for (int i = 0; i < 12; ++i) st->Fprev[i] = st->Ff[i];
  st->have_prev = st->knn_acc_flag;
  st->knn_acc_flag = 0;
  // final_transformation_ = transformation_ * final_transformation_ (icp_mod.hpp:249), kept in fp64
  double Fn[16];
#pragma unroll
  for (int c = 0; c < 4; ++c)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      double a = 0;
#pragma unroll
      for (int k = 0; k < 4; ++k) a += (double)Tf[4 * k + r] * st->F[4 * c + k];
      Fn[4 * c + r] = a;
    }
  {
    // Skip certificates (ope.h: skip_certificates): from the update on that moves no scene point by more than cert_thr the
    // accumulate launches keep per-query certificates.  The largest displacement over the scene's bounding sphere, centre c and
    // radius r in the scene's own frame: |Fn c - F c| + ||Rn - R||_F r (a trigger only: exactness never rests on it).  Sticky.
    const double cx = (double)st->src_c[0], cy = (double)st->src_c[1], cz = (double)st->src_c[2];
    double mv2 = 0.0, dr2 = 0.0;
#pragma unroll
    for (int r = 0; r < 3; ++r) {
      const double d0 = Fn[r] - st->F[r], d1 = Fn[4 + r] - st->F[4 + r], d2 = Fn[8 + r] - st->F[8 + r], d3 = Fn[12 + r] - st->F[12 + r];
      const double m = d0 * cx + d1 * cy + d2 * cz + d3;
      mv2 += m * m;
      dr2 += d0 * d0 + d1 * d1 + d2 * d2;
    }
    const float move = (float)(fast_sqrt(mv2) + fast_sqrt(dr2) * (double)st->src_r);
    st->last_move = move;
    if (move < st->cert_thr && st->cert_mode == 0) {
      st->cert_mode = 1;
      if (st->host_cert != nullptr) __hip_atomic_store(st->host_cert, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
  }
#pragma unroll
  for (int i = 0; i < 16; ++i) st->F[i] = Fn[i];
#pragma unroll
  for (int r = 0; r < 3; ++r)
#pragma unroll
    for (int c = 0; c < 4; ++c) st->Ff[4 * r + c] = (float)Fn[4 * c + r];
  if (st->use_reciprocal) {
    // inverse of the affine part by the adjugate (F is rigid unless the caller's guess was not)
    const double a = Fn[0], b = Fn[4], c = Fn[8], d = Fn[1], e = Fn[5], f = Fn[9], g = Fn[2], h = Fn[6], i = Fn[10];
    const double det = a * (e * i - f * h) - b * (d * i - f * g) + c * (d * h - e * g);
    const double id = 1.0 / det;
    const double M[9] = {(e * i - f * h) * id, (c * h - b * i) * id, (b * f - c * e) * id,
                         (f * g - d * i) * id, (a * i - c * g) * id, (c * d - a * f) * id,
                         (d * h - e * g) * id, (b * g - a * h) * id, (a * e - b * d) * id};
#pragma unroll
    for (int r = 0; r < 3; ++r) {
      st->Finv[4 * r + 0] = (float)M[3 * r]; st->Finv[4 * r + 1] = (float)M[3 * r + 1]; st->Finv[4 * r + 2] = (float)M[3 * r + 2];
      st->Finv[4 * r + 3] = (float)(-(M[3 * r] * Fn[12] + M[3 * r + 1] * Fn[13] + M[3 * r + 2] * Fn[14]));
    }
  }
  const int iterations = ++st->iterations;

  // DefaultConvergenceCriteria::hasConverged (uPCL) with the thresholds wired at icp_mod.hpp:164-168
  st->state = OPE_CONV_NOT_CONVERGED;
  int conv = 0;
  if (iterations >= st->max_iterations) {
    if (!st->failure_after_max_iter) { st->state = OPE_CONV_ITERATIONS; conv = 1; }
    st->converged = conv;
    st->done = 1;
    return;
  }
  const double cos_angle = 0.5 * ((double)Tf[0] + (double)Tf[5] + (double)Tf[10] - 1.0);
  const double tr2 = (double)Tf[12] * Tf[12] + (double)Tf[13] * Tf[13] + (double)Tf[14] * Tf[14];
  if (cos_angle >= st->rotation_threshold && tr2 <= st->translation_threshold) {
    st->state = OPE_CONV_TRANSFORM;
    conv = 1;
  } else {
    st->cur_mse = S[16] / n;
    const double diff = fabs(st->cur_mse - st->prev_mse);
    if (diff < st->mse_threshold_absolute) { st->state = OPE_CONV_ABS_MSE; conv = 1; }
    else if (diff / st->prev_mse < st->mse_threshold_relative) { st->state = OPE_CONV_REL_MSE; conv = 1; }
    else st->prev_mse = st->cur_mse;
  }
  st->converged = conv;
  st->done = conv;
}

// The update lane works on an LDS copy of the state: its ~60 dependent accesses then cost LDS
// latency instead of one L2 round trip each (the global version took 18 us of a 22 us launch).
__device__ __forceinline__ void state_to_lds(IcpState *dst, const IcpState *src) {
  constexpr int kWords = (int)(sizeof(IcpState) / 4);
  static_assert(sizeof(IcpState) % 4 == 0, "IcpState must be a whole number of dwords");
  for (int i = threadIdx.x; i < kWords; i += blockDim.x)
    reinterpret_cast<uint32_t *>(dst)[i] = reinterpret_cast<const uint32_t *>(src)[i];
}

// Fixed-order reduction of the block partials.  256 threads (4 partial rows each) so that the update
// lane may use the whole 512-register file instead of spilling its fp64 3x3 algebra.
constexpr int kRedBlock = 256;
__global__ __launch_bounds__(kRedBlock) void icp_reduce_update_kernel(IcpState *st, const double *__restrict__ partials,
                                                                            double *S, int nblocks, int do_update,
                                                                            uint32_t *work_counter) {
  if (st->done) return;
  // LDS tree in a fixed order (cross-lane fp64 shuffles serialised into ~200 dependent
  // ds_bpermutes and took 15 us): rows -> 256 per-thread sums -> 8 group sums -> total.
  __shared__ double s_part[kNumSums][kRedBlock];
  __shared__ double s_grp[kNumSums][8];
  __shared__ double s_S[kNumSumsMax];
  __shared__ IcpState s_st;
  if (do_update) state_to_lds(&s_st, st);
  const int nsums = (st->estimator == OPE_EST_POINT_TO_PLANE_LLS) ? kNumSumsMax : kNumSums;
  for (int base = 0; base < nsums; base += kNumSums) {   // 17 components per pass through the LDS tree
#pragma unroll
    for (int k = 0; k < kNumSums; ++k) {
      double v = 0.0;
      if (base + k < nsums) {
#pragma unroll
        for (int j = 0; j < kAccMaxBlocks / kRedBlock; ++j)
          // unconditional, independent loads: rows >= nblocks were zero-filled by ope_icp_begin
          v += partials[(base + k) * kAccMaxBlocks + (int)threadIdx.x + j * kRedBlock];
      }
      s_part[k][threadIdx.x] = v;
    }
    __syncthreads();
    {
      const int k = threadIdx.x >> 3, g = threadIdx.x & 7;
      if (k < kNumSums) {
        double v = 0.0;
#pragma unroll
        for (int j = 0; j < kRedBlock / 8; ++j) v += s_part[k][g * (kRedBlock / 8) + j];
        s_grp[k][g] = v;
      }
    }
    __syncthreads();
    if (threadIdx.x < kNumSums && base + (int)threadIdx.x < nsums) {
      double v = 0.0;
#pragma unroll
      for (int g = 0; g < 8; ++g) v += s_grp[threadIdx.x][g];
      S[base + threadIdx.x] = v;
      s_S[base + threadIdx.x] = v;
    }
    __syncthreads();
  }
  if (do_update) {
    __syncthreads();
    if (threadIdx.x == 0) icp_update_lane(&s_st, s_S);
    __syncthreads();
    state_to_lds(st, &s_st);
  }
}

// The sums are consumed: they are left at zero, ready for the next accumulate launch to add into (sharded runs).
// nsums: 17, or 44 with the point-to-plane estimator: a caller-owned sums buffer (ope_icp_set_sums_buffer) only has to hold
// what the estimator uses, and nothing past that is read or written.
__global__ __launch_bounds__(64) void icp_update_kernel(IcpState *st, double *S, int nsums, const float *Tk_ext) {
  if (st->done) return;
  __shared__ double s_S[kNumSumsMax];
  __shared__ IcpState s_st;
  state_to_lds(&s_st, st);
  if ((int)threadIdx.x < kNumSumsMax) s_S[threadIdx.x] = (int)threadIdx.x < nsums ? S[threadIdx.x] : 0.0;
  __syncthreads();
  if (threadIdx.x == 0) icp_update_lane(&s_st, s_S, Tk_ext);
  __syncthreads();
  if ((int)threadIdx.x < kNumSumsMax) s_st.S[threadIdx.x] = 0.0;   // S may be the state's own array
  __syncthreads();
  state_to_lds(st, &s_st);
  if ((int)threadIdx.x < nsums) S[threadIdx.x] = 0.0;
}

// The update launch of an overlapped run (see acc_launch_begin): on its own stream, resident while accumulate launch `seq`
// still runs.  The state is fetched first (nothing writes it during an accumulate launch), then lane 0 waits for the
// launch's tickets (the word counts up through the run: `tickets` = blocks of every overlapped launch so far, this one
// included), the sums are read and consumed as in icp_update_kernel and the update is published (chain[1] = seq + 1) to the blocks of launch seq + 1, which are waiting or about to start.
// A run that is over ("done") had no tickets to wait for: the word is published all the same, so that the launches still
// enqueued behind it drain.  192 VGPRs (it needs 96): the wave must fit into the block slot that api.hip leaves free on every
// XCD beside an accumulate launch (two of its waves per SIMD).
__global__ __launch_bounds__(64) __attribute__((amdgpu_num_vgpr(192))) void icp_update_chained_kernel(IcpState *st, int nsums, uint32_t *chain,
                                                                                                uint32_t seq, uint32_t tickets, uint32_t wait_ticks) {
  __shared__ double s_S[kNumSumsMax];
  __shared__ IcpState s_st;
  __shared__ uint32_t s_mode;   // 0: the run is over, 1: update, 2: the accumulate launch did not report in time
  const int t = (int)threadIdx.x;
  state_to_lds(&s_st, st);      // (written by the update before this one, a finished kernel of this stream: plain loads)
  if (t == 0) {
    uint32_t mode = 1u;
    if (__hip_atomic_load(&st->done, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) mode = 0u;
    else {
      const unsigned long long t0 = wall_clock64();
      while ((int32_t)(chain_load(chain) - tickets) < 0) {
        if (wall_clock64() - t0 > (unsigned long long)wait_ticks) { mode = 2u; break; }
        __builtin_amdgcn_s_sleep(1);
      }
    }
    s_mode = mode;
  }
  __syncthreads();
  const uint32_t mode = s_mode;
  constexpr int kSumsWord0 = (int)(offsetof(IcpState, S) / 4), kSumsWords = 2 * kNumSumsMax, kWords = (int)(sizeof(IcpState) / 4);
  if (mode == 1u) {
    // the sums as the blocks' atomics left them (the state's own array: overlapped runs have no caller-owned sums buffer)
    if (t < kNumSumsMax) s_S[t] = t < nsums ? __hip_atomic_load(&st->S[t], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0.0;
    __syncthreads();
    if (t == 0) icp_update_lane(&s_st, s_S, nullptr);
    __syncthreads();
    // The state goes back with plain stores — the next update and the host read it after this kernel has ended — EXCEPT the
    // sums: a zero left dirty in this XCD's L2 would be written back at the end of this kernel, over what the next launch's
    // blocks have added by then.  The consumed sums are zeroed, and the words the next launch reads while this kernel is
    // still running (transform rows, "done") are stored once more, with agent-scope atomics.
    for (int i = t; i < kWords; i += 64)
      if (i < kSumsWord0 || i >= kSumsWord0 + kSumsWords) reinterpret_cast<uint32_t *>(st)[i] = reinterpret_cast<const uint32_t *>(&s_st)[i];
    if (t < nsums) __hip_atomic_store(&st->S[t], 0.0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (t < 12) __hip_atomic_store(&st->Ff[t], s_st.Ff[t], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (t == 12) __hip_atomic_store(&st->done, s_st.done, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (t == 17) __hip_atomic_store(&st->last_move, s_st.last_move, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  } else if (mode == 2u && t == 0) {
    __hip_atomic_store(&st->chain_error, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_store(&st->done, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    atomicOr(chain + 2, 2u);
  }
  chain_wait_own_memory_ops();
  __syncthreads();
  if (t == 0) chain_store(chain + 1, seq + 1u);
}

// OPE_EST_POINT_TO_PLANE_LM: the Levenberg-Marquardt minimisation on the 91 sums of lm_stats_kernel (lm.hip, lm_solve.hpp)
// and the update step in one launch, one lane: nothing comes back to the host between two iterations.  S: the 17 sums of the
// accumulate launch (n and the MSE feed the convergence test); both sum buffers are left at zero.
__global__ __launch_bounds__(64) void icp_lm_update_kernel(IcpState *st, double *S, double *stats) {
  if (st->done) return;
  __shared__ double s_S[kNumSumsMax];
  __shared__ LmQuad s_q;
  __shared__ IcpState s_st;
  __shared__ float s_Tk[16];
  state_to_lds(&s_st, st);
  if ((int)threadIdx.x < kNumSumsMax) s_S[threadIdx.x] = (int)threadIdx.x < kNumSums ? S[threadIdx.x] : 0.0;
  lm_load_stats(stats, s_q);
  __syncthreads();
  if (threadIdx.x == 0) {
    lm_minimize_lane(s_q, (long long)s_S[0], s_Tk);
    icp_update_lane(&s_st, s_S, s_Tk);
  }
  __syncthreads();
  if ((int)threadIdx.x < kNumSumsMax) s_st.S[threadIdx.x] = 0.0;   // S may be the state's own array
  __syncthreads();
  state_to_lds(st, &s_st);
  if ((int)threadIdx.x < kNumSums) S[threadIdx.x] = 0.0;
  for (int k = threadIdx.x; k < 91; k += blockDim.x) stats[k] = 0.0;
}

// ------------------------------------------------------------------------------------------
// Peer-to-peer exchange of the sums and the update step in ONE launch (sharded runs on one node).
// Thread w < 2 * nsums sends half w & 1 of sum w >> 1: an 8-byte word {32 data bits, sequence number} stored (relaxed,
// system scope) into this rank's slot of parity seq & 1 in every rank's buffer, this rank's own included.  The same
// thread then polls word w of every rank's slot in its OWN buffer until the number matches — a word is complete the
// moment it is visible (aligned 8-byte stores are single-copy atomic), so no fence orders data before flags, the one
// assumption the protocol makes about the fabric.  Sums are added in rank order: every rank computes bit-identical
// totals, hence the same transform and the same "done".  Two parities: a rank that is one exchange ahead writes the
// other parity, and it cannot be two ahead because it needs every peer's words of the exchange in between.
// A peer that does not deliver within `timeout_ticks` (100 MHz ticks) ends the run with comm_error set: the kernel
// always terminates.
__global__ __launch_bounds__(256) void icp_p2p_update_kernel(IcpState *st, double *S, int nsums, P2pView pv, uint32_t seq,
                                                            unsigned long long timeout_ticks, int do_update, double *out_sums) {
  if (st != nullptr && st->done) return;   // (st == nullptr: the self-test at communicator set-up)
  __shared__ double s_S[kP2pMaxSums];
  __shared__ IcpState s_st;
  __shared__ uint32_t s_half[kP2pMaxRanks][kP2pSlotWords];
  __shared__ int s_fail;
  const int t = (int)threadIdx.x;
  const int nw = 2 * nsums;
  const size_t parity_off = (size_t)(seq & 1u) * kP2pMaxRanks * kP2pSlotWords;
  if (do_update) state_to_lds(&s_st, st);
  if (t == 0) s_fail = 0;
  __syncthreads();
  if (t < nw) {
    const double v = S[t >> 1];
    const uint32_t bits = (uint32_t)((t & 1) ? __double2hiint(v) : __double2loint(v));
    const unsigned long long word = ((unsigned long long)seq << 32) | bits;
    for (int r = 0; r < pv.nranks; ++r)
      __hip_atomic_store(pv.buf[r] + parity_off + (size_t)pv.rank * kP2pSlotWords + t, word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    const unsigned long long t0 = wall_clock64();
    const unsigned long long *mine = pv.buf[pv.rank] + parity_off + t;
    for (int r = 0; r < pv.nranks; ++r) {
      unsigned long long w;
      for (;;) {
        w = __hip_atomic_load(mine + (size_t)r * kP2pSlotWords, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        if ((uint32_t)(w >> 32) == seq) break;
        if (wall_clock64() - t0 > timeout_ticks) { s_fail = 1; break; }
        __builtin_amdgcn_s_sleep(1);
      }
      s_half[r][t] = (uint32_t)w;
    }
  }
  __syncthreads();
  if (t < kP2pMaxSums) {
    double acc = 0.0;
    if (t < nsums)
      for (int r = 0; r < pv.nranks; ++r) acc += __hiloint2double((int)s_half[r][2 * t + 1], (int)s_half[r][2 * t]);
    s_S[t] = acc;
  }
  __syncthreads();
  const bool failed = s_fail != 0;
  if (!do_update) {   // the exchange alone (self-test at set-up; the LM estimator's sums): hand the totals (or NaNs) back
    if (t < nsums) out_sums[t] = failed ? __longlong_as_double(0x7ff8000000000000ll) : s_S[t];
    if (failed && st != nullptr && t == 0) { st->comm_error = 1; st->done = 1; }
    return;
  }
  if (failed) {
    if (t == 0) { st->comm_error = 1; st->done = 1; }
    return;
  }
  if (t == 0) icp_update_lane(&s_st, s_S, nullptr);
  __syncthreads();
  if (t < kNumSumsMax) s_st.S[t] = 0.0;
  __syncthreads();
  state_to_lds(st, &s_st);
  if (t < nsums) S[t] = 0.0;   // consumed: ready for the next accumulate launch to add into
}

// ------------------------------------------------------------------------------------------
// The reference's injected "fixed correspondences" (setFixedCorrespondences, vPCL icp_mod.h:268; unused by its programs).
// fix: four float4 per pair {source point, source normal, target point, target normal}, gathered once by
// ope_icp_set_fixed_correspondences.  One launch after every accumulate launch adds the pairs' terms to the run's sums with
// the multiplicity the reference gives them:
//   1-NN estimation lists every given pair, distance field = (squared distance, float) * 1e10
//   (correspondence_estimation_mod.hpp:134-162); normal shooting lists none and sets the field to the squared distance to
//   the source normal's line (…normal_shooting_weighted.hpp:81-101); listed pairs pass through the rejectors like any other;
//   then the FIRST rejector alone is applied to the given pairs once more and the survivors are appended (icp_mod.hpp:210-224).
__global__ __launch_bounds__(64) void gather_fixed_pairs_kernel(CloudView src, CloudView tgt, const uint32_t *__restrict__ pos, uint32_t n,
                                                                float4 *__restrict__ fix) {
  const uint32_t f = blockIdx.x * 64u + threadIdx.x;
  if (f >= n) return;
  const uint32_t ps = pos[2 * f], pt = pos[2 * f + 1];
  const float4 z = make_float4(0.f, 0.f, 0.f, 0.f);
  fix[4 * f + 0] = src.xyzw[ps];
  fix[4 * f + 1] = src.nrm ? src.nrm[ps] : z;
  fix[4 * f + 2] = tgt.xyzw[pt];
  fix[4 * f + 3] = tgt.nrm ? tgt.nrm[pt] : z;
}

// seen: per pair {the distance field the reference writes back through the caller's pointer every iteration
// (correspondence_estimation_mod.hpp:159), bit 0 = listed and through every rejector | bit 1 = appended by the first rejector}
// as THIS launch saw them — ope_icp_fixed_correspondences reads the last launch's.
__global__ __launch_bounds__(64) void icp_fixed_pairs_kernel(const IcpState *__restrict__ st, const float4 *__restrict__ fix, uint32_t n, double *S,
                                                             float2 *__restrict__ seen) {
  if (st->done) return;
  // (every lane adds its pairs — f = lane, lane + 64, ... — in that order, then the 64 lane sums are added in lane order: the
  // pairs' share of the sums is the same bits from run to run, as ope_icp_params.deterministic_sums promises for the whole)
  __shared__ double s_lane[kNumSums][64];
  double acc[kNumSums];
#pragma unroll
  for (int k = 0; k < kNumSums; ++k) acc[k] = 0.0;
  const bool ns_mode = st->corr_mode == OPE_CORR_NORMAL_SHOOTING;
  const bool rej_sn = st->use_surface_normal_rej != 0, rej_so = st->use_self_occluded_rej != 0;
  const double thr_sn = st->surface_normal_thr, thr_so = st->self_occluded_thr;
  const double px = st->pivot[0], py = st->pivot[1], pz = st->pivot[2];
  for (uint32_t f = threadIdx.x; f < n; f += 64u) {
    const float4 s = fix[4 * f], sn = fix[4 * f + 1], t = fix[4 * f + 2], tn = fix[4 * f + 3];
    const float x = xform_row(st->Ff + 0, s.x, s.y, s.z), y = xform_row(st->Ff + 4, s.x, s.y, s.z), z = xform_row(st->Ff + 8, s.x, s.y, s.z);
    const float nx = rot_row(st->Ff + 0, sn.x, sn.y, sn.z), ny = rot_row(st->Ff + 4, sn.x, sn.y, sn.z), nz = rot_row(st->Ff + 8, sn.x, sn.y, sn.z);
    const float vx = __fsub_rn(t.x, x), vy = __fsub_rn(t.y, y), vz = __fsub_rn(t.z, z);
    float dist;
    if (!ns_mode) {
      const float d2 = __fadd_rn(__fadd_rn(__fmul_rn(vx, vx), __fmul_rn(vy, vy)), __fmul_rn(vz, vz));
      dist = (float)((double)d2 * 1e10);
    } else {
      const double cx = (double)ny * vz - (double)nz * vy, cy = (double)nz * vx - (double)nx * vz, cz = (double)nx * vy - (double)ny * vx;
      dist = (float)(cx * cx + cy * cy + cz * cz);
    }
    bool pass_sn = true, pass_so = true;
    if (rej_sn) pass_sn = (double)__fadd_rn(__fadd_rn(__fmul_rn(nx, tn.x), __fmul_rn(ny, tn.y)), __fmul_rn(nz, tn.z)) > thr_sn;
    if (rej_so) {
      const double sl = sqrt((double)__fadd_rn(__fadd_rn(__fmul_rn(x, x), __fmul_rn(y, y)), __fmul_rn(z, z)));
      pass_so = (double)nx * (-(double)x / sl) + (double)ny * (-(double)y / sl) + (double)nz * (-(double)z / sl) > thr_so;
    }
    const int listed = (!ns_mode && pass_sn && pass_so) ? 1 : 0;                            // listed by the estimation, through every rejector
    const int appended = ((rej_sn || rej_so) && (rej_sn ? pass_sn : pass_so)) ? 1 : 0;      // the first rejector alone, appended
    seen[f] = make_float2(dist, __int_as_float(listed | (appended << 1)));
    const int mult = listed + appended;
    if (mult == 0) continue;
    const double w = (double)mult;
    const double sx = (double)x - px, sy = (double)y - py, sz = (double)z - pz;
    const double tx = (double)t.x - px, ty = (double)t.y - py, tz = (double)t.z - pz;
    const double term[kNumSums] = {w, w * sx, w * sy, w * sz, w * tx, w * ty, w * tz,
                                   w * (tx * sx), w * (tx * sy), w * (tx * sz), w * (ty * sx), w * (ty * sy), w * (ty * sz),
                                   w * (tz * sx), w * (tz * sy), w * (tz * sz), w * (double)dist};
#pragma unroll
    for (int k = 0; k < kNumSums; ++k) acc[k] += term[k];
  }
#pragma unroll
  for (int k = 0; k < kNumSums; ++k) s_lane[k][threadIdx.x] = acc[k];
  __syncthreads();
  if (threadIdx.x < kNumSums) {
    double v = 0.0;
    for (int l = 0; l < 64; ++l) v += s_lane[threadIdx.x][l];
    if (v != 0.0) unsafeAtomicAdd(S + threadIdx.x, v);   // (one addition per component into sums no other launch touches at this point)
  }
}

void launch_gather_fixed_pairs(hipStream_t stream, const CloudView &src, const CloudView &tgt, const uint32_t *pos, uint32_t n, float4 *fix) {
  hipLaunchKernelGGL(gather_fixed_pairs_kernel, dim3((n + 63u) / 64u), dim3(64), 0, stream, src, tgt, pos, n, fix);
}
void launch_icp_fixed_pairs(hipStream_t stream, const IcpState *st, const float4 *fix, uint32_t n, double *S, float2 *seen) {
  hipLaunchKernelGGL(icp_fixed_pairs_kernel, dim3(1), dim3(64), 0, stream, st, fix, n, S, seen);
}

// ------------------------------------------------------------------------------------------
// Stand-alone TransformationEstimationSVD on n given pairs (poseestimator.cpp:429-435):
// the same 17 sums, about the first source point, then the same umeyama lane.
__global__ __launch_bounds__(256) void pairs_sums_kernel(const float *__restrict__ src, const float *__restrict__ tgt,
                                                          uint32_t n, double *__restrict__ partials /*[17][gridDim.x]*/) {
  __shared__ double s_red[4][kNumSums];
  const float px = src[0], py = src[1], pz = src[2];
  double acc[kNumSums];
#pragma unroll
  for (int k = 0; k < kNumSums; ++k) acc[k] = 0.0;
  for (uint32_t i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) {
    const double sx = (double)src[3 * i] - px, sy = (double)src[3 * i + 1] - py, sz = (double)src[3 * i + 2] - pz;
    const double tx = (double)tgt[3 * i] - px, ty = (double)tgt[3 * i + 1] - py, tz = (double)tgt[3 * i + 2] - pz;
    acc[0] += 1.0;
    acc[1] += sx; acc[2] += sy; acc[3] += sz;
    acc[4] += tx; acc[5] += ty; acc[6] += tz;
    acc[7] += tx * sx; acc[8] += tx * sy; acc[9] += tx * sz;
    acc[10] += ty * sx; acc[11] += ty * sy; acc[12] += ty * sz;
    acc[13] += tz * sx; acc[14] += tz * sy; acc[15] += tz * sz;
  }
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
  for (int k = 0; k < kNumSums; ++k) {
    const double v = wave_sum(acc[k]);
    if (lane == 0) s_red[wave][k] = v;
  }
  __syncthreads();
  if (threadIdx.x < kNumSums)
    partials[threadIdx.x * gridDim.x + blockIdx.x] =
        s_red[0][threadIdx.x] + s_red[1][threadIdx.x] + s_red[2][threadIdx.x] + s_red[3][threadIdx.x];
}

__global__ __launch_bounds__(64) void pairs_umeyama_kernel(const double *__restrict__ partials, int nblocks,
                                                            const float *__restrict__ src, float *__restrict__ out_T) {
  __shared__ double s_S[kNumSums];
  if (threadIdx.x < kNumSums) {
    double v = 0.0;
    for (int b = 0; b < nblocks; ++b) v += partials[threadIdx.x * nblocks + b];
    s_S[threadIdx.x] = v;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    const double pivot[3] = {(double)src[0], (double)src[1], (double)src[2]};
    double T[16];
    umeyama_from_sums(s_S, pivot, T);
#pragma unroll
    for (int i = 0; i < 16; ++i) out_T[i] = (float)T[i];
  }
}

void launch_pairs_svd(hipStream_t stream, const float *d_src, const float *d_tgt, uint32_t n, double *d_partials,
                      int nblocks, float *d_out_T) {
  hipLaunchKernelGGL(pairs_sums_kernel, dim3(nblocks), dim3(256), 0, stream, d_src, d_tgt, n, d_partials);
  hipLaunchKernelGGL(pairs_umeyama_kernel, dim3(1), dim3(64), 0, stream, d_partials, nblocks, d_src, d_out_T);
}

// ------------------------------------------------------------------------------------------
// plain searches (ope_nn_search / ope_knn_search) and the fitness pass
__global__ __launch_bounds__(256) void nn_search_kernel(CloudView q, BvhView tgt, const float *__restrict__ T,
                                                         int has_T, int32_t *__restrict__ out_idx,
                                                         float *__restrict__ out_d2) {
  __shared__ float s_stk[kMaxDepth + 1][256];
  float *stk = &s_stk[0][threadIdx.x];
  float F[12];
#pragma unroll
  for (int i = 0; i < 12; ++i) F[i] = has_T ? T[i] : ((i % 5 == 0) ? 1.f : 0.f);
  const uint32_t lane_id = threadIdx.x & 63u;
  for (uint32_t base = blockIdx.x * 256 + (threadIdx.x & ~63u); base < q.n; base += gridDim.x * 256) {
    const uint32_t i = base + lane_id;
    const bool active = i < q.n_valid;
    const float4 s = q.xyzw[active ? i : 0];
    float x = s.x, y = s.y, z = s.z;
    if (has_T) {
      x = xform_row(F + 0, s.x, s.y, s.z);
      y = xform_row(F + 4, s.x, s.y, s.z);
      z = xform_row(F + 8, s.x, s.y, s.z);
    }
    NearestVisitor v{active ? INFINITY : -INFINITY, kNoPos, 0};
    if (active) bvh_traverse(tgt, x, y, z, v, stk, 256);
    if (i < q.n) {
      const bool found = active && v.pos != kNoPos;
      out_idx[i] = found ? __float_as_int(tgt.pts[v.pos].w) : -1;
      out_d2[i] = found ? v.best : INFINITY;
    }
  }
}

__global__ __launch_bounds__(kKnnBlock) void knn_search_kernel(CloudView q, BvhView tgt, const float *__restrict__ T,
                                                                int has_T, int k, int32_t *__restrict__ out_idx,
                                                                float *__restrict__ out_d2) {
  extern __shared__ unsigned char s_dyn[];
  float *ld = reinterpret_cast<float *>(s_dyn) + threadIdx.x;
  uint32_t *lp = reinterpret_cast<uint32_t *>(s_dyn + sizeof(float) * kKnnBlock * kKnnMaxK) + threadIdx.x;
  __shared__ float s_stk[kMaxDepth + 1][kKnnBlock];
  float *stk = &s_stk[0][threadIdx.x];
  float F[12];
#pragma unroll
  for (int i = 0; i < 12; ++i) F[i] = has_T ? T[i] : ((i % 5 == 0) ? 1.f : 0.f);
  const uint32_t lane_id = threadIdx.x & 63u;
  for (uint32_t base = blockIdx.x * kKnnBlock + (threadIdx.x & ~63u); base < q.n; base += gridDim.x * kKnnBlock) {
    const uint32_t i = base + lane_id;
    const bool active = i < q.n_valid;
    const float4 s = q.xyzw[active ? i : 0];
    float x = s.x, y = s.y, z = s.z;
    if (has_T) {
      x = xform_row(F + 0, s.x, s.y, s.z);
      y = xform_row(F + 4, s.x, s.y, s.z);
      z = xform_row(F + 8, s.x, s.y, s.z);
    }
    KnnVisitor v{ld, lp, kKnnBlock, k, 0, active ? INFINITY : -INFINITY};
    if (active) bvh_traverse(tgt, x, y, z, v, stk, kKnnBlock);
    if (i < q.n) {
      for (int j = 0; j < k; ++j) {
        const bool have = j < v.count;
        out_idx[(size_t)i * k + j] = have ? __float_as_int(tgt.pts[lp[j * kKnnBlock]].w) : -1;
        out_d2[(size_t)i * k + j] = have ? ld[j * kKnnBlock] : INFINITY;
      }
    }
  }
}

// Registration::getFitnessScore: partial sums {Σ d2 (d2 <= max_range), count} per block.
// hint (optional): per query the leaf that held its match in the ICP run that has just ended over the same pair — the walk
// starts there, as the run's own walks did (a start leaf never changes what a walk finds)
__global__ __launch_bounds__(256) void fitness_kernel(CloudView q, BvhView tgt, const float *__restrict__ T,
                                                       double max_range, double *__restrict__ partials, const uint32_t *__restrict__ hint) {
  __shared__ double s_red[4][2];
  __shared__ float s_stk[kMaxDepth + 1][256];
  float *stk = &s_stk[0][threadIdx.x];
  float F[12];
#pragma unroll
  for (int i = 0; i < 12; ++i) F[i] = T[i];
  double sum = 0.0, cnt = 0.0;
  const uint32_t lane_id = threadIdx.x & 63u;
  for (uint32_t base = blockIdx.x * 256 + (threadIdx.x & ~63u); base < q.n_valid; base += gridDim.x * 256) {
    const uint32_t i = base + lane_id;
    const bool active = i < q.n_valid;
    const float4 s = q.xyzw[active ? i : base];
    const float x = xform_row(F + 0, s.x, s.y, s.z);
    const float y = xform_row(F + 4, s.x, s.y, s.z);
    const float z = xform_row(F + 8, s.x, s.y, s.z);
    NearestVisitor v{active ? INFINITY : -INFINITY, kNoPos, 0};
    if (active) bvh_traverse(tgt, x, y, z, v, stk, 256, hint ? hint[i] : 0u);
    if (active && v.pos != kNoPos && (double)v.best <= max_range) { sum += (double)v.best; cnt += 1.0; }
  }
  sum = wave_sum(sum);
  cnt = wave_sum(cnt);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (lane == 0) { s_red[wave][0] = sum; s_red[wave][1] = cnt; }
  __syncthreads();
  if (threadIdx.x < 2) {
    partials[2 * blockIdx.x + threadIdx.x] =
        s_red[0][threadIdx.x] + s_red[1][threadIdx.x] + s_red[2][threadIdx.x] + s_red[3][threadIdx.x];
  }
}

// ------------------------------------------------------------------------------------------
// host launchers (called from api.hip)
void launch_icp_accumulate(hipStream_t stream, int nblocks, int mode, bool nrm, bool recip, const CloudView &src,
                           const BvhView &tgt, const BvhView &srcix, const IcpState *st, double *partials,
                           int32_t *corr_match, float *corr_d2, uint32_t *work_counter, uint32_t *hint,
                           const uint32_t *chunk_order, uint32_t *chunk_cost, const uint32_t *plan_info, bool packet,
                           int k_normal_shooting, double *S_atomic, const uint32_t *slot_list, float *knn_rk, const uint32_t *plan_out,
                           hipEvent_t e0, hipEvent_t e1, bool measuring, uint32_t *chain, uint32_t chain_seq, float4 *cert_q, uint32_t *cert_pos,
                           uint32_t *pace, uint32_t launch_no, uint32_t wait_ticks, float *cert_l) {
  const uint32_t mflag = measuring ? 1u : 0u;
  // e0 / e1 (ope_icp_profile): the launch's own start and stop time stamps, taken by the dispatch itself (hipExtLaunchKernelGGL).
  // Round 3 measured what a hipEventRecord before and after every launch costs the loop it times: 7-11 us per iteration (two
  // more packets with barriers between dependent dispatches), 5 % of the number the bench reports.
#define OPE_KLAUNCH(KERNEL, BLK, LDS)                                                                                         \
  do {                                                                                                                        \
    if (e0 != nullptr)                                                                                                        \
      hipExtLaunchKernelGGL(KERNEL, dim3(nblocks), dim3(BLK), LDS, stream, e0, e1, 0, src, tgt, srcix, st, partials, corr_match, \
                            corr_d2, work_counter, hint, chunk_order, chunk_cost, plan_info, S_atomic, slot_list, knn_rk, plan_out, mflag, chain, chain_seq, cert_q, cert_pos, pace, launch_no, wait_ticks, cert_l); \
    else                                                                                                                      \
      hipLaunchKernelGGL(KERNEL, dim3(nblocks), dim3(BLK), LDS, stream, src, tgt, srcix, st, partials, corr_match, corr_d2,   \
                         work_counter, hint, chunk_order, chunk_cost, plan_info, S_atomic, slot_list, knn_rk, plan_out, mflag, chain, chain_seq, cert_q, cert_pos, pace, launch_no, wait_ticks, cert_l); \
  } while (0)
#define OPE_LAUNCH_ACC(M, N, R, BLK, LDS) OPE_KLAUNCH((icp_accumulate_kernel<M, N, R>), BLK, LDS)
  const bool certify = cert_q != nullptr && mode == 0 && !recip;   // the certifying instantiation (api.hip decides when)
  if (mode == 0 && !recip && packet) {
    if (certify) { if (nrm) OPE_KLAUNCH((icp_accumulate_kernel<0, true, false, true, 20, true>), kAccBlock, 0); else OPE_KLAUNCH((icp_accumulate_kernel<0, false, false, true, 20, true>), kAccBlock, 0); }
    else if (nrm) OPE_KLAUNCH((icp_accumulate_kernel<0, true, false, true>), kAccBlock, 0);
    else OPE_KLAUNCH((icp_accumulate_kernel<0, false, false, true>), kAccBlock, 0);
    return;
  }
  if (mode == 0) {
    if (recip) { if (nrm) OPE_LAUNCH_ACC(0, true, true, kAccBlock, 0); else OPE_LAUNCH_ACC(0, false, true, kAccBlock, 0); }
    else if (certify) { if (nrm) OPE_KLAUNCH((icp_accumulate_kernel<0, true, false, false, 20, true>), kAccBlock, 0); else OPE_KLAUNCH((icp_accumulate_kernel<0, false, false, false, 20, true>), kAccBlock, 0); }
    else       { if (nrm) OPE_LAUNCH_ACC(0, true, false, kAccBlock, 0); else OPE_LAUNCH_ACC(0, false, false, kAccBlock, 0); }
  } else {
    // normal shooting: the k-nearest list lives in registers for EVERY k <= 32 (instantiations at k rounded up to a
    // multiple of four, and at the class default 10): the LDS list of round 1 took 2.4 ms per C3 iteration where the
    // register list takes ~0.5 ms, and held the kernel at two waves per SIMD
#define OPE_LAUNCH_NS(KR) OPE_KLAUNCH((icp_accumulate_kernel<2, true, false, false, KR>), kKnnBlock, 0)
    const int k = k_normal_shooting;
    if (k == 10) OPE_LAUNCH_NS(10);
    else if (k <= 4) OPE_LAUNCH_NS(4);
    else if (k <= 8) OPE_LAUNCH_NS(8);
    else if (k <= 12) OPE_LAUNCH_NS(12);
    else if (k <= 16) OPE_LAUNCH_NS(16);
    else if (k <= 20) OPE_LAUNCH_NS(20);
    else if (k <= 24) OPE_LAUNCH_NS(24);
    else if (k <= 28) OPE_LAUNCH_NS(28);
    else OPE_LAUNCH_NS(32);
#undef OPE_LAUNCH_NS
  }
#undef OPE_LAUNCH_ACC
#undef OPE_KLAUNCH
}

// Blocks of the 1-NN accumulate kernel (tree or grid instantiation) that one CU holds at a time: a launch of more blocks
// than the GPU holds runs its surplus after the first blocks have ended, behind the costliest chunks (C3 without
// clutter: 109 -> 94 us when the launch is cut to the resident number).
int icp_accumulate_blocks_per_cu(bool nrm, bool packet, bool grid) {
  int nb = 0;
  hipError_t e;
  if (grid) {
    e = nrm ? hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, icp_accumulate_grid_kernel<true>, kAccBlock, 0)
            : hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, icp_accumulate_grid_kernel<false>, kAccBlock, 0);
  }
  else if (packet) e = nrm ? hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, icp_accumulate_kernel<0, true, false, true>, kAccBlock, 0)
                           : hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, icp_accumulate_kernel<0, false, false, true>, kAccBlock, 0);
  else e = nrm ? hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, icp_accumulate_kernel<0, true, false, false>, kAccBlock, 0)
               : hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, icp_accumulate_kernel<0, false, false, false>, kAccBlock, 0);
  return (e == hipSuccess && nb > 0) ? nb : 0;
}
// the same for the certifying tree instantiations (128 VGPRs, 4 waves per SIMD: their walks and list builds would spill at 80,
// and a launch that answers from certificates is short of bandwidth, not of waves)
int icp_accumulate_cert_blocks_per_cu(bool nrm, bool packet, bool grid) {
  int nc = 0;
  hipError_t ec;
  if (grid) ec = nrm ? hipOccupancyMaxActiveBlocksPerMultiprocessor(&nc, icp_accumulate_grid_kernel<true, true>, kAccBlock, 0)
                     : hipOccupancyMaxActiveBlocksPerMultiprocessor(&nc, icp_accumulate_grid_kernel<false, true>, kAccBlock, 0);
  else if (packet) ec = nrm ? hipOccupancyMaxActiveBlocksPerMultiprocessor(&nc, icp_accumulate_kernel<0, true, false, true, 20, true>, kAccBlock, 0)
                       : hipOccupancyMaxActiveBlocksPerMultiprocessor(&nc, icp_accumulate_kernel<0, false, false, true, 20, true>, kAccBlock, 0);
  else ec = nrm ? hipOccupancyMaxActiveBlocksPerMultiprocessor(&nc, icp_accumulate_kernel<0, true, false, false, 20, true>, kAccBlock, 0)
                : hipOccupancyMaxActiveBlocksPerMultiprocessor(&nc, icp_accumulate_kernel<0, false, false, false, 20, true>, kAccBlock, 0);
  return (ec == hipSuccess && nc > 0) ? nc : 0;
}

void launch_icp_accumulate_grid(hipStream_t stream, int nblocks, bool nrm, const CloudView &src, const BvhView &tgt, const GridView &grid,
                                const IcpState *st, double *partials, int32_t *corr_match, float *corr_d2, uint32_t *hint, uint32_t *ghint,
                                const uint32_t *qorder, unsigned char *qclass, const uint32_t *chunk_order, uint32_t *chunk_cost,
                                const uint32_t *plan_info, double *S_atomic, hipEvent_t e0, hipEvent_t e1, bool measuring, uint32_t *chain,
                                uint32_t chain_seq, uint32_t *pace, uint32_t launch_no, uint32_t wait_ticks, float4 *cert_q, uint32_t *cert_pos,
                                uint32_t *cert_stats, float *cert_l) {
  const uint32_t mflag = measuring ? 1u : 0u;
#define OPE_KLAUNCH(KERNEL)                                                                                                   \
  do {                                                                                                                        \
    if (e0 != nullptr)                                                                                                        \
      hipExtLaunchKernelGGL(KERNEL, dim3(nblocks), dim3(kAccBlock), 0, stream, e0, e1, 0, src, tgt, grid, st, partials, corr_match, \
                            corr_d2, hint, ghint, qorder, qclass, chunk_order, chunk_cost, plan_info, S_atomic, mflag, chain, chain_seq, pace, launch_no, wait_ticks, cert_q, cert_pos, cert_stats, cert_l); \
    else                                                                                                                      \
      hipLaunchKernelGGL(KERNEL, dim3(nblocks), dim3(kAccBlock), 0, stream, src, tgt, grid, st, partials, corr_match, corr_d2, \
                         hint, ghint, qorder, qclass, chunk_order, chunk_cost, plan_info, S_atomic, mflag, chain, chain_seq, pace, launch_no, wait_ticks, cert_q, cert_pos, cert_stats, cert_l); \
  } while (0)
  if (cert_q != nullptr) { if (nrm) OPE_KLAUNCH((icp_accumulate_grid_kernel<true, true>)); else OPE_KLAUNCH((icp_accumulate_grid_kernel<false, true>)); }
  else if (nrm) OPE_KLAUNCH((icp_accumulate_grid_kernel<true>));
  else OPE_KLAUNCH((icp_accumulate_grid_kernel<false>));
#undef OPE_KLAUNCH
}

void launch_icp_reduce_update(hipStream_t stream, IcpState *st, const double *partials, double *S, int nblocks,
                              bool do_update, uint32_t *work_counter) {
  hipLaunchKernelGGL(icp_reduce_update_kernel, dim3(1), dim3(kRedBlock), 0, stream, st, partials, S, nblocks,
                     do_update ? 1 : 0, work_counter);
}

void launch_icp_p2p_update(hipStream_t stream, IcpState *st, double *S, int nsums, const P2pView &pv, uint32_t seq,
                           unsigned long long timeout_ticks, bool do_update, double *out_sums) {
  hipLaunchKernelGGL(icp_p2p_update_kernel, dim3(1), dim3(256), 0, stream, st, S, nsums, pv, seq, timeout_ticks, do_update ? 1 : 0, out_sums);
}

void launch_icp_lm_update(hipStream_t stream, IcpState *st, double *S, double *stats) {
  hipLaunchKernelGGL(icp_lm_update_kernel, dim3(1), dim3(64), 0, stream, st, S, stats);
}

void launch_icp_update_chained(hipStream_t stream, IcpState *st, int nsums, uint32_t *chain, uint32_t seq, uint32_t tickets, uint32_t wait_ticks) {
  hipLaunchKernelGGL(icp_update_chained_kernel, dim3(1), dim3(64), 0, stream, st, nsums, chain, seq, tickets, wait_ticks);
}

void launch_icp_update(hipStream_t stream, IcpState *st, double *S, int nsums, const float *Tk_ext) {
  hipLaunchKernelGGL(icp_update_kernel, dim3(1), dim3(64), 0, stream, st, S, nsums, Tk_ext);
}

void launch_nn_search(hipStream_t stream, const CloudView &q, const BvhView &tgt, const float *d_T, int32_t *out_idx,
                      float *out_d2) {
  const int nblocks = (int)std::min<size_t>((q.n + 255) / 256, 4096);
  hipLaunchKernelGGL(nn_search_kernel, dim3(std::max(nblocks, 1)), dim3(256), 0, stream, q, tgt, d_T, d_T ? 1 : 0,
                     out_idx, out_d2);
}

void launch_knn_search(hipStream_t stream, const CloudView &q, const BvhView &tgt, const float *d_T, int k,
                       int32_t *out_idx, float *out_d2) {
  const int nblocks = (int)std::min<size_t>((q.n + kKnnBlock - 1) / kKnnBlock, 4096);
  const size_t lds = kKnnLdsBytes;
  hipLaunchKernelGGL(knn_search_kernel, dim3(std::max(nblocks, 1)), dim3(kKnnBlock), lds, stream, q, tgt, d_T,
                     d_T ? 1 : 0, k, out_idx, out_d2);
}

// Start leaves for the FIRST launch of a k-NN run (normal shooting): a k-NN walk from the root has no bound until its list is full
// and opens far more of the tree than it needs (the first launch of a BuildModel pair took 1.7-2x a later one).  One lane per
// sixteen Morton-adjacent queries walks the tree for the first of them — a 1-NN walk, pruned from the first box on — and hands
// the leaf it ends in to all sixteen: their neighbours are in or next to it.  A start leaf never changes what a walk finds.
// (1-NN runs gain nothing from it: a 1-NN walk from the root prunes from its first leaf on — first launch of C3 / C2 / a clean
// 1 M-point cluster 850-1000 / 245-370 / 475-550 us with or without: measured, k-NN runs only.)
__global__ __launch_bounds__(256) void seed_hints_kernel(CloudView src, BvhView tgt, const IcpState *__restrict__ st, uint32_t *__restrict__ hint) {
  __shared__ float s_stk[kMaxDepth + 1][256];
  float *stk = &s_stk[0][threadIdx.x];
  const uint32_t n_groups = (src.n_valid + 15u) / 16u;
  for (uint32_t g = blockIdx.x * 256u + threadIdx.x; g < n_groups; g += gridDim.x * 256u) {
    const uint32_t i0 = g * 16u;
    const float4 s = src.xyzw[i0];
    const float x = xform_row(st->Ff + 0, s.x, s.y, s.z), y = xform_row(st->Ff + 4, s.x, s.y, s.z), z = xform_row(st->Ff + 8, s.x, s.y, s.z);
    NearestVisitor v{INFINITY, kNoPos, 0};
    bvh_traverse(tgt, x, y, z, v, stk, 256);
    if (v.leaf != 0u)
      for (uint32_t i = i0; i < min(i0 + 16u, src.n_valid); ++i) hint[i] = v.leaf;
  }
}
void launch_seed_hints(hipStream_t stream, const CloudView &src, const BvhView &tgt, const IcpState *st, uint32_t *hint) {
  const uint32_t n_groups = (src.n_valid + 15u) / 16u;
  if (n_groups == 0) return;
  hipLaunchKernelGGL(seed_hints_kernel, dim3(std::min<uint32_t>((n_groups + 255u) / 256u, 4096u)), dim3(256), 0, stream, src, tgt, st, hint);
}

void launch_fitness(hipStream_t stream, int nblocks, const CloudView &q, const BvhView &tgt, const float *d_T,
                    double max_range, double *partials, const uint32_t *hint) {
  hipLaunchKernelGGL(fitness_kernel, dim3(nblocks), dim3(256), 0, stream, q, tgt, d_T, max_range, partials, hint);
}

}  // namespace ope

#ifdef OPE_DEVELOPER   // `make DEVELOPER=1`: instrumentation kernels are not part of the product library
extern "C" int ope_debug_cert_reasons(ope_ctx *ctx, uint32_t out[4], int reset) {
  OPE_HIP(ctx, hipStreamSynchronize(ctx->stream));
  OPE_HIP(ctx, hipMemcpy(out, ctx->d_work_counter + 44, 16, hipMemcpyDeviceToHost));
  if (reset) OPE_HIP(ctx, hipMemset(ctx->d_work_counter + 44, 0, 16));
  return OPE_OK;
}
// tools/chain_probe.py: switch the per-chunk path/packet counters of the tree kernel on (device buffer of 4 words per
// chunk, handed back by ope_debug_chunk_stats_read) or off (nullptr)
extern "C" int ope_debug_chunk_stats(ope_ctx *ctx, uint32_t n_chunks, uint32_t *read_into) {
  static uint32_t *d_buf = nullptr;
  static uint32_t cap = 0;
  if (read_into) {
    if (!d_buf || n_chunks > cap) return OPE_ESTATE;
    OPE_HIP(ctx, hipStreamSynchronize(ctx->stream));
    OPE_HIP(ctx, hipMemcpy(read_into, d_buf, 16 * (size_t)n_chunks, hipMemcpyDeviceToHost));
    return OPE_OK;
  }
  if (d_buf) { (void)hipFree(d_buf); d_buf = nullptr; cap = 0; }
  if (n_chunks) {
    OPE_HIP(ctx, hipMalloc((void **)&d_buf, 16 * (size_t)n_chunks));
    OPE_HIP(ctx, hipMemset(d_buf, 0, 16 * (size_t)n_chunks));
    cap = n_chunks;
  }
  OPE_HIP(ctx, hipStreamSynchronize(ctx->stream));
  OPE_HIP(ctx, hipMemcpyToSymbol(HIP_SYMBOL(ope::g_chunk_stats), &d_buf, sizeof d_buf));
  return OPE_OK;
}

// ------------------------------------------------------------------------------------------
// Developer instrumentation (tools/visit_stats.py): per-query node / leaf-point visit counts of the
// private per-lane traversal.  Not part of include/ope.h.
namespace ope {
struct CountingVisitor {
  float best;
  int points, nodes;
  __device__ __forceinline__ bool prune(float bound) const { return !(bound < best); }
  __device__ __forceinline__ void point(float d, const v4f &, uint32_t, uint32_t) {
    ++points;
    if (d < best) best = d;
  }
  __device__ __forceinline__ void on_node() { ++nodes; }
};

__global__ __launch_bounds__(256) void debug_visit_kernel(CloudView q, BvhView t, const float *__restrict__ T,
                                                           int32_t *nodes_out, int32_t *points_out) {
  __shared__ float s_stk[kMaxDepth + 1][256];
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= q.n_valid) return;
  const float4 s = q.xyzw[i];
  const float qx = xform_row(T + 0, s.x, s.y, s.z), qy = xform_row(T + 4, s.x, s.y, s.z), qz = xform_row(T + 8, s.x, s.y, s.z);
  CountingVisitor v{INFINITY, 0, 0};
  bvh_traverse(t, qx, qy, qz, v, &s_stk[0][threadIdx.x], 256);
  nodes_out[i] = v.nodes;
  points_out[i] = v.points;
}
}  // namespace ope

extern "C" int ope_debug_visit_counts(ope_ctx *ctx, const ope_cloud *q, const ope_index *ix, const float *T_colmajor,
                                      int32_t *nodes, int32_t *points) {
  using namespace ope;
  const size_t n = q->n_valid;
  int32_t *d_n, *d_p;
  float *d_T;
  float rows[12];
  for (int r = 0; r < 3; ++r)
    for (int c = 0; c < 4; ++c) rows[4 * r + c] = T_colmajor[4 * c + r];
  OPE_HIP(ctx, hipMalloc((void **)&d_n, 4 * n));
  OPE_HIP(ctx, hipMalloc((void **)&d_p, 4 * n));
  OPE_HIP(ctx, hipMalloc((void **)&d_T, sizeof rows));
  OPE_HIP(ctx, h2d_copy(ctx->stream, d_T, rows, sizeof rows));
  hipLaunchKernelGGL(debug_visit_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, ctx->stream, q->view(), ix->view(),
                     d_T, d_n, d_p);
  OPE_HIP(ctx, hipStreamSynchronize(ctx->stream));
  std::vector<int32_t> hn(n), hp(n);
  OPE_HIP(ctx, hipMemcpy(hn.data(), d_n, 4 * n, hipMemcpyDeviceToHost));
  OPE_HIP(ctx, hipMemcpy(hp.data(), d_p, 4 * n, hipMemcpyDeviceToHost));
  for (size_t i = 0; i < n; ++i) { nodes[q->perm[i]] = hn[i]; points[q->perm[i]] = hp[i]; }
  (void)hipFree(d_n); (void)hipFree(d_p); (void)hipFree(d_T);
  return OPE_OK;
}

// ------------------------------------------------------------------------------------------
// Developer instrumentation (tools/chunk_profile.py): one wave per 64-query chunk, start hints taken
// from the last ICP run; per chunk the elapsed shader cycles and the lane-maximum step counts.
namespace ope {
struct StepVisitor {
  float best;
  uint32_t pos, leaf;
  int nodes, points;
  __device__ __forceinline__ bool prune(float bound) const { return !(bound < best); }
  __device__ __forceinline__ void point(float d, const v4f &, uint32_t i, uint32_t lf) {
    ++points;
    if (d < best) { best = d; pos = i; leaf = lf; }
  }
  __device__ __forceinline__ void on_node() { ++nodes; }
};

#define OPE_STAMP(var)                                                                    \
  do {                                                                                     \
    __builtin_amdgcn_sched_barrier(0);                                                     \
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(var)::"memory");            \
    __builtin_amdgcn_sched_barrier(0);                                                     \
  } while (0)

__global__ __launch_bounds__(kAccBlock, kAccWavesPerSimd) void debug_chunk_kernel(CloudView src, BvhView tgt, const float *__restrict__ T,
                                                                     const uint32_t *__restrict__ hint, int use_hint,
                                                                     long long *__restrict__ out) {
  __shared__ float s_stk[kMaxDepth + 1][kAccBlock];
  const uint32_t lane_id = threadIdx.x & 63u;
  const uint32_t chunk = blockIdx.x * (kAccBlock / 64) + (threadIdx.x >> 6);
  const uint32_t base = chunk * 64u;
  if (base >= src.n_valid) return;
  const uint32_t i = base + lane_id;
  const bool active = i < src.n_valid;
  unsigned long long t0, t1, ta, tb;
  unsigned long long c_node = 0, c_leaf = 0, c_pop = 0, c_eager = 0, n_node = 0;
  int my_trips = 0;  // trips in which THIS lane still had work: sum over lanes / (64 * trips) = lane utilisation of the walk
  OPE_STAMP(t0);
  const float4 s4 = src.xyzw[active ? i : base];
  const float qx = xform_row(T + 0, s4.x, s4.y, s4.z), qy = xform_row(T + 4, s4.x, s4.y, s4.z), qz = xform_row(T + 8, s4.x, s4.y, s4.z);
  StepVisitor v{active ? INFINITY : -INFINITY, kNoPos, 0, 0, 0};
  // ---- instrumented copy of bvh_traverse's flat loop (no LDS top copy) ----
  const BvhView &t = tgt;
  float *stk = &s_stk[0][threadIdx.x];
  const int stk_stride = kAccBlock;
  const uint32_t leaf0 = 1u << t.depth;
  uint32_t node = 1, trail = 0;
  const uint32_t start_leaf = (active && use_hint) ? hint[i] : 0u;
  bool alive = active;
  OPE_STAMP(ta);
  if (alive && start_leaf != 0) {
    node = start_leaf;
    trail = leaf0 - 1u;
    const int D = t.depth;
    for (int k = 0; k < D; k += 4) {
      v4f a0, b0, c0, a1, b1, c1, a2, b2, c2, a3, b3, c3;
      const uint32_t s0 = (start_leaf >> k) ^ 1u;
      const uint32_t s1 = (k + 1 < D) ? ((start_leaf >> (k + 1)) ^ 1u) : s0;
      const uint32_t s2 = (k + 2 < D) ? ((start_leaf >> (k + 2)) ^ 1u) : s0;
      const uint32_t s3 = (k + 3 < D) ? ((start_leaf >> (k + 3)) ^ 1u) : s0;
      load_node(t, s0, a0, b0, c0); load_node(t, s1, a1, b1, c1);
      load_node(t, s2, a2, b2, c2); load_node(t, s3, a3, b3, c3);
      stk[(D - k) * stk_stride] = obb_dist2(a0, b0, c0, qx, qy, qz);
      if (k + 1 < D) stk[(D - k - 1) * stk_stride] = obb_dist2(a1, b1, c1, qx, qy, qz);
      if (k + 2 < D) stk[(D - k - 2) * stk_stride] = obb_dist2(a2, b2, c2, qx, qy, qz);
      if (k + 3 < D) stk[(D - k - 3) * stk_stride] = obb_dist2(a3, b3, c3, qx, qy, qz);
    }
  }
  OPE_STAMP(tb);
  c_eager += tb - ta;
  while (__ballot(alive) != 0ull) {
    bool do_pop = false;
    my_trips += alive ? 1 : 0;
    OPE_STAMP(ta);
    if (alive && node < leaf0) {
      v.on_node();
      v4f c0, c1, c2, c3, c4, c5;
      load_node(t, 2 * node, c0, c1, c2);
      load_node(t, 2 * node + 1, c3, c4, c5);
      const float d0 = obb_dist2(c0, c1, c2, qx, qy, qz);
      const float d1 = obb_dist2(c3, c4, c5, qx, qy, qz);
      const bool right = d1 < d0;
      const float dn = right ? d1 : d0, df = right ? d0 : d1;
      if (!v.prune(dn)) {
        node = 2 * node + (right ? 1u : 0u);
        const bool pend = !v.prune(df);
        trail = (trail << 1) | (pend ? 1u : 0u);
        if (pend) stk[(31 - __clz(node)) * stk_stride] = df;
      } else {
        do_pop = true;
      }
    }
    OPE_STAMP(tb);
    c_node += tb - ta; n_node += 1;
    const bool leaf_now = alive && !do_pop && node >= leaf0 && false;
    (void)leaf_now;
    OPE_STAMP(ta);
    if (alive && !do_pop && node >= leaf0) {
      const uint32_t j = node - leaf0;
      const uint32_t sb = (uint32_t)(((unsigned long long)j * t.n) >> t.depth);
      const uint32_t e = (uint32_t)(((unsigned long long)(j + 1) * t.n) >> t.depth);
      for (uint32_t q = sb; q < e; q += 4) {
        const uint32_t i1 = min(q + 1, e - 1), i2 = min(q + 2, e - 1), i3 = min(q + 3, e - 1);
        const v4f p0 = ld16(t.pts + q), p1 = ld16(t.pts + i1), p2 = ld16(t.pts + i2), p3 = ld16(t.pts + i3);
        v.point(sq_dist3(__fsub_rn(qx, p0.x), __fsub_rn(qy, p0.y), __fsub_rn(qz, p0.z)), p0, q, node);
        if (q + 1 < e) v.point(sq_dist3(__fsub_rn(qx, p1.x), __fsub_rn(qy, p1.y), __fsub_rn(qz, p1.z)), p1, i1, node);
        if (q + 2 < e) v.point(sq_dist3(__fsub_rn(qx, p2.x), __fsub_rn(qy, p2.y), __fsub_rn(qz, p2.z)), p2, i2, node);
        if (q + 3 < e) v.point(sq_dist3(__fsub_rn(qx, p3.x), __fsub_rn(qy, p3.y), __fsub_rn(qz, p3.z)), p3, i3, node);
      }
      do_pop = true;
    }
    OPE_STAMP(tb);
    c_leaf += tb - ta;
    OPE_STAMP(ta);
    if (alive && do_pop) {
      for (;;) {
        if (trail == 0) { alive = false; break; }
        const int k = __builtin_ctz(trail);
        node = (node >> k) ^ 1u;
        trail = (trail >> k) & ~1u;
        if (!v.prune(stk[(31 - __clz(node)) * stk_stride])) break;
      }
    }
    OPE_STAMP(tb);
    c_pop += tb - ta;
  }
  OPE_STAMP(t1);
  int mn = v.nodes, mp = v.points;
  int st = my_trips;
  for (int off = 32; off >= 1; off >>= 1) { mn = max(mn, __shfl_xor(mn, off, 64)); mp = max(mp, __shfl_xor(mp, off, 64)); st += __shfl_xor(st, off, 64); }
  if (lane_id == 0) {
    long long *o = out + 10 * (size_t)chunk;
    o[0] = (long long)(t1 - t0); o[1] = mn; o[2] = mp; o[3] = (long long)c_eager; o[4] = (long long)c_node;
    o[5] = (long long)c_leaf; o[6] = (long long)c_pop; o[7] = (long long)n_node; o[8] = st; o[9] = 0;
  }
}
}  // namespace ope

extern "C" int ope_debug_chunk_profile(ope_ctx *ctx, const ope_cloud *q, const ope_index *ix, const float *T_colmajor,
                                       int use_hint, long long *out /* n_chunks * 6 */) {
  using namespace ope;
  const size_t nch = (q->n_valid + 63) / 64;
  float rows[12];
  for (int r = 0; r < 3; ++r)
    for (int c = 0; c < 4; ++c) rows[4 * r + c] = T_colmajor[4 * c + r];
  float *d_T;
  long long *d_out;
  OPE_HIP(ctx, hipMalloc((void **)&d_T, sizeof rows));
  OPE_HIP(ctx, hipMalloc((void **)&d_out, sizeof(long long) * 10 * nch));
  OPE_HIP(ctx, hipMemset(d_out, 0, sizeof(long long) * 10 * nch));
  OPE_HIP(ctx, h2d_copy(ctx->stream, d_T, rows, sizeof rows));
  const unsigned nb = (unsigned)((nch + kAccBlock / 64 - 1) / (kAccBlock / 64));
  hipLaunchKernelGGL(debug_chunk_kernel, dim3(nb), dim3(kAccBlock), 0, ctx->stream, q->view(), ix->view(), d_T, ctx->d_hint,
                     (use_hint && ctx->d_hint) ? 1 : 0, d_out);
  OPE_HIP(ctx, hipStreamSynchronize(ctx->stream));
  OPE_HIP(ctx, hipMemcpy(out, d_out, sizeof(long long) * 10 * nch, hipMemcpyDeviceToHost));
  (void)hipFree(d_T); (void)hipFree(d_out);
  return OPE_OK;
}
#endif  // OPE_DEVELOPER

#ifdef OPE_KNN_STATS
extern "C" int ope_dev_knn_stats(unsigned long long out[8], int reset) {
  if (hipMemcpyFromSymbol(out, HIP_SYMBOL(ope::g_knn_stats), 64) != hipSuccess) return -1;
  if (reset) { unsigned long long z[8] = {0, 0, 0, 0, 0, 0, 0, 0}; if (hipMemcpyToSymbol(HIP_SYMBOL(ope::g_knn_stats), z, 64) != hipSuccess) return -1; }
  return 0;
}
#endif
