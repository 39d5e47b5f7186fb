// trace.cpp — optional roctx ranges around the host side of the hot path (the reference's only
// instrumentation is three pcl::ScopeTime blocks; the GPU side gets named ranges that rocprofv3 --marker-trace shows).
// Ranges: icp_iter { nn, reduce }, normals, fpfh_spfh, fpfh_weight, sacia, index_build, uniform_sampling.
// Off by default (ope_ctx_set_tracing); the roctx library is dlopen'ed on first use, so there is no link-time dependency.
#include <dlfcn.h>

#include <cstring>

#include "ope_internal.hpp"

namespace ope {

namespace {
struct Roctx {
  bool tried = false;
  int (*push)(const char *) = nullptr;
  int (*pop)() = nullptr;
};
Roctx &roctx() {
  static Roctx r;
  if (r.tried) return r;
  r.tried = true;
  void *h = nullptr;
  for (const char *name : {"librocprofiler-sdk-roctx.so.1", "libroctx64.so.4", "libroctx64.so"}) {
    h = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
    if (h) break;
  }
  if (!h) return r;
  r.push = reinterpret_cast<int (*)(const char *)>(dlsym(h, "roctxRangePushA"));
  r.pop = reinterpret_cast<int (*)()>(dlsym(h, "roctxRangePop"));
  if (!r.push || !r.pop) r.push = nullptr, r.pop = nullptr;
  return r;
}
}  // namespace

TraceRange::TraceRange(const ope_ctx *ctx, const char *name) : on_(false) {
  if (!ctx || !ctx->tracing) return;
  Roctx &r = roctx();
  if (!r.push) return;
  r.push(name);
  on_ = true;
}
TraceRange::~TraceRange() {
  if (on_) roctx().pop();
}

static hipEvent_t take_event(ope_ctx *ctx) {
  if (!ctx->kevent_pool.empty()) { hipEvent_t e = ctx->kevent_pool.back(); ctx->kevent_pool.pop_back(); return e; }
  hipEvent_t e = nullptr;
  if (hipEventCreate(&e) != hipSuccess) return nullptr;
  return e;
}

KernelTimer::KernelTimer(ope_ctx *ctx, const char *name, double algorithmic_bytes, bool start_now)
    : ctx_(ctx), name_(name), bytes_(algorithmic_bytes), slot_(-1), open_(false) {
  if (start_now) start();
}
void KernelTimer::start() {
  if (!ctx_ || !ctx_->ktime_on || slot_ >= 0) return;
  hipEvent_t e0 = take_event(ctx_), e1 = take_event(ctx_);
  if (!e0 || !e1) return;
  if (hipEventRecord(e0, ctx_->stream) != hipSuccess) return;
  slot_ = (int)ctx_->kstamps.size();
  ctx_->kstamps.push_back({name_, e0, e1, bytes_});
  open_ = true;
}
void KernelTimer::stop() {
  if (!open_) return;
  open_ = false;
  (void)hipEventRecord(ctx_->kstamps[(size_t)slot_].e1, ctx_->stream);
}
void KernelTimer::set_bytes(double b) {
  bytes_ = b;
  if (slot_ >= 0) ctx_->kstamps[(size_t)slot_].bytes = b;
}

}  // namespace ope

extern "C" int ope_profile_kernels(ope_ctx *ctx, int on) {
  if (!ctx) return OPE_EINVAL;
  if (hipStreamSynchronize(ctx->stream) != hipSuccess) return OPE_EHIP;
  for (auto &k : ctx->kstamps) { ctx->kevent_pool.push_back(k.e0); ctx->kevent_pool.push_back(k.e1); }
  ctx->kstamps.clear();
  ctx->ktime_on = on != 0;
  return OPE_OK;
}

extern "C" int ope_profile_kernels_read(ope_ctx *ctx, ope_kernel_time *out, size_t cap, size_t *n_out) {
  if (!ctx || !n_out || (cap && !out)) return OPE_EINVAL;
  if (hipStreamSynchronize(ctx->stream) != hipSuccess) return OPE_EHIP;
  size_t n = 0;
  for (const auto &k : ctx->kstamps) {
    float ms = 0.f;
    if (hipEventElapsedTime(&ms, k.e0, k.e1) != hipSuccess) continue;
    size_t j = 0;
    for (; j < n && j < cap; ++j)
      if (std::strncmp(out[j].name, k.name, sizeof out[j].name) == 0) break;
    if (j == n) {
      if (n < cap) {
        std::memset(&out[n], 0, sizeof out[n]);
        std::strncpy(out[n].name, k.name, sizeof out[n].name - 1);
      }
      ++n;
    }
    if (j < cap) { out[j].ms += ms; out[j].launches += 1; out[j].algorithmic_bytes += k.bytes; }
  }
  *n_out = n;
  return OPE_OK;
}

extern "C" int ope_ctx_set_tracing(ope_ctx *ctx, int on) {
  if (!ctx) return OPE_EINVAL;
  ctx->tracing = on != 0;
  return OPE_OK;
}
