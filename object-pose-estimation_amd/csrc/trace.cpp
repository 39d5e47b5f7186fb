// trace.cpp — optional roctx ranges around the host side of the hot path (the reference's only
// instrumentation is three pcl::ScopeTime blocks; the GPU side gets named ranges that rocprofv3 --marker-trace shows).
// Ranges: icp_iter { nn, reduce }, normals, fpfh_spfh, fpfh_weight, sacia, index_build, uniform_sampling.
// Off by default (ope_ctx_set_tracing); the roctx library is dlopen'ed on first use, so there is no link-time dependency.
#include <dlfcn.h>

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

}  // namespace ope

extern "C" int ope_ctx_set_tracing(ope_ctx *ctx, int on) {
  if (!ctx) return OPE_EINVAL;
  ctx->tracing = on != 0;
  return OPE_OK;
}
