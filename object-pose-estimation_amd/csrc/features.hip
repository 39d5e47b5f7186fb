// features.hip — coarse-stage kernels (normals, FPFH, uniform sampling, SAC-IA).  [work in progress]
#include "ope_internal.hpp"
using namespace ope;
extern "C" {
int ope_radius_search(ope_ctx *ctx, const ope_cloud *, const ope_index *, float, int, int32_t *, int32_t *, float *) {
  return set_err(ctx, OPE_ESTATE, "ope_radius_search: not implemented yet");
}
int ope_normals(ope_ctx *ctx, ope_cloud *, int, const float *, float *, float *) {
  return set_err(ctx, OPE_ESTATE, "ope_normals: not implemented yet");
}
int ope_fpfh(ope_ctx *ctx, const ope_cloud *, float, float *) { return set_err(ctx, OPE_ESTATE, "ope_fpfh: not implemented yet"); }
int ope_uniform_sampling(ope_ctx *ctx, const ope_cloud *, float, int32_t *, size_t *) {
  return set_err(ctx, OPE_ESTATE, "ope_uniform_sampling: not implemented yet");
}
void ope_sacia_default_params(ope_sacia_params *p) {
  if (!p) return;
  p->max_iterations = 400; p->nr_samples = 5; p->k_correspondences = 5; p->max_corr_dist = 0.05;
  p->min_sample_dist = 0.01f; p->seed = 1;
}
int ope_sacia(ope_ctx *ctx, const ope_cloud *, const float *, const ope_cloud *, const ope_index *, const float *,
              const ope_sacia_params *, const int32_t *, float *, double *, int32_t *) {
  return set_err(ctx, OPE_ESTATE, "ope_sacia: not implemented yet");
}
}
