"""Per-launch time of the normal-shooting search on a BuildModel-like pair (accumulated source of several views against one view):
how much of a pair is its first, unhinted launch."""
import importlib, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
ope = importlib.import_module("object-pose-estimation_amd"); synth = importlib.import_module("object-pose-estimation_amd.synth")
ctx = ope.Context(0)
frames = synth.frame_views(9, 500_000, n_azimuths=32)
for nsrc in (1, 8):
    src = np.concatenate(frames[:nsrc]); tgt = frames[nsrc if nsrc < 8 else 8]
    cs = ctx.upload(src); ctx.normals(cs, 12, fetch=False)
    ct = ctx.upload(tgt); ctx.normals(ct, 12, fetch=False)
    ix = ctx.build_index(ct)
    p = ope.default_icp_params(max_iterations=12, transformation_epsilon=0.0, euclidean_fitness_epsilon=0.0, mse_threshold_absolute=-1.0, check_every=0,
                               corr_mode=ope.CORR_NORMAL_SHOOTING, k_normal_shooting=20, use_surface_normal_rej=1, surface_normal_thr=0.7)
    ctx.icp_begin(cs, ix, p)
    ctx.icp_profile(12)
    ctx.icp_iterate(12); ctx.sync()
    ms = ctx.icp_profile_launches()
    ctx.icp_end()
    print(f"source {len(src)} points, target {len(tgt)}: per launch ms", " ".join(f"{v:.2f}" for v in ms), f"| first launch = {ms[0] / ms.sum() * 100:.0f} % of 12", flush=True)
