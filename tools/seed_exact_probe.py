"""Developer probe (DEVELOPER build, PROBE_LIB=dev): the start leaves seeded for the first k-NN launch must not change a
correspondence — one and three normal-shooting iterations of a BuildModel-like pair with and without them (OPE_NO_SEED)."""
import importlib, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
ope = importlib.import_module("object-pose-estimation_amd")
ope.LIB_PATH = os.path.join(os.path.dirname(ope.LIB_PATH), "libope_hip_dev.so")
synth = importlib.import_module("object-pose-estimation_amd.synth")
ctx = ope.Context(0)
frames = synth.frame_views(5, 500_000, n_azimuths=32)
src = np.concatenate(frames[:4]); tgt = frames[4]
cs = ctx.upload(src); ctx.normals(cs, 12, fetch=False)
ct = ctx.upload(tgt); ctx.normals(ct, 12, fetch=False)
ix = ctx.build_index(ct)
res = {}
for K in (1, 3):
    for seed in (True, False):
        if seed: os.environ.pop("OPE_NO_SEED", None)
        else: os.environ["OPE_NO_SEED"] = "1"
        p = ope.default_icp_params(max_iterations=K, transformation_epsilon=0.0, euclidean_fitness_epsilon=0.0, mse_threshold_absolute=-1.0,
                                   corr_mode=ope.CORR_NORMAL_SHOOTING, k_normal_shooting=20, use_surface_normal_rej=1, surface_normal_thr=0.7)
        out = ctx.icp(cs, ix, p)
        q, m, d = ctx.icp_correspondences(len(src))
        res[(K, seed)] = (q.copy(), m.copy(), d.copy(), out.T.copy())
    a, b = res[(K, True)], res[(K, False)]
    same_q = len(a[0]) == len(b[0]) and (a[0] == b[0]).all()
    print(f"K={K}: n_corr {len(a[0])} / {len(b[0])}, queries equal {same_q}, matches differing {int((a[1] != b[1]).sum()) if same_q else -1}, "
          f"d2 differing {int((a[2] != b[2]).sum()) if same_q else -1}, |T - T| {np.abs(a[3] - b[3]).max():.2e}")
