"""Developer probe: (1) the k-NN search bound must not change a single correspondence: normal-shooting ICP on the C1 fixture
with the bound (default) and without it (OPE_NO_KNN_BOUND=1, DEVELOPER build) — run this script twice and diff the dumps;
(2) run-to-run determinism of uniform sampling, normals and FPFH inside one process with other work in between."""
import importlib, os, sys, hashlib
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
ope = importlib.import_module("object-pose-estimation_amd")
if os.environ.get("PROBE_LIB"):
    ope.LIB_PATH = os.path.join(os.path.dirname(ope.LIB_PATH), f"libope_hip_{os.environ['PROBE_LIB']}.so")
pcd = importlib.import_module("object-pose-estimation_amd.pcd")
synth = importlib.import_module("object-pose-estimation_amd.synth")
GOLD = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")
model, _ = pcd.read_pcd(os.path.join(GOLD, "drill_model_decimated.pcd"))
scene = np.load(os.path.join(GOLD, "drill_scene_c1.npz"))["scene"]
ctx = ope.Context(0)
def h(a): return hashlib.md5(np.ascontiguousarray(a).tobytes()).hexdigest()[:10]
def prep(cloud):
    cloud = cloud[np.isfinite(cloud).all(1)]
    keys = cloud[ctx.uniform_sampling(ctx.upload(cloud), 0.008)]
    nrm, _ = ctx.normals(ctx.upload(keys), 30)
    ok = np.isfinite(nrm).all(1)
    return keys[ok], nrm[ok]
for rep in range(3):
    sk, sn = prep(model); tk, tn = prep(scene)
    big = ctx.upload(synth.scene_cloud(300_000)); ctx.build_index(big)      # other work in between (recycles the pool)
    print(f"rep {rep}: model keys {h(sk)} normals {h(sn)} scene keys {h(tk)} normals {h(tn)}")
# a guess that puts the model roughly on the scene: the fixture's ground truth
g = np.load(os.path.join(GOLD, "drill_scene_c1.npz"))
guess = g["T_gt"] if "T_gt" in g.files else None
cs = ctx.upload(sk, sn); ct = ctx.upload(tk, tn); ix = ctx.build_index(ct)
p = ope.default_icp_params(max_iterations=100, transformation_epsilon=1e-8, euclidean_fitness_epsilon=1e-8, corr_mode=ope.CORR_NORMAL_SHOOTING,
                           k_normal_shooting=20, use_surface_normal_rej=1, surface_normal_thr=0.7)
for K in (1, 2, 3, 5, 10, 30, 100):
    pk = ope.default_icp_params(max_iterations=K, transformation_epsilon=0.0, euclidean_fitness_epsilon=0.0, mse_threshold_absolute=-1.0,
                                corr_mode=ope.CORR_NORMAL_SHOOTING, k_normal_shooting=20, use_surface_normal_rej=1, surface_normal_thr=0.7)
    out = ctx.icp(cs, ix, pk, guess)
    q, m, d = ctx.icp_correspondences(len(sk))
    print(f"K={K:3d}: T {h(out.T)} n_corr {out.n_corr} corr q {h(q)} m {h(m)} d {h(d)}")
out = ctx.icp(cs, ix, p, guess)
print(f"converging run: iterations {out.iterations} T {h(out.T)} state {out.state}")
ctx.close()
