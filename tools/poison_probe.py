"""Developer probe: hashes of what the front-end and search entry points return.  Run once on the product library and once
on a build whose temporaries start out poisoned (make VARIANT=poison EXTRA=-DOPE_POISON_TMP; PROBE_LIB=poison): any
difference is a read of memory nobody wrote."""
import importlib, os, sys, hashlib
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
ope = importlib.import_module("object-pose-estimation_amd")
if os.environ.get("PROBE_LIB"):
    ope.LIB_PATH = os.path.join(os.path.dirname(ope.LIB_PATH), f"libope_hip_{os.environ['PROBE_LIB']}.so")
pcd = importlib.import_module("object-pose-estimation_amd.pcd")
synth = importlib.import_module("object-pose-estimation_amd.synth")
GOLD = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")
def h(a): return hashlib.md5(np.ascontiguousarray(a).tobytes()).hexdigest()[:10]
ctx = ope.Context(0)
model, _ = pcd.read_pcd(os.path.join(GOLD, "drill_model_decimated.pcd"))
scene = np.load(os.path.join(GOLD, "drill_scene_c1.npz"))["scene"]
for name, cloud in (("drill", model), ("scene", scene), ("synth60k", synth.scene_cloud(60_000)), ("synth1k", synth.model_surface(1000, 3))):
    for rep in range(2):
        c = ctx.upload(cloud)
        ks = ctx.uniform_sampling(c, 0.008)
        kc = ctx.upload(cloud[ks])
        nrm, curv = ctx.normals(kc, 30)
        f = ctx.fpfh(kc, 0.03)
        idx, d2 = ctx.nn(c, ctx.build_index(kc))
        ki, kd = ctx.knn(kc, ctx.build_index(c), 12)
        sor = ctx.statistical_outlier_removal(c, 30, 1.0)
        pt = ctx.pass_through(c, [-0.05, -0.05, -1], [0.05, 0.08, 1])
        vg = ctx.voxel_grid(c, 0.01)
        cc, ci = ctx.uniform_sampling_cloud(c, 0.01, want_idx=True)
        print(f"{name} rep {rep}: keys {h(ks)} nrm {h(np.nan_to_num(nrm))} curv {h(np.nan_to_num(curv))} fpfh {h(np.nan_to_num(f))} nn {h(idx)} {h(d2)} knn {h(ki)} {h(kd)} sor {h(sor)} pt {h(pt)} vg {h(vg)} usc {h(ci)} {h(ctx.download(cc))}", flush=True)
src = synth.scene_cloud(200_000); tgt = synth.model_surface(30_000, 1)
cs = ctx.upload(src); ct = ctx.upload(tgt)
for g in (0, 2, 1):
    ix = ctx.build_index(ct, grid=g)
    out = ctx.icp(cs, ix, ope.default_icp_params(max_iterations=12, mse_threshold_absolute=-1.0, check_every=0, deterministic_sums=0))
    q, m, d = ctx.icp_correspondences(len(src))
    print(f"icp grid={g}: n_corr {out.n_corr} m {h(m)} d {h(d)} T {np.round(out.T[:3,3],6).tolist()}", flush=True)
ctx.normals(cs, 30, fetch=False); ctx.normals(ct, 12, fetch=False)
ix = ctx.build_index(ct)
for est in (ope.EST_SVD, ope.EST_POINT_TO_PLANE_LLS, ope.EST_POINT_TO_PLANE_LM):
    out = ctx.icp(cs, ix, ope.default_icp_params(max_iterations=8, mse_threshold_absolute=-1.0, check_every=0, corr_mode=ope.CORR_NORMAL_SHOOTING, k_normal_shooting=20,
                                                 use_surface_normal_rej=1, estimator=est))
    q, m, d = ctx.icp_correspondences(len(src))
    print(f"ns est={est}: n_corr {out.n_corr} m {h(m)} d {h(d)} T {np.round(out.T[:3,3],6).tolist()}", flush=True)
ctx.close()
