"""Developer probe: run-to-run determinism of the C1 facade (tests/test_gpu_detect_and_localize.py's dense two-frame case): the
facade binary N times, the test's own preparation of the fine stage's inputs N times, the fine ICP on them N times."""
import hashlib, importlib, os, subprocess, sys, tempfile
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
ope = importlib.import_module("object-pose-estimation_amd")
pcd = importlib.import_module("object-pose-estimation_amd.pcd")
synth = importlib.import_module("object-pose-estimation_amd.synth")
EXE = os.path.join(ROOT, "object-pose-estimation_amd", "build", "detect_and_localize")
N = int(sys.argv[1]) if len(sys.argv) > 1 else 6
def h(a): return hashlib.md5(np.ascontiguousarray(a).tobytes()).hexdigest()[:10]

model = synth.model_surface(30_000, 1)
gt = np.eye(4); gt[:3, :3] = synth.rot_xyz(20.0, -15.0, 40.0); gt[:3, 3] = [0.03, -0.02, 0.7]
scene = (synth.model_surface(30_000, 2).astype(np.float64) @ gt[:3, :3].T + gt[:3, 3]).astype(np.float32)
M = np.eye(4); M[:3, :3] = synth.rot_xyz(0.5, 1.0, -1.0); M[:3, 3] = [0.002, 0.001, -0.002]
scene2 = (scene.astype(np.float64) @ M[:3, :3].T + M[:3, 3]).astype(np.float32)
tmp = tempfile.mkdtemp()
mp_, p1, p2 = os.path.join(tmp, "model.pcd"), os.path.join(tmp, "s1.pcd"), os.path.join(tmp, "s2.pcd")
pcd.write_pcd(mp_, model); pcd.write_pcd(p1, scene); pcd.write_pcd(p2, scene2)
for rep in range(N):
    r = subprocess.run([EXE, mp_, p1, p2, "--seed", "3"], capture_output=True, text=True, timeout=600)
    out = []
    for line in r.stdout.splitlines():
        if line.startswith("frame "):
            tok = line.split()
            out.append(f"it {tok[9]} coarse {h(np.array(tok[28:44], np.float64))} fine {h(np.array(tok[45:61], np.float64))} all {h(np.array([float(t) for t in tok[11:27]]))}")
    print(f"facade run {rep}: rc {r.returncode} | " + " | ".join(out), flush=True)

# the fine stage of frame 0 in this process: the facade's coarse pose applied to the model as the facade's host code applies it
if os.environ.get("PROBE_LIB"):
    ope.LIB_PATH = os.path.join(os.path.dirname(ope.LIB_PATH), f"libope_hip_{os.environ['PROBE_LIB']}.so")
tok = [l for l in r.stdout.splitlines() if l.startswith("frame ")][0].split()
Tc = np.array([float(v) for v in tok[28:44]]).reshape(4, 4).T.astype(np.float32)
P = model.astype(np.float32)
aligned = np.stack([((Tc[r_, 0] * P[:, 0] + Tc[r_, 1] * P[:, 1]) + Tc[r_, 2] * P[:, 2]) + Tc[r_, 3] for r_ in range(3)], axis=1).astype(np.float32)
ctx = ope.Context(0)
def prep(cloud):
    cloud = cloud[np.isfinite(cloud).all(1)]
    keys = cloud[ctx.uniform_sampling(ctx.upload(cloud), 0.008)]
    nrm, _ = ctx.normals(ctx.upload(keys), 30)
    ok = np.isfinite(nrm).all(1)
    return keys[ok], nrm[ok]
p = ope.default_icp_params(max_iterations=100, transformation_epsilon=1e-8, euclidean_fitness_epsilon=1e-8, corr_mode=ope.CORR_NORMAL_SHOOTING,
                           k_normal_shooting=20, use_surface_normal_rej=1, surface_normal_thr=0.7)
for det in (0, 1):
    p.deterministic_sums = det
    for rep in range(N):
        sk, sn = prep(aligned); tk, tn = prep(scene)
        cs = ctx.upload(sk, sn); ct = ctx.upload(tk, tn); ix = ctx.build_index(ct)
        o = ctx.icp(cs, ix, p)
        print(f"in-process det={det} {rep}: keys {h(sk)} {h(tk)} normals {h(sn)} {h(tn)} | icp it {o.iterations} T {h(o.T)} {np.asarray(o.T).ravel()[[3, 7, 11]]}", flush=True)
ctx.close()
