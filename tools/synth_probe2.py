"""CPU probe: SAC-IA success rate over seeds for a model-shape variant (oracle only)."""
import importlib, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import oracle
synth = importlib.import_module("object-pose-estimation_amd.synth")

def run(tag, crop=0.01, seeds=range(1, 9), n_scene=1_000_000, dens=0):
    scene = synth.scene_cloud(n_scene); model = synth.model_surface(100_000, 1); gt = synth.ground_truth_pose()
    lo, hi = synth.workspace_limits(crop)
    cs = scene[oracle.pass_through(scene, lo, hi)]
    sk = cs[oracle.uniform_sampling(cs, 0.01)]; mk = model[oracle.uniform_sampling(model, 0.01)]
    ns, _ = oracle.normals_knn(sk, 30); nm, _ = oracle.normals_knn(mk, 30)
    fs, _, _ = oracle.fpfh(sk, ns, 0.03); fm, _, _ = oracle.fpfh(mk, nm, 0.03)
    sub = cs[:: max(1, len(cs) // 30000)]
    p = oracle.default_icp_params(); p.max_iterations = 60; p.transformation_epsilon = 0.0; p.euclidean_fitness_epsilon = 0.0; p.mse_threshold_absolute = -1.0; p.acc_mode = 1
    ok = 0; errs = []
    for seed in seeds:
        T, err, it = oracle.sacia(mk, fm, sk, fs, seed=seed)
        r = oracle.icp(sub, model, p, guess=np.linalg.inv(T.astype(np.float64)).astype(np.float32))
        e = np.linalg.norm(r.T.astype(np.float64) - np.linalg.inv(gt))
        errs.append((round(float(np.linalg.norm(T - gt)), 2), round(float(e), 3), round(err, 3)))
        ok += e < 0.05
    print(f"{tag}: crop {crop} kp scene {len(sk)} model {len(mk)}  success {ok}/{len(list(seeds))}  {errs}", flush=True)

if __name__ == "__main__":
    run("current", 0.01)
    run("current", 0.003)
