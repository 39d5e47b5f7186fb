"""Developer probe: BuildModel registration loop (config C5 shape) on one GPU: F frames of N points each."""
import importlib, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
ope = importlib.import_module("object-pose-estimation_amd")
synth = importlib.import_module("object-pose-estimation_amd.synth")
bm = importlib.import_module("object-pose-estimation_amd.buildmodel")
F = int(sys.argv[1]) if len(sys.argv) > 1 else 4
N = int(sys.argv[2]) if len(sys.argv) > 2 else 500_000
t0 = time.perf_counter(); frames, poses = synth.frame_views(F, N, return_poses=True, n_azimuths=32); t1 = time.perf_counter()
print(f"generated {F} frames x {N} points in {t1-t0:.1f} s", flush=True)
ctx = ope.Context(0)
acc = frames[0]
for i in range(F - 1):
    t0 = time.perf_counter()
    aligned, pr = bm.get_icp_normal(ope, ctx, acc, frames[i + 1], 0.7, 500)
    dt = time.perf_counter() - t0
    want = poses[i + 1] @ np.linalg.inv(poses[i]) if i == 0 else None
    err = f" |T - generator| = {np.linalg.norm(pr.T - want):.4f}" if want is not None else ""
    print(f"pair {i}: source {len(acc)} target {len(frames[i+1])}: {dt*1e3:.0f} ms, {pr.iterations} iterations, converged {pr.converged}, fitness {pr.fitness:.3e}{err}", flush=True)
    acc = np.concatenate([aligned, frames[i + 1]], axis=0)
ctx.close()
