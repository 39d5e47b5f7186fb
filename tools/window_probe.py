"""Developer probe: kernel time of every accumulate launch over the first iterations of C3 from a coarse pose 0.13 away
from the generator's (what bench.py --steps 20 --warmup 5 times: launches 5..24)."""
import importlib, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
ope = importlib.import_module("object-pose-estimation_amd")
synth = importlib.import_module("object-pose-estimation_amd.synth")
scene, model = synth.config_clouds("C3")
gt_inv = np.linalg.inv(synth.ground_truth_pose())
a = np.deg2rad(4.0)
P = np.eye(4); P[:3, :3] = [[np.cos(a), -np.sin(a), 0], [np.sin(a), np.cos(a), 0], [0, 0, 1]]; P[:3, 3] = [0.004, -0.003, 0.002]
guess = (P @ gt_inv).astype(np.float32)
label = " ".join(f"{k[4:]}={v}" for k, v in sorted(os.environ.items()) if k.startswith("OPE_")) or "defaults"
print(f"[{label}] |guess - gt| = {np.linalg.norm(guess - gt_inv):.3f}")
ctx = ope.Context(0)
cs = ctx.upload(scene); ix = ctx.build_index(ctx.upload(model))
for rep in range(2):
    ctx.icp_begin(cs, ix, ope.default_icp_params(max_iterations=200, mse_threshold_absolute=-1.0, check_every=0), guess)
    ts = []
    for it in range(70):
        ctx.icp_profile(1); ctx.icp_iterate(1); ctx.sync()
        km, kn = ctx.icp_profile_read(); ts.append(km * 1e3)
    ctx.icp_end()
    print(f"[{label}] kernel us per launch: " + " ".join(f"{t:.0f}" for t in ts))
    print(f"[{label}] mean launches 5..24: {np.mean(ts[5:25]):.1f} us, 10..69: {np.mean(ts[10:70]):.1f} us", flush=True)
ctx.close()
