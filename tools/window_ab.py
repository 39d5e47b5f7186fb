"""Developer probe (DEVELOPER=1 variant build): the driver's window on the C3 frame — launches 5-24 from a pose a few degrees off
— per launch (HIP events) and per step (wall clock), plus the steady state; for whatever OPE_* switches the environment sets."""
import importlib, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
ope = importlib.import_module("object-pose-estimation_amd")
if os.environ.get("PROBE_LIB"):
    ope.LIB_PATH = os.path.join(os.path.dirname(ope.LIB_PATH), f"libope_hip_{os.environ['PROBE_LIB']}.so")
synth = importlib.import_module("object-pose-estimation_amd.synth")
label = " ".join(f"{k[4:]}={v}" for k, v in sorted(os.environ.items()) if k.startswith("OPE_")) or "defaults"
tgt = synth.model_surface(100_000, 1)
gt = np.linalg.inv(synth.ground_truth_pose()).astype(np.float32)
src = synth.scene_cloud(1_000_000, clutter_frac=0.10)
a = np.deg2rad(3.0)
R = np.array([[np.cos(a), -np.sin(a), 0], [np.sin(a), np.cos(a), 0], [0, 0, 1]])
T0 = np.eye(4); T0[:3, :3] = R; T0[:3, 3] = [0.004, -0.003, 0.002]
start = (T0 @ gt.astype(np.float64)).astype(np.float32)
grid = int(os.environ.get("PROBE_GRID", "0"))
for rep in range(int(sys.argv[1]) if len(sys.argv) > 1 else 2):
    ctx = ope.Context(0)
    cs = ctx.upload(src); ix = ctx.build_index(ctx.upload(tgt), grid=grid)
    p = ope.default_icp_params(max_iterations=400, mse_threshold_absolute=-1.0, check_every=0)
    ctx.icp_begin(cs, ix, p, start)
    ctx.icp_iterate(5); ctx.sync()
    ctx.icp_profile(20)
    t0 = time.perf_counter(); ctx.icp_iterate(20); ctx.sync(); dt = time.perf_counter() - t0
    ms = ctx.icp_profile_launches()
    ctx.icp_profile(0)
    ctx.icp_iterate(35); ctx.sync()
    ctx.icp_profile(100)
    t0 = time.perf_counter(); ctx.icp_iterate(100); ctx.sync(); dts = time.perf_counter() - t0
    ms2 = ctx.icp_profile_launches()
    T = ctx.icp_current_transform()
    ctx.icp_end(); ctx.close()
    print(f"[{label}] window: step {dt/20*1e6:6.1f} us, kernel {ms.mean()*1e3:6.1f} us | " + " ".join(f"{v*1e3:4.0f}" for v in ms)
          + f" || steady (launches 60-159): step {dts/100*1e6:6.1f} us, kernel {ms2.mean()*1e3:6.1f} us (max {ms2.max()*1e3:.0f}) | pose err {np.linalg.norm(np.asarray(T, np.float64) - gt):.4f}", flush=True)
