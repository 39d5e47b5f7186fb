"""Developer probe: the chain floor.  Only the clutter points of the C3 frame (91 k queries: at most one chunk per resident
wave), tree kernel, steady state from the generator's pose: how long does a launch take whose every chunk is a far chunk and
no wave shares its SIMD with more than one other?  Also the surface points alone, and both at a 1/8 shard's size."""
import importlib, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
ope = importlib.import_module("object-pose-estimation_amd")
if os.environ.get("PROBE_LIB"):   # an A/B build: make -C object-pose-estimation_amd VARIANT=<name> ...
    ope.LIB_PATH = os.path.join(os.path.dirname(ope.LIB_PATH), f"libope_hip_{os.environ['PROBE_LIB']}.so")
synth = importlib.import_module("object-pose-estimation_amd.synth")
label = " ".join([f"lib={os.environ['PROBE_LIB']}"] * bool(os.environ.get("PROBE_LIB")) + [f"{k[4:]}={v}" for k, v in sorted(os.environ.items()) if k.startswith("OPE_")]) or "defaults"
tgt = synth.model_surface(100_000, 1)
gt = np.linalg.inv(synth.ground_truth_pose()).astype(np.float32)
raw = synth.scene_cloud(1_000_000, shuffle=False)
n_surf = int(round(1_000_000 / 1.1))
rng = np.random.default_rng(5)
cases = {"clutter 91k": raw[n_surf:], "clutter 11k (1/8)": raw[n_surf:][rng.permutation(1_000_000 - n_surf)[:11364]],
         "surface 909k": raw[:n_surf], "surface 114k (1/8)": raw[:n_surf][rng.permutation(n_surf)[:113636]],
         "mix 125k (1/8)": raw[rng.permutation(1_000_000)[:125000]]}
# the pose the full frame's ICP settles on (the clutter's pull included), so that the surface points sit where they sit in the bench
ctx = ope.Context(0)
ix = ctx.build_index(ctx.upload(tgt), grid=0)
T_settled = ctx.icp(ctx.upload(raw), ix, ope.default_icp_params(max_iterations=60, mse_threshold_absolute=-1.0, check_every=0), gt).T
ctx.close()
for name, src in cases.items():
    res = []
    for r in range(2):
        ctx = ope.Context(0)
        cs = ctx.upload(src); ix = ctx.build_index(ctx.upload(tgt), grid=int(os.environ.get("PROBE_GRID", "0")))
        p = ope.default_icp_params(max_iterations=1, mse_threshold_absolute=-1.0, check_every=0)
        # frozen pose: every launch searches with the same transform (max_iterations = 1 would end the run: use the step-wise API without update)
        p = ope.default_icp_params(max_iterations=10_000, mse_threshold_absolute=-1.0, check_every=0)
        ctx.icp_begin(cs, ix, p, T_settled)
        for _ in range(40):
            ctx.icp_accumulate()
        ctx.sync()
        ctx.icp_profile(100)
        t0 = time.perf_counter()
        for _ in range(100):
            ctx.icp_accumulate()
        ctx.sync(); dt = time.perf_counter() - t0
        km, kn = ctx.icp_profile_read()
        ctx.icp_end(); ctx.close()
        res.append(km / kn * 1e3)
    print(f"[{label}] {name:20s}: kernel us " + " ".join(f"{k:6.1f}" for k in res), flush=True)
