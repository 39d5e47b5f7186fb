"""Developer probe: what a FAR query (clutter) costs in each search kernel, apart from the mix.
Scenes from the C3 generator, from the ground-truth pose: the mix, the surface points alone, and the clutter points alone
ten times over (jittered by 1 mm) so that the launch fills the GPU like the other two."""
import importlib, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
ope = importlib.import_module("object-pose-estimation_amd")
synth = importlib.import_module("object-pose-estimation_amd.synth")
tgt = synth.model_surface(100_000, 1)
guess = np.linalg.inv(synth.ground_truth_pose()).astype(np.float32)
raw = synth.scene_cloud(1_000_000, shuffle=False)
n_surf = int(round(1_000_000 / 1.1))
rng = np.random.default_rng(5)
near = raw[:n_surf]
clut = raw[n_surf:]
far10 = (np.repeat(clut, 10, axis=0) + rng.standard_normal((10 * len(clut), 3)).astype(np.float32) * 1e-3).astype(np.float32)
mix = raw[rng.permutation(len(raw))]
label = os.environ.get("OPE_NO_PACKET") and "nopacket" or "packet"
for sname, src in (("mix", mix), ("near", near), ("far10", far10)):
    for name, kw in (("tree", dict(grid=0)), ("grid", dict(grid=2))):
        ctx = ope.Context(0)
        cs = ctx.upload(src); ix = ctx.build_index(ctx.upload(tgt), **kw)
        p = ope.default_icp_params(max_iterations=141, mse_threshold_absolute=-1.0, check_every=0)
        ctx.icp_begin(cs, ix, p, guess)
        ctx.icp_iterate(40); ctx.sync()
        ctx.icp_profile(100)
        t0 = time.time(); ctx.icp_iterate(100); ctx.sync(); dt = time.time() - t0
        km, kn = ctx.icp_profile_read()
        out = ctx.icp_end()
        print(f"[{label}] {sname:6s} n {len(src):8d} {name:5s}: {dt/100*1e6:7.1f} us/iteration  kernel {km/kn*1e3:7.1f} us", flush=True)
        ctx.close()
