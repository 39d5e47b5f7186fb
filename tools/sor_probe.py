"""Developer probe: the outlier filter of the front end (device-resident form) on the C3 frame, wall-clock per call; run it under
rocprofv3 --kernel-trace --stats to see which launches the call is made of."""
import importlib, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
ope = importlib.import_module("object-pose-estimation_amd")
synth = importlib.import_module("object-pose-estimation_amd.synth")
scene = synth.scene_cloud(1_000_000)
lo, hi = synth.workspace_limits()
ctx = ope.Context(0)
c = ctx.upload(scene)
c2, _ = ctx.pass_through_cloud(c, lo, hi, want_idx=True)
for rep in range(4):
    ctx.sync(); t0 = time.perf_counter()
    c3 = ctx.statistical_outlier_removal_cloud(c2, 30, 1.0)
    c3 = c3[0] if isinstance(c3, tuple) else c3
    ctx.sync(); dt = time.perf_counter() - t0
    print(f"outlier removal of {c2.n} points -> {c3.n}: {dt * 1e3:.2f} ms", flush=True)
    c3.free()
ctx.close()
