"""Timing probe: normals (k = 12, 30) and statistical outlier removal of large clouds against their own index, HIP-event kernel times."""
import importlib, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
ope = importlib.import_module("object-pose-estimation_amd")
if os.environ.get("PROBE_LIB"):
    ope.LIB_PATH = os.path.join(os.path.dirname(ope.LIB_PATH), f"libope_hip_{os.environ['PROBE_LIB']}.so")
synth = importlib.import_module("object-pose-estimation_amd.synth")
ctx = ope.Context(0)
scene = synth.scene_cloud(1_000_000)
frames = synth.frame_views(4, 500_000, n_azimuths=32)
big = np.concatenate(frames)
for name, P in (("scene 1 M", scene), ("4 views 2 M", big)):
    c = ctx.upload(P)
    for rep in range(2):
        ctx.profile_kernels(True)
        ctx.normals(c, 12, fetch=False); ctx.normals(c, 30, fetch=False)
        keep = ctx.statistical_outlier_removal(c, 30, 1.0)
        k = ctx.profile_kernels_read()
        print(name, {kk: round(v["ms"], 3) for kk, v in k.items() if "normals" in kk or "sor" in kk}, len(keep), flush=True)
        ctx.profile_kernels(False)
