"""Developer probe: host wall-clock of the front-end stages on the C3 frame, first and later calls (allocations, lazy kernel
loading and host round trips are what these stages are made of: the kernels themselves take < 5 ms in all)."""
import importlib, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
ope = importlib.import_module("object-pose-estimation_amd")
if os.environ.get("PROBE_LIB"):
    ope.LIB_PATH = os.path.join(os.path.dirname(ope.LIB_PATH), f"libope_hip_{os.environ['PROBE_LIB']}.so")
synth = importlib.import_module("object-pose-estimation_amd.synth")
scene = synth.scene_cloud(1_000_000)
model = synth.model_surface(100_000, 1)
ctx = ope.Context(0)
lo_w, hi_w = synth.workspace_limits(0.01)
def t(label, fn):
    ctx.sync(); t0 = time.perf_counter(); r = fn(); ctx.sync(); print(f"   {label:34s} {(time.perf_counter()-t0)*1e3:8.2f} ms", flush=True); return r
for rep in range(3):
    print(f"-- pass {rep}")
    frame = t("upload 1M", lambda: ctx.upload(scene))
    crop, _ = t("pass_through_cloud", lambda: ctx.pass_through_cloud(frame, lo_w, hi_w))
    clus, _ = t("statistical_outlier_removal_cloud", lambda: ctx.statistical_outlier_removal_cloud(crop, 30, 1.0))
    kc, _ = t("uniform_sampling_cloud 0.01", lambda: ctx.uniform_sampling_cloud(clus, 0.01))
    t("normals k=30 (858 pts)", lambda: ctx.normals(kc, 30, fetch=False))
    f = t("fpfh r=0.03", lambda: ctx.fpfh(kc, 0.03))
    t("index over the cluster", lambda: ctx.build_index(clus))
ctx.close()
