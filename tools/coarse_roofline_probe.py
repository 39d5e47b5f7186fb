"""Coarse-stage kernels at a size where their roofline means something: the 49.6 k key points of the C3 frame (uniform sampling 0.01
of the raw 1 M-point frame, what tests/test_gpu_c3.py drives them with): normals k = 30 and FPFH r = 0.03, HIP-event times of the
launches (ope_profile_kernels) against the algorithmic bytes of SURVEY 8d; also the 1 M-point frame's normals (k = 12, BuildModel's).
    python tools/coarse_roofline_probe.py > profiles/rN_coarse_kernels_49k.txt"""
import importlib, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
ope = importlib.import_module("object-pose-estimation_amd")
synth = importlib.import_module("object-pose-estimation_amd.synth")
scene, model = synth.config_clouds("C3")
ctx = ope.Context(0)
frame = ctx.upload(scene)
keys, _ = ctx.uniform_sampling_cloud(frame, 0.01)
print(f"C3 frame: {len(scene)} points -> {keys.n} key points (leaf 0.01)")
def run(label, fn):
    fn()                                   # warm (loads the kernels)
    ctx.profile_kernels(True); fn(); t = ctx.profile_kernels_read(); ctx.profile_kernels(False)
    for name, rec in sorted(t.items()):
        gbs = rec["algorithmic_bytes"] / (rec["ms"] * 1e-3) / 1e9 if rec["ms"] > 0 else 0.0
        print(f"{label:34s} {name:28s} {rec['launches']:3d} launches {rec['ms']:8.3f} ms  {rec['algorithmic_bytes'] / 1e6:9.2f} MB algorithmic -> {gbs:7.1f} GB/s = {gbs / 8000:.4f} of 8 TB/s")
run("key points: normals k=30", lambda: ctx.normals(keys, 30, fetch=False))
run("key points: FPFH r=0.03", lambda: ctx.fpfh(keys, 0.03))
run("frame (1 M): normals k=12", lambda: ctx.normals(frame, 12, fetch=False))
run("frame: outlier removal meanK=30", lambda: ctx.statistical_outlier_removal_cloud(frame, 30, 1.0))
ctx.close()
