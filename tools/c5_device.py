"""Developer probe / profile target: config C5 (F frames of N points) through the device-resident BuildModel loop."""
import importlib, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
ope = importlib.import_module("object-pose-estimation_amd")
if os.environ.get("PROBE_LIB"):   # an A/B build of the library (make VARIANT=...)
    ope.LIB_PATH = os.path.join(os.path.dirname(ope.LIB_PATH), f"libope_hip_{os.environ['PROBE_LIB']}.so")
synth = importlib.import_module("object-pose-estimation_amd.synth")
bm = importlib.import_module("object-pose-estimation_amd.buildmodel")
F = int(sys.argv[1]) if len(sys.argv) > 1 else 8
N = int(sys.argv[2]) if len(sys.argv) > 2 else 500_000
cache = f"/tmp/c5_frames_{F}_{N}.npy"   # (6 s of CPU per frame: kept between the runs of one session)
if os.path.exists(cache):
    frames = list(np.load(cache, mmap_mode="r"))
else:
    frames = synth.frame_views(F, N, n_azimuths=32, workers=min(12, os.cpu_count() or 1))   # (child interpreters: safe in a script without a __main__ guard)
    np.save(cache, np.stack(frames))
ctx = ope.Context(0)
ctx.profile_kernels(True)
t0 = time.perf_counter()
res = bm.register_point_clouds(ope, ctx, frames)
dt = time.perf_counter() - t0
print(f"C5 {F} x {N}: {dt:.2f} s, iterations per pair {[p.iterations for p in res.pairs]}", flush=True)
for name, k in sorted(ctx.profile_kernels_read().items(), key=lambda kv: -kv[1]["ms"]):
    print(f"   {name:28s} {k['launches']:6d} launches {k['ms']:9.2f} ms  {k['algorithmic_bytes']/1e9:8.3f} GB algorithmic -> {k['algorithmic_bytes']/max(k['ms'],1e-9)/1e6:8.1f} GB/s")
ctx.close()
